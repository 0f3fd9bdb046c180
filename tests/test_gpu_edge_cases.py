"""Edge cases of the block system on the GPU against the oracle: ragged / empty rows
(generic-width kernels), rectangular blocks, ConstantNullspace and FullNullspace, mixed
boundary sets (automatic un-sharing of value arrays), the CN sub-block split, value
updates, argument errors."""
import numpy as np
import pytest
import scipy.sparse as sp

import common
from control_amd.multiblock import (ConstantNullspace, DirichletBCNullspace, FullNullspace,
                                    MultiBlockSystem, NoneNullspace)
from oracle import kkt_oracle as ko

pytestmark = pytest.mark.gpu


def ragged(n_rows, n_cols, seed, density=0.02, empty_every=7, heavy_every=50):
    """Random CSR with empty rows and a few very long rows (slice widths far from uniform)."""
    rng = np.random.default_rng(seed)
    A = sp.random(n_rows, n_cols, density=density, random_state=rng, format="lil")
    for r in range(0, n_rows, empty_every):
        A.rows[r], A.data[r] = [], []
    for r in range(3, n_rows, heavy_every):
        cols = np.sort(rng.choice(n_cols, size=min(n_cols, 40), replace=False))
        A.rows[r], A.data[r] = list(cols), list(rng.standard_normal(len(cols)))
    A = sp.csr_matrix(A)
    A.sort_indices()
    return A


def pair(nx0, nx1, blocks, n0, n1, ns0, ns1, ons0, ons1, **kw):
    g = MultiBlockSystem(nx0, nx1, *blocks, n_blocks_00=n0, n_blocks_11=n1, nullspace_0=ns0,
                         nullspace_1=ns1, **kw)
    okw = {k: v for k, v in kw.items() if k not in ("device", "options")}
    o = ko.OracleSystem(nx0, nx1, *blocks, n_blocks_00=n0, n_blocks_11=n1, nullspace_0=ons0,
                        nullspace_1=ons1, **okw)
    return g, o


@pytest.mark.parametrize("sell_r", ["2", "1"])
def test_ragged_rectangular_blocks_with_mixed_nullspaces(sell_r):
    """2 + 3 blocks of different sizes, rectangular couplings, every nullspace kind, empty
    rows, rows of 40 entries among rows of 3 (no uniform width -> generic kernels)."""
    nx0, nx1, n0, n1 = 301, 157, 2, 3
    b00 = {(i, j): None for i in range(n0) for j in range(n0)}
    b01 = {(i, j): None for i in range(n0) for j in range(n1)}
    b10 = {(i, j): None for i in range(n1) for j in range(n0)}
    b11 = {(i, j): None for i in range(n1) for j in range(n1)}
    b00[(0, 0)] = ragged(nx0, nx0, 1)
    b00[(1, 0)] = ragged(nx0, nx0, 2)
    b00[(1, 1)] = b00[(0, 0)]                      # shared object
    b01[(0, 1)] = ragged(nx0, nx1, 3)
    b01[(1, 2)] = ragged(nx0, nx1, 4)
    b10[(0, 0)] = ragged(nx1, nx0, 5)
    b10[(2, 1)] = ragged(nx1, nx0, 6)
    b11[(1, 1)] = ragged(nx1, nx1, 7)
    b11[(2, 0)] = ragged(nx1, nx1, 8)               # block row 0 of variable 1 has no 11 term
    bc0 = np.array([0, 5, 17, 300])
    bc1 = np.array([1, 2, 156])
    g, o = pair(nx0, nx1, (b00, b01, b10, b11), n0, n1,
                (DirichletBCNullspace(bc0, alpha=2.5), NoneNullspace()),
                (ConstantNullspace(alpha=0.5), DirichletBCNullspace(bc1), FullNullspace()),
                (ko.DirichletBCNullspace(bc0, alpha=2.5), ko.NoneNullspace()),
                (ko.ConstantNullspace(alpha=0.5), ko.DirichletBCNullspace(bc1),
                 ko.FullNullspace()), options={"sell_r": sell_r})
    for seed in range(3):
        x = common.rng_vector(o.N, 100 + seed)
        assert common.rel_err(g.mult(x), o.mult(x)) < 1e-13

    def pc(u_0, u_1, b_0, b_1):
        u_0[:] = 3.0 * b_0
        u_1[:] = b_1[::-1]
    x = common.rng_vector(o.N, 7)
    assert common.rel_err(g.pc_apply(x, pc), o.pc_apply(pc, x)) < 1e-14
    assert common.rel_err(g.pc_apply(x, None),
                          o.pc_apply(lambda a, b, c, d: (a.__setitem__(slice(None), c),
                                                         b.__setitem__(slice(None), d)),
                                     x)) < 1e-14


def test_shared_values_with_different_boundary_sets_are_unshared():
    """One matrix object used in two block columns whose Dirichlet sets differ: the column
    masks differ, so the library must give the blocks separate value arrays."""
    nx = 200
    A = ragged(nx, nx, 11, density=0.05, empty_every=1000)
    b00 = {(0, 0): A, (0, 1): A, (1, 0): None, (1, 1): A}
    z = {(i, j): None for i in range(2) for j in range(1)}
    zz = {(0, 0): None}
    bca, bcb = np.arange(0, 20), np.arange(100, 140)
    g = MultiBlockSystem(nx, 3, b00, z, {(0, 0): None, (0, 1): None}, zz, n_blocks_00=2,
                         n_blocks_11=1, nullspace_0=(DirichletBCNullspace(bca),
                                                     DirichletBCNullspace(bcb)))
    o = ko.OracleSystem(nx, 3, b00, z, {(0, 0): None, (0, 1): None}, zz, n_blocks_00=2,
                        n_blocks_11=1, nullspace_0=(ko.DirichletBCNullspace(bca),
                                                    ko.DirichletBCNullspace(bcb)))
    x = common.rng_vector(o.N, 3)
    assert common.rel_err(g.mult(x), o.mult(x)) < 1e-13
    assert g.info()["n_value_arrays"] == 1          # as added ...
    assert g.info()["bytes_device_values"] > g.info()["nnz_blocks"] // 3 * 8   # ... then cloned


def test_cn_sub_block_split():
    """CN with sub_n_blocks (the incompressible layout, preconditioner.py:471-525): T_1 on the
    first part of variable 0 and the second part of variable 1, T_2 on the others."""
    nx0, nx1, n0, n1 = 90, 40, 6, 4
    rngA = [ragged(nx0, nx0, 20 + k, density=0.06, empty_every=1000) for k in range(n0)]
    rngB = [ragged(nx1, nx1, 40 + k, density=0.08, empty_every=1000) for k in range(n1)]
    b00 = {(i, j): (rngA[i] if i == j else None) for i in range(n0) for j in range(n0)}
    b11 = {(i, j): (rngB[i] if i == j else None) for i in range(n1) for j in range(n1)}
    b01 = {(i, j): None for i in range(n0) for j in range(n1)}
    b10 = {(i, j): None for i in range(n1) for j in range(n0)}
    b01[(2, 1)] = ragged(nx0, nx1, 60)
    b10[(3, 5)] = ragged(nx1, nx0, 61)
    bc = np.array([0, 1, 2])
    g, o = pair(nx0, nx1, (b00, b01, b10, b11), n0, n1,
                tuple(DirichletBCNullspace(bc) for _ in range(n0)),
                tuple(ConstantNullspace() for _ in range(n1)),
                tuple(ko.DirichletBCNullspace(bc) for _ in range(n0)),
                tuple(ko.ConstantNullspace() for _ in range(n1)),
                CN=True, sub_n_blocks_00_0=3, sub_n_blocks_11_0=2)
    x = common.rng_vector(o.N, 5)
    assert common.rel_err(g.mult(x), o.mult(x)) < 1e-13


def test_update_block_values_and_rebuilt_preconditioner():
    """Picard-style re-linearisation: new values on the stored structure, operator and
    built-in preconditioner follow (kkt_update_block_values)."""
    p = common.heat_problem(n=10, n_t=6, share=False, time_dependent=True)
    q = common.heat_problem(n=10, n_t=6, share=False, time_dependent=False)
    gsys = common.gpu_system(p)
    mass, schur = (20, 0.5, 2.0), (10, 0.2, 2.1)
    gpc = common.gpu_pc(p, mass, schur)
    x = common.rng_vector(2 * p["m"] * p["sd"].n_dofs)
    gsys.pc_apply(x, gpc)                                  # builds the Schur matrices
    for quad in range(4):
        for (i, j), A in q["blocks"][quad].items():
            if A is not None:
                gsys.update_block_values(quad, i, j, A)
    osys = common.oracle_system(q)
    assert common.rel_err(gsys.mult(x), osys.mult(x)) < 1e-13
    assert common.rel_err(gsys.pc_apply(x, gpc),
                          osys.pc_apply(common.oracle_pc(q, mass, schur), x)) < 1e-10


def test_argument_errors():
    from control_amd import _lib
    nx = 20
    A = sp.identity(nx, format="csr")
    with pytest.raises(ValueError, match="Unexpected dimension of blocks"):
        MultiBlockSystem(nx, nx, {(0, 0): A}, {}, {(0, 0): A}, {(0, 0): A})
    B = sp.identity(nx + 1, format="csr")
    with pytest.raises(_lib.KktError, match="shape"):
        MultiBlockSystem(nx, nx, {(0, 0): B}, {(0, 0): None}, {(0, 0): None}, {(0, 0): None})
    with pytest.raises(_lib.KktError, match="out of range"):
        MultiBlockSystem(nx, nx, {(0, 0): A}, {(0, 0): None}, {(0, 0): None}, {(0, 0): None},
                         nullspace_0=(DirichletBCNullspace([nx + 3]),))
    g = MultiBlockSystem(nx, nx, {(0, 0): A}, {(0, 0): None}, {(0, 0): None}, {(0, 0): A})
    with pytest.raises(ValueError, match="linear_solver"):
        g.solve(np.zeros((1, nx)), np.zeros((1, nx)), np.ones((1, nx)), np.ones((1, nx)),
                solver_parameters={"linear_solver": "bicg", "relative_tolerance": 1e-8,
                                   "absolute_tolerance": 0.0})
    with pytest.raises(KeyError):      # relative_tolerance is required (preconditioner.py:739)
        g.solve(np.zeros((1, nx)), np.zeros((1, nx)), np.ones((1, nx)), np.ones((1, nx)),
                solver_parameters={})
    # identity system: one iteration, exact
    u0, u1 = np.zeros((1, nx)), np.zeros((1, nx))
    r = g.solve(u0, u1, np.ones((1, nx)), 2 * np.ones((1, nx)),
                solver_parameters={"relative_tolerance": 1e-12, "absolute_tolerance": 0.0,
                                   "monitor_convergence": False})
    assert r.reason > 0 and np.allclose(u0, 1.0) and np.allclose(u1, 2.0)


def test_full_size_operator_properties():
    """BASELINE configs[1] at full size (8.45 M unknowns, mode S to keep host assembly short):
    size-independent properties instead of an oracle run -- linearity, symmetry of the BE
    KKT operator (block_01 = block_10^T), boundary rows, and bitwise repeatability."""
    p = common.heat_problem(n=256, n_t=64)
    g = common.gpu_system(p)
    N = 2 * p["m"] * p["sd"].n_dofs
    x, y = common.rng_vector(N, 1), common.rng_vector(N, 2)
    Ax, Ay = g.mult(x), g.mult(y)
    assert np.array_equal(Ax, g.mult(x))
    a, b = 0.37, -1.9
    assert common.rel_err(g.mult(a * x + b * y), a * Ax + b * Ay) < 1e-13
    X = x.reshape(2 * p["m"], -1).copy()
    Y = y.reshape(2 * p["m"], -1).copy()
    X[:, p["nodes"]] = 0.0
    Y[:, p["nodes"]] = 0.0
    AX, AY = g.mult(X.ravel()), g.mult(Y.ravel())
    assert abs(X.ravel() @ AY - Y.ravel() @ AX) < 1e-10 * abs(X.ravel() @ AY)
    assert np.array_equal(Ax.reshape(2 * p["m"], -1)[:, p["nodes"]],
                          x.reshape(2 * p["m"], -1)[:, p["nodes"]])


def test_full_size_solve_reduces_true_residual_and_is_reproducible():
    p = common.heat_problem(n=256, n_t=64, beta=1e-2)
    g = common.gpu_system(p)
    m, nx = p["m"], p["sd"].n_dofs
    pc = common.gpu_pc(p, (20, 0.5, 2.0), (8, 0.07, 2.1))
    X = p["sd"].coords
    xs = np.stack([np.sin(np.pi * X[:, 0]) * np.sin(np.pi * X[:, 1]) * (1 + 0.01 * k)
                   for k in range(2 * m)])
    b = g.mult(xs.ravel()).reshape(2 * m, nx)
    sp_ = {"linear_solver": "fgmres", "gmres_restart": 10, "maximum_iterations": 20,
           "relative_tolerance": 1e-6, "absolute_tolerance": 0.0,
           "monitor_convergence": False, "preconditioner": True}
    runs = []
    for _ in range(2):
        u0, u1 = np.zeros((m, nx)), np.zeros((m, nx))
        r = g.solve(u0, u1, b[:m].copy(), b[m:].copy(), solver_parameters=sp_, pc_fn=pc)
        runs.append((np.vstack([u0, u1]), r.history))
    assert np.array_equal(runs[0][0], runs[1][0])              # deterministic reductions
    assert np.array_equal(runs[0][1], runs[1][1])
    u = runs[0][0]
    bc = b.copy()
    bc[:, p["nodes"]] = 0.0
    res = np.linalg.norm(bc.ravel() - g.mult(u.ravel()))
    assert res < 0.9 * np.linalg.norm(bc)                      # true residual really went down
    h = runs[0][1]
    assert abs(h[-1] - res) < 1e-6 * h[0]                      # monitored norm is the true one


def test_row_sorted_storage_is_bit_identical():
    """Structures with rows of very different lengths (P2: 9 or 19 non-zeros) are stored
    row-sorted inside windows of 8 slices (SELL-C-sigma); every row keeps its own fma chain,
    so operator and preconditioner are bit-identical to the unsorted storage -- with fewer
    padded slots."""
    outs, infos = [], []
    for flag in ("0", "1"):
        p = common.stokes_problem(n=8, n_t=4)
        outer, gpc = common.stokes_gpu(p, options={"sell_sort": flag})
        x = common.rng_vector(outer.info()["n_local"])
        outs.append((outer.mult(x), outer.pc_apply(x, gpc)))
        infos.append(outer.info())
    assert np.array_equal(outs[0][0], outs[1][0])
    assert np.array_equal(outs[0][1], outs[1][1])
    assert infos[1]["bytes_device_values"] < 0.85 * infos[0]["bytes_device_values"]


@pytest.mark.parametrize("CN", [False, True])
def test_width_switched_ragged_operator_is_bit_identical(CN):
    """Ragged structures in the operator apply (P2 velocity blocks: slices of 19 / 9 entries, the
    rectangular Stokes blocks: 7 / 4) run kkt_spmv_rows_ragged, where a wave picks the body
    unrolled for its slice's width; slices of any other width take the slot loop inside the same
    kernel.  Every row keeps its fma chain: the result equals the slot-loop kernel's
    ("ragged_switch" = "0") bit for bit, and the oracle's to round-off.  32 x 32 Taylor-Hood:
    interior windows of full 19- and 9-wide slices, boundary windows of other widths."""
    p = common.stokes_problem(n=32, n_t=4, CN=CN)
    outs = []
    for flag, xcd in (("1", "1"), ("0", "1"), ("1", "0")):   # "ragged_xcd": XCD-aware workgroup order
        outer, _ = common.stokes_gpu(p, options={"ragged_switch": flag, "ragged_xcd": xcd})
        info = outer.info()
        assert info["apply_switched"] == (info["apply_launches"] if flag == "1" else 0)
        x = common.rng_vector(info["n_local"])
        outs.append(outer.mult(x))
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    osys = common.stokes_oracle(p)[0]
    assert common.rel_err(outs[0], osys.mult(x)) < 1e-13


def test_mid_size_mesh_long_sweep_program_matches_plain_launches():
    """A mesh with more than 1 024 slices (401^2 nodes: 1 257) runs the data-flow sweep program
    with multi-wave workgroups; with 4-wave workgroups, two to a CU, programs of about 1 000
    phases timed out waiting for a neighbour (error -2).  The shapes in use -- one-wave
    workgroups, or one 8-wave workgroup per CU -- must reproduce the plain launches bit for
    bit on a program of 972 phases."""
    import bench

    class A:
        workload, n, n_t, beta, T, scheme, mode = "heat2d", 400, 12, 1e-4, 2.0, "BE", "G"
        schur_its, schur_emin, schur_emax = 80, 7e-4, 2.1
    p = bench.build_problem(A())
    x = common.rng_vector(2 * p["m"] * p["sd"].n_dofs)
    g = common.gpu_system(p)
    y = g.pc_apply(x, common.gpu_pc(p, p["mass"], p["schur"]))
    g2 = common.gpu_system(p, options={"persistent": "0"})
    y2 = g2.pc_apply(x, common.gpu_pc(p, p["mass"], p["schur"]))
    assert np.array_equal(y, y2)


def test_sweep_program_timeout_falls_back_to_plain_launches():
    """A persistent sweep program that gives up waiting for a neighbour (bounded spins) must not
    fail the call: the preconditioner is rebuilt as plain launches and the work redone.  The test
    hook makes tile 0 skip one hand-off, so its neighbours time out."""
    p = common.heat_problem(n=96, n_t=6)
    schur = (6, 0.05, 2.1)
    x = common.rng_vector(2 * p["m"] * p["sd"].n_dofs)
    ref = common.gpu_system(p, options={"persistent": "0"}).pc_apply(
        x, common.gpu_pc(p, (20, 0.5, 2.0), schur))
    g = common.gpu_system(p, options={"persistent": "1", "prog_mode": "tile", "debug_drop_handoff": "3"})
    got = g.pc_apply(x, common.gpu_pc(p, (20, 0.5, 2.0), schur))
    assert np.array_equal(got, ref)
    assert g.info()["program_fallbacks"] == 1
    msg = g._lib.kkt_last_error(g.handle).decode()
    assert "tile form" in msg and "plain launches" in msg
    # and inside a solve: it starts over from the caller's guess
    g2 = common.gpu_system(p, options={"persistent": "1", "prog_mode": "tile", "debug_drop_handoff": "3"})
    m, nx = p["m"], p["sd"].n_dofs
    b = x.reshape(2 * m, nx)
    sp_ = {"linear_solver": "gmres", "gmres_restart": 10, "maximum_iterations": 8,
           "relative_tolerance": 0.0, "absolute_tolerance": 0.0, "monitor_convergence": False,
           "preconditioner": True}
    out = []
    for gs in (g2, common.gpu_system(p, options={"persistent": "0"})):
        u0, u1 = np.zeros((m, nx)), np.zeros((m, nx))
        gs.solve(u0, u1, b[:m].copy(), b[m:].copy(), solver_parameters=sp_,
                 pc_fn=common.gpu_pc(p, (20, 0.5, 2.0), schur))
        out.append(np.vstack([u0, u1]))
    assert np.array_equal(out[0], out[1])
    assert g2.info()["program_fallbacks"] == 1


def test_tiles_that_are_not_co_resident_are_detected_at_kernel_entry():
    """The tile programs need every workgroup of their grid running at once.  Each tile checks in
    on a counter when the kernel starts and waits a bounded ~5 ms for the others; if they do not
    all arrive (test hook: tile 0 never checks in -- what a masked-off CU or a busy device looks
    like to the others) every tile leaves at once, the host says so on stderr, rebuilds the
    preconditioner as plain launches and redoes the application: same result, one fall-back, in
    well under the seconds a spin time-out deep inside a sweep used to take."""
    import time
    p = common.heat_problem(n=96, n_t=6)
    schur = (6, 0.05, 2.1)
    x = common.rng_vector(2 * p["m"] * p["sd"].n_dofs)
    ref = common.gpu_system(p, options={"persistent": "0"}).pc_apply(
        x, common.gpu_pc(p, (20, 0.5, 2.0), schur))
    g = common.gpu_system(p, options={"persistent": "1", "prog_mode": "tile", "debug_drop_handoff": "-1"})
    pc = common.gpu_pc(p, (20, 0.5, 2.0), schur)
    g._set_pc(pc)
    assert g.info()["sweep_form"] == 3
    t0 = time.time()
    got = g.pc_apply(x, pc)
    dt = time.time() - t0
    assert np.array_equal(got, ref)
    inf = g.info()
    assert inf["program_fallbacks"] == 1 and inf["sweep_form"] == 0
    msg = g._lib.kkt_last_error(g.handle).decode()
    assert "not co-resident" in msg and "plain launches" in msg, msg
    assert dt < 1.0, dt


def test_zero_right_hand_side_with_nonzero_guess_converges_to_zero():
    """KSPConvergedDefault's special case: with b = 0 the norm of the first residual takes the
    place of the (zero) right-hand-side norm, so a non-zero initial guess is iterated towards
    u = 0 instead of being reported as diverged at the first test (gmres, fgmres and the oracle
    alike).  FGMRES monitors the true residual: its iterate really reaches 0; left-preconditioned
    GMRES stops on the preconditioned norm (the BE preconditioner scales the last level by
    1 / epsilon), so only the stopping behaviour is compared there."""
    p = common.heat_problem(n=10, n_t=6, beta=1e-2)
    m, nx = p["m"], p["sd"].n_dofs
    mass, schur = (20, 0.5, 2.0), (12, 0.08, 2.1)
    for ksp in ("gmres", "fgmres"):
        sp_ = {"linear_solver": ksp, "gmres_restart": 10, "maximum_iterations": 100,
               "relative_tolerance": 1e-8, "absolute_tolerance": 0.0, "monitor_convergence": False}
        outs = []
        for sysm, pc in ((common.gpu_system(p), common.gpu_pc(p, mass, schur)),
                         (common.oracle_system(p), common.oracle_pc(p, mass, schur))):
            u0 = common.rng_vector(m * nx, 3).reshape(m, nx)
            u1 = common.rng_vector(m * nx, 4).reshape(m, nx)
            r = sysm.solve(u0, u1, np.zeros((m, nx)), np.zeros((m, nx)), solver_parameters=sp_,
                           pc_fn=pc)
            assert r.reason > 0
            assert r.history[-1] <= 1e-8 * r.history[0]
            if ksp == "fgmres":
                assert max(np.abs(u0).max(), np.abs(u1).max()) < 1e-5
            outs.append(r.its)
        # (free-running BE trajectories of two implementations separate: 77 against 70 FGMRES
        # iterations measured; what is compared is that both stop on the same test)
        assert abs(outs[0] - outs[1]) <= max(2, 0.15 * max(outs))


def test_updating_one_of_two_shared_blocks_leaves_the_other_alone():
    """Blocks given as one object share a value array on the device.  The reference assembles
    every block on its own, so new values for one of them must not reach the other (copy on
    write); re-sending identical values changes nothing."""
    nx = 150
    A = ragged(nx, nx, 11) + sp.identity(nx, format="csr")
    A.sort_indices()
    B = A.copy()
    B.data = B.data * 1.5 + 0.25
    b00 = {(0, 0): A, (0, 1): None, (1, 0): None, (1, 1): A}
    none = {(i, j): None for i in range(2) for j in range(2)}
    g, o = pair(nx, nx, (b00, dict(none), dict(none), dict(none)), 2, 2,
                (NoneNullspace(), NoneNullspace()), (NoneNullspace(), NoneNullspace()),
                (ko.NoneNullspace(), ko.NoneNullspace()), (ko.NoneNullspace(), ko.NoneNullspace()))
    x = common.rng_vector(o.N, 5)
    assert g.info()["n_value_arrays"] == 1
    g.update_block_values(0, 0, 0, A)                      # identical values: still shared
    assert g.info()["n_value_arrays"] == 1
    g.update_block_values(0, 0, 0, B)                      # new values: block (0, 0) only
    assert g.info()["n_value_arrays"] == 2
    o2 = ko.OracleSystem(nx, nx, {(0, 0): B, (0, 1): None, (1, 0): None, (1, 1): A}, dict(none),
                         dict(none), dict(none), n_blocks_00=2, n_blocks_11=2)
    assert common.rel_err(g.mult(x), o2.mult(x)) < 1e-13


def test_sharded_handle_rejects_blocks_whose_halo_is_not_exchanged():
    """One halo per column variable travels on a heat-type shard (x0 of level lo-1, x1 of level
    hi): a block that would read the other side must be refused, not multiplied with the
    wrong vector."""
    import ctypes as C
    from control_amd import _lib
    lib = _lib.load()
    h = C.c_void_p()
    assert lib.kkt_create(C.byref(h), 0) == 0
    try:
        nx = 40
        A = sp.identity(nx, format="csr")
        ip, ix, va = (np.ascontiguousarray(A.indptr, dtype=np.int32),
                      np.ascontiguousarray(A.indices, dtype=np.int32),
                      np.ascontiguousarray(A.data, dtype=np.float64))
        assert lib.kkt_set_layout(h, 6, 6, nx, nx, 0, -1, -1) == 0
        assert lib.kkt_set_shard(h, 1, 3) == 0             # owns levels [2, 4)
        add = lambda q, i, j: lib.kkt_add_block(                                  # noqa: E731
            h, q, i, j, nx, nx, ip.ctypes.data_as(_lib.c_i32p), ix.ctypes.data_as(_lib.c_i32p),
            va.ctypes.data_as(_lib.c_f64p), -1)
        assert add(2, 2, 1) == 0                           # Q10: x0 of level lo - 1: exchanged
        assert add(1, 3, 4) == 0                           # Q01: x1 of level hi: exchanged
        assert add(2, 3, 4) == -1                          # Q10 reading x0 of level hi
        assert b"lo-1, not hi" in lib.kkt_last_error(h)
        assert add(1, 2, 1) == -1                          # Q01 reading x1 of level lo - 1
    finally:
        lib.kkt_destroy(h)


def test_tile_coordinates_are_a_checked_hint():
    """``kkt_set_tile_coordinates``: wrong shapes and non-finite values are refused; coordinates
    that are useless for tiling (all rows at one point) still give the plain launches' result."""
    p = common.heat_problem(n=24, n_t=4)
    nx = p["sd"].n_dofs
    g = common.gpu_system(p, tile_coordinates=False, options={"prog_mode": "tile"})
    with pytest.raises(ValueError):
        g.set_tile_coordinates(np.zeros((nx + 1, 2)))
    bad = np.array(p["sd"].coords, dtype=np.float64)
    bad[3, 0] = np.nan
    with pytest.raises(Exception, match="finite"):
        g.set_tile_coordinates(bad)
    g.set_tile_coordinates(np.zeros((nx, 2)))              # degenerate: ties broken by row index
    x = common.rng_vector(2 * p["m"] * nx)
    pc = (20, 0.5, 2.0), (9, 0.05, 2.1)
    got = g.pc_apply(x, common.gpu_pc(p, *pc))
    plain = common.gpu_system(p, options={"persistent": "0"}).pc_apply(x, common.gpu_pc(p, *pc))
    assert np.array_equal(got, plain)
