"""GPU parity tests (run with -m gpu on an MI355X): HIP path vs the CPU oracle, through
the C-ABI.  Tolerances are fp64 round-off of differently ordered sums:
operator 1e-13, preconditioner 1e-10, Krylov residual histories 1e-6 relative (see
test_krylov_iterates_parity for why not tighter), known-answer tests 1e-13 in L2."""
import numpy as np
import pytest

import common
import kat

pytestmark = pytest.mark.gpu

MASS = (20, 0.5, 2.0)          # P1 2-D bounds, test_control.py:3477
SCHUR = (10, 0.2, 2.1)       # 10x10 mesh: Jacobi-scaled tau K + c M has kappa ~ 8


@pytest.mark.parametrize("CN", [False, True])
@pytest.mark.parametrize("share", [True, False])
def test_operator_parity(CN, share):
    p = common.heat_problem(n=10, n_t=10, CN=CN, share=share, time_dependent=not share)
    osys = common.oracle_system(p)
    gsys = common.gpu_system(p)
    for seed in range(3):
        x = common.rng_vector(osys.N, common.SEED + seed)
        assert common.rel_err(gsys.mult(x), osys.mult(x)) < 1e-13
    info = gsys.info()
    assert info["n_patterns"] == 1            # one shared index structure
    nblk = sum(a is not None for blk in p["blocks"] for a in blk.values())
    assert info["n_blocks_stored"] == nblk
    if not share:
        assert info["n_value_arrays"] == nblk  # mode G: every block its own values


def test_operator_parity_3d():
    p = common.heat_problem(space="p1_3d", n=6, n_t=5)
    osys, gsys = common.oracle_system(p), common.gpu_system(p)
    x = common.rng_vector(osys.N)
    assert common.rel_err(gsys.mult(x), osys.mult(x)) < 1e-13


@pytest.mark.parametrize("CN", [False, True])
def test_preconditioner_parity(CN):
    p = common.heat_problem(n=10, n_t=10, CN=CN)
    osys, gsys = common.oracle_system(p), common.gpu_system(p)
    opc, gpc = common.oracle_pc(p, MASS, SCHUR), common.gpu_pc(p, MASS, SCHUR)
    x = common.rng_vector(osys.N)
    ref = osys.pc_apply(opc, x)
    got = gsys.pc_apply(x, gpc)
    assert common.rel_err(got, ref) < 1e-10
    # boundary rows pass through unchanged: u = P pc(P b) + (I - P) b
    nodes = p["nodes"]
    G = got.reshape(2 * p["m"], -1)
    X = x.reshape(2 * p["m"], -1)
    assert np.array_equal(G[:, nodes], X[:, nodes])


@pytest.mark.parametrize("CN", [False, True])
def test_preconditioner_parity_many_workgroups(CN):
    """96x96 mesh: 9409 rows = 74 slices = 19 workgroups, so the persistent sweep program
    (pc_row_program) really synchronises neighbours; also run with it disabled."""
    import os
    p = common.heat_problem(n=96, n_t=6, CN=CN)
    osys = common.oracle_system(p)
    schur = (6, 0.05, 2.1)
    x = common.rng_vector(osys.N)
    ref = osys.pc_apply(common.oracle_pc(p, MASS, schur), x)
    got = common.gpu_system(p).pc_apply(x, common.gpu_pc(p, MASS, schur))
    assert common.rel_err(got, ref) < 1e-10
    # same arithmetic in the same order: plain launches, the counter form and the data-flow
    # form of the persistent program agree exactly
    # ("w": the opt-in data-flow form for any row width, matrix re-read every phase)
    # ("interleave": the batched mass solves with one vector per time level instead of the
    # iterates of four levels interleaved)
    for var, val in (("persistent", "0"), ("prog_mode", "flags"), ("prog_mode", "w"),
                     ("prog_mode", "dataflow"), ("prog_mode", "tile"), ("interleave", "0")):
        other = common.gpu_system(p, options={var: val}).pc_apply(
            x, common.gpu_pc(p, MASS, schur))
        assert np.array_equal(got, other), (var, val)
    # the tiles: boxes from the dof coordinates (the default of gpu_system) or parts from the
    # bisection of the sparsity graph -- the partition changes nothing in the result
    other = common.gpu_system(p, tile_coordinates=False, options={"prog_mode": "tile"}).pc_apply(
        x, common.gpu_pc(p, MASS, schur))
    assert np.array_equal(got, other)


KRYLOV_SCHUR = (12, 0.08, 2.1)   # beta = 1e-2 on the 10x10 mesh: kappa(D^-1 S) ~ 25


@pytest.mark.parametrize("ksp", ["gmres", "fgmres"])
@pytest.mark.parametrize("CN", [False, True])
def test_krylov_iterates_parity(ksp, CN):
    """Iterate-for-iterate agreement with the oracle on a manufactured right-hand side
    (b = A x*, x* smooth and zero on the boundary).  Two fp64 implementations of
    GMRES with classical Gram-Schmidt separate exponentially with the iteration number
    at a rate set by the conditioning of the preconditioned operator, so the comparison
    uses the moderately conditioned beta = 1e-2 system, where the solve takes 8-27
    iterations; on the beta = 1e-4 system with a random right-hand side the two
    trajectories agree to 1e-9 for ~5 iterations and to O(1) only after ~40."""
    p = common.heat_problem(n=10, n_t=10, CN=CN, beta=1e-2)
    osys, gsys = common.oracle_system(p), common.gpu_system(p)
    opc = common.oracle_pc(p, MASS, KRYLOV_SCHUR)
    gpc = common.gpu_pc(p, MASS, KRYLOV_SCHUR)
    m, nx = p["m"], p["sd"].n_dofs
    X = p["sd"].coords
    xs = np.stack([np.sin(np.pi * X[:, 0]) * np.sin(np.pi * X[:, 1]) * (1 + 0.1 * k)
                   for k in range(2 * m)])
    b = osys.mult(xs.ravel()).reshape(2 * m, nx)
    sp = {"linear_solver": ksp, "gmres_restart": 10, "maximum_iterations": 60,
          "relative_tolerance": 1e-6, "absolute_tolerance": 0.0,
          "monitor_convergence": False, "preconditioner": True}
    uo0, uo1 = np.zeros((m, nx)), np.zeros((m, nx))
    ro = osys.solve(uo0, uo1, b[:m], b[m:], solver_parameters=sp, pc_fn=opc)
    ug0, ug1 = np.zeros((m, nx)), np.zeros((m, nx))
    rg = gsys.solve(ug0, ug1, b[:m].copy(), b[m:].copy(), solver_parameters=sp, pc_fn=gpc)
    ho, hg = np.asarray(ro.history), np.asarray(rg.history)
    assert ro.reason > 0 and rg.reason == ro.reason
    if CN:
        # stated fp64 tolerance: same iteration count, every monitored norm to 1e-6 of its
        # own size, final solution to 1e-6 relative
        assert rg.its == ro.its and len(ho) == len(hg)
        assert np.max(np.abs(hg - ho) / ho) < 1e-6
        assert common.rel_err(np.vstack([ug0, ug1]), np.vstack([uo0, uo1])) < 1e-6
    else:
        # BE: the preconditioner scales the final-time block by 1/epsilon = 1e3
        # (control.py:2205-2206); Gram-Schmidt then cancels ~3 digits per step and ANY 1e-16
        # perturbation grows to 1e-4..1e-1 of the recurrence-estimated norms within one
        # restart cycle -- the oracle does that against itself, see
        # test_oracle.py::test_BE_iterates_are_ill_conditioned.  Checked here: the first
        # three norms to 1e-9, the count to +-1, all norms to a factor 1.5, and (fgmres,
        # which monitors the true residual) the solutions to 1e-5.
        n = min(len(ho), len(hg))
        assert abs(rg.its - ro.its) <= 1
        assert np.max(np.abs(hg[:3] - ho[:3]) / ho[:3]) < 1e-9
        assert np.all(hg[:n] < 1.5 * ho[:n]) and np.all(ho[:n] < 1.5 * hg[:n])
        if ksp == "fgmres":
            assert common.rel_err(np.vstack([ug0, ug1]), np.vstack([uo0, uo1])) < 1e-5


def test_identity_and_callback_pc():
    p = common.heat_problem(n=6, n_t=4)
    osys, gsys = common.oracle_system(p), common.gpu_system(p)
    x = common.rng_vector(osys.N)

    def ident(u_0, u_1, b_0, b_1):
        u_0[:] = b_0
        u_1[:] = b_1
    assert common.rel_err(gsys.pc_apply(x, None), osys.pc_apply(ident, x)) == 0.0

    def scaled(u_0, u_1, b_0, b_1):
        u_0[:] = 2.0 * b_0
        u_1[:] = -3.0 * b_1
    assert common.rel_err(gsys.pc_apply(x, scaled), osys.pc_apply(scaled, x)) == 0.0

    def broken(u_0, u_1, b_0, b_1):
        raise ValueError("boom")
    with pytest.raises(RuntimeError, match="Error encountered in PETSc solve"):
        gsys.pc_apply(x, broken)


def test_nonconvergence_raises():
    p = common.heat_problem(n=6, n_t=4)
    gsys = common.gpu_system(p)
    m, nx = p["m"], p["sd"].n_dofs
    b = common.rng_vector(2 * m * nx).reshape(2 * m, nx)
    sp = {"linear_solver": "gmres", "gmres_restart": 5, "maximum_iterations": 3,
          "relative_tolerance": 1e-12, "absolute_tolerance": 0.0,
          "monitor_convergence": False}
    with pytest.raises(RuntimeError, match="Solver failed to converge"):
        gsys.solve(np.zeros((m, nx)), np.zeros((m, nx)), b[:m], b[m:],
                   solver_parameters=sp)


# ---- the reference's known-answer tests through the GPU path
KAT_SCHUR = (40, 0.02, 2.2)


@pytest.mark.parametrize("CN", [False, True])
def test_kat_instationary(CN):
    from control_amd.blocks import instationary_blocks
    from control_amd.multiblock import (ChebSpec, DirichletBCNullspace, MultiBlockSystem,
                                        SchurPC)
    from oracle import kkt_oracle as ko
    p = kat.kat_instationary_CN() if CN else kat.kat_instationary_BE()
    sd, n_t, tau, beta = p["sd"], p["n_t"], p["tau"], p["beta"]
    b00, b01, b10, b11, m = instationary_blocks(sd.M, sd.K, tau, beta, n_t, CN)
    ns = tuple(DirichletBCNullspace(p["nodes"]) for _ in range(m))
    gsys = MultiBlockSystem(sd.n_dofs, sd.n_dofs, b00, b01, b10, b11, n_blocks_00=m,
                            n_blocks_11=m, nullspace_0=ns, nullspace_1=ns, CN=CN)
    pc = SchurPC(kind="CN" if CN else "BE", M=sd.M, beta=beta, bc_nodes=p["nodes"],
                 mass=ChebSpec(20, *kat.LAMBDA_V_BOUNDS), schur=ChebSpec(*KAT_SCHUR),
                 n_t=n_t, tau=tau)
    b_0, b_1 = (ko.apply_T_1(p["b_0"]), ko.apply_T_2(p["b_1"])) if CN else (p["b_0"], p["b_1"])
    v, z = np.zeros((m, sd.n_dofs)), np.zeros((m, sd.n_dofs))
    res = gsys.solve(v, z, b_0, b_1, solver_parameters=kat.SOLVER_PARAMETERS, pc_fn=pc)
    assert res.reason > 0
    if CN:
        v = np.vstack([np.zeros((1, sd.n_dofs)), v])
        z = np.vstack([z, np.zeros((1, sd.n_dofs))])
    assert kat.l2_norm(sd.M, v - p["v_ref"]) < 1.0e-13
    assert kat.l2_norm(sd.M, z - p["z_ref"]) < 1.0e-13


def test_kat_stationary():
    from control_amd.blocks import stationary_blocks
    from control_amd.multiblock import ChebSpec, MultiBlockSystem, SchurPC
    p = kat.kat_stationary()
    sd = p["sd"]
    b00, b01, b10, b11 = stationary_blocks(sd.M, p["D"], p["beta"])
    gsys = MultiBlockSystem(sd.n_dofs, sd.n_dofs, b00, b01, b10, b11)
    pc = SchurPC(kind="stationary", M=sd.M, beta=p["beta"], bc_nodes=p["nodes"],
                 mass=ChebSpec(20, *kat.LAMBDA_V_BOUNDS), schur=ChebSpec(*KAT_SCHUR))
    v, z = np.zeros((1, sd.n_dofs)), np.zeros((1, sd.n_dofs))
    res = gsys.solve(v, z, p["b_0"], p["b_1"], solver_parameters=kat.SOLVER_PARAMETERS,
                     pc_fn=pc)
    assert res.reason > 0
    assert kat.l2_norm(sd.M, v - p["v_ref"]) < 1.0e-13
    assert kat.l2_norm(sd.M, z - p["z_ref"]) < 1.0e-13


def _stokes_systems(p):
    """Outer Stokes-control system + velocity KKT + pressure commutator on the GPU."""
    from control_amd.multiblock import (ChebSpec, ConstantNullspace, DirichletBCNullspace,
                                        MultiBlockSystem, SchurPC, StokesPC)
    from oracle import kkt_oracle as ko
    import scipy.sparse as sp
    th, beta = p["th"], p["beta"]
    D, D_p = p["D"], p["D_p"]
    blocks = ko.stationary_incompressible_blocks(th.M_v, D, th.B, beta)
    nsv = DirichletBCNullspace(th.boundary_v)
    outer = MultiBlockSystem(th.n_v, th.n_p, *blocks, n_blocks_00=2, n_blocks_11=2,
                             nullspace_0=(nsv, nsv),
                             nullspace_1=(ConstantNullspace(), ConstantNullspace()))
    inner = MultiBlockSystem(th.n_v, th.n_v, {(0, 0): th.M_v}, {(0, 0): sp.csr_matrix(D.T)},
                             {(0, 0): D}, {(0, 0): sp.csr_matrix((-1.0 / beta) * th.M_v)},
                             nullspace_0=(nsv,), nullspace_1=(nsv,))
    comm = MultiBlockSystem(th.n_p, th.n_p, {(0, 0): th.M_p}, {(0, 0): sp.csr_matrix(D_p.T)},
                            {(0, 0): D_p}, {(0, 0): sp.csr_matrix((-1.0 / beta) * th.M_p)})
    mass, schur, kp = (20, 0.25, 1.5625), (40, 0.02, 2.2), (30, 0.02, 2.2)
    mp = (20,) + tuple(p["lambda_p_bounds"])
    inner_pc = SchurPC(kind="stationary", M=th.M_v, beta=beta, bc_nodes=th.boundary_v,
                       mass=ChebSpec(*mass), schur=ChebSpec(*schur))
    gpc = StokesPC(inner=inner, inner_pc=inner_pc, commutator=comm, B=th.B, K_p=th.K_p,
                   M_p=th.M_p, kp=ChebSpec(*kp), mp=ChebSpec(*mp))
    opc = ko.pc_stationary_incompressible(th.M_v, D, th.B, th.M_p, th.K_p, D_p, beta,
                                          th.boundary_v, ko.ChebSpec(*mass), ko.ChebSpec(*schur),
                                          ko.ChebSpec(*kp), ko.ChebSpec(*mp))
    osys = ko.OracleSystem(th.n_v, th.n_p, *blocks, n_blocks_00=2, n_blocks_11=2,
                           nullspace_0=(ko.DirichletBCNullspace(th.boundary_v),) * 2,
                           nullspace_1=(ko.ConstantNullspace(), ko.ConstantNullspace()))
    return outer, gpc, osys, opc


def test_stokes_control_operator_and_preconditioner_parity():
    """SURVEY 8f-1, stationary: rectangular divergence blocks, ConstantNullspace on the
    pressures, nested velocity solve and pressure Schur complement (control.py:802-1110)."""
    p = kat.kat_stationary_incompressible()
    outer, gpc, osys, opc = _stokes_systems(p)
    x = common.rng_vector(osys.N)
    assert common.rel_err(outer.mult(x), osys.mult(x)) < 1e-13
    # the nested 5-iteration GMRES amplifies round-off like any Krylov iterate comparison
    assert common.rel_err(outer.pc_apply(x, gpc), osys.pc_apply(opc, x)) < 1e-8


def test_kat_stationary_incompressible():
    """test/test_control.py:232-358 through the GPU path."""
    p = kat.kat_stationary_incompressible()
    th = p["th"]
    outer, gpc, _, _ = _stokes_systems(p)
    u0, u1 = np.zeros((2, th.n_v)), np.zeros((2, th.n_p))
    res = outer.solve(u0, u1, p["b_0"], p["b_1"], solver_parameters=p["solver_parameters"],
                      pc_fn=gpc)
    assert res.reason > 0

    def l2(M, e):
        return np.sqrt(abs(e @ (M @ e)))

    def demean(M, q):
        return q - (np.ones_like(q) @ (M @ q))
    assert l2(th.M_v, u0[0] - p["v_ref"]) < 1.0e-13
    assert l2(th.M_v, u0[1] - p["z_ref"]) < 1.0e-13
    assert l2(th.M_p, demean(th.M_p, u1[1]) - demean(th.M_p, p["p_ref"])) < 5.0e-13   # see
    assert l2(th.M_p, demean(th.M_p, u1[0]) - demean(th.M_p, p["mu_ref"])) < 1.0e-13  # oracle KAT


@pytest.mark.parametrize("CN", [False, True])
def test_preconditioner_parity_high_degree(CN):
    """The default benchmark preconditioner runs 80-140 Chebyshev sweeps per sub-solve: sweep
    programs of several thousand phases.  Same parity as at low degree."""
    p = common.heat_problem(n=48, n_t=8, CN=CN)
    osys = common.oracle_system(p)
    schur = (120, 0.002, 2.1)
    x = common.rng_vector(osys.N)
    ref = osys.pc_apply(common.oracle_pc(p, MASS, schur), x)
    got = common.gpu_system(p).pc_apply(x, common.gpu_pc(p, MASS, schur))
    assert common.rel_err(got, ref) < 1e-9


@pytest.mark.parametrize("CN", [False, True])
def test_timed_sweep_programs_give_the_preconditioner_result(CN):
    """``kkt_time_pc_sweeps`` (the measurement entry point behind ``roofline_sweeps`` of
    bench.py) replays the preconditioner step by step with a HIP event pair around every
    persistent sweep launch: the result is the preconditioner's, bit for bit, and the counts
    are the two time sweeps with ``n_levels * (its + 1)`` phases each."""
    import ctypes as C
    from control_amd import _lib
    p = common.heat_problem(n=24, n_t=6, CN=CN)
    gsys = common.gpu_system(p, options={"persistent": "1"})    # whatever KKT_PERSISTENT says
    gpc = common.gpu_pc(p, MASS, (12, 0.05, 2.2))
    x = common.rng_vector(2 * p["m"] * p["sd"].n_dofs)
    ref = gsys.pc_apply(x, gpc)
    lib, h = gsys._lib, gsys.handle
    d_x, d_y = C.c_void_p(), C.c_void_p()
    gsys._ck(lib.kkt_vec_alloc(h, C.byref(d_x)))
    gsys._ck(lib.kkt_vec_alloc(h, C.byref(d_y)))
    a, pa = _lib.f64(x)
    gsys._ck(lib.kkt_vec_upload(h, d_x, pa))
    ms, launches, phases = C.c_float(), C.c_int(), C.c_int64()
    gsys._ck(lib.kkt_time_pc_sweeps(h, d_x, d_y, C.byref(ms), C.byref(launches),
                                    C.byref(phases)))
    got = np.zeros_like(x)
    gsys._ck(lib.kkt_vec_download(h, d_y, got.ctypes.data_as(_lib.c_f64p)))
    assert np.array_equal(got, ref)
    assert launches.value >= 2 and ms.value > 0.0
    assert phases.value >= 2 * (p["m"] - 1) * 13 and phases.value <= 2 * (p["m"] + 1) * 14


# ---- MINRES through the C-ABI ("linear_solver": "minres", preconditioner.py:733)
def _spd_jacobi(p):
    d = p["sd"].M.diagonal()
    tau, beta = p["tau"], p["beta"]

    def pc(u_0, u_1, b_0, b_1):
        u_0[:] = b_0 / (tau * d)
        u_1[:] = b_1 / ((tau / beta) * d)
    return pc


@pytest.mark.parametrize("pc_kind", ["identity", "spd"])
def test_minres_iterates_parity(pc_kind):
    """Device MINRES against the oracle's restatement of PETSc's classic ``KSPSolve_MINRES``:
    same residual history over the first 25 steps (1e-9 relative) and the same converged
    solution.  Later steps are not compared entry by entry: the Lanczos recurrence amplifies
    the order of summation in the inner products -- the oracle run against itself with the
    dot products summed backwards agrees to 2e-13 after 30 steps, 1e-5 after 60 and differs
    by 2 % in the iteration count (1 115 / 1 137 with the identity, 131 / 130 with the SPD
    preconditioner) while the solutions agree to 2e-12."""
    p = common.heat_problem(n=6, n_t=4, CN=False)
    osys, gsys = common.oracle_system(p), common.gpu_system(p)
    m, nx = p["m"], p["sd"].n_dofs
    b = common.rng_vector(2 * m * nx).reshape(2 * m, nx)
    b[:, p["nodes"]] = 0.0
    pc = _spd_jacobi(p) if pc_kind == "spd" else None
    sp = {"linear_solver": "minres", "relative_tolerance": 1e-11, "absolute_tolerance": 0.0,
          "maximum_iterations": 6000, "monitor_convergence": False, "divergence limit": 1e12}
    vo, zo = np.zeros((m, nx)), np.zeros((m, nx))
    ro = osys.solve(vo, zo, b[:m], b[m:], solver_parameters=sp, pc_fn=pc)
    vg, zg = np.zeros((m, nx)), np.zeros((m, nx))
    rg = gsys.solve(vg, zg, b[:m], b[m:], solver_parameters=sp, pc_fn=pc)
    assert rg.reason == ro.reason == 2
    k = min(25, len(ro.history), len(rg.history))
    assert np.allclose(rg.history[:k], ro.history[:k], rtol=1e-9, atol=0.0)
    assert abs(rg.its - ro.its) <= max(3, ro.its // 20)
    assert common.rel_err(np.r_[vg.ravel(), zg.ravel()], np.r_[vo.ravel(), zo.ravel()]) < 1e-8
    assert np.all(vg[:, p["nodes"]] == 0.0)


def test_minres_reports_an_indefinite_preconditioner_and_refuses_right_side():
    p = common.heat_problem(n=6, n_t=4, CN=False)
    gsys = common.gpu_system(p)
    m, nx = p["m"], p["sd"].n_dofs
    b = common.rng_vector(2 * m * nx).reshape(2 * m, nx)
    b[:, p["nodes"]] = 0.0

    def negative(u_0, u_1, b_0, b_1):
        u_0[:] = -b_0
        u_1[:] = -b_1
    sp = {"linear_solver": "minres", "relative_tolerance": 1e-8, "absolute_tolerance": 0.0,
          "maximum_iterations": 50, "monitor_convergence": False, "preconditioner": True}
    r = gsys.solve(np.zeros((m, nx)), np.zeros((m, nx)), b[:m], b[m:], solver_parameters=sp,
                   pc_fn=negative)
    assert r.reason == -8 and r.its == 0          # KSP_DIVERGED_INDEFINITE_PC
    with pytest.raises(RuntimeError, match="Solver failed to converge"):
        gsys.solve(np.zeros((m, nx)), np.zeros((m, nx)), b[:m], b[m:],
                   solver_parameters=dict(sp, preconditioner=False), pc_fn=negative)
    with pytest.raises(Exception, match="left preconditioning"):
        gsys.solve(np.zeros((m, nx)), np.zeros((m, nx)), b[:m], b[m:],
                   solver_parameters=dict(sp, pc_side="right"))


@pytest.mark.parametrize("CN", [False, True])
def test_preconditioner_parity_3d_tile_form(CN):
    """3-D P1 (15 entries per row): the tile form for wide rows -- with the level update in the
    kernel (its matrix passing through the registers of the level matrix) where every update
    has one term, and without (option ``tile_unfused``, and wherever an update has two terms:
    one plain launch for b -= U u_prev and one tile launch per time level, hand-off tags counting
    through the launches) -- against the oracle and, bit for bit, against the plain launches."""
    p = common.heat_problem(space="p1_3d", n=16, n_t=5, CN=CN)
    osys = common.oracle_system(p)
    mass, schur = (20, 0.5, 2.5), (7, 0.05, 2.1)
    x = common.rng_vector(osys.N)
    ref = osys.pc_apply(common.oracle_pc(p, mass, schur), x)
    g = common.gpu_system(p, options={"prog_mode": "tile"})
    got = g.pc_apply(x, common.gpu_pc(p, mass, schur))
    assert common.rel_err(got, ref) < 1e-10
    again = g.pc_apply(x, common.gpu_pc(p, mass, schur))      # replay: tags of the first run are stale
    assert np.array_equal(got, again)
    plain = common.gpu_system(p, options={"persistent": "0"}).pc_apply(
        x, common.gpu_pc(p, mass, schur))
    assert np.array_equal(got, plain)
    assert g.info()["program_fallbacks"] == 0
    g2 = common.gpu_system(p, options={"prog_mode": "tile", "tile_unfused": "1"})
    assert np.array_equal(got, g2.pc_apply(x, common.gpu_pc(p, mass, schur)))
    assert g2.info()["program_fallbacks"] == 0
    g3 = common.gpu_system(p, tile_coordinates=False, options={"prog_mode": "tile"})
    assert np.array_equal(got, g3.pc_apply(x, common.gpu_pc(p, mass, schur)))   # graph bisection
    assert g3.info()["program_fallbacks"] == 0
