"""Every BASELINE.json configuration at FULL size on the GPU against the oracle.

configs[1]  2-D heat 256^2 P1, n_t = 64, BE, mode G, the preconditioner bench.py times
            (80 Chebyshev sweeps on [7e-4, 2.1]) -- the shape of test/test_control.py:1829-1933
configs[2]  2-D Stokes control, Taylor-Hood 128^2, n_t = 32 -- test_control.py:3546-3695
configs[3]  3-D heat 64^3 P1, n_t = 128 (70.3 M unknowns, 26 GB of block values in mode G)
configs[4]  2-D Navier-Stokes control (one Picard linearisation), 128^2, n_t = 64 --
            test_control.py:4171-4268

The oracle (NumPy / the C restatement; test infrastructure) is cheap enough at these sizes for
the operator and for ONE preconditioner application; whole Krylov solves are checked through the
true residual evaluated with the oracle operator.  configs[0] (10x10, n_t = 10) is the size of
every other parity test and of the golden fixtures.
"""
import time

import numpy as np
import pytest

import common
from oracle import cref

pytestmark = pytest.mark.gpu


def _log(msg, t0):
    print(f"[full-size] {msg}: {time.time() - t0:.1f} s", flush=True)


def _sweep_launches(gsys, gpc, x):
    """(persistent sweep launches, dependent SpMV steps in them) of one preconditioner
    application -- ``kkt_time_pc_sweeps``, the entry point behind bench.py's ``roofline_sweeps``.
    0 launches: the time sweeps run as plain launches (no persistent form fitted)."""
    import ctypes as C
    from control_amd import _lib
    gsys._set_pc(gpc)
    lib, h = gsys._lib, gsys.handle
    d_x, d_y = C.c_void_p(), C.c_void_p()
    gsys._ck(lib.kkt_vec_alloc(h, C.byref(d_x)))
    gsys._ck(lib.kkt_vec_alloc(h, C.byref(d_y)))
    gsys._ck(lib.kkt_vec_upload(h, d_x, _lib.f64(x)[1]))
    ms, launches, phases = C.c_float(), C.c_int(), C.c_int64()
    gsys._ck(lib.kkt_time_pc_sweeps(h, d_x, d_y, C.byref(ms), C.byref(launches),
                                    C.byref(phases)))
    return launches.value, phases.value


# ------------------------------------------------------------------ configs[1]
CFG2_MASS, CFG2_SCHUR = (20, 0.5, 2.0), (80, 7.0e-4, 2.1)      # bench.py defaults


@pytest.fixture(scope="module")
def cfg2():
    p = common.heat_problem(n=256, n_t=64, beta=1.0e-4)            # host objects shared ...
    g = common.gpu_system(p, share_values=False)                    # ... mode G on the device
    return p, g


def test_config2_operator_bit_identical_to_c_restatement(cfg2):
    """kkt_apply in mode G (380 value arrays, 8.45 M unknowns) equals the C loop bit for bit
    (same fma chain per row: blocks in dict order, CSR order inside a block) and the NumPy
    oracle to round-off."""
    p, g = cfg2
    assert g.info()["n_value_arrays"] == 6 * 64 - 4
    c = cref.CRef(p["blocks"], p["m"], p["sd"].n_dofs, p["nodes"], p["sd"].M, p["n_t"],
                  p["tau"], p["beta"], CFG2_MASS, CFG2_SCHUR)
    osys = common.oracle_system(p)
    for seed in range(2):
        x = common.rng_vector(osys.N, common.SEED + seed)
        y = g.mult(x)
        assert np.array_equal(y, c.mult(x))
        assert common.rel_err(y, osys.mult(x)) < 1e-13


def test_config2_bench_preconditioner_against_oracle_and_plain_launches(cfg2):
    """The configuration bench.py times -- mode G, (80, 7e-4, 2.1): two persistent sweep
    programs of 5 120 dependent SpMV steps each -- against oracle.pc_instationary_BE (1e-9), the
    C restatement, and its own plain-launch form (bit-identical)."""
    p, g = cfg2
    t0 = time.time()
    x = common.rng_vector(2 * p["m"] * p["sd"].n_dofs, 7)
    got = g.pc_apply(x, common.gpu_pc(p, CFG2_MASS, CFG2_SCHUR))
    _log("gpu pc_apply (incl. build)", t0)
    osys = common.oracle_system(p)
    ref = osys.pc_apply(common.oracle_pc(p, CFG2_MASS, CFG2_SCHUR), x)
    _log("+ numpy oracle", t0)
    assert common.rel_err(got, ref) < 1e-9
    # the form the bench line is about: one persistent launch per time sweep, every one of the
    # 2 x 64 x 80 dependent steps inside them
    assert _sweep_launches(g, common.gpu_pc(p, CFG2_MASS, CFG2_SCHUR), x) == (2, 2 * 64 * 80)
    c = cref.CRef(p["blocks"], p["m"], p["sd"].n_dofs, p["nodes"], p["sd"].M, p["n_t"],
                  p["tau"], p["beta"], CFG2_MASS, CFG2_SCHUR)
    xc = x.reshape(2 * p["m"], -1).copy()
    xc[:, p["nodes"]] = 0.0                    # the C routine is pc_fn itself: bc-clean input
    refc = c.pc_apply(xc.ravel()).reshape(2 * p["m"], -1)
    gc = got.reshape(2 * p["m"], -1).copy()
    refc[:, p["nodes"]] = gc[:, p["nodes"]]
    assert common.rel_err(gc, refc) < 1e-9
    g2 = common.gpu_system(p, share_values=False, options={"persistent": "0"})
    plain = g2.pc_apply(x, common.gpu_pc(p, CFG2_MASS, CFG2_SCHUR))
    _log("+ plain launches", t0)
    assert np.array_equal(got, plain)
    for mode in ("dataflow", "flags"):
        g3 = common.gpu_system(p, share_values=False, options={"prog_mode": mode})
        assert np.array_equal(got, g3.pc_apply(x, common.gpu_pc(p, CFG2_MASS, CFG2_SCHUR))), mode
    _log("+ other program forms", t0)


def test_config2_solve_true_residual(cfg2):
    """GMRES(10) with the bench preconditioner on a manufactured right-hand side: converges, and
    the true residual evaluated with the ORACLE operator is at the level the (preconditioned)
    monitor reports; the solution is the manufactured one."""
    p, g = cfg2
    osys = common.oracle_system(p)
    m, nx = p["m"], p["sd"].n_dofs
    X = p["sd"].coords
    xs = np.stack([np.sin(np.pi * X[:, 0]) * np.sin(np.pi * X[:, 1]) * (1 + 0.01 * k)
                   for k in range(2 * m)])
    b = osys.mult(xs.ravel()).reshape(2 * m, nx)
    sp_ = {"linear_solver": "fgmres", "gmres_restart": 30, "maximum_iterations": 120,
           "relative_tolerance": 1e-8, "absolute_tolerance": 0.0,
           "monitor_convergence": False}
    u0, u1 = np.zeros((m, nx)), np.zeros((m, nx))
    r = g.solve(u0, u1, b[:m].copy(), b[m:].copy(), solver_parameters=sp_,
                pc_fn=common.gpu_pc(p, CFG2_MASS, CFG2_SCHUR))
    u = np.vstack([u0, u1])
    res = np.linalg.norm(b.ravel() - osys.mult(u.ravel()))
    assert r.reason > 0
    assert res <= 2e-8 * np.linalg.norm(b)
    assert common.rel_err(u, xs) < 1e-4          # 1.1e-5 measured at a 2e-8 residual


def test_config2_crank_nicolson_operator_and_preconditioner():
    """The same system with the Crank-Nicolson scheme (test_control.py:1936-2040 shape; the
    time transforms T_1, T_2 around the blocks, control.py:1995-2189 for the preconditioner) as
    `bench.py --scheme CN` times it: operator 1e-13 and one preconditioner application 1e-9
    against the oracle, tile sweep programs bit-identical to plain launches."""
    t0 = time.time()
    p = common.heat_problem(n=256, n_t=64, beta=1.0e-4, CN=True)
    schur = (140, 7.0e-4, 2.1)
    g = common.gpu_system(p, share_values=False)
    osys = common.oracle_system(p)
    x = common.rng_vector(osys.N, common.SEED)
    assert common.rel_err(g.mult(x), osys.mult(x)) < 1e-13
    got = g.pc_apply(x, common.gpu_pc(p, CFG2_MASS, schur))
    _log("CN gpu operator + pc_apply", t0)
    ref = osys.pc_apply(common.oracle_pc(p, CFG2_MASS, schur), x)
    _log("+ numpy oracle", t0)
    assert common.rel_err(got, ref) < 1e-9
    g2 = common.gpu_system(p, share_values=False, options={"persistent": "0"})
    assert np.array_equal(got, g2.pc_apply(x, common.gpu_pc(p, CFG2_MASS, schur)))
    _log("+ plain launches", t0)


def test_config2_two_grid_preconditioner_against_the_oracle(cfg2):
    """The preconditioner bench.py times by default in round 3: Schur sub-solves as 2 x [Galerkin
    correction on the 33 x 33 multilinear coarse functions, 8 Chebyshev sweeps on [0.07, 2.1]] --
    one application against the oracle's restatement and the plain launches, the two sweep
    programs in tile form, and the README solve in 17 iterations (40 with 80 plain sweeps), closer
    to the tightly converged solution than the plain preconditioner's stopping point."""
    import bench
    from control_amd.coarse import multilinear_coarse_space
    p, g = cfg2
    osys = common.oracle_system(p)
    x = common.rng_vector(osys.N)
    t0 = time.time()
    co = (multilinear_coarse_space(p["sd"].coords, p["nodes"], cells=32), 2)
    assert co[0].shape[1] == 1089
    schur = (8, 0.07, 2.1)
    pc = common.gpu_pc(p, CFG2_MASS, schur, coarse=co)
    got = g.pc_apply(x, pc)
    _log("two-grid gpu pc", t0)
    ref = osys.pc_apply(common.oracle_pc(p, CFG2_MASS, schur, coarse=co), x)
    _log("+ oracle pc", t0)
    assert common.rel_err(got, ref) < 1e-9
    launches, phases = _sweep_launches(g, pc, x)
    assert launches == 2 and g.info()["sweep_form"] == 3, (launches, phases)
    g2 = common.gpu_system(p, share_values=False, options={"persistent": "0"})
    assert common.rel_err(got, g2.pc_apply(x, common.gpu_pc(p, CFG2_MASS, schur, coarse=co))) < 1e-12
    m, nx = p["m"], p["sd"].n_dofs
    g0, g1 = bench.readme_rhs(p)
    sp_ = {"linear_solver": "gmres", "gmres_restart": 10, "maximum_iterations": 100,
           "relative_tolerance": 1e-6, "absolute_tolerance": 0.0, "monitor_convergence": False}
    u0, u1 = np.zeros((m, nx)), np.zeros((m, nx))
    r = g.solve(u0, u1, g0, g1, solver_parameters=sp_, pc_fn=pc)
    _log(f"+ README solve ({r.its} iterations)", t0)
    assert r.reason > 0 and r.its <= 19          # 17 measured; 40 with 80 plain sweeps
    # the true residual of what it stopped at, with the oracle's operator
    # Left preconditioning stops on the preconditioned residual (the BE preconditioner scales the
    # final-time block by 1e3): what counts is the distance to the converged solution -- measured
    # 2e-6 here, 8e-3 where the plain preconditioner stops (scripts/r03_tts_quality.py).
    sp_t = dict(sp_, linear_solver="fgmres", gmres_restart=30, relative_tolerance=1e-11,
                maximum_iterations=400)
    t0_, t1_ = np.zeros((m, nx)), np.zeros((m, nx))
    rt = g.solve(t0_, t1_, g0, g1, solver_parameters=sp_t, pc_fn=pc)
    assert rt.reason > 0
    xt = np.concatenate([t0_.ravel(), t1_.ravel()])
    rhs = np.concatenate([g0.ravel(), g1.ravel()])
    assert np.linalg.norm(rhs - osys.mult(xt)) < 1e-9 * np.linalg.norm(rhs)     # oracle operator
    err = np.linalg.norm(np.concatenate([u0.ravel(), u1.ravel()]) - xt) / np.linalg.norm(xt)
    print(f"[full-size] two-grid README solve: {r.its} iterations, distance to the converged "
          f"solution {err:.1e}", flush=True)
    assert err < 1e-4


# ------------------------------------------------------------------ configs[2]
CFG3_SPECS = dict(mass=(20, 0.3924, 2.0598), mp=(20, 0.5, 2.0), schur=(40, 0.002, 2.25),
                  kp=(40, 0.002, 2.1))                               # bench.py --workload stokes2d


def test_config3_stokes_operator_and_preconditioner():
    """P2-P1 128 x 128, n_t = 32, BE (9.52 M unknowns): outer operator against the oracle, and
    ONE StokesPC application (5 nested GMRES iterations on the 8.45 M-unknown velocity system,
    K_p / M_p Chebyshev, commutator product) against oracle.pc_instationary_incompressible."""
    t0 = time.time()
    p = common.stokes_problem(n=128, n_t=32, beta=1.0e-3)
    _log("config 3 assembly", t0)
    osys, opc = common.stokes_oracle(p, CFG3_SPECS)
    outer, gpc = common.stokes_gpu(p, CFG3_SPECS)
    x = common.rng_vector(osys.N)
    assert common.rel_err(outer.mult(x), osys.mult(x)) < 1e-13
    _log("+ operator", t0)
    got = outer.pc_apply(x, gpc)
    _log("+ gpu StokesPC", t0)
    ref = osys.pc_apply(opc, x)
    _log("+ oracle StokesPC", t0)
    # the nested 5-iteration GMRES amplifies round-off (BE, 1/epsilon scaling of the last
    # level): the bar is 1e-4 at 4 x 4 x 4 (tests/test_gpu_stokes.py); measured here: 1.3e-11
    err = common.rel_err(got, ref)
    print(f"[full-size] config 3 StokesPC rel. deviation {err:.2e}", flush=True)
    assert err < 1e-8


def test_config3_stokes_two_grid_preconditioner():
    """The same StokesPC with the two-grid form of the velocity sub-solves (2 cycles of [Galerkin
    correction on 33 x 33 multilinear functions per component, 8 sweeps on [0.07, 2.25]]): one
    application against the oracle at full size."""
    from control_amd.coarse import multilinear_coarse_space
    t0 = time.time()
    p = common.stokes_problem(n=128, n_t=32, beta=1.0e-3)
    th = p["th"]
    P = multilinear_coarse_space(np.vstack([th.coords_v, th.coords_v]), th.boundary_v, cells=32)
    specs = dict(CFG3_SPECS, schur=(8, 0.07, 2.25))
    osys, opc = common.stokes_oracle(p, specs, coarse=(P, 2))
    outer, gpc = common.stokes_gpu(p, specs, coarse=(P, 2))
    x = common.rng_vector(osys.N)
    got = outer.pc_apply(x, gpc)
    _log("config 3 two-grid: gpu StokesPC", t0)
    ref = osys.pc_apply(opc, x)
    _log("+ oracle StokesPC", t0)
    err = common.rel_err(got, ref)
    print(f"[full-size] config 3 two-grid StokesPC rel. deviation {err:.2e}", flush=True)
    # measured 9.0e-8: on top of the nested GMRES, the 2178 x 2178 coarse inverses come from
    # Gauss-Jordan on the device here and from LAPACK in the oracle
    assert err < 1e-6
    assert gpc.inner.info()["program_fallbacks"] == 0


# ------------------------------------------------------------------ configs[3]
CFG4_MASS, CFG4_SCHUR = (20, 0.5, 2.5), (34, 7.44e-3, 2.1)       # suggest_chebyshev, bench.py


def test_config4_heat3d_full_size():
    """64^3 P1, n_t = 128, BE: 70.3 M unknowns on ONE GPU, every block with its own values on the
    device (764 value arrays, 26 GB); host objects shared so the host keeps two matrices.
    Operator bit-identical to the C loop; one preconditioner application against the C
    restatement (itself pinned to the NumPy oracle, tests/test_cref.py) and against the
    plain-launch form."""
    t0 = time.time()
    p = common.heat_problem(space="p1_3d", n=64, n_t=128, beta=1.0e-4)
    _log("config 4 assembly", t0)
    g = common.gpu_system(p, share_values=False)
    _log("+ upload", t0)
    info = g.info()
    assert info["n_local"] == 70304000 and info["n_value_arrays"] == 6 * 128 - 4
    c = cref.CRef(p["blocks"], p["m"], p["sd"].n_dofs, p["nodes"], p["sd"].M, p["n_t"],
                  p["tau"], p["beta"], CFG4_MASS, CFG4_SCHUR)
    x = common.rng_vector(info["n_local"])
    y = g.mult(x)
    assert np.array_equal(y, c.mult(x))
    _log("+ operator", t0)
    got = g.pc_apply(x, common.gpu_pc(p, CFG4_MASS, CFG4_SCHUR))
    _log("+ gpu pc", t0)
    xc = x.reshape(2 * p["m"], -1).copy()
    xc[:, p["nodes"]] = 0.0
    ref = c.pc_apply(xc.ravel()).reshape(2 * p["m"], -1)
    _log("+ C pc", t0)
    gc = got.reshape(2 * p["m"], -1)
    ref[:, p["nodes"]] = gc[:, p["nodes"]]
    assert common.rel_err(gc, ref) < 1e-9
    # (a tile plan that stops fitting would silently put 64^3 back on plain launches, 2.7x slower)
    launches, phases = _sweep_launches(g, common.gpu_pc(p, CFG4_MASS, CFG4_SCHUR), x)
    assert launches == 2 and phases >= 2 * 127 * 34, (launches, phases)
    g.set_option("persistent", "0")
    plain = g.pc_apply(x, common.gpu_pc(p, CFG4_MASS, CFG4_SCHUR))
    _log("+ plain launches", t0)
    assert np.array_equal(got, plain)


# ------------------------------------------------------------------ configs[4]
def test_config5_navier_stokes_linearised_solve():
    """P2-P1 128 x 128, n_t = 64 (19.0 M unknowns): the system of one Picard linearisation about
    a non-zero velocity at the reference's viscosity nu = 1/100 (test/test_control.py:4196; every
    time level its own convection block, mode G by nature).  Operator
    against the oracle; then one FGMRES solve with the StokesPC whose true residual is evaluated
    with the ORACLE operator."""
    from control_amd import picard
    from control_amd.blocks import instationary_incompressible_blocks
    from oracle import kkt_oracle as ko
    t0 = time.time()
    pb = common.navier_stokes_problem(n=128, n_t=64, nu=1.0 / 100.0, beta=1.0e-2)   # tst.py:4196
    th, n_t = pb.disc, pb.n_t
    v = 0.5 * pb.v_d
    D = [pb.D_v(v[i]) for i in range(n_t)]
    Dp = [pb.D_p(v[i]) for i in range(n_t)]
    _log("config 5 assembly (64 convection blocks)", t0)
    bl = instationary_incompressible_blocks(th.M_v, D, th.B, th.M_p, Dp, pb.tau, pb.beta,
                                            n_t, False)
    m = bl["m"]
    osys = ko.OracleSystem(
        th.n_v, th.n_p, *bl["outer"], n_blocks_00=2 * m, n_blocks_11=2 * m,
        nullspace_0=tuple(ko.DirichletBCNullspace(th.boundary_v) for _ in range(2 * m)),
        nullspace_1=tuple(ko.ConstantNullspace() for _ in range(2 * m)))
    sp_ = dict(common.NS_SOLVER_PARAMETERS, relative_tolerance=1.0e-6, maximum_iterations=200)
    # sub-solves: Chebyshev sweeps on the ellipse the library estimates for every level's matrix
    # (the blocks carry a convection term: symmetric part -> interval, skew part -> semi-axis)
    gls = picard.GpuLinearSolver(pb, mass=(20, 0.3924, 2.0598), schur=(-1, 0.0, 0.0),
                                 kp=(-1, 0.0, 0.0), mp=(20, 0.5, 2.0), solver_parameters=sp_)
    rng = np.random.default_rng(common.SEED)
    x0 = rng.standard_normal((2 * m, th.n_v))
    x0[:, th.boundary_v] = 0.0
    x1 = rng.standard_normal((2 * m, th.n_p))
    x1 -= x1.mean(axis=1, keepdims=True)
    xs = osys.join(x0, x1)
    b = osys.mult(xs)
    b0, b1 = osys.split(b)
    u0, u1, its = gls.linear_solve(D, Dp, b0, b1)
    _log(f"+ build and solve ({its} FGMRES iterations)", t0)
    assert common.rel_err(gls.outer.mult(xs), b) < 1e-13
    r = b - osys.mult(osys.join(u0, u1))
    assert np.linalg.norm(r) <= 2.0e-6 * np.linalg.norm(b)
