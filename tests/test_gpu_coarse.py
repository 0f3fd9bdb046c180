"""Two-grid form of the block-Schur preconditioner's sub-solves (kkt_pc_desc.coarse_*): cycles of
[Galerkin correction on a small coarse space, a few Jacobi-Chebyshev smoothing sweeps] in place
of ~1.6 sqrt(kappa) sweeps on the whole spectrum.  Not in the reference (its sub-solves are
BoomerAMG cycles, control/control.py:2277-2288): parity is GPU against the oracle's restatement
(oracle.kkt_oracle.coarse_chebyshev), and the solve against the plain Chebyshev preconditioner's."""
import numpy as np
import pytest

import common
from control_amd.coarse import multilinear_coarse_space

pytestmark = pytest.mark.gpu

MASS = (20, 0.5, 2.0)


def _coarse(p, cells, cycles):
    sd = p["sd"]
    return multilinear_coarse_space(sd.coords, p["nodes"], cells=cells), cycles


@pytest.mark.parametrize("CN", [False, True])
@pytest.mark.parametrize("cycles", [1, 2])
@pytest.mark.parametrize("persistent", ["0", "1"])
def test_coarse_preconditioner_matches_the_oracle(CN, cycles, persistent):
    p = common.heat_problem(n=40, n_t=6, CN=CN, beta=1e-4)
    co = _coarse(p, 5, cycles)
    schur = (6, 2.1 / 30.0, 2.1)
    osys = common.oracle_system(p)
    g = common.gpu_system(p, options={"persistent": persistent})
    x = common.rng_vector(osys.N)
    ref = osys.pc_apply(common.oracle_pc(p, MASS, schur, coarse=co), x)
    got = g.pc_apply(x, common.gpu_pc(p, MASS, schur, coarse=co))
    assert common.rel_err(got, ref) < 1e-10
    # and the correction really took part (a random vector is mostly high frequencies, which the
    # sweeps handle alone: the difference is small, but far above the parity bar)
    plain = g.pc_apply(x, common.gpu_pc(p, MASS, schur))
    assert common.rel_err(plain, ref) > 1e-8
    if persistent == "1":
        assert g.info()["sweep_form"] == 3          # the tile program ran the two-grid levels


def test_coarse_solve_needs_fewer_sweeps_and_no_more_iterations():
    """64^2 x 12: GMRES(10) with 1 x (correction + 8 sweeps) converges in no more iterations than
    with 1.6 sqrt(kappa) plain sweeps per sub-solve -- a fifth of the dependent SpMV steps."""
    import bench
    p = common.heat_problem(n=64, n_t=12, beta=1e-4)
    m, nx = p["m"], p["sd"].n_dofs
    g = common.gpu_system(p)
    g0, g1 = bench.readme_rhs(p)
    sp = {"linear_solver": "gmres", "gmres_restart": 10, "maximum_iterations": 200,
          "relative_tolerance": 1e-6, "absolute_tolerance": 0.0, "monitor_convergence": False}
    out = {}
    for tag, pc in (("plain", common.gpu_pc(p, MASS, (-1, 0.0, 0.0))),
                    ("coarse", common.gpu_pc(p, MASS, (8, 2.1 / 30.0, 2.1),
                                             coarse=_coarse(p, 8, 1)))):
        u0, u1 = np.zeros((m, nx)), np.zeros((m, nx))
        r = g.solve(u0, u1, g0, g1, solver_parameters=sp, pc_fn=pc)
        out[tag] = (r.its, np.vstack([u0, u1]), g.info()["sweep_its"])
    assert out["coarse"][0] <= out["plain"][0] + 2, out
    assert out["coarse"][2] == 8 and out["plain"][2] >= 20, (out["coarse"][2], out["plain"][2])
    # (left-preconditioned GMRES stops on the preconditioned residual: two preconditioners, two
    # stopping points)
    assert common.rel_err(out["coarse"][1], out["plain"][1]) < 1e-2


@pytest.mark.parametrize("CN", [False, True])
def test_stokes_velocity_sub_solves_in_two_grid_form(CN):
    """The StokesPC's nested velocity solve with two-grid sub-solves (vector P2: one copy of the
    multilinear coarse functions per velocity component): one application against the oracle."""
    p = common.stokes_problem(n=8, n_t=5 if CN else 4, CN=CN)
    th = p["th"]
    P = multilinear_coarse_space(np.vstack([th.coords_v, th.coords_v]), th.boundary_v, cells=4)
    assert P.shape == (th.n_v, 2 * 25)
    specs = dict(common.STOKES_SPECS, schur=(6, 0.07, 2.2))
    osys, opc = common.stokes_oracle(p, specs, coarse=(P, 1))
    outer, gpc = common.stokes_gpu(p, specs, coarse=(P, 1))
    x = common.rng_vector(osys.N)
    # (nested 5-iteration GMRES: the bar of tests/test_gpu_stokes.py)
    assert common.rel_err(outer.pc_apply(x, gpc), osys.pc_apply(opc, x)) < (1e-7 if CN else 1e-4)


@pytest.mark.parametrize("CN", [False, True])
def test_stokes_pressure_laplacian_solve_in_two_grid_form(CN):
    """The K_p solve of the StokesPC as 2 x [Galerkin correction with the constants deflated,
    6 sweeps] (kkt_pc_stokes_desc.kp_coarse_*): one application against the oracle, and a solve
    of a manufactured right-hand side."""
    p = common.stokes_problem(n=8, n_t=5 if CN else 4, CN=CN)
    th, m = p["th"], p["m"]
    Pp = multilinear_coarse_space(th.coords_p, (), cells=3)
    assert Pp.shape == (th.n_p, 16)
    assert np.allclose(np.asarray(Pp.sum(axis=1)).ravel(), 1.0)      # partition of unity
    specs = dict(common.STOKES_SPECS, kp=(6, 0.15, 2.1))
    osys, opc = common.stokes_oracle(p, specs, kp_coarse=(Pp, 2))
    outer, gpc = common.stokes_gpu(p, specs, kp_coarse=(Pp, 2))
    x = common.rng_vector(osys.N)
    assert common.rel_err(outer.pc_apply(x, gpc), osys.pc_apply(opc, x)) < (1e-7 if CN else 1e-4)
    rng = np.random.default_rng(common.SEED)
    x0 = rng.standard_normal((2 * m, th.n_v))
    x0[:, th.boundary_v] = 0.0
    x1 = rng.standard_normal((2 * m, th.n_p))
    x1 -= x1.mean(axis=1, keepdims=True)
    b0, b1 = osys.split(osys.mult(osys.join(x0, x1)))
    u0, u1 = np.zeros_like(x0), np.zeros_like(x1)
    res = outer.solve(u0, u1, b0, b1, pc_fn=gpc, solver_parameters={
        "linear_solver": "fgmres", "maximum_iterations": 200, "relative_tolerance": 1.0e-10,
        "absolute_tolerance": 1.0e-30, "monitor_convergence": False})
    assert res.reason > 0
    assert np.abs(u0 - x0).max() < 1.0e-6


def test_coarse_tile_program_soak_and_sharded_equivalents():
    """The tile program with coarse corrections, repeated: every application equals the first bit
    for bit (the coarse residual is summed in a fixed order) and the plain launches to round-off
    (other association of the restriction sums); no time-out."""
    p = common.heat_problem(n=96, n_t=8, beta=1e-4)
    co = _coarse(p, 12, 2)
    schur = (6, 0.07, 2.1)
    x = common.rng_vector(2 * p["m"] * p["sd"].n_dofs)
    g = common.gpu_system(p, options={"prog_mode": "tile"})
    pc = common.gpu_pc(p, MASS, schur, coarse=co)
    y0 = g.pc_apply(x, pc)
    assert g.info()["sweep_form"] == 3
    for _ in range(60):
        assert np.array_equal(g.pc_apply(x, pc), y0)
    assert g.info()["program_fallbacks"] == 0
    plain = common.gpu_system(p, options={"persistent": "0"}).pc_apply(
        x, common.gpu_pc(p, MASS, schur, coarse=co))
    assert common.rel_err(y0, plain) < 1e-12
