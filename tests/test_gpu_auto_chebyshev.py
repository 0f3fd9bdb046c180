"""The Chebyshev substitute of the reference's AMG sub-solves without hand-set numbers
(north_star: "per-time-step Laplacian/mass approximations are applied by Chebyshev SpMV
sweeps"; the reference itself gives no interval -- it calls BoomerAMG, control.py:2242-2431).
``ChebSpec(-1, 0, 0)``: the library estimates the Jacobi-scaled spectrum of every distinct
sub-solve matrix with a few Lanczos steps on the device and derives degree and intervals; the
first and last time levels carry other shifts (control.py:2241-2327) and get their own."""
import numpy as np
import pytest

import common
from control_amd.multiblock import ChebSpec, SchurPC

pytestmark = pytest.mark.gpu

AUTO = (-1, 0.0, 0.0)


def solve_readme_like(p, pc, max_it=150):
    g = common.gpu_system(p)
    m, nx = p["m"], p["sd"].n_dofs
    X = p["sd"].coords
    c = np.prod(np.cos(0.5 * np.pi * (X - 1.0)), axis=1)
    Mc = p["sd"].M @ c
    b0 = np.stack([p["tau"] * (i * p["tau"]) * Mc for i in range(m)])
    b1 = np.stack([p["tau"] * Mc for _ in range(m)])
    sp = {"linear_solver": "gmres", "gmres_restart": 10, "maximum_iterations": max_it,
          "relative_tolerance": 1e-6, "absolute_tolerance": 0.0, "monitor_convergence": False,
          "preconditioner": True}
    u0, u1 = np.zeros((m, nx)), np.zeros((m, nx))
    r = g.solve(u0, u1, b0, b1, solver_parameters=sp, pc_fn=pc)
    return r


@pytest.mark.parametrize("CN", [False, True])
@pytest.mark.parametrize("space,n,n_t", [("p1", 64, 16), ("p1", 128, 24), ("p1_3d", 16, 8)])
def test_default_sweeps_converge(space, n, n_t, CN):
    p = common.heat_problem(space=space, n=n, n_t=n_t, CN=CN, beta=1e-4)
    mass = (20, 0.5, 2.0) if space == "p1" else (20, 0.5, 2.5)
    r = solve_readme_like(p, common.gpu_pc(p, mass, AUTO))
    print(f"auto Chebyshev {space} {n} x {n_t} {'CN' if CN else 'BE'}: {r.its} iterations, "
          f"reason {r.reason}")
    assert r.reason > 0 and r.its <= 80


def test_estimated_interval_brackets_the_spectrum():
    """The per-matrix interval against a dense eigensolve of D^-1/2 (tau K + c M) D^-1/2 on a small
    mesh, through the convergence it buys: a fixed interval taken from the interior level alone
    (the first level's spectrum reaches lower) needs more iterations than the per-level ones."""
    p = common.heat_problem(n=32, n_t=8, CN=False, beta=1e-4)
    sd, tau = p["sd"], p["tau"]
    keep = np.setdiff1d(np.arange(sd.n_dofs), sd.boundary)

    def spectrum(c):
        L = (tau * sd.K + sd.M + c * sd.M).toarray()[np.ix_(keep, keep)]
        d = 1.0 / np.sqrt(np.diag(L))
        ev = np.linalg.eigvalsh(d[:, None] * L * d[None, :])
        return ev[0], ev[-1]
    lo_int, hi_int = spectrum(tau / np.sqrt(p["beta"]))
    lo_first, _ = spectrum(0.0)
    assert lo_first < 0.6 * lo_int              # the first level really needs its own interval
    its = int(np.ceil(1.6 * np.sqrt(hi_int / lo_int)))
    r_fixed = solve_readme_like(p, common.gpu_pc(p, (20, 0.5, 2.0), (its, 0.85 * lo_int, 1.05 * hi_int)))
    r_auto = solve_readme_like(p, common.gpu_pc(p, (20, 0.5, 2.0), AUTO))
    print(f"fixed interior interval: {r_fixed.its} its (reason {r_fixed.reason}); per level: {r_auto.its}")
    assert r_auto.reason > 0 and r_auto.its <= max(r_fixed.its, 1) + 2


def test_stokes_defaults_converge():
    """StokesPC with every Chebyshev parameter derived (inner sub-solves and K_p)."""
    specs = dict(mass=(20, 0.3924, 2.0598), schur=AUTO, kp=AUTO, mp=(20, 0.5, 2.0))
    p = common.stokes_problem(n=16, n_t=8, beta=1.0e-2)
    th, m = p["th"], p["m"]
    osys, _ = common.stokes_oracle(p)
    outer, gpc = common.stokes_gpu(p, specs)
    rng = np.random.default_rng(common.SEED)
    x0 = rng.standard_normal((2 * m, th.n_v))
    x0[:, th.boundary_v] = 0.0
    x1 = rng.standard_normal((2 * m, th.n_p))
    x1 -= x1.mean(axis=1, keepdims=True)
    b = osys.mult(osys.join(x0, x1))
    b0, b1 = osys.split(b)
    u0, u1 = np.zeros_like(x0), np.zeros_like(x1)
    res = outer.solve(u0, u1, b0, b1, pc_fn=gpc, solver_parameters={
        "linear_solver": "fgmres", "maximum_iterations": 300, "relative_tolerance": 1.0e-8,
        "absolute_tolerance": 1.0e-30, "monitor_convergence": False})
    assert res.reason > 0
    r = b - osys.mult(osys.join(u0, u1))
    assert np.linalg.norm(r) <= 2.0e-8 * np.linalg.norm(b)
