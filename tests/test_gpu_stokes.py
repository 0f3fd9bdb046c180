"""Instationary Stokes control on the GPU against the oracle (SURVEY 8f-1, configs[2])."""
import numpy as np
import pytest

import common

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("CN", [False, True])
def test_operator_and_preconditioner_parity(CN):
    p = common.stokes_problem(n=4, n_t=4, CN=CN)
    osys, opc = common.stokes_oracle(p)
    outer, gpc = common.stokes_gpu(p)
    x = common.rng_vector(osys.N)
    assert common.rel_err(outer.mult(x), osys.mult(x)) < 1e-13
    # nested 5-iteration GMRES on the velocity KKT system: BE iterates are ill-conditioned
    # (tests/test_oracle.py::test_BE_iterates_are_ill_conditioned), CN ones are not
    tol = 1e-7 if CN else 1e-4
    assert common.rel_err(outer.pc_apply(x, gpc), osys.pc_apply(opc, x)) < tol


@pytest.mark.parametrize("CN", [False, True])
def test_manufactured_solution(CN):
    p = common.stokes_problem(n=4, n_t=4, CN=CN)
    th, m = p["th"], p["m"]
    osys, _ = common.stokes_oracle(p)
    outer, gpc = common.stokes_gpu(p)
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
    assert np.abs(u1 - u1.mean(axis=1, keepdims=True) - x1).max() < 1.0e-5


def test_config3_shape_runs():
    """A larger instance (16x16 cells, n_t = 8): the solve converges and the residual of the
    returned solution, evaluated by the oracle operator, meets the tolerance."""
    p = common.stokes_problem(n=16, n_t=8, beta=1.0e-2)
    th, m = p["th"], p["m"]
    osys, _ = common.stokes_oracle(p)
    outer, gpc = common.stokes_gpu(p)
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
