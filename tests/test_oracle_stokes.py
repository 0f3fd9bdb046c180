"""Instationary Stokes control (BASELINE configs[2]) in the CPU oracle.

The reference holds no known-answer test for this driver (its instationary Stokes tests are
the convergence studies of ``test/test_control.py:3546-3790``, kind C in SURVEY 4.2, which
need Firedrake): parity unpinned beyond the pieces the stationary Stokes KAT and the heat
KATs pin (outer block algebra, ConstantNullspace, nested solve, both time schemes).  What is
checked here is the reference's kind-A recipe on a manufactured solution: ``b = A x_ref``,
solve, compare.
"""
import numpy as np
import pytest

import common


@pytest.mark.parametrize("CN", [False, True])
def test_instationary_stokes_manufactured_solution(CN):
    p = common.stokes_problem(n=4, n_t=4, CN=CN)
    th, m = p["th"], p["m"]
    osys, opc = common.stokes_oracle(p)
    rng = np.random.default_rng(common.SEED)
    x0 = rng.standard_normal((2 * m, th.n_v))
    x0[:, th.boundary_v] = 0.0
    x1 = rng.standard_normal((2 * m, th.n_p))
    x1 -= x1.mean(axis=1, keepdims=True)
    b0, b1 = osys.split(osys.mult(osys.join(x0, x1)))
    u0, u1 = np.zeros_like(x0), np.zeros_like(x1)
    res = osys.solve(u0, u1, b0, b1, pc_fn=opc, solver_parameters={
        "linear_solver": "fgmres", "maximum_iterations": 200, "relative_tolerance": 1.0e-10,
        "absolute_tolerance": 1.0e-30, "monitor_convergence": False})
    assert res.reason > 0
    assert np.abs(u0 - x0).max() < 1.0e-6
    assert np.abs(u1 - u1.mean(axis=1, keepdims=True) - x1).max() < 1.0e-5


def test_outer_block_layout():
    """The outer ``block_00`` is the velocity KKT system flattened to [v; zeta]
    (``control/control.py:3793-3829``) and ``block_01/10`` carry ``tau B^T`` / ``tau B``."""
    p = common.stokes_problem(n=2, n_t=3)
    bl, m, th = p["blocks"], p["m"], p["th"]
    b00, b01, b10, b11 = bl["outer"]
    i00, i01, i10, i11 = bl["inner"]
    assert len(b00) == (2 * m) ** 2 and all(v is None for v in b11.values())
    for (i, j), A in i01.items():
        assert b00[(i, m + j)] is A
    for (i, j), A in i10.items():
        assert b00[(m + i, j)] is A
    for (i, j), A in i11.items():
        assert b00[(m + i, m + j)] is A
    assert b00[(m - 1, m - 1)] is None                 # :3935
    assert abs(b10[(1, 1)] - p["tau"] * th.B).max() == 0.0
    assert abs(b01[(2 * m - 1, 2 * m - 1)] - p["tau"] * th.B.T).max() == 0.0
    assert b01[(0, 1)] is None


def test_two_grid_pressure_laplacian_solve_with_deflated_constants():
    """The two-grid form of the K_p solve (``CoarseSpace.deflate``): the Galerkin matrix of the
    Neumann Laplacian is singular with the constants in its kernel; the inverse of
    ``E + (trace E / n_c^2) 1 1^T`` is its pseudo-inverse on zero-mean right-hand sides, and a few
    cycles of [correction, sweeps] solve ``K_p x = b`` for a zero-mean ``b`` up to a constant."""
    from control_amd.coarse import multilinear_coarse_space
    from oracle import kkt_oracle as ko
    th = common.stokes_problem(n=8, n_t=3)["th"]
    K = th.K_p.tocsr()
    P = multilinear_coarse_space(th.coords_p, (), cells=3)
    assert np.allclose(np.asarray(P.sum(axis=1)).ravel(), 1.0)
    co = ko.CoarseSpace(P, 6, deflate=True)
    E = (P.T @ (K @ P)).toarray()
    assert abs(E @ np.ones(E.shape[0])).max() < 1e-12             # singular: constants
    Einv = ko.coarse_inverse(K, co)
    rng = np.random.default_rng(common.SEED)
    rc = rng.standard_normal(E.shape[0])
    rc -= rc.mean()
    assert np.abs(E @ (Einv @ rc) - rc).max() < 1e-10             # pseudo-inverse on 1-perp
    x_ref = rng.standard_normal(th.n_p)
    b = K @ x_ref
    spec = ko.ChebSpec(8, 0.15, 2.1)
    spec.coarse = co
    x = ko._inner_solve(K, spec, b)
    assert np.linalg.norm(K @ x - b) < 1e-6 * np.linalg.norm(b)
    e = (x - x.mean()) - (x_ref - x_ref.mean())
    assert np.linalg.norm(e) < 1e-5 * np.linalg.norm(x_ref)
