"""CPU checks of the oracle itself (no GPU): operator vs an independently assembled
global matrix, preconditioner exactness in the limit, and how ill-conditioned the
comparison of Krylov iterates is."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

import common
from oracle import kkt_oracle as ko


def global_matrix(p):
    """P A P + (I - P) of SURVEY 8.0, rows written out from M and K (BE and CN)."""
    sd, n_t, tau, beta, CN = p["sd"], p["n_t"], p["tau"], p["beta"], p["CN"]
    M, K, nx = sd.M, sd.K, sd.n_dofs
    m = p["m"]
    Z = None
    blk = [[Z] * (2 * m) for _ in range(2 * m)]
    if not CN:
        L = tau * K + M
        for i in range(m):
            if i < m - 1:
                blk[i][i] = tau * M
                blk[i][m + i + 1] = -M
            blk[i][m + i] = L.T
            blk[m + i][i] = L
            if i >= 1:
                blk[m + i][i - 1] = -M
                blk[m + i][m + i] = -(tau / beta) * M
        A = sp.bmat(blk, format="csr")
    else:
        h = 0.5 * tau
        raw = [[Z] * (2 * m) for _ in range(2 * m)]
        for i in range(m):
            raw[i][i] = h * M
            raw[i][m + i] = h * K.T + M
            raw[m + i][i] = h * K + M
            raw[m + i][m + i] = -(h / beta) * M
            if i >= 1:
                raw[i][i - 1] = h * M
                raw[m + i][i - 1] = h * K - M
            if i + 1 < m:
                raw[i][m + i + 1] = h * K.T - M
                raw[m + i][m + i + 1] = -(h / beta) * M
        R = sp.bmat(raw, format="csr")
        I = sp.identity(nx, format="csr")
        T1 = sp.bmat([[I if j in (i, i + 1) else None for j in range(m)] for i in range(m)])
        T2 = sp.bmat([[I if j in (i, i - 1) else None for j in range(m)] for i in range(m)])
        A = sp.block_diag([T1, T2]) @ R
    keep = np.ones(2 * m * nx)
    for k in range(2 * m):
        keep[k * nx + p["nodes"]] = 0.0
    Pm = sp.diags(keep)
    return (Pm @ A @ Pm + sp.diags(1.0 - keep)).tocsr()


@pytest.mark.parametrize("CN", [False, True])
def test_operator_matches_global_matrix(CN):
    p = common.heat_problem(n=6, n_t=5, CN=CN)
    osys = common.oracle_system(p)
    A = global_matrix(p)
    x = common.rng_vector(osys.N)
    assert common.rel_err(osys.mult(x), A @ x) < 1e-13


@pytest.mark.parametrize("CN", [False, True])
def test_krylov_solution_matches_direct_solve(CN):
    p = common.heat_problem(n=6, n_t=5, CN=CN, beta=1e-2)
    osys = common.oracle_system(p)
    A = global_matrix(p)
    m, nx = p["m"], p["sd"].n_dofs
    b = common.rng_vector(osys.N).reshape(2 * m, nx).copy()
    for k in range(2 * m):
        b[k, p["nodes"]] = 0.0
    x_direct = spla.spsolve(A.tocsc(), b.ravel())
    opc = common.oracle_pc(p, (20, 0.5, 2.0), (30, 0.05, 2.1))
    u0, u1 = np.zeros((m, nx)), np.zeros((m, nx))
    sp_ = {"linear_solver": "fgmres", "gmres_restart": 30, "maximum_iterations": 200,
           "relative_tolerance": 1e-12, "absolute_tolerance": 0.0, "monitor_convergence": False}
    res = osys.solve(u0, u1, b[:m], b[m:], solver_parameters=sp_, pc_fn=opc)
    assert res.reason > 0
    assert common.rel_err(np.vstack([u0, u1]), x_direct) < 1e-8


def test_time_transform_inverses():
    x = common.rng_vector(7 * 13).reshape(7, 13)
    assert common.rel_err(ko.apply_T_1_inv(ko.apply_T_1(x)), x) < 1e-14
    assert common.rel_err(ko.apply_T_2_inv(ko.apply_T_2(x)), x) < 1e-14


def test_BE_iterates_are_ill_conditioned():
    """Why BE iterate parity cannot be tight: perturbing the oracle's own preconditioner
    output by 2e-16 relative changes its monitored norms by > 1e-6 within ten iterations
    (the 1/epsilon scaling of the final-time block, control.py:2205-2206)."""
    p = common.heat_problem(n=10, n_t=10, CN=False, beta=1e-2)
    osys = common.oracle_system(p)
    opc = common.oracle_pc(p, (20, 0.5, 2.0), (12, 0.08, 2.1))
    rng = np.random.default_rng(1)

    def opc_pert(u0, u1, b0, b1):
        opc(u0, u1, b0, b1)
        u0 *= 1 + 2e-16 * rng.standard_normal(u0.shape)
        u1 *= 1 + 2e-16 * rng.standard_normal(u1.shape)
    m, nx = p["m"], p["sd"].n_dofs
    X = p["sd"].coords
    xs = np.stack([np.sin(np.pi * X[:, 0]) * np.sin(np.pi * X[:, 1]) * (1 + 0.1 * k)
                   for k in range(2 * m)])
    b = osys.mult(xs.ravel()).reshape(2 * m, nx)
    sp_ = {"linear_solver": "fgmres", "gmres_restart": 10, "maximum_iterations": 60,
           "relative_tolerance": 1e-6, "absolute_tolerance": 0.0,
           "monitor_convergence": False, "preconditioner": True}
    H = []
    for pc in (opc, opc_pert):
        u0, u1 = np.zeros((m, nx)), np.zeros((m, nx))
        H.append(np.asarray(osys.solve(u0, u1, b[:m], b[m:], solver_parameters=sp_,
                                       pc_fn=pc).history))
    n = min(len(H[0]), len(H[1]), 11)
    drift = np.abs(H[0][:n] - H[1][:n]) / H[0][:n]
    assert drift[:3].max() < 1e-9 and drift.max() > 1e-6


# ---- MINRES (reachable through "linear_solver": "minres", never used by the reference)
def _spd_jacobi(p):
    """A symmetric positive definite preconditioner for the BE system: Jacobi on
    ``tau M`` (state rows) and ``(tau / beta) M`` (adjoint rows)."""
    d = p["sd"].M.diagonal()
    tau, beta = p["tau"], p["beta"]

    def pc(u_0, u_1, b_0, b_1):
        u_0[:] = b_0 / (tau * d)
        u_1[:] = b_1 / ((tau / beta) * d)
    return pc


def test_minres_iterates_match_scipy():
    """The restated PETSc-classic MINRES against SciPy's independent implementation of
    Paige & Saunders: same iterates after k steps on the (symmetric) BE system with an SPD
    preconditioner.  This pins the algorithm; PETSc itself is not available (unpinned)."""
    p = common.heat_problem(n=6, n_t=4, CN=False)
    A = global_matrix(p)
    assert abs(A - A.T).max() < 1e-14
    osys = common.oracle_system(p)
    pc = _spd_jacobi(p)
    b = common.rng_vector(osys.N)
    for k in range(2 * p["m"]):
        b[k * p["sd"].n_dofs + p["nodes"]] = 0.0

    def B(v):
        return osys.pc_apply(pc, v)
    Bmat = spla.LinearOperator(A.shape, matvec=B, dtype=np.float64)
    for k in (1, 7, 40):
        x = np.zeros(osys.N)
        r = ko.minres(osys.mult, B, b, x, rtol=0.0, atol=0.0, divtol=1e300, max_it=k)
        xs, _ = spla.minres(A, b, M=Bmat, maxiter=k, rtol=1e-300)
        assert r.its == k and r.reason == ko.DIVERGED_ITS and len(r.history) == k + 1
        assert np.abs(x - xs).max() <= 1e-11 * np.abs(xs).max()
        assert all(b_ <= a_ * (1 + 1e-12) for a_, b_ in zip(r.history, r.history[1:]))


def test_minres_solves_the_system_and_flags_an_indefinite_preconditioner():
    p = common.heat_problem(n=6, n_t=4, CN=False)
    osys, m, nx = common.oracle_system(p), p["m"], p["sd"].n_dofs
    b = common.rng_vector(2 * m * nx).reshape(2 * m, nx)
    b[:, p["nodes"]] = 0.0
    sp_ = {"linear_solver": "minres", "relative_tolerance": 1e-12, "absolute_tolerance": 0.0,
           "maximum_iterations": 4000, "monitor_convergence": False}
    v, z = np.zeros((m, nx)), np.zeros((m, nx))
    res = osys.solve(v, z, b[:m], b[m:], solver_parameters=sp_, pc_fn=_spd_jacobi(p))
    assert res.reason == ko.CONVERGED_RTOL
    exact = spla.spsolve(global_matrix(p).tocsc(), b.ravel())
    assert common.rel_err(np.concatenate([v.ravel(), z.ravel()]), exact) < 1e-8

    def negative(u_0, u_1, b_0, b_1):
        u_0[:] = -b_0
        u_1[:] = -b_1
    res = osys.solve(np.zeros((m, nx)), np.zeros((m, nx)), b[:m], b[m:],
                     solver_parameters=dict(sp_, preconditioner=True), pc_fn=negative)
    assert res.reason == ko.DIVERGED_INDEFINITE_PC and res.its == 0
