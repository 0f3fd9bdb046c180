"""``Instationary.linear_solve`` end to end: right-hand sides, lifting of
inhomogeneous Dirichlet data, the solve, the post-solve assembly of the levels.

Data: the reference's manufactured heat-control problems (``test/test_control.py:1658-1826``
BE, ``1983-2137`` CN): exact solution linear in time, so both schemes are exact in time and
the P1 error falls with second order in h.  The reference prints the orders without
asserting them; here they are asserted.
"""
import numpy as np
import pytest

import common


@pytest.mark.parametrize("CN", [False, True])
def test_mms_heat_control_orders_with_the_oracle(CN):
    errs = []
    for N in (4, 8, 16):
        ctl, disc, ref_v, ref_zeta = common.mms_heat_control(N, CN)
        ksp = ctl.linear_solve(solver_parameters=common.MMS_SOLVER_PARAMETERS,
                               lambda_v_bounds=(0.5, 2.0), backend=common.OracleBackend())
        assert ksp.reason > 0
        # boundary data are back on the state, the adjoint vanishes there
        assert np.all(ctl._v[:, disc.boundary] == 1.0)
        assert np.all(ctl._zeta[:, disc.boundary] == 0.0)
        errs.append(common.mms_errors(ctl, disc, ref_v, ref_zeta))
    errs = np.array(errs)
    orders = np.log(errs[:-1] / errs[1:]) / np.log(2.0)
    # 4 -> 8 is pre-asymptotic (1.77 / 1.81 measured), 8 -> 16 gives 1.94 / 1.95
    assert orders[0].min() > 1.7 and orders[1].min() > 1.9, (errs, orders)


@pytest.mark.gpu
@pytest.mark.parametrize("CN", [False, True])
def test_mms_heat_control_on_the_gpu(CN):
    from control_amd.control import GpuBackend
    N = 16
    ctl, disc, ref_v, ref_zeta = common.mms_heat_control(N, CN)
    ksp = ctl.linear_solve(solver_parameters=common.MMS_SOLVER_PARAMETERS,
                           lambda_v_bounds=(0.5, 2.0), backend=GpuBackend(schur=(30, 0.02, 2.2)))
    assert ksp.getConvergedReason() > 0
    ref, *_ = common.mms_heat_control(N, CN)
    ref.linear_solve(solver_parameters=common.MMS_SOLVER_PARAMETERS,
                     lambda_v_bounds=(0.5, 2.0), backend=common.OracleBackend())
    assert np.abs(ctl._v - ref._v).max() < 1e-7
    assert np.abs(ctl._zeta - ref._zeta).max() < 1e-7
    ev, ez = common.mms_errors(ctl, disc, ref_v, ref_zeta)
    assert ev < 2e-2 and ez < 2e-2


@pytest.mark.parametrize("CN", [False, True])
def test_time_discretisation_orders_with_the_oracle(CN):
    """``test/test_control.py:1829-1980`` (BE) / ``2140-2294`` (CN): exact solution exponential in
    time.  The reference compares with the exact solution on ``N = 250`` for ``n_t = 4 .. 32``;
    here the time grids are nested (``n_t = 5, 9, 17, 33``: the step halves and the coarse levels
    are levels of the fine grid) on ``N = 24`` and the order is read off the differences of
    successive solutions at the common levels, in which the spatial error cancels: first order
    for BE (0.84 / 0.92 then 0.92 / 0.95 measured for state / adjoint), second for CN (2.21 /
    2.19 then 2.04 / 2.06).  This is the check that sees a wrong factor of tau or a wrong
    ``T_1`` / ``T_2`` in the rows: the linear-in-time problems above are exact in time for
    both schemes."""
    sols = {}
    for n_t in (5, 9, 17, 33):
        ctl, disc, _, _ = common.mms_heat_control_in_time(24, CN, n_t)
        ksp = ctl.linear_solve(solver_parameters=common.MMS_SOLVER_PARAMETERS,
                               lambda_v_bounds=(0.5, 2.0),
                               backend=common.OracleBackend(schur=(40, 0.01, 2.2)))
        assert ksp.reason > 0
        sols[n_t] = (ctl._v.copy(), ctl._zeta.copy())
    diffs = []
    for a, b in ((5, 9), (9, 17), (17, 33)):
        tau = 2.0 / (a - 1)
        diffs.append([np.sqrt(tau * sum(x @ (disc.M @ x) for x in sols[a][k] - sols[b][k][::2]))
                      for k in (0, 1)])
    diffs = np.array(diffs)
    orders = np.log(diffs[:-1] / diffs[1:]) / np.log(2.0)
    if CN:
        assert orders.min() > 1.9 and orders.max() < 2.4, orders
    else:
        assert orders.min() > 0.8 and orders.max() < 1.1, orders


@pytest.mark.gpu
@pytest.mark.parametrize("CN", [False, True])
def test_time_dependent_mms_on_the_gpu(CN):
    from control_amd.control import GpuBackend
    out = []
    for be in (GpuBackend(schur=(40, 0.01, 2.2)), common.OracleBackend(schur=(40, 0.01, 2.2))):
        ctl, disc, ref_v, ref_zeta = common.mms_heat_control_in_time(24, CN, 9)
        ksp = ctl.linear_solve(solver_parameters=common.MMS_SOLVER_PARAMETERS,
                               lambda_v_bounds=(0.5, 2.0), backend=be)
        assert ksp.reason > 0
        out.append((ctl._v.copy(), ctl._zeta.copy()))
    assert np.abs(out[0][0] - out[1][0]).max() < 1e-7
    assert np.abs(out[0][1] - out[1][1]).max() < 1e-7
    ev, ez = common.mms_errors(ctl, disc, ref_v, ref_zeta, n_t=9)
    assert ev < (0.02 if CN else 0.1) and ez < (0.04 if CN else 0.3)   # 0.010 / 0.022, 0.044 / 0.123


@pytest.mark.parametrize("CN", [False, True])
def test_mms_convection_diffusion_control_orders_with_the_oracle(CN):
    """``test/test_control.py:2297-2493`` (BE; the CN run keeps the BE data, see the helper):
    a non-symmetric forward operator that changes with the time level.  P1 orders asserted
    (the reference prints them)."""
    errs = []
    for N in (4, 8, 16):
        ctl, disc, ref_v, ref_zeta = common.mms_convection_diffusion_control(N, CN)
        ksp = ctl.linear_solve(solver_parameters=common.MMS_SOLVER_PARAMETERS,
                               lambda_v_bounds=(0.5, 2.0), backend=common.OracleBackend())
        assert ksp.reason > 0
        assert np.all(ctl._v[:, disc.boundary] == 1.0)
        assert np.all(ctl._zeta[:, disc.boundary] == 0.0)
        errs.append(common.mms_errors(ctl, disc, ref_v, ref_zeta))
    errs = np.array(errs)
    orders = np.log(errs[:-1] / errs[1:]) / np.log(2.0)
    # measured 1.77 / 1.81 (4 -> 8) and 1.94 / 1.95 (8 -> 16), BE and CN alike
    assert orders[0].min() > 1.7 and orders[1].min() > 1.9, (errs, orders)


@pytest.mark.gpu
@pytest.mark.parametrize("CN", [False, True])
def test_mms_convection_diffusion_control_on_the_gpu(CN):
    """Non-symmetric, level-dependent blocks through the C-ABI: the transposed blocks of the
    adjoint rows and the backward sweep of the preconditioner are genuinely different
    matrices here."""
    from control_amd.control import GpuBackend
    N = 16
    ctl, disc, ref_v, ref_zeta = common.mms_convection_diffusion_control(N, CN)
    ksp = ctl.linear_solve(solver_parameters=common.MMS_SOLVER_PARAMETERS,
                           lambda_v_bounds=(0.5, 2.0), backend=GpuBackend(schur=(30, 0.02, 2.2)))
    assert ksp.getConvergedReason() > 0
    ref, *_ = common.mms_convection_diffusion_control(N, CN)
    ref.linear_solve(solver_parameters=common.MMS_SOLVER_PARAMETERS,
                     lambda_v_bounds=(0.5, 2.0), backend=common.OracleBackend())
    assert np.abs(ctl._v - ref._v).max() < 1e-7
    assert np.abs(ctl._zeta - ref._zeta).max() < 1e-7
    ev, ez = common.mms_errors(ctl, disc, ref_v, ref_zeta)
    assert ev < 2e-2 and ez < 2e-2


@pytest.mark.parametrize("CN", [False, True])
def test_instationary_stokes_control_with_exact_sol_oracle(CN):
    """``test/test_control.py:3045-3172`` / ``3175-3302`` (the reference runs these without
    asserting anything): the driver's right-hand sides and Dirichlet lifting reproduce the
    exact velocity up to the discretisation error of the scheme."""
    ctl, th, true_v = common.stokes_exact_sol_control(CN, n=4, n_t=8)
    ksp = ctl.incompressible_linear_solve(lambda_v_bounds=(0.25, 1.5625),
                                          lambda_p_bounds=(0.25, 2.25),
                                          backend=common.OracleBackend())
    assert ksp.reason > 0
    tau = 1.0 / 7.0
    err = 0.0
    for i in range(8):
        d = ctl._v[i] - true_v(th.coords_v, i * tau)
        err += tau * (d @ (th.M_v @ d))
    # measured on this coarse instance (4x4 Q2-Q1, 8 levels): 4.6e-3 (BE), 5.6e-3 (CN)
    assert np.sqrt(err) < 1e-2
    # boundary data of every level are on the state; the adjoint vanishes there
    for i in range(8):
        assert np.array_equal(ctl._v[i, th.boundary_v], true_v(th.coords_v, i * tau)[th.boundary_v])
    assert np.all(ctl._zeta[:, th.boundary_v] == 0.0)


def _stokes_mms_errors(ctl, th, true_v, true_zeta, n_t=10, t_f=2.0):
    tau = t_f / (n_t - 1.0)
    ev = ez = 0.0
    for i in range(n_t):
        d = ctl._v[i] - true_v(th.coords_v, i * tau)
        ev += tau * (d @ (th.M_v @ d))
        d = ctl._zeta[i] - true_zeta(th.coords_v, i * tau)
        ez += tau * (d @ (th.M_v @ d))
    return np.sqrt(ev), np.sqrt(ez)


STOKES_MMS_BOUNDS = dict(lambda_v_bounds=(0.3924, 2.0598), lambda_p_bounds=(0.5, 2.0))   # :3476


@pytest.mark.parametrize("CN", [False, True])
def test_mms_instationary_stokes_control_orders_with_the_oracle(CN):
    """``test/test_control.py:3305-3543`` (BE) / ``3754-3962`` (CN), Taylor-Hood P2-P1 (degree 2
    of the reference's 2 and 3), beta = 1e-3, time-dependent inhomogeneous Dirichlet data.
    The reference prints the orders; here: at least third order for velocity and adjoint
    between N = 4 and N = 8 (measured against the nodal interpolant of the exact solution:
    3.84 / 3.91 BE, 3.72 / 3.97 CN)."""
    errs = []
    for N in ((2, 4, 8) if not CN else (4, 8)):
        ctl, th, true_v, true_zeta = common.mms_stokes_control_instationary(N, CN)
        ksp = ctl.incompressible_linear_solve(solver_parameters=common.MMS_SOLVER_PARAMETERS,
                                              backend=common.OracleBackend(schur=(40, 0.01, 2.3)),
                                              **STOKES_MMS_BOUNDS)
        assert ksp.reason > 0
        errs.append(_stokes_mms_errors(ctl, th, true_v, true_zeta))
    errs = np.array(errs)
    orders = np.log(errs[:-1] / errs[1:]) / np.log(2.0)
    assert orders.min() > 3.0, (errs, orders)
    assert errs[-1][0] < 1e-3 and errs[-1][1] < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("CN", [False, True])
def test_mms_instationary_stokes_control_on_the_gpu(CN):
    from control_amd.control import GpuBackend
    out = []
    for be in (GpuBackend(schur=(40, 0.01, 2.3)), common.OracleBackend(schur=(40, 0.01, 2.3))):
        ctl, th, true_v, true_zeta = common.mms_stokes_control_instationary(8, CN)
        ksp = ctl.incompressible_linear_solve(solver_parameters=common.MMS_SOLVER_PARAMETERS,
                                              backend=be, **STOKES_MMS_BOUNDS)
        assert ksp.reason > 0
        out.append((ctl._v.copy(), ctl._zeta.copy(), ctl._p.copy()))
    assert np.abs(out[0][0] - out[1][0]).max() < 1e-7
    assert np.abs(out[0][1] - out[1][1]).max() < 1e-8
    ev, ez = _stokes_mms_errors(ctl, th, true_v, true_zeta)
    assert ev < 1e-3 and ez < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("CN", [False, True])
def test_stokes_control_time_discretisation_orders_on_the_gpu(CN):
    """``test/test_control.py:3546-3751`` (BE) / ``3965-4168`` (CN): Stokes control with the
    exact solution ``exp(T - t) (x y^3, (x^4 - y^4) / 4)``, Taylor-Hood P2-P1, beta = 1e-3,
    T = 2, time-dependent Dirichlet data.  The reference runs N = 100 and prints the orders;
    here N = 32 (spatial error 1e-5, below the time errors compared) with n_t = 5, 9, 17 and the
    error against the exact velocity: first order for BE (0.89, 0.94 measured), second for
    CN (2.41, 1.91) -- the check of the incompressible driver's CN rows (``T_1`` / ``T_2`` on
    velocity and pressure blocks, ``sub_n_blocks``) with a real time-discretisation error.
    GPU only: the oracle's nested preconditioner needs minutes at this size."""
    from control_amd.control import GpuBackend
    errs = []
    for n_t in (5, 9, 17):
        ctl, th, true_v = common.stokes_exact_sol_control(CN, n=32, n_t=n_t, T_f=2.0, beta=1e-3,
                                                          taylor_hood=True)
        ksp = ctl.incompressible_linear_solve(solver_parameters=common.MMS_SOLVER_PARAMETERS,
                                              backend=GpuBackend(schur=(60, 0.002, 2.3)),
                                              **STOKES_MMS_BOUNDS)
        assert ksp.reason > 0
        tau = 2.0 / (n_t - 1.0)
        e2 = 0.0
        for i in range(n_t):
            d = ctl._v[i] - true_v(th.coords_v, i * tau)
            e2 += tau * (d @ (th.M_v @ d))
        errs.append(np.sqrt(e2))
    orders = np.log(np.array(errs[:-1]) / np.array(errs[1:])) / np.log(2.0)
    if CN:
        assert orders.min() > 1.7, (errs, orders)
    else:
        assert 0.8 < orders.min() and orders.max() < 1.2, (errs, orders)


@pytest.mark.gpu
@pytest.mark.parametrize("CN", [False, True])
def test_instationary_stokes_control_with_exact_sol_gpu(CN):
    from control_amd.control import GpuBackend
    sp_ = {"linear_solver": "fgmres", "fgmres_restart": 10, "maximum_iterations": 200,
           "relative_tolerance": 1.0e-9, "absolute_tolerance": 0.0, "monitor_convergence": False}
    ctl, th, true_v = common.stokes_exact_sol_control(CN, n=4, n_t=8)
    ksp = ctl.incompressible_linear_solve(lambda_v_bounds=(0.25, 1.5625),
                                          lambda_p_bounds=(0.25, 2.25), solver_parameters=sp_,
                                          backend=GpuBackend(schur=(30, 0.02, 2.2)))
    assert ksp.getConvergedReason() > 0
    ref, *_ = common.stokes_exact_sol_control(CN, n=4, n_t=8)
    ref.incompressible_linear_solve(lambda_v_bounds=(0.25, 1.5625),
                                    lambda_p_bounds=(0.25, 2.25), solver_parameters=sp_,
                                    backend=common.OracleBackend())
    assert np.abs(ctl._v - ref._v).max() < 1e-6
    assert np.abs(ctl._zeta - ref._zeta).max() < 1e-6


# ------------------------------------------------------------------ Control.Stationary

KAT_SP = {"linear_solver": "fgmres", "fgmres_restart": 10, "maximum_iterations": 500,
          "relative_tolerance": 1.0e-14, "absolute_tolerance": 1.0e-14,
          "monitor_convergence": False}


def _stationary_reference_problem(nonlinear=False):
    """``test/test_control.py:554-600`` (linear) / ``710-765`` (Picard): P1 on
    ``UnitSquareMesh(8, 8)``, ``-lapl(v) + alpha(v) v = m``, beta = 1, desired state
    ``sin(pi x) sin(pi y) exp(x + y)``, homogeneous Dirichlet conditions."""
    from control_amd.control import Stationary
    from control_amd.fem import unit_square_p1
    disc = unit_square_p1(8)

    def v_d(X):
        return np.sin(np.pi * X[:, 0]) * np.sin(np.pi * X[:, 1]) * np.exp(X[:, 0] + X[:, 1])
    if nonlinear:
        def forward(v_old):     # grad-grad + (2 + 0.5 v_old^2) mass, test_control.py:715-719
            return disc.K + disc.weighted_mass(
                lambda lam, cells: 2.0 + 0.5 * (v_old[cells] @ lam.T) ** 2)
    else:
        def forward(v_old):     # test_control.py:559-563
            return disc.K + 2.0 * disc.M
    return Stationary(disc, forward, desired_state=v_d, beta=1.0), disc, v_d


def test_stationary_linear_control_against_the_reduced_problem():
    """``test/test_control.py:554-707``: the reference checks the KKT solve against an
    independent minimisation of the reduced functional (L-BFGS through tlm_adjoint) with the
    bars 1e-8 (state) and 1e-6 (control).  Here the reduced problem -- quadratic for the
    linear state equation -- is solved exactly with dense algebra: same bars."""
    ctl, disc, v_d = _stationary_reference_problem()
    ksp = ctl.linear_solve(solver_parameters=KAT_SP, lambda_v_bounds=(0.5, 2.0),
                           backend=common.OracleBackend(schur=(40, 0.02, 2.2)))
    assert ksp.reason > 0
    inner = np.setdiff1d(np.arange(disc.n_dofs), disc.boundary)
    M = disc.M.toarray()[np.ix_(inner, inner)]
    A = (disc.K + 2.0 * disc.M).toarray()[np.ix_(inner, inner)]
    S = np.linalg.solve(A, M)                       # u = S m
    r = v_d(disc.coords)[inner]
    beta = 1.0                                      # J = |u - r|^2_M + beta^2 |m|^2_M
    m = np.linalg.solve(S.T @ M @ S + beta**2 * M, S.T @ (M @ r))
    u = S @ m

    def l2(e):
        return np.sqrt(abs(e @ (M @ e)))
    assert l2(ctl._v[inner] - u) < 1.0e-8           # test_control.py:701
    assert l2(ctl._zeta[inner] / beta - m) < 1.0e-6  # test_control.py:706
    assert np.all(ctl._v[disc.boundary] == 0.0) and np.all(ctl._zeta[disc.boundary] == 0.0)


def test_stationary_picard_loop_with_the_oracle():
    """``Stationary.non_linear_solve`` (``control.py:640-760``) on the reference's non-linear
    reaction problem: the residual norms fall monotonically and the fixed point satisfies the
    Picard optimality system ``M v + D(v)^T zeta = M v_d``, ``D(v) v = M zeta / beta``."""
    ctl, disc, v_d = _stationary_reference_problem(nonlinear=True)
    norms = ctl.non_linear_solve(solver_parameters=KAT_SP, lambda_v_bounds=(0.5, 2.0),
                                 max_non_linear_iter=30, relative_non_linear_tol=1.0e-10,
                                 backend=common.OracleBackend(schur=(40, 0.02, 2.2)))
    assert norms[-1] <= 1.0e-10 * norms[0]
    assert all(b < a for a, b in zip(norms, norms[1:]))
    D = ctl.construct_D_v(ctl._v)
    r0 = disc.M @ v_d(disc.coords) - disc.M @ ctl._v - D.T @ ctl._zeta
    r1 = -(D @ ctl._v) + disc.M @ ctl._zeta
    r0[disc.boundary] = r1[disc.boundary] = 0.0
    assert max(np.abs(r0).max(), np.abs(r1).max()) < 1.0e-9
    # the reference's own check (test_control.py:766-860): against the minimiser of the reduced
    # functional, 1e-8 on the state and 1e-6 on the control
    _, _, _, forward, jacobian = _gauss_newton_reference_problem()
    inner, Mi, u, m = _reduced_nonlinear_optimum(disc, v_d, forward, jacobian)
    assert np.sqrt(abs((ctl._v - u)[inner] @ (Mi @ (ctl._v - u)[inner]))) < 1.0e-8
    assert np.sqrt(abs((ctl._zeta[inner] - m) @ (Mi @ (ctl._zeta[inner] - m)))) < 1.0e-6


def _gauss_newton_reference_problem():
    """``test/test_control.py:866-930``: the forward form is the non-linear residual
    ``grad v . grad w + (2 + 0.5 v^2) v w``; with ``set_Gauss_Newton()`` the operator is its
    derivative ``grad-grad + (2 + 1.5 v^2) mass`` (``ufl.derivative``, ``control.py:314-320``)."""
    from control_amd.control import Stationary
    from control_amd.fem import unit_square_p1
    disc = unit_square_p1(8)

    def v_d(X):
        return np.sin(np.pi * X[:, 0]) * np.sin(np.pi * X[:, 1]) * np.exp(X[:, 0] + X[:, 1])

    def weighted(v, a, b):
        return disc.weighted_mass(lambda lam, cells: a + b * (v[cells] @ lam.T) ** 2)

    def forward(v):
        return disc.K + weighted(v, 2.0, 0.5)

    def jacobian(v):
        return disc.K + weighted(v, 2.0, 1.5)
    ctl = Stationary(disc, forward, desired_state=v_d, beta=1.0, forward_jacobian=jacobian)
    return ctl, disc, v_d, forward, jacobian


def _reduced_nonlinear_optimum(disc, v_d, forward, jacobian):
    """The reference's independent answer (``test_control.py:945-1014``): L-BFGS-B on the reduced
    functional ``|u(m) - r|^2 + beta^2 |m|^2`` (beta = 1) with ``u(m)`` from a Newton solve of
    ``-lapl u + (2 + 0.5 u^2) u = m``; the gradient comes from the adjoint equation instead of
    tlm_adjoint."""
    import scipy.optimize as so
    import scipy.sparse.linalg as spla
    inner = np.setdiff1d(np.arange(disc.n_dofs), disc.boundary)
    Mi = disc.M.tocsr()[inner][:, inner]
    r = v_d(disc.coords)

    def state(m_i):
        u = np.zeros(disc.n_dofs)
        for _ in range(50):
            res = (forward(u) @ u)[inner] - Mi @ m_i
            du = spla.spsolve(jacobian(u).tocsr()[inner][:, inner].tocsc(), -res)
            u[inner] += du
            if np.abs(du).max() < 1e-14:
                break
        return u

    def functional(m_i):
        u = state(m_i)
        e = (u - r)[inner]
        lam = spla.spsolve(jacobian(u).tocsr()[inner][:, inner].T.tocsc(), -2.0 * (Mi @ e))
        return e @ (Mi @ e) + m_i @ (Mi @ m_i), 2.0 * (Mi @ m_i) - Mi @ lam
    res = so.minimize(functional, np.zeros(len(inner)), jac=True, method="L-BFGS-B",
                      options={"ftol": 0.0, "gtol": 1.0e-12, "maxiter": 2000})
    return inner, Mi, state(res.x), res.x


def _check_gauss_newton_against_the_reduced_problem(backend):
    ctl, disc, v_d, forward, jacobian = _gauss_newton_reference_problem()
    ctl.set_Gauss_Newton()
    norms = ctl.non_linear_solve(solver_parameters=KAT_SP, lambda_v_bounds=(0.5, 2.0),
                                 max_non_linear_iter=100, relative_non_linear_tol=1.0e-8,
                                 backend=backend)
    assert norms[-1] <= 1.0e-8 * norms[0]
    inner, Mi, u, m = _reduced_nonlinear_optimum(disc, v_d, forward, jacobian)

    def l2(e):
        return np.sqrt(abs(e @ (Mi @ e)))
    assert l2((ctl._v - u)[inner]) < 1.0e-8            # test_control.py:1019
    assert l2(ctl._zeta[inner] / 1.0 - m) < 1.0e-6     # test_control.py:1024 (control = zeta / beta)
    return ctl


def test_GN_stationary_non_linear_control_with_reference_sol():
    """``test/test_control.py:866-1024`` restated for P1 (the reference also runs P2 and P3):
    Gauss-Newton on the non-linear reaction problem lands on the minimiser of the reduced
    functional, same bars (1e-8 state, 1e-6 control)."""
    _check_gauss_newton_against_the_reduced_problem(common.OracleBackend(schur=(40, 0.02, 2.2)))


@pytest.mark.gpu
def test_GN_stationary_non_linear_control_with_reference_sol_on_the_gpu():
    from control_amd.control import GpuBackend
    _check_gauss_newton_against_the_reduced_problem(GpuBackend(schur=(40, 0.02, 2.2)))


@pytest.mark.gpu
def test_stationary_drivers_on_the_gpu():
    from control_amd.control import GpuBackend
    for nonlinear in (False, True):
        ctl, disc, _ = _stationary_reference_problem(nonlinear)
        ref, _, _ = _stationary_reference_problem(nonlinear)
        for c, be in ((ctl, GpuBackend(schur=(40, 0.02, 2.2))),
                      (ref, common.OracleBackend(schur=(40, 0.02, 2.2)))):
            if nonlinear:
                c.non_linear_solve(solver_parameters=KAT_SP, lambda_v_bounds=(0.5, 2.0),
                                   max_non_linear_iter=30, relative_non_linear_tol=1.0e-10,
                                   backend=be)
            else:
                c.linear_solve(solver_parameters=KAT_SP, lambda_v_bounds=(0.5, 2.0), backend=be)
        assert np.abs(ctl._v - ref._v).max() < 1e-10
        assert np.abs(ctl._zeta - ref._zeta).max() < 1e-9


# ------------------------------------------------------ Instationary.non_linear_solve

def _reaction_heat_control(CN, n=8, n_t=5):
    """Instationary version of the reference's non-linear reaction problem
    (``test/test_control.py:715-719``): ``forward_form = grad-grad + (2 + 0.5 v_old^2) mass``."""
    from control_amd.control import Instationary
    from control_amd.fem import unit_square_p1
    disc = unit_square_p1(n)

    def forward(v_old, t):
        return disc.K + disc.weighted_mass(
            lambda lam, cells: 2.0 + 0.5 * (v_old[cells] @ lam.T) ** 2)

    def v_d(X, t):
        return (1.0 + t) * np.sin(np.pi * X[:, 0]) * np.sin(np.pi * X[:, 1]) * np.exp(X[:, 0])
    return Instationary(disc, forward, desired_state=v_d, beta=1.0e-2, CN=CN, n_t=n_t,
                        time_interval=(0.0, 1.0)), disc


@pytest.mark.parametrize("CN", [False, True])
def test_instationary_picard_loop_with_the_oracle(CN):
    ctl, disc = _reaction_heat_control(CN)
    norms = ctl.non_linear_solve(solver_parameters=KAT_SP, lambda_v_bounds=(0.5, 2.0),
                                 max_non_linear_iter=30, relative_non_linear_tol=1.0e-9,
                                 absolute_non_linear_tol=0.0,
                                 backend=common.OracleBackend(schur=(40, 0.02, 2.2)))
    assert norms[-1] <= 1.0e-9 * norms[0]
    assert all(b < a for a, b in zip(norms[1:], norms[2:]))
    # the driver's own residual at the returned fields is the last norm it reported
    r0, r1 = ctl.non_linear_res_eval(ctl._v, ctl._zeta, np.zeros(disc.n_dofs),
                                     ctl.construct_v_d(), ctl.construct_f())
    assert abs(np.sqrt(np.vdot(r0, r0) + np.vdot(r1, r1)) - norms[-1]) <= 1e-12 + 1e-6 * norms[-1]


@pytest.mark.gpu
@pytest.mark.parametrize("CN", [False, True])
def test_instationary_picard_loop_on_the_gpu(CN):
    from control_amd.control import GpuBackend
    out = []
    for be in (GpuBackend(schur=(40, 0.02, 2.2)), common.OracleBackend(schur=(40, 0.02, 2.2))):
        ctl, _ = _reaction_heat_control(CN)
        norms = ctl.non_linear_solve(solver_parameters=KAT_SP, lambda_v_bounds=(0.5, 2.0),
                                     max_non_linear_iter=30, relative_non_linear_tol=1.0e-9,
                                     absolute_non_linear_tol=0.0, backend=be)
        out.append((norms, ctl._v.copy(), ctl._zeta.copy()))
    assert len(out[0][0]) == len(out[1][0])
    assert np.abs(out[0][1] - out[1][1]).max() < 1e-9
    assert np.abs(out[0][2] - out[1][2]).max() < 1e-9


@pytest.mark.parametrize("CN", [False, True])
def test_instationary_gauss_newton_reaches_a_smaller_residual_than_it_starts_from(CN):
    """``Instationary.set_Gauss_Newton`` (``control.py:1835-1836``, ``construct_D_v``
    ``:1887-1896``): every ``D_v`` of the driver -- operator blocks, boundary lifting and the
    residual -- becomes the Gateaux derivative of the forward form.  For the cubic reaction
    term ``(2 + 0.5 v^2) v`` that is ``grad-grad + (2 + 1.5 v^2) mass``; the loop is the
    reference's (only the stationary variant is exercised by its tests,
    ``test/test_control.py:921``), so the checks are structural: the switch changes the
    blocks, refuses to engage without a derivative, and the loop still contracts."""
    from control_amd.control import Instationary
    ctl, disc = _reaction_heat_control(CN)
    with pytest.raises(ValueError):
        ctl.set_Gauss_Newton()

    def jacobian(v_old, t):
        return disc.K + disc.weighted_mass(
            lambda lam, cells: 2.0 + 1.5 * (v_old[cells] @ lam.T) ** 2)
    picard = ctl._forward
    ctl = Instationary(disc, picard, desired_state=ctl._desired_state, beta=1.0e-2, CN=CN,
                       n_t=5, time_interval=(0.0, 1.0), forward_jacobian=jacobian)
    v = np.linspace(0.0, 1.0, disc.n_dofs)
    A_picard = ctl.construct_D_v(v, 0.0)
    ctl.set_Gauss_Newton()
    A_newton = ctl.construct_D_v(v, 0.0)
    assert abs(A_newton - A_picard).max() > 1e-6
    assert abs(A_newton - jacobian(v, 0.0)).max() == 0.0
    norms = ctl.non_linear_solve(solver_parameters=KAT_SP, lambda_v_bounds=(0.5, 2.0),
                                 max_non_linear_iter=8, relative_non_linear_tol=1.0e-6,
                                 absolute_non_linear_tol=0.0,
                                 backend=common.OracleBackend(schur=(40, 0.02, 2.2)))
    assert len(norms) >= 2 and norms[-1] < 0.2 * norms[0]
    ctl.set_Gauss_Newton(False)
    assert abs(ctl.construct_D_v(v, 0.0) - A_picard).max() == 0.0


def _mms_poisson_control(N):
    """``test/test_control.py:122-230``: stationary Poisson control, P1 on
    ``UnitSquareMesh(N, N)``, beta = 1e-3, manufactured state and adjoint."""
    from control_amd.control import Stationary
    from control_amd.fem import unit_square_p1
    disc = unit_square_p1(N)
    beta = 1.0e-3

    def ref_v(X):
        return np.sin(np.pi * X[:, 0]) * np.sin(np.pi * X[:, 1]) * np.exp(X[:, 0] + X[:, 1])

    def ref_zeta(X):
        return np.sin(2 * np.pi * X[:, 0]) * np.sin(2 * np.pi * X[:, 1])

    def lapl_v(X):      # div(grad(sin(pi x) sin(pi y) exp(x + y)))
        x, y = X[:, 0], X[:, 1]
        sx, cx, sy, cy = np.sin(np.pi * x), np.cos(np.pi * x), np.sin(np.pi * y), np.cos(np.pi * y)
        e = np.exp(x + y)
        return e * (2.0 * (1.0 - np.pi**2) * sx * sy + 2.0 * np.pi * (cx * sy + sx * cy))

    def v_d(X):         # -lapl(zeta) + v
        return 8.0 * np.pi**2 * ref_zeta(X) + ref_v(X)

    def f(X):           # -lapl(v) - zeta / beta
        return -lapl_v(X) - ref_zeta(X) / beta
    return Stationary(disc, desired_state=v_d, force_f=f, beta=beta), disc, ref_v, ref_zeta


MMS_POISSON_SP = {"linear_solver": "fgmres", "fgmres_restart": 10, "maximum_iterations": 500,
                  "relative_tolerance": 1.0e-6, "absolute_tolerance": 1.0e-6,
                  "monitor_convergence": False}


def test_mms_stationary_poisson_control_orders():
    """Second order in h for state and adjoint (the reference prints the orders)."""
    errs = []
    for N in (8, 16, 32):
        ctl, disc, ref_v, ref_zeta = _mms_poisson_control(N)
        ksp = ctl.linear_solve(solver_parameters=MMS_POISSON_SP, lambda_v_bounds=(0.5, 2.0),
                               backend=common.OracleBackend(schur=(40, 0.01, 2.2)))
        assert ksp.reason > 0
        dv, dz = ctl._v - ref_v(disc.coords), ctl._zeta - ref_zeta(disc.coords)
        errs.append((np.sqrt(dv @ (disc.M @ dv)), np.sqrt(dz @ (disc.M @ dz))))
    errs = np.array(errs)
    orders = np.log(errs[:-1] / errs[1:]) / np.log(2.0)
    # 8 -> 16 is pre-asymptotic (1.67 / 1.78 measured), 16 -> 32 gives 1.91 / 1.95
    assert orders[0].min() > 1.6 and orders[1].min() > 1.9, (errs, orders)


@pytest.mark.gpu
def test_mms_stationary_poisson_control_on_the_gpu():
    from control_amd.control import GpuBackend
    ctl, disc, ref_v, ref_zeta = _mms_poisson_control(32)
    ksp = ctl.linear_solve(solver_parameters=MMS_POISSON_SP, lambda_v_bounds=(0.5, 2.0),
                           backend=GpuBackend(schur=(40, 0.01, 2.2)))
    assert ksp.getConvergedReason() > 0
    dv, dz = ctl._v - ref_v(disc.coords), ctl._zeta - ref_zeta(disc.coords)
    # discretisation errors at N = 32 (oracle backend: 6.2e-2 / 4.4e-3)
    assert np.sqrt(dv @ (disc.M @ dv)) < 8e-2 and np.sqrt(dz @ (disc.M @ dz)) < 6e-3


# ------------------------------------------------ Stationary.incompressible_linear_solve

def _mms_stokes_control(N):
    """``test/test_control.py:361-553``: stationary Stokes control, P2-P1 on
    ``RectangleMesh(N, N, 2, 2)``, beta = 1e-3, exact (v, p, zeta, mu), inhomogeneous Dirichlet
    data v = v_exact on the boundary."""
    from control_amd.control import Stationary
    from control_amd.fem import rectangle_p2p1
    th = rectangle_p2p1(N, N, 2.0, 2.0)
    beta = 1.0e-3

    def XY(P):
        return P[:, 0] - 1.0, P[:, 1] - 1.0

    def v_ex(P):
        x, y = XY(P)
        return np.concatenate([x * y**3, 0.25 * (x**4 - y**4)])

    def p_ex(P):
        x, y = XY(P)
        return 3.0 * x**2 * y - y**3

    def zeta_ex(P):
        x, y = XY(P)
        return np.concatenate([beta * 2.0 * y * (x**2 - 1.0)**2 * (y**2 - 1.0),
                               -beta * 2.0 * x * (x**2 - 1.0) * (y**2 - 1.0)**2])

    def mu_ex(P):
        x, y = XY(P)
        return beta * 4.0 * x * y

    def v_d(P):        # -lapl(zeta) + grad(mu) + v
        x, y = XY(P)
        lz1 = 2.0 * beta * (y * (y**2 - 1.0) * (12.0 * x**2 - 4.0) + 6.0 * y * (x**2 - 1.0)**2)
        lz2 = -2.0 * beta * (6.0 * x * (y**2 - 1.0)**2 + x * (x**2 - 1.0) * (12.0 * y**2 - 4.0))
        return np.concatenate([-lz1 + 4.0 * beta * y, -lz2 + 4.0 * beta * x]) + v_ex(P)

    def f(P):          # -lapl(v) + grad(p) - zeta / beta, and -lapl(v) + grad(p) = 0
        return -zeta_ex(P) / beta
    ctl = Stationary(th, desired_state=v_d, force_f=f, beta=beta, bcs_v=v_ex)
    return ctl, th, v_ex, p_ex, zeta_ex, mu_ex


MMS_STOKES_SP = {"linear_solver": "fgmres", "fgmres_restart": 10, "maximum_iterations": 200,
                 "relative_tolerance": 1.0e-10, "absolute_tolerance": 1.0e-10,
                 "monitor_convergence": False}


def _stokes_errors(ctl, th, v_ex, p_ex, zeta_ex, mu_ex):
    def l2(M, e):
        return np.sqrt(abs(e @ (M @ e)))

    def demean(M, q):
        return q - (np.ones_like(q) @ (M @ q)) / (np.ones_like(q) @ (M @ np.ones_like(q)))
    return (l2(th.M_v, ctl._v - v_ex(th.coords_v)), l2(th.M_v, ctl._zeta - zeta_ex(th.coords_v)),
            l2(th.M_p, demean(th.M_p, ctl._p) - demean(th.M_p, p_ex(th.coords_p))),
            l2(th.M_p, demean(th.M_p, ctl._mu) - demean(th.M_p, mu_ex(th.coords_p))))


def test_mms_stationary_stokes_control_orders():
    """Third order for the P2 velocities, second for the P1 pressures (the reference prints
    the orders); exercises the lifting of inhomogeneous Dirichlet data in all three rows."""
    errs = []
    for N in (4, 8, 16):
        ctl, th, *ex = _mms_stokes_control(N)
        ksp = ctl.incompressible_linear_solve(
            solver_parameters=MMS_STOKES_SP, lambda_v_bounds=(0.3924, 2.0598),
            lambda_p_bounds=(0.5, 2.0), backend=common.OracleBackend(schur=(40, 0.01, 2.25)))
        assert ksp.reason > 0
        errs.append(_stokes_errors(ctl, th, *ex))
    errs = np.array(errs)
    orders = np.log(errs[:-1] / errs[1:]) / np.log(2.0)
    assert orders[:, :2].min() > 2.6 and orders[:, 2:].min() > 1.7, (errs, orders)


@pytest.mark.gpu
def test_mms_stationary_stokes_control_on_the_gpu():
    from control_amd.control import GpuBackend
    ctl, th, *ex = _mms_stokes_control(8)
    ksp = ctl.incompressible_linear_solve(
        solver_parameters=MMS_STOKES_SP, lambda_v_bounds=(0.3924, 2.0598),
        lambda_p_bounds=(0.5, 2.0), backend=GpuBackend(schur=(40, 0.01, 2.25)))
    assert ksp.getConvergedReason() > 0
    ref, *_ = _mms_stokes_control(8)
    ref.incompressible_linear_solve(
        solver_parameters=MMS_STOKES_SP, lambda_v_bounds=(0.3924, 2.0598),
        lambda_p_bounds=(0.5, 2.0), backend=common.OracleBackend(schur=(40, 0.01, 2.25)))
    assert np.abs(ctl._v - ref._v).max() < 1e-7 and np.abs(ctl._zeta - ref._zeta).max() < 1e-8


def _stationary_navier_stokes_control(N=4, nu=1.0):
    """Shaped like ``test/test_control.py:1027-1092``: stationary Navier-Stokes control on
    P2-P1, ``forward_form = nu grad-grad + (w . grad)``, homogeneous velocity conditions."""
    from control_amd.control import Stationary
    from control_amd.fem import rectangle_p2p1
    th = rectangle_p2p1(N, N, 2.0, 2.0)

    def v_d(X):
        return np.concatenate([np.sin(0.5 * np.pi * X[:, 0]) ** 2 * np.sin(np.pi * X[:, 1]),
                               -np.sin(np.pi * X[:, 0]) * np.sin(0.5 * np.pi * X[:, 1]) ** 2])

    def D_v(w):
        return nu * th.K_v + th.convection_v(w)

    def D_p(w):
        return nu * th.K_p + th.convection_p(w)
    return Stationary(th, D_v, desired_state=v_d, beta=1.0e-2), th, D_p


def test_stationary_navier_stokes_picard_with_the_oracle():
    ctl, th, D_p = _stationary_navier_stokes_control()
    norms = ctl.incompressible_non_linear_solve(
        forward_operator_p=D_p, solver_parameters=MMS_STOKES_SP,
        lambda_v_bounds=(0.3924, 2.0598), lambda_p_bounds=(0.5, 2.0), max_non_linear_iter=20,
        relative_non_linear_tol=1.0e-8, absolute_non_linear_tol=0.0,
        backend=common.OracleBackend(schur=(40, 0.01, 2.25)))
    assert norms[-1] <= 1.0e-8 * norms[0]
    assert all(b < a for a, b in zip(norms[1:], norms[2:]))
    assert np.abs(th.B @ ctl._v).max() < 1e-8          # discretely divergence-free state


@pytest.mark.gpu
def test_stationary_navier_stokes_picard_on_the_gpu():
    from control_amd.control import GpuBackend
    out = []
    for be in (GpuBackend(schur=(40, 0.01, 2.25)), common.OracleBackend(schur=(40, 0.01, 2.25))):
        ctl, th, D_p = _stationary_navier_stokes_control()
        norms = ctl.incompressible_non_linear_solve(
            forward_operator_p=D_p, solver_parameters=MMS_STOKES_SP,
            lambda_v_bounds=(0.3924, 2.0598), lambda_p_bounds=(0.5, 2.0),
            max_non_linear_iter=20, relative_non_linear_tol=1.0e-8, absolute_non_linear_tol=0.0,
            backend=be)
        out.append((norms, ctl._v.copy(), ctl._zeta.copy()))
    assert len(out[0][0]) == len(out[1][0])
    assert np.abs(out[0][1] - out[1][1]).max() < 1e-7
    assert np.abs(out[0][2] - out[1][2]).max() < 1e-8


# ------------------------------------------------------ automatic Chebyshev parameters

def test_suggested_chebyshev_parameters_make_the_solve_converge():
    """``suggest_chebyshev``: interval from the spectrum of the Jacobi-scaled interior-level
    matrix, degree 1.5 sqrt(kappa).  With it the oracle converges on a 64^2 x 16 heat-control
    system on which 8 sweeps on [0.07, 2.1] do not (the situation of cfg 2)."""
    from control_amd.control import suggest_chebyshev
    p = common.heat_problem(n=64, n_t=16, beta=1.0e-4, T=2.0)
    sd, tau, m = p["sd"], p["tau"], p["m"]
    its, emin, emax = suggest_chebyshev(p["blocks"][2][(1, 1)], sd.M, tau / np.sqrt(p["beta"]),
                                        p["nodes"])
    assert 25 <= its <= 60 and 1e-3 < emin < 1e-2 and 2.0 < emax < 2.3
    import bench
    b_0, b_1 = bench.readme_rhs(p)
    sp_ = {"linear_solver": "gmres", "gmres_restart": 10, "relative_tolerance": 1.0e-6,
           "absolute_tolerance": 0.0, "maximum_iterations": 60, "monitor_convergence": False}
    osys = common.oracle_system(p)
    v, z = np.zeros((m, sd.n_dofs)), np.zeros((m, sd.n_dofs))
    r = osys.solve(v, z, b_0, b_1, solver_parameters=sp_,
                   pc_fn=common.oracle_pc(p, (20, 0.5, 2.0), (its, emin, emax)))
    assert r.reason > 0 and r.its <= 40
    with pytest.raises(RuntimeError):
        osys.solve(np.zeros_like(v), np.zeros_like(z), b_0, b_1, solver_parameters=sp_,
                   pc_fn=common.oracle_pc(p, (20, 0.5, 2.0), (8, 0.07, 2.1)))


def test_suggested_chebyshev_parameters_on_a_3d_block_need_no_factorisation():
    """Both ends of the spectrum come from plain Lanczos: a 3-D block (here 24^3; 64^3 takes
    2.7 s) costs seconds, where a shift-invert estimate -- a sparse LU with 3-D fill-in -- did
    not finish in minutes.  The interval must contain the dense spectrum of a small block."""
    import time
    from control_amd.blocks import instationary_blocks
    from control_amd.control import suggest_chebyshev
    from control_amd.fem import unit_cube_p1
    sd = unit_cube_p1(24)
    tau, beta = 2.0 / 31.0, 1.0e-4
    t0 = time.perf_counter()
    its, emin, emax = suggest_chebyshev(tau * sd.K + sd.M, sd.M, tau / np.sqrt(beta), sd.boundary)
    assert time.perf_counter() - t0 < 30.0
    assert 4 <= its <= 40 and 0.0 < emin < 0.2 and 1.5 < emax < 2.6
    small = unit_cube_p1(6)
    L = (tau * small.K + small.M + (tau / np.sqrt(beta)) * small.M).toarray()
    keep = np.setdiff1d(np.arange(small.n_dofs), small.boundary)
    L = L[np.ix_(keep, keep)]
    d = 1.0 / np.sqrt(np.diag(L))
    ev = np.linalg.eigvalsh(d[:, None] * L * d[None, :])
    _, lo, hi = suggest_chebyshev(tau * small.K + small.M, small.M, tau / np.sqrt(beta),
                                  small.boundary)
    assert lo <= ev[0] and ev[-1] <= hi and lo > 0.9 * ev[0] and hi < 1.1 * ev[-1]


@pytest.mark.gpu
def test_gpu_backend_default_parameters_converge():
    """``GpuBackend()`` without arguments picks the sweeps itself."""
    from control_amd.control import GpuBackend
    ctl, disc, ref_v, ref_zeta = common.mms_heat_control(32, False)
    be = GpuBackend()
    ksp = ctl.linear_solve(solver_parameters=common.MMS_SOLVER_PARAMETERS,
                           lambda_v_bounds=(0.5, 2.0), backend=be)
    assert ksp.getConvergedReason() > 0
    ev, ez = common.mms_errors(ctl, disc, ref_v, ref_zeta)
    assert ev < 1e-2 and ez < 1e-2
