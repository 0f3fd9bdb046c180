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
