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
