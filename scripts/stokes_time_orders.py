"""Time-discretisation orders of the instationary Stokes-control driver on the GPU (data of
test/test_control.py:3546-3751 / 3965-4168): nested time grids, differences at common levels."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import numpy as np
import common
from control_amd.control import GpuBackend

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for CN in (False, True):
    sols = {}
    for n_t in (5, 9, 17):
        t = time.time()
        ctl, th, tv = common.stokes_exact_sol_control(CN, n=N, n_t=n_t, T_f=2.0, beta=1e-3,
                                                      taylor_hood=True)
        ksp = ctl.incompressible_linear_solve(
            lambda_v_bounds=(0.3924, 2.0598), lambda_p_bounds=(0.5, 2.0),
            solver_parameters=common.MMS_SOLVER_PARAMETERS,
            backend=GpuBackend(schur=(60, 0.002, 2.3)))
        tau = 2.0 / (n_t - 1)
        err = np.sqrt(tau * sum((ctl._v[i] - tv(th.coords_v, i * tau)) @ (th.M_v @ (ctl._v[i] - tv(th.coords_v, i * tau))) for i in range(n_t)))
        sols[n_t] = (ctl._v.copy(), ctl._zeta.copy())
        print(CN, n_t, ksp.reason, ksp.its, f"error vs exact {err:.3e}", f"{time.time() - t:.1f} s", flush=True)
    ds = []
    for a, b in ((5, 9), (9, 17)):
        tau = 2.0 / (a - 1)
        ds.append([np.sqrt(tau * sum(x @ (th.M_v @ x) for x in sols[a][k] - sols[b][k][::2]))
                   for k in (0, 1)])
    d = np.array(ds)
    print("CN" if CN else "BE", "differences", d.tolist(), "orders", (np.log(d[0] / d[1]) / np.log(2)).tolist())
