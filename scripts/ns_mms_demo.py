"""Manufactured Navier-Stokes control (data of test/test_control.py:4371-4470) on the GPU:
Picard history and velocity error per mesh."""
import argparse, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import numpy as np
import common
from control_amd import picard

ap = argparse.ArgumentParser()
ap.add_argument("--sizes", default="8,16,32")
ap.add_argument("--cn", action="store_true")
ap.add_argument("--nu", type=float, default=1.0 / 50.0)
ap.add_argument("--its", type=int, default=40)
ap.add_argument("--emin", type=float, default=0.005)
a = ap.parse_args()
errs = []
for N in [int(k) for k in a.sizes.split(",")]:
    pb, v0, true_v = common.mms_navier_stokes_control(N, CN=a.cn, nu=a.nu)
    th = pb.disc
    s = dict(common.STOKES_SPECS)
    sp = dict(common.NS_SOLVER_PARAMETERS, maximum_iterations=200)
    gls = picard.GpuLinearSolver(pb, mass=s["mass"], schur=(a.its, a.emin, 2.25),
                                 kp=(a.its, a.emin, 2.1), mp=s["mp"], solver_parameters=sp)
    t = time.time()
    try:
        out = picard.incompressible_non_linear_solve(pb, gls, v=v0, max_non_linear_iter=10,
                                                     print_error_non_linear=False)
    except RuntimeError as e:
        print(N, "FAILED", e)
        continue
    tau = pb.tau
    ev = ez = 0.0
    for i in range(pb.n_t):
        d = out["v"][i] - true_v(i * tau)
        ev += tau * (d @ (th.M_v @ d))
        ez += tau * (out["zeta"][i] @ (th.M_v @ out["zeta"][i]))
    errs.append((np.sqrt(ev), np.sqrt(ez)))
    print(N, "converged", out["converged"], "norms", [f"{x:.2e}" for x in out["norms"]],
          "its", out["linear_iterations"], "errors", errs[-1], f"{time.time() - t:.1f} s", flush=True)
if len(errs) > 1:
    e = np.array(errs)
    print("orders", np.log(e[:-1] / e[1:]) / np.log(2.0))
