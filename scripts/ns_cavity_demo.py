"""Navier-Stokes control in the lid-driven cavity (data of test/test_control.py:4171-4268) on
the GPU: Picard history for a given mesh and viscosity."""
import argparse, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import numpy as np
import common
from control_amd import picard

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=32)
ap.add_argument("--n_t", type=int, default=10)
ap.add_argument("--nu", type=float, default=0.01)
ap.add_argument("--its", type=int, default=40)
ap.add_argument("--emin", type=float, default=0.005)
ap.add_argument("--cn", action="store_true")
a = ap.parse_args()
pb, v0, lid = common.navier_stokes_cavity_problem(n=a.n, n_t=a.n_t, CN=a.cn)
pb.nu = a.nu
s = dict(common.STOKES_SPECS)
s["schur"] = (a.its, a.emin, 2.25)
s["kp"] = (a.its, a.emin, 2.1)
sp = dict(common.NS_SOLVER_PARAMETERS)
sp["maximum_iterations"] = 200
gls = picard.GpuLinearSolver(pb, mass=s["mass"], schur=s["schur"], kp=s["kp"], mp=s["mp"],
                             solver_parameters=sp)
t = time.time()
try:
    out = picard.incompressible_non_linear_solve(pb, gls, v=v0, max_non_linear_iter=10)
    th = pb.disc
    print("converged", out["converged"], "linear iterations", out["linear_iterations"],
          f"{time.time() - t:.1f} s")
    print("bc kept", np.array_equal(out["v"][:, th.boundary_v], v0[:, th.boundary_v]),
          "max |B v|", max(np.abs(th.B @ out["v"][i]).max() for i in range(pb.n_t)))
except RuntimeError as e:
    print("FAILED:", e, f"{time.time() - t:.1f} s")
