"""Navier-Stokes control at the reference's own viscosities on the GPU (tooling): lid-driven cavity
nu = 1/100 (test/test_control.py:4171-4368) and the manufactured problem nu = 1/50 (:4371-4925),
sub-solves as Chebyshev sweeps on the ellipse of each matrix (estimated on the device, or given)."""
import argparse, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import numpy as np
import common
from control_amd import picard

ap = argparse.ArgumentParser()
ap.add_argument("--case", default="cavity", choices=["cavity", "mms"])
ap.add_argument("--n", type=int, default=8)
ap.add_argument("--n_t", type=int, default=10)
ap.add_argument("--nu", type=float, default=None)
ap.add_argument("--schur", default="auto")
ap.add_argument("--cn", action="store_true")
ap.add_argument("--max-it", type=int, default=100)
ap.add_argument("--rtol", type=float, default=1e-8)
ap.add_argument("--atol", type=float, default=0.0)
ap.add_argument("--nl-tol", type=float, default=1e-5)
a = ap.parse_args()
if a.case == "cavity":
    pb, v0, _ = common.navier_stokes_cavity_problem(n=a.n, n_t=a.n_t, CN=a.cn)
    pb.nu = a.nu if a.nu is not None else 0.01
    true_v = None
else:
    pb, v0, true_v = common.mms_navier_stokes_control(a.n, CN=a.cn, n_t=a.n_t,
                                                     nu=a.nu if a.nu is not None else 0.02)
schur = (-1, 0.0, 0.0) if a.schur == "auto" else eval(a.schur)
kp = (-1, 0.0, 0.0) if a.schur == "auto" else (schur[0], 0.02, 2.1)
s = common.STOKES_SPECS
sp = dict(common.NS_SOLVER_PARAMETERS, maximum_iterations=a.max_it, relative_tolerance=a.rtol,
          absolute_tolerance=a.atol)
gls = picard.GpuLinearSolver(pb, mass=(20, 0.3924, 2.0598), schur=schur, kp=kp, mp=(20, 0.5, 2.0),
                             solver_parameters=sp, options={"verbose": "1"})
t = time.time()
try:
    out = picard.incompressible_non_linear_solve(pb, gls, v=v0, max_non_linear_iter=10,
                                                 relative_non_linear_tol=a.nl_tol,
                                                 absolute_non_linear_tol=min(a.nl_tol, 1e-6),
                                                 print_error_non_linear=False)
    print(f"{a.case} n={a.n} nu={pb.nu} {'CN' if a.cn else 'BE'} schur={schur}: converged "
          f"{out['converged']} norms {['%.2e' % x for x in out['norms']]} linear its "
          f"{out['linear_iterations']} {time.time() - t:.1f} s", flush=True)
    if true_v is not None:
        th = pb.disc
        ev = sum(pb.tau * ((out["v"][i] - true_v(i * pb.tau)) @ (th.M_v @ (out["v"][i] - true_v(i * pb.tau))))
                 for i in range(pb.n_t)) ** 0.5
        print(f"   velocity error {ev:.3e}")
except RuntimeError as e:
    print(f"{a.case} n={a.n} nu={pb.nu} schur={schur}: FAILED {e} {time.time() - t:.1f} s", flush=True)
