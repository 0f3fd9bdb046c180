"""Navier-Stokes control at BASELINE configs[4] size on ONE GPU (the config names 8):
P2-P1 128x128, n_t = 64, BE, Picard.  Prints where an outer iteration spends its time."""
import argparse, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import numpy as np
import common
from control_amd import picard

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=128)
ap.add_argument("--n_t", type=int, default=64)
ap.add_argument("--nu", type=float, default=0.05)
ap.add_argument("--beta", type=float, default=1.0e-2)
ap.add_argument("--max-it", type=int, default=6)
ap.add_argument("--schur-its", type=int, default=30)
ap.add_argument("--schur-emin", type=float, default=0.005)
ap.add_argument("--monitor", action="store_true")
a = ap.parse_args()
t0 = time.time()
pb = common.navier_stokes_problem(n=a.n, n_t=a.n_t, nu=a.nu, beta=a.beta)
print(f"[demo] discretisation {time.time() - t0:.1f} s: n_v={pb.disc.n_v} n_p={pb.disc.n_p} "
      f"unknowns={2 * a.n_t * (pb.disc.n_v + pb.disc.n_p)}", flush=True)
sp_ = dict(common.NS_SOLVER_PARAMETERS, relative_tolerance=1.0e-6, maximum_iterations=200,
           monitor_convergence=a.monitor)
gls = picard.GpuLinearSolver(pb, mass=(20, 0.3924, 2.0598), schur=(a.schur_its, a.schur_emin, 2.25),
                             kp=(a.schur_its, a.schur_emin, 2.1), mp=(20, 0.5, 2.0), solver_parameters=sp_)
inner = gls.linear_solve
times = []


def timed(D, Dp, b_0, b_1):
    t = time.time()
    first = gls.outer is None
    bl_t = time.time()
    out = inner(D, Dp, b_0, b_1)
    times.append((first, time.time() - t, gls.outer.info()["last_solve_ms"], out[2]))
    print(f"[demo] linear solve: {'build' if first else 'update'} + solve {times[-1][1]:.1f} s, "
          f"GPU solve {times[-1][2] / 1e3:.2f} s, {out[2]} FGMRES iterations", flush=True)
    return out


gls.linear_solve = timed
t = time.time()
out = picard.incompressible_non_linear_solve(pb, gls, max_non_linear_iter=a.max_it,
                                             relative_non_linear_tol=1.0e-5)
print(f"[demo] total {time.time() - t:.1f} s, converged={out['converged']}, "
      f"norms={['%.3e' % x for x in out['norms']]}", flush=True)
