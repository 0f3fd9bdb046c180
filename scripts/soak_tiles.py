"""Soak test of the persistent tile sweep programs: the same preconditioner application many
times over, every result compared bit for bit with the first one (itself checked against the
plain launches), time-outs and fall-backs counted.  A race in the hand-off protocol (a granule
read before it was written, a buffer overwritten while a neighbour still reads it) would show as
a differing result or a time-out.  Environment: WORKLOAD (heat2d | heat3d), N, N_T, REPS."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
from control_amd import problems as common
import bench

class A: pass
a = A(); a.workload = os.environ.get("WORKLOAD", "heat2d"); a.n = int(os.environ.get("N", 256))
a.n_t = int(os.environ.get("N_T", 64)); a.beta = 1e-4; a.T = 2.0
a.scheme = os.environ.get("SCHEME", "BE"); a.mode = "G"
a.schur_its = int(os.environ.get("ITS", 80)); a.schur_emin = float(os.environ.get("EMIN", 7e-4)); a.schur_emax = 2.1
reps = int(os.environ.get("REPS", 500))
p = bench.build_problem(a)
g = common.gpu_system(p, share_values=False)
x = common.rng_vector(g.info()["n_local"])
pc = common.gpu_pc(p, p["mass"], p["schur"])
first = g.pc_apply(x, pc)
plain = common.gpu_system(p, share_values=False, options={"persistent": "0"}).pc_apply(
    x, common.gpu_pc(p, p["mass"], p["schur"]))
assert np.array_equal(first, plain), "tile programs differ from plain launches"
t0, bad = time.time(), 0
for k in range(reps):
    y = g.pc_apply(x, pc)
    if not np.array_equal(y, first):
        bad += 1
        print(f"application {k}: differs from the first in {np.count_nonzero(y != first)} entries", flush=True)
    if k % 100 == 99:
        print(f"{k + 1} applications, {bad} differing, fallbacks {g.info()['program_fallbacks']}, "
              f"{time.time() - t0:.0f} s", flush=True)
print(f"done: {reps} applications of {a.workload} n={a.n} n_t={a.n_t} {a.scheme}: {bad} differing, "
      f"fallbacks {g.info()['program_fallbacks']}")
sys.exit(1 if bad or g.info()["program_fallbacks"] else 0)
