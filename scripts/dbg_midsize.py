"""Diagnostic: set-up and one preconditioner application on a mid-size mesh (more than 1 024
slices: the data-flow sweep program runs with 4-wave workgroups)."""
import faulthandler, os, sys, time
faulthandler.dump_traceback_later(90, exit=True)
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import numpy as np
import common
import bench

class A: pass
a = A(); a.workload = "heat2d"; a.n = int(sys.argv[1]); a.n_t = int(sys.argv[2]); a.beta = 1e-4
a.T = 2.0; a.scheme = "BE"; a.mode = "G"; a.schur_its = int(sys.argv[3]) if len(sys.argv) > 3 else 8; a.schur_emin = float(sys.argv[4]) if len(sys.argv) > 4 else 0.07; a.schur_emax = 2.1
t = time.time(); p = bench.build_problem(a); print(f"problem {time.time()-t:.1f} s", flush=True)
t = time.time(); g = common.gpu_system(p, share_values=False); print(f"system {time.time()-t:.1f} s", flush=True)
t = time.time(); pc = common.gpu_pc(p, p["mass"], p["schur"]); g._set_pc(pc); g._ck(g._lib.kkt_sync(g.handle))
print(f"pc build {time.time()-t:.1f} s", flush=True)
x = common.rng_vector(g.info()["n_local"])
t = time.time(); y = g.pc_apply(x, pc); print(f"first apply {time.time()-t:.2f} s", flush=True)
t = time.time(); y2 = g.pc_apply(x, pc); print(f"second apply {time.time()-t:.3f} s, same {np.array_equal(y, y2)}", flush=True)
print("program fall-backs:", g.info()["program_fallbacks"], "| last error:", g._lib.kkt_last_error(g.handle).decode(), flush=True)
os.environ["KKT_PERSISTENT"] = "0"
g2 = common.gpu_system(p, share_values=False); pc2 = common.gpu_pc(p, p["mass"], p["schur"])
y3 = g2.pc_apply(x, pc2)
print("vs plain launches: max diff", np.abs(y - y3).max(), flush=True)
