import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import numpy as np
import common, bench
from control_amd.coarse import multilinear_coarse_space
n, n_t = int(sys.argv[1]), int(sys.argv[2])
its, emin, cycles = int(sys.argv[3]), float(sys.argv[4]), int(sys.argv[5])
p = common.heat_problem(n=n, n_t=n_t, beta=1e-4, share=False)
P = multilinear_coarse_space(p["sd"].coords, p["nodes"])
g = common.gpu_system(p, options={"verbose": "1"})
pc = common.gpu_pc(p, (20, 0.5, 2.0), (its, emin, 2.1), coarse=(P, cycles))
x = common.rng_vector(2 * p["m"] * p["sd"].n_dofs)
y = g.pc_apply(x, pc)
print("info", {k: v for k, v in g.info().items() if "sweep" in k or "fallback" in k})
print("last error:", g._lib.kkt_last_error(g.handle).decode())
g2 = common.gpu_system(p, options={"persistent": "0"})
y2 = g2.pc_apply(x, common.gpu_pc(p, (20, 0.5, 2.0), (its, emin, 2.1), coarse=(P, cycles)))
print("tile vs plain", common.rel_err(y, y2))
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 0
bad = 0
for i in range(reps):
    yi = g.pc_apply(x, pc)
    if not np.array_equal(yi, y):
        bad += 1
        print("application", i, "differs", common.rel_err(yi, y), flush=True)
    f = g.info()["program_fallbacks"]
    if f:
        print("fallback at application", i, ":", g._lib.kkt_last_error(g.handle).decode(), flush=True)
        break
print("soak", reps, "applications, differing", bad, "fallbacks", g.info()["program_fallbacks"])
