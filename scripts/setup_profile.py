"""Where the set-up time of cfg 2 goes (host side): cProfile around system construction and
preconditioner build."""
import cProfile, os, pstats, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import common
import bench

class A: pass
a = A(); a.workload = "heat2d"; a.n = 256; a.n_t = 64; a.beta = 1e-4; a.T = 2.0
a.scheme = "BE"; a.mode = "G"; a.schur_its = 80; a.schur_emin = 7e-4; a.schur_emax = 2.1
p = bench.build_problem(a)
g0 = common.gpu_system(common.heat_problem(n=8, n_t=4))     # context creation off the clock
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
g = common.gpu_system(p)
t1 = time.perf_counter()
g._set_pc(common.gpu_pc(p, p["mass"], p["schur"]))
g._ck(g._lib.kkt_sync(g.handle))
pr.disable()
t2 = time.perf_counter()
print(f"system {t1 - t0:.3f} s, preconditioner {t2 - t1:.3f} s")
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
