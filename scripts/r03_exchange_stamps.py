"""Tooling: sub-phases of the coarse exchange inside the tile program (library built with
-DKKT_XSTAMPS, e.g. make -C control_amd/csrc OBJDIR=../../build/xstamps OUT=../../build/xstamps/libkkt.so
EXTRA=-DKKT_XSTAMPS; run with KKT_LIB=$PWD/build/xstamps/libkkt.so)."""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
from control_amd import problems as common
import bench

class A: pass
a = A(); a.workload = "heat2d"; a.n = 256; a.n_t = 64; a.beta = 1e-4; a.T = 2.0; a.scheme = "BE"; a.mode = "G"
a.schur_its = 8; a.schur_emin = 0.07; a.schur_emax = 2.1; a.coarse_cycles = 2; a.coarse_cell = 8
p = bench.build_problem(a)
g = common.gpu_system(p, share_values=False, options={"stamps": "1", "no_graph": "1"})
pc = common.gpu_pc(p, p["mass"], p["schur"], coarse=p.get("coarse"))
lib, h = g._lib, g.handle
x = common.rng_vector(g.info()["n_local"])
g.pc_apply(x, pc)
n = 8 * 256
buf = (C.c_ulonglong * n)()
lib.kkt_debug_prog_stats.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
lib.kkt_debug_prog_stats(h, buf, n)
reps = 2
for _ in range(reps):
    g.pc_apply(x, pc)
lib.kkt_debug_prog_stats(h, buf, n)
d = np.array(buf[:], dtype=np.float64).reshape(-1, 8)
d = d[d[:, 7] > 0]
nx = 256 * reps          # exchanges per tile
names = {0: "restriction + publish", 1: "own polls (thread 0)", 2: "barrier after the polls",
         4: "coarse residual (sums of the partials)", 5: "owned products + publish",
         6: "products polled + barrier", 7: "whole exchange (incl. prolongation)"}
for c, nm in names.items():
    v = d[:, c] * 0.01 / nx
    print(f"{nm:40s} us each: mean {v.mean():6.2f}  min {v.min():6.2f}  max {v.max():6.2f}")
