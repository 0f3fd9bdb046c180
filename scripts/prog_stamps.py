"""Diagnostic: where a phase of the persistent sweep program spends its cycles.
Needs a library built with `make -C control_amd/csrc clean all EXTRA=-DKKT_STAMPS`."""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import numpy as np
import common
from control_amd import _lib
import bench

class A: pass
a = A(); a.workload="heat2d"; a.n=256; a.n_t=64; a.beta=1e-4; a.T=2.0; a.scheme="BE"; a.mode="G"
a.schur_its=8; a.schur_emin=0.07; a.schur_emax=2.1
p = bench.build_problem(a)
g = common.gpu_system(p)
g._set_pc(common.gpu_pc(p, p["mass"], p["schur"]))
lib, h = g._lib, g.handle
x = common.rng_vector(g.info()["n_local"])
y = g.pc_apply(x, g._pc_state)          # warm-up (also captures graphs)
n = 16 * 200
buf = (C.c_ulonglong * n)()
lib.kkt_debug_prog_stats.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
lib.kkt_debug_prog_stats(h, buf, n)      # reset
reps = 3
for _ in range(reps):
    g.pc_apply(x, g._pc_state)
lib.kkt_debug_prog_stats(h, buf, n)
d = np.array(buf[:], dtype=np.float64).reshape(-1, 16)
d = d[d[:, 2] > 0]
ph = d[:, 2]
print("workgroups", len(d), "phases per wg", ph[0] / reps)
# s_memtime ticks at 100 MHz? (constant clock) -> report raw ticks per phase
for name, col in (("st0 desc", 8), ("st1 own-row ops", 9), ("st2 mat loads", 10), ("st3 gather", 11),
                  ("st4 barrier", 12), ("st5 fma+epi+st", 13), ("st6 drain", 14)):
    v = d[:, col] / ph
    print(f"{name:16s} per phase: mean {v.mean():9.1f}  min {v.min():9.1f}  max {v.max():9.1f}")
