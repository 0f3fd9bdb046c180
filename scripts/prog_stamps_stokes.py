"""Diagnostic: stages of a phase of the counter-form sweep program on the Stokes-control
velocity system (P2, wide rows).  Needs `make -C control_amd/csrc clean all EXTRA=-DKKT_STAMPS`."""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import numpy as np
import common

p = common.stokes_problem(n=128, n_t=32, beta=1e-3, T=2.0, CN=False, share=False)
specs = dict(mass=(20, 0.3924, 2.0598), mp=(20, 0.5, 2.0), schur=(40, 0.002, 2.25),
             kp=(40, 0.002, 2.1))
outer, gpc = common.stokes_gpu(p, specs)
outer._set_pc(gpc)
inner = gpc.inner
lib = outer._lib
x = common.rng_vector(outer.info()["n_local"])
outer.pc_apply(x, gpc)                   # warm-up
n = 16 * 500
buf = (C.c_ulonglong * n)()
lib.kkt_debug_prog_stats.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
assert lib.kkt_debug_prog_stats(inner.handle, buf, n) == 0     # reset
outer.pc_apply(x, gpc)
assert lib.kkt_debug_prog_stats(inner.handle, buf, n) == 0
d = np.array(buf[:], dtype=np.float64).reshape(-1, 16)
d = d[d[:, 2] > 0]
ph = d[:, 2]
print("workgroups", len(d), "phases per workgroup", ph[0])
for name, col in (("top", 8), ("wait+barrier", 9), ("body issue", 10), ("drain", 11),
                  ("barrier+flag", 12)):
    v = d[:, col] / ph          # s_memtime ticks (100 MHz constant clock: 10 ns each)
    print(f"{name:14s} ticks per phase: mean {v.mean():8.1f}  min {v.min():8.1f}  max {v.max():8.1f}")
