"""Diagnostic: where a workgroup of the tile sweep program spends its time (option "stamps":
100 MHz ticks summed per tile in hand-offs, local steps and level prologues)."""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
from control_amd import problems as common
import bench

class A: pass
a = A(); a.workload = os.environ.get("WORKLOAD", "heat2d"); a.n = int(os.environ.get("N", 256))
a.n_t = int(os.environ.get("N_T", 64)); a.beta = 1e-4
a.T = 2.0; a.scheme = "BE"; a.mode = "G"
a.schur_its = int(os.environ.get("ITS", 80)); a.schur_emin = float(os.environ.get("EMIN", 7e-4)); a.schur_emax = 2.1
a.coarse_cycles = int(os.environ.get("COARSE_CYCLES", 0)); a.coarse_cell = int(os.environ.get("COARSE_CELL", 8))
p = bench.build_problem(a)
opts = {"stamps": "1", "no_graph": "1"}
for k in ("tile_depth", "tile_waves", "tile_poll_delay"):
    if os.environ.get(k.upper()):
        opts[k] = os.environ[k.upper()]
g = common.gpu_system(p, share_values=False, options=opts)
pc = common.gpu_pc(p, p["mass"], p["schur"], coarse=p.get("coarse"))
lib, h = g._lib, g.handle
x = common.rng_vector(g.info()["n_local"])
g.pc_apply(x, pc)
n = 8 * 256
buf = (C.c_ulonglong * n)()
lib.kkt_debug_prog_stats.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
lib.kkt_debug_prog_stats(h, buf, n)      # reset
reps = 2
for _ in range(reps):
    g.pc_apply(x, pc)
lib.kkt_debug_prog_stats(h, buf, n)
d = np.array(buf[:], dtype=np.float64).reshape(-1, 8)
d = d[d[:, 4] > 0]
us = 0.01   # 100 MHz ticks
print(f"tiles {len(d)}  hand-offs/tile {d[0,3]/reps:.0f}  steps/tile {d[0,4]/reps:.0f}  levels {d[0,5]/reps:.0f}")
for name, col, cnt in (("hand-off", 0, 3), ("local step", 1, 4), ("level prologue", 2, 5)):
    v = d[:, col] * us / d[:, cnt]
    tot = d[:, col] * us / reps / 1e3
    print(f"{name:15s} us each: mean {v.mean():7.3f} min {v.min():7.3f} max {v.max():7.3f}   total ms/application: mean {tot.mean():6.2f} max {tot.max():6.2f}")
if a.coarse_cycles > 0:
    nco = d[:, 5] * a.coarse_cycles
    print(f"coarse exchange (restriction .. prolongation): us each mean {(d[:,7]*us/nco).mean():.3f} max {(d[:,7]*us/nco).max():.3f}   total ms/application: mean {(d[:,7]*us/reps/1e3).mean():.2f}")
else:
    print(f"level prologue up to the update (operands in registers): us each mean {(d[:,7]*us/d[:,5]).mean():.3f} max {(d[:,7]*us/d[:,5]).max():.3f}")
print(f"poll rounds per hand-off: mean {(d[:,6]/d[:,3]).mean():.1f} max {(d[:,6]/d[:,3]).max():.1f}")
