"""Diagnostic: time of one preconditioner application (HIP events, kkt_time_pc_apply) over the
tile depth and the poll delay of the tile sweep program.  Environment as scripts/tile_stamps.py
(WORKLOAD, N, N_T, ITS, EMIN); DEPTHS / DELAYS / WAVES: space-separated lists."""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from control_amd import problems as common
import bench

class A: pass
a = A(); a.workload = os.environ.get("WORKLOAD", "heat2d"); a.n = int(os.environ.get("N", 256))
a.n_t = int(os.environ.get("N_T", 64)); a.beta = 1e-4
a.T = 2.0; a.scheme = "BE"; a.mode = "G"
a.schur_its = int(os.environ.get("ITS", 80)); a.schur_emin = float(os.environ.get("EMIN", 7e-4)); a.schur_emax = 2.1
p = bench.build_problem(a)
for waves in os.environ.get("WAVES", "0").split():
    for depth in os.environ.get("DEPTHS", "0 4 5 6 7 8 10").split():
        for delay in os.environ.get("DELAYS", "0 16 32").split():
            opts = {}
            if depth != "0": opts["tile_depth"] = depth
            if delay != "0": opts["tile_poll_delay"] = delay
            if waves != "0": opts["tile_waves"] = waves
            if os.environ.get("UNFUSED"): opts["tile_unfused"] = "1"
            g = common.gpu_system(p, share_values=False, options=opts)
            pc = common.gpu_pc(p, p["mass"], p["schur"])
            g._set_pc(pc)
            lib, h = g._lib, g.handle
            x = common.rng_vector(g.info()["n_local"])
            dx, dy = C.c_void_p(), C.c_void_p()
            g._ck(lib.kkt_vec_alloc(h, C.byref(dx))); g._ck(lib.kkt_vec_alloc(h, C.byref(dy)))
            from control_amd import _lib
            g._ck(lib.kkt_vec_upload(h, dx, _lib.f64(x)[1]))
            ms = C.c_float()
            g._ck(lib.kkt_time_pc_apply(h, dx, dy, 2, C.byref(ms)))
            g._ck(lib.kkt_time_pc_apply(h, dx, dy, 5, C.byref(ms)))
            print(f"waves {waves} depth {depth} delay {delay}: pc_apply {ms.value / 5:.3f} ms  fallbacks {g.info().get('program_fallbacks')}", flush=True)
            del pc, g
