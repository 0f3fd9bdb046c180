"""Randomised parity of the tile sweep programs: random mesh sizes (2-D P1, 3-D P1), level
counts, Chebyshev degrees, tile depths and workgroup sizes, BE and CN; every case bit for bit
against the plain launches, no time-outs.  Environment: CASES (default 60), SEED."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from control_amd import problems as common

rng = np.random.default_rng(int(os.environ.get("SEED", 1)))
bad = used = 0
for case in range(int(os.environ.get("CASES", 60))):
    three_d = rng.random() < 0.3
    n = int(rng.integers(4, 15)) if three_d else int(rng.integers(5, 90))
    n_t = int(rng.integers(2, 7))
    CN = bool(rng.random() < 0.4)
    its = int(rng.integers(2, 26))
    depth = int(rng.choice([0, 0, 1, 2, 3, 5, 8, 12]))
    waves = int(rng.choice([0, 0, 1, 2, 4, 8, 16]))
    p = common.heat_problem(space="p1_3d" if three_d else "p1", n=n, n_t=max(n_t, 3 if CN else 2), CN=CN)
    mass, schur = (int(rng.integers(1, 8)), 0.5, 2.5), (its, 0.05, 2.2)
    opts = {"prog_mode": "tile", "persistent": "1"}
    if depth: opts["tile_depth"] = str(depth)
    if waves: opts["tile_waves"] = str(waves)
    x = common.rng_vector(2 * p["m"] * p["sd"].n_dofs, 100 + case)
    tag = f"case {case}: {'3-D' if three_d else '2-D'} n={n} n_t={p['n_t']} CN={CN} its={its} depth={depth} waves={waves}"
    try:
        g = common.gpu_system(p, options=opts)
        got = g.pc_apply(x, common.gpu_pc(p, mass, schur))
        again = g.pc_apply(x, common.gpu_pc(p, mass, schur))
        plain = common.gpu_system(p, options={"persistent": "0"}).pc_apply(x, common.gpu_pc(p, mass, schur))
        ok = np.array_equal(got, plain) and np.array_equal(got, again) and g.info()["program_fallbacks"] == 0
        import ctypes as C
        from control_amd import _lib
        lib, h = g._lib, g.handle
        g._set_pc(common.gpu_pc(p, mass, schur))
        d_x, d_y = C.c_void_p(), C.c_void_p()
        g._ck(lib.kkt_vec_alloc(h, C.byref(d_x))); g._ck(lib.kkt_vec_alloc(h, C.byref(d_y)))
        g._ck(lib.kkt_vec_upload(h, d_x, _lib.f64(x)[1]))
        ms, nl, nph = C.c_float(), C.c_int(), C.c_int64()
        g._ck(lib.kkt_time_pc_sweeps(h, d_x, d_y, C.byref(ms), C.byref(nl), C.byref(nph)))
        used += 1 if nl.value > 0 else 0
        print(f"{tag}: {'ok' if ok else 'MISMATCH'} (persistent launches {nl.value}, fallbacks "
              f"{g.info()['program_fallbacks']})", flush=True)
        bad += 0 if ok else 1
    except Exception as e:      # noqa: BLE001
        print(f"{tag}: ERROR {type(e).__name__}: {e}", flush=True)
        bad += 1
print(f"done: {bad} bad, {used} cases ran persistent sweep launches")
sys.exit(1 if bad else 0)
