"""Randomised parity of the tile sweep programs with coarse corrections (two-grid levels): random
mesh sizes (2-D P1, 3-D P1), level counts, sweeps per cycle, cycles, coarse cells, tile depths and
workgroup sizes, BE and CN; every case against the plain-launch form to 1e-12 (the restriction
sums associate differently), a second application bit for bit, no time-outs.
Environment: CASES (default 60), SEED."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from control_amd import problems as common
from control_amd.coarse import multilinear_coarse_space

rng = np.random.default_rng(int(os.environ.get("SEED", 1)))
bad = used = 0
worst = 0.0
for case in range(int(os.environ.get("CASES", 60))):
    three_d = rng.random() < 0.3
    n = int(rng.integers(6, 15)) if three_d else int(rng.integers(12, 90))
    n_t = int(rng.integers(2, 7))
    CN = bool(rng.random() < 0.4)
    its = int(rng.integers(1, 14))
    cycles = int(rng.integers(1, 4))
    cells = int(rng.integers(2, max(3, n // 3)))
    depth = int(rng.choice([0, 0, 1, 2, 3, 5, 8]))
    waves = int(rng.choice([0, 0, 2, 4, 8, 16]))
    p = common.heat_problem(space="p1_3d" if three_d else "p1", n=n, n_t=max(n_t, 3 if CN else 2), CN=CN)
    P = multilinear_coarse_space(p["sd"].coords, p["nodes"], cells=cells)
    mass, schur = (int(rng.integers(1, 8)), 0.5, 2.5), (its, 0.07, 2.2)
    opts = {"prog_mode": "tile", "persistent": "1"}
    if depth: opts["tile_depth"] = str(depth)
    if waves: opts["tile_waves"] = str(waves)
    x = common.rng_vector(2 * p["m"] * p["sd"].n_dofs, 100 + case)
    tag = (f"case {case}: {'3-D' if three_d else '2-D'} n={n} n_t={p['n_t']} CN={CN} sweeps={its} "
           f"cycles={cycles} coarse={P.shape[1]} depth={depth} waves={waves}")
    try:
        g = common.gpu_system(p, options=opts)
        pc = common.gpu_pc(p, mass, schur, coarse=(P, cycles))
        got = g.pc_apply(x, pc)
        again = g.pc_apply(x, pc)
        form = g.info()["sweep_form"]
        plain = common.gpu_system(p, options={"persistent": "0"}).pc_apply(
            x, common.gpu_pc(p, mass, schur, coarse=(P, cycles)))
        err = float(np.linalg.norm(got - plain) / max(np.linalg.norm(plain), 1e-300))
        worst = max(worst, err)
        ok = err < 1e-12 and np.array_equal(got, again) and g.info()["program_fallbacks"] == 0
        used += 1 if form == 3 else 0
        print(f"{tag}: {'ok' if ok else 'MISMATCH'} (form {form}, deviation {err:.1e}, fallbacks "
              f"{g.info()['program_fallbacks']})", flush=True)
        bad += 0 if ok else 1
    except Exception as e:      # noqa: BLE001
        print(f"{tag}: ERROR {type(e).__name__}: {e}", flush=True)
        bad += 1
print(f"done: {bad} bad, {used} cases ran the tile program with coarse corrections, worst deviation "
      f"from the plain launches {worst:.1e}")
sys.exit(1 if bad else 0)
