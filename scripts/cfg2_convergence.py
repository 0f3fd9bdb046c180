"""Time-to-solution on cfg 2 (256^2 P1, n_t = 64, beta = 1e-4, BE) for several choices of the
Chebyshev substitute of the reference's AMG sub-solves: iterations and seconds to
rtol = 1e-6 (library default, control.py:3261-3266) with the README right-hand side."""
import argparse, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import numpy as np
import common

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=256)
ap.add_argument("--n_t", type=int, default=64)
ap.add_argument("--beta", type=float, default=1e-4)
ap.add_argument("--max-it", type=int, default=300)
ap.add_argument("--scheme", default="BE", choices=["BE", "CN"])
a = ap.parse_args()
p = common.heat_problem(n=a.n, n_t=a.n_t, beta=a.beta, T=2.0, CN=a.scheme == "CN", share=False)
sd, m, tau = p["sd"], p["m"], p["tau"]
g = common.gpu_system(p)
sys.path.insert(0, R)
import bench
b_0, b_1 = bench.readme_rhs(p)
ap2 = os.environ.get("KKT_SWEEP", "coarse")
SETS = {"coarse": ((8, 0.07, 2.1), (16, 0.02, 2.1), (30, 0.02, 2.1), (30, 0.005, 2.1),
                   (60, 0.002, 2.1), (100, 0.0007, 2.1)),
        "fine": ((60, 0.001, 2.1), (80, 0.001, 2.1), (80, 0.0007, 2.1), (100, 0.001, 2.1),
                 (100, 0.0007, 2.1), (140, 0.0005, 2.1), (140, 0.0007, 2.1)),
        # lower end at / below the smallest eigenvalue suggest_chebyshev reports (5.5e-4)
        "interval": ((90, 0.00055, 2.1), (100, 0.00055, 2.1), (124, 0.00055, 2.1),
                     (110, 0.0004, 2.1), (124, 0.0004, 2.1), (160, 0.0004, 2.1),
                     (124, 0.0003, 2.1), (180, 0.0003, 2.1))}
for ksp, restart in ((("gmres", 10), ("fgmres", 30)) if ap2 == "coarse" else
                     (("gmres", 10),) if ap2 == "interval" else
                     (("gmres", 10), ("gmres", 30))):
    for schur in SETS[ap2]:
        pc = common.gpu_pc(p, (20, 0.5, 2.0), schur)
        v, z = np.zeros((m, sd.n_dofs)), np.zeros((m, sd.n_dofs))
        sp = {"linear_solver": ksp, "gmres_restart": restart, "relative_tolerance": 1e-6,
              "absolute_tolerance": 0.0, "maximum_iterations": a.max_it,
              "monitor_convergence": False}
        t = time.time()
        try:
            r = g.solve(v, z, b_0, b_1, pc_fn=pc, solver_parameters=sp)
            its, ok = r.getIterationNumber(), "converged"
        except RuntimeError:
            its, ok = a.max_it, "NOT converged"
        ms = g.info()["last_solve_ms"]
        print(f"{ksp}({restart}) schur Chebyshev {schur}: {ok}, {its} iterations, "
              f"solve {ms / 1e3:.3f} s (wall {time.time() - t:.2f} s)", flush=True)
