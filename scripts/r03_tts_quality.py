"""What the library-default stopping test (gmres, left, rtol 1e-6) leaves behind with the plain
Chebyshev preconditioner and with the two-grid one: true residual and distance to a tightly
converged solution (cfg 2, README right-hand side)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import numpy as np
import common, bench
from control_amd.coarse import multilinear_coarse_space
n, n_t = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 64)
p = common.heat_problem(n=n, n_t=n_t, beta=1e-4)
m, nx = p["m"], p["sd"].n_dofs
g = common.gpu_system(p, share_values=False)
g0, g1 = bench.readme_rhs(p)
rhs = np.concatenate([g0.ravel(), g1.ravel()])
co = (multilinear_coarse_space(p["sd"].coords, p["nodes"], cells=n // 8), 1)
def mk(its, emin, cyc):
    return common.gpu_pc(p, (20, 0.5, 2.0), (its, emin, 2.1), coarse=(co[0], cyc))
pcs = {"plain 80": common.gpu_pc(p, (20, 0.5, 2.0), (80, 7e-4, 2.1)),
       "two-grid 1x10/.07": mk(10, 0.07, 1), "two-grid 1x16/.03": mk(16, 0.03, 1),
       "two-grid 1x20/.02": mk(20, 0.02, 1), "two-grid 2x6/.1": mk(6, 0.1, 2),
       "two-grid 2x8/.07": mk(8, 0.07, 2), "two-grid 2x10/.07": mk(10, 0.07, 2)}
def solve(pc, ksp, rtol, maxit=400, restart=10):
    sp = {"linear_solver": ksp, "gmres_restart": restart, "maximum_iterations": maxit, "relative_tolerance": rtol,
          "absolute_tolerance": 0.0, "monitor_convergence": False, "preconditioner": True}
    u0, u1 = np.zeros((m, nx)), np.zeros((m, nx))
    r = g.solve(u0, u1, g0, g1, solver_parameters=sp, pc_fn=pc)
    x = np.concatenate([u0.ravel(), u1.ravel()])
    return r, x
rt, xt = solve(pcs["two-grid 2x10/.07"], "fgmres", 1e-12, 600, 30)
print("tight: its", rt.its, "reason", rt.reason, "true residual", np.linalg.norm(rhs - g.mult(xt)) / np.linalg.norm(rhs))
for name, pc in pcs.items():
    for ksp, rtol, restart in (("gmres", 1e-6, 10), ("fgmres", 1e-6, 10), ("fgmres", 1e-6, 30)):
        r, x = solve(pc, ksp, rtol, 400, restart)
        ms = g.info()["last_solve_ms"]
        print(f"{name:18s} {ksp:6s}({restart}) {ms/1e3:6.3f} s: its {r.its:3d} reason {r.reason} true residual "
              f"{np.linalg.norm(rhs - g.mult(x)) / np.linalg.norm(rhs):.2e}  error vs tight "
              f"{np.linalg.norm(x - xt) / np.linalg.norm(xt):.2e}  v-part {np.linalg.norm((x - xt)[:m*nx]) / np.linalg.norm(xt[:m*nx]):.2e}")
