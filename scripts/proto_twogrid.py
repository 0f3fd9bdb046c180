"""CPU prototype (tooling, not product): how many dependent SpMV steps does a sub-solve of the
block-Schur preconditioner need when every `d` Jacobi-Chebyshev steps are followed by a
correction on the space of tile-wise constants (subdomain deflation: one coarse unknown per
sweep tile, Galerkin coarse matrix Z^T A Z inverted densely)?  Compares GMRES(10) iteration
counts on the README right-hand side against the plain degree-`its` Chebyshev sub-solves.

    python scripts/proto_twogrid.py --n 128 --n_t 32
"""
import argparse
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, R + "/tests")
import numpy as np
import scipy.sparse as sp

import common
import bench
from oracle import kkt_oracle as ko

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=128)
ap.add_argument("--n_t", type=int, default=32)
ap.add_argument("--beta", type=float, default=1e-4)
ap.add_argument("--tiles", type=int, default=16, help="tiles per axis")
ap.add_argument("--dim", type=int, default=2)
ap.add_argument("--max-it", type=int, default=120)
ap.add_argument("--variants", default="")
ap.add_argument("--T", type=float, default=2.0)
a = ap.parse_args()

p = common.heat_problem(space="p1" if a.dim == 2 else "p1_3d", n=a.n, n_t=a.n_t, beta=a.beta,
                        T=a.T, CN=False, share=True)
sd, m, tau = p["sd"], p["m"], p["tau"]
nodes = p["nodes"]
nx = sd.n_dofs
X = sd.coords
interior = np.ones(nx, bool)
interior[nodes] = False
# tiles: boxes on the interior nodes
tid = np.zeros(nx, np.int64)
for k in range(a.dim):
    q = np.minimum((X[:, k] * a.tiles).astype(np.int64), a.tiles - 1)
    tid = tid * a.tiles + q
ntile = a.tiles ** a.dim
rows = np.flatnonzero(interior)
Z = sp.csr_matrix((np.ones(rows.size), (rows, tid[rows])), shape=(nx, ntile))
if os.environ.get("PROTO_COARSE", "bilinear") == "bilinear":
    # multilinear interpolation from the grid of tile corners (boundary corners dropped)
    T = a.tiles
    rr, cc, vv = [], [], []
    import itertools
    for r in rows:
        xs_ = X[r] * T
        i0 = np.minimum(xs_.astype(int), T - 1)
        f = xs_ - i0
        for corner in itertools.product((0, 1), repeat=a.dim):
            ci = i0 + np.array(corner)
            wgt = np.prod([f[k] if corner[k] else 1 - f[k] for k in range(a.dim)])
            if np.all(ci > 0) and np.all(ci < T) and wgt > 0:
                idx = 0
                for k in range(a.dim):
                    idx = idx * (T - 1) + (ci[k] - 1)
                rr.append(r); cc.append(idx); vv.append(wgt)
    Z = sp.csr_matrix((vv, (rr, cc)), shape=(nx, (T - 1) ** a.dim))
print(f"nx {nx}, tiles {ntile}, rows per tile ~{rows.size / ntile:.0f}, tau {tau:.4f}")


class TwoGrid:
    """x = sum over cycles of [coarse correction, d Chebyshev-Jacobi smoothing steps]."""

    def __init__(self, At, cycles, d, lo_frac, emax, post_coarse=False):
        self.A = At
        self.dinv = 1.0 / At.diagonal()
        self.cycles, self.d = cycles, d
        self.emax = emax
        self.emin = emax / lo_frac
        E = (Z.T @ At @ Z).toarray()
        self.Einv = np.linalg.inv(E)
        self.post_coarse = post_coarse

    def coarse(self, r):
        return Z @ (self.Einv @ (Z.T @ r))

    def __call__(self, b):
        x = np.zeros_like(b)
        r = b.copy()
        for c in range(self.cycles):
            x += self.coarse(r)
            r = b - self.A @ x
            x += ko.chebyshev_jacobi(self.A, self.dinv, r, self.emin, self.emax, self.d)
            r = b - self.A @ x
        if self.post_coarse:
            x += self.coarse(r)
        return x

    def steps(self):
        return self.cycles * (self.d + 1)   # SpMVs: d - 1 in Chebyshev + 2 residuals


def make_pc(subsolver_factory):
    """pc_instationary_BE of the oracle with the Schur sub-solves replaced."""
    M = sd.M
    b00, b01, b10, b11 = p["blocks"]
    n_t, beta, eps = p["n_t"], p["beta"], 1e-3
    Mt = ko.assemble_with_bcs(M, nodes)
    shift = tau / beta ** 0.5
    cache = {}
    mass_spec = ko.ChebSpec(20, 0.5, 2.0 if a.dim == 2 else 2.5)

    def solve(blk, c, rhs):
        key = (id(blk), c)
        if key not in cache:
            At = ko.assemble_with_bcs(blk if c == 0.0 else blk + c * M, nodes)
            cache[key] = subsolver_factory(At)
        return cache[key](rhs)

    def pc_linear(u_0, u_1, b_0, b_1):
        for i in range(n_t):
            u_0[i] = ko._inner_solve(Mt, mass_spec, b_0[i].copy()) / tau
        u_0[n_t - 1] *= 1.0 / eps
        b = np.zeros_like(u_0)
        b[0] = b10[(0, 0)] @ u_0[0] - b_1[0]
        ko._bc(b[0], nodes)
        for i in range(1, n_t):
            b[i] = b10[(i, i)] @ u_0[i] + b10[(i, i - 1)] @ u_0[i - 1] - b_1[i]
            ko._bc(b[i], nodes)
        u_1[0] = solve(b10[(0, 0)], 0.0, b[0])
        for i in range(1, n_t):
            b[i] -= b10[(i, i - 1)] @ u_1[i - 1]
            ko._bc(b[i], nodes)
            u_1[i] = solve(b10[(i, i)], shift if i < n_t - 1 else eps ** 0.5 * shift, b[i])
        b = np.zeros_like(u_0)
        for i in range(n_t):
            b[i] = (M @ u_1[i]) * (tau if i < n_t - 1 else eps * tau)
            ko._bc(b[i], nodes)
        u_1[n_t - 1] = solve(b01[(n_t - 1, n_t - 1)], eps ** 0.5 * shift, b[n_t - 1])
        for i in range(n_t - 2, -1, -1):
            b[i] -= b01[(i, i + 1)] @ u_1[i + 1]
            ko._bc(b[i], nodes)
            u_1[i] = solve(b01[(i, i)], shift if i > 0 else 0.0, b[i])
    return pc_linear


osys = common.oracle_system(p)
g0, g1 = bench.readme_rhs(p)
emax = 2.1 if a.dim == 2 else 2.1


def run(name, factory, steps):
    pc = make_pc(factory)
    u0, u1 = np.zeros((m, nx)), np.zeros((m, nx))
    spar = {"linear_solver": "gmres", "gmres_restart": 10, "relative_tolerance": 1e-6,
            "absolute_tolerance": 0.0, "maximum_iterations": a.max_it,
            "monitor_convergence": False, "preconditioner": True}
    t = time.time()
    r = osys.solve(u0, u1, g0, g1, solver_parameters=spar, pc_fn=pc)
    print(f"{name:46s} steps/sub-solve {steps:4d}  GMRES(10) its {r.its:4d} reason {r.reason} "
          f"({time.time() - t:.0f} s)", flush=True)


# spectrum of a typical interior matrix
b10 = p["blocks"][2]
At = ko.assemble_with_bcs(b10[(1, 1)] + (tau / a.beta ** 0.5) * sd.M, nodes)
dinv = 1.0 / At.diagonal()
import scipy.sparse.linalg as spl
Dh = sp.diags(np.sqrt(dinv))
Sym = (Dh @ At @ Dh).tocsr()
lmax = spl.eigsh(Sym, k=1, which="LA", return_eigenvectors=False)[0]
lmin = spl.eigsh(Sym, k=1, sigma=0, which="LM", return_eigenvectors=False)[0]
print(f"Jacobi-scaled interior matrix: [{lmin:.3e}, {lmax:.3f}], kappa {lmax / lmin:.0f}, "
      f"1.6 sqrt(kappa) = {1.6 * (lmax / lmin) ** 0.5:.0f}")
deg = int(np.ceil(1.6 * (lmax / lmin) ** 0.5))

variants = a.variants.split(",") if a.variants else []
if not variants or "cheb" in variants:
    run(f"Chebyshev({deg}) on [{0.85 * lmin:.2e}, 2.1]",
        lambda A_: (lambda rhs, A_=A_, di=1.0 / A_.diagonal(): ko.chebyshev_jacobi(
            A_, di, rhs, 0.85 * lmin, emax, deg)), deg)
for spec in (variants or ["1x8:30", "2x8:30", "3x8:30", "2x12:60", "3x6:20", "4x6:20", "2x16:100",
                          "1x16:100", "1x24:200"]):
    if spec == "cheb":
        continue
    cyc, rest = spec.split("x")
    d, frac = rest.split(":")
    cyc, d, frac = int(cyc), int(d), float(frac)
    run(f"two-grid {cyc} x (coarse + Cheb({d}) on emax/{frac:g})",
        lambda A_: TwoGrid(A_, cyc, d, frac, emax), cyc * (d + 1))
