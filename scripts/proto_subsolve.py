"""CPU prototype (tooling): error reduction of one sub-solve (tau K + c M)^-1 by (a) plain
Jacobi-Chebyshev and (b) cycles of [tile-constant coarse correction, d Chebyshev smoothing
steps], in the energy norm, for a random, a smooth and a medium-frequency solution."""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, R + "/tests")
import numpy as np
import scipy.sparse as sp

import common
from oracle import kkt_oracle as ko

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n_t = int(sys.argv[2]) if len(sys.argv) > 2 else 32
T = int(sys.argv[3]) if len(sys.argv) > 3 else 16
beta = 1e-4
p = common.heat_problem(n=n, n_t=n_t, beta=beta)
sd, tau, nodes = p["sd"], p["tau"], p["nodes"]
nx = sd.n_dofs
At = ko.assemble_with_bcs(sd.K * tau + sd.M * (1 + tau / beta ** 0.5), nodes).tocsr()
dinv = 1 / At.diagonal()
X = sd.coords
interior = np.ones(nx, bool)
interior[nodes] = False


def mkP(T):
    """bilinear interpolation from the (T + 1)^2 grid of tile corners (boundary corners dropped)"""
    rows, cols, vals = [], [], []
    for r in np.flatnonzero(interior):
        x, y = X[r, 0] * T, X[r, 1] * T
        i, j = min(int(x), T - 1), min(int(y), T - 1)
        fx, fy = x - i, y - j
        for di, wx in ((0, 1 - fx), (1, fx)):
            for dj, wy in ((0, 1 - fy), (1, fy)):
                ci, cj = i + di, j + dj
                if 0 < ci < T and 0 < cj < T and wx * wy > 0:
                    rows.append(r); cols.append((ci - 1) * (T - 1) + (cj - 1)); vals.append(wx * wy)
    return sp.csr_matrix((vals, (rows, cols)), shape=(nx, (T - 1) ** 2))


def mkZ(T):
    tid = np.minimum((X[:, 0] * T).astype(int), T - 1) * T + np.minimum((X[:, 1] * T).astype(int), T - 1)
    rows = np.flatnonzero(interior)
    return sp.csr_matrix((np.ones(rows.size), (rows, tid[rows])), shape=(nx, T * T))


rng = np.random.default_rng(0)
xs = [rng.standard_normal(nx) * interior, np.sin(np.pi * X[:, 0]) * np.sin(np.pi * X[:, 1]),
      np.sin(5 * np.pi * X[:, 0]) * np.sin(3 * np.pi * X[:, 1])]


def err(solver):
    out = []
    for x in xs:
        b = At @ x
        e = solver(b) - x
        out.append(float(np.sqrt(e @ (At @ e)) / np.sqrt(x @ (At @ x))))
    return " ".join(f"{v:.2e}" for v in out)


import scipy.sparse.linalg as spl
Dh = sp.diags(np.sqrt(dinv))
lmin = spl.eigsh((Dh @ At @ Dh).tocsr(), k=1, sigma=0, which="LM", return_eigenvectors=False)[0]
deg = int(np.ceil(1.6 * (2.0 / lmin) ** 0.5))
print(f"n {n} tau {tau:.4f} lmin {lmin:.3e} degree {deg}")
print(f"cheb{deg}", err(lambda b: ko.chebyshev_jacobi(At, dinv, b, 0.85 * lmin, 2.1, deg)))
Z = mkP(T) if os.environ.get("PROTO_COARSE", "bilinear") == "bilinear" else mkZ(T)
print("coarse space", Z.shape[1], "functions")
E = (Z.T @ At @ Z).toarray()
Ei = np.linalg.inv(E)


def tg(b, cyc, d, frac, order):
    x = np.zeros_like(b)
    r = b.copy()
    for c in range(cyc):
        for what in order:
            if what == "c":
                x += Z @ (Ei @ (Z.T @ r))
            else:
                x += ko.chebyshev_jacobi(At, dinv, r, 2.1 / frac, 2.1, d)
            r = b - At @ x
    return x


for cyc, d, frac in ((1, 8, 30), (2, 8, 30), (3, 8, 30), (3, 8, 100), (2, 12, 60), (2, 16, 100),
                     (4, 6, 20), (1, 24, 200), (3, 12, 100), (2, 12, 200), (2, 16, 300)):
    print(f"T {T} cycles {cyc} d {d} frac {frac}: coarse-first {err(lambda b: tg(b, cyc, d, frac, 'cs'))}"
          f" | smooth-first {err(lambda b: tg(b, cyc, d, frac, 'sc'))}"
          f" | s-c-s {err(lambda b: tg(b, cyc, d // 2, frac, 'scs'))}")
