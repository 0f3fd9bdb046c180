#!/usr/bin/env python3
"""Generate the golden fixtures of tests/golden/ (BASELINE.json configs[0] sizes).

The reference cannot be imported in this pipeline (Firedrake/PETSc are not installed),
so these vectors come from this repository's CPU oracle, which is itself pinned by the
reference's known-answer tests (tests/test_oracle_kat.py).  They freeze the oracle's
outputs so that (a) a change to the oracle is noticed and (b) the GPU path can be checked
on a box where only the fixtures travel.

    python tests/golden/make_golden.py

Config: 2-D heat control, UnitSquareMesh 10x10 P1, n_t = 10, T = 2, homogeneous Dirichlet
boundary; beta = 1e-4 for operator/preconditioner vectors (README.md:14, 63-64) and
beta = 1e-2 for the Krylov histories (conditioning, see test_krylov_iterates_parity).
Inputs: numpy.random.default_rng(20241008 + k).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

import common  # noqa: E402

MASS = (20, 0.5, 2.0)
SCHUR = (10, 0.2, 2.1)
KSCHUR = (12, 0.08, 2.1)


def make(CN):
    out = {}
    p = common.heat_problem(n=10, n_t=10, CN=CN, beta=1e-4)
    osys = common.oracle_system(p)
    xs = np.stack([common.rng_vector(osys.N, common.SEED + k) for k in range(3)])
    out["x"] = xs
    out["Ax"] = np.stack([osys.mult(x) for x in xs])
    out["pc_x"] = osys.pc_apply(common.oracle_pc(p, MASS, SCHUR), xs[0])
    q = common.heat_problem(n=10, n_t=10, CN=CN, beta=1e-2)
    qsys = common.oracle_system(q)
    m, nx = q["m"], q["sd"].n_dofs
    X = q["sd"].coords
    xstar = np.stack([np.sin(np.pi * X[:, 0]) * np.sin(np.pi * X[:, 1]) * (1 + 0.1 * k)
                      for k in range(2 * m)])
    b = qsys.mult(xstar.ravel()).reshape(2 * m, nx)
    out["krylov_b"] = b
    for ksp in ("gmres", "fgmres"):
        sp = {"linear_solver": ksp, "gmres_restart": 10, "maximum_iterations": 60,
              "relative_tolerance": 1e-6, "absolute_tolerance": 0.0,
              "monitor_convergence": False, "preconditioner": True}
        u0, u1 = np.zeros((m, nx)), np.zeros((m, nx))
        r = qsys.solve(u0, u1, b[:m], b[m:], solver_parameters=sp,
                       pc_fn=common.oracle_pc(q, MASS, KSCHUR))
        out[f"{ksp}_history"] = np.asarray(r.history)
        out[f"{ksp}_its"] = np.int64(r.its)
        out[f"{ksp}_reason"] = np.int64(r.reason)
        out[f"{ksp}_solution"] = np.vstack([u0, u1])
    return out


if __name__ == "__main__":
    for CN in (False, True):
        name = os.path.join(HERE, f"config1_{'CN' if CN else 'BE'}.npz")
        np.savez_compressed(name, **make(CN))
        print(name, os.path.getsize(name), "bytes")
