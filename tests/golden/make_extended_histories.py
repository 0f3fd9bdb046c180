"""Which BE trajectory is the accurate one?  ("settle the BE trajectory question".)

The free-running FGMRES(10) trajectories of the HIP path and of the fp64 oracle on the smoke
problem separate after a few steps (105 against 127 iterations to rtol 1e-9 in the driver's smoke
run): classical Gram-Schmidt without refinement cancels ~3 digits per step and the BE
preconditioner scales the final-time block by 1 / epsilon = 1e3.  This script runs the SAME
algorithm -- the oracle's operator, preconditioner and FGMRES driver, unchanged code -- in
numpy.longdouble (x87 extended precision, unit round-off 1.1e-19 against 2.2e-16): every SpMV,
Chebyshev step, dot product, update and Givens rotation carries 11 more bits.  Its history is the
reference trajectory; the fp64 oracle history (recomputed here) and the HIP history (recorded on an
MI355X in tests/golden/smoke_histories.npz, same problem, same seeded right-hand side) are compared
with it step by step.  Writes tests/golden/extended_histories.npz.

    python tests/golden/make_extended_histories.py
"""
import os
import sys

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import common                                      # noqa: E402
from oracle import kkt_oracle as ko                # noqa: E402

LD = np.longdouble
MASS, SCHUR = (20, 0.5, 2.0), (12, 0.08, 2.1)      # __graft_entry__.smoke()


def system(p, dtype):
    cache = {}

    def cast(A):
        if A is None:
            return None
        if id(A) not in cache:
            cache[id(A)] = sp.csr_matrix(A).astype(dtype)
        return cache[id(A)]
    blocks = tuple({k: cast(v) for k, v in b.items()} for b in p["blocks"])
    sd, m = p["sd"], p["m"]
    ns = tuple(ko.DirichletBCNullspace(p["nodes"]) for _ in range(m))
    osys = ko.OracleSystem(sd.n_dofs, sd.n_dofs, *blocks, n_blocks_00=m, n_blocks_11=m,
                           nullspace_0=ns, nullspace_1=ns, CN=p["CN"], dtype=dtype)
    f = ko.pc_instationary_CN if p["CN"] else ko.pc_instationary_BE
    pc = f(cast(sd.M), blocks[1], blocks[2], p["n_t"], dtype(p["tau"]), dtype(p["beta"]),
           p["nodes"], ko.ChebSpec(*MASS), ko.ChebSpec(*SCHUR))
    return osys, pc


def history(p, dtype, max_it=300):
    osys, pc = system(p, dtype)
    m, nx = p["m"], p["sd"].n_dofs
    b = common.rng_vector(osys.N).astype(dtype).reshape(2 * m, nx).copy()
    b[:, p["nodes"]] = 0                            # correct_rhs (preconditioner.py:658-704)
    x = np.zeros(osys.N, dtype=dtype)
    res = ko.fgmres(osys.mult, lambda v: osys.pc_apply(pc, v), b.ravel(), x, restart=30,     # "fgmres_restart" is not read: PETSc default 30
                    rtol=dtype(1e-9), atol=dtype(0), divtol=dtype(1e4), max_it=max_it)
    assert x.dtype == dtype
    return np.array(res.history, dtype=dtype), x, res.its


def main(out):
    rec = np.load(os.path.join(HERE, "smoke_histories.npz"))
    data = {}
    for CN in (False, True):
        tag = "CN" if CN else "BE"
        p = common.heat_problem(n=10, n_t=10, CN=CN, beta=1e-2)
        h_ld, x_ld, its_ld = history(p, LD)
        h_64, x_64, its_64 = history(p, np.float64)
        h_gpu = rec[f"{tag}_gpu_history"]
        assert np.array_equal(h_64, rec[f"{tag}_oracle_history"]), "the recorded oracle history"
        data[f"{tag}_extended_history"] = h_ld.astype(np.float64)
        data[f"{tag}_oracle_history"] = h_64
        data[f"{tag}_gpu_history"] = h_gpu
        data[f"{tag}_extended_solution"] = x_ld.astype(np.float64)
        n = min(len(h_ld), len(h_64), len(h_gpu))
        e64 = np.abs(h_64[:n] - h_ld[:n].astype(float)) / h_ld[:n].astype(float)
        egp = np.abs(h_gpu[:n] - h_ld[:n].astype(float)) / h_ld[:n].astype(float)
        print(f"{tag}: iterations extended {its_ld}, fp64 oracle {its_64}, HIP {len(h_gpu) - 1}")
        for k in list(range(0, min(n, 60), 4)):
            print(f"   step {k:3d}  extended {float(h_ld[k]):.6e}  |oracle - ext| / ext {e64[k]:.1e}  "
                  f"|HIP - ext| / ext {egp[k]:.1e}")
        sol_o = rec[f"{tag}_oracle_solution"].ravel()
        sol_g = rec[f"{tag}_gpu_solution"].ravel()
        xl = x_ld.astype(float)
        print(f"   converged solutions against the extended one: oracle "
              f"{np.linalg.norm(sol_o - xl) / np.linalg.norm(xl):.1e}, HIP "
              f"{np.linalg.norm(sol_g - xl) / np.linalg.norm(xl):.1e}")
    np.savez_compressed(out, **data)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "extended_histories.npz"))
