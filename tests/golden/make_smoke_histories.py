#!/usr/bin/env python3
"""Residual histories of the smoke problem (__graft_entry__.smoke: 10x10 P1, n_t = 10,
beta = 1e-2, FGMRES(10) to rtol 1e-9) from BOTH sides -- the CPU oracle and the HIP path --
for BE and CN.  Needs an MI355X (run through gpurun; the file lands in gpurun_out/ and is
copied to tests/golden/smoke_histories.npz).  The two BE trajectories separate after a few
iterations (classical Gram-Schmidt on an ill-conditioned preconditioned operator,
tests/test_oracle.py::test_BE_iterates_are_ill_conditioned); they are committed so that the
separation is on record as data, and tests/test_golden.py checks what can be checked: the
oracle reproduces its own history, the CN histories agree, both BE runs converge to the same
solution."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))

import common  # noqa: E402

MASS, SCHUR = (20, 0.5, 2.0), (12, 0.08, 2.1)
SP = {"linear_solver": "fgmres", "fgmres_restart": 10, "maximum_iterations": 300,
      "relative_tolerance": 1e-9, "absolute_tolerance": 0.0, "monitor_convergence": False,
      "preconditioner": True}


def main(out):
    data = {}
    for CN in (False, True):
        tag = "CN" if CN else "BE"
        p = common.heat_problem(n=10, n_t=10, CN=CN, beta=1e-2)
        osys, gsys = common.oracle_system(p), common.gpu_system(p)
        m, nx = p["m"], p["sd"].n_dofs
        b = common.rng_vector(osys.N).reshape(2 * m, nx)
        uo = [np.zeros((m, nx)), np.zeros((m, nx))]
        ug = [np.zeros((m, nx)), np.zeros((m, nx))]
        ro = osys.solve(*uo, b[:m], b[m:], solver_parameters=SP,
                        pc_fn=common.oracle_pc(p, MASS, SCHUR))
        rg = gsys.solve(*ug, b[:m].copy(), b[m:].copy(), solver_parameters=SP,
                        pc_fn=common.gpu_pc(p, MASS, SCHUR))
        data[f"{tag}_oracle_history"] = np.asarray(ro.history)
        data[f"{tag}_gpu_history"] = np.asarray(rg.history)
        data[f"{tag}_oracle_solution"] = np.vstack(uo)
        data[f"{tag}_gpu_solution"] = np.vstack(ug)
        print(tag, "iterations oracle / gpu:", ro.its, rg.its, "solution deviation",
              common.rel_err(np.vstack(ug), np.vstack(uo)))
    np.savez_compressed(out, **data)
    print(out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "smoke_histories.npz"))
