"""The C/OpenMP restatement (cpu_baseline of bench.py) against the NumPy oracle, and --
on the GPU -- bit-for-bit against the HIP SpMV (same fma chain in the same order)."""
import numpy as np
import pytest

import common
from oracle import cref

MASS, SCHUR = (20, 0.5, 2.0), (12, 0.08, 2.1)


def build(share=True, n=10, n_t=10, beta=1e-2):
    p = common.heat_problem(n=n, n_t=n_t, CN=False, beta=beta, share=share,
                            time_dependent=not share)
    c = cref.CRef(p["blocks"], p["m"], p["sd"].n_dofs, p["nodes"], p["sd"].M, p["n_t"],
                  p["tau"], p["beta"], MASS, SCHUR)
    return p, c


@pytest.mark.parametrize("share", [True, False])
def test_c_restatement_matches_numpy_oracle(share):
    p, c = build(share)
    osys = common.oracle_system(p)
    x = common.rng_vector(osys.N)
    assert common.rel_err(c.mult(x), osys.mult(x)) < 1e-13
    opc = common.oracle_pc(p, MASS, SCHUR)
    assert common.rel_err(c.pc_apply(x), osys.pc_apply(opc, x)) < 1e-11
    m, nx = p["m"], p["sd"].n_dofs
    X = p["sd"].coords
    xs = np.stack([np.sin(np.pi * X[:, 0]) * np.sin(np.pi * X[:, 1]) * (1 + 0.1 * k)
                   for k in range(2 * m)])
    b = osys.mult(xs.ravel())
    sp = {"linear_solver": "gmres", "gmres_restart": 10, "maximum_iterations": 60,
          "relative_tolerance": 1e-6, "absolute_tolerance": 0.0,
          "monitor_convergence": False, "preconditioner": True}
    u0, u1 = np.zeros((m, nx)), np.zeros((m, nx))
    B = b.reshape(2 * m, nx)
    ro = osys.solve(u0, u1, B[:m], B[m:], solver_parameters=sp, pc_fn=opc)
    xc, its, reason, hist = c.gmres(b, np.zeros_like(b), max_it=60)
    assert reason == ro.reason and abs(its - ro.its) <= 1
    assert np.max(np.abs(hist[:3] - np.asarray(ro.history)[:3]) / np.asarray(ro.history)[:3]) < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("share", [True, False])
def test_gpu_spmv_is_bit_identical_to_c_restatement(share):
    """Integer-exact parity bar applied to fp64: the HIP block-row kernel and the C loop run
    the same fma chain per row (blocks in the reference's dict order, CSR order inside a
    block), so the KKT operator results must be identical to the last bit."""
    p, c = build(share, beta=1e-4)
    gsys = common.gpu_system(p)
    for seed in range(3):
        x = common.rng_vector(2 * p["m"] * p["sd"].n_dofs, common.SEED + seed)
        assert np.array_equal(gsys.mult(x), c.mult(x))
