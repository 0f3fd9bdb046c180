"""Step-locked Krylov parity (a-7: preconditioner.py:732-759, PETSc's KSPSolve_GMRES / FGMRES).

Two fp64 implementations of GMRES with classical Gram-Schmidt separate exponentially along a
trajectory (tests/test_oracle.py::test_BE_iterates_are_ill_conditioned), so comparing whole
residual histories tests conditioning, not the implementation.  Single steps are well
conditioned: here every inner step of the device loop starts from the ORACLE's basis
(kkt_debug_set_steplock) and what it produces -- the Gram-Schmidt coefficients h, ||w||, the
new basis vector, the Givens-updated residual norm, and the iterate built at the end of every
cycle -- is compared with the oracle's step from the same state.
"""
import ctypes as C

import numpy as np
import pytest

import common
from control_amd import _lib

pytestmark = pytest.mark.gpu

MASS, SCHUR = (20, 0.5, 2.0), (12, 0.08, 2.1)
RESTART = 10


def run_locked(p, ksp, side, n_steps):
    osys, gsys = common.oracle_system(p), common.gpu_system(p)
    opc, gpc = common.oracle_pc(p, MASS, SCHUR), common.gpu_pc(p, MASS, SCHUR)
    m, nx = p["m"], p["sd"].n_dofs
    N = 2 * m * nx
    b = common.rng_vector(N, 11).reshape(2 * m, nx)
    sp = {"linear_solver": ksp, "gmres_restart": RESTART, "maximum_iterations": n_steps,
          "relative_tolerance": 0.0, "absolute_tolerance": 0.0, "divergence limit": 1e300,
          "monitor_convergence": False, "preconditioner": True}
    if side:
        sp["pc_side"] = side
    trace = []
    uo = [np.zeros((m, nx)), np.zeros((m, nx))]
    ro = osys.solve(*uo, b[:m], b[m:], solver_parameters=sp, pc_fn=opc, trace=trace)
    assert len(trace) == n_steps
    V = np.zeros((n_steps, RESTART + 1, N))
    for s, t in enumerate(trace):
        assert t["its"] == s and t["it"] == s % RESTART
        V[s, :t["it"] + 1] = t["V"]
    h = np.zeros((n_steps, RESTART + 2))
    v_next = np.zeros((n_steps, N))
    lock = _lib.StepLock(n_steps=n_steps, restart=RESTART, V=V.ctypes.data_as(_lib.c_f64p),
                         h=h.ctypes.data_as(_lib.c_f64p),
                         v_next=v_next.ctypes.data_as(_lib.c_f64p))
    gsys._ck(gsys._lib.kkt_debug_set_steplock(gsys.handle, C.byref(lock)))
    ug = [np.zeros((m, nx)), np.zeros((m, nx))]
    rg = gsys.solve(*ug, b[:m].copy(), b[m:].copy(), solver_parameters=sp, pc_fn=gpc)
    gsys._ck(gsys._lib.kkt_debug_set_steplock(gsys.handle, None))
    return trace, h, v_next, ro, rg, np.vstack(uo), np.vstack(ug)


@pytest.mark.parametrize("ksp,side", [("gmres", None), ("gmres", "right"), ("fgmres", None)])
@pytest.mark.parametrize("CN", [False, True])
def test_every_krylov_step_from_the_oracle_state(ksp, side, CN):
    """25 steps of GMRES(10) (two restarts) on the beta = 1e-4 system -- the one whose BE
    trajectories cannot be compared -- step by step."""
    p = common.heat_problem(n=10, n_t=10, CN=CN, beta=1e-4)
    n_steps = 25
    trace, h, v_next, ro, rg, uo, ug = run_locked(p, ksp, side, n_steps)
    worst = dict(h=0.0, tt=0.0, w=0.0, v=0.0)
    for s, t in enumerate(trace):
        it = t["it"]
        w0 = np.sqrt(np.sum(t["h"] ** 2) + t["tt"] ** 2)      # ||w|| before the projection
        e_h = np.max(np.abs(h[s, :it + 1] - t["h"])) / w0
        e_tt = abs(h[s, it + 1] - t["tt"]) / w0
        e_w = np.linalg.norm(v_next[s] * h[s, it + 1] - t["v_next"] * t["tt"]) / w0
        e_v = np.linalg.norm(v_next[s] - t["v_next"])
        worst = dict(h=max(worst["h"], e_h), tt=max(worst["tt"], e_tt), w=max(worst["w"], e_w),
                     v=max(worst["v"], e_v))
        # the step is backward stable: everything it produces agrees to 1e-12 of its input
        assert e_h < 1e-12 and e_tt < 1e-12 and e_w < 1e-12, (s, e_h, e_tt, e_w)
        # the normalised vector amplifies by the cancellation ||w0|| / ||w|| of classical
        # Gram-Schmidt; still 1e-10 here
        assert e_v < 1e-10 * max(1.0, w0 / t["tt"]), (s, e_v, w0 / t["tt"])
    print(f"step-locked {ksp} {side or 'default'} {'CN' if CN else 'BE'}: worst {worst}")
    # Givens-updated norms (host arithmetic on the device's h) and the iterate after the last
    # cycle: same count, same values
    ho, hg = np.asarray(ro.history), np.asarray(rg.history)
    assert len(ho) == len(hg) and ro.its == rg.its == n_steps
    # (host-side Givens recurrences on h that agrees to 1e-15, and the restart norms from each
    # side's own iterate: the relative deviation grows from 1e-15 to 1.2e-9 over the 25 steps on
    # the BE system with right preconditioning)
    # (CN converges to round-off within the 25 steps: norms of 1e-12 ||r_0|| carry no digits, so
    # the bar is relative to the norm itself and to the first one)
    assert np.all(np.abs(hg - ho) <= 1e-8 * ho + 1e-11 * ho[0])
    assert common.rel_err(ug, uo) < 1e-8


@pytest.mark.parametrize("CN", [False, True])
def test_iterate_after_every_step_count(CN):
    """x_k for k = 1 .. 12 (through one restart): the solution update of KSPGMRESBuildSoln."""
    p = common.heat_problem(n=10, n_t=10, CN=CN, beta=1e-4)
    for k in range(1, 13):
        _, _, _, ro, rg, uo, ug = run_locked(p, "gmres", None, k)
        assert ro.its == rg.its == k
        assert common.rel_err(ug, uo) < 1e-9, k
