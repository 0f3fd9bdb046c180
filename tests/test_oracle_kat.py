"""Pin the CPU oracle against the reference's known-answer tests (CPU only).

Restates ``test/test_control.py:26-119`` (stationary), ``1243-1444`` (instationary BE)
and ``1447-1655`` (instationary CN): closed-form ``x_ref``, hand-built ``b = A x_ref``,
FGMRES to 1e-14, ``||x - x_ref||_L2 < 1e-13``.  The hypre sub-solves of the
reference's built-in preconditioner are replaced by Jacobi-Chebyshev (north_star);
the converged solution does not depend on that choice.
"""
import numpy as np
import pytest

import kat
from control_amd.blocks import instationary_blocks, stationary_blocks
from oracle import kkt_oracle as ko

MASS = ko.ChebSpec(20, *kat.LAMBDA_V_BOUNDS)
# Q2 8x8, Jacobi-scaled shifted stiffness: generous interval, high degree (tiny problem)
SCHUR = ko.ChebSpec(40, 0.02, 2.2)


def test_stationary_linear_control():
    p = kat.kat_stationary()
    sd = p["sd"]
    b00, b01, b10, b11 = stationary_blocks(sd.M, p["D"], p["beta"])
    sysm = ko.OracleSystem(sd.n_dofs, sd.n_dofs, b00, b01, b10, b11,
                           nullspace_0=(ko.DirichletBCNullspace(p["nodes"]),),
                           nullspace_1=(ko.DirichletBCNullspace(p["nodes"]),))
    pc = ko.pc_stationary(sd.M, p["D"], p["D"].T.tocsr(), p["beta"], p["nodes"],
                          MASS, SCHUR)
    v = np.zeros((1, sd.n_dofs))
    z = np.zeros((1, sd.n_dofs))
    res = sysm.solve(v, z, p["b_0"], p["b_1"],
                     solver_parameters=kat.SOLVER_PARAMETERS, pc_fn=pc)
    assert res.reason > 0
    assert kat.l2_norm(sd.M, v - p["v_ref"]) < 1.0e-13
    assert kat.l2_norm(sd.M, z - p["z_ref"]) < 1.0e-13


@pytest.mark.parametrize("CN", [False, True])
def test_instationary_linear_control(CN):
    p = kat.kat_instationary_CN() if CN else kat.kat_instationary_BE()
    sd, n_t, tau, beta = p["sd"], p["n_t"], p["tau"], p["beta"]
    b00, b01, b10, b11, m = instationary_blocks(sd.M, sd.K, tau, beta, n_t, CN)
    ns = tuple(ko.DirichletBCNullspace(p["nodes"]) for _ in range(m))
    sysm = ko.OracleSystem(sd.n_dofs, sd.n_dofs, b00, b01, b10, b11,
                           n_blocks_00=m, n_blocks_11=m, nullspace_0=ns,
                           nullspace_1=ns, CN=CN)
    if CN:
        pc = ko.pc_instationary_CN(sd.M, b01, b10, n_t, tau, beta, p["nodes"],
                                   MASS, SCHUR)
        # the library transforms the supplied rows itself (control.py:3242-3243)
        b_0 = ko.apply_T_1(p["b_0"])
        b_1 = ko.apply_T_2(p["b_1"])
    else:
        pc = ko.pc_instationary_BE(sd.M, b01, b10, n_t, tau, beta, p["nodes"],
                                   MASS, SCHUR)
        b_0, b_1 = p["b_0"], p["b_1"]
    v = np.zeros((m, sd.n_dofs))
    z = np.zeros((m, sd.n_dofs))
    res = sysm.solve(v, z, b_0, b_1, solver_parameters=kat.SOLVER_PARAMETERS,
                     pc_fn=pc)
    assert res.reason > 0
    if CN:       # control.py:3300-3309: v block i is time level i+1, zeta block i level i
        v_full = np.vstack([np.zeros((1, sd.n_dofs)), v])
        z_full = np.vstack([z, np.zeros((1, sd.n_dofs))])
    else:
        v_full, z_full = v, z
    assert kat.l2_norm(sd.M, v_full - p["v_ref"]) < 1.0e-13
    assert kat.l2_norm(sd.M, z_full - p["z_ref"]) < 1.0e-13
