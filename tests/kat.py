"""Known-answer problems restated from the reference's own tests (data only).

Each builder returns the closed-form reference solution and a right-hand side
``b = A x_ref`` assembled row by row from mass/stiffness products, independently of
the block dictionaries and of ``MultiBlockSystem`` -- exactly how the reference tests
build them:

* ``kat_stationary``      <- ``test/test_control.py:26-119``
* ``kat_instationary_BE`` <- ``test/test_control.py:1243-1444``
* ``kat_instationary_CN`` <- ``test/test_control.py:1447-1655``

Space: Q2 on ``UnitSquareMesh(8, 8, quadrilateral=True)``; beta = 1e-3; n_t = 5,
tau = 0.25; Chebyshev bounds (0.25, 1.5625) for the Jacobi-scaled Q2 mass matrix;
FGMRES to rtol = atol = 1e-14, at most 500 iterations; pass bar
``||x - x_ref||_{L2} < 1e-13``.
"""
import numpy as np

from control_amd.fem import unit_square_q2

LAMBDA_V_BOUNDS = (0.25, 1.5625)          # test_control.py:93, 1416, 1627
SOLVER_PARAMETERS = {"linear_solver": "fgmres",
                     "fgmres_restart": 10,    # ignored by the reference
                     "maximum_iterations": 500,
                     "relative_tolerance": 1.0e-14,
                     "absolute_tolerance": 1.0e-14,
                     "monitor_convergence": False}
BETA = 1.0e-3
N_T = 5
TAU = 0.25


def space():
    return unit_square_q2(8)


def l2_norm(M, e):
    """sqrt(assemble(inner(e, e) * dx)) for a block vector e of shape (n, nx)."""
    e = np.atleast_2d(e)
    return np.sqrt(abs(sum(float(ei @ (M @ ei)) for ei in e)))


def kat_stationary():
    sd = space()
    X, Y = sd.coords[:, 0], sd.coords[:, 1]
    M, K = sd.M, sd.K
    D = K + M                                     # forw_diff_operator, :34-36
    v_ref = X * np.exp(Y)
    z_ref = np.sin(np.pi * X) * np.sin(2.0 * np.pi * Y)
    b_0 = M @ v_ref + K @ z_ref + M @ z_ref       # :83-86
    b_1 = K @ v_ref + M @ v_ref - (1.0 / BETA) * (M @ z_ref)   # :87-90
    return dict(sd=sd, D=D, beta=BETA, v_ref=v_ref[None, :], z_ref=z_ref[None, :],
                b_0=b_0[None, :], b_1=b_1[None, :], nodes=np.zeros(0, dtype=np.int32))


def _refs(sd):
    X, Y = sd.coords[:, 0], sd.coords[:, 1]
    tau = TAU
    s34 = np.sin(3.0 * np.pi * X) * np.sin(4.0 * np.pi * Y)
    s12 = np.sin(np.pi * X) * np.sin(2.0 * np.pi * Y)
    xe = X * np.exp(Y) * s12
    v_ref = np.stack([0.0 * X, tau * s34, tau**2 * xe, tau**3 * s34, tau**4 * xe])
    z_ref = np.stack([s12, tau * s34, tau**2 * s12, tau**3 * s34, 0.0 * X])
    return v_ref, z_ref


def kat_instationary_BE():
    sd = space()
    M, K = sd.M, sd.K
    tau, beta, n_t = TAU, BETA, N_T
    v, z = _refs(sd)
    b_0 = np.zeros_like(v)
    b_1 = np.zeros_like(v)
    for i in range(n_t - 1):                       # :1334-1362
        b_0[i] = tau * (M @ v[i]) + tau * (K @ z[i]) + M @ z[i] - M @ z[i + 1]
    b_0[n_t - 1] = tau * (K @ z[n_t - 1]) + M @ z[n_t - 1]          # :1364-1366
    b_1[0] = tau * (K @ v[0]) + M @ v[0]                             # :1374-1376
    for i in range(1, n_t):                        # :1380-1408
        b_1[i] = (tau * (K @ v[i]) + M @ v[i] - M @ v[i - 1]
                  - (tau / beta) * (M @ z[i]))
    return dict(sd=sd, beta=beta, tau=tau, n_t=n_t, v_ref=v, z_ref=z,
                b_0=b_0, b_1=b_1, nodes=sd.boundary)


def kat_instationary_CN():
    sd = space()
    M, K = sd.M, sd.K
    tau, beta, n_t = TAU, BETA, N_T
    h = 0.5 * tau
    v, z = _refs(sd)
    m = n_t - 1
    b_0 = np.zeros((m, sd.n_dofs))
    b_1 = np.zeros((m, sd.n_dofs))
    # unknown block i holds v_{i+1} and zeta_i (control.py:3307-3309)
    for i in range(m):                             # :1541-1576
        b_0[i] = h * (M @ v[i + 1]) + h * (K @ z[i]) + M @ z[i]
        if i >= 1:
            b_0[i] += h * (M @ v[i])
        if i + 1 < m:
            b_0[i] += h * (K @ z[i + 1]) - M @ z[i + 1]
    for i in range(m):                             # :1585-1620
        b_1[i] = h * (K @ v[i + 1]) + M @ v[i + 1] - (h / beta) * (M @ z[i])
        if i >= 1:
            b_1[i] += h * (K @ v[i]) - M @ v[i]
        if i + 1 < m:
            b_1[i] -= (h / beta) * (M @ z[i + 1])
    return dict(sd=sd, beta=beta, tau=tau, n_t=n_t, v_ref=v, z_ref=z,
                b_0=b_0, b_1=b_1, nodes=sd.boundary)


def kat_stationary_incompressible():
    """``test/test_control.py:232-358``: Q2-Q1 on a 4x4 quadrilateral mesh, forward operator
    grad-grad + mass, beta = 1e-3, homogeneous Dirichlet velocity, ConstantNullspace on both
    pressures, Chebyshev bounds (0.25, 1.5625) velocity mass and (0.25, 2.25) pressure mass,
    FGMRES to 1e-15, all four fields within 1e-13 in L2 (pressures up to their means)."""
    from control_amd.fem import unit_square_q2q1
    th = unit_square_q2q1(4)
    X, Y = th.coords_v[:, 0], th.coords_v[:, 1]
    xp, yp = th.coords_p[:, 0], th.coords_p[:, 1]
    s12 = np.sin(np.pi * X) * np.sin(2.0 * np.pi * Y)
    s34 = np.sin(3.0 * np.pi * X) * np.sin(4.0 * np.pi * Y)
    v_ref = np.concatenate([X * np.exp(Y) * s12, s34])            # :278-280
    z_ref = np.concatenate([s12, s34])                            # :281-283
    p_ref = np.sin(np.pi * xp) * np.sin(2.0 * np.pi * yp)         # :287
    mu_ref = xp * np.exp(yp)                                      # :288
    beta = BETA
    D = th.K_v + th.M_v                                           # :243-245
    BT = th.B.T
    b_0 = th.M_v @ v_ref + D @ z_ref + BT @ mu_ref                # :295-298
    b_1 = D @ v_ref - (1.0 / beta) * (th.M_v @ z_ref) + BT @ p_ref   # :299-302
    b_2 = th.B @ v_ref                                            # :303
    b_3 = th.B @ z_ref                                            # :304
    return dict(th=th, D=D, D_p=th.K_p + th.M_p, beta=beta, v_ref=v_ref, z_ref=z_ref,
                p_ref=p_ref, mu_ref=mu_ref, b_0=np.stack([b_0, b_1]), b_1=np.stack([b_2, b_3]),
                lambda_p_bounds=(0.25, 2.25),
                solver_parameters={"linear_solver": "fgmres", "fgmres_restart": 10,
                                   "maximum_iterations": 500, "relative_tolerance": 1.0e-15,
                                   "absolute_tolerance": 1.0e-15,
                                   "monitor_convergence": False})
