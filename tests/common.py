"""Shared builders for the parity tests: one problem -> (oracle system, GPU system)."""
import numpy as np

from control_amd.problems import (SEED, STOKES_SPECS, gpu_pc, gpu_system, heat_problem,  # noqa: F401
                                  rng_vector, stokes_gpu, stokes_problem)


def oracle_system(p):
    from oracle import kkt_oracle as ko
    sd, m = p["sd"], p["m"]
    ns = tuple(ko.DirichletBCNullspace(p["nodes"]) for _ in range(m))
    return ko.OracleSystem(sd.n_dofs, sd.n_dofs, *p["blocks"], n_blocks_00=m,
                           n_blocks_11=m, nullspace_0=ns, nullspace_1=ns, CN=p["CN"])


def oracle_pc(p, mass, schur, coarse=None):
    from oracle import kkt_oracle as ko
    sd = p["sd"]
    b00, b01, b10, b11 = p["blocks"]
    f = ko.pc_instationary_CN if p["CN"] else ko.pc_instationary_BE
    sch = ko.ChebSpec(*schur)
    if coarse is not None:
        sch.coarse = ko.CoarseSpace(coarse[0], int(coarse[1]))
    return f(sd.M, b01, b10, p["n_t"], p["tau"], p["beta"], p["nodes"],
             ko.ChebSpec(*mass), sch)


def rel_err(a, b):
    return np.linalg.norm(np.ravel(a) - np.ravel(b)) / max(np.linalg.norm(np.ravel(b)), 1e-300)


# ------------------------------------------ instationary Stokes control



def stokes_oracle(p, specs=STOKES_SPECS, coarse=None, kp_coarse=None):
    from oracle import kkt_oracle as ko
    th, m, CN, bl = p["th"], p["m"], p["CN"], p["blocks"]
    kw = dict(sub_n_blocks_00_0=m, sub_n_blocks_11_0=m) if CN else {}
    osys = ko.OracleSystem(th.n_v, th.n_p, *bl["outer"], n_blocks_00=2 * m, n_blocks_11=2 * m,
                           nullspace_0=tuple(ko.DirichletBCNullspace(th.boundary_v)
                                             for _ in range(2 * m)),
                           nullspace_1=tuple(ko.ConstantNullspace() for _ in range(2 * m)),
                           CN=CN, **kw)
    schur = ko.ChebSpec(*specs["schur"])
    if coarse is not None:
        schur.coarse = ko.CoarseSpace(coarse[0], int(coarse[1]))
    kp = ko.ChebSpec(*specs["kp"])
    if kp_coarse is not None:      # two-grid K_p solve, constants deflated
        kp.coarse = ko.CoarseSpace(kp_coarse[0], int(kp_coarse[1]), deflate=True)
    opc = ko.pc_instationary_incompressible(
        th.M_v, bl["inner"], th.B, th.M_p, th.K_p, bl["commutator"], p["n_t"], p["tau"],
        p["beta"], th.boundary_v, ko.ChebSpec(*specs["mass"]), schur,
        kp, ko.ChebSpec(*specs["mp"]), CN=CN)
    return osys, opc


# ------------------------------------------ Navier-Stokes control, Picard loop

NS_SOLVER_PARAMETERS = {"linear_solver": "fgmres", "fgmres_restart": 10,
                        "maximum_iterations": 100, "relative_tolerance": 1.0e-8,
                        "absolute_tolerance": 0.0, "monitor_convergence": False}


def navier_stokes_problem(n=4, n_t=4, nu=0.1, beta=1.0e-2, T=2.0, CN=False):
    """A small instance shaped like ``test/test_control.py:4160-4270`` (P2-P1 on
    ``RectangleMesh(n, n, 2, 2)``, Picard convection, zero force and initial state) with a
    smooth desired state that vanishes on the boundary."""
    from control_amd.fem import rectangle_p2p1
    from control_amd.picard import NavierStokesControl
    th = rectangle_p2p1(n, n, 2.0, 2.0)
    X, Y = th.coords_v[:, 0], th.coords_v[:, 1]
    tau = T / (n_t - 1.0)
    v_d = np.stack([np.cos(0.5 * np.pi * i * tau) * np.concatenate([
        np.sin(0.5 * np.pi * X) ** 2 * np.sin(np.pi * Y),
        -np.sin(np.pi * X) * np.sin(0.5 * np.pi * Y) ** 2]) for i in range(n_t)])
    return NavierStokesControl(disc=th, nu=nu, beta=beta, n_t=n_t, T=T, v_d=v_d,
                               f=np.zeros((n_t, th.n_v)), CN=CN)


def navier_stokes_cavity_problem(n=8, n_t=10, CN=False):
    """``test/test_control.py:4171-4268`` (BE) / ``4271-4368`` (CN): Navier-Stokes control in the
    lid-driven cavity.  P2-P1 on ``RectangleMesh(n, n, 2, 2)``, nu = 1/100, beta = 1e-3,
    T = 2; the lid (boundary 4, ``y = 2``; its corners belong to the no-slip walls because the
    wall condition is applied last) moves with ``(min(t, 1), 0)``; the desired state is a pair
    of counter-rotating vortices modulated by ``cos(pi t / 2)``; zero force and initial state.
    Returns the problem and the initial iterate that carries the Dirichlet values of every
    time level (``control.py:4925-4959``: the loop starts from ``v`` with the conditions
    applied and solves for updates that vanish on the boundary)."""
    from control_amd.fem import rectangle_p2p1
    from control_amd.picard import NavierStokesControl
    th = rectangle_p2p1(n, n, 2.0, 2.0)
    T = 2.0
    tau = T / (n_t - 1.0)
    nn = th.n_v // 2
    X, Y = th.coords_v[:nn, 0], th.coords_v[:nn, 1]
    x, y = X - 1.0, Y - 1.0
    a, b = (100.0 / 49.0) ** 2, (100.0 / 99.0) ** 2
    c_1 = 1.0 - np.sqrt(a * (x - 0.5) ** 2 + b * y ** 2)
    c_2 = 1.0 - np.sqrt(a * (x + 0.5) ** 2 + b * y ** 2)
    vx = np.where(c_1 >= 0.0, c_1 * b * y, np.where(c_2 >= 0.0, -c_2 * b * y, 0.0))
    vy = np.where(c_1 >= 0.0, -c_1 * a * (x - 0.5), np.where(c_2 >= 0.0, c_2 * a * (x + 0.5), 0.0))
    shape = np.concatenate([vx, vy])
    v_d = np.stack([np.cos(0.5 * np.pi * i * tau) * shape for i in range(n_t)])
    pb = NavierStokesControl(disc=th, nu=1.0 / 100.0, beta=1.0e-3, n_t=n_t, T=T, v_d=v_d,
                             f=np.zeros((n_t, th.n_v)), CN=CN)
    lid = np.flatnonzero((np.abs(Y - 2.0) < 1e-12) & (X > 1e-12) & (X < 2.0 - 1e-12))
    v_init = np.zeros((n_t, th.n_v))
    for i in range(n_t):
        v_init[i, lid] = min(i * tau, 1.0)           # first component on the lid nodes
    return pb, v_init, lid


def mms_navier_stokes_control(N, CN=False, n_t=10, nu=1.0 / 50.0):
    """``test/test_control.py:4371-4470`` (BE) / ``4740-4840`` (CN): Navier-Stokes control with
    the manufactured solution ``v = (T - t) (x y^3, (x^4 - y^4) / 4)`` (divergence free),
    ``zeta = 0``, ``v_d = v``; nu = 1/50, beta = 1e-3, T = 2, P2-P1 on
    ``RectangleMesh(N, N, 2, 2)``; the force is ``-nu/2 lapl(v) + (v . grad) v - v_xy`` (the
    other half of the viscous term is a gradient and lands in the pressure).  Returns the
    problem, the initial iterate (exact boundary values on every level, zero inside) and the
    exact velocity."""
    from control_amd.fem import rectangle_p2p1
    from control_amd.picard import NavierStokesControl
    th = rectangle_p2p1(N, N, 2.0, 2.0)
    T, beta = 2.0, 1.0e-3
    tau = T / (n_t - 1.0)
    nn = th.n_v // 2
    x, y = th.coords_v[:nn, 0] - 1.0, th.coords_v[:nn, 1] - 1.0
    u1, u2 = x * y**3, 0.25 * (x**4 - y**4)
    v_xy = np.concatenate([u1, u2])
    lapl = np.concatenate([6.0 * x * y, 3.0 * x**2 - 3.0 * y**2])
    conv = np.concatenate([u1 * y**3 + u2 * 3.0 * x * y**2,        # (v_xy . grad) v_xy
                           u1 * x**3 - u2 * y**3])

    def true_v(t):
        return (T - t) * v_xy

    f = np.stack([-0.5 * nu * (T - i * tau) * lapl + (T - i * tau) ** 2 * conv - v_xy
                  for i in range(n_t)])
    v_d = np.stack([true_v(i * tau) for i in range(n_t)])
    pb = NavierStokesControl(disc=th, nu=nu, beta=beta, n_t=n_t, T=T, v_d=v_d, f=f,
                             v_0=true_v(0.0), CN=CN)
    v_init = np.zeros((n_t, th.n_v))
    v_init[:, th.boundary_v] = v_d[:, th.boundary_v]
    return pb, v_init, true_v


class OracleLinearSolver:
    """The linearised solve of one Picard iteration in the CPU oracle (rebuilt every time)."""

    def __init__(self, pb, specs=STOKES_SPECS, solver_parameters=NS_SOLVER_PARAMETERS):
        self.pb, self.specs, self.sp = pb, specs, solver_parameters

    def linear_solve(self, D, Dp, b_0, b_1):
        from control_amd.blocks import instationary_incompressible_blocks
        from oracle import kkt_oracle as ko
        pb, th, s = self.pb, self.pb.disc, self.specs
        bl = instationary_incompressible_blocks(th.M_v, list(D), th.B, th.M_p, list(Dp),
                                                pb.tau, pb.beta, pb.n_t, pb.CN)
        m = bl["m"]
        kw = dict(sub_n_blocks_00_0=m, sub_n_blocks_11_0=m) if pb.CN else {}
        osys = ko.OracleSystem(
            th.n_v, th.n_p, *bl["outer"], n_blocks_00=2 * m, n_blocks_11=2 * m,
            nullspace_0=tuple(ko.DirichletBCNullspace(th.boundary_v) for _ in range(2 * m)),
            nullspace_1=tuple(ko.ConstantNullspace() for _ in range(2 * m)), CN=pb.CN, **kw)
        opc = ko.pc_instationary_incompressible(
            th.M_v, bl["inner"], th.B, th.M_p, th.K_p, bl["commutator"], pb.n_t, pb.tau,
            pb.beta, th.boundary_v, ko.ChebSpec(*s["mass"]), ko.ChebSpec(*s["schur"]),
            ko.ChebSpec(*s["kp"]), ko.ChebSpec(*s["mp"]), CN=pb.CN)
        u_0, u_1 = np.zeros_like(b_0), np.zeros_like(b_1)
        res = osys.solve(u_0, u_1, b_0, b_1, solver_parameters=self.sp, pc_fn=opc)
        return u_0, u_1, res.its


# ------------------------------------------ Instationary.linear_solve driver

class OracleBackend:
    """``control_amd.control`` backend interface on the CPU oracle."""

    def __init__(self, schur=(30, 0.02, 2.2)):
        from oracle import kkt_oracle as ko
        self._ko, self.schur = ko, schur
        self.DirichletBCNullspace = ko.DirichletBCNullspace

    def MultiBlockSystem(self, *a, **kw):
        return self._ko.OracleSystem(*a, **kw)

    def construct_pc(self, kind, M, block_01, block_10, n_t, tau, beta, nodes, lambda_v_bounds,
                     epsilon):
        ko = self._ko
        mass, schur = ko.ChebSpec(20, *lambda_v_bounds), ko.ChebSpec(*self.schur)
        if kind == "stationary":
            return ko.pc_stationary(M, block_10[(0, 0)], block_01[(0, 0)], beta, nodes, mass,
                                    schur)
        if kind == "CN":
            return ko.pc_instationary_CN(M, block_01, block_10, n_t, tau, beta, nodes, mass,
                                         schur)
        return ko.pc_instationary_BE(M, block_01, block_10, n_t, tau, beta, nodes, mass, schur,
                                     epsilon=epsilon)


def mms_heat_control(N, CN, n_t=10):
    """``test/test_control.py:1658-1760`` (BE) / ``1983-2085`` (CN): heat control on
    ``RectangleMesh(N, N, 2, 2)``, P1, beta = 1, exact solution linear in time,
    ``v = 1`` on the boundary (inhomogeneous Dirichlet data)."""
    from control_amd.control import Instationary
    from control_amd.fem import rectangle_p1
    disc = rectangle_p1(N, N, 2.0, 2.0)
    beta, t_f = 1.0, 2.0

    def c(X):
        return np.cos(0.5 * np.pi * (X[:, 0] - 1.0)) * np.cos(0.5 * np.pi * (X[:, 1] - 1.0))

    def ref_v(X, t):
        return 1.0 + (t_f - t) * c(X)

    def ref_zeta(X, t):
        return (t_f - t) * c(X)

    def desired_state(X, t):      # zeta_space - lapl(zeta) + v
        return c(X) + 0.5 * np.pi**2 * (t_f - t) * c(X) + ref_v(X, t)

    def force_f(X, t):            # - v_space - lapl(v) - zeta / beta
        return -c(X) + 0.5 * np.pi**2 * (t_f - t) * c(X) - ref_zeta(X, t) / beta

    ctl = Instationary(disc, desired_state=desired_state, force_f=force_f, beta=beta, CN=CN,
                       n_t=n_t, initial_condition=lambda X: ref_v(X, 0.0),
                       time_interval=(0.0, t_f), bcs_v=lambda Xb, t: np.ones(len(Xb)))
    return ctl, disc, ref_v, ref_zeta


def mms_heat_control_in_time(N, CN, n_t):
    """``test/test_control.py:1829-1980`` (BE) / ``2140-2294`` (CN): heat control with an exact
    solution that is exponential in time -- ``zeta = (e^T - e^t) c(x)``, ``v = 1 + (c_1 +
    c_2(t)) c(x)``, ``f = 0`` -- so that the time discretisation error is what is measured
    (BE first order, CN second order).  The reference runs ``N = 250``; the spatial error of
    the meshes used here is below the time error of the coarse time grids compared."""
    from control_amd.control import Instationary
    from control_amd.fem import rectangle_p1
    disc = rectangle_p1(N, N, 2.0, 2.0)
    beta, t_f = 1.0, 2.0
    pi2 = np.pi * np.pi

    def c(X):
        return np.cos(0.5 * np.pi * (X[:, 0] - 1.0)) * np.cos(0.5 * np.pi * (X[:, 1] - 1.0))

    def ref_v(X, t):
        return 1.0 + ((2.0 / (pi2 * beta)) * np.exp(t_f)
                      - (2.0 / ((2.0 + pi2) * beta)) * np.exp(t)) * c(X)

    def ref_zeta(X, t):
        return (np.exp(t_f) - np.exp(t)) * c(X)

    def desired_state(X, t):
        return 1.0 + ((2.0 / (pi2 * beta) + 0.5 * pi2) * np.exp(t_f)
                      + (1.0 - 2.0 / ((2.0 + pi2) * beta) - 0.5 * pi2) * np.exp(t)) * c(X)

    ctl = Instationary(disc, desired_state=desired_state, force_f=None, beta=beta, CN=CN,
                       n_t=n_t, initial_condition=lambda X: ref_v(X, 0.0),
                       time_interval=(0.0, t_f), bcs_v=lambda Xb, t: np.ones(len(Xb)))
    return ctl, disc, ref_v, ref_zeta


def mms_convection_diffusion_control(N, CN, n_t=10):
    """``test/test_control.py:2297-2440``: the heat problem above with the time-dependent,
    divergence-free wind ``cos(pi t / 2) (2 y (1 - x^2), -2 x (1 - y^2))`` (coordinates shifted
    to the centre of ``[0, 2]^2``) in the state equation: ``forward_form = grad-grad +
    (wind . grad trial) test``, a non-symmetric operator that differs on every time level.
    The exact solution is the heat test's (linear in time).  The reference's CN variant
    (``2675-2860``) uses an exponential-in-time solution; with ``CN=True`` this helper keeps
    the BE data (still a manufactured solution of the same equations)."""
    from control_amd.control import Instationary
    from control_amd.fem import rectangle_p1
    disc = rectangle_p1(N, N, 2.0, 2.0)
    beta, t_f = 1.0, 2.0
    hp = 0.5 * np.pi

    def c(X):
        return np.cos(hp * (X[:, 0] - 1.0)) * np.cos(hp * (X[:, 1] - 1.0))

    def grad_c(X):
        x, y = X[:, 0] - 1.0, X[:, 1] - 1.0
        return np.stack([-hp * np.sin(hp * x) * np.cos(hp * y),
                         -hp * np.cos(hp * x) * np.sin(hp * y)], 1)

    def wind(X, t):
        x, y = X[:, 0] - 1.0, X[:, 1] - 1.0
        return np.cos(hp * t) * np.stack([2.0 * y * (1.0 - x * x), -2.0 * x * (1.0 - y * y)], 1)

    def ref_v(X, t):
        return 1.0 + (t_f - t) * c(X)

    def ref_zeta(X, t):
        return (t_f - t) * c(X)

    def w_grad_c(X, t):
        return np.einsum("nd,nd->n", wind(X, t), grad_c(X))

    def desired_state(X, t):      # zeta_space - lapl(zeta) - wind . grad(zeta) + v
        return (c(X) + 0.5 * np.pi**2 * (t_f - t) * c(X) - (t_f - t) * w_grad_c(X, t)
                + ref_v(X, t))

    def force_f(X, t):            # - v_space - lapl(v) + wind . grad(v) - zeta / beta
        return (-c(X) + 0.5 * np.pi**2 * (t_f - t) * c(X) + (t_f - t) * w_grad_c(X, t)
                - ref_zeta(X, t) / beta)

    def forward(v_old, t):
        return disc.K + disc.convection(lambda Xq: wind(Xq, t))

    ctl = Instationary(disc, forward, desired_state=desired_state, force_f=force_f, beta=beta,
                       CN=CN, n_t=n_t, initial_condition=lambda X: ref_v(X, 0.0),
                       time_interval=(0.0, t_f), bcs_v=lambda Xb, t: np.ones(len(Xb)))
    return ctl, disc, ref_v, ref_zeta


MMS_SOLVER_PARAMETERS = {"linear_solver": "fgmres", "fgmres_restart": 10,
                         "maximum_iterations": 200, "relative_tolerance": 1.0e-10,
                         "absolute_tolerance": 1.0e-10, "monitor_convergence": False}


def mms_errors(ctl, disc, ref_v, ref_zeta, n_t=10, t_f=2.0):
    """sqrt(tau) * L2 error over the time levels, as ``test_control.py:1809-1817``."""
    tau = t_f / (n_t - 1.0)
    ev = ez = 0.0
    for i in range(n_t):
        dv = ctl._v[i] - ref_v(disc.coords, i * tau)
        dz = ctl._zeta[i] - ref_zeta(disc.coords, i * tau)
        ev += dv @ (disc.M @ dv)
        ez += dz @ (disc.M @ dz)
    return np.sqrt(tau * ev), np.sqrt(tau * ez)


def _oracle_backend_stokes_pc(self, th, blocks, n_t, tau, beta, CN, lambda_v_bounds,
                              lambda_p_bounds, epsilon):
    ko = self._ko
    return ko.pc_instationary_incompressible(
        th.M_v, blocks["inner"], th.B, th.M_p, th.K_p, blocks["commutator"], n_t, tau, beta,
        th.boundary_v, ko.ChebSpec(20, *lambda_v_bounds), ko.ChebSpec(*self.schur),
        ko.ChebSpec(*self.schur), ko.ChebSpec(20, *lambda_p_bounds), CN=CN, epsilon=epsilon)


OracleBackend.construct_stokes_pc = _oracle_backend_stokes_pc
OracleBackend.ConstantNullspace = property(lambda self: self._ko.ConstantNullspace)


def stokes_exact_sol_control(CN, n=8, n_t=20, T_f=1.0, beta=1.0, taylor_hood=False):
    """``test/test_control.py:3045-3172`` (BE) / ``3175-3302`` (CN): instationary Stokes
    control with an exact solution, Q2-Q1 on ``RectangleMesh(8, 8, 2, 2, quadrilateral=True)``,
    beta = 1, time-dependent inhomogeneous Dirichlet data.  The time-convergence tests
    ``3546-3751`` (BE) / ``3965-4168`` (CN) use the same solution with ``T_f = 2``,
    ``beta = 1e-3`` on Taylor-Hood P2-P1 triangles (``taylor_hood=True``)."""
    from control_amd.control import Instationary
    from control_amd.fem import rectangle_p2p1, unit_square_q2q1
    th = rectangle_p2p1(n, n, 2.0, 2.0) if taylor_hood else unit_square_q2q1(n, 2.0)

    def true_v(X, t):
        x, y = X[:, 0] - 1.0, X[:, 1] - 1.0
        e = np.exp(T_f - t)
        return np.concatenate([e * x * y**3, 0.25 * e * (x**4 - y**4)])

    def desired_state(X, t):
        x, y = X[:, 0] - 1.0, X[:, 1] - 1.0
        e = np.exp(T_f - t)
        h0 = 4.0 * beta * y * (2.0 * (3.0 * x * x - 1.0) * (y * y - 1.0) + 3.0 * (x * x - 1.0)**2)
        h1 = -4.0 * beta * x * (3.0 * (y * y - 1.0)**2 + 2.0 * (x * x - 1.0) * (3.0 * y * y - 1.0))
        d0 = e * (x * y**3 + 2.0 * beta * y * (((x * x - 1.0)**2) * (y * y - 7.0)
                                               - 4.0 * (3.0 * x * x - 1.0) * (y * y - 1.0) + 2.0))
        d1 = e * (0.25 * (x**4 - y**4) - 2.0 * beta * x * (((y * y - 1.0)**2) * (x * x - 7.0)
                                                          - 4.0 * (x * x - 1.0) * (3.0 * y * y - 1.0)
                                                          - 2.0))
        return np.concatenate([d0 + h0, d1 + h1])

    def force_f(X, t):
        x, y = X[:, 0] - 1.0, X[:, 1] - 1.0
        e = np.exp(T_f - t)
        g0 = 2.0 * y * (x**2 - 1.0)**2 * (y**2 - 1.0)
        g1 = -2.0 * x * (x**2 - 1.0) * (y**2 - 1.0)**2
        f0 = e * (-x * y**3 - 2.0 * y * (x * x - 1.0)**2 * (y * y - 1.0))
        f1 = e * (0.25 * (y**4 - x**4) + 2.0 * x * (x * x - 1.0) * (y * y - 1.0)**2)
        return np.concatenate([f0 + g0, f1 + g1])

    ctl = Instationary(th, desired_state=desired_state, force_f=force_f, beta=beta,
                       initial_condition=lambda X: true_v(X, 0.0), time_interval=(0.0, T_f),
                       CN=CN, n_t=n_t, bcs_v=lambda Xb, t: true_v(Xb, t))
    return ctl, th, true_v


def mms_stokes_control_instationary(N, CN, n_t=10):
    """``test/test_control.py:3305-3460`` (BE) / ``3754-3905`` (CN): instationary Stokes control,
    Taylor-Hood P2-P1 on ``RectangleMesh(N, N, 2, 2)``, beta = 1e-3, polynomial exact solution
    linear in time, time-dependent inhomogeneous Dirichlet data."""
    from control_amd.control import Instationary
    from control_amd.fem import rectangle_p2p1
    th = rectangle_p2p1(N, N, 2.0, 2.0)
    t_f, beta = 2.0, 1.0e-3

    def xy(X):
        return X[:, 0] - 1.0, X[:, 1] - 1.0

    def v_space(X):
        x, y = xy(X)
        return np.concatenate([x * y**3, 0.25 * (x**4 - y**4)])

    def g(X):                      # zeta = beta (t_f - t) g
        x, y = xy(X)
        return np.concatenate([2.0 * y * (x**2 - 1.0)**2 * (y**2 - 1.0),
                               -2.0 * x * (x**2 - 1.0) * (y**2 - 1.0)**2])

    def lapl_g(X):
        x, y = xy(X)
        return np.concatenate([
            2.0 * y * (y**2 - 1.0) * (12.0 * x**2 - 4.0) + 12.0 * y * (x**2 - 1.0)**2,
            -12.0 * x * (y**2 - 1.0)**2 - 2.0 * x * (x**2 - 1.0) * (12.0 * y**2 - 4.0)])

    def true_v(X, t):
        return (t_f - t) * v_space(X)

    def true_zeta(X, t):
        return beta * (t_f - t) * g(X)

    def desired_state(X, t):       # v + zeta_space - lapl(zeta) + grad(mu), mu = beta (t_f - t) 4 x y
        x, y = xy(X)
        return (true_v(X, t) + beta * g(X) - beta * (t_f - t) * lapl_g(X)
                + beta * (t_f - t) * np.concatenate([4.0 * y, 4.0 * x]))

    def force_f(X, t):             # - v_space - lapl(v) + grad(p) - zeta / beta, -lapl(v) + grad(p) = 0
        return -v_space(X) - (t_f - t) * g(X)

    ctl = Instationary(th, desired_state=desired_state, force_f=force_f, beta=beta,
                       initial_condition=lambda X: true_v(X, 0.0), time_interval=(0.0, t_f),
                       CN=CN, n_t=n_t, bcs_v=lambda Xb, t: true_v(Xb, t))
    return ctl, th, true_v, true_zeta


def _oracle_backend_stokes_pc_stationary(self, th, D_v, D_p, beta, lambda_v_bounds,
                                         lambda_p_bounds):
    ko = self._ko
    return ko.pc_stationary_incompressible(
        th.M_v, D_v, th.B, th.M_p, th.K_p, D_p, beta, th.boundary_v,
        ko.ChebSpec(20, *lambda_v_bounds), ko.ChebSpec(*self.schur), ko.ChebSpec(*self.schur),
        ko.ChebSpec(20, *lambda_p_bounds))


OracleBackend.construct_stokes_pc_stationary = _oracle_backend_stokes_pc_stationary
