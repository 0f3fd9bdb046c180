"""``Control.Instationary.linear_solve`` without Firedrake: right-hand-side
construction, lifting of inhomogeneous Dirichlet data, the solve, and the post-solve
assembly of the time levels -- ``control/control.py:2800-3330`` on matrix-level input.

The UFL callables of the reference become array callables:

* ``forward_operator(v_i, t) -> csr``   (``forward_form``; default: the stiffness matrix)
* ``desired_state(coords, t) -> v_d``   nodal values; the driver forms ``assemble(inner(v_d,
  test) * dx) = M v_d`` as the reference does with the interpolated function
* ``force_f(coords, t) -> f``, ``initial_condition(coords) -> v_0``
* ``bcs_v(coords, t) -> values on disc.boundary`` (``None``: homogeneous)

The linear algebra runs on the GPU through ``control_amd.multiblock`` (``backend`` swaps in
another implementation of the same interface: the tests pass the CPU oracle).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

from .blocks import (conform_to, instationary_blocks, instationary_incompressible_blocks,
                     stationary_blocks, stationary_incompressible_blocks)

__all__ = ["Instationary", "Stationary", "GpuBackend", "suggest_chebyshev"]


def suggest_chebyshev(D, M, shift, nodes, safety=2.0):
    """Degree and interval ``(its, emin, emax)`` of the Jacobi-Chebyshev sweeps that stand in
    for the reference's AMG cycles on ``L = D + shift * M`` (Dirichlet rows removed).

    The interval is the spectrum of ``diag(L)^-1 L`` (both ends by plain Lanczos -- no
    factorisation, so 3-D blocks are as cheap as 2-D ones: 1.3 s for 256^2, 2.7 s for 64^3 on
    one core -- widened by 5 %); the degree is ``safety * sqrt(emax / emin)``.
    On the 256^2 P1 heat-control system (interior levels, shift tau / sqrt(beta)) this gives
    124 sweeps on [5.5e-4, 2.1]; the measured minimum for GMRES(10) to converge is 80 sweeps on
    [7e-4, 2.1], and 80 to 140 sweeps reach the solution in about the same time (DESIGN.md
    section 8).  Smaller systems prefer the larger factor (64^2 x 16: 2.25 is fastest).  Host-side
    (SciPy / ARPACK) on one spatial block."""
    import scipy.sparse.linalg as sla
    L = sp.csr_matrix(D) + float(shift) * sp.csr_matrix(M)
    keep = np.setdiff1d(np.arange(L.shape[0]), np.asarray(nodes, dtype=np.int64))
    L = L[keep][:, keep].tocsr()
    d = 1.0 / np.sqrt(L.diagonal())
    S = (sp.diags(d) @ (0.5 * (L + L.T)) @ sp.diags(d)).tocsr()   # symmetric part, Jacobi-scaled
    if S.shape[0] < 64:                                      # tiny blocks: dense
        ev = np.linalg.eigvalsh(S.toarray())
        emin, emax = float(ev[0]), float(ev[-1])
    else:
        ncv = min(60, S.shape[0] - 1)
        emax = float(sla.eigsh(S, k=1, which="LA", return_eigenvectors=False, tol=1e-3)[0])
        emin = float(sla.eigsh(S, k=1, which="SA", return_eigenvectors=False, tol=1e-2,
                               ncv=ncv, maxiter=50000)[0])
    emin, emax = 0.95 * emin, 1.05 * emax
    return int(np.ceil(safety * np.sqrt(emax / emin))), emin, emax


class GpuBackend:
    """The classes ``linear_solve`` instantiates (``preconditioner.preconditioner`` names).

    ``schur``: ``(its, emin, emax)`` of the Chebyshev sweeps that replace the reference's AMG
    sub-solves, or ``"auto"`` (default): ``suggest_chebyshev`` on the first time level's matrix."""

    def __init__(self, schur="auto", device=0, kp=None):
        from . import multiblock as mb
        self._mb, self.schur, self.device = mb, schur, device
        self.kp = kp if kp is not None else schur   # replacement of the AMG cycle on K_p
        self.DirichletBCNullspace = mb.DirichletBCNullspace
        self.ConstantNullspace = mb.ConstantNullspace

    def MultiBlockSystem(self, *a, **kw):
        return self._mb.MultiBlockSystem(*a, device=self.device, **kw)

    def construct_pc(self, kind, M, block_01, block_10, n_t, tau, beta, nodes, lambda_v_bounds,
                     epsilon):
        mb = self._mb
        schur = self.schur
        if isinstance(schur, str):
            # "auto": degree and one interval per sub-solve matrix from Lanczos estimates on the
            # device (kkt_pc_desc: schur_its = -1, schur_emin = 0); the first and last time levels
            # carry smaller shifts (control.py:2241-2327) and get their own, wider, intervals
            schur = (-1, 0.0, 0.0)
        return mb.SchurPC(kind=kind, M=M, beta=beta, bc_nodes=nodes,
                          mass=mb.ChebSpec(20, *lambda_v_bounds),      # control.py:1967-1982
                          schur=mb.ChebSpec(*schur), n_t=n_t, tau=tau, epsilon=epsilon)


    def _kp_spec(self, inner_pc):
        """Sweeps on ``K_p``: as given, or (``"auto"``) degree and lower bound of the velocity
        sub-solves on [emin, 2.1] (P1 pressure stiffness, Jacobi-scaled)."""
        if isinstance(self.kp, str):
            return (-1, 0.0, 0.0)       # kkt_pc_stokes_desc: follow the inner sub-solves
        return self.kp

    def construct_stokes_pc(self, th, blocks, n_t, tau, beta, CN, lambda_v_bounds,
                            lambda_p_bounds, epsilon):
        """``pc_fn`` of ``control.py:4318-4687`` as a ``StokesPC`` descriptor."""
        mb, m = self._mb, blocks["m"]
        nsv = mb.DirichletBCNullspace(th.boundary_v)
        inner = self.MultiBlockSystem(th.n_v, th.n_v, *blocks["inner"], n_blocks_00=m,
                                      n_blocks_11=m, nullspace_0=(nsv,) * m,
                                      nullspace_1=(nsv,) * m, CN=CN)
        comm = self.MultiBlockSystem(th.n_p, th.n_p, *blocks["commutator"], n_blocks_00=m,
                                     n_blocks_11=m)
        inner_pc = self.construct_pc("CN" if CN else "BE", th.M_v, blocks["inner"][1],
                                     blocks["inner"][2], n_t, tau, beta, th.boundary_v,
                                     lambda_v_bounds, epsilon)
        return mb.StokesPC(inner=inner, inner_pc=inner_pc, commutator=comm, B=th.B, K_p=th.K_p,
                           M_p=th.M_p, kp=mb.ChebSpec(*self._kp_spec(inner_pc)),
                           mp=mb.ChebSpec(20, *lambda_p_bounds), n_p_blocks=m, b_scale=tau,
                           post_scale=1.0 / tau**2, cn=CN)


    def construct_stokes_pc_stationary(self, th, D_v, D_p, beta, lambda_v_bounds,
                                       lambda_p_bounds):
        """``pc_fn`` of ``control.py:986-1085`` as a ``StokesPC`` descriptor."""
        mb = self._mb
        nsv = (mb.DirichletBCNullspace(th.boundary_v),)
        i00, i01, i10, i11 = stationary_blocks(th.M_v, D_v, beta)
        inner = self.MultiBlockSystem(th.n_v, th.n_v, i00, i01, i10, i11, nullspace_0=nsv,
                                      nullspace_1=nsv)
        comm = self.MultiBlockSystem(th.n_p, th.n_p, *stationary_blocks(th.M_p, D_p, beta))
        inner_pc = self.construct_pc("stationary", th.M_v, i01, i10, 1, 0.0, beta,
                                     th.boundary_v, lambda_v_bounds, 0.0)
        return mb.StokesPC(inner=inner, inner_pc=inner_pc, commutator=comm, B=th.B, K_p=th.K_p,
                           M_p=th.M_p, kp=mb.ChebSpec(*self._kp_spec(inner_pc)),
                           mp=mb.ChebSpec(20, *lambda_p_bounds))


def _apply_T_1(b):            # preconditioner.py:33-45
    out = b.copy()
    out[:-1] += b[1:]
    return out


def _apply_T_2(b):            # preconditioner.py:48-60
    out = b.copy()
    out[1:] += b[:-1]
    return out


class _VelocitySpace:
    """A Taylor-Hood discretisation seen as the space the velocity rows live on."""

    def __init__(self, th):
        nb = len(th.boundary_v) // 2
        self.M, self.K, self.n_dofs = th.M_v, th.K_v, th.n_v
        self.coords, self.boundary = th.coords_v, th.boundary_v
        self.bc_coords = th.coords_v[th.boundary_v[:nb]]


class Instationary:
    """``Control.Instationary`` (``control/control.py:1713-1836``).

    ``disc`` is a ``SpatialDiscretisation`` (scalar state, ``linear_solve``) or a
    ``TaylorHoodDiscretisation`` (velocity state, ``incompressible_linear_solve``: fields are
    component-major vectors, ``bcs_v(Xb, t)`` returns the values of both components on the
    boundary nodes ``Xb``, first component first)."""

    def __init__(self, disc, forward_operator=None, *, desired_state=None, force_f=None,
                 beta=1.0e-3, initial_condition=None, time_interval=(0.0, 1.0), CN=True,
                 n_t=20, bcs_v=None, forward_jacobian=None):
        self._th = disc if hasattr(disc, "M_v") else None
        if self._th is not None:
            disc = _VelocitySpace(disc)
        self._disc = disc
        self._forward = forward_operator or (lambda v, t: disc.K)
        self._jacobian = forward_jacobian
        self._Gauss_Newton = False
        self._desired_state, self._force_f = desired_state, force_f
        self._beta = float(beta)
        self._initial_condition = initial_condition
        self._time_interval = tuple(time_interval)
        self._CN, self._n_t = bool(CN), int(n_t)
        self._bcs_v = bcs_v
        n = disc.n_dofs
        self._v = np.zeros((n_t, n))
        self._zeta = np.zeros((n_t, n))
        if self._th is not None:
            self._p = np.zeros((n_t, self._th.n_p))
            self._mu = np.zeros((n_t, self._th.n_p))

    def set_Gauss_Newton(self, Gauss_Newton=True):   # control.py:1835-1836
        if Gauss_Newton and self._jacobian is None:
            raise ValueError("Gauss-Newton needs forward_jacobian")
        self._Gauss_Newton = bool(Gauss_Newton)

    def construct_D_v(self, v_n_help, t):    # control.py:1887-1896
        """``forward_form(trial, test, v, t)`` assembled, or after ``set_Gauss_Newton()`` its
        Gateaux derivative in ``v`` in the direction of ``trial`` (``forward_jacobian(v, t)``)."""
        return (self._jacobian if self._Gauss_Newton else self._forward)(v_n_help, t)

    # -- control.py:1898-1941
    def _times(self):
        t_0, T_f = self._time_interval
        tau = (T_f - t_0) / (self._n_t - 1.0)
        return t_0, T_f, tau

    def construct_f(self):
        t_0, _, tau = self._times()
        M, X = self._disc.M, self._disc.coords
        if self._force_f is None:
            return np.zeros((self._n_t, self._disc.n_dofs))
        return np.stack([M @ self._force_f(X, t_0 + i * tau) for i in range(self._n_t)])

    def construct_v_d(self):
        t_0, _, tau = self._times()
        M, X = self._disc.M, self._disc.coords
        return np.stack([M @ self._desired_state(X, t_0 + i * tau) for i in range(self._n_t)])

    def _bc_values(self, i):
        """``v_inhom`` of ``control.py:2995-2998``: zero except the Dirichlet values of level i."""
        t_0, _, tau = self._times()
        v = np.zeros(self._disc.n_dofs)
        if self._bcs_v is not None:
            Xb = getattr(self._disc, "bc_coords", None)
            if Xb is None:
                Xb = self._disc.coords[self._disc.boundary]
            v[self._disc.boundary] = self._bcs_v(Xb, t_0 + i * tau)
        return v

    def set_v(self, v):              # control.py:1838-1846
        self._v = np.array(v, dtype=np.float64)
        for i in range(self._n_t):
            self._v[i, self._disc.boundary] = self._bc_values(i)[self._disc.boundary]

    def set_zeta(self, zeta):        # control.py:1848-1856
        self._zeta = np.array(zeta, dtype=np.float64)
        self._zeta[:, self._disc.boundary] = 0.0

    def _velocity_rows(self, D, D_0, v_0, v_d, f, check_v_d, check_f):
        """Rows of the adjoint (``b_0``) and state (``b_1``) equations before the CN
        transforms: ``control.py:2991-3240`` (the same rows as ``3962-4245`` of the
        incompressible driver)."""
        disc, n_t, CN = self._disc, self._n_t, self._CN
        M, nodes = disc.M, disc.boundary
        _, _, tau = self._times()
        inhom = self._bcs_v is not None
        m = n_t - 1 if CN else n_t

        def bc_apply(b):             # homogenised bcs on a cofunction
            b[nodes] = 0.0
            return b

        b_0 = np.zeros((m, disc.n_dofs))
        b_1 = np.zeros((m, disc.n_dofs))
        if not CN:
            for i in range(n_t):                                     # control.py:2991-3130
                if check_v_d:
                    if i < n_t - 1:
                        b_0[i] = tau * v_d[i]
                        if inhom:
                            b_0[i] -= tau * (M @ self._bc_values(i))
                        bc_apply(b_0[i])
                else:
                    b_0[i] = v_d[i]
                if check_f:
                    if i == 0:
                        b_1[0] = tau * (D_0 @ v_0) + M @ v_0         # :3013-3017
                        if inhom:
                            vi = self._bc_values(0)
                            b_1[0] -= tau * (D_0 @ vi) + M @ vi
                    else:
                        b_1[i] = tau * f[i]
                        if inhom:
                            vi, vim = self._bc_values(i), self._bc_values(i - 1)
                            b_1[i] -= tau * (D[i] @ vi) + M @ vi
                            b_1[i] += M @ vim
                    bc_apply(b_1[i])
                else:
                    b_1[i] = f[i]
        else:
            h = 0.5 * tau
            for i in range(n_t - 1):                                 # control.py:3132-3216
                if check_v_d:
                    b_0[i] = h * (v_d[i] + v_d[i + 1])
                    if inhom:
                        b_0[i] -= h * (M @ self._bc_values(i + 1))
                        if i > 0:
                            b_0[i] -= h * (M @ self._bc_values(i))
                    bc_apply(b_0[i])
                else:
                    b_0[i] = v_d[i]
                if check_f:
                    b_1[i] = h * (f[i] + f[i + 1])
                    if inhom:
                        vi = self._bc_values(i + 1)
                        b_1[i] -= h * (D[i + 1] @ vi) + M @ vi
                        if i > 0:
                            vi = self._bc_values(i)
                            b_1[i] -= h * (D[i] @ vi) - M @ vi
                    bc_apply(b_1[i])
                else:
                    b_1[i] = f[i]
            if check_v_d:                                            # :3218-3226
                b_0[0] -= h * (M @ v_0)
                bc_apply(b_0[0])
            if check_f:                                              # :3228-3240
                b_1[0] -= h * (D_0 @ v_0) - M @ v_0
                bc_apply(b_1[0])
        return b_0, b_1

    def linear_solve(self, *, P=None, solver_parameters=None, lambda_v_bounds=None,
                     v_d=None, f=None, print_error=False, backend=None):
        """``control.py:2800-3330``.  Returns the KSP-like object of the solve; the fields are
        in ``self._v`` / ``self._zeta`` (all ``n_t`` levels, boundary values included)."""
        backend = backend or GpuBackend()
        disc, n_t, beta, CN = self._disc, self._n_t, self._beta, self._CN
        M, nodes = disc.M, disc.boundary
        t_0, T_f, tau = self._times()
        inhom = self._bcs_v is not None
        v_0 = (np.zeros(disc.n_dofs) if self._initial_condition is None
               else np.asarray(self._initial_condition(disc.coords), dtype=np.float64))
        check_f, check_v_d = f is None, v_d is None
        if check_f:
            f = self.construct_f()
        if check_v_d:
            v_d = self.construct_v_d()
        v_old = self._v
        D = [conform_to(self.construct_D_v(v_old[i], t_0 + i * tau), M) for i in range(n_t)]
        b00, b01, b10, b11, m = instationary_blocks(M, D, tau, beta, n_t, CN)
        D_0 = conform_to(self.construct_D_v(v_0, t_0), M)

        b_0, b_1 = self._velocity_rows(D, D_0, v_0, v_d, f, check_v_d, check_f)
        if CN:
            b_0 = _apply_T_1(b_0)                                    # :3242-3243
            b_1 = _apply_T_2(b_1)

        if P is None:                                                # :3245-3258
            pc_fn = backend.construct_pc("CN" if CN else "BE", M, b01, b10, n_t, tau, beta,
                                         nodes, lambda_v_bounds or (0.5, 2.0), 1.0e-3)
        else:
            pc_fn = P
        if solver_parameters is None:                                # :3260-3266
            solver_parameters = {"linear_solver": "gmres", "gmres_restart": 10,
                                 "maximum_iterations": 50, "relative_tolerance": 1.0e-6,
                                 "absolute_tolerance": 0.0, "monitor_convergence": print_error}
        ns = tuple(backend.DirichletBCNullspace(nodes) for _ in range(m))
        system = backend.MultiBlockSystem(disc.n_dofs, disc.n_dofs, b00, b01, b10, b11,
                                          n_blocks_00=m, n_blocks_11=m, nullspace_0=ns,
                                          nullspace_1=ns, CN=CN)
        v = np.zeros((m, disc.n_dofs))
        zeta = np.zeros((m, disc.n_dofs))
        ksp = system.solve(v, zeta, b_0, b_1, solver_parameters=solver_parameters, pc_fn=pc_fn)
        if CN:                                                       # :3300-3312
            v_new = np.zeros((n_t, disc.n_dofs))
            zeta_new = np.zeros((n_t, disc.n_dofs))
            if check_f and check_v_d:
                v_new[0] = v_0
            v_new[1:] = v
            zeta_new[:-1] = zeta
            self.set_v(v_new)
            self.set_zeta(zeta_new)
        else:
            self.set_v(v)
            self.set_zeta(zeta)
        return ksp

    def non_linear_res_eval(self, v_old, zeta_old, v_0, v_d, f):
        """``control.py:2442-2810``: residual rows of the (Picard-linearised) optimality system
        at ``(v_old, zeta_old)``, all ``n_t`` levels given; BE: ``n_t`` rows, CN: ``n_t - 1``."""
        disc, n_t, beta, CN = self._disc, self._n_t, self._beta, self._CN
        M, nodes = disc.M, disc.boundary
        t_0, _, tau = self._times()
        D = [sp.csr_matrix(self.construct_D_v(v_old[i], t_0 + i * tau)) for i in range(n_t)]
        if CN:
            m, h = n_t - 1, 0.5 * tau
            r0 = np.zeros((m, disc.n_dofs))
            r1 = np.zeros((m, disc.n_dofs))
            for i in range(m):
                r0[i] = (h * (v_d[i] + v_d[i + 1]) - h * (M @ (v_old[i] + v_old[i + 1]))
                         - (h * (D[i].T @ zeta_old[i]) + M @ zeta_old[i])
                         - (h * (D[i + 1].T @ zeta_old[i + 1]) - M @ zeta_old[i + 1]))
                r1[i] = (h * (f[i] + f[i + 1])
                         - (h * (D[i] @ v_old[i]) - M @ v_old[i])
                         - (h * (D[i + 1] @ v_old[i + 1]) + M @ v_old[i + 1])
                         + (h / beta) * (M @ (zeta_old[i] + zeta_old[i + 1])))
        else:
            r0 = np.zeros((n_t, disc.n_dofs))
            r1 = np.zeros((n_t, disc.n_dofs))
            D_0 = sp.csr_matrix(self.construct_D_v(v_0, t_0))
            for i in range(n_t):
                Dz = tau * (D[i].T @ zeta_old[i]) + M @ zeta_old[i]
                r0[i] = (tau * v_d[i] - tau * (M @ v_old[i]) - Dz + M @ zeta_old[i + 1]
                         if i < n_t - 1 else -Dz)
                Dv = tau * (D[i] @ v_old[i]) + M @ v_old[i]
                r1[i] = (tau * (D_0 @ v_0) + M @ v_0 - Dv if i == 0 else
                         tau * f[i] + M @ v_old[i - 1] - Dv + (tau / beta) * (M @ zeta_old[i]))
        r0[:, nodes] = 0.0
        r1[:, nodes] = 0.0
        return r0, r1

    def non_linear_solve(self, *, P=None, solver_parameters=None, lambda_v_bounds=None,
                         max_non_linear_iter=10, relative_non_linear_tol=1.0e-5,
                         absolute_non_linear_tol=1.0e-8, print_error_non_linear=False,
                         backend=None):
        """``control.py:3377-3560``: Picard loop around ``linear_solve``; returns the residual
        norms (initial one first)."""
        disc, n_t, CN = self._disc, self._n_t, self._CN
        v_0 = (np.zeros(disc.n_dofs) if self._initial_condition is None
               else np.asarray(self._initial_condition(disc.coords), dtype=np.float64))
        v_old, zeta_old = self._v.copy(), self._zeta.copy()
        if CN:
            v_old[0] = v_0                                           # control.py:3425-3426
        zeta_old[n_t - 1] = 0.0
        f, v_d = self.construct_f(), self.construct_v_d()

        def evaluate():
            r0, r1 = self.non_linear_res_eval(v_old, zeta_old, v_0, v_d, f)
            return r0, r1, float(np.sqrt(np.vdot(r0, r0) + np.vdot(r1, r1)))
        rhs_0, rhs_1, norm_0 = evaluate()
        norm_k, k, norms = norm_0, 0, [norm_0]
        while norm_k > relative_non_linear_tol * norm_0 and norm_k > absolute_non_linear_tol:
            self._v = v_old          # linear_solve linearises at self._v (control.py:2886)
            self.linear_solve(P=P, solver_parameters=solver_parameters,
                              lambda_v_bounds=lambda_v_bounds, v_d=rhs_0, f=rhs_1,
                              backend=backend)
            v_old = v_old + self._v
            for i in range(n_t):                                     # :3490-3493
                v_old[i, disc.boundary] = self._bc_values(i)[disc.boundary]
            if CN:
                v_old[0] = v_0
                v_old[0, disc.boundary] = self._bc_values(0)[disc.boundary]
            zeta_old = zeta_old + self._zeta
            zeta_old[:, disc.boundary] = 0.0
            self.set_v(v_old)
            self.set_zeta(zeta_old)
            rhs_0, rhs_1, norm_k = evaluate()
            norms.append(norm_k)
            k += 1
            if print_error_non_linear:
                print(f"Non-linear solver: iteration {k:d}, non-linear residual norm "
                      f"{norm_k:.16e}")
            if k + 1 > max_non_linear_iter:
                break
        return norms

    def set_p(self, p):              # control.py:1858-1865
        self._p = np.array(p, dtype=np.float64)

    def set_mu(self, mu):
        self._mu = np.array(mu, dtype=np.float64)

    def incompressible_linear_solve(self, nullspace_p=None, *, forward_operator_p=None, P=None,
                                    solver_parameters=None, lambda_v_bounds=None,
                                    lambda_p_bounds=None, v_d=None, f=None, div_v=None,
                                    div_zeta=None, print_error=False, backend=None):
        """``control.py:3592-4760``: the Stokes-type control solve.  ``nullspace_p`` is the
        nullspace class instance put on every pressure block (``ConstantNullspace()`` for
        enclosed flow, ``test/test_control.py:3167``); ``forward_operator_p(v_i, t)`` is the
        forward form on the pressure space (default ``K_p``, ``control.py:3783-3785``).
        Fields afterwards: ``_v``, ``_zeta`` (``n_t`` levels), ``_p``, ``_mu`` (``m`` levels)."""
        backend = backend or GpuBackend()
        th = self._th
        if th is None:
            raise ValueError("Undefined space_p")                    # control.py:3604-3608
        disc, n_t, beta, CN = self._disc, self._n_t, self._beta, self._CN
        nodes = disc.boundary
        t_0, T_f, tau = self._times()
        inhom = self._bcs_v is not None
        fwd_p = forward_operator_p or (lambda v, t: th.K_p)
        v_0 = (np.zeros(disc.n_dofs) if self._initial_condition is None
               else np.asarray(self._initial_condition(disc.coords), dtype=np.float64))
        check_f, check_v_d = f is None, v_d is None
        if check_f:
            f = self.construct_f()
        if check_v_d:
            v_d = self.construct_v_d()
        v_old = self._v
        D = [conform_to(self.construct_D_v(v_old[i], t_0 + i * tau), th.M_v) for i in range(n_t)]
        Dp = [conform_to(fwd_p(v_old[i], t_0 + i * tau), th.M_p) for i in range(n_t)]
        D_0 = conform_to(self.construct_D_v(v_0, t_0), th.M_v)
        bl = instationary_incompressible_blocks(th.M_v, D, th.B, th.M_p, Dp, tau, beta, n_t, CN)
        m = bl["m"]
        b_0_0, b_0_1 = self._velocity_rows(D, D_0, v_0, v_d, f, check_v_d, check_f)
        b_1_0 = np.zeros((m, th.n_p))
        b_1_1 = np.zeros((m, th.n_p))
        if div_v is None:                                            # :4088-4100, :4247-4258
            if inhom:
                for i in range(m):
                    b_1_0[i] -= tau * (th.B @ self._bc_values(i + 1 if CN else i))
        else:
            b_1_0[:] = div_v
        if div_zeta is not None:
            b_1_1[:] = div_zeta
        if CN:                                                       # :4266-4269
            b_0_0, b_0_1 = _apply_T_1(b_0_0), _apply_T_2(b_0_1)
            b_1_0, b_1_1 = _apply_T_2(b_1_0), _apply_T_1(b_1_1)
        b_0 = np.concatenate([b_0_0, b_0_1])
        b_1 = np.concatenate([b_1_0, b_1_1])

        if solver_parameters is None:                                # :4291-4297
            solver_parameters = {"linear_solver": "fgmres", "fgmres_restart": 10,
                                 "maximum_iterations": 100, "relative_tolerance": 1.0e-6,
                                 "absolute_tolerance": 0.0, "monitor_convergence": print_error}
        if P is None:
            pc_fn = backend.construct_stokes_pc(th, bl, n_t, tau, beta, CN,
                                                lambda_v_bounds or (0.3924, 2.0598),
                                                lambda_p_bounds or (0.5, 2.0), 1.0e-3)
        else:
            pc_fn = P
        nsv = tuple(backend.DirichletBCNullspace(nodes) for _ in range(2 * m))
        nsp = tuple((nullspace_p.__class__() if nullspace_p is not None
                     else backend.ConstantNullspace()) for _ in range(2 * m))
        kw = dict(sub_n_blocks_00_0=m, sub_n_blocks_11_0=m) if CN else {}
        system = backend.MultiBlockSystem(th.n_v, th.n_p, *bl["outer"], n_blocks_00=2 * m,
                                          n_blocks_11=2 * m, nullspace_0=nsv, nullspace_1=nsp,
                                          CN=CN, **kw)                # :4274-4289
        u_0 = np.zeros((2 * m, th.n_v))
        u_1 = np.zeros((2 * m, th.n_p))
        ksp = system.solve(u_0, u_1, b_0, b_1, solver_parameters=solver_parameters, pc_fn=pc_fn)
        v = np.zeros((n_t, th.n_v))                                  # :4698-4726
        zeta = np.zeros((n_t, th.n_v))
        if CN:
            if check_f and check_v_d:
                v[0] = v_0
            v[1:] = u_0[:m]
            zeta[:m] = u_0[m:]
        else:
            v[:] = u_0[:m]
            zeta[:] = u_0[m:]
        self.set_v(v)
        self.set_zeta(zeta)
        self.set_mu(u_1[:m])
        self.set_p(u_1[m:])
        return ksp


class Stationary:
    """``Control.Stationary`` (``control/control.py:100-800``) for a scalar state:
    ``linear_solve`` (``:487-604``) and the Picard / Gauss-Newton loop ``non_linear_solve``
    (``:640-760``).

    ``forward_operator(v_old) -> csr`` is ``forward_form(trial, test, v_old)`` assembled;
    ``forward_jacobian(v_old) -> csr`` its Gateaux derivative in the direction of ``trial``
    (``ufl.derivative``, ``control.py:314-318``), used after ``set_Gauss_Newton()``."""

    def __init__(self, disc, forward_operator=None, *, desired_state=None, force_f=None,
                 beta=1.0e-3, bcs_v=None, forward_jacobian=None):
        self._th = disc if hasattr(disc, "M_v") else None
        if self._th is not None:
            disc = _VelocitySpace(disc)
        self._disc = disc
        self._forward = forward_operator or (lambda v: disc.K)
        self._jacobian = forward_jacobian
        self._Gauss_Newton = False
        self._desired_state, self._force_f = desired_state, force_f
        self._beta = float(beta)
        self._bcs_v = bcs_v
        self._v = np.zeros(disc.n_dofs)
        self._zeta = np.zeros(disc.n_dofs)

    def set_Gauss_Newton(self):
        if self._jacobian is None:
            raise ValueError("Gauss-Newton needs forward_jacobian")
        self._Gauss_Newton = True

    def set_Picard(self):
        self._Gauss_Newton = False

    def construct_D_v(self, v_old):          # control.py:310-320
        # on the mass matrix's structure: the preconditioner adds multiples of M entry by entry
        return conform_to(self._jacobian(v_old) if self._Gauss_Newton
                          else self._forward(v_old), self._disc.M)

    def _v_inhom(self):
        v = np.zeros(self._disc.n_dofs)
        if self._bcs_v is not None:
            Xb = getattr(self._disc, "bc_coords", None)
            if Xb is None:
                Xb = self._disc.coords[self._disc.boundary]
            v[self._disc.boundary] = self._bcs_v(Xb)
        return v

    def set_v(self, v):                      # control.py:264-272
        self._v = np.array(v, dtype=np.float64)
        self._v[self._disc.boundary] = self._v_inhom()[self._disc.boundary]

    def set_zeta(self, zeta):                # control.py:274-283
        self._zeta = np.array(zeta, dtype=np.float64)
        self._zeta[self._disc.boundary] = 0.0

    def _data(self):
        M, X = self._disc.M, self._disc.coords
        f = ((M @ self._force_f(X)) if self._force_f is not None
             else np.zeros(self._disc.n_dofs))
        return M @ self._desired_state(X), f

    def linear_solve(self, *, P=None, solver_parameters=None, lambda_v_bounds=None, v_d=None,
                     f=None, print_error=False, backend=None):
        backend = backend or GpuBackend()
        disc, beta = self._disc, self._beta
        M, nodes = disc.M, disc.boundary
        inhom = self._bcs_v is not None
        D_v = self.construct_D_v(self._v)
        v_inhom = self._v_inhom()
        if f is None or v_d is None:
            v_d_data, f_data = self._data()
        if f is None:                        # construct_f, control.py:322-332
            f = f_data - (D_v @ v_inhom if inhom else 0.0)
            f[nodes] = 0.0 if inhom else f[nodes]
        if v_d is None:                      # construct_v_d, control.py:334-346
            v_d = v_d_data - (M @ v_inhom if inhom else 0.0)
            v_d[nodes] = 0.0 if inhom else v_d[nodes]
        b00, b01, b10, b11 = stationary_blocks(M, D_v, beta)
        if P is None:
            pc_fn = backend.construct_pc("stationary", M, b01, b10, 1, 0.0, beta, nodes,
                                         lambda_v_bounds or (0.5, 2.0), 0.0)
        else:
            pc_fn = P
        if solver_parameters is None:        # control.py:556-562
            solver_parameters = {"linear_solver": "gmres", "gmres_restart": 10,
                                 "maximum_iterations": 50, "relative_tolerance": 1.0e-6,
                                 "absolute_tolerance": 0.0, "monitor_convergence": print_error}
        ns = (backend.DirichletBCNullspace(nodes),)
        system = backend.MultiBlockSystem(disc.n_dofs, disc.n_dofs, b00, b01, b10, b11,
                                          nullspace_0=ns, nullspace_1=ns)
        v = np.zeros((1, disc.n_dofs))
        zeta = np.zeros((1, disc.n_dofs))
        ksp = system.solve(v, zeta, np.atleast_2d(v_d), np.atleast_2d(f),
                           solver_parameters=solver_parameters, pc_fn=pc_fn)
        self.set_v(v[0] + (v_inhom if inhom else 0.0))          # control.py:572-577
        self.set_zeta(zeta[0])
        return ksp

    def non_linear_res_eval(self, v_d, f, v_old, zeta_old, D_v):   # control.py:452-486
        M, nodes, beta = self._disc.M, self._disc.boundary, self._beta
        rhs_0 = v_d - M @ v_old - D_v.T @ zeta_old
        rhs_1 = f - D_v @ v_old + (1.0 / beta) * (M @ zeta_old)
        rhs_0[nodes] = 0.0
        rhs_1[nodes] = 0.0
        return rhs_0, rhs_1

    def non_linear_solve(self, *, P=None, solver_parameters=None, lambda_v_bounds=None,
                         max_non_linear_iter=10, relative_non_linear_tol=1.0e-5,
                         absolute_non_linear_tol=1.0e-8, print_error_non_linear=False,
                         backend=None):
        """``control.py:640-760``; returns the residual norms (initial one first)."""
        v_old, zeta_old = self._v.copy(), self._zeta.copy()
        v_d, f = self._data()
        D_v = self.construct_D_v(v_old)
        rhs_0, rhs_1 = self.non_linear_res_eval(v_d, f, v_old, zeta_old, D_v)
        norm_0 = float(np.sqrt(rhs_0 @ rhs_0 + rhs_1 @ rhs_1))
        norm_k, k, norms = norm_0, 0, [norm_0]
        while norm_k > relative_non_linear_tol * norm_0 and norm_k > absolute_non_linear_tol:
            self.linear_solve(P=P, solver_parameters=solver_parameters,
                              lambda_v_bounds=lambda_v_bounds, v_d=rhs_0, f=rhs_1,
                              backend=backend)
            v_old = v_old + self._v
            if self._bcs_v is not None:
                v_old[self._disc.boundary] = self._v_inhom()[self._disc.boundary]
            self.set_v(v_old)
            zeta_old = zeta_old + self._zeta
            self.set_zeta(zeta_old)
            zeta_old = self._zeta.copy()
            D_v = self.construct_D_v(v_old)
            rhs_0, rhs_1 = self.non_linear_res_eval(v_d, f, v_old, zeta_old, D_v)
            norm_k = float(np.sqrt(rhs_0 @ rhs_0 + rhs_1 @ rhs_1))
            norms.append(norm_k)
            k += 1
            if print_error_non_linear:
                print(f"Non-linear solver: iteration {k:d}, non-linear residual norm "
                      f"{norm_k:.16e}")
            if k + 1 > max_non_linear_iter:
                break
        return norms

    def incompressible_linear_solve(self, nullspace_p=None, *, forward_operator_p=None, P=None,
                                    solver_parameters=None, lambda_v_bounds=None,
                                    lambda_p_bounds=None, v_d=None, f=None, div_v=None,
                                    div_zeta=None, print_error=False, backend=None):
        """``control.py:802-1110``: stationary Stokes-type control.  Fields afterwards: ``_v``,
        ``_zeta`` (velocity space, component-major), ``_p``, ``_mu``."""
        backend = backend or GpuBackend()
        th = self._th
        if th is None:
            raise ValueError("Undefined space_p")                    # control.py:813-817
        disc, beta, nodes = self._disc, self._beta, self._disc.boundary
        M = disc.M
        inhom = self._bcs_v is not None
        D_v = self.construct_D_v(self._v)
        D_p = conform_to(forward_operator_p(self._v) if forward_operator_p else th.K_p, th.M_p)
        v_inhom = self._v_inhom()
        if f is None or v_d is None:
            v_d_data, f_data = self._data()
        if f is None:
            f = f_data - (D_v @ v_inhom if inhom else 0.0)
            if inhom:
                f[nodes] = 0.0
        if v_d is None:
            v_d = v_d_data - (M @ v_inhom if inhom else 0.0)
            if inhom:
                v_d[nodes] = 0.0
        if div_v is None:                                            # :866-871
            div_v = -(th.B @ v_inhom) if inhom else np.zeros(th.n_p)
        if div_zeta is None:
            div_zeta = np.zeros(th.n_p)
        b_0 = np.stack([v_d, f])
        b_1 = np.stack([div_v, div_zeta])
        blocks = stationary_incompressible_blocks(th.M_v, D_v, th.B, beta)
        if P is None:
            pc_fn = backend.construct_stokes_pc_stationary(
                th, D_v, D_p, beta, lambda_v_bounds or (0.3924, 2.0598),
                lambda_p_bounds or (0.5, 2.0))
        else:
            pc_fn = P
        if solver_parameters is None:                                # :921-927
            solver_parameters = {"linear_solver": "fgmres", "fgmres_restart": 10,
                                 "maximum_iterations": 100, "relative_tolerance": 1.0e-6,
                                 "absolute_tolerance": 0.0, "monitor_convergence": print_error}
        nsv = tuple(backend.DirichletBCNullspace(nodes) for _ in range(2))
        nsp = tuple((nullspace_p.__class__() if nullspace_p is not None
                     else backend.ConstantNullspace()) for _ in range(2))
        system = backend.MultiBlockSystem(th.n_v, th.n_p, *blocks, n_blocks_00=2, n_blocks_11=2,
                                          nullspace_0=nsv, nullspace_1=nsp)
        u_0 = np.zeros((2, th.n_v))
        u_1 = np.zeros((2, th.n_p))
        ksp = system.solve(u_0, u_1, b_0, b_1, solver_parameters=solver_parameters, pc_fn=pc_fn)
        self.set_v(u_0[0] + (v_inhom if inhom else 0.0))            # :1092-1100
        self.set_zeta(u_0[1])
        self._mu, self._p = u_1[0].copy(), u_1[1].copy()
        return ksp

    def incompressible_non_linear_solve(self, nullspace_p=None, *, forward_operator_p=None,
                                        P=None, solver_parameters=None, lambda_v_bounds=None,
                                        lambda_p_bounds=None, max_non_linear_iter=10,
                                        relative_non_linear_tol=1.0e-5,
                                        absolute_non_linear_tol=1.0e-8,
                                        print_error_non_linear=False, backend=None):
        """``control.py:1112-1480``: Picard loop of stationary Navier-Stokes-type control around
        ``incompressible_linear_solve``; returns the residual norms (initial one first)."""
        th = self._th
        if th is None:
            raise ValueError("Undefined space_p")
        disc, beta, nodes = self._disc, self._beta, self._disc.boundary
        M, B = disc.M, th.B
        BT = sp.csr_matrix(B.T)
        v_old, zeta_old = self._v.copy(), self._zeta.copy()
        p_old = getattr(self, "_p", np.zeros(th.n_p)).copy()
        mu_old = getattr(self, "_mu", np.zeros(th.n_p)).copy()
        v_d, f = self._data()

        def evaluate():                                              # :1272-1310
            D_v = self.construct_D_v(v_old)
            r00 = v_d - M @ v_old - D_v.T @ zeta_old - BT @ mu_old
            r01 = f - D_v @ v_old + (1.0 / beta) * (M @ zeta_old) - BT @ p_old
            r00[nodes] = 0.0
            r01[nodes] = 0.0
            r10, r11 = -(B @ v_old), -(B @ zeta_old)
            n = np.sqrt(r00 @ r00 + r01 @ r01 + r10 @ r10 + r11 @ r11)
            return r00, r01, r10, r11, float(n)
        r00, r01, r10, r11, norm_0 = evaluate()
        norm_k, k, norms = norm_0, 0, [norm_0]
        while norm_k > relative_non_linear_tol * norm_0 and norm_k > absolute_non_linear_tol:
            self._v = v_old          # linearisation point (control.py:839-840)
            self.incompressible_linear_solve(
                nullspace_p, forward_operator_p=forward_operator_p, P=P,
                solver_parameters=solver_parameters, lambda_v_bounds=lambda_v_bounds,
                lambda_p_bounds=lambda_p_bounds, v_d=r00, f=r01, div_v=r10, div_zeta=r11,
                backend=backend)
            v_old = v_old + self._v
            if self._bcs_v is not None:
                v_old[nodes] = self._v_inhom()[nodes]
            zeta_old = zeta_old + self._zeta
            zeta_old[nodes] = 0.0
            p_old = p_old + self._p
            mu_old = mu_old + self._mu
            self.set_v(v_old)
            self.set_zeta(zeta_old)
            self._p, self._mu = p_old.copy(), mu_old.copy()
            r00, r01, r10, r11, norm_k = evaluate()
            norms.append(norm_k)
            k += 1
            if print_error_non_linear:
                print(f"Non-linear solver: iteration {k:d}, non-linear residual norm "
                      f"{norm_k:.16e}")
            if k + 1 > max_non_linear_iter:
                break
        return norms
