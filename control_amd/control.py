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

from .blocks import instationary_blocks

__all__ = ["Instationary", "GpuBackend"]


class GpuBackend:
    """The classes ``linear_solve`` instantiates (``preconditioner.preconditioner`` names)."""

    def __init__(self, schur=(8, 0.07, 2.1), device=0):
        from . import multiblock as mb
        self._mb, self.schur, self.device = mb, schur, device
        self.DirichletBCNullspace = mb.DirichletBCNullspace

    def MultiBlockSystem(self, *a, **kw):
        return self._mb.MultiBlockSystem(*a, device=self.device, **kw)

    def construct_pc(self, kind, M, block_01, block_10, n_t, tau, beta, nodes, lambda_v_bounds,
                     epsilon):
        mb = self._mb
        return mb.SchurPC(kind=kind, M=M, beta=beta, bc_nodes=nodes,
                          mass=mb.ChebSpec(20, *lambda_v_bounds),      # control.py:1967-1982
                          schur=mb.ChebSpec(*self.schur), n_t=n_t, tau=tau, epsilon=epsilon)


def _apply_T_1(b):            # preconditioner.py:33-45
    out = b.copy()
    out[:-1] += b[1:]
    return out


def _apply_T_2(b):            # preconditioner.py:48-60
    out = b.copy()
    out[1:] += b[:-1]
    return out


class Instationary:
    """``Control.Instationary`` (``control/control.py:1713-1836``) for a scalar state."""

    def __init__(self, disc, forward_operator=None, *, desired_state=None, force_f=None,
                 beta=1.0e-3, initial_condition=None, time_interval=(0.0, 1.0), CN=True,
                 n_t=20, bcs_v=None):
        self._disc = disc
        self._forward = forward_operator or (lambda v, t: disc.K)
        self._desired_state, self._force_f = desired_state, force_f
        self._beta = float(beta)
        self._initial_condition = initial_condition
        self._time_interval = tuple(time_interval)
        self._CN, self._n_t = bool(CN), int(n_t)
        self._bcs_v = bcs_v
        n = disc.n_dofs
        self._v = np.zeros((n_t, n))
        self._zeta = np.zeros((n_t, n))

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
            v[self._disc.boundary] = self._bcs_v(self._disc.coords[self._disc.boundary],
                                                 t_0 + i * tau)
        return v

    def set_v(self, v):              # control.py:1838-1846
        self._v = np.array(v, dtype=np.float64)
        for i in range(self._n_t):
            self._v[i, self._disc.boundary] = self._bc_values(i)[self._disc.boundary]

    def set_zeta(self, zeta):        # control.py:1848-1856
        self._zeta = np.array(zeta, dtype=np.float64)
        self._zeta[:, self._disc.boundary] = 0.0

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
        D = [sp.csr_matrix(self._forward(v_old[i], t_0 + i * tau)) for i in range(n_t)]
        b00, b01, b10, b11, m = instationary_blocks(M, D, tau, beta, n_t, CN)
        D_0 = sp.csr_matrix(self._forward(v_0, t_0))

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
