"""Picard outer loop of instationary Navier-Stokes control (BE and CN).

Host-side mirror of ``Control.Instationary.incompressible_non_linear_solve``
(``control/control.py:4886-5232``): evaluate the non-linear residual at the current iterate
(``:2442-2620`` for the velocity rows, ``:4976-5082`` for the pressure couplings), solve the
linearised Stokes-control system for the update (``incompressible_linear_solve`` with the
convection term frozen at ``v_old``, ``construct_D_v`` ``:1887-1896``), add it, repeat until
``||r_k|| <= max(rtol ||r_0||, atol)`` or ``max_non_linear_iter``.

The loop, the residual and the re-assembly of the convection blocks stay on the host, as in
the reference; each outer iteration re-uploads only the values of the blocks that changed
(``kkt_update_block_values``: same sparsity structure, no index traffic, preconditioner
matrices refreshed on the device) and runs the linear solve on the GPU through
``MultiBlockSystem.solve`` with a ``StokesPC``.

Dirichlet velocity conditions: the updates vanish on the boundary (``bcs_v`` homogenised,
``bcs_zeta``); inhomogeneous, time-dependent values ride on the initial iterate ``v`` the
caller passes (``control.py:4925-4959`` applies the conditions to ``v_old`` before the loop),
as in the lid-driven cavity of ``test/test_control.py:4171-4268``.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np
import scipy.sparse as sp

from .blocks import instationary_incompressible_blocks

__all__ = ["NavierStokesControl", "GpuLinearSolver", "incompressible_non_linear_solve",
           "non_linear_res_eval"]


@dataclass
class NavierStokesControl:
    """What ``Control.Instationary`` holds for this driver: the Taylor-Hood discretisation,
    ``forward_form = nu (grad u, grad v) + ((w . grad) u, v)`` (``test_control.py:4194-4199``),
    nodal desired states ``v_d[i]`` and forces ``f[i]`` per time level, the initial
    condition, ``beta`` and the time interval."""
    disc: object
    nu: float
    beta: float
    n_t: int
    T: float
    v_d: np.ndarray                      # (n_t, n_v)
    f: np.ndarray                        # (n_t, n_v)
    v_0: np.ndarray = None               # (n_v,), zero when omitted
    t_0: float = 0.0
    CN: bool = False                     # Crank-Nicolson instead of backward Euler

    @property
    def tau(self):
        return (self.T - self.t_0) / (self.n_t - 1.0)

    def D_v(self, w):
        """``construct_D_v(v_trial, v_test, w, t)`` (Picard, ``control.py:1888-1889``)."""
        return _same_structure_sum(self.disc.K_v, [(self.nu, self.disc.K_v),
                                                   (1.0, self.disc.convection_v(w))])

    def D_p(self, w):
        """The same form on the pressure space (``control.py:3783-3785``)."""
        return _same_structure_sum(self.disc.K_p, [(self.nu, self.disc.K_p),
                                                   (1.0, self.disc.convection_p(w))])


def _apply_T_1(b):            # preconditioner.py:33-45
    out = b.copy()
    out[:-1] += b[1:]
    return out


def _apply_T_2(b):            # preconditioner.py:48-60
    out = b.copy()
    out[1:] += b[:-1]
    return out


def _same_structure_sum(like, terms):
    """``sum c_k A_k`` on the structure of ``like`` (explicit zeros kept, so the result can
    replace a stored block's values entry by entry)."""
    data = np.zeros_like(like.data)
    for c, A in terms:
        if A.nnz != like.nnz or not (np.array_equal(A.indptr, like.indptr)
                                     and np.array_equal(A.indices, like.indices)):
            raise ValueError("matrices assembled over different connectivities")
        data += c * A.data
    return sp.csr_matrix((data, like.indices.copy(), like.indptr.copy()), shape=like.shape)


def non_linear_res_eval(pb: NavierStokesControl, D, v, zeta, p, mu):
    """Residual rows of the BE Navier-Stokes control system at ``(v, zeta, p, mu)``
    (``control.py:2456-2620`` + ``5003-5041``); ``D[i] = D_v(v[i])``.  Returns
    ``(rhs_00, rhs_01, rhs_10, rhs_11)`` with Dirichlet rows zeroed (``bc.apply``)."""
    th, n_t, tau, beta = pb.disc, pb.n_t, pb.tau, pb.beta
    M, B = th.M_v, th.B
    BT = sp.csr_matrix(B.T)
    v_0 = np.zeros(th.n_v) if pb.v_0 is None else pb.v_0
    if pb.CN:
        # control.py:2621-2808 + 5043-5082: row i couples the levels i and i + 1; v[0] is the
        # initial condition, zeta[n_t - 1] = 0; mu, p live on the n_t - 1 intervals
        m, h = n_t - 1, 0.5 * tau
        r00 = np.zeros((m, th.n_v))
        r01 = np.zeros((m, th.n_v))
        for i in range(m):
            r00[i] = (h * (M @ (pb.v_d[i] + pb.v_d[i + 1])) - h * (M @ (v[i] + v[i + 1]))
                      - (h * (D[i].T @ zeta[i]) + M @ zeta[i])
                      - (h * (D[i + 1].T @ zeta[i + 1]) - M @ zeta[i + 1])
                      - tau * (BT @ mu[i]))
            r01[i] = (h * (M @ (pb.f[i] + pb.f[i + 1]))
                      - (h * (D[i] @ v[i]) - M @ v[i])
                      - (h * (D[i + 1] @ v[i + 1]) + M @ v[i + 1])
                      + (h / beta) * (M @ (zeta[i] + zeta[i + 1]))
                      - tau * (BT @ p[i]))
        r00[:, th.boundary_v] = 0.0
        r01[:, th.boundary_v] = 0.0
        r10 = np.stack([-(B @ v[i + 1]) for i in range(m)])
        r11 = np.stack([-(B @ zeta[i]) for i in range(m)])
        return r00, r01, r10, r11
    r00 = np.zeros((n_t, th.n_v))
    r01 = np.zeros((n_t, th.n_v))
    for i in range(n_t):
        Dz = tau * (D[i].T @ zeta[i]) + M @ zeta[i]
        if i < n_t - 1:          # :2469-2497, :2552-2584
            r00[i] = tau * (M @ pb.v_d[i]) - tau * (M @ v[i]) - Dz + M @ zeta[i + 1]
        else:                    # :2541-2546
            r00[i] = -Dz
        Dv = tau * (D[i] @ v[i]) + M @ v[i]
        if i == 0:               # :2499-2512: the initial-condition row
            D0 = pb.D_v(v_0)
            r01[0] = tau * (D0 @ v_0) + M @ v_0 - Dv
        else:                    # :2518-2540, :2586-2615
            r01[i] = (tau * (M @ pb.f[i]) + M @ v[i - 1] - Dv
                      + (tau / beta) * (M @ zeta[i]))
        r00[i] -= tau * (BT @ mu[i])        # :5004-5012
        r01[i] -= tau * (BT @ p[i])         # :5017-5025
    r00[:, th.boundary_v] = 0.0
    r01[:, th.boundary_v] = 0.0
    r10 = np.stack([-(B @ v[i]) for i in range(n_t)])       # :5030-5034
    r11 = np.stack([-(B @ zeta[i]) for i in range(n_t)])    # :5036-5041
    return r00, r01, r10, r11


class GpuLinearSolver:
    """The linearised solve of one Picard iteration on the GPU: builds the outer, inner
    (velocity KKT) and commutator (pressure) systems once, afterwards only re-uploads the
    values of the blocks that carry the re-linearised operator."""

    def __init__(self, pb: NavierStokesControl, *, mass, schur, kp, mp, solver_parameters,
                 device=0, comm=None, host_allreduce=None, options=None):
        """``comm`` (``control_amd.dist``): the three systems are time-sharded (BASELINE
        configs[4] names 8 GPUs) -- every rank runs the same Picard loop on the whole iterate
        (residual and re-linearisation are host work on replicated data, as cheap as in the
        reference), uploads the blocks of its own levels, solves for its shard of the update,
        and the shards are summed into the whole update with ``host_allreduce(array, op)``
        (in place over ranks, op 0 = sum: e.g. ``GlooTransport.allreduce``)."""
        self.pb, self.device = pb, device
        self.options = options          # execution options of the three systems (kkt_set_option)
        self.specs = dict(mass=mass, schur=schur, kp=kp, mp=mp)
        self.solver_parameters = solver_parameters
        self.outer = None
        self.uploads = 0
        self.dist = comm if comm is not None and comm.world > 1 else None
        self.host_allreduce = host_allreduce
        if self.dist is not None and host_allreduce is None:
            raise ValueError("a time-sharded GpuLinearSolver needs host_allreduce")

    def _blocks(self, D, Dp):
        th, pb = self.pb.disc, self.pb
        return instationary_incompressible_blocks(th.M_v, list(D), th.B, th.M_p, list(Dp),
                                                  pb.tau, pb.beta, pb.n_t, pb.CN)

    def _build(self, bl):
        from .multiblock import (ChebSpec, ConstantNullspace, DirichletBCNullspace,
                                 MultiBlockSystem, SchurPC, StokesPC)
        th, pb, m = self.pb.disc, self.pb, bl["m"]
        nsv = DirichletBCNullspace(th.boundary_v)
        kw = dict(sub_n_blocks_00_0=m, sub_n_blocks_11_0=m) if pb.CN else {}
        self.outer = MultiBlockSystem(
            th.n_v, th.n_p, *bl["outer"], n_blocks_00=2 * m, n_blocks_11=2 * m,
            nullspace_0=(nsv,) * (2 * m),
            nullspace_1=tuple(ConstantNullspace() for _ in range(2 * m)), device=self.device,
            CN=pb.CN, comm=self.dist, shard_families=2, options=self.options, **kw)
        self.inner = MultiBlockSystem(th.n_v, th.n_v, *bl["inner"], n_blocks_00=m,
                                      n_blocks_11=m, nullspace_0=(nsv,) * m,
                                      nullspace_1=(nsv,) * m, device=self.device, CN=pb.CN,
                                      comm=self.dist, options=self.options)
        if getattr(th, "coords_v", None) is not None and 2 * len(th.coords_v) == th.n_v:
            self.inner.set_tile_coordinates(np.vstack([th.coords_v, th.coords_v]))
        self.comm = MultiBlockSystem(th.n_p, th.n_p, *bl["commutator"], n_blocks_00=m,
                                     n_blocks_11=m, device=self.device, comm=self.dist,
                                     options=self.options)
        s = self.specs
        inner_pc = SchurPC(kind="CN" if pb.CN else "BE", M=th.M_v, beta=pb.beta, bc_nodes=th.boundary_v,
                           mass=ChebSpec(*s["mass"]), schur=ChebSpec(*s["schur"]), n_t=pb.n_t,
                           tau=pb.tau)
        self.pc = StokesPC(inner=self.inner, inner_pc=inner_pc, commutator=self.comm, B=th.B,
                           K_p=th.K_p, M_p=th.M_p, kp=ChebSpec(*s["kp"]), mp=ChebSpec(*s["mp"]),
                           n_p_blocks=m, b_scale=pb.tau, post_scale=1.0 / pb.tau**2, cn=pb.CN)

    def _update(self, bl):
        """Blocks that carry the linearised operator: all of ``block_01_int`` / ``block_10_int``
        (``control.py:3806-3808`` BE, ``3851-3885`` CN; the pure mass couplings among them are
        re-sent unchanged), their copies inside the outer ``block_00``, and the
        pressure-space analogues."""
        m = bl["m"]
        i00, i01, i10, i11 = bl["inner"]
        c00, c01, c10, c11 = bl["commutator"]
        own = self._owns
        for (i, j), A in i01.items():          # -> outer block_00 (i, m + j)
            if A is not None and own(i):
                self.inner.update_block_values(1, i, j, A)
                self.outer.update_block_values(0, i, m + j, A)
                self.uploads += 2
        for (i, j), A in i10.items():          # -> outer block_00 (m + i, j)
            if A is not None and own(i):
                self.inner.update_block_values(2, i, j, A)
                self.outer.update_block_values(0, m + i, j, A)
                self.uploads += 2
        for q, blk in ((1, c01), (2, c10)):
            for (i, j), A in blk.items():
                if A is not None and own(i):
                    self.comm.update_block_values(q, i, j, A)
                    self.uploads += 1

    def _owns(self, level):
        """Block rows of time level ``level`` live on this rank."""
        return self.dist is None or self.inner._lo <= level < self.inner._hi

    def linear_solve(self, D, Dp, b_0, b_1):
        bl = self._blocks(D, Dp)
        if self.outer is None:
            self._build(bl)
        else:
            self._update(bl)
        u_0 = np.zeros_like(b_0)
        u_1 = np.zeros_like(b_1)
        if self.dist is None:
            ksp = self.outer.solve(u_0, u_1, b_0, b_1, solver_parameters=self.solver_parameters,
                                   pc_fn=self.pc)
            return u_0, u_1, ksp.getIterationNumber()
        # this rank's levels of both block families of a variable: rows [lo, hi) and m + [lo, hi)
        m, lo, hi = bl["m"], self.inner._lo, self.inner._hi
        pick = list(range(lo, hi)) + list(range(m + lo, m + hi))
        l_0, l_1 = np.zeros_like(b_0[pick]), np.zeros_like(b_1[pick])
        ksp = self.outer.solve(l_0, l_1, np.ascontiguousarray(b_0[pick]),
                               np.ascontiguousarray(b_1[pick]),
                               solver_parameters=self.solver_parameters, pc_fn=self.pc)
        u_0[pick], u_1[pick] = l_0, l_1
        for u in (u_0, u_1):            # the other ranks' rows are zero here: a sum gathers
            flat = u.reshape(-1)
            self.host_allreduce(flat, 0)
        return u_0, u_1, ksp.getIterationNumber()


def incompressible_non_linear_solve(pb: NavierStokesControl, linear_solver, *,
                                    max_non_linear_iter=10, relative_non_linear_tol=1.0e-5,
                                    absolute_non_linear_tol=1.0e-8, v=None, zeta=None, p=None,
                                    mu=None, print_error_non_linear=True):
    """``control.py:4886-5232`` (BE).  ``linear_solver.linear_solve(D, Dp, b_0, b_1)`` returns
    the update ``(u_0, u_1, iterations)`` of the linearised system whose forward operator at
    time level ``i`` is ``D[i]`` (velocity space) / ``Dp[i]`` (pressure space).

    Returns a dict with the converged fields, the non-linear residual norms (``norm_0``
    first) and the linear iteration counts."""
    th, n_t, tau = pb.disc, pb.n_t, pb.tau
    m = n_t - 1 if pb.CN else n_t
    v = np.zeros((n_t, th.n_v)) if v is None else np.array(v, dtype=np.float64)
    zeta = np.zeros((n_t, th.n_v)) if zeta is None else np.array(zeta, dtype=np.float64)
    p = np.zeros((m, th.n_p)) if p is None else np.array(p, dtype=np.float64)
    mu = np.zeros((m, th.n_p)) if mu is None else np.array(mu, dtype=np.float64)
    if pb.CN:
        v[0] = np.zeros(th.n_v) if pb.v_0 is None else pb.v_0   # :4961-4962
    zeta[n_t - 1] = 0.0                                          # :4963

    def evaluate():
        D = [pb.D_v(v[i]) for i in range(n_t)]
        r = non_linear_res_eval(pb, D, v, zeta, p, mu)
        return D, r, float(np.sqrt(sum(np.vdot(x, x) for x in r)))

    D, (r00, r01, r10, r11), norm_0 = evaluate()
    norm_k = norm_0
    norms, lin_its = [norm_0], []
    if print_error_non_linear:
        print(f"Initial non-linear residual: {norm_0:.16e}")
    k = 0
    while norm_k > relative_non_linear_tol * norm_0 and norm_k > absolute_non_linear_tol:
        Dp = [pb.D_p(v[i]) for i in range(n_t)]
        s10, s11 = tau * r10, tau * r11                          # :5102-5105
        if pb.CN:    # the linear solve transforms the rows it is given (control.py:4266-4269)
            r00, r01 = _apply_T_1(r00), _apply_T_2(r01)
            s10, s11 = _apply_T_2(s10), _apply_T_1(s11)
        b_0 = np.concatenate([r00, r01])
        b_1 = np.concatenate([s10, s11])
        u_0, u_1, its = linear_solver.linear_solve(D, Dp, b_0, b_1)
        lin_its.append(its)
        if pb.CN:                # unknown block i: v at level i + 1, zeta at level i
            v[1:] += u_0[:m]
            zeta[:m] += u_0[m:]
        else:
            v += u_0[:m]                                         # :5127-5147
            zeta += u_0[m:]
        zeta[:, th.boundary_v] = 0.0
        mu += u_1[:m]            # pressure blocks: mu with the v rows, p with the zeta rows
        p += u_1[m:]
        D, (r00, r01, r10, r11), norm_k = evaluate()
        norms.append(norm_k)
        k += 1
        if print_error_non_linear:
            print(f"Non-linear solver: iteration {k:d}, non-linear residual norm {norm_k:.16e}")
        if k + 1 > max_non_linear_iter:                          # :5186-5187
            break
    return dict(v=v, zeta=zeta, p=p, mu=mu, norms=norms, linear_iterations=lin_its,
                converged=bool(norm_k <= relative_non_linear_tol * norm_0
                               or norm_k <= absolute_non_linear_tol))
