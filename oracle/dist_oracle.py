"""Time-sharded restatement of the oracle (TEST INFRASTRUCTURE, CPU, NumPy).

Mirrors, rank by rank, the distributed algorithm of ``control_amd/csrc`` (SURVEY 8e;
new functionality -- the reference does not distribute time): contiguous time-block
rows per rank, one ``N_x`` halo per neighbour and coupled variable for the operator, an
all-reduce for the Krylov inner products, and pipelined hand-offs for the time-serial
sweeps and scans of the block-Schur preconditioner (``control/control.py:2191-2438`` BE,
``1995-2189`` CN).  ``tests/test_dist_gloo.py`` runs it on 2 ranks over
``torch.distributed`` (gloo) and compares every shard with the single-rank oracle.

``comm`` needs: ``rank``, ``world``, ``allreduce(array) -> array`` (sum),
``send(array, dst)``, ``recv(n, src) -> array``.
"""
from __future__ import annotations

import numpy as np

from . import kkt_oracle as ko


def shard_range(m, rank, world):
    q, r = divmod(m, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


class ShardedHeatSystem:
    """BE / CN heat-control KKT system with Dirichlet nullspaces, rows [lo, hi) local."""

    def __init__(self, M, blocks, m, nodes, CN, comm):
        self.M, self.blocks, self.m, self.nodes, self.CN, self.comm = M, blocks, m, nodes, CN, comm
        self.lo, self.hi = shard_range(m, comm.rank, comm.world)
        self.nx = M.shape[0]
        self.up = comm.rank + 1 if self.hi < m else None
        self.dn = comm.rank - 1 if self.lo > 0 else None

    # -- neighbour hand-offs (every send has exactly one matching recv on the peer)
    def _xchg(self, send_up=None, send_dn=None, want_dn=False, want_up=False):
        c = self.comm
        if send_up is not None and self.up is not None:
            c.send(send_up, self.up)
        from_dn = c.recv(self.nx, self.dn) if want_dn and self.dn is not None else None
        if send_dn is not None and self.dn is not None:
            c.send(send_dn, self.dn)
        from_up = c.recv(self.nx, self.up) if want_up and self.up is not None else None
        return from_dn, from_up

    def split(self, x):
        n = self.hi - self.lo
        return x[:n * self.nx].reshape(n, self.nx), x[n * self.nx:].reshape(n, self.nx)

    def mult(self, x):
        b00, b01, b10, b11 = self.blocks
        lo, hi, m = self.lo, self.hi, self.m
        x0, x1 = self.split(np.asarray(x, dtype=np.float64))
        xc0, xc1 = x0.copy(), x1.copy()
        xc0[:, self.nodes] = 0.0
        xc1[:, self.nodes] = 0.0
        h0_dn, _ = self._xchg(send_up=xc0[-1], want_dn=True)         # x0 block lo-1
        _, h1_up = self._xchg(send_dn=xc1[0], want_up=True)          # x1 block hi

        def X0(j):
            return xc0[j - lo] if lo <= j < hi else h0_dn

        def X1(j):
            return xc1[j - lo] if lo <= j < hi else h1_up
        y0, y1 = np.zeros_like(x0), np.zeros_like(x1)
        for (i, j), A in b00.items():
            if A is not None and lo <= i < hi:
                y0[i - lo] += A @ X0(j)
        for (i, j), A in b01.items():
            if A is not None and lo <= i < hi:
                y0[i - lo] += A @ X1(j)
        for (i, j), A in b10.items():
            if A is not None and lo <= i < hi:
                y1[i - lo] += A @ X0(j)
        for (i, j), A in b11.items():
            if A is not None and lo <= i < hi:
                y1[i - lo] += A @ X1(j)
        if self.CN:
            # T_1 needs raw row 0 of the next rank, T_2 raw row -1 of the previous one
            _, r0_up = self._xchg(send_dn=y0[0].copy(), want_up=True)
            r1_dn, _ = self._xchg(send_up=y1[-1].copy(), want_dn=True)
            n0 = y0.copy()
            n0[:-1] += y0[1:]
            if r0_up is not None:
                n0[-1] += r0_up
            n1 = y1.copy()
            n1[1:] += y1[:-1]
            if r1_dn is not None:
                n1[0] += r1_dn
            y0, y1 = n0, n1
        y0[:, self.nodes] = x0[:, self.nodes]      # alpha = 1
        y1[:, self.nodes] = x1[:, self.nodes]
        return np.concatenate([y0.ravel(), y1.ravel()])

    # ---- scans with a carry from the neighbour rank
    def _scan_T1_inv(self, x):
        _, carry = self._xchg(want_up=True)
        y = x.copy()
        for i in range(y.shape[0] - 1, -1, -1):
            nxt = y[i + 1] if i + 1 < y.shape[0] else carry
            if nxt is not None:
                y[i] -= nxt
        self._xchg(send_dn=y[0])
        return y

    def _scan_T2_inv(self, x):
        carry, _ = self._xchg(want_dn=True)
        y = x.copy()
        for i in range(y.shape[0]):
            prv = y[i - 1] if i >= 1 else carry
            if prv is not None:
                y[i] -= prv
        self._xchg(send_up=y[-1])
        return y, carry

    def _T2(self, x, halo=None, exchange=True):
        if exchange:
            halo, _ = self._xchg(send_up=x[-1].copy(), want_dn=True)
        y = x.copy()
        y[1:] += x[:-1]
        if halo is not None:
            y[0] += halo
        return y

    def make_pc(self, n_t, tau, beta, mass, schur, epsilon=1.0e-3):
        M, nodes, lo, hi, m = self.M, self.nodes, self.lo, self.hi, self.m
        b00, b01, b10, b11 = self.blocks
        Mt = ko.assemble_with_bcs(M, nodes)
        mdinv = 1.0 / Mt.diagonal()
        cache = {}

        def inner(At, dinv, spec, rhs):
            if spec.its == 0:
                return dinv * rhs
            return ko.chebyshev_jacobi(At, dinv, rhs, spec.emin, spec.emax, spec.its, spec.eimag)

        co = getattr(schur, "coarse", None)

        def solve(blk, c, rhs):
            key = (id(blk), c)
            if key not in cache:
                At = ko.assemble_with_bcs(blk if c == 0.0 else blk + c * M, nodes)
                # two-grid form: the sub-solve of a level is local to the rank that owns it
                cache[key] = (At, 1.0 / At.diagonal(), blk,
                              ko.coarse_inverse(At, co) if co is not None else None)
            At, dinv, _, Einv = cache[key]
            if co is not None:
                return ko.coarse_chebyshev(At, dinv, rhs, schur, Einv)
            return inner(At, dinv, schur, rhs)

        def bc(v):
            v[nodes] = 0.0
            return v

        def pc_BE(b):
            b0, b1 = self.split(b)
            n = hi - lo
            shift = tau / beta**0.5
            u0 = np.zeros_like(b0)
            u1 = np.zeros_like(b0)
            for i in range(n):
                u0[i] = inner(Mt, mdinv, mass, bc(b0[i].copy())) * (1.0 / tau)
                if lo + i == m - 1:
                    u0[i] *= 1.0 / epsilon
            h_u0, _ = self._xchg(send_up=u0[-1], want_dn=True)
            B = np.zeros_like(b0)
            for i in range(n):
                g = lo + i
                B[i] = b10[(g, g)] @ u0[i]
                if g >= 1:
                    B[i] += b10[(g, g - 1)] @ (u0[i - 1] if i >= 1 else h_u0)
                B[i] -= bc(b1[i].copy())
                bc(B[i])

            def coef(g):
                return 0.0 if g == 0 else ((epsilon**0.5) * shift if g == m - 1 else shift)
            prev, _ = self._xchg(want_dn=True)
            for i in range(n):
                g = lo + i
                if g >= 1:
                    B[i] -= b10[(g, g - 1)] @ (u1[i - 1] if i >= 1 else prev)
                    bc(B[i])
                u1[i] = solve(b10[(g, g)], coef(g), B[i])
            self._xchg(send_up=u1[-1])
            for i in range(n):
                B[i] = bc((M @ u1[i]) * ((epsilon * tau) if lo + i == m - 1 else tau))
            _, nxt = self._xchg(want_up=True)
            for i in range(n - 1, -1, -1):
                g = lo + i
                if g <= m - 2:
                    B[i] -= b01[(g, g + 1)] @ (u1[i + 1] if i + 1 < n else nxt)
                    bc(B[i])
                u1[i] = solve(b01[(g, g)], coef(g), B[i])
            self._xchg(send_dn=u1[0])
            u0[:, nodes] = b0[:, nodes]
            u1[:, nodes] = b1[:, nodes]
            return np.concatenate([u0.ravel(), u1.ravel()])

        def pc_CN(b):
            b0, b1 = self.split(b)
            n = hi - lo
            c = 0.5 * tau / beta**0.5
            cM = c * M
            b0c = b0.copy()
            b0c[:, nodes] = 0.0
            b1c = b1.copy()
            b1c[:, nodes] = 0.0
            T = self._scan_T1_inv(b0c)
            u0 = np.zeros_like(b0)
            u1 = np.zeros_like(b0)
            for i in range(n):
                u0[i] = inner(Mt, mdinv, mass, T[i]) * (2.0 / tau)
            u0, h_u0 = self._scan_T2_inv(u0)
            B = np.zeros_like(b0)
            for i in range(n):
                g = lo + i
                B[i] = b10[(g, g)] @ u0[i]
                if g >= 1:
                    B[i] += b10[(g, g - 1)] @ (u0[i - 1] if i >= 1 else h_u0)
                bc(B[i])
            B = self._T2(B)
            for i in range(n):
                B[i] -= b1c[i]
                bc(B[i])
            B, _ = self._scan_T2_inv(B)
            prev, _ = self._xchg(want_dn=True)
            for i in range(n):
                g = lo + i
                if g >= 1:
                    p_ = u1[i - 1] if i >= 1 else prev
                    B[i] -= b10[(g, g - 1)] @ p_
                    B[i] -= cM @ p_
                    bc(B[i])
                u1[i] = solve(b10[(g, g)], c, B[i])
            self._xchg(send_up=u1[-1])
            u1 = self._T2(u1, halo=prev, exchange=False)
            for i in range(n):
                B[i] = bc((M @ u1[i]) * (0.5 * tau))
            _, nxt = self._xchg(want_up=True)
            for i in range(n - 1, -1, -1):
                g = lo + i
                if g <= m - 2:
                    B[i] -= (b01[(g, g + 1)] + cM) @ (u1[i + 1] if i + 1 < n else nxt)
                    bc(B[i])
                u1[i] = solve(b01[(g, g)], c, B[i])
            self._xchg(send_dn=u1[0])
            u0[:, nodes] = b0[:, nodes]
            u1[:, nodes] = b1[:, nodes]
            return np.concatenate([u0.ravel(), u1.ravel()])

        return pc_CN if self.CN else pc_BE

    def solve(self, b_local, pc, *, ksp="fgmres", restart=10, rtol=1e-6, max_it=60):
        b = np.asarray(b_local, dtype=np.float64).copy()
        b0, b1 = self.split(b)
        b0[:, self.nodes] = 0.0
        b1[:, self.nodes] = 0.0
        x = np.zeros_like(b)
        kw = dict(restart=restart, rtol=rtol, atol=0.0, divtol=1e4, max_it=max_it,
                  reduce=self.comm.allreduce)
        if ksp == "gmres":
            res = ko.gmres(self.mult, pc, b, x, **kw)
        else:
            res = ko.fgmres(self.mult, pc, b, x, **kw)
        x0, x1 = self.split(x)
        x0[:, self.nodes] = 0.0
        x1[:, self.nodes] = 0.0
        return x, res
