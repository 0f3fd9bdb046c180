"""CPU oracle (NumPy/SciPy) for the all-at-once KKT hot path.  TEST INFRASTRUCTURE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module; the product (``control_amd``) never does.

It restates, function by function, what the reference computes on the path
``MultiBlockSystem.solve()`` (all citations relative to ``/root/reference``):

* ``OracleSystem.mult``       <- ``preconditioner/preconditioner.py:375-543``
* ``OracleSystem.pc_apply``   <- ``preconditioner/preconditioner.py:562-656``
* ``OracleSystem.solve``      <- ``preconditioner/preconditioner.py:658-786``
* nullspace classes           <- ``preconditioner/preconditioner.py:75-213``
* ``apply_T_1/2`` (+inverses) <- ``preconditioner/preconditioner.py:33-60``,
                                 ``control/control.py:26-96``
* ``pc_instationary_BE/CN``   <- ``control/control.py:2191-2438`` / ``1995-2189``
* ``pc_stationary``           <- ``control/control.py:356-448``

Third-party arithmetic (PETSc ``KSP`` gmres/fgmres/chebyshev, ``PC`` jacobi;
petsc4py, version unpinned by the reference, absent from this image) is restated from
the published algorithms: ``gmres``/``fgmres`` follow ``KSPSolve_GMRES`` /
``KSPSolve_FGMRES`` (classical Gram-Schmidt without refinement, Givens rotations,
``KSPConvergedDefault`` with the right-hand-side norm as reference because the
reference sets ``setInitialGuessNonzero(True)``, ``preconditioner.py:743``);
``chebyshev_jacobi`` follows ``KSPSolve_Chebyshev`` (first kind, fixed bounds, zero
initial guess, exactly ``its`` steps).  hypre BoomerAMG sub-solves are replaced, as
BASELINE.json's ``north_star`` prescribes, by the same Jacobi-Chebyshev iteration with
a stated degree and stated bounds.

PARITY PINNING: the reference's own known-answer tests (``test/test_control.py:26-119``,
``1243-1444``, ``1447-1655``) pin the operator and the converged solution; this oracle is
checked against restatements of them in ``tests/test_oracle_kat.py``.  Krylov iterates
and iteration counts are pinned by NO reference test or fixture: for those, parity is
"unpinned" (oracle-vs-GPU agreement only).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np
import scipy.sparse as sp

# --------------------------------------------------------------------------- T ops


def apply_T_1(x):
    """``new_i = old_i + old_{i+1}`` (``preconditioner.py:33-45``).  x: (n, nx)."""
    y = x.copy()
    y[:-1] += x[1:]
    return y


def apply_T_2(x):
    """``new_i = old_i + old_{i-1}`` (``preconditioner.py:48-60``)."""
    y = x.copy()
    y[1:] += x[:-1]
    return y


def apply_T_1_inv(x):
    """Sequential: ``for i=n-2..0: x_i -= x_{i+1}`` (``control.py:63-78``)."""
    y = x.copy()
    for i in range(y.shape[0] - 2, -1, -1):
        y[i] -= y[i + 1]
    return y


def apply_T_2_inv(x):
    """Sequential: ``for i=1..n-1: x_i -= x_{i-1}`` (``control.py:81-96``)."""
    y = x.copy()
    for i in range(1, y.shape[0]):
        y[i] -= y[i - 1]
    return y


# ---------------------------------------------------------------------- nullspaces


class NoneNullspace:
    """``preconditioner.py:119-130``."""

    def lhs_right(self, x):
        pass

    def lhs_left(self, y):
        pass

    def extended_correct_lhs(self, x, y):
        pass

    def pc_extended_correct_soln(self, u, b):
        pass


class DirichletBCNullspace(NoneNullspace):
    """``preconditioner.py:158-197``: zero the listed dofs; add ``alpha * x`` there."""

    def __init__(self, nodes, *, alpha=1.0):
        self.nodes = np.asarray(nodes, dtype=np.int64)
        self.alpha = alpha

    def lhs_right(self, x):
        x[self.nodes] = 0.0

    def lhs_left(self, y):
        y[self.nodes] = 0.0

    def extended_correct_lhs(self, x, y):
        y[self.nodes] += self.alpha * x[self.nodes]

    def pc_extended_correct_soln(self, u, b):
        u[self.nodes] += 1.0 * b[self.nodes]


class ConstantNullspace(NoneNullspace):
    """``preconditioner.py:133-155``: shift by the arithmetic mean."""

    def __init__(self, *, alpha=1.0):
        self.alpha = alpha

    @staticmethod
    def _correct(x, y, alpha):
        y += alpha * x.sum() / float(x.size)

    def lhs_right(self, x):
        self._correct(x, x, -1.0)

    def lhs_left(self, y):
        self._correct(y, y, -1.0)

    def extended_correct_lhs(self, x, y):
        self._correct(x, y, self.alpha)

    def pc_extended_correct_soln(self, u, b):
        self._correct(b, u, 1.0)


class FullNullspace(NoneNullspace):
    """``preconditioner.py:200-213``."""

    def lhs_right(self, x):
        x[:] = 0.0

    def lhs_left(self, y):
        y[:] = 0.0

    def extended_correct_lhs(self, x, y):
        y[:] = x

    def pc_extended_correct_soln(self, u, b):
        u[:] = b


# ------------------------------------------------------------------- Krylov (PETSc)

CONVERGED_RTOL = 2
CONVERGED_ATOL = 3
CONVERGED_HAPPY_BREAKDOWN = 5
DIVERGED_ITS = -3
DIVERGED_DTOL = -4
DIVERGED_BREAKDOWN = -5
DIVERGED_NANORINF = -9


@dataclass
class KSPResult:
    """What the reference reads off the returned KSP (``preconditioner.py:786``)."""
    reason: int = 0
    its: int = 0
    rnorm: float = 0.0
    history: list = field(default_factory=list)

    def getConvergedReason(self):
        return self.reason

    def getIterationNumber(self):
        return self.its

    def getResidualNorm(self):
        return self.rnorm


class _ConvergedDefault:
    """``KSPConvergedDefault`` with the RHS norm as ``rnorm0`` (nonzero initial guess)."""

    def __init__(self, rtol, atol, divtol, rnorm0):
        self.rtol = rtol
        self.atol = atol
        self.divtol = divtol
        self.rnorm0 = rnorm0
        self.ttol = max(rtol * rnorm0, atol)

    def __call__(self, rnorm):
        if not np.isfinite(rnorm):
            return DIVERGED_NANORINF
        if self.rnorm0 == 0.0:
            # KSPConvergedDefault, "special case of zero RHS and nonzero guess": snorm = rnorm
            self.rnorm0 = rnorm
            self.ttol = max(self.rtol * rnorm, self.atol)
        if rnorm <= self.ttol:
            return CONVERGED_ATOL if rnorm < self.atol else CONVERGED_RTOL
        if rnorm >= self.divtol * self.rnorm0:
            return DIVERGED_DTOL
        return 0


def _gmres_driver(A, B, b, x, *, restart, rtol, atol, divtol, max_it, flexible,
                  right, monitor=None, reduce=None, trace=None):
    """GMRES(m) / FGMRES(m), structured like ``KSPSolve_GMRES`` + ``KSPGMRESCycle``.

    left  (gmres default):  Krylov on B A, monitored norm ||B r||.
    right (fgmres, or gmres with pc_side right): Krylov on A B, monitored norm ||r||.
    """
    n = b.size
    res = KSPResult()
    # KSPConvergedDefault at it == 0 with a nonzero guess: norm of the (preconditioned) rhs
    # `reduce` sums partial inner products over the ranks of a time-sharded run
    # (oracle/dist_oracle.py); None = single rank
    def mdot(Vk, w):
        h = Vk @ w
        return reduce(h) if reduce is not None else h

    def vnorm(v):
        return float(np.sqrt(mdot(v[None, :], v)[0])) if reduce is not None \
            else np.linalg.norm(v)

    rnorm0 = vnorm(b) if right else vnorm(B(b))
    conv = _ConvergedDefault(rtol, atol, divtol, rnorm0)
    haptol = 1.0e-30
    its = 0
    m = restart
    # (the arithmetic type follows the right-hand side: float64 everywhere in the parity tests;
    # numpy.longdouble in the extended-precision study of the BE trajectories,
    # tests/golden/make_extended_histories.py)
    dt = b.dtype
    V = np.zeros((m + 1, n), dtype=dt)
    Z = np.zeros((m, n), dtype=dt) if flexible else None
    H = np.zeros((m + 1, m), dtype=dt)
    cc = np.zeros(m, dtype=dt)
    ss = np.zeros(m, dtype=dt)
    grs = np.zeros(m + 1, dtype=dt)
    reason = 0
    while True:
        # KSPInitialResidual
        r = b - A(x)
        V[0] = r if right else B(r)
        rn = vnorm(V[0])
        res.history.append(rn)
        if monitor is not None:
            monitor(its, rn)
        res.rnorm = rn
        if rn == 0.0:
            reason = CONVERGED_ATOL
            break
        V[0] /= rn
        grs[:] = 0.0
        grs[0] = rn
        reason = conv(rn)
        it = 0
        H[:] = 0.0
        while not reason and it < m and its < max_it:
            if it:
                res.history.append(rn)
                if monitor is not None:
                    monitor(its, rn)
            if right:
                z = B(V[it])
                if flexible:
                    Z[it] = z
                w = A(z)
            else:
                w = B(A(V[it]))
            # classical Gram-Schmidt, no refinement (VecMDot, VecMAXPY)
            h = mdot(V[:it + 1], w)
            w = w - h @ V[:it + 1]
            H[:it + 1, it] = h
            tt = vnorm(w)
            if trace is not None:
                # step-locked parity (tests): the state this step started from and what it
                # produced, before the Givens update
                trace.append(dict(its=its, it=it, V=V[:it + 1].copy(), h=np.array(h, copy=True),
                                  tt=tt, v_next=(w / tt if tt > 0 else w).copy(), grs_it=grs[it],
                                  x=x.copy()))
            hapbnd = min(abs(tt / grs[it]), haptol)
            hapend = tt < hapbnd
            if not hapend:
                V[it + 1] = w / tt
            H[it + 1, it] = tt
            # KSPGMRESUpdateHessenberg
            for j in range(it):
                t = H[j, it]
                H[j, it] = cc[j] * t + ss[j] * H[j + 1, it]
                H[j + 1, it] = cc[j] * H[j + 1, it] - ss[j] * t
            if not hapend:
                t = np.sqrt(H[it, it] * H[it, it] + H[it + 1, it] * H[it + 1, it])
                if t == 0.0:
                    reason = DIVERGED_BREAKDOWN
                    break
                cc[it] = H[it, it] / t
                ss[it] = H[it + 1, it] / t
                grs[it + 1] = -(ss[it] * grs[it])
                grs[it] = cc[it] * grs[it]
                H[it, it] = cc[it] * H[it, it] + ss[it] * H[it + 1, it]
                rn = abs(grs[it + 1])
            else:
                rn = 0.0
            it += 1
            its += 1
            res.rnorm = rn
            if trace is not None:
                trace[-1]["rn"] = rn
            reason = conv(rn)
            if hapend and not reason:
                reason = DIVERGED_BREAKDOWN
        if it and (reason or its >= max_it):
            res.history.append(rn)
            if monitor is not None:
                monitor(its, rn)
        # KSPGMRESBuildSoln
        if it > 0:
            y = np.zeros(it, dtype=dt)
            for k in range(it - 1, -1, -1):
                y[k] = (grs[k] - H[k, k + 1:it] @ y[k + 1:]) / H[k, k]
            if flexible:
                x += y @ Z[:it]
            elif right:
                x += B(y @ V[:it])
            else:
                x += y @ V[:it]
        if reason:
            break
        if its >= max_it:
            reason = DIVERGED_ITS
            break
    res.reason = reason
    res.its = its
    return res


def gmres(A, B, b, x, **kw):
    kw.setdefault("right", False)
    return _gmres_driver(A, B, b, x, flexible=False, **kw)


def fgmres(A, B, b, x, **kw):
    kw.pop("right", None)
    return _gmres_driver(A, B, b, x, flexible=True, right=True, **kw)


DIVERGED_INDEFINITE_PC = -8
DIVERGED_INDEFINITE_MAT = -10


def minres(A, B, b, x, *, rtol, atol, divtol, max_it, monitor=None, restart=None,
           right=False, reduce=None):
    """Preconditioned MINRES (Paige & Saunders 1975), structured like the classic
    ``KSPSolve_MINRES`` of PETSc up to 3.18 (``src/ksp/ksp/impls/minres/minres.c``): Lanczos
    on ``A`` in the ``B`` inner product, one Givens rotation per step, monitored norm
    ``||B r_0||_2 * prod |s_k|`` (recurrence), ``B`` must be symmetric positive definite.

    Reachable in the reference through ``solver_parameters["linear_solver"] = "minres"``
    (``preconditioner.py:733``) but never used by it or its tests.  PETSc is
    not in this image and 3.19+ replaced the implementation (same iterates in exact
    arithmetic, another residual estimate): parity with PETSc is unpinned; the iterates
    are pinned against SciPy's independent ``minres`` in ``tests/test_oracle.py``.
    """
    haptol = 1.0e-50
    dot = (lambda u, v: float(np.dot(u, v))) if reduce is None else \
        (lambda u, v: float(reduce(np.array([np.dot(u, v)]))[0]))
    res = KSPResult()
    conv = _ConvergedDefault(rtol, atol, divtol, np.sqrt(dot(B(b), B(b))))
    n = b.size
    uold, vold, w, wold = np.zeros(n), np.zeros(n), np.zeros(n), np.zeros(n)
    r = b - A(x)
    z = B(r)
    nrm = np.sqrt(dot(z, z))
    dp = dot(r, z)
    res.rnorm = nrm
    if dp < haptol and nrm > haptol:
        res.reason = DIVERGED_INDEFINITE_PC
        return res
    res.history.append(nrm)
    if monitor is not None:
        monitor(0, nrm)
    res.reason = conv(nrm)
    if res.reason:
        return res
    beta = np.sqrt(abs(dp))
    eta = beta
    v, u = r * (1.0 / beta), z * (1.0 / beta)
    c = cold = 1.0
    s = sold = 0.0
    i = 0
    while i < max_it:
        res.its = i + 1
        r = A(u)                                   # Lanczos
        alpha = dot(u, r)
        z = B(r)
        r -= alpha * v
        z -= alpha * u
        r -= beta * vold
        z -= beta * uold
        betaold = beta
        dp = dot(r, z)
        beta = np.sqrt(abs(dp))
        coold, cold, soold, sold = cold, c, sold, s    # QR factorisation
        rho0 = cold * alpha - coold * sold * betaold
        rho1 = np.sqrt(rho0 * rho0 + beta * beta)
        rho2 = sold * alpha + coold * cold * betaold
        rho3 = soold * betaold
        c, s = rho0 / rho1, beta / rho1
        woold, wold = wold, w
        w = u - rho2 * wold
        w -= rho3 * woold
        w *= 1.0 / rho1
        x += (c * eta) * w
        if dp < haptol:                            # converged or indefinite: true residual
            nrm = np.sqrt(dot(A(x) - b, A(x) - b))
        else:
            nrm *= abs(s)
        res.rnorm = nrm
        res.history.append(nrm)
        if monitor is not None:
            monitor(i + 1, nrm)
        res.reason = conv(nrm)
        if res.reason:
            break
        if dp < haptol:
            res.reason = DIVERGED_INDEFINITE_MAT
            break
        eta = -s * eta
        vold, uold = v, u
        v, u = r * (1.0 / beta), z * (1.0 / beta)
        i += 1
    if not res.reason:
        res.reason = DIVERGED_ITS
    return res


def chebyshev_ellipse_coefficients(emin, emax, eimag, its):
    """Coefficients ``(c1, c2, c3)`` of steps ``2 .. its`` of the Chebyshev iteration for a
    spectrum inside the ellipse with centre ``d = (emax + emin) / 2`` and semi-axes
    ``a = (emax - emin) / 2`` (real direction) and ``eimag`` (imaginary direction):
    ``p_{k+1} = c1 p_{k-1} + c2 p_k + c3 D^-1 (b - A p_k)`` (Manteuffel, Numer. Math. 28, 1977:
    ``alpha_1 = 2d / (2d^2 - c^2)``, ``alpha_n = 1 / (d - (c^2 / 4) alpha_{n-1})``,
    ``beta_n = d alpha_n - 1`` with ``c^2 = a^2 - eimag^2``, which may be negative -- the
    recurrence only ever sees ``c^2``, so the arithmetic stays real).  The first step is
    ``p_1 = (1 / d) D^-1 b`` as for a real interval.  Not in the reference: its sub-solves on the
    non-symmetric convection blocks are BoomerAMG cycles (``control.py:2277-2288``); the sweep
    shape (one SpMV per step, three-term update) is the one north_star prescribes."""
    d = 0.5 * (emax + emin)
    a = 0.5 * (emax - emin)
    c2 = a * a - eimag * eimag
    out = []
    alpha = 1.0 / d
    for n in range(1, its):
        alpha = 1.0 / (d - (0.5 if n == 1 else 0.25) * c2 * alpha)
        beta = d * alpha - 1.0
        out.append((-beta, 1.0 + beta, alpha))
    return out


def chebyshev_jacobi(A, dinv, b, emin, emax, its, eimag=0.0):
    """``KSPSolve_Chebyshev`` (first kind) with ``PCJACOBI``, zero guess, ``its`` steps.

    Options of the reference: ``control/control.py:1973-1982`` (``ksp_max_it 20``,
    ``rtol = atol = 0``, fixed eigenvalue bounds, no estimation).  ``eimag > 0``: the same
    three-term sweep with the coefficients of an ellipse (``chebyshev_ellipse_coefficients``),
    for blocks with a convection term.
    """
    if eimag > 0.0:
        p_km1 = np.zeros_like(b)
        p_k = (2.0 / (emax + emin)) * (dinv * b) + p_km1
        for c1, c2, c3 in chebyshev_ellipse_coefficients(emin, emax, eimag, its):
            z = dinv * (b - A @ p_k)
            p_km1, p_k = p_k, c1 * p_km1 + c2 * p_k + c3 * z
        return p_k
    scale = 2.0 / (emax + emin)
    alpha = 1.0 - scale * emin
    mu = 1.0 / alpha
    omegaprod = 2.0 / alpha
    c_km1 = 1.0
    c_k = mu
    p_km1 = np.zeros_like(b)
    # zero guess: r = b; p_k = scale * B^-1 r + p_km1
    p_k = scale * (dinv * b) + p_km1
    for _ in range(1, its):
        c_kp1 = 2.0 * mu * c_k - c_km1
        omega = omegaprod * c_k / c_kp1
        r = b - A @ p_k
        z = dinv * r
        # VecAXPBYPCZ(p_kp1, 1-omega, omega, scale*omega, p_km1, p_k)
        p_kp1 = (1.0 - omega) * p_km1 + omega * p_k + (scale * omega) * z
        p_km1, p_k = p_k, p_kp1
        c_km1, c_k = c_k, c_kp1
    return p_k


def chebyshev_jacobi_from(A, dinv, b, x0, emin, emax, its, eimag=0.0):
    """``its`` steps of the same iteration from the initial guess ``x0`` (``KSPSolve_Chebyshev``
    with a non-zero guess: ``p_1 = x_0 + scale B (b - A x_0)``, then the three-term recurrence
    with ``p_0 = x_0`` and the coefficients of the zero-guess sequence): ``its`` SpMVs."""
    if its <= 0:
        return x0.copy()
    scale = 2.0 / (emax + emin)
    p_km1 = x0
    p_k = x0 + scale * (dinv * (b - A @ x0))
    if eimag > 0.0:
        coefs = chebyshev_ellipse_coefficients(emin, emax, eimag, its)
    else:
        alpha = 1.0 - scale * emin
        mu = 1.0 / alpha
        omegaprod = 2.0 / alpha
        c_km1, c_k = 1.0, mu
        coefs = []
        for _ in range(1, its):
            c_kp1 = 2.0 * mu * c_k - c_km1
            omega = omegaprod * c_k / c_kp1
            coefs.append((1.0 - omega, omega, scale * omega))
            c_km1, c_k = c_k, c_kp1
    for c1, c2, c3 in coefs:
        z = dinv * (b - A @ p_k)
        p_km1, p_k = p_k, c1 * p_km1 + c2 * p_k + c3 * z
    return p_k


@dataclass
class CoarseSpace:
    """Coarse space of the two-grid form of the sub-solves: ``P`` (n x n_c, scipy sparse; rows of
    Dirichlet dofs empty) and the number of cycles.  Not in the reference, which calls BoomerAMG
    for these solves (``control.py:2277-2288``): a cycle is the Galerkin coarse correction
    ``x += P (P^T A P)^-1 P^T (b - A x)`` followed by ``its`` Jacobi-Chebyshev smoothing sweeps
    on ``[emin, emax]`` (the upper part of the spectrum: what the coarse space does not see)."""
    P: object
    cycles: int = 1
    # the sub-solve matrix annihilates the constants (the pressure Laplacian K_p) and the columns of
    # P sum to the constant vector: invert E + (trace E / n_c^2) 1 1^T instead of the singular E
    deflate: bool = False


def coarse_chebyshev(A, dinv, b, spec, Einv):
    """Sub-solve with coarse corrections: ``cycles`` x [Galerkin correction, ``its`` smoothing
    sweeps from the corrected iterate].  The first correction acts on ``b`` (zero guess)."""
    P = spec.coarse.P
    x = np.zeros_like(b)
    for c in range(spec.coarse.cycles):
        r = b if c == 0 else b - A @ x
        x = x + P @ (Einv @ (P.T @ r))
        x = chebyshev_jacobi_from(A, dinv, b, x, spec.emin, spec.emax, spec.its, spec.eimag)
    return x


def coarse_inverse(At, coarse):
    """``(P^T A P)^-1`` of the bc-assembled sub-solve matrix (dense); with ``coarse.deflate`` the
    inverse of ``E + (trace E / n_c^2) 1 1^T``, which acts as the pseudo-inverse of the singular
    ``E`` on right-hand sides orthogonal to the constants."""
    P = coarse.P
    E = (P.T @ (At @ P)).toarray()
    if getattr(coarse, "deflate", False):
        E = E + np.trace(E) / float(E.shape[0]) ** 2
    return np.linalg.inv(E)


# ----------------------------------------------------------------------- the system


def _as_blocks(a, n, nx):
    a = np.asarray(a, dtype=np.float64)
    return a.reshape(n, nx)


class OracleSystem:
    """Restatement of ``MultiBlockSystem`` (``preconditioner.py:216-786``) on SciPy CSR.

    Vectors are NumPy arrays of shape ``(n_blocks, nx)``; the flat KKT vector is the
    ``n_blocks_00`` blocks of variable 0 followed by the ``n_blocks_11`` blocks of
    variable 1 (``preconditioner.py:286-287``).
    """

    def __init__(self, nx0, nx1, block_00, block_01, block_10, block_11, *,
                 n_blocks_00=1, n_blocks_11=1, sub_n_blocks_00_0=None,
                 sub_n_blocks_11_0=None, nullspace_0=None, nullspace_1=None,
                 CN=False, dtype=np.float64):
        n0, n1 = n_blocks_00, n_blocks_11
        self.dtype = dtype      # float64; numpy.longdouble only in the extended-precision study
        if nullspace_0 is None:
            nullspace_0 = tuple(NoneNullspace() for _ in range(n0))
        if nullspace_1 is None:
            nullspace_1 = tuple(NoneNullspace() for _ in range(n1))
        for blk, nr, nc in ((block_00, n0, n0), (block_01, n0, n1),
                            (block_10, n1, n0), (block_11, n1, n1)):
            if len(blk) != nr * nc:
                raise ValueError("Unexpected dimension of blocks")
        self.nx0, self.nx1, self.n0, self.n1 = nx0, nx1, n0, n1
        self.blocks = (block_00, block_01, block_10, block_11)
        self.nullspaces = tuple(nullspace_0) + tuple(nullspace_1)
        self.sub00 = sub_n_blocks_00_0
        self.sub11 = sub_n_blocks_11_0
        self.CN = CN
        self.N = n0 * nx0 + n1 * nx1

    # -- flat <-> blocks
    def split(self, x):
        k = self.n0 * self.nx0
        return (x[:k].reshape(self.n0, self.nx0),
                x[k:].reshape(self.n1, self.nx1))

    def join(self, a, b):
        return np.concatenate([np.ravel(a), np.ravel(b)])

    # -- preconditioner.py:375-543
    def mult(self, x):
        x0, x1 = self.split(np.asarray(x, dtype=self.dtype))
        n0, n1 = self.n0, self.n1
        xc0 = x0.copy()
        xc1 = x1.copy()
        for i in range(n0):
            self.nullspaces[i].lhs_right(xc0[i])
        for i in range(n1):
            self.nullspaces[n0 + i].lhs_right(xc1[i])
        y0 = np.zeros_like(x0)
        y1 = np.zeros_like(x1)
        b00, b01, b10, b11 = self.blocks
        for (i, j), A in b00.items():
            if A is not None:
                y0[i] += A @ xc0[j]
        for (i, j), A in b01.items():
            if A is not None:
                y0[i] += A @ xc1[j]
        for (i, j), A in b10.items():
            if A is not None:
                y1[i] += A @ xc0[j]
        for (i, j), A in b11.items():
            if A is not None:
                y1[i] += A @ xc1[j]
        if self.CN:
            if self.sub00 is None and self.sub11 is None:
                y0 = apply_T_1(y0)
                y1 = apply_T_2(y1)
            else:
                s0, s1 = self.sub00, self.sub11
                y0 = np.concatenate([apply_T_1(y0[:s0]), apply_T_2(y0[s0:])])
                y1 = np.concatenate([apply_T_2(y1[:s1]), apply_T_1(y1[s1:])])
        for i in range(n0):
            ns = self.nullspaces[i]
            ns.lhs_left(y0[i])
            ns.extended_correct_lhs(x0[i], y0[i])
        for i in range(n1):
            ns = self.nullspaces[n0 + i]
            ns.lhs_left(y1[i])
            ns.extended_correct_lhs(x1[i], y1[i])
        return self.join(y0, y1)

    # -- preconditioner.py:562-656
    def pc_apply(self, pc_fn, x):
        b0, b1 = self.split(np.asarray(x, dtype=self.dtype))
        n0, n1 = self.n0, self.n1
        b0c = b0.copy()
        b1c = b1.copy()
        for i in range(n0):
            self.nullspaces[i].lhs_left(b0c[i])
        for i in range(n1):
            self.nullspaces[n0 + i].lhs_left(b1c[i])
        u0 = np.zeros_like(b0)
        u1 = np.zeros_like(b1)
        pc_fn(u0, u1, b0c, b1c)
        for i in range(n0):
            ns = self.nullspaces[i]
            ns.lhs_right(u0[i])
            ns.pc_extended_correct_soln(u0[i], b0[i])
        for i in range(n1):
            ns = self.nullspaces[n0 + i]
            ns.lhs_right(u1[i])
            ns.pc_extended_correct_soln(u1[i], b1[i])
        return self.join(u0, u1)

    # -- preconditioner.py:337-345, 658-786
    def solve(self, u_0, u_1, b_0, b_1, *, solver_parameters=None, pc_fn=None,
              monitor=None, trace=None):
        if solver_parameters is None:
            solver_parameters = {}
        if pc_fn is None:
            def pc_fn(u_0, u_1, b_0, b_1):
                u_0[:] = b_0
                u_1[:] = b_1
        n0, n1 = self.n0, self.n1
        U0 = _as_blocks(u_0, n0, self.nx0)
        U1 = _as_blocks(u_1, n1, self.nx1)
        u0 = U0.copy()
        u1 = U1.copy()
        b0 = _as_blocks(b_0, n0, self.nx0).copy()
        b1 = _as_blocks(b_1, n1, self.nx1).copy()
        for i in range(n0):
            self.nullspaces[i].lhs_right(u0[i])      # correct_soln
            self.nullspaces[i].lhs_left(b0[i])       # correct_rhs
        for i in range(n1):
            self.nullspaces[n0 + i].lhs_right(u1[i])
            self.nullspaces[n0 + i].lhs_left(b1[i])
        u = self.join(u0, u1)
        b = self.join(b0, b1)
        sp_ = solver_parameters
        ksp_type = sp_.get("linear_solver", "fgmres")
        kw = dict(restart=sp_.get("gmres_restart", 30),
                  rtol=sp_["relative_tolerance"], atol=sp_["absolute_tolerance"],
                  divtol=sp_.get("divergence limit", None) or 1.0e4,
                  max_it=sp_.get("maximum_iterations", 1000), monitor=monitor)
        A = self.mult

        def B(v):
            return self.pc_apply(pc_fn, v)

        if ksp_type == "gmres":
            res = gmres(A, B, b, u, right=(sp_.get("pc_side", "left") == "right"), trace=trace,
                        **kw)
        elif ksp_type == "fgmres":
            res = fgmres(A, B, b, u, trace=trace, **kw)
        elif ksp_type == "minres":
            res = minres(A, B, b, u, **kw)
        else:
            raise ValueError(f"oracle restates gmres, fgmres and minres only, not {ksp_type}")
        u0, u1 = self.split(u)
        for i in range(n0):
            self.nullspaces[i].lhs_right(u0[i])
        for i in range(n1):
            self.nullspaces[n0 + i].lhs_right(u1[i])
        if not sp_.get("preconditioner", False) and res.reason <= 0:
            raise RuntimeError("Solver failed to converge")
        U0[:] = u0
        U1[:] = u1
        return res


# ---------------------------------------------------------- built-in preconditioners


def assemble_with_bcs(A, nodes):
    """Firedrake ``assemble(form, bcs=...)``: bc rows/cols zeroed, unit diagonal."""
    A = sp.csr_matrix(A, copy=True)
    A.sort_indices()
    n = A.shape[0]
    keep = np.ones(n)
    keep[nodes] = 0.0
    rows = np.repeat(np.arange(n), np.diff(A.indptr))
    A.data *= keep[rows] * keep[A.indices]
    isdiag = (rows == A.indices) & (keep[rows] == 0.0)
    if int(isdiag.sum()) != int((keep == 0.0).sum()):   # a bc row without a stored diagonal
        A = sp.csr_matrix(A + sp.diags(1.0 - keep))
        A.sort_indices()
    else:
        A.data[isdiag] = 1.0
    return A


@dataclass
class ChebSpec:
    """A Jacobi-Chebyshev inner solve: ``its`` steps on ``[emin, emax]``.

    ``its == 0`` means a single Jacobi application (the reference's ``preonly`` +
    ``jacobi`` branch, ``control/control.py:1984-1991``).
    """
    its: int
    emin: float
    emax: float
    eimag: float = 0.0      # > 0: imaginary semi-axis of the spectrum's ellipse (convection blocks)
    coarse: object = None   # CoarseSpace: two-grid cycles (its = smoothing sweeps per cycle)


def _inner_solve(At, spec, rhs, Einv=None):
    dinv = 1.0 / At.diagonal()
    if getattr(spec, "coarse", None) is not None:
        if Einv is None:       # (one inverse per matrix: kept on the specification)
            cache = spec.__dict__.setdefault("_einv_cache", {})
            if id(At) not in cache:
                cache[id(At)] = (At, coarse_inverse(At, spec.coarse))
            Einv = cache[id(At)][1]
        return coarse_chebyshev(At, dinv, rhs, spec, Einv)
    if spec.its == 0:
        return dinv * rhs
    return chebyshev_jacobi(At, dinv, rhs, spec.emin, spec.emax, spec.its, spec.eimag)


def _bc(v, nodes):
    v[nodes] = 0.0
    return v


def pc_stationary(M, D_v, D_zeta, beta, nodes, mass_spec, schur_spec):
    """``Stationary.construct_pc`` (``control/control.py:351-450``)."""
    Mt = assemble_with_bcs(M, nodes)
    S1 = assemble_with_bcs(D_v + (1.0 / beta**0.5) * M, nodes)
    S2 = assemble_with_bcs(D_zeta + (1.0 / beta**0.5) * M, nodes)

    co = getattr(schur_spec, "coarse", None)
    E1 = coarse_inverse(S1, co) if co is not None else None
    E2 = coarse_inverse(S2, co) if co is not None else None

    def pc_linear(u_0, u_1, b_0, b_1):
        u_0[0] = _inner_solve(Mt, mass_spec, b_0[0])
        b = D_v @ u_0[0] - b_1[0]
        _bc(b, nodes)
        u_1[0] = _inner_solve(S1, schur_spec, b, E1)
        b = M @ u_1[0]
        _bc(b, nodes)
        u_1[0] = _inner_solve(S2, schur_spec, b, E2)
    return pc_linear


def pc_instationary_BE(M, block_01, block_10, n_t, tau, beta, nodes, mass_spec,
                       schur_spec, epsilon=1.0e-3):
    """BE branch of ``Instationary.construct_pc`` (``control/control.py:2191-2438``)."""
    Mt = assemble_with_bcs(M, nodes)
    shift = tau / beta**0.5

    cache = {}

    def solve(blk, c, rhs):
        # The reference re-assembles `blk + c * M` with bcs inside every application
        # (control.py:2277-2288); the matrix does not change, so it is built once here.
        key = (id(blk), c)
        if key not in cache:
            At = assemble_with_bcs(blk if c == 0.0 else blk + c * M, nodes)
            Einv = coarse_inverse(At, schur_spec.coarse) if schur_spec.coarse is not None else None
            cache[key] = (At, 1.0 / At.diagonal(), blk, Einv)
        At, dinv, _, Einv = cache[key]
        if schur_spec.coarse is not None:
            return coarse_chebyshev(At, dinv, rhs, schur_spec, Einv)
        if schur_spec.its == 0:
            return dinv * rhs
        return chebyshev_jacobi(At, dinv, rhs, schur_spec.emin, schur_spec.emax,
                                schur_spec.its, schur_spec.eimag)

    def pc_linear(u_0, u_1, b_0, b_1):
        # (1,1)-block, control.py:2193-2206
        for i in range(n_t):
            u_0[i] = _inner_solve(Mt, mass_spec, b_0[i].copy())
            u_0[i] *= 1.0 / tau
        u_0[n_t - 1] *= 1.0 / epsilon
        # b = D_v u_0 - b_1, control.py:2208-2237
        b = np.zeros_like(u_0)
        b[0] = block_10[(0, 0)] @ u_0[0]
        b[0] -= b_1[0]
        _bc(b[0], nodes)
        for i in range(1, n_t):
            t = block_10[(i, i - 1)] @ u_0[i - 1]
            b[i] = block_10[(i, i)] @ u_0[i]
            b[i] += t
            b[i] -= b_1[i]
            _bc(b[i], nodes)
        # forward sweep, control.py:2241-2327
        u_1[0] = solve(block_10[(0, 0)], 0.0, b[0])
        for i in range(1, n_t - 1):
            b[i] -= block_10[(i, i - 1)] @ u_1[i - 1]
            _bc(b[i], nodes)
            u_1[i] = solve(block_10[(i, i)], shift, b[i])
        b[n_t - 1] -= block_10[(n_t - 1, n_t - 2)] @ u_1[n_t - 2]
        _bc(b[n_t - 1], nodes)
        u_1[n_t - 1] = solve(block_10[(n_t - 1, n_t - 1)], (epsilon**0.5) * shift,
                             b[n_t - 1])
        # b = tau M u_1, control.py:2330-2350
        b = np.zeros_like(u_0)
        for i in range(n_t - 1):
            b[i] = (M @ u_1[i]) * tau
            _bc(b[i], nodes)
        b[n_t - 1] = (M @ u_1[n_t - 1]) * (epsilon * tau)
        _bc(b[n_t - 1], nodes)
        # backward sweep, control.py:2353-2437
        u_1[n_t - 1] = solve(block_01[(n_t - 1, n_t - 1)], (epsilon**0.5) * shift,
                             b[n_t - 1])
        for i in range(n_t - 2, 0, -1):
            b[i] -= block_01[(i, i + 1)] @ u_1[i + 1]
            _bc(b[i], nodes)
            u_1[i] = solve(block_01[(i, i)], shift, b[i])
        b[0] -= block_01[(0, 1)] @ u_1[1]
        _bc(b[0], nodes)
        u_1[0] = solve(block_01[(0, 0)], 0.0, b[0])
    return pc_linear


def pc_instationary_CN(M, block_01, block_10, n_t, tau, beta, nodes, mass_spec,
                       schur_spec):
    """CN branch of ``Instationary.construct_pc`` (``control/control.py:1995-2189``)."""
    m = n_t - 1
    Mt = assemble_with_bcs(M, nodes)
    my_const = 0.5 * tau / beta**0.5
    cM = my_const * M
    ucache = {}

    def upper(i):
        blk = block_01[(i, i + 1)]
        if id(blk) not in ucache:
            ucache[id(blk)] = (blk + cM, blk)
        return ucache[id(blk)][0]

    cache = {}

    def solve(blk, c, rhs):
        # The reference re-assembles `blk + c * M` with bcs inside every application
        # (control.py:2277-2288); the matrix does not change, so it is built once here.
        key = (id(blk), c)
        if key not in cache:
            At = assemble_with_bcs(blk if c == 0.0 else blk + c * M, nodes)
            Einv = coarse_inverse(At, schur_spec.coarse) if schur_spec.coarse is not None else None
            cache[key] = (At, 1.0 / At.diagonal(), blk, Einv)
        At, dinv, _, Einv = cache[key]
        if schur_spec.coarse is not None:
            return coarse_chebyshev(At, dinv, rhs, schur_spec, Einv)
        if schur_spec.its == 0:
            return dinv * rhs
        return chebyshev_jacobi(At, dinv, rhs, schur_spec.emin, schur_spec.emax,
                                schur_spec.its, schur_spec.eimag)

    def pc_linear(u_0, u_1, b_0, b_1):
        # (1,1)-block, control.py:1997-2014
        b_0_help = apply_T_1_inv(b_0)
        for i in range(m):
            u_0[i] = _inner_solve(Mt, mass_spec, b_0_help[i].copy())
            u_0[i] *= 2.0 / tau
        u_0[:] = apply_T_2_inv(u_0)
        # b = T_2 (D_v u_0) - b_1, control.py:2016-2048
        b = np.zeros_like(u_0)
        b[0] = block_10[(0, 0)] @ u_0[0]
        _bc(b[0], nodes)
        for i in range(1, m):
            t = block_10[(i, i - 1)] @ u_0[i - 1]
            b[i] = block_10[(i, i)] @ u_0[i]
            b[i] += t
            _bc(b[i], nodes)
        b = apply_T_2(b)
        for i in range(m):
            b[i] -= b_1[i]
            _bc(b[i], nodes)
        # forward sweep, control.py:2050-2116
        b = apply_T_2_inv(b)
        u_1[0] = solve(block_10[(0, 0)], my_const, b[0])
        for i in range(1, m):
            b[i] -= block_10[(i, i - 1)] @ u_1[i - 1]
            b[i] -= cM @ u_1[i - 1]
            _bc(b[i], nodes)
            u_1[i] = solve(block_10[(i, i)], my_const, b[i])
        # control.py:2118-2133
        u_1[:] = apply_T_2(u_1)
        b = np.zeros_like(u_0)
        for i in range(m):
            b[i] = (M @ u_1[i]) * (0.5 * tau)
            _bc(b[i], nodes)
        # backward sweep, control.py:2135-2189
        u_1[m - 1] = solve(block_01[(m - 1, m - 1)], my_const, b[m - 1])
        for i in range(m - 2, -1, -1):
            b[i] -= upper(i) @ u_1[i + 1]
            _bc(b[i], nodes)
            u_1[i] = solve(block_01[(i, i)], my_const, b[i])
    return pc_linear


# ------------------------------------------------ incompressible (Stokes) control, stationary


def stationary_incompressible_blocks(M_v, D_v, B, beta):
    """Outer block system of ``Stationary.incompressible_linear_solve``
    (``control/control.py:896-919``): ``space_0`` blocks (v, zeta), ``space_1`` blocks (mu, p)."""
    D_zeta = sp.csr_matrix(D_v.T)
    B = sp.csr_matrix(B)
    B_T = sp.csr_matrix(B.T)
    b00 = {(0, 0): M_v, (0, 1): D_zeta, (1, 0): D_v, (1, 1): sp.csr_matrix((-1.0 / beta) * M_v)}
    b01 = {(0, 0): B_T, (0, 1): None, (1, 0): None, (1, 1): B_T}
    b10 = {(0, 0): B, (0, 1): None, (1, 0): None, (1, 1): B}
    b11 = {(0, 0): None, (0, 1): None, (1, 0): None, (1, 1): None}
    return b00, b01, b10, b11


def pc_stationary_incompressible(M_v, D_v, B, M_p, K_p, D_p, beta, nodes_v, mass_spec,
                                 schur_spec, kp_spec, mp_spec, inner_its=5):
    """``pc_fn`` of ``Stationary.incompressible_linear_solve`` (``control/control.py:986-1085``):
    ``inner_its`` GMRES iterations on the velocity KKT block with the stationary block-Schur
    preconditioner (``:993-1024``), then the pressure Schur complement
    ``M_p^-1 [[M_p, D_p^T], [D_p, -M_p/beta]] K_p^-1`` (``:1030-1082``).  ``kp_spec`` replaces the
    single BoomerAMG cycle on ``K_p`` (north_star), ``mp_spec`` is the reference's own
    20-step Jacobi-Chebyshev on ``M_p`` (``:957-971``).  ``D_p`` is the forward operator
    assembled on the pressure space (``:981``)."""
    D_zeta = sp.csr_matrix(D_v.T)
    nv = M_v.shape[0]
    ib00 = {(0, 0): M_v}
    ib01 = {(0, 0): D_zeta}
    ib10 = {(0, 0): D_v}
    ib11 = {(0, 0): sp.csr_matrix((-1.0 / beta) * M_v)}
    inner = OracleSystem(nv, nv, ib00, ib01, ib10, ib11,
                         nullspace_0=(DirichletBCNullspace(nodes_v),),
                         nullspace_1=(DirichletBCNullspace(nodes_v),))
    inner_pc = pc_stationary(M_v, D_v, D_zeta, beta, nodes_v, mass_spec, schur_spec)
    inner_sp = {"preconditioner": True, "linear_solver": "gmres",
                "maximum_iterations": inner_its, "relative_tolerance": 0.0,
                "absolute_tolerance": 0.0, "monitor_convergence": False}
    K_p = sp.csr_matrix(K_p)
    M_p = sp.csr_matrix(M_p)
    D_p = sp.csr_matrix(D_p)
    D_pT = sp.csr_matrix(D_p.T)

    def pc_fn(u_0, u_1, b_0, b_1):
        v = np.zeros(nv)
        z = np.zeros(nv)
        inner.solve(v[None, :], z[None, :], b_0[0:1], b_0[1:2], solver_parameters=inner_sp,
                    pc_fn=inner_pc)
        u_0[0], u_0[1] = v, z
        h0 = B @ v - b_1[0]
        h1 = B @ z - b_1[1]
        m0 = _inner_solve(K_p, kp_spec, h0)
        m1 = _inner_solve(K_p, kp_spec, h1)
        g0 = M_p @ m0 + D_pT @ m1
        g1 = D_p @ m0 + (-1.0 / beta) * (M_p @ m1)
        u_1[0] = _inner_solve(M_p, mp_spec, g0)
        u_1[1] = _inner_solve(M_p, mp_spec, g1)
    return pc_fn


# ---------------------------------------------- incompressible (Stokes) control, instationary


def pc_instationary_incompressible(M_v, inner_blocks, B, M_p, K_p, comm_blocks, n_t, tau, beta,
                                   nodes_v, mass_spec, schur_spec, kp_spec, mp_spec, CN=False,
                                   inner_its=5, epsilon=1.0e-3):
    """``pc_fn`` of ``Instationary.incompressible_linear_solve``: BE branch
    ``control/control.py:4515-4687``, CN branch ``:4318-4513``.

    ``inner_its`` GMRES iterations on the velocity KKT system with the instationary block-Schur
    preconditioner (``:4524-4553``), ``h = tau B u_0 - b_1`` scaled by ``1/tau^2``
    (``:4571-4601``; CN: ``T_2``/``T_1`` before the subtraction and their inverses after the
    scaling, ``:4407-4428``), ``K_p`` solve per block (``:4603-4619``), product with the
    pressure-space block system ``block_**_int_p`` (``:4625-4665``), ``M_p`` solve per block
    (``:4670-4684``)."""
    i00, i01, i10, i11 = inner_blocks
    c00, c01, c10, c11 = comm_blocks
    m = n_t - 1 if CN else n_t
    nv = M_v.shape[0]
    npr = M_p.shape[0]
    ns = tuple(DirichletBCNullspace(nodes_v) for _ in range(m))
    inner = OracleSystem(nv, nv, i00, i01, i10, i11, n_blocks_00=m, n_blocks_11=m,
                         nullspace_0=ns, nullspace_1=ns, CN=CN)
    if CN:
        inner_pc = pc_instationary_CN(M_v, i01, i10, n_t, tau, beta, nodes_v, mass_spec,
                                      schur_spec)
    else:
        inner_pc = pc_instationary_BE(M_v, i01, i10, n_t, tau, beta, nodes_v, mass_spec,
                                      schur_spec, epsilon=epsilon)
    inner_sp = {"preconditioner": True, "linear_solver": "gmres",
                "maximum_iterations": inner_its, "relative_tolerance": 0.0,
                "absolute_tolerance": 0.0, "monitor_convergence": False}
    B = sp.csr_matrix(B)
    K_p = sp.csr_matrix(K_p)
    M_p = sp.csr_matrix(M_p)

    def pc_fn(u_0, u_1, b_0, b_1):
        v = np.zeros((m, nv))
        z = np.zeros((m, nv))
        inner.solve(v, z, b_0[:m], b_0[m:], solver_parameters=inner_sp, pc_fn=inner_pc)
        u_0[:m], u_0[m:] = v, z
        h0 = np.stack([tau * (B @ v[i]) for i in range(m)])
        h1 = np.stack([tau * (B @ z[i]) for i in range(m)])
        if CN:
            h0, h1 = apply_T_2(h0), apply_T_1(h1)
        h0 = (h0 - b_1[:m]) * (1.0 / tau**2)
        h1 = (h1 - b_1[m:]) * (1.0 / tau**2)
        if CN:
            h0, h1 = apply_T_2_inv(h0), apply_T_1_inv(h1)
        m0 = np.stack([_inner_solve(K_p, kp_spec, h0[i]) for i in range(m)])
        m1 = np.stack([_inner_solve(K_p, kp_spec, h1[i]) for i in range(m)])
        g0 = np.zeros((m, npr))
        g1 = np.zeros((m, npr))
        for (i, j), A in c00.items():
            if A is not None:
                g0[i] += A @ m0[j]
        for (i, j), A in c01.items():
            if A is not None:
                g0[i] += A @ m1[j]
        for (i, j), A in c10.items():
            if A is not None:
                g1[i] += A @ m0[j]
        for (i, j), A in c11.items():
            if A is not None:
                g1[i] += A @ m1[j]
        for i in range(m):
            u_1[i] = _inner_solve(M_p, mp_spec, g0[i])
            u_1[m + i] = _inner_solve(M_p, mp_spec, g1[i])
    return pc_fn
