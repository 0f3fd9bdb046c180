"""Block-dict construction for the all-at-once KKT system (host side).

Mirrors which ``(i, j)`` blocks exist and their coefficients in
``Control.Instationary.linear_solve`` (``control/control.py:2889-2978``) and
``Control.Stationary.linear_solve`` (``control/control.py:547-554``), with SciPy CSR
matrices in place of UFL forms.  Dicts hold **all** ``n_row * n_col`` keys with
``None`` for structural zeros, as ``MultiBlockSystem`` requires
(``preconditioner/preconditioner.py:243-258``).

Blocks that are the same Python object share storage on the device ("mode S");
pass ``share=False`` to give every ``(i, j)`` block its own copy of the values
("mode G", what the reference stores).
"""
from __future__ import annotations

from typing import Sequence

import scipy.sparse as sp

__all__ = ["instationary_blocks", "stationary_blocks", "instationary_incompressible_blocks",
           "stationary_incompressible_blocks", "conform_to"]


def _csr(A):
    A = sp.csr_matrix(A)
    A.sort_indices()
    return A


def _axpby(a, A, b, B):
    """``a A + b B``; matrices with identical index arrays are combined entry by entry, so
    structural zeros survive and every block assembled over one connectivity keeps one
    sparsity structure (what ``kkt_update_block_values`` relies on)."""
    import numpy as np
    if (A.shape == B.shape and A.nnz == B.nnz and np.array_equal(A.indptr, B.indptr)
            and np.array_equal(A.indices, B.indices)):
        # index arrays are shared with A (blocks are never modified in place)
        return sp.csr_matrix((a * A.data + b * B.data, A.indices, A.indptr), shape=A.shape)
    return _csr(a * A + b * B)


def conform_to(A, like):
    """``A`` on the sparsity structure of ``like`` (explicit zeros where ``A`` has no entry).
    SciPy arithmetic drops entries that cancel or are zero, so a forward operator built as
    ``nu * K + N`` usually has fewer stored entries than the mass matrix; the block-Schur
    preconditioner forms ``D + c M`` entry by entry and needs one structure for both.
    Raises if ``A`` has an entry outside the structure of ``like``."""
    import numpy as np
    A, like = _csr(A), _csr(like)
    if A.shape != like.shape:
        raise ValueError("conform_to: shapes differ")
    if (A.nnz == like.nnz and np.array_equal(A.indptr, like.indptr)
            and np.array_equal(A.indices, like.indices)):
        return A
    rows = np.repeat(np.arange(A.shape[0]), np.diff(A.indptr))
    # position of every entry of A inside the row of `like` it belongs to
    key_like = like.indptr[:-1].astype(np.int64)
    lo = key_like[rows]
    hi = like.indptr[1:].astype(np.int64)[rows]
    pos = lo.copy()
    width = int(np.diff(like.indptr).max()) if like.nnz else 0
    step = 1
    while step < max(width, 1):
        step <<= 1
    off = np.zeros(len(rows), dtype=np.int64)
    idx = like.indices
    while step:
        cand = off + step
        ok = (lo + cand < hi) & (idx[np.minimum(lo + cand, len(idx) - 1)] <= A.indices)
        off = np.where(ok, cand, off)
        step >>= 1
    pos = lo + off
    if len(rows) and not np.array_equal(idx[pos], A.indices):
        raise ValueError("conform_to: the matrix has entries outside the target structure")
    data = np.zeros(like.nnz)
    data[pos] = A.data
    return sp.csr_matrix((data, like.indices, like.indptr), shape=like.shape)


_TRANSPOSE_PERM = {}


def _transpose(A):
    """``A^T`` as CSR.  For matrices with a symmetric sparsity structure (every FE matrix over
    one connectivity) the transposed values are a fixed permutation of the values, cached per
    structure: re-linearisation loops transpose hundreds of same-structure matrices."""
    import numpy as np
    A = _csr(A)
    if A.shape[0] != A.shape[1]:
        return _csr(A.T)
    key = (A.shape, A.nnz, hash(A.indptr.tobytes()))
    ent = _TRANSPOSE_PERM.get(key)
    if ent is None:
        T = _csr(sp.csr_matrix((np.arange(1, A.nnz + 1, dtype=np.float64), A.indices,
                                A.indptr), shape=A.shape).T)
        same = (T.nnz == A.nnz and np.array_equal(T.indptr, A.indptr)
                and np.array_equal(T.indices, A.indices))
        ent = (T.data.astype(np.int64) - 1) if same else False
        if len(_TRANSPOSE_PERM) > 16:
            _TRANSPOSE_PERM.clear()
        _TRANSPOSE_PERM[key] = ent
    if ent is False:
        return _csr(A.T)
    return sp.csr_matrix((A.data[ent], A.indices, A.indptr), shape=A.shape)


def _own(A, share):
    return A if share else A.copy()


def stationary_blocks(M, K, beta):
    """``control/control.py:547-554``: ``[[M, K^T], [K, -(1/beta) M]]``."""
    M = _csr(M)
    K = _csr(K)
    return ({(0, 0): M}, {(0, 0): _csr(K.T)}, {(0, 0): K},
            {(0, 0): _csr((-1.0 / beta) * M)})


def stationary_incompressible_blocks(M_v, D_v, B, beta):
    """Outer block system of ``Stationary.incompressible_linear_solve``
    (``control/control.py:896-919``): velocity-space blocks (v, zeta), pressure-space blocks
    (mu, p)."""
    M_v, D_v, B = _csr(M_v), _csr(D_v), _csr(B)
    B_T = _csr(B.T)
    b00 = {(0, 0): M_v, (0, 1): _transpose(D_v), (1, 0): D_v,
           (1, 1): _csr((-1.0 / beta) * M_v)}
    b01 = {(0, 0): B_T, (0, 1): None, (1, 0): None, (1, 1): B_T}
    b10 = {(0, 0): B, (0, 1): None, (1, 0): None, (1, 1): B}
    b11 = {(0, 0): None, (0, 1): None, (1, 0): None, (1, 1): None}
    return b00, b01, b10, b11


def instationary_blocks(M, K: Sequence, tau: float, beta: float, n_t: int,
                        CN: bool, *, share: bool = True):
    """Space-time KKT blocks.

    ``K[i]`` is the forward-operator matrix at time level ``i`` (``D_v_i`` of
    ``control/control.py:2903``); pass one matrix repeated for a time-invariant
    operator.  Returns ``(block_00, block_01, block_10, block_11, m)`` with ``m`` the
    number of blocks per variable (``n_t`` for BE, ``n_t - 1`` for CN).
    """
    M = _csr(M)
    if not isinstance(K, (list, tuple)):
        K = [K] * n_t
    if len(K) != n_t:
        raise ValueError("need one forward-operator matrix per time level")
    # one CSR object per distinct input matrix: a time-invariant operator (one matrix repeated)
    # then gives one block object per coefficient pair, and the device shares its values
    uniq = {}
    K = [uniq.setdefault(id(k), _csr(k)) for k in K]
    cache = {}

    def comb(a, i, b, transpose=False):
        """a * K_i (or its transpose) + b * M, cached so equal blocks are one object."""
        key = (a, id(K[i]), b, transpose)
        if share and key in cache:
            return cache[key]
        Ki = _transpose(K[i]) if transpose else K[i]
        A = _axpby(a, Ki, b, M)
        cache[key] = A
        return A

    def mass(c):
        key = ("M", c)
        if share and key in cache:
            return cache[key]
        A = _csr(c * M)
        cache[key] = A
        return A

    if not CN:
        m = n_t
        b00 = {(i, j): None for i in range(m) for j in range(m)}
        b01 = dict(b00)
        b10 = dict(b00)
        b11 = dict(b00)
        for i in range(n_t):
            # control/control.py:2907-2928 (rows 0..n_t-2) and 2960-2978 (last row)
            if i < n_t - 1:
                b00[(i, i)] = mass(tau)
                b01[(i, i + 1)] = mass(-1.0)
            b01[(i, i)] = comb(tau, i, 1.0, transpose=True)
            b10[(i, i)] = comb(tau, i, 1.0)
            if i >= 1:
                b10[(i, i - 1)] = mass(-1.0)
                b11[(i, i)] = mass(-tau / beta)
        return b00, b01, b10, b11, m

    m = n_t - 1
    h = 0.5 * tau
    b00 = {(i, j): None for i in range(m) for j in range(m)}
    b01 = dict(b00)
    b10 = dict(b00)
    b11 = dict(b00)
    for i in range(m):
        # control/control.py:2938-2958; D_v_i at level i, D_v_i_plus at level i+1
        if i >= 1:
            b00[(i, i - 1)] = mass(h)
            b10[(i, i - 1)] = comb(h, i, -1.0)
        b00[(i, i)] = mass(h)
        b01[(i, i)] = comb(h, i, 1.0, transpose=True)
        b10[(i, i)] = comb(h, i + 1, 1.0)
        b11[(i, i)] = mass(-h / beta)
        if i + 1 < m:
            b01[(i, i + 1)] = comb(h, i + 1, -1.0, transpose=True)
            b11[(i, i + 1)] = mass(-h / beta)
    return b00, b01, b10, b11, m


def instationary_incompressible_blocks(M_v, K_v, B, M_p, K_p, tau: float, beta: float,
                                       n_t: int, CN: bool, *, share: bool = True):
    """Block dicts of ``Instationary.incompressible_linear_solve``
    (``control/control.py:3750-3957``) for Stokes control.

    Returns a dict with

    * ``outer``: ``(block_00, block_01, block_10, block_11)`` over ``2m`` velocity-space and
      ``2m`` pressure-space blocks -- ``block_00`` is the velocity KKT system flattened to
      ``[v_0..v_{m-1}, zeta_0..zeta_{m-1}]`` (``:3793-3829`` BE, ``:3840-3895`` CN),
      ``block_01/10[(i, i)] = tau B^T / tau B`` (``:3750-3770``), ``block_11`` empty;
    * ``inner``: the velocity KKT dicts ``block_**_int`` (the heat-type system on ``M_v, K_v``);
    * ``commutator``: the pressure-space dicts ``block_**_int_p`` (the same construction on
      ``M_p, K_p``);
    * ``m``: blocks per variable (``n_t`` BE, ``n_t - 1`` CN).

    ``K_v`` / ``K_p`` are the forward operator assembled on the velocity / pressure space
    (``construct_D_v`` at ``:3779-3785``), one matrix or one per time level.
    """
    i00, i01, i10, i11, m = instationary_blocks(M_v, K_v, tau, beta, n_t, CN, share=share)
    c00, c01, c10, c11, _ = instationary_blocks(M_p, K_p, tau, beta, n_t, CN, share=share)
    n = 2 * m
    b00 = {(i, j): None for i in range(n) for j in range(n)}
    for (i, j), A in i00.items():
        b00[(i, j)] = A
    for (i, j), A in i01.items():
        b00[(i, m + j)] = A
    for (i, j), A in i10.items():
        b00[(m + i, j)] = A
    for (i, j), A in i11.items():
        b00[(m + i, m + j)] = A
    tB = _csr(tau * sp.csr_matrix(B))
    tBT = _csr(tB.T)
    b01 = {(i, j): None for i in range(n) for j in range(n)}
    b10 = dict(b01)
    b11 = dict(b01)
    for i in range(n):
        b01[(i, i)] = _own(tBT, share)
        b10[(i, i)] = _own(tB, share)
    return {"outer": (b00, b01, b10, b11), "inner": (i00, i01, i10, i11),
            "commutator": (c00, c01, c10, c11), "m": m}
