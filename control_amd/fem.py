"""Host-side finite-element input generation (NumPy/SciPy, no Firedrake).

The reference gets its spatial blocks from Firedrake ``assemble`` of UFL forms
(``preconditioner/preconditioner.py:305-328``).  Firedrake is not available in
this pipeline, so synthetic systems of the BASELINE.json shapes are produced
here: the mass matrix ``M = int u w`` and the stiffness matrix
``K = int grad u . grad w`` (the README's ``forw_diff_operator``,
``README.md:31-32``) on structured meshes, as SciPy CSR with sorted int32
column indices -- exactly what ``petscmat.getValuesCSR()`` would hand over.

This module is input generation only.  It is shared by the product host layer
(bench / smoke build their synthetic systems with it) and by the tests; it holds
no solver arithmetic.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import scipy.sparse as sp

__all__ = [
    "SpatialDiscretisation",
    "unit_square_p1",
    "unit_cube_p1",
    "unit_square_q2",
    "rectangle_p1",
]


@dataclass
class SpatialDiscretisation:
    """One spatial function space: matrices, boundary dofs and dof coordinates."""

    M: sp.csr_matrix          # mass matrix
    K: sp.csr_matrix          # stiffness matrix (grad-grad)
    coords: np.ndarray        # (N_x, dim) dof coordinates
    boundary: np.ndarray      # int32 sorted dof indices on the whole boundary
    name: str = ""
    cells: np.ndarray = None  # (n_cells, 3) vertex indices (P1 triangles only)

    @property
    def n_dofs(self) -> int:
        return self.M.shape[0]

    def weighted_mass(self, w) -> sp.csr_matrix:
        """``w(X) * inner(trial, test) * dx`` for a coefficient given by its values at
        quadrature points: ``w(lam)`` receives the barycentric coordinates ``(nq, 3)`` of
        Radon's 7-point rule and the cell connectivity and returns ``(n_cells, nq)`` values.
        P1 triangles; same structure as ``M``."""
        if self.cells is None:
            raise NotImplementedError("weighted_mass needs a P1 triangle discretisation")
        s15 = np.sqrt(15.0)
        a1, a2 = (6.0 - s15) / 21.0, (6.0 + s15) / 21.0
        w1, w2 = (155.0 - s15) / 1200.0, (155.0 + s15) / 1200.0
        lam = np.array([[1 / 3, 1 / 3, 1 / 3],
                        [a1, a1, 1 - 2 * a1], [a1, 1 - 2 * a1, a1], [1 - 2 * a1, a1, a1],
                        [a2, a2, 1 - 2 * a2], [a2, 1 - 2 * a2, a2], [1 - 2 * a2, a2, a2]])
        wq = np.array([9.0 / 40.0, w1, w1, w1, w2, w2, w2])
        X = self.coords[self.cells]
        d1, d2 = X[:, 1] - X[:, 0], X[:, 2] - X[:, 0]
        area = 0.5 * np.abs(d1[:, 0] * d2[:, 1] - d1[:, 1] * d2[:, 0])
        W = np.asarray(w(lam, self.cells)) * wq[None, :] * area[:, None]
        Me = np.einsum("cq,qa,qb->cab", W, lam, lam)
        return _assemble_like(Me, self.cells, self.cells, self.M.shape)


def _p1_convection(self, wind) -> sp.csr_matrix:
    """``inner(dot(grad(trial), wind), test) * dx`` for P1 triangles: ``wind(X) -> (n, 2)`` is
    evaluated at the points of Radon's 7-point rule (degree 5); same structure as ``M``."""
    if self.cells is None:
        raise NotImplementedError("convection needs a P1 triangle discretisation")
    s15 = np.sqrt(15.0)
    a1, a2 = (6.0 - s15) / 21.0, (6.0 + s15) / 21.0
    w1, w2 = (155.0 - s15) / 1200.0, (155.0 + s15) / 1200.0
    lam = np.array([[1 / 3, 1 / 3, 1 / 3],
                    [a1, a1, 1 - 2 * a1], [a1, 1 - 2 * a1, a1], [1 - 2 * a1, a1, a1],
                    [a2, a2, 1 - 2 * a2], [a2, 1 - 2 * a2, a2], [1 - 2 * a2, a2, a2]])
    wq = np.array([9.0 / 40.0, w1, w1, w1, w2, w2, w2])
    X = self.coords[self.cells]                                   # (nc, 3, 2)
    A = np.concatenate([np.ones((len(X), 3, 1)), X], axis=2)
    grads = np.transpose(np.linalg.inv(A)[:, 1:, :], (0, 2, 1))   # (nc, 3, 2): grad of phi_b
    d1, d2 = X[:, 1] - X[:, 0], X[:, 2] - X[:, 0]
    area = 0.5 * np.abs(d1[:, 0] * d2[:, 1] - d1[:, 1] * d2[:, 0])
    Xq = np.einsum("qa,cad->cqd", lam, X)                         # quadrature points
    W = np.asarray(wind(Xq.reshape(-1, 2))).reshape(len(X), len(wq), 2)
    wg = np.einsum("cqd,cbd->cqb", W, grads)                      # wind . grad(phi_b)
    Ce = np.einsum("cq,qa,cqb->cab", wq[None, :] * area[:, None], lam, wg)
    return _assemble_like(Ce, self.cells, self.cells, self.M.shape)


SpatialDiscretisation.convection = _p1_convection


def _canonical_csr(A: sp.spmatrix) -> sp.csr_matrix:
    A = sp.csr_matrix(A)
    A.sum_duplicates()
    A.sort_indices()
    A.indptr = A.indptr.astype(np.int32)
    A.indices = A.indices.astype(np.int32)
    A.data = np.ascontiguousarray(A.data, dtype=np.float64)
    return A


def _p1_simplex_assemble(coords: np.ndarray, cells: np.ndarray):
    """Mass and stiffness for P1 on simplices (dim 2 or 3), fully vectorised."""
    n_nodes, dim = coords.shape
    nv = dim + 1
    X = coords[cells]                                   # (nc, nv, dim)
    # barycentric gradients: rows of inv([1 x]) without the constant row
    A = np.concatenate([np.ones((len(cells), nv, 1)), X], axis=2)  # (nc,nv,nv)
    Ainv = np.linalg.inv(A)                             # columns = coefficients
    grads = np.transpose(Ainv[:, 1:, :], (0, 2, 1))     # (nc, nv, dim)
    fact = {2: 2.0, 3: 6.0}[dim]
    vol = np.abs(np.linalg.det(A)) / fact               # (nc,)
    Ke = np.einsum("cid,cjd->cij", grads, grads) * vol[:, None, None]
    Mref = (np.ones((nv, nv)) + np.eye(nv)) / ((dim + 1.0) * (dim + 2.0))
    Me = vol[:, None, None] * Mref[None, :, :]
    rows = np.repeat(cells, nv, axis=1).ravel()
    cols = np.tile(cells, (1, nv)).ravel()
    M = sp.coo_matrix((Me.ravel(), (rows, cols)), shape=(n_nodes, n_nodes))
    K = sp.coo_matrix((Ke.ravel(), (rows, cols)), shape=(n_nodes, n_nodes))
    return _canonical_csr(M), _canonical_csr(K)


def rectangle_p1(nx: int, ny: int, lx: float = 1.0, ly: float = 1.0,
                 diagonal: str = "right") -> SpatialDiscretisation:
    """P1 on a structured triangulation of ``[0,lx] x [0,ly]``.

    Node ``(i, j)`` has index ``j * (nx + 1) + i``.  ``diagonal="right"`` cuts every
    cell from its lower-left to its upper-right corner.
    """
    xs = np.linspace(0.0, lx, nx + 1)
    ys = np.linspace(0.0, ly, ny + 1)
    XX, YY = np.meshgrid(xs, ys, indexing="xy")
    coords = np.stack([XX.ravel(), YY.ravel()], axis=1)
    ii, jj = np.meshgrid(np.arange(nx), np.arange(ny), indexing="xy")
    n00 = (jj * (nx + 1) + ii).ravel()
    n10 = n00 + 1
    n01 = n00 + (nx + 1)
    n11 = n01 + 1
    if diagonal == "right":
        cells = np.concatenate([np.stack([n00, n10, n11], 1),
                                np.stack([n00, n11, n01], 1)], 0)
    elif diagonal == "left":
        cells = np.concatenate([np.stack([n00, n10, n01], 1),
                                np.stack([n10, n11, n01], 1)], 0)
    else:
        raise ValueError("diagonal must be 'right' or 'left'")
    M, K = _p1_simplex_assemble(coords, cells)
    onb = ((coords[:, 0] == xs[0]) | (coords[:, 0] == xs[-1])
           | (coords[:, 1] == ys[0]) | (coords[:, 1] == ys[-1]))
    return SpatialDiscretisation(M, K, coords,
                                 np.flatnonzero(onb).astype(np.int32),
                                 f"P1 {nx}x{ny}", cells)


def unit_square_p1(n: int, diagonal: str = "right") -> SpatialDiscretisation:
    """``UnitSquareMesh(n, n)`` with ``FunctionSpace(mesh, "Lagrange", 1)``."""
    return rectangle_p1(n, n, 1.0, 1.0, diagonal)


def unit_cube_p1(n: int) -> SpatialDiscretisation:
    """P1 on the 6-tetrahedra-per-cube (Kuhn) subdivision of the unit cube."""
    xs = np.linspace(0.0, 1.0, n + 1)
    ZZ, YY, XX = np.meshgrid(xs, xs, xs, indexing="ij")
    coords = np.stack([XX.ravel(), YY.ravel(), ZZ.ravel()], axis=1)
    np1 = n + 1

    def nid(i, j, k):
        return (k * np1 + j) * np1 + i

    kk, jj, ii = np.meshgrid(np.arange(n), np.arange(n), np.arange(n),
                             indexing="ij")
    ii, jj, kk = ii.ravel(), jj.ravel(), kk.ravel()
    cells = []
    # Kuhn: one tetrahedron per permutation of the axes, all sharing the main diagonal
    import itertools
    for perm in itertools.permutations(range(3)):
        off = np.zeros(3, dtype=np.int64)
        verts = [nid(ii, jj, kk)]
        for ax in perm:
            off = off.copy()
            off[ax] = 1
            verts.append(nid(ii + off[0], jj + off[1], kk + off[2]))
        cells.append(np.stack(verts, 1))
    cells = np.concatenate(cells, 0)
    M, K = _p1_simplex_assemble(coords, cells)
    onb = np.zeros(len(coords), dtype=bool)
    for d in range(3):
        onb |= (coords[:, d] == 0.0) | (coords[:, d] == 1.0)
    return SpatialDiscretisation(M, K, coords,
                                 np.flatnonzero(onb).astype(np.int32),
                                 f"P1 {n}^3")


def _p2_1d(n: int, length: float):
    """1-D quadratic Lagrange mass and stiffness on ``n`` elements (2n+1 nodes)."""
    h = length / n
    Me = h / 30.0 * np.array([[4.0, 2.0, -1.0], [2.0, 16.0, 2.0], [-1.0, 2.0, 4.0]])
    Ke = 1.0 / (3.0 * h) * np.array([[7.0, -8.0, 1.0], [-8.0, 16.0, -8.0],
                                     [1.0, -8.0, 7.0]])
    nn = 2 * n + 1
    M = sp.lil_matrix((nn, nn))
    K = sp.lil_matrix((nn, nn))
    for e in range(n):
        idx = np.array([2 * e, 2 * e + 1, 2 * e + 2])
        M[np.ix_(idx, idx)] += Me
        K[np.ix_(idx, idx)] += Ke
    return sp.csr_matrix(M), sp.csr_matrix(K), np.linspace(0.0, length, nn)


def unit_square_q2(n: int, length: float = 1.0) -> SpatialDiscretisation:
    """Q2 on ``UnitSquareMesh(n, n, quadrilateral=True)`` (tensor-product form); ``length``
    scales the square (``RectangleMesh(n, n, length, length, quadrilateral=True)``).

    This is the space of the reference's known-answer tests
    (``test/test_control.py:1244-1247``).  Dof ``(i, j)`` of the ``(2n+1)^2`` grid
    has index ``j * (2n + 1) + i``.
    """
    M1, K1, xs = _p2_1d(n, length)
    M = sp.kron(M1, M1)
    K = sp.kron(M1, K1) + sp.kron(K1, M1)
    XX, YY = np.meshgrid(xs, xs, indexing="xy")
    coords = np.stack([XX.ravel(), YY.ravel()], axis=1)
    onb = ((coords[:, 0] == 0.0) | (coords[:, 0] == xs[-1])
           | (coords[:, 1] == 0.0) | (coords[:, 1] == xs[-1]))
    return SpatialDiscretisation(_canonical_csr(M), _canonical_csr(K), coords,
                                 np.flatnonzero(onb).astype(np.int32),
                                 f"Q2 {n}x{n}")


# --------------------------------------------------------------- Taylor-Hood (Q2-Q1)

def _gauss(npts=4):
    x, w = np.polynomial.legendre.leggauss(npts)
    return 0.5 * (x + 1.0), 0.5 * w          # on [0, 1]


def _p1p2_1d(n: int, length: float):
    """1-D mixed matrices between P1 (n+1 nodes) test and P2 (2n+1 nodes) trial functions:
    ``MX[i, j] = int psi_i phi_j`` and ``DX[i, j] = int psi_i phi_j'``; plus P1 mass/stiffness."""
    h = length / n
    xq, wq = _gauss(4)
    psi = np.stack([1.0 - xq, xq])                                    # P1 on [0,1]
    phi = np.stack([2 * (xq - 0.5) * (xq - 1.0), 4 * xq * (1.0 - xq), 2 * xq * (xq - 0.5)])
    dphi = np.stack([4 * xq - 3.0, 4.0 - 8 * xq, 4 * xq - 1.0])       # d/d(xi)
    MXe = h * np.einsum("iq,jq,q->ij", psi, phi, wq)
    DXe = np.einsum("iq,jq,q->ij", psi, dphi, wq)                     # h * (1/h)
    M1e = h * np.einsum("iq,jq,q->ij", psi, psi, wq)
    K1e = (1.0 / h) * np.array([[1.0, -1.0], [-1.0, 1.0]])
    MX = sp.lil_matrix((n + 1, 2 * n + 1))
    DX = sp.lil_matrix((n + 1, 2 * n + 1))
    M1 = sp.lil_matrix((n + 1, n + 1))
    K1 = sp.lil_matrix((n + 1, n + 1))
    for e in range(n):
        ip = np.array([e, e + 1])
        iv = np.array([2 * e, 2 * e + 1, 2 * e + 2])
        MX[np.ix_(ip, iv)] += MXe
        DX[np.ix_(ip, iv)] += DXe
        M1[np.ix_(ip, ip)] += M1e
        K1[np.ix_(ip, ip)] += K1e
    return sp.csr_matrix(MX), sp.csr_matrix(DX), sp.csr_matrix(M1), sp.csr_matrix(K1)


@dataclass
class TaylorHoodDiscretisation:
    """Q2 velocity (two components, component-major dof order) and Q1 pressure on
    ``UnitSquareMesh(n, n, quadrilateral=True)`` -- the spaces of the reference's
    incompressible known-answer test (``test/test_control.py:232-237``)."""
    M_v: sp.csr_matrix        # vector mass
    K_v: sp.csr_matrix        # vector grad-grad
    B: sp.csr_matrix          # -(div v, q): shape (n_p, n_v)
    M_p: sp.csr_matrix
    K_p: sp.csr_matrix
    coords_v: np.ndarray      # (n_v / 2, 2) node coordinates of one component
    coords_p: np.ndarray
    boundary_v: np.ndarray    # Dirichlet dofs of the vector space (both components)
    elem: dict = None         # element data for re-assembly (P2-P1 triangles only)

    @property
    def n_v(self):
        return self.M_v.shape[0]

    def convection_v(self, w: np.ndarray) -> sp.csr_matrix:
        """``inner(dot(grad(trial), w), test) * dx`` on the velocity space for a P2 vector
        field ``w`` (component-major), the Picard linearisation of the Navier-Stokes
        convection term (``test/test_control.py:4194-4199``).  Same sparsity structure as
        ``M_v`` / ``K_v`` entry by entry, so it can be re-uploaded with
        ``kkt_update_block_values``."""
        data = self.convection_v_data(w)
        return sp.csr_matrix((data, self.K_v.indices.copy(), self.K_v.indptr.copy()),
                             shape=self.K_v.shape)

    def convection_v_data(self, w: np.ndarray) -> np.ndarray:
        """The values of ``convection_v(w)`` on the structure of ``K_v`` (no sparse-matrix
        construction: element entries are summed through a cached scatter map)."""
        e = self._need_elem()
        n2 = self.n_v // 2
        V = e["V"]
        wq = np.stack([w[:n2][V] @ e["phi"].T, w[n2:][V] @ e["phi"].T], axis=2)  # (ne, nq, 2)
        adv = np.matmul(e["gphi"], wq[..., None])[..., 0]              # (w . grad phi_b)
        Ne = np.einsum("eq,qa,eqb->eab", e["W"], e["phi"], adv)
        if "scatter_v" not in e:
            # CSR position of every element entry in the scalar P2 structure (one component)
            K2 = self.K_v[:n2, :n2].tocsr()
            K2.sort_indices()
            rows = np.repeat(V, V.shape[1], axis=1).ravel()
            cols = np.tile(V, (1, V.shape[1])).ravel()
            key = K2.indptr[rows].astype(np.int64)
            # position of `cols` inside each row's sorted index list
            pos = np.empty(len(rows), dtype=np.int64)
            for r0 in range(0, len(rows), 1 << 20):
                sl = slice(r0, r0 + (1 << 20))
                pos[sl] = [0] * 0 or _row_positions(K2, rows[sl], cols[sl])
            e["scatter_v"] = key + pos
            e["nnz2"] = K2.nnz
        d2 = np.bincount(e["scatter_v"], weights=Ne.ravel(), minlength=e["nnz2"])
        return np.concatenate([d2, d2])

    def convection_p(self, w: np.ndarray) -> sp.csr_matrix:
        """The same form on the pressure space (``construct_D_v(p_trial, p_test, ...)``,
        ``control/control.py:3783-3785``): structure of ``M_p`` / ``K_p``."""
        e = self._need_elem()
        n2 = self.n_v // 2
        V, Pn = e["V"], e["P"]
        wq = np.stack([e["phi"] @ w[:n2][V].T, e["phi"] @ w[n2:][V].T], axis=2)
        adv = np.einsum("qed,ecd->eqc", wq, e["glam"])                 # (w . grad lambda_c)
        Ne = np.einsum("eq,qa,eqc->eac", e["W"], e["lam"], adv)
        return _assemble_like(Ne, Pn, Pn, (self.n_p, self.n_p))

    def _need_elem(self):
        if self.elem is None:
            raise NotImplementedError("re-assembly is implemented for rectangle_p2p1 only")
        return self.elem

    @property
    def n_p(self):
        return self.M_p.shape[0]


def _row_positions(A, rows, cols):
    """Index of ``cols[k]`` inside row ``rows[k]`` of the CSR matrix ``A`` (sorted indices)."""
    out = np.empty(len(rows), dtype=np.int64)
    # vectorised binary search per entry inside its row segment
    lo = A.indptr[rows].astype(np.int64)
    hi = A.indptr[rows + 1].astype(np.int64)
    idx = A.indices
    base = lo.copy()
    n = (hi - lo).max()
    step = 1
    while step < n:
        step <<= 1
    pos = np.zeros(len(rows), dtype=np.int64)
    while step:
        cand = pos + step
        ok = (cand < hi - base) & (idx[np.minimum(base + cand, len(idx) - 1)] <= cols)
        pos = np.where(ok, cand, pos)
        step >>= 1
    out[:] = pos
    assert np.array_equal(idx[base + pos], cols)
    return out


def _assemble_like(Ee, rows, cols, shape):
    """Sum element matrices ``Ee[e, a, b]`` into CSR; entries are ordered by (row, column)
    whatever their values, so every matrix assembled over the same connectivity has the
    same ``indptr`` / ``indices``."""
    r = np.repeat(rows, cols.shape[1], axis=1).ravel()
    c = np.tile(cols, (1, rows.shape[1])).ravel()
    return _canonical_csr(sp.coo_matrix((Ee.ravel(), (r, c)), shape=shape))


def unit_square_q2q1(n: int, length: float = 1.0) -> TaylorHoodDiscretisation:
    sd = unit_square_q2(n, length)
    M2, K2 = sd.M, sd.K
    I2 = sp.identity(2, format="csr")
    M_v = _canonical_csr(sp.kron(I2, M2))
    K_v = _canonical_csr(sp.kron(I2, K2))
    MX, DX, M1, K1 = _p1p2_1d(n, length)
    # dof (i, j) -> j * n_nodes_x + i  (y index major), as in unit_square_q2
    Bx = sp.kron(MX, DX)          # int q  d(v_x)/dx : (y: psi*phi) x (x: psi*phi')
    By = sp.kron(DX, MX)          # int q  d(v_y)/dy
    B = _canonical_csr(-sp.hstack([Bx, By]))
    M_p = _canonical_csr(sp.kron(M1, M1))
    K_p = _canonical_csr(sp.kron(M1, K1) + sp.kron(K1, M1))
    xs = np.linspace(0.0, length, n + 1)
    XX, YY = np.meshgrid(xs, xs, indexing="xy")
    coords_p = np.stack([XX.ravel(), YY.ravel()], axis=1)
    nb = sd.boundary
    boundary_v = np.concatenate([nb, nb + sd.n_dofs]).astype(np.int32)
    return TaylorHoodDiscretisation(M_v, K_v, B, M_p, K_p, sd.coords, coords_p, boundary_v)


# ------------------------------------------------------- Taylor-Hood (P2-P1, triangles)

def rectangle_p2p1(nx: int, ny: int, lx: float = 1.0, ly: float = 1.0) -> TaylorHoodDiscretisation:
    """P2 velocity / P1 pressure on the right-diagonal triangulation of ``[0,lx] x [0,ly]``
    (``RectangleMesh(nx, ny, lx, ly)``, the spaces of the reference's Stokes-control tests,
    ``test/test_control.py:3546-3560``).  P2 dofs are the points of the ``(2nx+1) x (2ny+1)``
    grid, P1 dofs the mesh vertices, both in lexicographic (y-major) order; the velocity
    vector is component-major."""
    nvx, nvy = 2 * nx + 1, 2 * ny + 1
    npx = nx + 1

    def vid(i, j):          # P2 grid index
        return j * nvx + i

    def pid(i, j):
        return j * npx + i
    ii, jj = np.meshgrid(np.arange(nx), np.arange(ny), indexing="xy")
    ii, jj = ii.ravel(), jj.ravel()
    hx, hy = lx / nx, ly / ny
    # two triangles per cell: (00, 10, 11) and (00, 11, 01); local P2 order:
    # vertices 0,1,2 then midpoints of edges (0,1), (1,2), (0,2)
    tri_v, tri_p, tri_x = [], [], []
    for (a, b, c) in (((0, 0), (1, 0), (1, 1)), ((0, 0), (1, 1), (0, 1))):
        verts = [a, b, c]
        mids = [((a[0] + b[0]), (a[1] + b[1])), ((b[0] + c[0]), (b[1] + c[1])),
                ((a[0] + c[0]), (a[1] + c[1]))]
        v_idx = [vid(2 * ii + 2 * v[0], 2 * jj + 2 * v[1]) for v in verts] + \
                [vid(2 * ii + m[0], 2 * jj + m[1]) for m in mids]
        p_idx = [pid(ii + v[0], jj + v[1]) for v in verts]
        xy = [np.stack([(ii + v[0]) * hx, (jj + v[1]) * hy], 1) for v in verts]
        tri_v.append(np.stack(v_idx, 1))
        tri_p.append(np.stack(p_idx, 1))
        tri_x.append(np.stack(xy, 1))
    V = np.concatenate(tri_v, 0)            # (ne, 6)
    Pn = np.concatenate(tri_p, 0)           # (ne, 3)
    X = np.concatenate(tri_x, 0)            # (ne, 3, 2)
    ne = len(V)
    # degree-5 quadrature on the reference triangle (Radon's 7 points), barycentric: exact
    # for the mass matrix (degree 4) and for the convection form with a P2 field (degree 5)
    s15 = np.sqrt(15.0)
    a1, a2 = (6.0 - s15) / 21.0, (6.0 + s15) / 21.0
    w1, w2 = (155.0 - s15) / 1200.0, (155.0 + s15) / 1200.0
    lam = np.array([[1 / 3, 1 / 3, 1 / 3],
                    [a1, a1, 1 - 2 * a1], [a1, 1 - 2 * a1, a1], [1 - 2 * a1, a1, a1],
                    [a2, a2, 1 - 2 * a2], [a2, 1 - 2 * a2, a2], [1 - 2 * a2, a2, a2]])
    wq = 0.5 * np.array([9.0 / 40.0, w1, w1, w1, w2, w2, w2])
    # gradients of barycentric coordinates per element
    A = np.concatenate([np.ones((ne, 3, 1)), X], axis=2)
    Ainv = np.linalg.inv(A)
    glam = np.transpose(Ainv[:, 1:, :], (0, 2, 1))          # (ne, 3, 2)
    detJ = np.abs(np.linalg.det(A))                         # 2 * area
    l = lam                                                 # (nq, 3)
    phi = np.stack([l[:, 0] * (2 * l[:, 0] - 1), l[:, 1] * (2 * l[:, 1] - 1),
                    l[:, 2] * (2 * l[:, 2] - 1), 4 * l[:, 0] * l[:, 1],
                    4 * l[:, 1] * l[:, 2], 4 * l[:, 0] * l[:, 2]], 1)     # (nq, 6)
    # d phi / d lambda_k, (nq, 6, 3)
    dphi = np.zeros((len(l), 6, 3))
    for k in range(3):
        dphi[:, k, k] = 4 * l[:, k] - 1
    dphi[:, 3, 0], dphi[:, 3, 1] = 4 * l[:, 1], 4 * l[:, 0]
    dphi[:, 4, 1], dphi[:, 4, 2] = 4 * l[:, 2], 4 * l[:, 1]
    dphi[:, 5, 0], dphi[:, 5, 2] = 4 * l[:, 2], 4 * l[:, 0]
    gphi = np.einsum("qak,ekd->eqad", dphi, glam)           # (ne, nq, 6, 2)
    W = wq[None, :] * detJ[:, None]                         # (ne, nq)
    Me = np.einsum("eq,qa,qb->eab", W, phi, phi)
    Ke = np.einsum("eq,eqad,eqbd->eab", W, gphi, gphi)
    Bxe = -np.einsum("eq,qc,eqa->eca", W, l, gphi[..., 0])  # (ne, 3, 6)
    Bye = -np.einsum("eq,qc,eqa->eca", W, l, gphi[..., 1])
    Mpe = np.einsum("eq,qc,qd->ecd", W, l, l)
    Kpe = np.einsum("e,ecx,edx->ecd", 0.5 * detJ, glam, glam)
    n2, n1 = nvx * nvy, npx * (ny + 1)

    asm = _assemble_like
    M2 = asm(Me, V, V, (n2, n2))
    K2 = asm(Ke, V, V, (n2, n2))
    Bx = asm(Bxe, Pn, V, (n1, n2))
    By = asm(Bye, Pn, V, (n1, n2))
    M_p = asm(Mpe, Pn, Pn, (n1, n1))
    K_p = asm(Kpe, Pn, Pn, (n1, n1))
    I2 = sp.identity(2, format="csr")
    xs, ys = np.linspace(0, lx, nvx), np.linspace(0, ly, nvy)
    XX, YY = np.meshgrid(xs, ys, indexing="xy")
    coords_v = np.stack([XX.ravel(), YY.ravel()], 1)
    xp, yp = np.linspace(0, lx, npx), np.linspace(0, ly, ny + 1)
    XP, YP = np.meshgrid(xp, yp, indexing="xy")
    coords_p = np.stack([XP.ravel(), YP.ravel()], 1)
    onb = ((coords_v[:, 0] == 0) | (coords_v[:, 0] == xs[-1]) | (coords_v[:, 1] == 0)
           | (coords_v[:, 1] == ys[-1]))
    nb = np.flatnonzero(onb)
    return TaylorHoodDiscretisation(
        _canonical_csr(sp.kron(I2, M2)), _canonical_csr(sp.kron(I2, K2)),
        _canonical_csr(sp.hstack([Bx, By])), M_p, K_p, coords_v, coords_p,
        np.concatenate([nb, nb + n2]).astype(np.int32),
        elem=dict(V=V, P=Pn, W=W, phi=phi, gphi=gphi, lam=l, glam=glam))
