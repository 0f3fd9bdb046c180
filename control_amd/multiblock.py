"""``MultiBlockSystem`` on the GPU: host-side mirror of the reference's class.

Same constructor / ``solve`` signature, same argument meaning and error behaviour as
``preconditioner/preconditioner.py:216-786`` of sleveque/control, with assembled
matrices (SciPy CSR, ``(indptr, indices, data)`` triples, or anything exposing
``petscmat.getValuesCSR()``) in place of UFL forms, and NumPy arrays in place of
Firedrake ``Function``/``Cofunction``.  All arithmetic happens in ``libkkt.so`` (HIP,
gfx950); this module only marshals.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from . import _lib

__all__ = ["Nullspace", "NoneNullspace", "ConstantNullspace", "DirichletBCNullspace",
           "FullNullspace", "MultiBlockSystem", "SchurPC", "StokesPC", "KSPResult",
           "ChebSpec"]

Q00, Q01, Q10, Q11 = 0, 1, 2, 3
_KSP_TYPES = {"gmres": 0, "fgmres": 1, "minres": 2}
_PC_SIDES = {"left": 0, "right": 1, 0: 0, 1: 1}


# ------------------------------------------------------------------------ nullspaces
class Nullspace:
    """``preconditioner.py:75-116`` (the corrections themselves run on the device)."""


class NoneNullspace(Nullspace):
    """``preconditioner.py:119-130``."""


class ConstantNullspace(Nullspace):
    """``preconditioner.py:133-155``: arithmetic-mean shift (pressure blocks)."""

    def __init__(self, *, alpha=1.0):
        self._alpha = float(alpha)


class DirichletBCNullspace(Nullspace):
    """``preconditioner.py:158-197``.  ``bcs``: dof indices of the homogeneous Dirichlet
    boundary (what ``DirichletBC.nodes`` holds), or objects with a ``.nodes`` attribute."""

    def __init__(self, bcs, *, alpha=1.0):
        if hasattr(bcs, "nodes"):
            bcs = (bcs,)
        if isinstance(bcs, (list, tuple)) and len(bcs) and hasattr(bcs[0], "nodes"):
            for bc in bcs:
                fa = getattr(bc, "function_arg", 0)
                if not (fa == 0 or type(fa).__name__ == "Zero"):
                    raise ValueError("Homogeneous boundary conditions required")
            nodes = np.concatenate([np.asarray(bc.nodes).ravel() for bc in bcs])
        else:
            nodes = np.asarray(bcs).ravel()
        self._nodes = np.unique(nodes.astype(np.int32))
        self._alpha = float(alpha)


class FullNullspace(Nullspace):
    """``preconditioner.py:200-213``: equals a Dirichlet nullspace over every dof with
    ``alpha = 1`` (``y = x`` on the operator side, ``u = b`` on the preconditioner side)."""


# ------------------------------------------------------------------- preconditioner
@dataclass
class ChebSpec:
    """``its`` Jacobi-Chebyshev steps on ``[emin, emax]``; ``its == 0``: one Jacobi step.
    ``eimag > 0`` (Schur sub-solves only): the spectrum lies in the ellipse around the
    interval's mid-point with that imaginary semi-axis (blocks with a convection term);
    ``its = -1`` / ``emin <= 0``: degree / ellipse estimated per matrix on the device."""
    its: int
    emin: float = 0.0
    emax: float = 0.0
    eimag: float = 0.0
    coarse: object = None         # CoarseSpace: two-grid cycles (its = smoothing sweeps per cycle)


@dataclass
class CoarseSpace:
    """Coarse space of the two-grid form of the Schur sub-solves (``kkt_pc_desc.coarse_*``):
    ``P`` (n x n_c, scipy sparse / CSR triple; ``control_amd.coarse.multilinear_coarse_space``)
    and the number of cycles [Galerkin correction, ``its`` smoothing sweeps]."""
    P: object
    cycles: int = 1


@dataclass
class SchurPC:
    """Built-in block Schur preconditioner run on the GPU (``kkt_set_pc_schur``).

    The reference builds the same object as a Python closure in ``construct_pc``
    (``control/control.py:351-450``, ``1943-2440``); pass this descriptor as ``pc_fn``.
    ``kind``: "stationary" | "BE" | "CN".
    """
    kind: str
    M: object
    beta: float
    bc_nodes: Sequence[int]
    mass: ChebSpec
    schur: ChebSpec
    n_t: int = 1
    tau: float = 1.0
    epsilon: float = 1.0e-3


@dataclass
class StokesPC:
    """Built-in preconditioner of the incompressible control systems, run on the GPU
    (``kkt_set_pc_stokes``): the ``pc_fn`` closures of
    ``Stationary.incompressible_linear_solve`` (``control/control.py:986-1085``) and of
    ``Instationary.incompressible_linear_solve`` (BE ``control.py:4515-4687``, CN ``:4318-4513``).

    ``inner``: the velocity KKT ``MultiBlockSystem``; ``inner_pc``: its ``SchurPC``;
    ``commutator``: the pressure-space block system ``block_**_int_p``; ``B, K_p, M_p``:
    divergence, pressure stiffness and pressure mass matrices.
    """
    inner: object
    inner_pc: object
    commutator: object
    B: object
    K_p: object
    M_p: object
    kp: ChebSpec
    mp: ChebSpec
    n_p_blocks: int = 1
    b_scale: float = 1.0
    post_scale: float = 1.0
    inner_its: int = 5            # control.py:1005-1010
    cn: bool = False              # Crank-Nicolson branch (control.py:4318-4513)


class KSPResult:
    """What callers read off the KSP the reference returns (``preconditioner.py:786``)."""

    def __init__(self, reason, its, rnorm, history, info):
        self.reason, self.its, self.rnorm = reason, its, rnorm
        self.history = history
        self.info = info

    def getConvergedReason(self):
        return self.reason

    def getIterationNumber(self):
        return self.its

    def getResidualNorm(self):
        return self.rnorm

    def getConvergenceHistory(self):
        return np.asarray(self.history)


# --------------------------------------------------------------------------- helpers
def _space_dim(space):
    if isinstance(space, (int, np.integer)):
        return int(space)
    for attr in ("dim", "n_dofs"):
        if hasattr(space, attr):
            v = getattr(space, attr)
            return int(v() if callable(v) else v)
    raise TypeError("Space must be a primal space (an int number of dofs, or an object "
                    "with dim())")


def _as_csr(A):
    """-> (indptr int32, indices int32, data float64) with sorted columns."""
    if hasattr(A, "petscmat"):                       # Firedrake assembled matrix
        indptr, indices, data = A.petscmat.getValuesCSR()
    elif hasattr(A, "getValuesCSR"):                 # petsc4py Mat
        indptr, indices, data = A.getValuesCSR()
    elif isinstance(A, tuple) and len(A) == 3:
        indptr, indices, data = A
    else:                                            # SciPy sparse
        if not getattr(A, "has_sorted_indices", True):
            A = A.sorted_indices()
        A = A.tocsr()
        indptr, indices, data = A.indptr, A.indices, A.data
    return (np.ascontiguousarray(indptr, dtype=np.int32),
            np.ascontiguousarray(indices, dtype=np.int32),
            np.ascontiguousarray(data, dtype=np.float64))


def _array_of(v, n, nx):
    """NumPy (n, nx) view of a caller vector (array or Firedrake-like ``.dat.data``)."""
    if hasattr(v, "dat") and hasattr(v.dat, "data"):
        v = v.dat.data
    a = np.asarray(v)
    if a.dtype != np.float64 or a.size != n * nx:
        raise ValueError("vector of wrong size or dtype")
    return a.reshape(n, nx)


# execution options a developer script may pass through the environment (forwarded by the mirror)
_ENV_OPTION_KEYS = ("sell_r", "sell_sort", "no_graph", "persistent", "prog_mode", "prog_waves",
                    "prog_steps", "tile_depth", "tile_waves", "lanes", "lane_chunks",
                    "kernarg_ops", "shared_rows", "verbose", "stamps", "tile_poll_delay",
                    "tile_unfused", "stage_timers", "sell_sigma", "interleave")


# ------------------------------------------------------------------ the block system
class MultiBlockSystem:
    """GPU drop-in for ``preconditioner.py:216`` ``MultiBlockSystem``."""

    def __init__(self, space_0, space_1, block_00, block_01, block_10, block_11, *,
                 n_blocks_00=1, n_blocks_11=1, sub_n_blocks_00_0=None,
                 sub_n_blocks_11_0=None, nullspace_0=None, nullspace_1=None,
                 form_compiler_parameters=None, CN=False, device=0, comm=None, options=None,
                 share_values=True, shard_families=1):
        if nullspace_0 is None:
            nullspace_0 = tuple(NoneNullspace() for _ in range(n_blocks_00))
        if nullspace_1 is None:
            nullspace_1 = tuple(NoneNullspace() for _ in range(n_blocks_11))
        nx0, nx1 = _space_dim(space_0), _space_dim(space_1)

        def check_blocks(block, n_row_blocks, n_col_blocks):   # preconditioner.py:243-258
            if len(block) != n_row_blocks * n_col_blocks:
                raise ValueError("Unexpected dimension of blocks")
        check_blocks(block_00, n_blocks_00, n_blocks_00)
        check_blocks(block_01, n_blocks_00, n_blocks_11)
        check_blocks(block_10, n_blocks_11, n_blocks_00)
        check_blocks(block_11, n_blocks_11, n_blocks_11)

        self._lib = _lib.load()
        self._h = C.c_void_p()
        rc = self._lib.kkt_create(C.byref(self._h), int(device))
        if rc != 0:
            raise _lib.KktError(rc, self._lib.kkt_last_error(None).decode())
        self._n0, self._n1, self._nx0, self._nx1 = n_blocks_00, n_blocks_11, nx0, nx1
        self._CN = bool(CN)
        self._comm = comm
        self._cb_keep = []
        # kernel-form switches (kkt_set_option).  The library itself never reads the environment;
        # developer scripts may still say KKT_<KEY>=value, which this mirror forwards as explicit
        # calls (explicit `options` win; the debug_* hooks are never taken from the environment)
        opts = {k: os.environ["KKT_" + k.upper()] for k in _ENV_OPTION_KEYS
                if "KKT_" + k.upper() in os.environ}
        opts.update(options or {})
        for key, value in opts.items():
            self.set_option(key, value)
        self._ck(self._lib.kkt_set_layout(
            self._h, n_blocks_00, n_blocks_11, nx0, nx1, int(bool(CN)),
            -1 if sub_n_blocks_00_0 is None else int(sub_n_blocks_00_0),
            -1 if sub_n_blocks_11_0 is None else int(sub_n_blocks_11_0)))
        self._lo, self._hi = 0, n_blocks_00
        # time levels per block family (shard_families = 2: the outer incompressible system,
        # whose flat blocks are (v, zeta) and (mu, p) runs of time levels)
        self._mf = n_blocks_00 // int(shard_families)
        if comm is not None and comm.world > 1:
            if shard_families == 1:
                self._ck(self._lib.kkt_set_shard(self._h, comm.rank, comm.world))
            else:
                self._ck(self._lib.kkt_set_shard_families(self._h, comm.rank, comm.world,
                                                          int(shard_families)))
            lo, hi = C.c_int(), C.c_int()
            self._lib.kkt_shard_range(self._mf, comm.rank, comm.world, C.byref(lo), C.byref(hi))
            self._lo, self._hi = lo.value, hi.value
        self._sharded = comm is not None and comm.world > 1
        self._n0_loc = shard_families * (self._hi - self._lo) if self._sharded else n_blocks_00
        self._n1_loc = shard_families * (self._hi - self._lo) if self._sharded else n_blocks_11

        share_ids = {}
        self._structure = {}     # (quadrant, i, j) -> [nnz, hash of the index array or None, the array]
        for q, blk in ((Q00, block_00), (Q01, block_01), (Q10, block_10), (Q11, block_11)):
            for (i, j), A in blk.items():                      # dict order = apply order
                if A is None:
                    continue
                if self._sharded and not (self._lo <= i % self._mf < self._hi):
                    continue
                indptr, indices, data = _as_csr(A)
                nrows = len(indptr) - 1
                ncols = nx0 if q in (Q00, Q10) else nx1
                # the same Python object given for several (i, j) shares device storage
                # (share_values=False: every block gets its own device copy -- "mode G", what
                # the reference stores -- even when the host objects are shared)
                sid = share_ids.setdefault(id(A), len(share_ids)) if share_values else -1
                # the index array is kept referenced, so its address identifies it for good
                self._structure[(q, i, j)] = [len(data), None, indices]   # hash: on first need
                self._ck(self._lib.kkt_add_block(
                    self._h, q, i, j, nrows, ncols,
                    indptr.ctypes.data_as(_lib.c_i32p), indices.ctypes.data_as(_lib.c_i32p),
                    data.ctypes.data_as(_lib.c_f64p), sid))
        for k, ns in enumerate(tuple(nullspace_0) + tuple(nullspace_1)):
            nxk = nx0 if k < n_blocks_00 else nx1
            if isinstance(ns, DirichletBCNullspace):
                nodes, p = _lib.i32(ns._nodes)
                self._ck(self._lib.kkt_set_bc(self._h, k, len(nodes), p, ns._alpha))
            elif isinstance(ns, FullNullspace):
                nodes, p = _lib.i32(np.arange(nxk))
                self._ck(self._lib.kkt_set_bc(self._h, k, len(nodes), p, 1.0))
            elif isinstance(ns, ConstantNullspace):
                self._ck(self._lib.kkt_set_const_nullspace(self._h, k, ns._alpha))
            elif ns is None or isinstance(ns, NoneNullspace):
                pass
            else:
                raise TypeError("unknown nullspace type")
        self._ck(self._lib.kkt_finalize(self._h))
        if self._sharded:
            comm.attach(self)
        self._pc_state = None

    # -- plumbing
    def _ck(self, rc):
        if rc != 0:
            raise _lib.KktError(rc, self._lib.kkt_last_error(self._h).decode())

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._lib.kkt_destroy(h)
            self._h = None

    close = __del__

    @property
    def handle(self):
        return self._h

    @property
    def local_size(self):
        return self._n0_loc * self._nx0 + self._n1_loc * self._nx1

    def set_option(self, key, value):
        """``kkt_set_option``: which kernel form runs (include/kkt.h lists the keys)."""
        self._ck(self._lib.kkt_set_option(self._h, str(key).encode(), str(value).encode()))
        self._pc_state = None if key != "verbose" else getattr(self, "_pc_state", None)

    def set_tile_coordinates(self, coords):
        """``kkt_set_tile_coordinates``: coordinates of the dofs of a spatial block of variable 0,
        shape ``(N_x, dim)`` -- an optional hint for the tiling of the preconditioner's sweep
        programs (boxes instead of graph bisection; results do not depend on it).  Before the
        preconditioner is first used."""
        c = np.ascontiguousarray(np.asarray(coords, dtype=np.float64))
        if c.ndim != 2 or c.shape[0] != self._nx0 or not 1 <= c.shape[1] <= 3:
            raise ValueError("tile coordinates must have shape (N_x of space_0, 1..3)")
        self._ck(self._lib.kkt_set_tile_coordinates(self._h, c.shape[1], c.shape[0],
                                                    c.ctypes.data_as(_lib.c_f64p)))
        self._pc_state = None

    def info(self):
        inf = _lib.Info()
        self._ck(self._lib.kkt_get_info(self._h, C.byref(inf)))
        return inf.as_dict()

    def update_block_values(self, quadrant, i, j, A):
        """New values on a stored block's structure (Picard re-linearisation)."""
        _, indices, data = _as_csr(A)
        ref = self._structure.get((quadrant, i, j))
        # same index buffer as at construction (block sums share it): nothing to compare
        same = ref is not None and ref[0] == len(data)
        if same and ref[2].ctypes.data != indices.ctypes.data:
            if ref[1] is None:
                ref[1] = hash(ref[2].tobytes())
            same = ref[1] == hash(indices.tobytes())
        if not same:
            raise ValueError(f"block ({quadrant}; {i}, {j}): the new matrix does not have the "
                             "stored sparsity structure")
        self._ck(self._lib.kkt_update_block_values(
            self._h, quadrant, i, j, data.ctypes.data_as(_lib.c_f64p)))

    def _join(self, a0, a1):
        return np.concatenate([np.ravel(a0), np.ravel(a1)]).astype(np.float64, copy=False)

    # -- operator and preconditioner, host arrays (MultiBlockSystemMatrix.mult /
    #    Preconditioner.apply, preconditioner.py:375-543 / 562-656)
    def mult(self, x):
        x, px = _lib.f64(x)
        y = np.empty_like(x)
        self._ck(self._lib.kkt_apply(self._h, px, y.ctypes.data_as(_lib.c_f64p)))
        return y

    def pc_apply(self, x, pc_fn=None):
        self._set_pc(pc_fn)
        x, px = _lib.f64(x)
        y = np.empty_like(x)
        rc = self._lib.kkt_pc_apply(self._h, px, y.ctypes.data_as(_lib.c_f64p))
        if rc == -4:
            raise RuntimeError("Error encountered in PETSc solve")
        self._ck(rc)
        return y

    def _set_pc(self, pc_fn):
        if pc_fn is None:
            self._ck(self._lib.kkt_set_pc_identity(self._h))
            self._pc_state = None
        elif isinstance(pc_fn, SchurPC):
            if self._pc_state is pc_fn:
                return
            kinds = {"stationary": 0, "BE": 1, "CN": 2}
            indptr, indices, data = _as_csr(pc_fn.M)
            bc, pbc = _lib.i32(np.asarray(pc_fn.bc_nodes))
            co = getattr(pc_fn.schur, "coarse", None)
            ckw = {}
            if co is not None and int(co.cycles) > 0:
                p_ip, p_ix, p_v = _as_csr(co.P)
                if len(p_ip) - 1 != len(indptr) - 1:
                    raise ValueError("coarse space: P must have one row per spatial dof")
                ncoarse = int(co.P.shape[1]) if hasattr(co.P, "shape") else int(p_ix.max()) + 1
                ckw = dict(coarse_cycles=int(co.cycles), n_coarse=ncoarse,
                           p_indptr=p_ip.ctypes.data_as(_lib.c_i32p),
                           p_indices=p_ix.ctypes.data_as(_lib.c_i32p),
                           p_values=p_v.ctypes.data_as(_lib.c_f64p))
            d = _lib.PcDesc(
                kind=kinds[pc_fn.kind], n_t=int(pc_fn.n_t), tau=float(pc_fn.tau),
                beta=float(pc_fn.beta), epsilon=float(pc_fn.epsilon), nx=len(indptr) - 1,
                m_indptr=indptr.ctypes.data_as(_lib.c_i32p),
                m_indices=indices.ctypes.data_as(_lib.c_i32p),
                m_values=data.ctypes.data_as(_lib.c_f64p), n_bc=len(bc), bc_idx=pbc,
                mass_its=int(pc_fn.mass.its), mass_emin=float(pc_fn.mass.emin),
                mass_emax=float(pc_fn.mass.emax), schur_its=int(pc_fn.schur.its),
                schur_emin=float(pc_fn.schur.emin), schur_emax=float(pc_fn.schur.emax),
                schur_eimag=float(getattr(pc_fn.schur, "eimag", 0.0)), **ckw)
            self._ck(self._lib.kkt_set_pc_schur(self._h, C.byref(d)))
            self._pc_state = pc_fn
        elif isinstance(pc_fn, StokesPC):
            if self._pc_state is pc_fn:
                return
            inner, comm = pc_fn.inner, pc_fn.commutator
            inner._set_pc(pc_fn.inner_pc)
            # inner_solver_parameters of control.py:1005-1010: gmres, 5 iterations, no test
            self._ck(self._lib.kkt_set_krylov(inner._h, 0, -1, 30, 0.0, 0.0, -1.0,
                                              int(pc_fn.inner_its)))
            mats = [_as_csr(A) for A in (pc_fn.B, pc_fn.K_p, pc_fn.M_p)]
            d = _lib.PcStokesDesc(
                n_p_blocks=int(pc_fn.n_p_blocks), cn=int(bool(pc_fn.cn)), nv=self._nx0, np=self._nx1,
                b_scale=float(pc_fn.b_scale), post_scale=float(pc_fn.post_scale),
                kp_its=int(pc_fn.kp.its), kp_emin=float(pc_fn.kp.emin),
                kp_emax=float(pc_fn.kp.emax), mp_its=int(pc_fn.mp.its),
                mp_emin=float(pc_fn.mp.emin), mp_emax=float(pc_fn.mp.emax))
            for name, (ip, ix, va) in zip(("b", "kp", "mp"), mats):
                setattr(d, name + "_indptr", ip.ctypes.data_as(_lib.c_i32p))
                setattr(d, name + "_indices", ix.ctypes.data_as(_lib.c_i32p))
                setattr(d, name + "_values", va.ctypes.data_as(_lib.c_f64p))
            kpc = getattr(pc_fn.kp, "coarse", None)
            if kpc is not None:      # two-grid K_p solve (constants deflated in the library)
                kp_ip, kp_ix, kp_v = _as_csr(kpc.P)
                d.kp_coarse_cycles = int(kpc.cycles)
                d.kp_n_coarse = int(kpc.P.shape[1])
                d.kp_p_indptr = kp_ip.ctypes.data_as(_lib.c_i32p)
                d.kp_p_indices = kp_ix.ctypes.data_as(_lib.c_i32p)
                d.kp_p_values = kp_v.ctypes.data_as(_lib.c_f64p)
            self._ck(self._lib.kkt_set_pc_stokes(self._h, inner._h, comm._h, C.byref(d)))
            self._pc_state = pc_fn          # keeps inner and commutator alive
        elif callable(pc_fn):
            n0, n1, nx0, nx1 = self._n0_loc, self._n1_loc, self._nx0, self._nx1

            def trampoline(user, b0, b1, u0, u1):
                try:
                    B0 = np.ctypeslib.as_array(b0, shape=(n0, nx0))
                    B1 = np.ctypeslib.as_array(b1, shape=(n1, nx1))
                    U0 = np.ctypeslib.as_array(u0, shape=(n0, nx0))
                    U1 = np.ctypeslib.as_array(u1, shape=(n1, nx1))
                    B0.flags.writeable = False
                    B1.flags.writeable = False
                    pc_fn(U0, U1, B0, B1)
                    return 0
                except Exception:               # flag_errors, preconditioner.py:64-72
                    import traceback
                    traceback.print_exc()
                    return 1
            cb = _lib.PC_CALLBACK(trampoline)
            self._cb_keep = [cb]
            self._ck(self._lib.kkt_set_pc_callback(self._h, cb, None))
            self._pc_state = pc_fn
        else:
            raise TypeError("pc_fn must be None, a SchurPC / StokesPC descriptor or a callable")

    # -- preconditioner.py:337-786
    def solve(self, u_0, u_1, b_0, b_1, *, solver_parameters=None, pc_fn=None):
        if solver_parameters is None:
            solver_parameters = {}
        sp = solver_parameters
        U0 = _array_of(u_0, self._n0_loc, self._nx0)
        U1 = _array_of(u_1, self._n1_loc, self._nx1)
        B0 = _array_of(b_0, self._n0_loc, self._nx0)
        B1 = _array_of(b_1, self._n1_loc, self._nx1)
        self._set_pc(pc_fn)
        ksp_type = sp.get("linear_solver", "fgmres")
        if ksp_type not in _KSP_TYPES:
            raise ValueError(f"linear_solver {ksp_type!r} is not available "
                             "(gmres, fgmres and minres are)")
        side = _PC_SIDES[sp["pc_side"]] if "pc_side" in sp else -1
        divtol = sp.get("divergence limit", None)
        max_it = int(sp.get("maximum_iterations", 1000))
        self._ck(self._lib.kkt_set_krylov(
            self._h, _KSP_TYPES[ksp_type], side, int(sp.get("gmres_restart", 30)),
            float(sp["relative_tolerance"]), float(sp["absolute_tolerance"]),
            -1.0 if divtol is None else float(divtol), max_it))
        b, pb = _lib.f64(self._join(B0, B1))
        u, pu = _lib.f64(self._join(U0, U1))
        cap = max_it + 4 + max_it // max(1, int(sp.get("gmres_restart", 30)))
        hist = np.zeros(cap)
        its, reason, nh, rnorm = C.c_int(), C.c_int(), C.c_int(), C.c_double()
        rc = self._lib.kkt_solve(self._h, pb, pu, C.byref(its), C.byref(reason),
                                 C.byref(rnorm), hist.ctypes.data_as(_lib.c_f64p), cap,
                                 C.byref(nh))
        if rc == -4:
            raise RuntimeError("Error encountered in PETSc solve")
        self._ck(rc)
        history = hist[:min(nh.value, cap)].copy()
        if sp.get("monitor_convergence", True) and (self._comm is None or self._comm.rank == 0):
            # the reference prints from a KSP monitor (preconditioner.py:749-754)
            for it, r_norm in enumerate(history):
                print(f"KSP: iteration {it:d}, residual norm {r_norm:.16e}")
        if not sp.get("preconditioner", False) and reason.value <= 0:
            raise RuntimeError("Solver failed to converge")
        k = self._n0_loc * self._nx0
        U0[:] = u[:k].reshape(U0.shape)
        U1[:] = u[k:].reshape(U1.shape)
        return KSPResult(reason.value, its.value, rnorm.value, history, self.info())
