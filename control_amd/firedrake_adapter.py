"""Firedrake-facing adapter: the reference's ``MultiBlockSystem`` signature on
UFL forms, Firedrake ``Function`` / ``Cofunction`` vectors and ``DirichletBC`` objects, mapped
onto :mod:`control_amd.multiblock`.

The reference side swaps one import (``control/control.py:9``)::

    from control_amd.firedrake_adapter import *

Everything Firedrake-specific goes through three hooks that are looked up when first used, so
the module imports without Firedrake and the tests of this repository drive it with stand-in
objects of the same shape (``tests/test_firedrake_adapter.py``):

* ``_assemble(form, form_compiler_parameters)`` -- ``firedrake.assemble``
  (``preconditioner.py:305-328`` assembles every block form in the constructor);
* the assembled object's ``.petscmat.getValuesCSR()`` (read by ``multiblock._as_csr``);
* a vector's ``.sub(i).dat.data`` / ``.dat.data`` arrays and a boundary condition's ``.nodes``
  and ``.function_space().block_size`` (nodes of a vector space own ``block_size`` dofs each,
  interleaved, as in the ``Mat``).
"""
import numpy as np

from . import multiblock as mb
from .multiblock import (ChebSpec, ConstantNullspace, FullNullspace, NoneNullspace,  # noqa: F401
                         SchurPC, StokesPC)

__all__ = ["MultiBlockSystem", "DirichletBCNullspace", "ConstantNullspace", "NoneNullspace",
           "FullNullspace", "SchurPC", "StokesPC", "ChebSpec"]


def _assemble(form, form_compiler_parameters):
    from firedrake import assemble                   # absent here: the tests replace this hook
    return assemble(form, form_compiler_parameters=form_compiler_parameters)


def _block_size(space):
    for name in ("block_size", "value_size"):
        bs = getattr(space, name, None)
        if bs is not None:
            return int(bs() if callable(bs) else bs)
    return 1


def _space_dim(space):
    """Process-local number of dofs of a function space (``V.dim()`` in serial)."""
    if isinstance(space, (int, np.integer)):
        return int(space)
    if hasattr(space, "dof_dset"):                   # owned dofs x block size
        return int(space.dof_dset.size) * _block_size(space)
    d = space.dim
    return int(d() if callable(d) else d)


class DirichletBCNullspace(mb.DirichletBCNullspace):
    """``preconditioner.py:158-197`` on ``DirichletBC`` objects: ``bc.nodes`` are node numbers,
    a node of a vector-valued space owns ``block_size`` consecutive dofs."""

    def __init__(self, bcs, *, alpha=1.0):
        if hasattr(bcs, "nodes"):
            bcs = (bcs,)
        dofs = []
        for bc in bcs:
            fa = getattr(bc, "function_arg", 0)
            if not (isinstance(fa, (int, float)) and fa == 0 or type(fa).__name__ == "Zero"):
                raise ValueError("Homogeneous boundary conditions required")   # :166-169
            space = bc.function_space() if callable(getattr(bc, "function_space", None)) else None
            bs = _block_size(space) if space is not None else 1
            nodes = np.asarray(bc.nodes, dtype=np.int64).ravel()
            dofs.append((nodes[:, None] * bs + np.arange(bs)[None, :]).ravel())
        super().__init__(np.concatenate(dofs) if dofs else np.zeros(0, dtype=np.int64),
                         alpha=alpha)


def _vector_arrays(f, n):
    """The ``n`` sub-vectors of a (mixed) ``Function`` / ``Cofunction`` as writable arrays."""
    if n == 1 and not hasattr(f, "subfunctions") and not _has_subs(f):
        return [f.dat.data]
    return [f.sub(i).dat.data for i in range(n)]


def _has_subs(f):
    try:
        return len(f.dat) > 1                        # MixedDat
    except TypeError:
        return False


def _gather(f, n, nx):
    out = np.empty((n, nx))
    for i, a in enumerate(_vector_arrays(f, n)):
        out[i] = np.asarray(a, dtype=np.float64).ravel()
    return out


def _scatter(U, f, n):
    for i, a in enumerate(_vector_arrays(f, n)):
        a[...] = U[i].reshape(np.shape(a))


class MultiBlockSystem(mb.MultiBlockSystem):
    """``preconditioner.py:216-335`` with the reference's argument types."""

    def __init__(self, space_0, space_1, block_00, block_01, block_10, block_11, *,
                 form_compiler_parameters=None, **kw):
        if hasattr(space_0, "mesh") and hasattr(space_1, "mesh") \
                and space_0.mesh() != space_1.mesh():
            raise ValueError("Unexpected mesh")                      # :237-238
        fcp = form_compiler_parameters or {}
        assembled = {}                 # one assembled matrix per distinct form: blocks given
                                       # the same form share device storage (mode S)

        def asm(blocks):
            out = {}
            for ij, form in blocks.items():
                if form is None:
                    out[ij] = None
                    continue
                key = id(form)
                if key not in assembled:
                    assembled[key] = _assemble(form, fcp)
                out[ij] = assembled[key]
            return out
        self._spaces_fd = (space_0, space_1)
        super().__init__(_space_dim(space_0), _space_dim(space_1), asm(block_00), asm(block_01),
                         asm(block_10), asm(block_11), **kw)

    def solve(self, u_0, u_1, b_0, b_1, *, solver_parameters=None, pc_fn=None):
        """``preconditioner.py:337-786``: the initial guess is read from ``u_0, u_1`` and the
        solution written back into them; returns the KSP-like result object."""
        n0, n1, nx0, nx1 = self._n0_loc, self._n1_loc, self._nx0, self._nx1
        U0, U1 = _gather(u_0, n0, nx0), _gather(u_1, n1, nx1)
        B0, B1 = _gather(b_0, n0, nx0), _gather(b_1, n1, nx1)
        if pc_fn is not None and not isinstance(pc_fn, (mb.SchurPC, mb.StokesPC)):
            pc_fn = self._wrap_function_pc(pc_fn, u_0, u_1, b_0, b_1)
        ksp = super().solve(U0, U1, B0, B1, solver_parameters=solver_parameters, pc_fn=pc_fn)
        _scatter(U0, u_0, n0)
        _scatter(U1, u_1, n1)
        return ksp

    def _wrap_function_pc(self, pc_fn, u_0, u_1, b_0, b_1):
        """A user ``pc_fn(u_0, u_1, b_0, b_1)`` written on Firedrake vectors
        (``preconditioner.py:562-656`` hands it ``Function`` / ``Cofunction`` objects): called
        through the host-callback preconditioner on copies of the caller's vector types."""
        work = tuple(v.copy(deepcopy=True) for v in (u_0, u_1, b_0, b_1))
        n0, n1 = self._n0_loc, self._n1_loc

        def pc_arrays(U0, U1, B0, B1):
            w0, w1, c0, c1 = work
            _scatter(B0, c0, n0)
            _scatter(B1, c1, n1)
            for w, n in ((w0, n0), (w1, n1)):
                for a in _vector_arrays(w, n):
                    a[...] = 0.0
            pc_fn(w0, w1, c0, c1)
            U0[:] = _gather(w0, n0, self._nx0)
            U1[:] = _gather(w1, n1, self._nx1)
        return pc_arrays
