"""The Firedrake adapter driven by stand-in objects.

Firedrake cannot be installed in this pipeline, so the adapter's three contact points --
``assemble(form)``, ``.petscmat.getValuesCSR()`` and ``Function.sub(i).dat.data`` /
``DirichletBC.nodes`` -- are exercised with minimal objects of the same shape.  What is
checked is the adapter's own logic (dof numbering of vector spaces, form -> matrix caching,
gather / scatter of mixed vectors, wrapping of a user preconditioner written on vector
objects); the solver underneath is covered by the parity tests.  Parity against a real
Firedrake is unpinned (nothing in the reference's tests can run here)."""
import numpy as np
import pytest

import common
from control_amd import firedrake_adapter as fa


# ----------------------------------------------------------------- stand-ins
class _Dat:
    def __init__(self, a):
        self.data = a

    @property
    def data_ro(self):
        return self.data

    def __len__(self):
        raise TypeError


class _MixedDat:
    def __init__(self, n):
        self._n = n

    def __len__(self):
        return self._n


class _Fn:
    def __init__(self, shape):
        self.dat = _Dat(np.zeros(shape))


class _MixedFn:
    """``Function(MixedFunctionSpace([V] * n))``."""

    def __init__(self, n, shape):
        self._subs = [_Fn(shape) for _ in range(n)]
        self.dat = _MixedDat(n)

    def sub(self, i):
        return self._subs[i]

    def copy(self, deepcopy=False):
        out = _MixedFn(len(self._subs), self._subs[0].dat.data.shape)
        for a, b in zip(out._subs, self._subs):
            a.dat.data[...] = b.dat.data
        return out


class _Space:
    def __init__(self, n_nodes, bs=1, mesh="mesh"):
        self.block_size, self._n, self._mesh = bs, n_nodes, mesh

    def dim(self):
        return self._n * self.block_size

    def mesh(self):
        return self._mesh


class _BC:
    def __init__(self, space, nodes, value=0):
        self._space, self.nodes, self.function_arg = space, np.asarray(nodes), value

    def function_space(self):
        return self._space


class _PetscMat:
    def __init__(self, A):
        self._A = A.tocsr()

    def getValuesCSR(self):
        return self._A.indptr, self._A.indices, self._A.data


class _Assembled:
    def __init__(self, A):
        self.petscmat = _PetscMat(A)


class _Form:
    """A bilinear form: here just a holder of the matrix ``assemble`` would produce."""

    def __init__(self, A):
        self.A = A


@pytest.fixture
def counting_assemble(monkeypatch):
    calls = []

    def assemble(form, fcp):
        calls.append(form)
        return _Assembled(form.A)
    monkeypatch.setattr(fa, "_assemble", assemble)
    return calls


# ----------------------------------------------------------------- CPU: adapter logic
def test_module_imports_without_firedrake_and_fails_loudly_when_it_is_needed():
    with pytest.raises(ModuleNotFoundError):
        fa._assemble(object(), {})


def test_dirichlet_nodes_of_a_vector_space_become_interleaved_dofs():
    V = _Space(10, bs=2)
    ns = fa.DirichletBCNullspace([_BC(V, [1, 4]), _BC(V, [4, 7])], alpha=0.5)
    assert ns._nodes.tolist() == [2, 3, 8, 9, 14, 15] and ns._alpha == 0.5
    assert fa.DirichletBCNullspace(_BC(_Space(10), [3, 0]))._nodes.tolist() == [0, 3]
    with pytest.raises(ValueError, match="Homogeneous"):              # preconditioner.py:166
        fa.DirichletBCNullspace(_BC(V, [1], value=1.0))


def test_gather_and_scatter_of_mixed_vectors():
    f = _MixedFn(3, (4, 2))
    for i in range(3):
        f.sub(i).dat.data[...] = np.arange(8).reshape(4, 2) + 10 * i
    U = fa._gather(f, 3, 8)
    assert U.shape == (3, 8) and U[2].tolist() == list(range(20, 28))   # node-major, interleaved
    g = _MixedFn(3, (4, 2))
    fa._scatter(U, g, 3)
    assert all(np.array_equal(g.sub(i).dat.data, f.sub(i).dat.data) for i in range(3))
    single = _Fn((5,))
    single.dat.data[...] = np.arange(5.0)
    assert fa._gather(single, 1, 5).tolist() == [list(np.arange(5.0))]


def test_mesh_mismatch_is_refused(counting_assemble):
    with pytest.raises(ValueError, match="Unexpected mesh"):
        fa.MultiBlockSystem(_Space(4, mesh="a"), _Space(4, mesh="b"), {}, {}, {}, {})


# ----------------------------------------------------------------- GPU: end to end
def _heat_forms(n=8, n_t=4):
    from control_amd.blocks import instationary_blocks
    from control_amd.fem import unit_square_p1
    sd = unit_square_p1(n)
    tau, beta = 1.0 / (n_t - 1), 1e-2
    b00, b01, b10, b11, m = instationary_blocks(sd.M, sd.K, tau, beta, n_t, CN=False)
    forms = {}

    def as_forms(blk):           # the same matrix object -> the same form object
        return {ij: (None if A is None else forms.setdefault(id(A), _Form(A)))
                for ij, A in blk.items()}
    return sd, m, tau, beta, (b00, b01, b10, b11), tuple(as_forms(b) for b in
                                                         (b00, b01, b10, b11)), forms


@pytest.mark.gpu
def test_adapter_solve_matches_the_array_interface(counting_assemble):
    from control_amd import multiblock as mb
    sd, m, tau, beta, blocks, forms, distinct = _heat_forms()
    V = _Space(sd.n_dofs)
    bc = _BC(V, sd.boundary)
    ns = tuple(fa.DirichletBCNullspace(bc) for _ in range(m))
    system = fa.MultiBlockSystem(V, V, *forms, n_blocks_00=m, n_blocks_11=m,
                                 nullspace_0=ns, nullspace_1=ns)
    assert len(counting_assemble) == len(distinct)          # every distinct form assembled once
    rng = np.random.default_rng(3)
    B0, B1 = rng.standard_normal((2, m, sd.n_dofs))
    B0[:, sd.boundary] = B1[:, sd.boundary] = 0.0
    b_0, b_1, u_0, u_1 = (_MixedFn(m, (sd.n_dofs,)) for _ in range(4))
    fa._scatter(B0, b_0, m)
    fa._scatter(B1, b_1, m)
    sp = {"linear_solver": "fgmres", "gmres_restart": 30, "relative_tolerance": 1e-10,
          "absolute_tolerance": 0.0, "maximum_iterations": 200, "monitor_convergence": False}
    pc = mb.SchurPC(kind="BE", M=sd.M, beta=beta, bc_nodes=sd.boundary, n_t=m, tau=tau,
                    mass=mb.ChebSpec(20, 0.5, 2.0), schur=mb.ChebSpec(20, 0.05, 2.2))
    ksp = system.solve(u_0, u_1, b_0, b_1, solver_parameters=sp, pc_fn=pc)
    assert ksp.reason > 0

    ns2 = tuple(mb.DirichletBCNullspace(sd.boundary) for _ in range(m))
    ref = mb.MultiBlockSystem(sd.n_dofs, sd.n_dofs, *blocks, n_blocks_00=m, n_blocks_11=m,
                              nullspace_0=ns2, nullspace_1=ns2)
    U0, U1 = np.zeros((m, sd.n_dofs)), np.zeros((m, sd.n_dofs))
    ksp2 = ref.solve(U0, U1, B0, B1, solver_parameters=sp, pc_fn=pc)
    assert ksp2.its == ksp.its
    assert np.array_equal(fa._gather(u_0, m, sd.n_dofs), U0)
    assert np.array_equal(fa._gather(u_1, m, sd.n_dofs), U1)


@pytest.mark.gpu
def test_adapter_wraps_a_preconditioner_written_on_vector_objects(counting_assemble):
    """A ``pc_fn(u_0, u_1, b_0, b_1)`` in the reference's style (``preconditioner.py:562-656``
    hands it vector objects): here a block-Jacobi on the diagonal blocks, checked against
    the same preconditioner given on arrays."""
    from control_amd import multiblock as mb
    sd, m, tau, beta, blocks, forms, _ = _heat_forms()
    V = _Space(sd.n_dofs)
    ns = tuple(fa.DirichletBCNullspace(_BC(V, sd.boundary)) for _ in range(m))
    system = fa.MultiBlockSystem(V, V, *forms, n_blocks_00=m, n_blocks_11=m,
                                 nullspace_0=ns, nullspace_1=ns)
    d0 = blocks[0][(0, 0)].diagonal()
    seen = []

    def pc_functions(u_0, u_1, b_0, b_1):
        seen.append(type(u_0).__name__)
        for i in range(m):
            u_0.sub(i).dat.data[:] = b_0.sub(i).dat.data / d0
            u_1.sub(i).dat.data[:] = -b_1.sub(i).dat.data / d0

    def pc_arrays(U0, U1, B0, B1):
        U0[:] = B0 / d0
        U1[:] = -B1 / d0
    rng = np.random.default_rng(5)
    B0, B1 = rng.standard_normal((2, m, sd.n_dofs))
    B0[:, sd.boundary] = B1[:, sd.boundary] = 0.0
    sp = {"linear_solver": "fgmres", "gmres_restart": 30, "relative_tolerance": 1e-6,
          "absolute_tolerance": 0.0, "maximum_iterations": 60, "monitor_convergence": False,
          "preconditioner": True}      # iteration cap reached is not an error here
    b_0, b_1, u_0, u_1 = (_MixedFn(m, (sd.n_dofs,)) for _ in range(4))
    fa._scatter(B0, b_0, m)
    fa._scatter(B1, b_1, m)
    ksp = system.solve(u_0, u_1, b_0, b_1, solver_parameters=sp, pc_fn=pc_functions)
    assert seen and set(seen) == {"_MixedFn"}
    ns2 = tuple(mb.DirichletBCNullspace(sd.boundary) for _ in range(m))
    ref = mb.MultiBlockSystem(sd.n_dofs, sd.n_dofs, *blocks, n_blocks_00=m, n_blocks_11=m,
                              nullspace_0=ns2, nullspace_1=ns2)
    U0, U1 = np.zeros((m, sd.n_dofs)), np.zeros((m, sd.n_dofs))
    ksp2 = ref.solve(U0, U1, B0, B1, solver_parameters=sp, pc_fn=pc_arrays)
    assert ksp.its == ksp2.its
    assert np.array_equal(fa._gather(u_0, m, sd.n_dofs), U0)
