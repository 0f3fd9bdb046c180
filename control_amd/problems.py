"""Synthetic BASELINE.json workloads and their GPU systems (host-side builders).

Used by ``bench.py``, the examples and the tests; nothing here touches ``oracle/``.
Sizes and data follow SURVEY.md 8d: P1 mass / stiffness on structured triangulations (6-tet
Kuhn cubes in 3-D), homogeneous Dirichlet conditions on the whole boundary, ``[t_0, T_f] =
[0, 2]``, blocks as ``control/control.py:2907-2978`` builds them.
"""
import numpy as np

from .blocks import instationary_blocks
from .fem import unit_cube_p1, unit_square_p1, unit_square_q2

SEED = 20241008


def heat_problem(space="p1", n=10, n_t=10, beta=1.0e-4, T=2.0, CN=False, share=True,
                 time_dependent=False):
    """Config-1-shaped synthetic heat-control system (README example sizes by default)."""
    sd = {"p1": unit_square_p1, "q2": unit_square_q2, "p1_3d": unit_cube_p1}[space](n)
    tau = T / (n_t - 1.0)
    if time_dependent:
        # a forward operator that differs per time level (what Picard/NS produces): mode G
        K = [sd.K + (0.1 * i) * sd.M for i in range(n_t)]
    else:
        K = sd.K
    b00, b01, b10, b11, m = instationary_blocks(sd.M, K, tau, beta, n_t, CN, share=share)
    return dict(sd=sd, tau=tau, beta=beta, n_t=n_t, CN=CN, m=m,
                blocks=(b00, b01, b10, b11), nodes=sd.boundary)


def gpu_system(p, tile_coordinates=True, **kw):
    """``tile_coordinates``: hand the dof coordinates to the library as the tiling hint of the
    sweep programs (``kkt_set_tile_coordinates``; speed only)."""
    from .multiblock import DirichletBCNullspace, MultiBlockSystem
    sd, m = p["sd"], p["m"]
    ns = tuple(DirichletBCNullspace(p["nodes"]) for _ in range(m))
    g = MultiBlockSystem(sd.n_dofs, sd.n_dofs, *p["blocks"], n_blocks_00=m,
                         n_blocks_11=m, nullspace_0=ns, nullspace_1=ns, CN=p["CN"], **kw)
    if tile_coordinates and getattr(sd, "coords", None) is not None:
        g.set_tile_coordinates(sd.coords)
    return g


def gpu_pc(p, mass, schur, coarse=None):
    """``coarse``: ``(P, cycles)`` -- the two-grid form of the Schur sub-solves."""
    from .multiblock import ChebSpec, CoarseSpace, SchurPC
    sch = ChebSpec(*schur)
    if coarse is not None:
        sch.coarse = CoarseSpace(coarse[0], int(coarse[1]))
    return SchurPC(kind="CN" if p["CN"] else "BE", M=p["sd"].M, beta=p["beta"],
                   bc_nodes=p["nodes"], mass=ChebSpec(*mass), schur=sch,
                   n_t=p["n_t"], tau=p["tau"])


def rng_vector(n, seed=SEED):
    return np.random.default_rng(seed).standard_normal(n)


STOKES_SPECS = dict(mass=(20, 0.3, 1.9), schur=(30, 0.02, 2.2), kp=(30, 0.02, 2.2),
                    mp=(20, 0.25, 2.25))


def stokes_problem(n=4, n_t=4, beta=1.0e-2, T=2.0, CN=False, share=True):
    """Config-3-shaped system: P2-P1 on ``RectangleMesh(n, n, 2, 2)`` (BASELINE configs[2])."""
    from .blocks import instationary_incompressible_blocks
    from .fem import rectangle_p2p1
    th = rectangle_p2p1(n, n, 2.0, 2.0)
    tau = T / (n_t - 1.0)
    bl = instationary_incompressible_blocks(th.M_v, th.K_v, th.B, th.M_p, th.K_p, tau, beta,
                                            n_t, CN, share=share)
    return dict(th=th, tau=tau, beta=beta, n_t=n_t, CN=CN, m=bl["m"], blocks=bl)


def stokes_gpu(p, specs=STOKES_SPECS, options=None, comm=None, device=0, coarse=None,
               kp_coarse=None):
    """Outer system, velocity KKT system and pressure commutator on the GPU + the StokesPC."""
    from .multiblock import (ChebSpec, CoarseSpace, ConstantNullspace, DirichletBCNullspace,
                             MultiBlockSystem, SchurPC, StokesPC)
    th, m, CN, bl = p["th"], p["m"], p["CN"], p["blocks"]
    nsv = DirichletBCNullspace(th.boundary_v)
    kw = dict(sub_n_blocks_00_0=m, sub_n_blocks_11_0=m) if CN else {}
    outer = MultiBlockSystem(th.n_v, th.n_p, *bl["outer"], n_blocks_00=2 * m,
                             n_blocks_11=2 * m, nullspace_0=(nsv,) * (2 * m),
                             nullspace_1=tuple(ConstantNullspace() for _ in range(2 * m)),
                             CN=CN, options=options, comm=comm, shard_families=2, device=device,
                             **kw)
    inner = MultiBlockSystem(th.n_v, th.n_v, *bl["inner"], n_blocks_00=m, n_blocks_11=m,
                             nullspace_0=(nsv,) * m, nullspace_1=(nsv,) * m, CN=CN,
                             options=options, comm=comm, device=device)
    if getattr(th, "coords_v", None) is not None and 2 * len(th.coords_v) == th.n_v:
        inner.set_tile_coordinates(np.vstack([th.coords_v, th.coords_v]))   # component-major
    # the commutator product is a plain block product (control.py:4625-4665): no transforms
    # (time-sharded with `comm`: the outer system by levels of its two block families, the
    # velocity and commutator systems by their levels -- the same [lo, hi) on a rank)
    commutator = MultiBlockSystem(th.n_p, th.n_p, *bl["commutator"], n_blocks_00=m,
                                  n_blocks_11=m, options=options, comm=comm, device=device)
    schur = ChebSpec(*specs["schur"])
    if coarse is not None:      # (P, cycles): two-grid form of the velocity sub-solves
        schur.coarse = CoarseSpace(coarse[0], int(coarse[1]))
    inner_pc = SchurPC(kind="CN" if CN else "BE", M=th.M_v, beta=p["beta"],
                       bc_nodes=th.boundary_v, mass=ChebSpec(*specs["mass"]),
                       schur=schur, n_t=p["n_t"], tau=p["tau"])
    kp = ChebSpec(*specs["kp"])
    if kp_coarse is not None:   # (P_p, cycles): two-grid form of the pressure-Laplacian solve
        kp.coarse = CoarseSpace(kp_coarse[0], int(kp_coarse[1]))
    gpc = StokesPC(inner=inner, inner_pc=inner_pc, commutator=commutator, B=th.B, K_p=th.K_p,
                   M_p=th.M_p, kp=kp, mp=ChebSpec(*specs["mp"]),
                   n_p_blocks=m, b_scale=p["tau"], post_scale=1.0 / p["tau"]**2, cn=CN)
    return outer, gpc


