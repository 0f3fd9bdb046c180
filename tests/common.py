"""Shared builders for the parity tests: one problem -> (oracle system, GPU system)."""
import numpy as np

from control_amd.blocks import instationary_blocks, stationary_blocks
from control_amd.fem import unit_cube_p1, unit_square_p1, unit_square_q2

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


def oracle_system(p):
    from oracle import kkt_oracle as ko
    sd, m = p["sd"], p["m"]
    ns = tuple(ko.DirichletBCNullspace(p["nodes"]) for _ in range(m))
    return ko.OracleSystem(sd.n_dofs, sd.n_dofs, *p["blocks"], n_blocks_00=m,
                           n_blocks_11=m, nullspace_0=ns, nullspace_1=ns, CN=p["CN"])


def gpu_system(p, **kw):
    from control_amd.multiblock import DirichletBCNullspace, MultiBlockSystem
    sd, m = p["sd"], p["m"]
    ns = tuple(DirichletBCNullspace(p["nodes"]) for _ in range(m))
    return MultiBlockSystem(sd.n_dofs, sd.n_dofs, *p["blocks"], n_blocks_00=m,
                            n_blocks_11=m, nullspace_0=ns, nullspace_1=ns, CN=p["CN"], **kw)


def oracle_pc(p, mass, schur):
    from oracle import kkt_oracle as ko
    sd = p["sd"]
    b00, b01, b10, b11 = p["blocks"]
    f = ko.pc_instationary_CN if p["CN"] else ko.pc_instationary_BE
    return f(sd.M, b01, b10, p["n_t"], p["tau"], p["beta"], p["nodes"],
             ko.ChebSpec(*mass), ko.ChebSpec(*schur))


def gpu_pc(p, mass, schur):
    from control_amd.multiblock import ChebSpec, SchurPC
    return SchurPC(kind="CN" if p["CN"] else "BE", M=p["sd"].M, beta=p["beta"],
                   bc_nodes=p["nodes"], mass=ChebSpec(*mass), schur=ChebSpec(*schur),
                   n_t=p["n_t"], tau=p["tau"])


def rng_vector(n, seed=SEED):
    return np.random.default_rng(seed).standard_normal(n)


def rel_err(a, b):
    return np.linalg.norm(np.ravel(a) - np.ravel(b)) / max(np.linalg.norm(np.ravel(b)), 1e-300)
