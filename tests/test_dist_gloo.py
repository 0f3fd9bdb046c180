"""World-size-2 CPU test (gloo) of the time-sharded algorithm: partition, operator halos,
all-reduced Krylov products and the pipelined preconditioner sweeps, as restated in
oracle/dist_oracle.py, against the single-rank oracle.  The GPU library implements the
same program (csrc/pc.cpp, csrc/comm.cpp); its own 2/3-rank test is
tests/test_gpu_sharded.py."""
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, CN, q, two_grid=False):
    try:
        sys.path.insert(0, HERE)
        sys.path.insert(0, os.path.dirname(HERE))
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank,
                                world_size=world)
        import common
        from oracle import dist_oracle as do

        class Comm:
            def __init__(self):
                self.rank, self.world = rank, world

            def allreduce(self, a):
                t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).copy())
                dist.all_reduce(t)
                return t.numpy()

            def send(self, a, dst):
                dist.send(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).copy()), dst)

            def recv(self, n, src):
                t = torch.empty(n, dtype=torch.float64)
                dist.recv(t, src)
                return t.numpy()

        p = common.heat_problem(n=8, n_t=9 if CN else 7, CN=CN, beta=1e-2)
        m, nx = p["m"], p["sd"].n_dofs
        ssys = do.ShardedHeatSystem(p["sd"].M, p["blocks"], m, p["nodes"], CN, Comm())
        lo, hi = ssys.lo, ssys.hi
        osys = common.oracle_system(p)
        mass, schur = (20, 0.5, 2.0), (12, 0.08, 2.1)
        from oracle import kkt_oracle as ko
        coarse, sspec = None, ko.ChebSpec(*schur)
        if two_grid:
            # two-grid form of the Schur sub-solves (2 cycles of [coarse correction, 4 sweeps])
            from control_amd.coarse import multilinear_coarse_space
            P = multilinear_coarse_space(p["sd"].coords, p["nodes"], cells=4)
            schur = (4, 0.07, 2.1)
            coarse, sspec = (P, 2), ko.ChebSpec(*schur)
            sspec.coarse = ko.CoarseSpace(P, 2)
        opc = common.oracle_pc(p, mass, schur, coarse=coarse)
        spc = ssys.make_pc(p["n_t"], p["tau"], p["beta"], ko.ChebSpec(*mass), sspec)

        def shard(v):
            V = np.asarray(v).reshape(2 * m, nx)
            return np.concatenate([V[lo:hi].ravel(), V[m + lo:m + hi].ravel()])
        x = common.rng_vector(osys.N)
        e_op = common.rel_err(ssys.mult(shard(x)), shard(osys.mult(x)))
        e_pc = common.rel_err(spc(shard(x)), shard(osys.pc_apply(opc, x)))
        X = p["sd"].coords
        xs = np.stack([np.sin(np.pi * X[:, 0]) * np.sin(np.pi * X[:, 1]) * (1 + 0.1 * k)
                       for k in range(2 * m)])
        b = osys.mult(xs.ravel())
        sp = {"linear_solver": "fgmres", "gmres_restart": 10, "maximum_iterations": 60,
              "relative_tolerance": 1e-6, "absolute_tolerance": 0.0,
              "monitor_convergence": False, "preconditioner": True}
        B = b.reshape(2 * m, nx)
        u0, u1 = np.zeros((m, nx)), np.zeros((m, nx))
        ro = osys.solve(u0, u1, B[:m], B[m:], solver_parameters=sp, pc_fn=opc)
        xl, rs = ssys.solve(shard(b), spc)
        e_u = common.rel_err(xl, shard(np.vstack([u0, u1])))
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok", dict(e_op=e_op, e_pc=e_pc, e_u=e_u, its=(rs.its, ro.its),
                                lo=lo, hi=hi)))
    except Exception as e:
        import traceback
        q.put((rank, "error", traceback.format_exc() + repr(e)))


@pytest.mark.parametrize("CN,two_grid", [(False, False), (True, False), (False, True)])
def test_two_rank_sharded_algorithm_matches_single_rank(CN, two_grid):
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    world, port = 2, _free_port()
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, CN, q, two_grid))
             for r in range(world)]
    for pr in procs:
        pr.start()
    res = {}
    try:
        for _ in range(world):
            rank, status, payload = q.get(timeout=280)
            assert status == "ok", f"rank {rank}: {payload}"
            res[rank] = payload
    finally:
        for pr in procs:
            pr.join(timeout=20)
            if pr.is_alive():
                pr.kill()
    covered = []
    for r in range(world):
        d = res[r]
        covered += list(range(d["lo"], d["hi"]))
        assert d["e_op"] < 1e-13, d
        assert d["e_pc"] < 1e-10, d
        assert abs(d["its"][0] - d["its"][1]) <= (0 if CN else 1), d
        assert d["e_u"] < (1e-6 if CN else 1e-5), d
    assert covered == list(range(len(covered)))


def _transport_worker(rank, world, port, q):
    try:
        sys.path.insert(0, os.path.dirname(HERE))
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        from control_amd.dist import GlooTransport
        tr = GlooTransport(rank, world)
        a = np.arange(5, dtype=np.float64) + 10.0 * rank
        s = a.copy()
        tr.allreduce(s, 0)
        mx = a.copy()
        tr.allreduce(mx, 1)
        # ring: everybody sends its rank to the next rank and receives from the previous one
        dst = rank + 1 if rank + 1 < world else -1
        src = rank - 1 if rank > 0 else -1
        got = tr.sendrecv(np.full(3, float(rank)), dst, 3 if src >= 0 else 0, src)
        q.put((rank, s.tolist(), mx.tolist(), None if got is None else got.tolist()))
    except Exception:          # pragma: no cover
        import traceback
        q.put((rank, "error", traceback.format_exc(), None))


def test_gloo_rehearsal_transport():
    """``control_amd.dist.GlooTransport`` (the host-staged transport `bench.py` uses with
    ``KKT_TRANSPORT=gloo`` to rehearse N > 1 on one GPU): sum / max all-reduce and the
    neighbour hand-off, three ranks."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    world, port, q = 3, _free_port(), None
    q = ctx.Queue()
    procs = [ctx.Process(target=_transport_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    base = np.arange(5, dtype=np.float64)
    for rank, s, mx, got in out:
        assert s != "error", mx
        assert s == (3 * base + 30.0).tolist()
        assert mx == (base + 20.0).tolist()
        assert got == (None if rank == 0 else [float(rank - 1)] * 3)
