"""One rank of a time-sharded run (spawned by the tests, one process per rank).

Every rank builds the same synthetic problem, creates a time-sharded ``MultiBlockSystem``
on GPU 0 with a host-staged pipe transport, and compares its shard of
operator / preconditioner / solve results with the single-rank CPU oracle.
"""
import numpy as np


def run_rank(rank, world, conns, CN, ksp, out_q):
    try:
        import common
        from control_amd.dist import CallbackComm, PipeTransport, shard_range
        p = common.heat_problem(n=8, n_t=10, CN=CN, beta=1e-2)
        m, nx = p["m"], p["sd"].n_dofs
        lo, hi = shard_range(m, rank, world)
        tr = PipeTransport(rank, world, conns)
        comm = CallbackComm(rank, world, tr.allreduce, tr.sendrecv)
        gsys = common.gpu_system(p, comm=comm)
        osys = common.oracle_system(p)
        mass, schur = (20, 0.5, 2.0), (12, 0.08, 2.1)
        gpc, opc = common.gpu_pc(p, mass, schur), common.oracle_pc(p, mass, schur)

        def shard(v):   # global flat / (2m, nx) -> this rank's local flat vector
            V = np.asarray(v).reshape(2 * m, nx)
            return np.concatenate([V[lo:hi].ravel(), V[m + lo:m + hi].ravel()])

        x = common.rng_vector(osys.N)
        e_op = common.rel_err(gsys.mult(shard(x)), shard(osys.mult(x)))
        e_pc = common.rel_err(gsys.pc_apply(shard(x), gpc), shard(osys.pc_apply(opc, x)))
        X = p["sd"].coords
        xs = np.stack([np.sin(np.pi * X[:, 0]) * np.sin(np.pi * X[:, 1]) * (1 + 0.1 * k)
                       for k in range(2 * m)])
        b = osys.mult(xs.ravel()).reshape(2 * m, nx)
        sp = {"linear_solver": ksp, "gmres_restart": 10, "maximum_iterations": 60,
              "relative_tolerance": 1e-6, "absolute_tolerance": 0.0,
              "monitor_convergence": False, "preconditioner": True}
        uo0, uo1 = np.zeros((m, nx)), np.zeros((m, nx))
        ro = osys.solve(uo0, uo1, b[:m], b[m:], solver_parameters=sp, pc_fn=opc)
        ug0, ug1 = np.zeros((hi - lo, nx)), np.zeros((hi - lo, nx))
        rg = gsys.solve(ug0, ug1, b[lo:hi].copy(), b[m + lo:m + hi].copy(),
                        solver_parameters=sp, pc_fn=gpc)
        e_u = common.rel_err(np.vstack([ug0, ug1]), np.vstack([uo0[lo:hi], uo1[lo:hi]]))
        ho, hg = np.asarray(ro.history), np.asarray(rg.history)
        n = min(len(ho), len(hg))
        e_h = float(np.max(np.abs(hg[:n] - ho[:n]) / ho[:n]))
        out_q.put((rank, "ok", dict(e_op=e_op, e_pc=e_pc, e_u=e_u, e_h=e_h, its_g=rg.its,
                                    its_o=ro.its, hist=hg.tolist())))
    except Exception as e:   # report instead of hanging the other ranks' pipes
        import traceback
        out_q.put((rank, "error", traceback.format_exc() + repr(e)))
