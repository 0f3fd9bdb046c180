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
        y_shard = gsys.mult(shard(x))
        e_op = common.rel_err(y_shard, shard(osys.mult(x)))
        # interior block rows run while the halo travels; the rows that read it run after: the
        # same RowOps as on a single GPU, so the shard equals the single-rank result bit for bit
        single = common.gpu_system(p)
        op_bitwise = bool(np.array_equal(y_shard, shard(single.mult(x))))
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
                                    its_o=ro.its, hist=hg.tolist(), op_bitwise=op_bitwise)))
    except Exception as e:   # report instead of hanging the other ranks' pipes
        import traceback
        out_q.put((rank, "error", traceback.format_exc() + repr(e)))


def run_rank_stokes(rank, world, conns, out_q):
    _stokes_rank(rank, world, conns, out_q, False)


def run_rank_stokes_cn(rank, world, conns, out_q):
    _stokes_rank(rank, world, conns, out_q, True)


def _stokes_rank(rank, world, conns, out_q, CN):
    """Time-sharded instationary Stokes control: the outer system is sharded by levels of its two
    block families (Crank-Nicolson: the sub-block split of the time transforms is the family
    boundary), the nested velocity solve and the commutator product by theirs."""
    try:
        import common
        from control_amd.dist import CallbackComm, PipeTransport, shard_range
        p = common.stokes_problem(n=4, n_t=7 if CN else 6, CN=CN)
        th, m = p["th"], p["m"]
        lo, hi = shard_range(m, rank, world)
        nl = hi - lo
        tr = PipeTransport(rank, world, conns)
        comm = CallbackComm(rank, world, tr.allreduce, tr.sendrecv)
        osys, opc = common.stokes_oracle(p)
        outer, gpc = common.stokes_gpu(p, comm=comm)

        def shard(v):   # global flat -> [v_lo.., zeta_lo.. | mu_lo.., p_lo..]
            v0, v1 = osys.split(np.asarray(v))
            return np.concatenate([v0[lo:hi].ravel(), v0[m + lo:m + hi].ravel(),
                                   v1[lo:hi].ravel(), v1[m + lo:m + hi].ravel()])

        x = common.rng_vector(osys.N)
        e_op = common.rel_err(outer.mult(shard(x)), shard(osys.mult(x)))
        e_pc = common.rel_err(outer.pc_apply(shard(x), gpc), shard(osys.pc_apply(opc, x)))
        rng = np.random.default_rng(common.SEED)
        x0 = rng.standard_normal((2 * m, th.n_v))
        x0[:, th.boundary_v] = 0.0
        x1 = rng.standard_normal((2 * m, th.n_p))
        x1 -= x1.mean(axis=1, keepdims=True)
        b0, b1 = osys.split(osys.mult(osys.join(x0, x1)))
        pick = list(range(lo, hi)) + list(range(m + lo, m + hi))
        u0, u1 = np.zeros((2 * nl, th.n_v)), np.zeros((2 * nl, th.n_p))
        res = outer.solve(u0, u1, b0[pick].copy(), b1[pick].copy(), pc_fn=gpc, solver_parameters={
            "linear_solver": "fgmres", "maximum_iterations": 200, "relative_tolerance": 1.0e-10,
            "absolute_tolerance": 1.0e-30, "monitor_convergence": False})
        e_u0 = float(np.abs(u0 - x0[pick]).max())
        e_u1 = float(np.abs(u1 - u1.mean(axis=1, keepdims=True) - x1[pick]).max())
        out_q.put((rank, "ok", dict(e_op=e_op, e_pc=e_pc, e_u0=e_u0, e_u1=e_u1, reason=res.reason,
                                    its=res.its, hist=np.asarray(res.history).tolist())))
    except Exception as e:
        import traceback
        out_q.put((rank, "error", traceback.format_exc() + repr(e)))


def run_rank_picard(rank, world, conns, out_q):
    _picard_rank(rank, world, conns, out_q, False)


def run_rank_picard_cn(rank, world, conns, out_q):
    _picard_rank(rank, world, conns, out_q, True)


def _picard_rank(rank, world, conns, out_q, CN):
    """Time-sharded Picard loop of Navier-Stokes control (BASELINE configs[4] names 8 GPUs):
    lid-driven cavity of test/test_control.py:4171-4268 at 4 x 4, n_t = 6 (5 blocks with CN),
    nu = 0.2; every rank keeps the whole iterate, solves for its levels, the update is summed
    over the ranks on the host.  Compared with the same loop on one rank."""
    try:
        import common
        from control_amd import picard
        from control_amd.dist import CallbackComm, PipeTransport
        pb, v_init, _ = common.navier_stokes_cavity_problem(n=4, n_t=7 if CN else 6, CN=CN)
        pb.nu = 0.2
        s = common.STOKES_SPECS
        kw = dict(mass=s["mass"], schur=s["schur"], kp=s["kp"], mp=s["mp"],
                  solver_parameters=common.NS_SOLVER_PARAMETERS)
        ref = picard.incompressible_non_linear_solve(pb, picard.GpuLinearSolver(pb, **kw),
                                                     v=v_init, print_error_non_linear=False)
        tr = PipeTransport(rank, world, conns)
        comm = CallbackComm(rank, world, tr.allreduce, tr.sendrecv)
        gls = picard.GpuLinearSolver(pb, comm=comm, host_allreduce=tr.allreduce, **kw)
        out = picard.incompressible_non_linear_solve(pb, gls, v=v_init,
                                                     print_error_non_linear=False)
        out_q.put((rank, "ok", dict(
            converged=out["converged"], n=len(out["norms"]), n_ref=len(ref["norms"]),
            e_norms=float(max(abs(a - b) / ref["norms"][0]
                              for a, b in zip(out["norms"], ref["norms"]))),
            e_v=float(np.abs(out["v"] - ref["v"]).max()),
            e_p=float(np.abs(out["p"] - ref["p"]).max()),
            its=out["linear_iterations"], its_ref=ref["linear_iterations"],
            uploads=gls.uploads, hist=[float(x) for x in out["norms"]])))
    except Exception as e:
        import traceback
        out_q.put((rank, "error", traceback.format_exc() + repr(e)))


def run_rank_timeout(rank, world, conns, out_q):
    """A sweep program times out on ONE rank of a time shard (test hook: tile 0 of rank 0 skips a
    hand-off).  The decision to fall back must be collective -- the time-out word rides on the
    Krylov all-reduce -- or the ranks' sequences of all-reduces and hand-offs no longer match:
    both ranks fall back once, restart together and reproduce the plain-launch solve bit for bit."""
    try:
        import common
        from control_amd.dist import CallbackComm, PipeTransport, shard_range
        p = common.heat_problem(n=96, n_t=12)      # six levels per rank: the hook's third hand-off exists
        m, nx = p["m"], p["sd"].n_dofs
        lo, hi = shard_range(m, rank, world)
        tr = PipeTransport(rank, world, conns)
        schur, mass = (6, 0.05, 2.1), (20, 0.5, 2.0)
        b = common.rng_vector(2 * m * nx).reshape(2 * m, nx)
        sp_ = {"linear_solver": "gmres", "gmres_restart": 10, "maximum_iterations": 8,
               "relative_tolerance": 0.0, "absolute_tolerance": 0.0, "monitor_convergence": False,
               "preconditioner": True}
        out, its, falls, forms = [], [], [], []
        for opts in ({"persistent": "0"},
                     dict({"persistent": "1", "prog_mode": "tile"},
                          **({"debug_drop_handoff": "3"} if rank == 0 else {}))):
            comm = CallbackComm(rank, world, tr.allreduce, tr.sendrecv)
            g = common.gpu_system(p, comm=comm, options=opts)
            u0, u1 = np.zeros((hi - lo, nx)), np.zeros((hi - lo, nx))
            r = g.solve(u0, u1, b[lo:hi].copy(), b[m + lo:m + hi].copy(), solver_parameters=sp_,
                        pc_fn=common.gpu_pc(p, mass, schur))
            out.append(np.vstack([u0, u1]))
            its.append(r.its)
            inf = g.info()
            falls.append(int(inf["program_fallbacks"]))
            forms.append(int(inf["sweep_form"]))
            # and a single application through kkt_pc_apply on a fresh handle
            if opts.get("persistent") == "1":
                comm2 = CallbackComm(rank, world, tr.allreduce, tr.sendrecv)
                g2 = common.gpu_system(p, comm=comm2, options=opts)
                x = np.concatenate([b[lo:hi].ravel(), b[m + lo:m + hi].ravel()])
                y_prog = g2.pc_apply(x, common.gpu_pc(p, mass, schur))
                pc_falls = int(g2.info()["program_fallbacks"])
            else:
                x = np.concatenate([b[lo:hi].ravel(), b[m + lo:m + hi].ravel()])
                y_plain = g.pc_apply(x, common.gpu_pc(p, mass, schur))
        out_q.put((rank, "ok", dict(its=its, falls=falls, forms=forms, pc_falls=pc_falls,
                                    same=bool(np.array_equal(out[0], out[1])),
                                    pc_same=bool(np.array_equal(y_plain, y_prog)))))
    except Exception as e:
        import traceback
        out_q.put((rank, "error", traceback.format_exc() + repr(e)))


def run_rank_coarse(rank, world, conns, out_q):
    """Time-sharded solve with the two-grid sub-solves (tile program per rank): shard of the
    preconditioner application against the single-rank oracle, solve against the one-rank GPU."""
    try:
        import common
        from control_amd.coarse import multilinear_coarse_space
        from control_amd.dist import CallbackComm, PipeTransport, shard_range
        p = common.heat_problem(n=48, n_t=8, beta=1e-4)
        m, nx = p["m"], p["sd"].n_dofs
        lo, hi = shard_range(m, rank, world)
        tr = PipeTransport(rank, world, conns)
        comm = CallbackComm(rank, world, tr.allreduce, tr.sendrecv)
        co = (multilinear_coarse_space(p["sd"].coords, p["nodes"], cells=6), 1)
        mass, schur = (20, 0.5, 2.0), (8, 0.07, 2.1)
        g = common.gpu_system(p, comm=comm)
        osys = common.oracle_system(p)

        def shard(v):
            V = np.asarray(v).reshape(2 * m, nx)
            return np.concatenate([V[lo:hi].ravel(), V[m + lo:m + hi].ravel()])
        x = common.rng_vector(osys.N)
        e_pc = common.rel_err(g.pc_apply(shard(x), common.gpu_pc(p, mass, schur, coarse=co)),
                              shard(osys.pc_apply(common.oracle_pc(p, mass, schur, coarse=co), x)))
        import bench
        g0, g1 = bench.readme_rhs(p)
        sp_ = {"linear_solver": "gmres", "gmres_restart": 10, "maximum_iterations": 100,
               "relative_tolerance": 1e-6, "absolute_tolerance": 0.0, "monitor_convergence": False}
        u0, u1 = np.zeros((hi - lo, nx)), np.zeros((hi - lo, nx))
        r = g.solve(u0, u1, g0[lo:hi].copy(), g1[lo:hi].copy(), solver_parameters=sp_,
                    pc_fn=common.gpu_pc(p, mass, schur, coarse=co))
        one = common.gpu_system(p)
        v0, v1 = np.zeros((m, nx)), np.zeros((m, nx))
        r1 = one.solve(v0, v1, g0, g1, solver_parameters=sp_,
                       pc_fn=common.gpu_pc(p, mass, schur, coarse=co))
        e_u = common.rel_err(np.vstack([u0, u1]), np.vstack([v0[lo:hi], v1[lo:hi]]))
        out_q.put((rank, "ok", dict(e_pc=e_pc, its=r.its, its_one=r1.its, e_u=e_u,
                                    form=int(g.info()["sweep_form"]))))
    except Exception as e:
        import traceback
        out_q.put((rank, "error", traceback.format_exc() + repr(e)))
