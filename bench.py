#!/usr/bin/env python3
"""Benchmark of the all-at-once KKT hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one preconditioned Krylov iteration (one KKT operator apply, one block-Schur
preconditioner apply, one Gram-Schmidt sweep) of GMRES(10) -- the library default of the
reference (control/control.py:3260-3266) -- on the synthetic heat-control system of
BASELINE.json configs[1]: 2-D heat control, 256x256 P1, n_t = 64, beta = 1e-4, T = 2,
fp64, every (i, j) time block stored with its own values ("mode G", what the reference
stores, preconditioner.py:305-328).  W warm-up iterations, then exactly K timed ones
(rtol = 0, so the solver runs to max_it = K), inputs resident in HBM.

N > 1: under a launcher (torch.distributed.run sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*)
every process is one rank; started plainly (`python bench.py --gpus N`) this process only
spawns the N rank processes (fresh children, it never touches a GPU itself) and relays rank 0's
line.  Time-block rows are sharded over the ranks, RCCL over xGMI carries the halo vectors and
the Krylov all-reduces.  If RCCL does not start on N distinct GPUs the run FAILS (non-zero exit):
the host-staged gloo transport is only for rehearsals with all ranks on one GPU (KKT_DEVICE=0).

One JSON line on stdout (rank 0): metric/value = whole-job Krylov iterations per second;
`roofline` = the KKT block-row SpMV kernel (kkt_spmv_rows) against the 8 TB/s HBM peak,
timed with HIP events on the library's stream; `cpu_baseline` = the CPU oracle timed on
this box's host cores on a bounded sample of the same workload (C/OpenMP restatement).
No PyTorch: the GPU process binds libkkt.so through ctypes only.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
# what a float4 copy kernel reaches on this chip (MI355X_MICROARCH.md, "HBM bandwidth": measured
# copy rate; same figure in profiles/r01/README.md) -- a side figure next to the spec peak
MEASURED_COPY_GBS = 6290.0
SWEEP_FORMS = {0: "plain launches", 1: "row program (counters)", 2: "row program (data-flow)",
               3: "tile program"}


def sweep_plan(info):
    """How the time sweeps of the built-in preconditioner really run (from the library)."""
    return {"form": SWEEP_FORMS.get(int(info.get("sweep_form", 0)), "?"),
            "tiles": int(info.get("sweep_tiles", 0)), "threads": int(info.get("sweep_threads", 0)),
            "depth": int(info.get("sweep_depth", 0)),
            "row_slots": int(info.get("sweep_row_slots", 0)),
            "chebyshev_degree": int(info.get("sweep_its", 0)),
            "program_fallbacks": int(info.get("program_fallbacks", 0))}


def roofline_check(roof):
    """A fraction outside (0, 1] is a bookkeeping error: flag it in the line, keep the line."""
    f = roof.get("frac")
    if f is None or not (0.0 < f <= 1.0):
        roof["error"] = f"frac = {f}: not a fraction of the peak (bytes or time mis-counted)"
        return False
    return True


def pc_stages(gsys, lib, h, d_x, d_y):
    """One application of a block-Schur preconditioner replayed step by step (kkt_time_pc_stages)."""
    from control_amd import _lib
    ps = _lib.PcStageTimes()
    gsys._ck(lib.kkt_time_pc_stages(h, d_x, d_y, C.byref(ps)))     # warm-up
    gsys._ck(lib.kkt_time_pc_stages(h, d_x, d_y, C.byref(ps)))
    return {"time_sweeps": ps.sweeps_ms, "batched_steps": ps.batched_ms,
            "rank_handoffs": ps.comm_ms, "sum": ps.total_ms,
            "sweep_launches": int(ps.sweep_launches), "sweep_phases": int(ps.sweep_phases),
            "batched_launches": int(ps.batched_launches), "handoff_steps": int(ps.comm_steps),
            "note": "one application replayed step by step with an event after every step "
                    "(no hipGraph: launch gaps are inside the figures); rank_handoffs includes "
                    "the wait for the neighbour rank's stage of the sweep pipeline"}


def stage_breakdown(gsys, lib, h, d_b, d_u, d_x, d_y, n_local, its, krylov_type, with_pc_stages=True):
    """SURVEY 8e itemisation: a short solve with HIP events between the stages of every
    iteration (outside the timed region), and one preconditioner application step by step."""
    from control_amd import _lib
    out = {}
    try:
        gsys.set_option("stage_timers", "1")
        gsys._ck(lib.kkt_vec_upload(h, d_u, _lib.f64(np.zeros(n_local))[1]))
        gsys._ck(lib.kkt_set_krylov(h, krylov_type, -1, 10, 0.0, 0.0, 1e300, its))
        i_, r_, nh, rn = C.c_int(), C.c_int(), C.c_int(), C.c_double()
        gsys._ck(lib.kkt_solve_device(h, d_b, d_u, C.byref(i_), C.byref(r_), C.byref(rn), None, 0,
                                      C.byref(nh)))
        st = _lib.StageTimes()
        gsys._ck(lib.kkt_get_stage_times(h, C.byref(st)))
        gsys.set_option("stage_timers", "0")
        n = max(1, int(st.iterations))
        out["krylov_iteration_ms"] = {
            "operator": st.operator_ms / n, "preconditioner": st.pc_ms / n,
            "orthogonalisation": st.orth_ms / n, "allreduce": st.allreduce_ms / n,
            "other": st.other_ms / n, "sum": st.total_ms / n,
            "iterations": int(st.iterations), "operator_applies": int(st.operator_applies),
            "pc_applies": int(st.pc_applies),
            "note": "GPU time between HIP events on the library's stream, per iteration of a "
                    f"{int(st.iterations)}-iteration solve after the timed region (rank 0); "
                    "operator includes its halo exchange, allreduce the wait for the slowest rank, "
                    "other = residual set-up at restarts, normalisation, host round trips"}
    except Exception as e:      # noqa: BLE001 -- a side measurement must not lose the headline
        out["krylov_iteration_ms"] = {"error": f"{type(e).__name__}: {e}"}
    if with_pc_stages:
        try:
            out["preconditioner_application_ms"] = pc_stages(gsys, lib, h, d_x, d_y)
        except Exception as e:      # noqa: BLE001
            out["preconditioner_application_ms"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def measured_traffic(workload):
    """HBM bytes per kkt_spmv_rows launch from the committed rocprofv3 --pmc passes
    (profiles/*/traffic_kkt_spmv_rows*.json), for this exact workload; else None."""
    return measured_traffic_named("traffic_kkt_spmv_rows*.json", workload)


def measured_traffic_named(pattern, workload):
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", pattern))):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        if d.get("workload") == workload:
            best = d.get("hbm_bytes_per_launch_corrected")
    return best


def measured_sweep_traffic(workload, preconditioner, phases_per_launch):
    """HBM-side bytes per launch of the sweep program from the committed --pmc passes
    (profiles/*/traffic_pc_row_program_g.json) for this workload and preconditioner."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "traffic_pc_row_program_g.json"))
                    + glob.glob(os.path.join(ROOT, "profiles", "*", "traffic_pc_tile_sweep.json"))):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        if (d.get("workload") == workload and d.get("preconditioner") == preconditioner
                and d.get("phases_per_launch") == phases_per_launch):
            best = d.get("hbm_bytes_per_launch")
    return best


def build_problem(args):
    from control_amd.blocks import instationary_blocks
    from control_amd.fem import unit_cube_p1, unit_square_p1
    if args.workload == "heat2d":
        sd = unit_square_p1(args.n)
        mass_bounds = (0.5, 2.0)        # test_control.py:3477
    else:
        sd = unit_cube_p1(args.n)
        mass_bounds = (0.5, 2.5)        # Wathen's bound for P1 tetrahedra
    tau = args.T / (args.n_t - 1.0)
    CN = args.scheme == "CN"
    # host objects are shared either way; mode G asks the library for one device copy per
    # (i, j) block (share_values=False) -- 38.8 GB of host copies for configs[3] otherwise
    blocks = instationary_blocks(sd.M, sd.K, tau, args.beta, args.n_t, CN, share=True)
    schur = (args.schur_its, args.schur_emin, args.schur_emax)
    if getattr(args, "schur_auto", False):
        # degree and one interval per sub-solve matrix from Lanczos estimates on the device
        schur = (-1, 0.0, 0.0)
    coarse = None
    if getattr(args, "coarse_cycles", 0) > 0:
        # two-grid form of the sub-solves: Galerkin correction on the multilinear coarse space of
        # ~coarse_nodes functions, then `schur_its` smoothing sweeps on the upper part of the
        # spectrum (the reference runs BoomerAMG here)
        from control_amd.coarse import multilinear_coarse_space
        # coarse cells of --coarse-cell mesh widths (8: 33^2 coarse functions on 256^2, 9^3 on 64^3)
        cells = max(2, args.n // max(1, args.coarse_cell))
        coarse = (multilinear_coarse_space(sd.coords, sd.boundary, cells=cells), args.coarse_cycles)
    return dict(sd=sd, tau=tau, beta=args.beta, n_t=args.n_t, CN=CN, m=blocks[4],
                blocks=blocks[:4], nodes=sd.boundary, mass=(20,) + mass_bounds, schur=schur,
                coarse=coarse, share_values=(args.mode == "S"))


def readme_rhs(p):
    """Rows of the README example (v_d = t c, f = c, zero initial state; README.md:34-56) for all
    ``m`` blocks, as ``Instationary.linear_solve`` builds them (control.py:2991-3243): BE, or CN
    with ``T_1`` / ``T_2`` applied."""
    sd, m, tau, n_t = p["sd"], p["m"], p["tau"], p["n_t"]
    X = sd.coords
    cX = np.prod(np.cos(0.5 * np.pi * (X - 1.0)), axis=1)
    Mc = sd.M @ cX
    vd = np.stack([(i * tau) * Mc for i in range(n_t)])       # assemble(inner(v_d, test) dx)
    ff = np.stack([Mc for _ in range(n_t)])
    if p["CN"]:
        h2 = 0.5 * tau
        g0 = np.stack([h2 * (vd[i] + vd[i + 1]) for i in range(m)])
        g1 = np.stack([h2 * (ff[i] + ff[i + 1]) for i in range(m)])
    else:
        g0 = np.stack([tau * vd[i] * (i < n_t - 1) for i in range(m)])
        g1 = np.stack([tau * ff[i] * (i >= 1) for i in range(m)])
    g0[:, p["nodes"]] = 0.0
    g1[:, p["nodes"]] = 0.0
    if p["CN"]:                                              # apply_T_1 / apply_T_2, :3242
        t0_, t1_ = g0.copy(), g1.copy()
        t0_[:-1] += g0[1:]
        t1_[1:] += g1[:-1]
        g0, g1 = t0_, t1_
    return g0, g1


def usable_cores():
    """Host cores this process may really use: the scheduler affinity, capped by the cgroup
    CPU quota (a one-GPU box grants a 16-core share of a much larger host)."""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]))))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, quota // period))
        except (OSError, ValueError, IndexError):
            pass
    # every core the box grants (KKT_CPU_THREADS caps it for experiments)
    cap = os.environ.get("KKT_CPU_THREADS")
    return max(1, min(n, int(cap)) if cap else n)


def cpu_baseline(p, its_all, its_one):
    """The C/OpenMP restatement of the same algorithm (oracle/csrc/kkt_ref.c: per-block CSR
    SpMV as PETSc's MatMultAdd_SeqAIJ, the same block-Schur preconditioner with the same
    Chebyshev parameters, the same GMRES(10)) on this box's host cores, on a bounded
    sample: `its_all` iterations of the same system on all granted cores and `its_one` on one
    thread (0: skipped)."""
    import ctypes
    from control_amd import problems as common
    from oracle import cref
    if p["CN"]:
        return None   # the C restatement covers the BE benchmark configuration
    print("[bench] cpu baseline: building the C restatement's matrices", file=sys.stderr,
          flush=True)
    c = cref.CRef(p["blocks"], p["m"], p["sd"].n_dofs, p["nodes"], p["sd"].M, p["n_t"],
                  p["tau"], p["beta"], p["mass"], p["schur"], coarse=p.get("coarse"))
    b = common.rng_vector(2 * p["m"] * p["sd"].n_dofs)
    gomp = ctypes.CDLL("libgomp.so.1")
    cores = usable_cores()
    out = {}
    for threads, its in ((cores, its_all), (1, its_one)):
        if its < 1:
            continue
        gomp.omp_set_num_threads(threads)
        print(f"[bench] cpu baseline: {its} iterations on {threads} thread(s)",
              file=sys.stderr, flush=True)
        if threads == cores:
            c.gmres(b, np.zeros_like(b), rtol=0.0, max_it=1)      # warm-up, page-in
        t0 = time.perf_counter()
        _, n_it, _, _ = c.gmres(b, np.zeros_like(b), rtol=0.0, divtol=1e300, max_it=its)
        dt = time.perf_counter() - t0
        out[threads] = (n_it / dt, n_it, dt)
    v, n_it, dt = out[cores]
    res = {"value": v, "unit": "Krylov iterations/s", "cores": cores, "kind": "port",
           "sample": (f"{n_it} GMRES(10) iterations of the same system (same blocks, same "
                      f"preconditioner parameters) with oracle/csrc/kkt_ref.c, gcc -O3 "
                      f"-march=x86-64-v3 -fopenmp, {cores} threads, {dt:.1f} s")}
    if 1 in out and cores != 1:
        res["single_thread_value"] = out[1][0]
        res["single_thread_sample"] = f"{out[1][1]} iterations, {out[1][2]:.1f} s"
    return res


def bench_stokes(args, rank, world, local_rank):
    """BASELINE configs[2] (a parity/measurement case, not the headline line): instationary
    Stokes control, Taylor-Hood P2-P1 on RectangleMesh(n, n, 2, 2), outer FGMRES(10) with the
    StokesPC (5 nested GMRES iterations on the velocity KKT system per application).  Sub-solves:
    the setting with which the outer iteration converges at this size (see --kp-its in main()).
    N > 1: the outer, velocity and commutator systems are time-sharded over the ranks
    (csrc/pc_stokes.cpp); the time-to-solution leg runs at N = 1 only."""
    from control_amd import _lib
    from control_amd import problems as common
    from control_amd.dist import make_comm
    n = args.n if args.n != 256 else 128
    n_t = args.n_t if args.n_t != 64 else 32
    beta = args.beta if args.beta != 1.0e-4 else 1.0e-3
    CN = args.scheme == "CN"
    print("[bench] assembling the synthetic Stokes system", file=sys.stderr, flush=True)
    p = common.stokes_problem(n=n, n_t=n_t, beta=beta, T=args.T, CN=CN, share=(args.mode == "S"))
    # velocity sub-solves and the pressure-Laplacian solve of the commutator: degree and
    # per-matrix intervals from the library's spectrum estimates (its = -1: 1.6 sqrt(kappa)
    # sweeps) unless --schur-its / --kp-its hand-set them.  The pressure solve keeps its own
    # polynomial whatever form the velocity sub-solves take.
    specs = dict(mass=(20, 0.3924, 2.0598), mp=(20, 0.5, 2.0),
                 schur=(((8, 0.07, 2.25) if args.coarse_cycles > 0 else (-1, 0.0, 0.0))
                        if args.stokes_schur_its is None else
                        (args.stokes_schur_its, args.stokes_schur_emin, max(args.schur_emax, 2.25))),
                 kp=((-1, 0.0, 0.0) if args.kp_its < 0 else
                     (args.kp_its, args.kp_emin, args.schur_emax)))
    comm = make_comm(rank, world, local_rank) if world > 1 else None
    device = int(os.environ.get("KKT_DEVICE", local_rank))
    coarse = None
    if args.coarse_cycles and args.coarse_cycles > 0:
        # two-grid form of the velocity sub-solves: multilinear coarse functions per velocity
        # component on cells of --coarse-cell P2 node spacings
        from control_amd.coarse import multilinear_coarse_space
        th_ = p["th"]
        cells = max(2, (2 * n) // max(1, args.coarse_cell))
        coarse = (multilinear_coarse_space(np.vstack([th_.coords_v, th_.coords_v]), th_.boundary_v,
                                           cells=cells), args.coarse_cycles)
    kp_coarse = None
    if args.kp_coarse_cycles > 0:
        # two-grid form of the pressure-Laplacian solve: multilinear coarse functions on cells of
        # --kp-coarse-cell pressure-node spacings, the constants deflated in the library
        from control_amd.coarse import multilinear_coarse_space
        kp_coarse = (multilinear_coarse_space(p["th"].coords_p, (),
                                              cells=max(2, n // max(1, args.kp_coarse_cell))),
                     args.kp_coarse_cycles)
    lib_options = dict(kv.split("=", 1) for kv in getattr(args, "lib_option", []) or []) or None
    outer, gpc = common.stokes_gpu(p, specs, comm=comm, device=device, coarse=coarse,
                                   kp_coarse=kp_coarse, options=lib_options)
    lib, h = outer._lib, outer.handle
    if not args.only_spmv:
        outer._set_pc(gpc)
    info = outer.info()
    n_local = info["n_local"]

    def dvec(host=None):
        d = C.c_void_p()
        outer._ck(lib.kkt_vec_alloc(h, C.byref(d)))
        if host is not None:
            outer._ck(lib.kkt_vec_upload(h, d, _lib.f64(host)[1]))
        return d
    x = common.rng_vector(n_local, common.SEED + rank)
    d_x, d_y, d_u = dvec(x), dvec(), dvec()
    ms = C.c_float()
    outer._ck(lib.kkt_time_apply(h, d_x, d_y, 5, C.byref(ms)))
    outer._ck(lib.kkt_time_apply(h, d_x, d_y, args.spmv_reps, C.byref(ms)))
    spmv_ms = ms.value / args.spmv_reps
    if args.only_spmv:
        if rank == 0:
            alg_ = info["bytes_streamed"]
            th_ = p["th"]
            print(json.dumps({
                "config": {"workload": (f"2-D Stokes control, Taylor-Hood P2-P1 {n}x{n}, n_t={n_t}, "
                                        f"beta={beta:g}, T={args.T:g}, {args.scheme}, mode {args.mode}"),
                           "unknowns": int(2 * p["m"] * (th_.n_v + th_.n_p))},
                "roofline": {"achieved": alg_ / (spmv_ms * 1e-3) / 1e9,
                             "frac": alg_ / (spmv_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "algorithmic_bytes_per_launch": alg_, "launch_ms": spmv_ms,
                             "kernel_launches_per_apply": int(info.get("apply_launches", 0))}}))
        return 0
    outer._ck(lib.kkt_time_pc_apply(h, d_x, d_y, 1, C.byref(ms)))
    outer._ck(lib.kkt_time_pc_apply(h, d_x, d_y, 3, C.byref(ms)))
    pc_ms = ms.value / 3

    def run(max_it):
        outer._ck(lib.kkt_vec_upload(h, d_u, _lib.f64(np.zeros(n_local))[1]))
        outer._ck(lib.kkt_set_krylov(h, 1, -1, 10, 0.0, 0.0, 1e300, max_it))
        its, reason, nh, rn = C.c_int(), C.c_int(), C.c_int(), C.c_double()
        outer._ck(lib.kkt_comm_barrier(h))
        outer._ck(lib.kkt_sync(h))
        t0 = time.perf_counter()
        outer._ck(lib.kkt_solve_device(h, d_x, d_u, C.byref(its), C.byref(reason),
                                       C.byref(rn), None, 0, C.byref(nh)))
        outer._ck(lib.kkt_sync(h))
        outer._ck(lib.kkt_comm_barrier(h))
        dt = C.c_double(time.perf_counter() - t0)
        outer._ck(lib.kkt_comm_max(h, C.byref(dt)))
        return its.value, dt.value
    print(f"[bench] spmv {spmv_ms:.3f} ms, pc {pc_ms:.3f} ms; Krylov leg", file=sys.stderr,
          flush=True)
    if args.warmup > 0:
        run(args.warmup)
    its, dt = run(args.steps)
    stages = stage_breakdown(outer, lib, h, d_x, d_u, d_x, d_y, n_local, min(args.steps, 10), 1,
                             with_pc_stages=False)
    inner_info = gpc.inner.info() if hasattr(gpc, "inner") else {}
    # ---- time to solution (one GPU): b = A x* for a smooth x* (velocity zero on the boundary,
    # pressure blocks of zero mean), zero initial guess, outer FGMRES(10) to rtol 1e-6; says
    # whether a cheaper preconditioner application was bought with more outer iterations
    t_sol = None
    if world == 1:
        th_ = p["th"]
        m2 = 2 * p["m"]
        Xv, Xp = th_.coords_v, th_.coords_p
        sv = np.sin(0.5 * np.pi * Xv[:, 0]) * np.sin(0.5 * np.pi * Xv[:, 1])
        sv = np.concatenate([sv, -0.5 * sv])
        sv[th_.boundary_v] = 0.0
        sq = np.cos(0.5 * np.pi * Xp[:, 0]) * np.cos(np.pi * Xp[:, 1])
        sq -= sq.mean()
        xs = np.concatenate([np.concatenate([(1.0 + 0.1 * k) * sv for k in range(m2)]),
                             np.concatenate([(1.0 - 0.05 * k) * sq for k in range(m2)])])
        d_s, d_b = dvec(xs), dvec()
        outer._ck(lib.kkt_apply_device(h, d_s, d_b))
        outer._ck(lib.kkt_vec_upload(h, d_u, _lib.f64(np.zeros(n_local))[1]))
        outer._ck(lib.kkt_set_krylov(h, 1, -1, 10, 1e-6, 0.0, 1e300, args.tts_max_it))
        its_s, reason, nh, rn = C.c_int(), C.c_int(), C.c_int(), C.c_double()
        outer._ck(lib.kkt_sync(h))
        t0 = time.perf_counter()
        outer._ck(lib.kkt_solve_device(h, d_b, d_u, C.byref(its_s), C.byref(reason),
                                       C.byref(rn), None, 0, C.byref(nh)))
        outer._ck(lib.kkt_sync(h))
        dt_s = time.perf_counter() - t0
        got = np.empty(n_local)
        outer._ck(lib.kkt_vec_download(h, d_u, _lib.f64(got)[1]))
        nv_all = m2 * th_.n_v
        gp = got[nv_all:].reshape(m2, th_.n_p)
        gp = gp - gp.mean(axis=1, keepdims=True)     # pressures are fixed up to a constant
        t_sol = {"rhs": "A x* for a smooth x* (velocity sin sin, pressure cos cos of zero mean)",
                 "stopping_test": f"outer fgmres restart 10, rtol 1e-6, max {args.tts_max_it}",
                 "converged": bool(reason.value > 0), "iterations": int(its_s.value),
                 "seconds": dt_s,
                 "velocity_error": float(np.linalg.norm(got[:nv_all] - xs[:nv_all])
                                         / np.linalg.norm(xs[:nv_all])),
                 "pressure_error": float(np.linalg.norm(gp.ravel() - xs[nv_all:])
                                         / np.linalg.norm(xs[nv_all:]))}
    # the nested velocity solve's block-Schur preconditioner, itemised (5 applications of it and
    # 5 velocity-operator applies make up most of one StokesPC application)
    try:
        inner = gpc.inner
        ih, n_in = inner.handle, inner.info()["n_local"]
        dvi = []
        for seed in (1, 2):
            d = C.c_void_p()
            inner._ck(lib.kkt_vec_alloc(ih, C.byref(d)))
            inner._ck(lib.kkt_vec_upload(ih, d, _lib.f64(common.rng_vector(n_in, seed + rank))[1]))
            dvi.append(d)
        inner._ck(lib.kkt_time_apply(ih, dvi[0], dvi[1], 5, C.byref(ms)))
        inner._ck(lib.kkt_time_apply(ih, dvi[0], dvi[1], 20, C.byref(ms)))
        stages["velocity_preconditioner_application_ms"] = pc_stages(inner, lib, ih, dvi[0], dvi[1])
        stages["velocity_operator_apply_ms"] = ms.value / 20
    except Exception as e:      # noqa: BLE001 -- a side measurement must not lose the line
        stages["velocity_preconditioner_application_ms"] = {"error": f"{type(e).__name__}: {e}"}
    if rank != 0:
        return None
    alg = info["bytes_streamed"]          # what the launch must move (index arrays once)
    achieved = alg / (spmv_ms * 1e-3) / 1e9
    th = p["th"]
    sw = int(info.get("apply_switched", 0))
    roof = {"kernel": ("kkt_spmv_rows_ragged (width-switched: a wave runs the body unrolled for its "
                       "slice's width)" if sw else "kkt_spmv_rows<2,0> (slot loop)") +
                      f" (outer Stokes-control operator, this rank's shard; {sw} of "
                      f"{int(info.get('apply_launches', 0))} launches switched)",
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic_named(
                "traffic_stokes_outer_operator.json", f"stokes2d {n} {n_t} {args.scheme} {args.mode}")
            if world == 1 else None,
            "algorithmic_bytes_per_launch": alg, "launch_ms": spmv_ms,
            "kernel_launches_per_apply": int(info.get("apply_launches", 0)),
            "note": "one operator apply = kernel_launches_per_apply launches of the kernel (block rows "
                    "whose terms have two sparsity structures run in two) + the ConstantNullspace "
                    "corrections of the pressure blocks; bytes, time and traffic are per apply"}
    ok = roofline_check(roof)
    print(json.dumps({
        "metric": "Krylov iterations/s (preconditioned FGMRES(10), all-at-once Stokes-control KKT)",
        "value": its / dt, "unit": "Krylov iterations/s", "n_gpus": world, "steps": its,
        "warmup": args.warmup, "ms_per_step": 1e3 * dt / its, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": (f"2-D Stokes control, Taylor-Hood P2-P1 {n}x{n}, n_t={n_t}, "
                                f"beta={beta:g}, T={args.T:g}, {args.scheme}, mode {args.mode}"),
                   "unknowns": int(2 * p["m"] * (th.n_v + th.n_p)), "n_v": int(th.n_v),
                   "n_p": int(th.n_p),
                   "krylov": "outer fgmres restart 10; inner gmres, 5 iterations per application",
                   "preconditioner": f"StokesPC, Chebyshev (its, emin, emax): {specs}" + (
                       f"; velocity sub-solves {coarse[1]} x [Galerkin correction on "
                       f"{coarse[0].shape[1]} coarse functions + the schur sweeps]" if coarse else "") + (
                       f"; K_p solve {kp_coarse[1]} x [deflated Galerkin correction on "
                       f"{kp_coarse[0].shape[1]} coarse functions + the kp sweeps]" if kp_coarse else ""),
                   "parallelism": f"time-block rows over {world} GPU(s)",
                   "transport": (getattr(comm, "name", "rccl") if world > 1 else "none"),
                   "sweeps": sweep_plan(inner_info),
                   "pc_apply_ms": pc_ms, "kkt_apply_ms": spmv_ms,
                   "time_to_solution": t_sol},
        "stages": stages,
        "roofline": roof}))
    return 0 if ok else 4


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(n, argv, timeout_s):
    """`python bench.py --gpus N` without a launcher: this process starts the N ranks as fresh
    children (it has not touched, and never touches, a GPU), gives them the rendezvous of
    torch.distributed.run (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT), relays rank
    0's JSON line and returns the first non-zero exit code."""
    import subprocess
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                      text=True))
    t0, rc, out0 = time.time(), 0, ""
    try:
        import threading
        buf = []
        reader = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
        reader.start()
        live = set(range(n))
        while live:
            for r in sorted(live):
                code = procs[r].poll()
                if code is not None:
                    live.discard(r)
                    if code != 0 and rc == 0:
                        rc = code
                        print(f"[bench] rank {r} exited with code {code}", file=sys.stderr,
                              flush=True)
            if rc != 0 or time.time() - t0 > timeout_s:
                if rc == 0:
                    rc = 124
                    print(f"[bench] ranks still running after {timeout_s} s: stopping them",
                          file=sys.stderr, flush=True)
                break
            time.sleep(0.2)
        reader.join(timeout=5)
        out0 = buf[0] if buf else ""
    finally:
        for pr in procs:                    # exactly the children started here
            if pr.poll() is None:
                pr.terminate()
        for pr in procs:
            try:
                pr.wait(timeout=10)
            except subprocess.TimeoutExpired:
                pr.kill()
    lines = [ln for ln in out0.splitlines() if ln.strip()]
    if lines:
        print(lines[-1], flush=True)
    elif rc == 0:
        rc = 1
        print("[bench] rank 0 printed no line", file=sys.stderr)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="heat2d", choices=["heat2d", "heat3d", "stokes2d"])
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--n_t", type=int, default=64)
    ap.add_argument("--beta", type=float, default=1.0e-4)
    ap.add_argument("--T", type=float, default=2.0)
    ap.add_argument("--scheme", default="BE", choices=["BE", "CN"])
    ap.add_argument("--mode", default="G", choices=["G", "S"])
    # Chebyshev substitute of the reference's AMG sub-solves: the lightest setting with which
    # GMRES(10) converges on cfg 2 (scripts/cfg2_convergence.py)
    # (CN: 140 sweeps -- 80 to 100 do not converge there, scripts/cfg2_convergence.py --scheme CN)
    ap.add_argument("--schur-its", type=int, default=None)
    ap.add_argument("--schur-emin", type=float, default=None)
    ap.add_argument("--schur-emax", type=float, default=2.1)
    ap.add_argument("--schur-auto", action="store_true",
                    help="degree and per-matrix intervals of the sub-solves from spectrum "
                         "estimates on the device (kkt_pc_desc.schur_its = -1) instead of the flags")
    ap.add_argument("--coarse-cycles", type=int, default=None,
                    help="two-grid form of the Schur sub-solves: cycles of [coarse correction, "
                         "--schur-its sweeps on [--schur-emin, --schur-emax]]; 0: plain Chebyshev "
                         "(80 / 140 sweeps on [7e-4, 2.1]: the preconditioner rounds 1 and 2 "
                         "measured).  Default: 2 on heat2d and stokes2d, 0 on heat3d")
    # stokes2d, measured on 128 x 128 x 32 (profiles/r03/stokes_quality.txt): the outer FGMRES(10)
    # converges only with BOTH accurate velocity sub-solves (2 two-grid cycles on cells of 8 node
    # spacings; plain polynomials of 40 .. 160 sweeps, single cycles and cells of 16 all stall) and
    # a pressure-Laplacian polynomial that resolves the Neumann spectrum (600 sweeps on
    # [2e-4, 2.1]: 100 iterations; 300 on [5e-4, 2.1]: 145; 160 on [2e-3, 2.1]: 195; 80: 265;
    # -1, the spectrum estimate of a singular matrix: stalls)
    # ... or, cheaper per application and at 97-108 outer iterations, its two-grid form: 2 x
    # [Galerkin correction on 33^2 multilinear functions with the constants deflated, 12 sweeps on
    # [0.05, 2.1]] -- 6.4 s to solution against 6.7 s with the 600 plain sweeps (the default)
    ap.add_argument("--kp-its", type=int, default=None,
                    help="stokes2d: sweeps of the pressure-Laplacian solve (-1: from the spectrum); "
                         "default 12 per two-grid cycle, 600 with --kp-coarse-cycles 0")
    ap.add_argument("--kp-emin", type=float, default=None)
    ap.add_argument("--kp-coarse-cycles", type=int, default=2,
                    help="stokes2d: two-grid form of the pressure-Laplacian solve, cycles of [deflated "
                         "Galerkin correction, --kp-its sweeps on [--kp-emin, --schur-emax]]; 0: plain "
                         "polynomial")
    ap.add_argument("--kp-coarse-cell", type=int, default=4)
    ap.add_argument("--tts-max-it", type=int, default=600,
                    help="stokes2d: iteration cap of the time-to-solution solve")
    ap.add_argument("--coarse-cell", type=int, default=8,
                    help="coarse cells of this many mesh widths per axis")
    ap.add_argument("--spmv-reps", type=int, default=50)
    ap.add_argument("--cpu-its", type=int, default=24)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-config4", action="store_true",
                    help="skip the side leg on BASELINE configs[3] (3-D heat 64^3, n_t = 128)")
    ap.add_argument("--no-tile-coordinates", action="store_true",
                    help="do not pass the dof coordinates as the tiling hint of the sweep programs")
    ap.add_argument("--lib-option", action="append", default=[], metavar="KEY=VALUE",
                    help="kkt_set_option on the heat-control handle (A/B runs of kernel forms, e.g. "
                         "apply_xcd=1); printed in config")
    ap.add_argument("--only-spmv", action="store_true",
                    help="time the KKT SpMV only (counter-collection passes)")
    ap.add_argument("--launch-timeout", type=float, default=1700.0,
                    help="self-launched ranks (--gpus N without a launcher) are stopped after this "
                         "many seconds")
    args = ap.parse_args()
    if args.coarse_cycles is None:
        args.coarse_cycles = 0 if args.workload == "heat3d" else 2
    if args.kp_its is None:
        args.kp_its = 12 if args.kp_coarse_cycles > 0 else 600
    if args.kp_emin is None:
        args.kp_emin = 0.05 if args.kp_coarse_cycles > 0 else 0.0002
    # (the Stokes leg hand-sets its velocity sub-solves only when the flags are given)
    args.stokes_schur_its = args.schur_its
    args.stokes_schur_emin = 0.07 if args.schur_emin is None else args.schur_emin
    if args.coarse_cycles > 0 and args.workload != "stokes2d":
        # measured on 256^2 x 64 (profiles/r03, scripts/r03_tts_quality.py): BE 2 x 8 sweeps on
        # [0.07, 2.1] -- 17 iterations to the library's stopping test, and the setting that also
        # converges under fgmres(10) / fgmres(30) (single cycles are faster per iteration, 171
        # its/s, but stagnate under right-preconditioned FGMRES(10)); CN 2 x 16 on [0.03, 2.1]
        if args.schur_its is None:
            args.schur_its = 16 if args.scheme == "CN" else 8
        if args.schur_emin is None:
            args.schur_emin = 0.03 if args.scheme == "CN" else 0.07
    if args.schur_its is None:
        args.schur_its = 140 if args.scheme == "CN" and args.workload != "stokes2d" else 80
    if args.schur_emin is None:
        args.schur_emin = 0.0007

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return spawn_ranks(args.gpus, sys.argv[1:], args.launch_timeout)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    if args.workload == "stokes2d":
        return bench_stokes(args, rank, world, local_rank)
    out = measure_heat(args, rank, world, local_rank, tts=True)
    ok = True
    if rank == 0 and out is not None and args.only_spmv:
        print(json.dumps(out))
        return 0
    if rank == 0 and out is not None:
        ok = roofline_check(out["roofline"])
        p = out.pop("_problem")
        if not args.no_cpu_baseline and world == 1:      # rank 0 at N = 1 only (contract)
            out["cpu_baseline"] = cpu_baseline(p, args.cpu_its, max(1, args.cpu_its // 8))
    # BASELINE configs[3] (3-D heat 64^3 P1, n_t = 128: the 8-GPU configuration, which also fits
    # one GPU) beside the headline line, as its own object, time-sharded like the headline at
    # N > 1 (every rank takes part); failures are reported, not fatal
    if (not args.no_config4 and not args.only_spmv and args.workload == "heat2d"
            and args.n == 256 and args.n_t == 64):
        a4 = argparse.Namespace(**vars(args))
        a4.workload, a4.n, a4.n_t, a4.mode = "heat3d", 64, 128, args.mode
        # suggest_chebyshev on the interior-level block of this configuration (4.6 s of host
        # ARPACK, done once offline): (34, 7.44e-3, 2.093)
        a4.schur_its, a4.schur_emin, a4.schur_emax = 34, 7.44e-3, 2.1
        a4.coarse_cycles = 0      # (3-D: the plain sweeps are at their optimum, profiles/r03)
        a4.steps, a4.warmup, a4.spmv_reps = 10, 2, 10
        o4 = None
        try:
            o4 = measure_heat(a4, rank, world, local_rank, tts=True)
        except Exception as e:      # noqa: BLE001 -- a side leg must not lose the headline
            if rank == 0 and out is not None:
                out["config4"] = {"error": f"{type(e).__name__}: {e}"}
        if rank == 0 and out is not None and o4 is not None:
            roofline_check(o4["roofline"])
            p4 = o4.pop("_problem")
            out["config4"] = {k: o4[k] for k in ("value", "unit", "n_gpus", "steps", "ms_per_step",
                                                 "config", "stages", "roofline", "roofline_sweeps")
                              if k in o4}
            if not args.no_cpu_baseline and world == 1:
                try:        # a 2-iteration sample: an iteration of this system takes ~10 s of CPU
                    out["config4"]["cpu_baseline"] = cpu_baseline(p4, 2, 0)
                except Exception as e:      # noqa: BLE001
                    out["config4"]["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0 and out is not None:
        print(json.dumps(out))
    return 0 if ok else 4


def measure_heat(args, rank, world, local_rank, tts):
    """One heat-control workload: roofline leg (KKT SpMV), preconditioner timings, Krylov leg."""
    from control_amd import _lib
    from control_amd import problems as common
    from control_amd.dist import make_comm

    print(f"[bench] assembling the synthetic system ({args.workload} n={args.n} "
          f"n_t={args.n_t})", file=sys.stderr, flush=True)
    p = build_problem(args)
    comm = make_comm(rank, world, local_rank) if world > 1 else None
    # KKT_DEVICE pins every rank to one GPU (rehearsal of the N > 1 path on a one-GPU box,
    # with KKT_TRANSPORT=gloo); production: one GPU per local rank
    device = int(os.environ.get("KKT_DEVICE", local_rank))
    t_setup = time.perf_counter()
    lib_options = dict(kv.split("=", 1) for kv in getattr(args, "lib_option", []) or [])
    gsys = common.gpu_system(p, device=device, comm=comm, share_values=p["share_values"],
                             tile_coordinates=not args.no_tile_coordinates,
                             **({"options": lib_options} if lib_options else {}))
    lib, h = gsys._lib, gsys.handle
    if not args.only_spmv:       # (counter-collection passes of the operator need no preconditioner)
        gpc = common.gpu_pc(p, p["mass"], p["schur"], coarse=p.get("coarse"))
        gsys._set_pc(gpc)
    gsys._ck(lib.kkt_sync(h))
    t_setup = time.perf_counter() - t_setup     # CSR -> device storage + preconditioner build
    info = gsys.info()
    n_local = info["n_local"]

    def dvec(host=None):
        d = C.c_void_p()
        gsys._ck(lib.kkt_vec_alloc(h, C.byref(d)))
        if host is not None:
            a, pa = _lib.f64(host)
            gsys._ck(lib.kkt_vec_upload(h, d, pa))
        return d

    x = common.rng_vector(n_local, common.SEED + rank)
    d_x, d_y = dvec(x), dvec()

    # ---- roofline leg: the KKT block-row SpMV, HIP events on the library's stream.
    # achieved = bytes the launch must move (every value array once, every index structure
    # once -- they are shared on the device --, x in, y out) / launch time.  The SURVEY 8d CSR
    # formula (12 B per non-zero of every block) is kept as a named side figure: it prices index
    # bytes the device never streams and can exceed the HBM peak.
    ms = C.c_float()
    gsys._ck(lib.kkt_time_apply(h, d_x, d_y, 5, C.byref(ms)))          # warm-up
    gsys._ck(lib.kkt_time_apply(h, d_x, d_y, args.spmv_reps, C.byref(ms)))
    spmv_ms = ms.value / args.spmv_reps
    alg_bytes = info["bytes_streamed"]
    csr_bytes = info["bytes_algorithmic"]
    achieved = alg_bytes / (spmv_ms * 1e-3) / 1e9
    if args.only_spmv:
        workload_ = (f"{'2-D' if args.workload == 'heat2d' else '3-D'} heat control, "
                     f"{args.n}^{2 if args.workload == 'heat2d' else 3} P1, n_t={args.n_t}, "
                     f"beta={args.beta:g}, T={args.T:g}, {args.scheme}, mode {args.mode}")
        return {"config": {"workload": workload_, "unknowns": int(2 * p["m"] * p["sd"].n_dofs)},
                "roofline": {"achieved": achieved, "frac": achieved / HBM_PEAK_GBS,
                             "algorithmic_bytes_per_launch": alg_bytes, "launch_ms": spmv_ms,
                             "csr_formula_bytes_per_launch": csr_bytes}}
    gsys._ck(lib.kkt_time_pc_apply(h, d_x, d_y, 2, C.byref(ms)))
    gsys._ck(lib.kkt_time_pc_apply(h, d_x, d_y, 5, C.byref(ms)))
    pc_ms = ms.value / 5
    # the persistent sweep programs of one application, HIP events around each launch
    sweeps = None
    if world == 1:
        sw_ms, sw_n, sw_ph = C.c_float(), C.c_int(), C.c_int64()
        gsys._ck(lib.kkt_time_pc_sweeps(h, d_x, d_y, C.byref(sw_ms), C.byref(sw_n),
                                        C.byref(sw_ph)))
        if sw_n.value > 0:
            nx = p["sd"].n_dofs
            nnz_block = info["nnz_blocks"] // max(1, info["n_blocks_stored"])
            S_bytes = nnz_block * 12 + (nx + 1) * 4 + 16 * nx      # one spatial SpMV
            alg_sw = sw_ph.value * S_bytes
            sweeps = {
                "kernel": "sweep programs (the time sweeps of one preconditioner application)",
                "bound": "latency (hand-offs between dependent SpMV steps; operands on chip)",
                "launches": sw_n.value, "phases": int(sw_ph.value), "total_ms": sw_ms.value,
                "us_per_phase": 1e3 * sw_ms.value / sw_ph.value,
                "traffic": None, "hbm_GBs": None, "hbm_frac": None,
                "bytes_if_streamed": int(alg_sw),
                "GBs_if_streamed": alg_sw / (sw_ms.value * 1e-3) / 1e9,
                "note": "traffic = HBM bytes of all sweep launches of one application from the "
                        "committed PMC passes, hbm_GBs = traffic / total_ms, hbm_frac = hbm_GBs / "
                        f"{HBM_PEAK_GBS:.0f}: matrix, diagonal and iterates stay on chip (registers "
                        "/ LDS), the launches are bound by the hand-off latency between dependent "
                        "steps, not by HBM.  *_if_streamed = what plain launches would stream: "
                        "SpMV steps x S, S = 12 nnz + 4 (N_x + 1) + 16 N_x of one spatial block "
                        "(SURVEY 8d, B_pc); a side figure, not a fraction of anything"}

    # ---- Krylov leg: W warm-up iterations, then exactly K timed ones
    d_b, d_u = dvec(x), dvec()

    def run(max_it):
        gsys._ck(lib.kkt_vec_upload(h, d_u, _lib.f64(np.zeros(n_local))[1]))
        gsys._ck(lib.kkt_set_krylov(h, 0, -1, 10, 0.0, 0.0, 1e300, max_it))
        its, reason, nh, rn = C.c_int(), C.c_int(), C.c_int(), C.c_double()
        gsys._ck(lib.kkt_comm_barrier(h))
        gsys._ck(lib.kkt_sync(h))
        t0 = time.perf_counter()
        gsys._ck(lib.kkt_solve_device(h, d_b, d_u, C.byref(its), C.byref(reason),
                                      C.byref(rn), None, 0, C.byref(nh)))
        gsys._ck(lib.kkt_sync(h))
        gsys._ck(lib.kkt_comm_barrier(h))
        dt = C.c_double(time.perf_counter() - t0)
        gsys._ck(lib.kkt_comm_max(h, C.byref(dt)))
        return its.value, dt.value

    print(f"[bench] spmv {spmv_ms:.3f} ms, pc {pc_ms:.3f} ms; Krylov leg", file=sys.stderr,
          flush=True)
    if args.warmup > 0:
        run(args.warmup)
    its, dt = run(args.steps)
    if its != args.steps:
        raise RuntimeError(f"the solver ran {its} iterations, {args.steps} were asked for")

    # ---- SURVEY 8e itemisation (outside the timed region)
    stages = stage_breakdown(gsys, lib, h, d_b, d_u, d_x, d_y, n_local, min(args.steps, 20), 0)

    # ---- time to solution: the README right-hand side, library-default stopping test (gmres,
    # restart 10, rtol 1e-6, control.py:3261-3266), at most 300 iterations
    t_sol = None
    if tts:
        sd = p["sd"]
        g0, g1 = readme_rhs(p)
        lo = getattr(gsys, "_lo", 0)
        nloc = info["n_local"] // (2 * sd.n_dofs)
        r0, r1 = g0[lo:lo + nloc], g1[lo:lo + nloc]
        d_rhs = dvec(np.concatenate([r0.ravel(), r1.ravel()]))
        gsys._ck(lib.kkt_vec_upload(h, d_u, _lib.f64(np.zeros(n_local))[1]))
        gsys._ck(lib.kkt_set_krylov(h, 0, -1, 10, 1.0e-6, 0.0, -1.0, 300))
        s_its, s_reason, s_nh, s_rn = C.c_int(), C.c_int(), C.c_int(), C.c_double()
        gsys._ck(lib.kkt_comm_barrier(h))
        gsys._ck(lib.kkt_sync(h))
        t0 = time.perf_counter()
        gsys._ck(lib.kkt_solve_device(h, d_rhs, d_u, C.byref(s_its), C.byref(s_reason),
                                      C.byref(s_rn), None, 0, C.byref(s_nh)))
        gsys._ck(lib.kkt_sync(h))
        gsys._ck(lib.kkt_comm_barrier(h))
        s_dt = C.c_double(time.perf_counter() - t0)
        gsys._ck(lib.kkt_comm_max(h, C.byref(s_dt)))
        t_sol = {"rhs": "README example (v_d = t c, f = c), zero initial guess",
                 "stopping_test": "gmres restart 10, rtol 1e-6 (library default), max 300",
                 "converged": bool(s_reason.value > 0), "iterations": int(s_its.value),
                 "seconds": s_dt.value}
    info_end = gsys.info()       # after every leg: what ran, and whether anything fell back
    last_error = lib.kkt_last_error(h).decode() if info_end.get("program_fallbacks") else ""
    gsys.close()
    if rank != 0:
        return None
    workload = (f"{'2-D' if args.workload == 'heat2d' else '3-D'} heat control, "
                f"{args.n}^{2 if args.workload == 'heat2d' else 3} P1, n_t={args.n_t}, "
                f"beta={args.beta:g}, T={args.T:g}, {args.scheme}, mode {args.mode}")
    frac = achieved / HBM_PEAK_GBS
    plan = sweep_plan(info_end)
    out = {
        "metric": "Krylov iterations/s (preconditioned GMRES(10), all-at-once heat-control KKT)",
        "value": its / dt, "unit": "Krylov iterations/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / its,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": workload,
            "unknowns": int(2 * p["m"] * p["sd"].n_dofs),
            "krylov": "gmres, left preconditioning, restart 10, classical Gram-Schmidt",
            "preconditioner": (f"block Schur: mass Chebyshev {p['mass']}, "
                               + (f"Schur sub-solves {p['coarse'][1]} x [Galerkin correction on "
                                  f"{p['coarse'][0].shape[1]} multilinear coarse functions + "
                                  f"Chebyshev sweeps {p['schur']}]" if p.get("coarse") else
                                  f"Schur Chebyshev {p['schur']}") + " (its, emin, emax)"),
            "parallelism": f"time-block rows over {world} GPU(s)",
            "transport": (getattr(comm, "name", "rccl") if world > 1 else "none"),
            "sweep_tiles": ("boxes from the dof coordinates (kkt_set_tile_coordinates)"
                            if not args.no_tile_coordinates else "bisection of the sparsity graph"),
            "sweeps": plan,
            "pc_apply_ms": pc_ms, "kkt_apply_ms": spmv_ms, "setup_s": t_setup,
            "time_to_solution": t_sol},
        "stages": stages,
        "roofline": {
            "kernel": "kkt_spmv_rows (fused block-row SpMV of the KKT operator)",
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": frac,
            "traffic": measured_traffic(workload) if world == 1 else None,
            "algorithmic_bytes_per_launch": alg_bytes, "launch_ms": spmv_ms,
            "frac_of_measured_copy_rate": achieved / MEASURED_COPY_GBS,
            "csr_formula_bytes_per_launch": csr_bytes,
            "csr_formula_GBs": csr_bytes / (spmv_ms * 1e-3) / 1e9,
            "note": "algorithmic bytes = 8 B per stored non-zero of every value array + 4 B per "
                    "non-zero and 4 B per row of every distinct sparsity structure (blocks of "
                    "equal structure share one index array on the device) + 16 B per unknown; "
                    "csr_formula_* = SURVEY 8d mode-" + args.mode + " formula (12 B per non-zero "
                    "of every block), a side figure that may exceed the peak"},
        "_problem": p,
    }
    if plan["program_fallbacks"]:
        out["config"]["warning"] = ("a persistent sweep program timed out during this run and the "
                                    "preconditioner fell back to plain launches: value and "
                                    "pc_apply_ms measure that form (see config.sweeps): " + last_error)
    if sweeps is not None:
        # HBM-side bytes of one application (all sweep launches) from the committed PMC passes
        per_launch = measured_sweep_traffic(workload, out["config"]["preconditioner"],
                                            sweeps["phases"] // sweeps["launches"])
        if per_launch is not None:
            sweeps["traffic"] = per_launch * sweeps["launches"]
            sweeps["hbm_GBs"] = sweeps["traffic"] / (sweeps["total_ms"] * 1e-3) / 1e9
            sweeps["hbm_frac"] = sweeps["hbm_GBs"] / HBM_PEAK_GBS
        out["roofline_sweeps"] = sweeps
    return out


if __name__ == "__main__":
    sys.exit(main() or 0)
