// Preconditioner wrapper and Krylov loop on device-resident vectors.
//   System::pc_apply  <- Preconditioner.apply      (reference preconditioner.py:562-656)
//   System::solve     <- MultiBlockSystem.solve    (preconditioner.py:658-766) with the PETSc
//                        KSPSolve_GMRES / KSPSolve_FGMRES loop it calls at :758-759 restated:
//                        classical Gram-Schmidt without refinement, Givens rotations,
//                        KSPConvergedDefault against the norm of the (preconditioned) rhs
//                        because the initial guess is flagged non-zero (:743).
// All vector work is HIP kernels on the system's stream; the host sees one small
// device-to-host copy per iteration (the new Hessenberg column and the norm).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <string>
#include <vector>

#include "comm.hpp"
#include "pc.hpp"
#include "system.hpp"

namespace kkt {

void System::mdot(const double *w, const double *const *V, int nv, double *d_out) {
    for (int g = 0; g < nv; g += MDOT_MAX) {
        const int m = std::min(MDOT_MAX, nv - g);
        VecList L{};
        for (int i = 0; i < m; ++i) L.v[i] = V[g + i];
        launch_mdot(stream, w, L, m, n_local, d_red_scratch, d_out + g);
    }
    if (sharded) {
        clock.mark(stream, StageClock::ORTH);
        comm->allreduce_sum(d_out, nv, stream);
        clock.mark(stream, StageClock::ALLREDUCE);
    }
}

void System::norm2(const double *w, double *d_out) {
    const double *V[1] = {w};
    mdot(w, V, 1, d_out + 1);
    launch_norm2_finish(stream, d_out + 1, d_out);
}

void System::ensure_workspace(int restart, bool flexible) {
    if (d_V && ws_restart >= restart && (ws_flexible || !flexible)) return;
    auto F = [](double *&p) {
        if (p) (void)hipFree(p);
        p = nullptr;
    };
    F(d_V);
    F(d_Z);
    d_V = dev_alloc<double>((size_t)(restart + 1) * vec_stride);
    HIPCHK(hipMemsetAsync(d_V, 0, (size_t)(restart + 1) * vec_stride * 8, stream));
    if (flexible) {
        d_Z = dev_alloc<double>((size_t)restart * vec_stride);
        HIPCHK(hipMemsetAsync(d_Z, 0, (size_t)restart * vec_stride * 8, stream));
    }
    if (!d_t1) d_t1 = new_vec();
    if (!d_t2) d_t2 = new_vec();
    if (!d_rhs) d_rhs = new_vec();
    if (!d_red_scratch) d_red_scratch = dev_alloc<double>((size_t)REDUCE_BLOCKS * MDOT_MAX);
    F(d_hcol);
    F(d_coef);
    d_hcol = dev_alloc<double>(restart + 16);
    d_coef = dev_alloc<double>(restart + 16);
    if (h_pinned) (void)hipHostFree(h_pinned);
    HIPCHK(hipHostMalloc((void **)&h_pinned, (restart + 16) * sizeof(double), 0));
    ws_restart = restart;
    ws_flexible = flexible;
}

static void per_var(const System &S, double *y, const double *x, const double *mx,
                    const MaskJob *jobs) {
    if (S.nx0 == S.nx1) {
        launch_mask_blocks(S.stream, y, x, mx, jobs, S.n0_loc + S.n1_loc, S.nx0);
    } else {
        const int64_t o = (int64_t)S.n0_loc * S.nx0;
        launch_mask_blocks(S.stream, y, x, mx, jobs, S.n0_loc, S.nx0);
        launch_mask_blocks(S.stream, y + o, x + o, mx ? mx + o : nullptr, jobs + S.n0_loc,
                           S.n1_loc, S.nx1);
    }
}

void System::ns_project(double *y, const double *x) {
    per_var(*this, y, x, nullptr, d_mask_jobs);
    if (any_const_ns)
        launch_const_correct(stream, d_const_jobs, n_const_jobs, const_max_nx, y, nullptr, 0, d_sums);
}

void System::ns_pc_post(double *y, const double *u, const double *b) {
    per_var(*this, y, u, b, d_mask_jobs_one);
    if (any_const_ns)
        launch_const_correct(stream, d_const_jobs, n_const_jobs, const_max_nx, y, b, 1, d_sums);
}

// y = P pc_fn(P x) + (I - P) x   (preconditioner.py:562-656)
void System::pc_apply(const double *d_x, double *d_y) {
    if (!finalized) fail(KKT_ERR_STATE, "system not finalized");
    info.last_pc_applies++;
    if (pc && pc_stale) {
        pc->values_changed();
        pc_stale = false;
    }
    double *in, *out;
    if (pc) {
        in = pc->in();
        out = pc->out();
    } else {
        if (!d_pc_in) d_pc_in = new_vec();
        if (!d_pc_out) d_pc_out = new_vec();
        in = d_pc_in;
        out = d_pc_out;
    }
    ns_project(in, d_x);   // b_c = pc_pre_mult_corrected(b)
    if (pc) {
        pc->run();
    } else if (pc_cb) {
        std::vector<double> hb(n_local), hu(n_local, 0.0);
        HIPCHK(hipMemcpyAsync(hb.data(), in, n_local * 8, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        const int64_t o = (int64_t)n0_loc * nx0;
        if (pc_cb(pc_cb_user, hb.data(), hb.data() + o, hu.data(), hu.data() + o) != 0) {
            pc_cb_failed = true;   // the reference's _error_flag (preconditioner.py:64-72)
        }
        HIPCHK(hipMemcpyAsync(out, hu.data(), n_local * 8, hipMemcpyHostToDevice, stream));
        HIPCHK(hipStreamSynchronize(stream));
    } else {
        launch_copy(stream, out, in, n_local);   // default pc_fn: u = b (:342-345)
    }
    ns_pc_post(d_y, out, d_x);
}

// measurement variant of pc_apply: built-in preconditioner run as plain launches with events
// around its persistent programs (kkt_time_pc_sweeps)
void System::pc_apply_timed(const double *d_x, double *d_y, float *ms, int *launches,
                            int64_t *phases) {
    if (!finalized) fail(KKT_ERR_STATE, "system not finalized");
    if (!pc) fail(KKT_ERR_STATE, "no built-in preconditioner");
    if (sharded) fail(KKT_ERR_STATE, "kkt_time_pc_sweeps on a time-sharded handle");
    if (pc_stale) {
        pc->values_changed();
        pc_stale = false;
    }
    ns_project(pc->in(), d_x);
    pc->time_programs(ms, launches, phases);
    ns_pc_post(d_y, pc->out(), d_x);
}

void System::pc_apply_timed_stages(const double *d_x, double *d_y, kkt_pc_stage_times *out) {
    if (!finalized) fail(KKT_ERR_STATE, "system not finalized");
    if (!pc) fail(KKT_ERR_STATE, "no built-in preconditioner");
    if (pc_stale) {
        pc->values_changed();
        pc_stale = false;
    }
    ns_project(pc->in(), d_x);
    pc->time_stages(out);
    ns_pc_post(d_y, pc->out(), d_x);
}

// The persistent sweep programs spin with a bound; a time-out leaves an error word on the device.
// On time shards the decision to fall back must be the same on every rank (the preconditioner's
// hand-offs and the Krylov all-reduces are collective): the flags are combined over the ranks.
bool System::pc_timed_out_agreed(std::string *why) {
    if (!pc) return false;
    std::string local_why;
    const bool local = pc->timed_out(&local_why);
    bool any = local;
    if (sharded && comm) any = comm->max_host(local ? 1.0 : 0.0, stream) > 0.5;
    if (any && why)
        *why = local ? local_why
                     : std::string("persistent sweep kernel timed out on another rank of the time shard");
    return any;
}

bool System::pc_fallback_plain(const std::string &why) {
    if (!pc || !pc->fallback_plain()) return false;
    ++program_fallbacks;
    info.program_fallbacks = program_fallbacks;
    err = why + "; continued with plain launches";
    if (program_fallbacks == 1)
        std::fprintf(stderr, "[kkt] %s\n", err.c_str());
    return true;
}

namespace {

struct Conv {   // KSPConvergedDefault
    double rtol, atol, divtol, rnorm0, ttol;
    Conv(double rtol_, double atol_, double divtol_, double rnorm0_)
        : rtol(rtol_), atol(atol_), divtol(divtol_), rnorm0(rnorm0_),
          ttol(std::max(rtol_ * rnorm0_, atol_)) {}
    int operator()(double rn) {
        // "handle special case of zero RHS and nonzero guess": the first residual norm takes
        // the place of the (zero) right-hand-side norm
        if (rnorm0 == 0.0 && std::isfinite(rn)) {
            rnorm0 = rn;
            ttol = std::max(rtol * rn, atol);
        }
        if (!std::isfinite(rn)) return KKT_DIVERGED_NANORINF;
        if (rn <= ttol) return rn < atol ? KKT_CONVERGED_ATOL : KKT_CONVERGED_RTOL;
        if (rn >= divtol * rnorm0) return KKT_DIVERGED_DTOL;
        return 0;
    }
};

}  // namespace

// Preconditioned MINRES, the classic KSPSolve_MINRES of PETSc (<= 3.18): Lanczos on A in the
// B inner product, one Givens rotation per step, monitored norm ||B r_0|| prod |s_k|.  B must
// be symmetric positive definite (a user callback or the identity: the block-Schur
// descriptors are block triangular and are reported as KKT_DIVERGED_INDEFINITE_PC when they
// produce r.Br < 0).  Never used by the reference; reachable through the same
// "linear_solver" string, so it is here.  Two host reads per iteration (alpha, then r.z).
void System::solve_minres(const double *d_b, double *d_u, int *its_out, int *reason_out,
                          double *rnorm_out, double *hist, int hist_cap, int *hist_len) {
    if (ksp.pc_side == KKT_PC_RIGHT) fail(KKT_ERR_ARG, "minres supports left preconditioning only");
    ensure_workspace(std::max(ksp.restart, 9), false);
    pc_cb_failed = false;
    info.last_pc_applies = 0;
    info.last_op_applies = 0;
    auto Vp = [&](int k) { return d_V + (size_t)k * vec_stride; };
    double *R = Vp(0), *Z = Vp(1), *U = Vp(2), *V = Vp(3), *W = Vp(4), *UOLD = Vp(5),
           *VOLD = Vp(6), *WOLD = Vp(7), *WOOLD = Vp(8);
    auto dot = [&](const double *a, const double *c) {
        const double *L[1] = {c};
        mdot(a, L, 1, d_hcol);
        HIPCHK(hipMemcpyAsync(h_pinned, d_hcol, sizeof(double), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        return h_pinned[0];
    };
    auto axpy = [&](double *y, double a, const double *x) { launch_axpby(stream, y, a, x, 1.0, n_local); };
    auto scaled = [&](double *y, double a, const double *x) { launch_axpby(stream, y, a, x, 0.0, n_local); };
    int nh = 0;
    auto log = [&](double rn) {
        if (hist && nh < hist_cap) hist[nh] = rn;
        ++nh;
    };
    ns_project(d_u, d_u);
    ns_project(d_rhs, d_b);
    const double *b = d_rhs;
    sync();
    const auto t_begin = std::chrono::steady_clock::now();
    const double haptol = 1.0e-50;

    pc_apply(b, Z);
    Conv conv(ksp.rtol, ksp.atol, ksp.divtol, std::sqrt(dot(Z, Z)));
    for (double *p : {UOLD, VOLD, W, WOLD, WOOLD}) launch_fill(stream, p, 0.0, n_local);
    apply(d_u, R);
    launch_axpby(stream, R, 1.0, b, -1.0, n_local);   // r = b - A x
    pc_apply(R, Z);
    double np = std::sqrt(dot(Z, Z));
    double dp = dot(R, Z);
    int its = 0, reason = 0;
    if (dp < haptol && np > haptol) {
        reason = KKT_DIVERGED_INDEFINITE_PC;
    } else {
        log(np);
        reason = conv(np);
    }
    if (!reason) {
        double beta = std::sqrt(std::fabs(dp)), eta = beta;
        double c = 1.0, cold = 1.0, s = 0.0, sold = 0.0;
        scaled(V, 1.0 / beta, R);
        scaled(U, 1.0 / beta, Z);
        int i = 0;
        while (i < ksp.max_it) {
            its = i + 1;
            apply(U, R);
            const double alpha = dot(U, R);
            pc_apply(R, Z);
            axpy(R, -alpha, V);
            axpy(Z, -alpha, U);
            axpy(R, -beta, VOLD);
            axpy(Z, -beta, UOLD);
            const double betaold = beta;
            dp = dot(R, Z);
            beta = std::sqrt(std::fabs(dp));
            const double coold = cold, soold = sold;
            cold = c;
            sold = s;
            const double rho0 = cold * alpha - coold * sold * betaold;
            const double rho1 = std::sqrt(rho0 * rho0 + beta * beta);
            const double rho2 = sold * alpha + coold * cold * betaold;
            const double rho3 = soold * betaold;
            c = rho0 / rho1;
            s = beta / rho1;
            {   // w_oold <- w_old, w_old <- w; the new w goes into the buffer w_oold leaves
                double *scratch = WOOLD;
                WOOLD = WOLD;
                WOLD = W;
                W = scratch;
            }
            launch_copy(stream, W, U, n_local);
            axpy(W, -rho2, WOLD);
            axpy(W, -rho3, WOOLD);
            scaled(W, 1.0 / rho1, W);
            axpy(d_u, c * eta, W);
            if (dp < haptol) {   // converged or indefinite operator: true residual norm
                apply(d_u, VOLD);
                axpy(VOLD, -1.0, b);
                np = std::sqrt(dot(VOLD, VOLD));
            } else {
                np *= std::fabs(s);
            }
            log(np);
            reason = conv(np);
            if (reason) break;
            if (dp < haptol) {
                reason = KKT_DIVERGED_INDEFINITE_MAT;
                break;
            }
            eta = -s * eta;
            std::swap(VOLD, V);
            std::swap(UOLD, U);
            scaled(V, 1.0 / beta, R);
            scaled(U, 1.0 / beta, Z);
            ++i;
        }
        if (!reason) reason = KKT_DIVERGED_ITS;
    }
    ns_project(d_u, d_u);
    sync();
    if (pc) pc->check();
    info.last_solve_ms =
        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    if (its_out) *its_out = its;
    if (reason_out) *reason_out = reason;
    if (rnorm_out) *rnorm_out = np;
    if (hist_len) *hist_len = nh;
    if (pc_cb_failed) fail(KKT_ERR_CALLBACK, "Error encountered in preconditioner callback");
}

namespace {
struct ProgramTimeout {
    std::string why;
};
}  // namespace

// A sweep program that timed out (bounded spins; the kernel runs to its end, results invalid)
// is not fatal: the preconditioner is rebuilt as plain launches and the solve starts over from
// the caller's initial guess.
void System::solve(const double *d_b, double *d_u, int *its_out, int *reason_out,
                   double *rnorm_out, double *hist, int hist_cap, int *hist_len) {
    if (!finalized) fail(KKT_ERR_STATE, "system not finalized");
    double *u0 = nullptr;
    if (pc) {
        if (!d_guess) d_guess = new_vec();
        u0 = d_guess;
        launch_copy(stream, u0, d_u, n_local);
    }
    // the step-lock hook is consumed by this solve: its host arrays need not outlive it
    struct LockGuard {
        kkt_steplock &l;
        ~LockGuard() { l = kkt_steplock{}; }
    } lock_guard{steplock};
    try {
        solve_once(d_b, d_u, its_out, reason_out, rnorm_out, hist, hist_cap, hist_len);
    } catch (const ProgramTimeout &t) {
        // (thrown on every rank of a time shard at the same point of the iteration)
        if (!pc_fallback_plain(t.why)) fail(KKT_ERR_HIP, t.why);
        launch_copy(stream, d_u, u0, n_local);
        try {
            solve_once(d_b, d_u, its_out, reason_out, rnorm_out, hist, hist_cap, hist_len);
        } catch (const ProgramTimeout &t2) {
            fail(KKT_ERR_HIP, t2.why);
        }
    }
    info.program_fallbacks = program_fallbacks;
}

void System::solve_once(const double *d_b, double *d_u, int *its_out, int *reason_out,
                        double *rnorm_out, double *hist, int hist_cap, int *hist_len) {
    if (ksp.type == KKT_KSP_MINRES)
        return solve_minres(d_b, d_u, its_out, reason_out, rnorm_out, hist, hist_cap, hist_len);
    const bool flexible = ksp.type == KKT_KSP_FGMRES;
    bool right = flexible;
    if (!flexible && ksp.pc_side == KKT_PC_RIGHT) right = true;
    if (flexible && ksp.pc_side == KKT_PC_LEFT)
        fail(KKT_ERR_ARG, "fgmres supports right preconditioning only");
    const int m = ksp.restart;
    ensure_workspace(m, flexible);
    pc_cb_failed = false;
    info.last_pc_applies = 0;
    info.last_op_applies = 0;
    {
        const char *st = opt("stage_timers");
        clock.on = st && st[0] == '1';
    }
    // operator / preconditioner applications bracketed by stage marks
    auto apply = [&](const double *x, double *y) {
        clock.mark(stream, StageClock::OTHER);
        this->apply(x, y);
        clock.mark(stream, StageClock::OP);
    };
    auto pc_apply = [&](const double *x, double *y) {
        clock.mark(stream, StageClock::OTHER);
        this->pc_apply(x, y);
        clock.mark(stream, StageClock::PC);
    };
    auto Vp = [&](int k) { return d_V + (size_t)k * vec_stride; };
    auto Zp = [&](int k) { return d_Z + (size_t)k * vec_stride; };
    auto read_scalars = [&](const double *d_src, int n) {
        HIPCHK(hipMemcpyAsync(h_pinned, d_src, n * sizeof(double), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
    };
    int nh = 0;
    auto log = [&](double rn) {
        if (hist && nh < hist_cap) hist[nh] = rn;
        ++nh;
    };

    // corrected initial guess and right-hand side (preconditioner.py:658-704)
    ns_project(d_u, d_u);
    ns_project(d_rhs, d_b);
    const double *b = d_rhs;

    sync();
    const auto t_begin = std::chrono::steady_clock::now();
    clock.begin(stream);

    // rnorm0 = norm of the (preconditioned) right-hand side
    double rnorm0;
    if (right) {
        norm2(b, d_hcol);
    } else {
        pc_apply(b, d_t2);
        norm2(d_t2, d_hcol);
    }
    read_scalars(d_hcol, 1);
    rnorm0 = h_pinned[0];
    Conv conv(ksp.rtol, ksp.atol, ksp.divtol, rnorm0);
    const double haptol = 1.0e-30;

    std::vector<double> H((size_t)(m + 1) * m, 0.0), cc(m, 0.0), ss(m, 0.0), grs(m + 1, 0.0);
    auto Hm = [&](int r, int c) -> double & { return H[(size_t)c * (m + 1) + r]; };
    int its = 0, reason = 0;
    double rn = 0.0;
    while (true) {
        // KSPInitialResidual
        apply(d_u, d_t1);
        if (right) {
            launch_copy(stream, Vp(0), b, n_local);
            launch_axpby(stream, Vp(0), -1.0, d_t1, 1.0, n_local);
        } else {
            launch_copy(stream, d_t2, b, n_local);
            launch_axpby(stream, d_t2, -1.0, d_t1, 1.0, n_local);
            pc_apply(d_t2, Vp(0));
        }
        norm2(Vp(0), d_hcol);
        read_scalars(d_hcol, 1);
        rn = h_pinned[0];
        log(rn);
        if (rn == 0.0) {
            reason = KKT_CONVERGED_ATOL;
            break;
        }
        launch_scale_inv(stream, Vp(0), Vp(0), d_hcol, n_local);
        std::fill(grs.begin(), grs.end(), 0.0);
        grs[0] = rn;
        reason = conv(rn);
        int it = 0;
        std::fill(H.begin(), H.end(), 0.0);
        while (!reason && it < m && its < ksp.max_it) {
            if (it) log(rn);
            double *w = Vp(it + 1);
            const bool locked = steplock.n_steps > 0 && its < steplock.n_steps;
            if (locked) {
                // step-locked parity: this step starts from the caller's basis
                if (steplock.restart != m) fail(KKT_ERR_ARG, "step-lock restart differs from the KSP's");
                const double *src = steplock.V + (size_t)its * (m + 1) * n_local;
                for (int k = 0; k <= it; ++k)
                    HIPCHK(hipMemcpyAsync(Vp(k), src + (size_t)k * n_local, n_local * 8,
                                          hipMemcpyHostToDevice, stream));
                HIPCHK(hipStreamSynchronize(stream));
            }
            if (right) {
                double *z = flexible ? Zp(it) : d_t2;
                pc_apply(Vp(it), z);
                apply(z, w);
            } else {
                apply(Vp(it), d_t1);
                pc_apply(d_t1, w);
            }
            // classical Gram-Schmidt: h = V^T w; w -= V h; tt = ||w||
            std::vector<const double *> Vl(it + 1);
            for (int k = 0; k <= it; ++k) Vl[k] = Vp(k);
            clock.mark(stream, StageClock::OTHER);
            mdot(w, Vl.data(), it + 1, d_hcol);
            // the last update pass also leaves the partial sums of ||w||^2 (one pass over w
            // less; same chunks and summation order as mdot_stage1, so tt is bitwise the norm a
            // separate pass would give)
            double *d_tt = d_hcol + (it + 1);   // [tt, scratch]
            for (int g = 0; g <= it; g += MDOT_MAX) {
                const int mm = std::min(MDOT_MAX, it + 1 - g);
                VecList L{};
                for (int i = 0; i < mm; ++i) L.v[i] = Vl[g + i];
                if (g + MDOT_MAX > it)
                    launch_maxpy_norm(stream, w, L, d_hcol + g, -1.0, mm, n_local, d_red_scratch,
                                      d_tt + 1);
                else
                    launch_maxpy(stream, w, L, d_hcol + g, -1.0, mm, n_local);
            }
            clock.mark(stream, StageClock::ORTH);
            // time shards: the preconditioner's time-out word rides on this all-reduce, so that
            // every rank sees a time-out of any rank in the same iteration
            const unsigned *d_flag = (sharded && pc) ? pc->err_word() : nullptr;
            if (sharded) {
                launch_flag_to_double(stream, d_flag, d_tt + 2);
                comm->allreduce_sum(d_tt + 1, 2, stream);
                clock.mark(stream, StageClock::ALLREDUCE);
            }
            launch_norm2_finish(stream, d_tt + 1, d_tt);
            // (norm2_finish writes d_tt[0] and leaves d_tt[1]; the flag sits in d_tt[2])
            read_scalars(d_hcol, it + 2 + (sharded ? 2 : 0));
            if (pc) {
                std::string why;
                if (sharded) {
                    if (h_pinned[it + 3] > 0.0) {
                        if (!pc->timed_out(&why))
                            why = "persistent sweep kernel timed out on another rank of the time shard";
                        throw ProgramTimeout{why};
                    }
                } else if (pc->timed_out(&why)) {
                    throw ProgramTimeout{why};
                }
            }
            for (int k = 0; k <= it; ++k) Hm(k, it) = h_pinned[k];
            const double tt = h_pinned[it + 1];
            const double hapbnd = std::min(std::fabs(tt / grs[it]), haptol);
            const bool hapend = tt < hapbnd;
            if (!hapend) launch_scale_inv(stream, w, w, d_tt, n_local);
            if (locked) {
                double *ho = steplock.h + (size_t)its * (m + 2);
                for (int k = 0; k <= it + 1; ++k) ho[k] = h_pinned[k];
                HIPCHK(hipMemcpyAsync(steplock.v_next + (size_t)its * n_local, w, n_local * 8,
                                      hipMemcpyDeviceToHost, stream));
                HIPCHK(hipStreamSynchronize(stream));
            }
            Hm(it + 1, it) = tt;
            // KSPGMRESUpdateHessenberg
            for (int j = 0; j < it; ++j) {
                const double t = Hm(j, it);
                Hm(j, it) = cc[j] * t + ss[j] * Hm(j + 1, it);
                Hm(j + 1, it) = cc[j] * Hm(j + 1, it) - ss[j] * t;
            }
            if (!hapend) {
                const double t = std::sqrt(Hm(it, it) * Hm(it, it) + Hm(it + 1, it) * Hm(it + 1, it));
                if (t == 0.0) {
                    reason = KKT_DIVERGED_BREAKDOWN;
                    break;
                }
                cc[it] = Hm(it, it) / t;
                ss[it] = Hm(it + 1, it) / t;
                grs[it + 1] = -(ss[it] * grs[it]);
                grs[it] = cc[it] * grs[it];
                Hm(it, it) = cc[it] * Hm(it, it) + ss[it] * Hm(it + 1, it);
                rn = std::fabs(grs[it + 1]);
            } else {
                rn = 0.0;
            }
            ++it;
            ++its;
            reason = conv(rn);
            if (hapend && !reason) reason = KKT_DIVERGED_BREAKDOWN;
        }
        if (it && (reason || its >= ksp.max_it)) log(rn);
        // KSPGMRESBuildSoln
        if (it > 0) {
            std::vector<double> y(it, 0.0);
            for (int k = it - 1; k >= 0; --k) {
                double s = grs[k];
                for (int j = k + 1; j < it; ++j) s -= Hm(k, j) * y[j];
                y[k] = s / Hm(k, k);
            }
            HIPCHK(hipMemcpyAsync(d_coef, y.data(), it * sizeof(double), hipMemcpyHostToDevice,
                                  stream));
            HIPCHK(hipStreamSynchronize(stream));   // y is a stack-lifetime host buffer
            double *acc = d_u;
            if (right && !flexible) {
                launch_fill(stream, d_t1, 0.0, n_local);
                acc = d_t1;
            }
            for (int g = 0; g < it; g += MDOT_MAX) {
                const int mm = std::min(MDOT_MAX, it - g);
                VecList L{};
                for (int i = 0; i < mm; ++i) L.v[i] = flexible ? Zp(g + i) : Vp(g + i);
                launch_maxpy(stream, acc, L, d_coef + g, 1.0, mm, n_local);
            }
            if (right && !flexible) {
                pc_apply(d_t1, d_t2);
                launch_axpby(stream, d_u, 1.0, d_t2, 1.0, n_local);
            }
        }
        if (reason) break;
        if (its >= ksp.max_it) {
            reason = KKT_DIVERGED_ITS;
            break;
        }
    }
    // corrected solution (preconditioner.py:761-766)
    ns_project(d_u, d_u);
    clock.mark(stream, StageClock::OTHER);
    sync();
    if (pc) {
        std::string why;
        if (pc_timed_out_agreed(&why)) throw ProgramTimeout{why};
    }
    info.last_solve_ms =
        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    clock.finish(stage_times);
    stage_times.iterations = its;
    stage_times.operator_applies = info.last_op_applies;
    stage_times.pc_applies = info.last_pc_applies;
    if (its_out) *its_out = its;
    if (reason_out) *reason_out = reason;
    if (rnorm_out) *rnorm_out = rn;
    if (hist_len) *hist_len = nh;
    if (pc_cb_failed) fail(KKT_ERR_CALLBACK, "Error encountered in preconditioner callback");
}

}  // namespace kkt
