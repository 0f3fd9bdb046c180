// Preconditioner of the incompressible control systems, on the device.
#include <cmath>
#include <cstring>

#include "pc.hpp"

namespace kkt {

static VRef vabs(const double *p) {
    return p ? VRef{(int64_t)(uintptr_t)p, 0, 0} : VRef{0, -1, 0};
}

static RowOp base_op(const Pattern &P) {
    RowOp op{};
    op.col = P.d_col;
    op.perm = P.d_perm;
    op.slice_off = P.d_slice_off;
    op.uniform_w = P.uniform_w;
    op.nrows = (int32_t)P.nrows;
    op.nslices = P.nslices;
    op.y = op.y2 = op.yin = op.z = op.mx = op.b = op.pk = op.pkm1 = vabs(nullptr);
    op.rowmask = nullptr;
    op.ca = 1.0;
    op.post1 = op.post2 = 1.0;
    return op;
}

static RowLaunch upload_launch(const Pattern &P, std::vector<RowOp> &ops) {
    RowLaunch L;
    L.nops = (int)ops.size();
    L.max_slices = P.nslices;
    L.R = P.R;
    L.uniform_w = P.uniform_w;
    // every op applies one and the same matrix: the four-blocks-per-thread form applies
    L.shared_matrix = !ops.empty();
    for (const RowOp &op : ops)
        L.shared_matrix = L.shared_matrix && op.nterms == 1 && op.t[0].vals == ops[0].t[0].vals &&
                          op.col == ops[0].col && op.rowmask == ops[0].rowmask;
    L.d_ops = dev_upload(ops.data(), ops.size());
    return L;
}

StokesPC::DevMat StokesPC::upload(int64_t nrows, int64_t ncols, const int32_t *ip, const int32_t *ix,
                                  const double *v, bool want_dinv) {
    if (!ip || !ix || !v) fail(KKT_ERR_ARG, "kkt_set_pc_stokes: matrix missing");
    DevMat A;
    A.pat = S_.find_or_add_pattern(nrows, ncols, ip, ix);
    const Pattern &P = S_.patterns[A.pat];
    double *d_csr = dev_upload(v, (size_t)P.nnz);
    A.vals = dev_alloc<double>(P.npadded);
    owned_.push_back(A.vals);
    launch_csr_to_sell(S_.stream, d_csr, P.d_sell2csr, A.vals, P.npadded);
    if (want_dinv) {
        A.dinv = dev_alloc<double>(nrows);
        owned_.push_back(A.dinv);
        launch_extract_dinv(S_.stream, P.d_col, P.d_slice_off, A.vals, nullptr, A.dinv, (int)nrows,
                            P.nslices, P.R, P.d_perm);
    }
    HIPCHK(hipStreamSynchronize(S_.stream));
    HIPCHK(hipFree(d_csr));
    return A;
}

// KSPSolve_Chebyshev + PCJACOBI on all 2n pressure blocks in lock step (its == 0: Jacobi)
void StokesPC::emit_cheb(std::vector<RowLaunch> &dst, const DevMat &A, int its, double emin,
                         double emax, const double *b, double *out) {
    const Pattern &P = S_.patterns[A.pat];
    const int nb = 2 * n_;
    auto blk = [&](const double *base, int k) { return base + (int64_t)k * np_; };
    auto target = [&](int step, int k) -> double * {
        return step == std::max(its, 1) ? out + (int64_t)k * np_
                                         : P_[(step - 1) % 3] + (int64_t)k * np_;
    };
    std::vector<RowOp> ops(nb);
    const double scale = its > 0 ? 2.0 / (emax + emin) : 1.0;
    for (int k = 0; k < nb; ++k) {   // step 1: p_1 = scale D^-1 b (its == 0: D^-1 b)
        RowOp op = base_op(P);
        op.mode = EPI_CHEB;
        op.nterms = 0;
        op.uniform_w = 0;
        op.y = vabs(target(1, k));
        op.b = vabs(blk(b, k));
        op.dinv = A.dinv;
        op.c3 = scale;
        ops[k] = op;
    }
    dst.push_back(upload_launch(P, ops));
    if (its <= 1) return;
    const double alpha = 1.0 - scale * emin, mu = 1.0 / alpha, omegaprod = 2.0 / alpha;
    double c_km1 = 1.0, c_k = mu;
    for (int step = 2; step <= its; ++step) {
        const double c_kp1 = 2.0 * mu * c_k - c_km1;
        const double omega = omegaprod * c_k / c_kp1;
        for (int k = 0; k < nb; ++k) {
            RowOp op = base_op(P);
            op.mode = EPI_CHEB;
            op.nterms = 1;
            op.t[0].vals = A.vals;
            op.t[0].x = vabs(target(step - 1, k));
            op.y = vabs(target(step, k));
            op.b = vabs(blk(b, k));
            op.pk = vabs(target(step - 1, k));
            op.pkm1 = vabs(step >= 3 ? target(step - 2, k) : nullptr);
            op.dinv = A.dinv;
            op.c1 = 1.0 - omega;
            op.c2 = omega;
            op.c3 = scale * omega;
            ops[k] = op;
        }
        dst.push_back(upload_launch(P, ops));
        c_km1 = c_k;
        c_k = c_kp1;
    }
}

StokesPC::StokesPC(System &outer, System &inner, System &commutator, const kkt_pc_stokes_desc &d)
    : S_(outer), inner_(inner), comm_(commutator) {
    if (!outer.finalized || !inner.finalized || !commutator.finalized)
        fail(KKT_ERR_STATE, "kkt_set_pc_stokes needs three finalized systems");
    if (outer.device != inner.device || outer.device != commutator.device)
        fail(KKT_ERR_ARG, "all three systems must live on one GPU");
    cn_ = d.cn != 0;
    nv_ = d.nv;
    np_ = d.np;
    const int m_global = d.n_p_blocks;
    if (m_global < 1 || outer.n0 != 2 * m_global || outer.n1 != 2 * m_global ||
        outer.nx0 != nv_ || outer.nx1 != np_)
        fail(KKT_ERR_ARG, "outer system layout does not match the descriptor");
    // Time sharding (BASELINE configs[4] names 8 GPUs): the outer system is sharded by
    // levels of its two block families, the velocity and commutator systems by their levels --
    // the same [lo, hi) on one rank.  Everything of this preconditioner except the nested solve
    // and the commutator product is per pressure block, hence local; those two exchange their
    // own halos.  n_ below is the LOCAL number of pressure blocks per family.
    if (outer.sharded != inner.sharded || outer.sharded != commutator.sharded)
        fail(KKT_ERR_STATE, "outer, velocity and commutator systems must be sharded alike");
    if (outer.sharded) {
        if (outer.families != 2 || inner.families != 1 || commutator.families != 1 ||
            outer.lo != inner.lo || outer.hi != inner.hi || outer.lo != commutator.lo ||
            outer.hi != commutator.hi)
            fail(KKT_ERR_STATE, "shards of the three systems do not match");
    }
    n_ = outer.sharded ? outer.hi - outer.lo : m_global;
    if (inner.n_local != 2 * (int64_t)n_ * nv_)
        fail(KKT_ERR_ARG, "inner (velocity) system has the wrong size");
    if (commutator.n_local != 2 * (int64_t)n_ * np_)
        fail(KKT_ERR_ARG, "commutator (pressure) system has the wrong size");
    // kp_its == -1 / kp_emin <= 0: degree and lower bound of the velocity sub-solves (the
    // pressure Laplacian has the same h-dependence), upper bound from K_p itself
    if (d.kp_its < -1 || d.mp_its < 0) fail(KKT_ERR_ARG, "negative Chebyshev degree");
    if ((d.kp_emin > 0 && !(d.kp_emax > d.kp_emin)) ||
        (d.mp_its > 0 && !(d.mp_emax > d.mp_emin && d.mp_emin > 0)))
        fail(KKT_ERR_ARG, "Chebyshev bounds must satisfy 0 < emin < emax");
    sB_ = d.b_scale;
    s2_ = d.post_scale;
    kp_its_ = d.kp_its;
    mp_its_ = d.mp_its;
    B_ = upload(np_, nv_, d.b_indptr, d.b_indices, d.b_values, false);
    Kp_ = upload(np_, np_, d.kp_indptr, d.kp_indices, d.kp_values, true);
    Mp_ = upload(np_, np_, d.mp_indptr, d.mp_indices, d.mp_values, true);
    auto vec = [&](int64_t n) {
        double *p = dev_alloc<double>(n + 32);
        HIPCHK(hipMemsetAsync(p, 0, (n + 32) * sizeof(double), S_.stream));
        owned_.push_back(p);
        return p;
    };
    const int64_t n0 = 2 * (int64_t)n_ * nv_, n1 = 2 * (int64_t)n_ * np_;
    in_ = vec(n0 + n1);
    out_ = vec(n0 + n1);
    h_ = vec(n1);
    m_ = vec(n1);
    g_ = vec(n1);
    for (int k = 0; k < 3; ++k) P_[k] = vec(n1);
    if (cn_ && outer.sharded) {
        halo_a_ = vec(np_);
        halo_b_ = vec(np_);
    }
    // h_k = s2 (sB B u0_k - b1_k), k over the 2n blocks (v blocks then zeta blocks pair with
    // the mu blocks then p blocks: control.py:1030-1041, 4571-4601)
    {
        const Pattern &P = S_.patterns[B_.pat];
        std::vector<RowOp> ops;
        for (int k = 0; k < 2 * n_; ++k) {
            RowOp op = base_op(P);
            op.mode = EPI_LIN;
            op.nterms = 1;
            op.t[0].vals = B_.vals;
            op.t[0].x = vabs(out_ + (int64_t)k * nv_);
            op.y = vabs(h_ + (int64_t)k * np_);
            if (cn_) {   // the time transforms come between the product and the subtraction
                op.ca = sB_;
            } else {
                op.ca = s2_ * sB_;
                op.cz = -s2_;
                op.z = vabs(in_ + n0 + (int64_t)k * np_);
            }
            ops.push_back(op);
        }
        lin_.push_back(upload_launch(P, ops));
    }
    int kp_its = d.kp_its;
    double kp_emin = d.kp_emin, kp_emax = d.kp_emax;
    if (kp_its < 0 || kp_emin <= 0) {
        const SchurPC *sp = dynamic_cast<const SchurPC *>(inner.pc.get());
        if (!sp) fail(KKT_ERR_STATE, "automatic K_p sweeps need the built-in preconditioner on the inner system");
        if (sp->coarse_cycles() > 0) {
            // The velocity sub-solves are two-grid cycles: their few smoothing sweeps and their
            // interval say nothing about K_p.  Estimate K_p itself -- it is singular (constants),
            // so the Lanczos recurrence starts from a zero-mean vector and sees the non-zero
            // spectrum only -- and give the solve 5 sqrt(kappa) sweeps, at most 600: the outer
            // iteration is sensitive to this solve (128^2 x 32, profiles/r03/stokes_quality.txt:
            // 160 / 300 / 600 sweeps -> 195 / 145 / 100 outer iterations).
            const Spectrum ks = jacobi_spectrum(S_, Kp_.pat, Kp_.vals, Kp_.dinv, nullptr, 400, true);
            if (!(ks.emin > 0.0) || !(ks.emax > ks.emin)) fail(KKT_ERR_STATE, "no Chebyshev interval for K_p");
            if (kp_emin <= 0) {
                kp_emin = 0.85 * ks.emin;      // Ritz values approach the ends from inside
                kp_emax = 1.05 * ks.emax;
            }
            if (kp_its < 0)
                kp_its = std::max(4, std::min(600, (int)std::ceil(5.0 * std::sqrt(kp_emax / kp_emin))));
        }
        if (kp_its < 0) kp_its = sp->schur_its();
        if (kp_emin <= 0) {
            kp_emin = sp->typical_emin();
            // K_p is singular (constants): only the upper end of its spectrum is estimated
            kp_emax = 1.05 * jacobi_spectrum(S_, Kp_.pat, Kp_.vals, Kp_.dinv, nullptr, 60).emax;
            if (!(kp_emax > kp_emin)) fail(KKT_ERR_STATE, "no Chebyshev interval for K_p");
        }
    }
    kp_its_ = kp_its;
    emit_cheb(kp_steps_, Kp_, kp_its, kp_emin, kp_emax, h_, m_);
    emit_cheb(mp_steps_, Mp_, d.mp_its, d.mp_emin, d.mp_emax, g_, out_ + n0);
    HIPCHK(hipStreamSynchronize(S_.stream));
}

StokesPC::~StokesPC() {
    for (Chain *c : {&kp_chain_, &mp_chain_}) {
        if (c->exec) (void)hipGraphExecDestroy(c->exec);
        if (c->graph) (void)hipGraphDestroy(c->graph);
    }
    for (auto *v : {&lin_, &kp_steps_, &mp_steps_})
        for (auto &L : *v)
            if (L.d_ops) (void)hipFree(L.d_ops);
    for (void *p : owned_)
        if (p) (void)hipFree(p);
}

void StokesPC::run() {
    hipStream_t st = S_.stream;
    const int64_t n0 = 2 * (int64_t)n_ * nv_;
    const Bases B{{nullptr, nullptr, nullptr, nullptr}};
    // nested solve on the velocity KKT system from a zero guess (control.py:1012-1027); the
    // inner system works on its own stream: order the two streams by synchronising
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipMemsetAsync(out_, 0, n0 * sizeof(double), inner_.stream));
    int its = 0, reason = 0;
    inner_.solve(in_, out_, &its, &reason, nullptr, nullptr, 0, nullptr);
    HIPCHK(hipStreamSynchronize(inner_.stream));
    for (const RowLaunch &L : lin_)
        launch_rowops(st, L.d_ops, L.nops, L.max_slices, L.R, B, 1, L.uniform_w);
    if (cn_) {   // control.py:4407-4428
        const int64_t half = (int64_t)n_ * np_;
        // Time shards: T_2 / T_1 read the neighbour ranks' old blocks (one exchange each way);
        // the inverse transforms are scans through all levels -- a rank waits for the scanned
        // block of the rank before it, scans its levels and passes its own last one on.
        const bool sh = S_.sharded;
        const int up = sh && S_.rank + 1 < S_.world ? S_.rank + 1 : -1;
        const int dn = sh && S_.rank > 0 ? S_.rank - 1 : -1;
        double *last_a = h_ + (int64_t)(n_ - 1) * np_, *first_b = h_ + half;
        if (sh) {
            S_.comm->sendrecv(last_a, np_, up, halo_a_, np_, dn, st);
            S_.comm->sendrecv(first_b, np_, dn, halo_b_, np_, up, st);
        }
        launch_time_transform(st, h_, h_, 2, n_, np_, dn >= 0 ? halo_a_ : nullptr, nullptr);
        launch_time_transform(st, h_ + half, h_ + half, 1, n_, np_, nullptr,
                              up >= 0 ? halo_b_ : nullptr);
        launch_axpby(st, h_, -1.0, in_ + n0, 1.0, 2 * half);
        launch_axpby(st, h_, 0.0, in_ + n0, s2_, 2 * half);
        if (dn >= 0) S_.comm->sendrecv(nullptr, 0, -1, halo_a_, np_, dn, st);
        launch_time_transform(st, h_, h_, 4, n_, np_, dn >= 0 ? halo_a_ : nullptr, nullptr);
        if (up >= 0) S_.comm->sendrecv(last_a, np_, up, nullptr, 0, -1, st);
        if (up >= 0) S_.comm->sendrecv(nullptr, 0, -1, halo_b_, np_, up, st);
        launch_time_transform(st, h_ + half, h_ + half, 3, n_, np_, nullptr,
                              up >= 0 ? halo_b_ : nullptr);
        if (dn >= 0) S_.comm->sendrecv(first_b, np_, dn, nullptr, 0, -1, st);
    }
    run_chain(kp_chain_, kp_steps_);
    // g = C m: the pressure-space commutator block system (control.py:1056-1067, 4625-4665)
    HIPCHK(hipStreamSynchronize(st));
    comm_.apply(m_, g_);
    HIPCHK(hipStreamSynchronize(comm_.stream));
    run_chain(mp_chain_, mp_steps_);
}

// One Chebyshev chain of the pressure space: every block of a step applies the same matrix (four
// blocks per thread share its indices and values where that form applies); the chain works on
// fixed buffers, so it is captured once and replayed as a hipGraph (600 K_p steps of ~5 us of
// work each: the launch gaps were two thirds of their 16 us) unless "no_graph" is set.
void StokesPC::run_chain(Chain &c, const std::vector<RowLaunch> &steps) {
    hipStream_t st = S_.stream;
    const Bases B{{nullptr, nullptr, nullptr, nullptr}};
    auto launches = [&]() {
        for (const RowLaunch &L : steps)
            if (!L.shared_matrix ||
                !launch_rowops_shared(st, L.d_ops, L.nops, L.max_slices, L.R, L.uniform_w))
                launch_rowops(st, L.d_ops, L.nops, L.max_slices, L.R, B, 1, L.uniform_w);
    };
    const char *ng = S_.opt("no_graph");
    if ((ng && ng[0] == '1') || c.failed || steps.size() < 8) {
        launches();
        return;
    }
    if (!c.exec) {
        hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        if (e == hipSuccess) {
            launches();
            e = hipStreamEndCapture(st, &c.graph);
            if (e == hipSuccess) e = hipGraphInstantiate(&c.exec, c.graph, nullptr, nullptr, 0);
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            c.exec = nullptr;
            c.failed = true;          // plain launches of the same kernels from now on
            launches();
            return;
        }
    }
    HIPCHK(hipGraphLaunch(c.exec, st));
}

void StokesPC::check() {
    if (inner_.pc) inner_.pc->check();
}

}  // namespace kkt
