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
void StokesPC::emit_cheb(std::vector<ChainStep> &dst, const DevMat &A, int its, double emin,
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
    dst.push_back(ChainStep{upload_launch(P, ops)});
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
        dst.push_back(ChainStep{upload_launch(P, ops)});
        c_km1 = c_k;
        c_k = c_kp1;
    }
}

// ---- two-grid K_p solve (kkt_pc_stokes_desc.kp_coarse_*)
//
// K_p is a Neumann Laplacian: kappa of its Jacobi-scaled non-zero spectrum grows like h^-2 (1.3e4 on
// the 129^2 pressure grid of BASELINE configs[2]) and the outer iteration is sensitive to this
// solve -- 600 plain sweeps, each a 16 us launch, were the best plain setting.  Here: cycles x
// [x += P E^-1 P^T (b - K_p x) for every block; `its` sweeps on [emin, emax] from x].  The Galerkin
// matrix E = P^T K_p P inherits the constants as its kernel (the columns of P sum to one on every
// row), so E + (trace E / n_c^2) 1 1^T is inverted instead: on right-hand sides orthogonal to the
// constants -- P^T r of a residual in the range of K_p -- it acts as the pseudo-inverse.
void StokesPC::build_kp_coarse(const kkt_pc_stokes_desc &d) {
    const int nc = (int)d.kp_n_coarse;
    if (nc < 1 || !d.kp_p_indptr || !d.kp_p_indices || !d.kp_p_values)
        fail(KKT_ERR_ARG, "kkt_set_pc_stokes: coarse space of the K_p solve missing");
    hipStream_t st = S_.stream;
    std::vector<int32_t> ip(d.kp_p_indptr, d.kp_p_indptr + np_ + 1);
    if (ip[0] != 0) fail(KKT_ERR_ARG, "kp_p_indptr must start at 0");
    const int64_t nnz = ip[np_];
    std::vector<int32_t> ix(d.kp_p_indices, d.kp_p_indices + nnz);
    std::vector<double> pv(d.kp_p_values, d.kp_p_values + nnz);
    for (int32_t c : ix)
        if (c < 0 || c >= nc) fail(KKT_ERR_ARG, "kp_p_indices out of range");
    // P^T by a counting sort (entries of a column in ascending row order: the restriction's order)
    std::vector<int32_t> tip(nc + 1, 0), tix(nnz);
    std::vector<double> tv(nnz);
    for (int32_t c : ix) tip[c + 1]++;
    for (int j = 0; j < nc; ++j) tip[j + 1] += tip[j];
    {
        std::vector<int32_t> cur(tip.begin(), tip.end() - 1);
        for (int64_t r = 0; r < np_; ++r)
            for (int32_t q = ip[r]; q < ip[r + 1]; ++q) {
                const int32_t at = cur[ix[q]]++;
                tix[at] = (int32_t)r;
                tv[at] = pv[q];
            }
    }
    auto up = [&](const auto &v) {
        auto *p = dev_upload(v.data(), std::max<size_t>(1, v.size()));
        owned_.push_back((void *)p);
        return p;
    };
    CoarseDev &c = kp_coarse_;
    c.nc = nc;
    c.p_ip = up(ip);
    c.p_ix = up(ix);
    c.p_v = up(pv);
    c.pt_ip = up(tip);
    c.pt_ix = up(tix);
    c.pt_v = up(tv);
    c.rc = dev_alloc<double>((size_t)nc * 2 * n_);      // one coarse residual per pressure block
    c.ec = dev_alloc<double>((size_t)nc * 2 * n_);
    owned_.push_back(c.rc);
    owned_.push_back(c.ec);
    // E column by column with the kernels the sweeps use (fixed summation orders)
    const Pattern &P = S_.patterns[Kp_.pat];
    double *x = dev_alloc<double>(np_ + 32), *y = dev_alloc<double>(np_ + 32);
    double *E = dev_alloc<double>((size_t)nc * nc);
    RowOp op = base_op(P);
    op.mode = EPI_LIN;
    op.nterms = 1;
    op.t[0].vals = Kp_.vals;
    op.t[0].x = vabs(x);
    op.y = vabs(y);
    op.ca = 1.0;
    RowOp *d_op = dev_upload(&op, 1);
    const Bases B{{nullptr, nullptr, nullptr, nullptr}};
    for (int k = 0; k < nc; ++k) {
        launch_coarse_column(st, c, k, x, np_);
        launch_rowops(st, d_op, 1, P.nslices, P.R, B, 1, P.uniform_w);
        launch_coarse_restrict(st, c, y, E + k, nc);        // column k of the row-major E
    }
    // deflation of the constants: every entry + trace(E) / n_c^2 (diagonal summed on the host in
    // index order)
    {
        std::vector<double> diag(nc);
        HIPCHK(hipMemcpy2DAsync(diag.data(), sizeof(double), E, ((size_t)nc + 1) * sizeof(double),
                                sizeof(double), nc, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        double tr = 0.0;
        for (double v : diag) tr += v;
        launch_add_constant(st, E, tr / ((double)nc * (double)nc), (int64_t)nc * nc);
    }
    kp_einv_ = dev_alloc<double>((size_t)nc * nc);
    owned_.push_back(kp_einv_);
    int *d_piv = dev_alloc<int>(1);
    double *d_colbuf = dev_alloc<double>(nc + 1);
    unsigned *d_flag = dev_alloc<unsigned>(1);
    HIPCHK(hipMemsetAsync(d_flag, 0, sizeof(unsigned), st));
    launch_dense_inverse(st, E, kp_einv_, nc, d_piv, d_colbuf, d_flag);
    unsigned singular = 0;
    HIPCHK(hipMemcpyAsync(&singular, d_flag, sizeof singular, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (void *q : {(void *)x, (void *)y, (void *)E, (void *)d_op, (void *)d_piv, (void *)d_colbuf,
                    (void *)d_flag})
        (void)hipFree(q);
    if (singular)
        fail(KKT_ERR_STATE, "coarse matrix of the K_p solve is singular beyond the constants "
                            "(dependent coarse functions?)");
}

void StokesPC::emit_kp_two_grid(int its, double emin, double emax, const double *b, double *out) {
    const Pattern &P = S_.patterns[Kp_.pat];
    const int nb = 2 * n_;
    auto blk = [&](const double *base, int k) { return base + (int64_t)k * np_; };
    const double scale = 2.0 / (emax + emin);
    // coefficients of the sweeps 2 .. its (the recurrence of emit_cheb)
    std::vector<double> c1(its + 1, 0.0), c2(its + 1, 1.0), c3(its + 1, scale);
    {
        const double alpha = 1.0 - scale * emin, mu = 1.0 / alpha, omegaprod = 2.0 / alpha;
        double c_km1 = 1.0, c_k = mu;
        for (int step = 2; step <= its; ++step) {
            const double c_kp1 = 2.0 * mu * c_k - c_km1;
            const double omega = omegaprod * c_k / c_kp1;
            c1[step] = 1.0 - omega;
            c2[step] = omega;
            c3[step] = scale * omega;
            c_km1 = c_k;
            c_k = c_kp1;
        }
    }
    const double *xcur = nullptr;
    for (int cyc = 0; cyc < kp_cycles_; ++cyc) {
        const bool last_cycle = cyc + 1 == kp_cycles_;
        const double *r = b;
        if (cyc > 0) {          // r = b - K_p x
            std::vector<RowOp> ops(nb);
            for (int k = 0; k < nb; ++k) {
                RowOp op = base_op(P);
                op.mode = EPI_LIN;
                op.nterms = 1;
                op.t[0].vals = Kp_.vals;
                op.t[0].x = vabs(blk(xcur, k));
                op.y = vabs(blk(kp_r_, k));
                op.ca = -1.0;
                op.cz = 1.0;
                op.z = vabs(blk(b, k));
                ops[k] = op;
            }
            kp_steps_.push_back(ChainStep{upload_launch(P, ops)});
            r = kp_r_;
        }
        ChainStep cs;
        cs.coarse = true;
        cs.r = r;
        cs.x_in = xcur;
        cs.x_out = kp_x0_;
        kp_steps_.push_back(cs);
        auto target = [&](int step) -> double * {
            return (last_cycle && step == its) ? out : P_[(step - 1) % 3];
        };
        for (int step = 1; step <= its; ++step) {
            const double *pk = step == 1 ? kp_x0_ : target(step - 1);
            const double *pkm1 = step == 1 ? nullptr : (step == 2 ? kp_x0_ : target(step - 2));
            std::vector<RowOp> ops(nb);
            for (int k = 0; k < nb; ++k) {
                RowOp op = base_op(P);
                op.mode = EPI_CHEB;
                op.nterms = 1;
                op.t[0].vals = Kp_.vals;
                op.t[0].x = vabs(blk(pk, k));
                op.y = vabs(blk(target(step), k));
                op.b = vabs(blk(b, k));
                op.pk = vabs(blk(pk, k));
                op.pkm1 = vabs(pkm1 ? blk(pkm1, k) : nullptr);
                op.dinv = Kp_.dinv;
                op.c1 = step == 1 ? 0.0 : c1[step];
                op.c2 = step == 1 ? 1.0 : c2[step];
                op.c3 = step == 1 ? scale : c3[step];
                ops[k] = op;
            }
            kp_steps_.push_back(ChainStep{upload_launch(P, ops)});
        }
        xcur = target(its);
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
    kp_cycles_ = d.kp_coarse_cycles;
    if (kp_cycles_ < 0) fail(KKT_ERR_ARG, "negative number of coarse cycles for K_p");
    if (kp_cycles_ > 0) {
        if (kp_its < 1 || !(kp_emin > 0.0) || !(kp_emax > kp_emin))
            fail(KKT_ERR_ARG, "two-grid K_p solve needs kp_its >= 1 and 0 < kp_emin < kp_emax");
        build_kp_coarse(d);
        kp_r_ = vec(n1);
        kp_x0_ = vec(n1);
        emit_kp_two_grid(kp_its, kp_emin, kp_emax, h_, m_);
    } else {
        emit_cheb(kp_steps_, Kp_, kp_its, kp_emin, kp_emax, h_, m_);
    }
    emit_cheb(mp_steps_, Mp_, d.mp_its, d.mp_emin, d.mp_emax, g_, out_ + n0);
    HIPCHK(hipStreamSynchronize(S_.stream));
}

StokesPC::~StokesPC() {
    for (Chain *c : {&kp_chain_, &mp_chain_}) {
        if (c->exec) (void)hipGraphExecDestroy(c->exec);
        if (c->graph) (void)hipGraphDestroy(c->graph);
    }
    for (auto &L : lin_)
        if (L.d_ops) (void)hipFree(L.d_ops);
    for (auto *v : {&kp_steps_, &mp_steps_})
        for (auto &c : *v)
            if (c.L.d_ops) (void)hipFree(c.L.d_ops);
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
void StokesPC::run_chain(Chain &c, const std::vector<ChainStep> &steps) {
    hipStream_t st = S_.stream;
    const Bases B{{nullptr, nullptr, nullptr, nullptr}};
    auto launches = [&]() {
        for (const ChainStep &cs : steps) {
            if (cs.coarse) {      // all pressure blocks: one launch per stage
                launch_coarse_correction_batched(st, kp_coarse_, kp_einv_, cs.r, cs.x_in, cs.x_out,
                                                 np_, 2 * n_, np_);
                continue;
            }
            const RowLaunch &L = cs.L;
            if (!L.shared_matrix ||
                !launch_rowops_shared(st, L.d_ops, L.nops, L.max_slices, L.R, L.uniform_w))
                launch_rowops(st, L.d_ops, L.nops, L.max_slices, L.R, B, 1, L.uniform_w);
        }
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
