/*
 * CPU restatement in C (+OpenMP) of the hot path.  TEST INFRASTRUCTURE: the checker and
 * the `cpu_baseline` of bench.py; never linked into or called by the product.
 *
 * Same algorithm as oracle/kkt_oracle.py (which cites the reference lines), specialised to
 * what the benchmark runs: BE heat-control KKT operator (preconditioner.py:375-543 with
 * DirichletBCNullspace on every block), the BE block-Schur preconditioner
 * (control/control.py:2191-2438) with Jacobi-Chebyshev inner solves, and left-preconditioned
 * GMRES(m) as PETSc's KSPSolve_GMRES runs it for preconditioner.py:732-759.
 *
 * The SpMV is the loop of PETSc's MatMultAdd_SeqAIJ: one running sum per row, continued
 * across the blocks of a block row in the reference's dict order; here with fma() so that
 * the result is bit-identical to the GPU kernel's fma chain (tests/test_cref.py).
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int32_t nrows;
    const int32_t *indptr;
    const int32_t *indices;
    const double *vals;
} csr_t;

/* y[r] = fma-chain(y[r] ; A[r,:] x)  -- MatMultAdd */
static void spmv_add_p(const csr_t *A, const double *x, double *y, int par) {
#pragma omp parallel for schedule(static) if (par)
    for (int32_t r = 0; r < A->nrows; ++r) {
        double s = y[r];
        for (int32_t k = A->indptr[r]; k < A->indptr[r + 1]; ++k)
            s = fma(A->vals[k], x[A->indices[k]], s);
        y[r] = s;
    }
}

static void spmv_add(const csr_t *A, const double *x, double *y) { spmv_add_p(A, x, y, 1); }

void ref_spmv_add(const csr_t *A, const double *x, double *y) { spmv_add(A, x, y); }

typedef struct {
    int32_t m, nx;
    /* per quadrant q (00, 01, 10, 11): nb[q] blocks in the reference's dict order */
    int32_t nb[4];
    const int32_t *bi[4];
    const int32_t *bj[4];
    const csr_t *blk[4];
    const uint8_t *mask; /* nx: 1 on Dirichlet dofs (same set for every block) */
} ref_sys_t;

/* y = P A P x + (I - P) x, BE (no time transform) */
void ref_kkt_apply(const ref_sys_t *S, const double *x, double *y) {
    const int64_t nx = S->nx, N = 2 * (int64_t)S->m * nx;
    double *xc = (double *)malloc(N * sizeof(double));
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < N; ++p) {
        xc[p] = S->mask[p % nx] ? 0.0 : x[p];
        y[p] = 0.0;
    }
    /* block rows are independent; inside a row the blocks accumulate in dict order */
#pragma omp parallel for schedule(dynamic)
    for (int row = 0; row < 2 * S->m; ++row) {
        for (int q = (row < S->m ? 0 : 2); q < (row < S->m ? 2 : 4); ++q) {
            const int row_off = (q >= 2) ? S->m : 0, col_off = (q & 1) ? S->m : 0;
            for (int32_t b = 0; b < S->nb[q]; ++b)
                if (row_off + S->bi[q][b] == row)
                    spmv_add_p(&S->blk[q][b], xc + (int64_t)(col_off + S->bj[q][b]) * nx,
                               y + (int64_t)row * nx, 0);
        }
    }
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < N; ++p)
        if (S->mask[p % nx]) y[p] = x[p];
    free(xc);
}

/* KSPSolve_Chebyshev (first kind) + PCJACOBI, zero guess, `its` steps; out may alias nothing */
static void cheb(const csr_t *A, const double *dinv, const double *b, double emin, double emax,
                 int its, double *out, double *w0, double *w1, double *w2, int par) {
    const int32_t n = A->nrows;
    if (its == 0) {
#pragma omp parallel for schedule(static) if (par)
        for (int32_t r = 0; r < n; ++r) out[r] = dinv[r] * b[r];
        return;
    }
    const double scale = 2.0 / (emax + emin), alpha = 1.0 - scale * emin;
    const double mu = 1.0 / alpha, omegaprod = 2.0 / alpha;
    double c_km1 = 1.0, c_k = mu;
    double *pkm1 = w0, *pk = w1, *pkp1 = w2;
#pragma omp parallel for schedule(static) if (par)
    for (int32_t r = 0; r < n; ++r) {
        pkm1[r] = 0.0;
        pk[r] = scale * (dinv[r] * b[r]) + 0.0;
    }
    for (int i = 1; i < its; ++i) {
        const double c_kp1 = 2.0 * mu * c_k - c_km1;
        const double omega = omegaprod * c_k / c_kp1;
#pragma omp parallel for schedule(static) if (par)
        for (int32_t r = 0; r < n; ++r) {
            double s = 0.0;
            for (int32_t k = A->indptr[r]; k < A->indptr[r + 1]; ++k)
                s = fma(A->vals[k], pk[A->indices[k]], s);
            const double z = dinv[r] * (b[r] - s);
            pkp1[r] = (1.0 - omega) * pkm1[r] + omega * pk[r] + (scale * omega) * z;
        }
        double *t = pkm1;
        pkm1 = pk;
        pk = pkp1;
        pkp1 = t;
        c_km1 = c_k;
        c_k = c_kp1;
    }
    memcpy(out, pk, (size_t)n * sizeof(double));
}

/* Two-grid form of a sub-solve (oracle.kkt_oracle.coarse_chebyshev): cycles x [x += P Einv P^T
 * (b - A x); `its` Jacobi-Chebyshev sweeps from x (KSPSolve_Chebyshev with a non-zero guess)].
 * PT = P^T as CSR (nc rows); Einv = (P^T A P)^-1, nc x nc row-major; x (n), r (n), rc, ec (nc). */
static void coarse_cheb(const csr_t *A, const double *dinv, const double *b, double emin,
                        double emax, int its, int cycles, const csr_t *Pm, const csr_t *PT,
                        const double *Einv, int nc, double *out, double *w0, double *w1, double *w2,
                        double *r, double *rc, double *ec) {
    const int32_t n = A->nrows;
    const double scale = 2.0 / (emax + emin), alpha = 1.0 - scale * emin;
    const double mu = 1.0 / alpha, omegaprod = 2.0 / alpha;
    double *x = w0;      /* the iterate between cycles */
    for (int c = 0; c < cycles; ++c) {
        /* residual of the current iterate (zero guess in the first cycle) */
#pragma omp parallel for schedule(static)
        for (int32_t i = 0; i < n; ++i) {
            double s = 0.0;
            if (c > 0)
                for (int32_t k = A->indptr[i]; k < A->indptr[i + 1]; ++k)
                    s = fma(A->vals[k], x[A->indices[k]], s);
            r[i] = b[i] - s;
        }
#pragma omp parallel for schedule(static)
        for (int32_t j = 0; j < nc; ++j) {
            double s = 0.0;
            for (int32_t k = PT->indptr[j]; k < PT->indptr[j + 1]; ++k)
                s = fma(PT->vals[k], r[PT->indices[k]], s);
            rc[j] = s;
        }
#pragma omp parallel for schedule(static)
        for (int32_t j = 0; j < nc; ++j) {
            double s = 0.0;
            for (int32_t k = 0; k < nc; ++k) s = fma(Einv[(size_t)j * nc + k], rc[k], s);
            ec[j] = s;
        }
        /* p0 = x + P ec, into r (the residual is used up) */
#pragma omp parallel for schedule(static)
        for (int32_t i = 0; i < n; ++i) {
            double s = 0.0;
            for (int32_t k = Pm->indptr[i]; k < Pm->indptr[i + 1]; ++k)
                s = fma(Pm->vals[k], ec[Pm->indices[k]], s);
            r[i] = c > 0 ? x[i] + s : s;
        }
        /* `its` sweeps from p0: p1 = p0 + scale dinv (b - A p0), then the recurrence; the three
         * vectors rotate through r, w1, w2 */
        double *pkm1 = r, *pk = w1, *pkp1 = w2;
        double c_km1 = 1.0, c_k = mu;
#pragma omp parallel for schedule(static)
        for (int32_t i = 0; i < n; ++i) {
            double s = 0.0;
            for (int32_t k = A->indptr[i]; k < A->indptr[i + 1]; ++k)
                s = fma(A->vals[k], pkm1[A->indices[k]], s);
            pk[i] = pkm1[i] + scale * (dinv[i] * (b[i] - s));
        }
        for (int i2 = 1; i2 < its; ++i2) {
            const double c_kp1 = 2.0 * mu * c_k - c_km1;
            const double omega = omegaprod * c_k / c_kp1;
#pragma omp parallel for schedule(static)
            for (int32_t i = 0; i < n; ++i) {
                double s = 0.0;
                for (int32_t k = A->indptr[i]; k < A->indptr[i + 1]; ++k)
                    s = fma(A->vals[k], pk[A->indices[k]], s);
                const double z = dinv[i] * (b[i] - s);
                pkp1[i] = (1.0 - omega) * pkm1[i] + omega * pk[i] + (scale * omega) * z;
            }
            double *t = pkm1;
            pkm1 = pk;
            pk = pkp1;
            pkp1 = t;
            c_km1 = c_k;
            c_k = c_kp1;
        }
        memcpy(x, pk, (size_t)n * sizeof(double));
    }
    memcpy(out, x, (size_t)n * sizeof(double));
}

typedef struct {
    int32_t n_t, nx;
    double tau, beta, epsilon;
    const csr_t *M;    /* plain mass matrix */
    const csr_t *Mt;   /* bc-assembled mass matrix */
    const double *mdinv;
    const csr_t *D10;  /* n_t: block_10(i,i) */
    const csr_t *S10;  /* n_t: block_10(i,i-1) (entry 0 unused) */
    const csr_t *S01;  /* n_t: block_01(i,i+1) (entry n_t-1 unused) */
    const csr_t *F;    /* n_t: bc-assembled block_10(i,i) + c_i M */
    const double *const *Fdinv;
    const csr_t *G;    /* n_t: bc-assembled block_01(i,i) + c_i M */
    const double *const *Gdinv;
    const uint8_t *mask;
    int32_t mass_its, schur_its;
    double mass_emin, mass_emax, schur_emin, schur_emax;
    /* two-grid form of the Schur sub-solves (coarse_cycles == 0: plain Chebyshev) */
    int32_t coarse_cycles, nc;
    const csr_t *P, *PT;
    const double *const *FEinv;   /* n_t: (P^T F_i P)^-1 */
    const double *const *GEinv;
} ref_pc_t;

/* u = P pc_linear(P b) + (I - P) b, BE (control.py:2191-2438 inside preconditioner.py:562-656) */
void ref_pc_apply_BE(const ref_pc_t *P, const double *b, double *u) {
    const int32_t n = P->n_t, nx = P->nx;
    const int64_t N = 2 * (int64_t)n * nx;
    const double tau = P->tau, eps = P->epsilon;
    double *bc = (double *)malloc(N * sizeof(double));
    double *B = (double *)malloc((size_t)n * nx * sizeof(double));
    double *w = (double *)malloc(((size_t)5 * nx + 2 * (size_t)(P->coarse_cycles > 0 ? P->nc : 0)) *
                                 sizeof(double));
    double *t = w + 3 * (int64_t)nx;
    double *cr = w + 4 * (int64_t)nx, *crc = w + 5 * (int64_t)nx, *cec = crc + P->nc;
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < N; ++p) bc[p] = P->mask[p % nx] ? 0.0 : b[p];
    const double *b0 = bc, *b1 = bc + (int64_t)n * nx;
    double *u0 = u, *u1 = u + (int64_t)n * nx;
#define BLK(a, i) ((a) + (int64_t)(i) * nx)
    /* mass solves: independent per time level -> one level per thread */
#pragma omp parallel
    {
        double *ws = (double *)malloc((size_t)3 * nx * sizeof(double));
#pragma omp for schedule(dynamic)
        for (int i = 0; i < n; ++i) {
            cheb(P->Mt, P->mdinv, BLK(b0, i), P->mass_emin, P->mass_emax, P->mass_its, BLK(u0, i),
                 ws, ws + nx, ws + 2 * (int64_t)nx, 0);
            const double s1 = 1.0 / tau, s2 = (i == n - 1) ? 1.0 / eps : 1.0;
            double *ui = BLK(u0, i);
            for (int32_t r = 0; r < nx; ++r) ui[r] = s2 * (s1 * ui[r]);
        }
        free(ws);
    }
#pragma omp parallel
    {
        double *ts = (double *)malloc((size_t)nx * sizeof(double));
#pragma omp for schedule(dynamic)
        for (int i = 0; i < n; ++i) {
            double *Bi = BLK(B, i);
            memset(Bi, 0, (size_t)nx * sizeof(double));
            spmv_add_p(&P->D10[i], BLK(u0, i), Bi, 0);
            if (i >= 1) {
                memset(ts, 0, (size_t)nx * sizeof(double));
                spmv_add_p(&P->S10[i], BLK(u0, i - 1), ts, 0);
                for (int32_t r = 0; r < nx; ++r) Bi[r] += ts[r];
            }
            const double *b1i = BLK(b1, i);
            for (int32_t r = 0; r < nx; ++r) Bi[r] = P->mask[r] ? 0.0 : Bi[r] - b1i[r];
        }
        free(ts);
    }
    for (int i = 0; i < n; ++i) { /* forward sweep */
        double *Bi = BLK(B, i);
        if (i >= 1) {
            memset(t, 0, (size_t)nx * sizeof(double));
            spmv_add(&P->S10[i], BLK(u1, i - 1), t);
#pragma omp parallel for schedule(static)
            for (int32_t r = 0; r < nx; ++r) Bi[r] = P->mask[r] ? 0.0 : Bi[r] - t[r];
        }
        if (P->coarse_cycles > 0)
            coarse_cheb(&P->F[i], P->Fdinv[i], Bi, P->schur_emin, P->schur_emax, P->schur_its,
                        P->coarse_cycles, P->P, P->PT, P->FEinv[i], P->nc, BLK(u1, i), w, w + nx,
                        w + 2 * (int64_t)nx, cr, crc, cec);
        else
        cheb(&P->F[i], P->Fdinv[i], Bi, P->schur_emin, P->schur_emax, P->schur_its, BLK(u1, i), w,
             w + nx, w + 2 * (int64_t)nx, 1);
    }
#pragma omp parallel for schedule(dynamic)
    for (int i = 0; i < n; ++i) {
        double *Bi = BLK(B, i);
        memset(Bi, 0, (size_t)nx * sizeof(double));
        spmv_add_p(P->M, BLK(u1, i), Bi, 0);
        const double c = (i == n - 1) ? eps * tau : tau;
        for (int32_t r = 0; r < nx; ++r) Bi[r] = P->mask[r] ? 0.0 : Bi[r] * c;
    }
    for (int i = n - 1; i >= 0; --i) { /* backward sweep */
        double *Bi = BLK(B, i);
        if (i <= n - 2) {
            memset(t, 0, (size_t)nx * sizeof(double));
            spmv_add(&P->S01[i], BLK(u1, i + 1), t);
#pragma omp parallel for schedule(static)
            for (int32_t r = 0; r < nx; ++r) Bi[r] = P->mask[r] ? 0.0 : Bi[r] - t[r];
        }
        if (P->coarse_cycles > 0)
            coarse_cheb(&P->G[i], P->Gdinv[i], Bi, P->schur_emin, P->schur_emax, P->schur_its,
                        P->coarse_cycles, P->P, P->PT, P->GEinv[i], P->nc, BLK(u1, i), w, w + nx,
                        w + 2 * (int64_t)nx, cr, crc, cec);
        else
        cheb(&P->G[i], P->Gdinv[i], Bi, P->schur_emin, P->schur_emax, P->schur_its, BLK(u1, i), w,
             w + nx, w + 2 * (int64_t)nx, 1);
    }
#undef BLK
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < N; ++p)
        if (P->mask[p % nx]) u[p] = b[p];
    free(bc);
    free(B);
    free(w);
}

static double dotp(const double *a, const double *b, int64_t n) {
    double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
    for (int64_t p = 0; p < n; ++p) s += a[p] * b[p];
    return s;
}

/* Left-preconditioned GMRES(m), classical Gram-Schmidt, KSPConvergedDefault with the norm
 * of the preconditioned right-hand side (nonzero initial guess).  Returns iterations. */
int ref_gmres_BE(const ref_sys_t *S, const ref_pc_t *P, const double *b_in, double *x, int restart,
                 double rtol, double atol, double divtol, int max_it, double *hist, int hist_cap,
                 int *hist_len, int *reason_out) {
    const int64_t N = 2 * (int64_t)S->m * S->nx;
    const int m = restart;
    double *V = (double *)malloc((size_t)(m + 1) * N * sizeof(double));
    double *t1 = (double *)malloc(N * sizeof(double)), *t2 = (double *)malloc(N * sizeof(double));
    double *b = (double *)malloc(N * sizeof(double));
    double *H = (double *)calloc((size_t)(m + 1) * m, sizeof(double));
    double *cc = (double *)calloc(m, sizeof(double)), *ss = (double *)calloc(m, sizeof(double));
    double *grs = (double *)calloc(m + 1, sizeof(double)), *yv = (double *)calloc(m, sizeof(double));
#define Hm(r, c) H[(size_t)(c) * (m + 1) + (r)]
    for (int64_t p = 0; p < N; ++p) {
        const int msk = S->mask[p % S->nx];
        b[p] = msk ? 0.0 : b_in[p];
        if (msk) x[p] = 0.0;
    }
    ref_pc_apply_BE(P, b, t2);
    double rnorm0 = sqrt(dotp(t2, t2, N));
    double ttol = fmax(rtol * rnorm0, atol);
    int its = 0, reason = 0, nh = 0;
    double rn = 0.0;
#define LOG(v)                        \
    do {                              \
        if (hist && nh < hist_cap) hist[nh] = (v); \
        ++nh;                         \
    } while (0)
#define CONV(v) (!isfinite(v) ? -9 : ((v) <= ttol ? ((v) < atol ? 3 : 2) : ((v) >= divtol * rnorm0 ? -4 : 0)))
    for (;;) {
        ref_kkt_apply(S, x, t1);
        for (int64_t p = 0; p < N; ++p) t2[p] = b[p] - t1[p];
        ref_pc_apply_BE(P, t2, V);
        rn = sqrt(dotp(V, V, N));
        if (rnorm0 == 0.0) { /* KSPConvergedDefault: zero right-hand side, non-zero guess */
            rnorm0 = rn;
            ttol = fmax(rtol * rn, atol);
        }
        LOG(rn);
        if (rn == 0.0) {
            reason = 3;
            break;
        }
        for (int64_t p = 0; p < N; ++p) V[p] *= 1.0 / rn;
        memset(grs, 0, (m + 1) * sizeof(double));
        grs[0] = rn;
        reason = CONV(rn);
        int it = 0;
        memset(H, 0, (size_t)(m + 1) * m * sizeof(double));
        while (!reason && it < m && its < max_it) {
            if (it) LOG(rn);
            double *w = V + (size_t)(it + 1) * N;
            ref_kkt_apply(S, V + (size_t)it * N, t1);
            ref_pc_apply_BE(P, t1, w);
            for (int k = 0; k <= it; ++k) Hm(k, it) = dotp(V + (size_t)k * N, w, N);
            for (int k = 0; k <= it; ++k) {
                const double h = Hm(k, it);
                const double *vk = V + (size_t)k * N;
#pragma omp parallel for schedule(static)
                for (int64_t p = 0; p < N; ++p) w[p] -= h * vk[p];
            }
            const double tt = sqrt(dotp(w, w, N));
            for (int64_t p = 0; p < N; ++p) w[p] *= 1.0 / tt;
            Hm(it + 1, it) = tt;
            for (int j = 0; j < it; ++j) {
                const double t = Hm(j, it);
                Hm(j, it) = cc[j] * t + ss[j] * Hm(j + 1, it);
                Hm(j + 1, it) = cc[j] * Hm(j + 1, it) - ss[j] * t;
            }
            const double t = sqrt(Hm(it, it) * Hm(it, it) + Hm(it + 1, it) * Hm(it + 1, it));
            cc[it] = Hm(it, it) / t;
            ss[it] = Hm(it + 1, it) / t;
            grs[it + 1] = -(ss[it] * grs[it]);
            grs[it] = cc[it] * grs[it];
            Hm(it, it) = cc[it] * Hm(it, it) + ss[it] * Hm(it + 1, it);
            rn = fabs(grs[it + 1]);
            ++it;
            ++its;
            reason = CONV(rn);
        }
        if (it && (reason || its >= max_it)) LOG(rn);
        for (int k = it - 1; k >= 0; --k) {
            double s = grs[k];
            for (int j = k + 1; j < it; ++j) s -= Hm(k, j) * yv[j];
            yv[k] = s / Hm(k, k);
        }
        for (int k = 0; k < it; ++k) {
            const double *vk = V + (size_t)k * N;
#pragma omp parallel for schedule(static)
            for (int64_t p = 0; p < N; ++p) x[p] += yv[k] * vk[p];
        }
        if (reason) break;
        if (its >= max_it) {
            reason = -3;
            break;
        }
    }
    for (int64_t p = 0; p < N; ++p)
        if (S->mask[p % S->nx]) x[p] = 0.0;
    if (hist_len) *hist_len = nh;
    if (reason_out) *reason_out = reason;
    free(V); free(t1); free(t2); free(b); free(H); free(cc); free(ss); free(grs); free(yv);
    return its;
}
