// Spectrum of a Jacobi-scaled sub-solve matrix on the device: the interval for the
// Jacobi-Chebyshev sweeps that stand in for the reference's BoomerAMG cycles
// (control/control.py:2242-2431; BASELINE.json north_star).  The reference gives no interval for
// these solves (it uses AMG); a fixed hand-set one fits only the mesh it was found for.
//
// Lanczos through the conjugate-gradient recurrences with the Jacobi preconditioner
// (the coefficients alpha_k, beta_k of PCG define the Lanczos tridiagonal matrix of
// D^-1/2 A D^-1/2): k SpMVs with the existing block-row kernel, two reductions and three
// vector updates per step, the k x k tridiagonal eigenproblem on the host.  Extreme Ritz
// values converge first and from inside the spectrum, so the interval is widened a little.
#include <algorithm>
#include <cmath>
#include <vector>

#include "pc.hpp"

namespace kkt {

namespace {

// eigenvalues of the symmetric tridiagonal matrix (d, e) by implicit QL (EISPACK tql1)
std::vector<double> tridiag_eigenvalues(std::vector<double> d, std::vector<double> e) {
    const int n = (int)d.size();
    e.resize(n, 0.0);
    for (int l = 0; l < n; ++l) {
        int iter = 0, m;
        do {
            for (m = l; m < n - 1; ++m) {
                const double dd = std::fabs(d[m]) + std::fabs(d[m + 1]);
                if (std::fabs(e[m]) <= 1e-16 * dd) break;
            }
            if (m != l) {
                if (++iter > 200) break;
                double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
                double r = std::hypot(g, 1.0);
                g = d[m] - d[l] + e[l] / (g + (g >= 0 ? std::fabs(r) : -std::fabs(r)));
                double s = 1.0, c = 1.0, p = 0.0;
                int i;
                for (i = m - 1; i >= l; --i) {
                    double f = s * e[i];
                    const double b = c * e[i];
                    r = std::hypot(f, g);
                    e[i + 1] = r;
                    if (r == 0.0) {
                        d[i + 1] -= p;
                        e[m] = 0.0;
                        break;
                    }
                    s = f / r;
                    c = g / r;
                    g = d[i + 1] - p;
                    r = (d[i] - g) * s + 2.0 * c * b;
                    p = s * r;
                    d[i + 1] = g + p;
                    g = c * r - b;
                }
                if (r == 0.0 && i >= l) continue;
                d[l] -= p;
                e[l] = g;
                e[m] = 0.0;
            }
        } while (m != l);
    }
    std::sort(d.begin(), d.end());
    return d;
}

}  // namespace

// `zero_mean`: the matrix is singular with the constants in its kernel (a Neumann Laplacian):
// the start vector gets zero mean, so the recurrence stays in the range and the smallest Ritz
// value approaches the smallest NON-ZERO eigenvalue.
Spectrum jacobi_spectrum(System &S, int pattern, const double *vals, const double *dinv,
                         const uint8_t *rowmask, int max_steps, bool zero_mean) {
    const Pattern &P = S.patterns[pattern];
    const int64_t n = P.nrows;
    hipStream_t st = S.stream;
    auto vec = [&]() {
        double *p = dev_alloc<double>(n + 32);
        HIPCHK(hipMemsetAsync(p, 0, (n + 32) * sizeof(double), st));
        return p;
    };
    double *r = vec(), *z = vec(), *p = vec(), *w = vec();
    double *scratch = dev_alloc<double>((size_t)REDUCE_BLOCKS * MDOT_MAX);
    double *d_out = dev_alloc<double>(4);
    // deterministic start vector with all frequencies, zero on the boundary rows
    {
        std::vector<double> h(n);
        uint64_t sd = 0x9e3779b97f4a7c15ull;
        for (int64_t i = 0; i < n; ++i) {
            sd = sd * 6364136223846793005ull + 1442695040888963407ull;
            h[i] = (double)(sd >> 11) / (double)(1ull << 53) - 0.5;
        }
        if (rowmask) {
            std::vector<uint8_t> m(n);
            HIPCHK(hipMemcpy(m.data(), rowmask, n, hipMemcpyDeviceToHost));
            for (int64_t i = 0; i < n; ++i)
                if (m[i]) h[i] = 0.0;
        }
        if (zero_mean) {
            long double sum = 0.0L;
            for (int64_t i = 0; i < n; ++i) sum += h[i];
            const double mean = (double)(sum / (long double)n);
            for (int64_t i = 0; i < n; ++i) h[i] -= mean;
        }
        HIPCHK(hipMemcpy(r, h.data(), n * sizeof(double), hipMemcpyHostToDevice));
    }
    auto vabs = [](const double *q) { return q ? VRef{(int64_t)(uintptr_t)q, 0, 0} : VRef{0, -1, 0}; };
    // w = A p (boundary rows 0) and z = D^-1 r as single RowOps through the kernel arguments
    RowOp opA{}, opZ{};
    opA.col = P.d_col;
    opA.perm = P.d_perm;
    opA.slice_off = P.d_slice_off;
    opA.uniform_w = P.uniform_w;
    opA.nrows = (int32_t)n;
    opA.nslices = P.nslices;
    opA.nterms = 1;
    opA.mode = EPI_LIN;
    opA.t[0].vals = vals;
    opA.t[0].x = vabs(p);
    opA.y = vabs(w);
    opA.y2 = opA.yin = opA.z = opA.mx = opA.b = opA.pk = opA.pkm1 = vabs(nullptr);
    opA.ca = 1.0;
    opA.rowmask = rowmask;
    opZ = opA;
    opZ.nterms = 0;
    opZ.uniform_w = 0;
    std::vector<int32_t> zoff(P.nslices + 1, 0);
    int32_t *d_zoff = dev_upload(zoff.data(), zoff.size());
    opZ.slice_off = d_zoff;
    opZ.mode = EPI_CHEB;
    opZ.y = vabs(z);
    opZ.b = vabs(r);
    opZ.dinv = dinv;
    opZ.c1 = opZ.c2 = 0.0;
    opZ.c3 = 1.0;
    opZ.post1 = opZ.post2 = 1.0;
    RowOp *d_ops = nullptr;
    {
        RowOp both[2] = {opA, opZ};
        d_ops = dev_upload(both, 2);
    }
    const Bases B{{nullptr, nullptr, nullptr, nullptr}};
    auto run = [&](int which) {
        launch_rowops(st, d_ops + which, 1, P.nslices, P.R, B, 1, which == 0 ? P.uniform_w : 0);
    };
    double host[2];
    auto dot = [&](const double *a, const double *b) {
        VecList L{};
        L.v[0] = b;
        launch_mdot(st, a, L, 1, n, scratch, d_out);
        HIPCHK(hipMemcpyAsync(host, d_out, sizeof(double), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        return host[0];
    };
    run(1);                                   // z = D^-1 r
    launch_copy(st, p, z, n);
    double rz = dot(r, z);
    std::vector<double> alpha, beta;
    auto ritz = [&](double *lo_, double *hi_) {
        const int m = (int)alpha.size();
        std::vector<double> d(m), e(m > 1 ? m - 1 : 0);
        for (int k = 0; k < m; ++k) {
            d[k] = 1.0 / alpha[k] + (k > 0 ? beta[k - 1] / alpha[k - 1] : 0.0);
            if (k + 1 < m) e[k] = std::sqrt(beta[k]) / alpha[k];
        }
        const std::vector<double> ev = tridiag_eigenvalues(d, e);
        *lo_ = ev.front();
        *hi_ = ev.back();
    };
    double last_lo = 0.0;
    for (int k = 0; k < max_steps && rz > 0.0 && std::isfinite(rz); ++k) {
        // the smallest Ritz value is the slow one: stop when it moved by less than 1 % over the
        // last 10 steps
        if (k >= 30 && k % 10 == 0) {
            double lo_, hi_;
            ritz(&lo_, &hi_);
            if (last_lo > 0.0 && std::fabs(lo_ - last_lo) <= 0.01 * last_lo) break;
            last_lo = lo_;
        }
        run(0);                               // w = A p
        const double pAp = dot(p, w);
        if (!(pAp > 0.0) || !std::isfinite(pAp)) break;
        const double a = rz / pAp;
        alpha.push_back(a);
        launch_axpby(st, r, -a, w, 1.0, n);   // r -= a w
        run(1);                               // z = D^-1 r
        const double rz_new = dot(r, z);
        if (!(rz_new > 1e-28 * rz) || !std::isfinite(rz_new)) break;
        const double b = rz_new / rz;
        beta.push_back(b);
        launch_axpby(st, p, 1.0, z, b, n);    // p = z + b p
        rz = rz_new;
    }
    for (double *q : {r, z, p, w, scratch, d_out}) (void)hipFree(q);
    (void)hipFree(d_zoff);
    (void)hipFree(d_ops);
    Spectrum out{0.0, 0.0, (int)alpha.size()};
    const int m = (int)alpha.size();
    if (m == 0) return out;
    ritz(&out.emin, &out.emax);
    return out;
}

// rho(D^-1 S) for a skew-symmetric S (the convection part of a sub-solve matrix): its eigenvalues
// are +- i mu_j, so (D^-1 S)^2 is similar to a negative semi-definite matrix with eigenvalues
// -mu_j^2; power iteration v <- (D^-1 S)^2 v, mu^2 ~ ||(D^-1 S)^2 v|| / ||v||.  Two SpMVs and one
// reduction per step; the estimate approaches mu_max from below (the caller widens it).
double jacobi_skew_radius(System &S, int pattern, const double *skew_vals, const double *dinv,
                          const uint8_t *rowmask, int max_steps, int *steps_out) {
    const Pattern &P = S.patterns[pattern];
    const int64_t n = P.nrows;
    hipStream_t st = S.stream;
    auto vec = [&]() {
        double *p = dev_alloc<double>(n + 32);
        HIPCHK(hipMemsetAsync(p, 0, (n + 32) * sizeof(double), st));
        return p;
    };
    double *v = vec(), *z = vec();
    double *scratch = dev_alloc<double>((size_t)REDUCE_BLOCKS * MDOT_MAX);
    double *d_out = dev_alloc<double>(4);
    {
        std::vector<double> h(n);
        uint64_t sd = 0x2545f4914f6cdd1dull;
        for (int64_t i = 0; i < n; ++i) {
            sd = sd * 6364136223846793005ull + 1442695040888963407ull;
            h[i] = (double)(sd >> 11) / (double)(1ull << 53) - 0.5;
        }
        if (rowmask) {
            std::vector<uint8_t> m(n);
            HIPCHK(hipMemcpy(m.data(), rowmask, n, hipMemcpyDeviceToHost));
            for (int64_t i = 0; i < n; ++i)
                if (m[i]) h[i] = 0.0;
        }
        HIPCHK(hipMemcpy(v, h.data(), n * sizeof(double), hipMemcpyHostToDevice));
    }
    auto vabs = [](const double *q) { return q ? VRef{(int64_t)(uintptr_t)q, 0, 0} : VRef{0, -1, 0}; };
    // y = -D^-1 (0 - S x): the Chebyshev epilogue with c3 = -1 and no right-hand side
    RowOp op{};
    op.col = P.d_col;
    op.perm = P.d_perm;
    op.slice_off = P.d_slice_off;
    op.uniform_w = P.uniform_w;
    op.nrows = (int32_t)n;
    op.nslices = P.nslices;
    op.nterms = 1;
    op.mode = EPI_CHEB;
    op.t[0].vals = skew_vals;
    op.y2 = op.yin = op.z = op.mx = op.b = op.pk = op.pkm1 = vabs(nullptr);
    op.dinv = dinv;
    op.c1 = op.c2 = 0.0;
    op.c3 = -1.0;
    op.post1 = op.post2 = 1.0;
    op.rowmask = rowmask;
    RowOp both[2] = {op, op};
    both[0].t[0].x = vabs(v);
    both[0].y = vabs(z);
    both[1].t[0].x = vabs(z);
    both[1].y = vabs(v);
    RowOp *d_ops = dev_upload(both, 2);
    const Bases B{{nullptr, nullptr, nullptr, nullptr}};
    double host[2];
    auto norm = [&](const double *a) {
        VecList L{};
        L.v[0] = a;
        launch_mdot(st, a, L, 1, n, scratch, d_out);
        HIPCHK(hipMemcpyAsync(host, d_out, sizeof(double), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        return std::sqrt(host[0]);
    };
    double nv = norm(v), mu2 = 0.0, last = -1.0;
    int k = 0;
    for (; k < max_steps && nv > 0.0 && std::isfinite(nv); ++k) {
        launch_axpby(st, v, 0.0, z, 1.0 / nv, n);      // v /= ||v||
        launch_rowops(st, d_ops, 1, P.nslices, P.R, B, 1, P.uniform_w);       // z = D^-1 S v
        launch_rowops(st, d_ops + 1, 1, P.nslices, P.R, B, 1, P.uniform_w);   // v = D^-1 S z
        nv = norm(v);
        mu2 = nv;
        if (k >= 8 && std::fabs(mu2 - last) <= 0.005 * mu2) {
            ++k;
            break;
        }
        last = mu2;
    }
    if (steps_out) *steps_out = k;
    for (double *q : {v, z, scratch, d_out}) (void)hipFree(q);
    (void)hipFree(d_ops);
    return (mu2 > 0.0 && std::isfinite(mu2)) ? std::sqrt(mu2) : 0.0;
}

}  // namespace kkt
