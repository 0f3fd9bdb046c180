// CPU emulation of the tile sweep program (control_amd/csrc/tile_kernels.hip) on the plan that
// control_amd/csrc/tiles.cpp builds: same loop structure (rings, credit, hand-offs of the newest
// and the previous iterate), tiles advanced in lock step.  Compared bit for bit with the plain
// global recurrence.  Test infrastructure: checks the plan and the scheme, not the GPU kernel.
//   usage: tile_emu <nx> <ny> <ntiles> <depth (0 = auto)> <threads> <its> <nlevels>
//                   [mask-aware tiles 0/1] [components: interleaved uncoupled copies of the grid]
//                   [nz: > 1 = 3-D grid nx x ny x nz with the 15-point structure of Kuhn cubes]
//                   [coordinates 0/1: tiles by coordinate bisection instead of graph bisection]
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../control_amd/csrc/tiles.hpp"

using namespace kkt;
namespace kkt {
void fail(int, const std::string &m) { std::fprintf(stderr, "fail: %s\n", m.c_str()); std::exit(2); }
void hip_check(hipError_t, const char *, const char *, int) {}
}

int main(int argc, char **argv) {
    const int nx = argc > 1 ? std::atoi(argv[1]) : 33, ny = argc > 2 ? std::atoi(argv[2]) : 29;
    const int ntiles = argc > 3 ? std::atoi(argv[3]) : 12, depth_in = argc > 4 ? std::atoi(argv[4]) : 0;
    const int T = argc > 5 ? std::atoi(argv[5]) : 64, its = argc > 6 ? std::atoi(argv[6]) : 11;
    const int nlev = argc > 7 ? std::atoi(argv[7]) : 3;
    // P1-like 7-point structure on an nx x ny grid (right-diagonal triangulation); with
    // `ncomp` > 1: that many uncoupled copies, interleaved node by node (a vector-valued block)
    const int ncomp = argc > 9 ? std::atoi(argv[9]) : 1;
    const int nz = argc > 10 ? std::max(1, std::atoi(argv[10])) : 1;
    Pattern P;
    const int n = nx * ny * nz * ncomp;
    P.nrows = P.ncols = n;
    P.R = 2;
    P.h_indptr.push_back(0);
    const int dx[7] = {-1, 0, -1, 0, 1, 0, 1}, dy[7] = {-1, -1, 0, 0, 0, 1, 1};
    // Kuhn cubes: the three axes, the three face diagonals and the space diagonal, both ways
    const int ex[7] = {1, 0, 0, 1, 0, 1, 1}, ey[7] = {0, 1, 0, 1, 1, 0, 1}, ez[7] = {0, 0, 1, 0, 1, 1, 1};
    const int WS = nz > 1 ? 15 : 7;
    for (int k = 0; k < nz; ++k)
        for (int j = 0; j < ny; ++j)
            for (int i = 0; i < nx; ++i)
                for (int c = 0; c < ncomp; ++c) {
                    if (nz == 1) {
                        for (int q = 0; q < 7; ++q) {
                            const int ii = i + dx[q], jj = j + dy[q];
                            if (ii >= 0 && ii < nx && jj >= 0 && jj < ny)
                                P.h_indices.push_back((jj * nx + ii) * ncomp + c);
                        }
                    } else {
                        std::vector<int32_t> row;
                        row.push_back(((k * ny + j) * nx + i) * ncomp + c);
                        for (int q = 0; q < 7; ++q)
                            for (int sg = -1; sg <= 1; sg += 2) {
                                const int ii = i + sg * ex[q], jj = j + sg * ey[q], kk = k + sg * ez[q];
                                if (ii >= 0 && ii < nx && jj >= 0 && jj < ny && kk >= 0 && kk < nz)
                                    row.push_back(((kk * ny + jj) * nx + ii) * ncomp + c);
                            }
                        std::sort(row.begin(), row.end());
                        P.h_indices.insert(P.h_indices.end(), row.begin(), row.end());
                    }
                    P.h_indptr.push_back((int32_t)P.h_indices.size());
                }
    P.nnz = P.h_indices.size();
    P.max_width = WS;
    P.uniform_w = WS;
    P.nslices = (n + 127) / 128;
    for (int s = 0; s <= P.nslices; ++s) P.h_slice_off.push_back(WS * s);
    P.npadded = (int64_t)WS * P.nslices * 128;
    // boundary rows: masked (their iterates are zero); argv[8] = 0 keeps them inside the tiles
    std::vector<uint8_t> mask(n, 0);
    for (int k = 0; k < nz; ++k)
        for (int j = 0; j < ny; ++j)
            for (int i = 0; i < nx; ++i)
                if (i == 0 || j == 0 || i == nx - 1 || j == ny - 1 ||
                    (nz > 1 && (k == 0 || k == nz - 1)))
                    for (int c = 0; c < ncomp; ++c) mask[((k * ny + j) * nx + i) * ncomp + c] = 1;
    const bool mask_aware = argc > 8 ? std::atoi(argv[8]) != 0 : true;
    const bool use_coords = argc > 11 && std::atoi(argv[11]) != 0;
    std::vector<double> xyz;
    if (use_coords)
        for (int k = 0; k < nz; ++k)
            for (int j = 0; j < ny; ++j)
                for (int i = 0; i < nx; ++i)
                    for (int c = 0; c < ncomp; ++c) {
                        xyz.push_back(i);
                        xyz.push_back(j);
                        xyz.push_back(k);
                    }
    TilePlan tp;
    if (!build_tile_plan(P, ntiles, depth_in, T, 4, tp, mask_aware ? mask.data() : nullptr, 0,
                         use_coords ? xyz.data() : nullptr, 3)) {
        std::printf("plan does not fit\n");
        return 3;
    }
    std::printf("plan: %d tiles depth %d rpt %d nk_pad %d symmetric %d max own %lld rows %lld halo %lld red %.2f\n",
                tp.ntiles, tp.depth, tp.rpt, tp.nk_pad, (int)tp.symmetric, (long long)tp.max_own,
                (long long)tp.max_rows, (long long)tp.max_halo, tp.mean_redundancy);
    if (!tp.symmetric) return 4;
    const int depth = tp.depth, W = tp.W, RPT = tp.rpt, nkp = tp.nk_pad;
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    // SELL value arrays per level (F) and update matrices, masks
    auto rand_vals = [&]() {
        std::vector<double> v(P.npadded, 0.0);
        for (int r = 0; r < n; ++r)
            for (int k = 0; k < P.h_indptr[r + 1] - P.h_indptr[r]; ++k) {
                const int c = P.h_indices[P.h_indptr[r] + k];
                v[P.sell_index(r, k)] = mask[c] ? 0.0 : (c == r ? 4.0 + U(rng) : 0.3 * U(rng));
            }
        return v;
    };
    std::vector<std::vector<double>> F(nlev), Um(nlev), dinv(nlev), B(nlev), out_ref(nlev), out_emu(nlev);
    for (int l = 0; l < nlev; ++l) {
        F[l] = rand_vals();
        Um[l] = rand_vals();
        dinv[l].assign(n, 1.0);
        for (int r = 0; r < n; ++r)
            if (!mask[r])
                for (int k = 0; k < P.h_indptr[r + 1] - P.h_indptr[r]; ++k)
                    if (P.h_indices[P.h_indptr[r] + k] == r) dinv[l][r] = 1.0 / F[l][P.sell_index(r, k)];
        B[l].resize(n);
        for (auto &x : B[l]) x = U(rng);
        out_ref[l].assign(n, 0.0);
        out_emu[l].assign(n, 0.0);
    }
    std::vector<double> c1(its + 1), c2(its + 1), c3(its + 1);
    for (int s = 2; s <= its; ++s) { c1[s] = 0.1 * U(rng); c2[s] = 1.0 + 0.1 * U(rng); c3[s] = 0.4 + 0.1 * U(rng); }
    const double scale = 0.37, ca = -1.0, cy = 1.0;
    auto spmv_row = [&](const std::vector<double> &vals, const std::vector<double> &x, int r) {
        double acc = 0.0;
        for (int k = 0; k < W; ++k) {
            const int len = P.h_indptr[r + 1] - P.h_indptr[r];
            const double v = k < len ? vals[P.sell_index(r, k)] : 0.0;
            const int c = k < len ? P.h_indices[P.h_indptr[r] + k] : r;
            acc = std::fma(v, x[c], acc);
        }
        return acc;
    };
    // ---- reference: the plain recurrence on global vectors
    {
        std::vector<double> prev(n, 0.0), b(n), pa(n), pb(n), pn(n);
        for (int l = 0; l < nlev; ++l) {
            for (int r = 0; r < n; ++r) {
                if (l > 0) {
                    const double acc = spmv_row(Um[l], prev, r);
                    double t = ca * acc;
                    t = std::fma(cy, B[l][r], t);
                    b[r] = mask[r] ? 0.0 : t;
                    pb[r] = scale * (dinv[l][r] * b[r]);
                } else {
                    b[r] = B[l][r];
                    double t = 0.0;
                    t = std::fma(scale, dinv[l][r] * (b[r] - 0.0), t);
                    pb[r] = mask[r] ? 0.0 : 1.0 * (1.0 * t);
                }
            }
            for (int s = 2; s <= its; ++s) {
                for (int r = 0; r < n; ++r) {
                    const double acc = spmv_row(F[l], pb, r);
                    double t = s >= 3 ? c1[s] * pa[r] : 0.0;
                    t = std::fma(c2[s], pb[r], t);
                    t = std::fma(c3[s], dinv[l][r] * (b[r] - acc), t);
                    pn[r] = mask[r] ? 0.0 : 1.0 * (1.0 * t);
                }
                pa.swap(pb);
                pb.swap(pn);
            }
            out_ref[l] = pb;
            prev = pb;
        }
    }
    // ---- emulation of the kernel, tiles in lock step
    const int NT = tp.ntiles;
    std::vector<std::vector<double>> X(NT, std::vector<double>(2 * (size_t)nkp, NAN));
    for (int t = 0; t < NT; ++t) X[t][nkp - 1] = X[t][2 * (size_t)nkp - 1] = 0.0;   // zero slot
    std::vector<int> cur(NT, 0);
    std::vector<std::vector<double>> bl(NT, std::vector<double>((size_t)RPT * T, NAN));
    std::vector<double> Gn(n, NAN), Go(n, NAN);   // granule buffers (values), tags implied by lock step
    auto nt = [&](int t, int j) { return tp.n[(size_t)t * (TILE_MAX_DEPTH + 1) + j]; };
    auto grow = [&](int t, int l) { return tp.grow[(size_t)t * nkp + l]; };
    auto handoff = [&](bool both) {
        for (int t = 0; t < NT; ++t)
            for (int r = 0; r < nt(t, 0); ++r) {
                Gn[grow(t, r)] = X[t][cur[t] * nkp + r];
                if (both) Go[grow(t, r)] = X[t][(cur[t] ^ 1) * nkp + r];
            }
        for (int t = 0; t < NT; ++t)
            for (int l = nt(t, 0); l < nt(t, depth); ++l) {
                X[t][cur[t] * nkp + l] = Gn[grow(t, l)];
                if (both && l < nt(t, depth - 1)) X[t][(cur[t] ^ 1) * nkp + l] = Go[grow(t, l)];
            }
    };
    auto lval = [&](const std::vector<double> &vals, int t, int r, int k) {
        const int sl = r / T, tid = r % T;
        const size_t at = (((size_t)t * RPT + sl) * W + k) * T + tid;
        const int g = tp.gpos[at];
        return std::make_pair(g >= 0 ? vals[g] : 0.0, (int)tp.lcol[at]);
    };
    for (int l = 0; l < nlev; ++l) {
        int cr = 0;
        for (int t = 0; t < NT; ++t) {
            double *Xc = &X[t][cur[t] * nkp], *Xo = &X[t][(cur[t] ^ 1) * nkp];
            const int nk1 = nt(t, depth - 1);
            for (int r = 0; r < nk1; ++r) {
                const int g = grow(t, r);
                if (l > 0) {
                    double acc = 0.0;
                    for (int k = 0; k < W; ++k) { auto vc = lval(Um[l], t, r, k); acc = std::fma(vc.first, Xc[vc.second], acc); }
                    double tt = ca * acc;
                    tt = std::fma(cy, B[l][g], tt);
                    const double o = mask[g] ? 0.0 : tt;
                    bl[t][r] = o;
                    Xo[r] = scale * (dinv[l][g] * o);
                } else {
                    bl[t][r] = B[l][g];
                    double tt = 0.0;
                    tt = std::fma(scale, dinv[l][g] * (bl[t][r] - 0.0), tt);
                    Xo[r] = mask[g] ? 0.0 : 1.0 * (1.0 * tt);
                }
            }
            cur[t] ^= 1;
        }
        cr = depth - 1;
        for (int s = 2; s <= its; ++s) {
            if (cr == 0) { handoff(s >= 3); cr = depth; }
            for (int t = 0; t < NT; ++t) {
                double *Xc = &X[t][cur[t] * nkp], *Xo = &X[t][(cur[t] ^ 1) * nkp];
                const int nv = nt(t, cr - 1);
                for (int r = 0; r < nv; ++r) {
                    const int g = grow(t, r);
                    double acc = 0.0;
                    for (int k = 0; k < W; ++k) { auto vc = lval(F[l], t, r, k); acc = std::fma(vc.first, Xc[vc.second], acc); }
                    double tt = s >= 3 ? c1[s] * Xo[r] : 0.0;
                    tt = std::fma(c2[s], Xc[r], tt);
                    tt = std::fma(c3[s], dinv[l][g] * (bl[t][r] - acc), tt);
                    Xo[r] = mask[g] ? 0.0 : 1.0 * (1.0 * tt);
                }
                // rows beyond the valid region are stale: poison them so that a wrong read shows
                for (int r = nv; r < nt(t, depth); ++r) Xo[r] = NAN;
                Xo[nkp - 1] = 0.0;
                cur[t] ^= 1;
            }
            --cr;
        }
        for (int t = 0; t < NT; ++t)
            for (int r = 0; r < nt(t, 0); ++r) out_emu[l][grow(t, r)] = X[t][cur[t] * nkp + r];
        if (l + 1 < nlev) handoff(false);
    }
    long bad = 0;
    for (int l = 0; l < nlev; ++l)
        for (int r = 0; r < n; ++r) {
            const double a = out_ref[l][r], b = out_emu[l][r];
            if (!(a == b)) ++bad;
        }
    std::printf("mismatches: %ld of %d\n", bad, nlev * n);
    return bad ? 1 : 0;
}
