// Host side of the tile sweep program: graph partition of a sparsity structure into compact
// tiles, their rings, and the tile-local matrix layout (tiles.hpp).
#include "tiles.hpp"

#include <algorithm>
#include <cstdlib>
#include <numeric>
#include <set>

namespace kkt {

namespace {

struct Bisector {
    const Pattern &P;
    std::vector<int32_t> &part;
    std::vector<int32_t> stamp, dist, queue;
    int32_t cur = 0;

    // (rows left out of the partition never carry the current stamp, so the searches pass
    // around them)
    Bisector(const Pattern &P_, std::vector<int32_t> &part_)
        : P(P_), part(part_), stamp(P_.nrows, 0), dist(P_.nrows, 0) {
        queue.reserve(P_.nrows);
    }

    // BFS inside the subset stamped `cur`; unreached members get max + 1.  Returns a farthest
    // reached node.
    int32_t bfs(const std::vector<int32_t> &nodes, int32_t src) {
        for (int32_t v : nodes) dist[v] = -1;
        queue.clear();
        queue.push_back(src);
        dist[src] = 0;
        int32_t far = src;
        for (size_t h = 0; h < queue.size(); ++h) {
            const int32_t v = queue[h];
            far = v;
            for (int32_t q = P.h_indptr[v]; q < P.h_indptr[v + 1]; ++q) {
                const int32_t c = P.h_indices[q];
                if (c < (int32_t)P.nrows && stamp[c] == cur && dist[c] < 0) {
                    dist[c] = dist[v] + 1;
                    queue.push_back(c);
                }
            }
        }
        const int32_t big = dist[far] + 1;
        for (int32_t v : nodes)
            if (dist[v] < 0) dist[v] = big;
        return far;
    }

    void run(std::vector<int32_t> &nodes, int nparts, int base) {
        if (nparts <= 1 || nodes.size() <= 1) {
            for (int32_t v : nodes) part[v] = base;
            return;
        }
        ++cur;
        for (int32_t v : nodes) stamp[v] = cur;
        const int32_t a = bfs(nodes, nodes[0]);
        // A disconnected subset (vector-valued blocks whose components do not couple are copies
        // of one graph side by side): the component of the first row gets its share of the
        // parts, the rest theirs.  Ordered along one axis, the unreached rows would all sit at
        // "distance difference 0" and end up in tiles made of pieces of both components
        // (P2 velocity block 128^2: 777 ring rows around 509 own rows).
        // (only where both sides deserve a part of their own: a half of a connected mesh often
        // has slivers cut off from it, and those stay with the ordering below)
        const int nr = (int)(((int64_t)nparts * (int64_t)queue.size() + (int64_t)nodes.size() / 2) /
                             (int64_t)nodes.size());
        auto balanced = [&]() {      // no part more than 5 % above the average size
            const double avg = (double)nodes.size() / nparts;
            const double sr = (double)queue.size() / nr;
            const double su = (double)(nodes.size() - queue.size()) / (nparts - nr);
            return sr <= 1.05 * avg && su <= 1.05 * avg;
        };
        if (queue.size() < nodes.size() && nr >= 1 && nr <= nparts - 1 && balanced()) {
            std::vector<int32_t> reached(queue.begin(), queue.end()), rest;
            rest.reserve(nodes.size() - reached.size());
            const int32_t big = dist[a] + 1;
            for (int32_t v : nodes)
                if (dist[v] == big) rest.push_back(v);
            std::sort(reached.begin(), reached.end());
            std::vector<int32_t>().swap(nodes);
            run(reached, nr, base);
            run(rest, nparts - nr, base + nr);
            return;
        }
        const int32_t b = bfs(nodes, a);
        std::vector<int32_t> da(nodes.size());
        for (size_t i = 0; i < nodes.size(); ++i) da[i] = dist[nodes[i]];
        (void)bfs(nodes, b);
        // order along the "axis" between the two far-apart rows: d(a, .) - d(b, .)
        std::vector<int32_t> idx(nodes.size());
        std::iota(idx.begin(), idx.end(), 0);
        std::vector<int64_t> key(nodes.size());
        for (size_t i = 0; i < nodes.size(); ++i)
            key[i] = (int64_t)(da[i] - dist[nodes[i]]) * ((int64_t)1 << 40) +
                     (int64_t)da[i] * ((int64_t)1 << 32) + nodes[i];
        std::sort(idx.begin(), idx.end(), [&](int32_t x, int32_t y) { return key[x] < key[y]; });
        const int nl = nparts / 2;
        const size_t cut = (size_t)(((int64_t)nodes.size() * nl + nparts / 2) / nparts);
        std::vector<int32_t> left(cut), right(nodes.size() - cut);
        for (size_t i = 0; i < cut; ++i) left[i] = nodes[idx[i]];
        for (size_t i = cut; i < nodes.size(); ++i) right[i - cut] = nodes[idx[i]];
        std::vector<int32_t>().swap(nodes);
        run(left, nl, base);
        run(right, nparts - nl, base + nl);
    }
};

}  // namespace

void TilePlan::upload() {
    release();
    d_n = dev_upload(n.data(), n.size());
    d_grow = dev_upload(grow.data(), grow.size());
    d_gpos = dev_upload(gpos.data(), gpos.size());
    d_lcol = dev_upload(lcol.data(), lcol.size());
}

void TilePlan::release() {
    auto F = [](void *p) {
        if (p) (void)hipFree(p);
    };
    F(d_n);
    F(d_grow);
    F(d_gpos);
    F(d_lcol);
    d_n = d_grow = d_gpos = nullptr;
    d_lcol = nullptr;
}

bool build_tile_plan(const Pattern &P, int ntiles, int depth, int threads, int max_rpt,
                     TilePlan &out, const uint8_t *mask, int its, const double *coords, int dim,
                     int (*max_hslots)(int W, int rpt, int threads), int extra_handoffs) {
    if (P.nrows != P.ncols || ntiles < 1 || depth > TILE_MAX_DEPTH) return false;
    const bool auto_depth = depth <= 0;
    if (auto_depth) depth = TILE_MAX_DEPTH;
    if ((int64_t)ntiles > P.nrows) ntiles = (int)P.nrows;
    const int64_t nrows = P.nrows;
    out = TilePlan{};
    out.ntiles = ntiles;
    out.depth = depth;
    out.threads = threads;
    out.W = P.max_width;
    if (out.W < 1) return false;
    out.part.assign(nrows, -1);
    {
        std::vector<int32_t> all;
        all.reserve(nrows);
        for (int64_t r = 0; r < nrows; ++r)
            if (!mask || !mask[r]) all.push_back((int32_t)r);
        if ((int64_t)ntiles > (int64_t)all.size()) ntiles = (int)std::max<size_t>(1, all.size());
        out.ntiles = ntiles;
        if (coords != nullptr && dim >= 1) {
            // recursive coordinate bisection: cut the longest extent at the proportional
            // position (ties by the other coordinates, then by row: deterministic)
            struct Rcb {
                const double *c;
                int dim;
                std::vector<int32_t> &part;
                void run(std::vector<int32_t> &nodes, int nparts, int base) {
                    if (nparts <= 1 || nodes.size() <= 1) {
                        for (int32_t v : nodes) part[v] = base;
                        return;
                    }
                    int ax = 0;
                    double best = -1.0;
                    for (int a = 0; a < dim; ++a) {
                        double lo = 1e300, hi = -1e300;
                        for (int32_t v : nodes) {
                            lo = std::min(lo, c[(size_t)v * dim + a]);
                            hi = std::max(hi, c[(size_t)v * dim + a]);
                        }
                        if (hi - lo > best) {
                            best = hi - lo;
                            ax = a;
                        }
                    }
                    std::sort(nodes.begin(), nodes.end(), [&](int32_t x, int32_t y) {
                        for (int q = 0; q < dim; ++q) {
                            const int a = (ax + q) % dim;
                            const double cx = c[(size_t)x * dim + a], cy = c[(size_t)y * dim + a];
                            if (cx != cy) return cx < cy;
                        }
                        return x < y;
                    });
                    const int nl = nparts / 2;
                    const size_t cut = (size_t)(((int64_t)nodes.size() * nl + nparts / 2) / nparts);
                    std::vector<int32_t> left(nodes.begin(), nodes.begin() + cut),
                        right(nodes.begin() + cut, nodes.end());
                    std::vector<int32_t>().swap(nodes);
                    run(left, nl, base);
                    run(right, nparts - nl, base + nl);
                }
            } rcb{coords, dim, out.part};
            // Rows that share their coordinates are the components of a vector-valued space: every
            // component gets tiles of its own (a tile of both has the rings of both).  Component
            // = rank of a row among the rows at its point.
            std::vector<int32_t> order(all);
            std::sort(order.begin(), order.end(), [&](int32_t x, int32_t y) {
                for (int a = 0; a < dim; ++a) {
                    const double cx = coords[(size_t)x * dim + a], cy = coords[(size_t)y * dim + a];
                    if (cx != cy) return cx < cy;
                }
                return x < y;
            });
            auto same_point = [&](int32_t x, int32_t y) {
                for (int a = 0; a < dim; ++a)
                    if (coords[(size_t)x * dim + a] != coords[(size_t)y * dim + a]) return false;
                return true;
            };
            std::vector<std::vector<int32_t>> comp;
            for (size_t i = 0, k = 0; i < order.size(); ++i) {
                k = (i > 0 && same_point(order[i - 1], order[i])) ? k + 1 : 0;
                if (comp.size() <= k) comp.resize(k + 1);
                comp[k].push_back(order[i]);
            }
            bool even = comp.size() > 1 && (int)comp.size() <= ntiles && ntiles % (int)comp.size() == 0;
            for (const auto &c : comp) even = even && c.size() == comp[0].size();
            if (even) {
                const int per = ntiles / (int)comp.size();
                for (size_t k = 0; k < comp.size(); ++k) rcb.run(comp[k], per, (int)k * per);
            } else {
                rcb.run(all, ntiles, 0);
            }
        } else {
            Bisector B(P, out.part);
            B.run(all, ntiles, 0);
        }
    }
    // rings: L_0 = own rows, L_{j+1} = L_j + columns of the rows of L_j
    std::vector<std::vector<int32_t>> local(ntiles);   // global rows in local order
    out.n.assign((size_t)ntiles * (TILE_MAX_DEPTH + 1), 0);
    {
        std::vector<std::vector<int32_t>> own(ntiles);
        for (int64_t r = 0; r < nrows; ++r)
            if (out.part[r] >= 0) own[out.part[r]].push_back((int32_t)r);
        std::vector<int32_t> mark(nrows, -1);
        for (int t = 0; t < ntiles; ++t) {
            std::vector<int32_t> &L = local[t];
            L = own[t];
            for (int32_t r : L) mark[r] = t;
            int32_t *nt = &out.n[(size_t)t * (TILE_MAX_DEPTH + 1)];
            nt[0] = (int32_t)L.size();
            size_t first = 0;
            for (int j = 1; j <= depth; ++j) {
                const size_t last = L.size();
                std::vector<int32_t> ring;
                for (size_t i = first; i < last; ++i)
                    for (int32_t q = P.h_indptr[L[i]]; q < P.h_indptr[L[i] + 1]; ++q) {
                        const int32_t c = P.h_indices[q];
                        if (mark[c] != t && !(mask && mask[c])) {
                            mark[c] = t;
                            ring.push_back(c);
                        }
                    }
                std::sort(ring.begin(), ring.end());
                L.insert(L.end(), ring.begin(), ring.end());
                nt[j] = (int32_t)L.size();
                first = last;
            }
            for (int j = depth + 1; j <= TILE_MAX_DEPTH; ++j) nt[j] = nt[depth];
        }
    }
    // Modelled microseconds per dependent step: one hand-off per `d` steps plus the local steps,
    // fitted to per-tile timings on MI355X (256^2 P1, depths 4 .. 8, 16-byte granule polls;
    // DESIGN.md section 6).  A hand-off costs a + b per 1 000 granule pairs gathered (the newest
    // iterate on all rings, the previous one on all but the outermost), a local step c + e per
    // 1 000 rows computed: 1 024-thread workgroups 1.68 + 1.32 and 0.34 + 0.12, 512-thread
    // workgroups 1.5 + 1.5 and 0.38 + 0.16.  With the steps per level known, the hand-offs of a
    // level are counted whole (ceil(its / d): 80 steps at depth 7 are 12 rounds, at depth 6: 14).
    // the ring entries of a tile are gathered by `hslots` slots per thread; a variant with `rp`
    // row slots has rp + 1 of them unless the kernel table says otherwise
    auto halo_fits = [&](int rp, int64_t ring_rows) {
        const int cap = max_hslots ? max_hslots(out.W, rp, threads) : rp + 1;
        return (int64_t)cap * threads >= ring_rows;
    };
    auto model = [&](int d, double *us) -> bool {
        int64_t mk = 0, mr = 0, mh = 0, mo = 0;
        for (int t = 0; t < ntiles; ++t) {
            const int32_t *nt = &out.n[(size_t)t * (TILE_MAX_DEPTH + 1)];
            mk = std::max<int64_t>(mk, nt[d]);
            mr = std::max<int64_t>(mr, nt[d - 1]);
            mh = std::max<int64_t>(mh, nt[d] - nt[0]);
            mo = std::max<int64_t>(mo, nt[d - 1] - nt[0]);
        }
        int rp = (int)((mr + threads - 1) / threads);
        while (rp <= max_rpt && !halo_fits(rp, mh)) ++rp;
        if (mk > 65535 || rp > max_rpt) return false;
        const bool big = threads > 512;
        const double handoff = (big ? 1.68 : 1.5) + (big ? 1.32e-3 : 1.5e-3) * (double)(mh + mo);
        // (the per-row part of a step is its gathers and fmas: proportional to the row width;
        // fitted at W = 7 -- with it the P2 blocks, W = 19, get depth 2, measured best: 16.2
        // against 15.7 outer its/s at depth 3 and 15.6 at depth 1)
        const double wf = (double)out.W / 7.0;
        const double step = big ? 0.34 + 0.12e-3 * wf * (double)mr : 0.38 + 0.16e-3 * wf * (double)mr;
        // (two-grid levels: a hand-off behind every correction and one in front of the next
        // residual or at the level's end -- `extra_handoffs` per `its` sweeps, whose rings grow
        // with the depth like the others'; measured on cfg 2: depth 5 118 its/s, depth 8 112)
        if (its > 0)
            *us = ((double)((its + d - 1) / d + extra_handoffs) * handoff + its * step) / its;
        else
            *us = (handoff + d * step) / d;
        return true;
    };
    if (auto_depth) {
        // the optimum is flat: the shallowest depth within 1 % of the best (fewer redundant rows,
        // smaller rings) is taken
        double best = 1e300, per[TILE_MAX_DEPTH + 1];
        int n_ok = 0, best_d = 0;
        for (int d = 1; d <= TILE_MAX_DEPTH; ++d) {
            if (!model(d, &per[d])) break;
            best = std::min(best, per[d]);
            n_ok = d;
        }
        for (int d = 1; d <= n_ok && !best_d; ++d)
            if (per[d] <= 1.01 * best) best_d = d;
        if (best_d == 0) return false;
        depth = best_d;
        out.depth = depth;
        for (int t = 0; t < ntiles; ++t) {
            int32_t *nt = &out.n[(size_t)t * (TILE_MAX_DEPTH + 1)];
            for (int j = depth + 1; j <= TILE_MAX_DEPTH; ++j) nt[j] = nt[depth];
            local[t].resize(nt[depth]);
        }
    }
    if (!model(depth, &out.model_us)) return false;
    int64_t max_nk = 0, max_rows = 0, max_halo = 0, max_own = 0;
    double red = 0.0;
    for (int t = 0; t < ntiles; ++t) {
        const int32_t *nt = &out.n[(size_t)t * (TILE_MAX_DEPTH + 1)];
        max_nk = std::max<int64_t>(max_nk, nt[depth]);
        max_rows = std::max<int64_t>(max_rows, nt[depth - 1]);
        max_halo = std::max<int64_t>(max_halo, nt[depth] - nt[0]);
        max_own = std::max<int64_t>(max_own, nt[0]);
        red += nt[0] ? (double)nt[depth - 1] / nt[0] : 0.0;
    }
    out.max_own = max_own;
    out.max_rows = max_rows;
    out.max_halo = max_halo;
    out.mean_redundancy = red / ntiles;
    if (max_nk + 64 > 65535) return false;
    int rpt = std::max(1, (int)((max_rows + threads - 1) / threads));
    while (rpt <= max_rpt && !halo_fits(rpt, max_halo)) ++rpt;
    if (rpt > max_rpt) return false;
    out.rpt = rpt;
    out.hslots = std::max(1, (int)((max_halo + threads - 1) / threads));
    out.nk_pad = (int)((max_nk + 1 + 63) & ~(int64_t)63);   // + the zero slot at nk_pad - 1
    const int W = out.W, T = threads;
    out.grow.assign((size_t)ntiles * out.nk_pad, -1);
    out.lcol.assign((size_t)ntiles * rpt * W * T, 0);
    out.gpos.assign((size_t)ntiles * rpt * W * T, -1);
    std::vector<int32_t> lidx(nrows, -1);
    for (int t = 0; t < ntiles; ++t) {
        const std::vector<int32_t> &L = local[t];
        const int32_t *nt = &out.n[(size_t)t * (TILE_MAX_DEPTH + 1)];
        for (size_t l = 0; l < L.size(); ++l) {
            lidx[L[l]] = (int32_t)l;
            out.grow[(size_t)t * out.nk_pad + l] = L[l];
        }
        const size_t base = (size_t)t * rpt * W * T;
        for (int r = 0; r < rpt * T; ++r) {
            const int slot = r / T, tid = r % T;
            const bool live = r < nt[depth - 1];
            const int32_t g = live ? L[r] : -1;
            const int len = live ? P.h_indptr[g + 1] - P.h_indptr[g] : 0;
            for (int k = 0; k < W; ++k) {
                const size_t at = base + ((size_t)slot * W + k) * T + tid;
                if (k < len) {
                    const int32_t c = P.h_indices[P.h_indptr[g] + k];
                    out.lcol[at] = (uint16_t)((mask && mask[c]) ? out.nk_pad - 1 : lidx[c]);
                    out.gpos[at] = (int32_t)P.sell_index(g, k);
                } else {
                    out.lcol[at] = (uint16_t)(live ? r : 0);   // value 0: any valid index
                    out.gpos[at] = -1;
                }
            }
        }
        for (int32_t g : L) lidx[g] = -1;
    }
    // the write-after-read argument of the granule buffers needs a symmetric "gathers from"
    // relation between tiles
    {
        std::vector<std::set<int32_t>> reads(ntiles);
        for (int t = 0; t < ntiles; ++t) {
            const int32_t *nt = &out.n[(size_t)t * (TILE_MAX_DEPTH + 1)];
            for (int l = nt[0]; l < nt[depth]; ++l)
                if (out.part[local[t][l]] >= 0) reads[t].insert(out.part[local[t][l]]);
        }
        out.symmetric = true;
        for (int t = 0; t < ntiles && out.symmetric; ++t)
            for (int32_t u : reads[t])
                if (!reads[u].count(t)) out.symmetric = false;
    }
    return true;
}

}  // namespace kkt
