// Tile plan of the communication-avoiding sweep program (tile_kernels.hip).
//
// The time sweeps of the block-Schur preconditioner (reference control/control.py:2263-2295,
// 2375-2406) are chains of dependent spatial SpMV steps on one N_x-row block: the update
// b_i -= A u_{i-1} and `schur_its` Jacobi-Chebyshev steps per time level.  With one hand-off
// between workgroups per step the chain is bound by the hand-off latency (~2.3 us per step,
// DESIGN.md section 6).  Here the rows of the block are cut into one compact TILE per
// workgroup (graph partition of the sparsity structure), and a tile also holds the rows within
// graph distance `depth` of its own rows (its rings).  After one exchange of the newest
// iterates on the rings, a workgroup advances `depth` dependent SpMV steps out of LDS, on a
// region that shrinks by one ring per step -- redundant flops on the rings instead of hand-offs
// (a matrix-powers kernel).  Every row keeps the fma chain of the plain kernels, so results
// are bit-identical.
#pragma once
#include <stdint.h>

#include <vector>

#include "system.hpp"

namespace kkt {

constexpr int TILE_MAX_DEPTH = 16;

struct TilePlan {
    int ntiles = 0, depth = 0, threads = 0, W = 0;
    int rpt = 0;          // row slots per thread: rows [0, n[depth-1]) of a tile are computed
    int hslots = 0;       // ring-entry slots per thread a hand-off needs: ceil(max ring rows / threads)
    int nk_pad = 0;       // LDS vector length: max over tiles of n[depth], padded
    bool symmetric = false;   // tile A gathers from tile B <=> B gathers from A
    // host copies (per tile, fixed strides)
    std::vector<int32_t> n;       // [ntiles][TILE_MAX_DEPTH + 1]: n[j] = rows within distance j
    std::vector<int32_t> grow;    // [ntiles][nk_pad]: global row of a local index (-1: none)
    std::vector<uint16_t> lcol;   // [ntiles][rpt][W][threads]: local column of entry k of a row
    std::vector<int32_t> gpos;    // same shape: index into a SELL value array (-1: padding)
    std::vector<int32_t> part;    // tile of every global row
    // device copies
    int32_t *d_n = nullptr, *d_grow = nullptr, *d_gpos = nullptr;
    uint16_t *d_lcol = nullptr;
    int64_t max_own = 0, max_rows = 0, max_halo = 0;   // statistics (largest tile)
    double mean_redundancy = 0.0;                      // mean n[depth-1] / n[0]
    double model_us = 0.0;                             // modelled microseconds per dependent step
    void upload();
    void release();
};

// Partition the rows of `P` into `ntiles` compact parts (recursive graph bisection along the
// difference of the BFS distances from two far-apart rows) and build the rings up to `depth`.
// `depth` <= 0: the depth that minimises the modelled time per step (hand-off + depth local
// steps on the shrinking region) among those that fit.  Returns false when the structure does
// not fit the kernel's limits (local indices are 16-bit; `threads` * max_rpt rows per tile;
// threads * (max_rpt + 1) halo rows).
// `mask` (may be null): rows with mask[r] != 0 -- Dirichlet rows, whose iterates are exactly zero
// and whose columns are zeroed in every matrix of a sweep -- belong to no tile and to no ring;
// columns that point at them read the tile's permanent zero slot (local index nk_pad - 1).
// `its` (0: unknown): dependent steps per level, for the depth model (hand-offs counted whole).
// `coords` (may be null; nrows x dim): the rows' coordinates -- the parts are then boxes from
// recursive coordinate bisection (largest extent first) instead of graph bisection.
bool build_tile_plan(const Pattern &P, int ntiles, int depth, int threads, int max_rpt,
                     TilePlan &out, const uint8_t *mask = nullptr, int its = 0,
                     const double *coords = nullptr, int dim = 0,
                     int (*max_hslots)(int W, int rpt, int threads) = nullptr,
                     int extra_handoffs = 0);
// `max_hslots` (may be null: rpt + 1): ring-entry slots per thread the kernel variant with `rpt`
// row slots has; a plan whose rings need more is not made (a deeper one would lose the tile form).

}  // namespace kkt
