// Device-kernel interface of libkkt (host-callable launchers).  gfx950 only.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include <string>

namespace kkt {

// thrown by launch_tile_sweep when a memset or the launch itself fails (never silently)
struct TileLaunchError {
    std::string msg;
};

constexpr int MAX_TERMS = 8;      // blocks summed into one block row by one launch
constexpr int MDOT_MAX = 8;       // vectors per fused multi-dot / multi-axpy pass
constexpr int REDUCE_BLOCKS = 1024;  // fixed partial count -> deterministic reductions

// Reference to a vector operand.  base < 0: null; base == 0: `off` holds an absolute
// device pointer; base >= 1: element offset `off` from Bases::p[base - 1], which the
// caller supplies per launch (so one RowOp array serves any input/output vector).
struct VRef {
    int64_t off;
    int32_t base;
    int32_t pad_;
};
struct Bases { const double *p[4]; };

struct SpmvTerm {
    const double *vals;   // padded SELL values (layout of the RowOp's pattern)
    VRef x;               // the column block's vector (length ncols)
};

enum EpiMode : int32_t { EPI_LIN = 0, EPI_CHEB = 1 };

// One block-row operation.  Arrays of RowOp live in device memory; one launch executes
// many of them (blockIdx.y selects the RowOp).  acc = sum_t A_t x_t is accumulated
// term-major with one fused-multiply-add chain per row, in CSR order within a term --
// the order of the reference's sequence of MatMultAdd calls (preconditioner.py:406-432).
struct RowOp {
    const int32_t *col;        // SELL column indices of the shared pattern
    const int32_t *slice_off;  // nslices + 1 offsets, in slots
    const int32_t *perm;       // row stored at each position (-1: none), null: position == row
    int32_t nrows, nslices, nterms, mode;
    int32_t uniform_w;         // >= 0: every slice has this width (slice_off unused)
    int32_t pad0_;
    SpmvTerm t[MAX_TERMS];
    VRef y;
    VRef y2;                   // EPI_LIN only, optional: y2 = c3 * dinv * y
    // EPI_LIN:  y = ca*acc + cy*yin + cz*z;  masked rows: y = malpha * mx[r] (0 if !mx)
    double ca, cy, cz;
    VRef yin, z;
    const uint8_t *rowmask;
    VRef mx;
    double malpha;
    // EPI_CHEB: y = post2*(post1*(c1*pkm1 + c2*pk + c3*dinv*(b - acc))); masked rows use
    //           the exact result of the bc-assembled matrix on a bc-clean rhs: 0
    VRef b, pkm1, pk;
    const double *dinv;
    double c1, c2, c3, post1, post2;
};

void launch_rowops(hipStream_t s, const RowOp *d_ops, int nops, int max_slices, int R,
                   const Bases &bases, int tag, int uniform_w,
                   const RowOp *h_single = nullptr);
// uniform_w of a launch of ragged structures whose slots lie (>= 75 %) in slices of a width
// ragged_switch_width() accepts: the operator apply then runs kkt_spmv_rows_ragged, where a wave
// picks the body unrolled for its slice's width (other widths: the slot loop, in the same kernel)
constexpr int UNIFORM_W_SWITCH = -2;
// the same with one wave per workgroup, for launches of wide slices (mean width >= 10): the slices
// of a sort window differ in width (P2: 19, 19, 12, 9, 9, ...) and a four-wave workgroup holds its
// registers until its widest slice is done (P2: 0.565 -> 0.510 ms); narrow slices (the B^T blocks
// of the Stokes system: 4 / 5 / 7) move too few bytes per wave to be dispatched one by one
constexpr int UNIFORM_W_SWITCH_1WAVE = -3;
// every slice of the launch at most 7 wide: kkt_spmv_rows_ragged_narrow (bodies 3 .. 7 only, so
// that more waves are resident)
constexpr int UNIFORM_W_SWITCH_NARROW = -4;
bool ragged_switch_width(int w);
// XCD-aware workgroup order in the ragged kernel (default on; option "ragged_xcd" = "0": dispatch
// order).  P2 0.507 -> 0.46-0.48 ms, Stokes outer operator 0.778 -> 0.729 ms.  (On the fixed-width
// P1 launches the same relabelling was slower, profiles/DIARY_r01_r02.md 8.1b: their traffic is
// 1.1 x the bytes already; the ragged launches fetched 1.4 x.)
void set_ragged_xcd(bool on);
void set_apply_xcd(bool on);
void set_pc_xcd(bool on);       // ... and for the batched preconditioner steps (option "pc_xcd")    // the same order for the fixed-width operator launches (option "apply_xcd")

// Batched Chebyshev steps on ONE matrix with the iterates of four time levels interleaved
// (element (row r, level l) of a group at 4 r + l): a gather serves four levels with one 32-byte
// load instead of four 8-byte ones.  Right-hand sides and final outputs keep the API layout.
struct IlOp {
    const int32_t *col, *slice_off, *perm;
    const double *vals, *dinv;
    const uint8_t *rowmask;
    int32_t nrows, nslices, uniform_w, nlev;   // nlev <= 4 levels in this group
    const double *b[4];        // right-hand sides of the levels
    double *out[4];            // out[0] != null: the step writes the levels' outputs (last step)
    const double *x;           // interleaved p_k (null: first step, no matrix term)
    const double *pkm1;        // interleaved p_{k-1} or null
    double *y;                 // interleaved output (when out[0] == null)
    double c1, c2, c3;
    double post1[4], post2[4]; // per level (the last step's output scalings)
};
void launch_rowops_il(hipStream_t s, const IlOp *d_ops, int ngroups, int max_slices, int uniform_w);

// Batched launch for RowOps that all have ONE term with the same matrix and pattern
// (uniform width 1..8, R = 2): four time levels per thread.  Returns false if not applicable.
bool launch_rowops_shared(hipStream_t s, const RowOp *d_ops, int nops, int max_slices, int R,
                          int uniform_w);

// Operator apply for shared values: d_groups[g] = {first op, count <= 4} of runs of EPI_LIN RowOps
// with identical structure (same pattern, same term matrices); fixed slice widths 1..8, R = 2.
constexpr int ROW_GROUP_MAX = 4;
bool launch_rowops_grouped(hipStream_t s, const RowOp *d_ops, const int32_t *d_groups, int ngroups,
                           int max_slices, int R, int uniform_w, const Bases &bases);

// One persistent launch that runs `nphases` single-block RowOps in order, workgroup j
// waiting before each phase for the workgroups d_dep[2j] .. d_dep[2j+1] (kernels.hip).
int prog_flag_words(int nwg);
// data-flow form (tagged granules instead of phase counters), uniform widths 1..16 only
bool row_program_g_available(int R, int uniform_w);
int row_program_g_max_wgs(int uniform_w, int waves_per_wg);
// Compact record of a phase, read through the scalar cache one phase ahead.  kind 1 (STEP): a
// Chebyshev step that inherits matrix, diagonal, right-hand side and mask from the previous
// phase, with p_k = the previous phase's output and p_{k-1} (flags bit 0) the one before.
struct PhaseLite {
    uint32_t kind, flags;
    uint64_t y;
    double c1, c2, c3, post1, post2;
    uint64_t pad_;
};
static_assert(sizeof(PhaseLite) == 64, "one scalar cache line");
void launch_row_program_g(hipStream_t s, const RowOp *d_ops, const PhaseLite *d_lite, int nphases,
                          int nwg, int waves_per_wg, int uniform_w, unsigned long long *g0,
                          unsigned long long *g1, size_t granule_words, unsigned *d_err);
int row_program_max_wgs(int R, int uniform_w, int waves_per_wg, bool lowreg = false);
// data-flow form for any width (R = 2): granule hand-off, matrix re-read from L2 every phase
int row_program_gw_max_wgs(int waves_per_wg);
void launch_row_program_gw(hipStream_t s, const RowOp *d_ops, int nphases, int nwg,
                           int waves_per_wg, unsigned long long *g0, unsigned long long *g1,
                           size_t granule_words, unsigned *d_err);
void launch_row_program(hipStream_t s, const RowOp *d_ops, int nphases, int nwg, int waves_per_wg,
                        int R, int uniform_w, const int32_t *d_dep, unsigned *d_flags,
                        unsigned *d_err, bool lowreg = false);

// ---- tile sweep program (tile_kernels.hip; plan: tiles.hpp)
constexpr int TILE_DEPTH_MAX = 16;
struct TileCoef { double c1, c2, c3; };
// One time level of a sweep: optional update b = ca * (sum_t U_t x_prev) + cy * bin on the
// boundary-masked rows, then `its` Jacobi-Chebyshev steps with the matrix `vals` on b, result in
// `out` (control.py:2263-2295 / 2375-2406: "b_i += M u_{i-1}; solve").
// (128 bytes, aligned: the two scalar-cache lines of a level hold one level only)
struct alignas(128) TileLevel {
    const double *vals;        // SELL values of the level's matrix F_i
    const double *dinv;        // its Jacobi diagonal, inverted (1 on boundary rows)
    const double *bin;         // right-hand side before the update
    double *bout;              // where the updated right-hand side is stored (may be bin; may be null)
    double *out;               // u_i
    const double *x_prev;      // the vector the update multiplies
    const double *upd_vals[2]; // SELL values of the update terms
    int32_t n_upd;             // 0: no update at this level
    int32_t prev_in_lds;       // x_prev is the previous level's `out` (handed over inside the launch)
    double ca, cy;
    double p1_scale;           // first step: p_1 = p1_scale * dinv * b
    double post1, post2;       // factors of the last step
    const TileCoef *coef;      // steps 2 .. its: p_s = c1 p_{s-2} + c2 p_{s-1} + c3 dinv (b - F p_{s-1})
};
// Coarse corrections inside the tile program (two-grid form of the sub-solves): what a tile needs
// to restrict its own rows' residual to the coarse functions they touch (J_t), to publish those
// partial sums, to assemble the whole coarse residual from every tile's partials in a fixed order,
// to apply its rows of (P^T A P)^-1 and to prolong onto its own rows.  Device-resident plan,
// built by SchurPC from the tile plan and P (csrc/pc.cpp, build_tile_coarse).
struct TileCoarseDev {
    int32_t nc, jmax, n0max, nslots;
    int32_t nr_max;               // most restriction (= prolongation) entries of a tile
    int32_t cache_lists;          // 1: every tile copies its entries into LDS (they fit)
    int32_t cache_einv;           // 1: ... and keeps the rows of (P^T A P)^-1 it owns there
    int32_t nown;                 // most coarse functions a tile owns: ceil(nc / ntiles)
    const int32_t *nj;            // [ntiles] number of coarse functions the own rows touch
    const int32_t *jglob;         // [ntiles][jmax] their global numbers
    const int32_t *slot0;         // [ntiles] first slot of the tile's partial sums
    const int32_t *r_ip;          // [ntiles * jmax + 1] restriction lists, per (tile, k) ...
    const uint16_t *r_row;        // ... local own row
    const double *r_w;            // ... and weight, ascending rows
    const int32_t *p_ip;          // [ntiles * n0max + 1] prolongation entries of the own rows ...
    const uint16_t *p_k;          // ... index into the tile's J_t
    const double *p_w;
    const int32_t *c_ip;          // [nc + 1] slots that contribute to a coarse function ...
    const int32_t *c_slot;        // ... ascending (tile order)
    unsigned long long *cg[2];    // granule buffers of the partial sums, 2 words per slot
    uint32_t cg_bytes;
    unsigned long long *eg[2];    // ... and of the products (E^-1 r_c)_j, 2 words per coarse function
    uint32_t eg_bytes;
    // Columns of row j of (P^T A P)^-1 that can be non-zero: [e_lo[j], e_hi[j]), e_lo a multiple of
    // 64.  P^T A P is block diagonal over the connected components of its graph (a vector-valued
    // space: one block per component of the field) and so is its inverse -- Gauss-Jordan with
    // partial pivoting leaves exact zeros outside the blocks -- so the owned products run over the
    // row's block only and the rows kept in LDS are `ew` = max (e_hi - e_lo) long.
    const int32_t *e_lo, *e_hi;
    int32_t ew;
};
constexpr int TILE_COARSE_SLOTS = 8;   // partial-sum slots polled per thread, at most
struct TileArgs {
    const TileCoarseDev *coarse;              // null: plain Chebyshev levels
    const double *const *einv;                // per level: (P^T A P)^-1 of the level's matrix
    int32_t cycles;                           // coarse cycles per level (its = sweeps per cycle)
    uint32_t cepoch0;                         // coarse exchanges of this launch carry tags cepoch0 + 1, ...
    int32_t nlevels, its, depth, nk_pad, rpt, W;
    unsigned long long *gnew[2], *gold[2];   // granule buffers, 2 words per row each
    unsigned granule_bytes;
    unsigned *err;                            // word 0: error bits; words 8..: first time-out record
    int32_t stamps;                           // diagnostics: per-tile 100 MHz tick sums at err + 64
    int32_t poll_delay;                       // s_sleep units between publishing and the first poll
    int32_t debug_drop;                       // test hook: tile 0 skips publishing hand-off number
                                              // debug_drop (> 0), so its neighbours time out
    uint32_t epoch0;                          // hand-offs of this launch carry tags epoch0 + 1, ...
    int32_t clear;                            // zero the granule buffers before the launch (the
                                              // first tile launch of a replayed sequence)
    int32_t fused_update;                     // 0: kernel variant without the level update
    int32_t hslots;                           // ring-entry slots per thread the plan needs
};
bool tile_sweep_available(int W, int rpt, int threads, int hslots);   // hslots: ring-entry slots per thread
bool tile_sweep_coarse_available(int W, int rpt, int threads, int hslots);   // variant with coarse corrections
int tile_sweep_max_hslots(int W, int rpt, int threads);
int tile_sweep_max_rpt(int W, int threads);   // most row slots per thread of any variant (0: none)
// whether the variant for this shape computes the level update b -= U u_prev itself (narrow
// rows); wide rows (3-D P1) have no registers for it: the update stays a launch of its own
bool tile_sweep_fuses_update(int W, int max_terms);   // update terms of any level of the run
size_t tile_sweep_lds_bytes(int nk_pad, int its, int coarse_nc = 0, int coarse_nslots = 0,
                            int coarse_jmax = 0, int coarse_nr_max = 0,
                            int coarse_einv_rows = 0,
                            int coarse_einv_width = 0);   // rows of the coarse inverse kept in LDS
// workgroups of `threads` that are certainly co-resident (one per CU)
int tile_sweep_max_tiles(int W, int rpt, int threads, size_t lds_bytes, int hslots,
                         bool coarse = false);
void launch_tile_sweep(hipStream_t s, const TileArgs &a, const TileLevel *d_levels,
                       const int32_t *d_n, const int32_t *d_grow, const uint16_t *d_lcol,
                       const int32_t *d_gpos, const uint8_t *d_rowmask, int ntiles, int threads,
                       size_t granule_words, const TileCoarseDev *h_coarse = nullptr);

// ---- value-array preparation
void launch_csr_to_sell(hipStream_t s, const double *csr_vals, const int32_t *sell2csr,
                        double *sell_vals, int64_t n_padded);
void launch_mask_columns(hipStream_t s, double *sell_vals, const int32_t *col,
                         const uint8_t *colmask, int64_t n_padded);
// out = a + c * b (no fma contraction: matches "assemble(block + c * M)")
void launch_vals_axpy(hipStream_t s, double *out, const double *a, double c,
                      const double *b, int64_t n_padded);
// *flag |= 1 when a and b differ in any bit
void launch_vals_differ(hipStream_t s, const double *a, const double *b, int64_t n, unsigned *flag);
void launch_vals_sym_skew(hipStream_t s, const double *a, const int32_t *tpos, double *h,
                          double *sk, int64_t n, unsigned *nonsym);
// dinv[r] = rowmask[r] ? 1 : 1 / diag(A)[r]
void launch_extract_dinv(hipStream_t s, const int32_t *col, const int32_t *slice_off,
                         const double *vals, const uint8_t *rowmask, double *dinv,
                         int nrows, int nslices, int R, const int32_t *perm = nullptr);

// ---- vector kernels (all lengths in doubles)
void launch_copy(hipStream_t s, double *y, const double *x, int64_t n);
void launch_fill(hipStream_t s, double *y, double v, int64_t n);
// y = a*x + b*y
void launch_axpby(hipStream_t s, double *y, double a, const double *x, double b, int64_t n);
// y[k*nx + r] = mask_k[r] ? (mx ? alpha_k * mx[k*nx + r] : 0) : x[k*nx + r]
struct MaskJob { const uint8_t *mask; double alpha; };
void launch_mask_blocks(hipStream_t s, double *y, const double *x, const double *mx,
                        const MaskJob *d_jobs, int nblocks, int64_t nx);
// T_1 (kind 1) / T_2 (kind 2) from the raw rows `t` into `y` with the operator's Dirichlet
// post-correction fused (d_jobs: the n blocks' masks, xin: the operator's input at the same offset)
void launch_time_transform_mask(hipStream_t s, double *y, const double *t, const double *xin,
                                const MaskJob *d_jobs, int kind, int n, int64_t nx,
                                const double *lo_halo, const double *hi_halo);
// CN time transforms over `n` consecutive blocks of length nx (block stride nx):
// kind 1: T_1 (new_i = old_i + old_{i+1}); 2: T_2; 3: T_1^{-1}; 4: T_2^{-1}
// (preconditioner.py:33-60, control.py:63-96).  `lo_halo`/`hi_halo` (may be null) stand
// for the block before the first / after the last one on a time-sharded handle.
void launch_time_transform(hipStream_t s, double *y, const double *x, int kind, int n,
                           int64_t nx, const double *lo_halo, const double *hi_halo);
// y_k += shift_k for `n` blocks where shift_k = coef * sums[k] (ConstantNullspace)
void launch_block_shift(hipStream_t s, double *y, const double *sums, double coef, int n,
                        int64_t nx);
// ConstantNullspace on every block that carries one, in two launches:
// y_j -= mean(y_j); second == 1: y_j += mean(b_j); second == 2: y_j += alpha_j * mean(b_j)
struct ConstJob { int64_t off, nx; double c1, c2_one, c2_alpha; };
void launch_const_correct(hipStream_t s, const ConstJob *d_jobs, int njobs, int64_t max_nx,
                          double *y, const double *b, int second, double *sums);
void launch_const_center(hipStream_t s, const ConstJob *d_jobs, int njobs, int64_t max_nx,
                         const double *x, double *xc, double *sums);
void launch_block_sums(hipStream_t s, const double *x, double *sums, int n, int64_t nx,
                       double *scratch);

// ---- reductions (two fixed stages, no atomics: bitwise reproducible)
struct VecList { const double *v[MDOT_MAX]; };
// out[i] = <w, V_i>, i < nv <= MDOT_MAX.  scratch: REDUCE_BLOCKS * MDOT_MAX doubles.
void launch_mdot(hipStream_t s, const double *w, VecList V, int nv, int64_t n,
                 double *scratch, double *out);
// out[0] = sqrt(<w, w> + extra) where extra = (add ? *add : 0)
void launch_norm2_finish(hipStream_t s, const double *dot, double *out);
void launch_zero_bytes(hipStream_t s, void *p, size_t nbytes);   // kernel node, not a memset node
void launch_add_constant(hipStream_t s, double *y, double v, int64_t n);   // y[i] += v

// ---- coarse space of the two-grid sub-solves on the device: P by rows (prolongation) and by
// columns (restriction), scratch for one correction
struct CoarseDev {
    int nc = 0;
    const int32_t *p_ip = nullptr, *p_ix = nullptr;     // P: n rows
    const double *p_v = nullptr;
    const int32_t *pt_ip = nullptr, *pt_ix = nullptr;   // P^T: nc rows
    const double *pt_v = nullptr;
    double *rc = nullptr, *ec = nullptr;                // nc doubles each
};
// x_out = x_in + P E^-1 P^T r (x_in may be null or x_out); three launches, fixed summation orders
void launch_coarse_correction(hipStream_t s, const CoarseDev &c, const double *einv,
                              const double *r, const double *x_in, double *x_out, int64_t n);
// the same for nb vectors `vstride` apart in one launch per stage (c.rc, c.ec: nb * nc doubles)
void launch_coarse_correction_batched(hipStream_t s, const CoarseDev &c, const double *einv,
                                      const double *r, const double *x_in, double *x_out, int64_t n,
                                      int nb, int64_t vstride);
void launch_einv_outside(hipStream_t s, const double *einv, int nc, const int32_t *lo,
                         const int32_t *hi, unsigned *flag);
void launch_coarse_column(hipStream_t s, const CoarseDev &c, int k, double *x, int64_t n);
void launch_coarse_restrict(hipStream_t s, const CoarseDev &c, const double *r, double *rc,
                            int stride = 1);
// inv = a^-1 (n x n row-major, a destroyed) by Gauss-Jordan with partial pivoting on the device
void launch_dense_inverse(hipStream_t s, double *a, double *inv, int n, int *d_piv,
                          double *d_colbuf, unsigned *d_flag);
void launch_flag_to_double(hipStream_t s, const unsigned *flag, double *out);
// w += sign * sum_i coef[i] * V_i   (coef in device memory)
void launch_maxpy(hipStream_t s, double *w, VecList V, const double *coef, double sign,
                  int nv, int64_t n);
// the same update, and out_sq[0] = <w, w> of the updated w (summation order of launch_mdot)
void launch_maxpy_norm(hipStream_t s, double *w, VecList V, const double *coef, double sign,
                       int nv, int64_t n, double *scratch, double *out_sq);
// y = x * (1 / *norm)
void launch_scale_inv(hipStream_t s, double *y, const double *x, const double *norm,
                      int64_t n);

}  // namespace kkt
