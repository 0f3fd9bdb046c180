// Host-side objects of libkkt: the block system, its HBM layout, the apply plan.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include <map>
#include <memory>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/kkt.h"
#include "kernels.hpp"

namespace kkt {

struct Error {
    int code;
    std::string msg;
};
[[noreturn]] void fail(int code, const std::string &msg);
void hip_check(hipError_t e, const char *what, const char *file, int line);
#define HIPCHK(x) ::kkt::hip_check((x), #x, __FILE__, __LINE__)

template <class T>
T *dev_alloc(size_t n) {
    void *p = nullptr;
    if (n == 0) n = 1;
    HIPCHK(hipMalloc(&p, n * sizeof(T)));
    return static_cast<T *>(p);
}
template <class T>
T *dev_upload(const T *h, size_t n) {
    T *p = dev_alloc<T>(n);
    if (n) HIPCHK(hipMemcpy(p, h, n * sizeof(T), hipMemcpyHostToDevice));
    return p;
}

// One sparsity structure in SELL-64R layout, shared by every block that has it.
struct Pattern {
    int64_t nrows = 0, ncols = 0, nnz = 0;
    int R = 2;
    int nslices = 0;
    int64_t nslots = 0;     // sum of slice widths
    int64_t npadded = 0;    // nslots * 64 * R
    int max_width = 0;
    int uniform_w = -1;     // >= 0 when every slice has this width
    uint64_t hash = 0;
    std::vector<int32_t> h_indptr, h_indices;   // kept to prove equality, not just hash
    int32_t *d_col = nullptr;
    int32_t *d_slice_off = nullptr;
    std::vector<int32_t> h_slice_off;   // nslices + 1 offsets in slots (host copy)
    // index into a SELL value array of entry k (CSR order) of row r
    int64_t sell_index(int64_t r, int k) const {
        const int64_t C = 64 * R, pos = pos_of(r), s = pos / C, rin = pos - s * C;
        return ((int64_t)h_slice_off[s] + k) * C + (rin % 64) * R + rin / 64;
    }
    int32_t *d_sell2csr = nullptr;
    // row-sorted storage (SELL-C-sigma): row held by each position / position of each row;
    // empty and null when position == row
    std::vector<int32_t> h_row_of, h_pos_of;
    int32_t *d_perm = nullptr;
    int64_t row_of(int64_t pos) const { return h_row_of.empty() ? (pos < nrows ? pos : -1) : h_row_of[pos]; }
    int64_t pos_of(int64_t r) const { return h_pos_of.empty() ? r : h_pos_of[r]; }
};

struct ValueArray {
    int pattern = -1;
    double *d_vals = nullptr;
    int64_t share_id = -1;
    int colmask_set = -2;   // -2: not decided yet; -1: none; >= 0: id of the bc set applied
};

struct Block {
    int q, i, j;
    int va;
    int order;   // insertion order inside its quadrant (the reference's dict order)
};

struct NullspaceSpec {
    int kind = 0;   // 0 none, 1 Dirichlet, 2 constant
    double alpha = 1.0;
    int set_id = -1;   // Dirichlet: id into System::bc_sets
};

struct BcSet {
    std::vector<int32_t> idx;
    int64_t nx = 0;
    uint8_t *d_mask = nullptr;   // nx bytes
    int32_t *d_idx = nullptr;
};

// One launch of the fused row kernel.
struct RowLaunch {
    RowOp *d_ops = nullptr;
    int nops = 0;
    int max_slices = 0;
    int R = 2;
    int uniform_w = -1;   // > 0: every op with terms in this launch has this slice width
    bool single = false;  // nops == 1: h_op rides in the kernel arguments
    bool shared_matrix = false;   // every op has one term with the same matrix
    int32_t *d_groups = nullptr;  // operator apply, shared values: runs of ops of one structure
    int ngroups = 0;
    RowOp h_op;
};

struct TimeGroup {   // CN transform applied to a contiguous range of local blocks
    int first_local_block;   // flat local block index
    int n;
    int kind;                // 1: T_1, 2: T_2
    int64_t nx;
    double *d_halo = nullptr;   // time-sharded: the neighbour rank's block the transform reads
                                // (T_1: level hi from the rank above, T_2: level lo - 1 from below)
};

class PcBase;
class Comm;

struct KrylovCfg {
    int type = KKT_KSP_FGMRES;
    int pc_side = KKT_PC_SIDE_DEFAULT;
    int restart = 30;
    double rtol = 1e-6, atol = 0.0, divtol = 1e4;
    int max_it = 1000;
};

// HIP events between the stages of a Krylov iteration (option "stage_timers"; kkt_get_stage_times)
struct StageClock {
    enum { OP = 0, PC = 1, ORTH = 2, ALLREDUCE = 3, OTHER = 4, NSTAGES = 5 };
    bool on = false;
    std::vector<hipEvent_t> pool;
    std::vector<int> stage_of;     // stage of the interval that ends at event k (k >= 1)
    size_t used = 0;
    void begin(hipStream_t s);
    void mark(hipStream_t s, int stage);
    void finish(kkt_stage_times &out);   // synchronises on the last event
    ~StageClock();
};

struct System {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;

    // layout (global)
    bool layout_set = false, finalized = false;
    int n0 = 1, n1 = 1;
    int64_t nx0 = 0, nx1 = 0;
    bool CN = false;
    int sub00 = -1, sub11 = -1;
    // time shard
    int rank = 0, world = 1;
    int lo = 0, hi = 0;            // owned time levels of both variables (sharded only)
    bool sharded = false;
    // Block families: the flat blocks of a variable are `families` runs of mf = n / families time
    // levels (1: heat-type systems; 2: the outer incompressible system, velocity blocks (v, zeta)
    // and pressure blocks (mu, p), control.py:3654-3673).  A rank owns levels [lo, hi) of every
    // family; local flat block f * (hi - lo) + (level - lo).
    int families = 1, mf = 0;
    int level_of(int k) const { return sharded ? k % mf : k; }
    int family_of(int k) const { return sharded ? k / mf : 0; }
    bool owns(int k) const { return !sharded || (level_of(k) >= lo && level_of(k) < hi); }
    int local_of(int k) const { return sharded ? family_of(k) * (hi - lo) + level_of(k) - lo : k; }
    // halo blocks of a 2-family shard: [variable][family][0: level lo-1, 1: level hi]
    double *d_halo2[2][2][2] = {{{nullptr, nullptr}, {nullptr, nullptr}},
                                {{nullptr, nullptr}, {nullptr, nullptr}}};
    bool halo2_used[2][2][2] = {{{false, false}, {false, false}}, {{false, false}, {false, false}}};
    bool halo2_agreed = false;     // usage flags made consistent over the ranks (first apply)
    int n0_loc = 1, n1_loc = 1;    // local block counts
    int64_t n_local = 0;           // local vector length
    int64_t vec_stride = 0;        // padded allocation length of internal vectors

    int sell_R = 2;

    std::vector<Pattern> patterns;
    std::vector<ValueArray> values;
    std::map<std::tuple<int, int, int>, Block> blocks;
    std::map<int64_t, int> share_map;
    int order_counter[4] = {0, 0, 0, 0};
    std::vector<NullspaceSpec> nullspaces;   // per GLOBAL flat block
    std::vector<BcSet> bc_sets;

    // apply plan
    std::vector<RowLaunch> apply_launches;
    int first_halo_launch = 0;    // launches [first_halo_launch, end) hold the rows that read a halo
    hipStream_t comm_stream = nullptr;              // halo exchange, overlapped with interior rows
    hipEvent_t ev_x_ready = nullptr, ev_halo_ready = nullptr;
    std::vector<std::vector<RowOp>> h_apply_ops;   // host copies (value pointers are re-pointed
                                                   // when an update un-shares a value array)
    std::map<std::tuple<int, int, int>, std::tuple<int, int, int>> block_term;   // -> launch, op, term
    std::vector<TimeGroup> time_groups;
    bool fused_row_masks = false;     // Dirichlet epilogue fused into the row kernel
    bool any_const_ns = false;
    MaskJob *d_mask_jobs = nullptr;   // per local block (post-correction for CN)
    MaskJob *d_mask_jobs_one = nullptr;   // same masks with alpha = 1 (preconditioner side)
    double *d_pc_in = nullptr, *d_pc_out = nullptr;   // callback / identity preconditioner
    double *d_rhs = nullptr;          // corrected right-hand side of a solve
    double *d_guess = nullptr;        // the caller's initial guess (a solve may start over)
    double *d_xc = nullptr;           // ConstantNullspace-corrected copy of x
    double *d_tmp_y = nullptr;        // raw rows before the CN transform
    double *d_sums = nullptr;
    ConstJob *d_const_jobs = nullptr;   // ConstantNullspace blocks of this handle
    int n_const_jobs = 0;
    int64_t const_max_nx = 0;
    // halos (time-sharded): x0 block lo-1, x1 block hi; CN raw rows rho0_hi, rho1_{lo-1}
    double *d_halo_x0_lo = nullptr, *d_halo_x1_hi = nullptr;

    // byte accounting
    kkt_info info{};

    // preconditioner
    std::unique_ptr<PcBase> pc;
    bool pc_stale = false;   // block values changed since the preconditioner was built
    kkt_pc_callback pc_cb = nullptr;
    void *pc_cb_user = nullptr;
    bool pc_cb_failed = false;

    std::unique_ptr<Comm> comm;

    KrylovCfg ksp;
    kkt_steplock steplock{};   // test hook (kkt_debug_set_steplock); n_steps == 0: off
    // execution options (kkt_set_option); a key that was never set has its built-in default
    // (the library does not read the environment)
    std::map<std::string, std::string> options;
    StageClock clock;
    kkt_stage_times stage_times{};
    std::vector<double> tile_coords;   // kkt_set_tile_coordinates: N_x x tile_dim, or empty
    int tile_dim = 0;
    const char *opt(const char *key) const;
    // Krylov workspace (lazily sized)
    int ws_restart = 0;
    bool ws_flexible = false;
    double *d_V = nullptr, *d_Z = nullptr, *d_w = nullptr, *d_t1 = nullptr, *d_t2 = nullptr;
    double *d_red_scratch = nullptr, *d_hcol = nullptr, *d_coef = nullptr;
    double *h_pinned = nullptr;

    System() = default;
    ~System();

    // -- definition
    void set_layout(int n_blocks_00, int n_blocks_11, int64_t nx0_, int64_t nx1_, int CN_,
                    int s00, int s11);
    void set_shard(int rank_, int world_, int families_ = 1);
    void add_block(int q, int i, int j, int64_t nrows, int64_t ncols, const int32_t *indptr,
                   const int32_t *indices, const double *vals, int64_t share_id);
    void update_block_values(int q, int i, int j, const double *vals);
    void set_bc(int k, int64_t n, const int32_t *idx, double alpha);
    void set_const_ns(int k, double alpha);
    void finalize();

    // -- helpers
    int find_or_add_pattern(int64_t nrows, int64_t ncols, const int32_t *indptr,
                            const int32_t *indices);
    int add_bc_set(int64_t nx, int64_t n, const int32_t *idx);
    int new_value_array(int pattern, const double *csr_vals);
    int64_t block_nx(int flat_global_k) const { return flat_global_k < n0 ? nx0 : nx1; }
    // local flat block index -> offset in the local vector
    int64_t local_offset(int var, int local_i) const {
        return var == 0 ? (int64_t)local_i * nx0 : (int64_t)n0_loc * nx0 + (int64_t)local_i * nx1;
    }
    int global_row(int var, int local_i) const {
        (void)var;
        if (!sharded) return local_i;
        const int nl = hi - lo;
        return (local_i / nl) * mf + lo + local_i % nl;
    }
    double *new_vec();   // internal vector of vec_stride doubles (zeroed)

    // -- operations on device vectors of n_local doubles
    void apply(const double *d_x, double *d_y);
    void pc_apply(const double *d_x, double *d_y);
    void pc_apply_timed(const double *d_x, double *d_y, float *ms, int *launches, int64_t *phases);
    void solve(const double *d_b, double *d_u, int *its, int *reason, double *rnorm,
               double *hist, int hist_cap, int *hist_len);
    void solve_once(const double *d_b, double *d_u, int *its, int *reason, double *rnorm,
                    double *hist, int hist_cap, int *hist_len);
    int program_fallbacks = 0;   // sweep programs replaced by plain launches after a time-out
    // a sweep program of the preconditioner timed out on ANY rank (collective on time shards)
    bool pc_timed_out_agreed(std::string *why);
    // ... then every rank rebuilds its preconditioner as plain launches (collective); false if
    // there is nothing to fall back to
    bool pc_fallback_plain(const std::string &why);
    void pc_apply_timed_stages(const double *d_x, double *d_y, kkt_pc_stage_times *out);
    void solve_minres(const double *d_b, double *d_u, int *its, int *reason, double *rnorm,
                      double *hist, int hist_cap, int *hist_len);
    void ensure_workspace(int restart, bool flexible);
    // reductions over the whole (distributed) vector; results in device memory
    void mdot(const double *w, const double *const *V, int nv, double *d_out);
    void norm2(const double *w, double *d_out);   // d_out[0] = ||w||_2, d_out[1] scratch
    // y = P x: lhs_right / lhs_left of every block's nullspace (preconditioner.py:92-106)
    void ns_project(double *y, const double *x);
    // y = P u + (I - P) b on top of u stored in `u` (preconditioner.py:114-116)
    void ns_pc_post(double *y, const double *u, const double *b);
    void sync() { HIPCHK(hipStreamSynchronize(stream)); }
};

}  // namespace kkt
