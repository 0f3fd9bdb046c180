// Block system: definition, HBM layout (SELL-64R with shared index arrays), KKT apply.
// Restates MultiBlockSystem.__init__ / MultiBlockSystemMatrix.mult
// (reference preconditioner/preconditioner.py:216-335, 375-543) for the GPU.
#include "system.hpp"

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "comm.hpp"
#include "pc.hpp"

namespace kkt {

void fail(int code, const std::string &msg) { throw Error{code, msg}; }

void hip_check(hipError_t e, const char *what, const char *file, int line) {
    if (e != hipSuccess) {
        fail(KKT_ERR_HIP, std::string(hipGetErrorString(e)) + " in " + what + " at " + file +
                              ":" + std::to_string(line));
    }
}

const char *System::opt(const char *key) const {
    auto it = options.find(key);
    return it != options.end() ? it->second.c_str() : nullptr;
}

// ---- stage clock: one event per mark; the interval since the previous mark belongs to `stage`
void StageClock::begin(hipStream_t s) {
    used = 0;
    stage_of.clear();
    if (!on) return;
    mark(s, OTHER);
}
void StageClock::mark(hipStream_t s, int stage) {
    if (!on) return;
    if (used == pool.size()) {
        hipEvent_t e;
        HIPCHK(hipEventCreate(&e));
        pool.push_back(e);
    }
    HIPCHK(hipEventRecord(pool[used], s));
    stage_of.push_back(stage);
    ++used;
}
void StageClock::finish(kkt_stage_times &out) {
    out = kkt_stage_times{};
    if (!on || used < 2) return;
    HIPCHK(hipEventSynchronize(pool[used - 1]));
    double acc[NSTAGES] = {0, 0, 0, 0, 0};
    for (size_t k = 1; k < used; ++k) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, pool[k - 1], pool[k]));
        acc[stage_of[k]] += ms;
    }
    out.operator_ms = acc[OP];
    out.pc_ms = acc[PC];
    out.orth_ms = acc[ORTH];
    out.allreduce_ms = acc[ALLREDUCE];
    out.other_ms = acc[OTHER];
    out.total_ms = acc[OP] + acc[PC] + acc[ORTH] + acc[ALLREDUCE] + acc[OTHER];
}
StageClock::~StageClock() {
    for (hipEvent_t e : pool) (void)hipEventDestroy(e);
}

System::~System() {
    // device memory is released with the process or by hipDeviceReset; explicit frees keep
    // long-lived hosts (Picard loops creating many systems) from accumulating HBM
    (void)hipSetDevice(device);
    pc.reset();
    comm.reset();
    auto F = [](void *p) {
        if (p) (void)hipFree(p);
    };
    for (auto &p : patterns) {
        F(p.d_col);
        F(p.d_perm);
        F(p.d_slice_off);
        F(p.d_sell2csr);
    }
    for (auto &v : values) F(v.d_vals);
    for (auto &b : bc_sets) {
        F(b.d_mask);
        F(b.d_idx);
    }
    for (auto &l : apply_launches) {
        F(l.d_ops);
        F(l.d_groups);
    }
    F(d_mask_jobs);
    F(d_mask_jobs_one);
    F(d_pc_in);
    F(d_pc_out);
    F(d_rhs);
    F(d_guess);
    F(d_xc);
    F(d_tmp_y);
    F(d_sums);
    F(d_const_jobs);
    F(d_halo_x0_lo);
    F(d_halo_x1_hi);
    for (TimeGroup &g : time_groups) F(g.d_halo);
    for (auto &a : d_halo2)
        for (auto &b : a)
            for (double *q : b) F(q);
    F(d_V);
    F(d_Z);
    F(d_w);
    F(d_t1);
    F(d_t2);
    F(d_red_scratch);
    F(d_hcol);
    F(d_coef);
    if (h_pinned) (void)hipHostFree(h_pinned);
    if (ev_x_ready) (void)hipEventDestroy(ev_x_ready);
    if (ev_halo_ready) (void)hipEventDestroy(ev_halo_ready);
    if (comm_stream) (void)hipStreamDestroy(comm_stream);
    if (stream) (void)hipStreamDestroy(stream);
}

void System::set_layout(int n_blocks_00, int n_blocks_11, int64_t nx0_, int64_t nx1_, int CN_,
                        int s00, int s11) {
    if (layout_set) fail(KKT_ERR_STATE, "layout already set");
    if (n_blocks_00 < 1 || n_blocks_11 < 1 || nx0_ < 1 || nx1_ < 1)
        fail(KKT_ERR_ARG, "block counts and sizes must be positive");
    if (nx0_ >= (int64_t)1 << 31 || nx1_ >= (int64_t)1 << 31)
        fail(KKT_ERR_ARG, "spatial block size exceeds int32 indexing");
    if ((s00 >= 0) != (s11 >= 0))
        fail(KKT_ERR_ARG, "sub_n_blocks_00_0 and sub_n_blocks_11_0 must be given together");
    if (s00 > n_blocks_00 || s11 > n_blocks_11) fail(KKT_ERR_ARG, "sub_n_blocks out of range");
    n0 = n_blocks_00;
    n1 = n_blocks_11;
    nx0 = nx0_;
    nx1 = nx1_;
    CN = CN_ != 0;
    sub00 = s00;
    sub11 = s11;
    n0_loc = n0;
    n1_loc = n1;
    nullspaces.assign(n0 + n1, NullspaceSpec{});
    layout_set = true;
    const char *e = opt("sell_r");
    if (e && (e[0] == '1' || e[0] == '2')) sell_R = e[0] - '0';
}

void System::set_shard(int rank_, int world_, int families_) {
    if (!layout_set || finalized || !blocks.empty())
        fail(KKT_ERR_STATE, "kkt_set_shard must follow kkt_set_layout and precede blocks");
    if (world_ < 1 || rank_ < 0 || rank_ >= world_) fail(KKT_ERR_ARG, "bad rank/world");
    if (families_ != 1 && families_ != 2) fail(KKT_ERR_ARG, "1 or 2 block families");
    if (world_ == 1) return;
    if (n0 != n1) fail(KKT_ERR_ARG, "time sharding needs n_blocks_00 == n_blocks_11");
    if (n0 % families_ != 0) fail(KKT_ERR_ARG, "block count is not a multiple of the families");
    // a sub-block split (the incompressible outer system: other time transforms on the second
    // half of each variable's blocks, preconditioner.py:471-525) shards when the split is the
    // family boundary
    if (sub00 >= 0 && !(families_ == 2 && sub00 == n0 / 2 && sub11 == n1 / 2))
        fail(KKT_ERR_ARG, "time sharding with sub-block splits needs two families split at the "
                          "sub-block boundary");
    families = families_;
    mf = n0 / families;
    if (world_ > mf) fail(KKT_ERR_ARG, "more ranks than time levels");
    rank = rank_;
    world = world_;
    kkt_shard_range(mf, rank, world, &lo, &hi);
    sharded = true;
    n0_loc = n1_loc = families * (hi - lo);
}

// Fingerprint of an index array (a filter in front of the memcmp in find_or_add_pattern).
// FNV-1a over 8-byte words in four independent lanes: a byte-serial FNV spent 2.5 ms per
// 2 MB block on its multiply chain -- 0.9 s of the 1.06 s cfg 2 took to set up.
static uint64_t fnv1a(const void *data, size_t bytes, uint64_t h) {
    constexpr uint64_t prime = 1099511628211ull;
    const unsigned char *p = static_cast<const unsigned char *>(data);
    uint64_t lane[4] = {h, h ^ 0x9e3779b97f4a7c15ull, h ^ 0xc2b2ae3d27d4eb4full,
                        h ^ 0x165667b19e3779f9ull};
    size_t i = 0;
    for (; i + 32 <= bytes; i += 32) {
        uint64_t w[4];
        std::memcpy(w, p + i, 32);
        for (int k = 0; k < 4; ++k) lane[k] = (lane[k] ^ w[k]) * prime;
    }
    h = lane[0];
    for (int k = 1; k < 4; ++k) h = (h ^ (lane[k] >> 29) ^ lane[k]) * prime;
    for (; i < bytes; ++i) {
        h ^= p[i];
        h *= prime;
    }
    return h;
}

int System::find_or_add_pattern(int64_t nrows, int64_t ncols, const int32_t *indptr,
                                const int32_t *indices) {
    const int64_t nnz = indptr[nrows];
    uint64_t h = 1469598103934665603ull;
    h = fnv1a(&nrows, sizeof nrows, h);
    h = fnv1a(&ncols, sizeof ncols, h);
    h = fnv1a(indptr, (nrows + 1) * sizeof(int32_t), h);
    h = fnv1a(indices, nnz * sizeof(int32_t), h);
    for (size_t p = 0; p < patterns.size(); ++p) {
        const Pattern &P = patterns[p];
        if (P.hash == h && P.nrows == nrows && P.ncols == ncols && P.nnz == nnz &&
            std::memcmp(P.h_indptr.data(), indptr, (nrows + 1) * sizeof(int32_t)) == 0 &&
            std::memcmp(P.h_indices.data(), indices, nnz * sizeof(int32_t)) == 0)
            return (int)p;
    }
    Pattern P;
    P.nrows = nrows;
    P.ncols = ncols;
    P.nnz = nnz;
    P.hash = h;
    P.R = sell_R;
    P.h_indptr.assign(indptr, indptr + nrows + 1);
    P.h_indices.assign(indices, indices + nnz);
    const int C = 64 * P.R;
    P.nslices = (int)((nrows + C - 1) / C);
    for (int64_t r = 0; r < nrows; ++r)
        if (indptr[r + 1] < indptr[r]) fail(KKT_ERR_ARG, "indptr not monotone");
    std::vector<int32_t> off(P.nslices + 1, 0);
    // widths of the slices when position p holds row row_of(p)
    auto measure = [&](const std::vector<int32_t> *row_of) {
        P.max_width = 0;
        P.uniform_w = -1;
        for (int s = 0; s < P.nslices; ++s) {
            int w = 0;
            for (int64_t p = (int64_t)s * C; p < (int64_t)(s + 1) * C; ++p) {
                const int64_t r = row_of ? (*row_of)[p] : (p < nrows ? p : -1);
                if (r >= 0) w = std::max(w, indptr[r + 1] - indptr[r]);
            }
            P.max_width = std::max(P.max_width, w);
            if (s == 0)
                P.uniform_w = w;
            else if (P.uniform_w != w)
                P.uniform_w = -1;
            off[s + 1] = off[s] + w;
        }
    };
    measure(nullptr);
    // Nearly uniform structures (structured meshes: only the slices that hold boundary
    // rows are narrower) are padded to one width so that the fixed-width kernels apply;
    // accepted when it costs at most 3 % more slots.
    if (P.uniform_w < 0 && P.max_width >= 1 && P.max_width <= 16) {
        const int64_t uni = (int64_t)P.max_width * P.nslices;
        if ((uni - off[P.nslices]) * 100 <= 3 * (int64_t)off[P.nslices]) {
            for (int s = 0; s <= P.nslices; ++s) off[s] = s * P.max_width;
            P.uniform_w = P.max_width;
        }
    }
    // Rows of very different lengths inside a slice (P2 / Q2 spaces, unstructured meshes) waste
    // slots: sort the rows of every window of 8 slices by length (SELL-C-sigma) when that
    // saves at least 10 % of the storage.  Vectors keep their order; kernels reach the row
    // of a position through `perm`.
    const char *sort_env = opt("sell_sort");   // read per pattern: tests toggle it
    const bool allow_sort = !(sort_env && sort_env[0] == '0');
    if (allow_sort && nnz > 0) {
        // (window: 8 slices = the rows of two workgroups, unless "sell_sigma" says otherwise)
        const char *sg = opt("sell_sigma");
        const int sig_slices = sg ? std::max(1, std::min(64, std::atoi(sg))) : 8;
        const int64_t npos = (int64_t)P.nslices * C, sigma = sig_slices * (int64_t)C;
        std::vector<int32_t> cand(npos, -1);
        for (int64_t w0 = 0; w0 < nrows; w0 += sigma) {
            const int64_t w1 = std::min<int64_t>(nrows, w0 + sigma);
            std::vector<int32_t> rows((size_t)(w1 - w0));
            for (int64_t r = w0; r < w1; ++r) rows[r - w0] = (int32_t)r;
            std::stable_sort(rows.begin(), rows.end(), [&](int32_t a, int32_t b) {
                return indptr[a + 1] - indptr[a] > indptr[b + 1] - indptr[b];
            });
            for (int64_t r = w0; r < w1; ++r) cand[r] = rows[r - w0];
        }
        const int64_t before = off[P.nslices];
        const int uni_before = P.uniform_w, maxw_before = P.max_width;
        std::vector<int32_t> off_before = off;
        measure(&cand);
        if ((int64_t)off[P.nslices] * 10 <= before * 9) {
            P.uniform_w = -1;
            P.h_row_of = cand;
            P.h_pos_of.assign(nrows, -1);
            for (int64_t p = 0; p < npos; ++p)
                if (cand[p] >= 0) P.h_pos_of[cand[p]] = (int32_t)p;
            P.d_perm = dev_upload(cand.data(), cand.size());
        } else {
            off = off_before;
            P.uniform_w = uni_before;
            P.max_width = maxw_before;
        }
    }
    P.nslots = off[P.nslices];
    P.npadded = P.nslots * C;
    if (P.npadded >= (int64_t)1 << 31) fail(KKT_ERR_ARG, "block too large for int32 maps");
    std::vector<int32_t> col(P.npadded), map(P.npadded, -1);
    for (int s = 0; s < P.nslices; ++s) {
        const int w = off[s + 1] - off[s];
        for (int within = 0; within < C; ++within) {
            const int64_t pos = (int64_t)s * C + (within % P.R) * 64 + within / P.R;   // lane-major
            const int64_t r = P.row_of(pos);
            const int32_t self = (int32_t)std::min<int64_t>(r >= 0 ? r : nrows - 1, ncols - 1);
            const int len = r >= 0 ? indptr[r + 1] - indptr[r] : 0;
            for (int k = 0; k < w; ++k) {
                const int64_t p = ((int64_t)off[s] + k) * C + within;
                if (k < len) {
                    const int32_t c = indices[indptr[r] + k];
                    if (c < 0 || c >= ncols) fail(KKT_ERR_ARG, "column index out of range");
                    if (k > 0 && c <= indices[indptr[r] + k - 1])
                        fail(KKT_ERR_ARG, "column indices must be sorted and unique in a row");
                    col[p] = c;
                    map[p] = indptr[r] + k;
                } else {
                    col[p] = self;   // padding: value 0, gather stays local
                }
            }
        }
    }
    P.d_col = dev_upload(col.data(), col.size());
    P.d_slice_off = dev_upload(off.data(), off.size());
    P.h_slice_off = off;
    P.d_sell2csr = dev_upload(map.data(), map.size());
    patterns.push_back(std::move(P));
    info.bytes_device_index += patterns.back().npadded * 4 + (patterns.back().nslices + 1) * 4 +
                               (patterns.back().d_perm ? (int64_t)patterns.back().nslices * C * 4 : 0);
    return (int)patterns.size() - 1;
}

int System::new_value_array(int pattern, const double *csr_vals) {
    const Pattern &P = patterns[pattern];
    ValueArray va;
    va.pattern = pattern;
    va.d_vals = dev_alloc<double>(P.npadded);
    double *d_csr = dev_upload(csr_vals, (size_t)P.nnz);
    launch_csr_to_sell(stream, d_csr, P.d_sell2csr, va.d_vals, P.npadded);
    HIPCHK(hipStreamSynchronize(stream));
    HIPCHK(hipFree(d_csr));
    values.push_back(va);
    info.bytes_device_values += P.npadded * 8;
    return (int)values.size() - 1;
}

void System::add_block(int q, int i, int j, int64_t nrows, int64_t ncols,
                       const int32_t *indptr, const int32_t *indices, const double *vals,
                       int64_t share_id) {
    if (!layout_set || finalized) fail(KKT_ERR_STATE, "kkt_add_block: wrong state");
    if (q < 0 || q > 3 || !indptr || !indices || !vals) fail(KKT_ERR_ARG, "bad block args");
    const int nr = (q == KKT_Q00 || q == KKT_Q01) ? n0 : n1;
    const int nc = (q == KKT_Q00 || q == KKT_Q10) ? n0 : n1;
    const int64_t er = (q == KKT_Q00 || q == KKT_Q01) ? nx0 : nx1;
    const int64_t ec = (q == KKT_Q00 || q == KKT_Q10) ? nx0 : nx1;
    if (i < 0 || i >= nr || j < 0 || j >= nc) fail(KKT_ERR_ARG, "block index out of range");
    if (nrows != er || ncols != ec) fail(KKT_ERR_ARG, "block shape does not match the layout");
    if (sharded) {
        if (!owns(i)) fail(KKT_ERR_ARG, "block row not owned by this rank");
        const int lj = level_of(j);
        if (lj < lo - 1 || lj > hi) fail(KKT_ERR_ARG, "time sharding needs |i - j| <= 1 blocks");
        const bool col0 = q == KKT_Q00 || q == KKT_Q10;
        if (families == 1) {
            // one halo per column variable is exchanged: x0 of block lo-1 and x1 of block hi
            // (comm_exchange_x_halos) -- the couplings of the BE / CN stencils, control.py:2909-2953
            if (col0 && j == hi)
                fail(KKT_ERR_ARG, "time sharding: a block of column variable 0 may reach block lo-1, not hi");
            if (!col0 && j == lo - 1)
                fail(KKT_ERR_ARG, "time sharding: a block of column variable 1 may reach block hi, not lo-1");
        } else {
            // two families: whatever the stencil couples -- recorded per (variable, family, side)
            const int li = level_of(i);
            if (lj == li - 1) halo2_used[col0 ? 0 : 1][family_of(j)][0] = true;
            if (lj == li + 1) halo2_used[col0 ? 0 : 1][family_of(j)][1] = true;
        }
    }
    auto key = std::make_tuple(q, i, j);
    if (blocks.count(key)) fail(KKT_ERR_ARG, "block added twice");
    int va = -1;
    if (share_id >= 0) {
        auto it = share_map.find(share_id);
        if (it != share_map.end()) {
            va = it->second;
            const Pattern &P = patterns[values[va].pattern];
            if (P.nrows != nrows || P.ncols != ncols || P.nnz != indptr[nrows])
                fail(KKT_ERR_ARG, "share_id reused for a block of different structure");
        }
    }
    if (va < 0) {
        const int p = find_or_add_pattern(nrows, ncols, indptr, indices);
        va = new_value_array(p, vals);
        values[va].share_id = share_id;
        if (share_id >= 0) share_map[share_id] = va;
        info.n_value_arrays++;
        info.bytes_algorithmic += 12 * patterns[p].nnz + 4 * (nrows + 1);
    }
    blocks[key] = Block{q, i, j, va, order_counter[q]++};
    info.n_blocks_stored++;
    info.nnz_blocks += indptr[nrows];
    info.rows_blocks += nrows;
}

int System::add_bc_set(int64_t nx, int64_t n, const int32_t *idx) {
    std::vector<int32_t> v(idx, idx + n);
    std::sort(v.begin(), v.end());
    v.erase(std::unique(v.begin(), v.end()), v.end());
    for (int32_t k : v)
        if (k < 0 || k >= nx) fail(KKT_ERR_ARG, "Dirichlet dof index out of range");
    for (size_t s = 0; s < bc_sets.size(); ++s)
        if (bc_sets[s].nx == nx && bc_sets[s].idx == v) return (int)s;
    BcSet b;
    b.nx = nx;
    b.idx = v;
    std::vector<uint8_t> m(nx, 0);
    for (int32_t k : v) m[k] = 1;
    b.d_mask = dev_upload(m.data(), m.size());
    b.d_idx = dev_upload(v.data(), v.size());
    bc_sets.push_back(std::move(b));
    return (int)bc_sets.size() - 1;
}

void System::set_bc(int k, int64_t n, const int32_t *idx, double alpha) {
    if (!layout_set || finalized) fail(KKT_ERR_STATE, "kkt_set_bc: wrong state");
    if (k < 0 || k >= n0 + n1 || n < 0 || (n > 0 && !idx)) fail(KKT_ERR_ARG, "bad bc args");
    NullspaceSpec ns;
    ns.kind = 1;
    ns.alpha = alpha;
    ns.set_id = add_bc_set(block_nx(k), n, idx);
    nullspaces[k] = ns;
}

void System::set_const_ns(int k, double alpha) {
    if (!layout_set || finalized) fail(KKT_ERR_STATE, "kkt_set_const_nullspace: wrong state");
    if (k < 0 || k >= n0 + n1) fail(KKT_ERR_ARG, "bad block index");
    NullspaceSpec ns;
    ns.kind = 2;
    ns.alpha = alpha;
    nullspaces[k] = ns;
}

double *System::new_vec() {
    double *p = dev_alloc<double>(vec_stride);
    HIPCHK(hipMemsetAsync(p, 0, vec_stride * sizeof(double), stream));
    return p;
}

static VRef vref(int base, int64_t off) { return VRef{off, base, 0}; }
static VRef vnull() { return VRef{0, -1, 0}; }

void System::finalize() {
    if (!layout_set || finalized) fail(KKT_ERR_STATE, "kkt_finalize: wrong state");
    n_local = (int64_t)n0_loc * nx0 + (int64_t)n1_loc * nx1;
    vec_stride = (n_local + 31) & ~(int64_t)31;
    info.n_local = n_local;
    info.n_patterns = (int64_t)patterns.size();
    info.bytes_algorithmic += 16 * n_local;
    {
        // what one apply must move when every value array and every index structure is read
        // once (index arrays are shared between blocks of equal sparsity)
        std::vector<char> va_used(values.size(), 0), pat_used(patterns.size(), 0);
        for (auto &kv : blocks) va_used[kv.second.va] = 1;
        info.bytes_streamed = 16 * n_local;
        for (size_t v = 0; v < values.size(); ++v)
            if (va_used[v]) {
                info.bytes_streamed += 8 * patterns[values[v].pattern].nnz;
                pat_used[values[v].pattern] = 1;
            }
        for (size_t q = 0; q < patterns.size(); ++q)
            if (pat_used[q]) info.bytes_streamed += 4 * patterns[q].nnz + 4 * (patterns[q].nrows + 1);
    }

    // ---- column masks: A P of the operator P A P + alpha (I - P) (preconditioner.py:95-103)
    {
        std::map<std::pair<int, int>, int> clones;
        const size_t n_orig = values.size();
        for (auto &kv : blocks) {
            Block &b = kv.second;
            const int colk = (b.q == KKT_Q00 || b.q == KKT_Q10) ? b.j : n0 + b.j;
            const NullspaceSpec &ns = nullspaces[colk];
            const int want = ns.kind == 1 ? ns.set_id : -1;
            ValueArray &va = values[b.va];
            if (va.colmask_set == -2) {
                va.colmask_set = want;
            } else if (va.colmask_set != want) {
                auto ck = std::make_pair(b.va, want);
                auto it = clones.find(ck);
                if (it == clones.end()) {
                    const Pattern &P = patterns[va.pattern];
                    ValueArray c = va;
                    c.d_vals = dev_alloc<double>(P.npadded);
                    HIPCHK(hipMemcpy(c.d_vals, va.d_vals, P.npadded * 8, hipMemcpyDeviceToDevice));
                    c.colmask_set = want;
                    values.push_back(c);
                    info.bytes_device_values += P.npadded * 8;
                    it = clones.emplace(ck, (int)values.size() - 1).first;
                }
                b.va = it->second;
            }
        }
        (void)n_orig;
        for (auto &va : values) {
            if (va.colmask_set >= 0) {
                const Pattern &P = patterns[va.pattern];
                launch_mask_columns(stream, va.d_vals, P.d_col, bc_sets[va.colmask_set].d_mask,
                                    P.npadded);
            }
        }
    }

    any_const_ns = false;
    for (auto &ns : nullspaces) any_const_ns |= ns.kind == 2;
    if (any_const_ns && !d_xc) d_xc = new_vec();
    fused_row_masks = !CN;

    // ---- row plan: one RowOp per (row, run of same-pattern terms), chained by accumulation
    // Time-sharded handles keep the block rows that read a neighbour's level (through a halo
    // block) in launches of their own, behind the rows that do not: the halo exchange then runs
    // on the transport's stream while the interior rows compute.
    struct RowPlan {
        std::vector<std::vector<RowOp>> waves;   // waves[w] = ops of launch w
        std::vector<int> slices, R;
    } plan_int, plan_halo;
    std::map<std::tuple<int, int, int>, bool> term_in_halo_plan;
    auto put = [&](RowPlan &pl, size_t w, const RowOp &op, int nslices, int R) {
        if (pl.waves.size() <= w) {
            pl.waves.resize(w + 1);
            pl.slices.resize(w + 1, 0);
            pl.R.resize(w + 1, R);
        }
        if (!pl.waves[w].empty() && pl.R[w] != R) fail(KKT_ERR_STATE, "mixed SELL R in a launch");
        pl.R[w] = R;
        pl.waves[w].push_back(op);
        pl.slices[w] = std::max(pl.slices[w], nslices);
    };
    for (int var = 0; var < 2; ++var) {
        const int nloc = var == 0 ? n0_loc : n1_loc;
        const int64_t nxr = var == 0 ? nx0 : nx1;
        for (int il = 0; il < nloc; ++il) {
            const int gi = global_row(var, il);
            std::vector<const Block *> terms;
            for (int qq = 0; qq < 2; ++qq) {
                const int q = var == 0 ? (qq == 0 ? KKT_Q00 : KKT_Q01)
                                       : (qq == 0 ? KKT_Q10 : KKT_Q11);
                std::vector<const Block *> row;
                for (auto &kv : blocks)
                    if (kv.second.q == q && kv.second.i == gi) row.push_back(&kv.second);
                std::sort(row.begin(), row.end(),
                          [](const Block *a, const Block *b) { return a->order < b->order; });
                terms.insert(terms.end(), row.begin(), row.end());
            }
            bool row_halo = false;
            for (const Block *b : terms) {
                const int lj = level_of(b->j);
                row_halo = row_halo || (sharded && (lj < lo || lj >= hi));
            }
            RowPlan &pl = row_halo ? plan_halo : plan_int;
            const int64_t yoff = local_offset(var, il);
            const int flat_g = var == 0 ? gi : n0 + gi;
            const NullspaceSpec &rns = nullspaces[flat_g];
            size_t w = 0;
            size_t t0 = 0;
            bool wrote = false;
            while (t0 < terms.size() || !wrote) {
                RowOp op{};
                op.mode = EPI_LIN;
                op.nrows = (int32_t)nxr;
                op.y = vref(2, yoff);
                op.y2 = vnull();
                op.ca = 1.0;
                op.cy = 1.0;
                op.cz = 0.0;
                op.yin = wrote ? vref(2, yoff) : vnull();
                op.z = vnull();
                op.mx = vnull();
                op.b = op.pk = op.pkm1 = vnull();
                int R = sell_R, nslices = (int)((nxr + 64 * sell_R - 1) / (64 * sell_R));
                if (t0 < terms.size()) {
                    const int pat = values[terms[t0]->va].pattern;
                    const Pattern &P = patterns[pat];
                    op.col = P.d_col;
                    op.perm = P.d_perm;
                    op.slice_off = P.d_slice_off;
                    op.uniform_w = P.uniform_w;
                    op.nslices = P.nslices;
                    R = P.R;
                    nslices = P.nslices;
                    int nt = 0;
                    while (t0 < terms.size() && nt < MAX_TERMS &&
                           values[terms[t0]->va].pattern == pat) {
                        const Block *b = terms[t0];
                        const bool col0 = b->q == KKT_Q00 || b->q == KKT_Q10;
                        op.t[nt].vals = values[b->va].d_vals;
                        block_term[std::make_tuple(b->q, b->i, b->j)] =
                            std::make_tuple((int)w, pl.waves.size() > w ? (int)pl.waves[w].size() : 0, nt);
                        term_in_halo_plan[std::make_tuple(b->q, b->i, b->j)] = row_halo;
                        const int lj = level_of(b->j);
                        if (!sharded && nullspaces[col0 ? b->j : n0 + b->j].kind == 2) {
                            // column block with a ConstantNullspace: the term reads x - mean(x)
                            // from the handle's own buffer (System::apply centres these blocks
                            // only; every other block is read from the caller's x)
                            op.t[nt].x = VRef{(int64_t)(uintptr_t)(
                                d_xc + local_offset(col0 ? 0 : 1, local_of(b->j))), 0, 0};
                        } else if (!sharded || (lj >= lo && lj < hi)) {
                            op.t[nt].x = vref(1, local_offset(col0 ? 0 : 1, local_of(b->j)));
                        } else if (families == 1) {
                            op.t[nt].x = lj < lo ? vref(3, 0)    // halo below (block lo-1)
                                                 : vref(4, 0);   // halo above (block hi)
                        } else {
                            // two families: fixed halo buffers, addressed absolutely
                            const int v = col0 ? 0 : 1, f = family_of(b->j), side = lj < lo ? 0 : 1;
                            if (!d_halo2[v][f][side]) {
                                const int64_t nxh = v == 0 ? nx0 : nx1;
                                d_halo2[v][f][side] = dev_alloc<double>(nxh);
                                HIPCHK(hipMemset(d_halo2[v][f][side], 0, nxh * 8));
                            }
                            op.t[nt].x = VRef{(int64_t)(uintptr_t)d_halo2[v][f][side], 0, 0};
                        }
                        ++nt;
                        ++t0;
                    }
                    op.nterms = nt;
                } else {
                    // a row without blocks still produces y = 0 (+ corrections)
                    op.nterms = 0;
                    op.nslices = nslices;
                    op.col = nullptr;
                    // slice_off of zeros: every slice has width 0
                    std::vector<int32_t> z(nslices + 1, 0);
                    op.slice_off = dev_upload(z.data(), z.size());
                    op.uniform_w = 0;
                }
                const bool last = t0 >= terms.size();
                if (last && fused_row_masks && rns.kind == 1) {
                    // y = P y + alpha (I - P) x fused into the epilogue (preconditioner.py:
                    // 527-537).  Bases::p[0] is x after the ConstantNullspace correction, which
                    // leaves Dirichlet blocks untouched, so it is the original x here.
                    op.rowmask = bc_sets[rns.set_id].d_mask;
                    op.mx = vref(1, yoff);
                    op.malpha = rns.alpha;
                }
                put(pl, w, op, nslices, R);
                wrote = true;
                ++w;
            }
        }
    }
    std::vector<std::vector<RowOp>> waves = plan_int.waves;
    std::vector<int> wave_slices = plan_int.slices, wave_R = plan_int.R;
    first_halo_launch = (int)waves.size();
    for (size_t w = 0; w < plan_halo.waves.size(); ++w) {
        waves.push_back(plan_halo.waves[w]);
        wave_slices.push_back(plan_halo.slices[w]);
        wave_R.push_back(plan_halo.R[w]);
    }
    for (auto &kv : block_term)
        if (term_in_halo_plan[kv.first]) std::get<0>(kv.second) += first_halo_launch;
    for (size_t w = 0; w < waves.size(); ++w) {
        RowLaunch L;
        L.nops = (int)waves[w].size();
        L.max_slices = wave_slices[w];
        L.R = wave_R[w];
        L.uniform_w = 0;
        for (const RowOp &op : waves[w]) {
            if (op.nterms == 0) continue;
            if (L.uniform_w == 0)
                L.uniform_w = op.uniform_w;
            else if (L.uniform_w != op.uniform_w)
                L.uniform_w = -1;
        }
        // Shared values ("mode S"): ops whose terms use the same matrices in the same order are
        // made neighbours and served four at a time by the SpMM-shaped kernel
        // (kkt_spmv_rows_shared); pointless when every block has its own values (mode G).
        {
            auto same = [](const RowOp &a, const RowOp &b) {
                if (a.col != b.col || a.nterms != b.nterms || a.nslices != b.nslices ||
                    a.nrows != b.nrows || a.uniform_w != b.uniform_w || a.rowmask != b.rowmask ||
                    a.perm != b.perm)
                    return false;
                for (int t = 0; t < a.nterms; ++t)
                    if (a.t[t].vals != b.t[t].vals) return false;
                return true;
            };
            std::vector<RowOp> &ops = waves[w];
            std::vector<int> order;
            std::vector<char> used(ops.size(), 0);
            std::vector<int32_t> groups;
            const char *sg = opt("shared_rows");
            const bool allow = !(sg && sg[0] == '0') && L.R == 2 && L.uniform_w >= 1 &&
                               L.uniform_w <= 8;
            bool uniform_all = allow;
            for (const RowOp &op : ops) uniform_all = uniform_all && op.nterms > 0 && op.perm == nullptr;
            if (uniform_all) {
                for (size_t i = 0; i < ops.size(); ++i) {
                    if (used[i]) continue;
                    std::vector<int> run{(int)i};
                    used[i] = 1;
                    for (size_t j = i + 1; j < ops.size(); ++j)
                        if (!used[j] && same(ops[i], ops[j])) {
                            run.push_back((int)j);
                            used[j] = 1;
                        }
                    for (size_t q = 0; q < run.size(); q += ROW_GROUP_MAX) {
                        const int cnt = (int)std::min<size_t>(ROW_GROUP_MAX, run.size() - q);
                        groups.push_back((int32_t)order.size());
                        groups.push_back(cnt);
                        for (int e = 0; e < cnt; ++e) order.push_back(run[q + e]);
                    }
                }
                if (groups.size() / 2 < ops.size()) {      // something is shared
                    std::vector<RowOp> sorted;
                    std::vector<int> new_index(ops.size());
                    for (size_t q = 0; q < order.size(); ++q) {
                        sorted.push_back(ops[order[q]]);
                        new_index[order[q]] = (int)q;
                    }
                    for (auto &kv : block_term)
                        if (std::get<0>(kv.second) == (int)w)
                            std::get<1>(kv.second) = new_index[std::get<1>(kv.second)];
                    ops.swap(sorted);
                    L.d_groups = dev_upload(groups.data(), groups.size());
                    L.ngroups = (int)groups.size() / 2;
                    if (opt("verbose"))
                        std::fprintf(stderr, "[kkt] operator apply, launch %zu: %zu block rows in %d "
                                     "groups of equal structure (shared values)\n", w, ops.size(),
                                     L.ngroups);
                }
            }
        }
        // ragged launch: which kernel (kernels.hpp, UNIFORM_W_SWITCH)
        {
            const char *rs = opt("ragged_switch");
            if (L.uniform_w == -1 && L.R == 2 && !(rs && rs[0] == '0')) {
                int64_t slots = 0, covered = 0, nsl = 0;
                int widest = 0;
                for (const RowOp &op : waves[w]) {
                    if (op.nterms == 0) continue;
                    for (const Pattern &P : patterns) {
                        if (P.d_col != op.col) continue;
                        for (int s = 0; s < P.nslices; ++s) {
                            const int ws = P.h_slice_off[s + 1] - P.h_slice_off[s];
                            slots += (int64_t)ws * op.nterms;
                            nsl += op.nterms;
                            widest = std::max(widest, ws);
                            if (ragged_switch_width(ws)) covered += (int64_t)ws * op.nterms;
                        }
                        break;
                    }
                }
                if (slots > 0 && widest <= 7) {
                    L.uniform_w = UNIFORM_W_SWITCH_NARROW;
                    info.apply_switched++;
                } else if (slots > 0 && covered * 4 >= slots * 3) {
                    L.uniform_w = slots >= 10 * nsl ? UNIFORM_W_SWITCH_1WAVE : UNIFORM_W_SWITCH;
                    info.apply_switched++;
                }
                if (opt("verbose"))
                    std::fprintf(stderr, "[kkt] operator apply, launch %zu: ragged, %.1f %% of the "
                                 "slots in slices of an unrolled width -> %s\n", w,
                                 slots ? 100.0 * covered / slots : 0.0,
                                 L.uniform_w == UNIFORM_W_SWITCH ? "width-switched kernel" :
                                 L.uniform_w == UNIFORM_W_SWITCH_1WAVE ? "width-switched kernel, one "
                                 "wave per workgroup" :
                                 L.uniform_w == UNIFORM_W_SWITCH_NARROW ? "width-switched kernel for "
                                 "narrow slices" : "slot loop");
            }
        }
        L.d_ops = dev_upload(waves[w].data(), waves[w].size());
        info.apply_launches++;
        apply_launches.push_back(L);
    }
    h_apply_ops = waves;

    // ---- CN time transforms (preconditioner.py:437-525)
    if (CN) {
        if (sharded && families == 2) {
            // local blocks of a variable: family 0 levels [lo, hi), then family 1 levels [lo, hi)
            const int nl = hi - lo;
            const bool split = sub00 >= 0;     // without a split both families of a variable
            time_groups.push_back(TimeGroup{0, nl, 1, nx0});              // share its transform
            time_groups.push_back(TimeGroup{nl, nl, split ? 2 : 1, nx0});
            time_groups.push_back(TimeGroup{n0_loc, nl, 2, nx1});
            time_groups.push_back(TimeGroup{n0_loc + nl, nl, split ? 1 : 2, nx1});
        } else if (sub00 < 0) {
            time_groups.push_back(TimeGroup{0, n0_loc, 1, nx0});
            time_groups.push_back(TimeGroup{n0_loc, n1_loc, 2, nx1});
        } else {
            time_groups.push_back(TimeGroup{0, sub00, 1, nx0});
            time_groups.push_back(TimeGroup{sub00, n0 - sub00, 2, nx0});
            time_groups.push_back(TimeGroup{n0, sub11, 2, nx1});
            time_groups.push_back(TimeGroup{n0 + sub11, n1 - sub11, 1, nx1});
        }
    }
    // ---- post-correction jobs (Dirichlet) per local flat block, used when not fused
    {
        std::vector<MaskJob> jobs;
        for (int var = 0; var < 2; ++var) {
            const int nloc = var == 0 ? n0_loc : n1_loc;
            for (int il = 0; il < nloc; ++il) {
                const int gi = global_row(var, il);
                const NullspaceSpec &ns = nullspaces[var == 0 ? gi : n0 + gi];
                MaskJob j{nullptr, 0.0};
                if (ns.kind == 1) {
                    j.mask = bc_sets[ns.set_id].d_mask;
                    j.alpha = ns.alpha;
                }
                jobs.push_back(j);
            }
        }
        d_mask_jobs = dev_upload(jobs.data(), jobs.size());
        for (auto &j : jobs) j.alpha = 1.0;
        d_mask_jobs_one = dev_upload(jobs.data(), jobs.size());
    }
    if (any_const_ns) {
        std::vector<ConstJob> cj;
        for (int var = 0; var < 2; ++var) {
            const int nloc = var == 0 ? n0_loc : n1_loc;
            const int64_t nxv = var == 0 ? nx0 : nx1;
            for (int il = 0; il < nloc; ++il) {
                const int g = var == 0 ? global_row(0, il) : n0 + global_row(1, il);
                if (nullspaces[g].kind != 2) continue;
                cj.push_back({local_offset(var, il), nxv, -1.0 / (double)nxv, 1.0 / (double)nxv,
                              nullspaces[g].alpha / (double)nxv});
                const_max_nx = std::max(const_max_nx, nxv);
            }
        }
        n_const_jobs = (int)cj.size();
        d_const_jobs = dev_upload(cj.data(), cj.size());
        d_sums = dev_alloc<double>(2 * cj.size());
    }
    if (sharded) {
        d_halo_x0_lo = dev_alloc<double>(nx0);
        d_halo_x1_hi = dev_alloc<double>(nx1);
        HIPCHK(hipMemset(d_halo_x0_lo, 0, nx0 * 8));
        HIPCHK(hipMemset(d_halo_x1_hi, 0, nx1 * 8));
        for (TimeGroup &g : time_groups) {
            g.d_halo = dev_alloc<double>(g.nx);
            HIPCHK(hipMemset(g.d_halo, 0, g.nx * 8));
        }
    }
    HIPCHK(hipStreamSynchronize(stream));
    finalized = true;
}

void System::update_block_values(int q, int i, int j, const double *vals) {
    if (!finalized) fail(KKT_ERR_STATE, "kkt_update_block_values needs a finalized system");
    auto it = blocks.find(std::make_tuple(q, i, j));
    if (it == blocks.end()) fail(KKT_ERR_ARG, "no such block");
    Block &blk = it->second;
    int users = 0;
    for (auto &kv : blocks) users += kv.second.va == blk.va;
    const int pat = values[blk.va].pattern;
    const Pattern &P = patterns[pat];
    double *d_csr = dev_upload(vals, (size_t)P.nnz);
    auto fill = [&](double *dst, int colmask_set) {
        launch_csr_to_sell(stream, d_csr, P.d_sell2csr, dst, P.npadded);
        if (colmask_set >= 0)
            launch_mask_columns(stream, dst, P.d_col, bc_sets[colmask_set].d_mask, P.npadded);
    };
    if (users <= 1) {
        fill(values[blk.va].d_vals, values[blk.va].colmask_set);
        HIPCHK(hipStreamSynchronize(stream));
        HIPCHK(hipFree(d_csr));
        pc_stale = true;
        return;
    }
    // The value array is shared with other blocks (same object at construction): the
    // reference re-assembles every block on its own, so an update must not reach the sharers.
    // Identical values (a Picard loop re-sending its mass couplings) change nothing; new
    // values give this block a private array (copy on write) and its RowOp term is re-pointed.
    double *d_new = dev_alloc<double>(P.npadded);
    fill(d_new, values[blk.va].colmask_set);
    unsigned *d_flag = dev_alloc<unsigned>(1);
    HIPCHK(hipMemsetAsync(d_flag, 0, sizeof(unsigned), stream));
    launch_vals_differ(stream, d_new, values[blk.va].d_vals, P.npadded, d_flag);
    unsigned differ = 0;
    HIPCHK(hipMemcpyAsync(&differ, d_flag, sizeof differ, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    HIPCHK(hipFree(d_flag));
    HIPCHK(hipFree(d_csr));
    if (!differ) {
        HIPCHK(hipFree(d_new));
        return;
    }
    ValueArray c = values[blk.va];
    const double *d_old = c.d_vals;
    c.d_vals = d_new;
    c.share_id = -1;
    values.push_back(c);
    blk.va = (int)values.size() - 1;
    info.bytes_device_values += P.npadded * 8;
    info.n_value_arrays++;
    info.bytes_algorithmic += 12 * P.nnz + 4 * (P.nrows + 1);
    info.bytes_streamed += 8 * P.nnz;
    auto loc = block_term.find(std::make_tuple(q, i, j));
    if (loc == block_term.end()) fail(KKT_ERR_STATE, "block missing from the apply plan");
    const int L = std::get<0>(loc->second), o = std::get<1>(loc->second), t = std::get<2>(loc->second);
    RowOp &op = h_apply_ops[L][o];
    if (op.t[t].vals != d_old) fail(KKT_ERR_STATE, "apply plan out of step with the block table");
    op.t[t].vals = d_new;
    HIPCHK(hipMemcpy(apply_launches[L].d_ops + o, &op, sizeof(RowOp), hipMemcpyHostToDevice));
    apply_launches[L].ngroups = 0;   // the op left its group of equal structure: plain kernel
    pc_stale = true;
}

// y = A x  (preconditioner.py:375-543)
void System::apply(const double *d_x, double *d_y) {
    if (!finalized) fail(KKT_ERR_STATE, "system not finalized");
    info.last_op_applies++;
    const double *xin = d_x;
    const int nb = n0_loc + n1_loc;
    if (any_const_ns && !sharded) {
        // x_c = x - mean(x) on ConstantNullspace blocks (preconditioner.py:145-146, 384-393):
        // only those blocks are written to d_xc, and only the terms on them read it
        launch_const_center(stream, d_const_jobs, n_const_jobs, const_max_nx, d_x, d_xc, d_sums);
    } else if (any_const_ns) {
        // time shards: the halo exchange sends blocks of x_c, so all of it is formed
        launch_copy(stream, d_xc, d_x, n_local);
        launch_const_correct(stream, d_const_jobs, n_const_jobs, const_max_nx, d_xc, nullptr, 0,
                             d_sums);
        xin = d_xc;
    }
    if (sharded) {
        // the exchange runs on the transport's stream, behind whatever produced x ...
        if (!comm_stream) {
            HIPCHK(hipStreamCreateWithFlags(&comm_stream, hipStreamNonBlocking));
            HIPCHK(hipEventCreateWithFlags(&ev_x_ready, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&ev_halo_ready, hipEventDisableTiming));
        }
        HIPCHK(hipEventRecord(ev_x_ready, stream));
        HIPCHK(hipStreamWaitEvent(comm_stream, ev_x_ready, 0));
    }
    // Crank-Nicolson: the block rows leave the raw rows rho in a buffer of their own; the time
    // transform then writes y from it, every (level, dof) independently, with the Dirichlet
    // post-correction fused (one pass instead of a serial in-place transform and a mask pass)
    double *rows_out = d_y;
    if (CN) {
        if (!d_tmp_y) d_tmp_y = new_vec();
        rows_out = d_tmp_y;
    }
    Bases B{{xin, rows_out, d_halo_x0_lo, d_halo_x1_hi}};
    {
        const char *rx = opt("ragged_xcd");
        set_ragged_xcd(!(rx && rx[0] == '0'));
        const char *ax = opt("apply_xcd");
        set_apply_xcd(ax && ax[0] == '1');
    }
    auto launch = [&](const RowLaunch &L) {
        if (L.ngroups > 0 &&
            launch_rowops_grouped(stream, L.d_ops, L.d_groups, L.ngroups, L.max_slices, L.R,
                                  L.uniform_w, B))
            return;
        launch_rowops(stream, L.d_ops, L.nops, L.max_slices, L.R, B, 0, L.uniform_w);
    };
    // ... while the block rows that need no neighbour's level are already queued
    for (int w = 0; w < first_halo_launch; ++w) launch(apply_launches[w]);
    if (sharded) {
        hipStream_t compute = stream;
        stream = comm_stream;          // the exchange helpers work on S.stream
        try {
            if (families == 1)
                comm_exchange_x_halos(*this, xin);
            else
                comm_exchange_x_halos2(*this, xin);
        } catch (...) {
            stream = compute;
            throw;
        }
        stream = compute;
        HIPCHK(hipEventRecord(ev_halo_ready, comm_stream));
        HIPCHK(hipStreamWaitEvent(stream, ev_halo_ready, 0));
    }
    for (size_t w = (size_t)first_halo_launch; w < apply_launches.size(); ++w) launch(apply_launches[w]);
    if (CN) {
        if (sharded) comm_exchange_row_halos(*this, rows_out);
        for (const TimeGroup &g : time_groups) {
            const int64_t off = g.first_local_block < n0_loc
                                    ? (int64_t)g.first_local_block * nx0
                                    : (int64_t)n0_loc * nx0 +
                                          (int64_t)(g.first_local_block - n0_loc) * nx1;
            const double *lo_h = nullptr, *hi_h = nullptr;
            if (sharded) {
                if (g.kind == 1 && hi < mf) hi_h = g.d_halo;
                if (g.kind == 2 && lo > 0) lo_h = g.d_halo;
            }
            // y = P T rho + alpha (I - P) x (preconditioner.py:437-470, 527-537)
            launch_time_transform_mask(stream, d_y + off, rows_out + off, d_x + off,
                                       d_mask_jobs + g.first_local_block, g.kind, g.n, g.nx, lo_h,
                                       hi_h);
        }
    } else if (!fused_row_masks) {
        // y = P y + alpha (I - P) x on Dirichlet blocks (preconditioner.py:527-537)
        if (nx0 == nx1) {
            launch_mask_blocks(stream, d_y, d_y, d_x, d_mask_jobs, nb, nx0);
        } else {
            launch_mask_blocks(stream, d_y, d_y, d_x, d_mask_jobs, n0_loc, nx0);
            launch_mask_blocks(stream, d_y + (int64_t)n0_loc * nx0, d_y + (int64_t)n0_loc * nx0,
                               d_x + (int64_t)n0_loc * nx0, d_mask_jobs + n0_loc, n1_loc, nx1);
        }
    }
    if (any_const_ns) {
        // y -= mean(y); y += alpha * mean(x)  (preconditioner.py:137-152)
        launch_const_correct(stream, d_const_jobs, n_const_jobs, const_max_nx, d_y, d_x, 2, d_sums);
    }
}

}  // namespace kkt
