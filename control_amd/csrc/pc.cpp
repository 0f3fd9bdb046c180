#include "pc.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>

namespace kkt {

static VRef vabs(const double *p) {
    return p ? VRef{(int64_t)(uintptr_t)p, 0, 0} : VRef{0, -1, 0};
}

SchurPC::SchurPC(System &S, const kkt_pc_desc &d) : S_(S), d_(d) {
    if (!S.finalized) fail(KKT_ERR_STATE, "kkt_set_pc_schur needs a finalized system");
    if (d.kind < KKT_PC_STATIONARY || d.kind > KKT_PC_INSTATIONARY_CN)
        fail(KKT_ERR_ARG, "unknown preconditioner kind");
    if (!d.m_indptr || !d.m_indices || !d.m_values) fail(KKT_ERR_ARG, "mass matrix missing");
    if (S.nx0 != S.nx1 || d.nx != S.nx0)
        fail(KKT_ERR_ARG, "built-in preconditioner needs equal spatial spaces of size nx");
    nx_ = d.nx;
    if (d.kind == KKT_PC_STATIONARY)
        n_ = 1;
    else
        n_ = d.kind == KKT_PC_INSTATIONARY_BE ? d.n_t : d.n_t - 1;
    if (n_ != S.n0 || n_ != S.n1 || n_ < 1)
        fail(KKT_ERR_ARG, "n_t does not match the block counts of the system");
    lo_ = S.sharded ? S.lo : 0;
    hi_ = S.sharded ? S.hi : n_;
    if (d.kind != KKT_PC_STATIONARY && n_ < 2) fail(KKT_ERR_ARG, "need at least two blocks");
    // schur_its == -1: degree from the spectrum; schur_emin <= 0: interval from the spectrum
    if (d.mass_its < 0 || d.schur_its < -1) fail(KKT_ERR_ARG, "negative Chebyshev degree");
    if ((d.mass_its > 0 && !(d.mass_emax > d.mass_emin && d.mass_emin > 0)) ||
        (d.schur_emin > 0 && !(d.schur_emax > d.schur_emin)) || d.schur_eimag < 0)
        fail(KKT_ERR_ARG, "Chebyshev bounds must satisfy 0 < emin < emax, eimag >= 0");
    // deep copies: the caller's arrays are not kept (kkt.h conventions)
    m_indptr_.assign(d.m_indptr, d.m_indptr + nx_ + 1);
    m_indices_.assign(d.m_indices, d.m_indices + m_indptr_[nx_]);
    m_values_.assign(d.m_values, d.m_values + m_indptr_[nx_]);
    if (d.n_bc > 0) bc_idx_.assign(d.bc_idx, d.bc_idx + d.n_bc);
    if (d.coarse_cycles < 0) fail(KKT_ERR_ARG, "negative number of coarse cycles");
    if (d.coarse_cycles > 0) {
        if (!d.p_indptr || !d.p_indices || !d.p_values || d.n_coarse < 1)
            fail(KKT_ERR_ARG, "coarse_cycles > 0 needs the prolongation matrix P");
        if (d.n_coarse > 4096) fail(KKT_ERR_ARG, "coarse space too large (dense inverse): n_coarse <= 4096");
        coarse_cycles_ = d.coarse_cycles;
        p_indptr_.assign(d.p_indptr, d.p_indptr + nx_ + 1);
        const int64_t pn = p_indptr_[nx_];
        if (p_indptr_[0] != 0 || pn < 0) fail(KKT_ERR_ARG, "bad P indptr");
        p_indices_.assign(d.p_indices, d.p_indices + pn);
        p_values_.assign(d.p_values, d.p_values + pn);
        for (int64_t r = 0; r < nx_; ++r)
            if (p_indptr_[r + 1] < p_indptr_[r]) fail(KKT_ERR_ARG, "P indptr not monotone");
        for (int32_t c : p_indices_)
            if (c < 0 || c >= d.n_coarse) fail(KKT_ERR_ARG, "P column index out of range");
    }
    d_.p_indptr = d_.p_indices = nullptr;
    d_.p_values = nullptr;
    d_.m_indptr = d_.m_indices = nullptr;
    d_.m_values = nullptr;
    d_.bc_idx = nullptr;
    const char *e = S_.opt("no_graph");
    use_graph_ = !(e && e[0] == '1');
    e = S_.opt("persistent");
    use_programs_ = !(e && e[0] == '0');
    // opt-in: measured 0-3 % on cfg 2 (both lanes slow down when they share the chip, and the
    // smaller batches of a chunk are less efficient), DESIGN.md section 6
    e = S_.opt("lanes");
    use_lanes_ = e && e[0] == '1';
    build();
}

SchurPC::~SchurPC() {
    clear_program();
    tile_plan_.release();
    for (hipEvent_t e : events_) (void)hipEventDestroy(e);
    if (side_) (void)hipStreamDestroy(side_);
    for (void *p : owned_)
        if (p) (void)hipFree(p);
    for (double *p : einv_owned_) (void)hipFree(p);
}

void SchurPC::clear_program() {
    for (Segment &g : segments_) {
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        if (g.graph) (void)hipGraphDestroy(g.graph);
    }
    segments_.clear();
    for (auto &s : steps_)
        if ((s.kind == PcStep::ROWS || s.kind == PcStep::PROG) && s.rows.d_ops)
            (void)hipFree(s.rows.d_ops);
    for (auto &s : steps_)
        if (s.d_il) (void)hipFree(s.d_il);
    for (auto &s : steps_)
        if (s.d_lite) (void)hipFree(s.d_lite);
    for (auto &s : steps_)
        if (s.d_levels) (void)hipFree(s.d_levels);
    for (void *p : tile_owned_) (void)hipFree(p);
    tile_owned_.clear();
    sweep_levels_.clear();
    steps_.clear();
    n_events_ = 0;
    cur_lane_ = 0;
}

int SchurPC::emit_record(int lane) {
    PcStep s;
    s.kind = PcStep::EV_RECORD;
    s.lane = lane;
    s.ev = n_events_++;
    steps_.push_back(s);
    return s.ev;
}

void SchurPC::emit_wait(int lane, int ev) {
    PcStep s;
    s.kind = PcStep::EV_WAIT;
    s.lane = lane;
    s.ev = ev;
    steps_.push_back(s);
}

void SchurPC::values_changed() {
    // Schur matrices are sums with block values: rebuild them and the program
    clear_program();
    for (auto &kv : mats_) {
        (void)hipFree(kv.second.vals);
        (void)hipFree(kv.second.dinv);
    }
    mats_.clear();
    for (double *p : einv_owned_) (void)hipFree(p);
    einv_owned_.clear();
    if (d_.kind == KKT_PC_STATIONARY)
        build_stationary();
    else if (d_.kind == KKT_PC_INSTATIONARY_BE)
        build_BE();
    else
        build_CN();
    if (S_.opt("verbose") && d_.schur_emin <= 0)
        std::fprintf(stderr, "[kkt] Chebyshev sub-solves: degree %d; %lld Lanczos steps spent on "
                     "%zu matrices\n", schur_its_, (long long)spectrum_steps_, mats_.size());
    S_.info.sweep_form = 0;
    S_.info.sweep_tiles = S_.info.sweep_threads = S_.info.sweep_depth = S_.info.sweep_row_slots = 0;
    S_.info.sweep_its = schur_its_;
    fuse_programs();
}

// Shape and dependency tables of the row programs (counter and data-flow forms).
bool SchurPC::setup_row_programs() {
    const Pattern &P = S_.patterns[m_pat_];
    {
        int wpw = 0, nwg = 0;
        const char *pm0 = S_.opt("prog_mode");
        const bool try_g0 = !(pm0 && pm0[0] == 'f') && row_program_g_available(P.R, P.uniform_w);
        // the data-flow form for any width (matrix re-read from L2 every phase) is opt-in
        // (KKT_PROG_MODE=w): re-polling whole chunks of granules costs more fabric traffic
        // than the counter form's single gather round once rows are wide or row-sorted
        // (measured: Stokes P2 47 -> 121 ms, 3-D P1 3.3 -> 4.8 ms per application)
        const bool try_gw = pm0 && pm0[0] == 'w' && P.R == 2;
        const bool try_g = !try_gw && try_g0;
        // data-flow form: one wave per workgroup (no workgroup barrier on the critical path)
        // while all of them are co-resident; else 4 / 8 waves per workgroup
        const char *pw = S_.opt("prog_waves");
        const int first = pw ? std::atoi(pw) : (try_g ? 1 : 4);
        int n_cus = 0, dev_id = 0;
        if (hipGetDevice(&dev_id) != hipSuccess ||
            hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, dev_id) != hipSuccess)
            n_cus = 0;
        // Multi-wave workgroups: one per CU is the shape preferred (pass 0).  Two 4-wave
        // workgroups of the data-flow form on a CU did not stay co-resident reliably (below);
        // the counter form has run that way (P2 Stokes: 259 workgroups) without a failure, so it
        // remains the fallback (pass 1) when no one-per-CU shape fits.
        for (int pass = 0; pass < 2 && !wpw; ++pass)
        for (int cand : {first, 4, 8}) {
            if (cand < 1 || cand > 8) continue;
            if (pass == 0 && cand > 1 && !(pw && cand == first) &&
                (P.nslices + cand - 1) / cand > n_cus)
                continue;
            // (Round 1 fenced the data-flow form with two 4-wave workgroups on a CU off: programs
            // of about 1 000 phases on 401^2 -- 315 workgroups -- ran into the bounded spin.  Its
            // poll loop spun at full rate; an older wave that spins takes the issue slots and the
            // memory queue of the CU whose younger waves must produce what it waits for.  With
            // the back-off in the poll loop the same shape reproduces the plain launches bit
            // for bit (scripts/dbg_midsize.py 400 12 80, KKT_PROG_WAVES=4); a time-out is no
            // longer fatal either -- the step falls back to plain launches.)
            const int n = (P.nslices + cand - 1) / cand;
            int cap = row_program_max_wgs(P.R, P.uniform_w, cand);
            if (try_g) cap = std::min(cap, row_program_g_max_wgs(P.uniform_w, cand));
            if (try_gw) cap = std::min(cap, row_program_gw_max_wgs(cand));
            if (n <= cap) {
                wpw = cand;
                nwg = n;
                break;
            }
        }
        if (!wpw) {
            // more than 2 048 slices: the counter form held to 168 registers keeps three waves
            // per SIMD resident (kernels.hip, pc_row_program_lo)
            // (rows of up to 8 entries only: wider ones spill at 168 registers and run slower than
            // the plain launches -- 26.8 against 20 us per step on 64^3 P1)
            for (int cand : {4, 8}) {
                if (P.uniform_w < 1 || P.uniform_w > 8) break;
                const int n = (P.nslices + cand - 1) / cand;
                if (n <= row_program_max_wgs(P.R, P.uniform_w, cand, true)) {
                    wpw = cand;
                    nwg = n;
                    prog_lowreg_ = true;
                    break;
                }
            }
        }
        if (!wpw) return false;
        // workgroup j must wait for every workgroup whose rows it gathers from, and for
        // every workgroup that gathers from its rows (write-after-read on rotating buffers)
        const int64_t rpw = (int64_t)wpw * 64 * P.R;
        std::vector<int32_t> lo(nwg), hi(nwg);
        const int64_t npos = (int64_t)P.nslices * 64 * P.R;
        for (int j = 0; j < nwg; ++j) {
            // producers of the columns this workgroup gathers, as workgroup indices (positions
            // of the rows in the SELL storage: row-sorted structures permute them)
            int64_t wmin = j, wmax = j;
            const int64_t p1 = std::min<int64_t>(npos, (j + 1) * rpw);
            for (int64_t p = j * rpw; p < p1; ++p) {
                const int64_t r = P.row_of(p);
                if (r < 0) continue;
                for (int32_t q = P.h_indptr[r]; q < P.h_indptr[r + 1]; ++q) {
                    const int64_t w = P.pos_of(P.h_indices[q]) / rpw;
                    wmin = std::min(wmin, w);
                    wmax = std::max(wmax, w);
                }
            }
            lo[j] = (int32_t)wmin;
            hi[j] = (int32_t)std::min<int64_t>(nwg - 1, wmax);
        }
        std::vector<int32_t> slo = lo, shi = hi;
        for (int j = 0; j < nwg; ++j)
            for (int k = lo[j]; k <= hi[j]; ++k) {
                slo[k] = std::min(slo[k], (int32_t)j);
                shi[k] = std::max(shi[k], (int32_t)j);
            }
        std::vector<int32_t> dep(2 * (size_t)nwg);
        for (int j = 0; j < nwg; ++j) {
            if (shi[j] - slo[j] + 1 > 64) return false;   // one polling wave covers at most 64 neighbours
            dep[2 * j] = slo[j];
            dep[2 * j + 1] = shi[j];
        }
        // data-flow form: needs the exact gather relation between workgroups to be symmetric
        {
            const char *pm = S_.opt("prog_mode");
            const bool gw_ok = pm && pm[0] == 'w' && P.R == 2 &&
                               nwg <= row_program_gw_max_wgs(wpw);
            const bool g_ok = !gw_ok && row_program_g_available(P.R, P.uniform_w) &&
                              nwg <= row_program_g_max_wgs(P.uniform_w, wpw);
            bool want = !(pm && pm[0] == 'f') && (g_ok || gw_ok) && !prog_lowreg_;
            if (want) {
                // between waves (the unit that publishes and polls), in storage positions
                const int64_t rw = 64 * P.R;
                const int nw = P.nslices;
                std::vector<std::vector<int32_t>> reads(nw);
                for (int j = 0; j < nw; ++j) {
                    std::vector<char> seen(nw, 0);
                    for (int64_t p = j * rw; p < (j + 1) * rw; ++p) {
                        const int64_t r = P.row_of(p);
                        if (r < 0) continue;
                        for (int32_t q = P.h_indptr[r]; q < P.h_indptr[r + 1]; ++q)
                            seen[P.pos_of(P.h_indices[q]) / rw] = 1;
                    }
                    for (int k = 0; k < nw; ++k)
                        if (seen[k]) reads[j].push_back(k);
                }
                for (int j = 0; j < nw && want; ++j)
                    for (int32_t k : reads[j])
                        if (!std::binary_search(reads[k].begin(), reads[k].end(), (int32_t)j))
                            want = false;
            }
            prog_mode_ = want ? (g_ok ? 1 : 2) : 0;
            prog_granule_ = want;
            if (want) {
                granule_words_ = 2 * (size_t)P.nslices * 64 * P.R;
                d_g0_ = dev_alloc<unsigned long long>(granule_words_);
                d_g1_ = dev_alloc<unsigned long long>(granule_words_);
                owned_.push_back(d_g0_);
                owned_.push_back(d_g1_);
            }
        }
        prog_wpw_ = wpw;
        prog_nwg_ = nwg;
        if (S_.opt("verbose"))
            std::fprintf(stderr, "[kkt] sweep program: mode %d (0 counters, 1 data-flow, 2 data-flow "
                         "any width), %d workgroups x %d waves, %d slices\n",
                         prog_mode_, nwg, wpw, P.nslices);
        d_dep_ = dev_upload(dep.data(), dep.size());
        d_flags_ = dev_alloc<unsigned>(prog_flag_words(nwg));
        owned_.push_back(d_dep_);
        owned_.push_back(d_flags_);
    }
    return true;
}

// Tile plan and granule buffers of the tile sweep program (tile_kernels.hip), once per
// preconditioner.  False when the structure does not fit (then the other forms run).
bool SchurPC::prepare_tiles() {
    if (tile_tried_) return tile_ok_;
    tile_tried_ = true;
    const Pattern &P = S_.patterns[m_pat_];
    int n_cus = 0, dev_id = 0;
    if (hipGetDevice(&dev_id) != hipSuccess ||
        hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, dev_id) != hipSuccess ||
        n_cus < 1)
        return false;
    const char *tw = S_.opt("tile_waves");
    const char *td = S_.opt("tile_depth");
    const int depth = td ? std::atoi(td) : 0;
    // one workgroup per CU; tiny meshes get fewer tiles (at least 32 own rows each)
    const int ntiles = (int)std::max<int64_t>(1, std::min<int64_t>(n_cus, P.nrows / 32));
    // workgroup size: 1 024 threads (one row slot per thread, 128 registers) or 512 (up to three
    // row slots, 256 registers) -- whichever the model prefers, unless the caller chose
    int threads = 0;
    // Dirichlet rows belong to no tile (their iterates are exactly zero)
    std::vector<uint8_t> hmask;
    if (!bc_idx_.empty()) {
        hmask.assign(P.nrows, 0);
        for (int32_t k : bc_idx_) hmask[k] = 1;
    }
    const uint8_t *hm = hmask.empty() ? nullptr : hmask.data();
    // coordinates of the rows, if the caller gave them (kkt_set_tile_coordinates)
    const double *tc = (S_.tile_dim > 0 && (int64_t)S_.tile_coords.size() == P.nrows * S_.tile_dim)
                           ? S_.tile_coords.data() : nullptr;
    const int xh = coarse_cycles_ > 0 ? 2 : 0;     // extra hand-offs per cycle of a two-grid level
    if (tw) {
        threads = 64 * std::max(1, std::min(16, std::atoi(tw)));
        // (the depth model must only consider what a kernel variant exists for: wide rows have
        // one or two row slots, and a deeper plan that needs more would lose the tile form)
        if (!build_tile_plan(P, ntiles, depth, threads,
                             std::max(1, tile_sweep_max_rpt(P.max_width, threads)), tile_plan_, hm,
                             schur_its_, tc, S_.tile_dim, tile_sweep_max_hslots, xh))
            return false;
    } else {
        TilePlan big;
        const bool ok_big = tile_sweep_max_rpt(P.max_width, 1024) >= 1 &&
                            build_tile_plan(P, ntiles, depth, 1024, 1, big, hm, schur_its_, tc, S_.tile_dim,
                                            tile_sweep_max_hslots, xh) &&
                            tile_sweep_available(big.W, big.rpt, 1024, big.hslots);
        const bool ok_small = tile_sweep_max_rpt(P.max_width, 512) >= 1 &&
                              build_tile_plan(P, ntiles, depth, 512,
                                              tile_sweep_max_rpt(P.max_width, 512), tile_plan_, hm,
                                              schur_its_, tc, S_.tile_dim, tile_sweep_max_hslots, xh) &&
                              tile_sweep_available(tile_plan_.W, tile_plan_.rpt, 512, tile_plan_.hslots);
        // wide rows (P2: 19 entries) have no 1 024-thread variant -- 128 registers do not hold a
        // row -- but a 768-thread one: 12 waves with one row each instead of 8 with two
        TilePlan mid;
        const bool ok_mid = !ok_big && tile_sweep_max_rpt(P.max_width, 768) >= 1 &&
                            build_tile_plan(P, ntiles, depth, 768, 1, mid, hm, schur_its_, tc, S_.tile_dim,
                                            tile_sweep_max_hslots, xh) &&
                            tile_sweep_available(mid.W, mid.rpt, 768, mid.hslots);
        if (!ok_big && !ok_small && !ok_mid) return false;
        if (ok_mid && (!ok_small || (mid.max_rows > 512 && mid.model_us <= tile_plan_.model_us) ||
                       (coarse_cycles_ > 0 && mid.max_own > 128)))
            tile_plan_ = mid;
        // (1 024 threads only where more than 512 rows are computed: with fewer, half of the 16
        // waves never have a live row -- 32^3: 293 its/s with 512 threads, 283 with 1 024)
        // (two-grid levels: 16 waves also share the coarse exchange's work -- measured on cfg 2:
        // 1 024 threads at depth 5 118 its/s, 512 threads at depth 4 103)
        if (ok_big && (!ok_small || (big.max_rows > 512 && big.model_us <= tile_plan_.model_us) ||
                       (coarse_cycles_ > 0 && big.max_own > 128)))
            tile_plan_ = big;
        threads = tile_plan_.threads;
    }
    TilePlan &tp = tile_plan_;
    if (!tp.symmetric || !tile_sweep_available(tp.W, tp.rpt, threads, tp.hslots)) return false;
    // (solo levels of the stationary preconditioner run with mass_its, which may exceed schur_its_)
    int max_its = std::max(schur_its_, 2);
    for (const SweepLevel &lv : sweep_levels_) max_its = std::max(max_its, lv.its);
    size_t lds = tile_sweep_lds_bytes(tp.nk_pad, max_its);
    if (tp.ntiles > tile_sweep_max_tiles(tp.W, tp.rpt, threads, lds, tp.hslots)) return false;
    tile_lds_checked_ = lds;
    // two-grid levels: the variant with coarse corrections and its larger LDS footprint
    tile_coarse_ok_ = false;
    if (coarse_cycles_ > 0 && tile_sweep_coarse_available(tp.W, tp.rpt, threads, tp.hslots) &&
        build_tile_coarse()) {
        // the tiles' P entries go into LDS when everything fits in 150 KB, else they stay in memory
        // (and, room permitting, its rows of the coarse inverse: cache = 2)
        for (int cache = 2; cache >= 0 && !tile_coarse_ok_; --cache) {
            const size_t lds_c = tile_sweep_lds_bytes(tp.nk_pad, max_its, h_tile_coarse_.nc,
                                                      h_tile_coarse_.nslots, h_tile_coarse_.jmax,
                                                      cache ? h_tile_coarse_.nr_max : 0,
                                                      cache == 2 ? h_tile_coarse_.nown : 0,
                                                      h_tile_coarse_.ew);
            if (lds_c <= 150 * 1024 &&
                tp.ntiles <= tile_sweep_max_tiles(tp.W, tp.rpt, threads, lds_c, tp.hslots, true)) {
                tile_coarse_ok_ = true;
                h_tile_coarse_.cache_lists = cache >= 1;
                h_tile_coarse_.cache_einv = cache == 2;
                HIPCHK(hipMemcpy(d_tile_coarse_, &h_tile_coarse_, sizeof h_tile_coarse_,
                                 hipMemcpyHostToDevice));
            }
        }
    }
    tp.upload();
    const size_t words = 2 * (size_t)P.nrows;
    for (int i = 0; i < 4; ++i) {
        d_tg_[i] = dev_alloc<unsigned long long>(words);
        owned_.push_back(d_tg_[i]);
    }
    if (S_.opt("verbose"))
        std::fprintf(stderr, "[kkt] tile sweep program: %d tiles x %d threads, depth %d, W %d, "
                     "%d row slots per thread; largest tile: %lld own rows, %lld computed rows, "
                     "%lld ring rows; mean redundancy %.2f; modelled %.2f us per step\n", tp.ntiles,
                     threads, tp.depth, tp.W, tp.rpt, (long long)tp.max_own, (long long)tp.max_rows,
                     (long long)tp.max_halo, tp.mean_redundancy, tp.model_us);
    tile_ok_ = true;
    return true;
}

// Coarse corrections inside the tile program: per tile, the coarse functions its own rows touch
// (J_t) with their restriction lists, the prolongation entries of its own rows, the slots of its
// partial sums; per coarse function, the slots that contribute to it (ascending tile order).
bool SchurPC::build_tile_coarse() {
    const TilePlan &tp = tile_plan_;
    const int nt = tp.ntiles, nc = coarse_.nc;
    if (nt < 1 || nc < 1) return false;
    std::vector<int32_t> nj(nt, 0), slot0(nt, 0);
    std::vector<std::vector<int32_t>> J(nt);
    int n0max = 0;
    for (int t = 0; t < nt; ++t) {
        const int n0 = tp.n[(size_t)t * (TILE_MAX_DEPTH + 1)];
        n0max = std::max(n0max, n0);
        std::vector<int32_t> &j = J[t];
        for (int l = 0; l < n0; ++l) {
            const int32_t g = tp.grow[(size_t)t * tp.nk_pad + l];
            for (int32_t q = p_indptr_[g]; q < p_indptr_[g + 1]; ++q) j.push_back(p_indices_[q]);
        }
        std::sort(j.begin(), j.end());
        j.erase(std::unique(j.begin(), j.end()), j.end());
        nj[t] = (int32_t)j.size();
    }
    int jmax = 0, nslots = 0;
    for (int t = 0; t < nt; ++t) {
        jmax = std::max(jmax, nj[t]);
        slot0[t] = nslots;
        nslots += nj[t];
    }
    if (jmax < 1 || jmax > 65535 || nslots > 65536) return false;
    std::vector<int32_t> jglob((size_t)nt * jmax, -1), r_ip((size_t)nt * jmax + 1, 0),
        p_ip((size_t)nt * n0max + 1, 0), c_ip(nc + 1, 0), c_slot;
    std::vector<uint16_t> r_row, p_k;
    std::vector<double> r_w, p_w;
    std::vector<std::vector<int32_t>> contrib(nc);
    for (int t = 0; t < nt; ++t) {
        const int n0 = tp.n[(size_t)t * (TILE_MAX_DEPTH + 1)];
        const std::vector<int32_t> &j = J[t];
        std::vector<std::vector<std::pair<uint16_t, double>>> lists(j.size());
        for (int l = 0; l < n0; ++l) {
            const int32_t g = tp.grow[(size_t)t * tp.nk_pad + l];
            for (int32_t q = p_indptr_[g]; q < p_indptr_[g + 1]; ++q) {
                const int k = (int)(std::lower_bound(j.begin(), j.end(), p_indices_[q]) - j.begin());
                lists[k].push_back({(uint16_t)l, p_values_[q]});
                p_k.push_back((uint16_t)k);
                p_w.push_back(p_values_[q]);
            }
            p_ip[(size_t)t * n0max + l + 1] = (int32_t)p_k.size();
        }
        for (int l = n0; l < n0max; ++l) p_ip[(size_t)t * n0max + l + 1] = (int32_t)p_k.size();
        for (size_t k = 0; k < (size_t)jmax; ++k) {
            if (k < j.size()) {
                jglob[(size_t)t * jmax + k] = j[k];
                contrib[j[k]].push_back(slot0[t] + (int)k);
                for (auto &e : lists[k]) {
                    r_row.push_back(e.first);
                    r_w.push_back(e.second);
                }
            }
            r_ip[(size_t)t * jmax + k + 1] = (int32_t)r_row.size();
        }
    }
    for (int j = 0; j < nc; ++j) {
        for (int32_t sl : contrib[j]) c_slot.push_back(sl);
        c_ip[j + 1] = (int32_t)c_slot.size();
    }
    auto up = [&](const auto &v) {
        auto *p = dev_upload(v.data(), std::max<size_t>(1, v.size()));
        owned_.push_back((void *)p);
        return p;
    };
    TileCoarseDev &D = h_tile_coarse_;
    D.nc = nc;
    D.jmax = jmax;
    D.n0max = n0max;
    D.nslots = nslots;
    {
        int nrm = 1;
        for (int t = 0; t < nt; ++t)
            nrm = std::max(nrm, r_ip[(size_t)t * jmax + nj[t]] - r_ip[(size_t)t * jmax]);
        D.nr_max = nrm;
    }
    D.nj = up(nj);
    D.jglob = up(jglob);
    D.slot0 = up(slot0);
    D.r_ip = up(r_ip);
    if (r_row.empty()) r_row.push_back(0), r_w.push_back(0.0);
    D.r_row = up(r_row);
    D.r_w = up(r_w);
    D.p_ip = up(p_ip);
    if (p_k.empty()) p_k.push_back(0), p_w.push_back(0.0);
    D.p_k = up(p_k);
    D.p_w = up(p_w);
    D.c_ip = up(c_ip);
    D.c_slot = up(c_slot);
    D.cg_bytes = (uint32_t)((size_t)nslots * 16);
    D.eg_bytes = (uint32_t)((size_t)nc * 16);
    D.nown = (nc + nt - 1) / nt;
    {
        // Diagonal blocks of P^T A P (kernels.hpp, TileCoarseDev): coarse functions i and j are
        // coupled when P[r, i], A[r, s] and P[s, j] are stored entries for some rows r, s --
        // structure only, so the blocks hold for every matrix on this pattern (the Dirichlet
        // rows and columns only remove couplings).  Union-find over the coarse functions.
        const Pattern &P = S_.patterns[m_pat_];
        std::vector<int32_t> uf(nc);
        for (int j = 0; j < nc; ++j) uf[j] = j;
        auto find = [&](int32_t a) {
            while (uf[a] != a) a = uf[a] = uf[uf[a]];
            return a;
        };
        for (int64_t r = 0; r < P.nrows; ++r) {
            if (p_indptr_[r + 1] == p_indptr_[r]) continue;
            const int32_t a = find(p_indices_[p_indptr_[r]]);
            auto join = [&](int64_t row) {
                for (int32_t q = p_indptr_[row]; q < p_indptr_[row + 1]; ++q) {
                    const int32_t b = find(p_indices_[q]);
                    if (b != a) uf[b] = a;
                }
            };
            join(r);
            for (int32_t q = P.h_indptr[r]; q < P.h_indptr[r + 1]; ++q) join(P.h_indices[q]);
        }
        std::vector<int32_t> cmin(nc, nc), cmax(nc, -1), e_lo(nc), e_hi(nc);
        for (int j = 0; j < nc; ++j) {
            const int32_t c = find(j);
            cmin[c] = std::min(cmin[c], (int32_t)j);
            cmax[c] = std::max(cmax[c], (int32_t)j);
        }
        int ew = 64;
        for (int j = 0; j < nc; ++j) {
            const int32_t c = find(j);
            e_lo[j] = cmin[c] & ~63;
            e_hi[j] = cmax[c] + 1;
            ew = std::max(ew, e_hi[j] - e_lo[j]);
        }
        D.e_lo = up(e_lo);
        D.e_hi = up(e_hi);
        D.ew = ew;
        if (ew < nc) {
            // belt and braces: every inverse formed so far must be zero outside these ranges
            // (exact zeros: elimination never touches an entry outside a block); else full rows
            unsigned *d_flag = dev_alloc<unsigned>(1);
            HIPCHK(hipMemsetAsync(d_flag, 0, sizeof(unsigned), S_.stream));
            for (const double *inv : einv_owned_)
                launch_einv_outside(S_.stream, inv, nc, D.e_lo, D.e_hi, d_flag);
            unsigned bad = 0;
            HIPCHK(hipMemcpyAsync(&bad, d_flag, sizeof bad, hipMemcpyDeviceToHost, S_.stream));
            HIPCHK(hipStreamSynchronize(S_.stream));
            HIPCHK(hipFree(d_flag));
            if (bad) {
                std::fprintf(stderr, "[kkt] coarse inverse: entries outside the diagonal blocks of its "
                             "structure -- the tile program applies full rows\n");
                std::fill(e_lo.begin(), e_lo.end(), 0);
                std::fill(e_hi.begin(), e_hi.end(), nc);
                D.e_lo = up(e_lo);
                D.e_hi = up(e_hi);
                D.ew = nc;
            }
        }
        if (S_.opt("verbose"))
            std::fprintf(stderr, "[kkt] coarse inverse: rows of at most %d of %d columns (diagonal "
                         "blocks of P^T A P)\n", ew, nc);
    }
    for (int i = 0; i < 2; ++i) {
        D.cg[i] = dev_alloc<unsigned long long>((size_t)nslots * 2);
        HIPCHK(hipMemset(D.cg[i], 0, D.cg_bytes));
        owned_.push_back(D.cg[i]);
        D.eg[i] = dev_alloc<unsigned long long>((size_t)nc * 2);
        HIPCHK(hipMemset(D.eg[i], 0, D.eg_bytes));
        owned_.push_back(D.eg[i]);
    }
    d_tile_coarse_ = dev_upload(&D, 1);
    owned_.push_back(d_tile_coarse_);
    if (S_.opt("verbose"))
        std::fprintf(stderr, "[kkt] tile sweep program, coarse corrections: %d coarse functions, at most "
                     "%d per tile, %d partial-sum slots\n", nc, jmax, nslots);
    return true;
}

// Steps [k, e) are single-block steps: if they are exactly a run of recorded sweep levels the
// tile form can express, replace them by one TILE step.
bool SchurPC::fuse_tile_run(size_t k, size_t e, std::vector<PcStep> &out) {
    std::vector<const SweepLevel *> run;
    size_t cursor = k;
    while (cursor < e) {
        const SweepLevel *hit = nullptr;
        for (const SweepLevel &lv : sweep_levels_)
            if (lv.first == cursor) hit = &lv;
        if (!hit || !hit->eligible || hit->last > e) return false;
        run.push_back(hit);
        cursor = hit->last;
    }
    if (run.empty()) return false;
    const int its = run[0]->its;
    const bool coarse = run[0]->coarse;
    if (coarse && !tile_coarse_ok_) return false;
    {
        // residency and the dynamic-LDS attribute were checked for tile_lds_checked_ bytes
        const size_t need = coarse ? tile_sweep_lds_bytes(tile_plan_.nk_pad, its, h_tile_coarse_.nc,
                                                          h_tile_coarse_.nslots, h_tile_coarse_.jmax,
                                                          h_tile_coarse_.cache_lists
                                                              ? h_tile_coarse_.nr_max : 0,
                                                          h_tile_coarse_.cache_einv
                                                              ? h_tile_coarse_.nown : 0,
                                                          h_tile_coarse_.ew)
                                   : tile_sweep_lds_bytes(tile_plan_.nk_pad, its);
        if (need > tile_lds_checked_) {
            if (need > 150 * 1024 ||
                tile_plan_.ntiles > tile_sweep_max_tiles(tile_plan_.W, tile_plan_.rpt,
                                                         tile_plan_.threads, need, tile_plan_.hslots,
                                                         coarse))
                return false;
            tile_lds_checked_ = need;
        }
    }
    std::vector<TileLevel> levels;
    for (size_t i = 0; i < run.size(); ++i) {
        const SweepLevel &lv = *run[i];
        if (lv.its != its || lv.coarse != coarse ||
            (int)lv.coef.size() != (coarse ? its : its - 1))
            return false;
        TileLevel L = lv.lev;
        L.prev_in_lds = 0;
        // operands read from plain memory must not be produced inside this launch
        for (size_t j = 0; j < run.size(); ++j) {
            const TileLevel &o = run[j]->lev;
            if (L.bin == o.out || L.dinv == o.out) return false;
            if (j != i && o.bout && (L.bin == o.bout)) return false;
            if (L.n_upd > 0 && L.x_prev == o.out) {
                if (j + 1 != i) return false;       // only the previous level's result is on chip
                L.prev_in_lds = 1;
            }
            if (L.n_upd > 0 && o.bout && L.x_prev == o.bout) return false;
        }
        // identical coefficient tables share one device copy
        TileCoef *d_coef = nullptr;
        for (size_t j = 0; j < i && !d_coef; ++j)
            if (run[j]->coef.size() == lv.coef.size() &&
                std::memcmp(run[j]->coef.data(), lv.coef.data(), lv.coef.size() * sizeof(TileCoef)) == 0)
                d_coef = const_cast<TileCoef *>(levels[j].coef);
        if (!d_coef) {
            d_coef = dev_upload(lv.coef.data(), lv.coef.size());
            tile_owned_.push_back(d_coef);
        }
        L.coef = d_coef;
        levels.push_back(L);
    }
    // hand-offs of a level, at most (two-grid levels: per cycle one after the correction, one in
    // front of the residual, those of the sweeps; one at the level's end)
    const int per_launch = coarse ? coarse_cycles_ * (its / std::max(1, tile_plan_.depth) + 3) + 2
                                  : its / std::max(1, tile_plan_.depth) + 2;
    std::vector<const double *> einvs;
    for (const SweepLevel *lv : run) einvs.push_back(lv->einv);
    auto tile_step = [&](const TileLevel *lv, int n, int nphases, bool fused) {
        PcStep s;
        s.kind = PcStep::TILE;
        s.fused = fused;
        s.coarse = coarse;
        if (coarse) {
            s.d_einv = dev_upload(einvs.data(), einvs.size());
            tile_owned_.push_back((void *)s.d_einv);
            s.cepoch0 = tile_cepoch_cursor_;
            tile_cepoch_cursor_ += (uint32_t)(n * coarse_cycles_);
        }
        s.d_levels = dev_upload(lv, (size_t)n);
        s.nlevels = n;
        s.its = its;
        s.nphases = nphases;
        s.epoch0 = tile_epoch_cursor_;
        s.clear = !tile_cleared_;
        tile_cleared_ = true;
        tile_epoch_cursor_ += (uint32_t)(n * per_launch);
        out.push_back(s);
    };
    int max_terms = 0;
    for (const TileLevel &L : levels) max_terms = std::max(max_terms, (int)L.n_upd);
    if (coarse && !tile_sweep_fuses_update(tile_plan_.W, max_terms)) return false;
    if (tile_sweep_fuses_update(tile_plan_.W, max_terms) &&
        !(tile_plan_.W > 9 && S_.opt("tile_unfused") && !coarse)) {
        for (size_t q = k; q < e; ++q) (void)hipFree(steps_[q].rows.d_ops);
        tile_step(levels.data(), (int)levels.size(), (int)(e - k), true);
        return true;
    }
    // Wide rows: the kernel has no registers for the level update.  It stays the plain launch it
    // was (it also leaves p_1, which the tile launch recomputes from the same right-hand side),
    // and every level's Chebyshev steps become a tile launch of their own; the hand-off tags
    // keep counting through the launches of an application.
    for (size_t i = 0; i < run.size(); ++i) {
        const SweepLevel &lv = *run[i];
        TileLevel L = levels[i];
        size_t first_cheb = lv.first;
        if (L.n_upd > 0) {
            out.push_back(steps_[lv.first]);         // the update, as a plain single-block step
            first_cheb = lv.first + 1;
            L.n_upd = 0;
            L.prev_in_lds = 0;
            L.bin = lv.b_after;
            L.bout = nullptr;
        }
        for (size_t q = first_cheb; q < lv.last; ++q) (void)hipFree(steps_[q].rows.d_ops);
        tile_step(&L, 1, (int)(lv.last - first_cheb), false);
    }
    return true;
}

// Replace every run of >= 4 consecutive single-block steps (the time sweeps) by one
// persistent launch: the tile form (tile_kernels.hip: `depth` steps per hand-off) where it
// fits, else a row program with neighbour synchronisation (kernels.hip, pc_row_program*).
void SchurPC::fuse_programs() {
    if (!use_programs_) return;
    if (!d_err_) {
        d_err_ = dev_alloc<unsigned>(64 + 16 * 1024);   // word 0: error bits; rest: diagnostics
        HIPCHK(hipMemset(d_err_, 0, (64 + 16 * 1024) * sizeof(unsigned)));
        owned_.push_back(d_err_);
    }
    const char *pm_all = S_.opt("prog_mode");
    const bool tile_forced = pm_all && pm_all[0] == 't';
    const bool tile_wanted = tile_forced || !pm_all || pm_all[0] == 'a';
    const bool use_tiles = tile_wanted && prepare_tiles();
    tile_epoch_cursor_ = 0;
    tile_cepoch_cursor_ = 0;
    tile_cleared_ = false;
    if (!legacy_tried_) {
        legacy_tried_ = true;
        legacy_ok_ = setup_row_programs();
    }
    if (!use_tiles && !legacy_ok_) {
        use_programs_ = false;
        return;
    }
    // output pointer the next phase may gather from
    auto gather_out = [](const RowOp &op) -> int64_t {
        return (op.mode == EPI_LIN && op.y2.base >= 0) ? op.y2.off : op.y.off;
    };
    std::vector<PcStep> out;
    // one row program for the single-block steps [q0, q1)
    auto emit_prog = [&](size_t q0, size_t q1) {
        std::vector<RowOp> ops;
        for (size_t q = q0; q < q1; ++q) {
            ops.push_back(steps_[q].rows.h_op);
            (void)hipFree(steps_[q].rows.d_ops);
        }
        PcStep s;
        s.kind = PcStep::PROG;
        s.rows.d_ops = dev_upload(ops.data(), ops.size());
        s.nphases = (int)ops.size();
        {
            // compact records: Chebyshev steps that only continue the previous phase's
            // solve are marked STEP (kernels.hpp, PhaseLite)
            std::vector<PhaseLite> lite(ops.size());
            const char *ps = S_.opt("prog_steps");
            const bool use_steps = !(ps && ps[0] == '0');
            auto same = [](const VRef &a, const VRef &b) {
                return a.base == b.base && (a.base < 0 || a.off == b.off);
            };
            for (size_t e2 = 0; e2 < ops.size(); ++e2) {
                const RowOp &op = ops[e2];
                PhaseLite L{};
                L.kind = 0;
                if (use_steps && e2 >= 1) {
                    const RowOp &pv = ops[e2 - 1];
                    const bool has_pkm1 = op.pkm1.base >= 0;
                    bool stepk = op.mode == EPI_CHEB && pv.mode == EPI_CHEB &&
                                 op.nterms == 1 && pv.nterms == 1 &&
                                 op.t[0].vals == pv.t[0].vals && op.col == pv.col &&
                                 op.rowmask == pv.rowmask && op.dinv == pv.dinv &&
                                 op.nslices == pv.nslices && op.nrows == pv.nrows &&
                                 same(op.b, pv.b) && op.b.base == 0 && op.y.base == 0 &&
                                 op.pk.base == 0 && same(op.pk, pv.y) &&
                                 same(op.t[0].x, pv.y) &&
                                 (!has_pkm1 || (pv.pk.base == 0 && same(op.pkm1, pv.pk)));
                    if (stepk) {
                        L.kind = 1;
                        L.flags = has_pkm1 ? 1u : 0u;
                        L.y = (uint64_t)op.y.off;
                        L.c1 = op.c1;
                        L.c2 = op.c2;
                        L.c3 = op.c3;
                        L.post1 = op.post1;
                        L.post2 = op.post2;
                    }
                }
                lite[e2] = L;
            }
            s.d_lite = dev_upload(lite.data(), lite.size());
        }
        s.granule = prog_granule_;
        s.gmode = prog_mode_;
        out.push_back(s);
    };
    size_t k = 0;
    while (k < steps_.size()) {
        size_t e = k;
        bool has_coarse = false;
        // (the coarse corrections of two-grid levels sit between the single-block steps)
        while (e < steps_.size() && ((steps_[e].kind == PcStep::ROWS && steps_[e].rows.nops == 1) ||
                                     steps_[e].kind == PcStep::COARSE)) {
            has_coarse = has_coarse || steps_[e].kind == PcStep::COARSE;
            ++e;
        }
        if (e == k) {
            out.push_back(steps_[k++]);
            continue;
        }
        if (e - k >= 4 && use_tiles && fuse_tile_run(k, e, out)) {
            k = e;
            continue;
        }
        if (has_coarse) {      // two-grid levels outside the tile form stay plain launches
            for (size_t i = k; i < e; ++i) out.push_back(steps_[i]);
            k = e;
            continue;
        }
        // row programs; the data-flow form needs phase ph >= 1 to gather exactly what phase
        // ph - 1 produced, so a run is cut where that chain breaks
        size_t q = k;
        while (q < e) {
            size_t q1 = q + 1;
            while (q1 < e) {
                if (prog_granule_ && legacy_ok_) {
                    const RowOp &op = steps_[q1].rows.h_op, &prev = steps_[q1 - 1].rows.h_op;
                    bool chain = op.nterms > 0;
                    for (int t = 0; t < op.nterms; ++t)
                        chain &= op.t[t].x.base == 0 && op.t[t].x.off == gather_out(prev);
                    if (!chain) break;
                }
                ++q1;
            }
            if (q1 - q >= 4 && legacy_ok_)
                emit_prog(q, q1);
            else
                for (size_t i = q; i < q1; ++i) out.push_back(steps_[i]);
            q = q1;
        }
        k = e;
    }
    steps_.swap(out);
    // what runs, for the caller's records (kkt_info)
    int form = 0;
    for (const PcStep &s : steps_) {
        if (s.kind == PcStep::TILE) form = 3;
        if (s.kind == PcStep::PROG && form < 3) form = std::max(form, s.granule ? 2 : 1);
    }
    S_.info.sweep_form = form;
    S_.info.sweep_tiles = form == 3 ? tile_plan_.ntiles : 0;
    S_.info.sweep_threads = form == 3 ? tile_plan_.threads : 0;
    S_.info.sweep_depth = form == 3 ? tile_plan_.depth : 0;
    S_.info.sweep_row_slots = form == 3 ? tile_plan_.rpt : 0;
}

void SchurPC::debug_read(unsigned long long *out, int n) {
    if (!d_err_) return;
    HIPCHK(hipStreamSynchronize(S_.stream));
    HIPCHK(hipMemcpy(out, d_err_ + 64, (size_t)n * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemset(d_err_ + 64, 0, (size_t)n * 8));
}

// One application as plain launches with an event pair around every persistent program.
void SchurPC::time_programs(float *ms, int *launches, int64_t *phases) {
    hipStream_t st = S_.stream;
    *ms = 0.f;
    *launches = 0;
    *phases = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> evs;
    size_t k = 0;
    while (k < steps_.size()) {
        if (steps_[k].kind == PcStep::PROG || steps_[k].kind == PcStep::TILE) {
            hipEvent_t a, b;
            HIPCHK(hipEventCreate(&a));
            HIPCHK(hipEventCreate(&b));
            HIPCHK(hipEventRecord(a, st));
            replay(k, k + 1);
            HIPCHK(hipEventRecord(b, st));
            evs.push_back({a, b});
            *launches += 1;
            *phases += steps_[k].nphases;
        } else {
            replay(k, k + 1);
        }
        ++k;
    }
    HIPCHK(hipStreamSynchronize(st));
    for (auto &e : evs) {
        float t = 0.f;
        HIPCHK(hipEventElapsedTime(&t, e.first, e.second));
        *ms += t;
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
}

void PcBase::time_stages(kkt_pc_stage_times *out) {
    // no itemisation: the whole application between two events
    *out = kkt_pc_stage_times{};
    fail(KKT_ERR_STATE, "kkt_time_pc_stages needs the built-in block-Schur preconditioner");
}

// One application step by step, an event after every step; a step's time goes to the sweeps
// (persistent programs, or the single-block steps they replace), to the hand-offs between ranks,
// or to the batched steps over all time levels.
void SchurPC::time_stages(kkt_pc_stage_times *out) {
    hipStream_t st = S_.stream;
    *out = kkt_pc_stage_times{};
    std::vector<hipEvent_t> ev(steps_.size() + 1);
    for (auto &e : ev) HIPCHK(hipEventCreate(&e));
    HIPCHK(hipEventRecord(ev[0], st));
    for (size_t k = 0; k < steps_.size(); ++k) {
        replay(k, k + 1);
        // (side-lane steps are rare -- option "lanes" -- and are charged to the step that waits)
        HIPCHK(hipEventRecord(ev[k + 1], st));
    }
    HIPCHK(hipStreamSynchronize(st));
    for (size_t k = 0; k < steps_.size(); ++k) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, ev[k], ev[k + 1]));
        const PcStep &s = steps_[k];
        out->total_ms += ms;
        if (s.kind == PcStep::PROG || s.kind == PcStep::TILE) {
            out->sweeps_ms += ms;
            out->sweep_launches += 1;
            out->sweep_phases += s.nphases;
        } else if (s.kind == PcStep::ROWS && s.rows.nops == 1 && n_ > 1) {
            out->sweeps_ms += ms;          // a sweep step as a plain launch
            out->sweep_launches += 1;
            out->sweep_phases += 1;
        } else if (s.kind == PcStep::COMM) {
            out->comm_ms += ms;
            out->comm_steps += 1;
        } else {
            out->batched_ms += ms;
            if (s.kind == PcStep::ROWS || s.kind == PcStep::TIME || s.kind == PcStep::ROWS_IL)
                out->batched_launches += 1;
        }
    }
    for (auto &e : ev) (void)hipEventDestroy(e);
}

bool SchurPC::timed_out(std::string *why) {
    if (!d_err_) return false;
    unsigned e = 0;
    HIPCHK(hipMemcpyAsync(&e, d_err_, sizeof e, hipMemcpyDeviceToHost, S_.stream));
    HIPCHK(hipStreamSynchronize(S_.stream));
    if (!e) return false;
    unsigned rec[40] = {0};
    HIPCHK(hipMemcpy(rec, d_err_, sizeof rec, hipMemcpyDeviceToHost));
    HIPCHK(hipMemset(d_err_, 0, sizeof rec));
    std::string msg = "persistent sweep kernel timed out waiting for a neighbour workgroup";
    if (e & 4u)
        msg += " (tile form: tile " + std::to_string(rec[8]) + ", hand-off " +
               std::to_string(rec[9]) + ", local row " + std::to_string(rec[10]) +
               ", global row " + std::to_string(rec[11]) + ", tags seen new " +
               std::to_string(rec[12]) + "/" + std::to_string(rec[13]) + " old " +
               std::to_string(rec[14]) + "/" + std::to_string(rec[15]) +
               (rec[16] ? ", both iterates)" : ", newest iterate only)");
    if (e & 8u)
        msg = "persistent sweep kernel abandoned at entry: its workgroups are not co-resident (" +
              std::to_string(rec[17]) + " of " + std::to_string(rec[18]) +
              " tiles checked in within the bounded wait; tile form)";
    if (e & 2u)
        msg += " (data-flow form: workgroup " + std::to_string(rec[24]) + " wave " +
               std::to_string(rec[25]) + " lane " + std::to_string(rec[26]) + ", phase " +
               std::to_string(rec[27]) + ", expected tag " + std::to_string(rec[28]) +
               ", seen " + std::to_string(rec[29]) + "/" + std::to_string(rec[30]) +
               " at column " + std::to_string(rec[31]) + " of slice " + std::to_string(rec[32]) + ")";
    if (e & 1u) msg += " (counter form)";
    if (why) *why = msg;
    return true;
}

void SchurPC::check() {
    std::string why;
    if (timed_out(&why)) fail(KKT_ERR_HIP, why);
}

// The same steps as plain launches (bit-identical arithmetic, tests/test_gpu_parity.py).
bool SchurPC::fallback_plain() {
    if (!use_programs_) return false;
    use_programs_ = false;
    values_changed();
    return true;
}

const double *SchurPC::block_vals(int q, int i, int j) const {
    auto it = S_.blocks.find(std::make_tuple(q, i, j));
    if (it == S_.blocks.end())
        fail(KKT_ERR_STATE, "preconditioner needs block (" + std::to_string(q) + "," +
                                std::to_string(i) + "," + std::to_string(j) + ")");
    const ValueArray &va = S_.values[it->second.va];
    if (va.pattern != m_pat_)
        fail(KKT_ERR_STATE, "preconditioner blocks must share the mass matrix's sparsity");
    return va.d_vals;
}

void SchurPC::build() {
    hipStream_t st = S_.stream;
    bc_set_ = bc_idx_.empty() ? -1 : S_.add_bc_set(nx_, (int64_t)bc_idx_.size(), bc_idx_.data());
    mask_ = bc_set_ >= 0 ? S_.bc_sets[bc_set_].d_mask : nullptr;
    m_pat_ = S_.find_or_add_pattern(nx_, nx_, m_indptr_.data(), m_indices_.data());
    const Pattern &P = S_.patterns[m_pat_];
    {
        double *d_csr = dev_upload(m_values_.data(), m_values_.size());
        m_vals_ = dev_alloc<double>(P.npadded);
        owned_.push_back(m_vals_);
        launch_csr_to_sell(st, d_csr, P.d_sell2csr, m_vals_, P.npadded);
        if (mask_) launch_mask_columns(st, m_vals_, P.d_col, mask_, P.npadded);
        HIPCHK(hipStreamSynchronize(st));
        HIPCHK(hipFree(d_csr));
    }
    m_dinv_ = dev_alloc<double>(nx_);
    owned_.push_back(m_dinv_);
    launch_extract_dinv(st, P.d_col, P.d_slice_off, m_vals_, mask_, m_dinv_, (int)nx_, P.nslices,
                        P.R, P.d_perm);
    {
        std::vector<int32_t> z(P.nslices + 1, 0);
        zero_off_ = dev_upload(z.data(), z.size());
        owned_.push_back(zero_off_);
    }
    auto vec = [&](int64_t blocks) {
        double *p = dev_alloc<double>(blocks * nx_ + 32);
        HIPCHK(hipMemsetAsync(p, 0, (blocks * nx_ + 32) * sizeof(double), st));
        owned_.push_back(p);
        return p;
    };
    const int nl = hi_ - lo_;
    in_ = vec(2 * nl);
    out_ = vec(2 * nl);
    B_ = vec(nl);
    T_ = vec(nl);
    for (int k = 0; k < 3; ++k) P_[k] = vec(nl);
    h_u0_ = vec(1);
    h_u1_ = vec(1);
    h_t_ = vec(1);
    if (coarse_cycles_ > 0) {
        R_ = vec(1);
        build_coarse();
    }
    values_changed();
}

// tpos[position of (r, c)] = position of (c, r) in the SELL value arrays of the preconditioner's
// structure; null when some entry has no transposed partner (not a finite-element structure)
const int32_t *SchurPC::transpose_positions() {
    if (tpos_tried_) return d_tpos_;
    tpos_tried_ = true;
    const Pattern &P = S_.patterns[m_pat_];
    if (P.nrows != P.ncols) return nullptr;
    std::vector<int32_t> t((size_t)P.npadded, -1);
    for (int64_t r = 0; r < P.nrows; ++r)
        for (int32_t q = P.h_indptr[r]; q < P.h_indptr[r + 1]; ++q) {
            const int32_t c = P.h_indices[q];
            const int32_t *b = P.h_indices.data() + P.h_indptr[c];
            const int32_t *e = P.h_indices.data() + P.h_indptr[c + 1];
            const int32_t *hit = std::lower_bound(b, e, (int32_t)r);
            if (hit == e || *hit != (int32_t)r) return nullptr;
            t[(size_t)P.sell_index(r, q - P.h_indptr[r])] =
                (int32_t)P.sell_index(c, (int)(hit - b));
        }
    d_tpos_ = dev_upload(t.data(), t.size());
    owned_.push_back(d_tpos_);
    return d_tpos_;
}

// ---- two-grid form of the sub-solves: coarse space on the device, Galerkin inverses
void SchurPC::build_coarse() {
    const int nc = (int)d_.n_coarse;
    // P^T by a counting sort over the columns (entries of a column in ascending row order: the
    // restriction sums them in that order)
    pt_indptr_.assign(nc + 1, 0);
    for (int32_t c : p_indices_) pt_indptr_[c + 1]++;
    for (int j = 0; j < nc; ++j) pt_indptr_[j + 1] += pt_indptr_[j];
    pt_indices_.resize(p_indices_.size());
    pt_values_.resize(p_indices_.size());
    std::vector<int32_t> cur(pt_indptr_.begin(), pt_indptr_.end() - 1);
    for (int64_t r = 0; r < nx_; ++r)
        for (int32_t q = p_indptr_[r]; q < p_indptr_[r + 1]; ++q) {
            const int32_t at = cur[p_indices_[q]]++;
            pt_indices_[at] = (int32_t)r;
            pt_values_[at] = p_values_[q];
        }
    auto up = [&](const auto &v) {
        auto *p = dev_upload(v.data(), v.size());
        owned_.push_back((void *)p);
        return p;
    };
    coarse_.nc = nc;
    coarse_.p_ip = up(p_indptr_);
    coarse_.p_ix = up(p_indices_);
    coarse_.p_v = up(p_values_);
    coarse_.pt_ip = up(pt_indptr_);
    coarse_.pt_ix = up(pt_indices_);
    coarse_.pt_v = up(pt_values_);
    coarse_.rc = dev_alloc<double>(nc);
    coarse_.ec = dev_alloc<double>(nc);
    owned_.push_back(coarse_.rc);
    owned_.push_back(coarse_.ec);
}

// E = P^T A P column by column with the kernels the sweeps use (x = column k of P, y = A x with
// the boundary rows masked, E(:, k) = P^T y: fixed summation orders, so every rank and every run
// forms the same matrix), inverted on the device by Gauss-Jordan with partial pivoting.
double *SchurPC::coarse_inverse(const double *vals) {
    const Pattern &P = S_.patterns[m_pat_];
    hipStream_t st = S_.stream;
    const int nc = coarse_.nc;
    double *x = dev_alloc<double>(nx_ + 32), *y = dev_alloc<double>(nx_ + 32);
    double *E = dev_alloc<double>((size_t)nc * nc);
    auto vref = [](const double *q) { return q ? VRef{(int64_t)(uintptr_t)q, 0, 0} : VRef{0, -1, 0}; };
    RowOp op{};
    op.col = P.d_col;
    op.perm = P.d_perm;
    op.slice_off = P.d_slice_off;
    op.uniform_w = P.uniform_w;
    op.nrows = (int32_t)nx_;
    op.nslices = P.nslices;
    op.nterms = 1;
    op.mode = EPI_LIN;
    op.t[0].vals = vals;
    op.t[0].x = vref(x);
    op.y = vref(y);
    op.y2 = op.yin = op.z = op.mx = op.b = op.pk = op.pkm1 = vref(nullptr);
    op.ca = 1.0;
    op.rowmask = mask_;
    RowOp *d_op = dev_upload(&op, 1);
    const Bases B{{nullptr, nullptr, nullptr, nullptr}};
    for (int k = 0; k < nc; ++k) {
        launch_coarse_column(st, coarse_, k, x, nx_);
        launch_rowops(st, d_op, 1, P.nslices, P.R, B, 1, P.uniform_w);
        launch_coarse_restrict(st, coarse_, y, E + k, nc);     // column k of the row-major E
    }
    double *d_inv = dev_alloc<double>((size_t)nc * nc);
    int *d_piv = dev_alloc<int>(1);
    double *d_colbuf = dev_alloc<double>(nc + 1);     // multipliers; slot nc: the pivot
    unsigned *d_flag = dev_alloc<unsigned>(1);
    HIPCHK(hipMemsetAsync(d_flag, 0, sizeof(unsigned), st));
    launch_dense_inverse(st, E, d_inv, nc, d_piv, d_colbuf, d_flag);
    unsigned singular = 0;
    HIPCHK(hipMemcpyAsync(&singular, d_flag, sizeof singular, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    (void)hipFree(x);
    (void)hipFree(y);
    (void)hipFree(E);
    (void)hipFree(d_op);
    (void)hipFree(d_piv);
    (void)hipFree(d_colbuf);
    (void)hipFree(d_flag);
    if (singular) {
        (void)hipFree(d_inv);
        fail(KKT_ERR_STATE, "coarse matrix P^T A P is singular (a coarse function without support on "
                            "free rows, or dependent coarse functions)");
    }
    einv_owned_.push_back(d_inv);
    return d_inv;
}

void SchurPC::emit_coarse(const double *r, const double *x_in, double *x_out, const double *einv) {
    PcStep s;
    s.kind = PcStep::COARSE;
    s.cr = r;
    s.x = x_in;
    s.y = x_out;
    s.einv = einv;
    s.nx = nx_;
    s.lane = cur_lane_;
    steps_.push_back(s);
}

// coefficients (c1, c2, c3) of the Chebyshev steps 2 .. its (what emit_solves generates)
static std::vector<TileCoef> cheb_coefficients(int its, double emin, double emax, double eimag) {
    std::vector<TileCoef> out;
    const double scale = 2.0 / (emax + emin);
    const double alpha = 1.0 - scale * emin;
    const double mu = 1.0 / alpha, omegaprod = 2.0 / alpha;
    double c_km1 = 1.0, c_k = mu;
    const double ell_d = 0.5 * (emax + emin), ell_a = 0.5 * (emax - emin);
    const double ell_c2 = ell_a * ell_a - eimag * eimag;
    double ell_alpha = 1.0 / ell_d;
    for (int step = 2; step <= its; ++step) {
        if (eimag > 0.0) {
            ell_alpha = 1.0 / (ell_d - (step == 2 ? 0.5 : 0.25) * ell_c2 * ell_alpha);
            const double beta = ell_d * ell_alpha - 1.0;
            out.push_back(TileCoef{-beta, 1.0 + beta, ell_alpha});
        } else {
            const double c_kp1 = 2.0 * mu * c_k - c_km1;
            const double omega = omegaprod * c_k / c_kp1;
            out.push_back(TileCoef{1.0 - omega, omega, scale * omega});
            c_km1 = c_k;
            c_k = c_kp1;
        }
    }
    return out;
}

// One sub-solve in its two-grid form: [update of the right-hand side;] cycles x [Galerkin
// correction of the current iterate; `its` smoothing sweeps from it].
void SchurPC::emit_coarse_solve(const Lin *upd, const Solve &sv, const Mat &F) {
    const int its = std::max(1, schur_its_);
    SweepLevel lv;
    lv.first = steps_.size();
    if (upd) {
        Lin u = *upd;
        u.y2 = nullptr;
        emit_lin({u});
    }
    const double scale = 2.0 / (F.emax + F.emin);
    const std::vector<TileCoef> coef = cheb_coefficients(its, F.emin, F.emax, F.eimag);
    {
        // what the tile form runs as one two-grid level
        lv.its = its;
        lv.coarse = true;
        lv.einv = F.einv;
        lv.coef.push_back(TileCoef{0.0, 1.0, scale});
        lv.coef.insert(lv.coef.end(), coef.begin(), coef.end());
        TileLevel &L = lv.lev;
        L.vals = F.vals;
        L.dinv = F.dinv;
        L.out = sv.out;
        L.p1_scale = scale;
        L.post1 = sv.post1;
        L.post2 = sv.post2;
        bool ok = true;
        if (upd) {
            ok = !upd->terms.empty() && upd->terms.size() <= 2 && upd->cz == 0.0 && !upd->z &&
                 upd->yin != nullptr;
            for (const Term &t : upd->terms) ok = ok && t.x == upd->terms[0].x;
            if (ok) {
                L.bin = upd->yin;
                L.bout = upd->y == upd->yin ? nullptr : upd->y;
                lv.b_after = upd->y;
                L.x_prev = upd->terms[0].x;
                L.n_upd = (int32_t)upd->terms.size();
                for (size_t t = 0; t < upd->terms.size(); ++t) L.upd_vals[t] = upd->terms[t].vals;
                L.ca = upd->ca;
                L.cy = upd->cy;
            }
        } else {
            L.bin = sv.b;
            L.bout = nullptr;
            L.n_upd = 0;
        }
        lv.eligible = ok;
    }
    double *p0 = P_[2];
    const double *xcur = nullptr;
    for (int c = 0; c < coarse_cycles_; ++c) {
        const bool last_cycle = c + 1 == coarse_cycles_;
        const double *r = sv.b;
        if (c > 0) {
            // r = b - F x
            emit_lin({Lin{{Term{F.vals, xcur}}, R_, -1.0, 0.0, 1.0, nullptr, sv.b}});
            r = R_;
        }
        emit_coarse(r, xcur, p0, F.einv);
        auto target = [&](int step) -> double * {
            return (last_cycle && step == its) ? sv.out : P_[(step - 1) % 3];
        };
        for (int step = 1; step <= its; ++step) {
            const bool last = last_cycle && step == its;
            const double *pk = step == 1 ? p0 : target(step - 1);
            const double *pkm1 = step == 1 ? nullptr : (step == 2 ? p0 : target(step - 2));
            const TileCoef k = step == 1 ? TileCoef{0.0, 1.0, scale} : coef[step - 2];
            emit_cheb({Cheb{F.vals, F.dinv, sv.b, pk, pkm1, target(step), k.c1, k.c2, k.c3,
                            last ? sv.post1 : 1.0, last ? sv.post2 : 1.0}});
        }
        xcur = target(its);
    }
    lv.last = steps_.size();
    sweep_levels_.push_back(lv);
}

// base + c * M with bc rows/cols of `assemble(form, bcs=...)`, and its Jacobi diagonal
SchurPC::Mat SchurPC::schur_matrix(const double *base_vals, double c) {
    uint64_t bits;
    std::memcpy(&bits, &c, sizeof bits);
    auto key = std::make_pair(base_vals, bits);
    auto it = mats_.find(key);
    if (it != mats_.end()) return it->second;
    const Pattern &P = S_.patterns[m_pat_];
    hipStream_t st = S_.stream;
    Mat m;
    m.vals = dev_alloc<double>(P.npadded);
    m.dinv = dev_alloc<double>(nx_);
    launch_vals_axpy(st, m.vals, base_vals, c, m_vals_, P.npadded);
    if (mask_) launch_mask_columns(st, m.vals, P.d_col, mask_, P.npadded);
    launch_extract_dinv(st, P.d_col, P.d_slice_off, m.vals, mask_, m.dinv, (int)nx_, P.nslices,
                        P.R, P.d_perm);
    // an earlier matrix with the same shift and the same values (mode G stores one copy per time
    // level of a time-invariant operator) shares its spectrum estimate and its coarse inverse
    const Mat *twin = nullptr;
    if (d_.schur_emin <= 0 || coarse_cycles_ > 0) {
        for (auto &kv : mats_) {
            if (kv.first.second != bits || kv.second.emax <= 0.0) continue;
            unsigned *d_flag = dev_alloc<unsigned>(1);
            HIPCHK(hipMemsetAsync(d_flag, 0, sizeof(unsigned), st));
            launch_vals_differ(st, m.vals, kv.second.vals, P.npadded, d_flag);
            unsigned differ = 1;
            HIPCHK(hipMemcpyAsync(&differ, d_flag, sizeof differ, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            HIPCHK(hipFree(d_flag));
            if (!differ) {
                twin = &kv.second;
                break;
            }
        }
    }
    if (d_.schur_emin > 0) {
        m.emin = d_.schur_emin;
        m.emax = d_.schur_emax;
        m.eimag = d_.schur_eimag > 0 ? d_.schur_eimag : 0.0;
    } else if (twin) {
        // Interval from the matrix itself; the first and last levels carry other shifts
        // (control.py:2241-2327) and get their own, wider, intervals.
        m.emin = twin->emin;
        m.emax = twin->emax;
        m.eimag = twin->eimag;
    } else {
        // Blocks with a convection term are not symmetric: the interval comes from the
        // symmetric part H = (A + A^T) / 2 (Bendixson: Re lambda lies in the spectrum of
        // D^-1/2 H D^-1/2) and the ellipse's imaginary semi-axis from the spectral radius of
        // the skew part (|Im lambda| <= rho(D^-1/2 (A - A^T) / 2 D^-1/2)).
        double *hv = nullptr, *sv2 = nullptr;
        unsigned nonsym = 0;
        const int32_t *tpos = transpose_positions();
        if (tpos) {
            hv = dev_alloc<double>(P.npadded);
            sv2 = dev_alloc<double>(P.npadded);
            unsigned *d_flag = dev_alloc<unsigned>(1);
            HIPCHK(hipMemsetAsync(d_flag, 0, sizeof(unsigned), st));
            launch_vals_sym_skew(st, m.vals, tpos, hv, sv2, P.npadded, d_flag);
            HIPCHK(hipMemcpyAsync(&nonsym, d_flag, sizeof nonsym, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            HIPCHK(hipFree(d_flag));
        }
        const Spectrum sp = jacobi_spectrum(S_, m_pat_, nonsym ? hv : m.vals, m.dinv, mask_, 400);
        spectrum_steps_ += sp.steps;
        if (!(sp.emin > 0.0) || !(sp.emax > sp.emin))
            fail(KKT_ERR_STATE, "sub-solve matrix is not positive definite: no Chebyshev interval");
        m.emin = 0.85 * sp.emin;      // Ritz values lie inside the spectrum
        m.emax = 1.05 * sp.emax;
        if (nonsym) {
            int steps = 0;
            m.eimag = 1.1 * jacobi_skew_radius(S_, m_pat_, sv2, m.dinv, mask_, 40, &steps);
            spectrum_steps_ += 2 * steps;
        }
        if (hv) (void)hipFree(hv);
        if (sv2) (void)hipFree(sv2);
        // two-grid form: the sweeps smooth -- they cover the upper part of the spectrum, the
        // coarse space the rest (emax / 30: measured optimum for 8 sweeps at coarse cells of 8
        // to 16 mesh widths, scripts/proto_subsolve.py)
        if (coarse_cycles_ > 0) m.emin = std::max(m.emin, m.emax / 30.0);
    }
    if (coarse_cycles_ > 0) m.einv = twin && twin->einv ? twin->einv : coarse_inverse(m.vals);
    mats_[key] = m;
    return m;
}

// Degree of the sub-solves: as given, or 1.6 sqrt(kappa) of a typical (interior-level) matrix --
// the measured minimum for GMRES(10) to converge on 256^2 P1 is 1.46 sqrt(kappa).
// Crank-Nicolson: 2.6 sqrt(kappa) -- its preconditioner wraps the sweeps in the running sums
// T_2^-1 / T_2 over all time levels (control.py:2053, 2118), which accumulate the error of the
// inexact sub-solves: with 1.6 sqrt(kappa) GMRES(10) stalls on 256^2 x 64, with 2.5 it needs 15
// iterations.
int SchurPC::resolve_its(const Mat &typical) {
    typical_emin_ = typical.emin;
    typical_emax_ = typical.emax;
    if (d_.schur_its >= 0) return d_.schur_its;
    if (coarse_cycles_ > 0) return 8;      // smoothing sweeps per cycle
    const double factor = d_.kind == KKT_PC_INSTATIONARY_CN ? 2.6 : 1.6;
    int its = (int)std::ceil(factor * std::sqrt(typical.emax / typical.emin));
    its = std::max(4, std::min(600, its));
    if (S_.sharded && S_.comm) its = (int)S_.comm->max_host((double)its, S_.stream);
    return its;
}

void SchurPC::push_rows(std::vector<RowOp> &r) {
    const Pattern &P = S_.patterns[m_pat_];
    PcStep s;
    s.kind = PcStep::ROWS;
    s.rows.nops = (int)r.size();
    s.rows.max_slices = P.nslices;
    s.rows.R = P.R;
    s.rows.uniform_w = P.uniform_w;
    s.rows.d_ops = dev_upload(r.data(), r.size());
    const char *ko = S_.opt("kernarg_ops");
    const bool kernarg_ops = ko && ko[0] == '1';
    if (r.size() == 1) {
        s.rows.single = kernarg_ops;
        s.rows.h_op = r[0];
    }
    const char *sr = S_.opt("shared_rows");
    const bool use_shared = !(sr && sr[0] == '0');
    s.rows.shared_matrix = use_shared && r.size() >= 4;
    for (const RowOp &op : r)
        s.rows.shared_matrix = s.rows.shared_matrix && op.nterms == 1 &&
                               op.t[0].vals == r[0].t[0].vals && op.col == r[0].col &&
                               op.rowmask == r[0].rowmask && op.t[0].x.base == 0;
    s.lane = cur_lane_;
    steps_.push_back(s);
}

void SchurPC::emit_lin(const std::vector<Lin> &ops) {
    const Pattern &P = S_.patterns[m_pat_];
    std::vector<RowOp> r;
    for (const Lin &l : ops) {
        if ((int)l.terms.size() > MAX_TERMS) fail(KKT_ERR_STATE, "too many terms");
        RowOp op{};
        op.col = P.d_col;
        op.perm = P.d_perm;
        op.slice_off = l.terms.empty() ? zero_off_ : P.d_slice_off;
        op.uniform_w = l.terms.empty() ? 0 : P.uniform_w;
        op.nrows = (int32_t)nx_;
        op.nslices = P.nslices;
        op.nterms = (int32_t)l.terms.size();
        op.mode = EPI_LIN;
        for (size_t t = 0; t < l.terms.size(); ++t) {
            op.t[t].vals = l.terms[t].vals;
            op.t[t].x = vabs(l.terms[t].x);
        }
        op.y = vabs(l.y);
        op.y2 = vabs(l.y2);
        op.dinv = l.dinv;
        op.c3 = l.c3;
        op.ca = l.ca;
        op.cy = l.cy;
        op.cz = l.cz;
        op.yin = vabs(l.yin);
        op.z = vabs(l.z);
        op.rowmask = mask_;
        op.mx = vabs(nullptr);
        op.b = op.pk = op.pkm1 = vabs(nullptr);
        r.push_back(op);
    }
    push_rows(r);
}

void SchurPC::emit_cheb(const std::vector<Cheb> &ops) {
    const Pattern &P = S_.patterns[m_pat_];
    std::vector<RowOp> r;
    for (const Cheb &c : ops) {
        RowOp op{};
        op.col = P.d_col;
        op.perm = P.d_perm;
        op.slice_off = c.vals ? P.d_slice_off : zero_off_;
        op.uniform_w = c.vals ? P.uniform_w : 0;
        op.nrows = (int32_t)nx_;
        op.nslices = P.nslices;
        op.nterms = c.vals ? 1 : 0;
        op.mode = EPI_CHEB;
        if (c.vals) {
            op.t[0].vals = c.vals;
            op.t[0].x = vabs(c.pk);
        }
        op.y = vabs(c.y);
        op.y2 = vabs(nullptr);
        op.yin = op.z = op.mx = vabs(nullptr);
        op.rowmask = mask_;
        op.b = vabs(c.b);
        op.pk = vabs(c.pk);
        op.pkm1 = vabs(c.pkm1);
        op.dinv = c.dinv;
        op.c1 = c.c1;
        op.c2 = c.c2;
        op.c3 = c.c3;
        op.post1 = c.post1;
        op.post2 = c.post2;
        r.push_back(op);
    }
    push_rows(r);
}

void SchurPC::emit_time(double *y, const double *x, int kind, int n, const double *lo_halo,
                        const double *hi_halo) {
    PcStep s;
    s.kind = PcStep::TIME;
    s.y = y;
    s.x = x;
    s.tkind = kind;
    s.n = n;
    s.nx = nx_;
    s.lo_halo = lo_halo;
    s.hi_halo = hi_halo;
    s.lane = cur_lane_;
    steps_.push_back(s);
}

// KSPSolve_Chebyshev (first kind) + PCJACOBI, zero initial guess, exactly `its` steps
// (options of control.py:1973-1982); its == 0: one Jacobi application (control.py:1984-1991).
void SchurPC::emit_update_and_solve(Lin upd, const Solve &sv, int its, double emin, double emax,
                                    double eimag, const Mat *mat) {
    if (coarse_cycles_ > 0 && mat && mat->einv) {
        emit_coarse_solve(&upd, sv, *mat);
        return;
    }
    if (its == 0) {
        upd.y2 = sv.out;        // Jacobi: u = D^-1 b
        upd.dinv = sv.dinv;
        upd.c3 = 1.0;
        emit_lin({upd});
        return;
    }
    upd.y2 = its == 1 ? sv.out : P_[0];
    upd.dinv = sv.dinv;
    upd.c3 = 2.0 / (emax + emin);
    SweepLevel lv;
    lv.first = steps_.size();
    emit_lin({upd});
    emit_solves({sv}, its, emin, emax, P_, nx_, true, &lv.coef, eimag);
    lv.last = steps_.size();
    lv.its = its;
    // what the tile form can express: b = ca * sum_t U_t x + cy * b_in with one x
    bool ok = its >= 2 && !upd.terms.empty() && upd.terms.size() <= 2 && upd.cz == 0.0 && !upd.z &&
              upd.yin != nullptr;
    for (const Term &t : upd.terms) ok = ok && t.x == upd.terms[0].x;
    if (ok) {
        TileLevel &L = lv.lev;
        L.vals = sv.vals;
        L.dinv = sv.dinv;
        L.bin = upd.yin;
        // The plain steps update the right-hand side in place.  A tile re-computes the rows of
        // its rings from b_in, so an in-place store by the owner would race with its neighbours'
        // reads: the updated right-hand side stays on chip (nothing reads B_i after a sweep: the
        // next batched step overwrites it).
        L.bout = upd.y == upd.yin ? nullptr : upd.y;
        lv.b_after = upd.y;
        L.out = sv.out;
        L.x_prev = upd.terms[0].x;
        L.n_upd = (int32_t)upd.terms.size();
        for (size_t t = 0; t < upd.terms.size(); ++t) L.upd_vals[t] = upd.terms[t].vals;
        L.ca = upd.ca;
        L.cy = upd.cy;
        L.p1_scale = upd.c3;
        L.post1 = sv.post1;
        L.post2 = sv.post2;
        lv.eligible = true;
    }
    sweep_levels_.push_back(lv);
}

// Batched solves with ONE matrix (the mass solves: every time level the same M): the Chebyshev
// iterates of four levels interleaved, one IlOp per group and step (kernels.hpp).  Same
// coefficients and the same fma chain per row and level as emit_solves below.
bool SchurPC::emit_solves_interleaved(const std::vector<Solve> &sv, int its, double emin,
                                      double emax) {
    const char *o = S_.opt("interleave");
    if (o && o[0] == '0') return false;
    const size_t m = sv.size();
    const char *ln = S_.opt("lanes");
    if (ln && ln[0] != '0') return false;       // (chunks on two streams would share the buffers)
    for (const Solve &q : sv)
        if (q.vals != sv[0].vals || q.dinv != sv[0].dinv) return false;
    const Pattern &P = S_.patterns[m_pat_];
    const int ng = (int)((m + 3) / 4);
    const size_t need = (size_t)ng * 4 * (size_t)nx_;
    if (need > il_cap_) {
        for (int k = 0; k < 3; ++k) {
            il_P_[k] = dev_alloc<double>(need + 32);
            HIPCHK(hipMemsetAsync(il_P_[k], 0, (need + 32) * sizeof(double), S_.stream));
            owned_.push_back(il_P_[k]);
        }
        il_cap_ = need;
    }
    const double scale = 2.0 / (emax + emin);
    const double alpha = 1.0 - scale * emin, mu = 1.0 / alpha, omegaprod = 2.0 / alpha;
    double c_km1 = 1.0, c_k = mu;
    auto buf = [&](int step, int g) -> double * {
        return il_P_[(step - 1) % 3] + (size_t)g * 4 * (size_t)nx_;
    };
    for (int step = 1; step <= its; ++step) {
        double k1 = 0.0, k2 = 0.0, k3 = scale;
        if (step >= 2) {
            const double c_kp1 = 2.0 * mu * c_k - c_km1;
            const double omega = omegaprod * c_k / c_kp1;
            k1 = 1.0 - omega;
            k2 = omega;
            k3 = scale * omega;
            c_km1 = c_k;
            c_k = c_kp1;
        }
        const bool last = step == its;
        std::vector<IlOp> ops(ng);
        for (int g = 0; g < ng; ++g) {
            IlOp op{};
            op.col = P.d_col;
            op.slice_off = P.d_slice_off;
            op.perm = P.d_perm;
            op.vals = sv[0].vals;
            op.dinv = sv[0].dinv;
            op.rowmask = mask_;
            op.nrows = (int32_t)nx_;
            op.nslices = P.nslices;
            op.uniform_w = P.uniform_w;
            op.nlev = (int32_t)std::min<size_t>(4, m - (size_t)g * 4);
            for (int l = 0; l < op.nlev; ++l) {
                op.b[l] = sv[(size_t)g * 4 + l].b;
                op.out[l] = last ? sv[(size_t)g * 4 + l].out : nullptr;
            }
            op.x = step >= 2 ? buf(step - 1, g) : nullptr;
            op.pkm1 = step >= 3 ? buf(step - 2, g) : nullptr;
            op.y = last ? nullptr : buf(step, g);
            op.c1 = k1;
            op.c2 = k2;
            op.c3 = k3;
            for (int l = 0; l < 4; ++l) {
                const bool live = last && l < op.nlev;
                op.post1[l] = live ? sv[(size_t)g * 4 + l].post1 : 1.0;
                op.post2[l] = live ? sv[(size_t)g * 4 + l].post2 : 1.0;
            }
            ops[g] = op;
        }
        PcStep s;
        s.kind = PcStep::ROWS_IL;
        s.d_il = dev_upload(ops.data(), ops.size());
        s.il_groups = ng;
        s.il_slices = P.nslices;
        s.il_w = P.uniform_w;
        s.lane = cur_lane_;
        steps_.push_back(s);
    }
    return true;
}

void SchurPC::emit_solves(const std::vector<Solve> &sv, int its, double emin, double emax,
                          double *const P[3], int64_t pstride, bool first_done,
                          std::vector<TileCoef> *coef_out, double eimag, const Mat *mat) {
    if (coarse_cycles_ > 0 && mat && mat->einv && sv.size() == 1 && !first_done) {
        emit_coarse_solve(nullptr, sv[0], *mat);
        return;
    }
    const size_t m = sv.size();
    if (m >= 4 && !first_done && its >= 2 && eimag == 0.0 && pstride == nx_ &&
        emit_solves_interleaved(sv, its, emin, emax))
        return;
    // a single solve on a final right-hand side is a sweep level without update (the first
    // level of a sweep, the sub-solves of the stationary preconditioner)
    SweepLevel solo;
    const bool record_solo = m == 1 && !first_done && its >= 2;
    if (record_solo) {
        solo.first = steps_.size();
        coef_out = &solo.coef;
    }
    std::vector<Cheb> ops(m);
    if (its == 0) {
        for (size_t q = 0; q < m; ++q)
            ops[q] = Cheb{nullptr, sv[q].dinv, sv[q].b, nullptr, nullptr, sv[q].out,
                          0.0, 0.0, 1.0, sv[q].post1, sv[q].post2};
        emit_cheb(ops);
        return;
    }
    const double scale = 2.0 / (emax + emin);
    const double alpha = 1.0 - scale * emin;
    const double mu = 1.0 / alpha;
    const double omegaprod = 2.0 / alpha;
    double c_km1 = 1.0, c_k = mu;
    // step 1: p_1 = scale * D^-1 b
    auto target = [&](int step, size_t q) -> double * {
        return step == its ? sv[q].out : P[(step - 1) % 3] + (int64_t)q * pstride;
    };
    if (!first_done) {
        for (size_t q = 0; q < m; ++q) {
            const bool last = its == 1;
            ops[q] = Cheb{nullptr, sv[q].dinv, sv[q].b, nullptr, nullptr, target(1, q),
                          0.0, 0.0, scale, last ? sv[q].post1 : 1.0, last ? sv[q].post2 : 1.0};
        }
        emit_cheb(ops);
    }
    // eimag > 0: the spectrum lies in the ellipse with centre d = (emax + emin) / 2 and semi-axes
    // a = (emax - emin) / 2, eimag (blocks with a convection term).  Manteuffel's recurrence
    // (Numer. Math. 28, 1977) for the same three-term sweep: alpha_1 = 2d / (2d^2 - c^2),
    // alpha_n = 1 / (d - (c^2 / 4) alpha_{n-1}), beta_n = d alpha_n - 1 with c^2 = a^2 - eimag^2,
    // which may be negative -- only c^2 enters, the arithmetic stays real.  Restated in the
    // oracle (chebyshev_ellipse_coefficients); for eimag == 0 PETSc's form below is kept, bit
    // for bit what rounds 1 and 2 ran.
    const double ell_d = 0.5 * (emax + emin), ell_a = 0.5 * (emax - emin);
    const double ell_c2 = ell_a * ell_a - eimag * eimag;
    double ell_alpha = 1.0 / ell_d;
    for (int step = 2; step <= its; ++step) {
        double k1, k2, k3;
        if (eimag > 0.0) {
            ell_alpha = 1.0 / (ell_d - (step == 2 ? 0.5 : 0.25) * ell_c2 * ell_alpha);
            const double beta = ell_d * ell_alpha - 1.0;
            k1 = -beta;
            k2 = 1.0 + beta;
            k3 = ell_alpha;
        } else {
            const double c_kp1 = 2.0 * mu * c_k - c_km1;
            const double omega = omegaprod * c_k / c_kp1;
            k1 = 1.0 - omega;
            k2 = omega;
            k3 = scale * omega;
            c_km1 = c_k;
            c_k = c_kp1;
        }
        const bool last = step == its;
        for (size_t q = 0; q < m; ++q) {
            const double *pk = target(step - 1, q);
            const double *pkm1 = step >= 3 ? target(step - 2, q) : nullptr;
            ops[q] = Cheb{sv[q].vals, sv[q].dinv, sv[q].b, pk, pkm1, target(step, q),
                          k1, k2, k3,
                          last ? sv[q].post1 : 1.0, last ? sv[q].post2 : 1.0};
        }
        if (coef_out) coef_out->push_back(TileCoef{k1, k2, k3});
        emit_cheb(ops);
    }
    if (record_solo) {
        solo.last = steps_.size();
        solo.its = its;
        TileLevel &L = solo.lev;
        L.vals = sv[0].vals;
        L.dinv = sv[0].dinv;
        L.bin = sv[0].b;
        L.bout = nullptr;
        L.out = sv[0].out;
        L.n_upd = 0;
        L.p1_scale = scale;
        L.post1 = sv[0].post1;
        L.post2 = sv[0].post2;
        solo.eligible = true;
        sweep_levels_.push_back(solo);
    }
}

// ---- Control.Stationary.construct_pc, control.py:356-448
void SchurPC::build_stationary() {
    const double c = 1.0 / std::sqrt(d_.beta);
    const double *Dv = block_vals(KKT_Q10, 0, 0);
    const double *Dz = block_vals(KKT_Q01, 0, 0);
    double *b0 = in_, *b1 = in_ + nx_;
    double *u0 = out_, *u1 = out_ + nx_;
    emit_solves({Solve{m_vals_, m_dinv_, b0, u0}}, d_.mass_its, d_.mass_emin, d_.mass_emax, P_, nx_);
    emit_lin({Lin{{Term{Dv, u0}}, B_, 1.0, 0.0, -1.0, nullptr, b1}});
    Mat S1 = schur_matrix(Dv, c), S2 = schur_matrix(Dz, c);
    schur_its_ = resolve_its(S1);
    emit_solves({Solve{S1.vals, S1.dinv, B_, u1}}, schur_its_, S1.emin, S1.emax, P_, nx_, false, nullptr, S1.eimag, &S1);
    emit_lin({Lin{{Term{m_vals_, u1}}, B_, 1.0}});
    emit_solves({Solve{S2.vals, S2.dinv, B_, u1}}, schur_its_, S2.emin, S2.emax, P_, nx_, false, nullptr, S2.eimag, &S2);
}

// Time sharding: a rank owns blocks [lo, hi).  Everything that is independent
// per time level (mass solves, residual products) runs on the local blocks; the sweeps and
// the CN scans are serial in time, so ranks hand one N_x vector to the next rank and form a
// pipeline -- COMM steps between graph-captured segments of the program.

// ---- Control.Instationary.construct_pc, BE branch, control.py:2191-2438
void SchurPC::build_BE() {
    const int n = n_, lo = lo_, hi = hi_;
    const double tau = d_.tau, eps = d_.epsilon;
    const double shift = tau / std::sqrt(d_.beta);
    const int64_t nl = hi - lo;
    double *b0 = in_, *b1 = in_ + nl * nx_;
    double *u0 = out_, *u1 = out_ + nl * nx_;
    auto blk = [&](double *base, int i) { return base + (int64_t)(i - lo) * nx_; };
    const int up = hi < n ? S_.rank + 1 : -1, dn = lo > 0 ? S_.rank - 1 : -1;
    // Unsharded handles split the level range into chunks: the mass solves and the
    // right-hand-side products of chunk c run on the side lane while the forward sweep works
    // through chunk c - 1 on the main lane.
    const bool lanes = use_lanes_ && !S_.sharded && (hi - lo) >= 16;
    int n_chunks = 1;
    if (lanes) {
        const char *e = S_.opt("lane_chunks");
        n_chunks = std::max(2, std::min((hi - lo) / 4, e ? std::atoi(e) : 4));
    }
    std::vector<int> cfirst(n_chunks + 1);
    for (int c = 0; c <= n_chunks; ++c) cfirst[c] = lo + (int)((int64_t)(hi - lo) * c / n_chunks);
    std::vector<int> chunk_done(n_chunks, -1);
    auto coef = [&](int i) { return i == 0 ? 0.0 : (i == n - 1 ? std::sqrt(eps) * shift : shift); };
    {
        const int imid = (lo + hi) / 2;      // a typical level: decides the degree when it is derived
        schur_its_ = resolve_its(schur_matrix(block_vals(KKT_Q10, imid, imid), coef(imid)));
    }
    auto side_chunk = [&](int c) {
        const int c0 = cfirst[c], c1 = cfirst[c + 1];
        cur_lane_ = lanes ? 1 : 0;
        // (1,1)-block: u0_i = (1/tau) M~^-1 b0_i, last one also / epsilon   (2193-2206)
        {
            std::vector<Solve> sv;
            for (int i = c0; i < c1; ++i)
                sv.push_back(Solve{m_vals_, m_dinv_, blk(b0, i), blk(u0, i), 1.0 / tau,
                                   i == n - 1 ? 1.0 / eps : 1.0});
            double *const Pc[3] = {blk(P_[0], c0), blk(P_[1], c0), blk(P_[2], c0)};
            emit_solves(sv, d_.mass_its, d_.mass_emin, d_.mass_emax, Pc, nx_);
        }
        if (c == n_chunks - 1 && (up >= 0 || dn >= 0)) emit_comm(blk(u0, hi - 1), up, h_u0_, dn);
        // b_i = block_10(i,i) u0_i + block_10(i,i-1) u0_{i-1} - b1_i   (2208-2237)
        {
            std::vector<Lin> ops;
            for (int i = c0; i < c1; ++i) {
                Lin l;
                l.terms.push_back(Term{block_vals(KKT_Q10, i, i), blk(u0, i)});
                if (i >= 1)
                    l.terms.push_back(Term{block_vals(KKT_Q10, i, i - 1),
                                           i - 1 >= lo ? blk(u0, i - 1) : h_u0_});
                l.y = blk(B_, i);
                l.cz = -1.0;
                l.z = blk(b1, i);
                ops.push_back(l);
            }
            emit_lin(ops);
        }
        if (lanes) chunk_done[c] = emit_record(1);
        cur_lane_ = 0;
    };
    // forward sweep (2241-2327)
    auto sweep_chunk = [&](int c) {
        if (lanes) emit_wait(0, chunk_done[c]);
        for (int i = cfirst[c]; i < cfirst[c + 1]; ++i) {
            Mat F = schur_matrix(block_vals(KKT_Q10, i, i), coef(i));
            const Solve sv{F.vals, F.dinv, blk(B_, i), blk(u1, i)};
            if (i >= 1)
                emit_update_and_solve(Lin{{Term{block_vals(KKT_Q10, i, i - 1),
                                                i - 1 >= lo ? blk(u1, i - 1) : h_u1_}},
                                          blk(B_, i), -1.0, 1.0, 0.0, blk(B_, i), nullptr},
                                      sv, schur_its_, F.emin, F.emax, F.eimag, &F);
            else
                emit_solves({sv}, schur_its_, F.emin, F.emax, P_, nx_, false, nullptr, F.eimag, &F);
        }
    };
    if (lanes) {
        // host submission order interleaves the lanes: the sweep of chunk c (three calls) is
        // queued before the ~20 launches of chunk c + 1, so neither stream waits for the host
        const int e_start = emit_record(0);
        emit_wait(1, e_start);
        side_chunk(0);
        for (int c = 0; c < n_chunks; ++c) {
            sweep_chunk(c);
            if (c + 1 < n_chunks) side_chunk(c + 1);
        }
    } else {
        side_chunk(0);
        if (dn >= 0) emit_comm(nullptr, -1, h_u1_, dn);
        sweep_chunk(0);
    }
    if (up >= 0) emit_comm(blk(u1, hi - 1), up, nullptr, -1);
    // b_i = tau M u1_i (epsilon tau for the last)   (2330-2350)
    {
        std::vector<Lin> ops;
        for (int i = lo; i < hi; ++i)
            ops.push_back(Lin{{Term{m_vals_, blk(u1, i)}}, blk(B_, i),
                              i == n - 1 ? eps * tau : tau});
        emit_lin(ops);
    }
    // backward sweep (2353-2437)
    if (up >= 0) emit_comm(nullptr, -1, h_u1_, up);
    for (int i = hi - 1; i >= lo; --i) {
        Mat G = schur_matrix(block_vals(KKT_Q01, i, i), coef(i));
        const Solve sv{G.vals, G.dinv, blk(B_, i), blk(u1, i)};
        if (i <= n - 2)
            emit_update_and_solve(Lin{{Term{block_vals(KKT_Q01, i, i + 1),
                                            i + 1 < hi ? blk(u1, i + 1) : h_u1_}},
                                      blk(B_, i), -1.0, 1.0, 0.0, blk(B_, i), nullptr},
                                  sv, schur_its_, G.emin, G.emax, G.eimag, &G);
        else
            emit_solves({sv}, schur_its_, G.emin, G.emax, P_, nx_, false, nullptr, G.eimag, &G);
    }
    if (dn >= 0) emit_comm(blk(u1, lo), dn, nullptr, -1);
}

// ---- Control.Instationary.construct_pc, CN branch, control.py:1995-2189
void SchurPC::build_CN() {
    const int n = n_, lo = lo_, hi = hi_;
    const double tau = d_.tau;
    const double c = 0.5 * tau / std::sqrt(d_.beta);   // my_const, control.py:2051
    const int64_t nl = hi - lo;
    const int nloc = (int)nl;
    double *b0 = in_, *b1 = in_ + nl * nx_;
    double *u0 = out_, *u1 = out_ + nl * nx_;
    auto blk = [&](double *base, int i) { return base + (int64_t)(i - lo) * nx_; };
    const int up = hi < n ? S_.rank + 1 : -1, dn = lo > 0 ? S_.rank - 1 : -1;
    Mat cM = schur_matrix(nullptr, c);   // c * M~ (base absent): products with my_const * M
    {
        const int imid = (lo + hi) / 2;
        schur_its_ = resolve_its(schur_matrix(block_vals(KKT_Q10, imid, imid), c));
    }
    // (1,1)-block (1997-2014): T_1^-1 is a scan from the last block down
    if (up >= 0) emit_comm(nullptr, -1, h_t_, up);
    emit_time(T_, b0, 3, nloc, nullptr, up >= 0 ? h_t_ : nullptr);
    if (dn >= 0) emit_comm(blk(T_, lo), dn, nullptr, -1);
    {
        std::vector<Solve> sv;
        for (int i = lo; i < hi; ++i)
            sv.push_back(Solve{m_vals_, m_dinv_, blk(T_, i), blk(u0, i), 2.0 / tau, 1.0});
        emit_solves(sv, d_.mass_its, d_.mass_emin, d_.mass_emax, P_, nx_);
    }
    // T_2^-1: scan from the first block up; afterwards h_u0_ holds the final u0_{lo-1}
    if (dn >= 0) emit_comm(nullptr, -1, h_u0_, dn);
    emit_time(u0, u0, 4, nloc, dn >= 0 ? h_u0_ : nullptr, nullptr);
    if (up >= 0) emit_comm(blk(u0, hi - 1), up, nullptr, -1);
    // b = T_2 (D_v u0) - b1 (2016-2048)
    {
        std::vector<Lin> ops;
        for (int i = lo; i < hi; ++i) {
            Lin l;
            l.terms.push_back(Term{block_vals(KKT_Q10, i, i), blk(u0, i)});
            if (i >= 1)
                l.terms.push_back(Term{block_vals(KKT_Q10, i, i - 1),
                                       i - 1 >= lo ? blk(u0, i - 1) : h_u0_});
            l.y = blk(B_, i);
            ops.push_back(l);
        }
        emit_lin(ops);
    }
    if (up >= 0 || dn >= 0) emit_comm(blk(B_, hi - 1), up, h_t_, dn);   // old values
    emit_time(B_, B_, 2, nloc, dn >= 0 ? h_t_ : nullptr, nullptr);
    {
        std::vector<Lin> ops;
        for (int i = lo; i < hi; ++i)
            ops.push_back(Lin{{}, blk(B_, i), 0.0, 1.0, -1.0, blk(B_, i), blk(b1, i)});
        emit_lin(ops);
    }
    // forward sweep (2050-2116)
    if (dn >= 0) emit_comm(nullptr, -1, h_t_, dn);
    emit_time(B_, B_, 4, nloc, dn >= 0 ? h_t_ : nullptr, nullptr);
    if (up >= 0) emit_comm(blk(B_, hi - 1), up, nullptr, -1);
    if (dn >= 0) emit_comm(nullptr, -1, h_u1_, dn);
    for (int i = lo; i < hi; ++i) {
        Mat F = schur_matrix(block_vals(KKT_Q10, i, i), c);
        const Solve sv{F.vals, F.dinv, blk(B_, i), blk(u1, i)};
        if (i >= 1) {
            const double *prev = i - 1 >= lo ? blk(u1, i - 1) : h_u1_;
            emit_update_and_solve(Lin{{Term{block_vals(KKT_Q10, i, i - 1), prev},
                                       Term{cM.vals, prev}},
                                      blk(B_, i), -1.0, 1.0, 0.0, blk(B_, i), nullptr},
                                  sv, schur_its_, F.emin, F.emax, F.eimag, &F);
        } else {
            emit_solves({sv}, schur_its_, F.emin, F.emax, P_, nx_, false, nullptr, F.eimag, &F);
        }
    }
    if (up >= 0) emit_comm(blk(u1, hi - 1), up, nullptr, -1);
    // u1 = T_2 u1 (h_u1_ still holds the untransformed u1_{lo-1}); b_i = (tau/2) M u1_i
    emit_time(u1, u1, 2, nloc, dn >= 0 ? h_u1_ : nullptr, nullptr);
    {
        std::vector<Lin> ops;
        for (int i = lo; i < hi; ++i)
            ops.push_back(Lin{{Term{m_vals_, blk(u1, i)}}, blk(B_, i), 0.5 * tau});
        emit_lin(ops);
    }
    // backward sweep (2135-2189)
    if (up >= 0) emit_comm(nullptr, -1, h_u1_, up);
    for (int i = hi - 1; i >= lo; --i) {
        Mat G = schur_matrix(block_vals(KKT_Q01, i, i), c);
        const Solve sv{G.vals, G.dinv, blk(B_, i), blk(u1, i)};
        if (i <= n - 2) {
            Mat H = schur_matrix(block_vals(KKT_Q01, i, i + 1), c);
            emit_update_and_solve(Lin{{Term{H.vals, i + 1 < hi ? blk(u1, i + 1) : h_u1_}},
                                      blk(B_, i), -1.0, 1.0, 0.0, blk(B_, i), nullptr},
                                  sv, schur_its_, G.emin, G.emax, G.eimag, &G);
        } else {
            emit_solves({sv}, schur_its_, G.emin, G.emax, P_, nx_, false, nullptr, G.eimag, &G);
        }
    }
    if (dn >= 0) emit_comm(blk(u1, lo), dn, nullptr, -1);
}

void SchurPC::emit_comm(const double *send, int dst, double *recv, int src) {
    PcStep s;
    s.kind = PcStep::COMM;
    s.x = send;
    s.y = recv;
    s.dst = dst;
    s.src = src;
    s.nx = nx_;
    s.lane = cur_lane_;
    steps_.push_back(s);
}

void SchurPC::replay(size_t first, size_t last) {
    {
        const char *px = S_.opt("pc_xcd");
        set_pc_xcd(!(px && px[0] == '0'));
    }
    Bases B{{nullptr, nullptr, nullptr, nullptr}};
    if (n_events_ > 0 && !side_) {
        HIPCHK(hipStreamCreateWithFlags(&side_, hipStreamNonBlocking));
    }
    while ((int)events_.size() < n_events_) {
        hipEvent_t e;
        HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        events_.push_back(e);
    }
    for (size_t k = first; k < last; ++k) {
        const PcStep &s = steps_[k];
        hipStream_t st = s.lane == 0 ? S_.stream : side_;
        switch (s.kind) {
            case PcStep::EV_RECORD:
                HIPCHK(hipEventRecord(events_[s.ev], st));
                break;
            case PcStep::EV_WAIT:
                HIPCHK(hipStreamWaitEvent(st, events_[s.ev], 0));
                break;
            case PcStep::ROWS:
                if (s.rows.shared_matrix &&
                    launch_rowops_shared(st, s.rows.d_ops, s.rows.nops, s.rows.max_slices,
                                         s.rows.R, s.rows.uniform_w))
                    break;
                launch_rowops(st, s.rows.d_ops, s.rows.nops, s.rows.max_slices, s.rows.R, B, 1,
                              s.rows.uniform_w, s.rows.single ? &s.rows.h_op : nullptr);
                break;
            case PcStep::ROWS_IL:
                launch_rowops_il(st, s.d_il, s.il_groups, s.il_slices, s.il_w);
                break;
            case PcStep::TIME:
                launch_time_transform(st, s.y, s.x, s.tkind, s.n, s.nx, s.lo_halo, s.hi_halo);
                break;
            case PcStep::COPY:
                launch_copy(st, s.y, s.x, s.nx);
                break;
            case PcStep::COARSE:
                launch_coarse_correction(st, coarse_, s.einv, s.cr, s.x, s.y, s.nx);
                break;
            case PcStep::PROG: {
                const Pattern &P = S_.patterns[m_pat_];
                if (s.gmode == 2)
                    launch_row_program_gw(st, s.rows.d_ops, s.nphases, prog_nwg_, prog_wpw_, d_g0_,
                                          d_g1_, granule_words_, d_err_);
                else if (s.granule)
                    launch_row_program_g(st, s.rows.d_ops, s.d_lite, s.nphases, prog_nwg_, prog_wpw_,
                                         P.uniform_w, d_g0_, d_g1_, granule_words_, d_err_);
                else
                    launch_row_program(st, s.rows.d_ops, s.nphases, prog_nwg_, prog_wpw_, P.R,
                                       P.uniform_w, d_dep_, d_flags_, d_err_, prog_lowreg_);
                break;
            }
            case PcStep::TILE: {
                const TilePlan &tp = tile_plan_;
                TileArgs a{};
                a.coarse = s.coarse ? d_tile_coarse_ : nullptr;
                a.einv = s.d_einv;
                a.cycles = s.coarse ? coarse_cycles_ : 0;
                a.cepoch0 = s.cepoch0;
                a.nlevels = s.nlevels;
                a.its = s.its;
                a.depth = tp.depth;
                a.nk_pad = tp.nk_pad;
                a.rpt = tp.rpt;
                a.W = tp.W;
                a.gnew[0] = d_tg_[0];
                a.gnew[1] = d_tg_[1];
                a.gold[0] = d_tg_[2];
                a.gold[1] = d_tg_[3];
                const size_t words = 2 * (size_t)S_.patterns[m_pat_].nrows;
                a.granule_bytes = (unsigned)(words * sizeof(unsigned long long));
                a.err = d_err_;
                a.epoch0 = s.epoch0;
                a.clear = s.clear ? 1 : 0;
                a.fused_update = s.fused ? 1 : 0;
                a.hslots = tp.hslots;
                a.stamps = S_.opt("stamps") != nullptr;
                {
                    const char *dd = S_.opt("debug_drop_handoff");
                    a.debug_drop = dd ? std::atoi(dd) : 0;
                }
                {
                    const char *pd = S_.opt("tile_poll_delay");
                    // s_sleep units before the first poll: ~0.7 us; ~1.4 us for the big rings of
                    // 3-D tiles (measured optima: 256^2 P1 24, 64^3 P1 48)
                    a.poll_delay = pd ? std::atoi(pd) : 24;
                }
                try {
                    launch_tile_sweep(st, a, s.d_levels, tp.d_n, tp.d_grow, tp.d_lcol, tp.d_gpos,
                                      mask_, tp.ntiles, tp.threads, words,
                                      s.coarse ? &h_tile_coarse_ : nullptr);
                } catch (const TileLaunchError &e) {
                    fail(KKT_ERR_HIP, e.msg);
                }
                break;
            }
            case PcStep::COMM:
                if (!S_.comm) fail(KKT_ERR_STATE, "time-sharded system without a transport");
                S_.comm->sendrecv(s.x, s.nx, s.dst, s.y, s.nx, s.src, st);
                break;
        }
    }
}

// The program is a fixed launch sequence on fixed buffers: every maximal run of kernel
// steps between two COMM steps is captured once into a hipGraph and replayed.
void SchurPC::run() {
    hipStream_t st = S_.stream;
    if (segments_.empty()) {
        size_t k = 0;
        while (k < steps_.size()) {
            Segment g;
            g.first = k;
            if (steps_[k].kind == PcStep::COMM) {
                g.last = k + 1;
                g.comm = true;
            } else {
                size_t e = k;
                while (e < steps_.size() && steps_[e].kind != PcStep::COMM) ++e;
                g.last = e;
            }
            k = g.last;
            segments_.push_back(g);
        }
    }
    for (Segment &g : segments_) {
        if (g.comm || !use_graph_ || g.last - g.first < 4) {
            replay(g.first, g.last);
            continue;
        }
        if (!g.exec) {
            hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
            if (e == hipSuccess) {
                replay(g.first, g.last);
                e = hipStreamEndCapture(st, &g.graph);
                if (e == hipSuccess) e = hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0);
            }
            if (e != hipSuccess) {
                (void)hipGetLastError();
                g.exec = nullptr;
                use_graph_ = false;   // plain launches of the same kernels
                replay(g.first, g.last);
                continue;
            }
        }
        HIPCHK(hipGraphLaunch(g.exec, st));
    }
}

}  // namespace kkt
