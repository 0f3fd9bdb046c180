// Tile sweep program for gfx950: the serial time sweeps of the block-Schur preconditioner
// (reference control/control.py:2263-2295, 2375-2406) as ONE persistent launch in which a
// workgroup advances `depth` dependent SpMV steps per hand-off (matrix-powers kernel; plan and
// rationale: tiles.hpp).
//
// One workgroup per CU owns one tile: its own rows plus the rings within graph distance
// `depth`.  State per workgroup:
//   registers  matrix values and local column indices of the rows it ever computes
//              (rows [0, n[depth-1]) of the tile, RPT per thread), Jacobi diagonal, right-hand side
//   LDS        the two newest iterates on all n[depth] local rows
// A step computes p_s on the rows whose columns are still valid (one ring fewer than the step
// before) out of LDS; when the rings are used up, the own rows of the two newest iterates are
// published as tagged granules and the rings are re-gathered from the owners' granules
// ("the data is the flag", cdna_hip_programming.md Guideline 16 R2: no flag, no drain, no
// grid barrier).  Granule buffers alternate with the hand-off epoch; write-after-read safety
// is the argument of pc_row_program_g at tile granularity (a tile publishes epoch e + 1 only
// after it has seen epoch e of every tile it gathers from, and those tiles gather from it).
//
// Arithmetic: every row keeps the fma chain of the plain kernels (kernels.hip rowops_body:
// CSR order inside a term, terms in order) and the same epilogue expressions, so the results
// equal the plain launches bit for bit (tests/test_gpu_parity.py).
#include "kernels.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

namespace kkt {

#define KKT_GLOBAL __attribute__((address_space(1)))
typedef KKT_GLOBAL const double *gcd_p;
typedef KKT_GLOBAL double *gd_p;
typedef KKT_GLOBAL const int32_t *gci_p;
typedef KKT_GLOBAL const uint16_t *gcu16_p;
typedef KKT_GLOBAL const uint8_t *gcb_p;
typedef KKT_GLOBAL unsigned long long *gu64_p;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned TILE_SPIN_LIMIT = 1u << 21;

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory
// counter (s_waitcnt vmcnt(0)): every outstanding global load or write-through store would cost
// a round trip per local step.  Global data is ordered by the granule tags, not by barriers.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// one row's value as a 16-byte granule {lo, tag, hi, tag}: ONE write-through (sc1) store
__device__ __forceinline__ void publish(const __amdgpu_buffer_rsrc_t rs, int grow, double v,
                                        unsigned tag) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    const u32x4 g = {(unsigned)bits, tag, (unsigned)(bits >> 32), tag};
    __builtin_amdgcn_raw_buffer_store_b128(g, rs, grow * 16, 0, 16 /* sc1 */);
}

// the fields of a level that are needed before the level begins (its operands are requested
// under the last steps of the level before), by value
struct LevEarly {
    const double *vals, *dinv, *bin, *upd0;
    int32_t n_upd, prev_in_lds;
};
__device__ __forceinline__ LevEarly read_early(const TileLevel *p) {
    LevEarly f;
    f.vals = p->vals;
    f.dinv = p->dinv;
    f.bin = p->bin;
    f.upd0 = p->upd_vals[0];
    f.n_upd = p->n_upd;
    f.prev_in_lds = p->prev_in_lds;
    return f;
}

// (an opaque read of a packed column register: the decode -- shift / mask, times 8 plus the vector's
// base -- then stays inside the step loop instead of being hoisted into registers of its own:
// W registers per row slot)
template <typename P>
__device__ __forceinline__ P opaque_ptr(P p) {
    unsigned long long u = (unsigned long long)p;
    asm volatile("" : "+v"(u));
    return (P)u;
}
__device__ __forceinline__ unsigned opaque(unsigned x) {
    asm volatile("" : "+v"(x));
    return x;
}

// HPT_: ring-entry slots per thread of a hand-off (RPT + 1 unless the variant is short of registers)
// COARSE: levels are cycles of [Galerkin coarse correction, `its` smoothing sweeps] (two-grid form
// of the sub-solves); the plain variants compile none of it.
template <int W, int RPT, int TMAX, bool UPD, int HPT_ = RPT + 1, bool COARSE = false>
__global__ __launch_bounds__(TMAX) void pc_tile_sweep(
    const TileArgs A, const TileLevel *__restrict__ levels, const int32_t *__restrict__ n_all,
    const int32_t *__restrict__ grow_all, const uint16_t *__restrict__ lcol_all,
    const int32_t *__restrict__ gpos_all, const uint8_t *__restrict__ rowmask_) {
    constexpr int HPT = HPT_;
    extern __shared__ double X[];
    __shared__ int sn[TILE_DEPTH_MAX + 1];
    __shared__ int sdead;
    __shared__ unsigned long long sstat[8];
    const int T = blockDim.x, tid = threadIdx.x, tile = blockIdx.x;
    const int nkp = A.nk_pad, depth = A.depth, its = A.its;
    const int32_t *nt = n_all + (size_t)tile * (TILE_DEPTH_MAX + 1);
    const int n0 = nt[0], nk = nt[depth], nk1 = nt[depth - 1];
    if (tid <= TILE_DEPTH_MAX) sn[tid] = nt[tid];
    if (tid < 8) sstat[tid] = 0ull;
    if (tid == 0) {
        sdead = 0;
        // the zero slot: what columns of boundary rows read (their iterates are exactly 0)
        X[nkp - 1] = 0.0;
        X[2 * (size_t)nkp - 1] = 0.0;
    }
    // ---- residency check.  The hand-offs below assume that every tile of the grid is running at
    // the same time (one workgroup per CU).  If the device cannot place them all at once -- CUs
    // masked off, another process holding wave slots -- the resident tiles would each spin 2^21
    // times in their first hand-off before giving up.  Instead every tile checks in on a counter
    // (it only ever grows: each launch adds exactly gridDim.x, so the value a tile must see is the
    // next multiple above what its own atomic returned) and waits a bounded ~5 ms for the others;
    // a tile that does not see them all marks the launch as abandoned, and every tile -- those
    // still waiting and those scheduled later -- leaves at once.  The host then falls back to
    // plain launches (kkt_info.program_fallbacks).
    __shared__ int s_bail;
    if (tid == 0) {
        typedef KKT_GLOBAL unsigned long long *gu64a_p;
        const gu64a_p ctr = (gu64a_p)(A.err + 2), bail = (gu64a_p)(A.err + 4);
        const unsigned long long nt_ = gridDim.x;
        const bool skip = A.debug_drop < 0 && tile == 0;      // test hook: tile 0 never checks in
        unsigned long long old = 0ull;
        if (!skip) old = __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else old = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long target = (old / nt_ + 1ull) * nt_;
        int bad = 0;
        unsigned long long seen = old;
        // (4 096 polls of ~1.2 us: 5 ms -- long against any launch skew, 400 times shorter than
        // the bounded spin of a hand-off)
        for (int spin = 0; spin < 4096; ++spin) {
            seen = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (seen >= target) break;
            if (__hip_atomic_load(bail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == target) break;
            __builtin_amdgcn_s_sleep(8);
        }
        if (seen < target) {
            __hip_atomic_store(bail, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bad = 1;
        } else if (__hip_atomic_load(bail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == target) {
            bad = 1;
        }
        if (bad) {
            atomicOr(A.err, 8u);
            if (atomicCAS(A.err + 1, 0u, 1u) == 0u) {
                A.err[17] = (unsigned)(seen - (target - nt_));     // tiles that had checked in
                A.err[18] = (unsigned)nt_;
            }
        }
        s_bail = bad;
    }
    __syncthreads();
    if (s_bail) return;

    const gci_p grow = (gci_p)grow_all + (size_t)tile * nkp;
    const gcb_p rowmask = (gcb_p)rowmask_;

    // ---- what never changes during the launch: structure of the rows this thread computes.
    // Narrow rows (2-D P1): local column indices and the positions of the matrix values in a
    // SELL array stay in registers.  Wide rows (3-D P1, P2): columns are packed two to a
    // register and the positions are re-read from memory once per level and matrix.
    constexpr bool PACK = W > 9;
    constexpr int CH = PACK ? 5 : W;   // entries of a row handled at a time
    constexpr int WP = (W + 1) / 2;
    constexpr int CW = PACK ? WP : W, GW = PACK ? 1 : W;
    unsigned cpk[RPT][CW];
    int gpr[RPT][GW];
    int gr[RPT];
    bool msk[RPT];
    const gcu16_p lcol = (gcu16_p)lcol_all + (size_t)tile * RPT * W * T + tid;
    const gci_p gpos = (gci_p)gpos_all + (size_t)tile * RPT * W * T + tid;
    const gci_p gpos_base = gpos;
#pragma unroll
    for (int sl = 0; sl < RPT; ++sl) {
        const int r = sl * T + tid;
        if constexpr (PACK) {
#pragma unroll
            for (int k = 0; k < W; k += 2) {
                const unsigned lo = lcol[(size_t)(sl * W + k) * T];
                const unsigned hi = k + 1 < W ? lcol[(size_t)(sl * W + k + 1) * T] : 0u;
                cpk[sl][k / 2] = lo | (hi << 16);
            }
            gpr[sl][0] = 0;
        } else {
#pragma unroll
            for (int k = 0; k < W; ++k) {
                cpk[sl][k] = lcol[(size_t)(sl * W + k) * T];
                gpr[sl][k] = gpos[(size_t)(sl * W + k) * T];
            }
        }
        gr[sl] = r < nk1 ? grow[r] : -1;
        msk[sl] = gr[sl] >= 0 && rowmask != nullptr && rowmask[gr[sl]] != 0;
    }
    // (the 1 024-thread variants -- 128 registers -- read the packed columns opaquely, see
    // opaque(); where the decoded offsets fit into registers, hoisting them is 2 % faster)
    constexpr bool OPQ = PACK && TMAX > 512;
#define KKT_CPK(sl, j) (OPQ ? opaque(cpk[sl][j]) : cpk[sl][j])
#define KKT_COL(sl, k)                                                                       \
    (int)(PACK ? (((k) & 1) ? (KKT_CPK(sl, PACK ? (k) / 2 : 0) >> 16)                         \
                            : (KKT_CPK(sl, PACK ? (k) / 2 : 0) & 0xffffu))                    \
               : cpk[sl][PACK ? 0 : (k)])
#define KKT_GP(sl, k) (PACK ? gpos[(size_t)((sl) * W + (k)) * T] : gpr[sl][PACK ? 0 : (k)])
    // ... and the ring entries it gathers at a hand-off
    int hl[HPT], hg[HPT];
#pragma unroll
    for (int h = 0; h < HPT; ++h) {
        const int l = n0 + h * T + tid;
        hl[h] = l < nk ? l : -1;
        hg[h] = l < nk ? grow[l] : 0;
    }

    // step coefficients live in LDS behind the two iterates: a global load inside the step loop
    // would be drained by every workgroup barrier (vmcnt(0)), a round trip per step
    double *scoef = X + 2 * (size_t)nkp;
    // (coarse levels: the table also holds the first sweep after a correction, `its` entries)
    double *dump = scoef + 3 * (size_t)(COARSE ? (its > 0 ? its : 1) : (its > 1 ? its - 1 : 1));
    // coarse corrections: every tile's partial sums, the coarse residual, this tile's corrections
    double *SL = dump + 1, *RC = nullptr, *EC = nullptr;
    int c_nc = 0, c_jmax = 0, c_n0max = 0, c_nslots = 0, c_nj = 0, c_slot0 = 0;
    unsigned cepoch = A.cepoch0;
    if constexpr (COARSE) {
        const TileCoarseDev *cd = A.coarse;
        c_nc = cd->nc;
        c_jmax = cd->jmax;
        c_n0max = cd->n0max;
        c_nslots = cd->nslots;
        c_nj = cd->nj[tile];
        c_slot0 = cd->slot0[tile];
        RC = SL + c_nslots;
        EC = RC + c_nc;
    }
    // ... and what never changes between corrections, copied into LDS once (read from memory per
    // correction it was a chain of eight dependent round trips, ~25 us per level): the tile's
    // restriction lists and prolongation entries, the slot -> coarse function map
    double *RWc = nullptr, *PWc = nullptr;
    int *CSLc = nullptr, *CIPc = nullptr, *RIPc = nullptr, *JGc = nullptr;
    uint16_t *RROWc = nullptr, *PKc = nullptr;
    int pe0[RPT], pe1[RPT];
    bool c_cached = true;    // the tile's P entries are in LDS (3-D: 8 per row, they stay in memory)
    int c_re0 = 0, c_pq0 = 0;
    // Rows of (P^T A P)^-1: coarse function j belongs to tile j mod ntiles, which forms
    // (E^-1 r_c)_j for everybody and publishes it as a granule (every tile holds the whole coarse
    // residual anyway).  A tile forming the products for the functions its own rows touch read
    // 12 rows of 1 089 doubles from L2 per correction on cfg 2 -- 27 MB over the chip, 6 of the
    // exchange's 10 us -- four tiles each repeating the same product; the 4-5 rows a tile owns
    // stay in LDS while consecutive levels share the matrix (time-invariant operators).
    double *EINVc = nullptr;
    bool c_einv_cache = false;
    const void *einv_key = nullptr;
    const int c_nown = COARSE ? (c_nc > tile ? (c_nc - tile + (int)gridDim.x - 1) / (int)gridDim.x : 0) : 0;
    // the column range of the first row this wave owns (row j over its diagonal block only,
    // kernels.hpp TileCoarseDev), read once: few tiles own more rows than they have waves
    int c_lo0 = 0, c_hi0 = 0, c_ew = 0;
    if constexpr (COARSE) {
        const TileCoarseDev *cd = A.coarse;
        c_ew = cd->ew;
        if ((int)(threadIdx.x >> 6) < c_nown) {
            const int j0 = tile + (int)(threadIdx.x >> 6) * (int)gridDim.x;
            c_lo0 = cd->e_lo[j0];
            c_hi0 = cd->e_hi[j0];
        }
        c_cached = cd->cache_lists != 0;
        const int nrm = c_cached ? cd->nr_max : 0;
        RWc = EC + c_jmax;
        PWc = RWc + nrm;
        CSLc = reinterpret_cast<int *>(PWc + nrm);
        CIPc = CSLc + c_nslots;
        RIPc = CIPc + (c_nc + 1);
        JGc = RIPc + (c_jmax + 1);
        RROWc = reinterpret_cast<uint16_t *>(JGc + c_jmax);
        PKc = RROWc + nrm;
        c_einv_cache = cd->cache_einv != 0;
        // (8-byte aligned: behind the 2-byte lists, rounded up)
        EINVc = reinterpret_cast<double *>(
            (reinterpret_cast<uintptr_t>(PKc + nrm) + 7) & ~(uintptr_t)7);
        const gci_p rip = (gci_p)cd->r_ip + (size_t)tile * c_jmax;
        const gci_p pip = (gci_p)cd->p_ip + (size_t)tile * c_n0max;
        const int re0 = rip[0], re1 = rip[c_nj], pq0 = pip[0], pq1 = pip[nt[0]];
        c_re0 = re0;
        c_pq0 = pq0;
        const gcu16_p rrow = (gcu16_p)cd->r_row, pk = (gcu16_p)cd->p_k;
        const gcd_p rw = (gcd_p)cd->r_w, pw = (gcd_p)cd->p_w;
        if (c_cached) {
            for (int i = tid; i < re1 - re0; i += T) {
                RWc[i] = rw[re0 + i];
                RROWc[i] = rrow[re0 + i];
            }
            for (int i = tid; i < pq1 - pq0; i += T) {
                PWc[i] = pw[pq0 + i];
                PKc[i] = pk[pq0 + i];
            }
        }
        const gci_p cslot = (gci_p)cd->c_slot, cip = (gci_p)cd->c_ip;
        for (int i = tid; i < c_nslots; i += T) CSLc[i] = cslot[i];
        for (int i = tid; i <= c_nc; i += T) CIPc[i] = cip[i];
        const gci_p jg = (gci_p)cd->jglob + (size_t)tile * c_jmax;
        for (int i = tid; i <= c_nj; i += T) RIPc[i] = rip[i] - re0;
        for (int i = tid; i < c_nj; i += T) JGc[i] = jg[i];
#pragma unroll
        for (int sl = 0; sl < RPT; ++sl) {
            const int r = sl * T + tid;
            pe0[sl] = r < nt[0] ? pip[r] - pq0 : 0;
            pe1[sl] = r < nt[0] ? pip[r + 1] - pq0 : 0;
        }
        __syncthreads();
    }
    const void *coef_key = nullptr;
    int cur = 0;             // X + cur * nkp: the newest iterate; the other half: the one before
    unsigned epoch = A.epoch0;   // tags never repeat between the launches of one application
    bool dead = false;       // a spin timed out: stop waiting, run to the end, results invalid
    const void *vals_key = nullptr;
    double v[RPT][W];

    // diagnostics (option "stamps"): 100 MHz ticks this workgroup spent in hand-offs / local
    // steps / level prologues, counts, poll rounds -- 8 x 64-bit words per tile at err + 64
    unsigned long long *stat = reinterpret_cast<unsigned long long *>(A.err + 64) + (size_t)tile * 8;
    const bool stamps = A.stamps != 0;
    unsigned long long t_mark = stamps ? wall_clock64() : 0ull;
    // (summed in LDS and written out once at the end: a read-modify-write of memory per stamp is
    // a round trip that lands in the NEXT interval)
    auto lap = [&](int slot, unsigned long long count) {
        if (!stamps) return;
        const unsigned long long now = wall_clock64();
#ifndef KKT_XSTAMPS
        if (tid == 0) {
            sstat[slot] += now - t_mark;
            sstat[slot + 3] += count;
        }
#else
        // -DKKT_XSTAMPS: the slots hold the parts of the coarse exchange instead
        // (scripts/r03_exchange_stamps.py)
        if (tid == 0 && slot == 0) sstat[3] += count;
#endif
        t_mark = now;
    };

    // publish the own rows of the newest (and the previous) iterate, re-gather the rings
    // The fields of a level are uniform values behind scalar loads, and the first reader of a
    // line waits for it (1-2 us per dependent batch, every workgroup at the same moment; measured:
    // 2.8 us of a level's first step).  lgkmcnt is shared with LDS, so the wait cannot be pushed
    // past the next LDS barrier -- but a hand-off waits for memory anyway: what the NEXT level
    // needs early is read at the start of the current level's first hand-off, one field from
    // each of the two 64-byte lines of its struct, which the level's own reads then find cached.
    static_assert(sizeof(TileLevel) == 128, "two scalar-cache lines per level");
    LevEarly Nf{};
    bool nf_valid = false;
    const TileLevel *nf_want = nullptr;
    // `early`: the newest iterate's own rows went out from the step that computed them (the last
    // step of the round), under that step's remaining work; only the previous iterate is left
    auto handoff = [&](const bool both, const bool early = false) {
        lap(1, 0);
        if (nf_want != nullptr) {
            Nf = read_early(nf_want);
            nf_want = nullptr;
            nf_valid = true;
        }
        ++epoch;
        double *Xc = X + cur * nkp, *Xo = X + (cur ^ 1) * nkp;
        const __amdgpu_buffer_rsrc_t rn = __builtin_amdgcn_make_buffer_rsrc(
            (void *)A.gnew[epoch & 1], 0, (int)A.granule_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
            (void *)A.gold[epoch & 1], 0, (int)A.granule_bytes, 0x00020000);
#pragma unroll
        for (int sl = 0; sl < RPT; ++sl) {
            const int r = sl * T + tid;
            if (r < n0 && !(tile == 0 && A.debug_drop > 0 && (int)(epoch - A.epoch0) == A.debug_drop)) {
                if (!early) publish(rn, gr[sl], Xc[r], epoch);
                // (at depth 1 no ring row is ever computed: nobody reads the previous iterate)
                if (both && depth > 1) publish(ro, gr[sl], Xo[r], epoch);
            }
        }
        // a granule {lo, tag, hi, tag} is read back by ONE 16-byte sc1 load (half the requests of
        // two 8-byte loads; a torn read shows as unequal tags and is simply read again)
        u32x4 gn4[HPT], go4[HPT];
        bool want_o[HPT];
#pragma unroll
        for (int h = 0; h < HPT; ++h) {
            want_o[h] = both && hl[h] >= 0 && hl[h] < nk1;
            gn4[h] = u32x4{0u, 0u, 0u, 0u};         // tag 0 never matches a hand-off number
            go4[h] = u32x4{0u, 0u, 0u, 0u};
        }
        // a poll samples memory about half a round trip after it is issued; the neighbours'
        // granules, stored at about the same time as this tile's, take about one: polling at
        // once mostly fails and costs a second round trip
        for (int i = 0; i < A.poll_delay; ++i) __builtin_amdgcn_s_sleep(1);
        unsigned spins = 0;
        while (true) {
            asm volatile("" ::: "memory");   // the loads below are re-issued every round
            bool ok = true;
            // only what has not arrived yet is requested again
#pragma unroll
            for (int h = 0; h < HPT; ++h) {
                if (hl[h] >= 0 && (gn4[h].y != epoch || gn4[h].w != epoch))
                    gn4[h] = __builtin_amdgcn_raw_buffer_load_b128(rn, hg[h] * 16, 0, 16 /* sc1 */);
                if (want_o[h] && (go4[h].y != epoch || go4[h].w != epoch))
                    go4[h] = __builtin_amdgcn_raw_buffer_load_b128(ro, hg[h] * 16, 0, 16 /* sc1 */);
            }
#pragma unroll
            for (int h = 0; h < HPT; ++h) {
                if (hl[h] >= 0) ok &= gn4[h].y == epoch && gn4[h].w == epoch;
                if (want_o[h]) ok &= go4[h].y == epoch && go4[h].w == epoch;
            }
            if (ok || dead) break;
            if (++spins >= TILE_SPIN_LIMIT) {
                // record who waited for what: tile, epoch, local index, global row, tags seen
                dead = true;
                sdead = 1;
                atomicOr(A.err, 4u);
                if (atomicCAS(A.err + 1, 0u, 1u) == 0u) {
#pragma unroll
                    for (int h = 0; h < HPT; ++h)
                        if (hl[h] >= 0 && (gn4[h].y != epoch || gn4[h].w != epoch ||
                                           (want_o[h] && (go4[h].y != epoch || go4[h].w != epoch)))) {
                            A.err[8] = (unsigned)tile;
                            A.err[9] = epoch;
                            A.err[10] = (unsigned)hl[h];
                            A.err[11] = (unsigned)hg[h];
                            A.err[12] = gn4[h].y;
                            A.err[13] = gn4[h].w;
                            A.err[14] = go4[h].y;
                            A.err[15] = go4[h].w;
                            A.err[16] = both ? 1u : 0u;
                        }
                }
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int h = 0; h < HPT; ++h) {
            if (hl[h] >= 0)
                Xc[hl[h]] = __longlong_as_double(
                    (long long)((unsigned long long)gn4[h].x | ((unsigned long long)gn4[h].z << 32)));
            if (want_o[h])
                Xo[hl[h]] = __longlong_as_double(
                    (long long)((unsigned long long)go4[h].x | ((unsigned long long)go4[h].z << 32)));
        }
        // a wave that leaves this barrier knows every wave of the workgroup has finished reading
        // the granules of this epoch; only then may anyone publish the next one
        lds_barrier();
        dead = sdead != 0;
#ifndef KKT_XSTAMPS
        if (stamps && tid == 0) sstat[6] += spins;
#endif
        lap(0, 1);
    };

    // Operands of a level that live in registers through its steps: matrix values (re-loaded
    // only when the pointer changes: per level in mode G, once in mode S), Jacobi diagonal,
    // right-hand side.  Called for level 0 up front and for level l + 1 BEFORE the hand-off that
    // ends level l, so that the HBM fetch (every tile asks for its slice of the next matrix at
    // the same moment) runs under the hand-off's wait instead of after it.
    double dinv[RPT], b[RPT];
    // (wide rows: the positions of all slots first, then the values -- two dependent round trips
    // instead of two per chunk of five entries)
    auto load_vals = [&](const double *ptr, const bool update_matrix) {
        vals_key = (const void *)ptr;
        const gcd_p vp = (gcd_p)ptr;
        if constexpr (PACK) {
            // (the addresses of the positions must not be hoisted out of the level loop: they
            // are 2 W registers per row slot that nothing else could use -- 58 of the 245
            // registers of the two-slot 3-D variant)
            const gci_p gpos = opaque_ptr(gpos_base);
            constexpr int G = RPT * W > 30 ? 1 : RPT;   // slots whose positions are fetched together
#pragma unroll
            for (int s0 = 0; s0 < RPT; s0 += G) {
                int gp[G][W];
#pragma unroll
                for (int q = 0; q < G; ++q)
#pragma unroll
                    for (int k = 0; k < W; ++k) gp[q][k] = KKT_GP(s0 + q, k);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < G; ++q)
#pragma unroll
                    for (int k = 0; k < W; ++k)
                        v[s0 + q][k] = (gp[q][k] >= 0 && (!update_matrix || gr[s0 + q] >= 0))
                                           ? vp[gp[q][k]] : 0.0;
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int sl = 0; sl < RPT; ++sl)
#pragma unroll
                for (int k = 0; k < W; ++k) {
                    const int gp = KKT_GP(sl, k);
                    v[sl][k] = (gp >= 0 && (!update_matrix || gr[sl] >= 0)) ? vp[gp] : 0.0;
                }
        }
    };
    auto load_db = [&](const LevEarly &L) {
        const gcd_p dp = (gcd_p)L.dinv, bp = (gcd_p)L.bin;
#pragma unroll
        for (int sl = 0; sl < RPT; ++sl) {
            dinv[sl] = gr[sl] >= 0 ? dp[gr[sl]] : 0.0;
            b[sl] = gr[sl] >= 0 ? bp[gr[sl]] : 0.0;
        }
    };
    auto load_level = [&](const LevEarly &L) {
        if ((const void *)L.vals != vals_key) load_vals(L.vals, false);
        load_db(L);
    };
    // Wide rows with the level update in the kernel (SEQ): there are no registers for a second
    // matrix, so the update's matrix passes through the registers of the level matrix -- it is
    // requested in front of the hand-off that ends a level (the old matrix is dead by then),
    // multiplied after it, and only then is the new level's own matrix requested (one exposed
    // fetch per level instead of a kernel boundary, a plain update launch and the per-launch
    // structure loads).
    constexpr bool SEQ = PACK && UPD;
    bool v_is_u = false;
    if (A.nlevels > 0) {
        const LevEarly L0 = read_early(&levels[0]);
        if (SEQ && L0.n_upd > 0) {
            load_vals(L0.upd0, true);
            load_db(L0);
            v_is_u = true;
        } else {
            load_level(L0);
        }
    }

    // Narrow rows leave registers for the NEXT level's operands (its matrix, the matrix of its
    // update, diagonal, right-hand side): they are requested when the last round of local steps
    // of a level begins and land under those steps.  Requested at the level's end they cost an
    // HBM round trip per level twice over -- once in front of the hand-off's polls (loads return
    // in order), once in front of the update (measured: 3 us of level prologue in mode G).
    // Variants without that room (1 024 threads: 128 registers) still keep the values of the next
    // update's matrix out of the level's first step: they are requested together with the next
    // matrix in front of the hand-off that ends the level (PRE_U).
    // (two-grid levels are short -- a correction and a handful of sweeps -- and their exchanges
    // need the registers: no operands of the next level are held across them)
    constexpr bool PRE = UPD && !COARSE && W * RPT <= 14 && TMAX <= 512;
    constexpr bool PRE_U = UPD && !COARSE && !PRE && W * RPT <= 7;
    constexpr int PR = PRE ? RPT : 1, PW = PRE ? W : 1;
    constexpr int UR = (PRE || PRE_U) ? RPT : 1, UW = (PRE || PRE_U) ? W : 1;
    double vn[PR][PW], un[UR][UW], dn[PR], bn[PR];
    bool pre_v = false, pre_u = false, pre_issued = false, un_loaded = false;
    auto load_un = [&](const LevEarly &N) {
        if constexpr (PRE || PRE_U) {
            pre_u = N.n_upd > 0;
            if (pre_u) {
                const gcd_p up = (gcd_p)N.upd0;
#pragma unroll
                for (int sl = 0; sl < RPT; ++sl)
#pragma unroll
                    for (int k = 0; k < W; ++k) {
                        const int gp = KKT_GP(sl, k);
                        un[sl][k] = (gr[sl] >= 0 && gp >= 0) ? up[gp] : 0.0;
                    }
            }
            un_loaded = true;
        }
    };
    auto prefetch_level = [&](const LevEarly &N) {
        if constexpr (PRE) {
            pre_issued = true;
            pre_v = (const void *)N.vals != vals_key;
            if (pre_v) {
                const gcd_p vp = (gcd_p)N.vals;
#pragma unroll
                for (int sl = 0; sl < RPT; ++sl)
#pragma unroll
                    for (int k = 0; k < W; ++k) {
                        const int gp = KKT_GP(sl, k);
                        vn[sl][k] = gp >= 0 ? vp[gp] : 0.0;
                    }
            }
            load_un(N);
            const gcd_p dp = (gcd_p)N.dinv, bp = (gcd_p)N.bin;
#pragma unroll
            for (int sl = 0; sl < RPT; ++sl) {
                dn[sl] = gr[sl] >= 0 ? dp[gr[sl]] : 0.0;
                bn[sl] = gr[sl] >= 0 ? bp[gr[sl]] : 0.0;
            }
        }
    };
    auto adopt_level = [&](const LevEarly &N) {
        if constexpr (PRE) {
            if (pre_v) {
                vals_key = (const void *)N.vals;
#pragma unroll
                for (int sl = 0; sl < RPT; ++sl)
#pragma unroll
                    for (int k = 0; k < W; ++k) v[sl][k] = vn[sl][k];
            }
#pragma unroll
            for (int sl = 0; sl < RPT; ++sl) {
                dinv[sl] = dn[sl];
                b[sl] = bn[sl];
            }
        }
    };

    for (int lev = 0; lev < A.nlevels; ++lev) {
        const TileLevel &L = levels[lev];
        nf_valid = false;
        nf_want = lev + 1 < A.nlevels ? &levels[lev + 1] : nullptr;
        const bool have_u = (PRE || PRE_U) && un_loaded && pre_u;   // un = values of upd_vals[0]
        const bool use_v = SEQ && v_is_u;                            // v = values of upd_vals[0]
        pre_issued = false;
        un_loaded = false;
        v_is_u = false;
        if ((const void *)L.coef != coef_key) {
            coef_key = (const void *)L.coef;
            const gcd_p cp = (gcd_p)(const double *)L.coef;
            __syncthreads();   // nobody still reads the old table
            for (int i = tid; i < 3 * (COARSE ? its : its - 1); i += T) scoef[i] = cp[i];
            // (the barrier after the first step of the level orders these writes before their use)
        }
        // Everything loaded for the level is pinned in registers here: a value whose load is
        // conditional would otherwise be waited for at its first use INSIDE the step loop
        // (s_waitcnt vmcnt(0) there would also drain whatever the loop keeps in flight).
#pragma unroll
        for (int sl = 0; sl < RPT; ++sl) {
#pragma unroll
            for (int k = 0; k < W; ++k) asm volatile("" : "+v"(v[sl][k]));
            asm volatile("" : "+v"(dinv[sl]));
            asm volatile("" : "+v"(b[sl]));
        }
        if (!COARSE && stamps && tid == 0) sstat[7] += wall_clock64() - t_mark;   // prologue up to the update
        const double post1 = L.post1, post2 = L.post2;
        int cr;
        {
            double *Xc = X + cur * nkp, *Xo = X + (cur ^ 1) * nkp;
            if (UPD && L.n_upd > 0) {
                if (!L.prev_in_lds) {
                    // produced before this launch (or by another rank): plain memory
                    const gcd_p xp = (gcd_p)L.x_prev;
                    for (int l = tid; l < nk; l += T) Xc[l] = xp[grow[l]];
                    lds_barrier();
                }
                // b = ca * (sum_t U_t x_prev) + cy * b_in, 0 on boundary rows; p_1 = scale D^-1 b
                double acc[RPT];
#pragma unroll
                for (int sl = 0; sl < RPT; ++sl) acc[sl] = 0.0;
                if constexpr (PRE || PRE_U) {
                    if (have_u) {
                        // all gathers in flight before the first fma (the chains of the slots
                        // then run side by side instead of one LDS round trip per entry)
                        double xu[UR][UW];
#pragma unroll
                        for (int sl = 0; sl < RPT; ++sl)
#pragma unroll
                            for (int k = 0; k < W; ++k) xu[sl][k] = Xc[KKT_COL(sl, k)];
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int k = 0; k < W; ++k)
#pragma unroll
                            for (int sl = 0; sl < RPT; ++sl)
                                acc[sl] = __builtin_fma(un[sl][k], xu[sl][k], acc[sl]);
                    }
                }
                if constexpr (SEQ) {
                    if (use_v) {
#pragma unroll
                        for (int sl = 0; sl < RPT; ++sl) {
                            if (sl * T + (tid & ~63) >= nk1) continue;
#pragma unroll
                            for (int k0 = 0; k0 < W; k0 += CH) {
                                double xv[CH];
#pragma unroll
                                for (int k = 0; k < CH; ++k)
                                    if (k0 + k < W) xv[k] = Xc[KKT_COL(sl, k0 + k)];
#pragma unroll
                                for (int k = 0; k < CH; ++k)
                                    if (k0 + k < W)
                                        acc[sl] = __builtin_fma(v[sl][k0 + k], xv[k], acc[sl]);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    }
                }
                // (SEQ: one update term, the host's condition for this variant -- with this
                // loop compiled in, the two-slot variant spills 88 registers)
                if constexpr (!SEQ)
                for (int t = have_u ? 1 : 0; t < L.n_upd; ++t) {
                    const gcd_p up = (gcd_p)L.upd_vals[t];
#pragma unroll
                    for (int sl = 0; sl < RPT; ++sl) {
                        if (sl * T + (tid & ~63) >= nk1) continue;
#pragma unroll
                        for (int k0 = 0; k0 < W; k0 += CH) {
                            int gp[CH];
                            double vu[CH];
#pragma unroll
                            for (int k = 0; k < CH; ++k)
                                if (k0 + k < W) gp[k] = KKT_GP(sl, k0 + k);
#pragma unroll
                            for (int k = 0; k < CH; ++k)
                                if (k0 + k < W)
                                    vu[k] = (gr[sl] >= 0 && gp[k] >= 0) ? up[gp[k]] : 0.0;
#pragma unroll
                            for (int k = 0; k < CH; ++k)
                                if (k0 + k < W)
                                    acc[sl] = __builtin_fma(vu[k], Xc[KKT_COL(sl, k0 + k)], acc[sl]);
                            if constexpr (PACK) __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
                const gd_p bo = (gd_p)L.bout;
#pragma unroll
                for (int sl = 0; sl < RPT; ++sl) {
                    const int r = sl * T + tid;
                    double out;
                    if (msk[sl]) {
                        out = 0.0;
                    } else {
                        double t = L.ca * acc[sl];
                        t += L.cy * b[sl];
                        out = t;
                    }
                    const double out2 = L.p1_scale * (dinv[sl] * out);
                    b[sl] = out;
                    if (r < nk1) Xo[r] = out2;
                    if (r < n0 && bo != nullptr) bo[gr[sl]] = out;
                }
                if constexpr (SEQ) {
                    // the level's own matrix, now that the update's has been used
                    if ((const void *)L.vals != vals_key) load_vals(L.vals, false);
                }
            } else {
                // first step of a solve on a right-hand side that is already final
                const bool last = its == 1;
                const double q1 = last ? post1 : 1.0, q2 = last ? post2 : 1.0;
#pragma unroll
                for (int sl = 0; sl < RPT; ++sl) {
                    const int r = sl * T + tid;
                    double out = 0.0;
                    if (!msk[sl]) {
                        double t = 0.0;
                        t += L.p1_scale * (dinv[sl] * (b[sl] - 0.0));
                        out = q2 * (q1 * t);
                    }
                    if (r < nk1) Xo[r] = out;
                }
            }
            cr = depth - 1;
            lds_barrier();
            cur ^= 1;
        }
        const bool has_next = lev + 1 < A.nlevels;
        if (PRE && has_next && its - 1 <= cr) {
            Nf = read_early(&levels[lev + 1]);        // (a level without a hand-off of its own)
            nf_want = nullptr;
            nf_valid = true;
            prefetch_level(Nf);
        }
#pragma unroll
        for (int sl = 0; sl < RPT; ++sl) {
            if constexpr (SEQ) {
#pragma unroll
                for (int k = 0; k < W; ++k) asm volatile("" : "+v"(v[sl][k]));
            }
            asm volatile("" : "+v"(dinv[sl]));
            asm volatile("" : "+v"(b[sl]));
        }
        lap(2, 1);
        if constexpr (COARSE) {
            // ---- two-grid level: cycles of [coarse correction of the iterate; `its` sweeps]
            // One SpMV step of this path on the rows [0, nv): mode 0 the Chebyshev update
            // p+ = c1 p- + c2 p + c3 D^-1 (b - A p) into the older buffer, mode 1 the residual
            // b - A p.  The fma chain of a row is the plain kernels' (slot order).
            auto cstep = [&](const int mode, const int nv, const double cf1, const double cf2,
                             const double cf3, const bool has_old, const double q1,
                             const double q2) {
                double *Xc = X + cur * nkp, *Xo = X + (cur ^ 1) * nkp;
#pragma unroll
                for (int sl = 0; sl < RPT; ++sl) {
                    if (sl * T + (tid & ~63) >= nv) continue;      // wave-uniform
                    const int r = sl * T + tid;
                    double acc = 0.0;
#pragma unroll
                    for (int k0 = 0; k0 < W; k0 += CH) {
                        double xv[CH];
#pragma unroll
                        for (int k = 0; k < CH; ++k)
                            if (k0 + k < W) xv[k] = Xc[KKT_COL(sl, k0 + k)];
#pragma unroll
                        for (int k = 0; k < CH; ++k)
                            if (k0 + k < W) acc = __builtin_fma(v[sl][k0 + k], xv[k], acc);
                        if constexpr (PACK) __builtin_amdgcn_sched_barrier(0);
                    }
                    if (r < nv) {
                        double out = 0.0;
                        if (!msk[sl]) {
                            if (mode == 0) {
                                const double e0 = Xo[r], e1 = Xc[r];
                                double t = has_old ? cf1 * e0 : 0.0;
                                t += cf2 * e1;
                                t += cf3 * (dinv[sl] * (b[sl] - acc));
                                out = q2 * (q1 * t);
                            } else {
                                out = b[sl] - acc;
                            }
                        }
                        Xo[r] = out;
                    }
                }
            };
            const TileCoarseDev *cd = A.coarse;
            const gcd_p einv = (gcd_p)A.einv[lev];
            const int wave = tid >> 6, lane = tid & 63, nwaves = T >> 6;
            for (int cyc = 0; cyc < A.cycles; ++cyc) {
                // ---- residual of the current iterate on the own rows, into the older buffer
                if (cyc == 0) {
                    double *Xo = X + (cur ^ 1) * nkp;
#pragma unroll
                    for (int sl = 0; sl < RPT; ++sl) {
                        const int r = sl * T + tid;
                        if (r < n0) Xo[r] = b[sl];          // zero guess: r = b
                    }
                } else {
                    if (cr == 0) {                          // the first ring must be valid
                        handoff(false);
                        cr = depth;
                    }
                    cstep(1, n0, 0.0, 0.0, 0.0, false, 1.0, 1.0);
                }
                lds_barrier();
                lap(1, cyc == 0 ? 0 : 1);      // (stamps: the residual step counts as a step)
#ifdef KKT_XSTAMPS
                unsigned long long xs_t = wall_clock64();
                auto xs = [&](int slot) {
                    const unsigned long long n_ = wall_clock64();
                    if (tid == 0) sstat[slot] += n_ - xs_t;
                    xs_t = n_;
                };
#define KKT_XS(slot) xs(slot)
#else
#define KKT_XS(slot)
#endif
                // ---- restriction: partial sums over the own rows for the coarse functions they
                // touch, a wave per function, lanes stride its list, fixed butterfly
                ++cepoch;
                {
                    const double *Rb = X + (cur ^ 1) * nkp;
                    const __amdgpu_buffer_rsrc_t rc_ = __builtin_amdgcn_make_buffer_rsrc(
                        (void *)cd->cg[cepoch & 1], 0, (int)cd->cg_bytes, 0x00020000);
                    for (int k = wave; k < c_nj; k += nwaves) {
                        const int e0 = RIPc[k], e1 = RIPc[k + 1];
                        double a = 0.0;
                        if (c_cached) {
                            for (int e = e0 + lane; e < e1; e += 64)
                                a = __builtin_fma(RWc[e], Rb[RROWc[e]], a);
                        } else {
                            const gcu16_p rrow = (gcu16_p)cd->r_row + c_re0;
                            const gcd_p rw = (gcd_p)cd->r_w + c_re0;
                            for (int e = e0 + lane; e < e1; e += 64)
                                a = __builtin_fma(rw[e], Rb[rrow[e]], a);
                        }
#pragma unroll
                        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
                        if (lane == 0) publish(rc_, c_slot0 + k, a, cepoch);
                    }
                    // ---- every tile's partial sums ("the data is the flag"), then the coarse
                    // residual: contributions of a function summed in slot (= tile) order
                    // (4 slots per thread in flight; more slots than 4 T: further rounds, whose
                    // granules have mostly landed by then.  8 in flight made the compiler spill
                    // ~200 registers in the 512-thread variants.)
                    constexpr int CSL = 4;
                    KKT_XS(0);      // restriction + publish
                    for (int i = 0; i < A.poll_delay; ++i) __builtin_amdgcn_s_sleep(1);
                    for (int base = 0; base < c_nslots; base += CSL * T) {
                        u32x4 g[CSL];
#pragma unroll
                        for (int q = 0; q < CSL; ++q) g[q] = u32x4{0u, 0u, 0u, 0u};
                        unsigned spins = 0;
                        while (true) {
                            asm volatile("" ::: "memory");
                            bool ok = true;
#pragma unroll
                            for (int q = 0; q < CSL; ++q) {
                                const int sidx = base + q * T + tid;
                                if (sidx < c_nslots && (g[q].y != cepoch || g[q].w != cepoch))
                                    g[q] = __builtin_amdgcn_raw_buffer_load_b128(rc_, sidx * 16, 0, 16);
                            }
#pragma unroll
                            for (int q = 0; q < CSL; ++q) {
                                const int sidx = base + q * T + tid;
                                if (sidx < c_nslots) ok &= g[q].y == cepoch && g[q].w == cepoch;
                            }
                            if (ok || dead) break;
                            if (++spins >= TILE_SPIN_LIMIT) {
                                dead = true;
                                sdead = 1;
                                atomicOr(A.err, 4u);
                                if (atomicCAS(A.err + 1, 0u, 1u) == 0u) {
                                    A.err[8] = (unsigned)tile;
                                    A.err[9] = cepoch;
                                    A.err[10] = 0xffffu;            // a coarse exchange
                                    A.err[11] = (unsigned)(base + tid);
                                    A.err[12] = g[0].y;
                                    A.err[13] = g[0].w;
                                    A.err[16] = 2u;
                                }
                                break;
                            }
                            __builtin_amdgcn_s_sleep(1);
                        }
#pragma unroll
                        for (int q = 0; q < CSL; ++q) {
                            const int sidx = base + q * T + tid;
                            if (sidx < c_nslots)
                                SL[sidx] = __longlong_as_double((long long)(
                                    (unsigned long long)g[q].x | ((unsigned long long)g[q].z << 32)));
                        }
                    }
                    KKT_XS(1);      // polls of this thread
                    lds_barrier();
                    KKT_XS(2);      // barrier: the slowest thread's polls
                    dead = sdead != 0;
                    for (int j = tid; j < c_nc; j += T) {
                        double a = 0.0;
                        for (int q = CIPc[j]; q < CIPc[j + 1]; ++q) a += SL[CSLc[q]];
                        RC[j] = a;
                    }
                    lds_barrier();
                    KKT_XS(4);      // coarse residual
                    // ---- the coarse functions this tile owns (j = tile, tile + ntiles, ...): a wave
                    // per function, the product published as a granule of the same exchange number
                    const __amdgpu_buffer_rsrc_t re_ = __builtin_amdgcn_make_buffer_rsrc(
                        (void *)cd->eg[cepoch & 1], 0, (int)cd->eg_bytes, 0x00020000);
                    const int ntl = (int)gridDim.x;
                    // (row j only over the columns of its diagonal block, [e_lo[j], e_hi[j]) with
                    // e_lo a multiple of 64: a lane keeps the columns it had, the terms left out
                    // are exact zeros -- kernels.hpp, TileCoarseDev)
                    const gci_p elo = (gci_p)cd->e_lo, ehi = (gci_p)cd->e_hi;
                    if (c_einv_cache && (const void *)einv != einv_key) {
                        // (the iterates' barrier above ordered every earlier read of EINVc)
                        for (int i = wave; i < c_nown; i += nwaves) {
                            const int j = tile + i * ntl;
                            const int lo = i == wave ? c_lo0 : elo[j], hi = i == wave ? c_hi0 : ehi[j];
                            const gcd_p row = einv + (size_t)j * c_nc;
                            for (int q = lo + lane; q < hi; q += 64)
                                EINVc[(size_t)i * c_ew + (q - lo)] = row[q];
                        }
                        einv_key = (const void *)einv;
                        // (a wave reads back only the row it wrote: no barrier needed)
                    }
                    for (int i = wave; i < c_nown; i += nwaves) {
                        double a = 0.0;
                        const int j = tile + i * ntl;
                        const int lo = i == wave ? c_lo0 : elo[j], hi = i == wave ? c_hi0 : ehi[j];
                        if (c_einv_cache) {
                            // (reading six entries ahead of the fma chain was slower: 128 its/s
                            // against 133 -- the 1 024-thread variant has no registers to spare)
                            const double *row = EINVc + (size_t)i * c_ew;
                            for (int q = lo + lane; q < hi; q += 64)
                                a = __builtin_fma(row[q - lo], RC[q], a);
                        } else {
                            // rows from L2: eight loads of a lane in flight before the first fma
                            const gcd_p row = einv + (size_t)j * c_nc;
                            constexpr int EU = 8;
                            for (int q0 = lo + lane; q0 < hi; q0 += 64 * EU) {
                                double rv[EU];
#pragma unroll
                                for (int u = 0; u < EU; ++u) {
                                    const int q = q0 + 64 * u;
                                    rv[u] = q < hi ? row[q] : 0.0;
                                }
#pragma unroll
                                for (int u = 0; u < EU; ++u) {
                                    const int q = q0 + 64 * u;
                                    if (q < hi) a = __builtin_fma(rv[u], RC[q], a);
                                }
                            }
                        }
#pragma unroll
                        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
                        if (lane == 0) publish(re_, tile + i * ntl, a, cepoch);
                    }
                    KKT_XS(5);      // owned products + publish
                    // ---- the products for the functions the own rows touch, from their owners
                    for (int k0 = 0; k0 < c_nj; k0 += T) {
                        const int k = k0 + tid;
                        u32x4 g = u32x4{0u, 0u, 0u, 0u};
                        unsigned spins = 0;
                        while (true) {
                            asm volatile("" ::: "memory");
                            bool ok = true;
                            if (k < c_nj && (g.y != cepoch || g.w != cepoch)) {
                                g = __builtin_amdgcn_raw_buffer_load_b128(re_, JGc[k] * 16, 0, 16);
                                ok = g.y == cepoch && g.w == cepoch;
                            }
                            if (__all(ok) || dead) break;
                            if (++spins >= TILE_SPIN_LIMIT) {
                                dead = true;
                                sdead = 1;
                                atomicOr(A.err, 4u);
                                if (atomicCAS(A.err + 1, 0u, 1u) == 0u) {
                                    A.err[8] = (unsigned)tile;
                                    A.err[9] = cepoch;
                                    A.err[10] = 0xfffeu;            // products of a coarse exchange
                                    A.err[11] = (unsigned)(k < c_nj ? JGc[k] : 0);
                                    A.err[12] = g.y;
                                    A.err[13] = g.w;
                                    A.err[16] = 2u;
                                }
                                break;
                            }
                            __builtin_amdgcn_s_sleep(1);
                        }
                        if (k < c_nj)
                            EC[k] = __longlong_as_double((long long)(
                                (unsigned long long)g.x | ((unsigned long long)g.z << 32)));
                    }
                    lds_barrier();
                    dead = sdead != 0;
                    KKT_XS(6);      // products polled + barrier
                    // ---- prolongation onto the own rows of the current iterate
                    double *Xc = X + cur * nkp;
#pragma unroll
                    for (int sl = 0; sl < RPT; ++sl) {
                        const int r = sl * T + tid;
                        if (r < n0) {
                            double a = 0.0;
                            if (c_cached) {
                                for (int e = pe0[sl]; e < pe1[sl]; ++e)
                                    a = __builtin_fma(PWc[e], EC[PKc[e]], a);
                            } else {
                                const gcu16_p pk = (gcu16_p)cd->p_k + c_pq0;
                                const gcd_p pw = (gcd_p)cd->p_w + c_pq0;
                                for (int e = pe0[sl]; e < pe1[sl]; ++e)
                                    a = __builtin_fma(pw[e], EC[pk[e]], a);
                            }
                            Xc[r] = cyc == 0 ? a : Xc[r] + a;
                        }
                    }
                }
                // (the barrier inside the hand-off orders these stores before anyone's gathers)
                lds_barrier();
                if (stamps) {      // the coarse exchange, restriction to prolongation: slot 7
                    const unsigned long long now = wall_clock64();
                    if (tid == 0) sstat[7] += now - t_mark;
                    t_mark = now;
                }
                handoff(false);
                cr = depth;
                // ---- smoothing sweeps from the corrected iterate
                for (int s = 1; s <= its; ++s) {
                    if (cr == 0) {
                        handoff(s >= 2);
                        cr = depth;
                    }
                    const bool last = cyc + 1 == A.cycles && s == its;
                    cstep(0, sn[cr - 1], scoef[3 * (s - 1)], scoef[3 * (s - 1) + 1],
                          scoef[3 * (s - 1) + 2], s >= 2, last ? post1 : 1.0, last ? post2 : 1.0);
                    lds_barrier();
                    cur ^= 1;
                    --cr;
                }
            }
        } else {
            // ---- Chebyshev steps 2 .. its, `depth` of them per hand-off
            // (row count and coefficients of a step are read from LDS one step ahead: they land with
            // the barrier's own wait instead of in front of the step's gathers)
            int nv_next = sn[(cr == 0 ? depth : cr) - 1];
            double cn1 = scoef[0], cn2 = scoef[1], cn3 = scoef[2];
            bool early_done = false;   // the step before published the own rows already
            for (int s = 2; s <= its; ++s) {
                if (cr == 0) {
                    handoff(s >= 3, early_done);
                    early_done = false;
                    cr = depth;
                    // the last round of the level: the next level's operands travel under it
                    if (PRE && has_next && its - s < depth) prefetch_level(Nf);
                }
                const int nv = nv_next;
                const double cf1 = cn1, cf2 = cn2, cf3 = cn3;
                const bool last = s == its;
                const double q1 = last ? post1 : 1.0, q2 = last ? post2 : 1.0;
                const bool has_old = s >= 3;
                double *Xc = X + cur * nkp, *Xo = X + (cur ^ 1) * nkp;
                // a hand-off follows this step inside the level
                // (wide rows only -- on narrow rows at depth 5-7 the stores in the step cost more
                // than the hand-off gains: 0.45 -> 0.51 us per step, 3.58 -> 3.37 us per hand-off)
                const bool pub_next = PACK && cr == 1 && s < its;
                const bool pub_drop =
                    tile == 0 && A.debug_drop > 0 && (int)(epoch + 1 - A.epoch0) == A.debug_drop;
                early_done = pub_next;
                const __amdgpu_buffer_rsrc_t rn_next = __builtin_amdgcn_make_buffer_rsrc(
                    (void *)A.gnew[(epoch + 1) & 1], 0, (int)A.granule_bytes, 0x00020000);
                if constexpr (PACK) {
    #pragma unroll
                    for (int sl = 0; sl < RPT; ++sl) {
                        // wave-uniform: none of this wave's 64 rows of the slot is live on the
                        // shrunken region
                        if (sl * T + (tid & ~63) >= nv) continue;
                        const int r = sl * T + tid;
                        double acc = 0.0;
    #pragma unroll
                        for (int k0 = 0; k0 < W; k0 += CH) {
                            // (wide rows: a bounded number of gathers in flight, or the registers of
                            // the matrix values spill)
                            double xv[CH];
    #pragma unroll
                            for (int k = 0; k < CH; ++k)
                                if (k0 + k < W) xv[k] = Xc[KKT_COL(sl, k0 + k)];
    #pragma unroll
                            for (int k = 0; k < CH; ++k)
                                if (k0 + k < W) acc = __builtin_fma(v[sl][k0 + k], xv[k], acc);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        if (r < nv) {
                            const double e0 = Xo[r], e1 = Xc[r];
                            double out = 0.0;
                            if (!msk[sl]) {
                                double t = has_old ? cf1 * e0 : 0.0;
                                t += cf2 * e1;
                                t += cf3 * (dinv[sl] * (b[sl] - acc));
                                out = q2 * (q1 * t);
                            }
                            Xo[r] = out;
                            // last step of a round: the own rows leave at once, with the number of
                            // the hand-off that follows (the rest of the step runs under their flight)
                            if (pub_next && !pub_drop && r < n0)
                                publish(rn_next, gr[sl], out, epoch + 1);
                        }
                    }
                } else {
                    // Narrow rows: a step is a latency chain (gather, W dependent fmas, epilogue,
                    // store, barrier), not a throughput problem -- the slots of a thread run side by
                    // side: every gather of every live slot is issued before the first fma.  Live
                    // slots are a prefix (rows of a slot lie behind those of the one before); a wave
                    // none of whose 64 rows of a slot is live on the shrunken region skips the slot.
                    int nl = 0;
                    long long keep[RPT];
    #pragma unroll
                    for (int sl = 0; sl < RPT; ++sl) {
                        if (sl * T + (tid & ~63) < nv) nl = sl + 1;
                        keep[sl] = msk[sl] ? 0ll : -1ll;
                        asm volatile("" : "+v"(keep[sl]));
                    }
                    auto body = [&](auto NLc) {
                        constexpr int NL = decltype(NLc)::value;
                        double xv[NL][W], e0[NL], e1[NL], acc[NL];
    #pragma unroll
                        for (int sl = 0; sl < NL; ++sl) {
    #pragma unroll
                            for (int k = 0; k < W; ++k) xv[sl][k] = Xc[KKT_COL(sl, k)];
                            const int r = sl * T + tid;
                            e0[sl] = Xo[r];
                            e1[sl] = Xc[r];
                            acc[sl] = 0.0;
                        }
                        __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
                        for (int k = 0; k < W; ++k)
    #pragma unroll
                            for (int sl = 0; sl < NL; ++sl)
                                acc[sl] = __builtin_fma(v[sl][k], xv[sl][k], acc[sl]);
    #pragma unroll
                        for (int sl = 0; sl < NL; ++sl) {
                            const int r = sl * T + tid;
                            // (boundary rows: a bit mask, not a branch or a select the compiler
                            // turns into one -- a branch per slot would put the chains of the slots
                            // one behind the other again)
                            double t = has_old ? cf1 * e0[sl] : 0.0;
                            t += cf2 * e1[sl];
                            t += cf3 * (dinv[sl] * (b[sl] - acc[sl]));
                            const double out = __longlong_as_double(
                                __double_as_longlong(q2 * (q1 * t)) & keep[sl]);
                            // rows behind the live region store into a dump slot: an unconditional
                            // store keeps the compiler from sinking the chain into a branch per slot
                            *(r < nv ? Xo + r : dump) = out;
                        }
                    };
                    if (nl == 1) body(std::integral_constant<int, 1>{});
                    if constexpr (RPT >= 2)
                        if (nl == 2) body(std::integral_constant<int, 2>{});
                    if constexpr (RPT >= 3)
                        if (nl == 3) body(std::integral_constant<int, 3>{});
                }
                if (s < its) {
                    const int crn = cr - 1 == 0 ? depth : cr - 1;
                    nv_next = sn[crn - 1];
                    cn1 = scoef[3 * (s - 1)];
                    cn2 = scoef[3 * (s - 1) + 1];
                    cn3 = scoef[3 * (s - 1) + 2];
                }
                lds_barrier();
                cur ^= 1;
                --cr;
            }
        }
        lap(1, (unsigned long long)(its - 1));
        // ---- result of the level: own rows to memory; the next level's update multiplies it
        {
            double *Xc = X + cur * nkp;
            const gd_p op = (gd_p)L.out;
#pragma unroll
            for (int sl = 0; sl < RPT; ++sl) {
                const int r = sl * T + tid;
                if (r < n0) op[gr[sl]] = Xc[r];
            }
            if (lev + 1 < A.nlevels) {
                if (!nf_valid) Nf = read_early(&levels[lev + 1]);
                nf_want = nullptr;
                const LevEarly &N = Nf;
                if (!(PRE && pre_issued)) {
                    if (SEQ && N.n_upd > 0) {
                        load_vals(N.upd0, true);
                        load_db(N);
                        v_is_u = true;
                    } else {
                        load_level(N);
                        if constexpr (PRE_U) load_un(N);
                    }
                }
                if (N.n_upd > 0 && N.prev_in_lds) handoff(false);
                if (PRE && pre_issued) adopt_level(N);
            }
        }
    }
    if (stamps && tid == 0)
        for (int i = 0; i < 8; ++i) stat[i] += sstat[i];
}

#undef KKT_COL
#undef KKT_CPK
#undef KKT_GP

typedef void (*tile_fn)(const TileArgs, const TileLevel *, const int32_t *, const int32_t *,
                        const uint16_t *, const int32_t *, const uint8_t *);
// Register budget per thread by workgroup size: 512 threads -> 256, 1024 -> 128.  Narrow rows
// (2-D P1) run 512 threads with up to three row slots or 1 024 with one, level update fused.
// Wide rows (3-D P1: 15 entries, 30 registers of matrix values per row slot) run 512 threads with
// two slots in the variant without the fused update (244 registers; with it 256 + 78 spilled).
static tile_fn pick_tile_coarse(int W, int rpt, int threads, int hslots);
static tile_fn pick_tile(int W, int rpt, int threads, bool fused = true, int hslots = 0,
                         bool coarse = false) {
    if (coarse) return fused ? pick_tile_coarse(W, rpt, threads, hslots) : nullptr;
    if (hslots <= 0) hslots = 1;
#define KKT_T(w)                                                           \
    if (W == w) {                                                          \
        if (threads <= 512) {                                              \
            switch (rpt) {                                                 \
                case 1: return pc_tile_sweep<w, 1, 512, true>;             \
                case 2: return pc_tile_sweep<w, 2, 512, true>;             \
                case 3: return pc_tile_sweep<w, 3, 512, true>;             \
                default: return nullptr;                                   \
            }                                                              \
        }                                                                  \
        return rpt == 1 ? pc_tile_sweep<w, 1, 1024, true> : nullptr;       \
    }
    KKT_T(5) KKT_T(7) KKT_T(9)
#undef KKT_T
    // Wide rows: with the level update (one term, its matrix passing through the registers of
    // the level matrix) or without (the update stays a plain launch, one tile launch per level)
    // 3-D tiles whose rows and ring fit one slot each: 1 024 threads (16 waves hide the step's
    // latency chain better than 8 with two slots)
    if (W == 15 && threads > 512 && rpt == 1 && hslots <= 1)
        return fused ? pc_tile_sweep<15, 1, 1024, true, 1> : pc_tile_sweep<15, 1, 1024, false, 1>;
    if (W == 15 && threads <= 512) {
        if (rpt == 1) return fused ? pc_tile_sweep<15, 1, 512, true> : pc_tile_sweep<15, 1, 512, false>;
        if (rpt == 2) return fused ? pc_tile_sweep<15, 2, 512, true> : pc_tile_sweep<15, 2, 512, false>;
    }
    // P2 velocity blocks (9 or 19 entries per row, row-sorted storage): one row slot
    if (W == 19 && threads <= 512 && rpt == 1)
        return fused ? pc_tile_sweep<19, 1, 512, true> : pc_tile_sweep<19, 1, 512, false>;
    if (W == 19 && threads <= 512 && rpt == 2)      // 215 registers
        return fused ? pc_tile_sweep<19, 2, 512, true> : pc_tile_sweep<19, 2, 512, false>;
    if (W == 19 && threads > 512 && threads <= 768 && rpt == 1)
        return fused ? pc_tile_sweep<19, 1, 768, true> : pc_tile_sweep<19, 1, 768, false>;
    return nullptr;
}

// variants with coarse corrections (two-grid levels), level update fused
static tile_fn pick_tile_coarse(int W, int rpt, int threads, int hslots) {
    if (hslots <= 0) hslots = 1;
#define KKT_TC(w)                                                                  \
    if (W == w) {                                                                  \
        if (threads <= 512) {                                                      \
            switch (rpt) {                                                         \
                case 1: return pc_tile_sweep<w, 1, 512, true, 2, true>;            \
                case 2: return pc_tile_sweep<w, 2, 512, true, 3, true>;            \
                case 3: return pc_tile_sweep<w, 3, 512, true, 4, true>;            \
                default: return nullptr;                                           \
            }                                                                      \
        }                                                                          \
        return rpt == 1 ? pc_tile_sweep<w, 1, 1024, true, 2, true> : nullptr;      \
    }
    KKT_TC(5) KKT_TC(7) KKT_TC(9)
#undef KKT_TC
    if (W == 15 && threads > 512 && rpt == 1 && hslots <= 1)
        return pc_tile_sweep<15, 1, 1024, true, 1, true>;
    if (W == 15 && threads <= 512) {
        if (rpt == 1) return pc_tile_sweep<15, 1, 512, true, 2, true>;
        if (rpt == 2) return pc_tile_sweep<15, 2, 512, true, 3, true>;
    }
    if (W == 19 && threads <= 512 && rpt == 1) return pc_tile_sweep<19, 1, 512, true, 2, true>;
    if (W == 19 && threads <= 512 && rpt == 2) return pc_tile_sweep<19, 2, 512, true, 3, true>;
    // (two-grid levels are short chains of steps: 16 waves with one row each run a step's 19
    // gathers in half the time of 8 waves with two)
    if (W == 19 && threads > 512 && threads <= 768 && rpt == 1)
        return pc_tile_sweep<19, 1, 768, true, 2, true>;
    return nullptr;
}

bool tile_sweep_coarse_available(int W, int rpt, int threads, int hslots) {
    return threads >= 64 && threads <= 1024 && threads % 64 == 0 && hslots <= rpt + 1 &&
           pick_tile_coarse(W, rpt, threads, hslots) != nullptr;
}

bool tile_sweep_fuses_update(int W, int max_terms) {
    return W <= 9 || ((W == 15 || W == 19) && max_terms <= 1);
}

size_t tile_sweep_lds_bytes(int nk_pad, int its, int coarse_nc, int coarse_nslots, int coarse_jmax,
                            int coarse_nr_max, int coarse_einv_rows, int coarse_einv_width) {
    if (coarse_nc > 0 && coarse_einv_rows > 0)
        return tile_sweep_lds_bytes(nk_pad, its, coarse_nc, coarse_nslots, coarse_jmax,
                                    coarse_nr_max, 0) +
               (size_t)coarse_einv_rows * (coarse_einv_width > 0 ? coarse_einv_width : coarse_nc) *
                   sizeof(double) + 8;
    if (coarse_nc > 0)
        return (2 * (size_t)nk_pad + 3 * (size_t)std::max(1, its) + 1 + (size_t)coarse_nslots +
                (size_t)coarse_nc + (size_t)coarse_jmax + 2 * (size_t)coarse_nr_max) * sizeof(double) +
               ((size_t)coarse_nslots + (size_t)coarse_nc + 1 + 2 * (size_t)coarse_jmax + 1) * sizeof(int) +
               2 * (size_t)coarse_nr_max * sizeof(uint16_t) + 16;
    return (2 * (size_t)nk_pad + 3 * (size_t)std::max(1, its - 1) + 1) * sizeof(double);
}

int tile_sweep_max_rpt(int W, int threads) {
    int best = 0;
    for (int rpt = 1; rpt <= 4; ++rpt)
        if (pick_tile(W, rpt, threads, true, 1)) best = rpt;
    return best;
}

int tile_sweep_max_hslots(int W, int rpt, int threads) {
    int best = 0;
    for (int h = 1; h <= rpt + 1; ++h)
        if (pick_tile(W, rpt, threads, true, h)) best = h;
    return best;
}

bool tile_sweep_available(int W, int rpt, int threads, int hslots) {
    return threads >= 64 && threads <= 1024 && threads % 64 == 0 && hslots <= rpt + 1 &&
           pick_tile(W, rpt, threads, true, hslots) != nullptr;
}

int tile_sweep_max_tiles(int W, int rpt, int threads, size_t lds_bytes, int hslots, bool coarse) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    // both variants of a width (with / without the level update) must be resident
    // (coarse: the one variant with corrections)
    for (int fused = coarse ? 1 : 0; fused < 2; ++fused) {
        tile_fn f = pick_tile(W, rpt, threads, fused != 0, hslots, coarse);
        int per_cu = 0;
        if (!f) return 0;
        if (lds_bytes > 48 * 1024 &&
            hipFuncSetAttribute((const void *)f, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_bytes) != hipSuccess)
            return 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, f, threads, lds_bytes) != hipSuccess)
            return 0;
        if (per_cu < 1) return 0;
    }
    return cus;   // one workgroup per CU
}

void launch_tile_sweep(hipStream_t s, const TileArgs &a, const TileLevel *d_levels,
                       const int32_t *d_n, const int32_t *d_grow, const uint16_t *d_lcol,
                       const int32_t *d_gpos, const uint8_t *d_rowmask, int ntiles, int threads,
                       size_t granule_words, const TileCoarseDev *h_coarse) {
    if (a.nlevels <= 0 || ntiles <= 0) return;
    // tags of an earlier application must not match this one's hand-off numbers
    auto chk = [](hipError_t e, const char *what) {
        if (e != hipSuccess)
            throw TileLaunchError{std::string("tile sweep program: ") + what + ": " +
                                  hipGetErrorString(e)};
    };
    (void)hipGetLastError();
    if (a.clear) {
        for (int i = 0; i < 2; ++i) {
            launch_zero_bytes(s, a.gnew[i], granule_words * sizeof(unsigned long long));
            launch_zero_bytes(s, a.gold[i], granule_words * sizeof(unsigned long long));
        }
        if (h_coarse)
            for (int i = 0; i < 2; ++i) {
                launch_zero_bytes(s, h_coarse->cg[i], h_coarse->cg_bytes);
                launch_zero_bytes(s, h_coarse->eg[i], h_coarse->eg_bytes);
            }
        chk(hipGetLastError(), "clearing the granule buffers");
    }
    const size_t lds = h_coarse ? tile_sweep_lds_bytes(a.nk_pad, a.its, h_coarse->nc,
                                                       h_coarse->nslots, h_coarse->jmax,
                                                       h_coarse->cache_lists ? h_coarse->nr_max : 0,
                                                       h_coarse->cache_einv ? h_coarse->nown : 0,
                                                       h_coarse->ew)
                                : tile_sweep_lds_bytes(a.nk_pad, a.its);
    tile_fn f = pick_tile(a.W, a.rpt, threads, a.fused_update != 0, a.hslots, h_coarse != nullptr);
    if (!f) throw TileLaunchError{"tile sweep program: no kernel variant for this plan"};
    (void)hipGetLastError();
    hipLaunchKernelGGL(f, dim3(ntiles), dim3(threads), lds, s, a, d_levels, d_n, d_grow, d_lcol,
                       d_gpos, d_rowmask);
    chk(hipGetLastError(), "launch");
}

}  // namespace kkt
