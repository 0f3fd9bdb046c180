// Hand-written HIP kernels of libkkt for gfx950 (MI355X, CDNA4): 64-wide wavefronts,
// HBM-bound fp64 streaming.  No MFMA: the path is sparse, ~0.17 flop/byte.
#include "kernels.hpp"

#include <hip/hip_runtime.h>

#include <cstddef>

namespace kkt {

typedef double d2 __attribute__((ext_vector_type(2)));
typedef int i2 __attribute__((ext_vector_type(2)));
// vector blocks start at multiples of nx doubles: only 8-byte alignment is guaranteed
typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));

// Pointers that arrive inside RowOp descriptors (loaded from memory) are generic to the
// compiler, which then emits flat_load/flat_store: those count on vmcnt AND lgkmcnt and
// force full `s_waitcnt vmcnt(0) lgkmcnt(0)` drains, and are not a valid hand-off form
// (Guideline 16).  Everything here is hipMalloc memory: cast to the global address space.
#define KKT_GLOBAL __attribute__((address_space(1)))
typedef KKT_GLOBAL const double *gcd_p;
typedef KKT_GLOBAL double *gd_p;
typedef KKT_GLOBAL const int32_t *gci_p;
typedef KKT_GLOBAL const uint8_t *gcb_p;
typedef KKT_GLOBAL const d2 *gcd2_p;
typedef KKT_GLOBAL const i2 *gci2_p;

__device__ __forceinline__ gcd_p resolve(const VRef &r, const Bases &B) {
    if (r.base < 0) return nullptr;
    if (r.base == 0) return (gcd_p) reinterpret_cast<const double *>(r.off);
    const double *p = r.base == 1 ? B.p[0] : r.base == 2 ? B.p[1] : r.base == 3 ? B.p[2] : B.p[3];
    return (gcd_p)(p + r.off);
}

// Buffers that must be zero before a persistent launch (tags, flags) are cleared by a KERNEL,
// not by hipMemsetAsync: inside a captured preconditioner application that is a kernel node
// instead of a memset node (rocprofv3 of ROCm 7.2 crashed in hipGraphLaunch of the graphs that
// held memset nodes next to the persistent kernels; a graph of kernel nodes only is the
// conservative shape).
void launch_zero_bytes(hipStream_t s, void *p, size_t nbytes);

// ---------------------------------------------------------------- fused block-row SpMV
//
// Layout ("SELL-64R"): rows are cut into slices of C = 64*R consecutive rows; slice s
// has width w_s = max row length in it and owns slots [slice_off[s], slice_off[s+1]).
// Entry k of row (s*C + q*64 + lane) sits at (slice_off[s] + k)*C + lane*R + q, so one
// wave-instruction reads 64*R consecutive values (R=2: 16 B per lane, 1 KiB per wave --
// the widest coalesced access); lane l owns rows l and l + 64 of the slice, so each x
// gather of a wave touches the columns of 64 consecutive rows (half the cache lines of
// an adjacent-row pairing) and y is stored as R coalesced 512-byte rows.
//
// One workgroup = 4 waves = 4 slices; blockIdx.y picks the RowOp (block row).  All RowOp
// fields are wave-uniform and come in through scalar loads.
//
// NT = true (the KKT operator apply): matrix values are streamed once with non-temporal
// loads so they do not evict the shared index array and the x windows from the XCD's L2.
// NT = false (preconditioner steps): the same few MB of matrix are re-read by every
// Chebyshev step of a sweep and should stay in L2.
//
// The per-row sum is term-major with one fma chain per row in CSR order (the order of
// the reference's MatMultAdd sequence, preconditioner.py:406-432).  For slice widths up
// to 16 the loop is fully unrolled at the exact width so that all value loads of a term,
// then all of its x gathers, are in flight together: small launches (one 66k-row block in
// the preconditioner sweeps) are latency-bound and this removes dependent round trips.

// COH = true (persistent row programs): vector operands are exchanged between workgroups
// inside one launch, so every vector load is an agent-scope relaxed atomic load
// (global_load ... sc1: bypasses this CU's L1, served by L2) and every vector store an
// agent-scope relaxed atomic store (write-through) -- the R1 hand-off form of
// cdna_hip_programming.md Guideline 16 that needs no acquire fence.
template <bool COH>
__device__ __forceinline__ double ldv(gcd_p p) {
    if constexpr (COH) {
        const unsigned long long u = __hip_atomic_load(
            (KKT_GLOBAL unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return __longlong_as_double((long long)u);
    } else {
        return *p;
    }
}
template <bool COH>
__device__ __forceinline__ void stv(gd_p p, double v) {
    if constexpr (COH) {
        __hip_atomic_store((KKT_GLOBAL unsigned long long *)p,
                           (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    } else {
        *p = v;
    }
}

template <int R, bool NT>
__device__ __forceinline__ void load_vals(gcd_p p, double (&v)[R]) {
    if constexpr (R == 2) {
        d2 t;
        if constexpr (NT)
            t = __builtin_nontemporal_load((gcd2_p)p);
        else
            t = *(gcd2_p)p;
        v[0] = t.x;
        v[1] = t.y;
    } else {
        if constexpr (NT)
            v[0] = __builtin_nontemporal_load(p);
        else
            v[0] = *p;
    }
}

template <int R>
__device__ __forceinline__ void load_cols(gci_p p, int (&c)[R]) {
    if constexpr (R == 2) {
        const i2 t = *(gci2_p)p;
        c[0] = t.x;
        c[1] = t.y;
    } else {
        c[0] = *p;
    }
}

// Data-flow synchronisation of the persistent programs (see pc_row_program_g below): the
// vector a phase produces also travels as {tag, 32 value bits} granules, two per row.
struct NoGran {
    static constexpr bool on = false;
    static constexpr bool hoist = false;
};
// Counter form of the persistent programs, slices without a compile-time width.  A phase of
// such a program was a chain of dependent round trips: slice offset -> row numbers -> row
// mask -> (sc1) own-row operands -> column indices -> (sc1) gathers, chunk by chunk.  What does
// not change while consecutive phases work on the same structure -- slice offset and width,
// row numbers, row mask, diagonal -- is kept in registers across phases, and the first chunk of
// column indices and matrix values of the phase is fetched before the wave waits for its
// neighbours, so that after the wait the own-row operands and the first gathers go out
// back to back.
constexpr int HOIST_KC = 12;
#ifndef RAGGED_CASES
#define RAGGED_CASES KKT_WX(4) KKT_WX(5) KKT_WX(7) KKT_WX(9) KKT_WX(12) KKT_WX(15) KKT_WX(19)
#endif
#ifndef RAGGED_CH
#define RAGGED_CH 8
#endif
template <int R>
struct Hoist {
    static constexpr bool on = false;
    static constexpr bool hoist = true;
    const void *key_slice, *key_perm, *key_mask, *key_dinv;   // what the cache was read for
    int key_uw, key_nrows;
    int off0, w;
    int row[R];
    bool masked[R];
    double dinv[R];
    int c[HOIST_KC][R];                 // chunk 0 of term 0 of the phase being prepared
    double v[HOIST_KC][R];
};
struct Gran {
    static constexpr bool on = true;
    static constexpr bool hoist = false;
    const unsigned long long *xg;   // granules written by the previous phase
    unsigned long long *yg;         // granules this phase writes
    unsigned ep_in, ep_out;         // tags to expect (0: gather plain memory) / to publish
    unsigned *err;
    bool dead;                      // a spin timed out: stop waiting, results invalid
};
constexpr unsigned GRAN_SPIN_LIMIT = 1u << 20;

// `terms(t)` yields the t-th SpmvTerm of the op: from the descriptor in memory, or unpacked
// from a wave-held copy (persistent programs), so that no descriptor array is indexed
// dynamically in registers.
// FENCE (the width-switched ragged kernel): a scheduling barrier at every chunk boundary keeps the
// compiler from hoisting the loads of later chunks over earlier ones, so that the allocation is
// the indices + three chunks whatever W is (unfenced it grows by 14 registers per slot: 250 at
// W = 16).
// NSEG > 1 (the width-switched kernel, slices of NSEG * W slots: the B blocks of the Stokes
// system are 2 x 19 wide): a slice is worked off in NSEG segments of W slots per term -- same fma
// chain (term-major, slots ascending), the W index registers re-loaded per (term, segment).
template <int R, bool NT, int W, bool COH, class TermFn, int CH = 8, bool FENCE = false,
          int NSEG = 1>
__device__ __forceinline__ void accumulate_exact(const RowOp &op, const TermFn &terms,
                                                 const Bases &bases, size_t base,
                                                 double (&acc)[R]) {
    constexpr int C = 64 * R;                   // CH: slots per register chunk
    // (buffer gathers in the fixed-width launches too -- 138 -> 124 registers at width 7, four
    // waves per SIMD instead of three -- change nothing: 0.279-0.283 ms either way on cfg 2)
    constexpr bool BUFG = FENCE;
    constexpr int NCH = (W + CH - 1) / CH;
    const gci_p colp = (gci_p)op.col + base;
    int c[W][R];
#pragma unroll
    for (int k = 0; k < W; ++k) load_cols<R>(colp + (size_t)k * C, c[k]);
    const int nterms = op.nterms * NSEG;        // (term, segment) pairs, term-major
    constexpr size_t SEG = (size_t)W * C;
    double vn[CH][R];                           // values of the next (term, chunk)
    {
        const gcd_p vp = (gcd_p)terms(0).vals + base;
#pragma unroll
        for (int k = 0; k < (W < CH ? W : CH); ++k) load_vals<R, NT>(vp + (size_t)k * C, vn[k]);
    }
    for (int t = 0; t < nterms; ++t) {
        const SpmvTerm tm = terms(t / NSEG);
        const gcd_p x = resolve(tm.x, bases);
        const gcd_p vcur = (gcd_p)tm.vals + base + (size_t)(t % NSEG) * SEG;
        const gcd_p vnext = (t + 1 < nterms) ? (gcd_p)terms((t + 1) / NSEG).vals + base +
                                                   (size_t)((t + 1) % NSEG) * SEG
                                             : vcur;
        if constexpr (NSEG > 1) {
            if (t > 0) {
#pragma unroll
                for (int k = 0; k < W; ++k)
                    load_cols<R>(colp + (size_t)(t % NSEG) * SEG + (size_t)k * C, c[k]);
            }
        }
        // (FENCE) gathers as buffer loads: uniform descriptor of x + one 32-bit byte offset per
        // gather -- global loads keep a 64-bit offset pair per slot alive across the term loop
        __amdgpu_buffer_rsrc_t xr;
        if constexpr (BUFG)
            xr = __builtin_amdgcn_make_buffer_rsrc((void *)(const double *)x, 0, -1, 0x00020000);
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            constexpr int dummy = 0;
            (void)dummy;
            const int k0 = ch * CH;
            const int n = (W - k0) < CH ? (W - k0) : CH;
            double v[CH][R], xv[CH][R];
#pragma unroll
            for (int k = 0; k < CH; ++k)
                if (k < n) {
#pragma unroll
                    for (int q = 0; q < R; ++q) {
                        v[k][q] = vn[k][q];
                        if constexpr (BUFG) {
                            typedef unsigned u2 __attribute__((ext_vector_type(2)));
                            const u2 g = __builtin_amdgcn_raw_buffer_load_b64(
                                xr, (unsigned)c[k0 + k][q] << 3, 0, 0);
                            xv[k][q] = __hiloint2double((int)g.y, (int)g.x);
                        } else {
                            xv[k][q] = ldv<COH>(x + c[k0 + k][q]);
                        }
                    }
                }
            // next chunk's (or next term's first chunk's) values go in flight now
            if (ch + 1 < NCH) {
                const int k1 = k0 + CH;
                const int n1 = (W - k1) < CH ? (W - k1) : CH;
#pragma unroll
                for (int k = 0; k < CH; ++k)
                    if (k < n1) load_vals<R, NT>(vcur + (size_t)(k1 + k) * C, vn[k]);
            } else if (t + 1 < nterms) {
#pragma unroll
                for (int k = 0; k < (W < CH ? W : CH); ++k)
                    load_vals<R, NT>(vnext + (size_t)k * C, vn[k]);
            }
#pragma unroll
            for (int k = 0; k < CH; ++k)
                if (k < n) {
#pragma unroll
                    for (int q = 0; q < R; ++q) acc[q] = __builtin_fma(v[k][q], xv[k][q], acc[q]);
                }
            if constexpr (FENCE) __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int R, bool NT, bool COH, class TermFn, class GranT>
__device__ __forceinline__ void accumulate_generic(const RowOp &op, const TermFn &terms,
                                                   const Bases &bases, size_t base, int w,
                                                   double (&acc)[R], GranT &gran) {
    constexpr int C = 64 * R;
    const gci_p colp = (gci_p)op.col + base;
    const int nterms = op.nterms;
    if constexpr (GranT::hoist) {
        // as the COH path below, with the chunk registers owned by the caller: chunk 0 of
        // term 0 was loaded by hoist_prepare before the wave waited for its neighbours
        static_assert(COH, "hoisted operands belong to the persistent programs");
        constexpr int KC = HOIST_KC;
        for (int t = 0; t < (w > 0 ? nterms : 0); ++t) {
            const SpmvTerm tm = terms(t);
            const gcd_p vp = (gcd_p)tm.vals + base;
            const gcd_p x = resolve(tm.x, bases);
            if (t > 0) {
#pragma unroll
                for (int k = 0; k < KC; ++k)
                    load_cols<R>(colp + (size_t)(k < w ? k : w - 1) * C, gran.c[k]);
            }
            for (int k0 = 0; k0 < w; k0 += KC) {
                double xv[KC][R];
                if (t > 0 || k0 > 0) {
#pragma unroll
                    for (int k = 0; k < KC; ++k) {
                        const int kk = k0 + k < w ? k0 + k : w - 1;
                        load_vals<R, NT>(vp + (size_t)kk * C, gran.v[k]);
                    }
                }
#pragma unroll
                for (int k = 0; k < KC; ++k)
#pragma unroll
                    for (int q = 0; q < R; ++q) xv[k][q] = ldv<COH>(x + gran.c[k][q]);
                if (k0 + KC < w) {
#pragma unroll
                    for (int k = 0; k < KC; ++k) {
                        const int kk = k0 + KC + k < w ? k0 + KC + k : w - 1;
                        load_cols<R>(colp + (size_t)kk * C, gran.c[k]);
                    }
                }
#pragma unroll
                for (int k = 0; k < KC; ++k)
#pragma unroll
                    for (int q = 0; q < R; ++q)
                        acc[q] = __builtin_fma(k0 + k < w ? gran.v[k][q] : 0.0, xv[k][q], acc[q]);
            }
        }
    } else if constexpr (COH) {
        // Persistent programs are latency-bound: a phase is a chain of dependent round trips
        // (indices -> gathers through L2 -> fma).  Whole chunks of KC slots are in flight at
        // once and the next chunk's indices travel with the current chunk's gathers; slots
        // past the slice width re-read the last valid slot with a zero value
        // (fma(0, x, acc) == acc), so there is no branch inside the chunk.
        constexpr int KC = GranT::on ? 8 : 12;   // granule polls hold 4 registers per value
        for (int t = 0; t < (w > 0 ? nterms : 0); ++t) {
            const SpmvTerm tm = terms(t);
            const gcd_p vp = (gcd_p)tm.vals + base;
            const gcd_p x = resolve(tm.x, bases);
            int c[KC][R];
#pragma unroll
            for (int k = 0; k < KC; ++k)
                load_cols<R>(colp + (size_t)(k < w ? k : w - 1) * C, c[k]);
            for (int k0 = 0; k0 < w; k0 += KC) {
                double v[KC][R], xv[KC][R];
                bool plain = true;
                if constexpr (GranT::on) plain = gran.ep_in == 0u;
#pragma unroll
                for (int k = 0; k < KC; ++k) {
                    const int kk = k0 + k < w ? k0 + k : w - 1;
                    load_vals<R, NT>(vp + (size_t)kk * C, v[k]);
                }
                if (plain) {
#pragma unroll
                    for (int k = 0; k < KC; ++k)
#pragma unroll
                        for (int q = 0; q < R; ++q) xv[k][q] = ldv<COH>(x + c[k][q]);
                }
                if constexpr (GranT::on) {
                    if (!plain) {
                        // "the data is the flag": re-read the chunk's granules until every
                        // tag carries the producing phase
                        typedef KKT_GLOBAL const unsigned long long *gcu64_p;
                        const gcu64_p xg = (gcu64_p)gran.xg;
                        unsigned spins = 0;
                        while (true) {
                            unsigned long long ga[KC][R], gb[KC][R];
#pragma unroll
                            for (int k = 0; k < KC; ++k)
#pragma unroll
                                for (int q = 0; q < R; ++q) {
                                    const gcu64_p g = xg + 2 * (size_t)c[k][q];
                                    ga[k][q] = __hip_atomic_load(g, __ATOMIC_RELAXED,
                                                                 __HIP_MEMORY_SCOPE_AGENT);
                                    gb[k][q] = __hip_atomic_load(g + 1, __ATOMIC_RELAXED,
                                                                 __HIP_MEMORY_SCOPE_AGENT);
                                }
                            bool ok = true;
#pragma unroll
                            for (int k = 0; k < KC; ++k)
#pragma unroll
                                for (int q = 0; q < R; ++q) {
                                    ok &= (unsigned)(ga[k][q] >> 32) == gran.ep_in &&
                                          (unsigned)(gb[k][q] >> 32) == gran.ep_in;
                                    xv[k][q] = __longlong_as_double((long long)(
                                        (ga[k][q] & 0xffffffffull) | (gb[k][q] << 32)));
                                }
                            if (__all(ok) || gran.dead) break;
                            if (++spins >= GRAN_SPIN_LIMIT) {
                                gran.dead = true;
                                if ((threadIdx.x & 63) == 0) atomicOr(gran.err, 2u);
                                break;
                            }
                        }
                    }
                }
                if (k0 + KC < w) {
#pragma unroll
                    for (int k = 0; k < KC; ++k) {
                        const int kk = k0 + KC + k < w ? k0 + KC + k : w - 1;
                        load_cols<R>(colp + (size_t)kk * C, c[k]);
                    }
                }
#pragma unroll
                for (int k = 0; k < KC; ++k)
#pragma unroll
                    for (int q = 0; q < R; ++q)
                        acc[q] = __builtin_fma(k0 + k < w ? v[k][q] : 0.0, xv[k][q], acc[q]);
            }
        }
    } else {
        // (measured on the Stokes operator, same box: chunks of 10 slots with all loads of a chunk
        // in flight and padded tails ran exactly as fast as this loop, 0.886 against 0.889 ms, and
        // chunks guarded by scalar branches slower, 1.11 ms -- the waves of this kernel are not
        // short of memory-level parallelism)
        for (int t = 0; t < nterms; ++t) {
            const SpmvTerm tm = terms(t);
            const gcd_p vp = (gcd_p)tm.vals + base;
            const gcd_p x = resolve(tm.x, bases);
#pragma unroll 4
            for (int k = 0; k < w; ++k) {
                int c[R];
                double v[R];
                load_cols<R>(colp + (size_t)k * C, c);
                load_vals<R, NT>(vp + (size_t)k * C, v);
#pragma unroll
                for (int q = 0; q < R; ++q)
                    acc[q] = __builtin_fma(v[q], ldv<COH>(x + c[q]), acc[q]);
            }
        }
    }
}

// WFIX > 0: the launcher knows every slice of every RowOp in the launch has width WFIX
// (structured meshes: 7 for 2-D P1, 15 for 3-D P1) and picks the kernel unrolled for it.
template <int R, bool NT, int WFIX, bool COH, class TermFn, class GranT>
__device__ __forceinline__ void rowops_body(const RowOp &op, const TermFn &terms,
                                            const Bases &bases, const int s, GranT &gran) {
    const int lane = threadIdx.x & 63;
    if (s >= op.nslices) return;
    constexpr int C = 64 * R;
    int off0, w;
    if constexpr (GranT::hoist) {
        off0 = gran.off0;
        w = gran.w;
    } else if constexpr (WFIX > 0) {
        w = WFIX;
        off0 = s * WFIX;
    } else if (op.uniform_w >= 0) {   // same width everywhere: no slice_off round trip
        w = op.uniform_w;
        off0 = s * w;
    } else {
        off0 = ((gci_p)op.slice_off)[s];
        w = ((gci_p)op.slice_off)[s + 1] - off0;
    }
    const size_t base = (size_t)off0 * C + (size_t)lane * R;
    const int r0 = s * C + lane;   // position of q is r0 + 64 * q
    const int nrows = op.nrows;
    // row-sorted structures (SELL-C-sigma): the row stored at a position comes from perm
    int row[R];
    if constexpr (GranT::hoist) {
#pragma unroll
        for (int q = 0; q < R; ++q) row[q] = gran.row[q];
    } else {
        const gci_p perm = (gci_p)op.perm;
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const int pos = r0 + 64 * q;
            row[q] = perm ? perm[pos] : (pos < nrows ? pos : -1);
        }
    }

    // epilogue operands first: their latency overlaps the matrix stream
    double e0[R], e1[R], e2[R], e3[R];
    bool masked[R];
    const bool lin = op.mode == EPI_LIN;
    gcd_p pa, pb, pc, pd;
    const gcb_p rowmask = (gcb_p)op.rowmask;
    if (lin) {
        pa = resolve(op.yin, bases);
        pb = resolve(op.z, bases);
        pc = resolve(op.mx, bases);
        pd = op.y2.base >= 0 ? (gcd_p)op.dinv : nullptr;
    } else {
        pa = resolve(op.pkm1, bases);
        pb = resolve(op.pk, bases);
        pc = resolve(op.b, bases);
        pd = (gcd_p)op.dinv;
    }
#pragma unroll
    for (int q = 0; q < R; ++q) {
        const int r = row[q];
        const bool in = r >= 0;
        if constexpr (GranT::hoist) {
            masked[q] = gran.masked[q];
            e3[q] = pd ? gran.dinv[q] : 0.0;
        } else {
            masked[q] = in && rowmask != nullptr && rowmask[r] != 0;
            e3[q] = (in && pd) ? pd[r] : 0.0;   // dinv: never written inside a launch
        }
        e0[q] = (in && pa) ? ldv<COH>(pa + r) : 0.0;
        e1[q] = (in && pb) ? ldv<COH>(pb + r) : 0.0;
        e2[q] = (in && pc) ? ldv<COH>(pc + r) : 0.0;
    }

    double acc[R];
#pragma unroll
    for (int q = 0; q < R; ++q) acc[q] = 0.0;
    if (op.nterms > 0) {
        if constexpr (WFIX > 0) {
            accumulate_exact<R, NT, WFIX, COH>(op, terms, bases, base, acc);
        } else if constexpr (WFIX == -1) {
            static_assert(NT && !COH && R == 2 && !GranT::on && !GranT::hoist, "operator apply only");
            // Ragged structures in the operator apply (P2: rows of 9 / 19, Q2: 9 / 15 / 25, the
            // rectangular blocks of the Stokes system: 4 / 7): a wave is one slice, so its width
            // is wave-uniform and the wave can run the body unrolled for exactly that width --
            // the indices of the slice stay in registers for all terms and the next chunk of
            // matrix values is in flight under the gathers, as in the fixed-width launches.  The
            // slot loop below loads (index, value, gather) per slot and term: two dependent round
            // trips per four slots, during which no value load is outstanding -- the value stream
            // and the rest added up instead of overlapping (profiles/r03/spmv_ragged_forms.md).
            // Every row keeps its fma chain (term-major, CSR order): results do not change.
            switch (__builtin_amdgcn_readfirstlane(w)) {
#define KKT_WX(n) case n: accumulate_exact<R, NT, n, COH, TermFn, RAGGED_CH, true>(op, terms, bases, base, acc); break;
                RAGGED_CASES
#undef KKT_WX
                case 24: accumulate_exact<R, NT, 12, COH, TermFn, RAGGED_CH, true, 2>(op, terms, bases, base, acc); break;
                case 38: accumulate_exact<R, NT, 19, COH, TermFn, RAGGED_CH, true, 2>(op, terms, bases, base, acc); break;
                default: accumulate_generic<R, NT, COH>(op, terms, bases, base, w, acc, gran);
            }
        } else if constexpr (WFIX == -2) {
            // the same for launches whose slices are at most 7 wide (the B^T terms of the Stokes
            // system: 4 / 5 / 7, one term): few bytes per wave, so the waves resident count -- the
            // allocation is the 7-wide body's instead of the 19-wide one's
            switch (__builtin_amdgcn_readfirstlane(w)) {
#define KKT_WX(n) case n: accumulate_exact<R, NT, n, COH, TermFn, RAGGED_CH, true>(op, terms, bases, base, acc); break;
                KKT_WX(3) KKT_WX(4) KKT_WX(5) KKT_WX(6) KKT_WX(7)
#undef KKT_WX
                default: accumulate_generic<R, NT, COH>(op, terms, bases, base, w, acc, gran);
            }
        } else {
            accumulate_generic<R, NT, COH>(op, terms, bases, base, w, acc, gran);
        }
    }

    const gd_p y = (gd_p)resolve(op.y, bases);
    double out[R], out2[R];
#pragma unroll
    for (int q = 0; q < R; ++q) {
        out2[q] = 0.0;
        if (lin) {
            // y = ca*acc + cy*yin + cz*z; masked rows: malpha * mx (0 without mx)
            if (masked[q]) {
                out[q] = pc ? op.malpha * e2[q] : 0.0;
            } else {
                double v = op.ca * acc[q];
                if (pa) v += op.cy * e0[q];
                if (pb) v += op.cz * e1[q];
                out[q] = v;
            }
            // optional second output: first Chebyshev step on the row just formed,
            // y2 = c3 * D^-1 y  (p_1 = scale D^-1 b with b = y)
            out2[q] = op.c3 * (e3[q] * out[q]);
        } else {
            // (1-w) p_{k-1} + w p_k + (scale w) D^-1 (b - A p_k): VecAXPBYPCZ order; masked
            // rows: the bc-assembled matrix on a bc-clean right-hand side gives exactly 0
            if (masked[q]) {
                out[q] = 0.0;
            } else {
                double v = pa ? op.c1 * e0[q] : 0.0;
                if (pb) v += op.c2 * e1[q];
                v += op.c3 * (e3[q] * (e2[q] - acc[q]));
                out[q] = op.post2 * (op.post1 * v);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < R; ++q)
        if (row[q] >= 0) stv<COH>(y + row[q], out[q]);
    const bool has_y2 = lin && op.y2.base >= 0;
    if (has_y2) {
        const gd_p y2 = (gd_p)resolve(op.y2, bases);
#pragma unroll
        for (int q = 0; q < R; ++q)
            if (row[q] >= 0) stv<COH>(y2 + row[q], out2[q]);
    }
    if constexpr (GranT::on) {
        // the vector the next phase gathers (p_1 after a fused update, else y), as granules
        typedef KKT_GLOBAL unsigned long long *gu64w_p;
        const gu64w_p yg = (gu64w_p)gran.yg;
#pragma unroll
        for (int q = 0; q < R; ++q)
            if (row[q] >= 0) {
                const unsigned long long bits =
                    (unsigned long long)__double_as_longlong(has_y2 ? out2[q] : out[q]);
                const unsigned long long tag = (unsigned long long)gran.ep_out << 32;
                __hip_atomic_store(yg + 2 * (size_t)row[q], tag | (bits & 0xffffffffull),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(yg + 2 * (size_t)row[q] + 1, tag | (bits >> 32),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
    }
}

template <int R, bool NT, int WFIX, bool COH, class TermFn>
__device__ __forceinline__ void rowops_body(const RowOp &op, const TermFn &terms,
                                            const Bases &bases, const int s) {
    NoGran none;
    rowops_body<R, NT, WFIX, COH>(op, terms, bases, s, none);
}

// Two entry points over one body so that profiles separate the KKT operator apply (the
// roofline kernel of bench.py) from the many small block-row steps of the preconditioner.
// XCD-aware order (xcd != 0; gridDim.x a multiple of 8): workgroup x of a block row runs on XCD
// x % 8, which takes the k-th contiguous eighth of THIS block row's workgroups (eighths differ by
// at most one workgroup).  Returns -1 for a workgroup beyond its eighth.
__device__ __forceinline__ int xcd_workgroup(int nwg, int xcd) {
    if (!xcd) return (int)blockIdx.x;
    const int k = (int)(blockIdx.x & 7), j = (int)(blockIdx.x >> 3);
    const int lo = (k * nwg) >> 3, hi = ((k + 1) * nwg) >> 3;
    return j < hi - lo ? lo + j : -1;
}
// the batched preconditioner steps (pc_rows_shared*, pc_rows_il) in the same order (default on,
// option "pc_xcd" = "0": dispatch order): each XCD gathers from an eighth of every iterate instead
// of all of it -- cfg 2: batched steps 1.045 -> 0.869 ms per application; 64^3: no change
static bool g_pc_xcd = true;
void set_pc_xcd(bool on) { g_pc_xcd = on; }
template <int R, int WFIX>
__global__ __launch_bounds__(256) void kkt_spmv_rows(const RowOp *__restrict__ ops,
                                                     const Bases bases, const int xcd) {
    const RowOp &op = ops[blockIdx.y];
    const int wg = xcd_workgroup((op.nslices + 3) >> 2, xcd);
    if (wg < 0) return;
    rowops_body<R, true, WFIX, false>(op, [&](int t) { return op.t[t]; }, bases,
                                      wg * 4 + (threadIdx.x >> 6));
}
// Ragged structures (WFIX = -1: a wave picks the body unrolled for its slice's width).  The
// allocation is the widest body's; held to two waves per SIMD, where the fixed-width launches
// of 15-wide rows (64^3 P1) run as well.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void kkt_spmv_rows_ragged(const RowOp *__restrict__ ops, const Bases bases, const int per_xcd) {
    const RowOp &op = ops[blockIdx.y];
    // per_xcd > 0 (gridDim.x = 8 * per_xcd): workgroups are dealt round-robin over the 8 XCDs, so
    // workgroup x of a block row runs on XCD x % 8; XCD k takes the k-th contiguous eighth of the
    // block row's workgroups -- its L2 then holds an eighth of the index array and of every x
    // window instead of (nearly) all of both
    // (an eighth of THIS block row's workgroups: the block rows of a launch differ in size)
    const int wpw = (int)(blockDim.x >> 6);
    const int wg = xcd_workgroup((op.nslices + wpw - 1) / wpw, per_xcd);
    if (wg < 0) return;
    rowops_body<2, true, -1, false>(op, [&](int t) { return op.t[t]; }, bases,
                                    wg * wpw + (int)(threadIdx.x >> 6));
}
__global__ __launch_bounds__(256) void kkt_spmv_rows_ragged_narrow(const RowOp *__restrict__ ops,
                                                                   const Bases bases, const int per_xcd) {
    const RowOp &op = ops[blockIdx.y];
    const int wg = xcd_workgroup((op.nslices + 3) >> 2, per_xcd);
    if (wg < 0) return;
    rowops_body<2, true, -2, false>(op, [&](int t) { return op.t[t]; }, bases,
                                    wg * 4 + (int)(threadIdx.x >> 6));
}
template <int R, int WFIX>
__global__ __launch_bounds__(256) void pc_rows(const RowOp *__restrict__ ops,
                                               const Bases bases) {
    const RowOp &op = ops[blockIdx.y];
    rowops_body<R, false, WFIX, false>(op, [&](int t) { return op.t[t]; }, bases,
                                       blockIdx.x * 4 + (threadIdx.x >> 6));
}
// One RowOp passed by value: the descriptor arrives with the kernel arguments instead of
// through a dependent load -- one round trip less on the latency-bound sweep steps.
template <int R, int WFIX>
__global__ __launch_bounds__(256) void pc_row_step(const RowOp op, const Bases bases) {
    rowops_body<R, false, WFIX, false>(op, [&](int t) { return op.t[t]; }, bases,
                                       blockIdx.x * 4 + (threadIdx.x >> 6));
}


// ---------------------------------------------------------------- persistent row program
//
// The time sweeps of the block-Schur preconditioner are chains of dependent single-block
// steps (2 n_t solves x schur_its Chebyshev steps); as separate launches each costs ~5 us
// of kernel boundary + cold dependent loads.  Here ONE launch walks the whole chain:
// workgroup j owns the same slices in every phase, and before phase ph it waits only for
// the workgroups whose rows it gathers from (and that gather from it) to have finished
// phase ph-1 -- a neighbour barrier through per-workgroup phase counters, not a grid
// barrier.  Hand-off form: sc1 payload stores, every wave drains (vmcnt(0)), workgroup
// barrier, one lane publishes the counter; consumer: one wave polls relaxed, workgroup
// barrier, then sc1 loads only (Guideline 16, R1 with sc1 loads in place of the acquire).
// Correctness does not depend on placement; residency does: the grid is at most one
// workgroup per CU.  Every spin is bounded: on a time-out the error word is set and the
// kernel runs to its end (results invalid, reported by the host) instead of hanging.
constexpr int FLAG_STRIDE = 32;                 // one 128-byte line per workgroup counter
constexpr unsigned PROG_SPIN_LIMIT = 1u << 20;

// A phase descriptor (one RowOp, < 512 bytes) is fetched by ONE wave-wide vector load --
// lane l holds bytes 8l..8l+7 -- a whole phase ahead, and its fields are moved to scalar
// registers with v_readlane.  Reading the descriptor field by field through the scalar
// cache cost a dependent ~1 us miss per touched line and phase (scripts/prog_stamps.py).
static_assert(sizeof(RowOp) <= 512 && sizeof(RowOp) % 8 == 0,
              "RowOp must fit one 64 x 8-byte wave load");
__device__ __forceinline__ unsigned long long rl64(unsigned long long v, int lane) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), lane);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long load_desc_words(const RowOp *op) {
    const KKT_GLOBAL unsigned long long *p = (const KKT_GLOBAL unsigned long long *)op;
    const int lane = threadIdx.x & 63;
    return (size_t)lane * 8 < sizeof(RowOp) ? p[lane] : 0ull;
}
// every field of a wave-held descriptor except the term array, as wave-uniform values
#define KKT_U64(f) rl64(desc, (int)(offsetof(RowOp, f) / 8))
#define KKT_U32(f) ((int32_t)(unsigned)(KKT_U64(f) >> (8 * (offsetof(RowOp, f) % 8))))
#define KKT_UF64(f) __longlong_as_double((long long)KKT_U64(f))
#define KKT_UPTR(T, f) ((T)(uintptr_t)KKT_U64(f))
#define KKT_UVREF(f) VRef{(int64_t)KKT_U64(f.off), KKT_U32(f.base), 0}
__device__ __forceinline__ RowOp unpack_desc_head(unsigned long long desc) {
    RowOp op;
    op.col = KKT_UPTR(const int32_t *, col);
    op.slice_off = KKT_UPTR(const int32_t *, slice_off);
    op.perm = KKT_UPTR(const int32_t *, perm);
    op.nrows = KKT_U32(nrows);
    op.nslices = KKT_U32(nslices);
    op.nterms = KKT_U32(nterms);
    op.mode = KKT_U32(mode);
    op.uniform_w = KKT_U32(uniform_w);
    op.y = KKT_UVREF(y);
    op.y2 = KKT_UVREF(y2);
    op.ca = KKT_UF64(ca);
    op.cy = KKT_UF64(cy);
    op.cz = KKT_UF64(cz);
    op.yin = KKT_UVREF(yin);
    op.z = KKT_UVREF(z);
    op.rowmask = KKT_UPTR(const uint8_t *, rowmask);
    op.mx = KKT_UVREF(mx);
    op.malpha = KKT_UF64(malpha);
    op.b = KKT_UVREF(b);
    op.pkm1 = KKT_UVREF(pkm1);
    op.pk = KKT_UVREF(pk);
    op.dinv = KKT_UPTR(const double *, dinv);
    op.c1 = KKT_UF64(c1);
    op.c2 = KKT_UF64(c2);
    op.c3 = KKT_UF64(c3);
    op.post1 = KKT_UF64(post1);
    op.post2 = KKT_UF64(post2);
    return op;
}
__device__ __forceinline__ SpmvTerm unpack_desc_term(unsigned long long desc, int t) {
    const int w0 = (int)(offsetof(RowOp, t) / 8) + t * (int)(sizeof(SpmvTerm) / 8);
    SpmvTerm tm;
    tm.vals = (const double *)(uintptr_t)rl64(desc, w0 + (int)(offsetof(SpmvTerm, vals) / 8));
    tm.x.off = (int64_t)rl64(desc, w0 + (int)(offsetof(SpmvTerm, x) / 8));
    tm.x.base = (int32_t)(unsigned)rl64(desc, w0 + (int)(offsetof(SpmvTerm, x) / 8) + 1);
    tm.x.pad_ = 0;
    return tm;
}

// Everything of phase `op` that can be read before the neighbours have finished the
// previous phase (see Hoist).  Matrix structure, values, row mask and diagonal are never
// written inside a launch.
template <int R>
__device__ __forceinline__ void hoist_prepare(Hoist<R> &h, const RowOp &op, const SpmvTerm &t0,
                                              int s) {
    constexpr int C = 64 * R;
    const int lane = threadIdx.x & 63;
    if (s >= op.nslices) return;
    if (h.key_slice != (const void *)op.slice_off || h.key_perm != (const void *)op.perm ||
        h.key_uw != op.uniform_w || h.key_nrows != op.nrows) {
        h.key_slice = (const void *)op.slice_off;
        h.key_perm = (const void *)op.perm;
        h.key_uw = op.uniform_w;
        h.key_nrows = op.nrows;
        h.key_mask = h.key_dinv = (const void *)(uintptr_t)1;   // rows changed: re-read below
        if (op.uniform_w >= 0) {
            h.w = op.uniform_w;
            h.off0 = s * h.w;
        } else {
            h.off0 = ((gci_p)op.slice_off)[s];
            h.w = ((gci_p)op.slice_off)[s + 1] - h.off0;
        }
        const gci_p perm = (gci_p)op.perm;
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const int pos = s * C + lane + 64 * q;
            h.row[q] = perm ? perm[pos] : (pos < op.nrows ? pos : -1);
        }
    }
    if (h.key_mask != (const void *)op.rowmask) {
        h.key_mask = (const void *)op.rowmask;
        const gcb_p rowmask = (gcb_p)op.rowmask;
#pragma unroll
        for (int q = 0; q < R; ++q)
            h.masked[q] = h.row[q] >= 0 && rowmask != nullptr && rowmask[h.row[q]] != 0;
    }
    if (h.key_dinv != (const void *)op.dinv) {
        h.key_dinv = (const void *)op.dinv;
        const gcd_p dinv = (gcd_p)op.dinv;
#pragma unroll
        for (int q = 0; q < R; ++q)
            h.dinv[q] = (h.row[q] >= 0 && dinv != nullptr) ? dinv[h.row[q]] : 0.0;
    }
    if (op.nterms > 0 && h.w > 0) {
        const size_t base = (size_t)h.off0 * C + (size_t)lane * R;
        const gci_p colp = (gci_p)op.col + base;
        const gcd_p vp = (gcd_p)t0.vals + base;
        const int w = h.w;
#pragma unroll
        for (int k = 0; k < HOIST_KC; ++k) {
            const size_t kk = (size_t)(k < w ? k : w - 1) * C;
            load_cols<R>(colp + kk, h.c[k]);
            load_vals<R, false>(vp + kk, h.v[k]);
        }
    }
}

template <int R, int WFIX>
__device__ __forceinline__ void row_program_body(const RowOp *__restrict__ ops, int nphases,
                                                 const int2 *__restrict__ dep, unsigned *flags,
                                                 unsigned *err) {
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int wpw = blockDim.x >> 6;
    const int j = blockIdx.x;
    const int s = j * wpw + wave;
    const int2 d = dep[j];
    const Bases B{{nullptr, nullptr, nullptr, nullptr}};
    bool dead = false;   // a spin timed out: stop waiting, run to the end, results invalid
    unsigned long long dnext = load_desc_words(ops);
#ifdef KKT_STAMPS
    // diagnostic build: per-workgroup cycle sums of the stages of a phase (scripts/prog_stamps.py)
    unsigned long long *sdbg = reinterpret_cast<unsigned long long *>(err + 64) + blockIdx.x * 16;
    unsigned long long t_prev = __builtin_amdgcn_s_memtime();
#define KKT_FSTAGE(i)                                                  \
    do {                                                               \
        const unsigned long long t_now = __builtin_amdgcn_s_memtime(); \
        if (threadIdx.x == 0) sdbg[8 + (i)] += t_now - t_prev;         \
        t_prev = t_now;                                                \
    } while (0)
#else
#define KKT_FSTAGE(i)
#endif
    Hoist<R> hoist;
    hoist.key_slice = hoist.key_perm = hoist.key_mask = hoist.key_dinv =
        (const void *)(uintptr_t)1;   // matches no structure
    hoist.key_uw = hoist.key_nrows = -2;
    hoist.off0 = hoist.w = 0;
    for (int ph = 0; ph < nphases; ++ph) {
        const unsigned long long desc = dnext;
        if (ph + 1 < nphases) dnext = load_desc_words(ops + ph + 1);   // lands during this phase
        const RowOp op = unpack_desc_head(desc);
        if constexpr (WFIX == 0) hoist_prepare<R>(hoist, op, unpack_desc_term(desc, 0), s);
        KKT_FSTAGE(0);   // loop top: descriptor, cached structure, first chunk issued
        if (ph > 0) {
            if (wave == 0) {
                const int jj = d.x + lane;
                unsigned spins = 0;
                bool ok;
                do {
                    ok = jj > d.y || __hip_atomic_load(flags + (size_t)jj * FLAG_STRIDE,
                                                       __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)ph;
                    if (__all(ok) || dead) break;
                    __builtin_amdgcn_s_sleep(1);
                } while (++spins < PROG_SPIN_LIMIT);
                if (!__all(ok)) {
                    dead = true;
                    if (lane == 0) atomicOr(err, 1u);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __syncthreads();
        }
        KKT_FSTAGE(1);   // wait for the neighbours' counters + workgroup barrier
        if constexpr (WFIX == 0)
            rowops_body<R, false, WFIX, true>(
                op, [&](int t) { return unpack_desc_term(desc, t); }, B, s, hoist);
        else
            rowops_body<R, false, WFIX, true>(
                op, [&](int t) { return unpack_desc_term(desc, t); }, B, s);
        KKT_FSTAGE(2);   // loads, fma chain, store issue
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        KKT_FSTAGE(3);   // store drain
        __syncthreads();
        if (threadIdx.x == 0)
            __hip_atomic_store(flags + (size_t)j * FLAG_STRIDE, (unsigned)(ph + 1), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        KKT_FSTAGE(4);   // workgroup barrier + counter store
#ifdef KKT_STAMPS
        if (threadIdx.x == 0) sdbg[2] += 1;
#endif
    }
}
#undef KKT_FSTAGE

template <int R, int WFIX>
__global__ __launch_bounds__(512) void pc_row_program(const RowOp *__restrict__ ops, int nphases,
                                                       const int2 *__restrict__ dep,
                                                       unsigned *flags, unsigned *err) {
    row_program_body<R, WFIX>(ops, nphases, dep, flags, err);
}
// The same program held to 168 registers (three waves per SIMD, 3 072 wave slots on 256 CUs):
// for blocks with more than 2 048 slices of 128 rows -- 512^2 P1, the 64^3 blocks of BASELINE
// configs[3] -- every wave of the persistent launch must be resident, and two waves per SIMD at
// full register use are 2 048.
template <int R, int WFIX>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(3, 3)))
void pc_row_program_lo(const RowOp *__restrict__ ops, int nphases, const int2 *__restrict__ dep,
                       unsigned *flags, unsigned *err) {
    row_program_body<R, WFIX>(ops, nphases, dep, flags, err);
}


// Data-flow form for structures without a small fixed width (P2 / Q2 / 3-D P1): the phases
// of pc_row_program with the hand-off of pc_row_program_g -- gathers poll tagged granules in
// chunks, outputs are published as granules, and there is no counter, no drain and no
// workgroup barrier.  Matrix values and indices are re-read from L2 every phase (too wide to
// keep in registers).  Needs the same symmetric gather relation between waves and the same
// "phase ph gathers what phase ph - 1 produced" chain as pc_row_program_g.
template <int R>
__global__ __launch_bounds__(512) void pc_row_program_gw(const RowOp *__restrict__ ops, int nphases,
                                                          unsigned long long *g0,
                                                          unsigned long long *g1, unsigned *err) {
    const int wave = threadIdx.x >> 6;
    const int s = blockIdx.x * (blockDim.x >> 6) + wave;
    const Bases B{{nullptr, nullptr, nullptr, nullptr}};
    Gran gran;
    gran.err = err;
    gran.dead = false;
    unsigned long long dnext = load_desc_words(ops);
    for (int ph = 0; ph < nphases; ++ph) {
        const unsigned long long desc = dnext;
        if (ph + 1 < nphases) dnext = load_desc_words(ops + ph + 1);
        const RowOp op = unpack_desc_head(desc);
        gran.xg = (ph & 1) ? g0 : g1;
        gran.yg = (ph & 1) ? g1 : g0;
        gran.ep_in = (ph > 0 && op.nterms > 0) ? (unsigned)ph : 0u;
        gran.ep_out = (unsigned)(ph + 1);
        rowops_body<R, false, 0, true>(
            op, [&](int t) { return unpack_desc_term(desc, t); }, B, s, gran);
    }
}

// ---- persistent row program, data-flow form ("the data is the flag", Guideline 16 R2)
//
// Every vector element that crosses workgroups inside the program travels as two 8-byte
// granules {tag = phase epoch, 32 value bits}, each written by ONE aligned 8-byte
// agent-scope store; a gather re-reads its granules until every tag carries the epoch of
// the producing phase.  No flag, no drain (s_waitcnt vmcnt(0)), no poll wave: a hand-off
// costs one store -> visible -> load latency.  Two granule buffers alternate between
// phases.  Write-after-read: a workgroup stores its phase-ph granules only after its own
// barrier of phase ph, which follows its gathers; whoever observes one of those granules
// therefore knows that workgroup has finished reading generation ph-1, and overwrites the
// buffer holding it (the alternate one) only after observing them -- which needs the
// gather relation between workgroups to be symmetric (checked on the host).
typedef KKT_GLOBAL unsigned long long *gu64_p;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define KKT_D64(field) rl64(desc, (int)(offsetof(RowOp, field) / 8))
#define KKT_DPTR(type, field) ((type)(uintptr_t)KKT_D64(field))
#define KKT_DF64(field) __longlong_as_double((long long)KKT_D64(field))
#define KKT_D32(field)                                                                   \
    ((int)(unsigned)(rl64(desc, (int)(offsetof(RowOp, field) / 8)) >>                    \
                     (8 * (offsetof(RowOp, field) % 8))))
__device__ __forceinline__ gcd_p vref_ptr(long long off, int base) {
    return base == 0 ? (gcd_p)(const double *)(uintptr_t)off : (gcd_p) nullptr;
}

// Register-resident state across phases.  A sweep is a chain of solves; inside one solve the
// matrix, its Jacobi diagonal, the right-hand side row and the boundary mask do not change
// and the Chebyshev operands p_k[r], p_{k-1}[r] are values this very thread produced one and
// two phases earlier.  Re-loading any of them costs a ~1.2 us dependent round trip per phase
// (measured, scripts/prog_stamps.py), so the kernel keeps a small pointer-keyed cache in
// registers: a same-row operand whose address equals a cached one is taken from the cache
// (bitwise the value in memory), and matrix values / column indices are re-loaded only when
// their pointer changes.  What remains on the critical path of a Chebyshev phase is the
// descriptor fetch and ONE gather round trip.
template <int R, int W>
__global__ __launch_bounds__(512) void pc_row_program_g(const RowOp *__restrict__ ops,
                                                         const PhaseLite *__restrict__ lite,
                                                         int nphases, unsigned long long *g0,
                                                         unsigned long long *g1, unsigned gbytes,
                                                         unsigned *err) {
    constexpr int C = 64 * R;
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int s = blockIdx.x * (blockDim.x >> 6) + wave;
    const size_t base = (size_t)s * W * C + (size_t)lane * R;
    const int r0 = s * C + lane;

    // cache slots: 0 = y of ph-1, 1 = y2 of ph-1, 2 = gather output of ph-2, 3 = b, 4 = dinv
    const void *kp0 = nullptr, *kp1 = nullptr, *kp2 = nullptr, *kp3 = nullptr, *kp4 = nullptr;
    double kv0[R], kv1[R], kv2[R], kv3[R], kv4[R];
#pragma unroll
    for (int q = 0; q < R; ++q) kv0[q] = kv1[q] = kv2[q] = kv3[q] = kv4[q] = 0.0;
    const void *cols_ptr = nullptr, *vals_ptr = nullptr, *mask_ptr = nullptr;
    int c[W][R];
    double v[W][R];
    bool masked[R];
    // what this wave published in the previous phase, by row within the slice: columns that
    // are rows of the own slice are gathered from here instead of from memory
    __shared__ double own_lds[8][C];
#pragma unroll
    for (int q = 0; q < R; ++q) {
        masked[q] = false;
        own_lds[wave][lane + 64 * q] = 0.0;
    }

#define KKT_FETCH(dst, ptr)                                                      \
    do {                                                                         \
        const void *key_ = (const void *)(const double *)(ptr);                  \
        _Pragma("unroll") for (int q = 0; q < R; ++q) {                          \
            const int r_ = r0 + 64 * q;                                          \
            const bool in_ = active && r_ < nrows;                               \
            double val_ = 0.0;                                                   \
            if ((ptr) && in_) {                                                  \
                if (key_ == kp0) val_ = kv0[q];                                  \
                else if (key_ == kp1) val_ = kv1[q];                             \
                else if (key_ == kp2) val_ = kv2[q];                             \
                else if (key_ == kp3) val_ = kv3[q];                             \
                else if (key_ == kp4) val_ = kv4[q];                             \
                else val_ = (ptr)[r_];                                           \
            }                                                                    \
            dst[q] = val_;                                                       \
        }                                                                        \
    } while (0)

#ifdef KKT_STAMPS
    unsigned long long *sdbg = reinterpret_cast<unsigned long long *>(err + 64) + blockIdx.x * 16;
    unsigned long long t_prev = __builtin_amdgcn_s_memtime();
#define KKT_STAGE(i)                                                       \
    do {                                                                   \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");        \
        const unsigned long long t_now = __builtin_amdgcn_s_memtime();     \
        if (threadIdx.x == 0) sdbg[8 + (i)] += t_now - t_prev;             \
        t_prev = t_now;                                                    \
    } while (0)
#else
#define KKT_STAGE(i)
#endif
    auto load_desc = [&](int ph) -> unsigned long long {
        const KKT_GLOBAL unsigned long long *p =
            (const KKT_GLOBAL unsigned long long *)(ops + ph);
        return (size_t)lane * 8 < sizeof(RowOp) ? p[lane] : 0ull;
    };
    unsigned long long dnext = load_desc(0);
    bool dead = false;   // a spin timed out: stop waiting, run to the end, results invalid
    // state that a STEP phase (PhaseLite, kind 1) inherits from the last fully decoded phase
    int op_nslices = 0, nrows = 0;
    PhaseLite lnext = lite[0];
    for (int ph = 0; ph < nphases; ++ph) {
        const unsigned long long desc = dnext;
        const PhaseLite L = lnext;
        if (ph + 1 < nphases) {
            dnext = load_desc(ph + 1);   // in flight during this phase
            lnext = lite[ph + 1];        // 64 bytes through the scalar cache, one phase ahead
        }
        // A STEP phase is a Chebyshev step of the solve the previous phase belongs to: same
        // matrix, diagonal, right-hand side and mask; p_k is what the previous phase produced
        // and p_{k-1} what the phase before it produced.  Everything it needs beyond the 64-byte
        // record is already in registers -- no descriptor decoding, no pointer compares.
        const bool step = L.kind == 1u;
        int op_nterms;
        bool lin;
        gcd_p p_y, p_y2, pa, pb, pc, pd;
        const void *d_col = cols_ptr, *d_vals0 = vals_ptr;
        double k_ca = 0.0, k_cy = 0.0, k_cz = 0.0, k_malpha = 0.0;
        double k_c1, k_c2, k_c3, k_post1, k_post2;
        double e0[R], e1[R], e2[R], e3[R];
        if (step) {
            op_nterms = 1;
            lin = false;
            p_y = (gcd_p)(const double *)(uintptr_t)L.y;
            p_y2 = nullptr;
            pa = (L.flags & 1u) ? (gcd_p)(const double *)kp2 : nullptr;
            pb = (gcd_p)(const double *)kp0;
            pc = (gcd_p)(const double *)kp3;
            pd = (gcd_p)(const double *)kp4;
            k_c1 = L.c1;
            k_c2 = L.c2;
            k_c3 = L.c3;
            k_post1 = L.post1;
            k_post2 = L.post2;
#pragma unroll
            for (int q = 0; q < R; ++q) {
                e0[q] = kv2[q];
                e1[q] = kv0[q];
                e2[q] = kv3[q];
                e3[q] = kv4[q];
            }
        } else {
            op_nslices = KKT_D32(nslices);
            nrows = KKT_D32(nrows);
            op_nterms = KKT_D32(nterms);
            lin = KKT_D32(mode) == EPI_LIN;
        }
        const bool active = s < op_nslices;
        const int nterms = active ? op_nterms : 0;
        KKT_STAGE(0);   // descriptor
        // this phase's granule buffer as a raw buffer resource: one 16-byte sc1 store per row
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
            (void *)((ph & 1) ? g1 : g0), 0, (int)gbytes, 0x00020000);
        const unsigned ep_in = (ph > 0 && op_nterms > 0) ? (unsigned)ph : 0u;
        const unsigned ep_out = (unsigned)(ph + 1);
        if (!step) {
            // program descriptors carry absolute pointers only (base 0) or null (base < 0)
            p_y = vref_ptr((long long)KKT_D64(y.off), KKT_D32(y.base));
            p_y2 = vref_ptr((long long)KKT_D64(y2.off), KKT_D32(y2.base));
            const void *d_rowmask = KKT_DPTR(const void *, rowmask);
            d_col = KKT_DPTR(const void *, col);
            d_vals0 = KKT_DPTR(const void *, t[0].vals);
            const gcd_p d_dinv = (gcd_p)KKT_DPTR(const double *, dinv);

            // ---- operands of this thread's own rows (cache first)
            if (lin) {
                pa = vref_ptr((long long)KKT_D64(yin.off), KKT_D32(yin.base));
                pb = vref_ptr((long long)KKT_D64(z.off), KKT_D32(z.base));
                pc = vref_ptr((long long)KKT_D64(mx.off), KKT_D32(mx.base));
                pd = p_y2 ? d_dinv : nullptr;
            } else {
                pa = vref_ptr((long long)KKT_D64(pkm1.off), KKT_D32(pkm1.base));
                pb = vref_ptr((long long)KKT_D64(pk.off), KKT_D32(pk.base));
                pc = vref_ptr((long long)KKT_D64(b.off), KKT_D32(b.base));
                pd = d_dinv;
            }
            if (d_rowmask != mask_ptr) {
                mask_ptr = d_rowmask;
                const gcb_p rowmask = (gcb_p)(const uint8_t *)d_rowmask;
#pragma unroll
                for (int q = 0; q < R; ++q) {
                    const int r = r0 + 64 * q;
                    masked[q] = active && r < nrows && rowmask != nullptr && rowmask[r] != 0;
                }
            }
            KKT_FETCH(e0, pa);
            KKT_FETCH(e1, pb);
            KKT_FETCH(e2, pc);
            KKT_FETCH(e3, pd);
            // b (Chebyshev) and dinv stay the same through a solve: pin them
            if (!lin) {
                kp3 = (const void *)(const double *)pc;
#pragma unroll
                for (int q = 0; q < R; ++q) kv3[q] = e2[q];
            }
            if (pd) {
                kp4 = (const void *)(const double *)pd;
#pragma unroll
                for (int q = 0; q < R; ++q) kv4[q] = e3[q];
            }
        }

        KKT_STAGE(1);   // own-row operands (cache or memory)
        // ---- matrix structure and values: re-load only when the pointer changes
        double xv[W][R];
        if (nterms > 0) {
            if (d_col != cols_ptr) {
                cols_ptr = d_col;
                const gci_p colp = (gci_p)(const int32_t *)d_col + base;
#pragma unroll
                for (int k = 0; k < W; ++k) load_cols<R>(colp + (size_t)k * C, c[k]);
            }
            if (d_vals0 != vals_ptr) {
                vals_ptr = d_vals0;
                const gcd_p vp = (gcd_p)(const double *)d_vals0 + base;
#pragma unroll
                for (int k = 0; k < W; ++k) load_vals<R, false>(vp + (size_t)k * C, v[k]);
            }
            KKT_STAGE(2);   // index / matrix value loads (only when they change)
            if (ep_in == 0) {   // produced before this launch: plain vector
                const gcd_p x = vref_ptr((long long)KKT_D64(t[0].x.off), KKT_D32(t[0].x.base));
#pragma unroll
                for (int k = 0; k < W; ++k)
#pragma unroll
                    for (int q = 0; q < R; ++q) xv[k][q] = x[c[k][q]];
            } else {
                // Columns that are rows of this very slice were produced by this wave one phase
                // ago: they come out of its LDS copy, bitwise the published values.  Only the
                // other columns are polled in memory (two 8-byte agent-scope loads of
                // {tag, lo} and {tag, hi}: 16-byte sc1 buffer LOADS were measured to return
                // stale lines, scripts/microbench/xcd_handoff.hip); lanes that need nothing
                // from memory all aim at one granule of their own slice, which already
                // carries this phase's tag, so the poll stays branch-free and nearly free.
                const gu64_p xg = (gu64_p)((ph & 1) ? g0 : g1);   // written by phase ph-1
                unsigned long long ga[W][R], gb[W][R];
                unsigned spins = 0;
                bool ok;
                while (true) {
#pragma unroll
                    for (int k = 0; k < W; ++k)
#pragma unroll
                        for (int q = 0; q < R; ++q) {
                            const bool own = (unsigned)(c[k][q] - s * C) < (unsigned)C;
                            const gu64_p g = xg + 2 * (size_t)(own ? s * C : c[k][q]);
                            ga[k][q] = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            gb[k][q] = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    ok = true;
#pragma unroll
                    for (int k = 0; k < W; ++k)
#pragma unroll
                        for (int q = 0; q < R; ++q)
                            ok &= (unsigned)(ga[k][q] >> 32) == ep_in &&
                                  (unsigned)(gb[k][q] >> 32) == ep_in;
                    if (__all(ok) || dead || ++spins >= PROG_SPIN_LIMIT) break;
                    // back off: a wave that spins at full rate takes the issue slots (older wave
                    // first, MI355X_MICROARCH.md "Two waves per SIMD") and the memory queue of
                    // the very CU whose other waves may have to produce what it waits for
                    __builtin_amdgcn_s_sleep(2);
                }
                if (!__all(ok)) {
                    // who waited for what: a tag below the expected one means the producer has
                    // not run (or is not resident), above it an overwritten granule
                    if (!dead && !ok && atomicCAS(err + 1, 0u, 1u) == 0u) {
                        bool mine = false;
#pragma unroll
                        for (int k = 0; k < W; ++k)
#pragma unroll
                            for (int q = 0; q < R; ++q) {
                                const bool bad = (unsigned)(ga[k][q] >> 32) != ep_in ||
                                                 (unsigned)(gb[k][q] >> 32) != ep_in;
                                if (bad && !mine) {
                                    mine = true;
                                    err[24] = (unsigned)blockIdx.x;
                                    err[25] = (unsigned)wave;
                                    err[26] = (unsigned)lane;
                                    err[27] = (unsigned)ph;
                                    err[28] = ep_in;
                                    err[29] = (unsigned)(ga[k][q] >> 32);
                                    err[30] = (unsigned)(gb[k][q] >> 32);
                                    err[31] = (unsigned)c[k][q];
                                    err[32] = (unsigned)(c[k][q] / C);   // producing slice
                                }
                            }
                    }
                    dead = true;
                    if (lane == 0) atomicOr(err, 2u);
                }
#pragma unroll
                for (int k = 0; k < W; ++k)
#pragma unroll
                    for (int q = 0; q < R; ++q) {
                        const int rel = c[k][q] - s * C;
                        const bool own = (unsigned)rel < (unsigned)C;
                        const double vo = own_lds[wave][rel & (C - 1)];
                        const double vm = __longlong_as_double(
                            (long long)((ga[k][q] & 0xffffffffull) | (gb[k][q] << 32)));
                        xv[k][q] = own ? vo : vm;
                    }
            }
        }
        KKT_STAGE(3);   // gather
        __syncthreads();   // this workgroup has finished reading generation ep_in
        KKT_STAGE(4);   // barrier

        double acc[R];
#pragma unroll
        for (int q = 0; q < R; ++q) acc[q] = 0.0;
        for (int t = 0; t < nterms; ++t) {
            if (t > 0) {   // further terms of an update phase: values are not cached
                vals_ptr = nullptr;
                const gcd_p vp = (gcd_p)(const double *)(uintptr_t)rl64(
                                     desc, (int)((offsetof(RowOp, t) + sizeof(SpmvTerm) * t +
                                                  offsetof(SpmvTerm, vals)) / 8)) + base;
#pragma unroll
                for (int k = 0; k < W; ++k) load_vals<R, false>(vp + (size_t)k * C, v[k]);
            }
#pragma unroll
            for (int k = 0; k < W; ++k)
#pragma unroll
                for (int q = 0; q < R; ++q) acc[q] = __builtin_fma(v[k][q], xv[k][q], acc[q]);
        }

        const gd_p y = (gd_p)p_y;
        const gd_p y2 = lin ? (gd_p)p_y2 : nullptr;
        if (!step) {   // decoded after the gather: these moves overlap the poll's round trip
            k_ca = KKT_DF64(ca);
            k_cy = KKT_DF64(cy);
            k_cz = KKT_DF64(cz);
            k_malpha = KKT_DF64(malpha);
            k_c1 = KKT_DF64(c1);
            k_c2 = KKT_DF64(c2);
            k_c3 = KKT_DF64(c3);
            k_post1 = KKT_DF64(post1);
            k_post2 = KKT_DF64(post2);
        }
        double outv[R], out2v[R];
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const int r = r0 + 64 * q;
            double out = 0.0, out2 = 0.0;
            if (lin) {
                if (masked[q]) {
                    out = pc ? k_malpha * e2[q] : 0.0;
                } else {
                    out = k_ca * acc[q];
                    if (pa) out += k_cy * e0[q];
                    if (pb) out += k_cz * e1[q];
                }
                out2 = k_c3 * (e3[q] * out);
            } else if (!masked[q]) {
                double t = pa ? k_c1 * e0[q] : 0.0;
                if (pb) t += k_c2 * e1[q];
                t += k_c3 * (e3[q] * (e2[q] - acc[q]));
                out = k_post2 * (k_post1 * t);
            }
            outv[q] = out;
            out2v[q] = out2;
            if (active && r < nrows) {
                y[r] = out;
                if (y2) y2[r] = out2;
                // the vector the next phase gathers: p_1 after a fused update, else y
                const unsigned long long bits =
                    (unsigned long long)__double_as_longlong(y2 ? out2 : out);
                const u32x4 g = {(unsigned)bits, ep_out, (unsigned)(bits >> 32), ep_out};
                __builtin_amdgcn_raw_buffer_store_b128(g, ry, r * 16, 0, 16 /* sc1 */);
            }
            own_lds[wave][lane + 64 * q] = y2 ? out2 : out;
        }
        // rotate the cache: what was produced one phase ago becomes "two phases ago"
#ifdef KKT_STAMPS
        {
            const unsigned long long t_now = __builtin_amdgcn_s_memtime();
            if (threadIdx.x == 0) {
                sdbg[8 + 5] += t_now - t_prev;   // fma + epilogue + store issue (no drain)
                sdbg[2] += 1;
            }
            t_prev = t_now;
        }
        KKT_STAGE(6);   // store drain (diagnostic only)
#endif
        {
            const bool had2 = kp1 != nullptr;
            kp2 = had2 ? kp1 : kp0;
#pragma unroll
            for (int q = 0; q < R; ++q) kv2[q] = had2 ? kv1[q] : kv0[q];
        }
        kp0 = (const void *)(const double *)(gcd_p)y;
        kp1 = y2 ? (const void *)(const double *)(gcd_p)y2 : nullptr;
#pragma unroll
        for (int q = 0; q < R; ++q) {
            kv0[q] = outv[q];
            kv1[q] = out2v[q];
        }
        // a fused update wrote the right-hand side of the solve that follows: pin it as b
        if (lin && y2) {
            kp3 = kp0;
#pragma unroll
            for (int q = 0; q < R; ++q) kv3[q] = outv[q];
        }
    }
#undef KKT_FETCH
#undef KKT_STAGE
}

typedef void (*progg_fn)(const RowOp *, const PhaseLite *, int, unsigned long long *,
                         unsigned long long *, unsigned, unsigned *);
static progg_fn pick_program_g(int uniform_w) {
    switch (uniform_w) {
#define KKT_W(n) case n: return pc_row_program_g<2, n>;
        KKT_W(1) KKT_W(2) KKT_W(3) KKT_W(4) KKT_W(5) KKT_W(6) KKT_W(7)
#undef KKT_W
        default: return nullptr;   // wider rows: register pressure, use the counter form
    }
}
bool row_program_g_available(int R, int uniform_w) { return R == 2 && pick_program_g(uniform_w); }
int row_program_g_max_wgs(int uniform_w, int waves_per_wg) {
    int dev = 0, cus = 0, per_cu = 0;
    progg_fn f = pick_program_g(uniform_w);
    if (!f || hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, f, 64 * waves_per_wg, 0) != hipSuccess)
        return 0;
    if (per_cu > 4) per_cu = 4;
    return cus * per_cu;
}
void launch_row_program_g(hipStream_t s, const RowOp *d_ops, const PhaseLite *d_lite, int nphases,
                          int nwg, int waves_per_wg, int uniform_w, unsigned long long *g0,
                          unsigned long long *g1, size_t granule_words, unsigned *d_err) {
    if (nphases <= 0 || nwg <= 0) return;
    // tags of an earlier launch must not match this launch's epochs
    launch_zero_bytes(s, g0, granule_words * sizeof(unsigned long long));
    launch_zero_bytes(s, g1, granule_words * sizeof(unsigned long long));
    hipLaunchKernelGGL(pick_program_g(uniform_w), dim3(nwg), dim3(64 * waves_per_wg), 0, s, d_ops,
                       d_lite, nphases, g0, g1,
                       (unsigned)(granule_words * sizeof(unsigned long long)), d_err);
}

int row_program_gw_max_wgs(int waves_per_wg) {
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pc_row_program_gw<2>,
                                                     64 * waves_per_wg, 0) != hipSuccess)
        return 0;
    if (per_cu > 4) per_cu = 4;
    return cus * per_cu;
}
void launch_row_program_gw(hipStream_t s, const RowOp *d_ops, int nphases, int nwg,
                           int waves_per_wg, unsigned long long *g0, unsigned long long *g1,
                           size_t granule_words, unsigned *d_err) {
    if (nphases <= 0 || nwg <= 0) return;
    launch_zero_bytes(s, g0, granule_words * sizeof(unsigned long long));
    launch_zero_bytes(s, g1, granule_words * sizeof(unsigned long long));
    hipLaunchKernelGGL(pc_row_program_gw<2>, dim3(nwg), dim3(64 * waves_per_wg), 0, s, d_ops,
                       nphases, g0, g1, d_err);
}

int prog_flag_words(int nwg) { return nwg * FLAG_STRIDE; }

typedef void (*prog_fn)(const RowOp *, int, const int2 *, unsigned *, unsigned *);
static prog_fn pick_program(int R, int uniform_w, bool lowreg = false) {
    if (R != 2) return pc_row_program<1, 0>;
    if (lowreg) {
        switch (uniform_w) {
            case 5: return pc_row_program_lo<2, 5>;
            case 7: return pc_row_program_lo<2, 7>;
            default: return nullptr;
        }
    }
    switch (uniform_w) {
#define KKT_W(n) case n: return pc_row_program<2, n>;
        KKT_W(1) KKT_W(2) KKT_W(3) KKT_W(4) KKT_W(5) KKT_W(6) KKT_W(7) KKT_W(8)
        KKT_W(9) KKT_W(10) KKT_W(11) KKT_W(12) KKT_W(13) KKT_W(14) KKT_W(15) KKT_W(16)
#undef KKT_W
        default: return pc_row_program<2, 0>;
    }
}

// Workgroups of 64*waves_per_wg threads that are certainly co-resident on this device
// (occupancy query of the chosen instantiation, at most 4 per CU: MI355X_MICROARCH.md,
// residency and cooperative launch).
int row_program_max_wgs(int R, int uniform_w, int waves_per_wg, bool lowreg) {
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (!pick_program(R, uniform_w, lowreg)) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pick_program(R, uniform_w, lowreg),
                                                     64 * waves_per_wg, 0) != hipSuccess)
        return 0;
    if (per_cu > 4) per_cu = 4;
    return cus * per_cu;
}

void launch_row_program(hipStream_t s, const RowOp *d_ops, int nphases, int nwg, int waves_per_wg,
                        int R, int uniform_w, const int32_t *d_dep, unsigned *d_flags,
                        unsigned *d_err, bool lowreg) {
    if (nphases <= 0 || nwg <= 0) return;
    launch_zero_bytes(s, d_flags, (size_t)prog_flag_words(nwg) * sizeof(unsigned));
    const dim3 grid(nwg), block(64 * waves_per_wg);
    hipLaunchKernelGGL(pick_program(R, uniform_w, lowreg), grid, block, 0, s, d_ops, nphases,
                       reinterpret_cast<const int2 *>(d_dep), d_flags, d_err);
}

// Batched steps whose RowOps all apply the SAME matrix (the mass-Chebyshev steps and the
// mass products of the preconditioner: one M for every time level): a thread loads its two
// rows' indices and values once and serves NB time levels with them, so the per-non-zero
// load issue -- the actual limiter of these launches, not HBM -- drops from (index, value,
// 2 gathers) to (2 gathers + 1/NB of the rest).
template <int W, int NB>
__global__ __launch_bounds__(256) void pc_rows_shared(const RowOp *__restrict__ ops, int nops,
                                                      const int xcd) {
    constexpr int R = 2, C = 128;
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int g0 = blockIdx.y * NB;
    const RowOp &op0 = ops[g0];
    const int wg = xcd_workgroup((op0.nslices + 3) >> 2, xcd);
    if (wg < 0) return;
    const int s = wg * 4 + wave;
    if (s >= op0.nslices) return;
    const size_t base = (size_t)s * W * C + (size_t)lane * R;
    const int r0 = s * C + lane;
    const int nrows = op0.nrows;
    const Bases bases{{nullptr, nullptr, nullptr, nullptr}};
    int c[W][R];
    double v[W][R];
    {
        const gci_p colp = (gci_p)op0.col + base;
        const gcd_p vp = (gcd_p)op0.t[0].vals + base;
#pragma unroll
        for (int k = 0; k < W; ++k) load_cols<R>(colp + (size_t)k * C, c[k]);
#pragma unroll
        for (int k = 0; k < W; ++k) load_vals<R, false>(vp + (size_t)k * C, v[k]);
    }
    const gcb_p rowmask = (gcb_p)op0.rowmask;
    bool masked[R];
#pragma unroll
    for (int q = 0; q < R; ++q) {
        const int r = r0 + 64 * q;
        masked[q] = r < nrows && rowmask != nullptr && rowmask[r] != 0;
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        if (g0 + b >= nops) break;
        const RowOp &op = ops[g0 + b];
        const bool lin = op.mode == EPI_LIN;
        gcd_p pa, pb, pc, pd;
        if (lin) {
            pa = resolve(op.yin, bases);
            pb = resolve(op.z, bases);
            pc = resolve(op.mx, bases);
            pd = op.y2.base >= 0 ? (gcd_p)op.dinv : nullptr;
        } else {
            pa = resolve(op.pkm1, bases);
            pb = resolve(op.pk, bases);
            pc = resolve(op.b, bases);
            pd = (gcd_p)op.dinv;
        }
        double e0[R], e1[R], e2[R], e3[R];
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const int r = r0 + 64 * q;
            const bool in = r < nrows;
            e0[q] = (in && pa) ? pa[r] : 0.0;
            e1[q] = (in && pb) ? pb[r] : 0.0;
            e2[q] = (in && pc) ? pc[r] : 0.0;
            e3[q] = (in && pd) ? pd[r] : 0.0;
        }
        const gcd_p x = resolve(op.t[0].x, bases);
        double xv[W][R];
#pragma unroll
        for (int k = 0; k < W; ++k)
#pragma unroll
            for (int q = 0; q < R; ++q) xv[k][q] = x[c[k][q]];
        double acc[R];
#pragma unroll
        for (int q = 0; q < R; ++q) acc[q] = 0.0;
#pragma unroll
        for (int k = 0; k < W; ++k)
#pragma unroll
            for (int q = 0; q < R; ++q) acc[q] = __builtin_fma(v[k][q], xv[k][q], acc[q]);
        const gd_p y = (gd_p)resolve(op.y, bases);
        const gd_p y2 = lin && op.y2.base >= 0 ? (gd_p)resolve(op.y2, bases) : nullptr;
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const int r = r0 + 64 * q;
            if (r >= nrows) continue;
            double out, out2 = 0.0;
            if (lin) {
                if (masked[q]) {
                    out = pc ? op.malpha * e2[q] : 0.0;
                } else {
                    out = op.ca * acc[q];
                    if (pa) out += op.cy * e0[q];
                    if (pb) out += op.cz * e1[q];
                }
                out2 = op.c3 * (e3[q] * out);
            } else if (masked[q]) {
                out = 0.0;
            } else {
                double t = pa ? op.c1 * e0[q] : 0.0;
                if (pb) t += op.c2 * e1[q];
                t += op.c3 * (e3[q] * (e2[q] - acc[q]));
                out = op.post2 * (op.post1 * t);
            }
            y[r] = out;
            if (y2) y2[r] = out2;
        }
    }
}

// The same for structures without one fixed width (P2 / Q2, row-sorted storage): indices and
// values are loaded in chunks of KC slots and serve NB time levels; every row keeps the fma
// chain of the plain kernel (ascending k), so the results are bit-identical.
template <int NB, int KC>
__global__ __launch_bounds__(256) void pc_rows_shared_g(const RowOp *__restrict__ ops, int nops,
                                                        const int xcd) {
    constexpr int R = 2, C = 128;
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int g0 = blockIdx.y * NB;
    const RowOp &op0 = ops[g0];
    const int wg = xcd_workgroup((op0.nslices + 3) >> 2, xcd);
    if (wg < 0) return;
    const int s = wg * 4 + wave;
    if (s >= op0.nslices) return;
    int off0, w;
    if (op0.uniform_w >= 0) {
        w = op0.uniform_w;
        off0 = s * w;
    } else {
        off0 = ((gci_p)op0.slice_off)[s];
        w = ((gci_p)op0.slice_off)[s + 1] - off0;
    }
    const size_t base = (size_t)off0 * C + (size_t)lane * R;
    const int nrows = op0.nrows;
    const Bases bases{{nullptr, nullptr, nullptr, nullptr}};
    int row[R];
    {
        const gci_p perm = (gci_p)op0.perm;
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const int pos = s * C + lane + 64 * q;
            row[q] = perm ? perm[pos] : (pos < nrows ? pos : -1);
        }
    }
    const gcb_p rowmask = (gcb_p)op0.rowmask;
    bool masked[R];
#pragma unroll
    for (int q = 0; q < R; ++q)
        masked[q] = row[q] >= 0 && rowmask != nullptr && rowmask[row[q]] != 0;
    const int nb = nops - g0 < NB ? nops - g0 : NB;
    gcd_p x[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) x[b] = resolve(ops[g0 + (b < nb ? b : 0)].t[0].x, bases);
    double acc[NB][R];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int q = 0; q < R; ++q) acc[b][q] = 0.0;
    const gci_p colp = (gci_p)op0.col + base;
    const gcd_p vp = (gcd_p)op0.t[0].vals + base;
    for (int k0 = 0; k0 < w; k0 += KC) {
        int c[KC][R];
        double v[KC][R];
#pragma unroll
        for (int k = 0; k < KC; ++k)
            if (k0 + k < w) {
                load_cols<R>(colp + (size_t)(k0 + k) * C, c[k]);
                load_vals<R, false>(vp + (size_t)(k0 + k) * C, v[k]);
            }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            double xv[KC][R];
#pragma unroll
            for (int k = 0; k < KC; ++k)
                if (k0 + k < w) {
#pragma unroll
                    for (int q = 0; q < R; ++q) xv[k][q] = x[b][c[k][q]];
                }
#pragma unroll
            for (int k = 0; k < KC; ++k)
                if (k0 + k < w) {
#pragma unroll
                    for (int q = 0; q < R; ++q)
                        acc[b][q] = __builtin_fma(v[k][q], xv[k][q], acc[b][q]);
                }
        }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        if (b >= nb) break;
        const RowOp &op = ops[g0 + b];
        const bool lin = op.mode == EPI_LIN;
        gcd_p pa, pb, pc, pd;
        if (lin) {
            pa = resolve(op.yin, bases);
            pb = resolve(op.z, bases);
            pc = resolve(op.mx, bases);
            pd = op.y2.base >= 0 ? (gcd_p)op.dinv : nullptr;
        } else {
            pa = resolve(op.pkm1, bases);
            pb = resolve(op.pk, bases);
            pc = resolve(op.b, bases);
            pd = (gcd_p)op.dinv;
        }
        const gd_p y = (gd_p)resolve(op.y, bases);
        const gd_p y2 = lin && op.y2.base >= 0 ? (gd_p)resolve(op.y2, bases) : nullptr;
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const int r = row[q];
            if (r < 0) continue;
            const double e0 = pa ? pa[r] : 0.0, e1 = pb ? pb[r] : 0.0;
            const double e2 = pc ? pc[r] : 0.0, e3 = pd ? pd[r] : 0.0;
            double out, out2 = 0.0;
            if (lin) {
                if (masked[q]) {
                    out = pc ? op.malpha * e2 : 0.0;
                } else {
                    out = op.ca * acc[b][q];
                    if (pa) out += op.cy * e0;
                    if (pb) out += op.cz * e1;
                }
                out2 = op.c3 * (e3 * out);
            } else if (masked[q]) {
                out = 0.0;
            } else {
                double t = pa ? op.c1 * e0 : 0.0;
                if (pb) t += op.c2 * e1;
                t += op.c3 * (e3 * (e2 - acc[b][q]));
                out = op.post2 * (op.post1 * t);
            }
            y[r] = out;
            if (y2) y2[r] = out2;
        }
    }
}

// ---- batched Chebyshev steps, four time levels interleaved (IlOp, kernels.hpp)
//
// The batched mass steps are bound by the number of L1 -> L2 requests in flight, and with four
// levels per thread seven of eight requests of a slot pair were the 8-byte gathers of the four
// iterates (profiles/r03/spmv_ragged_forms.md).  With the levels of a group interleaved a row's
// four values are one 32-byte load.  Same fma chain per (row, level) as every other form.
typedef double d4 __attribute__((ext_vector_type(4)));
typedef KKT_GLOBAL const d4 *gcd4_p;
typedef KKT_GLOBAL d4 *gd4_p;

template <int KC>
__global__ __launch_bounds__(256) void pc_rows_il(const IlOp *__restrict__ ops, const int xcd) {
    constexpr int R = 2, C = 128;
    const IlOp &op = ops[blockIdx.y];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wg = xcd_workgroup((op.nslices + 3) >> 2, xcd);
    if (wg < 0) return;
    const int s = wg * 4 + wave;
    if (s >= op.nslices) return;
    int off0, w;
    if (op.uniform_w >= 0) {
        w = op.uniform_w;
        off0 = s * w;
    } else {
        off0 = ((gci_p)op.slice_off)[s];
        w = ((gci_p)op.slice_off)[s + 1] - off0;
    }
    const size_t base = (size_t)off0 * C + (size_t)lane * R;
    int row[R];
    {
        const gci_p perm = (gci_p)op.perm;
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const int pos = s * C + lane + 64 * q;
            row[q] = perm ? perm[pos] : (pos < op.nrows ? pos : -1);
        }
    }
    const gcb_p rowmask = (gcb_p)op.rowmask;
    bool masked[R];
#pragma unroll
    for (int q = 0; q < R; ++q)
        masked[q] = row[q] >= 0 && rowmask != nullptr && rowmask[row[q]] != 0;
    d4 acc[R];
#pragma unroll
    for (int q = 0; q < R; ++q) acc[q] = d4{0.0, 0.0, 0.0, 0.0};
    const gcd4_p x4 = (gcd4_p)op.x;
    if (x4 != nullptr) {
        const gci_p colp = (gci_p)op.col + base;
        const gcd_p vp = (gcd_p)op.vals + base;
        for (int k0 = 0; k0 < w; k0 += KC) {
            int c[KC][R];
            double v[KC][R];
#pragma unroll
            for (int k = 0; k < KC; ++k)
                if (k0 + k < w) {
                    load_cols<R>(colp + (size_t)(k0 + k) * C, c[k]);
                    load_vals<R, false>(vp + (size_t)(k0 + k) * C, v[k]);
                }
            d4 xv[KC][R];
#pragma unroll
            for (int k = 0; k < KC; ++k)
                if (k0 + k < w) {
#pragma unroll
                    for (int q = 0; q < R; ++q) xv[k][q] = x4[c[k][q]];
                }
#pragma unroll
            for (int k = 0; k < KC; ++k)
                if (k0 + k < w) {
#pragma unroll
                    for (int q = 0; q < R; ++q) {
                        acc[q].x = __builtin_fma(v[k][q], xv[k][q].x, acc[q].x);
                        acc[q].y = __builtin_fma(v[k][q], xv[k][q].y, acc[q].y);
                        acc[q].z = __builtin_fma(v[k][q], xv[k][q].z, acc[q].z);
                        acc[q].w = __builtin_fma(v[k][q], xv[k][q].w, acc[q].w);
                    }
                }
        }
    }
    const gcd4_p pkm1 = (gcd4_p)op.pkm1;
    const gcd_p dinv = (gcd_p)op.dinv;
#pragma unroll
    for (int q = 0; q < R; ++q) {
        const int r = row[q];
        if (r < 0) continue;
        double o[4];
        if (masked[q]) {
#pragma unroll
            for (int l = 0; l < 4; ++l) o[l] = 0.0;
        } else {
            const d4 e1 = x4 ? x4[r] : d4{0.0, 0.0, 0.0, 0.0};
            const d4 e0 = pkm1 ? pkm1[r] : d4{0.0, 0.0, 0.0, 0.0};
            const double e3 = dinv[r];
            const double a_[4] = {acc[q].x, acc[q].y, acc[q].z, acc[q].w};
            const double p0_[4] = {e0.x, e0.y, e0.z, e0.w}, p1_[4] = {e1.x, e1.y, e1.z, e1.w};
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                const double e2 = l < op.nlev ? ((gcd_p)op.b[l])[r] : 0.0;
                double t = pkm1 ? op.c1 * p0_[l] : 0.0;
                if (x4) t += op.c2 * p1_[l];
                t += op.c3 * (e3 * (e2 - a_[l]));
                o[l] = op.post2[l] * (op.post1[l] * t);
            }
        }
        if (op.out[0] != nullptr) {
#pragma unroll
            for (int l = 0; l < 4; ++l)
                if (l < op.nlev) ((gd_p)op.out[l])[r] = o[l];
        } else {
            ((gd4_p)op.y)[r] = d4{o[0], o[1], o[2], o[3]};
        }
    }
}

void launch_rowops_il(hipStream_t s, const IlOp *d_ops, int ngroups, int max_slices, int uniform_w) {
    if (ngroups <= 0 || max_slices <= 0) return;
    const int gx = (max_slices + 3) / 4, xcd = g_pc_xcd ? 1 : 0;
    const dim3 grid(xcd ? (gx + 7) / 8 * 8 : gx, ngroups), block(256);
    if (uniform_w > 0 && uniform_w <= 8)
        hipLaunchKernelGGL((pc_rows_il<8>), grid, block, 0, s, d_ops, xcd);
    else
        hipLaunchKernelGGL((pc_rows_il<4>), grid, block, 0, s, d_ops, xcd);
}

// KKT operator apply with shared values ("mode S", time-invariant blocks): block rows whose
// terms use the same matrices in the same order -- the interior rows of the BE / CN stencils,
// control.py:2907-2978 -- differ only in the vectors they read and write.  One thread loads the
// indices and values of its two rows once per term and serves up to NB block rows with them
// (SpMM shape: the per-non-zero load issue drops from index + value + gather to
// gather + (index + value) / NB).  Every block row keeps the fma chain of kkt_spmv_rows (terms
// in order, CSR order inside a term): bit-identical results.
// `groups[g]` = {first op, count <= NB} of a run of RowOps with identical structure.
template <int W, int NB>
__global__ __launch_bounds__(256) void kkt_spmv_rows_shared(const RowOp *__restrict__ ops,
                                                            const int2 *__restrict__ groups,
                                                            const Bases bases) {
    constexpr int R = 2, C = 128;
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int s = blockIdx.x * 4 + wave;
    const int2 grp = groups[blockIdx.y];
    const int g0 = grp.x, nb = grp.y;
    const RowOp &op0 = ops[g0];
    if (s >= op0.nslices) return;
    const size_t base = (size_t)s * W * C + (size_t)lane * R;
    const int r0 = s * C + lane;
    const int nrows = op0.nrows;
    double acc[NB][R];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int q = 0; q < R; ++q) acc[b][q] = 0.0;
    const gci_p colp = (gci_p)op0.col + base;
    int c[W][R];
    if (op0.nterms > 0) {
#pragma unroll
        for (int k = 0; k < W; ++k) load_cols<R>(colp + (size_t)k * C, c[k]);
    }
    for (int t = 0; t < op0.nterms; ++t) {
        const gcd_p vp = (gcd_p)op0.t[t].vals + base;
        double v[W][R];
#pragma unroll
        for (int k = 0; k < W; ++k) load_vals<R, false>(vp + (size_t)k * C, v[k]);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (b >= nb) break;
            const gcd_p x = resolve(ops[g0 + b].t[t].x, bases);
            double xv[W][R];
#pragma unroll
            for (int k = 0; k < W; ++k)
#pragma unroll
                for (int q = 0; q < R; ++q) xv[k][q] = x[c[k][q]];
#pragma unroll
            for (int k = 0; k < W; ++k)
#pragma unroll
                for (int q = 0; q < R; ++q) acc[b][q] = __builtin_fma(v[k][q], xv[k][q], acc[b][q]);
        }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        if (b >= nb) break;
        const RowOp &op = ops[g0 + b];
        const gcd_p pa = resolve(op.yin, bases), pb = resolve(op.z, bases),
                    pc = resolve(op.mx, bases);
        const gcb_p rowmask = (gcb_p)op.rowmask;
        const gd_p y = (gd_p)resolve(op.y, bases);
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const int r = r0 + 64 * q;
            if (r >= nrows) continue;
            const bool masked = rowmask != nullptr && rowmask[r] != 0;
            double out;
            if (masked) {
                out = pc ? op.malpha * pc[r] : 0.0;
            } else {
                double vv = op.ca * acc[b][q];
                if (pa) vv += op.cy * pa[r];
                if (pb) vv += op.cz * pb[r];
                out = vv;
            }
            y[r] = out;
        }
    }
}

bool launch_rowops_grouped(hipStream_t s, const RowOp *d_ops, const int32_t *d_groups, int ngroups,
                           int max_slices, int R, int uniform_w, const Bases &bases) {
    constexpr int NB = 4;
    if (R != 2 || ngroups <= 0) return false;
    const dim3 grid((max_slices + 3) / 4, ngroups), block(256);
    const int2 *g = reinterpret_cast<const int2 *>(d_groups);
    switch (uniform_w) {
#define KKT_W(n) case n: hipLaunchKernelGGL((kkt_spmv_rows_shared<n, NB>), grid, block, 0, s, d_ops, g, bases); return true;
        KKT_W(1) KKT_W(2) KKT_W(3) KKT_W(4) KKT_W(5) KKT_W(6) KKT_W(7) KKT_W(8)
#undef KKT_W
        default: return false;
    }
}

bool launch_rowops_shared(hipStream_t s, const RowOp *d_ops, int nops, int max_slices, int R,
                          int uniform_w) {
    constexpr int NB = 4;
    if (R != 2 || nops < NB) return false;
    const int gx = (max_slices + 3) / 4, xcd = g_pc_xcd ? 1 : 0;
    const dim3 grid(xcd ? (gx + 7) / 8 * 8 : gx, (nops + NB - 1) / NB), block(256);
    switch (uniform_w) {
#define KKT_W(n) case n: hipLaunchKernelGGL((pc_rows_shared<n, NB>), grid, block, 0, s, d_ops, nops, xcd); return true;
        KKT_W(1) KKT_W(2) KKT_W(3) KKT_W(4) KKT_W(5) KKT_W(6) KKT_W(7) KKT_W(8)
#undef KKT_W
        default:
            hipLaunchKernelGGL((pc_rows_shared_g<NB, 4>), grid, block, 0, s, d_ops, nops, xcd);
            return true;
    }
}

static bool g_apply_xcd = false;
void set_apply_xcd(bool on) { g_apply_xcd = on; }
template <int R, int WFIX>
static void launch_one(hipStream_t s, dim3 grid, const RowOp *d_ops, const Bases &bases, int tag,
                       const RowOp *h_single) {
    if (h_single && tag != 0)
        hipLaunchKernelGGL((pc_row_step<R, WFIX>), grid, dim3(256), 0, s, *h_single, bases);
    else if (tag == 0)
        hipLaunchKernelGGL((kkt_spmv_rows<R, WFIX>),
                           dim3(g_apply_xcd ? (grid.x + 7) / 8 * 8 : grid.x, grid.y), dim3(256), 0, s,
                           d_ops, bases, g_apply_xcd ? 1 : 0);
    else
        hipLaunchKernelGGL((pc_rows<R, WFIX>), grid, dim3(256), 0, s, d_ops, bases);
}

static bool g_ragged_xcd = true;
void set_ragged_xcd(bool on) { g_ragged_xcd = on; }

void launch_rowops(hipStream_t s, const RowOp *d_ops, int nops, int max_slices, int R,
                   const Bases &bases, int tag, int uniform_w, const RowOp *h_single) {
    if (nops <= 0 || max_slices <= 0) return;
    dim3 grid((max_slices + 3) / 4, nops);
    if (nops != 1) h_single = nullptr;
    if (R != 2) {
        launch_one<1, 0>(s, grid, d_ops, bases, tag, h_single);
        return;
    }
    switch (uniform_w) {
#define KKT_W(n) case n: launch_one<2, n>(s, grid, d_ops, bases, tag, h_single); break;
        KKT_W(1) KKT_W(2) KKT_W(3) KKT_W(4) KKT_W(5) KKT_W(6) KKT_W(7) KKT_W(8)
        KKT_W(9) KKT_W(10) KKT_W(11) KKT_W(12) KKT_W(13) KKT_W(14) KKT_W(15) KKT_W(16)
#undef KKT_W
        case UNIFORM_W_SWITCH_NARROW:
            if (tag == 0) {
                const int per = g_ragged_xcd ? ((max_slices + 3) / 4 + 7) / 8 : 0;
                hipLaunchKernelGGL(kkt_spmv_rows_ragged_narrow, dim3(per ? 8 * per : grid.x, nops),
                                   dim3(256), 0, s, d_ops, bases, per);
                break;
            }
            launch_one<2, 0>(s, grid, d_ops, bases, tag, h_single);
            break;
        case UNIFORM_W_SWITCH:   // ragged, most slots in slices of a width the switch kernel unrolls
            if (tag == 0) {
                const int per = g_ragged_xcd ? ((max_slices + 3) / 4 + 7) / 8 : 0;
                hipLaunchKernelGGL(kkt_spmv_rows_ragged, dim3(per ? 8 * per : grid.x, nops),
                                   dim3(256), 0, s, d_ops, bases, per);
                break;
            }
            launch_one<2, 0>(s, grid, d_ops, bases, tag, h_single);
            break;
        case UNIFORM_W_SWITCH_1WAVE:
            if (tag == 0) {
                // one wave per workgroup: the slices of a window differ in width (19, 19, 12, 9, ...),
                // and a four-wave workgroup holds its registers until its widest slice is done
                const int per = g_ragged_xcd ? (max_slices + 7) / 8 : 0;
                hipLaunchKernelGGL(kkt_spmv_rows_ragged, dim3(per ? 8 * per : max_slices, nops),
                                   dim3(64), 0, s, d_ops, bases, per);
                break;
            }
            [[fallthrough]];
        default: launch_one<2, 0>(s, grid, d_ops, bases, tag, h_single); break;
    }
}

bool ragged_switch_width(int w) {
    switch (w) {
        case 24: case 38: return true;
#define KKT_WX(n) case n: return true;
        RAGGED_CASES
#undef KKT_WX
        default: return false;
    }
}

// ------------------------------------------------------------- value-array preparation

static inline int grid_for(int64_t n, int per_block = 256, int cap = 256 * 8) {
    int64_t g = (n + per_block - 1) / per_block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

__global__ void csr_to_sell_kernel(const double *__restrict__ csr,
                                   const int32_t *__restrict__ map,
                                   double *__restrict__ out, int64_t n) {
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < n;
         p += (int64_t)gridDim.x * blockDim.x) {
        const int32_t m = map[p];
        out[p] = m >= 0 ? csr[m] : 0.0;
    }
}
void launch_csr_to_sell(hipStream_t s, const double *csr_vals, const int32_t *sell2csr,
                        double *sell_vals, int64_t n_padded) {
    hipLaunchKernelGGL(csr_to_sell_kernel, dim3(grid_for(n_padded)), dim3(256), 0, s,
                       csr_vals, sell2csr, sell_vals, n_padded);
}

__global__ void mask_columns_kernel(double *__restrict__ vals,
                                    const int32_t *__restrict__ col,
                                    const uint8_t *__restrict__ colmask, int64_t n) {
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < n;
         p += (int64_t)gridDim.x * blockDim.x) {
        if (colmask[col[p]]) vals[p] = 0.0;
    }
}
void launch_mask_columns(hipStream_t s, double *sell_vals, const int32_t *col,
                         const uint8_t *colmask, int64_t n_padded) {
    hipLaunchKernelGGL(mask_columns_kernel, dim3(grid_for(n_padded)), dim3(256), 0, s,
                       sell_vals, col, colmask, n_padded);
}

__global__ void vals_axpy_kernel(double *__restrict__ out, const double *__restrict__ a,
                                 double c, const double *__restrict__ b, int64_t n) {
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < n;
         p += (int64_t)gridDim.x * blockDim.x) {
        out[p] = __dadd_rn(a ? a[p] : 0.0, __dmul_rn(c, b[p]));
    }
}
void launch_vals_axpy(hipStream_t s, double *out, const double *a, double c,
                      const double *b, int64_t n_padded) {
    hipLaunchKernelGGL(vals_axpy_kernel, dim3(grid_for(n_padded)), dim3(256), 0, s, out, a,
                       c, b, n_padded);
}

__global__ void vals_differ_kernel(const double *__restrict__ a, const double *__restrict__ b,
                                   int64_t n, unsigned *flag) {
    bool d = false;
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < n;
         p += (int64_t)gridDim.x * blockDim.x)
        d |= __double_as_longlong(a[p]) != __double_as_longlong(b[p]);
    if (d) atomicOr(flag, 1u);
}
void launch_vals_differ(hipStream_t s, const double *a, const double *b, int64_t n, unsigned *flag) {
    hipLaunchKernelGGL(vals_differ_kernel, dim3(grid_for(n)), dim3(256), 0, s, a, b, n, flag);
}

template <int R>
__global__ void extract_dinv_kernel(const int32_t *__restrict__ col,
                                    const int32_t *__restrict__ slice_off,
                                    const double *__restrict__ vals,
                                    const uint8_t *__restrict__ rowmask,
                                    double *__restrict__ dinv, int nrows, int nslices,
                                    const int32_t *__restrict__ perm) {
    constexpr int C = 64 * R;
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;   // position in the SELL storage
    if (pos >= nslices * C) return;
    const int r = perm ? perm[pos] : (pos < nrows ? pos : -1);
    if (r < 0) return;
    const int s = pos / C;
    const int rin = pos - s * C;
    const int within = (rin % 64) * R + rin / 64;
    const int off0 = slice_off[s];
    const int w = slice_off[s + 1] - off0;
    double d = 1.0;
    for (int k = 0; k < w; ++k) {
        const size_t p = ((size_t)off0 + k) * C + within;
        if (col[p] == r) {
            d = vals[p];
            break;
        }
    }
    const bool masked = rowmask != nullptr && rowmask[r] != 0;
    dinv[r] = masked ? 1.0 : 1.0 / d;
}
void launch_extract_dinv(hipStream_t s, const int32_t *col, const int32_t *slice_off,
                         const double *vals, const uint8_t *rowmask, double *dinv,
                         int nrows, int nslices, int R, const int32_t *perm) {
    dim3 grid((nslices * 64 * R + 255) / 256);
    if (R == 2)
        hipLaunchKernelGGL(extract_dinv_kernel<2>, grid, dim3(256), 0, s, col, slice_off,
                           vals, rowmask, dinv, nrows, nslices, perm);
    else
        hipLaunchKernelGGL(extract_dinv_kernel<1>, grid, dim3(256), 0, s, col, slice_off,
                           vals, rowmask, dinv, nrows, nslices, perm);
}

// ----------------------------------------------------------------------- vector kernels

__global__ void copy_kernel(double *__restrict__ y, const double *__restrict__ x,
                            int64_t n) {
    const int64_t n2 = n >> 1;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < n2; p += stride)
        reinterpret_cast<d2 *>(y)[p] = reinterpret_cast<const d2 *>(x)[p];
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) y[n - 1] = x[n - 1];
}
void launch_copy(hipStream_t s, double *y, const double *x, int64_t n) {
    if (n <= 0 || y == x) return;
    hipLaunchKernelGGL(copy_kernel, dim3(grid_for(n >> 1)), dim3(256), 0, s, y, x, n);
}

__global__ void fill_kernel(double *__restrict__ y, double v, int64_t n) {
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < n;
         p += (int64_t)gridDim.x * blockDim.x)
        y[p] = v;
}
__global__ void add_constant_kernel(double *__restrict__ y, double v, int64_t n) {
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < n;
         p += (int64_t)gridDim.x * blockDim.x)
        y[p] += v;
}
void launch_add_constant(hipStream_t s, double *y, double v, int64_t n) {
    hipLaunchKernelGGL(add_constant_kernel, dim3(grid_for(n)), dim3(256), 0, s, y, v, n);
}
// (sizes are multiples of 4 bytes: flag words and 8-byte granule halves)
__global__ void zero_words_kernel(unsigned *__restrict__ p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x)
        p[i] = 0u;
}
void launch_zero_bytes(hipStream_t s, void *p, size_t nbytes) {
    const size_t n = nbytes / 4;
    if (n == 0) return;
    hipLaunchKernelGGL(zero_words_kernel, dim3(grid_for((int64_t)n)), dim3(256), 0, s,
                       (unsigned *)p, n);
}

void launch_fill(hipStream_t s, double *y, double v, int64_t n) {
    if (n <= 0) return;
    hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n)), dim3(256), 0, s, y, v, n);
}

__global__ void axpby_kernel(double *__restrict__ y, double a, const double *__restrict__ x,
                             double b, int64_t n) {
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < n;
         p += (int64_t)gridDim.x * blockDim.x)
        y[p] = a * x[p] + b * y[p];
}
void launch_axpby(hipStream_t s, double *y, double a, const double *x, double b,
                  int64_t n) {
    if (n <= 0) return;
    hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n)), dim3(256), 0, s, y, a, x, b, n);
}

__global__ void mask_blocks_kernel(double *__restrict__ y, const double *__restrict__ x,
                                   const double *__restrict__ mx,
                                   const MaskJob *__restrict__ jobs, int64_t nx) {
    const MaskJob job = jobs[blockIdx.y];
    const int64_t off = (int64_t)blockIdx.y * nx;
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < nx;
         r += (int64_t)gridDim.x * blockDim.x) {
        const bool masked = job.mask != nullptr && job.mask[r] != 0;
        double v;
        if (masked)
            v = mx ? job.alpha * mx[off + r] : 0.0;
        else
            v = x[off + r];
        y[off + r] = v;
    }
}
void launch_mask_blocks(hipStream_t s, double *y, const double *x, const double *mx,
                        const MaskJob *d_jobs, int nblocks, int64_t nx) {
    if (nblocks <= 0 || nx <= 0) return;
    dim3 grid(grid_for(nx, 256, 64), nblocks);
    hipLaunchKernelGGL(mask_blocks_kernel, grid, dim3(256), 0, s, y, x, mx, d_jobs, nx);
}

__global__ void time_transform_kernel(double *__restrict__ y, const double *__restrict__ x,
                                      int kind, int n, int64_t nx,
                                      const double *__restrict__ lo_halo,
                                      const double *__restrict__ hi_halo) {
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < nx;
         r += (int64_t)gridDim.x * blockDim.x) {
        if (kind == 1) {          // T_1: new_i = old_i + old_{i+1}
            double cur = x[r];
            for (int i = 0; i < n; ++i) {
                double nxt = 0.0;
                if (i + 1 < n)
                    nxt = x[(int64_t)(i + 1) * nx + r];
                else if (hi_halo)
                    nxt = hi_halo[r];
                const bool has = (i + 1 < n) || hi_halo;
                y[(int64_t)i * nx + r] = has ? cur + nxt : cur;
                cur = nxt;
            }
        } else if (kind == 2) {   // T_2: new_i = old_i + old_{i-1}
            double prev = lo_halo ? lo_halo[r] : 0.0;
            bool has = lo_halo != nullptr;
            for (int i = 0; i < n; ++i) {
                const double cur = x[(int64_t)i * nx + r];
                y[(int64_t)i * nx + r] = has ? cur + prev : cur;
                prev = cur;
                has = true;
            }
        } else if (kind == 3) {   // T_1^{-1}: for i = n-2..0: x_i -= x_{i+1} (updated)
            double nxt = hi_halo ? hi_halo[r] : 0.0;
            bool has = hi_halo != nullptr;
            for (int i = n - 1; i >= 0; --i) {
                double cur = x[(int64_t)i * nx + r];
                if (has) cur -= nxt;
                y[(int64_t)i * nx + r] = cur;
                nxt = cur;
                has = true;
            }
        } else {                  // T_2^{-1}: for i = 1..n-1: x_i -= x_{i-1} (updated)
            double prev = lo_halo ? lo_halo[r] : 0.0;
            bool has = lo_halo != nullptr;
            for (int i = 0; i < n; ++i) {
                double cur = x[(int64_t)i * nx + r];
                if (has) cur -= prev;
                y[(int64_t)i * nx + r] = cur;
                prev = cur;
                has = true;
            }
        }
    }
}
// T_1 / T_2 out of place, with the Dirichlet post-correction of the operator fused
// (preconditioner.py:437-470 then 527-537): y_i = masked ? alpha * xin_i : t_i + t_{i+-1}.
// One thread per (level, dof): the in-place form above walks the levels serially per dof (66 049
// threads for 63 dependent loads each); raw rows `t` live in a buffer of their own, so every
// output is independent and the pass runs at the HBM rate (16 B read, 8 B written per unknown).
__global__ __launch_bounds__(256) void time_transform_mask_kernel(
    double *__restrict__ y, const double *__restrict__ t, const double *__restrict__ xin,
    const MaskJob *__restrict__ jobs, int kind, int n, int64_t nx,
    const double *__restrict__ lo_halo, const double *__restrict__ hi_halo) {
    const int i = blockIdx.y;
    const MaskJob job = jobs[i];
    const int64_t off = (int64_t)i * nx;
    // 2 dofs per thread: 16-byte accesses
    for (int64_t r = 2 * (blockIdx.x * (int64_t)blockDim.x + threadIdx.x); r < nx;
         r += 2 * (int64_t)gridDim.x * blockDim.x) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int64_t rr = r + q;
            if (rr >= nx) break;
            const double cur = t[off + rr];
            double other = 0.0;
            bool has;
            if (kind == 1) {
                has = (i + 1 < n) || hi_halo != nullptr;
                if (i + 1 < n) other = t[off + nx + rr];
                else if (hi_halo) other = hi_halo[rr];
            } else {
                has = i > 0 || lo_halo != nullptr;
                if (i > 0) other = t[off - nx + rr];
                else if (lo_halo) other = lo_halo[rr];
            }
            double v = has ? cur + other : cur;
            if (job.mask != nullptr && job.mask[rr] != 0) v = job.alpha * xin[off + rr];
            y[off + rr] = v;
        }
    }
}
void launch_time_transform_mask(hipStream_t s, double *y, const double *t, const double *xin,
                                const MaskJob *d_jobs, int kind, int n, int64_t nx,
                                const double *lo_halo, const double *hi_halo) {
    if (n <= 0 || nx <= 0) return;
    dim3 grid(grid_for((nx + 1) / 2, 256, 256), n);
    hipLaunchKernelGGL(time_transform_mask_kernel, grid, dim3(256), 0, s, y, t, xin, d_jobs, kind,
                       n, nx, lo_halo, hi_halo);
}

void launch_time_transform(hipStream_t s, double *y, const double *x, int kind, int n,
                           int64_t nx, const double *lo_halo, const double *hi_halo) {
    if (n <= 0 || nx <= 0) return;
    hipLaunchKernelGGL(time_transform_kernel, dim3(grid_for(nx)), dim3(256), 0, s, y, x,
                       kind, n, nx, lo_halo, hi_halo);
}

// one workgroup per block: fixed-order sum (deterministic)
__global__ void block_sums_kernel(const double *__restrict__ x, double *__restrict__ sums,
                                  int64_t nx) {
    __shared__ double sh[256];
    const double *xb = x + (int64_t)blockIdx.x * nx;
    double a = 0.0;
    for (int64_t r = threadIdx.x; r < nx; r += 256) a += xb[r];
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[blockIdx.x] = sh[0];
}
void launch_block_sums(hipStream_t s, const double *x, double *sums, int n, int64_t nx,
                       double *) {
    if (n <= 0) return;
    hipLaunchKernelGGL(block_sums_kernel, dim3(n), dim3(256), 0, s, x, sums, nx);
}
__global__ void block_shift_kernel(double *__restrict__ y, const double *__restrict__ sums,
                                   double coef, int64_t nx) {
    const double sh = coef * sums[blockIdx.y];
    double *yb = y + (int64_t)blockIdx.y * nx;
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < nx;
         r += (int64_t)gridDim.x * blockDim.x)
        yb[r] += sh;
}
void launch_block_shift(hipStream_t s, double *y, const double *sums, double coef, int n,
                        int64_t nx) {
    if (n <= 0) return;
    dim3 grid(grid_for(nx, 256, 64), n);
    hipLaunchKernelGGL(block_shift_kernel, grid, dim3(256), 0, s, y, sums, coef, nx);
}

// ConstantNullspace on all its blocks at once (preconditioner.py:137-152): sums of block j of
// `a` (and of `b` when given) by one workgroup each, in a fixed order; then
// y_j = (y_j + c1_j * sum_a_j) + c2_j * sum_b_j, the two additions the per-block form made.
__global__ void const_sums_kernel(const ConstJob *__restrict__ jobs, int njobs,
                                  const double *__restrict__ a, const double *__restrict__ b,
                                  double *__restrict__ sums) {
    __shared__ double sh[256];
    const ConstJob j = jobs[blockIdx.x];
    const double *xb = (blockIdx.y == 0 ? a : b) + j.off;
    double acc = 0.0;
    // one chain per thread, eight loads in flight (the additions keep their order)
#pragma unroll 8
    for (int64_t r = threadIdx.x; r < j.nx; r += 256) acc += xb[r];
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[blockIdx.y * njobs + blockIdx.x] = sh[0];
}
__global__ void const_shift_kernel(const ConstJob *__restrict__ jobs, int njobs,
                                   double *y, const double *src, const double *__restrict__ sums,
                                   int second) {
    const ConstJob j = jobs[blockIdx.y];
    const double s1 = j.c1 * sums[blockIdx.y];
    const double s2 = second ? (second == 2 ? j.c2_alpha : j.c2_one) * sums[njobs + blockIdx.y] : 0.0;
    double *yb = y + j.off;
    const double *sb = src + j.off;
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < j.nx;
         r += (int64_t)gridDim.x * blockDim.x) {
        double v = sb[r] + s1;
        if (second) v += s2;
        yb[r] = v;
    }
}
void launch_const_correct(hipStream_t s, const ConstJob *d_jobs, int njobs, int64_t max_nx,
                          double *y, const double *b, int second, double *sums) {
    if (njobs <= 0) return;
    hipLaunchKernelGGL(const_sums_kernel, dim3(njobs, second ? 2 : 1), dim3(256), 0, s, d_jobs,
                       njobs, y, b, sums);
    hipLaunchKernelGGL(const_shift_kernel, dim3(grid_for(max_nx, 256, 64), njobs), dim3(256), 0,
                       s, d_jobs, njobs, y, y, sums, second);
}
// xc_j = x_j - mean(x_j) on the ConstantNullspace blocks only, out of place (the other blocks of xc
// are not written: the operator's terms read them from x itself)
void launch_const_center(hipStream_t s, const ConstJob *d_jobs, int njobs, int64_t max_nx,
                         const double *x, double *xc, double *sums) {
    if (njobs <= 0) return;
    hipLaunchKernelGGL(const_sums_kernel, dim3(njobs, 1), dim3(256), 0, s, d_jobs, njobs, x,
                       (const double *)nullptr, sums);
    hipLaunchKernelGGL(const_shift_kernel, dim3(grid_for(max_nx, 256, 64), njobs), dim3(256), 0,
                       s, d_jobs, njobs, xc, x, sums, 0);
}

// -------------------------------------------------------------------------- reductions

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// stage 1: REDUCE_BLOCKS workgroups, each a fixed contiguous chunk; lane partials are
// combined by wavefront shuffles, then the 4 wave results in LDS, in a fixed order.
template <int NV>
__global__ __launch_bounds__(256) void mdot_stage1(const double *__restrict__ w, VecList V,
                                                   int64_t n, double *__restrict__ part) {
    __shared__ double sh[4][MDOT_MAX];
    const int64_t chunk = ((n + REDUCE_BLOCKS - 1) / REDUCE_BLOCKS + 1) & ~(int64_t)1;
    const int64_t lo = (int64_t)blockIdx.x * chunk;
    int64_t hi = lo + chunk;
    if (hi > n) hi = n;
    double acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = 0.0;
    // chunk is even and lo is even -> 16-byte aligned double2 accesses
    int64_t p = lo + 2 * (int64_t)threadIdx.x;
    for (; p + 1 < hi; p += 512) {
        const d2 wv = *reinterpret_cast<const d2 *>(w + p);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const d2 vv = *reinterpret_cast<const d2 *>(V.v[i] + p);
            acc[i] = __builtin_fma(wv.x, vv.x, acc[i]);
            acc[i] = __builtin_fma(wv.y, vv.y, acc[i]);
        }
    }
    if (p < hi) {
        const double wv = w[p];
#pragma unroll
        for (int i = 0; i < NV; ++i) acc[i] = __builtin_fma(wv, V.v[i][p], acc[i]);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const double t = wave_sum(acc[i]);
        if (lane == 0) sh[wave][i] = t;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        const int i = threadIdx.x;
        part[(int64_t)blockIdx.x * MDOT_MAX + i] = ((sh[0][i] + sh[1][i]) + sh[2][i]) + sh[3][i];
    }
}

// stage 2: one workgroup per vector sums the REDUCE_BLOCKS partials in a fixed tree
__global__ __launch_bounds__(256) void mdot_stage2(const double *__restrict__ part,
                                                   double *__restrict__ out) {
    __shared__ double sh[256];
    const int i = blockIdx.x;
    double a = 0.0;
    for (int b = threadIdx.x; b < REDUCE_BLOCKS; b += 256) a += part[(int64_t)b * MDOT_MAX + i];
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[i] = sh[0];
}

void launch_mdot(hipStream_t s, const double *w, VecList V, int nv, int64_t n,
                 double *scratch, double *out) {
    if (nv <= 0) return;
    dim3 g(REDUCE_BLOCKS), b(256);
    switch (nv) {
        case 1: hipLaunchKernelGGL(mdot_stage1<1>, g, b, 0, s, w, V, n, scratch); break;
        case 2: hipLaunchKernelGGL(mdot_stage1<2>, g, b, 0, s, w, V, n, scratch); break;
        case 3: hipLaunchKernelGGL(mdot_stage1<3>, g, b, 0, s, w, V, n, scratch); break;
        case 4: hipLaunchKernelGGL(mdot_stage1<4>, g, b, 0, s, w, V, n, scratch); break;
        case 5: hipLaunchKernelGGL(mdot_stage1<5>, g, b, 0, s, w, V, n, scratch); break;
        case 6: hipLaunchKernelGGL(mdot_stage1<6>, g, b, 0, s, w, V, n, scratch); break;
        case 7: hipLaunchKernelGGL(mdot_stage1<7>, g, b, 0, s, w, V, n, scratch); break;
        default: hipLaunchKernelGGL(mdot_stage1<8>, g, b, 0, s, w, V, n, scratch); break;
    }
    hipLaunchKernelGGL(mdot_stage2, dim3(nv), b, 0, s, scratch, out);
}

__global__ void norm2_finish_kernel(const double *dot, double *out) {
    out[0] = sqrt(dot[0]);
}
void launch_norm2_finish(hipStream_t s, const double *dot, double *out) {
    hipLaunchKernelGGL(norm2_finish_kernel, dim3(1), dim3(1), 0, s, dot, out);
}

// symmetric and skew parts of a matrix on a structurally symmetric pattern: tpos[k] = position of
// the transposed entry of position k (-1: padding).  *nonsym is set when a pair of stored entries
// differs (pairs with an exact zero on one side are boundary columns zeroed by mask_columns: the
// products they take part in are masked anyway).
__global__ void vals_sym_skew_kernel(const double *__restrict__ a, const int32_t *__restrict__ tpos,
                                     double *__restrict__ h, double *__restrict__ sk, int64_t n,
                                     unsigned *__restrict__ nonsym) {
    bool any = false;
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < n;
         p += (int64_t)gridDim.x * blockDim.x) {
        const int32_t t = tpos[p];
        const double x = a[p];
        if (t < 0) {
            h[p] = x;
            sk[p] = 0.0;
            continue;
        }
        const double y = a[t];
        h[p] = 0.5 * (x + y);
        sk[p] = 0.5 * (x - y);
        if (x != 0.0 && y != 0.0 && fabs(x - y) > 1e-12 * (fabs(x) + fabs(y))) any = true;
    }
    if (any) atomicOr(nonsym, 1u);
}
void launch_vals_sym_skew(hipStream_t s, const double *a, const int32_t *tpos, double *h,
                          double *sk, int64_t n, unsigned *nonsym) {
    if (n <= 0) return;
    hipLaunchKernelGGL(vals_sym_skew_kernel, dim3(grid_for(n)), dim3(256), 0, s, a, tpos, h, sk, n,
                       nonsym);
}

// ---- coarse correction of the two-grid sub-solves (plain-launch form): r_c = P^T r,
// e_c = E^-1 r_c, x_out = x_in + P e_c.  Fixed summation orders (no atomics): a workgroup per
// coarse function sums its column of P in a fixed tree; a workgroup per row of E^-1 its row.
__global__ __launch_bounds__(256) void coarse_restrict_kernel(
    const int32_t *__restrict__ pt_ip, const int32_t *__restrict__ pt_ix,
    const double *__restrict__ pt_v, const double *__restrict__ r, double *__restrict__ rc,
    int stride) {
    __shared__ double sh[256];
    const int j = blockIdx.x;
    double a = 0.0;
    for (int32_t q = pt_ip[j] + threadIdx.x; q < pt_ip[j + 1]; q += 256)
        a = __builtin_fma(pt_v[q], r[pt_ix[q]], a);
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) rc[(size_t)j * stride] = sh[0];
}
__global__ __launch_bounds__(256) void coarse_dense_kernel(const double *__restrict__ einv,
                                                           const double *__restrict__ rc,
                                                           double *__restrict__ ec, int nc) {
    __shared__ double sh[256];
    const int j = blockIdx.x;
    const double *row = einv + (size_t)j * nc;
    double a = 0.0;
    for (int k = threadIdx.x; k < nc; k += 256) a = __builtin_fma(row[k], rc[k], a);
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) ec[j] = sh[0];
}
__global__ void coarse_prolong_kernel(const int32_t *__restrict__ p_ip,
                                      const int32_t *__restrict__ p_ix,
                                      const double *__restrict__ p_v,
                                      const double *__restrict__ ec,
                                      const double *__restrict__ x_in, double *__restrict__ x_out,
                                      int64_t n) {
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n;
         r += (int64_t)gridDim.x * blockDim.x) {
        double a = 0.0;
        for (int32_t q = p_ip[r]; q < p_ip[r + 1]; ++q) a = __builtin_fma(p_v[q], ec[p_ix[q]], a);
        x_out[r] = x_in ? x_in[r] + a : a;
    }
}
void launch_coarse_correction(hipStream_t s, const CoarseDev &c, const double *einv,
                              const double *r, const double *x_in, double *x_out, int64_t n) {
    if (c.nc <= 0) return;
    hipLaunchKernelGGL(coarse_restrict_kernel, dim3(c.nc), dim3(256), 0, s, c.pt_ip, c.pt_ix, c.pt_v,
                       r, c.rc, 1);
    hipLaunchKernelGGL(coarse_dense_kernel, dim3(c.nc), dim3(256), 0, s, einv, c.rc, c.ec, c.nc);
    hipLaunchKernelGGL(coarse_prolong_kernel, dim3(grid_for(n)), dim3(256), 0, s, c.p_ip, c.p_ix,
                       c.p_v, c.ec, x_in, x_out, n);
}
// The same correction for `nb` vectors `vstride` apart (the pressure blocks of the Stokes
// preconditioner): blockIdx.y picks the vector; every sum in the order of the kernels above, so a
// block's result equals the one-vector launches bit for bit.  rc / ec: nb * nc doubles each.
__global__ __launch_bounds__(256) void coarse_restrict_batched_kernel(
    const int32_t *__restrict__ pt_ip, const int32_t *__restrict__ pt_ix,
    const double *__restrict__ pt_v, const double *__restrict__ r, double *__restrict__ rc,
    int nc, int64_t vstride) {
    __shared__ double sh[256];
    const int j = blockIdx.x;
    const double *rb = r + (size_t)blockIdx.y * vstride;
    double a = 0.0;
    for (int32_t q = pt_ip[j] + threadIdx.x; q < pt_ip[j + 1]; q += 256)
        a = __builtin_fma(pt_v[q], rb[pt_ix[q]], a);
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) rc[(size_t)blockIdx.y * nc + j] = sh[0];
}
__global__ __launch_bounds__(256) void coarse_dense_batched_kernel(const double *__restrict__ einv,
                                                                   const double *__restrict__ rc,
                                                                   double *__restrict__ ec, int nc) {
    __shared__ double sh[256];
    const int j = blockIdx.x;
    const double *row = einv + (size_t)j * nc;
    const double *rcb = rc + (size_t)blockIdx.y * nc;
    double a = 0.0;
    for (int k = threadIdx.x; k < nc; k += 256) a = __builtin_fma(row[k], rcb[k], a);
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) ec[(size_t)blockIdx.y * nc + j] = sh[0];
}
__global__ void coarse_prolong_batched_kernel(const int32_t *__restrict__ p_ip,
                                              const int32_t *__restrict__ p_ix,
                                              const double *__restrict__ p_v,
                                              const double *__restrict__ ec,
                                              const double *__restrict__ x_in,
                                              double *__restrict__ x_out, int64_t n, int nc,
                                              int64_t vstride) {
    const double *ecb = ec + (size_t)blockIdx.y * nc;
    const size_t off = (size_t)blockIdx.y * vstride;
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n;
         r += (int64_t)gridDim.x * blockDim.x) {
        double a = 0.0;
        for (int32_t q = p_ip[r]; q < p_ip[r + 1]; ++q) a = __builtin_fma(p_v[q], ecb[p_ix[q]], a);
        x_out[off + r] = x_in ? x_in[off + r] + a : a;
    }
}
void launch_coarse_correction_batched(hipStream_t s, const CoarseDev &c, const double *einv,
                                      const double *r, const double *x_in, double *x_out, int64_t n,
                                      int nb, int64_t vstride) {
    if (c.nc <= 0 || nb <= 0) return;
    hipLaunchKernelGGL(coarse_restrict_batched_kernel, dim3(c.nc, nb), dim3(256), 0, s, c.pt_ip,
                       c.pt_ix, c.pt_v, r, c.rc, c.nc, vstride);
    hipLaunchKernelGGL(coarse_dense_batched_kernel, dim3(c.nc, nb), dim3(256), 0, s, einv, c.rc, c.ec,
                       c.nc);
    hipLaunchKernelGGL(coarse_prolong_batched_kernel, dim3(grid_for(n, 256, 256), nb), dim3(256), 0, s,
                       c.p_ip, c.p_ix, c.p_v, c.ec, x_in, x_out, n, c.nc, vstride);
}
// set-up: x = column k of P (dense), and one restricted column of A P into E (column-major scratch)
__global__ void coarse_column_kernel(const int32_t *__restrict__ pt_ip,
                                     const int32_t *__restrict__ pt_ix,
                                     const double *__restrict__ pt_v, int k,
                                     double *__restrict__ x) {
    for (int32_t q = pt_ip[k] + blockIdx.x * blockDim.x + threadIdx.x; q < pt_ip[k + 1];
         q += gridDim.x * blockDim.x)
        x[pt_ix[q]] = pt_v[q];
}
void launch_coarse_column(hipStream_t s, const CoarseDev &c, int k, double *x, int64_t n) {
    launch_zero_bytes(s, x, (size_t)n * sizeof(double));
    hipLaunchKernelGGL(coarse_column_kernel, dim3(8), dim3(256), 0, s, c.pt_ip, c.pt_ix, c.pt_v, k, x);
}
void launch_coarse_restrict(hipStream_t s, const CoarseDev &c, const double *r, double *rc,
                            int stride) {
    hipLaunchKernelGGL(coarse_restrict_kernel, dim3(c.nc), dim3(256), 0, s, c.pt_ip, c.pt_ix, c.pt_v,
                       r, rc, stride);
}

// Dense inverse by Gauss-Jordan with partial pivoting on the device (set-up of the two-grid
// sub-solves: (P^T A P)^-1, n up to a few thousand).  Three small launches per pivot, every sum
// in a fixed order.  a (n x n, row-major) is destroyed; inv receives the inverse; *flag != 0:
// singular.
__global__ __launch_bounds__(256) void gj_pivot_kernel(const double *__restrict__ a, int n, int c,
                                                       int *__restrict__ piv,
                                                       double *__restrict__ pivval,
                                                       unsigned *__restrict__ flag) {
    __shared__ double bv[256];
    __shared__ int bi[256];
    double best = -1.0;
    int at = c;
    for (int r = c + (int)threadIdx.x; r < n; r += 256) {
        const double v = fabs(a[(size_t)r * n + c]);
        if (v > best) {
            best = v;
            at = r;
        }
    }
    bv[threadIdx.x] = best;
    bi[threadIdx.x] = at;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) {
            const double o = bv[threadIdx.x + st];
            const int oi = bi[threadIdx.x + st];
            if (o > bv[threadIdx.x] || (o == bv[threadIdx.x] && oi < bi[threadIdx.x])) {
                bv[threadIdx.x] = o;
                bi[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        piv[0] = bi[0];
        pivval[0] = a[(size_t)bi[0] * n + c];
        if (!(bv[0] > 0.0) || !isfinite(bv[0])) atomicOr(flag, 1u);
    }
}
__global__ void gj_swap_scale_kernel(double *__restrict__ a, double *__restrict__ inv, int n, int c,
                                     const int *__restrict__ piv, double *__restrict__ colbuf) {
    const int p = piv[0];
    const double d = colbuf[n];     // the pivot, saved by gj_pivot_kernel (this kernel overwrites it)
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < 2 * n;
         j += gridDim.x * blockDim.x) {
        double *m = j < n ? a : inv;
        const int col = j < n ? j : j - n;
        const double vp = m[(size_t)p * n + col], vc = m[(size_t)c * n + col];
        m[(size_t)p * n + col] = vc;                     // row p <- old row c
        m[(size_t)c * n + col] = d != 0.0 ? vp / d : 0.0;    // row c <- old row p, scaled
    }
}
__global__ void gj_column_kernel(const double *__restrict__ a, int n, int c,
                                 double *__restrict__ colbuf) {
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x)
        colbuf[r] = r == c ? 0.0 : a[(size_t)r * n + c];
}
__global__ __launch_bounds__(256) void gj_eliminate_kernel(double *__restrict__ a,
                                                           double *__restrict__ inv, int n, int c,
                                                           const double *__restrict__ colbuf) {
    const int r = blockIdx.y;
    const double f = colbuf[r];
    if (f == 0.0) return;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < 2 * n; j += gridDim.x * blockDim.x) {
        double *m = j < n ? a : inv;
        const int col = j < n ? j : j - n;
        m[(size_t)r * n + col] -= f * m[(size_t)c * n + col];
    }
}
// Set-up check of the tile program's column ranges (TileCoarseDev::e_lo / e_hi): flag = 1 if row j
// of a coarse inverse has a non-zero entry outside [lo_j, hi_j).
__global__ __launch_bounds__(256) void einv_outside_kernel(const double *__restrict__ einv, int nc,
                                                           const int32_t *__restrict__ lo,
                                                           const int32_t *__restrict__ hi,
                                                           unsigned *__restrict__ flag) {
    const int j = blockIdx.x;
    const double *row = einv + (size_t)j * nc;
    const int l = lo[j], h = hi[j];
    bool bad = false;
    for (int q = threadIdx.x; q < nc; q += 256)
        if ((q < l || q >= h) && row[q] != 0.0) bad = true;
    if (bad) atomicOr(flag, 1u);
}
void launch_einv_outside(hipStream_t s, const double *einv, int nc, const int32_t *lo,
                         const int32_t *hi, unsigned *flag) {
    if (nc > 0) hipLaunchKernelGGL(einv_outside_kernel, dim3(nc), dim3(256), 0, s, einv, nc, lo, hi, flag);
}

__global__ void gj_identity_kernel(double *__restrict__ inv, int n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < (size_t)n * n;
         i += (size_t)gridDim.x * blockDim.x)
        inv[i] = (i / n == i % n) ? 1.0 : 0.0;
}
void launch_dense_inverse(hipStream_t s, double *a, double *inv, int n, int *d_piv,
                          double *d_colbuf, unsigned *d_flag) {
    hipLaunchKernelGGL(gj_identity_kernel, dim3(grid_for((int64_t)n * n)), dim3(256), 0, s, inv, n);
    const dim3 g1(grid_for(2 * n, 256, 64));
    const dim3 ge(grid_for(2 * n, 256, 8), n);
    for (int c = 0; c < n; ++c) {
        hipLaunchKernelGGL(gj_pivot_kernel, dim3(1), dim3(256), 0, s, a, n, c, d_piv, d_colbuf + n,
                           d_flag);
        hipLaunchKernelGGL(gj_swap_scale_kernel, g1, dim3(256), 0, s, a, inv, n, c, d_piv, d_colbuf);
        hipLaunchKernelGGL(gj_column_kernel, dim3(grid_for(n)), dim3(256), 0, s, a, n, c, d_colbuf);
        hipLaunchKernelGGL(gj_eliminate_kernel, ge, dim3(256), 0, s, a, inv, n, c, d_colbuf);
    }
}

// the time-out word of the sweep programs as a summand of the Krylov all-reduce (time shards)
__global__ void flag_to_double_kernel(const unsigned *flag, double *out) {
    out[0] = (flag != nullptr && flag[0] != 0u) ? 1.0 : 0.0;
}
void launch_flag_to_double(hipStream_t s, const unsigned *flag, double *out) {
    hipLaunchKernelGGL(flag_to_double_kernel, dim3(1), dim3(1), 0, s, flag, out);
}

template <int NV>
__global__ __launch_bounds__(256) void maxpy_kernel(double *__restrict__ w, VecList V,
                                                    const double *__restrict__ coef,
                                                    double sign, int64_t n) {
    double c[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) c[i] = coef[i];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < n; p += stride) {
        double a = 0.0;
#pragma unroll
        for (int i = 0; i < NV; ++i) a = __builtin_fma(c[i], V.v[i][p], a);
        w[p] = __builtin_fma(sign, a, w[p]);
    }
}
void launch_maxpy(hipStream_t s, double *w, VecList V, const double *coef, double sign,
                  int nv, int64_t n) {
    if (nv <= 0 || n <= 0) return;
    dim3 g(grid_for(n)), b(256);
    switch (nv) {
        case 1: hipLaunchKernelGGL(maxpy_kernel<1>, g, b, 0, s, w, V, coef, sign, n); break;
        case 2: hipLaunchKernelGGL(maxpy_kernel<2>, g, b, 0, s, w, V, coef, sign, n); break;
        case 3: hipLaunchKernelGGL(maxpy_kernel<3>, g, b, 0, s, w, V, coef, sign, n); break;
        case 4: hipLaunchKernelGGL(maxpy_kernel<4>, g, b, 0, s, w, V, coef, sign, n); break;
        case 5: hipLaunchKernelGGL(maxpy_kernel<5>, g, b, 0, s, w, V, coef, sign, n); break;
        case 6: hipLaunchKernelGGL(maxpy_kernel<6>, g, b, 0, s, w, V, coef, sign, n); break;
        case 7: hipLaunchKernelGGL(maxpy_kernel<7>, g, b, 0, s, w, V, coef, sign, n); break;
        default: hipLaunchKernelGGL(maxpy_kernel<8>, g, b, 0, s, w, V, coef, sign, n); break;
    }
}

// maxpy with the partial sums of the squared norm of the result, in the chunks and the
// summation order of mdot_stage1 (so the norm equals a separate mdot(w, w) bit for bit);
// followed by mdot_stage2 on one column.
template <int NV>
__global__ __launch_bounds__(256) void maxpy_norm_kernel(double *__restrict__ w, VecList V,
                                                         const double *__restrict__ coef,
                                                         double sign, int64_t n,
                                                         double *__restrict__ part) {
    __shared__ double sh[4];
    double c[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) c[i] = coef[i];
    const int64_t chunk = ((n + REDUCE_BLOCKS - 1) / REDUCE_BLOCKS + 1) & ~(int64_t)1;
    const int64_t lo = (int64_t)blockIdx.x * chunk;
    int64_t hi = lo + chunk;
    if (hi > n) hi = n;
    double acc = 0.0;
    int64_t p = lo + 2 * (int64_t)threadIdx.x;
    for (; p + 1 < hi; p += 512) {
        double ax = 0.0, ay = 0.0;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const d2 vv = *reinterpret_cast<const d2 *>(V.v[i] + p);
            ax = __builtin_fma(c[i], vv.x, ax);
            ay = __builtin_fma(c[i], vv.y, ay);
        }
        d2 wv = *reinterpret_cast<const d2 *>(w + p);
        wv.x = __builtin_fma(sign, ax, wv.x);
        wv.y = __builtin_fma(sign, ay, wv.y);
        *reinterpret_cast<d2 *>(w + p) = wv;
        acc = __builtin_fma(wv.x, wv.x, acc);
        acc = __builtin_fma(wv.y, wv.y, acc);
    }
    if (p < hi) {
        double a = 0.0;
#pragma unroll
        for (int i = 0; i < NV; ++i) a = __builtin_fma(c[i], V.v[i][p], a);
        const double wv = __builtin_fma(sign, a, w[p]);
        w[p] = wv;
        acc = __builtin_fma(wv, wv, acc);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const double t = wave_sum(acc);
    if (lane == 0) sh[wave] = t;
    __syncthreads();
    if (threadIdx.x == 0)
        part[(int64_t)blockIdx.x * MDOT_MAX] = ((sh[0] + sh[1]) + sh[2]) + sh[3];
}
void launch_maxpy_norm(hipStream_t s, double *w, VecList V, const double *coef, double sign,
                       int nv, int64_t n, double *scratch, double *out_sq) {
    if (nv <= 0 || n <= 0) return;
    dim3 g(REDUCE_BLOCKS), b(256);
    switch (nv) {
#define KKT_N(k) case k: hipLaunchKernelGGL(maxpy_norm_kernel<k>, g, b, 0, s, w, V, coef, sign, n, scratch); break;
        KKT_N(1) KKT_N(2) KKT_N(3) KKT_N(4) KKT_N(5) KKT_N(6) KKT_N(7)
        default: hipLaunchKernelGGL(maxpy_norm_kernel<8>, g, b, 0, s, w, V, coef, sign, n, scratch); break;
#undef KKT_N
    }
    hipLaunchKernelGGL(mdot_stage2, dim3(1), b, 0, s, scratch, out_sq);
}

__global__ void scale_inv_kernel(double *__restrict__ y, const double *__restrict__ x,
                                 const double *__restrict__ norm, int64_t n) {
    const double inv = 1.0 / norm[0];
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < n;
         p += (int64_t)gridDim.x * blockDim.x)
        y[p] = x[p] * inv;
}
void launch_scale_inv(hipStream_t s, double *y, const double *x, const double *norm,
                      int64_t n) {
    if (n <= 0) return;
    hipLaunchKernelGGL(scale_inv_kernel, dim3(grid_for(n)), dim3(256), 0, s, y, x, norm, n);
}

}  // namespace kkt
