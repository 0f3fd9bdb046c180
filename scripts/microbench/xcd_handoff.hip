// Micro-benchmark: per-phase latency of a neighbour hand-off chain between one-wave workgroups,
// (a) spread over all XCDs, (b) confined to one XCD; store flavours plain / sc1.
// Build: hipcc -O3 --offload-arch=gfx950 xcd_handoff.hip -o xcd_handoff
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef unsigned long long u64;
#define GLOBAL __attribute__((address_space(1)))

__device__ __forceinline__ int xcc_id() {
    return __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11));   // HW_REG_XCC_ID[3:0]
}

template <bool SC1_STORE>
__device__ __forceinline__ void put(u64 *p, u64 v) {
    if constexpr (SC1_STORE)
        __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
        __builtin_nontemporal_store(v, p);   // keeps nothing in L1; line stays in the XCD L2
}
__device__ __forceinline__ u64 get(const u64 *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sc1: L2-served
}

// mode 0: every workgroup takes part (index = blockIdx.x), grid == n
// mode 1: only workgroups on XCC `target` take part (ticket order), grid == 8 * n + slack
template <bool SC1_STORE, bool PLAIN_STORE>
__global__ __launch_bounds__(64) void chain(u64 *g0, u64 *g1, int n, int phases, int mode, int target,
                                            unsigned *counter, unsigned *err, int *xcc_of) {
    int i;
    const int lane = threadIdx.x;
    if (mode == 0) {
        i = blockIdx.x;
    } else {
        if (xcc_id() != target) return;
        unsigned t = 0;
        if (lane == 0) t = atomicAdd(counter, 1u);
        t = __shfl(t, 0);
        if ((int)t >= n) return;
        i = (int)t;
    }
    if (lane == 0) xcc_of[i] = xcc_id();
    const int L = i > 0 ? i - 1 : i, Rr = i + 1 < n ? i + 1 : i;
    u64 acc = (u64)(i * 64 + lane);
    // phase 0 payload
    {
        const u64 v = ((u64)0 << 32) | (acc & 0xffffffffu);
        if constexpr (PLAIN_STORE) g0[i * 64 + lane] = v; else put<SC1_STORE>(g0 + i * 64 + lane, v);
    }
    bool dead = false;
    for (int p = 1; p <= phases; ++p) {
        u64 *src = (p & 1) ? g0 : g1, *dst = (p & 1) ? g1 : g0;
        u64 a, b, c;
        unsigned spins = 0;
        const u64 want = (u64)(p - 1);
        do {
            a = get(src + L * 64 + lane);
            b = get(src + i * 64 + ((lane + 1) & 63));
            c = get(src + Rr * 64 + lane);
            const bool ok = (a >> 32) == want && (b >> 32) == want && (c >> 32) == want;
            if (__all(ok) || dead) break;
        } while (++spins < (1u << 22));
        if (spins >= (1u << 22)) { dead = true; if (lane == 0) atomicOr(err, 1u); }
        acc = (a + b + c) & 0xffffffffu;
        const u64 v = ((u64)p << 32) | acc;
        if constexpr (PLAIN_STORE) dst[i * 64 + lane] = v; else put<SC1_STORE>(dst + i * 64 + lane, v);
    }
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// 16-byte granules {lo, tag, hi, tag} through raw buffer loads / stores with sc1
template <int NLOADS, int LDAUX, int STAUX, bool LD16, bool ST16>
__global__ __launch_bounds__(64) void chain16(u64 *g0, u64 *g1, int n, int phases, int mode, int target,
                                              unsigned *counter, unsigned *err, int *xcc_of) {
    const int lane = threadIdx.x;
    const int i = blockIdx.x;
    if (lane == 0) xcc_of[i] = xcc_id();
    const unsigned bytes = (unsigned)n * 64 * 16;
    int nb[NLOADS];   // neighbour rows this lane gathers (wave i-2 .. i+2, clamped)
    for (int k = 0; k < NLOADS; ++k) {
        int w = i + (k % 5) - 2;
        w = w < 0 ? 0 : (w >= n ? n - 1 : w);
        nb[k] = w * 64 + ((lane + k) & 63);
    }
    unsigned acc = i * 64 + lane;
    {
        const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc((void *)g0, 0, (int)bytes, 0x00020000);
        const u32x4 v = {acc, 0u, acc, 0u};
        __builtin_amdgcn_raw_buffer_store_b128(v, r0, (i * 64 + lane) * 16, 0, 17);
    }
    bool dead = false;
    for (int p = 1; p <= phases; ++p) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)((p & 1) ? g0 : g1), 0, (int)bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void *)((p & 1) ? g1 : g0), 0, (int)bytes, 0x00020000);
        u32x4 g[NLOADS];
        unsigned spins = 0;
        const unsigned want = (unsigned)(p - 1);
        bool ok;
        do {
            ok = true;
#pragma unroll
            for (int k = 0; k < NLOADS; ++k) {
                if constexpr (LD16) {
                    g[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, nb[k] * 16, 0, LDAUX);
                } else {
                    const u64 *src = (p & 1) ? g0 : g1;
                    const u64 a = get(src + 2 * nb[k]), b = get(src + 2 * nb[k] + 1);
                    g[k] = u32x4{(unsigned)a, (unsigned)(a >> 32), (unsigned)b, (unsigned)(b >> 32)};
                }
            }
#pragma unroll
            for (int k = 0; k < NLOADS; ++k) ok &= g[k].y == want && g[k].w == want;
            if (__all(ok) || dead) break;
        } while (++spins < (1u << 22));
        if (spins >= (1u << 22)) { dead = true; if (lane == 0) atomicOr(err, 1u); }
        unsigned a = 0;
#pragma unroll
        for (int k = 0; k < NLOADS; ++k) a += g[k].x;
        const u32x4 v = {a, (unsigned)p, a ^ 1u, (unsigned)p};
        if constexpr (ST16) {
            __builtin_amdgcn_raw_buffer_store_b128(v, rd, (i * 64 + lane) * 16, 0, STAUX);
        } else {
            u64 *dst = (p & 1) ? g1 : g0;
            put<true>(dst + 2 * (i * 64 + lane), ((u64)p << 32) | a);
            put<true>(dst + 2 * (i * 64 + lane) + 1, ((u64)p << 32) | (a ^ 1u));
        }
    }
}
// same traffic with 8-byte relaxed agent atomics (2 per granule)
template <int NLOADS>
__global__ __launch_bounds__(64) void chain8x2(u64 *g0, u64 *g1, int n, int phases, int mode, int target,
                                               unsigned *counter, unsigned *err, int *xcc_of) {
    const int lane = threadIdx.x;
    const int i = blockIdx.x;
    if (lane == 0) xcc_of[i] = xcc_id();
    int nb[NLOADS];
    for (int k = 0; k < NLOADS; ++k) {
        int w = i + (k % 5) - 2;
        w = w < 0 ? 0 : (w >= n ? n - 1 : w);
        nb[k] = w * 64 + ((lane + k) & 63);
    }
    unsigned acc = i * 64 + lane;
    put<true>(g0 + 2 * (i * 64 + lane), (u64)acc);
    put<true>(g0 + 2 * (i * 64 + lane) + 1, (u64)acc);
    bool dead = false;
    for (int p = 1; p <= phases; ++p) {
        u64 *src = (p & 1) ? g0 : g1, *dst = (p & 1) ? g1 : g0;
        u64 ga[NLOADS], gb[NLOADS];
        unsigned spins = 0;
        const u64 want = (u64)(p - 1);
        bool ok;
        do {
            ok = true;
#pragma unroll
            for (int k = 0; k < NLOADS; ++k) { ga[k] = get(src + 2 * nb[k]); gb[k] = get(src + 2 * nb[k] + 1); }
#pragma unroll
            for (int k = 0; k < NLOADS; ++k) ok &= (ga[k] >> 32) == want && (gb[k] >> 32) == want;
            if (__all(ok) || dead) break;
        } while (++spins < (1u << 22));
        if (spins >= (1u << 22)) { dead = true; if (lane == 0) atomicOr(err, 1u); }
        unsigned a = 0;
#pragma unroll
        for (int k = 0; k < NLOADS; ++k) a += (unsigned)ga[k];
        put<true>(dst + 2 * (i * 64 + lane), ((u64)p << 32) | a);
        put<true>(dst + 2 * (i * 64 + lane) + 1, ((u64)p << 32) | (a ^ 1u));
    }
}

int main(int argc, char **argv) {
    const int phases = 4000;
    u64 *g0, *g1; unsigned *counter, *err; int *xcc;
    const int maxn = 2048;
    CHK(hipMalloc(&g0, maxn * 64 * 16)); CHK(hipMalloc(&g1, maxn * 64 * 16));
    CHK(hipMalloc(&counter, 4)); CHK(hipMalloc(&err, 4)); CHK(hipMalloc(&xcc, maxn * 4));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    auto run = [&](const char *name, auto kern, int n, int mode, int target) {
        for (int rep = 0; rep < 2; ++rep) {
            CHK(hipMemset(g0, 0xff, maxn * 64 * 16)); CHK(hipMemset(g1, 0xff, maxn * 64 * 16));
            CHK(hipMemset(counter, 0, 4)); CHK(hipMemset(err, 0, 4)); CHK(hipMemset(xcc, 0xff, maxn * 4));
            CHK(hipDeviceSynchronize());
            const int grid = mode == 0 ? n : 8 * n + 64;
            CHK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, 0, g0, g1, n, phases, mode, target, counter, err, xcc);
            CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize());
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            unsigned herr, hc; CHK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost)); CHK(hipMemcpy(&hc, counter, 4, hipMemcpyDeviceToHost));
            std::vector<int> hx(n); CHK(hipMemcpy(hx.data(), xcc, n * 4, hipMemcpyDeviceToHost));
            int hist[9] = {0}; for (int v : hx) hist[v >= 0 && v < 8 ? v : 8]++;
            if (rep == 1)
                printf("%-34s n=%4d mode=%d: %.3f us/phase err=%u tickets=%u xcc hist %d %d %d %d %d %d %d %d (none %d)\n",
                       name, n, mode, 1e3 * ms / phases, herr, hc, hist[0], hist[1], hist[2], hist[3], hist[4], hist[5], hist[6], hist[7], hist[8]);
        }
    };
    {
        const int n = 512;
        run("ld16 sc1 / st16 sc1", chain16<8, 16, 16, true, true>, n, 0, 0);
        run("ld16 sc0sc1 / st16 sc0sc1", chain16<8, 17, 17, true, true>, n, 0, 0);
        run("ld8x2 atomic / st16 sc1", chain16<8, 16, 16, false, true>, n, 0, 0);
        run("ld8x2 atomic / st16 sc0sc1", chain16<8, 16, 17, false, true>, n, 0, 0);
        run("ld16 sc1 / st8x2 atomic", chain16<8, 16, 16, true, false>, n, 0, 0);
        run("ld16 sc0sc1 / st8x2 atomic", chain16<8, 17, 16, true, false>, n, 0, 0);
        run("ld8x2 / st8x2 (atomics)", chain16<8, 16, 16, false, false>, n, 0, 0);
        run("8Bx2 atomics, 8 loads, all XCDs", chain8x2<8>, n, 0, 0);
        run("8Bx2 atomics, 14 loads, all XCDs", chain8x2<14>, n, 0, 0);
    }
    if (0)
    for (int n : {512}) {
        run("sc1 store, all XCDs", chain<true, false>, n, 0, 0);
        run("nt store, all XCDs", chain<false, false>, n, 0, 0);
        run("plain store, all XCDs", chain<false, true>, n, 0, 0);
        run("sc1 store, one XCD", chain<true, false>, n, 1, 0);
        run("nt store, one XCD", chain<false, false>, n, 1, 0);
        run("plain store, one XCD", chain<false, true>, n, 1, 0);
    }
    return 0;
}
