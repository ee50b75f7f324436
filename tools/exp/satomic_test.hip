// Standalone probe (not part of the product): do scalar-memory atomics (s_atomic_add) work on gfx950, where do they
// execute when the counter is touched by one XCD only, and what does a ticket cost while the CU streams?
//   hipcc --offload-arch=gfx950 -O3 -o satomic_test satomic_test.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t s_ticket(uint32_t* counter, uint32_t add) {
    uint32_t v = add;
    asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(counter) : "memory");
    return v;
}

// every wave draws K tickets from the counter of its XCD (counters 64 B apart); records them
__global__ void k_tickets(uint32_t* counters, uint32_t* out, uint32_t K, uint32_t per_xcd) {
    const uint32_t xcc = __builtin_amdgcn_s_getreg(63508) & 0xfu;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) / 64;
    uint32_t* c = counters + (per_xcd ? xcc * 16 : 0);
    for (uint32_t k = 0; k < K; ++k) {
        const uint32_t t = s_ticket(c, 1);
        if ((threadIdx.x & 63) == 0) out[(size_t)wave * K + k] = (xcc << 28) | t;
    }
}

// streaming copy where each wave draws its 64-float4 units as tickets (mode 1: scalar atomic per XCD, mode 2: vector
// atomic agent scope single counter, mode 0: static grid-stride)
template <int MODE>
__global__ __launch_bounds__(256) void k_stream(const f32x4* __restrict__ src, f32x4* __restrict__ dst, uint32_t units64,
                                               uint32_t* counters, uint32_t* xcd_base, uint32_t* xcd_end) {
    const uint32_t lane = threadIdx.x & 63;
    if constexpr (MODE == 0) {
        const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) / 64, nw = gridDim.x * blockDim.x / 64;
        for (uint32_t u = wave; u < units64; u += nw) {
            f32x4 a = __builtin_nontemporal_load(src + (size_t)u * 64 + lane);
            __builtin_nontemporal_store(a, dst + (size_t)u * 64 + lane);
        }
    } else if constexpr (MODE == 1) {
        const uint32_t xcc = __builtin_amdgcn_s_getreg(63508) & 0xfu;
        const uint32_t b = xcd_base[xcc], e = xcd_end[xcc];
        uint32_t* c = counters + xcc * 16;
        uint32_t t = b + s_ticket(c, 1);
        while (t < e) {
            f32x4 a = __builtin_nontemporal_load(src + (size_t)t * 64 + lane);
            const uint32_t tn = b + s_ticket(c, 1);
            __builtin_nontemporal_store(a, dst + (size_t)t * 64 + lane);
            t = tn;
        }
    } else {
        uint32_t t = 0;
        if (lane == 0) t = atomicAdd(counters, 1u);
        t = __builtin_amdgcn_readfirstlane(t);
        while (t < units64) {
            f32x4 a = __builtin_nontemporal_load(src + (size_t)t * 64 + lane);
            uint32_t tn = 0;
            if (lane == 0) tn = atomicAdd(counters, 1u);
            tn = __builtin_amdgcn_readfirstlane(tn);
            __builtin_nontemporal_store(a, dst + (size_t)t * 64 + lane);
            t = tn;
        }
    }
}

int main() {
    uint32_t *d_c, *d_out;
    CK(hipMalloc(&d_c, 4096));
    const uint32_t grid = 2048, block = 256, K = 8, waves = grid * block / 64;
    CK(hipMalloc(&d_out, (size_t)waves * K * 4));
    std::vector<uint32_t> h((size_t)waves * K);
    for (int per_xcd = 1; per_xcd >= 0; --per_xcd) {
        CK(hipMemset(d_c, 0, 4096));
        hipLaunchKernelGGL(k_tickets, dim3(grid), dim3(block), 0, 0, d_c, d_out, K, (uint32_t)per_xcd);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), d_out, h.size() * 4, hipMemcpyDeviceToHost));
        uint32_t hc[1024];
        CK(hipMemcpy(hc, d_c, 4096, hipMemcpyDeviceToHost));
        // per XCD: tickets must be a permutation of 0..n-1 when the counter is per XCD
        std::vector<std::vector<uint32_t>> by(16);
        for (uint32_t v : h) by[v >> 28].push_back(v & 0x0fffffff);
        bool ok = true; size_t tot = 0;
        for (int x = 0; x < 16; ++x) {
            if (by[x].empty()) continue;
            std::sort(by[x].begin(), by[x].end());
            bool perm = true;
            for (size_t i = 0; i < by[x].size(); ++i) perm &= by[x][i] == i;
            printf("  per_xcd=%d xcc %d: %zu tickets, permutation of 0..n-1: %s, final counter word %u\n", per_xcd, x, by[x].size(), perm ? "yes" : "NO", hc[per_xcd ? x * 16 : 0]);
            ok &= perm; tot += by[x].size();
        }
        if (!per_xcd) {   // one counter shared by all XCDs: coherent only if the atomic executes beyond the XCD's L2
            std::vector<uint32_t> all; for (auto& b : by) all.insert(all.end(), b.begin(), b.end());
            std::sort(all.begin(), all.end()); bool perm = true;
            for (size_t i = 0; i < all.size(); ++i) perm &= all[i] == i;
            printf("  shared counter: all %zu tickets a permutation: %s\n", all.size(), perm ? "yes" : "NO");
        }
    }
    // ticket cost under streaming: 100 MB copy (50 in / 50 out)
    const uint32_t units64 = 50u * 1000 * 1000 / 1024;
    f32x4 *src, *dst; CK(hipMalloc(&src, (size_t)units64 * 1024)); CK(hipMalloc(&dst, (size_t)units64 * 1024));
    CK(hipMemset(src, 1, (size_t)units64 * 1024));
    uint32_t hb[16], he[16];
    for (int x = 0; x < 8; ++x) { hb[x] = (uint32_t)((uint64_t)units64 * x / 8); he[x] = (uint32_t)((uint64_t)units64 * (x + 1) / 8); }
    uint32_t *d_b, *d_e; CK(hipMalloc(&d_b, 64)); CK(hipMalloc(&d_e, 64));
    CK(hipMemcpy(d_b, hb, 32, hipMemcpyHostToDevice)); CK(hipMemcpy(d_e, he, 32, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 3; ++mode) {
        for (uint32_t g : {1024u, 2048u}) {
            float best = 1e9f;
            for (int rep = 0; rep < 20; ++rep) {
                CK(hipMemsetAsync(d_c, 0, 4096, 0));
                CK(hipEventRecord(e0, 0));
                if (mode == 0) hipLaunchKernelGGL(k_stream<0>, dim3(g), dim3(256), 0, 0, src, dst, units64, d_c, d_b, d_e);
                if (mode == 1) hipLaunchKernelGGL(k_stream<1>, dim3(g), dim3(256), 0, 0, src, dst, units64, d_c, d_b, d_e);
                if (mode == 2) hipLaunchKernelGGL(k_stream<2>, dim3(g), dim3(256), 0, 0, src, dst, units64, d_c, d_b, d_e);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep >= 3) best = std::min(best, ms);
            }
            printf("stream mode %d grid %u: best %.2f us (%u units of 1 KiB)\n", mode, g, best * 1e3f, units64);
        }
    }
    // verify mode 1 copied everything
    CK(hipMemset(dst, 0, (size_t)units64 * 1024)); CK(hipMemset(d_c, 0, 4096));
    hipLaunchKernelGGL(k_stream<1>, dim3(2048), dim3(256), 0, 0, src, dst, units64, d_c, d_b, d_e);
    CK(hipDeviceSynchronize());
    std::vector<unsigned char> hd((size_t)units64 * 1024); CK(hipMemcpy(hd.data(), dst, hd.size(), hipMemcpyDeviceToHost));
    size_t bad = 0; for (unsigned char c : hd) bad += c != 1;
    printf("mode 1 copy: %zu wrong bytes\n", bad);
    return 0;
}
