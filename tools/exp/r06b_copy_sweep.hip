// Standalone probe (not part of the product): what is the fastest LONE launch that moves C4's traffic -- 60 MB read from five streams of
// 12 / 12 / 16 / 16 / 4 bytes per vertex, 40 MB written to three of 12 / 12 / 16 -- over 8 rotating buffer sets (800 MB: every byte HBM's)?
// The product's lbs_skin_dyn takes 18.0 - 19.6 us and bench.py's no-math copy 17.7 - 18.2; a long launch of the same mix reaches 0.74 - 0.79 of
// 8 TB/s.  This sweeps launch SHAPES of a no-math kernel with the product's access pattern: persistent grids of 1 - 8 workgroups per CU x
// 256 / 512 / 1024 threads with one or two units requested ahead, and one-shot grids (one 64-vertex unit per wave, the hardware deals the
// workgroups), each with nt / default loads and nt / sc1 / default stores.  Per-dispatch times from the dispatch's own start / stop events.
//   hipcc --offload-arch=gfx950 -O3 -o r06b_copy_sweep r06b_copy_sweep.hip && ./r06b_copy_sweep [sets]
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct Set { const float *pos, *nrm, *tan, *wgt; const uint32_t* idx; float *op, *on, *ot; };
struct Bufs { __amdgpu_buffer_rsrc_t pos, nrm, tan, wgt, idx, op, on, ot; };
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rs(const void* p, uint32_t bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000); }
__device__ __forceinline__ Bufs bufs(const Set& s, uint32_t n) {
    return Bufs{rs(s.pos, n * 12), rs(s.nrm, n * 12), rs(s.tan, n * 16), rs(s.wgt, n * 16), rs(s.idx, n * 4), rs(s.op, n * 12), rs(s.on, n * 12), rs(s.ot, n * 16)};
}
struct V { u32x3 p, n; u32x4 t, w; uint32_t id; };
template <int LA> __device__ __forceinline__ V ld(const Bufs& b, uint32_t v) {
    V r;
    r.p = __builtin_amdgcn_raw_buffer_load_b96(b.pos, v * 12u, 0, LA);
    r.n = __builtin_amdgcn_raw_buffer_load_b96(b.nrm, v * 12u, 0, LA);
    r.t = __builtin_amdgcn_raw_buffer_load_b128(b.tan, v * 16u, 0, LA);
    r.w = __builtin_amdgcn_raw_buffer_load_b128(b.wgt, v * 16u, 0, LA);
    r.id = __builtin_amdgcn_raw_buffer_load_b32(b.idx, v * 4u, 0, LA);
    return r;
}
template <int SA> __device__ __forceinline__ void st(const Bufs& b, uint32_t v, const V& r) {
    // (every loaded word reaches a store, so nothing is optimised away)
    __builtin_amdgcn_raw_buffer_store_b96(u32x3{r.p.x ^ r.w.x, r.p.y ^ r.id, r.p.z}, b.op, v * 12u, 0, SA);
    __builtin_amdgcn_raw_buffer_store_b96(u32x3{r.n.x ^ r.w.y, r.n.y ^ r.w.z, r.n.z ^ r.w.w}, b.on, v * 12u, 0, SA);
    __builtin_amdgcn_raw_buffer_store_b128(r.t, b.ot, v * 16u, 0, SA);
}

// persistent: workgroup b owns the contiguous unit range [b T / G, (b + 1) T / G); its waves take every WPB-th unit; AHEAD units requested ahead
template <int BLOCK, int AHEAD, int LA, int SA>
__global__ __launch_bounds__(BLOCK) void k_persist(Set s, uint32_t n, uint32_t units) {
    const Bufs b = bufs(s, n);
    constexpr uint32_t WPB = BLOCK / 64;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t u0 = (uint32_t)(((uint64_t)blockIdx.x * units) / gridDim.x), u1 = (uint32_t)(((uint64_t)(blockIdx.x + 1) * units) / gridDim.x);
    uint32_t u = u0 + wave;
    if (u >= u1) return;
    V a = ld<LA>(b, u * 64 + lane), c;
    bool hc = false;
    if (AHEAD == 2 && u + WPB < u1) { c = ld<LA>(b, (u + WPB) * 64 + lane); hc = true; }
    for (;;) {
        const uint32_t un = u + WPB * AHEAD;
        V nx;
        const bool hn = un < u1;
        if (hn) nx = ld<LA>(b, un * 64 + lane);
        asm volatile("" : "+v"(a.p), "+v"(a.n), "+v"(a.t), "+v"(a.w), "+v"(a.id));
        st<SA>(b, u * 64 + lane, a);
        if (AHEAD == 1) { if (!hn) break; a = nx; u = un; }
        else { if (!hc) break; a = c; u += WPB; c = nx; hc = hn; }
    }
}

// one-shot: one unit per wave, the dispatcher deals the workgroups
template <int BLOCK, int LA, int SA>
__global__ __launch_bounds__(BLOCK) void k_oneshot(Set s, uint32_t n, uint32_t units) {
    const Bufs b = bufs(s, n);
    const uint32_t u = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    if (u >= units) return;
    const uint32_t v = u * 64 + (threadIdx.x & 63);
    const V a = ld<LA>(b, v);
    st<SA>(b, v, a);
}

// one-shot, UPW units per wave all requested up front
template <int BLOCK, int UPW, int LA, int SA>
__global__ __launch_bounds__(BLOCK) void k_oneshot_n(Set s, uint32_t n, uint32_t units) {
    const Bufs b = bufs(s, n);
    const uint32_t w = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    V a[UPW];
#pragma unroll
    for (int k = 0; k < UPW; ++k) a[k] = ld<LA>(b, (w * UPW + k) * 64 + lane);     // (past the end: zeros, stores dropped)
#pragma unroll
    for (int k = 0; k < UPW; ++k) st<SA>(b, (w * UPW + k) * 64 + lane, a[k]);
}

static std::vector<Set> g_sets;
static uint32_t g_n = 1000000, g_units;
static hipStream_t g_s;
static hipEvent_t g_e0, g_e1;

template <typename F> static void run(const char* name, F launch) {
    std::vector<float> t;
    for (int it = 0; it < 8 + 200; ++it) {
        const Set& s = g_sets[it % g_sets.size()];
        launch(s);
        CK(hipGetLastError());
        CK(hipEventSynchronize(g_e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, g_e0, g_e1));
        if (it >= 8) t.push_back(ms * 1e3f);
    }
    std::sort(t.begin(), t.end());
    double sum = 0;
    for (float x : t) sum += x;
    printf("{\"form\": \"%s\", \"sets\": %zu, \"avg_us\": %.2f, \"median_us\": %.2f, \"min_us\": %.2f, \"frac_of_8TBps_avg\": %.3f}\n", name, g_sets.size(), sum / t.size(),
           t[t.size() / 2], t[0], 100e6 / (sum / t.size() * 1e-6) / 8e12);
    fflush(stdout);
}

#define PERSIST(BLOCK, AHEAD, LA, SA, PER_CU)                                                                                            \
    run("persistent block=" #BLOCK " ahead=" #AHEAD " load_aux=" #LA " store_aux=" #SA " per_cu=" #PER_CU, [&](const Set& s) {          \
        hipExtLaunchKernelGGL((k_persist<BLOCK, AHEAD, LA, SA>), dim3(256 * PER_CU), dim3(BLOCK), 0, g_s, g_e0, g_e1, 0, s, g_n, g_units); \
    })
#define ONESHOT(BLOCK, LA, SA)                                                                                                           \
    run("one-shot block=" #BLOCK " load_aux=" #LA " store_aux=" #SA, [&](const Set& s) {                                                 \
        hipExtLaunchKernelGGL((k_oneshot<BLOCK, LA, SA>), dim3((g_units + BLOCK / 64 - 1) / (BLOCK / 64)), dim3(BLOCK), 0, g_s, g_e0, g_e1, 0, s, g_n, g_units); \
    })
#define ONESHOTN(BLOCK, UPW, LA, SA)                                                                                                     \
    run("one-shot block=" #BLOCK " units_per_wave=" #UPW " load_aux=" #LA " store_aux=" #SA, [&](const Set& s) {                        \
        const uint32_t waves = (g_units + UPW - 1) / UPW;                                                                                \
        hipExtLaunchKernelGGL((k_oneshot_n<BLOCK, UPW, LA, SA>), dim3((waves + BLOCK / 64 - 1) / (BLOCK / 64)), dim3(BLOCK), 0, g_s, g_e0, g_e1, 0, s, g_n, g_units); \
    })

int main(int argc, char** argv) {
    const int n_sets = argc > 1 ? atoi(argv[1]) : 8;
    g_units = (g_n + 63) / 64;
    CK(hipStreamCreate(&g_s));
    CK(hipEventCreate(&g_e0));
    CK(hipEventCreate(&g_e1));
    for (int k = 0; k < n_sets; ++k) {   // every stream its own allocation, as fyx_malloc_streams places them
        Set s;
        void* p[8];
        const size_t bytes[8] = {12, 12, 16, 16, 4, 12, 12, 16};
        for (int j = 0; j < 8; ++j) { CK(hipMalloc(&p[j], bytes[j] * g_n + 4096)); CK(hipMemset(p[j], j + 1, bytes[j] * g_n)); }
        s.pos = (float*)p[0]; s.nrm = (float*)p[1]; s.tan = (float*)p[2]; s.wgt = (float*)p[3]; s.idx = (uint32_t*)p[4];
        s.op = (float*)p[5]; s.on = (float*)p[6]; s.ot = (float*)p[7];
        g_sets.push_back(s);
    }
    CK(hipDeviceSynchronize());
    // aux: 0 default, 2 nt, 16 sc1
    PERSIST(256, 2, 2, 16, 4);      // the product's shape and policies
    PERSIST(256, 1, 2, 16, 4);
    PERSIST(256, 2, 2, 2, 4);
    PERSIST(256, 2, 0, 0, 4);
    PERSIST(256, 2, 2, 16, 2);
    PERSIST(256, 2, 2, 16, 8);
    PERSIST(512, 2, 2, 16, 2);
    PERSIST(512, 2, 2, 16, 4);
    PERSIST(1024, 2, 2, 16, 1);
    PERSIST(1024, 2, 2, 16, 2);
    PERSIST(1024, 1, 2, 16, 2);
    ONESHOT(64, 2, 16);
    ONESHOT(256, 2, 16);
    ONESHOT(256, 2, 2);
    ONESHOT(256, 0, 0);
    ONESHOT(1024, 2, 16);
    ONESHOTN(256, 2, 2, 16);
    ONESHOTN(256, 4, 2, 16);
    ONESHOTN(256, 4, 2, 2);
    ONESHOTN(512, 4, 2, 16);
    ONESHOTN(256, 8, 2, 16);
    return 0;
}
