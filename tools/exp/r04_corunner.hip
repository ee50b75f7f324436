// Round 4 experiment: synthetic co-runners for the crowd's skinning launch (tools/exp/r04_corunner.py).  What does a kernel on
// ANOTHER stream cost the exact crowd kernel, by what it does?
//   kind 0  occupy:  every wave sleeps (s_sleep, no memory, no VALU) until `us` microseconds have passed
//   kind 1  chase:   every lane follows `hops` dependent 16-byte loads through a 64 MB table (memory latency, like the sampler)
//   kind 2  scatter: every lane writes 16 bytes to `hops` scattered places (partial lines, like the sampler's record parts)
//   kind 3  valu:    every lane runs `hops` x 64 dependent v_fma
//   kind 4  records: like 2, but three adjacent lanes write one contiguous 48-byte record
// VGPR footprint: template R (registers pinned live across the body).
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/exp/libs/libcorunner.so tools/exp/r04_corunner.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int R>
__global__ __launch_bounds__(256) void corun(int kind, uint32_t us, uint32_t hops, const uint4* __restrict__ table, uint4* __restrict__ out, uint32_t mask) {
    float keep[R];
#pragma unroll
    for (int i = 0; i < R; ++i) keep[i] = (float)(threadIdx.x + i);
#pragma unroll
    for (int i = 0; i < R; ++i) asm volatile("" : "+v"(keep[i]));
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (kind == 0) {
        const uint64_t t0 = wall_clock64();
        while (wall_clock64() - t0 < (uint64_t)us * 100ull) __builtin_amdgcn_s_sleep(8);
    } else if (kind == 1) {
        uint32_t p = (gid * 2654435761u) & mask;
        for (uint32_t h = 0; h < hops; ++h) { const uint4 v = table[p]; p = (v.x + gid) & mask; }
        if (p == 0xffffffffu) out[0] = make_uint4(p, 0, 0, 0);
    } else if (kind == 2) {
        uint32_t p = (gid * 2654435761u) & mask;
        for (uint32_t h = 0; h < hops; ++h) { out[p] = make_uint4(gid, h, 0, 0); p = (p * 1664525u + 1013904223u) & mask; }
    } else if (kind == 4) {      // records: three adjacent lanes write the three 16-byte parts of one 48-byte record at a scattered place
        const uint32_t rec = gid / 3u, part = gid % 3u;
        uint32_t p = ((rec * 2654435761u) & mask) / 3u * 3u;
        for (uint32_t h = 0; h < hops; ++h) { out[p + part] = make_uint4(gid, h, 0, 0); p = ((p * 1664525u + 1013904223u) & mask) / 3u * 3u; }
    } else {
        float a = keep[0], b = 1.0001f;
        for (uint32_t h = 0; h < hops * 64u; ++h) a = __builtin_fmaf(a, b, 0.5f);
        keep[0] = a;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < R; ++i) { asm volatile("" : "+v"(keep[i])); s += keep[i]; }
    if (s == 123.456f) out[1] = make_uint4(1, 2, 3, 4);
}

static hipStream_t g_stream = nullptr;
static uint4 *g_table = nullptr, *g_out = nullptr;
static const uint32_t kEntries = 1u << 22;   // 4 M x 16 B = 64 MB each
static hipEvent_t g_e0 = nullptr, g_e1 = nullptr;

extern "C" int corun_init() {
    if (hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking) != hipSuccess) return 1;
    if (hipMalloc(&g_table, (size_t)kEntries * 16) != hipSuccess || hipMalloc(&g_out, (size_t)kEntries * 16) != hipSuccess) return 2;
    uint4* h = (uint4*)malloc((size_t)kEntries * 16);
    uint32_t x = 12345u;
    for (uint32_t i = 0; i < kEntries; ++i) { x = x * 1664525u + 1013904223u; h[i] = make_uint4(x >> 7, i, 0, 0); }
    hipMemcpy(g_table, h, (size_t)kEntries * 16, hipMemcpyHostToDevice);
    free(h);
    hipEventCreate(&g_e0); hipEventCreate(&g_e1);
    return 0;
}
// launches the co-runner on its own stream; regs: 32 / 64 / 128 / 160
extern "C" int corun_launch(int kind, int regs, uint32_t grid, uint32_t us, uint32_t hops) {
    const uint32_t mask = kEntries - 1;
    hipEventRecord(g_e0, g_stream);
    if (regs <= 32) hipLaunchKernelGGL(corun<24>, dim3(grid), dim3(256), 0, g_stream, kind, us, hops, g_table, g_out, mask);
    else if (regs <= 64) hipLaunchKernelGGL(corun<48>, dim3(grid), dim3(256), 0, g_stream, kind, us, hops, g_table, g_out, mask);
    else if (regs <= 128) hipLaunchKernelGGL(corun<112>, dim3(grid), dim3(256), 0, g_stream, kind, us, hops, g_table, g_out, mask);
    else hipLaunchKernelGGL(corun<150>, dim3(grid), dim3(256), 0, g_stream, kind, us, hops, g_table, g_out, mask);
    hipEventRecord(g_e1, g_stream);
    return (int)hipGetLastError();
}
extern "C" float corun_last_ms() { float ms = 0.f; hipEventSynchronize(g_e1); hipEventElapsedTime(&ms, g_e0, g_e1); return ms; }
extern "C" int corun_sync() { return (int)hipStreamSynchronize(g_stream); }
