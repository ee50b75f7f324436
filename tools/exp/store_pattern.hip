// Standalone probe (not part of the product): how fast can 400 MB of skinned output be WRITTEN in the crowd kernel's
// pattern (three streams [instance][vertex] of 12 / 12 / 16 bytes, a workgroup = one 512-vertex tile x a run of
// instances), against a linear fill?   hipcc --offload-arch=gfx950 -O3 -o store_pattern store_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x3 __attribute__((ext_vector_type(3)));

template <bool NT, typename T> __device__ __forceinline__ void st(T* p, T v) { if constexpr (NT) __builtin_nontemporal_store(v, p); else *p = v; }

__global__ __launch_bounds__(256) void k_fill(f32x4* dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) __builtin_nontemporal_store(f32x4{1, 2, 3, 4}, dst + i);
}

// MODE 0: tile fastest over blockIdx (product before the XCD ranking); 1: XCD-ranked; 2: chunk fastest
// VPL: vertices per lane (1 or 2: a tile of 512 or 1024 vertices)
template <bool NT, int MODE, int VPL>
__global__ __launch_bounds__(512) void k_crowd(float* op, float* on, float* ot, uint32_t n_verts, uint32_t n_inst, uint32_t tiles, uint32_t ipb) {
    uint32_t tile, chunk;
    const uint32_t chunks = (n_inst + ipb - 1) / ipb;
    if (MODE == 0) { tile = blockIdx.x % tiles; chunk = blockIdx.x / tiles; }
    else if (MODE == 1) {
        const uint32_t G = gridDim.x, x = blockIdx.x & 7u;
        uint32_t rank = blockIdx.x >> 3;
        for (uint32_t q = 0; q < x; ++q) rank += (G - q + 7u) >> 3;
        tile = rank % tiles; chunk = rank / tiles;
    } else { chunk = blockIdx.x % chunks; tile = blockIdx.x / chunks; }
    uint32_t i0 = chunk * ipb, i1 = min(i0 + ipb, n_inst), istep = 1;
    if (MODE == 3) {   // instances interleaved over the `chunks` workgroups of a tile: all resident workgroups write one narrow band
        tile = blockIdx.x % tiles; chunk = blockIdx.x / tiles;
        i0 = chunk; i1 = n_inst; istep = chunks;
    }
    for (uint32_t inst = i0; inst < i1; inst += istep) {
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            const uint32_t v = tile * (512 * VPL) + j * 512 + threadIdx.x;
            if (v >= n_verts) continue;
            const size_t ov = (size_t)inst * n_verts + v;
            const float f = (float)inst;
            st<NT>(reinterpret_cast<f32x3*>(op + ov * 3), f32x3{f, 1, 2});
            st<NT>(reinterpret_cast<f32x3*>(on + ov * 3), f32x3{f, 3, 4});
            st<NT>(reinterpret_cast<f32x4*>(ot + ov * 4), f32x4{f, 5, 6, 7});
        }
    }
}

// one (tile, instance) per workgroup; WHAT: 1 = pos only (x3), 2 = tan only (x4), 3 = pos as x4 (48 lanes of a wave cover its 768 B),
// 4 = all three with pos / nrm as x4 over 48 lanes
template <int WHAT>
__global__ __launch_bounds__(512) void k_one(float* op, float* on, float* ot, uint32_t n_verts, uint32_t tiles) {
    const uint32_t tile = blockIdx.x % tiles, inst = blockIdx.x / tiles;
    const uint32_t v = tile * 512 + threadIdx.x, lane = threadIdx.x & 63u, wv0 = tile * 512 + (threadIdx.x & ~63u);
    const float f = (float)inst;
    const size_t ov = (size_t)inst * n_verts + v;
    if (WHAT == 1 && v < n_verts) __builtin_nontemporal_store(f32x3{f, 1, 2}, reinterpret_cast<f32x3*>(op + ov * 3));
    if ((WHAT == 2 || WHAT == 4) && v < n_verts) __builtin_nontemporal_store(f32x4{f, 5, 6, 7}, reinterpret_cast<f32x4*>(ot + ov * 4));
    if (WHAT == 3 || WHAT == 4) {   // the wave's 64 x 12 B = 48 x 16 B
        const size_t wbase = ((size_t)inst * n_verts + wv0) * 3;     // in floats
        if (lane < 48 && wv0 + 64 <= n_verts) {
            __builtin_nontemporal_store(f32x4{f, 1, 2, 3}, reinterpret_cast<f32x4*>(op + wbase) + lane);
            if (WHAT == 4) __builtin_nontemporal_store(f32x4{f, 1, 2, 3}, reinterpret_cast<f32x4*>(on + wbase) + lane);
        }
    }
}

int main(int argc, char** argv) {
    const uint32_t n_verts = argc > 1 ? atoi(argv[1]) : 10000, n_inst = 1000;
    const size_t nv = (size_t)n_verts * n_inst;
    float *op, *on, *ot;
    CK(hipMalloc(&op, nv * 12 + 256)); CK(hipMalloc(&on, nv * 12 + 256)); CK(hipMalloc(&ot, nv * 16 + 256));
    f32x4* lin; CK(hipMalloc(&lin, nv * 40));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto launch) {
        float best = 1e9f, sum = 0; int n = 0;
        for (int rep = 0; rep < 14; ++rep) {
            CK(hipEventRecord(e0, 0)); launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep >= 4) { best = std::min(best, ms); sum += ms; ++n; }
        }
        printf("%-44s verts %u: avg %.2f us  best %.2f us  (%.2f TB/s)\n", name, n_verts, sum / n * 1e3f, best * 1e3f, nv * 40.0 / (sum / n * 1e-3) / 1e12);
    };
    timeit("linear fill", [&] { hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, lin, nv * 40 / 16); });
    {
        const uint32_t tiles = (n_verts + 511) / 512;
        const double b1 = nv * 12.0, b2 = nv * 16.0;
        auto t2 = [&](const char* name, double bytes, auto launch) {
            float sum = 0; int n = 0;
            for (int rep = 0; rep < 14; ++rep) {
                CK(hipEventRecord(e0, 0)); launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep >= 4) { sum += ms; ++n; }
            }
            printf("%-44s verts %u: avg %.2f us  (%.2f TB/s)\n", name, n_verts, sum / n * 1e3f, bytes / (sum / n * 1e-3) / 1e12);
        };
        t2("one-shot: pos only, x3 per lane", b1, [&] { hipLaunchKernelGGL((k_one<1>), dim3(tiles * n_inst), dim3(512), 0, 0, op, on, ot, n_verts, tiles); });
        t2("one-shot: tan only, x4 per lane", b2, [&] { hipLaunchKernelGGL((k_one<2>), dim3(tiles * n_inst), dim3(512), 0, 0, op, on, ot, n_verts, tiles); });
        t2("one-shot: pos only, as x4 over 48 lanes", b1, [&] { hipLaunchKernelGGL((k_one<3>), dim3(tiles * n_inst), dim3(512), 0, 0, op, on, ot, n_verts, tiles); });
        t2("one-shot: all three, pos / nrm as x4 over 48 lanes", nv * 40.0, [&] { hipLaunchKernelGGL((k_one<4>), dim3(tiles * n_inst), dim3(512), 0, 0, op, on, ot, n_verts, tiles); });
    }
    {   // one-shot workgroups: one (tile, instance) each, instance-major order
        const uint32_t tiles = (n_verts + 511) / 512;
        timeit("crowd nt ONE instance per WG (tile fastest)", [&] { hipLaunchKernelGGL((k_crowd<true, 0, 1>), dim3(tiles * n_inst), dim3(512), 0, 0, op, on, ot, n_verts, n_inst, tiles, 1u); });
        timeit("crowd plain ONE instance per WG (tile fastest)", [&] { hipLaunchKernelGGL((k_crowd<false, 0, 1>), dim3(tiles * n_inst), dim3(512), 0, 0, op, on, ot, n_verts, n_inst, tiles, 1u); });
        timeit("crowd nt TWO instances per WG", [&] { hipLaunchKernelGGL((k_crowd<true, 0, 1>), dim3(tiles * ((n_inst + 1) / 2)), dim3(512), 0, 0, op, on, ot, n_verts, n_inst, tiles, 2u); });
        timeit("crowd nt FOUR instances per WG", [&] { hipLaunchKernelGGL((k_crowd<true, 0, 1>), dim3(tiles * ((n_inst + 3) / 4)), dim3(512), 0, 0, op, on, ot, n_verts, n_inst, tiles, 4u); });
    }
    for (uint32_t ipb : {16u, 8u}) {
        const uint32_t tiles = (n_verts + 511) / 512, chunks = (n_inst + ipb - 1) / ipb, tiles2 = (n_verts + 1023) / 1024;
        char nm[96];
        snprintf(nm, 96, "crowd nt tile-fastest ipb %u", ipb);
        timeit(nm, [&] { hipLaunchKernelGGL((k_crowd<true, 0, 1>), dim3(tiles * chunks), dim3(512), 0, 0, op, on, ot, n_verts, n_inst, tiles, ipb); });
        snprintf(nm, 96, "crowd nt xcd-ranked ipb %u", ipb);
        timeit(nm, [&] { hipLaunchKernelGGL((k_crowd<true, 1, 1>), dim3(tiles * chunks), dim3(512), 0, 0, op, on, ot, n_verts, n_inst, tiles, ipb); });
        snprintf(nm, 96, "crowd nt chunk-fastest ipb %u", ipb);
        timeit(nm, [&] { hipLaunchKernelGGL((k_crowd<true, 2, 1>), dim3(tiles * chunks), dim3(512), 0, 0, op, on, ot, n_verts, n_inst, tiles, ipb); });
        snprintf(nm, 96, "crowd nt interleaved instances, %u per WG", ipb);
        timeit(nm, [&] { hipLaunchKernelGGL((k_crowd<true, 3, 1>), dim3(tiles * chunks), dim3(512), 0, 0, op, on, ot, n_verts, n_inst, tiles, ipb); });
        snprintf(nm, 96, "crowd plain xcd-ranked ipb %u", ipb);
        timeit(nm, [&] { hipLaunchKernelGGL((k_crowd<false, 1, 1>), dim3(tiles * chunks), dim3(512), 0, 0, op, on, ot, n_verts, n_inst, tiles, ipb); });
        snprintf(nm, 96, "crowd nt xcd-ranked 2 verts/lane ipb %u", ipb);
        timeit(nm, [&] { hipLaunchKernelGGL((k_crowd<true, 1, 2>), dim3(tiles2 * chunks), dim3(512), 0, 0, op, on, ot, n_verts, n_inst, tiles2, ipb); });
    }
    return 0;
}
