// What one "level" of a dependent LDS chain costs a lone wave on gfx950: read (depends on the previous level's write) ->
// seven dependent VALU instructions -> write -> [nothing | s_barrier | __syncthreads].  One workgroup, 64 or 256 threads.
//   hipcc -O3 --offload-arch=gfx950 -o tools/exp/lds_level tools/exp/lds_level.hip && tools/exp/lds_level
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void chain(float* out, unsigned long long* t, int levels) {
    __shared__ float l[4096];
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += blockDim.x) l[i] = 1.0f + i * 1e-6f;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    const unsigned long long c0 = clock64();
    float y = 0.f;
    for (int lv = 0; lv < levels; ++lv) {
        const float* p = l + ((lv * 64 + (tid & ~15)) & 4095 & ~63) + (tid & 3);
        const float a0 = p[0], a1 = p[4], a2 = p[8], a3 = p[12];
        y = a0 * 1.0001f;
        y = a1 * 0.5f + y;
        y = a2 * 0.25f + y;
        y = a3 * 0.125f + y;
        l[(((lv + 1) * 64 + tid) & 4095)] = y;
        if (MODE == 1) __builtin_amdgcn_s_barrier();
        if (MODE == 2) __syncthreads();
        if (MODE == 3) { __builtin_amdgcn_s_waitcnt(0xc07f); }   // lgkmcnt(0) only
    }
    const unsigned long long c1 = clock64();
    const unsigned long long t1 = wall_clock64();
    if (tid == 0) { t[0] = t1 - t0; t[1] = c1 - c0; }
    out[tid] = y;
}

int main() {
    float* out; unsigned long long* t;
    hipMalloc(&out, 4096); hipMalloc(&t, 64);
    const char* names[4] = {"no barrier", "s_barrier", "__syncthreads", "s_waitcnt lgkmcnt(0)"};
    for (int block : {64, 256})
        for (int mode = 0; mode < 4; ++mode) {
            unsigned long long h[2] = {0, 0};
            for (int rep = 0; rep < 3; ++rep) {
                const int levels = 1000;
                if (mode == 0) hipLaunchKernelGGL(chain<0>, dim3(1), dim3(block), 0, 0, out, t, levels);
                if (mode == 1) hipLaunchKernelGGL(chain<1>, dim3(1), dim3(block), 0, 0, out, t, levels);
                if (mode == 2) hipLaunchKernelGGL(chain<2>, dim3(1), dim3(block), 0, 0, out, t, levels);
                if (mode == 3) hipLaunchKernelGGL(chain<3>, dim3(1), dim3(block), 0, 0, out, t, levels);
                hipDeviceSynchronize();
                hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
            }
            printf("{\"block\": %d, \"mode\": \"%s\", \"ns_per_level\": %.1f, \"cycles_per_level\": %.1f}\n", block, names[mode], h[0] * 10.0 / 1000, h[1] / 1000.0);
        }
    return 0;
}
