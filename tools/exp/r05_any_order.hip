// Round 5: do two kernels of ONE stream overlap when the second is launched with hipExtAnyOrderLaunch (AQL barrier bit clear)?
// K1: a bandwidth-bound copy (~tens of us).  K2: a latency-bound pointer chase on a few thousand waves.  Prints one JSON line.
//   hipcc --offload-arch=gfx950 -O2 -o tools/exp/_build/r05_any_order tools/exp/r05_any_order.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(512) void copy_kernel(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}

__global__ __launch_bounds__(256) void chase_kernel(const unsigned* __restrict__ next, unsigned* __restrict__ out, unsigned steps, unsigned mask) {
    unsigned i = (blockIdx.x * blockDim.x + threadIdx.x) & mask;
    for (unsigned s = 0; s < steps; ++s) i = next[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = i;
}

int main() {
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const size_t n = (size_t)6 << 20;      // 96 MB in, 96 MB out
    uint4 *a, *b;
    CK(hipMalloc(&a, n * 16));
    CK(hipMalloc(&b, n * 16));
    CK(hipMemset(a, 1, n * 16));
    const unsigned m = 1u << 22, threads = 4096 * 256;
    std::vector<unsigned> h(m);
    for (unsigned i = 0; i < m; ++i) h[i] = (unsigned)(((unsigned long long)i * 2654435761ull + 12345) & (m - 1));
    unsigned *next, *out;
    CK(hipMalloc(&next, m * 4));
    CK(hipMalloc(&out, threads * 4));
    CK(hipMemcpy(next, h.data(), m * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int reps = 300;
    auto timed = [&](int mode, float* us) -> int {      // 0: K1 only, 1: K2 only, 2: K1 then K2 (barrier), 3: K1 then K2 any-order
        for (int w = 0; w < 2; ++w) {
            if (w) CK(hipEventRecord(e0, s));
            for (int k = 0; k < (w ? reps : 20); ++k) {
                if (mode != 1) hipLaunchKernelGGL(copy_kernel, dim3(1024), dim3(512), 0, s, a, b, n);
                if (mode == 1 || mode == 2) hipLaunchKernelGGL(chase_kernel, dim3(4096), dim3(256), 0, s, next, out, 12u, m - 1);
                if (mode == 3) hipExtLaunchKernelGGL(chase_kernel, dim3(4096), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, next, out, 12u, m - 1);
            }
        }
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        *us = ms * 1e3f / reps;
        return 0;
    };
    float t[4];
    for (int mode = 0; mode < 4; ++mode)
        if (timed(mode, &t[mode])) return 1;
    printf("{\"copy_us\": %.2f, \"chase_us\": %.2f, \"copy_then_chase_us\": %.2f, \"copy_then_chase_any_order_us\": %.2f}\n", t[0], t[1], t[2], t[3]);
    return 0;
}
