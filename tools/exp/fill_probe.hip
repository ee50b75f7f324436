// Standalone probe: which form of a plain 400 MB fill reaches the rate of hipMemsetAsync (6.8 TB/s)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(void* p, uint32_t bytes) { return __builtin_amdgcn_make_buffer_rsrc(p, 0, bytes, 0x00020000); }

// POL: 0 plain, 1 nt builtin, 16 sc1 (buffer), 17 sc0sc1, 2 nt (buffer)   PER: float4 per lane per iteration, contiguous per lane (PER x 16 B)
template <int POL, int PER, bool STRIDE>
__global__ __launch_bounds__(256) void k_fill(f32x4* dst, size_t n4) {
    const f32x4 v = {1, 2, 3, 4};
    if (STRIDE) {
        for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * PER; i < n4; i += (size_t)gridDim.x * 256 * PER)
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                if (POL == 0) dst[i + j] = v; else __builtin_nontemporal_store(v, dst + i + j);
            }
    } else {   // one chunk per workgroup
        const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * PER;
        if (i < n4) {
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                if (POL == 0) dst[i + j] = v; else __builtin_nontemporal_store(v, dst + i + j);
            }
        }
    }
}
// lanes interleaved: iteration j writes lane's 16 B at (base + j * 256 lanes): every instruction a dense 4 KB per workgroup
template <int POL, int PER>
__global__ __launch_bounds__(256) void k_fill_dense(f32x4* dst, size_t n4) {
    const f32x4 v = {1, 2, 3, 4};
    const size_t base = (size_t)blockIdx.x * 256 * PER + threadIdx.x;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const size_t i = base + (size_t)j * 256;
        if (i < n4) { if (POL == 0) dst[i] = v; else __builtin_nontemporal_store(v, dst + i); }
    }
}
int main() {
    const size_t bytes = 400u * 1000 * 1000, n4 = bytes / 16;
    f32x4* d; CK(hipMalloc(&d, bytes + 4096));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto launch) {
        float best = 1e9f, sum = 0; int n = 0;
        for (int rep = 0; rep < 14; ++rep) {
            CK(hipEventRecord(e0, 0)); launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep >= 4) { best = std::min(best, ms); sum += ms; ++n; }
        }
        printf("%-52s avg %.2f us  best %.2f us  (%.2f TB/s)\n", name, sum / n * 1e3f, best * 1e3f, bytes / (sum / n * 1e-3) / 1e12);
    };
    timeit("hipMemsetAsync", [&] { CK(hipMemsetAsync(d, 1, bytes, 0)); });
    timeit("grid-stride 2048x256 nt PER1", [&] { hipLaunchKernelGGL((k_fill<1, 1, true>), dim3(2048), dim3(256), 0, 0, d, n4); });
    timeit("grid-stride 2048x256 plain PER1", [&] { hipLaunchKernelGGL((k_fill<0, 1, true>), dim3(2048), dim3(256), 0, 0, d, n4); });
    timeit("grid-stride 4096x256 plain PER1", [&] { hipLaunchKernelGGL((k_fill<0, 1, true>), dim3(4096), dim3(256), 0, 0, d, n4); });
    timeit("grid-stride 2048x256 plain PER4 (64 B/lane)", [&] { hipLaunchKernelGGL((k_fill<0, 4, true>), dim3(2048), dim3(256), 0, 0, d, n4); });
    timeit("one chunk per WG plain PER1", [&] { hipLaunchKernelGGL((k_fill<0, 1, false>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, d, n4); });
    timeit("one chunk per WG nt PER1", [&] { hipLaunchKernelGGL((k_fill<1, 1, false>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, d, n4); });
    timeit("one chunk per WG plain PER4 (64 B/lane)", [&] { hipLaunchKernelGGL((k_fill<0, 4, false>), dim3((unsigned)((n4 + 1023) / 1024)), dim3(256), 0, 0, d, n4); });
    timeit("dense 4 x 4 KB per WG plain", [&] { hipLaunchKernelGGL((k_fill_dense<0, 4>), dim3((unsigned)((n4 + 1023) / 1024)), dim3(256), 0, 0, d, n4); });
    timeit("dense 4 x 4 KB per WG nt", [&] { hipLaunchKernelGGL((k_fill_dense<1, 4>), dim3((unsigned)((n4 + 1023) / 1024)), dim3(256), 0, 0, d, n4); });
    timeit("dense 16 x 4 KB per WG plain", [&] { hipLaunchKernelGGL((k_fill_dense<0, 16>), dim3((unsigned)((n4 + 4095) / 4096)), dim3(256), 0, 0, d, n4); });
    timeit("dense 16 x 4 KB per WG nt", [&] { hipLaunchKernelGGL((k_fill_dense<1, 16>), dim3((unsigned)((n4 + 4095) / 4096)), dim3(256), 0, 0, d, n4); });
    return 0;
}
