// Standalone probe (not part of the product): what does rocprofv3's FETCH_SIZE report, and what does the memory system fetch, when a
// kernel reads 64 contiguous bytes out of every 128-byte line (the pose sampler's key pairs) instead of streaming whole lines?
//   hipcc --offload-arch=gfx950 -O3 -o fetch_granule fetch_granule.hip ;  rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- ./fetch_granule
// Kernels (each over a 1 GiB buffer, far beyond the 256 MiB Infinity Cache, every line touched at most once):
//   k_dense      every lane reads 16 B, lanes contiguous: 1 GiB read                       (the guide's calibration case)
//   k_half_lines four lanes read the FIRST 64 B of every 128-B line: 512 MiB requested
//   k_half_odd   four lanes read the SECOND 64 B of every 128-B line: 512 MiB requested
//   k_quarter    two lanes read 32 B of every 128-B line: 256 MiB requested
//   k_half_256   four lanes read 64 B of every 256 B: 256 MiB requested, every other line untouched
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CHUNK_LANES, int STRIDE_B, int OFF_B>
__global__ __launch_bounds__(256) void k_sparse(const char* __restrict__ src, size_t n_chunks, float* out) {
    // chunk c = CHUNK_LANES consecutive lanes reading 16 B each at src + c * STRIDE_B + OFF_B
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t c = t / CHUNK_LANES, l = t % CHUNK_LANES;
    if (c >= n_chunks) return;
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + c * STRIDE_B + OFF_B + l * 16);
    if (v.x == 12345.678f && v.y == -1.0f) out[0] = v.z;
}

int main() {
    const size_t bytes = (size_t)1 << 30;
    char* src; float* out;
    CK(hipMalloc(&src, bytes)); CK(hipMalloc(&out, 64));
    CK(hipMemset(src, 0, bytes));
    CK(hipDeviceSynchronize());
    auto run = [&](const char* name, auto kernel, int chunk_lanes, int stride) {
        const size_t n_chunks = bytes / stride, threads = n_chunks * chunk_lanes;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, 0, (const char*)src, n_chunks, out);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("%s: %zu bytes requested per launch, %zu lines touched, %.1f us -> %.2f TB/s requested, %.2f TB/s if whole lines move\n", name,
               n_chunks * chunk_lanes * 16, bytes / 128 / (stride / 128), best * 1e3, n_chunks * chunk_lanes * 16 / (best * 1e-3) / 1e12,
               (double)(bytes / (stride / 128)) / (best * 1e-3) / 1e12);
    };
    run("k_dense", k_sparse<8, 128, 0>, 8, 128);
    run("k_half_lines", k_sparse<4, 128, 0>, 4, 128);
    run("k_half_odd", k_sparse<4, 128, 64>, 4, 128);
    run("k_quarter", k_sparse<2, 128, 0>, 2, 128);
    run("k_half_256", k_sparse<4, 256, 0>, 4, 256);
    return 0;
}
