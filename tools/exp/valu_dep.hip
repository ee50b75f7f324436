// Standalone probe (not part of the product): how many independent dependency chains a wave needs to keep a SIMD's VALU busy
// with packed-f32 / single-f32 mul+add, at 1 / 2 / 4 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 -o valu_dep valu_dep.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef float f32x2 __attribute__((ext_vector_type(2)));

// CH independent chains, each chain: alternately pk_mul and pk_add on its own accumulator (every instruction depends on the previous one of its chain)
template <int CH, bool PK>
__global__ __launch_bounds__(256) void k(float* out, int trips, float s) {
    f32x2 p[8]; float a[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; p[i] = f32x2{a[i], a[i] + 0.5f}; }
    f32x2 sp = {s, s * 1.0001f};
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int r = 0; r < 32 / CH; ++r) {     // 64 instructions per trip in total
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if constexpr (PK) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[c]) : "v"(sp));
                else asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[c]) : "v"(s));
            }
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if constexpr (PK) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[c]) : "v"(sp));
                else asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[c]) : "v"(s));
            }
        }
    }
    float r = 0;
    for (int i = 0; i < 8; ++i) r += a[i] + p[i].x + p[i].y;
    if (r == 12345.678f) out[0] = r;
}

template <int CH, bool PK>
void run(float* d) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int trips = 2000;
    for (int wps : {1, 2, 4, 6, 8}) {
        const int grid = 256 * wps;
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL((k<CH, PK>), dim3(grid), dim3(256), 0, 0, d, trips, 1.0000001f);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep >= 2 && ms < best) best = ms;
        }
        const double instr_per_simd = (double)trips * 64 * wps;
        printf("%s chains/wave %d waves/SIMD %d: %.2f ns per wave-instruction per SIMD; one wave issues every %.2f ns\n", PK ? "pk_mul/pk_add" : "mul/add      ", CH, wps,
               best * 1e6 / instr_per_simd, best * 1e6 / ((double)trips * 64));
    }
}

int main() {
    float* d; CK(hipMalloc(&d, 4096));
    run<1, true>(d); run<2, true>(d); run<4, true>(d); run<8, true>(d);
    run<1, false>(d); run<2, false>(d); run<4, false>(d); run<8, false>(d);
    return 0;
}
