// Standalone probe (not part of the product): issue rate of the f32 VALU forms the skinning arithmetic can be written in,
// per SIMD, at 1 / 2 / 4 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef float f32x2 __attribute__((ext_vector_type(2)));

// 16 independent instructions per asm block, ITER blocks per loop trip
#define REP16(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int trips, float s) {
    float a[8]; f32x2 p[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; p[i] = f32x2{a[i], a[i] + 0.5f}; }
    f32x2 sp = {s, s * 1.0001f};
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if constexpr (KIND == 0) {
#define OP(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
                REP16(OP)
#undef OP
            } else if constexpr (KIND == 1) {
#define OP(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
                REP16(OP)
#undef OP
            } else if constexpr (KIND == 2) {
#define OP(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(s));
                REP16(OP)
#undef OP
            } else if constexpr (KIND == 3) {
#define OP(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(sp));
                REP16(OP)
#undef OP
            } else if constexpr (KIND == 4) {
#define OP(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(sp));
                REP16(OP)
#undef OP
            } else if constexpr (KIND == 5) {
#define OP(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(sp));
                REP16(OP)
#undef OP
            } else if constexpr (KIND == 6) {   // broadcast of the low half of the multiplier (op_sel_hi:[1,0])
#define OP(i) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(p[i]) : "v"(sp));
                REP16(OP)
#undef OP
            } else if constexpr (KIND == 7) {   // alternating mul / add on scalars (the exact chain's mix)
#define OP(i) asm volatile("v_mul_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %2" : "+v"(a[i]), "+v"(a[(i + 4) & 7]) : "v"(s));
                REP16(OP)
#undef OP
            } else if constexpr (KIND == 8) {   // VOP3 mul with an SGPR operand
#define OP(i) asm volatile("v_mul_f32_e64 %0, %0, %1" : "+v"(a[i]) : "s"(s));
                REP16(OP)
#undef OP
            }
        }
    }
    float r = 0;
    for (int i = 0; i < 8; ++i) r += a[i] + p[i].x + p[i].y;
    if (r == 12345.678f) out[0] = r;
}

template <int KIND>
void run(const char* name, float* d, int per_instr_mult = 1) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int trips = 2000;
    for (int wps : {1, 2, 4, 8}) {            // waves per SIMD: 256-thread blocks, wps blocks per CU
        const int grid = 256 * wps;
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL((k<KIND>), dim3(grid), dim3(256), 0, 0, d, trips, 1.0000001f);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep >= 2 && ms < best) best = ms;
        }
        const double instr_per_simd = (double)trips * 64 * per_instr_mult * wps;   // each SIMD hosts wps waves
        printf("%-34s waves/SIMD %d: %8.1f us  -> %.2f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz)\n", name, wps,
               best * 1e3, best * 1e6 / instr_per_simd, best * 1e6 / instr_per_simd * 2.4);
    }
}

int main() {
    float* d; CK(hipMalloc(&d, 4096));
    run<0>("v_mul_f32", d); run<1>("v_add_f32", d); run<2>("v_fma_f32", d);
    run<3>("v_pk_mul_f32", d); run<4>("v_pk_add_f32", d); run<5>("v_pk_fma_f32", d);
    run<6>("v_pk_mul_f32 op_sel_hi:[1,0]", d); run<7>("v_mul_f32 + v_add_f32 pairs", d, 2); run<8>("v_mul_f32_e64 (SGPR operand)", d);
    return 0;
}
