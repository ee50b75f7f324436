// What a branch costs a lone wave on gfx950: 1024 taken s_branch (each to the next instruction), 1024 not-taken
// s_cbranch_scc1, 1024 taken s_cbranch_scc0, 1024 s_cbranch_execz not taken, and the same with a second wave beside it.
//   hipcc -O3 --offload-arch=gfx950 -o tools/exp/branch_cost tools/exp/branch_cost.hip && tools/exp/branch_cost
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void k(unsigned long long* t) {
    for (int pass = 0; pass < 3; ++pass) {
        const unsigned long long c0 = clock64();
        if (MODE == 0) asm volatile(".rept 1024\n s_branch 0\n .endr" ::: "memory");
        if (MODE == 1) asm volatile("s_cmp_eq_u32 0, 1\n .rept 1024\n s_cbranch_scc1 0\n .endr" ::: "memory", "scc");
        if (MODE == 2) asm volatile("s_cmp_eq_u32 0, 1\n .rept 1024\n s_cbranch_scc0 0\n .endr" ::: "memory", "scc");
        if (MODE == 3) asm volatile(".rept 1024\n s_cbranch_execz 0\n .endr" ::: "memory");
        if (MODE == 4) asm volatile(".rept 1024\n s_branch 15\n .rept 15\n s_nop 0\n .endr\n .endr" ::: "memory");   // each hop lands in the next 64-byte line
        if (MODE == 5) asm volatile(".rept 1024\n s_nop 0\n .endr" ::: "memory");
        const unsigned long long c1 = clock64();
        if (threadIdx.x == 0 && blockIdx.x == 0) t[pass] = c1 - c0;
    }
}

int main() {
    unsigned long long* t;
    hipMalloc(&t, 64);
    unsigned long long h[3];
    const char* names[6] = {"s_branch taken (next instruction)", "s_cbranch_scc1 not taken", "s_cbranch_scc0 taken", "s_cbranch_execz not taken", "s_branch taken (next 64-byte line)", "s_nop"};
#define RUN(M) for (int block : {64, 128}) { hipLaunchKernelGGL(k<M>, dim3(1), dim3(block), 0, 0, t); hipLaunchKernelGGL(k<M>, dim3(1), dim3(block), 0, 0, t); hipDeviceSynchronize(); hipMemcpy(h, t, 24, hipMemcpyDeviceToHost); \
        printf("{\"what\": \"%s\", \"threads\": %d, \"cycles_each\": [%.1f, %.1f, %.1f]}\n", names[M], block, h[0] / 1024.0, h[1] / 1024.0, h[2] / 1024.0); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5)
    return 0;
}
