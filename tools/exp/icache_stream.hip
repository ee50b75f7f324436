// How fast a lone wave gets COLD code on gfx950: a straight line of N s_nop (4 bytes each) timed on its first pass (the
// instruction cache has never seen it) and on its second (it has), for several N; and with a sample of VALU code.
//   hipcc -O3 --offload-arch=gfx950 -o tools/exp/icache_stream tools/exp/icache_stream.hip && tools/exp/icache_stream
#include <hip/hip_runtime.h>
#include <cstdio>

#define BODY(N) asm volatile(".rept " #N "\n s_nop 0\n .endr" ::: "memory")
#define VBODY(N) asm volatile(".rept " #N "\n v_add_f32 %0, %0, %0\n .endr" : "+v"(y) :: "memory")

template <int KB>
__global__ void nops(unsigned long long* t) {
    for (int pass = 0; pass < 3; ++pass) {
        const unsigned long long c0 = clock64();
        if (KB == 4) BODY(1024);
        if (KB == 16) BODY(4096);
        if (KB == 32) BODY(8192);
        if (KB == 48) BODY(12288);
        const unsigned long long c1 = clock64();
        if (threadIdx.x == 0) t[pass] = c1 - c0;
    }
}
template <int KB>
__global__ void valu(unsigned long long* t, float* out) {
    float y = threadIdx.x;
    for (int pass = 0; pass < 3; ++pass) {
        const unsigned long long c0 = clock64();
        if (KB == 16) VBODY(4096);
        if (KB == 4) VBODY(1024);
        const unsigned long long c1 = clock64();
        if (threadIdx.x == 0) t[pass] = c1 - c0;
    }
    out[threadIdx.x] = y;
}

int main() {
    unsigned long long* t; float* out;
    hipMalloc(&t, 64); hipMalloc(&out, 1024);
    unsigned long long h[3];
#define RUN(K, KB, ...) for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(K<KB>, dim3(1), dim3(64), 0, 0, __VA_ARGS__); hipDeviceSynchronize(); hipMemcpy(h, t, 24, hipMemcpyDeviceToHost); \
        printf("{\"kernel\": \"%s\", \"code_KB\": %d, \"launch\": %d, \"cycles_pass\": [%llu, %llu, %llu], \"cold_cycles_per_64B_line\": %.1f}\n", #K, KB, rep, h[0], h[1], h[2], (double)h[0] / (KB * 16)); }
    RUN(nops, 4, t) RUN(nops, 16, t) RUN(nops, 32, t) RUN(nops, 48, t)
    RUN(valu, 4, t, out) RUN(valu, 16, t, out)
    return 0;
}
