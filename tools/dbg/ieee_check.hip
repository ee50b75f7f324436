#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k(const float* a, const float* b, float* s, float* d, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { s[i] = sqrtf(a[i]); d[i] = b[i] / s[i]; }
}
int main() {
    const int n = 1 << 20;
    std::vector<float> a(n), b(n), s(n), d(n);
    srand(1);
    for (int i = 0; i < n; ++i) { a[i] = 0.5f + (float)rand() / RAND_MAX; b[i] = (float)rand() / RAND_MAX - 0.5f; }
    float *da, *db, *ds, *dd;
    hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&dd, n * 4);
    hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(da, db, ds, dd, n);
    hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost); hipMemcpy(d.data(), dd, n * 4, hipMemcpyDeviceToHost);
    int bs = 0, bd = 0;
    for (int i = 0; i < n; ++i) {
        volatile float hs = sqrtf(a[i]);
        volatile float hd = b[i] / s[i];
        if (hs != s[i]) ++bs;
        if (hd != d[i]) ++bd;
    }
    printf("sqrt mismatches %d, div mismatches %d of %d\n", bs, bd, n);
    return 0;
}
