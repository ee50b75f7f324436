// Standalone ablation harness (not part of the product): which part of the skinning kernel costs
// the ~2 us it sits above the pure-stream ceiling?  hipcc --offload-arch=gfx950 -O3 -o stream_exp stream_exp.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <string>
#include <chrono>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

constexpr int NV = 1000000, NB = 256, SETS = 8;

struct Bufs { float *pos, *nrm, *tan, *wgt; uint32_t* idx; float *opos, *onrm, *otan; };

template <bool NT, typename T> __device__ __forceinline__ T ldg(const T* p) { if constexpr (NT) return __builtin_nontemporal_load(p); else return *p; }
template <bool NT, typename T> __device__ __forceinline__ void stg(T* p, T v) { if constexpr (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// MODE 0: passthrough streams only; 1: + palette staging; 2: + LDS gather (sum rows); 3: full math (fused); 4: full math unfused
template <int BLOCK, int MODE, bool NT>
__global__ __launch_bounds__(BLOCK) void k_lbs(Bufs b, const float* pal, uint32_t n_verts, uint32_t total_units) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f32x4* rows = reinterpret_cast<f32x4*>(smem);
    constexpr uint32_t WPB = BLOCK / 64;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t u_begin = (uint32_t)(((uint64_t)blockIdx.x * total_units) / gridDim.x);
    const uint32_t u_end = (uint32_t)(((uint64_t)(blockIdx.x + 1) * total_units) / gridDim.x);
    bool staged = false;
    for (uint32_t u = u_begin + wave; u < u_end || !staged; u += WPB) {
        const bool active = u < u_end;
        uint32_t v = u * 64 + lane;
        const bool live = active && v < n_verts;
        const uint32_t vs = live ? v : 0;
        float px = ldg<NT>(b.pos + (size_t)vs * 3), py = ldg<NT>(b.pos + (size_t)vs * 3 + 1), pz = ldg<NT>(b.pos + (size_t)vs * 3 + 2);
        float nx = ldg<NT>(b.nrm + (size_t)vs * 3), ny = ldg<NT>(b.nrm + (size_t)vs * 3 + 1), nz = ldg<NT>(b.nrm + (size_t)vs * 3 + 2);
        f32x4 t = ldg<NT>(reinterpret_cast<const f32x4*>(b.tan) + vs);
        f32x4 w = ldg<NT>(reinterpret_cast<const f32x4*>(b.wgt) + vs);
        uint32_t id = ldg<NT>(b.idx + vs);
        if (!staged) {
            if constexpr (MODE >= 1) {
                for (uint32_t bn = threadIdx.x; bn < NB; bn += BLOCK) {
                    const f32x4* m = reinterpret_cast<const f32x4*>(pal + (size_t)bn * 16);
                    f32x4 c0 = m[0], c1 = m[1], c2 = m[2], c3 = m[3];
                    rows[bn * 3 + 0] = f32x4{c0.x, c1.x, c2.x, c3.x};
                    rows[bn * 3 + 1] = f32x4{c0.y, c1.y, c2.y, c3.y};
                    rows[bn * 3 + 2] = f32x4{c0.z, c1.z, c2.z, c3.z};
                }
                __syncthreads();
            }
            staged = true;
        }
        float ox = px + w.x, oy = py + w.y, oz = pz + w.z, onx = nx + w.w, ony = ny, onz = nz + __uint_as_float(id) * 0.f;
        float otx = t.x, oty = t.y, otz = t.z;
        if constexpr (MODE == 2) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t bn = (id >> (8 * k)) & 0xff;
                f32x4 r0 = rows[bn * 3], r1 = rows[bn * 3 + 1], r2 = rows[bn * 3 + 2];
                ox += r0.x + r0.y + r0.z + r0.w; oy += r1.x + r1.y + r1.z + r1.w; oz += r2.x + r2.y + r2.z + r2.w;
            }
        }
        if constexpr (MODE >= 3) {
            ox = oy = oz = onx = ony = onz = otx = oty = otz = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t bn = (id >> (8 * k)) & 0xff;
                const float wk = w[k];
                f32x4 r0 = rows[bn * 3], r1 = rows[bn * 3 + 1], r2 = rows[bn * 3 + 2];
                if constexpr (MODE == 3) {
#define DOT(r, a_, b_, c_) __builtin_fmaf(r.z, c_, __builtin_fmaf(r.y, b_, r.x * a_))
                    ox = __builtin_fmaf(__builtin_fmaf(r0.z, pz, __builtin_fmaf(r0.y, py, __builtin_fmaf(r0.x, px, r0.w))), wk, ox);
                    oy = __builtin_fmaf(__builtin_fmaf(r1.z, pz, __builtin_fmaf(r1.y, py, __builtin_fmaf(r1.x, px, r1.w))), wk, oy);
                    oz = __builtin_fmaf(__builtin_fmaf(r2.z, pz, __builtin_fmaf(r2.y, py, __builtin_fmaf(r2.x, px, r2.w))), wk, oz);
                    onx = __builtin_fmaf(DOT(r0, nx, ny, nz), wk, onx); ony = __builtin_fmaf(DOT(r1, nx, ny, nz), wk, ony); onz = __builtin_fmaf(DOT(r2, nx, ny, nz), wk, onz);
                    otx = __builtin_fmaf(DOT(r0, t.x, t.y, t.z), wk, otx); oty = __builtin_fmaf(DOT(r1, t.x, t.y, t.z), wk, oty); otz = __builtin_fmaf(DOT(r2, t.x, t.y, t.z), wk, otz);
                } else {
#define DOTX(r, a_, b_, c_) ((r.x * a_ + r.y * b_) + r.z * c_)
                    ox = ox + (DOTX(r0, px, py, pz) + r0.w) * wk; oy = oy + (DOTX(r1, px, py, pz) + r1.w) * wk; oz = oz + (DOTX(r2, px, py, pz) + r2.w) * wk;
                    onx = onx + DOTX(r0, nx, ny, nz) * wk; ony = ony + DOTX(r1, nx, ny, nz) * wk; onz = onz + DOTX(r2, nx, ny, nz) * wk;
                    otx = otx + DOTX(r0, t.x, t.y, t.z) * wk; oty = oty + DOTX(r1, t.x, t.y, t.z) * wk; otz = otz + DOTX(r2, t.x, t.y, t.z) * wk;
                }
            }
        }
        if (live) {
            stg<NT>(b.opos + (size_t)v * 3, ox); stg<NT>(b.opos + (size_t)v * 3 + 1, oy); stg<NT>(b.opos + (size_t)v * 3 + 2, oz);
            stg<NT>(b.onrm + (size_t)v * 3, onx); stg<NT>(b.onrm + (size_t)v * 3 + 1, ony); stg<NT>(b.onrm + (size_t)v * 3 + 2, onz);
            stg<NT>(reinterpret_cast<f32x4*>(b.otan) + v, f32x4{otx, oty, otz, t.w});
        }
        if (!active) break;
    }
}


// ---- candidate structure: palette loads FIRST (older in the in-order vmcnt), then vertex loads; ping-pong prefetch
struct VIn { float px, py, pz, nx, ny, nz; f32x4 t, w; uint32_t id; };
template <bool NT> __device__ __forceinline__ VIn ldv(const Bufs& b, uint32_t vs) {
    VIn r;
    r.px = ldg<NT>(b.pos + (size_t)vs * 3); r.py = ldg<NT>(b.pos + (size_t)vs * 3 + 1); r.pz = ldg<NT>(b.pos + (size_t)vs * 3 + 2);
    r.nx = ldg<NT>(b.nrm + (size_t)vs * 3); r.ny = ldg<NT>(b.nrm + (size_t)vs * 3 + 1); r.nz = ldg<NT>(b.nrm + (size_t)vs * 3 + 2);
    r.t = ldg<NT>(reinterpret_cast<const f32x4*>(b.tan) + vs);
    r.w = ldg<NT>(reinterpret_cast<const f32x4*>(b.wgt) + vs);
    r.id = ldg<NT>(b.idx + vs);
    return r;
}
template <bool FUSED, bool NT> __device__ __forceinline__ void skin_store(const Bufs& b, const f32x4* rows, const VIn& c, uint32_t v, bool live) {
    float ox = 0, oy = 0, oz = 0, onx = 0, ony = 0, onz = 0, otx = 0, oty = 0, otz = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t bn = (c.id >> (8 * k)) & 0xff;
        const float wk = c.w[k];
        f32x4 r0 = rows[bn * 3], r1 = rows[bn * 3 + 1], r2 = rows[bn * 3 + 2];
        if constexpr (FUSED) {
            ox = __builtin_fmaf(__builtin_fmaf(r0.z, c.pz, __builtin_fmaf(r0.y, c.py, __builtin_fmaf(r0.x, c.px, r0.w))), wk, ox);
            oy = __builtin_fmaf(__builtin_fmaf(r1.z, c.pz, __builtin_fmaf(r1.y, c.py, __builtin_fmaf(r1.x, c.px, r1.w))), wk, oy);
            oz = __builtin_fmaf(__builtin_fmaf(r2.z, c.pz, __builtin_fmaf(r2.y, c.py, __builtin_fmaf(r2.x, c.px, r2.w))), wk, oz);
            onx = __builtin_fmaf(DOT(r0, c.nx, c.ny, c.nz), wk, onx); ony = __builtin_fmaf(DOT(r1, c.nx, c.ny, c.nz), wk, ony); onz = __builtin_fmaf(DOT(r2, c.nx, c.ny, c.nz), wk, onz);
            otx = __builtin_fmaf(DOT(r0, c.t.x, c.t.y, c.t.z), wk, otx); oty = __builtin_fmaf(DOT(r1, c.t.x, c.t.y, c.t.z), wk, oty); otz = __builtin_fmaf(DOT(r2, c.t.x, c.t.y, c.t.z), wk, otz);
        } else {
            ox = ox + (DOTX(r0, c.px, c.py, c.pz) + r0.w) * wk; oy = oy + (DOTX(r1, c.px, c.py, c.pz) + r1.w) * wk; oz = oz + (DOTX(r2, c.px, c.py, c.pz) + r2.w) * wk;
            onx = onx + DOTX(r0, c.nx, c.ny, c.nz) * wk; ony = ony + DOTX(r1, c.nx, c.ny, c.nz) * wk; onz = onz + DOTX(r2, c.nx, c.ny, c.nz) * wk;
            otx = otx + DOTX(r0, c.t.x, c.t.y, c.t.z) * wk; oty = oty + DOTX(r1, c.t.x, c.t.y, c.t.z) * wk; otz = otz + DOTX(r2, c.t.x, c.t.y, c.t.z) * wk;
        }
    }
    if (live) {
        stg<NT>(b.opos + (size_t)v * 3, ox); stg<NT>(b.opos + (size_t)v * 3 + 1, oy); stg<NT>(b.opos + (size_t)v * 3 + 2, oz);
        stg<NT>(b.onrm + (size_t)v * 3, onx); stg<NT>(b.onrm + (size_t)v * 3 + 1, ony); stg<NT>(b.onrm + (size_t)v * 3 + 2, onz);
        stg<NT>(reinterpret_cast<f32x4*>(b.otan) + v, f32x4{otx, oty, otz, c.t.w});
    }
}
// STAGE_FIRST: palette loads issued before the first vertex loads.  PIPE: 0 none, 1 ping-pong prefetch
template <int BLOCK, bool FUSED, bool STAGE_FIRST, int PIPE, bool NT>
__global__ __launch_bounds__(BLOCK) void k_lbs2(Bufs b, const float* pal, uint32_t n_verts, uint32_t total_units) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f32x4* rows = reinterpret_cast<f32x4*>(smem);
    constexpr uint32_t WPB = BLOCK / 64;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t u_begin = (uint32_t)(((uint64_t)blockIdx.x * total_units) / gridDim.x);
    const uint32_t u_end = (uint32_t)(((uint64_t)(blockIdx.x + 1) * total_units) / gridDim.x);
    uint32_t u = u_begin + wave;
    const bool stager = threadIdx.x < NB;   // NB <= BLOCK in this harness
    f32x4 c0, c1, c2, c3;
    VIn A, B;
    uint32_t vA = u * 64 + lane, vB;
    if constexpr (STAGE_FIRST) {
        if (stager) { const f32x4* m = reinterpret_cast<const f32x4*>(pal + (size_t)threadIdx.x * 16); c0 = m[0]; c1 = m[1]; c2 = m[2]; c3 = m[3]; }
        if (u < u_end) A = ldv<NT>(b, vA < n_verts ? vA : 0);
    } else {
        if (u < u_end) A = ldv<NT>(b, vA < n_verts ? vA : 0);
        if (stager) { const f32x4* m = reinterpret_cast<const f32x4*>(pal + (size_t)threadIdx.x * 16); c0 = m[0]; c1 = m[1]; c2 = m[2]; c3 = m[3]; }
    }
    if (stager) {
        const uint32_t bn = threadIdx.x;
        rows[bn * 3 + 0] = f32x4{c0.x, c1.x, c2.x, c3.x};
        rows[bn * 3 + 1] = f32x4{c0.y, c1.y, c2.y, c3.y};
        rows[bn * 3 + 2] = f32x4{c0.z, c1.z, c2.z, c3.z};
    }
    __syncthreads();
    if constexpr (PIPE == 0) {
        while (u < u_end) {
            skin_store<FUSED, NT>(b, rows, A, vA, vA < n_verts);
            u += WPB; vA = u * 64 + lane;
            if (u < u_end) A = ldv<NT>(b, vA < n_verts ? vA : 0);
        }
    } else {
        while (u < u_end) {
            const uint32_t u1 = u + WPB; vB = u1 * 64 + lane;
            if (u1 < u_end) B = ldv<NT>(b, vB < n_verts ? vB : 0);
            skin_store<FUSED, NT>(b, rows, A, vA, vA < n_verts);
            if (u1 >= u_end) break;
            const uint32_t u2 = u1 + WPB; vA = u2 * 64 + lane;
            if (u2 < u_end) A = ldv<NT>(b, vA < n_verts ? vA : 0);
            skin_store<FUSED, NT>(b, rows, B, vB, vB < n_verts);
            u = u2;
        }
    }
}


// ---- instrumented variant: per-wave wall-clock stamps (100 MHz constant counter)
template <int BLOCK, int MODE>   // MODE 0: passthrough no staging, 1: staging only (passthrough), 3: full fused
__global__ __launch_bounds__(BLOCK) void k_dbg(Bufs b, const float* pal, uint32_t n_verts, uint32_t total_units, unsigned long long* stamps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f32x4* rows = reinterpret_cast<f32x4*>(smem);
    constexpr uint32_t WPB = BLOCK / 64;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long t0 = wall_clock64();
    const uint32_t u_begin = (uint32_t)(((uint64_t)blockIdx.x * total_units) / gridDim.x);
    const uint32_t u_end = (uint32_t)(((uint64_t)(blockIdx.x + 1) * total_units) / gridDim.x);
    uint32_t u = u_begin + wave;
    uint32_t vA = u * 64 + lane;
    VIn A;
    if (u < u_end) A = ldv<true>(b, vA < n_verts ? vA : 0);
    if constexpr (MODE >= 1) {
        if (threadIdx.x < NB) {
            const f32x4* m = reinterpret_cast<const f32x4*>(pal + (size_t)threadIdx.x * 16);
            f32x4 c0 = m[0], c1 = m[1], c2 = m[2], c3 = m[3];
            const uint32_t bn = threadIdx.x;
            rows[bn * 3 + 0] = f32x4{c0.x, c1.x, c2.x, c3.x};
            rows[bn * 3 + 1] = f32x4{c0.y, c1.y, c2.y, c3.y};
            rows[bn * 3 + 2] = f32x4{c0.z, c1.z, c2.z, c3.z};
        }
        __syncthreads();
    }
    const unsigned long long t1 = wall_clock64();
    unsigned long long t2 = 0;
    int n = 0;
    while (u < u_end) {
        if constexpr (MODE == 3) skin_store<true, true>(b, rows, A, vA, vA < n_verts);
        else {
            if (vA < n_verts) {
                stg<true>(b.opos + (size_t)vA * 3, A.px + A.w.x); stg<true>(b.opos + (size_t)vA * 3 + 1, A.py); stg<true>(b.opos + (size_t)vA * 3 + 2, A.pz);
                stg<true>(b.onrm + (size_t)vA * 3, A.nx); stg<true>(b.onrm + (size_t)vA * 3 + 1, A.ny); stg<true>(b.onrm + (size_t)vA * 3 + 2, A.nz + __uint_as_float(A.id) * 0.f);
                stg<true>(reinterpret_cast<f32x4*>(b.otan) + vA, A.t);
            }
        }
        if (n == 0) t2 = wall_clock64();
        ++n;
        u += WPB; vA = u * 64 + lane;
        if (u < u_end) A = ldv<true>(b, vA < n_verts ? vA : 0);
    }
    __builtin_amdgcn_s_waitcnt(0);   // all stores acked
    const unsigned long long t3 = wall_clock64();
    if (lane == 0) {
        unsigned long long* o = stamps + ((size_t)blockIdx.x * WPB + wave) * 4;
        o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3;
    }
}

// same bytes, all-float4 planes (15 in, 10 out per 4 vertices), no math
template <int BLOCK, bool NT>
__global__ __launch_bounds__(BLOCK) void k_copy4(const f32x4* src, f32x4* dst, uint32_t groups) {
    for (uint32_t g = blockIdx.x * BLOCK + threadIdx.x; g < groups; g += gridDim.x * BLOCK) {
        f32x4 acc[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) acc[i] = ldg<NT>(src + (size_t)i * groups + g);
#pragma unroll
        for (int i = 10; i < 15; ++i) acc[i - 10] += ldg<NT>(src + (size_t)i * groups + g);
#pragma unroll
        for (int i = 0; i < 10; ++i) stg<NT>(dst + (size_t)i * groups + g, acc[i]);
    }
}

int main(int argc, char** argv) {
    int steps = argc > 1 ? atoi(argv[1]) : 300;
    CK(hipSetDevice(0));
    hipStream_t s; CK(hipStreamCreate(&s));
    std::vector<Bufs> sets(SETS);
    std::vector<uint32_t> hidx(NV);
    for (int v = 0; v < NV; ++v) { uint32_t b0 = (uint32_t)(((uint64_t)v * NB) / NV); hidx[v] = (b0 & 255) | (((b0 + 1) & 255) << 8) | (((b0 + 255) & 255) << 16) | (((b0 + 3) & 255) << 24); }
    std::vector<float> hf(NV * 4, 0.25f);
    for (auto& b : sets) {
        CK(hipMalloc(&b.pos, NV * 12 + 4096)); CK(hipMalloc(&b.nrm, NV * 12 + 4096)); CK(hipMalloc(&b.tan, NV * 16 + 4096)); CK(hipMalloc(&b.wgt, NV * 16 + 4096));
        CK(hipMalloc(&b.idx, NV * 4 + 4096)); CK(hipMalloc(&b.opos, NV * 12 + 4096)); CK(hipMalloc(&b.onrm, NV * 12 + 4096)); CK(hipMalloc(&b.otan, NV * 16 + 4096));
        CK(hipMemcpy(b.pos, hf.data(), NV * 12, hipMemcpyHostToDevice)); CK(hipMemcpy(b.nrm, hf.data(), NV * 12, hipMemcpyHostToDevice));
        CK(hipMemcpy(b.tan, hf.data(), NV * 16, hipMemcpyHostToDevice)); CK(hipMemcpy(b.wgt, hf.data(), NV * 16, hipMemcpyHostToDevice));
        CK(hipMemcpy(b.idx, hidx.data(), NV * 4, hipMemcpyHostToDevice));
    }
    float* pal; CK(hipMalloc(&pal, NB * 64));
    std::vector<float> hp(NB * 16, 0.f); for (int b = 0; b < NB; ++b) { hp[b * 16] = hp[b * 16 + 5] = hp[b * 16 + 10] = hp[b * 16 + 15] = 1.f; hp[b * 16 + 12] = 0.001f * b; }
    CK(hipMemcpy(pal, hp.data(), NB * 64, hipMemcpyHostToDevice));
    std::vector<f32x4*> csrc(SETS), cdst(SETS);
    const uint32_t groups = NV / 4;
    for (int i = 0; i < SETS; ++i) { CK(hipMalloc(&csrc[i], (size_t)groups * 15 * 16)); CK(hipMalloc(&cdst[i], (size_t)groups * 10 * 16)); CK(hipMemset(csrc[i], 0, (size_t)groups * 15 * 16)); }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint32_t units = (NV + 63) / 64;

    auto timeit = [&](const char* name, auto launch) {
        std::vector<float> ts;
        for (int r = 0; r < 3; ++r) {
            for (int i = 0; i < 20; ++i) launch(i);
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < steps; ++i) launch(i);
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms * 1e3f / steps);
        }
        std::sort(ts.begin(), ts.end());
        printf("%-44s %7.2f us  %7.1f GB/s\n", name, ts[1], 100e6 / ts[1] / 1e3);
        CK(hipGetLastError());
    };
#define RUN_LBS(BLOCK, BPCU, MODE, NT) timeit("lbs block=" #BLOCK " bpcu=" #BPCU " mode=" #MODE " nt=" #NT, [&](int i) { \
        hipLaunchKernelGGL((k_lbs<BLOCK, MODE, NT>), dim3(256 * BPCU), dim3(BLOCK), NB * 48, s, sets[i % SETS], pal, (uint32_t)NV, units); })
#define RUN_C4(BLOCK, BPCU, NT) timeit("copy4 block=" #BLOCK " bpcu=" #BPCU " nt=" #NT, [&](int i) { \
        hipLaunchKernelGGL((k_copy4<BLOCK, NT>), dim3(256 * BPCU), dim3(BLOCK), 0, s, csrc[i % SETS], cdst[i % SETS], groups); })
    RUN_C4(256, 4, true); RUN_C4(256, 8, true); RUN_C4(256, 2, true); RUN_C4(256, 8, false);
    RUN_LBS(256, 8, 0, true); RUN_LBS(256, 4, 0, true); RUN_LBS(512, 4, 0, true); RUN_LBS(1024, 2, 0, true); RUN_LBS(256, 8, 0, false);
    RUN_LBS(256, 8, 1, true); RUN_LBS(512, 4, 1, true); RUN_LBS(1024, 2, 1, true);
    RUN_LBS(256, 8, 2, true); RUN_LBS(512, 4, 2, true); RUN_LBS(1024, 2, 2, true);
    RUN_LBS(256, 8, 3, true); RUN_LBS(256, 4, 3, true); RUN_LBS(512, 4, 3, true); RUN_LBS(1024, 2, 3, true);
    RUN_LBS(256, 8, 4, true); RUN_LBS(256, 4, 4, true); RUN_LBS(512, 4, 4, true); RUN_LBS(1024, 2, 4, true);

#define RUN_L2(BLOCK, BPCU, FUSED, SF, PIPE) timeit("lbs2 block=" #BLOCK " bpcu=" #BPCU " fused=" #FUSED " stagefirst=" #SF " pipe=" #PIPE, [&](int i) { \
        hipLaunchKernelGGL((k_lbs2<BLOCK, FUSED, SF, PIPE, true>), dim3(256 * BPCU), dim3(BLOCK), NB * 48, s, sets[i % SETS], pal, (uint32_t)NV, units); })
    RUN_L2(256, 8, true, false, 0); RUN_L2(256, 8, true, true, 0); RUN_L2(256, 8, true, false, 1); RUN_L2(256, 8, true, true, 1);
    RUN_L2(256, 4, true, true, 1); RUN_L2(256, 2, true, true, 1); RUN_L2(256, 6, true, true, 1);
    RUN_L2(512, 4, true, true, 1); RUN_L2(512, 2, true, true, 1); RUN_L2(1024, 1, true, true, 1); RUN_L2(1024, 2, true, true, 1);
    RUN_L2(256, 8, false, true, 0); RUN_L2(256, 8, false, true, 1); RUN_L2(256, 4, false, true, 1); RUN_L2(512, 2, false, true, 1); RUN_L2(512, 4, false, true, 1); RUN_L2(1024, 1, false, true, 1);

    {
        const int G = 256 * 8, W = G * 4;
        unsigned long long* dst; CK(hipMalloc(&dst, (size_t)W * 4 * 8));
        std::vector<unsigned long long> h((size_t)W * 4);
        auto analyze = [&](const char* name) {
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h.data(), dst, h.size() * 8, hipMemcpyDeviceToHost));
            unsigned long long base = ~0ull; for (int w = 0; w < W; ++w) base = std::min(base, h[w * 4]);
            std::vector<double> t0, t1, t2, t3;
            for (int w = 0; w < W; ++w) { t0.push_back((h[w*4]-base)*0.01); t1.push_back((h[w*4+1]-base)*0.01); t2.push_back((h[w*4+2]-base)*0.01); t3.push_back((h[w*4+3]-base)*0.01); }
            auto pct = [](std::vector<double> v, double p) { std::sort(v.begin(), v.end()); return v[(size_t)(p * (v.size() - 1))]; };
            printf("%s (us since first wave start): ", name);
            const char* names[4] = {"start", "staged", "unit0done", "end"}; std::vector<double>* vs[4] = {&t0, &t1, &t2, &t3};
            for (int k = 0; k < 4; ++k) printf(" %s[p0 %.2f p10 %.2f p50 %.2f p90 %.2f p100 %.2f]", names[k], pct(*vs[k], 0), pct(*vs[k], 0.1), pct(*vs[k], 0.5), pct(*vs[k], 0.9), pct(*vs[k], 1.0));
            printf("\n");
        };
        for (int rep = 0; rep < 2; ++rep) {
            for (int i = 0; i < 9; ++i) hipLaunchKernelGGL((k_dbg<256, 0>), dim3(G), dim3(256), NB * 48, s, sets[i % SETS], pal, (uint32_t)NV, units, dst);
            analyze("dbg mode0 passthrough ");
            for (int i = 0; i < 9; ++i) hipLaunchKernelGGL((k_dbg<256, 1>), dim3(G), dim3(256), NB * 48, s, sets[i % SETS], pal, (uint32_t)NV, units, dst);
            analyze("dbg mode1 +staging    ");
            for (int i = 0; i < 9; ++i) hipLaunchKernelGGL((k_dbg<256, 3>), dim3(G), dim3(256), NB * 48, s, sets[i % SETS], pal, (uint32_t)NV, units, dst);
            analyze("dbg mode3 full fused  ");
        }
    }

    {   // E3: alternate launches over K streams; wall time per launch (host clock around a full sync)
        hipStream_t ss[4]; for (auto& x : ss) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
        for (int K : {1, 2, 3, 4}) {
            for (int variant = 0; variant < 2; ++variant) {
                auto launch = [&](int i) {
                    hipStream_t st = ss[i % K];
                    if (variant == 0) hipLaunchKernelGGL((k_lbs<256, 0, true>), dim3(256 * 8), dim3(256), NB * 48, st, sets[i % SETS], pal, (uint32_t)NV, units);
                    else hipLaunchKernelGGL((k_lbs2<256, false, true, 0, true>), dim3(256 * 8), dim3(256), NB * 48, st, sets[i % SETS], pal, (uint32_t)NV, units);
                };
                for (int i = 0; i < 40; ++i) launch(i);
                CK(hipDeviceSynchronize());
                std::vector<double> ts;
                for (int r = 0; r < 3; ++r) {
                    auto t0 = std::chrono::steady_clock::now();
                    for (int i = 0; i < steps; ++i) launch(i);
                    CK(hipDeviceSynchronize());
                    ts.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / steps);
                }
                std::sort(ts.begin(), ts.end());
                printf("streams=%d %s  %7.2f us/launch  %7.1f GB/s\n", K, variant ? "lbs2 unfused" : "passthrough ", ts[1], 100e6 / ts[1] / 1e3);
            }
        }
    }
    return 0;
}
