// lbs_leaves.h -- the per-vertex device code of the skinning kernels: vector types, cache-policy loads / stores, palette staging into the
// packed-math LDS layout, skin_vertex (the reference's operation order, see lbs_kernels.hip's head) and the vertex streams as
// buffer resources.  Shared by lbs_kernels.hip (every skinning kernel) and anim_kernels.hip (the one-launch frame that goes on to
// the vertices): ONE definition, so a vertex skinned inside a pose launch has the bits of one skinned by lbs_skin.
#pragma once
#include "fyx_internal.h"

namespace fyx {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <bool NT, typename T>
__device__ __forceinline__ T ldg(const T* p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <bool NT, typename T>
__device__ __forceinline__ void stg(T* p, T v) {
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

template <bool NT>
__device__ __forceinline__ void ld3(const float* p, float& x, float& y, float& z) {
    x = ldg<NT>(p); y = ldg<NT>(p + 1); z = ldg<NT>(p + 2);
}
template <bool NT>
__device__ __forceinline__ void st3(float* p, float x, float y, float z) {
    stg<NT>(p, x); stg<NT>(p + 1, y); stg<NT>(p + 2, z);
}

// ---------------------------------------------------------------------------------------
// Palette staging: global column-major mat4 -> three float4 per bone in LDS, arranged for
// PACKED f32 math (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 work on register pairs and can
// broadcast either half of a pair):
//     A = (m00, m10, m01, m11)   B = (m02, m12, t0, t1)   C = (m20, m21, m22, t2)
// so the x and y rows of a matrix sit side by side in a 64-bit register pair
//   (pos.x, pos.y) = ((A.lo*px + A.hi*py) + B.lo*pz) + B.hi           3 pk_mul + 3 pk_add
// while a single coefficient m_rk is the lo or hi half of one of those pairs, so normal and
// tangent are transformed together as the pair (n_r, t_r) = (m_r0*(nx,tx) + m_r1*(ny,ty)) +
// m_r2*(nz,tz) with the coefficient broadcast by op_sel -- no register shuffling.  Every output
// component keeps exactly the reference's operation order (each pk instruction rounds its two
// lanes independently), so the packed EXACT path stays bit-identical to the CPU loop while
// issuing ~half the VALU instructions of the scalar form.
// row3[b] = (m30, m31, m32, m33) is only read by the projective path.
// Returns (per thread) whether any staged matrix is projective.
// ---------------------------------------------------------------------------------------
struct PaletteRegs {
    f32x4 c0, c1, c2, c3;
};

// Issue the global loads of this thread's bone (n_bones <= 256 <= BLOCK: at most one bone each).
__device__ __forceinline__ PaletteRegs palette_fetch(const float* __restrict__ pal, uint32_t n_bones,
                                                     int tid) {
    // unconditional (clamped) so that no branch sits between these loads, the vertex loads that
    // follow and the LDS commit: the compiler's vmcnt bookkeeping stays exact only in straight-line code
    const uint32_t b = (uint32_t)tid < n_bones ? (uint32_t)tid : n_bones - 1;
    const f32x4* m = reinterpret_cast<const f32x4*>(pal + (size_t)b * 16);
    PaletteRegs r;
    r.c0 = m[0]; r.c1 = m[1]; r.c2 = m[2]; r.c3 = m[3];
    return r;
}

// Write the fetched bone to LDS in the packed-math layout; true if that matrix is projective.
__device__ __forceinline__ bool palette_commit(const PaletteRegs& r, uint32_t n_bones, f32x4* rows,
                                               f32x4* row3, int tid) {
    if ((uint32_t)tid >= n_bones) return false;
    rows[tid * 3 + 0] = f32x4{r.c0.x, r.c0.y, r.c1.x, r.c1.y};
    rows[tid * 3 + 1] = f32x4{r.c2.x, r.c2.y, r.c3.x, r.c3.y};
    rows[tid * 3 + 2] = f32x4{r.c0.z, r.c1.z, r.c2.z, r.c3.z};
    row3[tid] = f32x4{r.c0.w, r.c1.w, r.c2.w, r.c3.w};
    return !(r.c0.w == 0.0f && r.c1.w == 0.0f && r.c2.w == 0.0f && r.c3.w == 1.0f);
}

// Whole-palette staging by a workgroup of any size (AABB kernel).
__device__ __forceinline__ bool stage_palette(const float* __restrict__ pal, uint32_t n_bones,
                                              f32x4* rows, f32x4* row3, int tid, int nthreads) {
    bool projective = false;
    for (uint32_t b = tid; b < n_bones; b += nthreads) {
        const f32x4* m = reinterpret_cast<const f32x4*>(pal + (size_t)b * 16);
        f32x4 c0 = m[0], c1 = m[1], c2 = m[2], c3 = m[3];
        rows[b * 3 + 0] = f32x4{c0.x, c0.y, c1.x, c1.y};
        rows[b * 3 + 1] = f32x4{c2.x, c2.y, c3.x, c3.y};
        rows[b * 3 + 2] = f32x4{c0.z, c1.z, c2.z, c3.z};
        row3[b] = f32x4{c0.w, c1.w, c2.w, c3.w};
        projective |= !(c0.w == 0.0f && c1.w == 0.0f && c2.w == 0.0f && c3.w == 1.0f);
    }
    return projective;
}

// mat3 rows * v, reference order ((m0*x + m1*y) + m2*z); used by the projective normaliser.
template <bool EXACT>
__device__ __forceinline__ float dot3(f32x4 r, float x, float y, float z) {
    if constexpr (EXACT) return (r.x * x + r.y * y) + r.z * z;
    else return __builtin_fmaf(r.z, z, __builtin_fmaf(r.y, y, r.x * x));
}

// (a*x + b*y) + c*z on pairs; fused: fma(c, z, fma(b, y, a*x)).
template <bool EXACT>
__device__ __forceinline__ f32x2 dot3p(f32x2 a, f32x2 x, f32x2 b, f32x2 y, f32x2 c, f32x2 z) {
    if constexpr (EXACT) return (a * x + b * y) + c * z;
    else return __builtin_elementwise_fma(c, z, __builtin_elementwise_fma(b, y, a * x));
}
template <bool EXACT>
__device__ __forceinline__ f32x2 accp(f32x2 a, f32x2 r, f32x2 w) {
    if constexpr (EXACT) return a + r * w;
    else return __builtin_elementwise_fma(r, w, a);
}
template <bool EXACT>
__device__ __forceinline__ float acc(float a, float r, float w) {
    if constexpr (EXACT) return a + r * w;
    else return __builtin_fmaf(r, w, a);
}
__device__ __forceinline__ f32x2 splat(float v) { return f32x2{v, v}; }

struct Skinned {
    float px, py, pz, nx, ny, nz, tx, ty, tz;
};

// One vertex, four influences.  MASK bit0 position, bit1 normal, bit2 tangent.
// SEQ: the four influences one after another (three LDS rows live at a time instead of twelve: ~74 instead of ~112 VGPRs in
// the crowd kernel) -- the same operations in the same order per output component, so the same bits.
template <bool EXACT, int MASK, bool PROJ, bool SEQ = false>
__device__ __forceinline__ Skinned skin_vertex_impl(const f32x4* __restrict__ rows,
                                               const f32x4* __restrict__ row3,
                                               uint32_t id, f32x4 w, float px, float py, float pz,
                                               float nx, float ny, float nz, float tx, float ty,
                                               float tz) {
    f32x2 o_pxy = {0.f, 0.f}, o_x = {0.f, 0.f}, o_y = {0.f, 0.f}, o_z = {0.f, 0.f};  // o_r = (n_r, t_r)
    float o_pz = 0.f;
    const f32x2 vx = {nx, tx}, vy = {ny, ty}, vz = {nz, tz};
#pragma unroll(SEQ ? 1 : 4)
    for (int k = 0; k < 4; ++k) {
        const uint32_t b = (id >> (8 * k)) & 0xffu;
        const float wk = k == 0 ? w.x : k == 1 ? w.y : k == 2 ? w.z : w.w;
        const f32x4 A = rows[b * 3 + 0];
        const f32x4 B = rows[b * 3 + 1];
        const f32x4 C = rows[b * 3 + 2];
        if constexpr (MASK & 1) {
            f32x2 xy;
            float z;
            if constexpr (EXACT) {
                xy = dot3p<true>(A.xy, splat(px), A.zw, splat(py), B.xy, splat(pz)) + B.zw;
                z = ((C.x * px + C.y * py) + C.z * pz) + C.w;
            } else {
                xy = __builtin_elementwise_fma(
                    B.xy, splat(pz),
                    __builtin_elementwise_fma(A.zw, splat(py), __builtin_elementwise_fma(A.xy, splat(px), B.zw)));
                z = __builtin_fmaf(C.z, pz, __builtin_fmaf(C.y, py, __builtin_fmaf(C.x, px, C.w)));
            }
            if constexpr (PROJ) {
                const f32x4 r3 = row3[b];
                const float n = dot3<EXACT>(r3, px, py, pz) + r3.w;
                if (n != 0.0f) { xy.x = xy.x / n; xy.y = xy.y / n; z = z / n; }
            }
            o_pxy = accp<EXACT>(o_pxy, xy, splat(wk));
            o_pz = acc<EXACT>(o_pz, z, wk);
        }
        if constexpr ((MASK & 6) != 0) {
            // (n_r, t_r) for r = x, y, z; with only one of the two streams present the other lane
            // carries zeros and is never stored.
            const f32x2 rx = dot3p<EXACT>(splat(A.x), vx, splat(A.z), vy, splat(B.x), vz);
            const f32x2 ry = dot3p<EXACT>(splat(A.y), vx, splat(A.w), vy, splat(B.y), vz);
            const f32x2 rz = dot3p<EXACT>(splat(C.x), vx, splat(C.y), vy, splat(C.z), vz);
            o_x = accp<EXACT>(o_x, rx, splat(wk));
            o_y = accp<EXACT>(o_y, ry, splat(wk));
            o_z = accp<EXACT>(o_z, rz, splat(wk));
        }
    }
    Skinned o;
    o.px = o_pxy.x; o.py = o_pxy.y; o.pz = o_pz;
    o.nx = o_x.x; o.ny = o_y.x; o.nz = o_z.x;
    o.tx = o_x.y; o.ty = o_y.y; o.tz = o_z.y;
    return o;
}

// Fused mode for VALU-bound launches (crowds): blend the four matrices first, M = sum_k w_k M_k, then transform
// position / normal / tangent once.  For affine palettes this is the same linear map as sum_k (M_k v) w_k -- only
// the rounding differs (a few 1e-7 relative, inside the 1e-5 bar of lbs.exact=0) -- and costs ~40 packed VALU
// instead of ~138: 24 for the blend (three float4 per bone, two pk-FMAs each) and 15 for the three transforms.
template <int MASK>
__device__ __forceinline__ Skinned skin_vertex_blended(const f32x4* __restrict__ rows, uint32_t id, f32x4 w, float px,
                                                       float py, float pz, float nx, float ny, float nz, float tx,
                                                       float ty, float tz) {
    f32x2 a_lo, a_hi, b_lo, b_hi, c_lo, c_hi;
    {
        const uint32_t b0 = id & 0xffu;
        const f32x4 A = rows[b0 * 3 + 0], B = rows[b0 * 3 + 1], C = rows[b0 * 3 + 2];
        const f32x2 ww = splat(w[0]);
        a_lo = A.xy * ww; a_hi = A.zw * ww; b_lo = B.xy * ww; b_hi = B.zw * ww; c_lo = C.xy * ww; c_hi = C.zw * ww;
    }
#pragma unroll
    for (int k = 1; k < 4; ++k) {
        const uint32_t b = (id >> (8 * k)) & 0xffu;
        const f32x4 A = rows[b * 3 + 0], B = rows[b * 3 + 1], C = rows[b * 3 + 2];
        const f32x2 ww = splat(w[k]);
        a_lo = __builtin_elementwise_fma(A.xy, ww, a_lo); a_hi = __builtin_elementwise_fma(A.zw, ww, a_hi);
        b_lo = __builtin_elementwise_fma(B.xy, ww, b_lo); b_hi = __builtin_elementwise_fma(B.zw, ww, b_hi);
        c_lo = __builtin_elementwise_fma(C.xy, ww, c_lo); c_hi = __builtin_elementwise_fma(C.zw, ww, c_hi);
    }
    // blended rows: (m00,m10) (m01,m11) (m02,m12) (t0,t1) (m20,m21) (m22,t2)
    Skinned o;
    o.px = o.py = o.pz = o.nx = o.ny = o.nz = o.tx = o.ty = o.tz = 0.f;
    if constexpr (MASK & 1) {
        const f32x2 xy = __builtin_elementwise_fma(b_lo, splat(pz), __builtin_elementwise_fma(a_hi, splat(py),
                         __builtin_elementwise_fma(a_lo, splat(px), b_hi)));
        o.px = xy.x; o.py = xy.y;
        o.pz = __builtin_fmaf(c_hi.x, pz, __builtin_fmaf(c_lo.y, py, __builtin_fmaf(c_lo.x, px, c_hi.y)));
    }
    if constexpr ((MASK & 6) != 0) {
        const f32x2 vx = {nx, tx}, vy = {ny, ty}, vz = {nz, tz};
        const f32x2 rx = __builtin_elementwise_fma(splat(b_lo.x), vz, __builtin_elementwise_fma(splat(a_hi.x), vy, splat(a_lo.x) * vx));
        const f32x2 ry = __builtin_elementwise_fma(splat(b_lo.y), vz, __builtin_elementwise_fma(splat(a_hi.y), vy, splat(a_lo.y) * vx));
        const f32x2 rz = __builtin_elementwise_fma(splat(c_hi.x), vz, __builtin_elementwise_fma(splat(c_lo.y), vy, splat(c_lo.x) * vx));
        o.nx = rx.x; o.ny = ry.x; o.nz = rz.x;
        o.tx = rx.y; o.ty = ry.y; o.tz = rz.y;
    }
    return o;
}

// `projective` is workgroup-uniform: the affine path (the only kind of palette Fyrox produces)
// is one straight-line block of packed math; the homogeneous divide lives in its own copy.
template <bool EXACT, int MASK, bool BLEND_FIRST = false, bool SEQ = false>
__device__ __forceinline__ Skinned skin_vertex(const f32x4* __restrict__ rows,
                                               const f32x4* __restrict__ row3, bool projective,
                                               uint32_t id, f32x4 w, float px, float py, float pz,
                                               float nx, float ny, float nz, float tx, float ty,
                                               float tz) {
    if (projective) return skin_vertex_impl<EXACT, MASK, true, SEQ>(rows, row3, id, w, px, py, pz, nx, ny, nz, tx, ty, tz);
    if constexpr (!EXACT && BLEND_FIRST) return skin_vertex_blended<MASK>(rows, id, w, px, py, pz, nx, ny, nz, tx, ty, tz);
    return skin_vertex_impl<EXACT, MASK, false, SEQ>(rows, row3, id, w, px, py, pz, nx, ny, nz, tx, ty, tz);
}

// p and n are kept as whole 96-bit values (one register triple each): a loop that carries a vertex from one
// iteration to the next then carries the triple a dwordx3 load fills, instead of six scalars the register allocator
// is free to scatter (and has to gather again with moves that wait for the load).
template <int MASK>
struct VertexIn {
    f32x3 p, n;
    f32x4 t, w;
    uint32_t id;
};

template <bool NT, int MASK>
__device__ __forceinline__ VertexIn<MASK> load_vertex(const LbsArgs& a, uint32_t vs) {
    VertexIn<MASK> r;
    r.p = r.n = f32x3{0.f, 0.f, 0.f};
    r.t = f32x4{0.f, 0.f, 0.f, 0.f};
    float x, y, z;
    if constexpr (MASK & 1) { ld3<NT>(a.pos + (size_t)vs * 3, x, y, z); r.p = f32x3{x, y, z}; }
    if constexpr (MASK & 2) { ld3<NT>(a.nrm + (size_t)vs * 3, x, y, z); r.n = f32x3{x, y, z}; }
    if constexpr (MASK & 4) r.t = ldg<NT>(reinterpret_cast<const f32x4*>(a.tan) + vs);
    r.w = ldg<NT>(reinterpret_cast<const f32x4*>(a.wgt) + vs);
    r.id = ldg<NT>(a.idx + vs);
    return r;
}

template <int MASK>
__device__ __forceinline__ void pin_vertex(VertexIn<MASK>& r) {
    asm volatile("" : "+v"(r.p), "+v"(r.n));
    asm volatile("" : "+v"(r.t), "+v"(r.w), "+v"(r.id));
}

// ---------------------------------------------------------------------------------------
// Vertex streams as buffer resources (lbs_skin_dyn): buffer_load / buffer_store carry the cache policy in the
// instruction (aux bits below), the compiler knows them (exact vmcnt counts, hazards), and an access past the
// stream's last byte is dropped by the hardware (loads return 0) -- no per-lane bounds test.
//   aux: 1 = sc0, 2 = nt, 16 = sc1   (gfx940+ cache-policy bits of the raw-buffer builtins)
// ---------------------------------------------------------------------------------------
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
struct VtxBuffers {
    __amdgpu_buffer_rsrc_t pos, nrm, tan, wgt, idx, out_pos, out_nrm, out_tan;
};
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_stream(const void* p, uint32_t bytes) {
    // null stream -> zero records: every access is out of range
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, p ? bytes : 0u, 0x00020000);
}
__device__ __forceinline__ VtxBuffers make_vtx_buffers(const LbsArgs& a) {
    VtxBuffers b;
    b.pos = make_stream(a.pos, a.n_verts * 12u);
    b.nrm = make_stream(a.nrm, a.n_verts * 12u);
    b.tan = make_stream(a.tan, a.n_verts * 16u);
    b.wgt = make_stream(a.wgt, a.n_verts * 16u);
    b.idx = make_stream(a.idx, a.n_verts * 4u);
    b.out_pos = make_stream(a.out_pos, a.n_verts * 12u);
    b.out_nrm = make_stream(a.out_nrm, a.n_verts * 12u);
    b.out_tan = make_stream(a.out_tan, a.n_verts * 16u);
    return b;
}
template <int MASK, int AUX>
__device__ __forceinline__ VertexIn<MASK> load_vertex_buf(const VtxBuffers& b, uint32_t v) {
    VertexIn<MASK> r;
    r.p = r.n = f32x3{0.f, 0.f, 0.f};
    r.t = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (MASK & 1) r.p = __builtin_bit_cast(f32x3, __builtin_amdgcn_raw_buffer_load_b96(b.pos, v * 12u, 0, AUX));
    if constexpr (MASK & 2) r.n = __builtin_bit_cast(f32x3, __builtin_amdgcn_raw_buffer_load_b96(b.nrm, v * 12u, 0, AUX));
    if constexpr (MASK & 4) r.t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b.tan, v * 16u, 0, AUX));
    r.w = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b.wgt, v * 16u, 0, AUX));
    r.id = __builtin_amdgcn_raw_buffer_load_b32(b.idx, v * 4u, 0, AUX);
    return r;
}
template <int MASK, int AUX>
__device__ __forceinline__ void store_vertex_buf(const VtxBuffers& b, uint32_t v, const Skinned& o, float tw) {
    if constexpr (MASK & 1)
        __builtin_amdgcn_raw_buffer_store_b96(__builtin_bit_cast(u32x3, f32x3{o.px, o.py, o.pz}), b.out_pos, v * 12u, 0, AUX);
    if constexpr (MASK & 2)
        __builtin_amdgcn_raw_buffer_store_b96(__builtin_bit_cast(u32x3, f32x3{o.nx, o.ny, o.nz}), b.out_nrm, v * 12u, 0, AUX);
    if constexpr (MASK & 4)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{o.tx, o.ty, o.tz, tw}), b.out_tan, v * 16u, 0, AUX);
}

}  // namespace fyx
