/*
 * fyrox_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY). See fyrox_oracle.h.
 *
 * Scalar IEEE f32, compiled -O2 -ffp-contract=off -fno-fast-math so that every
 * a*b+c below is two rounded operations, as in the Rust reference.
 * file:line citations are relative to /root/reference.
 */
#include "fyrox_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ======================================================================== */
/* nalgebra 0.35 leaves (un-vendored dependency; semantics restated).        */
/* ======================================================================== */

void fo_mat4_identity(float out[16]) {
    memset(out, 0, 16 * sizeof(float));
    out[0] = out[5] = out[10] = out[15] = 1.0f;
}

/* nalgebra Matrix*Matrix for statically sized operands: gemm -> one gemv per
 * result column -> axcpy per k:  y = (1*a_col_k)*b_kj            (k == 0)
 *                                y = (1*a_col_k)*b_kj + 1*y      (k > 0)
 * Call sites: scene/graph/mod.rs:1216, scene/mesh/mod.rs:497,787-788. */
void fo_mat4_mul(const float a[16], const float b[16], float out[16]) {
    float r[16];
    for (int j = 0; j < 4; ++j) {
        for (int i = 0; i < 4; ++i) {
            float y = a[0 * 4 + i] * b[j * 4 + 0];
            for (int k = 1; k < 4; ++k) {
                y = a[k * 4 + i] * b[j * 4 + k] + y;
            }
            r[j * 4 + i] = y;
        }
    }
    memcpy(out, r, sizeof r);
}

/* mat3(M) * v with nalgebra's gemv order ((m_i0*x + m_i1*y) + m_i2*z).
 * Used for normals/tangents, spec: fyrox-material/src/shader/standard/opengl/
 * standard.shader:192-200 (`mat3(m0) * inputNormal`). */
void fo_mat3_of_mat4_mul_vec(const float m[16], const float v[3], float out[3]) {
    float r[3];
    for (int i = 0; i < 3; ++i) {
        float y = m[0 * 4 + i] * v[0];
        y = m[1 * 4 + i] * v[1] + y;
        y = m[2 * 4 + i] * v[2] + y;
        r[i] = y;
    }
    out[0] = r[0]; out[1] = r[1]; out[2] = r[2];
}

/* nalgebra Matrix4::transform_point:
 *   n = normalizer.tr_dot(pt) + m33          (3-element dot: (a+b)+c)
 *   if n != 0 { (transform*pt + translation) / n } else { transform*pt + translation }
 * Call sites: scene/mesh/mod.rs:482-484, 515-517. */
void fo_mat4_transform_point(const float m[16], const float p[3], float out[3]) {
    float r[3];
    fo_mat3_of_mat4_mul_vec(m, p, r);
    r[0] = r[0] + m[12];
    r[1] = r[1] + m[13];
    r[2] = r[2] + m[14];
    float n = ((m[3] * p[0] + m[7] * p[1]) + m[11] * p[2]) + m[15];
    if (n != 0.0f) {
        r[0] = r[0] / n; r[1] = r[1] / n; r[2] = r[2] / n;
    }
    out[0] = r[0]; out[1] = r[1]; out[2] = r[2];
}

/* nalgebra Matrix4::transform_vector: n = normalizer.tr_dot(v);
 * if n != 0 { transform * (v / n) } else { transform * v }.
 * Call sites: scene/mesh/mod.rs:114,117 (static-mesh path; kept for completeness). */
void fo_mat4_transform_vector(const float m[16], const float v[3], float out[3]) {
    float n = (m[3] * v[0] + m[7] * v[1]) + m[11] * v[2];
    float t[3] = { v[0], v[1], v[2] };
    if (n != 0.0f) { t[0] = v[0] / n; t[1] = v[1] / n; t[2] = v[2] / n; }
    fo_mat3_of_mat4_mul_vec(m, t, out);
}

/* nalgebra dotx() 4-element special case: a=a0b0 b=a1b1 c=a2b2 d=a3b3; a+=c; b+=d; a+b */
float fo_vec4_dot(const float a[4], const float b[4]) {
    float x = a[0] * b[0];
    float y = a[1] * b[1];
    float z = a[2] * b[2];
    float w = a[3] * b[3];
    x += z;
    y += w;
    return x + y;
}

/* nalgebra Quaternion * Quaternion (Hamilton); storage (i,j,k,w). */
void fo_quat_mul(const float a[4], const float b[4], float out[4]) {
    float w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    float i = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    float j = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
    float k = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
    out[0] = i; out[1] = j; out[2] = k; out[3] = w;
}

/* UnitQuaternion::from_axis_angle(unit axis e_axis, angle):
 *   (s, c) = sin_cos(angle / 2);  q = (w = c, v = axis * s) */
void fo_quat_from_axis_angle(int axis, float angle, float out[4]) {
    float half = angle / 2.0f;
    float s = sinf(half), c = cosf(half);
    float e[3] = { 0.0f, 0.0f, 0.0f };
    e[axis] = 1.0f;
    out[0] = e[0] * s; out[1] = e[1] * s; out[2] = e[2] * s; out[3] = c;
}

/* Unit::new_normalize / normalize_mut: n = sqrt(dot(q,q)); q_i / n */
void fo_quat_normalize(const float q[4], float out[4]) {
    float n = sqrtf(fo_vec4_dot(q, q));
    out[0] = q[0] / n; out[1] = q[1] / n; out[2] = q[2] / n; out[3] = q[3] / n;
}

/* UnitQuaternion::to_rotation_matrix (call site scene/transform.rs:424-425).
 * Output column-major 3x3: out[col*3+row], matching `pr[..]`/`r[..]` indexing there. */
void fo_quat_to_mat3(const float q[4], float out[9]) {
    float i = q[0], j = q[1], k = q[2], w = q[3];
    float ww = w * w, ii = i * i, jj = j * j, kk = k * k;
    float ij = i * j * 2.0f, wk = w * k * 2.0f, wj = w * j * 2.0f;
    float ik = i * k * 2.0f, jk = j * k * 2.0f, wi = w * i * 2.0f;
    /* row-major constructor arguments m11 m12 m13 / m21 m22 m23 / m31 m32 m33 */
    float m11 = ww + ii - jj - kk, m12 = ij - wk, m13 = wj + ik;
    float m21 = wk + ij, m22 = ww - ii + jj - kk, m23 = jk - wi;
    float m31 = ik - wj, m32 = wi + jk, m33 = ww - ii - jj + kk;
    out[0] = m11; out[1] = m21; out[2] = m31;
    out[3] = m12; out[4] = m22; out[5] = m32;
    out[6] = m13; out[7] = m23; out[8] = m33;
}

/* nalgebra Vector::lerp: self * (1 - t) + rhs * t   (NOT lerpf's a+(b-a)*t) */
void fo_vec_lerp(const float* a, const float* b, float t, int n, float* out) {
    float omt = 1.0f - t;
    for (int i = 0; i < n; ++i) out[i] = a[i] * omt + b[i] * t;
}

/* fyrox-animation/src/value.rs:449-459  nlerp(a, b, w):
 *   if a.dot(b) < 0 { a = -a };  a.nlerp(b, w) = normalize(a*(1-w) + b*w) */
void fo_quat_nlerp_shortest(const float a_in[4], const float b[4], float w, float out[4]) {
    float a[4] = { a_in[0], a_in[1], a_in[2], a_in[3] };
    if (fo_vec4_dot(a, b) < 0.0f) {
        a[0] = -a[0]; a[1] = -a[1]; a[2] = -a[2]; a[3] = -a[3];
    }
    float l[4];
    fo_vec_lerp(a, b, w, 4, l);
    fo_quat_normalize(l, out);
}

/* ======================================================================== */
/* fyrox-math                                                                */
/* ======================================================================== */

/* fyrox-math/src/lib.rs:206-208 */
float fo_lerpf(float a, float b, float t) { return a + (b - a) * t; }

/* fyrox-math/src/lib.rs:212-221 */
float fo_cubicf(float p0, float p1, float t, float m0, float m1) {
    float t2 = t * t;
    float t3 = t2 * t;
    float scale = fabsf(p1 - p0);
    return (2.0f * t3 - 3.0f * t2 + 1.0f) * p0
         + (t3 - 2.0f * t2 + t) * m0 * scale
         + (-2.0f * t3 + 3.0f * t2) * p1
         + (t3 - t2) * m1 * scale;
}

/* fyrox-math/src/lib.rs:179-203 */
float fo_wrapf(float n, float min_limit, float max_limit) {
    if (n >= min_limit && n <= max_limit) return n;
    if (max_limit == 0.0f && min_limit == 0.0f) return 0.0f;
    max_limit -= min_limit;
    float offset = min_limit;
    min_limit = 0.0f;
    n -= offset;
    float num_of_max = floorf(fabsf(n / max_limit));
    if (n >= max_limit) {
        n -= num_of_max * max_limit;
    } else if (n < min_limit) {
        n += (num_of_max + 1.0f) * max_limit;
    }
    return n + offset;
}

/* fyrox-math/src/curve.rs:25-31 */
float fo_stepf(float p0, float p1, float t) { return (t == 1.0f) ? p1 : p0; }

/* fyrox-math/src/lib.rs:725-740 */
void fo_quat_from_euler(const float e[3], int order, float out[4]) {
    float qx[4], qy[4], qz[4], t[4];
    fo_quat_from_axis_angle(0, e[0], qx);
    fo_quat_from_axis_angle(1, e[1], qy);
    fo_quat_from_axis_angle(2, e[2], qz);
    switch (order) {
    default:
    case 0: fo_quat_mul(qz, qy, t); fo_quat_mul(t, qx, out); break; /* XYZ => qz*qy*qx */
    case 1: fo_quat_mul(qy, qz, t); fo_quat_mul(t, qx, out); break; /* XZY => qy*qz*qx */
    case 2: fo_quat_mul(qx, qz, t); fo_quat_mul(t, qy, out); break; /* YZX => qx*qz*qy */
    case 3: fo_quat_mul(qz, qx, t); fo_quat_mul(t, qy, out); break; /* YXZ => qz*qx*qy */
    case 4: fo_quat_mul(qy, qx, t); fo_quat_mul(t, qz, out); break; /* ZXY => qy*qx*qz */
    case 5: fo_quat_mul(qx, qy, t); fo_quat_mul(t, qz, out); break; /* ZYX => qx*qy*qz */
    }
}

/* fyrox-math/src/curve.rs:87-132  interpolate(): dispatch on the LEFT key's kind.
 * Cubic left key contributes its right_tangent; a Cubic right key its left_tangent,
 * any other right kind contributes 0.0. */
float fo_key_interpolate(float lv, int lkind, float l_right_tangent,
                         float rv, int rkind, float r_left_tangent, float t) {
    switch (lkind) {
    case FO_KEY_CONSTANT: return fo_stepf(lv, rv, t);
    case FO_KEY_LINEAR:   return fo_lerpf(lv, rv, t);
    default:
        if (rkind == FO_KEY_CUBIC) return fo_cubicf(lv, rv, t, l_right_tangent, r_left_tangent);
        return fo_cubicf(lv, rv, t, l_right_tangent, 0.0f);
    }
}

static float fo_interp_keys(const fo_curve* c, size_t l, size_t r, float location) {
    float t = (location - c->location[l]) / (c->location[r] - c->location[l]);
    float lrt = (c->kind[l] == FO_KEY_CUBIC) ? c->right_tangent[l] : 0.0f;
    float rlt = (c->kind[r] == FO_KEY_CUBIC) ? c->left_tangent[r] : 0.0f;
    return fo_key_interpolate(c->value[l], c->kind[l], lrt, c->value[r], c->kind[r], rlt, t);
}

/* fyrox-math/src/curve.rs:254-314  Curve::fetch_at / value_at */
float fo_curve_value_at(const fo_curve* c, float location, size_t* hint) {
    size_t n = c->n_keys;
    if (n == 0) return 0.0f;
    if (location <= c->location[0]) { *hint = 0; return c->value[0]; }
    if (location >= c->location[n - 1]) { *hint = n - 1; return c->value[n - 1]; }
    /* hinted span [hint-1, hint) */
    size_t h = *hint;
    size_t hl = h > 0 ? h - 1 : 0;
    if (hl < n && h < n) {
        if (location >= c->location[hl] && location < c->location[h])
            return fo_interp_keys(c, hl, h, location);
    }
    /* partition_point(|k| k.location < location) */
    size_t lo = 0, hi = n;
    while (lo < hi) {
        size_t mid = lo + (hi - lo) / 2;
        if (c->location[mid] < location) lo = mid + 1; else hi = mid;
    }
    *hint = lo;
    size_t l = lo > 0 ? lo - 1 : 0;
    return fo_interp_keys(c, l, lo, location);
}

/* fyrox-animation/src/container.rs:287-297 (+ :182-282) */
int fo_track_fetch(const fo_curve* curves, uint32_t n_curves, int kind, float time,
                   size_t hints[4], float out[4]) {
    switch (kind) {
    case FO_KIND_REAL:
        if (n_curves < 1) return 0;
        out[0] = fo_curve_value_at(&curves[0], time, &hints[0]);
        return 1;
    case FO_KIND_VEC2:
        if (n_curves < 2) return 0;
        for (int i = 0; i < 2; ++i) out[i] = fo_curve_value_at(&curves[i], time, &hints[i]);
        return 2;
    case FO_KIND_VEC3:
        if (n_curves < 3) return 0;
        for (int i = 0; i < 3; ++i) out[i] = fo_curve_value_at(&curves[i], time, &hints[i]);
        return 3;
    case FO_KIND_VEC4:
        if (n_curves < 4) return 0;
        for (int i = 0; i < 4; ++i) out[i] = fo_curve_value_at(&curves[i], time, &hints[i]);
        return 4;
    case FO_KIND_QUAT_EULER: {
        if (n_curves < 3) return 0;
        float e[3];
        for (int i = 0; i < 3; ++i) e[i] = fo_curve_value_at(&curves[i], time, &hints[i]);
        fo_quat_from_euler(e, 0, out);
        return 4;
    }
    case FO_KIND_QUAT: {
        if (n_curves < 4) return 0;
        float q[4];
        for (int i = 0; i < 4; ++i) q[i] = fo_curve_value_at(&curves[i], time, &hints[i]);
        /* UnitQuaternion::from_quaternion(Quaternion::new(w, x, y, z)) : storage (x,y,z,w) */
        fo_quat_normalize(q, out);
        return 4;
    }
    default: return 0;
    }
}

/* ======================================================================== */
/* scene: local transform, hierarchy, palette                                */
/* ======================================================================== */

void fo_transform_default(fo_transform* t) {
    memset(t, 0, sizeof *t);
    t->local_rotation[3] = 1.0f;
    t->pre_rotation[3] = 1.0f;
    t->local_scale[0] = t->local_scale[1] = t->local_scale[2] = 1.0f;
    t->post_rotation_matrix[0] = t->post_rotation_matrix[4] = t->post_rotation_matrix[8] = 1.0f;
}

/* fyrox-impl/src/scene/transform.rs:421-540, expression for expression. */
void fo_calculate_local_transform(const fo_transform* tr, float out[16]) {
    const float* por = tr->post_rotation_matrix;
    float pr[9], r[9];
    fo_quat_to_mat3(tr->pre_rotation, pr);
    fo_quat_to_mat3(tr->local_rotation, r);
    float sx = tr->local_scale[0], sy = tr->local_scale[1], sz = tr->local_scale[2];
    float tx = tr->local_position[0], ty = tr->local_position[1], tz = tr->local_position[2];
    float rpx = tr->rotation_pivot[0], rpy = tr->rotation_pivot[1], rpz = tr->rotation_pivot[2];
    float rox = tr->rotation_offset[0], roy = tr->rotation_offset[1], roz = tr->rotation_offset[2];
    float spx = tr->scaling_pivot[0], spy = tr->scaling_pivot[1], spz = tr->scaling_pivot[2];
    float sox = tr->scaling_offset[0], soy = tr->scaling_offset[1], soz = tr->scaling_offset[2];

    float a0 = pr[0] * r[0] + pr[3] * r[1] + pr[6] * r[2];
    float a1 = pr[1] * r[0] + pr[4] * r[1] + pr[7] * r[2];
    float a2 = pr[2] * r[0] + pr[5] * r[1] + pr[8] * r[2];
    float a3 = pr[0] * r[3] + pr[3] * r[4] + pr[6] * r[5];
    float a4 = pr[1] * r[3] + pr[4] * r[4] + pr[7] * r[5];
    float a5 = pr[2] * r[3] + pr[5] * r[4] + pr[8] * r[5];
    float a6 = pr[0] * r[6] + pr[3] * r[7] + pr[6] * r[8];
    float a7 = pr[1] * r[6] + pr[4] * r[7] + pr[7] * r[8];
    float a8 = pr[2] * r[6] + pr[5] * r[7] + pr[8] * r[8];
    float f0 = por[0] * a0 + por[1] * a3 + por[2] * a6;
    float f1 = por[0] * a1 + por[1] * a4 + por[2] * a7;
    float f2 = por[0] * a2 + por[1] * a5 + por[2] * a8;
    float f3 = por[3] * a0 + por[4] * a3 + por[5] * a6;
    float f4 = por[3] * a1 + por[4] * a4 + por[5] * a7;
    float f5 = por[3] * a2 + por[4] * a5 + por[5] * a8;
    float f6 = por[6] * a0 + por[7] * a3 + por[8] * a6;
    float f7 = por[6] * a1 + por[7] * a4 + por[8] * a7;
    float f8 = por[6] * a2 + por[7] * a5 + por[8] * a8;
    float m0 = sx * f0, m1 = sx * f1, m2 = sx * f2, m3 = 0.0f;
    float m4 = sy * f3, m5 = sy * f4, m6 = sy * f5, m7 = 0.0f;
    float m8 = sz * f6, m9 = sz * f7, m10 = sz * f8, m11 = 0.0f;
    float k0 = spx * f0, k1 = spy * f3, k2 = spz * f6;
    float m12 = rox + rpx + tx - rpx * f0 - rpy * f3 - rpz * f6
        + sox * f0 + k0 + soy * f3 + k1 + soz * f6 + k2 - sx * k0 - sy * k1 - sz * k2;
    float k3 = spx * f1, k4 = spy * f4, k5 = spz * f7;
    float m13 = roy + rpy + ty - rpx * f1 - rpy * f4 - rpz * f7
        + sox * f1 + k3 + soy * f4 + k4 + soz * f7 + k5 - sx * k3 - sy * k4 - sz * k5;
    float k6 = spx * f2, k7 = spy * f5, k8 = spz * f8;
    float m14 = roz + rpz + tz - rpx * f2 - rpy * f5 - rpz * f8
        + sox * f2 + k6 + soy * f5 + k7 + soz * f8 + k8 - sx * k6 - sy * k7 - sz * k8;
    float m15 = 1.0f;
    /* Matrix4::new(m0,m4,m8,m12, m1,m5,m9,m13, m2,m6,m10,m14, m3,m7,m11,m15) is
     * row-major argument order => column-major storage is m0..m15 in order. */
    out[0] = m0; out[1] = m1; out[2] = m2; out[3] = m3;
    out[4] = m4; out[5] = m5; out[6] = m6; out[7] = m7;
    out[8] = m8; out[9] = m9; out[10] = m10; out[11] = m11;
    out[12] = m12; out[13] = m13; out[14] = m14; out[15] = m15;
}

/* fyrox-impl/src/scene/graph/mod.rs:1199-1241: global = parent.global * local,
 * root's parent = identity (an invalid parent handle). Requires parent[i] < i. */
void fo_update_global_transforms(const float* local, const int32_t* parent, uint32_t n, float* global) {
    float ident[16];
    fo_mat4_identity(ident);
    for (uint32_t i = 0; i < n; ++i) {
        const float* pg = (parent[i] >= 0) ? &global[(size_t)parent[i] * 16] : ident;
        fo_mat4_mul(pg, &local[(size_t)i * 16], &global[(size_t)i * 16]);
    }
}

/* fyrox-impl/src/scene/mesh/mod.rs:781-793 (== :492-499):
 * palette[i] = bone.global_transform() * bone.inv_bind_pose_transform() */
void fo_palette(const float* global, const float* inv_bind, uint32_t n, float* out) {
    for (uint32_t i = 0; i < n; ++i)
        fo_mat4_mul(&global[(size_t)i * 16], &inv_bind[(size_t)i * 16], &out[(size_t)i * 16]);
}

/* ======================================================================== */
/* LBS                                                                       */
/* ======================================================================== */

/* One vertex.
 * position: fyrox-impl/src/scene/mesh/mod.rs:501-522
 *     position = 0; for k in 0..4: position += M[idx_k].transform_point(p).scale(w_k)
 * normal / tangent.xyz: standard.shader:192-200
 *     local = 0; local += mat3(M_k) * v * w_k   (no inverse-transpose, no normalise)
 * tangent.w passes through (standard.shader:212 uses vertexTangent.w afterwards). */
static inline void fo_lbs_vertex(const float* p, const float* nv, const float* tv,
                                 const float* w, const uint8_t* idx, const float* palette,
                                 float* op, float* on, float* ot) {
    float ap[3] = { 0.0f, 0.0f, 0.0f };
    float an[3] = { 0.0f, 0.0f, 0.0f };
    float at[3] = { 0.0f, 0.0f, 0.0f };
    for (int k = 0; k < 4; ++k) {
        const float* m = &palette[(size_t)idx[k] * 16];
        float wk = w[k];
        float r[3];
        if (op) {
            fo_mat4_transform_point(m, p, r);
            ap[0] += r[0] * wk; ap[1] += r[1] * wk; ap[2] += r[2] * wk;
        }
        if (on) {
            fo_mat3_of_mat4_mul_vec(m, nv, r);
            an[0] += r[0] * wk; an[1] += r[1] * wk; an[2] += r[2] * wk;
        }
        if (ot) {
            fo_mat3_of_mat4_mul_vec(m, tv, r);
            at[0] += r[0] * wk; at[1] += r[1] * wk; at[2] += r[2] * wk;
        }
    }
    if (op) { op[0] = ap[0]; op[1] = ap[1]; op[2] = ap[2]; }
    if (on) { on[0] = an[0]; on[1] = an[1]; on[2] = an[2]; }
    if (ot) { ot[0] = at[0]; ot[1] = at[1]; ot[2] = at[2]; ot[3] = tv[3]; }
}

static int fo_check_indices(uint32_t n_verts, const uint8_t* indices, uint32_t n_bones) {
    if (n_bones >= 256) return 0;
    for (size_t i = 0; i < (size_t)n_verts * 4; ++i)
        if (indices[i] >= n_bones) return -1; /* Rust would panic on the slice index */
    return 0;
}

int fo_lbs_skin(uint32_t n_verts, const float* pos, const float* nrm, const float* tan,
                const float* weights, const uint8_t* indices,
                const float* palette, uint32_t n_bones,
                float* out_pos, float* out_nrm, float* out_tan) {
    if (fo_check_indices(n_verts, indices, n_bones)) return -1;
    if (!nrm) out_nrm = NULL;
    if (!tan) out_tan = NULL;
    for (size_t v = 0; v < n_verts; ++v) {
        fo_lbs_vertex(&pos[v * 3], nrm ? &nrm[v * 3] : NULL, tan ? &tan[v * 4] : NULL,
                      &weights[v * 4], &indices[v * 4], palette,
                      out_pos ? &out_pos[v * 3] : NULL,
                      out_nrm ? &out_nrm[v * 3] : NULL,
                      out_tan ? &out_tan[v * 4] : NULL);
    }
    return 0;
}

int fo_omp_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* Same arithmetic, OpenMP over vertices.  NOT present in the reference (its loop is
 * serial and there is no rayon on this path); a generous CPU upper bound only. */
int fo_lbs_skin_omp(uint32_t n_verts, const float* pos, const float* nrm, const float* tan,
                    const float* weights, const uint8_t* indices,
                    const float* palette, uint32_t n_bones,
                    float* out_pos, float* out_nrm, float* out_tan, int n_threads) {
    if (fo_check_indices(n_verts, indices, n_bones)) return -1;
    if (!nrm) out_nrm = NULL;
    if (!tan) out_tan = NULL;
    (void)n_threads;
    long long nv = (long long)n_verts;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(n_threads > 0 ? n_threads : omp_get_max_threads())
#endif
    for (long long v = 0; v < nv; ++v) {
        fo_lbs_vertex(&pos[v * 3], nrm ? &nrm[v * 3] : NULL, tan ? &tan[v * 4] : NULL,
                      &weights[v * 4], &indices[v * 4], palette,
                      out_pos ? &out_pos[v * 3] : NULL,
                      out_nrm ? &out_nrm[v * 3] : NULL,
                      out_tan ? &out_tan[v * 4] : NULL);
    }
    return 0;
}

static float fo_rd_f32le(const uint8_t* p) {
    uint32_t u = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
    float f;
    memcpy(&f, &u, 4);
    return f;
}

/* fyrox-impl/src/scene/mesh/mod.rs:470-526 skinned branch; per-field little-endian
 * reads as VertexReadTrait does (scene/mesh/buffer.rs:1279-1321); AABB default and
 * add_point per fyrox-math/src/aabb.rs:33-38,86-104. */
uint32_t fo_accurate_world_bounding_box(const uint8_t* aos, uint32_t n_verts, uint32_t stride,
                                        int off_pos, int off_weights, int off_indices,
                                        const float* palette, uint32_t n_bones, float aabb[6]) {
    aabb[0] = aabb[1] = aabb[2] = FLT_MAX;
    aabb[3] = aabb[4] = aabb[5] = -FLT_MAX;
    if (off_pos < 0 || off_weights < 0 || off_indices < 0) return 0; /* `else { break }` on first vertex */
    uint32_t v = 0;
    for (; v < n_verts; ++v) {
        const uint8_t* base = aos + (size_t)v * stride;
        float p[3], w[4];
        uint8_t idx[4];
        for (int i = 0; i < 3; ++i) p[i] = fo_rd_f32le(base + off_pos + 4 * i);
        for (int i = 0; i < 4; ++i) w[i] = fo_rd_f32le(base + off_weights + 4 * i);
        for (int i = 0; i < 4; ++i) idx[i] = base[off_indices + i];
        for (int i = 0; i < 4; ++i) if (idx[i] >= n_bones) return v; /* panic in Rust */
        float o[3];
        fo_lbs_vertex(p, NULL, NULL, w, idx, palette, o, NULL, NULL);
        for (int i = 0; i < 3; ++i) {
            if (o[i] < aabb[i]) aabb[i] = o[i];
            if (o[i] > aabb[3 + i]) aabb[3 + i] = o[i];
        }
    }
    return v;
}

/* ======================================================================== */
/* Blend shapes                                                              */
/* ======================================================================== */

/* IEEE binary16 -> binary32, exact (what texelFetch on an RGB16F texture returns). */
float fo_half_to_float(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1fu, man = h & 0x3ffu, bits;
    if (exp == 0) {
        if (man == 0) { bits = sign; }
        else { /* subnormal: normalise */
            int e = -1;
            do { ++e; man <<= 1; } while (!(man & 0x400u));
            bits = sign | (uint32_t)(127 - 15 - e) << 23 | (man & 0x3ffu) << 13;
        }
    } else if (exp == 31) { bits = sign | 0x7f800000u | man << 13; }
    else { bits = sign | (exp + 127 - 15) << 23 | man << 13; }
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

/* fyrox-material/src/shader/standard/opengl/standard.shader:167-173 with S_FetchBlendShapeOffsets
 * (fyrox-graphics-gl/src/shaders/shared.glsl:371-378) over the RGB16F volume that
 * BlendShapesContainer::from_lists packs (fyrox-impl/src/scene/mesh/surface.rs:116-217):
 *   texel (3*v + {0,1,2}) of layer i = position / normal / tangent offset of vertex v, shape i;
 *   for i in 0..n_shapes: p += off.position * w[i]; n += off.normal * w[i]; t.xyz += off.tangent * w[i]
 * (GLSL may contract a*b+c; this restatement does not -- parity unpinned, see the header).
 * storage: n_shapes planes of plane_vertices * 9 halfs.  nrm / tan (and their outputs) may be NULL. */
void fo_apply_blend_shapes(uint32_t n_verts, const float* pos, const float* nrm, const float* tan,
                           const uint16_t* storage, uint32_t plane_vertices, uint32_t n_shapes,
                           const float* weights, float* out_pos, float* out_nrm, float* out_tan) {
    for (uint32_t v = 0; v < n_verts; ++v) {
        float p[3] = { pos[v * 3], pos[v * 3 + 1], pos[v * 3 + 2] };
        float n[3] = { 0, 0, 0 }, t[4] = { 0, 0, 0, 0 };
        if (nrm) memcpy(n, nrm + (size_t)v * 3, 12);
        if (tan) memcpy(t, tan + (size_t)v * 4, 16);
        for (uint32_t i = 0; i < n_shapes; ++i) {
            const uint16_t* rec = storage + ((size_t)i * plane_vertices + v) * 9;
            float w = weights[i];
            for (int k = 0; k < 3; ++k) {
                p[k] = p[k] + fo_half_to_float(rec[k]) * w;
                n[k] = n[k] + fo_half_to_float(rec[3 + k]) * w;
                t[k] = t[k] + fo_half_to_float(rec[6 + k]) * w;
            }
        }
        memcpy(out_pos + (size_t)v * 3, p, 12);
        if (nrm && out_nrm) memcpy(out_nrm + (size_t)v * 3, n, 12);
        if (tan && out_tan) memcpy(out_tan + (size_t)v * 4, t, 16);
    }
}

/* ---------------------------------------------------------------------------------------
 * glTF importer: curve simplification (fyrox-impl/src/resource/gltf/simplify.rs:39-140), run on every imported curve
 * (gltf/animation.rs:155-163, :292) with the binding's epsilon / max_step (animation.rs:50-65): decides which keys a
 * glTF-built track has.  Restated function by function, recursion included.
 * ------------------------------------------------------------------------------------- */
static void fo_find_points_in_span(const float* x, const float* y, uint8_t* keep, size_t start, size_t end, float epsilon) {
    if (end <= start + 1) return;                                   /* simplify.rs:114-116 */
    const float x0 = x[start], y0 = y[start];
    const float slope = (y[end] - y0) / (x[end] - x0);              /* :119 */
    size_t far_index = 0;
    float far_dist = 0.0f;
    for (size_t i = start + 1; i < end; ++i) {                      /* :122-129 */
        const float y_line = y0 + slope * (x[i] - x0);
        const float dist = fabsf(y[i] - y_line);
        if (far_dist < dist) { far_dist = dist; far_index = i; }
    }
    if (far_index == 0 || far_dist < epsilon) return;               /* :131-133 */
    keep[far_index] = 1;
    fo_find_points_in_span(x, y, keep, start, far_index, epsilon);
    fo_find_points_in_span(x, y, keep, far_index, end, epsilon);
}

static size_t fo_find_step(size_t start, const float* y, size_t n, const uint8_t* keep, float max_step) {   /* :86-102 */
    const float start_y = y[start];
    for (size_t i = start + 1; i < n; ++i) {
        const float step = fabsf(y[i] - start_y);
        if (step > max_step) return (i - 1 > start + 1) ? i - 1 : start + 1;
        else if (keep[i]) return i;
    }
    return n - 1;
}

/* find_important_points (simplify.rs:39-66): indices of the kept points into out (room for n); returns their number.
 * max_step: INFINITY = no step limit (is_finite() false). */
uint32_t fo_find_important_points(const float* x, const float* y, uint32_t n, float epsilon, float max_step, uint32_t* out) {
    if (n == 0) return 0;
    uint8_t* keep = (uint8_t*)calloc(n, 1);
    const size_t end = (size_t)n - 1;
    keep[0] = 1;
    keep[end] = 1;
    fo_find_points_in_span(x, y, keep, 0, end, epsilon);
    if (isfinite(max_step)) {                                       /* limit_step_size, :69-81 */
        size_t i = 1;
        while (i < end) {
            if (keep[i]) { i += 1; }
            else {
                const size_t next = fo_find_step(i - 1, y, n, keep, max_step);
                keep[next] = 1;
                i = (next + 1 > i + 1) ? next + 1 : i + 1;
            }
        }
    }
    uint32_t m = 0;
    for (uint32_t i = 0; i < n; ++i) if (keep[i]) out[m++] = i;
    if (m == 2 && fabsf(y[out[0]] - y[out[1]]) < epsilon) m = 1;    /* :62-64 */
    free(keep);
    return m;
}

/* ---------------------------------------------------------------------------------------
 * BlendSpace::triangulate (fyrox-animation/src/machine/node/blendspace.rs:416-447).  The reference inserts the points,
 * in order, into a spade::DelaunayTriangulation (crate `spade` 2.x -- not vendored, not buildable here) and lists every
 * inner face as the origins of its three edges.  Restated from the published algorithm + the one vector the reference
 * holds (blendspace.rs:455-484: the unit square -> [2, 0, 1], [3, 0, 2]):
 *   * the triangles are the Delaunay triangulation of the points; for co-circular points the diagonal that exists when
 *     the later point arrives stays (insertion order decides: incremental insertion, strict in-circle test);
 *   * each triangle is counter-clockwise and starts at its NEWEST point (the face was made by that point's insertion),
 *     triangles are listed by newest point, then by the other two.
 * The second rule is what the fixture shows; spade's face order for larger inputs is NOT pinned by anything in the
 * reference ("parity unpinned" beyond the fixture).  Here: Bowyer-Watson, written as cavity re-triangulation over an
 * explicit triangle list with a far-away bounding triangle; all predicates in double on f32 inputs.
 * Returns the number of triangles (0 for fewer than three points), -1 when a coordinate is not finite (spade's insert
 * fails: the reference returns false with no triangles).  A point equal to an earlier one adds nothing.
 * ------------------------------------------------------------------------------------- */
static double fo_orient2d(const double* a, const double* b, const double* c) {
    return (b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0]);
}
static int fo_in_circle(const double* a, const double* b, const double* c, const double* p) {   /* a, b, c counter-clockwise */
    const double ax = a[0] - p[0], ay = a[1] - p[1], bx = b[0] - p[0], by = b[1] - p[1], cx = c[0] - p[0], cy = c[1] - p[1];
    const double det = (ax * ax + ay * ay) * (bx * cy - cx * by) - (bx * bx + by * by) * (ax * cy - cx * ay) + (cx * cx + cy * cy) * (ax * by - bx * ay);
    return det > 0.0;
}

int32_t fo_blend_space_triangulate(const float* xy, uint32_t n, uint32_t* out_tri, uint32_t capacity) {
    for (uint32_t i = 0; i < 2 * n; ++i) if (!isfinite(xy[i])) return -1;
    if (n < 3) return 0;
    double lo[2] = {xy[0], xy[1]}, hi[2] = {xy[0], xy[1]};
    for (uint32_t i = 0; i < n; ++i)
        for (int k = 0; k < 2; ++k) { if (xy[2 * i + k] < lo[k]) lo[k] = xy[2 * i + k]; if (xy[2 * i + k] > hi[k]) hi[k] = xy[2 * i + k]; }
    double ext = (hi[0] - lo[0] > hi[1] - lo[1]) ? hi[0] - lo[0] : hi[1] - lo[1];
    if (ext <= 0.0) ext = 1.0;
    const double cx = 0.5 * (lo[0] + hi[0]), cy = 0.5 * (lo[1] + hi[1]), R = ext * 1.0e4;
    double* P = (double*)malloc(sizeof(double) * 2 * ((size_t)n + 3));
    for (uint32_t i = 0; i < n; ++i) { P[2 * i] = xy[2 * i]; P[2 * i + 1] = xy[2 * i + 1]; }
    P[2 * n] = cx - 2.0 * R; P[2 * n + 1] = cy - R;              /* the bounding triangle: vertices n, n + 1, n + 2 (ccw) */
    P[2 * n + 2] = cx + 2.0 * R; P[2 * n + 3] = cy - R;
    P[2 * n + 4] = cx; P[2 * n + 5] = cy + 2.0 * R;
    size_t cap_t = 16 * ((size_t)n + 4), nt = 0;
    uint32_t* T = (uint32_t*)malloc(sizeof(uint32_t) * 3 * cap_t);
    uint32_t* E = (uint32_t*)malloc(sizeof(uint32_t) * 2 * 3 * cap_t);
    T[0] = n; T[1] = n + 1; T[2] = n + 2; nt = 1;
    for (uint32_t p = 0; p < n; ++p) {
        int dup = 0;
        for (uint32_t q = 0; q < p && !dup; ++q) dup = (xy[2 * q] == xy[2 * p] && xy[2 * q + 1] == xy[2 * p + 1]);
        if (dup) continue;
        /* the cavity: every triangle whose circumcircle strictly contains p; its boundary edges get a triangle with p */
        size_t ne = 0, keep = 0;
        for (size_t t = 0; t < nt; ++t) {
            const uint32_t a = T[3 * t], b = T[3 * t + 1], c = T[3 * t + 2];
            if (fo_in_circle(&P[2 * a], &P[2 * b], &P[2 * c], &P[2 * p])) {
                const uint32_t e[3][2] = {{a, b}, {b, c}, {c, a}};
                for (int k = 0; k < 3; ++k) { E[2 * ne] = e[k][0]; E[2 * ne + 1] = e[k][1]; ++ne; }
            } else {
                T[3 * keep] = a; T[3 * keep + 1] = b; T[3 * keep + 2] = c; ++keep;
            }
        }
        nt = keep;
        for (size_t i = 0; i < ne; ++i) {
            int shared = 0;                                         /* an edge inside the cavity appears twice, reversed */
            for (size_t j = 0; j < ne && !shared; ++j) shared = (j != i && E[2 * j] == E[2 * i + 1] && E[2 * j + 1] == E[2 * i]);
            if (shared) continue;
            const uint32_t a = E[2 * i], b = E[2 * i + 1];
            if (fo_orient2d(&P[2 * a], &P[2 * b], &P[2 * p]) <= 0.0) continue;   /* p on the edge's line: a sliver of zero area */
            if (nt < cap_t) { T[3 * nt] = p; T[3 * nt + 1] = a; T[3 * nt + 2] = b; ++nt; }
        }
    }
    /* inner faces: no bounding vertex; newest point first, counter-clockwise; listed by (newest, second, third) */
    size_t m = 0;
    for (size_t t = 0; t < nt; ++t) {
        uint32_t v[3] = {T[3 * t], T[3 * t + 1], T[3 * t + 2]};
        if (v[0] >= n || v[1] >= n || v[2] >= n) continue;
        const int top = (v[0] > v[1] && v[0] > v[2]) ? 0 : (v[1] > v[2] ? 1 : 2);
        T[3 * m] = v[top]; T[3 * m + 1] = v[(top + 1) % 3]; T[3 * m + 2] = v[(top + 2) % 3]; ++m;
    }
    for (size_t i = 1; i < m; ++i) {                                /* insertion sort, lexicographic */
        uint32_t k[3] = {T[3 * i], T[3 * i + 1], T[3 * i + 2]};
        size_t j = i;
        while (j > 0 && (T[3 * (j - 1)] > k[0] || (T[3 * (j - 1)] == k[0] && (T[3 * (j - 1) + 1] > k[1] || (T[3 * (j - 1) + 1] == k[1] && T[3 * (j - 1) + 2] > k[2]))))) {
            T[3 * j] = T[3 * (j - 1)]; T[3 * j + 1] = T[3 * (j - 1) + 1]; T[3 * j + 2] = T[3 * (j - 1) + 2]; --j;
        }
        T[3 * j] = k[0]; T[3 * j + 1] = k[1]; T[3 * j + 2] = k[2];
    }
    for (size_t i = 0; i < m && i < capacity; ++i) { out_tri[3 * i] = T[3 * i]; out_tri[3 * i + 1] = T[3 * i + 1]; out_tri[3 * i + 2] = T[3 * i + 2]; }
    free(P); free(T); free(E);
    return (int32_t)m;
}
