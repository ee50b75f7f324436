// anim_leaves.h -- the decision-making leaves of the pose kernels as __host__ __device__ code: the SAME functions run in the
// kernels (anim_kernels.hip) and, compiled for the host, behind the fyx_debug_* entry points the CPU tests call against the
// oracle (tests/test_span_value_at.py, tests/test_fold_classifier.py) -- so what the CPU suite checks is the code the GPU
// runs, not a model of it.  Both passes are compiled with -ffp-contract=off: the arithmetic is the same unfused IEEE f32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../../include/fyrox_hip.h"
#include "fyx_internal.h"

#define FYX_HD __host__ __device__ __forceinline__

namespace fyx {

typedef float f4 __attribute__((ext_vector_type(4)));

FYX_HD uint32_t f2u(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_uint(v);
#else
    uint32_t u;
    memcpy(&u, &v, 4);
    return u;
#endif
}
FYX_HD float u2f(uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float v;
    memcpy(&v, &u, 4);
    return v;
#endif
}
FYX_HD float absf_(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return fabsf(v);
#else
    return __builtin_fabsf(v);
#endif
}

// ---------------------------------------------------------------------------------------
// fyrox-math leaves (fyrox-math/src/lib.rs:206-221)
// ---------------------------------------------------------------------------------------
FYX_HD float lerpf_(float a, float b, float t) { return a + (b - a) * t; }

FYX_HD float cubicf_(float p0, float p1, float t, float m0, float m1) {
    const float t2 = t * t;
    const float t3 = t2 * t;
    const float scale = absf_(p1 - p0);
    return (2.0f * t3 - 3.0f * t2 + 1.0f) * p0 + (t3 - 2.0f * t2 + t) * m0 * scale +
           (-2.0f * t3 + 3.0f * t2) * p1 + (t3 - t2) * m1 * scale;
}

// CurveKey::interpolate (curve.rs:87-132): dispatch on the LEFT key's kind.  la / ra = {value, kind bits, left tangent, right tangent}.
FYX_HD float interpolate_loaded(float ll, float rl, f4 la, f4 ra, float location) {
    const float t = (location - ll) / (rl - ll);
    const uint32_t lk = f2u(la.y), rk = f2u(ra.y);
    if (lk == FYX_KEY_CONSTANT) return t == 1.0f ? ra.x : la.x;
    if (lk == FYX_KEY_LINEAR) return lerpf_(la.x, ra.x, t);
    return cubicf_(la.x, ra.x, t, la.w, rk == FYX_KEY_CUBIC ? ra.z : 0.0f);
}

// The same dispatch on a SPAN RECORD's part for one curve: cv = {left key's value, right key's value, left key's right tangent,
// right key's left tangent if the right key is cubic else 0} -- the four numbers CurveKey::interpolate reads of the two keys -- and
// the left key's kind (the builder of the records makes the selection of the last one: same values into the same operations).
FYX_HD float interpolate_span(float ll, float rl, uint32_t lk, f4 cv, float location) {
    const float t = (location - ll) / (rl - ll);
    if (lk == FYX_KEY_CONSTANT) return t == 1.0f ? cv.y : cv.x;
    if (lk == FYX_KEY_LINEAR) return lerpf_(cv.x, cv.y, t);
    return cubicf_(cv.x, cv.y, t, cv.z, cv.w);
}
// stride of a track's span records in f4 (header + one part per curve), and curve c's kind out of the header's third word
FYX_HD uint32_t span_stride(uint32_t need) { return need + 1u; }
FYX_HD uint32_t span_kind(f4 header, uint32_t c) { return (f2u(header.z) >> (8u * c)) & 0xffu; }

// ---------------------------------------------------------------------------------------
// Curve::value_at (curve.rs:254-314) for the three or four curves of ONE track at once, on the track's span records
// (`sp`: LDS in the crowd sampler, plain memory on the host; n keys, `stride` f4 per span; record layout: TrackHot in
// fyx_internal.h): the curves share their key times, so the decisions -- clamp at the ends, the hinted span [hint - 1, hint),
// else partition_point(k.location < time) -- are taken once, in the reference's order, and every curve's hint becomes the
// same value.  The search result is found without searching when it is the hint itself (time on the right key) or a
// neighbour (playback crossed a key: at 60 frames a second over 30 keys a second half the instances do every frame).
// Returns the new hint.
// ---------------------------------------------------------------------------------------
template <typename SP>
FYX_HD uint32_t span_track_value_at(SP sp, uint32_t n, uint32_t stride, int need, float time, uint32_t h, float (&val)[4]) {
    SP last = sp + (size_t)(n - 2u) * stride;
    const float l_first = sp[0].x, l_last = last[0].y;
    if (time <= l_first) {
#pragma unroll
        for (int c = 0; c < 4; ++c) if (c < need) val[c] = sp[1 + c].x;              // first key's value
        return 0u;
    }
    if (time >= l_last) {
#pragma unroll
        for (int c = 0; c < 4; ++c) if (c < need) val[c] = last[1 + c].y;            // last key's value
        return n - 1u;
    }
    // right key of the span that holds the time: key `hint` if the hinted span holds it, else the first key at or after the time
    uint32_t right = 0u;                        // 0: not found yet (the first key lies before the time)
    f4 locs = f4{0.f, 0.f, 0.f, 0.f};
    if (h >= 1u && h < n) {
        locs = sp[(size_t)(h - 1u) * stride];
        // (time on the right key: the hinted test fails and the search returns hint -- unless the left key has the same
        // location, then it returns an earlier key: duplicates go to the search)
        if (time >= locs.x && time <= locs.y && locs.x < locs.y) right = h;
        else if (time > locs.y && h + 1u < n) {
            locs = sp[(size_t)h * stride];
            if (time <= locs.y) right = h + 1u;               // (time > its left key: that is the hinted span's right key)
        } else if (time < locs.x && h >= 2u) {
            locs = sp[(size_t)(h - 2u) * stride];
            if (time > locs.x) right = h - 1u;                // (time < its right key)
        }
    }
    if (!right) {                                // partition_point(k.location < time) over the keys
        uint32_t lo = 0u, hi = n;
        while (lo < hi) {
            const uint32_t mid = lo + (hi - lo) / 2u;
            const float l_mid = mid + 1u < n ? sp[(size_t)mid * stride].x : l_last;
            if (l_mid < time) lo = mid + 1u; else hi = mid;
        }
        right = lo;                              // 1 <= lo <= n - 1: first < time < last
        locs = sp[(size_t)(right - 1u) * stride];
    }
    SP r = sp + (size_t)(right - 1u) * stride;
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (c < need) val[c] = interpolate_span(locs.x, locs.y, span_kind(locs, (uint32_t)c), r[1 + c], time);
    return right;
}

// ---------------------------------------------------------------------------------------
// STRAIGHT fold programs: [PUSH^d] BLEND_ANIM^k [POP_BLEND^d] [MASK] APPLY END with 1 <= k <= kStraightOps -- a machine whose
// layers after the first are off, in one state or in one transition between states whose roots are single clips, or one
// state whose root is one blend node (BASELINE configs 2, 3, 5; the host writes such programs without the PUSHes, see
// Planner::emit_blend).  Every pose the PUSHes open is empty when its child is popped into it, and an empty pose becomes a
// COPY of the other (pose.rs:41-47, weight ignored), so the result is the k operands blended in order into one accumulator
// -- what pose_update's straight form computes without the interpreter.
// A second straight shape is the AnimationPlayer's program, APPLY_ANIM^k END (k <= kStraightOps): k poses applied in turn.
// The decision is taken from four 64-bit masks (bit p: op p is a PUSH / BLEND_ANIM / POP_BLEND / APPLY_ANIM; the kernel makes
// them with one ballot each over the program held in its lanes, the host with a loop) and the opcodes of the ops of the tail.
// ---------------------------------------------------------------------------------------
constexpr uint32_t kStraightOps = 4;

struct StraightShape {
    uint32_t d, k;     // leading PUSHes, BLEND_ANIMs behind them (player: k = the APPLY_ANIMs)
    bool mask;         // an OP_MASK sits between the pops and the APPLY
    bool player;       // APPLY_ANIM^k END (AnimationPlayer: every enabled animation's pose applied in turn, plan_player)
    bool straight;
};

FYX_HD uint32_t ctz64_(uint64_t v) { return (uint32_t)__builtin_ctzll(v); }   // v != 0

// code_at(pc): the opcode (low byte of op.x) of op pc, pc < 64; ops at and past n_ops read as OP_END (0).
template <typename CodeAt>
FYX_HD StraightShape classify_fold_program(uint32_t n_ops, uint64_t m_push, uint64_t m_blend, uint64_t m_pop, uint64_t m_apply_anim,
                                           CodeAt code_at) {
    StraightShape s{0u, 0u, false, false, false};
    if (n_ops < 2u || n_ops > 64u) return s;
    const uint32_t applies = ctz64_(~m_apply_anim | (1ull << 63));
    if (applies) {   // the player's program: nothing but APPLY_ANIMs ahead of the END
        s.k = applies;
        s.player = true;
        s.straight = applies <= kStraightOps && n_ops == applies + 1u && code_at(applies) == (uint32_t)OP_END;
        return s;
    }
    if (n_ops < 3u) return s;
    s.d = ctz64_(~m_push | (1ull << 63));
    s.k = ctz64_(~(m_blend >> s.d) | (1ull << 63));
    const uint32_t pops = ctz64_(~(m_pop >> ((s.d + s.k) & 63u)) | (1ull << 63));
    uint32_t tail = 2u * s.d + s.k;
    if (tail + 2u > 64u) return s;
    s.mask = code_at(tail) == (uint32_t)OP_MASK;
    if (s.mask) ++tail;
    if (tail + 1u >= 64u) return s;
    s.straight = s.k >= 1u && s.k <= kStraightOps && pops == s.d && s.d + 1u < (uint32_t)kMaxFoldDepth && n_ops == tail + 2u &&
                 code_at(tail) == (uint32_t)OP_APPLY && code_at(tail + 1u) == (uint32_t)OP_END;
    return s;
}

// Host form: the masks from the program as it lies in memory (ops[p].x's low byte is the opcode).
inline StraightShape classify_fold_program_host(const uint32_t* ops_xy /* n_ops pairs {x, y} */, uint32_t n_ops) {
    uint64_t m_push = 0, m_blend = 0, m_pop = 0, m_apply_anim = 0;
    const uint32_t n = n_ops < 64u ? n_ops : 64u;
    for (uint32_t p = 0; p < n; ++p) {
        const uint32_t code = ops_xy[2 * p] & 0xffu;
        if (code == (uint32_t)OP_PUSH) m_push |= 1ull << p;
        else if (code == (uint32_t)OP_BLEND_ANIM) m_blend |= 1ull << p;
        else if (code == (uint32_t)OP_POP_BLEND) m_pop |= 1ull << p;
        else if (code == (uint32_t)OP_APPLY_ANIM) m_apply_anim |= 1ull << p;
    }
    return classify_fold_program(n_ops, m_push, m_blend, m_pop, m_apply_anim,
                                 [&](uint32_t pc) -> uint32_t { return pc < n_ops ? (ops_xy[2 * pc] & 0xffu) : 0u; });
}

}  // namespace fyx
