// anim_kernels.hip -- gfx950 kernels of the pose path that feeds the skinning kernel:
//   pose_sample  : keyframe sampling of every ticked animation of every instance
//                  (Curve::value_at fyrox-math/src/curve.rs:254-314, TrackDataContainer::fetch
//                  fyrox-animation/src/container.rs:287-297, Animation::update_pose lib.rs:895-914)
//   pose_update  : per instance, per node: run the instance's fold program (AnimationPose::
//                  blend_with semantics, pose.rs:41-101 / value.rs:221-230,438-459; the program is
//                  the Machine/Layer/PoseNode evaluation order flattened by the host control
//                  plane), apply the result to the node's local TRS (scene/animation/mod.rs:147-186),
//                  build the local matrix (scene/transform.rs:421-540) and propagate
//                  global = parent.global * local (scene/graph/mod.rs:1199-1241)
//   palette_gather: palette[b] = global[bone_b] * inv_bind[bone_b] (scene/mesh/mod.rs:781-793)
//
// Mapping to the hardware.  A pose is tiny next to the vertices it drives (10 floats per bone),
// so this is latency-/launch-bound work, not bandwidth-bound: the design goal is ONE block per
// instance that never leaves the chip between "sampled poses" and "global matrices":
//   * thread = node; the fold program of an instance is block-uniform, so the interpreter has no
//     divergence and its operands are scalar loads; nested blend accumulators live in VGPRs
//     (the nesting is unrolled at compile time, kMaxFoldDepth levels -- no scratch, no LDS);
//     the common programs -- a few clips blended in a row, which is what the host writes for a
//     machine in one state or one transition -- skip the interpreter (straight form);
//   * the samplers find their track through a 32-byte descriptor per (animation, node, binding)
//     (CrowdDesc) and sample tracks whose curves share their key times from span records;
//   * local and global matrices of the instance stay in LDS across the level-synchronous
//     hierarchy walk (64 B x 2 per node; 1024 nodes = 128 KiB of the CU's 160 KiB);
//   * records are float4-packed and node-contiguous, so a wave's loads are dense.
// Arithmetic: f32, reference operation order, no FMA contraction (-ffp-contract=off), IEEE
// divide and sqrt -- bit-identical to the CPU path except sin/cos of Euler tracks (<= 1 ulp f32
// sincosf; libm's sinf is not reproducible bit-for-bit on a GPU).
#include "fyx_internal.h"

#include "../../include/fyrox_hip.h"
#include "lbs_leaves.h"    // the skinning kernels' per-vertex code: the one-launch frame goes on to the vertices
#include "anim_leaves.h"   // lerpf_, cubicf_, interpolate_loaded, span_track_value_at, classify_fold_program (host + device)

#include <hip/hip_ext.h>

namespace fyx {

constexpr int kFrameSkinLoadAux = 2, kFrameSkinStoreAux = 16;     // nt loads, sc1 stores (lbs_skin_dyn's pair)

thread_local LaunchEvents g_launch_events;

// Experiment builds only (tools/exp/r05_stamps_build.sh compiles with -DFYX_FRAME_STAMPS): thread 0 of every workgroup of the one-launch
// frame leaves wall-clock stamps (100 MHz) -- 0 entry, 1 first requests issued, 2 the samplers have reported, 3 fold done, 4 local
// matrices in LDS, 5 walk done, 6 palette in LDS / stores issued, 7 done.  Not in the product library.
#ifdef FYX_FRAME_STAMPS
__device__ unsigned long long g_fstamp[1024 * 8];
#define FSTAMP(i) do { if (threadIdx.x == 0) g_fstamp[(blockIdx.x & 1023u) * 8u + (i)] = wall_clock64(); } while (0)
#else
#define FSTAMP(i) do {} while (0)
#endif
// Experiment builds only (tools/exp/r06_scene_stamps_build.sh, -DFYX_SCENE_STAMPS): thread 0 of every workgroup of the scene's sampler leaves
// wall-clock stamps (100 MHz) -- 0 entry, 1 the job's parameters have arrived, 2 this lane's curve is sampled, 3 the record's stores are issued.
#ifdef FYX_SCENE_STAMPS
__device__ unsigned long long g_sstamp[8192 * 4];
#define SSTAMP(i) do { if (threadIdx.x == 0) g_sstamp[(blockIdx.x & 8191u) * 4u + (i)] = wall_clock64(); } while (0)
__device__ unsigned int g_spath[4];      // curve samples by exit: 0 inside the hinted span, 1 next span, 2 previous span, 3 the general path
#define SPATH(i) atomicAdd(&g_spath[i], 1u)
#else
#define SSTAMP(i) do {} while (0)
#define SPATH(i) do {} while (0)
#endif
// a launch that takes the armed timeline events, if any (option debug.timeline)
#define FYX_TL_LAUNCH(kernel, grid, block, lds, s, ...)                                                                      \
    do {                                                                                                                      \
        if (g_launch_events.start) {                                                                                          \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, s, g_launch_events.start, g_launch_events.stop, 0, __VA_ARGS__);  \
            g_launch_events = LaunchEvents();                                                                                 \
        } else hipLaunchKernelGGL(kernel, grid, block, lds, s, __VA_ARGS__);                                                  \
    } while (0)

// (LP / AP: pointers to the key locations / {value, kind, tangents} records -- global memory, or LDS where the crowd
// sampler has staged the curve)
template <typename LP, typename AP>
__device__ __forceinline__ float interpolate_keys(LP loc, AP aux, uint32_t l, uint32_t r, float location) {
    return interpolate_loaded(loc[l], loc[r], aux[l], aux[r], location);
}

// Curve::value_at with the caller's span hint (curve.rs:254-314).
// A sample is a chain of dependent loads, and this kernel is bound by that latency and by the number of cache lines
// it touches, not by bytes: the first and last key (clamping) come with the track record (TrackDev, one line per
// track), the two keys of the hinted span are fetched in ONE round trip right after the hint is known (four
// independent loads); a hint that is off by one span costs one more round trip, and only a hint further off pays
// for the binary search.  The decisions are taken in the reference's order on the same values.
struct CurveEnds { float l_first, l_last, v_first, v_last; };   // first / last key of the curve: location, value
struct CurveKeys {   // the two keys of the hinted span
    float l_hl, l_h;
    f4 a_hl, a_h;
};

__device__ __forceinline__ CurveEnds curve_ends(const TrackDev* tk, int c) {
    return CurveEnds{tk->first_loc[c], tk->last_loc[c], tk->first_val[c], tk->last_val[c]};
}

template <typename LP, typename AP>
__device__ __forceinline__ CurveKeys curve_fetch(LP loc, AP aux, uint32_t n, uint32_t h) {
    CurveKeys k;
    const uint32_t nn = n ? n : 1;                                    // an empty curve reads key 0 of its successor, unused
    const uint32_t hc = h < nn ? h : nn - 1, hl = hc > 0 ? hc - 1 : 0;   // clamped: addresses stay inside the curve
    k.l_hl = loc[hl]; k.l_h = loc[hc];
    k.a_hl = aux[hl]; k.a_h = aux[hc];
    return k;
}

// The decisions of value_at, in the reference's order, on values fetched earlier.
// NEIGHBOURS: resolve a hint that is off by one span without the binary search (one extra round trip instead of
// ~five).  Worth it where a round trip is a cold HBM access (many animators with their own key data: -5 %); on a
// crowd, whose few curves sit in L2 and whose bound is the number of memory instructions, it measured 6 % slower.
template <bool NEIGHBOURS, typename LP, typename AP>
__device__ __forceinline__ float curve_eval(const CurveKeys& k, const CurveEnds& e, LP loc, AP aux, uint32_t n, float location,
                                            uint32_t& hint) {
    if (n == 0) return 0.0f;
    const uint32_t h = hint;
    if (location <= e.l_first) { hint = 0; return e.v_first; }
    if (location >= e.l_last) { hint = n - 1; return e.v_last; }
    if (h < n) {
        if (location >= k.l_hl && location < k.l_h) return interpolate_loaded(k.l_hl, k.l_h, k.a_hl, k.a_h, location);
        // The hint missed.  The reference now runs partition_point(|k| k.location < location) = the first key at or
        // after `location`.  Keys are sorted, so if key i - 1 lies before `location` and key i at or after it, that
        // index IS i -- no search needed.  Playback is continuous (curve.rs:293-297 has this as a TODO): i is almost
        // always the hinted index itself (location sits exactly on its right key), the next one (forward playback
        // crossed a key) or the previous one (reverse playback).  Same hint, same two keys, same arithmetic.
        if (h >= 1 && k.l_hl < location && location <= k.l_h)
            return interpolate_loaded(k.l_hl, k.l_h, k.a_hl, k.a_h, location);                       // i == h
        if constexpr (NEIGHBOURS) {
            if (h + 1 < n && k.l_h < location) {
                const float l_n = loc[h + 1];
                if (location <= l_n) {
                    hint = h + 1;
                    return interpolate_loaded(k.l_h, l_n, k.a_h, aux[h + 1], location);              // i == h + 1
                }
            } else if (h >= 2 && location <= k.l_hl) {
                const float l_p = loc[h - 2];
                if (l_p < location) {
                    hint = h - 1;
                    return interpolate_loaded(l_p, k.l_hl, aux[h - 2], k.a_hl, location);            // i == h - 1
                }
            }
        }
    }
    uint32_t lo = 0, hi = n;  // partition_point(|k| k.location < location)
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (loc[mid] < location) lo = mid + 1; else hi = mid;
    }
    hint = lo;
    return interpolate_keys(loc, aux, lo > 0 ? lo - 1 : 0, lo, location);
}

template <bool NEIGHBOURS = true, typename LP, typename AP>
__device__ __forceinline__ float curve_value_at(LP loc, AP aux, uint32_t n, const CurveEnds& ends, float location, uint32_t& hint) {
    if (n == 0) return 0.0f;
    const CurveKeys k = curve_fetch(loc, aux, n, hint);
    return curve_eval<NEIGHBOURS>(k, ends, loc, aux, n, location, hint);
}

// Span hint of (animation a, track, curve c, instance): Curve::value_at's `&mut usize`.
__device__ __forceinline__ uint32_t* hint_ptr(const PoseFrameDev& f, uint32_t a, uint32_t track, uint32_t c, uint32_t inst) {
    return f.hints + (((size_t)a * f.max_tracks + track) * 4 + c) * f.n_instances + inst;
}

// nalgebra leaves, in nalgebra's operation order
__device__ __forceinline__ float dot4(f4 a, f4 b) {
    float x = a.x * b.x, y = a.y * b.y;
    const float z = a.z * b.z, w = a.w * b.w;
    x += z;
    y += w;
    return x + y;
}
__device__ __forceinline__ f4 quat_normalize(f4 q) {
    const float n = sqrtf(dot4(q, q));
    return f4{q.x / n, q.y / n, q.z / n, q.w / n};
}
__device__ __forceinline__ f4 quat_mul(f4 a, f4 b) {  // Hamilton product, storage (i,j,k,w)
    const float w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    const float i = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    const float j = a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
    const float k = a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
    return f4{i, j, k, w};
}
// Rotation value of a lane group whose lanes base..base+3 sampled the (up to four) curves of a
// rotation track: UnitQuaternion -> from_quaternion (normalise, container.rs:277-279);
// UnitQuaternionEuler -> qz * qy * qx (fyrox-math/src/lib.rs:725-740), lanes base..base+2 each
// evaluating sin/cos of their own half angle once.  has_r / rkind are uniform over the group;
// every lane of the wave must call this (cross-lane shuffles).
__device__ __forceinline__ f4 group_rotation(float v, int has_r, int rkind, int base) {
    const float r0 = __shfl(v, base, 64), r1 = __shfl(v, base + 1, 64);
    const float r2 = __shfl(v, base + 2, 64), r3 = __shfl(v, base + 3, 64);
    f4 q = f4{0.f, 0.f, 0.f, 1.f};
    if (__any(has_r && rkind == FYX_KIND_QUAT_EULER)) {
        // (axis * sin(angle/2), cos(angle/2)) of this lane's own angle; lanes base.. are x,y,z
        // f32 sin/cos (<= 1 ulp each): libm's sinf/cosf cannot be matched bit for bit on a GPU anyway (DESIGN.md
        // section 2), and the f64 evaluation used before cost more than the whole rest of the sample
        const float half = v / 2.0f;
        float sn, cs;
        sincosf(half, &sn, &cs);
        const float sx = __shfl(sn, base, 64), cx = __shfl(cs, base, 64);
        const float sy = __shfl(sn, base + 1, 64), cy = __shfl(cs, base + 1, 64);
        const float sz = __shfl(sn, base + 2, 64), cz = __shfl(cs, base + 2, 64);
        if (has_r && rkind == FYX_KIND_QUAT_EULER) {
            const f4 qx = f4{1.0f * sx, 0.0f * sx, 0.0f * sx, cx};
            const f4 qy = f4{0.0f * sy, 1.0f * sy, 0.0f * sy, cy};
            const f4 qz = f4{0.0f * sz, 0.0f * sz, 1.0f * sz, cz};
            q = quat_mul(quat_mul(qz, qy), qx);
        }
    }
    if (has_r && rkind == FYX_KIND_QUAT) q = quat_normalize(f4{r0, r1, r2, r3});
    return q;
}

// Key records as the `loc[i]` / `aux[i]` the curve functions index (pose_sample: one line per span instead of two).
#ifndef FYX_KEYREC
#define FYX_KEYREC 1
#endif
struct RecLoc {
    const KeyRec* p;
    __device__ __forceinline__ float operator[](uint32_t i) const { return p[i].loc; }
};
struct RecAux {
    const KeyRec* p;
    __device__ __forceinline__ f4 operator[](uint32_t i) const {
        const float4 a = p[i].aux;
        return f4{a.x, a.y, a.z, a.w};
    }
};

// ---------------------------------------------------------------------------------------
// One curve of one track for ONE instance: Curve::value_at (curve.rs:254-314) with the instance's span hint, which it keeps up to
// date.  (Round 4 measured a FUSED sample + update launch for single characters on top of this function -- one lane per (animation,
// node), its ten curves one after another, then the update body behind a barrier: C2 18.6 us against 15.8, C5 25.5 against 18.0 with
// the two launches; ten dependent sample chains per lane cost more than a launch boundary.  Not kept.)
// ---------------------------------------------------------------------------------------
// hp / hint: the curve's span hint and where it lives (the caller requested it together with the descriptor)
__device__ __forceinline__ float sample_curve(const PoseFrameDev& f, uint32_t a, const CrowdDesc& d, uint32_t c, uint32_t* hp, uint32_t hint, float time) {
    const uint32_t track = d.track;
    float v = 0.0f;
    // Steady playback: the time lies strictly inside the hinted span [key hint - 1, key hint).  Curve::value_at then
    // clamps nothing (first.location <= left < time < right <= last.location), takes its hinted span and leaves the
    // hint alone (curve.rs:254-314) -- and the track's span record (TrackHot) holds everything that needs: one cache
    // line for the three curves of a Vector3 track, two for a quaternion's four.  Where the clip's tracks share one time grid
    // the records lie [span][track] (TracksData::d_span_rows, round 6): the 192 records a 64-node character reads of a clip in
    // a frame are one dense run, not a line here and a line there.  (Requesting the neighbouring span -- the next row -- together
    // with the hinted one, so that a crossed key costs no second round trip, was measured on top of that: 19.4 against 19.8 us,
    // not kept; the frames that take 20 us instead of 12 are not slow because of that trip.)
    bool sampled = false;
    if (d.spans && hint >= 1 && hint < d.n_keys) {
        const uint32_t stride = d.valid >> 8;       // (f4 to the next span's record: the track's own table, or a row of [span][track])
        const f4* r = reinterpret_cast<const f4*>(d.spans) + (size_t)(hint - 1) * stride;
        f4 locs = r[0];
        if (locs.x < time && time < locs.y) {
            v = interpolate_span(locs.x, locs.y, span_kind(locs, c), r[1 + c], time);
            sampled = true;
            SPATH(0);
        } else if (locs.y < time && hint + 1 < d.n_keys) {
            // playback crossed the span's right key: if the time lies strictly inside the NEXT span, nothing is clamped
            // there either, the hinted span fails, and partition_point(k.location < time) is hint + 1 (every key up to
            // `hint` lies before the time, key hint + 1 after it): the reference interpolates keys hint, hint + 1
            r += stride;
            locs = r[0];
            if (locs.x < time && time < locs.y) {
                v = interpolate_span(locs.x, locs.y, span_kind(locs, c), r[1 + c], time);
                *hp = hint + 1;
                sampled = true;
                SPATH(1);
            }
        } else if (time < locs.x && hint >= 2) {
            // reverse playback crossed the left key: strictly inside the PREVIOUS span the search returns hint - 1
            r -= stride;
            locs = r[0];
            if (locs.x < time && time < locs.y) {
                v = interpolate_span(locs.x, locs.y, span_kind(locs, c), r[1 + c], time);
                *hp = hint - 1;
                sampled = true;
                SPATH(2);
            }
        }
    }
    if (!sampled && d.spans && d.n_keys >= 2u) {
        // The hint is far off: a looping animation wrapped (forwards: the time is back in the FIRST span with the hint at the last key;
        // backwards: the other way round), or it has just left a clamp at either end.  A time STRICTLY inside any span s -- key s - 1 before
        // it, key s after it -- is interpolated over exactly those two keys whatever the incoming hint is: nothing is clamped, no other
        // span's test can pass (keys are sorted), and partition_point(k.location < time) is s, which becomes the hint (curve.rs:254-314).
        // So the first and the last span are worth one more round trip before the per-curve search (round 6: in a scene of 256 characters
        // ~17 clips wrap every frame, and the workgroups that searched for them were what the whole launch waited for).
        const uint32_t stride = d.valid >> 8;
        const f4* first = reinterpret_cast<const f4*>(d.spans);
        const f4* last = first + (size_t)(d.n_keys - 2u) * stride;
        const f4 lf = first[0], ll = last[0], pf = first[1 + c], pl = last[1 + c];
        if (lf.x < time && time < lf.y) {
            v = interpolate_span(lf.x, lf.y, span_kind(lf, c), pf, time);
            *hp = 1u;
            sampled = true;
            SPATH(1);
        } else if (ll.x < time && time < ll.y) {
            v = interpolate_span(ll.x, ll.y, span_kind(ll, c), pl, time);
            *hp = d.n_keys - 1u;
            sampled = true;
            SPATH(2);
        }
    }
    if (!sampled) {   // everything else, decided in the reference's order on the per-curve records
        SPATH(3);
        const AnimDev an = f.anims[a];
        const TrackDev* tk = an.tracks + track;
        const uint32_t fk = tk->first_key[c];
#if FYX_KEYREC
        v = curve_value_at(RecLoc{an.key_rec + fk}, RecAux{an.key_rec + fk}, tk->n_keys[c], curve_ends(tk, (int)c), time, hint);
#else
        v = curve_value_at(an.key_loc + fk, reinterpret_cast<const f4*>(an.key_aux) + fk, tk->n_keys[c],
                           curve_ends(tk, (int)c), time, hint);
#endif
        *hp = hint;
    }
    return v;
}

// ---------------------------------------------------------------------------------------
// pose_sample: sixteen lanes per (animation, instance, node), ONE LANE PER CURVE.  A curve sample is
// a chain of dependent loads (hint -> key locations -> key values), so a thread that walked the ten
// curves of a node one after another would pay that latency ten times; here the ten chains of a node
// run side by side and the lanes exchange their results with cross-lane shuffles.
// Lane j of a group owns dword j of the node's 48-byte pose record, so the store is one dense 48-byte
// span per group (192 B per wave):
//     j = 0..2  position x,y,z     j = 3  present bits (1 Position, 2 Scale, 4 Rotation)
//     j = 4..7  rotation i,j,k,w   j = 8..10  scale x,y,z     j = 11  zero      j = 12..15 idle
// Euler tracks: lanes 4,5,6 each evaluate sin/cos of their own half angle once, every rotation lane
// then forms qz * qy * qx (fyrox-math/src/lib.rs:725-740) from the shuffled values.
// ---------------------------------------------------------------------------------------
// Grid: x = 16-node slices of one instance, y = instance, z = animation -- no index arithmetic beyond shifts
// (the kernel is VALU-issue bound: ~64 K waves of a few hundred instructions each for the C3 crowd).
// WRITE_THROUGH: the record is stored with agent-scope (sc1) stores, which go through the XCD's L2 to memory -- for the one-launch frame,
// whose update workgroups (on other XCDs, each with its own L2) read the records while this kernel is still running.
template <bool WRITE_THROUGH = false>
__device__ __forceinline__ void pose_sample_body(const PoseFrameDev& f, uint32_t bx, uint32_t by, uint32_t bz) {
    const uint32_t lane = threadIdx.x & 63u, j = threadIdx.x & 15u, gbase = lane & ~15u;
    const uint32_t a = bz, inst = by;
    const uint32_t node = (bx * 256u + threadIdx.x) >> 4;
    {
        if (node >= f.n_nodes) return;                                   // uniform across the group
        // which binding / curve this lane serves
        int bind = -1, c = 0;
        if (j < 3) { bind = FYX_BIND_POSITION; c = (int)j; }
        else if (j >= 4 && j < 8) { bind = FYX_BIND_ROTATION; c = (int)j - 4; }
        else if (j >= 8 && j < 11) { bind = FYX_BIND_SCALE; c = (int)j - 8; }
        // The animator's descriptor of (animation, node, binding) -- CrowdDesc: slot table, track record and TrackHot resolved
        // by the host -- is two 16-byte loads off a kernel argument; the chain animation record -> slot -> TrackHot it replaces
        // was three dependent round trips ahead of the hint.  (Lane 3 writes the present bits: they are in every descriptor.)
        // Requested BEFORE the tick flag is looked at: its address does not depend on it, and the early return below would
        // otherwise put a round trip of its own in front of this one.
        const CrowdDesc d = f.crowd[((size_t)a * f.n_nodes + node) * 3 + (uint32_t)(bind >= 0 ? bind : 0)];
        // ... and the lane's span hint with it: its slot is a function of (animation, node, binding, curve, instance), nothing the
        // descriptor has to say first (round 5: one dependent round trip less in a kernel that is made of them)
        uint32_t* hp = f.slot_hints + ((((size_t)a * f.n_nodes + node) * 3 + (uint32_t)(bind >= 0 ? bind : 0)) * 4 + (uint32_t)c) * f.n_instances + inst;
        uint32_t hint = 0;
        if (bind >= 0) hint = *hp;
        const float time = f.times[(size_t)inst * f.n_anims + a];
        if (!(f.ticked[(size_t)inst * f.n_anims + a] & 1u)) return;      // uniform across the block
        const size_t item = ((size_t)a * f.n_instances + inst) * f.n_nodes + node;
        const int32_t track = bind >= 0 && (d.valid || d.kind >= 0) ? (int32_t)d.track : -1;
        const bool has_prop = j == 3 && (d.present & 8u);
        int kind = -1, need = 0;
        bool valid = false;
        float v = 0.0f;
        if (track >= 0) {
            kind = d.kind;
            need = (int)d.need;
            valid = d.valid != 0;                                 // else fetch() -> None
            if (valid && c < need) v = sample_curve(f, a, d, (uint32_t)c, hp, hint, time);
        }
#ifdef FYX_SCENE_STAMPS
        asm volatile("" : "+v"(v));
        SSTAMP(2);
#endif
        const int has_p = __shfl((int)valid, (int)gbase + 0, 64);
        const int has_r = __shfl((int)valid, (int)gbase + 4, 64);
        const int has_s = __shfl((int)valid, (int)gbase + 8, 64);
        const int rkind = __shfl(kind, (int)gbase + 4, 64);

        const f4 q = group_rotation(v, has_r, rkind, (int)gbase + 4);

        float out;
        if (j < 3 || (j >= 8 && j < 11)) out = v;   // an absent binding sampled nothing: 0
        else if (j == 3) out = __uint_as_float((has_p ? 1u : 0u) | (has_s ? 2u : 0u) | (has_r ? 4u : 0u) | (has_prop ? 8u : 0u) | (d.present & 16u));   // (16: the node's list holds a value that fits no binding)
        else if (j == 4) out = q.x;
        else if (j == 5) out = q.y;
        else if (j == 6) out = q.z;
        else if (j == 7) out = q.w;
        else out = 0.0f;
        if (j < 12) {
            float* dst = reinterpret_cast<float*>(f.anim_pose) + item * 12 + j;
            if constexpr (WRITE_THROUGH) __hip_atomic_store(dst, out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else *dst = out;
        }
    }
}

// The frame's control block inside the kernel arguments (CtrlInline, fyx_internal.h): the control pointers are offsets from the
// start of `inl`, which the kernel reads where it lies -- in the kernel-argument segment, ordinary device-visible memory.
// KARG_OFF: where `inl` lies in the kernel-argument segment (the parameters before it are 8-byte aligned structs, CtrlInline is
// 4-byte aligned: it follows them directly).  The address comes from the segment pointer, not from `&inl`: taking the address of
// the by-value parameter makes the compiler copy the whole kilobyte to scratch in some kernels.
template <size_t KARG_OFF>
__device__ __forceinline__ PoseFrameDev ctrl_resolve(PoseFrameDev f, const CtrlInline& inl) {
    if (inl.bytes) {
#if defined(__HIP_DEVICE_COMPILE__)
        const char* cb = reinterpret_cast<const char*>((uintptr_t)__builtin_amdgcn_kernarg_segment_ptr()) + KARG_OFF;
#else
        const char* cb = nullptr;    // (the host pass only parses this)
#endif
        f.times = reinterpret_cast<const float*>(cb + reinterpret_cast<uintptr_t>(f.times));
        f.ticked = reinterpret_cast<const uint8_t*>(cb + reinterpret_cast<uintptr_t>(f.ticked));
        f.ops = reinterpret_cast<const uint2*>(cb + reinterpret_cast<uintptr_t>(f.ops));
        f.prog_off = reinterpret_cast<const uint32_t*>(cb + reinterpret_cast<uintptr_t>(f.prog_off));
        if (f.slices) f.slices = reinterpret_cast<const float2*>(cb + reinterpret_cast<uintptr_t>(f.slices));
        if (f.rm_ops) f.rm_ops = reinterpret_cast<const uint4*>(cb + reinterpret_cast<uintptr_t>(f.rm_ops));
        if (f.rm_prog_off) f.rm_prog_off = reinterpret_cast<const uint32_t*>(cb + reinterpret_cast<uintptr_t>(f.rm_prog_off));
    }
    return f;
}
// (the host lays the sections out 16-byte aligned from the start of CtrlInline, and the kernels read rm_ops as uint4: the offsets
// of CtrlInline in the kernel-argument segment must be multiples of 16)
static_assert(sizeof(PoseFrameDev) % 16 == 0 && (sizeof(PoseFrameDev) + sizeof(RigDev)) % 16 == 0 && alignof(CtrlInline) == 16, "kernel-argument layout of (PoseFrameDev[, RigDev], CtrlInline)");
constexpr size_t kInlAfterFrame = sizeof(PoseFrameDev), kInlAfterFrameAndRig = sizeof(PoseFrameDev) + sizeof(RigDev);
static const CtrlInline kNoInline = {};

__global__ __launch_bounds__(256) void pose_sample_kernel(PoseFrameDev f, CtrlInline inl) { pose_sample_body(ctrl_resolve<kInlAfterFrame>(f, inl), blockIdx.x, blockIdx.y, blockIdx.z); }

// Scene forms of the kernels in this file (fyx_scene_update): ONE launch covers the same stage of MANY animators.
// Block b of the launch looks up (job, x, y, z) in a table that depends only on the scene's shape -- which animator
// it works for and which block of that animator's own grid it stands for -- copies the job's parameter block out of
// HBM (uniform address, loaded before any store: scalar loads, exactly like kernel arguments) and runs the body above.
// A job's frame parameters: the job array is resident (it changes when the scene does, not every frame), its control pointers
// are offsets into the frame's control block `ctrl` -- the only thing a steady frame uploads.
__device__ __forceinline__ PoseFrameDev scene_frame_of(const SceneJobDev* __restrict__ jobs, uint32_t job, const char* __restrict__ ctrl) {
    PoseFrameDev f = jobs[job].f;
    f.times = reinterpret_cast<const float*>(ctrl + reinterpret_cast<uintptr_t>(f.times));
    f.ticked = reinterpret_cast<const uint8_t*>(ctrl + reinterpret_cast<uintptr_t>(f.ticked));
    f.ops = reinterpret_cast<const uint2*>(ctrl + reinterpret_cast<uintptr_t>(f.ops));
    f.prog_off = reinterpret_cast<const uint32_t*>(ctrl + reinterpret_cast<uintptr_t>(f.prog_off));
    if (f.slices) f.slices = reinterpret_cast<const float2*>(ctrl + reinterpret_cast<uintptr_t>(f.slices));
    if (f.rm_ops) f.rm_ops = reinterpret_cast<const uint4*>(ctrl + reinterpret_cast<uintptr_t>(f.rm_ops));
    if (f.rm_prog_off) f.rm_prog_off = reinterpret_cast<const uint32_t*>(ctrl + reinterpret_cast<uintptr_t>(f.rm_prog_off));
    return f;
}

__global__ __launch_bounds__(256) void pose_sample_scene_kernel(const SceneJobDev* __restrict__ jobs, const char* __restrict__ ctrl, const uint4* __restrict__ blocks) {
    SSTAMP(0);
    const uint4 b = blocks[blockIdx.x];
    const PoseFrameDev f = scene_frame_of(jobs, b.x, ctrl);
#ifdef FYX_SCENE_STAMPS
    { uint32_t nn = f.n_nodes; asm volatile("" : "+s"(nn)); SSTAMP(1); }
#endif
    pose_sample_body(f, b.y, b.z, b.w);
    SSTAMP(3);
}

// ---------------------------------------------------------------------------------------
// Crowd form of pose_sample: the lanes of a wave are 64 INSTANCES of one (animation, node).
//
// The form above puts the curves of one instance side by side, so the lanes of every key load hit up to 64 different
// cache lines (ten scattered requests per curve sample -- the texture-addresser, not VALU or HBM, sets its 51 us on
// the C3 crowd).  A crowd samples the SAME curves at different times: with instances on the lanes, a load of a
// curve's keys touches the few lines around the instances' playback positions, the span hints (instance-minor) are
// one dense span, and the only scattered access left is the 16-byte store of the wave's part of the pose record.
// Same arithmetic, same order: bit-identical to the form above.
// ---------------------------------------------------------------------------------------
constexpr uint32_t kCurveLdsKeys = 128;   // curves up to this many keys are staged in LDS by the crowd sampler (2.5 KB per wave)
constexpr uint32_t kSpanLdsF4 = 512;      // ... and a track's span records up to this many 16-byte words (8 KB per wave: 64 spans of a Vector3 track)

// (span_track_value_at -- Curve::value_at for the curves of one track on its span records -- lives in anim_leaves.h)

// BLOCK: 64, or 256 = four waves (256 instances) of the same (animation, node, binding) that stage the track's span records
// TOGETHER: a quarter of the staging loads and of the LDS per wave (8 KB per wave let 20 waves onto a CU; the kernel is made of
// memory latency, and what hides it is waves).
template <uint32_t BLOCK>
__device__ __forceinline__ void pose_sample_crowd_body(const PoseFrameDev& f, uint32_t bx, uint32_t by, uint32_t bz) {
    // One wave = 64 instances of one (animation, node, BINDING): the position, scale and rotation tracks of a node
    // are sampled by three different waves, so a thread walks at most four curves (the chain of dependent loads
    // is what bounds this kernel) and the three 16-byte parts of the pose record have one writer each.
    const uint32_t inst = bx * BLOCK + threadIdx.x, a = bz;
    const uint32_t node = by / 3u;
    const int bind = (int)(by - node * 3u);           // FYX_BIND_POSITION, _SCALE, _ROTATION
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    // a lane without work (past the crowd, or an animation that did not tick for its instance) still helps to stage
    // the curves below
    const bool active = inst < f.n_instances && (f.ticked[(size_t)(inst < f.n_instances ? inst : 0) * f.n_anims + a] & 1u);
    if constexpr (BLOCK == 64) {
        if (!__any(active)) return;
    }
    const float time = active ? f.times[(size_t)inst * f.n_anims + a] : 0.0f;
    // span records of the track: one area for the workgroup; per-curve staging: one per wave (a lone wave uses the span area for
    // both: its per-curve path runs after the span records are done with)
    __shared__ __attribute__((aligned(16))) f4 s_stage[kSpanLdsF4];
    static_assert(kSpanLdsF4 * 16 >= kCurveLdsKeys * 20, "the per-curve staging fits the span area");
    constexpr uint32_t kCurveF4 = (kCurveLdsKeys * 20 + 15) / 16;
    __shared__ __attribute__((aligned(16))) f4 s_curves[BLOCK == 64 ? 1 : (BLOCK / 64) * kCurveF4];
    f4* s_own = BLOCK == 64 ? s_stage : s_curves + wave * kCurveF4;
    float val[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    // what the final store needs to know about the node: its present bits, and for the rotation wave whether / how it is keyed
    uint32_t present = 0;
    bool rot_valid = false;
    int rot_kind = -1;
    bool sampled = false;                              // this lane's values are in val[]
    bool described = false;                            // (wave-uniform) present / rot_valid / rot_kind are known

    // The track's SPAN RECORDS first (round 3; TrackHot: curves with the same key times -- per span the two locations and both
    // keys of every curve), found through the animator's descriptor of this (animation, node, binding) in ONE scalar load and
    // staged once per wave in one round trip; Curve::value_at is then decided once per instance for the three or four curves
    // (span_track_value_at).  Needs every hint of the track to be the same (they are unless a caller set them apart); what is
    // left -- those lanes, tracks without span records, records that do not fit the staging area -- takes the per-curve path below.
    if (f.crowd) {
        const CrowdDesc d = f.crowd[((size_t)a * f.n_nodes + node) * 3 + (uint32_t)bind];      // wave-uniform: scalar loads
        present = d.present;
        rot_valid = bind == FYX_BIND_ROTATION && d.valid;
        rot_kind = d.kind;
        described = true;
        if (!d.valid) {
            sampled = true;                            // fetch() -> None: nothing to sample, the record's part is zeros
        } else {
            const uint32_t stride = span_stride(d.need), n = d.n_keys;
            const uint32_t n_f4 = n >= 2u ? (n - 1u) * stride : 0u;
            if (d.spans && n_f4 && n_f4 <= kSpanLdsF4) {
                uint32_t h0[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (c < (int)d.need && active) h0[c] = *hint_ptr(f, a, d.track, (uint32_t)c, inst);
                {   // all of a lane's loads go out before its first LDS write: one round trip, not one per 1 KB
                    const f4* gs = reinterpret_cast<const f4*>(d.spans);
                    f4 t[kSpanLdsF4 / BLOCK];
#pragma unroll
                    for (uint32_t k = 0; k < kSpanLdsF4 / BLOCK; ++k) {
                        const uint32_t i = threadIdx.x + k * BLOCK;
                        t[k] = gs[i < n_f4 ? i : 0u];
                    }
#pragma unroll
                    for (uint32_t k = 0; k < kSpanLdsF4 / BLOCK; ++k) {
                        const uint32_t i = threadIdx.x + k * BLOCK;
                        if (i < n_f4) s_stage[i] = t[k];
                    }
                }
                if constexpr (BLOCK == 64) __builtin_amdgcn_wave_barrier();
                else __syncthreads();                 // (d is the same for the whole workgroup: every wave gets here)
                const uint32_t h = h0[0];
                const bool one_hint = h0[1] == h && h0[2] == h && (d.need < 4u || h0[3] == h);
                if (active && one_hint) {
                    const uint32_t new_hint = span_track_value_at((const __attribute__((address_space(3))) f4*)s_stage, n, stride, (int)d.need, time, h, val);
                    if (new_hint != h) {
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if (c < (int)d.need) *hint_ptr(f, a, d.track, (uint32_t)c, inst) = new_hint;
                    }
                    sampled = true;
                }
            }
        }
    }
    const bool general = active && !sampled;          // everything else, decided in the reference's order on the per-curve records
    if (!described || __any(general)) {
        const AnimDev an = f.anims[a];
        const int32_t* st = an.slot_track + (size_t)node * 4;     // wave-uniform: scalar loads
        // which bindings the animation provides for this node (all three: the position wave writes the present bits)
        bool valid[3];
        int kinds[3];
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            valid[b] = false;
            kinds[b] = -1;
            if (st[b] < 0) continue;
            const TrackDev* tk = an.tracks + st[b];
            const int kind = tk->kind;
            const int need = kind == FYX_KIND_QUAT ? 4 : (kind == FYX_KIND_VEC3 || kind == FYX_KIND_QUAT_EULER) ? 3 : 0;
            const bool fits = (b == FYX_BIND_ROTATION) ? (kind == FYX_KIND_QUAT || kind == FYX_KIND_QUAT_EULER)
                                                       : (kind == FYX_KIND_VEC3);
            valid[b] = fits && need > 0 && (int)tk->n_curves >= need;   // else fetch() -> None
            kinds[b] = kind;
        }
        present = (valid[FYX_BIND_POSITION] ? 1u : 0u) | (valid[FYX_BIND_SCALE] ? 2u : 0u) | (valid[FYX_BIND_ROTATION] ? 4u : 0u) | (st[3] >= 0 ? 8u : 0u);
        rot_valid = bind == FYX_BIND_ROTATION && valid[FYX_BIND_ROTATION];
        rot_kind = kinds[FYX_BIND_ROTATION];
        if (valid[bind]) {
            // (fetching the hints and keys of all four curves up front was measured slower: 44 vs 37 us on C3)
            const int32_t track = st[bind];
            const TrackDev* tk = an.tracks + track;
            const int need = kinds[bind] == FYX_KIND_QUAT ? 4 : 3;
            // the hints of the track's curves are independent of each other: one round trip for all of them
            uint32_t h0[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < need && general) h0[c] = *hint_ptr(f, a, (uint32_t)track, (uint32_t)c, inst);
            // The 64 instances of the wave sit at 64 different playback positions, so a key load from global memory is a
            // gather of 64 different addresses -- and the texture addresser takes about a cycle per address (measured:
            // ~58 cycles per such instruction and CU, five of them per curve, is what this kernel's time was made of).
            // A curve is small (20 B per key), so the wave copies the WHOLE curve into LDS with dense loads instead and
            // gathers from there, where 64 different addresses cost a few cycles.  Longer curves keep the global path.
            f4* s_aux = s_own;
            float* s_loc = reinterpret_cast<float*>(s_own + kCurveLdsKeys);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (c >= need) break;
                uint32_t hint = h0[c];
                const uint32_t fk = tk->first_key[c], nk = tk->n_keys[c];
                const float* gl = an.key_loc + fk;
                const f4* ga = reinterpret_cast<const f4*>(an.key_aux) + fk;
                if (nk <= kCurveLdsKeys) {   // wave-uniform
                    __builtin_amdgcn_wave_barrier();                // every lane is done with the previous curve
                    for (uint32_t i = lane; i < nk; i += 64u) { s_loc[i] = gl[i]; s_aux[i] = ga[i]; }
                    __builtin_amdgcn_wave_barrier();
                    if (general)
                        val[c] = curve_value_at<false>((const __attribute__((address_space(3))) float*)s_loc,
                                                       (const __attribute__((address_space(3))) f4*)s_aux, nk, curve_ends(tk, c), time, hint);
                } else if (general) {
                    val[c] = curve_value_at<false>(gl, ga, nk, curve_ends(tk, c), time, hint);
                }
                if (general && hint != h0[c]) *hint_ptr(f, a, (uint32_t)track, (uint32_t)c, inst) = hint;
            }
        }
    }
    if (!active) return;
    f4* rec = reinterpret_cast<f4*>(f.anim_pose) + (((size_t)a * f.n_instances + inst) * f.n_nodes + node) * 3;
    if (bind == FYX_BIND_POSITION) {
        rec[0] = f4{val[0], val[1], val[2], __uint_as_float(present)};
    } else if (bind == FYX_BIND_SCALE) {
        rec[2] = f4{val[0], val[1], val[2], 0.0f};
    } else {
        f4 q = f4{0.f, 0.f, 0.f, 1.f};
        if (rot_valid) {
            if (rot_kind == FYX_KIND_QUAT) {
                q = quat_normalize(f4{val[0], val[1], val[2], val[3]});
            } else {   // (axis * sin(angle/2), cos(angle/2)) per axis, then qz * qy * qx (fyrox-math/src/lib.rs:725-740)
                float sx, cx, sy, cy, sz, cz;
                sincosf(val[0] / 2.0f, &sx, &cx);
                sincosf(val[1] / 2.0f, &sy, &cy);
                sincosf(val[2] / 2.0f, &sz, &cz);
                const f4 qx = f4{1.0f * sx, 0.0f * sx, 0.0f * sx, cx};
                const f4 qy = f4{0.0f * sy, 1.0f * sy, 0.0f * sy, cy};
                const f4 qz = f4{0.0f * sz, 0.0f * sz, 1.0f * sz, cz};
                q = quat_mul(quat_mul(qz, qy), qx);
            }
        }
        rec[1] = q;
    }
}

template <uint32_t BLOCK>
__global__ __launch_bounds__(BLOCK) void pose_sample_crowd_kernel(PoseFrameDev f, CtrlInline inl) { pose_sample_crowd_body<BLOCK>(ctrl_resolve<kInlAfterFrame>(f, inl), blockIdx.x, blockIdx.y, blockIdx.z); }

__global__ __launch_bounds__(64) void pose_sample_crowd_scene_kernel(const SceneJobDev* __restrict__ jobs, const char* __restrict__ ctrl, const uint4* __restrict__ blocks) {
    const uint4 b = blocks[blockIdx.x];
    const PoseFrameDev f = scene_frame_of(jobs, b.x, ctrl);
    pose_sample_crowd_body<64>(f, b.y, b.z, b.w);
}

hipError_t launch_pose_sample(const PoseFrameDev& f, hipStream_t s, const CtrlInline* inl) {
    if (f.n_anims == 0 || f.n_instances == 0 || f.n_nodes == 0) return hipSuccess;
    if (f.n_instances > 65535u || f.n_anims > 65535u || f.n_nodes * 3u > 65535u) return hipErrorInvalidValue;   // grid limits
    if (f.sample_form == 2 || (f.sample_form == 0 && f.n_instances >= 32)) {
        if (f.n_instances > 64u)
            FYX_TL_LAUNCH(pose_sample_crowd_kernel<256>, dim3((f.n_instances + 255) / 256, f.n_nodes * 3, f.n_anims), dim3(256), 0, s, f, inl ? *inl : kNoInline);
        else
            hipLaunchKernelGGL(pose_sample_crowd_kernel<64>, dim3(1, f.n_nodes * 3, f.n_anims), dim3(64), 0, s, f, inl ? *inl : kNoInline);
    } else {
        FYX_TL_LAUNCH(pose_sample_kernel, dim3((f.n_nodes * 16 + 255) / 256, f.n_instances, f.n_anims), dim3(256), 0, s, f, inl ? *inl : kNoInline);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// Root motion.
//
// root_motion_kernel = Animation::update_root_motion (fyrox-animation/src/lib.rs:498-661) for every
// ticked animation with RootMotionSettings: sixteen lanes per (animation, instance), ONE LANE PER
// CURVE SAMPLE as in pose_sample.  The reference fetches the first Position / Rotation track of the
// tracks data at cycle_start_time, cycle_end_time and time_slice.start, all of which are one of
// {time_slice.start, time_slice.end}: lanes 0..7 sample at the slice start, lanes 8..15 at the slice
// end (within each half: 0..2 position xyz, 4..7 rotation), with fresh hints as HintContainer::
// default() gives.  Lane 0 then runs the scalar update against the root node's sampled pose record,
// stores the animation's RootMotion and rewrites the record (the root stops moving).
//
// root_motion_fold_kernel: one thread per instance runs the instance's root-motion program (the
// order in which the reference's pose nodes / layers / machine called clone_into and blend_with on
// each other's poses, recorded by the host control plane) over the persistent slots.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void root_motion_body(const PoseFrameDev& f, uint32_t bx, uint32_t n_blocks) {
    const uint32_t lane = threadIdx.x & 63u, j = threadIdx.x & 15u, gbase = lane & ~15u;
    const uint32_t items = f.n_anims * f.n_instances;
    const uint32_t groups_per_pass = (n_blocks * blockDim.x) >> 4;
    for (uint32_t item = (bx * blockDim.x + threadIdx.x) >> 4; item < items; item += groups_per_pass) {
        const uint32_t a = item / f.n_instances, inst = item - a * f.n_instances;
        const uint32_t flags = f.ticked[(size_t)inst * f.n_anims + a];
        const AnimDev an = f.anims[a];
        if (!(flags & 1u) || an.rm_node < 0) continue;  // uniform across the group
        const float2 slice = f.slices[(size_t)inst * f.n_anims + a];
        const uint32_t half = j >> 3, c8 = j & 7u;
        const float time = half ? slice.y : slice.x;
        int32_t track = -1;
        int c = 0;
        if (c8 < 3) { track = an.rm_pos_track; c = (int)c8; }
        else if (c8 >= 4) { track = an.rm_rot_track; c = (int)c8 - 4; }
        int kind = -1;
        bool valid = false;
        float v = 0.0f;
        if (track >= 0) {
            const TrackDev* tk = an.tracks + track;
            kind = tk->kind;
            const int need = kind == FYX_KIND_QUAT ? 4 : (kind == FYX_KIND_VEC3 || kind == FYX_KIND_QUAT_EULER) ? 3 : 0;
            const bool fits = (c8 >= 4) ? (kind == FYX_KIND_QUAT || kind == FYX_KIND_QUAT_EULER) : (kind == FYX_KIND_VEC3);
            valid = fits && need > 0 && (int)tk->n_curves >= need;  // else .and_then(..) -> unwrap_or_default()
            if (valid && c < need) {
                uint32_t hint = 0;
                const uint32_t fk = tk->first_key[c];
                v = curve_value_at(an.key_loc + fk, reinterpret_cast<const f4*>(an.key_aux) + fk, tk->n_keys[c], curve_ends(tk, c), time, hint);
            }
        }
        const int sub = (int)(gbase + half * 8u);
        const int has_p = __shfl((int)valid, sub, 64);
        const int has_r = __shfl((int)valid, sub + 4, 64);
        const int rkind = __shfl(kind, sub + 4, 64);
        const f4 q = group_rotation(v, has_r, rkind, sub + 4);   // identity when the fetch fails
        const float pv = has_p ? v : 0.0f;                        // Vector3::default()
        // gather both halves on every lane
        float ps[3], pe[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            ps[k] = __shfl(pv, (int)gbase + k, 64);
            pe[k] = __shfl(pv, (int)gbase + 8 + k, 64);
        }
        f4 qs, qe;
        qs.x = __shfl(q.x, (int)gbase, 64); qs.y = __shfl(q.y, (int)gbase, 64);
        qs.z = __shfl(q.z, (int)gbase, 64); qs.w = __shfl(q.w, (int)gbase, 64);
        qe.x = __shfl(q.x, (int)gbase + 8, 64); qe.y = __shfl(q.y, (int)gbase + 8, 64);
        qe.z = __shfl(q.z, (int)gbase + 8, 64); qe.w = __shfl(q.w, (int)gbase + 8, 64);
        if (j != 0) continue;

        const bool new_loop = (flags & 2u) != 0, fwd = (flags & 4u) != 0;
        RootMotionDev* st = f.rm_anim + (size_t)a * f.n_instances + inst;
        RootMotionDev prev = *st;
        if (!prev.has) {  // self.root_motion.clone().unwrap_or_default()
            prev = RootMotionDev{};
            prev.delta_rotation[3] = 1.0f;
            prev.prev_rotation[3] = 1.0f;
        }
        RootMotionDev rm = RootMotionDev{};
        rm.has = 1u;
        rm.delta_rotation[3] = 1.0f;
        rm.prev_rotation[3] = 1.0f;
        f4* rec = reinterpret_cast<f4*>(f.anim_pose) + ((size_t)a * f.n_instances * f.n_nodes + (size_t)inst * f.n_nodes + (uint32_t)an.rm_node) * 3;
        f4 r0 = rec[0], r1 = rec[1];
        const uint32_t present = __float_as_uint(r0.w);
        if (present & 1u) {  // ValueBinding::Position / TrackValue::Vector3
            const float p[3] = {r0.x, r0.y, r0.z};
            const float* cyc_start = fwd ? ps : pe;   // speed > 0 ? time_slice.start : time_slice.end
            const float* cyc_end = fwd ? pe : ps;
            if (new_loop) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    rm.prev_position[k] = cyc_start[k];
                    rm.position_offset_remainder[k] = cyc_end[k] - p[k];
                }
                rm.rem_flags |= 1u;
            } else {
#pragma unroll
                for (int k = 0; k < 3; ++k) rm.prev_position[k] = p[k];
            }
            float delta[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float remainder = ((prev.rem_flags & 1u) && !(an.rm_ignore & 16u)) ? prev.position_offset_remainder[k] : 0.0f;  // .take().unwrap_or_default()
                const float current_offset = p[k] - prev.prev_position[k];
                delta[k] = current_offset + remainder;
            }
            rm.delta_position[0] = (an.rm_ignore & 1u) ? 0.0f : delta[0];
            rm.delta_position[1] = (an.rm_ignore & 2u) ? 0.0f : delta[1];
            rm.delta_position[2] = (an.rm_ignore & 4u) ? 0.0f : delta[2];
            r0.x = (an.rm_ignore & 1u) ? p[0] : ps[0];   // start_position = fetch(time_slice.start)
            r0.y = (an.rm_ignore & 2u) ? p[1] : ps[1];
            r0.z = (an.rm_ignore & 4u) ? p[2] : ps[2];
        }
        if ((present & 4u) && !(an.rm_ignore & 8u)) {  // ValueBinding::Rotation / UnitQuaternion
            const f4 pose_rotation = r1;
            const f4 cyc_start = fwd ? qs : qe, cyc_end = fwd ? qe : qs;
            auto conj = [](f4 x) { return f4{-x.x, -x.y, -x.z, x.w}; };  // UnitQuaternion::inverse
            if (new_loop) {
                const f4 rem = quat_mul(conj(cyc_end), pose_rotation);
                rm.prev_rotation[0] = cyc_start.x; rm.prev_rotation[1] = cyc_start.y;
                rm.prev_rotation[2] = cyc_start.z; rm.prev_rotation[3] = cyc_start.w;
                rm.rotation_remainder[0] = rem.x; rm.rotation_remainder[1] = rem.y;
                rm.rotation_remainder[2] = rem.z; rm.rotation_remainder[3] = rem.w;
                rm.rem_flags |= 2u;
            } else {
                rm.prev_rotation[0] = pose_rotation.x; rm.prev_rotation[1] = pose_rotation.y;
                rm.prev_rotation[2] = pose_rotation.z; rm.prev_rotation[3] = pose_rotation.w;
            }
            const f4 remainder = ((prev.rem_flags & 2u) && !(an.rm_ignore & 32u))
                ? f4{prev.rotation_remainder[0], prev.rotation_remainder[1], prev.rotation_remainder[2], prev.rotation_remainder[3]}
                : f4{0.f, 0.f, 0.f, 1.f};
            const f4 pp = f4{prev.prev_rotation[0], prev.prev_rotation[1], prev.prev_rotation[2], prev.prev_rotation[3]};
            const f4 current_relative_rotation = quat_mul(conj(pp), pose_rotation);
            const f4 d = quat_mul(remainder, current_relative_rotation);
            rm.delta_rotation[0] = d.x; rm.delta_rotation[1] = d.y; rm.delta_rotation[2] = d.z; rm.delta_rotation[3] = d.w;
            r1 = qs;
        }
        *st = rm;
        rec[0] = r0;
        rec[1] = r1;
    }
}

// RootMotion::blend_with (lib.rs:340-343) on {delta_position, has}{delta_rotation} slot pairs.
__device__ __forceinline__ void rm_blend(f4& sp, f4& sr, f4 op, f4 orr, float w) {
    if (__float_as_uint(sp.w) == 0u) { sp = f4{0.f, 0.f, 0.f, __uint_as_float(1u)}; sr = f4{0.f, 0.f, 0.f, 1.f}; }
    if (__float_as_uint(op.w) == 0u) { op = f4{0.f, 0.f, 0.f, 0.f}; orr = f4{0.f, 0.f, 0.f, 1.f}; }
    const float omw = 1.0f - w;
    sp.x = sp.x * omw + op.x * w;
    sp.y = sp.y * omw + op.y * w;
    sp.z = sp.z * omw + op.z * w;
    f4 a = sr;
    if (dot4(a, orr) < 0.0f) a = f4{-a.x, -a.y, -a.z, -a.w};
    sr = quat_normalize(f4{a.x * omw + orr.x * w, a.y * omw + orr.y * w, a.z * omw + orr.z * w, a.w * omw + orr.w * w});
}

__device__ __forceinline__ void root_motion_fold_body(const PoseFrameDev& f, uint32_t bx) {
    const uint32_t inst = bx * blockDim.x + threadIdx.x;
    if (inst >= f.n_instances) return;
    f4* slots = reinterpret_cast<f4*>(f.rm_slots) + (size_t)inst * f.n_rm_slots * 2;
    uint32_t pc = f.rm_prog_off[inst];
    const uint32_t end = f.rm_prog_off[inst + 1];
    if (pc >= end) return;
    uint4 op = f.rm_ops[pc];
    while (true) {
        ++pc;
        const uint4 next = pc < end ? f.rm_ops[pc] : make_uint4(RM_END, 0, 0, 0);  // fetched ahead of the dependent slot traffic
        if (op.x == RM_END) break;
        f4* d = slots + (size_t)op.y * 2;
        if (op.x == RM_SET_ANIM) {
            const RootMotionDev* r = f.rm_anim + (size_t)op.z * f.n_instances + inst;
            const uint32_t has = r->has;
            d[0] = f4{r->delta_position[0], r->delta_position[1], r->delta_position[2], __uint_as_float(has ? 1u : 0u)};
            d[1] = f4{r->delta_rotation[0], r->delta_rotation[1], r->delta_rotation[2], r->delta_rotation[3]};
        } else {
            const f4* sl = slots + (size_t)op.z * 2;
            const f4 s0 = sl[0], s1 = sl[1];
            if (op.x == RM_COPY) {
                d[0] = s0;
                d[1] = s1;
            } else {  // RM_BLEND
                f4 d0 = d[0], d1 = d[1];
                rm_blend(d0, d1, s0, s1, __uint_as_float(op.w));
                d[0] = d0;
                d[1] = d1;
            }
        }
        op = next;
    }
}

__global__ __launch_bounds__(256) void root_motion_kernel(PoseFrameDev f, CtrlInline inl) { root_motion_body(ctrl_resolve<kInlAfterFrame>(f, inl), blockIdx.x, gridDim.x); }
__global__ __launch_bounds__(64) void root_motion_fold_kernel(PoseFrameDev f, CtrlInline inl) { root_motion_fold_body(ctrl_resolve<kInlAfterFrame>(f, inl), blockIdx.x); }

__global__ __launch_bounds__(256) void root_motion_scene_kernel(const SceneJobDev* __restrict__ jobs, const char* __restrict__ ctrl, const uint4* __restrict__ blocks) {
    const uint4 b = blocks[blockIdx.x];          // {job, block of the job, blocks of the job, -}
    const PoseFrameDev f = scene_frame_of(jobs, b.x, ctrl);
    root_motion_body(f, b.y, b.z);
}
__global__ __launch_bounds__(64) void root_motion_fold_scene_kernel(const SceneJobDev* __restrict__ jobs, const char* __restrict__ ctrl, const uint4* __restrict__ blocks) {
    const uint4 b = blocks[blockIdx.x];
    const PoseFrameDev f = scene_frame_of(jobs, b.x, ctrl);
    root_motion_fold_body(f, b.y);
}

hipError_t launch_root_motion(const PoseFrameDev& f, bool run_program, hipStream_t s, const CtrlInline* inl) {
    const uint64_t items = (uint64_t)f.n_anims * f.n_instances;
    if (items && f.rm_anim) {
        uint64_t grid = (items * 16 + 255) / 256;
        if (grid > (uint64_t)kCUs * 16) grid = (uint64_t)kCUs * 16;
        hipLaunchKernelGGL(root_motion_kernel, dim3((uint32_t)grid), dim3(256), 0, s, f, inl ? *inl : kNoInline);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    if (run_program && f.rm_slots && f.rm_ops && f.n_instances) {
        hipLaunchKernelGGL(root_motion_fold_kernel, dim3((f.n_instances + 63) / 64), dim3(64), 0, s, f, inl ? *inl : kNoInline);
        return hipGetLastError();
    }
    return hipSuccess;
}

// ---------------------------------------------------------------------------------------
// Fold interpreter
// ---------------------------------------------------------------------------------------
struct Acc {
    float px, py, pz, sx, sy, sz;
    f4 r;
    uint32_t mask;
};

__device__ __forceinline__ Acc load_rec(const f4* __restrict__ rec) {
    const f4 a = rec[0], b = rec[1], c = rec[2];
    Acc o;
    o.px = a.x; o.py = a.y; o.pz = a.z;
    o.mask = __float_as_uint(a.w);
    o.r = b;
    o.sx = c.x; o.sy = c.y; o.sz = c.z;
    return o;
}

// NodePose::blend_with (pose.rs:41-47): an empty self becomes a copy of other (weight ignored);
// otherwise every value of self with a same-binding value in other is blended
// (value.rs:438-444), values only in other are dropped.
__device__ __forceinline__ void blend(Acc& self, const Acc& o, float w) {
    if (self.mask == 0) { self = o; return; }
    const uint32_t both = self.mask & o.mask;
    const float omw = 1.0f - w;  // nalgebra lerp: self * (1 - t) + rhs * t
    if (both & 1u) {
        self.px = self.px * omw + o.px * w;
        self.py = self.py * omw + o.py * w;
        self.pz = self.pz * omw + o.pz * w;
    }
    if (both & 2u) {
        self.sx = self.sx * omw + o.sx * w;
        self.sy = self.sy * omw + o.sy * w;
        self.sz = self.sz * omw + o.sz * w;
    }
    if (both & 4u) {  // value.rs:449-454 nlerp: flip self when dot < 0, then normalize(lerp)
        f4 a = self.r;
        if (dot4(a, o.r) < 0.0f) a = f4{-a.x, -a.y, -a.z, -a.w};
        const f4 l = f4{a.x * omw + o.r.x * w, a.y * omw + o.r.y * w, a.z * omw + o.r.z * w,
                        a.w * omw + o.r.w * w};
        self.r = quat_normalize(l);
    }
}

// The values of `self` that have a same-binding value in `o`, blended; no copy of an empty self (the DUP fold decides that for
// both of its records at once).
__device__ __forceinline__ void blend_values(Acc& self, const Acc& o, float w) {
    const uint32_t keep = self.mask;
    if (keep == 0) return;
    blend(self, o, w);          // (self.mask != 0: the value branch)
    self.mask = keep;
}

struct FoldCtx {
    const uint2* __restrict__ ops;
    uint2 my_op;                          // op `lane` of the program (END beyond its end): ops 0..63 come from registers
    uint32_t n_ops;
    const f4* __restrict__ anim_pose;     // base of [n_anims][n_instances][n_nodes][3]
    const uint8_t* __restrict__ layer_masks;
    size_t anim_stride;                   // records between animations = n_instances * n_nodes
    size_t rec_index;                     // inst * n_nodes + node
    uint32_t n_nodes, node;
    uint32_t pc;
    float pop_w;
    bool done;
    // node transform being written (BoundValueCollectionExt::apply)
    float tpx, tpy, tpz, tsx, tsy, tsz;
    f4 tr;
    bool dirty;
};

__device__ __forceinline__ void apply_pose(FoldCtx& cx, const Acc& a) {
    if (a.mask & 1u) { cx.tpx = a.px; cx.tpy = a.py; cx.tpz = a.pz; }
    if (a.mask & 2u) { cx.tsx = a.sx; cx.tsy = a.sy; cx.tsz = a.sz; }
    if (a.mask & 4u) cx.tr = a.r;
    cx.dirty |= a.mask != 0;
}

// The program is the same for every thread of the workgroup (one instance), so its first 64 ops sit in the lanes of a
// register pair (one load at kernel start) and op k is a v_readlane away -- no dependent scalar load per op.
__device__ __forceinline__ uint2 fold_op(const FoldCtx& cx, uint32_t pc) {
    if (pc < 64u) {
        uint2 op;
        op.x = (uint32_t)__builtin_amdgcn_readlane((int)cx.my_op.x, (int)pc);
        op.y = (uint32_t)__builtin_amdgcn_readlane((int)cx.my_op.y, (int)pc);
        return op;
    }
    // (the program is the same for every lane: say so, or the interpreter's control flow and counters are compiled as divergent)
    uint2 op = pc < cx.n_ops ? cx.ops[pc] : make_uint2(OP_END, 0u);
    op.x = (uint32_t)__builtin_amdgcn_readfirstlane((int)op.x);
    op.y = (uint32_t)__builtin_amdgcn_readfirstlane((int)op.y);
    return op;
}

__device__ __forceinline__ Acc acc_empty() {
    Acc a;
    a.px = a.py = a.pz = a.sx = a.sy = a.sz = 0.f;
    a.r = f4{0.f, 0.f, 0.f, 1.f};
    a.mask = 0;
    return a;
}

template <int D>
__device__ __forceinline__ void run_fold(FoldCtx& cx, Acc& acc) {
    for (;;) {
        const uint2 op = fold_op(cx, cx.pc++);
        const uint32_t code = op.x & 0xffu, arg = op.x >> 8;
        const float w = __uint_as_float(op.y);
        switch (code) {
            case OP_BLEND_ANIM: {
                const Acc o = load_rec(cx.anim_pose + ((size_t)arg * cx.anim_stride + cx.rec_index) * 3);
                blend(acc, o, w);
                break;
            }
            case OP_PUSH:
                if constexpr (D + 1 < kMaxFoldDepth) {
                    Acc child;
                    child.px = child.py = child.pz = child.sx = child.sy = child.sz = 0.f;
                    child.r = f4{0.f, 0.f, 0.f, 1.f};
                    child.mask = 0;
                    run_fold<D + 1>(cx, child);
                    if (cx.done) return;
                    blend(acc, child, cx.pop_w);
                } else {
                    cx.done = true;  // deeper than the host ever emits (validated at build time)
                    return;
                }
                break;
            case OP_POP_BLEND:
                cx.pop_w = w;
                return;
            case OP_RESET:
                acc.mask = 0;
                break;
            case OP_MASK:
                if (cx.layer_masks[(size_t)arg * cx.n_nodes + cx.node]) acc.mask = 0;
                break;
            case OP_APPLY:
                apply_pose(cx, acc);
                break;
            case OP_APPLY_ANIM: {
                const Acc o = load_rec(cx.anim_pose + ((size_t)arg * cx.anim_stride + cx.rec_index) * 3);
                apply_pose(cx, o);
                break;
            }
            default:  // OP_END
                cx.done = true;
                return;
        }
    }
}

// The fold over animations whose node poses hold SEVERAL values of one binding (PoseFrameDev::shadows).  In the reference such a list
// is blended value by value against the FIRST same-binding value of the other pose (value.rs:438-444) and applied in order, the last
// applicable value winning (scene/animation/mod.rs:147-186); a value whose kind does not fit its binding blends with nothing and is
// never applied, but it is a value: the list is not empty and nothing is copied over it (pose.rs:41-47).  Every value of a list goes
// its own way through the blends, so two of them are all that can ever be observed: `a`, the value that will be APPLIED, and `f`, the
// one a later blend READS when this pose is the `other` -- each a whole record.  Emptiness is the list's: f's mask (bit 16: values that
// fit no binding).  Same ops, same order as run_fold.
struct AccPair { Acc a, f; };
__device__ __forceinline__ void blend_dup(AccPair& self, const Acc& oa, const Acc& of, float w) {
    if (self.f.mask == 0) { self.a = oa; self.f = of; return; }      // NodePose::blend_with: an empty pose becomes a copy
    blend_values(self.a, of, w);
    blend_values(self.f, of, w);
}

template <int D>
__device__ __forceinline__ void run_fold_dup(FoldCtx& cx, AccPair& acc) {
    for (;;) {
        const uint2 op = fold_op(cx, cx.pc++);
        const uint32_t code = op.x & 0xffu, arg = op.x >> 8;     // (arg: the animation's first device record, 2 a)
        const float w = __uint_as_float(op.y);
        switch (code) {
            case OP_BLEND_ANIM: {
                const Acc oa = load_rec(cx.anim_pose + ((size_t)arg * cx.anim_stride + cx.rec_index) * 3);
                const Acc of = load_rec(cx.anim_pose + ((size_t)(arg + 1u) * cx.anim_stride + cx.rec_index) * 3);
                blend_dup(acc, oa, of, w);
                break;
            }
            case OP_PUSH:
                if constexpr (D + 1 < kMaxFoldDepth) {
                    AccPair child;
                    child.a = acc_empty();
                    child.f = acc_empty();
                    run_fold_dup<D + 1>(cx, child);
                    if (cx.done) return;
                    blend_dup(acc, child.a, child.f, cx.pop_w);
                } else {
                    cx.done = true;
                    return;
                }
                break;
            case OP_POP_BLEND:
                cx.pop_w = w;
                return;
            case OP_RESET:
                acc.a.mask = acc.f.mask = 0;
                break;
            case OP_MASK:
                if (cx.layer_masks[(size_t)arg * cx.n_nodes + cx.node]) acc.a.mask = acc.f.mask = 0;
                break;
            case OP_APPLY:
                apply_pose(cx, acc.a);
                break;
            case OP_APPLY_ANIM: {
                const Acc o = load_rec(cx.anim_pose + ((size_t)arg * cx.anim_stride + cx.rec_index) * 3);
                apply_pose(cx, o);
                break;
            }
            default:  // OP_END
                cx.done = true;
                return;
        }
    }
}

// ---------------------------------------------------------------------------------------
// Property{..} bindings (morph-target weights, and any other numeric property the animation editor can key).
//
// In the reference such a value is one more BoundValue in its node's pose (pose.rs:107-121), blended by
// TrackValue::blend_with -- Real => lerpf, Vector2/3/4 => nalgebra lerp, UnitQuaternion => nlerp, different variants
// => no-op (value.rs:221-230) -- and written through reflection after a numeric cast to the property's machine type
// (value.rs:232-427; the cast is the Rust shim's: it gets the f32 lanes and the value's variant).  Here every
// (node, property) pair of an animator is a SLOT with its own thread.  What couples a slot to the rest of its node
// is NodePose::blend_with's rule "an empty node pose becomes a copy of the other" (pose.rs:41-47): emptiness is a
// property of the whole node, so bit 3 of a pose record's present bits says "this animation holds a Property value
// for the node" (the node threads of pose_update see it in their mask), and a slot thread carries the node's mask
// along with its own {value[4], variant, present} through the same fold program.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void property_sample_body(const PoseFrameDev& f, uint32_t bx, uint32_t by, uint32_t bz) {
    const uint32_t slot = bx * 256u + threadIdx.x, inst = by, a = bz;
    if (slot >= f.n_prop_slots) return;
    if (!(f.ticked[(size_t)inst * f.n_anims + a] & 1u)) return;
    const AnimDev an = f.anims[a];
    const int32_t track = an.prop_track[slot];
    PropRec out;
    out.v[0] = out.v[1] = out.v[2] = out.v[3] = 0.0f;
    out.present = out.kind = out.pad[0] = out.pad[1] = 0u;
    if (track >= 0) {
        // TrackDataContainer::fetch (container.rs:182-297): 1 / 2 / 3 / 4 curves for Real / Vector2 / Vector3 / Vector4,
        // 3 Euler angles -> qz * qy * qx, 4 components -> normalised quaternion; None when curves are missing
        const TrackDev* tk = an.tracks + track;
        const int kind = tk->kind;
        const int need = kind == FYX_KIND_REAL ? 1 : kind == FYX_KIND_VEC2 ? 2 : (kind == FYX_KIND_VEC3 || kind == FYX_KIND_QUAT_EULER) ? 3 : 4;
        if ((int)tk->n_curves >= need) {
            const float time = f.times[(size_t)inst * f.n_anims + a];
            float val[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (c >= need) break;
                uint32_t* hp = hint_ptr(f, a, (uint32_t)track, (uint32_t)c, inst);
                uint32_t hint = *hp;
                const uint32_t fk = tk->first_key[c];
                val[c] = curve_value_at(an.key_loc + fk, reinterpret_cast<const f4*>(an.key_aux) + fk, tk->n_keys[c], curve_ends(tk, c), time, hint);
                *hp = hint;
            }
            if (kind == FYX_KIND_QUAT) {
                const f4 q = quat_normalize(f4{val[0], val[1], val[2], val[3]});
                val[0] = q.x; val[1] = q.y; val[2] = q.z; val[3] = q.w;
            } else if (kind == FYX_KIND_QUAT_EULER) {
                float sx, cx, sy, cy, sz, cz;
                sincosf(val[0] / 2.0f, &sx, &cx);
                sincosf(val[1] / 2.0f, &sy, &cy);
                sincosf(val[2] / 2.0f, &sz, &cz);
                const f4 qx = f4{1.0f * sx, 0.0f * sx, 0.0f * sx, cx};
                const f4 qy = f4{0.0f * sy, 1.0f * sy, 0.0f * sy, cy};
                const f4 qz = f4{0.0f * sz, 0.0f * sz, 1.0f * sz, cz};
                const f4 q = quat_mul(quat_mul(qz, qy), qx);
                val[0] = q.x; val[1] = q.y; val[2] = q.z; val[3] = q.w;
            }
            out.v[0] = val[0]; out.v[1] = val[1]; out.v[2] = val[2]; out.v[3] = val[3];
            out.present = 1u;
            out.kind = kind == FYX_KIND_QUAT_EULER || kind == FYX_KIND_QUAT ? (uint32_t)FYX_VALUE_QUAT : (uint32_t)kind;   // TrackValue variant
        }
    }
    f.prop_pose[((size_t)a * f.n_instances + inst) * f.n_prop_slots + slot] = out;
}

__global__ __launch_bounds__(256) void property_sample_kernel(PoseFrameDev f, CtrlInline inl) { property_sample_body(ctrl_resolve<kInlAfterFrame>(f, inl), blockIdx.x, blockIdx.y, blockIdx.z); }
__global__ __launch_bounds__(256) void property_sample_scene_kernel(const SceneJobDev* __restrict__ jobs, const char* __restrict__ ctrl, const uint4* __restrict__ blocks) {
    const uint4 b = blocks[blockIdx.x];
    const PoseFrameDev f = scene_frame_of(jobs, b.x, ctrl);
    property_sample_body(f, b.y, b.z, b.w);
}

hipError_t launch_property_sample(const PoseFrameDev& f, hipStream_t s, const CtrlInline* inl) {
    if (!f.n_prop_slots || !f.n_anims || !f.n_instances) return hipSuccess;
    hipLaunchKernelGGL(property_sample_kernel, dim3((f.n_prop_slots + 255) / 256, f.n_instances, f.n_anims), dim3(256), 0, s, f, inl ? *inl : kNoInline);
    return hipGetLastError();
}

struct PAcc {
    f4 v;
    uint32_t kind;        // TrackValue variant (FYX_VALUE_*)
    uint32_t present;     // this slot holds a value
    uint32_t node_mask;   // present bits of the whole node (0 == the node's pose is empty)
};

__device__ __forceinline__ PAcc pacc_empty() { return PAcc{f4{0.0f, 0.0f, 0.0f, 0.0f}, 0u, 0u, 0u}; }

// TrackValue::blend_with (value.rs:221-230)
__device__ __forceinline__ void pblend(PAcc& self, const PAcc& o, float w) {
    if (self.node_mask == 0) { self = o; return; }                      // copy, weight ignored
    if (!(self.present && o.present) || self.kind != o.kind) return;     // different variants: no-op
    const float omw = 1.0f - w;                                          // nalgebra lerp: self * (1 - t) + rhs * t
    switch (self.kind) {
        case FYX_VALUE_REAL: self.v.x = lerpf_(self.v.x, o.v.x, w); break;    // a + (b - a) * w
        case FYX_VALUE_VEC2:
            self.v.x = self.v.x * omw + o.v.x * w; self.v.y = self.v.y * omw + o.v.y * w;
            break;
        case FYX_VALUE_VEC3:
            self.v.x = self.v.x * omw + o.v.x * w; self.v.y = self.v.y * omw + o.v.y * w; self.v.z = self.v.z * omw + o.v.z * w;
            break;
        case FYX_VALUE_VEC4:
            self.v.x = self.v.x * omw + o.v.x * w; self.v.y = self.v.y * omw + o.v.y * w;
            self.v.z = self.v.z * omw + o.v.z * w; self.v.w = self.v.w * omw + o.v.w * w;
            break;
        default: {   // value.rs:449-454 nlerp: flip self when dot < 0, then normalize(lerp)
            f4 a = self.v;
            if (dot4(a, o.v) < 0.0f) a = f4{-a.x, -a.y, -a.z, -a.w};
            self.v = quat_normalize(f4{a.x * omw + o.v.x * w, a.y * omw + o.v.y * w, a.z * omw + o.v.z * w, a.w * omw + o.v.w * w});
            break;
        }
    }
}

struct PFoldCtx {
    const uint2* __restrict__ ops;
    const PropRec* __restrict__ prop_pose;     // [n_anims][n_instances][n_slots]
    const f4* __restrict__ anim_pose;          // node records (for the node's present bits)
    const uint8_t* __restrict__ layer_masks;
    size_t prop_stride, prop_index;            // n_instances * n_slots, inst * n_slots + slot
    size_t rec_stride, rec_index;              // n_instances * n_nodes, inst * n_nodes + node
    uint32_t n_nodes, node, pc;
    float pop_w;
    bool done;
    f4 out_v;
    uint32_t out_kind;
    bool out_set;
};

__device__ __forceinline__ PAcc pload(const PFoldCtx& cx, uint32_t a) {
    const f4* p = reinterpret_cast<const f4*>(cx.prop_pose + (size_t)a * cx.prop_stride + cx.prop_index);
    const f4 v = p[0], m = p[1];
    PAcc o;
    o.v = v;
    o.present = __float_as_uint(m.x);
    o.kind = __float_as_uint(m.y);
    o.node_mask = __float_as_uint(cx.anim_pose[((size_t)a * cx.rec_stride + cx.rec_index) * 3].w);
    return o;
}

template <int D>
__device__ __forceinline__ void run_fold_prop(PFoldCtx& cx, PAcc& acc) {
    for (;;) {
        const uint2 op = cx.ops[cx.pc++];
        const uint32_t code = op.x & 0xffu, arg = op.x >> 8;
        const float w = __uint_as_float(op.y);
        switch (code) {
            case OP_BLEND_ANIM: pblend(acc, pload(cx, arg), w); break;
            case OP_PUSH:
                if constexpr (D + 1 < kMaxFoldDepth) {
                    PAcc child = pacc_empty();
                    run_fold_prop<D + 1>(cx, child);
                    if (cx.done) return;
                    pblend(acc, child, cx.pop_w);
                } else {
                    cx.done = true;
                    return;
                }
                break;
            case OP_POP_BLEND: cx.pop_w = w; return;
            case OP_RESET: acc = pacc_empty(); break;
            case OP_MASK:
                if (cx.layer_masks[(size_t)arg * cx.n_nodes + cx.node]) acc = pacc_empty();
                break;
            case OP_APPLY:
                if (acc.present) { cx.out_v = acc.v; cx.out_kind = acc.kind; cx.out_set = true; }
                break;
            case OP_APPLY_ANIM: {
                const PAcc o = pload(cx, arg);
                if (o.present) { cx.out_v = o.v; cx.out_kind = o.kind; cx.out_set = true; }
                break;
            }
            default: cx.done = true; return;
        }
    }
}

// The same fold over two records per animation (PoseFrameDev::shadows; run_fold_dup above has the rule): `a` the slot's value that
// will be applied (the LAST of its list), `f` the one a blend reads of it (the FIRST).
template <int D>
__device__ __forceinline__ void run_fold_prop_dup(PFoldCtx& cx, PAcc& a, PAcc& f) {
    auto blend2 = [](PAcc& sa, PAcc& sf, const PAcc& oa, const PAcc& of, float w) {
        if (sf.node_mask == 0) { sa = oa; sf = of; return; }
        if (sa.node_mask != 0) pblend(sa, of, w);
        pblend(sf, of, w);
    };
    for (;;) {
        const uint2 op = cx.ops[cx.pc++];
        const uint32_t code = op.x & 0xffu, arg = op.x >> 8;
        const float w = __uint_as_float(op.y);
        switch (code) {
            case OP_BLEND_ANIM: blend2(a, f, pload(cx, arg), pload(cx, arg + 1u), w); break;
            case OP_PUSH:
                if constexpr (D + 1 < kMaxFoldDepth) {
                    PAcc ca = pacc_empty(), cf = pacc_empty();
                    run_fold_prop_dup<D + 1>(cx, ca, cf);
                    if (cx.done) return;
                    blend2(a, f, ca, cf, cx.pop_w);
                } else {
                    cx.done = true;
                    return;
                }
                break;
            case OP_POP_BLEND: cx.pop_w = w; return;
            case OP_RESET: a = pacc_empty(); f = pacc_empty(); break;
            case OP_MASK:
                if (cx.layer_masks[(size_t)arg * cx.n_nodes + cx.node]) { a = pacc_empty(); f = pacc_empty(); }
                break;
            case OP_APPLY:
                if (a.present) { cx.out_v = a.v; cx.out_kind = a.kind; cx.out_set = true; }
                break;
            case OP_APPLY_ANIM: {
                const PAcc o = pload(cx, arg);
                if (o.present) { cx.out_v = o.v; cx.out_kind = o.kind; cx.out_set = true; }
                break;
            }
            default: cx.done = true; return;
        }
    }
}

__device__ __forceinline__ void property_update_body(const PoseFrameDev& f, uint32_t bx, uint32_t by) {
    const uint32_t slot = bx * 64u + threadIdx.x, inst = by;
    if (slot >= f.n_prop_slots) return;
    PFoldCtx cx;
    cx.ops = f.ops + f.prog_off[inst];
    cx.prop_pose = f.prop_pose;
    cx.anim_pose = reinterpret_cast<const f4*>(f.anim_pose);
    cx.layer_masks = f.layer_masks;
    cx.prop_stride = (size_t)f.n_instances * f.n_prop_slots;
    cx.prop_index = (size_t)inst * f.n_prop_slots + slot;
    cx.n_nodes = f.n_nodes;
    cx.node = (uint32_t)f.prop_node[slot];
    cx.rec_stride = (size_t)f.n_instances * f.n_nodes;
    cx.rec_index = (size_t)inst * f.n_nodes + cx.node;
    cx.pc = 0;
    cx.pop_w = 0.f;
    cx.done = false;
    cx.out_v = f4{0.f, 0.f, 0.f, 0.f};
    cx.out_kind = 0u;
    cx.out_set = false;
    PAcc acc = pacc_empty();
    if (f.shadows) {
        PAcc accf = pacc_empty();
        while (!cx.done) run_fold_prop_dup<0>(cx, acc, accf);
    } else {
        while (!cx.done) run_fold_prop<0>(cx, acc);
    }
    if (cx.out_set) {
        f4* o = reinterpret_cast<f4*>(f.prop_out + (size_t)inst * f.n_prop_slots + slot);
        o[0] = cx.out_v;
        o[1] = f4{__uint_as_float(1u), __uint_as_float(cx.out_kind), 0.0f, 0.0f};
    }
}

__global__ __launch_bounds__(64) void property_update_kernel(PoseFrameDev f, CtrlInline inl) { property_update_body(ctrl_resolve<kInlAfterFrame>(f, inl), blockIdx.x, blockIdx.y); }
__global__ __launch_bounds__(64) void property_update_scene_kernel(const SceneJobDev* __restrict__ jobs, const char* __restrict__ ctrl, const uint4* __restrict__ blocks) {
    const uint4 b = blocks[blockIdx.x];
    const PoseFrameDev f = scene_frame_of(jobs, b.x, ctrl);
    property_update_body(f, b.y, b.z);
}

hipError_t launch_property_update(const PoseFrameDev& f, hipStream_t s, const CtrlInline* inl) {
    if (!f.n_prop_slots || !f.n_instances) return hipSuccess;
    hipLaunchKernelGGL(property_update_kernel, dim3((f.n_prop_slots + 63) / 64, f.n_instances), dim3(64), 0, s, f, inl ? *inl : kNoInline);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void blend_shape_weights_kernel(const PropRec* __restrict__ prop_out, uint32_t n_prop_slots,
                                                                  uint32_t n_instances, const int32_t* __restrict__ slots,
                                                                  const float* __restrict__ defaults, uint32_t n_shapes,
                                                                  float* __restrict__ out) {
    const uint32_t e = blockIdx.x * 256u + threadIdx.x;
    if (e >= n_instances * n_shapes) return;
    const uint32_t inst = e / n_shapes, k = e - inst * n_shapes;
    float w = defaults[k];
    const int32_t sl = slots[k];
    if (sl >= 0 && (uint32_t)sl < n_prop_slots) {
        const PropRec p = prop_out[(size_t)inst * n_prop_slots + sl];
        if (p.present && p.kind == FYX_VALUE_REAL) w = p.v[0];   // BlendShape::weight is an f32: only a Real value casts to it
    }
    out[e] = w / 100.0f;   // bs.weight / 100.0 (scene/mesh/mod.rs:797)
}

hipError_t launch_blend_shape_weights(const PropRec* prop_out, uint32_t n_prop_slots, uint32_t n_instances,
                                      const int32_t* d_slots, const float* d_defaults, uint32_t n_shapes, float* d_out,
                                      hipStream_t s) {
    const uint64_t total = (uint64_t)n_instances * n_shapes;
    if (!total) return hipSuccess;
    if (total > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(blend_shape_weights_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, s, prop_out,
                       n_prop_slots, n_instances, d_slots, d_defaults, n_shapes, d_out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// local matrix (Transform::calculate_local_transform, scene/transform.rs:421-540) and mat4 product
// ---------------------------------------------------------------------------------------
struct M3 { float m[9]; };  // column-major, m[col*3+row]

// UnitQuaternion::to_rotation_matrix
__device__ __forceinline__ M3 quat_to_m3(f4 q) {
    const float i = q.x, j = q.y, k = q.z, w = q.w;
    const float ww = w * w, ii = i * i, jj = j * j, kk = k * k;
    const float ij = i * j * 2.0f, wk = w * k * 2.0f, wj = w * j * 2.0f;
    const float ik = i * k * 2.0f, jk = j * k * 2.0f, wi = w * i * 2.0f;
    M3 o;
    o.m[0] = ww + ii - jj - kk; o.m[1] = wk + ij;           o.m[2] = ik - wj;
    o.m[3] = ij - wk;           o.m[4] = ww - ii + jj - kk; o.m[5] = wi + jk;
    o.m[6] = wj + ik;           o.m[7] = jk - wi;           o.m[8] = ww - ii - jj + kk;
    return o;
}

// Writes the 16 floats of the local matrix (column-major) for one node.
__device__ __forceinline__ void local_matrix(const float* __restrict__ st, float px, float py, float pz,
                                             f4 rot, float sx, float sy, float sz, float* out) {
    const f4 pre = f4{st[0], st[1], st[2], st[3]};
    const float* por = st + 4;                       // post_rotation_matrix, column-major 3x3
    const float rox = st[13], roy = st[14], roz = st[15];
    const float rpx = st[16], rpy = st[17], rpz = st[18];
    const float sox = st[19], soy = st[20], soz = st[21];
    const float spx = st[22], spy = st[23], spz = st[24];
    const M3 pr = quat_to_m3(pre), r = quat_to_m3(rot);
    float a[9], fm[9];
#pragma unroll
    for (int c = 0; c < 3; ++c)      // a = pr * r (columns of r)
#pragma unroll
        for (int row = 0; row < 3; ++row)
            a[c * 3 + row] = pr.m[row] * r.m[c * 3] + pr.m[3 + row] * r.m[c * 3 + 1] + pr.m[6 + row] * r.m[c * 3 + 2];
#pragma unroll
    for (int c = 0; c < 3; ++c)      // f = a * por  with por indexed as in the reference (por[3c+k])
#pragma unroll
        for (int row = 0; row < 3; ++row)
            fm[c * 3 + row] = por[c * 3] * a[row] + por[c * 3 + 1] * a[3 + row] + por[c * 3 + 2] * a[6 + row];
    const float s[3] = {sx, sy, sz};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        out[c * 4 + 0] = s[c] * fm[c * 3 + 0];
        out[c * 4 + 1] = s[c] * fm[c * 3 + 1];
        out[c * 4 + 2] = s[c] * fm[c * 3 + 2];
        out[c * 4 + 3] = 0.0f;
    }
    const float ro[3] = {rox, roy, roz}, rp[3] = {rpx, rpy, rpz}, t[3] = {px, py, pz};
#pragma unroll
    for (int row = 0; row < 3; ++row) {
        const float f0 = fm[row], f3 = fm[3 + row], f6 = fm[6 + row];
        const float k0 = spx * f0, k1 = spy * f3, k2 = spz * f6;
        out[12 + row] = ro[row] + rp[row] + t[row] - rpx * f0 - rpy * f3 - rpz * f6 + sox * f0 + k0 +
                        soy * f3 + k1 + soz * f6 + k2 - sx * k0 - sy * k1 - sz * k2;
    }
    out[15] = 1.0f;
}

// nalgebra Matrix4 * Matrix4: per result column an axpy chain over k (a_col_k * b_kj + y).
__device__ __forceinline__ void mat4_mul(const float* a, const float* b, float* out) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float y = a[i] * b[j * 4];
            y = a[4 + i] * b[j * 4 + 1] + y;
            y = a[8 + i] * b[j * 4 + 2] + y;
            y = a[12 + i] * b[j * 4 + 3] + y;
            out[j * 4 + i] = y;
        }
}

// ---------------------------------------------------------------------------------------
// pose_update: one workgroup per instance.
// ---------------------------------------------------------------------------------------
// An in-grid wait gave up: say so where the host looks (DeviceError, pinned host-coherent memory) -- plain system-scope stores, the
// code last; every workgroup that gives up writes the same kind of record, the last one stays.
__device__ __forceinline__ void report_frame_wait(const FrameSync& w, uint32_t seen) {
    if (!w.err) return;
    DeviceError* e = reinterpret_cast<DeviceError*>(w.err);
    __hip_atomic_store(&e->block, (uint32_t)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&e->seen, seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&e->target, w.target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&e->tag, w.tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&e->code, (uint32_t)kDevErrFrameWait, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// MODE: kUpdNoProgram -- the transforms as they are (fyx_animator_update_transforms); kUpdGeneral -- straight form + interpreter;
// kUpdStraight -- every program of the launch is straight (the host classified them with the same function,
// classify_fold_program): the interpreter and its kMaxFoldDepth nested accumulators are not compiled in, which is what lets a
// crowd's update kernel sit beside the previous frame's skinning (<= 128 VGPRs instead of ~440: anim.overlap).
// pal_mem: where the rig's palette outputs lie in memory, for callers whose RigDev is a register copy (the scene form: indexing a
// register copy with the loop counter would put the array in scratch).
// QUIET (a skinning workgroup of the one-launch frame, frame_skin_body): everything up to the global matrices in LDS, NOTHING stored to
// memory, then the palette of `skin`'s bone list straight into the skinning kernels' LDS layout (behind the update's own LDS areas).
// Returns false when an in-grid wait timed out (reported through FrameSync::err; nothing was computed).
// INV: the update workgroup's agent-scope acquire behind the wait (one character: by the book; a scene's one-launch frame has hundreds of
// update workgroups and leaves it out for the reason the skinning workgroups do, see the WAIT block)
// LATE (with QUIET): the palette's bone -> node words and inverse bind columns are requested behind the walk instead of at the top -- a
// scene's workgroups hide that latency behind one another, and twenty registers less across the fold are a wave more per SIMD.
template <int MODE, int PACK = 1, bool WAIT = false, bool WIDE = false, bool QUIET = false, bool INV = true, bool LATE = false>
__device__ __forceinline__ bool pose_update_body(const PoseFrameDev& f, const RigDev& rig, const uint32_t inst, const uint32_t first_ops = 0,
                                                 const PaletteOutDev* __restrict__ pal_mem = nullptr, const FrameSync* wait = nullptr,
                                                 const FrameSkinJob* skin = nullptr) {
    static_assert(!QUIET || (WIDE && PACK == 1), "the quiet form belongs to the 256-thread wide-walk launches");
    constexpr bool PROGRAM = MODE != kUpdNoProgram;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // PACK > 1: the workgroup is PACK independent waves, one instance each (rigs of <= 64 nodes: pose_update_pack_kernel) -- a
    // wave is then its own "workgroup": its index in it is the lane, its LDS area its own, its barrier the wave's program order
    const uint32_t tid = PACK > 1 ? (threadIdx.x & 63u) : threadIdx.x;
    const uint32_t bdim = PACK > 1 ? 64u : blockDim.x;
    // (WIDE: two more slots each -- n_nodes: the identity among the globals, n_nodes + 1: the padding's node -- see the walk)
    constexpr uint32_t kSlots = WIDE ? 2u : 0u;
    float* l_local = lds + (PACK > 1 ? (size_t)(threadIdx.x >> 6) * rig.n_nodes * 32 : 0);   // [n_nodes][16]
    float* l_global = l_local + (size_t)(rig.n_nodes + kSlots) * 16;                          // [n_nodes][16]
    auto sync = [] {
        if constexpr (PACK > 1) {    // (LDS operations of one wave execute in program order: only the compiler has to keep it)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else {
            __syncthreads();
        }
    };
    const size_t inst_base = (size_t)inst * rig.n_nodes;
    if constexpr (WAIT) FSTAMP(0);

    // Everything that does not depend on the fold is requested FIRST, so that its (cold) latency runs under the
    // fold's chain of dependent loads instead of after it: what this thread will do in the hierarchy walk (its <= 4
    // entries of the depth-sorted node list: node, level, parent -- so a level of the walk costs LDS traffic and a
    // barrier only) and, per node, the 112 bytes of the rig's static transform parts.
    constexpr int kEntries = WIDE ? 0 : PACK > 1 ? 1 : kMaxRigNodes / 256;   // launch_pose_update: block = min(256, n_nodes rounded up to 64); packed: <= 64 nodes on 64 lanes
    uint32_t w_node[kEntries ? kEntries : 1], w_level[kEntries ? kEntries : 1];
    int32_t w_par[kEntries ? kEntries : 1];
#pragma unroll
    for (int k = 0; k < kEntries; ++k) {
        const uint32_t i = tid + (uint32_t)k * bdim;
        w_level[k] = 0xffffffffu;
        w_node[k] = 0;
        w_par[k] = -1;
        if (i < rig.n_nodes) {
            const uint32_t e = rig.walk[i];            // node | (parent + 1) << 10 | depth << 21: one load, no chain
            w_node[k] = e & 1023u;
            w_par[k] = (int32_t)((e >> 10) & 2047u) - 1;
            w_level[k] = e >> 21;
        }
    }
    // WIDE (a workgroup of 256 threads for one character: the launches with the control block in their arguments): the walk below
    // gives every ELEMENT of a node's matrix a lane -- sixteen nodes of a level at a time -- so a level is five dependent VALU
    // instead of 112, and a lone wave issues an instruction every ~4 cycles whatever it is.  Its table (RigDev::walk's second
    // part: the host's chunks of sixteen entries, levels padded to whole chunks) is staged here, where nothing waits for it.
    uint32_t* s_chunks = reinterpret_cast<uint32_t*>(lds + (size_t)(rig.n_nodes + kSlots) * 32);   // [n_chunks + 1][16]
    if constexpr (WIDE) {
        const uint32_t* gc = rig.walk + rig.n_nodes;
        for (uint32_t i = threadIdx.x; i < rig.n_chunks * 16u; i += blockDim.x) s_chunks[i] = gc[i];
        if (threadIdx.x < 16u) {
            l_global[(size_t)rig.n_nodes * 16 + threadIdx.x] = (threadIdx.x % 5u == 0u) ? 1.0f : 0.0f;   // what a root is multiplied by
            l_local[(size_t)(rig.n_nodes + 1u) * 16 + threadIdx.x] = 0.0f;                               // the padding's "local matrix"
            s_chunks[rig.n_chunks * 16u + threadIdx.x] = 0u;                                            // (read ahead by the last chunk)
        }
    }
    // QUIET: the bone -> node words of the palette this workgroup builds (four columns per thread: <= 256 bones), requested with the
    // first loads of the kernel; the inverse bind columns they lead to follow just ahead of the wait for the samplers.
    int32_t q_node[4] = {-1, -1, -1, -1};
    f4 q_bb[4];
    if constexpr (QUIET && !LATE) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t e = threadIdx.x + (uint32_t)k * 256u;
            if (e < skin->n_bones * 4u) q_node[k] = skin->bone_nodes[e >> 2];
        }
    }
    // the instance's fold program: op `lane` into every wave's registers while ALL lanes are still active (v_readlane
    // reads a lane's register whatever EXEC says, but only active lanes load)
    const uint2* prog = nullptr;
    uint32_t n_ops = 0;
    uint2 my_op = make_uint2(OP_END, 0u);
    if constexpr (PROGRAM) {
        // (instance 0's program starts at op 0; where the launch knows its length -- CtrlInline::first_ops, one character's
        // frame -- the program is requested without a round trip for its offsets first)
        uint32_t p0 = 0;
        if (inst == 0 && first_ops) {
            n_ops = first_ops - 1u;
        } else {
            p0 = f.prog_off[inst];
            n_ops = f.prog_off[inst + 1] - p0;
        }
        prog = f.ops + p0;
        const uint32_t lane = tid & 63u;
        if (lane < n_ops) my_op = prog[lane];
    }
    // STRAIGHT programs ([PUSH^d] BLEND_ANIM^k [POP_BLEND^d] [MASK] APPLY END, k <= kStraightOps: anim_leaves.h) skip the
    // interpreter: the same calls to blend() in the same order as run_fold makes, without its dispatch (an op of it costs a
    // lone wave ~0.7 us, memory round trip included) and with all k operand records requested together.
    uint32_t st_d = 0, st_k = 0;
    bool straight = false, st_mask = false, st_player = false;
    if constexpr (PROGRAM) {
        const uint32_t code = my_op.x & 0xffu;
        const uint64_t m_push = __ballot(code == OP_PUSH), m_blend = __ballot(code == OP_BLEND_ANIM), m_pop = __ballot(code == OP_POP_BLEND);
        const uint64_t m_apply_anim = __ballot(code == OP_APPLY_ANIM);
        const StraightShape sh = classify_fold_program(n_ops, m_push, m_blend, m_pop, m_apply_anim, [&](uint32_t pc) -> uint32_t {
            return (uint32_t)__builtin_amdgcn_readlane((int)my_op.x, (int)(pc & 63u)) & 0xffu; });
        st_d = sh.d; st_k = sh.k; st_mask = sh.mask; st_player = sh.player;
        straight = sh.straight;
        if constexpr (MODE == kUpdStraight) straight = true;   // the host's promise (same classifier); no second path to fall into
        if constexpr (MODE == kUpdDup) straight = false;       // two records per animation: the interpreter's DUP form only
    }
    // Every lane of every wave walks the fold, also the lanes past the last node (they fold the last node's operand records onto
    // an identity transform and store nothing): fold_op reads the program out of the lanes' registers with v_readlane, and a lane that is inactive when
    // its register is read is undefined by the LLVM contract -- with a rig of 24 nodes and a program of 40 ops the ops
    // 24..39 would sit in lanes that a `node < n_nodes` loop has switched off.
    for (uint32_t node_base = 0; node_base < rig.n_nodes; node_base += bdim) {   // workgroup-uniform trip count
        const bool live = node_base + tid < rig.n_nodes;
        const uint32_t node = live ? node_base + tid : rig.n_nodes - 1;
        f4* trs = reinterpret_cast<f4*>(f.node_trs) + (inst_base + node) * 3;
        // a lane past the last node folds an identity transform, not the last node's record: the lane that owns that node (maybe in
        // another wave of the block) writes its folded TRS back below, and an unsynchronised read of it here would be a race
        f4 t0 = f4{0.f, 0.f, 0.f, 0.f}, t1 = f4{0.f, 0.f, 0.f, 1.f}, t2 = f4{1.f, 1.f, 1.f, 0.f};
        if (live) { t0 = trs[0]; t1 = trs[1]; t2 = trs[2]; }
        float st[28];
        {
            const f4* sp = reinterpret_cast<const f4*>(rig.statics + (size_t)node * 28);
#pragma unroll
            for (int q = 0; q < 7; ++q) {
                const f4 v = sp[q];
                st[q * 4] = v.x; st[q * 4 + 1] = v.y; st[q * 4 + 2] = v.z; st[q * 4 + 3] = v.w;
            }
        }
        if constexpr (QUIET && !LATE) {
            if (node_base == 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    q_bb[k] = f4{0.f, 0.f, 0.f, 0.f};
                    if (q_node[k] >= 0) q_bb[k] = reinterpret_cast<const f4*>(rig.inv_bind)[(size_t)q_node[k] * 4 + (threadIdx.x & 3u)];
                }
            }
        }
        if constexpr (WAIT) {
            // one-launch frame: the sampler's workgroups run in this grid too; everything above was requested without them
            if (node_base == 0) {
                FSTAMP(1);
                uint32_t* status = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(lds) + wide_update_lds(rig.n_nodes, rig.n_chunks));
                if (threadIdx.x == 0) {
                    // Bounded (FrameSync::timeout_ticks of the 100 MHz clock, option anim.wait_timeout_ms).  The samplers of this grid take
                    // microseconds and hold places of their own (fyx_internal.h: FrameSync); should the counter stay short all the same --
                    // a word of it overwritten from outside, a dispatcher that holds the samplers back -- the workgroup REPORTS and
                    // computes nothing: a frame is late or it is right, never silently made of stale records.
                    const uint64_t t0 = wall_clock64();
                    uint32_t seen = 0, ok = 1;
                    const uint32_t* word = wait->counter + (size_t)(blockIdx.x % kFrameCounterReplicas) * (kFrameCounterStride / 4u);
                    while ((int32_t)((seen = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) - wait->target) < 0) {
                        __builtin_amdgcn_s_sleep(1);
                        if (wall_clock64() - t0 > (uint64_t)wait->timeout_ticks) { ok = 0; break; }
                    }
                    if (!ok) report_frame_wait(*wait, seen);
                    *status = ok;
                }
                __syncthreads();
                if (*status == 0) return false;
                // The records are read behind this, from where the samplers put them.  The update workgroup takes the agent-scope acquire
                // by the book (it invalidates its XCD's L2).  The skinning workgroups -- hundreds, all after the same few KB -- must NOT:
                // measured (tools/exp/r05_stamps.py), 196 of them invalidating the L2s under one another turn a 0.7 us fold into 5 us (every
                // record load of every workgroup goes to memory, all to the same few lines).  They do not need to either: the L2s were
                // invalidated when this kernel started, nothing on the chip reads a record line between then and the samplers' last
                // acknowledged write-through store (the records' only readers are behind this wait), so no L2 can hold a stale copy;
                // the first reader of an XCD misses to memory, where the record is, and the rest hit.  Program order is the compiler's to keep.
                if constexpr (QUIET || !INV) __atomic_signal_fence(__ATOMIC_SEQ_CST);
                else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                FSTAMP(2);
            }
        }
        FoldCtx cx;
        cx.tpx = t0.x; cx.tpy = t0.y; cx.tpz = t0.z;
        cx.tr = t1;
        cx.tsx = t2.x; cx.tsy = t2.y; cx.tsz = t2.z;
        cx.dirty = false;
        if constexpr (PROGRAM) {
            cx.ops = prog;
            cx.n_ops = n_ops;
            cx.my_op = my_op;
            cx.anim_pose = reinterpret_cast<const f4*>(f.anim_pose);
            cx.layer_masks = f.layer_masks;
            cx.anim_stride = (size_t)f.n_instances * f.n_nodes;
            cx.rec_index = inst_base + node;
            cx.n_nodes = f.n_nodes;
            cx.node = node;
            if (straight) {
                Acc rec[kStraightOps];
                float rw[kStraightOps];
#pragma unroll
                for (uint32_t i = 0; i < kStraightOps; ++i) {
                    if (i >= st_k) break;
                    const uint2 op = fold_op(cx, st_d + i);
                    rec[i] = load_rec(cx.anim_pose + ((size_t)(op.x >> 8) * cx.anim_stride + cx.rec_index) * 3);
                    rw[i] = __uint_as_float(op.y);
                }
                if (st_player) {      // APPLY_ANIM^k: every pose written to the node in turn (the later one wins per value)
#pragma unroll
                    for (uint32_t i = 0; i < kStraightOps; ++i) {
                        if (i >= st_k) break;
                        apply_pose(cx, rec[i]);
                    }
                } else {
                    bool masked = false;
                    if (st_mask) masked = cx.layer_masks[(size_t)(fold_op(cx, 2u * st_d + st_k).x >> 8) * cx.n_nodes + cx.node] != 0;
                    Acc acc = acc_empty();
#pragma unroll
                    for (uint32_t i = 0; i < kStraightOps; ++i) {
                        if (i >= st_k) break;
                        blend(acc, rec[i], rw[i]);
                    }
                    if (masked) acc.mask = 0;
                    apply_pose(cx, acc);
                }
            } else if constexpr (MODE == kUpdGeneral) {
                cx.pc = 0;
                cx.pop_w = 0.f;
                cx.done = false;
                Acc acc = acc_empty();
                while (!cx.done) run_fold<0>(cx, acc);  // a stray POP at depth 0 is ignored
            } else if constexpr (MODE == kUpdDup) {
                cx.pc = 0;
                cx.pop_w = 0.f;
                cx.done = false;
                AccPair acc;
                acc.a = acc_empty();
                acc.f = acc_empty();
                while (!cx.done) run_fold_dup<0>(cx, acc);
            }
            if constexpr (WAIT) { if (node_base == 0) FSTAMP(3); }
            if (!QUIET && cx.dirty && live) {    // (a quiet workgroup leaves the write-back to the update workgroup: same values)
                trs[0] = f4{cx.tpx, cx.tpy, cx.tpz, 0.f};
                trs[1] = cx.tr;
                trs[2] = f4{cx.tsx, cx.tsy, cx.tsz, 0.f};
            }
        }
        float m[16];
        local_matrix(st, cx.tpx, cx.tpy, cx.tpz, cx.tr, cx.tsx, cx.tsy, cx.tsz, m);
        if (live) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                reinterpret_cast<f4*>(l_local + (size_t)node * 16)[c] = f4{m[c * 4], m[c * 4 + 1], m[c * 4 + 2], m[c * 4 + 3]};
        }
    }
    sync();
    if constexpr (WAIT) FSTAMP(4);

    // level-synchronous global = parent.global * local; a root multiplies by the identity, as
    // the reference does for a node without a valid parent.  What each thread does in the walk was fetched at the top.
    if constexpr (WIDE) {
        // Lane (g, i, j) forms element (row i, column j) of the matrix of group g's node: mat4_mul's chain for that element.  No
        // lane asks whether it has a node (padding entries multiply the identity by zeros into a slot nobody reads) or whether its
        // node has a parent (a root's parent slot holds the identity): a chunk is straight-line code, and the barrier closes a level.
        // Measured (tools/exp/r04_stamps.py, nine levels): 1.3 us against 2.1 us with per-level offsets and a lane test, 2.2 - 3.1 us
        // for the walk in which a lane forms a whole matrix.
        const uint32_t g = threadIdx.x >> 4, e = threadIdx.x & 15u, i = e & 3u, j = e >> 2;
        const uint32_t* cp = s_chunks + g;
        uint32_t ent = cp[0];
        for (uint32_t ch = 0; ch < rig.n_chunks; ++ch) {
            cp += 16;
            const uint32_t nent = cp[0];      // the next chunk's entry: in a register before it is needed
            const uint32_t node = ent & 2047u, slot = (ent >> 11) & 2047u;
            const f4 b = reinterpret_cast<const f4*>(l_local + (size_t)node * 16)[j];
            const float* a = l_global + (size_t)slot * 16 + i;
            const float a0 = a[0], a1 = a[4], a2 = a[8], a3 = a[12];
            float y = a0 * b.x;
            y = a1 * b.y + y;
            y = a2 * b.z + y;
            y = a3 * b.w + y;
            l_global[(size_t)node * 16 + e] = y;          // e = j * 4 + i
            if (__builtin_amdgcn_readfirstlane((int)ent) & (1 << 22)) sync();     // (the flag is the same in all sixteen entries of a chunk)
            ent = nent;
        }
    }
    for (uint32_t lv = 0; !WIDE && lv < rig.n_levels; ++lv) {
#pragma unroll
        for (int k = 0; k < kEntries; ++k) {
            if (w_level[k] != lv) continue;
            const uint32_t node = w_node[k];
            const int32_t par = w_par[k];
            float pg[16], lm[16], g[16];
            if (par >= 0) {
#pragma unroll
                for (int q = 0; q < 16; ++q) pg[q] = l_global[(size_t)par * 16 + q];
            } else {
#pragma unroll
                for (int q = 0; q < 16; ++q) pg[q] = (q % 5 == 0) ? 1.0f : 0.0f;
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) lm[q] = l_local[(size_t)node * 16 + q];
            mat4_mul(pg, lm, g);
#pragma unroll
            for (int c = 0; c < 4; ++c)
                reinterpret_cast<f4*>(l_global + (size_t)node * 16)[c] = f4{g[c * 4], g[c * 4 + 1], g[c * 4 + 2], g[c * 4 + 3]};
        }
        sync();
    }
    if constexpr (WAIT) FSTAMP(5);
    if constexpr (QUIET) {
        // bone_matrices[b] = global(bone_b) * inv_bind(bone_b) (the epilogue's expression, element for element) into the packed-math
        // layout skin_vertex reads (lbs_leaves.h): a thread holds column c = (m0c, m1c, m2c, m3c) of bone b
        f32x4* rows = reinterpret_cast<f32x4*>(reinterpret_cast<char*>(lds) + wide_update_lds(rig.n_nodes, rig.n_chunks) + 16u);
        f32x4* row3 = rows + 3u * skin->n_bones;
        uint32_t* wave_flag = reinterpret_cast<uint32_t*>(row3 + skin->n_bones);
        if constexpr (LATE) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t e = threadIdx.x + (uint32_t)k * 256u;
                if (e < skin->n_bones * 4u) q_node[k] = skin->bone_nodes[e >> 2];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                q_bb[k] = f4{0.f, 0.f, 0.f, 0.f};
                if (q_node[k] >= 0) q_bb[k] = reinterpret_cast<const f4*>(rig.inv_bind)[(size_t)q_node[k] * 4 + (threadIdx.x & 3u)];
            }
        }
        bool pj = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t e = threadIdx.x + (uint32_t)k * 256u;
            if (e >= skin->n_bones * 4u) continue;
            const uint32_t b = e >> 2, c = e & 3u;
            f4 y;
            if (q_node[k] < 0) {
                y = f4{c == 0 ? 1.0f : 0.0f, c == 1 ? 1.0f : 0.0f, c == 2 ? 1.0f : 0.0f, c == 3 ? 1.0f : 0.0f};
            } else {
                const f4 b4 = q_bb[k];
                const f4* a = reinterpret_cast<const f4*>(l_global) + (size_t)q_node[k] * 4;
                const f4 a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3];
                y.x = a0.x * b4.x; y.x = a1.x * b4.y + y.x; y.x = a2.x * b4.z + y.x; y.x = a3.x * b4.w + y.x;
                y.y = a0.y * b4.x; y.y = a1.y * b4.y + y.y; y.y = a2.y * b4.z + y.y; y.y = a3.y * b4.w + y.y;
                y.z = a0.z * b4.x; y.z = a1.z * b4.y + y.z; y.z = a2.z * b4.z + y.z; y.z = a3.z * b4.w + y.z;
                y.w = a0.w * b4.x; y.w = a1.w * b4.y + y.w; y.w = a2.w * b4.z + y.w; y.w = a3.w * b4.w + y.w;
            }
            float* r = reinterpret_cast<float*>(rows + b * 3u);
            *reinterpret_cast<f32x2*>(r + 2u * c) = f32x2{y.x, y.y};
            r[8u + c] = y.z;
            reinterpret_cast<float*>(row3 + b)[c] = y.w;
            pj |= y.w != (c == 3u ? 1.0f : 0.0f);
        }
        const bool wave_pj = __any(pj) != 0;
        if ((threadIdx.x & 63u) == 0) wave_flag[threadIdx.x >> 6] = wave_pj ? 1u : 0u;
        __syncthreads();
        FSTAMP(6);
        return true;
    }
    // The global matrices leave the chip once, after the walk: a store inside the level loop would have every
    // level's barrier wait for its write acknowledgement (s_waitcnt vmcnt(0) ahead of s_barrier: ~0.65 us per level
    // measured, against ~0.1 us for the LDS-only level).
    f4* gout = reinterpret_cast<f4*>(f.global + inst_base * 16);
    f4* lout = reinterpret_cast<f4*>(f.local + inst_base * 16);
    const f4* gin = reinterpret_cast<const f4*>(l_global);
    const f4* lin = reinterpret_cast<const f4*>(l_local);
    for (uint32_t i = tid; i < rig.n_nodes * 4; i += bdim) {
        gout[i] = gin[i];
        lout[i] = lin[i];
    }
    // Registered palettes (Surface::bones of the meshes this rig drives): bone_matrices[b] = global(bone_b) *
    // inv_bind(bone_b) (scene/mesh/mod.rs:781-793) straight from the matrices still in LDS -- no separate gather
    // launch, no re-read of the global matrices.  One thread per output COLUMN (four 16-byte LDS reads, one 16-byte
    // load of inv_bind's column, one 16-byte store), nalgebra's column-axpy order per component.
    // (four columns per thread and pass: their bone -> node loads go out together, then their inverse bind columns -- two round
    // trips for the pass, not two per column; this is the tail of a kernel that is one wave per character)
    for (uint32_t p = 0; p < rig.n_pal; ++p) {
        const PaletteOutDev po = pal_mem ? pal_mem[p] : rig.pal[p];
        f4* out = reinterpret_cast<f4*>(po.out + (size_t)inst * po.n_bones * 16);
        const uint32_t n_cols = po.n_bones * 4;
        for (uint32_t e0 = tid; e0 < n_cols; e0 += 4u * bdim) {
            int32_t node[4];
            f4 bb[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t e = e0 + (uint32_t)k * bdim;
                node[k] = e < n_cols ? po.bone_nodes[e >> 2] : -1;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t e = e0 + (uint32_t)k * bdim;
                bb[k] = f4{0.f, 0.f, 0.f, 0.f};
                if (node[k] >= 0) bb[k] = reinterpret_cast<const f4*>(rig.inv_bind)[(size_t)node[k] * 4 + (e & 3u)];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t e = e0 + (uint32_t)k * bdim;
                if (e >= n_cols) continue;
                const uint32_t j = e & 3u;
                f4 y;
                if (node[k] < 0) {
                    y = f4{j == 0 ? 1.0f : 0.0f, j == 1 ? 1.0f : 0.0f, j == 2 ? 1.0f : 0.0f, j == 3 ? 1.0f : 0.0f};
                } else {
                    const f4 b4 = bb[k];
                    const f4* a = reinterpret_cast<const f4*>(l_global) + (size_t)node[k] * 4;
                    const f4 a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3];
                    y.x = a0.x * b4.x; y.x = a1.x * b4.y + y.x; y.x = a2.x * b4.z + y.x; y.x = a3.x * b4.w + y.x;
                    y.y = a0.y * b4.x; y.y = a1.y * b4.y + y.y; y.y = a2.y * b4.z + y.y; y.y = a3.y * b4.w + y.y;
                    y.z = a0.z * b4.x; y.z = a1.z * b4.y + y.z; y.z = a2.z * b4.z + y.z; y.z = a3.z * b4.w + y.z;
                    y.w = a0.w * b4.x; y.w = a1.w * b4.y + y.w; y.w = a2.w * b4.z + y.w; y.w = a3.w * b4.w + y.w;
                }
                out[e] = y;
            }
        }
    }
    if constexpr (WAIT) {
        FSTAMP(6);
#ifdef FYX_FRAME_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        FSTAMP(7);
#endif
    }
    return true;
}

// Two kernel-argument shapes: with the frame's control block inside the arguments (one character, CtrlInline) and without
// (crowds, scenes, update_transforms: 1 KB less to copy per launch, and no SGPR pressure from a parameter nobody reads).
template <int MODE>
__global__ __launch_bounds__(256) void pose_update_inl_kernel(PoseFrameDev f, RigDev rig, CtrlInline inl) { pose_update_body<MODE, 1, false, true>(ctrl_resolve<kInlAfterFrameAndRig>(f, inl), rig, blockIdx.x, inl.first_ops); }
template <int MODE>
__global__ __launch_bounds__(256) void pose_update_kernel(PoseFrameDev f, RigDev rig) { pose_update_body<MODE>(f, rig, blockIdx.x); }
// One character's frame in one launch (FrameSync, fyx_internal.h).
__device__ __forceinline__ void frame_sample_block(const PoseFrameDev& fr, const FrameSync& fs, uint32_t block) {
    FSTAMP(0);
    const uint32_t bx = block % fs.sx, t = block / fs.sx;
    pose_sample_body<true>(fr, bx, t % fs.sy, t / fs.sy);
    // Every wave: its part of the records is visible device-wide before the workgroup reports.  The records are the only thing
    // of this half that the update half reads, and they were stored with agent-scope (write-through) stores: once those are
    // acknowledged (vmcnt = 0) they are where every XCD finds them, and the L2 write-back an agent-scope release FENCE would
    // add (buffer_wbl2: ~0.7 us of one character's 9 - 11 us; tools/exp/r04_character_ab.py, call 38) has nothing left to do.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x < kFrameCounterReplicas)
        __hip_atomic_fetch_add(fs.counter + (size_t)threadIdx.x * (kFrameCounterStride / 4u), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    FSTAMP(7);
}

template <int MODE>
__global__ __launch_bounds__(256) void pose_frame_inl_kernel(PoseFrameDev f, RigDev rig, CtrlInline inl, FrameSync fs) {
    const PoseFrameDev fr = ctrl_resolve<kInlAfterFrameAndRig>(f, inl);
    if (blockIdx.x < fs.n_sample_blocks) {
        frame_sample_block(fr, fs, blockIdx.x);
        return;
    }
    pose_update_body<MODE, 1, true, true>(fr, rig, blockIdx.x - fs.n_sample_blocks, inl.first_ops, nullptr, &fs);
}

// A skinning workgroup of the one-launch frame (FrameSkin, fyx_internal.h): four waves; wave w of the workgroup takes units
// u0 + w, u0 + w + 4, ... of the workgroup's share [u0, u1) of one instance of one job, two units in flight.  The per-vertex code is
// lbs_skin_dyn's (load_vertex_buf / skin_vertex / store_vertex_buf on buffer resources: a lane past the mesh's end loads zeros and
// stores nothing; a stream the job does not have -- no normals, an output not wanted -- is a resource of zero bytes), the palette
// in LDS is made of the same expressions as the update kernel's palette epilogue: the vertices are lbs_skin's on the palette
// the update workgroup writes to memory, bit for bit.
// PREFETCH: the wave's first two units are requested before the pose is recomputed (one character: the chip is idle and the latency is the
// frame's) -- or behind it (a scene's one-launch frame: other workgroups hide the latency, and 30 registers less held across the update
// body put a third wave on every SIMD).
template <int MODE, bool EXACT, bool WAIT = true, bool PREFETCH = true>
__device__ __forceinline__ void frame_skin_body(const PoseFrameDev& fr, const RigDev& rig, uint32_t first_ops, const FrameSync& fs, const FrameSkin& sk, uint32_t b) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    FrameSkinJob j = sk.job[0];
#pragma unroll
    for (int k = 1; k < kMaxFrameSkins; ++k)
        if ((uint32_t)k < sk.n_jobs && b >= sk.job[k].block0) j = sk.job[k];
    const uint32_t rb = b - j.block0, inst = rb / j.blocks_per_inst, part = rb - inst * j.blocks_per_inst;
    const uint32_t upi = (j.n_verts + 63u) >> 6;
    const uint32_t u0 = (uint32_t)(((uint64_t)part * upi) / j.blocks_per_inst), u1 = (uint32_t)(((uint64_t)(part + 1u) * upi) / j.blocks_per_inst);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    LbsArgs a;
    a.pos = j.pos; a.nrm = j.nrm; a.tan = j.tan; a.wgt = j.wgt; a.idx = j.idx;
    a.palette = nullptr;
    const size_t ov = (size_t)inst * j.n_verts;
    a.out_pos = j.out_pos ? j.out_pos + ov * 3 : nullptr;
    a.out_nrm = j.out_nrm ? j.out_nrm + ov * 3 : nullptr;
    a.out_tan = j.out_tan ? j.out_tan + ov * 4 : nullptr;
    a.n_verts = j.n_verts; a.n_bones = j.n_bones; a.n_instances = 1;
    const VtxBuffers vb = make_vtx_buffers(a);
    constexpr uint32_t kNoVertex = 0x0fffffffu;     // past the end of every stream: loads give zeros, stores are dropped
    // the wave's first two units: nothing of them depends on the pose
    uint32_t uA = u0 + wave, uB = uA + 4u;
    uint32_t vA = uA < u1 ? uA * 64u + lane : kNoVertex, vB = uB < u1 ? uB * 64u + lane : kNoVertex;
    VertexIn<7> A, B;
    if constexpr (PREFETCH) {
        A = load_vertex_buf<7, kFrameSkinLoadAux>(vb, vA);
        B = load_vertex_buf<7, kFrameSkinLoadAux>(vb, vB);
    }

    if (!pose_update_body<MODE, 1, WAIT, true, true, true, !PREFETCH>(fr, rig, inst, first_ops, nullptr, &fs, &j)) return;
    if constexpr (!PREFETCH) {
        A = load_vertex_buf<7, kFrameSkinLoadAux>(vb, vA);
        B = load_vertex_buf<7, kFrameSkinLoadAux>(vb, vB);
    }

    const f32x4* rows = reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(lds) + wide_update_lds(rig.n_nodes, rig.n_chunks) + 16u);
    const f32x4* row3 = rows + 3u * j.n_bones;
    const uint32_t* wave_flag = reinterpret_cast<const uint32_t*>(row3 + j.n_bones);
    const bool projective = (wave_flag[0] | wave_flag[1] | wave_flag[2] | wave_flag[3]) != 0;
    auto process = [&](VertexIn<7>& c_, uint32_t v_c) {
        pin_vertex(c_);
        const Skinned o = skin_vertex<EXACT, 7>(rows, row3, projective, c_.id, c_.w, c_.p.x, c_.p.y, c_.p.z,
                                                c_.n.x, c_.n.y, c_.n.z, c_.t.x, c_.t.y, c_.t.z);
        store_vertex_buf<7, kFrameSkinStoreAux>(vb, v_c, o, c_.t.w);
    };
    while (uA < u1) {     // wave-uniform
        process(A, vA);
        uA += 8u;
        if (uA < u1) { vA = uA * 64u + lane; A = load_vertex_buf<7, kFrameSkinLoadAux>(vb, vA); }
        if (uB < u1) {
            process(B, vB);
            uB += 8u;
            if (uB < u1) { vB = uB * 64u + lane; B = load_vertex_buf<7, kFrameSkinLoadAux>(vb, vB); }
        }
    }
#ifdef FYX_FRAME_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    FSTAMP(7);
#endif
}

// The launch: [sampler workgroups][one update workgroup per instance][skinning workgroups].
template <int MODE, bool EXACT>
__global__ __launch_bounds__(256) void pose_frame_skin_kernel(PoseFrameDev f, RigDev rig, CtrlInline inl, FrameSync fs, FrameSkin sk) {
    const PoseFrameDev fr = ctrl_resolve<kInlAfterFrameAndRig>(f, inl);
    if (blockIdx.x < fs.n_sample_blocks) {
        frame_sample_block(fr, fs, blockIdx.x);
        return;
    }
    const uint32_t u = blockIdx.x - fs.n_sample_blocks;
    if (u < fr.n_instances) {
        pose_update_body<MODE, 1, true, true>(fr, rig, u, inl.first_ops, nullptr, &fs);
        return;
    }
    frame_skin_body<MODE, EXACT>(fr, rig, inl.first_ops, fs, sk, u - fr.n_instances);
}

// Crowds of small rigs (<= 64 nodes: one wave per instance): PACK instances per workgroup (four: one per SIMD).  The waves share
// nothing; what the packing buys is WHERE they land when the launch runs beside a crowd's skinning (anim.overlap): that kernel's
// workgroups take half a CU each (two 128-VGPR waves on every SIMD), a lone update wave of ~160 VGPRs that slips into such a
// place keeps a skinning workgroup out of it for its ~20 us, and 1000 lone waves can hold every such place of the chip; four
// waves that arrive together take one.  Measured on the C3 frame (profiles/r04_frame_study, call17): 97.9 against 99.5 us exact,
// 82.8 against 83.8 fused, the same bits.  A 128-VGPR form (two such workgroups per place; the statics requested after the
// fold) spills 148 bytes and is slower (100.4).  (A workgroup past the crowd's end repeats the last instance: same values,
// same places.)
template <int MODE, int PACK>
__global__ __launch_bounds__(64 * PACK) void pose_update_pack_kernel(PoseFrameDev f, RigDev rig) {
    const uint32_t inst = blockIdx.x * PACK + (threadIdx.x >> 6);
    pose_update_body<MODE, PACK>(f, rig, inst < f.n_instances ? inst : f.n_instances - 1u);
}

// Scene form: every job of one launch has the same block size; the dynamic LDS is sized for the largest rig among them.
template <int MODE, bool WIDE = false>
__global__ __launch_bounds__(256) void pose_update_scene_kernel(const SceneJobDev* __restrict__ jobs, const char* __restrict__ ctrl, const uint4* __restrict__ blocks) {
    const uint4 b = blocks[blockIdx.x];
    const PoseFrameDev f = scene_frame_of(jobs, b.x, ctrl);
    RigDev rig = jobs[b.x].rig;
    pose_update_body<MODE, 1, false, WIDE>(f, rig, b.y, 0u, jobs[b.x].rig.pal);
}

// The scene's update stage that also skins (fyx_animator_set_skin_output on animators of a fyx_scene_update): block {job, instance, 0, -}
// updates (and writes poses, matrices, palettes) as pose_update_scene_kernel does; block {job, k, 1, -} is skinning workgroup k of the job:
// it recomputes the character's pose on chip and skins its share (frame_skin_body without the in-grid wait: the samplers were the
// previous launch).  One launch where there were two, no palette round trip, the update's latency chain under other characters' stores.
template <int MODE, bool EXACT>
__global__ __launch_bounds__(256) void pose_update_skin_scene_kernel(const SceneJobDev* __restrict__ jobs, const char* __restrict__ ctrl, const uint4* __restrict__ blocks) {
    const uint4 b = blocks[blockIdx.x];
    const PoseFrameDev f = scene_frame_of(jobs, b.x, ctrl);
    RigDev rig = jobs[b.x].rig;
    if (b.z == 0u) {
        pose_update_body<MODE, 1, false, true>(f, rig, b.y, 0u, jobs[b.x].rig.pal);
        return;
    }
    const FrameSkin sk = jobs[b.x].sk;
    FrameSync none;
    none.counter = nullptr; none.target = 0; none.n_sample_blocks = 0; none.sx = none.sy = 0; none.timeout_ticks = 0; none.err = nullptr; none.tag = 0;
    frame_skin_body<MODE, EXACT, false>(f, rig, 0u, none, sk, b.y);
}

// A scene of characters as ONE launch (kStageFrame): block {job, k, kind, -} is sampler workgroup k of the job (kind 0), its update
// workgroup for instance k (1) or its skinning workgroup k (2), job after job.  The update and skinning workgroups wait on their JOB's
// frame counter, which only that job's sampler workgroups add to -- workgroups with LOWER indices, so with a dispatcher that hands out
// each XCD's share of a grid in index order the lowest unfinished workgroup of the launch can always run: it is a sampler, or all it
// waits for is done.  What it buys: a character's skinning runs while later characters are still being sampled (the sampler's DRAM
// latency under the skinning's streaming, which one launch per stage serialises), one launch where there were three.  A wait is
// bounded and loud as in the one-character frame (FrameSync::err).
template <int MODE, bool EXACT>
__global__ __launch_bounds__(256) void scene_frame_kernel(const SceneJobDev* __restrict__ jobs, const char* __restrict__ ctrl, const uint4* __restrict__ blocks, SceneWait w) {
    const uint4 b = blocks[blockIdx.x];
    const PoseFrameDev f = scene_frame_of(jobs, b.x, ctrl);
    FrameSync fs;
    fs.counter = jobs[b.x].counter;
    fs.target = reinterpret_cast<const uint32_t*>(ctrl + w.o_targets)[b.x];
    fs.n_sample_blocks = jobs[b.x].n_sample_blocks;
    fs.sx = jobs[b.x].sx;
    fs.sy = f.n_instances;
    fs.timeout_ticks = w.timeout_ticks;
    fs.err = w.err;
    fs.tag = jobs[b.x].tag;
    if (b.z == 0u) {
        frame_sample_block(f, fs, b.y);
        return;
    }
    RigDev rig = jobs[b.x].rig;
    if (b.z == 1u) {
        pose_update_body<MODE, 1, true, true, false, false>(f, rig, b.y, 0u, jobs[b.x].rig.pal, &fs);
        return;
    }
    const FrameSkin sk = jobs[b.x].sk;
    frame_skin_body<MODE, EXACT, true, false>(f, rig, 0u, fs, sk, b.y);
}

template <typename K, typename... Args>
static hipError_t launch_update_one(K kernel, uint32_t grid, uint32_t block, size_t lds, hipStream_t s, Args... args) {
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    FYX_TL_LAUNCH(kernel, dim3(grid), dim3(block), lds, s, args...);
    return hipGetLastError();
}

// matrices + the wide walk's tables (pose_update_body<.., WIDE>)
static size_t wide_walk_lds(const RigDev& rig) { return wide_update_lds(rig.n_nodes, rig.n_chunks); }

hipError_t launch_pose_update(const PoseFrameDev& f, const RigDev& rig, int mode, hipStream_t s, const CtrlInline* inl, int pack) {
    if (f.n_instances == 0 || rig.n_nodes == 0) return hipSuccess;
    const uint32_t block = 64u * update_block_waves(rig.n_nodes, f.n_instances);
    const size_t lds = (size_t)rig.n_nodes * 32 * sizeof(float);
    const bool in_args = inl && inl->bytes && mode != kUpdNoProgram && mode != kUpdDup;
    if (in_args) {    // (a control block that fits the arguments is a few instances: 256 threads, the wide walk)
        if (mode == kUpdStraight) return launch_update_one(pose_update_inl_kernel<kUpdStraight>, f.n_instances, 256u, wide_walk_lds(rig), s, f, rig, *inl);
        return launch_update_one(pose_update_inl_kernel<kUpdGeneral>, f.n_instances, 256u, wide_walk_lds(rig), s, f, rig, *inl);
    }
    if (pack > 1 && rig.n_nodes <= 64u && f.n_instances >= 64u && mode == kUpdStraight) {
        const uint32_t p = pack >= 4 ? 4u : 2u, grid = (f.n_instances + p - 1u) / p;
        if (p == 4u) return launch_update_one(pose_update_pack_kernel<kUpdStraight, 4>, grid, 64u * p, lds * p, s, f, rig);
        return launch_update_one(pose_update_pack_kernel<kUpdStraight, 2>, grid, 64u * p, lds * p, s, f, rig);
    }
    if (mode == kUpdStraight) return launch_update_one(pose_update_kernel<kUpdStraight>, f.n_instances, block, lds, s, f, rig);
    if (mode == kUpdGeneral) return launch_update_one(pose_update_kernel<kUpdGeneral>, f.n_instances, block, lds, s, f, rig);
    if (mode == kUpdDup) return launch_update_one(pose_update_kernel<kUpdDup>, f.n_instances, block, lds, s, f, rig);
    return launch_update_one(pose_update_kernel<kUpdNoProgram>, f.n_instances, block, lds, s, f, rig);
}

hipError_t launch_pose_frame(const PoseFrameDev& f, const RigDev& rig, int mode, hipStream_t s, const CtrlInline& inl, uint32_t* counter, uint32_t* counter_total,
                             const FrameSync& wait, const FrameSkin* skin, bool exact) {
    FrameSync fs = wait;
    fs.counter = counter;
    fs.sx = (f.n_nodes * 16 + 255) / 256;
    fs.sy = f.n_instances;
    fs.n_sample_blocks = fs.sx * fs.sy * f.n_anims;
    fs.target = *counter_total + fs.n_sample_blocks;
    size_t lds = wide_walk_lds(rig) + 16;      // (+ the wait's status word)
    uint32_t grid = fs.n_sample_blocks + f.n_instances;
    hipError_t e;
    if (skin && skin->n_blocks) {
        uint32_t max_bones = 0;
        for (uint32_t k = 0; k < skin->n_jobs; ++k) max_bones = std::max(max_bones, skin->job[k].n_bones);
        lds = wide_walk_lds(rig) + frame_skin_lds(max_bones);
        grid += skin->n_blocks;
        if (mode == kUpdStraight) e = exact ? launch_update_one(pose_frame_skin_kernel<kUpdStraight, true>, grid, 256u, lds, s, f, rig, inl, fs, *skin)
                                            : launch_update_one(pose_frame_skin_kernel<kUpdStraight, false>, grid, 256u, lds, s, f, rig, inl, fs, *skin);
        else e = exact ? launch_update_one(pose_frame_skin_kernel<kUpdGeneral, true>, grid, 256u, lds, s, f, rig, inl, fs, *skin)
                       : launch_update_one(pose_frame_skin_kernel<kUpdGeneral, false>, grid, 256u, lds, s, f, rig, inl, fs, *skin);
    } else {
        // (first_ops of instance 0 only: pose_update_body looks at it for inst == 0)
        e = mode == kUpdStraight ? launch_update_one(pose_frame_inl_kernel<kUpdStraight>, grid, 256u, lds, s, f, rig, inl, fs)
                                 : launch_update_one(pose_frame_inl_kernel<kUpdGeneral>, grid, 256u, lds, s, f, rig, inl, fs);
    }
    if (e == hipSuccess) *counter_total = fs.target;      // a launch that was refused adds nothing to the counter
    return e;
}

// ---------------------------------------------------------------------------------------
// palette[inst][b] = global[inst][bone_b] * inv_bind[bone_b]; one thread per output element.
// An invalid bone handle (negative node) yields the identity (scene/mesh/mod.rs:789-791).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void palette_gather_kernel(const float* __restrict__ global,
                                                             const float* __restrict__ inv_bind,
                                                             const int32_t* __restrict__ bone_nodes,
                                                             uint32_t n_nodes, uint32_t n_bones,
                                                             uint32_t n_instances, float* __restrict__ out) {
    const uint64_t total = (uint64_t)n_instances * n_bones * 16;
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t i = (uint32_t)(e & 3), j = (uint32_t)((e >> 2) & 3);
        const uint64_t mb = e >> 4;
        const uint32_t b = (uint32_t)(mb % n_bones), inst = (uint32_t)(mb / n_bones);
        const int32_t node = bone_nodes[b];
        float y;
        if (node < 0) {
            y = (i == j) ? 1.0f : 0.0f;
        } else {
            const float* a = global + ((size_t)inst * n_nodes + node) * 16;
            const float* bb = inv_bind + (size_t)node * 16;
            y = a[i] * bb[j * 4];
            y = a[4 + i] * bb[j * 4 + 1] + y;
            y = a[8 + i] * bb[j * 4 + 2] + y;
            y = a[12 + i] * bb[j * 4 + 3] + y;
        }
        out[e] = y;
    }
}

hipError_t launch_palette_gather(const float* d_global, const float* d_inv_bind, const int32_t* d_bone_nodes,
                                 uint32_t n_nodes, uint32_t n_bones, uint32_t n_instances, float* d_out,
                                 hipStream_t s) {
    const uint64_t total = (uint64_t)n_instances * n_bones * 16;
    if (total == 0) return hipSuccess;
    uint64_t grid = (total + 255) / 256;
    if (grid > (uint64_t)kCUs * 16) grid = (uint64_t)kCUs * 16;
    hipLaunchKernelGGL(palette_gather_kernel, dim3((uint32_t)grid), dim3(256), 0, s, d_global, d_inv_bind,
                       d_bone_nodes, n_nodes, n_bones, n_instances, d_out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// Scene launches (fyx_scene_update): the grids of launch_pose_sample / _property_sample / _root_motion / _pose_update /
// _property_update, flattened into per-stage block tables so that one launch serves every animator of a scene.
// ---------------------------------------------------------------------------------------
void scene_blocks(uint32_t job, const SceneJobShape& s, std::vector<uint4> (&t)[kSceneStages]) {
    if (s.n_instances == 0 || s.n_nodes == 0) return;
    if (s.n_anims) {
        if (s.sample_form == 2 || (s.sample_form == 0 && s.n_instances >= 32)) {
            const uint32_t gx = (s.n_instances + 63) / 64;
            for (uint32_t a = 0; a < s.n_anims; ++a)
                for (uint32_t y = 0; y < s.n_nodes * 3; ++y)
                    for (uint32_t x = 0; x < gx; ++x) t[kStageSampleCrowd].push_back(make_uint4(job, x, y, a));
        } else {
            const uint32_t gx = (s.n_nodes * 16 + 255) / 256;
            for (uint32_t a = 0; a < s.n_anims; ++a)
                for (uint32_t i = 0; i < s.n_instances; ++i)
                    for (uint32_t x = 0; x < gx; ++x) t[kStageSample].push_back(make_uint4(job, x, i, a));
        }
        if (s.n_prop_slots) {
            const uint32_t gx = (s.n_prop_slots + 255) / 256;
            for (uint32_t a = 0; a < s.n_anims; ++a)
                for (uint32_t i = 0; i < s.n_instances; ++i)
                    for (uint32_t x = 0; x < gx; ++x) t[kStagePropSample].push_back(make_uint4(job, x, i, a));
        }
        if (s.root_motion) {
            const uint64_t items = (uint64_t)s.n_anims * s.n_instances;
            uint64_t grid = (items * 16 + 255) / 256;
            if (grid > (uint64_t)kCUs * 16) grid = (uint64_t)kCUs * 16;
            for (uint32_t x = 0; x < (uint32_t)grid; ++x) t[kStageRootMotion].push_back(make_uint4(job, x, (uint32_t)grid, 0));
        }
    }
    if (s.root_motion_program)
        for (uint32_t x = 0; x < (s.n_instances + 63) / 64; ++x) t[kStageRootMotionFold].push_back(make_uint4(job, x, 0, 0));
    const uint32_t block = update_block_waves(s.n_nodes, s.n_instances);
    for (uint32_t i = 0; i < s.n_instances; ++i) t[kStageUpdate64 + (int)block - 1].push_back(make_uint4(job, i, 0, 0));
    if (block == 4u)
        for (uint32_t k = 0; k < s.skin_blocks; ++k) t[kStageUpdate256].push_back(make_uint4(job, k, 1, 0));
    if (s.n_prop_slots) {
        const uint32_t gx = (s.n_prop_slots + 63) / 64;
        for (uint32_t i = 0; i < s.n_instances; ++i)
            for (uint32_t x = 0; x < gx; ++x) t[kStagePropUpdate].push_back(make_uint4(job, x, i, 0));
    }
}

hipError_t launch_scene(const SceneJobDev* d_jobs, const char* d_ctrl, const uint4* const (&d_tables)[kSceneStages],
                        const uint32_t (&n_blocks)[kSceneStages], const size_t (&lds_bytes)[kSceneStages], bool all_straight, bool wide256, hipStream_t s,
                        bool skin256, bool exact, const SceneWait* frame) {
    if (frame) {      // the whole scene in one launch
        if (!n_blocks[kStageFrame]) return hipSuccess;
        auto one = [&](auto kernel) -> hipError_t {
            const size_t lds = lds_bytes[kStageFrame];
            if (lds > 64 * 1024) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e != hipSuccess) return e;
            }
            FYX_TL_LAUNCH(kernel, dim3(n_blocks[kStageFrame]), dim3(256), lds, s, d_jobs, d_ctrl, d_tables[kStageFrame], *frame);
            return hipGetLastError();
        };
        if (all_straight) return exact ? one(&scene_frame_kernel<kUpdStraight, true>) : one(&scene_frame_kernel<kUpdStraight, false>);
        return exact ? one(&scene_frame_kernel<kUpdGeneral, true>) : one(&scene_frame_kernel<kUpdGeneral, false>);
    }
    auto go = [&](int stage, auto kernel, uint32_t block, size_t lds) {
        if (n_blocks[stage]) hipLaunchKernelGGL(kernel, dim3(n_blocks[stage]), dim3(block), lds, s, d_jobs, d_ctrl, d_tables[stage]);
    };
    go(kStageSample, pose_sample_scene_kernel, 256, 0);
    go(kStageSampleCrowd, pose_sample_crowd_scene_kernel, 64, 0);
    go(kStagePropSample, property_sample_scene_kernel, 256, 0);
    go(kStageRootMotion, root_motion_scene_kernel, 256, 0);
    go(kStageRootMotionFold, root_motion_fold_scene_kernel, 64, 0);
    auto big_lds = [&](const void* kernel, size_t lds) -> hipError_t {
        return lds > 64 * 1024 ? hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) : hipSuccess;
    };
    for (int k = kStageUpdate64; k <= kStageUpdate256; ++k) {
        if (!n_blocks[k]) continue;
        const uint32_t block = 64u * (uint32_t)(k - kStageUpdate64 + 1);
        hipError_t e = hipSuccess;
        if (k == kStageUpdate256 && wide256 && skin256) {     // (LDS: the wide walk's tables + the skinning workgroups' palette, sized by the caller)
            auto fused = [&](auto kernel) {
                e = big_lds(reinterpret_cast<const void*>(kernel), lds_bytes[k]);
                if (e == hipSuccess) go(k, kernel, block, lds_bytes[k]);
            };
            if (all_straight) { if (exact) fused(&pose_update_skin_scene_kernel<kUpdStraight, true>); else fused(&pose_update_skin_scene_kernel<kUpdStraight, false>); }
            else { if (exact) fused(&pose_update_skin_scene_kernel<kUpdGeneral, true>); else fused(&pose_update_skin_scene_kernel<kUpdGeneral, false>); }
        } else if (k == kStageUpdate256 && wide256) {     // (the stage's LDS was sized for the wide walk's tables by the caller)
            if (all_straight) { e = big_lds(reinterpret_cast<const void*>(&pose_update_scene_kernel<kUpdStraight, true>), lds_bytes[k]); if (e == hipSuccess) go(k, pose_update_scene_kernel<kUpdStraight, true>, block, lds_bytes[k]); }
            else { e = big_lds(reinterpret_cast<const void*>(&pose_update_scene_kernel<kUpdGeneral, true>), lds_bytes[k]); if (e == hipSuccess) go(k, pose_update_scene_kernel<kUpdGeneral, true>, block, lds_bytes[k]); }
        } else {
            if (all_straight) { e = big_lds(reinterpret_cast<const void*>(&pose_update_scene_kernel<kUpdStraight>), lds_bytes[k]); if (e == hipSuccess) go(k, pose_update_scene_kernel<kUpdStraight>, block, lds_bytes[k]); }
            else { e = big_lds(reinterpret_cast<const void*>(&pose_update_scene_kernel<kUpdGeneral>), lds_bytes[k]); if (e == hipSuccess) go(k, pose_update_scene_kernel<kUpdGeneral>, block, lds_bytes[k]); }
        }
        if (e != hipSuccess) return e;
    }
    go(kStagePropUpdate, property_update_scene_kernel, 64, 0);
    return hipGetLastError();
}

}  // namespace fyx

#ifdef FYX_SCENE_STAMPS
extern "C" int fyx_exp_scene_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fyx::g_sstamp), sizeof fyx::g_sstamp); }
extern "C" int fyx_exp_scene_stamps_clear() { static unsigned long long z[8192 * 4]; static unsigned int zp[4]; (void)hipMemcpyToSymbol(HIP_SYMBOL(fyx::g_spath), zp, sizeof zp); return (int)hipMemcpyToSymbol(HIP_SYMBOL(fyx::g_sstamp), z, sizeof z); }
extern "C" int fyx_exp_scene_paths(unsigned int* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fyx::g_spath), sizeof fyx::g_spath); }
#endif
#ifdef FYX_FRAME_STAMPS
extern "C" int fyx_exp_frame_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fyx::g_fstamp), sizeof fyx::g_fstamp); }
#endif
