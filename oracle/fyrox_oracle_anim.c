/*
 * fyrox_oracle_anim.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY). See fyrox_oracle.h.
 *
 * Restatement of the fyrox-animation pose path: AnimationTracksData / Animation::tick /
 * AnimationPose fold semantics / Machine (layers, states, transitions, pose nodes,
 * parameters, masks) and the scene-side `apply`.
 * Deliberately written the way the reference is: per-node value lists matched by
 * binding, recursive pose-node evaluation with cached output poses -- NOT the dense
 * batched form the HIP path uses -- so that the two share no structure.
 *
 * Signals/events (lib.rs:471-496), root motion (lib.rs:498-661, pose.rs:73-100,
 * play.rs:97) and the layer event queue (layer.rs:590-706, event.rs:53-90) are restated
 * too.  Not restated (need unavailable crates or the engine's reflection):
 * StateAction::EnableRandomAnimation (rand), Property bindings through reflection
 * (value.rs:404-427; Property values still blend here, they are just not applied to
 * nodes), BlendSpace triangulation (spade crate; triangles are inputs).
 *
 * file:line citations are relative to /root/reference.
 */
#include "fyrox_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ======================================================================== */
/* Poses (fyrox-animation/src/pose.rs, value.rs)                             */
/* ======================================================================== */

typedef struct fo_node_pose {
    int n, cap;
    fo_bound_value* vals;
} fo_node_pose;

struct fo_pose {
    int n_nodes; /* node ids are 0..n_nodes-1; "not in the map" == "empty values" (see header) */
    fo_node_pose* nodes;
    /* pose.rs:54 root_motion: Option<RootMotion>; only the two public fields take part in
     * pose blending (lib.rs:340-343) */
    int rm_has;
    float rm_dp[3]; /* delta_position */
    float rm_dr[4]; /* delta_rotation (i,j,k,w) */
};

static void rm_default(float dp[3], float dr[4]) { /* RootMotion::default(): zero offset, identity */
    dp[0] = dp[1] = dp[2] = 0.0f;
    dr[0] = dr[1] = dr[2] = 0.0f; dr[3] = 1.0f;
}

static void pose_reserve_nodes(fo_pose* p, int n_nodes) {
    if (n_nodes <= p->n_nodes) return;
    p->nodes = (fo_node_pose*)realloc(p->nodes, (size_t)n_nodes * sizeof(fo_node_pose));
    for (int i = p->n_nodes; i < n_nodes; ++i) { p->nodes[i].n = 0; p->nodes[i].cap = 0; p->nodes[i].vals = NULL; }
    p->n_nodes = n_nodes;
}

fo_pose* fo_pose_new(void) { return (fo_pose*)calloc(1, sizeof(fo_pose)); }

void fo_pose_free(fo_pose* p) {
    if (!p) return;
    for (int i = 0; i < p->n_nodes; ++i) free(p->nodes[i].vals);
    free(p->nodes);
    free(p);
}

/* pose.rs:125-129  reset(): clear every node's values, keep the nodes -- and keep root_motion:
 * a pose node's output (blend.rs:145, blendspace.rs:127, layer.rs:596, mod.rs:352) starts every
 * frame with LAST frame's root motion still in place */
void fo_pose_reset(fo_pose* p) {
    for (int i = 0; i < p->n_nodes; ++i) p->nodes[i].n = 0;
}

static void node_push(fo_node_pose* np, const fo_bound_value* v) {
    if (np->n == np->cap) {
        np->cap = np->cap ? np->cap * 2 : 4;
        np->vals = (fo_bound_value*)realloc(np->vals, (size_t)np->cap * sizeof(fo_bound_value));
    }
    np->vals[np->n++] = *v;
}

/* pose.rs:107-121  add_to_node_pose */
void fo_pose_add(fo_pose* p, int node, const fo_bound_value* v) {
    pose_reserve_nodes(p, node + 1);
    node_push(&p->nodes[node], v);
}

static void node_copy(fo_node_pose* dst, const fo_node_pose* src) {
    dst->n = 0;
    for (int i = 0; i < src->n; ++i) node_push(dst, &src->vals[i]);
}

/* pose.rs:58-74  clone_into: dest.reset(); every node of self replaces dest's values */
void fo_pose_clone_into(const fo_pose* src, fo_pose* dst) {
    fo_pose_reset(dst);
    pose_reserve_nodes(dst, src->n_nodes);
    for (int i = 0; i < src->n_nodes; ++i) node_copy(&dst->nodes[i], &src->nodes[i]);
    dst->rm_has = src->rm_has; /* :73 dest.root_motion.clone_from(&self.root_motion) */
    memcpy(dst->rm_dp, src->rm_dp, sizeof dst->rm_dp);
    memcpy(dst->rm_dr, src->rm_dr, sizeof dst->rm_dr);
}

/* pose.rs:78-85 */
void fo_pose_set_root_motion(fo_pose* p, int has, const float dp[3], const float dr[4]) {
    p->rm_has = has;
    if (has) { memcpy(p->rm_dp, dp, 12); memcpy(p->rm_dr, dr, 16); }
}
int fo_pose_root_motion(const fo_pose* p, float dp[3], float dr[4]) {
    if (p->rm_has) { memcpy(dp, p->rm_dp, 12); memcpy(dr, p->rm_dr, 16); } else rm_default(dp, dr);
    return p->rm_has;
}

/* value.rs:221-230  TrackValue::blend_with (mismatched variants: no-op) */
static void value_blend(fo_bound_value* a, const fo_bound_value* b, float w) {
    if (a->kind != b->kind) return;
    switch (a->kind) {
    case FO_VAL_REAL: a->v[0] = fo_lerpf(a->v[0], b->v[0], w); break;
    case FO_VAL_VEC2: fo_vec_lerp(a->v, b->v, w, 2, a->v); break;
    case FO_VAL_VEC3: fo_vec_lerp(a->v, b->v, w, 3, a->v); break;
    case FO_VAL_VEC4: fo_vec_lerp(a->v, b->v, w, 4, a->v); break;
    case FO_VAL_QUAT: { float o[4]; fo_quat_nlerp_shortest(a->v, b->v, w, o); memcpy(a->v, o, sizeof o); break; }
    default: break;
    }
}

/* pose.rs:41-47 NodePose::blend_with + value.rs:438-444 BoundValueCollection::blend_with */
static void node_blend(fo_node_pose* self, const fo_node_pose* other, float w) {
    if (self->n == 0) { /* empty -> plain copy of the other, weight ignored */
        node_copy(self, other);
        return;
    }
    for (int i = 0; i < self->n; ++i) {
        for (int j = 0; j < other->n; ++j) { /* .find(|v| v.binding == value.binding): first match */
            if (other->vals[j].binding == self->vals[i].binding) {
                value_blend(&self->vals[i], &other->vals[j], w);
                break;
            }
        }
    }
}

/* pose.rs:89-101 AnimationPose::blend_with */
void fo_pose_blend_with(fo_pose* self, const fo_pose* other, float w) {
    pose_reserve_nodes(self, other->n_nodes);
    for (int i = 0; i < other->n_nodes; ++i) node_blend(&self->nodes[i], &other->nodes[i], w);
    /* :98-100 self.root_motion.get_or_insert_with(Default::default)
     *             .blend_with(&other.root_motion.clone().unwrap_or_default(), weight)
     * lib.rs:340-343: delta_position.lerp (nalgebra a*(1-t)+b*t), delta_rotation value.rs nlerp */
    if (!self->rm_has) { self->rm_has = 1; rm_default(self->rm_dp, self->rm_dr); }
    float odp[3], odr[4], dr[4];
    if (other->rm_has) { memcpy(odp, other->rm_dp, 12); memcpy(odr, other->rm_dr, 16); } else rm_default(odp, odr);
    fo_vec_lerp(self->rm_dp, odp, w, 3, self->rm_dp);
    fo_quat_nlerp_shortest(self->rm_dr, odr, w, dr);
    memcpy(self->rm_dr, dr, sizeof dr);
}

/* layer.rs:700-702  final_pose.poses_mut().retain(|h,_| mask.should_animate(*h)) */
static void pose_drop_nodes(fo_pose* p, const int* excluded, int n_excluded) {
    for (int i = 0; i < n_excluded; ++i)
        if (excluded[i] >= 0 && excluded[i] < p->n_nodes) p->nodes[excluded[i]].n = 0;
}

int fo_pose_node_capacity(const fo_pose* p) { return p->n_nodes; }
int fo_pose_value_count(const fo_pose* p, int node) {
    return (node >= 0 && node < p->n_nodes) ? p->nodes[node].n : 0;
}
int fo_pose_get_value(const fo_pose* p, int node, int i, fo_bound_value* out) {
    if (node < 0 || node >= p->n_nodes || i < 0 || i >= p->nodes[node].n) return -1;
    *out = p->nodes[node].vals[i];
    return 0;
}

/* scene/animation/mod.rs:147-186 BoundValueCollectionExt::apply +
 * scene/transform.rs:202-260 set_position/set_rotation/set_scale.  The setters store
 * the value when it differs (or the transform is already dirty); storing an equal
 * value is unobservable, so this simply assigns.  Values whose variant does not fit
 * the binding are skipped (the reference logs an error). */
void fo_pose_apply(const fo_pose* p, fo_transform* nodes, int n_nodes) {
    for (int n = 0; n < p->n_nodes && n < n_nodes; ++n) {
        const fo_node_pose* np = &p->nodes[n];
        for (int i = 0; i < np->n; ++i) {
            const fo_bound_value* bv = &np->vals[i];
            switch (bv->binding) {
            case FO_BIND_POSITION:
                if (bv->kind == FO_VAL_VEC3) memcpy(nodes[n].local_position, bv->v, 12);
                break;
            case FO_BIND_SCALE:
                if (bv->kind == FO_VAL_VEC3) memcpy(nodes[n].local_scale, bv->v, 12);
                break;
            case FO_BIND_ROTATION:
                if (bv->kind == FO_VAL_QUAT) memcpy(nodes[n].local_rotation, bv->v, 16);
                break;
            default: break; /* Property: reflection, not restated */
            }
        }
    }
}

/* ======================================================================== */
/* AnimationTracksData / Track  (lib.rs:66-110, track.rs:100-205)            */
/* ======================================================================== */

typedef struct fo_track {
    int binding; /* FO_BIND_* or >= FO_BIND_PROPERTY0 */
    int kind;    /* FO_KIND_* */
    uint32_t n_curves;
    fo_curve curves[4];
} fo_track;

struct fo_tracks {
    int n, cap;
    fo_track* t;
};

fo_tracks* fo_tracks_new(void) { return (fo_tracks*)calloc(1, sizeof(fo_tracks)); }

static void* dup_mem(const void* src, size_t bytes) {
    void* p = malloc(bytes ? bytes : 1);
    if (src && bytes) memcpy(p, src, bytes);
    return p;
}

int fo_tracks_add_track(fo_tracks* td, int binding, int kind, uint32_t n_curves, const fo_curve* curves) {
    if (n_curves > 4) return -1;
    if (td->n == td->cap) {
        td->cap = td->cap ? td->cap * 2 : 16;
        td->t = (fo_track*)realloc(td->t, (size_t)td->cap * sizeof(fo_track));
    }
    fo_track* t = &td->t[td->n];
    memset(t, 0, sizeof *t);
    t->binding = binding;
    t->kind = kind;
    t->n_curves = n_curves;
    for (uint32_t c = 0; c < n_curves; ++c) {
        uint32_t k = curves[c].n_keys;
        t->curves[c].n_keys = k;
        t->curves[c].location = (const float*)dup_mem(curves[c].location, k * sizeof(float));
        t->curves[c].value = (const float*)dup_mem(curves[c].value, k * sizeof(float));
        t->curves[c].kind = (const uint8_t*)dup_mem(curves[c].kind, k);
        t->curves[c].left_tangent = (const float*)dup_mem(curves[c].left_tangent, k * sizeof(float));
        t->curves[c].right_tangent = (const float*)dup_mem(curves[c].right_tangent, k * sizeof(float));
    }
    return td->n++;
}

void fo_tracks_free(fo_tracks* td) {
    if (!td) return;
    for (int i = 0; i < td->n; ++i)
        for (uint32_t c = 0; c < td->t[i].n_curves; ++c) {
            free((void*)td->t[i].curves[c].location);
            free((void*)td->t[i].curves[c].value);
            free((void*)td->t[i].curves[c].kind);
            free((void*)td->t[i].curves[c].left_tangent);
            free((void*)td->t[i].curves[c].right_tangent);
        }
    free(td->t);
    free(td);
}

int fo_tracks_count(const fo_tracks* td) { return td->n; }

/* track.rs:184-189 Track::fetch -> BoundValue{binding, value} */
static int track_fetch(const fo_track* t, float time, size_t hints[4], fo_bound_value* out) {
    float v[4] = { 0, 0, 0, 0 };
    int n = fo_track_fetch(t->curves, t->n_curves, t->kind, time, hints, v);
    if (n == 0) return 0; /* None */
    out->binding = t->binding;
    switch (t->kind) {
    case FO_KIND_REAL: out->kind = FO_VAL_REAL; break;
    case FO_KIND_VEC2: out->kind = FO_VAL_VEC2; break;
    case FO_KIND_VEC3: out->kind = FO_VAL_VEC3; break;
    case FO_KIND_VEC4: out->kind = FO_VAL_VEC4; break;
    default: out->kind = FO_VAL_QUAT; break;
    }
    memcpy(out->v, v, sizeof v);
    return 1;
}

/* ======================================================================== */
/* Animation (lib.rs)                                                        */
/* ======================================================================== */

typedef struct fo_track_binding { /* track.rs:40-97 TrackBinding: enabled, target, fetch_hints */
    int bound;  /* has an entry in track_bindings */
    int enabled;
    int target;
    size_t hints[4];
} fo_track_binding;

typedef struct fo_signal { float time; int enabled; } fo_signal; /* signal.rs: AnimationSignal */

typedef struct fo_root_motion { /* lib.rs:325-336 RootMotion */
    float delta_position[3];
    float delta_rotation[4];
    float prev_position[3];
    int has_position_offset_remainder;
    float position_offset_remainder[3];
    float prev_rotation[4];
    int has_rotation_remainder;
    float rotation_remainder[4];
} fo_root_motion;

static void root_motion_default(fo_root_motion* r) {
    memset(r, 0, sizeof *r);
    r->delta_rotation[3] = 1.0f; /* UnitQuaternion::default() == identity */
    r->prev_rotation[3] = 1.0f;
}

struct fo_animation {
    const fo_tracks* tracks;
    fo_track_binding* bindings; /* one per track (track_bindings map keyed by track id) */
    float speed, time_position, slice_start, slice_end;
    int enabled, looped;
    fo_pose* pose;
    /* signals: Vec<AnimationSignal>, events: VecDeque<AnimationEvent> (signal index stands for
     * the {signal_id, name} pair), max_event_capacity (default 32, lib.rs:941) */
    int n_signals;
    fo_signal* signals;
    int n_events, ev_cap, ev_head; /* ring buffer of signal indices */
    int* events;
    size_t max_event_capacity;
    /* root_motion_settings: Option<RootMotionSettings>, root_motion: Option<RootMotion> */
    int has_rm_settings, rm_node, rm_ignore_x, rm_ignore_y, rm_ignore_z, rm_ignore_rotations;
    int has_root_motion;
    fo_root_motion root_motion;
};

/* lib.rs:928-950 Default: speed 1, time 0, enabled, looped, time_slice 0..0 */
fo_animation* fo_animation_new(const fo_tracks* td) {
    fo_animation* a = (fo_animation*)calloc(1, sizeof *a);
    a->tracks = td;
    a->bindings = (fo_track_binding*)calloc((size_t)(td->n ? td->n : 1), sizeof(fo_track_binding));
    a->speed = 1.0f;
    a->enabled = 1;
    a->looped = 1;
    a->pose = fo_pose_new();
    a->max_event_capacity = 32;
    return a;
}

void fo_animation_free(fo_animation* a) {
    if (!a) return;
    fo_pose_free(a->pose);
    free(a->bindings);
    free(a->signals);
    free(a->events);
    free(a);
}

/* lib.rs add_signal / signals_mut */
int fo_animation_add_signal(fo_animation* a, float time, int enabled) {
    a->signals = (fo_signal*)realloc(a->signals, (size_t)(a->n_signals + 1) * sizeof(fo_signal));
    a->signals[a->n_signals].time = time;
    a->signals[a->n_signals].enabled = enabled;
    return a->n_signals++;
}
void fo_animation_set_signal_enabled(fo_animation* a, int signal, int enabled) {
    if (signal >= 0 && signal < a->n_signals) a->signals[signal].enabled = enabled;
}
void fo_animation_set_max_event_capacity(fo_animation* a, uint32_t cap) { a->max_event_capacity = cap; } /* :380 */
int fo_animation_event_count(const fo_animation* a) { return a->n_events; }
static void events_push_back(fo_animation* a, int signal) {
    if (a->n_events == a->ev_cap) {
        int ncap = a->ev_cap ? a->ev_cap * 2 : 16;
        int* ne = (int*)malloc((size_t)ncap * sizeof(int));
        for (int i = 0; i < a->n_events; ++i) ne[i] = a->events[(a->ev_head + i) % a->ev_cap];
        free(a->events);
        a->events = ne; a->ev_cap = ncap; a->ev_head = 0;
    }
    a->events[(a->ev_head + a->n_events) % a->ev_cap] = signal;
    ++a->n_events;
}
/* lib.rs:680-682 pop_event: events.pop_front(); -1 == None */
int fo_animation_pop_event(fo_animation* a) {
    if (!a->n_events) return -1;
    int s = a->events[a->ev_head];
    a->ev_head = (a->ev_head + 1) % a->ev_cap;
    --a->n_events;
    return s;
}
void fo_animation_clear_events(fo_animation* a) { a->n_events = 0; a->ev_head = 0; } /* take_events / events_mut().clear() */

/* lib.rs:664-666 set_root_motion_settings; node < 0 == None */
void fo_animation_set_root_motion_settings(fo_animation* a, int node, int ignore_x, int ignore_y,
                                           int ignore_z, int ignore_rotations) {
    a->has_rm_settings = node >= 0;
    a->rm_node = node;
    a->rm_ignore_x = ignore_x; a->rm_ignore_y = ignore_y; a->rm_ignore_z = ignore_z;
    a->rm_ignore_rotations = ignore_rotations;
}
/* lib.rs:674-676 root_motion(); returns 0 for None */
int fo_animation_root_motion(const fo_animation* a, float delta_position[3], float delta_rotation[4]) {
    if (!a->has_root_motion) { rm_default(delta_position, delta_rotation); return 0; }
    memcpy(delta_position, a->root_motion.delta_position, 12);
    memcpy(delta_rotation, a->root_motion.delta_rotation, 16);
    return 1;
}

/* add_track_with_binding / track_bindings_mut: target < 0 removes the binding */
void fo_animation_bind(fo_animation* a, int track, int target_node, int enabled) {
    if (track < 0 || track >= a->tracks->n) return;
    fo_track_binding* b = &a->bindings[track];
    memset(b, 0, sizeof *b);
    if (target_node < 0) return;
    b->bound = 1;
    b->enabled = enabled;
    b->target = target_node;
}

/* lib.rs:432-440 */
void fo_animation_set_time_position(fo_animation* a, float time) {
    if (a->looped) {
        a->time_position = fo_wrapf(time, a->slice_start, a->slice_end);
    } else {
        /* f32::clamp: max(min) then min(max); NaN stays NaN */
        float t = time;
        if (t < a->slice_start) t = a->slice_start;
        if (t > a->slice_end) t = a->slice_end;
        a->time_position = t;
    }
}
/* lib.rs:445-452 */
void fo_animation_set_time_slice(fo_animation* a, float start, float end) {
    a->slice_start = start;
    a->slice_end = end;
    fo_animation_set_time_position(a, a->time_position);
}
void fo_animation_set_speed(fo_animation* a, float s) { a->speed = s; }
void fo_animation_set_loop(fo_animation* a, int l) { a->looped = l; }
void fo_animation_set_enabled(fo_animation* a, int e) { a->enabled = e; }
void fo_animation_rewind(fo_animation* a) { fo_animation_set_time_position(a, a->slice_start); } /* :460 */
float fo_animation_time_position(const fo_animation* a) { return a->time_position; }
int fo_animation_is_enabled(const fo_animation* a) { return a->enabled; }
/* lib.rs:736-738 */
int fo_animation_has_ended(const fo_animation* a) {
    return !a->looped && fabsf(a->time_position - a->slice_end) <= FLT_EPSILON;
}
const fo_pose* fo_animation_pose(const fo_animation* a) { return a->pose; }

/* lib.rs:895-914 update_pose */
static void animation_update_pose(fo_animation* a) {
    fo_pose_reset(a->pose);
    for (int i = 0; i < a->tracks->n; ++i) {
        fo_track_binding* b = &a->bindings[i];
        if (!b->bound) continue;
        if (b->enabled) {
            fo_bound_value bv;
            if (track_fetch(&a->tracks->t[i], a->time_position, b->hints, &bv))
                fo_pose_add(a->pose, b->target, &bv);
        }
    }
}

/* lib.rs:507-519 fetch_position_at_time: the FIRST track of the tracks data whose binding is
 * Position (whatever node it is bound to, enabled or not), fresh hints, Vector3 or default */
static void rm_fetch_position(const fo_tracks* td, float time, float out[3]) {
    out[0] = out[1] = out[2] = 0.0f;
    for (int i = 0; i < td->n; ++i) {
        if (td->t[i].binding != FO_BIND_POSITION) continue;
        size_t hints[4] = { 0, 0, 0, 0 };
        fo_bound_value bv;
        if (track_fetch(&td->t[i], time, hints, &bv) && bv.kind == FO_VAL_VEC3) memcpy(out, bv.v, 12);
        return; /* .find() stops at the first Position track even when its fetch fails */
    }
}
/* lib.rs:521-534 fetch_rotation_at_time */
static void rm_fetch_rotation(const fo_tracks* td, float time, float out[4]) {
    out[0] = out[1] = out[2] = 0.0f; out[3] = 1.0f;
    for (int i = 0; i < td->n; ++i) {
        if (td->t[i].binding != FO_BIND_ROTATION) continue;
        size_t hints[4] = { 0, 0, 0, 0 };
        fo_bound_value bv;
        if (track_fetch(&td->t[i], time, hints, &bv) && bv.kind == FO_VAL_QUAT) memcpy(out, bv.v, 16);
        return;
    }
}
/* nalgebra UnitQuaternion::inverse() == conjugate: (-i,-j,-k,w) */
static void quat_conj(const float q[4], float out[4]) { out[0] = -q[0]; out[1] = -q[1]; out[2] = -q[2]; out[3] = q[3]; }

/* lib.rs:498-661 update_root_motion */
static void animation_update_root_motion(fo_animation* a, float prev_time_position) {
    if (!a->has_rm_settings) return;
    const fo_tracks* td = a->tracks;
    fo_root_motion prev;
    if (a->has_root_motion) prev = a->root_motion; else root_motion_default(&prev);
    int new_loop_cycle_started = a->looped
        && ((a->speed > 0.0f && a->time_position < prev_time_position)
            || (a->speed < 0.0f && a->time_position > prev_time_position));
    float cycle_start_time = a->speed > 0.0f ? a->slice_start : a->slice_end;
    float cycle_end_time = a->speed > 0.0f ? a->slice_end : a->slice_start;
    fo_root_motion rm;
    root_motion_default(&rm);
    if (a->rm_node >= 0 && a->rm_node < a->pose->n_nodes) {
        fo_node_pose* np = &a->pose->nodes[a->rm_node];
        for (int i = 0; i < np->n; ++i) {
            fo_bound_value* bv = &np->vals[i];
            if (bv->binding == FO_BIND_POSITION) {
                if (bv->kind != FO_VAL_VEC3) continue;
                float pose_position[3] = { bv->v[0], bv->v[1], bv->v[2] };
                if (new_loop_cycle_started) {
                    float end[3];
                    rm_fetch_position(td, cycle_start_time, rm.prev_position);
                    rm_fetch_position(td, cycle_end_time, end);
                    rm.has_position_offset_remainder = 1;
                    for (int k = 0; k < 3; ++k) rm.position_offset_remainder[k] = end[k] - pose_position[k];
                } else {
                    memcpy(rm.prev_position, pose_position, 12);
                }
                float remainder[3] = { 0, 0, 0 };
                if (prev.has_position_offset_remainder) { /* .take().unwrap_or_default() */
                    memcpy(remainder, prev.position_offset_remainder, 12);
                    prev.has_position_offset_remainder = 0;
                }
                float delta[3];
                for (int k = 0; k < 3; ++k) {
                    float current_offset = pose_position[k] - prev.prev_position[k];
                    delta[k] = current_offset + remainder[k];
                }
                rm.delta_position[0] = a->rm_ignore_x ? 0.0f : delta[0];
                rm.delta_position[1] = a->rm_ignore_y ? 0.0f : delta[1];
                rm.delta_position[2] = a->rm_ignore_z ? 0.0f : delta[2];
                float start_position[3];
                rm_fetch_position(td, a->slice_start, start_position);
                bv->v[0] = a->rm_ignore_x ? pose_position[0] : start_position[0];
                bv->v[1] = a->rm_ignore_y ? pose_position[1] : start_position[1];
                bv->v[2] = a->rm_ignore_z ? pose_position[2] : start_position[2];
            } else if (bv->binding == FO_BIND_ROTATION) {
                if (bv->kind != FO_VAL_QUAT) continue;
                if (a->rm_ignore_rotations) continue;
                float pose_rotation[4] = { bv->v[0], bv->v[1], bv->v[2], bv->v[3] };
                if (new_loop_cycle_started) {
                    float end[4], inv[4];
                    rm_fetch_rotation(td, cycle_start_time, rm.prev_rotation);
                    rm_fetch_rotation(td, cycle_end_time, end);
                    quat_conj(end, inv);
                    rm.has_rotation_remainder = 1;
                    fo_quat_mul(inv, pose_rotation, rm.rotation_remainder);
                } else {
                    memcpy(rm.prev_rotation, pose_rotation, 16);
                }
                float remainder[4] = { 0, 0, 0, 1 };
                if (prev.has_rotation_remainder) {
                    memcpy(remainder, prev.rotation_remainder, 16);
                    prev.has_rotation_remainder = 0;
                }
                float inv_prev[4], current_relative_rotation[4];
                quat_conj(prev.prev_rotation, inv_prev);
                fo_quat_mul(inv_prev, pose_rotation, current_relative_rotation);
                fo_quat_mul(remainder, current_relative_rotation, rm.delta_rotation);
                rm_fetch_rotation(td, a->slice_start, bv->v);
            }
        }
    }
    a->root_motion = rm;
    a->has_root_motion = 1;
}

/* lib.rs:471-496 tick: pose at the OLD time, signals over (t, t + dt*speed], advance, root motion */
void fo_animation_tick(fo_animation* a, float dt) {
    animation_update_pose(a);
    float current_time_position = a->time_position;
    float new_time_position = current_time_position + dt * a->speed;
    for (int i = 0; i < a->n_signals; ++i) {
        const fo_signal* sg = &a->signals[i];
        if (!sg->enabled) continue;
        /* operator precedence as written at :478-482: a || (b && c && cap) -- the capacity cap
         * binds to the negative-speed branch only */
        if ((a->speed >= 0.0f && (current_time_position < sg->time && new_time_position >= sg->time))
            || (a->speed < 0.0f && (current_time_position > sg->time && new_time_position <= sg->time)
                && (size_t)a->n_events < a->max_event_capacity))
            events_push_back(a, i);
    }
    float prev_time_position = current_time_position;
    fo_animation_set_time_position(a, new_time_position);
    animation_update_root_motion(a, prev_time_position);
}

/* ======================================================================== */
/* Machine (machine/mod.rs, layer.rs, state.rs, transition.rs, node/ *.rs)    */
/* ======================================================================== */

typedef struct fo_param {
    int kind; /* FO_PARAM_* */
    float f[2];
    uint32_t u;
} fo_param;

typedef struct fo_blend_input { int source; int weight_param; float weight_const; float blend_time; } fo_blend_input;

typedef struct fo_pose_node {
    int type; /* FO_NODE_* */
    int animation;          /* Play */
    int n_inputs;           /* Blend / ByIndex: inputs; BlendSpace: points */
    fo_blend_input* inputs;
    int param;              /* ByIndex: index parameter; BlendSpace: sampling parameter */
    int has_prev_index;     /* ByIndex state: prev_index: Cell<Option<u32>> */
    uint32_t prev_index;
    float blend_time;       /* ByIndex state */
    float* points;          /* BlendSpace: xy per point */
    int n_tris;
    uint32_t* tris;
    fo_pose* output;
} fo_pose_node;

typedef struct fo_action { int kind; int animation; int n_choices; int* choices; } fo_action;
typedef struct fo_state {
    int root;
    int n_enter, n_leave;
    fo_action *enter, *leave;
} fo_state;

typedef struct fo_transition {
    int source, dest;
    float transition_time, elapsed_time, blend_factor;
    int n_logic;
    int* logic; /* prefix-encoded LogicNode tree */
} fo_transition;

typedef struct fo_layer {
    float weight;
    int n_nodes, n_states, n_transitions;
    fo_pose_node* nodes;
    fo_state* states;
    fo_transition* transitions;
    int active_state, active_transition; /* -1 == Handle::NONE */
    int entry_state;                     /* layer.rs:105: set by set_entry_state only */
    int n_excluded;
    int* excluded;
    fo_pose* final_pose;
    /* layer.rs:117,182 events: FixedEventQueue::new(2048) (event.rs:53-90): push drops the
     * event when the queue is full, pop takes from the front */
    int n_events, ev_head;
    int (*events)[3]; /* {FO_EVENT_*, a, b}, ring of FO_LAYER_EVENT_LIMIT */
} fo_layer;

#define FO_LAYER_EVENT_LIMIT 2048
static void layer_push_event(fo_layer* L, int kind, int a, int b) {
    if (L->n_events >= FO_LAYER_EVENT_LIMIT) return;
    if (!L->events) L->events = (int(*)[3])malloc(sizeof(int[3]) * FO_LAYER_EVENT_LIMIT);
    int* e = L->events[(L->ev_head + L->n_events) % FO_LAYER_EVENT_LIMIT];
    e[0] = kind; e[1] = a; e[2] = b;
    ++L->n_events;
}
/* layer.rs:284-286 pop_event; returns 0 for None */
int fo_layer_pop_event(fo_machine* m, int layer, int out[3]);

struct fo_machine {
    uint64_t rng; /* StateAction::EnableRandomAnimation: see apply_actions */
    int n_params, n_layers;
    fo_param* params;
    fo_layer* layers;
    fo_pose* final_pose;
};

fo_machine* fo_machine_new(void) {
    fo_machine* m = (fo_machine*)calloc(1, sizeof *m);
    m->final_pose = fo_pose_new();
    return m;
}

void fo_machine_free(fo_machine* m) {
    if (!m) return;
    for (int l = 0; l < m->n_layers; ++l) {
        fo_layer* L = &m->layers[l];
        for (int i = 0; i < L->n_nodes; ++i) {
            free(L->nodes[i].inputs); free(L->nodes[i].points); free(L->nodes[i].tris);
            fo_pose_free(L->nodes[i].output);
        }
        for (int i = 0; i < L->n_states; ++i) {
            for (int k = 0; k < L->states[i].n_enter; ++k) free(L->states[i].enter[k].choices);
            for (int k = 0; k < L->states[i].n_leave; ++k) free(L->states[i].leave[k].choices);
            free(L->states[i].enter); free(L->states[i].leave);
        }
        for (int i = 0; i < L->n_transitions; ++i) free(L->transitions[i].logic);
        free(L->nodes); free(L->states); free(L->transitions); free(L->excluded); free(L->events);
        fo_pose_free(L->final_pose);
    }
    free(m->layers); free(m->params);
    fo_pose_free(m->final_pose);
    free(m);
}

int fo_machine_add_parameter(fo_machine* m, int kind, float f0, float f1, uint32_t u) {
    m->params = (fo_param*)realloc(m->params, (size_t)(m->n_params + 1) * sizeof(fo_param));
    fo_param* p = &m->params[m->n_params];
    p->kind = kind; p->f[0] = f0; p->f[1] = f1; p->u = u;
    return m->n_params++;
}
void fo_machine_set_parameter(fo_machine* m, int index, int kind, float f0, float f1, uint32_t u) {
    if (index < 0 || index >= m->n_params) return;
    fo_param* p = &m->params[index];
    p->kind = kind; p->f[0] = f0; p->f[1] = f1; p->u = u;
}

/* layer.rs Default: weight 1.0, no states */
int fo_machine_add_layer(fo_machine* m, float weight) {
    m->layers = (fo_layer*)realloc(m->layers, (size_t)(m->n_layers + 1) * sizeof(fo_layer));
    fo_layer* L = &m->layers[m->n_layers];
    memset(L, 0, sizeof *L);
    L->weight = weight;
    L->active_state = -1;
    L->active_transition = -1;
    L->entry_state = -1;
    L->final_pose = fo_pose_new();
    return m->n_layers++;
}
void fo_layer_set_weight(fo_machine* m, int layer, float w) { m->layers[layer].weight = w; }
void fo_layer_set_mask(fo_machine* m, int layer, const int* excluded, int n) {
    fo_layer* L = &m->layers[layer];
    free(L->excluded);
    L->excluded = (int*)dup_mem(excluded, (size_t)n * sizeof(int));
    L->n_excluded = n;
}

static fo_pose_node* layer_new_node(fo_layer* L, int type) {
    L->nodes = (fo_pose_node*)realloc(L->nodes, (size_t)(L->n_nodes + 1) * sizeof(fo_pose_node));
    fo_pose_node* n = &L->nodes[L->n_nodes++];
    memset(n, 0, sizeof *n);
    n->type = type;
    n->animation = -1;
    n->param = -1;
    n->output = fo_pose_new();
    return n;
}

int fo_layer_add_play(fo_machine* m, int layer, int animation) {
    fo_layer* L = &m->layers[layer];
    fo_pose_node* n = layer_new_node(L, FO_NODE_PLAY);
    n->animation = animation;
    return L->n_nodes - 1;
}

int fo_layer_add_blend(fo_machine* m, int layer, int n_inputs, const int* sources,
                       const int* weight_params, const float* weight_consts) {
    fo_layer* L = &m->layers[layer];
    fo_pose_node* n = layer_new_node(L, FO_NODE_BLEND);
    n->n_inputs = n_inputs;
    n->inputs = (fo_blend_input*)calloc((size_t)(n_inputs ? n_inputs : 1), sizeof(fo_blend_input));
    for (int i = 0; i < n_inputs; ++i) {
        n->inputs[i].source = sources[i];
        n->inputs[i].weight_param = weight_params ? weight_params[i] : -1;
        n->inputs[i].weight_const = weight_consts ? weight_consts[i] : 0.0f;
    }
    return L->n_nodes - 1;
}

int fo_layer_add_blend_by_index(fo_machine* m, int layer, int index_param, int n_inputs,
                                const int* sources, const float* blend_times) {
    fo_layer* L = &m->layers[layer];
    fo_pose_node* n = layer_new_node(L, FO_NODE_BLEND_BY_INDEX);
    n->param = index_param;
    n->n_inputs = n_inputs;
    n->inputs = (fo_blend_input*)calloc((size_t)(n_inputs ? n_inputs : 1), sizeof(fo_blend_input));
    for (int i = 0; i < n_inputs; ++i) {
        n->inputs[i].source = sources[i];
        n->inputs[i].blend_time = blend_times[i];
    }
    return L->n_nodes - 1;
}

int fo_layer_add_blend_space(fo_machine* m, int layer, int sampling_param, int n_points,
                             const float* points_xy, const int* sources, int n_tris,
                             const uint32_t* tris) {
    fo_layer* L = &m->layers[layer];
    fo_pose_node* n = layer_new_node(L, FO_NODE_BLEND_SPACE);
    n->param = sampling_param;
    n->n_inputs = n_points;
    n->inputs = (fo_blend_input*)calloc((size_t)(n_points ? n_points : 1), sizeof(fo_blend_input));
    for (int i = 0; i < n_points; ++i) n->inputs[i].source = sources[i];
    n->points = (float*)dup_mem(points_xy, (size_t)n_points * 2 * sizeof(float));
    n->n_tris = n_tris;
    n->tris = (uint32_t*)dup_mem(tris, (size_t)n_tris * 3 * sizeof(uint32_t));
    return L->n_nodes - 1;
}

/* layer.rs:229-235 add_state: the first state becomes the active one */
int fo_layer_add_state(fo_machine* m, int layer, int root_node) {
    fo_layer* L = &m->layers[layer];
    L->states = (fo_state*)realloc(L->states, (size_t)(L->n_states + 1) * sizeof(fo_state));
    fo_state* s = &L->states[L->n_states];
    memset(s, 0, sizeof *s);
    s->root = root_node;
    if (L->active_state < 0) L->active_state = L->n_states;
    return L->n_states++;
}
/* layer.rs:209-212 */
void fo_layer_set_entry_state(fo_machine* m, int layer, int state) {
    m->layers[layer].active_state = state;
    m->layers[layer].entry_state = state;
}
/* layer.rs:288-296 reset: every transition's elapsed_time and blend_factor return to 0 (transition.rs:311-314) and
 * active_state = entry_state -- which is NONE unless set_entry_state was called; active_transition is NOT cleared */
void fo_layer_reset(fo_machine* m, int layer) {
    fo_layer* L = &m->layers[layer];
    for (int t = 0; t < L->n_transitions; ++t) L->transitions[t].elapsed_time = L->transitions[t].blend_factor = 0.0f;
    L->active_state = L->entry_state;
}

void fo_state_add_action(fo_machine* m, int layer, int state, int on_enter, int kind, int animation) {
    fo_state* s = &m->layers[layer].states[state];
    fo_action** arr = on_enter ? &s->enter : &s->leave;
    int* cnt = on_enter ? &s->n_enter : &s->n_leave;
    *arr = (fo_action*)realloc(*arr, (size_t)(*cnt + 1) * sizeof(fo_action));
    (*arr)[*cnt].kind = kind;
    (*arr)[*cnt].animation = animation;
    (*arr)[*cnt].n_choices = 0;
    (*arr)[*cnt].choices = NULL;
    ++*cnt;
}

/* state.rs:85 StateAction::EnableRandomAnimation(Vec<Handle>) */
void fo_state_add_random_action(fo_machine* m, int layer, int state, int on_enter, const int* animations, int n) {
    fo_state_add_action(m, layer, state, on_enter, FO_ACTION_ENABLE_RANDOM, -1);
    fo_state* s = &m->layers[layer].states[state];
    fo_action* a = on_enter ? &s->enter[s->n_enter - 1] : &s->leave[s->n_leave - 1];
    a->n_choices = n;
    a->choices = (int*)malloc((size_t)(n ? n : 1) * sizeof(int));
    if (n) memcpy(a->choices, animations, (size_t)n * sizeof(int));
}

void fo_machine_set_random_state(fo_machine* m, uint64_t state) { m->rng = state; }

int fo_layer_add_transition(fo_machine* m, int layer, int source, int dest, float time,
                            const int* logic, int n_logic) {
    fo_layer* L = &m->layers[layer];
    L->transitions = (fo_transition*)realloc(L->transitions, (size_t)(L->n_transitions + 1) * sizeof(fo_transition));
    fo_transition* t = &L->transitions[L->n_transitions];
    memset(t, 0, sizeof *t);
    t->source = source; t->dest = dest; t->transition_time = time;
    t->logic = (int*)dup_mem(logic, (size_t)n_logic * sizeof(int));
    t->n_logic = n_logic;
    return L->n_transitions++;
}

int fo_layer_pop_event(fo_machine* m, int layer, int out[3]) {
    fo_layer* L = &m->layers[layer];
    if (!L->n_events) return 0;
    memcpy(out, L->events[L->ev_head], sizeof(int[3]));
    L->ev_head = (L->ev_head + 1) % FO_LAYER_EVENT_LIMIT;
    --L->n_events;
    return 1;
}
int fo_layer_active_state(const fo_machine* m, int layer) { return m->layers[layer].active_state; }
int fo_layer_active_transition(const fo_machine* m, int layer) { return m->layers[layer].active_transition; }
const fo_pose* fo_layer_pose(const fo_machine* m, int layer) { return m->layers[layer].final_pose; }
const fo_pose* fo_machine_pose(const fo_machine* m) { return m->final_pose; }

static const fo_param* get_param(const fo_machine* m, int idx) {
    return (idx >= 0 && idx < m->n_params) ? &m->params[idx] : NULL;
}

/* transition.rs:141-173 LogicNode::calculate_value over the prefix encoding */
static int logic_eval(const int* code, int n, int* pc, const fo_machine* m,
                      fo_animation* const* anims, int n_anims) {
    if (*pc >= n) return 0;
    int op = code[(*pc)++];
    switch (op) {
    case FO_LOGIC_PARAM: {
        int idx = (*pc < n) ? code[(*pc)++] : -1;
        const fo_param* p = get_param(m, idx);
        return (p && p->kind == FO_PARAM_RULE) ? (p->u != 0) : 0;
    }
    case FO_LOGIC_AND: { int l = logic_eval(code, n, pc, m, anims, n_anims); int r = logic_eval(code, n, pc, m, anims, n_anims); return l & r; }
    case FO_LOGIC_OR:  { int l = logic_eval(code, n, pc, m, anims, n_anims); int r = logic_eval(code, n, pc, m, anims, n_anims); return l | r; }
    case FO_LOGIC_XOR: { int l = logic_eval(code, n, pc, m, anims, n_anims); int r = logic_eval(code, n, pc, m, anims, n_anims); return l ^ r; }
    case FO_LOGIC_NOT: return !logic_eval(code, n, pc, m, anims, n_anims);
    case FO_LOGIC_IS_ANIMATION_ENDED: {
        int a = (*pc < n) ? code[(*pc)++] : -1;
        /* .ok().is_none_or(|a| a.has_ended()): an invalid handle counts as ended */
        if (a < 0 || a >= n_anims || !anims[a]) return 1;
        return fo_animation_has_ended(anims[a]);
    }
    default: return 0;
    }
}

/* fyrox-math/src/lib.rs:291-313, :326-328 */
static void barycentric_2d(const float p[2], const float a[2], const float b[2], const float c[2], float out[3]) {
    float v0[2] = { b[0] - a[0], b[1] - a[1] };
    float v1[2] = { c[0] - a[0], c[1] - a[1] };
    float v2[2] = { p[0] - a[0], p[1] - a[1] };
    float d00 = v0[0] * v0[0] + v0[1] * v0[1];
    float d01 = v0[0] * v1[0] + v0[1] * v1[1];
    float d11 = v1[0] * v1[0] + v1[1] * v1[1];
    float d20 = v2[0] * v0[0] + v2[1] * v0[1];
    float d21 = v2[0] * v1[0] + v2[1] * v1[1];
    float inv_denom = 1.0f / (d00 * d11 - d01 * d01);
    float v = (d11 * d20 - d01 * d21) * inv_denom;
    float w = (d00 * d21 - d01 * d20) * inv_denom;
    out[0] = 1.0f - v - w; out[1] = v; out[2] = w;
}

/* blendspace.rs:338-414 BlendSpace::fetch_weights. Returns 0 for None. */
int fo_blend_space_fetch_weights(int n_points, const float* pts, int n_tris, const uint32_t* tris,
                                 const float sp[2], int idx[3], float w[3]) {
    if (n_points == 0) return 0;
    if (n_points == 1) { idx[0] = idx[1] = idx[2] = 0; w[0] = 1.0f; w[1] = 0.0f; w[2] = 0.0f; return 1; }
    if (n_points == 2) {
        float e[2] = { pts[2] - pts[0], pts[3] - pts[1] };
        float tp[2] = { sp[0] - pts[0], sp[1] - pts[1] };
        float t = (tp[0] * e[0] + tp[1] * e[1]) / (e[0] * e[0] + e[1] * e[1]);
        if (t >= 0.0f && t <= 1.0f) {
            idx[0] = 0; idx[1] = 1; idx[2] = 0; w[0] = 1.0f - t; w[1] = t; w[2] = 0.0f;
            return 1;
        }
    }
    for (int k = 0; k < n_tris; ++k) {
        uint32_t ia = tris[k * 3], ib = tris[k * 3 + 1], ic = tris[k * 3 + 2];
        float bc[3];
        barycentric_2d(sp, &pts[ia * 2], &pts[ib * 2], &pts[ic * 2], bc);
        if (bc[0] >= 0.0f && bc[1] >= 0.0f && bc[0] + bc[1] < 1.0f) {
            idx[0] = (int)ia; idx[1] = (int)ib; idx[2] = (int)ic;
            w[0] = bc[0]; w[1] = bc[1]; w[2] = bc[2];
            return 1;
        }
    }
    float min_distance = FLT_MAX;
    int found = 0;
    for (int k = 0; k < n_tris; ++k) {
        for (int e = 0; e < 3; ++e) {
            uint32_t a = tris[k * 3 + e], b = tris[k * 3 + (e + 1) % 3];
            const float* pa = &pts[a * 2];
            const float* pb = &pts[b * 2];
            float edge[2] = { pb[0] - pa[0], pb[1] - pa[1] };
            float tp[2] = { sp[0] - pa[0], sp[1] - pa[1] };
            float t = (tp[0] * edge[0] + tp[1] * edge[1]) / (edge[0] * edge[0] + edge[1] * edge[1]);
            if (t >= 0.0f && t <= 1.0f) {
                float proj[2] = { pa[0] + edge[0] * t, pa[1] + edge[1] * t };
                float dx = sp[0] - proj[0], dy = sp[1] - proj[1];
                float distance = sqrtf(dx * dx + dy * dy); /* metric_distance = (a-b).norm() */
                if (distance < min_distance) {
                    min_distance = distance;
                    idx[0] = (int)a; idx[1] = (int)b; idx[2] = (int)b;
                    w[0] = 1.0f - t; w[1] = t; w[2] = 0.0f;
                    found = 1;
                }
            }
        }
    }
    return found;
}

typedef struct eval_ctx {
    fo_machine* m;
    fo_layer* L;
    fo_animation* const* anims;
    int n_anims;
    float dt;
} eval_ctx;

/* AnimationPoseSource::eval_pose for every node type; returns the node's cached output. */
static const fo_pose* node_eval(eval_ctx* c, int handle) {
    if (handle < 0 || handle >= c->L->n_nodes) return NULL; /* nodes.try_borrow failed */
    fo_pose_node* n = &c->L->nodes[handle];
    switch (n->type) {
    case FO_NODE_PLAY: /* play.rs:86-100: a stale output is kept when the animation handle is invalid */
        if (n->animation >= 0 && n->animation < c->n_anims && c->anims[n->animation]) {
            const fo_animation* an = c->anims[n->animation];
            fo_pose_clone_into(an->pose, n->output);
            /* :97 output_pose.set_root_motion(animation.root_motion().cloned()) */
            fo_pose_set_root_motion(n->output, an->has_root_motion, an->root_motion.delta_position,
                                    an->root_motion.delta_rotation);
        }
        return n->output;
    case FO_NODE_BLEND: /* blend.rs:136-164 */
        fo_pose_reset(n->output);
        for (int i = 0; i < n->n_inputs; ++i) {
            float weight;
            if (n->inputs[i].weight_param < 0) {
                weight = n->inputs[i].weight_const;
            } else {
                const fo_param* p = get_param(c->m, n->inputs[i].weight_param);
                weight = (p && p->kind == FO_PARAM_WEIGHT) ? p->f[0] : 0.0f;
            }
            const fo_pose* src = node_eval(c, n->inputs[i].source);
            if (src) fo_pose_blend_with(n->output, src, weight);
        }
        return n->output;
    case FO_NODE_BLEND_BY_INDEX: { /* blend.rs:306-361 */
        fo_pose_reset(n->output);
        const fo_param* p = get_param(c->m, n->param);
        if (p && p->kind == FO_PARAM_INDEX) {
            uint32_t current = p->u;
            int applied = 0;
            if (n->has_prev_index) {
                if (n->prev_index != current) {
                    if (n->prev_index < (uint32_t)n->n_inputs && current < (uint32_t)n->n_inputs) {
                        const fo_blend_input* prev_in = &n->inputs[n->prev_index];
                        const fo_blend_input* cur_in = &n->inputs[current];
                        float bt = n->blend_time + c->dt;       /* (blend_time + dt).min(cur.blend_time) */
                        if (cur_in->blend_time < bt) bt = cur_in->blend_time; /* f32::min */
                        n->blend_time = bt;
                        float interpolator = n->blend_time / cur_in->blend_time;
                        const fo_pose* pp = node_eval(c, prev_in->source); /* nodes[..]: index panics if invalid */
                        if (pp) fo_pose_blend_with(n->output, pp, 1.0f - interpolator);
                        const fo_pose* cp = node_eval(c, cur_in->source);
                        if (cp) fo_pose_blend_with(n->output, cp, interpolator);
                        if (interpolator >= 1.0f) {
                            n->prev_index = current;
                            n->blend_time = 0.0f;
                        }
                        applied = 1;
                    }
                }
            } else {
                n->has_prev_index = 1;
                n->prev_index = current;
            }
            if (!applied) {
                n->blend_time = 0.0f;
                if (current < (uint32_t)n->n_inputs) {
                    const fo_pose* cp = node_eval(c, n->inputs[current].source);
                    if (cp) fo_pose_clone_into(cp, n->output);
                }
            }
        }
        return n->output;
    }
    case FO_NODE_BLEND_SPACE: { /* blendspace.rs:118-150 */
        fo_pose_reset(n->output);
        const fo_param* p = get_param(c->m, n->param);
        if (p && p->kind == FO_PARAM_SAMPLING_POINT) {
            int idx[3]; float w[3];
            if (fo_blend_space_fetch_weights(n->n_inputs, n->points, n->n_tris, n->tris, p->f, idx, w)) {
                int sa = n->inputs[idx[0]].source, sb = n->inputs[idx[1]].source, sc = n->inputs[idx[2]].source;
                int ok = sa >= 0 && sa < c->L->n_nodes && sb >= 0 && sb < c->L->n_nodes && sc >= 0 && sc < c->L->n_nodes;
                if (ok) {
                    fo_pose_blend_with(n->output, node_eval(c, sa), w[0]);
                    fo_pose_blend_with(n->output, node_eval(c, sb), w[1]);
                    fo_pose_blend_with(n->output, node_eval(c, sc), w[2]);
                }
            }
        }
        return n->output;
    }
    default: return NULL;
    }
}

/* ---- collect_animation_events (node/play.rs:106-122, blend.rs:172-222, :370-438, blendspace.rs:157-218) ---- */
typedef struct ev_out { int* pairs; int cap, n; } ev_out; /* (animation, signal index) pairs */

static void ev_push(ev_out* o, int anim, int sig) {
    if (o->n < o->cap) { o->pairs[o->n * 2] = anim; o->pairs[o->n * 2 + 1] = sig; }
    ++o->n;
}

static void node_collect_events(const fo_machine* m, const fo_layer* L, int handle, fo_animation* const* anims,
                                int n_anims, int strategy, ev_out* out) {
    if (handle < 0 || handle >= L->n_nodes) return; /* nodes.try_borrow failed */
    const fo_pose_node* n = &L->nodes[handle];
    switch (n->type) {
    case FO_NODE_PLAY: /* every queued event of the animation, in queue order; nothing is removed */
        if (n->animation >= 0 && n->animation < n_anims && anims[n->animation]) {
            const fo_animation* a = anims[n->animation];
            for (int i = 0; i < a->n_events; ++i) ev_push(out, n->animation, a->events[(a->ev_head + i) % a->ev_cap]);
        }
        return;
    case FO_NODE_BLEND: {
        if (strategy == FO_EVENTS_ALL) {
            for (int i = 0; i < n->n_inputs; ++i) node_collect_events(m, L, n->inputs[i].source, anims, n_anims, strategy, out);
            return;
        }
        /* filter_map(weight.value(params)) then max_by (LAST of equal maxima) / min_by (FIRST of equal minima),
         * partial_cmp().unwrap_or(Equal) */
        int best = -1; float bw = 0.0f;
        for (int i = 0; i < n->n_inputs; ++i) {
            float w;
            if (n->inputs[i].weight_param < 0) w = n->inputs[i].weight_const;
            else {
                const fo_param* p = get_param(m, n->inputs[i].weight_param);
                if (!p || p->kind != FO_PARAM_WEIGHT) continue; /* value() -> None */
                w = p->f[0];
            }
            if (best < 0) { best = i; bw = w; continue; }
            if (strategy == FO_EVENTS_MAX_WEIGHT) { if (!(w < bw)) { best = i; bw = w; } } /* >= or unordered: later wins */
            else { if (w < bw) { best = i; bw = w; } }                                      /* strictly smaller: first wins */
        }
        if (best >= 0) node_collect_events(m, L, n->inputs[best].source, anims, n_anims, strategy, out);
        return;
    }
    case FO_NODE_BLEND_BY_INDEX: {
        const fo_param* p = get_param(m, n->param);
        if (!p || p->kind != FO_PARAM_INDEX || !n->has_prev_index) return;
        uint32_t cur = p->u;
        if (n->prev_index != cur) {
            if (n->prev_index < (uint32_t)n->n_inputs && cur < (uint32_t)n->n_inputs) {
                const fo_blend_input* pi = &n->inputs[n->prev_index];
                const fo_blend_input* ci = &n->inputs[cur];
                float interpolator = n->blend_time / ci->blend_time;
                if (strategy == FO_EVENTS_ALL) {
                    node_collect_events(m, L, pi->source, anims, n_anims, strategy, out);
                    node_collect_events(m, L, ci->source, anims, n_anims, strategy, out);
                } else if (strategy == FO_EVENTS_MAX_WEIGHT) {
                    node_collect_events(m, L, (interpolator < 0.5f ? pi : ci)->source, anims, n_anims, strategy, out);
                } else {
                    node_collect_events(m, L, (interpolator < 0.5f ? ci : pi)->source, anims, n_anims, strategy, out);
                }
            }
        } else if (cur < (uint32_t)n->n_inputs) {
            node_collect_events(m, L, n->inputs[cur].source, anims, n_anims, strategy, out);
        }
        return;
    }
    case FO_NODE_BLEND_SPACE: {
        const fo_param* p = get_param(m, n->param);
        if (!p || p->kind != FO_PARAM_SAMPLING_POINT) return;
        int idx[3]; float w[3];
        if (!fo_blend_space_fetch_weights(n->n_inputs, n->points, n->n_tris, n->tris, p->f, idx, w)) return;
        int src[3] = { n->inputs[idx[0]].source, n->inputs[idx[1]].source, n->inputs[idx[2]].source };
        for (int k = 0; k < 3; ++k) if (src[k] < 0 || src[k] >= L->n_nodes) return;
        if (strategy == FO_EVENTS_ALL) {
            for (int k = 0; k < 3; ++k) node_collect_events(m, L, src[k], anims, n_anims, strategy, out);
            return;
        }
        int best = 0;
        for (int k = 1; k < 3; ++k) {
            if (strategy == FO_EVENTS_MAX_WEIGHT) { if (!(w[k] < w[best])) best = k; }
            else { if (w[k] < w[best]) best = k; }
        }
        node_collect_events(m, L, src[best], anims, n_anims, strategy, out);
        return;
    }
    default: return;
    }
}

/* layer.rs:308-401 MachineLayer::collect_active_animations_events.  source[4] = {kind 0 Invalid / 1 State /
 * 2 Transition, handle, source state, dest state}; returns the number of events (pairs hold at most cap). */
int fo_layer_collect_active_animations_events(const fo_machine* m, int layer, fo_animation* const* anims, int n_anims,
                                              int strategy, int* pairs, int cap, int source[4]) {
    const fo_layer* L = &m->layers[layer];
    ev_out out = { pairs, cap, 0 };
    source[0] = 0; source[1] = source[2] = source[3] = -1;
    if (L->active_state >= 0 && L->active_state < L->n_states) {
        source[0] = 1; source[1] = L->active_state;
        node_collect_events(m, L, L->states[L->active_state].root, anims, n_anims, strategy, &out);
    } else if (L->active_transition >= 0 && L->active_transition < L->n_transitions) {
        const fo_transition* tr = &L->transitions[L->active_transition];
        if (tr->source >= 0 && tr->source < L->n_states && tr->dest >= 0 && tr->dest < L->n_states) {
            source[0] = 2; source[1] = L->active_transition; source[2] = tr->source; source[3] = tr->dest;
            if (strategy == FO_EVENTS_ALL) {
                node_collect_events(m, L, L->states[tr->source].root, anims, n_anims, strategy, &out);
                node_collect_events(m, L, L->states[tr->dest].root, anims, n_anims, strategy, &out);
            } else {
                int pick_source = strategy == FO_EVENTS_MAX_WEIGHT ? tr->blend_factor < 0.5f : !(tr->blend_factor < 0.5f);
                node_collect_events(m, L, L->states[pick_source ? tr->source : tr->dest].root, anims, n_anims, strategy, &out);
            }
        }
    }
    return out.n;
}

/* node/mod.rs:116-150 collect_animations (set semantics via the `seen` array) */
static void node_collect(const fo_layer* L, int handle, unsigned char* seen, int n_anims) {
    if (handle < 0 || handle >= L->n_nodes) return;
    const fo_pose_node* n = &L->nodes[handle];
    if (n->type == FO_NODE_PLAY) {
        if (n->animation >= 0 && n->animation < n_anims) seen[n->animation] = 1;
        return;
    }
    for (int i = 0; i < n->n_inputs; ++i) node_collect(L, n->inputs[i].source, seen, n_anims);
}

/* The reference's EnableRandomAnimation draws from rand::thread_rng() (state.rs:109), which cannot be restated: what
 * is restated is everything around the draw (which handles, when, what an invalid or absent choice does) with the
 * generator the product documents in fyrox_hip.h -- splitmix64, index = high word of draw * n. */
static uint64_t splitmix64_next(uint64_t* state) {
    uint64_t z = (*state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* state.rs:88-116 StateAction::apply */
static void apply_actions(fo_machine* m, const fo_action* acts, int n, fo_animation* const* anims, int n_anims) {
    for (int i = 0; i < n; ++i) {
        if (acts[i].kind == FO_ACTION_ENABLE_RANDOM) {
            if (acts[i].n_choices <= 0) continue; /* choose() of an empty iterator: None */
            const uint64_t draw = splitmix64_next(&m->rng);
            const int pick = acts[i].choices[(uint32_t)(((unsigned __int128)draw * (uint32_t)acts[i].n_choices) >> 64)];
            if (pick >= 0 && pick < n_anims && anims[pick]) fo_animation_set_enabled(anims[pick], 1);
            continue;
        }
        int a = acts[i].animation;
        if (a < 0 || a >= n_anims || !anims[a]) continue;
        switch (acts[i].kind) {
        case FO_ACTION_REWIND: fo_animation_rewind(anims[a]); break;
        case FO_ACTION_ENABLE: fo_animation_set_enabled(anims[a], 1); break;
        case FO_ACTION_DISABLE: fo_animation_set_enabled(anims[a], 0); break;
        default: break;
        }
    }
}

/* layer.rs:590-706 MachineLayer::evaluate_pose */
static const fo_pose* layer_evaluate(fo_machine* m, fo_layer* L, fo_animation* const* anims, int n_anims, float dt) {
    fo_pose_reset(L->final_pose);
    if (L->active_state >= 0 || L->active_transition >= 0) {
        eval_ctx c = { m, L, anims, n_anims, dt };
        for (int s = 0; s < L->n_states; ++s) node_eval(&c, L->states[s].root); /* state.update */

        if (L->active_transition < 0) {
            for (int t = 0; t < L->n_transitions; ++t) {
                fo_transition* tr = &L->transitions[t];
                if (tr->dest == L->active_state || tr->source != L->active_state) continue;
                int pc = 0;
                if (logic_eval(tr->logic, tr->n_logic, &pc, m, anims, n_anims)) {
                    if (L->active_state >= 0 && L->active_state < L->n_states)
                        apply_actions(m, L->states[L->active_state].leave, L->states[L->active_state].n_leave, anims, n_anims);
                    layer_push_event(L, FO_EVENT_STATE_LEAVE, L->active_state, -1);      /* :620 */
                    if (tr->dest >= 0 && tr->dest < L->n_states)
                        apply_actions(m, L->states[tr->dest].enter, L->states[tr->dest].n_enter, anims, n_anims);
                    layer_push_event(L, FO_EVENT_STATE_ENTER, tr->dest, -1);             /* :634 */
                    L->active_state = -1;
                    L->active_transition = t;
                    layer_push_event(L, FO_EVENT_ACTIVE_TRANSITION_CHANGED, t, -1);      /* :645 */
                    break;
                }
            }
        }

        if (L->active_transition >= 0) {
            fo_transition* tr = &L->transitions[L->active_transition];
            /* states[..].pose(&nodes) = root.pose(): the cached output, no re-evaluation */
            if (tr->source >= 0 && tr->source < L->n_states) {
                int r = L->states[tr->source].root;
                if (r >= 0 && r < L->n_nodes) fo_pose_blend_with(L->final_pose, L->nodes[r].output, 1.0f - tr->blend_factor);
            }
            if (tr->dest >= 0 && tr->dest < L->n_states) {
                int r = L->states[tr->dest].root;
                if (r >= 0 && r < L->n_nodes) fo_pose_blend_with(L->final_pose, L->nodes[r].output, tr->blend_factor);
            }
            /* transition.rs:315-321 update */
            tr->elapsed_time += dt;
            if (tr->elapsed_time > tr->transition_time) tr->elapsed_time = tr->transition_time;
            tr->blend_factor = tr->elapsed_time / tr->transition_time;
            /* :301-303 is_done */
            if (fabsf(tr->transition_time - tr->elapsed_time) <= FLT_EPSILON) {
                tr->elapsed_time = 0.0f; /* reset */
                tr->blend_factor = 0.0f;
                L->active_transition = -1;
                layer_push_event(L, FO_EVENT_ACTIVE_TRANSITION_CHANGED, -1, -1);         /* :673 */
                L->active_state = tr->dest;
                layer_push_event(L, FO_EVENT_ACTIVE_STATE_CHANGED, tr->source, tr->dest); /* :677 */
            }
        } else {
            if (L->active_state >= 0 && L->active_state < L->n_states) {
                int r = L->states[L->active_state].root;
                if (r >= 0 && r < L->n_nodes) fo_pose_clone_into(L->nodes[r].output, L->final_pose);
            }
        }
    }
    pose_drop_nodes(L->final_pose, L->excluded, L->n_excluded);
    return L->final_pose;
}

/* machine/mod.rs:344-382 Machine::evaluate_pose */
const fo_pose* fo_machine_evaluate_pose(fo_machine* m, fo_animation* const* anims, int n_anims, float dt) {
    fo_pose_reset(m->final_pose);
    unsigned char* seen = (unsigned char*)calloc((size_t)(n_anims ? n_anims : 1), 1);
    for (int l = 0; l < m->n_layers; ++l) {
        fo_layer* L = &m->layers[l];
        int check[3] = { L->active_state, -1, -1 };
        if (L->active_transition >= 0 && L->active_transition < L->n_transitions) {
            check[1] = L->transitions[L->active_transition].source;
            check[2] = L->transitions[L->active_transition].dest;
        }
        for (int k = 0; k < 3; ++k)
            if (check[k] >= 0 && check[k] < L->n_states) node_collect(L, L->states[check[k]].root, seen, n_anims);
    }
    for (int a = 0; a < n_anims; ++a)
        if (seen[a] && anims[a] && anims[a]->enabled) fo_animation_tick(anims[a], dt);
    free(seen);
    for (int l = 0; l < m->n_layers; ++l) {
        fo_layer* L = &m->layers[l];
        float weight = L->weight;
        const fo_pose* pose = layer_evaluate(m, L, anims, n_anims, dt);
        fo_pose_blend_with(m->final_pose, pose, weight);
    }
    return m->final_pose;
}
