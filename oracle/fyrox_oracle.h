/*
 * fyrox_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE ONLY, NOT PRODUCT CODE).
 *
 * A plain-C, scalar-f32 restatement of the Fyrox skeletal-animation hot path
 * (pose sampling/blending -> local TRS -> hierarchy -> bone palette -> 4-weight
 * linear-blend skinning).  Each function cites the reference file:line it
 * follows (paths relative to /root/reference).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may
 * load this library.  The product path (fyrox_amd/ + libfyrox_hip.so) never
 * links, imports or calls anything in oracle/.
 *
 * PARITY PINNING.  The reference is Rust and cannot be compiled here (no
 * rustc/cargo), and its arithmetic leaves live in the un-vendored crate
 * nalgebra 0.35 (fyrox-math/Cargo.toml; no Cargo.lock).  So:
 *   - PINNED by the reference's own unit-test vectors (tests/golden/ .json files,
 *     transcribed from the cited #[test] bodies): Curve::value_at /
 *     CurveKey::interpolate (fyrox-math/src/curve.rs:429-566), wrapf
 *     (fyrox-math/src/lib.rs:1142-1147), quat_from_euler (lib.rs:1462-1478),
 *     BlendSpace::fetch_weights (machine/node/blendspace.rs:456-537),
 *     vertex-buffer fixture (scene/mesh/buffer.rs:1678-1720),
 *     graph hierarchy (scene/graph/mod.rs:2602-2739).
 *   - PARITY UNPINNED (no reference test asserts any output): LBS, palette,
 *     Transform::calculate_local_transform, every blend_with, Machine.  For
 *     those the oracle's authority is line-by-line fidelity to the cited code
 *     and to nalgebra's published operation order (restated in the comments),
 *     plus the self-consistency properties in tests/test_oracle_*.py.
 *
 * Build: `make -C oracle` (gcc -O2 -ffp-contract=off: Rust never fuses a*b+c).
 * All matrices are nalgebra-layout column-major float[16]: m[col*4 + row].
 * Quaternions are nalgebra storage order float[4] = (i, j, k, w).
 */
#ifndef FYROX_ORACLE_H
#define FYROX_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------- nalgebra leaves ---------- */
void fo_mat4_identity(float out[16]);
void fo_mat4_mul(const float a[16], const float b[16], float out[16]);
void fo_mat4_transform_point(const float m[16], const float p[3], float out[3]);
void fo_mat4_transform_vector(const float m[16], const float v[3], float out[3]);
void fo_mat3_of_mat4_mul_vec(const float m[16], const float v[3], float out[3]);
float fo_vec4_dot(const float a[4], const float b[4]);
void fo_quat_mul(const float a[4], const float b[4], float out[4]);
void fo_quat_from_axis_angle(int axis, float angle, float out[4]);
void fo_quat_normalize(const float q[4], float out[4]);
void fo_quat_to_mat3(const float q[4], float out9_colmajor[9]);
void fo_vec_lerp(const float* a, const float* b, float t, int n, float* out);
void fo_quat_nlerp_shortest(const float a[4], const float b[4], float w, float out[4]);

/* ---------- fyrox-math ---------- */
float fo_lerpf(float a, float b, float t);
float fo_cubicf(float p0, float p1, float t, float m0, float m1);
float fo_wrapf(float n, float min_limit, float max_limit);
float fo_stepf(float p0, float p1, float t);
/* order: 0=XYZ 1=XZY 2=YZX 3=YXZ 4=ZXY 5=ZYX */
void fo_quat_from_euler(const float euler[3], int order, float out[4]);

/* Curve key kinds */
enum { FO_KEY_CONSTANT = 0, FO_KEY_LINEAR = 1, FO_KEY_CUBIC = 2 };

typedef struct fo_curve {
    uint32_t n_keys;
    const float* location;     /* sorted ascending */
    const float* value;
    const uint8_t* kind;       /* FO_KEY_* */
    const float* left_tangent; /* only read for FO_KEY_CUBIC keys */
    const float* right_tangent;
} fo_curve;

float fo_key_interpolate(float left_value, int left_kind, float left_right_tangent,
                         float right_value, int right_kind, float right_left_tangent, float t);
float fo_curve_value_at(const fo_curve* c, float location, size_t* hint);

/* TrackValueKind */
enum {
    FO_KIND_REAL = 0, FO_KIND_VEC2 = 1, FO_KIND_VEC3 = 2, FO_KIND_VEC4 = 3,
    FO_KIND_QUAT_EULER = 4, FO_KIND_QUAT = 5
};
/* returns number of floats written to out (0 == None) */
int fo_track_fetch(const fo_curve* curves, uint32_t n_curves, int kind, float time,
                   size_t hints[4], float out[4]);

/* ---------- scene ---------- */
typedef struct fo_transform {
    float local_position[3];
    float local_rotation[4];
    float local_scale[3];
    float pre_rotation[4];
    float post_rotation_matrix[9]; /* Matrix3, column-major; identity by default */
    float rotation_offset[3];
    float rotation_pivot[3];
    float scaling_offset[3];
    float scaling_pivot[3];
} fo_transform;
void fo_transform_default(fo_transform* t);
void fo_calculate_local_transform(const fo_transform* t, float out[16]);
/* parent[i] < i or -1 (root): global = parent.global * local, depth first == topological */
void fo_update_global_transforms(const float* local, const int32_t* parent, uint32_t n, float* global);
void fo_palette(const float* global, const float* inv_bind, uint32_t n, float* out);

/* ---------- LBS ---------- */
/* packed streams: pos 3N, nrm 3N, tan 4N (w passed through), weights 4N, indices 4N u8.
 * any of nrm/tan/out_* may be NULL.  Returns 0, or -1 if an index >= n_bones. */
int fo_lbs_skin(uint32_t n_verts, const float* pos, const float* nrm, const float* tan,
                const float* weights, const uint8_t* indices,
                const float* palette, uint32_t n_bones,
                float* out_pos, float* out_nrm, float* out_tan);
/* OpenMP variant (NOT in the reference; labelled as such) */
int fo_lbs_skin_omp(uint32_t n_verts, const float* pos, const float* nrm, const float* tan,
                    const float* weights, const uint8_t* indices,
                    const float* palette, uint32_t n_bones,
                    float* out_pos, float* out_nrm, float* out_tan, int n_threads);
/* Mesh::accurate_world_bounding_box skinned branch over an AoS vertex buffer.
 * aabb = {min xyz, max xyz}.  Returns vertices consumed. */
uint32_t fo_accurate_world_bounding_box(const uint8_t* aos, uint32_t n_verts, uint32_t stride,
                                        int off_pos, int off_weights, int off_indices,
                                        const float* palette, uint32_t n_bones, float aabb[6]);
int fo_omp_max_threads(void);
/* blend shapes ahead of skinning (standard.shader:167-173, surface.rs:116-217) */
float fo_half_to_float(uint16_t h);
void fo_apply_blend_shapes(uint32_t n_verts, const float* pos, const float* nrm, const float* tan,
                           const uint16_t* storage, uint32_t plane_vertices, uint32_t n_shapes,
                           const float* weights, float* out_pos, float* out_nrm, float* out_tan);

/* ---------- fyrox-animation pose path (fyrox_oracle_anim.c) ---------- */
/* ValueBinding (value.rs:355-373): ids >= FO_BIND_PROPERTY0 stand for distinct Property{name,..} */
enum { FO_BIND_POSITION = 0, FO_BIND_SCALE = 1, FO_BIND_ROTATION = 2, FO_BIND_PROPERTY0 = 3 };
/* TrackValue variants (value.rs:199-214) */
enum { FO_VAL_REAL = 0, FO_VAL_VEC2 = 1, FO_VAL_VEC3 = 2, FO_VAL_VEC4 = 3, FO_VAL_QUAT = 4 };
typedef struct fo_bound_value { int binding; int kind; float v[4]; } fo_bound_value;

typedef struct fo_pose fo_pose;           /* AnimationPose<T>, T = small non-negative int */
fo_pose* fo_pose_new(void);
void fo_pose_free(fo_pose*);
void fo_pose_reset(fo_pose*);
void fo_pose_add(fo_pose*, int node, const fo_bound_value*);
void fo_pose_clone_into(const fo_pose* src, fo_pose* dst);
void fo_pose_blend_with(fo_pose* self, const fo_pose* other, float weight);
int fo_pose_node_capacity(const fo_pose*);
int fo_pose_value_count(const fo_pose*, int node);
int fo_pose_get_value(const fo_pose*, int node, int i, fo_bound_value* out);
void fo_pose_apply(const fo_pose*, fo_transform* nodes, int n_nodes);
/* pose.rs:78-85; root_motion() returns 0 for None (outputs then hold RootMotion::default()) */
void fo_pose_set_root_motion(fo_pose*, int has, const float delta_position[3], const float delta_rotation[4]);
int fo_pose_root_motion(const fo_pose*, float delta_position[3], float delta_rotation[4]);

typedef struct fo_tracks fo_tracks;       /* AnimationTracksData */
fo_tracks* fo_tracks_new(void);
void fo_tracks_free(fo_tracks*);
int fo_tracks_add_track(fo_tracks*, int binding, int kind, uint32_t n_curves, const fo_curve* curves);
int fo_tracks_count(const fo_tracks*);

typedef struct fo_animation fo_animation; /* Animation<T> */
fo_animation* fo_animation_new(const fo_tracks*);
void fo_animation_free(fo_animation*);
void fo_animation_bind(fo_animation*, int track, int target_node, int enabled);
void fo_animation_set_time_position(fo_animation*, float);
void fo_animation_set_time_slice(fo_animation*, float start, float end);
void fo_animation_set_speed(fo_animation*, float);
void fo_animation_set_loop(fo_animation*, int);
void fo_animation_set_enabled(fo_animation*, int);
void fo_animation_rewind(fo_animation*);
float fo_animation_time_position(const fo_animation*);
int fo_animation_is_enabled(const fo_animation*);
int fo_animation_has_ended(const fo_animation*);
const fo_pose* fo_animation_pose(const fo_animation*);
void fo_animation_tick(fo_animation*, float dt);
/* signals (index == the {id, name} pair), events queue (lib.rs:471-496, 680-700) */
int fo_animation_add_signal(fo_animation*, float time, int enabled);
void fo_animation_set_signal_enabled(fo_animation*, int signal, int enabled);
void fo_animation_set_max_event_capacity(fo_animation*, uint32_t cap);
int fo_animation_event_count(const fo_animation*);
int fo_animation_pop_event(fo_animation*); /* signal index, -1 == None */
void fo_animation_clear_events(fo_animation*);
/* root motion (lib.rs:302-343, 498-676); node < 0 == settings None */
void fo_animation_set_root_motion_settings(fo_animation*, int node, int ignore_x, int ignore_y,
                                           int ignore_z, int ignore_rotations);
int fo_animation_root_motion(const fo_animation*, float delta_position[3], float delta_rotation[4]);

enum { FO_PARAM_WEIGHT = 0, FO_PARAM_RULE = 1, FO_PARAM_INDEX = 2, FO_PARAM_SAMPLING_POINT = 3 };
enum { FO_NODE_PLAY = 0, FO_NODE_BLEND = 1, FO_NODE_BLEND_BY_INDEX = 2, FO_NODE_BLEND_SPACE = 3 };
enum { FO_ACTION_NONE = 0, FO_ACTION_REWIND = 1, FO_ACTION_ENABLE = 2, FO_ACTION_DISABLE = 3, FO_ACTION_ENABLE_RANDOM = 4 };
/* LogicNode, prefix encoded: PARAM p | AND a b | OR a b | XOR a b | NOT a | IS_ANIMATION_ENDED anim */
enum { FO_LOGIC_PARAM = 0, FO_LOGIC_AND = 1, FO_LOGIC_OR = 2, FO_LOGIC_XOR = 3, FO_LOGIC_NOT = 4,
       FO_LOGIC_IS_ANIMATION_ENDED = 5 };

/* machine/event.rs:30-51 Event<T>: {kind, a, b}: StateEnter(a) StateLeave(a)
 * ActiveStateChanged{prev: a, new: b} ActiveTransitionChanged(a) (-1 == Handle::NONE) */
enum { FO_EVENT_STATE_ENTER = 0, FO_EVENT_STATE_LEAVE = 1, FO_EVENT_ACTIVE_STATE_CHANGED = 2,
       FO_EVENT_ACTIVE_TRANSITION_CHANGED = 3 };

typedef struct fo_machine fo_machine;     /* Machine<T> */
fo_machine* fo_machine_new(void);
void fo_machine_free(fo_machine*);
int fo_machine_add_parameter(fo_machine*, int kind, float f0, float f1, uint32_t u);
void fo_machine_set_parameter(fo_machine*, int index, int kind, float f0, float f1, uint32_t u);
int fo_machine_add_layer(fo_machine*, float weight);
void fo_layer_set_weight(fo_machine*, int layer, float w);
void fo_layer_set_mask(fo_machine*, int layer, const int* excluded_nodes, int n);
int fo_layer_add_play(fo_machine*, int layer, int animation);
int fo_layer_add_blend(fo_machine*, int layer, int n_inputs, const int* sources,
                       const int* weight_params /* -1 = constant */, const float* weight_consts);
int fo_layer_add_blend_by_index(fo_machine*, int layer, int index_param, int n_inputs,
                                const int* sources, const float* blend_times);
int fo_layer_add_blend_space(fo_machine*, int layer, int sampling_param, int n_points,
                             const float* points_xy, const int* sources, int n_tris,
                             const uint32_t* tris);
int fo_layer_add_state(fo_machine*, int layer, int root_node);
void fo_layer_set_entry_state(fo_machine*, int layer, int state);
void fo_layer_reset(fo_machine*, int layer); /* MachineLayer::reset, layer.rs:288-296 */
void fo_state_add_action(fo_machine*, int layer, int state, int on_enter, int kind, int animation);
void fo_state_add_random_action(fo_machine*, int layer, int state, int on_enter, const int* animations, int n);
void fo_machine_set_random_state(fo_machine*, uint64_t state);
int fo_layer_add_transition(fo_machine*, int layer, int source, int dest, float time,
                            const int* logic, int n_logic);
int fo_layer_pop_event(fo_machine*, int layer, int out[3]); /* 0 == None */
/* AnimationEventCollectionStrategy (node/mod.rs:177-184) and MachineLayer::collect_active_animations_events */
enum { FO_EVENTS_ALL = 0, FO_EVENTS_MAX_WEIGHT = 1, FO_EVENTS_MIN_WEIGHT = 2 };
int fo_layer_collect_active_animations_events(const fo_machine*, int layer, fo_animation* const* anims, int n_anims,
                                              int strategy, int* pairs, int cap, int source[4]);
int fo_layer_active_state(const fo_machine*, int layer);
int fo_layer_active_transition(const fo_machine*, int layer);
const fo_pose* fo_layer_pose(const fo_machine*, int layer);
const fo_pose* fo_machine_pose(const fo_machine*);
int fo_blend_space_fetch_weights(int n_points, const float* pts_xy, int n_tris, const uint32_t* tris,
                                 const float sampling_point[2], int idx[3], float w[3]);
const fo_pose* fo_machine_evaluate_pose(fo_machine*, fo_animation* const* anims, int n_anims, float dt);

/* gltf/simplify.rs:39-66 find_important_points; blendspace.rs:416-447 triangulate (see fyrox_oracle.c) */
uint32_t fo_find_important_points(const float* x, const float* y, uint32_t n, float epsilon, float max_step, uint32_t* out);
int32_t fo_blend_space_triangulate(const float* xy, uint32_t n, uint32_t* out_tri, uint32_t capacity);

#ifdef __cplusplus
}
#endif
#endif
