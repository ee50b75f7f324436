// Internal declarations shared by the HIP translation units of libfyrox_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <vector>
#include <stdint.h>

namespace fyx {

constexpr int kCUs = 256;  // MI355X

struct LbsTuning {
    int blocks_per_cu = 4;   // lbs_skin's persistent grid = kCUs * blocks_per_cu (capped by the work)
    int exact = 1;           // 1: reference operation order, unfused; 0: FMA
    int crowd = -1;          // instanced launches: -1 auto (crowd kernel from 4 instances), 0 never, 1 always
    int crowd_lean = 0;      // 1: register-lean crowd kernel at two workgroups per CU (leaves room for other kernels' waves, see lbs_kernels.hip)
    int crowd_ipb = 0;       // instances per workgroup run; 0 = auto
    int dyn = 1;             // large single-instance launches: 1 = lbs_skin_dyn (units drawn from an LDS ticket counter), 0 = lbs_skin
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;   // option lbs.timing: this launch's own start / stop events (dispatch timestamps)
};

// Option debug.timeline: the next launch made through FYX_TL_LAUNCH (anim_kernels.hip) or FYX_LAUNCH (lbs_kernels.hip, through
// LbsTuning) carries these as its own start / stop events (the dispatch's timestamps); set by the caller for ONE launch.
struct LaunchEvents { hipEvent_t start = nullptr, stop = nullptr; };
extern thread_local LaunchEvents g_launch_events;

struct LbsArgs {
    const float* pos;        // 3N packed xyz
    const float* nrm;        // 3N or null
    const float* tan;        // 4N or null
    const float* wgt;        // 4N
    const uint32_t* idx;     // N (4 x u8 packed little-endian: idx0 in the low byte)
    const float* palette;    // n_instances * n_bones * 16, column-major mat4
    float* out_pos;          // n_instances * 3N or null
    float* out_nrm;
    float* out_tan;
    uint32_t n_verts;
    uint32_t n_bones;
    uint32_t n_instances;
};

// Launch the skinning kernel.  Returns hipSuccess or the launch error.
hipError_t launch_lbs(const LbsArgs& a, const LbsTuning& t, hipStream_t stream);

// Batched launch (fyx_lbs_skin_batch): ONE launch skins many (mesh, palette) pairs.  A segment is one instance of one
// job; segments are laid end to end in a common numbering of 64-vertex units and the workgroups of the launch take
// equal contiguous unit ranges of the whole batch, exactly as lbs_skin does for the instances of one mesh.
struct LbsSegDev {
    const float* pos;
    const float* nrm;
    const float* tan;
    const float* wgt;
    const uint32_t* idx;
    const float* palette;    // this instance's n_bones matrices
    float* out_pos;          // this instance's outputs (null where the launch's mask has no bit)
    float* out_nrm;
    float* out_tan;
    uint32_t n_verts;
    uint32_t n_bones;
    uint32_t unit0;          // first unit of the segment in the batch
    uint32_t pad;
};
// grid for a batch of total_units units
uint32_t lbs_batch_grid(uint32_t total_units, const LbsTuning& t);
// d_block_seg[b] = the segment that holds workgroup b's first unit.  mask: bit 0 position, 1 normal, 2 tangent (the
// same for every segment of the launch); max_bones sizes the LDS.
hipError_t launch_lbs_batch(const LbsSegDev* d_segs, uint32_t n_segs, const uint32_t* d_block_seg, uint32_t grid,
                            uint32_t total_units, uint32_t max_bones, int mask, const LbsTuning& t, hipStream_t stream);

// Extended launch: blend shapes before skinning and / or interleaved output (lbs_skin_ex).
struct LbsExArgs {
    LbsArgs a;                   // out_* are the SoA outputs (ignored when out_aos is set)
    const uint16_t* shapes;      // re-tiled f16 offsets [shape][tile][9][64], or null
    const float* shape_w;        // [n_instances][n_shapes]
    uint32_t n_shapes;
    uint32_t tiles_per_shape;
    unsigned char* out_aos;      // interleaved output [n_instances][n_verts][out_stride], or null
    uint32_t out_stride;
    int off_pos, off_nrm, off_tan;   // byte offsets inside a vertex, -1 = do not write
    // vertex-buffer-in / vertex-buffer-out launch (launch_lbs_aos): the mesh's own interleaved bytes, same
    // stride and offsets as the output
    const unsigned char* in_aos;
    int in_off_wgt, in_off_idx;
};
hipError_t launch_lbs_ex(const LbsExArgs& x, const LbsTuning& t, hipStream_t stream);

// Batched whole-span launches (vertex buffer in -> vertex buffer out, optional blend shapes): one segment per
// (job, instance), as LbsSegDev.
struct LbsExSegDev {
    const unsigned char* in_aos;
    unsigned char* out_aos;      // this instance's output vertex buffer
    const float* palette;        // this instance's matrices
    const uint16_t* shapes;
    const float* shape_w;        // this instance's weights
    uint32_t n_verts, n_bones, n_shapes, tiles_per_shape;
    uint32_t stride;
    int32_t off_pos, off_nrm, off_tan, in_off_wgt, in_off_idx;
    uint32_t unit0, pad;
};
// The kernel variant is picked by the layout bucket (16-byte accesses per lane and span: 4, 5, 8 or 10 -- see
// lbs_aos_bucket) and whether any blend shapes are applied; every segment of a launch shares both.
uint32_t lbs_aos_bucket(uint32_t stride);   // 0 = unsupported stride
// Persistent grid of a batch launch (what is resident for its LDS footprint, capped by the work).
hipError_t lbs_aos_batch_grid(uint32_t total_units, uint32_t max_bones, uint32_t max_stride, uint32_t bucket, bool shapes,
                              const LbsTuning& t, uint32_t* grid);
hipError_t launch_lbs_aos_batch(const LbsExSegDev* d_segs, uint32_t n_segs, const uint32_t* d_block_seg, uint32_t grid,
                                uint32_t total_units, uint32_t max_bones, uint32_t max_stride, uint32_t bucket, bool shapes,
                                const LbsTuning& t, hipStream_t stream);
// whole-span variant: out = in with position / normal / tangent.xyz replaced; stride <= 160, multiple of 4
hipError_t launch_lbs_aos(const LbsExArgs& x, const LbsTuning& t, hipStream_t stream);
// engine RGB16F volume -> device tile layout (see lbs_kernels.hip)
hipError_t launch_retile_blend_shapes(const uint16_t* d_src, uint32_t n_verts, uint32_t plane_vertices,
                                      uint32_t n_shapes, uint16_t* d_dst, hipStream_t stream);

// AoS -> SoA de-interleave (device to device).  off_* in bytes, -1 = absent.
hipError_t launch_deinterleave(const uint8_t* d_aos, uint32_t n_verts, uint32_t stride,
                               int off_pos, int off_nrm, int off_tan, int off_wgt, int off_idx,
                               float* d_pos, float* d_nrm, float* d_tan, float* d_wgt,
                               uint32_t* d_idx, hipStream_t stream);

// max over all 4N index bytes -> *d_out (single uint32).
hipError_t launch_max_bone_index(const uint32_t* d_idx, uint32_t n_verts, uint32_t* d_out,
                                 hipStream_t stream);

// skinned-position AABB. d_partials: scratch of 6 * n_blocks floats (n_blocks returned by
// aabb_partial_blocks()).  Result in d_out[6] = {min xyz, max xyz}.
uint32_t aabb_partial_blocks(uint32_t n_verts);
// Per-instance boxes of an instanced mesh (a.palette = n_instances palettes): d_out[n_instances][6]; d_partials holds
// aabb_inst_slices() * n_instances * 6 floats when there is more than one slice.
uint32_t aabb_inst_slices(uint32_t n_verts, uint32_t n_instances);
hipError_t launch_skinned_aabb_inst(const LbsArgs& a, float* d_partials, float* d_out, hipStream_t s);
hipError_t launch_skinned_aabb(const LbsArgs& a, float* d_partials, float* d_out,
                               hipStream_t stream);
// min/max over an already skinned packed xyz stream of n points.
hipError_t launch_points_aabb(const float* d_xyz, uint64_t n_points, float* d_partials,
                              float* d_out, hipStream_t stream);

// out[i] = a[i] * b[i] (mat4, nalgebra operation order)
hipError_t launch_palette(const float* d_global, const float* d_inv_bind, uint32_t n,
                          float* d_out, hipStream_t stream);

// control blocks: copies `bytes` (rounded up to 16: both blocks are padded) from a pinned, device-visible host block to device memory
hipError_t launch_ctrl_copy(const void* h_src, void* d_dst, size_t bytes, hipStream_t stream);

// calibration: read 48*units bytes from d_src, write 32*units bytes to d_dst
hipError_t launch_stream_copy(const float* d_src, float* d_dst, uint32_t units, int blocks_per_cu,
                              hipStream_t stream);

}  // namespace fyx

// ---------------------------------------------------------------------------------------
// Skeletal-animation pose path (anim_kernels.hip).  All pointers device.
// ---------------------------------------------------------------------------------------
namespace fyx {

constexpr int kMaxRigNodes = 1024;   // local+global matrices of one instance live in LDS
constexpr int kMaxFoldDepth = 8;     // nested pose accumulators held in VGPRs

// One track of an AnimationTracksData: TrackValueKind + up to four curves (key ranges).
// The first and last key of every curve ride along: Curve::value_at clamps against them on every sample, and here
// they come with the cache line the sample reads anyway instead of from four more lines of the key arrays.
struct alignas(128) TrackDev {
    int32_t kind;            // FYX_KIND_*
    uint32_t n_curves;
    uint32_t first_key[4];
    uint32_t n_keys[4];
    float first_loc[4], last_loc[4];   // location of each curve's first / last key (0 for an empty curve)
    float first_val[4], last_val[4];   // and their values
};
static_assert(sizeof(TrackDev) == 128, "one cache line per track");

// One key as the per-instance sampler reads it: {value, kind bits, left tangent, right tangent} and the location in ONE
// 32-byte record, so that the two keys of a span are 64 contiguous bytes (one cache line three times out of four)
// instead of one line of the location array plus one of the value array.  The crowd sampler, which copies whole
// curves into LDS with dense loads, keeps the two arrays.
struct alignas(32) KeyRec {
    float4 aux;
    float loc;
    float pad[3];
};
static_assert(sizeof(KeyRec) == 32, "two keys per 64 bytes");

// What the per-instance sampler reads of a track FIRST (16 bytes: the three tracks of a bone share a cache line, where their
// TrackDev records are three lines), and the track's SPAN RECORDS.  Importers write the curves of a track on common key times
// (glTF samplers, FBX curve nodes); for such a track every span [key i - 1, key i) of ALL its curves is one record
//     f4 {loc[i-1], loc[i], the curves' left-key kinds (8 bits each), -}   then per curve c:
//     f4 {value[c][i-1], value[c][i], right tangent[c][i-1], left tangent[c][i] if key i is cubic else 0}
// of 64 bytes (three curves) or 80 (four) -- exactly what CurveKey::interpolate reads of the two keys (round 5; rounds 3 - 4 held both
// keys' full 16-byte records: 128 / 256 bytes per span, of which a sample used 7 words per curve) -- so a bone's ten curve samples
// touch three or four lines instead of ~12.5 of per-curve key records, and a scene of characters with their own clips moves 208
// bytes per (animation, node) instead of 512 (profiles/r03_crowd_study/fetch_granule.log: the memory system moves whole lines).  A sample whose time lies strictly inside its hinted span -- the steady state of
// playback -- needs nothing else; every other case (clamping at the ends, a hint that moved, a time exactly on a key, curves with
// their own key times) takes the general path over TrackDev / KeyRec, which decides everything in the reference's order.
struct TrackHot {
    int32_t kind;            // FYX_KIND_*
    uint32_t n_curves;
    uint32_t n_keys;         // keys per curve of a track that has span records
    uint32_t span_first;     // first f4 of the track's span records in AnimDev::spans, or kNoSpans
};
static_assert(sizeof(TrackHot) == 16, "eight tracks per cache line");
constexpr uint32_t kNoSpans = 0xffffffffu;

// What the crowd sampler needs to know about one (animation, node, binding) of an animator, in ONE scalar load: without it a wave
// finds its track through the animation's record, the node's slot table, the track's record and its TrackHot -- a chain of four
// dependent loads ahead of the first byte of key data, in a kernel that is made of such latency.
struct CrowdDesc {
    const float4* spans;     // the track's span records (TrackHot::span_first resolved); null: none, the general path samples it
    uint32_t track;          // index into the tracks data (the span hints are kept per track)
    uint32_t n_keys;         // keys per curve
    uint32_t need;           // curves the value is made of: 3 or 4
    int32_t kind;            // FYX_KIND_*
    uint32_t valid;          // bit 0: the animation provides this binding for the node (TrackDataContainer::fetch gives Some); bits 8 ..: f4 from one
                             //   span's record to the next in `spans` (need + 1 in a track's own table, the row stride in TracksData::d_span_rows)
    uint32_t present;        // the node's present bits in this animation: 1 Position, 2 Scale, 4 Rotation, 8 a Property value
};
static_assert(sizeof(CrowdDesc) == 32, "one s_load_dwordx8");

// One animation of an animator (shared by all its instances).
struct AnimDev {
    const TrackDev* tracks;
    const TrackHot* hot;         // [n_tracks]
    const float4* spans;         // span records of the tracks whose curves share their key times
    const float* key_loc;        // locations of all keys of the tracks data
    const float4* key_aux;       // {value, kind bits, left tangent, right tangent} per key
    const KeyRec* key_rec;       // the same keys, one record each
    const int32_t* slot_track;   // [n_nodes][4]: track feeding Position/Scale/Rotation of a node (-1 none); entry 3 is
                                 //   >= 0 when the animation holds a Property value for the node
    const int32_t* prop_track;   // [n_prop_slots]: Real track feeding a (node, property) slot of the animator, -1 none
    uint32_t n_tracks;
    // RootMotionSettings (lib.rs:307-319): rm_node < 0 = None; rm_ignore bits 1 x, 2 y, 4 z, 8 rotations; 16 / 32: the remainders of the last
    // loop were taken by an earlier Position / Rotation value of the root node's list (AnimationDef::multi);
    // rm_pos_track / rm_rot_track: FIRST track of the tracks data bound to Position / Rotation
    // (fetch_position_at_time / fetch_rotation_at_time, lib.rs:507-534), -1 none
    int32_t rm_node;
    uint32_t rm_ignore;
    int32_t rm_pos_track;
    int32_t rm_rot_track;
    uint32_t pad;
};

// Animation::root_motion: Option<RootMotion> (lib.rs:325-336), one per (animation, instance).
struct RootMotionDev {
    float delta_position[3];
    uint32_t has;                     // Option is Some
    float delta_rotation[4];
    float prev_position[3];
    uint32_t rem_flags;               // 1: position_offset_remainder is Some, 2: rotation_remainder is Some
    float prev_rotation[4];
    float position_offset_remainder[3];
    uint32_t pad;
    float rotation_remainder[4];
};
static_assert(sizeof(RootMotionDev) == 96, "RootMotionDev layout");

// Root-motion program ops (one uint4 each: x = opcode, y = dst slot, z = src slot / animation,
// w = f32 weight bits).  A slot is the root_motion field of one AnimationPose that outlives the
// frame: every pose node's output, every layer's final pose, the machine's final pose.
enum : uint32_t {
    RM_END = 0,
    RM_SET_ANIM = 1,   // slot[dst] = animation[src].root_motion.clone()        (play.rs:97)
    RM_BLEND = 2,      // slot[dst].get_or_insert_default().blend_with(slot[src] or default, w)  (pose.rs:98-100)
    RM_COPY = 3,       // slot[dst] = slot[src].clone()                          (pose.rs:73)
};

// Fold program ops (one uint2 each: x = opcode | a << 8, y = float weight bits).
enum : uint32_t {
    OP_END = 0,
    OP_BLEND_ANIM = 1,   // acc.blend_with(animation[a].pose, w)
    OP_PUSH = 2,         // open a nested, empty accumulator
    OP_POP_BLEND = 3,    // parent.blend_with(nested, w)
    OP_RESET = 4,        // acc.reset()
    OP_MASK = 5,         // drop the node from acc if layer mask a excludes it
    OP_APPLY = 6,        // write acc to the node's local transform
    OP_APPLY_ANIM = 7,   // write animation[a].pose to the node's local transform
};

// A palette the update kernel writes by itself at the end of every frame (fyx_animator_set_palette_output).
struct PaletteOutDev {
    const int32_t* bone_nodes;   // [n_bones] rig node of each bone, < 0: invalid handle (identity)
    float* out;                  // [n_instances][n_bones][16]
    uint32_t n_bones;
    uint32_t pad;
};
constexpr int kMaxPaletteOutputs = 4;

struct RigDev {
    const float* inv_bind;       // [n_nodes][16]
    PaletteOutDev pal[kMaxPaletteOutputs];
    uint32_t n_pal;
    const float* statics;        // [n_nodes][28]: pre_rotation(4) post_rotation_matrix(9)
                                 //   rotation_offset(3) rotation_pivot(3) scaling_offset(3) scaling_pivot(3) pad(3)
    const uint32_t* walk;        // [n_nodes] the nodes sorted by depth, one word each: node | (parent + 1) << 10 | depth << 21
                                 //   (kMaxRigNodes = 1024: 10 + 11 + 11 bits) -- one load where node -> depth, parent were two
                                 //   behind them [n_chunks][16] entries of the wide walk: node | parent slot << 11 | last-of-level << 22
    uint32_t n_nodes;
    uint32_t n_levels;
    uint32_t n_chunks;
    uint32_t pad1;               // (PoseFrameDev, RigDev) is a multiple of 16 bytes: the CtrlInline behind them is 16-byte aligned
};

// One Property{..} value: the TrackValue's f32 lanes, its variant and whether it is there (== fyx_property_value)
struct PropRec {
    float v[4];
    uint32_t present;
    uint32_t kind;       // FYX_VALUE_*
    uint32_t pad[2];
};
static_assert(sizeof(PropRec) == 32, "two 16-byte halves");

struct PoseFrameDev {
    const AnimDev* anims;
    uint32_t n_anims;
    uint32_t n_instances;
    uint32_t n_nodes;
    const float* times;          // [n_instances][n_anims] sample time of ticked animations
    const uint8_t* ticked;       // [n_instances][n_anims] bit 0: ticked this frame; bit 1: the tick started a new
                                 //   loop cycle; bit 2: speed > 0 (root motion, lib.rs:539-554)
    const uint2* ops;            // all instances' programs
    const uint32_t* prog_off;    // [n_instances] offset of each instance's program in ops
    const uint8_t* layer_masks;  // [n_layers][n_nodes] 1 = excluded
    uint32_t* hints;             // [n_anims][max_tracks][4 curves][n_instances]: instance-minor, so the crowd sampler's
                                 //   lanes (= instances of one curve) touch one dense span
    uint32_t max_tracks;
    uint32_t sample_form;        // 0 auto (instances on the lanes from 32 instances), 1 curves on the lanes, 2 instances on the lanes
    uint32_t* slot_hints;        // [n_anims][n_nodes][3 bindings][4 curves][n_instances]: the per-instance sampler's span hints, indexed by what the
                                 //   lane knows BEFORE it has its descriptor (round 5: the hint's load no longer waits for the descriptor's)
    float4* anim_pose;           // [n_anims][n_instances][n_nodes][3]
    float4* node_trs;            // [n_instances][n_nodes][3]: {pos,_} {rot} {scale,_}
    float* local;                // [n_instances][n_nodes][16]
    float* global;               // [n_instances][n_nodes][16]
    // Property{..} bindings (value.rs:221-230, :404-427): one slot per (node, property) of the animator
    uint32_t n_prop_slots;
    uint32_t shadows;            // 1: the animator keeps TWO device animations per animation -- 2 a: what the animation's pose APPLIES (per
                                 //   binding the LAST applicable value of the node's list), 2 a + 1: what a blend READS of it as `other` (the
                                 //   FIRST value of the binding; BoundValueCollection::blend_with's find, value.rs:438-444).  n_anims counts both;
                                 //   the ops of the frame's programs name 2 a.  0: one record per animation (every list holds one value per binding)
    const int32_t* prop_node;    // [n_prop_slots] node of each slot
    PropRec* prop_pose;          // [n_anims][n_instances][n_prop_slots]
    PropRec* prop_out;           // [n_instances][n_prop_slots] value applied last (present = has been applied)
    // root motion (all null / 0 unless the animator tracks root motion)
    const float2* slices;        // [n_instances][n_anims] time_slice {start, end}
    RootMotionDev* rm_anim;      // [n_anims][n_instances]
    float4* rm_slots;            // [n_instances][n_rm_slots][2]: {delta_position, has}{delta_rotation}
    uint32_t n_rm_slots;
    const uint4* rm_ops;         // all instances' root-motion programs
    const uint32_t* rm_prog_off; // [n_instances + 1]
    const CrowdDesc* crowd;      // [n_anims][n_nodes][3 bindings] (Position, Scale, Rotation): the crowd sampler's descriptors
};

// A small per-frame control block rides INSIDE the kernel arguments of the single-animator launches (one character: a few
// hundred bytes of sample times, tick flags and fold program) instead of in a device block uploaded by an H2D copy on its own
// stream -- that copy, its event and the stream wait were ~8 us of a 25 us frame.  `bytes` != 0: the control pointers of the
// PoseFrameDev beside it (times, ticked, ops, prog_off, slices, rm_ops, rm_prog_off) hold OFFSETS from the start of this struct.
constexpr uint32_t kCtrlInlineBytes = 1008;
struct alignas(16) CtrlInline {
    uint32_t bytes;          // 0: the control block is in device memory and the pointers are pointers
    uint32_t first_ops;      // 1 + the number of ops of instance 0's fold program (which starts at op 0), 0: not given --
                             //   the update kernel of ONE character then asks for its program without asking for its offsets first
    uint32_t pad[2];
    uint32_t words[kCtrlInlineBytes / 4];
};
static_assert(sizeof(CtrlInline) == 16 + kCtrlInlineBytes, "header + payload");

struct SceneJobDev;     // (below: it holds a FrameSkin)
struct SceneWait;
// The stages of a scene frame.  Each has a table of {job, x, y, z} per block (uint4), built by scene_blocks() from the
// jobs' shapes alone (so it is uploaded once per scene, not per frame) in the order the launch runs them.
enum SceneStage : int {
    kStageSample = 0,      // pose_sample, curves on the lanes      {job, node slice, instance, animation}
    kStageSampleCrowd,     // pose_sample, instances on the lanes   {job, instance slice, node * 3 + binding, animation}
    kStagePropSample,      //                                       {job, slot slice, instance, animation}
    kStageRootMotion,      //                                       {job, block, blocks of the job, -}
    kStageRootMotionFold,  //                                       {job, instance slice, -, -}
    kStageUpdate64,        // pose_update with 64 / 128 / 192 / 256 threads: {job, instance, -, -}
    kStageUpdate128,
    kStageUpdate192,
    kStageUpdate256,
    kStagePropUpdate,      //                                       {job, slot slice, instance, -}
    kStageFrame,           // the WHOLE frame of a scene of characters in one launch (scene_frame_kernel), job after job:
                           //   {job, sampler block, 0, -} ..., {job, instance, 1, -} ..., {job, skinning block, 2, -} ...
    kSceneStages
};
struct SceneJobShape {   // what the tables depend on
    uint32_t n_anims, n_instances, n_nodes, n_prop_slots, sample_form;
    bool root_motion, root_motion_program;
    uint32_t skin_blocks = 0;     // skinning workgroups of the job in the update stage (the 256-thread stage only: {job, block, 1, -} behind the job's {job, instance, 0, -})
};
// Appends job `job`'s blocks to the per-stage tables.
void scene_blocks(uint32_t job, const SceneJobShape& s, std::vector<uint4> (&tables)[kSceneStages]);
// One launch per non-empty stage.  d_tables[k] / n_blocks[k]: the device copy of stage k's table; lds_bytes[k]: dynamic
// LDS of the update stages (the largest rig of the stage).
// all_straight: every fold program of every job is straight (classify_fold_program, anim_leaves.h): the update stages run
// the kernel form without the interpreter.
// skin256: the 256-thread update stage's table also holds the jobs' skinning workgroups (SceneJobShape::skin_blocks): ONE launch updates
// and skins (the skinning workgroups recompute their character's pose on chip, FrameSkin); exact: lbs.exact for them.
// frame: null, or the scene runs as ONE launch (kStageFrame's table; every other stage's is ignored).
hipError_t launch_scene(const SceneJobDev* d_jobs, const char* d_ctrl, const uint4* const (&d_tables)[kSceneStages],
                        const uint32_t (&n_blocks)[kSceneStages], const size_t (&lds_bytes)[kSceneStages], bool all_straight, bool wide256, hipStream_t s,
                        bool skin256 = false, bool exact = true, const SceneWait* frame = nullptr);

// `inl` (optional): the frame's control block travelling in the kernel arguments, see CtrlInline.
hipError_t launch_pose_sample(const PoseFrameDev& f, hipStream_t s, const CtrlInline* inl = nullptr);
// mode: kUpdNoProgram -- the transforms as they are; kUpdGeneral -- fold programs of any shape; kUpdStraight -- the caller
// has classified EVERY instance's program as straight (classify_fold_program_host, anim_leaves.h): a kernel without the
// interpreter (a third of the registers).
// kUpdDup -- fold programs of any shape over TWO records per animation (PoseFrameDev::shadows: animations with several values of one
// binding on a node, see anim_model.h AnimationDef::dup): the plain one-workgroup-per-instance launch only.
enum : int { kUpdNoProgram = 0, kUpdGeneral = 1, kUpdStraight = 2, kUpdDup = 3 };
// Waves per workgroup of the update kernel (one workgroup per instance): one per 64 nodes, at most four -- and four for an
// animator of few instances: the chip is empty then, and the kernel's strided tail (matrix stores, palette columns: 256 columns for
// 64 bones) runs over four waves instead of one.  A crowd keeps the smallest block: its waves compete with the skinning kernel's.
// LDS of an update workgroup with the wide walk: (n_nodes + 2) local and global matrices, the chunk table + one chunk read ahead
__host__ __device__ inline size_t wide_update_lds(uint32_t n_nodes, uint32_t n_chunks) { return (size_t)(n_nodes + 2u) * 128u + (size_t)(n_chunks + 1u) * 64u; }
constexpr size_t kLdsPerWorkgroup = 160u * 1024u;
inline uint32_t update_block_waves(uint32_t n_nodes, uint32_t n_instances) {
    uint32_t w = (n_nodes + 63u) / 64u;
    if (w > 4u) w = 4u;
    if (n_instances <= 64u) w = 4u;
    return w < 1u ? 1u : w;
}
hipError_t launch_pose_update(const PoseFrameDev& f, const RigDev& rig, int mode, hipStream_t s, const CtrlInline* inl = nullptr, int pack = 0);
// One character's frame in ONE launch (option anim.one_launch): the sampler's workgroups first, the update's behind them IN THE
// SAME GRID.  An update workgroup fetches what does not depend on the sampled poses (walk table, program, statics, the node's own
// transforms), then waits until `counter` -- every sampler workgroup adds one after its records are visible device-wide -- has
// reached `target`.  Workgroups are dispatched in index order, so a waiting update workgroup never holds a place a sampler
// workgroup needs.  What it saves is the launch boundary between the two kernels (~3 us of a 12 us pose path).
// The counter is kFrameCounterReplicas words kFrameCounterStride bytes apart, all kept equal: every sampler workgroup adds one to EACH
// (sixteen lanes, one fire-and-forget atomic instruction), a waiting workgroup polls replica (its index % replicas).  A frame that
// skins has hundreds of waiting workgroups; on ONE word their polls queue in front of the samplers' adds in that word's memory
// channel (measured: the counter was seen 1.3 us later with 391 pollers than with one).
constexpr uint32_t kFrameCounterReplicas = 16, kFrameCounterStride = 4096 + 256;
struct FrameSync {
    uint32_t* counter;           // replica 0 of the animator's device counter, only ever added to
    uint32_t target;             // its value when all sampler workgroups of THIS frame have reported (wraps: compared as a signed difference)
    uint32_t n_sample_blocks;    // workgroups [0, n_sample_blocks) of the grid sample, the rest update
    uint32_t sx, sy;             // the sampler's own grid (x, y; z follows), flattened x fastest
    uint32_t timeout_ticks;      // how long a workgroup waits for the counter, in ticks of the 100 MHz wall clock (option anim.wait_timeout_ms)
    uint32_t* err;               // the context's device-error block (DeviceError below): a wait that times out reports here and the
                                 //   workgroup computes NOTHING -- the next API call returns FYX_ERR_HIP
    uint64_t tag;                // what the report names: the animator's id
};
// What a kernel reports to the host when it gives up (pinned, host-coherent memory; checked by fyx_sync and by every pose entry).
struct DeviceError {
    uint32_t code;               // 0 none; kDevErrFrameWait: an in-grid wait of a one-launch frame timed out
    uint32_t block;              // the workgroup that gave up
    uint32_t seen, target;       // the counter's value and what it was waited to reach
    uint64_t tag;                // FrameSync::tag
};
enum : uint32_t { kDevErrFrameWait = 1 };

// The frame goes on to the vertices (fyx_animator_set_skin_output): the launch that samples and updates one character ALSO holds the
// workgroups that skin its meshes.  A skinning workgroup requests its first vertices at once (60 bytes per vertex, none of them
// depends on the pose), then does what the update workgroup does -- waits for the samplers, folds, builds the local matrices, walks
// the hierarchy -- WITHOUT storing any of it, forms the palette of ITS mesh straight into the skinning kernels' LDS layout and
// skins: recomputing a 64-node pose in a workgroup that would otherwise sleep is free on an empty chip, and it takes the launch
// boundary, the palette's trip through memory and a second in-grid wait out of the character's critical path.  Same device
// functions as the update kernel and lbs_skin: same bits.
constexpr int kMaxFrameSkins = 4;
struct FrameSkinJob {
    const float* pos; const float* nrm; const float* tan; const float* wgt; const uint32_t* idx;   // the mesh's streams
    float* out_pos; float* out_nrm; float* out_tan;    // [n_instances][n_verts], null: not wanted
    const int32_t* bone_nodes;   // the job's bone list [n_bones] (rig node of each bone, < 0: identity)
    uint32_t n_verts, n_bones;
    uint32_t block0;             // the job's first workgroup among the launch's skinning workgroups
    uint32_t blocks_per_inst;    // workgroups per instance: an instance's 64-vertex units are dealt evenly over them
};
struct FrameSkin {
    FrameSkinJob job[kMaxFrameSkins];
    uint32_t n_jobs;
    uint32_t n_blocks;           // skinning workgroups of the launch
};
// Skinning workgroups of one launch: the whole grid stays resident at once -- two 256-thread workgroups per CU at the register
// budget of the kernel without the interpreter (230 VGPRs), one per CU with it (404) -- so no workgroup ever waits for one that
// has no place to run, whatever order the dispatcher takes them in.
constexpr uint32_t kFrameSkinMaxBlocks = 448, kFrameSkinMaxBlocksGeneral = 192;
constexpr uint32_t kSceneSkinMaxUnits = 4096;      // a scene's update launch takes the skin outputs along up to this many 64-vertex units (~260 k vertices): see scene_frame
constexpr uint32_t kFrameSkinAutoBlocks = 224;     // what anim.frame_skin_units = 0 aims for: a CU per skinning workgroup (256 CUs, the pose workgroups beside them)

// fyx_scene_update: one parameter block per animator of the scene, read by the *_scene_kernel forms, whose block tables say which
// job a block works for.  Its control pointers are OFFSETS into the frame's control block (the kernels get the block's address
// beside the array), so the array itself stays on the device from frame to frame and is sent again only when its bytes change:
// an animator's device state or palette outputs moved, or a fold program changed its length.
struct SceneJobDev {
    PoseFrameDev f;
    RigDev rig;
    FrameSkin sk;        // the animator's skin outputs the scene's update launch skins itself (n_jobs = 0: none)
    // the one-launch scene frame (kStageFrame): the animator's frame counter (FrameSync::counter, 16 replicas), what a report names, and
    // its sampler grid -- the frame's TARGET travels in the control block (it changes every frame, this record does not)
    uint32_t* counter;
    uint64_t tag;
    uint32_t n_sample_blocks, sx;
};
// What every workgroup of scene_frame_kernel gets besides its job: where the per-job targets lie in the control block, how long a wait
// lasts and where it reports.
struct SceneWait {
    uint32_t o_targets;          // byte offset of uint32_t targets[n_jobs] in the frame's control block
    uint32_t timeout_ticks;
    uint32_t* err;
};

// wait: timeout_ticks, err and tag of the frame's FrameSync (the rest is filled in here); skin: null, or the meshes the launch skins itself
hipError_t launch_pose_frame(const PoseFrameDev& f, const RigDev& rig, int mode, hipStream_t s, const CtrlInline& inl, uint32_t* counter, uint32_t* counter_total,
                             const FrameSync& wait, const FrameSkin* skin = nullptr, bool exact = true);
// LDS a skinning workgroup of the one-launch frame needs behind the update's: a status word, the palette rows, a flag per wave
__host__ __device__ inline size_t frame_skin_lds(uint32_t max_bones) { return 16u + (size_t)max_bones * 64u + 64u; }

// Animation::update_root_motion for every ticked animation that has settings (after pose_sample:
// rewrites the root node's pose record), then the per-instance root-motion program (machine mode).
hipError_t launch_root_motion(const PoseFrameDev& f, bool run_program, hipStream_t s, const CtrlInline* inl = nullptr);
// Property slots: sample the Real tracks of every ticked animation / run the instance's fold program on them
hipError_t launch_property_sample(const PoseFrameDev& f, hipStream_t s, const CtrlInline* inl = nullptr);
hipError_t launch_property_update(const PoseFrameDev& f, hipStream_t s, const CtrlInline* inl = nullptr);
// out[inst][k] = (has(inst, slots[k]) ? value(inst, slots[k]) : defaults[k]) / 100  (mesh/mod.rs:794-798)
hipError_t launch_blend_shape_weights(const PropRec* prop_out, uint32_t n_prop_slots, uint32_t n_instances,
                                      const int32_t* d_slots, const float* d_defaults, uint32_t n_shapes, float* d_out,
                                      hipStream_t s);
// out[inst][b] = global[inst][bone_nodes[b]] * inv_bind[bone_nodes[b]] (identity for a negative node)
hipError_t launch_palette_gather(const float* d_global, const float* d_inv_bind, const int32_t* d_bone_nodes,
                                 uint32_t n_nodes, uint32_t n_bones, uint32_t n_instances, float* d_out,
                                 hipStream_t s);

}  // namespace fyx
