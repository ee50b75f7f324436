/*
 * fyrox_hip.h -- C ABI of libfyrox_hip.so: the MI355X (gfx950) implementation of Fyrox's
 * per-frame skeletal-animation hot path (pose -> palette -> 4-weight linear-blend skinning).
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.  Fyrox has no
 * FFI seam on this path today (its only dynamic boundary is the Rust-ABI game-plugin dylib,
 * fyrox-impl/src/plugin/dylib.rs:52-76), so each entry point below names the Rust item whose
 * body a maintainer replaces with a call to it (file:line relative to the Fyrox repo root;
 * the Rust `extern "C"` block and the patched call sites are in INTEGRATION.md).
 *
 * Conventions
 *   - every function returns 0 (FYX_OK) or a negative fyx_status; nothing aborts, throws or
 *     unwinds across the boundary; fyx_last_error(ctx) gives the message of the last failure.
 *   - matrices: nalgebra layout, column-major float[16] (m[col*4+row]) -- exactly the bytes of
 *     a Rust `Matrix4<f32>`; a palette is `&[Matrix4<f32>]` reinterpreted as float*.
 *   - quaternions: nalgebra storage order (i, j, k, w).
 *   - "host" pointers are ordinary process memory; "device" pointers are HBM addresses valid
 *     on the context's GPU (from fyx_malloc, or any HIP allocation of the same process).
 *   - a fyx_ctx is single-threaded (`!Sync`): the reference drives this path from its one
 *     update/render thread (fyrox-impl/src/engine/mod.rs:1634-1733).
 *   - *_device calls are asynchronous on the context's stream; host variants synchronise.
 */
#ifndef FYROX_HIP_H
#define FYROX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fyx_ctx fyx_ctx;

typedef enum fyx_status {
    FYX_OK = 0,
    FYX_ERR_INVALID_ARG = -1,   /* null pointer, zero stride, attribute offset outside the vertex */
    FYX_ERR_NO_DEVICE = -2,     /* no gfx950 device / bad ordinal */
    FYX_ERR_HIP = -3,           /* a HIP runtime call failed (message has the hipError string) */
    FYX_ERR_OOM = -4,           /* device allocation failed */
    FYX_ERR_UNKNOWN_ID = -5,    /* mesh / clip / skeleton id not registered */
    FYX_ERR_BONE_INDEX = -6,    /* a vertex references bone >= n_bones (Rust: slice-index panic,
                                   scene/mesh/mod.rs:514) */
    FYX_ERR_MISSING_ATTRIBUTE = -7, /* attribute required by the call is absent in the mesh
                                   (Rust: VertexFetchError::NoSuchAttribute, buffer.rs:1279) */
    FYX_ERR_UNSUPPORTED = -8
} fyx_status;

/* ---- context ------------------------------------------------------------------------- */

/* One context per engine thread per GPU.  Owns a HIP stream and all device memory keyed by ids. */
int fyx_init(fyx_ctx** out_ctx, int device_ordinal);
void fyx_shutdown(fyx_ctx* ctx);
const char* fyx_last_error(const fyx_ctx* ctx);   /* never NULL; valid until the next call */
const char* fyx_version(void);
/* Borrow an externally owned hipStream_t (e.g. the renderer's); NULL restores the own stream. */
int fyx_set_stream(fyx_ctx* ctx, void* hip_stream);
void* fyx_get_stream(fyx_ctx* ctx);
int fyx_sync(fyx_ctx* ctx);
/* Stream semantics.  Everything is ordered on the context stream EXCEPT fyx_lbs_skin_device /
 * fyx_lbs_skin_streams launches: with option "lbs.streams" = K > 1 (default 2) those are dealt
 * round-robin onto K internal worker streams so that the head of one launch overlaps the tail of
 * the previous one (independent meshes / frames).  Each such launch is ordered AFTER everything
 * enqueued on the context stream before the call, but NOT with respect to other skinning
 * launches: two launches that touch the same output buffer need a fyx_join() between them.
 * fyx_join makes the context stream wait (GPU-side, no host block) for every in-flight launch;
 * every other entry point that uses the context stream (uploads, copies, fyx_sync, fyx_timer_*,
 * host-variant calls, fyx_palette*) joins implicitly first.  A caller that borrowed a stream
 * with fyx_set_stream must call fyx_join before consuming skinned output on that stream. */
int fyx_join(fyx_ctx* ctx);
/* Options.  Unknown keys return FYX_ERR_INVALID_ARG.
 *   skinning:
 *     "lbs.exact"        1 (default) = the reference's operation order, no FMA contraction: bit-identical to the CPU path
 *                        (subnormal numbers kept, overflow to +-inf, NaN where the CPU has NaN -- WHICH NaN, its sign and payload,
 *                        is specified neither by IEEE 754 nor by Rust: x86 makes 0xffc00000 of 0 / 0, this device 0x7fc00000);
 *                        0 = fused multiply-adds, within 1e-5 relative (north_star's tolerance); the crowd kernel then blends
 *                        the four matrices first and transforms once (the same linear map, different rounding)
 *     "lbs.streams"      1..4 worker streams for independent skinning launches, see fyx_join (default 2)
 *     "lbs.blocks_per_cu" persistent grid of lbs_skin (and of the lbs.dyn = 0 batch) = CUs x this (default 4)
 *     "lbs.dyn"          single-instance launches from 512 K vertices on, and every fyx_lbs_skin_batch launch: 1 (default) = the
 *                        kernels whose waves draw their 64-vertex units from a per-workgroup ticket counter (lbs_skin_dyn,
 *                        lbs_skin_batch_dyn); 0 = lbs_skin's fixed deal.  Same bits.
 *     "lbs.crowd"        instanced launches: -1 (default) = the crowd kernel from 4 instances on, 0 never, 1 always (it keeps
 *                        a tile of 512 vertices in registers and loops over instances)
 *     "lbs.crowd_ipb"    instances per workgroup run of the crowd kernel, 0 = auto
 *     "lbs.crowd_lean"   1 = the crowd kernel in its register-lean form at two workgroups per CU: ~6 % slower alone, but it
 *                        leaves register room for the next frame's pose kernels to run beside it -- use with "anim.overlap"
 *     "lbs.timing"       per-launch events, see fyx_debug_kernel_time
 *   pose path:
 *     "anim.threads" / "anim.split"  host threads that plan a crowd's frame and instances per planning task; crowds below
 *                        2 x anim.split stay on the calling thread
 *     "anim.sample_form" 0 auto, 1 curves of one instance on the lanes, 2 instances of one curve on the lanes -- same
 *                        results, the crowd form is picked from 32 instances on
 *     "anim.overlap"     0 (default) = a frame is one dependent chain on the context stream.
 *                        1 = whole frames alternate between TWO streams: a pose update (fyx_*_update, fyx_scene_update) starts a
 *                        frame on the other stream, the skinning launches that follow go there too, in order behind it.  Frame
 *                        n + 1's pose kernels so run beside frame n's skinning; its only cross-stream edge is "behind frame n's
 *                        pose update".  The caller alternates two palette buffers per animator: a pose update must not be given
 *                        a palette buffer that a skinning launch issued since the previous pose update reads (INTEGRATION.md;
 *                        fyx_animator_set_palette_output_pair registers both once).  The skinning launches of two consecutive
 *                        frames are NOT ordered against each other: a caller that skins itself alternates its vertex outputs
 *                        as well (what a renderer that draws frame n while frame n + 1 is skinned does anyway).
 *                        "lbs.streams" is not used then.  C3: frame 0.113 -> 0.094 - 0.101 ms (a scene of 256 small characters gains nothing
 *                        since its skinning is one round of resident workgroups: 0.051 ms either way).  (A second pipelined form -- streams by kind -- measured the same for scenes and the one-stream time
 *                        for crowds; it is "debug.overlap" = 2, for experiments: DESIGN.md section 4.)
 *     "anim.update_lean" 1 (default) = a frame whose fold programs are ALL straight (a few clips blended in a row: the common
 *                        machines; the host classifies with the kernel's own function) runs the update kernel built without the
 *                        fold interpreter: a third of the registers, so its waves fit beside a running skinning kernel
 *     "anim.one_launch"  1 (default) = an animator whose control block fits the kernel arguments (one character, a few instances)
 *                        and that tracks neither root motion nor property values runs its sampler and its update as ONE launch:
 *                        the update's workgroups lie behind the sampler's in the same grid and wait on a device counter the
 *                        sampler's workgroups add to.  0 = two launches.  Same results bit for bit.
 *                        ASSUMPTION, and what happens when it fails: a waiting workgroup must never hold the place a workgroup it
 *                        waits for needs.  The launch is sized so that its WHOLE grid is resident at once on an otherwise idle
 *                        MI355X (then no order of dispatch can starve a sampler); beside other kernels it relies on the
 *                        dispatcher handing out a grid's workgroups in index order (samplers first), which gfx9 hardware does
 *                        and HIP does not promise.  A wait is therefore BOUNDED ("anim.wait_timeout_ms"): a workgroup that
 *                        gives up writes a report and that launch computes NOTHING (never a frame made of stale records).  The next
 *                        fyx_sync or pose update sees the report, sets "anim.one_launch" = 0 for the context (separate launches, no
 *                        in-grid wait) and RUNS THE FRAME AGAIN that way before it returns: no frame is lost, the call returns
 *                        FYX_OK, fyx_last_error holds a warning naming the animator, the counter and its target, and
 *                        fyx_get_option("debug.frames_reissued") counts such frames (fyx_get_option("anim.one_launch") reads 0).
 *     "anim.frame_skin"  1 (default) = a one-launch frame also holds the workgroups that skin the animator's skin outputs
 *                        (fyx_animator_set_skin_output), and the update stage of a SMALL scene (fyx_scene_update, up to ~260 k skinned
 *                        vertices) those of its animators; 0 = the update call issues the skinning launches behind the pose launch(es).
 *                        (Two further forms measured slower on large scenes are "debug.frame_skin" = 2 / 3, for experiments.)
 *                        "anim.frame_skin_units": 64-vertex units per wave of those workgroups, 0 (default) = the smallest depth
 *                        that gives every skinning workgroup a CU of its own (C2: 1, C5: 2)
 *     "anim.wait_timeout_ms" 1 .. 30000 (default 500): how long an in-grid wait of a one-launch frame lasts before it reports
 *     "anim.update_pack" 4 (default), 2 or 0: a crowd (>= 64 instances) of a rig of at most 64 nodes runs that lean update kernel
 *                        with this many instances per workgroup (one wave each, nothing shared): beside a crowd's skinning the
 *                        four waves take ONE of the places a skinning workgroup leaves instead of up to four (C3 frame 0.0979
 *                        against 0.0995 ms, the same bits); 0 = one workgroup per instance
 *     "anim.ctrl_upload" how a frame's control block reaches the GPU when it does not fit the kernel arguments: 0 = a copy on
 *                        an upload stream of its own + two events, 1 = a copy on the consuming stream, 2 (default) = a copy
 *                        kernel on the consuming stream that reads the pinned block (no copy command, no event, no second stream;
 *                        C3: the update call costs the host 46 us instead of 53, the frame 0.115 instead of 0.116 - 0.120 ms)
 *     "anim.inline_ctrl" 1 (default) = a frame's control block -- sample times, tick flags, fold program -- of at most 1 KB, i.e.
 *                        one character or a handful of instances, travels inside the kernel arguments of the pose kernels
 *                        instead of through a device block and an H2D copy on the upload stream; 0 = always the copy.  Same
 *                        kernels, same results; a single character's frame 0.031 -> 0.018 ms
 *   streams:
 *     "streams.priority" 1 = the context's own stream (pose path) is created with the highest stream priority, the launch streams
 *                        (skinning) with the lowest.  Default 0: measured on MI355X / ROCm 7.0, kernels of a high-priority
 *                        stream run ~3 x slower (the crowd sampler 52 us instead of 17) and the calls that order streams of
 *                        different priority cost the host ~150 us per frame (profiles/r04_frame_study/)
 *     "streams.pose_cus" N > 0 (multiple of 8) = the own stream may use N CUs only and the launch streams the other 256 - N
 *                        (hipExtStreamCreateWithCUMask); 0 (default) = no masks.  Changing either re-creates the streams.
 *   exchange:
 *     "comm.form"        see fyx_allgather_skinned
 *   measurement:
 *     "debug.timeline"   see fyx_debug_timeline
 *     "debug.host_times" see fyx_debug_host_times
 * The kernel-variant switches of rounds 1 - 2 (workgroup sizes, prefetch depth, cache policy, work distribution, timeline
 * probe) were experiments; their results are in DESIGN.md 5 and the code in the history (tools/exp/README.md). */
int fyx_set_option(fyx_ctx* ctx, const char* key, int value);
/* Measurement aid (option "lbs.timing" = 1): every fyx_lbs_skin_device launch carries its own start / stop events
 * (hipExtLaunchKernel: the dispatch's begin / end timestamps, i.e. the kernel's own duration as a kernel trace
 * reports it, without the gap between dependent launches).  Waits for the work in flight, returns the sum of the
 * durations of the launches made since the last call and their number, and starts over.  At most 8192 launches
 * between two calls.  Replaces nothing in the reference. */
int fyx_debug_kernel_time(fyx_ctx* ctx, double* total_us, uint32_t* n_launches);
/* Measurement aid (option "debug.timeline" = 1): the pose_sample / pose_update launches of fyx_*_update and every
 * fyx_lbs_skin_device launch carry their own start / stop events.  Waits for the work in flight, writes up to `capacity`
 * records {kind: 0 skinning, 1 pose_sample, 2 pose_update, 3 control-block copy kernel; start / stop in microseconds after the first record's start} in
 * launch order, returns their number and starts over -- which kernels of a pipelined frame really ran beside which.  (A frame that
 * runs as ONE launch, option "anim.one_launch", has no record of kind 1: its sampler is inside the kind-2 launch.)
 * At most 16384 launches between two calls.  Replaces nothing in the reference. */
int fyx_debug_timeline(fyx_ctx* ctx, int32_t* kinds, double* start_us, double* stop_us, uint32_t capacity, uint32_t* n_records);
/* Measurement aid (option "debug.host_times" = 1): what fyx_scene_update's sections cost the CALLING THREAD, summed in microseconds
 * since the last call: [0] control plane (every animator's frame planned), [1] device state, launch plans and the job array,
 * [2] control block written and its upload enqueued, [3] the stages' launches, [4] event records behind them, [5] the skin
 * outputs' batched launch (list, cached plan, launch), [6] number of frames, [7] number of those that were steady frames of an unchanged
 * scene (launch plans, job array and the control block's programs kept: only clocks and tick flags written).  Up to 8 values; starts over. */
int fyx_debug_host_times(fyx_ctx* ctx, double* out_us, uint32_t capacity);
int fyx_get_option(fyx_ctx* ctx, const char* key, int* value);

/* GPU-side timing on the context's stream (hipEvent pair): begin records an event, end records
 * a second one, waits for it and returns the elapsed milliseconds between the two. */
int fyx_timer_begin(fyx_ctx* ctx);
int fyx_timer_end(fyx_ctx* ctx, float* out_ms);

/* ---- device memory (for callers without their own HIP allocator) ---------------------- */
int fyx_malloc(fyx_ctx* ctx, size_t bytes, void** out_device_ptr);
int fyx_free(fyx_ctx* ctx, void* device_ptr);
/* The OUTPUT streams of skinning launches (position / normal / tangent of every buffer set a renderer rotates through), allocated the way
 * that measured fastest: `n` allocations of `bytes[i]` bytes, EACH ONE ITS OWN device allocation -- never ranges carved out of one block
 * (C3's three output streams inside one 499 MB block: 77 - 80 us per launch at every spacing tried, as three allocations 61 - 63 us;
 * profiles/r05_placement_pool/).  out_device_ptrs[i] = NULL for bytes[i] = 0; on failure nothing stays allocated.  Free each with
 * fyx_free.  (The reference allocates a mesh's buffers one by one too: scene/mesh/buffer.rs:404-415.) */
int fyx_malloc_streams(fyx_ctx* ctx, uint32_t n, const size_t* bytes, void** out_device_ptrs);
int fyx_memcpy_h2d(fyx_ctx* ctx, void* dst_device, const void* src_host, size_t bytes);
int fyx_memcpy_d2h(fyx_ctx* ctx, void* dst_host, const void* src_device, size_t bytes);

/* ---- mesh registry -------------------------------------------------------------------- */

/* Upload one SurfaceData vertex buffer (AoS bytes + attribute byte offsets; -1 = absent) and
 * de-interleave it on the GPU into the SoA streams the skinning kernel reads.
 * Replaces nothing in the reference -- it is the once-per-modification step of the new
 * `SurfaceData::skin_into` (next to scene/mesh/surface.rs:265), keyed by
 * `SurfaceResource::key()` (surface.rs:1332) and re-run when
 * `VertexBuffer::modifications_count()` (scene/mesh/buffer.rs:909) changes.
 * Attribute formats are AnimatedVertex's (scene/mesh/vertex.rs:139-155): position f32x3,
 * normal f32x3, tangent f32x4, bone weights f32x4, bone indices u8x4.
 * off_pos, off_weights, off_indices are required; normal/tangent optional. */
int fyx_mesh_upload(fyx_ctx* ctx, uint64_t mesh_id, const uint8_t* aos, uint32_t n_verts,
                    uint32_t stride, int off_pos, int off_normal, int off_tangent,
                    int off_weights, int off_indices);
/* Same, from host SoA arrays (pos 3N, normal 3N or NULL, tangent 4N or NULL, weights 4N,
 * indices 4N u8). */
int fyx_mesh_upload_soa(fyx_ctx* ctx, uint64_t mesh_id, uint32_t n_verts, const float* pos,
                        const float* normal, const float* tangent, const float* weights,
                        const uint8_t* indices);
int fyx_mesh_free(fyx_ctx* ctx, uint64_t mesh_id);
/* n_verts, largest bone index referenced (+1 = minimum legal palette length), attribute mask
 * (bit0 normal, bit1 tangent). Any out pointer may be NULL. */
int fyx_mesh_info(fyx_ctx* ctx, uint64_t mesh_id, uint32_t* n_verts, uint32_t* max_bone_index,
                  uint32_t* attr_mask);
/* Device addresses of the SoA streams of a registered mesh (NULL where absent). */
int fyx_mesh_streams(fyx_ctx* ctx, uint64_t mesh_id, const float** d_pos, const float** d_normal,
                     const float** d_tangent, const float** d_weights, const uint32_t** d_indices);

/* ---- linear-blend skinning ------------------------------------------------------------ */

/* Skin `n_instances` copies of mesh `mesh_id`, instance i using
 * palette[i*n_bones .. (i+1)*n_bones).  Host palette in, host vertices out; synchronous.
 *   position: out = sum_k (M[idx_k].transform_point(p)) * w_k      scene/mesh/mod.rs:501-522
 *   normal / tangent.xyz: out = sum_k (mat3(M[idx_k]) * v) * w_k   standard.shader:192-200
 *   tangent.w: copied through.
 * out_* are packed per instance: out_pos[(i*n_verts+v)*3], out_normal likewise, out_tangent
 * [(i*n_verts+v)*4]; each may be NULL to skip that stream.  out_aabb (6 floats: min xyz,
 * max xyz over all instances' skinned positions; NULL to skip) is what
 * `Mesh::accurate_world_bounding_box` (scene/mesh/mod.rs:470-526) returns for the surface. */
int fyx_lbs_skin(fyx_ctx* ctx, uint64_t mesh_id, const float* palette, uint32_t n_bones,
                 uint32_t n_instances, float* out_pos, float* out_normal, float* out_tangent,
                 float* out_aabb);
/* Same with device-resident palette and outputs; asynchronous on the context stream. */
int fyx_lbs_skin_device(fyx_ctx* ctx, uint64_t mesh_id, const float* d_palette, uint32_t n_bones,
                        uint32_t n_instances, float* d_out_pos, float* d_out_normal,
                        float* d_out_tangent);

/* Batch form for a scene of many skinned meshes (Engine::render collects every Mesh's surfaces per frame,
 * scene/mesh/mod.rs:774-802): jobs[k] means fyx_lbs_skin_device(ctx, mesh_id, d_palette, n_bones, n_instances,
 * d_out_pos, d_out_normal, d_out_tangent) with identical results, but all jobs are skinned by ONE kernel launch
 * whose workgroups split the whole batch's vertices evenly (jobs of 16 or more instances keep their own crowd launch).
 * Every job is validated before anything is launched; asynchronous, like the single form.  The job table is kept on
 * the device and only re-sent when it differs from the previous call's. */
typedef struct fyx_skin_job {
    uint64_t mesh_id;
    const float* d_palette;      /* n_instances * n_bones column-major mat4 */
    uint32_t n_bones;
    uint32_t n_instances;
    float* d_out_pos;            /* n_instances * n_verts * 3, or NULL */
    float* d_out_normal;         /* n_instances * n_verts * 3, or NULL */
    float* d_out_tangent;        /* n_instances * n_verts * 4, or NULL */
} fyx_skin_job;
int fyx_lbs_skin_batch(fyx_ctx* ctx, const fyx_skin_job* jobs, uint32_t n_jobs);
/* Blend shapes (morph targets).  `storage` = the bytes of BlendShapesContainer::blend_shape_storage
 * (fyrox-impl/src/scene/mesh/surface.rs:116-217): an RGB16F volume of width*3 x height x n_shapes
 * texels, i.e. per shape `plane_vertices` (= width * height >= n_verts) records of three f16 triples
 * {position, normal, tangent offset}, vertex v at record v -- exactly what S_FetchBlendShapeOffsets
 * (fyrox-graphics-gl/src/shaders/shared.glsl:371-378) reads.  Uploaded once per SurfaceData change
 * (host pointer); n_shapes = 0 removes them.  At most FYX_MAX_BLEND_SHAPES
 * (ShaderDefinition::MAX_BLEND_SHAPE_WEIGHT_GROUPS * 4, fyrox-material/src/shader/mod.rs:616). */
#define FYX_MAX_BLEND_SHAPES 128
int fyx_mesh_set_blend_shapes(fyx_ctx* ctx, uint64_t mesh_id, uint32_t n_shapes, const uint16_t* storage,
                              uint32_t plane_vertices);

/* Skinning with everything the standard shader's vertex stage does before the world transform
 * (fyrox-material/src/shader/standard/opengl/standard.shader:157-200): blend-shape offsets are
 * added to position / normal / tangent.xyz first (for i in 0..n: v += offset_i * weight_i, i
 * ascending, unfused under lbs.exact=1), then the four-influence blend.  All pointers device.
 *   d_blend_shape_weights: [n_instances][n_blend_shapes], the values SurfaceInstanceData carries
 *     (BlendShape::weight / 100, scene/mesh/mod.rs:794-798); n_blend_shapes must be 0 (skip) or the
 *     mesh's shape count.
 *   Outputs: either the SoA streams of fyx_lbs_skin_device, or ONE interleaved vertex buffer
 *     d_out_vertices [n_instances][n_verts][stride], in one of two forms:
 *     - out_stride == 0: the mesh's OWN vertex layout (the one given to fyx_mesh_upload, which
 *       keeps the VertexBuffer bytes resident).  Every output vertex is the input vertex with
 *       position / normal / tangent.xyz replaced -- texture coordinates, tangent.w, bone weights
 *       and indices pass through -- i.e. a complete render-ready vertex buffer
 *       (scene/mesh/buffer.rs:404-415; consumer: renderer/cache/geometry.rs:84-93).  This form
 *       moves whole 64-vertex spans (2 * stride bytes of HBM traffic per vertex) and is the fast
 *       one; stride <= 160.  out_off_* are ignored.
 *     - out_stride > 0: any other layout.  position (12 B), normal (12 B) and tangent (xyzw, 16 B,
 *       w passed through) go to their byte offsets (< 0: not written; stride and offsets multiples
 *       of 4); all other bytes of the buffer are left untouched.  Up to a stride of 160 bytes a
 *       64-vertex span is laid out on chip and written as consecutive dwords with the untouched ones
 *       masked off (1 M vertices into a 68-byte layout: 34 us; every lane storing its own pieces,
 *       which wider strides still do, took 77). */
typedef struct fyx_skin_desc {
    const float* d_palette;
    uint32_t n_bones;
    uint32_t n_instances;
    const float* d_blend_shape_weights;
    uint32_t n_blend_shapes;
    float* d_out_pos;
    float* d_out_normal;
    float* d_out_tangent;
    uint8_t* d_out_vertices;
    uint32_t out_stride;
    int32_t out_off_pos, out_off_normal, out_off_tangent;
} fyx_skin_desc;
int fyx_lbs_skin_ex(fyx_ctx* ctx, uint64_t mesh_id, const fyx_skin_desc* desc);
/* Batch form of fyx_lbs_skin_ex, as fyx_lbs_skin_batch is of fyx_lbs_skin_device: job k = fyx_lbs_skin_ex(ctx,
 * mesh_ids[k], &descs[k]), identical results.  Jobs that write vertex buffers in the mesh's own layout (out_stride 0,
 * with or without blend shapes) are skinned by one launch per layout class; plain SoA jobs by one launch per output
 * set; the remaining forms (blend shapes into SoA, custom interleaved layouts) keep one launch each. */
int fyx_lbs_skin_ex_batch(fyx_ctx* ctx, const uint64_t* mesh_ids, const fyx_skin_desc* descs, uint32_t n_jobs);

/* Raw-stream form (no registry): all pointers device; d_indices is 4 x u8 per vertex. No bone
 * index validation (caller guarantees indices < n_bones). Asynchronous. */
int fyx_lbs_skin_streams(fyx_ctx* ctx, uint32_t n_verts, const float* d_pos, const float* d_normal,
                         const float* d_tangent, const float* d_weights, const uint32_t* d_indices,
                         const float* d_palette, uint32_t n_bones, uint32_t n_instances,
                         float* d_out_pos, float* d_out_normal, float* d_out_tangent);
/* Bounding box only (positions are skinned in registers and never stored).
 * Replaces the body of Mesh::accurate_world_bounding_box, scene/mesh/mod.rs:470-526. */
int fyx_skinned_aabb(fyx_ctx* ctx, uint64_t mesh_id, const float* palette, uint32_t n_bones,
                     float out_aabb[6]);
/* The same box for EVERY instance of an instanced mesh, device to device: d_palette holds n_instances palettes
 * (n_bones column-major mat4 each, e.g. written by fyx_animator_set_palette_output), d_out_aabb receives
 * n_instances x {min xyz, max xyz}.  One launch (two when an instance's vertices are cut in slices) on the context
 * stream, asynchronous, nothing visits the host: what a culling pass over a crowd needs of
 * Mesh::accurate_world_bounding_box (scene/mesh/mod.rs:470-526), which the reference would call once per instance.
 * An empty mesh gives the reference's default box (+MAX, -MAX) (fyrox-math/src/aabb.rs:33-40).  At most 65535 instances. */
int fyx_skinned_aabb_device(fyx_ctx* ctx, uint64_t mesh_id, const float* d_palette, uint32_t n_bones,
                            uint32_t n_instances, float* d_out_aabb);

/* ---- calibration ---------------------------------------------------------------------- */

/* A pure HBM stream with the skinning kernel's read/write mix and access width and no arithmetic:
 * reads 48*units bytes from d_src and writes 32*units bytes to d_dst (units = 1 250 000 moves the
 * same 100 MB as one 1 M-vertex skinning launch).  Used to measure the achievable ceiling next to
 * the skinning kernel and to calibrate rocprofv3 FETCH_SIZE/WRITE_SIZE on a known byte count. */
int fyx_calib_stream_copy(fyx_ctx* ctx, const float* d_src, float* d_dst, uint32_t units);

/* ---- palette -------------------------------------------------------------------------- */

/* out[i] = global[i] * inv_bind[i]  (scene/mesh/mod.rs:781-793).  Host in/out, synchronous. */
int fyx_palette(fyx_ctx* ctx, const float* global, const float* inv_bind, uint32_t n, float* out);
int fyx_palette_device(fyx_ctx* ctx, const float* d_global, const float* d_inv_bind, uint32_t n,
                       float* d_out);

/* ====================================================================================== */
/* Pose path: keyframes -> animation poses -> blending state machine -> node transforms    */
/* -> global matrices -> bone palettes, for a batch of instances (a crowd) per call.         */
/*                                                                                          */
/* Split of work.  The per-instance CONTROL plane (time advance, transitions, parameters,   */
/* blend weights: a few scalars per instance) runs on the host inside this library and       */
/* mirrors the reference's Animation / Machine code line by line; it emits, per frame and    */
/* instance, the sample times and a small fold program.  Every per-BONE operation (curve     */
/* sampling, pose blending, apply, local matrix, hierarchy, palette) runs in HIP kernels;    */
/* poses, node transforms, matrices and palettes never leave HBM.                            */
/* ====================================================================================== */

#define FYX_ALL_INSTANCES 0xffffffffu

/* ValueBinding (fyrox-animation/src/value.rs:355-373). */
enum { FYX_BIND_POSITION = 0, FYX_BIND_SCALE = 1, FYX_BIND_ROTATION = 2,
       /* ValueBinding::Property{name, value_type}: FYX_BIND_PROPERTY0 + id, the id standing for the name
        * (bindings compare by name and type, value.rs:355-373).  Tracks of every TrackValueKind. */
       FYX_BIND_PROPERTY0 = 3 };
/* TrackValueKind (container.rs:40-62) */
enum { FYX_KIND_REAL = 0, FYX_KIND_VEC2 = 1, FYX_KIND_VEC3 = 2, FYX_KIND_VEC4 = 3,
       FYX_KIND_QUAT_EULER = 4, FYX_KIND_QUAT = 5 };
/* CurveKeyKind (fyrox-math/src/curve.rs:34-45) */
enum { FYX_KEY_CONSTANT = 0, FYX_KEY_LINEAR = 1, FYX_KEY_CUBIC = 2 };

/* One Track (track.rs:103-107): binding + TrackDataContainer{kind, curves}.  The keys of its
 * curves follow each other in the key arrays passed to fyx_tracks_data_upload. */
typedef struct fyx_track_desc {
    int32_t binding;          /* FYX_BIND_* */
    int32_t kind;             /* FYX_KIND_*; Position/Scale need VEC3, Rotation QUAT or QUAT_EULER */
    uint32_t n_curves;        /* curves.len(), 0..4; fewer than the kind needs => the track fetches None */
    uint32_t curve_n_keys[4]; /* keys per curve, sorted by location as Curve keeps them */
} fyx_track_desc;

/* AnimationTracksData (fyrox-animation/src/lib.rs:66-110): uploaded once, shared by any number
 * of animations.  Key arrays hold sum(curve_n_keys) entries in track-major, curve-major order.
 * Tangents are read only for FYX_KEY_CUBIC keys (CurveKeyKind::Cubic{left_tangent,right_tangent}).
 * Key locations must be finite and non-decreasing within each curve, as Curve keeps them (Curve::from and add_key sort,
 * curve.rs:176-236): value_at's search, its span hints and the end clamps rely on the order.  Unsorted or non-finite
 * locations are refused with FYX_ERR_INVALID_ARG (they are not sorted here: a caller that sends them has bypassed Curve). */
int fyx_tracks_data_upload(fyx_ctx* ctx, uint64_t tracks_id, uint32_t n_tracks,
                           const fyx_track_desc* tracks, uint32_t n_keys,
                           const float* key_location, const float* key_value,
                           const uint8_t* key_kind, const float* key_left_tangent,
                           const float* key_right_tangent);
int fyx_tracks_data_free(fyx_ctx* ctx, uint64_t tracks_id);

/* The fields of scene::transform::Transform that calculate_local_transform reads
 * (fyrox-impl/src/scene/transform.rs:96-120, :421-540).  post_rotation_matrix is the cached
 * Matrix3 (column-major) that set_post_rotation stores. */
typedef struct fyx_transform {
    float local_position[3];
    float local_rotation[4];      /* (i, j, k, w) */
    float local_scale[3];
    float pre_rotation[4];
    float post_rotation_matrix[9];
    float rotation_offset[3];
    float rotation_pivot[3];
    float scaling_offset[3];
    float scaling_pivot[3];
} fyx_transform;

/* A rig: the scene nodes one animated model instance consists of (bones and whatever sits
 * between them), in an order where parent[i] < i (or -1: no parent inside the rig; such a node
 * multiplies by the identity exactly as Graph::update_global_transform_recursively does for an
 * invalid parent handle, scene/graph/mod.rs:1210-1216).  `transforms` are the initial local
 * transforms every instance starts from; inv_bind (n_nodes x 16, NULL = identity) is
 * Base::inv_bind_pose_transform (scene/base.rs:710-712).  At most 1024 nodes. */
int fyx_rig_create(fyx_ctx* ctx, uint64_t rig_id, uint32_t n_nodes, const int32_t* parent,
                   const fyx_transform* transforms, const float* inv_bind);
int fyx_rig_free(fyx_ctx* ctx, uint64_t rig_id);

/* Surface::bones (scene/mesh/surface.rs:1255): the rig nodes whose matrices form a palette,
 * in bone-index order.  A negative entry is an invalid handle (identity matrix). */
int fyx_bone_list_create(fyx_ctx* ctx, uint64_t bones_id, uint64_t rig_id, uint32_t n_bones,
                         const int32_t* bone_nodes);
int fyx_bone_list_free(fyx_ctx* ctx, uint64_t bones_id);

/* An animator: n_instances copies of one rig, each with its own AnimationContainer state
 * (AnimationPlayer, scene/animation/mod.rs:190-346) and optionally its own Machine
 * (AnimationBlendingStateMachine, scene/animation/absm.rs).  Structure (animations, tracks
 * bindings, machine graph) is shared by the instances; state (times, speeds, enabled flags,
 * parameters, active states, transition progress, node transforms) is per instance. */
int fyx_animator_create(fyx_ctx* ctx, uint64_t animator_id, uint64_t rig_id, uint32_t n_instances);
int fyx_animator_free(fyx_ctx* ctx, uint64_t animator_id);

/* AnimationContainer::add + Animation::set_tracks_data + track_bindings (lib.rs:860-870):
 * track_target[t] = rig node the t-th track drives (negative: no TrackBinding for the track),
 * track_enabled[t] = TrackBinding::enabled (NULL: all enabled).  The new animation has the
 * reference's defaults: speed 1, looped, enabled, time 0, time_slice 0..0 (lib.rs:928-950).
 * Several tracks may feed one binding of one node, and a track's kind need not fit its binding, exactly as in the reference: every
 * enabled track's value goes into its node's list in track order (lib.rs:895-914); blends pair each value with the FIRST same-binding
 * value of the other pose (value.rs:438-444, kinds that differ: no-op), and the list is applied in order, the last fitting value
 * winning (scene/animation/mod.rs:147-186).  Such an animator runs its machine on a two-record fold (no one-launch frame, scenes run
 * its members one by one); span hints of tracks that are neither first nor last of their binding are not kept warm. */
int fyx_animator_add_animation(fyx_ctx* ctx, uint64_t animator_id, uint64_t tracks_id,
                               const int32_t* track_target, const uint8_t* track_enabled,
                               uint32_t* out_animation);
/* AnimationContainer::remove (lib.rs:1007): the index stays taken (indices of other animations do not move, as pool
 * handles do not) but no longer resolves: nothing ticks it, IsAnimationEnded conditions on it are true, state actions
 * skip it, per-animation calls answer FYX_ERR_INVALID_ARG -- and a PlayAnimation node that still names it keeps
 * handing out the pose it copied last, exactly as play.rs:93-99 does for a handle that stopped resolving. */
int fyx_animator_remove_animation(fyx_ctx* ctx, uint64_t animator_id, uint32_t animation);
int fyx_animation_set_track_enabled(fyx_ctx* ctx, uint64_t animator_id, uint32_t animation,
                                    uint32_t track, int enabled);
/* Per-instance Animation state; instance = FYX_ALL_INSTANCES addresses every instance.
 * Semantics of lib.rs:432-460 (set_time_position wraps or clamps into the time slice;
 * set_time_slice re-applies it), :695-748. */
int fyx_animation_set_time_slice(fyx_ctx* ctx, uint64_t animator_id, uint32_t animation,
                                 uint32_t instance, float start, float end);
int fyx_animation_set_time_position(fyx_ctx* ctx, uint64_t animator_id, uint32_t animation,
                                    uint32_t instance, float time);
int fyx_animation_set_speed(fyx_ctx* ctx, uint64_t animator_id, uint32_t animation,
                            uint32_t instance, float speed);
int fyx_animation_set_loop(fyx_ctx* ctx, uint64_t animator_id, uint32_t animation,
                           uint32_t instance, int looped);
int fyx_animation_set_enabled(fyx_ctx* ctx, uint64_t animator_id, uint32_t animation,
                              uint32_t instance, int enabled);
int fyx_animation_rewind(fyx_ctx* ctx, uint64_t animator_id, uint32_t animation, uint32_t instance);
int fyx_animation_get_state(fyx_ctx* ctx, uint64_t animator_id, uint32_t animation,
                            uint32_t instance, float* time_position, int* enabled, int* has_ended);

/* AnimationSignal + the animation's event queue (fyrox-animation/src/signal.rs, lib.rs:471-496,
 * :680-700).  A signal is addressed by the index fyx_animation_add_signal returned (the shim keeps
 * the {Uuid, name} pair per index); signals are shared by the instances, event queues and
 * max_event_capacity (default 32) are per instance.  During a tick every enabled signal with
 * time in (t, t + dt*speed] (speed >= 0) resp. [t + dt*speed, t) (speed < 0) pushes its index;
 * as in the reference the capacity cap applies to the negative-speed case only (the `||`/`&&`
 * precedence at lib.rs:478-482). */
int fyx_animation_add_signal(fyx_ctx* ctx, uint64_t animator_id, uint32_t animation, float time,
                             int enabled, uint32_t* out_signal);
int fyx_animation_set_signal_enabled(fyx_ctx* ctx, uint64_t animator_id, uint32_t animation,
                                     uint32_t signal, int enabled);
int fyx_animation_set_max_event_capacity(fyx_ctx* ctx, uint64_t animator_id, uint32_t animation,
                                         uint32_t instance, uint32_t capacity);
/* Animation::pop_event: *out_signal = front of the queue, or -1 when it is empty */
int fyx_animation_pop_event(fyx_ctx* ctx, uint64_t animator_id, uint32_t animation,
                            uint32_t instance, int32_t* out_signal);
int fyx_animation_event_count(fyx_ctx* ctx, uint64_t animator_id, uint32_t animation,
                              uint32_t instance, uint32_t* out_count);
/* Animation::take_events / events_mut().clear() */
int fyx_animation_clear_events(fyx_ctx* ctx, uint64_t animator_id, uint32_t animation, uint32_t instance);

/* Root motion (lib.rs:302-343, :498-676).  Settings are shared by the instances; node < 0 = None.
 * With settings, every tick extracts the root node's frame-to-frame motion into the animation's
 * RootMotion and rewrites the root node's sampled position / rotation to their value at the start
 * of the time slice (per ignore_* flag), on the GPU, before the pose is blended or applied.
 * Setting any settings turns root-motion tracking on for the animator (see below). */
int fyx_animation_set_root_motion_settings(fyx_ctx* ctx, uint64_t animator_id, uint32_t animation,
                                           int32_t node, int ignore_x_movement, int ignore_y_movement,
                                           int ignore_z_movement, int ignore_rotations);
/* Option<RootMotion> as the engine's scripts read it: has == 0 is None (the other fields then
 * hold RootMotion::default(): zero offset, identity rotation). */
typedef struct fyx_root_motion {
    float delta_position[3];
    uint32_t has;
    float delta_rotation[4];      /* (i, j, k, w) */
} fyx_root_motion;
/* Root-motion tracking also maintains AnimationPose::root_motion of every pose node, layer and
 * the machine (pose.rs:54-100: clone_into copies it, blend_with lerps / nlerps it, reset() keeps
 * it -- so a blend node starts each frame from LAST frame's value, which is reproduced).  It can
 * be switched on without any settings (every pose then carries None / default as in the
 * reference); switching it off frees the state. */
int fyx_animator_track_root_motion(fyx_ctx* ctx, uint64_t animator_id, int enabled);
/* Animation::root_motion() of every instance: host_out[n_instances].  Synchronous. */
int fyx_animation_read_root_motion(fyx_ctx* ctx, uint64_t animator_id, uint32_t animation,
                                   fyx_root_motion* host_out);
/* Machine::pose().root_motion() (layer < 0) or MachineLayer's final pose (layer >= 0) of every
 * instance, as of the last fyx_absm_update: host_out[n_instances].  Synchronous. */
int fyx_absm_read_root_motion(fyx_ctx* ctx, uint64_t animator_id, int32_t layer, fyx_root_motion* host_out);

/* Property{..} bindings (value.rs:355-373, applied through reflection at :404-427; the glTF importer animates
 * BlendShape weights this way, resource/gltf/animation.rs:395-420, and the animation editor keys any numeric
 * property).  Tracks of every TrackValueKind are taken: Real, Vector2/3/4, UnitQuaternion, UnitQuaternionEuler.  A track with
 * binding FYX_BIND_PROPERTY0 + id bound to node n animates the (n, id) "slot" of the animator; slots
 * are created by fyx_animator_add_animation in order of first appearance.  Such a value is part of its
 * node's pose exactly as in the reference: it is blended by TrackValue::blend_with (value.rs:221-230: lerpf, vector
 * lerp or nlerp by variant; values of different variants do not blend), dropped when
 * only the other pose holds it, copied -- weight ignored -- when the node's own pose is empty, removed
 * by a layer mask on the node, and it makes the node's pose non-empty for the node's Position /
 * Rotation / Scale values too.  The applied values stay on the device as f32 lanes + variant; the shim reads them
 * back and writes them through reflection -- the numeric cast to the property's machine type (bool, integers, f64,
 * vectors of those: value.rs:232-352) is the shim's, `as` conversions of the lanes -- or feeds them to the
 * skinning kernel directly. */
enum { FYX_VALUE_REAL = 0, FYX_VALUE_VEC2 = 1, FYX_VALUE_VEC3 = 2, FYX_VALUE_VEC4 = 3, FYX_VALUE_QUAT = 4 };  /* TrackValue variants */
typedef struct fyx_property_value {
    float value[4];      /* Real: [0]; Vector2/3/4: xy / xyz / xyzw; UnitQuaternion: i, j, k, w */
    uint32_t present;    /* 0: no value (the other fields are then 0) */
    uint32_t kind;       /* FYX_VALUE_* */
    uint32_t reserved[2];
} fyx_property_value;
int fyx_animator_property_count(fyx_ctx* ctx, uint64_t animator_id, uint32_t* out_count);
/* *out_slot = slot of (node, property id), or -1 when no animation drives it */
int fyx_animator_property_slot(fyx_ctx* ctx, uint64_t animator_id, int32_t node, int32_t property_id,
                               int32_t* out_slot);
/* host_out[n_instances][n_slots].  animation < 0: the values applied so far (present = the property has been
 * written at least once); animation >= 0: that animation's current pose (present = the pose holds the value).
 * Synchronous. */
int fyx_animator_read_properties(fyx_ctx* ctx, uint64_t animator_id, int32_t animation, fyx_property_value* host_out);
/* d_out[n_instances][n_shapes] = (applied value of slots[k], or default_weights[k] when slots[k] < 0 or
 * nothing has been applied yet) / 100 -- the `blend_shapes_weights` Mesh::collect_render_data hands to
 * the renderer (scene/mesh/mod.rs:794-798), ready for fyx_lbs_skin_ex.  slots / default_weights are
 * host arrays of n_shapes entries; asynchronous on the context stream. */
int fyx_animator_blend_shape_weights(fyx_ctx* ctx, uint64_t animator_id, uint32_t n_shapes,
                                     const int32_t* slots, const float* default_weights, float* d_out);

/* ---- Machine (fyrox-animation/src/machine) ------------------------------------------- */
/* Parameter (machine/parameter.rs:37-60) */
enum { FYX_PARAM_WEIGHT = 0, FYX_PARAM_RULE = 1, FYX_PARAM_INDEX = 2, FYX_PARAM_SAMPLING_POINT = 3 };
/* StateAction (machine/state.rs:62-116) */
enum { FYX_ACTION_NONE = 0, FYX_ACTION_REWIND_ANIMATION = 1, FYX_ACTION_ENABLE_ANIMATION = 2,
       FYX_ACTION_DISABLE_ANIMATION = 3, FYX_ACTION_ENABLE_RANDOM_ANIMATION = 4 /* fyx_state_add_random_action */ };
/* LogicNode (machine/transition.rs:107-131), prefix encoded into an int array:
 * PARAMETER p | AND a b | OR a b | XOR a b | NOT a | IS_ANIMATION_ENDED animation */
enum { FYX_LOGIC_PARAMETER = 0, FYX_LOGIC_AND = 1, FYX_LOGIC_OR = 2, FYX_LOGIC_XOR = 3,
       FYX_LOGIC_NOT = 4, FYX_LOGIC_IS_ANIMATION_ENDED = 5 };

/* Parameters are addressed by index instead of name (the shim resolves names once); an index
 * that is out of range, or holds another kind than the reader expects, behaves like a missing
 * or mistyped name in the reference (weight 0.0 / rule false / node yields an empty pose). */
int fyx_machine_add_parameter(fyx_ctx* ctx, uint64_t animator_id, int kind, float f0, float f1,
                              uint32_t u, uint32_t* out_parameter);
int fyx_machine_set_parameter(fyx_ctx* ctx, uint64_t animator_id, uint32_t parameter,
                              uint32_t instance, int kind, float f0, float f1, uint32_t u);
int fyx_machine_add_layer(fyx_ctx* ctx, uint64_t animator_id, float weight, uint32_t* out_layer);
int fyx_layer_set_weight(fyx_ctx* ctx, uint64_t animator_id, uint32_t layer, float weight);
/* LayerMask (machine/mask.rs): rig nodes the layer must not animate */
int fyx_layer_set_mask(fyx_ctx* ctx, uint64_t animator_id, uint32_t layer,
                       const int32_t* excluded_nodes, uint32_t n);
/* PoseNode::PlayAnimation (machine/node/play.rs) */
int fyx_layer_add_play_animation(fyx_ctx* ctx, uint64_t animator_id, uint32_t layer,
                                 uint32_t animation, uint32_t* out_node);
/* PoseNode::BlendAnimations (node/blend.rs:60-164): weight_parameter[i] < 0 = PoseWeight::Constant */
int fyx_layer_add_blend_animations(fyx_ctx* ctx, uint64_t animator_id, uint32_t layer,
                                   uint32_t n_inputs, const int32_t* pose_sources,
                                   const int32_t* weight_parameters, const float* weight_constants,
                                   uint32_t* out_node);
/* PoseNode::BlendAnimationsByIndex (node/blend.rs:200-361) */
int fyx_layer_add_blend_animations_by_index(fyx_ctx* ctx, uint64_t animator_id, uint32_t layer,
                                            int32_t index_parameter, uint32_t n_inputs,
                                            const int32_t* pose_sources, const float* blend_times,
                                            uint32_t* out_node);
/* PoseNode::BlendSpace (node/blendspace.rs); triangles = the Delaunay triangulation the
 * reference caches in BlendSpace::triangles (3 point indices each) */
int fyx_layer_add_blend_space(fyx_ctx* ctx, uint64_t animator_id, uint32_t layer,
                              int32_t sampling_parameter, uint32_t n_points, const float* points_xy,
                              const int32_t* pose_sources, uint32_t n_triangles,
                              const uint32_t* triangles, uint32_t* out_node);
/* MachineLayer::add_state (layer.rs:229-235): the state becomes the ACTIVE one of every instance that has no active
 * state -- the first state of a new layer, but also a state added while a transition is in flight (active_state is NONE
 * then), exactly as in the reference.  It does not become the ENTRY state: only fyx_layer_set_entry_state sets that
 * (layer.rs:209-212: active_state and entry_state of every instance), and fyx_layer_reset returns to it. */
int fyx_layer_add_state(fyx_ctx* ctx, uint64_t animator_id, uint32_t layer, int32_t root_node,
                        uint32_t* out_state);
int fyx_layer_set_entry_state(fyx_ctx* ctx, uint64_t animator_id, uint32_t layer, uint32_t state);
int fyx_state_add_action(fyx_ctx* ctx, uint64_t animator_id, uint32_t layer, uint32_t state,
                         int on_enter, int action, uint32_t animation);
/* StateAction::EnableRandomAnimation(handles) (state.rs:85, :108-114): one of `animations`, chosen uniformly, is
 * enabled (an invalid handle chosen enables nothing; an empty list draws nothing).  The reference draws from
 * rand::thread_rng(), which is not reproducible; this library gives every instance its own splitmix64 stream:
 *     state += 0x9E3779B97F4A7C15; z = state; z = (z ^ z >> 30) * 0xBF58476D1CE4E5B9;
 *     z = (z ^ z >> 27) * 0x94D049BB133111EB; draw = z ^ z >> 31;   index = (draw * n) >> 64
 * so a run can be repeated.  fyx_animator_set_random_seed sets an instance's state to `seed`; with
 * FYX_ALL_INSTANCES instance i gets seed + (i + 1) * 0x9E3779B97F4A7C15 (distinct streams; with seed 0 that is
 * the default). */
int fyx_state_add_random_action(fyx_ctx* ctx, uint64_t animator_id, uint32_t layer, uint32_t state, int on_enter,
                                const uint32_t* animations, uint32_t n_animations);
int fyx_animator_set_random_seed(fyx_ctx* ctx, uint64_t animator_id, uint32_t instance, uint64_t seed);
/* Transition (machine/transition.rs:180-323) */
int fyx_layer_add_transition(fyx_ctx* ctx, uint64_t animator_id, uint32_t layer, uint32_t source,
                             uint32_t dest, float transition_time, const int32_t* condition,
                             uint32_t n_condition, uint32_t* out_transition);
int fyx_layer_get_state(fyx_ctx* ctx, uint64_t animator_id, uint32_t layer, uint32_t instance,
                        int32_t* active_state, int32_t* active_transition);
/* ---- run-time edits of a machine -------------------------------------------------------
 * The engine edits a Machine IN PLACE between two frames -- Machine::add_layer / remove_layer / insert_layer /
 * pop_layer / layers_mut (machine/mod.rs:280-312), MachineLayer::node_mut / nodes_mut / transition_mut /
 * transitions_mut / state_mut / states_mut (layer.rs:412-525), Transition::set_condition (transition.rs:290),
 * BlendSpace::set_points / points_mut / set_sampling_parameter (node/blendspace.rs:246-310), the public fields of the
 * blend nodes -- and the next evaluate_pose sees the edit with every other piece of run-time state untouched.  The
 * builder calls above only append (which already works between frames: the per-instance state grows with the
 * definition).  Every other edit is made by re-sending the definition:
 *     read the run-time state (fyx_layer_get_state, fyx_layer_get_transition_state, fyx_layer_get_node_state,
 *     fyx_machine_get_parameter)  ->  fyx_machine_clear  ->  builder calls for the edited machine  ->  put the state
 *     back (fyx_layer_set_state, ..._set_transition_state, ..._set_node_state), translated through the caller's
 *     handle -> index maps (a state / transition / node whose handle no longer resolves is simply not restored).
 * What the reference keeps in the objects, and therefore what there is to carry over: MachineLayer::active_state /
 * active_transition (layer.rs:103-109), Transition::elapsed_time / blend_factor (transition.rs:188-201),
 * BlendAnimationsByIndex::prev_index / blend_time (node/blend.rs:260-264) and the parameter values.  Animations, their
 * clocks, event queues and sampled poses belong to the AnimationContainer and are not touched by any of this. */
/* Drops the parameters, the layers and the per-instance machine state (pending layer events included: pop them first).
 * The root-motion records of the poses (AnimationPose::root_motion persists from frame to frame: reset() keeps it,
 * pose.rs:125-129) stay on the device and are matched to the re-sent definition BY POSITION -- node n of layer l, layer
 * l's final pose, the machine's -- so they carry over exactly as long as the surviving pose nodes keep their indices
 * (a definition flattened in pool order does, unless a node in the middle of a pool was freed). */
int fyx_machine_clear(fyx_ctx* ctx, uint64_t animator_id);
int fyx_machine_get_parameter(fyx_ctx* ctx, uint64_t animator_id, uint32_t parameter, uint32_t instance, int* kind,
                              float* f0, float* f1, uint32_t* u);
/* Sets MachineLayer::active_state / active_transition (-1 = Handle::NONE); no actions run, no events are queued.
 * instance may be FYX_ALL_INSTANCES. */
int fyx_layer_set_state(fyx_ctx* ctx, uint64_t animator_id, uint32_t layer, uint32_t instance, int32_t active_state,
                        int32_t active_transition);
int fyx_layer_get_transition_state(fyx_ctx* ctx, uint64_t animator_id, uint32_t layer, uint32_t instance,
                                   uint32_t transition, float* elapsed_time, float* blend_factor);
int fyx_layer_set_transition_state(fyx_ctx* ctx, uint64_t animator_id, uint32_t layer, uint32_t instance,
                                   uint32_t transition, float elapsed_time, float blend_factor);
/* BlendAnimationsByIndex's prev_index (has_prev = 0: None) and blend_time; `node` must be such a node. */
int fyx_layer_get_node_state(fyx_ctx* ctx, uint64_t animator_id, uint32_t layer, uint32_t instance, uint32_t node,
                             int* has_prev, uint32_t* prev_index, float* blend_time);
int fyx_layer_set_node_state(fyx_ctx* ctx, uint64_t animator_id, uint32_t layer, uint32_t instance, uint32_t node,
                             int has_prev, uint32_t prev_index, float blend_time);
/* MachineLayer::reset (layer.rs:288-296): every transition's elapsed_time and blend_factor return to 0 and
 * active_state = entry_state (Handle::NONE on a layer whose entry state was never set: add_state does not set it);
 * active_transition is left as it is, as in the reference.  instance may be FYX_ALL_INSTANCES. */
int fyx_layer_reset(fyx_ctx* ctx, uint64_t animator_id, uint32_t layer, uint32_t instance);
/* machine::Event (machine/event.rs:30-51) and MachineLayer::pop_event (layer.rs:284-286); the
 * queue holds at most 2048 events per layer and instance, further ones are dropped
 * (FixedEventQueue, event.rs:53-90).  Handles are indices, -1 = Handle::NONE. */
enum { FYX_EVENT_STATE_ENTER = 0,               /* a = state */
       FYX_EVENT_STATE_LEAVE = 1,               /* a = state */
       FYX_EVENT_ACTIVE_STATE_CHANGED = 2,      /* a = prev, b = new */
       FYX_EVENT_ACTIVE_TRANSITION_CHANGED = 3  /* a = transition or -1 */ };
typedef struct fyx_layer_event { int32_t kind, a, b; } fyx_layer_event;
/* MachineLayer::collect_active_animations_events (layer.rs:308-401): the queued events of the animations behind the
 * active state (or the active transition's states), filtered by AnimationEventCollectionStrategy (node/mod.rs:
 * 177-184; blend nodes pick the source with the largest / smallest weight, Iterator::max_by / min_by tie rules
 * included).  Nothing is removed from the animations' queues.  *n_events = number found (out_events holds at most
 * `capacity`); signal = the index fyx_animation_add_signal returned. */
enum { FYX_EVENTS_ALL = 0, FYX_EVENTS_MAX_WEIGHT = 1, FYX_EVENTS_MIN_WEIGHT = 2 };
typedef struct fyx_animation_event { uint32_t animation; int32_t signal; } fyx_animation_event;
/* AnimationEventsSource: kind 0 Invalid, 1 State{handle}, 2 Transition{handle, source_state, dest_state} */
typedef struct fyx_events_source { int32_t kind, handle, source_state, dest_state; } fyx_events_source;
int fyx_layer_collect_active_animations_events(fyx_ctx* ctx, uint64_t animator_id, uint32_t layer,
                                               uint32_t instance, int strategy,
                                               fyx_animation_event* out_events, uint32_t capacity,
                                               uint32_t* n_events, fyx_events_source* out_source);
/* *out_has = 0 when the queue is empty */
int fyx_layer_pop_event(fyx_ctx* ctx, uint64_t animator_id, uint32_t layer, uint32_t instance,
                        fyx_layer_event* out_event, int* out_has);

/* ---- per frame ----------------------------------------------------------------------- */
/* AnimationPlayer::update = AnimationContainerExt::update_animations (scene/animation/mod.rs:
 * 83-88, 340-346): every enabled animation ticks (pose sampled at its current time, then the
 * time advances, lib.rs:471-496) and its pose is applied to the nodes, in container order. */
int fyx_animation_player_update(fyx_ctx* ctx, uint64_t animator_id, float dt);
/* AnimationBlendingStateMachine::update (absm.rs:311-326) = Machine::evaluate_pose
 * (machine/mod.rs:344-382) + AnimationPoseExt::apply_internal. */
int fyx_absm_update(fyx_ctx* ctx, uint64_t animator_id, float dt);
/* Graph::update ticks EVERY node of a scene (scene/graph/mod.rs:1459-1502 -> update_node :1415-1440: `node.update(ctx)` for every pool slot), i.e.
 * every AnimationPlayer and every AnimationBlendingStateMachine once per frame.  Batch form of the two calls
 * above for a scene of many animators: each listed animator gets fyx_absm_update if it has a machine,
 * fyx_animation_player_update otherwise, with results identical to calling them one by one in list order -- but the
 * host control planes run on the planner threads side by side, all control data travels in one upload, and each
 * stage of the frame is ONE kernel launch over all animators (a scene of 256 distinct characters costs what one
 * costs in launches).  An id may appear once.  Palettes registered with fyx_animator_set_palette_output are
 * written as usual.
 * Failure: arguments (unknown or repeated ids) are checked before anything changes.  After that the host control plane of
 * EVERY listed animator advances by dt (clocks, transitions, event queues, random draws) before the first GPU call, so an
 * error returned from there on -- a HIP error: allocation, upload, launch -- leaves the host state one frame ahead of the
 * device's poses, exactly as a failure inside fyx_absm_update does for one animator; the context's device state is
 * suspect after a HIP error anyway.  The control plane itself cannot fail at this point: its only error (pose nodes
 * nested deeper than the fold interpreter's accumulators) is refused when the node is added (fyx_layer_add_*). */
int fyx_scene_update(fyx_ctx* ctx, const uint64_t* animator_ids, uint32_t n_animators, float dt);
/* All three calls also refresh the local and global matrices of every node of every instance
 * (Transform::matrix + Graph::update_hierarchical_data); this one does only that (after
 * fyx_animator_set_local_trs, or for a rig nothing animates). */
int fyx_animator_update_transforms(fyx_ctx* ctx, uint64_t animator_id);
/* d_out_palette[(i * n_bones + b) * 16 ..] = global(i, bone_b) * inv_bind(bone_b): the
 * `bone_matrices` of Mesh::collect_render_data (scene/mesh/mod.rs:781-793), ready for
 * fyx_lbs_skin_device(.., n_instances).  Asynchronous on the context stream. */
int fyx_animator_palette(fyx_ctx* ctx, uint64_t animator_id, uint64_t bones_id, float* d_out_palette);
/* The same palette, written by the update calls themselves: after fyx_animation_player_update /
 * fyx_absm_update / fyx_animator_update_transforms the buffer holds the frame's palettes (the matrices are
 * multiplied while still on chip; no extra launch).  d_out_palette = NULL unregisters.  At most 4 outputs per
 * animator (one per skinned surface of the model, typically 1); the bone list must stay registered; the buffer
 * must be 16-byte aligned (matrix columns are stored as 16-byte words). */
int fyx_animator_set_palette_output(fyx_ctx* ctx, uint64_t animator_id, uint64_t bones_id, float* d_out_palette);
/* The same registration with TWO buffers, for pipelined frames ("anim.overlap" = 1: whole frames alternate between two streams
 * of the library).  The frames of the first stream write d_out_palette, those of the second d_out_palette_alt -- so frame n + 1's
 * update never writes the palette frame n's skinning still reads, and the caller registers once instead of swapping the pointer
 * every frame (a scene of 256 animators: 256 calls per frame, each of which invalidates the animator's cached launch plans).
 * Skin outputs registered on the bone list (fyx_animator_set_skin_output) skin from the frame's own buffer; a caller that skins
 * itself asks fyx_animator_current_palette.  Without "anim.overlap" only d_out_palette is written.  d_out_palette_alt = NULL is
 * fyx_animator_set_palette_output.  Both 16-byte aligned, distinct. */
int fyx_animator_set_palette_output_pair(fyx_ctx* ctx, uint64_t animator_id, uint64_t bones_id, float* d_out_palette, float* d_out_palette_alt);
/* The buffer of a registered palette output that the animator's most recent update call wrote (or writes: the call is
 * asynchronous) -- for a pair, the one of the frame's stream.  Host-side lookup, no GPU work. */
int fyx_animator_current_palette(fyx_ctx* ctx, uint64_t animator_id, uint64_t bones_id, float** d_palette);
/* The frame goes on to the vertices.  In the engine one character's frame is ONE dependent chain: Machine::evaluate_pose
 * (fyrox-animation/src/machine/mod.rs:344-382) -> Graph::update_hierarchical_data (scene/graph/mod.rs:1199-1241) -> the bone
 * matrices of Mesh::collect_render_data (scene/mesh/mod.rs:781-793) -> the skinning loop (mesh/mod.rs:501-522,
 * standard.shader:157-216).  After this call every fyx_animation_player_update / fyx_absm_update / fyx_scene_update /
 * fyx_animator_update_transforms of the animator ALSO skins mesh `mesh_id` with the palette of `bones_id` -- which must be a palette
 * output of the animator (fyx_animator_set_palette_output: the palette is still written to its buffer) -- into the outputs
 * ([n_instances][n_verts], packed as fyx_lbs_skin_device packs them; NULL = that stream is not wanted), with the results
 * fyx_lbs_skin_device(mesh_id, that palette, n_bones, n_instances, outputs) right behind the update would give, bit for bit
 * (option lbs.exact as there).  For one character (the frames that run as one launch: anim.one_launch) the launch that
 * samples and updates also holds the workgroups that skin: they request their vertices before the pose exists and form the
 * palette on chip -- no second launch, no palette round trip through memory (option anim.frame_skin = 0: separate launches).
 * Crowds, root motion, property tracks, meshes beyond ~450 k vertices: the same skinning launches fyx_lbs_skin_device makes,
 * issued by the update call.  All three outputs NULL removes the entry; removing the palette output removes its skin outputs.
 * At most 4 per animator.  Under anim.overlap the palette buffers alternate as before (the caller's pointer swap, or a pair
 * registered once: fyx_animator_set_palette_output_pair); the vertex outputs are the same buffers every frame, so the library
 * orders frame n + 1's skinning of them behind frame n's (frame n + 1's pose kernels still run beside frame n's skinning).  A mesh
 * that is freed while registered makes the next update fail with FYX_ERR_UNKNOWN_ID. */
int fyx_animator_set_skin_output(fyx_ctx* ctx, uint64_t animator_id, uint64_t bones_id, uint64_t mesh_id, float* d_out_pos,
                                 float* d_out_normal, float* d_out_tangent);

/* Transform::set_position / set_rotation / set_scale of one node for a range of instances
 * (placing the members of a crowd): trs = n_instances x {pos xyz, rot ijkw, scale xyz}. */
int fyx_animator_set_local_trs(fyx_ctx* ctx, uint64_t animator_id, uint32_t node,
                               uint32_t first_instance, uint32_t n_instances, const float* trs);

/* Read back per-instance state (synchronous).  Layouts, all [n_instances][n_nodes][..]:
 * LOCAL_TRS 12 floats {pos xyz, 0, rot ijkw, scale xyz, 0}; LOCAL/GLOBAL_MATRIX 16 floats;
 * ANIMATION_POSE + animation: the animation's current pose, 12 floats {pos xyz, present-bits
 * as u32 (1 Position, 2 Scale, 4 Rotation), rot ijkw, scale xyz, 0}. */
/* Present bits: 8 = the node's pose holds a Property value, 16 = it holds a value whose kind fits no binding (never applied, blends with
 * nothing, but the pose is not empty: pose.rs:41-47).  A node's pose is a LIST (pose.rs:107-121) and may hold several values of one
 * binding; ANIMATION_POSE shows per binding the value the pose APPLIES (the last one whose kind fits, scene/animation/mod.rs:147-186),
 * ANIMATION_BLEND_VIEW + animation the one a blend READS when this pose is the other operand (the first one, value.rs:438-444; bit
 * clear when that one's kind does not fit).  The two are the same record unless tracks share a binding of a node AND the animator has
 * a machine (an AnimationPlayer blends nothing: it keeps the apply view only and answers both selectors with it). */
enum { FYX_READ_LOCAL_TRS = 0, FYX_READ_LOCAL_MATRIX = 1, FYX_READ_GLOBAL_MATRIX = 2,
       FYX_READ_ANIMATION_POSE = 16, FYX_READ_ANIMATION_BLEND_VIEW = 65536 };
int fyx_animator_read(fyx_ctx* ctx, uint64_t animator_id, int what, float* host_out);
/* Device address of the same arrays (what as above), for consumers on the GPU. */
int fyx_animator_device_ptr(fyx_ctx* ctx, uint64_t animator_id, int what, void** out_device_ptr);

/* ---- multi-GPU: the one exchange step of the path --------------------------------------
 * A scene larger than one GPU is sharded by contiguous vertex range (each GPU holds and skins only its
 * slice; the <= 16 KiB palette is replicated); crowds shard by instance range.  Neither needs any
 * communication to skin.  Only a consumer that wants the WHOLE skinned buffer on every GPU pays one
 * all-gather (RCCL over xGMI), which these calls provide without any other framework in the process:
 * one process (or thread) per GPU, each with its own fyx_ctx; rank 0 calls fyx_comm_unique_id and the
 * host application hands the 128 bytes to the other ranks (a file, a pipe, MPI, ...); every rank then
 * calls fyx_comm_init (collective: returns when all n_ranks have joined).  librccl.so is opened on
 * first use; without it these calls return FYX_ERR_UNSUPPORTED and nothing else is affected. */
#define FYX_COMM_ID_BYTES 128
int fyx_comm_unique_id(fyx_ctx* ctx, uint8_t out_id[FYX_COMM_ID_BYTES]);
int fyx_comm_init(fyx_ctx* ctx, const uint8_t id[FYX_COMM_ID_BYTES], int rank, int n_ranks);
int fyx_comm_shutdown(fyx_ctx* ctx);
/* d_recv[r * count .. (r + 1) * count) = rank r's d_send[0 .. count): ncclAllGather on the context stream,
 * ordered after every skinning launch in flight (it starts with the GPU-side join of fyx_join).  Equal
 * counts on all ranks (pad the last shard).  Asynchronous. */
int fyx_allgather_f32(fyx_ctx* ctx, const float* d_send, size_t count, float* d_recv);
/* The cut BASELINE config 4 names ("1 M verts / 256 bones, vertex-range sharded across 8 GPUs"): rank r of n_ranks owns
 * the contiguous vertices [*begin, *end) of an n_verts mesh; boundaries fall on whole 256-vertex groups (four 64-vertex
 * work units), groups are dealt as evenly as integers allow, so shards are RAGGED (1 000 000 over 8 GPUs: 124 928 /
 * 124 928 / 125 184 / ... / 124 992 vertices).  Pure function: no context, no GPU.  The reference has no counterpart
 * (one process, one GPU); this replaces nothing and exists for the all-gather below. */
int fyx_shard_vertex_range(uint32_t n_verts, int rank, int n_ranks, uint32_t* begin, uint32_t* end);
/* The PADDED cut of exchange form 2 (option "comm.form" = 2): every shard is the same *shard_verts = ceil(groups / n_ranks) * 256
 * vertices long, rank r owns [r * shard_verts, min((r + 1) * shard_verts, n_verts)) and every GPU's full streams hold
 * n_ranks * shard_verts vertices (1 000 000 over 8 GPUs: shards of 125 184, buffers of 1 001 472 vertices, the last rank skins
 * 123 712).  Pure function.  FYX_ERR_UNSUPPORTED when n_ranks * shard_verts does not fit 32 bits. */
int fyx_shard_vertex_range_padded(uint32_t n_verts, int rank, int n_ranks, uint32_t* begin, uint32_t* end, uint32_t* shard_verts);
/* rank and size of the context's communicator as RCCL reports them (ncclCommUserRank / ncclCommCount). */
int fyx_comm_info(fyx_ctx* ctx, int* rank, int* n_ranks);
/* The exchange step, once per frame: every non-null d_*_all is a FULL skinned stream of the n_verts mesh (3 / 3 / 4
 * floats per vertex) of which this rank has written its own shard IN PLACE (give fyx_lbs_skin_device the addresses
 * d_pos_all + 3 * begin, ...); on return (stream-ordered) every rank holds every shard.  ONE grouped RCCL operation
 * for all streams and all (ragged) shards -- ncclGroupStart, one ncclBroadcast per (stream, rank) rooted at the shard's
 * owner, ncclGroupEnd -- on the context stream, after the GPU-side join of the skinning launches in flight.  All ranks
 * must pass the same n_verts and the same set of non-null streams.
 * Option "comm.form" (fyx_set_option, the same value on every rank) picks how the shards travel inside that one group:
 * 0 (default) the broadcasts above; 1 point to point -- every rank ncclSend's its shard to each other rank and ncclRecv's
 * each other shard where it belongs (2 (n - 1) calls per stream and rank; RCCL fuses grouped send / recv into one kernel
 * over the xGMI links, no root and no tree).  Same bytes in the same places either way.
 * 2: ONE in-place ncclAllGather per stream over EQUAL shards -- the collective RCCL tunes hardest.  The shards are then the
 * padded cut of fyx_shard_vertex_range_padded (not the ragged one) and every d_*_all holds n_ranks * shard_verts vertices; the
 * vertices past n_verts in the last shard(s) are padding (whatever the buffer held travels; nothing reads it).  Because that form
 * WRITES n_ranks * shard_verts vertices per stream -- more than n_verts -- it runs only through fyx_allgather_skinned_padded[_all],
 * which are told what the buffers hold (capacity_verts, in vertices, the same for every stream) and return FYX_ERR_INVALID_ARG when
 * that is less than n_ranks * shard_verts (FYX_ERR_UNSUPPORTED when the product does not fit 32 bits); with "comm.form" = 2 the
 * calls without a capacity refuse (FYX_ERR_INVALID_ARG): an option never changes how much a call writes.  The padded entry points
 * always use the all-gather form, whatever the option says. */
int fyx_allgather_skinned(fyx_ctx* ctx, uint32_t n_verts, float* d_pos_all, float* d_normal_all, float* d_tangent_all);
int fyx_allgather_skinned_padded(fyx_ctx* ctx, uint32_t n_verts, uint32_t capacity_verts, float* d_pos_all, float* d_normal_all, float* d_tangent_all);
/* ONE PROCESS driving several GPUs -- the engine is one process with one update thread, so this is the
 * form its shim uses: contexts ctxs[0..n) made by fyx_init on n different devices, every call from the same thread.
 * RCCL requires a thread that drives several communicators to issue each collective for all of them inside one group,
 * hence the array forms:
 *   fyx_comm_init_all       ncclGetUniqueId + one grouped ncclCommInitRank per context: ctxs[i] becomes rank i of n
 *                           (no id to hand around).  Fails as a whole: no context keeps a half-made communicator.
 *   fyx_allgather_skinned_all   fyx_allgather_skinned for every GPU at once: d_*_all[i] is context i's FULL stream on its
 *                           own device, of which it has written shard i (fyx_shard_vertex_range(n_verts, i, n)) in
 *                           place; a stream is either null (the array pointer) for all GPUs or non-null for all.  Each
 *                           GPU's part is ordered on its own context stream after its skinning launches in flight.
 * Errors are reported on ctxs[0] (fyx_last_error).  The per-context calls above stay the form for one process (or thread)
 * per GPU; the two must not be mixed on one communicator. */
int fyx_comm_init_all(fyx_ctx* const* ctxs, int n);
int fyx_allgather_skinned_all(fyx_ctx* const* ctxs, int n, uint32_t n_verts, float* const* d_pos_all,
                              float* const* d_normal_all, float* const* d_tangent_all);
int fyx_allgather_skinned_padded_all(fyx_ctx* const* ctxs, int n, uint32_t n_verts, uint32_t capacity_verts, float* const* d_pos_all,
                                     float* const* d_normal_all, float* const* d_tangent_all);

/* ---- importer / editor helpers (pure host functions: no context, no GPU) ---------------- */
/* Which points of a sampled curve the glTF importer keeps (fyrox-impl/src/resource/gltf/simplify.rs:39-66 find_important_points,
 * applied to every imported curve by gltf/animation.rs:155-163 with the binding's epsilon / max_step, :50-65: Position 0.001 / inf,
 * Rotation pi/180 / pi/4, Scale 0.1 / inf, morph weights 0.001 / inf): Ramer-Douglas-Peucker on (x, y) with the reference's
 * vertical distance, then at most max_step between kept values (max_step = INFINITY: no limit), and a curve of two equal values
 * collapses to one key.  out_indices has room for n; *out_count = the number kept.  Pinned by the reference's 14 tests
 * (simplify.rs:145-229, tests/golden/fyrox_unit_vectors.json). */
int fyx_curve_simplify(const float* x, const float* y, uint32_t n, float epsilon, float max_step, uint32_t* out_indices, uint32_t* out_count);
/* The triangles of a BlendSpace (blendspace.rs:416-447 triangulate, called by fetch_weights after set_points): what
 * fyx_layer_add_blend_space takes as `triangles` when the engine does not bring its own.  Delaunay triangulation of the points in
 * insertion order; every triangle counter-clockwise, starting at its newest point, triangles listed by newest point -- the
 * reference's fixture (blendspace.rs:455-484: the unit square -> [2, 0, 1], [3, 0, 2]).  The reference delegates to the `spade`
 * crate, whose face order for larger inputs nothing in the reference pins; results of fetch_weights that depend on it (the fold
 * order of a triangle's three poses) are exact against the engine only with the engine's own triangles.  Fewer than three points:
 * no triangles (the reference returns false); a non-finite coordinate: FYX_ERR_INVALID_ARG.  Writes min(*out_count, capacity)
 * triangles of three point indices. */
int fyx_blend_space_triangulate(const float* points_xy, uint32_t n_points, uint32_t* out_triangles, uint32_t capacity, uint32_t* out_count);

/* ---- control plane without a GPU ----------------------------------------------------- */
/* A context with no device: registry and control-plane calls work, every call that would touch
 * the GPU returns FYX_ERR_NO_DEVICE.  It computes no poses and no vertices -- it exists so the
 * host logic (time advance, transitions, emitted fold programs) can be unit-tested anywhere. */
int fyx_init_control_only(fyx_ctx** out_ctx);
/* Advance the control plane of every instance by one frame exactly as fyx_animation_player_update
 * (mode 0) / fyx_absm_update (mode 1) would, and copy what they would send to the GPU:
 * times and ticked are [n_instances][n_animations]; program_offset is [n_instances + 1]; ops are
 * {opcode | arg << 8, f32 weight bits} pairs (opcodes: 0 END, 1 BLEND_ANIM, 2 PUSH, 3 POP_BLEND,
 * 4 RESET, 5 MASK, 6 APPLY, 7 APPLY_ANIM).  *n_ops returns the number of pairs needed.
 * A program is the reference's sequence of blend_with calls; a PUSH ... POP_BLEND pair (a sub-tree
 * evaluated into a pose of its own) appears only where it changes the result -- a sub-tree blended
 * into a pose nothing has been blended into yet is written in place, a one-clip sub-tree is one
 * BLEND_ANIM with the outer weight (NodePose::blend_with's copy rule, pose.rs:41-47).
 * ticked: bit 0 = the animation ticked; bit 1 = that tick started a new loop cycle; bit 2 =
 * speed > 0 (what Animation::update_root_motion derives, lib.rs:539-554). */
int fyx_animator_plan(fyx_ctx* ctx, uint64_t animator_id, int mode, float dt, float* times,
                      uint8_t* ticked, uint32_t* program_offset, uint32_t* ops,
                      uint32_t ops_capacity, uint32_t* n_ops);
/* mode -1: do not advance anything, copy what was planned last (e.g. by fyx_scene_plan).
 *
 * The host half of fyx_scene_update: every listed animator is planned (machine mode where it has a machine) on
 * the planner threads, nothing is sent to a GPU; read the results with fyx_animator_plan(.., mode -1, ..). */
int fyx_scene_plan(fyx_ctx* ctx, const uint64_t* animator_ids, uint32_t n_animators, float dt);
/* The block table fyx_scene_update would use for one stage of this scene (stages in launch order: 0 sample, curves on
 * the lanes; 1 sample, instances on the lanes; 2 property sample; 3 root motion; 4 root-motion fold; 5-8 update with
 * 64 / 128 / 192 / 256 threads; 9 property update): {job, x, y, z} per workgroup, *n_blocks = how many there are.
 * Depends on the animators' shapes only; needs no GPU (a test hook, like fyx_animator_plan). */
int fyx_debug_scene_tables(fyx_ctx* ctx, const uint64_t* animator_ids, uint32_t n_animators, int stage,
                           uint32_t* out_blocks, uint32_t capacity, uint32_t* n_blocks);

/* Test hooks for the CPU suite (no GPU, no context for the last two).
 *   fyx_debug_rig_walk: the rig's hierarchy-walk table as the update kernel reads it -- one word per node in level order:
 *     node | (parent + 1) << 10 | depth << 21 (any negative parent is a root: field 0).
 *   fyx_debug_rig_chunks: the same order as the one-character kernels walk it -- chunks of sixteen entries, every level padded to
 *     whole chunks: node | parent slot << 11 | (last chunk of its level) << 22; parent slot n_nodes = "no parent" (the identity),
 *     node n_nodes + 1 = padding.
 *   fyx_debug_span_value_at / fyx_debug_classify_fold_program: the kernels' own decision-making leaves (csrc/anim_leaves.h,
 *     __host__ __device__) compiled for the host -- Curve::value_at for the `need` (3 or 4) curves of one track on its span
 *     records (n_keys - 1 records of need + 1 float4: {loc[i-1], loc[i], the curves' left-key kinds as 8-bit fields of a u32, -}
 *     then per curve {value of key i - 1, value of key i, right tangent of key i - 1, left tangent of key i if that key is cubic
 *     else 0}), returning the values and the new hint; and the classifier that decides whether a fold
 *     program {opcode | arg << 8, f32 weight bits} x n_ops is "straight" (d leading PUSHes, k operands, a MASK before the APPLY,
 *     or the AnimationPlayer's APPLY_ANIM^k END) -- the function by which the host picks the update kernel's lean form. */
/* Test hook of the one-launch frame's in-grid wait: adds `delta` to the animator's device counter (the word the frame's sampler
 * workgroups add to and its update / skinning workgroups wait on) WITHOUT telling the host side -- a negative delta is "a word
 * overwritten from outside": the next one-launch frame's waits time out (option anim.wait_timeout_ms), that frame computes nothing,
 * and the next fyx_sync / pose entry returns FYX_ERR_HIP with the report.  Synchronous. */
int fyx_debug_frame_counter_add(fyx_ctx* ctx, uint64_t animator_id, int32_t delta);
int fyx_debug_rig_walk(fyx_ctx* ctx, uint64_t rig_id, uint32_t* out_words, uint32_t capacity, uint32_t* n_words);
int fyx_debug_rig_chunks(fyx_ctx* ctx, uint64_t rig_id, uint32_t* out_words, uint32_t capacity, uint32_t* n_words);
int fyx_debug_span_value_at(const float* span_records, uint32_t n_keys, uint32_t need, float time, uint32_t hint, float out_values[4], uint32_t* out_hint);
int fyx_debug_classify_fold_program(const uint32_t* ops_xy, uint32_t n_ops, uint32_t* out_d, uint32_t* out_k, int* out_mask, int* out_player,
                                    int* out_straight);

/* The root-motion program of the frame fyx_animator_plan planned last (mode 1, tracking on):
 * program_offset is [n_instances + 1]; ops are {opcode, dst slot, src slot | animation, f32
 * weight bits} quadruples (opcodes: 0 END, 1 SET_ANIM, 2 BLEND, 3 COPY).  Slots: for each layer
 * its pose nodes in handle order, then the layer's final pose; the machine's final pose is the
 * last slot (*n_slots).  slices is [n_instances][n_animations][2] = time_slice {start, end}. */
int fyx_animator_plan_root_motion(fyx_ctx* ctx, uint64_t animator_id, uint32_t* program_offset,
                                  uint32_t* ops, uint32_t ops_capacity, uint32_t* n_ops,
                                  uint32_t* n_slots, float* slices);

#ifdef __cplusplus
}
#endif
#endif /* FYROX_HIP_H */
