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
/* Kernel tuning knobs (block size, grid multiple, prefetch, exact vs fused arithmetic).
 * Unknown keys return FYX_ERR_INVALID_ARG.  Keys: "lbs.block", "lbs.blocks_per_cu", "lbs.prefetch",
 * "lbs.exact" (1 = reference operation order, no FMA contraction: bit-identical to the CPU path;
 * 0 = fused multiply-add, within 1e-5 relative), "lbs.nt" (non-temporal loads/stores),
 * "lbs.streams" (1..4 worker streams for independent skinning launches, see fyx_join). */
int fyx_set_option(fyx_ctx* ctx, const char* key, int value);
int fyx_get_option(fyx_ctx* ctx, const char* key, int* value);

/* GPU-side timing on the context's stream (hipEvent pair): begin records an event, end records
 * a second one, waits for it and returns the elapsed milliseconds between the two. */
int fyx_timer_begin(fyx_ctx* ctx);
int fyx_timer_end(fyx_ctx* ctx, float* out_ms);

/* ---- device memory (for callers without their own HIP allocator) ---------------------- */
int fyx_malloc(fyx_ctx* ctx, size_t bytes, void** out_device_ptr);
int fyx_free(fyx_ctx* ctx, void* device_ptr);
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

#ifdef __cplusplus
}
#endif
#endif /* FYROX_HIP_H */
