// fyx_ctx.h -- the context object and the helpers every C-ABI translation unit shares.
#pragma once
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <unordered_map>

#include "../../include/fyrox_hip.h"
#include "fyx_internal.h"

namespace fyx {
struct Mesh {
    uint32_t n_verts = 0;
    uint32_t max_bone_index = 0;
    void* block = nullptr;  // one allocation, streams carved at 256-byte boundaries
    float* pos = nullptr;
    float* nrm = nullptr;
    float* tan = nullptr;
    float* wgt = nullptr;
    uint32_t* idx = nullptr;
    // BlendShapesContainer offsets, re-tiled (own allocation; replaced by fyx_mesh_set_blend_shapes)
    uint16_t* shapes = nullptr;
    uint32_t n_shapes = 0;
    // the interleaved VertexBuffer bytes as uploaded by fyx_mesh_upload (padded by one 64-vertex unit), kept for
    // the vertex-buffer-in / vertex-buffer-out skinning path; null after fyx_mesh_upload_soa
    unsigned char* aos = nullptr;
    uint32_t stride = 0;
    int off_pos = -1, off_nrm = -1, off_tan = -1, off_wgt = -1, off_idx = -1;
};
struct AnimStore;  // anim_api.hip: tracks data, rigs, animators, bone lists
void anim_store_destroy(AnimStore*);
struct Comm;        // comm_api.hip: RCCL communicator (dlopen'ed on first use)
void comm_destroy(Comm*);
// Double-buffered pinned staging + device block for per-frame control data (the pose path's per-animator and
// per-scene control blocks, the job tables of fyx_lbs_skin_batch): frame k+1 is written and uploaded while frame k's
// kernels run.
struct CtrlBuffers {
    void* d[2] = {nullptr, nullptr};
    size_t d_bytes[2] = {0, 0};
    hipEvent_t d_consumed[2] = {nullptr, nullptr};   // the kernels that read d[slot] have finished
    bool d_in_use[2] = {false, false};
    hipStream_t d_consumer[2] = {nullptr, nullptr};  // the stream d_consumed[slot] was recorded on
    bool h_by_consumed[2] = {false, false};          // the staging block is free when d_consumed[slot] is (in-stream copies: no event of its own)
    void* h[2] = {nullptr, nullptr};
    size_t h_bytes[2] = {0, 0};
    hipEvent_t h_ev[2] = {nullptr, nullptr};
    bool h_busy[2] = {false, false};
    int next = 0;
};

struct SkinBatch;   // fyx_api.hip: cached tables of fyx_lbs_skin_batch
void skin_batch_destroy(SkinBatch*);

class PlanPool;     // anim_api.hip: host threads that plan a crowd's frame
void plan_pool_destroy(PlanPool*);

}  // namespace fyx
using fyx::Mesh;

struct fyx_ctx {
    int device = 0;          // -1: control-only context (no GPU; data-path calls fail)
    hipStream_t own_stream = nullptr;
    hipStream_t upload_stream = nullptr;   // per-frame control blocks of the pose path (anim_api.hip)
    hipStream_t stream = nullptr;
    fyx::LbsTuning lbs;
    std::unordered_map<uint64_t, Mesh> meshes;
    uint64_t mesh_gen = 0;   // bumped by everything that changes a mesh record (cached batch plans are made from them)
    std::string err = "";
    // scratch: staging for host-variant calls, grown on demand
    void* scratch = nullptr;
    size_t scratch_bytes = 0;
    float* aabb_partials = nullptr;  // 6 * 2048 floats + 8
    uint32_t* d_u32 = nullptr;       // 1 word
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // Worker streams for independent skinning launches (see "stream semantics" in fyrox_hip.h).
    static constexpr int kMaxWorkers = 4;
    int timing = 0;     // option "lbs.timing": fyx_lbs_skin_device launches carry their own start / stop events
    std::vector<hipEvent_t> timing_ev;   // pairs, in launch order (fyx_debug_kernel_time sums and clears)
    size_t timing_used = 0;
    int n_workers = 2;  // option "lbs.streams"; 1 = launch on the context stream itself
    hipStream_t workers[kMaxWorkers] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t worker_done[kMaxWorkers] = {nullptr, nullptr, nullptr, nullptr};
    bool worker_busy[kMaxWorkers] = {false, false, false, false};
    uint64_t worker_seen[kMaxWorkers] = {0, 0, 0, 0};
    // anim.overlap: whole frames alternate between TWO streams -- frame n (its control block, pose kernels and the skinning launches
    // that follow) runs in order on stream n & 1 (0: the context stream, 1: alt_stream), so that frame n + 1's pose update runs
    // beside frame n's skinning.  The only cross-stream edge of a frame is "behind the previous frame's pose update" (pose_done);
    // everything else a frame depends on -- frame n - 2's skinning, which read the palette buffer it rewrites -- lies earlier on
    // its own stream.  See enter_pose.
    hipStream_t alt_stream = nullptr;
    hipEvent_t alt_done = nullptr;       // join: the context stream waits for what is on alt_stream
    bool alt_busy = false;
    int frame_idx = 0;                   // the stream the current frame runs on
    hipEvent_t pose_done[2] = {nullptr, nullptr};   // recorded behind a frame's last pose kernel, one per stream
    int pose_done_on = -1;               // stream of the last pose update, -1: none yet
    // The skinning the LIBRARY issues for registered skin outputs (fyx_animator_set_skin_output) writes the same vertex buffers every
    // frame: under anim.overlap the launches of frame n + 1 are ordered behind those of frame n on the other stream (skin_outputs_order /
    // skin_outputs_issued).  Skinning calls of the caller are not: their output buffers are the caller's to alternate.
    hipEvent_t skin_done[2] = {nullptr, nullptr};
    int skin_done_on = -1;
    // anim.overlap = 2 (streams by kind: pose kernels on the context stream, skinning on alt_stream): skin_done[k] is recorded on alt_stream at the
    // start of the frames of parity k ^ 1 (skin_mark: it has been), skin_waits_pose: the frame's first skinning launch still has to wait for its pose update
    bool skin_mark[2] = {false, false};
    bool skin_waits_pose = false;
    int stream_priority = 0; // option "streams.priority": 1 = the context's own stream (the pose path: short latency-bound kernels) is created
                             //   with the highest priority, the launch streams (skinning: long bandwidth-bound kernels) with the lowest
    int pose_cus = 0;        // option "streams.pose_cus": N > 0 = the context's own stream may only use N CUs (spread over the XCDs) and the
                             //   launch streams only the other 256 - N (hipExtStreamCreateWithCUMask); 0 = no masks
    int ctrl_mode = 2;       // option "anim.ctrl_upload": how a control block travels -- 0 its own upload stream + events, 1 a copy on the
                             //   consuming stream, 2 (default) a copy kernel on the consuming stream reading the pinned block
    uint64_t options_gen = 1;  // bumped by every fyx_set_option (cached launch plans of a scene are made from the options)
    int host_times_on = 0;   // option "debug.host_times": fyx_scene_update adds up what its sections cost the calling thread (fyx_debug_host_times)
    double host_times[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int timeline_on = 0;     // option "debug.timeline": pose_sample / pose_update / fyx_lbs_skin_device launches carry their own events
    struct TimelineRec { int kind; hipEvent_t start, stop; };
    std::vector<TimelineRec> timeline;       // in launch order (fyx_debug_timeline reads and clears)
    hipEvent_t fork_ev = nullptr;
    uint64_t fork_gen = 0;
    bool primary_dirty = true;  // context-stream work enqueued since the last fork event
    int next_worker = 0;
    fyx::Comm* comm = nullptr;
    int comm_form = 0;       // option "comm.form": how fyx_allgather_skinned moves the shards (0 one broadcast per shard, 1 grouped send / recv)
    fyx::AnimStore* anim = nullptr;
    int plan_threads = 8;    // option "anim.threads": host threads planning a crowd's frame (1 = the calling thread only)
    int sample_form = 0;     // option "anim.sample_form": 0 auto, 1 curves on the lanes, 2 instances on the lanes
    int pose_overlap = 0;    // option "anim.overlap": 1 = pose updates do not wait for in-flight skinning launches (see enter_pose)
    int inline_ctrl = 1;     // option "anim.inline_ctrl": 1 = a control block of <= 1 KB travels in the kernel arguments (no H2D copy)
    int one_launch = 1;      // option "anim.one_launch": one character's sampler and update kernels as one launch (FrameSync)
    int frame_skin = 1;      // option "anim.frame_skin": that launch also holds the workgroups that skin the animator's skin outputs (FrameSkin)
    int frame_skin_units = 0;   // option "anim.frame_skin_units": 64-vertex units a wave of those workgroups is given (as far as kFrameSkinMaxBlocks
                                //   allows); 0 = the smallest depth that gives every skinning workgroup a CU of its own (kFrameSkinAutoBlocks)
    int wait_timeout_ms = 500;  // option "anim.wait_timeout_ms": how long an in-grid wait of the one-launch frame lasts before it reports
    fyx::DeviceError* dev_err = nullptr;   // pinned, host-coherent: what a kernel that gave up wrote (check_device_error)
    bool reissuing = false;                // check_device_error is re-running a frame whose in-grid wait gave up (no second report is acted on meanwhile)
    int frames_reissued = 0;               // how many frames that has happened to (fyx_get_option "debug.frames_reissued")
    int upd_pack = 4;        // option "anim.update_pack": 0, 2 or 4 instances of a small rig (<= 64 nodes) per workgroup of a crowd's update launch
    int upd_lean = 1;        // option "anim.update_lean": 1 = frames whose fold programs are all straight run the update kernel without the interpreter
    int plan_split = 2048;   // option "anim.split": instances per planning task
    fyx::PlanPool* plan_pool = nullptr;
    fyx::SkinBatch* skin_batch[2] = {nullptr, nullptr};     // two cached batches (the frames of the two frame streams skin from different palettes)
    int skin_batch_last = 0;
};


namespace fyx {
int fail(fyx_ctx* c, int code, const char* fmt, ...);
int hip_fail(fyx_ctx* c, hipError_t e, const char* what);
size_t align_up(size_t x, size_t a);
int join_workers(fyx_ctx* c);
int bind_device(fyx_ctx* c);      // hipSetDevice(ctx's device) for the calling thread
int enter_primary(fyx_ctx* c);
// Host-side wait for everything the context has in flight on any of its streams.
int sync_all(fyx_ctx* c);
// Entry of the pose path (fyx_*_update, fyx_scene_update, fyx_animator_palette); *out = the stream its work goes to.  Default:
// enter_primary and the context stream.  With option anim.overlap = 1 a pose update starts a new FRAME on the other of the
// context's two frame streams (see fyx_ctx::alt_stream): frame n + 1's pose kernels (latency-bound, a few waves per CU) run
// beside frame n's skinning (bandwidth-bound).  The pose path touches nothing a skinning launch reads except the palette
// buffers it is told to write, which the caller therefore alternates: a pose update must not be given a palette buffer that
// a skinning launch issued since the previous pose update reads.
int enter_pose(fyx_ctx* c, hipStream_t* out);
// Behind the last pose kernel of the entry: the next pose entry (on the other stream, under anim.overlap) orders itself behind this point.
int exit_pose(fyx_ctx* c);
// The stream of an ordered skinning launch (fyx_lbs_skin_batch and friends): the current frame's under anim.overlap, else the context stream (joined).
int enter_skin(fyx_ctx* c, hipStream_t* out);
// Around the library's own skinning of registered skin outputs on the frame's stream `st` (see fyx_ctx::skin_done): `order` before a
// launch that writes them (also a pose launch that holds skinning workgroups), `issued` behind skinning launches of their own.
int skin_outputs_order(fyx_ctx* c, hipStream_t st, bool pose_launch);
int pose_behind_all_skinning(fyx_ctx* c, hipStream_t ps);
int skin_outputs_issued(fyx_ctx* c, hipStream_t st);
int ensure_scratch(fyx_ctx* c, size_t bytes);
// What kernels reported since the last look (fyx_ctx::dev_err): FYX_OK, or FYX_ERR_HIP with the report as the context's message;
// the block is cleared and the one-launch frame switched off for the context (anim.one_launch = 0: the multi-launch path has no in-grid wait).
int check_device_error(fyx_ctx* c);
// The latest frame of the animator `tag` names (or of the scene it was last updated in), run again as separate launches -- no in-grid
// wait -- and waited for: what check_device_error does about a one-launch frame that reported a timed-out wait (anim_api.hip).
int reissue_frame(fyx_ctx* c, uint64_t tag);
// Kernel arguments of a skinning launch of a registered mesh, validated as fyx_lbs_skin_device validates them.
int skin_args_of(fyx_ctx* c, uint64_t mesh_id, const float* d_palette, uint32_t n_bones, uint32_t n_instances,
                 float* d_out_pos, float* d_out_normal, float* d_out_tangent, fyx::LbsArgs* out);
// debug.timeline: arms g_launch_events for the next launch and files the pair under `kind` (0 skinning, 1 pose_sample, 2 pose_update)
int timeline_arm(fyx_ctx* c, int kind);
void free_ctrl(CtrlBuffers& B);
// Claims the next slot with room for `total` bytes; *h / *d are its staging and device blocks.
int ctrl_acquire(fyx_ctx* c, CtrlBuffers& B, size_t total, int* slot_out, char** h, char** d);
// Sends slot's staging block to its device block on the upload stream; `consumer` (default: the context stream) waits for it.
int ctrl_upload(fyx_ctx* c, CtrlBuffers& B, int slot, size_t total, hipStream_t consumer = nullptr);
// Marks the point on `consumer` after which the device block may be overwritten.
int ctrl_consumed(fyx_ctx* c, CtrlBuffers& B, int slot, hipStream_t consumer = nullptr);
}  // namespace fyx

#define FYX_HIP(c, call)                                               \
    do {                                                               \
        hipError_t e_ = (call);                                        \
        if (e_ != hipSuccess) return fyx::hip_fail((c), e_, #call);    \
    } while (0)

// Entry points whose context is called `c` bind the calling thread to the context's GPU first (see bind_device);
// fyx_init / fyx_init_control_only, which have no context yet, use FYX_GUARD_BEGIN_NOCTX.
#define FYX_GUARD_BEGIN_NOCTX try {
// ... and start from a clean slate: the runtime keeps the LAST error of the thread until somebody asks for it (hipGetLastError), and the
// launch helpers ask after every launch -- an allocation that failed and was reported two calls ago (or a failed call of the application's
// own) must not come back as this call's launch error.
#define FYX_GUARD_BEGIN try { if ((c) && (c)->device >= 0) { (void)hipSetDevice((c)->device); (void)hipGetLastError(); }
#define FYX_GUARD_END(c)                                                        \
    } catch (const std::bad_alloc&) {                                           \
        return fyx::fail((c), FYX_ERR_OOM, "host allocation failed");           \
    } catch (...) {                                                             \
        return fyx::fail((c), FYX_ERR_HIP, "unexpected C++ exception");         \
    }
