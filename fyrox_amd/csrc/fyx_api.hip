// fyx_api.hip -- the C ABI of libfyrox_hip.so (see include/fyrox_hip.h).
// Owns: the context (device, stream, options, scratch), the mesh registry (device SoA streams
// keyed by mesh id) and all argument validation.  No exception ever crosses the boundary.
#include "fyx_ctx.h"

namespace fyx {
// The events of the pose path -- a frame's pose update done (pose_done), a control block
// consumed -- only order kernels of THIS device that read what kernels of this device wrote: their release need not be a
// system-scope one (which writes the L2 back and invalidates it under whatever kernel is running).  FYX_EVENT_SCOPE=system restores
// the runtime's default for an A/B.  The events around the launch streams (worker_done, fork_ev) keep the default: what is ordered
// behind them may be an RCCL exchange, or kernels that read what other GPUs wrote into this one's memory.
static unsigned order_event_flags() {
    static const unsigned flags = [] {
        const char* e = getenv("FYX_EVENT_SCOPE");
        return (e && !strcmp(e, "system")) ? (unsigned)hipEventDisableTiming : (unsigned)(hipEventDisableTiming | hipEventDisableSystemFence);
    }();
    return flags;
}

int fail(fyx_ctx* c, int code, const char* fmt, ...) {
    if (c) {
        char buf[1024];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        c->err = buf;
    }
    return code;
}

int hip_fail(fyx_ctx* c, hipError_t e, const char* what) {
    const int code = (e == hipErrorOutOfMemory) ? FYX_ERR_OOM : FYX_ERR_HIP;
    (void)hipGetLastError();      // reported here: not to be found again by the next launch's hipGetLastError (see FYX_GUARD_BEGIN)
    return fail(c, code, "%s: %s (%d)", what, hipGetErrorString(e), (int)e);
}

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Make the context stream wait for every in-flight worker launch (GPU-side join, no host wait).
int join_workers(fyx_ctx* c) {
    if (c->alt_busy) {
        FYX_HIP(c, hipEventRecord(c->alt_done, c->alt_stream));
        FYX_HIP(c, hipStreamWaitEvent(c->stream, c->alt_done, 0));
        c->alt_busy = false;
    }
    for (int w = 0; w < fyx_ctx::kMaxWorkers; ++w) {
        if (!c->worker_busy[w]) continue;
        FYX_HIP(c, hipEventRecord(c->worker_done[w], c->workers[w]));
        FYX_HIP(c, hipStreamWaitEvent(c->stream, c->worker_done[w], 0));
        c->worker_busy[w] = false;
    }
    return FYX_OK;
}

// The calling thread's current HIP device becomes the context's: allocations, launches and event calls all act on
// "the current device", and nothing says that the thread that calls in is the one that called fyx_init, or that it has
// not used another context (another GPU) in between.  hipSetDevice is a thread-local store when nothing changes.
int bind_device(fyx_ctx* c) {
    if (c->device < 0) return fail(c, FYX_ERR_NO_DEVICE, "control-only context: this call needs a GPU");
    FYX_HIP(c, hipSetDevice(c->device));
    return FYX_OK;
}

// Called by every entry point that enqueues work on (or synchronises) the context stream.
int enter_primary(fyx_ctx* c) {
    if (int rc = bind_device(c)) return rc;
    int rc = join_workers(c);
    c->primary_dirty = true;
    return rc;
}

int sync_all(fyx_ctx* c) {
    if (int rc = enter_primary(c)) return rc;
    FYX_HIP(c, hipStreamSynchronize(c->stream));
    return FYX_OK;
}

int check_device_error(fyx_ctx* c) {
    fyx::DeviceError* e = c->dev_err;
    if (!e || __atomic_load_n(&e->code, __ATOMIC_ACQUIRE) == 0) return FYX_OK;
    const fyx::DeviceError r = *e;
    // Only the code is taken back, with an exchange: kernels of later frames may be writing a report of their own right now (theirs
    // then stays for the next check instead of being zeroed half-written).
    __atomic_exchange_n(&e->code, 0u, __ATOMIC_SEQ_CST);
    if (r.code == fyx::kDevErrFrameWait) {
        // The frame whose wait gave up computed NOTHING (no pose, no palette, no vertex: a frame is late or it is right) -- and the
        // engine's chain never skips a frame (machine/mod.rs:344-382 -> mesh/mod.rs:781-793): the one-launch form is switched off for
        // the context and the animator's latest frame is run again NOW as separate launches, which cannot wait for anything, and waited
        // for.  The caller gets FYX_OK and correct outputs; what happened stays readable in fyx_last_error and in "debug.frames_reissued".
        if (c->one_launch) { c->one_launch = 0; ++c->options_gen; }      // (cached scene plans are made from the options)
        if (c->reissuing) return FYX_OK;      // (a second workgroup's report of the same frame, seen while that frame is being re-run)
        c->reissuing = true;
        const int rc = fyx::reissue_frame(c, r.tag);
        c->reissuing = false;
        if (rc) return rc;      // (fyx_last_error says what the re-run failed with)
        ++c->frames_reissued;
        fail(c, FYX_OK,
             "warning: one-launch frame of animator %llu: workgroup %u waited %d ms for the frame's sampler workgroups (counter %u, target %u) and gave "
             "up; that launch computed nothing and the frame was run AGAIN as separate launches (sampler, update, skinning; no in-grid wait): its "
             "outputs are correct.  The one-launch form relies on the sampler workgroups of a grid being dispatched before the workgroups that wait "
             "for them; anim.one_launch is now 0 for this context",
             (unsigned long long)r.tag, r.block, c->wait_timeout_ms, r.seen, r.target);
        return FYX_OK;
    }
    return fail(c, FYX_ERR_HIP, "a kernel reported error %u (workgroup %u)", r.code, r.block);
}

static hipError_t make_stream(fyx_ctx* c, bool pose, hipStream_t* out);

int enter_pose(fyx_ctx* c, hipStream_t* out) {
    if (int rc = check_device_error(c)) return rc;      // what an earlier frame's kernels reported
    if (!c->pose_overlap) {
        *out = c->stream;
        return enter_primary(c);
    }
    if (int rc = bind_device(c)) return rc;
    const int idx = c->frame_idx ^ 1;
    c->frame_idx = idx;
    if (c->pose_overlap == 2 && c->alt_stream) {
        // Streams BY KIND: every pose kernel on the context stream, every skinning launch on the second stream, each in order.  Frame n's
        // pose path writes the palette buffers frame n - 2's skinning read (pairs: fyx_animator_set_palette_output_pair, or the caller's
        // two buffers): it waits for the skinning issued up to the start of frame n - 1, an event recorded a frame ago -- usually long
        // since signalled, so that no queue sits blocked on another (what a blocked queue costs on this runtime: ~12 us per hop).
        FYX_HIP(c, hipEventRecord(c->skin_done[idx ^ 1], c->alt_stream));      // the skinning of the frames up to n - 1
        c->skin_mark[idx ^ 1] = true;
        if (c->skin_mark[idx]) FYX_HIP(c, hipStreamWaitEvent(c->stream, c->skin_done[idx], 0));      // ... up to n - 2
        *out = c->stream;
        return FYX_OK;
    }
    if (!c->alt_stream) {
        // (streams.priority / streams.pose_cus: a frame stream carries pose kernels; the second stream of anim.overlap = 2 carries skinning only)
        FYX_HIP(c, make_stream(c, c->pose_overlap != 2, &c->alt_stream));
        // (the join of the frame stream into the context stream keeps the system scope: under anim.overlap the frame streams carry the
        // skinning launches too, and what waits behind this event on the context stream may be an RCCL exchange or a copy to the host)
        FYX_HIP(c, hipEventCreateWithFlags(&c->alt_done, hipEventDisableTiming));
        for (int k = 0; k < 2; ++k) FYX_HIP(c, hipEventCreateWithFlags(&c->pose_done[k], order_event_flags()));
        for (int k = 0; k < 2; ++k) FYX_HIP(c, hipEventCreateWithFlags(&c->skin_done[k], order_event_flags()));
        if (c->pose_overlap == 2) {      // (the first frame of the mode: nothing on the second stream yet)
            *out = c->stream;
            return FYX_OK;
        }
    }
    hipStream_t target = idx ? c->alt_stream : c->stream;
    if (idx) {
        // whatever other calls have put on the context stream since the last fork (uploads, copies, a borrowed stream's work)
        if (c->primary_dirty || c->stream != c->own_stream) {
            if (!c->fork_ev) FYX_HIP(c, hipEventCreateWithFlags(&c->fork_ev, hipEventDisableTiming));
            FYX_HIP(c, hipEventRecord(c->fork_ev, c->stream));
            ++c->fork_gen;
            c->primary_dirty = false;
            FYX_HIP(c, hipStreamWaitEvent(target, c->fork_ev, 0));
        }
        c->alt_busy = true;
    }
    // behind the previous frame's pose update (the animators' persistent state: hints, pose records, node transforms)
    if (c->pose_done_on >= 0 && c->pose_done_on != idx) FYX_HIP(c, hipStreamWaitEvent(target, c->pose_done[c->pose_done_on], 0));
    *out = target;
    return FYX_OK;
}

int exit_pose(fyx_ctx* c) {
    if (!c->pose_overlap || !c->alt_stream) return FYX_OK;
    const int idx = c->frame_idx;
    if (c->pose_overlap == 2) {      // the frame's skinning launches (second stream) wait for this point of the context stream
        FYX_HIP(c, hipEventRecord(c->pose_done[idx], c->stream));
        c->pose_done_on = idx;
        c->skin_waits_pose = true;
        return FYX_OK;
    }
    // work was enqueued on alt_stream since enter_pose: whatever joined the streams in between (ensure_device_state's sync_all, a
    // control block that grew -- both hit an animator's FIRST frame) cleared the flag, and a fyx_sync / readback that follows the
    // update directly must still wait for these kernels
    if (idx) c->alt_busy = true;
    FYX_HIP(c, hipEventRecord(c->pose_done[idx], idx ? c->alt_stream : c->stream));
    c->pose_done_on = idx;
    return FYX_OK;
}

// anim.overlap = 2, a pose launch that itself writes vertex outputs (a frame that skins in its own launch) or a palette buffer without a
// second one: behind ALL the skinning issued so far, not only the frames up to n - 2.
int pose_behind_all_skinning(fyx_ctx* c, hipStream_t ps) {
    if (c->pose_overlap != 2 || !c->alt_stream || !c->skin_mark[c->frame_idx ^ 1]) return FYX_OK;
    FYX_HIP(c, hipStreamWaitEvent(ps, c->skin_done[c->frame_idx ^ 1], 0));
    return FYX_OK;
}

int skin_outputs_order(fyx_ctx* c, hipStream_t st, bool pose_launch) {
    // (2: skinning launches are one stream's, in order; a POSE launch that writes vertex outputs lies behind all of them)
    if (c->pose_overlap == 2) return pose_launch ? pose_behind_all_skinning(c, st) : FYX_OK;
    if (c->skin_done_on >= 0 && c->skin_done_on != c->frame_idx) FYX_HIP(c, hipStreamWaitEvent(st, c->skin_done[c->skin_done_on], 0));
    c->skin_done_on = -1;     // (what follows on `st` lies behind it; a pose launch that skins is covered by the frame's pose_done)
    return FYX_OK;
}

int skin_outputs_issued(fyx_ctx* c, hipStream_t st) {
    if (!c->pose_overlap || !c->alt_stream || c->pose_overlap == 2) return FYX_OK;      // (2: the skinning launches are one stream's, in order)
    FYX_HIP(c, hipEventRecord(c->skin_done[c->frame_idx], st));
    c->skin_done_on = c->frame_idx;
    return FYX_OK;
}

int enter_skin(fyx_ctx* c, hipStream_t* out) {
    if (c->pose_overlap == 2 && c->alt_stream) {
        if (int rc = bind_device(c)) return rc;
        if (c->primary_dirty) {      // other calls have put work on the context stream (uploads, copies): behind all of it, the frame's pose update included
            if (!c->fork_ev) FYX_HIP(c, hipEventCreateWithFlags(&c->fork_ev, hipEventDisableTiming));
            FYX_HIP(c, hipEventRecord(c->fork_ev, c->stream));
            ++c->fork_gen;
            c->primary_dirty = false;
            c->skin_waits_pose = false;
            FYX_HIP(c, hipStreamWaitEvent(c->alt_stream, c->fork_ev, 0));
        } else if (c->skin_waits_pose && c->pose_done_on >= 0) {      // the frame's first skinning launch: behind its pose update
            FYX_HIP(c, hipStreamWaitEvent(c->alt_stream, c->pose_done[c->pose_done_on], 0));
            c->skin_waits_pose = false;
        }
        c->alt_busy = true;
        *out = c->alt_stream;
        return FYX_OK;
    }
    if (c->pose_overlap && c->alt_stream) {
        if (int rc = bind_device(c)) return rc;
        *out = c->frame_idx ? c->alt_stream : c->stream;
        if (c->frame_idx) c->alt_busy = true;
        return FYX_OK;
    }
    *out = c->stream;
    return enter_primary(c);
}

// The context's streams.  own_stream carries the pose path (a chain of short, latency-bound kernels), the launch streams the
// skinning (long, bandwidth-bound): with priorities (option streams.priority, default) a workgroup slot that frees up goes to
// the pose kernels first, so frame n + 1's pose update makes its way under frame n's skinning (anim.overlap) instead of
// queueing behind it; with streams.pose_cus = N the two kinds of stream get disjoint sets of CUs instead.
static hipError_t make_stream(fyx_ctx* c, bool pose, hipStream_t* out) {
    if (c->pose_cus > 0 && c->pose_cus < fyx::kCUs) {
        // CU-mask bits are dealt round-robin over the XCDs (bit i = CU i / 8 of XCD i % 8 in the order the driver numbers them):
        // the low N bits are N / 8 CUs of every XCD
        uint32_t mask[fyx::kCUs / 32];
        for (int wd = 0; wd < fyx::kCUs / 32; ++wd) {
            uint32_t m = 0;
            for (int b = 0; b < 32; ++b) {
                const bool low = wd * 32 + b < c->pose_cus;
                if (low == pose) m |= 1u << b;
            }
            mask[wd] = m;
        }
        return hipExtStreamCreateWithCUMask(out, fyx::kCUs / 32, mask);
    }
    if (c->stream_priority) {
        int least = 0, greatest = 0;
        hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (e != hipSuccess) return e;
        return hipStreamCreateWithPriority(out, hipStreamNonBlocking, pose ? greatest : least);
    }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}

// Re-creates the context's own streams after streams.priority / streams.pose_cus changed (everything in flight is waited for).
static int recreate_streams(fyx_ctx* c) {
    if (c->device < 0) return FYX_OK;
    if (int rc = enter_primary(c)) return rc;
    FYX_HIP(c, hipStreamSynchronize(c->stream));
    if (c->upload_stream) FYX_HIP(c, hipStreamSynchronize(c->upload_stream));
    for (int w = 0; w < fyx_ctx::kMaxWorkers; ++w) {
        if (!c->workers[w]) continue;
        FYX_HIP(c, hipStreamSynchronize(c->workers[w]));
        FYX_HIP(c, hipStreamDestroy(c->workers[w]));
        c->workers[w] = nullptr;
        c->worker_busy[w] = false;
        c->worker_seen[w] = 0;
    }
    if (c->alt_stream) {
        FYX_HIP(c, hipStreamSynchronize(c->alt_stream));
        FYX_HIP(c, hipStreamDestroy(c->alt_stream));
        c->alt_stream = nullptr;
        FYX_HIP(c, make_stream(c, c->pose_overlap != 2, &c->alt_stream));
        c->alt_busy = false;
        c->pose_done_on = -1;
        c->skin_done_on = -1;
        c->skin_mark[0] = c->skin_mark[1] = false;
        c->skin_waits_pose = false;
        c->frame_idx = 0;
    }
    const bool own_current = c->stream == c->own_stream;
    hipStream_t fresh = nullptr;
    FYX_HIP(c, make_stream(c, true, &fresh));
    FYX_HIP(c, hipStreamSynchronize(c->own_stream));
    FYX_HIP(c, hipStreamDestroy(c->own_stream));
    c->own_stream = fresh;
    if (own_current) c->stream = fresh;
    c->primary_dirty = true;
    return FYX_OK;
}

// Pick the stream for an independent skinning launch: a worker, ordered after everything that
// was on the context stream at this moment (one fork event per batch of context-stream work).
int acquire_launch_stream(fyx_ctx* c, hipStream_t* out) {
    if (c->pose_overlap && c->alt_stream) return enter_skin(c, out);    // the frame's own stream: behind its pose update, in order
    if (c->n_workers <= 1) {
        int rc = enter_primary(c);
        *out = c->stream;
        return rc;
    }
    if (int rc = bind_device(c)) return rc;
    const int w = c->next_worker;
    c->next_worker = (w + 1) % c->n_workers;
    if (!c->workers[w]) {
        FYX_HIP(c, make_stream(c, false, &c->workers[w]));
        if (!c->worker_done[w]) FYX_HIP(c, hipEventCreateWithFlags(&c->worker_done[w], hipEventDisableTiming));
    }
    if (!c->fork_ev) FYX_HIP(c, hipEventCreateWithFlags(&c->fork_ev, hipEventDisableTiming));
    if (c->primary_dirty || c->stream != c->own_stream) {  // a borrowed stream may have foreign work
        FYX_HIP(c, hipEventRecord(c->fork_ev, c->stream));
        ++c->fork_gen;
        c->primary_dirty = false;
    }
    if (c->worker_seen[w] != c->fork_gen) {
        FYX_HIP(c, hipStreamWaitEvent(c->workers[w], c->fork_ev, 0));
        c->worker_seen[w] = c->fork_gen;
    }
    c->worker_busy[w] = true;
    *out = c->workers[w];
    return FYX_OK;
}

void free_ctrl(CtrlBuffers& B) {
    for (int i = 0; i < 2; ++i) {
        if (B.d[i]) (void)hipFree(B.d[i]);
        if (B.d_consumed[i]) (void)hipEventDestroy(B.d_consumed[i]);
        if (B.h[i]) (void)hipHostFree(B.h[i]);
        if (B.h_ev[i]) (void)hipEventDestroy(B.h_ev[i]);
    }
    B = CtrlBuffers();
}

int ctrl_acquire(fyx_ctx* c, CtrlBuffers& B, size_t total, int* slot_out, char** h, char** d) {
    const int slot = B.next;
    B.next ^= 1;
    if (B.h_busy[slot]) {      // the copy out of the staging block: its own event, or the event behind the kernels that followed it
        FYX_HIP(c, hipEventSynchronize(B.h_by_consumed[slot] ? B.d_consumed[slot] : B.h_ev[slot]));
        B.h_busy[slot] = false;
    }
    if (total > B.h_bytes[slot]) {
        if (B.h[slot]) FYX_HIP(c, hipHostFree(B.h[slot]));
        B.h[slot] = nullptr;
        const size_t want = align_up(total + total / 2, 4096);
        // coherent (fine-grained): what the host has written is what a kernel reading the block directly sees (anim.ctrl_upload = 2)
        FYX_HIP(c, hipHostMalloc(&B.h[slot], want, hipHostMallocCoherent));
        B.h_bytes[slot] = want;
    }
    if (!B.h_ev[slot]) FYX_HIP(c, hipEventCreateWithFlags(&B.h_ev[slot], hipEventDisableTiming));
    if (total > B.d_bytes[slot]) {
        if (int rc = enter_primary(c)) return rc;            // whoever still reads the old block, on any launch stream
        FYX_HIP(c, hipStreamSynchronize(c->stream));
        if (B.d_in_use[slot] && B.d_consumer[slot] && B.d_consumer[slot] != c->stream) FYX_HIP(c, hipEventSynchronize(B.d_consumed[slot]));
        if (B.d[slot]) (void)hipFree(B.d[slot]);
        B.d[slot] = nullptr;
        const size_t want = align_up(total + total / 2, 4096);
        FYX_HIP(c, hipMalloc(&B.d[slot], want));
        B.d_bytes[slot] = want;
        B.d_in_use[slot] = false;
    }
    if (!B.d_consumed[slot]) FYX_HIP(c, hipEventCreateWithFlags(&B.d_consumed[slot], order_event_flags()));
    *slot_out = slot;
    *h = static_cast<char*>(B.h[slot]);
    *d = static_cast<char*>(B.d[slot]);
    return FYX_OK;
}

// How the control block travels (option anim.ctrl_upload):
//   0  its own stream: the block has no dependence on the kernels already queued on the consuming stream (on ONE stream that is the
//      previous frame's skinning, ~100 us of work), so it travels beside them and only the frame's first kernel waits for it --
//      a copy, two event records and two stream waits per frame on the host's clock;
//   1  a copy on the consuming stream itself, 2 a copy KERNEL there that reads the pinned block directly (a launch instead of a
//      copy command): no event, no wait -- the order is the stream's, and the staging block is free again when the event behind
//      the frame's kernels (ctrl_consumed) is.  Right when the consuming stream carries only pose work (anim.overlap: the
//      skinning is on the launch streams) or short frames.
int ctrl_upload(fyx_ctx* c, CtrlBuffers& B, int slot, size_t total, hipStream_t consumer) {
    hipStream_t cs = consumer ? consumer : c->stream;
    if (c->ctrl_mode == 0) {
        if (!c->upload_stream) FYX_HIP(c, hipStreamCreateWithFlags(&c->upload_stream, hipStreamNonBlocking));
        if (B.d_in_use[slot]) FYX_HIP(c, hipStreamWaitEvent(c->upload_stream, B.d_consumed[slot], 0));
        FYX_HIP(c, hipMemcpyAsync(B.d[slot], B.h[slot], total, hipMemcpyHostToDevice, c->upload_stream));
        FYX_HIP(c, hipEventRecord(B.h_ev[slot], c->upload_stream));
        FYX_HIP(c, hipStreamWaitEvent(cs, B.h_ev[slot], 0));
        B.h_busy[slot] = true;
        B.h_by_consumed[slot] = false;
        return FYX_OK;
    }
    // the block's last readers ran on another stream (fyx_set_stream in between): order behind them
    if (B.d_in_use[slot] && B.d_consumer[slot] != cs) FYX_HIP(c, hipStreamWaitEvent(cs, B.d_consumed[slot], 0));
    if (c->ctrl_mode == 2) {
        if (int rc = timeline_arm(c, 3)) return rc;
        FYX_HIP(c, fyx::launch_ctrl_copy(B.h[slot], B.d[slot], total, cs));
        fyx::g_launch_events = fyx::LaunchEvents();
    }
    else FYX_HIP(c, hipMemcpyAsync(B.d[slot], B.h[slot], total, hipMemcpyHostToDevice, cs));
    B.h_busy[slot] = true;
    B.h_by_consumed[slot] = true;     // ctrl_consumed follows every upload
    return FYX_OK;
}

int ctrl_consumed(fyx_ctx* c, CtrlBuffers& B, int slot, hipStream_t consumer) {
    hipStream_t cs = consumer ? consumer : c->stream;
    FYX_HIP(c, hipEventRecord(B.d_consumed[slot], cs));
    B.d_in_use[slot] = true;
    B.d_consumer[slot] = cs;
    return FYX_OK;
}

// fyx_lbs_skin_batch: the tables of the last batch (a scene sends the same one every frame: no re-upload then)
struct SkinBatch {
    CtrlBuffers ctrl;
    std::vector<char> last;     // bytes of the last uploaded tables
    int last_slot = -1;
    std::vector<char> build;    // scratch of the current call
    // fyx_lbs_skin_batch called again with the SAME job array (a scene's frame: same meshes, same buffers), no mesh changed since,
    // same launch options: the plan and its tables are the previous call's (256 jobs: the call costs the host 14 us to plan)
    std::vector<char> jobs_key;
    uint64_t key_mesh_gen = 0;
    fyx::LbsTuning key_tuning;
    void* plan = nullptr;       // the BatchPlan of jobs_key (defined below, outside this namespace)
    void (*plan_free)(void*) = nullptr;
    struct Placed { size_t o_segs = 0, o_blocks = 0; uint32_t grid = 0; };
    Placed ps[8], pa[8];
    bool placed_valid = false;
};
void skin_batch_destroy(SkinBatch* b) {
    if (!b) return;
    free_ctrl(b->ctrl);
    if (b->plan && b->plan_free) b->plan_free(b->plan);
    delete b;
}

int timeline_arm(fyx_ctx* c, int kind) {
    if (!c->timeline_on) return FYX_OK;
    if (c->timeline.size() >= 16384) return fail(c, FYX_ERR_INVALID_ARG, "debug.timeline: read the records (fyx_debug_timeline) every 16384 launches");
    fyx_ctx::TimelineRec r{kind, nullptr, nullptr};
    FYX_HIP(c, hipEventCreate(&r.start));
    FYX_HIP(c, hipEventCreate(&r.stop));
    c->timeline.push_back(r);
    fyx::g_launch_events.start = r.start;
    fyx::g_launch_events.stop = r.stop;
    return FYX_OK;
}

int ensure_scratch(fyx_ctx* c, size_t bytes) {
    if (bytes <= c->scratch_bytes) return FYX_OK;
    if (c->scratch) {
        FYX_HIP(c, hipStreamSynchronize(c->stream));
        FYX_HIP(c, hipFree(c->scratch));
        c->scratch = nullptr;
        c->scratch_bytes = 0;
    }
    const size_t want = align_up(bytes + bytes / 4, 1 << 20);
    FYX_HIP(c, hipMalloc(&c->scratch, want));
    c->scratch_bytes = want;
    return FYX_OK;
}

}  // namespace fyx

namespace {
using namespace fyx;

void free_mesh(Mesh& m) {
    if (m.block) (void)hipFree(m.block);
    if (m.shapes) (void)hipFree(m.shapes);
    if (m.aos) (void)hipFree(m.aos);
    m = Mesh();
}

// Allocate the SoA streams for n vertices.  Each stream is padded to a whole number of
// 1024-vertex chunks so vector loads of a ragged tail stay inside the allocation.
int alloc_mesh(fyx_ctx* c, Mesh& m, uint32_t n, bool has_nrm, bool has_tan) {
    const size_t np = align_up((size_t)n, 1024) + 1024;
    const size_t b_pos = align_up(np * 12, 256), b_nrm = has_nrm ? b_pos : 0;
    const size_t b_tan = has_tan ? align_up(np * 16, 256) : 0, b_wgt = align_up(np * 16, 256);
    const size_t b_idx = align_up(np * 4, 256);
    const size_t total = b_pos + b_nrm + b_tan + b_wgt + b_idx;
    void* blk = nullptr;
    FYX_HIP(c, hipMalloc(&blk, total));
    hipError_t e = hipMemsetAsync(blk, 0, total, c->stream);
    if (e != hipSuccess) { (void)hipFree(blk); return hip_fail(c, e, "hipMemsetAsync"); }
    char* p = static_cast<char*>(blk);
    m.block = blk;
    m.n_verts = n;
    m.pos = reinterpret_cast<float*>(p); p += b_pos;
    m.nrm = has_nrm ? reinterpret_cast<float*>(p) : nullptr; p += b_nrm;
    m.tan = has_tan ? reinterpret_cast<float*>(p) : nullptr; p += b_tan;
    m.wgt = reinterpret_cast<float*>(p); p += b_wgt;
    m.idx = reinterpret_cast<uint32_t*>(p);
    return FYX_OK;
}

int finish_upload(fyx_ctx* c, uint64_t mesh_id, Mesh& m) {
    hipError_t e = hipSuccess;
    e = fyx::launch_max_bone_index(m.idx, m.n_verts, c->d_u32, c->stream);
    if (e != hipSuccess) { free_mesh(m); return hip_fail(c, e, "max_bone_index"); }
    uint32_t mx = 0;
    e = hipMemcpyAsync(&mx, c->d_u32, 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { free_mesh(m); return hip_fail(c, e, "upload sync"); }
    m.max_bone_index = mx;
    auto it = c->meshes.find(mesh_id);
    if (it != c->meshes.end()) { free_mesh(it->second); c->meshes.erase(it); }
    c->meshes.emplace(mesh_id, m);
    ++c->mesh_gen;
    return FYX_OK;
}

Mesh* find_mesh(fyx_ctx* c, uint64_t id) {
    auto it = c->meshes.find(id);
    return it == c->meshes.end() ? nullptr : &it->second;
}

int check_skin_args(fyx_ctx* c, const Mesh* m, uint64_t mesh_id, const void* palette,
                    uint32_t n_bones, uint32_t n_instances) {
    if (!m) return fail(c, FYX_ERR_UNKNOWN_ID, "mesh %llu is not registered", (unsigned long long)mesh_id);
    if (!palette) return fail(c, FYX_ERR_INVALID_ARG, "palette is null");
    if (n_bones == 0 || n_bones > 256)
        return fail(c, FYX_ERR_INVALID_ARG, "n_bones=%u outside 1..256 (bone indices are u8)", n_bones);
    if (n_instances == 0) return fail(c, FYX_ERR_INVALID_ARG, "n_instances is 0");
    if (m->n_verts > 0 && m->max_bone_index >= n_bones)
        return fail(c, FYX_ERR_BONE_INDEX,
                    "mesh %llu references bone %u but the palette has %u matrices",
                    (unsigned long long)mesh_id, m->max_bone_index, n_bones);
    return FYX_OK;
}

fyx::LbsArgs make_args(const Mesh& m, const float* d_palette, uint32_t n_bones, uint32_t n_inst,
                       float* op, float* on, float* ot) {
    fyx::LbsArgs a;
    a.pos = m.pos; a.nrm = m.nrm; a.tan = m.tan; a.wgt = m.wgt; a.idx = m.idx;
    a.palette = d_palette;
    a.out_pos = op; a.out_nrm = on; a.out_tan = ot;
    a.n_verts = m.n_verts; a.n_bones = n_bones; a.n_instances = n_inst;
    return a;
}

}  // namespace

namespace fyx {
int skin_args_of(fyx_ctx* c, uint64_t mesh_id, const float* d_palette, uint32_t n_bones, uint32_t n_instances,
                 float* d_out_pos, float* d_out_normal, float* d_out_tangent, fyx::LbsArgs* out) {
    const Mesh* m = find_mesh(c, mesh_id);
    if (int rc = check_skin_args(c, m, mesh_id, d_palette, n_bones, n_instances)) return rc;
    if (d_out_normal && !m->nrm) return fail(c, FYX_ERR_MISSING_ATTRIBUTE, "mesh has no Normal attribute");
    if (d_out_tangent && !m->tan) return fail(c, FYX_ERR_MISSING_ATTRIBUTE, "mesh has no Tangent attribute");
    *out = make_args(*m, d_palette, n_bones, n_instances, d_out_pos, d_out_normal, d_out_tangent);
    return FYX_OK;
}
}  // namespace fyx

namespace {
// Validation and kernel arguments of one extended skinning job (fyx_lbs_skin_ex and its batch form).
int build_ex_args(fyx_ctx* c, uint64_t mesh_id, const fyx_skin_desc* d, fyx::LbsExArgs& x, bool& whole_spans) {
    if (!d) return fail(c, FYX_ERR_INVALID_ARG, "desc is null");
    const Mesh* m = find_mesh(c, mesh_id);
    int rc = check_skin_args(c, m, mesh_id, d->d_palette, d->n_bones, d->n_instances);
    if (rc) return rc;
    memset(&x, 0, sizeof x);
    x.a = make_args(*m, d->d_palette, d->n_bones, d->n_instances, d->d_out_pos, d->d_out_normal, d->d_out_tangent);
    whole_spans = false;
    if (d->d_out_vertices && d->out_stride == 0) {
        // the mesh's own layout: vertex buffer in, vertex buffer out
        if (d->d_out_pos || d->d_out_normal || d->d_out_tangent)
            return fail(c, FYX_ERR_INVALID_ARG, "give either the interleaved output or the SoA outputs");
        if (!m->aos && m->n_verts)
            return fail(c, FYX_ERR_MISSING_ATTRIBUTE, "out_stride 0 (the mesh's own vertex layout) needs a mesh uploaded with fyx_mesh_upload");
        if ((m->stride & 3u) || m->stride > 160 || ((m->off_pos | m->off_wgt | m->off_idx) & 3) ||
            (m->off_nrm >= 0 && (m->off_nrm & 3)) || (m->off_tan >= 0 && (m->off_tan & 3)))
            return fail(c, FYX_ERR_UNSUPPORTED, "vertex layout (stride %u) must be 4-byte aligned and at most 160 bytes", m->stride);
        if (reinterpret_cast<uintptr_t>(d->d_out_vertices) & 3u) return fail(c, FYX_ERR_UNSUPPORTED, "output buffer is not 4-byte aligned");
        whole_spans = true;
        x.out_aos = d->d_out_vertices;
        x.out_stride = m->stride;
        x.off_pos = m->off_pos; x.off_nrm = m->off_nrm; x.off_tan = m->off_tan;
        x.in_aos = m->aos;
        x.in_off_wgt = m->off_wgt; x.in_off_idx = m->off_idx;
    } else if (d->d_out_vertices) {
        if (d->d_out_pos || d->d_out_normal || d->d_out_tangent)
            return fail(c, FYX_ERR_INVALID_ARG, "give either the interleaved output or the SoA outputs");
        if ((d->out_stride & 3u)) return fail(c, FYX_ERR_UNSUPPORTED, "out_stride %u is not a multiple of 4", d->out_stride);
        struct { int off; uint32_t size; const char* name; bool have; } f[] = {
            {d->out_off_pos, 12, "Position", true}, {d->out_off_normal, 12, "Normal", m->nrm != nullptr},
            {d->out_off_tangent, 16, "Tangent", m->tan != nullptr}};
        for (auto& a : f) {
            if (a.off < 0) continue;
            if (!a.have) return fail(c, FYX_ERR_MISSING_ATTRIBUTE, "mesh has no %s attribute", a.name);
            if (a.off & 3) return fail(c, FYX_ERR_UNSUPPORTED, "%s offset %d is not 4-byte aligned", a.name, a.off);
            if ((uint64_t)a.off + a.size > d->out_stride)
                return fail(c, FYX_ERR_INVALID_ARG, "%s at offset %d does not fit vertex size %u", a.name, a.off, d->out_stride);
        }
        if (reinterpret_cast<uintptr_t>(d->d_out_vertices) & 3u) return fail(c, FYX_ERR_UNSUPPORTED, "output buffer is not 4-byte aligned");
        x.out_aos = d->d_out_vertices;
        x.out_stride = d->out_stride;
        x.off_pos = d->out_off_pos; x.off_nrm = d->out_off_normal; x.off_tan = d->out_off_tangent;
    } else {
        if (d->d_out_normal && !m->nrm) return fail(c, FYX_ERR_MISSING_ATTRIBUTE, "mesh has no Normal attribute");
        if (d->d_out_tangent && !m->tan) return fail(c, FYX_ERR_MISSING_ATTRIBUTE, "mesh has no Tangent attribute");
    }
    if (d->n_blend_shapes) {
        if (d->n_blend_shapes != m->n_shapes)
            return fail(c, FYX_ERR_INVALID_ARG, "%u blend-shape weights for a mesh with %u blend shapes", d->n_blend_shapes, m->n_shapes);
        if (!d->d_blend_shape_weights) return fail(c, FYX_ERR_INVALID_ARG, "blend-shape weights are null");
        x.shapes = m->shapes;
        x.shape_w = d->d_blend_shape_weights;
        x.n_shapes = m->shapes ? d->n_blend_shapes : 0;
        x.tiles_per_shape = (m->n_verts + 63) / 64;
    }
    return FYX_OK;
}

enum ExKind { kExWholeSpans, kExPlain, kExGeneral };
ExKind ex_kind(const fyx::LbsExArgs& x, bool whole_spans) {
    if (whole_spans) return kExWholeSpans;
    return (!x.out_aos && !x.n_shapes) ? kExPlain : kExGeneral;   // nothing extended asked for: the plain kernel
}

// Everything a batch call launches: segments for the batched kernels, the rest one launch each.
struct BatchPlan {
    struct SoaGroup {
        std::vector<fyx::LbsSegDev> segs;
        uint32_t units = 0, max_bones = 0;
        bool big = false;      // a mesh whose streams do not fit a buffer resource's 32-bit range: lbs_skin_batch, not its dyn form
        fyx::LbsTuning tuning(const fyx::LbsTuning& t) const { fyx::LbsTuning r = t; if (big) r.dyn = 0; return r; }
    };
    struct AosGroup { std::vector<fyx::LbsExSegDev> segs; uint32_t units = 0, max_bones = 0, max_stride = 0; };
    SoaGroup soa[8];                 // by output mask
    AosGroup aos[8];                 // by (layout bucket, blend shapes)
    std::vector<fyx::LbsArgs> crowds;     // instanced jobs big enough for the crowd kernel
    std::vector<fyx::LbsExArgs> general;  // blend shapes into SoA, scattered interleaved output
    static constexpr uint32_t kCrowdFrom = 16;

    static int aos_slot(uint32_t bucket, bool shapes) { return (bucket == 4 ? 0 : bucket == 5 ? 1 : bucket == 8 ? 2 : 3) * 2 + (shapes ? 1 : 0); }
    static uint32_t aos_bucket_of(int slot) { static const uint32_t b[4] = {4, 5, 8, 10}; return b[slot / 2]; }

    int add_soa(fyx_ctx* c, const fyx::LbsArgs& a) {
        if (a.n_verts == 0) return FYX_OK;
        if (a.n_instances >= kCrowdFrom) { crowds.push_back(a); return FYX_OK; }
        const int mask = (a.out_pos ? 1 : 0) | ((a.out_nrm && a.nrm) ? 2 : 0) | ((a.out_tan && a.tan) ? 4 : 0);
        if (!mask) return FYX_OK;
        SoaGroup& G = soa[mask];
        const uint32_t upi = (a.n_verts + 63) / 64;
        for (uint32_t i = 0; i < a.n_instances; ++i) {
            if ((uint64_t)G.units + upi > 0xffffffffull) return fail(c, FYX_ERR_UNSUPPORTED, "batch too large for one launch");
            fyx::LbsSegDev sg;
            sg.pos = a.pos; sg.nrm = a.nrm; sg.tan = a.tan; sg.wgt = a.wgt; sg.idx = a.idx;
            sg.palette = a.palette + (size_t)i * a.n_bones * 16;
            sg.out_pos = a.out_pos ? a.out_pos + (size_t)i * a.n_verts * 3 : nullptr;
            sg.out_nrm = a.out_nrm ? a.out_nrm + (size_t)i * a.n_verts * 3 : nullptr;
            sg.out_tan = a.out_tan ? a.out_tan + (size_t)i * a.n_verts * 4 : nullptr;
            sg.n_verts = a.n_verts;
            sg.n_bones = a.n_bones;
            sg.unit0 = G.units;
            sg.pad = 0;
            G.segs.push_back(sg);
            G.units += upi;
            G.max_bones = std::max(G.max_bones, a.n_bones);
            G.big = G.big || a.n_verts > 0x0fffffffu;
        }
        return FYX_OK;
    }

    int add_aos(fyx_ctx* c, const fyx::LbsExArgs& x) {
        if (x.a.n_verts == 0) return FYX_OK;
        const uint32_t bucket = fyx::lbs_aos_bucket(x.out_stride);
        if (!bucket) return fail(c, FYX_ERR_UNSUPPORTED, "vertex layout (stride %u)", x.out_stride);
        AosGroup& G = aos[aos_slot(bucket, x.n_shapes > 0)];
        const uint32_t upi = (x.a.n_verts + 63) / 64;
        for (uint32_t i = 0; i < x.a.n_instances; ++i) {
            if ((uint64_t)G.units + upi > 0xffffffffull) return fail(c, FYX_ERR_UNSUPPORTED, "batch too large for one launch");
            fyx::LbsExSegDev sg;
            sg.in_aos = x.in_aos;
            sg.out_aos = x.out_aos + (size_t)i * x.a.n_verts * x.out_stride;
            sg.palette = x.a.palette + (size_t)i * x.a.n_bones * 16;
            sg.shapes = x.shapes;
            sg.shape_w = x.n_shapes ? x.shape_w + (size_t)i * x.n_shapes : nullptr;
            sg.n_verts = x.a.n_verts; sg.n_bones = x.a.n_bones; sg.n_shapes = x.n_shapes; sg.tiles_per_shape = x.tiles_per_shape;
            sg.stride = x.out_stride;
            sg.off_pos = x.off_pos; sg.off_nrm = x.off_nrm; sg.off_tan = x.off_tan;
            sg.in_off_wgt = x.in_off_wgt; sg.in_off_idx = x.in_off_idx;
            sg.unit0 = G.units;
            sg.pad = 0;
            G.segs.push_back(sg);
            G.units += upi;
            G.max_bones = std::max(G.max_bones, x.a.n_bones);
            G.max_stride = std::max(G.max_stride, x.out_stride);
        }
        return FYX_OK;
    }
};

// d_block_seg[b] = the segment holding workgroup b's first unit
template <typename Seg>
void fill_block_segs(uint32_t* bs, uint32_t grid, uint32_t units, const std::vector<Seg>& segs) {
    uint32_t sg = 0;
    for (uint32_t b = 0; b < grid; ++b) {
        const uint32_t u = (uint32_t)(((uint64_t)b * units) / grid);
        while (sg + 1 < segs.size() && segs[sg + 1].unit0 <= u) ++sg;
        bs[b] = sg;
    }
}

// Tables to the device (unless they are the previous call's), then one launch per non-empty group and per leftover job.
void batch_plan_free(void* p) { delete static_cast<BatchPlan*>(p); }

// reuse: P is the plan of the previous call (SkinBatch::plan) and nothing it was made from changed -- its tables are on the device.
// The cached batch a call works in: `which` < 0 picks the one used longest ago.
SkinBatch& batch_entry(fyx_ctx* c, int which) {
    const int e = which >= 0 ? which : 1 - c->skin_batch_last;
    if (!c->skin_batch[e]) c->skin_batch[e] = new SkinBatch();
    c->skin_batch_last = e;
    return *c->skin_batch[e];
}

int run_batch_plan(fyx_ctx* c, SkinBatch& B, BatchPlan& P, bool reuse = false) {
    // One launch that fills the chip: nothing to gain from a worker stream, and on the context stream the table
    // buffers have a single consumer to order their reuse against.
    hipStream_t st = nullptr;
    if (int sr = enter_skin(c, &st)) return sr;
    using Placed = SkinBatch::Placed;
    Placed ps[8], pa[8];
    size_t total = 0;
    if (!reuse) B.jobs_key.clear();      // whatever tables this call uploads replace the ones a cached plan points at
    reuse = reuse && B.placed_valid && B.last_slot >= 0;
    if (reuse) {
        memcpy(ps, B.ps, sizeof ps);
        memcpy(pa, B.pa, sizeof pa);
        total = B.last.size();
    }
    B.placed_valid = false;
    if (!reuse) {
    for (int k = 1; k < 8; ++k) {
        if (P.soa[k].segs.empty()) continue;
        ps[k].grid = fyx::lbs_batch_grid(P.soa[k].units, P.soa[k].tuning(c->lbs));
        ps[k].o_segs = total;
        total += align_up(P.soa[k].segs.size() * sizeof(fyx::LbsSegDev), 256);
        ps[k].o_blocks = total;
        total += align_up((size_t)ps[k].grid * 4, 256);
    }
    for (int k = 0; k < 8; ++k) {
        BatchPlan::AosGroup& G = P.aos[k];
        if (G.segs.empty()) continue;
        FYX_HIP(c, fyx::lbs_aos_batch_grid(G.units, G.max_bones, G.max_stride, BatchPlan::aos_bucket_of(k), (k & 1) != 0, c->lbs, &pa[k].grid));
        pa[k].o_segs = total;
        total += align_up(G.segs.size() * sizeof(fyx::LbsExSegDev), 256);
        pa[k].o_blocks = total;
        total += align_up((size_t)pa[k].grid * 4, 256);
    }
    }
    if (total) {
        if (!reuse) {
        B.build.assign(total, 0);
        for (int k = 1; k < 8; ++k) {
            if (P.soa[k].segs.empty()) continue;
            memcpy(B.build.data() + ps[k].o_segs, P.soa[k].segs.data(), P.soa[k].segs.size() * sizeof(fyx::LbsSegDev));
            fill_block_segs(reinterpret_cast<uint32_t*>(B.build.data() + ps[k].o_blocks), ps[k].grid, P.soa[k].units, P.soa[k].segs);
        }
        for (int k = 0; k < 8; ++k) {
            if (P.aos[k].segs.empty()) continue;
            memcpy(B.build.data() + pa[k].o_segs, P.aos[k].segs.data(), P.aos[k].segs.size() * sizeof(fyx::LbsExSegDev));
            fill_block_segs(reinterpret_cast<uint32_t*>(B.build.data() + pa[k].o_blocks), pa[k].grid, P.aos[k].units, P.aos[k].segs);
        }
        }
        int slot = B.last_slot;
        const bool same = reuse || (slot >= 0 && B.last.size() == total && memcmp(B.last.data(), B.build.data(), total) == 0);
        if (!same) {
            char *h = nullptr, *d = nullptr;
            if (int rc = ctrl_acquire(c, B.ctrl, total, &slot, &h, &d)) return rc;
            memcpy(h, B.build.data(), total);
            if (int rc = ctrl_upload(c, B.ctrl, slot, total, st)) return rc;
            B.last.swap(B.build);
            B.last_slot = slot;
        } else if (B.ctrl.h_busy[slot] && !B.ctrl.h_by_consumed[slot]) {
            FYX_HIP(c, hipStreamWaitEvent(st, B.ctrl.h_ev[slot], 0));   // a borrowed stream may have changed since the upload
        } else if (B.ctrl.d_in_use[slot] && B.ctrl.d_consumer[slot] != st) {
            FYX_HIP(c, hipStreamWaitEvent(st, B.ctrl.d_consumed[slot], 0));   // (in-stream upload: the event behind its first readers covers it)
        }
        const char* d = static_cast<const char*>(B.ctrl.d[slot]);
        for (int k = 1; k < 8; ++k) {
            const BatchPlan::SoaGroup& G = P.soa[k];
            if (G.segs.empty()) continue;
            FYX_HIP(c, fyx::launch_lbs_batch(reinterpret_cast<const fyx::LbsSegDev*>(d + ps[k].o_segs), (uint32_t)G.segs.size(),
                                             reinterpret_cast<const uint32_t*>(d + ps[k].o_blocks), ps[k].grid, G.units, G.max_bones,
                                             k, G.tuning(c->lbs), st));
        }
        for (int k = 0; k < 8; ++k) {
            const BatchPlan::AosGroup& G = P.aos[k];
            if (G.segs.empty()) continue;
            FYX_HIP(c, fyx::launch_lbs_aos_batch(reinterpret_cast<const fyx::LbsExSegDev*>(d + pa[k].o_segs), (uint32_t)G.segs.size(),
                                                 reinterpret_cast<const uint32_t*>(d + pa[k].o_blocks), pa[k].grid, G.units,
                                                 G.max_bones, G.max_stride, BatchPlan::aos_bucket_of(k), (k & 1) != 0, c->lbs, st));
        }
        if (int rc = ctrl_consumed(c, B.ctrl, slot, st)) return rc;
        memcpy(B.ps, ps, sizeof ps);
        memcpy(B.pa, pa, sizeof pa);
        B.placed_valid = true;
    }
    for (const fyx::LbsArgs& a : P.crowds) FYX_HIP(c, fyx::launch_lbs(a, c->lbs, st));   // vertices held in registers across the instances
    for (const fyx::LbsExArgs& x : P.general) FYX_HIP(c, fyx::launch_lbs_ex(x, c->lbs, st));
    return FYX_OK;
}

}  // namespace

extern "C" {

const char* fyx_version(void) { return "fyrox_hip 0.2.0 (gfx950)"; }

int fyx_init(fyx_ctx** out_ctx, int device_ordinal) {
    if (!out_ctx) return FYX_ERR_INVALID_ARG;
    *out_ctx = nullptr;
    FYX_GUARD_BEGIN_NOCTX
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return FYX_ERR_NO_DEVICE;
    if (device_ordinal < 0 || device_ordinal >= n) return FYX_ERR_NO_DEVICE;
    if (hipSetDevice(device_ordinal) != hipSuccess) return FYX_ERR_NO_DEVICE;
    fyx_ctx* c = new fyx_ctx();
    c->device = device_ordinal;
    if (make_stream(c, true, &c->own_stream) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&c->aabb_partials), (6 * 2048 + 8) * sizeof(float)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&c->d_u32), 64) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&c->dev_err), sizeof(fyx::DeviceError), hipHostMallocCoherent) != hipSuccess) {
        fyx_shutdown(c);
        return FYX_ERR_HIP;
    }
    memset(c->dev_err, 0, sizeof(fyx::DeviceError));
    c->stream = c->own_stream;
    *out_ctx = c;
    return FYX_OK;
    FYX_GUARD_END(nullptr)
}

void fyx_shutdown(fyx_ctx* c) {
    if (!c) return;
    if (c->device >= 0) (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    fyx::anim_store_destroy(c->anim);
    c->anim = nullptr;
    fyx::comm_destroy(c->comm);
    c->comm = nullptr;
    fyx::plan_pool_destroy(c->plan_pool);
    for (fyx::SkinBatch* b : c->skin_batch) fyx::skin_batch_destroy(b);
    c->plan_pool = nullptr;
    for (auto& kv : c->meshes) free_mesh(kv.second);
    if (c->scratch) (void)hipFree(c->scratch);
    if (c->aabb_partials) (void)hipFree(c->aabb_partials);
    if (c->d_u32) (void)hipFree(c->d_u32);
    if (c->dev_err) (void)hipHostFree(c->dev_err);
    for (hipEvent_t e : c->timing_ev) (void)hipEventDestroy(e);
    c->timing_ev.clear();
    for (const fyx_ctx::TimelineRec& r : c->timeline) { (void)hipEventDestroy(r.start); (void)hipEventDestroy(r.stop); }
    c->timeline.clear();
    for (int w = 0; w < fyx_ctx::kMaxWorkers; ++w) {
        if (c->workers[w]) { (void)hipStreamSynchronize(c->workers[w]); (void)hipStreamDestroy(c->workers[w]); }
        if (c->worker_done[w]) (void)hipEventDestroy(c->worker_done[w]);
    }
    if (c->alt_stream) { (void)hipStreamSynchronize(c->alt_stream); (void)hipStreamDestroy(c->alt_stream); }
    if (c->alt_done) (void)hipEventDestroy(c->alt_done);
    for (int k = 0; k < 2; ++k)
        if (c->pose_done[k]) (void)hipEventDestroy(c->pose_done[k]);
    for (int k = 0; k < 2; ++k)
        if (c->skin_done[k]) (void)hipEventDestroy(c->skin_done[k]);
    if (c->fork_ev) (void)hipEventDestroy(c->fork_ev);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->upload_stream) { (void)hipStreamSynchronize(c->upload_stream); (void)hipStreamDestroy(c->upload_stream); }
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

const char* fyx_last_error(const fyx_ctx* c) { return c ? c->err.c_str() : "null context"; }

int fyx_set_stream(fyx_ctx* c, void* s) {
    if (!c) return FYX_ERR_INVALID_ARG;
    if (int rc = enter_primary(c)) return rc;
    c->stream = s ? static_cast<hipStream_t>(s) : c->own_stream;
    return FYX_OK;
}

void* fyx_get_stream(fyx_ctx* c) { return c ? static_cast<void*>(c->stream) : nullptr; }

int fyx_join(fyx_ctx* c) {
    if (!c) return FYX_ERR_INVALID_ARG;
    return enter_primary(c);
}

int fyx_sync(fyx_ctx* c) {
    if (!c) return FYX_ERR_INVALID_ARG;
    if (int rc = enter_primary(c)) return rc;
    FYX_HIP(c, hipStreamSynchronize(c->stream));
    return check_device_error(c);     // what the kernels that have now finished reported (an in-grid wait that gave up)
}

int fyx_timer_begin(fyx_ctx* c) {
    if (!c) return FYX_ERR_INVALID_ARG;
    if (!c->ev0) { FYX_HIP(c, hipEventCreate(&c->ev0)); FYX_HIP(c, hipEventCreate(&c->ev1)); }
    if (int rc = enter_primary(c)) return rc;
    FYX_HIP(c, hipEventRecord(c->ev0, c->stream));
    return FYX_OK;
}

int fyx_timer_end(fyx_ctx* c, float* out_ms) {
    if (!c || !out_ms) return FYX_ERR_INVALID_ARG;
    if (!c->ev0) return fail(c, FYX_ERR_INVALID_ARG, "fyx_timer_end without fyx_timer_begin");
    if (int rc = enter_primary(c)) return rc;
    FYX_HIP(c, hipEventRecord(c->ev1, c->stream));
    FYX_HIP(c, hipEventSynchronize(c->ev1));
    FYX_HIP(c, hipEventElapsedTime(out_ms, c->ev0, c->ev1));
    return FYX_OK;
}

static int* option_slot(fyx_ctx* c, const char* key) {
    if (!key) return nullptr;
    if (!strcmp(key, "lbs.blocks_per_cu")) return &c->lbs.blocks_per_cu;
    if (!strcmp(key, "lbs.exact")) return &c->lbs.exact;
    if (!strcmp(key, "lbs.streams")) return &c->n_workers;
    if (!strcmp(key, "lbs.crowd")) return &c->lbs.crowd;
    if (!strcmp(key, "lbs.crowd_ipb")) return &c->lbs.crowd_ipb;
    if (!strcmp(key, "lbs.crowd_lean")) return &c->lbs.crowd_lean;
    if (!strcmp(key, "lbs.timing")) return &c->timing;
    if (!strcmp(key, "lbs.dyn")) return &c->lbs.dyn;
    if (!strcmp(key, "comm.form")) return &c->comm_form;
    if (!strcmp(key, "anim.threads")) return &c->plan_threads;
    if (!strcmp(key, "anim.split")) return &c->plan_split;
    if (!strcmp(key, "anim.overlap") || !strcmp(key, "debug.overlap")) return &c->pose_overlap;
    if (!strcmp(key, "anim.inline_ctrl")) return &c->inline_ctrl;
    if (!strcmp(key, "anim.sample_form")) return &c->sample_form;
    if (!strcmp(key, "anim.update_lean")) return &c->upd_lean;
    if (!strcmp(key, "anim.update_pack")) return &c->upd_pack;
    if (!strcmp(key, "anim.one_launch")) return &c->one_launch;
    if (!strcmp(key, "anim.frame_skin") || !strcmp(key, "debug.frame_skin")) return &c->frame_skin;
    if (!strcmp(key, "anim.frame_skin_units")) return &c->frame_skin_units;
    if (!strcmp(key, "anim.wait_timeout_ms")) return &c->wait_timeout_ms;
    if (!strcmp(key, "anim.ctrl_upload")) return &c->ctrl_mode;
    if (!strcmp(key, "streams.priority")) return &c->stream_priority;
    if (!strcmp(key, "streams.pose_cus")) return &c->pose_cus;
    if (!strcmp(key, "debug.timeline")) return &c->timeline_on;
    if (!strcmp(key, "debug.host_times")) return &c->host_times_on;
    if (!strcmp(key, "debug.frames_reissued")) return &c->frames_reissued;
    return nullptr;
}

int fyx_set_option(fyx_ctx* c, const char* key, int value) {
    if (!c) return FYX_ERR_INVALID_ARG;
    int* slot = option_slot(c, key);
    if (!slot) return fail(c, FYX_ERR_INVALID_ARG, "unknown option '%s'", key ? key : "(null)");
    // (options_gen -- what cached scene plans are compared against -- moves where a value is accepted, below: a refused value changes nothing)
    if (slot == &c->n_workers) {
        if (value < 1 || value > fyx_ctx::kMaxWorkers)
            return fail(c, FYX_ERR_INVALID_ARG, "lbs.streams must be 1..%d", fyx_ctx::kMaxWorkers);
        if (int rc = enter_primary(c)) return rc;
        c->next_worker = 0;
    }
    if (slot == &c->lbs.crowd && (value < -1 || value > 1))
        return fail(c, FYX_ERR_INVALID_ARG, "lbs.crowd must be -1 (auto), 0 or 1");
    if (slot == &c->lbs.crowd_ipb && (value < 0 || value > 4096))
        return fail(c, FYX_ERR_INVALID_ARG, "lbs.crowd_ipb must be 0 (auto) .. 4096");
    if (slot == &c->comm_form && (value < 0 || value > 2)) return fail(c, FYX_ERR_INVALID_ARG, "comm.form must be 0 (broadcasts), 1 (send / recv) or 2 (one all-gather over padded shards)");
    if (slot == &c->sample_form && (value < 0 || value > 2)) return fail(c, FYX_ERR_INVALID_ARG, "anim.sample_form must be 0, 1 or 2");
    if (slot == &c->inline_ctrl && value != 0 && value != 1) return fail(c, FYX_ERR_INVALID_ARG, "anim.inline_ctrl must be 0 or 1");
    // (the forms that did not pay -- streams by kind, the scene's update stage skinning whatever its size, the scene as one launch --
    // are reachable under their debug.* names only: the public options describe one way to do each thing)
    const bool debug_key = !strncmp(key, "debug.", 6);
    if (slot == &c->pose_overlap && (value < 0 || value > (debug_key ? 2 : 1)))
        return fail(c, FYX_ERR_INVALID_ARG, debug_key ? "debug.overlap must be 0, 1 or 2 (streams by kind)" : "anim.overlap must be 0 or 1");
    if (slot == &c->plan_split && value < 1) return fail(c, FYX_ERR_INVALID_ARG, "anim.split must be >= 1");
    if (slot == &c->plan_threads && (value < 1 || value > 64))
        return fail(c, FYX_ERR_INVALID_ARG, "anim.threads must be 1..64");
    if (slot == &c->lbs.blocks_per_cu && (value < 1 || value > 64))
        return fail(c, FYX_ERR_INVALID_ARG, "lbs.blocks_per_cu must be 1..64");
    if (slot == &c->ctrl_mode && (value < 0 || value > 2)) return fail(c, FYX_ERR_INVALID_ARG, "anim.ctrl_upload must be 0 (upload stream), 1 (copy in stream) or 2 (copy kernel)");
    if (slot == &c->stream_priority && value != 0 && value != 1) return fail(c, FYX_ERR_INVALID_ARG, "streams.priority must be 0 or 1");
    if (slot == &c->pose_cus && (value < 0 || value >= fyx::kCUs || (value & 7))) return fail(c, FYX_ERR_INVALID_ARG, "streams.pose_cus must be 0 or a multiple of 8 below %d", fyx::kCUs);
    if (slot == &c->upd_lean && value != 0 && value != 1) return fail(c, FYX_ERR_INVALID_ARG, "anim.update_lean must be 0 or 1");
    if (slot == &c->one_launch && value != 0 && value != 1) return fail(c, FYX_ERR_INVALID_ARG, "anim.one_launch must be 0 or 1");
    if (slot == &c->frame_skin && (value < 0 || value > (debug_key ? 3 : 1)))
        return fail(c, FYX_ERR_INVALID_ARG, debug_key ? "debug.frame_skin must be 0 .. 3" : "anim.frame_skin must be 0 or 1");
    if (slot == &c->frame_skin_units && (value < 0 || value > 64)) return fail(c, FYX_ERR_INVALID_ARG, "anim.frame_skin_units must be 0 (auto) .. 64");
    if (slot == &c->wait_timeout_ms && (value < 1 || value > 30000)) return fail(c, FYX_ERR_INVALID_ARG, "anim.wait_timeout_ms must be 1..30000");
    if (slot == &c->upd_pack && value != 0 && value != 2 && value != 4) return fail(c, FYX_ERR_INVALID_ARG, "anim.update_pack must be 0, 2 or 4");
    ++c->options_gen;
    if ((slot == &c->stream_priority || slot == &c->pose_cus) && *slot != value) {
        const int old = *slot;
        *slot = value;
        if (int rc = recreate_streams(c)) { *slot = old; return rc; }
        return FYX_OK;
    }
    if (slot == &c->pose_overlap && *slot != value && c->device >= 0) {
        if (int rc = enter_primary(c)) return rc;      // switching the mode joins everything once
        c->frame_idx = 0;
        c->pose_done_on = -1;
        c->skin_done_on = -1;
        c->skin_mark[0] = c->skin_mark[1] = false;
        c->skin_waits_pose = false;
        if (c->alt_stream && (c->stream_priority || c->pose_cus > 0)) {      // the second stream changes its kind between the two modes
            *slot = value;
            return recreate_streams(c);
        }
    }
    *slot = value;
    return FYX_OK;
}

int fyx_get_option(fyx_ctx* c, const char* key, int* value) {
    if (!c || !value) return FYX_ERR_INVALID_ARG;
    int* slot = option_slot(c, key);
    if (!slot) return fail(c, FYX_ERR_INVALID_ARG, "unknown option '%s'", key ? key : "(null)");
    *value = *slot;
    return FYX_OK;
}

int fyx_malloc(fyx_ctx* c, size_t bytes, void** out) {
    if (!c || !out) return FYX_ERR_INVALID_ARG;
    *out = nullptr;
    if (bytes == 0) return FYX_OK;
    if (int rc = bind_device(c)) return rc;
    FYX_HIP(c, hipMalloc(out, bytes));
    return FYX_OK;
}

int fyx_malloc_streams(fyx_ctx* c, uint32_t n, const size_t* bytes, void** out) {
    if (!c) return FYX_ERR_INVALID_ARG;
    if (n && (!bytes || !out)) return fail(c, FYX_ERR_INVALID_ARG, "fyx_malloc_streams: null array");
    for (uint32_t i = 0; i < n; ++i) out[i] = nullptr;
    if (int rc = bind_device(c)) return rc;
    for (uint32_t i = 0; i < n; ++i) {
        if (bytes[i] == 0) continue;
        const hipError_t e = hipMalloc(&out[i], bytes[i]);      // one allocation per stream: the point of the call
        if (e != hipSuccess) {
            for (uint32_t k = 0; k < i; ++k) {
                if (out[k]) (void)hipFree(out[k]);
                out[k] = nullptr;
            }
            out[i] = nullptr;
            return fail(c, e == hipErrorOutOfMemory ? FYX_ERR_OOM : FYX_ERR_HIP, "fyx_malloc_streams: stream %u of %u (%zu bytes): %s", i, n, bytes[i], hipGetErrorString(e));
        }
    }
    return FYX_OK;
}

int fyx_free(fyx_ctx* c, void* p) {
    if (!c) return FYX_ERR_INVALID_ARG;
    if (!p) return FYX_OK;
    if (int rc = enter_primary(c)) return rc;
    FYX_HIP(c, hipStreamSynchronize(c->stream));
    FYX_HIP(c, hipFree(p));
    return FYX_OK;
}

int fyx_memcpy_h2d(fyx_ctx* c, void* dst, const void* src, size_t bytes) {
    if (!c || (bytes && (!dst || !src))) return fail(c, FYX_ERR_INVALID_ARG, "null pointer");
    if (!bytes) return FYX_OK;
    if (int rc = enter_primary(c)) return rc;
    FYX_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    FYX_HIP(c, hipStreamSynchronize(c->stream));
    return FYX_OK;
}

int fyx_memcpy_d2h(fyx_ctx* c, void* dst, const void* src, size_t bytes) {
    if (!c || (bytes && (!dst || !src))) return fail(c, FYX_ERR_INVALID_ARG, "null pointer");
    if (!bytes) return FYX_OK;
    if (int rc = enter_primary(c)) return rc;
    FYX_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    FYX_HIP(c, hipStreamSynchronize(c->stream));
    return FYX_OK;
}

// ---- mesh registry ------------------------------------------------------------------------

int fyx_mesh_upload(fyx_ctx* c, uint64_t mesh_id, const uint8_t* aos, uint32_t n_verts,
                    uint32_t stride, int off_pos, int off_normal, int off_tangent, int off_weights,
                    int off_indices) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    if (n_verts && !aos) return fail(c, FYX_ERR_INVALID_ARG, "vertex bytes are null");
    if (stride == 0) return fail(c, FYX_ERR_INVALID_ARG, "vertex stride is 0");
    if (off_pos < 0 || off_weights < 0 || off_indices < 0)
        return fail(c, FYX_ERR_MISSING_ATTRIBUTE,
                    "Position, BoneWeight and BoneIndices attributes are required for skinning");
    struct { int off; uint32_t size; const char* name; } f[] = {
        {off_pos, 12, "Position"}, {off_normal, 12, "Normal"}, {off_tangent, 16, "Tangent"},
        {off_weights, 16, "BoneWeight"}, {off_indices, 4, "BoneIndices"}};
    for (auto& a : f)
        if (a.off >= 0 && (uint64_t)a.off + a.size > stride)
            return fail(c, FYX_ERR_INVALID_ARG, "%s at offset %d does not fit vertex size %u", a.name,
                        a.off, stride);
    Mesh m;
    if (int jr = enter_primary(c)) return jr;
    int rc = alloc_mesh(c, m, n_verts, off_normal >= 0, off_tangent >= 0);
    if (rc) return rc;
    if (n_verts) {
        // the interleaved bytes stay resident (padded by a 64-vertex unit so whole-span loads of a ragged tail
        // stay inside the allocation): fyx_lbs_skin_ex's vertex-buffer output path reads them directly
        const size_t bytes = (size_t)n_verts * stride;
        const size_t padded = (align_up((size_t)n_verts, 64) + 64) * stride;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&m.aos), padded);
        if (e != hipSuccess) { free_mesh(m); return hip_fail(c, e, "hipMalloc(vertex buffer)"); }
        e = hipMemsetAsync(m.aos + bytes, 0, padded - bytes, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(m.aos, aos, bytes, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess)
            e = fyx::launch_deinterleave(m.aos, n_verts, stride,
                                         off_pos, off_normal, off_tangent, off_weights, off_indices,
                                         m.pos, m.nrm, m.tan, m.wgt, m.idx, c->stream);
        if (e != hipSuccess) { free_mesh(m); return hip_fail(c, e, "mesh upload"); }
        m.stride = stride;
        m.off_pos = off_pos; m.off_nrm = off_normal; m.off_tan = off_tangent; m.off_wgt = off_weights; m.off_idx = off_indices;
    }
    return finish_upload(c, mesh_id, m);
    FYX_GUARD_END(c)
}

int fyx_mesh_upload_soa(fyx_ctx* c, uint64_t mesh_id, uint32_t n_verts, const float* pos,
                        const float* normal, const float* tangent, const float* weights,
                        const uint8_t* indices) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    if (n_verts && (!pos || !weights || !indices))
        return fail(c, FYX_ERR_MISSING_ATTRIBUTE,
                    "Position, BoneWeight and BoneIndices streams are required for skinning");
    Mesh m;
    if (int jr = enter_primary(c)) return jr;
    int rc = alloc_mesh(c, m, n_verts, normal != nullptr, tangent != nullptr);
    if (rc) return rc;
    if (n_verts) {
        const size_t n = n_verts;
        hipError_t e = hipMemcpyAsync(m.pos, pos, n * 12, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess && normal) e = hipMemcpyAsync(m.nrm, normal, n * 12, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess && tangent) e = hipMemcpyAsync(m.tan, tangent, n * 16, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(m.wgt, weights, n * 16, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(m.idx, indices, n * 4, hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) { free_mesh(m); return hip_fail(c, e, "mesh upload"); }
    }
    return finish_upload(c, mesh_id, m);
    FYX_GUARD_END(c)
}

int fyx_mesh_free(fyx_ctx* c, uint64_t mesh_id) {
    if (!c) return FYX_ERR_INVALID_ARG;
    auto it = c->meshes.find(mesh_id);
    if (it == c->meshes.end())
        return fail(c, FYX_ERR_UNKNOWN_ID, "mesh %llu is not registered", (unsigned long long)mesh_id);
    if (int rc = enter_primary(c)) return rc;
    FYX_HIP(c, hipStreamSynchronize(c->stream));
    free_mesh(it->second);
    c->meshes.erase(it);
    ++c->mesh_gen;
    return FYX_OK;
}

int fyx_mesh_info(fyx_ctx* c, uint64_t mesh_id, uint32_t* n_verts, uint32_t* max_bone_index,
                  uint32_t* attr_mask) {
    if (!c) return FYX_ERR_INVALID_ARG;
    const Mesh* m = find_mesh(c, mesh_id);
    if (!m) return fail(c, FYX_ERR_UNKNOWN_ID, "mesh %llu is not registered", (unsigned long long)mesh_id);
    if (n_verts) *n_verts = m->n_verts;
    if (max_bone_index) *max_bone_index = m->max_bone_index;
    if (attr_mask) *attr_mask = (m->nrm ? 1u : 0u) | (m->tan ? 2u : 0u);
    return FYX_OK;
}

int fyx_mesh_streams(fyx_ctx* c, uint64_t mesh_id, const float** d_pos, const float** d_normal,
                     const float** d_tangent, const float** d_weights, const uint32_t** d_indices) {
    if (!c) return FYX_ERR_INVALID_ARG;
    const Mesh* m = find_mesh(c, mesh_id);
    if (!m) return fail(c, FYX_ERR_UNKNOWN_ID, "mesh %llu is not registered", (unsigned long long)mesh_id);
    if (d_pos) *d_pos = m->pos;
    if (d_normal) *d_normal = m->nrm;
    if (d_tangent) *d_tangent = m->tan;
    if (d_weights) *d_weights = m->wgt;
    if (d_indices) *d_indices = m->idx;
    return FYX_OK;
}

// ---- skinning -----------------------------------------------------------------------------

int fyx_lbs_skin_device(fyx_ctx* c, uint64_t mesh_id, const float* d_palette, uint32_t n_bones,
                        uint32_t n_instances, float* d_out_pos, float* d_out_normal,
                        float* d_out_tangent) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    const Mesh* m = find_mesh(c, mesh_id);
    int rc = check_skin_args(c, m, mesh_id, d_palette, n_bones, n_instances);
    if (rc) return rc;
    if (d_out_normal && !m->nrm)
        return fail(c, FYX_ERR_MISSING_ATTRIBUTE, "mesh has no Normal attribute");
    if (d_out_tangent && !m->tan)
        return fail(c, FYX_ERR_MISSING_ATTRIBUTE, "mesh has no Tangent attribute");
    const fyx::LbsArgs a = make_args(*m, d_palette, n_bones, n_instances, d_out_pos, d_out_normal, d_out_tangent);
    hipStream_t st;
    if (int sr = acquire_launch_stream(c, &st)) return sr;
    if (c->timing) {   // this launch's own start / stop events
        if (c->timing_used + 2 > c->timing_ev.size()) {
            if (c->timing_ev.size() >= 2 * 8192) return fail(c, FYX_ERR_INVALID_ARG, "lbs.timing: read the times (fyx_debug_kernel_time) every 8192 launches");
            for (int k = 0; k < 2; ++k) {
                hipEvent_t e = nullptr;
                FYX_HIP(c, hipEventCreate(&e));
                c->timing_ev.push_back(e);
            }
        }
        fyx::LbsTuning t = c->lbs;
        t.ev_start = c->timing_ev[c->timing_used];
        t.ev_stop = c->timing_ev[c->timing_used + 1];
        const int mask = (a.out_pos ? 1 : 0) | ((a.out_nrm && a.nrm) ? 2 : 0) | ((a.out_tan && a.tan) ? 4 : 0);
        FYX_HIP(c, fyx::launch_lbs(a, t, st));
        if (mask && a.n_verts && a.n_instances) c->timing_used += 2;   // the pair is claimed only by a launch that recorded it
        return FYX_OK;
    }
    if (c->timeline_on) {
        if (int rc = timeline_arm(c, 0)) return rc;
        fyx::LbsTuning t = c->lbs;
        t.ev_start = fyx::g_launch_events.start;
        t.ev_stop = fyx::g_launch_events.stop;
        fyx::g_launch_events = fyx::LaunchEvents();
        FYX_HIP(c, fyx::launch_lbs(a, t, st));
        return FYX_OK;
    }
    FYX_HIP(c, fyx::launch_lbs(a, c->lbs, st));
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_debug_timeline(fyx_ctx* c, int32_t* kinds, double* start_us, double* stop_us, uint32_t capacity, uint32_t* n_records) {
    if (!c || !n_records) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    if (int rc = enter_primary(c)) return rc;
    FYX_HIP(c, hipStreamSynchronize(c->stream));
    uint32_t n = 0;
    hipEvent_t base = c->timeline.empty() ? nullptr : c->timeline.front().start;
    for (const fyx_ctx::TimelineRec& r : c->timeline) {
        float a = 0.f, b = 0.f;
        // a pair that never got its timestamps (a launch that launched nothing) is skipped
        if (hipEventElapsedTime(&a, base, r.start) != hipSuccess || hipEventElapsedTime(&b, base, r.stop) != hipSuccess) { (void)hipGetLastError(); continue; }
        if (n < capacity && kinds && start_us && stop_us) { kinds[n] = r.kind; start_us[n] = (double)a * 1e3; stop_us[n] = (double)b * 1e3; }
        ++n;
    }
    *n_records = n;
    for (const fyx_ctx::TimelineRec& r : c->timeline) { (void)hipEventDestroy(r.start); (void)hipEventDestroy(r.stop); }
    c->timeline.clear();
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_debug_host_times(fyx_ctx* c, double* out_us, uint32_t capacity) {
    if (!c || (capacity && !out_us)) return FYX_ERR_INVALID_ARG;
    for (uint32_t k = 0; k < capacity; ++k) out_us[k] = k < 8 ? c->host_times[k] : 0.0;
    for (double& t : c->host_times) t = 0.0;
    return FYX_OK;
}

int fyx_debug_kernel_time(fyx_ctx* c, double* total_us, uint32_t* n_launches) {
    if (!c || !total_us || !n_launches) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    if (int rc = enter_primary(c)) return rc;
    FYX_HIP(c, hipStreamSynchronize(c->stream));
    double sum = 0.0;
    uint32_t counted = 0;
    for (size_t k = 0; k + 1 < c->timing_used; k += 2) {
        float ms = 0.f;
        // a pair whose elapsed time cannot be read (never recorded) is skipped: one bad pair must not wedge the counters
        if (hipEventElapsedTime(&ms, c->timing_ev[k], c->timing_ev[k + 1]) != hipSuccess) { (void)hipGetLastError(); continue; }
        sum += (double)ms * 1e3;
        ++counted;
    }
    *total_us = sum;
    *n_launches = counted;
    c->timing_used = 0;
    // the events go back to the runtime (a measurement leg may have made thousands)
    for (hipEvent_t e : c->timing_ev) (void)hipEventDestroy(e);
    c->timing_ev.clear();
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_lbs_skin_batch(fyx_ctx* c, const fyx_skin_job* jobs, uint32_t n_jobs) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    if (n_jobs && !jobs) return fail(c, FYX_ERR_INVALID_ARG, "jobs is null");
    if (n_jobs == 0) return FYX_OK;
    const size_t key_bytes = (size_t)n_jobs * sizeof(fyx_skin_job);
    // (two cached batches: under anim.overlap a scene's frames alternate between two job arrays -- the palette buffers of the two frame streams)
    for (int e = 0; e < 2; ++e) {
        const int k = e ? 1 - c->skin_batch_last : c->skin_batch_last;      // the last one first: a scene on one stream hits it every frame
        SkinBatch* Bk = c->skin_batch[k];
        if (Bk && Bk->plan && Bk->jobs_key.size() == key_bytes && Bk->key_mesh_gen == c->mesh_gen && memcmp(&Bk->key_tuning, &c->lbs, sizeof c->lbs) == 0 &&
            memcmp(Bk->jobs_key.data(), jobs, key_bytes) == 0)
            return run_batch_plan(c, batch_entry(c, k), *static_cast<BatchPlan*>(Bk->plan), true);
    }
    SkinBatch& B = batch_entry(c, -1);
    B.jobs_key.clear();
    if (!B.plan) { B.plan = new BatchPlan(); B.plan_free = batch_plan_free; }
    BatchPlan& P = *static_cast<BatchPlan*>(B.plan);
    P = BatchPlan();
    for (uint32_t j = 0; j < n_jobs; ++j) {   // every job is validated before anything is launched
        const fyx_skin_job& J = jobs[j];
        const Mesh* m = find_mesh(c, J.mesh_id);
        if (int rc = check_skin_args(c, m, J.mesh_id, J.d_palette, J.n_bones, J.n_instances)) return rc;
        if (J.d_out_normal && !m->nrm) return fail(c, FYX_ERR_MISSING_ATTRIBUTE, "job %u: mesh has no Normal attribute", j);
        if (J.d_out_tangent && !m->tan) return fail(c, FYX_ERR_MISSING_ATTRIBUTE, "job %u: mesh has no Tangent attribute", j);
        if (int rc = P.add_soa(c, make_args(*m, J.d_palette, J.n_bones, J.n_instances, J.d_out_pos, J.d_out_normal, J.d_out_tangent))) return rc;
    }
    if (int rc = run_batch_plan(c, B, P)) return rc;
    B.jobs_key.assign(reinterpret_cast<const char*>(jobs), reinterpret_cast<const char*>(jobs) + key_bytes);
    B.key_mesh_gen = c->mesh_gen;
    B.key_tuning = c->lbs;
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_mesh_set_blend_shapes(fyx_ctx* c, uint64_t mesh_id, uint32_t n_shapes, const uint16_t* storage,
                              uint32_t plane_vertices) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    Mesh* m = find_mesh(c, mesh_id);
    if (!m) return fail(c, FYX_ERR_UNKNOWN_ID, "mesh %llu is not registered", (unsigned long long)mesh_id);
    if (n_shapes > FYX_MAX_BLEND_SHAPES)
        return fail(c, FYX_ERR_UNSUPPORTED, "%u blend shapes (the shader's weight array holds %d)", n_shapes, FYX_MAX_BLEND_SHAPES);
    if (n_shapes && !storage) return fail(c, FYX_ERR_INVALID_ARG, "storage is null");
    if (n_shapes && plane_vertices < m->n_verts)
        return fail(c, FYX_ERR_INVALID_ARG, "a %u-texel-triple plane cannot hold %u vertices", plane_vertices, m->n_verts);
    if (int jr = enter_primary(c)) return jr;
    ++c->mesh_gen;
    FYX_HIP(c, hipStreamSynchronize(c->stream));  // nothing may still read the old offsets
    if (m->shapes) { FYX_HIP(c, hipFree(m->shapes)); m->shapes = nullptr; }
    m->n_shapes = 0;
    if (n_shapes == 0 || m->n_verts == 0) { m->n_shapes = n_shapes; return FYX_OK; }
    const size_t src_bytes = (size_t)n_shapes * plane_vertices * 18;
    const uint32_t tiles = (m->n_verts + 63) / 64;
    const size_t dst_bytes = (size_t)n_shapes * tiles * 9 * 64 * 2;
    int rc = ensure_scratch(c, src_bytes);
    if (rc) return rc;
    FYX_HIP(c, hipMalloc(reinterpret_cast<void**>(&m->shapes), dst_bytes));
    FYX_HIP(c, hipMemcpyAsync(c->scratch, storage, src_bytes, hipMemcpyHostToDevice, c->stream));
    FYX_HIP(c, fyx::launch_retile_blend_shapes(static_cast<const uint16_t*>(c->scratch), m->n_verts, plane_vertices,
                                               n_shapes, m->shapes, c->stream));
    FYX_HIP(c, hipStreamSynchronize(c->stream));
    m->n_shapes = n_shapes;
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_lbs_skin_ex(fyx_ctx* c, uint64_t mesh_id, const fyx_skin_desc* d) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    fyx::LbsExArgs x;
    bool whole_spans = false;
    if (int rc = build_ex_args(c, mesh_id, d, x, whole_spans)) return rc;
    hipStream_t st;
    if (int sr = acquire_launch_stream(c, &st)) return sr;
    switch (ex_kind(x, whole_spans)) {
        case kExWholeSpans: FYX_HIP(c, fyx::launch_lbs_aos(x, c->lbs, st)); break;
        case kExPlain: FYX_HIP(c, fyx::launch_lbs(x.a, c->lbs, st)); break;
        default: FYX_HIP(c, fyx::launch_lbs_ex(x, c->lbs, st)); break;
    }
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_lbs_skin_ex_batch(fyx_ctx* c, const uint64_t* mesh_ids, const fyx_skin_desc* descs, uint32_t n_jobs) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    if (n_jobs && (!mesh_ids || !descs)) return fail(c, FYX_ERR_INVALID_ARG, "null job arrays");
    if (n_jobs == 0) return FYX_OK;
    BatchPlan P;
    for (uint32_t j = 0; j < n_jobs; ++j) {   // every job is validated before anything is launched
        fyx::LbsExArgs x;
        bool whole_spans = false;
        if (int rc = build_ex_args(c, mesh_ids[j], &descs[j], x, whole_spans)) return rc;
        int rc = FYX_OK;
        switch (ex_kind(x, whole_spans)) {
            case kExWholeSpans: rc = P.add_aos(c, x); break;
            case kExPlain: rc = P.add_soa(c, x.a); break;
            default: if (x.a.n_verts) P.general.push_back(x); break;
        }
        if (rc) return rc;
    }
    return run_batch_plan(c, batch_entry(c, -1), P);
    FYX_GUARD_END(c)
}

int fyx_lbs_skin_streams(fyx_ctx* c, uint32_t n_verts, const float* d_pos, const float* d_normal,
                         const float* d_tangent, const float* d_weights, const uint32_t* d_indices,
                         const float* d_palette, uint32_t n_bones, uint32_t n_instances,
                         float* d_out_pos, float* d_out_normal, float* d_out_tangent) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    if (!d_palette || !d_weights || !d_indices)
        return fail(c, FYX_ERR_INVALID_ARG, "palette, weights and indices are required");
    if (n_bones == 0 || n_bones > 256)
        return fail(c, FYX_ERR_INVALID_ARG, "n_bones=%u outside 1..256 (bone indices are u8)", n_bones);
    if (d_out_pos && !d_pos) return fail(c, FYX_ERR_MISSING_ATTRIBUTE, "no Position stream");
    if (d_out_normal && !d_normal) return fail(c, FYX_ERR_MISSING_ATTRIBUTE, "no Normal stream");
    if (d_out_tangent && !d_tangent) return fail(c, FYX_ERR_MISSING_ATTRIBUTE, "no Tangent stream");
    fyx::LbsArgs a;
    a.pos = d_pos; a.nrm = d_normal; a.tan = d_tangent; a.wgt = d_weights; a.idx = d_indices;
    a.palette = d_palette;
    a.out_pos = d_out_pos; a.out_nrm = d_out_normal; a.out_tan = d_out_tangent;
    a.n_verts = n_verts; a.n_bones = n_bones; a.n_instances = n_instances;
    hipStream_t st;
    if (int sr = acquire_launch_stream(c, &st)) return sr;
    FYX_HIP(c, fyx::launch_lbs(a, c->lbs, st));
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_lbs_skin(fyx_ctx* c, uint64_t mesh_id, const float* palette, uint32_t n_bones,
                 uint32_t n_instances, float* out_pos, float* out_normal, float* out_tangent,
                 float* out_aabb) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    const Mesh* m = find_mesh(c, mesh_id);
    int rc = check_skin_args(c, m, mesh_id, palette, n_bones, n_instances);
    if (rc) return rc;
    if (out_normal && !m->nrm) return fail(c, FYX_ERR_MISSING_ATTRIBUTE, "mesh has no Normal attribute");
    if (out_tangent && !m->tan) return fail(c, FYX_ERR_MISSING_ATTRIBUTE, "mesh has no Tangent attribute");
    if (int jr = enter_primary(c)) return jr;
    const size_t nv = (size_t)m->n_verts * n_instances;
    const size_t b_pal = align_up((size_t)n_bones * n_instances * 64, 256);
    const bool need_pos = out_pos || out_aabb;
    const size_t b_pos = need_pos ? align_up(nv * 12 + 64, 256) : 0;
    const size_t b_nrm = out_normal ? align_up(nv * 12 + 64, 256) : 0;
    const size_t b_tan = out_tangent ? align_up(nv * 16 + 64, 256) : 0;
    rc = ensure_scratch(c, b_pal + b_pos + b_nrm + b_tan + 64);
    if (rc) return rc;
    char* p = static_cast<char*>(c->scratch);
    float* d_pal = reinterpret_cast<float*>(p); p += b_pal;
    float* d_pos = need_pos ? reinterpret_cast<float*>(p) : nullptr; p += b_pos;
    float* d_nrm = out_normal ? reinterpret_cast<float*>(p) : nullptr; p += b_nrm;
    float* d_tan = out_tangent ? reinterpret_cast<float*>(p) : nullptr; p += b_tan;
    float* d_aabb = reinterpret_cast<float*>(p);
    FYX_HIP(c, hipMemcpyAsync(d_pal, palette, (size_t)n_bones * n_instances * 64, hipMemcpyHostToDevice, c->stream));
    const fyx::LbsArgs a = make_args(*m, d_pal, n_bones, n_instances, d_pos, d_nrm, d_tan);
    FYX_HIP(c, fyx::launch_lbs(a, c->lbs, c->stream));
    if (out_pos && nv) FYX_HIP(c, hipMemcpyAsync(out_pos, d_pos, nv * 12, hipMemcpyDeviceToHost, c->stream));
    if (out_normal && nv) FYX_HIP(c, hipMemcpyAsync(out_normal, d_nrm, nv * 12, hipMemcpyDeviceToHost, c->stream));
    if (out_tangent && nv) FYX_HIP(c, hipMemcpyAsync(out_tangent, d_tan, nv * 16, hipMemcpyDeviceToHost, c->stream));
    if (out_aabb) {
        FYX_HIP(c, fyx::launch_points_aabb(d_pos, nv, c->aabb_partials, d_aabb, c->stream));
        FYX_HIP(c, hipMemcpyAsync(out_aabb, d_aabb, 24, hipMemcpyDeviceToHost, c->stream));
    }
    FYX_HIP(c, hipStreamSynchronize(c->stream));
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_skinned_aabb(fyx_ctx* c, uint64_t mesh_id, const float* palette, uint32_t n_bones,
                     float out_aabb[6]) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    if (!out_aabb) return fail(c, FYX_ERR_INVALID_ARG, "out_aabb is null");
    const Mesh* m = find_mesh(c, mesh_id);
    int rc = check_skin_args(c, m, mesh_id, palette, n_bones, 1);
    if (rc) return rc;
    if (int jr = enter_primary(c)) return jr;
    rc = ensure_scratch(c, (size_t)n_bones * 64 + 256);
    if (rc) return rc;
    float* d_pal = static_cast<float*>(c->scratch);
    float* d_aabb = d_pal + (size_t)n_bones * 16;
    FYX_HIP(c, hipMemcpyAsync(d_pal, palette, (size_t)n_bones * 64, hipMemcpyHostToDevice, c->stream));
    const fyx::LbsArgs a = make_args(*m, d_pal, n_bones, 1, nullptr, nullptr, nullptr);
    FYX_HIP(c, fyx::launch_skinned_aabb(a, c->aabb_partials, d_aabb, c->stream));
    FYX_HIP(c, hipMemcpyAsync(out_aabb, d_aabb, 24, hipMemcpyDeviceToHost, c->stream));
    FYX_HIP(c, hipStreamSynchronize(c->stream));
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_skinned_aabb_device(fyx_ctx* c, uint64_t mesh_id, const float* d_palette, uint32_t n_bones, uint32_t n_instances,
                            float* d_out_aabb) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    if (!d_out_aabb) return fail(c, FYX_ERR_INVALID_ARG, "d_out_aabb is null");
    const Mesh* m = find_mesh(c, mesh_id);
    int rc = check_skin_args(c, m, mesh_id, d_palette, n_bones, n_instances);
    if (rc) return rc;
    if (n_instances > 65535u) return fail(c, FYX_ERR_UNSUPPORTED, "at most 65535 instances per call");
    if (int jr = enter_primary(c)) return jr;
    const uint32_t slices = fyx::aabb_inst_slices(m->n_verts, n_instances);
    float* d_partials = nullptr;
    if (slices > 1) {
        rc = ensure_scratch(c, (size_t)slices * n_instances * 24);
        if (rc) return rc;
        d_partials = static_cast<float*>(c->scratch);
    }
    const fyx::LbsArgs a = make_args(*m, d_palette, n_bones, n_instances, nullptr, nullptr, nullptr);
    FYX_HIP(c, fyx::launch_skinned_aabb_inst(a, d_partials, d_out_aabb, c->stream));
    return FYX_OK;
    FYX_GUARD_END(c)
}

// ---- calibration --------------------------------------------------------------------------

int fyx_calib_stream_copy(fyx_ctx* c, const float* d_src, float* d_dst, uint32_t units) {
    if (!c) return FYX_ERR_INVALID_ARG;
    if (units && (!d_src || !d_dst)) return fail(c, FYX_ERR_INVALID_ARG, "null pointer");
    FYX_GUARD_BEGIN
    if (int jr = enter_primary(c)) return jr;
    FYX_HIP(c, fyx::launch_stream_copy(d_src, d_dst, units, c->lbs.blocks_per_cu, c->stream));
    return FYX_OK;
    FYX_GUARD_END(c)
}

// ---- palette ------------------------------------------------------------------------------

int fyx_palette_device(fyx_ctx* c, const float* d_global, const float* d_inv_bind, uint32_t n,
                       float* d_out) {
    if (!c) return FYX_ERR_INVALID_ARG;
    if (n && (!d_global || !d_inv_bind || !d_out)) return fail(c, FYX_ERR_INVALID_ARG, "null pointer");
    FYX_GUARD_BEGIN
    if (int jr = enter_primary(c)) return jr;
    FYX_HIP(c, fyx::launch_palette(d_global, d_inv_bind, n, d_out, c->stream));
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_palette(fyx_ctx* c, const float* global, const float* inv_bind, uint32_t n, float* out) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    if (n == 0) return FYX_OK;
    if (!global || !inv_bind || !out) return fail(c, FYX_ERR_INVALID_ARG, "null pointer");
    const size_t b = (size_t)n * 64;
    if (int jr = enter_primary(c)) return jr;
    int rc = ensure_scratch(c, 3 * align_up(b, 256));
    if (rc) return rc;
    char* p = static_cast<char*>(c->scratch);
    float* dg = reinterpret_cast<float*>(p);
    float* di = reinterpret_cast<float*>(p + align_up(b, 256));
    float* dout = reinterpret_cast<float*>(p + 2 * align_up(b, 256));
    FYX_HIP(c, hipMemcpyAsync(dg, global, b, hipMemcpyHostToDevice, c->stream));
    FYX_HIP(c, hipMemcpyAsync(di, inv_bind, b, hipMemcpyHostToDevice, c->stream));
    FYX_HIP(c, fyx::launch_palette(dg, di, n, dout, c->stream));
    FYX_HIP(c, hipMemcpyAsync(out, dout, b, hipMemcpyDeviceToHost, c->stream));
    FYX_HIP(c, hipStreamSynchronize(c->stream));
    return FYX_OK;
    FYX_GUARD_END(c)
}

}  // extern "C"
