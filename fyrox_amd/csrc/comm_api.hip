// comm_api.hip -- the one exchange step of the path: an RCCL all-gather of the skinned streams when a scene is
// sharded by vertex range over several GPUs and a consumer needs the whole buffer on every GPU.
// One process (or thread) per GPU, each with its own fyx_ctx; ranks exchange the 128-byte unique id through the
// host application (as they would exchange any start-up datum).  librccl.so is opened on first use, so the
// library has no RCCL dependency for single-GPU use.
#include <dlfcn.h>

#include "fyx_ctx.h"

namespace fyx {

struct RcclId { char internal[FYX_COMM_ID_BYTES]; };   // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef void* RcclComm;

struct Comm {
    void* lib = nullptr;
    int (*get_unique_id)(RcclId*) = nullptr;
    int (*comm_init_rank)(RcclComm*, int, RcclId, int) = nullptr;
    int (*comm_destroy)(RcclComm) = nullptr;
    int (*all_gather)(const void*, void*, size_t, int, RcclComm, hipStream_t) = nullptr;
    int (*broadcast)(const void*, void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
    int (*send)(const void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;     // optional (exchange form 1)
    int (*recv)(void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
    int (*group_start)() = nullptr;
    int (*group_end)() = nullptr;
    int (*comm_count)(RcclComm, int*) = nullptr;
    int (*comm_user_rank)(RcclComm, int*) = nullptr;
    const char* (*get_error_string)(int) = nullptr;
    RcclComm comm = nullptr;
    int rank = 0, n_ranks = 0;
};

namespace {

int load_rccl(fyx_ctx* c, Comm& k) {
    if (k.lib) return FYX_OK;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail(c, FYX_ERR_UNSUPPORTED, "librccl.so not found: %s", dlerror());
    k.get_unique_id = reinterpret_cast<int (*)(RcclId*)>(dlsym(h, "ncclGetUniqueId"));
    k.comm_init_rank = reinterpret_cast<int (*)(RcclComm*, int, RcclId, int)>(dlsym(h, "ncclCommInitRank"));
    k.comm_destroy = reinterpret_cast<int (*)(RcclComm)>(dlsym(h, "ncclCommDestroy"));
    k.all_gather = reinterpret_cast<int (*)(const void*, void*, size_t, int, RcclComm, hipStream_t)>(dlsym(h, "ncclAllGather"));
    k.get_error_string = reinterpret_cast<const char* (*)(int)>(dlsym(h, "ncclGetErrorString"));
    k.broadcast = reinterpret_cast<int (*)(const void*, void*, size_t, int, int, RcclComm, hipStream_t)>(dlsym(h, "ncclBroadcast"));
    k.send = reinterpret_cast<int (*)(const void*, size_t, int, int, RcclComm, hipStream_t)>(dlsym(h, "ncclSend"));
    k.recv = reinterpret_cast<int (*)(void*, size_t, int, int, RcclComm, hipStream_t)>(dlsym(h, "ncclRecv"));
    k.group_start = reinterpret_cast<int (*)()>(dlsym(h, "ncclGroupStart"));
    k.group_end = reinterpret_cast<int (*)()>(dlsym(h, "ncclGroupEnd"));
    k.comm_count = reinterpret_cast<int (*)(RcclComm, int*)>(dlsym(h, "ncclCommCount"));
    k.comm_user_rank = reinterpret_cast<int (*)(RcclComm, int*)>(dlsym(h, "ncclCommUserRank"));
    if (!k.get_unique_id || !k.comm_init_rank || !k.comm_destroy || !k.all_gather || !k.broadcast || !k.group_start ||
        !k.group_end || !k.comm_count || !k.comm_user_rank) {
        dlclose(h);
        return fail(c, FYX_ERR_UNSUPPORTED, "librccl.so lacks the expected entry points");
    }
    k.lib = h;
    return FYX_OK;
}

Comm& comm_of(fyx_ctx* c) {
    if (!c->comm) c->comm = new Comm();
    return *c->comm;
}

int rccl_fail(fyx_ctx* c, const Comm& k, int rc, const char* what) {
    return fail(c, FYX_ERR_HIP, "%s: %s (%d)", what, k.get_error_string ? k.get_error_string(rc) : "RCCL error", rc);
}

// [g * groups / n_ranks] groups of kShardAlign vertices each: the same cut as fyrox_amd/sharding.py::vertex_range
constexpr uint32_t kShardAlign = 256;
uint32_t shard_cut(uint32_t n_verts, uint64_t g, uint64_t n_ranks) {
    const uint64_t groups = ((uint64_t)n_verts + kShardAlign - 1) / kShardAlign;
    const uint64_t v = (g * groups / n_ranks) * kShardAlign;
    return v < n_verts ? (uint32_t)v : n_verts;
}

// The PADDED cut of exchange form 2: every rank's shard is the same S = ceil(groups / n_ranks) * kShardAlign vertices long -- rank r
// owns [r S, min((r + 1) S, n_verts)) -- and every GPU's buffers hold n_ranks * S vertices, so that ONE in-place ncclAllGather per
// stream (sendbuff = recvbuff + rank * count: RCCL's best-tuned collective) moves everything.  1 M vertices over 8 GPUs:
// S = 125 184, 1 001 472 vertices of buffer, the last rank skins 123 712.
uint32_t padded_shard(uint32_t n_verts, uint64_t n_ranks) {
    const uint64_t groups = ((uint64_t)n_verts + kShardAlign - 1) / kShardAlign;
    return (uint32_t)(((groups + n_ranks - 1) / n_ranks) * kShardAlign);
}

// The padded form moves n_ranks * shard vertices per stream: it exists only behind entry points that are told what the buffers hold
// (fyx_allgather_skinned_padded*).  The ragged entry points refuse comm.form = 2 -- an option must not change how much a call writes.
int padded_form_check(fyx_ctx* c, bool padded, uint32_t n_verts, uint32_t capacity_verts, int n_ranks) {
    if (!padded) {
        if (c->comm_form == 2)
            return fail(c, FYX_ERR_INVALID_ARG, "comm.form = 2 (one all-gather over padded shards) writes n_ranks * shard_verts vertices per stream: "
                                                 "call fyx_allgather_skinned_padded[_all], which takes the buffers' capacity");
        return FYX_OK;
    }
    const uint64_t need = (uint64_t)padded_shard(n_verts, (uint64_t)n_ranks) * (uint64_t)n_ranks;
    if (need > 0xffffffffull) return fail(c, FYX_ERR_UNSUPPORTED, "n_ranks * shard_verts = %llu does not fit 32 bits", (unsigned long long)need);
    if ((uint64_t)capacity_verts < need)
        return fail(c, FYX_ERR_INVALID_ARG, "the padded exchange of %u vertices over %d ranks writes %llu vertices per stream; the buffers hold %u "
                                             "(fyx_shard_vertex_range_padded: n_ranks * shard_verts)", n_verts, n_ranks, (unsigned long long)need, capacity_verts);
    return FYX_OK;
}

// One rank's calls for one stream of the exchange, inside an open RCCL group.  `base` is that rank's full buffer, `me` its rank.
//   form 0: one broadcast per shard, in place (root = the shard's owner);
//   form 1: the rank sends its own shard to every other rank and receives every other shard where it belongs -- point to point,
//           what RCCL turns into one fused send/recv kernel over the xGMI links (no root, no tree).
int enqueue_stream_exchange(const Comm& k, int form, float* base, uint32_t width, uint32_t n_verts, int me, int n_ranks, hipStream_t st) {
    constexpr int kNcclFloat32 = 7;   // rccl.h: ncclFloat32
    if (form == 2) {     // equal padded shards: one in-place all-gather
        const size_t count = (size_t)padded_shard(n_verts, (uint64_t)n_ranks) * width;
        return k.all_gather(base + (size_t)me * count, base, count, kNcclFloat32, k.comm, st);
    }
    const uint32_t mb = shard_cut(n_verts, (uint64_t)me, (uint64_t)n_ranks), me_e = shard_cut(n_verts, (uint64_t)me + 1, (uint64_t)n_ranks);
    for (int r = 0; r < n_ranks; ++r) {
        const uint32_t b = shard_cut(n_verts, (uint64_t)r, (uint64_t)n_ranks), e = shard_cut(n_verts, (uint64_t)r + 1, (uint64_t)n_ranks);
        float* at = base + (size_t)b * width;      // rank r's shard: sent from there by r, received there by all
        if (form == 0) {
            if (e == b) continue;
            if (int rc = k.broadcast(at, at, (size_t)(e - b) * width, kNcclFloat32, r, k.comm, st)) return rc;
        } else if (r != me) {
            if (me_e > mb)
                if (int rc = k.send(base + (size_t)mb * width, (size_t)(me_e - mb) * width, kNcclFloat32, r, k.comm, st)) return rc;
            if (e > b)
                if (int rc = k.recv(at, (size_t)(e - b) * width, kNcclFloat32, r, k.comm, st)) return rc;
        }
    }
    return 0;
}

}  // namespace

void comm_destroy(Comm* k) {
    if (!k) return;
    if (k->comm && k->comm_destroy) (void)k->comm_destroy(k->comm);
    // librccl stays loaded: unloading a library with live HIP state at process exit is not worth the risk
    delete k;
}

}  // namespace fyx

using namespace fyx;

extern "C" {

int fyx_comm_unique_id(fyx_ctx* c, uint8_t out_id[FYX_COMM_ID_BYTES]) {
    if (!c || !out_id) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    if (int rc = bind_device(c)) return rc;
    Comm& k = comm_of(c);
    if (int rc = load_rccl(c, k)) return rc;
    RcclId id;
    const int rc = k.get_unique_id(&id);
    if (rc) return rccl_fail(c, k, rc, "ncclGetUniqueId");
    memcpy(out_id, id.internal, FYX_COMM_ID_BYTES);
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_comm_init(fyx_ctx* c, const uint8_t id[FYX_COMM_ID_BYTES], int rank, int n_ranks) {
    if (!c || !id) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    if (c->device < 0) return fail(c, FYX_ERR_NO_DEVICE, "control-only context");
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(c, FYX_ERR_INVALID_ARG, "rank %d of %d", rank, n_ranks);
    Comm& k = comm_of(c);
    if (k.comm) return fail(c, FYX_ERR_INVALID_ARG, "this context already has a communicator");
    if (int rc = load_rccl(c, k)) return rc;
    FYX_HIP(c, hipSetDevice(c->device));
    RcclId rid;
    memcpy(rid.internal, id, FYX_COMM_ID_BYTES);
    const int rc = k.comm_init_rank(&k.comm, n_ranks, rid, rank);
    if (rc) { k.comm = nullptr; return rccl_fail(c, k, rc, "ncclCommInitRank"); }
    k.rank = rank;
    k.n_ranks = n_ranks;
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_comm_shutdown(fyx_ctx* c) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    if (!c->comm || !c->comm->comm) return FYX_OK;
    if (int rc = enter_primary(c)) return rc;
    FYX_HIP(c, hipStreamSynchronize(c->stream));
    const int rc = c->comm->comm_destroy(c->comm->comm);
    c->comm->comm = nullptr;
    if (rc) return rccl_fail(c, *c->comm, rc, "ncclCommDestroy");
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_allgather_f32(fyx_ctx* c, const float* d_send, size_t count, float* d_recv) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    if (!c->comm || !c->comm->comm) return fail(c, FYX_ERR_INVALID_ARG, "no communicator: call fyx_comm_init first");
    if (count && (!d_send || !d_recv)) return fail(c, FYX_ERR_INVALID_ARG, "null buffer");
    if (count == 0) return FYX_OK;
    // on the context stream, after every skinning launch in flight (the shard must be complete before it is sent)
    if (int rc = enter_primary(c)) return rc;
    constexpr int kNcclFloat32 = 7;   // rccl.h: ncclFloat32
    const int rc = c->comm->all_gather(d_send, d_recv, count, kNcclFloat32, c->comm->comm, c->stream);
    if (rc) return rccl_fail(c, *c->comm, rc, "ncclAllGather");
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_shard_vertex_range(uint32_t n_verts, int rank, int n_ranks, uint32_t* begin, uint32_t* end) {
    if (!begin || !end || n_ranks < 1 || rank < 0 || rank >= n_ranks) return FYX_ERR_INVALID_ARG;
    *begin = shard_cut(n_verts, (uint64_t)rank, (uint64_t)n_ranks);
    *end = shard_cut(n_verts, (uint64_t)rank + 1, (uint64_t)n_ranks);
    return FYX_OK;
}

int fyx_shard_vertex_range_padded(uint32_t n_verts, int rank, int n_ranks, uint32_t* begin, uint32_t* end, uint32_t* shard_verts) {
    if (!begin || !end || !shard_verts || n_ranks < 1 || rank < 0 || rank >= n_ranks) return FYX_ERR_INVALID_ARG;
    const uint64_t S = padded_shard(n_verts, (uint64_t)n_ranks);
    if (S * (uint64_t)n_ranks > 0xffffffffull) return FYX_ERR_UNSUPPORTED;
    const uint64_t b = S * (uint64_t)rank, e = b + S;
    *begin = (uint32_t)(b < n_verts ? b : n_verts);
    *end = (uint32_t)(e < n_verts ? e : n_verts);
    *shard_verts = (uint32_t)S;
    return FYX_OK;
}

int fyx_comm_info(fyx_ctx* c, int* rank, int* n_ranks) {
    if (!c || !rank || !n_ranks) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    if (!c->comm || !c->comm->comm) return fail(c, FYX_ERR_INVALID_ARG, "no communicator: call fyx_comm_init first");
    Comm& k = *c->comm;
    int rc = k.comm_user_rank(k.comm, rank);
    if (rc) return rccl_fail(c, k, rc, "ncclCommUserRank");
    rc = k.comm_count(k.comm, n_ranks);
    if (rc) return rccl_fail(c, k, rc, "ncclCommCount");
    return FYX_OK;
    FYX_GUARD_END(c)
}

// capacity_verts: vertices every d_*_all holds (the padded form writes n_ranks * shard_verts of them); 0: the ragged forms of the option
static int allgather_one(fyx_ctx* c, uint32_t n_verts, uint32_t capacity_verts, bool padded, float* d_pos_all, float* d_normal_all, float* d_tangent_all) {
    {
    if (!c->comm || !c->comm->comm) return fail(c, FYX_ERR_INVALID_ARG, "no communicator: call fyx_comm_init first");
    if (n_verts == 0 || (!d_pos_all && !d_normal_all && !d_tangent_all)) return FYX_OK;
    Comm& k = *c->comm;
    if (int rc = padded_form_check(c, padded, n_verts, capacity_verts, k.n_ranks)) return rc;
    // on the context stream, after every skinning launch in flight (the shard must be complete before it is sent)
    if (int rc = enter_primary(c)) return rc;
    const int form = padded ? 2 : c->comm_form;
    if (form == 1 && (!k.send || !k.recv)) return fail(c, FYX_ERR_UNSUPPORTED, "comm.form=1 needs ncclSend / ncclRecv, which this librccl lacks");
    struct { float* p; uint32_t width; } streams[3] = {{d_pos_all, 3}, {d_normal_all, 3}, {d_tangent_all, 4}};
    int rc = k.group_start();
    if (rc) return rccl_fail(c, k, rc, "ncclGroupStart");
    int first_err = 0;
    for (const auto& s : streams) {
        if (!s.p || first_err) continue;
        first_err = enqueue_stream_exchange(k, form, s.p, s.width, n_verts, k.rank, k.n_ranks, c->stream);
    }
    rc = k.group_end();     // always closed, also after a failed call inside the group
    if (first_err) return rccl_fail(c, k, first_err, form == 2 ? "ncclAllGather" : form ? "ncclSend / ncclRecv" : "ncclBroadcast");
    if (rc) return rccl_fail(c, k, rc, "ncclGroupEnd");
    return FYX_OK;
    }
}

int fyx_allgather_skinned(fyx_ctx* c, uint32_t n_verts, float* d_pos_all, float* d_normal_all, float* d_tangent_all) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    return allgather_one(c, n_verts, 0, false, d_pos_all, d_normal_all, d_tangent_all);
    FYX_GUARD_END(c)
}

int fyx_allgather_skinned_padded(fyx_ctx* c, uint32_t n_verts, uint32_t capacity_verts, float* d_pos_all, float* d_normal_all, float* d_tangent_all) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    return allgather_one(c, n_verts, capacity_verts, true, d_pos_all, d_normal_all, d_tangent_all);
    FYX_GUARD_END(c)
}

// ---- one process, several GPUs: every collective of the process is issued by one thread inside one RCCL group ----

int fyx_comm_init_all(fyx_ctx* const* ctxs, int n) {
    if (!ctxs || n < 1) return FYX_ERR_INVALID_ARG;
    for (int i = 0; i < n; ++i)
        if (!ctxs[i]) return FYX_ERR_INVALID_ARG;
    fyx_ctx* c = ctxs[0];
    FYX_GUARD_BEGIN
    for (int i = 0; i < n; ++i) {
        if (ctxs[i]->device < 0) return fail(c, FYX_ERR_NO_DEVICE, "context %d is control-only", i);
        if (ctxs[i]->comm && ctxs[i]->comm->comm) return fail(c, FYX_ERR_INVALID_ARG, "context %d already has a communicator", i);
        for (int j = 0; j < i; ++j)
            if (ctxs[j] == ctxs[i] || ctxs[j]->device == ctxs[i]->device)
                return fail(c, FYX_ERR_INVALID_ARG, "contexts %d and %d are on the same GPU (one rank per GPU)", j, i);
    }
    Comm& k0 = comm_of(c);
    if (int rc = load_rccl(c, k0)) return rc;
    for (int i = 1; i < n; ++i)
        if (int rc = load_rccl(ctxs[i], comm_of(ctxs[i]))) return fail(c, rc, "context %d: %s", i, ctxs[i]->err.c_str());
    FYX_HIP(c, hipSetDevice(c->device));
    RcclId id;
    int rc = k0.get_unique_id(&id);
    if (rc) return rccl_fail(c, k0, rc, "ncclGetUniqueId");
    rc = k0.group_start();
    if (rc) return rccl_fail(c, k0, rc, "ncclGroupStart");
    int first_err = 0;
    for (int i = 0; i < n && !first_err; ++i) {
        Comm& k = *ctxs[i]->comm;
        if (hipSetDevice(ctxs[i]->device) != hipSuccess) { first_err = -1; break; }
        first_err = k.comm_init_rank(&k.comm, n, id, i);
    }
    rc = k0.group_end();
    (void)hipSetDevice(c->device);
    if (first_err || rc) {
        for (int i = 0; i < n; ++i) {      // no half-made communicator is left behind
            Comm& k = *ctxs[i]->comm;
            if (k.comm) { (void)k.comm_destroy(k.comm); k.comm = nullptr; }
        }
        if (first_err == -1) return fail(c, FYX_ERR_HIP, "hipSetDevice failed while joining the communicator");
        return rccl_fail(c, k0, first_err ? first_err : rc, first_err ? "ncclCommInitRank" : "ncclGroupEnd");
    }
    for (int i = 0; i < n; ++i) { ctxs[i]->comm->rank = i; ctxs[i]->comm->n_ranks = n; }
    return FYX_OK;
    FYX_GUARD_END(c)
}

static int allgather_all(fyx_ctx* const* ctxs, int n, uint32_t n_verts, uint32_t capacity_verts, bool padded, float* const* d_pos_all,
                         float* const* d_normal_all, float* const* d_tangent_all) {
    if (!ctxs || n < 1) return FYX_ERR_INVALID_ARG;
    for (int i = 0; i < n; ++i)
        if (!ctxs[i]) return FYX_ERR_INVALID_ARG;
    fyx_ctx* c = ctxs[0];
    FYX_GUARD_BEGIN
    for (int i = 0; i < n; ++i) {
        const Comm* k = ctxs[i]->comm;
        if (!k || !k->comm) return fail(c, FYX_ERR_INVALID_ARG, "context %d has no communicator: call fyx_comm_init_all first", i);
        if (k->n_ranks != n || k->rank != i) return fail(c, FYX_ERR_INVALID_ARG, "context %d is rank %d of %d: pass the contexts of fyx_comm_init_all, in order", i, k->rank, k->n_ranks);
    }
    if (n_verts == 0 || (!d_pos_all && !d_normal_all && !d_tangent_all)) return FYX_OK;
    if (int rc = padded_form_check(c, padded, n_verts, capacity_verts, n)) return rc;
    float* const* sets[3] = {d_pos_all, d_normal_all, d_tangent_all};
    for (int s = 0; s < 3; ++s)
        if (sets[s])
            for (int i = 0; i < n; ++i)
                if (!sets[s][i]) return fail(c, FYX_ERR_INVALID_ARG, "stream %d is null on context %d (every GPU holds the same set of streams)", s, i);
    // every GPU's shard must be complete before it is sent: the GPU-side join of its skinning launches, on its own stream
    for (int i = 0; i < n; ++i)
        if (int rc = enter_primary(ctxs[i])) return i == 0 ? rc : fail(c, rc, "context %d: %s", i, ctxs[i]->err.c_str());
    Comm& k0 = *c->comm;
    const int form = padded ? 2 : c->comm_form;
    if (form == 1 && (!k0.send || !k0.recv)) return fail(c, FYX_ERR_UNSUPPORTED, "comm.form=1 needs ncclSend / ncclRecv, which this librccl lacks");
    const uint32_t widths[3] = {3, 3, 4};
    int rc = k0.group_start();
    if (rc) return rccl_fail(c, k0, rc, "ncclGroupStart");
    int first_err = 0;
    for (int i = 0; i < n && !first_err; ++i) {
        Comm& k = *ctxs[i]->comm;
        if (hipSetDevice(ctxs[i]->device) != hipSuccess) { first_err = -1; break; }
        for (int s = 0; s < 3 && !first_err; ++s) {
            if (!sets[s]) continue;
            first_err = enqueue_stream_exchange(k, form, sets[s][i], widths[s], n_verts, i, n, ctxs[i]->stream);
        }
    }
    rc = k0.group_end();
    (void)hipSetDevice(c->device);
    if (first_err == -1) return fail(c, FYX_ERR_HIP, "hipSetDevice failed inside the exchange");
    if (first_err) return rccl_fail(c, k0, first_err, form == 2 ? "ncclAllGather" : form ? "ncclSend / ncclRecv" : "ncclBroadcast");
    if (rc) return rccl_fail(c, k0, rc, "ncclGroupEnd");
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_allgather_skinned_all(fyx_ctx* const* ctxs, int n, uint32_t n_verts, float* const* d_pos_all, float* const* d_normal_all,
                              float* const* d_tangent_all) {
    return allgather_all(ctxs, n, n_verts, 0, false, d_pos_all, d_normal_all, d_tangent_all);
}

int fyx_allgather_skinned_padded_all(fyx_ctx* const* ctxs, int n, uint32_t n_verts, uint32_t capacity_verts, float* const* d_pos_all,
                                     float* const* d_normal_all, float* const* d_tangent_all) {
    return allgather_all(ctxs, n, n_verts, capacity_verts, true, d_pos_all, d_normal_all, d_tangent_all);
}

}  // extern "C"
