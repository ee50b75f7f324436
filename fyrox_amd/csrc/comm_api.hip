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
    if (!k.get_unique_id || !k.comm_init_rank || !k.comm_destroy || !k.all_gather) {
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
    if (c->device < 0) return fail(c, FYX_ERR_NO_DEVICE, "control-only context");
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

}  // extern "C"
