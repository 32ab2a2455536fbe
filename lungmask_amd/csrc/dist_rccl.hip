// RCCL communicator inside the engine (include/lungmask_hip.h: lm_dist_*): the one collective the slice-sharded pipeline needs --
// an equal-size all-gather of device buffers -- enqueued on the engine's own stream, so that it is ordered against the engine's
// kernels by the stream and the host never waits for it.  RCCL is bound at run time (dlopen): the library has no link-time
// dependency on it, a single-GPU user never loads it, and inside a process that already carries an RCCL (torch's) that copy
// is the one used.  Only the five entry points below are needed; their C declarations are restated here (rccl.h: ncclUniqueId is
// 128 opaque bytes, ncclResult_t 0 = success, ncclChar = 0).
#include <dlfcn.h>

#include <cstring>

#include "engine.h"

using namespace lm;

namespace {

struct NcclUniqueId {
    char internal[128];
};
typedef int (*GetUniqueIdFn)(NcclUniqueId*);
typedef int (*CommInitRankFn)(void** comm, int nranks, NcclUniqueId id, int rank);
typedef int (*AllGatherFn)(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t stream);
typedef int (*CommDestroyFn)(void* comm);
typedef const char* (*GetErrorStringFn)(int);

struct Rccl {
    void* lib = nullptr;
    GetUniqueIdFn get_unique_id = nullptr;
    CommInitRankFn comm_init_rank = nullptr;
    AllGatherFn all_gather = nullptr;
    CommDestroyFn comm_destroy = nullptr;
    GetErrorStringFn error_string = nullptr;
};

Rccl* rccl() {
#ifdef LM_EMU_BUILD
    return nullptr;
#else
    static Rccl r;
    static bool tried = false;
    if (!tried) {
        tried = true;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* n : names)  // a copy that is already mapped (torch's) first
            if ((r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD)) != nullptr) break;
        if (!r.lib)
            for (const char* n : names)
                if ((r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL)) != nullptr) break;
        if (r.lib) {
            r.get_unique_id = reinterpret_cast<GetUniqueIdFn>(dlsym(r.lib, "ncclGetUniqueId"));
            r.comm_init_rank = reinterpret_cast<CommInitRankFn>(dlsym(r.lib, "ncclCommInitRank"));
            r.all_gather = reinterpret_cast<AllGatherFn>(dlsym(r.lib, "ncclAllGather"));
            r.comm_destroy = reinterpret_cast<CommDestroyFn>(dlsym(r.lib, "ncclCommDestroy"));
            r.error_string = reinterpret_cast<GetErrorStringFn>(dlsym(r.lib, "ncclGetErrorString"));
            if (!r.get_unique_id || !r.comm_init_rank || !r.all_gather || !r.comm_destroy) r.lib = nullptr;
        }
    }
    return r.lib ? &r : nullptr;
#endif
}

int rccl_fail(const char* what, int rc) {
    Rccl* r = rccl();
    set_error("%s failed: %s", what, (r && r->error_string) ? r->error_string(rc) : "RCCL error");
    return LM_ERR_DEVICE;
}

}  // namespace

extern "C" {

int lm_dist_unique_id(uint8_t* id_out) {
    if (!id_out) return LM_ERR_INVALID;
    Rccl* r = rccl();
    if (!r) {
        set_error("lm_dist_unique_id: no RCCL library could be loaded (librccl.so)");
        return LM_ERR_DEVICE;
    }
    NcclUniqueId id;
    const int rc = r->get_unique_id(&id);
    if (rc != 0) return rccl_fail("ncclGetUniqueId", rc);
    std::memcpy(id_out, &id, sizeof id);
    return LM_OK;
}

int lm_dist_init(lm_engine* e, int rank, int world, const uint8_t* id) {
    if (!e || world < 1 || rank < 0 || rank >= world || (world > 1 && !id)) {
        set_error("lm_dist_init: bad arguments (rank %d of %d)", rank, world);
        return LM_ERR_INVALID;
    }
    if (e->dist_world != 0) {
        set_error("lm_dist_init: this engine already has a communicator (lm_dist_destroy first)");
        return LM_ERR_INVALID;
    }
    if (world > 1 || id != nullptr) {  // (a world of one WITHOUT an id needs no library: its all-gather is a copy on the engine's stream)
        Rccl* r = rccl();
        if (!r) {
            set_error("lm_dist_init: no RCCL library could be loaded (librccl.so)");
            return LM_ERR_DEVICE;
        }
        LM_HIP(hipSetDevice(e->device));
        NcclUniqueId uid;
        std::memcpy(&uid, id, sizeof uid);
        void* comm = nullptr;
        const int rc = r->comm_init_rank(&comm, world, uid, rank);
        if (rc != 0) return rccl_fail("ncclCommInitRank", rc);
        e->dist_comm = comm;
    }
    e->dist_rank = rank;
    e->dist_world = world;
    return LM_OK;
}

int lm_dist_rank(lm_engine* e) { return (e && e->dist_world) ? e->dist_rank : LM_ERR_INVALID; }
int lm_dist_world(lm_engine* e) { return (e && e->dist_world) ? e->dist_world : LM_ERR_INVALID; }

int lm_dist_all_gather(lm_engine* e, const void* send_dev, void* recv_dev, size_t bytes) {
    if (!e || e->dist_world == 0 || (bytes && (!send_dev || !recv_dev))) {
        set_error("lm_dist_all_gather: no communicator (lm_dist_init) or null buffers");
        return LM_ERR_INVALID;
    }
    if (bytes == 0) return LM_OK;
    LM_HIP(hipSetDevice(e->device));
    if (e->dist_comm == nullptr) {  // world of one, no library
        if (send_dev != recv_dev) LM_HIP(hipMemcpyAsync(recv_dev, send_dev, bytes, hipMemcpyDeviceToDevice, e->stream));
        return LM_OK;
    }
    const int rc = rccl()->all_gather(send_dev, recv_dev, bytes, /*ncclChar*/ 0, e->dist_comm, e->stream);
    if (rc != 0) return rccl_fail("ncclAllGather", rc);
    return LM_OK;
}

int lm_dist_destroy(lm_engine* e) {
    if (!e) return LM_ERR_INVALID;
    if (e->dist_comm) {
        (void)hipSetDevice(e->device);
        (void)hipStreamSynchronize(e->stream);
        Rccl* r = rccl();
        if (r) (void)r->comm_destroy(e->dist_comm);
        e->dist_comm = nullptr;
    }
    e->dist_world = 0;
    e->dist_rank = 0;
    return LM_OK;
}

}  // extern "C"
