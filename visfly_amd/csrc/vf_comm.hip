// vf_comm.hip -- the one exchange step of the path: sum of the flat gradient buffer over the ranks (RCCL over xGMI).
//
// Replaces what torch.distributed / DDP would do around optimizer.step() if the reference's PPO / BPTT loops
// (utils/algorithms/PPO.py:285-292, BPTT.py:127-134) ran data-parallel: ONE ncclAllReduce of the flat fp32 gradient (+ the
// loss statistics riding in its tail) per optimiser step, enqueued on the caller's stream straight from C -- no Python
// dispatcher, no side stream, no event pair per call.  torch.distributed stays the bootstrap: it carries rank 0's
// ncclUniqueId to the other ranks (visfly_amd/parallel.py).
//
// RCCL is resolved at run time from the library the process already maps (torch ships its own librccl.so; linking a
// second copy would give two runtimes that cannot share a communicator), see vf_comm_library().
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>

#include "vf_common.hpp"

struct vf_comm {
    ncclComm_t comm = nullptr;
    int world = 0, rank = 0;
};

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl& rccl()
{
    static Rccl r;
    return r;
}

int need_library(const char* who)
{
    if (!rccl().handle) return vf::fail(VF_ESTATE, "%s: vf_comm_library() has not been called", who);
    return VF_OK;
}

int nccl_fail(const char* what, ncclResult_t rc)
{
    return vf::fail(VF_EHIP, "%s failed: %s", what, rccl().GetErrorString ? rccl().GetErrorString(rc) : "?");
}

}  // namespace

extern "C" {

int vf_comm_library(const char* path)
{
    Rccl& r = rccl();
    if (r.handle) return VF_OK;
    if (!path) return vf::fail(VF_EINVAL, "vf_comm_library: null path");
    void* h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
    if (!h) return vf::fail(VF_EINVAL, "vf_comm_library: dlopen(%s): %s", path, dlerror());
#define VF_SYM(field, name)                                                                         \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, name));                                  \
    if (!r.field) return vf::fail(VF_EINVAL, "vf_comm_library: %s has no symbol %s", path, name);
    VF_SYM(GetUniqueId, "ncclGetUniqueId")
    VF_SYM(CommInitRank, "ncclCommInitRank")
    VF_SYM(AllReduce, "ncclAllReduce")
    VF_SYM(CommDestroy, "ncclCommDestroy")
    VF_SYM(GetErrorString, "ncclGetErrorString")
#undef VF_SYM
    r.handle = h;
    return VF_OK;
}

int vf_comm_unique_id(uint8_t* id)
{
    if (int rc = need_library("vf_comm_unique_id")) return rc;
    if (!id) return vf::fail(VF_EINVAL, "vf_comm_unique_id: null argument");
    ncclUniqueId u;
    ncclResult_t rc = rccl().GetUniqueId(&u);
    if (rc != ncclSuccess) return nccl_fail("ncclGetUniqueId", rc);
    static_assert(sizeof(u) == VF_COMM_ID_BYTES, "ncclUniqueId size");
    memcpy(id, &u, sizeof(u));
    return VF_OK;
}

int vf_comm_init(const uint8_t* id, int32_t world, int32_t rank, vf_comm** out)
{
    if (int rc = need_library("vf_comm_init")) return rc;
    if (!id || !out || world < 1 || rank < 0 || rank >= world) return vf::fail(VF_EINVAL, "vf_comm_init: bad argument");
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    vf_comm* c = new vf_comm;
    c->world = world;
    c->rank = rank;
    ncclResult_t rc = rccl().CommInitRank(&c->comm, world, u, rank);   // collective over the ranks; uses the current HIP device
    if (rc != ncclSuccess) {
        delete c;
        return nccl_fail("ncclCommInitRank", rc);
    }
    *out = c;
    return VF_OK;
}

int vf_allreduce_grads(vf_comm* c, float* buf, int64_t n, vf_stream_t stream)
{
    if (!c || !c->comm) return vf::fail(VF_EINVAL, "vf_allreduce_grads: null communicator");
    if (!buf || n <= 0) return vf::fail(VF_EINVAL, "vf_allreduce_grads: null buffer or n <= 0");
    ncclResult_t rc = rccl().AllReduce(buf, buf, (size_t)n, ncclFloat, ncclSum, c->comm, vf::as_stream(stream));
    if (rc != ncclSuccess) return nccl_fail("ncclAllReduce", rc);
    return VF_OK;
}

int vf_allreduce_f64(vf_comm* c, double* buf, int64_t n, vf_stream_t stream)
{
    if (!c || !c->comm) return vf::fail(VF_EINVAL, "vf_allreduce_f64: null communicator");
    if (!buf || n <= 0) return vf::fail(VF_EINVAL, "vf_allreduce_f64: null buffer or n <= 0");
    ncclResult_t rc = rccl().AllReduce(buf, buf, (size_t)n, ncclDouble, ncclSum, c->comm, vf::as_stream(stream));
    if (rc != ncclSuccess) return nccl_fail("ncclAllReduce", rc);
    return VF_OK;
}

void vf_comm_destroy(vf_comm* c)
{
    if (!c) return;
    if (c->comm && rccl().CommDestroy) (void)rccl().CommDestroy(c->comm);
    delete c;
}

}  // extern "C"
