// The one exchange of the data-parallel training step -- the sum of the flat gradient over the ranks (SURVEY.md section 8 e) --
// directly on RCCL: ncclAllReduce on the ctx stream, no torch tensor and no host synchronisation between the backward pass and
// Adam.  librccl.so is opened at run time (dlopen): inference-only deployments and CPU-side tools do not need it, and
// libsr_hip.so keeps no link-time dependency on it.  The host program only has to carry the 128-byte communicator id from rank 0
// to the others (dp.py does it over torch.distributed, which it has for the launch contract anyway).
#include "sr_internal.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>

using namespace srh;

namespace {

struct Rccl {
    void* so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.so = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.so) break;
        }
        if (!r.so) {
            r.err = std::string("librccl.so could not be opened: ") + (dlerror() ? dlerror() : "?");
            return;
        }
        r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.so, "ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.so, "ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.so, "ncclCommDestroy");
        r.AllReduce = (decltype(r.AllReduce))dlsym(r.so, "ncclAllReduce");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.so, "ncclGetErrorString");
        if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce || !r.GetErrorString) r.err = "librccl.so lacks an expected symbol";
    });
    return &r;
}

int rfail(sr_ctx* c, const char* what, ncclResult_t e) { return fail(c, SR_ERR_HIP, "%s: %s", what, rccl()->GetErrorString ? rccl()->GetErrorString(e) : "RCCL error"); }

}  // namespace

namespace srh {

int rccl_allreduce_f32(sr_ctx* c, void* comm, float* d_buf, size_t n, hipStream_t st) {
    Rccl* r = rccl();
    if (!r->err.empty()) return fail(c, SR_ERR_HIP, "%s", r->err.c_str());
    ncclResult_t e = r->AllReduce(d_buf, d_buf, n, ncclFloat32, ncclSum, (ncclComm_t)comm, st);
    return e == ncclSuccess ? SR_OK : rfail(c, "ncclAllReduce", e);
}

}  // namespace srh

extern "C" {

int sr_rccl_unique_id(unsigned char* id128) {
    if (!id128) return fail(nullptr, SR_ERR_ARG, "id is NULL");
    Rccl* r = rccl();
    if (!r->err.empty()) return fail(nullptr, SR_ERR_HIP, "%s", r->err.c_str());
    ncclUniqueId id;
    ncclResult_t e = r->GetUniqueId(&id);
    if (e != ncclSuccess) return rfail(nullptr, "ncclGetUniqueId", e);
    memcpy(id128, id.internal, NCCL_UNIQUE_ID_BYTES);
    return SR_OK;
}

int sr_rccl_comm_create(void** comm, const unsigned char* id128, int rank, int nranks, int device) {
    if (!comm || !id128) return fail(nullptr, SR_ERR_ARG, "NULL argument");
    *comm = nullptr;
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(nullptr, SR_ERR_ARG, "rank %d of %d", rank, nranks);
    Rccl* r = rccl();
    if (!r->err.empty()) return fail(nullptr, SR_ERR_HIP, "%s", r->err.c_str());
    HIPCHK(nullptr, hipSetDevice(device));
    ncclUniqueId id;
    memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t cm = nullptr;
    ncclResult_t e = r->CommInitRank(&cm, nranks, id, rank);
    if (e != ncclSuccess) return rfail(nullptr, "ncclCommInitRank", e);
    *comm = cm;
    return SR_OK;
}

int sr_rccl_comm_destroy(void* comm) {
    if (!comm) return SR_OK;
    Rccl* r = rccl();
    if (!r->err.empty()) return fail(nullptr, SR_ERR_HIP, "%s", r->err.c_str());
    ncclResult_t e = r->CommDestroy((ncclComm_t)comm);
    return e == ncclSuccess ? SR_OK : rfail(nullptr, "ncclCommDestroy", e);
}

int sr_rccl_allreduce_f32(void* comm, float* d_buf, int64_t n, void* hip_stream) {
    if (!comm || !d_buf || n < 1) return fail(nullptr, SR_ERR_ARG, "bad argument");
    return rccl_allreduce_f32(nullptr, comm, d_buf, (size_t)n, (hipStream_t)hip_stream);
}

}  // extern "C"
