// Multi-GPU exchange step of the sharded align: source points are split into
// contiguous ranges, one per rank (one process per GPU), the map is replicated,
// and every residual evaluation ends in ONE collective over xGMI: an
// all-gather of LOM_NSUMS f64 per rank, summed in rank order on the host so
// the result does not depend on the collective's internal tree.
//
// The reference has no counterpart (single process, src/voxel_grid.h:217
// std::execution::par is its only parallelism).
//
// RCCL (573 MB) is loaded lazily with dlopen the first time a communicator is
// requested, so single-GPU users never pay for it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>

#include "lom_internal.hpp"

namespace {

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl &rccl()
{
    static Rccl r;
    if (r.lib) return r;
    // reuse a copy the process already holds (e.g. the one torch loaded), else load ROCm's
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
        if (r.lib) break;
    }
    for (int i = 0; !r.lib && i < 3; i++) r.lib = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!r.lib) return r;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.lib, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.lib, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))dlsym(r.lib, "ncclAllGather");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.GetErrorString;
    return r;
}

int comm_error(lom_map *m, const char *what, ncclResult_t e)
{
    std::string s = what;
    s += ": ";
    s += rccl().GetErrorString ? rccl().GetErrorString(e) : "rccl error";
    return lom::set_error(m, LOM_ERR_COMM, s.c_str());
}

}  // namespace

namespace lom {

int comm_allgather_sums(lom_map *m, const double *d_send, double *d_recv, int count)
{
    Rccl &r = rccl();
    if (!r.ok || !m->comm) return set_error(m, LOM_ERR_COMM, "communicator not initialised");
    const ncclResult_t e = r.AllGather(d_send, d_recv, (size_t)count, ncclFloat64, (ncclComm_t)m->comm, m->stream);
    if (e != ncclSuccess) return comm_error(m, "ncclAllGather", e);
    return LOM_OK;
}

}  // namespace lom

extern "C" {

int lom_comm_unique_id(char id_out[LOM_COMM_ID_BYTES])
{
    static_assert(LOM_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    if (!id_out) return LOM_ERR_ARG;
    Rccl &r = rccl();
    if (!r.ok) return lom::set_error(nullptr, LOM_ERR_COMM, "librccl.so could not be loaded");
    ncclUniqueId id;
    const ncclResult_t e = r.GetUniqueId(&id);
    if (e != ncclSuccess) return comm_error(nullptr, "ncclGetUniqueId", e);
    std::memcpy(id_out, id.internal, LOM_COMM_ID_BYTES);
    return LOM_OK;
}

int lom_comm_init(lom_map *m, int rank, int nranks, const char id_in[LOM_COMM_ID_BYTES])
{
    if (!m || !id_in || nranks < 1 || rank < 0 || rank >= nranks) return LOM_ERR_ARG;
    if (m->comm) return lom::set_error(m, LOM_ERR_STATE, "communicator already initialised");
    Rccl &r = rccl();
    if (!r.ok) return lom::set_error(m, LOM_ERR_COMM, "librccl.so could not be loaded");
    LOM_HIP(m, hipSetDevice(m->device));
    ncclUniqueId id;
    std::memcpy(id.internal, id_in, LOM_COMM_ID_BYTES);
    ncclComm_t c = nullptr;
    const ncclResult_t e = r.CommInitRank(&c, nranks, id, rank);
    if (e != ncclSuccess) return comm_error(m, "ncclCommInitRank", e);
    m->comm = c;
    m->rank = rank;
    m->nranks = nranks;
    return LOM_OK;
}

int lom_comm_finalize(lom_map *m)
{
    if (!m) return LOM_ERR_ARG;
    if (!m->comm) return LOM_OK;
    (void)hipSetDevice(m->device);
    (void)hipStreamSynchronize(m->stream);
    rccl().CommDestroy((ncclComm_t)m->comm);
    m->comm = nullptr;
    m->rank = 0;
    m->nranks = 1;
    return LOM_OK;
}

}  // extern "C"
