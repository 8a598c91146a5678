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
//
// Second transport, same contract (lom_comm_init_host): the reduced system is consumed by
// the HOST-side solver of every rank, and each rank's host already holds its own 32 sums
// ~15 us after the evaluation (resident server, pinned mailbox).  For ranks of one node the
// hosts can exchange those 256 bytes through POSIX shared memory in about a microsecond;
// the device-side collective costs a launch plus the collective's latency per LM iteration.
#include <dlfcn.h>
#include <fcntl.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstring>
#include <ctime>

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

// one 512-byte slot per (buffer, rank): sequence word + LOM_NSUMS doubles, cache-line separated
struct HostSlot {
    volatile unsigned long long seq;
    double data[LOM_NSUMS];
    char pad[512 - 8 - LOM_NSUMS * 8];
};
static_assert(sizeof(HostSlot) == 512, "slot size");

// A rank that gives up on an exchange (deadline, or lom_host_comm_abort) marks BOTH of its slots with this
// bit: whoever waits for that rank now or later sees it and fails too, instead of pairing with a slot that
// was published for an exchange its owner has walked away from.  An abandoned exchange object stays broken.
constexpr unsigned long long kAbandoned = 1ull << 63;

struct lom_host_comm {
    HostSlot *slots = nullptr;  // [2 buffers][nranks]
    size_t bytes = 0;
    std::string name;
    int rank = 0, nranks = 1;
    unsigned long long seq = 0;
    double timeout_s = 60.0;  // deadline of one exchange (lom_host_comm_set_timeout)
    bool broken = false;      // this rank abandoned an exchange, or saw a peer that had
    std::string error;
};

namespace {

double mono_s()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

void abandon(lom_host_comm *hc, const std::string &why)
{
    hc->broken = true;
    hc->error = why;
    for (int b = 0; b < 2; b++)
        __atomic_store_n(&hc->slots[(size_t)b * hc->nranks + hc->rank].seq, kAbandoned | hc->seq, __ATOMIC_RELEASE);
}

// publish this rank's slot of exchange `seq` (payload written by the caller), then wait for every rank's;
// `take(r, slot)` consumes rank r's payload in rank order.  Double buffered: a rank can start exchange k+2
// only after every rank finished k+1, i.e. after every rank is done reading exchange k, whose buffer it
// then reuses.
// What the return value does NOT promise: that every rank returns the same.  Slots are consumed in rank order and a
// rank gives up by its own clock, so a rank that has taken rank A's slot and then sees the late rank publish returns
// LOM_OK while A -- waiting for the same late rank -- reaches its deadline in that instant and returns LOM_ERR_COMM
// (the two-generals residue of any deadline; a second "done" round would only move the window).  What IS promised: the
// object is broken from then on for everybody -- A marked both its slots, so the ranks that came through fail at
// their NEXT exchange, at once.  For the agreement after a device-to-device align (match.hip align_device) this means:
// in that one instant a rank may return the align's pose with LOM_OK while a peer returns LOM_ERR_COMM for the same
// align; the rank that came through learns it at its next call, which fails at once instead of computing alone.
template <typename Take>
int exchange(lom_host_comm *hc, HostSlot *slots, unsigned long long seq, double timeout_s, Take take)
{
    __atomic_store_n(&slots[hc->rank].seq, seq, __ATOMIC_RELEASE);
    const double t0 = mono_s();
    for (int r = 0; r < hc->nranks; r++) {
        unsigned long long spins = 0;
        for (;;) {
            const unsigned long long v = __atomic_load_n(&slots[r].seq, __ATOMIC_ACQUIRE);
            if (v == seq) break;
            if (v & kAbandoned) {
                abandon(hc, "host exchange: rank " + std::to_string(r) + " abandoned exchange " +
                                std::to_string(v & ~kAbandoned) + " (this rank is at " + std::to_string(seq) + ")");
                return LOM_ERR_COMM;
            }
            __builtin_ia32_pause();
            if ((++spins & 0xFFFF) == 0 && mono_s() - t0 > timeout_s) {
                abandon(hc, "host exchange timed out waiting for rank " + std::to_string(r));
                return LOM_ERR_COMM;
            }
        }
        take(r, slots[r]);
    }
    return LOM_OK;
}

}  // namespace

namespace lom {

int host_comm_rank(void *hc, int *rank, int *nranks, unsigned long long *seq)
{
    lom_host_comm *c = reinterpret_cast<lom_host_comm *>(hc);
    if (!c) return LOM_ERR_ARG;
    *rank = c->rank;
    *nranks = c->nranks;
    if (seq) *seq = c->seq;  // operations so far: the same number on every rank
    return LOM_OK;
}

const char *host_comm_error(void *hc)
{
    lom_host_comm *c = reinterpret_cast<lom_host_comm *>(hc);
    return c ? c->error.c_str() : "";
}

int host_comm_allreduce_deadline(void *hcv, double *buf, int count, double timeout_s)
{
    lom_host_comm *hc = reinterpret_cast<lom_host_comm *>(hcv);
    if (!hc || !buf || count < 0 || count > LOM_NSUMS) return LOM_ERR_ARG;
    if (hc->broken) return LOM_ERR_COMM;  // hc->error says which exchange was abandoned, and by whom
    const unsigned long long seq = ++hc->seq;
    HostSlot *slots = hc->slots + (size_t)(seq & 1) * hc->nranks;
    HostSlot &me = slots[hc->rank];
    for (int k = 0; k < count; k++) me.data[k] = buf[k];
    for (int k = 0; k < count; k++) buf[k] = 0.0;
    return exchange(hc, slots, seq, timeout_s, [&](int, HostSlot &s) {
        for (int k = 0; k < count; k++) buf[k] += s.data[k];  // rank order
    });
}

int host_exchange_sums(lom_map *m, const double *mine, double *out)
{
    lom_host_comm *hc = reinterpret_cast<lom_host_comm *>(m->host_comm);
    if (!hc) return set_error(m, LOM_ERR_COMM, "host exchange not attached");
    for (int k = 0; k < LOM_NSUMS; k++) out[k] = mine[k];
    const int rc = lom_host_comm_allreduce(hc, out, LOM_NSUMS);
    if (rc != LOM_OK) return set_error(m, rc, hc->error.c_str());
    return LOM_OK;
}

}  // namespace lom

extern "C" {

int lom_host_comm_create(int rank, int nranks, const char id_in[LOM_COMM_ID_BYTES], lom_host_comm **out)
{
    if (!out || !id_in || nranks < 1 || nranks > 64 || rank < 0 || rank >= nranks) return LOM_ERR_ARG;
    *out = nullptr;
    lom_host_comm *hc = new (std::nothrow) lom_host_comm();
    if (!hc) return LOM_ERR_OOM;
    char name[64];
    std::snprintf(name, sizeof name, "/lom_%02x%02x%02x%02x%02x%02x%02x%02x", (unsigned char)id_in[0],
                  (unsigned char)id_in[1], (unsigned char)id_in[2], (unsigned char)id_in[3], (unsigned char)id_in[4],
                  (unsigned char)id_in[5], (unsigned char)id_in[6], (unsigned char)id_in[7]);
    hc->name = name;
    hc->rank = rank;
    hc->nranks = nranks;
    hc->bytes = sizeof(HostSlot) * 2 * (size_t)nranks;
    // every rank opens with O_CREAT; the segment starts zero-filled (seq 0 = nothing published)
    const int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)hc->bytes) != 0) {
        if (fd >= 0) close(fd);
        delete hc;
        return LOM_ERR_COMM;
    }
    void *p = mmap(nullptr, hc->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) {
        delete hc;
        return LOM_ERR_COMM;
    }
    hc->slots = reinterpret_cast<HostSlot *>(p);
    *out = hc;
    return LOM_OK;
}

int lom_host_comm_set_timeout(lom_host_comm *hc, double seconds)
{
    if (!hc || !(seconds > 0.0)) return LOM_ERR_ARG;
    hc->timeout_s = seconds;
    return LOM_OK;
}

int lom_host_comm_abort(lom_host_comm *hc)
{
    if (!hc) return LOM_ERR_ARG;
    if (!hc->broken) abandon(hc, "host exchange aborted by this rank");
    return LOM_OK;
}

const char *lom_host_comm_last_error(const lom_host_comm *hc) { return hc ? hc->error.c_str() : ""; }

// In-place sum over the ranks of buf[0..count), count <= LOM_NSUMS, added in rank order on every
// rank (bitwise the same result everywhere).
int lom_host_comm_allreduce(lom_host_comm *hc, double *buf, int count)
{
    if (!hc) return LOM_ERR_ARG;
    return lom::host_comm_allreduce_deadline(hc, buf, count, hc->timeout_s);
}

// Raw bytes through the same double-buffered slots (a slot's data area is 256 bytes).
int lom_host_comm_allgather(lom_host_comm *hc, const void *mine, size_t bytes, void *all_out)
{
    if (!hc || !mine || !all_out || bytes == 0 || bytes > sizeof(double) * LOM_NSUMS) return LOM_ERR_ARG;
    if (hc->broken) return LOM_ERR_COMM;
    const unsigned long long seq = ++hc->seq;
    HostSlot *slots = hc->slots + (size_t)(seq & 1) * hc->nranks;
    std::memcpy(slots[hc->rank].data, mine, bytes);
    return exchange(hc, slots, seq, hc->timeout_s, [&](int r, HostSlot &s) {
        std::memcpy(static_cast<char *>(all_out) + (size_t)r * bytes, s.data, bytes);
    });
}

void lom_host_comm_destroy(lom_host_comm *hc)
{
    if (!hc) return;
    munmap(hc->slots, hc->bytes);
    shm_unlink(hc->name.c_str());  // every rank unlinks; later calls fail harmlessly
    delete hc;
}

int lom_comm_attach_host(lom_map *m, lom_host_comm *hc)
{
    if (!m) return LOM_ERR_ARG;
    if (m->comm) return lom::set_error(m, LOM_ERR_STATE, "an RCCL communicator is already attached");
    if (m->p2p) lom::p2p_detach(m);
    m->host_comm = hc;  // NULL detaches; the caller keeps ownership
    m->rank = hc ? hc->rank : 0;
    m->nranks = hc ? hc->nranks : 1;
    return LOM_OK;
}

int lom_comm_host_id(char id_out[LOM_COMM_ID_BYTES])
{
    if (!id_out) return LOM_ERR_ARG;
    std::memset(id_out, 0, LOM_COMM_ID_BYTES);
    const int fd = open("/dev/urandom", O_RDONLY);
    if (fd < 0 || read(fd, id_out, 16) != 16) {
        if (fd >= 0) close(fd);
        return lom::set_error(nullptr, LOM_ERR_COMM, "/dev/urandom not readable");
    }
    close(fd);
    return LOM_OK;
}

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
    if (m->comm || m->host_comm) return lom::set_error(m, LOM_ERR_STATE, "communicator already initialised");
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
    if (m->p2p) lom::p2p_detach(m);
    if (m->host_comm) {  // attached host exchange: detach (its owner destroys it)
        m->host_comm = nullptr;
        m->rank = 0;
        m->nranks = 1;
    }
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
