// Data-parallel exchange over RCCL (xGMI) behind the C ABI: communicator set-up, in-place SUM all-reduce, and the SyncBN moment exchange.
//
// What it replaces: the reference has no working multi-GPU path -- nn.DataParallel is commented out (models/model.py:186-188) and the
// vendored SynchronizedBatchNorm2d (models/sync_batchnorm/batchnorm.py:48-125, comm.py: a master / slave queue pair that reduces
// [sum, sumsq, count] on the master and broadcasts mean / inv_std back) is never constructed.  SURVEY.md section 8b / 8e: one process
// per GPU, RCCL; SyncBN = all-reduce of the per-GPU moments; gradients = SUM all-reduce of the flat arena.
//
// RCCL is resolved at RUN time (dlopen of the librccl.so.1 already in the process -- PyTorch-ROCm ships one -- or the system's), so that
// libpylc_hip.so neither links a second copy next to torch's nor fails to load on a box without RCCL: pylc_comm_* then return
// PYLC_ERR_UNSUPPORTED with the loader's message.  Every call is enqueued on the caller's stream; nothing here synchronises.
#include "common.h"
#include <dlfcn.h>
#include <mutex>

namespace pylc {
namespace {

typedef struct { char internal[128]; } NcclUniqueId;      // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void* NcclComm;
enum { kNcclSuccess = 0, kNcclFloat32 = 7, kNcclFloat64 = 8, kNcclSum = 0 };      // ncclDataType_t / ncclRedOp_t values of nccl.h / rccl.h

struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(NcclUniqueId*) = nullptr;
    int (*CommInitRank)(NcclComm*, int, NcclUniqueId, int) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    int (*CommDestroy)(NcclComm) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    char why[256] = "";
};

void rccl_resolve(Rccl& r);

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;          // two host threads may enter their first pylc_comm_* call together
    std::call_once(once, [] { rccl_resolve(r); });
    return r;
}

void rccl_resolve(Rccl& r) {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);      // the copy already in the process (torch's), if any
        if (r.handle) break;
    }
    for (const char* n : names) {
        if (r.handle) break;
        r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    }
    if (!r.handle) {
        snprintf(r.why, sizeof(r.why), "librccl.so.1 not loadable: %s", dlerror());
        return;
    }
#define PYLC_SYM(field, name)                                                       \
    *reinterpret_cast<void**>(&r.field) = dlsym(r.handle, name);                   \
    if (!r.field) { snprintf(r.why, sizeof(r.why), "RCCL symbol %s missing", name); r.handle = nullptr; return; }
    PYLC_SYM(GetUniqueId, "ncclGetUniqueId")
    PYLC_SYM(CommInitRank, "ncclCommInitRank")
    PYLC_SYM(AllReduce, "ncclAllReduce")
    PYLC_SYM(CommDestroy, "ncclCommDestroy")
    PYLC_SYM(GetErrorString, "ncclGetErrorString")
#undef PYLC_SYM
}

struct Comm { NcclComm comm; int rank, world; };

#define PYLC_RCCL_READY(r)                                                                                  \
    Rccl& r = rccl();                                                                                       \
    if (!r.handle) return fail(PYLC_ERR_UNSUPPORTED, "pylc_comm: %s", r.why)
#define PYLC_RCCL(r, call, what)                                                                            \
    do {                                                                                                    \
        const int rc_ = (call);                                                                             \
        if (rc_ != kNcclSuccess) return fail(PYLC_ERR_HIP, "%s: RCCL error %d (%s)", what, rc_, r.GetErrorString(rc_)); \
    } while (0)

}  // namespace
}  // namespace pylc

using namespace pylc;

// Local probe: can this process reach an RCCL at all?  No GPU call, no collective -- what every rank asks first, so that the decision to
// start a communicator hand-shake is taken by all ranks together (pylc_amd/parallel.py try_native_comm)
extern "C" int pylc_comm_available(void) {
    PYLC_RCCL_READY(r);
    return PYLC_OK;
}

// 128 bytes that identify a new communicator: produced on ONE rank, handed to every rank by the caller (torch.distributed store, a file, MPI)
extern "C" int pylc_comm_unique_id(void* id_out) {
    PYLC_REQUIRE(id_out != nullptr, "comm_unique_id: null output");
    PYLC_RCCL_READY(r);
    PYLC_RCCL(r, r.GetUniqueId(static_cast<NcclUniqueId*>(id_out)), "ncclGetUniqueId");
    return PYLC_OK;
}

// Collective over all `world` ranks; the calling thread's current HIP device is the rank's GPU
extern "C" int pylc_comm_init(const void* id, int rank, int world, void** comm_out) {
    PYLC_REQUIRE(id != nullptr && comm_out != nullptr && world >= 1 && rank >= 0 && rank < world, "comm_init: bad arguments (rank %d of %d)", rank, world);
    PYLC_RCCL_READY(r);
    NcclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    Comm* c = new Comm{nullptr, rank, world};
    const int rc = r.CommInitRank(&c->comm, world, uid, rank);
    if (rc != kNcclSuccess) {
        delete c;
        return fail(PYLC_ERR_HIP, "ncclCommInitRank: RCCL error %d (%s)", rc, r.GetErrorString(rc));
    }
    *comm_out = c;
    return PYLC_OK;
}

// In-place SUM all-reduce of `count` elements (dtype 0 = fp32, 1 = fp64) on `stream`: the gradient buckets of the flat arena, the loss
// head's partial sums, the BatchNorm backward's [sum g xhat | sum g]
extern "C" int pylc_comm_allreduce(void* comm, void* buf, long long count, int dtype, void* stream) {
    PYLC_REQUIRE(comm != nullptr && buf != nullptr && count > 0 && (dtype == 0 || dtype == 1), "comm_allreduce: bad arguments");
    PYLC_RCCL_READY(r);
    Comm* c = static_cast<Comm*>(comm);
    PYLC_RCCL(r, r.AllReduce(buf, buf, (size_t)count, dtype == 0 ? kNcclFloat32 : kNcclFloat64, kNcclSum, c->comm, as_stream(stream)), "ncclAllReduce");
    return PYLC_OK;
}

// SyncBN forward exchange: in-place SUM of this rank's fp64 moments [sum x | sum x^2 | n] (2 C + 1 doubles, written by
// pylc_bn_local_moments, consumed by pylc_bn_finalize_moments) -- batchnorm.py:66-68,113-125's reduce + broadcast as one all-reduce
extern "C" int pylc_comm_syncbn_reduce(void* comm, double* moments, int channels, void* stream) {
    PYLC_REQUIRE(channels > 0, "comm_syncbn_reduce: bad channel count");
    return pylc_comm_allreduce(comm, moments, 2ll * channels + 1, 1, stream);
}

extern "C" int pylc_comm_destroy(void* comm) {
    PYLC_REQUIRE(comm != nullptr, "comm_destroy: null communicator");
    PYLC_RCCL_READY(r);
    Comm* c = static_cast<Comm*>(comm);
    PYLC_RCCL(r, r.CommDestroy(c->comm), "ncclCommDestroy");
    delete c;
    return PYLC_OK;
}
