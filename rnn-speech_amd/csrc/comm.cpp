// Data-parallel exchange behind the C ABI: an RCCL communicator (xGMI inside a node) created from a unique id
// that rank 0 makes and the host broadcasts by any means it likes, plus the two collectives the training path
// needs -- the fp32 SUM all-reduce of the flat gradient buffer (one exchange per optimiser step;
// N ranks x batch b == the reference's mini_batch_size = N accumulation, models/AcousticModel.py:391-406) and a
// broadcast for restoring replicas from rank 0's checkpoint.
//
// RCCL is bound at run time (dlopen/dlsym), not at link time: a process that already carries an RCCL (PyTorch
// ships one) must keep using that single copy, and a single-GPU user needs none at all.
#include "common.h"
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include <mutex>
#include <string>

// The handful of RCCL / NCCL ABI types this file needs, declared here so that a build host without the RCCL headers still builds
// the library (values are those of nccl.h / rccl.h, a stable ABI: NCCL_UNIQUE_ID_BYTES 128, ncclSuccess 0, ncclSum 0, ncclFloat32 7)
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
typedef int ncclDataType_t;
typedef int ncclRedOp_t;
}
namespace {
constexpr ncclResult_t ncclSuccess = 0;
constexpr ncclRedOp_t ncclSum = 0;
constexpr ncclDataType_t ncclFloat32 = 7;

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;          // (optional: diagnostics only)
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    std::string path, why;      // the shared object that was bound / why none was
    bool ok = false;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // an RCCL that is already in the process first (RTLD_NOLOAD), then the ROCm installation's
        const char* names[] = {"librccl.so", "librccl.so.1"};
        for (const char* n : names)
            if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        for (const char* n : names)
            if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (!r.handle) r.handle = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!r.handle) {
            const char* e = dlerror();
            r.why = e ? e : "dlopen(librccl.so) failed";
            return;
        }
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.handle, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.handle, "ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.handle, "ncclCommDestroy"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(r.handle, "ncclAllReduce"));
        r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(dlsym(r.handle, "ncclBroadcast"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.handle, "ncclGetErrorString"));
        r.CommCount = reinterpret_cast<decltype(r.CommCount)>(dlsym(r.handle, "ncclCommCount"));
        r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(dlsym(r.handle, "ncclCommUserRank"));
        r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(dlsym(r.handle, "ncclGetVersion"));
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.Broadcast && r.GetErrorString;
        if (!r.ok) r.why = "a required nccl* symbol is missing from the bound library";
        Dl_info info;
        if (r.GetUniqueId && dladdr(reinterpret_cast<void*>(r.GetUniqueId), &info) && info.dli_fname) r.path = info.dli_fname;
        if (amdspeech::runtime_switch("AMDSPEECH_COMM_DEBUG", 0)) fprintf(stderr, "amdspeech comm: bound RCCL at %s\n", r.path.c_str());
    });
    return r;
}

struct Comm {
    ncclComm_t nccl = nullptr;
    int rank = 0, world = 1;
};

int need_rccl() {
    if (rccl().ok) return AMDSPEECH_OK;
    amdspeech::set_error("comm: RCCL is not available (%s)", rccl().why.c_str());
    return AMDSPEECH_EUNSUPPORTED;
}

#define AS_CHECK_RCCL(expr)                                                                        \
    do {                                                                                           \
        ncclResult_t r__ = (expr);                                                                 \
        if (r__ != ncclSuccess) {                                                                  \
            amdspeech::set_error("%s failed: %s", #expr, rccl().GetErrorString(r__));              \
            return AMDSPEECH_EHIP;                                                                 \
        }                                                                                          \
    } while (0)

}  // namespace

// Can this process bind RCCL at all?  dlopen + the symbol check, nothing else: ncclGetUniqueId (what the bootstrap used to probe
// with on every rank) creates a bootstrap root -- a listening socket and a thread -- that only rank 0's id ever gets connections on.
extern "C" int amdspeech_comm_available(void) {
    return need_rccl();
}

extern "C" int amdspeech_comm_unique_id(void* id_out) {
    AS_CHECK_ARG(id_out != nullptr, "comm_unique_id: null pointer");
    if (int rc = need_rccl()) return rc;
    static_assert(sizeof(ncclUniqueId) == AMDSPEECH_COMM_ID_BYTES, "unique id size");
    AS_CHECK_RCCL(rccl().GetUniqueId(static_cast<ncclUniqueId*>(id_out)));
    return AMDSPEECH_OK;
}

extern "C" int amdspeech_comm_init(const void* id, int rank, int world, void** comm_out) {
    AS_CHECK_ARG(id && comm_out, "comm_init: null pointer");
    AS_CHECK_ARG(world >= 1 && rank >= 0 && rank < world, "comm_init: rank %d of %d", rank, world);
    if (int rc = need_rccl()) return rc;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    Comm* c = new (std::nothrow) Comm;
    AS_CHECK_ARG(c != nullptr, "comm_init: out of memory");
    c->rank = rank; c->world = world;
    ncclResult_t r = rccl().CommInitRank(&c->nccl, world, uid, rank);      // (uses the calling thread's current device)
    if (r != ncclSuccess) {
        amdspeech::set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, world, rccl().GetErrorString(r));
        delete c;
        return AMDSPEECH_EHIP;
    }
    *comm_out = c;
    return AMDSPEECH_OK;
}

extern "C" int amdspeech_comm_destroy(void* comm) {
    if (!comm) return AMDSPEECH_OK;
    Comm* c = static_cast<Comm*>(comm);
    if (c->nccl && rccl().ok) (void)rccl().CommDestroy(c->nccl);
    delete c;
    return AMDSPEECH_OK;
}

extern "C" int amdspeech_comm_info(void* comm, int* rank, int* world, int* rccl_version, char* lib_path, int lib_path_len) {
    AS_CHECK_ARG(comm != nullptr, "comm_info: null communicator");
    Comm* c = static_cast<Comm*>(comm);
    int r = c->rank, w = c->world, v = 0;
    // asked of the communicator itself where the library can answer (what RCCL believes, not what the launcher said)
    if (rccl().CommCount) AS_CHECK_RCCL(rccl().CommCount(c->nccl, &w));
    if (rccl().CommUserRank) AS_CHECK_RCCL(rccl().CommUserRank(c->nccl, &r));
    if (rccl().GetVersion) (void)rccl().GetVersion(&v);
    if (rank) *rank = r;
    if (world) *world = w;
    if (rccl_version) *rccl_version = v;
    if (lib_path && lib_path_len > 0) snprintf(lib_path, (size_t)lib_path_len, "%s", rccl().path.c_str());
    return AMDSPEECH_OK;
}

extern "C" int amdspeech_allreduce_sum_f32(void* comm, void* stream, float* buf, long n) {
    AS_CHECK_ARG(comm && buf && n >= 0, "allreduce_sum_f32: bad argument");
    Comm* c = static_cast<Comm*>(comm);
    if (n == 0) return AMDSPEECH_OK;
    // ONE in-place collective over the whole flat buffer (25 MB at 3x512, 169 MB at 5x1024): RCCL pipelines it over
    // the xGMI rings itself; nothing is gained by bucketing a buffer that is complete when backward ends
    AS_CHECK_RCCL(rccl().AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, c->nccl, static_cast<hipStream_t>(stream)));
    return AMDSPEECH_OK;
}

extern "C" int amdspeech_broadcast_f32(void* comm, void* stream, float* buf, long n, int root) {
    AS_CHECK_ARG(comm && buf && n >= 0, "broadcast_f32: bad argument");
    Comm* c = static_cast<Comm*>(comm);
    AS_CHECK_ARG(root >= 0 && root < c->world, "broadcast_f32: root %d of %d", root, c->world);
    if (n == 0) return AMDSPEECH_OK;
    AS_CHECK_RCCL(rccl().Broadcast(buf, buf, (size_t)n, ncclFloat32, root, c->nccl, static_cast<hipStream_t>(stream)));
    return AMDSPEECH_OK;
}
