// shapegan_amd/csrc/comm.cpp — the gradient exchange of the data-parallel path behind the C ABI: RCCL all-reduce of the flat
// fp32 gradient buffer of one network per optimizer step, stream-ordered, without torch.distributed on the hot path.
//
// Reference: nn.DataParallel's reduce-add of replica gradients (train_hybrid_progressive_gan.py:62-68) — here one process per
// GPU, identical replicas, one (or two: tail slice early, head slice at the end of backward) ncclAllReduce(sum) per update;
// the 1/world factor is applied by the optimizer kernel (grad_scale).  Built into its own library (libshapegan_comm.so) so
// that single-GPU users of libshapegan_hip.so do not need RCCL.
//
// Ordering: every communicator owns a side stream.  sg_allreduce_launch records an event on the caller's compute stream,
// makes the side stream wait for it (the slice's gradients are complete), and enqueues the all-reduce there — later backward
// kernels on the compute stream overlap with it.  sg_allreduce_wait makes the compute stream wait for everything enqueued so
// far.  No host synchronisation anywhere.
//
// Which RCCL: the library does NOT link librccl.  sg_comm_bind(path) opens the RCCL the host process names — the one
// torch.distributed's "nccl" backend has already mapped (torch/lib/librccl.so), so that both exchanges of a process live in ONE
// RCCL instance of ONE version — and resolves the nine entry points it needs from it; rccl.h is used for types only.  The header
// version this file was compiled against and the version the bound library reports are both readable (sg_comm_versions) and a
// different MAJOR version is refused (round 6: the ROCm 7.2 header is 2.27.7, torch 2.10+rocm7.0 carries 2.26.6; the 2.x entry
// points used here have not changed their signatures).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <string.h>

#include "../../include/shapegan_hip.h"

namespace {
struct RcclApi {
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    decltype(&ncclCommCuDevice) CommCuDevice = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    void* handle = nullptr;
    int runtime_version = 0;
};
RcclApi g_rccl;
}  // namespace
#define ncclGetUniqueId g_rccl.GetUniqueId
#define ncclCommInitRank g_rccl.CommInitRank
#define ncclCommDestroy g_rccl.CommDestroy
#define ncclCommCount g_rccl.CommCount
#define ncclCommUserRank g_rccl.CommUserRank
#define ncclCommCuDevice g_rccl.CommCuDevice
#define ncclGetVersion g_rccl.GetVersion
#define ncclAllReduce g_rccl.AllReduce
#define ncclGetErrorString g_rccl.GetErrorString

struct sg_comm {
    ncclComm_t comm;
    hipStream_t side;
    hipEvent_t ready, done;
    int rank, world, device;
    int pending;
};

static thread_local char g_comm_err[512];
#define COMM_FAIL(...)                               \
    do {                                             \
        snprintf(g_comm_err, 512, __VA_ARGS__);      \
        return -2;                                   \
    } while (0)
#define COMM_HIP(call)                                                                       \
    do {                                                                                     \
        hipError_t e__ = (call);                                                             \
        if (e__ != hipSuccess) COMM_FAIL("%s: %s failed: %s", __func__, #call, hipGetErrorString(e__)); \
    } while (0)
#define COMM_NCCL(call)                                                                        \
    do {                                                                                       \
        ncclResult_t r__ = (call);                                                             \
        if (r__ != ncclSuccess) COMM_FAIL("%s: %s failed: %s", __func__, #call, ncclGetErrorString(r__)); \
    } while (0)

extern "C" {

const char* sg_comm_last_error(void) { return g_comm_err; }

#define COMM_BOUND() \
    if (!g_rccl.handle) COMM_FAIL("%s: no RCCL bound: call sg_comm_bind first", __func__)

// Binds the RCCL at `path` (NULL: "librccl.so.1" by the loader's search order — an already mapped one is found first).  Idempotent
// for the same library; a second, different library is refused.
int sg_comm_bind(const char* path) {
    const char* name = path && path[0] ? path : "librccl.so.1";
    void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (!h) COMM_FAIL("sg_comm_bind: dlopen(%s) failed: %s", name, dlerror());
    if (g_rccl.handle) {
        const bool same = g_rccl.handle == h;
        dlclose(h);
        if (same) return 0;
        COMM_FAIL("sg_comm_bind: another RCCL is already bound");
    }
    RcclApi a;
    a.handle = h;
#define SG_SYM(field, sym)                                                              \
    a.field = reinterpret_cast<decltype(a.field)>(dlsym(h, sym));                      \
    if (!a.field) {                                                                     \
        dlclose(h);                                                                     \
        COMM_FAIL("sg_comm_bind: %s does not export %s", name, sym);                   \
    }
    SG_SYM(GetUniqueId, "ncclGetUniqueId")
    SG_SYM(CommInitRank, "ncclCommInitRank")
    SG_SYM(CommDestroy, "ncclCommDestroy")
    SG_SYM(CommCount, "ncclCommCount")
    SG_SYM(CommUserRank, "ncclCommUserRank")
    SG_SYM(CommCuDevice, "ncclCommCuDevice")
    SG_SYM(GetVersion, "ncclGetVersion")
    SG_SYM(AllReduce, "ncclAllReduce")
    SG_SYM(GetErrorString, "ncclGetErrorString")
#undef SG_SYM
    int v = 0;
    if (a.GetVersion(&v) != ncclSuccess) {
        dlclose(h);
        COMM_FAIL("sg_comm_bind: ncclGetVersion of %s failed", name);
    }
    a.runtime_version = v;
    const int runtime_major = v >= 10000 ? v / 10000 : v / 1000;
    if (runtime_major != NCCL_MAJOR) {
        dlclose(h);
        COMM_FAIL("sg_comm_bind: %s is RCCL %d, this library was compiled against the %d.%d.%d header: major versions differ", name, v,
                  NCCL_MAJOR, NCCL_MINOR, NCCL_PATCH);
    }
    g_rccl = a;
    return 0;
}

// header_version: NCCL_VERSION_CODE of the rccl.h this file was compiled against; runtime_version: what the bound library's
// ncclGetVersion reports (0 before sg_comm_bind).
int sg_comm_versions(int* header_version, int* runtime_version) {
    if (header_version) *header_version = NCCL_VERSION_CODE;
    if (runtime_version) *runtime_version = g_rccl.runtime_version;
    return 0;
}

size_t sg_allreduce_unique_id_bytes(void) { return sizeof(ncclUniqueId); }

int sg_allreduce_unique_id(void* id_out, size_t bytes) {
    COMM_BOUND();
    if (!id_out || bytes < sizeof(ncclUniqueId)) COMM_FAIL("sg_allreduce_unique_id: need %zu bytes", sizeof(ncclUniqueId));
    ncclUniqueId id;
    COMM_NCCL(ncclGetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return 0;
}

int sg_allreduce_init(sg_comm** out, int rank, int world, const void* unique_id, size_t id_bytes, int device) {
    if (!out || !unique_id || id_bytes < sizeof(ncclUniqueId) || world < 1 || rank < 0 || rank >= world)
        COMM_FAIL("sg_allreduce_init: bad argument");
    COMM_BOUND();
    COMM_HIP(hipSetDevice(device));
    sg_comm* c = new sg_comm();
    c->rank = rank;
    c->world = world;
    c->device = device;
    c->pending = 0;
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclResult_t r = ncclCommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        delete c;
        COMM_FAIL("sg_allreduce_init: ncclCommInitRank failed: %s", ncclGetErrorString(r));
    }
    // every error path below releases what was created before it
    hipError_t e = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking);
    if (e == hipSuccess) {
        e = hipEventCreateWithFlags(&c->ready, hipEventDisableTiming);
        if (e == hipSuccess) {
            e = hipEventCreateWithFlags(&c->done, hipEventDisableTiming);
            if (e == hipSuccess) {
                *out = c;
                return 0;
            }
            (void)hipEventDestroy(c->ready);
        }
        (void)hipStreamDestroy(c->side);
    }
    (void)ncclCommDestroy(c->comm);
    delete c;
    COMM_FAIL("sg_allreduce_init: creating the side stream / events failed: %s", hipGetErrorString(e));
}

// What the communicator itself reports (read back for bench.py's "comm" record): the number of ranks RCCL formed the
// communicator over (ncclCommCount), this rank's index in it (ncclCommUserRank), its device (ncclCommCuDevice) and the RCCL
// version code (ncclGetVersion; e.g. 22606 = 2.26.6).
int sg_allreduce_info(sg_comm* c, int* ranks, int* rank, int* device, int* rccl_version) {
    if (!c) COMM_FAIL("sg_allreduce_info: bad argument");
    int v = 0;
    if (ranks) COMM_NCCL(ncclCommCount(c->comm, ranks));
    if (rank) COMM_NCCL(ncclCommUserRank(c->comm, rank));
    if (device) COMM_NCCL(ncclCommCuDevice(c->comm, device));
    if (rccl_version) {
        COMM_NCCL(ncclGetVersion(&v));
        *rccl_version = v;
    }
    return 0;
}

int sg_allreduce_launch(sg_comm* c, float* buf, long count, hipStream_t compute_stream) {
    if (!c || !buf || count <= 0) COMM_FAIL("sg_allreduce_launch: bad argument");
    COMM_HIP(hipEventRecord(c->ready, compute_stream));
    COMM_HIP(hipStreamWaitEvent(c->side, c->ready, 0));
    COMM_NCCL(ncclAllReduce(buf, buf, (size_t)count, ncclFloat32, ncclSum, c->comm, c->side));
    c->pending++;
    return 0;
}

int sg_allreduce_wait(sg_comm* c, hipStream_t compute_stream) {
    if (!c) COMM_FAIL("sg_allreduce_wait: bad argument");
    if (c->pending) {
        COMM_HIP(hipEventRecord(c->done, c->side));
        COMM_HIP(hipStreamWaitEvent(compute_stream, c->done, 0));
        c->pending = 0;
    }
    return 0;
}

int sg_allreduce_destroy(sg_comm* c) {
    if (!c) return 0;
    (void)hipStreamSynchronize(c->side);
    (void)ncclCommDestroy(c->comm);
    (void)hipEventDestroy(c->ready);
    (void)hipEventDestroy(c->done);
    (void)hipStreamDestroy(c->side);
    delete c;
    return 0;
}

}  // extern "C"
