// RCCL helpers of the C ABI: the one exchange step of the data-parallel path (sum / average of every parameter gradient
// over the ranks, one process per GPU over xGMI) for hosts that do not bring torch.distributed.
//
// The reference has no collective call site of its own: README.md:40 defers to PL's Trainer(gpus=N), i.e. torch DDP's
// bucketed NCCL all-reduce of the gradients, divided by the world size.
//
// RCCL is resolved at RUN time, not linked: inside a PyTorch process the copy that torch already loaded is used (two
// RCCL instances in one process would each open their own xGMI / IPC state), otherwise the system librccl.so.1; a host
// without RCCL still loads libw2l_hip.so and only these entry points report an error.
#include "common.h"
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <rccl/rccl.h>

namespace {

struct Rccl {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl g_rccl;
std::once_flag g_rccl_once;

void rccl_resolve() {
    static const char* names[] = {"librccl.so.1", "librccl.so"};
    void* h = nullptr;
    for (const char* n : names)                       // a copy that is already mapped (torch's) wins
        if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    for (const char* n : names)
        if (!h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!h) return;
    Rccl r;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    r.CommCount = (decltype(r.CommCount))dlsym(h, "ncclCommCount");
    r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
    r.Broadcast = (decltype(r.Broadcast))dlsym(h, "ncclBroadcast");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.CommCount && r.AllReduce && r.Broadcast && r.GetErrorString;
    g_rccl = r;
}

const Rccl* rccl() {
    std::call_once(g_rccl_once, rccl_resolve);
    return g_rccl.ok ? &g_rccl : nullptr;
}

#define W2L_RCCL_OR_FAIL(R)                                                                             \
    const Rccl* R = rccl();                                                                             \
    W2L_CHECK_ARG(R != nullptr, "RCCL not available: librccl.so.1 could not be loaded (or lacks the nccl* entry points)")

#define W2L_CHECK_RCCL(R, expr)                                                                         \
    do {                                                                                                \
        ncclResult_t _r = (expr);                                                                       \
        if (_r != ncclSuccess) {                                                                        \
            w2l_set_error("%s failed: %s", #expr, (R)->GetErrorString(_r));                             \
            return 1000 + (int)_r;                                                                      \
        }                                                                                               \
    } while (0)

}  // namespace

static_assert(NCCL_UNIQUE_ID_BYTES == 128, "W2L_RCCL_ID_BYTES in w2l_hip.h must match RCCL's unique id");

extern "C" int w2l_rccl_available(void) { return rccl() != nullptr ? 1 : 0; }

// path of the RCCL shared object whose entry points are in use ("" without RCCL): inside a PyTorch process this must be
// the copy torch itself links, not a second instance
extern "C" const char* w2l_rccl_library(void) {
    const Rccl* r = rccl();
    Dl_info info;
    if (r == nullptr || dladdr((const void*)r->AllReduce, &info) == 0 || info.dli_fname == nullptr) return "";
    return info.dli_fname;
}

extern "C" int w2l_rccl_unique_id(void* id_host) {
    W2L_CHECK_ARG(id_host != nullptr, "rccl_unique_id: null pointer");
    W2L_RCCL_OR_FAIL(r);
    ncclUniqueId id;
    W2L_CHECK_RCCL(r, r->GetUniqueId(&id));
    memcpy(id_host, id.internal, NCCL_UNIQUE_ID_BYTES);
    return 0;
}

extern "C" int w2l_rccl_init(const void* id_host, int rank, int world, void** comm_out) {
    W2L_CHECK_ARG(id_host && comm_out, "rccl_init: null pointer");
    W2L_CHECK_ARG(world >= 1 && rank >= 0 && rank < world, "rccl_init: rank %d outside world %d", rank, world);
    W2L_RCCL_OR_FAIL(r);
    ncclUniqueId id;
    memcpy(id.internal, id_host, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t comm = nullptr;
    W2L_CHECK_RCCL(r, r->CommInitRank(&comm, world, id, rank));      // collective over the ranks; binds the current device
    *comm_out = (void*)comm;
    return 0;
}

extern "C" int w2l_rccl_world(void* comm, int* world_out) {
    W2L_CHECK_ARG(comm && world_out, "rccl_world: null pointer");
    W2L_RCCL_OR_FAIL(r);
    W2L_CHECK_RCCL(r, r->CommCount((ncclComm_t)comm, world_out));
    return 0;
}

extern "C" int w2l_rccl_all_reduce(void* comm, void* buf, int64_t count, int dtype, int average, void* stream) {
    W2L_CHECK_ARG(comm && buf, "rccl_all_reduce: null pointer");
    W2L_CHECK_ARG(count >= 0, "rccl_all_reduce: negative count");
    W2L_CHECK_ARG(dtype == 0 || dtype == 1, "rccl_all_reduce: dtype %d (0 = fp32, 1 = bf16)", dtype);
    if (count == 0) return 0;
    W2L_RCCL_OR_FAIL(r);
    W2L_CHECK_RCCL(r, r->AllReduce(buf, buf, (size_t)count, dtype == 0 ? ncclFloat32 : ncclBfloat16,
                                   average ? ncclAvg : ncclSum, (ncclComm_t)comm, (hipStream_t)stream));
    return 0;
}

extern "C" int w2l_rccl_broadcast(void* comm, void* buf, int64_t bytes, int root, void* stream) {
    W2L_CHECK_ARG(comm && buf, "rccl_broadcast: null pointer");
    W2L_CHECK_ARG(bytes >= 0 && root >= 0, "rccl_broadcast: bad size / root");
    if (bytes == 0) return 0;
    W2L_RCCL_OR_FAIL(r);
    W2L_CHECK_RCCL(r, r->Broadcast(buf, buf, (size_t)bytes, ncclUint8, root, (ncclComm_t)comm, (hipStream_t)stream));
    return 0;
}

extern "C" int w2l_rccl_destroy(void* comm) {
    if (comm == nullptr) return 0;
    W2L_RCCL_OR_FAIL(r);
    W2L_CHECK_RCCL(r, r->CommDestroy((ncclComm_t)comm));
    return 0;
}
