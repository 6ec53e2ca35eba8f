// comm.hip — RCCL communicator behind the C ABI (one process per GPU, collectives over xGMI).
//
// Reference: utils/distributed.py:16-209 (Horovod 0.16.4 + NCCL: allreduce_ of one flat gradient
// buffer, broadcast_ of parameters, allgather of pickled objects).  librccl is loaded lazily with
// dlopen so that libuniter_hip.so has no hard link dependency on it (single-GPU users never touch it).
#include "common.cuh"
#include "../../include/uniter_hip.h"

#include <dlfcn.h>
#include <string.h>

namespace {

// minimal RCCL surface (ABI-stable NCCL 2 signatures)
typedef struct { char internal[128]; } rcclUniqueId;
typedef void* rcclComm_t;
enum { RCCL_SUM = 0 };
enum { RCCL_INT8 = 0, RCCL_UINT8 = 1, RCCL_FLOAT32 = 7, RCCL_BFLOAT16 = 9 };

struct Api {
    void* lib = nullptr;
    int (*GetUniqueId)(rcclUniqueId*) = nullptr;
    int (*CommInitRank)(rcclComm_t*, int, rcclUniqueId, int) = nullptr;
    int (*CommDestroy)(rcclComm_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, rcclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
Api g_api;

int load_api() {
    if (g_api.lib) return 0;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* lib = nullptr;
    for (const char* n : names) {
        lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (lib) break;
    }
    if (!lib) { uh_set_error("uniter_comm: cannot dlopen librccl (%s)", dlerror()); return -2; }
#define SYM(field, name)                                                                   \
    *(void**)(&g_api.field) = dlsym(lib, name);                                            \
    if (!g_api.field) { uh_set_error("uniter_comm: librccl lacks %s", name); return -2; }
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(AllReduce, "ncclAllReduce");
    SYM(Broadcast, "ncclBroadcast");
    SYM(AllGather, "ncclAllGather");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    g_api.lib = lib;
    return 0;
}

#define NCCL_CHECK(expr)                                                                             \
    do {                                                                                             \
        int _r = (expr);                                                                             \
        if (_r != 0) {                                                                               \
            uh_set_error("%s: %s -> %s", __func__, #expr, g_api.GetErrorString ? g_api.GetErrorString(_r) : "?"); \
            return 1000 + _r;                                                                        \
        }                                                                                            \
    } while (0)

struct Comm {
    rcclComm_t comm;
    int rank, world;
};

}  // namespace

extern "C" {

int uniter_comm_unique_id(uint8_t id_out[128]) {
    UH_CHECK_ARG(id_out != nullptr, "null pointer");
    int rc = load_api();
    if (rc) return rc;
    rcclUniqueId id;
    NCCL_CHECK(g_api.GetUniqueId(&id));
    memcpy(id_out, id.internal, 128);
    return 0;
}

int uniter_comm_init(const uint8_t id[128], int32_t rank, int32_t world, void** comm_out) {
    UH_CHECK_ARG(id != nullptr && comm_out != nullptr, "null pointer");
    UH_CHECK_ARG(world >= 1 && rank >= 0 && rank < world, "bad rank / world size");
    int rc = load_api();
    if (rc) return rc;
    rcclUniqueId uid;
    memcpy(uid.internal, id, 128);
    Comm* c = new Comm();
    c->rank = rank; c->world = world; c->comm = nullptr;
    int r = g_api.CommInitRank(&c->comm, world, uid, rank);
    if (r != 0) {
        uh_set_error("uniter_comm_init: ncclCommInitRank -> %s", g_api.GetErrorString(r));
        delete c;
        return 1000 + r;
    }
    *comm_out = c;
    return 0;
}

int uniter_comm_destroy(void* comm) {
    if (comm == nullptr) return 0;
    Comm* c = (Comm*)comm;
    if (g_api.CommDestroy && c->comm) (void)g_api.CommDestroy(c->comm);
    delete c;
    return 0;
}

int uniter_comm_allreduce(void* comm, void* buf, int64_t count, int32_t dtype, void* stream) {
    UH_CHECK_ARG(comm != nullptr && buf != nullptr && count > 0, "null pointer / empty buffer");
    UH_CHECK_ARG(dtype == 0 || dtype == 1, "dtype must be 0 (bf16) or 1 (fp32)");
    Comm* c = (Comm*)comm;
    NCCL_CHECK(g_api.AllReduce(buf, buf, (size_t)count, dtype == 0 ? RCCL_BFLOAT16 : RCCL_FLOAT32, RCCL_SUM, c->comm,
                               (hipStream_t)stream));
    return 0;
}

int uniter_comm_broadcast(void* comm, void* buf, int64_t bytes, int32_t root, void* stream) {
    UH_CHECK_ARG(comm != nullptr && buf != nullptr && bytes > 0, "null pointer / empty buffer");
    Comm* c = (Comm*)comm;
    UH_CHECK_ARG(root >= 0 && root < c->world, "bad root rank");
    NCCL_CHECK(g_api.Broadcast(buf, buf, (size_t)bytes, RCCL_UINT8, root, c->comm, (hipStream_t)stream));
    return 0;
}

int uniter_comm_allgather(void* comm, const void* send, void* recv, int64_t bytes_per_rank, void* stream) {
    UH_CHECK_ARG(comm != nullptr && send != nullptr && recv != nullptr && bytes_per_rank > 0, "null pointer / empty buffer");
    Comm* c = (Comm*)comm;
    NCCL_CHECK(g_api.AllGather(send, recv, (size_t)bytes_per_rank, RCCL_UINT8, c->comm, (hipStream_t)stream));
    return 0;
}

}  // extern "C"
