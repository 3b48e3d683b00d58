// Collectives of the sharded-state mode issued from the library itself: RCCL calls on the context's own stream, no
// host callback, no stream synchronisation around them (the callback route of plm_ctx_set_collective costs two per
// collective).  One process per GPU; the host creates a 128-byte id on rank 0 (plm_rccl_unique_id), hands it to every
// rank by its own means (MPI, a file, torch.distributed) and each rank attaches it (plm_ctx_attach_rccl,
// plm_fit_sharded_rccl).  librccl is resolved at run time: a host that already has RCCL in the process (PyTorch ships
// its own copy) shares that copy, everything else loads /opt/rocm's; the library has no link-time dependency on it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "../../include/plm_hip.h"
#include "plm_internal.h"

namespace {

struct Api {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    std::string error;
};

Api &api() {
    static Api a;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *env = getenv("PLM_RCCL_LIB");
        // a copy that is already mapped (RTLD_NOLOAD matches loaded objects by soname) wins: two RCCL copies in one
        // process work, but double the bootstrap threads and the staging buffers
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        if (env && *env) a.handle = dlopen(env, RTLD_NOW | RTLD_LOCAL);
        for (int pass = 0; pass < 2 && !a.handle && !(env && *env); pass++)
            for (const char *n : names) {
                a.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
                if (a.handle) break;
            }
        if (!a.handle) {
            const char *e = dlerror();
            a.error = std::string("librccl not found (") + (e ? e : "no dlerror") + "); set PLM_RCCL_LIB";
            return;
        }
#define PLM_SYM(NAME)                                                                        \
    a.NAME = (decltype(a.NAME))dlsym(a.handle, "nccl" #NAME);                                \
    if (!a.NAME && a.error.empty()) a.error = "librccl lacks nccl" #NAME;
        PLM_SYM(GetUniqueId) PLM_SYM(CommInitRank) PLM_SYM(CommDestroy) PLM_SYM(AllReduce) PLM_SYM(Broadcast)
        PLM_SYM(Send) PLM_SYM(Recv) PLM_SYM(GroupStart) PLM_SYM(GroupEnd) PLM_SYM(GetErrorString) PLM_SYM(GetVersion)
#undef PLM_SYM
    });
    return a;
}

thread_local std::string g_err;
int rfail(const char *what, ncclResult_t r) {
    Api &a = api();
    g_err = std::string(what) + ": " + (a.GetErrorString ? a.GetErrorString(r) : "rccl error");
    return -1;
}

}  // namespace

struct PlmRccl {
    ncclComm_t comm = nullptr;
    int nranks = 0, rank = 0;
};

const char *plm_rccl_error() { return g_err.c_str(); }

int plm_rccl_id(void *id128) {
    Api &a = api();
    if (!a.error.empty()) { g_err = a.error; return -1; }
    static_assert(sizeof(ncclUniqueId) == PLM_RCCL_ID_BYTES, "id size");
    ncclUniqueId id;
    const ncclResult_t r = a.GetUniqueId(&id);
    if (r != ncclSuccess) return rfail("ncclGetUniqueId", r);
    memcpy(id128, &id, sizeof(id));
    return 0;
}

int plm_rccl_version() {
    Api &a = api();
    int v = 0;
    if (!a.error.empty() || a.GetVersion(&v) != ncclSuccess) return 0;
    return v;
}

// collective over all ranks: the calling thread's current HIP device is the rank's GPU
int plm_rccl_init(const void *id128, int nranks, int rank, PlmRccl **out) {
    Api &a = api();
    if (!a.error.empty()) { g_err = a.error; return -1; }
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    PlmRccl *p = new PlmRccl();
    p->nranks = nranks;
    p->rank = rank;
    const ncclResult_t r = a.CommInitRank(&p->comm, nranks, id, rank);
    if (r != ncclSuccess) {
        delete p;
        return rfail("ncclCommInitRank", r);
    }
    *out = p;
    return 0;
}

void plm_rccl_destroy(PlmRccl *p) {
    if (!p) return;
    if (p->comm) api().CommDestroy(p->comm);
    delete p;
}

// the operations of plm_collective_cb (include/plm_hip.h), enqueued on st; counts are bytes per rank
int plm_rccl_collective(PlmRccl *p, int op, void *send, void *recv, const int64_t *scounts, const int64_t *rcounts,
                        hipStream_t st) {
    Api &a = api();
    ncclResult_t r = ncclSuccess;
    switch (op) {
    case PLM_COLL_ALLREDUCE_F64:
        r = a.AllReduce(send, send, (size_t)scounts[0] / sizeof(double), ncclDouble, ncclSum, p->comm, st);
        return r == ncclSuccess ? 0 : rfail("ncclAllReduce(f64)", r);
    case PLM_COLL_ALLREDUCE_F32:
        r = a.AllReduce(send, send, (size_t)scounts[0] / sizeof(float), ncclFloat, ncclSum, p->comm, st);
        return r == ncclSuccess ? 0 : rfail("ncclAllReduce(f32)", r);
    case PLM_COLL_BROADCAST:
        r = a.Broadcast(send, send, (size_t)scounts[0], ncclInt8, (int)rcounts[0], p->comm, st);
        return r == ncclSuccess ? 0 : rfail("ncclBroadcast", r);
    case PLM_COLL_ALLTOALL: {
        // messages in rank order on both sides; xGMI is point to point, so the n-1 transfers of a rank run on
        // different links at once -- one group, RCCL schedules them
        size_t soff = 0, roff = 0;
        if ((r = a.GroupStart()) != ncclSuccess) return rfail("ncclGroupStart", r);
        for (int k = 0; k < p->nranks && r == ncclSuccess; k++) {
            if (k == p->rank) {
                if (scounts[k] != rcounts[k]) {
                    a.GroupEnd();
                    g_err = "all-to-all: a rank's message to itself has different send and receive sizes";
                    return -1;
                }
                if (scounts[k] > 0 &&
                    hipMemcpyAsync((char *)recv + roff, (const char *)send + soff, (size_t)scounts[k],
                                   hipMemcpyDeviceToDevice, st) != hipSuccess) {
                    a.GroupEnd();
                    g_err = "all-to-all: device copy of the local message failed";
                    return -1;
                }
            } else {
                if (scounts[k] > 0) r = a.Send((const char *)send + soff, (size_t)scounts[k], ncclInt8, k, p->comm, st);
                if (r == ncclSuccess && rcounts[k] > 0)
                    r = a.Recv((char *)recv + roff, (size_t)rcounts[k], ncclInt8, k, p->comm, st);
            }
            soff += (size_t)scounts[k];
            roff += (size_t)rcounts[k];
        }
        const ncclResult_t e = a.GroupEnd();
        if (r != ncclSuccess) return rfail("ncclSend/ncclRecv", r);
        return e == ncclSuccess ? 0 : rfail("ncclGroupEnd", e);
    }
    default:
        g_err = "unknown collective";
        return -1;
    }
}
