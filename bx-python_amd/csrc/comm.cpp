// comm.cpp -- the path's only collective in the C ABI: an int64 sum all-reduce over RCCL (xGMI inside a node).
//
// The reference keeps one tree / bitset per chromosome (scripts/interval_join.py:21-28, lib/bx/bitset_builders.py:31-45);
// sharded by chromosome over the GPUs of a node, the only thing the ranks ever exchange is the vector of per-chromosome
// overlap totals.  A host that is not Python (the Cython extension of INTEGRATION.md, a C driver) reaches it here:
//
//     rank 0:  bxmi_comm_unique_id(id)              -> 128 bytes, handed to the other ranks by whatever launched them
//     all:     bxmi_comm_create(&c, id, rank, world)   (RCCL communicator on the CURRENT device, one process per GPU)
//              bxmi_allreduce_i64(c, buf_dev, n, stream)   in place, stream-ordered
//              bxmi_comm_destroy(c)
//
// librccl.so is opened on first use (dlopen): libbxmi itself does not link against it, so everything else keeps loading
// on a box without RCCL, and the call fails loudly there.
#include <dlfcn.h>

#include <mutex>

#include "common.hpp"

namespace {

constexpr int kUniqueIdBytes = 128;  // NCCL_UNIQUE_ID_BYTES (rccl.h:40)
struct UniqueId {
    char internal[kUniqueIdBytes];
};
typedef void *Comm;
// rccl.h: ncclResult_t 0 = success; ncclInt64 = 4, ncclSum = 0
typedef int (*GetUniqueIdFn)(UniqueId *);
typedef int (*CommInitRankFn)(Comm *, int, UniqueId, int);
typedef int (*CommDestroyFn)(Comm);
typedef int (*AllReduceFn)(const void *, void *, size_t, int, int, Comm, hipStream_t);
typedef const char *(*GetErrorStringFn)(int);

struct Rccl {
    void *lib = nullptr;
    GetUniqueIdFn get_unique_id = nullptr;
    CommInitRankFn comm_init_rank = nullptr;
    CommDestroyFn comm_destroy = nullptr;
    AllReduceFn all_reduce = nullptr;
    GetErrorStringFn error_string = nullptr;
    std::string why;
};

Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // First choice: the librccl that sits beside the HIP runtime libbxmi itself is bound to.  A process can hold two HIP
        // runtimes (a PyTorch wheel brings its own libamdhip64 and librccl; libbxmi links against /opt/rocm's): a communicator
        // and the streams / buffers handed to it must belong to the same one, and a bare dlopen("librccl.so") returns whichever
        // copy the process loaded first (found by the round-6 GPU suite: torch imported before libbxmi -> "unhandled cuda error").
        std::string beside;
        Dl_info info;
        if (dladdr(reinterpret_cast<const void *>(&hipGetDeviceCount), &info) && info.dli_fname) {
            beside = info.dli_fname;
            const size_t slash = beside.rfind('/');
            beside = slash == std::string::npos ? std::string() : beside.substr(0, slash + 1) + "librccl.so";
        }
        for (const char *name : {beside.c_str(), "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            if (!*name) continue;
            r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
        }
        if (!r.lib) {
            const char *e = dlerror();
            r.why = e ? e : "dlopen failed";
            return;
        }
        r.get_unique_id = reinterpret_cast<GetUniqueIdFn>(dlsym(r.lib, "ncclGetUniqueId"));
        r.comm_init_rank = reinterpret_cast<CommInitRankFn>(dlsym(r.lib, "ncclCommInitRank"));
        r.comm_destroy = reinterpret_cast<CommDestroyFn>(dlsym(r.lib, "ncclCommDestroy"));
        r.all_reduce = reinterpret_cast<AllReduceFn>(dlsym(r.lib, "ncclAllReduce"));
        r.error_string = reinterpret_cast<GetErrorStringFn>(dlsym(r.lib, "ncclGetErrorString"));
        if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_reduce) r.why = "librccl.so lacks the nccl* entry points";
    });
    return r;
}

int need_rccl(const char *who)
{
    Rccl &r = rccl();
    if (!r.why.empty()) return bxmi::fail(BXMI_EHIP, "%s: RCCL is not available (%s)", who, r.why.c_str());
    return BXMI_OK;
}

int rccl_fail(const char *what, int rc)
{
    Rccl &r = rccl();
    return bxmi::fail(BXMI_EHIP, "%s: %s", what, r.error_string ? r.error_string(rc) : "RCCL error");
}

}  // namespace

struct bxmi_comm {
    Comm comm = nullptr;
    int rank = 0, world = 1;
};

extern "C" int bxmi_comm_unique_id(void *id128)
{
    if (!id128) return bxmi::fail(BXMI_EINVAL, "bxmi_comm_unique_id: id is NULL");
    BXMI_TRY(need_rccl("bxmi_comm_unique_id"));
    UniqueId id;
    const int rc = rccl().get_unique_id(&id);
    if (rc != 0) return rccl_fail("ncclGetUniqueId", rc);
    memcpy(id128, id.internal, kUniqueIdBytes);
    return BXMI_OK;
}

extern "C" int bxmi_comm_create(bxmi_comm_t **out, const void *id128, int rank, int world)
{
    if (!out || !id128 || world < 1 || rank < 0 || rank >= world) return bxmi::fail(BXMI_EINVAL, "bxmi_comm_create: bad arguments");
    BXMI_TRY(need_rccl("bxmi_comm_create"));
    UniqueId id;
    memcpy(id.internal, id128, kUniqueIdBytes);
    bxmi_comm *c = new (std::nothrow) bxmi_comm();
    if (!c) return bxmi::fail(BXMI_ENOMEM, "bxmi_comm_create: host allocation failed");
    const int rc = rccl().comm_init_rank(&c->comm, world, id, rank);
    if (rc != 0) {
        delete c;
        return rccl_fail("ncclCommInitRank", rc);
    }
    c->rank = rank, c->world = world;
    *out = c;
    return BXMI_OK;
}

extern "C" int bxmi_comm_destroy(bxmi_comm_t *c)
{
    if (!c) return BXMI_OK;
    int rc = 0;
    if (c->comm) rc = rccl().comm_destroy(c->comm);
    delete c;
    return rc == 0 ? BXMI_OK : rccl_fail("ncclCommDestroy", rc);
}

extern "C" int bxmi_allreduce_i64(bxmi_comm_t *c, int64_t *buf_dev, int64_t n, void *stream)
{
    if (!c || !c->comm || n < 0 || (n > 0 && !buf_dev)) return bxmi::fail(BXMI_EINVAL, "bxmi_allreduce_i64: bad arguments");
    // (n must be the same on every rank, as for any collective: with n == 0 everywhere nobody enters RCCL)
    if (n == 0) return BXMI_OK;
    const int rc = rccl().all_reduce(buf_dev, buf_dev, (size_t)n, /* ncclInt64 */ 4, /* ncclSum */ 0, c->comm, bxmi::as_stream(stream));
    return rc == 0 ? BXMI_OK : rccl_fail("ncclAllReduce", rc);
}
