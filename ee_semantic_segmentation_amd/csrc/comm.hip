// Data-parallel collectives of the training step: RCCL called DIRECTLY on a stream the caller owns.
//
// The reference has no collective at all (nn.DataParallel is commented out at train_funcs.py:72-74), so
// this replaces nothing in it: it is the exchange step of the build's own data-parallel path.
//
// Why not torch.distributed's ProcessGroupNCCL for the data path ("captured-event abort"): c10d
// records every collective's end event on ITS internal stream and a watchdog thread polls those events.  On HIP an
// event counts as "captured" as soon as the stream it was last recorded on is capturing, so a long-finished warm-up
// collective aborts the process once that internal stream is forked into the HIP-graph capture of the training step
// (scripts/captured_event_repro.hip).  Here there is no Work object, no watchdog and no hidden stream: the collective
// is one RCCL kernel enqueued on the stream passed in, eager or inside a capture alike.
//
// librccl is bound at run time (dlopen), so libeeseg.so itself has no link-time dependency on it: a process that has
// torch loaded gets torch's own copy (same HIP runtime), a plain C host gets /opt/rocm/lib/librccl.so.1.
#include "eeseg_common.h"

#include <dlfcn.h>
#include <string.h>

#include <mutex>

namespace {

typedef void* nccl_comm;
struct nccl_uid { char internal[128]; };
typedef int (*fn_get_version)(int*);
typedef int (*fn_get_unique_id)(nccl_uid*);
typedef int (*fn_comm_init_rank)(nccl_comm*, int, nccl_uid, int);
typedef int (*fn_comm_destroy)(nccl_comm);
typedef int (*fn_comm_async_error)(nccl_comm, int*);
typedef const char* (*fn_error_string)(int);
typedef int (*fn_all_reduce)(const void*, void*, size_t, int, int, nccl_comm, hipStream_t);
typedef int (*fn_all_gather)(const void*, void*, size_t, int, nccl_comm, hipStream_t);
typedef int (*fn_broadcast)(const void*, void*, size_t, int, int, nccl_comm, hipStream_t);
typedef int (*fn_reduce_scatter)(const void*, void*, size_t, int, int, nccl_comm, hipStream_t);

struct Rccl {
    void* handle = nullptr;
    fn_get_version get_version = nullptr;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_comm_async_error comm_async_error = nullptr;
    fn_error_string error_string = nullptr;
    fn_all_reduce all_reduce = nullptr;
    fn_all_gather all_gather = nullptr;
    fn_broadcast broadcast = nullptr;
    fn_reduce_scatter reduce_scatter = nullptr;
    char where[96] = "";
};

Rccl g_rccl;            // bound once; immutable afterwards (the "once-initialised cache" SURVEY 8b allows)
std::once_flag g_once;
char g_bind_error[256] = "";

void bind_rccl() {
    // already mapped by the host process (torch's libtorch_hip.so needs librccl.so)?  then use exactly that copy
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (int pass = 0; pass < 2 && !h; ++pass)
        for (const char* n : names) {
            h = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
            if (h) { snprintf(g_rccl.where, sizeof g_rccl.where, "%s%s", n, pass == 0 ? " (already loaded)" : ""); break; }
        }
    if (!h) { snprintf(g_bind_error, sizeof g_bind_error, "librccl.so.1 not found: %s", dlerror()); return; }
    Rccl r = g_rccl;
    r.handle = h;
#define EESEG_SYM(field, name)                                                                     \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, name));                                 \
    if (!r.field) { snprintf(g_bind_error, sizeof g_bind_error, "librccl: no symbol %s", name); return; }
    EESEG_SYM(get_version, "ncclGetVersion")
    EESEG_SYM(get_unique_id, "ncclGetUniqueId")
    EESEG_SYM(comm_init_rank, "ncclCommInitRank")
    EESEG_SYM(comm_destroy, "ncclCommDestroy")
    EESEG_SYM(comm_async_error, "ncclCommGetAsyncError")
    EESEG_SYM(error_string, "ncclGetErrorString")
    EESEG_SYM(all_reduce, "ncclAllReduce")
    EESEG_SYM(all_gather, "ncclAllGather")
    EESEG_SYM(broadcast, "ncclBroadcast")
    EESEG_SYM(reduce_scatter, "ncclReduceScatter")
#undef EESEG_SYM
    g_rccl = r;
}

const Rccl* rccl() {
    std::call_once(g_once, bind_rccl);
    return g_rccl.handle && g_rccl.broadcast ? &g_rccl : nullptr;
}

struct Comm {
    uint32_t magic;
    nccl_comm comm;
    int world, rank, device;
};
constexpr uint32_t kMagic = 0xEE5EC033u;

// RCCL's enums (rccl.h): ncclInt32 2, ncclInt64 4, ncclFloat32 7, ncclFloat64 8, ncclBfloat16 9; ncclSum 0, ncclMax 2, ncclAvg 4
int nccl_dtype(int dt) {
    switch (dt) {
        case EESEG_COMM_F32: return 7;
        case EESEG_COMM_BF16: return 9;
        case EESEG_COMM_F64: return 8;
        case EESEG_COMM_I32: return 2;
        case EESEG_COMM_I64: return 4;
        default: return -1;
    }
}
int nccl_op(int op) {
    switch (op) {
        case EESEG_COMM_SUM: return 0;
        case EESEG_COMM_AVG: return 4;
        case EESEG_COMM_MAX: return 2;
        default: return -1;
    }
}

}  // namespace

#define EESEG_RCCL(r, call, what)                                                              \
    do {                                                                                       \
        int rc_ = (call);                                                                      \
        if (rc_ != 0) {                                                                        \
            eeseg_set_error("%s failed: RCCL error %d (%s)", what, rc_, (r)->error_string(rc_)); \
            return EESEG_ERR_HIP;                                                              \
        }                                                                                      \
    } while (0)

#define EESEG_COMM_ARG(c, handle)                                                              \
    Comm* c = static_cast<Comm*>(handle);                                                      \
    EESEG_CHECK(c && c->magic == kMagic, EESEG_ERR_ARG, "not a live eeseg communicator handle"); \
    const Rccl* R = rccl();                                                                    \
    EESEG_CHECK(R, EESEG_ERR_HIP, "%s", g_bind_error)

extern "C" {

int eeseg_comm_available(int* rccl_version) {
    const Rccl* R = rccl();
    EESEG_CHECK(R, EESEG_ERR_HIP, "%s", g_bind_error);
    int v = 0;
    EESEG_RCCL(R, R->get_version(&v), "ncclGetVersion");
    if (rccl_version) *rccl_version = v;
    return EESEG_OK;
}

int eeseg_comm_unique_id(void* id128) {
    EESEG_CHECK(id128, EESEG_ERR_ARG, "comm_unique_id: NULL buffer");
    const Rccl* R = rccl();
    EESEG_CHECK(R, EESEG_ERR_HIP, "%s", g_bind_error);
    nccl_uid id;
    EESEG_RCCL(R, R->get_unique_id(&id), "ncclGetUniqueId");
    memcpy(id128, id.internal, sizeof id.internal);
    return EESEG_OK;
}

int eeseg_comm_create(const void* id128, int world, int rank, void** comm_out) {
    EESEG_CHECK(id128 && comm_out, EESEG_ERR_ARG, "comm_create: NULL argument");
    EESEG_CHECK(world >= 1 && rank >= 0 && rank < world, EESEG_ERR_ARG, "comm_create: rank %d of %d", rank, world);
    const Rccl* R = rccl();
    EESEG_CHECK(R, EESEG_ERR_HIP, "%s", g_bind_error);
    int dev = -1;
    EESEG_HIP(hipGetDevice(&dev));        // the communicator lives on the caller's current device
    nccl_uid id;
    memcpy(id.internal, id128, sizeof id.internal);
    nccl_comm c = nullptr;
    EESEG_RCCL(R, R->comm_init_rank(&c, world, id, rank), "ncclCommInitRank");
    *comm_out = new Comm{kMagic, c, world, rank, dev};
    return EESEG_OK;
}

int eeseg_comm_destroy(void* comm) {
    EESEG_COMM_ARG(c, comm);
    c->magic = 0;
    int rc = R->comm_destroy(c->comm);
    delete c;
    EESEG_CHECK(rc == 0, EESEG_ERR_HIP, "ncclCommDestroy failed: RCCL error %d (%s)", rc, R->error_string(rc));
    return EESEG_OK;
}

int eeseg_comm_info(void* comm, int* world, int* rank, int* device) {
    EESEG_COMM_ARG(c, comm);
    if (world) *world = c->world;
    if (rank) *rank = c->rank;
    if (device) *device = c->device;
    return EESEG_OK;
}

int eeseg_comm_check(void* comm) {
    EESEG_COMM_ARG(c, comm);
    int async = 0;
    EESEG_RCCL(R, R->comm_async_error(c->comm, &async), "ncclCommGetAsyncError");
    EESEG_CHECK(async == 0, EESEG_ERR_HIP, "communicator reports asynchronous RCCL error %d (%s)", async, R->error_string(async));
    return EESEG_OK;
}

int eeseg_comm_all_reduce(void* comm, void* buf, int64_t count, int dtype, int op, void* stream) {
    EESEG_COMM_ARG(c, comm);
    const int dt = nccl_dtype(dtype), o = nccl_op(op);
    EESEG_CHECK(dt >= 0 && o >= 0, EESEG_ERR_ARG, "comm_all_reduce: dtype %d / op %d", dtype, op);
    EESEG_CHECK(buf && count > 0, EESEG_ERR_ARG, "comm_all_reduce: empty buffer");
    EESEG_RCCL(R, R->all_reduce(buf, buf, (size_t)count, dt, o, c->comm, static_cast<hipStream_t>(stream)), "ncclAllReduce");
    return EESEG_OK;
}

int eeseg_comm_all_gather(void* comm, const void* send, void* recv, int64_t bytes_per_rank, void* stream) {
    EESEG_COMM_ARG(c, comm);
    EESEG_CHECK(send && recv && bytes_per_rank > 0, EESEG_ERR_ARG, "comm_all_gather: empty buffer");
    EESEG_RCCL(R, R->all_gather(send, recv, (size_t)bytes_per_rank, /*ncclUint8*/ 1, c->comm, static_cast<hipStream_t>(stream)),
               "ncclAllGather");
    return EESEG_OK;
}

int eeseg_comm_reduce_scatter(void* comm, const void* send, void* recv, int64_t count_per_rank, int dtype, int op, void* stream) {
    EESEG_COMM_ARG(c, comm);
    const int dt = nccl_dtype(dtype), o = nccl_op(op);
    EESEG_CHECK(dt >= 0 && o >= 0, EESEG_ERR_ARG, "comm_reduce_scatter: dtype %d / op %d", dtype, op);
    EESEG_CHECK(send && recv && count_per_rank > 0, EESEG_ERR_ARG, "comm_reduce_scatter: empty buffer");
    EESEG_RCCL(R, R->reduce_scatter(send, recv, (size_t)count_per_rank, dt, o, c->comm, static_cast<hipStream_t>(stream)),
               "ncclReduceScatter");
    return EESEG_OK;
}

int eeseg_comm_broadcast(void* comm, void* buf, int64_t bytes, int root, void* stream) {
    EESEG_COMM_ARG(c, comm);
    EESEG_CHECK(buf && bytes > 0 && root >= 0 && root < c->world, EESEG_ERR_ARG, "comm_broadcast: bad argument");
    EESEG_RCCL(R, R->broadcast(buf, buf, (size_t)bytes, /*ncclUint8*/ 1, root, c->comm, static_cast<hipStream_t>(stream)),
               "ncclBroadcast");
    return EESEG_OK;
}

}  // extern "C"
