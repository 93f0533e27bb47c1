// RCCL behind the C ABI: the gradient all-reduce of data-parallel training and the all-gather of walker / chain
// state (SURVEY 8 b6 / e), over xGMI on one node.  librccl is resolved at run time (dlopen of the soname, so a
// process that already holds RCCL -- e.g. through PyTorch -- shares that copy and a build box without a GPU can still
// load this library); the communicator lives in the per-device context and every collective is enqueued on the
// caller's stream like any other entry.
#include "common.h"

#include <dlfcn.h>
#include <string.h>
#include <mutex>

namespace linna {

// the part of rccl.h this file needs (RCCL keeps NCCL's ABI: result / datatype / op enums, 128-byte unique id)
typedef struct { char internal[LINNA_COMM_ID_BYTES]; } UniqueId;
typedef void* Comm;
enum { kSuccess = 0 };
enum { kFloat32 = 7 };       // ncclFloat32
enum { kSum = 0 };           // ncclSum

struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, Comm, hipStream_t) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*GetVersion)(int*) = nullptr;
};

static Rccl g_rccl;
static std::once_flag g_rccl_once;
static int g_rccl_rc = LINNA_ERR_UNSUPPORTED;

static void rccl_load() {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
    }
    if (!h) { set_error("RCCL: cannot load librccl.so.1 (%s)", dlerror()); return; }
    g_rccl.lib = h;
#define SYM(field, name) \
    *reinterpret_cast<void**>(&g_rccl.field) = dlsym(h, name); \
    if (!g_rccl.field) { set_error("RCCL: symbol %s missing", name); return; }
    SYM(GetUniqueId, "ncclGetUniqueId")
    SYM(CommInitRank, "ncclCommInitRank")
    SYM(CommDestroy, "ncclCommDestroy")
    SYM(AllReduce, "ncclAllReduce")
    SYM(AllGather, "ncclAllGather")
    SYM(Broadcast, "ncclBroadcast")
    SYM(GetErrorString, "ncclGetErrorString")
    SYM(GetVersion, "ncclGetVersion")
#undef SYM
    g_rccl_rc = LINNA_OK;
}

static int rccl() {
    std::call_once(g_rccl_once, rccl_load);
    if (g_rccl_rc != LINNA_OK && !g_rccl.lib) set_error("RCCL: librccl.so.1 not available in this process");
    return g_rccl_rc;
}

static int check_rccl(int r, const char* what) {
    if (r == kSuccess) return LINNA_OK;
    set_error("%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error");
    return LINNA_ERR_HIP;
}

struct CommState { Comm comm = nullptr; int rank = 0, nranks = 1; };

}  // namespace linna

using namespace linna;

// the context owns the communicator (api.hip: linna_ctx::comm, released by linna_ctx_destroy through linna_comm_destroy)
void** linna_ctx_comm_slot(linna_ctx_t* ctx);
int linna_ctx_device(const linna_ctx_t* ctx);

#define TRYC(expr) do { int rc__ = (expr); if (rc__ != LINNA_OK) return rc__; } while (0)

static CommState* state_of(linna_ctx_t* ctx) {
    return ctx ? static_cast<CommState*>(*linna_ctx_comm_slot(ctx)) : nullptr;
}

int linna_comm_unique_id(void* id) try {
    if (!id) { set_error("comm_unique_id: null id"); return LINNA_ERR_INVALID; }
    TRYC(rccl());
    UniqueId u;
    TRYC(check_rccl(g_rccl.GetUniqueId(&u), "ncclGetUniqueId"));
    memcpy(id, u.internal, LINNA_COMM_ID_BYTES);
    return LINNA_OK;
} LINNA_CATCH_INT

int linna_comm_init(linna_ctx_t* ctx, int rank, int nranks, const void* id) try {
    if (!ctx || !id || nranks < 1 || rank < 0 || rank >= nranks) {
        set_error("comm_init: bad arguments (rank %d of %d)", rank, nranks);
        return LINNA_ERR_INVALID;
    }
    if (state_of(ctx)) { set_error("comm_init: this context already holds a communicator"); return LINNA_ERR_INVALID; }
    TRYC(rccl());
    TRYC(check_hip(hipSetDevice(linna_ctx_device(ctx)), "hipSetDevice"));
    UniqueId u;
    memcpy(u.internal, id, LINNA_COMM_ID_BYTES);
    Comm c = nullptr;
    TRYC(check_rccl(g_rccl.CommInitRank(&c, nranks, u, rank), "ncclCommInitRank"));
    CommState* st = new (std::nothrow) CommState();
    if (!st) { (void)g_rccl.CommDestroy(c); return LINNA_ERR_INVALID; }
    st->comm = c; st->rank = rank; st->nranks = nranks;
    *linna_ctx_comm_slot(ctx) = st;
    return LINNA_OK;
} LINNA_CATCH_INT

int linna_comm_destroy(linna_ctx_t* ctx) try {
    CommState* st = state_of(ctx);
    if (!st) return LINNA_OK;
    int rc = LINNA_OK;
    if (st->comm && g_rccl.CommDestroy) rc = check_rccl(g_rccl.CommDestroy(st->comm), "ncclCommDestroy");
    delete st;
    *linna_ctx_comm_slot(ctx) = nullptr;
    return rc;
} LINNA_CATCH_INT

int linna_comm_info(linna_ctx_t* ctx, int* rank, int* nranks, int* rccl_version) try {
    CommState* st = state_of(ctx);
    if (rank) *rank = st ? st->rank : 0;
    if (nranks) *nranks = st ? st->nranks : 0;           // 0: no communicator
    if (rccl_version) {
        *rccl_version = 0;
        if (rccl() == LINNA_OK) (void)g_rccl.GetVersion(rccl_version);
    }
    return LINNA_OK;
} LINNA_CATCH_INT

int linna_allreduce_sum_f32(linna_ctx_t* ctx, float* buf, size_t n, void* stream) try {
    CommState* st = state_of(ctx);
    if (!st) { set_error("allreduce_sum_f32: linna_comm_init has not been called on this context"); return LINNA_ERR_INVALID; }
    if (n == 0) return LINNA_OK;
    if (!buf) { set_error("allreduce_sum_f32: null buffer"); return LINNA_ERR_INVALID; }
    return check_rccl(g_rccl.AllReduce(buf, buf, n, kFloat32, kSum, st->comm, reinterpret_cast<hipStream_t>(stream)), "ncclAllReduce");
} LINNA_CATCH_INT

int linna_allgather_f32(linna_ctx_t* ctx, const float* send, float* recv, size_t n_per_rank, void* stream) try {
    CommState* st = state_of(ctx);
    if (!st) { set_error("allgather_f32: linna_comm_init has not been called on this context"); return LINNA_ERR_INVALID; }
    if (n_per_rank == 0) return LINNA_OK;
    if (!send || !recv) { set_error("allgather_f32: null buffer"); return LINNA_ERR_INVALID; }
    return check_rccl(g_rccl.AllGather(send, recv, n_per_rank, kFloat32, st->comm, reinterpret_cast<hipStream_t>(stream)), "ncclAllGather");
} LINNA_CATCH_INT

int linna_broadcast_f32(linna_ctx_t* ctx, float* buf, size_t n, int root, void* stream) try {
    CommState* st = state_of(ctx);
    if (!st) { set_error("broadcast_f32: linna_comm_init has not been called on this context"); return LINNA_ERR_INVALID; }
    if (n == 0) return LINNA_OK;
    if (!buf || root < 0 || root >= st->nranks) { set_error("broadcast_f32: bad arguments"); return LINNA_ERR_INVALID; }
    return check_rccl(g_rccl.Broadcast(buf, buf, n, kFloat32, root, st->comm, reinterpret_cast<hipStream_t>(stream)), "ncclBroadcast");
} LINNA_CATCH_INT
