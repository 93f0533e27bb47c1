// C-ABI entry points of liblinna_hip.so (declared in include/linna_hip.h) and the host-side
// orchestration above the kernels: network forward/backward as chains of fused GEMMs, the
// serving pipeline (Log_prob), the training loss.  No device allocation, no sync.
#include "common.h"

#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <cstring>
#include <string>
#include <vector>
#include <atomic>
#include <new>
#include <stdlib.h>

namespace linna {

static thread_local std::string g_err;

void set_error(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}

int caught_exception(const char* what) noexcept {
    // (the text is built in a fixed buffer; if even the assignment cannot allocate, the previous text stays)
    try { set_error("C++ exception at the C boundary: %s", what ? what : "unknown type"); } catch (...) {}
    return LINNA_ERR_INTERNAL;
}

int check_hip(hipError_t e, const char* what) {
    if (e == hipSuccess) return LINNA_OK;
    set_error("%s: %s", what, hipGetErrorString(e));
    return LINNA_ERR_HIP;
}

static inline int ld4(int w) { return (w + 3) & ~3; }
static inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }

#define TRY(expr) do { int rc__ = (expr); if (rc__ != LINNA_OK) return rc__; } while (0)

// a descriptor struct filled against another layout of include/linna_hip.h is refused, not read at wrong offsets
#define CHECK_STRUCT(ptr, T, who) do { if ((ptr)->struct_size != sizeof(T)) { \
        set_error("%s: " #T "::struct_size is %u, this library's sizeof is %zu -- set it to sizeof(" #T ") of the header you built against; " \
                  "a mismatch means that header is not this library's (LINNA_ABI_VERSION %d)", who, (unsigned)(ptr)->struct_size, sizeof(T), LINNA_ABI_VERSION); \
        return LINNA_ERR_INVALID; } } while (0)
static GemmArgs gemm_zero() {
    GemmArgs a;
    std::memset(&a, 0, sizeof a);
    a.struct_size = (uint32_t)sizeof(GemmArgs);
    a.npairs = 1;
    a.alpha0 = 1.f;
    return a;
}
static void set_pair(GemmArgs& a, int i, const float* A, int lda, int alay, const float* B, int ldb, int blay, int K) {
    a.p[i].A = A; a.p[i].lda = lda; a.p[i].alay = alay;
    a.p[i].B = B; a.p[i].ldb = ldb; a.p[i].blay = blay; a.p[i].K = K;
}

}  // namespace linna

using namespace linna;

// Weight epoch: bumped by every entry that may change network parameters (linna_adamw_step, a
// graph replay, linna_weights_changed()); a log-probability object re-lays its fragment-order
// weight copy (net_stream.hip) when the epoch moved since the copy was made.
static std::atomic<unsigned long long> g_weights_epoch{1};

struct linna_ctx {
    int device;
    hipStream_t aux = nullptr;               // second stream: parameter-gradient GEMMs run beside the dX chain
    std::vector<hipEvent_t> events;          // fork/join markers (no timing)
    int overlap = -1;                        // -1 unknown, 0 off (env LINNA_BWD_STREAMS=0), 1 on
    int group = -1;                          // -1 unknown, 0 off (env LINNA_BWD_GROUP=0), 1 on
    unsigned* counters = nullptr;            // zeroed, self-resetting arrival counters (fused loss)
    int loss_fused = -1;                     // -1 unknown, 0 off (env LINNA_LOSS_FUSED=0), 1 on
    void* comm = nullptr;                    // RCCL communicator state (comm.hip), set by linna_comm_init
};
void** linna_ctx_comm_slot(linna_ctx_t* ctx) { return &ctx->comm; }
int linna_ctx_device(const linna_ctx_t* ctx) { return ctx->device; }
struct linna_graph { hipGraph_t graph; hipGraphExec_t exec; };
// zeroed, self-resetting arrival counters of the context (allocated by linna_ctx_create)
static unsigned* ctx_counters(linna_ctx* ctx, hipStream_t) { return ctx ? ctx->counters : nullptr; }

// The whole-network kernel (net_stream.hip) reads the weights from a copy in MFMA fragment order.  Its 16-row
// engine and its small-batch engines (8 / 4 rows per workgroup) read different orders, so there are two copies,
// each re-laid lazily when the weights moved (g_weights_epoch) since it was made.
struct StreamCopy {
    float* buf[2] = {nullptr, nullptr};      // [0]: 16-row engine, [1]: small-batch engines
    unsigned long long epoch[2] = {0, 0};    // epoch each copy was made at (0 = never)
    size_t floats = 0;
    bool ready() const { return buf[0] && buf[1]; }
    int alloc(size_t nf) {
        floats = nf;
        for (int k = 0; k < 2; ++k)
            if (hipMalloc(reinterpret_cast<void**>(&buf[k]), nf * sizeof(float)) != hipSuccess) { release(); return LINNA_ERR_HIP; }
        return LINNA_OK;
    }
    void release() {
        for (int k = 0; k < 2; ++k) { if (buf[k]) (void)hipFree(buf[k]); buf[k] = nullptr; epoch[k] = 0; }
    }
};

struct linna_net {
    linna_ctx* ctx;

    StreamCopy packed;                       // fragment-order weight streams for the one-launch training forward
    int stream_fwd = -1;
    StreamCopy packed_loss;                  // ... for forward + chi^2-ratio loss in one launch (linna_net_forward_loss)
    NsDense loss_dn{nullptr, 0, nullptr, nullptr};   // the inverse covariance that stream ends in
    int stream_loss = -1;                    // -1 unknown, 0 no (not eligible / LINNA_LOSS_STREAM=0), 1 yes
    StreamCopy packed_dx[2];                 // ... for the one-launch dX chain of the backward ([1]: down to the network input)
    StreamCopy packed_tb;                    // ... for forward + loss + dX chain in ONE launch (linna_net_train_step on the small-batch engines)
    int stream_tb = -1;                      // -1 unknown, 0 no (not eligible / LINNA_TRAIN_MERGED=0), 1 yes
    int as_merged = -1;                      // which streams as_args describes: 1 = packed_tb, 0 = packed_loss + packed_dx[0]
    AsArgs as_args;                          // linna_net_adamw_step's descriptor table, valid for (as_params, as_n, as_k)
    const float* as_params = nullptr; size_t as_n = 0; int as_k = -1; int as_state = -1;   // as_state: -1 unknown, 0 unsupported, 1 ready
    int upd_state = -1; int upd_B = 0;       // linna_net_train_step_update: -1 unknown, 0 unsupported, 1 every parameter gradient of the step is in the grouped launch
    int stream_bwd[2] = {-1, -1};            // -1 unknown, 0 no (network out of reach / LINNA_BWD_STREAM=0), 1 yes                     // -1 unknown, 0 no (network out of reach / LINNA_FWD_STREAM=0), 1 yes
    std::vector<linna_layer_t> L;   // without the trailing INSKIP
    std::vector<linna_layer_t> Lfull;   // with it: what the serving programs of the whole-network kernel are built from
    int in_size, out_size;
    bool has_inskip;
    linna_layer_t inskip;
    int max_w, max_c;
};

static int stream_copy_refresh(StreamCopy& sc, const linna_net* n, int rows, void* stream, const float** out, int prog = 0,
                               const NsDense* dn = nullptr, int serve = 0);

struct FwdLayout {
    std::vector<size_t> t_off, y_off;   // float offsets, per op (y_off of the last op unused)
    size_t total;
};

static FwdLayout fwd_layout(const linna_net* n, int B) {
    FwdLayout f;
    size_t off = 0;
    const int nl = (int)n->L.size();
    f.t_off.assign(nl, 0);
    f.y_off.assign(nl, 0);
    for (int i = 0; i < nl; ++i) {
        const linna_layer_t& l = n->L[i];
        if (l.op == LINNA_OP_RESBLOCK) { f.t_off[i] = off; off += (size_t)B * ld4(l.C); }
        if (i + 1 < nl) { f.y_off[i] = off; off += (size_t)B * ld4(l.N); }
    }
    f.total = off;
    return f;
}

extern "C" {

// ------------------------------------------------------------------ runtime
int linna_abi_version(void) { return LINNA_ABI_VERSION; }
const char* linna_last_error(void) { return g_err.c_str(); }
// diagnostic: raise inside a guarded entry (tests/test_abi.py checks that the barrier turns it into a code and a text)
int linna_debug_raise(int kind) try {
    if (kind == 1) throw std::bad_alloc();
    if (kind == 2) { std::vector<int> v; v.reserve(v.max_size() + 1); }      // std::length_error, as a planner's vector would
    if (kind == 3) throw 42;                                                  // not derived from std::exception
    if (kind == 4) { std::string s; (void)s.at(7); }                          // std::out_of_range
    return LINNA_OK;
} LINNA_CATCH_INT

int linna_ctx_create(int device, linna_ctx_t** out) try {
    if (!out) { set_error("ctx_create: null out"); return LINNA_ERR_INVALID; }
    int n = 0;
    TRY(check_hip(hipGetDeviceCount(&n), "hipGetDeviceCount"));
    if (device < 0 || device >= n) { set_error("ctx_create: device %d of %d", device, n); return LINNA_ERR_INVALID; }
    hipDeviceProp_t prop;
    TRY(check_hip(hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties"));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
        set_error("ctx_create: device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
        return LINNA_ERR_UNSUPPORTED;
    }
    linna_ctx* c = new (std::nothrow) linna_ctx();
    if (!c) return LINNA_ERR_INVALID;
    c->device = device;
    // the context's device-side state is allocated HERE, so that no launch ever allocates: the arrival counters of the
    // one-launch loss
    int prev = 0;
    (void)hipGetDevice(&prev);
    int rc = check_hip(hipSetDevice(device), "hipSetDevice");
    if (rc == LINNA_OK) rc = check_hip(hipMalloc(reinterpret_cast<void**>(&c->counters), 64), "hipMalloc(counters)");
    if (rc == LINNA_OK) rc = check_hip(hipMemset(c->counters, 0, 64), "hipMemset(counters)");
    (void)hipSetDevice(prev);
    if (rc != LINNA_OK) { (void)linna_ctx_destroy(c); return rc; }
    *out = c;
    return LINNA_OK;
} LINNA_CATCH_INT
int linna_ctx_destroy(linna_ctx_t* ctx) try {
    if (ctx) {
        (void)linna_comm_destroy(ctx);
        for (hipEvent_t e : ctx->events) (void)hipEventDestroy(e);
        if (ctx->aux) (void)hipStreamDestroy(ctx->aux);
        if (ctx->counters) (void)hipFree(ctx->counters);
    }
    delete ctx;
    return LINNA_OK;
} LINNA_CATCH_INT
int linna_stream_sync(void* stream) try { return check_hip(hipStreamSynchronize(S(stream)), "hipStreamSynchronize"); } LINNA_CATCH_INT

int linna_graph_begin(void* stream) try {
    return check_hip(hipStreamBeginCapture(S(stream), hipStreamCaptureModeThreadLocal), "hipStreamBeginCapture");
} LINNA_CATCH_INT
int linna_graph_end(void* stream, linna_graph_t** out) try {
    hipGraph_t g = nullptr;
    TRY(check_hip(hipStreamEndCapture(S(stream), &g), "hipStreamEndCapture"));
    hipGraphExec_t e = nullptr;
    int rc = check_hip(hipGraphInstantiate(&e, g, nullptr, nullptr, 0), "hipGraphInstantiate");
    if (rc != LINNA_OK) { (void)hipGraphDestroy(g); return rc; }
    *out = new linna_graph{g, e};
    return LINNA_OK;
} LINNA_CATCH_INT
int linna_graph_launch(linna_graph_t* g, void* stream) try {
    if (!g) { set_error("graph_launch: null graph"); return LINNA_ERR_INVALID; }
    g_weights_epoch.fetch_add(1);            // the graph may hold an AdamW step
    return check_hip(hipGraphLaunch(g->exec, S(stream)), "hipGraphLaunch");
} LINNA_CATCH_INT
int linna_graph_destroy(linna_graph_t* g) try {
    if (!g) return LINNA_OK;
    (void)hipGraphExecDestroy(g->exec);
    (void)hipGraphDestroy(g->graph);
    delete g;
    return LINNA_OK;
} LINNA_CATCH_INT
int linna_event_create(void** ev) try {
    hipEvent_t e;
    TRY(check_hip(hipEventCreate(&e), "hipEventCreate"));
    *ev = e;
    return LINNA_OK;
} LINNA_CATCH_INT
int linna_event_record(void* ev, void* stream) try { return check_hip(hipEventRecord((hipEvent_t)ev, S(stream)), "hipEventRecord"); } LINNA_CATCH_INT
int linna_event_elapsed_ms(void* a, void* b, float* ms) try {
    TRY(check_hip(hipEventSynchronize((hipEvent_t)b), "hipEventSynchronize"));
    return check_hip(hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b), "hipEventElapsedTime");
} LINNA_CATCH_INT
int linna_event_destroy(void* ev) try { return check_hip(hipEventDestroy((hipEvent_t)ev), "hipEventDestroy"); } LINNA_CATCH_INT

// ------------------------------------------------------------------ GEMM
int linna_gemm_f32(linna_ctx_t*, const linna_gemm_t* d, void* stream) try {
    if (!d) { set_error("gemm: null descriptor"); return LINNA_ERR_INVALID; }
    CHECK_STRUCT(d, linna_gemm_t, "gemm");
    return gemm_launch(*d, S(stream));
} LINNA_CATCH_INT
int linna_gemm_dot_slots(int M, int N) try { return gemm_slots(M, N); } LINNA_CATCH_INT

// ------------------------------------------------------------------ layers
int linna_linear_fwd(linna_ctx_t*, const float* X, int ldx, const float* W, int ldw, const float* b, float* Y, int ldy,
                     int B, int K, int N, int relu, float alpha, const float* R, int ldr, void* stream) try {
    GemmArgs a = gemm_zero();
    set_pair(a, 0, X, ldx, LAY_K, W, ldw, LAY_K, K);
    a.M = B; a.N = N; a.C = Y; a.ldc = ldy; a.bias0 = b; a.alpha0 = alpha; a.R = R; a.ldr = ldr; a.relu = relu;
    return gemm_launch(a, S(stream));
} LINNA_CATCH_INT

int linna_resblock_fwd(linna_ctx_t*, const float* X, int ldx, const float* W1, const float* b1, const float* W2,
                       const float* b2, const float* Ws, float* T, int ldt, float* Y, int ldy, int B, int K, int C,
                       int N, void* stream) try {
    if (!Ws && K != N) { set_error("resblock: identity skip needs K == N"); return LINNA_ERR_INVALID; }
    TRY(linna_linear_fwd(nullptr, X, ldx, W1, ld4(K), b1, T, ldt, B, K, C, 1, 1.f, nullptr, 0, stream));
    GemmArgs a = gemm_zero();
    set_pair(a, 0, T, ldt, LAY_K, W2, ld4(C), LAY_K, C);
    a.M = B; a.N = N; a.C = Y; a.ldc = ldy; a.bias0 = b2; a.alpha0 = 0.1f; a.relu = 1;
    if (Ws) { a.npairs = 2; set_pair(a, 1, X, ldx, LAY_K, Ws, ld4(K), LAY_K, K); }
    else { a.R = X; a.ldr = ldx; }
    return gemm_launch(a, S(stream));
} LINNA_CATCH_INT

int linna_linear_bwd(linna_ctx_t*, const float* dY, int lddy, const float* X, int ldx, const float* W, int ldw,
                     float* dX, int lddx, const float* Xmask, int ldxm, float* dW, int lddw, float* db, int B, int K,
                     int N, float scale, void* stream) try {
    if (dW) {   // dW[n][k] = scale * sum_b dY[b][n] X[b][k]
        GemmArgs a = gemm_zero();
        set_pair(a, 0, dY, lddy, LAY_MN, X, ldx, LAY_MN, B);
        a.M = N; a.N = K; a.C = dW; a.ldc = lddw; a.alpha0 = scale;
        TRY(gemm_launch(a, S(stream)));
    }
    if (db) TRY(launch_colsum(dY, lddy, B, N, scale, db, S(stream)));
    if (dX) {   // dX[b][k] = scale * sum_n dY[b][n] W[n][k]
        GemmArgs a = gemm_zero();
        set_pair(a, 0, dY, lddy, LAY_K, W, ldw, LAY_MN, N);
        a.M = B; a.N = K; a.C = dX; a.ldc = lddx; a.alpha0 = scale; a.mask = Xmask; a.ldmask = ldxm;
        TRY(gemm_launch(a, S(stream)));
    }
    return LINNA_OK;
} LINNA_CATCH_INT

// ------------------------------------------------------------------ network
int linna_net_create(linna_ctx_t* ctx, const linna_layer_t* layers, int nlayers, int in_size, linna_net_t** out) try {
    if (!layers || nlayers < 1 || !out) { set_error("net_create: bad arguments"); return LINNA_ERR_INVALID; }
    CHECK_STRUCT(layers, linna_layer_t, "net_create");          // (the first entry's size is the array's stride: check it before walking)
    for (int i = 1; i < nlayers; ++i) CHECK_STRUCT(layers + i, linna_layer_t, "net_create");
    linna_net* n = new linna_net();
    n->ctx = ctx; n->in_size = in_size; n->has_inskip = false; n->max_w = in_size; n->max_c = 4;
    int width = in_size;
    for (int i = 0; i < nlayers; ++i) {
        const linna_layer_t& l = layers[i];
        if (l.op == LINNA_OP_INSKIP) {
            if (i != nlayers - 1 || l.K != in_size || l.N != width) {
                set_error("net_create: INSKIP must be the last op, K = in_size, N = network width");
                delete n; return LINNA_ERR_INVALID;
            }
            n->has_inskip = true; n->inskip = l;
            continue;
        }
        if (l.K != width) { set_error("net_create: op %d expects K=%d, got %d", i, width, l.K); delete n; return LINNA_ERR_INVALID; }
        if (l.op == LINNA_OP_RESBLOCK) {
            if (!l.Ws && l.K != l.N) { set_error("net_create: op %d identity skip with K != N", i); delete n; return LINNA_ERR_INVALID; }
            if (l.C > n->max_c) n->max_c = l.C;
        } else if (l.op != LINNA_OP_LINEAR) { set_error("net_create: unknown op %d", l.op); delete n; return LINNA_ERR_INVALID; }
        width = l.N;
        if (width > n->max_w) n->max_w = width;
        n->L.push_back(l);
    }
    const linna_layer_t& last = n->L.back();
    if (last.op != LINNA_OP_LINEAR || last.relu) {
        set_error("net_create: the last op must be a LINEAR without ReLU"); delete n; return LINNA_ERR_INVALID;
    }
    if (n->has_inskip && n->L[0].op != LINNA_OP_LINEAR) {
        set_error("net_create: INSKIP needs a LINEAR first op"); delete n; return LINNA_ERR_INVALID;
    }
    n->out_size = width;
    n->Lfull = n->L;
    if (n->has_inskip) n->Lfull.push_back(n->inskip);
    *out = n;
    return LINNA_OK;
} LINNA_CATCH_INT
int linna_net_destroy(linna_net_t* net) try {
    if (net) {
        net->packed.release(); net->packed_dx[0].release(); net->packed_dx[1].release(); net->packed_loss.release();
        net->packed_tb.release();
    }
    delete net;
    return LINNA_OK;
} LINNA_CATCH_INT

// Device-side weight copies of the one-launch paths (net_stream.hip): decided once per network, allocated by
// linna_net_prepare -- or on first use, when the caller did not prepare and the stream is not capturing.
static void net_ensure_fwd(linna_net* n, bool may_alloc) {
    const int nl = (int)n->L.size();
    if (n->stream_fwd < 0) {
        const char* e = getenv("LINNA_FWD_STREAM");
        n->stream_fwd = !n->has_inskip && !(e && e[0] == '0') && net_stream_eligible(n->L.data(), nl, n->in_size) ? 1 : 0;
    }
    if (n->stream_fwd == 1 && !n->packed.ready() && may_alloc) {
        if (n->packed.alloc(net_stream_packed_floats(n->L.data(), nl, n->in_size)) != LINNA_OK) n->stream_fwd = 0;
    }
}
static void net_ensure_dx(linna_net* n, int wi, bool may_alloc) {
    const int nl = (int)n->L.size();
    if (n->stream_bwd[wi] < 0) {
        const char* e = getenv("LINNA_BWD_STREAM");
        n->stream_bwd[wi] = !n->has_inskip && (nl >= 2 || wi) && !(e && e[0] == '0') &&
                            net_stream_dx_eligible(n->L.data(), nl, n->in_size, wi) ? 1 : 0;
    }
    StreamCopy& sc = n->packed_dx[wi];
    if (n->stream_bwd[wi] == 1 && !sc.ready() && may_alloc) {
        if (sc.alloc(net_stream_dx_packed_floats(n->L.data(), nl, n->in_size, wi)) != LINNA_OK) n->stream_bwd[wi] = 0;
    }
}
int linna_net_prepare(linna_net_t* n, int backward, int input_grad) try {
    if (!n) { set_error("net_prepare: null network"); return LINNA_ERR_INVALID; }
    net_ensure_fwd(n, true);
    if (backward) net_ensure_dx(n, input_grad ? 1 : 0, true);
    if (n->ctx && backward) {                                // the auxiliary stream and its events, for the same reason
        linna_ctx* ctx = n->ctx;
        if (!ctx->aux) TRY(check_hip(hipStreamCreateWithFlags(&ctx->aux, hipStreamNonBlocking), "hipStreamCreate"));
        while ((int)ctx->events.size() < 2 * (int)n->L.size() + 4) {
            hipEvent_t e;
            TRY(check_hip(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate"));
            ctx->events.push_back(e);
        }
    }
    return LINNA_OK;
} LINNA_CATCH_INT

size_t linna_net_fwd_ws_bytes(const linna_net_t* n, int B) try { return (fwd_layout(n, B).total + 16) * sizeof(float); } LINNA_CATCH_SIZE
// backward scratch: one buffer per op for the gradient wrt that op's input (no reuse: the
// parameter-gradient GEMMs of an op may still be reading it on the auxiliary stream while the dX
// chain moves on) + one dT buffer per residual block
static size_t bwd_floats(const linna_net* n, int B) {
    size_t f = 0;
    for (size_t i = 0; i < n->L.size(); ++i) {
        f += (size_t)B * ld4(n->L[i].K);
        if (n->L[i].op == LINNA_OP_RESBLOCK) f += (size_t)B * ld4(n->L[i].C);
    }
    return f;
}
size_t linna_net_bwd_ws_bytes(const linna_net_t* n, int B) try { return (bwd_floats(n, B) + 16) * sizeof(float); } LINNA_CATCH_SIZE

int linna_net_forward(linna_net_t* n, const float* X, int ldx, int B, void* ws, float* OUT, int ldo,
                      const linna_colmap_t* om, void* stream) try {
    if (!n || !X || !OUT || B < 1) { set_error("net_forward: bad arguments"); return LINNA_ERR_INVALID; }
    const FwdLayout f = fwd_layout(n, B);
    float* w = static_cast<float*>(ws);
    const int nl = (int)n->L.size();
    if (nl > 1 && !w) { set_error("net_forward: workspace required"); return LINNA_ERR_INVALID; }
    if ((!om || !om->cexp) && !n->has_inskip) {
        // ONE launch (net_stream.hip, STORE): at batch 500 the ten layer GEMMs are 10-30 us of latency each.  The
        // fragment-order weight copy is re-laid whenever the weights moved (every optimiser step: ~10 us).
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(S(stream), &cap);
        net_ensure_fwd(n, cap == hipStreamCaptureStatusNone);
        if (n->stream_fwd == 1 && n->packed.ready()) {
            const int rows = net_stream_rows(B);
            const float* packed = nullptr;
            TRY(stream_copy_refresh(n->packed, n, rows, stream, &packed));
            std::vector<float*> y(nl), t(nl);
            std::vector<int> ldy(nl), ldt(nl);
            for (int i = 0; i < nl; ++i) {
                const bool last = i == nl - 1;
                y[i] = last ? OUT : w + f.y_off[i]; ldy[i] = last ? ldo : ld4(n->L[i].N);
                t[i] = n->L[i].op == LINNA_OP_RESBLOCK ? w + f.t_off[i] : nullptr; ldt[i] = ld4(n->L[i].C);
            }
            return launch_net_stream_store(n->L.data(), nl, n->in_size, packed, X, ldx, B, y.data(), ldy.data(), t.data(),
                                           ldt.data(), om ? om->cscale : nullptr, om ? om->cshift : nullptr, rows, S(stream));
        }
    }
    const float* hin = X; int ldh = ldx;
    for (int i = 0; i < nl; ++i) {
        const linna_layer_t& l = n->L[i];
        const bool last = (i == nl - 1);
        float* Y = last ? OUT : w + f.y_off[i];
        const int ldy = last ? ldo : ld4(l.N);
        if (l.op == LINNA_OP_LINEAR) {
            GemmArgs a = gemm_zero();
            a.M = B; a.N = l.N; a.C = Y; a.ldc = ldy; a.relu = l.relu;
            if (last && n->has_inskip) {
                // out = (hin W^T + b) + alpha * (X0 Wl^T + bl): pair 0 carries the scaled input skip
                const linna_layer_t& s = n->inskip;
                a.npairs = 2;
                set_pair(a, 0, X, ldx, LAY_K, s.W, ld4(s.K), LAY_K, s.K);
                a.bias0 = s.b; a.alpha0 = s.alpha;
                set_pair(a, 1, hin, ldh, LAY_K, l.W, ld4(l.K), LAY_K, l.K);
                a.bias1 = l.b;
            } else {
                set_pair(a, 0, hin, ldh, LAY_K, l.W, ld4(l.K), LAY_K, l.K);
                a.bias0 = l.b;
            }
            if (last && om) {
                a.cscale = om->cscale; a.cshift = om->cshift; a.cexp = om->cexp; a.cpost = om->cpost; a.cshift2 = om->cshift2;
            }
            TRY(gemm_launch(a, S(stream)));
        } else {
            float* T = w + f.t_off[i];
            TRY(linna_resblock_fwd(nullptr, hin, ldh, l.W1, l.b1, l.W2, l.b2, l.Ws, T, ld4(l.C), Y, ldy, B, l.K, l.C, l.N, stream));
        }
        hin = Y; ldh = ldy;
    }
    return LINNA_OK;
} LINNA_CATCH_INT

static void net_ensure_loss(linna_net* n, const NsDense& dn) {
    const int nl = (int)n->L.size();
    const char* e = getenv("LINNA_LOSS_STREAM");
    const bool ok = !n->has_inskip && !(e && e[0] == '0') && net_stream_dense_eligible(n->L.data(), nl, n->in_size, dn);
    n->packed_loss.release();
    n->stream_loss = 0; n->loss_dn = dn;
    if (ok && n->packed_loss.alloc(net_stream_dense_packed_floats(n->L.data(), nl, n->in_size, dn)) == LINNA_OK) n->stream_loss = 1;
    // the same loss behind the one-launch training step (forward + loss + dX chain in one weight stream)
    const char* m = getenv("LINNA_TRAIN_MERGED");
    const char* b = getenv("LINNA_BWD_STREAM");
    const bool okm = ok && n->stream_loss == 1 && !(m && m[0] == '0') && !(b && b[0] == '0') && nl >= 2 &&
                     net_stream_tb_eligible(n->L.data(), nl, n->in_size, dn);
    n->packed_tb.release();
    n->stream_tb = 0; n->as_merged = -1; n->as_state = -1;
    if (okm && n->packed_tb.alloc(net_stream_tb_packed_floats(n->L.data(), nl, n->in_size, dn)) == LINNA_OK) n->stream_tb = 1;
}
// The one-launch training step serves this batch size (the 4-row engine only) -- decided the same way by every entry
// that touches the training streams of a step (train_step, train_step_update, adamw_step)
static bool net_tb_usable(const linna_net* n, int B) {
    return n->stream_tb == 1 && n->packed_tb.ready() && net_stream_rows(B) == 4;     // (batches of up to 1024 rows)
}
int linna_loss_targets(linna_ctx_t*, const linna_loss_desc_t* d, const float* Y, int ldy, int nrows, float* YN, int ldyn, void* stream) try {
    if (!d || !Y || !YN || nrows < 1) { set_error("loss_targets: bad arguments"); return LINNA_ERR_INVALID; }
    CHECK_STRUCT(d, linna_loss_desc_t, "loss_targets");
    if (ldyn < d->nout) { set_error("loss_targets: ldyn %d < nout %d", ldyn, d->nout); return LINNA_ERR_INVALID; }
    return launch_loss_targets(Y, ldy, nrows, *d, YN, ldyn, S(stream));
} LINNA_CATCH_INT
int linna_net_prepare_loss(linna_net_t* n, const linna_loss_desc_t* d) try {
    if (!n || !d) { set_error("net_prepare_loss: null argument"); return LINNA_ERR_INVALID; }
    CHECK_STRUCT(d, linna_loss_desc_t, "net_prepare_loss");
    const NsDense dn{d->Cinv, d->ldc, nullptr, nullptr};
    if (n->stream_loss < 0 || n->loss_dn.S != dn.S || n->loss_dn.lds != dn.lds) net_ensure_loss(n, dn);
    return LINNA_OK;
} LINNA_CATCH_INT
// Forward pass of a training step AND its loss in ONE launch (net_stream.hip, STORE == 3): the batch rows are gathered
// from the resident set and X-transformed in the kernel's prologue, every activation the backward needs is stored, the
// network's normalised-space inverse covariance is the program's last segment and the finish writes the per-row loss and
// d loss / d pred.  Replaces linna_gather_xform + linna_net_forward + linna_chi2_ratio_loss_fwd_bwd (seven launches for
// nout > 64) when the network + loss fit the whole-network kernel; LINNA_ERR_UNSUPPORTED otherwise (the caller then
// runs that sequence).  The batch mean is a second, tiny launch (fixed summation order).
struct NetUpdate { float* params; float* m; float* v; size_t n; float* hyper; float b1, b2, eps; };
static int net_backward_impl(linna_net_t* n, const float* X, int ldx, int B, void* fwd_ws, void* bwd_ws, const float* dOUT,
                             int lddo, float* dX, int lddx, int pg, void* stream, const NsPost* post, const NetUpdate* upd = nullptr,
                             bool dx_done = false, const GemmPost* gpost = nullptr);
static int net_forward_loss_impl(linna_net_t* n, const linna_loss_desc_t* d, const float* X, int ldx, const int* ROWS, int B,
                                 const int* lg, const float* xmean, const float* xstd, float* XB, int ldxb, void* ws, float* PRED,
                                 int ldp, const float* YN, int ldyn, const float* den, float inv_batch, float* loss_rows,
                                 float* loss_mean, float* dPRED, int lddp, float* hyper, int* step_dev, float b1, float b2,
                                 void* stream, bool defer_post);
int linna_net_forward_loss(linna_net_t* n, const linna_loss_desc_t* d, const float* X, int ldx, const int* ROWS, int B,
                           const int* lg, const float* xmean, const float* xstd, float* XB, int ldxb, void* ws, float* PRED,
                           int ldp, const float* YN, int ldyn, const float* den, float inv_batch, float* loss_rows,
                           float* loss_mean, float* dPRED, int lddp, float* hyper, int* step_dev, float b1, float b2,
                           void* stream) try {
    if (!d) { set_error("net_forward_loss: null loss descriptor"); return LINNA_ERR_INVALID; }
    CHECK_STRUCT(d, linna_loss_desc_t, "net_forward_loss");
    return net_forward_loss_impl(n, d, X, ldx, ROWS, B, lg, xmean, xstd, XB, ldxb, ws, PRED, ldp, YN, ldyn, den, inv_batch, loss_rows,
                                 loss_mean, dPRED, lddp, hyper, step_dev, b1, b2, stream, false);
} LINNA_CATCH_INT
// One optimiser step up to the gradients in ONE call: linna_net_forward_loss followed by linna_net_backward(param_grads = 1)
// on the rows it gathered, with the step's two single-thread jobs (batch mean of the loss, AdamW step counter and bias
// corrections) riding in the backward's dX-chain launch as one extra workgroup instead of a launch of their own between
// the two whole-network launches.  LINNA_ERR_UNSUPPORTED exactly when linna_net_forward_loss is.
// Forward + loss + dX chain of a training step in ONE launch (net_stream.hip TRB); the caller has checked net_tb_usable.
static int net_train_merged_impl(linna_net_t* n, const linna_loss_desc_t* d, const float* X, int ldx, const int* ROWS, int B,
                                 const int* lg, const float* xmean, const float* xstd, float* XB, int ldxb, void* fwd_ws, float* PRED,
                                 int ldp, const float* YN, int ldyn, const float* den, float inv_batch, float* loss_rows,
                                 float* dPRED, int lddp, void* bwd_ws, float* hyper, int* step_dev, float b1, float b2, void* stream) {
    if (!n || !d || !X || !xmean || !xstd || !XB || !PRED || !YN || !den || !loss_rows || !dPRED || !fwd_ws || !bwd_ws || B < 1) {
        set_error("net_train_step: bad arguments"); return LINNA_ERR_INVALID;
    }
    if (d->nout != n->out_size) { set_error("net_train_step: loss for %d outputs, network has %d", d->nout, n->out_size); return LINNA_ERR_INVALID; }
    const int nl = (int)n->L.size();
    const FwdLayout f = fwd_layout(n, B);
    float* w = static_cast<float*>(fwd_ws);
    float* bw = static_cast<float*>(bwd_ws);
    const int rows = net_stream_rows(B);
    const float* packed = nullptr;
    TRY(stream_copy_refresh(n->packed_tb, n, rows, stream, &packed, 4, &n->loss_dn));
    std::vector<float*> y(nl), t(nl), dprev(nl, nullptr), dt(nl, nullptr);
    std::vector<const float*> hinp(nl, nullptr);
    std::vector<int> ldy_(nl), ldt(nl), ldpv(nl, 0), ldhv(nl, 0), lddt(nl, 0);
    for (int i = 0; i < nl; ++i) {
        const bool last = i == nl - 1;
        y[i] = last ? PRED : w + f.y_off[i]; ldy_[i] = last ? ldp : ld4(n->L[i].N);
        t[i] = n->L[i].op == LINNA_OP_RESBLOCK ? w + f.t_off[i] : nullptr; ldt[i] = ld4(n->L[i].C);
    }
    float* cur = bw;
    for (int i = nl - 1; i >= 1; --i) {                            // net_backward_impl's workspace walk (no input gradient)
        const linna_layer_t& l = n->L[i];
        dprev[i] = cur; ldpv[i] = ld4(l.K);
        cur += (size_t)B * ld4(l.K);
        const bool hin_relu = n->L[i - 1].op == LINNA_OP_RESBLOCK || n->L[i - 1].relu;
        hinp[i] = hin_relu ? w + f.y_off[i - 1] : nullptr; ldhv[i] = ld4(n->L[i - 1].N);
        if (l.op == LINNA_OP_RESBLOCK) { dt[i] = cur; lddt[i] = ld4(l.C); cur += (size_t)B * ld4(l.C); }
    }
    const NsTrainLoss L{YN, ldyn, den, inv_batch, loss_rows, dPRED, lddp};
    const bool prep = hyper && step_dev;
    const NsPost post{nullptr, 0, 0.f, nullptr, prep ? step_dev : nullptr, prep ? hyper : nullptr, b1, b2};
    return launch_net_stream_train_bwd(n->L.data(), nl, n->in_size, packed, X, ldx, ROWS, B, lg, xmean, xstd, XB, ldxb, y.data(),
                                       ldy_.data(), t.data(), ldt.data(), L, n->loss_dn, dprev.data(), ldpv.data(), hinp.data(),
                                       ldhv.data(), dt.data(), lddt.data(), rows, S(stream), prep ? &post : nullptr);
}
// (the loss descriptor's stream state, as net_forward_loss_impl establishes it)
static int net_train_ensure_loss(linna_net_t* n, const linna_loss_desc_t* d, void* stream) {
    if (!n || !d) { set_error("net_train_step: null argument"); return LINNA_ERR_INVALID; }
    const NsDense dn{d->Cinv, d->ldc, nullptr, nullptr};
    if (n->stream_loss < 0 || n->loss_dn.S != dn.S || n->loss_dn.lds != dn.lds) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(S(stream), &cap);
        if (cap != hipStreamCaptureStatusNone) {
            set_error("net_train_step: first use inside a stream capture (call linna_net_prepare_loss before)"); return LINNA_ERR_UNSUPPORTED;
        }
        net_ensure_loss(n, dn);
    }
    return LINNA_OK;
}
int linna_net_train_step(linna_net_t* n, const linna_loss_desc_t* d, const float* X, int ldx, const int* ROWS, int B,
                         const int* lg, const float* xmean, const float* xstd, float* XB, int ldxb, void* fwd_ws, float* PRED,
                         int ldp, const float* YN, int ldyn, const float* den, float inv_batch, float* loss_rows,
                         float* loss_mean, float* dPRED, int lddp, void* bwd_ws, float* hyper, int* step_dev, float b1, float b2,
                         void* stream) try {
    if (!bwd_ws) { set_error("net_train_step: backward workspace required"); return LINNA_ERR_INVALID; }
    if (!d) { set_error("net_train_step: null loss descriptor"); return LINNA_ERR_INVALID; }
    CHECK_STRUCT(d, linna_loss_desc_t, "net_train_step");
    TRY(net_train_ensure_loss(n, d, stream));
    if (net_tb_usable(n, B)) {
        // two launches: forward + loss + dX chain, then every parameter gradient (the batch mean of the loss riding in it)
        TRY(net_train_merged_impl(n, d, X, ldx, ROWS, B, lg, xmean, xstd, XB, ldxb, fwd_ws, PRED, ldp, YN, ldyn, den, inv_batch,
                                  loss_rows, dPRED, lddp, bwd_ws, hyper, step_dev, b1, b2, stream));
        const GemmPost gp{loss_rows, loss_mean ? B : 0, inv_batch, loss_mean};
        return net_backward_impl(n, XB, ldxb, B, fwd_ws, bwd_ws, dPRED, lddp, nullptr, 0, 1, stream, nullptr, nullptr, true, &gp);
    }
    TRY(net_forward_loss_impl(n, d, X, ldx, ROWS, B, lg, xmean, xstd, XB, ldxb, fwd_ws, PRED, ldp, YN, ldyn, den, inv_batch,
                              loss_rows, loss_mean, dPRED, lddp, hyper, step_dev, b1, b2, stream, true));
    const bool prep = hyper && step_dev;
    const NsPost post{loss_rows, (loss_mean || prep) ? B : 0, inv_batch, loss_mean, prep ? step_dev : nullptr, prep ? hyper : nullptr, b1, b2};
    return net_backward_impl(n, XB, ldxb, B, fwd_ws, bwd_ws, dPRED, lddp, nullptr, 0, 1, stream, post.n ? &post : nullptr);
} LINNA_CATCH_INT
static int net_forward_loss_impl(linna_net_t* n, const linna_loss_desc_t* d, const float* X, int ldx, const int* ROWS, int B,
                                 const int* lg, const float* xmean, const float* xstd, float* XB, int ldxb, void* ws, float* PRED,
                                 int ldp, const float* YN, int ldyn, const float* den, float inv_batch, float* loss_rows,
                                 float* loss_mean, float* dPRED, int lddp, float* hyper, int* step_dev, float b1, float b2,
                                 void* stream, bool defer_post) {
    if (!n || !d || !X || !xmean || !xstd || !XB || !PRED || !YN || !den || !loss_rows || !dPRED || B < 1) {
        set_error("net_forward_loss: bad arguments"); return LINNA_ERR_INVALID;
    }
    if (d->nout != n->out_size) { set_error("net_forward_loss: loss for %d outputs, network has %d", d->nout, n->out_size); return LINNA_ERR_INVALID; }
    const int nl = (int)n->L.size();
    const NsDense dn{d->Cinv, d->ldc, nullptr, nullptr};
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(S(stream), &cap);
    if (n->stream_loss < 0 || n->loss_dn.S != dn.S || n->loss_dn.lds != dn.lds) {
        if (cap != hipStreamCaptureStatusNone) {
            set_error("net_forward_loss: first use inside a stream capture (call linna_net_prepare_loss before)"); return LINNA_ERR_UNSUPPORTED;
        }
        net_ensure_loss(n, dn);
    }
    if (n->stream_loss != 1) { set_error("net_forward_loss: this network / loss does not run the whole-network kernel"); return LINNA_ERR_UNSUPPORTED; }
    const FwdLayout f = fwd_layout(n, B);
    float* w = static_cast<float*>(ws);
    if (nl > 1 && !w) { set_error("net_forward_loss: workspace required"); return LINNA_ERR_INVALID; }
    const int rows = net_stream_rows(B);
    const float* packed = nullptr;
    TRY(stream_copy_refresh(n->packed_loss, n, rows, stream, &packed, 0, &n->loss_dn));
    std::vector<float*> y(nl), t(nl);
    std::vector<int> ldy_(nl), ldt(nl);
    for (int i = 0; i < nl; ++i) {
        const bool last = i == nl - 1;
        y[i] = last ? PRED : w + f.y_off[i]; ldy_[i] = last ? ldp : ld4(n->L[i].N);
        t[i] = n->L[i].op == LINNA_OP_RESBLOCK ? w + f.t_off[i] : nullptr; ldt[i] = ld4(n->L[i].C);
    }
    const NsTrainLoss L{YN, ldyn, den, inv_batch, loss_rows, dPRED, lddp};
    TRY(launch_net_stream_train(n->L.data(), nl, n->in_size, packed, X, ldx, ROWS, B, lg, xmean, xstd, XB, ldxb, y.data(),
                                ldy_.data(), t.data(), ldt.data(), L, n->loss_dn, rows, S(stream)));
    if (defer_post) return LINNA_OK;                 // linna_net_train_step: they ride in the backward's dX launch
    // the batch mean -- and, when the caller hands in its AdamW state, the step counter and bias corrections of the
    // update that will follow this step's backward (linna_adamw_step(prepared = 1)): two single-thread jobs, one launch
    if (loss_mean && hyper && step_dev) return launch_sum_scale_prepare(loss_rows, B, inv_batch, loss_mean, step_dev, hyper, b1, b2, S(stream));
    if (hyper && step_dev) TRY(launch_adamw_prepare(hyper, step_dev, b1, b2, S(stream)));
    if (loss_mean) TRY(launch_sum_scale(loss_rows, B, inv_batch, loss_mean, S(stream)));
    return LINNA_OK;
}

int linna_net_train_launches(const linna_net_t* n, int B) try {
    if (!n || B < 1) { set_error("net_train_launches: bad arguments"); return LINNA_ERR_INVALID; }
    if (n->stream_loss != 1) return 0;
    if (net_tb_usable(n, B)) return 2;
    return n->stream_bwd[0] == 1 ? 3 : 0;
} LINNA_CATCH_INT
int linna_net_stream_state(const linna_net_t* n, int* fwd, int* dx, int* dx_input) try {
    if (!n) { set_error("net_stream_state: null network"); return LINNA_ERR_INVALID; }
    if (fwd) *fwd = n->stream_fwd;
    if (dx) *dx = n->stream_bwd[0];
    if (dx_input) *dx_input = n->stream_bwd[1];
    return LINNA_OK;
} LINNA_CATCH_INT

int linna_net_backward(linna_net_t* n, const float* X, int ldx, int B, void* fwd_ws, void* bwd_ws, const float* dOUT,
                       int lddo, float* dX, int lddx, int pg, void* stream) try {
    return net_backward_impl(n, X, ldx, B, fwd_ws, bwd_ws, dOUT, lddo, dX, lddx, pg, stream, nullptr);
} LINNA_CATCH_INT
// `post`: the loss mean / AdamW step constants of this step (linna_net_train_step): they ride in the one-launch dX chain
// as an extra workgroup, or run as the launch of their own they otherwise are, in front of the GEMM chain
// `upd`: the optimiser rides in the grouped parameter-gradient launch (linna_net_train_step_update; the caller has checked
// net_update_supported: every parameter gradient of the step goes into that launch)
// `dx_done`: the dX chain already ran (inside the one-launch training step): only the parameter gradients are left;
// `gpost`: the batch mean of the loss rows rides in the grouped parameter-gradient launch as one extra workgroup
static int net_backward_impl(linna_net_t* n, const float* X, int ldx, int B, void* fwd_ws, void* bwd_ws, const float* dOUT,
                             int lddo, float* dX, int lddx, int pg, void* stream, const NsPost* post, const NetUpdate* upd,
                             bool dx_done, const GemmPost* gpost) {
    if (!n || !X || !dOUT || !bwd_ws || B < 1) { set_error("net_backward: bad arguments"); return LINNA_ERR_INVALID; }
    const FwdLayout f = fwd_layout(n, B);
    const float* w = static_cast<const float*>(fwd_ws);
    float* bw = static_cast<float*>(bwd_ws);
    const int nl = (int)n->L.size();
    hipStream_t st = S(stream);

    // ---- auxiliary stream for the parameter gradients (off the dX critical path)
    linna_ctx* ctx = n->ctx;
    bool overlap = false;
    if (pg && ctx) {
        if (ctx->overlap < 0) {
            const char* e = getenv("LINNA_BWD_STREAMS");
            ctx->overlap = (e && e[0] == '0') ? 0 : 1;
        }
        if (ctx->overlap == 1) {
            if (!ctx->aux) TRY(check_hip(hipStreamCreateWithFlags(&ctx->aux, hipStreamNonBlocking), "hipStreamCreate"));
            while ((int)ctx->events.size() < 2 * nl + 4) {
                hipEvent_t e;
                TRY(check_hip(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate"));
                ctx->events.push_back(e);
            }
            overlap = true;
        }
    }
    int next_event = 0;
    void* aux = overlap ? (void*)ctx->aux : stream;
    // Parameter gradients: the bias column sums go to the auxiliary stream as they become possible; the dW GEMMs
    // (1-128 tiles each, 251 together for ChtoModelv2(33,33)) are collected and launched as ONE grid after the
    // dX chain -- 13 launches of ~18 us each on the auxiliary stream were the critical path of the step.
    if (pg && ctx && ctx->group < 0) {
        const char* e = getenv("LINNA_BWD_GROUP");
        ctx->group = (e && e[0] == '0') ? 0 : 1;
    }
    const bool grouping = pg && ctx && ctx->group == 1;
    GemmGroupArgs grp;                  // the grouped parameter-gradient launch: descriptors by value, filled as we go
    grp.nprob = 0;
    GemmGroupArgsS grpu;                // ... and its form with the optimiser in the epilogue
    GemmUpdate gu;
    grpu.nprob = 0;
    long long pdiff = 0;
    if (upd) {
        ::memset(static_cast<void*>(&gu), 0, sizeof(gu));
        const linna_layer_t& l0 = n->L[0];
        const float* w0 = l0.op == LINNA_OP_RESBLOCK ? l0.W1 : l0.W;
        const float* g0 = l0.op == LINNA_OP_RESBLOCK ? l0.gW1 : l0.gW;
        pdiff = w0 - g0;
        gu.pdiff = pdiff; gu.mdiff = (upd->m - upd->params) + pdiff; gu.vdiff = (upd->v - upd->params) + pdiff;
        gu.hyper = upd->hyper; gu.beta1 = upd->b1; gu.beta2 = upd->b2; gu.eps = upd->eps; gu.small = n->as_args.small;
    }
    auto as_range_of = [&](const float* param, int kind) -> const AsRange* {     // the flat-buffer tensor that starts at `param`
        const long long off = param - upd->params;
        for (int i = 0; i < n->as_args.nr; ++i)
            if (n->as_args.r[i].kind == kind && (long long)n->as_args.r[i].off4 * 4 == off) return &n->as_args.r[i];
        return nullptr;
    };
    int grp_blocks = 0;
    bool aux_used = false;
    auto fork = [&]() -> int {       // work enqueued on aux after this sees everything enqueued on st so far
        if (!overlap) return LINNA_OK;
        aux_used = true;
        hipEvent_t e = ctx->events[next_event++];
        TRY(check_hip(hipEventRecord(e, st), "hipEventRecord"));
        return check_hip(hipStreamWaitEvent(ctx->aux, e, 0), "hipStreamWaitEvent");
    };
    auto param_grads = [&](const float* dY, int lddy, const float* Xin, int ldxin, float* dW, int lddw, float* db, int K,
                           int N, float scale) -> int {
        GemmArgs a = gemm_zero();    // dW[n][k] = scale * sum_b dY[b][n] X[b][k]
        set_pair(a, 0, dY, lddy, LAY_MN, Xin, ldxin, LAY_MN, B);
        a.M = N; a.N = K; a.C = dW; a.ldc = lddw; a.alpha0 = scale;
        if (upd) {
            if (!(grouping && grpu.nprob < GEMM_UPD_MAX && gemm_group_ok(a))) { set_error("net_backward: update outside the grouped launch"); return LINNA_ERR_INVALID; }
            const AsRange* rw = as_range_of(dW + pdiff, 0);
            const AsRange* rb = db ? as_range_of(db + pdiff, 1) : nullptr;
            if (!rw || (db && !rb)) { set_error("net_backward: update of a tensor outside the flat parameter buffer"); return LINNA_ERR_INVALID; }
            const int i = grpu.nprob++;
            grpu.p[i] = GemmGroupProb{dY, Xin, dW, db, lddy, ldxin, lddw, B, N, K, scale, grp_blocks};
            gu.pl[i][0] = n->as_args.w[rw->idx].pl[0]; gu.pl[i][1] = n->as_args.w[rw->idx].pl[1];
            if (rb) gu.bias[i] = n->as_args.b[rb->idx];
            grp_blocks += gemm_group_blocks(a);
            return LINNA_OK;
        }
        if (grouping && grp.nprob < GEMM_GROUP_MAX && gemm_group_ok(a)) {
            // one tile grid for every dW of the step; the bias gradient (column sums of dY) rides in the same tiles
            grp.p[grp.nprob++] = GemmGroupProb{dY, Xin, dW, db, lddy, ldxin, lddw, B, N, K, scale, grp_blocks};
            grp_blocks += gemm_group_blocks(a);
            return LINNA_OK;
        }
        TRY(fork()); TRY(gemm_launch(a, S(aux)));
        if (db) { TRY(fork()); TRY(launch_colsum(dY, lddy, B, N, scale, db, S(aux))); }
        return LINNA_OK;
    };

    if (n->has_inskip && pg) {
        const linna_layer_t& s = n->inskip;
        TRY(param_grads(dOUT, lddo, X, ldx, s.gW, ld4(s.K), s.gb, s.K, s.N, s.alpha));
    }
    // The dX chain -- one GEMM per op, each waiting for the one before (140 us of 300 at batch 500) -- as ONE launch of
    // the whole-network kernel over the transposed weights (net_stream.hip, STORE == 2), when the network has such a
    // program.  The loop below then only collects the parameter gradients.
    bool fused_dx = dx_done;
    const int wi = dX ? 1 : 0;
    if (!dx_done && !n->has_inskip && (nl >= 2 || dX)) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(st, &cap);
        net_ensure_dx(n, wi, cap == hipStreamCaptureStatusNone);
        StreamCopy& sc = n->packed_dx[wi];
        if (n->stream_bwd[wi] == 1 && sc.ready()) {
            const int rows = net_stream_rows(B);
            const float* packed = nullptr;
            TRY(stream_copy_refresh(sc, n, rows, stream, &packed, 1 + wi));
            std::vector<float*> dprev(nl, nullptr), dt(nl, nullptr);
            std::vector<const float*> hinp(nl, nullptr), tp(nl, nullptr);
            std::vector<int> ldpv(nl, 0), ldhv(nl, 0), lddt(nl, 0), ldtv(nl, 0);
            float* cur = bw;
            for (int i = nl - 1; i >= (wi ? 0 : 1); --i) {                 // the same workspace walk as the loop below
                const linna_layer_t& l = n->L[i];
                dprev[i] = (i == 0) ? dX : cur; ldpv[i] = (i == 0) ? lddx : ld4(l.K);
                if (i > 0) cur += (size_t)B * ld4(l.K);
                const bool hin_relu = (i > 0) && (n->L[i - 1].op == LINNA_OP_RESBLOCK || n->L[i - 1].relu);
                hinp[i] = hin_relu ? w + f.y_off[i - 1] : nullptr; ldhv[i] = (i == 0) ? ldx : ld4(n->L[i - 1].N);
                if (l.op == LINNA_OP_RESBLOCK) {
                    dt[i] = cur; lddt[i] = ld4(l.C); cur += (size_t)B * ld4(l.C);
                    tp[i] = w + f.t_off[i]; ldtv[i] = ld4(l.C);
                }
            }
            TRY(launch_net_stream_dx(n->L.data(), nl, n->in_size, packed, dOUT, lddo, B, dprev.data(), ldpv.data(), hinp.data(),
                                     ldhv.data(), dt.data(), lddt.data(), tp.data(), ldtv.data(), wi, rows, st, post));
            fused_dx = true;
        }
    }
    if (post && !fused_dx) {
        if (post->out && post->step) TRY(launch_sum_scale_prepare(post->rows, post->n, post->scale, post->out, post->step, post->hyper, post->b1, post->b2, st));
        else if (post->out) TRY(launch_sum_scale(post->rows, post->n, post->scale, post->out, st));
        else if (post->step) TRY(launch_adamw_prepare(post->hyper, post->step, post->b1, post->b2, st));
    }
    const float* dcur = dOUT; int ldd = lddo;
    float* cursor = bw;
    for (int i = nl - 1; i >= 0; --i) {
        const linna_layer_t& l = n->L[i];
        const float* hin = (i == 0) ? X : w + f.y_off[i - 1];
        const int ldh = (i == 0) ? ldx : ld4(n->L[i - 1].N);
        const bool need_dx = (i > 0) || (dX != nullptr);
        float* dprev = (i == 0) ? dX : cursor;
        const int ldp = (i == 0) ? lddx : ld4(l.K);
        if (i > 0) cursor += (size_t)B * ld4(l.K);
        // hin went through a ReLU iff the producing op is a resblock or a linear with relu
        const bool hin_relu = (i > 0) && (n->L[i - 1].op == LINNA_OP_RESBLOCK || n->L[i - 1].relu);
        const float* mask = hin_relu ? hin : nullptr;
        if (l.op == LINNA_OP_LINEAR) {
            if (pg) {                // dW, db need only dcur (already produced on st) and hin
                TRY(param_grads(dcur, ldd, hin, ldh, l.gW, ld4(l.K), l.gb, l.K, l.N, 1.f));
            }
            if (need_dx && !fused_dx) {
                GemmArgs a = gemm_zero();
                a.M = B; a.N = l.K; a.C = dprev; a.ldc = ldp; a.mask = mask; a.ldmask = ldh;
                if (i == 0 && n->has_inskip) {
                    const linna_layer_t& s = n->inskip;
                    a.npairs = 2;
                    set_pair(a, 0, dOUT, lddo, LAY_K, s.W, ld4(s.K), LAY_MN, s.N);
                    a.alpha0 = s.alpha;
                    set_pair(a, 1, dcur, ldd, LAY_K, l.W, ld4(l.K), LAY_MN, l.N);
                } else {
                    set_pair(a, 0, dcur, ldd, LAY_K, l.W, ld4(l.K), LAY_MN, l.N);
                }
                TRY(gemm_launch(a, st));
            }
        } else {
            const float* T = w + f.t_off[i];
            const int ldt = ld4(l.C);
            float* dT = cursor;
            cursor += (size_t)B * ldt;
            if (!fused_dx) {   // dT = 0.1 * (dcur W2) * (T > 0)
                GemmArgs a = gemm_zero();
                set_pair(a, 0, dcur, ldd, LAY_K, l.W2, ld4(l.C), LAY_MN, l.N);
                a.M = B; a.N = l.C; a.C = dT; a.ldc = ldt; a.alpha0 = 0.1f; a.mask = T; a.ldmask = ldt;
                TRY(gemm_launch(a, st));
            }
            if (pg) {
                TRY(param_grads(dcur, ldd, T, ldt, l.gW2, ld4(l.C), l.gb2, l.C, l.N, 0.1f));   // (after dT)
                TRY(param_grads(dT, ldt, hin, ldh, l.gW1, ld4(l.K), l.gb1, l.K, l.C, 1.f));
                if (l.Ws) TRY(param_grads(dcur, ldd, hin, ldh, l.gWs, ld4(l.K), nullptr, l.K, l.N, 1.f));
            }
            if (need_dx && !fused_dx) {   // dprev = (dT W1 + dcur Ws [+ dcur]) * (hin > 0)
                GemmArgs a = gemm_zero();
                set_pair(a, 0, dT, ldt, LAY_K, l.W1, ld4(l.K), LAY_MN, l.C);
                a.M = B; a.N = l.K; a.C = dprev; a.ldc = ldp; a.mask = mask; a.ldmask = ldh;
                if (l.Ws) { a.npairs = 2; set_pair(a, 1, dcur, ldd, LAY_K, l.Ws, ld4(l.K), LAY_MN, l.N); }
                else { a.R = dcur; a.ldr = ldd; }
                TRY(gemm_launch(a, st));
            }
        }
        dcur = dprev; ldd = ldp;
    }
    if (grp.nprob) { TRY(gemm_launch_group(grp, grp_blocks, st, gpost)); gpost = nullptr; }     // every dY is on `st` by now: one grid over all the dW tiles
    if (grpu.nprob) { TRY(gemm_launch_group_update(grpu, gu, grp_blocks, st, gpost)); gpost = nullptr; }
    if (gpost && gpost->n > 0) TRY(launch_sum_scale(gpost->rows, gpost->n, gpost->scale, gpost->out, st));   // (no grouped launch to ride in)
    if (overlap && aux_used) {       // join: the caller's stream continues only after every gradient is written
        hipEvent_t e = ctx->events[next_event++];
        TRY(check_hip(hipEventRecord(e, ctx->aux), "hipEventRecord"));
        TRY(check_hip(hipStreamWaitEvent(st, e, 0), "hipStreamWaitEvent"));
    }
    return LINNA_OK;
}

// ------------------------------------------------------------------ prior map
int linna_prior_map_fwd(linna_ctx_t*, const float* Z, int ldz, int B, int nin, const int* is_flat, const float* a1,
                        const float* a2, const int* lg, const float* xmean, const float* xstd, float* X, int ldx,
                        float* TH, int ldt, void* stream) try {
    if (ldx < nin || ldz < nin) { set_error("prior_map_fwd: leading dimension < nin"); return LINNA_ERR_INVALID; }
    return launch_prior_map_fwd(Z, ldz, B, nin, is_flat, a1, a2, lg, xmean, xstd, X, ldx, TH, ldt, S(stream));
} LINNA_CATCH_INT
int linna_prior_map_bwd(linna_ctx_t*, const float* Z, int ldz, int B, int nin, const int* is_flat, const float* a1,
                        const float* a2, const int* lg, const float* xstd, const float* dX, int lddx, float* dZ,
                        int lddz, void* stream) try {
    return launch_prior_map_bwd(Z, ldz, B, nin, is_flat, a1, a2, lg, xstd, dX, lddx, dZ, lddz, S(stream));
} LINNA_CATCH_INT

// ------------------------------------------------------------------ log-likelihood
int linna_gauss_loglike_diag(linna_ctx_t*, const float* D, int ldd, int B, int nout, const float* w, const float* Z,
                             int ldz, int nin, float T, float* out, void* stream) try {
    return launch_loglike_diag(D, ldd, B, nout, w, Z, ldz, nin, T, out, S(stream));
} LINNA_CATCH_INT
// factored: Sm holds L with S = L L^T and the row-dot is |d L|^2 (linna_logprob_desc_t::Sfac)
static int loglike_dense_impl(const float* D, int ldd, int B, int nout, const float* Sm, int lds, bool factored,
                              const float* Z, int ldz, int nin, float T, float* scratch, float* out, void* stream) {
    const int slots = gemm_slots(B, nout);
    GemmArgs a = gemm_zero();          // rows of (D S) dotted with D (or with themselves), no C store
    set_pair(a, 0, D, ldd, LAY_K, Sm, lds, LAY_MN, nout);
    a.M = B; a.N = nout; a.dotwith = D; a.lddot = ldd; a.dot_partial = scratch; a.dot_slots = slots;
    if (factored) a.flags |= LINNA_GEMM_DOT_SELF;
    TRY(gemm_launch(a, S(stream)));
    return launch_loglike_finish(scratch, slots, slots, B, Z, ldz, nin, T, out, S(stream));
}
int linna_gauss_loglike_dense(linna_ctx_t*, const float* D, int ldd, int B, int nout, const float* Sm, int lds,
                              const float* Z, int ldz, int nin, float T, float* scratch, float* out, void* stream) try {
    return loglike_dense_impl(D, ldd, B, nout, Sm, lds, false, Z, ldz, nin, T, scratch, out, stream);
} LINNA_CATCH_INT

}  // extern "C"

// ------------------------------------------------------------------ serving pipeline
struct linna_logprob {
    linna_ctx* ctx;
    linna_net* net;
    linna_logprob_desc_t d;
    StreamCopy packed;                       // fragment-order weight streams (net_stream.hip), or not allocated
    bool grad_fused = false;                 // the streams also hold the backward segments (ReLU MLPs)
    StreamCopy packed_g2;                    // forward + dX chain down to the input in one stream (any network: residual blocks, ...)
    bool grad2 = false;
    bool dense_fused = false;                // the streams end in the dense inverse covariance (output map folded in)
    int dense_tri = 2;                       // NsDense::tri, fixed when the object is created (the stream's size depends on it)
    NsDense dense() const { return NsDense{d.Sfac ? d.Sfac : d.S, d.lds, d.outmap.cscale, d.outmap.cshift, d.Sfac ? 1 : 0, dense_tri}; }
};

struct LpLayout { size_t x0, fwd, d, part, dh, bwd, dx, total; int slots; };
static LpLayout lp_layout(const linna_logprob* lp, int B, int with_grad) {
    LpLayout L;
    size_t off = 0;
    auto take = [&](size_t nfloats) { size_t o = off; off += (nfloats + 3) & ~(size_t)3; return o; };
    L.x0 = take((size_t)B * ld4(lp->d.nin));
    L.fwd = take(linna_net_fwd_ws_bytes(lp->net, B) / sizeof(float));
    L.d = take((size_t)B * ld4(lp->d.nout));
    L.slots = gemm_slots(B, lp->d.nout);
    L.part = take((size_t)B * L.slots);
    L.dh = L.bwd = L.dx = 0;
    if (with_grad) {
        L.dh = take((size_t)B * ld4(lp->d.nout));
        L.bwd = take(linna_net_bwd_ws_bytes(lp->net, B) / sizeof(float));
        L.dx = take((size_t)B * ld4(lp->d.nin));
    }
    L.total = off;
    return L;
}

static bool fused_enabled() {
    static const bool on = !(getenv("LINNA_DISABLE_FUSED") && getenv("LINNA_DISABLE_FUSED")[0] == '1');
    return on;
}
static int stream_copy_refresh(StreamCopy& sc, const linna_net* n, int rows, void* stream, const float** out, int prog,
                               const NsDense* dn, int serve) {
    const int k = rows < 16 ? 1 : 0;
    const unsigned long long epoch = g_weights_epoch.load();
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(S(stream), &cap);
    const std::vector<linna_layer_t>& LL = prog == 0 ? n->Lfull : n->L;     // (the forward programs carry the input skip)
    if (cap != hipStreamCaptureStatusNone) {
        // a captured launch carries its own re-layout, so that every replay sees the weights of that
        // moment; the copy is not valid for direct launches until they redo it
        TRY(launch_net_stream_pack(LL.data(), (int)LL.size(), n->in_size, sc.buf[k], rows, prog, dn, S(stream), serve));
        sc.epoch[k] = 0;
    } else if (sc.epoch[k] != epoch) {
        TRY(launch_net_stream_pack(LL.data(), (int)LL.size(), n->in_size, sc.buf[k], rows, prog, dn, S(stream), serve));
        sc.epoch[k] = epoch;
    }
    *out = sc.buf[k];
    return LINNA_OK;
}
// The copy the engine for `B` rows reads, re-laid if the weights moved since it was made; *rows: that engine.
static int lp_refresh_stream(linna_logprob* lp, int B, void* stream, const float** packed, int* rows) {
    *rows = net_stream_rows(B);
    const NsDense dn = lp->dense();
    return stream_copy_refresh(lp->packed, lp->net, *rows, stream, packed, 0, lp->dense_fused ? &dn : nullptr, 1);   // serve: SIDE segments on the 16-row engine
}

static int lp_forward(linna_logprob* lp, const float* Z, int ldz, int B, float* w, const LpLayout& L, float* lnP,
                      float* TH, int ldt, void* stream, bool keep_activations, const int* gate = nullptr) {
    const linna_logprob_desc_t& d = lp->d;
    const int ldx = ld4(d.nin), ldd = ld4(d.nout);
    const linna_net* n = lp->net;
    if (!keep_activations && fused_enabled() && lp->packed.ready() && (!d.outmap.cexp || (d.outmap.cpost && d.outmap.cshift2 && !lp->dense_fused))) {
        // whole-network kernel (net_stream.hip): prior map -> every layer -> output transform -> diagonal
        // log-likelihood in ONE launch, weights streamed from the fragment-order copy
        const float* packed = nullptr; int rows = 16;
        TRY(lp_refresh_stream(lp, B, stream, &packed, &rows));
        if (lp->dense_fused) {
            // dense covariance: the output map is folded into the stream's last layer and the inverse covariance is its
            // last segment -- lnP comes out of the same launch
            const NsDense dn = lp->dense();
            return launch_net_stream(n->Lfull.data(), (int)n->Lfull.size(), n->in_size, packed, Z, ldz, B, d.nin, d.is_flat, d.a1, d.a2,
                                     d.log10_flag, d.xmean, d.xstd, nullptr, nullptr, nullptr, d.temperature, lnP, nullptr, 0,
                                     TH, ldt, nullptr, nullptr, gate, rows, &dn, S(stream));
        }
        TRY(launch_net_stream(n->Lfull.data(), (int)n->Lfull.size(), n->in_size, packed, Z, ldz, B, d.nin, d.is_flat, d.a1, d.a2,
                              d.log10_flag, d.xmean, d.xstd, d.outmap.cscale, d.outmap.cshift, d.w, d.temperature,
                              d.w ? lnP : nullptr, d.w ? nullptr : w + L.d, ldd, TH, ldt, nullptr, nullptr, gate, rows, nullptr,
                              S(stream), d.outmap.cexp ? d.outmap.cpost : nullptr, d.outmap.cexp ? d.outmap.cshift2 : nullptr));
        if (d.w) return LINNA_OK;
        return loglike_dense_impl(w + L.d, ldd, B, d.nout, d.Sfac ? d.Sfac : d.S, d.lds, d.Sfac != nullptr, Z, ldz, d.nin, d.temperature,
                                         w + L.part, lnP, stream);
    }
    TRY(launch_prior_map_fwd(Z, ldz, B, d.nin, d.is_flat, d.a1, d.a2, d.log10_flag, d.xmean, d.xstd, w + L.x0, ldx, TH,
                             ldt, S(stream)));
    TRY(linna_net_forward(lp->net, w + L.x0, ldx, B, w + L.fwd, w + L.d, ldd, &d.outmap, stream));
    if (d.w) return launch_loglike_diag(w + L.d, ldd, B, d.nout, d.w, Z, ldz, d.nin, d.temperature, lnP, S(stream));
    return loglike_dense_impl(w + L.d, ldd, B, d.nout, d.Sfac ? d.Sfac : d.S, d.lds, d.Sfac != nullptr, Z, ldz, d.nin, d.temperature,
                                     w + L.part, lnP, stream);
}

extern "C" {

int linna_logprob_create(linna_ctx_t* ctx, linna_net_t* net, const linna_logprob_desc_t* desc, linna_logprob_t** out) try {
    if (!net || !desc || !out) { set_error("logprob_create: null argument"); return LINNA_ERR_INVALID; }
    CHECK_STRUCT(desc, linna_logprob_desc_t, "logprob_create");
    if (desc->nin != net->in_size || desc->nout != net->out_size) {
        set_error("logprob_create: network is %d->%d, descriptor says %d->%d", net->in_size, net->out_size, desc->nin, desc->nout);
        return LINNA_ERR_INVALID;
    }
    if (!desc->w && !desc->S) { set_error("logprob_create: need S (dense) or w (diagonal)"); return LINNA_ERR_INVALID; }
    if (!(desc->temperature > 0.f)) { set_error("logprob_create: temperature must be > 0"); return LINNA_ERR_INVALID; }
    linna_logprob* lp = new linna_logprob{ctx, net, *desc};
    lp->dense_tri = net_stream_dense_tri(-1);
    const NsDense dn = lp->dense();
    const bool want_dense = !desc->w && desc->S && !desc->outmap.cexp &&
                            !(getenv("LINNA_DENSE_FUSED") && getenv("LINNA_DENSE_FUSED")[0] == '0');
    if (want_dense && net_stream_dense_eligible(net->Lfull.data(), (int)net->Lfull.size(), net->in_size, dn)) {
        lp->dense_fused = true;
        if (lp->packed.alloc(net_stream_dense_packed_floats(net->Lfull.data(), (int)net->Lfull.size(), net->in_size, dn)) != LINNA_OK) {
            set_error("logprob_create: hipMalloc(weight stream) failed");
            delete lp; return LINNA_ERR_HIP;
        }
    } else if (net_stream_eligible(net->Lfull.data(), (int)net->Lfull.size(), net->in_size)) {
        const size_t nf = net_stream_packed_floats(net->Lfull.data(), (int)net->Lfull.size(), net->in_size);
        lp->grad_fused = net_stream_has_grad(net->Lfull.data(), (int)net->Lfull.size(), net->in_size) && !desc->outmap.cexp &&
                         !(getenv("LINNA_DISABLE_FUSED_GRAD") && getenv("LINNA_DISABLE_FUSED_GRAD")[0] == '1');
        if (lp->packed.alloc(nf) != LINNA_OK) {
            set_error("logprob_create: hipMalloc(weight stream) failed");
            delete lp; return LINNA_ERR_HIP;
        }
    }
    // lnP + gradient in one launch for the networks the MLP-only fused gradient does not cover (residual blocks, SPLIT
    // segments): forward program + dX chain in one weight stream, gates from the activations the same launch stored
    if (!lp->grad_fused && desc->w && desc->gscale && !desc->outmap.cexp && !net->has_inskip &&
        !(getenv("LINNA_DISABLE_FUSED_GRAD") && getenv("LINNA_DISABLE_FUSED_GRAD")[0] == '1') &&
        net_stream_dxi_eligible(net->L.data(), (int)net->L.size(), net->in_size)) {
        if (lp->packed_g2.alloc(net_stream_dxi_packed_floats(net->L.data(), (int)net->L.size(), net->in_size)) == LINNA_OK) lp->grad2 = true;
    }
    *out = lp;
    return LINNA_OK;
} LINNA_CATCH_INT
int linna_logprob_destroy(linna_logprob_t* lp) try {
    if (lp) { lp->packed.release(); lp->packed_g2.release(); }
    delete lp;
    return LINNA_OK;
} LINNA_CATCH_INT
int linna_weights_changed(linna_ctx_t*) try { g_weights_epoch.fetch_add(1); return LINNA_OK; } LINNA_CATCH_INT
int linna_program_describe(const linna_layer_t* layers, int nlayers, int in_size, int rows, int dense_nout, char* buf, size_t n) try {
    if (!layers || nlayers < 1 || !buf || !n) { set_error("program_describe: bad arguments"); return LINNA_ERR_INVALID; }
    CHECK_STRUCT(layers, linna_layer_t, "program_describe");
    for (int i = 1; i < nlayers; ++i) CHECK_STRUCT(layers + i, linna_layer_t, "program_describe");
    // (pointers are only compared, never read: a placeholder stands for the dense inverse covariance)
    static float dummy;
    if (dense_nout == -1) return net_stream_describe(layers, nlayers, in_size, 3, nullptr, rows, 1, buf, n);   // the one-launch gradient's program
    const int dn_cols = dense_nout < -1 ? -dense_nout : dense_nout;         // < -1: the factored form (chi^2 = |d L|^2) of -dense_nout columns
    NsDense dn{&dummy, (dn_cols + 3) & ~3, nullptr, nullptr, dense_nout < -1 ? 1 : 0, net_stream_dense_tri(-1)};
    return net_stream_describe(layers, nlayers, in_size, 0, dn_cols > 0 ? &dn : nullptr, rows, 1, buf, n);
} LINNA_CATCH_INT
int linna_dense_tri(int mode) try {
    if (mode < -1 || mode > 2) { set_error("linna_dense_tri: %d (-1 query, 0, 1 or 2)", mode); return LINNA_ERR_INVALID; }
    return net_stream_dense_tri(mode);
} LINNA_CATCH_INT
// which launches of linna_slice_half_step are folded into their neighbours (bit 0: the one stepping-out round's logic into
// the first shrinking round; bit 1: the set-up into the first evaluation's prologue; A/B switch, results identical either way)
static std::atomic<int> g_slice_fusion{7};
static bool slice_derive_enabled() { return (g_slice_fusion.load() & 1) != 0; }
int linna_slice_fusion(int mode) try {
    if (mode < -1 || mode > 7) { set_error("linna_slice_fusion: %d (-1 query, or a mask of bits 0-2)", mode); return LINNA_ERR_INVALID; }
    return mode < 0 ? g_slice_fusion.load() : g_slice_fusion.exchange(mode);
} LINNA_CATCH_INT
int linna_engine_rows(int rows) try {
    const int prev = net_stream_force_rows(rows);
    if (prev < 0) { set_error("linna_engine_rows: %d (0, 4, 8 or 16)", rows); return LINNA_ERR_INVALID; }
    // the packed-stream copies are cached per (16-row | small-batch) layout, not per engine: whatever was laid out under the
    // previous setting is re-laid before the next launch
    if (prev != rows) g_weights_epoch.fetch_add(1);
    return prev;
} LINNA_CATCH_INT
size_t linna_logprob_ws_bytes(const linna_logprob_t* lp, int B, int with_grad) try {
    return (lp_layout(lp, B, with_grad).total + 16) * sizeof(float);
} LINNA_CATCH_SIZE

int linna_logprob_eval(linna_logprob_t* lp, const float* Z, int ldz, int B, void* ws, float* lnP, float* TH, int ldt,
                       void* stream) try {
    if (!lp || !Z || !ws || !lnP || B < 1) { set_error("logprob_eval: bad arguments"); return LINNA_ERR_INVALID; }
    const LpLayout L = lp_layout(lp, B, 0);
    return lp_forward(lp, Z, ldz, B, static_cast<float*>(ws), L, lnP, TH, ldt, stream, false);
} LINNA_CATCH_INT

int linna_logprob_eval_if(linna_logprob_t* lp, const float* Z, int ldz, int B, void* ws, float* lnP, float* TH, int ldt,
                          const int* gate, void* stream) try {
    if (!lp || !Z || !ws || !lnP || B < 1) { set_error("logprob_eval_if: bad arguments"); return LINNA_ERR_INVALID; }
    const LpLayout L = lp_layout(lp, B, 0);
    return lp_forward(lp, Z, ldz, B, static_cast<float*>(ws), L, lnP, TH, ldt, stream, false, gate);
} LINNA_CATCH_INT

// `list` / `count` / `mul`: only the trial points list[0 .. count[0] * mul) are evaluated (device-side count; the launch is
// sized for all nrep * ns); `b_engine`: the batch size the engine is chosen for (the expected number of live rows)
// the first shrinking round behind one stepping-out round: what the evaluation needs to place its own trials (NsArgs::sl_*)
struct SliceDerive { const float* Z0; const float* L; const float* R; const float* Ze; int m, nt; uint64_t seed; const int* step_dev; int stream_id; const int* flags; };
static int lp_eval_slice_points(linna_logprob_t* lp, const float* coords, int ldc, int ndim, const int* S_idx, int ns,
                                const float* DIR, int ldd, const float* w, int nrep, float* lnP, const int* gate,
                                const int* list, const int* count, int mul, int b_engine, void* stream, const SliceDerive* sd = nullptr,
                                const SliceBegin* sb = nullptr) {
    if (!lp || !coords || !S_idx || !DIR || !w || !lnP || ns < 1 || nrep < 1) {
        set_error("logprob_eval_slice_points: bad arguments"); return LINNA_ERR_INVALID;
    }
    const linna_logprob_desc_t& d = lp->d;
    if (ndim != d.nin) { set_error("logprob_eval_slice_points: ndim %d, log-probability has %d parameters", ndim, d.nin); return LINNA_ERR_INVALID; }
    if (!fused_enabled() || !lp->packed.ready() || (d.outmap.cexp && (!d.w || !d.outmap.cpost || !d.outmap.cshift2)) ||
        (!d.w && !lp->dense_fused) || d.nin > 64) {
        set_error("logprob_eval_slice_points: this log-probability does not run the whole-network kernel");
        return LINNA_ERR_UNSUPPORTED;          // the caller falls back to linna_slice_points + linna_logprob_eval_if
    }
    const float* packed = nullptr; int rows = 16;
    TRY(lp_refresh_stream(lp, b_engine > 0 ? b_engine : nrep * ns, stream, &packed, &rows));
    const linna_net* n = lp->net;
    NsMove mv{const_cast<float*>(coords), ldc, nullptr, S_idx, w, 0, list, ns, 0ull, count, mul, 0, 0.f, nullptr, 1};
    mv.sb = sb;
    if (sd) {
        mv.sl_Z0 = sd->Z0; mv.sl_L = sd->L; mv.sl_R = sd->R; mv.sl_Zt = sd->Ze; mv.sl_m = sd->m; mv.sl_nt = sd->nt;
        mv.sl_seed = sd->seed; mv.sl_step = sd->step_dev; mv.sl_stream = sd->stream_id; mv.sl_flags = sd->flags;
    }
    const NsDense dn = lp->dense();
    const bool df = lp->dense_fused;
    return launch_net_stream(n->Lfull.data(), (int)n->Lfull.size(), n->in_size, packed, DIR, ldd, nrep * ns, d.nin, d.is_flat, d.a1,
                             d.a2, d.log10_flag, d.xmean, d.xstd, df ? nullptr : d.outmap.cscale, df ? nullptr : d.outmap.cshift,
                             df ? nullptr : d.w, d.temperature, lnP, nullptr, 0, nullptr, 0, &mv, nullptr, gate, rows,
                             df ? &dn : nullptr, S(stream), d.outmap.cexp ? d.outmap.cpost : nullptr,
                             d.outmap.cexp ? d.outmap.cshift2 : nullptr);
}
int linna_logprob_eval_slice_points(linna_logprob_t* lp, const float* coords, int ldc, int ndim, const int* S_idx, int ns,
                                    const float* DIR, int ldd, const float* w, int nrep, float* lnP, const int* gate,
                                    void* stream) try {
    return lp_eval_slice_points(lp, coords, ldc, ndim, S_idx, ns, DIR, ldd, w, nrep, lnP, gate, nullptr, nullptr, 0, 0, stream);
} LINNA_CATCH_INT

// One half step of the ensemble slice sampler (zeus behind sampler.py:728-735) in ONE call: the differential-move directions
// and slice heights, `nexp_rounds` speculative stepping-out rounds of `m_sched[r]` bracket ends per side, `nshr_rounds`
// shrinking rounds of `nt_sched[r]` trials with the commit in the last one -- 1 + 2 (nexp_rounds + nshr_rounds) launches, none of which the host waits for.
// Every evaluation is the whole-network kernel with the trial points formed in its prologue (never written to memory);
// rounds behind the one that finished the last walker are gated off on the device.
int linna_slice_half_step(linna_logprob_t* lp, float* coords, int ldc, int ndim, float* logp, const int* S_idx, int ns,
                          const float* ccoords, int ldcc, const int* C_idx, int nc, const float* mu, uint64_t seed,
                          int* step_dev, int half, const int* m_sched, int nexp_rounds, const int* nt_sched, int nshr_rounds,
                          float* DIR, int ldd, float* state, int* flags, float* W, float* Wd, float* Zt, int* list, int* counters,
                          int zero_totals, int bump_step, const int* expect_rows, int maxsteps, void* stream) try {
    if (!lp || !coords || !logp || !S_idx || !ccoords || !C_idx || !mu || !step_dev || !DIR || !state || !flags || !W || !Wd ||
        !Zt || !list || !counters || ns < 1 || nc < 2 || !m_sched || !nt_sched || nexp_rounds < 1 || nshr_rounds < 1 || (half != 0 && half != 1)) {
        set_error("slice_half_step: bad arguments"); return LINNA_ERR_INVALID;
    }
    if (maxsteps < 1) { set_error("slice_half_step: maxsteps %d < 1", maxsteps); return LINNA_ERR_INVALID; }
    for (int r = 0; r < nexp_rounds; ++r) if (m_sched[r] < 1) { set_error("slice_half_step: m_sched[%d] = %d", r, m_sched[r]); return LINNA_ERR_INVALID; }
    for (int r = 0; r < nshr_rounds; ++r) if (nt_sched[r] < 1) { set_error("slice_half_step: nt_sched[%d] = %d", r, nt_sched[r]); return LINNA_ERR_INVALID; }
    const linna_logprob_desc_t& d = lp->d;
    if (ndim != d.nin) { set_error("slice_half_step: ndim %d, log-probability has %d parameters", ndim, d.nin); return LINNA_ERR_INVALID; }
    if (!fused_enabled() || !lp->packed.ready() || (d.outmap.cexp && (!d.w || !d.outmap.cpost || !d.outmap.cshift2)) ||
        (!d.w && !lp->dense_fused) || d.nin > 64) {
        set_error("slice_half_step: this log-probability does not run the whole-network kernel");
        return LINNA_ERR_UNSUPPORTED;          // the caller falls back to the round-by-round entries
    }
    float* const Z0 = state; float* const L = state + ns; float* const R = state + 2 * ns;
    float* const Wacc = state + 3 * ns; float* const Zacc = state + 4 * ns;
    hipStream_t st = S(stream);
    // the set-up of the half step: a launch of its own, or done by the first evaluation in its prologue (SliceBegin)
    const bool begin_fused = (g_slice_fusion.load() & 2) != 0;
    const SliceBegin sb{logp, ccoords, ldcc, C_idx, nc, mu, seed, step_dev, half, m_sched[0], DIR, ldd, Z0, L, R, flags, counters,
                        nexp_rounds + nshr_rounds, zero_totals, maxsteps};
    if (!begin_fused)
        TRY(launch_slice_begin(logp, S_idx, ns, ccoords, ldcc, C_idx, nc, ndim, mu, seed, step_dev, half, DIR, ldd, Z0, L, R, flags, W, m_sched[0],
                               counters, nexp_rounds + nshr_rounds, zero_totals, maxsteps, st));
    int slot = 4;
    // ONE stepping-out round (small ensembles): its logic kernel is not launched -- the first shrinking round's evaluation
    // derives its trial points from the stepping-out round's results in its prologue (NsArgs::sl_*), and the first shrinking
    // round's logic kernel does the bookkeeping of both.  The lnP of that round go to W (whose bracket ends are spent).
    const bool derive = slice_derive_enabled() && nexp_rounds == 1 && m_sched[0] <= 16 && nt_sched[0] <= 32 && nt_sched[0] <= 2 * m_sched[0];
    // rounds after the first evaluate only the walkers still active: the logic kernel of round r lists their trial points
    // (list[pos * nrep + j] = j ns + k, pos = the walker's rank among the active ones) and counts them in counters[slot];
    // round r + 1's launch is sized for all of them, runs the engine chosen for the expected number (`expect_rows`: what the
    // caller has seen in the usage counters of its earlier calls; without it a quarter of the walkers per round) and leaves
    // at the counted one.  Because only those walkers are evaluated, the later rounds can look
    // further ahead for nothing (m_sched / nt_sched grow) and the call needs few rounds.
    for (int r = 0; r < nexp_rounds; ++r, ++slot) {
        const int m = m_sched[r], m_next = r + 1 < nexp_rounds ? m_sched[r + 1] : 0;
        TRY(lp_eval_slice_points(lp, coords, ldc, ndim, S_idx, ns, DIR, ldd, W, 2 * m, Zt, nullptr, r > 0 ? list : nullptr,
                                 r > 0 ? counters + slot - 1 : nullptr, 2 * m, expect_rows && r > 0 ? std::max(1, expect_rows[r]) : std::max(1, (2 * m * ns) >> (2 * r)), stream, nullptr,
                                 r == 0 && begin_fused ? &sb : nullptr));
        if (!derive)
            TRY(launch_slice_expand_multi(Z0, Zt, L, R, S_idx, flags, ns, m, m_next, counters, slot, r > 0 ? slot - 1 : -1, W, Wd, list, seed,
                                          step_dev, 2 + half, nt_sched[0], st));
    }
    int trials = 0;
    for (int r = 0; r < nshr_rounds; ++r, ++slot) {
        const int nt = nt_sched[r], nt_next = r + 1 < nshr_rounds ? nt_sched[r + 1] : 0;
        trials += nt;
        const bool dv = derive && r == 0;
        SliceDerive sd{Z0, L, R, Zt, m_sched[0], nt, seed, step_dev, 2 + half, flags};
        TRY(lp_eval_slice_points(lp, coords, ldc, ndim, S_idx, ns, DIR, ldd, Wd, nt, dv ? W : Zt, nullptr, r > 0 ? list : nullptr,
                                 r > 0 ? counters + slot - 1 : nullptr, nt,
                                 expect_rows && r > 0 ? std::max(1, expect_rows[nexp_rounds + r]) : std::max(1, (nt * ns) >> (2 * r)), stream, dv ? &sd : nullptr));
        const bool last = r + 1 == nshr_rounds;         // the commit (and the step counter) ride in the last round's logic kernel
        SliceRound sr{Z0, dv ? W : Zt, L, R, S_idx, Wd, flags, Wacc, Zacc, ns, counters, slot, r > 0 ? slot - 1 : -1, nt, nt_next, trials, list,
                      seed, step_dev, 2 + half, last ? coords : nullptr, ldc, ndim, logp, DIR, ldd, last && bump_step ? 1 : 0,
                      Zt, dv ? m_sched[0] : 0, 4};
        TRY(launch_slice_shrink_multi(sr, st));
    }
    return LINNA_OK;
} LINNA_CATCH_INT

int linna_stretch_half_step(linna_logprob_t* lp, float* coords, int ldc, int ndim, float* logp, const int* S_idx, int ns,
                            const float* ccoords, int ldcc, const int* C_idx, int nc, uint64_t seed, const int* step_dev,
                            int step_offset, int stream_id, float a, int* naccept, void* stream) try {
    if (!lp || !coords || !logp || !S_idx || !ccoords || !C_idx || !step_dev || ns < 1 || nc < 1) {
        set_error("stretch_half_step: bad arguments"); return LINNA_ERR_INVALID;
    }
    const linna_logprob_desc_t& d = lp->d;
    if (ndim != d.nin) { set_error("stretch_half_step: ndim %d, log-probability has %d parameters", ndim, d.nin); return LINNA_ERR_INVALID; }
    if (!fused_enabled() || !lp->packed.ready() || (d.outmap.cexp && (!d.w || !d.outmap.cpost || !d.outmap.cshift2)) ||
        (!d.w && !lp->dense_fused) || d.nin > 64) {
        set_error("stretch_half_step: this log-probability does not run the whole-network kernel");
        return LINNA_ERR_UNSUPPORTED;          // the caller falls back to propose / eval / accept
    }
    const float* packed = nullptr; int rows = 16;
    TRY(lp_refresh_stream(lp, ns, stream, &packed, &rows));
    const linna_net* n = lp->net;
    NsMove mv{coords, ldc, logp, S_idx, ccoords, ldcc, C_idx, nc, seed, step_dev, step_offset, stream_id, a, naccept, 0};
    const NsDense dn = lp->dense();
    const bool df = lp->dense_fused;
    return launch_net_stream(n->Lfull.data(), (int)n->Lfull.size(), n->in_size, packed, nullptr, 0, ns, d.nin, d.is_flat, d.a1,
                             d.a2, d.log10_flag, d.xmean, d.xstd, df ? nullptr : d.outmap.cscale, df ? nullptr : d.outmap.cshift,
                             df ? nullptr : d.w, d.temperature, nullptr, nullptr, 0, nullptr, 0, &mv, nullptr, nullptr, rows,
                             df ? &dn : nullptr, S(stream), d.outmap.cexp ? d.outmap.cpost : nullptr,
                             d.outmap.cexp ? d.outmap.cshift2 : nullptr);
} LINNA_CATCH_INT


int linna_stretch_run(linna_logprob_t* lp, float* coords, int ldc, int ndim, float* logp, int nw, const int* splits,
                      int split_stride, int nsteps, uint64_t seed, const int* step_dev, int step_offset, float a, int* naccept,
                      float* chain, float* logps, void* stream) try {
    if (!lp || !coords || !logp || !splits || !step_dev || nw < 2 || (nw & 1) || nsteps < 1 || split_stride < 0 ||
        (chain != nullptr) != (logps != nullptr)) {
        set_error("stretch_run: bad arguments"); return LINNA_ERR_INVALID;
    }
    const linna_logprob_desc_t& d = lp->d;
    if (ndim != d.nin) { set_error("stretch_run: ndim %d, log-probability has %d parameters", ndim, d.nin); return LINNA_ERR_INVALID; }
    if (!fused_enabled() || !lp->packed.ready() || (d.outmap.cexp && (!d.w || !d.outmap.cpost || !d.outmap.cshift2)) ||
        (!d.w && !lp->dense_fused) || d.nin > 64) {
        set_error("stretch_run: this log-probability does not run the whole-network kernel");
        return LINNA_ERR_UNSUPPORTED;          // the caller loops over linna_stretch_half_step / the three-launch form
    }
    const int ns = nw / 2;
    const float* packed = nullptr; int rows = 16;
    TRY(lp_refresh_stream(lp, ns, stream, &packed, &rows));
    const linna_net* n = lp->net;
    const NsDense dn = lp->dense();
    const bool df = lp->dense_fused;
    for (int i = 0; i < nsteps; ++i) {
        const int* sp = splits + (size_t)i * split_stride;
        for (int h = 0; h < 2; ++h) {
            NsMove mv{coords, ldc, logp, sp + h * ns, coords, ldc, sp + (1 - h) * ns, ns, seed, step_dev, step_offset + i, h, a, naccept, 0};
            if (chain) { mv.chain = chain + (size_t)i * nw * ndim; mv.lps = logps + (size_t)i * nw; }
            TRY(launch_net_stream(n->Lfull.data(), (int)n->Lfull.size(), n->in_size, packed, nullptr, 0, ns, d.nin, d.is_flat, d.a1,
                                  d.a2, d.log10_flag, d.xmean, d.xstd, df ? nullptr : d.outmap.cscale, df ? nullptr : d.outmap.cshift,
                                  df ? nullptr : d.w, d.temperature, nullptr, nullptr, 0, nullptr, 0, &mv, nullptr, nullptr, rows,
                                  df ? &dn : nullptr, S(stream), d.outmap.cexp ? d.outmap.cpost : nullptr,
                                  d.outmap.cexp ? d.outmap.cshift2 : nullptr));
        }
    }
    return LINNA_OK;
} LINNA_CATCH_INT

// ---- convergence statistics of a chain (autocorr.hip)
static bool ac_shape_ok(int ndim, int nwp) { return ndim >= 1 && nwp >= 64 && (nwp & 63) == 0; }
int linna_chain_append_t(linna_ctx_t*, const float* block, int ldb, int nsteps, int nw, int ndim, int wstride, float* CT, int nwp,
                         int64_t row0, void* stream) try {
    if (!block || !CT || nsteps < 1 || nw < 1 || wstride < 1 || ldb < ndim || row0 < 0 || !ac_shape_ok(ndim, nwp) ||
        nw > nwp || nsteps > 65535 || (size_t)64 * (ndim + 1) * sizeof(float) > 64 * 1024) {
        set_error("chain_append_t: bad arguments"); return LINNA_ERR_INVALID;
    }
    return launch_chain_append_t(block, ldb, nsteps, nw, ndim, wstride, CT, nwp, row0, S(stream));
} LINNA_CATCH_INT
static bool ac_cover_ok(int nwp, int nwc, int nlive) { return nwc >= 64 && (nwc & 63) == 0 && nwc <= nwp && nlive >= 1 && nlive <= nwc; }
int linna_acorr_update(linna_ctx_t*, const float* CT, int ndim, int nwp, int nwc, int64_t a0, int64_t a1, int64_t lo, int64_t hi, int k0,
                       int k1, double* Ssum, double* Tsum, int remove, void* stream) try {
    if (!CT || !Ssum || !ac_shape_ok(ndim, nwp) || !ac_cover_ok(nwp, nwc, 1) || k0 < 0 || k1 < k0 || (k0 & 31) || (k1 & 31) || lo < 0 ||
        hi < lo || a0 < lo || a1 > hi || a1 < a0 || (k1 - k0) / 32 > 4 * 65535 || hi > 0x7fffff00) {
        set_error("acorr_update: bad arguments (lag ranges are multiples of 32, anchors inside [lo, hi))"); return LINNA_ERR_INVALID;
    }
    return launch_acorr_update(CT, ndim, nwp, nwc, a0, a1, lo, hi, k0, k1, Ssum, Tsum, remove, S(stream));
} LINNA_CATCH_INT
size_t linna_acorr_scratch_bytes(int ndim, int nwc, int kuse) try {
    if (ndim < 1 || nwc < 64 || (nwc & 63) || kuse < 0) return 0;
    return acorr_scratch_doubles(ndim * nwc, kuse, ndim) * sizeof(double);
} LINNA_CATCH_SIZE
int linna_acorr_tau(linna_ctx_t*, const float* CT, int ndim, int nwp, int nwc, int nlive, int64_t lo, int64_t hi, int kuse,
                    const double* Ssum, const double* Tsum, double c, double* scratch, double* out, void* stream) try {
    if (!CT || !Ssum || !Tsum || !scratch || !out || !ac_shape_ok(ndim, nwp) || !ac_cover_ok(nwp, nwc, nlive) || lo < 0 || hi <= lo ||
        kuse < 0 || (int64_t)kuse > hi - lo - 1 || kuse / 32 + 1 > 65535 || !(c > 0.0) || hi > 0x7fffff00) {
        set_error("acorr_tau: bad arguments (0 <= kuse <= hi - lo - 1)"); return LINNA_ERR_INVALID;
    }
    return launch_acorr_tau(CT, ndim, nwp, nwc, nlive, lo, hi, kuse, Ssum, Tsum, c, scratch, out, S(stream));
} LINNA_CATCH_INT
int linna_chain_meanstd(linna_ctx_t*, const float* CT, int ndim, int nwp, int nws, int64_t t0, int64_t tm, int64_t t1, double* out,
                        void* stream) try {
    if (!CT || !out || !ac_shape_ok(ndim, nwp) || nws < 1 || nws > nwp || t0 < 0 || tm < t0 || t1 < tm || ndim > 65535) {
        set_error("chain_meanstd: bad arguments"); return LINNA_ERR_INVALID;
    }
    return launch_chain_meanstd(CT, ndim, nwp, nws, t0, tm, t1, out, S(stream));
} LINNA_CATCH_INT

}  // extern "C"

// lnP and its gradient at Z; `leap` (hm_* of an NsGrad, the rest unset): the leapfrog's kick and drift behind it -- in the
// finish of the one-launch forms, as a launch of its own behind the others
static int logprob_grad_impl(linna_logprob_t* lp, const float* Z, int ldz, int B, void* ws, float* lnP, float* G, int ldg,
                             const NsGrad* leap, void* stream) {
    if (!lp || !Z || !ws || !lnP || !G || B < 1) { set_error("logprob_grad: bad arguments"); return LINNA_ERR_INVALID; }
    const linna_logprob_desc_t& d = lp->d;
    if (d.outmap.cexp) { set_error("logprob_grad: ypositive (exp) output map has no gradient path"); return LINNA_ERR_UNSUPPORTED; }
    if (!d.gscale || (!d.w && !d.Ssym)) { set_error("logprob_grad: descriptor lacks gscale / Ssym"); return LINNA_ERR_INVALID; }
    if (fused_enabled() && lp->packed.ready() && lp->grad_fused && d.w) {
        // lnP and d lnP / d z in ONE launch: forward segments, turnaround, backward segments over W^T (net_stream.hip)
        const float* packed = nullptr; int rows = 16;
        TRY(lp_refresh_stream(lp, B, stream, &packed, &rows));
        const linna_net* n = lp->net;
        NsGrad gr{d.gscale, G, ldg, nullptr, 0, nullptr, nullptr, 0.f, 0.f};
        if (leap) { gr.hm_p = leap->hm_p; gr.hm_ldp = leap->hm_ldp; gr.hm_q = leap->hm_q; gr.hm_mass = leap->hm_mass; gr.hm_ek = leap->hm_ek; gr.hm_ed = leap->hm_ed; }
        return launch_net_stream(n->Lfull.data(), (int)n->Lfull.size(), n->in_size, packed, Z, ldz, B, d.nin, d.is_flat, d.a1, d.a2,
                                 d.log10_flag, d.xmean, d.xstd, d.outmap.cscale, d.outmap.cshift, d.w, d.temperature, lnP,
                                 nullptr, 0, nullptr, 0, nullptr, &gr, nullptr, rows, nullptr, S(stream));
    }
    const LpLayout L = lp_layout(lp, B, 1);
    float* w = static_cast<float*>(ws);
    const int ldx = ld4(d.nin), ldd = ld4(d.nout);
    if (fused_enabled() && lp->grad2 && lp->packed_g2.ready() && d.w) {
        // ONE launch for any network: forward segments (activations kept in the workspace), turnaround, dX chain down to
        // the input, prior map's derivative (net_stream.hip, GRAD + STORE == 2) -- six launches otherwise
        linna_net* n = lp->net;
        const int nl = (int)n->L.size();
        const int rows = net_stream_rows(B);
        const float* packed = nullptr;
        TRY(stream_copy_refresh(lp->packed_g2, n, rows, stream, &packed, 3));
        const FwdLayout f = fwd_layout(n, B);
        float* base = w + L.fwd;
        std::vector<float*> y(nl, nullptr), t(nl, nullptr);
        std::vector<int> ldy(nl, 0), ldt(nl, 0);
        for (int i = 0; i < nl; ++i) {
            if (i < nl - 1) { y[i] = base + f.y_off[i]; ldy[i] = ld4(n->L[i].N); }
            if (n->L[i].op == LINNA_OP_RESBLOCK) { t[i] = base + f.t_off[i]; ldt[i] = ld4(n->L[i].C); }
        }
        NsGrad gr{d.gscale, G, ldg, nullptr, 0, nullptr, nullptr, 0.f, 0.f};
        if (leap) { gr.hm_p = leap->hm_p; gr.hm_ldp = leap->hm_ldp; gr.hm_q = leap->hm_q; gr.hm_mass = leap->hm_mass; gr.hm_ek = leap->hm_ek; gr.hm_ed = leap->hm_ed; }
        return launch_net_stream_grad2(n->L.data(), nl, n->in_size, packed, Z, ldz, B, d.nin, d.is_flat, d.a1, d.a2, d.log10_flag,
                                       d.xmean, d.xstd, d.outmap.cscale, d.outmap.cshift, d.w, d.temperature, lnP, gr, y.data(),
                                       ldy.data(), t.data(), ldt.data(), rows, S(stream));
    }
    TRY(lp_forward(lp, Z, ldz, B, w, L, lnP, nullptr, 0, stream, true));
    if (d.w) {
        TRY(launch_loglike_diag_grad(w + L.d, ldd, B, d.nout, d.w, d.gscale, d.temperature, w + L.dh, ldd, S(stream)));
    } else {   // dH = -(1/T) * (D Ssym) * gscale
        GemmArgs a = gemm_zero();
        set_pair(a, 0, w + L.d, ldd, LAY_K, d.Ssym, d.lds, LAY_MN, d.nout);
        a.M = B; a.N = d.nout; a.C = w + L.dh; a.ldc = ldd; a.alpha0 = -1.f / d.temperature; a.cscale = d.gscale;
        TRY(gemm_launch(a, S(stream)));
    }
    TRY(linna_net_backward(lp->net, w + L.x0, ldx, B, w + L.fwd, w + L.bwd, w + L.dh, ldd, w + L.dx, ldx, 0, stream));
    TRY(launch_prior_map_bwd(Z, ldz, B, d.nin, d.is_flat, d.a1, d.a2, d.log10_flag, d.xstd, w + L.dx, ldx, G, ldg, S(stream)));
    if (leap) return launch_hmc_kick_drift(B, d.nin, leap->hm_mass, leap->hm_ek, leap->hm_ed, G, ldg, leap->hm_p, leap->hm_ldp, leap->hm_q, ldz, S(stream));
    return LINNA_OK;
}

extern "C" {

int linna_logprob_grad(linna_logprob_t* lp, const float* Z, int ldz, int B, void* ws, float* lnP, float* G, int ldg,
                       void* stream) try {
    return logprob_grad_impl(lp, Z, ldz, B, ws, lnP, G, ldg, nullptr, stream);
} LINNA_CATCH_INT

// One leapfrog step's gradient, kick and drift (HMCSampler.py:35-49): lnP and G = d lnP / d z at Q, then P += eps_kick G and
// Q += eps_drift P / mass.  ONE launch where linna_logprob_grad is one (the kick and the drift ride in its finish).
int linna_logprob_grad_leapfrog(linna_logprob_t* lp, float* Q, int ldq, int B, void* ws, float* lnP, float* G, int ldg, float* P,
                                int ldp, const float* mass, float eps_kick, float eps_drift, void* stream) try {
    if (!P || !mass || !Q) { set_error("logprob_grad_leapfrog: bad arguments"); return LINNA_ERR_INVALID; }
    NsGrad leap{nullptr, nullptr, 0, P, ldp, Q, mass, eps_kick, eps_drift};
    return logprob_grad_impl(lp, Q, ldq, B, ws, lnP, G, ldg, &leap, stream);
} LINNA_CATCH_INT

// ------------------------------------------------------------------ training
// scratch layout for the loss entry points: DELTA[B][ld] | U[B][ld] | partial[B][slots]
static size_t loss_scratch_floats(int B, int nout) { return (size_t)B * (2 * (size_t)ld4(nout) + gemm_slots(B, nout)); }

static int chi2_partials(const linna_loss_desc_t* d, int mode, const float* PRED, int ldp, const float* Y, int ldy,
                         const int* ROWS, int B, float* scratch, bool keepU, hipStream_t st) {
    const int ld = ld4(d->nout), slots = gemm_slots(B, d->nout);
    float* DELTA = scratch;
    float* U = scratch + (size_t)B * ld;
    float* part = scratch + 2 * (size_t)B * ld;
    TRY(launch_loss_delta(mode, PRED, ldp, Y, ldy, ROWS, B, *d, DELTA, ld, st));
    GemmArgs a = gemm_zero();
    set_pair(a, 0, DELTA, ld, LAY_K, d->Cinv, d->ldc, LAY_MN, d->nout);
    a.M = B; a.N = d->nout; a.C = keepU ? U : nullptr; a.ldc = ld;
    a.dotwith = DELTA; a.lddot = ld; a.dot_partial = part; a.dot_slots = slots;
    return gemm_launch(a, st);
}

size_t linna_loss_scratch_bytes(int B, int nout) try { return (loss_scratch_floats(B, nout) + 16) * sizeof(float); } LINNA_CATCH_SIZE

int linna_chi2_md(linna_ctx_t*, const linna_loss_desc_t* d, const float* Y, int ldy, int nrows, float* scratch,
                  float* den, void* stream) try {
    if (!d) { set_error("chi2_md: null loss descriptor"); return LINNA_ERR_INVALID; }
    CHECK_STRUCT(d, linna_loss_desc_t, "chi2_md");
    const int slots = gemm_slots(nrows, d->nout);
    TRY(chi2_partials(d, 1, nullptr, 0, Y, ldy, nullptr, nrows, scratch, false, S(stream)));
    return launch_loss_rows(0, scratch + 2 * (size_t)nrows * ld4(d->nout), slots, slots, nrows, nullptr, nullptr,
                            0.5f * (float)d->nout, den, S(stream));
} LINNA_CATCH_INT

int linna_chi2_ratio_loss_fwd_bwd(linna_ctx_t* ctx, const linna_loss_desc_t* d, const float* PRED, int ldp, const float* Y,
                                  int ldy, const float* den, const int* ROWS, int B, float* scratch, float* loss_rows,
                                  float* loss_mean, float* dPRED, int lddp, float inv_batch, void* stream) try {
    if (!d) { set_error("chi2_ratio_loss_fwd_bwd: null loss descriptor"); return LINNA_ERR_INVALID; }
    CHECK_STRUCT(d, linna_loss_desc_t, "chi2_ratio_loss_fwd_bwd");
    const int ld = ld4(d->nout), slots = gemm_slots(B, d->nout);
    hipStream_t st = S(stream);
    if (ctx && d->nout <= 64 && lddp >= 0) {
        // five launches of 5-14 us each (delta, U = delta Cinv, row sums, mean, gradient) for 2 MFLOP: one kernel
        if (ctx->loss_fused < 0) {
            const char* e = getenv("LINNA_LOSS_FUSED");
            ctx->loss_fused = (e && e[0] == '0') ? 0 : 1;
        }
        unsigned* const cnt = ctx->loss_fused == 1 ? ctx_counters(ctx, st) : nullptr;
        if (cnt)
            return launch_loss_fused_small(PRED, ldp, Y, ldy, ROWS, B, *d, den, inv_batch, loss_rows, loss_mean, dPRED, lddp,
                                           cnt, st);
    }
    TRY(chi2_partials(d, 0, PRED, ldp, Y, ldy, ROWS, B, scratch, dPRED != nullptr, st));
    TRY(launch_loss_rows(1, scratch + 2 * (size_t)B * ld, slots, slots, B, den, ROWS, 0.f, loss_rows, st));
    if (loss_mean) TRY(launch_sum_scale(loss_rows, B, inv_batch, loss_mean, st));
    if (dPRED) TRY(launch_loss_grad(scratch + (size_t)B * ld, ld, Y, ldy, ROWS, B, d->nout, d->data_norm, den, inv_batch, dPRED, lddp, st));
    return LINNA_OK;
} LINNA_CATCH_INT

int linna_val_rows(linna_ctx_t*, const linna_loss_desc_t* d, const float* PRED, int ldp, const float* Y, int ldy,
                   const float* den, int B, float* scratch, float* loss_rows, float* frac_rows, void* stream) try {
    if (!d) { set_error("val_rows: null loss descriptor"); return LINNA_ERR_INVALID; }
    CHECK_STRUCT(d, linna_loss_desc_t, "val_rows");
    const int ld = ld4(d->nout), slots = gemm_slots(B, d->nout);
    hipStream_t st = S(stream);
    TRY(chi2_partials(d, 0, PRED, ldp, Y, ldy, nullptr, B, scratch, false, st));
    TRY(launch_loss_rows(1, scratch + 2 * (size_t)B * ld, slots, slots, B, den, nullptr, 0.f, loss_rows, st));
    TRY(chi2_partials(d, 2, PRED, ldp, Y, ldy, nullptr, B, scratch, false, st));
    return launch_val_frac(scratch + 2 * (size_t)B * ld, slots, slots, B, den, frac_rows, st);
} LINNA_CATCH_INT

int linna_val_metrics(linna_ctx_t*, const float* loss_rows, const float* frac_rows, int n, const float* last_train_loss, float* out,
                      void* stream) try {
    if (!loss_rows || !frac_rows || !out || n < 1 || n > (1 << 16)) {
        set_error("val_metrics: bad arguments (1 <= n <= 65536 validation rows)"); return LINNA_ERR_INVALID;
    }
    return launch_val_metrics(loss_rows, frac_rows, n, last_train_loss, out, S(stream));
} LINNA_CATCH_INT

int linna_gather_xform(linna_ctx_t*, const float* X, int ldx, const int* ROWS, int B, int nin, const int* lg,
                       const float* xmean, const float* xstd, float* XB, int ldxb, void* stream) try {
    return launch_gather_xform(X, ldx, ROWS, B, nin, lg, xmean, xstd, XB, ldxb, S(stream));
} LINNA_CATCH_INT

int linna_adamw_step(linna_ctx_t*, float* p, const float* g, float* m, float* v, size_t n, float* hyper, int* step_dev,
                     float b1, float b2, float eps, int prepared, void* stream) try {
    if (!p || !g || !m || !v || !hyper || !step_dev) { set_error("adamw_step: null pointer"); return LINNA_ERR_INVALID; }
    g_weights_epoch.fetch_add(1);
    // The step counter and the bias corrections are a single-thread launch of their own (folding them into the update
    // with an arrival counter measured 5 us slower than the extra launch): in front of the update here, or -- `prepared`
    // -- already advanced by linna_net_forward_loss, in the launch that takes the batch mean of the loss.
    return launch_adamw(p, g, m, v, n, hyper, prepared ? nullptr : step_dev, b1, b2, eps, S(stream));
} LINNA_CATCH_INT

// AdamW over the network's flat parameter buffer AND the re-layout of the updated weights into the two weight streams a
// training step reads (linna_net_forward_loss's and the backward's dX chain), in ONE launch: what linna_adamw_step
// followed by the two lazy re-layouts of the next step does in three.  `B`: the batch size the step runs at (it
// selects the engine, hence the stream layout).  LINNA_ERR_UNSUPPORTED when the network does not train through those
// two streams, or `params[n]` is not exactly its tensors back to back: the caller then uses linna_adamw_step.
// The placement tables of the flat parameter buffer `p[n]` in the two training streams (net_stream_adamw_args), cached.
static int net_ensure_as_args(linna_net_t* net, int B, const float* p, size_t n) {
    static const bool off = getenv("LINNA_ADAMW_STREAMS") && getenv("LINNA_ADAMW_STREAMS")[0] == '0';
    const int merged = net_tb_usable(net, B) ? 1 : 0;
    if (off || net->stream_loss != 1 || (!merged && (net->stream_bwd[0] != 1 || !net->packed_loss.ready() || !net->packed_dx[0].ready()))) {
        set_error("the network does not train through the whole-network streams"); return LINNA_ERR_UNSUPPORTED;
    }
    const int rows = net_stream_rows(B), k = rows < 16 ? 1 : 0;
    if (net->as_state < 0 || net->as_params != p || net->as_n != n || net->as_k != k || net->as_merged != merged) {
        net->as_params = p; net->as_n = n; net->as_k = k; net->as_merged = merged;
        net->as_state = (merged ? net_stream_adamw_args(net->L.data(), (int)net->L.size(), net->in_size, rows, p, n, net->packed_tb.buf[k],
                                                        &net->loss_dn, nullptr, &net->as_args, 1)
                                : net_stream_adamw_args(net->L.data(), (int)net->L.size(), net->in_size, rows, p, n, net->packed_loss.buf[k],
                                                        &net->loss_dn, net->packed_dx[0].buf[k], &net->as_args)) == LINNA_OK ? 1 : 0;
        net->upd_state = -1;
    }
    return net->as_state == 1 ? LINNA_OK : LINNA_ERR_UNSUPPORTED;   // (the error text is net_stream_adamw_args')
}

int linna_net_adamw_step(linna_net_t* net, int B, float* p, const float* g, float* m, float* v, size_t n, float* hyper,
                         int* step_dev, float b1, float b2, float eps, int prepared, void* stream) try {
    if (!net || !p || !g || !m || !v || !hyper || !step_dev || B < 1) { set_error("net_adamw_step: bad arguments"); return LINNA_ERR_INVALID; }
    TRY(net_ensure_as_args(net, B, p, n));
    const int rows = net_stream_rows(B), k = rows < 16 ? 1 : 0;
    // both streams must hold the CURRENT weights and their constant parts before they are patched in place
    const float* dummy = nullptr;
    const bool merged = net->as_merged == 1;
    if (merged) {
        TRY(stream_copy_refresh(net->packed_tb, net, rows, stream, &dummy, 4, &net->loss_dn));
    } else {
        TRY(stream_copy_refresh(net->packed_loss, net, rows, stream, &dummy, 0, &net->loss_dn));
        TRY(stream_copy_refresh(net->packed_dx[0], net, rows, stream, &dummy, 1, nullptr));
    }
    if (!prepared) TRY(launch_adamw_prepare(hyper, step_dev, b1, b2, S(stream)));
    TRY(launch_adamw_streams(net->as_args, p, g, m, v, hyper, b1, b2, eps, S(stream)));
    const unsigned long long epoch = g_weights_epoch.fetch_add(1) + 1;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(S(stream), &cap);
    if (cap == hipStreamCaptureStatusNone) {
        if (merged) net->packed_tb.epoch[k] = epoch;
        else { net->packed_loss.epoch[k] = epoch; net->packed_dx[0].epoch[k] = epoch; }
    }
    return LINNA_OK;
} LINNA_CATCH_INT

// ONE optimiser step in ONE call and THREE launches: linna_net_train_step with AdamW in the epilogue of its grouped
// parameter-gradient launch -- every 64 x 64 gradient tile updates its block of the weight matrix (and its moments) as soon as
// it exists and writes the updated block into the two weight streams the next step reads.  For one rank (data-parallel
// training all-reduces the gradients between backward and update: linna_net_train_step + linna_net_adamw_step).
// LINNA_ERR_UNSUPPORTED -- before anything is launched -- when the network does not train through the streams, `params[n]`
// is not its tensors back to back, or some parameter gradient of the step falls outside the grouped launch.
int linna_net_train_step_update(linna_net_t* net, const linna_loss_desc_t* d, const float* X, int ldx, const int* ROWS, int B,
                                const int* lg, const float* xmean, const float* xstd, float* XB, int ldxb, void* fwd_ws, float* PRED,
                                int ldp, const float* YN, int ldyn, const float* den, float inv_batch, float* loss_rows,
                                float* loss_mean, float* dPRED, int lddp, void* bwd_ws, float* params, float* m, float* v,
                                size_t n, float* hyper, int* step_dev, float b1, float b2, float eps, void* stream) try {
    if (!net || !params || !m || !v || !hyper || !step_dev || !bwd_ws || B < 1) { set_error("net_train_step_update: bad arguments"); return LINNA_ERR_INVALID; }
    if (!d) { set_error("net_train_step_update: null loss descriptor"); return LINNA_ERR_INVALID; }
    CHECK_STRUCT(d, linna_loss_desc_t, "net_train_step_update");
    static const bool off = getenv("LINNA_ADAMW_IN_GEMM") && getenv("LINNA_ADAMW_IN_GEMM")[0] == '0';
    if (off || net->has_inskip) { set_error("net_train_step_update: switched off / input-skip network"); return LINNA_ERR_UNSUPPORTED; }
    TRY(net_train_ensure_loss(net, d, stream));
    TRY(net_ensure_as_args(net, B, params, n));
    if (net->upd_state < 0 || net->upd_B != B) {
        // every parameter gradient of the step must be a problem of the grouped launch, and the gradient pointers of the layer
        // table must mirror the parameter buffer (same distance for every tensor)
        linna_ctx* ctx = net->ctx;
        if (ctx && ctx->group < 0) { const char* e = getenv("LINNA_BWD_GROUP"); ctx->group = (e && e[0] == '0') ? 0 : 1; }
        bool ok = ctx && ctx->group == 1;
        int nprob = 0;
        long long diff = 0; bool have = false;
        auto same = [&](const float* w, const float* g) { if (!w) return; if (!g) { ok = false; return; } if (!have) { diff = w - g; have = true; } else if (w - g != diff) ok = false; };
        auto prob = [&](int M, int N) {
            GemmArgs a = gemm_zero();
            a.npairs = 1; a.p[0].alay = LAY_MN; a.p[0].blay = LAY_MN; a.p[0].lda = ld4(M); a.p[0].ldb = ld4(N); a.p[0].K = B; a.M = M; a.N = N;
            if (!gemm_group_ok(a)) ok = false;
            ++nprob;
        };
        for (const linna_layer_t& l : net->L) {
            if (l.op == LINNA_OP_LINEAR) { prob(l.N, l.K); same(l.W, l.gW); same(l.b, l.gb); }
            else if (l.op == LINNA_OP_RESBLOCK) {
                prob(l.N, l.C); prob(l.C, l.K); if (l.Ws) prob(l.N, l.K);
                same(l.W1, l.gW1); same(l.b1, l.gb1); same(l.W2, l.gW2); same(l.b2, l.gb2); same(l.Ws, l.gWs);
            } else ok = false;
        }
        net->upd_state = (ok && nprob <= GEMM_UPD_MAX) ? 1 : 0;
        net->upd_B = B;
    }
    if (net->upd_state != 1) { set_error("net_train_step_update: a parameter gradient of this network falls outside the grouped launch"); return LINNA_ERR_UNSUPPORTED; }
    const NetUpdate upd{params, m, v, n, hyper, b1, b2, eps};
    const bool merged = net->as_merged == 1;
    if (merged) {
        // TWO launches: forward + loss + dX chain (AdamW's step constants riding in it), then every parameter gradient with the
        // optimiser in the tiles' epilogue (the batch mean of the loss riding in it)
        TRY(net_train_merged_impl(net, d, X, ldx, ROWS, B, lg, xmean, xstd, XB, ldxb, fwd_ws, PRED, ldp, YN, ldyn, den, inv_batch,
                                  loss_rows, dPRED, lddp, bwd_ws, hyper, step_dev, b1, b2, stream));
        const GemmPost gp{loss_rows, loss_mean ? B : 0, inv_batch, loss_mean};
        TRY(net_backward_impl(net, XB, ldxb, B, fwd_ws, bwd_ws, dPRED, lddp, nullptr, 0, 1, stream, nullptr, &upd, true, &gp));
    } else {
        TRY(net_forward_loss_impl(net, d, X, ldx, ROWS, B, lg, xmean, xstd, XB, ldxb, fwd_ws, PRED, ldp, YN, ldyn, den, inv_batch,
                                  loss_rows, loss_mean, dPRED, lddp, hyper, step_dev, b1, b2, stream, true));
        const NsPost post{loss_rows, B, inv_batch, loss_mean, step_dev, hyper, b1, b2};
        TRY(net_backward_impl(net, XB, ldxb, B, fwd_ws, bwd_ws, dPRED, lddp, nullptr, 0, 1, stream, &post, &upd));
    }
    const int k = net_stream_rows(B) < 16 ? 1 : 0;
    const unsigned long long epoch = g_weights_epoch.fetch_add(1) + 1;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(S(stream), &cap);
    if (cap == hipStreamCaptureStatusNone) {
        if (merged) net->packed_tb.epoch[k] = epoch;
        else { net->packed_loss.epoch[k] = epoch; net->packed_dx[0].epoch[k] = epoch; }
    }
    return LINNA_OK;
} LINNA_CATCH_INT

// ------------------------------------------------------------------ moves
int linna_stretch_propose(linna_ctx_t*, const float* coords, int ldc, int ndim, const int* S_idx, int ns,
                          const float* ccoords, int ldcc, const int* C_idx, int nc, uint64_t seed, const int* step_dev,
                          int stream_id, float a, float* Q, int ldq, float* factors, void* stream) try {
    if (ns < 1 || nc < 1) { set_error("stretch_propose: empty walker set"); return LINNA_ERR_INVALID; }
    return launch_stretch_propose(coords, ldc, ndim, S_idx, ns, ccoords, ldcc, C_idx, nc, seed, step_dev, stream_id, a, Q,
                                  ldq, factors, S(stream));
} LINNA_CATCH_INT
int linna_stretch_accept(linna_ctx_t*, float* coords, int ldc, int ndim, float* logp, const int* S_idx, int ns,
                         const float* Q, int ldq, const float* logp_new, const float* factors, uint64_t seed,
                         const int* step_dev, int stream_id, int* naccept, void* stream) try {
    return launch_stretch_accept(coords, ldc, ndim, logp, S_idx, ns, Q, ldq, logp_new, factors, seed, step_dev, stream_id, naccept, S(stream));
} LINNA_CATCH_INT
int linna_hmc_init(linna_ctx_t*, int B, int ndim, const float* mass, uint64_t seed, const int* step_dev, const float* lnp,
                   const float* P0, int ldp0, float* P, int ldp, float* H0, void* stream) try {
    return launch_hmc_init(B, ndim, mass, seed, step_dev, lnp, P0, ldp0, P, ldp, H0, S(stream));
} LINNA_CATCH_INT
int linna_hmc_start(linna_ctx_t*, int B, int ndim, const float* mass, uint64_t seed, const int* step_dev, const float* lnp,
                    const float* P0, int ldp0, const float* G, int ldg, float eps_kick, float eps_drift, const float* X, int ldx,
                    float* P, int ldp, float* Q, int ldq, float* H0, void* stream) try {
    if (B < 1 || ndim < 1 || !mass || !step_dev || !lnp || !G || !X || !P || !Q || !H0) { set_error("hmc_start: bad arguments"); return LINNA_ERR_INVALID; }
    return launch_hmc_start(B, ndim, mass, seed, step_dev, lnp, P0, ldp0, G, ldg, eps_kick, eps_drift, X, ldx, P, ldp, Q, ldq, H0, S(stream));
} LINNA_CATCH_INT
int linna_hmc_kick_drift(linna_ctx_t*, int B, int ndim, const float* mass, float ek, float ed, const float* G, int ldg,
                         float* P, int ldp, float* Q, int ldq, void* stream) try {
    return launch_hmc_kick_drift(B, ndim, mass, ek, ed, G, ldg, P, ldp, Q, ldq, S(stream));
} LINNA_CATCH_INT
int linna_hmc_accept(linna_ctx_t*, int B, int ndim, const float* mass, uint64_t seed, const int* step_dev, const float* H0,
                     const float* P, int ldp, const float* Qn, int ldq, const float* lnp_new, const float* Gn, int ldg,
                     const float* U, float* X, int ldx, float* lnp, float* G, int* naccept, void* stream) try {
    return launch_hmc_accept(B, ndim, mass, seed, step_dev, H0, P, ldp, Qn, ldq, lnp_new, Gn, ldg, U, X, ldx, lnp, G, naccept, S(stream));
} LINNA_CATCH_INT
int linna_step_increment(linna_ctx_t*, int* step_dev, void* stream) try { return launch_step_increment(step_dev, S(stream)); } LINNA_CATCH_INT

int linna_slice_init(linna_ctx_t*, const float* logp, const int* S_idx, int ns, const float* cc, int ldcc, const int* C_idx,
                     int nc, int ndim, const float* mu, uint64_t seed, const int* step_dev, int stream_id, float* DIR,
                     int ldd, float* Z0, float* L, float* R, int* flags, int maxsteps, void* stream) try {
    if (ns < 1 || nc < 2) { set_error("slice_init: need >= 2 complementary walkers"); return LINNA_ERR_INVALID; }
    if (maxsteps < 1) { set_error("slice_init: maxsteps %d < 1", maxsteps); return LINNA_ERR_INVALID; }
    return launch_slice_init(logp, S_idx, ns, cc, ldcc, C_idx, nc, ndim, mu, seed, step_dev, stream_id, DIR, ldd, Z0, L, R,
                             flags, maxsteps, S(stream));
} LINNA_CATCH_INT
int linna_slice_points(linna_ctx_t*, const float* coords, int ldc, int ndim, const int* S_idx, int ns, const float* DIR,
                       int ldd, const float* w, float* Q, int ldq, int nrep, void* stream) try {
    if (nrep < 1) { set_error("slice_points: nrep < 1"); return LINNA_ERR_INVALID; }
    return launch_slice_points(coords, ldc, ndim, S_idx, ns, DIR, ldd, w, Q, ldq, nrep, S(stream));
} LINNA_CATCH_INT
int linna_slice_expand(linna_ctx_t*, const float* Z0, const float* ZL, const float* ZR, float* L, float* R, int* flags,
                       int ns, int* counters, int slot, void* stream) try {
    return launch_slice_expand(Z0, ZL, ZR, L, R, flags, ns, counters, slot, S(stream));
} LINNA_CATCH_INT
int linna_slice_draw(linna_ctx_t*, const float* L, const float* R, const int* S_idx, float* W, const int* flags, int ns,
                     uint64_t seed, const int* step_dev, int stream_id, int round, int ntrial, void* stream) try {
    if (ntrial < 1) { set_error("slice_draw: ntrial < 1"); return LINNA_ERR_INVALID; }
    return launch_slice_draw(L, R, S_idx, W, flags, ns, seed, step_dev, stream_id, round, ntrial, S(stream));
} LINNA_CATCH_INT
int linna_slice_shrink(linna_ctx_t*, const float* Z0, const float* Zt, float* L, float* R, const float* W, int* flags,
                       float* Wacc, float* Zacc, int ns, int* counters, int slot, int ntrial, void* stream) try {
    if (ntrial < 1) { set_error("slice_shrink: ntrial < 1"); return LINNA_ERR_INVALID; }
    return launch_slice_shrink(Z0, Zt, L, R, W, flags, Wacc, Zacc, ns, counters, slot, ntrial, S(stream));
} LINNA_CATCH_INT
int linna_slice_commit(linna_ctx_t*, float* coords, int ldc, int ndim, float* logp, const int* S_idx, int ns,
                       const float* DIR, int ldd, const float* Wacc, const float* Zacc, void* stream) try {
    return launch_slice_commit(coords, ldc, ndim, logp, S_idx, ns, DIR, ldd, Wacc, Zacc, S(stream));
} LINNA_CATCH_INT

}  // extern "C"
