/* liblinna_hip.so -- C ABI of the MI355X-native LINNA emulator hot path.
 *
 * The reference (chto/linna) has no FFI: its hot path is Python calling torch ops.  This
 * header is the boundary a maintainer binds instead (ctypes stub in INTEGRATION.md); each
 * entry point names the reference call site it replaces (paths relative to the reference
 * root).
 *
 * Conventions
 *  - every function returns int: 0 = OK, <0 = error; linna_last_error() gives the text
 *    (thread-local).  No C++ exception crosses the boundary.
 *  - every pointer named like a matrix/vector is a CALLER-OWNED DEVICE pointer (fp32,
 *    row-major, leading dimension in elements).  Handles own small device-side state allocated when
 *    they are created or first used outside a stream capture (linna_net_t / linna_logprob_t: the weights
 *    re-laid in MFMA fragment order; linna_ctx_t: the descriptor table of the grouped parameter-gradient
 *    launch, 64 bytes of arrival counters); activations, workspaces, inputs and outputs are the caller's.
 *    Nothing synchronises (except linna_stream_sync and linna_event_elapsed_ms); all work is enqueued on
 *    `stream` (a hipStream_t passed as void*).
 *  - handles (linna_ctx_t, linna_net_t, ...) are small host objects; a handle is
 *    single-threaded, distinct handles are independent.
 *  - leading dimensions of activation/workspace matrices produced by the library are
 *    rounded up to a multiple of 4 floats (16-byte rows for vector loads).
 */
#ifndef LINNA_HIP_H
#define LINNA_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LINNA_ABI_VERSION 11   /* 11: every descriptor struct a caller fills (linna_gemm_t, linna_layer_t, linna_logprob_desc_t, linna_loss_desc_t) starts with `uint32_t struct_size` = its sizeof in the caller's header, checked by the entries that take it -- a binding built against another layout is refused instead of read at wrong offsets; linna_slice_init / linna_slice_half_step(maxsteps): zeus' stepping-out budget; the shrinking test is zeus' `Z0 < lnP`; 10: + linna_slice_fusion, linna_slice_half_step(expect_rows); 9: + linna_stretch_run, linna_chain_append_t, linna_acorr_*, linna_chain_meanstd, linna_val_metrics, linna_loss_desc_t::ylog; 8: + the exception barrier (LINNA_ERR_INTERNAL, linna_debug_raise) and linna_logprob_desc_t GREW by one pointer (Sfac, appended: a binding compiled against the v7 struct must be rebuilt -- linna_logprob_create copies the struct at the new size); 4: + linna_comm_* (RCCL); 5: + linna_net_prepare, linna_net_forward_loss; 6: + linna_net_adamw_step, linna_net_train_step, linna_net_train_step_update; 7: + linna_engine_rows, linna_slice_half_step, linna_program_describe, linna_net_train_launches, linna_logprob_grad_leapfrog, linna_hmc_start */

typedef struct linna_ctx linna_ctx_t;
typedef struct linna_net linna_net_t;
typedef struct linna_logprob linna_logprob_t;
typedef struct linna_graph linna_graph_t;

/* ------------------------------------------------------------------ runtime */
int linna_abi_version(void);
const char* linna_last_error(void);
/* Error codes: 0 OK; -1 invalid argument; -2 a HIP / RCCL call failed; -3 unsupported (the caller takes the documented
 * other route); -4 a C++ exception (std::bad_alloc, std::length_error ... from the host-side planners) was caught at
 * this boundary -- every entry below is a function-try-block, the text names the exception.  size_t-returning entries
 * return 0 in that case. */
#define LINNA_OK 0
#define LINNA_ERR_INVALID (-1)
#define LINNA_ERR_HIP (-2)
#define LINNA_ERR_UNSUPPORTED (-3)
#define LINNA_ERR_INTERNAL (-4)
/* diagnostic: throws inside a guarded entry -- 1 std::bad_alloc, 2 std::length_error, 3 a non-std exception,
 * 4 std::out_of_range, else nothing -- and returns what the barrier made of it (tests/test_abi.py) */
int linna_debug_raise(int kind);
int linna_ctx_create(int device, linna_ctx_t** out);
int linna_ctx_destroy(linna_ctx_t* ctx);
int linna_stream_sync(void* stream);
/* hipGraph capture of everything enqueued on `stream` between begin and end. */
int linna_graph_begin(void* stream);
int linna_graph_end(void* stream, linna_graph_t** out);
int linna_graph_launch(linna_graph_t* g, void* stream);
int linna_graph_destroy(linna_graph_t* g);
/* HIP-event timing helpers for bench.py (events recorded on the launch stream). */
int linna_event_create(void** ev);
int linna_event_record(void* ev, void* stream);
int linna_event_elapsed_ms(void* start, void* stop, float* ms);   /* synchronises on stop */
int linna_event_destroy(void* ev);

/* ------------------------------------------------------------------ collectives (RCCL over xGMI)
 * One communicator per context (= per device = per rank: one process per GPU).  Replaces, for the data path, what the
 * reference does with torch DistributedDataParallel and `lr = lr*size` (linna/predictor_gpu.py:246, 265-266) for the
 * training gradient, and with its MPI pool (linna/util.py:99-256, chtoPool) for walker / chain state.
 * Rank 0 calls linna_comm_unique_id and hands the LINNA_COMM_ID_BYTES bytes to every rank out of band (a file, a TCP
 * store, torch.distributed's store ...); every rank then calls linna_comm_init (collective, blocks until all ranks
 * arrive).  The collectives are enqueued on `stream` like every other entry and are hipGraph-capturable; all ranks must
 * issue them in the same order.  librccl.so.1 is resolved at run time: without it these entries fail with a text in
 * linna_last_error(), the rest of the library is unaffected. */
#define LINNA_COMM_ID_BYTES 128
int linna_comm_unique_id(void* id /* [LINNA_COMM_ID_BYTES] host bytes, out */);
int linna_comm_init(linna_ctx_t* ctx, int rank, int nranks, const void* id /* [LINNA_COMM_ID_BYTES] */);
int linna_comm_destroy(linna_ctx_t* ctx);            /* also done by linna_ctx_destroy */
/* nranks = 0 when the context holds no communicator; rccl_version as ncclGetVersion reports it (0 if unavailable) */
int linna_comm_info(linna_ctx_t* ctx, int* rank, int* nranks, int* rccl_version);
/* buf[n] <- sum over ranks, in place: the flat fp32 gradient buffer (+ the step's scalar loss) of a data-parallel
 * optimiser step (linna/predictor_gpu.py:265-266, 282-287) */
int linna_allreduce_sum_f32(linna_ctx_t* ctx, float* buf, size_t n, void* stream);
/* recv[r*n_per_rank .. (r+1)*n_per_rank) <- send[0 .. n_per_rank) of rank r: complementary walkers per half step,
 * chain blocks per flush (linna/util.py:143-156, 258-289; linna/sampler.py:346-368) */
int linna_allgather_f32(linna_ctx_t* ctx, const float* send, float* recv, size_t n_per_rank, void* stream);
/* buf[n] of rank `root` to every rank (learning rate of the range test, stop flag of the epoch controller:
 * linna/predictor_gpu.py:223-245, 387-393) */
int linna_broadcast_f32(linna_ctx_t* ctx, float* buf, size_t n, int root, void* stream);

/* ------------------------------------------------------------------ generic fused GEMM
 * C = epi( alpha0 * (A0.B0 + bias0) + (A1.B1 + bias1) ),  fp32 MFMA, exact fp32.
 * Layout codes: 0 = contraction index contiguous (A[M][K], B[N][K]);
 *               1 = k-major (A[K][M], B[K][N]).
 * epi(v): v += R; relu; v = mask>0 ? v : 0; v = v*cscale + cshift; [exp: v = exp(v)*cpost
 * + cshift2]; store C (if C != NULL); row-dot partial sums with `dotwith`. */
typedef struct {
    const float* A; const float* B;
    int lda, ldb, K;
    int alay, blay;
} linna_gemm_pair_t;

typedef struct {
    uint32_t struct_size;   /* = sizeof of this struct in the CALLER's header; checked by every entry that takes it (LINNA_ERR_INVALID otherwise) */
    linna_gemm_pair_t p[2];
    int npairs;
    int M, N;
    float* C; int ldc;
    const float* bias0; const float* bias1;
    float alpha0;
    const float* R; int ldr;
    int relu;
    const float* mask; int ldmask;
    const float* cscale; const float* cshift;
    int cexp; const float* cpost; const float* cshift2;
    const float* dotwith; int lddot;
    float* dot_partial; int dot_slots;
    int flags;   /* 0 in production; see LINNA_GEMM_* below (tile override / ablation for tools/) */
} linna_gemm_t;

#define LINNA_GEMM_TILE_MASK 0x7        /* 0 = automatic, 1.. = force tile configuration n-1 */
#define LINNA_GEMM_ABL_NOLOAD 0x10      /* timing only: skip global loads after the first K tile */
#define LINNA_GEMM_ABL_NOSTAGE 0x20     /* timing only: skip LDS restaging + barrier */
#define LINNA_GEMM_ABL_NOMFMA 0x40      /* timing only: skip the MFMAs */
#define LINNA_GEMM_NOSPLIT 0x80         /* timing only: no K split inside the workgroup for small grids */
#define LINNA_GEMM_DOT_SELF 0x100       /* the row-dot is taken with the product itself: sum_n C[m][n]^2 (dotwith only says "on") */

int linna_gemm_f32(linna_ctx_t* ctx, const linna_gemm_t* desc, void* stream);
int linna_gemm_dot_slots(int M, int N);

/* ------------------------------------------------------------------ emulator layers
 * linna_linear_fwd: Y = act(alpha*(X W^T + b) + R)          nn.Linear + F.relu, nn.py:121,125-130
 * linna_resblock_fwd: T = relu(X W1^T + b1);
 *                     Y = relu(0.1*(T W2^T + b2) + X Ws^T)  (Ws NULL: + X)   nn.py:53-54
 * linna_linear_bwd: dX = (dY W) [* (Xmask>0)], dW = dY^T X, db = colsum dY   autograd of the above,
 *                   predictor_gpu.py:285; any of dX/dW/db may be NULL.       */
int linna_linear_fwd(linna_ctx_t* ctx, const float* X, int ldx, const float* W, int ldw, const float* b,
                     float* Y, int ldy, int B, int K, int N, int relu, float alpha,
                     const float* R, int ldr, void* stream);
/* weights of the block use the packed convention of linna_layer_t (row stride LINNA_LD(K)) */
int linna_resblock_fwd(linna_ctx_t* ctx, const float* X, int ldx, const float* W1, const float* b1,
                       const float* W2, const float* b2, const float* Ws, float* T, int ldt,
                       float* Y, int ldy, int B, int K, int C, int N, void* stream);
int linna_linear_bwd(linna_ctx_t* ctx, const float* dY, int lddy, const float* X, int ldx,
                     const float* W, int ldw, float* dX, int lddx, const float* Xmask, int ldxm,
                     float* dW, int lddw, float* db, int B, int K, int N, float scale, void* stream);

/* ------------------------------------------------------------------ whole network
 * A network is an ordered list of ops (nn.py:110-133, 185-198, 351-374).  Parameter
 * pointers reference the caller's flat parameter buffer; gradient pointers (may be NULL)
 * reference the caller's flat gradient buffer with the same layout.
 * Packed weight convention: a weight matrix [N][K] is stored row-major with row stride
 * LINNA_LD(K) = K rounded up to a multiple of 4 floats (pad entries zero), 16-byte aligned,
 * so every row can be streamed by 16-byte LDS-DMA. */
#define LINNA_LD(k) (((k) + 3) & ~3)
#define LINNA_OP_LINEAR 0     /* Y = [relu](X W^T + b)                                         */
#define LINNA_OP_RESBLOCK 1   /* nn.py:11-56                                                    */
#define LINNA_OP_INSKIP 2     /* out += alpha*(X0 W^T + b), X0 = network input (nn.py:195)     */

typedef struct {
    uint32_t struct_size;   /* = sizeof of this struct in the CALLER's header; checked by every entry that takes it (LINNA_ERR_INVALID otherwise) */
    int op;
    int K, C, N;
    int relu;
    float alpha;
    const float *W, *b;                       /* linear / inskip */
    const float *W1, *b1, *W2, *b2, *Ws;      /* resblock; Ws NULL => identity skip */
    float *gW, *gb, *gW1, *gb1, *gW2, *gb2, *gWs;
} linna_layer_t;

/* Output-side column epilogue fused into the last GEMM (util.py:532-542, 457-458):
 *   v = v*cscale + cshift;  if cexp: v = exp(v)*cpost + cshift2.  Any pointer may be NULL. */
typedef struct {
    const float* cscale; const float* cshift;
    int cexp; const float* cpost; const float* cshift2;
} linna_colmap_t;

int linna_net_create(linna_ctx_t* ctx, const linna_layer_t* layers, int nlayers, int in_size,
                     linna_net_t** out);
int linna_net_destroy(linna_net_t* net);
/* Allocates every device-side copy the one-launch paths of this network use (fragment-order weight streams of the
 * forward; with `backward` those of the dX chain, `input_grad`: down to the network input) and the context's auxiliary
 * stream -- so that nothing is allocated on the launch path and the FIRST step can be captured into a hipGraph.
 * Without this call the copies are allocated on first use outside a capture. */
int linna_net_prepare(linna_net_t* net, int backward, int input_grad);
/* bytes of activation workspace needed for a batch of B rows (forward, all activations
 * kept) and for the backward scratch. */
size_t linna_net_fwd_ws_bytes(const linna_net_t* net, int B);
size_t linna_net_bwd_ws_bytes(const linna_net_t* net, int B);
/* forward: X[B][ldx] -> OUT[B][ldo] (network output after `outmap`, NULL = identity).
 * `ws` keeps every intermediate activation for a following backward.  Predictor.predict,
 * predictor_gpu.py:461-504; training forward, predictor_gpu.py:278. */
int linna_net_forward(linna_net_t* net, const float* X, int ldx, int B, void* ws, float* OUT, int ldo,
                      const linna_colmap_t* outmap, void* stream);
/* backward from dOUT[B][lddo] (gradient wrt the raw network output): parameter gradients
 * into the layers' g* pointers when param_grads != 0, input gradient into dX (may be
 * NULL).  torch autograd at predictor_gpu.py:285 / HMCSampler.py:32. */
int linna_net_backward(linna_net_t* net, const float* X, int ldx, int B, void* fwd_ws, void* bwd_ws,
                       const float* dOUT, int lddo, float* dX, int lddx, int param_grads, void* stream);
/* Diagnostics: which parts of this network's forward / backward ran (or will run) as ONE launch of the
 * whole-network kernel instead of one GEMM per op.  Each flag is -1 (not decided yet: decided at the first
 * call of that kind), 0 or 1.  fwd: linna_net_forward; dx: the dX chain of linna_net_backward with dX == NULL
 * (training); dx_input: with dX != NULL (gradient with respect to the network input).  Any pointer may be NULL. */
/* Launches of one optimiser step through linna_net_train_step_update for a batch of B rows, once linna_net_prepare_loss
 * has seen the loss: 2 = forward + loss + dX chain in ONE launch of the whole-network kernel (the 4-row engine:
 * B <= 1024) and the grouped parameter-gradient launch with the optimiser in its epilogue; 3 = forward + loss and dX chain
 * as two launches; 0 = the network or its loss trains layer by layer. */
int linna_net_train_launches(const linna_net_t* net, int B);
int linna_net_stream_state(const linna_net_t* net, int* fwd, int* dx, int* dx_input);

/* ------------------------------------------------------------------ prior map + input transform
 * util.py:339-347 (Transform) fused with util.py:483-497 (X_transform_class):
 *   theta_j = flat ? 0.5*(1+erf(z_j/sqrt2))*width_j + a1_j : z_j*a2_j + a1_j
 *   x_j = ((log10_j ? log10(theta_j) : theta_j) - xmean_j) / xstd_j
 * is_flat/log10 flags are int32 device arrays of length nin; THETA may be NULL. */
int linna_prior_map_fwd(linna_ctx_t* ctx, const float* Z, int ldz, int B, int nin, const int* is_flat,
                        const float* a1, const float* a2, const int* log10_flag, const float* xmean,
                        const float* xstd, float* X, int ldx, float* THETA, int ldt, void* stream);
/* chain rule back to z and the prior term: dz_j = dx_j/xstd_j * [1/(theta_j ln10)] *
 * dtheta_j/dz_j - z_j. */
int linna_prior_map_bwd(linna_ctx_t* ctx, const float* Z, int ldz, int B, int nin, const int* is_flat,
                        const float* a1, const float* a2, const int* log10_flag, const float* xstd,
                        const float* dX, int lddx, float* dZ, int lddz, void* stream);

/* ------------------------------------------------------------------ Gaussian log-likelihood
 * util.py:953-955 + :1013-1016 + :1160-1165, row-wise for a batch of walkers:
 *   out_b = (-0.5 * d_b S d_b^T)/T - 0.5*|z_b|^2 ; NaN -> -inf,   d = D row (model - data)
 * diag: S = diag(w) (coalesced row read + wavefront shuffle reduction, HBM-bound);
 * dense: MFMA GEMM D.S fused with the row-dot, then a finishing reduction.
 * `scratch` for dense: B * linna_gemm_dot_slots(B, nout) floats. */
int linna_gauss_loglike_diag(linna_ctx_t* ctx, const float* D, int ldd, int B, int nout, const float* w,
                             const float* Z, int ldz, int nin, float temperature, float* out,
                             void* stream);
int linna_gauss_loglike_dense(linna_ctx_t* ctx, const float* D, int ldd, int B, int nout,
                              const float* S, int lds, const float* Z, int ldz, int nin,
                              float temperature, float* scratch, float* out, void* stream);

/* ------------------------------------------------------------------ full serving pipeline
 * One call = Log_prob.__call__ (util.py:990-1021) for B walkers: prior map -> X transform
 * -> network -> Y transform -> *sigma -> log-likelihood/T + ln prior.  The struct holds
 * device pointers to constants (all caller-owned). */
typedef struct {
    uint32_t struct_size;   /* = sizeof of this struct in the CALLER's header; checked by every entry that takes it (LINNA_ERR_INVALID otherwise) */
    int nin, nout;
    const int* is_flat; const float* a1; const float* a2;       /* priors, [nin] */
    const int* log10_flag; const float* xmean; const float* xstd;
    linna_colmap_t outmap;        /* maps raw network output to d = model - data */
    const float* S; int lds;      /* dense inverse covariance [nout][nout] (or NULL) */
    const float* Ssym;            /* 0.5*(S+S^T) for the gradient (dense) */
    const float* w;               /* diagonal of S when S is diagonal (or NULL) */
    const float* gscale;          /* [nout] d(d)/d(raw output) = y_std*sigma, for the gradient */
    float temperature;
    const float* Sfac;            /* optional, dense only: lower-triangular L [nout][lds] with S = L L^T (float64 Cholesky of S on
                                   * the host, rounded to fp32).  lnP is then taken as |d L|^2 instead of d S d^T: the same one
                                   * GEMM, but a sum of squares -- no cancellation between the stiff and the soft directions of an
                                   * ill-conditioned covariance (error ~ sqrt(cond) eps instead of ~ cond eps).
                                   * L MUST be lower triangular (zeros above the diagonal, as numpy.linalg.cholesky returns it): for
                                   * nout > 512 the second column pass starts at row 512 -- the block above it is never read.
                                   * NULL: the direct form.  The gradient keeps S (Ssym). */
} linna_logprob_desc_t;

int linna_logprob_create(linna_ctx_t* ctx, linna_net_t* net, const linna_logprob_desc_t* desc,
                         linna_logprob_t** out);
int linna_logprob_destroy(linna_logprob_t* lp);
/* Networks of LINEAR / RESBLOCK ops up to 1024 wide (every model of nn.py) are served by ONE
 * whole-network kernel from a fragment-order copy of the weights owned by the linna_logprob_t
 * (net_stream.hip): 16 walker rows per workgroup for batches that fill the GPU, 8 or 4 rows per
 * workgroup below 2048 / 1024 walkers; with a dense S the output map is folded into the copy's last
 * layer and S is the program's last segment, so lnP still comes out of the one launch.
 * The copy is refreshed automatically
 * after linna_adamw_step and after any linna_graph_launch; a caller that overwrites parameter
 * memory by other means (hipMemcpy of a checkpoint, a torch-side copy_) calls this once
 * afterwards -- the reference has no counterpart because `model.load_state_dict`
 * (predictor_gpu.py:439-445) rebinds the tensors the forward pass reads. */
int linna_weights_changed(linna_ctx_t* ctx);
/* Rows per workgroup of the whole-network kernel: 0 (default) = chosen per launch from the batch size -- the fewest
 * of 4 / 8 / 16 that still fit the batch into one workgroup per CU; 4, 8 or 16 = that engine for every launch of the
 * process (tests and measurements; results differ between engines in the last bits: another summation order over k).
 * Returns the previous setting, or LINNA_ERR_INVALID.  Process-wide, not a per-launch argument: the launch path reads
 * one atomic instead of the environment. */
int linna_engine_rows(int rows);
/* How the whole-network kernel's dense log-likelihood segment (chi^2 = |d L|^2, L the lower-triangular Cholesky factor of the
 * inverse covariance; util.py:953-955) skips the factor's zero upper triangle, for log-probability objects created AFTERWARDS:
 * 0 not at all, 1 the second column pass starts at row 512, 2 (default; LINNA_DENSE_TRI) additionally, for 960 < nout <= 1024,
 * the balanced assignment: wave w takes the 64-column blocks w and 15 - w, each from its first non-zero row.  Same sums in
 * the same order in every mode (the skipped products are zeros).  -1 queries.  Returns the previous mode (tests, A/B). */
int linna_dense_tri(int mode);
/* Which launches of linna_slice_half_step are folded into their neighbours, as a mask (default: all that apply): bit 0 -- with ONE
 * stepping-out round, its logic kernel is not launched: the first shrinking round's evaluation derives its trial points from
 * the bracket ends' lnP in its prologue and that round's logic kernel does the bookkeeping of both; bit 1 -- the set-up of the
 * half step (directions, slice heights, initial brackets, the usage counters' roll) is not a launch of its own: the first
 * evaluation does it in its prologue, each row for its walker, the rows of bracket end 0 writing it for the later launches;
 * bit 2 -- reserved.  The chain is identical under every mask (same Philox counters, same arithmetic).  -1 queries.
 * Returns the previous mask (tests, A/B). */
int linna_slice_fusion(int mask);
/* The serving program the whole-network kernel would run for this op list on the engine of `rows` rows per workgroup
 * (dense_nout > 0: with a dense inverse covariance of that size as its last segment, in the direct form d S d^T; dense_nout < -1:
 * of -dense_nout columns in the factored form |d L|^2 under the current linna_dense_tri mode; dense_nout == -1: the program of
 * the one-launch gradient instead, forward segments then the dX chain down to the input), as text: a header line, then one
 * line per segment ("WIDE|SPLIT|SIDE steps passes ncg kc dst zext N").  Host-side planning only -- nothing is launched,
 * no pointer is read -- for tests and diagnostics.  Returns the number of segments, 0 when the network is outside the
 * kernel's reach. */
int linna_program_describe(const linna_layer_t* layers, int nlayers, int in_size, int rows, int dense_nout, char* buf,
                           size_t n);
size_t linna_logprob_ws_bytes(const linna_logprob_t* lp, int B, int with_grad);
/* lnP[B]; THETA[B][ldt] optional (physical parameters, for chain_transformed). */
int linna_logprob_eval(linna_logprob_t* lp, const float* Z, int ldz, int B, void* ws, float* lnP,
                       float* THETA, int ldt, void* stream);
/* The same, gated on a device int: when gate[0] == 0 at the time the kernel starts, the whole-network
 * kernel computes nothing and the contents of lnP are unspecified (the layer-by-layer path ignores
 * the gate and computes).  For
 * host loops that queue the next round of a data-dependent iteration before they know whether it is
 * needed (the ensemble slice sampler's stepping-out / shrinking rounds). */
int linna_logprob_eval_if(linna_logprob_t* lp, const float* Z, int ldz, int B, void* ws, float* lnP,
                          float* THETA, int ldt, const int* gate, void* stream);
/* lnP at the ensemble slice sampler's trial points  coords[S[k]] + w[j*ns + k] * DIR[k]  (j < nrep), in ONE
 * launch and without writing the points to memory (= linna_slice_points + linna_logprob_eval_if).  lnP[nrep*ns].
 * LINNA_ERR_UNSUPPORTED under the conditions of linna_stretch_half_step: the caller then uses the two entries. */
int linna_logprob_eval_slice_points(linna_logprob_t* lp, const float* coords, int ldc, int ndim, const int* S_idx,
                                    int ns, const float* DIR, int ldd, const float* w, int nrep, float* lnP,
                                    const int* gate, void* stream);
/* lnP[B] and d lnP / d z [B][ldg]  (intended semantics of util.py:1023-1035; HMCSampler.py:32). */
int linna_logprob_grad(linna_logprob_t* lp, const float* Z, int ldz, int B, void* ws, float* lnP,
                       float* G, int ldg, void* stream);
/* One leapfrog step's gradient, kick and drift (HMCSampler.py:35-49: p += eps dlnp(q); q += eps p / m): lnP and G at Q, then
 * P += eps_kick * G and Q += eps_drift * P / mass (= linna_logprob_grad + linna_hmc_kick_drift, same arithmetic).  ONE launch
 * where linna_logprob_grad is one: the kick and the drift ride in the finish of the whole-network kernel. */
int linna_logprob_grad_leapfrog(linna_logprob_t* lp, float* Q, int ldq, int B, void* ws, float* lnP, float* G, int ldg,
                                float* P, int ldp, const float* mass, float eps_kick, float eps_drift, void* stream);

/* ------------------------------------------------------------------ training
 * chi^2-ratio loss of util.py:1070-1088,1114-1116 on the raw network output PRED:
 *   delta = mask ? 0 : ((y/sigma - ymean)/ystd - pred)  [ylog: log(y/sigma) for y/sigma];  chi2 = delta Cinv delta^T
 *   loss_b = chi2 / den_b ; L = inv_batch * sum_b loss_b ;  dPRED = -2 (delta Cinv) inv_batch/den_b
 *   (Cinv symmetric; inv_batch = 1/global batch so data-parallel shards sum to the mean)
 * den_b = max(chisqMd_b, nout/2) is precomputed per dataset row (linna_chi2_md).
 * ROWS (int32, may be NULL = identity) selects minibatch rows out of the resident dataset. */
typedef struct {
    uint32_t struct_size;   /* = sizeof of this struct in the CALLER's header; checked by every entry that takes it (LINNA_ERR_INVALID otherwise) */
    int nout;
    const float* sigma; const float* ymean; const float* ystd;   /* [nout] */
    const float* data_norm;                                       /* [nout] */
    const float* Cinv; int ldc;                                   /* [nout][nout], symmetric */
    int ylog;   /* ABI 9 (fills what was tail padding: the struct's size is unchanged): 1 = `ypositive` training, the target is
                 * normalised as (log(y/sigma) - ymean)/ystd (util.py:567-571, 1444-1447); ymean / ystd / data_norm / Cinv are
                 * then the log-space constants the caller computed (util.py:1444-1447, 573-586) */
} linna_loss_desc_t;

/* floats of scratch the three loss entry points need for a batch of B rows, in bytes */
size_t linna_loss_scratch_bytes(int B, int nout);
int linna_chi2_md(linna_ctx_t* ctx, const linna_loss_desc_t* d, const float* Y, int ldy, int nrows,
                  float* scratch, float* den, void* stream);
int linna_chi2_ratio_loss_fwd_bwd(linna_ctx_t* ctx, const linna_loss_desc_t* d, const float* PRED,
                                  int ldp, const float* Y, int ldy, const float* den, const int* ROWS,
                                  int B, float* scratch, float* loss_rows, float* loss_mean,
                                  float* dPRED, int lddp, float inv_batch, void* stream);
/* Minibatch gather + X transform + network forward (activations kept for linna_net_backward) + the loss above, its
 * per-row values, batch mean and gradient -- predictor_gpu.py:274-285 up to loss.backward() -- in ONE launch of the
 * whole-network kernel when the network and the loss fit it (LINNA_ERR_UNSUPPORTED otherwise: run linna_gather_xform,
 * linna_net_forward and linna_chi2_ratio_loss_fwd_bwd instead).  X[n][ldx]: the resident, untransformed training
 * inputs; XB[B][ldxb]: the transformed batch (out; the first layer's parameter gradient reads it); ws: forward
 * workspace of linna_net_fwd_ws_bytes(net, B); PRED[B][ldp]: raw network output (out); YN[n][ldyn]: the normalised
 * targets of the resident set (linna_loss_targets). */
int linna_net_prepare_loss(linna_net_t* net, const linna_loss_desc_t* d);   /* allocates its weight stream (outside a capture) */
/* YN[i][j] = (Y[i][j]/sigma_j - ymean_j)/ystd_j, NaN where the element is masked (util.py:1072): the normalised targets
 * of a whole data set, computed once; linna_net_forward_loss subtracts its prediction from them */
int linna_loss_targets(linna_ctx_t* ctx, const linna_loss_desc_t* d, const float* Y, int ldy, int nrows, float* YN, int ldyn,
                       void* stream);
int linna_net_forward_loss(linna_net_t* net, const linna_loss_desc_t* d, const float* X, int ldx, const int* ROWS, int B,
                           const int* log10_flag, const float* xmean, const float* xstd, float* XB, int ldxb, void* ws,
                           float* PRED, int ldp, const float* YN, int ldyn, const float* den, float inv_batch,
                           float* loss_rows, float* loss_mean, float* dPRED, int lddp,
                           /* optional (NULL, NULL, 0, 0): the AdamW state of the update that follows this step's backward --
                            * its step counter and bias corrections are then advanced here, in the launch that takes the
                            * batch mean, and linna_adamw_step is called with prepared = 1 */
                           float* hyper, int* step_dev, float beta1, float beta2, void* stream);
/* The optimiser step up to its gradients in ONE call (predictor_gpu.py:274-285): linna_net_forward_loss, then
 * linna_net_backward(param_grads = 1) on the gathered rows XB and the stored activations (`fwd_ws`, scratch `bwd_ws` of
 * linna_net_bwd_ws_bytes).  The batch mean of the loss and the AdamW step constants ride in the backward's dX-chain launch
 * as one extra workgroup (no launch of their own); follow with linna_net_adamw_step / linna_adamw_step(prepared = 1).
 * LINNA_ERR_UNSUPPORTED exactly when linna_net_forward_loss is. */
int linna_net_train_step(linna_net_t* net, const linna_loss_desc_t* d, const float* X, int ldx, const int* ROWS, int B,
                         const int* log10_flag, const float* xmean, const float* xstd, float* XB, int ldxb, void* fwd_ws,
                         float* PRED, int ldp, const float* YN, int ldyn, const float* den, float inv_batch,
                         float* loss_rows, float* loss_mean, float* dPRED, int lddp, void* bwd_ws,
                         float* hyper, int* step_dev, float beta1, float beta2, void* stream);
/* validation pieces (util.py:1124-1127): per-row loss and chisq_nnd/chisq_Md. */
int linna_val_rows(linna_ctx_t* ctx, const linna_loss_desc_t* d, const float* PRED, int ldp,
                   const float* Y, int ldy, const float* den, int B, float* scratch, float* loss_rows,
                   float* frac_rows, void* stream);
/* The epoch's record for the host-side controller (predictor_gpu.py:289-312, 339-401) in ONE 4-float device array instead of
 * two row vectors: out = { *last_train_loss (NaN if NULL), lower median of loss_rows[n] (torch.median), max of frac_rows[n],
 * lower median of frac_rows[n] } -- Val_metric_fn's three numbers (util.py:1124-1127) behind linna_val_rows; a NaN anywhere makes
 * the statistic NaN, as torch's reductions do.  n <= 65536. */
int linna_val_metrics(linna_ctx_t* ctx, const float* loss_rows, const float* frac_rows, int n, const float* last_train_loss,
                      float* out, void* stream);
/* gather + input transform of a minibatch: XB[i] = (X[rows[i]] (log10 on flagged cols) - mean)/std */
int linna_gather_xform(linna_ctx_t* ctx, const float* X, int ldx, const int* ROWS, int B, int nin,
                       const int* log10_flag, const float* xmean, const float* xstd, float* XB, int ldxb,
                       void* stream);
/* torch.optim.AdamW step on a flat parameter vector (predictor_gpu.py:267,287).
 * `hyper` is a 4-float DEVICE array [lr, weight_decay, bc1, sqrt(bc2)]: the caller writes
 * lr/weight_decay (so a captured graph replays with new values), the library increments the
 * device int32 `step_dev` and refreshes the two bias-correction slots before the update. */
int linna_adamw_step(linna_ctx_t* ctx, float* p, const float* g, float* m, float* v, size_t n,
                     float* hyper, int* step_dev, float beta1, float beta2, float eps,
                     int prepared /* 1: linna_net_forward_loss already advanced step_dev / hyper[2..3] for this step */,
                     void* stream);
/* The same update over the flat buffer `p[n]` that holds exactly `net`'s tensors back to back (weights with rows padded
 * to 4 floats, biases padded to 4), AND the re-layout of the updated weights into the two fragment-order weight streams
 * a training step reads (linna_net_forward_loss's, the backward's dX chain) -- one launch instead of the update plus the
 * two lazy re-layouts of the next step.  `B`: the batch size the steps run at (selects the engine, hence the stream
 * layout).  LINNA_ERR_UNSUPPORTED when the network does not train through those two streams or the buffer is laid out
 * differently: use linna_adamw_step then (the streams re-lay themselves).  (predictor_gpu.py:287 `optim.step()`.) */
int linna_net_adamw_step(linna_net_t* net, int B, float* p, const float* g, float* m, float* v, size_t n,
                         float* hyper, int* step_dev, float beta1, float beta2, float eps, int prepared, void* stream);
/* ONE optimiser step (predictor_gpu.py:274-287: forward, loss, backward, optim.step()) in ONE call and THREE launches:
 * linna_net_train_step with AdamW in the epilogue of its grouped parameter-gradient launch -- every gradient tile updates
 * its block of `params` / `m` / `v` (flat buffers laid out like the gradient buffer the layer table points into) and writes
 * the updated block into both weight streams of the next step.  One rank only: data-parallel training all-reduces the
 * gradients between backward and update (linna_net_train_step + linna_net_adamw_step).  LINNA_ERR_UNSUPPORTED, before
 * anything is launched, when linna_net_train_step / linna_net_adamw_step would be, or some parameter gradient of this
 * network does not fit the grouped launch. */
int linna_net_train_step_update(linna_net_t* net, const linna_loss_desc_t* d, const float* X, int ldx, const int* ROWS, int B,
                                const int* log10_flag, const float* xmean, const float* xstd, float* XB, int ldxb,
                                void* fwd_ws, float* PRED, int ldp, const float* YN, int ldyn, const float* den,
                                float inv_batch, float* loss_rows, float* loss_mean, float* dPRED, int lddp, void* bwd_ws,
                                float* params, float* m, float* v, size_t n, float* hyper, int* step_dev, float beta1,
                                float beta2, float eps, void* stream);

/* ------------------------------------------------------------------ ensemble / HMC moves
 * Stretch move (emcee StretchMove, called at sampler.py:493-495,530): for the active half
 *   zz = ((a-1)u+1)^2/a ; q = c[r] - (c[r]-s) zz ; factor = (ndim-1) log zz
 * with Philox4x32-10 draws keyed (seed; walker, step, stream).  `S_idx` lists the active rows of
 * `coords`, `C_idx` the complementary rows of `ccoords` (the same array on one GPU; the
 * all-gathered complement of every rank when walkers are sharded). */
int linna_stretch_propose(linna_ctx_t* ctx, const float* coords, int ldc, int ndim, const int* S_idx,
                          int ns, const float* ccoords, int ldcc, const int* C_idx, int nc, uint64_t seed,
                          const int* step_dev, int stream_id, float a, float* Q, int ldq, float* factors,
                          void* stream);
int linna_stretch_accept(linna_ctx_t* ctx, float* coords, int ldc, int ndim, float* logp,
                         const int* S_idx, int ns, const float* Q, int ldq, const float* logp_new,
                         const float* factors, uint64_t seed, const int* step_dev, int stream_id,
                         int* naccept, void* stream);
/* One ensemble half step in ONE launch: propose + log-probability of the proposals + accept, fused
 * around the whole-network kernel (net_stream.hip).  Same Philox counters and arithmetic as the
 * three entries above: bit-identical results.  The Philox step is step_dev[0] + step_offset, so a
 * host loop can count iterations itself and skip linna_step_increment (one launch less per
 * iteration); a captured graph passes step_offset = 0 and increments the device counter.  Returns LINNA_ERR_UNSUPPORTED (and launches
 * nothing) when the log-probability object does not run the whole-network kernel (dense inverse
 * covariance, ypositive output, more than 64 parameters, a network outside net_stream's reach):
 * the caller then uses linna_stretch_propose / linna_logprob_eval / linna_stretch_accept. */
int linna_stretch_half_step(linna_logprob_t* lp, float* coords, int ldc, int ndim, float* logp,
                            const int* S_idx, int ns, const float* ccoords, int ldcc, const int* C_idx,
                            int nc, uint64_t seed, const int* step_dev, int step_offset, int stream_id,
                            float a, int* naccept, void* stream);
/* leapfrog pieces for batched per-walker HMC (HMCSampler.py:26-54, sampler.py:67-98). */
/* P0 (standard-normal draws [B][ldp0]) and U (uniforms [B]) are optional: NULL = Philox draws. */
int linna_hmc_init(linna_ctx_t* ctx, int B, int ndim, const float* mass, uint64_t seed,
                   const int* step_dev, const float* lnp, const float* P0, int ldp0, float* P, int ldp,
                   float* H0, void* stream);
/* linna_hmc_init, the first half kick and the first drift in one launch (HMCSampler.py:26-37): P ~ N(0, m),
 * H0 = P^2 / 2m - lnp, P += eps_kick * G, Q = X + eps_drift * P / m. */
int linna_hmc_start(linna_ctx_t* ctx, int B, int ndim, const float* mass, uint64_t seed, const int* step_dev,
                    const float* lnp, const float* P0, int ldp0, const float* G, int ldg, float eps_kick, float eps_drift,
                    const float* X, int ldx, float* P, int ldp, float* Q, int ldq, float* H0, void* stream);
int linna_hmc_kick_drift(linna_ctx_t* ctx, int B, int ndim, const float* mass, float eps_kick,
                         float eps_drift, const float* G, int ldg, float* P, int ldp, float* Q, int ldq,
                         void* stream);
int linna_hmc_accept(linna_ctx_t* ctx, int B, int ndim, const float* mass, uint64_t seed,
                     const int* step_dev, const float* H0, const float* P, int ldp, const float* Qnew,
                     int ldq, const float* lnp_new, const float* Gnew, int ldg, const float* U, float* X,
                     int ldx, float* lnp, float* G, int* naccept, void* stream);
int linna_step_increment(linna_ctx_t* ctx, int* step_dev, void* stream);

/* Ensemble slice sampling (zeus DifferentialMove behind sampler.py:728-735): per active walker a
 * direction mu*(c_a - c_b) from two distinct complementary walkers, a slice height
 * Z0 = logp + log u, a unit bracket [L, R] placed at random around 0; stepping out pushes an end
 * out while the density there exceeds Z0; shrinking draws w ~ U(L, R) and pulls the bracket in to
 * rejected trials (accepted iff Z0 < lnP there, zeus' comparison).  `mu_dev[0]` is the whole scale of the direction: a caller
 * with zeus' semantics of mu (direction 2 mu (c_a - c_b), moves.py) passes 2 mu.  `maxsteps` is zeus' stepping-out budget
 * (its default: 10000): at most J = floor(maxsteps u) steps to the left and maxsteps - 1 - J to the right (Philox stream
 * `stream_id`, sub-counter 1).  flags[3k..3k+2] = {steps left of J while the left end is still stepping out (0: closed), the
 * same for the right end and K, still shrinking}; counters[0..2] = {expansions, contractions, walkers still active after
 * this call} (device int32, caller zeroes). */
int linna_slice_init(linna_ctx_t* ctx, const float* logp, const int* S_idx, int ns, const float* ccoords, int ldcc,
                     const int* C_idx, int nc, int ndim, const float* mu_dev, uint64_t seed, const int* step_dev,
                     int stream_id, float* DIR, int ldd, float* Z0, float* L, float* R, int* flags, int maxsteps,
                     void* stream);
/* Q[j*ns + k] = coords[S[k]] + w[j*ns + k] * DIR[k], j < nrep */
int linna_slice_points(linna_ctx_t* ctx, const float* coords, int ldc, int ndim, const int* S_idx, int ns,
                       const float* DIR, int ldd, const float* w, float* Q, int ldq, int nrep, void* stream);
/* counters: [0] expansions, [1] contractions (zeus' mu tuning), [slot] walkers still active after this call */
int linna_slice_expand(linna_ctx_t* ctx, const float* Z0, const float* ZL, const float* ZR, float* L, float* R,
                       int* flags, int ns, int* counters, int slot, void* stream);
/* `ntrial` shrink trials per call, placed as the sequential procedure would place them if every
 * earlier one were rejected (the bracket after a rejection depends on where the trial fell, not on
 * its density): W[j*ns + k], Ztrial[j*ns + k]; Philox sub-counters round+1 .. round+ntrial.  Two
 * trials per round fill the GPU (2 x nw/2 points per launch) and halve the number of rounds; the
 * accepted point is the one the one-trial-per-round procedure accepts. */
int linna_slice_draw(linna_ctx_t* ctx, const float* L, const float* R, const int* S_idx, float* W, const int* flags,
                     int ns, uint64_t seed, const int* step_dev, int stream_id, int round, int ntrial, void* stream);
int linna_slice_shrink(linna_ctx_t* ctx, const float* Z0, const float* Ztrial, float* L, float* R, const float* W,
                       int* flags, float* Wacc, float* Zacc, int ns, int* counters, int slot, int ntrial,
                       void* stream);
int linna_slice_commit(linna_ctx_t* ctx, float* coords, int ldc, int ndim, float* logp, const int* S_idx, int ns,
                       const float* DIR, int ldd, const float* Wacc, const float* Zacc, void* stream);
/* One half step of the ensemble slice sampler in ONE call (replaces the round-by-round use of the six entries above for
 * log-probabilities the whole-network kernel serves; LINNA_ERR_UNSUPPORTED otherwise): directions + slice heights
 * (linna_slice_init's arithmetic and Philox counters: stream `half`), `nexp_rounds` stepping-out rounds, round r evaluating
 * the `m_sched[r]` bracket ends per side the sequential loop `while lnP(L) > Z0: L -= 1` would visit next, `nshr_rounds`
 * shrinking rounds of `nt_sched[r]` trials each placed as if its predecessors were rejected (linna_slice_draw's rule, stream
 * 2 + half, sub-counter = trials of the earlier rounds + j + 1), and the commit (in the last shrinking round's kernel;
 * bump_step != 0: step_dev[0] += 1 there as well, which saves the caller its linna_step_increment behind the second half
 * step of an iteration): at most 1 + 2 (nexp_rounds + nshr_rounds) launches on `stream` (linna_slice_fusion folds the set-up
 * and, with one stepping-out round, that round's logic into the evaluations: 5 launches for 1 + 2 rounds), no host
 * synchronisation.  The accepted points are those of the one-point-per-round procedure.
 * Rounds after the first evaluate only the walkers still active (listed and counted on the device), so a schedule that
 * looks further ahead each round (1, 2, 4, 8 ...) costs little and keeps the number of rounds small; rounds behind the one
 * that finished the last walker leave at once.  m_sched / nt_sched: HOST arrays; M = max m_sched, T = max nt_sched:
 *   state[5 ns]  : Z0 | L | R | Wacc | Zacc;  flags[3 ns];  W[2 M ns], Wd[T ns], Zt[max(2 M, T) ns],
 *   list[max(2 M, T) ns] (the trial points of the walkers still active, for the rounds after the first): scratch
 *   counters[5 + 2 (nexp_rounds + nshr_rounds)]: [0] expansions, [1] contractions (zeroed first when zero_totals),
 *       [2] walkers the rounds of a call left unfinished -- they keep their position; the caller treats a non-zero
 *       count as zeus treats its `maxsteps` (an error) -- [3] evaluated points (both cumulative), [4 + r] walkers still
 *       active after round r of THIS call (stepping-out rounds first), [4 + nr + r] the same summed over the earlier calls,
 *       [4 + 2 nr] the number of calls (nr = nexp_rounds + nshr_rounds): how much of the look-ahead a run uses.
 * expect_rows (HOST array of nr ints, or null): the number of trial points the caller expects round r to evaluate (entry r for
 *   the stepping-out rounds, nexp_rounds + r for the shrinking ones; the first round of each kind evaluates every walker and
 *   its entry is ignored) -- the evaluation of a later round is sized for every walker but runs the ENGINE (4 / 8 / 16 rows per
 *   workgroup) that suits this number: 60 points on the 16-row engine take 56 us, on the 4-row engine 29.  null: a quarter of
 *   the previous round's.  A wrong expectation costs time, never results of another chain... the lnP of one point differs in
 *   the last bits between engines (another summation order), as everywhere else in this library.
 * maxsteps: zeus' stepping-out budget, as in linna_slice_init.
 * zeus' EnsembleSampler behind sampler.py:728-735. */
int linna_slice_half_step(linna_logprob_t* lp, float* coords, int ldc, int ndim, float* logp, const int* S_idx, int ns,
                          const float* ccoords, int ldcc, const int* C_idx, int nc, const float* mu, uint64_t seed,
                          int* step_dev, int half, const int* m_sched, int nexp_rounds, const int* nt_sched,
                          int nshr_rounds, float* DIR, int ldd, float* state, int* flags, float* W, float* Wd, float* Zt,
                          int* list, int* counters, int zero_totals, int bump_step, const int* expect_rows, int maxsteps,
                          void* stream);

/* `nsteps` ensemble iterations in ONE call: 2 nsteps launches of linna_stretch_half_step's kernel with the same Philox
 * counters (step = step_dev[0] + step_offset + i, stream = half), so the chain is bit-identical to a host loop over that
 * entry.  Iteration i moves the walkers splits[i * split_stride + 0 .. ns) against splits[i * split_stride + ns .. 2 ns) and
 * then the other way round (emcee's RedBlueMove draws a random equal split per iteration; split_stride = 0 reuses one split;
 * DEVICE int32, ns = nw / 2).  chain != NULL: the kernels' finish also writes each walker's position and log-probability
 * after iteration i to chain[i][nw][ndim] / logps[i][nw] (every walker moves in exactly one of the two half steps), which
 * replaces two device copies per iteration; the whole ensemble is on this rank (no complementary exchange).  Replaces the
 * per-walker loop of emcee's sample() behind linna/sampler.py:530.  LINNA_ERR_UNSUPPORTED as linna_stretch_half_step. */
int linna_stretch_run(linna_logprob_t* lp, float* coords, int ldc, int ndim, float* logp, int nw, const int* splits,
                      int split_stride, int nsteps, uint64_t seed, const int* step_dev, int step_offset, float a,
                      int* naccept, float* chain, float* logps, void* stream);

/* ------------------------------------------------------------------ convergence statistics of a chain (autocorr.hip)
 * The reference recomputes emcee's integrated autocorrelation time of the WHOLE chain every 100 iterations
 * (linna/sampler.py:532-552 `get_autocorr_time(tol=0)`; zeus: :667-696, first 20 % discarded) and compares the halves of the
 * chain's tail (`checkmeanstd`, :370-387).  Here the statistics' copy of the chain is CT[row][ndim][nwp] (fp32; time x
 * parameter x walker lane, nwp = walkers rounded up to a multiple of 64, padding zero; every wstride-th walker -- the subset
 * a routine check may average over -- takes the first ceil(nw / wstride) lanes, the others follow in order) and the
 * estimator runs on RUNNING sums in float64 (the only double-precision pointers of this ABI) over the first nwc lanes of
 * every parameter (nwc a multiple of 64, <= nwp; nlive of them are walkers):
 *   Ssum[k][ndim][nwc] = sum_{t = lo+k}^{hi-1} x_t x_{t-k}  (k < K, K a multiple of 32),  Tsum[ndim][nwc] = sum_{t=lo}^{hi-1} x_t,
 * x relative to the series' row 0.  The caller zeroes the sums once and then keeps them current:
 *   linna_chain_append_t : rows row0 .. row0 + nsteps of CT from a chain block[nsteps][nw][ldb]
 *   linna_acorr_update   : remove = 0: rows [a0, a1) entered at the end of the window [lo, hi) (hi = a1): adds their products
 *                          with the rows up to K - 1 back (never before lo) for the lags [k0, k1); the same call with
 *                          [a0, a1) = [lo, hi) computes lags from scratch (more lags on demand).  remove = 1: rows [a0, a1)
 *                          = [lo, new lo) leave at the front: subtracts their products with the rows up to K - 1 ahead.
 *                          Tsum is updated by the call whose k0 is 0.
 *   linna_acorr_tau      : emcee's estimate from the sums for the lags 0..kuse (kuse <= min(K, hi - lo) - 1): per series
 *                          acf_k = S_k - m (2 T - tail_k - head_k) + (N - k) m^2, normalised by acf_0, averaged over the nlive
 *                          walkers; tau_M = 2 sum_{k<=M} f_k - 1 at the first M with M >= c tau_M.  out[ndim] = tau (NaN as the
 *                          host estimator gives it for a constant series), out[ndim + d] = M, out[2 ndim + d] = 1 when no such
 *                          M <= kuse exists and kuse < hi - lo - 1 (the caller adds lags and asks again), else 0.
 *                          scratch: linna_acorr_scratch_bytes(ndim, nwc, kuse) bytes.
 *   linna_chain_meanstd  : out[d][h][2] = mean and population standard deviation of parameter d over rows [t0, tm) (h = 0) and
 *                          [tm, t1) (h = 1), the first nws lanes (all walkers): the moments checkmeanstd compares. */
int linna_chain_append_t(linna_ctx_t* ctx, const float* block, int ldb, int nsteps, int nw, int ndim, int wstride, float* CT,
                         int nwp, int64_t row0, void* stream);
int linna_acorr_update(linna_ctx_t* ctx, const float* CT, int ndim, int nwp, int nwc, int64_t a0, int64_t a1, int64_t lo,
                       int64_t hi, int k0, int k1, double* Ssum, double* Tsum, int remove, void* stream);
size_t linna_acorr_scratch_bytes(int ndim, int nwc, int kuse);
int linna_acorr_tau(linna_ctx_t* ctx, const float* CT, int ndim, int nwp, int nwc, int nlive, int64_t lo, int64_t hi, int kuse,
                    const double* Ssum, const double* Tsum, double c, double* scratch, double* out, void* stream);
int linna_chain_meanstd(linna_ctx_t* ctx, const float* CT, int ndim, int nwp, int nws, int64_t t0, int64_t tm, int64_t t1,
                        double* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LINNA_HIP_H */
