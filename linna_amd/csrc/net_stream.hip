// Whole-network serving kernel ("network stream"): ONE launch evaluates util.Log_prob.__call__
// (util.py:990-1021) for 16 walkers per workgroup -- prior map + input transform (util.py:339-347,
// 483-497), every layer, the output transform and the diagonal Gaussian log-likelihood
// (util.py:953-955) -- for the reference's own architectures, ChtoModelv2 / ChtoModelsimple
// (nn.py:59-133, 300-374: Linear + three residual blocks + three Linears), and for plain MLPs of
// any width up to 1024 (BASELINE configs 2/5: 33 -> 512 x 4 -> 33).
//
// MI355X mapping
//  * one 512-thread workgroup (8 waves, two per SIMD) per CU owns 16 walker rows; activations stay
//    in LDS for the whole network, double-buffered [2][16][LD] fp32 (LD = widest row + 4: the
//    stride makes ds_read_b128 and the epilogue's ds_write_b32 conflict-free);
//  * weights never touch LDS.  With 16 rows per workgroup no wave shares a weight with another
//    wave, so they are re-laid ONCE per weight update into MFMA FRAGMENT ORDER (ns_pack_kernel):
//    the B operand of a step's column tile is 1 KiB contiguous, lane-linear, and one coalesced
//    global_load_dwordx4 puts it straight into the registers the MFMA reads.  Every wave owns ONE
//    contiguous stream over the whole network; a ring of R register sets keeps R-1 steps (20 KiB
//    per wave) in flight across layer boundaries with counted s_waitcnt vmcnt;
//  * v_mfma_f32_16x16x4_f32 (exact fp32) with the k-permutation trick: lane group kq owns 4
//    consecutive k, element s of the 128-bit A and B fragments feeds MFMA s;
//  * ONE copy of the step loop serves every layer: the network is a small PROGRAM of segments.
//
// Program = list of segments, each a GEMM over the activation rows held in LDS:
//   WIDE    (N > 256, or a short K): the N columns are split over the 8 waves in passes of 512 (wave w owns column
//           tiles 4w..4w+3 of a pass), every wave runs all K steps; epilogue bias(+ReLU) -> the
//           OTHER activation buffer; one barrier after the last pass.
//   SPLIT   N <= 256: the columns form ncg = 1, 2 or 4 groups of 64 and K is split over the
//           8/ncg waves of a group (wave = kpart*ncg + group), partial sums are reduced through
//           LDS, bias(+ReLU) -> the CURRENT buffer at a column offset (two barriers).  Keeps all
//           eight waves on real columns where WIDE would leave most of its 32 tiles empty.
// A residual block  y = relu(0.1 (W2 relu(W1 x + b1) + b2) + Ws x)  (nn.py:45-56) is SPLIT
// (h = relu(W1 x + b1), written right behind x in the same buffer) + ONE GEMM over the
// concatenated K = [x ; h] with the concatenated weight [Ws | 0.1 W2] and bias 0.1 b2.
// One step = 16 k x 64 columns = 1 ds_read_b128 (A) + 4 coalesced 1-KiB global loads (B, fragment
// order, see pack below) + 16 v_mfma_f32_16x16x4_f32.
// After the last segment the output row block sits in LDS: output transform, optional store of
// d, diagonal Gaussian log-likelihood (util.py:953-955) with 32-lane shuffles.
//
// Small batches (ROWS = 8 or 4 instantiations).  With 16 rows per workgroup a batch of B rows occupies B/16 of
// the 256 CUs and the launch lasts as long as one workgroup needs for the whole network, whatever B is: an
// ensemble half step of 2048 proposals (128 workgroups), a training batch of 500 (32), the reference's own
// ensembles of 4..128 walkers (1..8).  The same program runs with 8 or 4 rows per workgroup on
// v_mfma_f32_4x4x1_16b_f32: one instruction multiplies 4 rows by the wave's 64 columns at k = 1 (lane = column;
// CBSZ = 4 broadcasts the A values of block ABID to all 16 blocks, so the ONE ds_read_b128 per step still fetches
// 16 k of every row: block b of the read holds row set b & 3, k chunk b >> 2), at the same flop rate per
// instruction cycle as 16x16x4 (measured: tools/probe/mfma4_probe.hip).  Four accumulator chains per row set (one
// per k chunk) keep dependent instructions 32 cycles apart.  The weight stream is the same bytes in another order
// (lane = column, see ns_pack_kernel), so a workgroup of 8 (4) rows needs 2x (4x) the weight bandwidth per flop:
// 64 (128) B/clk/CU at full MFMA rate against the 64 B/clk a CU's vector L1 delivers -- the small engines are
// L1-fill bound, which still halves the time of a launch that cannot fill the chip.
#include "common.h"
#include <stdlib.h>
#include <type_traits>
#include <vector>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>

namespace linna {

constexpr int NS_ROWS = 16;                // rows per workgroup of the large-batch engine (and the LDS layout bound)
constexpr int NS_NW = 8;                 // waves per workgroup
constexpr int NS_NT = 4;                 // 16-column tiles per wave and step
constexpr int NS_MAXSEG = 20;
constexpr int NS_MAXRUN = 40;
constexpr unsigned NS_STEP_B = NS_NT * 1024;
constexpr int NS_LDS_BYTES = 160 * 1024;
#ifndef NS_R
#define NS_R 6
#endif
#ifndef NS_PRE
#define NS_PRE 2
#endif
enum { NS_WIDE = 0, NS_SPLIT = 1, NS_SIDE = 2 };

struct NsSeg {            // kernel-side view of a segment
    int type, steps, passes, bias_off;
    int dst_col, relu, kslice, zext;   // SPLIT: kslice = k offset between K parts (16*steps), zext = columns written
                                       // WIDE with a SHORT second pass (zext > 0): pass 1 runs zext steps from k offset kslice -- the
                                       // lower-triangular factor of a dense inverse covariance has no rows k < 512 in columns >= 512
    int ncg_log2;                      // SPLIT: log2 of the number of 64-column groups
    int mask_store, mask_apply;        // GRAD: 1 + LDS slot of the ReLU sign bits this WIDE segment records / applies (0: none)
    int x0_n;                          // ... that many columns of them (a multiple of 16)
    int x0_col;                        // > 0: the network INPUT rows (kept aside in LDS) are copied to this column of the segment's
                                       // input buffer before it runs (ChtoModelv2_linear's input skip, nn.py:160-163,195)
    int kcl, side_off;                 // SIDE: log2 of the k chunks per wave and step; float offset of its weights from `packed`
};

struct NsArgs {
    const float* Z; int ldz; int B; int nin;
    const int* is_flat; const float* a1; const float* a2; const int* lg;
    const float* xmean; const float* xstd;
    const float* packed;                // [8 waves][G][4][64 lanes][4] then the packed biases
    int G, nseg, LD, kpad0, nout, bias_total;
    int Gstride, nseg_f;                // steps per wave in the packed stream; forward segments (GRAD: the rest is the backward)
    const float* gscale; float* Gout; int ldg;   // GRAD: d(d)/d(raw output); d lnP / d z
    float* hm_p; int hm_ldp; float* hm_q; const float* hm_mass; float hm_ek, hm_ed;   // GRAD: leapfrog kick + drift in the finish
    const float* cscale; const float* cshift; const float* w; float T;
    const float* cpost; const float* cshift2;   // ypositive output map (util.py:540): d = exp(raw cscale + cshift) cpost + cshift2
    float* lnP; float* D; int ldd; float* TH; int ldt;
    unsigned long long* stamps;
    const int* gate;                    // optional: every workgroup leaves at once when gate[0] == 0 (speculatively queued rounds)
    // STORE instantiation (training / validation forward, nn.py:110-133): the input rows are taken as they are
    // (already X-transformed), every segment's output ALSO goes to global memory for the backward, no likelihood
    float* gout[NS_MAXSEG]; int gld[NS_MAXSEG]; int gn[NS_MAXSEG];
    // STORE == 2 (the dX chain of a training step): a segment's output is zeroed where gmask (the stored forward
    // activation whose gradient it is) is not positive, before it is stored and handed to the next segment
    const float* gmask[NS_MAXSEG]; int gmld[NS_MAXSEG];
    // GRAD + STORE == 2 (lnP and its gradient in one launch, any network): the backward needs the SIGN of the forward
    // activations only, and it needs it in this very workgroup -- one bit per (row, column) in LDS ([ROWS][nbw] words at
    // float offset bits_off) instead of the activations in global memory.  gbit / mbit: the first bit column (a multiple of
    // 64) of the tensor a segment writes / is gated by, -1 = none.  The merged training launch (GRAD + STORE == 3) gates the
    // same way -- its activations still go to global memory, the parameter gradients need them.
    int nbw, bits_off;
    int gbit[NS_MAXSEG], mbit[NS_MAXSEG];
    // STORE == 3 (one launch = gather + input transform + training forward + chi^2-ratio loss and its gradient,
    // predictor_gpu.py:274-285 with util.py:1070-1116): Z is the RESIDENT training set X[n][ldz], row `t_rows[b]` is
    // transformed in the prologue (and stored to t_xb for the first layer's parameter gradient); the last network layer's
    // epilogue turns pred into delta = mask ? 0 : ynorm - pred in LDS (ynorm, with its mask, precomputed for the whole
    // set: one load per element); the program's last segment is U = delta Cinv; the finish writes
    // loss_b = delta.U / den and d loss / d pred = -2 U inv_batch / den
    const int* t_rows; float* t_xb; int t_ldxb;
    const float* t_Y; int t_ldy;            // NORMALISED targets of the whole set, NaN where masked (linna_loss_targets)
    const float* t_den; float t_inv_batch; float* t_loss_rows; float* t_dP; int t_lddp;
    // dense inverse covariance as the program's last segment: U = d S sits at column u_col of the current buffer
    // (u_same) or at column 0 with d in the other buffer; the finish takes chi2 = d . U
    int dense, u_col, u_same;
    int x0_keep;                        // the program copies the input rows somewhere later (NsSeg::x0_col): keep them in LDS
    // STORE == 2 rider (linna_net_train_step): ONE extra workgroup of the launch takes the batch mean of the loss rows the
    // forward + loss launch wrote and advances AdamW's step counter / bias corrections -- sum_scale_prepare_kernel's job,
    // on a CU the dX chain leaves idle instead of a launch of its own between the two
    const float* p_rows; int p_n; float p_scale; float* p_out; int* p_step; float* p_hyper; float p_b1, p_b2;
    // stretch move fused around the evaluation (MOVE instantiation; emcee StretchMove behind sampler.py:493-495)
    float* mv_coords; int mv_ldc; float* mv_logp; const int* mv_S;
    const float* mv_cc; int mv_ldcc; const int* mv_C; int mv_nc;
    unsigned long long mv_seed; const int* mv_step; int mv_step_off; int mv_stream; float mv_a; int* mv_naccept;
    float* mv_chain; float* mv_lps;     // MOVE == 1: row of the chain block this iteration fills (linna_stretch_run), [nw][nin] / [nw]
    // MOVE == 2, sl_Zt != null: the FIRST shrinking round of a half step whose stepping-out was one round of sl_m bracket ends per
    // side -- the trial weight of row j ns + k is derived here from that round's results instead of being read: bracket
    // [L, R] pushed out while lnP at the ends exceeds Z0 (slice_expand_multi_kernel), then trial j placed as if its
    // predecessors were rejected (slice_draw_dev: Philox (walker, step, stream, sub j + 1)).  Saves the launch between them.
    const float* sl_Z0; const float* sl_L; const float* sl_R; const float* sl_Zt; int sl_m, sl_nt;
    unsigned long long sl_seed; const int* sl_step; int sl_stream; const int* sl_flags;
    SliceBegin sb;                      // MOVE == 2, sb.logp != null: the half step's set-up in this launch's prologue (common.h)
    NsSeg seg[NS_MAXSEG];
};

// ------------------------------------------------------------------ weight re-layout
struct NsPackSeg {
    const float* Wa; int lda, Ka, Kapad;          // first K part (Wa NULL: identity)
    const float* Wb; int ldb, Kb; float alpha;    // second K part, scaled (residual blocks)
    const float* b; float bscale;
    const float* b2; float b2scale;                // second bias term (input skip: alpha * bl)
    int N, type, steps, passes, bias_off, bias_pad, ncg;
    int transA;                                   // Wa is read transposed: value(n, k) = Wa[k][n] (backward segments)
    int transB;                                   // the same for Wb
    const float* rscale; const float* rshift;     // per output column: weights and bias * rscale, bias + rshift (folded output map)
    int kc, side_off;                             // SIDE segments: k chunks per wave and step (2 or 4), float offset of their block
    int koff2;                                    // WIDE with a short second pass: k offset of pass 1 (NsSeg::kslice)
};
// SIDE segments (serving programs of the 16-row engine): a SPLIT segment of <= 32 output columns -- the hidden
// h = relu(W1 x + b1) of a residual block, 1000 -> 16 and 500 -> 32 in ChtoModelv2 -- costs the step loop 8 + 4 steps of
// which three quarters / half multiply zero weights (a wave's four 16-column tiles need 64 columns).  As a SIDE segment it
// leaves the weight stream: its weights sit in a block of their own behind the biases, laid out so that a wave's four tiles
// are (column tile, k chunk) pairs -- kc = 4 chunks of 16 columns, or 2 chunks of 32 -- and the whole segment is at most
// two steps per wave, run inside the run-end code of the segment before it (loads issued before that segment's epilogue,
// by inline asm: the compiler's counted waits for the ring never see them).  The step loop itself is untouched.
//   side[w][s][t][lane][e]:  n = 16 (t % (4 / kc)) + li,  k = 16 (kc (w steps + s) + t / (4 / kc)) + 4 kq + e
struct NsPackArgs {
    NsPackSeg seg[NS_MAXSEG];
    int run_seg[NS_MAXRUN], run_pass[NS_MAXRUN], run_first[NS_MAXRUN + 1];
    int nseg, nrun, G, bias_total;
    int small;                                     // layout of the 4x4x1 engines: lane = column, load t = k chunk
    float* out;                                    // weights, then biases
};
// stream[w][g][t][lane][e], run r = (segment, pass), s = g - first[r], li = lane & 15, kq = lane >> 4:
//   WIDE   n = 16 (32 pass + 4 w + t) + li,  k = 16 s + 4 kq + e
//   SPLIT  n = 64 (w % ncg) + 16 t + li,     k = 16 ((w / ncg) steps + s) + 4 kq + e
// value = [Wa | alpha Wb](n, k), zero outside.  Small-batch engines (p.small): the same 64 columns x 16 k per
// (w, g) with lane = column and load t = k chunk:  n = ... + lane,  k = ... + 4 t + e.
__global__ void ns_pack_kernel(NsPackArgs p) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nw4 = (size_t)NS_NW * p.G * NS_NT * 64;
    if (idx < nw4) {
        const int lane = (int)(idx & 63);
        size_t q = idx >> 6;
        const int t = (int)(q % NS_NT); q /= NS_NT;
        const int g = (int)(q % p.G);
        const int w = (int)(q / p.G);
        int r = 0;
        while (r + 1 < p.nrun && g >= p.run_first[r + 1]) ++r;
        const NsPackSeg& S = p.seg[p.run_seg[r]];
        const int s = g - p.run_first[r];
        const int nl = p.small ? lane : 16 * t + (lane & 15), kl = p.small ? 4 * t : 4 * (lane >> 4);
        int n, k0;
        if (S.type == NS_WIDE && S.koff2 < 0) {        // balanced triangular factor: block w from row 64 w, then block 15 - w
            const int n0 = S.steps - 4 * w, blk = s < n0 ? w : 15 - w;
            n = 64 * blk + nl; k0 = 16 * (s < n0 ? s : s - n0) + 64 * blk + kl;
        } else if (S.type == NS_WIDE) { n = 512 * p.run_pass[r] + 64 * w + nl; k0 = 16 * s + kl + (p.run_pass[r] ? S.koff2 : 0); }
        else { n = 64 * (w % S.ncg) + nl; k0 = 16 * ((w / S.ncg) * S.steps + s) + kl; }
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (n < S.N && S.Wa && !S.transA && k0 + 3 < S.Ka && (S.lda & 3) == 0 && (reinterpret_cast<uintptr_t>(S.Wa) & 15) == 0) {
            // the common case: four consecutive k of one weight row, 16 bytes aligned (rows are padded to 4 floats)
            v = *reinterpret_cast<const f32x4*>(S.Wa + (size_t)n * S.lda + k0);
            if (S.rscale) { const float r = S.rscale[n]; v = f32x4{v[0] * r, v[1] * r, v[2] * r, v[3] * r}; }
        } else if (n < S.N) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = k0 + e;
                if (k < S.Kapad) {
                    if (k < S.Ka) v[e] = !S.Wa ? (k == n ? 1.f : 0.f) : S.transA ? S.Wa[(size_t)k * S.lda + n] : S.Wa[(size_t)n * S.lda + k];
                } else if (k - S.Kapad < S.Kb) {
                    v[e] = S.alpha * (S.transB ? S.Wb[(size_t)(k - S.Kapad) * S.ldb + n] : S.Wb[(size_t)n * S.ldb + (k - S.Kapad)]);
                }
            }
            if (S.rscale) { const float r = S.rscale[n]; v = f32x4{v[0] * r, v[1] * r, v[2] * r, v[3] * r}; }
        }
        reinterpret_cast<f32x4*>(p.out)[idx] = v;
        return;
    }
    const size_t j = idx - nw4;                       // packed biases, one float per thread
    if (j >= (size_t)p.bias_total) return;
    int si = 0;
    while (si + 1 < p.nseg && (int)j >= p.seg[si + 1].bias_off) ++si;
    const NsPackSeg& S = p.seg[si];
    const int c = (int)j - S.bias_off;
    float bv = (c < S.N && S.b) ? S.bscale * S.b[c] : 0.f;
    if (c < S.N && S.b2) bv += S.b2scale * S.b2[c];
    if (c < S.N && S.rscale) bv *= S.rscale[c];
    if (c < S.N && S.rshift) bv += S.rshift[c];
    p.out[nw4 * 4 + j] = bv;
}

__global__ void ns_pack_side_kernel(NsPackArgs p) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // one f32x4 of one SIDE segment's block
    size_t base = 0;
    for (int si = 0; si < p.nseg; ++si) {
        const NsPackSeg& S = p.seg[si];
        if (S.type != NS_SIDE) continue;
        const size_t n4 = (size_t)NS_NW * S.steps * NS_NT * 64;
        if (idx >= base + n4) { base += n4; continue; }
        size_t q = idx - base;
        const int lane = (int)(q & 63); q >>= 6;
        const int t = (int)(q % NS_NT); q /= NS_NT;
        const int st = (int)(q % S.steps);
        const int w = (int)(q / S.steps);
        const int tpc = NS_NT / S.kc;
        const int n = 16 * (t % tpc) + (lane & 15);
        const int k0 = 16 * (S.kc * (w * S.steps + st) + t / tpc) + 4 * (lane >> 4);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (n < S.N) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = k0 + e;
                if (S.Wa) { if (k < S.Ka) v[e] = S.Wa[(size_t)n * S.lda + k]; }
                else if (k < S.Kb) v[e] = S.alpha * (S.transB ? S.Wb[(size_t)k * S.ldb + n] : S.Wb[(size_t)n * S.ldb + k]);   // d/dh: the second K part alone
            }
        }
        reinterpret_cast<f32x4*>(p.out + S.side_off)[idx - base] = v;
        return;
    }
}

__device__ __forceinline__ float ns_prior_theta(float z, int flat, float a1, float a2) {
    float u = 0.5f * (1.f + erff(z / 1.41421356237309515f));
    asm volatile("" : "+v"(u));                    // computed unconditionally: no branch on the loaded flag
    return (flat ? u : z) * a2 + a1;
}

// ------------------------------------------------------------------ the kernel
// MOVE: one ensemble half step in the launch.  Row k of the batch is walker S[k]: the prologue draws the
// stretch proposal q = c + z (s - c) from the complementary walkers (what linna_stretch_propose writes
// to memory), the network evaluates lnP(q), the finish applies the Metropolis test of
// linna_stretch_accept and updates coords / logp / naccept in place.  Same Philox counters, same
// arithmetic: bit-identical to the three-launch sequence.
// GRAD: lnP AND d lnP / d z in the launch (what torch.autograd.grad(lnP, x) yields at HMCSampler.py:32,40,48),
// for ReLU MLPs: every hidden layer records the sign bits of its output in LDS (one word per lane: the
// backward GEMM of the next layer has the same lane <-> (row, column) map), after the last layer the
// finish turns the output rows into d lnP / d out in place, and the SAME step loop runs on through the
// backward segments -- W^T in fragment order, streamed right behind the forward weights -- ending in the
// prior map's derivative.
// STORE: the forward pass of a training step in one launch.  The activations the backward needs (every op's
// output, every residual block's hidden h) are written to global memory from the epilogues with inline-asm
// stores: the compiler does not see them, so its counted vmcnt waits for the weight stream stay counted
// (stores only ever make the hardware counter read higher, i.e. the waits conservative).
template <int R, int MOVE, bool GRAD, int STORE, int ROWS>
__global__ __launch_bounds__(64 * NS_NW, 1) void net_stream_kernel(NsArgs a) {
    constexpr int NT = NS_NT, NW = NS_NW;
    constexpr int RG = 32;                         // threads per walker row in prologue / reduce / finish
    constexpr bool SM = ROWS < 16;                 // 4x4x1 engine: ROWS / 4 row sets, lane = column
    constexpr int RS = SM ? ROWS / 4 : 1;
    constexpr int NQ = SM ? RS : NT;               // result quads per lane: (row set) or (column tile)
    constexpr int NACC = SM ? 4 * RS : NT;
    constexpr bool K4 = !SM && (STORE == 0 || (GRAD && STORE == 2));   // 16-row engine, serving and the one-launch gradient: programs may hold SIDE segments
    // (the one-launch gradient with SIDE segments in its forward half was measured SLOWER: 156.4 against 154.2 us at ChtoModelv2(33,33))
    // TRB: a whole training step's network work in ONE launch (linna_net_train_step): gather + transform + forward with the
    // activations kept + chi^2-ratio loss (STORE == 3) as the forward half, the loss finish as the TURNAROUND (loss rows,
    // d loss / d pred to memory AND into LDS as the input of the first backward segment), then the dX chain (STORE == 2's
    // epilogues: gates from the activations this very launch stored, read through the L2) -- one prologue and one launch
    // boundary less than forward + loss and dX chain as two launches, the weight ring never drained in between.
    constexpr bool TRB = GRAD && STORE == 3;
    constexpr bool DXE = STORE == 2 || TRB;        // epilogues of dX segments: gate by the stored activation, store
    constexpr bool G2 = GRAD && STORE == 2;        // lnP + gradient in one launch: nothing of the forward half goes to global memory
    // the gates are sign bits in LDS (NsArgs::nbw).  The merged training launch can gate the same way (LB = G2 || TRB: its
    // launcher fills gbit / mbit), measured SLOWER on its 4-row engine (155.4 against 152.4 us per step at (26,457): the
    // gate loads are not what its run ends wait for, the ballots and bit writes of every forward epilogue are extra) -- off.
    constexpr bool LB = G2;
    constexpr bool LATE_REFILL = true;             // see `step`
    static_assert(ROWS == 16 || ROWS == 8 || ROWS == 4, "rows per workgroup");
    static_assert(R % 2 == 0, "the A double buffer alternates with the ring slot parity");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int LD = a.LD, ABUF = ROWS * LD;
    // The per-segment tables are indexed at run time.  Through `a` (a by-value argument) such an index makes the compiler
    // keep a private copy of the whole 2.3 KB block whenever it cannot split it -- scratch traffic at every run end, and which
    // instantiation is hit changes with unrelated edits (STORE == 1 on the 16-row engine in round 2, the merged training
    // launch with the refill one slot back in round 3).  Through the kernel-argument segment itself they are scalar loads.
    // (Only in the instantiations where that copy has appeared -- the training ones: the serving ones and the one-launch
    // gradient lose 0.3-0.7 % to the explicit pointer; tests/test_abi.py watches every kernel's scratch size.)
    constexpr bool KA = STORE == 3 || STORE == 1 || GRAD;    // (R4: every GRAD instantiation -- G2 since it holds SIDE code, the MLP gradient since the short-second-pass selects)
    const NsArgs* const ka = KA ? reinterpret_cast<const NsArgs*>((const void*)__builtin_amdgcn_kernarg_segment_ptr()) : &a;
    float* const act = smem;                       // [2][ROWS][LD]
    float* const lbias = smem + 2 * ABUF;          // packed biases of every segment
    unsigned* const lbits = reinterpret_cast<unsigned*>(smem + a.bits_off);   // G2: sign bits [ROWS][nbw]
    const int nbw = a.nbw;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    const int row0 = blockIdx.x * ROWS;
    if (a.gate && a.gate[0] == 0) return;
    // MOVE == 2 with a row list (the later rounds of linna_slice_half_step): only the trial points of the walkers still
    // active are evaluated -- row r of the launch is trial point mv_C[r] (= j ns + k), mv_step[0] * mv_step_off rows in
    // all, counted on the device; the workgroups behind them leave at once
    int mv_rows = a.B;
    if constexpr (MOVE == 2) {
        if (a.mv_C) {
            mv_rows = a.mv_step[0] * a.mv_step_off;
            if (row0 >= mv_rows) return;
        }
    }
    if constexpr ((STORE == 2 && !GRAD) || TRB) {
        if ((a.p_n > 0 || a.p_step) && blockIdx.x == gridDim.x - 1) {
            // sum_scale_prepare_kernel's arithmetic in its order: 1024 strided partial sums (two per thread here),
            // sixteen wave sums, added in wave order
            float* const part = smem;
            float acc0 = 0.f, acc1 = 0.f;
            for (int i = tid; i < a.p_n; i += 1024) acc0 += a.p_rows[i];
            for (int i = tid + 512; i < a.p_n; i += 1024) acc1 += a.p_rows[i];
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) { acc0 += __shfl_xor(acc0, o, 64); acc1 += __shfl_xor(acc1, o, 64); }
            if (lane == 0) { part[wave] = acc0; part[8 + wave] = acc1; }
            __syncthreads();
            if (tid == 0 && a.p_out) {
                float t = 0.f;
                for (int w = 0; w < 16; ++w) t += part[w];
                a.p_out[0] = t * a.p_scale;
            }
            if (tid == 64 && a.p_step) {
                const int t = ++a.p_step[0];
                a.p_hyper[2] = (float)(1.0 - pow((double)a.p_b1, (double)t));
                a.p_hyper[3] = (float)sqrt(1.0 - pow((double)a.p_b2, (double)t));
            }
            return;
        }
    }
#ifdef NS_STAMPS
    unsigned long long* const lstamp = reinterpret_cast<unsigned long long*>(lbias + ((a.bias_total + 3) & ~3) + 32 + (a.x0_keep ? 1024 : 0)) + wave * 32;   // (not for GRAD: its masks live there)
    int nstamp = 0;
#define NS_STAMP() do { const unsigned long long t_ = __builtin_readcyclecounter(); \
        if (lane == 0 && nstamp < 32) lstamp[nstamp] = t_; ++nstamp; } while (0)
#define NS_STAMPS_FLUSH() do { if (lane < 32) a.stamps[((size_t)blockIdx.x * NW + wave) * 32 + lane] = lane < nstamp ? lstamp[lane] : 0ull; } while (0)
#else
#define NS_STAMP() do {} while (0)
#define NS_STAMPS_FLUSH() do {} while (0)
#endif
    NS_STAMP();

    // ---- 1. every small load of the kernel, up front, straight-line (no branch on a loaded value)
    const int prt = tid / RG, pc0 = tid % RG;
    const bool prow = prt < ROWS;                   // (ROWS < 16: the thread rows past ROWS only keep the barriers company)
    const int pr = SM ? min(prt, ROWS - 1) : prt;
    int grow = min(row0 + pr, a.B - 1);
    if constexpr (MOVE == 2) {
        if (a.mv_C) grow = a.mv_C[min(row0 + pr, mv_rows - 1)];     // the trial point this row evaluates
    }
    int zsrc = grow;                                // STORE == 3: the row of the resident set this batch row is
    float zden = 1.f;
    if constexpr (STORE == 3) {
        zsrc = a.t_rows ? a.t_rows[grow] : grow;
        zden = a.t_den[zsrc];
    }
    const int kpad0 = a.kpad0, nin = a.nin, nout = a.nout, nseg = a.nseg;
    const int nlast = (TRB ? a.nseg_f : nseg) - 2;  // STORE == 3: the network's last layer (the loss segment follows it)
    constexpr int ZPRE = 2;
    float zr[ZPRE], za1[ZPRE], za2[ZPRE], zxm[ZPRE], zxs[ZPRE]; int zfl[ZPRE], zlg[ZPRE];
    const int* const lgp = a.lg ? a.lg : a.is_flat;
    int mv_wk = 0; float mv_factor = 0.f, mv_lnp_old = 0.f, mv_logu = 0.f;
    if constexpr (MOVE == 1) {
        mv_wk = a.mv_S[grow];
        const U4 rb = walker_bits(a.mv_seed, (uint32_t)mv_wk, (uint32_t)(a.mv_step[0] + a.mv_step_off), (uint32_t)a.mv_stream, 0u);
        const float t = (a.mv_a - 1.f) * u01(rb.x) + 1.f;
        const float zf = t * t / a.mv_a;
        const int j = (int)(((uint64_t)rb.y * (uint64_t)a.mv_nc) >> 32);
        const int wc = a.mv_C[j];
        mv_factor = ((float)nin - 1.f) * logf(zf);
        mv_logu = logf(u01(rb.z));
        mv_lnp_old = a.mv_logp[mv_wk];
#pragma unroll
        for (int i = 0; i < ZPRE; ++i) {
            const int c = min(pc0 + i * RG, nin - 1);
            const float cr = a.mv_cc[(size_t)wc * a.mv_ldcc + c], sx = a.mv_coords[(size_t)mv_wk * a.mv_ldc + c];
            zr[i] = cr - (cr - sx) * zf;
        }
    }
    if constexpr (MOVE == 2) {
        // trial points of the ensemble slice sampler, never written to memory: row j*ns + k is
        // coords[S[k]] + w[j*ns + k] * DIR[k]   (what linna_slice_points materialises)
        const int k = grow % a.mv_nc;                                   // mv_nc: ns (walkers per half ensemble)
        const int wk = a.mv_S[k];
        float wgt;
        if (a.sb.logp) {
            // slice_begin_kernel's arithmetic, per row: two distinct complementary walkers, the direction between them, a uniform
            // height under the density, a unit bracket placed uniformly around 0; this row's end of it, jt steps out
            const SliceBegin& b = a.sb;
            const int jt = grow / a.mv_nc;
            const U4 rb = walker_bits(b.seed, (uint32_t)wk, (uint32_t)b.step[0], (uint32_t)b.half, 0u);
            const int ia = (int)(((uint64_t)rb.x * (uint64_t)b.nc) >> 32);
            int ib = (int)(((uint64_t)rb.y * (uint64_t)(b.nc - 1)) >> 32);
            ib += (ib >= ia);
            const int wa = b.C[ia], wb = b.C[ib];
            const float mu = b.mu[0];
            const float l = -u01(rb.w);
            wgt = jt < b.m ? l - (float)jt : l + 1.f + (float)(jt - b.m);
            const bool first = jt == 0 && prow && row0 + pr < a.B;      // the rows that write the walker's state for the later launches
#pragma unroll
            for (int i = 0; i < ZPRE; ++i) {
                const int c = min(pc0 + i * RG, nin - 1);
                const float dir = mu * (b.cc[(size_t)wa * b.ldcc + c] - b.cc[(size_t)wb * b.ldcc + c]);
                zr[i] = a.mv_coords[(size_t)wk * a.mv_ldc + c] + wgt * dir;
                if (first && pc0 + i * RG < nin) b.DIR[(size_t)k * b.ldd + c] = dir;
            }
            if (first && pc0 == 0) {
                b.Z0[k] = b.logp[wk] + logf(u01(rb.z));
                b.L[k] = l; b.R[k] = l + 1.f;
                int J, K;
                slice_budget(b.seed, (uint32_t)wk, (uint32_t)b.step[0], (uint32_t)b.half, b.maxsteps, J, K);
                b.flags[3 * k] = J; b.flags[3 * k + 1] = K; b.flags[3 * k + 2] = 1;
            }
            if (blockIdx.x == 0 && tid == 0) {
                for (int i = 0; i < b.nslots; ++i) {       // [4 + nslots + i]: the same counts summed over the calls so far (usage statistics)
                    b.counters[4 + b.nslots + i] += b.counters[4 + i];
                    b.counters[4 + i] = 0;
                }
                b.counters[4 + 2 * b.nslots] += 1;         // calls
                if (b.zero_totals) { b.counters[0] = 0; b.counters[1] = 0; }
            }
        } else {
        if (a.sl_Zt) {
            // lanes 0 .. 2 m - 1 of the row's 32 hold "lnP at bracket end j exceeds Z0"; the count of leading ones per side is
            // the number of steps out.  Lane i < nt holds the uniform of trial i; the row walks the trials up to its own.
            const int ns_ = a.mv_nc, m = a.sl_m, jt = grow / ns_;
            const float z0 = a.sl_Z0[k];
            const float zend = a.sl_Zt[(size_t)min(pc0, 2 * m - 1) * ns_ + k];
            const unsigned long long bal = __ballot(pc0 < 2 * m && zend > z0);
            const unsigned bits = (unsigned)(bal >> (lane & 32));
            const unsigned lm = bits & ((1u << m) - 1u), rm = (bits >> m) & ((1u << m) - 1u);
            // (never more steps than the budget of that side has left: slice_side_steps in common.h)
            const int nl = min(__builtin_ctz(~lm), a.sl_flags[3 * k]), nr = min(__builtin_ctz(~rm), a.sl_flags[3 * k + 1]);
            float l = a.sl_L[k], r = a.sl_R[k];
            for (int j = 0; j < m; ++j) {                               // (one unit at a time: the rounding of slice_expand_multi_kernel)
                if (j < nl) l -= 1.f;
                if (j < nr) r += 1.f;
            }
            const U4 tb = walker_bits(a.sl_seed, (uint32_t)wk, (uint32_t)a.sl_step[0], (uint32_t)a.sl_stream, (uint32_t)(pc0 + 1));
            const int myu = __float_as_int(u01(tb.x));
            wgt = 0.f;
            for (int jj = 0; jj < a.sl_nt; ++jj) {
                const int ulo = __builtin_amdgcn_readlane(myu, jj), uhi = __builtin_amdgcn_readlane(myu, 32 + jj);
                const float u = __int_as_float((lane & 32) ? uhi : ulo);
                const float w = l + u * (r - l);
                if (jj == jt) wgt = w;
                if (jj < jt) { if (w < 0.f) l = w; else r = w; }
            }
        } else {
            wgt = a.mv_cc[grow];                                        // mv_cc: w[nrep * ns]
        }
#pragma unroll
        for (int i = 0; i < ZPRE; ++i) {
            const int c = min(pc0 + i * RG, nin - 1);
            zr[i] = a.mv_coords[(size_t)wk * a.mv_ldc + c] + wgt * a.Z[(size_t)k * a.ldz + c];   // Z: DIR[ns][ldz]
        }
        }
    }
#pragma unroll
    for (int i = 0; i < ZPRE; ++i) {
        const int c = min(pc0 + i * RG, nin - 1);
        if constexpr (MOVE == 0) zr[i] = a.Z[(size_t)(STORE == 3 ? zsrc : grow) * a.ldz + c];
        zfl[i] = a.is_flat[c]; za1[i] = a.a1[c]; za2[i] = a.a2[c];
        zlg[i] = lgp[c]; zxm[i] = a.xmean[c]; zxs[i] = a.xstd[c];
    }
    const int nb4 = (a.bias_total + 3) >> 2;       // packed biases, 16 bytes per thread and round
    const f32x4* const bsrc = reinterpret_cast<const f32x4*>(a.packed + (size_t)NW * a.Gstride * NT * 256);
    constexpr int BMAX = 3;                        // 3 x 512 x 4 floats >= every eligible network's biases
    f32x4 breg[BMAX];
#pragma unroll
    for (int i = 0; i < BMAX; ++i) {
        const int j = tid + i * 64 * NW;
        breg[i] = bsrc[min(j, nb4 - 1)];
    }
    constexpr int FIN = 2;
    const float* const csp = a.cscale ? a.cscale : a.xmean;   // always a readable pointer
    const float* const ctp = a.cshift ? a.cshift : a.xmean;
    const float* const wtp = a.w ? a.w : a.xmean;
    float fcs[FIN], fct[FIN], fw[FIN], fgs[FIN];
#pragma unroll
    for (int i = 0; i < FIN; ++i) {
        const int cc = min(pc0 + i * RG, nout - 1);
        if constexpr (GRAD && !TRB) fgs[i] = a.gscale[cc]; else fgs[i] = 0.f;
        const int c1 = a.cscale ? cc : 0, c2 = a.cshift ? cc : 0, c3 = a.w ? cc : 0;
        const float cs = csp[c1], ct = ctp[c2], ww = wtp[c3];
        fcs[i] = a.cscale ? cs : 1.f; fct[i] = a.cshift ? ct : 0.f; fw[i] = a.w ? ww : 0.f;
    }

    // ---- 2. weight stream: wave-uniform base + 32-bit per-lane offset + immediate
    const char* const wbase = reinterpret_cast<const char*>(a.packed) + (size_t)wave * a.Gstride * NS_STEP_B;
    const unsigned wlast = (unsigned)(a.G - 1) * NS_STEP_B;
    unsigned woff = 0;
    const unsigned voff = 16u * (unsigned)lane;
    f32x4 Bq[R][NT];
    auto wload = [&](int t) { return *reinterpret_cast<const f32x4*>(wbase + (size_t)(voff + woff) + t * 1024); };
    auto wadvance = [&]() { woff = min(woff + NS_STEP_B, wlast); };   // past the end: reload the last step (never used)
    constexpr int PRE = NS_PRE < R ? NS_PRE : R;
    auto prefetch = [&](auto Uc) {
        constexpr int U = decltype(Uc)::value;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            Bq[U][t] = wload(t);
            __builtin_amdgcn_sched_barrier(0);     // keep the issue order: the loop's counted vmcnt depends on it
        }
        wadvance();
    };
#define NS_PF(U, LO, HI) if constexpr (U >= LO && U < HI) prefetch(std::integral_constant<int, U>{});
#define NS_PF_ALL(LO, HI) NS_PF(0, LO, HI) NS_PF(1, LO, HI) NS_PF(2, LO, HI) NS_PF(3, LO, HI) NS_PF(4, LO, HI) NS_PF(5, LO, HI) \
    NS_PF(6, LO, HI) NS_PF(7, LO, HI)
    NS_PF_ALL(0, PRE)
#ifdef NS_STAMPS_FINE
    NS_STAMP();                                    // every small load and the first weight slots requested
#endif

    // ---- 3. network input x = X_transform(Transform(z)) into buffer 0, zero padded to kpad0; biases to LDS
    float zz = 0.f;
    float theta[ZPRE];
#pragma unroll
    for (int i = 0; i < ZPRE; ++i) {
        const int c = pc0 + i * RG;
        const bool in = c < nin;
        const float z = in ? zr[i] : 0.f;
        zz += z * z;
        float th = ns_prior_theta(z, zfl[i], za1[i], za2[i]);
        float lt = log10f(th);
        asm volatile("" : "+v"(lt));
        theta[i] = th;
        const float t = (a.lg && zlg[i]) ? lt : th;
        float x = in ? (t - zxm[i]) / zxs[i] : 0.f;
        if constexpr ((STORE == 1 || STORE == 2) && !GRAD) x = z;   // rows arrive transformed
        if constexpr (STORE == 3) {                 // X_transform of a gathered row (util.py:483-497), as linna_gather_xform
            float lz = log10f(z);
            asm volatile("" : "+v"(lz));
            const float tz = (a.lg && zlg[i]) ? lz : z;
            x = in ? (tz - zxm[i]) / zxs[i] : 0.f;
            if (prow && row0 + pr < a.B && c < a.t_ldxb)
                asm volatile("global_store_dword %0, %1, off" :: "v"(a.t_xb + (size_t)(row0 + pr) * a.t_ldxb + c), "v"(x) : "memory");
        }
        if (c < kpad0 && prow) act[pr * LD + c] = x;
    }
#ifdef NS_STAMPS_FINE
    NS_STAMP();                                    // first 64 input columns transformed and in LDS
#endif
    // inputs wider than ZPRE*RG = 64 columns (none of the reference's models; <= 256 supported) and the zero
    // pad of a SPLIT first segment: plain loop, loads waited in place
    if constexpr ((STORE == 1 || STORE == 2) && !GRAD) {
        // rows arrive transformed (the dX chain's input is d loss / d pred, 457 or 1000 columns wide): every thread of the
        // workgroup copies, four independent loads in flight each -- the per-row loop below waits for every load in place
        const int wcols = kpad0 - ZPRE * RG;
        if (wcols > 0) {
            constexpr int CP = 4;
            for (int base = 0; base < ROWS * wcols; base += CP * 64 * NW) {
                float v[CP];
#pragma unroll
                for (int u = 0; u < CP; ++u) {
                    const int idx = min(base + u * 64 * NW + tid, ROWS * wcols - 1);
                    const int r = idx / wcols, c = ZPRE * RG + idx % wcols;
                    v[u] = a.Z[(size_t)min(row0 + r, a.B - 1) * a.ldz + min(c, nin - 1)];
                }
#pragma unroll
                for (int u = 0; u < CP; ++u) {
                    const int idx = base + u * 64 * NW + tid;
                    if (idx < ROWS * wcols) {
                        const int r = idx / wcols, c = ZPRE * RG + idx % wcols;
                        act[r * LD + c] = c < nin ? v[u] : 0.f;
                    }
                }
            }
        }
    } else
#pragma unroll 1
    for (int c = pc0 + ZPRE * RG; c < kpad0; c += RG) {
        float x = 0.f;
        if (c < nin) {
            const float z = a.Z[(size_t)grow * a.ldz + c];
            zz += z * z;
            const float th = ns_prior_theta(z, a.is_flat[c], a.a1[c], a.a2[c]);
            const float t = (a.lg && a.lg[c]) ? log10f(th) : th;
            x = (t - a.xmean[c]) / a.xstd[c];
            if constexpr ((STORE == 1 || STORE == 2) && !GRAD) x = z;
            if constexpr (STORE == 3) {
                const float zz3 = a.Z[(size_t)zsrc * a.ldz + c];
                x = (((a.lg && a.lg[c]) ? log10f(zz3) : zz3) - a.xmean[c]) / a.xstd[c];
            }
        }
        if constexpr (STORE == 3) {
            if (prow && row0 + pr < a.B && c < a.t_ldxb)
                asm volatile("global_store_dword %0, %1, off" :: "v"(a.t_xb + (size_t)(row0 + pr) * a.t_ldxb + c), "v"(x) : "memory");
        }
        if (prow) act[pr * LD + c] = x;
    }
    __builtin_amdgcn_sched_barrier(0);
#ifdef NS_STAMPS_FINE
    NS_STAMP();                                    // the rest of the input in LDS
#endif
    NS_PF_ALL(PRE, (LATE_REFILL ? R - 1 : R))
#undef NS_PF_ALL
#undef NS_PF
#pragma unroll
    for (int o = RG / 2; o >= 1; o >>= 1) zz += __shfl_xor(zz, o, 64);
#pragma unroll
    for (int i = 0; i < BMAX; ++i) {
        const int j = tid + i * 64 * NW;
        if (j < nb4) reinterpret_cast<f32x4*>(lbias)[j] = breg[i];
    }
    int* const lsrc = reinterpret_cast<int*>(lbias + ((a.bias_total + 3) & ~3));      // STORE == 3: [ROWS] set rows, [ROWS] den
    float* const lden = reinterpret_cast<float*>(lsrc + 16);
    if constexpr (STORE == 3) {
        if (prow && pc0 == 0) { lsrc[pr] = zsrc; lden[pr] = zden; }
    }
    float* const lx0 = lden + 16;                  // [ROWS][64]: the input rows, for a later input-skip segment
    if (a.x0_keep && prow) {
#pragma unroll
        for (int i = 0; i < ZPRE; ++i) lx0[pr * 64 + pc0 + i * RG] = (pc0 + i * RG < kpad0) ? act[pr * LD + pc0 + i * RG] : 0.f;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                  // raw: __syncthreads() would drain the weight stream
    asm volatile("" ::: "memory");
    NS_STAMP();

    // ---- 4. the step loop
    const uint32_t act_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)act;
    f32x4 acc[NACC];
    f32x4 Aq[2];
    // 4x4x1 engine: block b = lane >> 2 of the A read holds row set b & 3 (rows wrap below ROWS: never selected), k chunk b >> 2
    const int sm_arow = (4 * ((lane >> 2) & 3) + (lane & 3)) % ROWS, sm_achunk = lane >> 4;
    int si = 0, pass = 0, P = 0, kleft;
    int s_type, s_steps, s_passes, s_bias, s_dst, s_relu, s_kslice, s_zext, s_ncgl, s_mstore = 0, s_mapply = 0, s_x0col = 0, s_x0n = 0, s_kcl = 0;
    unsigned* const lmask = reinterpret_cast<unsigned*>(lbias + ((a.bias_total + 3) & ~3));   // GRAD: [slot][512 lanes]
    float lnp_grad = 0.f;                          // GRAD: lnP, stored at the very end (no store next to the weight loads)
    float* s_gout = nullptr; int s_gld = 0, s_gn = 0;   // STORE: global destination of the current segment's output
    const float* s_gmask = nullptr; int s_gmld = 0;     // STORE == 2: forward activation gating it
    int s_gbit = -1, s_mbit = -1;                       // LB: bit column of the signs this segment writes / is gated by
    auto gstore = [&](float* p, float v) { asm volatile("global_store_dword %0, %1, off" :: "v"(p), "v"(v) : "memory"); };
    uint32_t ap;
    auto a_read = [&](f32x4& dst) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(ap) : "memory");
        ap += 64;
    };
    // (kernel-argument arrays are indexed through readfirstlane: one instantiation -- STORE == 1 on the 16-row engine --
    // could not prove the segment index uniform and copied the whole 2 KB argument block to scratch)
    auto load_seg = [&]() {
        const NsSeg S = ka->seg[__builtin_amdgcn_readfirstlane(si)];
        s_type = S.type; kleft = s_steps = S.steps; s_passes = S.passes; s_bias = S.bias_off;
        s_dst = S.dst_col; s_relu = S.relu; s_kslice = S.kslice; s_zext = S.zext; s_ncgl = S.ncg_log2; s_x0col = S.x0_col; s_x0n = S.x0_n;
        if constexpr (GRAD) { s_mstore = S.mask_store; s_mapply = S.mask_apply; }
        if constexpr (STORE) { const int j = __builtin_amdgcn_readfirstlane(si); s_gout = ka->gout[j]; s_gld = ka->gld[j]; s_gn = ka->gn[j]; }
        if constexpr (DXE) { const int j = __builtin_amdgcn_readfirstlane(si); s_gmask = ka->gmask[j]; s_gmld = ka->gmld[j]; }
        if constexpr (LB) { const int j = __builtin_amdgcn_readfirstlane(si); s_gbit = ka->gbit[j]; s_mbit = ka->mbit[j]; }
    };
    auto begin_run = [&]() {                       // accumulators and A pointer of run (si, pass)
        const int arow = SM ? sm_arow : li, ak = SM ? 4 * sm_achunk : 4 * kq;
#pragma unroll
        for (int t = 0; t < NACC; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (s_type == NS_WIDE) {
            // the wave's 64-column block of this run: 8 pass + wave -- or, balanced triangular factor (zext < 0), w then 15 - w
            const int blk = s_zext < 0 ? (pass ? 15 - wave : wave) : 8 * pass + wave;
            if constexpr (SM) {
                const float b = lbias[s_bias + 64 * blk + lane];
#pragma unroll
                for (int r = 0; r < RS; ++r) acc[4 * r] = f32x4{b, b, b, b};
            } else {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float b = lbias[s_bias + 16 * (4 * blk + t) + li];
                    acc[t] = f32x4{b, b, b, b};
                }
            }
            if (s_zext < 0) kleft = s_steps - 4 * blk;             // its rows start at 64 blk
            ap = act_lds + 4u * (uint32_t)(P * ABUF + arow * LD + ak + (s_zext < 0 ? 64 * blk : (pass ? s_kslice : 0)));   // (kslice: 0 but for a short second pass)
        } else {
            ap = act_lds + 4u * (uint32_t)(P * ABUF + arow * LD + ak + (wave >> s_ncgl) * s_kslice);
        }
    };
    auto lds_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    load_seg();
    begin_run();
    a_read(Aq[0]);

    auto step = [&](auto Uc, auto Refill) {
        constexpr int U = decltype(Uc)::value;
        constexpr bool refill = decltype(Refill)::value;
        // the refill of this step goes to the slot the PREVIOUS step consumed: a load whose target the MFMAs just issued
        // still read waits for them at issue (measured: the same loads one slot back, 1.2-1.4 % off every launch; R - 1
        // steps are in flight instead of R).  Not in the merged training launch: its register allocation does not survive
        // the longer slot lifetimes (2.3 KB of scratch per lane, +40 % on the step).
        constexpr int RU = LATE_REFILL ? (U + R - 1) % R : U;
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(Aq[U & 1]) :: "memory");
        a_read(Aq[(U + 1) & 1]);                   // next step's A (speculative at a run end)
        const f32x4 av = Aq[U & 1];
        if constexpr (SM) {
            // acc[4 r + c] += A(rows of set r, k chunk c, element e) x B(k chunk c = load c, element e); ABID = block 4 c + r
#define NS_M4(r, c, e) acc[4 * (r) + (c)] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[e], Bq[U][c][e], acc[4 * (r) + (c)], 4, 4 * (c) + (r), 0);
#define NS_M4R(r, e) NS_M4(r, 0, e) NS_M4(r, 1, e) NS_M4(r, 2, e) NS_M4(r, 3, e)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                NS_M4R(0, e)
                if constexpr (RS > 1) { NS_M4R(1, e) }
            }
#undef NS_M4R
#undef NS_M4
            if constexpr (refill) {
#pragma unroll
                for (int t = 0; t < NT; ++t) Bq[RU][t] = wload(t);
            }
        } else {
#pragma unroll
            for (int h = 0; h < NT; h += 2) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    acc[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], Bq[U][h][s], acc[h], 0, 0, 0);
                    acc[h + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], Bq[U][h + 1][s], acc[h + 1], 0, 0, 0);
                }
                if constexpr (refill) {
                    Bq[RU][h] = wload(h);
                    Bq[RU][h + 1] = wload(h + 1);
                }
            }
        }
        if constexpr (refill) wadvance();
        if (--kleft == 0) {
            // ---- end of run (si, pass).  The next segment's descriptor is requested first: the scalar
            // load's latency then hides under the epilogue stores and the barrier.
            const int nxi = __builtin_amdgcn_readfirstlane(min(si + 1, nseg - 1));
            const NsSeg NX = ka->seg[nxi];
            float* nx_gout = nullptr; int nx_gld = 0, nx_gn = 0;
            const float* nx_gmask = nullptr; int nx_gmld = 0;
            if constexpr (STORE) { nx_gout = ka->gout[nxi]; nx_gld = ka->gld[nxi]; nx_gn = ka->gn[nxi]; }
            if constexpr (DXE) { nx_gmask = ka->gmask[nxi]; nx_gmld = ka->gmld[nxi]; }
            int nx_gbit = -1, nx_mbit = -1;
            if constexpr (LB) { nx_gbit = ka->gbit[nxi]; nx_mbit = ka->mbit[nxi]; }
            const int cur_steps = s_steps;
            auto take_seg = [&](const NsSeg& X) {
                s_type = X.type; s_steps = X.steps; s_passes = X.passes; s_bias = X.bias_off;
                s_dst = X.dst_col; s_relu = X.relu; s_kslice = X.kslice; s_zext = X.zext; s_ncgl = X.ncg_log2; s_x0col = X.x0_col; s_x0n = X.x0_n;
                if constexpr (K4) s_kcl = X.kcl;
                if constexpr (GRAD) { s_mstore = X.mask_store; s_mapply = X.mask_apply; }
                kleft = X.steps;
            };
            auto take_next = [&]() {
                take_seg(NX);
                if constexpr (STORE) { s_gout = nx_gout; s_gld = nx_gld; s_gn = nx_gn; }
                if constexpr (DXE) { s_gmask = nx_gmask; s_gmld = nx_gmld; }
                if constexpr (LB) { s_gbit = nx_gbit; s_mbit = nx_mbit; }
            };
            // SIDE segment next (and this run is its predecessor's last): its weights are requested NOW, by loads the compiler
            // does not see, so that they fly under this segment's epilogue and barrier
            f32x4 sdw[2][NT];
            bool side_next = false;
            if constexpr (K4) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int t = 0; t < NT; ++t) sdw[s2][t] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (NX.type == NS_SIDE && si + 1 < nseg && (s_type != NS_WIDE || pass + 1 == s_passes)) {
                    side_next = true;
                    const char* const sb = reinterpret_cast<const char*>(a.packed + NX.side_off) + (size_t)wave * NX.steps * NS_STEP_B + voff;
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(sdw[0][t]) : "v"(sb + t * 1024) : "memory");
                    if (NX.steps > 1) {
#pragma unroll
                        for (int t = 0; t < NT; ++t)
                            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(sdw[1][t]) : "v"(sb + NS_STEP_B + t * 1024) : "memory");
                    }
                }
            }
            bool seg_done = true;
#ifdef NS_STAMPS_FINE
            const bool fine = si >= NS_STAMPS_FINE && si < NS_STAMPS_FINE + 4;
            if (fine) NS_STAMP();                  // run end reached (last MFMA issued)
#endif
            // result quads: fin[q][e] is (row, column) = SM ? (4 q + e, lane) : (4 kq + e, 16 q + li) of the wave's 64 columns
            if constexpr (SM) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) acc[q] = (acc[4 * q] + acc[4 * q + 1]) + (acc[4 * q + 2] + acc[4 * q + 3]);
            }
#define fin acc
#define q_row(q, e) (SM ? 4 * (q) + (e) : 4 * kq + (e))
#define q_col(q) (SM ? lane : 16 * (q) + li)
            if (s_type == NS_WIDE) {
                float* const nxt = act + (P ^ 1) * ABUF + s_dst + (s_zext < 0 ? 64 * (pass ? 15 - wave : wave) : 512 * pass + 64 * wave);
                // STORE == 3, last network layer: the targets and the per-column constants of delta, fetched by inline-asm
                // loads the compiler does not count (a visible load in this loop body would turn its counted vmcnt waits
                // for the weight ring into vmcnt(0) in EVERY step); one explicit wait for all of them
                float ty[NQ][4];                        // STORE == 3: normalised targets; STORE == 2: gates
                if constexpr (STORE == 3) {
                    if (si == nlast) {
#pragma unroll
                        for (int t = 0; t < NQ; ++t) {
                            const int dc = min(512 * pass + 64 * wave + q_col(t), nout - 1);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float* py = a.t_Y + (size_t)lsrc[q_row(t, e)] * a.t_ldy + dc;
                                asm volatile("global_load_dword %0, %1, off" : "=v"(ty[t][e]) : "v"(py) : "memory");
                            }
                        }
#pragma unroll
                        for (int t = 0; t < NQ; ++t)
                            asm volatile("s_waitcnt vmcnt(0)" : "+v"(ty[t][0]), "+v"(ty[t][1]), "+v"(ty[t][2]), "+v"(ty[t][3]) :: "memory");
                    }
                }
                unsigned long long gw[SM ? NQ : 1][4];      // G2: the 64 sign bits of this wave's columns, per result row
                if constexpr (LB) {
                    if (s_mbit >= 0) {
                        const int cw = 512 * pass + 64 * wave, w0 = (s_mbit + cw) >> 5;
                        const bool mine = cw < ((s_gn + 63) & ~63);             // (past the tensor: padding columns, their gradients are zero)
#pragma unroll
                        for (int t = 0; t < (SM ? NQ : 1); ++t)
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                gw[t][e] = mine ? *reinterpret_cast<const unsigned long long*>(lbits + (SM ? 4 * t + e : 4 * kq + e) * nbw + w0) : 0ull;
                    }
                }
                if constexpr (DXE && !LB) {
                    // the gates (stored forward activations), by loads the compiler does not count -- a visible load in this
                    // loop body makes every step's wait for the weight ring a vmcnt(0) -- with one explicit wait
                    if (s_gmask) {
#pragma unroll
                        for (int t = 0; t < NQ; ++t) {
                            const int mc = min(512 * pass + 64 * wave + q_col(t), s_gn - 1);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float* pg = s_gmask + (size_t)min(row0 + q_row(t, e), a.B - 1) * s_gmld + mc;
                                if constexpr (GRAD)     // written earlier in this very launch: served by the L2, not this CU's L1
                                    asm volatile("global_load_dword %0, %1, off sc1" : "=v"(ty[t][e]) : "v"(pg) : "memory");
                                else
                                    asm volatile("global_load_dword %0, %1, off" : "=v"(ty[t][e]) : "v"(pg) : "memory");
                            }
                        }
#pragma unroll
                        for (int t = 0; t < NQ; ++t)
                            asm volatile("s_waitcnt vmcnt(0)" : "+v"(ty[t][0]), "+v"(ty[t][1]), "+v"(ty[t][2]), "+v"(ty[t][3]) :: "memory");
                    }
                }
                unsigned mbits = 0xFFFFu;
                if constexpr (GRAD) {
                    if (s_mapply) mbits = lmask[(s_mapply - 1 + pass) * (64 * NW) + threadIdx.x];   // sign bits of this very (row, col)
                    if (s_mstore) {
                        unsigned m = 0;
#pragma unroll
                        for (int t = 0; t < NQ; ++t)
#pragma unroll
                            for (int e = 0; e < 4; ++e) m |= (fin[t][e] > 0.f ? 1u : 0u) << (4 * t + e);
                        lmask[(s_mstore - 1 + pass) * (64 * NW) + threadIdx.x] = m;
                    }
                }
#pragma unroll
                for (int t = 0; t < NQ; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {  // 16x16x4 C/D layout: col = lane&15, row = 4*(lane>>4) + e; 4x4x1: col = lane, row = 4 t + e
                        float v = fin[t][e];
                        if constexpr (GRAD) v = ((mbits >> (4 * t + e)) & 1u) ? v : 0.f;
                        v = s_relu ? fmaxf(v, 0.f) : v;
                        if constexpr (LB) {
                            if (s_mbit >= 0 && !((gw[SM ? t : 0][e] >> (SM ? lane : 16 * t + li)) & 1ull)) v = 0.f;
                        } else if constexpr (DXE) {
                            if (s_gmask && !(ty[t][e] > 0.f)) v = 0.f;
                        }
                        float v_lds = v;
                        if constexpr (STORE == 3) {
                            if (si == nlast) {                 // the network's last layer: delta replaces pred in LDS
                                const int dc = 512 * pass + 64 * wave + q_col(t);
                                const float yn = ty[t][e];
                                v_lds = dc < nout ? (isnan(yn) ? -0.f : (yn - v) + 0.f) : 0.f;   // -0: "masked", read back by the finish
                            }
                        }
                        nxt[q_row(t, e) * LD + q_col(t)] = v_lds;
                        if constexpr (LB) {
                            if (s_gbit >= 0) {                 // the sign of this activation, for the gate of its gradient
                                const unsigned long long bb = __ballot(v > 0.f);
                                const int cw = 512 * pass + 64 * wave;          // (a wave whose 64 columns lie past the tensor's
                                const bool mine = cw < ((s_gn + 63) & ~63);     //  bit range writes nothing: the next row's bits, or the
                                const int c0 = s_gbit + cw;                     //  neighbouring workgroup's LDS, sit there)
                                if constexpr (SM) {            // lane = column: 64 columns of row 4 t + e
                                    if (lane == 0 && mine) *reinterpret_cast<unsigned long long*>(lbits + (4 * t + e) * nbw + (c0 >> 5)) = bb;
                                } else {                       // 16 columns of tile t for each of the rows 4 kq + e
                                    if (li == 0 && mine) reinterpret_cast<unsigned short*>(lbits + (4 * kq + e) * nbw)[(c0 + 16 * t) >> 4] = (unsigned short)(bb >> (16 * kq));
                                }
                            }
                        }
                        if constexpr (STORE && !G2) {
                            const int grow_ = row0 + q_row(t, e), gcol = 512 * pass + 64 * wave + q_col(t);
                            if (s_gout && grow_ < a.B && gcol < s_gn) {
                                float vs = v;                      // the network's last output carries the column affine
                                if constexpr (STORE == 1)
                                    if (si == nseg - 1) vs = vs * (a.cscale ? a.cscale[gcol] : 1.f) + (a.cshift ? a.cshift[gcol] : 0.f);
                                gstore(s_gout + (size_t)grow_ * s_gld + gcol, vs);
                            }
                        }
                    }
#ifdef NS_STAMPS_FINE
                if (fine) NS_STAMP();              // epilogue done (LDS writes and stores issued)
#endif
                if (++pass == s_passes) {
                    lds_barrier();
                    NS_STAMP();
                    P ^= 1; pass = 0; ++si;
                    take_next();
                } else {
                    kleft = s_zext > 0 ? s_zext : cur_steps;    // (a short second pass: NsSeg::zext)
                    seg_done = false;
                }
            } else {
                // [8 waves][ROWS][64 cols]; the column is swizzled per row so that the reduce below reads without bank
                // conflicts: col ^= 16*(row>>2) (16 rows), col ^= 32*(row&1) (4x4x1 engines)
                float* const part = act + (P ^ 1) * ABUF;
                constexpr int PW = ROWS * 64;
#pragma unroll
                for (int t = 0; t < NQ; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int rr = q_row(t, e);
                        part[wave * PW + rr * 64 + (q_col(t) ^ (SM ? 32 * (rr & 1) : 16 * kq))] = fin[t][e];
                    }
#ifdef NS_STAMPS_FINE
                if (fine) NS_STAMP();              // partials written (issued)
#endif
                lds_barrier();
#ifdef NS_STAMPS_FINE
                if (fine) NS_STAMP();              // first barrier passed
#endif
                // thread (row sr, lane sc0 of RGS): columns sc0, sc0+RGS, ...; wave of (K part kp, group cg) = kp*ncg + cg.
                // 16 rows: the prologue's 32 threads per row; 8 / 4 rows: ALL 512 threads, 64 / 128 per row (with 32 per row
                // three quarters of a 4-row workgroup watched the other quarter reduce)
                constexpr int RGS = SM ? 64 * NW / ROWS : RG;
                const int sr = SM ? tid / RGS : pr, sc0 = SM ? tid % RGS : pc0;
                const bool srow = SM ? true : prow;
                float* const cur = act + P * ABUF + sr * LD + s_dst;
                const int ncol = 64 << s_ncgl, nkp = NW >> s_ncgl;
                const int sw = SM ? 32 * (sr & 1) : 16 * (sr >> 2);
                constexpr int NGJ = 256 / RGS;          // (SPLIT outputs are <= 256 columns: 8 / 4 / 2 per thread)
                float sg[NGJ];
                if constexpr (DXE && !LB) {
                    if (s_gmask) {
#pragma unroll
                        for (int j = 0; j < NGJ; ++j) {
                            const float* pg = s_gmask + (size_t)min(row0 + sr, a.B - 1) * s_gmld + min(sc0 + RGS * j, s_gn - 1);
                            if constexpr (GRAD)
                                asm volatile("global_load_dword %0, %1, off sc1" : "=v"(sg[j]) : "v"(pg) : "memory");
                            else
                                asm volatile("global_load_dword %0, %1, off" : "=v"(sg[j]) : "v"(pg) : "memory");
                        }
#pragma unroll
                        for (int j = 0; j < NGJ; ++j) asm volatile("s_waitcnt vmcnt(0)" : "+v"(sg[j]) :: "memory");
                    }
                }
                int gj = 0;
                if constexpr ((!LB && !DXE && !STORE) || (LB && !SM)) {
                    // serving (and the one-launch gradient on the 16-row engine): FOUR adjacent columns per thread and trip, as 16-byte LDS accesses (the swizzle moves whole groups of
                    // 16 / 32 columns, the biases start at multiples of 64): a quarter of the LDS instructions of the column-per-
                    // trip loop below, and all K parts of a trip in flight -- 256 output columns were 8 trips of ~130 cycles
                    // each behind the barrier.  The sums in the order of the loop below.
                    auto reduce4 = [&](auto NKc) {
                        constexpr int NK = decltype(NKc)::value;
                        for (int c = 4 * sc0; c < (srow ? s_zext : 0); c += 4 * RGS) {
                            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
                            if (c < ncol) {
                                const float* src = part + (c >> 6) * PW + sr * 64 + ((c & 63) ^ sw);
                                constexpr int CH = LB && NK > 4 ? 4 : NK;        // K parts in flight (the gradient launch has no 32 registers to spare)
                                const f32x4 b = *reinterpret_cast<const f32x4*>(lbias + s_bias + c);
#pragma unroll
                                for (int k0 = 0; k0 < NK; k0 += CH) {
                                    f32x4 x[CH];
#pragma unroll
                                    for (int kp = 0; kp < CH; ++kp) x[kp] = *reinterpret_cast<const f32x4*>(src + ((k0 + kp) << s_ncgl) * PW);
#pragma unroll
                                    for (int kp = 0; kp < CH; ++kp) v += x[kp];
                                }
                                v += b;
                                if (s_relu) v = f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
                            }
                            if constexpr (LB) {
                                // the gate: four sign bits of the tensor this gradient belongs to (bit columns s_mbit + c ..; c & 31 <= 28)
                                const bool mine = c < ((s_gn + 63) & ~63);
                                if (s_mbit >= 0) {
                                    const unsigned nib = mine ? (lbits[sr * nbw + ((s_mbit + c) >> 5)] >> (c & 31)) & 0xFu : 0u;
#pragma unroll
                                    for (int e = 0; e < 4; ++e) v[e] = ((nib >> e) & 1u) ? v[e] : 0.f;
                                }
                                *reinterpret_cast<f32x4*>(cur + c) = v;
                                if (s_gbit >= 0) {
                                    // the signs of these activations: the eight threads of a 32-column word OR their nibbles together
                                    unsigned w32 = ((v[0] > 0.f ? 1u : 0u) | (v[1] > 0.f ? 2u : 0u) | (v[2] > 0.f ? 4u : 0u) | (v[3] > 0.f ? 8u : 0u)) << (c & 31);
                                    w32 |= __shfl_xor(w32, 1, 64); w32 |= __shfl_xor(w32, 2, 64); w32 |= __shfl_xor(w32, 4, 64);
                                    if ((sc0 & 7) == 0 && mine) lbits[sr * nbw + ((s_gbit + c) >> 5)] = w32;
                                }
                            } else {
                                *reinterpret_cast<f32x4*>(cur + c) = v;
                            }
                        }
                    };
                    if (nkp == NW) reduce4(std::integral_constant<int, NW>{});
                    else if (nkp == NW / 2) reduce4(std::integral_constant<int, NW / 2>{});
                    else reduce4(std::integral_constant<int, 2>{});
                } else
                for (int c = sc0; c < (srow ? s_zext : 0); c += RGS, ++gj) {
                    float v = 0.f;
                    if (c < ncol) {
                        const float* src = part + (c >> 6) * PW + sr * 64 + ((c & 63) ^ sw);
                        // the partials of this column's K parts, all reads in flight (8, 4 or 2 of them: the same sums in the
                        // same order as one loop over kp < nkp)
                        if (nkp == NW) {
                            float x[NW];
#pragma unroll
                            for (int kp = 0; kp < NW; ++kp) x[kp] = src[(kp << s_ncgl) * PW];
#pragma unroll
                            for (int kp = 0; kp < NW; ++kp) v += x[kp];
                        } else if (nkp == NW / 2) {
                            float x[NW / 2];
#pragma unroll
                            for (int kp = 0; kp < NW / 2; ++kp) x[kp] = src[(kp << s_ncgl) * PW];
#pragma unroll
                            for (int kp = 0; kp < NW / 2; ++kp) v += x[kp];
                        } else {
                            const float x0 = src[0], x1 = src[(1 << s_ncgl) * PW];
                            v += x0; v += x1;
                        }
                        v += lbias[s_bias + c];
                        if (s_relu) v = fmaxf(v, 0.f);
                        if constexpr (LB) {
                            if (s_mbit >= 0 && !(c < ((s_gn + 63) & ~63) && ((lbits[sr * nbw + ((s_mbit + c) >> 5)] >> (c & 31)) & 1u))) v = 0.f;
                        } else if constexpr (DXE) {
                            float gv = 1.f;             // (dynamic register-array index: a select chain)
#pragma unroll
                            for (int j = 0; j < NGJ; ++j) gv = gj == j ? sg[j] : gv;
                            if (s_gmask && !(gv > 0.f)) v = 0.f;
                        }
                        if constexpr (STORE && !G2) {
                            if (s_gout && row0 + sr < a.B && c < s_gn) {
                                float vs = v;
                                if constexpr (STORE == 1)
                                    if (si == nseg - 1) vs = vs * (a.cscale ? a.cscale[c] : 1.f) + (a.cshift ? a.cshift[c] : 0.f);
                                gstore(s_gout + (size_t)(row0 + sr) * s_gld + c, vs);
                            }
                        }
                    }
                    cur[c] = v;
                    if constexpr (LB) {
                        if (s_gbit >= 0) {                  // (every lane of a wave runs the same trips of this loop)
                            const unsigned long long bb = __ballot(v > 0.f);
                            const bool mine = c < ((s_gn + 63) & ~63);
                            if constexpr (SM) { if (lane == 0 && mine) *reinterpret_cast<unsigned long long*>(lbits + sr * nbw + ((s_gbit + c) >> 5)) = bb; }
                            else { if ((lane & 31) == 0 && mine) lbits[sr * nbw + ((s_gbit + c) >> 5)] = (unsigned)(bb >> (lane & 32)); }
                        }
                    }
                }
                if constexpr (STORE == 3) {
                    if (si == nlast) {
                        // the network's last layer (nout <= 256): thread (row sr, lane sc0) turns its own columns
                        // sc0 + RGS j of pred into delta, in place (asm loads: see the WIDE epilogue)
                        constexpr int NJ = NGJ;
                        const int ysrc = lsrc[sr];
                        float sy[NJ];
#pragma unroll
                        for (int j = 0; j < NJ; ++j) {
                            const int c = min(sc0 + RGS * j, nout - 1);
                            asm volatile("global_load_dword %0, %1, off" : "=v"(sy[j]) : "v"(a.t_Y + (size_t)ysrc * a.t_ldy + c) : "memory");
                        }
#pragma unroll
                        for (int j = 0; j < NJ; ++j) asm volatile("s_waitcnt vmcnt(0)" : "+v"(sy[j]) :: "memory");
#pragma unroll
                        for (int j = 0; j < NJ; ++j) {
                            const int c = sc0 + RGS * j;
                            if (c < nout && srow) cur[c] = isnan(sy[j]) ? -0.f : (sy[j] - cur[c]) + 0.f;
                        }
                    }
                }
#ifdef NS_STAMPS_FINE
                if (fine) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); NS_STAMP(); }   // reduce done
#endif
                lds_barrier();
                NS_STAMP();
                ++si;
                take_next();
            }
            if constexpr (GRAD) {
                if (seg_done && si == a.nseg_f) {   // (seg_done: not again after a pass of the first backward segment)
                    if constexpr (DXE && !LB)       // the forward activations every wave stored are in memory before any gate load
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if constexpr (TRB) {
                        // ---- turnaround of a training step = the loss finish: delta and U = delta Cinv sit in LDS (U at column
                        // u_col of buffer P, or at column 0 with delta in the other buffer).  loss_b = delta . U / den
                        // (util.py:1086-1088); d loss / d pred = -2 U inv_batch / den, zero where delta was masked, goes to
                        // memory (the last layer's parameter gradient reads it) AND over delta / U in LDS: the input rows of the
                        // first backward segment (their padding columns hold the zeros the forward left there).
                        {
                            const float* const F = act + P * ABUF + pr * LD;
                            const bool rok = prow && row0 + pr < a.B;
                            const float* const Dv = a.u_same ? F : act + (P ^ 1) * ABUF + pr * LD;
                            const float* const Uv = a.u_same ? F + a.u_col : F;
                            float chi = 0.f;
                            for (int c = pc0; c < nout; c += RG) chi += Dv[c] * Uv[c];
#pragma unroll
                            for (int o = RG / 2; o >= 1; o >>= 1) chi += __shfl_xor(chi, o, 64);
                            if (rok && pc0 == 0) gstore(a.t_loss_rows + row0 + pr, chi / lden[pr]);
                        }
                        lds_barrier();                  // every row's chi is taken before delta is overwritten (u_same)
                        {
                            constexpr int TPR = 64 * NW / ROWS;
                            const int fr = tid / TPR, fc = tid % TPR;
                            float* const Fr = act + P * ABUF + fr * LD;
                            const float* const Dr = a.u_same ? Fr : act + (P ^ 1) * ABUF + fr * LD;
                            const float* const Ur = a.u_same ? Fr + a.u_col : Fr;
                            const float dr = lden[fr];
                            const bool rowok = row0 + fr < a.B;
                            for (int c = fc; c < a.t_lddp; c += TPR) {
                                float g = 0.f;
                                if (c < nout && rowok) {
                                    const bool masked = __float_as_uint(Dr[c]) == 0x80000000u;
                                    g = masked ? 0.f : (-2.f * Ur[c]) * a.t_inv_batch / dr;
                                }
                                if (rowok) gstore(a.t_dP + (size_t)(row0 + fr) * a.t_lddp + c, g);
                                if (c < nout) Fr[c] = g;
                            }
                        }
                        lds_barrier();
                    } else {
                    // ---- turnaround: the output rows (bias added) sit in buffer P.  lnP as in the finish, and
                    // d lnP / d out = -(d w) gscale / T written over them: the input of the first backward segment
                    float* const F = act + P * ABUF + pr * LD;
                    float chi = 0.f;
#pragma unroll
                    for (int i = 0; i < FIN; ++i) {
                        const int c = pc0 + i * RG;
                        if (c < nout && prow) {
                            const float d = F[c] * fcs[i] + fct[i];
                            chi += (d * fw[i]) * d;
                            F[c] = -(d * fw[i]) * fgs[i] / a.T;
                        }
                    }
#pragma unroll
                    for (int o = RG / 2; o >= 1; o >>= 1) chi += __shfl_xor(chi, o, 64);
                    lnp_grad = (-0.5f * chi) / a.T + (-0.5f * zz);
                    lds_barrier();
                    }
                }
            }
#undef fin
#undef q_row
#undef q_col
            if (seg_done && si < nseg && s_x0col > 0) {
                // input skip: the kept input rows go behind this segment's regular input (one GEMM over [h ; x0])
                if (prow) {
                    float* const dstp = act + P * ABUF + pr * LD + s_x0col;
#pragma unroll
                    for (int i = 0; i < ZPRE; ++i)
                        if (pc0 + i * RG < s_x0n) dstp[pc0 + i * RG] = lx0[pr * 64 + pc0 + i * RG];
                }
                lds_barrier();
            }
            if constexpr (K4) {
                if (side_next && seg_done) {
                    // ---- SIDE run: the segment now in s_* (taken above), whole, right here.  Wave w owns the k range
                    // [w kslice, (w + 1) kslice) (ncg = 1); a step covers kc chunks of 16 k: tile t of the weights is
                    // (column tile t % (4 / kc), chunk t / (4 / kc)).
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                    const bool kc4 = s_kcl == 2, kc1 = s_kcl == 0;
                    const uint32_t abase = act_lds + 4u * (uint32_t)(P * ABUF + li * LD + 4 * kq + wave * s_kslice);
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        if (s2 < s_steps) {
                            const uint32_t as = abase + (uint32_t)s2 * (64u << s_kcl);
                            f32x4 af0, af1, af2 = f32x4{0.f, 0.f, 0.f, 0.f}, af3 = f32x4{0.f, 0.f, 0.f, 0.f};
                            asm volatile("ds_read_b128 %0, %1" : "=v"(af0) : "v"(as) : "memory");
                            af1 = f32x4{0.f, 0.f, 0.f, 0.f};
                            if (!kc1) asm volatile("ds_read_b128 %0, %1 offset:64" : "=v"(af1) : "v"(as) : "memory");
                            if (kc4) {
                                asm volatile("ds_read_b128 %0, %1 offset:128" : "=v"(af2) : "v"(as) : "memory");
                                asm volatile("ds_read_b128 %0, %1 offset:192" : "=v"(af3) : "v"(as) : "memory");
                            }
                            if (s2 == 0)
                                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                                             : "+v"(af0), "+v"(af1), "+v"(af2), "+v"(af3), "+v"(sdw[0][0]), "+v"(sdw[0][1]), "+v"(sdw[0][2]),
                                               "+v"(sdw[0][3]), "+v"(sdw[1][0]), "+v"(sdw[1][1]), "+v"(sdw[1][2]), "+v"(sdw[1][3]) :: "memory");
                            else
                                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af0), "+v"(af1), "+v"(af2), "+v"(af3) :: "memory");
                            // tile t multiplies chunk t (kc = 4), t >> 1 (kc = 2) or the one chunk (kc = 1)
                            const f32x4 f1 = kc4 ? af1 : af0, f2 = kc4 ? af2 : kc1 ? af0 : af1, f3 = kc4 ? af3 : kc1 ? af0 : af1;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af0[e], sdw[s2][0][e], acc[0], 0, 0, 0);
                                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(f1[e], sdw[s2][1][e], acc[1], 0, 0, 0);
                            }
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(f2[e], sdw[s2][2][e], acc[2], 0, 0, 0);
                                acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(f3[e], sdw[s2][3][e], acc[3], 0, 0, 0);
                            }
                        }
                    }
                    int treal = NT;
                    if (kc4) { acc[0] = (acc[0] + acc[1]) + (acc[2] + acc[3]); treal = 1; }
                    else if (!kc1) { acc[0] = acc[0] + acc[2]; acc[1] = acc[1] + acc[3]; treal = 2; }
                    float* const part = act + (P ^ 1) * ABUF;       // [8 waves][16 rows][64]: the SPLIT layout and swizzle
                    constexpr int PW = ROWS * 64;
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        if (t < treal) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) part[wave * PW + (4 * kq + e) * 64 + ((16 * t + li) ^ (16 * kq))] = acc[t][e];
                        }
#ifdef NS_STAMPS_FINE
                    if (fine) NS_STAMP();          // SIDE: MFMAs issued, partials written
#endif
                    lds_barrier();
#ifdef NS_STAMPS_FINE
                    if (fine) NS_STAMP();          // SIDE: first barrier passed
#endif
                    {                                  // four adjacent columns per thread, 16-byte accesses: the SPLIT reduce's form
                        float* const cur = act + P * ABUF + pr * LD + s_dst;
                        const int ncol = 16 * treal, sw = 16 * (pr >> 2);
                        for (int c = 4 * pc0; c < s_zext; c += 4 * RG) {
                            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
                            if (c < ncol) {
                                const float* src = part + pr * 64 + (c ^ sw);
                                constexpr int CH = LB ? 4 : NW;
                                const f32x4 b = *reinterpret_cast<const f32x4*>(lbias + s_bias + c);
#pragma unroll
                                for (int k0 = 0; k0 < NW; k0 += CH) {
                                    f32x4 x[CH];
#pragma unroll
                                    for (int kp = 0; kp < CH; ++kp) x[kp] = *reinterpret_cast<const f32x4*>(src + (k0 + kp) * PW);
#pragma unroll
                                    for (int kp = 0; kp < CH; ++kp) v += x[kp];
                                }
                                v += b;
                                if (s_relu) v = f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
                            }
                            if constexpr (LB) {         // a backward SIDE segment (d/dh) is gated by the sign bits of h; a forward one records them
                                const bool mine = c < ((s_gn + 63) & ~63);
                                if (s_mbit >= 0) {
                                    const unsigned nib = mine ? (lbits[pr * nbw + ((s_mbit + c) >> 5)] >> (c & 31)) & 0xFu : 0u;
#pragma unroll
                                    for (int e = 0; e < 4; ++e) v[e] = ((nib >> e) & 1u) ? v[e] : 0.f;
                                }
                                *reinterpret_cast<f32x4*>(cur + c) = v;
                                if (s_gbit >= 0) {
                                    unsigned w32 = ((v[0] > 0.f ? 1u : 0u) | (v[1] > 0.f ? 2u : 0u) | (v[2] > 0.f ? 4u : 0u) | (v[3] > 0.f ? 8u : 0u)) << (c & 31);
                                    w32 |= __shfl_xor(w32, 1, 64); w32 |= __shfl_xor(w32, 2, 64); w32 |= __shfl_xor(w32, 4, 64);
                                    if ((pc0 & 7) == 0 && mine) lbits[pr * nbw + ((s_gbit + c) >> 5)] = w32;
                                }
                            } else {
                                *reinterpret_cast<f32x4*>(cur + c) = v;
                            }
                        }
                    }
#ifdef NS_STAMPS_FINE
                    if (fine) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); NS_STAMP(); }   // SIDE: reduce done
#endif
                    lds_barrier();
                    NS_STAMP();
                    ++si;
                    const int n2i = __builtin_amdgcn_readfirstlane(min(si, nseg - 1));
                    const NsSeg N2 = ka->seg[n2i];
                    take_seg(N2);
                    if constexpr (LB) { s_gbit = ka->gbit[n2i]; s_mbit = ka->mbit[n2i]; s_gn = ka->gn[n2i]; }
                }
            }
            if (si < nseg) {
                begin_run();
                a_read(Aq[(U + 1) & 1]);           // replaces the speculative fragment
            }
#ifdef NS_STAMPS_FINE
            if (fine) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); NS_STAMP(); }   // next run ready to issue
#endif
        }
    };
    using T_ = std::true_type; using F_ = std::false_type;
#define NS_STEP(U, RF) if constexpr (U < R) step(std::integral_constant<int, U>{}, RF{});
    const int ngroups = a.G / R, rem = a.G - ngroups * R;
#pragma unroll 1
    for (int it = 0; it < ngroups; ++it) {
        NS_STEP(0, T_) NS_STEP(1, T_) NS_STEP(2, T_) NS_STEP(3, T_) NS_STEP(4, T_) NS_STEP(5, T_) NS_STEP(6, T_) NS_STEP(7, T_)
    }
#define NS_TAIL(U) if constexpr (U < R - 1) { if (rem > U) step(std::integral_constant<int, U>{}, F_{}); }
    NS_TAIL(0) NS_TAIL(1) NS_TAIL(2) NS_TAIL(3) NS_TAIL(4) NS_TAIL(5) NS_TAIL(6)
#undef NS_STEP
#undef NS_TAIL
    // The last speculative A read.  Both fragments are operands of the wait: the compiler does not know that the
    // inline-asm ds_read lands later, and a fragment nobody reads again would otherwise be dead at once -- its
    // registers could be handed to an accumulator of the final step, which the returning LDS data then overwrites.
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(Aq[0]), "+v"(Aq[1]) :: "memory");
    NS_STAMP();

    if constexpr (((STORE == 1 || STORE == 2) && !GRAD) || TRB) { NS_STAMPS_FLUSH(); return; }   // every output is in global memory already
    if constexpr (STORE == 3) {
        // ---- 5 (loss).  delta and U = delta Cinv sit in LDS (as d and U of the dense serving program): chi2 = delta . U,
        // loss_b = chi2 / den (util.py:1086-1088), d loss / d pred = -2 U inv_batch / den, zero where delta was masked
        const float* const F = act + P * ABUF + pr * LD;
        const bool rok = prow && row0 + pr < a.B;
        const float* const Dv = a.u_same ? F : act + (P ^ 1) * ABUF + pr * LD;
        const float* const Uv = a.u_same ? F + a.u_col : F;
        float chi = 0.f;
        for (int c = pc0; c < nout; c += RG) chi += Dv[c] * Uv[c];
#pragma unroll
        for (int o = RG / 2; o >= 1; o >>= 1) chi += __shfl_xor(chi, o, 64);
        if (rok && pc0 == 0) a.t_loss_rows[row0 + pr] = chi / lden[pr];
        {
            // d loss / d pred by ALL threads of the workgroup, 512 / ROWS per row (the row's 32 threads alone walked a
            // 457-wide row in 15 rounds while three quarters of the workgroup waited: 8 k cycles at the kernel's tail)
            constexpr int TPR = 64 * NW / ROWS;
            const int fr = tid / TPR, fc = tid % TPR;
            if (row0 + fr < a.B) {
                const float* const Fr = act + P * ABUF + fr * LD;
                const float* const Dr = a.u_same ? Fr : act + (P ^ 1) * ABUF + fr * LD;
                const float* const Ur = a.u_same ? Fr + a.u_col : Fr;
                const float dr = lden[fr];
                for (int c = fc; c < a.t_lddp; c += TPR) {
                    float g = 0.f;
                    if (c < nout) {
                        const bool masked = __float_as_uint(Dr[c]) == 0x80000000u;
                        g = masked ? 0.f : (-2.f * Ur[c]) * a.t_inv_batch / dr;
                    }
                    a.t_dP[(size_t)(row0 + fr) * a.t_lddp + c] = g;
                }
            }
        }
        NS_STAMP();
        NS_STAMPS_FLUSH();
        return;
    }
    // ---- 5 (GRAD). d lnP / d x sits in buffer P: the derivative of the input transform and of the prior map
    // (util.py:339-347, 483-497), minus z for the Gaussian prior term; lnP from the turnaround
    if constexpr (GRAD) {
        const float* const F = act + P * ABUF + pr * LD;
        const bool rok = prow && row0 + pr < a.B;
        if (rok) {
#pragma unroll
            for (int j = 0; j < ZPRE; ++j) {
                const int c = pc0 + j * RG;
                if (c < nin) {
                    float g = F[c] / zxs[j];
                    if (a.lg && zlg[j]) g = g / (theta[j] * 2.30258509299404568f);
                    const float z = zr[j];
                    const float dth = zfl[j] ? za2[j] * (expf(-0.5f * z * z) * 0.398942280401432678f) : za2[j];
                    const float gz = g * dth - z;
                    a.Gout[(size_t)(row0 + pr) * a.ldg + c] = gz;
                    if (a.hm_p) {
                        // the leapfrog's kick with this gradient and the drift to the next position (hmc_kick_drift_kernel's
                        // arithmetic: the launch between two gradient evaluations it replaces was 4.5 us of nothing)
                        float* const pp = a.hm_p + (size_t)(row0 + pr) * a.hm_ldp + c;
                        float pm = *pp;
                        if (a.hm_ek != 0.f) { pm += a.hm_ek * gz; *pp = pm; }
                        if (a.hm_ed != 0.f) a.hm_q[(size_t)(row0 + pr) * a.ldz + c] = z + a.hm_ed * (pm / a.hm_mass[c]);
                    }
                }
            }
            if (pc0 == 0) a.lnP[row0 + pr] = isnan(lnp_grad) ? -INFINITY : lnp_grad;
        }
        if constexpr (STORE == 2) { NS_STAMP(); NS_STAMPS_FLUSH(); }     // (diagnostic build; the MLP-only GRAD keeps its masks where the stamps would sit)
        return;
    }
    // ---- 5. output rows are in buffer P (bias added, no ReLU): output transform, d, log-likelihood
    {
        const float* const F = act + P * ABUF + pr * LD;
        const bool rok = prow && row0 + pr < a.B;
        float chi = 0.f;
        auto column = [&](int c, float cs, float ct, float ww) {
            float d = F[c] * cs + ct;
            if (a.cpost) d = expf(d) * a.cpost[c] + a.cshift2[c];
            if (a.D && rok) a.D[(size_t)(row0 + pr) * a.ldd + c] = d;
            chi += (d * ww) * d;
        };
        if (a.dense) {
            // the last segment multiplied d (output map folded into the last layer) by the dense inverse covariance
            const float* const Dv = a.u_same ? F : act + (P ^ 1) * ABUF + pr * LD;
            const float* const U = a.u_same ? F + a.u_col : F;
            const bool fac = a.dense == 2;      // the segment multiplied by L (S = L L^T): chi2 = |d L|^2, a sum of squares
            for (int c = pc0; c < nout; c += RG) {
                const float d = Dv[c];
                if (a.D && rok) a.D[(size_t)(row0 + pr) * a.ldd + c] = d;
                const float u = U[c];
                chi += (fac ? u : d) * u;
            }
        } else {
#pragma unroll
            for (int i = 0; i < FIN; ++i)
                if (pc0 + i * RG < nout) column(pc0 + i * RG, fcs[i], fct[i], fw[i]);
            for (int c = pc0 + FIN * RG; c < nout; c += RG)        // wide outputs: constants straight from memory
                column(c, a.cscale ? a.cscale[c] : 1.f, a.cshift ? a.cshift[c] : 0.f, a.w ? a.w[c] : 0.f);
        }
#pragma unroll
        for (int o = RG / 2; o >= 1; o >>= 1) chi += __shfl_xor(chi, o, 64);
        float lnp_new = (-0.5f * chi) / a.T + (-0.5f * zz);
        lnp_new = isnan(lnp_new) ? -INFINITY : lnp_new;
        if constexpr (MOVE == 2) {
            if (a.lnP && pc0 == 0 && prow && row0 + pr < mv_rows) a.lnP[a.mv_C ? grow : row0 + pr] = lnp_new;
        } else {
            if (a.lnP && (a.w || a.dense) && pc0 == 0 && rok) a.lnP[row0 + pr] = lnp_new;
        }
        if constexpr (MOVE == 1) {
            // Metropolis test of the stretch move (linna_stretch_accept); every lane of the row agrees
            const bool mv_acc = rok && mv_factor + lnp_new - mv_lnp_old > mv_logu;
            if (mv_acc) {
#pragma unroll
                for (int j = 0; j < ZPRE; ++j)
                    if (pc0 + j * RG < nin) a.mv_coords[(size_t)mv_wk * a.mv_ldc + pc0 + j * RG] = zr[j];
                if (pc0 == 0) {
                    a.mv_logp[mv_wk] = lnp_new;
                    if (a.mv_naccept) a.mv_naccept[mv_wk] += 1;
                }
            }
            if (a.mv_chain && rok) {
                // the walker's position after this iteration goes straight into the chain block (a walker moves in ONE of the
                // two half steps of an iteration: the two launches together fill the row)
#pragma unroll
                for (int j = 0; j < ZPRE; ++j)
                    if (pc0 + j * RG < nin)
                        a.mv_chain[(size_t)mv_wk * nin + pc0 + j * RG] = mv_acc ? zr[j] : a.mv_coords[(size_t)mv_wk * a.mv_ldc + pc0 + j * RG];
                if (pc0 == 0) a.mv_lps[mv_wk] = mv_acc ? lnp_new : mv_lnp_old;
            }
        }
        if (a.TH && rok) {
#pragma unroll
            for (int j = 0; j < ZPRE; ++j)
                if (pc0 + j * RG < nin) a.TH[(size_t)(row0 + pr) * a.ldt + pc0 + j * RG] = theta[j];
            for (int c = pc0 + ZPRE * RG; c < nin; c += RG)     // wide inputs: theta recomputed rather than kept
                a.TH[(size_t)(row0 + pr) * a.ldt + c] = ns_prior_theta(a.Z[(size_t)grow * a.ldz + c], a.is_flat[c], a.a1[c], a.a2[c]);
        }
    }
    NS_STAMP();
    NS_STAMPS_FLUSH();
#undef NS_STAMP
#undef NS_STAMPS_FLUSH
}

// ---------------------------------------------------------------------------- host side: the program
struct NsProgram {
    std::vector<NsPackSeg> pack;
    std::vector<NsSeg> seg;
    int G = 0, LD = 0, kpad0 = 0, nout = 0, bias_total = 0;
    int Gstride = 0, nseg_f = 0, mask_slots = 0;            // G: forward steps; Gstride: forward + backward steps
    size_t lds_bytes = 0, lds_bytes_grad = 0, packed_floats = 0;   // LDS of the 16-row engine (lds_for: any engine)
    int dense = 0, u_col = 0, u_same = 0;                   // dense inverse covariance appended as the last segment
    int x0_keep = 0;                                        // an input-skip segment copies the network input later
    size_t side_f4 = 0;                                     // 16-byte vectors of all SIDE blocks
    size_t lds_for(int rows, bool grad) const {
        size_t b = (size_t)(2 * rows * LD + ((bias_total + 3) & ~3)) * sizeof(float) + 128 + (x0_keep ? 4096 : 0);   // + [16] set rows, [16] den (STORE == 3), kept input rows
#ifdef NS_STAMPS
        b += NS_NW * 32 * 8;
#endif
        return b + (grad ? (size_t)mask_slots * 64 * NS_NW * sizeof(unsigned) : 0);
    }
    bool ok = false, grad_ok = false;                       // grad_ok: backward segments appended (ReLU MLPs)
    bool dxi_ok = false;                                    // the dX chain down to the input appended (NS_PROG_FWD_DXI)
    bool train_ok = false;                                  // forward + loss + dX chain (NS_PROG_TRAIN)
    std::vector<int> seg_op, seg_hidden;                    // forward segments: op index; 1 = the hidden h of a residual block
                                                            // (dX-chain program: 1 = d/dh of a residual block, else d/d(input) of the op)
};

static int ceil16(int k) { return (k + 15) & ~15; }
// steps of a segment in every wave's stream (a WIDE segment's second pass may be shorter: NsSeg::zext)
// zext < 0: the BALANCED triangular assignment of a 16-block lower-triangular factor -- wave w multiplies column block w
// (rows from 64 w) and then block 15 - w (rows from 64 (15 - w)): steps - 4 w and steps - 4 (15 - w) steps, 2 steps - 60 in
// every wave
static int ns_seg_steps(const NsSeg& s) {
    if (s.type == NS_WIDE && s.zext < 0) return 2 * s.steps - 60;
    return s.type == NS_WIDE && s.zext > 0 ? s.steps + (s.passes - 1) * s.zext : s.steps * s.passes;
}
static std::atomic<int> g_dense_tri{-1};
static int ns_dense_tri_resolved() {
    int m = g_dense_tri.load(std::memory_order_relaxed);
    if (m < 0) {
        const char* env = getenv("LINNA_DENSE_TRI");
        int v = env ? atoi(env) : 2;
        if (v < 0 || v > 2) v = 2;
        int expect = -1;
        g_dense_tri.compare_exchange_strong(expect, v);
        m = g_dense_tri.load(std::memory_order_relaxed);
    }
    return m;
}
int net_stream_dense_tri(int mode) {
    const int prev = ns_dense_tri_resolved();
    if (mode >= 0 && mode <= 2) g_dense_tri.store(mode);
    return prev;
}

// Translate the op list into segments; ok = false when something does not fit this kernel.
enum { NS_PROG_FWD = 0, NS_PROG_FWD_NOGRAD = 1, NS_PROG_DX = 2, NS_PROG_DX_INPUT = 3, NS_PROG_FWD_DENSE = 4, NS_PROG_FWD_DXI = 5,
       NS_PROG_TRAIN = 6 };   // TRAIN: forward + loss segment (FWD_DENSE with the loss's inverse covariance) followed by the dX chain down to op 1
static NsProgram ns_build_one(const linna_layer_t* layers, int nl, int in_size, int mode, const NsDense* dn = nullptr, bool k4 = false);
static NsProgram ns_build(const linna_layer_t* layers, int nl, int in_size, bool k4 = false) {
    NsProgram p = ns_build_one(layers, nl, in_size, NS_PROG_FWD, nullptr, k4);
    if (!p.ok) p = ns_build_one(layers, nl, in_size, NS_PROG_FWD_NOGRAD, nullptr, k4);      // the backward half may be what did not fit
    return p;
}
// mode NS_PROG_DX: the dX chain of a training step as a program of its own (linna_net_backward's order): the rows are
// d loss / d output, the segments run over the transposed weights from the last op down to op 1 (NS_PROG_DX_INPUT: op 0).
// A residual block y = relu(0.1 (W2 h + b2) + Ws x), h = relu(W1 x + b1) comes back as  dh = 0.1 (dy W2) [h > 0],
// written behind dy, and ONE GEMM over [dy ; dh] with [Ws^T | W1^T]; the gate of every output (the stored forward
// activation) is applied by the kernel's STORE == 2 epilogue.
// mode NS_PROG_FWD_DENSE: the forward program with the output map (d = raw * cscale + cshift) folded into the last
// layer's weights and bias, and the dense inverse covariance appended as one more bias-free segment U = d S -- the
// Gaussian log-likelihood (util.py:953-955) with a dense covariance then needs no GEMM launch of its own.
// k4: SPLIT segments of <= 32 columns become SIDE segments where they fit (see NsPackArgs): the kernel's K4 path, i.e. the
// serving instantiations of the 16-row engine only.
static NsProgram ns_build_one(const linna_layer_t* layers, int nl, int in_size, int mode, const NsDense* dn, bool k4) {
    NsProgram p;
    const bool allow_grad = mode == NS_PROG_FWD, dx_prog = mode == NS_PROG_DX || mode == NS_PROG_DX_INPUT;
    const bool train = mode == NS_PROG_TRAIN;
    if ((mode == NS_PROG_FWD_DENSE || train) && (!dn || !dn->S)) return p;
    if (nl < 1 || in_size < 1 || in_size > 256) return p;
    // 1. linear maps: [Wa | alpha Wb] over K = [Kapad ; Kb], N outputs, written to dst_col (same_buf: into the input's buffer)
    struct Lin { const float* Wa; int lda, Ka, Kapad; const float* Wb; int ldb, Kb; float alpha; const float* b; float bscale;
                 int N, relu, dst_col; bool same_buf; int transA = 0; int force_wide = 0; int mask_apply_of = -1; int op = -1;
                 int transB = 0; const float* rscale = nullptr; const float* rshift = nullptr;
                 const float* b2 = nullptr; float b2scale = 0.f; int x0_col = 0; };
    std::vector<Lin> lins;
    int width = in_size;
    // NS_PROG_FWD_DXI: the forward program followed by the dX chain down to the network input in ONE stream -- lnP and
    // d lnP / d z in one launch for ANY network (residual blocks, SPLIT segments): the forward segments store their
    // activations, the backward segments gate on them (the kernel's GRAD + STORE == 2 instantiation)
    const bool fwd_dxi = mode == NS_PROG_FWD_DXI;
    for (int i = 0; i < nl && (dx_prog || fwd_dxi || train); ++i) {      // shape checks as in the forward program
        const linna_layer_t& l = layers[i];
        if (l.K != width || l.N < 1 || l.N > 1024 || l.K > 1024) return p;
        if (l.op == LINNA_OP_LINEAR) { if (l.alpha != 1.f) return p; }
        else if (l.op == LINNA_OP_RESBLOCK) { if (l.C < 1 || l.C > 64 || (!l.Ws && l.K != l.N)) return p; }
        else return p;
        width = l.N;
    }
    auto push_dx = [&](int first) {
      for (int i = nl - 1; i >= first; --i) {
        const linna_layer_t& l = layers[i];
        const int npad = ceil16(l.N);
        if (l.op == LINNA_OP_LINEAR) {
            Lin B{l.W, (l.K + 3) & ~3, l.N, npad, nullptr, 0, 0, 0.f, nullptr, 0.f, l.K, 0, 0, false};
            B.transA = 1; B.op = i;
            lins.push_back(B);
        } else {
            Lin A{nullptr, 0, 0, 0, l.W2, (l.C + 3) & ~3, l.N, 0.1f, nullptr, 0.f, l.C, 0, npad, true};     // dh behind dy
            A.transB = 1; A.op = i;
            lins.push_back(A);
            Lin B{l.Ws, (l.K + 3) & ~3, l.N, npad, l.W1, (l.K + 3) & ~3, l.C, 1.f, nullptr, 0.f, l.K, 0, 0, false};
            B.transA = 1; B.transB = 1; B.op = i;
            lins.push_back(B);
        }
      }
    };
    if (dx_prog) push_dx(mode == NS_PROG_DX_INPUT ? 0 : 1);
    width = in_size;
    for (int i = 0; i < nl && !dx_prog; ++i) {
        const linna_layer_t& l = layers[i];
        if (l.K != width && l.op != LINNA_OP_INSKIP) return p;
        if (l.op == LINNA_OP_LINEAR) {
            if (l.alpha != 1.f || l.N < 1 || l.N > 1024) return p;
            lins.push_back(Lin{l.W, (l.K + 3) & ~3, l.K, ceil16(l.K), nullptr, 0, 0, 0.f, l.b, 1.f, l.N, l.relu, 0, false});
            lins.back().op = i;
        } else if (l.op == LINNA_OP_RESBLOCK) {
            if (l.C < 1 || l.C > 64 || l.N < 1 || l.N > 1024 || (!l.Ws && l.K != l.N)) return p;
            const int inpad = ceil16(l.K);
            lins.push_back(Lin{l.W1, (l.K + 3) & ~3, l.K, inpad, nullptr, 0, 0, 0.f, l.b1, 1.f, l.C, 1, inpad, true});   // h behind x
            lins.back().op = i;
            lins.push_back(Lin{l.Ws, (l.K + 3) & ~3, l.K, inpad, l.W2, (l.C + 3) & ~3, l.C, 0.1f, l.b2, 0.1f, l.N, 1, 0, false});
            lins.back().op = i;
        } else if (l.op == LINNA_OP_INSKIP) {
            // out = last(h) + alpha (x0 Wl^T + bl) (nn.py:195): the last layer becomes ONE GEMM over [h ; x0] with [W | alpha Wl]
            // and bias b + alpha bl; x0 (the network input, kept in LDS) is copied behind h before the segment runs
            if (i != nl - 1 || lins.empty() || l.K != in_size || l.N != width || in_size > 64 || mode == NS_PROG_FWD) return p;
            Lin& last = lins.back();
            if (last.Wb || last.same_buf || last.relu || !last.Wa) return p;
            last.Wb = l.W; last.ldb = (l.K + 3) & ~3; last.Kb = l.K; last.alpha = l.alpha;
            last.b2 = l.b; last.b2scale = l.alpha; last.x0_col = last.Kapad;
            p.x0_keep = 1;
            continue;
        } else {
            return p;
        }
        width = l.N;
    }
    if (dx_prog) in_size = layers[nl - 1].N;                   // the rows of this program are d loss / d output
    if (lins.empty() || lins.back().relu || (int)lins.size() > NS_MAXSEG) return p;
    if (mode == NS_PROG_FWD_DENSE || train) {
        Lin& last = lins.back();
        if (last.Wb || last.same_buf || last.dst_col) return p;               // the last op must be a plain linear layer
        last.rscale = dn->cscale; last.rshift = dn->cshift;
        const int no = last.N, npad = ceil16(no);
        if ((int)lins.size() + 1 > NS_MAXSEG) return p;
        const bool behind = no <= 256;                                          // SPLIT: U behind d in the same buffer
        Lin Q{dn->S, dn->lds, no, npad, nullptr, 0, 0, 0.f, nullptr, 0.f, no, 0, behind ? npad : 0, behind};
        Q.op = nl;
        // factored: the matrix is L (S = L L^T, not symmetric) and the segment must produce d L: U_n = sum_k d_k L[k][n] -- the
        // pack kernel reads it transposed (the row-dot GEMM of the layered path reads S[k][n] as it is)
        if (dn->factored) Q.transA = 1;
        lins.push_back(Q);
        p.dense = 1; p.u_col = Q.dst_col; p.u_same = behind ? 1 : 0;
    }
    const int nfwd = (int)lins.size();
    if (fwd_dxi) {
        if (in_size > 64 || lins.back().N > 64) return p;      // (prologue / turnaround constants are held for <= 64 columns)
        push_dx(0);
        if ((int)lins.size() > NS_MAXSEG) return p;
    }
    if (train) {                                               // the turnaround works on the rows in LDS: any width
        if (nl < 2) return p;
        push_dx(1);
        if ((int)lins.size() > NS_MAXSEG) return p;
    }
    // Backward (d lnP / d z) for plain ReLU MLPs whose hidden layers come out as WIDE segments: the backward
    // GEMM of layer l is a forward-shaped segment over W_l^T (contraction N_l, K_l outputs); for l >= 1 it is
    // forced WIDE so that its lane <-> (row, column) map equals that of the layer whose sign bits it applies.
    bool want_grad = allow_grad && in_size <= 64 && lins.back().N <= 64 && 2 * nfwd <= NS_MAXSEG;
    for (int i = 0; i < nfwd && want_grad; ++i) {
        const Lin& L = lins[i];
        if (L.Wb || L.same_buf || L.dst_col || !L.Wa) want_grad = false;
        if (i < nfwd - 1 && (!L.relu || L.N <= 256)) want_grad = false;       // hidden layers must be WIDE (N > 256)
    }
    if (want_grad) {
        for (int i = nfwd - 1; i >= 0; --i) {
            const Lin F = lins[i];
            Lin Bk{F.Wa, F.lda, F.N, ceil16(F.N), nullptr, 0, 0, 0.f, nullptr, 0.f, F.Ka, 0, 0, false};
            Bk.transA = 1; Bk.force_wide = i >= 1; Bk.mask_apply_of = i - 1;
            lins.push_back(Bk);
        }
    }

    // 2. segment shapes.  SPLIT when it must write into its own input buffer (h of a residual block) or when
    //    splitting K over the idle waves saves at least three steps; WIDE otherwise.
    int bias_off = 0, G = 0, zero_off = -1, zero_pad = 0;
    std::vector<int> in_ext(lins.size());
    for (size_t i = 0; i < lins.size(); ++i) {
        const Lin& L = lins[i];
        const int ksteps = (L.Kapad + ceil16(L.Kb)) / 16;
        NsSeg s; NsPackSeg q;
        std::memset(&s, 0, sizeof(s));
        s.relu = L.relu; s.dst_col = L.dst_col; s.bias_off = bias_off;
        const int passes = (L.N + 511) / 512;
        int ncg = L.N <= 64 ? 1 : L.N <= 128 ? 2 : L.N <= 256 ? 4 : 0;
        const int split_steps = ncg ? (ksteps + NS_NW / ncg - 1) / (NS_NW / ncg) : 0;
        const bool split = ncg && !L.force_wide && (L.same_buf || split_steps + 3 <= ksteps * passes);
        if (L.same_buf && !ncg) return p;
        const int kc = L.N <= 16 ? 4 : L.N <= 32 ? 2 : 1;
        const int side_steps = (ksteps + NS_NW * kc - 1) / (NS_NW * kc);
        // (never the first segment nor the last forward one: the kernel runs a SIDE segment between two runs of the step loop;
        // kc = 1, <= 64 columns: the SPLIT mapping itself, run out of the stream -- 250 -> 64 is two steps per wave)
        // ... and, in the one-launch gradient, d/dh of a residual block (0.1 dy W2 gated by h: 500 / 250 / 125 -> 16 / 32 / 64,
        // the second K part alone, read transposed): one SIDE step each instead of 4 / 2 / 1 SPLIT steps and a SPLIT boundary
        // (the LAST forward segment too where nothing follows it in the launch -- serving programs without a backward half:
        // ChtoModelv2's 33 -> 33 last layer is one SIDE step instead of a three-step WIDE run)
        const bool last_ok = (int)i == nfwd - 1 && !want_grad && !fwd_dxi && !train && !dx_prog;
        const bool side_fwd = ((int)i < nfwd - 1 || last_ok) && !L.Wb && !L.transA && L.Wa;
        const bool side_bwd = fwd_dxi && (int)i > nfwd && !L.Wa && L.Wb && L.transB && L.Kapad == 0 && !L.relu;
        const bool side_pays = split || (side_fwd && !L.force_wide && side_steps < ksteps * passes);   // (a short WIDE run of <= 64 columns)
        const bool side = k4 && side_pays && ncg == 1 && side_steps <= 2 && i > 0 && (side_fwd || side_bwd) && p.seg.back().type != NS_SIDE &&
                          !L.rscale && !L.rshift && !L.b2 && !L.x0_col;
        if (side) {
            s.type = NS_SIDE; s.steps = side_steps; s.passes = 1; s.kslice = 16 * kc * s.steps;
            s.ncg_log2 = 0; s.kcl = kc == 4 ? 2 : kc == 2 ? 1 : 0;
            s.zext = 64;
            in_ext[i] = NS_NW * s.kslice;
            q.bias_pad = 64; q.ncg = 1;
        } else if (split) {
            s.type = NS_SPLIT; s.steps = split_steps; s.passes = 1; s.kslice = 16 * s.steps;
            s.ncg_log2 = ncg == 1 ? 0 : ncg == 2 ? 1 : 2;
            s.zext = 64 * ncg;
            in_ext[i] = (NS_NW / ncg) * s.kslice;
            q.bias_pad = 64 * ncg; q.ncg = ncg;
        } else {
            s.type = NS_WIDE; s.steps = ksteps; s.passes = passes;
            in_ext[i] = 16 * ksteps;
            q.bias_pad = 512 * passes; q.ncg = 1;
        }
        q.Wa = L.Wa; q.lda = L.lda; q.Ka = L.Ka; q.Kapad = L.Kapad; q.Wb = L.Wb; q.ldb = L.ldb; q.Kb = L.Kb; q.alpha = L.alpha;
        q.b = L.b; q.bscale = L.bscale; q.N = L.N; q.type = s.type; q.steps = s.steps; q.passes = s.passes; q.bias_off = bias_off;
        q.transA = L.transA; q.transB = L.transB; q.rscale = L.rscale; q.rshift = L.rshift; q.b2 = L.b2; q.b2scale = L.b2scale;
        q.kc = side ? kc : 1; q.side_off = 0;
        s.x0_col = L.x0_col; s.x0_n = L.x0_col ? ceil16(L.Kb) : 0;
        if (L.transA && !dx_prog) {                                     // backward segments have no bias: ONE shared block of zeros
            if (zero_off < 0) { zero_off = bias_off; zero_pad = 0; }
            s.bias_off = q.bias_off = zero_off;
            const int grow = std::max(0, q.bias_pad - zero_pad);
            zero_pad += grow; q.bias_pad = grow;            // (the first backward segment's record carries the block; later ones extend it)
        }
        bias_off += q.bias_pad;
        // the Cholesky factor of a dense inverse covariance (NsDense::factored) is lower triangular: in the second pass (columns
        // >= 512) the rows k < 512 are zero -- that pass starts at k = 512 (bit-identical: the skipped products are zeros)
        // tri = 2 and exactly 16 column blocks (960 < nout <= 1024): the balanced assignment instead (ns_seg_steps) -- the zero
        // rows of EVERY 64-column block are skipped, not only those of the second pass, and every wave runs the same number
        // of steps: 66 instead of 94 at nout = 1000
        if (dn && dn->tri > 0 && s.type == NS_WIDE && dn->factored && L.Wa == dn->S && passes == 2 && ksteps > 32 && !train) {
            if (dn->tri == 2 && (L.N + 63) / 64 == 16 && ksteps > 60) s.zext = -1;
            else { s.kslice = 512; s.zext = ksteps - 32; }
        }
        q.koff2 = s.type == NS_WIDE ? (s.zext < 0 ? -1 : s.kslice) : 0;
        if (!side) G += ns_seg_steps(s);           // (a SIDE segment is not part of the weight stream)
        p.seg.push_back(s); p.pack.push_back(q);
    }
    if (want_grad) {
        // sign-bit slots: one per pass of every hidden forward segment; the backward segment of layer i+1 applies them
        int slot = 0;
        for (int i = 0; i < nfwd - 1; ++i) {
            if (p.seg[i].type != NS_WIDE) { want_grad = false; break; }
            p.seg[i].mask_store = 1 + slot;
            slot += p.seg[i].passes;
        }
        for (size_t j = nfwd; j < lins.size() && want_grad; ++j) {
            const int of = lins[j].mask_apply_of;
            if (of < 0) continue;
            if (p.seg[j].type != NS_WIDE || p.seg[j].passes != p.seg[of].passes) { want_grad = false; break; }
            p.seg[j].mask_apply = p.seg[of].mask_store;
        }
        if (want_grad) p.mask_slots = slot;
        else {   // drop the backward half again
            for (size_t j = lins.size(); j-- > (size_t)nfwd;) { G -= ns_seg_steps(p.seg[j]); bias_off -= p.pack[j].bias_pad; }
            p.seg.resize(nfwd); p.pack.resize(nfwd); lins.resize(nfwd); in_ext.resize(nfwd);
            for (int i = 0; i < nfwd; ++i) p.seg[i].mask_store = 0;
        }
    }
    int Gf = 0;
    for (int i = 0; i < nfwd; ++i) if (p.seg[i].type != NS_SIDE) Gf += ns_seg_steps(p.seg[i]);

    // 3. every column a segment reads must have been WRITTEN (finite; zero where the weights are zero):
    //    track the defined prefix [0, def) of the current buffer and widen the zero fill of the last
    //    SPLIT writer (or of the prologue) where a consumer reads further.
    p.kpad0 = in_ext[0];
    int def = p.kpad0, writer = -1;                         // writer: segment whose write ends at `def` (-1 prologue, -2 fixed)
    int maxext = 64;
    for (size_t i = 0; i < p.seg.size(); ++i) {
        NsSeg& s = p.seg[i];
        if (in_ext[i] > def) {
            if (writer == -1) p.kpad0 = def = in_ext[i];
            else if (writer >= 0) { NsSeg& w = p.seg[writer]; w.zext = in_ext[i] - w.dst_col; def = in_ext[i]; }
            else return p;
        }
        maxext = std::max(maxext, in_ext[i]);
        if (s.type == NS_WIDE) {
            if (s.dst_col != 0) return p;
            def = 512 * s.passes; writer = -2;
            maxext = std::max(maxext, def);
        } else {
            if (s.dst_col > def) return p;
            if (s.dst_col == 0) def = s.zext;               // overwrites its input from column 0
            else def = std::max(def, s.dst_col + s.zext);
            writer = (int)i;
        }
    }
    for (const NsSeg& s : p.seg) if (s.type != NS_WIDE) maxext = std::max(maxext, s.dst_col + s.zext);
    if (p.kpad0 > (dx_prog ? 1024 : 256)) return p;
    p.nout = lins[nfwd - 1].N;
    p.G = Gf; p.Gstride = G; p.nseg_f = nfwd; p.grad_ok = want_grad;
    for (size_t i = 0; i < lins.size(); ++i) { p.seg_op.push_back(lins[i].op); p.seg_hidden.push_back(lins[i].same_buf ? 1 : 0); }
    p.dxi_ok = fwd_dxi;
    p.train_ok = train;
    p.bias_total = bias_off;
    p.LD = std::max(((maxext + 63) & ~63) + 4, 516);        // >= 516: SPLIT partials need [8][rows][64] floats in one buffer
    if (bias_off > 3 * 64 * NS_NW * 4) return p;            // BMAX rounds of float4 per thread
    p.lds_bytes = (size_t)(2 * NS_ROWS * p.LD + ((bias_off + 3) & ~3)) * sizeof(float) + 128 + (p.x0_keep ? 4096 : 0);
#ifdef NS_STAMPS
    p.lds_bytes += NS_NW * 32 * 8;
#endif
    if (p.lds_bytes > (size_t)NS_LDS_BYTES) return p;
    p.lds_bytes_grad = p.lds_bytes + (size_t)p.mask_slots * 64 * NS_NW * sizeof(unsigned);
    if (p.grad_ok && p.lds_bytes_grad > (size_t)NS_LDS_BYTES) return p;   // (never for the eligible shapes)
    int nrun = 0;
    for (const NsSeg& s : p.seg) nrun += s.passes;
    if (nrun > NS_MAXRUN) return p;
    p.packed_floats = (size_t)NS_NW * G * NS_NT * 256 + (size_t)((bias_off + 3) & ~3);
    for (size_t i = 0; i < p.seg.size(); ++i)
        if (p.seg[i].type == NS_SIDE) {            // blocks of their own behind the biases
            p.seg[i].side_off = p.pack[i].side_off = (int)p.packed_floats;
            p.packed_floats += (size_t)NS_NW * p.seg[i].steps * NS_NT * 256;
            p.side_f4 += (size_t)NS_NW * p.seg[i].steps * NS_NT * 64;
        }
    p.ok = true;
    return p;
}

static const NsProgram& ns_build_prog(const linna_layer_t* layers, int nl, int in_size, int prog, const NsDense* dn = nullptr, bool k4 = false);
bool net_stream_eligible(const linna_layer_t* layers, int nl, int in_size) { return ns_build_prog(layers, nl, in_size, 0, nullptr).ok; }
size_t net_stream_packed_floats(const linna_layer_t* layers, int nl, int in_size) {
    return ns_build_prog(layers, nl, in_size, 0, nullptr).packed_floats;
}

// Engine for a batch of B rows: the fewest rows per workgroup that still fit the batch into one workgroup per CU
// (linna_engine_rows forces one for tests and measurements; LINNA_NS_ROWS in the environment sets the initial value,
// read ONCE -- the launch path reads an atomic, not the environment).
static std::atomic<int> g_forced_rows{-1};
static int ns_forced_rows_resolved() {             // -1 (never read) -> the environment's value, once
    int forced = g_forced_rows.load(std::memory_order_relaxed);
    if (forced < 0) {
        const char* const env = getenv("LINNA_NS_ROWS");
        const int v = env ? atoi(env) : 0;
        forced = (v == 4 || v == 8 || v == 16) ? v : 0;
        int expect = -1;
        g_forced_rows.compare_exchange_strong(expect, forced);
        forced = g_forced_rows.load(std::memory_order_relaxed);
    }
    return forced;
}
int net_stream_force_rows(int rows) {
    if (rows != 0 && rows != 4 && rows != 8 && rows != 16) return -1;
    (void)ns_forced_rows_resolved();               // the environment's value is what `prev = engine_rows(4); ...; engine_rows(prev)` must restore
    return g_forced_rows.exchange(rows);
}
int net_stream_rows(int B) {
    const int forced = ns_forced_rows_resolved();
    if (forced) return forced;
    static int ncu = 0;
    if (!ncu) {
        int dev = 0; hipDeviceProp_t pr;
        ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0)
                  ? pr.multiProcessorCount : 256;
    }
    return B <= 4 * ncu ? 4 : B <= 8 * ncu ? 8 : 16;
}

// SIDE segments: serving launches (linna_logprob_*, the fused sampler moves) on the 16-row engine; LINNA_NS_K4=0 turns them off
bool net_stream_k4(int rows, int serve) {
    static const bool on = !(getenv("LINNA_NS_K4") && getenv("LINNA_NS_K4")[0] == '0');
    return on && serve && rows == 16;
}

// SIDE segments in the forward half of the one-launch gradient (16-row engine).  Round 3 measured them slower there (with the
// activations stored for the gates); with the gates as sign bits in LDS and the per-segment tables read through the
// kernel-argument segment (without that the instantiation spilled 2.5 KB per lane): 151.4 -> 148.7 us at ChtoModelv2(33,33),
// 4096 chains (NOTES R4).  LINNA_G2_SIDE=0 turns them off.
bool net_stream_g2_side() {
    static const bool on = !(getenv("LINNA_G2_SIDE") && getenv("LINNA_G2_SIDE")[0] == '0');
    return on;
}

static NsProgram ns_build_prog_uncached(const linna_layer_t* layers, int nl, int in_size, int prog, const NsDense* dn, bool k4) {
    if (k4 && prog == 3) {
        NsProgram p = ns_build_one(layers, nl, in_size, NS_PROG_FWD_DXI, nullptr, true);
        if (p.ok && p.dxi_ok) return p;
    }
    if (k4 && prog == 0) {                     // the serving programs of the 16-row engine; the plain program where this one does not fit
        NsProgram p = dn ? ns_build_one(layers, nl, in_size, NS_PROG_FWD_DENSE, dn, true) : ns_build(layers, nl, in_size, true);
        if (p.ok) return p;
    }
    if (prog == 0 && dn) return ns_build_one(layers, nl, in_size, NS_PROG_FWD_DENSE, dn);
    if (prog == 3) return ns_build_one(layers, nl, in_size, NS_PROG_FWD_DXI);
    if (prog == 4) return ns_build_one(layers, nl, in_size, NS_PROG_TRAIN, dn);
    return prog == 0 ? ns_build(layers, nl, in_size) : ns_build_one(layers, nl, in_size, prog == 2 ? NS_PROG_DX_INPUT : NS_PROG_DX);
}
// Kernel-configuration cache: a program is a pure function of the op list (shapes AND parameter pointers:
// the pack descriptors carry them), the program kind and the dense descriptor, so it is built once and looked up by
// those bytes on every later launch -- no segment planning, no vector allocation on the launch path.  Entries live for
// the life of the library (a handful per network; the table is cleared if it ever reaches 256 entries).
static const NsProgram& ns_build_prog(const linna_layer_t* layers, int nl, int in_size, int prog, const NsDense* dn, bool k4) {
    static std::mutex mu;
    static std::unordered_map<std::string, std::unique_ptr<NsProgram>> cache;
    std::string key;
    key.reserve((size_t)nl * sizeof(linna_layer_t) + 64);
    key.append(reinterpret_cast<const char*>(layers), (size_t)nl * sizeof(linna_layer_t));
    const int hdr[4] = {nl, in_size, prog, k4 ? 1 : 0};
    key.append(reinterpret_cast<const char*>(hdr), sizeof(hdr));
    if (dn) {                                               // field by field: the struct has padding bytes
        const void* const ptrs[3] = {dn->S, dn->cscale, dn->cshift};
        key.append(reinterpret_cast<const char*>(ptrs), sizeof(ptrs));
        key.append(reinterpret_cast<const char*>(&dn->lds), sizeof(int));
        key.append(reinterpret_cast<const char*>(&dn->factored), sizeof(int));
        key.append(reinterpret_cast<const char*>(&dn->tri), sizeof(int));
    }
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(key);
    if (it != cache.end()) return *it->second;
    if (cache.size() >= 256) cache.clear();
    auto ins = cache.emplace(std::move(key), std::unique_ptr<NsProgram>(new NsProgram(ns_build_prog_uncached(layers, nl, in_size, prog, dn, k4))));
    return *ins.first->second;
}
static int ns_g2_cols(const NsProgram& p, const linna_layer_t* layers, int nl);
// Text form of a program (tests, diagnostics): one line per segment, "type steps passes ncg kc dst_col zext".
int net_stream_describe(const linna_layer_t* layers, int nl, int in_size, int prog, const NsDense* dn, int rows, int serve, char* buf,
                        size_t n) {
    const NsProgram& p = ns_build_prog(layers, nl, in_size, prog, dn, (prog == 0 && net_stream_k4(rows, serve)) || (prog == 3 && rows == 16 && net_stream_g2_side()));
    std::string out = p.ok ? "ok" : "not eligible";
    char line[160];
    snprintf(line, sizeof line, " G %d Gstride %d nseg_f %d LD %d kpad0 %d packed_floats %zu grad %d\n", p.G, p.Gstride, p.nseg_f, p.LD,
             p.kpad0, p.packed_floats, (int)p.grad_ok);
    out += line;
    for (size_t i = 0; p.ok && i < p.seg.size(); ++i) {
        const NsSeg& g = p.seg[i];
        snprintf(line, sizeof line, "%s steps %d passes %d ncg %d kc %d dst %d zext %d N %d\n",
                 g.type == NS_WIDE ? "WIDE" : g.type == NS_SPLIT ? "SPLIT" : "SIDE", g.steps, g.passes, 1 << g.ncg_log2, 1 << g.kcl, g.dst_col,
                 g.zext, p.pack[i].N);
        out += line;
    }
    if (p.ok && prog == 3) {                        // the one-launch gradient: its LDS with the sign-bit matrix, on the 16-row engine
        const int cols = ns_g2_cols(p, layers, nl);
        const size_t need = ((p.lds_for(NS_ROWS, true) + 7) & ~(size_t)7) + (size_t)NS_ROWS * (cols / 32) * sizeof(unsigned);
        snprintf(line, sizeof line, "lds %zu of %d bytes with %d sign-bit columns: %s\n", need, NS_LDS_BYTES, cols,
                 need <= (size_t)NS_LDS_BYTES && p.dxi_ok ? "one launch" : "layered");
        out += line;
    }
    if (buf && n) { snprintf(buf, n, "%s", out.c_str()); }
    return p.ok ? (int)p.seg.size() : 0;
}
bool net_stream_dense_eligible(const linna_layer_t* layers, int nl, int in_size, const NsDense& dn) {
    return ns_build_prog(layers, nl, in_size, 0, &dn).ok;
}
size_t net_stream_dense_packed_floats(const linna_layer_t* layers, int nl, int in_size, const NsDense& dn) {
    return ns_build_prog(layers, nl, in_size, 0, &dn).packed_floats;
}
bool net_stream_tb_eligible(const linna_layer_t* layers, int nl, int in_size, const NsDense& dn) {
    const NsProgram& p = ns_build_prog(layers, nl, in_size, 4, &dn);
    return p.ok && p.train_ok;
}
size_t net_stream_tb_packed_floats(const linna_layer_t* layers, int nl, int in_size, const NsDense& dn) {
    return ns_build_prog(layers, nl, in_size, 4, &dn).packed_floats;
}
bool net_stream_dx_eligible(const linna_layer_t* layers, int nl, int in_size, int with_input) {
    return nl >= (with_input ? 1 : 2) && ns_build_prog(layers, nl, in_size, with_input ? 2 : 1).ok;
}
size_t net_stream_dx_packed_floats(const linna_layer_t* layers, int nl, int in_size, int with_input) {
    return ns_build_prog(layers, nl, in_size, with_input ? 2 : 1).packed_floats;
}

// prog: 0 the forward program (+ the fused gradient's backward half), 1 / 2 the dX chain without / with op 0
int launch_net_stream_pack(const linna_layer_t* layers, int nl, int in_size, float* packed, int rows, int prog,
                           const NsDense* dn, hipStream_t s, int serve) {
    const NsProgram& p = ns_build_prog(layers, nl, in_size, prog, dn, (prog == 0 && net_stream_k4(rows, serve)) || (prog == 3 && rows == 16 && net_stream_g2_side()));
    if (!p.ok) { set_error("net_stream: network not eligible"); return LINNA_ERR_UNSUPPORTED; }
    NsPackArgs a;
    ::memset(static_cast<void*>(&a), 0, sizeof(a));
    a.nseg = (int)p.seg.size(); a.G = p.Gstride; a.bias_total = p.bias_total; a.out = packed;
    a.small = rows < 16;
    int nrun = 0, first = 0;
    for (int i = 0; i < a.nseg; ++i) {
        a.seg[i] = p.pack[i];
        if (p.seg[i].type == NS_SIDE) continue;                 // not in the stream: ns_pack_side_kernel below
        if (p.seg[i].type == NS_WIDE && p.seg[i].zext < 0) {     // balanced triangular: ONE run of the stream, split per wave (ns_pack_kernel)
            a.run_seg[nrun] = i; a.run_pass[nrun] = 0; a.run_first[nrun] = first;
            first += ns_seg_steps(p.seg[i]); ++nrun;
            continue;
        }
        for (int ps = 0; ps < p.seg[i].passes; ++ps) {
            a.run_seg[nrun] = i; a.run_pass[nrun] = ps; a.run_first[nrun] = first;
            first += (ps > 0 && p.seg[i].type == NS_WIDE && p.seg[i].zext > 0) ? p.seg[i].zext : p.seg[i].steps; ++nrun;
        }
    }
    a.run_first[nrun] = first; a.nrun = nrun;
    const size_t total = (size_t)NS_NW * p.Gstride * NS_NT * 64 + (size_t)p.bias_total;
    hipLaunchKernelGGL(ns_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
    if (p.side_f4) hipLaunchKernelGGL(ns_pack_side_kernel, dim3((unsigned)((p.side_f4 + 255) / 256)), dim3(256), 0, s, a);
    return check_hip(hipGetLastError(), "net_stream pack launch");
}

template <int MOVE, bool GRAD, int STORE, int ROWS>
static int ns_launch_rows(const NsArgs& a, int B, size_t lds_bytes, hipStream_t s, int extra = 0) {
    static bool attr_set = false;
    if (!attr_set) {
        const int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&net_stream_kernel<NS_R, MOVE, GRAD, STORE, ROWS>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, NS_LDS_BYTES), "hipFuncSetAttribute");
        if (rc != LINNA_OK) return rc;
        attr_set = true;
    }
    hipLaunchKernelGGL((net_stream_kernel<NS_R, MOVE, GRAD, STORE, ROWS>), dim3((B + ROWS - 1) / ROWS + extra), dim3(64 * NS_NW), lds_bytes, s, a);
    return check_hip(hipGetLastError(), "net_stream launch");
}
template <int MOVE, bool GRAD, int STORE = 0>
static int ns_launch_kernel(const NsArgs& a0, int B, const NsProgram& p, int rows, hipStream_t s, int extra = 0, size_t lds_extra = 0) {
    const size_t lds = p.lds_for(rows, GRAD) + lds_extra;
#ifdef NS_STAMPS
    // diagnostic build: every launch writes its phase stamps to the buffer LINNA_FUSED_STAMPS names (tools/ns_stamps*.py)
    NsArgs a = a0;
    a.stamps = getenv("LINNA_FUSED_STAMPS") ? reinterpret_cast<unsigned long long*>(strtoull(getenv("LINNA_FUSED_STAMPS"), nullptr, 16)) : nullptr;
    if (!a.stamps) { set_error("net_stream: NS_STAMPS build needs LINNA_FUSED_STAMPS"); return LINNA_ERR_INVALID; }
#else
    const NsArgs& a = a0;
#endif
    if (rows == 4) return ns_launch_rows<MOVE, GRAD, STORE, 4>(a, B, lds, s, extra);
    if constexpr (STORE == 3 && GRAD) {
        // the one-launch training step exists for the 4-row engine only (batches up to 1024 rows; the caller checks): on the
        // 8-row engine it was measured SLOWER than its two halves (batch 1500 at (26,457): 218.5 against 210.9 us per step)
    } else {
        if (rows == 8) return ns_launch_rows<MOVE, GRAD, STORE, 8>(a, B, lds, s, extra);
        if (rows == 16) return ns_launch_rows<MOVE, GRAD, STORE, 16>(a, B, lds, s, extra);
    }
    set_error("net_stream: %d rows per workgroup", rows);
    return LINNA_ERR_INVALID;
}

// ------------------------------------------------------------------ AdamW that writes the weight streams itself
// A training step ends with AdamW over the flat parameter buffer and begins with two re-layouts of the updated weights
// (ns_pack_kernel: the forward + loss stream and the dX-chain stream): three launches that each read or write every
// parameter.  Here the update runs once, in the flat buffer's own order, and every thread puts its four updated values
// where the two streams want them: the forward stream holds four consecutive k of one weight row as ONE 16-byte vector
// (one store), the dX-chain stream holds the transposed matrix (four 4-byte stores; neighbouring lanes fill
// neighbouring vectors).  Biases go to the forward stream's bias block.  Constant parts of a stream (zero padding, the
// loss's inverse covariance) are written by the ordinary re-layout once and never touched here.
constexpr int AS_BLOCK = 64;               // one wave per block: ~1200 blocks for 1.3 M parameters, five per CU
__device__ __forceinline__ void as_update(f32x4& P4, const f32x4& G4, f32x4& M4, f32x4& V4, float lr, float wd, float bc1,
                                          float sbc2, float beta1, float beta2, float eps) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {                   // adamw_kernel's arithmetic, operation for operation
        const float gi = G4[e];
        float pi = P4[e] * (1.f - lr * wd);
        float mi = M4[e];
        mi = mi + (gi - mi) * (1.f - beta1);
        const float vi = V4[e] * beta2 + (1.f - beta2) * gi * gi;
        const float denom = sqrtf(vi) / sbc2 + eps;
        pi = pi - (lr / bc1) * (mi / denom);
        P4[e] = pi; M4[e] = mi; V4[e] = vi;
    }
}

// One work item = a 4 x 4 block of a weight matrix (rows n0..n0+3, columns k0..k0+3; 16-byte loads and stores throughout:
// the forward stream takes the block's rows as four vectors, the dX-chain stream its columns) or four bias elements.
__global__ __launch_bounds__(AS_BLOCK) void adamw_streams_kernel(AsArgs a, float* __restrict__ p, const float* __restrict__ g,
                                                            float* __restrict__ m, float* __restrict__ v,
                                                            const float* __restrict__ hyper, float beta1, float beta2, float eps) {
    int ri = 0;
    while (ri + 1 < a.nr && blockIdx.x >= a.r[ri + 1].blk0) ++ri;
    const AsRange R = a.r[ri];
    const unsigned it = (blockIdx.x - R.blk0) * AS_BLOCK + threadIdx.x;
    if (it >= R.n4) return;
    const float lr = hyper[0], wd = hyper[1], bc1 = hyper[2], sbc2 = hyper[3];
    if (R.kind == 1) {
        const size_t i = ((size_t)R.off4 + it) * 4;
        const f32x4 G4 = *reinterpret_cast<const f32x4*>(g + i);
        f32x4 P4 = *reinterpret_cast<const f32x4*>(p + i), M4 = *reinterpret_cast<const f32x4*>(m + i), V4 = *reinterpret_cast<const f32x4*>(v + i);
        as_update(P4, G4, M4, V4, lr, wd, bc1, sbc2, beta1, beta2, eps);
        *reinterpret_cast<f32x4*>(p + i) = P4; *reinterpret_cast<f32x4*>(m + i) = M4; *reinterpret_cast<f32x4*>(v + i) = V4;
        const AsBias B = a.b[R.idx];
        if (B.out) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if ((int)(4 * it) + e < B.N) B.out[4 * it + e] = B.scale * P4[e];
        }
        return;
    }
    const AsMat& W = a.w[R.idx];
    const unsigned ld4 = (unsigned)W.ld >> 2;
    const int n0 = 4 * (int)(it / ld4), k0 = 4 * (int)(it % ld4);
    f32x4 P4[4], G4[4], M4[4], V4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const size_t i = (size_t)R.off4 * 4 + (size_t)min(n0 + r, W.N - 1) * W.ld + k0;     // (rows past N: reread the last, not stored)
        G4[r] = *reinterpret_cast<const f32x4*>(g + i); P4[r] = *reinterpret_cast<const f32x4*>(p + i);
        M4[r] = *reinterpret_cast<const f32x4*>(m + i); V4[r] = *reinterpret_cast<const f32x4*>(v + i);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        as_update(P4[r], G4[r], M4[r], V4[r], lr, wd, bc1, sbc2, beta1, beta2, eps);
        if (n0 + r < W.N) {
            const size_t i = (size_t)R.off4 * 4 + (size_t)(n0 + r) * W.ld + k0;
            *reinterpret_cast<f32x4*>(p + i) = P4[r]; *reinterpret_cast<f32x4*>(m + i) = M4[r]; *reinterpret_cast<f32x4*>(v + i) = V4[r];
        } else {
            P4[r] = f32x4{0.f, 0.f, 0.f, 0.f};      // the streams' padding
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const AsPlace& q = W.pl[j];
        if (!q.out) continue;
        if (!q.trans) {
            // row n0 + r, columns k0..k0+3: one vector of the stream (koff and k0 are multiples of 4; pad columns hold zeros)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (n0 + r < W.N)
                    *reinterpret_cast<f32x4*>(q.out + as_slot(q, a.small, n0 + r, q.koff + k0)) =
                        f32x4{q.scale * P4[r][0], q.scale * P4[r][1], q.scale * P4[r][2], q.scale * P4[r][3]};
        } else {
            // column k0 + e, rows n0..n0+3: one vector of the transposed stream (rows past N: the zeros of its padding)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (k0 + e < q.ncols)
                    *reinterpret_cast<f32x4*>(q.out + as_slot(q, a.small, k0 + e, q.koff + n0)) =
                        f32x4{q.scale * P4[0][e], q.scale * P4[1][e], q.scale * P4[2][e], q.scale * P4[3][e]};
        }
    }
}

// Descriptor table of adamw_streams_kernel for the flat buffer `params[nflat]` the layers' parameters live in, the
// forward(+loss) stream `s_fwd` (program 0 with `dn`) and the dX-chain stream `s_dx` (program 1).  LINNA_ERR_UNSUPPORTED
// when the buffer is not exactly the layers' tensors back to back, or a stream folds something into the weights that
// an element-wise scatter cannot reproduce (output maps, a second bias).
int net_stream_adamw_args(const linna_layer_t* layers, int nl, int in_size, int rows, const float* params, size_t nflat,
                          float* s_fwd, const NsDense* dn, float* s_dx, AsArgs* out, int merged) {
    // merged: ONE stream holds the forward + loss segments [0, nseg_f) and the dX chain [nseg_f, nseg) (NS_PROG_TRAIN)
    const NsProgram& pf = merged ? ns_build_prog(layers, nl, in_size, 4, dn) : ns_build_prog(layers, nl, in_size, 0, dn);
    const NsProgram& pd = merged ? pf : ns_build_prog(layers, nl, in_size, 1);
    if (merged) s_dx = s_fwd;
    if (!pf.ok || !pd.ok || !s_fwd || !s_dx || (merged && !pf.train_ok)) { set_error("adamw_streams: no forward / dX-chain program"); return LINNA_ERR_UNSUPPORTED; }
    const size_t f_lo = 0, f_hi = merged ? (size_t)pf.nseg_f : pf.pack.size();
    const size_t d_lo = merged ? (size_t)pf.nseg_f : 0, d_hi = pd.pack.size();
    ::memset(static_cast<void*>(out), 0, sizeof(*out));
    out->small = rows < 16;
    struct T { const float* ptr; int N, K, bias; };
    std::vector<T> ts;
    for (int i = 0; i < nl; ++i) {
        const linna_layer_t& l = layers[i];
        if (l.op == LINNA_OP_LINEAR) { ts.push_back({l.W, l.N, l.K, 0}); ts.push_back({l.b, l.N, 0, 1}); }
        else if (l.op == LINNA_OP_RESBLOCK) {
            ts.push_back({l.W1, l.C, l.K, 0}); ts.push_back({l.b1, l.C, 0, 1});
            ts.push_back({l.W2, l.N, l.C, 0}); ts.push_back({l.b2, l.N, 0, 1});
            if (l.Ws) ts.push_back({l.Ws, l.N, l.K, 0});
        } else { set_error("adamw_streams: op %d", l.op); return LINNA_ERR_UNSUPPORTED; }
    }
    std::sort(ts.begin(), ts.end(), [](const T& x, const T& y) { return x.ptr < y.ptr; });
    if ((int)ts.size() > AS_MAXR) { set_error("adamw_streams: %d tensors", (int)ts.size()); return LINNA_ERR_UNSUPPORTED; }
    auto runs_first = [](const NsProgram& p, int seg, int pass) {
        int first = 0;
        for (int i = 0; i < seg; ++i) first += ns_seg_steps(p.seg[i]);
        return first + pass * p.seg[seg].steps;
    };
    auto place = [&](const NsProgram& p, float* base, const float* W, AsPlace* q, size_t lo, size_t hi) -> int {
        for (size_t i = lo; i < hi; ++i) {
            const NsPackSeg& S = p.pack[i];
            const bool isA = S.Wa == W, isB = S.Wb == W;
            if (!isA && !isB) continue;
            if (q->out) { set_error("adamw_streams: a weight matrix twice in one stream"); return LINNA_ERR_UNSUPPORTED; }
            if (S.rscale || S.rshift || S.b2) { set_error("adamw_streams: folded output map"); return LINNA_ERR_UNSUPPORTED; }
            q->out = base; q->scale = isA ? 1.f : S.alpha; q->trans = isA ? S.transA : S.transB; q->koff = isA ? 0 : S.Kapad;
            q->ncols = S.N; q->type = S.type; q->ncg = S.ncg; q->steps = S.steps; q->G = p.Gstride;
            q->first0 = runs_first(p, (int)i, 0); q->first1 = p.seg[i].passes > 1 ? runs_first(p, (int)i, 1) : q->first0;
            if (p.seg[i].passes > 2) { set_error("adamw_streams: %d passes", p.seg[i].passes); return LINNA_ERR_UNSUPPORTED; }
        }
        return LINNA_OK;
    };
    size_t off = 0;
    unsigned blk = 0;
    int nw = 0, nb = 0;
    for (size_t i = 0; i < ts.size(); ++i) {
        const T& t = ts[i];
        if (!t.ptr || t.ptr != params + off) { set_error("adamw_streams: the parameters are not one contiguous buffer"); return LINNA_ERR_UNSUPPORTED; }
        const int ld = t.bias ? 0 : (t.K + 3) & ~3;
        const size_t nf = t.bias ? (size_t)((t.N + 3) & ~3) : (size_t)t.N * ld;
        AsRange& R = out->r[i];
        R.off4 = (unsigned)(off / 4); R.blk0 = blk; R.kind = (short)t.bias;
        R.n4 = t.bias ? (unsigned)(nf / 4) : (unsigned)((t.N + 3) / 4) * (unsigned)(ld / 4);      // work items (see the kernel)
        blk += (R.n4 + AS_BLOCK - 1) / AS_BLOCK;
        if (t.bias) {
            if (nb >= AS_MAXB) { set_error("adamw_streams: biases"); return LINNA_ERR_UNSUPPORTED; }
            R.idx = (short)nb;
            AsBias& B = out->b[nb++];
            B.N = t.N;
            for (size_t j = f_lo; j < f_hi; ++j) {
                const NsPackSeg& S = pf.pack[j];
                if (S.b != t.ptr) continue;
                if (B.out || S.rscale || S.rshift || S.b2) { set_error("adamw_streams: bias folded or used twice"); return LINNA_ERR_UNSUPPORTED; }
                B.out = s_fwd + (size_t)NS_NW * pf.Gstride * NS_NT * 256 + S.bias_off; B.scale = S.bscale;
            }
            for (size_t j = d_lo; j < d_hi; ++j) if (pd.pack[j].b == t.ptr) { set_error("adamw_streams: bias in the dX program"); return LINNA_ERR_UNSUPPORTED; }
        } else {
            if (nw >= AS_MAXW) { set_error("adamw_streams: weight matrices"); return LINNA_ERR_UNSUPPORTED; }
            R.idx = (short)nw;
            AsMat& W = out->w[nw++];
            W.N = t.N; W.ld = ld;
            int rc = place(pf, s_fwd, t.ptr, &W.pl[0], f_lo, f_hi);
            if (rc == LINNA_OK) rc = place(pd, s_dx, t.ptr, &W.pl[1], d_lo, d_hi);
            if (rc != LINNA_OK) return rc;
            if (!W.pl[0].out) { set_error("adamw_streams: a weight matrix outside the forward stream"); return LINNA_ERR_UNSUPPORTED; }
        }
        off += nf;
    }
    if (off != nflat) { set_error("adamw_streams: %zu of %zu floats covered", off, nflat); return LINNA_ERR_UNSUPPORTED; }
    // every pack segment's weights must have been found among the tensors (the loss's constant matrix excepted)
    out->nr = (int)ts.size(); out->nblocks = blk;
    return LINNA_OK;
}

int launch_adamw_streams(const AsArgs& a, float* p, const float* g, float* m, float* v, const float* hyper, float b1, float b2,
                         float eps, hipStream_t s) {
    hipLaunchKernelGGL(adamw_streams_kernel, dim3(a.nblocks), dim3(AS_BLOCK), 0, s, a, p, g, m, v, hyper, b1, b2, eps);
    return check_hip(hipGetLastError(), "adamw_streams launch");
}

bool net_stream_has_grad(const linna_layer_t* layers, int nl, int in_size) { return ns_build_prog(layers, nl, in_size, 0).grad_ok; }
// bit columns of the one-launch gradient's sign matrix: every tensor a gate asks for, each rounded up to 64 columns
static int ns_g2_cols(const NsProgram& p, const linna_layer_t* layers, int nl) {
    int n = 0;
    for (int i = 0; i < p.nseg_f; ++i) {
        const int op = p.seg_op[i];
        if (p.seg_hidden[i]) n += (layers[op].C + 63) & ~63;
        else if (op < nl - 1) n += (layers[op].N + 63) & ~63;
    }
    return n;
}
bool net_stream_dxi_eligible(const linna_layer_t* layers, int nl, int in_size) {
    const NsProgram& p = ns_build_prog(layers, nl, in_size, 3);
    if (!p.ok || !p.dxi_ok) return false;
    return ((p.lds_for(NS_ROWS, true) + 7) & ~(size_t)7) + (size_t)NS_ROWS * (ns_g2_cols(p, layers, nl) / 32) * sizeof(unsigned) <= (size_t)NS_LDS_BYTES;
}
size_t net_stream_dxi_packed_floats(const linna_layer_t* layers, int nl, int in_size) {
    const size_t a = ns_build_prog(layers, nl, in_size, 3).packed_floats;
    return net_stream_g2_side() ? std::max(a, ns_build_prog(layers, nl, in_size, 3, nullptr, true).packed_floats) : a;
}

int launch_net_stream(const linna_layer_t* layers, int nl, int in_size, const float* packed, const float* Z, int ldz, int B,
                      int nin, const int* is_flat, const float* a1, const float* a2, const int* lg, const float* xmean,
                      const float* xstd, const float* cscale, const float* cshift, const float* w, float T, float* lnP,
                      float* D, int ldd, float* TH, int ldt, const NsMove* mv, const NsGrad* gr, const int* gate, int rows,
                      const NsDense* dn, hipStream_t s, const float* cpost, const float* cshift2) {
    const NsProgram& p = ns_build_prog(layers, nl, in_size, 0, dn, net_stream_k4(rows, 1));
    if ((cpost != nullptr) != (cshift2 != nullptr) || (cpost && (dn || gr))) {
        set_error("net_stream: the exp output map needs cpost and cshift2, and has no dense / gradient program"); return LINNA_ERR_INVALID;
    }
    if (!p.ok) { set_error("net_stream: network not eligible"); return LINNA_ERR_UNSUPPORTED; }
    if (dn && (w || gr || cscale || cshift)) { set_error("net_stream: the dense program carries its own output map and has no fused gradient"); return LINNA_ERR_INVALID; }
    if (mv && (nin > 64 || (!w && !dn))) { set_error("net_stream: fused sampler moves need <= 64 parameters and a log-likelihood in the launch"); return LINNA_ERR_UNSUPPORTED; }
    if (gr && (!p.grad_ok || !w || !lnP || !gr->gscale || !gr->G || mv)) { set_error("net_stream: no fused gradient for this network / likelihood"); return LINNA_ERR_UNSUPPORTED; }
    NsArgs a;
    ::memset(static_cast<void*>(&a), 0, sizeof(a));
    a.Z = Z; a.ldz = ldz; a.B = B; a.nin = nin;
    a.is_flat = is_flat; a.a1 = a1; a.a2 = a2; a.lg = lg; a.xmean = xmean; a.xstd = xstd;
    a.packed = packed;
    a.Gstride = p.Gstride; a.nseg_f = p.nseg_f;
    a.G = gr ? p.Gstride : p.G;                        // forward only: stop after the forward segments
    a.nseg = gr ? (int)p.seg.size() : p.nseg_f;
    a.LD = p.LD; a.kpad0 = p.kpad0; a.nout = p.nout; a.bias_total = p.bias_total;
    a.cscale = cscale; a.cshift = cshift; a.w = w; a.T = T;
    a.cpost = cpost; a.cshift2 = cshift2;
    a.dense = p.dense ? (dn && dn->factored ? 2 : 1) : 0; a.u_col = p.u_col; a.u_same = p.u_same; a.x0_keep = p.x0_keep;
    a.lnP = lnP; a.D = D; a.ldd = ldd; a.TH = TH; a.ldt = ldt;
    for (int i = 0; i < (int)p.seg.size(); ++i) a.seg[i] = p.seg[i];
    a.stamps = nullptr; a.gate = gate;
    if (mv) {
        a.mv_coords = mv->coords; a.mv_ldc = mv->ldc; a.mv_logp = mv->logp; a.mv_S = mv->S;
        a.mv_cc = mv->cc; a.mv_ldcc = mv->ldcc; a.mv_C = mv->C; a.mv_nc = mv->nc;
        a.mv_seed = mv->seed; a.mv_step = mv->step; a.mv_step_off = mv->step_off; a.mv_stream = mv->stream; a.mv_a = mv->a; a.mv_naccept = mv->naccept;
        a.mv_chain = mv->chain; a.mv_lps = mv->lps;
        a.sl_Z0 = mv->sl_Z0; a.sl_L = mv->sl_L; a.sl_R = mv->sl_R; a.sl_Zt = mv->sl_Zt; a.sl_m = mv->sl_m; a.sl_nt = mv->sl_nt;
        a.sl_seed = mv->sl_seed; a.sl_step = mv->sl_step; a.sl_stream = mv->sl_stream; a.sl_flags = mv->sl_flags;
        if (mv->sb) a.sb = *mv->sb;
        if (mv->slice) return ns_launch_kernel<2, false>(a, B, p, rows, s);
        return ns_launch_kernel<1, false>(a, B, p, rows, s);
    }
    if (gr) {
        a.gscale = gr->gscale; a.Gout = gr->G; a.ldg = gr->ldg;
        a.hm_p = gr->hm_p; a.hm_ldp = gr->hm_ldp; a.hm_q = gr->hm_q; a.hm_mass = gr->hm_mass; a.hm_ek = gr->hm_ek; a.hm_ed = gr->hm_ed;
        return ns_launch_kernel<0, true>(a, B, p, rows, s);
    }
    return ns_launch_kernel<0, false>(a, B, p, rows, s);
}

// Training / validation forward: X[B][ldx] (transformed inputs) -> every op's output in global memory.
// `y[i]`, `ldy[i]`: destination of op i's output; `t[i]`, `ldt[i]`: of the hidden h of residual block i.
int launch_net_stream_store(const linna_layer_t* layers, int nl, int in_size, const float* packed, const float* X, int ldx,
                            int B, float* const* y, const int* ldy, float* const* t, const int* ldt, const float* cscale,
                            const float* cshift, int rows, hipStream_t s) {
    const NsProgram& p = ns_build_prog(layers, nl, in_size, 0);
    if (!p.ok) { set_error("net_stream: network not eligible"); return LINNA_ERR_UNSUPPORTED; }
    NsArgs a;
    ::memset(static_cast<void*>(&a), 0, sizeof(a));
    a.Z = X; a.ldz = ldx; a.B = B; a.nin = in_size;
    // the prologue's transform constants are loaded (and ignored): any readable arrays of >= in_size entries
    a.is_flat = reinterpret_cast<const int*>(X); a.a1 = X; a.a2 = X; a.lg = nullptr; a.xmean = X; a.xstd = X;
    a.packed = packed;
    a.Gstride = p.Gstride; a.nseg_f = p.nseg_f; a.G = p.G; a.nseg = p.nseg_f;
    a.LD = p.LD; a.kpad0 = p.kpad0; a.nout = p.nout; a.bias_total = p.bias_total;
    a.T = 1.f;
    a.cscale = cscale; a.cshift = cshift;                   // column affine of the last output (Y transforms), or null
    for (int i = 0; i < (int)p.seg.size(); ++i) a.seg[i] = p.seg[i];
    for (int i = 0; i < p.nseg_f; ++i) {
        const int op = p.seg_op[i];
        if (p.seg_hidden[i]) { a.gout[i] = t[op]; a.gld[i] = ldt[op]; a.gn[i] = layers[op].C; }
        else { a.gout[i] = y[op]; a.gld[i] = ldy[op]; a.gn[i] = layers[op].N; }
    }
    return ns_launch_kernel<0, false, 1>(a, B, p, rows, s);
}


// lnP and d lnP / d z in ONE launch for any network the forward and dX-chain programs cover (GRAD + STORE == 2): forward
// segments store the activations the gates need (`y[i]` / `t[i]`: output of op i / hidden h of residual block i, in the
// caller's workspace), the turnaround forms d lnP / d out, the dX chain runs down to the network input gating on those
// activations, the finish applies the prior map's derivative.  Diagonal covariance.
int launch_net_stream_grad2(const linna_layer_t* layers, int nl, int in_size, const float* packed, const float* Z, int ldz, int B,
                            int nin, const int* is_flat, const float* a1, const float* a2, const int* lg, const float* xmean,
                            const float* xstd, const float* cscale, const float* cshift, const float* w, float T, float* lnP,
                            const NsGrad& gr, float* const* y, const int* ldy, float* const* t, const int* ldt, int rows,
                            hipStream_t s) {
    const NsProgram& p = ns_build_prog(layers, nl, in_size, 3, nullptr, rows == 16 && net_stream_g2_side());
    if (!p.ok || !p.dxi_ok) { set_error("net_stream: no forward + dX program for this network"); return LINNA_ERR_UNSUPPORTED; }
    if (!w || !lnP || !gr.gscale || !gr.G) { set_error("net_stream: the one-launch gradient needs a diagonal covariance"); return LINNA_ERR_INVALID; }
    NsArgs a;
    ::memset(static_cast<void*>(&a), 0, sizeof(a));
    a.Z = Z; a.ldz = ldz; a.B = B; a.nin = nin;
    a.is_flat = is_flat; a.a1 = a1; a.a2 = a2; a.lg = lg; a.xmean = xmean; a.xstd = xstd;
    a.packed = packed;
    a.Gstride = p.Gstride; a.nseg_f = p.nseg_f; a.G = p.Gstride; a.nseg = (int)p.seg.size();
    a.LD = p.LD; a.kpad0 = p.kpad0; a.nout = p.nout; a.bias_total = p.bias_total;
    a.cscale = cscale; a.cshift = cshift; a.w = w; a.T = T; a.lnP = lnP;
    a.gscale = gr.gscale; a.Gout = gr.G; a.ldg = gr.ldg;
    a.hm_p = gr.hm_p; a.hm_ldp = gr.hm_ldp; a.hm_q = gr.hm_q; a.hm_mass = gr.hm_mass; a.hm_ek = gr.hm_ek; a.hm_ed = gr.hm_ed;
    for (int i = 0; i < (int)p.seg.size(); ++i) a.seg[i] = p.seg[i];
    // What the backward gates on is the SIGN of a forward activation, and the workgroup that needs it is the one that
    // computed it: one bit per (row, column) in LDS (NsArgs::nbw).  Round 2 kept the activations themselves in the caller's
    // workspace (y / t: 40 MB written and read back per 4096-chain launch of ChtoModelv2(33,33), in bursts at the run ends of
    // 256 workgroups in step, each read an exposed L2 round trip behind a drained weight ring); gout / gmask are flags now.
    (void)y; (void)ldy; (void)t; (void)ldt;
    std::vector<int> base_y(nl, -1), base_t(nl, -1);
    int ncolbits = 0;
    for (int i = 0; i < NS_MAXSEG; ++i) a.gbit[i] = a.mbit[i] = -1;
    for (int i = 0; i < p.nseg_f; ++i) {                      // forward: keep the signs a gate will ask for
        const int op = p.seg_op[i];
        if (p.seg_hidden[i]) { base_t[op] = ncolbits; a.gn[i] = layers[op].C; }
        else if (op < nl - 1) { base_y[op] = ncolbits; a.gn[i] = layers[op].N; }
        else continue;
        a.gbit[i] = ncolbits;
        ncolbits += (a.gn[i] + 63) & ~63;
    }
    for (int i = p.nseg_f; i < (int)p.seg.size(); ++i) {
        const int op = p.seg_op[i];
        if (p.seg_hidden[i]) {                                // d/dh of residual block op: gated by its h
            a.mbit[i] = base_t[op]; a.gn[i] = layers[op].C;
            if (a.mbit[i] < 0) { set_error("net_stream: a gate of the one-launch gradient has no producer"); return LINNA_ERR_UNSUPPORTED; }
        } else {                                              // d/d(input of op): gated by the producing op's output, if it went through a ReLU
            const bool relu_in = op > 0 && (layers[op - 1].op == LINNA_OP_RESBLOCK || layers[op - 1].relu);
            a.mbit[i] = relu_in ? base_y[op - 1] : -1; a.gn[i] = layers[op].K;
            if (relu_in && a.mbit[i] < 0) { set_error("net_stream: a gate of the one-launch gradient has no producer"); return LINNA_ERR_UNSUPPORTED; }
        }
    }
    a.nbw = ncolbits / 32;
    const size_t lds0 = (p.lds_for(rows, true) + 7) & ~(size_t)7;
    a.bits_off = (int)(lds0 / sizeof(float));
    const size_t lds_bits = (size_t)rows * a.nbw * sizeof(unsigned);
    if (lds0 + lds_bits > (size_t)NS_LDS_BYTES) { set_error("net_stream: the one-launch gradient's sign bits do not fit the LDS"); return LINNA_ERR_UNSUPPORTED; }
    return ns_launch_kernel<0, true, 2>(a, B, p, rows, s, 0, lds0 + lds_bits - p.lds_for(rows, true));
}

// Training forward + loss in one launch (STORE == 3): X[n][ldx] the resident set, ROWS the batch (null: rows 0..B-1);
// `dn` = {Cinv, ldc, null, null}: the inverse covariance in the network's normalised output space as the last segment.
int launch_net_stream_train(const linna_layer_t* layers, int nl, int in_size, const float* packed, const float* X, int ldx,
                            const int* ROWS, int B, const int* lg, const float* xmean, const float* xstd, float* XB, int ldxb,
                            float* const* y, const int* ldy, float* const* t, const int* ldt, const NsTrainLoss& L,
                            const NsDense& dn, int rows, hipStream_t s) {
    const NsProgram& p = ns_build_prog(layers, nl, in_size, 0, &dn);
    if (!p.ok || !p.dense) { set_error("net_stream: network + loss not eligible"); return LINNA_ERR_UNSUPPORTED; }
    NsArgs a;
    ::memset(static_cast<void*>(&a), 0, sizeof(a));
    a.Z = X; a.ldz = ldx; a.B = B; a.nin = in_size;
    a.is_flat = reinterpret_cast<const int*>(xmean); a.a1 = xmean; a.a2 = xmean;     // loaded and ignored
    a.lg = lg; a.xmean = xmean; a.xstd = xstd;
    a.packed = packed;
    a.Gstride = p.Gstride; a.nseg_f = p.nseg_f; a.G = p.G; a.nseg = p.nseg_f;
    a.LD = p.LD; a.kpad0 = p.kpad0; a.nout = p.nout; a.bias_total = p.bias_total;
    a.T = 1.f;
    a.dense = p.dense; a.u_col = p.u_col; a.u_same = p.u_same;
    for (int i = 0; i < (int)p.seg.size(); ++i) a.seg[i] = p.seg[i];
    for (int i = 0; i < p.nseg_f; ++i) {
        const int op = p.seg_op[i];
        if (op >= nl) continue;                                   // the loss segment stores nothing
        if (p.seg_hidden[i]) { a.gout[i] = t[op]; a.gld[i] = ldt[op]; a.gn[i] = layers[op].C; }
        else { a.gout[i] = y[op]; a.gld[i] = ldy[op]; a.gn[i] = layers[op].N; }
    }
    a.t_rows = ROWS; a.t_xb = XB; a.t_ldxb = ldxb;
    a.t_Y = L.YN; a.t_ldy = L.ldyn;
    a.t_den = L.den; a.t_inv_batch = L.inv_batch; a.t_loss_rows = L.loss_rows; a.t_dP = L.dP; a.t_lddp = L.lddp;
    return ns_launch_kernel<0, false, 3>(a, B, p, rows, s);
}

// A training step's network work in ONE launch (TRB: GRAD + STORE == 3): launch_net_stream_train's forward + loss, its
// finish as the turnaround, then launch_net_stream_dx's chain down to op 1 -- same arguments as the two of them.  `post`:
// only the AdamW step constants ride here (the loss rows are not complete before every workgroup's turnaround: the batch
// mean rides in the parameter-gradient launch instead).
int launch_net_stream_train_bwd(const linna_layer_t* layers, int nl, int in_size, const float* packed, const float* X, int ldx,
                                const int* ROWS, int B, const int* lg, const float* xmean, const float* xstd, float* XB, int ldxb,
                                float* const* y, const int* ldy, float* const* t, const int* ldt, const NsTrainLoss& L,
                                const NsDense& dn, float* const* dprev, const int* ldp, const float* const* hin, const int* ldh,
                                float* const* dt, const int* lddt, int rows, hipStream_t s, const NsPost* post) {
    const NsProgram& p = ns_build_prog(layers, nl, in_size, 4, &dn);
    if (!p.ok || !p.train_ok || !p.dense) { set_error("net_stream: network + loss have no one-launch training program"); return LINNA_ERR_UNSUPPORTED; }
    if (rows != 4) { set_error("net_stream: the one-launch training step runs on the 4-row engine"); return LINNA_ERR_UNSUPPORTED; }
    NsArgs a;
    ::memset(static_cast<void*>(&a), 0, sizeof(a));
    a.Z = X; a.ldz = ldx; a.B = B; a.nin = in_size;
    a.is_flat = reinterpret_cast<const int*>(xmean); a.a1 = xmean; a.a2 = xmean;     // loaded and ignored
    a.lg = lg; a.xmean = xmean; a.xstd = xstd;
    a.packed = packed;
    a.Gstride = p.Gstride; a.nseg_f = p.nseg_f; a.G = p.Gstride; a.nseg = (int)p.seg.size();
    a.LD = p.LD; a.kpad0 = p.kpad0; a.nout = p.nout; a.bias_total = p.bias_total;
    a.T = 1.f;
    a.dense = p.dense; a.u_col = p.u_col; a.u_same = p.u_same;
    for (int i = 0; i < (int)p.seg.size(); ++i) a.seg[i] = p.seg[i];
    for (int i = 0; i < (int)p.seg.size(); ++i) {
        const int op = p.seg_op[i];
        if (i < p.nseg_f) {
            if (op >= nl) continue;                               // the loss segment stores nothing
            if (p.seg_hidden[i]) { a.gout[i] = t[op]; a.gld[i] = ldt[op]; a.gn[i] = layers[op].C; }
            else { a.gout[i] = y[op]; a.gld[i] = ldy[op]; a.gn[i] = layers[op].N; }
        } else if (p.seg_hidden[i]) {                             // d/dh of residual block op, gated by its stored h
            a.gout[i] = dt[op]; a.gld[i] = lddt[op]; a.gn[i] = layers[op].C; a.gmask[i] = t[op]; a.gmld[i] = ldt[op];
        } else {                                                  // d/d(input of op), gated by the producing op's output
            a.gout[i] = dprev[op]; a.gld[i] = ldp[op]; a.gn[i] = layers[op].K; a.gmask[i] = hin[op]; a.gmld[i] = ldh[op];
        }
    }
    // the gates of the backward half: sign bits in LDS (see launch_net_stream_grad2) -- the activations go to memory all the
    // same (the parameter gradients read them), but no gate is read back from there
    std::vector<int> base_y(nl, -1), base_t(nl, -1);
    int ncolbits = 0;
    for (int i = 0; i < NS_MAXSEG; ++i) a.gbit[i] = a.mbit[i] = -1;
    for (int i = 0; i < p.nseg_f; ++i) {
        const int op = p.seg_op[i];
        if (op >= nl) continue;
        if (p.seg_hidden[i]) base_t[op] = ncolbits; else if (op < nl - 1) base_y[op] = ncolbits; else continue;
        a.gbit[i] = ncolbits;
        ncolbits += (a.gn[i] + 63) & ~63;
    }
    for (int i = p.nseg_f; i < (int)p.seg.size(); ++i) {
        const int op = p.seg_op[i];
        if (!a.gmask[i]) continue;
        a.mbit[i] = p.seg_hidden[i] ? base_t[op] : (op > 0 ? base_y[op - 1] : -1);
        if (a.mbit[i] < 0) { set_error("net_stream: a gate of the one-launch training step has no producer"); return LINNA_ERR_UNSUPPORTED; }
    }
    a.nbw = ncolbits / 32;
    const size_t lds0 = (p.lds_for(rows, true) + 7) & ~(size_t)7;
    a.bits_off = (int)(lds0 / sizeof(float));
    const size_t lds_bits = (size_t)rows * a.nbw * sizeof(unsigned);
    if (lds0 + lds_bits > (size_t)NS_LDS_BYTES) { set_error("net_stream: the training step's sign bits do not fit the LDS"); return LINNA_ERR_UNSUPPORTED; }
    a.t_rows = ROWS; a.t_xb = XB; a.t_ldxb = ldxb;
    a.t_Y = L.YN; a.t_ldy = L.ldyn;
    a.t_den = L.den; a.t_inv_batch = L.inv_batch; a.t_loss_rows = L.loss_rows; a.t_dP = L.dP; a.t_lddp = L.lddp;
    int extra = 0;
    if (post && post->step) { a.p_step = post->step; a.p_hyper = post->hyper; a.p_b1 = post->b1; a.p_b2 = post->b2; extra = 1; }
    return ns_launch_kernel<0, true, 3>(a, B, p, rows, s, extra, lds0 + lds_bits - p.lds_for(rows, true));
}
}  // namespace linna

namespace linna {
// The dX chain of a training step in one launch (what linna_net_backward otherwise runs as one GEMM per op):
// dOUT[B][lddo] -> for every op i >= first (1, or 0 with_input) the gradient with respect to its input, gated by the
// stored forward activation hin[i] (null: no gate), into dprev[i]; for residual blocks also d/dh into dt[i], gated by
// the stored h (t[i]).
int launch_net_stream_dx(const linna_layer_t* layers, int nl, int in_size, const float* packed, const float* dOUT, int lddo,
                         int B, float* const* dprev, const int* ldp, const float* const* hin, const int* ldh,
                         float* const* dt, const int* lddt, const float* const* t, const int* ldt, int with_input, int rows,
                         hipStream_t s, const NsPost* post) {
    const NsProgram& p = ns_build_prog(layers, nl, in_size, with_input ? 2 : 1);
    if (!p.ok) { set_error("net_stream: no dX-chain program for this network"); return LINNA_ERR_UNSUPPORTED; }
    NsArgs a;
    ::memset(static_cast<void*>(&a), 0, sizeof(a));
    const int nout = layers[nl - 1].N;
    a.Z = dOUT; a.ldz = lddo; a.B = B; a.nin = nout;
    a.is_flat = reinterpret_cast<const int*>(dOUT); a.a1 = dOUT; a.a2 = dOUT; a.lg = nullptr; a.xmean = dOUT; a.xstd = dOUT;
    a.packed = packed;
    a.Gstride = p.Gstride; a.nseg_f = p.nseg_f; a.G = p.G; a.nseg = p.nseg_f;
    a.LD = p.LD; a.kpad0 = p.kpad0; a.nout = p.nout; a.bias_total = p.bias_total;
    a.T = 1.f;
    for (int i = 0; i < (int)p.seg.size(); ++i) a.seg[i] = p.seg[i];
    for (int i = 0; i < p.nseg_f; ++i) {
        const int op = p.seg_op[i];
        if (p.seg_hidden[i]) { a.gout[i] = dt[op]; a.gld[i] = lddt[op]; a.gn[i] = layers[op].C; a.gmask[i] = t[op]; a.gmld[i] = ldt[op]; }
        else { a.gout[i] = dprev[op]; a.gld[i] = ldp[op]; a.gn[i] = layers[op].K; a.gmask[i] = hin[op]; a.gmld[i] = ldh[op]; }
    }
    if (post && post->n > 0) {                    // the rider (see NsArgs::p_rows): one more workgroup
        a.p_rows = post->rows; a.p_n = post->n; a.p_scale = post->scale; a.p_out = post->out;
        a.p_step = post->step; a.p_hyper = post->hyper; a.p_b1 = post->b1; a.p_b2 = post->b2;
        return ns_launch_kernel<0, false, 2>(a, B, p, rows, s, 1);
    }
    return ns_launch_kernel<0, false, 2>(a, B, p, rows, s);
}
}  // namespace linna
