// Internal declarations shared by the HIP translation units of liblinna_hip.so.
// gfx950 (MI355X / CDNA4) only: 64-lane wavefronts, fp32-input MFMA, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <exception>
#include "../../include/linna_hip.h"


typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace linna {

// ------------------------------------------------------------------ Philox4x32-10 (Salmon et al. 2011)
struct U4 { uint32_t x, y, z, w; };
__device__ __forceinline__ U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c.x;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c.z;
        U4 n;
        n.x = (uint32_t)(p1 >> 32) ^ c.y ^ k0;
        n.y = (uint32_t)p1;
        n.z = (uint32_t)(p0 >> 32) ^ c.w ^ k1;
        n.w = (uint32_t)p0;
        c = n;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return c;
}
__device__ __forceinline__ float u01(uint32_t b) { return ((float)(b >> 8) + 0.5f) * (1.0f / 16777216.0f); }

// draws for walker w at (step, stream): counter = (w, step, stream, sub), key = seed
__device__ __forceinline__ U4 walker_bits(uint64_t seed, uint32_t w, uint32_t step, uint32_t stream, uint32_t sub) {
    U4 c = {w, step, stream, sub};
    return philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
}

// ---- the logic of one shrinking round of linna_slice_half_step for walker k of the half ensemble (pointwise.hip's comment
// block "one-call half step" has the procedure).  Shared by slice_shrink_multi_kernel and by the tail of the whole-network
// kernel (net_stream.hip: the last workgroup of a round's evaluation runs this over the walkers, no launch of its own).
// counters: [0] expansions, [1] contractions, [2] walkers left unfinished by the rounds of a call (sticky),
//           [3] evaluated points, [4 + r] walkers still active after round r (expand rounds first, then shrink rounds)
// The set-up of a half step (slice_begin_kernel: differential-move direction, slice height, initial bracket, flags, the usage
// counters' roll) as the first stepping-out round's evaluation does it in its own prologue (net_stream.hip, MOVE == 2): row
// j ns + k forms its walker's direction and bracket end itself; the rows j = 0 also write them for the later launches.
struct SliceBegin {
    const float* logp; const float* cc; int ldcc; const int* C; int nc; const float* mu; uint64_t seed; const int* step; int half, m;
    float* DIR; int ldd; float* Z0; float* L; float* R; int* flags; int* counters; int nslots, zero_totals;
    int maxsteps;                                       // zeus' stepping-out budget (slice_budget)
};
// zeus' stepping-out budget (Neal 2003, Fig. 3; zeus ensemble.py: J = floor(maxsteps U), K = (maxsteps - 1) - J): at most J steps
// to the left and K to the right.  flags[3k] / flags[3k + 1] hold what is LEFT of them while that side is still stepping out and
// 0 once it is closed (an end below the slice, or the budget spent); flags[3k + 2]: still shrinking.
__device__ __forceinline__ void slice_budget(uint64_t seed, uint32_t wk, uint32_t step, uint32_t stream, int maxsteps, int& J, int& K) {
    const U4 b = walker_bits(seed, wk, step, stream, 1u);
    J = min((int)floorf((float)maxsteps * u01(b.x)), maxsteps - 1);       // (the fp32 product may round up to maxsteps itself)
    K = maxsteps - 1 - J;
}
// a side with `f` steps left saw `above` leading bracket ends over the slice among the m evaluated: steps taken; f becomes what is left
__device__ __forceinline__ int slice_side_steps(int& f, int above, int m) {
    const int n = min(above, f);
    f = (n == m && n < f) ? f - n : 0;
    return n;
}
struct SliceRound {
    const float* Z0; const float* Zt;                  // slice heights [ns]; lnP of this round's trials, [j ns + k]
    float* L; float* R; const int* S; float* W;        // brackets, the half ensemble's walkers, trial weights [j ns + k] (read; the next round's written)
    int* flags; float* Wacc; float* Zacc; int ns;
    int* counters; int slot, prev_slot, ntrial, nt_next, trials_so_far;
    int* list; uint64_t seed; int* step_dev; int stream_id;
    float* coords; int ldc, ndim; float* logp; const float* DIR; int ldd, bump;   // coords != null: the call's last round commits
    // m_derive > 0: this is the first shrinking round behind ONE stepping-out round of m_derive ends per side whose logic kernel
    // was not launched (the evaluation derived its trials itself, NsArgs::sl_*): its bookkeeping is done here first, from Ze
    const float* Ze; int m_derive, eslot;
};
__device__ __forceinline__ void slice_draw_dev(int k, int wk, float l, float r, float* __restrict__ W, int ns, uint64_t seed,
                                               uint32_t step, int stream_id, int round, int ntrial) {
    for (int j = 0; j < ntrial; ++j) {
        const U4 b = walker_bits(seed, (uint32_t)wk, step, (uint32_t)stream_id, (uint32_t)(round + j + 1));
        const float w = l + u01(b.x) * (r - l);
        W[(size_t)j * ns + k] = w;
        if (w < 0.f) l = w; else r = w;
    }
}
// stepping out over the m ends per side one round evaluated (Zt[j ns + k]: lnP at L - j, j < m, and at R + (j - m)); returns
// whether the walker is still stepping out
__device__ __forceinline__ bool slice_expand_walker(int k, int ns, int m, float z0, const float* __restrict__ Zt, float& l, float& r,
                                                    int* __restrict__ flags, int* __restrict__ counters) {
    int n = 0;
    int fl = flags[3 * k], fr = flags[3 * k + 1];
    if (fl) {
        int j = 0;
        for (; j < m; ++j) { if (!(Zt[(size_t)j * ns + k] > z0)) break; }
        const int nl = slice_side_steps(fl, j, m);
        for (int i = 0; i < nl; ++i) l -= 1.f;
        n += nl;
        flags[3 * k] = fl;
    }
    if (fr) {
        int j = 0;
        for (; j < m; ++j) { if (!(Zt[(size_t)(m + j) * ns + k] > z0)) break; }
        const int nr = slice_side_steps(fr, j, m);
        for (int i = 0; i < nr; ++i) r += 1.f;
        n += nr;
        flags[3 * k + 1] = fr;
    }
    if (n) atomicAdd(counters + 0, n);
    return (flags[3 * k] | flags[3 * k + 1]) != 0;
}
__device__ __forceinline__ void slice_round_walker(const SliceRound& a, int k) {
    const int ns = a.ns;
    if (k == 0) {
        if (a.m_derive) atomicAdd(a.counters + 3, 2 * a.m_derive * ns);
        atomicAdd(a.counters + 3, a.ntrial * (a.prev_slot < 0 ? ns : a.counters[a.prev_slot]));      // the points this round evaluated
    }
    if (k >= ns) return;
    const uint32_t step = (uint32_t)a.step_dev[0];
    if (a.m_derive) {
        float l = a.L[k], r = a.R[k];
        const bool out = slice_expand_walker(k, ns, a.m_derive, a.Z0[k], a.Ze, l, r, a.flags, a.counters);
        a.L[k] = l; a.R[k] = r;
        if (out) atomicAdd(a.counters + a.eslot, 1);
        else slice_draw_dev(k, a.S[k], l, r, a.W, ns, a.seed, step, a.stream_id, 0, a.ntrial);   // the trials the evaluation derived
    }
    const bool mine = a.flags[3 * k + 2] && !(a.flags[3 * k] | a.flags[3 * k + 1]) &&      // not done, and its bracket closed
                      !(a.prev_slot >= 0 && a.counters[a.prev_slot] == 0);
    if (mine) {
        int ncon = 0;
        bool active = true;
        float l = a.L[k], r = a.R[k];
        const float z0 = a.Z0[k];
        for (int j = 0; j < a.ntrial && active; ++j) {
            const float zt = a.Zt[(size_t)j * ns + k], w = a.W[(size_t)j * ns + k];
            if (!(z0 < zt)) {                               // zeus accepts iff Z0 < lnP(x'); NaN rejects
                if (w < 0.f) l = w; else r = w;
                ++ncon;
                if (r - l < 1e-30f) { active = false; a.Wacc[k] = 0.f; a.Zacc[k] = z0; }   // degenerate: stay put
            } else {
                active = false; a.Wacc[k] = w; a.Zacc[k] = zt;
            }
        }
        a.L[k] = l; a.R[k] = r;
        if (ncon) atomicAdd(a.counters + 1, ncon);
        if (active) {
            const int pos = atomicAdd(a.counters + a.slot, 1);
            slice_draw_dev(k, a.S[k], l, r, a.W, ns, a.seed, step, a.stream_id, a.trials_so_far, a.nt_next);   // the next round's trials
            for (int j = 0; j < a.nt_next; ++j) a.list[(size_t)pos * a.nt_next + j] = j * ns + k;
        } else {
            a.flags[3 * k + 2] = 0;
        }
    }
    if (a.coords) {
        // the move of every finished walker (slice_commit_checked_kernel's arithmetic); a walker the rounds of the call left
        // unfinished stays where it is and is counted; `bump` advances the device step counter behind an iteration's second half step
        if (a.flags[3 * k] | a.flags[3 * k + 1] | a.flags[3 * k + 2]) {
            atomicAdd(a.counters + 2, 1);
        } else if (a.Wacc[k] != 0.f) {
            const int wk = a.S[k];
            const float wa = a.Wacc[k];
            for (int d = 0; d < a.ndim; ++d) a.coords[(size_t)wk * a.ldc + d] += wa * a.DIR[(size_t)k * a.ldd + d];
            a.logp[wk] = a.Zacc[k];
        }
        // (the last round draws no further trials: no thread of this launch uses the counter's value, whichever it reads)
        if (a.bump && k == 0) a.step_dev[0] = (int)step + 1;
    }
}

// counters[i] += the sum over the block's waves of v (wave-uniform); every thread of the block calls
__device__ __forceinline__ void block_add_counter(int* __restrict__ counter, int v, int* lds_slot) {
    if (threadIdx.x == 0) *lds_slot = 0;
    __syncthreads();
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(lds_slot, v);
    __syncthreads();
    if (threadIdx.x == 0 && *lds_slot) atomicAdd(counter, *lds_slot);
}
// ---- the same logic with a WAVE per walker (all 64 lanes call with the same k): the bracket ends / trials of a round are loaded,
// and the Philox draws of the next round made, one per lane; what is sequential in the procedure (a bracket shrinking trial by
// trial) is a short uniform loop over register values.  One thread per walker ran 16-32 dependent Philox draws and as many
// dependent loads: 5-10 us per logic kernel, a third of a 128-walker iteration.  Same arithmetic in the same order: same chain.
__device__ __forceinline__ float wave_lane_f(float v, int j) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j)); }
// trial j + 1 of (round0 ...) placed as if its predecessors were rejected (slice_draw_dev); lane j < cnt returns w_j
__device__ __forceinline__ float slice_draw_wave(int lane, int wk, float l, float r, uint64_t seed, uint32_t step, int stream_id,
                                                 int round0, int cnt) {
    const U4 b = walker_bits(seed, (uint32_t)wk, step, (uint32_t)stream_id, (uint32_t)(round0 + lane + 1));
    const float u = u01(b.x);
    float mine = 0.f;
    for (int j = 0; j < cnt; ++j) {
        const float w = l + wave_lane_f(u, j) * (r - l);
        if (lane == j) mine = w;
        if (w < 0.f) l = w; else r = w;
    }
    return mine;
}
// stepping out over m <= 32 ends per side: lanes 0..m-1 the left ends, 32..32+m-1 the right ones; fl / fr updated
__device__ __forceinline__ void slice_expand_wave(int lane, int k, int ns, int m, float z0, const float* __restrict__ Zt, float& l, float& r,
                                                  int& fl, int& fr, int* __restrict__ flags, int& nexp) {
    const int side = lane >> 5, j = lane & 31;
    const float ze = j < m ? Zt[(size_t)(side * m + j) * ns + k] : 0.f;
    const unsigned long long bal = __ballot(j < m && ze > z0);
    const int fl0 = fl, fr0 = fr;
    const int nl = slice_side_steps(fl, __builtin_ctzll(~(unsigned long long)(uint32_t)bal), m);
    const int nr = slice_side_steps(fr, __builtin_ctzll(~(unsigned long long)(uint32_t)(bal >> 32)), m);
    for (int i = 0; i < nl; ++i) l -= 1.f;
    for (int i = 0; i < nr; ++i) r += 1.f;
    if (lane == 0 && fl != fl0) flags[3 * k] = fl;
    if (lane == 0 && fr != fr0) flags[3 * k + 1] = fr;
    nexp += nl + nr;
}
// nexp / ncon: this walker's expansions / contractions, which the caller adds to counters[0] / [1] (summed over the block first:
// one atomic per walker on one address cost 10-15 us of a 4096-walker round)
__device__ __forceinline__ void slice_round_wave(const SliceRound& a, int k, int lane, int& nexp, int& ncon_out) {
    const int ns = a.ns;
    if (a.ntrial > 64 || a.nt_next > 64 || a.m_derive > 32) {       // (schedules beyond a wave's lanes: one lane, the plain procedure)
        if (lane == 0) slice_round_walker(a, k);                    //  -- counts itself
        return;
    }
    if (k == 0 && lane == 0) {
        if (a.m_derive) atomicAdd(a.counters + 3, 2 * a.m_derive * ns);
        atomicAdd(a.counters + 3, a.ntrial * (a.prev_slot < 0 ? ns : a.counters[a.prev_slot]));      // the points this round evaluated
    }
    if (k >= ns) return;
    const uint32_t step = (uint32_t)a.step_dev[0];
    const int wk = a.S[k];
    const float z0 = a.Z0[k];
    int fl = a.flags[3 * k], fr = a.flags[3 * k + 1], fs = a.flags[3 * k + 2];
    float l = a.L[k], r = a.R[k];
    float w = 0.f;                                     // lane j: trial j of this round
    bool have_w = false;
    if (a.m_derive) {
        slice_expand_wave(lane, k, ns, a.m_derive, z0, a.Ze, l, r, fl, fr, a.flags, nexp);
        if (lane == 0) { a.L[k] = l; a.R[k] = r; }
        if (fl | fr) {
            if (lane == 0) atomicAdd(a.counters + a.eslot, 1);
        } else {
            w = slice_draw_wave(lane, wk, l, r, a.seed, step, a.stream_id, 0, a.ntrial);   // the trials the evaluation derived
            if (lane < a.ntrial) a.W[(size_t)lane * ns + k] = w;
            have_w = true;
        }
    }
    const bool mine = fs && !(fl | fr) && !(a.prev_slot >= 0 && a.counters[a.prev_slot] == 0);   // not done, and its bracket closed
    float wacc = 0.f, zacc = 0.f;
    bool done_now = false;
    if (mine) {
        const bool in = lane < a.ntrial;
        const float zt = in ? a.Zt[(size_t)lane * ns + k] : 0.f;
        if (!have_w) w = in ? a.W[(size_t)lane * ns + k] : 0.f;
        const unsigned long long okm = __ballot(in && z0 < zt);          // zeus accepts iff Z0 < lnP(x'); NaN rejects
        const int ja = okm ? __builtin_ctzll(okm) : a.ntrial;          // the first trial inside the slice
        int ncon = 0;
        bool active = true;
        for (int j = 0; j < ja; ++j) {
            const float wj = wave_lane_f(w, j);
            if (wj < 0.f) l = wj; else r = wj;
            ++ncon;
            if (r - l < 1e-30f) { active = false; wacc = 0.f; zacc = z0; break; }   // degenerate: stay put
        }
        if (active && ja < a.ntrial) { active = false; wacc = wave_lane_f(w, ja); zacc = wave_lane_f(zt, ja); }
        if (lane == 0) { a.L[k] = l; a.R[k] = r; }
        ncon_out += ncon;
        if (active) {
            int pos = 0;
            if (lane == 0) pos = atomicAdd(a.counters + a.slot, 1);
            pos = __builtin_amdgcn_readfirstlane(pos);
            const float wn = slice_draw_wave(lane, wk, l, r, a.seed, step, a.stream_id, a.trials_so_far, a.nt_next);   // the next round's trials
            if (lane < a.nt_next) {
                a.W[(size_t)lane * ns + k] = wn;
                a.list[(size_t)pos * a.nt_next + lane] = lane * ns + k;
            }
        } else {
            done_now = true;
            fs = 0;
            if (lane == 0) { a.flags[3 * k + 2] = 0; a.Wacc[k] = wacc; a.Zacc[k] = zacc; }
        }
    }
    if (a.coords) {
        if (fl | fr | fs) {
            if (lane == 0) atomicAdd(a.counters + 2, 1);
        } else {
            if (!done_now) { wacc = a.Wacc[k]; zacc = a.Zacc[k]; }
            if (wacc != 0.f) {
                for (int d = lane; d < a.ndim; d += 64) a.coords[(size_t)wk * a.ldc + d] += wacc * a.DIR[(size_t)k * a.ldd + d];
                if (lane == 0) a.logp[wk] = zacc;
            }
        }
        if (a.bump && k == 0 && lane == 0) a.step_dev[0] = (int)step + 1;
    }
}

void set_error(const char* fmt, ...);
int check_hip(hipError_t e, const char* what);
// The exception barrier of the C ABI (include/linna_hip.h: "No C++ exception crosses the boundary").  Every extern "C"
// entry is a function-try-block closed by one of these: the host side uses std::vector / std::string / std::unordered_map
// (program planning, descriptor tables), so bad_alloc / length_error can be thrown under an entry; it becomes a return
// code and a text in linna_last_error() instead of unwinding into ctypes (= std::terminate in the caller's process).
int caught_exception(const char* what) noexcept;      // sets the text, returns LINNA_ERR_INTERNAL (api.hip)
#define LINNA_CATCH_INT \
    catch (const std::exception& e_) { return ::linna::caught_exception(e_.what()); } \
    catch (...) { return ::linna::caught_exception(nullptr); }
#define LINNA_CATCH_SIZE \
    catch (const std::exception& e_) { (void)::linna::caught_exception(e_.what()); return 0; } \
    catch (...) { (void)::linna::caught_exception(nullptr); return 0; }

// Operand storage seen by the GEMM: LAY_K = contraction index contiguous
// (A as [M][K], B as [N][K]); LAY_MN = k-major (A as [K][M], B as [K][N]).
enum { LAY_K = 0, LAY_MN = 1 };

typedef linna_gemm_pair_t GemmPair;
typedef linna_gemm_t GemmArgs;   // the public descriptor is passed to the kernel by value

// pointwise.hip launchers
int launch_prior_map_fwd(const float* Z, int ldz, int B, int nin, const int* is_flat, const float* a1, const float* a2,
                         const int* lg, const float* xmean, const float* xstd, float* X, int ldx, float* TH, int ldt,
                         hipStream_t s);
int launch_prior_map_bwd(const float* Z, int ldz, int B, int nin, const int* is_flat, const float* a1, const float* a2,
                         const int* lg, const float* xstd, const float* dX, int lddx, float* dZ, int lddz, hipStream_t s);
int launch_loglike_diag(const float* D, int ldd, int B, int nout, const float* w, const float* Z, int ldz, int nin,
                        float T, float* out, hipStream_t s);
int launch_loglike_finish(const float* partial, int slots_ld, int nslots, int B, const float* Z, int ldz, int nin,
                          float T, float* out, hipStream_t s);
int launch_loglike_diag_grad(const float* D, int ldd, int B, int nout, const float* w, const float* gscale, float T,
                             float* dH, int lddh, hipStream_t s);
int launch_loss_delta(int mode, const float* PRED, int ldp, const float* Y, int ldy, const int* ROWS, int B,
                      const linna_loss_desc_t& d, float* DELTA, int ldd, hipStream_t s);
int launch_loss_fused_small(const float* PRED, int ldp, const float* Y, int ldy, const int* ROWS, int B,
                            const linna_loss_desc_t& d, const float* den, float inv_batch, float* loss_rows, float* loss_mean,
                            float* dP, int lddp, unsigned* counter, hipStream_t s);
int launch_loss_rows(int mode, const float* partial, int slots_ld, int nslots, int B, const float* den, const int* ROWS,
                     float floorv, float* out, hipStream_t s);
int launch_loss_grad(const float* U, int ldu, const float* Y, int ldy, const int* ROWS, int B, int nout,
                     const float* data_norm, const float* den, float inv_batch, float* dP, int lddp, hipStream_t s);
int launch_sum_scale(const float* v, int n, float scale, float* out, hipStream_t s);
int launch_sum_scale_prepare(const float* v, int n, float scale, float* out, int* step_dev, float* hyper, float b1, float b2,
                             hipStream_t s);
int launch_val_frac(const float* partial, int slots_ld, int nslots, int B, const float* den, float* frac, hipStream_t s);
int launch_val_metrics(const float* loss, const float* frac, int n, const float* last, float* out, hipStream_t s);
int launch_gather_xform(const float* X, int ldx, const int* ROWS, int B, int nin, const int* lg, const float* xmean,
                        const float* xstd, float* XB, int ldxb, hipStream_t s);
int launch_colsum(const float* dZ, int ld, int B, int N, float scale, float* db, hipStream_t s);
int launch_adamw(float* p, const float* g, float* m, float* v, size_t n, float* hyper, int* step_dev, float b1, float b2,
                 float eps, hipStream_t s);
int launch_adamw_prepare(float* hyper, int* step_dev, float b1, float b2, hipStream_t s);
int launch_stretch_propose(const float* coords, int ldc, int ndim, const int* S, int ns, const float* ccoords, int ldcc,
                           const int* C, int nc, uint64_t seed, const int* step_dev, int stream_id, float a, float* Q,
                           int ldq, float* factors, hipStream_t s);
int launch_stretch_accept(float* coords, int ldc, int ndim, float* logp, const int* S, int ns, const float* Q, int ldq,
                          const float* lp_new, const float* factors, uint64_t seed, const int* step_dev, int stream_id,
                          int* naccept, hipStream_t s);
int launch_hmc_init(int B, int ndim, const float* mass, uint64_t seed, const int* step_dev, const float* lnp,
                    const float* P0, int ldp0, float* P, int ldp, float* H0, hipStream_t s);
int launch_hmc_start(int B, int ndim, const float* mass, uint64_t seed, const int* step_dev, const float* lnp, const float* P0,
                     int ldp0, const float* G, int ldg, float ek, float ed, const float* X, int ldx, float* P, int ldp, float* Q,
                     int ldq, float* H0, hipStream_t s);
int launch_hmc_kick_drift(int B, int ndim, const float* mass, float ek, float ed, const float* G, int ldg, float* P,
                          int ldp, float* Q, int ldq, hipStream_t s);
int launch_hmc_accept(int B, int ndim, const float* mass, uint64_t seed, const int* step_dev, const float* H0,
                      const float* P, int ldp, const float* Qn, int ldq, const float* lnp_new, const float* Gn, int ldg,
                      const float* U, float* X, int ldx, float* lnp, float* G, int* naccept, hipStream_t s);
int launch_slice_init(const float* logp, const int* S, int ns, const float* cc, int ldcc, const int* C, int nc, int ndim,
                      const float* mu, uint64_t seed, const int* step_dev, int stream_id, float* DIR, int ldd, float* Z0,
                      float* L, float* R, int* flags, int maxsteps, hipStream_t s);
int launch_slice_points(const float* coords, int ldc, int ndim, const int* S, int ns, const float* DIR, int ldd,
                        const float* w, float* Q, int ldq, int nrep, hipStream_t s);
int launch_slice_expand(const float* Z0, const float* ZL, const float* ZR, float* L, float* R, int* flags, int ns,
                        int* counters, int slot, hipStream_t s);
int launch_slice_draw(const float* L, const float* R, const int* S, float* W, const int* flags, int ns, uint64_t seed,
                      const int* step_dev, int stream_id, int round, int ntrial, hipStream_t s);
int launch_slice_shrink(const float* Z0, const float* Zt, float* L, float* R, const float* W, int* flags, float* Wacc,
                        float* Zacc, int ns, int* counters, int slot, int ntrial, hipStream_t s);
int launch_slice_begin(const float* logp, const int* S, int ns, const float* cc, int ldcc, const int* C, int nc, int ndim,
                       const float* mu, uint64_t seed, const int* step_dev, int stream_id, float* DIR, int ldd, float* Z0, float* L,
                       float* R, int* flags, float* W, int m, int* counters, int nslots, int zero_totals, int maxsteps, hipStream_t s);
int launch_slice_expand_multi(const float* Z0, const float* Zt, float* L, float* R, const int* S, int* flags, int ns, int m,
                              int m_next, int* counters, int slot, int prev_slot, float* W, float* Wd, int* list, uint64_t seed,
                              const int* step_dev, int stream_id_shrink, int ntrial, hipStream_t s);
int launch_slice_shrink_multi(const SliceRound& a, hipStream_t s);
int launch_slice_commit_checked(float* coords, int ldc, int ndim, float* logp, const int* S, int ns, const float* DIR, int ldd,
                                const float* Wacc, const float* Zacc, const int* flags, int* counters, hipStream_t s);
int launch_slice_commit(float* coords, int ldc, int ndim, float* logp, const int* S, int ns, const float* DIR, int ldd,
                        const float* Wacc, const float* Zacc, hipStream_t s);
int launch_step_increment(int* step, hipStream_t s);

// Dense inverse covariance for the whole-network kernel: the output map d = raw * cscale + cshift is folded into the last
// layer of the weight stream and S (symmetric, [nout][lds]) is appended as one more segment, chi2 = d . (d S).
struct NsDense { const float* S; int lds; const float* cscale; const float* cshift; int factored = 0;   // factored: S holds L (S = L L^T), chi2 = |d L|^2
                 int tri = 2; };         // how the factor's zero upper triangle is skipped (net_stream_dense_tri): 0 not, 1 short second pass, 2 balanced blocks
// process-wide default of NsDense::tri for log-probability objects created afterwards (LINNA_DENSE_TRI); returns the previous value
int net_stream_dense_tri(int mode);
// net_stream.hip (program-driven whole-network kernel: residual blocks, widths up to 1024)
bool net_stream_eligible(const linna_layer_t* layers, int nl, int in_size);
size_t net_stream_packed_floats(const linna_layer_t* layers, int nl, int in_size);
// rows per workgroup for a batch of B rows: 16 (v_mfma_f32_16x16x4_f32), or 8 / 4 (v_mfma_f32_4x4x1_16b_f32) when 16-row
// workgroups would leave CUs idle.  The 16-row engine and the small ones read different orders of the weight stream.
int net_stream_rows(int B);
int net_stream_force_rows(int rows);   // 0 automatic, 4 / 8 / 16 forced; returns the previous setting, -1 for an invalid value
int launch_net_stream_pack(const linna_layer_t* layers, int nl, int in_size, float* packed, int rows, int prog,
                           const NsDense* dn, hipStream_t s, int serve = 0);      // prog 0 + dn: the forward program with the dense segment
bool net_stream_dense_eligible(const linna_layer_t* layers, int nl, int in_size, const NsDense& dn);
int net_stream_describe(const linna_layer_t* layers, int nl, int in_size, int prog, const NsDense* dn, int rows, int serve, char* buf,
                        size_t n);
size_t net_stream_dense_packed_floats(const linna_layer_t* layers, int nl, int in_size, const NsDense& dn);
// the dX chain of a training step as a program of the same kernel (prog 1: ops nl-1..1, prog 2: down to op 0)
bool net_stream_dx_eligible(const linna_layer_t* layers, int nl, int in_size, int with_input);
size_t net_stream_dx_packed_floats(const linna_layer_t* layers, int nl, int in_size, int with_input);
// A small job that rides in the dX-chain launch as one extra workgroup (linna_net_train_step): the batch mean of the loss
// rows (out = scale * sum rows[n], sum_scale_prepare_kernel's order) and AdamW's step counter / bias corrections
struct NsPost { const float* rows; int n; float scale; float* out; int* step; float* hyper; float b1, b2; };
struct NsTrainLoss;
bool net_stream_tb_eligible(const linna_layer_t* layers, int nl, int in_size, const NsDense& dn);
size_t net_stream_tb_packed_floats(const linna_layer_t* layers, int nl, int in_size, const NsDense& dn);
int launch_net_stream_dx(const linna_layer_t* layers, int nl, int in_size, const float* packed, const float* dOUT, int lddo,
                         int B, float* const* dprev, const int* ldp, const float* const* hin, const int* ldh,
                         float* const* dt, const int* lddt, const float* const* t, const int* ldt, int with_input, int rows,
                         hipStream_t s, const NsPost* post = nullptr);
// sampler moves fused around the evaluation.  slice == 0: stretch half step, rows of the batch are the walkers
// S[0..B).  slice == 1: rows are the slice sampler's trial points coords[S[k]] + cc[row] * DIR[k], k = row % nc
// (DIR is passed as the launch's Z / ldz; cc = w[nrep * ns], nc = ns; nothing is written back).
struct NsMove {
    float* coords; int ldc; float* logp; const int* S;
    const float* cc; int ldcc; const int* C; int nc;
    unsigned long long seed; const int* step; int step_off; int stream; float a; int* naccept;
    int slice;
    float* chain = nullptr; float* lps = nullptr;      // stretch move: this iteration's row of the chain block ([nw][ndim], [nw]); null: none
    // slice == 1, sl_Zt != null: the trial weights are derived in the kernel from the one stepping-out round's results (NsArgs::sl_*)
    const float* sl_Z0 = nullptr; const float* sl_L = nullptr; const float* sl_R = nullptr; const float* sl_Zt = nullptr;
    int sl_m = 0, sl_nt = 0; unsigned long long sl_seed = 0; const int* sl_step = nullptr; int sl_stream = 0;
    const int* sl_flags = nullptr;                      // the stepping-out budgets left (flags[3k], flags[3k + 1])
    const SliceBegin* sb = nullptr;                     // slice == 1: this evaluation is the half step's first and sets it up (SliceBegin)
};
// training / validation forward: every op's output stored for the backward (STORE instantiation)
int launch_net_stream_store(const linna_layer_t* layers, int nl, int in_size, const float* packed, const float* X, int ldx,
                            int B, float* const* y, const int* ldy, float* const* t, const int* ldt, const float* cscale,
                            const float* cshift, int rows, hipStream_t s);
// training forward + chi^2-ratio loss in one launch (STORE == 3)
struct NsTrainLoss { const float* YN; int ldyn;      // normalised targets of the whole set, NaN where masked
                     const float* den; float inv_batch; float* loss_rows; float* dP; int lddp; };
int launch_loss_targets(const float* Y, int ldy, int n, const linna_loss_desc_t& d, float* YN, int ldyn, hipStream_t s);
int launch_net_stream_train(const linna_layer_t* layers, int nl, int in_size, const float* packed, const float* X, int ldx,
                            const int* ROWS, int B, const int* lg, const float* xmean, const float* xstd, float* XB, int ldxb,
                            float* const* y, const int* ldy, float* const* t, const int* ldt, const NsTrainLoss& L,
                            const NsDense& dn, int rows, hipStream_t s);
// the same followed by the dX chain down to op 1 in the SAME launch (GRAD + STORE == 3; prog 4)
int launch_net_stream_train_bwd(const linna_layer_t* layers, int nl, int in_size, const float* packed, const float* X, int ldx,
                                const int* ROWS, int B, const int* lg, const float* xmean, const float* xstd, float* XB, int ldxb,
                                float* const* y, const int* ldy, float* const* t, const int* ldt, const NsTrainLoss& L,
                                const NsDense& dn, float* const* dprev, const int* ldp, const float* const* hin, const int* ldh,
                                float* const* dt, const int* lddt, int rows, hipStream_t s, const NsPost* post);
// gradient fused behind the evaluation (plain ReLU MLPs, diagonal covariance): G = d lnP / d z
// hm_*: a leapfrog kick and drift riding in the gradient's finish (HMCSampler.py:35-49): P += ek G; Q += ed P / mass, Q the
// launch's own input rows (hm_p == nullptr: none)
struct NsGrad { const float* gscale; float* G; int ldg; float* hm_p; int hm_ldp; float* hm_q; const float* hm_mass; float hm_ek, hm_ed; };
// AdamW that also writes the two weight streams of a training step (net_stream.hip: adamw_streams_kernel; gemm.hip: the
// update epilogue of the grouped parameter-gradient launch).  An AsPlace says where the elements of a weight matrix sit
// in one fragment-order stream (segment type 0 = WIDE, 1 = SPLIT; see net_stream.hip).
constexpr int AS_MAXR = 26, AS_MAXW = 14, AS_MAXB = 12;
struct AsPlace { float* out; float scale; int trans, koff, ncols, type, ncg, steps, G, first0, first1; };
struct AsMat { int N, ld; AsPlace pl[2]; };            // pl[0]: forward(+loss) stream, pl[1]: dX-chain stream
struct AsBias { float* out; float scale; int N; };
struct AsRange { unsigned off4, n4, blk0; short kind, idx; };   // a tensor of the flat buffer, in units of 4 floats
struct AsArgs { AsRange r[AS_MAXR]; AsMat w[AS_MAXW]; AsBias b[AS_MAXB]; int nr, small; unsigned nblocks; };
// float index of element (column nn, depth kk) of a segment in its stream (4 step tiles per wave and step, 64 lanes x 4 floats each)
__device__ __forceinline__ size_t as_slot(const AsPlace& q, int small, int nn, int kk) {
    const int nl = nn & 63, ks = kk >> 4, kr = kk & 15;
    int w, g;
    if (q.type == 0) { w = (nn & 511) >> 6; g = ((nn >> 9) ? q.first1 : q.first0) + ks; }
    else { const int kp = ks / q.steps; w = kp * q.ncg + (nn >> 6); g = q.first0 + (ks - kp * q.steps); }
    const int t = small ? kr >> 2 : nl >> 4, lane = small ? nl : (nl & 15) + 16 * (kr >> 2);
    return ((((size_t)w * q.G + g) * 4 + t) * 64 + lane) * 4 + (kr & 3);
}
// torch.optim.AdamW's single-tensor update of one element (pointwise.hip adamw_kernel's arithmetic, operation for operation)
__device__ __forceinline__ void adamw_one(float& pi, float gi, float& mi, float& vi, float lr, float wd, float bc1, float sbc2,
                                          float beta1, float beta2, float eps) {
    pi = pi * (1.f - lr * wd);
    mi = mi + (gi - mi) * (1.f - beta1);
    vi = vi * beta2 + (1.f - beta2) * gi * gi;
    const float denom = sqrtf(vi) / sbc2 + eps;
    pi = pi - (lr / bc1) * (mi / denom);
}
int net_stream_adamw_args(const linna_layer_t* layers, int nl, int in_size, int rows, const float* params, size_t nflat,
                          float* s_fwd, const NsDense* dn, float* s_dx, AsArgs* out, int merged = 0);
int launch_adamw_streams(const AsArgs& a, float* p, const float* g, float* m, float* v, const float* hyper, float b1, float b2,
                         float eps, hipStream_t s);
bool net_stream_has_grad(const linna_layer_t* layers, int nl, int in_size);
// forward + dX chain down to the input in one stream (any network both programs cover; diagonal covariance)
bool net_stream_dxi_eligible(const linna_layer_t* layers, int nl, int in_size);
size_t net_stream_dxi_packed_floats(const linna_layer_t* layers, int nl, int in_size);
int launch_net_stream_grad2(const linna_layer_t* layers, int nl, int in_size, const float* packed, const float* Z, int ldz, int B,
                            int nin, const int* is_flat, const float* a1, const float* a2, const int* lg, const float* xmean,
                            const float* xstd, const float* cscale, const float* cshift, const float* w, float T, float* lnP,
                            const NsGrad& gr, float* const* y, const int* ldy, float* const* t, const int* ldt, int rows,
                            hipStream_t s);
int launch_net_stream(const linna_layer_t* layers, int nl, int in_size, const float* packed, const float* Z, int ldz, int B,
                      int nin, const int* is_flat, const float* a1, const float* a2, const int* lg, const float* xmean,
                      const float* xstd, const float* cscale, const float* cshift, const float* w, float T, float* lnP,
                      float* D, int ldd, float* TH, int ldt, const NsMove* mv, const NsGrad* gr, const int* gate, int rows,
                      const NsDense* dn, hipStream_t s, const float* cpost = nullptr, const float* cshift2 = nullptr);

// autocorr.hip: convergence statistics of a walker chain (running lagged products, emcee's estimator, checkmeanstd's moments)
int launch_chain_append_t(const float* block, int ldb, int nsteps, int nw, int ndim, int wstride, float* CT, int nwp,
                          int64_t row0, hipStream_t s);
int launch_acorr_update(const float* CT, int nd, int nwp, int nwc, int64_t a0, int64_t a1, int64_t lo, int64_t hi, int k0, int k1,
                        double* S, double* T, int remove, hipStream_t s);
size_t acorr_scratch_doubles(int nser, int kuse, int nd);
int launch_acorr_tau(const float* CT, int ndim, int nwp, int nwc, int nlive, int64_t lo, int64_t hi, int kuse, const double* S,
                     const double* T, double c, double* scratch, double* out, hipStream_t s);
int launch_chain_meanstd(const float* CT, int ndim, int nwp, int nws, int64_t t0, int64_t tm, int64_t t1, double* out,
                         hipStream_t s);

int gemm_slots(int M, int N);            // number of row-dot partial slots gemm_launch will write
int gemm_launch(const GemmArgs& a, hipStream_t stream);
// several independent k-major x k-major problems in ONE grid (gemm.hip: gemm_group_kernel): C[M][N] = alpha * A^T B over
// K (the batch), db[M] = alpha * column sums of A (null: none); descriptors by value in the kernel arguments
struct GemmGroupProb { const float* A; const float* B; float* C; float* db; int lda, ldb, ldc, K, M, N; float alpha; int first; };
constexpr int GEMM_GROUP_MAX = 48;
struct GemmGroupArgs { GemmGroupProb p[GEMM_GROUP_MAX]; int nprob; };
// The same launch with the optimiser in its epilogue (linna_net_train_step_update): every tile applies AdamW to the block of
// the weight matrix whose gradient it has just formed (parameters / moments at fixed distances from the gradient buffer)
// and puts the updated block into the two weight streams of the next step -- no AdamW launch, no re-layout.
constexpr int GEMM_UPD_MAX = 16;
struct GemmGroupArgsS { GemmGroupProb p[GEMM_UPD_MAX]; int nprob; };
struct GemmUpdate { long long pdiff, mdiff, vdiff;           // parameter / moment pointers = gradient pointer + these (floats)
                    const float* hyper; float beta1, beta2, eps; int small;
                    AsPlace pl[GEMM_UPD_MAX][2]; AsBias bias[GEMM_UPD_MAX]; };
struct GemmUpd1 { long long pdiff, mdiff, vdiff; const float* hyper; float beta1, beta2, eps; int small; AsPlace pl[2]; AsBias bias; };
struct GemmPost { const float* rows; int n; float scale; float* out; };   // batch mean of the loss rows riding as one extra workgroup
int gemm_launch_group_update(const GemmGroupArgsS& g, const GemmUpdate& u, int nblocks, hipStream_t stream, const GemmPost* post = nullptr);
bool gemm_group_ok(const GemmArgs& a);
int gemm_group_blocks(const GemmArgs& a);
int gemm_launch_group(const GemmGroupArgs& g, int nblocks, hipStream_t stream, const GemmPost* post = nullptr);

}  // namespace linna
