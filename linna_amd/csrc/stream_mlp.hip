// Whole-network serving kernel, second generation ("weight stream"), for ReLU MLP emulators whose
// hidden layers are all 512 wide (BASELINE configs 2/5: 33 -> 512 x 4 -> 33).  ONE launch evaluates
// util.Log_prob.__call__ (util.py:990-1021) for 16 walkers per workgroup: prior map + input
// transform (util.py:339-347, 483-497), every nn.Linear + ReLU (nn.py:121-130 shaped), the output
// transform and the Gaussian log-likelihood (util.py:953-955).
//
// What changed against fused_mlp.hip (kept as the fallback for other widths), and why
// (profiles/r01_d_*: stamps + ablations of the first kernel):
//  * weights no longer pass through LDS.  With 16 rows per workgroup no wave shares a weight with
//    another wave, so the LDS ring only cost instructions (s_mov m0 + s_nop + DMA + ds_read per
//    KiB).  The weights are re-laid once per weight update into MFMA FRAGMENT ORDER
//    (pack_weight_stream below): the B operand of step g, column tile t is 1 KiB contiguous,
//    lane-linear, so one fully coalesced global_load_dwordx4 puts it straight into the registers
//    the MFMA reads.  Every wave owns one contiguous stream over ALL layers; a ring of R register
//    sets keeps R-1 steps (>= 5 KiB per wave) in flight across layer boundaries;
//  * one step = 16 k x (16*NT) columns: 1 ds_read_b128 (A, activations), NT global loads, 4*NT
//    MFMAs (v_mfma_f32_16x16x4_f32, exact fp32; the k-permutation trick: lane group kq owns 4
//    consecutive k, element s of the 128-bit fragments feeds MFMA s).  ~6 non-MFMA instructions
//    per 16 MFMAs instead of ~30;
//  * ONE copy of the step loop for all layers (a 6 KiB kernel instead of 60 KiB: the first kernel
//    unrolled per layer and ran every layer from a cold instruction cache);
//  * activations double-buffered in LDS ([2][16][516] fp32, row stride 516 = conflict-free
//    ds_read_b128 and ds_write_b32): one raw s_barrier per layer; biases staged in LDS once, the
//    accumulators START at the bias so the epilogue is max(acc, 0) + store;
//  * the narrow last layer splits K over the waves (same step shape), reduces through LDS and
//    finishes the likelihood with lane shuffles.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace linna {

constexpr int SM_ROWS = 16;             // walker rows per workgroup
constexpr int SM_H = 512;               // hidden width this kernel is specialised for
constexpr int SM_LD = SM_H + 4;         // activation row stride (floats)
constexpr int SM_ABUF = SM_ROWS * SM_LD;
constexpr int SM_MAX_LAYERS = 6;
constexpr int SM_KSLICES = 8;           // K slices of the last layer (64 k each)
#ifndef SM_PRE
#define SM_PRE 2                        // ring slots requested before the network input is computed
#endif

struct StreamArgs {
    const float* Z; int ldz; int B; int nin;
    const int* is_flat; const float* a1; const float* a2; const int* lg;
    const float* xmean; const float* xstd;
    const float* packed;                // [waves][G][NT][64 lanes][4]
    const float* bias[SM_MAX_LAYERS];
    int ksteps[SM_MAX_LAYERS];          // 16-k steps per layer (per wave for the last layer)
    int nl, G, nout;
    const float* cscale; const float* cshift; const float* w; float T;
    float* lnP; float* D; int ldd; float* TH; int ldt;
    unsigned long long* stamps;         // diagnostic builds only (-DSM_STAMPS)
};

__device__ __forceinline__ float sm_prior_theta(float z, int flat, float a1, float a2) {
    float u = 0.5f * (1.f + erff(z / 1.41421356237309515f));
    asm volatile("" : "+v"(u));                    // computed unconditionally: no branch on the loaded flag
    return (flat ? u : z) * a2 + a1;
}

// ------------------------------------------------------------------ weight re-layout
// stream[w][g][t][lane][e]: hidden layer l, step s (g = first[l] + s):
//      W_l[n = 16*(w*NT + t) + li][k = 16 s + 4 kq + e]          (li = lane & 15, kq = lane >> 4)
// last layer, wave w = (kslice, tg) with 4/NT waves per K slice:
//      W_L[n = 16*(tg*NT + t) + li][k = 64 kslice + 16 s + 4 kq + e]
// zero outside the matrix.
struct PackArgs {
    const float* W[SM_MAX_LAYERS]; int K[SM_MAX_LAYERS]; int N[SM_MAX_LAYERS]; int ldw[SM_MAX_LAYERS];
    int first[SM_MAX_LAYERS + 1];
    int nl, G, NT, NW;
    float* out;
};
__global__ void pack_weight_stream(PackArgs p) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // one float4 per thread
    const size_t total = (size_t)p.NW * p.G * p.NT * 64;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    size_t q = idx >> 6;
    const int t = (int)(q % p.NT); q /= p.NT;
    const int g = (int)(q % p.G);
    const int w = (int)(q / p.G);
    int l = 0;
    while (l + 1 < p.nl && g >= p.first[l + 1]) ++l;
    const int s = g - p.first[l];
    const int li = lane & 15, kq = lane >> 4;
    int n, k;
    if (l < p.nl - 1) { n = 16 * (w * p.NT + t) + li; k = 16 * s + 4 * kq; }
    else {
        const int per = 4 / p.NT, ks = w / per, tg = w % per;
        n = 16 * (tg * p.NT + t) + li; k = 64 * ks + 16 * s + 4 * kq;
    }
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (n < p.N[l]) {
        const float* row = p.W[l] + (size_t)n * p.ldw[l];
#pragma unroll
        for (int e = 0; e < 4; ++e) if (k + e < p.K[l]) v[e] = row[k + e];
    }
    reinterpret_cast<f32x4*>(p.out)[idx] = v;
}

// ------------------------------------------------------------------ the kernel
template <int NW, int R>
__global__ __launch_bounds__(64 * NW, 1) void stream_mlp_kernel(StreamArgs a) {
    constexpr int NT = 32 / NW;                    // 16-column tiles per wave (512 / 16 / NW)
    constexpr int RG = 4 * NW;                     // threads per walker row in the prologue / finish
    constexpr unsigned STEP_B = NT * 1024;         // bytes of one step of one wave's stream
    static_assert(NT == 4 || NT == 2, "8 or 16 waves");
    static_assert(R % 2 == 0, "the A double buffer alternates with the ring slot parity");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const act = smem;                       // [2][16][516]
    float* const lbias = smem + 2 * SM_ABUF;       // [nl-1][512]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    const int row0 = blockIdx.x * SM_ROWS;
#ifdef SM_STAMPS
    // stamps are parked in LDS and written out at the very end: a global store inside the step loop
    // would change the compiler's vmcnt bookkeeping and with it the thing being measured
    unsigned long long* const lstamp = reinterpret_cast<unsigned long long*>(smem + 2 * SM_ABUF + (SM_MAX_LAYERS - 1) * SM_H) + wave * 16;
    int nstamp = 0;
#define SM_STAMP() do { const unsigned long long t_ = __builtin_readcyclecounter(); \
        if (lane == 0) lstamp[nstamp] = t_; ++nstamp; } while (0)
#else
#define SM_STAMP() do {} while (0)
#endif
    SM_STAMP();

    // ---- 1. every small load of the kernel, issued up front in straight-line code (no branch may
    // depend on a loaded value here: a branch would put a full memory round trip in front of the
    // weight stream): z and its prior/transform constants for this thread's ZPRE columns (one walker
    // row per RG threads), the hidden biases, and the constants of the finish.
    const int pr = tid / RG, pc0 = tid % RG;
    const int grow = min(row0 + pr, a.B - 1);
    const int kpad0 = 16 * a.ksteps[0];
    const int nl = a.nl, nin = a.nin, nout = a.nout;
    constexpr int ZPRE = 2;
    float zr[ZPRE], za1[ZPRE], za2[ZPRE], zxm[ZPRE], zxs[ZPRE]; int zfl[ZPRE], zlg[ZPRE];
    const int* const lgp = a.lg ? a.lg : a.is_flat;            // always a readable pointer
#pragma unroll
    for (int i = 0; i < ZPRE; ++i) {
        const int c = min(pc0 + i * RG, nin - 1);
        zr[i] = a.Z[(size_t)grow * a.ldz + c];
        zfl[i] = a.is_flat[c]; za1[i] = a.a1[c]; za2[i] = a.a2[c];
        zlg[i] = lgp[c]; zxm[i] = a.xmean[c]; zxs[i] = a.xstd[c];
    }
    const float* const b0 = a.bias[0]; const float* const b1 = a.bias[1]; const float* const b2 = a.bias[2];
    const float* const b3 = a.bias[3]; const float* const b4 = a.bias[4];
    constexpr int BMAX = ((SM_MAX_LAYERS - 1) * SM_H + 64 * NW - 1) / (64 * NW);
    float breg[BMAX];
#pragma unroll
    for (int i = 0; i < BMAX; ++i) {
        const int j = tid + i * 64 * NW, l = j >> 9;
        const bool ok = l < nl - 1;                           // hidden layer (also bounds j)
        const float* bp = l == 0 ? b0 : l == 1 ? b1 : l == 2 ? b2 : l == 3 ? b3 : b4;
        breg[i] = (ok ? bp : b0)[ok ? (j & (SM_H - 1)) : 0];
    }
    constexpr int FIN = 64 / RG > 0 ? 64 / RG : 1;           // output columns per thread in the finish
    const float* const blast = a.bias[SM_MAX_LAYERS - 1];    // the host stores the last layer's bias here
    const float* const csp = a.cscale ? a.cscale : blast;
    const float* const ctp = a.cshift ? a.cshift : blast;
    const float* const wtp = a.w ? a.w : blast;
    float fb[FIN], fcs[FIN], fct[FIN], fw[FIN];
#pragma unroll
    for (int i = 0; i < FIN; ++i) {
        const int c = min(pc0 + i * RG, nout - 1);
        fb[i] = blast[c];
        const float cs = csp[c], ct = ctp[c], ww = wtp[c];
        fcs[i] = a.cscale ? cs : 1.f; fct[i] = a.cshift ? ct : 0.f; fw[i] = a.w ? ww : 0.f;
    }

    // ---- 2. start the weight stream: R steps in flight
    // address = wave-uniform base (SGPR pair) + 32-bit per-lane offset (VGPR) + immediate
    const char* const wbase = reinterpret_cast<const char*>(a.packed) + (size_t)wave * a.G * STEP_B;
    const unsigned wlast = (unsigned)(a.G - 1) * STEP_B;      // offset of the last step
    unsigned woff = 0;                                        // offset of the next step to load (uniform)
    const unsigned voff = 16u * (unsigned)lane;
    f32x4 Bq[R][NT];
    auto wload = [&](int t) {
        return *reinterpret_cast<const f32x4*>(wbase + (size_t)(voff + woff) + t * 1024);
    };
#ifdef SM_NOADVANCE   // timing-only ablation: every load hits the same 4 KiB (L1-resident); results are wrong
    auto wadvance = [&]() { woff = min(woff, wlast); };
#else
    auto wadvance = [&]() { woff = min(woff + STEP_B, wlast); };   // past the end: reload the last step (never used)
#endif
    // (the scheduler must not reorder these: the step loop's counted vmcnt waits are derived from
    // the issue order, and one reversed pair on the entry path degrades every iteration to vmcnt(0))
    // Only the first SM_PRE steps go out before the network input is computed: the compiler waits
    // for z with vmcnt(0), so everything issued by then is on the critical path of the first MFMA
    // (all 256 workgroups start together: R steps each would be a 48 MB burst out of the L2s).
    constexpr int PRE = SM_PRE < R ? SM_PRE : R;
    auto prefetch = [&](auto Uc) {
        constexpr int U = decltype(Uc)::value;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            Bq[U][t] = wload(t);
            __builtin_amdgcn_sched_barrier(0);
        }
        wadvance();
    };
#define SM_PF(U, LO, HI) if constexpr (U >= LO && U < HI) prefetch(std::integral_constant<int, U>{});
#define SM_PF_ALL(LO, HI) SM_PF(0, LO, HI) SM_PF(1, LO, HI) SM_PF(2, LO, HI) SM_PF(3, LO, HI) SM_PF(4, LO, HI) SM_PF(5, LO, HI) \
    SM_PF(6, LO, HI) SM_PF(7, LO, HI) SM_PF(8, LO, HI) SM_PF(9, LO, HI) SM_PF(10, LO, HI) SM_PF(11, LO, HI)
    SM_PF_ALL(0, PRE)

    // ---- 3. prologue: x = X_transform(Transform(z)) into act[0], zero padded to 16*ksteps[0]; biases to LDS.
    // No global store here (theta is written at the very end): a store in flight next to the weight
    // loads would force the compiler to full vmcnt(0) waits inside the step loop.
    float zz = 0.f;
    float theta[ZPRE];
#pragma unroll
    for (int i = 0; i < ZPRE; ++i) {
        const int c = pc0 + i * RG;
        const bool in = c < nin;
        const float z = in ? zr[i] : 0.f;
        zz += z * z;
        float th = sm_prior_theta(z, zfl[i], za1[i], za2[i]);
        float lt = log10f(th);
        asm volatile("" : "+v"(lt));               // both candidates exist: the select below stays a v_cndmask
        theta[i] = th;
        const float t = (a.lg && zlg[i]) ? lt : th;
        const float x = in ? (t - zxm[i]) / zxs[i] : 0.f;
        if (c < kpad0) act[pr * SM_LD + c] = x;
    }
    __builtin_amdgcn_sched_barrier(0);
    SM_PF_ALL(PRE, R)
#undef SM_PF_ALL
#undef SM_PF
#pragma unroll
    for (int o = RG / 2; o >= 1; o >>= 1) zz += __shfl_xor(zz, o, 64);
#pragma unroll
    for (int i = 0; i < BMAX; ++i) {
        const int j = tid + i * 64 * NW;
        if ((j >> 9) < nl - 1) lbias[j] = breg[i];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                  // raw: __syncthreads() would drain the weight stream
    asm volatile("" ::: "memory");
    SM_STAMP();

    // ---- 4. the step loop: floor(G/R) groups of R steps with the refill loads, then G%R tail steps.
    // No exits from inside a group: the compiler's vmcnt bookkeeping then sees every ring slot's
    // loads followed by exactly (R-1)*NT younger ones and waits with a counted vmcnt.
    const int colbase = wave * 16 * NT;            // first column of this wave in a hidden layer
    const int lastks = wave / (4 / NT), lasttg = wave % (4 / NT);
    const uint32_t act_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)act;
    f32x4 acc[NT];
    f32x4 Aq[2];
    int layer = 0, p = 0, kleft = a.ksteps[0];
    uint32_t ap;                                   // per-lane LDS byte address of the next A fragment
    // A fragments are read with inline asm one step ahead of their MFMAs (the compiler would merge
    // the two buffers and read right before the use); lgkmcnt(0) at the top of the next step
    auto a_read = [&](f32x4& dst) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(ap) : "memory");
        ap += 64;
    };
    auto begin_layer = [&]() {                     // accumulators and A pointer of `layer`
        const bool last = layer == nl - 1;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float b = last ? 0.f : lbias[layer * SM_H + colbase + 16 * t + li];
            acc[t] = f32x4{b, b, b, b};
        }
        ap = act_lds + 4u * (uint32_t)(p * SM_ABUF + li * SM_LD + 4 * kq + (last ? 64 * lastks : 0));
    };
    begin_layer();
    a_read(Aq[0]);

    auto step = [&](auto Uc, auto Refill) {        // one step on ring slot U
        constexpr int U = decltype(Uc)::value;
        constexpr bool refill = decltype(Refill)::value;
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(Aq[U & 1]) :: "memory");
        a_read(Aq[(U + 1) & 1]);                   // next step's A (speculative at a layer end)
        const f32x4 av = Aq[U & 1];
#pragma unroll
        for (int h = 0; h < NT; h += 2) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], Bq[U][h][s], acc[h], 0, 0, 0);
                acc[h + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], Bq[U][h + 1][s], acc[h + 1], 0, 0, 0);
            }
            if constexpr (refill) {                // the two fragments just consumed <- step +R
                Bq[U][h] = wload(h);
                Bq[U][h + 1] = wload(h + 1);
            }
        }
        if constexpr (refill) wadvance();
        if (--kleft == 0 && layer < nl - 1) {
            // ---- hidden layer complete: ReLU, publish into the other buffer, one barrier
            float* const nxt = act + (p ^ 1) * SM_ABUF;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e)   // C/D layout of v_mfma_f32_16x16x4_f32: col = lane&15, row = 4*(lane>>4) + e
                    nxt[(4 * kq + e) * SM_LD + colbase + 16 * t + li] = fmaxf(acc[t][e], 0.f);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            SM_STAMP();
            p ^= 1; ++layer;
            kleft = a.ksteps[layer];
            begin_layer();
            a_read(Aq[(U + 1) & 1]);               // replaces the speculative fragment
        }
    };
    using T_ = std::true_type; using F_ = std::false_type;
#define SM_STEP(U, RF) if constexpr (U < R) step(std::integral_constant<int, U>{}, RF{});
    const int ngroups = a.G / R, rem = a.G - ngroups * R;
#pragma unroll 1
    for (int it = 0; it < ngroups; ++it) {
        SM_STEP(0, T_) SM_STEP(1, T_) SM_STEP(2, T_) SM_STEP(3, T_) SM_STEP(4, T_) SM_STEP(5, T_)
        SM_STEP(6, T_) SM_STEP(7, T_) SM_STEP(8, T_) SM_STEP(9, T_) SM_STEP(10, T_) SM_STEP(11, T_)
    }
#define SM_TAIL(U) if constexpr (U < R - 1) { if (rem > U) step(std::integral_constant<int, U>{}, F_{}); }
    SM_TAIL(0) SM_TAIL(1) SM_TAIL(2) SM_TAIL(3) SM_TAIL(4) SM_TAIL(5) SM_TAIL(6) SM_TAIL(7) SM_TAIL(8) SM_TAIL(9) SM_TAIL(10)
#undef SM_STEP
#undef SM_TAIL
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the last speculative A read
    SM_STAMP();

    // ---- 5. last layer: reduce the K slices through LDS, output transform, log-likelihood
    {
        float* const part = act + (p ^ 1) * SM_ABUF;           // [8 slices][16 rows][64 cols], column ^= 16*(row>>2)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int col = 16 * (lasttg * NT + t) + li;
                part[lastks * 1024 + (4 * kq + e) * 64 + (col ^ (16 * kq))] = acc[t][e];
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const bool rok = row0 + pr < a.B;
        float chi = 0.f;
#pragma unroll
        for (int i = 0; i < FIN; ++i) {
            const int c = pc0 + i * RG;
            if (c < nout) {
                float v = 0.f;
#pragma unroll
                for (int ks = 0; ks < SM_KSLICES; ++ks) v += part[ks * 1024 + pr * 64 + (c ^ (16 * (pr >> 2)))];
                v += fb[i];
                const float d = v * fcs[i] + fct[i];
                if (a.D && rok) a.D[(size_t)(row0 + pr) * a.ldd + c] = d;
                chi += (d * fw[i]) * d;
            }
        }
#pragma unroll
        for (int o = RG / 2; o >= 1; o >>= 1) chi += __shfl_xor(chi, o, 64);
        if (a.lnP && a.w && pc0 == 0 && rok) {
            const float v = (-0.5f * chi) / a.T + (-0.5f * zz);
            a.lnP[row0 + pr] = isnan(v) ? -INFINITY : v;
        }
        if (a.TH && rok) {
#pragma unroll
            for (int i = 0; i < ZPRE; ++i)
                if (pc0 + i * RG < nin) a.TH[(size_t)(row0 + pr) * a.ldt + pc0 + i * RG] = theta[i];
        }
    }
    SM_STAMP();
#ifdef SM_STAMPS
    if (lane < 16) a.stamps[((size_t)blockIdx.x * NW + wave) * 16 + lane] = lane < nstamp ? lstamp[lane] : 0ull;
#endif
#undef SM_STAMP
}

// ---------------------------------------------------------------------------- host side
#ifndef SM_NW
#define SM_NW 8
#endif
#ifndef SM_R
#define SM_R 6
#endif

bool stream_mlp_eligible(const linna_layer_t* layers, int nl, int in_size) {
    if (nl < 2 || nl > SM_MAX_LAYERS) return false;
    int k = in_size;
    if (k < 1 || k > 2 * 4 * SM_NW) return false;            // ZPRE * RG input columns
    for (int i = 0; i < nl; ++i) {
        const linna_layer_t& l = layers[i];
        if (l.op != LINNA_OP_LINEAR || l.K != k) return false;
        if (i < nl - 1) { if (!l.relu || l.N != SM_H) return false; }
        else if (l.relu || l.N > 64 || l.K != SM_H) return false;
        k = l.N;
    }
    return true;
}

static int stream_steps(const linna_layer_t* layers, int nl, int* ksteps) {   // returns G
    int G = 0;
    for (int i = 0; i < nl; ++i) {
        ksteps[i] = (i < nl - 1) ? (layers[i].K + 15) / 16 : (SM_H / SM_KSLICES) / 16;
        G += ksteps[i];
    }
    return G;
}

size_t stream_mlp_packed_floats(const linna_layer_t* layers, int nl) {
    int ks[SM_MAX_LAYERS];
    return (size_t)SM_NW * stream_steps(layers, nl, ks) * (32 / SM_NW) * 256;
}

int launch_pack_weight_stream(const linna_layer_t* layers, int nl, float* packed, hipStream_t s) {
    PackArgs p;
    p.nl = nl; p.NW = SM_NW; p.NT = 32 / SM_NW; p.out = packed;
    int ks[SM_MAX_LAYERS];
    p.G = stream_steps(layers, nl, ks);
    p.first[0] = 0;
    for (int i = 0; i < nl; ++i) {
        p.W[i] = layers[i].W; p.K[i] = layers[i].K; p.N[i] = layers[i].N; p.ldw[i] = (layers[i].K + 3) & ~3;
        p.first[i + 1] = p.first[i] + ks[i];
    }
    for (int i = nl; i < SM_MAX_LAYERS; ++i) { p.W[i] = nullptr; p.K[i] = p.N[i] = p.ldw[i] = 0; p.first[i + 1] = p.first[nl]; }
    const size_t total = (size_t)p.NW * p.G * p.NT * 64;
    hipLaunchKernelGGL(pack_weight_stream, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p);
    return check_hip(hipGetLastError(), "pack_weight_stream launch");
}

int launch_stream_mlp(const linna_layer_t* layers, int nl, const float* packed, const float* Z, int ldz, int B, int nin,
                      const int* is_flat, const float* a1, const float* a2, const int* lg, const float* xmean,
                      const float* xstd, const float* cscale, const float* cshift, const float* w, float T, float* lnP,
                      float* D, int ldd, float* TH, int ldt, hipStream_t s) {
    StreamArgs a;
    a.Z = Z; a.ldz = ldz; a.B = B; a.nin = nin;
    a.is_flat = is_flat; a.a1 = a1; a.a2 = a2; a.lg = lg; a.xmean = xmean; a.xstd = xstd;
    a.packed = packed;
    a.G = stream_steps(layers, nl, a.ksteps);
    for (int i = 0; i < SM_MAX_LAYERS; ++i) a.bias[i] = layers[i < nl ? i : nl - 1].b;   // [5] is always the last layer's
    for (int i = nl; i < SM_MAX_LAYERS; ++i) a.ksteps[i] = 0;
    a.nl = nl; a.nout = layers[nl - 1].N;
    a.cscale = cscale; a.cshift = cshift; a.w = w; a.T = T;
    a.lnP = lnP; a.D = D; a.ldd = ldd; a.TH = TH; a.ldt = ldt;
    a.stamps = nullptr;
#ifdef SM_STAMPS
    a.stamps = getenv("LINNA_FUSED_STAMPS") ? reinterpret_cast<unsigned long long*>(strtoull(getenv("LINNA_FUSED_STAMPS"), nullptr, 16)) : nullptr;
    if (!a.stamps) { set_error("stream_mlp: SM_STAMPS build needs LINNA_FUSED_STAMPS"); return LINNA_ERR_INVALID; }
#endif
#ifdef SM_STAMPS
    constexpr size_t lds = (size_t)(2 * SM_ABUF + (SM_MAX_LAYERS - 1) * SM_H) * sizeof(float) + SM_NW * 16 * 8;
#else
    constexpr size_t lds = (size_t)(2 * SM_ABUF + (SM_MAX_LAYERS - 1) * SM_H) * sizeof(float);
#endif
    static bool attr_set = false;
    if (!attr_set) {
        const int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_mlp_kernel<SM_NW, SM_R>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute");
        if (rc != LINNA_OK) return rc;
        attr_set = true;
    }
    hipLaunchKernelGGL((stream_mlp_kernel<SM_NW, SM_R>), dim3((B + SM_ROWS - 1) / SM_ROWS), dim3(64 * SM_NW), lds, s, a);
    return check_hip(hipGetLastError(), "stream_mlp launch");
}

}  // namespace linna
