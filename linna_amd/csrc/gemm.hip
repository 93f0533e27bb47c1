// fp32-input MFMA GEMM for gfx950 with fused epilogues.  Replaces every torch nn.Linear /
// F.relu / `@` on the reference hot path (linna/nn.py:53-54,121-130; util.py:1077-1085; and
// the autograd of those, predictor_gpu.py:285).
//
// Arithmetic is exact fp32 (v_mfma_f32_32x32x2_f32 == k-ordered fmaf chain): the parity
// path against the reference's fp32 CPU GEMMs.
//
// Structure (one workgroup = (WM*TM*32) x (WN*TN*32) outputs, BK = 32):
//  * operand tiles travel global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per
//    wave-instruction) into an NS-deep ring; the only waits in the K loop are a counted
//    s_waitcnt vmcnt(N) and ONE raw s_barrier per K tile, so NS-1 tiles are always in flight;
//  * the LDS image is unpadded (DMA writes lane-linear); K-contiguous tiles are XOR-swizzled
//    on the SOURCE address and on the fragment read (chunk ^= (row>>1)&7) so that the
//    ds_read_b128 fragment reads are bank-conflict free;
//  * within an 8-deep k group, lane half h = lane>>5 owns k = 8g+4h+{0..3}: one ds_read_b128
//    per operand feeds four MFMA steps (a permutation of k, identical for A and B);
//  * a partial last K tile, and whole operands that are not 16-byte aligned, are staged
//    through registers with clamped, masked element loads into the same LDS image;
//  * out-of-range rows/columns of an edge tile read clamped (valid) addresses; their
//    products are never stored.
#include "common.h"
#include <algorithm>
#include <stdlib.h>
#include <map>
#include <mutex>

namespace linna {

constexpr int BK = 32;
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));   // vmcnt(N): lgkmcnt/expcnt fields left at max
    asm volatile("" ::: "memory");
}
// s_waitcnt vmcnt(y * LPT) for the runtime y in [0, Y]: a tile with y younger tiles still in flight
template <int Y, int LPT> __device__ __forceinline__ void wait_younger(int y) {
    if constexpr (Y == 0) wait_vmcnt<0>();
    else { if (y >= Y) wait_vmcnt<Y * LPT>(); else wait_younger<Y - 1, LPT>(y); }
}

// One operand tile: ROWS (along M or N) x BK floats, ROWS*128 bytes, staged by NW waves.
template <int ROWS, int LAY, int NW>
struct Tile {
    static constexpr int SIZE = ROWS * BK;                 // floats
    static constexpr int NI = (ROWS * BK * 4 / 1024) / NW;  // 1 KiB DMA instructions per wave per tile
    static_assert(NI >= 1 && (ROWS * BK * 4 / 1024) % NW == 0, "tile/wave mismatch");

    // LDS float offset of logical element (row r, k) -- K-contiguous: 16-byte chunk c=k/4 lives
    // at chunk slot c ^ ((r>>1)&7); k-major: [k][ROWS] plain.
    __device__ static __forceinline__ int slot(int r, int c) { return r * BK + 4 * (c ^ ((r >> 1) & 7)); }

    // Issue this wave's share of the tile as LDS-DMA.  `row0`/`nrows`: first row of the tile and
    // the operand's extent along the tiled dimension; `ld` multiple of 4 and base 16-B aligned.
    // `klast` (k-major operands): k rows past it read row klast again -- the partial last K tile rides the ring too, and
    // the consumer zeroes those rows of ONE operand in LDS before it multiplies.
    __device__ static __forceinline__ void dma(const float* __restrict__ g, int ld, int row0, int nrows, int k0,
                                               float* stage, int wave, int lane, int klast = 0x7fffffff) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int ins = wave * NI + j;
            const float* src;
            if (LAY == LAY_K) {
                const int r = ins * 8 + (lane >> 3);
                const int c = (lane & 7) ^ ((r >> 1) & 7);
                src = g + (size_t)min(row0 + r, nrows - 1) * ld + k0 + 4 * c;
            } else {
                constexpr int CH = ROWS / 4, KR = 64 / CH;          // chunks per k-row, k-rows per instruction
                const int k = ins * KR + lane / CH;
                const int col = min(row0 + 4 * (lane % CH), ld - 4);  // stays inside the (padded) row
                src = g + (size_t)min(k0 + k, klast) * ld + col;
            }
            __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(stage + ins * 256), 16, 0, 0);
        }
    }

    // Register-staged fallback for a partial K tile or an unaligned operand: clamped element
    // loads, zero beyond K / beyond nrows, written into the same LDS image.
    __device__ static __forceinline__ void stage_slow(const float* __restrict__ g, int ld, int row0, int nrows, int k0,
                                                      int K, float* stage, int tid) {
        constexpr int NT = NW * 64;
        for (int f = tid; f < ROWS * BK / 4; f += NT) {
            f32x4 v;
            int dst;
            if (LAY == LAY_K) {
                const int r = f >> 3, c = f & 7;
                const int gr = row0 + r;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int gk = k0 + 4 * c + e;
                    const float x = g[(size_t)min(gr, nrows - 1) * ld + min(gk, K - 1)];
                    v[e] = (gr < nrows && gk < K) ? x : 0.f;
                }
                dst = slot(r, c);
            } else {
                constexpr int CH = ROWS / 4;
                const int k = f / CH, cc = f % CH;
                const int gk = k0 + k;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int gr = row0 + 4 * cc + e;
                    const float x = g[(size_t)min(gk, K - 1) * ld + min(gr, nrows - 1)];
                    v[e] = (gr < nrows && gk < K) ? x : 0.f;
                }
                dst = k * ROWS + 4 * cc;
            }
            *reinterpret_cast<f32x4*>(stage + dst) = v;
        }
    }

    // Fragment of the 32-row slab starting at `slab0`, k group g: 4 values = 4 MFMA steps.
    __device__ static __forceinline__ f32x4 frag(const float* stage, int slab0, int g, int lane) {
        const int i = lane & 31, h = lane >> 5;
        if (LAY == LAY_K) {
            return *reinterpret_cast<const f32x4*>(stage + slot(slab0 + i, 2 * g + h));
        } else {
            f32x4 v;
#pragma unroll
            for (int s = 0; s < 4; ++s) v[s] = stage[(8 * g + 4 * h + s) * ROWS + slab0 + i];
            return v;
        }
    }
};

// KS > 1: the workgroup holds KS groups of WM*WN waves; group kg streams K tiles kg, kg+KS, ... through
// its own slice of every ring stage and the groups' accumulators are summed through LDS in fixed order
// before the epilogue.  For grids that leave CUs idle: one output tile's K loop is a serial chain of
// 64-cycle MFMAs on one wave per SIMD, and splitting K inside the workgroup is the parallelism left.
// kz > 1: K is ALSO split over kz workgroups per output tile (grid = kz x tiles rounded up to a multiple of 8,
// so that the workgroups of one tile share an XCD).  Every workgroup parks its partial tile in `ks_scratch`
// and bumps the tile's counter; the LAST one to arrive sums the kz partials in fixed order (its own included,
// so the result does not depend on who was last), runs the epilogue and resets the counter.  No workgroup
// ever waits for another.  For the few-tile, long-K GEMMs of a batch-500 training step, which otherwise run
// as one serial MFMA chain per tile on 8-64 of the 256 CUs.
struct KSplit { int kz; float* scratch; int* counters; };

#ifdef GEMM_STAMPS
// diagnostic build (python linna_amd/_build.py --stamps --source=gemm.hip -DGEMM_STAMPS; tools/gemm_stamps.py): cycle
// counter at four points of every workgroup of the LAST gemm_body launch -- entry, first K tile requested, K loop done,
// epilogue stores drained -- plus the hardware id (which CU the workgroup ran on)
__device__ unsigned long long g_gemm_stamps[8 * 1024];
__device__ unsigned long long g_gemm_rt[2 * 1024];       // s_memrealtime (100 MHz, one origin for the whole chip) at entry and at the end
#define GEMM_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 1024) { g_gemm_stamps[blockIdx.x * 8 + (i)] = __builtin_readcyclecounter(); \
        if ((i) == 0) g_gemm_rt[blockIdx.x * 2] = __builtin_amdgcn_s_memrealtime(); \
        if ((i) == 3) g_gemm_rt[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define GEMM_STAMP(i) do {} while (0)
#endif

// CH2 (the grouped parameter-gradient launches): a 64 x 64 workgroup gives every wave ONE accumulator tile, and sixteen
// MFMAs per K tile that each wait for the one before run at ~100-115 cycles apiece instead of 64 (tools/gemm_stamps.py: a
// workgroup alone on its CU spends 1.65 k cycles per K tile in fragments + MFMAs for 1.0 k of matrix pipe).  With CH2 the k
// steps alternate between two accumulators -- dependent MFMAs sit two issues apart -- which are added once after the K
// loop (one fixed order: even k steps + odd k steps; every update form of a training step shares this kernel).
template <int WM, int WN, int TM, int TN, int ALAY, int BLAY, int NS, int KS, bool UPD = false, bool CH2 = false>
__device__ __forceinline__ void gemm_body(const GemmArgs& a, const int block_all, const KSplit ks, float* const db = nullptr,
                                          const GemmUpd1* const up = nullptr) {
    static_assert(!CH2 || (TM == 1 && TN == 1), "two accumulator chains: one tile per wave");
    constexpr int NW = WM * WN, NT = NW * 64;                 // waves / threads of one K group
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    using TA = Tile<BM, ALAY, NW>;
    using TB = Tile<BN, BLAY, NW>;
    constexpr int LPT = TA::NI + TB::NI;                      // DMA instructions per wave per K tile
    extern __shared__ __attribute__((aligned(16))) float smem[];  // NS x KS x (A tile | B tile)
    constexpr int STAGE = TA::SIZE + TB::SIZE;

    GEMM_STAMP(0);
#ifdef GEMM_STAMPS
    if (threadIdx.x == 0 && blockIdx.x < 1024) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_gemm_stamps[blockIdx.x * 8 + 4] = ((unsigned long long)xcc << 32) | hw;
    }
#endif
    const int lane = threadIdx.x & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kg = wave_all / NW, wave = wave_all % NW;       // K group, wave within the group
    const int tid = threadIdx.x % NT;                         // thread within the group
    const int wm = wave / WN, wn = wave % WN;

    // XCD-aware tile order: blocks b and b+8 share an XCD (private L2), so give every XCD a
    // contiguous run of tiles; consecutive tiles walk N first and therefore share the A panel.
    const int ntm = (a.M + BM - 1) / BM, ntn = (a.N + BN - 1) / BN;
    const int nwg = ntm * ntn;
    const int nwgp = (nwg + 7) & ~7;                          // blocks per K split (padded: same XCD for every split of a tile)
    const int kzi = ks.kz > 1 ? block_all / nwgp : 0;         // K split of this workgroup
    const int block = ks.kz > 1 ? block_all % nwgp : block_all;
    if (block >= nwg) return;
    const int gid = kzi * KS + kg, ngrp = ks.kz * KS;         // K group among all groups of this tile
    int tile;
    {
        const int b = block, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    const int tm_ = tile / ntn, tn_ = tile % ntn;
    const int m0 = tm_ * BM, n0 = tn_ * BN;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    f32x16 acc_odd;                                           // CH2: the odd k steps' chain
#pragma unroll
    for (int e = 0; e < 16; ++e) acc_odd[e] = 0.f;

    // Bias gradient fused into the parameter-gradient GEMM (db != null; k-major A = dY[batch][n], one K group): the column
    // sums of the A tile over the batch rows, taken from LDS by the tiles of the first tile column -- wave w adds the
    // k rows w, w + NW, ... of every K tile for column `lane` -- summed over the waves after the K loop.
    float csum = 0.f;
    const bool do_colsum = db != nullptr && ALAY == LAY_MN && KS == 1 && TM == 1 && BM == 64 && tn_ == 0 && ks.kz <= 1;
    auto compute = [&](const float* st) {
        const float* cA = st;
        const float* cB = st + TA::SIZE;
        if (do_colsum) {
#pragma unroll
            for (int kk = 0; kk < BK / NW; ++kk) csum += cA[(wave + kk * NW) * BM + lane];
        }
        f32x4 af[2][TM], bf[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[0][i] = TA::frag(cA, (wm * TM + i) * 32, 0, lane);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[0][j] = TB::frag(cB, (wn * TN + j) * 32, 0, lane);
#pragma unroll
        for (int g = 0; g < BK / 8; ++g) {
            if (g + 1 < BK / 8) {        // next k group's fragments under this group's MFMAs
#pragma unroll
                for (int i = 0; i < TM; ++i) af[(g + 1) & 1][i] = TA::frag(cA, (wm * TM + i) * 32, g + 1, lane);
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[(g + 1) & 1][j] = TB::frag(cB, (wn * TN + j) * 32, g + 1, lane);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if (CH2 && (s & 1)) acc_odd = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g & 1][i][s], bf[g & 1][j][s], acc_odd, 0, 0, 0);
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g & 1][i][s], bf[g & 1][j][s], acc[i][j], 0, 0, 0);
                    }
        }
    };

    // ---- per-column epilogue constants, loaded BEFORE any LDS-DMA is in flight and pinned, so that
    // their latency hides under the K loop and the compiler's wait for them does not drain the ring
    const bool two = a.npairs > 1;
    float e_b[TN], e_cs[TN], e_ct[TN], e_cp[TN], e_ct2[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + (wn * TN + j) * 32 + (lane & 31);
        const bool cok = col < a.N;
        const float* bp = two ? a.bias1 : a.bias0;
        e_b[j] = (cok && bp) ? bp[col] : 0.f;
        e_cs[j] = (cok && a.cscale) ? a.cscale[col] : 1.f;
        e_ct[j] = (cok && a.cshift) ? a.cshift[col] : 0.f;
        e_cp[j] = (cok && a.cpost) ? a.cpost[col] : 1.f;
        e_ct2[j] = (cok && a.cshift2) ? a.cshift2[col] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        asm volatile("" : "+v"(e_b[j]), "+v"(e_cs[j]), "+v"(e_ct[j]), "+v"(e_cp[j]), "+v"(e_ct2[j]));
    }

    // UPD: the parameters and moments of this tile's elements are requested HERE, before the first K tile -- they do not
    // depend on the product, and behind the K loop their latency was a third of the epilogue (tools/gemm_stamps.py).  They
    // are older than every LDS-DMA, so the ring's counted waits are untouched.
    float pv[16], mv[16], vv[16];
    if constexpr (UPD) {
        const int col = n0 + wn * 32 + (lane & 31);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = m0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
            const float* const gp = a.C + (size_t)min(row, a.M - 1) * a.ldc + min(col, a.N - 1);
            pv[e] = gp[up->pdiff]; mv[e] = gp[up->mdiff]; vv[e] = gp[up->vdiff];
        }
    }

    for (int pi = 0; pi < a.npairs; ++pi) {
        const GemmPair p = a.p[pi];
        const bool dma_ok = ((p.lda & 3) == 0) && ((p.ldb & 3) == 0) &&
                            (((reinterpret_cast<uintptr_t>(p.A) | reinterpret_cast<uintptr_t>(p.B)) & 15) == 0);
        const int nfull = dma_ok ? p.K / BK : 0;                // full K tiles streamed by LDS-DMA
        const int nall = (p.K + BK - 1) / BK;
        // ... and the partial last one with them where both operands are k-major (the parameter-gradient products: K = the
        // batch, 500 = 15 x 32 + 20): its rows past K are re-reads of row K - 1, zeroed in A's LDS image before the MFMAs
        // (through registers it cost 7-11 k of a workgroup's 65 k cycles)
        const bool tail_dma = dma_ok && ALAY == LAY_MN && BLAY == LAY_MN && KS == 1 && ks.kz <= 1 && nall > nfull;
        const int nfast = nfull + (tail_dma ? 1 : 0);
        const int nmine = nfast > gid ? (nfast - gid + ngrp - 1) / ngrp : 0;   // ... of which this group takes gid, gid+ngrp, ...
        const int niter = (nfast + ngrp - 1) / ngrp;            // ring turns (same for every group of the workgroup: barriers)
        auto stage_of = [&](int i) { return smem + ((i % NS) * KS + kg) * STAGE; };
        auto issue = [&](int i) {
            float* st = stage_of(i);
            const int t = gid + i * ngrp;
            TA::dma(p.A, p.lda, m0, a.M, t * BK, st, wave, lane, p.K - 1);
            TB::dma(p.B, p.ldb, n0, a.N, t * BK, st + TA::SIZE, wave, lane, p.K - 1);
        };
        if (nfast > 0) {
#pragma unroll
            for (int i = 0; i < NS - 1; ++i)
                if (i < nmine) issue(i);
            GEMM_STAMP(1);
#ifdef GEMM_STAMPS
            unsigned long long tw = 0, tb = 0, tc = 0;
#endif
            for (int it = 0; it < niter; ++it) {
#ifdef GEMM_STAMPS
                const unsigned long long s0 = __builtin_readcyclecounter();
#endif
                // tile `it` of this group must have landed; up to NS-2 younger ones stay in flight
                if (it < nmine) wait_younger<NS - 2, LPT>(min(NS - 2, nmine - 1 - it));
#ifdef GEMM_STAMPS
                const unsigned long long s1 = __builtin_readcyclecounter();
#endif
                __builtin_amdgcn_s_barrier();      // all waves' DMA for this turn landed; stage (it-1)%NS is free
                asm volatile("" ::: "memory");
#ifdef GEMM_STAMPS
                const unsigned long long s2 = __builtin_readcyclecounter();
#endif
                if (it + NS - 1 < nmine) issue(it + NS - 1);
                if (tail_dma && it == nfull) {     // (block-uniform) the partial tile: A's k rows past K to zero
                    float* const za = stage_of(it) + (p.K - nfull * BK) * BM;
                    const int nz = (nall * BK - p.K) * BM / 4;
                    for (int f = tid; f < nz; f += NT) reinterpret_cast<f32x4*>(za)[f] = f32x4{0.f, 0.f, 0.f, 0.f};
                    __syncthreads();
                }
                if (it < nmine) compute(stage_of(it));
#ifdef GEMM_STAMPS
                const unsigned long long s3 = __builtin_readcyclecounter();
                tw += s1 - s0; tb += s2 - s1; tc += s3 - s2;
#endif
            }
            __builtin_amdgcn_s_barrier();          // last tile fully read before anything restages
#ifdef GEMM_STAMPS
            if (threadIdx.x == 0 && blockIdx.x < 1024) {
                g_gemm_stamps[blockIdx.x * 8 + 5] = tw; g_gemm_stamps[blockIdx.x * 8 + 6] = tb; g_gemm_stamps[blockIdx.x * 8 + 7] = tc;
            }
#endif
        }
        if (kzi == 0) {
            for (int kt = nfast; kt < nall; ++kt) {    // partial tile / unaligned operand: register path, group 0 of split 0
                if (kg == 0) {
                    TA::stage_slow(p.A, p.lda, m0, a.M, kt * BK, p.K, smem, tid);
                    TB::stage_slow(p.B, p.ldb, n0, a.N, kt * BK, p.K, smem + TA::SIZE, tid);
                }
                __syncthreads();
                if (kg == 0) compute(smem);
                __syncthreads();
            }
        }
        if (pi == 0 && a.npairs > 1) {             // acc <- alpha0 * (acc + bias0) before pair 1 accumulates
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + (wn * TN + j) * 32 + (lane & 31);
                const float b0 = (gid == 0 && a.bias0 && col < a.N) ? a.bias0[col] : 0.f;  // the bias once, not per K group
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = a.alpha0 * (acc[i][j][e] + b0);
            }
        }
    }

    if constexpr (CH2) {
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[0][0][e] = acc[0][0][e] + acc_odd[e];
    }
    GEMM_STAMP(2);
    if (db != nullptr && ALAY == LAY_MN && KS == 1 && TM == 1 && BM == 64 && ks.kz <= 1) {   // (block-uniform)
        if (do_colsum) {
            __syncthreads();
            smem[wave * 64 + lane] = csum;
            __syncthreads();
            if (wave == 0 && m0 + lane < a.M) {
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) t += smem[w * 64 + lane];
                const float gb = a.alpha0 * t;
                db[m0 + lane] = gb;
                if constexpr (UPD) {                // the bias element's AdamW step, and its slot in the forward stream's bias block
                    float* const gp = db + m0 + lane;
                    float pi = gp[up->pdiff], mi = gp[up->mdiff], vi = gp[up->vdiff];
                    adamw_one(pi, gb, mi, vi, up->hyper[0], up->hyper[1], up->hyper[2], up->hyper[3], up->beta1, up->beta2, up->eps);
                    gp[up->pdiff] = pi; gp[up->mdiff] = mi; gp[up->vdiff] = vi;
                    if (up->bias.out) up->bias.out[m0 + lane] = up->bias.scale * pi;
                }
            }
        }
    }

    // ---------------------------------------------------------------- K groups -> group 0 (fixed order)
    if constexpr (KS > 1) {
        __syncthreads();
        constexpr int E = TM * TN * 16;
        float* const red = smem;                   // [KS-1][E][NT]: consecutive threads, consecutive floats
        if (kg > 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) red[((kg - 1) * E + (i * TN + j) * 16 + e) * NT + tid] = acc[i][j][e];
        }
        __syncthreads();
        if (kg > 0) return;
#pragma unroll
        for (int g = 1; g < KS; ++g)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] += red[((g - 1) * E + (i * TN + j) * 16 + e) * NT + tid];
    }

    // ---------------------------------------------------------------- K splits -> the last workgroup to arrive
    if (ks.kz > 1) {
        constexpr int E = TM * TN * 16;
        __shared__ int s_last;
        float* const mine = ks.scratch + ((size_t)tile * ks.kz + kzi) * (E * NT);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) mine[((i * TN + j) * 16 + e) * NT + tid] = acc[i][j][e];
        __threadfence();                               // partial visible device-wide before the counter moves
        __syncthreads();
        if (threadIdx.x == 0) s_last = atomicAdd(ks.counters + tile, 1) == ks.kz - 1;
        __syncthreads();
        if (!s_last) return;
        __threadfence();                               // acquire: the other splits' partials
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        for (int z = 0; z < ks.kz; ++z) {
            const float* src = ks.scratch + ((size_t)tile * ks.kz + z) * (E * NT);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        acc[i][j][e] += __builtin_nontemporal_load(src + ((i * TN + j) * 16 + e) * NT + tid);
        }
        if (threadIdx.x == 0) ks.counters[tile] = 0;   // every split has arrived: ready for the next launch
    }

    // ---------------------------------------------------------------- epilogue
    // C/D layout of v_mfma_f32_32x32x2_f32: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5).
    const int h = lane >> 5;
    if constexpr (UPD) {
        // Parameter-gradient tile with the optimiser behind it: C = alpha acc is the gradient of W[row][col]; AdamW on that
        // element (parameters and moments sit at fixed distances from the gradient), then the updated 4-row groups into the
        // weight streams -- one 16-byte vector per group where the stream holds the matrix transposed, single floats where it
        // holds rows (neighbouring lanes complete the vector).
        static_assert(TM == 1 && TN == 1, "update epilogue: 64 x 64 workgroups");
        const int col = n0 + wn * 32 + (lane & 31);
        const bool cok = col < a.N;
        const float lr = up->hyper[0], wd = up->hyper[1], bc1 = up->hyper[2], sbc2 = up->hyper[3];
        // the loads were requested before the K loop; here the arithmetic, then every store (parameters and moments are
        // reached through one pointer plus offsets, so a store of element e and a load of element e + 1 may alias as far as
        // the compiler knows: an element-by-element loop would be sixteen dependent memory round trips per lane)
        bool okv[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) okv[e] = cok && m0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h < a.M;
        float gv[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            gv[e] = a.alpha0 * acc[0][0][e];
            adamw_one(pv[e], gv[e], mv[e], vv[e], lr, wd, bc1, sbc2, up->beta1, up->beta2, up->eps);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            if (okv[e]) {
                const int row = m0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                float* const gp = a.C + (size_t)row * a.ldc + col;
                *gp = gv[e]; gp[up->pdiff] = pv[e]; gp[up->mdiff] = mv[e]; gp[up->vdiff] = vv[e];
            } else {
                pv[e] = 0.f;                        // the streams' padding
            }
        }
#pragma unroll
        for (int eg = 0; eg < 4; ++eg) {
            const int r0 = m0 + wm * 32 + 8 * eg + 4 * h;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const AsPlace& q = up->pl[j];
                if (!q.out || !cok || r0 >= a.M) continue;
                if (q.trans) {
                    *reinterpret_cast<f32x4*>(q.out + as_slot(q, up->small, col, q.koff + r0)) =
                        f32x4{q.scale * pv[4 * eg], q.scale * pv[4 * eg + 1], q.scale * pv[4 * eg + 2], q.scale * pv[4 * eg + 3]};
                } else {
                    // rows r0 .. r0 + 3 (r0 a multiple of 4) sit 4 floats apart in either stream layout: one slot computation
                    const size_t s0 = as_slot(q, up->small, r0, q.koff + col);
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4)
                        if (r0 + e4 < a.M) q.out[s0 + 4 * e4] = q.scale * pv[4 * eg + e4];
                }
            }
        }
#ifdef GEMM_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        GEMM_STAMP(3);
#endif
        return;
    }
    float dot[TM][16];
    if (a.dotwith) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) dot[i][e] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + (wn * TN + j) * 32 + (lane & 31);
        const bool cok = col < a.N;
        const float b_first = two ? 0.f : e_b[j], b_last = two ? e_b[j] : 0.f;
        const float cs = e_cs[j], ct = e_ct[j], cp = e_cp[j], ct2 = e_ct2[j];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (!cok || row >= a.M) continue;
                float v = acc[i][j][e];
                v = two ? (v + b_last) : a.alpha0 * (v + b_first);
                if (a.R) v += a.R[(size_t)row * a.ldr + col];
                if (a.relu) v = fmaxf(v, 0.f);
                if (a.mask) v = (a.mask[(size_t)row * a.ldmask + col] > 0.f) ? v : 0.f;
                if (a.cscale || a.cshift) v = v * cs + ct;
                if (a.cexp) v = expf(v) * cp + ct2;
                if (a.C) a.C[(size_t)row * a.ldc + col] = v;
                if (a.dotwith) dot[i][e] += v * ((a.flags & LINNA_GEMM_DOT_SELF) ? v : a.dotwith[(size_t)row * a.lddot + col]);
            }
        }
    }
    if (a.dotwith) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = dot[i][e];
#pragma unroll
                for (int o = 16; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
                const int row = m0 + (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if ((lane & 31) == 0 && row < a.M)
                    a.dot_partial[(size_t)row * a.dot_slots + tn_ * WN + wn] = v;
            }
    }
}

template <int WM, int WN, int TM, int TN, int ALAY, int BLAY, int NS, int KS>
__global__ __launch_bounds__(WM * WN * KS * 64) void gemm_kernel(GemmArgs a, KSplit ks) {
    gemm_body<WM, WN, TM, TN, ALAY, BLAY, NS, KS>(a, blockIdx.x, ks);
}

// Grouped launch: ONE grid over the tiles of several independent problems that share the tile
// configuration and the operand layouts (the 13 parameter-gradient GEMMs of a training step: 250-370
// tiles that fill the chip once, instead of 13 launches of 1-128 tiles each), with the bias gradients
// (column sums of dY) taken inside the tiles of the first tile column.  The descriptors travel BY VALUE as
// kernel arguments (56 bytes per problem): no device table, nothing to upload, capturable as it is; the
// search for a block's problem runs on the scalar unit over the kernarg segment.
// The batch mean of a training step's loss rows as one extra workgroup of the grouped parameter-gradient launch (the
// one-launch forward + loss + dX chain cannot hold it: the rows are complete only when every workgroup has passed its
// turnaround).  sum_scale_prepare_kernel's arithmetic in its order: 1024 strided partial sums (four per thread here),
// sixteen wave sums, added in wave order.
template <int NT = 256>                   // threads of the workgroup (256 or 128): the same 1024 partial sums, 1024 / NT per thread
__device__ __forceinline__ void gemm_post_mean(const GemmPost& q) {
    __shared__ float post_part[16];
    constexpr int NP = 1024 / NT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float acc[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) acc[j] = 0.f;
#pragma unroll
    for (int j = 0; j < NP; ++j)
        for (int i = tid + NT * j; i < q.n; i += 1024) acc[j] += q.rows[i];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) acc[j] += __shfl_xor(acc[j], o, 64);
        if (lane == 0) post_part[wave + (NT / 64) * j] = acc[j];
    }
    __syncthreads();
    if (tid == 0) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += post_part[w];
        q.out[0] = t * q.scale;
    }
}

template <int WM, int WN, int TM, int TN, int ALAY, int BLAY, int NS, int KS>
__global__ __launch_bounds__(WM * WN * KS * 64) void gemm_group_kernel(const GemmGroupArgs g, const GemmPost post) {
    if (post.n > 0 && blockIdx.x == gridDim.x - 1) { gemm_post_mean<WM * WN * KS * 64>(post); return; }
    int p = 0;
    while (p + 1 < g.nprob && (int)blockIdx.x >= g.p[p + 1].first) ++p;
    const GemmGroupProb q = g.p[p];
    GemmArgs a;
    a.p[0].A = q.A; a.p[0].B = q.B; a.p[0].lda = q.lda; a.p[0].ldb = q.ldb; a.p[0].K = q.K; a.p[0].alay = ALAY; a.p[0].blay = BLAY;
    a.p[1] = a.p[0];
    a.npairs = 1; a.M = q.M; a.N = q.N; a.C = q.C; a.ldc = q.ldc;
    a.bias0 = nullptr; a.bias1 = nullptr; a.alpha0 = q.alpha; a.R = nullptr; a.ldr = 0; a.relu = 0; a.mask = nullptr; a.ldmask = 0;
    a.cscale = nullptr; a.cshift = nullptr; a.cexp = 0; a.cpost = nullptr; a.cshift2 = nullptr;
    a.dotwith = nullptr; a.lddot = 0; a.dot_partial = nullptr; a.dot_slots = 0; a.flags = 0;
    gemm_body<WM, WN, TM, TN, ALAY, BLAY, NS, KS, false, true>(a, (int)blockIdx.x - q.first, KSplit{1, nullptr, nullptr}, q.db);
}

template <int WM, int WN, int TM, int TN, int ALAY, int BLAY, int NS, int KS>
__global__ __launch_bounds__(WM * WN * KS * 64) void gemm_group_update_kernel(const GemmGroupArgsS g, const GemmUpdate u, const GemmPost post) {
    if (post.n > 0 && blockIdx.x == gridDim.x - 1) { gemm_post_mean<WM * WN * KS * 64>(post); return; }
    int p = 0;
    while (p + 1 < g.nprob && (int)blockIdx.x >= g.p[p + 1].first) ++p;
    const GemmGroupProb q = g.p[p];
    GemmArgs a;
    a.p[0].A = q.A; a.p[0].B = q.B; a.p[0].lda = q.lda; a.p[0].ldb = q.ldb; a.p[0].K = q.K; a.p[0].alay = ALAY; a.p[0].blay = BLAY;
    a.p[1] = a.p[0];
    a.npairs = 1; a.M = q.M; a.N = q.N; a.C = q.C; a.ldc = q.ldc;
    a.bias0 = nullptr; a.bias1 = nullptr; a.alpha0 = q.alpha; a.R = nullptr; a.ldr = 0; a.relu = 0; a.mask = nullptr; a.ldmask = 0;
    a.cscale = nullptr; a.cshift = nullptr; a.cexp = 0; a.cpost = nullptr; a.cshift2 = nullptr;
    a.dotwith = nullptr; a.lddot = 0; a.dot_partial = nullptr; a.dot_slots = 0; a.flags = 0;
    GemmUpd1 up;
    up.pdiff = u.pdiff; up.mdiff = u.mdiff; up.vdiff = u.vdiff; up.hyper = u.hyper; up.beta1 = u.beta1; up.beta2 = u.beta2;
    up.eps = u.eps; up.small = u.small; up.pl[0] = u.pl[p][0]; up.pl[1] = u.pl[p][1]; up.bias = u.bias[p];
    gemm_body<WM, WN, TM, TN, ALAY, BLAY, NS, KS, true, true>(a, (int)blockIdx.x - q.first, KSplit{1, nullptr, nullptr}, q.db, &up);
}

// ---------------------------------------------------------------------------- launcher
struct TileCfg { int wm, wn, tm, tn; };
static const TileCfg kCfgs[3] = {{2, 2, 2, 1}, {2, 2, 1, 1}, {1, 1, 1, 1}};   // 128x64, 64x64, 32x32

static int pick_cfg(int M, int N, int flags = 0) {
    if (flags & LINNA_GEMM_TILE_MASK) return (flags & LINNA_GEMM_TILE_MASK) - 1;
    auto tiles = [&](int c) {
        const int bm = kCfgs[c].wm * kCfgs[c].tm * 32, bn = kCfgs[c].wn * kCfgs[c].tn * 32;
        return (long)((M + bm - 1) / bm) * ((N + bn - 1) / bn);
    };
    // 64x64 workgroups are the default (measured: they beat single-wave 32x32 tiles even when they
    // leave CUs idle, tools/gemm_bench.py; 32x64 two-wave tiles that double the grid gain nothing either:
    // a wave's 32x32x32 chain of sixteen 64-cycle MFMAs per K tile is the unit of time, and only a
    // 16x16x4-based split of the tile over more SIMDs would shorten it); 128x64 only once 64x64 already
    // oversubscribes the chip; 32x32 for outputs that fit a single such tile.
    if (tiles(1) >= 2048) return 0;
    if (M <= 32 && N <= 32) return 2;
    return 1;
}

int gemm_slots(int M, int N) {
    const TileCfg& c = kCfgs[pick_cfg(M, N)];
    const int bn = c.wn * c.tn * 32;
    return ((N + bn - 1) / bn) * c.wn;
}

template <int WM, int WN, int TM, int TN, int ALAY, int BLAY, int NS, int KS = 1>
static int launch_one(const GemmArgs& a, hipStream_t stream, KSplit ks = KSplit{1, nullptr, nullptr}) {
    constexpr int NT = WM * WN * KS * 64, BM = WM * TM * 32, BN = WN * TN * 32;
    const int ntm = (a.M + BM - 1) / BM, ntn = (a.N + BN - 1) / BN;
    const size_t lds = (size_t)NS * KS * (BM + BN) * BK * sizeof(float);
    static bool attr_set = false;
    if (!attr_set && lds > 65536) {
        const int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<WM, WN, TM, TN, ALAY, BLAY, NS, KS>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute");
        if (rc != LINNA_OK) return rc;
        attr_set = true;
    }
    const int grid = ks.kz > 1 ? ks.kz * ((ntm * ntn + 7) & ~7) : ntm * ntn;
    hipLaunchKernelGGL((gemm_kernel<WM, WN, TM, TN, ALAY, BLAY, NS, KS>), dim3(grid), dim3(NT), lds, stream, a, ks);
    return check_hip(hipGetLastError(), "gemm launch");
}

// Per-stream scratch of the cross-workgroup K split: partial tiles + arrival counters (zeroed once; every
// launch leaves them zero).  Launches on one stream are ordered, so one buffer per stream is enough.
struct KsScratch { float* scratch = nullptr; int* counters = nullptr; };
static constexpr int KS_MAX_TILES = 64, KS_MAX_KZ = 8;
static KSplit ksplit_for(hipStream_t stream, int kz) {
    static std::mutex mu;
    static std::map<hipStream_t, KsScratch> pool;
    std::lock_guard<std::mutex> lock(mu);
    KsScratch& k = pool[stream];
    if (!k.scratch) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(stream, &cap);
        if (cap != hipStreamCaptureStatusNone) return KSplit{1, nullptr, nullptr};     // no allocation inside a capture
        const size_t bytes = (size_t)KS_MAX_TILES * KS_MAX_KZ * 64 * 64 * sizeof(float);
        if (hipMalloc(reinterpret_cast<void**>(&k.scratch), bytes) != hipSuccess) { k.scratch = nullptr; return KSplit{1, nullptr, nullptr}; }
        if (hipMalloc(reinterpret_cast<void**>(&k.counters), KS_MAX_TILES * sizeof(int)) != hipSuccess ||
            hipMemset(k.counters, 0, KS_MAX_TILES * sizeof(int)) != hipSuccess) {
            (void)hipFree(k.scratch); k.scratch = nullptr; k.counters = nullptr;
            return KSplit{1, nullptr, nullptr};
        }
    }
    return KSplit{kz, k.scratch, k.counters};
}

template <int ALAY, int BLAY>
static int launch_lay(const GemmArgs& a, int cfg, hipStream_t stream) {
    switch (cfg) {
        case 0: return launch_one<2, 2, 2, 1, ALAY, BLAY, 3>(a, stream);   // 72 KiB LDS: 2 blocks / CU
        case 1: {
            // A grid that leaves most CUs idle is bound by ONE wave's chain of 64-cycle MFMAs along K
            // (tools/gemm_small.py: 0.8 us per 32-k tile whatever the ring depth): split K over 4 or 2
            // wave groups inside the workgroup (16 / 8 waves, 128 KiB LDS, one block per CU).
            const long tiles = (long)((a.M + 63) / 64) * ((a.N + 63) / 64);
            const int K = a.npairs > 1 ? (a.p[0].K > a.p[1].K ? a.p[0].K : a.p[1].K) : a.p[0].K;
            if (!(a.flags & LINNA_GEMM_NOSPLIT) && K >= 128) {
                // few tiles and a long K: split K over several workgroups per tile (up to ~128 workgroups in all)
                const int ktiles = a.p[0].K / BK + (a.npairs > 1 ? a.p[1].K / BK : 0);
                int kz = std::min(std::min(KS_MAX_KZ, 128 / (int)std::max(1L, tiles)), ktiles / 4);
                // (measured, tools/gemm_small.py: the device-scope fence + scratch round trip costs 5-10 us, so the
                //  split pays only when a handful of tiles would otherwise run a >= 16-tile K loop each)
                if (tiles <= 16 && ktiles >= 12 && kz >= 2 && !a.dotwith) {
                    const KSplit ks = ksplit_for(stream, kz);
                    if (ks.kz > 1) return launch_one<2, 2, 1, 1, ALAY, BLAY, 4, 1>(a, stream, ks);
                }
                if (tiles <= 64) return launch_one<2, 2, 1, 1, ALAY, BLAY, 2, 4>(a, stream);
                if (tiles <= 128) return launch_one<2, 2, 1, 1, ALAY, BLAY, 4, 2>(a, stream);
            }
            return launch_one<2, 2, 1, 1, ALAY, BLAY, 4>(a, stream);       // 64 KiB LDS: 2 blocks / CU
        }
        default: return launch_one<1, 1, 1, 1, ALAY, BLAY, 4>(a, stream);  // 32 KiB LDS: 5 blocks / CU
    }
}

// Grouped launch of k-major x k-major problems on 64x64 tiles (4-stage ring); see gemm_group_kernel.
bool gemm_group_ok(const GemmArgs& a) {
    const int cfg = pick_cfg(a.M, a.N, a.flags);
    return a.npairs == 1 && a.p[0].alay == LAY_MN && a.p[0].blay == LAY_MN && (cfg == 1 || cfg == 2) && !a.dotwith &&
           ((a.p[0].lda | a.p[0].ldb) & 3) == 0;
}
int gemm_group_blocks(const GemmArgs& a) { return ((a.M + 63) / 64) * ((a.N + 63) / 64); }
int gemm_launch_group(const GemmGroupArgs& g, int nblocks, hipStream_t stream, const GemmPost* post) {
    constexpr size_t lds = (size_t)4 * (64 + 64) * BK * sizeof(float);
    const GemmPost q = post ? *post : GemmPost{nullptr, 0, 0.f, nullptr};
    hipLaunchKernelGGL((gemm_group_kernel<2, 2, 1, 1, LAY_MN, LAY_MN, 4, 1>), dim3(nblocks + (q.n > 0 ? 1 : 0)), dim3(256), lds, stream, g, q);
    return check_hip(hipGetLastError(), "gemm group launch");
}

int gemm_launch_group_update(const GemmGroupArgsS& g, const GemmUpdate& u, int nblocks, hipStream_t stream, const GemmPost* post) {
    constexpr size_t lds = (size_t)4 * (64 + 64) * BK * sizeof(float);
    const GemmPost q = post ? *post : GemmPost{nullptr, 0, 0.f, nullptr};
    hipLaunchKernelGGL((gemm_group_update_kernel<2, 2, 1, 1, LAY_MN, LAY_MN, 4, 1>), dim3(nblocks + (q.n > 0 ? 1 : 0)), dim3(256), lds, stream, g, u, q);
    return check_hip(hipGetLastError(), "gemm group update launch");
}

int gemm_launch(const GemmArgs& a, hipStream_t stream) {
    if (a.M <= 0 || a.N <= 0 || a.npairs < 1 || a.npairs > 2 || a.p[0].K <= 0) {
        set_error("gemm: bad shape M=%d N=%d K=%d pairs=%d", a.M, a.N, a.p[0].K, a.npairs);
        return LINNA_ERR_INVALID;
    }
    if (a.npairs == 2 && (a.p[0].alay != a.p[1].alay || a.p[0].blay != a.p[1].blay || a.p[1].K <= 0)) {
        set_error("gemm: operand pairs must share layouts");
        return LINNA_ERR_INVALID;
    }
    if (a.dotwith && (!a.dot_partial || a.dot_slots < gemm_slots(a.M, a.N))) {
        set_error("gemm: row-dot partial buffer too small");
        return LINNA_ERR_INVALID;
    }
    const int cfg = pick_cfg(a.M, a.N, a.flags);
    const int al = a.p[0].alay, bl = a.p[0].blay;
    if (al == LAY_K && bl == LAY_K) return launch_lay<LAY_K, LAY_K>(a, cfg, stream);
    if (al == LAY_K && bl == LAY_MN) return launch_lay<LAY_K, LAY_MN>(a, cfg, stream);
    if (al == LAY_MN && bl == LAY_MN) return launch_lay<LAY_MN, LAY_MN>(a, cfg, stream);
    set_error("gemm: unsupported layout combination A=%d B=%d", al, bl);
    return LINNA_ERR_UNSUPPORTED;
}

}  // namespace linna

#ifdef GEMM_STAMPS
extern "C" int linna_debug_gemm_stamps(unsigned long long* out, int nwords) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(linna::g_gemm_stamps), sizeof(unsigned long long) * (size_t)nwords, 0, hipMemcpyDeviceToHost);
}
extern "C" int linna_debug_gemm_realtime(unsigned long long* out, int nwords) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(linna::g_gemm_rt), sizeof(unsigned long long) * (size_t)nwords, 0, hipMemcpyDeviceToHost);
}
#endif
