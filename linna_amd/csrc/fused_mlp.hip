// Whole-network serving kernel for plain ReLU MLP emulators (BASELINE configs 2/5: 33 -> 512 x 4
// -> 33): ONE launch evaluates util.Log_prob.__call__ (util.py:990-1021) for 16 walkers per
// workgroup -- prior map + input transform (util.py:339-347, 483-497), every nn.Linear + ReLU
// (nn.py:121-130 shaped), the output transform and the Gaussian log-likelihood.
//
// MI355X mapping
//  * one 512-thread workgroup (8 waves, two per SIMD) per CU owns 16 walker rows; the activation
//    buffer [16][512] fp32 stays in LDS for the whole network (32 KiB): a wave keeps its output
//    block of a layer in registers until every wave has finished reading the layer's input;
//  * a layer's N columns are split over the 8 waves; every wave streams ITS OWN weight rows
//    through a private 4-stage LDS ring (4 x 4 KiB, 32 rows x 32 k per stage) with LDS-DMA
//    (global_load_lds_dwordx4).  Wave-private rings need no barrier in the K loop, only a
//    counted s_waitcnt vmcnt.  The weight stream is a flat list of "runs" (one run = the K
//    tiles of one 32-column block) over all layers, so the ring keeps prefetching across
//    layer boundaries (weights do not depend on activations);
//  * v_mfma_f32_16x16x4_f32 (exact fp32): A = activation rows (ds_read_b128, XOR-swizzled by
//    row), B = weight rows (ds_read_b128, swizzled on the DMA source address).  One wave per
//    SIMD can hide only ~5 other instructions per 32-cycle MFMA, so the tile step is lean:
//    per-lane offsets are precomputed, DMA sources are scalar base + per-lane offset, the 4
//    DMA issues of a tile are interleaved between its 16 MFMAs, run changes are a rare path;
//  * two raw s_barriers per layer (input fully read / output visible); the narrow last layer
//    splits K over the waves and reduces through LDS, then the log-likelihood is finished
//    with 16-lane shuffles.
// 160 KiB of LDS per workgroup: 32 KiB activations + 128 KiB rings.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace linna {

constexpr int FM = 16;            // walker rows per workgroup
constexpr int ACT_LD = 512;       // floats per activation row (max layer width)
constexpr int FNW = 8;            // waves per workgroup (two per SIMD: one wave's LDS/DMA stalls hide under the other's MFMAs)
constexpr int FNS = 4;            // ring stages per wave
constexpr int STAGE_F = 32 * 32;  // floats per stage: 32 weight rows x 32 k (4 KiB, 4 DMA instructions)
constexpr int FUSED_MAX_LAYERS = 6;
#ifndef FUSED_ABL
#define FUSED_ABL 0   // timing-only ablations for tools/: 1 no DMA, 2 no B-fragment reads, 4 no MFMA, 8 no A reads
#endif

struct FusedLayer { const float* W; const float* b; int K, N, ldw, pad; };

struct FusedArgs {
    const float* Z; int ldz; int B; int nin;
    const int* is_flat; const float* a1; const float* a2; const int* lg;
    const float* xmean; const float* xstd;
    FusedLayer L[FUSED_MAX_LAYERS];
    int nl;
    const float* cscale; const float* cshift; const float* w; float T;
    float* lnP; float* D; int ldd; float* TH; int ldt;
    const float* wlimit;          // highest address a 16-byte weight read may start at
    unsigned long long* stamps;   // diagnostic only (env LINNA_FUSED_STAMPS): [block][wave][16] cycle stamps
};

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

__device__ __forceinline__ int act_off(int row, int k) {          // swizzled activation address (floats)
    return row * ACT_LD + 4 * ((k >> 2) ^ (row & 15)) + (k & 3);
}
__device__ __forceinline__ float prior_theta_f(float z, int flat, float a1, float a2) {
    if (flat) return (0.5f * (1.f + erff(z / 1.41421356237309515f))) * a2 + a1;
    return z * a2 + a1;
}

template <int NL>
__global__ __launch_bounds__(64 * FNW, 2) void fused_mlp_kernel(FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* const act0 = smem;
    float* const ring = smem + FM * ACT_LD + wave * (FNS * STAGE_F);
    const int row0 = blockIdx.x * FM;
    const int li = lane & 15, kq = lane >> 4;
    int nstamp = 0;
    auto stamp = [&]() {           // diagnostic builds only: a.stamps is NULL in production
        if (a.stamps) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (lane == 0) a.stamps[((size_t)blockIdx.x * FNW + wave) * 16 + nstamp] = t;
            ++nstamp;
        }
    };
    stamp();

    // ---- biases of the hidden layers into registers BEFORE any LDS-DMA is in flight:
    // bias[l][cb][t] belongs to column wave*N/8 + cb*32 + 16*t + li
    constexpr int MAXCB = ACT_LD / (32 * FNW);        // 32-column blocks per wave (2)
    float bias[NL > 1 ? NL - 1 : 1][MAXCB][2];
#pragma unroll
    for (int l = 0; l < NL - 1; ++l)
#pragma unroll
        for (int cb = 0; cb < MAXCB; ++cb)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int nper = a.L[l].N / FNW;
                const int col = wave * nper + cb * 32 + 16 * t + li;
                bias[l][cb][t] = (cb * 32 < nper) ? a.L[l].b[col] : 0.f;
            }

    // make every bias register "used" here, so the compiler's vmcnt wait for those loads sits
    // before the first LDS-DMA instead of draining the ring at each epilogue
#pragma unroll
    for (int l = 0; l < NL - 1; ++l)
#pragma unroll
        for (int cb = 0; cb < MAXCB; ++cb)
#pragma unroll
            for (int t = 0; t < 2; ++t) asm volatile("" : "+v"(bias[l][cb][t]));

    // ---- per-lane constants of the tile step
    // A fragment of k tile kt, group g: act[li][32*(kt ^ hb) + aoff[g]]   (chunk ^= row swizzle)
    const int hb = (li >> 3) & 1;
    const int aoff0 = 4 * ((kq) ^ (li & 7)), aoff1 = 4 * ((4 + kq) ^ (li & 7));
    // B fragment (g, t): stage[r*32 + 4*((4g+kq) ^ ((r>>1)&7))], r = 16t + li
    int boff[2][2];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int r = 16 * t + li;
            boff[g][t] = r * 32 + 4 * ((4 * g + kq) ^ ((r >> 1) & 7));
        }

    // ---- weight stream: a flat list of runs.  Run = consecutive K tiles (32 rows x 32 k) of one
    // 32-column block.  Hidden layer l, block cb: rows wave*N/4 + 32cb.., tiles kt = 0..nkt-1
    // (stride 32 floats); last layer, block cb: rows 32cb.., tiles kt = wave, wave+8, ... (stride 256).
    int prl = 0, prcb = 0;                 // producer: next run to open
    int prem = 0;                          // tiles left in the open run
    const float* pw = nullptr;             // source of the open run's next tile (row 0, chunk 0)
    int pstride = 0, pldw = 0, prows = 32; // floats between tiles; weight row stride; valid rows in the block
    bool pedge = false;                    // tile needs clamped addressing
    int rowoff[4];                         // per-lane float offset of DMA instruction j (for pldw)
    unsigned rowoff_b[4];                  // the same in bytes (32-bit offset for scalar-base addressing)
#pragma unroll
    for (int j = 0; j < 4; ++j) { rowoff[j] = 0; rowoff_b[j] = 0; }
    int pslot = 0, issued = 0;
    auto open_run = [&]() {                // rare path: position the producer on the next non-empty run
        while (prl < NL) {
            const FusedLayer L = a.L[prl];
            const int nkt = (L.K + 31) >> 5;
            if (prl < NL - 1) {
                pw = L.W + (size_t)(wave * (L.N / FNW) + prcb * 32) * L.ldw;
                prem = nkt; pstride = 32; prows = 32;
            } else {
                pw = L.W + (size_t)(prcb * 32) * L.ldw + wave * 32;
                prem = (nkt - wave + FNW - 1) / FNW; pstride = 32 * FNW; prows = min(32, L.N - prcb * 32);
            }
            if (pldw != L.ldw) {
                pldw = L.ldw;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = j * 8 + (lane >> 3);
                    rowoff[j] = r * pldw + 4 * ((lane & 7) ^ ((r >> 1) & 7));
                    rowoff_b[j] = 4u * (unsigned)rowoff[j];
                }
            }
            // clamp when rows run past N, when the K padding runs past the row, or near the buffer end
            pedge = (prows < 32) || (nkt * 32 > L.ldw) || (pw + (size_t)32 * pldw + 32 * nkt > a.wlimit);
            ++prcb;
            const int nblocks = (prl < NL - 1) ? (L.N / (32 * FNW)) : ((L.N + 31) >> 5);
            if (prcb >= nblocks) { prcb = 0; ++prl; }
            if (prem > 0) return;
        }
        prem = 0;
    };
    auto dma = [&](int j) {                // one 1-KiB DMA instruction of the producer's current tile
        const float* src = pw + rowoff[j];
        if (pedge) {
            const int r = j * 8 + (lane >> 3);
            src = pw + min(r, prows - 1) * pldw + 4 * ((lane & 7) ^ ((r >> 1) & 7));
            src = src < a.wlimit ? src : a.wlimit;
        }
#if FUSED_ABL & 1
        return;
#endif
        __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(ring + pslot + j * 256), 16, 0, 0);
    };
    auto tile_issued = [&]() {             // after the 4 DMAs of a tile
        pw += pstride;
        pslot = (pslot + STAGE_F == FNS * STAGE_F) ? 0 : pslot + STAGE_F;
        ++issued;
        if (--prem == 0) open_run();
    };
    auto total_tiles = [&]() {
        int n = 0;
#pragma unroll
        for (int l = 0; l < NL - 1; ++l) n += (a.L[l].N / (32 * FNW)) * ((a.L[l].K + 31) >> 5);
        const int nkt = (a.L[NL - 1].K + 31) >> 5;
        n += ((nkt - wave + FNW - 1) / FNW) * ((a.L[NL - 1].N + 31) >> 5);
        return n;
    };
    const int ntiles = total_tiles();
    stamp();
    open_run();
    // fragments are read one tile ahead of the MFMAs, so the DMA issued under tile c's MFMAs refills
    // tile c's own stage with tile c+FNS: the prologue fills the whole ring
#pragma unroll 1
    for (int t = 0; t < FNS; ++t) {
        if (prem > 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) dma(j);
            tile_issued();
        }
    }

    // the weight ring is filling: compute the network input meanwhile
    // ---- prologue: x = X_transform(Transform(z)) for 16 rows, zero padded to a multiple of 32
    const int kpad0 = (a.L[0].K + 31) & ~31;
    float zz = 0.f;
    {
        const int r = tid >> 5, c0 = tid & 31;
        const int grow = min(row0 + r, a.B - 1);
        for (int c = c0; c < kpad0; c += 32) {
            float x = 0.f;
            if (c < a.nin) {
                const float z = a.Z[(size_t)grow * a.ldz + c];
                zz += z * z;
                const float th = prior_theta_f(z, a.is_flat[c], a.a1[c], a.a2[c]);
                if (a.TH && row0 + r < a.B) a.TH[(size_t)grow * a.ldt + c] = th;
                const float t = (a.lg && a.lg[c]) ? log10f(th) : th;
                x = (t - a.xmean[c]) / a.xstd[c];
            }
            act0[act_off(r, c)] = x;
        }
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) zz += __shfl_xor(zz, o, 64);     // 32-lane row groups
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();          // raw barrier: __syncthreads() would drain the DMA ring
    asm volatile("" ::: "memory");

    int consumed = 0, cslot = 0;
    auto layer_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    // Fragments of one tile (A: 2 x b128, B: 4 x b128).
    struct Frags { f32x4 a0, a1, b00, b01, b10, b11; };
    auto wait_next = [&]() {   // the tile about to be READ (index `consumed`) has landed
        // when tile c is read, tiles up to c+FNS-2 have been issued: FNS-2 = 2 younger tiles (4 DMAs
        // each) may stay in flight; the last tiles of the list drain
        if (consumed + 3 <= ntiles) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (consumed + 2 <= ntiles) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    auto read_frags = [&](const float* arow, int kt) {
        wait_next();
        const float* st = ring + cslot;
        const float* ak = arow + 32 * (kt ^ hb);
        Frags f;
#if FUSED_ABL & 8
        f.a0 = f32x4{1.f, 2.f, 3.f, (float)kt}; f.a1 = f.a0;
#else
        f.a0 = *reinterpret_cast<const f32x4*>(ak + aoff0);
        f.a1 = *reinterpret_cast<const f32x4*>(ak + aoff1);
#endif
#if FUSED_ABL & 2
        f.b00 = f32x4{1.f, 2.f, 3.f, (float)cslot}; f.b01 = f.b00; f.b10 = f.b00; f.b11 = f.b00;
#else
        f.b00 = *reinterpret_cast<const f32x4*>(st + boff[0][0]);
        f.b01 = *reinterpret_cast<const f32x4*>(st + boff[0][1]);
        f.b10 = *reinterpret_cast<const f32x4*>(st + boff[1][0]);
        f.b11 = *reinterpret_cast<const f32x4*>(st + boff[1][1]);
#endif
        cslot = (cslot + STAGE_F == FNS * STAGE_F) ? 0 : cslot + STAGE_F;
        ++consumed;
        return f;
    };
    // 16 MFMAs of one tile; the 4 DMA issues that refill the stage this tile's fragments came
    // from are spread between them (its ds_reads have completed: the fragments are in registers).
    auto mfma_tile = [&](const Frags& f, f32x4 (&acc)[2]) {
        const bool feed = prem > 0;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#if FUSED_ABL & 4
            acc[0][s] += f.a0[s] * f.b00[s]; acc[1][s] += f.a0[s] * f.b01[s];
#else
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a0[s], f.b00[s], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a0[s], f.b01[s], acc[1], 0, 0, 0);
#endif
            if ((s & 1) && feed) dma(s >> 1);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#if FUSED_ABL & 4
            acc[0][s] += f.a1[s] * f.b10[s]; acc[1][s] += f.a1[s] * f.b11[s];
#else
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a1[s], f.b10[s], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a1[s], f.b11[s], acc[1], 0, 0, 0);
#endif
            if ((s & 1) && feed) dma(2 + (s >> 1));
        }
        if (feed) tile_issued();
    };
    // One run: K tiles kt0, kt0+step, ... (n of them); the fragments of tile i+1 are read while the
    // MFMAs of tile i execute.
    // ---- steady-state fast path: 4 consecutive tiles of a hidden-layer run with every ring slot a
    // compile-time immediate (4 tiles = one turn of the 4-stage ring), DMA sources as scalar base +
    // 32-bit per-lane offset, and no per-DMA branches.  Entered only when the ring phase is aligned
    // (next tile to read sits in slot 1), the producer run has >= 4 plain tiles left and the consumer
    // run has >= 5 tiles left.
    static_assert(FNS == 4, "fast path assumes a 4-stage ring");
    const float* const bp00 = ring + boff[0][0];
    const float* const bp01 = ring + boff[0][1];
    const float* const bp10 = ring + boff[1][0];
    const float* const bp11 = ring + boff[1][1];
    auto fast4 = [&](Frags& fa, Frags& fb, const float* arow, int kt, f32x4 (&acc)[2]) {
        // A fragment of tile kt+u (kt multiple of 4): arow + 32*((kt+u) ^ hb) + aoff_g
        //   = arow + 32*kt + {32*hb, 32-32*hb, 64+32*hb, 96-32*hb}[u] + aoff_g
        const float* ae = arow + 32 * kt + 32 * hb;           // even u: + 32*u
        const float* ao = arow + 32 * kt - 32 * hb;           // odd  u: + 32*u
        const char* wb = reinterpret_cast<const char*>(pw);   // uniform
        const unsigned stride_b = 4u * (unsigned)pstride;
        auto rd = [&](auto U, Frags& f) {                      // fragments of tile kt+U+1, living in slot (U+1)&3
            constexpr int u1 = decltype(U)::value + 1;
            constexpr int slot = u1 & 3;
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            const float* ak = (u1 & 1) ? ao : ae;
            // (u1 == 4 belongs to the next group: 32*4 floats further)
#if FUSED_ABL & 8
            asm volatile("" : "+v"(f.a0), "+v"(f.a1));
#else
            f.a0 = *reinterpret_cast<const f32x4*>(ak + 32 * u1 + aoff0);
            f.a1 = *reinterpret_cast<const f32x4*>(ak + 32 * u1 + aoff1);
#endif
#if FUSED_ABL & 2
            asm volatile("" : "+v"(f.b00), "+v"(f.b01), "+v"(f.b10), "+v"(f.b11));
#else
            f.b00 = *reinterpret_cast<const f32x4*>(bp00 + slot * STAGE_F);
            f.b01 = *reinterpret_cast<const f32x4*>(bp01 + slot * STAGE_F);
            f.b10 = *reinterpret_cast<const f32x4*>(bp10 + slot * STAGE_F);
            f.b11 = *reinterpret_cast<const f32x4*>(bp11 + slot * STAGE_F);
#endif
        };
        auto mm = [&](auto U, const Frags& f) {                // MFMAs of tile kt+U; refill its slot U with tile +4
            constexpr int u = decltype(U)::value;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a0[s4], f.b00[s4], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a0[s4], f.b01[s4], acc[1], 0, 0, 0);
                if ((s4 & 1) && !(FUSED_ABL & 1)) {
                    const int jj = s4 >> 1;
                    __builtin_amdgcn_global_load_lds((gbl_void_t*)(wb + u * stride_b + rowoff_b[jj]),
                                                     (lds_void_t*)(ring + u * STAGE_F + jj * 256), 16, 0, 0);
                }
            }
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a1[s4], f.b10[s4], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a1[s4], f.b11[s4], acc[1], 0, 0, 0);
                if ((s4 & 1) && !(FUSED_ABL & 1)) {
                    const int jj = 2 + (s4 >> 1);
                    __builtin_amdgcn_global_load_lds((gbl_void_t*)(wb + u * stride_b + rowoff_b[jj]),
                                                     (lds_void_t*)(ring + u * STAGE_F + jj * 256), 16, 0, 0);
                }
            }
        };
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
        rd(I0{}, fb); mm(I0{}, fa);
        rd(I1{}, fa); mm(I1{}, fb);
        rd(I2{}, fb); mm(I2{}, fa);
        rd(I3{}, fa); mm(I3{}, fb);
        pw += 4 * pstride;
        prem -= 4; issued += 4; consumed += 4;
        if (prem == 0) open_run();
    };
    auto run_tiles = [&](const float* arow, int kt0, int step, int n, f32x4 (&acc)[2]) {
        if (n <= 0) return;
        Frags fa = read_frags(arow, kt0), fb;
        int i = 0;
#pragma unroll 1
        while (i < n) {
            if (step == 1 && n - i >= 5 && prem >= 4 && !pedge && cslot == STAGE_F && ((kt0 + i) & 3) == 0 &&
                consumed + 6 <= ntiles) {
                fast4(fa, fb, arow, kt0 + i, acc);
                i += 4;
                continue;
            }
            if (i + 1 < n) {
                fb = read_frags(arow, kt0 + (i + 1) * step);
                mfma_tile(fa, acc);
                fa = fb;
            } else {
                mfma_tile(fa, acc);
            }
            ++i;
        }
    };
    auto barrier_raw = [&]() {     // every wave has finished READING the activation buffer
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    stamp();
    // Waves w and w+4 share a SIMD and run the same program: left alone they reach their DMA-issue
    // and MFMA phases together and the matrix pipe idles.  Delay the second-dispatched half by about
    // half a tile after every rendezvous (MI355X_MICROARCH.md, two waves per SIMD, item 9).
#ifndef FUSED_STAGGER
#define FUSED_STAGGER 8
#endif
    auto stagger = [&]() {
        if (FUSED_STAGGER > 0 && wave >= FNW / 2) __builtin_amdgcn_s_sleep(FUSED_STAGGER);
    };
    stagger();
    // ---- hidden layers
    const float* arow = act0 + li * ACT_LD;
    const uint32_t act_lds = (uint32_t)(uintptr_t)(lds_void_t*)act0;
#pragma unroll
    for (int l = 0; l < NL - 1; ++l) {
        const int nkt = (a.L[l].K + 31) >> 5, nper = a.L[l].N / FNW, ncb = nper >> 5;
        f32x4 acc[MAXCB][2];
#pragma unroll
        for (int cb = 0; cb < MAXCB; ++cb) {
            acc[cb][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc[cb][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (cb < ncb) run_tiles(arow, 0, 1, nkt, acc[cb]);
        }
        stamp();
        barrier_raw();           // the layer's input is dead: overwrite it with the output
#pragma unroll
        for (int cb = 0; cb < MAXCB; ++cb) {
            if (cb >= ncb) break;
            // C/D layout of v_mfma_f32_16x16x4_f32: col = lane&15, row = 4*(lane>>4) + e
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int col = wave * nper + cb * 32 + 16 * t + li;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // inline-asm LDS store: a compiler-visible ds_write would be ordered behind the
                    // in-flight LDS-DMA with a vmcnt(0) and drain the weight ring
                    const float v = fmaxf(acc[cb][t][e] + bias[l][cb][t], 0.f);
                    const uint32_t addr = act_lds + 4u * (uint32_t)act_off(4 * kq + e, col);
                    asm volatile("ds_write_b32 %0, %1" :: "v"(addr), "v"(v) : "memory");
                }
            }
        }
        layer_barrier();         // output visible (N is a multiple of 256: the next K needs no zero padding)
        stagger();
        stamp();
    }

    // ---- last layer: K split over the waves, all N (<= 64) columns per wave
    {
        constexpr int l = NL - 1;
        float* part = act0;                             // [8 waves][16 rows][64 cols] = 32 KiB (reuses the buffer)
        const int N = a.L[l].N;
        const int nkt = (a.L[l].K + 31) >> 5;
        f32x4 acc[2][2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            acc[cb][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc[cb][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (cb * 32 < N) run_tiles(arow, wave, FNW, (nkt - wave + FNW - 1) / FNW, acc[cb]);
        }
        stamp();
        barrier_raw();           // all waves are done with the last layer's input (and all DMA has drained)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) part[wave * 1024 + (4 * kq + e) * 64 + cb * 32 + 16 * t + li] = acc[cb][t][e];
        layer_barrier();
        // finish: thread (row = tid>>5, c = tid&31) owns columns c, c+32
        const int r = tid >> 5, c0 = tid & 31;
        const bool rok = row0 + r < a.B;
        float chi = 0.f;
        for (int c = c0; c < N; c += 32) {
            float v = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < FNW; ++w8) v += part[w8 * 1024 + r * 64 + c];
            v += a.L[l].b[c];
            const float d = v * (a.cscale ? a.cscale[c] : 1.f) + (a.cshift ? a.cshift[c] : 0.f);
            if (a.D && rok) a.D[(size_t)(row0 + r) * a.ldd + c] = d;
            if (a.w) chi += (d * a.w[c]) * d;
        }
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) chi += __shfl_xor(chi, o, 64);
        if (a.lnP && a.w && c0 == 0 && rok) {
            const float v = (-0.5f * chi) / a.T + (-0.5f * zz);
            a.lnP[row0 + r] = isnan(v) ? -INFINITY : v;
        }
        stamp();
    }
}

// ---------------------------------------------------------------------------- host side
bool fused_mlp_eligible(const linna_layer_t* layers, int nl, int in_size) {
    if (nl < 2 || nl > FUSED_MAX_LAYERS) return false;
    int k = in_size;
    for (int i = 0; i < nl; ++i) {
        const linna_layer_t& l = layers[i];
        if (l.op != LINNA_OP_LINEAR || l.K != k || l.K > ACT_LD) return false;
        if (i < nl - 1) { if (!l.relu || (l.N % (32 * FNW)) || l.N > ACT_LD) return false; }
        else if (l.relu || l.N > 64) return false;
        k = l.N;
    }
    return true;
}

template <int NL>
static int launch_nl(const FusedArgs& a, hipStream_t s) {
    constexpr size_t lds = (size_t)(FM * ACT_LD + FNW * FNS * STAGE_F) * sizeof(float);   // 160 KiB
    static bool attr_set = false;
    if (!attr_set) {
        const int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_mlp_kernel<NL>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute");
        if (rc != LINNA_OK) return rc;
        attr_set = true;
    }
    hipLaunchKernelGGL((fused_mlp_kernel<NL>), dim3((a.B + FM - 1) / FM), dim3(64 * FNW), lds, s, a);
    return check_hip(hipGetLastError(), "fused_mlp launch");
}

int launch_fused_mlp(const linna_layer_t* layers, int nl, const float* param_end, const float* Z, int ldz, int B, int nin,
                     const int* is_flat, const float* a1, const float* a2, const int* lg, const float* xmean,
                     const float* xstd, const float* cscale, const float* cshift, const float* w, float T, float* lnP,
                     float* D, int ldd, float* TH, int ldt, hipStream_t s) {
    FusedArgs a;
    a.Z = Z; a.ldz = ldz; a.B = B; a.nin = nin;
    a.is_flat = is_flat; a.a1 = a1; a.a2 = a2; a.lg = lg; a.xmean = xmean; a.xstd = xstd;
    const float* hi = nullptr;
    for (int i = 0; i < nl; ++i) {
        a.L[i].W = layers[i].W; a.L[i].b = layers[i].b; a.L[i].K = layers[i].K; a.L[i].N = layers[i].N;
        a.L[i].ldw = (layers[i].K + 3) & ~3; a.L[i].pad = 0;
        const float* end = layers[i].W + (size_t)layers[i].N * a.L[i].ldw;
        if (!hi || end > hi) hi = end;
    }
    for (int i = nl; i < FUSED_MAX_LAYERS; ++i) a.L[i] = a.L[nl - 1];
    a.nl = nl;
    a.cscale = cscale; a.cshift = cshift; a.w = w; a.T = T;
    a.lnP = lnP; a.D = D; a.ldd = ldd; a.TH = TH; a.ldt = ldt;
    a.wlimit = (param_end ? param_end : hi) - 4;         // last address a 16-byte read may start at
    a.stamps = getenv("LINNA_FUSED_STAMPS") ? reinterpret_cast<unsigned long long*>(strtoull(getenv("LINNA_FUSED_STAMPS"), nullptr, 16)) : nullptr;
    switch (nl) {
        case 2: return launch_nl<2>(a, s);
        case 3: return launch_nl<3>(a, s);
        case 4: return launch_nl<4>(a, s);
        case 5: return launch_nl<5>(a, s);
        case 6: return launch_nl<6>(a, s);
        default: set_error("fused_mlp: %d layers unsupported", nl); return LINNA_ERR_UNSUPPORTED;
    }
}

}  // namespace linna
