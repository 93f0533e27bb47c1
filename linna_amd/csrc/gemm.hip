// fp32-input MFMA GEMM for gfx950: LDS-tiled, register-staged double buffering, one
// barrier per K-tile, fused epilogues.  Replaces every torch nn.Linear / F.relu / `@` on
// the reference hot path (linna/nn.py:53-54,121-130; util.py:1077-1085; autograd of them).
//
// Arithmetic is exact fp32 (v_mfma_f32_32x32x2_f32 == k-ordered fmaf chain), which is the
// parity path against the reference's fp32 CPU GEMMs.
//
// Tile = (WM*TM*32) x (WN*TN*32) outputs per workgroup, BK = 32.  Within an 8-deep k
// group, lane half h = lane>>5 owns k = 8g+4h+{0..3}: one ds_read_b128 per operand feeds
// four MFMA steps (the contraction order is a permutation of k, identical for A and B).
#include "common.h"

namespace linna {

constexpr int BK = 32;

template <int ROWS, int LAY, int NT>
struct Stager {
    // ROWS = tile extent along M (or N); the tile holds ROWS x BK floats.
    static constexpr int NV = ROWS * BK / 4 / NT;
    static constexpr int LD = (LAY == LAY_K) ? (BK + 4) : (ROWS + 4);
    static constexpr int SIZE = (LAY == LAY_K) ? ROWS * LD : BK * LD;
    f32x4 r[NV];

    __device__ __forceinline__ void load(const float* __restrict__ g, int ld, int row0, int nrows,
                                         int k0, int K, bool vec, int tid) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int f = tid + i * NT;
            if (LAY == LAY_K) {
                const int rr = f / (BK / 4), kk = (f % (BK / 4)) * 4;
                const int gr = row0 + rr, gk = k0 + kk;
                const float* p = g + (size_t)gr * ld + gk;
                if (gr < nrows && vec && gk + 3 < K) {
                    r[i] = *reinterpret_cast<const f32x4*>(p);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) r[i][e] = (gr < nrows && gk + e < K) ? p[e] : 0.f;
                }
            } else {
                constexpr int RQ = ROWS / 4;
                const int kk = f / RQ, rr = (f % RQ) * 4;
                const int gk = k0 + kk, gr = row0 + rr;
                const float* p = g + (size_t)gk * ld + gr;
                if (gk < K && vec && gr + 3 < nrows) {
                    r[i] = *reinterpret_cast<const f32x4*>(p);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) r[i][e] = (gk < K && gr + e < nrows) ? p[e] : 0.f;
                }
            }
        }
    }
    __device__ __forceinline__ void store(float* lds, int tid) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int f = tid + i * NT;
            if (LAY == LAY_K) {
                const int rr = f / (BK / 4), kk = (f % (BK / 4)) * 4;
                *reinterpret_cast<f32x4*>(lds + rr * LD + kk) = r[i];
            } else {
                constexpr int RQ = ROWS / 4;
                const int kk = f / RQ, rr = (f % RQ) * 4;
                *reinterpret_cast<f32x4*>(lds + kk * LD + rr) = r[i];
            }
        }
    }
};

// Fragment for one 32-row (or 32-col) MFMA slab, k group g: 4 values = 4 MFMA steps.
template <int LAY, int LD>
__device__ __forceinline__ f32x4 read_frag(const float* lds, int slab0, int g, int lane) {
    const int i = lane & 31, h = lane >> 5;
    if (LAY == LAY_K) {
        return *reinterpret_cast<const f32x4*>(lds + (slab0 + i) * LD + 8 * g + 4 * h);
    } else {
        f32x4 v;
#pragma unroll
        for (int s = 0; s < 4; ++s) v[s] = lds[(8 * g + 4 * h + s) * LD + slab0 + i];
        return v;
    }
}

template <int WM, int WN, int TM, int TN, int ALAY, int BLAY>
__global__ __launch_bounds__(WM * WN * 64) void gemm_kernel(GemmArgs a) {
    constexpr int NT = WM * WN * 64;
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    using SA = Stager<BM, ALAY, NT>;
    using SB = Stager<BN, BLAY, NT>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const lA0 = smem;
    float* const lB0 = smem + 2 * SA::SIZE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    // XCD-aware tile order: blocks b and b+8 share an XCD (private L2), so give every XCD a
    // contiguous run of tiles; consecutive tiles walk N first and therefore share the A panel.
    const int ntm = (a.M + BM - 1) / BM, ntn = (a.N + BN - 1) / BN;
    const int nwg = ntm * ntn;
    int tile;
    {
        const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
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

    const int nk0 = (a.p[0].K + BK - 1) / BK;
    const int nk1 = a.npairs > 1 ? (a.p[1].K + BK - 1) / BK : 0;
    const int nk = nk0 + nk1;

    SA sa; SB sb;
    auto gload = [&](int kt) {
        const int pi = kt >= nk0 ? 1 : 0;
        const GemmPair& p = a.p[pi];
        const int k0 = (pi ? kt - nk0 : kt) * BK;
        const bool va = ((p.lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.A) & 15) == 0);
        const bool vb = ((p.ldb & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.B) & 15) == 0);
        sa.load(p.A, p.lda, m0, a.M, k0, p.K, va, tid);
        sb.load(p.B, p.ldb, n0, a.N, k0, p.K, vb, tid);
    };

    gload(0);
    sa.store(lA0, tid);
    sb.store(lB0, tid);
    __syncthreads();

    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) gload(kt + 1);               // issue early, written after the MFMAs
        if (kt == nk0 && nk1 > 0) {
            // switch from pair 0 to pair 1: acc <- alpha0 * (acc + bias0)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + (wn * TN + j) * 32 + (lane & 31);
                const float b0 = (a.bias0 && col < a.N) ? a.bias0[col] : 0.f;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = a.alpha0 * (acc[i][j][e] + b0);
            }
        }
        const float* cA = lA0 + cur * SA::SIZE;
        const float* cB = lB0 + cur * SB::SIZE;
#pragma unroll
        for (int g = 0; g < BK / 8; ++g) {
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = read_frag<ALAY, SA::LD>(cA, (wm * TM + i) * 32, g, lane);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = read_frag<BLAY, SB::LD>(cB, (wn * TN + j) * 32, g, lane);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            sa.store(lA0 + (cur ^ 1) * SA::SIZE, tid);
            sb.store(lB0 + (cur ^ 1) * SB::SIZE, tid);
        }
        __syncthreads();
        cur ^= 1;
    }

    // ---------------------------------------------------------------- epilogue
    // C/D layout of v_mfma_f32_32x32x2_f32: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5).
    const int h = lane >> 5;
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
        float b_first = 0.f, b_last = 0.f, cs = 1.f, ct = 0.f, cp = 1.f, ct2 = 0.f;
        if (cok) {
            if (nk1 > 0) { b_last = a.bias1 ? a.bias1[col] : 0.f; }
            else { b_first = a.bias0 ? a.bias0[col] : 0.f; }
            if (a.cscale) cs = a.cscale[col];
            if (a.cshift) ct = a.cshift[col];
            if (a.cpost) cp = a.cpost[col];
            if (a.cshift2) ct2 = a.cshift2[col];
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (!cok || row >= a.M) continue;
                float v = acc[i][j][e];
                v = (nk1 > 0) ? (v + b_last) : a.alpha0 * (v + b_first);
                if (a.R) v += a.R[(size_t)row * a.ldr + col];
                if (a.relu) v = fmaxf(v, 0.f);
                if (a.mask) v = (a.mask[(size_t)row * a.ldmask + col] > 0.f) ? v : 0.f;
                if (a.cscale || a.cshift) v = v * cs + ct;
                if (a.cexp) v = expf(v) * cp + ct2;
                if (a.C) a.C[(size_t)row * a.ldc + col] = v;
                if (a.dotwith) dot[i][e] += v * a.dotwith[(size_t)row * a.lddot + col];
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

// ---------------------------------------------------------------------------- launcher
struct TileCfg { int wm, wn, tm, tn; };
static const TileCfg kCfgs[3] = {{2, 2, 2, 1}, {2, 2, 1, 1}, {1, 1, 1, 1}};   // 128x64, 64x64, 32x32

static int pick_cfg(int M, int N) {
    // Largest tile that still gives the chip >= 256 workgroups (one per CU); else the
    // smallest tile, many single-wave workgroups per CU.
    for (int c = 0; c < 3; ++c) {
        const int bm = kCfgs[c].wm * kCfgs[c].tm * 32, bn = kCfgs[c].wn * kCfgs[c].tn * 32;
        const long t = (long)((M + bm - 1) / bm) * ((N + bn - 1) / bn);
        if (t >= 256) return c;
    }
    return 2;
}

int gemm_slots(int M, int N) {
    const TileCfg& c = kCfgs[pick_cfg(M, N)];
    const int bn = c.wn * c.tn * 32;
    return ((N + bn - 1) / bn) * c.wn;
}

template <int WM, int WN, int TM, int TN, int ALAY, int BLAY>
static int launch_one(const GemmArgs& a, hipStream_t stream) {
    constexpr int NT = WM * WN * 64, BM = WM * TM * 32, BN = WN * TN * 32;
    const int ntm = (a.M + BM - 1) / BM, ntn = (a.N + BN - 1) / BN;
    const size_t lds = 2 * (Stager<BM, ALAY, NT>::SIZE + Stager<BN, BLAY, NT>::SIZE) * sizeof(float);
    hipLaunchKernelGGL((gemm_kernel<WM, WN, TM, TN, ALAY, BLAY>), dim3(ntm * ntn), dim3(NT), lds, stream, a);
    return check_hip(hipGetLastError(), "gemm launch");
}

template <int ALAY, int BLAY>
static int launch_lay(const GemmArgs& a, int cfg, hipStream_t stream) {
    switch (cfg) {
        case 0: return launch_one<2, 2, 2, 1, ALAY, BLAY>(a, stream);
        case 1: return launch_one<2, 2, 1, 1, ALAY, BLAY>(a, stream);
        default: return launch_one<1, 1, 1, 1, ALAY, BLAY>(a, stream);
    }
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
    const int cfg = pick_cfg(a.M, a.N);
    const int al = a.p[0].alay, bl = a.p[0].blay;
    if (al == LAY_K && bl == LAY_K) return launch_lay<LAY_K, LAY_K>(a, cfg, stream);
    if (al == LAY_K && bl == LAY_MN) return launch_lay<LAY_K, LAY_MN>(a, cfg, stream);
    if (al == LAY_MN && bl == LAY_MN) return launch_lay<LAY_MN, LAY_MN>(a, cfg, stream);
    set_error("gemm: unsupported layout combination A=%d B=%d", al, bl);
    return LINNA_ERR_UNSUPPORTED;
}

}  // namespace linna
