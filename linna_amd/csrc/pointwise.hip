// Element-wise and reduction kernels of the LINNA hot path for gfx950: prior map + input
// transform, Gaussian log-likelihood reductions (wavefront-shuffle), chi^2-ratio loss
// pieces, AdamW, bias-gradient column sums, Philox-driven ensemble / HMC moves.
// All HBM-bound: coalesced row reads, 64-lane shuffle reductions, no LDS round trips
// except the cross-wave column sum.
#include "common.h"
#include <string.h>
#include <math.h>

namespace linna {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

static inline dim3 grid1d(size_t n, int block) { return dim3((unsigned)((n + block - 1) / block)); }

// ------------------------------------------------------------------ prior map (util.py:339-347, 483-497)
__device__ __forceinline__ float prior_theta(float z, int flat, float a1, float a2) {
    if (flat) return (0.5f * (1.f + erff(z / 1.41421356237309515f))) * a2 + a1;   // a2 = hi - lo
    return z * a2 + a1;
}

__global__ void prior_map_fwd_kernel(const float* __restrict__ Z, int ldz, int B, int nin,
                                     const int* __restrict__ is_flat, const float* __restrict__ a1,
                                     const float* __restrict__ a2, const int* __restrict__ lg,
                                     const float* __restrict__ xmean, const float* __restrict__ xstd,
                                     float* __restrict__ X, int ldx, float* __restrict__ TH, int ldt) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * ldx) return;
    const int row = (int)(idx / ldx), col = (int)(idx % ldx);
    if (col >= nin) { X[idx] = 0.f; return; }
    const float z = Z[(size_t)row * ldz + col];
    const float th = prior_theta(z, is_flat[col], a1[col], a2[col]);
    if (TH) TH[(size_t)row * ldt + col] = th;
    const float t = (lg && lg[col]) ? log10f(th) : th;
    X[idx] = (t - xmean[col]) / xstd[col];
}

__global__ void prior_map_bwd_kernel(const float* __restrict__ Z, int ldz, int B, int nin,
                                     const int* __restrict__ is_flat, const float* __restrict__ a1,
                                     const float* __restrict__ a2, const int* __restrict__ lg,
                                     const float* __restrict__ xstd, const float* __restrict__ dX, int lddx,
                                     float* __restrict__ dZ, int lddz) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * nin) return;
    const int row = (int)(idx / nin), col = (int)(idx % nin);
    const float z = Z[(size_t)row * ldz + col];
    const int flat = is_flat[col];
    float g = dX[(size_t)row * lddx + col] / xstd[col];
    if (lg && lg[col]) g = g / (prior_theta(z, flat, a1[col], a2[col]) * 2.30258509299404568f);
    const float dth = flat ? a2[col] * (expf(-0.5f * z * z) * 0.398942280401432678f) : a2[col];
    dZ[(size_t)row * lddz + col] = g * dth - z;
}

// ------------------------------------------------------------------ Gaussian log-likelihood
// LPR lanes per walker row (a power of two: 64 for wide rows, down to 1): a wavefront covers 64/LPR
// rows, so a 33-wide row costs 16 lanes, not a whole wavefront of mostly idle ones.  Coalesced
// 16-byte reads of d, w and z when the rows are 16-byte aligned (ld multiple of 4), shuffle
// reduction over the LPR lanes of a row.
// out = (-0.5 * sum_j (d_j w_j) d_j)/T - 0.5 sum z^2 ; NaN -> -inf (util.py:953-955,1013-1016,1165)
template <int LPR>
__global__ __launch_bounds__(256) void loglike_diag_kernel(const float* __restrict__ D, int ldd, int B, int nout,
                                                           const float* __restrict__ w,
                                                           const float* __restrict__ Z, int ldz, int nin,
                                                           float T, float* __restrict__ out) {
    constexpr int RPW = 64 / LPR;                                  // rows per wavefront
    const int lane = threadIdx.x & 63, sub = lane % LPR;
    const int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / LPR;
    const bool rok = row < B;
    const int r = rok ? row : B - 1;                               // idle lanes read a valid row, store nothing
    const float* d = D + (size_t)r * ldd;
    const float* z = Z + (size_t)r * ldz;
    float acc = 0.f, zz = 0.f;
    const bool al = (ldd & 3) == 0 && ((reinterpret_cast<uintptr_t>(D) | reinterpret_cast<uintptr_t>(w)) & 15) == 0;
    if (al) {
        for (int j = sub * 4; j < nout; j += LPR * 4) {
            const f32x4 dv = *reinterpret_cast<const f32x4*>(d + j);   // (row padded to a multiple of 4: in bounds)
            if (j + 3 < nout) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(w + j);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc += (dv[e] * wv[e]) * dv[e];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (j + e < nout) acc += (dv[e] * w[j + e]) * dv[e];
            }
        }
    } else {
        for (int j = sub; j < nout; j += LPR) { const float dv = d[j]; acc += (dv * w[j]) * dv; }
    }
    if ((ldz & 3) == 0 && (reinterpret_cast<uintptr_t>(Z) & 15) == 0) {
        for (int j = sub * 4; j < nin; j += LPR * 4) {
            const f32x4 zv = *reinterpret_cast<const f32x4*>(z + j);
#pragma unroll
            for (int e = 0; e < 4; ++e) if (j + e < nin) zz += zv[e] * zv[e];
        }
    } else {
        for (int j = sub; j < nin; j += LPR) { const float zv = z[j]; zz += zv * zv; }
    }
#pragma unroll
    for (int o = LPR / 2; o >= 1; o >>= 1) { acc += __shfl_xor(acc, o, 64); zz += __shfl_xor(zz, o, 64); }
    if (sub == 0 && rok) {
        const float v = (-0.5f * acc) / T + (-0.5f * zz);
        out[row] = isnan(v) ? -INFINITY : v;
    }
}

__global__ __launch_bounds__(256) void loglike_finish_kernel(const float* __restrict__ partial, int slots_ld, int nslots,
                                                             int B, const float* __restrict__ Z, int ldz, int nin,
                                                             float T, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    float acc = 0.f, zz = 0.f;
    for (int s = lane; s < nslots; s += 64) acc += partial[(size_t)row * slots_ld + s];
    for (int j = lane; j < nin; j += 64) { const float zv = Z[(size_t)row * ldz + j]; zz += zv * zv; }
    acc = wave_sum(acc);
    zz = wave_sum(zz);
    if (lane == 0) {
        const float v = (-0.5f * acc) / T + (-0.5f * zz);
        out[row] = isnan(v) ? -INFINITY : v;
    }
}

// dH = -(1/T) * gscale_j * w_j * d_j   (gradient of the diagonal log-likelihood wrt raw net output)
__global__ void loglike_diag_grad_kernel(const float* __restrict__ D, int ldd, int B, int nout,
                                         const float* __restrict__ w, const float* __restrict__ gscale,
                                         float T, float* __restrict__ dH, int lddh) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * lddh) return;
    const int row = (int)(idx / lddh), col = (int)(idx % lddh);
    dH[idx] = col < nout ? -(D[(size_t)row * ldd + col] * w[col]) * gscale[col] / T : 0.f;
}

// ------------------------------------------------------------------ chi^2-ratio loss pieces (util.py:1070-1088)
// mode 0: delta = ynorm - pred ; mode 1: ynorm - data_norm ; mode 2: pred - data_norm ; masked -> 0
__global__ void loss_delta_kernel(int mode, const float* __restrict__ PRED, int ldp, const float* __restrict__ Y, int ldy,
                                  const int* __restrict__ ROWS, int B, int nout, const float* __restrict__ sigma,
                                  const float* __restrict__ ymean, const float* __restrict__ ystd, int ylog,
                                  const float* __restrict__ data_norm, float* __restrict__ DELTA, int ldd) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * ldd) return;
    const int i = (int)(idx / ldd), j = (int)(idx % ldd);
    if (j >= nout) { DELTA[idx] = 0.f; return; }
    const int r = ROWS ? ROWS[i] : i;
    const float y = Y[(size_t)r * ldy + j];
    const float dn = data_norm[j];
    const bool masked = (y == 1e-30f) | (y == 1e10f) | (dn == 1e-30f);
    const float yn = ((ylog ? logf(y / sigma[j]) : y / sigma[j]) - ymean[j]) / ystd[j];   // ylog: util.py:567-571 (ypositive)
    float v;
    if (mode == 0) v = yn - PRED[(size_t)i * ldp + j];
    else if (mode == 1) v = yn - dn;
    else v = PRED[(size_t)i * ldp + j] - dn;
    DELTA[idx] = masked ? 0.f : v;
}

// Normalised targets of a whole data set, once: YN = (y / sigma - ymean) / ystd (loss_delta_kernel's formula), NaN where
// the element is masked (util.py:1072) -- what the one-launch training forward subtracts its prediction from.
__global__ void loss_targets_kernel(const float* __restrict__ Y, int ldy, int n, int nout, const float* __restrict__ sigma,
                                    const float* __restrict__ ymean, const float* __restrict__ ystd, int ylog,
                                    const float* __restrict__ data_norm, float* __restrict__ YN, int ldyn) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)n * ldyn) return;
    const int i = (int)(idx / ldyn), j = (int)(idx % ldyn);
    if (j >= nout) { YN[idx] = 0.f; return; }
    const float y = Y[(size_t)i * ldy + j];
    const bool masked = (y == 1e-30f) | (y == 1e10f) | (data_norm[j] == 1e-30f);
    YN[idx] = masked ? __builtin_nanf("") : ((ylog ? logf(y / sigma[j]) : y / sigma[j]) - ymean[j]) / ystd[j];
}

// chi2_b = sum_s partial[b][s]; mode 0: out[b] = max(chi2, floor) (denominator, util.py:1086)
// mode 1: out[b] = chi2 / den[r]
__global__ void loss_rows_kernel(int mode, const float* __restrict__ partial, int slots_ld, int nslots, int B,
                                 const float* __restrict__ den, const int* __restrict__ ROWS, float floorv,
                                 float* __restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float c = 0.f;
    for (int s = 0; s < nslots; ++s) c += partial[(size_t)b * slots_ld + s];
    if (mode == 0) out[b] = c < floorv ? floorv : c;
    else out[b] = c / den[ROWS ? ROWS[b] : b];
}

// dPRED[i][j] = masked ? 0 : -2 * U[i][j] * inv_batch / den   (U = delta Cinv, Cinv symmetric)
__global__ void loss_grad_kernel(const float* __restrict__ U, int ldu, const float* __restrict__ Y, int ldy,
                                 const int* __restrict__ ROWS, int B, int nout, const float* __restrict__ data_norm,
                                 const float* __restrict__ den, float inv_batch, float* __restrict__ dP, int lddp) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * lddp) return;
    const int i = (int)(idx / lddp), j = (int)(idx % lddp);
    if (j >= nout) { dP[idx] = 0.f; return; }
    const int r = ROWS ? ROWS[i] : i;
    const float y = Y[(size_t)r * ldy + j];
    const bool masked = (y == 1e-30f) | (y == 1e10f) | (data_norm[j] == 1e-30f);
    dP[idx] = masked ? 0.f : (-2.f * U[(size_t)i * ldu + j]) * inv_batch / den[r];
}

// The whole chi2-ratio loss of a minibatch (util.py:1070-1116) for nout <= 64 in ONE launch: one wave per row, lane j
// owns column j.  delta (loss_delta_kernel, mode 0), U = delta Cinv from a copy of Cinv in LDS (k-loop, delta_k by
// shuffle), chi2 = delta . U (wave sum), loss_b = chi2 / den, d loss / d pred (loss_grad_kernel), and the batch mean:
// the block that arrives last sums loss_rows in index order (same value whatever the arrival order).
// `counter` is a zeroed int that wraps back to zero by itself (atomicInc).
__global__ __launch_bounds__(256) void loss_fused_small_kernel(
        const float* __restrict__ PRED, int ldp, const float* __restrict__ Y, int ldy, const int* __restrict__ ROWS, int B,
        int nout, const float* __restrict__ sigma, const float* __restrict__ ymean, const float* __restrict__ ystd, int ylog,
        const float* __restrict__ data_norm, const float* __restrict__ Cinv, int ldc, const float* __restrict__ den,
        float inv_batch, float* __restrict__ loss_rows, float* __restrict__ loss_mean, float* __restrict__ dP, int lddp,
        unsigned* counter) {
    __shared__ float C[64 * 65];
    __shared__ float part[4];
    __shared__ int last;
    for (int i = threadIdx.x; i < nout * nout; i += 256) C[(i / nout) * 65 + (i % nout)] = Cinv[(size_t)(i / nout) * ldc + (i % nout)];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + wave;
    if (b < B) {
        const int r = ROWS ? ROWS[b] : b;
        const bool in = lane < nout;
        const int j = in ? lane : 0;
        const float y = Y[(size_t)r * ldy + j], dn = data_norm[j];
        const bool masked = !in | (y == 1e-30f) | (y == 1e10f) | (dn == 1e-30f);
        const float yn = ((ylog ? logf(y / sigma[j]) : y / sigma[j]) - ymean[j]) / ystd[j];
        const float delta = masked ? 0.f : yn - PRED[(size_t)b * ldp + j];
        float u = 0.f;
        for (int k = 0; k < nout; ++k) u += __shfl(delta, k, 64) * C[k * 65 + j];
        const float chi = wave_sum(in ? delta * u : 0.f);
        const float dr = den[r];
        if (lane == 0) loss_rows[b] = chi / dr;
        if (dP && lane < lddp) dP[(size_t)b * lddp + lane] = masked ? 0.f : (-2.f * u) * inv_batch / dr;
    }
    if (!loss_mean) return;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) last = atomicInc(counter, gridDim.x - 1) == gridDim.x - 1;
    __syncthreads();
    if (!last) return;
    __threadfence();
    float acc = 0.f;
    for (int i = threadIdx.x; i < B; i += 256) acc += __builtin_nontemporal_load(loss_rows + i);
    acc = wave_sum(acc);
    if (lane == 0) part[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) loss_mean[0] = (((part[0] + part[1]) + part[2]) + part[3]) * inv_batch;
}

// deterministic single-block sum: out[0] = scale * sum_i v[i]
__global__ __launch_bounds__(1024) void sum_scale_kernel(const float* __restrict__ v, int n, float scale, float* __restrict__ out) {
    __shared__ float part[16];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 1024) acc += v[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += part[w];
        out[0] = t * scale;
    }
}
// The same with the AdamW step counter and bias corrections advanced on the side (adamw_prepare_kernel): both are
// single-thread jobs of every optimiser step, one launch instead of two.
__global__ __launch_bounds__(1024) void sum_scale_prepare_kernel(const float* __restrict__ v, int n, float scale, float* __restrict__ out,
                                                                 int* step, float* hyper, float beta1, float beta2) {
    __shared__ float part[16];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 1024) acc += v[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += part[w];
        out[0] = t * scale;
    }
    if (threadIdx.x == 64) {
        const int t = ++step[0];
        hyper[2] = (float)(1.0 - pow((double)beta1, (double)t));
        hyper[3] = (float)sqrt(1.0 - pow((double)beta2, (double)t));
    }
}

// frac[b] = |nnd_b / den_b - 1|  (util.py:1126)
__global__ void val_frac_kernel(const float* __restrict__ partial, int slots_ld, int nslots, int B,
                                const float* __restrict__ den, float* __restrict__ frac) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float c = 0.f;
    for (int s = 0; s < nslots; ++s) c += partial[(size_t)b * slots_ld + s];
    frac[b] = fabsf(c / den[b] - 1.f);
}

// The three validation metrics of an epoch (util.py:1124-1127: torch.median(loss), torch.max(frac), torch.median(frac)) on the
// device, so that the epoch's controller reads ONE small record instead of two row vectors: out[0] = *last (the epoch's last
// training loss, NaN if null), out[1] = lower median of loss[n], out[2] = max of frac[n], out[3] = lower median of frac[n].
// Selection by rank counting (thread i counts the elements before x_i in sorted order; n <= a few 10^4: n^2 compares spread over
// n threads); a NaN anywhere makes the statistic NaN, as torch.median / torch.max do.
__global__ __launch_bounds__(256) void val_metrics_kernel(const float* __restrict__ loss, const float* __restrict__ frac, int n,
                                                          const float* __restrict__ last, float* __restrict__ out) {
    __shared__ float tl[256], tf[256];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool in = i < n;
    const float xl = in ? loss[i] : 0.f, xf = in ? frac[i] : 0.f;
    int rl = 0, rf = 0, nanl = 0, nanf = 0;
    for (int j0 = 0; j0 < n; j0 += 256) {
        const int j = j0 + threadIdx.x;
        tl[threadIdx.x] = j < n ? loss[j] : INFINITY;
        tf[threadIdx.x] = j < n ? frac[j] : INFINITY;
        __syncthreads();
        const int m = min(256, n - j0);
        for (int k = 0; k < m; ++k) {
            const float a = tl[k], b = tf[k];
            rl += (a < xl) | ((a == xl) & (j0 + k < i));
            rf += (b < xf) | ((b == xf) & (j0 + k < i));
            nanl |= a != a;
            nanf |= b != b;
        }
        __syncthreads();
    }
    if (!in) return;
    const int mid = (n - 1) / 2;
    if (i == 0) {
        out[0] = last ? last[0] : __builtin_nanf("");
        if (nanl) out[1] = __builtin_nanf("");
        if (nanf) { out[2] = __builtin_nanf(""); out[3] = __builtin_nanf(""); }
    }
    if (!nanl && rl == mid) out[1] = xl;
    if (!nanf && rf == n - 1) out[2] = xf;
    if (!nanf && rf == mid) out[3] = xf;
}

// ------------------------------------------------------------------ minibatch gather + X transform
__global__ void gather_xform_kernel(const float* __restrict__ X, int ldx, const int* __restrict__ ROWS, int B, int nin,
                                    const int* __restrict__ lg, const float* __restrict__ xmean,
                                    const float* __restrict__ xstd, float* __restrict__ XB, int ldxb) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * ldxb) return;
    const int i = (int)(idx / ldxb), j = (int)(idx % ldxb);
    if (j >= nin) { XB[idx] = 0.f; return; }
    const int r = ROWS ? ROWS[i] : i;
    float t = X[(size_t)r * ldx + j];
    if (lg && lg[j]) t = log10f(t);
    XB[idx] = (t - xmean[j]) / xstd[j];
}

// ------------------------------------------------------------------ bias gradient: db[n] = scale * sum_b dZ[b][n]
// One block per 64 columns, 16 waves: wave w sums rows w, w+16, ... with four independent
// accumulators (four row loads in flight per lane), then a fixed-order LDS reduction over the
// waves -- deterministic, and ~30 row loads deep instead of 125.
__global__ __launch_bounds__(1024) void colsum_kernel(const float* __restrict__ dZ, int ld, int B, int N, float scale,
                                                      float* __restrict__ db) {
    __shared__ float part[16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (col < N) {
        const float* p = dZ + col;
        int b = wave;
        for (; b + 48 < B; b += 64) {
            a0 += p[(size_t)b * ld]; a1 += p[(size_t)(b + 16) * ld];
            a2 += p[(size_t)(b + 32) * ld]; a3 += p[(size_t)(b + 48) * ld];
        }
        for (; b < B; b += 16) a0 += p[(size_t)b * ld];
    }
    part[wave][lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (wave == 0 && col < N) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += part[w][lane];
        db[col] = scale * t;
    }
}

// ------------------------------------------------------------------ AdamW (torch.optim.AdamW single-tensor update)
// hyper = [lr, weight_decay, bc1 = 1-b1^t, sqrt(bc2) = sqrt(1-b2^t)]
__global__ void adamw_prepare_kernel(int* step, float* hyper, float beta1, float beta2) {
    const int t = ++step[0];
    hyper[2] = (float)(1.0 - pow((double)beta1, (double)t));
    hyper[3] = (float)sqrt(1.0 - pow((double)beta2, (double)t));
}

__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, size_t n, const float* __restrict__ hyper, float beta1,
                             float beta2, float eps) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float lr = hyper[0], wd = hyper[1], bc1 = hyper[2], sbc2 = hyper[3];
    const float gi = g[i];
    float pi = p[i] * (1.f - lr * wd);
    float mi = m[i];
    mi = mi + (gi - mi) * (1.f - beta1);
    const float vi = v[i] * beta2 + (1.f - beta2) * gi * gi;
    const float denom = sqrtf(vi) / sbc2 + eps;
    pi = pi - (lr / bc1) * (mi / denom);
    p[i] = pi; m[i] = mi; v[i] = vi;
}

// Philox4x32-10 and the per-walker counter convention live in common.h (shared with net_stream.hip)

// ------------------------------------------------------------------ stretch move (emcee StretchMove / RedBlueMove)
__global__ void stretch_propose_kernel(const float* __restrict__ coords, int ldc, int ndim, const int* __restrict__ S,
                                       int ns, const float* __restrict__ ccoords, int ldcc, const int* __restrict__ C,
                                       int nc, uint64_t seed,
                                       const int* __restrict__ step_dev, int stream_id, float a,
                                       float* __restrict__ Q, int ldq, float* __restrict__ factors) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)ns * ldq) return;
    const int k = (int)(idx / ldq), d = (int)(idx % ldq);
    if (d >= ndim) { Q[idx] = 0.f; return; }
    const int wk = S[k];
    const U4 r = walker_bits(seed, (uint32_t)wk, (uint32_t)step_dev[0], (uint32_t)stream_id, 0u);
    const float t = (a - 1.f) * u01(r.x) + 1.f;
    const float zz = t * t / a;
    int j = (int)(((uint64_t)r.y * (uint64_t)nc) >> 32);
    const int wc = C[j];
    const float cr = ccoords[(size_t)wc * ldcc + d], s = coords[(size_t)wk * ldc + d];
    Q[idx] = cr - (cr - s) * zz;
    if (d == 0) factors[k] = ((float)ndim - 1.f) * logf(zz);
}

__global__ void stretch_accept_kernel(float* __restrict__ coords, int ldc, int ndim, float* __restrict__ logp,
                                      const int* __restrict__ S, int ns, const float* __restrict__ Q, int ldq,
                                      const float* __restrict__ lp_new, const float* __restrict__ factors,
                                      uint64_t seed, const int* __restrict__ step_dev, int stream_id,
                                      int* __restrict__ naccept) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ns) return;
    const int wk = S[k];
    const U4 r = walker_bits(seed, (uint32_t)wk, (uint32_t)step_dev[0], (uint32_t)stream_id, 0u);
    const float lnpdiff = factors[k] + lp_new[k] - logp[wk];
    if (lnpdiff > logf(u01(r.z))) {
        for (int d = 0; d < ndim; ++d) coords[(size_t)wk * ldc + d] = Q[(size_t)k * ldq + d];
        logp[wk] = lp_new[k];
        if (naccept) atomicAdd(naccept + wk, 1);
    }
}

// ------------------------------------------------------------------ batched per-walker HMC (HMCSampler.py:26-59)
__device__ __forceinline__ float normal_draw(uint64_t seed, uint32_t w, uint32_t step, uint32_t stream, int d) {
    const U4 r = walker_bits(seed, w, step, stream, (uint32_t)(d >> 1) + 1u);
    const float u1 = (d & 1) ? u01(r.z) : u01(r.x), u2 = (d & 1) ? u01(r.w) : u01(r.y);
    return sqrtf(-2.f * logf(u1)) * cosf(6.28318530717958648f * u2);
}

// One WAVE per chain, lane = dimension (a thread per chain walked its row with a stride of a row: 20-29 us per launch at
// 4096 chains x 33, more than four of the leapfrog's own launches); the kinetic energy is summed in dimension order.
__device__ __forceinline__ float wave_ordered_sum(float t, int n) {
    float acc = 0.f;
    for (int j = 0; j < n; ++j) acc += __shfl(t, j, 64);
    return acc;
}

// P ~ N(0, m), H0 = P^2 / 2m - lnP; G != nullptr: also the first half kick and the first drift of the leapfrog,
// P += ek G, Q = X + ed P / m (hmc_kick_drift_kernel's arithmetic)
__global__ void hmc_start_kernel(int B, int ndim, const float* __restrict__ mass, uint64_t seed,
                                 const int* __restrict__ step_dev, const float* __restrict__ lnp,
                                 const float* __restrict__ P0, int ldp0, const float* __restrict__ G, int ldg, float ek, float ed,
                                 const float* __restrict__ X, int ldx, float* __restrict__ P, int ldp, float* __restrict__ Q,
                                 int ldq, float* __restrict__ H0) {
    const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B) return;
    float ke = 0.f;
    for (int d0 = 0; d0 < ndim; d0 += 64) {
        const int d = d0 + lane;
        float t = 0.f;
        if (d < ndim) {
            const float m = mass[d];
            const float n01 = P0 ? P0[(size_t)b * ldp0 + d] : normal_draw(seed, (uint32_t)b, (uint32_t)step_dev[0], 1u, d);
            float p = n01 * sqrtf(m);
            t = p * p / m;
            if (G) {
                if (ek != 0.f) p += ek * G[(size_t)b * ldg + d];
                const float x = X[(size_t)b * ldx + d];
                Q[(size_t)b * ldq + d] = ed != 0.f ? x + ed * (p / m) : x;
            }
            P[(size_t)b * ldp + d] = p;
        }
        ke += wave_ordered_sum(t, min(64, ndim - d0));
    }
    if (lane == 0) H0[b] = 0.5f * ke - lnp[b];
}

// P += eps_kick * G ; Q += eps_drift * P / m   (either eps may be 0)
__global__ void hmc_kick_drift_kernel(int B, int ndim, const float* __restrict__ mass, float ek, float ed,
                                      const float* __restrict__ G, int ldg, float* __restrict__ P, int ldp,
                                      float* __restrict__ Q, int ldq) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * ndim) return;
    const int b = (int)(idx / ndim), d = (int)(idx % ndim);
    float p = P[(size_t)b * ldp + d];
    if (ek != 0.f) { p += ek * G[(size_t)b * ldg + d]; P[(size_t)b * ldp + d] = p; }
    if (ed != 0.f) Q[(size_t)b * ldq + d] += ed * (p / mass[d]);
}

__global__ void hmc_accept_kernel(int B, int ndim, const float* __restrict__ mass, uint64_t seed,
                                  const int* __restrict__ step_dev, const float* __restrict__ H0,
                                  const float* __restrict__ P, int ldp, const float* __restrict__ Qn, int ldq,
                                  const float* __restrict__ lnp_new, const float* __restrict__ Gn, int ldg,
                                  const float* __restrict__ U,
                                  float* __restrict__ X, int ldx, float* __restrict__ lnp, float* __restrict__ G,
                                  int* __restrict__ naccept) {
    const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;   // a wave per chain
    if (b >= B) return;
    float ke = 0.f;
    for (int d0 = 0; d0 < ndim; d0 += 64) {
        const int d = d0 + lane;
        float t = 0.f;
        if (d < ndim) { const float p = P[(size_t)b * ldp + d]; t = p * p / mass[d]; }
        ke += wave_ordered_sum(t, min(64, ndim - d0));
    }
    const float ln = lnp_new[b];
    const float H1 = 0.5f * ke - ln;
    const U4 r = walker_bits(seed, (uint32_t)b, (uint32_t)step_dev[0], 2u, 0u);
    const float ratio = expf(fminf(H0[b] - H1, 0.f));
    const float u = U ? U[b] : u01(r.x);
    if (isfinite(ln) && u < ratio) {
        for (int d = lane; d < ndim; d += 64) {
            X[(size_t)b * ldx + d] = Qn[(size_t)b * ldq + d];
            G[(size_t)b * ldg + d] = Gn[(size_t)b * ldg + d];
        }
        if (lane == 0) {
            lnp[b] = ln;
            if (naccept) atomicAdd(naccept + b, 1);
        }
    }
}

__global__ void step_increment_kernel(int* step) { step[0] += 1; }

// ------------------------------------------------------------------ ensemble slice sampling (zeus, Karamanis & Beutler 2021)
// state per active walker k: direction DIR[k][:], slice height Z0, bracket [L, R] in units of the
// direction, flags aL/aR (still stepping out) and aS (still shrinking).
__global__ void slice_init_kernel(const float* __restrict__ logp, const int* __restrict__ S, int ns,
                                  const float* __restrict__ cc, int ldcc, const int* __restrict__ C, int nc, int ndim,
                                  const float* __restrict__ mu, uint64_t seed, const int* __restrict__ step_dev,
                                  int stream_id, float* __restrict__ DIR, int ldd, float* __restrict__ Z0,
                                  float* __restrict__ L, float* __restrict__ R, int* __restrict__ flags, int maxsteps) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)ns * ldd) return;
    const int k = (int)(idx / ldd), d = (int)(idx % ldd);
    const int wk = S[k];
    const U4 r = walker_bits(seed, (uint32_t)wk, (uint32_t)step_dev[0], (uint32_t)stream_id, 0u);
    // differential move: two DISTINCT complementary walkers
    const int ia = (int)(((uint64_t)r.x * (uint64_t)nc) >> 32);
    int ib = (int)(((uint64_t)r.y * (uint64_t)(nc - 1)) >> 32);
    ib += (ib >= ia);
    if (d < ndim) DIR[idx] = mu[0] * (cc[(size_t)C[ia] * ldcc + d] - cc[(size_t)C[ib] * ldcc + d]);
    else DIR[idx] = 0.f;
    if (d == 0) {
        Z0[k] = logp[wk] + logf(u01(r.z));           // log of a uniform height under the density
        const float l = -u01(r.w);
        L[k] = l; R[k] = l + 1.f;
        int J, K;
        slice_budget(seed, (uint32_t)wk, (uint32_t)step_dev[0], (uint32_t)stream_id, maxsteps, J, K);
        flags[3 * k] = J; flags[3 * k + 1] = K; flags[3 * k + 2] = 1;
    }
}

// Q[j*ns + k][:] = X[S[k]][:] + w[j*ns + k] * DIR[k][:]   for j < nrep
__global__ void slice_points_kernel(const float* __restrict__ coords, int ldc, int ndim, const int* __restrict__ S, int ns,
                                    const float* __restrict__ DIR, int ldd, const float* __restrict__ w,
                                    float* __restrict__ Q, int ldq, int nrep) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)nrep * ns * ldq) return;
    const int row = (int)(idx / ldq), d = (int)(idx % ldq), k = row % ns;
    Q[idx] = d < ndim ? coords[(size_t)S[k] * ldc + d] + w[row] * DIR[(size_t)k * ldd + d] : 0.f;
}

// stepping out: while the density at an end is above the slice, push that end out by one unit
__global__ void slice_expand_kernel(const float* __restrict__ Z0, const float* __restrict__ ZL, const float* __restrict__ ZR,
                                    float* __restrict__ L, float* __restrict__ R, int* __restrict__ flags, int ns,
                                    int* __restrict__ counters, int slot) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ns) return;
    int n = 0;
    int fl = flags[3 * k], fr = flags[3 * k + 1];                           // steps left of the budget; 0: that side is closed
    if (fl) { const int s = slice_side_steps(fl, ZL[k] > Z0[k] ? 1 : 0, 1); if (s) L[k] -= 1.f; n += s; flags[3 * k] = fl; }
    if (fr) { const int s = slice_side_steps(fr, ZR[k] > Z0[k] ? 1 : 0, 1); if (s) R[k] += 1.f; n += s; flags[3 * k + 1] = fr; }
    if (n) atomicAdd(counters + 0, n);                                      // [0] expansions
    if (fl | fr) atomicAdd(counters + slot, 1);                             // [slot] still-active count
}

// ntrial trials per launch, drawn as the SEQUENTIAL procedure would draw them if every earlier one
// were rejected: the bracket after a rejection depends on where the trial fell, not on its density,
// so trial j+1 can be placed before trial j has been evaluated.  W[j*ns + k]; Philox sub-counter
// round + j + 1 (the stream of single-trial rounds).
__global__ void slice_draw_kernel(const float* __restrict__ L, const float* __restrict__ R, const int* __restrict__ S,
                                  float* __restrict__ W, const int* __restrict__ flags, int ns, uint64_t seed,
                                  const int* __restrict__ step_dev, int stream_id, int round, int ntrial) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ns || !flags[3 * k + 2]) return;
    float l = L[k], r = R[k];
    for (int j = 0; j < ntrial; ++j) {
        const U4 b = walker_bits(seed, (uint32_t)S[k], (uint32_t)step_dev[0], (uint32_t)stream_id, (uint32_t)(round + j + 1));
        const float w = l + u01(b.x) * (r - l);
        W[(size_t)j * ns + k] = w;
        if (w < 0.f) l = w; else r = w;
    }
}

// shrinking: accept the first trial inside the slice, otherwise pull the bracket in to each rejected trial
__global__ void slice_shrink_kernel(const float* __restrict__ Z0, const float* __restrict__ Zt, float* __restrict__ L,
                                    float* __restrict__ R, const float* __restrict__ W, int* __restrict__ flags,
                                    float* __restrict__ Wacc, float* __restrict__ Zacc, int ns, int* __restrict__ counters,
                                    int slot, int ntrial) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ns || !flags[3 * k + 2]) return;
    int ncon = 0;
    bool active = true;
    for (int j = 0; j < ntrial && active; ++j) {
        const float zt = Zt[(size_t)j * ns + k], w = W[(size_t)j * ns + k];
        if (!(Z0[k] < zt)) {                                 // zeus accepts iff Z0 < lnP(x'); NaN rejects
            if (w < 0.f) L[k] = w; else R[k] = w;
            ++ncon;
            if (R[k] - L[k] < 1e-30f) { active = false; Wacc[k] = 0.f; Zacc[k] = Z0[k]; }   // degenerate: stay put
        } else {
            active = false; Wacc[k] = w; Zacc[k] = zt;
        }
    }
    if (ncon) atomicAdd(counters + 1, ncon);              // [1] contractions
    if (active) atomicAdd(counters + slot, 1);            // [slot] still-active count
    else flags[3 * k + 2] = 0;
}

__global__ void slice_commit_kernel(float* __restrict__ coords, int ldc, int ndim, float* __restrict__ logp,
                                    const int* __restrict__ S, int ns, const float* __restrict__ DIR, int ldd,
                                    const float* __restrict__ Wacc, const float* __restrict__ Zacc) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)ns * ndim) return;
    const int k = (int)(idx / ndim), d = (int)(idx % ndim);
    const int wk = S[k];
    if (Wacc[k] != 0.f) {
        coords[(size_t)wk * ldc + d] += Wacc[k] * DIR[(size_t)k * ldd + d];
        if (d == 0) logp[wk] = Zacc[k];
    }
}

// ---- one-call half step (linna_slice_half_step): the same procedure with SPECULATIVE rounds, so that a half step is a
// fixed sequence of launches the host enqueues in one call and never waits for.  Stepping out: a round evaluates the
// bracket ends the sequential loop would visit next, L, L-1, ..., L-(m-1) and R, R+1, ..., R+(m-1), in ONE launch, and
// the logic below walks them in the loop's order; shrinking: `ntrial` trials per round, each placed as if its
// predecessors were rejected (slice_draw_kernel's rule).  Same Philox counters, same comparisons, same accepted point
// as the one-point-per-round procedure -- only the count of evaluated-and-discarded points differs.  Rounds after the
// one that finishes the last walker are gated off on the device (the evaluation leaves at once on a zero count, the logic
// kernels return per walker on its flags).
// counters: [0] expansions, [1] contractions, [2] walkers left unfinished by the rounds of a call (sticky),
//           [3] evaluated points, [4 + r] walkers still active after round r (expand rounds first, then shrink rounds)
__global__ void slice_begin_kernel(const float* __restrict__ logp, const int* __restrict__ S, int ns,
                                   const float* __restrict__ cc, int ldcc, const int* __restrict__ C, int nc, int ndim,
                                   const float* __restrict__ mu, uint64_t seed, const int* __restrict__ step_dev,
                                   int stream_id, float* __restrict__ DIR, int ldd, float* __restrict__ Z0,
                                   float* __restrict__ L, float* __restrict__ R, int* __restrict__ flags,
                                   float* __restrict__ W, int m, int* __restrict__ counters, int nslots, int zero_totals,
                                   int maxsteps) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx == 0) {
        for (int i = 0; i < nslots; ++i) {             // [4 + nslots + i]: the same counts summed over the calls so far (usage statistics)
            counters[4 + nslots + i] += counters[4 + i];
            counters[4 + i] = 0;
        }
        counters[4 + 2 * nslots] += 1;                 // calls
        if (zero_totals) { counters[0] = 0; counters[1] = 0; }
    }
    if (idx >= (size_t)ns * ldd) return;
    const int k = (int)(idx / ldd), d = (int)(idx % ldd);
    const int wk = S[k];
    const U4 r = walker_bits(seed, (uint32_t)wk, (uint32_t)step_dev[0], (uint32_t)stream_id, 0u);
    const int ia = (int)(((uint64_t)r.x * (uint64_t)nc) >> 32);
    int ib = (int)(((uint64_t)r.y * (uint64_t)(nc - 1)) >> 32);
    ib += (ib >= ia);
    if (d < ndim) DIR[idx] = mu[0] * (cc[(size_t)C[ia] * ldcc + d] - cc[(size_t)C[ib] * ldcc + d]);
    else DIR[idx] = 0.f;
    if (d == 0) {
        Z0[k] = logp[wk] + logf(u01(r.z));
        const float l = -u01(r.w);
        L[k] = l; R[k] = l + 1.f;
        int J, K;
        slice_budget(seed, (uint32_t)wk, (uint32_t)step_dev[0], (uint32_t)stream_id, maxsteps, J, K);
        flags[3 * k] = J; flags[3 * k + 1] = K; flags[3 * k + 2] = 1;
        for (int j = 0; j < m; ++j) { W[(size_t)j * ns + k] = l - (float)j; W[(size_t)(m + j) * ns + k] = l + 1.f + (float)j; }
    }
}

// Zt[j*ns + k]: lnP at L - j (j < m) and at R + (j - m) (m <= j < 2m) of the bracket this round started from; the next
// round looks at m_next ends per side (0: this is the last stepping-out round)
__device__ __forceinline__ int slice_expand_multi_wave(const float* __restrict__ Z0, const float* __restrict__ Zt, float* __restrict__ L,
                                                       float* __restrict__ R, const int* __restrict__ S, int* __restrict__ flags, int ns,
                                                       int m, int m_next, int* __restrict__ counters, int slot, int prev_slot, float* __restrict__ W,
                                                       float* __restrict__ Wd, int* __restrict__ list, uint64_t seed,
                                                       const int* __restrict__ step_dev, int stream_id_shrink, int ntrial, int k, int lane) {
    if (k == 0 && lane == 0) atomicAdd(counters + 3, 2 * m * (prev_slot < 0 ? ns : counters[prev_slot]));   // the points this round evaluated
    if (k >= ns) return 0;
    int fl = flags[3 * k], fr = flags[3 * k + 1];
    if (!(fl | fr)) return 0;
    if (prev_slot >= 0 && counters[prev_slot] == 0) return 0;     // (never: a walker with a flag set was counted)
    const float z0 = Z0[k];
    float l = L[k], r = R[k];
    if (m > 32 || m_next > 32 || ntrial > 64) {                     // (schedules beyond a wave's lanes: one lane, the plain procedure)
        if (lane) return 0;
        const bool out = slice_expand_walker(k, ns, m, z0, Zt, l, r, flags, counters);        // (counts itself)
        L[k] = l; R[k] = r;
        if (out) {
            const int pos = atomicAdd(counters + slot, 1);
            for (int j = 0; j < m_next; ++j) { W[(size_t)j * ns + k] = l - (float)j; W[(size_t)(m_next + j) * ns + k] = r + (float)j; }
            for (int j = 0; j < 2 * m_next; ++j) list[(size_t)pos * 2 * m_next + j] = j * ns + k;
        } else {
            slice_draw_dev(k, S[k], l, r, Wd, ns, seed, (uint32_t)step_dev[0], stream_id_shrink, 0, ntrial);
        }
        return 0;
    }
    int nexp = 0;
    slice_expand_wave(lane, k, ns, m, z0, Zt, l, r, fl, fr, flags, nexp);
    if (lane == 0) { L[k] = l; R[k] = r; }
    if (fl | fr) {
        int pos = 0;
        if (lane == 0) pos = atomicAdd(counters + slot, 1);      // rank among the walkers still stepping out: its rows of the next launch
        pos = __builtin_amdgcn_readfirstlane(pos);
        const int side = lane >> 5, j = lane & 31;
        if (j < m_next) W[(size_t)(side * m_next + j) * ns + k] = side ? r + (float)j : l - (float)j;
        if (lane < 2 * m_next) list[(size_t)pos * 2 * m_next + lane] = lane * ns + k;
    } else {
        const float w = slice_draw_wave(lane, S[k], l, r, seed, (uint32_t)step_dev[0], stream_id_shrink, 0, ntrial);   // first shrink round's trials
        if (lane < ntrial) Wd[(size_t)lane * ns + k] = w;
    }
    return nexp;
}
__global__ void slice_expand_multi_kernel(const float* __restrict__ Z0, const float* __restrict__ Zt, float* __restrict__ L,
                                          float* __restrict__ R, const int* __restrict__ S, int* __restrict__ flags, int ns,
                                          int m, int m_next, int* __restrict__ counters, int slot, int prev_slot, float* __restrict__ W,
                                          float* __restrict__ Wd, int* __restrict__ list, uint64_t seed,
                                          const int* __restrict__ step_dev, int stream_id_shrink, int ntrial) {
    __shared__ int sums[1];
    const int k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;      // a wave per walker
    const int nexp = slice_expand_multi_wave(Z0, Zt, L, R, S, flags, ns, m, m_next, counters, slot, prev_slot, W, Wd, list, seed, step_dev,
                                             stream_id_shrink, ntrial, k, lane);
    block_add_counter(counters + 0, nexp, sums);
}

// one shrinking round (SliceRound / slice_round_walker in common.h); with `coords` the call's LAST one: the move of every
// finished walker is applied and `bump` advances the device step counter -- two launches less per iteration
__global__ void slice_shrink_multi_kernel(const SliceRound a) {
    __shared__ int sums[2];
    int nexp = 0, ncon = 0;
    slice_round_wave(a, blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), threadIdx.x & 63, nexp, ncon);     // a wave per walker
    block_add_counter(a.counters + 0, nexp, sums);
    block_add_counter(a.counters + 1, ncon, sums + 1);
}

// the move of every finished walker; a walker the rounds of the call left unfinished stays where it is and is counted
__global__ void slice_commit_checked_kernel(float* __restrict__ coords, int ldc, int ndim, float* __restrict__ logp,
                                            const int* __restrict__ S, int ns, const float* __restrict__ DIR, int ldd,
                                            const float* __restrict__ Wacc, const float* __restrict__ Zacc,
                                            const int* __restrict__ flags, int* __restrict__ counters) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)ns * ndim) return;
    const int k = (int)(idx / ndim), d = (int)(idx % ndim);
    if (flags[3 * k] | flags[3 * k + 1] | flags[3 * k + 2]) {
        if (d == 0) atomicAdd(counters + 2, 1);
        return;
    }
    const int wk = S[k];
    if (Wacc[k] != 0.f) {
        coords[(size_t)wk * ldc + d] += Wacc[k] * DIR[(size_t)k * ldd + d];
        if (d == 0) logp[wk] = Zacc[k];
    }
}

// ------------------------------------------------------------------ host-side launchers (namespace-internal)
#define LAUNCH_CHECK(name) return check_hip(hipGetLastError(), name)

int launch_prior_map_fwd(const float* Z, int ldz, int B, int nin, const int* is_flat, const float* a1, const float* a2,
                         const int* lg, const float* xmean, const float* xstd, float* X, int ldx, float* TH, int ldt,
                         hipStream_t s) {
    hipLaunchKernelGGL(prior_map_fwd_kernel, grid1d((size_t)B * ldx, 256), dim3(256), 0, s, Z, ldz, B, nin, is_flat, a1,
                       a2, lg, xmean, xstd, X, ldx, TH, ldt);
    LAUNCH_CHECK("prior_map_fwd");
}
int launch_prior_map_bwd(const float* Z, int ldz, int B, int nin, const int* is_flat, const float* a1, const float* a2,
                         const int* lg, const float* xstd, const float* dX, int lddx, float* dZ, int lddz, hipStream_t s) {
    hipLaunchKernelGGL(prior_map_bwd_kernel, grid1d((size_t)B * nin, 256), dim3(256), 0, s, Z, ldz, B, nin, is_flat, a1,
                       a2, lg, xstd, dX, lddx, dZ, lddz);
    LAUNCH_CHECK("prior_map_bwd");
}
template <int LPR>
static void launch_loglike_diag_lpr(const float* D, int ldd, int B, int nout, const float* w, const float* Z, int ldz, int nin,
                                    float T, float* out, hipStream_t s) {
    constexpr int rows_per_block = 4 * (64 / LPR);
    hipLaunchKernelGGL(loglike_diag_kernel<LPR>, dim3((B + rows_per_block - 1) / rows_per_block), dim3(256), 0, s, D, ldd, B,
                       nout, w, Z, ldz, nin, T, out);
}
int launch_loglike_diag(const float* D, int ldd, int B, int nout, const float* w, const float* Z, int ldz, int nin,
                        float T, float* out, hipStream_t s) {
    const int q = ((nout > nin ? nout : nin) + 3) / 4;            // 16-byte chunks of the longer of the two rows
    if (q > 32) launch_loglike_diag_lpr<64>(D, ldd, B, nout, w, Z, ldz, nin, T, out, s);
    else if (q > 16) launch_loglike_diag_lpr<32>(D, ldd, B, nout, w, Z, ldz, nin, T, out, s);
    else if (q > 8) launch_loglike_diag_lpr<16>(D, ldd, B, nout, w, Z, ldz, nin, T, out, s);
    else if (q > 4) launch_loglike_diag_lpr<8>(D, ldd, B, nout, w, Z, ldz, nin, T, out, s);
    else launch_loglike_diag_lpr<4>(D, ldd, B, nout, w, Z, ldz, nin, T, out, s);
    LAUNCH_CHECK("loglike_diag");
}
int launch_loglike_finish(const float* partial, int slots_ld, int nslots, int B, const float* Z, int ldz, int nin,
                          float T, float* out, hipStream_t s) {
    hipLaunchKernelGGL(loglike_finish_kernel, dim3((B + 3) / 4), dim3(256), 0, s, partial, slots_ld, nslots, B, Z, ldz,
                       nin, T, out);
    LAUNCH_CHECK("loglike_finish");
}
int launch_loglike_diag_grad(const float* D, int ldd, int B, int nout, const float* w, const float* gscale, float T,
                             float* dH, int lddh, hipStream_t s) {
    hipLaunchKernelGGL(loglike_diag_grad_kernel, grid1d((size_t)B * lddh, 256), dim3(256), 0, s, D, ldd, B, nout, w,
                       gscale, T, dH, lddh);
    LAUNCH_CHECK("loglike_diag_grad");
}
int launch_loss_delta(int mode, const float* PRED, int ldp, const float* Y, int ldy, const int* ROWS, int B,
                      const linna_loss_desc_t& d, float* DELTA, int ldd, hipStream_t s) {
    hipLaunchKernelGGL(loss_delta_kernel, grid1d((size_t)B * ldd, 256), dim3(256), 0, s, mode, PRED, ldp, Y, ldy, ROWS,
                       B, d.nout, d.sigma, d.ymean, d.ystd, d.ylog, d.data_norm, DELTA, ldd);
    LAUNCH_CHECK("loss_delta");
}
int launch_loss_targets(const float* Y, int ldy, int n, const linna_loss_desc_t& d, float* YN, int ldyn, hipStream_t s) {
    hipLaunchKernelGGL(loss_targets_kernel, grid1d((size_t)n * ldyn, 256), dim3(256), 0, s, Y, ldy, n, d.nout, d.sigma, d.ymean,
                       d.ystd, d.ylog, d.data_norm, YN, ldyn);
    LAUNCH_CHECK("loss_targets");
}
int launch_loss_fused_small(const float* PRED, int ldp, const float* Y, int ldy, const int* ROWS, int B,
                            const linna_loss_desc_t& d, const float* den, float inv_batch, float* loss_rows, float* loss_mean,
                            float* dP, int lddp, unsigned* counter, hipStream_t s) {
    hipLaunchKernelGGL(loss_fused_small_kernel, dim3((B + 3) / 4), dim3(256), 0, s, PRED, ldp, Y, ldy, ROWS, B, d.nout, d.sigma,
                       d.ymean, d.ystd, d.ylog, d.data_norm, d.Cinv, d.ldc, den, inv_batch, loss_rows, loss_mean, dP, lddp, counter);
    LAUNCH_CHECK("loss_fused_small");
}
int launch_loss_rows(int mode, const float* partial, int slots_ld, int nslots, int B, const float* den, const int* ROWS,
                     float floorv, float* out, hipStream_t s) {
    hipLaunchKernelGGL(loss_rows_kernel, grid1d(B, 256), dim3(256), 0, s, mode, partial, slots_ld, nslots, B, den, ROWS,
                       floorv, out);
    LAUNCH_CHECK("loss_rows");
}
int launch_loss_grad(const float* U, int ldu, const float* Y, int ldy, const int* ROWS, int B, int nout,
                     const float* data_norm, const float* den, float inv_batch, float* dP, int lddp, hipStream_t s) {
    hipLaunchKernelGGL(loss_grad_kernel, grid1d((size_t)B * lddp, 256), dim3(256), 0, s, U, ldu, Y, ldy, ROWS, B, nout,
                       data_norm, den, inv_batch, dP, lddp);
    LAUNCH_CHECK("loss_grad");
}
int launch_sum_scale(const float* v, int n, float scale, float* out, hipStream_t s) {
    hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(1024), 0, s, v, n, scale, out);
    LAUNCH_CHECK("sum_scale");
}
int launch_sum_scale_prepare(const float* v, int n, float scale, float* out, int* step_dev, float* hyper, float b1, float b2,
                             hipStream_t s) {
    hipLaunchKernelGGL(sum_scale_prepare_kernel, dim3(1), dim3(1024), 0, s, v, n, scale, out, step_dev, hyper, b1, b2);
    LAUNCH_CHECK("sum_scale_prepare");
}
int launch_val_frac(const float* partial, int slots_ld, int nslots, int B, const float* den, float* frac, hipStream_t s) {
    hipLaunchKernelGGL(val_frac_kernel, grid1d(B, 256), dim3(256), 0, s, partial, slots_ld, nslots, B, den, frac);
    LAUNCH_CHECK("val_frac");
}
int launch_val_metrics(const float* loss, const float* frac, int n, const float* last, float* out, hipStream_t s) {
    hipLaunchKernelGGL(val_metrics_kernel, grid1d(n, 256), dim3(256), 0, s, loss, frac, n, last, out);
    LAUNCH_CHECK("val_metrics");
}
int launch_gather_xform(const float* X, int ldx, const int* ROWS, int B, int nin, const int* lg, const float* xmean,
                        const float* xstd, float* XB, int ldxb, hipStream_t s) {
    hipLaunchKernelGGL(gather_xform_kernel, grid1d((size_t)B * ldxb, 256), dim3(256), 0, s, X, ldx, ROWS, B, nin, lg,
                       xmean, xstd, XB, ldxb);
    LAUNCH_CHECK("gather_xform");
}
int launch_colsum(const float* dZ, int ld, int B, int N, float scale, float* db, hipStream_t s) {
    hipLaunchKernelGGL(colsum_kernel, dim3((N + 63) / 64), dim3(1024), 0, s, dZ, ld, B, N, scale, db);
    LAUNCH_CHECK("colsum");
}
int launch_adamw(float* p, const float* g, float* m, float* v, size_t n, float* hyper, int* step_dev, float b1, float b2,
                 float eps, hipStream_t s) {
    if (step_dev) hipLaunchKernelGGL(adamw_prepare_kernel, dim3(1), dim3(1), 0, s, step_dev, hyper, b1, b2);   // null: already advanced
    hipLaunchKernelGGL(adamw_kernel, grid1d(n, 256), dim3(256), 0, s, p, g, m, v, n, hyper, b1, b2, eps);
    LAUNCH_CHECK("adamw");
}
int launch_adamw_prepare(float* hyper, int* step_dev, float b1, float b2, hipStream_t s) {
    hipLaunchKernelGGL(adamw_prepare_kernel, dim3(1), dim3(1), 0, s, step_dev, hyper, b1, b2);
    LAUNCH_CHECK("adamw_prepare");
}
int launch_stretch_propose(const float* coords, int ldc, int ndim, const int* S, int ns, const float* ccoords, int ldcc,
                           const int* C, int nc, uint64_t seed, const int* step_dev, int stream_id, float a, float* Q,
                           int ldq, float* factors, hipStream_t s) {
    hipLaunchKernelGGL(stretch_propose_kernel, grid1d((size_t)ns * ldq, 256), dim3(256), 0, s, coords, ldc, ndim, S, ns,
                       ccoords, ldcc, C, nc, seed, step_dev, stream_id, a, Q, ldq, factors);
    LAUNCH_CHECK("stretch_propose");
}
int launch_stretch_accept(float* coords, int ldc, int ndim, float* logp, const int* S, int ns, const float* Q, int ldq,
                          const float* lp_new, const float* factors, uint64_t seed, const int* step_dev, int stream_id,
                          int* naccept, hipStream_t s) {
    hipLaunchKernelGGL(stretch_accept_kernel, grid1d(ns, 256), dim3(256), 0, s, coords, ldc, ndim, logp, S, ns, Q, ldq,
                       lp_new, factors, seed, step_dev, stream_id, naccept);
    LAUNCH_CHECK("stretch_accept");
}
int launch_hmc_init(int B, int ndim, const float* mass, uint64_t seed, const int* step_dev, const float* lnp,
                    const float* P0, int ldp0, float* P, int ldp, float* H0, hipStream_t s) {
    hipLaunchKernelGGL(hmc_start_kernel, dim3((B + 3) / 4), dim3(256), 0, s, B, ndim, mass, seed, step_dev, lnp, P0, ldp0,
                       (const float*)nullptr, 0, 0.f, 0.f, (const float*)nullptr, 0, P, ldp, (float*)nullptr, 0, H0);
    LAUNCH_CHECK("hmc_init");
}
int launch_hmc_start(int B, int ndim, const float* mass, uint64_t seed, const int* step_dev, const float* lnp, const float* P0,
                     int ldp0, const float* G, int ldg, float ek, float ed, const float* X, int ldx, float* P, int ldp, float* Q,
                     int ldq, float* H0, hipStream_t s) {
    hipLaunchKernelGGL(hmc_start_kernel, dim3((B + 3) / 4), dim3(256), 0, s, B, ndim, mass, seed, step_dev, lnp, P0, ldp0, G, ldg, ek,
                       ed, X, ldx, P, ldp, Q, ldq, H0);
    LAUNCH_CHECK("hmc_start");
}
int launch_hmc_kick_drift(int B, int ndim, const float* mass, float ek, float ed, const float* G, int ldg, float* P,
                          int ldp, float* Q, int ldq, hipStream_t s) {
    hipLaunchKernelGGL(hmc_kick_drift_kernel, grid1d((size_t)B * ndim, 256), dim3(256), 0, s, B, ndim, mass, ek, ed, G,
                       ldg, P, ldp, Q, ldq);
    LAUNCH_CHECK("hmc_kick_drift");
}
int launch_hmc_accept(int B, int ndim, const float* mass, uint64_t seed, const int* step_dev, const float* H0,
                      const float* P, int ldp, const float* Qn, int ldq, const float* lnp_new, const float* Gn, int ldg,
                      const float* U, float* X, int ldx, float* lnp, float* G, int* naccept, hipStream_t s) {
    hipLaunchKernelGGL(hmc_accept_kernel, dim3((B + 3) / 4), dim3(256), 0, s, B, ndim, mass, seed, step_dev, H0, P, ldp, Qn,
                       ldq, lnp_new, Gn, ldg, U, X, ldx, lnp, G, naccept);
    LAUNCH_CHECK("hmc_accept");
}
int launch_slice_init(const float* logp, const int* S, int ns, const float* cc, int ldcc, const int* C, int nc, int ndim,
                      const float* mu, uint64_t seed, const int* step_dev, int stream_id, float* DIR, int ldd, float* Z0,
                      float* L, float* R, int* flags, int maxsteps, hipStream_t s) {
    hipLaunchKernelGGL(slice_init_kernel, grid1d((size_t)ns * ldd, 256), dim3(256), 0, s, logp, S, ns, cc, ldcc, C, nc, ndim,
                       mu, seed, step_dev, stream_id, DIR, ldd, Z0, L, R, flags, maxsteps);
    LAUNCH_CHECK("slice_init");
}
int launch_slice_points(const float* coords, int ldc, int ndim, const int* S, int ns, const float* DIR, int ldd,
                        const float* w, float* Q, int ldq, int nrep, hipStream_t s) {
    hipLaunchKernelGGL(slice_points_kernel, grid1d((size_t)nrep * ns * ldq, 256), dim3(256), 0, s, coords, ldc, ndim, S, ns, DIR,
                       ldd, w, Q, ldq, nrep);
    LAUNCH_CHECK("slice_points");
}
int launch_slice_expand(const float* Z0, const float* ZL, const float* ZR, float* L, float* R, int* flags, int ns,
                        int* counters, int slot, hipStream_t s) {
    hipLaunchKernelGGL(slice_expand_kernel, grid1d(ns, 256), dim3(256), 0, s, Z0, ZL, ZR, L, R, flags, ns, counters, slot);
    LAUNCH_CHECK("slice_expand");
}
int launch_slice_draw(const float* L, const float* R, const int* S, float* W, const int* flags, int ns, uint64_t seed,
                      const int* step_dev, int stream_id, int round, int ntrial, hipStream_t s) {
    hipLaunchKernelGGL(slice_draw_kernel, grid1d(ns, 256), dim3(256), 0, s, L, R, S, W, flags, ns, seed, step_dev,
                       stream_id, round, ntrial);
    LAUNCH_CHECK("slice_draw");
}
int launch_slice_shrink(const float* Z0, const float* Zt, float* L, float* R, const float* W, int* flags, float* Wacc,
                        float* Zacc, int ns, int* counters, int slot, int ntrial, hipStream_t s) {
    hipLaunchKernelGGL(slice_shrink_kernel, grid1d(ns, 256), dim3(256), 0, s, Z0, Zt, L, R, W, flags, Wacc, Zacc, ns,
                       counters, slot, ntrial);
    LAUNCH_CHECK("slice_shrink");
}
int launch_slice_begin(const float* logp, const int* S, int ns, const float* cc, int ldcc, const int* C, int nc, int ndim,
                       const float* mu, uint64_t seed, const int* step_dev, int stream_id, float* DIR, int ldd, float* Z0, float* L,
                       float* R, int* flags, float* W, int m, int* counters, int nslots, int zero_totals, int maxsteps, hipStream_t s) {
    hipLaunchKernelGGL(slice_begin_kernel, grid1d((size_t)ns * ldd, 256), dim3(256), 0, s, logp, S, ns, cc, ldcc, C, nc, ndim, mu,
                       seed, step_dev, stream_id, DIR, ldd, Z0, L, R, flags, W, m, counters, nslots, zero_totals, maxsteps);
    LAUNCH_CHECK("slice_begin");
}
int launch_slice_expand_multi(const float* Z0, const float* Zt, float* L, float* R, const int* S, int* flags, int ns, int m,
                              int m_next, int* counters, int slot, int prev_slot, float* W, float* Wd, int* list, uint64_t seed,
                              const int* step_dev, int stream_id_shrink, int ntrial, hipStream_t s) {
    hipLaunchKernelGGL(slice_expand_multi_kernel, dim3((ns + 15) / 16), dim3(1024), 0, s, Z0, Zt, L, R, S, flags, ns, m, m_next, counters, slot,
                       prev_slot, W, Wd, list, seed, step_dev, stream_id_shrink, ntrial);
    LAUNCH_CHECK("slice_expand_multi");
}
int launch_slice_shrink_multi(const SliceRound& a, hipStream_t s) {
    hipLaunchKernelGGL(slice_shrink_multi_kernel, dim3((a.ns + 15) / 16), dim3(1024), 0, s, a);
    LAUNCH_CHECK("slice_shrink_multi");
}
int launch_slice_commit_checked(float* coords, int ldc, int ndim, float* logp, const int* S, int ns, const float* DIR, int ldd,
                                const float* Wacc, const float* Zacc, const int* flags, int* counters, hipStream_t s) {
    hipLaunchKernelGGL(slice_commit_checked_kernel, grid1d((size_t)ns * ndim, 256), dim3(256), 0, s, coords, ldc, ndim, logp, S, ns,
                       DIR, ldd, Wacc, Zacc, flags, counters);
    LAUNCH_CHECK("slice_commit_checked");
}
int launch_slice_commit(float* coords, int ldc, int ndim, float* logp, const int* S, int ns, const float* DIR, int ldd,
                        const float* Wacc, const float* Zacc, hipStream_t s) {
    hipLaunchKernelGGL(slice_commit_kernel, grid1d((size_t)ns * ndim, 256), dim3(256), 0, s, coords, ldc, ndim, logp, S,
                       ns, DIR, ldd, Wacc, Zacc);
    LAUNCH_CHECK("slice_commit");
}
int launch_step_increment(int* step, hipStream_t s) {
    hipLaunchKernelGGL(step_increment_kernel, dim3(1), dim3(1), 0, s, step);
    LAUNCH_CHECK("step_increment");
}

}  // namespace linna
