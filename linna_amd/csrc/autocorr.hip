// Convergence statistics of a walker chain on gfx950, incrementally.
//
// The reference asks emcee for the integrated autocorrelation time of the WHOLE chain every 100 iterations
// (linna/sampler.py:538 `get_autocorr_time(tol=0)`; zeus callback :667-696 with the first 20 % discarded) and compares the
// two halves of the chain's tail (`checkmeanstd`, :370-387).  emcee's estimator: per (walker, parameter) series the
// autocovariance of the mean-removed series, normalised by its lag-0 value, averaged over the walkers; tau_M = 2 sum_{k<=M} f_k - 1
// at the first M with M >= c tau_M (Sokal's window, c = 5).  Only lags up to that window enter the result, so instead of an
// FFT over the whole chain per check this file keeps, per series, the RUNNING lagged products
//     S_k = sum_{t = lo+k}^{hi-1} x_t x_{t-k},  k = 0..K        and        T = sum_{t = lo}^{hi-1} x_t
// of the window [lo, hi) of the chain in float64 and updates them from the rows that entered (the 100 new steps) or left
// (zeus' moving discard) since the last check: O(rows x K) per series instead of O(n log n).  With the series mean m = T / N,
// N = hi - lo,
//     sum_{t=lo}^{hi-k-1} (x_t - m)(x_{t+k} - m) = S_k - m (2 T - tail_k - head_k) + (N - k) m^2
// (head_k / tail_k: the sums of the first / last k rows of the window) is exactly what the FFT yields.  x is taken relative to
// the series' first stored row (the autocovariance does not see a constant shift), which keeps the cancellation in the
// formula above at the level of (drift / sigma)^2 instead of (mean / sigma)^2.
//
// Layout: the statistics' copy of the chain is time x parameter x walker, CT[row][d][w] fp32 with the walker count padded to a
// multiple of 64 (padding = 0): a wavefront's lanes are 64 walkers of ONE parameter, every load of a chain row is one coalesced
// 256-byte line, the walker average of emcee's estimator is a wavefront reduction.  Sums are S[k][d][w] float64.
//
// Bound: the update is balanced between the float64 vector pipe (one v_fma_f64 per (anchor row, lag, series): MI355X has no
// faster float64 matrix rate than vector rate) and memory (each wave re-reads a window of TT + R - 1 chain rows per TT x R
// products from L2); everything else here is one pass over K x series values.
#include "common.h"
#include <math.h>

namespace linna {

constexpr int AC_R = 32;        // lags per wavefront (accumulators per lane)
constexpr int AC_TT = 16;       // anchor rows per tile
constexpr int AC_WAVES = 4;     // wavefronts (lag blocks) per workgroup
constexpr int AC_CK = 32;       // lags per chunk of the finalising kernels

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ------------------------------------------------------------------ transposing append
// block[nsteps][nw][ldb] (the sampler's chain block, walker-major) -> CT[row0 + i][d][lane(w)].  Every wstride-th walker --
// the subset the routine checks average over -- takes the first nws lanes of a parameter, the others follow in order:
// lane(w) = w / wstride if w % wstride == 0, else nws + w - w / wstride - 1.
__global__ __launch_bounds__(256) void chain_append_t_kernel(const float* __restrict__ block, int ldb, int nw, int ndim,
                                                             int wstride, int nws, float* __restrict__ CT, int nwp, int64_t row0) {
    extern __shared__ float tile[];                  // [64][ndim + 1]
    const int i = blockIdx.y, w0 = blockIdx.x * 64;
    const int ldt = ndim + 1;
    const float* src = block + (size_t)i * nw * ldb;
    for (int e = threadIdx.x; e < 64 * ndim; e += 256) {
        const int wl = e / ndim, d = e - wl * ndim;
        const int lane = w0 + wl;                    // destination lane -> source walker
        int w = -1;
        if (lane < nws) w = lane * wstride;
        else if (lane < nw && wstride > 1) { const int m = lane - nws; w = m + m / (wstride - 1) + 1; }     // the m-th walker that is not a multiple of wstride
        tile[wl * ldt + d] = w >= 0 ? src[(size_t)w * ldb + d] : 0.f;
    }
    __syncthreads();
    float* dst = CT + (size_t)(row0 + i) * ndim * nwp;
    for (int e = threadIdx.x; e < 64 * ndim; e += 256) {
        const int d = e >> 6, wl = e & 63;
        dst[(size_t)d * nwp + w0 + wl] = tile[wl * ldt + d];
    }
}

// ------------------------------------------------------------------ running lagged products
// One wavefront = 64 series (walkers of one parameter) x AC_R lags; it walks the anchor rows in tiles of AC_TT with the
// partner rows of its lags in a SLIDING register window: of the AC_TT + AC_R - 1 partner rows a tile multiplies, all but
// AC_TT were the previous tile's, so a tile costs 2 AC_TT row loads (its anchors and its new partners) for AC_TT x AC_R
// fused multiply-adds; the next tile's rows are requested before this tile's arithmetic.
// DIR = -1 (rows entered at the end): anchors t in [a0, a1), partner t - k, valid while >= lo.
// DIR = +1 (rows left at the front):  anchors s in [a0, a1), partner s + k, valid while < hi; the caller passes sign = -1.
// Series index s in [0, nd * nwc): parameter d = s / nwc, lane w = s % nwc of CT[row][d][nwp]; sums S[k][nd * nwc].
template <int DIR>
__global__ __launch_bounds__(64 * AC_WAVES) void acorr_update_kernel(const float* __restrict__ CT, int nd, int nwp, int nwc,
                                                                     int a0, int a1, int lo, int hi, int k0, int k1,
                                                                     double* __restrict__ S, double* __restrict__ T, double sign) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = blockIdx.x * 64 + lane;
    const int kb = k0 + (blockIdx.y * AC_WAVES + wave) * AC_R;
    if (kb >= k1) return;                                   // (wave-uniform)
    const int d = (blockIdx.x * 64) / nwc;
    const size_t ns = (size_t)nd * nwp, nser = (size_t)nd * nwc;
    const float* col = CT + (size_t)d * nwp + (s - d * nwc);
    const double ref = (double)col[0];
    auto ldrow = [&](int r) { r = min(max(r, lo), hi - 1); return col[(size_t)r * ns]; };
    constexpr int NW = AC_TT + AC_R - 1, OLD = AC_R - 1;
    double acc[AC_R], win[NW], an[AC_TT];
#pragma unroll
    for (int r = 0; r < AC_R; ++r) acc[r] = 0.0;
    double tsum = 0.0;
    int t0 = a0, wb = DIR < 0 ? a0 - kb - OLD : a0 + kb;
#pragma unroll
    for (int j = 0; j < OLD; ++j) {
        const int r = wb + j;
        const double v = (double)ldrow(r) - ref;
        win[j] = (r >= lo && r < hi) ? v : 0.0;
    }
    float rw[AC_TT], ra[AC_TT];
#pragma unroll
    for (int i = 0; i < AC_TT; ++i) { rw[i] = ldrow(wb + OLD + i); ra[i] = ldrow(t0 + i); }
    for (; t0 < a1; t0 += AC_TT, wb += AC_TT) {
#pragma unroll
        for (int i = 0; i < AC_TT; ++i) {
            const int r = wb + OLD + i, t = t0 + i;
            win[OLD + i] = (r >= lo && r < hi) ? (double)rw[i] - ref : 0.0;
            an[i] = t < a1 ? (double)ra[i] - ref : 0.0;
            tsum += an[i];
        }
        if (t0 + AC_TT < a1) {
#pragma unroll
            for (int i = 0; i < AC_TT; ++i) { rw[i] = ldrow(wb + AC_TT + OLD + i); ra[i] = ldrow(t0 + AC_TT + i); }
        }
#pragma unroll
        for (int i = 0; i < AC_TT; ++i)
#pragma unroll
            for (int r = 0; r < AC_R; ++r)
                acc[r] = __builtin_fma(an[i], win[DIR < 0 ? i - r + OLD : i + r], acc[r]);
#pragma unroll
        for (int j = 0; j < OLD; ++j) win[j] = win[j + AC_TT];
    }
#pragma unroll
    for (int r = 0; r < AC_R; ++r) {
        double* p = S + (size_t)(kb + r) * nser + s;
        *p = *p + sign * acc[r];
    }
    if (kb == 0 && T) T[s] = T[s] + sign * tsum;
}

// ------------------------------------------------------------------ tau from the sums
// sums of AC_CK rows at the head / tail of the window: HC[c][s] = sum_{j in chunk c} x[lo + j], TC[c][s] = sum x[hi - 1 - j]
__global__ __launch_bounds__(64) void acorr_chunksum_kernel(const float* __restrict__ CT, int nd, int nwp, int nwc, int lo, int hi,
                                                            double* __restrict__ HC, double* __restrict__ TC) {
    const int s = blockIdx.x * 64 + threadIdx.x, c = blockIdx.y;
    const int d = (blockIdx.x * 64) / nwc;
    const size_t ns = (size_t)nd * nwp, nser = (size_t)nd * nwc;
    const float* col = CT + (size_t)d * nwp + (s - d * nwc);
    const double ref = (double)col[0];
    const int N = hi - lo;
    double h = 0.0, t = 0.0;
#pragma unroll 8
    for (int j = c * AC_CK; j < (c + 1) * AC_CK; ++j) {
        const bool ok = j < N;
        const double vh = (double)col[(size_t)(ok ? lo + j : lo) * ns] - ref;
        const double vt = (double)col[(size_t)(ok ? hi - 1 - j : lo) * ns] - ref;
        h += ok ? vh : 0.0;
        t += ok ? vt : 0.0;
    }
    HC[(size_t)c * nser + s] = h;
    TC[(size_t)c * nser + s] = t;
}
// in place: chunk sums -> sums of all chunks before (one thread per series walks the chunks)
__global__ __launch_bounds__(64) void acorr_chunkscan_kernel(int nser, int nchunk, double* __restrict__ HC, double* __restrict__ TC) {
    const int s = blockIdx.x * 64 + threadIdx.x;
    double h = 0.0, t = 0.0;
    for (int c = 0; c < nchunk; ++c) {
        double* ph = HC + (size_t)c * nser + s; double* pt = TC + (size_t)c * nser + s;
        const double vh = *ph, vt = *pt;
        *ph = h; *pt = t;
        h += vh; t += vt;
    }
}

// rho_k = acf_k / acf_0 per series, summed over the 64 walkers of the wavefront: P[k][group]
__global__ __launch_bounds__(64) void acorr_rho_kernel(const float* __restrict__ CT, int nd, int nwp, int nwc, int nlive, int lo, int hi,
                                                       int kuse, const double* __restrict__ S, const double* __restrict__ T,
                                                       const double* __restrict__ HC, const double* __restrict__ TC,
                                                       double* __restrict__ P) {
    const int lane = threadIdx.x, g = blockIdx.x, c = blockIdx.y;
    const int s = g * 64 + lane;
    const int d = (g * 64) / nwc, w = s - d * nwc;
    const size_t ns = (size_t)nd * nwp, nser = (size_t)nd * nwc;
    const float* col = CT + (size_t)d * nwp + w;
    const double ref = (double)col[0];
    const int N = hi - lo;
    const double Tt = T[s], m = Tt / (double)N;
    const double acf0 = S[s] - (double)N * m * m;            // = S_0 - m (2T) + N m^2
    double hp = HC[(size_t)c * nser + s], tp = TC[(size_t)c * nser + s];
    const bool live = w < nlive;
    const int ngroups = (int)(nser >> 6);
    const int kend = min((c + 1) * AC_CK, kuse + 1);
    for (int k = c * AC_CK; k < kend; ++k) {
        const double acf = S[(size_t)k * nser + s] - m * (2.0 * Tt - tp - hp) + (double)(N - k) * m * m;
        const double rho = live ? acf / acf0 : 0.0;          // 0 / 0 -> NaN, as the host estimator
        const double sum = wave_sum_f64(rho);
        if (lane == 0) P[(size_t)k * ngroups + g] = sum;
        hp += (double)col[(size_t)(lo + k) * ns] - ref;       // (k <= kuse <= N - 1: both rows exist)
        tp += (double)col[(size_t)(hi - 1 - k) * ns] - ref;
    }
}
// f[d][k] = mean over walkers = (sum of the parameter's wavefront partials) / nlive
__global__ __launch_bounds__(256) void acorr_fmean_kernel(const double* __restrict__ P, int ngroups, int gpd, int nlive, int L, int nd,
                                                          double* __restrict__ F) {
    const int k = blockIdx.x * 256 + threadIdx.x, d = blockIdx.y;
    if (k >= L) return;
    double v = 0.0;
    for (int g = 0; g < gpd; ++g) v += P[(size_t)k * ngroups + d * gpd + g];
    F[(size_t)d * L + k] = v / (double)nlive;
}

// per parameter: tau_k = 2 cumsum(f)_k - 1; emcee's auto_window
// out[d] = tau, out[nd + d] = window, out[2 nd + d] = status (0 done; 1 no window within kuse and kuse < N - 1: more lags needed)
__global__ __launch_bounds__(256) void acorr_window_kernel(const double* __restrict__ F, int kuse, int N, double cfac, int nd,
                                                           double* __restrict__ out) {
    __shared__ double part[256];
    __shared__ int first_false;
    const int d = blockIdx.x, tid = threadIdx.x;
    const int L = kuse + 1, seg = (L + 255) / 256;
    const int k0 = min(L, tid * seg), k1 = min(L, k0 + seg);
    const double* f = F + (size_t)d * L;
    double loc = 0.0;
    for (int k = k0; k < k1; ++k) loc += f[k];
    part[tid] = loc;
    if (tid == 0) first_false = L;
    __syncthreads();
    if (tid == 0) {                                         // exclusive scan of 256 partial sums
        double run = 0.0;
        for (int i = 0; i < 256; ++i) { const double v = part[i]; part[i] = run; run += v; }
    }
    __syncthreads();
    double run = part[tid];
    int mine = L;
    for (int k = k0; k < k1; ++k) {
        run += f[k];
        const double tau = 2.0 * run - 1.0;
        if (!((double)k < cfac * tau)) { mine = k; break; }
    }
    if (mine < L) atomicMin(&first_false, mine);
    __syncthreads();
    int kf = first_false;
    double status = 0.0;
    if (kf == L) {
        if (kuse >= N - 1) kf = 0;                           // every lag of the chain looked at, all true: numpy's argmin gives 0
        else { status = 1.0; kf = kuse; }
    }
    if (kf >= k0 && kf < k1) {
        double r2 = part[tid];
        for (int k = k0; k <= kf; ++k) r2 += f[k];
        double tau = 2.0 * r2 - 1.0;
        if (first_false == 0) tau = NAN;                     // m[0] false: a NaN series (np.any(m) false -> taus[-1] = NaN)
        out[d] = tau;
        out[nd + d] = (double)kf;
        out[2 * nd + d] = status;
    }
}

// ------------------------------------------------------------------ checkmeanstd's moments (sampler.py:370-387)
// rows [t0, tm) and [tm, t1) of parameter d, all walkers: mean and population standard deviation -> out[d][half][2]
__global__ __launch_bounds__(256) void chain_meanstd_kernel(const float* __restrict__ CT, int ndim, int nwp, int nws, int64_t t0,
                                                            int64_t tm, int64_t t1, double* __restrict__ out) {     // nws: live lanes (all walkers)
    __shared__ double sh[2][4];
    const int d = blockIdx.x, half = blockIdx.y;
    const int64_t r0 = half ? tm : t0, r1 = half ? t1 : tm;
    const size_t ns = (size_t)ndim * nwp;
    const float* base = CT + (size_t)d * nwp;
    const double ref = (double)base[(size_t)t0 * ns];
    double s1 = 0.0, s2 = 0.0;
    const int64_t total = (r1 - r0) * nwp;
    for (int64_t e = threadIdx.x; e < total; e += 256) {
        const int64_t t = r0 + e / nwp; const int w = (int)(e % nwp);
        if (w < nws) {
            const double v = (double)base[(size_t)t * ns + w] - ref;
            s1 += v; s2 = __builtin_fma(v, v, s2);
        }
    }
    s1 = wave_sum_f64(s1); s2 = wave_sum_f64(s2);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[0][wave] = s1; sh[1][wave] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double a = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3], b = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
        const double n = (double)(r1 - r0) * (double)nws;
        const double mu = a / n;
        double var = b / n - mu * mu;
        if (var < 0.0) var = 0.0;
        out[((size_t)d * 2 + half) * 2 + 0] = mu + ref;
        out[((size_t)d * 2 + half) * 2 + 1] = sqrt(var);
    }
}

// ------------------------------------------------------------------ launchers
int launch_chain_append_t(const float* block, int ldb, int nsteps, int nw, int ndim, int wstride, float* CT, int nwp,
                          int64_t row0, hipStream_t s) {
    const int nws = (nw + wstride - 1) / wstride;
    hipLaunchKernelGGL(chain_append_t_kernel, dim3(nwp / 64, nsteps), dim3(256), (size_t)64 * (ndim + 1) * sizeof(float), s,
                       block, ldb, nw, ndim, wstride, nws, CT, nwp, row0);
    return check_hip(hipGetLastError(), "chain_append_t");
}

int launch_acorr_update(const float* CT, int nd, int nwp, int nwc, int64_t a0, int64_t a1, int64_t lo, int64_t hi, int k0, int k1,
                        double* S, double* T, int remove, hipStream_t s) {
    if (a1 <= a0 || k1 <= k0) return LINNA_OK;
    const int nblk = (k1 - k0) / AC_R;
    const dim3 grid(nd * nwc / 64, (nblk + AC_WAVES - 1) / AC_WAVES), block(64 * AC_WAVES);
    if (remove) hipLaunchKernelGGL((acorr_update_kernel<1>), grid, block, 0, s, CT, nd, nwp, nwc, (int)a0, (int)a1, (int)lo, (int)hi, k0, k1, S, T, -1.0);
    else hipLaunchKernelGGL((acorr_update_kernel<-1>), grid, block, 0, s, CT, nd, nwp, nwc, (int)a0, (int)a1, (int)lo, (int)hi, k0, k1, S, T, 1.0);
    return check_hip(hipGetLastError(), "acorr_update");
}

size_t acorr_scratch_doubles(int nser, int kuse, int nd) {
    const size_t nchunk = (size_t)kuse / AC_CK + 1;
    return 2 * nchunk * nser + nchunk * AC_CK * (size_t)(nser / 64) + (size_t)nd * (kuse + 1);
}

int launch_acorr_tau(const float* CT, int ndim, int nwp, int nwc, int nlive, int64_t lo, int64_t hi, int kuse, const double* S,
                     const double* T, double c, double* scratch, double* out, hipStream_t s) {
    const int nser = ndim * nwc;
    const int nchunk = kuse / AC_CK + 1, L = kuse + 1;
    double* HC = scratch;
    double* TC = HC + (size_t)nchunk * nser;
    double* P = TC + (size_t)nchunk * nser;
    double* F = P + (size_t)nchunk * AC_CK * (nser / 64);
    hipLaunchKernelGGL(acorr_chunksum_kernel, dim3(nser / 64, nchunk), dim3(64), 0, s, CT, ndim, nwp, nwc, (int)lo, (int)hi, HC, TC);
    hipLaunchKernelGGL(acorr_chunkscan_kernel, dim3(nser / 64), dim3(64), 0, s, nser, nchunk, HC, TC);
    hipLaunchKernelGGL(acorr_rho_kernel, dim3(nser / 64, nchunk), dim3(64), 0, s, CT, ndim, nwp, nwc, nlive, (int)lo, (int)hi, kuse, S, T, HC, TC, P);
    hipLaunchKernelGGL(acorr_fmean_kernel, dim3((L + 255) / 256, ndim), dim3(256), 0, s, P, nser / 64, nwc / 64, nlive, L, ndim, F);
    hipLaunchKernelGGL(acorr_window_kernel, dim3(ndim), dim3(256), 0, s, F, kuse, (int)(hi - lo), c, ndim, out);
    return check_hip(hipGetLastError(), "acorr_tau");
}

int launch_chain_meanstd(const float* CT, int ndim, int nwp, int nws, int64_t t0, int64_t tm, int64_t t1, double* out,
                         hipStream_t s) {
    hipLaunchKernelGGL(chain_meanstd_kernel, dim3(ndim, 2), dim3(256), 0, s, CT, ndim, nwp, nws, t0, tm, t1, out);
    return check_hip(hipGetLastError(), "chain_meanstd");
}

}  // namespace linna
