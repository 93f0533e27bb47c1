// Micro-probe for v_mfma_f32_4x4x1_16b_f32 on gfx950: (1) operand / result layout and the A-broadcast
// controls (CBSZ = 4: the A values of block ABID feed all 16 blocks), (2) sustained rate as a function of
// the number of independent accumulator chains and waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 mfma4_probe.hip -o mfma4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int ABID>
__global__ void layout(const float* a, const float* b, float* d_plain, float* d_bcast) {
    const int l = threadIdx.x;
    f32x4 z = {0, 0, 0, 0};
    f32x4 p = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], z, 0, 0, 0);
    f32x4 q = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], z, 4, ABID, 0);
    for (int v = 0; v < 4; ++v) { d_plain[v * 64 + l] = p[v]; d_bcast[v * 64 + l] = q[v]; }
}

template <int NACC>
__global__ __launch_bounds__(1024) void rate(int iters, float* out, unsigned long long* clk) {
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f, s = 0.f;
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 4, 3, 0);
    }
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
}

template <int NACC>
void run(int wps, int iters) {
    const int threads = 64 * 4 * wps, blocks = 256;
    float* out; unsigned long long* clk;
    hipMalloc(&out, sizeof(float) * threads * blocks); hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) rate<NACC><<<blocks, threads>>>(iters, out, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    rate<NACC><<<blocks, threads>>>(iters, out, clk);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double flop = (double)blocks * (threads / 64) * iters * 8.0 * NACC * 512.0;    // 16 blocks x 4 x 4 x 1 x 2
    printf("mfma 4x4x1  chains %2d  waves/SIMD %d : %7.1f us  %6.1f TFLOP/s  clock %.0f MHz  %.2f cyc/MFMA/SIMD\n", NACC, wps,
           ms * 1e3, flop / (ms * 1e-3) / 1e12, (double)h[0] / ((double)h[1] / 100.0), (double)h[0] / (iters * 8.0 * NACC) / wps);
    hipFree(out); hipFree(clk);
}

int main() {
    float ha[64], hb[64], hp[256], hq[256];
    for (int l = 0; l < 64; ++l) { ha[l] = 1.0f + l; hb[l] = 100.0f + 3 * l; }
    float *a, *b, *p, *q;
    hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&p, 1024); hipMalloc(&q, 1024);
    hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
    layout<5><<<1, 64>>>(a, b, p, q);
    hipMemcpy(hp, p, 1024, hipMemcpyDeviceToHost); hipMemcpy(hq, q, 1024, hipMemcpyDeviceToHost);
    // expectation: plain  D[v][l] = A[4 (l/4) + v] * B[l];  broadcast (cbsz 4, abid 5)  D[v][l] = A[4*5 + v] * B[l]
    int bad_p = 0, bad_q = 0;
    for (int v = 0; v < 4; ++v)
        for (int l = 0; l < 64; ++l) {
            if (hp[v * 64 + l] != ha[4 * (l / 4) + v] * hb[l]) ++bad_p;
            if (hq[v * 64 + l] != ha[4 * 5 + v] * hb[l]) ++bad_q;
        }
    printf("layout: plain mismatches %d, broadcast(cbsz=4, abid=5) mismatches %d\n", bad_p, bad_q);
    if (bad_p || bad_q) {
        printf("plain row0: "); for (int l = 0; l < 12; ++l) printf("%g ", hp[l]); printf("\nbcast row0: ");
        for (int l = 0; l < 12; ++l) printf("%g ", hq[l]); printf("\nbcast row1: ");
        for (int l = 0; l < 12; ++l) printf("%g ", hq[64 + l]); printf("\n");
    }
    const int iters = 20000;
    run<1>(1, iters); run<2>(1, iters); run<4>(1, iters); run<8>(1, iters);
    run<2>(2, iters); run<4>(2, iters); run<8>(2, iters); run<8>(4, iters);
    return 0;
}
