// Micro-probe: sustained fp32 MFMA rate on gfx950 as a function of instruction shape, number of
// independent accumulator chains per wave and waves per SIMD; also reports the shader clock
// under that load (s_memtime / s_memrealtime).  Build: hipcc --offload-arch=gfx950 -O3 mfma_probe.hip -o mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND, int NACC>
__global__ __launch_bounds__(1024) void probe(int iters, float* out, unsigned long long* clk) {
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    float s = 0.f;
    if constexpr (KIND == 16) {
        f32x4 acc[NACC];
        for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
    } else {
        f32x16 acc[NACC];
        for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
}

template <int KIND, int NACC>
void run(int waves_per_simd, int iters) {
    const int threads = 64 * 4 * waves_per_simd, blocks = 256;
    float* out; unsigned long long* clk;
    hipMalloc(&out, sizeof(float) * threads * blocks); hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<KIND, NACC><<<blocks, threads>>>(iters, out, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<KIND, NACC><<<blocks, threads>>>(iters, out, clk);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double flop = (double)blocks * (threads / 64) * iters * 8.0 * NACC * (KIND == 16 ? 2048.0 : 4096.0);
    const double mhz = (double)h[0] / ((double)h[1] / 100.0);   // s_memrealtime: 100 MHz
    const double cyc_per_mfma = (double)h[0] / (iters * 8.0 * NACC) / waves_per_simd;
    printf("mfma %dx%d  chains %d  waves/SIMD %d : %7.1f us  %6.1f TFLOP/s  clock %.0f MHz  %.1f cyc/MFMA/SIMD\n", KIND, KIND, NACC,
           waves_per_simd, ms * 1e3, flop / (ms * 1e-3) / 1e12, mhz, cyc_per_mfma);
    hipFree(out); hipFree(clk);
}

int main() {
    const int it = 2000;
    run<16, 1>(1, it); run<16, 2>(1, it); run<16, 4>(1, it); run<16, 8>(1, it);
    run<16, 1>(2, it); run<16, 2>(2, it); run<16, 4>(2, it);
    run<16, 2>(4, it); run<16, 4>(4, it);
    run<32, 1>(1, it); run<32, 2>(1, it); run<32, 4>(1, it);
    run<32, 1>(2, it); run<32, 2>(2, it); run<32, 2>(4, it);
    return 0;
}
