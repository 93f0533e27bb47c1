// Micro-probe for the GROUP broadcasts of v_mfma_f32_4x4x1_16b_f32 on gfx950: CBSZ = 2 (the A values of block 4 g + ABID feed the
// four blocks of group g = lanes 16 g .. 16 g + 15) and CBSZ = 3 (block 8 g + ABID feeds the eight blocks of group g = lanes
// 32 g .. 32 g + 31) -- what a SIDE segment of the 4x4x1 engines needs (a 16- / 32-lane group per k chunk).
// Build: hipcc --offload-arch=gfx950 -O3 mfma4_group_probe.hip -o mfma4_group_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CBSZ, int ABID>
__global__ void layout(const float* a, const float* b, float* d) {
    const int l = threadIdx.x;
    f32x4 z = {0, 0, 0, 0};
    f32x4 q = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], z, CBSZ, ABID, 0);
    for (int v = 0; v < 4; ++v) d[v * 64 + l] = q[v];
}

template <int CBSZ, int ABID>
int check(const float* a, const float* b, float* d, const float* ha, const float* hb) {
    float hq[256];
    layout<CBSZ, ABID><<<1, 64>>>(a, b, d);
    hipMemcpy(hq, d, 1024, hipMemcpyDeviceToHost);
    const int gs = 1 << CBSZ;                       // blocks per group
    int bad = 0;
    for (int v = 0; v < 4; ++v)
        for (int l = 0; l < 64; ++l) {
            const int blk = l / 4, src = (blk / gs) * gs + ABID;
            if (hq[v * 64 + l] != ha[4 * src + v] * hb[l]) ++bad;
        }
    printf("cbsz %d abid %d: mismatches %d (expect D[v][l] = A[4 ((l/4 / %d) %d + abid) + v] B[l])\n", CBSZ, ABID, bad, gs, gs);
    return bad;
}

int main() {
    float ha[64], hb[64];
    for (int l = 0; l < 64; ++l) { ha[l] = 1.0f + l; hb[l] = 100.0f + 3 * l; }
    float *a, *b, *d;
    hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024);
    hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
    int bad = 0;
    bad += check<2, 0>(a, b, d, ha, hb); bad += check<2, 1>(a, b, d, ha, hb); bad += check<2, 3>(a, b, d, ha, hb);
    bad += check<3, 0>(a, b, d, ha, hb); bad += check<3, 1>(a, b, d, ha, hb); bad += check<3, 5>(a, b, d, ha, hb);
    bad += check<4, 5>(a, b, d, ha, hb); bad += check<1, 1>(a, b, d, ha, hb);
    printf(bad ? "UNEXPECTED\n" : "ok\n");
    return bad != 0;
}
