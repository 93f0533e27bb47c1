// Micro-probe for the next step named in DESIGN section 8: two workgroups on different CUs sharing one weight stream
// have to hand each other half of every segment's activations.  What does one such exchange cost on gfx950?
// Each workgroup of a pair writes `nfloats` floats to its outbox in global memory, releases a flag, waits (BOUNDED spin)
// for its partner's flag and reads the partner's outbox back; `iters` rounds, cycles per round from the shader clock.
// Pairs (b, b + 8) share an XCD (workgroup b runs on XCD b % 8: their exchange stays in that XCD's L2); pairs (b, b + 1)
// sit on neighbouring XCDs (the exchange crosses the fabric).  Grid = 250 workgroups: one per CU, all resident.
// Build: hipcc --offload-arch=gfx950 -O3 pair_probe.hip -o pair_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int MAXSPIN = 1 << 22;      // a partner that never arrives ends the wait, it does not hang the GPU

__global__ __launch_bounds__(512) void exchange(float* box, int* flag, int nfloats, int iters, int stride, unsigned long long* cycles,
                                                int* timeouts, float* check) {
    const int b = blockIdx.x;
    // partner: stride 8 -> (b, b + 8) within groups of 16; stride 1 -> (b, b ^ 1)
    const int partner = stride == 8 ? ((b % 16) < 8 ? b + 8 : b - 8) : (b ^ 1);
    if (partner >= (int)gridDim.x) return;
    // two outboxes per workgroup, used alternately: the partner rewrites the one of round `it` in round it + 2, which it
    // enters only after it has seen our flag of round it + 1 -- set after we finished reading round `it`
    float* const mine0 = box + (size_t)b * nfloats;
    const float* const theirs0 = box + (size_t)partner * nfloats;
    const size_t half = (size_t)gridDim.x * nfloats;
    float acc = 0.f;
    int lost = 0;
    __shared__ int s_ok;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 1; it <= iters; ++it) {
        float* const mine = mine0 + (it & 1) * half;
        const float* const theirs = theirs0 + (it & 1) * half;
        for (int i = threadIdx.x; i < nfloats; i += blockDim.x) mine[i] = (float)(b * 1000 + it) + 0.001f * i;
        __threadfence();                                   // the outbox is visible device-wide before the flag moves
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_store(flag + b, it, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while (__hip_atomic_load(flag + partner, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < it && ++spins < MAXSPIN)
                __builtin_amdgcn_s_sleep(1);
            s_ok = spins < MAXSPIN;
        }
        __syncthreads();
        if (!s_ok) { ++lost; break; }
        for (int i = threadIdx.x; i < nfloats; i += blockDim.x)
            acc += __builtin_nontemporal_load(theirs + i) - ((float)(partner * 1000 + it) + 0.001f * i);
        __syncthreads();
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { cycles[b] = t1 - t0; timeouts[b] = lost; }
    atomicAdd(check + b, fabsf(acc));
}

// The same exchange for pairs that share an XCD, without agent-scope release / acquire (on gfx950 these write the XCD's L2
// back to memory and invalidate it: tens of microseconds): the vector L1 writes through, so once the stores have left the
// CU (vmcnt(0)) the pair's common L2 holds them; flag and data are then read with sc1 loads, which bypass the reader's L1.
__global__ __launch_bounds__(512) void exchange_l2(float* box, int* flag, int nfloats, int iters, unsigned long long* cycles,
                                                   int* timeouts, float* check) {
    const int b = blockIdx.x;
    const int partner = (b % 16) < 8 ? b + 8 : b - 8;
    if (partner >= (int)gridDim.x) return;
    float* const mine0 = box + (size_t)b * nfloats;
    const float* const theirs0 = box + (size_t)partner * nfloats;
    const size_t half = (size_t)gridDim.x * nfloats;
    float acc = 0.f;
    int lost = 0;
    __shared__ int s_ok;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 1; it <= iters; ++it) {
        float* const mine = mine0 + (it & 1) * half;
        const float* const theirs = theirs0 + (it & 1) * half;
        for (int i = threadIdx.x; i < nfloats; i += blockDim.x) mine[i] = (float)(b * 1000 + it) + 0.001f * i;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's stores are in the L2
        __syncthreads();
        if (threadIdx.x == 0) {
            asm volatile("global_store_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" :: "v"(flag + b), "v"(it) : "memory");
            int spins = 0, seen = 0;
            do {
                asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(seen) : "v"(flag + partner) : "memory");
            } while (seen < it && ++spins < MAXSPIN);
            s_ok = spins < MAXSPIN;
        }
        __syncthreads();
        if (!s_ok) { ++lost; break; }
        for (int i = threadIdx.x; i < nfloats; i += blockDim.x) {
            float v;
            asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(theirs + i) : "memory");
            acc += v - ((float)(partner * 1000 + it) + 0.001f * i);
        }
        __syncthreads();
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { cycles[b] = t1 - t0; timeouts[b] = lost; }
    atomicAdd(check + b, fabsf(acc));
}

int main() {
    const int grid = 250, iters = 200;
    float* box; int* flag; unsigned long long* cyc; int* to; float* chk;
    hipMalloc(&box, (size_t)2 * grid * 8192 * sizeof(float));
    hipMalloc(&flag, grid * sizeof(int)); hipMalloc(&cyc, grid * sizeof(unsigned long long));
    hipMalloc(&to, grid * sizeof(int)); hipMalloc(&chk, grid * sizeof(float));
    for (int stride : {8, 1})
        for (int nfloats : {256, 2048, 8192})
            for (int threads : {256, 512}) {
                hipMemset(flag, 0, grid * sizeof(int)); hipMemset(chk, 0, grid * sizeof(float));
                hipLaunchKernelGGL(exchange, dim3(grid), dim3(threads), 0, 0, box, flag, nfloats, iters, stride, cyc, to, chk);
                if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
                std::vector<unsigned long long> c(grid); std::vector<int> t(grid); std::vector<float> k(grid);
                hipMemcpy(c.data(), cyc, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
                hipMemcpy(t.data(), to, grid * sizeof(int), hipMemcpyDeviceToHost);
                hipMemcpy(k.data(), chk, grid * sizeof(float), hipMemcpyDeviceToHost);
                double mean = 0; int lost = 0; float bad = 0;
                for (int i = 0; i < grid; ++i) { mean += (double)c[i] / iters; lost += t[i]; bad += k[i]; }
                printf("%s pairs, %5d floats (%2d KB) per exchange, %d threads: %7.0f cycles per round; timeouts %d; data error %.3g\n",
                       stride == 8 ? "same-XCD " : "cross-XCD", nfloats, nfloats * 4 / 1024, threads, mean / grid, lost, bad);
            }
    for (int nfloats : {256, 2048, 8192})
        for (int threads : {256, 512}) {
            hipMemset(flag, 0, grid * sizeof(int)); hipMemset(chk, 0, grid * sizeof(float));
            hipLaunchKernelGGL(exchange_l2, dim3(grid), dim3(threads), 0, 0, box, flag, nfloats, iters, cyc, to, chk);
            if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
            std::vector<unsigned long long> c(grid); std::vector<int> t(grid); std::vector<float> k(grid);
            hipMemcpy(c.data(), cyc, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            hipMemcpy(t.data(), to, grid * sizeof(int), hipMemcpyDeviceToHost);
            hipMemcpy(k.data(), chk, grid * sizeof(float), hipMemcpyDeviceToHost);
            double mean = 0; int lost = 0; float bad = 0;
            for (int i = 0; i < grid; ++i) { mean += (double)c[i] / iters; lost += t[i]; bad += k[i]; }
            printf("same-XCD pairs through their L2 (sc1 loads, no agent-scope fence), %5d floats (%2d KB), %d threads: %7.0f cycles per round; timeouts %d; data error %.3g\n",
                   nfloats, nfloats * 4 / 1024, threads, mean / grid, lost, bad);
        }
    return 0;
}