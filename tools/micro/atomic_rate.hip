// L2-atomic ceilings of the chip, for the roofline of the HBM-scratch router (BASELINE config 5: bound by dependent L2 atomics, not by
// bandwidth).  Two numbers:
//   rate     agent-scope atomicMin (no return needed, returning form used: the router reads the old value) on random 4-byte words of a
//            buffer far larger than L2, every lane its own address, as many in flight as the chip takes -> sustained atomics / s
//   latency  one lane, a chain of DEPENDENT returning atomics (the address of the next one comes out of the previous one) -> ns each
// Build: hipcc --offload-arch=gfx950 -O3 -o atomic_rate atomic_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

__global__ void k_rate(uint32_t* __restrict__ buf, uint32_t mask, int per_thread, uint32_t* __restrict__ sink) {
    uint32_t x = mix(blockIdx.x * blockDim.x + threadIdx.x + 1u), acc = 0;
    for (int i = 0; i < per_thread; i++) {
        x = mix(x + i);
        acc += __hip_atomic_fetch_min(buf + (x & mask), x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (acc == 0x12345u) sink[0] = acc;
}
__global__ void k_latency(uint32_t* __restrict__ buf, uint32_t mask, int n, uint32_t* __restrict__ sink) {
    uint32_t x = 12345u;
    for (int i = 0; i < n; i++) {
        const uint32_t old = __hip_atomic_fetch_min(buf + (x & mask), x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        x = mix(x ^ old ^ (uint32_t)i);
    }
    sink[0] = x;
}

int main() {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    uint32_t* sink; CK(hipMalloc(&sink, 64));
    for (int lg : {22, 26, 30}) {          // 16 MiB (in L2 / MALL), 256 MiB, 4 GiB of words
        const size_t words = (size_t)1 << lg;
        uint32_t* buf; CK(hipMalloc(&buf, words * 4)); CK(hipMemset(buf, 0xFF, words * 4));
        const uint32_t mask = (uint32_t)(words - 1);
        for (int blocks : {512, 2048, 8192}) {
            const int per = 256; float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_rate, dim3(blocks), dim3(256), 0, 0, buf, mask, per, sink);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            printf("rate    footprint %6zu MiB  %5d x 256 threads x %d atomics: %8.3f ms  %7.2f G atomics/s\n", words * 4 >> 20, blocks, per, best,
                   (double)blocks * 256 * per / best / 1e6);
        }
        {
            const int n = 20000; float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_latency, dim3(1), dim3(1), 0, 0, buf, mask, n, sink);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            printf("latency footprint %6zu MiB  one lane, %d dependent atomics: %8.3f ms  %7.1f ns each\n", words * 4 >> 20, n, best, best * 1e6 / n);
        }
        CK(hipFree(buf));
    }
    return 0;
}
