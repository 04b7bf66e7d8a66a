// Device write-bandwidth probe: what a pure-write kernel can sustain on this GPU as a function of footprint and of
// the store pattern.  Build: hipcc --offload-arch=gfx950 -O3 -o write_bw write_bw.hip ; run: ./write_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// grid-stride over the whole buffer (what a plain fill does)
template <bool NT>
__global__ void k_grid(f4* __restrict__ p, size_t n4) {
    const f4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        if (NT) __builtin_nontemporal_store(v, p + i); else p[i] = v;
    }
}
// one workgroup per contiguous chunk (what the step kernel does: one env = one 2.5 MB run)
template <bool NT>
__global__ void k_chunk(f4* __restrict__ p, size_t chunk4) {
    const f4 v = {1.f, 2.f, 3.f, 4.f};
    f4* q = p + (size_t)blockIdx.x * chunk4;
    for (size_t i = threadIdx.x; i < chunk4; i += blockDim.x) {
        if (NT) __builtin_nontemporal_store(v, q + i); else q[i] = v;
    }
}

// the fused step kernel's order: the workgroup owning a chunk (= env) of `planes` planes of plane4 float4 each writes
// 4 KB of plane 0, 4 KB of plane 1, ... then the next 4 KB of every plane
template <bool NT>
__global__ void k_rot(f4* __restrict__ p, int planes, int plane4) {
    const f4 v = {1.f, 2.f, 3.f, 4.f};
    f4* q = p + (size_t)blockIdx.x * planes * plane4;
    for (int cb = 0; cb < plane4; cb += blockDim.x) {
        const int i = cb + threadIdx.x;
        if (i < plane4)
            for (int pl = 0; pl < planes; pl++) {
                if (NT) __builtin_nontemporal_store(v, q + (size_t)pl * plane4 + i); else q[(size_t)pl * plane4 + i] = v;
            }
    }
}
// globally ordered tickets: a workgroup repeatedly takes the next `blk4`-float4 block of the whole buffer
template <bool NT>
__global__ void k_ticket(f4* __restrict__ p, size_t n4, int blk4, unsigned* __restrict__ counter) {
    const f4 v = {1.f, 2.f, 3.f, 4.f};
    __shared__ unsigned s_t;
    const size_t nblk = (n4 + blk4 - 1) / blk4;
    for (;;) {
        if (threadIdx.x == 0) s_t = atomicAdd(counter, 1u);
        __syncthreads();
        const size_t t = s_t;
        __syncthreads();
        if (t >= nblk) break;
        f4* q = p + t * blk4;
        const size_t lim = (t + 1) * (size_t)blk4 <= n4 ? blk4 : n4 - t * blk4;
        for (size_t i = threadIdx.x; i < lim; i += blockDim.x) {
            if (NT) __builtin_nontemporal_store(v, q + i); else q[i] = v;
        }
    }
}

int main(int argc, char** argv) {
    const double gbs[] = {4.0, 10.9, 36.0};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (double gb : gbs) {
        size_t bytes = (size_t)(gb * 1e9) & ~(size_t)((1 << 22) - 1);
        f4* p; if (hipMalloc(&p, bytes) != hipSuccess) { printf("alloc %.1f GB failed\n", gb); continue; }
        const size_t n4 = bytes / 16;
        const size_t chunk4 = (2u << 20) / 16 + 8192;             // ~2.1 MB per workgroup
        const int nchunks = (int)(n4 / chunk4);
        const int nrot = (int)(n4 / (37 * 2160));
        unsigned* ctr; CK(hipMalloc(&ctr, 4));
        struct { const char* name; int kind; } tests[] = {{"grid plain 2048x256", 0}, {"grid NT    2048x256", 1}, {"grid NT   16384x256", 2},
                                                          {"chunk plain  wg=256", 3}, {"chunk NT     wg=256", 4}, {"memset", 5},
                                                          {"rot37 plain  wg=256", 6}, {"rot37 NT     wg=256", 7},
                                                          {"ticket 34.5K plain", 8}, {"ticket 34.5K NT", 9}, {"ticket 138K NT", 10},
                                                          {"chunk plain wg=1024", 11}, {"rot37 plain wg=1024", 12}};
        for (auto& t : tests) {
            float best = 1e30f;
            for (int rep = 0; rep < 4; rep++) {
                CK(hipEventRecord(e0));
                switch (t.kind) {
                    case 0: hipLaunchKernelGGL(k_grid<false>, dim3(2048), dim3(256), 0, 0, p, n4); break;
                    case 1: hipLaunchKernelGGL(k_grid<true>, dim3(2048), dim3(256), 0, 0, p, n4); break;
                    case 2: hipLaunchKernelGGL(k_grid<true>, dim3(16384), dim3(256), 0, 0, p, n4); break;
                    case 3: hipLaunchKernelGGL(k_chunk<false>, dim3(nchunks), dim3(256), 0, 0, p, chunk4); break;
                    case 4: hipLaunchKernelGGL(k_chunk<true>, dim3(nchunks), dim3(256), 0, 0, p, chunk4); break;
                    case 5: CK(hipMemsetAsync(p, 0, bytes, 0)); break;
                    case 6: hipLaunchKernelGGL(k_rot<false>, dim3(nrot), dim3(256), 0, 0, p, 37, 2160); break;
                    case 7: hipLaunchKernelGGL(k_rot<true>, dim3(nrot), dim3(256), 0, 0, p, 37, 2160); break;
                    case 8: CK(hipMemsetAsync(ctr, 0, 4, 0)); hipLaunchKernelGGL(k_ticket<false>, dim3(2048), dim3(256), 0, 0, p, n4, 2160, ctr); break;
                    case 9: CK(hipMemsetAsync(ctr, 0, 4, 0)); hipLaunchKernelGGL(k_ticket<true>, dim3(2048), dim3(256), 0, 0, p, n4, 2160, ctr); break;
                    case 10: CK(hipMemsetAsync(ctr, 0, 4, 0)); hipLaunchKernelGGL(k_ticket<true>, dim3(2048), dim3(256), 0, 0, p, n4, 8640, ctr); break;
                    case 11: hipLaunchKernelGGL(k_chunk<false>, dim3(nchunks), dim3(1024), 0, 0, p, chunk4); break;
                    case 12: hipLaunchKernelGGL(k_rot<false>, dim3(nrot), dim3(1024), 0, 0, p, 37, 2160); break;
                }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0 && ms < best) best = ms;
            }
            const double wr = (t.kind == 3 || t.kind == 4 || t.kind == 11) ? (double)nchunks * chunk4 * 16
                            : (t.kind == 6 || t.kind == 7 || t.kind == 12) ? (double)nrot * 37 * 2160 * 16 : (double)bytes;
            printf("%5.1f GB  %-22s %8.3f ms  %6.2f TB/s\n", gb, t.name, best, wr / best / 1e9);
        }
        CK(hipFree(p));
    }
    return 0;
}
