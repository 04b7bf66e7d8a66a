// Write-pattern probe for the unit writer of UNALIGNED channel planes (design-derived ispd18_test1 regions: N = 7650 floats per
// plane, N % 4 = 2).  A unit = 7 consecutive planes = ONE contiguous run of 7*N floats starting at float (2 + 7*rank)*N of the
// env's row.  Patterns (constant values, nontemporal float4 stores, 1024 workgroups x 256 threads, grid-stride over the units):
//   A  plane rotation: slot t of each of the 7 planes in turn; slots are 16-byte aligned, a wave's 1 KB run starts at any 16 B
//      boundary (what xr_unit_stream did in round 2)
//   D  piece rotation: the run is cut into 7 pieces at 4 KB multiples of a 128-byte aligned start; slot t of each piece in turn:
//      every wave store covers 8 whole 128-byte lines
//   C  the run as one sequential stream of aligned float4
//   R  reference: aligned planes (N = 8640), plane rotation (the aligned unit writer)
// Build: hipcc --offload-arch=gfx950 -O3 -o write_unaligned write_unaligned.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_units(float* __restrict__ out, long long stride, int N, int kper, int nunits, int mode) {
    const f4 v = {1.f, 0.f, 0.f, 1.f};
    const int tid = threadIdx.x;
    for (int u = blockIdx.x; u < nunits; u += gridDim.x) {
        const int e = u / kper, rank = u - e * kper;
        float* row = out + (long long)e * stride;
        const long long a = (long long)(2 + 7 * rank) * N;
        if (mode == 0) {             // A
            for (int t = tid; t < (N >> 2) + 1; t += 256)
#pragma unroll
                for (int pl = 0; pl < 7; pl++) {
                    const long long p0 = a + (long long)pl * N, s0 = (p0 + 3) >> 2;
                    const int r = (int)((s0 << 2) - p0);
                    if (t < ((N - r) >> 2)) __builtin_nontemporal_store(v, reinterpret_cast<f4*>(row + ((s0 + t) << 2)));
                }
        } else if (mode == 1) {      // D
            const long long S0 = ((a + 31) >> 5) << 3, S1 = ((a + 7LL * N) >> 5) << 3;
            const int nsl = (int)(S1 - S0), piece = ((nsl + 7 * 256 - 1) / (7 * 256)) * 256;
            for (int it = 0; it < piece; it += 256)
#pragma unroll
                for (int p = 0; p < 7; p++) {
                    const long long s = S0 + (long long)p * piece + it + tid;
                    if (s < S1) __builtin_nontemporal_store(v, reinterpret_cast<f4*>(row + (s << 2)));
                }
            if (tid < 32) { row[a + tid] = 1.f; row[a + 7LL * N - 1 - tid] = 1.f; }      // (ragged ends, roughly)
        } else {                     // C
            const long long S0 = (a + 3) >> 2, S1 = (a + 7LL * N) >> 2;
            for (long long s = S0 + tid; s < S1; s += 256) __builtin_nontemporal_store(v, reinterpret_cast<f4*>(row + (s << 2)));
        }
    }
}

int main() {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int B = 4096, kper = 12;
    struct { const char* name; int N; int mode; } tests[] = {{"R aligned planes N=8640, plane rotation", 8640, 0}, {"A unaligned N=7650, plane rotation (16 B aligned slots)", 7650, 0},
                                                             {"D unaligned N=7650, 4 KB piece rotation (128 B aligned)", 7650, 1}, {"C unaligned N=7650, sequential run", 7650, 2},
                                                             {"D aligned N=8640, piece rotation", 8640, 1}, {"A N=7649 (N%4=1)", 7649, 0}, {"D N=7649", 7649, 1}};
    for (auto& t : tests) {
        const long long stride = (((long long)(2 + 7 * 36) * t.N) + 3) & ~3LL;
        float* p; CK(hipMalloc(&p, (size_t)B * stride * 4));
        float best = 1e30f;
        for (int rep = 0; rep < 4; rep++) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_units, dim3(1024), dim3(256), 0, 0, p, stride, t.N, kper, B * kper, t.mode);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        const double bytes = (double)B * kper * 7 * t.N * 4;
        printf("%-60s %8.3f ms  %6.2f TB/s\n", t.name, best, bytes / best / 1e9);
        CK(hipFree(p));
    }
    return 0;
}
