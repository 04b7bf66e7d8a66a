// Microbenchmark of xr_line_pass on one workgroup: cycles per pass pair for H (L=24, stride 361),
// V (L=40, stride 9) and column (L=9, stride 1) lines.   hipcc --offload-arch=gfx950 -O3 pass_bench.hip -o pass_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../xroute_env_amd/csrc/xr_kernels.hip"

template <int MODE>
__global__ void bench(long long* out, int reps, int nlines) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t* field = reinterpret_cast<uint32_t*>(smem);
    uint32_t* el4 = field + 8664;
    const int tid = threadIdx.x;
    for (int i = tid; i < 8664; i += blockDim.x) field[i] = XR_W_UNREACHED | ((i * 2654435761u >> 28) == 0 ? 2u : 0u);
    for (int i = tid; i < 64; i += blockDim.x) el4[i] = 1600;
    if (tid == 0) field[4000] = 1;
    __syncthreads();
    const int X = 24, Y = 40, Z = 9, SY = 9, SX = 361;
    int fl = 0;
    unsigned long long acc = 0;
    long long t0 = clock64();
    for (int r = 0; r < reps; r++) {
        if (tid < nlines) {
            if (MODE == 0) {          // H line
                const int y = (tid * 7) % Y, z = 2 * (tid % 5);
                const int base = y * SY + z;
                acc |= xr_line_pass<true, true, XR_CH, false>(field, el4, base, SX, X, 0u, 12800u, 0xC0000000u, fl);
                acc |= xr_line_pass<false, true, XR_CH, false>(field, el4, base, SX, X, 0u, 12800u, 0xC0000000u, fl);
            } else if (MODE == 1) {   // V line
                const int x = (tid * 5) % X, z = 1 + 2 * (tid % 4);
                const int base = x * SX + z;
                acc |= xr_line_pass<true, true, XR_CH, false>(field, el4, base, SY, Y, 0u, 12800u, 0xC0000000u, fl);
                acc |= xr_line_pass<false, true, XR_CH, false>(field, el4, base, SY, Y, 0u, 12800u, 0xC0000000u, fl);
            } else {                  // column, exact 9
                const int c = (tid * 37) % (X * Y);
                const int x = c / Y, y = c - x * Y;
                acc |= xr_line_pass<true, false, 9, true>(field, nullptr, x * SX + y * SY, 1, 9, 3200u, 12800u, 0xC0000000u, fl);
                acc |= xr_line_pass<false, false, 9, true>(field, nullptr, x * SX + y * SY, 1, 9, 3200u, 12800u, 0xC0000000u, fl);
            }
        }
    }
    long long t1 = clock64();
    if (tid == 0) { out[0] = (t1 - t0) / reps; out[1] = (long long)acc + fl; }
}

int main() {
    long long* d; hipMalloc(&d, 64);
    long long h[2];
    for (int threads : {64, 256}) for (int nl : {1, 64, 256}) {
        if (nl > threads) continue;
        for (int mode = 0; mode < 3; mode++) {
            for (int it = 0; it < 2; it++) {
                if (mode == 0) hipLaunchKernelGGL(bench<0>, dim3(1), dim3(threads), 36000, 0, d, 200, nl);
                if (mode == 1) hipLaunchKernelGGL(bench<1>, dim3(1), dim3(threads), 36000, 0, d, 200, nl);
                if (mode == 2) hipLaunchKernelGGL(bench<2>, dim3(1), dim3(threads), 36000, 0, d, 200, nl);
                hipDeviceSynchronize();
            }
            hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
            printf("threads %3d lines %3d mode %s: %lld cycles per pass pair\n", threads, nl, mode == 0 ? "H(24)" : mode == 1 ? "V(40)" : "C(9) ", h[0]);
        }
    }
    return 0;
}
