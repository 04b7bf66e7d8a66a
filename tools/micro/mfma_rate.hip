// The matrix instructions the towers use.  (1) One workgroup of W waves per SIMD, every wave 16384 instructions on 4 independent accumulators: the cycles of
// WAVE 0 per instruction — they do not grow with W: the SIMD issues oldest-first, wave 0 owns the pipe until it is done (which is why the per-wave stage
// cycles of the instrumented tower build form a staircase).  (2) The whole chip, wall clock: instructions per second = 16 cycles per 16x16x32 bf16 and 32 per
// 16x16x4 fp32 instruction and SIMD (the dense peaks of the data sheet).   hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip && ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ void rate(float* out, unsigned long long* cyc, int n) {
    f4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    u4 x = {threadIdx.x, 1u, 2u, 3u}, y = {5u, threadIdx.x, 7u, 9u};
    float xf = (float)threadIdx.x, yf = 2.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; i++) {
        if (KIND == 0) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, x), __builtin_bit_cast(bf8, y), a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, x), __builtin_bit_cast(bf8, y), a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, x), __builtin_bit_cast(bf8, y), a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, x), __builtin_bit_cast(bf8, y), a3, 0, 0, 0);
        } else {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xf, yf, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xf, yf, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(xf, yf, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(xf, yf, a3, 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}

int main() {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 4096 * sizeof(float)); (void)hipMalloc(&cyc, 64);
    const int n = 4096;
    for (int kind = 0; kind < 2; kind++)
        for (int w = 1; w <= 4; w++) {
            if (kind == 0) hipLaunchKernelGGL(rate<0>, dim3(1), dim3(256 * w), 0, 0, out, cyc, n);
            else hipLaunchKernelGGL(rate<1>, dim3(1), dim3(256 * w), 0, 0, out, cyc, n);
            unsigned long long c = 0; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            printf("%s, %d wave(s) per SIMD: wave 0 takes %.2f cycles per instruction (%llu cycles for %d instructions)\n",
                   kind == 0 ? "v_mfma_f32_16x16x32_bf16" : "v_mfma_f32_16x16x4_f32  ", w, (double)c / (4.0 * n), c, 4 * n);
        }
    // the whole chip: 2048 workgroups of 1024 threads, wall clock -> instructions per second
    for (int kind = 0; kind < 2; kind++) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        const int nb = 2048, nn = 8192;
        float* big; (void)hipMalloc(&big, (size_t)nb * 1024 * sizeof(float));
        unsigned long long* cb; (void)hipMalloc(&cb, nb * 8);
        for (int rep = 0; rep < 2; rep++) {
            (void)hipEventRecord(e0, 0);
            if (kind == 0) hipLaunchKernelGGL(rate<0>, dim3(nb), dim3(1024), 0, 0, big, cb, nn);
            else hipLaunchKernelGGL(rate<1>, dim3(nb), dim3(1024), 0, 0, big, cb, nn);
            (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        }
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        const double inst = (double)nb * 16 * 4.0 * nn, flop = inst * (kind == 0 ? 16384.0 : 2048.0);
        printf("%s, whole chip (2048 workgroups of 16 waves): %.3f ms for %.3g instructions = %.1f TFLOP/s = %.1f ns per instruction and SIMD (1024 SIMDs)\n",
               kind == 0 ? "v_mfma_f32_16x16x32_bf16" : "v_mfma_f32_16x16x4_f32  ", ms, inst, flop / (ms * 1e-3) / 1e12, ms * 1e6 / (inst / 1024.0));
    }
    return 0;
}
