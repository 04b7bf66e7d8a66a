// Shader clock under load: a latency-bound LDS pointer chase (what a route is) timed with clock64() (shader cycles) and
// wall_clock64() (100 MHz), alone and while the other workgroups of the same launch stream float4 stores to HBM (what the unit
// writers of the queue-form step kernel do).  cycles / microsecond = the clock the chase really ran at.
//   hipcc --offload-arch=gfx950 -O3 -o clock_probe clock_probe.hip && ./clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void __launch_bounds__(256) probe(float4* out, size_t n4, int writers_every, int chase_iters, long long* res) {
    __shared__ int s_next[1024];
    const int tid = threadIdx.x;
    const bool writer = writers_every > 0 && (blockIdx.x % writers_every) != 0;
    if (writer) {                                   // stream stores until the chasers are done (bounded)
        const size_t per = n4 / gridDim.x;
        float4* p = out + per * blockIdx.x;
        for (int rep = 0; rep < 64; rep++)
            for (size_t i = tid; i < per; i += 256) p[i] = make_float4(rep, 1, 2, 3);
        return;
    }
    for (int i = tid; i < 1024; i += 256) s_next[i] = (i * 197 + 31) & 1023;
    __syncthreads();
    const long long c0 = clock64(), w0 = wall_clock64();
    int v = tid;
    for (int i = 0; i < chase_iters; i++) v = s_next[v];
    const long long c1 = clock64(), w1 = wall_clock64();
    if (tid == 0) { res[blockIdx.x * 3] = c1 - c0; res[blockIdx.x * 3 + 1] = w1 - w0; res[blockIdx.x * 3 + 2] = v; }
}

int main() {
    const size_t bytes = 8ull << 30;
    float4* out; CK(hipMalloc(&out, bytes));
    const int blocks = 1024;
    long long* res; CK(hipMalloc(&res, blocks * 3 * sizeof(long long)));
    std::vector<long long> h(blocks * 3);
    for (int every : {0, 2, 4, 8}) {
        for (int rep = 0; rep < 3; rep++) {
            CK(hipMemset(res, 0, blocks * 3 * sizeof(long long)));
            hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, 0, out, bytes / 16, every, 40000, res);
            CK(hipDeviceSynchronize());
        }
        CK(hipMemcpy(h.data(), res, blocks * 3 * sizeof(long long), hipMemcpyDeviceToHost));
        double cyc = 0, wall = 0; int n = 0;
        for (int b = 0; b < blocks; b++) if (h[b * 3 + 1] > 0) { cyc += h[b * 3]; wall += h[b * 3 + 1]; n++; }
        printf("writers: %s  chasers %4d  chase %8.0f cycles in %7.1f us  -> %6.0f MHz, %.1f cycles per dependent LDS read\n",
               every == 0 ? "none            " : every == 2 ? "1 of 2 workgroups" : every == 4 ? "3 of 4 workgroups" : "7 of 8 workgroups",
               n, cyc / n, wall / n / 100.0, cyc / wall * 100.0, cyc / n / 40000.0);
    }
    return 0;
}
