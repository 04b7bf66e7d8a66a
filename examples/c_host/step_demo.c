/*
 * step_demo.c — a plain C host driving libxroute_hip.so through include/xroute_hip.h, no Python, no torch:
 * what a maintainer of the reference's (C++) simulator side would write.
 *
 *   gcc -std=c11 -D__HIP_PLATFORM_AMD__ step_demo.c -I../../include -I/opt/rocm/include -L../../xroute_env_amd -lxroute_hip \
 *       -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/../../xroute_env_amd -Wl,-rpath,/opt/rocm/lib -o step_demo
 *   ./step_demo region.bin
 *
 * region.bin (written by tests/test_gpu_c_host.py): int32 X,Y,Z,n_nets, m0[3], then xs[X], ys[Y] (int32),
 * layer_dir[Z] (uint8, padded to 4 bytes), nodes[X*Y*Z] (uint32).
 * Routes the nets in ascending order and prints, per step:  action dv dwl dvia done path_len hash nets_left obs_sum
 * Odd actions go through xr_batch_step (route only) + xr_batch_observation, even ones through xr_batch_step_observe (the one-call
 * step: route + reference-layout fp32 observation); obs_fnv = position-weighted sum (mod 2^64) of the (2 + 7K) * N fp32 words of the observation, copied to the host.
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "xroute_hip.h"

#define CHECK(call) do { int32_t _rc = (call); if (_rc != XR_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, _rc, xr_last_error()); return 2; } } while (0)
#define HIP(call) do { hipError_t _e = (call); if (_e != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(_e)); return 3; } } while (0)

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s region.bin\n", argv[0]); return 1; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror("open"); return 1; }
    int32_t hdr[7];
    if (fread(hdr, 4, 7, f) != 7) return 1;
    const int X = hdr[0], Y = hdr[1], Z = hdr[2], n_nets = hdr[3];
    const int N = X * Y * Z, zpad = (Z + 3) & ~3;
    int32_t* xs = malloc(4 * X); int32_t* ys = malloc(4 * Y); uint8_t* ld = malloc(zpad); uint32_t* nodes = malloc(4 * (size_t)N);
    if (fread(xs, 4, X, f) != (size_t)X || fread(ys, 4, Y, f) != (size_t)Y || fread(ld, 1, zpad, f) != (size_t)zpad ||
        fread(nodes, 4, N, f) != (size_t)N) { fprintf(stderr, "short read\n"); return 1; }
    fclose(f);

    xr_config cfg;
    xr_config_default(&cfg);
    cfg.n_envs = 1;
    xr_batch* b = NULL;
    CHECK(xr_batch_create(&cfg, &b));
    xr_region_desc d = {X, Y, Z, xs, ys, ld, nodes, n_nets, {hdr[4], hdr[5], hdr[6]}};
    hipStream_t st;
    HIP(hipStreamCreate(&st));
    CHECK(xr_batch_load_regions(b, &d, 1, st));
    CHECK(xr_batch_reset(b, NULL, 0, st));

    int32_t *d_act, *d_delta, *d_plen, *d_nleg; uint8_t* d_done; uint64_t* d_hash; float* d_obs;
    HIP(hipMalloc((void**)&d_act, 4)); HIP(hipMalloc((void**)&d_delta, 12)); HIP(hipMalloc((void**)&d_plen, 4));
    HIP(hipMalloc((void**)&d_done, 1)); HIP(hipMalloc((void**)&d_hash, 8)); HIP(hipMalloc((void**)&d_nleg, 4));
    int64_t stride = 0;
    CHECK(xr_batch_sizes(b, NULL, NULL, NULL, NULL, NULL, NULL, &stride));
    HIP(hipMalloc((void**)&d_obs, (size_t)stride * 4));
    float* h_obs = malloc((size_t)stride * 4);
    for (int32_t a = 1; a <= n_nets; a++) {
        int32_t delta[3], plen, nleg; uint8_t done; uint64_t hash;
        HIP(hipMemcpyAsync(d_act, &a, 4, hipMemcpyHostToDevice, st));
        if (a & 1) {
            CHECK(xr_batch_step(b, d_act, st));
            CHECK(xr_batch_observation(b, d_obs, stride, 0, 1, st));
        } else {
            CHECK(xr_batch_step_observe(b, d_act, d_obs, stride, st));
        }
        CHECK(xr_batch_fetch(b, XR_FETCH_DELTA, d_delta, 12, st));
        CHECK(xr_batch_fetch(b, XR_FETCH_PATH_LEN, d_plen, 4, st));
        CHECK(xr_batch_fetch(b, XR_FETCH_DONE, d_done, 1, st));
        CHECK(xr_batch_fetch(b, XR_FETCH_HASH, d_hash, 8, st));
        HIP(hipMemcpyAsync(delta, d_delta, 12, hipMemcpyDeviceToHost, st));
        HIP(hipMemcpyAsync(&plen, d_plen, 4, hipMemcpyDeviceToHost, st));
        HIP(hipMemcpyAsync(&done, d_done, 1, hipMemcpyDeviceToHost, st));
        CHECK(xr_batch_fetch(b, XR_FETCH_NLEGAL, d_nleg, 4, st));
        HIP(hipMemcpyAsync(&hash, d_hash, 8, hipMemcpyDeviceToHost, st));
        HIP(hipMemcpyAsync(&nleg, d_nleg, 4, hipMemcpyDeviceToHost, st));
        HIP(hipStreamSynchronize(st));
        const size_t nfl = (size_t)(2 + 7 * nleg) * (size_t)N;
        HIP(hipMemcpy(h_obs, d_obs, nfl * 4, hipMemcpyDeviceToHost));
        uint64_t fnv = 0;                      /* position-weighted sum of the 32-bit words, mod 2^64 */
        const uint32_t* pw = (const uint32_t*)h_obs;
        for (size_t i = 0; i < nfl; i++) fnv += (uint64_t)pw[i] * (uint64_t)(i + 1);
        printf("%d %d %d %d %d %d %llu %d %llu\n", a, delta[0], delta[1], delta[2], (int)done, plen, (unsigned long long)hash, nleg,
               (unsigned long long)fnv);
    }
    CHECK(xr_batch_destroy(b));
    return 0;
}
