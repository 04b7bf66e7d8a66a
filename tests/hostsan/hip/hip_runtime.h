// tests/hostsan/hip/hip_runtime.h — TEST INFRASTRUCTURE, never part of the product.
//
// A host-memory stand-in for the HIP runtime calls that xroute_env_amd/csrc/xr_batch.cpp makes, so that the product's host-side C++
// (argument checks, region / guide / state-blob validators, size arithmetic, staging buffers, error paths) can be built with
// g++ -fsanitize=address,undefined and driven without a GPU (tests/test_host_sanitizers.py).  "Device" memory is plain malloc (so
// the sanitizer's red zones sit around every device buffer and catch a mis-sized host-to-device copy), streams and events are dummies,
// kernels do not exist: the launchers of stub_launch.cpp return success without computing anything.  No result of this build is ever
// compared with anything — the parity tests run on the real library on a real GPU.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1 };
typedef struct xr_stub_stream* hipStream_t;
typedef struct xr_stub_event* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
struct hipDeviceProp_t { int multiProcessorCount; size_t sharedMemPerBlock; size_t totalGlobalMem; };

// allocations above this many bytes fail (tests shrink it to walk the out-of-memory paths); -1 = no limit
extern "C" int64_t xr_stub_alloc_limit;
extern "C" int64_t xr_stub_alloc_live;     // bytes currently allocated (leak check of the error paths)

static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : e == hipErrorOutOfMemory ? "out of memory (stub)" : "invalid value (stub)"; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidValue; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { p->multiProcessorCount = 256; p->sharedMemPerBlock = 160 * 1024; p->totalGlobalMem = (size_t)1 << 30; return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) {
    if (xr_stub_alloc_limit >= 0 && (int64_t)n > xr_stub_alloc_limit) { *p = nullptr; return hipErrorOutOfMemory; }
    // the size rides in a 16-byte header so that hipFree can keep the live-bytes count
    uint8_t* q = static_cast<uint8_t*>(malloc(n + 16));
    if (!q) { *p = nullptr; return hipErrorOutOfMemory; }
    *reinterpret_cast<uint64_t*>(q) = n;
    xr_stub_alloc_live += (int64_t)n;
    *p = q + 16;
    return hipSuccess;
}
static inline hipError_t hipFree(void* p) {
    if (!p) return hipSuccess;
    uint8_t* q = static_cast<uint8_t*>(p) - 16;
    xr_stub_alloc_live -= (int64_t)*reinterpret_cast<uint64_t*>(q);
    free(q);
    return hipSuccess;
}
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { if (n) memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { if (n) memset(d, v, n); return hipSuccess; }
typedef void* hipDeviceptr_t;
static inline hipError_t hipMemsetD32Async(hipDeviceptr_t d, int v, size_t count, hipStream_t = nullptr) {
    uint32_t* q = static_cast<uint32_t*>(d);
    for (size_t i = 0; i < count; i++) q[i] = (uint32_t)v;
    return hipSuccess;
}
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = reinterpret_cast<hipStream_t>(malloc(1)); return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = reinterpret_cast<hipEvent_t>(malloc(1)); return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
static inline hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = nullptr) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
