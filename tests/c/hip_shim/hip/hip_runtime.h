// TEST INFRASTRUCTURE: a host-memory stand-in for the handful of HIP runtime calls that
// lance_amd/csrc/index_file.cpp makes, so that the file <-> "device" glue (staging, list shards, save) can be compiled
// for the CPU and run under AddressSanitizer in this GPU-less container (tests/c/index_io_host_harness.cpp).
// "Device memory" is malloc memory; streams are synchronous.  Never part of the product build.
#pragma once
#include <cstdlib>
#include <cstring>

typedef int hipError_t;
typedef void *hipStream_t;
typedef void *hipEvent_t;
typedef void *hipGraphExec_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
#define hipHostMallocDefault 0

inline const char *hipGetErrorString(hipError_t) { return "shim error"; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { return hipMalloc(p, n); }
inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { if (n) memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { if (n) memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
