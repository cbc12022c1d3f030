/* Test infrastructure (never shipped): the handful of HIP runtime entry points libmscnn_dist.so calls, backed by host memory, so
 * that the REAL RcclGather / mscnn_dist_* code path can run on a box without a GPU (LD_PRELOAD in tests/test_dist_cpu.py).
 * "Device" pointers are host pointers; streams are synchronous. */
#include <stdlib.h>
#include <string.h>
typedef int hipError_t;
typedef void* hipStream_t;
hipError_t hipSetDevice(int d) { (void)d; return 0; }
hipError_t hipGetDevice(int* d) { *d = 0; return 0; }
hipError_t hipMalloc(void** p, size_t n) { *p = calloc(1, n ? n : 1); return *p ? 0 : 2; }
hipError_t hipFree(void* p) { free(p); return 0; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned flags) { (void)flags; *p = calloc(1, n ? n : 1); return *p ? 0 : 2; }
hipError_t hipHostFree(void* p) { free(p); return 0; }
hipError_t hipMemset(void* p, int v, size_t n) { memset(p, v, n); return 0; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int kind, hipStream_t st) { (void)kind; (void)st; memcpy(d, s, n); return 0; }
hipError_t hipMemsetD32Async(void* p, int v, size_t count, hipStream_t st) { (void)st; for (size_t i = 0; i < count; ++i) ((int*)p)[i] = v; return 0; }
hipError_t hipStreamSynchronize(hipStream_t st) { (void)st; return 0; }
typedef void* hipEvent_t;
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned f) { (void)f; *s = (void*)1; return 0; }
hipError_t hipStreamDestroy(hipStream_t s) { (void)s; return 0; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned f) { (void)f; *e = (void*)1; return 0; }
hipError_t hipEventDestroy(hipEvent_t e) { (void)e; return 0; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) { (void)e; (void)s; return 0; }
hipError_t hipEventSynchronize(hipEvent_t e) { (void)e; return 0; }
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned f) { (void)s; (void)e; (void)f; return 0; }
const char* hipGetErrorString(hipError_t e) { return e ? "fake hip error" : "no error"; }
