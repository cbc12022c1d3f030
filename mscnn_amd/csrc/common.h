// Shared host-side helpers for libmscnn_hip.so (gfx950 only; no CUDA/dual-path code).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include "../../include/mscnn_hip.h"

namespace mscnn {

void set_error(const char* fmt, ...);

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// hipPeekAtLastError after each launch mirrors CUDA_POST_KERNEL_CHECK (device_alternate.hpp:76),
// but reports through the return code instead of aborting.
#define MSCNN_HIP_TRY(expr)                                                              \
  do {                                                                                   \
    hipError_t e__ = (expr);                                                             \
    if (e__ != hipSuccess) {                                                             \
      ::mscnn::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e__)); \
      return MSCNN_ERR_HIP;                                                              \
    }                                                                                    \
  } while (0)

#define MSCNN_POST_LAUNCH() MSCNN_HIP_TRY(hipPeekAtLastError())

#define MSCNN_REQUIRE(cond, ...)                 \
  do {                                           \
    if (!(cond)) {                               \
      ::mscnn::set_error(__VA_ARGS__);           \
      return MSCNN_ERR_BAD_ARG;                  \
    }                                            \
  } while (0)

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Tuning knobs come from the descriptors (mscnn_conv_desc::tune_*).  Only a debug build (make EXTRA=-DMSCNN_TUNING_ENV)
// lets an environment variable override them for quick A/B runs; the product library never reads the environment.
#ifdef MSCNN_TUNING_ENV
int tune_env(const char* name, int dflt);
#else
inline int tune_env(const char*, int dflt) { return dflt; }
#endif

}  // namespace mscnn
