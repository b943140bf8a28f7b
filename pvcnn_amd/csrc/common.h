// common.h -- shared host/device helpers of libpvcnn_hip.so (gfx950 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/pvcnn_hip.h"

namespace pvcnn {

constexpr int kWave = 64;                      // CDNA wavefront
constexpr int kLdsBytesPerCU = 160 * 1024;     // gfx950 LDS per CU (and max per workgroup)
constexpr int kNumCU = 256;                    // MI355X

// thread-local error string (api.hip)
void set_error(const char *fmt, ...);
// hipGetLastError() -> return code (+ error string); 0 when clean
int check_launch(const char *what);

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace pvcnn

#define PVCNN_REQUIRE(cond, msg)                                       \
  do {                                                                 \
    if (!(cond)) {                                                     \
      pvcnn::set_error("%s: %s", __func__, msg);                       \
      return PVCNN_ERR_INVALID_ARGUMENT;                               \
    }                                                                  \
  } while (0)

#define PVCNN_HIP_TRY(expr)                                            \
  do {                                                                 \
    hipError_t e_ = (expr);                                            \
    if (e_ != hipSuccess) {                                            \
      pvcnn::set_error("%s: %s failed: %s", __func__, #expr, hipGetErrorString(e_)); \
      return static_cast<int>(e_);                                     \
    }                                                                  \
  } while (0)
