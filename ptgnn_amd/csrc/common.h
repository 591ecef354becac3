// Shared helpers for libptgnn_amd (gfx950 only; no portability layer by design).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/ptgnn_amd.h"

namespace ptgnn_amd {

void set_error(const char *fmt, ...);
void count_launch(int kernel_id);   // PTGNN_AMD_KERNEL_* of include/ptgnn_amd.h

#define PTGNN_REQUIRE(cond, code, ...)            \
  do {                                            \
    if (!(cond)) {                                \
      ::ptgnn_amd::set_error(__VA_ARGS__);        \
      return (code);                              \
    }                                             \
  } while (0)

#define PTGNN_HIP(expr)                                                                    \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) {                                                                \
      ::ptgnn_amd::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),        \
                             __FILE__, __LINE__);                                          \
      return PTGNN_AMD_EHIP;                                                               \
    }                                                                                      \
  } while (0)

#define PTGNN_LAUNCH_CHECK()                                                               \
  do {                                                                                     \
    hipError_t _e = hipGetLastError();                                                     \
    if (_e != hipSuccess) {                                                                \
      ::ptgnn_amd::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e),    \
                             __FILE__, __LINE__);                                          \
      return PTGNN_AMD_EHIP;                                                               \
    }                                                                                      \
  } while (0)

constexpr int kWave = 64;        // CDNA wavefront
constexpr int kNumXcd = 8;       // MI355X: 8 XCDs, block b is observed on XCD b % 8

// XCD-aware tile order: consecutive tiles go to the SAME XCD (contiguous node ranges -- i.e. whole
// graphs of a disjoint-union batch -- share one 4 MiB L2), while hardware round-robins blockIdx
// over the XCDs.  Performance only; correctness never depends on placement.
__device__ __forceinline__ int64_t xcd_swizzle(int64_t block, int64_t nblocks) {
  const int64_t per = (nblocks + kNumXcd - 1) / kNumXcd;
  const int64_t tile = (block % kNumXcd) * per + block / kNumXcd;
  return tile;  // may be >= nblocks for the ragged tail: caller must bounds-check
}

inline int64_t xcd_padded_blocks(int64_t nblocks) {
  const int64_t per = (nblocks + kNumXcd - 1) / kNumXcd;
  return per * kNumXcd;
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace ptgnn_amd
