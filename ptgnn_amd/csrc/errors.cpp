// Thread-local error message + version for libptgnn_amd.
#include <stdarg.h>
#include <stdio.h>

#include <atomic>

#include "../../include/ptgnn_amd.h"

namespace ptgnn_amd {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// Launch counters of the GEMM kernel families (ptgnn_amd_launch_count): which kernel a call was dispatched to is a
// shape / size / mode decision the tests assert on (a parity test that silently ran the tile kernel pins nothing about
// the streaming one).  Relaxed atomics on the host side of a launch; never read by the library itself.
static std::atomic<int64_t> g_launches[PTGNN_AMD_KERNEL_COUNT_];
void count_launch(int kernel_id) {
  if (kernel_id >= 0 && kernel_id < PTGNN_AMD_KERNEL_COUNT_) g_launches[kernel_id].fetch_add(1, std::memory_order_relaxed);
}
}  // namespace ptgnn_amd

extern "C" int64_t ptgnn_amd_launch_count(int kernel_id) {
  if (kernel_id < 0 || kernel_id >= PTGNN_AMD_KERNEL_COUNT_) return -1;
  return ptgnn_amd::g_launches[kernel_id].load(std::memory_order_relaxed);
}

extern "C" const char *ptgnn_amd_launch_name(int kernel_id) {
  static const char *const names[PTGNN_AMD_KERNEL_COUNT_] = {
      "k_stream_linear", "k_stream_linear_ring", "k_stream_gru", "k_stream_gru_ring", "k_stream_edge",
      "k_stream_edge_shared", "k_stream_edge_v2", "k_wgrad_stream", "k_linear_tlp", "k_gru", "k_edge_linear",
      "k_edge_wgrad", "k_gather_update"};
  return kernel_id >= 0 && kernel_id < PTGNN_AMD_KERNEL_COUNT_ ? names[kernel_id] : nullptr;
}

extern "C" int ptgnn_amd_version(void) { return PTGNN_AMD_VERSION; }
extern "C" const char *ptgnn_amd_last_error(void) { return ptgnn_amd::g_err; }
