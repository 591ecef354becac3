// Internal interface of the streaming GEMM core (stream_gemm.hip) used by the C-ABI entry points in
// dense_f32.hip / edge_gemm.hip.  Each stream_* launcher returns 1 when it took the call, 0 when the
// shape / mode is not its own (the caller then runs the round-1 tile kernel).
#pragma once
#include "common.h"

namespace ptgnn_amd {

constexpr int kStreamMaxTypes = 64;

struct StreamEdgeTable {
  const int64_t *src[kStreamMaxTypes];
  const int64_t *dst[kStreamMaxTypes];     // unused entries when the message has no target-state half
  const float *w[kStreamMaxTypes];
  int64_t edge_off[kStreamMaxTypes + 1];   // prefix of edges (global message row of the type's edge 0)
  int32_t unit_off[kStreamMaxTypes + 1];   // prefix of 32-edge units
  int32_t wg_off[kStreamMaxTypes + 1];     // prefix of workgroups apportioned to each type (filled by stream_edge)
  int32_t num_types;
};

int num_compute_units();

int stream_gemm_mode();          // 0 tile kernels, 1 streaming kernels (both exact fp32, same bits)
void stream_gemm_set_mode(int mode);

// `addend` (nullable): y = act(x W^T + b) + addend, [rows, n_out] with leading dimension ld_add
int stream_linear(const float *x, int64_t rows, int32_t k, int64_t ld_x, const float *w, int32_t n_out,
                  const float *bias, int act, float *y, int64_t ld_y, hipStream_t st, const float *addend = nullptr,
                  int64_t ld_add = 0);
int stream_gru(const float *a, int64_t ld_a, const float *h, int64_t ld_h, const float *w_ih,
               const float *w_hh, const float *b_ih, const float *b_hh, int64_t n, int32_t m, int32_t hd,
               float *out, int64_t ld_out, float *gates, hipStream_t st);
int stream_edge_supported(int32_t state_dim, int32_t msg_dim, int use_dst);
int stream_edge_masked_supported(int32_t state_dim, int32_t msg_dim);   // the bit-mask dropout forms (fence-free kernel)
// per-edge dropout as a bit mask (ptgnn_amd_dropout_bitmask): mode 1 = on the gathered input rows (forward), 2 = on the
// output rows (input gradient); `bits` = [message rows][ld] dwords, bit b of dword c = column 32 c + b
struct StreamEdgeMask {
  int mode;
  const uint32_t *bits;
  int ld, col0;
  float scale;
};
int stream_edge(const StreamEdgeTable &tab, const float *x, int64_t ld_x, int64_t num_rows, int32_t state_dim,
                int use_dst, int32_t msg_dim, int act, float *msg, int64_t ld_msg, int64_t msg_row_base,
                hipStream_t st, const StreamEdgeMask *mask = nullptr);

// the same GEMM over a table written on the device (ptgnn_amd_unique_sources); `edge_table_budget()` workgroups at most
int edge_table_budget();
int stream_edge_indirect(const StreamEdgeTable *tab_dev, const float *const *w_per_type, int num_types, const float *x,
                         int64_t ld_x, int64_t num_rows, int32_t state_dim, int32_t msg_dim, int act, float *msg,
                         int64_t ld_msg, hipStream_t st);

}  // namespace ptgnn_amd
