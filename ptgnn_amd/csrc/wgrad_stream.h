// Internal interface of the streaming weight-gradient kernels (wgrad_stream.hip), used by the C-ABI entry points in
// edge_wgrad.hip.  stream_wgrad returns 1 when it took the call, 0 when the shape is not its own (the caller then
// runs the tile kernel), < 0 on a launch error.
#pragma once
#include "dense_common.h"

namespace ptgnn_amd {

constexpr int kWsTypes = 64;

struct WsTable {
  const int64_t *src[kWsTypes];
  const int64_t *dst[kWsTypes];     // null entries when the input has no target-state half
  int64_t edge_off[kWsTypes + 1];   // prefix of edges: global d_msg row of the type's edge 0
  int32_t wg_off[kWsTypes + 1];     // prefix of workgroups (filled by stream_wgrad)
  int32_t num_types;
};

size_t stream_wgrad_workspace_floats(int64_t num_edges, int num_types, int msg_dim, int in_dim);
int stream_wgrad(const WsTable &tab, const float *x, int64_t ld_x, int64_t num_rows, int state_dim, int use_dst,
                 const float *gm, int64_t ld_gm, int64_t gm_row_base, int msg_dim, float dropout_p, uint64_t dropout_seed,
                 float *grad_w, int type_base, float *grad_b, float *workspace, size_t workspace_floats,
                 hipStream_t st, const uint32_t *mask_bits = nullptr /* keep bits instead of the hash */);

}  // namespace ptgnn_amd
