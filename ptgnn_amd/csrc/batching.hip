// Device side of the disjoint-union minibatch assembly
// (GraphNeuralNetworkModel.extend_minibatch_with / finalize_minibatch, graphneuralnetwork.py:386-493):
// the reference adds the running node offset to every per-graph int32 edge endpoint / reference id on
// the host (one numpy add per graph and edge type), concatenates, converts to int64 and uploads each
// tensor separately; node_to_graph_idx comes out of a Python generator.  Here the host ships the RAW
// int32 arrays in one staging buffer plus a small segment table, and one launch produces every int64
// index tensor of the minibatch:
//
//     out[i] = (i < n_in ? in[i] : 0) + seg_add[s],      seg_start[s] <= i < seg_start[s+1]
//
// (offset-add segments: s = one graph's slice of one array, seg_add = that graph's first node id;
//  fill segments (i >= n_in): node_to_graph_idx and reference_node_graph_idx, seg_add = graph index).
// HBM-bound integer work: 4 B read + 8 B written per element; bit-exact by construction.
#include "common.h"

namespace ptgnn_amd {
namespace {

constexpr int kItems = 4;   // elements per thread

__global__ __launch_bounds__(256) void k_batch_offsets(const int32_t *__restrict__ in, int64_t n_in,
                                                       const int64_t *__restrict__ seg_start,
                                                       const int64_t *__restrict__ seg_add,
                                                       int num_segments, int64_t n_out,
                                                       int64_t *__restrict__ out) {
  const int64_t base = (int64_t)blockIdx.x * (256 * kItems);
  if (base >= n_out) return;
  const int64_t last = (base + 256 * kItems < n_out ? base + 256 * kItems : n_out) - 1;
  // segments touched by this block (uniform): [s_lo, s_hi]
  auto find = [&](int64_t i) {
    int lo = 0, hi = num_segments;                 // invariant: seg_start[lo] <= i < seg_start[hi]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (seg_start[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
  };
  const int s_lo = find(base), s_hi = find(last);
#pragma unroll
  for (int k = 0; k < kItems; ++k) {
    const int64_t i = base + k * 256 + threadIdx.x;
    if (i >= n_out) break;
    int lo = s_lo, hi = s_hi + 1;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (seg_start[mid] <= i) lo = mid; else hi = mid;
    }
    const int64_t v = i < n_in ? (int64_t)in[i] : 0;
    out[i] = v + seg_add[lo];
  }
}

}  // namespace
}  // namespace ptgnn_amd

using namespace ptgnn_amd;

extern "C" int ptgnn_amd_batch_offsets_i64(const int32_t *in, int64_t n_in, const int64_t *seg_start,
                                           const int64_t *seg_add, int32_t num_segments,
                                           int64_t n_out, int64_t *out, void *stream_) {
  PTGNN_REQUIRE(n_in >= 0 && n_out >= n_in && num_segments >= 0, PTGNN_AMD_EINVAL, "batch_offsets: bad sizes");
  if (n_out == 0) return PTGNN_AMD_OK;
  PTGNN_REQUIRE(num_segments > 0 && seg_start && seg_add && out && (n_in == 0 || in), PTGNN_AMD_EINVAL,
                "batch_offsets: null pointer");
  const int64_t blocks = (n_out + 256 * kItems - 1) / (256 * kItems);
  PTGNN_REQUIRE(blocks < ((int64_t)1 << 31), PTGNN_AMD_EUNSUPPORTED, "batch_offsets: too many elements");
  k_batch_offsets<<<(unsigned)blocks, 256, 0, (hipStream_t)stream_>>>(in, n_in, seg_start, seg_add,
                                                                     num_segments, n_out, out);
  PTGNN_LAUNCH_CHECK();
  return PTGNN_AMD_OK;
}
