// reduce = "mul" of the scatter seam (abstractmessagepassing.py:44-50 passes the aggregation name straight to
// torch_scatter.scatter, whose reduce set is sum / add / mul / mean / min / max).  No shipped ptgnn
// configuration aggregates with a product, so this is a plain, deterministic kernel rather than a variant of the fused
// gather / segment-reduce (whose template space it would grow by a quarter): one destination row per group of LPR =
// dim / 4 lanes (16 B per lane), the row's CSR slots folded IN ORDER -- torch_scatter's CPU kernel multiplies in edge
// order, and the plan's stable sort keeps that order inside a row --, rows without in-edges stay 1 (Reducer<MUL>::init()
// and no masked_fill afterwards, unlike max / min).  HBM-bound: E * (4 dim + 4) + N * 4 dim bytes.
#include "common.h"

namespace ptgnn_amd {
namespace {

template <int VEC>
__global__ __launch_bounds__(256) void k_segment_mul(const float *__restrict__ msg, int64_t ld_msg,
                                                     const int32_t *__restrict__ rowptr, const int32_t *__restrict__ perm,
                                                     int64_t num_rows, int dim, int lanes_per_row, float *__restrict__ out,
                                                     int64_t ld_out) {
  const int rows_per_block = 256 / lanes_per_row;
  const int64_t row = (int64_t)blockIdx.x * rows_per_block + threadIdx.x / lanes_per_row;
  const int g = threadIdx.x % lanes_per_row;
  if (row >= num_rows) return;
  const int beg = rowptr[row], end = rowptr[row + 1];
  for (int col = g * VEC; col < dim; col += lanes_per_row * VEC) {
    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 1.0f;
    for (int i = beg; i < end; ++i) {
      const float *m = msg + (int64_t)perm[i] * ld_msg + col;
      if constexpr (VEC == 4) {
        const float4 t = *reinterpret_cast<const float4 *>(m);
        acc[0] *= t.x; acc[1] *= t.y; acc[2] *= t.z; acc[3] *= t.w;
      } else {
        acc[0] *= m[0];
      }
    }
    float *o = out + row * ld_out + col;
    if constexpr (VEC == 4) *reinterpret_cast<float4 *>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    else o[0] = acc[0];
  }
}

}  // namespace
}  // namespace ptgnn_amd

using namespace ptgnn_amd;

extern "C" int ptgnn_amd_segment_mul_f32(const float *msg, int64_t ld_msg, const int32_t *rowptr, const int32_t *perm,
                                         int64_t num_rows, int64_t num_edges, int32_t dim, float *out, int64_t ld_out,
                                         void *stream_) {
  PTGNN_REQUIRE(num_rows >= 0 && num_edges >= 0 && dim > 0, PTGNN_AMD_EINVAL, "segment_mul: bad sizes");
  if (num_rows == 0) return PTGNN_AMD_OK;
  PTGNN_REQUIRE(rowptr && out && (num_edges == 0 || (msg && perm)), PTGNN_AMD_EINVAL, "segment_mul: null pointer");
  PTGNN_REQUIRE(ld_out >= dim && (num_edges == 0 || ld_msg >= dim), PTGNN_AMD_EINVAL, "segment_mul: bad leading dimension");
  const bool vec4 = dim % 4 == 0 && ld_msg % 4 == 0 && ld_out % 4 == 0 && aligned16(msg) && aligned16(out);
  int lanes = 1;
  const int want = vec4 ? (dim + 3) / 4 : dim;
  while (lanes < want && lanes < 64) lanes <<= 1;
  const int rows_per_block = 256 / lanes;
  const int64_t blocks = (num_rows + rows_per_block - 1) / rows_per_block;
  PTGNN_REQUIRE(blocks < ((int64_t)1 << 31), PTGNN_AMD_EUNSUPPORTED, "segment_mul: too many rows");
  hipStream_t st = (hipStream_t)stream_;
  if (vec4) k_segment_mul<4><<<(unsigned)blocks, 256, 0, st>>>(msg, ld_msg, rowptr, perm, num_rows, dim, lanes, out, ld_out);
  else k_segment_mul<1><<<(unsigned)blocks, 256, 0, st>>>(msg, ld_msg, rowptr, perm, num_rows, dim, lanes, out, ld_out);
  PTGNN_LAUNCH_CHECK();
  return PTGNN_AMD_OK;
}
