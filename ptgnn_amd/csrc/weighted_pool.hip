// WeightedSumVarSizedElementReduce (ptgnn/neuralmodels/reduceops/varsizedsummary.py:68-81), the pooling of the VarMisuse
// GGNN stack's GruGlobalStateUpdate (varmisuse/train.py:87-92, globalgraphexchange.py:37-45):
//
//     score_i = sigmoid(x_i . w)                    nn.Linear(D, 1, bias=False) + torch.sigmoid
//     out[g]  = sum_{i : map[i] == g} score_i x_i   scatter_sum over element_to_sample_map
//
// Forward = k_score_scale (this file: the gemv, the sigmoid and the broadcast multiply of the reference in ONE streaming
// pass, y_i = score_i x_i rounded like the reference's product) followed by the package's ordinary segment sum over the
// map's plan (gather_reduce.hip) -- the SAME in-order fold every aggregation uses, so the pool of a graph of <= 2048 nodes
// adds its rows in the order the reference's CPU scatter_add_ does, and a sharded run (whole graphs per rank) reproduces
// the unsharded pool bit for bit.  (Round 6 first folded score, scaling and sum into one kernel over position slices of
// each segment: 1.3x less traffic on a 20 MB table that sits in the Infinity Cache anyway, but a fold order of its own --
// 3e-5 from the oracle on 2 000-node graphs, where this form is within 3e-6.)
//
// Backward (training): with go = d out,
//     d x_i = score_i go[g_i] + (go[g_i] . x_i) score_i (1 - score_i) w
//     d w   = sum_i (go[g_i] . x_i) score_i (1 - score_i) x_i
// one streaming pass for d x plus per-workgroup partial rows of d w that a second launch adds in a fixed order (no float
// atomics: d w is a fixed function of its inputs).
#include "common.h"

namespace ptgnn_amd {
namespace {

__device__ __forceinline__ float pool_sigmoid(float v) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
}

// sum of `v` over the `lanes` (power of two <= 64) consecutive lanes of a row group: xor butterfly, every lane of the
// group ends with the same bits
__device__ __forceinline__ float group_sum(float v, int lanes) {
  for (int o = lanes >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

constexpr int kPoolThreads = 256;
// COLS = chunks (float4 or float) of one row a lane holds: 1 for dim <= lanes * VEC (hidden 64 ... 256), up to 16

// y[i, :] = sigmoid(x[i, :] . w) * x[i, :]  -- one row per group of `lanes` lanes, rows independent (any order)
template <int VEC, int COLS>
__global__ __launch_bounds__(kPoolThreads) void k_score_scale(const float *__restrict__ x, int64_t ld_x,
                                                              const float *__restrict__ w, int64_t n, int dim, int lanes,
                                                              float *__restrict__ y, int64_t ld_y) {
  const int groups = kPoolThreads / lanes;
  const int g = threadIdx.x % lanes;
  float wv[COLS][VEC];
#pragma unroll
  for (int c = 0; c < COLS; ++c) {
    const int col = (c * lanes + g) * VEC;
#pragma unroll
    for (int v = 0; v < VEC; ++v) wv[c][v] = col + v < dim ? w[col + v] : 0.0f;
  }
  // every lane of a row group runs the same trip count: the shuffles of group_sum stay inside active lanes
  for (int64_t i = (int64_t)blockIdx.x * groups + threadIdx.x / lanes; i < n; i += (int64_t)gridDim.x * groups) {
    const float *row = x + i * ld_x;
    float xv[COLS][VEC];
    float dot = 0.0f;
#pragma unroll
    for (int c = 0; c < COLS; ++c) {
      const int col = (c * lanes + g) * VEC;
      if (col < dim) {
        if constexpr (VEC == 4) {
          const float4 t = *reinterpret_cast<const float4 *>(row + col);
          xv[c][0] = t.x; xv[c][1] = t.y; xv[c][2] = t.z; xv[c][3] = t.w;
        } else {
          xv[c][0] = row[col];
        }
      } else {
#pragma unroll
        for (int v = 0; v < VEC; ++v) xv[c][v] = 0.0f;
      }
#pragma unroll
      for (int v = 0; v < VEC; ++v) dot = fmaf(xv[c][v], wv[c][v], dot);
    }
    const float s = pool_sigmoid(group_sum(dot, lanes));
#pragma unroll
    for (int c = 0; c < COLS; ++c) {
      const int col = (c * lanes + g) * VEC;
      if (col < dim) {
        if constexpr (VEC == 4)
          *reinterpret_cast<float4 *>(y + i * ld_y + col) = make_float4(s * xv[c][0], s * xv[c][1], s * xv[c][2], s * xv[c][3]);
        else
          y[i * ld_y + col] = s * xv[c][0];
      }
    }
  }
}

// out[r, :] = sum over the `parts` rows partial[r * parts + s, :], s ascending
__global__ __launch_bounds__(256) void k_fold_partials(const float *__restrict__ partial, int parts, int dim, int64_t rows,
                                                        float *__restrict__ out, int64_t ld_out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * dim) return;
  const int64_t r = i / dim;
  const int col = (int)(i % dim);
  const float *p = partial + r * parts * dim + col;
  float t = 0.0f;
  for (int s = 0; s < parts; ++s) t += p[(int64_t)s * dim];
  out[r * ld_out + col] = t;
}

constexpr int kBwdRowsPerBlock = 256;     // rows one workgroup of the backward walks

template <int VEC, int COLS>
__global__ __launch_bounds__(kPoolThreads) void k_weighted_pool_backward(
    const float *__restrict__ x, int64_t ld_x, const float *__restrict__ w, const int64_t *__restrict__ map,
    const float *__restrict__ go, int64_t ld_go, int64_t n, int dim, int lanes, float *__restrict__ gx, int64_t ld_gx,
    float *__restrict__ gw_partial) {
  extern __shared__ float lds[];
  const int groups = kPoolThreads / lanes;
  const int grp = threadIdx.x / lanes, g = threadIdx.x % lanes;
  const int64_t lo = (int64_t)blockIdx.x * kBwdRowsPerBlock;
  const int64_t hi = lo + kBwdRowsPerBlock < n ? lo + kBwdRowsPerBlock : n;
  float wv[COLS][VEC], acc[COLS][VEC];
#pragma unroll
  for (int c = 0; c < COLS; ++c) {
    const int col = (c * lanes + g) * VEC;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      wv[c][v] = col + v < dim ? w[col + v] : 0.0f;
      acc[c][v] = 0.0f;
    }
  }
  for (int64_t i = lo + grp; i < hi; i += groups) {
    const float *row = x + i * ld_x;
    const float *grow = go + map[i] * ld_go;
    float xv[COLS][VEC], gv[COLS][VEC];
    float dot = 0.0f, gdot = 0.0f;
#pragma unroll
    for (int c = 0; c < COLS; ++c) {
      const int col = (c * lanes + g) * VEC;
      if (col < dim) {
        if constexpr (VEC == 4) {
          const float4 t = *reinterpret_cast<const float4 *>(row + col);
          const float4 u = *reinterpret_cast<const float4 *>(grow + col);
          xv[c][0] = t.x; xv[c][1] = t.y; xv[c][2] = t.z; xv[c][3] = t.w;
          gv[c][0] = u.x; gv[c][1] = u.y; gv[c][2] = u.z; gv[c][3] = u.w;
        } else {
          xv[c][0] = row[col];
          gv[c][0] = grow[col];
        }
      } else {
#pragma unroll
        for (int v = 0; v < VEC; ++v) xv[c][v] = gv[c][v] = 0.0f;
      }
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        dot = fmaf(xv[c][v], wv[c][v], dot);
        gdot = fmaf(xv[c][v], gv[c][v], gdot);
      }
    }
    const float s = pool_sigmoid(group_sum(dot, lanes));
    const float ds = group_sum(gdot, lanes) * s * (1.0f - s);      // d loss / d (x_i . w)
#pragma unroll
    for (int c = 0; c < COLS; ++c) {
      const int col = (c * lanes + g) * VEC;
      float o[VEC];
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        o[v] = fmaf(ds, wv[c][v], s * gv[c][v]);
        acc[c][v] = fmaf(ds, xv[c][v], acc[c][v]);
      }
      if (col < dim) {
        if constexpr (VEC == 4) *reinterpret_cast<float4 *>(gx + i * ld_gx + col) = make_float4(o[0], o[1], o[2], o[3]);
        else gx[i * ld_gx + col] = o[0];
      }
    }
  }
#pragma unroll
  for (int c = 0; c < COLS; ++c) {
    const int col = (c * lanes + g) * VEC;
#pragma unroll
    for (int v = 0; v < VEC; ++v)
      if (col + v < dim) lds[grp * dim + col + v] = acc[c][v];
  }
  __syncthreads();
  float *dst = gw_partial + (int64_t)blockIdx.x * dim;
  for (int col = threadIdx.x; col < dim; col += kPoolThreads) {
    float t = 0.0f;
    for (int r = 0; r < groups; ++r) t += lds[r * dim + col];
    dst[col] = t;
  }
}

struct PoolShape {
  bool vec4;
  int lanes, cols;     // cols: template COLS (1, 2, 4, 8 or 16), 0 = dim too wide
};

PoolShape pool_shape(int dim, bool aligned) {
  PoolShape s;
  s.vec4 = aligned && dim % 4 == 0;
  const int want = s.vec4 ? dim / 4 : dim;
  s.lanes = 1;
  while (s.lanes < want && s.lanes < 64) s.lanes <<= 1;
  const int need = (want + s.lanes - 1) / s.lanes;
  s.cols = 1;
  while (s.cols < need) s.cols <<= 1;
  if (s.cols > (s.vec4 ? 4 : 16)) s.cols = 0;
  return s;
}

// instantiate KERNEL<VEC, COLS> for the shape and launch it
#define POOL_DISPATCH(KERNEL, SH, GRID, LDS, ST, ...)                                                       \
  do {                                                                                                      \
    if ((SH).vec4) {                                                                                        \
      if ((SH).cols == 1) KERNEL<4, 1><<<(GRID), kPoolThreads, (LDS), (ST)>>>(__VA_ARGS__);                  \
      else if ((SH).cols == 2) KERNEL<4, 2><<<(GRID), kPoolThreads, (LDS), (ST)>>>(__VA_ARGS__);             \
      else KERNEL<4, 4><<<(GRID), kPoolThreads, (LDS), (ST)>>>(__VA_ARGS__);                                 \
    } else {                                                                                                \
      if ((SH).cols == 1) KERNEL<1, 1><<<(GRID), kPoolThreads, (LDS), (ST)>>>(__VA_ARGS__);                  \
      else if ((SH).cols <= 4) KERNEL<1, 4><<<(GRID), kPoolThreads, (LDS), (ST)>>>(__VA_ARGS__);             \
      else KERNEL<1, 16><<<(GRID), kPoolThreads, (LDS), (ST)>>>(__VA_ARGS__);                                \
    }                                                                                                       \
  } while (0)

}  // namespace
}  // namespace ptgnn_amd

using namespace ptgnn_amd;

extern "C" int ptgnn_amd_score_scale_f32(const float *x, int64_t ld_x, const float *w, int64_t num_rows, int32_t dim,
                                         float *y, int64_t ld_y, void *stream_) {
  PTGNN_REQUIRE(num_rows >= 0 && dim > 0, PTGNN_AMD_EINVAL, "score_scale: bad sizes");
  if (num_rows == 0) return PTGNN_AMD_OK;
  PTGNN_REQUIRE(x && w && y, PTGNN_AMD_EINVAL, "score_scale: null pointer");
  PTGNN_REQUIRE(ld_x >= dim && ld_y >= dim, PTGNN_AMD_EINVAL, "score_scale: bad leading dimension");
  const PoolShape sh = pool_shape(dim, ld_x % 4 == 0 && ld_y % 4 == 0 && aligned16(x) && aligned16(y));
  PTGNN_REQUIRE(sh.cols > 0, PTGNN_AMD_EUNSUPPORTED, "score_scale: dim %d exceeds 1024", dim);
  const int groups = kPoolThreads / sh.lanes;
  int64_t blocks = (num_rows + groups - 1) / groups;
  if (blocks > 256 * 16) blocks = 256 * 16;       // grid-stride beyond 16 workgroups per CU
  hipStream_t st = (hipStream_t)stream_;
  POOL_DISPATCH(k_score_scale, sh, (unsigned)blocks, 0, st, x, ld_x, w, num_rows, dim, sh.lanes, y, ld_y);
  PTGNN_LAUNCH_CHECK();
  return PTGNN_AMD_OK;
}

extern "C" size_t ptgnn_amd_weighted_pool_backward_workspace_bytes(int64_t num_elements, int32_t dim) {
  if (num_elements <= 0 || dim <= 0) return 0;
  return (size_t)((num_elements + kBwdRowsPerBlock - 1) / kBwdRowsPerBlock) * dim * sizeof(float);
}

extern "C" int ptgnn_amd_weighted_pool_backward_f32(const float *x, int64_t ld_x, const float *w, const int64_t *map,
                                                    const float *grad_out, int64_t ld_go, int64_t num_elements,
                                                    int32_t dim, float *grad_x, int64_t ld_gx, float *grad_w,
                                                    void *workspace, size_t workspace_bytes, void *stream_) {
  PTGNN_REQUIRE(num_elements >= 0 && dim > 0, PTGNN_AMD_EINVAL, "weighted_pool_backward: bad sizes");
  PTGNN_REQUIRE(grad_w, PTGNN_AMD_EINVAL, "weighted_pool_backward: null pointer");
  hipStream_t st = (hipStream_t)stream_;
  if (num_elements == 0) {
    PTGNN_HIP(hipMemsetAsync(grad_w, 0, (size_t)dim * sizeof(float), st));
    return PTGNN_AMD_OK;
  }
  PTGNN_REQUIRE(x && w && map && grad_out && grad_x, PTGNN_AMD_EINVAL, "weighted_pool_backward: null pointer");
  PTGNN_REQUIRE(ld_x >= dim && ld_go >= dim && ld_gx >= dim, PTGNN_AMD_EINVAL, "weighted_pool_backward: bad leading dimension");
  const bool al = ld_x % 4 == 0 && ld_go % 4 == 0 && ld_gx % 4 == 0 && aligned16(x) && aligned16(grad_out) && aligned16(grad_x);
  const PoolShape sh = pool_shape(dim, al);
  PTGNN_REQUIRE(sh.cols > 0, PTGNN_AMD_EUNSUPPORTED, "weighted_pool_backward: dim %d exceeds 1024", dim);
  const int64_t blocks = (num_elements + kBwdRowsPerBlock - 1) / kBwdRowsPerBlock;
  const size_t need = (size_t)blocks * dim * sizeof(float);
  PTGNN_REQUIRE(workspace && workspace_bytes >= need, PTGNN_AMD_EWORKSPACE,
                "weighted_pool_backward: workspace of %zu bytes, need %zu", workspace_bytes, need);
  PTGNN_REQUIRE(blocks < ((int64_t)1 << 31), PTGNN_AMD_EUNSUPPORTED, "weighted_pool_backward: too many elements");
  float *partial = static_cast<float *>(workspace);
  const size_t lds = (size_t)(kPoolThreads / sh.lanes) * dim * sizeof(float);
  POOL_DISPATCH(k_weighted_pool_backward, sh, (unsigned)blocks, lds, st, x, ld_x, w, map, grad_out, ld_go, num_elements, dim,
                sh.lanes, grad_x, ld_gx, partial);
  PTGNN_LAUNCH_CHECK();
  // d w [dim] = the workgroups' partial rows added in workgroup order: fold them as ONE "row" of `blocks` parts
  k_fold_partials<<<(unsigned)((dim + 255) / 256), 256, 0, st>>>(partial, (int)blocks, dim, 1, grad_w, dim);
  PTGNN_LAUNCH_CHECK();
  return PTGNN_AMD_OK;
}
