// WeightedSumVarSizedElementReduce (ptgnn/neuralmodels/reduceops/varsizedsummary.py:68-81), the pooling of the VarMisuse
// GGNN stack's GruGlobalStateUpdate (varmisuse/train.py:87-92, globalgraphexchange.py:37-45):
//
//     score_i = sigmoid(x_i . w)                    nn.Linear(D, 1, bias=False) + torch.sigmoid
//     out[g]  = sum_{i : map[i] == g} score_i x_i   scatter_sum over element_to_sample_map
//
// The reference runs a [N, D] x [D, 1] gemv, a sigmoid, a broadcast multiply that materialises [N, D], and the
// scatter.  Here the score, the scaling and the segment sum are ONE pass over x: N * D * 4 bytes read, G * D * 4 written
// -- HBM-bound streaming work (cfg4: 80 k x 64 = 20 MB).
//
// Segments are graphs: few (tens) and long (thousands of rows).  One lane group per segment folding in the reference's
// serial order is what the package's ordinary segment sum does for rows of <= 2048 slots -- ~250 dependent round trips,
// 0.36 ms per pool at cfg4, 60 % on top of the whole 8-layer forward (measured, round 6) -- and segments beyond 2048 rows
// leave that order anyway (hub chunks).  So every segment is cut into CHUNKS OF 128 ROWS COUNTED FROM ITS OWN START:
// workgroup b folds chunk c of segment g (k_pool_chunk_starts gives the (g, c) of every workgroup), a last launch adds the
// chunk partials of a segment in chunk order.  No float atomics; the value of a segment is a fixed function of ITS rows
// and their order -- independent of where the segment sits in the batch, so a sharded run that keeps whole graphs on a
// rank reproduces the unsharded pool bit for bit.  The fold order is not the reference's serial one: a sum of ~2 000
// fp32 rows of magnitude <= 1 differs by ~3e-5 between ANY two orders (the fp32 oracle itself sits 2e-5 from float64 on
// such pools); the tests hold the pool, and the GRU update behind it, to "no further from float64 than 2 x the oracle".
//
// Backward (training): with go = d out,
//     d x_i = score_i go[g_i] + (go[g_i] . x_i) score_i (1 - score_i) w
//     d w   = sum_i (go[g_i] . x_i) score_i (1 - score_i) x_i
// one streaming pass for d x plus per-workgroup partial rows of d w that a second launch adds in a fixed order.
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

constexpr int kPoolChunk = 128;       // rows of one chunk, counted from the start of its segment

// chunk_start[g] = chunks of the segments in front of g (chunk_start[G] = all chunks); one workgroup
__global__ __launch_bounds__(1024) void k_pool_chunk_starts(const int32_t *__restrict__ rowptr, int num_segments,
                                                           int32_t *__restrict__ chunk_start) {
  __shared__ int wsum[16];
  __shared__ int carry;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < num_segments; base += 1024) {
    const int gseg = base + threadIdx.x;
    const int c = gseg < num_segments ? (rowptr[gseg + 1] - rowptr[gseg] + kPoolChunk - 1) / kPoolChunk : 0;
    int inc = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int run = carry + inc - c;
    for (int v = 0; v < wave; ++v) run += wsum[v];
    if (gseg < num_segments) chunk_start[gseg] = run;
    __syncthreads();
    if (threadIdx.x == 1023) carry = run + c;
    __syncthreads();
  }
  if (threadIdx.x == 0) chunk_start[num_segments] = carry;
}

// x rows are read through `perm` (plan order -> element id): a sorted map (node_to_graph_idx) makes it the identity.
template <int VEC, int COLS>
__global__ __launch_bounds__(kPoolThreads) void k_weighted_pool_partial(
    const float *__restrict__ x, int64_t ld_x, const float *__restrict__ w, const int32_t *__restrict__ rowptr,
    const int32_t *__restrict__ perm, int dim, int lanes, int num_segments, const int32_t *__restrict__ chunk_start,
    float *__restrict__ partial) {
  extern __shared__ float lds[];                       // [groups, dim] partial rows of the workgroup
  const int b = blockIdx.x;
  if (b >= chunk_start[num_segments]) return;          // the grid is the host's upper bound n / 128 + G
  int seg = 0, hi_seg = num_segments;                  // chunk_start[seg] <= b < chunk_start[seg + 1] (workgroup-uniform)
  while (hi_seg - seg > 1) {
    const int mid = (seg + hi_seg) >> 1;
    if (chunk_start[mid] <= b) seg = mid; else hi_seg = mid;
  }
  const int beg = rowptr[seg], end = rowptr[seg + 1];
  const int lo = beg + (b - chunk_start[seg]) * kPoolChunk;
  const int hi = lo + kPoolChunk < end ? lo + kPoolChunk : end;
  const int groups = kPoolThreads / lanes;
  const int grp = threadIdx.x / lanes, g = threadIdx.x % lanes;
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
  for (int p = lo + grp; p < hi; p += groups) {        // the loop bound is uniform per row group: shuffles stay in step
    const float *row = x + (int64_t)perm[p] * ld_x;
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
    for (int c = 0; c < COLS; ++c)
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc[c][v] = fmaf(s, xv[c][v], acc[c][v]);
  }
#pragma unroll
  for (int c = 0; c < COLS; ++c) {
    const int col = (c * lanes + g) * VEC;
#pragma unroll
    for (int v = 0; v < VEC; ++v)
      if (col + v < dim) lds[grp * dim + col + v] = acc[c][v];
  }
  __syncthreads();
  float *dst = partial + (int64_t)blockIdx.x * dim;    // [chunks, dim], the chunks of a segment consecutive
  for (int col = threadIdx.x; col < dim; col += kPoolThreads) {
    float t = 0.0f;
    for (int r = 0; r < groups; ++r) t += lds[r * dim + col];   // fixed order
    dst[col] = t;
  }
}

// out[g, :] = the chunk partials of segment g added in chunk order (a segment without elements pools to 0)
__global__ __launch_bounds__(256) void k_fold_segments(const float *__restrict__ partial,
                                                        const int32_t *__restrict__ chunk_start, int dim, int64_t segments,
                                                        float *__restrict__ out, int64_t ld_out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= segments * dim) return;
  const int64_t g = i / dim;
  const int col = (int)(i % dim);
  float t = 0.0f;
  for (int c = chunk_start[g]; c < chunk_start[g + 1]; ++c) t += partial[(int64_t)c * dim + col];
  out[g * ld_out + col] = t;
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

int64_t pool_chunk_bound(int64_t segments, int64_t elements) { return elements / kPoolChunk + segments; }

size_t pool_starts_bytes(int64_t segments) { return (size_t)((segments + 1 + 3) / 4 * 4) * sizeof(int32_t); }

}  // namespace
}  // namespace ptgnn_amd

using namespace ptgnn_amd;

extern "C" size_t ptgnn_amd_weighted_pool_workspace_bytes(int64_t num_segments, int64_t num_elements, int32_t dim) {
  if (num_segments <= 0 || dim <= 0) return 0;
  return pool_starts_bytes(num_segments) + (size_t)pool_chunk_bound(num_segments, num_elements) * dim * sizeof(float);
}

extern "C" int ptgnn_amd_weighted_pool_f32(const float *x, int64_t ld_x, const float *w, const int32_t *rowptr,
                                           const int32_t *perm, int64_t num_segments, int64_t num_elements, int32_t dim,
                                           float *out, int64_t ld_out, void *workspace, size_t workspace_bytes,
                                           void *stream_) {
  PTGNN_REQUIRE(num_segments >= 0 && num_elements >= 0 && dim > 0, PTGNN_AMD_EINVAL, "weighted_pool: bad sizes");
  if (num_segments == 0) return PTGNN_AMD_OK;
  PTGNN_REQUIRE(w && rowptr && out && (num_elements == 0 || (x && perm)), PTGNN_AMD_EINVAL, "weighted_pool: null pointer");
  PTGNN_REQUIRE(ld_out >= dim && (num_elements == 0 || ld_x >= dim), PTGNN_AMD_EINVAL, "weighted_pool: bad leading dimension");
  const PoolShape sh = pool_shape(dim, ld_x % 4 == 0 && aligned16(x));
  PTGNN_REQUIRE(sh.cols > 0, PTGNN_AMD_EUNSUPPORTED, "weighted_pool: dim %d exceeds 1024", dim);
  const int64_t bound = pool_chunk_bound(num_segments, num_elements);
  const size_t need = pool_starts_bytes(num_segments) + (size_t)bound * dim * sizeof(float);
  PTGNN_REQUIRE(workspace && workspace_bytes >= need, PTGNN_AMD_EWORKSPACE, "weighted_pool: workspace of %zu bytes, need %zu",
                workspace_bytes, need);
  PTGNN_REQUIRE(bound < ((int64_t)1 << 31) && num_elements < ((int64_t)1 << 31) && num_segments < ((int64_t)1 << 31),
                PTGNN_AMD_EUNSUPPORTED, "weighted_pool: too many segments / elements");
  hipStream_t st = (hipStream_t)stream_;
  int32_t *chunk_start = static_cast<int32_t *>(workspace);
  float *partial = reinterpret_cast<float *>(static_cast<char *>(workspace) + pool_starts_bytes(num_segments));
  k_pool_chunk_starts<<<1, 1024, 0, st>>>(rowptr, (int)num_segments, chunk_start);
  PTGNN_LAUNCH_CHECK();
  if (bound > 0) {
    const size_t lds = (size_t)(kPoolThreads / sh.lanes) * dim * sizeof(float);
    POOL_DISPATCH(k_weighted_pool_partial, sh, (unsigned)bound, lds, st, x, ld_x, w, rowptr, perm, dim, sh.lanes,
                  (int)num_segments, chunk_start, partial);
    PTGNN_LAUNCH_CHECK();
  }
  const int64_t total = num_segments * dim;
  k_fold_segments<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(partial, chunk_start, dim, num_segments, out, ld_out);
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
