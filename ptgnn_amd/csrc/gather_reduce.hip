// Fused gather -> (+dst term) -> segment reduce -> (GELU / LayerNorm) over a dst-sorted CSR.
// Contract + reference lines: include/ptgnn_amd.h (ptgnn_amd_gather_reduce_f32).
//
// Mapping (HBM/L2-bound, no MFMA on purpose):
//   * one destination row per group of LPR lanes, LPR = msg_dim/4 rounded to {16,32,64}; each lane
//     owns CH float4 column chunks => a 64-lane wave reads 1 KiB of message rows per
//     wave-instruction (16 B/lane, the coalescing sweet spot), 64/LPR rows per wave;
//   * the in-edges of a row are contiguous in `col` (CSR) and folded IN ORDER, so fp32 sums are
//     deterministic and follow the reference's message order; no atomics;
//   * the edge loop is unrolled x4 with all 4 row loads issued before the first use, to keep
//     >= 4 KiB per wave in flight against ~1-2 us gather latency;
//   * consecutive row tiles run on the same XCD (xcd_swizzle) so one graph of a disjoint-union
//     batch keeps its node states in a single 4 MiB L2.
// Algorithmic bytes per edge: 4*M (message row) + 4 (col) ; per node: 4*M (out) [+ 4*M dst term].
#include <float.h>

#include "common.h"

namespace ptgnn_amd {
namespace {

__device__ __forceinline__ float gelu_erf(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, LPR);
  return v;
}

struct Args {
  const float *ysrc;
  const float *ydst;
  int64_t ld_y;
  int64_t ld_yd;
  const int32_t *rowptr;
  const int32_t *col;
  int32_t type_bits;
  int64_t num_nodes;
  int32_t msg_dim;
  const float *ln_gamma;
  const float *ln_beta;
  float ln_eps;
  float *out;
  int64_t ld_out;
  int32_t *argout;
  int64_t num_tiles;
  int32_t epi;
  const int32_t *mask_arg;   // MASKED: [num source rows of this launch, M] winning forward slot
  const int32_t *mask_slot;  // MASKED: [E] forward slot of each slot of THIS plan
};

// VEC = 4: float4 path (msg_dim % 4 == 0, all bases/lds 16-B aligned); VEC = 1: generic.
// MASKED (sum only): the gathered row is an output gradient that only flows where the forward max/min
// picked this very edge:  value = (mask_arg[src, c] == mask_slot[i]) ? ysrc[src, c] : 0.
template <int VEC, int LPR, int CH, int REDUCE, bool HAS_DST, bool HAS_ARG, bool MASKED = false>
__global__ __launch_bounds__(256) void k_gather_reduce(Args a) {
  constexpr int ROWS_PER_BLOCK = 256 / LPR;
  const int64_t tile = xcd_swizzle(blockIdx.x, gridDim.x);
  if (tile >= a.num_tiles) return;
  const int g = threadIdx.x % LPR;
  const int64_t row = tile * ROWS_PER_BLOCK + threadIdx.x / LPR;
  // column block (only > 0 when msg_dim exceeds LPR*VEC*CH; epilogues are then disabled by host)
  const int cbase = blockIdx.y * (LPR * VEC * CH);
  if (row >= a.num_nodes) return;  // whole lane-group exits together (no cross-group shuffles)

  const int beg = a.rowptr[row], end = a.rowptr[row + 1];
  const int32_t tmask = (1 << a.type_bits) - 1;
  const int M = a.msg_dim;
  const int EPI = a.epi;  // wave-uniform

  float acc[CH][VEC];
  int arg[CH][VEC];
  constexpr float kInit = REDUCE == PTGNN_AMD_MAX ? -FLT_MAX : (REDUCE == PTGNN_AMD_MIN ? FLT_MAX : 0.f);
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int v = 0; v < VEC; ++v) { acc[c][v] = kInit; arg[c][v] = -1; }

  auto load_row = [&](const float *base, float (&dst)[CH][VEC]) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int colx = cbase + (g + c * LPR) * VEC;
      if constexpr (VEC == 4) {
        if (colx < M) {
          const float4 t = *reinterpret_cast<const float4 *>(base + colx);
          dst[c][0] = t.x; dst[c][1] = t.y; dst[c][2] = t.z; dst[c][3] = t.w;
        } else {
#pragma unroll
          for (int v = 0; v < VEC; ++v) dst[c][v] = 0.f;
        }
      } else {
        dst[c][0] = colx < M ? base[colx] : 0.f;
      }
    }
  };

  auto apply_mask = [&](float (&m)[CH][VEC], int64_t srow, int i) {
    if constexpr (MASKED) {
      const int want = a.mask_slot[i];
      const int32_t *ar = a.mask_arg + srow * (int64_t)M;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int colx = cbase + (g + c * LPR) * VEC;
        if constexpr (VEC == 4) {
          if (colx < M) {
            const int4 w = *reinterpret_cast<const int4 *>(ar + colx);
            m[c][0] = w.x == want ? m[c][0] : 0.f; m[c][1] = w.y == want ? m[c][1] : 0.f;
            m[c][2] = w.z == want ? m[c][2] : 0.f; m[c][3] = w.w == want ? m[c][3] : 0.f;
          }
        } else {
          if (colx < M) m[c][0] = ar[colx] == want ? m[c][0] : 0.f;
        }
      }
    }
  };

  auto fold = [&](const float (&m)[CH][VEC], int slot) {
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        if (REDUCE == PTGNN_AMD_MAX) {
          if (m[c][v] > acc[c][v]) { acc[c][v] = m[c][v]; if (HAS_ARG) arg[c][v] = slot; }
        } else if (REDUCE == PTGNN_AMD_MIN) {
          if (m[c][v] < acc[c][v]) { acc[c][v] = m[c][v]; if (HAS_ARG) arg[c][v] = slot; }
        } else {
          acc[c][v] += m[c][v];
        }
      }
  };

  const float *dst_base = HAS_DST ? a.ydst + row * a.ld_yd : nullptr;

  constexpr int U = 4;
  int i = beg;
  for (; i + U <= end; i += U) {
    int32_t pk[U];
#pragma unroll
    for (int u = 0; u < U; ++u) pk[u] = a.col[i + u];
    float m[U][CH][VEC];
    float d[U][CH][VEC];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t s = pk[u] >> a.type_bits;
      const int t = pk[u] & tmask;
      load_row(a.ysrc + s * a.ld_y + (int64_t)t * M, m[u]);
      apply_mask(m[u], s, i + u);
      if (HAS_DST) load_row(dst_base + (int64_t)t * M, d[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (HAS_DST) {
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
          for (int v = 0; v < VEC; ++v) m[u][c][v] += d[u][c][v];
      }
      fold(m[u], i + u);
    }
  }
  for (; i < end; ++i) {
    const int32_t pk = a.col[i];
    const int64_t s = pk >> a.type_bits;
    const int t = pk & tmask;
    float m[CH][VEC];
    load_row(a.ysrc + s * a.ld_y + (int64_t)t * M, m);
    apply_mask(m, s, i);
    if (HAS_DST) {
      float d[CH][VEC];
      load_row(dst_base + (int64_t)t * M, d);
#pragma unroll
      for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int v = 0; v < VEC; ++v) m[c][v] += d[c][v];
    }
    fold(m, i);
  }

  const int deg = end - beg;
  if (REDUCE == PTGNN_AMD_MEAN) {
    const float cnt = (float)(deg < 1 ? 1 : deg);
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc[c][v] = acc[c][v] / cnt;
  }
  if ((REDUCE == PTGNN_AMD_MAX || REDUCE == PTGNN_AMD_MIN) && deg == 0) {
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc[c][v] = 0.f;  // torch_scatter: empty segment -> 0
  }

  if (EPI & PTGNN_AMD_EPI_GELU) {
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc[c][v] = gelu_erf(acc[c][v]);
  }
  if (EPI & PTGNN_AMD_EPI_LAYERNORM) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
      for (int v = 0; v < VEC; ++v) s += ((g + c * LPR) * VEC + v < M) ? acc[c][v] : 0.f;
    const float mean = group_sum<LPR>(s) / (float)M;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const float dlt = acc[c][v] - mean;
        q += ((g + c * LPR) * VEC + v < M) ? dlt * dlt : 0.f;
      }
    const float rstd = 1.0f / sqrtf(group_sum<LPR>(q) / (float)M + a.ln_eps);
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const int colx = (g + c * LPR) * VEC + v;
        if (colx < M) acc[c][v] = (acc[c][v] - mean) * rstd * a.ln_gamma[colx] + a.ln_beta[colx];
      }
  }

  float *orow = a.out + row * a.ld_out;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int colx = cbase + (g + c * LPR) * VEC;
    if (colx >= M) continue;
    if constexpr (VEC == 4) {
      *reinterpret_cast<float4 *>(orow + colx) = make_float4(acc[c][0], acc[c][1], acc[c][2], acc[c][3]);
      if (HAS_ARG)
        *reinterpret_cast<int4 *>(a.argout + row * (int64_t)M + colx) =
            make_int4(arg[c][0], arg[c][1], arg[c][2], arg[c][3]);
    } else {
      orow[colx] = acc[c][0];
      if (HAS_ARG) a.argout[row * (int64_t)M + colx] = arg[c][0];
    }
  }
}

template <int VEC, int LPR, int CH, int REDUCE, bool HAS_DST>
int launch2(const Args &a, int epi, int col_blocks, hipStream_t stream) {
  constexpr int ROWS_PER_BLOCK = 256 / LPR;
  Args b = a;
  b.epi = epi;
  b.num_tiles = (a.num_nodes + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
  dim3 grid((unsigned)xcd_padded_blocks(b.num_tiles), (unsigned)col_blocks);
  if constexpr (REDUCE == PTGNN_AMD_MAX || REDUCE == PTGNN_AMD_MIN) {
    if (a.argout) {
      k_gather_reduce<VEC, LPR, CH, REDUCE, HAS_DST, true><<<grid, 256, 0, stream>>>(b);
      PTGNN_LAUNCH_CHECK();
      return PTGNN_AMD_OK;
    }
  }
  k_gather_reduce<VEC, LPR, CH, REDUCE, HAS_DST, false><<<grid, 256, 0, stream>>>(b);
  PTGNN_LAUNCH_CHECK();
  return PTGNN_AMD_OK;
}

template <int VEC, int LPR, int CH>
int launch1(const Args &a, int reduce, int epi, int col_blocks, hipStream_t s) {
  const bool d = a.ydst != nullptr;
  switch (reduce) {
    case PTGNN_AMD_SUM:
      return d ? launch2<VEC, LPR, CH, PTGNN_AMD_SUM, true>(a, epi, col_blocks, s)
               : launch2<VEC, LPR, CH, PTGNN_AMD_SUM, false>(a, epi, col_blocks, s);
    case PTGNN_AMD_MEAN:
      return d ? launch2<VEC, LPR, CH, PTGNN_AMD_MEAN, true>(a, epi, col_blocks, s)
               : launch2<VEC, LPR, CH, PTGNN_AMD_MEAN, false>(a, epi, col_blocks, s);
    case PTGNN_AMD_MAX:
      return d ? launch2<VEC, LPR, CH, PTGNN_AMD_MAX, true>(a, epi, col_blocks, s)
               : launch2<VEC, LPR, CH, PTGNN_AMD_MAX, false>(a, epi, col_blocks, s);
    default:
      return d ? launch2<VEC, LPR, CH, PTGNN_AMD_MIN, true>(a, epi, col_blocks, s)
               : launch2<VEC, LPR, CH, PTGNN_AMD_MIN, false>(a, epi, col_blocks, s);
  }
}

template <int VEC, int LPR, int CH>
int launch_masked(const Args &a, int col_blocks, hipStream_t stream) {
  constexpr int ROWS_PER_BLOCK = 256 / LPR;
  Args b = a;
  b.epi = 0;
  b.num_tiles = (a.num_nodes + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
  dim3 grid((unsigned)xcd_padded_blocks(b.num_tiles), (unsigned)col_blocks);
  k_gather_reduce<VEC, LPR, CH, PTGNN_AMD_SUM, false, false, true><<<grid, 256, 0, stream>>>(b);
  PTGNN_LAUNCH_CHECK();
  return PTGNN_AMD_OK;
}

}  // namespace
}  // namespace ptgnn_amd

using namespace ptgnn_amd;

extern "C" int ptgnn_amd_gather_reduce_masked_f32(const float *grad, int64_t ld_grad,
                                                  const int32_t *arg, const int32_t *rowptr,
                                                  const int32_t *col, const int32_t *slot_of,
                                                  int64_t num_rows, int32_t msg_dim, float *out,
                                                  int64_t ld_out, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PTGNN_REQUIRE(num_rows >= 0 && msg_dim > 0, PTGNN_AMD_EINVAL, "gather_reduce_masked: bad sizes");
  if (num_rows == 0) return PTGNN_AMD_OK;
  PTGNN_REQUIRE(grad && arg && rowptr && col && slot_of && out && ld_out >= msg_dim && ld_grad >= msg_dim,
                PTGNN_AMD_EINVAL, "gather_reduce_masked: null/ld");
  Args a{grad, nullptr, ld_grad, ld_grad, rowptr, col, 0, num_rows, msg_dim, nullptr, nullptr, 0.f,
         out, ld_out, nullptr, 0, 0, arg, slot_of};
  const bool vec4 = (msg_dim % 4 == 0) && (ld_grad % 4 == 0) && (ld_out % 4 == 0) && aligned16(grad) &&
                    aligned16(out) && aligned16(arg);
  if (vec4) {
    if (msg_dim <= 64) return launch_masked<4, 16, 1>(a, 1, stream);
    if (msg_dim <= 128) return launch_masked<4, 32, 1>(a, 1, stream);
    if (msg_dim <= 256) return launch_masked<4, 64, 1>(a, 1, stream);
    return launch_masked<4, 64, 2>(a, (msg_dim + 511) / 512, stream);
  }
  if (msg_dim <= 64) return launch_masked<1, 64, 1>(a, 1, stream);
  return launch_masked<1, 64, 4>(a, (msg_dim + 255) / 256, stream);
}

extern "C" int ptgnn_amd_gather_reduce_f32(const float *ysrc, int64_t ld_y, const float *ydst,
                                           int64_t ld_yd, const int32_t *rowptr, const int32_t *col,
                                           int32_t type_bits, int64_t num_nodes, int32_t msg_dim,
                                           int reduce, int epilogue, const float *ln_gamma,
                                           const float *ln_beta, float ln_eps, float *out,
                                           int64_t ld_out, int32_t *argout, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PTGNN_REQUIRE(num_nodes >= 0 && msg_dim > 0, PTGNN_AMD_EINVAL, "gather_reduce: bad sizes");
  PTGNN_REQUIRE(reduce >= PTGNN_AMD_SUM && reduce <= PTGNN_AMD_MIN, PTGNN_AMD_EINVAL,
                "gather_reduce: unknown reduce %d", reduce);
  PTGNN_REQUIRE(epilogue >= 0 && epilogue <= 3, PTGNN_AMD_EINVAL, "gather_reduce: bad epilogue");
  PTGNN_REQUIRE(type_bits >= 0 && type_bits < 16, PTGNN_AMD_EINVAL, "gather_reduce: bad type_bits");
  if (num_nodes == 0) return PTGNN_AMD_OK;
  PTGNN_REQUIRE(rowptr && out && ld_out >= msg_dim, PTGNN_AMD_EINVAL, "gather_reduce: null/ld");
  PTGNN_REQUIRE(ysrc && col, PTGNN_AMD_EINVAL, "gather_reduce: null ysrc/col");
  PTGNN_REQUIRE(!(epilogue & PTGNN_AMD_EPI_LAYERNORM) || (ln_gamma && ln_beta), PTGNN_AMD_EINVAL,
                "gather_reduce: LayerNorm epilogue needs gamma/beta");
  PTGNN_REQUIRE(argout == nullptr || reduce >= PTGNN_AMD_MAX, PTGNN_AMD_EINVAL,
                "gather_reduce: argout only with max/min");

  Args a{ysrc, ydst, ld_y, ydst ? ld_yd : ld_y, rowptr, col, type_bits, num_nodes, msg_dim, ln_gamma, ln_beta, ln_eps,
         out, ld_out, argout, 0, 0, nullptr, nullptr};
  const bool vec4 = (msg_dim % 4 == 0) && (ld_y % 4 == 0) && (!ydst || ld_yd % 4 == 0) && (ld_out % 4 == 0) && aligned16(ysrc) &&
                    aligned16(out) && (!ydst || aligned16(ydst)) && (!argout || aligned16(argout));
  const bool row_epi = (epilogue & PTGNN_AMD_EPI_LAYERNORM) != 0;
  if (vec4) {
    if (msg_dim <= 64) return launch1<4, 16, 1>(a, reduce, epilogue, 1, stream);
    if (msg_dim <= 128) return launch1<4, 32, 1>(a, reduce, epilogue, 1, stream);
    if (msg_dim <= 256) return launch1<4, 64, 1>(a, reduce, epilogue, 1, stream);
    if (msg_dim <= 512) return launch1<4, 64, 2>(a, reduce, epilogue, 1, stream);
    PTGNN_REQUIRE(!row_epi, PTGNN_AMD_EUNSUPPORTED,
                  "gather_reduce: LayerNorm epilogue supports msg_dim <= 512 (got %d)", msg_dim);
    return launch1<4, 64, 2>(a, reduce, epilogue, (msg_dim + 511) / 512, stream);
  }
  if (msg_dim <= 64) return launch1<1, 64, 1>(a, reduce, epilogue, 1, stream);
  if (msg_dim <= 256) return launch1<1, 64, 4>(a, reduce, epilogue, 1, stream);
  PTGNN_REQUIRE(!row_epi, PTGNN_AMD_EUNSUPPORTED,
                "gather_reduce: unaligned LayerNorm epilogue supports msg_dim <= 256 (got %d)", msg_dim);
  return launch1<1, 64, 4>(a, reduce, epilogue, (msg_dim + 255) / 256, stream);
}

// ---------------------------------------------------------------------------------------------
// plain row gather (general per-edge path + task-head indexing)
// ---------------------------------------------------------------------------------------------
namespace ptgnn_amd {
namespace {
__global__ __launch_bounds__(256) void k_gather_rows(const float *__restrict__ x, int64_t ld_x,
                                                     const int64_t *__restrict__ idx,
                                                     int64_t n_idx, int dim,
                                                     float *__restrict__ out, int64_t ld_out,
                                                     int vec4) {
  // one wave per output row, lanes stride the row
  const int lane = threadIdx.x & 63;
  const int64_t w = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = w; r < n_idx; r += nw) {
    const float *src = x + idx[r] * ld_x;
    float *dst = out + r * ld_out;
    if (vec4) {
      for (int c = lane * 4; c < dim; c += 256)
        *reinterpret_cast<float4 *>(dst + c) = *reinterpret_cast<const float4 *>(src + c);
    } else {
      for (int c = lane; c < dim; c += 64) dst[c] = src[c];
    }
  }
}
}  // namespace
}  // namespace ptgnn_amd

extern "C" int ptgnn_amd_gather_rows_f32(const float *x, int64_t ld_x, const int64_t *idx,
                                         int64_t n_idx, int32_t dim, float *out, int64_t ld_out,
                                         void *stream_) {
  PTGNN_REQUIRE(n_idx >= 0 && dim > 0, PTGNN_AMD_EINVAL, "gather_rows: bad sizes");
  if (n_idx == 0) return PTGNN_AMD_OK;
  PTGNN_REQUIRE(x && idx && out, PTGNN_AMD_EINVAL, "gather_rows: null pointer");
  const int vec4 = (dim % 4 == 0) && (ld_x % 4 == 0) && (ld_out % 4 == 0) && aligned16(x) && aligned16(out);
  int64_t blocks = (n_idx + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  k_gather_rows<<<(unsigned)blocks, 256, 0, (hipStream_t)stream_>>>(x, ld_x, idx, n_idx, dim, out,
                                                                    ld_out, vec4);
  PTGNN_LAUNCH_CHECK();
  return PTGNN_AMD_OK;
}
