// Row epilogue of the MLP message-passing layer as a differentiable pair: y = LayerNorm(GELU(x)) over the rows
// of the aggregated message matrix, and its backward.  Contract + reference lines: include/ptgnn_amd.h
// (ptgnn_amd_row_epilogue_f32 / _backward_f32).
//
// In inference the same arithmetic is the epilogue of the gather/segment-reduce kernel (gather_reduce.hip); the
// training step needs it as an autograd node: the pre-activation rows must survive for the backward, and the
// backward itself (d LayerNorm -> d GELU -> column sums for d gamma / d beta) was three torch kernels of
// ~100 us each at [116 k, 64] -- their row reductions are written for wide rows -- plus the exact-erf GELU pair.
//
// Mapping (HBM-bound: forward reads + writes N*M floats, backward reads 2, writes 1): one row per group of LPR
// lanes, LPR = M/4 rounded up to {16, 32, 64} (float4 columns, CH column chunks per lane), like the aggregation
// kernel, so a wave moves 1 KiB per instruction whatever the row width; the row statistics are LPR-wide
// shuffles.  The backward walks rows with a grid stride so that every lane accumulates d gamma / d beta of ITS
// columns in registers; a workgroup folds its row groups through LDS and stores one partial row, and a second
// small kernel sums the partial rows in a fixed order (deterministic, no float atomics).
// Numerics: the forward is statement for statement the epilogue of gather_reduce.hip (two-pass mean / variance,
// exact-erf GELU), so the training forward equals the inference forward bit for bit.
#include <type_traits>

#include "common.h"

namespace ptgnn_amd {
namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ float gelu_erf(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

__device__ __forceinline__ float gelu_erf_grad(float x) {
  // d/dx [x Phi(x)] = Phi(x) + x phi(x)
  return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * expf(-0.5f * x * x);
}

template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, LPR);
  return v;
}

struct RowArgs {
  const float *x; int64_t ld_x;
  const float *dy; int64_t ld_dy;
  float *out; int64_t ld_out;          // forward: y; backward: dx
  const float *gamma, *beta;
  float eps;
  int64_t rows;
  int M;
  int flags;                           // PTGNN_AMD_EPI_*
  float *partial;                      // backward: [gridDim.x][2 * M] (d gamma | d beta) per workgroup
};

// Row state of one lane group: CH chunks of VEC columns per lane; column of (c, v) = (g + c * LPR) * VEC + v.
template <int VEC, int LPR, int CH>
struct Row {
  float a[CH][VEC];
  __device__ __forceinline__ void load(const float *row, int g, int M, float fill = 0.f) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int col = (g + c * LPR) * VEC;
      if constexpr (VEC == 4) {
        if (col < M) {
          const float4 t = *reinterpret_cast<const float4 *>(row + col);
          a[c][0] = t.x; a[c][1] = t.y; a[c][2] = t.z; a[c][3] = t.w;
        } else {
          a[c][0] = a[c][1] = a[c][2] = a[c][3] = fill;
        }
      } else {
        a[c][0] = col < M ? row[col] : fill;
      }
    }
  }
  __device__ __forceinline__ void store(float *row, int g, int M) const {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int col = (g + c * LPR) * VEC;
      if (col >= M) continue;
      if constexpr (VEC == 4) *reinterpret_cast<float4 *>(row + col) = make_float4(a[c][0], a[c][1], a[c][2], a[c][3]);
      else row[col] = a[c][0];
    }
  }
};

// mean and 1/std of the row held in `r` (two passes, as the aggregation epilogue does)
template <int VEC, int LPR, int CH>
__device__ __forceinline__ void row_stats(const Row<VEC, LPR, CH> &r, int g, int M, float eps, float &mean, float &rstd) {
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int v = 0; v < VEC; ++v) s += ((g + c * LPR) * VEC + v < M) ? r.a[c][v] : 0.f;
  mean = group_sum<LPR>(s) / (float)M;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      const float d = r.a[c][v] - mean;
      q += ((g + c * LPR) * VEC + v < M) ? d * d : 0.f;
    }
  rstd = 1.0f / sqrtf(group_sum<LPR>(q) / (float)M + eps);
}

// Rows per lane group and pass: ONE.  Several rows requested up front (2 / 4 / 8 per group) were measured at
// [116 k, 64] and [116 k, 128] and lose -- forward 22.1 -> 24.8 / 25.6 / 26.4 us, backward 35.4 -> 36.1 / 39.7 / 46.1 us
// (HIP events; a torch copy of the same matrix takes 12 us): the extra live rows cost occupancy and buy nothing.  Nor is
// the exact-erf GELU what the forward waits for: an Abramowitz-Stegun erf on v_exp_f32 / v_rcp_f32 left it at 22 us
// (backward 35 -> 32 us); not adopted, the forward must stay the fused aggregation epilogue's arithmetic bit for bit.
template <int LPR> constexpr int rows_in_flight() { return 1; }

template <int VEC, int LPR, int CH>
__device__ __forceinline__ void row_forward(const RowArgs &p, Row<VEC, LPR, CH> &r, int g) {
  if (p.flags & PTGNN_AMD_EPI_GELU) {
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
      for (int v = 0; v < VEC; ++v) r.a[c][v] = gelu_erf(r.a[c][v]);
  }
  if (p.flags & PTGNN_AMD_EPI_LAYERNORM) {
    float mean, rstd;
    row_stats(r, g, p.M, p.eps, mean, rstd);
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const int col = (g + c * LPR) * VEC + v;
        if (col < p.M) r.a[c][v] = (r.a[c][v] - mean) * rstd * p.gamma[col] + p.beta[col];
      }
  }
}

template <int VEC, int LPR, int CH>
__global__ __launch_bounds__(kBlock) void k_row_epilogue(RowArgs p) {
  constexpr int G = kBlock / LPR, R = rows_in_flight<LPR>();
  const int g = threadIdx.x % LPR;
  const int64_t row0 = ((int64_t)blockIdx.x * G + threadIdx.x / LPR) * R;
  if (row0 >= p.rows) return;          // whole lane group leaves together
  Row<VEC, LPR, CH> r[R];
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const int64_t row = row0 + i < p.rows ? row0 + i : p.rows - 1;
    r[i].load(p.x + row * p.ld_x, g, p.M);
  }
#pragma unroll
  for (int i = 0; i < R; ++i) {
    row_forward(p, r[i], g);
    if (row0 + i < p.rows) r[i].store(p.out + (row0 + i) * p.ld_out, g, p.M);
  }
}

template <int VEC, int LPR, int CH>
__global__ __launch_bounds__(kBlock) void k_row_epilogue_backward(RowArgs p) {
  constexpr int G = kBlock / LPR;
  __shared__ float fold[kBlock * VEC * CH * 2];
  const int g = threadIdx.x % LPR, grp = threadIdx.x / LPR;
  const bool ln = (p.flags & PTGNN_AMD_EPI_LAYERNORM) != 0, gelu = (p.flags & PTGNN_AMD_EPI_GELU) != 0;
  float gam[CH][VEC], dgam[CH][VEC], dbet[CH][VEC];
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      const int col = (g + c * LPR) * VEC + v;
      gam[c][v] = (ln && col < p.M) ? p.gamma[col] : 0.f;
      dgam[c][v] = dbet[c][v] = 0.f;
    }
  // rows this lane group owns: R consecutive rows per pass (all requested up front), passes with a grid stride
  constexpr int R = rows_in_flight<LPR>();
  for (int64_t base = ((int64_t)blockIdx.x * G + grp) * R; base < p.rows; base += (int64_t)gridDim.x * G * R) {
    Row<VEC, LPR, CH> xs[R], ds[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int64_t rr = base + i < p.rows ? base + i : p.rows - 1;
      xs[i].load(p.x + rr * p.ld_x, g, p.M);
      ds[i].load(p.dy + rr * p.ld_dy, g, p.M);
    }
#pragma unroll
    for (int i = 0; i < R; ++i) {
    const int64_t row = base + i;
    if (row >= p.rows) break;            // whole lane group: `row` is uniform inside it
    Row<VEC, LPR, CH> x = xs[i], d = ds[i], h;
    h = x;
    if (gelu) {
#pragma unroll
      for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int v = 0; v < VEC; ++v) h.a[c][v] = gelu_erf(x.a[c][v]);
    }
    if (ln) {
      float mean, rstd;
      row_stats(h, g, p.M, p.eps, mean, rstd);
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          const bool in = (g + c * LPR) * VEC + v < p.M;
          const float xhat = in ? (h.a[c][v] - mean) * rstd : 0.f;
          const float dy = in ? d.a[c][v] : 0.f;
          dgam[c][v] += dy * xhat;
          dbet[c][v] += dy;
          const float dxh = dy * gam[c][v];
          h.a[c][v] = xhat;            // keep x-hat
          d.a[c][v] = dxh;             // and d x-hat
          s1 += dxh;
          s2 += dxh * xhat;
        }
      const float m1 = group_sum<LPR>(s1) / (float)p.M, m2 = group_sum<LPR>(s2) / (float)p.M;
#pragma unroll
      for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int v = 0; v < VEC; ++v) d.a[c][v] = rstd * (d.a[c][v] - m1 - h.a[c][v] * m2);
    }
    if (gelu) {
#pragma unroll
      for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int v = 0; v < VEC; ++v) d.a[c][v] *= gelu_erf_grad(x.a[c][v]);
    }
    d.store(p.out + row * p.ld_out, g, p.M);
    }
  }
  if (!ln) return;
  // fold the G lane groups of this workgroup (fixed order) and store one partial row
  constexpr int W = LPR * VEC * CH;       // padded row width held by one group
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      fold[(grp * W + (g + c * LPR) * VEC + v) * 2] = dgam[c][v];
      fold[(grp * W + (g + c * LPR) * VEC + v) * 2 + 1] = dbet[c][v];
    }
  __syncthreads();
  for (int col = threadIdx.x; col < p.M; col += kBlock) {
    float sg = 0.f, sb = 0.f;
    for (int q = 0; q < G; ++q) {
      sg += fold[(q * W + col) * 2];
      sb += fold[(q * W + col) * 2 + 1];
    }
    p.partial[(int64_t)blockIdx.x * 2 * p.M + col] = sg;
    p.partial[(int64_t)blockIdx.x * 2 * p.M + p.M + col] = sb;
  }
}

// out[j] = sum over partial rows, j < 2 M: 64 columns x 16 row slices per workgroup; a slice adds its rows in
// ascending order (8 independent loads per step), the 16 slice sums meet in LDS in a fixed order (deterministic)
constexpr int kSumSlices = 16;
__global__ __launch_bounds__(64 * kSumSlices) void k_partial_rows_sum(const float *__restrict__ partial, int nrows,
                                                                      int width, float *__restrict__ dgamma,
                                                                      float *__restrict__ dbeta, int M) {
  __shared__ float s[kSumSlices][64];
  const int c = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + c;
  float acc = 0.f;
  if (j < width) {
#pragma unroll 8
    for (int r = slice; r < nrows; r += kSumSlices) acc += partial[(int64_t)r * width + j];
  }
  s[slice][c] = acc;
  __syncthreads();
  if (slice == 0 && j < width) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < kSumSlices; ++q) t += s[q][c];
    if (j < M) dgamma[j] = t; else dbeta[j - M] = t;
  }
}

constexpr int kMaxBackwardBlocks = 2048;
int backward_blocks(int64_t rows, int rows_per_block_pass) {
  // eight workgroups per CU keep the loads of this HBM-bound pass in flight; every workgroup costs one partial row
  int64_t b = (rows + rows_per_block_pass - 1) / rows_per_block_pass;
  if (b > kMaxBackwardBlocks) b = kMaxBackwardBlocks;
  return (int)(b < 1 ? 1 : b);
}

template <typename F>
int dispatch(bool vec4, int M, F f) {
  if (vec4) {
    if (M <= 64) return f(std::integral_constant<int, 4>{}, std::integral_constant<int, 16>{}, std::integral_constant<int, 1>{});
    if (M <= 128) return f(std::integral_constant<int, 4>{}, std::integral_constant<int, 32>{}, std::integral_constant<int, 1>{});
    if (M <= 256) return f(std::integral_constant<int, 4>{}, std::integral_constant<int, 64>{}, std::integral_constant<int, 1>{});
    return f(std::integral_constant<int, 4>{}, std::integral_constant<int, 64>{}, std::integral_constant<int, 2>{});
  }
  if (M <= 64) return f(std::integral_constant<int, 1>{}, std::integral_constant<int, 64>{}, std::integral_constant<int, 1>{});
  if (M <= 256) return f(std::integral_constant<int, 1>{}, std::integral_constant<int, 64>{}, std::integral_constant<int, 4>{});
  return f(std::integral_constant<int, 1>{}, std::integral_constant<int, 64>{}, std::integral_constant<int, 8>{});
}

int check_common(const char *who, const float *x, int64_t ld_x, int64_t rows, int32_t dim, int32_t flags,
                 const float *gamma) {
  PTGNN_REQUIRE(rows >= 0 && dim > 0 && dim <= 512, PTGNN_AMD_EUNSUPPORTED, "%s: row width %d outside (0, 512]", who, dim);
  PTGNN_REQUIRE(flags > 0 && flags <= PTGNN_AMD_EPI_GELU_LAYERNORM, PTGNN_AMD_EINVAL, "%s: flags must name GELU and/or LayerNorm", who);
  PTGNN_REQUIRE(rows == 0 || (x != nullptr && ld_x >= dim), PTGNN_AMD_EINVAL, "%s: bad input matrix", who);
  PTGNN_REQUIRE(!(flags & PTGNN_AMD_EPI_LAYERNORM) || gamma != nullptr, PTGNN_AMD_EINVAL, "%s: LayerNorm needs gamma", who);
  return PTGNN_AMD_OK;
}

}  // namespace
}  // namespace ptgnn_amd

using namespace ptgnn_amd;

extern "C" int ptgnn_amd_row_epilogue_f32(const float *x, int64_t ld_x, int64_t rows, int32_t dim, int32_t flags,
                                          const float *ln_gamma, const float *ln_beta, float ln_eps, float *y,
                                          int64_t ld_y, void *stream_) {
  hipStream_t st = (hipStream_t)stream_;
  if (int rc = check_common("row_epilogue", x, ld_x, rows, dim, flags, ln_gamma)) return rc;
  PTGNN_REQUIRE(!(flags & PTGNN_AMD_EPI_LAYERNORM) || ln_beta != nullptr, PTGNN_AMD_EINVAL, "row_epilogue: LayerNorm needs beta");
  PTGNN_REQUIRE(rows == 0 || (y != nullptr && ld_y >= dim), PTGNN_AMD_EINVAL, "row_epilogue: bad output matrix");
  if (rows == 0) return PTGNN_AMD_OK;
  RowArgs p{};
  p.x = x; p.ld_x = ld_x; p.out = y; p.ld_out = ld_y; p.gamma = ln_gamma; p.beta = ln_beta; p.eps = ln_eps;
  p.rows = rows; p.M = dim; p.flags = flags;
  const bool vec4 = dim % 4 == 0 && ld_x % 4 == 0 && ld_y % 4 == 0 && aligned16(x) && aligned16(y);
  return dispatch(vec4, dim, [&](auto VEC, auto LPR, auto CH) -> int {
    constexpr int GR = kBlock / decltype(LPR)::value * rows_in_flight<decltype(LPR)::value>();
    k_row_epilogue<decltype(VEC)::value, decltype(LPR)::value, decltype(CH)::value>
        <<<(unsigned)((rows + GR - 1) / GR), kBlock, 0, st>>>(p);
    PTGNN_LAUNCH_CHECK();
    return (int)PTGNN_AMD_OK;
  });
}

namespace ptgnn_amd {
namespace {
// out = grad * (keep ? scale : 0) * act'(y), y = act(.) the forward output BEFORE dropout: the backward of the MLP-MP
// node update's Tanh + Dropout (mlpmessagepassing.py:62-66) as one pass instead of torch's two elementwise kernels
template <int ACT, bool KEEP>
__global__ __launch_bounds__(256) void k_act_dropout_backward(const float *__restrict__ grad, const float *__restrict__ y,
                                                              const uint8_t *__restrict__ keep, float scale, int64_t n4,
                                                              float *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float4 g = reinterpret_cast<const float4 *>(grad)[i];
  const float4 v = reinterpret_cast<const float4 *>(y)[i];
  float m[4] = {1.f, 1.f, 1.f, 1.f};
  if constexpr (KEEP) {
    const uchar4 k = reinterpret_cast<const uchar4 *>(keep)[i];
    m[0] = k.x ? scale : 0.f; m[1] = k.y ? scale : 0.f; m[2] = k.z ? scale : 0.f; m[3] = k.w ? scale : 0.f;
  }
  auto d = [](float yv) {
    if constexpr (ACT == PTGNN_AMD_ACT_TANH) return 1.0f - yv * yv;
    if constexpr (ACT == PTGNN_AMD_ACT_RELU) return yv > 0.f ? 1.0f : 0.f;
    return 1.0f;
  };
  reinterpret_cast<float4 *>(out)[i] = make_float4(g.x * m[0] * d(v.x), g.y * m[1] * d(v.y), g.z * m[2] * d(v.z),
                                                   g.w * m[3] * d(v.w));
}
}  // namespace
}  // namespace ptgnn_amd

extern "C" int ptgnn_amd_act_dropout_backward_f32(const float *grad, const float *y, const uint8_t *keep, float scale,
                                                  int act, int64_t n, float *out, void *stream_) {
  PTGNN_REQUIRE(n >= 0 && act >= PTGNN_AMD_ACT_NONE && act <= PTGNN_AMD_ACT_RELU, PTGNN_AMD_EINVAL,
                "act_dropout_backward: bad arguments");
  if (n == 0) return PTGNN_AMD_OK;
  PTGNN_REQUIRE(grad && y && out, PTGNN_AMD_EINVAL, "act_dropout_backward: null pointer");
  PTGNN_REQUIRE(n % 4 == 0 && aligned16(grad) && aligned16(y) && aligned16(out) && (!keep || ((uintptr_t)keep & 3u) == 0),
                PTGNN_AMD_EUNSUPPORTED, "act_dropout_backward: needs n %% 4 == 0 and 16-byte aligned arrays");
  const int64_t n4 = n / 4;
  PTGNN_REQUIRE((n4 + 255) / 256 < ((int64_t)1 << 31), PTGNN_AMD_EUNSUPPORTED, "act_dropout_backward: too many elements");
  const unsigned grid = (unsigned)((n4 + 255) / 256);
  hipStream_t st = (hipStream_t)stream_;
#define PTGNN_ADB(ACTV)                                                                                   \
  do {                                                                                                    \
    if (keep) k_act_dropout_backward<ACTV, true><<<grid, 256, 0, st>>>(grad, y, keep, scale, n4, out);    \
    else k_act_dropout_backward<ACTV, false><<<grid, 256, 0, st>>>(grad, y, keep, scale, n4, out);        \
  } while (0)
  if (act == PTGNN_AMD_ACT_TANH) PTGNN_ADB(PTGNN_AMD_ACT_TANH);
  else if (act == PTGNN_AMD_ACT_RELU) PTGNN_ADB(PTGNN_AMD_ACT_RELU);
  else PTGNN_ADB(PTGNN_AMD_ACT_NONE);
#undef PTGNN_ADB
  PTGNN_LAUNCH_CHECK();
  return PTGNN_AMD_OK;
}

extern "C" size_t ptgnn_amd_row_epilogue_workspace_bytes(int64_t rows, int32_t dim) {
  if (rows <= 0 || dim <= 0) return 0;
  return (size_t)kMaxBackwardBlocks * 2 * (size_t)dim * sizeof(float) + 256;
}

extern "C" int ptgnn_amd_row_epilogue_backward_f32(const float *x, int64_t ld_x, const float *grad_y, int64_t ld_gy,
                                                   int64_t rows, int32_t dim, int32_t flags, const float *ln_gamma,
                                                   float ln_eps, float *grad_x, int64_t ld_gx, float *grad_gamma,
                                                   float *grad_beta, void *workspace, size_t workspace_bytes,
                                                   void *stream_) {
  hipStream_t st = (hipStream_t)stream_;
  if (int rc = check_common("row_epilogue_backward", x, ld_x, rows, dim, flags, ln_gamma)) return rc;
  const bool ln = (flags & PTGNN_AMD_EPI_LAYERNORM) != 0;
  PTGNN_REQUIRE(rows == 0 || (grad_y && ld_gy >= dim && grad_x && ld_gx >= dim), PTGNN_AMD_EINVAL,
                "row_epilogue_backward: bad gradient matrices");
  PTGNN_REQUIRE(!ln || (grad_gamma && grad_beta), PTGNN_AMD_EINVAL, "row_epilogue_backward: LayerNorm needs grad_gamma / grad_beta");
  if (rows == 0) {
    if (ln) {
      PTGNN_HIP(hipMemsetAsync(grad_gamma, 0, sizeof(float) * dim, st));
      PTGNN_HIP(hipMemsetAsync(grad_beta, 0, sizeof(float) * dim, st));
    }
    return PTGNN_AMD_OK;
  }
  PTGNN_REQUIRE(!ln || (workspace && workspace_bytes >= ptgnn_amd_row_epilogue_workspace_bytes(rows, dim)),
                PTGNN_AMD_EWORKSPACE, "row_epilogue_backward: workspace %zu < %zu", workspace_bytes,
                ptgnn_amd_row_epilogue_workspace_bytes(rows, dim));
  RowArgs p{};
  p.x = x; p.ld_x = ld_x; p.dy = grad_y; p.ld_dy = ld_gy; p.out = grad_x; p.ld_out = ld_gx; p.gamma = ln_gamma;
  p.eps = ln_eps; p.rows = rows; p.M = dim; p.flags = flags;
  p.partial = ln ? (float *)(((uintptr_t)workspace + 255) & ~(uintptr_t)255) : nullptr;
  const bool vec4 = dim % 4 == 0 && ld_x % 4 == 0 && ld_gy % 4 == 0 && ld_gx % 4 == 0 && aligned16(x) &&
                    aligned16(grad_y) && aligned16(grad_x);
  int nblocks = 0;
  const int rc = dispatch(vec4, dim, [&](auto VEC, auto LPR, auto CH) -> int {
    constexpr int GR = kBlock / decltype(LPR)::value * rows_in_flight<decltype(LPR)::value>();
    nblocks = backward_blocks(rows, GR);
    k_row_epilogue_backward<decltype(VEC)::value, decltype(LPR)::value, decltype(CH)::value>
        <<<(unsigned)nblocks, kBlock, 0, st>>>(p);
    PTGNN_LAUNCH_CHECK();
    return (int)PTGNN_AMD_OK;
  });
  if (rc != PTGNN_AMD_OK || !ln) return rc;
  k_partial_rows_sum<<<(unsigned)((2 * dim + 63) / 64), 64 * kSumSlices, 0, st>>>(p.partial, nblocks, 2 * dim, grad_gamma,
                                                                        grad_beta, dim);
  PTGNN_LAUNCH_CHECK();
  return PTGNN_AMD_OK;
}
