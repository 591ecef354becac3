// Dense blocks of the message-passing layers on the matrix cores, exact fp32
// (v_mfma_f32_32x32x2_f32: bit-for-bit a k-ordered fmaf chain, 64 FLOP/clk/SIMD).
//   ptgnn_amd_linear_f32   : y = act(x W^T + b)            (pre-transform, MLP update, residual mix)
//   ptgnn_amd_gru_cell_f32 : h' = GRUCell(a, h), gate GEMMs + gate math in one kernel
// Contracts + reference lines: include/ptgnn_amd.h.
//
// Tiling (per 256-thread workgroup = 4 waves, one per SIMD):
//   linear: 128 x 128 output tile, K in chunks of 32 through LDS (row stride 33 floats ->
//           conflict-free ds_read_b32 for both MFMA operands), each wave owns 64 x 64 =
//           2 x 2 MFMA tiles (64 accumulator VGPRs); next chunk's global loads are issued before
//           the current chunk's MFMAs (register-staged software pipeline).
//   gru   : 128 rows x 32 state features; each wave owns 32 rows and FOUR 32x32 accumulators
//           (r, z, i_n, h_n) whose C-fragment maps coincide, so the gate math is per-lane
//           register arithmetic in the epilogue and the [n, 3H] gate matrices never exist.
// fp32 MFMA is 1/16 of the bf16 rate, so LDS/global traffic is far from limiting: MFMA-bound.
#include "common.h"

namespace ptgnn_amd {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int BK = 32;
constexpr int LDS_LD = BK + 1;

// Stage a [ROWS x 32] K-chunk of a row-major matrix into registers (ROWS*32/256 floats/thread).
// RowMap maps tile row -> matrix row (or -1 for "zero row").
template <int ROWS, bool ALIGNED, typename RowMap>
struct Stager {
  static constexpr int NV4 = ROWS * BK / 4 / 256;  // float4 per thread
  float4 v[NV4];

  __device__ __forceinline__ void load(const float *__restrict__ base, int64_t ld, int k0, int K,
                                       RowMap rm) {
#pragma unroll
    for (int r = 0; r < NV4; ++r) {
      const int f = threadIdx.x + r * 256;
      const int row = f >> 3, c4 = (f & 7) * 4;
      const int64_t mrow = rm(row);
      const int kk = k0 + c4;
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      if (mrow >= 0) {
        const float *p = base + mrow * ld + kk;
        if (ALIGNED) {
          if (kk < K) t = *reinterpret_cast<const float4 *>(p);  // K % 4 == 0 => whole float4 valid
        } else {
          if (kk + 0 < K) t.x = p[0];
          if (kk + 1 < K) t.y = p[1];
          if (kk + 2 < K) t.z = p[2];
          if (kk + 3 < K) t.w = p[3];
        }
      }
      v[r] = t;
    }
  }

  __device__ __forceinline__ void store(float *__restrict__ lds) const {
#pragma unroll
    for (int r = 0; r < NV4; ++r) {
      const int f = threadIdx.x + r * 256;
      const int row = f >> 3, c4 = (f & 7) * 4;
      float *q = lds + row * LDS_LD + c4;
      q[0] = v[r].x; q[1] = v[r].y; q[2] = v[r].z; q[3] = v[r].w;
    }
  }
};

struct RowClamp {  // plain matrices: tile row -> base_row + row if < limit
  int64_t base, limit;
  __device__ __forceinline__ int64_t operator()(int row) const {
    const int64_t r = base + row;
    return r < limit ? r : -1;
  }
};

struct GateRows {  // GRU weights: tile row (gate*32 + jj) -> gate*H + j0 + jj
  int j0, H;
  __device__ __forceinline__ int64_t operator()(int row) const {
    const int gate = row >> 5, j = j0 + (row & 31);
    return j < H ? (int64_t)gate * H + j : -1;
  }
};

__device__ __forceinline__ float act_apply(float v, int act) {
  if (act == PTGNN_AMD_ACT_TANH) return tanhf(v);
  if (act == PTGNN_AMD_ACT_RELU) return v > 0.f ? v : 0.f;
  return v;
}

// ---------------------------------------------------------------------------------------------
// linear
// ---------------------------------------------------------------------------------------------
template <bool ALIGNED>
__global__ __launch_bounds__(256, 2) void k_linear(const float *__restrict__ x, int64_t rows, int K,
                                                   int64_t ld_x, const float *__restrict__ w,
                                                   int n_out, const float *__restrict__ bias,
                                                   int act, float *__restrict__ y, int64_t ld_y,
                                                   int64_t row_tiles, int col_tiles) {
  __shared__ float As[128 * LDS_LD];
  __shared__ float Bs[128 * LDS_LD];

  const int64_t tile = xcd_swizzle(blockIdx.x, gridDim.x);
  if (tile >= row_tiles * col_tiles) return;
  const int64_t rt = tile / col_tiles;
  const int ct = (int)(tile % col_tiles);
  const int64_t row0 = rt * 128;
  const int col0 = ct * 128;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, hi = lane >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  Stager<128, ALIGNED, RowClamp> sa, sb;
  const RowClamp ra{row0, rows}, rb{col0, n_out};
  const int nchunks = (K + BK - 1) / BK;
  sa.load(x, ld_x, 0, K, ra);
  sb.load(w, K, 0, K, rb);
  for (int c = 0; c < nchunks; ++c) {
    __syncthreads();  // previous chunk's readers are done
    sa.store(As);
    sb.store(Bs);
    __syncthreads();
    if (c + 1 < nchunks) {
      sa.load(x, ld_x, (c + 1) * BK, K, ra);
      sb.load(w, K, (c + 1) * BK, K, rb);
    }
    const float *ap = As + (wm * 64 + li) * LDS_LD + hi;
    const float *bp = Bs + (wn * 64 + li) * LDS_LD + hi;
#pragma unroll
    for (int ks = 0; ks < BK / 2; ++ks) {
      const float a0 = ap[ks * 2], a1 = ap[32 * LDS_LD + ks * 2];
      const float b0 = bp[ks * 2], b1 = bp[32 * LDS_LD + ks * 2];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
  }

  // C fragment: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = col0 + wn * 64 + j * 32 + li;
    if (col >= n_out) continue;
    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = row0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (row < rows) y[row * ld_y + col] = act_apply(acc[i][j][r] + bv, act);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// fused GRU cell
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

template <bool ALIGNED>
__global__ __launch_bounds__(256, 2) void k_gru(const float *__restrict__ a, int64_t ld_a,
                                                const float *__restrict__ h, int64_t ld_h,
                                                const float *__restrict__ w_ih,
                                                const float *__restrict__ w_hh,
                                                const float *__restrict__ b_ih,
                                                const float *__restrict__ b_hh, int64_t n, int M,
                                                int H, float *__restrict__ out, int64_t ld_out,
                                                int64_t row_tiles, int col_tiles) {
  __shared__ float As[128 * LDS_LD];
  __shared__ float Bs[96 * LDS_LD];

  const int64_t tile = xcd_swizzle(blockIdx.x, gridDim.x);
  if (tile >= row_tiles * col_tiles) return;
  const int64_t rt = tile / col_tiles;
  const int ct = (int)(tile % col_tiles);
  const int64_t row0 = rt * 128;
  const int j0 = ct * 32;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, hi = lane >> 5;

  f32x16 acc_r, acc_z, acc_in, acc_hn;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc_r[r] = 0.f; acc_z[r] = 0.f; acc_in[r] = 0.f; acc_hn[r] = 0.f; }

  Stager<128, ALIGNED, RowClamp> sa;
  Stager<96, ALIGNED, GateRows> sb;
  const RowClamp ra{row0, n};
  const GateRows rg{j0, H};

  // phase 0: K over the aggregated messages (a, W_ih) -> r, z, i_n
  // phase 1: K over the previous state     (h, W_hh) -> r, z, h_n
  const int chunks0 = (M + BK - 1) / BK, chunks1 = (H + BK - 1) / BK;
  const int total = chunks0 + chunks1;
  auto issue = [&](int c) {
    if (c < chunks0) {
      sa.load(a, ld_a, c * BK, M, ra);
      sb.load(w_ih, M, c * BK, M, rg);
    } else {
      sa.load(h, ld_h, (c - chunks0) * BK, H, ra);
      sb.load(w_hh, H, (c - chunks0) * BK, H, rg);
    }
  };
  issue(0);
  for (int c = 0; c < total; ++c) {
    __syncthreads();
    sa.store(As);
    sb.store(Bs);
    __syncthreads();
    if (c + 1 < total) issue(c + 1);
    const float *ap = As + (wave * 32 + li) * LDS_LD + hi;
    const float *bp = Bs + li * LDS_LD + hi;
    if (c < chunks0) {
#pragma unroll
      for (int ks = 0; ks < BK / 2; ++ks) {
        const float av = ap[ks * 2];
        const float br = bp[ks * 2], bz = bp[32 * LDS_LD + ks * 2], bn = bp[64 * LDS_LD + ks * 2];
        acc_r = __builtin_amdgcn_mfma_f32_32x32x2f32(av, br, acc_r, 0, 0, 0);
        acc_z = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bz, acc_z, 0, 0, 0);
        acc_in = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bn, acc_in, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < BK / 2; ++ks) {
        const float av = ap[ks * 2];
        const float br = bp[ks * 2], bz = bp[32 * LDS_LD + ks * 2], bn = bp[64 * LDS_LD + ks * 2];
        acc_r = __builtin_amdgcn_mfma_f32_32x32x2f32(av, br, acc_r, 0, 0, 0);
        acc_z = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bz, acc_z, 0, 0, 0);
        acc_hn = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bn, acc_hn, 0, 0, 0);
      }
    }
  }

  const int j = j0 + li;
  if (j >= H) return;
  const float bir = b_ih[j], biz = b_ih[H + j], bin = b_ih[2 * H + j];
  const float bhr = b_hh[j], bhz = b_hh[H + j], bhn = b_hh[2 * H + j];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t row = row0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
    if (row >= n) continue;
    // torch.nn.GRUCell: gi = W_ih x + b_ih, gh = W_hh h + b_hh
    const float rg_ = sigmoidf_((acc_r[r] + bir) + bhr);
    const float zg = sigmoidf_((acc_z[r] + biz) + bhz);
    const float ng = tanhf((acc_in[r] + bin) + rg_ * (acc_hn[r] + bhn));
    const float hv = h[row * ld_h + j];
    out[row * ld_out + j] = (1.0f - zg) * ng + zg * hv;
  }
}

}  // namespace
}  // namespace ptgnn_amd

using namespace ptgnn_amd;

extern "C" int ptgnn_amd_linear_f32(const float *x, int64_t rows, int32_t k, int64_t ld_x,
                                    const float *w, int32_t n_out, const float *bias, int act,
                                    float *y, int64_t ld_y, void *stream_) {
  PTGNN_REQUIRE(rows >= 0 && k > 0 && n_out > 0, PTGNN_AMD_EINVAL, "linear: bad sizes");
  PTGNN_REQUIRE(act >= 0 && act <= PTGNN_AMD_ACT_RELU, PTGNN_AMD_EINVAL, "linear: bad act");
  if (rows == 0) return PTGNN_AMD_OK;
  PTGNN_REQUIRE(x && w && y && ld_x >= k && ld_y >= n_out, PTGNN_AMD_EINVAL, "linear: null/ld");
  const int64_t row_tiles = (rows + 127) / 128;
  const int col_tiles = (n_out + 127) / 128;
  const int64_t nblk = xcd_padded_blocks(row_tiles * col_tiles);
  PTGNN_REQUIRE(nblk < ((int64_t)1 << 31), PTGNN_AMD_EUNSUPPORTED, "linear: grid too large");
  const bool al = (k % 4 == 0) && (ld_x % 4 == 0) && aligned16(x) && aligned16(w);
  if (al)
    k_linear<true><<<(unsigned)nblk, 256, 0, (hipStream_t)stream_>>>(x, rows, k, ld_x, w, n_out, bias,
                                                                     act, y, ld_y, row_tiles, col_tiles);
  else
    k_linear<false><<<(unsigned)nblk, 256, 0, (hipStream_t)stream_>>>(x, rows, k, ld_x, w, n_out, bias,
                                                                      act, y, ld_y, row_tiles, col_tiles);
  PTGNN_LAUNCH_CHECK();
  return PTGNN_AMD_OK;
}

extern "C" int ptgnn_amd_gru_cell_f32(const float *a, int64_t ld_a, const float *h, int64_t ld_h,
                                      const float *w_ih, const float *w_hh, const float *b_ih,
                                      const float *b_hh, int64_t n, int32_t m, int32_t hd,
                                      float *out, int64_t ld_out, void *stream_) {
  PTGNN_REQUIRE(n >= 0 && m > 0 && hd > 0, PTGNN_AMD_EINVAL, "gru_cell: bad sizes");
  if (n == 0) return PTGNN_AMD_OK;
  PTGNN_REQUIRE(a && h && w_ih && w_hh && b_ih && b_hh && out, PTGNN_AMD_EINVAL, "gru_cell: null");
  PTGNN_REQUIRE(ld_a >= m && ld_h >= hd && ld_out >= hd, PTGNN_AMD_EINVAL, "gru_cell: bad ld");
  PTGNN_REQUIRE(out != h, PTGNN_AMD_EINVAL, "gru_cell: in-place update is not supported");
  const int64_t row_tiles = (n + 127) / 128;
  const int col_tiles = (hd + 31) / 32;
  const int64_t nblk = xcd_padded_blocks(row_tiles * col_tiles);
  PTGNN_REQUIRE(nblk < ((int64_t)1 << 31), PTGNN_AMD_EUNSUPPORTED, "gru_cell: grid too large");
  const bool al = (m % 4 == 0) && (hd % 4 == 0) && (ld_a % 4 == 0) && (ld_h % 4 == 0) &&
                  aligned16(a) && aligned16(h) && aligned16(w_ih) && aligned16(w_hh);
  if (al)
    k_gru<true><<<(unsigned)nblk, 256, 0, (hipStream_t)stream_>>>(a, ld_a, h, ld_h, w_ih, w_hh, b_ih, b_hh,
                                                                  n, m, hd, out, ld_out, row_tiles, col_tiles);
  else
    k_gru<false><<<(unsigned)nblk, 256, 0, (hipStream_t)stream_>>>(a, ld_a, h, ld_h, w_ih, w_hh, b_ih, b_hh,
                                                                   n, m, hd, out, ld_out, row_tiles, col_tiles);
  PTGNN_LAUNCH_CHECK();
  return PTGNN_AMD_OK;
}
