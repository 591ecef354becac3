// EXPERIMENT (not part of libptgnn_amd): y = x W^T with every fp32 operand split EXACTLY into three bf16
// pieces (8 + 8 + 8 significand bits) and the product formed from the six largest piece products on the
// bf16 MFMA (v_mfma_f32_32x32x16_bf16, fp32 accumulate):
//     a b ~= a1 b1 + a1 b2 + a2 b1 + a1 b3 + a2 b2 + a3 b1        (dropped terms <= 2^-24 |a b|)
// Question for round 2: how fast is this against the exact-fp32 MFMA kernel, and how
// accurate against float64?  Restricted shape: rows % 128 == 0, K % 32 == 0, n_out % 128 == 0.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

namespace {
constexpr int BK = 32;            // fp32 columns per K chunk
constexpr int LDK = BK + 8;       // bf16 per LDS row (80 B: 16-byte aligned, staggers the banks)
constexpr int PIECE = 128 * LDK;  // bf16 elements of one piece of one operand tile

__device__ __forceinline__ uint32_t fbits(float v) { return __float_as_uint(v); }

// x -> (hi, mid, lo) with hi + mid + lo == x exactly; each piece has <= 8 significand bits (a bf16)
__device__ __forceinline__ void split3(float x, uint32_t &h, uint32_t &m, uint32_t &l) {
  const uint32_t hb = fbits(x) & 0xFFFF0000u;
  const float r1 = x - __uint_as_float(hb);
  const uint32_t mb = fbits(r1) & 0xFFFF0000u;
  const float r2 = r1 - __uint_as_float(mb);
  h = hb >> 16; m = mb >> 16; l = fbits(r2) >> 16;
}

// four consecutive fp32 -> three 8-byte groups of four bf16, written to the three piece tiles
__device__ __forceinline__ void stage4(const float4 v, uint16_t *tile, int row, int k) {
  uint32_t h[4], m[4], l[4];
  split3(v.x, h[0], m[0], l[0]); split3(v.y, h[1], m[1], l[1]);
  split3(v.z, h[2], m[2], l[2]); split3(v.w, h[3], m[3], l[3]);
  uint16_t *p = tile + row * LDK + k;
  *reinterpret_cast<uint2 *>(p) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
  *reinterpret_cast<uint2 *>(p + PIECE) = make_uint2(m[0] | (m[1] << 16), m[2] | (m[3] << 16));
  *reinterpret_cast<uint2 *>(p + 2 * PIECE) = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
}

__global__ __launch_bounds__(256, 2) void k_linear_split(const float *__restrict__ x, int64_t rows, int K,
                                                         const float *__restrict__ w, int n_out,
                                                         float *__restrict__ y, int terms) {
  __shared__ __attribute__((aligned(16))) uint16_t As[3 * PIECE];
  __shared__ __attribute__((aligned(16))) uint16_t Bs[3 * PIECE];
  const int col_tiles = n_out / 128;
  const int64_t tile = blockIdx.x;
  const int64_t row0 = (tile / col_tiles) * 128;
  const int col0 = (int)(tile % col_tiles) * 128;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, hi = lane >> 5;

  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 va[4], vb[4];
  auto issue = [&](int c) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = threadIdx.x + r * 256;
      const int row = f >> 3, c4 = (f & 7) * 4;
      va[r] = *reinterpret_cast<const float4 *>(x + (row0 + row) * K + c * BK + c4);
      vb[r] = *reinterpret_cast<const float4 *>(w + (int64_t)(col0 + row) * K + c * BK + c4);
    }
  };
  const int nchunks = K / BK;
  issue(0);
  for (int c = 0; c < nchunks; ++c) {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = threadIdx.x + r * 256;
      const int row = f >> 3, c4 = (f & 7) * 4;
      stage4(va[r], As, row, c4);
      stage4(vb[r], Bs, row, c4);
    }
    __syncthreads();
    if (c + 1 < nchunks) issue(c + 1);
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8 a[2][3], b[2][3];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          a[i][p] = *reinterpret_cast<const bf16x8 *>(As + p * PIECE + (wm * 64 + i * 32 + li) * LDK + ks * 16 + hi * 8);
          b[i][p] = *reinterpret_cast<const bf16x8 *>(Bs + p * PIECE + (wn * 64 + i * 32 + li) * LDK + ks * 16 + hi * 8);
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          // smallest terms first
          if (terms >= 6) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], acc[i][j], 0, 0, 0);
          }
          if (terms >= 3) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
        }
    }
  }
  // C fragment: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = row0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        y[row * n_out + col0 + wn * 64 + j * 32 + li] = acc[i][j][r];
      }
}
}  // namespace

extern "C" int split_bf16_linear(const float *x, int64_t rows, int k, const float *w, int n_out, float *y,
                                 int terms, void *stream) {
  if (rows % 128 || k % 32 || n_out % 128) return -1;
  const int64_t tiles = rows / 128 * (n_out / 128);
  k_linear_split<<<(unsigned)tiles, 256, 0, (hipStream_t)stream>>>(x, rows, k, w, n_out, y, terms);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
