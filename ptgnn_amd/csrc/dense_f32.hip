// Dense blocks of the message-passing layers on the matrix cores, exact fp32
// (v_mfma_f32_32x32x2_f32: bit-for-bit a k-ordered fmaf chain, 64 FLOP/clk/SIMD): the C-ABI entry points
//   ptgnn_amd_linear_f32   : y = act(x W^T + b)            (pre-transform, MLP update, residual mix)
//   ptgnn_amd_gru_cell_f32 : h' = GRUCell(a, h), gate GEMMs + gate math in one kernel
// and their 128 x 128 TILE kernels.  Contracts + reference lines: include/ptgnn_amd.h.
//
// Each entry point first offers the call to the streaming weight-stationary core (stream_gemm.hip: GEMM modes
// 1 and 2); the tile kernels below run GEMM mode 0 and every shape the streaming core does not take (K % 64 != 0,
// unaligned rows, slabs beyond LDS, few rows per wave).  One output tile per 256-thread workgroup (4 waves = one
// per SIMD), 3-4 workgroups per CU, XCD-swizzled tile order:
//   linear: 128 x 128 (or 128 x 64 for narrow outputs) tile, K chunks of 32 through LDS (row stride 33 floats ->
//           conflict-free ds_read_b32 for both MFMA operands), next chunk prefetched into registers under the
//           MFMAs; each wave owns 64 x 64 (64 x 32) = 2 x 2 (2 x 1) MFMA tiles; epilogue through an LDS slab.
//   gru   : 128 rows x 32 state features; each wave owns 32 rows and FOUR 32x32 accumulators (r, z, i_n, h_n)
//           whose C-fragment maps coincide, so the gate math is per-lane register arithmetic in the epilogue and
//           the [n, 3H] gate matrices never exist; the previous state is re-read (L2-hot) in the epilogue.
#include <stdlib.h>

#include <type_traits>

#include "dense_common.h"
#include "stream_gemm.h"

namespace ptgnn_amd {
namespace {

// ---------------------------------------------------------------------------------------------
// linear, one tile per workgroup: single LDS buffer, two barriers per K-chunk, next chunk prefetched into
// registers under the MFMAs; latency is hidden by 3-5 co-resident workgroups per CU and the hardware
// dispatcher balances the tail.
// ---------------------------------------------------------------------------------------------
template <bool ALIGNED, int ACT, int NJ>
__global__ __launch_bounds__(256, (NJ == 1 ? 4 : 3)) void k_linear_tlp(
    const float *__restrict__ x, int64_t rows, int K, int64_t ld_x, const float *__restrict__ w,
    int n_out, const float *__restrict__ bias, float *__restrict__ y, int64_t ld_y,
    int64_t num_tiles, int col_tiles, int vec_store) {
  constexpr int BN = 64 * NJ;
  constexpr int B_FLOATS = BN * LDS_LD;
  constexpr int SLAB_LD = 32 * NJ + 4;
  constexpr int SLAB_FLOATS = 32 * SLAB_LD;
  constexpr int OPER = TILE_FLOATS + B_FLOATS;
  constexpr int kLds = OPER > 4 * SLAB_FLOATS ? OPER : 4 * SLAB_FLOATS;
  __shared__ __attribute__((aligned(16))) float smem[kLds];
  float *const As = smem, *const Bs = smem + TILE_FLOATS;

  const int64_t tile = xcd_swizzle(blockIdx.x, gridDim.x);
  if (tile >= num_tiles) return;
  const int64_t row0 = (tile / col_tiles) * 128;
  const int col0 = (int)(tile % col_tiles) * BN;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, hi = lane >> 5;

  f32x16 acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  Stager<128, ALIGNED, RowClamp> sa;
  Stager<BN, ALIGNED, RowClamp> sb;
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
    const float *ap = As + (wm * 64 + li) * LDS_LD + 4 * hi;   // K order of a chunk: kcol()
    const float *bp = Bs + (wn * 32 * NJ + li) * LDS_LD + 4 * hi;   // K order of a chunk: kcol()
#pragma unroll
    for (int ks = 0; ks < BK / 2; ++ks) {
      float a[2], b[NJ];
      a[0] = ap[kcol(ks)];
      a[1] = ap[32 * LDS_LD + kcol(ks)];
#pragma unroll
      for (int j = 0; j < NJ; ++j) b[j] = bp[j * 32 * LDS_LD + kcol(ks)];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }

  __syncthreads();  // operand buffers are reused as the epilogue slabs
  float *const slab = smem + wave * SLAB_FLOATS;
  constexpr int LPRW = 8 * NJ;             // lanes per output row (float4 each)
  constexpr int RPI = 64 / LPRW;           // rows per wave-instruction
  constexpr int NIT = 32 / RPI;
  const int c4 = (lane % LPRW) * 4;
  const int rsub = lane / LPRW;
  const int gcol = col0 + wn * 32 * NJ + c4;
  // Interior tiles (all of them but the last row/column of tiles) take a straight-line path: the
  // guarded variant below compiles to a read -> wait -> ~8 branches -> store chain per float4 and
  // cost 22-27 % of the kernel (profiles/r01_notes.md).
  const bool interior = (vec_store & 1) && row0 + 128 <= rows && col0 + BN <= n_out &&
                        (bias == nullptr || (vec_store & 2));
  if (interior) {
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) bv = *reinterpret_cast<const float4 *>(bias + gcol);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          slab[((r & 3) + 8 * (r >> 2) + 4 * hi) * SLAB_LD + j * 32 + li] = acc[i][j][r];
      __builtin_amdgcn_wave_barrier();
      float4 v[NIT];
#pragma unroll
      for (int it = 0; it < NIT; ++it)
        v[it] = *reinterpret_cast<const float4 *>(slab + (it * RPI + rsub) * SLAB_LD + c4);
      __builtin_amdgcn_wave_barrier();
      float *dst = y + (row0 + wm * 64 + i * 32 + rsub) * ld_y + gcol;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        float4 o;
        o.x = act_apply<ACT>(v[it].x + bv.x);
        o.y = act_apply<ACT>(v[it].y + bv.y);
        o.z = act_apply<ACT>(v[it].z + bv.z);
        o.w = act_apply<ACT>(v[it].w + bv.w);
        *reinterpret_cast<float4 *>(dst + (int64_t)it * RPI * ld_y) = o;
      }
    }
    return;
  }
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (bias) {
    if (gcol + 0 < n_out) bv.x = bias[gcol + 0];
    if (gcol + 1 < n_out) bv.y = bias[gcol + 1];
    if (gcol + 2 < n_out) bv.z = bias[gcol + 2];
    if (gcol + 3 < n_out) bv.w = bias[gcol + 3];
  }
#pragma unroll 1
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        slab[((r & 3) + 8 * (r >> 2) + 4 * hi) * SLAB_LD + j * 32 + li] = i == 0 ? acc[0][j][r] : acc[1][j][r];
    __builtin_amdgcn_wave_barrier();
    const int64_t rbase = row0 + wm * 64 + i * 32;
#pragma unroll 1
    for (int it = 0; it < NIT; ++it) {
      const int rl = it * RPI + rsub;
      const int64_t grow = rbase + rl;
      float4 v = *reinterpret_cast<const float4 *>(slab + rl * SLAB_LD + c4);
      v.x = act_apply<ACT>(v.x + bv.x);
      v.y = act_apply<ACT>(v.y + bv.y);
      v.z = act_apply<ACT>(v.z + bv.z);
      v.w = act_apply<ACT>(v.w + bv.w);
      if (grow < rows) {
        float *dst = y + grow * ld_y + gcol;
        if ((vec_store & 1) && gcol + 3 < n_out) {
          *reinterpret_cast<float4 *>(dst) = v;
        } else {
          if (gcol + 0 < n_out) dst[0] = v.x;
          if (gcol + 1 < n_out) dst[1] = v.y;
          if (gcol + 2 < n_out) dst[2] = v.z;
          if (gcol + 3 < n_out) dst[3] = v.w;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------------------------
// fused GRU cell
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

template <bool ALIGNED>
__global__ __launch_bounds__(256, 3) void k_gru(const float *__restrict__ a, int64_t ld_a,
                                                const float *__restrict__ h, int64_t ld_h,
                                                const float *__restrict__ w_ih,
                                                const float *__restrict__ w_hh,
                                                const float *__restrict__ b_ih,
                                                const float *__restrict__ b_hh, int64_t n, int M,
                                                int H, float *__restrict__ out, int64_t ld_out,
                                                int64_t num_tiles, int col_tiles,
                                                float *__restrict__ gates) {
  constexpr int B_FLOATS = 96 * LDS_LD;
  __shared__ __attribute__((aligned(16))) float smem[TILE_FLOATS + B_FLOATS];
  float *const As = smem, *const Bs = smem + TILE_FLOATS;

  const int64_t tile = xcd_swizzle(blockIdx.x, gridDim.x);
  if (tile >= num_tiles) return;
  const int64_t row0 = (tile / col_tiles) * 128;
  const int j0 = (int)(tile % col_tiles) * 32;

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
  const float *arow = As + (wave * 32) * LDS_LD;
  const float *ap = arow + li * LDS_LD + 4 * hi;   // K order of a chunk: kcol()
  const float *bp = Bs + li * LDS_LD + 4 * hi;   // K order of a chunk: kcol()
  for (int c = 0; c < chunks0; ++c) {
    __syncthreads();
    sa.store(As);
    sb.store(Bs);
    __syncthreads();
    if (c + 1 < total) issue(c + 1);
#pragma unroll
    for (int ks = 0; ks < BK / 2; ++ks) {
      const float av = ap[kcol(ks)];
      const float br = bp[kcol(ks)], bz = bp[32 * LDS_LD + kcol(ks)], bn = bp[64 * LDS_LD + kcol(ks)];
      acc_r = __builtin_amdgcn_mfma_f32_32x32x2f32(av, br, acc_r, 0, 0, 0);
      acc_z = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bz, acc_z, 0, 0, 0);
      acc_in = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bn, acc_in, 0, 0, 0);
    }
  }
  for (int c = chunks0; c < total; ++c) {
    __syncthreads();
    sa.store(As);
    sb.store(Bs);
    __syncthreads();
    if (c + 1 < total) issue(c + 1);
#pragma unroll
    for (int ks = 0; ks < BK / 2; ++ks) {
      const float av = ap[kcol(ks)];
      const float br = bp[kcol(ks)], bz = bp[32 * LDS_LD + kcol(ks)], bn = bp[64 * LDS_LD + kcol(ks)];
      acc_r = __builtin_amdgcn_mfma_f32_32x32x2f32(av, br, acc_r, 0, 0, 0);
      acc_z = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bz, acc_z, 0, 0, 0);
      acc_hn = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bn, acc_hn, 0, 0, 0);
    }
  }

  // ---- gate math in registers (the four C fragments share one lane->element map) ----
  const int j = j0 + li;
  if (j >= H) return;
  const float bir = b_ih[j], biz = b_ih[H + j], bin = b_ih[2 * H + j];
  const float bhr = b_hh[j], bhz = b_hh[H + j], bhn = b_hh[2 * H + j];
  // the previous state of this lane's 16 output elements: re-read from global memory (L2-hot -- the K
  // loop just streamed these rows) instead of holding 16 more registers across the whole K loop; rows
  // past the end are clamped, their results are never stored
  float hprev[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    int64_t row = row0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
    row = row < n ? row : n - 1;
    hprev[r] = h[row * ld_h + j];
  }
  // torch.nn.GRUCell: gi = W_ih x + b_ih, gh = W_hh h + b_hh.
  // No mul+add contraction in the gate math: the 16 elements of a lane are 16 different ROWS, and the
  // compiler otherwise fuses some element pairs (packed fp32 ops) and not others, which makes a row's
  // result depend on where it sits in the tile (1 ulp) -- sharded and unsharded runs must agree bit for
  // bit.  Every op below is individually rounded, like the reference's torch ops.
  float res[16];
  {
#pragma clang fp contract(off)
    if (gates) {
      // training: keep r, z, n and gh_n = W_hn h + b_hn ([n, 4H], gate-major) for the backward
      // (ptgnn_amd_gru_cell_backward_gates_f32); same arithmetic as the inference branch below
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = row0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        const float rg_ = sigmoidf_((acc_r[r] + bir) + bhr);
        const float zg = sigmoidf_((acc_z[r] + biz) + bhz);
        const float hn = acc_hn[r] + bhn;
        const float ng = tanhf((acc_in[r] + bin) + rg_ * hn);
        res[r] = (1.0f - zg) * ng + zg * hprev[r];
        if (row < n) {
          float *gp = gates + row * (int64_t)(4 * H) + j;
          gp[0] = rg_; gp[H] = zg; gp[2 * H] = ng; gp[3 * H] = hn;
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float rg_ = sigmoidf_((acc_r[r] + bir) + bhr);
        const float zg = sigmoidf_((acc_z[r] + biz) + bhz);
        const float hn = acc_hn[r] + bhn;
        const float ng = tanhf((acc_in[r] + bin) + rg_ * hn);
        res[r] = (1.0f - zg) * ng + zg * hprev[r];
      }
    }
  }
  float *dst = out + (row0 + wave * 32 + 4 * hi) * ld_out + j;
  if (row0 + 128 <= n) {  // interior tile: straight-line stores (2 x 128 B per wave-instruction)
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[(int64_t)((r & 3) + 8 * (r >> 2)) * ld_out] = res[r];
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t row = row0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (row < n) out[row * ld_out + j] = res[r];
    }
  }
}

// Backward of the GRU gate math (everything between the two gate GEMMs and h'):
//   h' = (1 - z) n + z h,  n = tanh(gi_n + r gh_n),  r, z = sigmoid(gi + gh)
//   d_n = g (1 - z), d_z = g (h - n), d_h(direct) = g z, d_npre = d_n (1 - n^2),
//   d_gi = [d_r r(1-r), d_z z(1-z), d_npre],  d_gh = [same, same, d_npre r],  d_r = d_npre gh_n
// One thread per 4 hidden units of one row; 7 reads + 7 writes of H floats per row: HBM-bound.
__global__ __launch_bounds__(256) void k_gru_gates_backward(const float *__restrict__ g, int64_t ld_g,
                                                            const float *__restrict__ gates,
                                                            const float *__restrict__ h, int64_t ld_h,
                                                            int64_t n, int H, float *__restrict__ d_gi,
                                                            float *__restrict__ d_gh,
                                                            float *__restrict__ d_h) {
  const int q = H / 4;
  const int64_t total = n * q;
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / q;
    const int c = (int)(i - row * q) * 4;
    const float4 gv = *reinterpret_cast<const float4 *>(g + row * ld_g + c);
    const float4 hv = *reinterpret_cast<const float4 *>(h + row * ld_h + c);
    const float *gp = gates + row * (int64_t)(4 * H) + c;
    const float4 r = *reinterpret_cast<const float4 *>(gp);
    const float4 z = *reinterpret_cast<const float4 *>(gp + H);
    const float4 nn = *reinterpret_cast<const float4 *>(gp + 2 * H);
    const float4 hn = *reinterpret_cast<const float4 *>(gp + 3 * H);
    float4 o_r, o_z, o_n, o_hn, o_h;
#define PTGNN_GRU_BWD(C)                                         \
    {                                                            \
      const float dn = gv.C * (1.0f - z.C);                      \
      const float dz = gv.C * (hv.C - nn.C);                     \
      const float dnp = dn * (1.0f - nn.C * nn.C);               \
      o_h.C = gv.C * z.C;                                        \
      o_n.C = dnp;                                               \
      o_hn.C = dnp * r.C;                                        \
      o_r.C = (dnp * hn.C) * (r.C * (1.0f - r.C));               \
      o_z.C = dz * (z.C * (1.0f - z.C));                         \
    }
    PTGNN_GRU_BWD(x) PTGNN_GRU_BWD(y) PTGNN_GRU_BWD(z) PTGNN_GRU_BWD(w)
#undef PTGNN_GRU_BWD
    float *gi = d_gi + row * (int64_t)(3 * H) + c, *gh = d_gh + row * (int64_t)(3 * H) + c;
    *reinterpret_cast<float4 *>(gi) = o_r;
    *reinterpret_cast<float4 *>(gi + H) = o_z;
    *reinterpret_cast<float4 *>(gi + 2 * H) = o_n;
    *reinterpret_cast<float4 *>(gh) = o_r;
    *reinterpret_cast<float4 *>(gh + H) = o_z;
    *reinterpret_cast<float4 *>(gh + 2 * H) = o_hn;
    *reinterpret_cast<float4 *>(d_h + row * (int64_t)H + c) = o_h;
  }
}

}  // namespace

int num_compute_units() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}

// PTGNN_AMD_LINEAR_NJ=1|2 forces the 128x64 | 128x128 tile (default: by n_out)
static int linear_nj() {  // 0 = heuristic
  static int nj = -1;
  if (nj < 0) {
    const char *e = getenv("PTGNN_AMD_LINEAR_NJ");
    nj = e ? atoi(e) : 0;
    if (nj < 0 || nj > 2) nj = 0;
  }
  return nj;
}

}  // namespace ptgnn_amd

using namespace ptgnn_amd;

extern "C" int ptgnn_amd_linear_f32(const float *x, int64_t rows, int32_t k, int64_t ld_x,
                                    const float *w, int32_t n_out, const float *bias, int act,
                                    float *y, int64_t ld_y, void *stream_) {
  PTGNN_REQUIRE(rows >= 0 && k > 0 && n_out > 0, PTGNN_AMD_EINVAL, "linear: bad sizes");
  PTGNN_REQUIRE(act >= 0 && act <= PTGNN_AMD_ACT_RELU, PTGNN_AMD_EINVAL, "linear: bad act");
  if (rows == 0) return PTGNN_AMD_OK;
  PTGNN_REQUIRE(x && w && y && ld_x >= k && ld_y >= n_out, PTGNN_AMD_EINVAL, "linear: null/ld");
  if (stream_linear(x, rows, k, ld_x, w, n_out, bias, act, y, ld_y, (hipStream_t)stream_)) {
    PTGNN_LAUNCH_CHECK();
    return PTGNN_AMD_OK;
  }
  int nj = linear_nj();
  if (nj == 0) nj = n_out <= 128 ? 1 : 2;  // narrow outputs: finer tiles, more workgroups per CU
  const int bn = 64 * nj;
  const int64_t row_tiles = (rows + 127) / 128;
  const int col_tiles = (n_out + bn - 1) / bn;
  const int64_t num_tiles = row_tiles * col_tiles;
  PTGNN_REQUIRE(num_tiles < ((int64_t)1 << 31), PTGNN_AMD_EUNSUPPORTED, "linear: too many tiles");
  const unsigned grid = (unsigned)xcd_padded_blocks(num_tiles);
  const bool al = (k % 4 == 0) && (ld_x % 4 == 0) && aligned16(x) && aligned16(w);
  const int vec_store = ((ld_y % 4 == 0) && aligned16(y) ? 1 : 0) | ((bias && aligned16(bias) && n_out % 4 == 0) ? 2 : 0);
  hipStream_t st = (hipStream_t)stream_;
#define PTGNN_LINEAR_LAUNCH(AL, ACT, NJ)                                                         \
  k_linear_tlp<AL, ACT, NJ><<<grid, 256, 0, st>>>(x, rows, k, ld_x, w, n_out, bias, y, ld_y,     \
                                                  num_tiles, col_tiles, vec_store)
#define PTGNN_LINEAR_ACT(AL, NJ)                                                          \
  do {                                                                                    \
    if (act == PTGNN_AMD_ACT_TANH) PTGNN_LINEAR_LAUNCH(AL, PTGNN_AMD_ACT_TANH, NJ);       \
    else if (act == PTGNN_AMD_ACT_RELU) PTGNN_LINEAR_LAUNCH(AL, PTGNN_AMD_ACT_RELU, NJ);  \
    else PTGNN_LINEAR_LAUNCH(AL, PTGNN_AMD_ACT_NONE, NJ);                                 \
  } while (0)
  if (al) { if (nj == 1) PTGNN_LINEAR_ACT(true, 1); else PTGNN_LINEAR_ACT(true, 2); }
  else    { if (nj == 1) PTGNN_LINEAR_ACT(false, 1); else PTGNN_LINEAR_ACT(false, 2); }
#undef PTGNN_LINEAR_ACT
#undef PTGNN_LINEAR_LAUNCH
  PTGNN_LAUNCH_CHECK();
  count_launch(PTGNN_AMD_KERNEL_TILE_LINEAR);
  return PTGNN_AMD_OK;
}

// y = act(x W^T + b) + addend in ONE launch: the streaming kernels add the block in their store epilogue.  Shapes the
// streaming core does not take answer EUNSUPPORTED (the host then runs ptgnn_amd_linear_f32 and adds).
extern "C" int ptgnn_amd_linear_add_f32(const float *x, int64_t rows, int32_t k, int64_t ld_x, const float *w,
                                        int32_t n_out, const float *bias, int act, const float *addend,
                                        int64_t ld_add, float *y, int64_t ld_y, void *stream_) {
  PTGNN_REQUIRE(rows >= 0 && k > 0 && n_out > 0, PTGNN_AMD_EINVAL, "linear_add: bad sizes");
  PTGNN_REQUIRE(act >= 0 && act <= PTGNN_AMD_ACT_RELU, PTGNN_AMD_EINVAL, "linear_add: bad act");
  if (rows == 0) return PTGNN_AMD_OK;
  PTGNN_REQUIRE(x && w && y && addend && ld_x >= k && ld_y >= n_out && ld_add >= n_out, PTGNN_AMD_EINVAL,
                "linear_add: null/ld");
  PTGNN_REQUIRE(addend != y, PTGNN_AMD_EINVAL, "linear_add: in-place accumulation is not supported");
  if (stream_linear(x, rows, k, ld_x, w, n_out, bias, act, y, ld_y, (hipStream_t)stream_, addend, ld_add)) {
    PTGNN_LAUNCH_CHECK();
    return PTGNN_AMD_OK;
  }
  set_error("linear_add: rows=%lld k=%d n_out=%d is not a shape of the streaming GEMM (use ptgnn_amd_linear_f32 and add)",
            (long long)rows, k, n_out);
  return PTGNN_AMD_EUNSUPPORTED;
}

static int gru_launch(const float *a, int64_t ld_a, const float *h, int64_t ld_h, const float *w_ih,
                      const float *w_hh, const float *b_ih, const float *b_hh, int64_t n, int32_t m,
                      int32_t hd, float *out, int64_t ld_out, float *gates, void *stream_) {
  PTGNN_REQUIRE(n >= 0 && m > 0 && hd > 0, PTGNN_AMD_EINVAL, "gru_cell: bad sizes");
  if (n == 0) return PTGNN_AMD_OK;
  PTGNN_REQUIRE(a && h && w_ih && w_hh && b_ih && b_hh && out, PTGNN_AMD_EINVAL, "gru_cell: null");
  PTGNN_REQUIRE(ld_a >= m && ld_h >= hd && ld_out >= hd, PTGNN_AMD_EINVAL, "gru_cell: bad ld");
  PTGNN_REQUIRE(out != h, PTGNN_AMD_EINVAL, "gru_cell: in-place update is not supported");
  if (stream_gru(a, ld_a, h, ld_h, w_ih, w_hh, b_ih, b_hh, n, m, hd, out, ld_out, gates, (hipStream_t)stream_)) {
    PTGNN_LAUNCH_CHECK();
    return PTGNN_AMD_OK;
  }
  const int64_t row_tiles = (n + 127) / 128;
  const int col_tiles = (hd + 31) / 32;
  const int64_t num_tiles = row_tiles * col_tiles;
  PTGNN_REQUIRE(num_tiles < ((int64_t)1 << 31), PTGNN_AMD_EUNSUPPORTED, "gru_cell: too many tiles");
  const unsigned grid = (unsigned)xcd_padded_blocks(num_tiles);
  const bool al = (m % 4 == 0) && (hd % 4 == 0) && (ld_a % 4 == 0) && (ld_h % 4 == 0) &&
                  aligned16(a) && aligned16(h) && aligned16(w_ih) && aligned16(w_hh);
  if (al)
    k_gru<true><<<grid, 256, 0, (hipStream_t)stream_>>>(a, ld_a, h, ld_h, w_ih, w_hh, b_ih, b_hh, n, m, hd,
                                                        out, ld_out, num_tiles, col_tiles, gates);
  else
    k_gru<false><<<grid, 256, 0, (hipStream_t)stream_>>>(a, ld_a, h, ld_h, w_ih, w_hh, b_ih, b_hh, n, m, hd,
                                                         out, ld_out, num_tiles, col_tiles, gates);
  PTGNN_LAUNCH_CHECK();
  count_launch(PTGNN_AMD_KERNEL_TILE_GRU);
  return PTGNN_AMD_OK;
}

extern "C" int ptgnn_amd_gru_cell_f32(const float *a, int64_t ld_a, const float *h, int64_t ld_h,
                                      const float *w_ih, const float *w_hh, const float *b_ih,
                                      const float *b_hh, int64_t n, int32_t m, int32_t hd,
                                      float *out, int64_t ld_out, void *stream_) {
  return gru_launch(a, ld_a, h, ld_h, w_ih, w_hh, b_ih, b_hh, n, m, hd, out, ld_out, nullptr, stream_);
}

extern "C" int ptgnn_amd_gru_cell_train_f32(const float *a, int64_t ld_a, const float *h, int64_t ld_h,
                                            const float *w_ih, const float *w_hh, const float *b_ih,
                                            const float *b_hh, int64_t n, int32_t m, int32_t hd,
                                            float *out, int64_t ld_out, float *gates, void *stream_) {
  PTGNN_REQUIRE(gates != nullptr || n == 0, PTGNN_AMD_EINVAL, "gru_cell_train: gates is null");
  return gru_launch(a, ld_a, h, ld_h, w_ih, w_hh, b_ih, b_hh, n, m, hd, out, ld_out, gates, stream_);
}

extern "C" int ptgnn_amd_gru_cell_backward_gates_f32(const float *grad_out, int64_t ld_grad_out,
                                                     const float *gates, const float *h, int64_t ld_h,
                                                     int64_t n, int32_t hd, float *d_gi, float *d_gh,
                                                     float *d_h, void *stream_) {
  PTGNN_REQUIRE(n >= 0 && hd > 0, PTGNN_AMD_EINVAL, "gru_cell_backward_gates: bad sizes");
  if (n == 0) return PTGNN_AMD_OK;
  PTGNN_REQUIRE(grad_out && gates && h && d_gi && d_gh && d_h, PTGNN_AMD_EINVAL,
                "gru_cell_backward_gates: null pointer");
  PTGNN_REQUIRE(hd % 4 == 0 && ld_grad_out % 4 == 0 && ld_h % 4 == 0 && ld_grad_out >= hd && ld_h >= hd &&
                    aligned16(grad_out) && aligned16(gates) && aligned16(h) && aligned16(d_gi) &&
                    aligned16(d_gh) && aligned16(d_h),
                PTGNN_AMD_EUNSUPPORTED, "gru_cell_backward_gates: needs hd %% 4 == 0 and 16-byte aligned rows");
  const int64_t items = n * (hd / 4);
  int64_t blocks = (items + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  k_gru_gates_backward<<<(unsigned)blocks, 256, 0, (hipStream_t)stream_>>>(grad_out, ld_grad_out, gates, h,
                                                                          ld_h, n, hd, d_gi, d_gh, d_h);
  PTGNN_LAUNCH_CHECK();
  return PTGNN_AMD_OK;
}
