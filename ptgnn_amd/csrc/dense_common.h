// Shared device helpers of the fp32-MFMA kernels (dense_f32.hip, edge_gemm.hip).
#pragma once
#include "common.h"

namespace ptgnn_amd {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int BK = 32;
constexpr int LDS_LD = BK + 1;

// Stage a [ROWS x 32] K-chunk of a row-major matrix through registers into LDS, one float4 "part"
// at a time so the traffic can be threaded between MFMAs.  RowMap maps (tile row, part) -> matrix row and
// CLAMPS it into range: out-of-range tile rows read some valid row instead of being predicated,
// because a branch around a load makes hipcc's waitcnt pass fall back to vmcnt(0) drains.  Such
// rows only feed output rows/columns that are never stored.  The K tail must be exact, so chunk
// columns >= K are zeroed on the way into LDS.
template <int ROWS, bool ALIGNED, typename RowMap>
struct Stager {
  static constexpr int NV4 = ROWS * BK / 4 / 256;  // float4 per thread
  float4 v[NV4];
  int kvalid;  // how many of this thread's 4 chunk columns are < K (0..4)

  __device__ __forceinline__ void load_part(const float *__restrict__ base, int64_t ld, int k0, int K,
                                            RowMap rm, int r) {
    const int f = threadIdx.x + r * 256;
    const int row = f >> 3, c4 = (f & 7) * 4;
    const int64_t mrow = rm(row, r);
    const int kk = k0 + c4;
    kvalid = K - kk;
    if (ALIGNED) {
      // K % 4 == 0: a float4 is entirely valid or entirely past the end
      const int kc = kk < K ? kk : 0;
      v[r] = *reinterpret_cast<const float4 *>(base + mrow * ld + kc);
    } else {
      const float *p = base + mrow * ld;
      float4 t;
      t.x = p[kk + 0 < K ? kk + 0 : 0];
      t.y = p[kk + 1 < K ? kk + 1 : 0];
      t.z = p[kk + 2 < K ? kk + 2 : 0];
      t.w = p[kk + 3 < K ? kk + 3 : 0];
      v[r] = t;
    }
  }

  __device__ __forceinline__ void store_part(float *__restrict__ lds, int r) const {
    const int f = threadIdx.x + r * 256;
    const int row = f >> 3, c4 = (f & 7) * 4;
    float *q = lds + row * LDS_LD + c4;
    q[0] = kvalid > 0 ? v[r].x : 0.f;
    q[1] = kvalid > 1 ? v[r].y : 0.f;
    q[2] = kvalid > 2 ? v[r].z : 0.f;
    q[3] = kvalid > 3 ? v[r].w : 0.f;
  }

  __device__ __forceinline__ void load(const float *__restrict__ base, int64_t ld, int k0, int K, RowMap rm) {
#pragma unroll
    for (int r = 0; r < NV4; ++r) load_part(base, ld, k0, K, rm, r);
  }

  __device__ __forceinline__ void store(float *__restrict__ lds) const {
#pragma unroll
    for (int r = 0; r < NV4; ++r) store_part(lds, r);
  }
};

struct RowClamp {  // plain matrices: tile row -> min(base_row + row, limit - 1)
  int64_t base, limit;
  __device__ __forceinline__ int64_t operator()(int row, int /*part*/) const {
    const int64_t r = base + row;
    return r < limit ? r : limit - 1;
  }
};

struct GateRows {  // GRU weights: tile row (gate*32 + jj) -> gate*H + min(j0 + jj, H - 1)
  int j0, H;
  __device__ __forceinline__ int64_t operator()(int row, int /*part*/) const {
    const int gate = row >> 5, j = j0 + (row & 31);
    return (int64_t)gate * H + (j < H ? j : H - 1);
  }
};

// K order inside a 32-wide chunk, shared by EVERY exact-fp32 kernel of the library (tile and streaming):
// MFMA step s of a chunk multiplies columns 8 (s >> 2) + (s & 3) (lanes 0-31) and that + 4 (lanes 32-63), i.e.
// the 16-byte pieces the streaming kernels load per lane (stream_gemm.hip).  One fixed accumulation order per
// row means a result does not depend on which kernel, tile position, shard or layer form produced it.
__device__ __forceinline__ constexpr int kcol(int s) { return (s >> 2) * 8 + (s & 3); }

// Workgroup barrier that orders LDS only.  __syncthreads() also drains vmcnt, which would stall
// every K-chunk on the prefetch loads still in flight and on the previous tile's epilogue stores
// (cdna_hip_programming.md section 5: "the ~20% stall").  The staged global loads are ordered by
// the register dependence of the ds_write that consumes them (compiler-counted vmcnt(N)).
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// tanh on the hardware transcendentals (v_exp_f32, v_rcp_f32: 1 ulp each; absolute error ~1.2e-7): libm tanhf is
// ~40 VALU instructions per element, and VALU work is not hidden under fp32 MFMAs (profiles/r02_notes.md) -- in the
// MLP-MP update GEMM (mlpmessagepassing.py:62-63: Tanh after the dense layer) it was a third of the kernel
__device__ __forceinline__ float fast_tanh(float v) {
  return 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.8853900817779268f * v)) - 1.0f;
}

template <int ACT>
__device__ __forceinline__ float act_apply(float v) {
  if constexpr (ACT == PTGNN_AMD_ACT_TANH) return fast_tanh(v);
  if constexpr (ACT == PTGNN_AMD_ACT_RELU) return v > 0.f ? v : 0.f;
  return v;
}

// Per-edge dropout of the GGNN message input (gatedmessagepassing.py:57-61: Linear(Dropout([x_src ; f_e]))).
// The keep mask is a counter-based hash of (seed, message row, input column), so forward, the input
// gradient and the weight gradient regenerate the SAME mask without an [E, H] mask tensor in HBM.
// One 32-bit hash serves a column pair: bits 0-15 decide the even column, bits 16-31 the odd one;
// an element is kept when its 16 bits >= thr = round(p * 65536), and kept values are scaled by
// 1 / (1 - p) like nn.Dropout.  tests/helpers.py restates this in numpy for the parity tests.
struct DropoutParams {
  uint64_t seed;
  uint32_t thr;        // 0 => dropout off
  float scale;
  int32_t half_width;  // (forward input width) / 2
};

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x21f0aaadu; x ^= x >> 15; x *= 0x735a2d97u; x ^= x >> 15;
  return x;
}

__device__ __forceinline__ uint32_t dropout_bits(const DropoutParams &d, int64_t row, int col_pair) {
  const uint64_t idx = (uint64_t)row * (uint64_t)d.half_width + (uint64_t)col_pair;
  uint32_t h = mix32((uint32_t)idx ^ (uint32_t)d.seed);
  return mix32(h + (uint32_t)(idx >> 32) * 0x9e3779b9u + (uint32_t)(d.seed >> 32));
}

// v = 4 consecutive columns starting at the (multiple-of-4) column `col` of message row `row`
__device__ __forceinline__ float4 dropout_apply4(const DropoutParams &d, int64_t row, int col, float4 v) {
  const uint32_t b0 = dropout_bits(d, row, col >> 1), b1 = dropout_bits(d, row, (col >> 1) + 1);
  v.x = (b0 & 0xffffu) >= d.thr ? v.x * d.scale : 0.f;
  v.y = (b0 >> 16) >= d.thr ? v.y * d.scale : 0.f;
  v.z = (b1 & 0xffffu) >= d.thr ? v.z * d.scale : 0.f;
  v.w = (b1 >> 16) >= d.thr ? v.w * d.scale : 0.f;
  return v;
}

inline DropoutParams make_dropout(float p, uint64_t seed, int width) {
  DropoutParams d;
  d.seed = seed;
  d.thr = p > 0.f ? (uint32_t)((double)p * 65536.0 + 0.5) : 0u;
  d.scale = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
  d.half_width = width / 2;
  return d;
}

// Epilogue staging: a wave parks a 32 x 64 slab of its C fragments in LDS (row stride 68 floats:
// conflict-free ds_write_b32 in the MFMA C layout, 16-B aligned rows for ds_read_b128) and streams
// it out as float4 rows -- 256 contiguous bytes per row.
constexpr int TILE_FLOATS = 128 * LDS_LD;

}  // namespace
}  // namespace ptgnn_amd
