// EXPERIMENT (not part of libptgnn_amd): what does the exact-fp32 MFMA sustain on this chip with NOTHING else
// in the loop?  Register-only loops of v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32, operands random or zero,
// 1 or 2 waves per SIMD.  Answers whether the ~62 % matrix-pipe duty seen in every fp32 GEMM structure
// (profiles/r02_notes.md) is a property of the kernels or of the pipe under load.
#include <hip/hip_runtime.h>
#include <stdint.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int NACC>
__global__ __launch_bounds__(512, 2) void k_peak32(const float *__restrict__ src, float *__restrict__ out, int iters) {
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = src[(threadIdx.x + i * 512) & 4095]; b[i] = src[(threadIdx.x * 7 + i * 131) & 4095]; }
  f32x16 acc[NACC];
  for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[(i + n) & 7], acc[n], 0, 0, 0);
  }
  float s = 0.f;
  for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(512, 2) void k_peak16(const float *__restrict__ src, float *__restrict__ out, int iters) {
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = src[(threadIdx.x + i * 512) & 4095]; b[i] = src[(threadIdx.x * 7 + i * 131) & 4095]; }
  f32x4 acc[NACC];
  for (int n = 0; n < NACC; ++n) for (int r = 0; r < 4; ++r) acc[n][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[(i + n) & 7], acc[n], 0, 0, 0);
  }
  float s = 0.f;
  for (int n = 0; n < NACC; ++n) for (int r = 0; r < 4; ++r) s += acc[n][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// variant: grid blocks of `threads` (256 = 1 wave per SIMD, 512 = 2 waves per SIMD); kind 0 = 32x32x2, 1 = 16x16x4
extern "C" int mfma_peak(const float *src, float *out, int iters, int threads, int blocks, int kind, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  if (kind == 0) k_peak32<4><<<blocks, threads, 0, st>>>(src, out, iters);
  else k_peak16<8><<<blocks, threads, 0, st>>>(src, out, iters);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
