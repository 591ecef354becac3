// EXPERIMENT: which companion instructions cost fp32-MFMA issue slots?  Loop of v_mfma_f32_32x32x2_f32 (4
// accumulators, 2 waves per SIMD) with, per 16 MFMAs: V independent v_fma_f32, D ds_read_b128, G
// global_load_dwordx4 (L2-resident).  Compare TFLOP/s against the bare loop.
#include <hip/hip_runtime.h>
#include <stdint.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int V, int D, int G, bool BF16>
__global__ __launch_bounds__(512, 2) void k_mix(const float *__restrict__ src, float *__restrict__ out, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = src[i & 4095];
  __syncthreads();
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = src[(threadIdx.x + i * 512) & 4095]; b[i] = src[(threadIdx.x * 7 + i * 131) & 4095]; }
  f32x16 acc[4];
  for (int n = 0; n < 4; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = a[i];
  float4 dacc = make_float4(0.f, 0.f, 0.f, 0.f), gacc = dacc;
  const float4 *lp = reinterpret_cast<const float4 *>(lds) + (threadIdx.x & 63) * 9;
  const float4 *gp = reinterpret_cast<const float4 *>(src) + (threadIdx.x & 255);
  typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
  bf16x8 ab, bb;
  for (int i = 0; i < 8; ++i) { ab[i] = (__bf16)a[i]; bb[i] = (__bf16)b[i]; }
  for (int it = 0; it < iters; ++it) {
    float4 dv[D > 0 ? D : 1], gv[G > 0 ? G : 1];
#pragma unroll
    for (int d = 0; d < D; ++d) dv[d] = lp[(it + d * 64) & 1023];
#pragma unroll
    for (int g = 0; g < G; ++g) gv[g] = gp[((it + g * 8) & 3) * 256];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        if constexpr (BF16) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[n], 0, 0, 0);
        else acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[(i + n) & 7], acc[n], 0, 0, 0);
      }
#pragma unroll
    for (int k = 0; k < V; ++k) v[k & 7] = __builtin_fmaf(v[k & 7], 1.0001f, 0.5f);
#pragma unroll
    for (int d = 0; d < D; ++d) { dacc.x += dv[d].x; dacc.y += dv[d].w; }
#pragma unroll
    for (int g = 0; g < G; ++g) { gacc.x += gv[g].x; gacc.y += gv[g].w; }
  }
  float s = dacc.x + dacc.y + gacc.x + gacc.y;
  for (int i = 0; i < 8; ++i) s += v[i];
  for (int n = 0; n < 4; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

#define CASE(ID, V, D, G, B) if (variant == ID) k_mix<V, D, G, B><<<blocks, 512, 0, st>>>(src, out, iters);
extern "C" int mfma_mix(const float *src, float *out, int iters, int blocks, int variant, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  CASE(0, 0, 0, 0, false) CASE(1, 16, 0, 0, false) CASE(2, 64, 0, 0, false) CASE(3, 0, 4, 0, false)
  CASE(4, 0, 16, 0, false) CASE(5, 0, 0, 2, false) CASE(6, 0, 0, 8, false) CASE(7, 16, 4, 2, false)
  CASE(8, 0, 0, 0, true) CASE(9, 64, 0, 0, true) CASE(10, 256, 0, 0, false)
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
