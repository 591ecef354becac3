// EXPERIMENT (round 2, for the "finer units" item of DESIGN section 9): in what order does v_mfma_f32_16x16x4_f32
// accumulate its four k products, and is a chain of them bit-identical to the fmaf chain v_mfma_f32_32x32x2_f32
// realises (k ascending, fused multiply-add)?  If yes, a 16-row unit variant of the streaming kernels can keep the
// "one K accumulation order for every exact-fp32 kernel" rule.
//   hipcc -O3 --offload-arch=gfx950 mfma_16x16x4_order_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// one wave: C[16][16] = A[16][K] . B[K][16], k in steps of 4
__global__ void k16(const float *A, const float *B, float *C, int K) {
  const int l = threadIdx.x, i = l & 15, q = l >> 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < K; k0 += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i * K + k0 + q], B[(k0 + q) * 16 + i], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) C[(4 * q + r) * 16 + i] = acc[r];
}

// one wave: C[32][32] = A[32][K] . B[K][32], k in steps of 2 (the instruction the product kernels use)
__global__ void k32(const float *A, const float *B, float *C, int K) {
  const int l = threadIdx.x, i = l & 31, h = l >> 5;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int k0 = 0; k0 < K; k0 += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * K + k0 + h], B[(k0 + h) * 32 + i], acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = acc[r];
}

int main() {
  const int K = 64;
  std::vector<float> A(32 * K), B(K * 32), C16(256), C32(1024);
  srand(7);
  for (auto &x : A) x = (float)rand() / 2147483648.0f * 2.f - 1.f;
  for (auto &x : B) x = (float)rand() / 2147483648.0f * 2.f - 1.f;
  // the 16x16 problem uses rows 0..15 of A and columns 0..15 of B (B16[k][j] = B[k][j])
  std::vector<float> B16(K * 16);
  for (int k = 0; k < K; ++k) for (int j = 0; j < 16; ++j) B16[k * 16 + j] = B[k * 32 + j];
  float *dA, *dB, *dB16, *dC16, *dC32;
  CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dB16, B16.size() * 4));
  CK(hipMalloc(&dC16, 1024)); CK(hipMalloc(&dC32, 4096));
  CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB16, B16.data(), B16.size() * 4, hipMemcpyHostToDevice));
  k16<<<1, 64>>>(dA, dB16, dC16, K);
  k32<<<1, 64>>>(dA, dB, dC32, K);
  CK(hipMemcpy(C16.data(), dC16, 1024, hipMemcpyDeviceToHost));
  CK(hipMemcpy(C32.data(), dC32, 4096, hipMemcpyDeviceToHost));
  int bad_fma = 0, bad_fma32 = 0, bad_mul_add = 0, bad_pair = 0, bad_vs32 = 0, bad_rev = 0;
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      float f = 0.f, ma = 0.f, pr = 0.f, rv = 0.f;
      for (int k = 0; k < K; ++k) f = fmaf(A[i * K + k], B[k * 32 + j], f);
      for (int k = 0; k < K; ++k) { volatile float p = A[i * K + k] * B[k * 32 + j]; ma = ma + p; }
      for (int k0 = 0; k0 < K; k0 += 4) {
        const float p0 = A[i * K + k0] * B[k0 * 32 + j], p1 = A[i * K + k0 + 1] * B[(k0 + 1) * 32 + j];
        const float p2 = A[i * K + k0 + 2] * B[(k0 + 2) * 32 + j], p3 = A[i * K + k0 + 3] * B[(k0 + 3) * 32 + j];
        volatile float s01 = p0 + p1, s23 = p2 + p3;
        volatile float s = s01 + s23;
        pr = pr + s;
        float t = rv;
        for (int d = 3; d >= 0; --d) t = fmaf(A[i * K + k0 + d], B[(k0 + d) * 32 + j], t);
        rv = t;
      }
      const float got = C16[i * 16 + j];
      bad_fma += memcmp(&got, &f, 4) != 0;
      bad_mul_add += memcmp(&got, &ma, 4) != 0;
      bad_pair += memcmp(&got, &pr, 4) != 0;
      bad_rev += memcmp(&got, &rv, 4) != 0;
      bad_vs32 += memcmp(&got, &C32[i * 32 + j], 4) != 0;
    }
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      float f = 0.f;
      for (int k = 0; k < K; ++k) f = fmaf(A[i * K + k], B[k * 32 + j], f);
      bad_fma32 += memcmp(&C32[i * 32 + j], &f, 4) != 0;
    }
  printf("K=%d  32x32x2 vs host fmaf chain (k ascending): %d / 1024 differ\n", K, bad_fma32);
  printf("16x16x4 vs host fmaf chain (k ascending): %d / 256 differ | vs mul-then-add chain: %d | vs pairwise-in-4: %d | "
         "vs fmaf chain with k descending inside each 4: %d | vs the 32x32x2 result of the same rows/cols: %d\n",
         bad_fma, bad_mul_add, bad_pair, bad_rev, bad_vs32);
  return 0;
}
