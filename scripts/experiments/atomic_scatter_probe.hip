// EXPERIMENT (round 2): what does a scatter-max by device-scope atomics cost on MI355X, against the plain
// row stores the grouped edge GEMM does today?  cfg3 shape: E = 625 130 message rows x M = 128 into N = 115 772
// destination rows (random destinations), the value pattern of a GEMM epilogue (a wave writes 8 rows x 32 columns
// per instruction as dwordx4 -- here: each lane owns 4 consecutive columns of one row).
//   hipcc -O3 --offload-arch=gfx950 atomic_scatter_probe.hip -o /tmp/atomic_probe && /tmp/atomic_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t key_of(float v) {
  const uint32_t b = __float_as_uint(v);
  return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);   // monotone float -> uint
}

// MODE 0: plain dwordx4 stores to message row e (what the GEMM epilogue does today: sequential rows)
// MODE 1: plain dwordx4 stores to row dst[e] (random rows, races ignored: bandwidth probe only)
// MODE 2: 4 x atomicMax(uint key) per lane into row dst[e]   (agent scope, relaxed)
// MODE 3: as 2, but only lanes whose value beats a plain (stale-tolerant) read issue the atomic
template <int MODE>
__global__ __launch_bounds__(256) void k_scatter(const float *__restrict__ msg, const int *__restrict__ dst, int64_t E,
                                                 int M, uint32_t *__restrict__ out, float *__restrict__ out_f) {
  const int lpr = M / 4;                       // lanes per row
  const int64_t rows_per_blk = 256 / lpr;
  for (int64_t e0 = (int64_t)blockIdx.x * rows_per_blk; e0 < E; e0 += (int64_t)gridDim.x * rows_per_blk) {
    const int64_t e = e0 + threadIdx.x / lpr;
    if (e >= E) continue;
    const int c = (threadIdx.x % lpr) * 4;
    const float4 v = *reinterpret_cast<const float4 *>(msg + e * M + c);
    if (MODE == 0) {
      *reinterpret_cast<float4 *>(out_f + e * M + c) = v;
    } else if (MODE == 1) {
      *reinterpret_cast<float4 *>(out_f + (int64_t)dst[e] * M + c) = v;
    } else {
      uint32_t *p = out + (int64_t)dst[e] * M + c;
      const uint32_t k0 = key_of(v.x), k1 = key_of(v.y), k2 = key_of(v.z), k3 = key_of(v.w);
      if (MODE == 3) {
        const uint4 cur = *reinterpret_cast<const uint4 *>(p);
        if (k0 > cur.x) __hip_atomic_fetch_max(p + 0, k0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (k1 > cur.y) __hip_atomic_fetch_max(p + 1, k1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (k2 > cur.z) __hip_atomic_fetch_max(p + 2, k2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (k3 > cur.w) __hip_atomic_fetch_max(p + 3, k3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        __hip_atomic_fetch_max(p + 0, k0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_max(p + 1, k1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_max(p + 2, k2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_max(p + 3, k3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

template <int MODE>
float run(const float *msg, const int *dst, int64_t E, int M, uint32_t *out, float *out_f, int64_t N, int reps) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  std::vector<float> ts;
  for (int r = 0; r < reps + 2; ++r) {
    if (MODE >= 2) CK(hipMemsetAsync(out, 0, (size_t)N * M * 4, 0));
    CK(hipEventRecord(a, 0));
    k_scatter<MODE><<<4096, 256, 0, 0>>>(msg, dst, E, M, out, out_f);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (r >= 2) ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  return ts[ts.size() / 2];
}

int main() {
  const int64_t N = 115772, E = 625130;
  for (int M : {128, 64}) {
    std::vector<float> h((size_t)E * M);
    std::vector<int> hd(E);
    srand(1);
    for (auto &x : h) x = (float)rand() / 2147483648.0f - 0.5f;
    for (auto &d : hd) d = (int)(((int64_t)rand() * 32768 + rand()) % N);
    float *msg, *out_f; int *dst; uint32_t *out;
    CK(hipMalloc(&msg, (size_t)E * M * 4)); CK(hipMalloc(&out_f, (size_t)E * M * 4));
    CK(hipMalloc(&dst, E * 4)); CK(hipMalloc(&out, (size_t)N * M * 4));
    CK(hipMemcpy(msg, h.data(), (size_t)E * M * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dst, hd.data(), E * 4, hipMemcpyHostToDevice));
    const float t0 = run<0>(msg, dst, E, M, out, out_f, N, 9);
    const float t1 = run<1>(msg, dst, E, M, out, out_f, N, 9);
    const float t2 = run<2>(msg, dst, E, M, out, out_f, N, 9);
    const float t3 = run<3>(msg, dst, E, M, out, out_f, N, 9);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a, 0)); CK(hipMemsetAsync(out, 0, (size_t)N * M * 4, 0)); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float tm; CK(hipEventElapsedTime(&tm, a, b));
    printf("M=%d E=%lld N=%lld: copy rows %.1f us | random-row stores %.1f us | atomic max (4/lane) %.1f us | read-then-atomic %.1f us | memset out %.1f us\n",
           M, (long long)E, (long long)N, t0 * 1e3, t1 * 1e3, t2 * 1e3, t3 * 1e3, tm * 1e3);
    CK(hipFree(msg)); CK(hipFree(out_f)); CK(hipFree(dst)); CK(hipFree(out));
  }
  return 0;
}
