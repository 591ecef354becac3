// EXPERIMENT (round 4): what does ONE wave-wide vector-memory instruction cost a wave that is otherwise issuing fp32 MFMAs?
// Loop of v_mfma_f32_32x32x2_f32 (4 accumulators, 2 waves per SIMD, 256 workgroups) with, per 16 MFMAs:
//   S  global_store_dwordx4 in the streaming kernels' epilogue shape (8 rows x 128 B per instruction), either into a small
//      cache-resident window (RESIDENT) or streaming through a 512 MB buffer,
//   G  global_load_dwordx4 in the A-operand shape (32 gathered rows x 2 x 16 B), rows pseudo-random in a 59 MB table.
// TFLOP/s against the bare loop gives cycles per memory instruction (profiles/r04_notes.md 6).
#include <hip/hip_runtime.h>
#include <stdint.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int S, int G, bool RESIDENT, int NT = 0>
__global__ __launch_bounds__(512, 2) void k_probe(const float *__restrict__ table, int64_t table_rows, float *__restrict__ out,
                                                  int64_t out_rows, int iters, float *__restrict__ sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t wid = (int64_t)blockIdx.x * 8 + wave;
  float a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = 1.0f + 0.001f * ((threadIdx.x + i) & 7); b[i] = 0.5f + 0.001f * ((threadIdx.x * 3 + i) & 7); }
  f32x16 acc[4];
  for (int n = 0; n < 4; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  float4 gacc = make_float4(0.f, 0.f, 0.f, 0.f);
  uint32_t rng = (uint32_t)(wid * 64 + lane) * 2654435761u + 12345u;
  const int srow = lane >> 3, scol = (lane & 7) * 4;
  for (int it = 0; it < iters; ++it) {
    float4 gv[G > 0 ? G : 1];
#pragma unroll
    for (int g = 0; g < G; ++g) {   // 32 rows (lane & 31) x 2 halves, 512-B rows
      rng = rng * 1664525u + 1013904223u;
      const int64_t row = (int64_t)((rng >> 8) % (uint32_t)table_rows);
      const int64_t r32 = __shfl(row, lane & 31);                       // both halves of a row read the same row
      gv[g] = *reinterpret_cast<const float4 *>(table + r32 * 128 + (lane >> 5) * 4 + g * 8);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[(i + n) & 3], acc[n], 0, 0, 0);
#pragma unroll
    for (int g = 0; g < G; ++g) { gacc.x += gv[g].x; gacc.y += gv[g].w; }
#pragma unroll
    for (int s = 0; s < S; ++s) {   // 8 rows x 128 B, like one (column block, row group) of a unit's epilogue
      // RESIDENT: 64 units (1 MB) shared by everybody; streaming: every wave cycles through 16 units of its own (2048 waves x
      // 256 KB = 512 MB in flight: nothing survives in the caches between two visits); addresses cost one shift-and-add
      const int64_t unit = RESIDENT ? (wid & 63) : (wid * 16 + (it & 15));
      float *p = out + (unit * 32 + 8 * (s & 3) + srow) * 128 + 32 * ((s >> 2) & 3) + scol;
      using f32x4 = __attribute__((ext_vector_type(4))) float;
      const f32x4 v = {acc[s & 3][0], acc[s & 3][1], gacc.x, (float)it};
      if constexpr (NT == 1) __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(p));
      else if constexpr (NT >= 100) {   // buffer store with cache-policy bits NT - 100 (1 = sc0, 2 = nt, 16 = sc1)
        using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0x7fffffff, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc, (int)((p - out) * 4), 0, NT - 100);
      } else *reinterpret_cast<f32x4 *>(p) = v;
    }
  }
  float sum = gacc.x + gacc.y;
  for (int n = 0; n < 4; ++n) for (int r = 0; r < 16; ++r) sum += acc[n][r];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

// loads in the A-operand shape (32 gathered 512-B rows x 2 halves x 16 B), rows random in the table, results consumed one
// iteration later (16 MFMAs x 2 waves of cover: the load latency is hidden, what remains is what ISSUING the load costs);
// BUF: raw_buffer_load_b128 through a descriptor of the table + a 32-bit byte offset instead of global_load_dwordx4
template <int G, bool BUF>
__global__ __launch_bounds__(512, 2) void k_probe_loads(const float *__restrict__ table, int64_t table_rows, int iters,
                                                        float *__restrict__ sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t wid = (int64_t)blockIdx.x * 8 + wave;
  float a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = 1.0f + 0.001f * ((threadIdx.x + i) & 7); b[i] = 0.5f + 0.001f * ((threadIdx.x * 3 + i) & 7); }
  f32x16 acc[4];
  for (int n = 0; n < 4; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  using f32x4 = __attribute__((ext_vector_type(4))) float;
  using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
  f32x4 cur[G > 0 ? G : 1], nxt[G > 0 ? G : 1];
  for (int g = 0; g < G; ++g) cur[g] = f32x4{0.f, 0.f, 0.f, 0.f};
  float gacc = 0.f;
  uint32_t rng = (uint32_t)(wid * 32 + (lane & 31)) * 2654435761u + 12345u;     // both halves of a row share the row
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(table), 0, 0x7fffffff, 0x00020000);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      rng = rng * 1664525u + 1013904223u;
      const uint32_t row = (rng >> 8) % (uint32_t)table_rows;
      const uint32_t off = row * 512u + (lane >> 5) * 16u + g * 32u;          // bytes
      if constexpr (BUF) nxt[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, 0));
      else nxt[g] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(table) + off);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[(i + n) & 3], acc[n], 0, 0, 0);
#pragma unroll
    for (int g = 0; g < G; ++g) { gacc += cur[g].x + cur[g].w; cur[g] = nxt[g]; }
  }
  float sum = gacc;
  for (int g = 0; g < G; ++g) sum += cur[g].y;
  for (int n = 0; n < 4; ++n) for (int r = 0; r < 16; ++r) sum += acc[n][r];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

#define CASEL(ID, G, B) if (variant == ID) k_probe_loads<G, B><<<blocks, 512, 0, st>>>(table, table_rows, iters, sink);
#define CASE(ID, S, G, R) if (variant == ID) k_probe<S, G, R><<<blocks, 512, 0, st>>>(table, table_rows, out, out_rows, iters, sink);
#define CASENT(ID, S, G, R) if (variant == ID) k_probe<S, G, R, 1><<<blocks, 512, 0, st>>>(table, table_rows, out, out_rows, iters, sink);
#define CASEB(ID, S, AUX) if (variant == ID) k_probe<S, 0, false, 100 + AUX><<<blocks, 512, 0, st>>>(table, table_rows, out, out_rows, iters, sink);
extern "C" int vmem_probe(const float *table, int64_t table_rows, float *out, int64_t out_rows, int iters, int blocks,
                          int variant, float *sink, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  CASE(0, 0, 0, true) CASE(1, 1, 0, true) CASE(2, 2, 0, true) CASE(3, 4, 0, true) CASE(4, 1, 0, false) CASE(5, 2, 0, false)
  CASE(6, 4, 0, false) CASENT(12, 1, 0, false) CASENT(13, 2, 0, false) CASENT(14, 4, 0, false) CASEB(15, 1, 0) CASEB(16, 1, 16) CASEB(17, 1, 17) CASEB(18, 1, 1) CASEB(19, 1, 3) CASEB(20, 2, 16) CASEB(21, 2, 17) CASEL(22, 1, false) CASEL(23, 1, true) CASEL(24, 2, false) CASEL(25, 2, true) CASEL(26, 4, false) CASEL(27, 4, true) CASE(7, 0, 1, true) CASE(8, 0, 2, true) CASE(9, 0, 4, true) CASE(10, 1, 1, false) CASE(11, 2, 2, false)
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
