// Weight gradient of the per-edge message Linear, every edge type in one launch:
//
//     dW_t[m, k] = sum_{e < E_t}  d_msg[off_t + e, m] * in_t[e, k],
//     in_t[e, :] = [ X[src_t[e], :] ; X[dst_t[e], :] (optional) ]   (optionally with the dropout mask
//                                                                     of the forward, regenerated)
//
// i.e. the autograd of `edge_transformation_layer(...)` in gatedmessagepassing.py:57-61 /
// mlpmessagepassing.py:96-98 for all T types, without materialising the gathered [E, K] input.
//
// It is a GEMM whose reduction dimension is the EDGE list (long) and whose output is tiny
// ([M, K] per type), so the edge list is cut into chunks: one workgroup owns (chunk, 128 x 128 output
// tile), accumulates its partial on fp32 MFMA and writes it to a workspace; k_wgrad_reduce then adds
// the partials of a type in chunk order.  Two stages instead of float atomics keeps the result
// deterministic (bit-identical run to run), like every other reduction in this library.
//
// LDS layout is reduction-major ("[edge][m]" and "[edge][k]"): both operands are read from HBM as
// contiguous rows (d_msg rows, gathered X rows), stored with ds_write_b128 as they are, and the MFMA
// operand reads walk consecutive floats of one LDS row -- no transposes anywhere.  A wave's two MFMA row blocks
// interleave (block i holds rows 2 l + i of its 64), so one ds_read_b64 feeds both blocks of an operand.
// Partial tiles are stored in FRAGMENT order (each lane's accumulator quads as contiguous float4s: 16 coalesced
// dwordx4 stores per wave instead of 64 strided dword stores); k_wgrad_reduce reads them in the same order and
// un-permutes on its (much rarer) stores.
#include "dense_common.h"
#include "stream_gemm.h"
#include "wgrad_stream.h"

namespace ptgnn_amd {
namespace {

constexpr int kMaxTypesW = 64;
constexpr int WG_LD = 132;            // floats per LDS row: 128 + 4 (rows stay 16-byte aligned)
#ifndef PTGNN_WGRAD_STEP
#define PTGNN_WGRAD_STEP 32
#endif
constexpr int STEP = PTGNN_WGRAD_STEP;   // edges per LDS stage (multiple of 8).  32 / 48 measured level, 64 loses a third (two workgroups per CU): profiles/r03_notes.md
constexpr int kTile = 128 * 128;      // floats per partial tile

struct WgradTable {
  const int64_t *src[kMaxTypesW];
  const int64_t *dst[kMaxTypesW];     // null when the input has no target-state half
  int64_t edge_off[kMaxTypesW + 1];   // prefix of edges: global d_msg row of the type's edge 0
  int32_t chunk_off[kMaxTypesW + 1];  // prefix of edge chunks
  int32_t num_types;
};

// COLSUM: additionally emit the column sums of d_msg (= the bias gradient of a dense Linear) from the
// A tiles already in LDS, for the k-tile-0 workgroups.
template <bool DROP, bool COLSUM>
__global__ __launch_bounds__(256, 3) void k_edge_wgrad(
    WgradTable tab, const float *__restrict__ x, int64_t ld_x, int64_t num_rows, int H, int use_dst,
    const float *__restrict__ gm, int64_t ld_gm, int M, int64_t gm_row_base, int chunk_edges,
    int mtiles, int ktiles, float *__restrict__ partial, int chunk_base, DropoutParams drop,
    float *__restrict__ colsum_partial) {
  __shared__ __attribute__((aligned(16))) float As[STEP * WG_LD];
  __shared__ __attribute__((aligned(16))) float Bs[STEP * WG_LD];

  const int tiles_per_chunk = mtiles * ktiles;
  const int64_t total = (int64_t)tab.chunk_off[tab.num_types] * tiles_per_chunk;
  const int64_t tile = xcd_swizzle(blockIdx.x, gridDim.x);
  if (tile >= total) return;
  const int chunk = (int)(tile / tiles_per_chunk);
  const int rem = (int)(tile % tiles_per_chunk);
  const int mt = rem / ktiles, kt = rem % ktiles;
  int lo = 0, hi_t = tab.num_types;
  while (hi_t - lo > 1) {
    const int mid = (lo + hi_t) >> 1;
    if (tab.chunk_off[mid] <= chunk) lo = mid; else hi_t = mid;
  }
  const int t = lo;
  const int64_t n_edges = tab.edge_off[t + 1] - tab.edge_off[t];
  const int64_t e_begin = (int64_t)(chunk - tab.chunk_off[t]) * chunk_edges;
  const int64_t e_end = e_begin + chunk_edges < n_edges ? e_begin + chunk_edges : n_edges;
  const int64_t gm_row0 = gm_row_base + tab.edge_off[t];
  const int K = use_dst ? 2 * H : H;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, hi = lane >> 5;

  // staging geometry: thread -> (4 edges of the stage, one float4 column of each operand)
  const int erow = threadIdx.x >> 5;               // + 8 r
  const int c4 = (threadIdx.x & 31) * 4;
  int ca = mt * 128 + c4;                          // d_msg column (clamped: extra columns are never stored)
  ca = ca <= M - 4 ? ca : M - 4;
  int kc = kt * 128 + c4;                          // input column
  kc = kc <= K - 4 ? kc : K - 4;
  const bool from_dst = kc >= H;
  const int cb = from_dst ? kc - H : kc;
  const int64_t *__restrict__ idx = from_dst ? tab.dst[t] : tab.src[t];

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nsteps = (int)((e_end - e_begin + STEP - 1) / STEP);
  // software pipeline: node ids of stage s+2 and rows of stage s+1 are in flight under the MFMAs of stage s.
  // A thread stages NR = STEP / 8 edges of each operand (one float4 column of each); fixed-trip unrolled loops
  // over plain arrays keep them in registers (arrays captured by a lambda ended up in scratch memory).
  constexpr int NR = STEP / 8;
  int64_t nid[NR];
  float4 va[NR], vb[NR];
  auto edge_of = [&](int s_, int r_) -> int64_t {
    const int64_t e_ = e_begin + (int64_t)s_ * STEP + erow + r_ * 8;
    return e_ < n_edges ? e_ : n_edges - 1;
  };
#define WG_LOAD_IDX(S)                                                                                   \
  _Pragma("unroll") for (int r_ = 0; r_ < NR; ++r_) {                                                    \
    int64_t v_ = edge_of(S, r_);                                                                         \
    if (idx) { /* uniform: a null index list means "row e of x" (dense weight gradient) */              \
      v_ = idx[v_];                                                                                      \
      v_ = v_ < 0 ? 0 : v_;                                                                              \
      v_ = v_ < num_rows ? v_ : num_rows - 1;   /* ids were range-checked by the plan build */           \
    }                                                                                                    \
    nid[r_] = v_;                                                                                        \
  }
#define WG_LOAD_ROWS(S)                                                                                  \
  _Pragma("unroll") for (int r_ = 0; r_ < NR; ++r_) {                                                    \
    va[r_] = *reinterpret_cast<const float4 *>(gm + (gm_row0 + edge_of(S, r_)) * ld_gm + ca);            \
    vb[r_] = *reinterpret_cast<const float4 *>(x + nid[r_] * ld_x + cb);                                 \
  }
  WG_LOAD_IDX(0)
  WG_LOAD_ROWS(0)
  if (nsteps > 1) { WG_LOAD_IDX(1) }

  float colsum = 0.f;
  for (int s = 0; s < nsteps; ++s) {
    __syncthreads();
    const bool tail = s + 1 == nsteps;   // rows past the chunk must add nothing; only the last stage can hold any
#pragma unroll
    for (int r_ = 0; r_ < NR; ++r_) {
      const int64_t e_ = e_begin + (int64_t)s * STEP + erow + r_ * 8;
      float4 a_ = va[r_], b_ = vb[r_];
      if constexpr (DROP) b_ = dropout_apply4(drop, gm_row0 + e_, kc, b_);
      if (tail && e_ >= e_end) {   /* componentwise: `cond ? float4 : float4` becomes a pointer select through scratch */
        a_ = make_float4(0.f, 0.f, 0.f, 0.f);
        b_ = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      *reinterpret_cast<float4 *>(As + (erow + r_ * 8) * WG_LD + c4) = a_;
      *reinterpret_cast<float4 *>(Bs + (erow + r_ * 8) * WG_LD + c4) = b_;
    }
    __syncthreads();
    if (s + 1 < nsteps) {
      WG_LOAD_ROWS(s + 1)
      if (s + 2 < nsteps) { WG_LOAD_IDX(s + 2) }
    }
    if constexpr (COLSUM) {
      if (kt == 0 && threadIdx.x < 128) {
        float cs0 = 0.f, cs1 = 0.f;
#pragma unroll
        for (int e = 0; e < STEP; e += 2) {
          cs0 += As[e * WG_LD + threadIdx.x];
          cs1 += As[(e + 1) * WG_LD + threadIdx.x];
        }
        colsum += cs0 + cs1;
      }
    }
    const float *ap = As + hi * WG_LD + wm * 64 + 2 * li;
    const float *bp = Bs + hi * WG_LD + wn * 64 + 2 * li;
#pragma unroll
    for (int ks = 0; ks < STEP / 2; ++ks) {
      const float2 a2 = *reinterpret_cast<const float2 *>(ap + ks * 2 * WG_LD);
      const float2 b2 = *reinterpret_cast<const float2 *>(bp + ks * 2 * WG_LD);
      const float a[2] = {a2.x, a2.y}, b[2] = {b2.x, b2.y};
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
#undef WG_LOAD_IDX
#undef WG_LOAD_ROWS
  if constexpr (COLSUM) {
    if (kt == 0 && threadIdx.x < 128)
      colsum_partial[((int64_t)(chunk_base + chunk) * mtiles + mt) * 128 + threadIdx.x] = colsum;
  }
  // partial tile in fragment order: float4 f = ((wave * 4 + i * 2 + j) * 4 + q) * 64 + lane holds accumulator
  // registers 4 q .. 4 q + 3 of block (i, j), i.e. tile rows wm * 64 + 2 * (8 q + 4 hi + {0, 1, 2, 3}) + i at
  // tile column wn * 64 + 2 * li + j (see frag_coords)
  float4 *const out = reinterpret_cast<float4 *>(partial + ((int64_t)(chunk_base + chunk) * tiles_per_chunk + rem) * kTile);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        out[((wave * 4 + i * 2 + j) * 4 + q) * 64 + lane] =
            make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
}

// float4 f of a fragment-ordered partial tile -> tile row of its first component (the four components are rows
// m0, m0 + 2, m0 + 4, m0 + 6) and its tile column
__device__ __forceinline__ void frag_coords(int f, int &m0, int &k) {
  const int lane = f & 63, q = (f >> 6) & 3, blk = (f >> 8) & 3, wave = f >> 10;
  const int i = blk >> 1, j = blk & 1, wm = wave >> 1, wn = wave & 1;
  m0 = wm * 64 + 2 * (8 * q + 4 * (lane >> 5)) + i;
  k = wn * 64 + 2 * (lane & 31) + j;
}

// grad_w[t][m][k .. k+3] = sum of the type's partial tiles.  Eight lanes share one output float4: lane
// `sub` adds chunks sub, sub+8, ... in ascending order, then the eight sums meet in a fixed xor
// butterfly -- a fixed summation order, so the result is bit-identical run to run.
constexpr int kSplit = 8;
__global__ __launch_bounds__(256) void k_wgrad_reduce(WgradTable tab, const float *__restrict__ partial,
                                                      int chunk_base, int mtiles, int ktiles, int M,
                                                      int K, float *__restrict__ grad_w, int type_base) {
  const int tiles_per_chunk = mtiles * ktiles;
  const int64_t per_type = (int64_t)tiles_per_chunk * (kTile / 4);     // float4s of one type's tiles
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t i = gid / kSplit;
  const int sub = (int)(gid % kSplit);
  if (i >= per_type * tab.num_types) return;      // whole 8-lane groups leave together
  const int t = (int)(i / per_type);
  const int rem = (int)(i % per_type);
  const int tile = rem / (kTile / 4), f = rem % (kTile / 4);
  int m0, k;
  frag_coords(f, m0, k);
  m0 += (tile / ktiles) * 128;
  k += (tile % ktiles) * 128;
  const int64_t off = (int64_t)tile * kTile + (int64_t)f * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int c = tab.chunk_off[t] + sub; c < tab.chunk_off[t + 1]; c += kSplit) {
    const float4 p = *reinterpret_cast<const float4 *>(
        partial + (int64_t)(chunk_base + c) * tiles_per_chunk * kTile + off);
    s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
  }
#pragma unroll
  for (int d = 1; d < kSplit; d <<= 1) {
    s.x += __shfl_xor(s.x, d); s.y += __shfl_xor(s.y, d);
    s.z += __shfl_xor(s.z, d); s.w += __shfl_xor(s.w, d);
  }
  if (sub == 0 && k < K) {
    float *g = grad_w + ((int64_t)(type_base + t) * M + m0) * K + k;
    if (m0 < M) g[0] = s.x;
    if (m0 + 2 < M) g[2 * (int64_t)K] = s.y;
    if (m0 + 4 < M) g[4 * (int64_t)K] = s.z;
    if (m0 + 6 < M) g[6 * (int64_t)K] = s.w;
  }
}

// grad_b[m] = sum over chunks of the column-sum partials (dense form: one "type")
__global__ __launch_bounds__(256) void k_colsum_reduce(const float *__restrict__ colsum_partial,
                                                       int num_chunks, int mtiles, int M,
                                                       float *__restrict__ grad_b) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  const int m = gid / kSplit, sub = gid % kSplit;
  if (m >= M) return;
  float s = 0.f;
  for (int c = sub; c < num_chunks; c += kSplit)
    s += colsum_partial[((int64_t)c * mtiles + (m >> 7)) * 128 + (m & 127)];
#pragma unroll
  for (int d = 1; d < kSplit; d <<= 1) s += __shfl_xor(s, d);
  if (sub == 0) grad_b[m] = s;
}

// Edges per chunk.  The kernel is resident three workgroups to a CU (__launch_bounds__(256, 3)), i.e. 768 at a time
// on an MI355X, and every workgroup does the same work -- so the launch should be ONE wave of workgroups: round 2
// aimed at ~1024 ("4 per CU"), which ran as 1.3 rounds, a third of the chip idle through the second (the dense GRU
// weight gradient: 987 workgroups, 171 us against a 72 us MFMA floor).  `slots` whole rounds, with room for the one
// extra chunk every edge type may add.
constexpr int kWgradPerCu = 3;
inline int chunk_edges_for(int64_t num_edges, int mtiles, int ktiles, int num_types = 1) {
  const int64_t slots = (int64_t)kWgradPerCu * num_compute_units();
  const int64_t tiles = (int64_t)mtiles * ktiles;
  int64_t rounds = 1;
  int64_t ch;
  for (;;) {   // smallest number of rounds whose chunks are not longer than the pipeline can use
    int64_t budget = slots * rounds / tiles - num_types;
    if (budget < 1) budget = 1;
    ch = (num_edges + budget - 1) / budget;
    if (ch <= 8192 || rounds >= 64) break;
    ++rounds;
  }
  ch = (ch + STEP - 1) / STEP * STEP;
  if (ch < 256) ch = 256;
  return (int)ch;
}

}  // namespace
}  // namespace ptgnn_amd

using namespace ptgnn_amd;

extern "C" size_t ptgnn_amd_edge_wgrad_workspace_bytes(int64_t num_edges, int32_t num_types,
                                                       int32_t msg_dim, int32_t in_dim) {
  if (num_edges <= 0 || num_types <= 0 || msg_dim <= 0 || in_dim <= 0) return 0;
  const int mtiles = (msg_dim + 127) / 128, ktiles = (in_dim + 127) / 128;
  const int ch = chunk_edges_for(num_edges, mtiles, ktiles, num_types);
  const int64_t chunks = num_edges / ch + num_types;   // upper bound of sum_t ceil(E_t / ch)
  const size_t tile_form = (size_t)chunks * mtiles * (ktiles * kTile + 128) * sizeof(float);   // tiles + column-sum partials
  const size_t stream_form = stream_wgrad_workspace_floats(num_edges, num_types, msg_dim, in_dim) * sizeof(float);
  return tile_form > stream_form ? tile_form : stream_form;
}

static int weight_grad_launch(const float *x, int64_t ld_x, int64_t num_rows, int32_t state_dim,
                              const int64_t *const *src_per_type, const int64_t *const *dst_per_type,
                              const int64_t *edges_per_type, const float *grad_msg, int64_t ld_grad_msg,
                              int32_t num_types, int32_t msg_dim, float dropout_p, uint64_t dropout_seed,
                              float *grad_w, void *workspace, size_t workspace_bytes, void *stream_,
                              bool identity_rows, float *grad_b, const uint32_t *mask_bits = nullptr) {
  PTGNN_REQUIRE(num_types >= 0 && state_dim > 0 && msg_dim > 0, PTGNN_AMD_EINVAL, "edge_weight_grad: bad sizes");
  PTGNN_REQUIRE(state_dim % 4 == 0 && msg_dim % 4 == 0, PTGNN_AMD_EUNSUPPORTED,
                "edge_weight_grad: needs state_dim %% 4 == 0 and msg_dim %% 4 == 0 (got %d, %d)",
                state_dim, msg_dim);
  PTGNN_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, PTGNN_AMD_EINVAL, "edge_weight_grad: bad dropout p");
  PTGNN_REQUIRE(dropout_p == 0.f || dst_per_type == nullptr, PTGNN_AMD_EUNSUPPORTED,
                "edge_weight_grad: dropout with a target-state half is not a reference configuration");
  if (num_types == 0) return PTGNN_AMD_OK;
  PTGNN_REQUIRE(x && (src_per_type || identity_rows) && edges_per_type && grad_msg && grad_w,
                PTGNN_AMD_EINVAL, "edge_weight_grad: null pointer");
  PTGNN_REQUIRE(ld_x % 4 == 0 && ld_grad_msg % 4 == 0 && ld_grad_msg >= msg_dim && aligned16(x) &&
                    aligned16(grad_msg) && aligned16(grad_w),
                PTGNN_AMD_EUNSUPPORTED, "edge_weight_grad: rows must be 16-byte aligned");
  const int use_dst = dst_per_type != nullptr;
  const int K = state_dim * (use_dst ? 2 : 1);
  const int mtiles = (msg_dim + 127) / 128, ktiles = (K + 127) / 128;
  int64_t E = 0;
  for (int t = 0; t < num_types; ++t) {
    PTGNN_REQUIRE(edges_per_type[t] >= 0, PTGNN_AMD_EINVAL, "edge_weight_grad: negative edge count");
    E += edges_per_type[t];
  }
  const int ch = chunk_edges_for(E, mtiles, ktiles, num_types);
  PTGNN_REQUIRE(workspace_bytes >= ptgnn_amd_edge_wgrad_workspace_bytes(E, num_types, msg_dim, K) &&
                    (E == 0 || workspace),
                PTGNN_AMD_EINVAL, "edge_weight_grad: workspace too small (%zu bytes)", workspace_bytes);
  PTGNN_REQUIRE(!grad_b || (identity_rows && num_types == 1 && dropout_p == 0.f), PTGNN_AMD_EINVAL,
                "edge_weight_grad: column sums are a dense-form output");
  const DropoutParams drop = make_dropout(dropout_p, dropout_seed, state_dim);
  hipStream_t st = (hipStream_t)stream_;
  int64_t row_base = 0;
  int64_t chunk_base = 0;
  const int64_t total_chunk_bound = E / ch + num_types;   // as in ptgnn_amd_edge_wgrad_workspace_bytes
  for (int t0 = 0; t0 < num_types; t0 += kMaxTypesW) {
    WgradTable tab;
    tab.num_types = (num_types - t0 < kMaxTypesW) ? (num_types - t0) : kMaxTypesW;
    tab.edge_off[0] = 0;
    tab.chunk_off[0] = 0;
    for (int t = 0; t < tab.num_types; ++t) {
      const int64_t n = edges_per_type[t0 + t];
      PTGNN_REQUIRE(n == 0 || identity_rows || (src_per_type[t0 + t] && (!use_dst || dst_per_type[t0 + t])),
                    PTGNN_AMD_EINVAL, "edge_weight_grad: null table entry for type %d", t0 + t);
      tab.src[t] = identity_rows ? nullptr : src_per_type[t0 + t];
      tab.dst[t] = use_dst ? dst_per_type[t0 + t] : nullptr;
      tab.edge_off[t + 1] = tab.edge_off[t] + n;
      const int64_t chunks = tab.chunk_off[t] + (n + ch - 1) / ch;
      PTGNN_REQUIRE(chunk_base + chunks < ((int64_t)1 << 30), PTGNN_AMD_EUNSUPPORTED,
                    "edge_weight_grad: too many chunks");
      tab.chunk_off[t + 1] = (int32_t)chunks;
    }
    {   // the streaming form (wgrad_stream.hip) takes widths that are multiples of 32
      WsTable ws;
      ws.num_types = tab.num_types;
      for (int t = 0; t < tab.num_types; ++t) {
        ws.src[t] = tab.src[t];
        ws.dst[t] = tab.dst[t];
        ws.edge_off[t] = tab.edge_off[t];
      }
      ws.edge_off[tab.num_types] = tab.edge_off[tab.num_types];
      const int taken = stream_wgrad(ws, x, ld_x, num_rows, state_dim, use_dst, grad_msg, ld_grad_msg, row_base, msg_dim,
                                     dropout_p, dropout_seed, grad_w, t0, grad_b, (float *)workspace, workspace_bytes / sizeof(float), st,
                                     mask_bits);
      PTGNN_REQUIRE(taken >= 0, PTGNN_AMD_EHIP, "edge_weight_grad: streaming launch failed");
      if (taken == 1) {
        row_base += tab.edge_off[tab.num_types];
        chunk_base += tab.chunk_off[tab.num_types];
        continue;
      }
      PTGNN_REQUIRE(mask_bits == nullptr, PTGNN_AMD_EUNSUPPORTED,
                    "edge_weight_grad_masked: state_dim=%d msg_dim=%d is not a shape of the streaming weight-gradient "
                    "kernel (use ptgnn_amd_edge_weight_grad_f32)", state_dim, msg_dim);
    }
    const int64_t total = (int64_t)tab.chunk_off[tab.num_types] * mtiles * ktiles;
    if (total > 0) {
      const unsigned grid = (unsigned)xcd_padded_blocks(total);
      float *const colsum_ws = (float *)workspace + total_chunk_bound * mtiles * ktiles * kTile;
#define PTGNN_WGRAD_LAUNCH(DROP, COLSUM)                                                              \
  k_edge_wgrad<DROP, COLSUM><<<grid, 256, 0, st>>>(tab, x, ld_x, num_rows, state_dim, use_dst, grad_msg, \
                                                   ld_grad_msg, msg_dim, row_base, ch, mtiles, ktiles, \
                                                   (float *)workspace, (int)chunk_base, drop, colsum_ws)
      if (grad_b) PTGNN_WGRAD_LAUNCH(false, true);
      else if (drop.thr != 0) PTGNN_WGRAD_LAUNCH(true, false);
      else PTGNN_WGRAD_LAUNCH(false, false);
#undef PTGNN_WGRAD_LAUNCH
      PTGNN_LAUNCH_CHECK();
      count_launch(PTGNN_AMD_KERNEL_TILE_WGRAD);
      if (grad_b) {
        k_colsum_reduce<<<(unsigned)((msg_dim * kSplit + 255) / 256), 256, 0, st>>>(
            colsum_ws, tab.chunk_off[tab.num_types], mtiles, msg_dim, grad_b);
        PTGNN_LAUNCH_CHECK();
      }
    } else if (grad_b) {
      PTGNN_HIP(hipMemsetAsync(grad_b, 0, sizeof(float) * msg_dim, st));
    }
    const int64_t outs = (int64_t)tab.num_types * mtiles * ktiles * (kTile / 4) * kSplit;
    k_wgrad_reduce<<<(unsigned)((outs + 255) / 256), 256, 0, st>>>(
        tab, (const float *)workspace, (int)chunk_base, mtiles, ktiles, msg_dim, K, grad_w, t0);
    PTGNN_LAUNCH_CHECK();
    row_base += tab.edge_off[tab.num_types];
    chunk_base += tab.chunk_off[tab.num_types];
  }
  return PTGNN_AMD_OK;
}

extern "C" int ptgnn_amd_edge_weight_grad_f32(const float *x, int64_t ld_x, int64_t num_rows, int32_t state_dim,
                                              const int64_t *const *src_per_type,
                                              const int64_t *const *dst_per_type,
                                              const int64_t *edges_per_type, const float *grad_msg,
                                              int64_t ld_grad_msg, int32_t num_types, int32_t msg_dim,
                                              float dropout_p, uint64_t dropout_seed, float *grad_w,
                                              void *workspace, size_t workspace_bytes, void *stream_) {
  PTGNN_REQUIRE(num_rows > 0 || num_types == 0, PTGNN_AMD_EINVAL, "edge_weight_grad: num_rows must be positive");
  return weight_grad_launch(x, ld_x, num_rows, state_dim, src_per_type, dst_per_type, edges_per_type, grad_msg,
                            ld_grad_msg, num_types, msg_dim, dropout_p, dropout_seed, grad_w, workspace,
                            workspace_bytes, stream_, false, nullptr);
}

extern "C" int ptgnn_amd_edge_weight_grad_masked_supported(int32_t state_dim, int32_t msg_dim) {
  return state_dim % 128 == 0 && msg_dim % 32 == 0 ? 1 : 0;   // the streaming kernel's dropout form: 128-wide k tiles
}

extern "C" int ptgnn_amd_edge_weight_grad_masked_f32(const float *x, int64_t ld_x, int64_t num_rows, int32_t state_dim,
                                                     const int64_t *const *src_per_type, const int64_t *edges_per_type,
                                                     const float *grad_msg, int64_t ld_grad_msg, int32_t num_types,
                                                     int32_t msg_dim, float dropout_p, const uint32_t *mask_bits,
                                                     float *grad_w, void *workspace, size_t workspace_bytes,
                                                     void *stream_) {
  PTGNN_REQUIRE(num_rows > 0 || num_types == 0, PTGNN_AMD_EINVAL, "edge_weight_grad_masked: num_rows must be positive");
  PTGNN_REQUIRE(dropout_p > 0.f && dropout_p < 1.f && mask_bits != nullptr, PTGNN_AMD_EINVAL,
                "edge_weight_grad_masked: needs 0 < p < 1 and a mask");
  PTGNN_REQUIRE(ptgnn_amd_edge_weight_grad_masked_supported(state_dim, msg_dim), PTGNN_AMD_EUNSUPPORTED,
                "edge_weight_grad_masked: state_dim=%d msg_dim=%d is not a shape of the streaming weight-gradient kernel",
                state_dim, msg_dim);
  return weight_grad_launch(x, ld_x, num_rows, state_dim, src_per_type, nullptr, edges_per_type, grad_msg, ld_grad_msg,
                            num_types, msg_dim, dropout_p, 0, grad_w, workspace, workspace_bytes, stream_, false, nullptr,
                            mask_bits);
}

// grad_w [n_out, k] = grad_y^T [n_out, rows] . x [rows, k]: the same split-row GEMM with the identity
// row map (nn.Linear / nn.GRUCell weight gradients; rocBLAS runs this long-reduction TN shape at
// ~23 TFLOP/s on MI355X, profiles/r01_notes.md).
extern "C" int ptgnn_amd_linear_weight_grad_f32(const float *x, int64_t ld_x, int32_t k,
                                                const float *grad_y, int64_t ld_grad_y, int64_t rows,
                                                int32_t n_out, float *grad_w, float *grad_b,
                                                void *workspace, size_t workspace_bytes, void *stream_) {
  PTGNN_REQUIRE(rows >= 0, PTGNN_AMD_EINVAL, "linear_weight_grad: negative row count");
  const int64_t counts[1] = {rows};
  return weight_grad_launch(x, ld_x, rows, k, nullptr, nullptr, counts, grad_y, ld_grad_y, 1, n_out, 0.f, 0,
                            grad_w, workspace, workspace_bytes, stream_, true, grad_b);
}
