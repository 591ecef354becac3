// Per-edge message GEMM for ALL edge types in one launch (grouped GEMM with gathered A rows):
//
//     msg[off_t + e, :] = [ X[src_t[e], :] ; X[dst_t[e], :] (optional) ] . W_t^T          e < E_t
//
// written in the reference's message order (type-major, then edge order), i.e. exactly the matrix
// `torch.cat(all_messages)` of gatedmessagepassing.py:50-64 / mlpmessagepassing.py:81-108 -- minus the
// F.embedding outputs, the torch.cat copies and the T small launches.  The segment reduce then reads
// these rows through the plan's `perm`.
//
// Why it exists next to the per-node pre-transform (ptgnn_amd_linear_f32 on [W_0; ...; W_{T-1}]):
// the pre-transform costs 2 N T H M FLOP and writes an [N, T M] table; with many sparse edge types
// (program graphs: T ~ 17-23, E/N ~ 5.4) this kernel needs 2 E H M FLOP (3x less) and writes [E, M].
// The host picks per minibatch (ptgnn_amd/layers.py).
//
// Two kernels behind the entry points: the streaming weight-stationary kernel (stream_gemm.hip: `k_stream_edge`,
// GEMM modes 1 / 2, every shape whose [M, K] weight slab fits LDS) and, here, the 128-row tile kernel for GEMM
// mode 0, the remaining shapes and the two dropout forms (training).  Both accumulate K in the same order.
//
// Structure of the tile kernel: the one-tile-per-workgroup fp32-MFMA kernel of dense_f32.hip with
//   * a tile -> (edge type, first edge) decode through a small table in the kernel arguments,
//   * the A-operand row map replaced by the int64 source (and destination) indices of the tile's
//     128 edges, read straight from ptgnn's adjacency tensors once per tile,
//   * the B operand = that type's own nn.Linear weight (no stacked copy needed).
// Bound: MFMA fp32 for the math; the gather reads E*H*4 bytes of L2/MALL-resident node states.
#include <stdlib.h>

#include <type_traits>

#include "dense_common.h"
#include "stream_gemm.h"

namespace ptgnn_amd {
namespace {

constexpr int kMaxTypes = 64;

struct EdgeTypeTable {
  const int64_t *src[kMaxTypes];
  const int64_t *dst[kMaxTypes];     // null entries when the message has no target-state half
  const float *w[kMaxTypes];
  int64_t edge_off[kMaxTypes + 1];   // prefix of edges (global message row of the type's edge 0)
  int32_t tile_off[kMaxTypes + 1];   // prefix of 128-edge tiles
  int32_t num_types;
};

// FEAT kernels: per-edge feature rows that extend the A operand behind the gathered state halves
// (gatedmessagepassing.py:57-61 `cat([edge_source_states, features], -1)`, mlpmessagepassing.py:96-98): row e of
// feat[t] belongs to edge e of type t, F columns (F % 4 == 0; the chunk tail is zeroed on the way into LDS)
struct EdgeFeat {
  const float *feat[kMaxTypes];
  int64_t ld;
  int F;
};
struct NoFeat {};

struct GatherRows {  // tile row r -> node id of edge (e0 + r); rows past the type's end repeat its last edge
  int64_t idx[4];
  // indexed by the staging part (a compile-time constant after unrolling): `idx[row >> 5]` is a
  // dynamic index to the compiler and put the array in scratch memory
  __device__ __forceinline__ int64_t operator()(int /*row*/, int part) const { return idx[part]; }
};

// DROP: 0 none; 1 = dropout on the gathered input rows (training forward: W_t . Dropout(x_src));
// 2 = dropout on the OUTPUT rows (training backward: d x_gathered = (d msg . W_t) * mask, with x = d msg
// read through an identity index and w = W_t^T).  Both index the mask by (global message row, column
// of the forward input), see DropoutParams.
template <int ACT, int NJ, int DROP, bool FEAT = false>
__global__ __launch_bounds__(256, (NJ == 1 ? 4 : 3)) void k_edge_linear(
    EdgeTypeTable tab, const float *__restrict__ x, int64_t ld_x, int64_t num_rows, int H, int use_dst, int n_out,
    float *__restrict__ msg, int64_t ld_msg, int64_t msg_row_base, int col_tiles, DropoutParams drop,
    std::conditional_t<FEAT, EdgeFeat, NoFeat> ef) {
  constexpr int BN = 64 * NJ;
  constexpr int B_FLOATS = BN * LDS_LD;
  constexpr int SLAB_LD = 32 * NJ + 4;
  constexpr int SLAB_FLOATS = 32 * SLAB_LD;
  constexpr int OPER = TILE_FLOATS + B_FLOATS;
  constexpr int kLds = OPER > 4 * SLAB_FLOATS ? OPER : 4 * SLAB_FLOATS;
  __shared__ __attribute__((aligned(16))) float smem[kLds];
  float *const As = smem, *const Bs = smem + TILE_FLOATS;

  const int64_t total_tiles = (int64_t)tab.tile_off[tab.num_types] * col_tiles;
  const int64_t tile = xcd_swizzle(blockIdx.x, gridDim.x);
  if (tile >= total_tiles) return;
  const int etile = (int)(tile / col_tiles);
  const int col0 = (int)(tile % col_tiles) * BN;
  // which edge type owns this tile (<= 6 steps over a table that lives in SGPRs)
  int lo = 0, hi_t = tab.num_types;
  while (hi_t - lo > 1) {
    const int mid = (lo + hi_t) >> 1;
    if (tab.tile_off[mid] <= etile) lo = mid; else hi_t = mid;
  }
  const int t = lo;
  const int64_t e0 = (int64_t)(etile - tab.tile_off[t]) * 128;
  const int64_t n_edges = tab.edge_off[t + 1] - tab.edge_off[t];   // > 0: empty types own no tiles
  const int64_t out_row0 = msg_row_base + tab.edge_off[t] + e0;
  const int Hs = use_dst ? 2 * H : H;
  int K = Hs;                                   // row stride of w = the whole message input
  if constexpr (FEAT) K += ef.F;
  const float *__restrict__ w = tab.w[t];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, hi = lane >> 5;

  // the node ids of this thread's four staging rows, read once per tile
  GatherRows gs, gd;
  [[maybe_unused]] GatherRows ge;               // FEAT: the edge's own row of the type's feature matrix
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int64_t e = e0 + (threadIdx.x >> 3) + r * 32;
    e = e < n_edges ? e : n_edges - 1;
    if constexpr (FEAT) ge.idx[r] = e;
    // ids were range-checked by the plan build (ptgnn_amd_csr_build); clamp anyway so a bad id can never fault
    int64_t si = tab.src[t][e], di = use_dst ? tab.dst[t][e] : 0;
    si = si < 0 ? 0 : (si < num_rows ? si : num_rows - 1);
    di = di < 0 ? 0 : (di < num_rows ? di : num_rows - 1);
    gs.idx[r] = si;
    gd.idx[r] = di;
  }

  f32x16 acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  Stager<128, true, GatherRows> sa;
  Stager<BN, true, RowClamp> sb;
  const RowClamp rb{col0, n_out};
  const int hchunks = H / BK;               // host guarantees H % 32 == 0
  const int schunks = Hs / BK;
  const int nchunks = (K + BK - 1) / BK;    // (K % 32 != 0 only with features: the feature chunks carry the tail)
  auto issue = [&](int c) {
    if (c < hchunks) sa.load(x, ld_x, c * BK, H, gs);
    else if (!FEAT || c < schunks) sa.load(x, ld_x, (c - hchunks) * BK, H, gd);
    if constexpr (FEAT) {
      if (c >= schunks) sa.load(ef.feat[t], ef.ld, (c - schunks) * BK, ef.F, ge);
    }
    sb.load(w, K, c * BK, K, rb);
  };
  issue(0);
  for (int c = 0; c < nchunks; ++c) {
    __syncthreads();
    if constexpr (DROP == 1) {   // mask this chunk's gathered rows in registers on their way into LDS
      const int dcol = c * BK + (threadIdx.x & 7) * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        sa.v[r] = dropout_apply4(drop, out_row0 + (threadIdx.x >> 3) + r * 32, dcol, sa.v[r]);
    }
    sa.store(As);
    sb.store(Bs);
    __syncthreads();
    if (c + 1 < nchunks) issue(c + 1);
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
  constexpr int LPRW = 8 * NJ, RPI = 64 / LPRW;
  const int c4 = (lane % LPRW) * 4, rsub = lane / LPRW;
  const int gcol = col0 + wn * 32 * NJ + c4;
  const int64_t rows_left = n_edges - e0;   // valid rows of this tile
  constexpr int NIT = 32 / RPI;
  if (rows_left >= 128 && col0 + BN <= n_out) {
    // interior tile: straight-line slab round trip + float4 stores (a guarded epilogue compiles to a
    // read -> wait -> branches -> store chain per float4; profiles/r01_notes.md)
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
      float *dst = msg + (out_row0 + wm * 64 + i * 32 + rsub) * ld_msg + gcol;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        float4 o;
        o.x = act_apply<ACT>(v[it].x); o.y = act_apply<ACT>(v[it].y);
        o.z = act_apply<ACT>(v[it].z); o.w = act_apply<ACT>(v[it].w);
        if constexpr (DROP == 2)
          o = dropout_apply4(drop, out_row0 + wm * 64 + i * 32 + rsub + it * RPI, gcol, o);
        *reinterpret_cast<float4 *>(dst + (int64_t)it * RPI * ld_msg) = o;
      }
    }
    return;
  }
#pragma unroll 1
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        slab[((r & 3) + 8 * (r >> 2) + 4 * hi) * SLAB_LD + j * 32 + li] = i == 0 ? acc[0][j][r] : acc[1][j][r];
    __builtin_amdgcn_wave_barrier();
#pragma unroll 1
    for (int it = 0; it < NIT; ++it) {
      const int rl = it * RPI + rsub;
      const int trow = wm * 64 + i * 32 + rl;
      float4 v = *reinterpret_cast<const float4 *>(slab + rl * SLAB_LD + c4);
      v.x = act_apply<ACT>(v.x); v.y = act_apply<ACT>(v.y);
      v.z = act_apply<ACT>(v.z); v.w = act_apply<ACT>(v.w);
      if constexpr (DROP == 2) v = dropout_apply4(drop, out_row0 + trow, gcol < n_out ? gcol : 0, v);
      if (trow < rows_left && gcol < n_out)   // host guarantees n_out % 4 == 0 and 16-B aligned rows
        *reinterpret_cast<float4 *>(msg + (out_row0 + trow) * ld_msg + gcol) = v;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// keep bits of the per-edge dropout (ptgnn_amd_dropout_bitmask): one thread per dword = 32 columns of one message row,
// the hash of dense_common.h evaluated once per layer call instead of inside three GEMMs
__global__ __launch_bounds__(256) void k_dropout_bitmask(DropoutParams d, int64_t rows, int words_per_row,
                                                         uint32_t *__restrict__ bits) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * words_per_row) return;
  const int64_t row = i / words_per_row;
  const int c = (int)(i - row * words_per_row);
  uint32_t w = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const uint32_t h = dropout_bits(d, row, c * 16 + j);
    w |= ((h & 0xffffu) >= d.thr ? 1u : 0u) << (2 * j);
    w |= ((h >> 16) >= d.thr ? 1u : 0u) << (2 * j + 1);
  }
  bits[i] = w;
}

}  // namespace
}  // namespace ptgnn_amd

using namespace ptgnn_amd;

// `mask_bits` (nullable): the keep mask as bits; with it the dropout forms run on the streaming kernel (which has no
// hash form) or fail with EUNSUPPORTED -- callers ask ptgnn_amd_edge_linear_masked_supported first
static int edge_linear_launch(const float *x, int64_t ld_x, int64_t num_rows, int32_t state_dim,
                              const int64_t *const *src_per_type, const int64_t *const *dst_per_type,
                              const int64_t *edges_per_type, const float *const *w_per_type,
                              int32_t num_types, int32_t msg_dim, int act, float *msg, int64_t ld_msg,
                              int dropout_mode, float dropout_p, uint64_t dropout_seed, void *stream_,
                              const uint32_t *mask_bits = nullptr, const float *const *feat_per_type = nullptr,
                              int64_t ld_feat = 0, int32_t feat_dim = 0) {
  PTGNN_REQUIRE(num_types >= 0 && state_dim > 0 && msg_dim > 0, PTGNN_AMD_EINVAL, "edge_linear: bad sizes");
  PTGNN_REQUIRE(act >= 0 && act <= PTGNN_AMD_ACT_RELU, PTGNN_AMD_EINVAL, "edge_linear: bad act");
  PTGNN_REQUIRE(state_dim % 32 == 0 && msg_dim % 4 == 0, PTGNN_AMD_EUNSUPPORTED,
                "edge_linear: needs state_dim %% 32 == 0 and msg_dim %% 4 == 0 (got %d, %d)", state_dim,
                msg_dim);
  PTGNN_REQUIRE(dropout_mode >= 0 && dropout_mode <= 2 && dropout_p >= 0.f && dropout_p < 1.f,
                PTGNN_AMD_EINVAL, "edge_linear: bad dropout mode / probability");
  if (dropout_p == 0.f) dropout_mode = 0;
  PTGNN_REQUIRE(dropout_mode == 0 || (act == PTGNN_AMD_ACT_NONE && dst_per_type == nullptr),
                PTGNN_AMD_EUNSUPPORTED, "edge_linear: dropout needs act none and no target-state half");
  PTGNN_REQUIRE(feat_dim >= 0 && (feat_dim == 0 || feat_per_type), PTGNN_AMD_EINVAL, "edge_linear: bad feature arguments");
  PTGNN_REQUIRE(feat_dim == 0 || (feat_dim % 4 == 0 && ld_feat % 4 == 0 && ld_feat >= feat_dim && dropout_mode == 0),
                PTGNN_AMD_EUNSUPPORTED,
                "edge_linear: edge features need feat_dim %% 4 == 0, 16-byte aligned rows and no dropout (got %d)", feat_dim);
  if (num_types == 0) return PTGNN_AMD_OK;
  PTGNN_REQUIRE(num_rows > 0, PTGNN_AMD_EINVAL, "edge_linear: num_rows must be positive");
  PTGNN_REQUIRE(x && src_per_type && edges_per_type && w_per_type && msg, PTGNN_AMD_EINVAL,
                "edge_linear: null pointer");
  PTGNN_REQUIRE(ld_x % 4 == 0 && ld_msg % 4 == 0 && ld_msg >= msg_dim && aligned16(x) && aligned16(msg),
                PTGNN_AMD_EUNSUPPORTED, "edge_linear: x/msg rows must be 16-byte aligned");
  const int use_dst = dst_per_type != nullptr;
  int nj = msg_dim <= 64 ? 1 : 2;
  if (const char *e = getenv("PTGNN_AMD_EDGE_NJ")) {   // developer A/B knob
    const int v = atoi(e);
    if (v == 1 || v == 2) nj = v;
  }
  const int col_tiles = (msg_dim + 64 * nj - 1) / (64 * nj);
  // the mask is indexed by the column of the FORWARD input: the A operand in mode 1, the output in mode 2
  const DropoutParams drop = make_dropout(dropout_p, dropout_seed, dropout_mode == 2 ? msg_dim : state_dim);
  hipStream_t st = (hipStream_t)stream_;
  int64_t row_base = 0;
  // streaming core (stream_gemm.hip): the shapes it tiles; the dropout forms only with the mask as bits.  A 256-wide
  // output (the input gradient of the last Typilus layer, whose input is the 256-wide concat residual) is two column
  // slabs of 128, each a launch of its own.
  const int col_slabs = (msg_dim == 256 && stream_edge_supported(state_dim, 128, use_dst)) ? 2 : 1;
  const int slab_dim = msg_dim / col_slabs;
  bool streaming = (dropout_mode == 0 || mask_bits != nullptr) && stream_edge_supported(state_dim, slab_dim, use_dst) &&
                   ld_x % 4 == 0 && aligned16(x) && feat_dim == 0;   // (feature rows: the tile kernel's third K phase)
  const int in_dim = use_dst ? 2 * state_dim : state_dim;
  const int mask_words = (dropout_mode == 2 ? msg_dim : state_dim) / 32;   // the mask covers the FORWARD input row
  for (int cs = 0; streaming && cs < col_slabs; ++cs) {
  row_base = 0;
  for (int t0 = 0; streaming && t0 < num_types; t0 += kStreamMaxTypes) {
    StreamEdgeTable tab;
    tab.num_types = (num_types - t0 < kStreamMaxTypes) ? (num_types - t0) : kStreamMaxTypes;
    tab.edge_off[0] = 0;
    tab.unit_off[0] = 0;
    for (int t = 0; t < tab.num_types; ++t) {
      const int64_t n = edges_per_type[t0 + t];
      PTGNN_REQUIRE(n >= 0, PTGNN_AMD_EINVAL, "edge_linear: negative edge count");
      PTGNN_REQUIRE(n == 0 || (src_per_type[t0 + t] && w_per_type[t0 + t] &&
                               (!use_dst || dst_per_type[t0 + t])),
                    PTGNN_AMD_EINVAL, "edge_linear: null table entry for type %d", t0 + t);
      PTGNN_REQUIRE(n == 0 || aligned16(w_per_type[t0 + t]), PTGNN_AMD_EUNSUPPORTED,
                    "edge_linear: weight of type %d is not 16-byte aligned", t0 + t);
      tab.src[t] = src_per_type[t0 + t];
      tab.dst[t] = use_dst ? dst_per_type[t0 + t] : src_per_type[t0 + t];
      tab.w[t] = w_per_type[t0 + t] ? w_per_type[t0 + t] + (size_t)cs * slab_dim * in_dim : nullptr;
      tab.edge_off[t + 1] = tab.edge_off[t] + n;
      const int64_t units = tab.unit_off[t] + (n + 31) / 32;
      PTGNN_REQUIRE(units < ((int64_t)1 << 30), PTGNN_AMD_EUNSUPPORTED, "edge_linear: too many units");
      tab.unit_off[t + 1] = (int32_t)units;
    }
    StreamEdgeMask mk;
    mk.mode = mask_bits ? dropout_mode : 0; mk.bits = mask_bits; mk.ld = mask_words;
    mk.col0 = dropout_mode == 2 ? cs * (slab_dim / 32) : 0;
    mk.scale = drop.scale;
    if (stream_edge(tab, x, ld_x, num_rows, state_dim, use_dst, slab_dim, act, msg + cs * slab_dim, ld_msg, row_base, st,
                    mk.mode ? &mk : nullptr) != 1) {
      // "not taken" (the dynamic-LDS attribute was refused, e.g. a first use inside a graph capture): like
      // stream_linear / stream_gru, fall through to the tile kernel, which recomputes every type chunk
      streaming = false;
      row_base = 0;
      break;
    }
    PTGNN_LAUNCH_CHECK();
    row_base += tab.edge_off[tab.num_types];
  }
  }
  if (streaming) return PTGNN_AMD_OK;
  PTGNN_REQUIRE(mask_bits == nullptr, PTGNN_AMD_EUNSUPPORTED,
                "edge_linear_masked: state_dim=%d msg_dim=%d mode=%d is not a shape of the streaming edge GEMM "
                "(use ptgnn_amd_edge_linear_dropout_f32)", state_dim, msg_dim, dropout_mode);
  row_base = 0;
  for (int t0 = 0; t0 < num_types; t0 += kMaxTypes) {
    EdgeTypeTable tab;
    EdgeFeat ef;
    ef.ld = ld_feat; ef.F = feat_dim;
    tab.num_types = (num_types - t0 < kMaxTypes) ? (num_types - t0) : kMaxTypes;
    tab.edge_off[0] = 0;
    tab.tile_off[0] = 0;
    for (int t = 0; t < tab.num_types; ++t) {
      const int64_t n = edges_per_type[t0 + t];
      PTGNN_REQUIRE(n >= 0, PTGNN_AMD_EINVAL, "edge_linear: negative edge count");
      PTGNN_REQUIRE(n == 0 || (src_per_type[t0 + t] && w_per_type[t0 + t] &&
                               (!use_dst || dst_per_type[t0 + t])),
                    PTGNN_AMD_EINVAL, "edge_linear: null table entry for type %d", t0 + t);
      PTGNN_REQUIRE(n == 0 || aligned16(w_per_type[t0 + t]), PTGNN_AMD_EUNSUPPORTED,
                    "edge_linear: weight of type %d is not 16-byte aligned", t0 + t);
      if (feat_dim) {
        PTGNN_REQUIRE(n == 0 || (feat_per_type[t0 + t] && aligned16(feat_per_type[t0 + t])), PTGNN_AMD_EINVAL,
                      "edge_linear: feature rows of type %d are null or not 16-byte aligned", t0 + t);
        ef.feat[t] = feat_per_type[t0 + t];
      }
      tab.src[t] = src_per_type[t0 + t];
      tab.dst[t] = use_dst ? dst_per_type[t0 + t] : nullptr;
      tab.w[t] = w_per_type[t0 + t];
      tab.edge_off[t + 1] = tab.edge_off[t] + n;
      const int64_t tiles = tab.tile_off[t] + (n + 127) / 128;
      PTGNN_REQUIRE(tiles < ((int64_t)1 << 30), PTGNN_AMD_EUNSUPPORTED, "edge_linear: too many tiles");
      tab.tile_off[t + 1] = (int32_t)tiles;
    }
    const int64_t total_tiles = (int64_t)tab.tile_off[tab.num_types] * col_tiles;
    if (total_tiles > 0) {
      const unsigned grid = (unsigned)xcd_padded_blocks(total_tiles);
#define PTGNN_EDGE_LAUNCH(ACT, NJ, DROP)                                                            \
  k_edge_linear<ACT, NJ, DROP, false><<<grid, 256, 0, st>>>(tab, x, ld_x, num_rows, state_dim, use_dst, msg_dim, msg, \
                                                            ld_msg, row_base, col_tiles, drop, NoFeat{})
#define PTGNN_EDGE_LAUNCH_FEAT(ACT, NJ)                                                             \
  k_edge_linear<ACT, NJ, 0, true><<<grid, 256, 0, st>>>(tab, x, ld_x, num_rows, state_dim, use_dst, msg_dim, msg, \
                                                        ld_msg, row_base, col_tiles, drop, ef)
      if (feat_dim) {
        if (nj == 1) {
          if (act == PTGNN_AMD_ACT_TANH) PTGNN_EDGE_LAUNCH_FEAT(PTGNN_AMD_ACT_TANH, 1);
          else if (act == PTGNN_AMD_ACT_RELU) PTGNN_EDGE_LAUNCH_FEAT(PTGNN_AMD_ACT_RELU, 1);
          else PTGNN_EDGE_LAUNCH_FEAT(PTGNN_AMD_ACT_NONE, 1);
        } else {
          if (act == PTGNN_AMD_ACT_TANH) PTGNN_EDGE_LAUNCH_FEAT(PTGNN_AMD_ACT_TANH, 2);
          else if (act == PTGNN_AMD_ACT_RELU) PTGNN_EDGE_LAUNCH_FEAT(PTGNN_AMD_ACT_RELU, 2);
          else PTGNN_EDGE_LAUNCH_FEAT(PTGNN_AMD_ACT_NONE, 2);
        }
      } else if (dropout_mode == 1) {
        if (nj == 1) PTGNN_EDGE_LAUNCH(PTGNN_AMD_ACT_NONE, 1, 1); else PTGNN_EDGE_LAUNCH(PTGNN_AMD_ACT_NONE, 2, 1);
      } else if (dropout_mode == 2) {
        if (nj == 1) PTGNN_EDGE_LAUNCH(PTGNN_AMD_ACT_NONE, 1, 2); else PTGNN_EDGE_LAUNCH(PTGNN_AMD_ACT_NONE, 2, 2);
      } else if (nj == 1) {
        if (act == PTGNN_AMD_ACT_TANH) PTGNN_EDGE_LAUNCH(PTGNN_AMD_ACT_TANH, 1, 0);
        else if (act == PTGNN_AMD_ACT_RELU) PTGNN_EDGE_LAUNCH(PTGNN_AMD_ACT_RELU, 1, 0);
        else PTGNN_EDGE_LAUNCH(PTGNN_AMD_ACT_NONE, 1, 0);
      } else {
        if (act == PTGNN_AMD_ACT_TANH) PTGNN_EDGE_LAUNCH(PTGNN_AMD_ACT_TANH, 2, 0);
        else if (act == PTGNN_AMD_ACT_RELU) PTGNN_EDGE_LAUNCH(PTGNN_AMD_ACT_RELU, 2, 0);
        else PTGNN_EDGE_LAUNCH(PTGNN_AMD_ACT_NONE, 2, 0);
      }
#undef PTGNN_EDGE_LAUNCH
#undef PTGNN_EDGE_LAUNCH_FEAT
      PTGNN_LAUNCH_CHECK();
      count_launch(PTGNN_AMD_KERNEL_TILE_EDGE);
    }
    row_base += tab.edge_off[tab.num_types];
  }
  return PTGNN_AMD_OK;
}

extern "C" int ptgnn_amd_edge_linear_f32(const float *x, int64_t ld_x, int64_t num_rows, int32_t state_dim,
                                         const int64_t *const *src_per_type,
                                         const int64_t *const *dst_per_type,
                                         const int64_t *edges_per_type,
                                         const float *const *w_per_type, int32_t num_types,
                                         int32_t msg_dim, int act, float *msg, int64_t ld_msg,
                                         void *stream_) {
  return edge_linear_launch(x, ld_x, num_rows, state_dim, src_per_type, dst_per_type, edges_per_type, w_per_type,
                            num_types, msg_dim, act, msg, ld_msg, 0, 0.f, 0, stream_);
}

extern "C" int ptgnn_amd_edge_linear_feat_f32(const float *x, int64_t ld_x, int64_t num_rows, int32_t state_dim,
                                              const int64_t *const *src_per_type, const int64_t *const *dst_per_type,
                                              const float *const *feat_per_type, int64_t ld_feat, int32_t feat_dim,
                                              const int64_t *edges_per_type, const float *const *w_per_type,
                                              int32_t num_types, int32_t msg_dim, int act, float *msg, int64_t ld_msg,
                                              void *stream) {
  PTGNN_REQUIRE(feat_dim > 0, PTGNN_AMD_EINVAL, "edge_linear_feat: feat_dim must be positive (use ptgnn_amd_edge_linear_f32)");
  return edge_linear_launch(x, ld_x, num_rows, state_dim, src_per_type, dst_per_type, edges_per_type, w_per_type,
                            num_types, msg_dim, act, msg, ld_msg, 0, 0.f, 0, stream, nullptr, feat_per_type, ld_feat,
                            feat_dim);
}

extern "C" int ptgnn_amd_edge_linear_shared_f32(const float *x, int64_t ld_x, int64_t num_rows, int32_t state_dim,
                                                const void *edge_table, const float *const *w_per_type,
                                                int32_t num_types, int32_t msg_dim, int act, float *msg,
                                                int64_t ld_msg, void *stream_) {
  PTGNN_REQUIRE(num_types > 0 && state_dim > 0 && msg_dim > 0 && num_rows > 0, PTGNN_AMD_EINVAL,
                "edge_linear_shared: bad sizes");
  PTGNN_REQUIRE(act == PTGNN_AMD_ACT_NONE || act == PTGNN_AMD_ACT_TANH || act == PTGNN_AMD_ACT_RELU, PTGNN_AMD_EINVAL,
                "edge_linear_shared: bad activation");
  PTGNN_REQUIRE(x && edge_table && w_per_type && msg, PTGNN_AMD_EINVAL, "edge_linear_shared: null pointer");
  PTGNN_REQUIRE(ld_x % 4 == 0 && ld_msg % 4 == 0 && ld_msg >= msg_dim && aligned16(x) && aligned16(msg),
                PTGNN_AMD_EUNSUPPORTED, "edge_linear_shared: x/msg rows must be 16-byte aligned");
  for (int t = 0; t < num_types; ++t)
    PTGNN_REQUIRE(w_per_type[t] && aligned16(w_per_type[t]), PTGNN_AMD_EINVAL,
                  "edge_linear_shared: weight of type %d is null or not 16-byte aligned", t);
  const int taken = stream_edge_indirect((const StreamEdgeTable *)edge_table, w_per_type, num_types, x, ld_x, num_rows,
                                         state_dim, msg_dim, act, msg, ld_msg, (hipStream_t)stream_);
  PTGNN_REQUIRE(taken == 1, PTGNN_AMD_EUNSUPPORTED,
                "edge_linear_shared: state_dim=%d msg_dim=%d num_types=%d is not a shape of the streaming edge GEMM",
                state_dim, msg_dim, num_types);
  PTGNN_LAUNCH_CHECK();
  return PTGNN_AMD_OK;
}

extern "C" int ptgnn_amd_edge_linear_shared_supported(int32_t state_dim, int32_t msg_dim, int32_t num_types) {
  return num_types <= kStreamMaxTypes && stream_edge_supported(state_dim, msg_dim, 0) ? 1 : 0;
}

extern "C" int ptgnn_amd_edge_linear_dropout_f32(const float *x, int64_t ld_x, int64_t num_rows,
                                                 int32_t state_dim,
                                                 const int64_t *const *src_per_type,
                                                 const int64_t *edges_per_type,
                                                 const float *const *w_per_type, int32_t num_types,
                                                 int32_t msg_dim, float *msg, int64_t ld_msg,
                                                 int dropout_mode, float dropout_p,
                                                 uint64_t dropout_seed, void *stream_) {
  return edge_linear_launch(x, ld_x, num_rows, state_dim, src_per_type, nullptr, edges_per_type, w_per_type,
                            num_types, msg_dim, PTGNN_AMD_ACT_NONE, msg, ld_msg, dropout_mode, dropout_p,
                            dropout_seed, stream_);
}

extern "C" size_t ptgnn_amd_dropout_bitmask_bytes(int64_t rows, int32_t width) {
  if (rows < 0 || width <= 0 || width % 32 != 0) return 0;
  return (size_t)rows * (size_t)(width / 32) * sizeof(uint32_t);
}

extern "C" int ptgnn_amd_dropout_bitmask(int64_t rows, int32_t width, float dropout_p, uint64_t dropout_seed,
                                         uint32_t *bits, void *stream_) {
  PTGNN_REQUIRE(rows >= 0 && width > 0 && width % 32 == 0, PTGNN_AMD_EINVAL,
                "dropout_bitmask: width must be a positive multiple of 32 (got %d)", width);
  PTGNN_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, PTGNN_AMD_EINVAL, "dropout_bitmask: bad probability");
  if (rows == 0) return PTGNN_AMD_OK;
  PTGNN_REQUIRE(bits != nullptr, PTGNN_AMD_EINVAL, "dropout_bitmask: null pointer");
  const DropoutParams d = make_dropout(dropout_p, dropout_seed, width);
  const int64_t words = rows * (width / 32);
  PTGNN_REQUIRE((words + 255) / 256 < ((int64_t)1 << 31), PTGNN_AMD_EUNSUPPORTED, "dropout_bitmask: too many rows");
  k_dropout_bitmask<<<(unsigned)((words + 255) / 256), 256, 0, (hipStream_t)stream_>>>(d, rows, width / 32, bits);
  PTGNN_LAUNCH_CHECK();
  return PTGNN_AMD_OK;
}

extern "C" int ptgnn_amd_edge_linear_masked_supported(int32_t state_dim, int32_t msg_dim, int dropout_mode) {
  if (dropout_mode != 1 && dropout_mode != 2) return 0;
  const int slab_dim = msg_dim == 256 ? 128 : msg_dim;
  return stream_edge_masked_supported(state_dim, slab_dim) ? 1 : 0;
}

extern "C" int ptgnn_amd_edge_linear_masked_f32(const float *x, int64_t ld_x, int64_t num_rows, int32_t state_dim,
                                                const int64_t *const *src_per_type, const int64_t *edges_per_type,
                                                const float *const *w_per_type, int32_t num_types, int32_t msg_dim,
                                                float *msg, int64_t ld_msg, int dropout_mode, float dropout_p,
                                                const uint32_t *mask_bits, void *stream_) {
  PTGNN_REQUIRE(dropout_mode == 1 || dropout_mode == 2, PTGNN_AMD_EINVAL, "edge_linear_masked: dropout_mode must be 1 or 2");
  PTGNN_REQUIRE(dropout_p > 0.f && dropout_p < 1.f && mask_bits != nullptr, PTGNN_AMD_EINVAL,
                "edge_linear_masked: needs 0 < p < 1 and a mask");
  PTGNN_REQUIRE(ptgnn_amd_edge_linear_masked_supported(state_dim, msg_dim, dropout_mode), PTGNN_AMD_EUNSUPPORTED,
                "edge_linear_masked: state_dim=%d msg_dim=%d is not a shape of the streaming edge GEMM", state_dim, msg_dim);
  return edge_linear_launch(x, ld_x, num_rows, state_dim, src_per_type, nullptr, edges_per_type, w_per_type, num_types,
                            msg_dim, PTGNN_AMD_ACT_NONE, msg, ld_msg, dropout_mode, dropout_p, 0, stream_, mask_bits);
}
