// Streaming form of the weight-gradient GEMMs (the training step's  dW_t = d_msg_t^T . in_t  over the edges of a type, and
// dW = g^T . x over the rows of a dense Linear / GRU gate GEMM):
//
//     C[m, k] = sum_e  A[e, m] * B[e, k]        A = d_msg rows (contiguous), B = gathered / plain input rows
//
// The reduction dimension is the ROW index, the output is tiny.  The tile kernel (edge_wgrad.hip) stages both operands
// through LDS in 32-row steps behind two workgroup barriers each and sits at 60 % matrix-pipe duty with 22-24 % of its
// wave cycles waiting on memory (profiles/r03_wgrad_pmc.txt).  Here NEITHER operand touches LDS and there is no barrier
// in the loop:
//   * a wave owns a whole output tile (up to 128 x 128: 16 accumulator blocks = 256 VGPRs -- one wave per SIMD, which
//     gfx950's 512-register budget allows) and a contiguous range of rows;
//   * an MFMA step multiplies two rows (lanes 0-31: row e, lanes 32-63: row e + 1).  The row index of a C block is a
//     free permutation, so lane li loads the 16 bytes A[e][4 li .. 4 li + 3] and uses component c as its operand of
//     block c (block c covers the columns 4 li + c): ONE dwordx4 per operand per step feeds 16 MFMAs, every load
//     is a full 512-byte row per half wave;
//   * rows are prefetched 8 steps (8 x 1024 MFMA cycles) ahead in registers, node ids a block further;
//   * the four waves of a workgroup (same type and tile, consecutive row ranges) meet once, at the end, through LDS
//     (a row of C blocks at a time); one partial tile per workgroup goes to the workspace, a second small launch adds the partials of a
//     type in a fixed order (deterministic, like every reduction of this library).
// Shapes it takes: message and input widths that are multiples of 32 (tiles of 32 / 64 / 128 columns); the rest stays
// on the tile kernel.
#include <mutex>
#include <unordered_map>

#include "dense_common.h"
#include "stream_gemm.h"
#include "wgrad_stream.h"

namespace ptgnn_amd {
namespace {

#ifndef PTGNN_WS_DEPTH
#define PTGNN_WS_DEPTH 8
#endif
constexpr int kDepth = PTGNN_WS_DEPTH;   // k-steps of rows in flight per wave
constexpr int kWavesPerWg = 4;

struct WsArgs {
  WsTable tab;
  const float *x; int64_t ld_x; int64_t num_rows; int H; int use_dst;
  const float *gm; int64_t ld_gm; int64_t gm_row_base;
  int ch;                        // rows per wave
  int mtiles, ktiles;
  float *partial;                // [workgroups][WA * WB]
  float *colsum_partial;         // [groups * mtiles][WA] (dense form, k-tile 0)
  DropoutParams drop;
  const uint32_t *mask;          // DROP == 2: keep bits [d_msg rows][mask_ld] dwords (ptgnn_amd_dropout_bitmask)
  int mask_ld;
};

template <int N> struct Vec;
template <> struct Vec<4> { using T = float4; };
template <> struct Vec<2> { using T = float2; };
template <> struct Vec<1> { using T = float; };

template <int N>
__device__ __forceinline__ void load_vec(const float *p, float (&v)[N]) {
  if constexpr (N == 4) {
    const float4 t = *reinterpret_cast<const float4 *>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else if constexpr (N == 2) {
    const float2 t = *reinterpret_cast<const float2 *>(p);
    v[0] = t.x; v[1] = t.y;
  } else {
    v[0] = *p;
  }
}

template <int N>
__device__ __forceinline__ void store_vec(float *p, const float (&v)[N]) {
  if constexpr (N == 4) *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
  else if constexpr (N == 2) *reinterpret_cast<float2 *>(p) = make_float2(v[0], v[1]);
  else *p = v[0];
}

// DROP: 0 none, 1 = keep mask from the hash of dense_common.h (~55 VALU instructions per step), 2 = keep mask as bits
// (one dword load + ~13 VALU per step; fp32 MFMA shares the vector lanes with the VALU, so this is kernel time)
template <int NBA, int NBB, int DROP, bool COLSUM, bool GATHER>
__global__ __launch_bounds__(kWavesPerWg * 64, 1) void k_wgrad_stream(WsArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int WA = 32 * NBA, WB = 32 * NBB, D = kDepth;
  constexpr int kTileFloats = WA * WB;
  // The tiles of one row group read the same rows: they take consecutive positions on ONE XCD (block b runs on XCD b % 8),
  // so the second and third reader of a row hit that XCD's L2 instead of HBM.
  const int wg = (int)xcd_swizzle(blockIdx.x, p.tab.wg_off[p.tab.num_types]);
  if (wg >= p.tab.wg_off[p.tab.num_types]) return;      // ragged tail of the padded grid
  int t;
  {
    int lo = 0, hi_t = p.tab.num_types;
    while (hi_t - lo > 1) {
      const int mid = (lo + hi_t) >> 1;
      if (p.tab.wg_off[mid] <= wg) lo = mid; else hi_t = mid;
    }
    t = lo;
  }
  const int tiles = p.mtiles * p.ktiles;
  const int local = wg - p.tab.wg_off[t];
  const int g = local / tiles, tile = local - g * tiles;
  const int mt = tile / p.ktiles, kt = tile - mt * p.ktiles;
  const int64_t n_edges = p.tab.edge_off[t + 1] - p.tab.edge_off[t];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, hi = lane >> 5;
  const int64_t e0 = ((int64_t)g * kWavesPerWg + wave) * p.ch;
  const int64_t e1 = e0 + p.ch < n_edges ? e0 + p.ch : n_edges;
  const int nsteps = e1 > e0 ? (int)((e1 - e0 + 1) >> 1) : 0;
  const int64_t gm_row0 = p.gm_row_base + p.tab.edge_off[t];
  const float *const a_col = p.gm + (mt * WA + NBA * li);
  const int colb = kt * WB + NBB * li;                  // column of the input row [x_src ; x_dst]
  const bool from_dst = p.use_dst && colb >= p.H;        // per LANE: a tile may hold columns of both halves
  const float *const b_col = p.x + (from_dst ? colb - p.H : colb);
  const int64_t *__restrict__ idx = from_dst ? p.tab.dst[t] : p.tab.src[t];   // null: row e of x (dense form)

  f32x16 acc[NBA][NBB];
#pragma unroll
  for (int i = 0; i < NBA; ++i)
#pragma unroll
    for (int j = 0; j < NBB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float cs[NBA];
#pragma unroll
  for (int i = 0; i < NBA; ++i) cs[i] = 0.f;

  // edge of (step, this lane), clamped into the wave's range (steps past the end load valid memory and add zeros)
  auto edge_at = [&](int step) -> int64_t {
    const int64_t e = e0 + 2 * (int64_t)step + hi;
    return e < e1 ? e : e1 - 1;
  };
  // Node ids travel as the raw low dword of the int64 entry and are clamped where they are USED: clamping at the load
  // would put an s_waitcnt vmcnt(0) -- a drain of every row in flight -- behind each id load.  (The lists were
  // range-checked by the plan build; the clamp only keeps a corrupted entry inside the table.)
  const unsigned last_row = (unsigned)(p.num_rows - 1);
  auto raw_id = [&](int64_t e) -> int {
    if constexpr (GATHER) return reinterpret_cast<const int *>(idx)[2 * e];
    else return (int)e;
  };
  // row offsets are one 32 x 32 -> 64 bit multiply each (the host checks that rows and leading dimensions fit 32 bits)
  const unsigned ldx = (unsigned)p.ld_x, ldg = (unsigned)p.ld_gm;
  auto x_off = [&](int raw) -> uint64_t {
    const unsigned u = (unsigned)raw;
    return (uint64_t)(u < last_row ? u : last_row) * ldx;
  };
  auto gm_off = [&](int64_t e) -> uint64_t { return (uint64_t)(unsigned)(gm_row0 + e) * ldg; };
  if (nsteps > 0) {
    float a[D][NBA], b[D][NBB];
    int nid1[D], nid2[D];                                 // node ids of the next block and of the one after
    uint32_t mw[D];                                       // DROP == 2: the mask dword of each row in flight
    const int mword = colb >> 5, mshift = colb & 31;      // this lane's four columns inside the row's mask
    auto mask_at = [&](int64_t e) -> const uint32_t * {
      return p.mask + (uint64_t)(unsigned)(gm_row0 + e) * (unsigned)p.mask_ld + mword;
    };
    auto masked4 = [&](float (&bv)[NBB], uint32_t w) {
      if constexpr (NBB == 4) {
        const uint32_t m = w >> mshift;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int t = ((int)(m << (31 - i))) >> 31;
          bv[i] = __uint_as_float(__float_as_uint(bv[i] * p.drop.scale) & (uint32_t)t);
        }
      }
    };
#pragma unroll
    for (int s = 0; s < D; ++s) {
      const int64_t e = edge_at(s);
      load_vec<NBA>(a_col + gm_off(e), a[s]);
      load_vec<NBB>(b_col + x_off(raw_id(e)), b[s]);
      if constexpr (DROP == 2) mw[s] = *mask_at(e);
    }
#pragma unroll
    for (int s = 0; s < D; ++s) nid1[s] = raw_id(edge_at(D + s));
    // Everything of the prologue lands before the loop: the compiler orders the prologue's loads freely, and the wait
    // it then puts at the top of the loop has to hold for that order as well as for the steady state's (it came out as
    // a drain of all rows in flight, once per block).
    __builtin_amdgcn_s_waitcnt(0);
    const int nblocks = (nsteps + D - 1) / D;
    int j = 0;
    // Blocks whose own rows, refills (block j + 1) and id loads (block j + 2) all lie inside the wave's range take no
    // masks and no clamps of the row index, and their addresses advance by constants: four or five vector ALU
    // instructions a step instead of ~25.  That matters because fp32 MFMA shares the vector lanes with the VALU on
    // gfx950 -- address arithmetic between the MFMAs is not hidden behind them (profiles/r02_notes.md).
    const int nfull = (int)((e1 - e0) >> 1) / D;                          // blocks made of whole, valid steps
    const int nfast = nfull > 2 ? nfull - 2 : 0;
    if (nfast > 0) {
      const char *pa = reinterpret_cast<const char *>(a_col + gm_off(e0 + 2 * D + hi));       // step D = block 1, slot 0
      const uint64_t astep = 8ull * ldg;                                                       // two rows, in bytes
      const unsigned ldxb = 4u * ldx;
      const char *const bbase = reinterpret_cast<const char *>(b_col);
      const char *pb = bbase + (uint64_t)(unsigned)(e0 + 2 * D + hi) * ldxb;                   // dense form: row e of x
      const uint64_t bstep = 2ull * ldxb;
      const uint32_t *pm = nullptr;
      if constexpr (DROP == 2) pm = mask_at(e0 + 2 * D + hi);
      const unsigned mstep = 2u * (unsigned)p.mask_ld;
      const int *pid = nullptr;
      if constexpr (GATHER) pid = reinterpret_cast<const int *>(idx) + 2 * (e0 + hi) + 4 * (2 * D);   // block 2, slot 0
      for (; j < nfast; ++j) {
        if constexpr (GATHER) {
#pragma unroll
          for (int s = 0; s < D; ++s) nid2[s] = pid[4 * s];
          pid += 4 * D;
        }
#pragma unroll
        for (int s = 0; s < D; ++s) {
          float bv[NBB];
#pragma unroll
          for (int i = 0; i < NBB; ++i) bv[i] = b[s][i];
          if constexpr (DROP == 1) {
            const int64_t e = e0 + 2 * (int64_t)(j * D + s) + hi;
            const float4 m4 = dropout_apply4(p.drop, gm_row0 + e, colb, make_float4(bv[0], bv[1], bv[2], bv[3]));
            bv[0] = m4.x; bv[1] = m4.y; bv[2] = m4.z; bv[3] = m4.w;
          }
          if constexpr (DROP == 2) masked4(bv, mw[s]);
          if constexpr (COLSUM) {
#pragma unroll
            for (int i = 0; i < NBA; ++i) cs[i] += a[s][i];
          }
#ifdef PTGNN_WS_PROBE_NOMFMA
          acc[0][0][s] += a[s][0] * bv[0] + a[s][NBA - 1] * bv[NBB - 1];
#else
#pragma unroll
          for (int i = 0; i < NBA; ++i)
#pragma unroll
            for (int k = 0; k < NBB; ++k)
              acc[i][k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][i], bv[k], acc[i][k], 0, 0, 0);
#endif
          __builtin_amdgcn_sched_barrier(0);               // (see the fences of the masked loop below)
#ifndef PTGNN_WS_PROBE_NOLOAD
          load_vec<NBA>(reinterpret_cast<const float *>(pa), a[s]);
          pa += astep;
          if constexpr (GATHER) {
            const unsigned u = (unsigned)nid1[s];
            load_vec<NBB>(reinterpret_cast<const float *>(bbase + (uint64_t)(u < last_row ? u : last_row) * ldxb), b[s]);
          } else {
            load_vec<NBB>(reinterpret_cast<const float *>(pb), b[s]);
            pb += bstep;
          }
          if constexpr (DROP == 2) { mw[s] = *pm; pm += mstep; }
#endif
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (GATHER) {
#pragma unroll
          for (int s = 0; s < D; ++s) nid1[s] = nid2[s];
        }
      }
      if constexpr (!GATHER) {
#pragma unroll
        for (int s = 0; s < D; ++s) nid1[s] = raw_id(edge_at((j + 1) * D + s));
      }
    }
    // the last blocks of the range (and ranges shorter than three blocks): masked rows, clamped indices
    for (; j < nblocks; ++j) {
#pragma unroll
      for (int s = 0; s < D; ++s) nid2[s] = raw_id(edge_at((j + 2) * D + s));
#pragma unroll
      for (int s = 0; s < D; ++s) {
        const int step = j * D + s;
        const int64_t e = e0 + 2 * (int64_t)step + hi;
        const bool valid = e < e1;
        float av[NBA], bv[NBB];
#pragma unroll
        for (int i = 0; i < NBA; ++i) av[i] = valid ? a[s][i] : 0.f;
#pragma unroll
        for (int i = 0; i < NBB; ++i) bv[i] = b[s][i];
        static_assert(DROP == 0 || NBB == 4, "dropout masks are applied per float4 of the input row");
        if constexpr (DROP == 1) {
          const float4 m4 = dropout_apply4(p.drop, gm_row0 + (valid ? e : e1 - 1), colb,
                                           make_float4(bv[0], bv[1], bv[2], bv[3]));
          bv[0] = m4.x; bv[1] = m4.y; bv[2] = m4.z; bv[3] = m4.w;
        }
        if constexpr (DROP == 2) masked4(bv, mw[s]);
        if constexpr (COLSUM) {
#pragma unroll
          for (int i = 0; i < NBA; ++i) cs[i] += av[i];
        }
#pragma unroll
        for (int i = 0; i < NBA; ++i)
#pragma unroll
          for (int k = 0; k < NBB; ++k)
            acc[i][k] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[k], acc[i][k], 0, 0, 0);
        // refill the slot with the same step of the next block.  The fences keep the scheduler from gathering the
        // eight refills into one group at the top of the loop (which puts a full memory latency in front of every
        // block): the load has to leave right behind the MFMAs that read the slot.
        __builtin_amdgcn_sched_barrier(0);
        const int64_t en = edge_at((j + 1) * D + s);
        load_vec<NBA>(a_col + gm_off(en), a[s]);
        load_vec<NBB>(b_col + x_off(nid1[s]), b[s]);
        if constexpr (DROP == 2) mw[s] = *mask_at(en);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int s = 0; s < D; ++s) nid1[s] = nid2[s];
    }
  }

  // ---- the four waves' tiles meet in LDS, one row of C blocks (a quarter of a 128-wide tile) at a time.  The
  // accumulators only ever move accumulator register -> LDS: adding into them in place would need all 256 of them in
  // architectural VGPRs at once, which does not fit next to anything else (the first version of this kernel spilled
  // here).  The sums are taken by all 256 threads from LDS, ((w0 + w1) + (w2 + w3)), and go straight to the partial.
  // C block (i, k), register r of lane (li, hi) = tile row NBA * ((r & 3) + 8 (r >> 2) + 4 hi) + i, column NBB * li + k
  {
    float *const out = p.partial + (int64_t)wg * kTileFloats;
    constexpr int kPiece = NBB * 16 * 64;                 // floats of one wave's row of blocks
#pragma unroll
    for (int i = 0; i < NBA; ++i) {
      if (i > 0) __syncthreads();
#pragma unroll
      for (int k = 0; k < NBB; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) lds[wave * kPiece + (k * 16 + r) * 64 + lane] = acc[i][k][r];
      __syncthreads();
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const int pair = n * 256 + (int)threadIdx.x;      // (r, lane) of the fragment layout
        const int r = pair >> 6, ln = pair & 63;
        float v[NBB];
#pragma unroll
        for (int k = 0; k < NBB; ++k) {
          const float *const q = lds + (k * 16 + r) * 64 + ln;
          v[k] = (q[0] + q[kPiece]) + (q[2 * kPiece] + q[3 * kPiece]);
        }
        const int m = NBA * ((r & 3) + 8 * (r >> 2) + 4 * (ln >> 5)) + i;
        store_vec<NBB>(out + m * WB + NBB * (ln & 31), v);
      }
    }
  }
  if constexpr (COLSUM) {
    if (kt == 0) {   // column sums of A (= the bias gradient): 8 partial sums per column (4 waves x 2 row parities)
      __syncthreads();
#pragma unroll
      for (int i = 0; i < NBA; ++i) lds[(wave * 2 + hi) * WA + NBA * li + i] = cs[i];
      __syncthreads();
      if ((int)threadIdx.x < WA) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 2 * kWavesPerWg; ++q) s += lds[q * WA + threadIdx.x];
        p.colsum_partial[((int64_t)g * p.mtiles + mt) * WA + threadIdx.x] = s;
      }
    }
  }
}

// grad_w[t][m][k .. k+3] = sum over the type's workgroups of their partial tiles.  Eight lanes share one output float4:
// lane `sub` adds groups sub, sub + 8, ... in ascending order, then the eight sums meet in a fixed xor butterfly.
constexpr int kSplit = 8;
__global__ __launch_bounds__(256) void k_wgrad_stream_reduce(WsTable tab, const float *__restrict__ partial, int mtiles,
                                                             int ktiles, int WA, int WB, int M, int K,
                                                             float *__restrict__ grad_w, int type_base) {
  const int tiles = mtiles * ktiles;
  const int tile4 = WA * WB / 4;
  const int64_t per_type = (int64_t)tiles * tile4;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t i = gid / kSplit;
  const int sub = (int)(gid % kSplit);
  if (i >= per_type * tab.num_types) return;      // whole 8-lane groups leave together
  const int t = (int)(i / per_type);
  const int rem = (int)(i % per_type);
  const int tile = rem / tile4, f = rem % tile4;
  const int groups = (tab.wg_off[t + 1] - tab.wg_off[t]) / tiles;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int g = sub; g < groups; g += kSplit) {
    const float4 v = *reinterpret_cast<const float4 *>(
        partial + ((int64_t)tab.wg_off[t] + (int64_t)g * tiles + tile) * (WA * WB) + (int64_t)f * 4);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
#pragma unroll
  for (int d = 1; d < kSplit; d <<= 1) {
    s.x += __shfl_xor(s.x, d); s.y += __shfl_xor(s.y, d);
    s.z += __shfl_xor(s.z, d); s.w += __shfl_xor(s.w, d);
  }
  if (sub == 0) {
    const int m = (tile / ktiles) * WA + f / (WB / 4), k = (tile % ktiles) * WB + (f % (WB / 4)) * 4;
    *reinterpret_cast<float4 *>(grad_w + ((int64_t)(type_base + t) * M + m) * K + k) = s;
  }
}

__global__ __launch_bounds__(256) void k_wgrad_stream_colsum(const float *__restrict__ colsum_partial, int groups,
                                                             int mtiles, int WA, int M, float *__restrict__ grad_b) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  const int m = gid / kSplit, sub = gid % kSplit;
  if (m >= M) return;
  float s = 0.f;
  for (int g = sub; g < groups; g += kSplit) s += colsum_partial[((int64_t)g * mtiles + m / WA) * WA + m % WA];
#pragma unroll
  for (int d = 1; d < kSplit; d <<= 1) s += __shfl_xor(s, d);
  if (sub == 0) grad_b[m] = s;
}

template <typename Kern>
bool set_lds(Kern kern, size_t bytes) {
  static std::mutex mu;
  static std::unordered_map<uint64_t, size_t> done;
  int dev = 0;
  (void)hipGetDevice(&dev);
  const void *fn = reinterpret_cast<const void *>(kern);
  const uint64_t key = (uint64_t)(uintptr_t)fn * 64u + (uint64_t)dev;
  std::lock_guard<std::mutex> lock(mu);
  auto it = done.find(key);
  if (it != done.end() && it->second >= bytes) return true;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  done[key] = bytes;
  return true;
}

int blocks_for(int width) { return width % 128 == 0 ? 4 : (width % 64 == 0 ? 2 : (width % 32 == 0 ? 1 : 0)); }

// (Sizing the launch for every workgroup the small tiles could keep resident -- 3 per CU at 64 x 64, 2 at 64 x 128 --
// was measured on the hidden-64 shapes and lost: 116 k x [64 x 64] 43 -> 57 us, the README architecture's edge form
// 165 -> 174 us.  More workgroups mean more partial tiles for the reduce launch and shorter row runs per wave.)
}  // namespace

size_t stream_wgrad_workspace_floats(int64_t num_edges, int num_types, int msg_dim, int in_dim) {
  const int nba = blocks_for(msg_dim), nbb = blocks_for(in_dim);
  if (nba == 0 || nbb == 0) return 0;
  const int tiles = (msg_dim / (32 * nba)) * (in_dim / (32 * nbb));
  const int64_t wgs = (int64_t)num_compute_units() + (int64_t)num_types * tiles + tiles;
  return (size_t)wgs * (size_t)(32 * nba) * (32 * nbb) + (size_t)wgs * 128;
}

// 1 = taken, 0 = not this kernel's shape / mode (the caller runs the tile kernel), < 0 = error
int stream_wgrad(const WsTable &tab_in, const float *x, int64_t ld_x, int64_t num_rows, int state_dim, int use_dst,
                 const float *gm, int64_t ld_gm, int64_t gm_row_base, int msg_dim, float dropout_p, uint64_t dropout_seed,
                 float *grad_w, int type_base, float *grad_b, float *workspace, size_t workspace_floats,
                 hipStream_t st, const uint32_t *mask_bits) {
  static const bool off = getenv("PTGNN_AMD_WGRAD_STREAM") && getenv("PTGNN_AMD_WGRAD_STREAM")[0] == '0';
  if (off) return 0;
  const DropoutParams drop = make_dropout(dropout_p, dropout_seed, state_dim);
  const int K = state_dim * (use_dst ? 2 : 1);
  const int nba = blocks_for(msg_dim), nbb = blocks_for(K);
  if (nba == 0 || nbb == 0) return 0;
  // (a k-tile may straddle the [src ; dst] seam: which half -- table, id list -- a lane reads is decided per lane, and
  //  a lane's NBB columns never straddle it because state_dim % 4 == 0)
  if (drop.thr != 0 && nbb != 4) return 0;
  if (mask_bits && (state_dim % 32 != 0 || use_dst)) return 0;
  if (ld_x >= ((int64_t)1 << 29) || ld_gm >= ((int64_t)1 << 29) || num_rows >= ((int64_t)1 << 31)) return 0;
  const int WA = 32 * nba, WB = 32 * nbb;
  const int mtiles = msg_dim / WA, ktiles = K / WB, tiles = mtiles * ktiles;
  WsArgs p;
  p.tab = tab_in;
  int64_t E = p.tab.edge_off[p.tab.num_types];
  if (E == 0 || gm_row_base + E >= ((int64_t)1 << 32)) return 0;
  // rows per wave: one resident round of workgroups (one 4-wave workgroup per CU)
  const int64_t slots = num_compute_units();
  int64_t budget = slots / tiles - p.tab.num_types;
  if (budget < 1) budget = 1;
  int64_t ch = (E + kWavesPerWg * budget - 1) / (kWavesPerWg * budget);
  ch = (ch + 1) & ~(int64_t)1;
  if (ch < 64) ch = 64;
  int64_t total = 0;
  p.tab.wg_off[0] = 0;
  for (int t = 0; t < p.tab.num_types; ++t) {
    const int64_t n = p.tab.edge_off[t + 1] - p.tab.edge_off[t];
    total += (n + kWavesPerWg * ch - 1) / (kWavesPerWg * ch) * tiles;
    p.tab.wg_off[t + 1] = (int32_t)total;
  }
  if (total == 0) return 0;
  const size_t need = (size_t)total * WA * WB + (size_t)total * 128;
  if (need > workspace_floats) return 0;
  p.x = x; p.ld_x = ld_x; p.num_rows = num_rows; p.H = state_dim; p.use_dst = use_dst;
  p.gm = gm; p.ld_gm = ld_gm; p.gm_row_base = gm_row_base; p.ch = (int)ch; p.mtiles = mtiles; p.ktiles = ktiles;
  p.partial = workspace;
  p.colsum_partial = workspace + (size_t)total * WA * WB;
  p.drop = drop;
  p.mask = mask_bits; p.mask_ld = state_dim / 32;
  size_t lds = (size_t)kWavesPerWg * nbb * 16 * 64 * sizeof(float);   // one row of C blocks per wave
  if (lds < (size_t)2 * kWavesPerWg * WA * sizeof(float)) lds = (size_t)2 * kWavesPerWg * WA * sizeof(float);
  const bool colsum = grad_b != nullptr, dropout = drop.thr != 0;
  bool gather = false;
  for (int t = 0; t < p.tab.num_types; ++t) gather = gather || p.tab.src[t] != nullptr;
#define PTGNN_WS_LAUNCH(NBA_, NBB_, DROP_, CS_)                                      \
  do {                                                                               \
    auto kern = gather ? k_wgrad_stream<NBA_, NBB_, DROP_, false, true>              \
                       : k_wgrad_stream<NBA_, NBB_, 0, CS_, false>;                  \
    if (!set_lds(kern, lds)) return 0;                                               \
    kern<<<(unsigned)xcd_padded_blocks(total), kWavesPerWg * 64, lds, st>>>(p);                         \
  } while (0)
#define PTGNN_WS_NBB(NBA_)                                                           \
  do {                                                                               \
    if (nbb == 4) {                                                                  \
      if (dropout && mask_bits) PTGNN_WS_LAUNCH(NBA_, 4, 2, false);                  \
      else if (dropout) PTGNN_WS_LAUNCH(NBA_, 4, 1, false);                          \
      else if (colsum) PTGNN_WS_LAUNCH(NBA_, 4, 0, true);                            \
      else PTGNN_WS_LAUNCH(NBA_, 4, 0, false);                                       \
    } else if (nbb == 2) {                                                           \
      if (colsum) PTGNN_WS_LAUNCH(NBA_, 2, 0, true);                                 \
      else PTGNN_WS_LAUNCH(NBA_, 2, 0, false);                                       \
    } else {                                                                         \
      if (colsum) PTGNN_WS_LAUNCH(NBA_, 1, 0, true);                                 \
      else PTGNN_WS_LAUNCH(NBA_, 1, 0, false);                                       \
    }                                                                                \
  } while (0)
  if (nba == 4) PTGNN_WS_NBB(4); else if (nba == 2) PTGNN_WS_NBB(2); else PTGNN_WS_NBB(1);
#undef PTGNN_WS_NBB
#undef PTGNN_WS_LAUNCH
  if (hipGetLastError() != hipSuccess) return -1;
  count_launch(PTGNN_AMD_KERNEL_WGRAD_STREAM);
  const int64_t outs = (int64_t)p.tab.num_types * tiles * (WA * WB / 4) * kSplit;
  k_wgrad_stream_reduce<<<(unsigned)((outs + 255) / 256), 256, 0, st>>>(p.tab, p.partial, mtiles, ktiles, WA, WB,
                                                                         msg_dim, K, grad_w, type_base);
  if (colsum) {
    const int groups = (int)(total / tiles);
    k_wgrad_stream_colsum<<<(unsigned)((msg_dim * kSplit + 255) / 256), 256, 0, st>>>(p.colsum_partial, groups, mtiles,
                                                                                    WA, msg_dim, grad_b);
  }
  if (hipGetLastError() != hipSuccess) return -1;
  return 1;
}

}  // namespace ptgnn_amd
