// Graph plan construction: per-edge-type int64 adjacency lists -> one destination-sorted CSR.
// See include/ptgnn_amd.h (ptgnn_amd_csr_build) for the contract and the reference lines replaced.
//
// The plan is a STABLE sort of the edges by plan row (the order of a numpy stable argsort: tests compare bit
// for bit), built by hand-written kernels only -- no vendor sort, no vendor scan:
//
//   k_split_count    per sub-tile of the edge list: LDS histogram of one digit of the key (reads only the key column
//                    of the int64 lists), stored as the tile's row of the aggregate table
//   k_tile_scan      aggregate table -> exclusive prefix over the tiles, in place, + the digit totals
//   k_split_scatter  one pass over the lists: stable scatter by that digit.  Ranks: a wave owns a CONTIGUOUS run of
//                    records and walks it in rounds of 64; the rank of a record among the equal digits of its round is
//                    a ballot match, the count of its wave's earlier rounds sits in a WAVE-PRIVATE LDS counter row (no
//                    workgroup barrier per round).  The records then leave through LDS in sorted order, so that the
//                    global stores of a digit are contiguous runs: written straight from the ranking lanes every record
//                    is its own partial-line write request, and the request rate of the L2s (~50 per clock chip-wide),
//                    not bytes, bounds these kernels (profiles/r03_notes.md)
//   k_plan_buckets   MSD form only: one workgroup per bucket of consecutive rows: stable counting sort by the low row
//                    bits -> rowptr, col, perm and the hub list; the bucket's window is staged in LDS
//
// Forms (choose_path):
//   * MSD, row ids <= 18 bits and <= 4 M edges (every minibatch): count / scan / scatter on the high bits with 8-byte
//     records (low row bits | position, payload), then k_plan_buckets over <= 512-row buckets;
//   * MSD, <= 21 bits and <= 4 M edges (backward plans over rows = src * T + type): 12-byte records, buckets of up to
//     4096 rows whose wave-private counters are packed 16-bit pairs (144 KB of LDS per workgroup);
//   * LSD, anything larger (BASELINE config 5: 1.25 M rows / 12.5 M edges per GPU): ceil(bits / 9) passes of count /
//     scan / scatter from the low digit up, the last one writing col / perm / sorted keys, rowptr from the keys.  Every
//     pass is tiled over the input order, so a power-law hub costs nothing extra (the workgroup that owns a hub's
//     bucket in the MSD form walked 200 k records alone: 0.66 ms).
// HBM-bound integer work in principle; at minibatch size a chain of dependent launches.
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <unordered_map>

#include "common.h"

namespace ptgnn_amd {
namespace {

constexpr int kMaxTypes = 64;

struct TypeTable {
  const int64_t *src[kMaxTypes];
  const int64_t *dst[kMaxTypes];
  int64_t offset[kMaxTypes + 1];  // exclusive prefix of edges_per_type
  int32_t num_types;
  int32_t type_base;  // global type id of entry 0 (for > kMaxTypes chunking)
};

// mode 0: rows = dst,            payload = (src << type_bits) | type   (forward plan)
// mode 1: rows = src,            payload = (dst << type_bits) | type   (transposed plan)
// mode 2: rows = src * T + type, payload = dst                         (backward of the message table:
//         row r of the [N*T, M] gradient view sums the output gradients of its out-edges)
struct EdgeRec {
  uint32_t key;
  int32_t packed;
};

// Range guard (the reference device-asserts in F.embedding on a bad node id; here a bad id must never
// become an out-of-bounds rowptr write or gather): ids outside [0, limit) are clamped to 0 and counted in
// `bad` (nullable, accumulated -- the host reads it back asynchronously and raises).
struct RangeGuard {
  int64_t num_rows;   // plan rows (key domain)
  int64_t src_rows;   // rows of the table the payload indexes (mode 0 only; 0 = unchecked)
  int32_t *bad;
};

__device__ __forceinline__ void guard_record(const RangeGuard &g, int64_t &key, int64_t &payload_id, bool count) {
  const bool kbad = key < 0 || key >= g.num_rows;
  const bool pbad = g.src_rows > 0 && (payload_id < 0 || payload_id >= g.src_rows);
  if (kbad) key = 0;
  if (pbad) payload_id = 0;
  if ((kbad || pbad) && count && g.bad) atomicAdd(g.bad, 1);
}

// type of edge e of the type-major concatenation, known to lie in [lo, hi): binary search over the offsets
// (<= 6 steps; the table is a kernel argument).  Among equal offsets (empty types) the last one wins, which
// is the non-empty one.
__device__ __forceinline__ int type_of(const TypeTable &tab, int64_t e, int lo, int hi) {
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (tab.offset[mid] <= e) lo = mid; else hi = mid;
  }
  return lo;
}

// Types the edges [e_first, e_last] of one workgroup's tile can have, [lo, hi).  Both ends are workgroup-uniform,
// so the two searches run on the scalar unit; a tile inside ONE type (the rule: 4096 consecutive edges) then needs
// no per-lane search at all and indexes the pointer table with a uniform value (scalar loads instead of a
// per-lane walk over the argument table).
struct TypeSpan { int lo, hi; };
__device__ __forceinline__ TypeSpan tile_types(const TypeTable &tab, int64_t e_first, int64_t e_last) {
  TypeSpan sp;
  sp.lo = __builtin_amdgcn_readfirstlane(type_of(tab, e_first, 0, tab.num_types));
  sp.hi = __builtin_amdgcn_readfirstlane(type_of(tab, e_last, sp.lo, tab.num_types)) + 1;
  return sp;
}

// edge e of the type-major concatenation -> (plan row, col payload); `span`: see tile_types
__device__ __forceinline__ EdgeRec edge_record(const TypeTable &tab, int64_t e, int32_t type_bits, int mode,
                                               int total_types, const RangeGuard &guard, bool count,
                                               TypeSpan span) {
  const int lo = span.hi - span.lo == 1 ? span.lo : type_of(tab, e, span.lo, span.hi);
  const int64_t i = e - tab.offset[lo];
  int64_t s = tab.src[lo][i], d = tab.dst[lo][i];
  const int64_t ty = tab.type_base + lo;
  EdgeRec r;
  if (mode == 2) {
    int64_t key = s * total_types + ty;
    if (s < 0) key = -1;
    int64_t none = 0;
    guard_record(guard, key, none, count);
    r.key = (uint32_t)key;
    r.packed = (int32_t)d;
  } else {
    if (mode == 1) { const int64_t t = s; s = d; d = t; }
    guard_record(guard, d, s, count);
    r.key = (uint32_t)d;
    r.packed = (int32_t)((s << type_bits) | ty);
  }
  return r;
}

// plan row of edge e (the sort key), reading only the column that holds it; same clamping as edge_record
__device__ __forceinline__ uint32_t edge_key(const TypeTable &tab, int64_t e, int mode, int total_types,
                                             const RangeGuard &guard, TypeSpan span) {
  const int lo = span.hi - span.lo == 1 ? span.lo : type_of(tab, e, span.lo, span.hi);
  const int64_t i = e - tab.offset[lo];
  int64_t key;
  if (mode == 2) {
    const int64_t s_ = tab.src[lo][i];
    key = s_ < 0 ? -1 : s_ * total_types + (tab.type_base + lo);
  } else {
    key = mode == 1 ? tab.src[lo][i] : tab.dst[lo][i];
  }
  return (key < 0 || key >= guard.num_rows) ? 0u : (uint32_t)key;
}

// (key, position, payload) of every edge as three arrays: the output of an LSD pre-pass, and of k_pack for
// batches with more than kMaxTypes edge types
struct Soa {
  uint32_t *key;
  int32_t *pos;
  int32_t *packed;
};

__global__ __launch_bounds__(256) void k_pack(TypeTable tab, int32_t type_bits, int mode, int total_types,
                                              Soa out, int64_t pos_base, RangeGuard guard) {
  const int64_t total = tab.offset[tab.num_types];
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const EdgeRec r = edge_record(tab, e, type_bits, mode, total_types, guard, true, TypeSpan{0, tab.num_types});
    const int64_t g = pos_base + e;
    out.pos[g] = (int32_t)g;
    out.key[g] = r.key;
    out.packed[g] = r.packed;
  }
}

// ---- the split kernels (one stable partition pass over the edge list by one digit of the key) --------------
constexpr int kMaxBins = 512;
#ifndef PTGNN_SPLIT_WAVES
#define PTGNN_SPLIT_WAVES 8
#endif
#ifndef PTGNN_SPLIT_ROUNDS
#define PTGNN_SPLIT_ROUNDS 8
#endif
#ifndef PTGNN_BUCKET_WAVES
#define PTGNN_BUCKET_WAVES 16
#endif
constexpr int kSplitWaves = PTGNN_SPLIT_WAVES;                     // 8 or 16 (one digit per thread needs >= 512 threads)
constexpr int kSplitThreads = kSplitWaves * 64;
constexpr int kSplitRounds = PTGNN_SPLIT_ROUNDS;                   // records per lane per sub-tile
constexpr int kBucketWaves = PTGNN_BUCKET_WAVES;
constexpr int kWaveRun = 64 * kSplitRounds;                        // consecutive records one wave ranks
constexpr int kSubTile = kSplitThreads * kSplitRounds;             // 4096 records
constexpr int kMaxTiles = 8192;     // rows of the aggregate table (8192 x 512 x 4 B = 16 MB at most): beyond that a
                                    // scatter workgroup walks several sub-tiles
constexpr int kPosBits = 22;        // 8-byte record: x = low row bits << 22 | position (E <= 4 M)
constexpr int kSmallLowBits = 9;    // 8-byte records: buckets of <= 512 rows
constexpr int kBigLowBits = 12;     // 12-byte records: buckets of <= 4096 rows

// The caller-owned control block of the C ABI (zero at rest).  Round 2's build kept digit totals and a tile
// counter in it; the totals now come out of k_tile_scan as plain stores, so nothing touches the block any more --
// it stays in the signature so that callers of the round-2 ABI keep working.
struct PlanControl {
  int32_t reserved[4];
};

// Workgroup barrier that orders LDS only: __syncthreads() also drains vmcnt, i.e. it would wait for the records
// prefetched for the next sub-tile and for the write-out stores still in flight.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// exclusive scan of v over ALL threads of the block (NW waves; tmp: NW ints of LDS); returns the exclusive prefix
template <int NW>
__device__ __forceinline__ int block_scan(int v, int *tmp) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  lds_barrier();                   // tmp may still be read by the previous scan's consumers
  if (lane == 63) tmp[wave] = inc;
  lds_barrier();
  int prior = 0;
  for (int w = 0; w < wave; ++w) prior += tmp[w];
  return prior + inc - v;
}

// lanes of this wave that hold the same digit (valid lanes only), as a 64-bit mask
__device__ __forceinline__ unsigned long long match_digit(bool valid, int digit, int bits) {
  unsigned long long same = __ballot(valid);
  for (int b = 0; b < bits; ++b) {
    const bool bit = (digit >> b) & 1;
    const unsigned long long bal = __ballot(bit);
    same &= bit ? bal : ~bal;
  }
  return same;
}

// One workgroup per SUB-tile (the scatter kernel walks the `subs` sub-tiles of a tile in sequence; the histogram
// has no such order to keep, and 8 x as many workgroups keep more loads in flight).  subs > 1: the sub-tiles of
// a tile add into its aggregate row with global atomics (the host zero-fills the table first).
template <bool LISTS>
__global__ __launch_bounds__(kSplitThreads) void k_split_count(TypeTable tab, Soa in, int mode, int total_types,
                                                               int64_t n, int shift, uint32_t mask, int bins,
                                                               int subs, int32_t *__restrict__ agg,
                                                               int32_t *hub_count, RangeGuard guard) {
  __shared__ int lh[kMaxBins];
  for (int j = threadIdx.x; j < bins; j += kSplitThreads) lh[j] = 0;
  if (hub_count && blockIdx.x == 0 && threadIdx.x == 0) *hub_count = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kSubTile;
  const int tile = (int)(blockIdx.x / (unsigned)subs);
  if (base < n) {
    TypeSpan span{0, 1};
    if constexpr (LISTS) span = tile_types(tab, base, (base + kSubTile < n ? base + kSubTile : n) - 1);
    uint32_t key[kSplitRounds];
#pragma unroll
    for (int r = 0; r < kSplitRounds; ++r) {
      const int64_t e = base + r * kSplitThreads + threadIdx.x;
      key[r] = 0u;
      if (e < n) {
        if constexpr (LISTS) key[r] = edge_key(tab, e, mode, total_types, guard, span);
        else key[r] = in.key[e];
      }
    }
#pragma unroll
    for (int r = 0; r < kSplitRounds; ++r)
      if (base + r * kSplitThreads + threadIdx.x < n) atomicAdd(&lh[(key[r] >> shift) & mask], 1);
  }
  __syncthreads();
  const int bp = (bins + 3) & ~3;                     // row stride of the aggregate table (dwordx4 reads)
  for (int j = threadIdx.x; j < bp; j += kSplitThreads) {
    const int c = j < bins ? lh[j] : 0;
    if (subs == 1) agg[(int64_t)tile * bp + j] = c;
    else if (c) atomicAdd(&agg[(int64_t)tile * bp + j], c);
  }
}

// Aggregate table: per-tile digit counts -> exclusive prefix over the tiles, in place (column-wise scan of
// [ntiles][bp]).  Round 2 had every scatter workgroup add up the rows of all earlier tiles itself: quadratic in the
// tile count, and up to nine dependent L2 round trips in the prologue of the late tiles.  One workgroup per 2
// digits (64-256 workgroups); thread (g, c) owns the rows [g R, (g+1) R) of column c.
constexpr int kScanCols = 2;
__global__ __launch_bounds__(1024) void k_tile_scan(int32_t *__restrict__ agg, int ntiles, int bp,
                                                    int32_t *__restrict__ totals /* [bp] column sums */) {
  constexpr int G = 1024 / kScanCols;             // row groups per column = 8 waves of consecutive threads
  __shared__ int wtot[16];
  const int c = threadIdx.x / G, g = threadIdx.x % G;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = blockIdx.x * kScanCols + c;
  const int R = (ntiles + G - 1) / G;
  const int r0 = g * R, r1 = r0 + R < ntiles ? r0 + R : ntiles;
  int sum = 0;
  if (col < bp) {
    for (int r = r0; r < r1; r += 8) {
      int v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = r + u < r1 ? agg[(int64_t)(r + u) * bp + col] : 0;
#pragma unroll
      for (int u = 0; u < 8; ++u) sum += v[u];
    }
  }
  // exclusive scan of the group sums along the column: within the wave by shuffles, across the column's 8 waves
  // through LDS
  int inc = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wtot[wave] = inc;
  __syncthreads();
  constexpr int WPC = G / 64;                     // waves per column
  int run = inc - sum;
  for (int w = c * WPC; w < wave; ++w) run += wtot[w];
  if (g == G - 1 && col < bp) totals[col] = run + sum;
  if (col < bp) {
    for (int r = r0; r < r1; r += 8) {
      int v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = r + u < r1 ? agg[(int64_t)(r + u) * bp + col] : 0;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (r + u < r1) agg[(int64_t)(r + u) * bp + col] = run;
        run += v[u];
      }
    }
  }
}

// where the scattered records go
constexpr int DST_SOA = 0;    // pre-pass: (key, position, payload) arrays
constexpr int DST_REC2 = 1;   // int2 (low row bits << 22 | position, payload)
constexpr int DST_REC3 = 2;   // int2 (key, payload) + int position
constexpr int DST_FINAL = 3;  // last LSD pass: col (payload), perm (position) and the sorted keys

struct SplitOut {
  Soa soa;               // DST_SOA; DST_FINAL: key = sorted keys, pos = perm (nullable), packed = col
  int2 *recs;
  int32_t *rpos;
  uint32_t low_mask;
};

template <bool LISTS, int DST>
__global__ __launch_bounds__(kSplitThreads) void k_split_scatter(TypeTable tab, Soa in, int32_t type_bits, int mode,
                                                                 int total_types, int64_t n, int shift,
                                                                 uint32_t mask, int dbits, int bins, int subs,
                                                                 const int32_t *__restrict__ totals,
                                                                 const int32_t *__restrict__ agg, SplitOut out,
                                                                 RangeGuard guard) {
  // The sub-tile in sorted order (key, payload, position): records leave through LDS so that the global stores of
  // one digit are contiguous runs -- written straight from the ranking lanes every record is its own partial-line
  // write request, and the request rate of the L2s, not bytes, bounded the kernel (profiles/r03_notes.md).  The
  // wave counter rows live in the same storage: they are dead once every lane holds its staging slot.
  constexpr int kStageInts = 3 * kSubTile > kSplitWaves * kMaxBins ? 3 * kSubTile : kSplitWaves * kMaxBins;
  __shared__ __attribute__((aligned(16))) int stage[kStageInts];
  __shared__ int cursor[kMaxBins];     // global position of the digit's next record
  __shared__ int lstart[kMaxBins];     // where the digit starts in the sub-tile's sorted order
  __shared__ int tmp[kSplitWaves];
  int *const wcnt = stage;
  uint32_t *const st_key = reinterpret_cast<uint32_t *>(stage);
  int32_t *const st_packed = stage + kSubTile;
  int32_t *const st_pos = stage + 2 * kSubTile;
  const int tile = blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int *const wrow = wcnt + wave * kMaxBins;
  int placed = 0;                                    // threads < bins: records of digit threadIdx.x in the previous sub-tile
  // wave w owns records [sbase + w * kWaveRun, + kWaveRun) of a sub-tile: round r = its r-th group of 64
  EdgeRec rec[kSplitRounds];
  int32_t opos[kSplitRounds];
  auto load_sub = [&](int sub) {                      // all loads of a sub-tile in flight together
    const int64_t sb = ((int64_t)tile * subs + sub) * kSubTile;
    const int64_t last = (sb + kSubTile < n ? sb + kSubTile : n) - 1;
    TypeSpan span{0, 1};
    if constexpr (LISTS) {
      if (sb < n) span = tile_types(tab, sb, last);
    }
#pragma unroll
    for (int r = 0; r < kSplitRounds; ++r) {
      const int64_t e = sb + (int64_t)wave * kWaveRun + lane + r * 64;
      rec[r] = EdgeRec{0u, 0};
      opos[r] = (int32_t)e;
      if (e < n) {
        if constexpr (LISTS) {
          rec[r] = edge_record(tab, e, type_bits, mode, total_types, guard, true, span);
        } else {
          rec[r].key = in.key[e];
          rec[r].packed = in.packed[e];
          opos[r] = in.pos[e];
        }
      }
    }
  };
  load_sub(0);                                       // in flight under the prologue's scan
  {   // where each bucket starts in the output: prefix of the digit totals (k_tile_scan) + the digit's records in
      // all earlier tiles: row `tile` of the aggregate table, which k_tile_scan has turned into exclusive prefixes
    const bool mine = (int)threadIdx.x < bins;
    const int v = mine ? totals[threadIdx.x] : 0;
    const int before = mine ? agg[(int64_t)tile * ((bins + 3) & ~3) + threadIdx.x] : 0;
    const int ex = block_scan<kSplitWaves>(v, tmp);
    if (mine) cursor[threadIdx.x] = ex + before;
  }
  __syncthreads();
  for (int s = 0; s < subs; ++s) {
    const int64_t sbase = ((int64_t)tile * subs + s) * kSubTile;
    if (sbase >= n) break;                            // workgroup-uniform
    const int nsub = (int)(n - sbase < kSubTile ? n - sbase : kSubTile);
    const int64_t wbase = sbase + (int64_t)wave * kWaveRun + lane;
    for (int j = lane; j < bins; j += 64) wrow[j] = 0;
    __builtin_amdgcn_wave_barrier();                  // DS ops of one wave execute in order; this pins the compiler
    int rk[kSplitRounds];
#pragma unroll
    for (int r = 0; r < kSplitRounds; ++r) {
      const bool valid = wbase + r * 64 < n;
      const int digit = (int)((rec[r].key >> shift) & mask);
      const unsigned long long same = match_digit(valid, digit, dbits);
      const int below = __popcll(same & ((1ull << lane) - 1ull));
      const int prior = valid ? wrow[digit] : 0;      // this wave's earlier rounds
      __builtin_amdgcn_wave_barrier();
      if (valid && below == 0) wrow[digit] = prior + __popcll(same);
      __builtin_amdgcn_wave_barrier();
      rk[r] = prior + below;
    }
    __syncthreads();   // every wave: ranking done, and done with the previous sub-tile's write-out
    int mine = 0;
    if ((int)threadIdx.x < bins) {   // wave rows -> offsets inside the digit; all counts first (independent LDS reads)
      cursor[threadIdx.x] += placed;
      int c[kSplitWaves];
#pragma unroll
      for (int w = 0; w < kSplitWaves; ++w) c[w] = wcnt[w * kMaxBins + threadIdx.x];
#pragma unroll
      for (int w = 0; w < kSplitWaves; ++w) {
        wcnt[w * kMaxBins + threadIdx.x] = mine;
        mine += c[w];
      }
      placed = mine;
    }
    {
      const int ex = block_scan<kSplitWaves>(mine, tmp);   // synchronises
      if ((int)threadIdx.x < bins) lstart[threadIdx.x] = ex;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kSplitRounds; ++r) {          // staging slot of every record, into registers
      const int digit = (int)((rec[r].key >> shift) & mask);
      if (wbase + r * 64 < n) rk[r] += lstart[digit] + wrow[digit];
    }
    __syncthreads();                                  // the counter rows are dead: their storage stages the records
#pragma unroll
    for (int r = 0; r < kSplitRounds; ++r) {
      if (wbase + r * 64 < n) {
        st_key[rk[r]] = rec[r].key;
        st_packed[rk[r]] = rec[r].packed;
        st_pos[rk[r]] = opos[r];
      }
    }
    if (s + 1 < subs) load_sub(s + 1);                // the next sub-tile's records travel under this write-out
    lds_barrier();
    for (int i = threadIdx.x; i < nsub; i += kSplitThreads) {   // sorted order: a digit's records are neighbours
      const uint32_t key = st_key[i];
      const int digit = (int)((key >> shift) & mask);
      const int64_t pos = (int64_t)cursor[digit] + (i - lstart[digit]);
      if constexpr (DST == DST_SOA) {
        out.soa.key[pos] = key;
        out.soa.pos[pos] = st_pos[i];
        out.soa.packed[pos] = st_packed[i];
      } else if constexpr (DST == DST_FINAL) {
        out.soa.key[pos] = key;
        if (out.soa.pos) out.soa.pos[pos] = st_pos[i];
        out.soa.packed[pos] = st_packed[i];
      } else if constexpr (DST == DST_REC2) {
        out.recs[pos] = make_int2((int)(((key & out.low_mask) << kPosBits) | (uint32_t)st_pos[i]), st_packed[i]);
      } else {
        out.recs[pos] = make_int2((int)key, st_packed[i]);
        out.rpos[pos] = st_pos[i];
      }
    }
    lds_barrier();     // the next sub-tile clears its counter rows inside the staging storage
  }
}

// ---- second level: one workgroup per bucket -----------------------------------------------------------------
// Stable counting sort of the bucket's records by the low row bits.  Records are ranked in super-chunks of
// NW x 512; inside one, wave w owns a contiguous run, counts it into its private counter row (packed 16-bit
// pairs: a run holds <= 512 records and a super-chunk <= 8192 at NW = 16, so neither the counts nor their
// prefix over the waves can carry into the neighbour), one pass over the rows turns them into offsets, and the
// wave ranks its run round by round against them.  BIG: 12-byte records, digit = (key >> pshift) & mask.
constexpr int kBucketRounds = 8;
constexpr int kStageCap = 5632;   // records of a bucket whose (col, perm) window is staged in LDS (8-byte path)

// arr[idx] += inc over the valid lanes of a wave (LDS).  When every valid lane hits the SAME counter -- the rounds
// of a hub row -- one lane adds the lot: 64 same-address LDS atomics serialise.
__device__ __forceinline__ void lds_count(uint32_t *arr, int idx, uint32_t inc, bool valid) {
  const unsigned long long vm = __ballot(valid);
  if (vm == 0ull) return;
  const int first = __builtin_amdgcn_readfirstlane(__shfl(idx, __ffsll((long long)vm) - 1, 64));
  if (__ballot(valid && idx != first) == 0ull) {
    if ((threadIdx.x & 63) == (unsigned)(__ffsll((long long)vm) - 1)) atomicAdd(&arr[first], inc * (uint32_t)__popcll(vm));
  } else if (valid) {
    atomicAdd(&arr[idx], inc);
  }
}

template <int NW, bool BIG>
__global__ __launch_bounds__(NW * 64) void k_plan_buckets(
    const int2 *__restrict__ recs, const int32_t *__restrict__ rpos, const int32_t *__restrict__ totals, int bins,
    int low_bits, int64_t num_rows, int64_t num_edges, int32_t *__restrict__ rowptr,
    int32_t *__restrict__ col, int32_t *__restrict__ perm, int32_t hub_threshold,
    int32_t hub_chunk, int32_t *__restrict__ hub_entries, int32_t *__restrict__ hub_count, int stage_cap) {
  extern __shared__ __attribute__((aligned(16))) uint32_t dyn[];
  constexpr int NT = NW * 64;
  constexpr int SC = NT * kBucketRounds;             // records per super-chunk
  constexpr int WPT = (1 << (kBigLowBits - 1)) / 512;  // counter words a thread may own (2048 words / 512 threads)
  __shared__ int tmp[NW];
  __shared__ int bucket_start_s, bucket_size_s;
  const int lbins = 1 << low_bits, lmask = lbins - 1;
  const int words = (lbins + 1) >> 1;
  uint32_t *const offs = dyn;                         // [lbins] degrees -> row starts -> + records placed so far
  uint32_t *const wcnt = dyn + lbins;                 // [NW][words] packed wave counters
  int32_t *const st_col = reinterpret_cast<int32_t *>(wcnt + NW * words);   // [stage_cap] the bucket's col window
  int32_t *const st_perm = st_col + stage_cap;                              // [stage_cap] ... and its perm window
  const int b = blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint32_t *const wrow = wcnt + wave * words;
  {   // where this bucket starts in the record array: prefix of the digit totals (bins <= 512 <= NT)
    const int v = (int)threadIdx.x < bins ? totals[threadIdx.x] : 0;
    const int ex = block_scan<NW>(v, tmp);
    if ((int)threadIdx.x == b) { bucket_start_s = ex; bucket_size_s = v; }
  }
  for (int j = threadIdx.x; j < lbins; j += NT) offs[j] = 0;
  __syncthreads();
  const int s = bucket_start_s, e = s + bucket_size_s;
  const bool staged = !BIG && e - s <= stage_cap;
  auto digit_of = [&](int2 r) -> int {
    return BIG ? (int)((uint32_t)r.x & (uint32_t)lmask) : (int)(((uint32_t)r.x >> kPosBits) & (uint32_t)lmask);
  };

  // records of one super-chunk, wave-striped: wave w owns [cb + w * run, + run), run a multiple of 64
  int2 rc[kBucketRounds];
  int32_t rp[BIG ? kBucketRounds : 1];
  int run = 0;
  auto load_chunk = [&](int cb) {
    const int left = e - cb < SC ? e - cb : SC;
    run = (((left + NW - 1) / NW) + 63) & ~63;
#pragma unroll
    for (int k = 0; k < kBucketRounds; ++k) {
      const int i = cb + wave * run + k * 64 + lane;
      const bool valid = k * 64 < run && i < e;
      rc[k] = valid ? recs[i] : make_int2(0, 0);
      if constexpr (BIG) rp[k] = valid ? rpos[i] : 0;
    }
  };
  auto valid_at = [&](int cb, int k) -> bool { return k * 64 < run && cb + wave * run + k * 64 + lane < e; };

  // in-degrees of the bucket's rows: the first super-chunk from registers (it is ranked from them below), the
  // rest of a long bucket in a strided pass
  load_chunk(s);
#pragma unroll
  for (int k = 0; k < kBucketRounds; ++k)
    if (k * 64 < run) lds_count(offs, digit_of(rc[k]), 1u, valid_at(s, k));
  for (int i0 = s + SC; i0 < e; i0 += 4 * NT) {     // four independent loads per thread in flight
    int2 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * NT + threadIdx.x;
      v[u] = i < e ? recs[i] : make_int2(0, 0);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) lds_count(offs, digit_of(v[u]), 1u, i0 + u * NT + (int)threadIdx.x < e);
  }
  __syncthreads();
  {   // degrees -> row starts (+ rowptr, hub rows); thread t owns the consecutive digits [t * per, (t+1) * per)
    const int per = (lbins + NT - 1) / NT;
    const int d0 = threadIdx.x * per;
    uint32_t deg[(1 << kBigLowBits) / 512];
    int sum = 0;
#pragma unroll
    for (int k = 0; k < (1 << kBigLowBits) / 512; ++k) {
      deg[k] = (k < per && d0 + k < lbins) ? offs[d0 + k] : 0u;
      sum += (int)deg[k];
    }
    int ex = block_scan<NW>(sum, tmp);       // synchronises: every degree has been read
#pragma unroll
    for (int k = 0; k < (1 << kBigLowBits) / 512; ++k) {
      if (k < per && d0 + k < lbins) {
        offs[d0 + k] = (uint32_t)ex;
        {
          const int64_t row = ((int64_t)b << low_bits) + d0 + k;
          if (row < num_rows) {
            rowptr[row] = s + ex;
            if (row == num_rows - 1) rowptr[num_rows] = (int32_t)num_edges;
            if (hub_entries && hub_threshold > 0 && (int)deg[k] > hub_threshold) {
              const int beg = s + ex, end = beg + (int)deg[k];
              const int c0 = beg / hub_chunk, c1 = (end - 1) / hub_chunk;
              const int at = atomicAdd(hub_count, c1 - c0 + 1);
              for (int c = c0; c <= c1; ++c) {
                hub_entries[2 * (at + c - c0)] = c;
                hub_entries[2 * (at + c - c0) + 1] = (int32_t)row;
              }
            }
          }
        }
        ex += (int)deg[k];
      }
    }
  }
  uint32_t pend[WPT];   // packed totals of the previous super-chunk, owned per counter word
#pragma unroll
  for (int k = 0; k < WPT; ++k) pend[k] = 0u;
  for (int cb = s; cb < e; cb += SC) {
    if (cb > s) load_chunk(cb);
    for (int j = lane; j < words; j += 64) wrow[j] = 0u;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < kBucketRounds; ++k) {
      if (k * 64 < run) {
        // packed pair: digit d counts in half (d & 1) of word d >> 1; (word, increment) is uniform iff d is
        const int d = digit_of(rc[k]);
        const bool valid = valid_at(cb, k);
        const unsigned long long vm = __ballot(valid);
        if (vm != 0ull) {
          const int lead = __ffsll((long long)vm) - 1;
          const int first = __builtin_amdgcn_readfirstlane(__shfl(d, lead, 64));
          if (__ballot(valid && d != first) == 0ull) {
            if (lane == lead) atomicAdd(&wrow[first >> 1], (uint32_t)__popcll(vm) << ((first & 1) * 16));
          } else if (valid) {
            atomicAdd(&wrow[d >> 1], 1u << ((d & 1) * 16));
          }
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < WPT; ++k) {   // counter word t + k * NT: both halves prefix at once over the waves
      const int word = threadIdx.x + k * NT;
      if (word < words) {
        if (pend[k]) {                 // rows advance by what the previous super-chunk placed
          offs[2 * word] += pend[k] & 0xffffu;
          if (2 * word + 1 < lbins) offs[2 * word + 1] += pend[k] >> 16;
        }
        uint32_t c[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) c[w] = wcnt[w * words + word];
        uint32_t acc = 0u;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          wcnt[w * words + word] = acc;
          acc += c[w];
        }
        pend[k] = acc;
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kBucketRounds; ++k) {
      if (k * 64 < run) {              // wave-uniform
        const bool valid = valid_at(cb, k);
        const int d = digit_of(rc[k]);
        const unsigned long long same = match_digit(valid, d, low_bits);
        const int below = __popcll(same & ((1ull << lane) - 1ull));
        const int sh = (d & 1) * 16;
        const uint32_t cur = valid ? (wrow[d >> 1] >> sh) & 0xffffu : 0u;
        __builtin_amdgcn_wave_barrier();
        if (valid && below == 0) atomicAdd(&wrow[d >> 1], (uint32_t)__popcll(same) << sh);
        __builtin_amdgcn_wave_barrier();
        if (valid) {
          const int at = (int)offs[d] + (int)cur + below;     // slot inside the bucket
          if constexpr (BIG) {
            col[s + at] = rc[k].y;
            if (perm) perm[s + at] = rp[k];
          } else {
            if (staged) {
              st_col[at] = rc[k].y;
              st_perm[at] = rc[k].x & ((1 << kPosBits) - 1);
            } else {
              col[s + at] = rc[k].y;
              if (perm) perm[s + at] = rc[k].x & ((1 << kPosBits) - 1);
            }
          }
        }
      }
    }
    // no barrier: the next super-chunk clears and counts wave-private rows only, and `offs` moves in its prefix
    // step, behind the barrier every wave reaches after this ranking step
  }
  if constexpr (!BIG) {
    if (staged) {   // the bucket's window leaves as whole lines instead of one 4-byte request per record and array
      __syncthreads();
      for (int i = threadIdx.x; i < e - s; i += NT) {
        col[s + i] = st_col[i];
        if (perm) perm[s + i] = st_perm[i];
      }
    }
  }
}

// rowptr from the sorted keys (plans with LSD pre-passes, whose buckets hold several rows per low digit)
__global__ __launch_bounds__(256) void k_rowptr_from_keys(const uint32_t *__restrict__ keys_sorted, int64_t num_edges,
                                                          int64_t num_rows, int32_t *__restrict__ rowptr) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= num_edges) return;
  const int64_t k = keys_sorted[i];
  const int64_t kprev = (i == 0) ? -1 : (int64_t)keys_sorted[i - 1];
  for (int64_t v = kprev + 1; v <= k; ++v) rowptr[v] = (int32_t)i;  // rows (kprev, k] start here
  if (i == num_edges - 1)
    for (int64_t v = k + 1; v <= num_rows; ++v) rowptr[v] = (int32_t)num_edges;
}

__global__ __launch_bounds__(256) void k_zero_rowptr(int32_t *__restrict__ rowptr, int64_t num_rows) {
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v <= num_rows; v += (int64_t)gridDim.x * blockDim.x)
    rowptr[v] = 0;
}

// (chunk, row) pairs of every row longer than `threshold`, appended in arbitrary order (consumers
// treat the pairs independently); *count must be 0 on entry.
__global__ __launch_bounds__(256) void k_hub_list(const int32_t *__restrict__ rowptr, int64_t num_rows,
                                                  int32_t threshold, int32_t chunk,
                                                  int32_t *__restrict__ entries,
                                                  int32_t *__restrict__ count) {
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < num_rows;
       r += (int64_t)gridDim.x * blockDim.x) {
    const int beg = rowptr[r], end = rowptr[r + 1];
    if (end - beg <= threshold) continue;
    const int c0 = beg / chunk, c1 = (end - 1) / chunk;
    const int base = atomicAdd(count, c1 - c0 + 1);
    for (int c = c0; c <= c1; ++c) {
      entries[2 * (base + c - c0)] = c;
      entries[2 * (base + c - c0) + 1] = (int32_t)r;
    }
  }
}

__global__ __launch_bounds__(256) void k_max_degree(const int32_t *__restrict__ rowptr, int64_t num_rows,
                                                    int32_t *__restrict__ out) {
  int best = 0;
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < num_rows;
       r += (int64_t)gridDim.x * blockDim.x) {
    const int d = rowptr[r + 1] - rowptr[r];
    best = d > best ? d : best;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int other = __shfl_xor(best, o, 64);
    best = other > best ? other : best;
  }
  if ((threadIdx.x & 63) == 0 && best > 0) atomicMax(out, best);
}

__global__ __launch_bounds__(256) void k_validate(const int64_t *__restrict__ idx, int64_t n,
                                                  int64_t num_nodes, int32_t *bad) {
  int local = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = idx[i];
    local += (v < 0 || v >= num_nodes);
  }
  if (local) atomicAdd(bad, local);
}

// ---- host side ----------------------------------------------------------------------------------------------
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

int end_bit_for(int64_t num_rows) {
  int b = 1;
  while (((int64_t)1 << b) < num_rows) ++b;
  return b;
}

std::atomic<int> g_force_path{0};   // test switch (relaxed; results never depend on it), ptgnn_amd_set_plan_path: 0 auto, 1 force the wide-record MSD form, 2 force the LSD form

// How a build of (edges, rows) is split into passes.
//   MSD (<= 4 M edges, <= 21 row bits): first level on the high bits, k_plan_buckets on the low bits -- the lists
//        are read 1.5 times, records written and read once;
//   LSD (everything larger): ceil(bits / 9) passes of the same two kernels from the low digit up, the last one
//        writing col / perm / sorted keys, rowptr from the keys.  Every pass is tiled over the INPUT order, so a
//        power-law hub (one row holding > 1 % of 12.5 M edges) costs nothing extra; the MSD form leaves such a row
//        to the one workgroup that owns its bucket (measured: 0.66 ms for that workgroup alone).
struct PlanPath {
  int bits;           // row id bits
  bool lsd;
  bool big;           // MSD: 12-byte records (row bits > 18)
  int low_bits;       // MSD second level: rows per bucket = 2^low_bits
  int npass;          // split passes (MSD: 1)
  int width[4];       // LSD: bits of each pass, lowest first
  int bins;           // buckets of the last (MSD: the only) split pass
  int subs;           // sub-tiles per workgroup of the scatter kernel
  int64_t ntiles;
  size_t bucket_lds;  // dynamic LDS of k_plan_buckets
};

PlanPath choose_path(int64_t num_edges, int64_t num_rows) {
  PlanPath P{};
  P.bits = end_bit_for(num_rows);
  P.subs = (int)((num_edges + (int64_t)kSubTile * kMaxTiles - 1) / ((int64_t)kSubTile * kMaxTiles));
  if (P.subs < 1) P.subs = 1;
  P.ntiles = (num_edges + (int64_t)kSubTile * P.subs - 1) / ((int64_t)kSubTile * P.subs);
  const bool msd_ok = num_edges <= ((int64_t)1 << kPosBits) && P.bits <= 9 + kBigLowBits;
  P.lsd = !msd_ok || g_force_path == 2;
  if (P.lsd) {
    P.npass = (P.bits + 8) / 9;
    int left = P.bits;
    for (int i = 0; i < P.npass; ++i) {
      P.width[i] = (left + (P.npass - i) - 1) / (P.npass - i);
      left -= P.width[i];
    }
    const int top_shift = P.bits - P.width[P.npass - 1];
    P.bins = (int)((num_rows + (((int64_t)1 << top_shift) - 1)) >> top_shift);
    if (P.bins < 1) P.bins = 1;
    return P;
  }
  P.npass = 1;
  P.big = P.bits > 2 * kSmallLowBits || g_force_path == 1;
  const int lmax = P.big ? kBigLowBits : kSmallLowBits;
  // buckets of ~4096 edges on average keep every CU busy in the second level
  int l = 0;
  while (l < lmax && l < P.bits && (((int64_t)2 << l) * num_edges <= (int64_t)4096 * num_rows)) ++l;
  if (l < 3) l = P.bits < 3 ? P.bits : 3;
  if (P.bits - l > 9) l = P.bits - 9;
  P.low_bits = l;
  P.bins = (int)((num_rows + (((int64_t)1 << l) - 1)) >> l);
  if (P.bins < 1) P.bins = 1;
  const size_t lbins = (size_t)1 << l;
  P.bucket_lds = 4 * lbins + (size_t)kBucketWaves * 4 * ((lbins + 1) / 2) + (P.big ? 0 : (size_t)kStageCap * 8);
  return P;
}

struct WsLayout {
  size_t soa_a, soa_b, recs, rpos, agg, control, total;
};

void layout(int64_t num_edges, const PlanPath &P, WsLayout *L) {
  const size_t e4 = align_up((size_t)num_edges * 4, 256);
  size_t o = 0;
  L->soa_a = o; o += 3 * e4;                           // k_pack (> 64 edge types) / LSD ping
  L->soa_b = o; o += P.lsd ? 3 * e4 : 0;               // LSD pong
  L->recs = o;  o += P.lsd ? e4 : 2 * e4;              // MSD: records; LSD: the sorted keys
  L->rpos = o;  o += (!P.lsd && P.big) ? e4 : 0;
  L->agg = o;   o += align_up((size_t)(P.ntiles > 0 ? P.ntiles : 1) * kMaxBins * 4, 256);
  L->control = o; o += align_up((size_t)kMaxBins * 4, 256);  // digit totals of the current pass
  L->total = o + 256;
}

// Raise a kernel's dynamic-LDS limit once per (device, kernel): the attribute call is not legal while a stream is
// being captured into a hipGraph, and the warm-up launch outside the capture has made it.
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

}  // namespace
}  // namespace ptgnn_amd

using namespace ptgnn_amd;

extern "C" int ptgnn_amd_type_bits(int32_t num_types) {
  int b = 0;
  while ((1 << b) < num_types) ++b;
  return b;
}

extern "C" size_t ptgnn_amd_csr_control_bytes(void) { return sizeof(PlanControl); }

extern "C" int ptgnn_amd_set_plan_path(int path) {
  PTGNN_REQUIRE(path >= 0 && path <= 2, PTGNN_AMD_EINVAL, "set_plan_path: 0 auto, 1 wide-record MSD, 2 LSD");
  g_force_path = path;
  return PTGNN_AMD_OK;
}

extern "C" size_t ptgnn_amd_csr_workspace_bytes(int64_t num_edges, int64_t num_nodes) {
  if (num_edges < 0 || num_nodes < 0) return 0;
  WsLayout L;
  layout(num_edges, choose_path(num_edges, num_nodes), &L);
  return L.total;
}

extern "C" int ptgnn_amd_csr_build(const int64_t *const *src_per_type,
                                   const int64_t *const *dst_per_type,
                                   const int64_t *edges_per_type, int32_t num_types,
                                   int64_t num_nodes, int64_t num_src_rows, int swap_src_dst,
                                   int32_t *rowptr,
                                   int32_t *col, int32_t *perm, int32_t *max_degree,
                                   int32_t hub_threshold, int32_t *hub_entries, int32_t *hub_count,
                                   int32_t *bad_index_count, void *control,
                                   void *workspace, size_t workspace_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PTGNN_REQUIRE(num_types >= 0 && num_nodes >= 0, PTGNN_AMD_EINVAL, "csr_build: negative size");
  PTGNN_REQUIRE(rowptr != nullptr, PTGNN_AMD_EINVAL, "csr_build: rowptr is null");
  PTGNN_REQUIRE(num_types == 0 || (src_per_type && dst_per_type && edges_per_type),
                PTGNN_AMD_EINVAL, "csr_build: null type tables");
  int64_t num_edges = 0;
  for (int t = 0; t < num_types; ++t) {
    PTGNN_REQUIRE(edges_per_type[t] >= 0, PTGNN_AMD_EINVAL, "csr_build: negative edge count");
    PTGNN_REQUIRE(edges_per_type[t] == 0 || (src_per_type[t] && dst_per_type[t]),
                  PTGNN_AMD_EINVAL, "csr_build: null adjacency list for type %d", t);
    num_edges += edges_per_type[t];
  }
  PTGNN_REQUIRE(swap_src_dst >= 0 && swap_src_dst <= 2, PTGNN_AMD_EINVAL, "csr_build: bad mode");
  const int type_bits = ptgnn_amd_type_bits(num_types);
  // mode 2: `num_nodes` is the number of plan rows = source rows * num_types (caller passes it so)
  const int64_t src_rows = num_src_rows > num_nodes ? num_src_rows : num_nodes;
  PTGNN_REQUIRE(num_edges < ((int64_t)1 << 31) && num_nodes < ((int64_t)1 << 31) &&
                    (src_rows << (swap_src_dst == 2 ? 0 : type_bits)) < ((int64_t)1 << 31),
                PTGNN_AMD_EUNSUPPORTED,
                "csr_build: num_edges=%lld / rows=%lld / source rows=%lld x 2^%d exceed the int32 plan format",
                (long long)num_edges, (long long)num_nodes, (long long)src_rows, type_bits);
  PTGNN_REQUIRE(num_edges == 0 || col != nullptr, PTGNN_AMD_EINVAL, "csr_build: col is null");
  const RangeGuard guard{num_nodes, swap_src_dst == 0 ? src_rows : 0, bad_index_count};
  const RangeGuard no_guard{num_nodes, 0, nullptr};
  const bool hubs = hub_entries && hub_count && hub_threshold > 0;

  if (num_edges == 0) {
    const int64_t zb = (num_nodes + 256) / 256;
    k_zero_rowptr<<<(unsigned)(zb < 1024 ? zb : 1024), 256, 0, stream>>>(rowptr, num_nodes);
    PTGNN_LAUNCH_CHECK();
    if (hubs) PTGNN_HIP(hipMemsetAsync(hub_count, 0, sizeof(int32_t), stream));
    if (max_degree) PTGNN_HIP(hipMemsetAsync(max_degree, 0, sizeof(int32_t), stream));
    return PTGNN_AMD_OK;
  }

  const PlanPath P = choose_path(num_edges, num_nodes);
  WsLayout L;
  layout(num_edges, P, &L);
  PTGNN_REQUIRE(workspace_bytes >= L.total && workspace, PTGNN_AMD_EWORKSPACE,
                "csr_build: workspace %zu < required %zu", workspace_bytes, L.total);
  char *ws = (char *)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  const size_t e4 = align_up((size_t)num_edges * 4, 256);
  auto soa_at = [&](size_t off) {
    Soa t;
    t.key = (uint32_t *)(ws + off);
    t.pos = (int32_t *)(ws + off + e4);
    t.packed = (int32_t *)(ws + off + 2 * e4);
    return t;
  };
  const Soa soa[2] = {soa_at(L.soa_a), soa_at(L.soa_b)};
  int32_t *agg = (int32_t *)(ws + L.agg);
  int32_t *totals = (int32_t *)(ws + L.control);   // [<= 512] digit totals of the current pass (k_tile_scan)
  (void)control;

  TypeTable tab{};
  int cur = -1;     // index of the Soa that holds the current records (-1: the lists)
  if (num_types <= kMaxTypes) {
    tab.num_types = num_types;
    for (int t = 0; t < num_types; ++t) {
      tab.src[t] = src_per_type[t];
      tab.dst[t] = dst_per_type[t];
      tab.offset[t + 1] = tab.offset[t] + edges_per_type[t];
    }
  } else {
    // more edge types than one kernel-argument table holds: narrow the lists into (key, position, payload)
    // arrays chunk by chunk; the passes below then read those
    int64_t base = 0;
    for (int t0 = 0; t0 < num_types; t0 += kMaxTypes) {
      TypeTable tt{};
      tt.num_types = (num_types - t0 < kMaxTypes) ? (num_types - t0) : kMaxTypes;
      tt.type_base = t0;
      for (int t = 0; t < tt.num_types; ++t) {
        tt.src[t] = src_per_type[t0 + t];
        tt.dst[t] = dst_per_type[t0 + t];
        tt.offset[t + 1] = tt.offset[t] + edges_per_type[t0 + t];
      }
      const int64_t chunk = tt.offset[tt.num_types];
      if (chunk > 0) {
        const int64_t blocks = (chunk + 255) / 256;
        k_pack<<<(unsigned)(blocks < 4096 ? blocks : 4096), 256, 0, stream>>>(tt, type_bits, swap_src_dst, num_types,
                                                                              soa[0], base, guard);
        PTGNN_LAUNCH_CHECK();
      }
      base += chunk;
    }
    cur = 0;
  }

  const unsigned grid = (unsigned)P.ntiles;
  const unsigned cgrid = (unsigned)(P.ntiles * P.subs);           // the count kernel: one workgroup per sub-tile
  // one split pass: count -> tile scan -> scatter
  auto split = [&](int pass, int shift, uint32_t mask, int dbits, int nb, int dst, const SplitOut &out) -> int {
    int32_t *hc = pass == 0 && hubs ? hub_count : nullptr;
    const int bp = (nb + 3) & ~3;
    if (P.subs > 1) PTGNN_HIP(hipMemsetAsync(agg, 0, (size_t)P.ntiles * bp * 4, stream));
    if (cur < 0)
      k_split_count<true><<<cgrid, kSplitThreads, 0, stream>>>(tab, Soa{}, swap_src_dst, num_types, num_edges, shift,
                                                              mask, nb, P.subs, agg, hc, guard);
    else
      k_split_count<false><<<cgrid, kSplitThreads, 0, stream>>>(tab, soa[cur], swap_src_dst, num_types, num_edges,
                                                               shift, mask, nb, P.subs, agg, hc, no_guard);
    PTGNN_LAUNCH_CHECK();
    k_tile_scan<<<(unsigned)((bp + kScanCols - 1) / kScanCols), 1024, 0, stream>>>(agg, (int)P.ntiles, bp, totals);
    PTGNN_LAUNCH_CHECK();
#define PTGNN_SCATTER(DST_)                                                                                           \
  do {                                                                                                                \
    if (cur < 0)                                                                                                      \
      k_split_scatter<true, DST_><<<grid, kSplitThreads, 0, stream>>>(tab, Soa{}, type_bits, swap_src_dst, num_types, \
                                                                     num_edges, shift, mask, dbits, nb, P.subs,      \
                                                                     totals, agg, out, guard);                       \
    else                                                                                                              \
      k_split_scatter<false, DST_><<<grid, kSplitThreads, 0, stream>>>(tab, soa[cur], type_bits, swap_src_dst,       \
                                                                      num_types, num_edges, shift, mask, dbits, nb,  \
                                                                      P.subs, totals, agg, out, no_guard);           \
  } while (0)
    switch (dst) {
      case DST_SOA: PTGNN_SCATTER(DST_SOA); break;
      case DST_REC2: PTGNN_SCATTER(DST_REC2); break;
      case DST_REC3: PTGNN_SCATTER(DST_REC3); break;
      default: PTGNN_SCATTER(DST_FINAL); break;
    }
#undef PTGNN_SCATTER
    PTGNN_LAUNCH_CHECK();
    return PTGNN_AMD_OK;
  };
  auto bits_for = [](int nb) { int d = 0; while ((1 << d) < nb) ++d; return d; };

  if (P.lsd) {
    uint32_t *keys_sorted = (uint32_t *)(ws + L.recs);
    int shift = 0;
    for (int i = 0; i < P.npass; ++i) {
      const bool last = i == P.npass - 1;
      SplitOut out{};
      int rc;
      if (!last) {
        const int dstidx = cur == 0 ? 1 : 0;
        out.soa = soa[dstidx];
        rc = split(i, shift, (1u << P.width[i]) - 1u, P.width[i], 1 << P.width[i], DST_SOA, out);
        if (rc != PTGNN_AMD_OK) return rc;
        cur = dstidx;
      } else {
        out.soa.key = keys_sorted;
        out.soa.pos = perm;
        out.soa.packed = col;
        rc = split(i, shift, 0xffffffffu, bits_for(P.bins), P.bins, DST_FINAL, out);
        if (rc != PTGNN_AMD_OK) return rc;
      }
      shift += P.width[i];
    }
    const int64_t blocks = (num_edges + 255) / 256;
    k_rowptr_from_keys<<<(unsigned)blocks, 256, 0, stream>>>(keys_sorted, num_edges, num_nodes, rowptr);
    PTGNN_LAUNCH_CHECK();
    if (hubs && num_edges > hub_threshold) {
      const int64_t hb = (num_nodes + 255) / 256;
      k_hub_list<<<(unsigned)(hb < 2048 ? hb : 2048), 256, 0, stream>>>(rowptr, num_nodes, hub_threshold, 1024,
                                                                        hub_entries, hub_count);
      PTGNN_LAUNCH_CHECK();
    }
  } else {
    SplitOut out{};
    out.recs = (int2 *)(ws + L.recs);
    out.rpos = (int32_t *)(ws + L.rpos);
    out.low_mask = (1u << P.low_bits) - 1u;
    const int rc = split(0, P.low_bits, 0xffffffffu, bits_for(P.bins), P.bins, P.big ? DST_REC3 : DST_REC2, out);
    if (rc != PTGNN_AMD_OK) return rc;
    const unsigned bgrid = (unsigned)P.bins;
    const int hub_thr = hubs ? hub_threshold : 0;
    if (P.big) {
      auto kern = k_plan_buckets<kBucketWaves, true>;
      PTGNN_REQUIRE(set_lds(kern, P.bucket_lds), PTGNN_AMD_EHIP, "csr_build: %zu B of LDS refused", P.bucket_lds);
      kern<<<bgrid, kBucketWaves * 64, P.bucket_lds, stream>>>(out.recs, out.rpos, totals, P.bins, P.low_bits, num_nodes,
                                                              num_edges, rowptr, col, perm, hub_thr, 1024, hub_entries,
                                                              hub_count, 0);
    } else {
      k_plan_buckets<kBucketWaves, false><<<bgrid, kBucketWaves * 64, P.bucket_lds, stream>>>(
          out.recs, nullptr, totals, P.bins, P.low_bits, num_nodes, num_edges, rowptr, col, perm, hub_thr, 1024,
          hub_entries, hub_count, kStageCap);
    }
    PTGNN_LAUNCH_CHECK();
  }
  if (max_degree) {
    PTGNN_HIP(hipMemsetAsync(max_degree, 0, sizeof(int32_t), stream));
    const int64_t mb = (num_nodes + 255) / 256;
    k_max_degree<<<(unsigned)(mb < 1024 ? mb : 1024), 256, 0, stream>>>(rowptr, num_nodes, max_degree);
    PTGNN_LAUNCH_CHECK();
  }
  return PTGNN_AMD_OK;
}

extern "C" int ptgnn_amd_hub_list(const int32_t *rowptr, int64_t num_rows, int32_t hub_threshold,
                                   int32_t *hub_entries, int32_t *hub_count, void *stream_) {
  PTGNN_REQUIRE(rowptr && hub_entries && hub_count && num_rows >= 0 && hub_threshold >= 2048,
                PTGNN_AMD_EINVAL, "hub_list: bad arguments");
  if (num_rows == 0) return PTGNN_AMD_OK;
  const int64_t hb = (num_rows + 255) / 256;
  k_hub_list<<<(unsigned)(hb < 2048 ? hb : 2048), 256, 0, (hipStream_t)stream_>>>(
      rowptr, num_rows, hub_threshold, 1024, hub_entries, hub_count);
  PTGNN_LAUNCH_CHECK();
  return PTGNN_AMD_OK;
}

extern "C" int ptgnn_amd_validate_indices(const int64_t *idx, int64_t n, int64_t num_nodes,
                                          int32_t *bad_count, void *stream_) {
  PTGNN_REQUIRE(n >= 0 && bad_count && (n == 0 || idx), PTGNN_AMD_EINVAL, "validate: bad args");
  if (n == 0) return PTGNN_AMD_OK;
  const int64_t blocks = (n + 255) / 256;
  k_validate<<<(unsigned)(blocks < 2048 ? blocks : 2048), 256, 0, (hipStream_t)stream_>>>(
      idx, n, num_nodes, bad_count);
  PTGNN_LAUNCH_CHECK();
  return PTGNN_AMD_OK;
}
