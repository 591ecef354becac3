// Graph plan construction: per-edge-type int64 adjacency lists -> one destination-sorted CSR.
// See include/ptgnn_amd.h (ptgnn_amd_csr_build) for the contract and the reference lines replaced.
//
// Pipeline (all on `stream`, no host sync):
//   1. k_pack   : walk the T lists (pointer table in the kernel argument segment), narrow to
//                 int32, emit key = row (dst), payload = packed (src << type_bits | type) and the
//                 edge position in the type-major concatenation.
//   2. stable LSD radix sort of (key, position) over ceil(log2(rows)) bits, <= 9 bits per pass
//      (18-bit node ids = 2 passes).  Hand-written for the sizes that matter here: rocPRIM's onesweep
//      sort runs 1.1 M pairs as ~140 long-running workgroups (29 us per pass on MI355X, latency-bound
//      with half the CUs idle); these kernels use one workgroup per 1024 pairs (one pair per lane):
//        k_radix_hist    per-workgroup digit histogram            -> hist[digit][workgroup]
//        exclusive scan  of the digit-major histogram (rocPRIM device scan: plumbing, 1 small launch)
//        k_radix_scatter stable rank = earlier waves' count (LDS) + same-digit lanes below (ballots)
//      PTGNN_AMD_SORT=rocprim selects the library sort instead (A/B + fallback).
//   3. k_finish : payload gather into CSR order + rowptr from key boundaries.
// HBM-bound integer work: 8 B/edge read once, O(passes * 16 B/edge) inside the sort.
#include <stdlib.h>
#include <string.h>

#include <rocprim/rocprim.hpp>

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

__global__ __launch_bounds__(256) void k_pack(TypeTable tab, int32_t type_bits, int mode, int total_types,
                                              uint32_t *__restrict__ keys,
                                              int32_t *__restrict__ pos,
                                              int32_t *__restrict__ packed, int64_t pos_base, RangeGuard guard) {
  const int64_t total = tab.offset[tab.num_types];
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const EdgeRec r = edge_record(tab, e, type_bits, mode, total_types, guard, true, TypeSpan{0, tab.num_types});
    const int64_t g = pos_base + e;
    pos[g] = (int32_t)g;
    keys[g] = r.key;
    packed[g] = r.packed;
  }
}

__global__ __launch_bounds__(256) void k_finish(const uint32_t *__restrict__ keys_sorted,
                                                const int32_t *__restrict__ pos_sorted,
                                                const int32_t *__restrict__ packed,
                                                int64_t num_edges, int64_t num_nodes,
                                                int32_t *__restrict__ rowptr,
                                                int32_t *__restrict__ col,
                                                int32_t *__restrict__ perm) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (num_edges == 0) {
    for (int64_t v = i; v <= num_nodes; v += (int64_t)gridDim.x * blockDim.x) rowptr[v] = 0;
    return;
  }
  if (i >= num_edges) return;
  const int32_t p = pos_sorted[i];
  col[i] = packed[p];
  if (perm) perm[i] = p;
  const int64_t k = keys_sorted[i];
  const int64_t kprev = (i == 0) ? -1 : (int64_t)keys_sorted[i - 1];
  for (int64_t v = kprev + 1; v <= k; ++v) rowptr[v] = (int32_t)i;  // rows (kprev, k] start here
  if (i == num_edges - 1)
    for (int64_t v = k + 1; v <= num_nodes; ++v) rowptr[v] = (int32_t)num_edges;
}

// ---- stable LSD radix sort of (key, value) pairs, one pair per lane, 1024 pairs per workgroup ------
constexpr int kSortBlock = 1024;
constexpr int kMaxBins = 512;

__global__ __launch_bounds__(kSortBlock) void k_radix_hist(const uint32_t *__restrict__ keys, int64_t n,
                                                           int shift, int bits,
                                                           int32_t *__restrict__ hist, int64_t nblocks) {
  __shared__ int lh[kMaxBins];
  const int bins = 1 << bits;
  for (int j = threadIdx.x; j < bins; j += kSortBlock) lh[j] = 0;
  __syncthreads();
  const int64_t i = blockIdx.x * (int64_t)kSortBlock + threadIdx.x;
  if (i < n) atomicAdd(&lh[(keys[i] >> shift) & (bins - 1)], 1);
  __syncthreads();
  for (int j = threadIdx.x; j < bins; j += kSortBlock) hist[(int64_t)j * nblocks + blockIdx.x] = lh[j];
}

__global__ __launch_bounds__(kSortBlock) void k_radix_scatter(
    const uint32_t *__restrict__ keys_in, const int32_t *__restrict__ vals_in,
    uint32_t *__restrict__ keys_out, int32_t *__restrict__ vals_out, int64_t n, int shift, int bits,
    const int32_t *__restrict__ offs /* scanned hist */, int64_t nblocks) {
  __shared__ int wave_cnt[(kSortBlock / 64) * kMaxBins];
  const int bins = 1 << bits;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int j = threadIdx.x; j < (kSortBlock / 64) * bins; j += kSortBlock) wave_cnt[j] = 0;
  const int64_t i = blockIdx.x * (int64_t)kSortBlock + threadIdx.x;
  const bool valid = i < n;
  const uint32_t key = valid ? keys_in[i] : 0u;
  const int32_t val = valid ? vals_in[i] : 0;
  const int digit = (int)((key >> shift) & (uint32_t)(bins - 1));
  // lanes of this wave that hold the same digit (ballots are wave-wide, 64 bits)
  unsigned long long same = __ballot(valid);
  for (int b = 0; b < bits; ++b) {
    const bool bit = (digit >> b) & 1;
    const unsigned long long bal = __ballot(bit);
    same &= bit ? bal : ~bal;
  }
  const int rank = __popcll(same & ((1ull << lane) - 1ull));
  __syncthreads();                                     // wave_cnt is zeroed
  if (valid && rank == 0) wave_cnt[wave * bins + digit] = __popcll(same);
  __syncthreads();
  for (int d = threadIdx.x; d < bins; d += kSortBlock) {   // exclusive prefix over the 16 waves
    int run = 0;
    for (int w = 0; w < kSortBlock / 64; ++w) {
      const int t = wave_cnt[w * bins + d];
      wave_cnt[w * bins + d] = run;
      run += t;
    }
  }
  __syncthreads();
  if (valid) {
    const int64_t pos = (int64_t)offs[(int64_t)digit * nblocks + blockIdx.x] + wave_cnt[wave * bins + digit] + rank;
    keys_out[pos] = key;
    vals_out[pos] = val;
  }
}

// ---- two-level plan build for minibatch-sized graphs (rows <= 2^18, <= 64 edge types, <= 4 M edges) --
// The LSD sort above needs 2 x (histogram, scan, scatter) + pack + finish + hub list = 11 dependent
// launches whose 4-byte scatters land two-at-a-time in random cache lines.  Rows are node ids, i.e.
// roughly uniformly populated, so the plan is built MSD-first in THREE launches:
//   k_plan_count    per 4096-edge tile: LDS histogram of the HIGH row bits (reads only the key column of the
//                   int64 lists), stored as the tile's row of the aggregate table; digit totals by one global
//                   atomic per (tile, digit) into the control block (4 replicas against same-address contention)
//   k_plan_scatter  one pass over the lists: stable scatter of 8-byte records (low row bits | position,
//                   payload) into <= 512 buckets of 2^low_bits consecutive rows.  The cross-tile prefix of every
//                   digit is the column sum of the aggregate rows of all earlier tiles (complete: previous
//                   launch): all 1024 threads, dwordx4, 32 rows per step, plain cached loads -- no flags, no
//                   inter-workgroup communication or ordering.  Quadratic in the tile count, which the 4 M-edge
//                   ceiling of this path bounds at 1024 (72 MB of L2 reads at 1.1 M edges).  A decoupled
//                   look-back over (flag | value) status words was measured against it: its agent-scope loads go
//                   to the memory side on this multi-L2 part (~1 us per dependent step) and it was never faster
//                   (625 k edges / 153 tiles: 18.0 vs 15.7 us per launch; 1.1 M edges / 269 tiles: 28.2 vs 28.5)
//   k_plan_buckets  one workgroup per bucket: histogram of the LOW bits = the in-degrees -> rowptr and
//                   the hub list directly; stable counting sort of the bucket into col / perm; the last
//                   workgroup to have read the control block zeroes it again
// Keys are never materialised, the second level works inside a few-KiB window of the output, and the edge
// lists are read 1.5 times (8 + 16 B/edge) against 2 x 16 B/edge + a scan launch for the histogram/scan/scatter
// form this replaces.  Stability of both levels = the order of a numpy stable argsort (tests: bit-exact).
//
// Control block (PlanControl, ptgnn_amd_csr_control_bytes()): digit totals + one counter, ZERO AT REST -- the
// caller zero-fills it once, hands it to every build on ONE stream, and finds it zero-filled again after each
// build.  A null control pointer makes the library carve one out of the workspace and zero it with a memset
// node per build.
constexpr int kMsdBlock = 1024;
constexpr int kTileRounds = 4;
constexpr int kTileEdges = kMsdBlock * kTileRounds;   // edges per tile of k_plan_count / k_plan_scatter
constexpr int kPosBits = 22;                          // record.x = low row bits << 22 | position (E <= 4 M)

constexpr int kTotalReplicas = 4;
struct PlanControl {
  int32_t totals[kTotalReplicas][kMaxBins];   // digit totals, replica = tile % 4
  int32_t done;                               // k_plan_buckets workgroups that have finished reading `totals`
  int32_t pad[3];
};

__device__ __forceinline__ int digit_total(const PlanControl *ctl, int d) {
  int v = 0;
#pragma unroll
  for (int r = 0; r < kTotalReplicas; ++r) v += ctl->totals[r][d];
  return v;
}

// exclusive scan of v over threads 0 .. 511 of a 1024-thread block (tmp: 8 ints of LDS); returns the
// exclusive prefix
__device__ __forceinline__ int block_scan_512(int v, int *tmp) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (wave < 8 && lane == 63) tmp[wave] = inc;
  __syncthreads();
  int prior = 0;
  if (wave < 8)
    for (int w = 0; w < wave; ++w) prior += tmp[w];
  return prior + inc - v;
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

__global__ __launch_bounds__(kMsdBlock) void k_plan_count(TypeTable tab, int mode, int total_types, int64_t n,
                                                          int low_bits, int bins, PlanControl *ctl,
                                                          int32_t *__restrict__ agg, int32_t *hub_count,
                                                          RangeGuard guard) {
  __shared__ int lh[kMaxBins];
  for (int j = threadIdx.x; j < bins; j += kMsdBlock) lh[j] = 0;
  if (hub_count && blockIdx.x == 0 && threadIdx.x == 0) *hub_count = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kTileEdges;
  const TypeSpan span = tile_types(tab, base, (base + kTileEdges < n ? base + kTileEdges : n) - 1);
#pragma unroll
  for (int r = 0; r < kTileRounds; ++r) {
    const int64_t e = base + r * kMsdBlock + threadIdx.x;
    if (e < n) atomicAdd(&lh[edge_key(tab, e, mode, total_types, guard, span) >> low_bits], 1);
  }
  __syncthreads();
  const int bp = (bins + 3) & ~3;                     // row stride of the aggregate table (dwordx4 reads)
  for (int j = threadIdx.x; j < bp; j += kMsdBlock) {
    const int c = j < bins ? lh[j] : 0;
    agg[(int64_t)blockIdx.x * bp + j] = c;
    if (c) atomicAdd(&ctl->totals[blockIdx.x % kTotalReplicas][j], c);
  }
}

// stable rank of this lane's digit inside a 1024-thread block: (earlier waves' count, rank in wave)
// via wave ballots + a [16][bins] LDS table; returns the block-local exclusive rank contribution
// wave_cnt[wave][digit] + rank (valid lanes only) and leaves per-digit block totals in `run_out`
// for threads < bins.
__device__ __forceinline__ int block_stable_rank(bool valid, int digit, int bits, int bins, int *wave_cnt,
                                                 int *run_out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int j = threadIdx.x; j < (kMsdBlock / 64) * bins; j += kMsdBlock) wave_cnt[j] = 0;
  unsigned long long same = __ballot(valid);
  for (int b = 0; b < bits; ++b) {
    const bool bit = (digit >> b) & 1;
    const unsigned long long bal = __ballot(bit);
    same &= bit ? bal : ~bal;
  }
  const int rank = __popcll(same & ((1ull << lane) - 1ull));
  __syncthreads();
  if (valid && rank == 0) wave_cnt[wave * bins + digit] = __popcll(same);
  __syncthreads();
  int run = 0;
  if (threadIdx.x < bins) {     // bins <= 512 < block: one digit per thread
    // all 16 counts first (independent LDS reads in flight together), then the prefix: written as a
    // read-modify-write loop the accesses form a 16-deep dependent chain of LDS round trips
    constexpr int W = kMsdBlock / 64;
    int c[W];
#pragma unroll
    for (int w = 0; w < W; ++w) c[w] = wave_cnt[w * bins + threadIdx.x];
#pragma unroll
    for (int w = 0; w < W; ++w) {
      wave_cnt[w * bins + threadIdx.x] = run;
      run += c[w];
    }
  }
  *run_out = run;
  __syncthreads();
  return valid ? wave_cnt[wave * bins + digit] + rank : 0;
}

// 8 waves per SIMD = two 16-wave workgroups per CU (<= 64 VGPRs): with one, the tiles beyond 256 run as a second
// round and double the kernel time at cfg2's 269 tiles
__global__ __launch_bounds__(kMsdBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_plan_scatter(TypeTable tab, int32_t type_bits, int mode,
                                                            int total_types, int64_t n, int low_bits,
                                                            int high_bits, int bins, const PlanControl *ctl,
                                                            const int32_t *__restrict__ agg,
                                                            int2 *__restrict__ recs, RangeGuard guard) {
  __shared__ __attribute__((aligned(16))) int wave_cnt[(kMsdBlock / 64) * kMaxBins];
  __shared__ int base[kMaxBins];
  __shared__ int tile_excl[kMaxBins];
  __shared__ int round_base[kTileRounds][kMaxBins];
  __shared__ int tmp[8];
  const int tile = blockIdx.x;
  {   // where each bucket starts in the record array: prefix of the digit totals (complete: previous launch)
    const int v = threadIdx.x < bins ? digit_total(ctl, threadIdx.x) : 0;
    const int ex = block_scan_512(v, tmp);
    if (threadIdx.x < bins) base[threadIdx.x] = ex;
  }
  const uint32_t low_mask = (1u << low_bits) - 1u;
  EdgeRec rec[kTileRounds];
  const TypeSpan span = tile_types(tab, (int64_t)tile * kTileEdges,
                                   ((int64_t)(tile + 1) * kTileEdges < n ? (int64_t)(tile + 1) * kTileEdges : n) - 1);
#pragma unroll
  for (int r = 0; r < kTileRounds; ++r) {            // issued first: in flight under the prefix sums
    const int64_t e = (int64_t)tile * kTileEdges + r * kMsdBlock + threadIdx.x;
    rec[r] = EdgeRec{0u, 0};
    if (e < n) rec[r] = edge_record(tab, e, type_bits, mode, total_types, guard, true, span);
  }
  {
    // exclusive cross-tile prefix of every digit = column sums of the aggregate rows of tiles 0 .. tile-1:
    // thread (g, q) adds rows g, g+8, ... for digits 4q .. 4q+3; the 8 partials meet in LDS (the rank table's
    // storage, not yet in use)
    const int bp = (bins + 3) & ~3;
    const int q = threadIdx.x & 127, g = threadIdx.x >> 7;
    int4 acc = make_int4(0, 0, 0, 0);
    if (4 * q < bp) {
#pragma unroll 4
      for (int r = g; r < tile; r += 8) {
        const int4 v = *reinterpret_cast<const int4 *>(agg + (int64_t)r * bp + 4 * q);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
      *reinterpret_cast<int4 *>(wave_cnt + g * kMaxBins + 4 * q) = acc;
    }
    __syncthreads();
    if (threadIdx.x < bins) {
      int e8 = 0;
#pragma unroll
      for (int gg = 0; gg < 8; ++gg) e8 += wave_cnt[gg * kMaxBins + threadIdx.x];
      tile_excl[threadIdx.x] = e8;
    }
    __syncthreads();                                 // the rank rounds clear wave_cnt next
  }
  int local[kTileRounds];
  int mine = 0;                                      // threads < bins: this tile's count of digit threadIdx.x
#pragma unroll
  for (int r = 0; r < kTileRounds; ++r) {
    const int64_t e = (int64_t)tile * kTileEdges + r * kMsdBlock + threadIdx.x;
    int run;
    local[r] = block_stable_rank(e < n, (int)(rec[r].key >> low_bits), high_bits, bins, wave_cnt, &run);
    if (threadIdx.x < bins) {
      round_base[r][threadIdx.x] = mine;
      mine += run;
    }
    __syncthreads();                                 // wave_cnt is cleared again by the next round
  }
#pragma unroll
  for (int r = 0; r < kTileRounds; ++r) {
    const int64_t e = (int64_t)tile * kTileEdges + r * kMsdBlock + threadIdx.x;
    if (e < n) {
      const int digit = (int)(rec[r].key >> low_bits);
      const int64_t pos = (int64_t)base[digit] + tile_excl[digit] + round_base[r][digit] + local[r];
      recs[pos] = make_int2((int)(((rec[r].key & low_mask) << kPosBits) | (uint32_t)e), rec[r].packed);
    }
  }
}

__global__ __launch_bounds__(kMsdBlock) void k_plan_buckets(
    const int2 *__restrict__ recs, PlanControl *ctl, int bins, int low_bits,
    int64_t num_rows, int64_t num_edges, int32_t *__restrict__ rowptr, int32_t *__restrict__ col,
    int32_t *__restrict__ perm, int32_t hub_threshold, int32_t hub_chunk, int32_t *__restrict__ hub_entries,
    int32_t *__restrict__ hub_count) {
  __shared__ int wave_cnt[(kMsdBlock / 64) * kMaxBins];
  __shared__ int offs[kMaxBins];
  __shared__ int tmp[8];
  __shared__ int bucket_start_s, bucket_size_s, last_s;
  const int b = blockIdx.x;
  const int lbins = 1 << low_bits, mask = lbins - 1;
  {   // where this bucket starts in the record array: prefix of the digit totals
    const int v = threadIdx.x < bins ? digit_total(ctl, threadIdx.x) : 0;
    const int ex = block_scan_512(v, tmp);
    if (threadIdx.x == b) { bucket_start_s = ex; bucket_size_s = v; }
  }
  for (int j = threadIdx.x; j < lbins; j += kMsdBlock) offs[j] = 0;
  __syncthreads();
  // every thread of this workgroup has its totals in registers: the last workgroup to get here puts the
  // control block back to its zero-at-rest state
  if (threadIdx.x == 0) last_s = atomicAdd(&ctl->done, 1) == (int)gridDim.x - 1;
  const int s = bucket_start_s, e = s + bucket_size_s;
  // the first 4096 records of the bucket (all of it, as a rule) stay in registers for both passes: one memory
  // round trip with four loads in flight instead of two passes of dependent ones
  constexpr int KC = kTileRounds;
  int2 rc[KC];
#pragma unroll
  for (int k = 0; k < KC; ++k) {
    const int i = s + k * kMsdBlock + threadIdx.x;
    rc[k] = i < e ? recs[i] : make_int2(0, 0);
  }
#pragma unroll
  for (int k = 0; k < KC; ++k)
    if (s + k * kMsdBlock + (int)threadIdx.x < e) atomicAdd(&offs[(rc[k].x >> kPosBits) & mask], 1);
  for (int i = s + KC * kMsdBlock + threadIdx.x; i < e; i += kMsdBlock)
    atomicAdd(&offs[(recs[i].x >> kPosBits) & mask], 1);
  __syncthreads();
  if (last_s) {
    for (int j = threadIdx.x; j < kTotalReplicas * kMaxBins; j += kMsdBlock) (&ctl->totals[0][0])[j] = 0;
    if (threadIdx.x == 0) ctl->done = 0;
  }
  {   // in-degrees -> rowptr (+ hub rows); offs becomes the running write cursor of each row
    const int deg = threadIdx.x < lbins ? offs[threadIdx.x] : 0;
    __syncthreads();
    const int ex = block_scan_512(deg, tmp);
    const int64_t row = ((int64_t)b << low_bits) + threadIdx.x;
    if (threadIdx.x < lbins) {
      offs[threadIdx.x] = ex;
      if (row < num_rows) {
        rowptr[row] = s + ex;
        if (row == num_rows - 1) rowptr[num_rows] = (int32_t)num_edges;
        if (hub_entries && hub_threshold > 0 && deg > hub_threshold) {
          const int beg = s + ex, end = beg + deg;
          const int c0 = beg / hub_chunk, c1 = (end - 1) / hub_chunk;
          const int at = atomicAdd(hub_count, c1 - c0 + 1);
          for (int c = c0; c <= c1; ++c) {
            hub_entries[2 * (at + c - c0)] = c;
            hub_entries[2 * (at + c - c0) + 1] = (int32_t)row;
          }
        }
      }
    }
  }
  __syncthreads();
  auto place = [&](bool valid, int2 r) {   // one 1024-record chunk, in record order
    const int digit = (r.x >> kPosBits) & mask;
    int run;
    const int local = block_stable_rank(valid, digit, low_bits, lbins, wave_cnt, &run);   // syncs inside
    if (valid) {
      const int pos = s + offs[digit] + local;
      col[pos] = r.y;
      if (perm) perm[pos] = r.x & ((1 << kPosBits) - 1);
    }
    __syncthreads();
    if (threadIdx.x < lbins) offs[threadIdx.x] += run;
    // the next chunk's block_stable_rank synchronises before offs is read again
  };
#pragma unroll
  for (int k = 0; k < KC; ++k)
    if (s + k * kMsdBlock < e) place(s + k * kMsdBlock + (int)threadIdx.x < e, rc[k]);   // workgroup-uniform test
  for (int cb = s + KC * kMsdBlock; cb < e; cb += kMsdBlock) {
    const int i = cb + threadIdx.x;
    place(i < e, i < e ? recs[i] : make_int2(0, 0));
  }
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

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct WsLayout {
  size_t keys_in, keys_out, pos_in, pos_out, packed, sort_tmp, sort_tmp_bytes, hist, hist_scan, control, total;
};

// The one-pair-per-lane kernels win up to a few million edges (minibatch sizes: 0.11 vs 0.15 ms at
// 1.1 M, 0.085 vs 0.16 ms at 0.6 M edges); beyond that rocPRIM's many-items-per-thread onesweep
// coalesces its scatters better (0.67 vs 0.80 ms at 12.5 M).  PTGNN_AMD_SORT=rocprim|custom forces one.
bool use_rocprim_sort(int64_t num_edges) {
  static int v = -1;
  if (v < 0) {
    const char *e = getenv("PTGNN_AMD_SORT");
    v = !e ? 0 : (strcmp(e, "rocprim") == 0 ? 1 : (strcmp(e, "custom") == 0 || strcmp(e, "lsd") == 0 ? 2 : 0));
  }
  if (v == 1) return true;
  if (v == 2 || v == 3) return false;
  return num_edges > ((int64_t)4 << 20);
}

// two-level build: needs every row id in 18 bits (high digit <= 9 bits over <= 9-bit buckets) and one
// type table; PTGNN_AMD_SORT=lsd forces the flat LSD sort for A/B runs
bool use_msd_build(int64_t num_edges, int64_t num_rows, int num_types) {
  const char *e = getenv("PTGNN_AMD_SORT");
  if (e && strcmp(e, "msd") != 0) return false;
  return num_edges > 0 && num_edges <= ((int64_t)4 << 20) && num_rows <= ((int64_t)1 << 18) &&
         num_types <= kMaxTypes;
}

int end_bit_for(int64_t num_nodes) {
  int b = 1;
  while (((int64_t)1 << b) < num_nodes) ++b;
  return b;
}

hipError_t sort_tmp_bytes(int64_t num_edges, int64_t num_nodes, size_t *bytes) {
  *bytes = 0;
  if (num_edges == 0) return hipSuccess;
  size_t a = 0, b = 0;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, a, (uint32_t *)nullptr, (uint32_t *)nullptr,
                                           (int32_t *)nullptr, (int32_t *)nullptr, (size_t)num_edges, 0,
                                           end_bit_for(num_nodes), (hipStream_t)0);
  if (e != hipSuccess) return e;
  const size_t nh = (size_t)kMaxBins * (size_t)((num_edges + kSortBlock - 1) / kSortBlock);
  e = rocprim::exclusive_scan(nullptr, b, (int32_t *)nullptr, (int32_t *)nullptr, 0, nh,
                              rocprim::plus<int32_t>(), (hipStream_t)0);
  *bytes = a > b ? a : b;
  return e;
}

bool layout(int64_t num_edges, int64_t num_nodes, WsLayout *L) {
  size_t tmp = 0;
  if (sort_tmp_bytes(num_edges, num_nodes, &tmp) != hipSuccess) return false;
  const size_t e4 = align_up((size_t)num_edges * 4, 256);
  size_t o = 0;
  L->keys_in = o;  o += e4;
  L->keys_out = o; o += e4;
  L->pos_in = o;   o += e4;
  L->pos_out = o;  o += e4;
  L->packed = o;   o += e4;
  L->sort_tmp = o; o += align_up(tmp, 256);
  L->sort_tmp_bytes = tmp;
  const size_t nh = align_up((size_t)kMaxBins * (size_t)((num_edges + kSortBlock - 1) / kSortBlock) * 4, 256);
  L->hist = o;      o += nh;
  L->hist_scan = o; o += nh;
  L->control = o;   o += align_up(sizeof(PlanControl), 256);   // only used when the caller passes no control block
  L->total = o + 256;
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

extern "C" size_t ptgnn_amd_csr_workspace_bytes(int64_t num_edges, int64_t num_nodes) {
  if (num_edges < 0 || num_nodes < 0) return 0;
  WsLayout L;
  if (!layout(num_edges, num_nodes, &L)) return 0;
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
  PTGNN_REQUIRE(num_edges < ((int64_t)1 << 31) &&
                    (src_rows << (swap_src_dst == 2 ? 0 : type_bits)) < ((int64_t)1 << 31),
                PTGNN_AMD_EUNSUPPORTED,
                "csr_build: num_edges=%lld / source rows=%lld x 2^%d exceed the int32 plan format",
                (long long)num_edges, (long long)src_rows, type_bits);
  PTGNN_REQUIRE(num_edges == 0 || col != nullptr, PTGNN_AMD_EINVAL, "csr_build: col is null");
  const RangeGuard guard{num_nodes, swap_src_dst == 0 ? src_rows : 0, bad_index_count};
  WsLayout L;
  PTGNN_REQUIRE(layout(num_edges, num_nodes, &L), PTGNN_AMD_EHIP, "csr_build: sort size query failed");
  PTGNN_REQUIRE(workspace_bytes >= L.total && (workspace || L.total == 0), PTGNN_AMD_EWORKSPACE,
                "csr_build: workspace %zu < required %zu", workspace_bytes, L.total);
  char *ws = (char *)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  uint32_t *keys_in = (uint32_t *)(ws + L.keys_in), *keys_out = (uint32_t *)(ws + L.keys_out);
  int32_t *pos_in = (int32_t *)(ws + L.pos_in), *pos_out = (int32_t *)(ws + L.pos_out);
  int32_t *packed = (int32_t *)(ws + L.packed);

  if (use_msd_build(num_edges, num_nodes, num_types)) {
    TypeTable tab;
    tab.num_types = num_types;
    tab.type_base = 0;
    tab.offset[0] = 0;
    for (int t = 0; t < num_types; ++t) {
      tab.src[t] = src_per_type[t];
      tab.dst[t] = dst_per_type[t];
      tab.offset[t + 1] = tab.offset[t] + edges_per_type[t];
    }
    const int total_bits = end_bit_for(num_nodes);
    // buckets of ~4096 edges on average keep every CU busy in the second level; <= 9 bits per level
    int low_bits = 0;
    while (low_bits < 9 && low_bits < total_bits &&
           (((int64_t)2 << low_bits) * num_edges <= (int64_t)4096 * num_nodes)) ++low_bits;
    if (low_bits < 3) low_bits = total_bits < 3 ? total_bits : 3;
    if (total_bits - low_bits > 9) low_bits = total_bits - 9;
    const int high_bits = total_bits - low_bits;
    const int bins = (int)((num_nodes + ((int64_t)1 << low_bits) - 1) >> low_bits);
    const int64_t ntiles = (num_edges + kTileEdges - 1) / kTileEdges;
    int32_t *agg = (int32_t *)(ws + L.hist);        // [ntiles][bins rounded up to 4] per-tile digit counts
    int2 *recs = (int2 *)ws;      // 8 B/edge over the (unused) key/pos/payload buffers of the LSD path
    PlanControl *ctl = (PlanControl *)control;
    if (ctl == nullptr) {         // no caller-owned control block: one inside the workspace, zeroed per build
      ctl = (PlanControl *)(ws + L.control);
      PTGNN_HIP(hipMemsetAsync(ctl, 0, sizeof(PlanControl), stream));
    }
    const bool hubs = hub_entries && hub_count && hub_threshold > 0;
    k_plan_count<<<(unsigned)ntiles, kMsdBlock, 0, stream>>>(tab, swap_src_dst, num_types, num_edges, low_bits, bins,
                                                            ctl, agg, hubs ? hub_count : nullptr, guard);
    PTGNN_LAUNCH_CHECK();
    k_plan_scatter<<<(unsigned)ntiles, kMsdBlock, 0, stream>>>(tab, type_bits, swap_src_dst, num_types, num_edges,
                                                              low_bits, high_bits, bins, ctl, agg, recs, guard);
    PTGNN_LAUNCH_CHECK();
    k_plan_buckets<<<(unsigned)bins, kMsdBlock, 0, stream>>>(recs, ctl, bins, low_bits, num_nodes, num_edges,
                                                             rowptr, col, perm, hubs ? hub_threshold : 0, 1024,
                                                             hub_entries, hub_count);
    PTGNN_LAUNCH_CHECK();
    if (max_degree) {
      PTGNN_HIP(hipMemsetAsync(max_degree, 0, sizeof(int32_t), stream));
      const int64_t mb = (num_nodes + 255) / 256;
      k_max_degree<<<(unsigned)(mb < 1024 ? mb : 1024), 256, 0, stream>>>(rowptr, num_nodes, max_degree);
      PTGNN_LAUNCH_CHECK();
    }
    return PTGNN_AMD_OK;
  }
  if (num_edges > 0) {
    int64_t base = 0;
    for (int t0 = 0; t0 < num_types; t0 += kMaxTypes) {
      TypeTable tab;
      tab.num_types = (num_types - t0 < kMaxTypes) ? (num_types - t0) : kMaxTypes;
      tab.type_base = t0;
      tab.offset[0] = 0;
      for (int t = 0; t < tab.num_types; ++t) {
        tab.src[t] = src_per_type[t0 + t];
        tab.dst[t] = dst_per_type[t0 + t];
        tab.offset[t + 1] = tab.offset[t] + edges_per_type[t0 + t];
      }
      const int64_t chunk = tab.offset[tab.num_types];
      if (chunk > 0) {
        const int64_t blocks = (chunk + 255) / 256;
        k_pack<<<(unsigned)(blocks < 4096 ? blocks : 4096), 256, 0, stream>>>(
            tab, type_bits, swap_src_dst, num_types, keys_in, pos_in, packed, base, guard);
        PTGNN_LAUNCH_CHECK();
      }
      base += chunk;
    }
    size_t tmp = L.sort_tmp_bytes;
    if (use_rocprim_sort(num_edges)) {
      PTGNN_HIP(rocprim::radix_sort_pairs(ws + L.sort_tmp, tmp, keys_in, keys_out, pos_in, pos_out,
                                          (size_t)num_edges, 0, end_bit_for(num_nodes), stream));
    } else {
      const int total_bits = end_bit_for(num_nodes);
      const int passes = (total_bits + 8) / 9;
      const int bits = (total_bits + passes - 1) / passes;          // <= 9
      const int64_t nblocks = (num_edges + kSortBlock - 1) / kSortBlock;
      int32_t *hist = (int32_t *)(ws + L.hist), *hscan = (int32_t *)(ws + L.hist_scan);
      uint32_t *ka = keys_in, *kb = keys_out;
      int32_t *va = pos_in, *vb = pos_out;
      for (int p = 0; p < passes; ++p) {
        const int shift = p * bits;
        k_radix_hist<<<(unsigned)nblocks, kSortBlock, 0, stream>>>(ka, num_edges, shift, bits, hist, nblocks);
        PTGNN_LAUNCH_CHECK();
        size_t stmp = L.sort_tmp_bytes;
        PTGNN_HIP(rocprim::exclusive_scan(ws + L.sort_tmp, stmp, hist, hscan, 0,
                                          (size_t)((int64_t)(1 << bits) * nblocks),
                                          rocprim::plus<int32_t>(), stream));
        k_radix_scatter<<<(unsigned)nblocks, kSortBlock, 0, stream>>>(ka, va, kb, vb, num_edges, shift, bits,
                                                                      hscan, nblocks);
        PTGNN_LAUNCH_CHECK();
        uint32_t *tk = ka; ka = kb; kb = tk;
        int32_t *tv = va; va = vb; vb = tv;
      }
      keys_out = ka;   // the sorted pairs live in whichever buffer the last pass wrote
      pos_out = va;
    }
  }
  const int64_t work = num_edges > 0 ? num_edges : 1;
  const int64_t blocks = num_edges > 0 ? (work + 255) / 256 : 64;
  k_finish<<<(unsigned)blocks, 256, 0, stream>>>(keys_out, pos_out, packed, num_edges, num_nodes,
                                                 rowptr, col, perm);
  PTGNN_LAUNCH_CHECK();
  if (hub_entries && hub_count && hub_threshold > 0) {
    PTGNN_HIP(hipMemsetAsync(hub_count, 0, sizeof(int32_t), stream));
    if (num_edges > hub_threshold && num_nodes > 0) {
      const int64_t hb = (num_nodes + 255) / 256;
      k_hub_list<<<(unsigned)(hb < 2048 ? hb : 2048), 256, 0, stream>>>(rowptr, num_nodes, hub_threshold,
                                                                        1024, hub_entries, hub_count);
      PTGNN_LAUNCH_CHECK();
    }
  }
  if (max_degree) {
    PTGNN_HIP(hipMemsetAsync(max_degree, 0, sizeof(int32_t), stream));
    if (num_edges > 0 && num_nodes > 0) {
      const int64_t mb = (num_nodes + 255) / 256;
      k_max_degree<<<(unsigned)(mb < 1024 ? mb : 1024), 256, 0, stream>>>(rowptr, num_nodes, max_degree);
      PTGNN_LAUNCH_CHECK();
    }
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
