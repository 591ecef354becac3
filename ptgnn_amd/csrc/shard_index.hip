// Per-minibatch index bookkeeping of a destination-range shard (ptgnn_amd/sharded.py, SURVEY.md 8e / 8f-3 "halo send
// lists"): which remote source rows this rank needs (de-duplicated, sorted, hence grouped by owner), and the edge
// endpoints remapped into the local table [own rows | halo rows].  The reference has no counterpart (its multi-GPU mode
// is whole-batch data parallelism, distributedtrainer.py:250-297); this replaces the chain of ~20 torch launches
// (index_fill_ / cumsum / nonzero / where / gathers over bool + int32 arrays of the GLOBAL node count) that round 2
// used -- at BASELINE config 4 it cost more than the layers it prepared.
//
//   k_shard_mark     one pass over the edge lists: a BIT per remote source id (atomicOr into a bitmap of the global id
//                    space: 1.25 MB at 10 M nodes, where round 2 held a bool + an int32 per node), local ids of the
//                    destinations and of the own sources, own-source edges per edge type
//   k_shard_blocks   population count per 1024-word block of the bitmap
//   k_shard_compact  per block: exclusive prefix of the set bits -> slot of every word, the sorted id list
//   k_shard_counts   halo rows per owner range (slot(bounds[p+1]) - slot(bounds[p])) + totals for the ONE host read-back
//   k_shard_remap    second pass over the edge lists: remote source -> n_local + its halo slot
// HBM-bound integer work; nothing here synchronises with the host.
#include "common.h"
#include "stream_gemm.h"

namespace ptgnn_amd {
namespace {

constexpr int kShardTypes = 64;
constexpr int kBlockWords = 256;     // bitmap words per workgroup of the rank kernels: one word per thread

struct ShardTable {
  const int64_t *src[kShardTypes];
  const int64_t *dst[kShardTypes];
  int64_t offset[kShardTypes + 1];
  int32_t num_types;
  int32_t type_base;
};

__device__ __forceinline__ int shard_type_of(const ShardTable &tab, int64_t e) {
  int lo = 0, hi = tab.num_types;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (tab.offset[mid] <= e) lo = mid; else hi = mid;
  }
  return lo;
}

// stats layout (int64): [0, world) halo rows per owner | [world] remote-source edges | [world + 1] halo rows in all |
//                       [world + 2, world + 2 + T) own-source edges per edge type
__global__ __launch_bounds__(256) void k_shard_mark(ShardTable tab, int64_t edge_base, int64_t lo, int64_t hi,
                                                    int64_t total_nodes, uint32_t *__restrict__ bitmap,
                                                    int64_t *__restrict__ local_src, int64_t *__restrict__ local_dst,
                                                    unsigned long long *__restrict__ stats, int world,
                                                    int32_t *__restrict__ bad_index_count) {
  __shared__ int own_cnt[kShardTypes];
  __shared__ int remote_cnt;
  __shared__ int bad_cnt;
  if (threadIdx.x < kShardTypes) own_cnt[threadIdx.x] = 0;
  if (threadIdx.x == 0) { remote_cnt = 0; bad_cnt = 0; }
  __syncthreads();
  const int64_t n = tab.offset[tab.num_types];
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int t = shard_type_of(tab, e);
    const int64_t i = e - tab.offset[t];
    int64_t s = tab.src[t][i];
    const int64_t d = tab.dst[t][i];
    // A global source id outside [0, total_nodes) must not leave the bitmap: it is clamped -- and COUNTED here, because
    // after the remap it is a valid own or halo row that the plan build's range guard can no longer tell from a good one
    // (the reference device-asserts on it in F.embedding, gatedmessagepassing.py:54-56).  Destinations outside [lo, hi)
    // become local ids outside [0, hi - lo), which the plan build does count.
    const bool bad = s < 0 || s >= total_nodes;
    if (bad) atomicAdd(&bad_cnt, 1);
    s = s < 0 ? 0 : (s < total_nodes ? s : total_nodes - 1);
    local_dst[edge_base + e] = d - lo;
    const bool own = s >= lo && s < hi;
    if (own) local_src[edge_base + e] = s - lo;
    else atomicOr(&bitmap[s >> 5], 1u << (s & 31));
    // counters: one LDS atomic per wave when the wave sits inside one edge type (the rule), else one per lane
    const unsigned long long own_m = __ballot(own), rem_m = __ballot(!own);
    const int t0 = __builtin_amdgcn_readfirstlane(t);
    const bool leader = (threadIdx.x & 63) == (unsigned)(__ffsll((long long)(own_m | rem_m)) - 1);
    if (__ballot(t != t0) == 0ull) {
      if (leader && own_m) atomicAdd(&own_cnt[t0], __popcll(own_m));
    } else if (own) {
      atomicAdd(&own_cnt[t], 1);
    }
    if (leader && rem_m) atomicAdd(&remote_cnt, __popcll(rem_m));
  }
  __syncthreads();
  if ((int)threadIdx.x < tab.num_types && own_cnt[threadIdx.x])
    atomicAdd(&stats[world + 2 + tab.type_base + threadIdx.x], (unsigned long long)own_cnt[threadIdx.x]);
  if (threadIdx.x == 0 && remote_cnt) atomicAdd(&stats[world], (unsigned long long)remote_cnt);
  if (threadIdx.x == 0 && bad_cnt && bad_index_count) atomicAdd(bad_index_count, bad_cnt);
}

__global__ __launch_bounds__(256) void k_shard_blocks(const uint32_t *__restrict__ bitmap, int64_t words,
                                                      int32_t *__restrict__ block_sum) {
  __shared__ int part[4];
  const int64_t w0 = (int64_t)blockIdx.x * kBlockWords;
  int c = 0;
#pragma unroll
  for (int k = 0; k < kBlockWords / 256; ++k) {
    const int64_t w = w0 + k * 256 + threadIdx.x;
    c += w < words ? __popc(bitmap[w]) : 0;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) block_sum[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// one workgroup per kBlockWords-word block; thread t owns word t of the block (at one word per thread the serial part --
// a store per set bit -- is 4x shorter than with the four words a thread of the first version owned: 19.4 -> 8.6 us on
// the 2 M-bit map of BASELINE config 3's (type, source) pairs, every fourth bit set)
// KEY_MOD: the ids are (edge type, source) keys t * num_src_rows + src of the shared-message bookkeeping and leave as the
// source id (what the grouped GEMM gathers by) -- round 3 rewrote the list in a launch of its own (k_uniq_sources)
template <bool KEY_MOD>
__global__ __launch_bounds__(256) void k_shard_compact(const uint32_t *__restrict__ bitmap, int64_t words,
                                                       const int32_t *__restrict__ block_sum,
                                                       int32_t *__restrict__ word_slot, int64_t *__restrict__ need_ids,
                                                       int64_t num_src_rows, int64_t capacity) {
  static_assert(kBlockWords == 256, "one bitmap word per thread");
  __shared__ int wsum[4];
  __shared__ int base_s;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  {   // slots in front of this block = sum of the earlier blocks' counts
    int c = 0;
    for (int j = threadIdx.x; j < (int)blockIdx.x; j += 256) c += block_sum[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if (lane == 0) wsum[wave] = c;
    __syncthreads();
    if (threadIdx.x == 0) base_s = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
  const int64_t w = (int64_t)blockIdx.x * kBlockWords + threadIdx.x;
  uint32_t bits = w < words ? bitmap[w] : 0u;
  const int mine = __popc(bits);
  int inc = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  __syncthreads();                       // base_s / wsum of the prefix above are read by now
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int run = base_s + inc - mine;
  for (int v = 0; v < wave; ++v) run += wsum[v];
  if (w < words) {
    word_slot[w] = run;
    while (bits) {
      const int b = __ffs((int)bits) - 1;
      bits &= bits - 1;
      int64_t id = w * 32 + b;
      if constexpr (KEY_MOD) {
        id -= (id / num_src_rows) * num_src_rows;
        if (run >= capacity) break;        // never happens for a plan's own keys (capacity = min(E, rows * T))
      }
      need_ids[run++] = id;
    }
  }
}

__device__ __forceinline__ int64_t slot_of(const uint32_t *bitmap, const int32_t *word_slot, int64_t words,
                                           int64_t id, int64_t total_marked) {
  const int64_t w = id >> 5;
  if (w >= words) return total_marked;
  return (int64_t)word_slot[w] + __popc(bitmap[w] & ((1u << (id & 31)) - 1u));
}

__global__ __launch_bounds__(64) void k_shard_counts(const uint32_t *__restrict__ bitmap,
                                                     const int32_t *__restrict__ word_slot, int64_t words,
                                                     const int32_t *__restrict__ block_sum, int nblocks,
                                                     const int64_t *__restrict__ bounds, int world,
                                                     unsigned long long *__restrict__ stats) {
  int c = 0;
  for (int j = threadIdx.x; j < nblocks; j += 64) c += block_sum[j];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  const int64_t total_marked = c;
  if (threadIdx.x == 0) stats[world + 1] = (unsigned long long)total_marked;
  for (int p = threadIdx.x; p < world; p += 64) {
    const int64_t a = slot_of(bitmap, word_slot, words, bounds[p], total_marked);
    const int64_t b = slot_of(bitmap, word_slot, words, bounds[p + 1], total_marked);
    stats[p] = (unsigned long long)(b - a);
  }
}

__global__ __launch_bounds__(256) void k_shard_remap(ShardTable tab, int64_t edge_base, int64_t lo, int64_t hi,
                                                     int64_t total_nodes, int64_t n_local,
                                                     const uint32_t *__restrict__ bitmap,
                                                     const int32_t *__restrict__ word_slot,
                                                     int64_t *__restrict__ local_src) {
  const int64_t n = tab.offset[tab.num_types];
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int t = shard_type_of(tab, e);
    int64_t s = tab.src[t][e - tab.offset[t]];
    s = s < 0 ? 0 : (s < total_nodes ? s : total_nodes - 1);
    if (s < lo || s >= hi) {
      const int64_t w = s >> 5;
      local_src[edge_base + e] = n_local + word_slot[w] + __popc(bitmap[w] & ((1u << (s & 31)) - 1u));
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Unique (edge type, source) pairs of a plan.  A GGNN message  W_t x[src]  (gatedmessagepassing.py:52-58) depends on the
// edge only through (t, src): edges that share the pair share the message row.  The same bitmap machinery ranks the
// pairs -- key = t * num_src_rows + src, so the rows come out type-major (what the grouped per-edge GEMM wants) and
// ascending in src inside a type -- and gives every CSR slot the row of its pair.  Integer bookkeeping only; which edges
// share a row never changes a value (a message row is the same fmaf chain wherever it is computed).
//   k_uniq_mark     one pass over the plan's col array ((src << type_bits) | type): a bit per pair that occurs
//   k_uniq_pack / k_shard_compact<true>  rank of every set bit; the sorted keys leave as source node ids (the "adjacency
//                   list" of the de-duplicated GEMM)
//   k_uniq_remap    CSR slot -> message row; its last workgroup computes the rows per edge type (+ the total, for the one
//                   host read-back) and the grouped GEMM's launch table
// (round 4: five launches incl. the flag memset, where round 3 ran seven -- at this size every launch is ~4 us of ramp)
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int64_t uniq_key(int32_t c, int type_bits, int num_types, int64_t num_src_rows) {
  const uint32_t u = (uint32_t)c;
  int64_t t = u & ((1u << type_bits) - 1u), s = u >> type_bits;
  t = t < num_types ? t : num_types - 1;             // a plan never holds such entries; keep the key inside the bitmap
  s = s < num_src_rows ? s : num_src_rows - 1;
  return t * num_src_rows + s;
}

// A byte per pair, plain stores (every writer stores the same 1): bit-granular marks need device-scope atomics, which
// execute at the memory side -- 53 us for the 625 k edges of BASELINE config 3 against ~8 us for the byte stores.
__global__ __launch_bounds__(256) void k_uniq_mark(const int32_t *__restrict__ col, int64_t num_edges, int type_bits,
                                                   int num_types, int64_t num_src_rows, uint8_t *__restrict__ flags) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < num_edges; i += (int64_t)gridDim.x * blockDim.x)
    flags[uniq_key(col[i], type_bits, num_types, num_src_rows)] = 1;
}

// flags -> bitmap (thread t of a workgroup: the 32 flags of word t) + the workgroup's population count (k_shard_blocks
// of the shard index, fused); `flags` is padded to whole words and zero behind the last key
__global__ __launch_bounds__(256) void k_uniq_pack(const uint8_t *__restrict__ flags, int64_t words,
                                                   uint32_t *__restrict__ bitmap, int32_t *__restrict__ block_sum) {
  __shared__ int part[4];
  const int64_t w = (int64_t)blockIdx.x * kBlockWords + threadIdx.x;
  uint32_t bits = 0;
  if (w < words) {
    const uint4 a = reinterpret_cast<const uint4 *>(flags)[2 * w], b = reinterpret_cast<const uint4 *>(flags)[2 * w + 1];
    const uint32_t v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int k = 0; k < 8; ++k) {      // four flag bytes (0 / 1) -> four bits
      const uint32_t x = v[k];
      bits |= ((x & 1u) | ((x >> 7) & 2u) | ((x >> 14) & 4u) | ((x >> 21) & 8u)) << (4 * k);
    }
    bitmap[w] = bits;
  }
  int c = __popc(bits);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) block_sum[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// one wave: rows per edge type (+ total) and the grouped GEMM's launch table
__device__ __forceinline__ void uniq_counts_wave(const uint32_t *__restrict__ bitmap,
                                                 const int32_t *__restrict__ word_slot, int64_t words,
                                                 const int32_t *__restrict__ block_sum, int nblocks, int num_types,
                                                 int64_t num_src_rows, int64_t *__restrict__ counts,
                                                 const int64_t *unique_src, StreamEdgeTable *__restrict__ table,
                                                 int budget_cus) {
  const int lane = threadIdx.x & 63;
  int c = 0;
  for (int j = lane; j < nblocks; j += 64) c += block_sum[j];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  const int64_t total = c;
  if (lane == 0) counts[num_types] = total;
  int64_t first = 0, rows = 0;                     // lane t: the rows of edge type t (num_types <= 64 with a table)
  for (int t = lane; t < num_types; t += 64) {
    const int64_t a = slot_of(bitmap, word_slot, words, (int64_t)t * num_src_rows, total);
    const int64_t b = slot_of(bitmap, word_slot, words, (int64_t)(t + 1) * num_src_rows, total);
    counts[t] = b - a;
    first = a; rows = b - a;
  }
  if (!table) return;
  // The grouped per-edge GEMM's table (stream_gemm.h), on the device: the same apportioning of one workgroup per CU
  // to the edge types as stream_edge() computes on the host -- proportional start, then the spare workgroups go to /
  // the excess comes from the type whose load per workgroup moves the maximum least.
  const bool live = lane < num_types;
  const int units = live ? (int)((rows + 31) / 32) : 0;
  int unit_incl = units;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(unit_incl, o, 64);
    if (lane >= o) unit_incl += v;
  }
  const int total_units = __shfl(unit_incl, 63, 64);
  int budget = budget_cus;
  if (budget > total_units / 8 + 1) budget = total_units / 8 + 1;
  int w = units == 0 ? 0 : (int)((int64_t)units * budget / (total_units > 0 ? total_units : 1));
  if (units > 0 && w == 0) w = 1;
  auto wave_sum = [](int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
  };
  int sum = wave_sum(w);
  const int nonempty = wave_sum(units > 0 ? 1 : 0);
  if (budget < nonempty) budget = nonempty;
  for (int it = 0; it < 4 * 64 + budget_cus && sum != budget; ++it) {
    const bool give = sum < budget;
    // candidate value of this lane; the winner is the largest load per workgroup (give) / the smallest load after the
    // cut (take), the lowest edge type among equals
    double v = -1.0;
    if (units > 0 && (give || w > 1)) v = give ? (double)units / w : (double)units / (w - 1);
    double best = v;
    int who = v >= 0.0 ? lane : 64;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double ov = __shfl_xor(best, o, 64);
      const int ow = __shfl_xor(who, o, 64);
      const bool take_other = ow < 64 && (who == 64 || (give ? (ov > best || (ov == best && ow < who))
                                                              : (ov < best || (ov == best && ow < who))));
      if (take_other) { best = ov; who = ow; }
    }
    if (who == 64) break;
    if (lane == who) w += give ? 1 : -1;
    sum += give ? 1 : -1;
  }
  int wg_incl = w;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(wg_incl, o, 64);
    if (lane >= o) wg_incl += v;
  }
  if (live) {
    table->src[lane] = unique_src + first;
    table->dst[lane] = unique_src + first;
    table->w[lane] = nullptr;
    table->edge_off[lane] = first;
    table->unit_off[lane] = unit_incl - units;
    table->wg_off[lane] = wg_incl - w;
  }
  if (lane >= num_types) {   // every prefix entry past the last type holds the totals: the GEMM kernel fetches the arrays one
    table->unit_off[lane] = total_units;   // entry per lane and searches them with a ballot, without knowing num_types
    table->wg_off[lane] = sum;
  }
  if (lane == 0) {
    table->edge_off[num_types] = total;
    table->unit_off[kStreamMaxTypes] = total_units;
    table->wg_off[kStreamMaxTypes] = sum;
    table->num_types = num_types;
  }
}

// CSR slot -> message row for every edge; the LAST workgroup's first wave does not remap but runs the one-wave count /
// launch-table pass (round 3: a launch of its own, 8.7 us of dependent latency in front of the remap -- the two only
// share their inputs, so they run side by side now)
__global__ __launch_bounds__(256) void k_uniq_remap(const int32_t *__restrict__ col, int64_t num_edges, int type_bits,
                                                    int num_types, int64_t num_src_rows,
                                                    const uint32_t *__restrict__ bitmap,
                                                    const int32_t *__restrict__ word_slot, int64_t words,
                                                    int32_t *__restrict__ slot_row,
                                                    const int32_t *__restrict__ block_sum, int nblocks,
                                                    int64_t *__restrict__ counts, const int64_t *unique_src,
                                                    StreamEdgeTable *__restrict__ table, int budget_cus) {
  if (blockIdx.x == gridDim.x - 1) {
    if (threadIdx.x < 64)
      uniq_counts_wave(bitmap, word_slot, words, block_sum, nblocks, num_types, num_src_rows, counts, unique_src, table,
                       budget_cus);
    return;
  }
  const int64_t stride = (int64_t)(gridDim.x - 1) * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < num_edges; i += stride) {
    const int64_t key = uniq_key(col[i], type_bits, num_types, num_src_rows);
    const int64_t w = key >> 5;
    slot_row[i] = word_slot[w] + __popc(bitmap[w] & ((1u << (key & 31)) - 1u));
  }
}

inline size_t sh_align(size_t x) { return (x + 255) / 256 * 256; }

}  // namespace
}  // namespace ptgnn_amd

using namespace ptgnn_amd;

extern "C" size_t ptgnn_amd_shard_index_workspace_bytes(int64_t total_nodes) {
  if (total_nodes < 0) return 0;
  const size_t words = (size_t)((total_nodes + 31) / 32) + 1;
  const size_t blocks = (words + kBlockWords - 1) / kBlockWords;
  return sh_align(words * 4) * 2 + sh_align(blocks * 4) + 512;
}

extern "C" int ptgnn_amd_shard_index(const int64_t *const *src_per_type, const int64_t *const *dst_per_type,
                                     const int64_t *edges_per_type, int32_t num_types, int64_t lo, int64_t hi,
                                     const int64_t *bounds, int32_t world, int64_t total_nodes, int64_t *local_src,
                                     int64_t *local_dst, int64_t *need_ids, int64_t need_capacity, int64_t *stats,
                                     int32_t *bad_index_count, void *workspace, size_t workspace_bytes, void *stream_) {
  hipStream_t st = (hipStream_t)stream_;
  PTGNN_REQUIRE(num_types > 0 && world > 0 && lo >= 0 && hi >= lo && total_nodes >= hi, PTGNN_AMD_EINVAL,
                "shard_index: bad sizes");
  PTGNN_REQUIRE(src_per_type && dst_per_type && edges_per_type && bounds && stats, PTGNN_AMD_EINVAL,
                "shard_index: null pointer");
  int64_t num_edges = 0;
  for (int t = 0; t < num_types; ++t) {
    PTGNN_REQUIRE(edges_per_type[t] >= 0 && (edges_per_type[t] == 0 || (src_per_type[t] && dst_per_type[t])),
                  PTGNN_AMD_EINVAL, "shard_index: bad adjacency list %d", t);
    num_edges += edges_per_type[t];
  }
  PTGNN_REQUIRE(num_edges == 0 || (local_src && local_dst), PTGNN_AMD_EINVAL, "shard_index: null output");
  const int64_t remote_nodes = total_nodes - (hi - lo);
  PTGNN_REQUIRE(need_capacity >= (num_edges < remote_nodes ? num_edges : remote_nodes) && (need_ids || need_capacity == 0),
                PTGNN_AMD_EINVAL, "shard_index: need_ids holds %lld ids, up to %lld may be written",
                (long long)need_capacity, (long long)(num_edges < remote_nodes ? num_edges : remote_nodes));
  PTGNN_REQUIRE(workspace_bytes >= ptgnn_amd_shard_index_workspace_bytes(total_nodes) && workspace, PTGNN_AMD_EWORKSPACE,
                "shard_index: workspace too small");
  const int64_t words = (total_nodes + 31) / 32 + 1;
  const int nblocks = (int)((words + kBlockWords - 1) / kBlockWords);
  char *ws = (char *)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  uint32_t *bitmap = (uint32_t *)ws;
  int32_t *word_slot = (int32_t *)(ws + sh_align((size_t)words * 4));
  int32_t *block_sum = (int32_t *)(ws + 2 * sh_align((size_t)words * 4));
  PTGNN_HIP(hipMemsetAsync(bitmap, 0, (size_t)words * 4, st));
  PTGNN_HIP(hipMemsetAsync(stats, 0, sizeof(int64_t) * (size_t)(world + 2 + num_types), st));
  auto for_each_table = [&](auto &&launch) -> int {
    int64_t base = 0;
    for (int t0 = 0; t0 < num_types; t0 += kShardTypes) {
      ShardTable tab{};
      tab.num_types = num_types - t0 < kShardTypes ? num_types - t0 : kShardTypes;
      tab.type_base = t0;
      for (int t = 0; t < tab.num_types; ++t) {
        tab.src[t] = src_per_type[t0 + t];
        tab.dst[t] = dst_per_type[t0 + t];
        tab.offset[t + 1] = tab.offset[t] + edges_per_type[t0 + t];
      }
      const int64_t n = tab.offset[tab.num_types];
      if (n > 0) {
        const int64_t blocks = (n + 255) / 256;
        launch(tab, base, (unsigned)(blocks < 8192 ? blocks : 8192));
        PTGNN_LAUNCH_CHECK();
      }
      base += n;
    }
    return PTGNN_AMD_OK;
  };
  int rc = for_each_table([&](const ShardTable &tab, int64_t base, unsigned grid) {
    k_shard_mark<<<grid, 256, 0, st>>>(tab, base, lo, hi, total_nodes, bitmap, local_src, local_dst,
                                      (unsigned long long *)stats, world, bad_index_count);
  });
  if (rc != PTGNN_AMD_OK) return rc;
  k_shard_blocks<<<(unsigned)nblocks, 256, 0, st>>>(bitmap, words, block_sum);
  PTGNN_LAUNCH_CHECK();
  k_shard_compact<false><<<(unsigned)nblocks, 256, 0, st>>>(bitmap, words, block_sum, word_slot, need_ids, 1, 0);
  PTGNN_LAUNCH_CHECK();
  k_shard_counts<<<1, 64, 0, st>>>(bitmap, word_slot, words, block_sum, nblocks, bounds, world,
                                   (unsigned long long *)stats);
  PTGNN_LAUNCH_CHECK();
  rc = for_each_table([&](const ShardTable &tab, int64_t base, unsigned grid) {
    k_shard_remap<<<grid, 256, 0, st>>>(tab, base, lo, hi, total_nodes, hi - lo, bitmap, word_slot, local_src);
  });
  return rc;
}

extern "C" size_t ptgnn_amd_edge_table_bytes(void) { return sizeof(StreamEdgeTable); }

extern "C" size_t ptgnn_amd_unique_sources_workspace_bytes(int64_t num_src_rows, int32_t num_types) {
  if (num_src_rows < 0 || num_types <= 0) return 0;
  const int64_t keys = num_src_rows * num_types;
  const size_t words = (size_t)((keys + 31) / 32) + 1;
  return ptgnn_amd_shard_index_workspace_bytes(keys) + sh_align(words * 32);      // + a flag byte per (type, source)
}

extern "C" int ptgnn_amd_unique_sources(const int32_t *col, int64_t num_edges, int32_t type_bits, int32_t num_types,
                                        int64_t num_src_rows, int32_t *slot_row, int64_t *unique_src,
                                        int64_t capacity, int64_t *counts, void *edge_table, void *workspace,
                                        size_t workspace_bytes, void *stream_) {
  hipStream_t st = (hipStream_t)stream_;
  PTGNN_REQUIRE(num_edges >= 0 && num_types > 0 && num_src_rows > 0 && type_bits >= 0 && type_bits < 31 &&
                    num_types <= (1 << type_bits),
                PTGNN_AMD_EINVAL, "unique_sources: bad sizes");
  const int64_t keys = num_src_rows * num_types;
  PTGNN_REQUIRE(keys < ((int64_t)1 << 31), PTGNN_AMD_EUNSUPPORTED, "unique_sources: %lld (type, source) pairs",
                (long long)keys);
  PTGNN_REQUIRE(counts && (num_edges == 0 || (col && slot_row)), PTGNN_AMD_EINVAL, "unique_sources: null pointer");
  const int64_t most = num_edges < keys ? num_edges : keys;
  PTGNN_REQUIRE(capacity >= most && (unique_src || most == 0), PTGNN_AMD_EINVAL,
                "unique_sources: unique_src holds %lld ids, up to %lld may be written", (long long)capacity,
                (long long)most);
  PTGNN_REQUIRE(workspace_bytes >= ptgnn_amd_unique_sources_workspace_bytes(num_src_rows, num_types) && workspace,
                PTGNN_AMD_EWORKSPACE, "unique_sources: workspace too small");
  const int64_t words = (keys + 31) / 32 + 1;
  const int nblocks = (int)((words + kBlockWords - 1) / kBlockWords);
  char *ws = (char *)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  uint32_t *bitmap = (uint32_t *)ws;
  int32_t *word_slot = (int32_t *)(ws + sh_align((size_t)words * 4));
  int32_t *block_sum = (int32_t *)(ws + 2 * sh_align((size_t)words * 4));
  uint8_t *flags = (uint8_t *)(ws + 2 * sh_align((size_t)words * 4) + sh_align((size_t)nblocks * 4) + 256);
  PTGNN_HIP(hipMemsetAsync(flags, 0, (size_t)words * 32, st));
  const int64_t eblocks = (num_edges + 255) / 256;
  const unsigned egrid = (unsigned)(eblocks < 8192 ? (eblocks > 0 ? eblocks : 1) : 8192);
  if (num_edges > 0) {
    k_uniq_mark<<<egrid, 256, 0, st>>>(col, num_edges, type_bits, num_types, num_src_rows, flags);
    PTGNN_LAUNCH_CHECK();
  }
  k_uniq_pack<<<(unsigned)nblocks, 256, 0, st>>>(flags, words, bitmap, block_sum);
  PTGNN_LAUNCH_CHECK();
  k_shard_compact<true><<<(unsigned)nblocks, 256, 0, st>>>(bitmap, words, block_sum, word_slot, unique_src, num_src_rows,
                                                           capacity);
  PTGNN_LAUNCH_CHECK();
  PTGNN_REQUIRE(!edge_table || num_types <= kStreamMaxTypes, PTGNN_AMD_EUNSUPPORTED,
                "unique_sources: an edge table holds at most %d edge types", kStreamMaxTypes);
  // remap over the edges + (last workgroup) the counts / launch table: one launch; with no edges only the counts run
  k_uniq_remap<<<egrid + 1, 256, 0, st>>>(col, num_edges, type_bits, num_types, num_src_rows, bitmap, word_slot, words,
                                          slot_row, block_sum, nblocks, counts, unique_src,
                                          (StreamEdgeTable *)edge_table, edge_table_budget());
  PTGNN_LAUNCH_CHECK();
  return PTGNN_AMD_OK;
}
