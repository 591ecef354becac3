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

namespace ptgnn_amd {
namespace {

constexpr int kShardTypes = 64;
constexpr int kBlockWords = 1024;

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
                                                    unsigned long long *__restrict__ stats, int world) {
  __shared__ int own_cnt[kShardTypes];
  __shared__ int remote_cnt;
  if (threadIdx.x < kShardTypes) own_cnt[threadIdx.x] = 0;
  if (threadIdx.x == 0) remote_cnt = 0;
  __syncthreads();
  const int64_t n = tab.offset[tab.num_types];
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int t = shard_type_of(tab, e);
    const int64_t i = e - tab.offset[t];
    int64_t s = tab.src[t][i];
    const int64_t d = tab.dst[t][i];
    s = s < 0 ? 0 : (s < total_nodes ? s : total_nodes - 1);     // a bad id must not leave the bitmap (the plan
    local_dst[edge_base + e] = d - lo;                           // build's range guard reports it)
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

// one workgroup per 1024-word block; thread t owns the 4 consecutive words 4 t .. 4 t + 3
__global__ __launch_bounds__(256) void k_shard_compact(const uint32_t *__restrict__ bitmap, int64_t words,
                                                       const int32_t *__restrict__ block_sum,
                                                       int32_t *__restrict__ word_slot, int64_t *__restrict__ need_ids) {
  __shared__ int wsum[4];
  __shared__ int base_s;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  {   // halo slots in front of this block = sum of the earlier blocks' counts
    int c = 0;
    for (int j = threadIdx.x; j < (int)blockIdx.x; j += 256) c += block_sum[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if (lane == 0) wsum[wave] = c;
    __syncthreads();
    if (threadIdx.x == 0) base_s = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
  const int64_t w0 = (int64_t)blockIdx.x * kBlockWords + 4 * threadIdx.x;
  uint32_t v[4];
  int mine = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    v[k] = w0 + k < words ? bitmap[w0 + k] : 0u;
    mine += __popc(v[k]);
  }
  int inc = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int run = base_s + inc - mine;
  for (int w = 0; w < wave; ++w) run += wsum[w];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (w0 + k < words) {
      word_slot[w0 + k] = run;
      uint32_t bits = v[k];
      while (bits) {
        const int b = __ffs((int)bits) - 1;
        bits &= bits - 1;
        need_ids[run++] = (w0 + k) * 32 + b;
      }
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
                                     void *workspace, size_t workspace_bytes, void *stream_) {
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
                                      (unsigned long long *)stats, world);
  });
  if (rc != PTGNN_AMD_OK) return rc;
  k_shard_blocks<<<(unsigned)nblocks, 256, 0, st>>>(bitmap, words, block_sum);
  PTGNN_LAUNCH_CHECK();
  k_shard_compact<<<(unsigned)nblocks, 256, 0, st>>>(bitmap, words, block_sum, word_slot, need_ids);
  PTGNN_LAUNCH_CHECK();
  k_shard_counts<<<1, 64, 0, st>>>(bitmap, word_slot, words, block_sum, nblocks, bounds, world,
                                   (unsigned long long *)stats);
  PTGNN_LAUNCH_CHECK();
  rc = for_each_table([&](const ShardTable &tab, int64_t base, unsigned grid) {
    k_shard_remap<<<grid, 256, 0, st>>>(tab, base, lo, hi, total_nodes, hi - lo, bitmap, word_slot, local_src);
  });
  return rc;
}
