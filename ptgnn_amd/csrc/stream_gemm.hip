// Streaming ("skinny") GEMM core of the message-passing layers: rows >> K, n_out.
//
//     C[rows, n_out] = A[rows, K] . W[n_out, K]^T          K in {64 .. ~400}, n_out small
//
// Every dense block of the hot path has this shape (per-edge message Linear: 625 k gathered rows x
// K = 128; GRU gates: 116 k rows x K = 256; MLP update: 200 k rows x K = 128).  The round-1 kernels
// tiled it like a square GEMM -- 128 x 128 output tile per workgroup, both operands staged through LDS
// per 32-wide K chunk, two workgroup barriers per chunk -- and landed at 0.64 of the fp32-MFMA peak
// with "time = t(MFMA) + t(everything else)".  What round 2 measured (profiles/r02_notes.md): the
// f32-input MFMA sustains 155 TFLOP/s on its own, but it runs at the fp32 VECTOR rate and VALU work
// beyond about one instruction per MFMA is not hidden under it -- it adds to it; LDS reads are free.  So
// the rule for exact fp32 on CDNA4 is: as few non-MFMA instructions per MFMA as possible.
//
// Structure (MI355X-first: 160 KB LDS per CU, 512 VGPRs per SIMD):
//   * the WEIGHT SLAB is stationary: a persistent 8-wave workgroup (one per CU) copies its [BN, K] slice
//     of W into LDS once per run (BN = 128 columns, or the 3 x 32 gate rows of one GRU feature tile) --
//     68 KB at K = 128, 100-150 KB for the GRU at K = 256-384;
//   * A never touches LDS: a wave owns 32 rows ("unit") and loads its MFMA A fragments straight from
//     global memory.  The K index of an MFMA step is a free permutation as long as A and B agree, so
//     lane (row li, half hi) takes the 16-byte pieces k = 32c + 8g + 4hi + {0..3} (`kcol()` in
//     dense_common.h): every global load is a dwordx4, every LDS read of the matching B fragment a
//     conflict-free ds_read_b128, and there is no staging VALU, ds_write or workgroup barrier in steady
//     state;
//   * the four pieces of a chunk are the four sectors of one cache line per row and are refilled as a burst,
//     one chunk pair ahead and ACROSS unit boundaries (the next unit's first chunks load under the current
//     unit's last MFMAs and its epilogue);
//   * epilogues go through a wave-private 8 x 32 transposing slab: dwordx4 stores of 8 rows x 128 B instead
//     of dword stores in the C-fragment layout; the GRU's gate math runs on v_exp_f32 / v_rcp_f32;
//   * work = balanced runs of units in group-major order (group = column slab / GRU feature tile / edge type;
//     the edge kernel apportions its workgroups to the types), units claimed dynamically from an LDS
//     counter inside a run; the dense kernels map runs so that the column slabs of one row range share an
//     XCD (L2 reuse of A).
// The accumulation order over K inside a row is ONE fixed permutation shared with the tile kernels, so a
// result does not depend on the kernel, the tile position, the shard or the layer form that produced it.
//
// Kernel families (ptgnn_amd_set_gemm_mode / PTGNN_AMD_GEMM), both exact fp32 -- v_mfma_f32_32x32x2_f32, bit for bit an
// fmaf chain in ONE K order:
//   1  the streaming kernels of this file (default);
//   0  the round-1 tile kernels (dense_f32.hip / edge_gemm.hip), also the fallback for shapes this file does not take
//      (K % 64 != 0, unaligned rows, slabs that do not fit LDS).
// (Rounds 2-4 carried a third, opt-in mode: f32 emulated by an exact 3 x bf16 operand split on the bf16 MFMA.  It ran
//  the headline step 1.17x faster at the same error against float64, never reached a useful fraction of the bf16 peak
//  -- 0.35 of a six-product ceiling: the unit is bound by its ~40 wave-wide memory instructions, not by the split VALU --
//  and was removed in round 5 rather than carried as a second arithmetic nobody benchmarks against;
//  profiles/r02_notes.md, r04_notes.md and DESIGN.md 9 hold its numbers.)
#include <stdlib.h>

#include <mutex>
#include <unordered_map>

#include "dense_common.h"
#include <atomic>

#include "stream_gemm.h"

namespace ptgnn_amd {
namespace {

using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

constexpr int kLdsBudget = 160 * 1024;
constexpr size_t kEpiBytes = 16 + 8 * 8 * 36 * sizeof(float);   // unit counter + 8 wave-private transposing slabs

__device__ __forceinline__ f32x16 zero16() {
  f32x16 v;
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = 0.f;
  return v;
}

// per-lane A row pointers (+ this lane's piece offset folded in) of the current and the next unit;
// phase 1 = the second K range of two-source kernels (GRU: previous state, MLP edge form: target state)
struct ARows {
  const float *c0, *c1, *n0, *n1;
};

// float4 piece g of chunk cc of the lane's row; cc >= nch addresses the NEXT unit (cross-unit prefetch).
// All conditions are wave-uniform: selects, no branches (a branch around a load costs a vmcnt(0) drain).
__device__ __forceinline__ float4 load_piece(const ARows &r, int cc, int ch0, int nch, int g) {
  const bool nx = cc >= nch;
  const int c = nx ? cc - nch : cc;
  const float *b0 = nx ? r.n0 : r.c0;
  const float *b1 = nx ? r.n1 : r.c1;
  const float *p = c < ch0 ? b0 + c * 32 : b1 + (c - ch0) * 32;
  // fp32 MFMA (K = 2 per step): pieces 8 g + 4 hi
  return *reinterpret_cast<const float4 *>(p + g * 8);
}

__device__ __forceinline__ int lane_piece_offset(int hi) { return hi * 4; }

// ---- one K chunk (32 columns) of one unit ---------------------------------------------------------
// NBLK column blocks of the slab per step; GRU maps block 2 to accumulator 3 in phase 1 (h_n).
template <int NBLK, int NACC, int PH, bool GRU>
__device__ __forceinline__ void chunk_f32(f32x16 (&acc)[NACC], float4 (&buf)[4], float4 (&bcur)[NBLK],
                                          const float *bl, int cbs, int kofs, int kofs_next, const ARows &rows,
                                          int cnext, int ch0, int nch) {
  // bcur holds the B fragments of piece 0 on entry and those of the NEXT chunk's piece 0 on exit: the LDS reads
  // run one piece ahead of the MFMAs.  The four pieces of a chunk are the four 32-byte sectors of ONE cache line
  // per row, so they are refilled as a burst once the chunk is consumed (one L2 fetch per line; refilled one at a
  // time, 16 MFMAs apart, each piece re-fetched the line after the other waves had flushed it from the 32 KB L1:
  // measured 11 % of the kernel).  sched_barriers pin the order: hipcc otherwise sinks the LDS reads below the MFMAs,
  // and counts vmcnt per loop only when every path into the loop issued the loads in the same order.
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float4 bn[NBLK];
    const int ko = g < 3 ? kofs + (g + 1) * 8 : kofs_next;
#pragma unroll
    for (int n = 0; n < NBLK; ++n) bn[n] = *reinterpret_cast<const float4 *>(bl + n * cbs + ko);
    __builtin_amdgcn_sched_barrier(0);
    const float4 a = buf[g];
#define PTGNN_STEP(C)                                                                     \
    _Pragma("unroll") for (int n = 0; n < NBLK; ++n) {                                    \
      const int t = (GRU && PH == 1 && n == 2) ? 3 : n;                                   \
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.C, bcur[n].C, acc[t], 0, 0, 0);     \
    }
    PTGNN_STEP(x) PTGNN_STEP(y) PTGNN_STEP(z) PTGNN_STEP(w)
#undef PTGNN_STEP
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int n = 0; n < NBLK; ++n) bcur[n] = bn[n];
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    buf[g] = load_piece(rows, cnext, ch0, nch, g);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// The slab in LDS: [rows][K + 4] floats.
struct Slab {
  int ld;       // row stride in floats
  __device__ __forceinline__ Slab(int K, int /*rows*/) { ld = K + 4; }
  static size_t bytes(int K, int rows) { return (size_t)rows * (K + 4) * 4; }

  // store 4 consecutive K values of slab row r
  __device__ __forceinline__ void put4(float *smem, int r, int k, float4 v) const {
    *reinterpret_cast<float4 *>(smem + r * ld + k) = v;
  }
};

// Slab fill: `src(r, q)` = global address of the float4 `q` of slab row `r` (always loadable), `live(r)` = whether the
// row is real (else zeros).  The loads of a batch of 8 are ALL issued before the first LDS store: written as one load ->
// store per iteration the compiler waits for every float4 at once, and the fill is 8-18 dependent L2 round trips --
// 4-9 us of every launch, most of the kernels' size-independent cost (profiles/r04_notes.md 9).
template <int NT, typename SrcFn, typename LiveFn>
__device__ __forceinline__ void fill_slab(float *smem, const Slab &sl, int rows, int kq, SrcFn src, LiveFn live) {
  constexpr int UB = 8;
  const int total = rows * kq;
  for (int i0 = threadIdx.x; i0 < total; i0 += NT * UB) {
    float4 v[UB];
    int rr[UB], qq[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int i = i0 + u * NT;
      const int ic = i < total ? i : total - 1;
      rr[u] = ic / kq;
      qq[u] = ic - rr[u] * kq;
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      v[u] = *reinterpret_cast<const float4 *>(src(rr[u], qq[u]));
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      if (i0 + u * NT < total) sl.put4(smem, rr[u], qq[u] * 4, live(rr[u]) ? v[u] : make_float4(0.f, 0.f, 0.f, 0.f));
    }
  }
}

// K loop of one unit: chunks [0, ch0) are phase 0, [ch0, nch) phase 1; a0/a1 hold chunks 0/1 on entry
// and the next unit's chunks 0/1 on exit.  ch0 and nch are even.
template <int NBLK, int NACC, bool GRU>
__device__ __forceinline__ void unit_kloop(f32x16 (&acc)[NACC], float4 (&a0)[4], float4 (&a1)[4],
                                           const float *smem, const Slab &sl, int li, int hi,
                                           const ARows &rows, int ch0, int nch) {
  const float *bl = smem + li * sl.ld + hi * 4;
  const int cbs = 32 * sl.ld;
  const int e0 = GRU ? ch0 : nch;
  float4 bcur[NBLK];
#pragma unroll
  for (int n = 0; n < NBLK; ++n) bcur[n] = *reinterpret_cast<const float4 *>(bl + n * cbs);
  for (int c = 0; c < e0; c += 2) {
    const int kn = c + 2 < nch ? c * 32 + 64 : 0;   // the chunk after next (0: first chunk of the next unit)
    chunk_f32<NBLK, NACC, 0, GRU>(acc, a0, bcur, bl, cbs, c * 32, c * 32 + 32, rows, c + 2, ch0, nch);
    chunk_f32<NBLK, NACC, 0, GRU>(acc, a1, bcur, bl, cbs, c * 32 + 32, kn, rows, c + 3, ch0, nch);
  }
  if constexpr (GRU) {
    for (int c = ch0; c < nch; c += 2) {
      const int kn = c + 2 < nch ? c * 32 + 64 : 0;
      chunk_f32<NBLK, NACC, 1, GRU>(acc, a0, bcur, bl, cbs, c * 32, c * 32 + 32, rows, c + 2, ch0, nch);
      chunk_f32<NBLK, NACC, 1, GRU>(acc, a1, bcur, bl, cbs, c * 32 + 32, kn, rows, c + 3, ch0, nch);
    }
  }
}

// hipcc counts vmcnt per loop only when every path into the loop header issued the pending loads in the
// SAME order as the loop body does (the header wait is the most conservative of its predecessors).  So:
// the first unit's loads are pinned in steady-state order, and each unit starts from a drained queue
// (`unit_fence`): the epilogue's stores share the counter with the prefetched loads, and a mixed
// load/store queue makes the compiler fall back to vmcnt(0) on EVERY chunk instead of once per unit.
__device__ __forceinline__ void prologue_loads(float4 (&a0)[4], float4 (&a1)[4], const ARows &rows, int ch0,
                                               int nch) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    a0[g] = load_piece(rows, 0, ch0, nch, g);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    a1[g] = load_piece(rows, 1, ch0, nch, g);
    __builtin_amdgcn_sched_barrier(0);
  }
}

__device__ __forceinline__ void unit_fence() {
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt / lgkmcnt untouched
  __builtin_amdgcn_sched_barrier(0);
}

// Dynamic unit claiming: the waves of a workgroup pull units of the current run / segment from one LDS
// counter (one returning LDS atomic per unit), so they finish within one unit of each other whatever
// their relative speed.  `claim` returns an index >= `count` when the segment is exhausted.
__device__ __forceinline__ int claim_unit(int *counter) {
  int v = 0;
  if ((threadIdx.x & 63) == 0) v = atomicAdd(counter, 1);
  return __builtin_amdgcn_readfirstlane(v);
}

// ---------------------------------------------------------------------------------------------------
// linear: y = act(x W^T + b)
// ---------------------------------------------------------------------------------------------------
// Epilogue stores.  A C fragment holds, per lane, 4 consecutive ROWS of one column (x 4 row groups): stored as
// is, a unit costs 64 dword stores of 2 x 128 B each -- store-issue bound, measured 16-21 % of the kernels.  A
// wave-private 8 x 32 transposing slab in LDS (1.1 KB per wave; DS traffic is free next to fp32 MFMAs) turns a
// row group into ONE dwordx4 store per lane: 8 rows x 128 contiguous bytes.
constexpr int kTqLd = 36;                    // floats per slab row (32 + 4: 16-byte aligned rows)
constexpr int kTqFloats = 8 * kTqLd;         // per wave

// v[i] = value of row (4 hi + i), column li of the row group; returns this lane's float4 = row lane/8, cols 4 (lane%8)
__device__ __forceinline__ float4 tq_transpose(float *tq, int lane, int li, int hi, float v0, float v1, float v2,
                                               float v3) {
  float *w = tq + (4 * hi) * kTqLd + li;
  w[0] = v0; w[kTqLd] = v1; w[2 * kTqLd] = v2; w[3 * kTqLd] = v3;
  __builtin_amdgcn_wave_barrier();            // DS ops of one wave execute in order; this pins the compiler
  const float4 o = *reinterpret_cast<const float4 *>(tq + (lane >> 3) * kTqLd + (lane & 7) * 4);
  __builtin_amdgcn_wave_barrier();
  return o;
}

// the reverse: this lane's float4 (row lane/8, cols 4 (lane%8)) -> values of rows 4 hi + {0..3}, column li
__device__ __forceinline__ void tq_untranspose(float *tq, int lane, int li, int hi, float4 in, float (&v)[4]) {
  *reinterpret_cast<float4 *>(tq + (lane >> 3) * kTqLd + (lane & 7) * 4) = in;
  __builtin_amdgcn_wave_barrier();
  const float *r = tq + (4 * hi) * kTqLd + li;
  v[0] = r[0]; v[1] = r[kTqLd]; v[2] = r[2 * kTqLd]; v[3] = r[3 * kTqLd];
  __builtin_amdgcn_wave_barrier();
}

// Row stores of the epilogues go out as BUFFER stores (descriptor in SGPRs + a 32-bit byte offset per lane), not as
// global_store_dwordx4 with a 64-bit address per lane.  Measured under an fp32-MFMA stream (scripts/experiments/
// vmem_issue_probe.py, profiles/r04_notes.md 7): a wave-wide `global_store_dwordx4` to memory that is not cache-resident
// costs the issuing wave ~190-210 cycles, the same store as `buffer_store_dwordx4` ~15-30 -- and a unit issues sixteen.
// The descriptor's base is the unit's first row (wave-uniform) and its size ends with the unit's last VALID row, so the
// hardware drops the lanes of rows past a type's / the matrix's end: no per-lane condition, no branch around the store.
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rows_descriptor(float *base /* wave-uniform */, int64_t ld, int64_t rows_left,
                                                                  int width /* columns the unit writes from `base` */) {
  const uint64_t b = reinterpret_cast<uint64_t>(base);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
  const int64_t rows = rows_left < 32 ? rows_left : 32;
  const int64_t bytes = rows > 0 ? ((rows - 1) * ld + width) * 4 : 0;       // ends behind the last valid row's columns
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float *>(((uint64_t)hi << 32) | lo), 0,
                                           __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
}

// (the whole offset travels in the VGPR operand: the hardware range-checks voffset + the immediate against the
//  descriptor's size and adds the scalar offset AFTER the check, so a row term passed as soffset would escape it)
__device__ __forceinline__ void store_row4(__amdgpu_buffer_rsrc_t rsrc, int lane_bytes, int uniform_bytes, float4 o) {
  const u32x4 v = {__float_as_uint(o.x), __float_as_uint(o.y), __float_as_uint(o.z), __float_as_uint(o.w)};
  __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, lane_bytes + uniform_bytes, 0, 0);
}

// one 32 x 32 C block -> y (bias + activation on the way), 4 dwordx4 stores per lane
// BIAS = false: no bias add at all (the edge GEMMs: `c + 0.f` is not foldable under IEEE rules -- it turns -0 into +0 --
// so the literal zero they passed cost 64 v_add_f32 per unit, 1.5 % of the unit's MFMA time)
// `addblk` (nullable, wave-uniform): the same block of a matrix that is ADDED to the result after the activation
// (y = act(x W^T + b) + addend: the GRU backward's `d_h + d_gh W_hh`, dense.py) -- one dwordx4 load per store, in
// the stores' own row / column layout.
template <int ACT, bool BIAS = true>
__device__ __forceinline__ void store_block_tq(const f32x16 &c, float *tq, float *yblk /* row0, col0 of the block */,
                                               int64_t ld_y, float bv, int64_t rows_left, int lane, int li, int hi,
                                               const float *addblk = nullptr, int64_t ld_add = 0) {
  auto val = [&](int r) { return BIAS ? act_apply<ACT>(c[r] + bv) : act_apply<ACT>(c[r]); };
  float4 addv[4];
  if (addblk) {    // uniform branch; rows past the end re-read the last valid row (their stores are dropped)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      int64_t r = 8 * q + (lane >> 3);
      r = r < rows_left ? r : rows_left - 1;
      addv[q] = *reinterpret_cast<const float4 *>(addblk + r * ld_add + (lane & 7) * 4);
    }
  }
#ifdef PTGNN_GLOBAL_STORES   // A/B (scripts/build_variant.sh): the round-1..3 form, global_store_dwordx4 per lane
  float *yp = yblk + (int64_t)(lane >> 3) * ld_y + (lane & 7) * 4;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float4 o = tq_transpose(tq, lane, li, hi, val(4 * q), val(4 * q + 1), val(4 * q + 2), val(4 * q + 3));
    if (addblk) { o.x += addv[q].x; o.y += addv[q].y; o.z += addv[q].z; o.w += addv[q].w; }
    if (8 * q + (lane >> 3) < rows_left) *reinterpret_cast<float4 *>(yp + (int64_t)(8 * q) * ld_y) = o;
  }
#else
  const __amdgpu_buffer_rsrc_t rsrc = rows_descriptor(yblk, ld_y, rows_left, 32);
  const int ldb = (int)ld_y * 4;
  const int lane_bytes = (lane >> 3) * ldb + (lane & 7) * 16;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float4 o = tq_transpose(tq, lane, li, hi, val(4 * q), val(4 * q + 1), val(4 * q + 2), val(4 * q + 3));
    if (addblk) { o.x += addv[q].x; o.y += addv[q].y; o.z += addv[q].z; o.w += addv[q].w; }
    store_row4(rsrc, lane_bytes, 8 * q * ldb, o);
  }
#endif
}

struct LinearArgs {
  const float *x; int64_t rows; int K; int64_t ld_x;
  const float *w; int n_out; const float *bias; int act;
  float *y; int64_t ld_y;
  const float *addend; int64_t ld_add;   // nullable: y = act(x W^T + b) + addend (vec_store launches only)
  int nrb, ncs, rps, run_len;
  int lds_floats;            // slab size in floats (the unit counter sits behind it)
  int vec_store;             // y rows are 16-byte aligned: float4 stores through the transposing slab
};

template <int ACT>
__device__ __forceinline__ void store_block(const f32x16 &c, float *yp, int64_t ld_y, float bv, int64_t row0,
                                            int64_t rows, int hi) {
  // C fragment: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 hi
  if (row0 + 32 <= rows) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
      yp[(int64_t)((r & 3) + 8 * (r >> 2)) * ld_y] = act_apply<ACT>(c[r] + bv);
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rr = (r & 3) + 8 * (r >> 2);
      if (row0 + rr + 4 * hi < rows) yp[(int64_t)rr * ld_y] = act_apply<ACT>(c[r] + bv);
    }
  }
}

// bias + activation + store of one unit's NB column blocks; leaves the accumulators zeroed
template <int NB>
__device__ __forceinline__ void linear_epilogue(const LinearArgs &p, f32x16 (&acc)[NB], float *tq, int64_t row0,
                                                int col_base, int lane, int li, int hi) {
#pragma unroll
  for (int n = 0; n < NB; ++n) {
    const int col = col_base + n * 32 + li;
    if (col_base + n * 32 < p.n_out) {   // n_out % 32 == 0: whole column blocks
      const float bv = p.bias ? p.bias[col] : 0.f;
      if (p.vec_store) {
        float *yb = p.y + row0 * p.ld_y + col_base + n * 32;
        const int64_t left = p.rows - row0;
        // the bias goes into the accumulators under a uniform branch, the store adds nothing: a bias-free Linear (the
        // stacked edge pre-transform, the GRU's gradient GEMMs) must not pay 16 `c + 0.f` per block (not foldable: -0 + 0)
        if (p.bias) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[n][r] += bv;
        }
        const float *ab = p.addend ? p.addend + row0 * p.ld_add + col_base + n * 32 : nullptr;
        if (p.act == PTGNN_AMD_ACT_TANH) store_block_tq<PTGNN_AMD_ACT_TANH, false>(acc[n], tq, yb, p.ld_y, 0.f, left, lane, li, hi, ab, p.ld_add);
        else if (p.act == PTGNN_AMD_ACT_RELU) store_block_tq<PTGNN_AMD_ACT_RELU, false>(acc[n], tq, yb, p.ld_y, 0.f, left, lane, li, hi, ab, p.ld_add);
        else store_block_tq<PTGNN_AMD_ACT_NONE, false>(acc[n], tq, yb, p.ld_y, 0.f, left, lane, li, hi, ab, p.ld_add);
      } else {
        float *yp = p.y + (row0 + 4 * hi) * p.ld_y + col;
        if (p.act == PTGNN_AMD_ACT_TANH) store_block<PTGNN_AMD_ACT_TANH>(acc[n], yp, p.ld_y, bv, row0, p.rows, hi);
        else if (p.act == PTGNN_AMD_ACT_RELU) store_block<PTGNN_AMD_ACT_RELU>(acc[n], yp, p.ld_y, bv, row0, p.rows, hi);
        else store_block<PTGNN_AMD_ACT_NONE>(acc[n], yp, p.ld_y, bv, row0, p.rows, hi);
      }
    }
    acc[n] = zero16();
  }
}

template <int NB>
__global__ __launch_bounds__(512, 2) void k_stream_linear(LinearArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NT = 512, BN = 32 * NB;
  const int b = blockIdx.x, G = gridDim.x;
  int slab, part;
  if (G % (kNumXcd * p.ncs) == 0) {   // the column slabs of one row range share an XCD (L2 reuse of A)
    const int xcd = b % kNumXcd, j = b / kNumXcd;
    slab = j % p.ncs;
    part = xcd + kNumXcd * (j / p.ncs);
  } else {
    slab = b / p.rps;
    part = b % p.rps;
  }
  const int rb0 = part * p.run_len;
  const int rb1 = rb0 + p.run_len < p.nrb ? rb0 + p.run_len : p.nrb;
  if (rb0 >= rb1) return;
  const int count = rb1 - rb0;

  const Slab sl(p.K, BN);
  int *counter = reinterpret_cast<int *>(smem + p.lds_floats);
  float *const tq = smem + p.lds_floats + 4 + (threadIdx.x >> 6) * kTqFloats;   // this wave's transposing slab
  if (threadIdx.x == 0) *counter = 0;
  const int col_base = slab * BN;
  fill_slab<NT>(smem, sl, BN, p.K >> 2,
                       [&](int r, int q) {
                         const int wr = col_base + r < p.n_out ? col_base + r : p.n_out - 1;
                         return p.w + (int64_t)wr * p.K + q * 4;
                       },
                       [&](int r) { return col_base + r < p.n_out; });
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int li = lane & 31, hi = lane >> 5;
  const int nch = p.K >> 5;
  const int lofs = lane_piece_offset(hi);
  auto rowp = [&](int u) {   // u: unit index inside the run (clamped to the run)
    int64_t row = (int64_t)(rb0 + (u < count ? u : count - 1)) * 32 + li;
    row = row < p.rows ? row : p.rows - 1;
    return p.x + row * p.ld_x + lofs;
  };
  int cur = claim_unit(counter);
  if (cur >= count) return;
  int nxt = claim_unit(counter);
  ARows rows;
  rows.c0 = rows.c1 = rowp(cur);
  rows.n0 = rows.n1 = rowp(nxt);
  float4 a0[4], a1[4];
  prologue_loads(a0, a1, rows, nch, nch);
  f32x16 acc[NB];
#pragma unroll
  for (int n = 0; n < NB; ++n) acc[n] = zero16();

  while (cur < count) {
    const int nn = claim_unit(counter);
    unit_fence();
    unit_kloop<NB, NB, false>(acc, a0, a1, smem, sl, li, hi, rows, nch, nch);
    const int64_t row0 = (int64_t)(rb0 + cur) * 32;
    linear_epilogue<NB>(p, acc, tq, row0, col_base, lane, li, hi);
    cur = nxt;
    nxt = nn;
    rows.c0 = rows.c1 = rows.n0;
    rows.n0 = rows.n1 = rowp(nxt);
  }
}

// ---------------------------------------------------------------------------------------------------
// fused GRU cell: unit = 32 rows x 32 state features; slab = the 3 x 32 gate rows of the feature tile,
// each [W_ih row | W_hh row] (K = M + H); accumulators r, z, i_n, h_n share one C-fragment map.
// ---------------------------------------------------------------------------------------------------
struct GruArgs {
  const float *a; int64_t ld_a; const float *h; int64_t ld_h;
  const float *w_ih, *w_hh, *b_ih, *b_hh;
  int64_t n; int M, H;
  float *out; int64_t ld_out; float *gates;
  int nrb, ncs, rps, run_len;
  int lds_floats;
};

// sigmoid / tanh on the hardware transcendentals (v_exp_f32, v_rcp_f32: 1 ulp each): absolute error
// ~1.5e-7, against ~1700 VALU instructions per unit for the libm forms -- which made the epilogue 40 % of
// a unit's MFMA time
__device__ __forceinline__ float fast_sigmoid(float v) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
}

// The same gate math on PAIRS of elements: gfx950's packed fp32 ops (v_pk_add_f32 / v_pk_mul_f32) process two floats per
// lane per instruction with the results of the scalar ops (IEEE round-to-nearest per element), and VALU work is not
// hidden under fp32 MFMAs -- the ~300 adds / muls of a unit's gate math were 5 % of its MFMA time.  The transcendentals stay
// scalar.  Every expression below is the scalar one above, element for element: identical bits.
using f32x2 = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ f32x2 splat2(float v) { return f32x2{v, v}; }
__device__ __forceinline__ f32x2 rcp_1_plus_exp2(f32x2 t) {     // 1 / (1 + 2^t) per element
  const f32x2 d = f32x2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + splat2(1.0f);
  return f32x2{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
}

// Epilogue of one GRU unit (32 rows x 32 features), per row group of 8 rows: previous state in as one dwordx4 per
// lane (8 rows x 128 B of the h tile), un-transposed through the wave's slab into the C layout; gate math per lane;
// h' (and, in training, the gates) back through the slab as dwordx4 stores.  No mul+add contraction in the gate
// math: a row's result must not depend on which slot of the lane it occupies (sharded == unsharded bit for bit).
__device__ __forceinline__ void gru_epilogue(const GruArgs &p, const f32x16 (&acc)[4], float *tq, int64_t row0,
                                             int j0, int lane, int li, int hi, float bir, float biz, float bin,
                                             float bhr, float bhz, float bhn) {
  const int trow = lane >> 3, tcol = (lane & 7) * 4;
#ifndef PTGNN_GLOBAL_STORES
  // the unit's output rows (and, in training, its gate rows) through buffer descriptors that end with the last valid row
  const __amdgpu_buffer_rsrc_t out_rsrc = rows_descriptor(p.out + row0 * p.ld_out + j0, p.ld_out, p.n - row0, 32);
  const __amdgpu_buffer_rsrc_t gate_rsrc =
      rows_descriptor(p.gates ? p.gates + row0 * (int64_t)(4 * p.H) + j0 : p.out, 4 * (int64_t)p.H, p.n - row0, 3 * p.H + 32);
#endif
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    int64_t hrow = row0 + 8 * q + trow;
    const bool rvalid = hrow < p.n;
    hrow = rvalid ? hrow : p.n - 1;
    float hp[4];
    tq_untranspose(tq, lane, li, hi, *reinterpret_cast<const float4 *>(p.h + hrow * p.ld_h + j0 + tcol), hp);
    float res[4], rg[4], zg[4], ng[4], hn[4];
    {
#pragma clang fp contract(off)
#pragma unroll
      for (int i = 0; i < 4; i += 2) {
        const int r = 4 * q + i;
        const f32x2 ar = {acc[0][r], acc[0][r + 1]}, az = {acc[1][r], acc[1][r + 1]};
        const f32x2 an = {acc[2][r], acc[2][r + 1]}, ah = {acc[3][r], acc[3][r + 1]};
        const f32x2 hp2 = {hp[i], hp[i + 1]};
        const f32x2 r2 = rcp_1_plus_exp2(((ar + splat2(bir)) + splat2(bhr)) * splat2(-1.4426950408889634f));   // fast_sigmoid
        const f32x2 z2 = rcp_1_plus_exp2(((az + splat2(biz)) + splat2(bhz)) * splat2(-1.4426950408889634f));
        const f32x2 h2 = ah + splat2(bhn);
        const f32x2 pre = (an + splat2(bin)) + r2 * h2;
        const f32x2 n2 = splat2(2.0f) * rcp_1_plus_exp2(pre * splat2(-2.8853900817779268f)) - splat2(1.0f);     // fast_tanh
        const f32x2 o2 = (splat2(1.0f) - z2) * n2 + z2 * hp2;
        rg[i] = r2.x; rg[i + 1] = r2.y; zg[i] = z2.x; zg[i + 1] = z2.y; hn[i] = h2.x; hn[i + 1] = h2.y;
        ng[i] = n2.x; ng[i + 1] = n2.y; res[i] = o2.x; res[i + 1] = o2.y;
      }
    }
    const float4 o = tq_transpose(tq, lane, li, hi, res[0], res[1], res[2], res[3]);
#ifdef PTGNN_GLOBAL_STORES
    if (rvalid) *reinterpret_cast<float4 *>(p.out + hrow * p.ld_out + j0 + tcol) = o;
#else
    store_row4(out_rsrc, trow * (int)p.ld_out * 4 + tcol * 4, 8 * q * (int)p.ld_out * 4, o);   // rows past n: dropped
#endif
    if (p.gates) {   // training: r, z, n, gh_n for the backward ([n, 4H], gate-major)
      const float4 o_r = tq_transpose(tq, lane, li, hi, rg[0], rg[1], rg[2], rg[3]);
      const float4 o_z = tq_transpose(tq, lane, li, hi, zg[0], zg[1], zg[2], zg[3]);
      const float4 o_n = tq_transpose(tq, lane, li, hi, ng[0], ng[1], ng[2], ng[3]);
      const float4 o_h = tq_transpose(tq, lane, li, hi, hn[0], hn[1], hn[2], hn[3]);
#ifdef PTGNN_GLOBAL_STORES
      float *gp = p.gates + hrow * (int64_t)(4 * p.H) + j0 + tcol;
      if (rvalid) {
        *reinterpret_cast<float4 *>(gp) = o_r;
        *reinterpret_cast<float4 *>(gp + p.H) = o_z;
        *reinterpret_cast<float4 *>(gp + 2 * p.H) = o_n;
        *reinterpret_cast<float4 *>(gp + 3 * p.H) = o_h;
      }
#else
      const int gl = trow * 16 * p.H + tcol * 4, gu = 8 * q * 16 * p.H;       // bytes: lane part, uniform part
      store_row4(gate_rsrc, gl, gu, o_r);
      store_row4(gate_rsrc, gl, gu + 4 * p.H, o_z);
      store_row4(gate_rsrc, gl, gu + 8 * p.H, o_n);
      store_row4(gate_rsrc, gl, gu + 12 * p.H, o_h);
#endif
    }
  }
}

__global__ __launch_bounds__(512, 2) void k_stream_gru(GruArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NT = 512;
  const int b = blockIdx.x, G = gridDim.x;
  int slab, part;
  if (G % (kNumXcd * p.ncs) == 0) {
    const int xcd = b % kNumXcd, j = b / kNumXcd;
    slab = j % p.ncs;
    part = xcd + kNumXcd * (j / p.ncs);
  } else {
    slab = b / p.rps;
    part = b % p.rps;
  }
  const int rb0 = part * p.run_len;
  const int rb1 = rb0 + p.run_len < p.nrb ? rb0 + p.run_len : p.nrb;
  if (rb0 >= rb1) return;
  const int count = rb1 - rb0;

  const int K = p.M + p.H;
  const Slab sl(K, 96);
  int *counter = reinterpret_cast<int *>(smem + p.lds_floats);
  float *const tq = smem + p.lds_floats + 4 + (threadIdx.x >> 6) * kTqFloats;
  if (threadIdx.x == 0) *counter = 0;
  const int j0 = slab * 32;
  {
    const int mq = p.M >> 2;
    fill_slab<NT>(smem, sl, 96, K >> 2,
                         [&](int r, int q) {
                           const int gate = r >> 5, jj = j0 + (r & 31);   // H % 32 == 0: always a valid feature
                           return q < mq ? p.w_ih + ((int64_t)gate * p.H + jj) * p.M + q * 4
                                         : p.w_hh + ((int64_t)gate * p.H + jj) * p.H + (q - mq) * 4;
                         },
                         [](int) { return true; });
  }
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int li = lane & 31, hi = lane >> 5;
  const int ch0 = p.M >> 5, nch = K >> 5;
  const int lofs = lane_piece_offset(hi);
  auto clampr = [&](int u) {
    const int64_t row = (int64_t)(rb0 + (u < count ? u : count - 1)) * 32 + li;
    return row < p.n ? row : p.n - 1;
  };
  int cur = claim_unit(counter);
  if (cur >= count) return;
  int nxt = claim_unit(counter);
  ARows rows;
  {
    const int64_t r0 = clampr(cur), r1 = clampr(nxt);
    rows.c0 = p.a + r0 * p.ld_a + lofs; rows.c1 = p.h + r0 * p.ld_h + lofs;
    rows.n0 = p.a + r1 * p.ld_a + lofs; rows.n1 = p.h + r1 * p.ld_h + lofs;
  }
  float4 a0[4], a1[4];
  prologue_loads(a0, a1, rows, ch0, nch);
  f32x16 acc[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) acc[n] = zero16();

  const int j = j0 + li;
  const float bir = p.b_ih[j], biz = p.b_ih[p.H + j], bin = p.b_ih[2 * p.H + j];
  const float bhr = p.b_hh[j], bhz = p.b_hh[p.H + j], bhn = p.b_hh[2 * p.H + j];

  while (cur < count) {
    const int nn = claim_unit(counter);
    unit_fence();
    unit_kloop<3, 4, true>(acc, a0, a1, smem, sl, li, hi, rows, ch0, nch);
    gru_epilogue(p, acc, tq, (int64_t)(rb0 + cur) * 32, j0, lane, li, hi, bir, biz, bin, bhr, bhz, bhn);
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[n] = zero16();
    cur = nxt;
    nxt = nn;
    rows.c0 = rows.n0; rows.c1 = rows.n1;
    const int64_t r1 = clampr(nxt);
    rows.n0 = p.a + r1 * p.ld_a + lofs; rows.n1 = p.h + r1 * p.ld_h + lofs;
  }
}

// ---------------------------------------------------------------------------------------------------
// fused GRU cell whose weight slab does NOT fit LDS (K = M + H > ~400: BASELINE config 5, H = M = 256): the
// 3 x 32 gate rows of the feature tile stream through a double-buffered ring of 64-column panels instead (2 x 26 KB),
// refilled by the workgroup itself under its own MFMAs; A still goes straight from global memory into the MFMA
// operand registers.  The waves of a workgroup walk the panels in lockstep (one LDS barrier per 64 columns), so the
// workgroup is 4 waves and two of them share a CU: while one sits in its epilogue or at a barrier the other's waves
// keep the matrix pipes busy.  Chunks are consumed in the same order, with the same k permutation, as by
// k_stream_gru and the tile kernel: the three produce identical bits.
// ---------------------------------------------------------------------------------------------------
constexpr int kRingPanel = 64;                         // K columns per panel (two 32-wide chunks)
constexpr int kRingLd = kRingPanel + 4;                // floats per panel row
constexpr int kRingPanelFloats = 96 * kRingLd;
constexpr int kRingWaves = 4;

__global__ __launch_bounds__(kRingWaves * 64, 2) void k_stream_gru_ring(GruArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.x, G = gridDim.x;
  int slab, part;
  if (G % (kNumXcd * p.ncs) == 0) {
    const int xcd = b % kNumXcd, j = b / kNumXcd;
    slab = j % p.ncs;
    part = xcd + kNumXcd * (j / p.ncs);
  } else {
    slab = b / p.rps;
    part = b % p.rps;
  }
  const int rb0 = part * p.run_len;
  const int rb1 = rb0 + p.run_len < p.nrb ? rb0 + p.run_len : p.nrb;
  if (rb0 >= rb1) return;
  const int count = rb1 - rb0;
  const int rounds = (count + kRingWaves - 1) / kRingWaves;

  const int K = p.M + p.H;
  const int npan = K / kRingPanel;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, hi = lane >> 5;
  float *const tq = smem + 2 * kRingPanelFloats + wave * kTqFloats;
  const int j0 = slab * 32;
  const int ch0 = p.M >> 5, nch = K >> 5;

  // this thread's six float4 of a weight panel: element e = t + 256 q -> slab row e / 16 = (t >> 4) + 16 q, i.e.
  // gate q >> 1, feature j0 + (t >> 4) + 16 (q & 1); columns 4 (t & 15).  One per-thread offset, the rest is uniform
  // (kept as 32-bit offsets on purpose: twelve hoisted 64-bit row pointers spilled the kernel)
  const int trow = threadIdx.x >> 4, tc4 = (threadIdx.x & 15) * 4;
  const int base_ih = (j0 + trow) * p.M + tc4, base_hh = (j0 + trow) * p.H + tc4;
  auto panel_src = [&](int pan, int q) -> const float * {
    const int k0 = pan * kRingPanel;                  // M % 64 == 0: a panel lies in W_ih or in W_hh
    const int rq = (q >> 1) * p.H + (q & 1) * 16;     // row offset of this float4 inside the [3H, .] matrix
    return k0 < p.M ? p.w_ih + (base_ih + rq * p.M + k0) : p.w_hh + (base_hh + rq * p.H + (k0 - p.M));
  };
  auto panel_dst = [&](int buf, int q) -> float * {
    return smem + buf * kRingPanelFloats + (trow + 16 * q) * kRingLd + tc4;
  };
  {
#pragma unroll
    for (int q = 0; q < 6; ++q) *reinterpret_cast<float4 *>(panel_dst(0, q)) = *reinterpret_cast<const float4 *>(panel_src(0, q));
  }
  __syncthreads();

  const int lofs = lane_piece_offset(hi);
  auto clampr = [&](int u) {
    const int64_t row = (int64_t)(rb0 + (u < count ? u : count - 1)) * 32 + li;
    return row < p.n ? row : p.n - 1;
  };
  ARows rows;
  {
    const int64_t r0 = clampr(wave), r1 = clampr(wave + kRingWaves);
    rows.c0 = p.a + r0 * p.ld_a + lofs; rows.c1 = p.h + r0 * p.ld_h + lofs;
    rows.n0 = p.a + r1 * p.ld_a + lofs; rows.n1 = p.h + r1 * p.ld_h + lofs;
  }
  float4 a0[4], a1[4];
  prologue_loads(a0, a1, rows, ch0, nch);
  f32x16 acc[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) acc[n] = zero16();

  const int j = j0 + li;
  const float bir = p.b_ih[j], biz = p.b_ih[p.H + j], bin = p.b_ih[2 * p.H + j];
  const float bhr = p.b_hh[j], bhz = p.b_hh[p.H + j], bhn = p.b_hh[2 * p.H + j];
  const int cbs = 32 * kRingLd;

  int P = 0;                                          // panels consumed so far: panel P sits in buffer P & 1
  for (int i = 0; i < rounds; ++i) {
    const int u = i * kRingWaves + wave;              // this wave's unit of the round (may be past the run: computed
    unit_fence();                                     // on clamped rows, not stored -- every wave keeps the barriers)
    // one panel = two chunks; the next panel's rows ride under its MFMAs in two halves (3 + 3 float4 per thread:
    // registers), each written into the buffer every wave left at the last barrier once its chunk is done
#define PTGNN_RING_PANEL(PH)                                                                                   \
  do {                                                                                                         \
    const float *bl = smem + (P & 1) * kRingPanelFloats + li * kRingLd + hi * 4;                               \
    float *const nb = smem + ((P + 1) & 1) * kRingPanelFloats + trow * kRingLd + tc4;                          \
    const int nextpan = pan + 1 < npan ? pan + 1 : 0;                                                          \
    float4 w0 = *reinterpret_cast<const float4 *>(panel_src(nextpan, 0));                                      \
    float4 w1 = *reinterpret_cast<const float4 *>(panel_src(nextpan, 1));                                      \
    float4 w2 = *reinterpret_cast<const float4 *>(panel_src(nextpan, 2));                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    float4 bcur[3];                                                                                            \
    _Pragma("unroll") for (int n = 0; n < 3; ++n) bcur[n] = *reinterpret_cast<const float4 *>(bl + n * cbs);   \
    chunk_f32<3, 4, PH, true>(acc, a0, bcur, bl, cbs, 0, 32, rows, 2 * pan + 2, ch0, nch);                     \
    *reinterpret_cast<float4 *>(nb) = w0;                                                                      \
    *reinterpret_cast<float4 *>(nb + 16 * kRingLd) = w1;                                                       \
    *reinterpret_cast<float4 *>(nb + 32 * kRingLd) = w2;                                                       \
    w0 = *reinterpret_cast<const float4 *>(panel_src(nextpan, 3));                                             \
    w1 = *reinterpret_cast<const float4 *>(panel_src(nextpan, 4));                                             \
    w2 = *reinterpret_cast<const float4 *>(panel_src(nextpan, 5));                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    chunk_f32<3, 4, PH, true>(acc, a1, bcur, bl, cbs, 32, 0, rows, 2 * pan + 3, ch0, nch);                     \
    *reinterpret_cast<float4 *>(nb + 48 * kRingLd) = w0;                                                       \
    *reinterpret_cast<float4 *>(nb + 64 * kRingLd) = w1;                                                       \
    *reinterpret_cast<float4 *>(nb + 80 * kRingLd) = w2;                                                       \
    lds_barrier();   /* panel P + 1 is complete; everyone is done reading panel P */                           \
  } while (0)
    int pan = 0;
    for (; pan < (ch0 >> 1); ++pan, ++P) PTGNN_RING_PANEL(0);
    for (; pan < npan; ++pan, ++P) PTGNN_RING_PANEL(1);
#undef PTGNN_RING_PANEL
    if (u < count) gru_epilogue(p, acc, tq, (int64_t)(rb0 + u) * 32, j0, lane, li, hi, bir, biz, bin, bhr, bhz, bhn);
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[n] = zero16();
    rows.c0 = rows.n0; rows.c1 = rows.n1;
    const int64_t r1 = clampr(u + 2 * kRingWaves);
    rows.n0 = p.a + r1 * p.ld_a + lofs; rows.n1 = p.h + r1 * p.ld_h + lofs;
  }
}

// ---------------------------------------------------------------------------------------------------
// linear whose [128, K] weight slab does not fit LDS (K > ~300: the GRU's input gradients  d_a = d_gates . W_ih,
// d_h = d_gates . W_hh  with K = 3 H = 384 at H = 128, which the slab kernel sent back to the tile kernel at 0.64 of
// the MFMA peak): the same panel ring as k_stream_gru_ring, 128 output columns x 64 K columns per panel (2 x 34 KB),
// four waves a workgroup, two workgroups a CU.  Same chunk order and k permutation as k_stream_linear and the tile
// kernel: identical bits.
// ---------------------------------------------------------------------------------------------------
constexpr int kLinRingPanelFloats = 128 * kRingLd;

__global__ __launch_bounds__(kRingWaves * 64, 2) void k_stream_linear_ring(LinearArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.x, G = gridDim.x;
  int slab, part;
  if (G % (kNumXcd * p.ncs) == 0) {
    const int xcd = b % kNumXcd, j = b / kNumXcd;
    slab = j % p.ncs;
    part = xcd + kNumXcd * (j / p.ncs);
  } else {
    slab = b / p.rps;
    part = b % p.rps;
  }
  const int rb0 = part * p.run_len;
  const int rb1 = rb0 + p.run_len < p.nrb ? rb0 + p.run_len : p.nrb;
  if (rb0 >= rb1) return;
  const int count = rb1 - rb0;
  const int rounds = (count + kRingWaves - 1) / kRingWaves;

  const int K = p.K;
  const int npan = K / kRingPanel, nch = K >> 5;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, hi = lane >> 5;
  float *const tq = smem + 2 * kLinRingPanelFloats + wave * kTqFloats;
  const int col_base = slab * 128;

  // this thread's eight float4 of a weight panel: element e = t + 256 q -> panel row (t >> 4) + 16 q, columns 4 (t & 15)
  const int trow = threadIdx.x >> 4, tc4 = (threadIdx.x & 15) * 4;
  const int wbase = (col_base + trow) * K + tc4;        // 32-bit offsets (see k_stream_gru_ring)
  const int wq = 16 * K;
  {
    float *const d = smem + trow * kRingLd + tc4;
#pragma unroll
    for (int q = 0; q < 8; ++q)
      *reinterpret_cast<float4 *>(d + 16 * q * kRingLd) = *reinterpret_cast<const float4 *>(p.w + (wbase + q * wq));
  }
  __syncthreads();

  const int lofs = lane_piece_offset(hi);
  auto rowp = [&](int u) {
    int64_t row = (int64_t)(rb0 + (u < count ? u : count - 1)) * 32 + li;
    row = row < p.rows ? row : p.rows - 1;
    return p.x + row * p.ld_x + lofs;
  };
  ARows rows;
  rows.c0 = rows.c1 = rowp(wave);
  rows.n0 = rows.n1 = rowp(wave + kRingWaves);
  float4 a0[4], a1[4];
  prologue_loads(a0, a1, rows, nch, nch);
  f32x16 acc[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) acc[n] = zero16();
  const int cbs = 32 * kRingLd;

  int P = 0;                                          // panels consumed so far: panel P sits in buffer P & 1
  for (int i = 0; i < rounds; ++i) {
    const int u = i * kRingWaves + wave;              // may be past the run: computed on clamped rows, not stored
    unit_fence();
    for (int pan = 0; pan < npan; ++pan, ++P) {
      const float *bl = smem + (P & 1) * kLinRingPanelFloats + li * kRingLd + hi * 4;
      float *const nb = smem + ((P + 1) & 1) * kLinRingPanelFloats + trow * kRingLd + tc4;
      const float *const src = p.w + (wbase + (pan + 1 < npan ? pan + 1 : 0) * kRingPanel);
      float4 w0 = *reinterpret_cast<const float4 *>(src);
      float4 w1 = *reinterpret_cast<const float4 *>(src + wq);
      float4 w2 = *reinterpret_cast<const float4 *>(src + 2 * wq);
      float4 w3 = *reinterpret_cast<const float4 *>(src + 3 * wq);
      __builtin_amdgcn_sched_barrier(0);
      float4 bcur[4];
#pragma unroll
      for (int n = 0; n < 4; ++n) bcur[n] = *reinterpret_cast<const float4 *>(bl + n * cbs);
      chunk_f32<4, 4, 0, false>(acc, a0, bcur, bl, cbs, 0, 32, rows, 2 * pan + 2, nch, nch);
      *reinterpret_cast<float4 *>(nb) = w0;
      *reinterpret_cast<float4 *>(nb + 16 * kRingLd) = w1;
      *reinterpret_cast<float4 *>(nb + 32 * kRingLd) = w2;
      *reinterpret_cast<float4 *>(nb + 48 * kRingLd) = w3;
      w0 = *reinterpret_cast<const float4 *>(src + 4 * wq);
      w1 = *reinterpret_cast<const float4 *>(src + 5 * wq);
      w2 = *reinterpret_cast<const float4 *>(src + 6 * wq);
      w3 = *reinterpret_cast<const float4 *>(src + 7 * wq);
      __builtin_amdgcn_sched_barrier(0);
      chunk_f32<4, 4, 0, false>(acc, a1, bcur, bl, cbs, 32, 0, rows, 2 * pan + 3, nch, nch);
      *reinterpret_cast<float4 *>(nb + 64 * kRingLd) = w0;
      *reinterpret_cast<float4 *>(nb + 80 * kRingLd) = w1;
      *reinterpret_cast<float4 *>(nb + 96 * kRingLd) = w2;
      *reinterpret_cast<float4 *>(nb + 112 * kRingLd) = w3;
      lds_barrier();   // panel P + 1 is complete; everyone is done reading panel P
    }
    if (u < count) linear_epilogue<4>(p, acc, tq, (int64_t)(rb0 + u) * 32, col_base, lane, li, hi);
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[n] = zero16();
    rows.c0 = rows.c1 = rows.n0;
    rows.n0 = rows.n1 = rowp(u + 2 * kRingWaves);
  }
}

// ---------------------------------------------------------------------------------------------------
// grouped per-edge GEMM: unit = 32 edges of one type; flat balanced runs over the type-major unit order;
// a run reloads the slab (W_t) when it crosses into the next edge type.
// ---------------------------------------------------------------------------------------------------
// The two prefix arrays of an edge table (65 entries each, those past num_types hold the totals), one entry per lane,
// searched with a ballot; single entries come back through v_readlane.
struct TablePrefixes {
  int v_wg, v_unit, wg_total, unit_total;
  __device__ __forceinline__ explicit TablePrefixes(const StreamEdgeTable &tab) {
    const int l = threadIdx.x & 63;
    v_wg = tab.wg_off[l];
    v_unit = tab.unit_off[l];
    wg_total = tab.wg_off[kStreamMaxTypes];
    unit_total = tab.unit_off[kStreamMaxTypes];
  }
  __device__ __forceinline__ static int entry(int v, int total, int i) {   // prefix entry i (wave-uniform)
    return i >= kStreamMaxTypes ? total : __builtin_amdgcn_readlane(v, i);
  }
  // units [u, u_end) of workgroup `wg`: its share of the units of the type that owns it
  __device__ __forceinline__ void run_of_workgroup(int wg, int &u, int &u_end) const {
    const int lo = __popcll(__ballot(v_wg <= wg)) - 1;     // last type whose first workgroup is <= this one
    const int wg0 = entry(v_wg, wg_total, lo), u0 = entry(v_unit, unit_total, lo);
    const int64_t w = entry(v_wg, wg_total, lo + 1) - wg0, part = wg - wg0;
    const int64_t units = entry(v_unit, unit_total, lo + 1) - u0;
    u = u0 + (int)(part * units / w);
    u_end = u0 + (int)((part + 1) * units / w);
  }
  __device__ __forceinline__ void type_of_unit(int u, int &t, int &u0, int &u1) const {
    t = __popcll(__ballot(v_unit <= u)) - 1;
    u0 = entry(v_unit, unit_total, t);
    u1 = entry(v_unit, unit_total, t + 1);
  }
};

struct EdgeArgs {
  StreamEdgeTable tab;
  const StreamEdgeTable *tab_dev;   // INDIRECT kernels: the table in device memory (tab then only carries w / num_types)
  const float *x; int64_t ld_x; int H; int use_dst; int M; int act;
  float *msg; int64_t ld_msg; int64_t msg_row_base;
  int64_t num_rows;          // rows of x (gathered ids are clamped into it)
  int run_len, lds_floats;
};

// INDIRECT: edge counts, unit split and workgroup apportioning are read from a table in DEVICE memory that an earlier
// kernel of the same stream wrote (ptgnn_amd_unique_sources: the row counts of the shared-message form are only
// known on the device, and fetching them would stall the host once per minibatch); the weights still come by value.
// Waves per workgroup of the edge kernel (A/B knob, scripts/build_variant.sh ... -DPTGNN_EDGE_WAVES=12): a third wave per
// SIMD would cover the per-unit wait for the store acknowledgements (profiles/r03_notes.md 12) if the kernel fits 170
// VGPRs.  8 is what ships.
#ifndef PTGNN_EDGE_WAVES
#define PTGNN_EDGE_WAVES 8
#endif
constexpr int kEdgeWaves = PTGNN_EDGE_WAVES;

template <int NB, bool INDIRECT>
__global__ __launch_bounds__(kEdgeWaves * 64, kEdgeWaves == 8 ? 2 : 3) void k_stream_edge(EdgeArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NT = kEdgeWaves * 64, BN = 32 * NB;
  const StreamEdgeTable *tabp;
  if constexpr (INDIRECT) tabp = p.tab_dev; else tabp = &p.tab;
  const StreamEdgeTable &tab = *tabp;
  // INDIRECT: the two prefix arrays of the device-resident table are fetched ONCE, one entry per lane (the writer fills all
  // 65 entries, those past num_types with the totals), and searched with a ballot: a binary search over device memory
  // is five dependent L2 round trips, twice, in front of every launch's first MFMA (profiles/r04_notes.md 9)
  // (the by-value table of the kernel arguments is searched the same way: its binary searches were ten dependent scalar
  //  loads; stream_edge() fills the entries past num_types like the device-side writer)
  const TablePrefixes tp(tab);
  if constexpr (INDIRECT) {
    if ((int)blockIdx.x >= tp.wg_total) return;   // the launch is sized for the largest apportioning
  }
  // Workgroups are apportioned to edge types in proportion to their units, and a type's units are split evenly
  // over its workgroups: no run crosses a type boundary (a mid-run slab reload + barrier made the ~T affected
  // workgroups the stragglers that set the kernel time).
  int u, u_end;
  tp.run_of_workgroup((int)blockIdx.x, u, u_end);
  if (u >= u_end) return;
  const int K = p.use_dst ? 2 * p.H : p.H;
  const Slab sl(K, BN);
  int *counter = reinterpret_cast<int *>(smem + p.lds_floats);
  float *const tq = smem + p.lds_floats + 4 + (threadIdx.x >> 6) * kTqFloats;
  const int lane = threadIdx.x & 63;
  const int li = lane & 31, hi = lane >> 5;
  const int ch0 = p.H >> 5, nch = K >> 5;
  const int lofs = lane_piece_offset(hi);

  while (u < u_end) {
    int t, t_u0, t_u1;                  // edge type of unit u and the type's unit range
    tp.type_of_unit(u, t, t_u0, t_u1);
    const int seg_end = t_u1 < u_end ? t_u1 : u_end;
    const int ub = u - t_u0;                        // first unit of the segment inside the type
    const int count = seg_end - u;
    // every table field the unit loop needs is read HERE: with the table in device memory (INDIRECT) a read inside the
    // loop -- after the kernel's own stores -- cannot be proven unclobbered, becomes a vector load whose result is needed
    // at once, and the `s_waitcnt vmcnt(0)` in front of it makes the compiler give up counting the A-row loads of the
    // whole loop (the table-driven kernel waited for every refill in full: 17.6 % of its wave cycles)
    const int64_t type_row0 = tab.edge_off[t];
    const int64_t n_edges = tab.edge_off[t + 1] - type_row0;
    // The id lists are GLOBAL memory.  Read out of a table in device memory (INDIRECT) the pointers are generic as far as
    // the compiler knows, their loads become flat_load, and a pending FLAT access makes the wait-count pass treat the
    // whole queue as out of order: `s_waitcnt vmcnt(0) lgkmcnt(0)` in front of every chunk pair -- every A-row refill
    // waited for in full, every B-fragment read drained -- while the ids of the unit after next are in flight, i.e.
    // always (found in the ISA of the table-driven kernel; the by-value kernel, whose pointers come from the kernel
    // arguments, counts `vmcnt(7) .. (4)` per piece).
    using GlobalIds = const __attribute__((address_space(1))) int64_t *;
    const GlobalIds src = (GlobalIds)tab.src[t];
    const GlobalIds dst = (GlobalIds)tab.dst[t];
    __syncthreads();   // every wave is done with the previous slab and its counter
    if (threadIdx.x == 0) *counter = 0;
    {
      const float *w = p.tab.w[t];
      fill_slab<NT>(smem, sl, BN, K >> 2,
                           [&](int r, int q) { return w + (int64_t)(r < p.M ? r : p.M - 1) * K + q * 4; },
                           [&](int r) { return r < p.M; });
    }
    __syncthreads();

    int cur = claim_unit(counter);
    if (cur < count) {
      int nxt = claim_unit(counter);
      int nn = claim_unit(counter);   // the node ids of a unit are fetched two units ahead of its rows
          auto edge_of = [&](int unit) {
        const int64_t e = (int64_t)(ub + (unit < count ? unit : count - 1)) * 32 + li;
        return e < n_edges ? e : n_edges - 1;
      };
      auto node_row = [&](int64_t id) {   // ids were range-checked by the plan build; clamp anyway
        id = id < 0 ? 0 : id;
        return p.x + (id < p.num_rows ? id : p.num_rows - 1) * p.ld_x + lofs;
      };
      ARows rows;
      {
        const int64_t e0 = edge_of(cur), e1 = edge_of(nxt);
        rows.c0 = node_row(src[e0]); rows.c1 = node_row(dst[e0]);
        rows.n0 = node_row(src[e1]); rows.n1 = node_row(dst[e1]);
      }
      float4 a0[4], a1[4];
      prologue_loads(a0, a1, rows, ch0, nch);
      f32x16 acc[NB];
#pragma unroll
      for (int n = 0; n < NB; ++n) acc[n] = zero16();
      while (cur < count) {
        const int n3 = claim_unit(counter);
        const int64_t e_nn = edge_of(nn);
        unit_fence();
        const int64_t s_nn = src[e_nn], d_nn = dst[e_nn];   // lands under this unit's MFMAs
        unit_kloop<NB, NB, false>(acc, a0, a1, smem, sl, li, hi, rows, ch0, nch);
        const int64_t e_row0 = (int64_t)(ub + cur) * 32;
        const int64_t out_row0 = p.msg_row_base + type_row0 + e_row0;
#pragma unroll
        for (int n = 0; n < NB; ++n) {
          if (n * 32 < p.M) {   // host guarantees 16-byte aligned message rows
            float *yb = p.msg + out_row0 * p.ld_msg + n * 32;
            const int64_t left = n_edges - e_row0;
            if (p.act == PTGNN_AMD_ACT_TANH) store_block_tq<PTGNN_AMD_ACT_TANH, false>(acc[n], tq, yb, p.ld_msg, 0.f, left, lane, li, hi);
            else if (p.act == PTGNN_AMD_ACT_RELU) store_block_tq<PTGNN_AMD_ACT_RELU, false>(acc[n], tq, yb, p.ld_msg, 0.f, left, lane, li, hi);
            else store_block_tq<PTGNN_AMD_ACT_NONE, false>(acc[n], tq, yb, p.ld_msg, 0.f, left, lane, li, hi);
          }
          acc[n] = zero16();
        }
        cur = nxt; nxt = nn; nn = n3;
        rows.c0 = rows.n0; rows.c1 = rows.n1;
        rows.n0 = node_row(s_nn); rows.n1 = node_row(d_nn);
      }
    }
    u = seg_end;
  }
}

// ---------------------------------------------------------------------------------------------------
// grouped per-edge GEMM with the reference's per-edge dropout (gatedmessagepassing.py:57-61, training): exact fp32,
// K in {64, 128, 256}, M in {64, 128}, no activation.  The keep mask arrives as ONE BIT per element
// (`ptgnn_amd_dropout_bitmask`: the hash of dense_common.h evaluated once per layer call instead of inside three GEMMs --
// the hash costs ~55 VALU instructions per 16 MFMAs, and fp32 MFMA shares the vector lanes with the VALU):
//   DROP 1  mask on the gathered INPUT rows (forward): one mask dword per (message row, 32-column chunk) rides with the
//           chunk's four row pieces; 3-4 VALU per element (bit -> all-ones, scale, and);
//   DROP 2  mask on the OUTPUT rows (input gradient, d in = (d msg . W) * mask): the row's mask dwords are fetched at the
//           top of the unit and applied to the transposed float4 right before its store.
// Same chunk order, k permutation and MFMA sequence per accumulator as k_stream_edge and the tile kernel: identical bits.
//
// The unit body is straight-line -- the K loop is unrolled over a compile-time chunk count and every row store is issued
// unconditionally (lanes of rows past a type's end write to a sink buffer instead of being masked off) -- so the
// compiler counts every wait across the unit boundary: `s_waitcnt vmcnt(N)` per piece, no drain anywhere in the loop.
// This was built as the "fence-free" form of k_stream_edge and MEASURED AS SUCH (DROP 0, same shapes,
// alternating launches in one process, profiles/r04_notes.md 1): with the queue shape of the prologue matched to the
// loop body (sixteen sink stores, so that the unit starts on `vmcnt(25)` and never waits for a store acknowledgement)
// it is 4-7 % SLOWER than k_stream_edge with its per-unit drain; without them (a small `vmcnt` at the unit top that does
// cover the previous unit's stores, like the drain) it is level with it; re-inserting the drain costs nothing; sending
// every store to the cache-resident sink gains 2 %.  So the stores' acknowledgement was never what the edge GEMM waits
// for (PMC: 6 % of wave cycles in s_waitcnt here against 11 % there, at MORE total cycles).  k_stream_edge therefore
// stays the DROP 0 kernel and only the dropout forms are instantiated from this one.
// ---------------------------------------------------------------------------------------------------
}  // namespace

// sink of the unconditional stores (rows past a type's end): external linkage, so the stores cannot be proven dead
__device__ float4 ptgnn_amd_edge_store_sink[64 + 32];

namespace {

struct EdgeV2Args {
  EdgeArgs e;
  const uint32_t *mask;     // DROP != 0: keep bits, [message rows][mask_ld] dwords, bit b of dword c = column 32 c + b
  int mask_ld;              // dwords per mask row
  int mask_col0;            // DROP == 2: first dword of this launch's column slab
  float scale;              // 1 / (1 - p)
};

__device__ __forceinline__ float keep_scaled(float v, uint32_t m, int bit, float scale) {
  const int t = ((int)(m << (31 - bit))) >> 31;   // v_bfe_i32: the keep bit spread over the word
  return __uint_as_float(__float_as_uint(v * scale) & (uint32_t)t);
}

// one K chunk: MFMAs on the four pieces in `buf` (+ mask word `mw`), then the refill of `buf` (+ `mw`) from `refill`
template <int NB, int DROP>
__device__ __forceinline__ void chunk_v2(f32x16 (&acc)[NB], float4 (&buf)[4], uint32_t &mw, float4 (&bcur)[NB],
                                         const float *bl, int cbs, int kofs, int kofs_next, const float *refill,
                                         const uint32_t *mrefill, int hi, float scale) {
  uint32_t m = 0;
  if constexpr (DROP == 1) m = mw >> (4 * hi);
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float4 bn[NB];
    const int ko = g < 3 ? kofs + (g + 1) * 8 : kofs_next;
#pragma unroll
    for (int n = 0; n < NB; ++n) bn[n] = *reinterpret_cast<const float4 *>(bl + n * cbs + ko);
    __builtin_amdgcn_sched_barrier(0);
    float4 a = buf[g];
    if constexpr (DROP == 1) {
      a.x = keep_scaled(a.x, m, 8 * g + 0, scale); a.y = keep_scaled(a.y, m, 8 * g + 1, scale);
      a.z = keep_scaled(a.z, m, 8 * g + 2, scale); a.w = keep_scaled(a.w, m, 8 * g + 3, scale);
    }
#define PTGNN_STEP(C)                                                                     \
    _Pragma("unroll") for (int n = 0; n < NB; ++n)                                        \
      acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.C, bcur[n].C, acc[n], 0, 0, 0);
    PTGNN_STEP(x) PTGNN_STEP(y) PTGNN_STEP(z) PTGNN_STEP(w)
#undef PTGNN_STEP
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int n = 0; n < NB; ++n) bcur[n] = bn[n];
  }
  if constexpr (DROP == 1) {   // first: the chunk's pieces are waited for one at a time, and the mask with the first
    mw = *mrefill;
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    buf[g] = *reinterpret_cast<const float4 *>(refill + g * 8);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// VAR: 0 = table by value, 1 = table in device memory (shared message rows), 2 = by value with a target-state half
// (input row = [x_src ; x_dst], K = 2 H: the first NCH / 2 chunks come from the source row)
template <int NB, int NCH, int DROP, int VAR>
__global__ __launch_bounds__(512, 2) void k_stream_edge_v2(EdgeV2Args q) {
  constexpr bool INDIRECT = VAR == 1;
  constexpr int CH0 = VAR == 2 ? NCH / 2 : NCH;   // chunks of the source half
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NT = 512, BN = 32 * NB, K = 32 * NCH;
  const EdgeArgs &p = q.e;
  const StreamEdgeTable *tabp;
  if constexpr (INDIRECT) tabp = p.tab_dev; else tabp = &p.tab;
  const StreamEdgeTable &tab = *tabp;
  const TablePrefixes tp(tab);
  if constexpr (INDIRECT) {
    if ((int)blockIdx.x >= tp.wg_total) return;
  }
  int u, u_end;
  tp.run_of_workgroup((int)blockIdx.x, u, u_end);
  if (u >= u_end) return;
  const Slab sl(K, BN);
  int *counter = reinterpret_cast<int *>(smem + p.lds_floats);
  float *const tq = smem + p.lds_floats + 4 + (threadIdx.x >> 6) * kTqFloats;
  const int lane = threadIdx.x & 63;
  const int li = lane & 31, hi = lane >> 5;
  const int lofs = hi * 4;
  const int srow = lane >> 3, scol = (lane & 7) * 4;          // this lane's row / column in the store layout
  [[maybe_unused]] float *const sink = reinterpret_cast<float *>(ptgnn_amd_edge_store_sink) + lane * 4;
  const float *const bl = smem + li * sl.ld + hi * 4;
  const int cbs = 32 * sl.ld;

  while (u < u_end) {
    int t, t_u0, t_u1;
    tp.type_of_unit(u, t, t_u0, t_u1);
    const int seg_end = t_u1 < u_end ? t_u1 : u_end;
    const int ub = u - t_u0;
    const int count = seg_end - u;
    const int64_t type_row0 = tab.edge_off[t];
    const int64_t n_edges = tab.edge_off[t + 1] - type_row0;
    using GlobalIds = const __attribute__((address_space(1))) int64_t *;
    const GlobalIds src = (GlobalIds)tab.src[t];
    const GlobalIds dst = (GlobalIds)tab.dst[t];
    __syncthreads();
    if (threadIdx.x == 0) *counter = 0;
    {
      const float *w = p.tab.w[t];
      fill_slab<NT>(smem, sl, BN, K >> 2, [&](int r, int q) { return w + (int64_t)r * K + q * 4; },
                           [](int) { return true; });
    }
    __syncthreads();

    int cur = claim_unit(counter);
    if (cur < count) {
      int nxt = claim_unit(counter);
      auto edge_of = [&](int unit) {
        const int64_t e = (int64_t)(ub + (unit < count ? unit : count - 1)) * 32 + li;
        return e < n_edges ? e : n_edges - 1;
      };
      auto node_row = [&](int64_t id) {
        id = id < 0 ? 0 : id;
        return p.x + (id < p.num_rows ? id : p.num_rows - 1) * p.ld_x + lofs;
      };
      // chunk cc of a unit whose source / target rows are r0 / r1
      auto chunk_ptr = [&](const float *r0, const float *r1, int cc) {
        return cc < CH0 ? r0 + cc * 32 : r1 + (cc - CH0) * 32;
      };
      const int64_t mrow_base = p.msg_row_base + type_row0;
      auto mask_row = [&](int64_t e) { return q.mask + (mrow_base + e) * q.mask_ld; };   // DROP == 1: e = edge_of(unit)

      const float *c0, *c1, *n0, *n1;           // row pointers of the current and the next unit
      const uint32_t *mc = nullptr, *mn = nullptr;
      {
        const int64_t e0 = edge_of(cur);
        c0 = node_row(src[e0]); c1 = node_row(dst[e0]);
        if constexpr (DROP == 1) mc = mask_row(e0);
      }
      float4 a0[4], a1[4];
      uint32_t m0 = 0, m1 = 0;
      if constexpr (DROP == 1) { m0 = mc[0]; __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        a0[g] = *reinterpret_cast<const float4 *>(chunk_ptr(c0, c1, 0) + g * 8);
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (DROP == 1) { m1 = mc[1]; __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        a1[g] = *reinterpret_cast<const float4 *>(chunk_ptr(c0, c1, 1) + g * 8);
        __builtin_amdgcn_sched_barrier(0);
      }
      int64_t e_n = edge_of(nxt);
      int64_t s_n = src[e_n], d_n = dst[e_n];    // consumed inside the first unit
      __builtin_amdgcn_sched_barrier(0);
      f32x16 acc[NB];
#pragma unroll
      for (int n = 0; n < NB; ++n) acc[n] = zero16();
      float4 bcur[NB];
#pragma unroll
      for (int n = 0; n < NB; ++n) bcur[n] = *reinterpret_cast<const float4 *>(bl + n * cbs);

      while (true) {
        const int n2 = claim_unit(counter);
        const int64_t e_row0 = (int64_t)(ub + cur) * 32;
        const int64_t out_row0 = mrow_base + e_row0;
        const int64_t left = n_edges - e_row0;
        uint4 mq[4];
        if constexpr (DROP == 2) {   // keep bits of this lane's four output rows, all column blocks of the slab
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int64_t rr = 8 * r + srow < left ? 8 * r + srow : left - 1;
            const uint32_t *mp = q.mask + (out_row0 + rr) * q.mask_ld + q.mask_col0;
            if constexpr (NB == 4) mq[r] = *reinterpret_cast<const uint4 *>(mp);
            else { const uint2 v = *reinterpret_cast<const uint2 *>(mp); mq[r] = make_uint4(v.x, v.y, 0u, 0u); }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
#pragma unroll
        for (int c = 0; c < NCH; c += 2) {
          if (c == NCH - 2) {        // the next unit's rows are first needed by this chunk pair's refills
            __builtin_amdgcn_sched_barrier(0);
            // the ids were requested during the previous unit's epilogue: pin their first use HERE (the compiler hoists the
            // address arithmetic -- and with it the wait for the ids -- to the top of the unit otherwise)
            asm volatile("" : "+v"(s_n), "+v"(d_n));
            n0 = node_row(s_n); n1 = node_row(d_n);
            if constexpr (DROP == 1) mn = mask_row(e_n);
            __builtin_amdgcn_sched_barrier(0);
          }
          const float *r0 = c + 2 < NCH ? chunk_ptr(c0, c1, c + 2) : chunk_ptr(n0, n1, 0);
          const float *r1 = c + 2 < NCH ? chunk_ptr(c0, c1, c + 3) : chunk_ptr(n0, n1, 1);
          const uint32_t *mr0 = nullptr, *mr1 = nullptr;
          if constexpr (DROP == 1) {
            mr0 = c + 2 < NCH ? mc + (c + 2) : mn;
            mr1 = c + 2 < NCH ? mc + (c + 3) : mn + 1;
          }
          const int kn = c + 2 < NCH ? c * 32 + 64 : 0;
          chunk_v2<NB, DROP>(acc, a0, m0, bcur, bl, cbs, c * 32, c * 32 + 32, r0, mr0, hi, q.scale);
          chunk_v2<NB, DROP>(acc, a1, m1, bcur, bl, cbs, c * 32 + 32, kn, r1, mr1, hi, q.scale);
        }
        e_n = edge_of(n2);
        s_n = src[e_n]; d_n = dst[e_n];          // consumed one unit later
        __builtin_amdgcn_sched_barrier(0);
        // epilogue: C fragments -> rows through the wave's transposing slab, every store issued
        {
          float *const yp = p.msg + (out_row0 + srow) * p.ld_msg + scol;
          (void)yp;
#ifndef PTGNN_GLOBAL_STORES
          const __amdgpu_buffer_rsrc_t msg_rsrc = rows_descriptor(p.msg + out_row0 * p.ld_msg, p.ld_msg, left, 32 * NB);
          const int ldb = (int)p.ld_msg * 4;
#endif
          // column block outer, row group inner (k_stream_edge's order; the other nesting measured 4 % slower)
#pragma unroll
          for (int n = 0; n < NB; ++n) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#ifdef PTGNN_GLOBAL_STORES
              float *const yr = 8 * r + srow < left ? yp + (int64_t)(8 * r) * p.ld_msg : sink;
#endif
              float4 o = tq_transpose(tq, lane, li, hi, acc[n][4 * r], acc[n][4 * r + 1], acc[n][4 * r + 2],
                                      acc[n][4 * r + 3]);
              if constexpr (DROP == 2) {
                const uint32_t w = n == 0 ? mq[r].x : n == 1 ? mq[r].y : n == 2 ? mq[r].z : mq[r].w;
                const uint32_t m = w >> scol;
                o.x = keep_scaled(o.x, m, 0, q.scale); o.y = keep_scaled(o.y, m, 1, q.scale);
                o.z = keep_scaled(o.z, m, 2, q.scale); o.w = keep_scaled(o.w, m, 3, q.scale);
              }
#ifdef PTGNN_GLOBAL_STORES
              *reinterpret_cast<float4 *>(yr + n * 32) = o;
#else
              store_row4(msg_rsrc, srow * ldb + (scol + n * 32) * 4, 8 * r * ldb, o);   // rows past the type's end: dropped
#endif
              __builtin_amdgcn_sched_barrier(0);
            }
          }
#pragma unroll
          for (int n = 0; n < NB; ++n) acc[n] = zero16();
        }
        if (nxt >= count) break;
        cur = nxt; nxt = n2;
        c0 = n0; c1 = n1;
        if constexpr (DROP == 1) mc = mn;
      }
    }
    u = seg_end;
  }
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
std::atomic<int> g_mode{-1};   // test / developer switch: relaxed, results never depend on it

bool debug_on() {
  static int v = -1;
  if (v < 0) v = getenv("PTGNN_AMD_DEBUG") ? 1 : 0;
  return v == 1;
}

// Raise a kernel's dynamic-LDS limit once per (kernel, device, size): the attribute call is not legal while a
// stream is being captured into a hipGraph, and the warm-up launch outside the capture has made it.
template <typename Kern>
bool set_lds(Kern kern, size_t bytes) {
  static std::mutex mu;
  static std::unordered_map<uint64_t, size_t> done;     // (kernel, device): the attribute is per device
  const void *fn = reinterpret_cast<const void *>(kern);
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t key = (uint64_t)(uintptr_t)fn * 64u + (uint64_t)dev;
  std::lock_guard<std::mutex> lock(mu);
  auto it = done.find(key);
  if (it != done.end() && it->second >= bytes) return true;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    if (debug_on()) fprintf(stderr, "[ptgnn_amd] stream kernel: %zu B of LDS refused (%s) -> tile kernel\n", bytes, hipGetErrorString(e));
    return false;
  }
  done[key] = bytes;
  return true;
}

// runs per slab for the dense kernels: one 8-wave workgroup per CU
void dense_runs(int nrb, int ncs, int &rps, int &run_len, int max_wg = 0) {
  if (max_wg <= 0) max_wg = num_compute_units();
  rps = max_wg / ncs;
  if (rps < 1) rps = 1;
  if (rps >= kNumXcd && (rps / kNumXcd * kNumXcd) * ncs * 10 >= max_wg * 9) rps = rps / kNumXcd * kNumXcd;
  if (rps > nrb) rps = nrb;
  run_len = (nrb + rps - 1) / rps;
  rps = (nrb + run_len - 1) / run_len;   // drop runs that would be empty
}

}  // namespace

int stream_gemm_mode() {
  int mode = g_mode.load(std::memory_order_relaxed);
  if (mode < 0) {
    const char *e = getenv("PTGNN_AMD_GEMM");
    const int m = e ? atoi(e) : 1;
    mode = (m == 0 || m == 1) ? m : 1;
    g_mode.store(mode, std::memory_order_relaxed);
  }
  return mode;
}

void stream_gemm_set_mode(int mode) { g_mode.store(mode, std::memory_order_relaxed); }

#define PTGNN_STREAM_DISPATCH_NB(NBV, KERN, ...)     \
  switch (NBV) {                                      \
    case 1: KERN(1, __VA_ARGS__); break;              \
    case 2: KERN(2, __VA_ARGS__); break;              \
    case 3: KERN(3, __VA_ARGS__); break;              \
    default: KERN(4, __VA_ARGS__); break;             \
  }

int stream_linear(const float *x, int64_t rows, int32_t k, int64_t ld_x, const float *w, int32_t n_out,
                  const float *bias, int act, float *y, int64_t ld_y, hipStream_t st, const float *addend,
                  int64_t ld_add) {
  const int mode = stream_gemm_mode();
  if (mode == 0) return 0;
  // the add-epilogue lives in the dwordx4 store path only
  if (addend && (ld_y % 4 != 0 || !aligned16(y) || ld_add % 4 != 0 || !aligned16(addend))) return 0;
  if (k % 64 != 0 || n_out % 32 != 0 || ld_x % 4 != 0 || !aligned16(x) || !aligned16(w)) return 0;
  if (rows >= ((int64_t)1 << 31) * 32) return 0;
  int bn = n_out >= 128 ? 128 : n_out;
  // 64-column slabs where the [128, K] slab does not fit LDS but the [64, K] one does (300 < K <= 576: the GRU backward's
  // K = 3 H input-gradient GEMMs): the weights stay RESIDENT and A is read once per 64-column slab, against the panel ring's
  // refill barriers every 64 k.  Measured (scripts/linear_ring_bench.py, profiles/r05_notes.md 7): 116 k x 384 -> 128
  // 127.6 -> 113.0 us (0.57 -> 0.64 of the MFMA peak), 1.25 M x 512 -> 256 2657 -> 2443 us (0.78 -> 0.85); same bits (one K
  // order).  PTGNN_AMD_LINEAR_BN = 128 | 64 forces either form where it fits (A/B).
  {
    const bool fits128 = Slab::bytes(k, 128) + kEpiBytes <= (size_t)kLdsBudget;
    const bool fits64 = Slab::bytes(k, 64) + kEpiBytes <= (size_t)kLdsBudget;
    const char *bn_env = getenv("PTGNN_AMD_LINEAR_BN");
    const int want = bn_env ? atoi(bn_env) : 0;
    const char *ring_force = getenv("PTGNN_AMD_LINEAR_RING");          // "1" forces the ring, whose slabs are 128 columns
    const bool ring_forced = ring_force && ring_force[0] == '1';
    if (!ring_forced && n_out % 128 == 0 && n_out <= 256 && fits64 && (want == 64 || (want != 128 && !fits128))) bn = 64;
  }
  const int nb = bn / 32;
  const char *ring_env = getenv("PTGNN_AMD_LINEAR_RING");                     // "1": force (A/B, parity tests at small K)
  const bool force_ring = ring_env && ring_env[0] == '1', no_ring = ring_env && ring_env[0] == '0';
  const char *force = getenv("PTGNN_AMD_FORCE_STREAM");        // tests: replay small reference fixtures on these kernels
  const bool forced = force && force[0] == '1';
  const bool ring_shape = n_out % 128 == 0 && (int64_t)n_out * k < ((int64_t)1 << 30) && (rows >= 32 * 64 || forced);
  const bool ring = ring_shape && (force_ring || (!no_ring && Slab::bytes(k, bn) + kEpiBytes > (size_t)kLdsBudget));
  if (!ring) {
    // measured (profiles/r02_notes.md): the persistent kernel pays one slab copy per workgroup and re-reads
    // A once per column slab, so it only beats the tile kernel with >= 3 units per wave and few slabs
    const int64_t units = (rows + 31) / 32 * ((n_out + bn - 1) / bn);
    if (!forced && (units < (int64_t)num_compute_units() * 8 * 3 || (n_out + bn - 1) / bn > 4)) return 0;
  }
  const size_t slab = Slab::bytes(k, bn);
  const size_t lds = slab + 16 + 8 * kTqFloats * sizeof(float);
  LinearArgs p;
  p.x = x; p.rows = rows; p.K = k; p.ld_x = ld_x; p.w = w; p.n_out = n_out; p.bias = bias; p.act = act;
  p.y = y; p.ld_y = ld_y; p.addend = addend; p.ld_add = ld_add;
  p.nrb = (int)((rows + 31) / 32);
  p.ncs = (n_out + bn - 1) / bn;
  p.vec_store = (ld_y % 4 == 0 && aligned16(y)) ? 1 : 0;
  if (ring) {
    // the slab does not fit: stream the weights through the panel ring
    const size_t rlds = (size_t)(2 * kLinRingPanelFloats + kRingWaves * kTqFloats) * sizeof(float);
    dense_runs(p.nrb, p.ncs, p.rps, p.run_len, 2 * num_compute_units());   // two 4-wave workgroups per CU
    p.lds_floats = 0;
    auto kern = k_stream_linear_ring;
    if (!set_lds(kern, rlds)) return 0;
    kern<<<(unsigned)(p.ncs * p.rps), kRingWaves * 64, rlds, st>>>(p);
    count_launch(PTGNN_AMD_KERNEL_STREAM_LINEAR_RING);
    return 1;
  }
  if (lds > (size_t)kLdsBudget) return 0;
  dense_runs(p.nrb, p.ncs, p.rps, p.run_len);
  p.lds_floats = (int)(slab / 4);
  const unsigned grid = (unsigned)(p.ncs * p.rps);
#define PTGNN_K(NBV, UNUSED)                                                        \
  do {                                                                              \
    auto kern = k_stream_linear<NBV>;                                               \
    if (!set_lds(kern, lds)) return 0;                                              \
    kern<<<grid, 512, lds, st>>>(p);                                                \
  } while (0)
  PTGNN_STREAM_DISPATCH_NB(nb, PTGNN_K, 0);
#undef PTGNN_K
  count_launch(PTGNN_AMD_KERNEL_STREAM_LINEAR);
  return 1;
}

int stream_gru(const float *a, int64_t ld_a, const float *h, int64_t ld_h, const float *w_ih,
               const float *w_hh, const float *b_ih, const float *b_hh, int64_t n, int32_t m, int32_t hd,
               float *out, int64_t ld_out, float *gates, hipStream_t st) {
  const int mode = stream_gemm_mode();
  if (mode == 0) return 0;
  if (m % 64 != 0 || hd % 64 != 0 || ld_a % 4 != 0 || ld_h % 4 != 0 || ld_out % 4 != 0 || !aligned16(a) ||
      !aligned16(h) || !aligned16(out) || (gates && !aligned16(gates)) || !aligned16(w_ih) || !aligned16(w_hh))
    return 0;
  if (n >= ((int64_t)1 << 31) * 32) return 0;
  const int K = m + hd;
  const size_t slab = Slab::bytes(K, 96);
  const size_t lds = slab + 16 + 8 * kTqFloats * sizeof(float);
  GruArgs p;
  p.a = a; p.ld_a = ld_a; p.h = h; p.ld_h = ld_h; p.w_ih = w_ih; p.w_hh = w_hh; p.b_ih = b_ih; p.b_hh = b_hh;
  p.n = n; p.M = m; p.H = hd; p.out = out; p.ld_out = ld_out; p.gates = gates;
  p.nrb = (int)((n + 31) / 32);
  p.ncs = hd / 32;
  const char *ring_env = getenv("PTGNN_AMD_GRU_RING");                        // "1": A/B + parity tests at small K
  const bool force_ring = ring_env && ring_env[0] == '1';
  if ((lds > (size_t)kLdsBudget && !(ring_env && ring_env[0] == '0')) || force_ring) {
    // the slab does not fit (K > ~400): stream the weights through the panel ring
    const size_t rlds = (size_t)(2 * kRingPanelFloats + kRingWaves * kTqFloats) * sizeof(float);
    dense_runs(p.nrb, p.ncs, p.rps, p.run_len, 2 * num_compute_units());   // two 4-wave workgroups per CU
    p.lds_floats = 0;
    auto kern = k_stream_gru_ring;
    if (!set_lds(kern, rlds)) return 0;
    kern<<<(unsigned)(p.ncs * p.rps), kRingWaves * 64, rlds, st>>>(p);
    count_launch(PTGNN_AMD_KERNEL_STREAM_GRU_RING);
    return 1;
  }
  if (lds > (size_t)kLdsBudget) return 0;
  dense_runs(p.nrb, p.ncs, p.rps, p.run_len);
  p.lds_floats = (int)(slab / 4);
  const unsigned grid = (unsigned)(p.ncs * p.rps);
  {
    auto kern = k_stream_gru;
    if (!set_lds(kern, lds)) return 0;
    kern<<<grid, 512, lds, st>>>(p);
  }
  count_launch(PTGNN_AMD_KERNEL_STREAM_GRU);
  return 1;
}

// The dropout forms take: exact fp32, no activation, no target-state half, K in {64, 128, 256}, message width 64 or 128.
static bool edge_v2_shape(int32_t state_dim, int32_t msg_dim, int use_dst, int act) {
  if (stream_gemm_mode() != 1 || act != PTGNN_AMD_ACT_NONE || use_dst) return false;
  const int K = state_dim;
  if (!(K == 64 || K == 128 || K == 256)) return false;
  if (!(msg_dim == 64 || msg_dim == 128)) return false;
  return Slab::bytes(K, msg_dim) + kEpiBytes <= (size_t)kLdsBudget;
}

template <int DROP>
static int edge_v2_launch(const EdgeV2Args &q, int K, int msg_dim, unsigned grid, size_t lds, hipStream_t st) {
#define PTGNN_K(NBV, NCHV)                                            \
  do {                                                                \
    auto kern = k_stream_edge_v2<NBV, NCHV, DROP, 0>;                 \
    if (!set_lds(kern, lds)) return 0;                                \
    kern<<<grid, 512, lds, st>>>(q);                                  \
    count_launch(PTGNN_AMD_KERNEL_STREAM_EDGE_V2);                    \
    return 1;                                                         \
  } while (0)
  if (msg_dim == 128) {
    if (K == 64) PTGNN_K(4, 2);
    if (K == 128) PTGNN_K(4, 4);
    if (K == 256) PTGNN_K(4, 8);
  } else {
    if (K == 64) PTGNN_K(2, 2);
    if (K == 128) PTGNN_K(2, 4);
    if (K == 256) PTGNN_K(2, 8);
  }
#undef PTGNN_K
  return 0;
}

// 0 = not taken, 1 = streaming
constexpr size_t kEdgeEpiBytes = 16 + (size_t)kEdgeWaves * kTqFloats * sizeof(float);   // = kEpiBytes at 8 waves

static int edge_plan(int32_t state_dim, int32_t msg_dim, int use_dst, size_t *slab_bytes) {
  const int mode = stream_gemm_mode();
  if (mode == 0) return 0;
  const int K = use_dst ? 2 * state_dim : state_dim;
  if (K % 64 != 0 || state_dim % 32 != 0 || msg_dim % 32 != 0 || msg_dim > 128) return 0;
  // 12 % ahead of the tile kernel at K = 128 and 9 % at K = 256 (cfg3's last layer, 133 KB slab)
  if (Slab::bytes(K, msg_dim) + kEdgeEpiBytes <= (size_t)kLdsBudget) {
    *slab_bytes = Slab::bytes(K, msg_dim);
    return 1;
  }
  return 0;
}

int stream_edge_supported(int32_t state_dim, int32_t msg_dim, int use_dst) {
  size_t b;
  return edge_plan(state_dim, msg_dim, use_dst, &b) != 0;
}

int stream_edge_masked_supported(int32_t state_dim, int32_t msg_dim) {
  return edge_v2_shape(state_dim, msg_dim, 0, PTGNN_AMD_ACT_NONE) ? 1 : 0;
}

int stream_edge(const StreamEdgeTable &tab, const float *x, int64_t ld_x, int64_t num_rows, int32_t state_dim,
                int use_dst, int32_t msg_dim, int act, float *msg, int64_t ld_msg, int64_t msg_row_base,
                hipStream_t st, const StreamEdgeMask *mask) {
  size_t slab = 0;
  const int kind = edge_plan(state_dim, msg_dim, use_dst, &slab);
  const bool v2 = edge_v2_shape(state_dim, msg_dim, use_dst, act);
  if (mask && mask->mode != 0 && !v2) return 0;        // the dropout forms exist in k_stream_edge_v2 only
  if (kind == 0) return 0;
  const int nb = msg_dim / 32;
  const size_t lds = slab + 16 + kEdgeWaves * kTqFloats * sizeof(float);
  const int total = tab.unit_off[tab.num_types];
  if (total == 0) return 1;
  EdgeArgs p;
  p.tab = tab; p.tab_dev = nullptr; p.x = x; p.ld_x = ld_x; p.H = state_dim; p.use_dst = use_dst; p.M = msg_dim; p.act = act;
  p.msg = msg; p.ld_msg = ld_msg; p.msg_row_base = msg_row_base; p.num_rows = num_rows;
  // apportion the workgroups (one per CU) to the edge types: proportional start, then hand the spare ones to /
  // take the excess from the type whose load per workgroup moves the maximum least
  int budget = num_compute_units();
  if (budget > total / 8 + 1) budget = total / 8 + 1;      // at least ~one unit per wave
  int w[kStreamMaxTypes], sum = 0, nonempty = 0;
  for (int t = 0; t < tab.num_types; ++t) {
    const int64_t u_t = tab.unit_off[t + 1] - tab.unit_off[t];
    w[t] = u_t == 0 ? 0 : (int)(u_t * budget / total);
    if (u_t > 0 && w[t] == 0) w[t] = 1;
    sum += w[t];
    nonempty += u_t > 0;
  }
  if (budget < nonempty) budget = nonempty;
  while (sum != budget) {
    int best = -1;
    double best_v = 0.0;
    for (int t = 0; t < tab.num_types; ++t) {
      const double u_t = (double)(tab.unit_off[t + 1] - tab.unit_off[t]);
      if (u_t == 0) continue;
      if (sum < budget) {                       // give to the most loaded
        const double v = u_t / w[t];
        if (best < 0 || v > best_v) { best = t; best_v = v; }
      } else if (w[t] > 1) {                    // take where the load after the cut stays smallest
        const double v = u_t / (w[t] - 1);
        if (best < 0 || v < best_v) { best = t; best_v = v; }
      }
    }
    if (best < 0) break;
    w[best] += sum < budget ? 1 : -1;
    sum += sum < budget ? 1 : -1;
  }
  p.tab.wg_off[0] = 0;
  for (int t = 0; t < tab.num_types; ++t) p.tab.wg_off[t + 1] = p.tab.wg_off[t] + w[t];
  for (int t = tab.num_types + 1; t <= kStreamMaxTypes; ++t) {   // the kernels search the prefixes with a ballot over all
    p.tab.wg_off[t] = p.tab.wg_off[tab.num_types];               // 65 entries: those past the last type hold the totals
    p.tab.unit_off[t] = p.tab.unit_off[tab.num_types];
  }
  p.run_len = 0;
  p.lds_floats = (int)(slab / 4);
  const unsigned grid = (unsigned)p.tab.wg_off[tab.num_types];
  if (mask && mask->mode != 0) {
    EdgeV2Args q;
    q.e = p;
    q.mask = mask->bits; q.mask_ld = mask->ld; q.mask_col0 = mask->col0; q.scale = mask->scale;
    return mask->mode == 1 ? edge_v2_launch<1>(q, state_dim, msg_dim, grid, lds, st)
                           : edge_v2_launch<2>(q, state_dim, msg_dim, grid, lds, st);
  }
#define PTGNN_K(NBV, UNUSED)                                  \
  do {                                                        \
    auto kern = k_stream_edge<NBV, false>;                    \
    if (!set_lds(kern, lds)) return 0;                        \
    kern<<<grid, kEdgeWaves * 64, lds, st>>>(p);              \
  } while (0)
  PTGNN_STREAM_DISPATCH_NB(nb, PTGNN_K, 0);
#undef PTGNN_K
  count_launch(PTGNN_AMD_KERNEL_STREAM_EDGE);
  return 1;
}

// The grouped per-edge GEMM over a table that lives in device memory (no target-state half; the arithmetic mode the
// per-edge launch of the same shape would use).  The launch
// is one workgroup per CU; workgroups beyond the table's apportioning leave at once.
int stream_edge_indirect(const StreamEdgeTable *tab_dev, const float *const *w_per_type, int num_types, const float *x,
                         int64_t ld_x, int64_t num_rows, int32_t state_dim, int32_t msg_dim, int act, float *msg,
                         int64_t ld_msg, hipStream_t st) {
  if (num_types > kStreamMaxTypes) return 0;
  size_t slab = 0;
  const int kind = edge_plan(state_dim, msg_dim, 0, &slab);      // the arithmetic the per-edge launch would use
  if (kind == 0) return 0;
  const int nb = msg_dim / 32;
  const size_t lds = slab + 16 + kEdgeWaves * kTqFloats * sizeof(float);
  EdgeArgs p;
  p.tab.num_types = num_types;
  for (int t = 0; t < num_types; ++t) p.tab.w[t] = w_per_type[t];
  p.tab_dev = tab_dev;
  p.x = x; p.ld_x = ld_x; p.H = state_dim; p.use_dst = 0; p.M = msg_dim; p.act = act;
  p.msg = msg; p.ld_msg = ld_msg; p.msg_row_base = 0; p.num_rows = num_rows;
  p.run_len = 0;
  p.lds_floats = (int)(slab / 4);
  const unsigned grid = (unsigned)edge_table_budget();
#define PTGNN_K(NBV, UNUSED)                                  \
  do {                                                        \
    auto kern = k_stream_edge<NBV, true>;                     \
    if (!set_lds(kern, lds)) return 0;                        \
    kern<<<grid, kEdgeWaves * 64, lds, st>>>(p);              \
  } while (0)
  PTGNN_STREAM_DISPATCH_NB(nb, PTGNN_K, 0);
#undef PTGNN_K
  count_launch(PTGNN_AMD_KERNEL_STREAM_EDGE_SHARED);
  return 1;
}

int edge_table_budget() {
  const int cus = num_compute_units();
  return cus > kStreamMaxTypes ? cus : kStreamMaxTypes;
}

}  // namespace ptgnn_amd

extern "C" int ptgnn_amd_set_gemm_mode(int mode) {
  if (mode != 0 && mode != 1) {
    ptgnn_amd::set_error("set_gemm_mode: mode must be 0 (tile kernels) or 1 (streaming kernels); the 3 x bf16 split mode "
                         "of rounds 2-4 was removed");
    return PTGNN_AMD_EINVAL;
  }
  ptgnn_amd::stream_gemm_set_mode(mode);
  return PTGNN_AMD_OK;
}

extern "C" int ptgnn_amd_get_gemm_mode(void) { return ptgnn_amd::stream_gemm_mode(); }
