// EXPERIMENT, not built into libptgnn_amd.so (round 2; results in profiles/r02_notes.md).
//
// "Walk" variant of the gather / segment-reduce main kernel: a lane group takes R consecutive destination rows =
// ONE contiguous CSR slot range, reads `col` two batches of U slots ahead and the message rows one batch ahead
// of the fold, parks finished rows in registers.  Bit-identical to k_gather_reduce (2 800 cases on the GPU), but
// SLOWER on every BASELINE shape except the max reduce of the cfg5 shard: the double buffers + parked rows cost
// 100-190 VGPRs (3-4 waves per SIMD against 8), and this kernel's throughput is set by bytes in flight per CU.
//   cfg3 segment max: 97 us (one row per lane group) -> 102 (R2 U4) / 109 (R4 U4) / 109 (R4 U8)
//   cfg2 table sum:   103 -> 119 / 152 / 190;    cfg5 shard sum: 4.19 ms -> 4.36 / 5.04 / 6.22
// What helped instead (shipped): reduce_pf<8> in gather_reduce.hip -- 8 slots per round trip, next group's col
// entries prefetched, 63-70 VGPRs.  This file drops into gather_reduce.hip behind k_gather_reduce (it uses RowOp
// with the reset() / store_from() / finish<PRE>() helpers of that experiment) and is kept for the record only.

// ------------------------------------------------------------------------------------------------
// walk kernel: R consecutive rows per lane group = one contiguous range of CSR slots
// ------------------------------------------------------------------------------------------------
// R consecutive destination rows own ONE contiguous slot range, so a lane group that takes R rows walks a
// slot stream whose addresses do not depend on the row boundaries: `col` is read two batches of U slots
// ahead and the message rows one batch ahead of the fold, and the rowptr -> col -> row chain of dependent
// round trips is paid once per R rows instead of once per row (and once per 4 slots of a long row: a
// 4000-edge row of a power-law graph was ~1000 serial col -> row round trips on one lane group).  Slots are
// folded in CSR order, so results are bit-identical to k_gather_reduce.  Finished rows are parked in
// registers and stored after the walk, LayerNorm gamma / beta are preloaded: the loop issues nothing but the
// prefetches, in one fixed order (cols of batch j+2, rows of batch j+1, fold batch j).
template <int LPR, int CH, int REDUCE, bool HAS_DST, bool HAS_ARG, int R, int U>
__global__ __launch_bounds__(256) void k_gather_reduce_walk(Args a) {
  using Op = RowOp<4, LPR, CH, REDUCE, HAS_DST, HAS_ARG, false>;
  constexpr int G = 256 / LPR;
  constexpr int UD = HAS_DST ? U : 1;
  constexpr int RA = HAS_ARG ? R : 1;
  const int64_t tile = xcd_swizzle(blockIdx.x, gridDim.x);
  if (tile >= a.num_tiles) return;
  const int64_t row0 = (tile * G + threadIdx.x / LPR) * R;
  if (row0 >= a.num_nodes) return;   // whole lane group exits together
  const int nrows = a.num_nodes - row0 < R ? (int)(a.num_nodes - row0) : R;
  const int g = threadIdx.x % LPR;

  int bnd[R + 1];   // slot boundaries of the R rows (rows beyond the matrix are empty)
#pragma unroll
  for (int k = 0; k <= R; ++k) bnd[k] = a.rowptr[row0 + (k < nrows ? k : nrows)];
  if (a.hub_threshold > 0) {
    bool hub = false;
#pragma unroll
    for (int k = 0; k < R; ++k) hub = hub || (bnd[k + 1] - bnd[k] > a.hub_threshold);
    if (hub) {   // rare: the chunk kernel owns the hub rows, the others take the per-row path
#pragma unroll 1
      for (int k = 0; k < nrows; ++k) {
        const int b = a.rowptr[row0 + k], e = a.rowptr[row0 + k + 1];
        if (e - b > a.hub_threshold) continue;
        Op o(a, g, 0);
        o.reduce(row0 + k, b, e, 1);
        o.finish_and_store(row0 + k, e - b);
      }
      return;
    }
  }

  Op op(a, g, 0);
  float gw[CH][4], gb[CH][4];
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int colx = (g + c * LPR) * 4 + v;
      const bool on = (a.epi & PTGNN_AMD_EPI_LAYERNORM) && colx < op.M;
      gw[c][v] = on ? a.ln_gamma[colx] : 0.f;
      gb[c][v] = on ? a.ln_beta[colx] : 0.f;
    }
  float res[R][CH][4] = {};
  int rarg[RA][CH][4] = {};

  const int s0 = bnd[0], s1 = bnd[R];
  const int last = s1 > 0 ? s1 - 1 : 0;    // the host launches this kernel only for plans with >= 1 slot
  auto pick = [&](int j) {                 // bnd[min(j, R)] for a lane-group-uniform j >= 1
    int v = bnd[R];
#pragma unroll
    for (int r = R - 1; r >= 1; --r) v = j == r ? bnd[r] : v;
    return v;
  };
  int k = 0, cur_beg = s0, cur_end = bnd[1];   // fold cursor: row k owns slots [cur_beg, cur_end)
  int kl = 0, ld_end = bnd[1];                 // load cursor (destination term only)

  auto row_done = [&]() __attribute__((always_inline)) {   // row k is complete: epilogue, park it, next row
    op.template finish<true>(cur_end - cur_beg, gw, gb);
#pragma unroll
    for (int r = 0; r < R; ++r) {   // selects, not `if (r == k)`: hipcc turns that into a scratch array indexed by k
      const bool here = r == k;
#pragma unroll
      for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          res[r][c][v] = here ? op.acc[c][v] : res[r][c][v];
          if constexpr (HAS_ARG) rarg[r][c][v] = here ? op.arg[c][v] : rarg[r][c][v];
        }
    }
    op.reset();
    ++k;
    cur_beg = cur_end;
    cur_end = pick(k + 1);
  };
  auto load_cols = [&](int32_t (&pk)[U], int sb) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int s = sb + u;
      pk[u] = a.col[s < last ? s : last];
    }
  };
  auto issue_rows = [&](float (&m)[U][CH][4], float (&d)[UD][CH][4], const int32_t (&pk)[U], int sb)
                        __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t src = pk[u] >> a.type_bits;
      const int t = pk[u] & op.tmask;
      op.load_row(a.ysrc + src * a.ld_y + (int64_t)t * op.M, m[u]);
      if constexpr (HAS_DST) {
        int s = sb + u;
        s = s < last ? s : last;
        while (s >= ld_end && kl < nrows - 1) {   // s < s1 = bnd[nrows] unless the stream is empty
          ++kl;
          ld_end = pick(kl + 1);
        }
        op.load_row(a.ydst + (row0 + kl) * a.ld_yd + (int64_t)t * op.M, d[u]);
      }
    }
  };
  auto fold_one = [&](float (&m)[CH][4], float (&d)[CH][4], int slot) __attribute__((always_inline)) {
    if constexpr (HAS_DST) {
#pragma unroll
      for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int v = 0; v < 4; ++v) m[c][v] += d[c][v];
    }
    op.fold(m, slot);
  };
  auto fold_batch = [&](float (&m)[U][CH][4], float (&d)[UD][CH][4], int sb) __attribute__((always_inline)) {
    if (cur_end - sb > U) {   // the whole batch lies inside row k (cur_end <= s1, so all U slots exist)
#pragma unroll
      for (int u = 0; u < U; ++u) fold_one(m[u], d[HAS_DST ? u : 0], sb + u);
      return;
    }
    int nb = s1 - sb;
    nb = nb < 0 ? 0 : (nb > U ? U : nb);
    int u0 = 0;
    for (;;) {
      int lim = cur_end - sb;   // slots [u0, lim) of this batch belong to row k
      lim = lim < nb ? lim : nb;
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (u >= u0 && u < lim) fold_one(m[u], d[HAS_DST ? u : 0], sb + u);
      if (cur_end - sb > nb) break;   // row k continues in the next batch
      row_done();
      u0 = lim;
      if (k >= nrows) break;
    }
  };

  {
    int32_t pkA[U], pkB[U];
    float mA[U][CH][4], mB[U][CH][4];
    float dA[UD][CH][4], dB[UD][CH][4];
    load_cols(pkA, s0);
    if (s0 + U < s1) load_cols(pkB, s0 + U);
    issue_rows(mA, dA, pkA, s0);
    int sb = s0;
    for (;;) {
      if (sb + U < s1) {
        if (sb + 2 * U < s1) load_cols(pkA, sb + 2 * U);
        issue_rows(mB, dB, pkB, sb + U);
      }
      fold_batch(mA, dA, sb);
      if (k >= nrows) break;
      sb += U;
      if (sb + U < s1) {
        if (sb + 2 * U < s1) load_cols(pkB, sb + 2 * U);
        issue_rows(mA, dA, pkA, sb + U);
      }
      fold_batch(mB, dB, sb);
      if (k >= nrows) break;
      sb += U;
    }
  }

#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (r < nrows)
      op.store_from(res[r], rarg[HAS_ARG ? r : 0], a.out + (row0 + r) * a.ld_out,
                    HAS_ARG ? a.argout + (row0 + r) * (int64_t)op.M : nullptr);
  }
}

