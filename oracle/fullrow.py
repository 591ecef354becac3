"""TEST INFRASTRUCTURE ONLY -- never imported by the product package `ptgnn_amd`.

Every-row parity of ONE message-passing layer on a graph whose whole-graph CPU evaluation does not fit the host
(BASELINE config 5 per-GPU shard: 1.25 M rows, 12.5 M in-edges, H = 256 -- the reference's per-edge form would gather
a 12.8 GB [E, 256] matrix).  The oracle (`mp_oracle.layer_on_rows`: the reference's arithmetic restricted to a set of
destination rows) walks ALL rows in chunks of bounded edge count (<= 2 GB resident), in fp32 -- the reference's
arithmetic -- and, for the rows where fp32 itself cannot hold 1e-5 (un-normalised sums of >= 32 messages feeding a
GRU), in float64 for attribution.  Used by bench.py (`config5_shard.parity`, rows_checked == N) and by
tests/test_gpu_fullrow.py; both hand in the GPU result, this file only checks it.
"""
import time
from typing import Dict, List, Tuple

import torch

from oracle import mp_oracle as O

TOL = 1e-5


def _dst_sorted(adjacency_lists):
    """Per edge type the edges stably sorted by destination: a row's in-edges stay in their original relative order
    (= the order the reference folds them in), and the in-edges of a row range become one slice."""
    out = []
    for src, dst in adjacency_lists:
        d_sorted, order = torch.sort(dst, stable=True)
        out.append((src[order], d_sorted))
    return out


def full_row_parity(spec: Dict, adjacency_lists, x: torch.Tensor, got: torch.Tensor, max_edges: int = 1 << 20,
                    attribute_from_degree: int = 32, threads: int = 16) -> Dict:
    """got [N, H'] (CPU tensor: the GPU layer's output) against the oracle on EVERY row.

    Bars (the ones the sampled check of rounds 2-4 used, now over all rows):
      * rows with < `attribute_from_degree` in-edges: |got - oracle_fp32| <= 1e-5;
      * all other rows: either <= 1e-5 as well, or no further from a float64 evaluation than 2 x the oracle's own
        fp32 arithmetic is (reported per in-degree bucket).
    """
    prev_threads = torch.get_num_threads()
    torch.set_num_threads(max(1, min(threads, prev_threads if prev_threads > 0 else threads)))
    try:
        return _run(spec, adjacency_lists, x, got, max_edges, attribute_from_degree)
    finally:
        torch.set_num_threads(prev_threads)


def _run(spec, adjacency_lists, x, got, max_edges, attribute_from_degree):
    t0 = time.perf_counter()
    n = x.shape[0]
    deg = torch.zeros(n, dtype=torch.int64)
    for _, dst in adjacency_lists:
        deg += torch.bincount(dst, minlength=n)
    adj = _dst_sorted(adjacency_lists)
    starts = [torch.searchsorted(d, torch.arange(n + 1)) for _, d in adj]      # per type: first edge of every row
    err = torch.empty(n, dtype=torch.float32)                                   # per-row max |got - oracle fp32|
    chunks = 0
    with torch.no_grad():
        for lo, hi in O.row_chunks(deg, max_edges):
            sub = [(s[int(st[lo]): int(st[hi])], d[int(st[lo]): int(st[hi])]) for (s, d), st in zip(adj, starts)]
            want = O.layer_on_rows(x, sub, spec, torch.arange(lo, hi))
            err[lo:hi] = (got[lo:hi] - want).abs().amax(dim=1)
            chunks += 1
    low = deg < attribute_from_degree
    res = {"rows_checked": int(n), "rows_total": int(n), "edges_checked": int(deg.sum()), "chunks": chunks,
           "tol": TOL, "max_abs_all_rows": float(err.max()),
           "max_abs_rows_below_%d_in_edges" % attribute_from_degree: float(err[low].max()) if bool(low.any()) else 0.0,
           "rows_below_%d_in_edges" % attribute_from_degree: int(low.sum()),
           "rows_over_tol": int((err > TOL).sum()),
           "against": "oracle/mp_oracle.py layer_on_rows over ALL rows in chunks of <= %d edges" % max_edges}
    res["strict_1e-5"] = bool(res["max_abs_all_rows"] <= TOL)
    ok = res["max_abs_rows_below_%d_in_edges" % attribute_from_degree] <= TOL
    # float64 attribution of every row that is over the bar (only rows of >= attribute_from_degree in-edges may be)
    over = torch.nonzero(err > TOL).flatten()
    if int(over.numel()):
        ok = ok and bool((deg[over] >= attribute_from_degree).all())
        spec64 = O.cast_spec(spec, torch.float64)
        x64 = x.double()
        buckets: Dict[str, Dict] = {}
        pos, worst_ours, worst_ref = 0, 0.0, 0.0
        csum = torch.cumsum(deg[over], 0)
        with torch.no_grad():
            while pos < int(over.numel()):
                base = int(csum[pos - 1]) if pos else 0
                end = int(torch.searchsorted(csum, torch.tensor(base + max_edges // 2), right=True))
                end = min(int(over.numel()), max(end, pos + 1))
                rows = over[pos:end]
                pick = torch.zeros(n, dtype=torch.bool)
                pick[rows] = True
                sub = []
                for s, d in adj:
                    m = pick[d]
                    sub.append((s[m], d[m]))
                w64 = O.layer_on_rows(x64, sub, spec64, rows)
                w32 = O.layer_on_rows(x, sub, spec, rows)
                ours = (got[rows].double() - w64).abs().amax(dim=1)
                ref = (w32.double() - w64).abs().amax(dim=1)
                for blo, bhi in ((32, 128), (128, 512), (512, 2049), (2049, 1 << 62)):
                    m = (deg[rows] >= blo) & (deg[rows] < bhi)
                    if bool(m.any()):
                        b = buckets.setdefault("in_degree_%d_%s" % (blo, "up" if bhi > 1 << 40 else bhi - 1),
                                               {"rows": 0, "ours_vs_fp64": 0.0, "oracle_fp32_vs_fp64": 0.0})
                        b["rows"] += int(m.sum())
                        b["ours_vs_fp64"] = max(b["ours_vs_fp64"], float(ours[m].max()))
                        b["oracle_fp32_vs_fp64"] = max(b["oracle_fp32_vs_fp64"], float(ref[m].max()))
                worst_ours, worst_ref = max(worst_ours, float(ours.max())), max(worst_ref, float(ref.max()))
                pos = end
        res["float64_attribution"] = {"rows": int(over.numel()), "ours_vs_fp64": worst_ours,
                                      "oracle_fp32_vs_fp64": worst_ref, "buckets": buckets}
        ok = ok and all(b["ours_vs_fp64"] <= max(TOL, 2.0 * b["oracle_fp32_vs_fp64"]) for b in buckets.values())
    res["ok"] = bool(ok)
    res["seconds"] = round(time.perf_counter() - t0, 1)
    return res


def segment_reduce_all_rows(messages_of, adjacency_lists, n: int, dim: int, reduce: str, got: torch.Tensor,
                            max_edges: int = 1 << 21) -> Dict:
    """The aggregation on its own, every row, BIT FOR BIT: `messages_of(src_ids, type)` -> [len, dim] fp32 rows of the
    message table the GPU aggregated (downloaded from the GPU, so both sides fold identical numbers); the C
    restatement of torch_scatter's CPU kernel (oracle/scatter_ref.c: serial, edge order) reduces them chunk by chunk.
    max / min must agree on every row; sum / mean on every row that the GPU folds serially (in-degree <= the hub
    threshold handed in by the caller through `got`'s companion mask -- see tests/test_gpu_fullrow.py)."""
    import ctypes
    import os
    import numpy as np
    so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libscatter_ref.so")
    lib = ctypes.CDLL(so)
    fn = lib.ptgnn_oracle_scatter_f32
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int,
                   ctypes.c_void_p, ctypes.c_void_p]
    code = {"sum": 0, "add": 0, "mean": 1, "max": 2, "min": 3}[reduce]
    deg = torch.zeros(n, dtype=torch.int64)
    for _, dst in adjacency_lists:
        deg += torch.bincount(dst, minlength=n)
    adj = _dst_sorted(adjacency_lists)
    starts = [torch.searchsorted(d, torch.arange(n + 1)) for _, d in adj]
    mismatch = torch.zeros(n, dtype=torch.bool)
    for lo, hi in O.row_chunks(deg, max_edges):
        msgs, tgt = [], []
        for t, ((s, d), st) in enumerate(zip(adj, starts)):
            a, b = int(st[lo]), int(st[hi])
            msgs.append(messages_of(s[a:b], t))
            tgt.append(d[a:b] - lo)
        m = np.ascontiguousarray(torch.cat(msgs).numpy(), dtype=np.float32)
        idx = np.ascontiguousarray(torch.cat(tgt).numpy(), dtype=np.int64)
        out = np.empty((hi - lo, dim), dtype=np.float32)
        rc = fn(m.ctypes.data, idx.ctypes.data, m.shape[0], dim, hi - lo, code, out.ctypes.data, None)
        assert rc == 0, rc
        mismatch[lo:hi] = torch.from_numpy((out != got[lo:hi].numpy()).any(axis=1))
    return {"rows_checked": int(n), "rows_not_bit_identical": int(mismatch.sum()),
            "max_in_degree_of_a_mismatch": int(deg[mismatch].max()) if bool(mismatch.any()) else 0,
            "min_in_degree_of_a_mismatch": int(deg[mismatch].min()) if bool(mismatch.any()) else 0}
