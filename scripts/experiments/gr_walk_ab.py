"""Micro-benchmarks (and, in round 2, the A/B harness) of the gather / segment-reduce main kernel on the GPU box:

    python scripts/experiments/gr_walk_ab.py [check] [time] [scale]

The product build has ONE main kernel, so by default every table has one column ("cur").  The round-2 experiment
build read two developer knobs per call -- PTGNN_AMD_GR_WALK = RU (the walk kernel kept in
scripts/experiments/gather_reduce_walk.hip) and PTGNN_AMD_GR_U = 4 | 8 (reduce_pf<U>; U = 8 is what shipped) --
and GR_VARIANTS=old,pf4,pf8,w24,w48 selected the columns; results in profiles/r02_notes.md.

`check`: the walk kernel variants (PTGNN_AMD_GR_WALK = RU) must reproduce the one-row-per-lane-group kernel
bit for bit (same CSR fold order) on a sweep of widths / reduces / destination terms / args / epilogues / hub and
tail cases.  `time`: median HIP-event time per variant on the BASELINE shapes (cfg2, cfg3, cfg4, cfg5 shard), plus
two diagnostics: the cfg3 reduce with an identity `col` (what slot-ordered messages would cost) and the cfg5
shard with a lower hub threshold (how much of its time is long rows folded serially)."""
import os
import statistics
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ptgnn_amd import ops, workloads  # noqa: E402

# variant -> (PTGNN_AMD_GR_WALK, PTGNN_AMD_GR_U)
VARIANTS = {"cur": ("0", "0"), "old": ("0", "0"), "pf4": ("0", "4"), "pf8": ("0", "8"), "w24": ("24", "0"), "w48": ("48", "0")}
CODES = [v for v in os.environ.get("GR_VARIANTS", "cur").split(",")]
DEV = "cuda"


def setcode(c):
    w, u = VARIANTS.get(c, ("0", "0"))
    os.environ["PTGNN_AMD_GR_WALK"] = w
    os.environ["PTGNN_AMD_GR_U"] = u


def timeit(fn, n=15, warm=3):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); e.synchronize()
        ts.append(s.elapsed_time(e))
    return statistics.median(ts)


def graph(N, E, T, seed, skew=None):
    rng = np.random.RandomState(seed)
    adj = []
    for t in range(T):
        e = E // T
        src = rng.randint(0, N, size=e)
        if skew is None:
            dst = rng.randint(0, N, size=e)
        else:   # a few heavy rows
            dst = np.where(rng.rand(e) < skew, rng.randint(0, 3, size=e), rng.randint(0, N, size=e))
        adj.append((torch.from_numpy(src.astype(np.int64)).to(DEV), torch.from_numpy(dst.astype(np.int64)).to(DEV)))
    return adj


def same(a, b):
    if isinstance(a, tuple):
        return all(torch.equal(x, y) for x, y in zip(a, b))
    return torch.equal(a, b)


def check():
    """Identity on the product build: the fused table form (gather + destination term + reduce) must equal, bit for
    bit, the segment reduce of the materialised messages y[src, t] + yd[dst, t] (same adds, same CSR fold order) --
    this crosses the group-of-8 / group-of-4 / load-once-destination-term kernel variants against each other."""
    bad = 0
    n_cases = 0
    shapes = [(1, 1, 1, None), (3, 7, 2, None), (37, 200, 3, None), (5003, 30000, 3, None), (4096, 2000, 1, None),
              (20011, 150000, 5, 0.15), (3000, 40000, 1, 0.5), (50021, 400000, 1, None)]
    for (N, E, T, skew) in shapes:
        adj = graph(N, E, T, seed=N + E, skew=skew)
        plan = ops.build_plan(adj, N)
        Etot = plan.num_edges
        hub = (plan.rowptr[1:] - plan.rowptr[:-1]) > ops.HUB_THRESHOLD   # hub rows fold chunk-wise: not bit-comparable
        for M in (32, 64, 128, 256, 512):
            torch.manual_seed(M + N)
            y = torch.randn(N, T * M, device=DEV)
            yd = torch.randn(N, T * M, device=DEV)
            msgs = torch.cat([y[s][:, t * M:(t + 1) * M] + yd[d][:, t * M:(t + 1) * M] for t, (s, d) in enumerate(adj)])
            plain = torch.cat([y[s][:, t * M:(t + 1) * M] for t, (s, d) in enumerate(adj)])
            for red in ("sum", "mean", "max", "min"):
                for with_dst in (True, False):
                    want = ops.segment_reduce(msgs if with_dst else plain, plan, red)
                    got = ops.gather_reduce(y, plan, M, red, ydst=yd if with_dst else None)
                    n_cases += 1
                    ok = torch.equal(want[~hub], got[~hub])
                    if red in ("max", "min"):
                        ok = ok and torch.equal(want, got)
                    if not ok:
                        bad += 1
                        print(f"MISMATCH N={N} E={Etot} T={T} M={M} {red} dst={with_dst}: "
                              f"max|d|={float((want - got).abs().max()):.3e}", flush=True)
        torch.cuda.synchronize()
        print(f"checked N={N} E={Etot} T={T} skew={skew}: cumulative {n_cases} cases, {bad} mismatches", flush=True)
    return bad


def table(title, fn, nbytes, n=15):
    row = []
    for c in CODES:
        setcode(c)
        ms = timeit(fn, n=n)
        row.append(f"{c}: {ms * 1e3:7.1f} us {nbytes / ms / 1e9:4.2f}")
    setcode(CODES[0])
    print(f"{title:58s} | " + " | ".join(row), flush=True)


def cfg3_adj():
    mb = workloads.batched_graphs(48, 2500, 8, 2.2, seed=1234)
    adj = [(s.to(DEV), d.to(DEV)) for s, d in mb["adjacency_lists"]]
    N = mb["num_nodes"]
    ident = torch.arange(N, device=DEV)
    return adj + [(d, s) for s, d in adj] + [(ident, ident)], N


def scale_probe():
    """Is the cfg3 reduce held back by fixed costs (ramp, tail, launch)?  The same batch at 1x / 4x, and the
    box's plain copy rate at the same byte count."""
    for g in (48, 192):
        mb = workloads.batched_graphs(g, 2500, 8, 2.2, seed=1234)
        adj = [(s.to(DEV), d.to(DEV)) for s, d in mb["adjacency_lists"]]
        N = mb["num_nodes"]
        ident = torch.arange(N, device=DEV)
        adj = adj + [(d, s) for s, d in adj] + [(ident, ident)]
        plan = ops.build_plan(adj, N)
        E, M = plan.num_edges, 128
        msg = torch.randn(E, M, device=DEV)
        nb = E * (4.0 * M + 4) + N * (4.0 * M + 4)
        table(f"{g} graphs: segment max N={N} E={E} M=128", lambda: ops.segment_reduce(msg, plan, "max"), nb)
        out = torch.empty_like(msg)
        table(f"   torch copy_ of the same [E,128] matrix (2x bytes)", lambda: out.copy_(msg), 2.0 * E * M * 4)
        half = msg[: E // 2]
        table(f"   torch sum over dim 0 (read-only stream)", lambda: torch.sum(msg, dim=0), 1.0 * E * M * 4)
        del msg, out


def times():
    # cfg3: segment reduce of the [E, 128] message matrix (max), 17 edge types
    adj, N = cfg3_adj()
    plan = ops.build_plan(adj, N)
    E, M = plan.num_edges, 128
    msg = torch.randn(E, M, device=DEV)
    nb = E * (4.0 * M + 4) + N * (4.0 * M + 4)
    table(f"cfg3 segment max N={N} E={E} M=128 (perm)", lambda: ops.segment_reduce(msg, plan, "max"), nb)
    ident = torch.arange(E, device=DEV, dtype=torch.int32)
    table("cfg3 segment max, identity col (slot-ordered messages)",
          lambda: ops.gather_reduce(msg, plan, M, "max", type_bits=0, col=ident), nb)
    table("cfg3 segment max + arg (training forward)", lambda: ops.segment_reduce(msg, plan, "max", return_arg=True),
          nb + N * 4.0 * M)
    msg256 = torch.randn(E, 64, device=DEV)
    table("cfg3 batch, M=64 segment max (README arch)", lambda: ops.segment_reduce(msg256, plan, "max"),
          E * (4.0 * 64 + 4) + N * (4.0 * 64 + 4))
    del msg, msg256
    # cfg2: table form with destination term, sum, GELU + LayerNorm epilogue
    N2, E2 = 200_000, 1_100_000
    a2 = workloads.random_graph(N2, E2, seed=1234)
    a2 = [(a2[0][0].to(DEV), a2[0][1].to(DEV))]
    p2 = ops.build_plan(a2, N2)
    y, yd = torch.randn(N2, M, device=DEV), torch.randn(N2, M, device=DEV)
    g, b = torch.ones(M, device=DEV), torch.zeros(M, device=DEV)
    nb2 = E2 * (4.0 * M + 4) + N2 * (4.0 * M + 4) + N2 * 4.0 * M
    table("cfg2 table+dst sum gelu+ln N=200k E=1.1M M=128",
          lambda: ops.gather_reduce(y, p2, M, "sum", ydst=yd, epilogue=ops.EPI_GELU | ops.EPI_LAYERNORM, ln_weight=g, ln_bias=b), nb2)
    table("cfg2 table sum (no dst, no epilogue)", lambda: ops.gather_reduce(y, p2, M, "sum"), nb2 - N2 * 4.0 * M)
    del y, yd
    # cfg5 per-GPU shard
    N5, E5, M5 = 1_250_000, 12_500_000, 256
    a5 = workloads.power_law_graph(N5, E5, alpha=0.8, seed=1234)
    a5 = [(a5[0][0].to(DEV), a5[0][1].to(DEV))]
    y5 = torch.randn(N5, M5, device=DEV)
    nb5 = E5 * (4.0 * M5 + 4) + N5 * (4.0 * M5 + 4)
    for thr in (4096, 2048):
        ops.HUB_THRESHOLD = thr
        p5 = ops.build_plan(a5, N5)
        table(f"cfg5 shard sum M=256 hub_threshold={thr}", lambda: ops.gather_reduce(y5, p5, M5, "sum"), nb5, n=7)
    ops.HUB_THRESHOLD = 4096
    p5 = ops.build_plan(a5, N5)
    table("cfg5 shard max M=256", lambda: ops.gather_reduce(y5, p5, M5, "max"), nb5, n=7)
    # uniform destinations at the cfg5 size: the same bytes without long rows
    au = workloads.random_graph(N5, E5, seed=7)
    au = [(au[0][0].to(DEV), au[0][1].to(DEV))]
    pu = ops.build_plan(au, N5)
    table("cfg5 size, UNIFORM destinations, sum M=256", lambda: ops.gather_reduce(y5, pu, M5, "sum"), nb5, n=7)


if __name__ == "__main__":
    what = sys.argv[1:] or ["check", "time"]
    rc = 0
    if "check" in what:
        rc = check()
        print("CHECK", "FAILED" if rc else "OK", flush=True)
    if "scale" in what:
        scale_probe()
    if "time" in what:
        times()
    sys.exit(1 if rc else 0)
