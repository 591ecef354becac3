"""The `cpu_baseline` leg: the CPU restatement of the reference path (oracle/mp_oracle.py; kind "port" -- the reference's
own modules live in /root/reference, which does not exist on the GPU box) timed on this box's host cores, and the
full-size parity of the GPU output against it.  The ONLY bench module besides the per-config parity checks that imports
`oracle`; nothing here runs inside a timed GPU region."""
import os
import time

import torch

from benchmarks.common import PARITY_TOL

CPU_FORWARD_BUDGET_S = 6.0    # per thread count: a warm-up slower than this is reported as is (no timed repeats)


def cpu_model() -> str:
    """CPU model string of the host the baseline ran on ("CPU model string")."""
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or "unknown"


def _timed_forwards(fn, n_timed=3, budget=None):
    """1 warm-up + n timed forwards, median (SURVEY.md 8d / BASELINE.md 3).  Time-boxed: when the warm-up alone
    exceeds the budget (e.g. 256 threads on a cgroup-limited host: 105 s per forward) its time is the figure and
    the repeats are skipped, so the default bench run stays within minutes.  Returns (seconds, output, n_timed)."""
    t0 = time.perf_counter()
    out = fn()
    warm = time.perf_counter() - t0
    if warm > (CPU_FORWARD_BUDGET_S if budget is None else budget):
        return warm, out, 0
    ts = []
    for _ in range(n_timed):
        t0 = time.perf_counter()
        out = fn()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2], out, n_timed


def _thread_counts():
    """{1, 8, 32, all}, `all` capped at the cores this process may actually use."""
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    try:   # cgroup v2 CPU quota of the container, if any
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            usable = max(1, min(usable, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return sorted({c for c in (1, 8, 32, usable) if c <= usable})


def _sweep(fn):
    """seconds per forward by thread count.  Counts are tried in increasing order; once doubling-plus the threads
    no longer buys 20 % (the oracle is bandwidth / framework bound well before 256 threads) larger counts are
    skipped -- oversubscribed runs cost minutes and are never the baseline."""
    sweep, out, prev = {}, None, None
    for c in _thread_counts():
        if prev is not None and c > 32 and sweep[str(prev[0])] > 0.8 * prev[1]:
            sweep[str(c)] = None
            continue
        torch.set_num_threads(c)
        sec, out, _ = _timed_forwards(fn)
        if sweep:
            prev = (c, min(v for v in sweep.values() if v is not None))
        sweep[str(c)] = round(sec, 4)
    return sweep, out


def cpu_baseline_cfg2(st, gpu_out):
    """The CPU restatement of the reference path (kind "port": the reference's own modules live in
    /root/reference, which does not exist on the GPU box) on this box's host cores, on the SAME config-2 inputs:
    thread sweep {1, 8, 32, all}, 1 warm-up + 3 timed forwards each, median; parity of the GPU output against
    the oracle output at full size."""
    from oracle import mp_oracle as O
    spec = st["layer"].export_weights()
    x, adj = st["cpu_x"], st["cpu_adj"]
    feats = [torch.empty(st["E"], 0)]
    with torch.no_grad():
        sweep, want = _sweep(lambda: O.mlp_mp_layer(x, adj, feats, spec))
    best = min((k for k in sweep if sweep[k] is not None), key=lambda k: sweep[k])
    parity = {"max_abs": float((gpu_out.cpu() - want).abs().max()), "tol": PARITY_TOL, "n": st["N"],
              "against": "oracle/mp_oracle.py at full size (N=200k, E=1.1M)"}
    return {"value": round(st["E"] / sweep[best], 1), "unit": "edges/s", "cores": int(best), "kind": "port",
            "host_cpus": os.cpu_count(), "cpu_model": cpu_model(), "seconds_by_threads": sweep,
            "value_1_thread": round(st["E"] / sweep["1"], 1),
            "sample": "kind \"port\": the torch-CPU fp32 RESTATEMENT of the reference layer (oracle/mp_oracle.py), not the "
                      "reference's own module (no /root/reference on the GPU box; the two time within 5-25 % of each other in "
                      "the authoring container, BASELINE.md 3). cfg2 full size (N=200k, E=1.1M), 1 MLP-MP layer forward; per "
                      "thread count 1 warm-up + 3 timed, median; best thread count reported in `cores`"}, parity


def cpu_baseline_cfg3(st, gpu_out):
    """CPU restatement (kind "port") of the 8-layer GGNN stack on this box's host cores.
    Thread sweep on a BOUNDED sample (the first 8 of the 48 graphs, same weights: 1 warm-up + 3 timed forwards per
    thread count, median), then ONE forward of the full batch at the best thread count -- timed, and kept as the
    full-size parity reference for the GPU output."""
    from oracle import mp_oracle as O
    from ptgnn_amd import workloads
    small = workloads.batched_graphs(8, 2500, 8, 2.2, seed=1234)
    xs = workloads.node_states(small["num_nodes"], st["H"], seed=5)
    e_small = 2 * sum(int(a[0].shape[0]) for a in small["adjacency_lists"]) + small["num_nodes"]
    with torch.no_grad():
        sweep, _ = _sweep(lambda: O.gnn_forward(xs, small["adjacency_lists"], st["specs"], True, True))
        ranked = sorted((k for k in sweep if sweep[k] is not None), key=lambda k: sweep[k])
        # the FULL batch (the inputs the GPU line is measured on) at the TWO best thread counts of the sample sweep (an
        # 8-graph sample can rank them wrongly for the 48-graph batch: VERDICT r03 weak #14), 1 warm-up + 3 timed
        # forwards each, median; the better one is `value`
        full_by_threads, full, best = {}, None, ranked[0]
        for cand in ranked[:2]:
            torch.set_num_threads(int(cand))
            sec, (want_c, n_edges_c), n_timed_c = _timed_forwards(
                lambda: O.gnn_forward(st["cpu_x"], st["cpu_adj"], st["specs"], True, True), budget=9.0)
            full_by_threads[cand] = round(sec, 3)
            if full is None or sec < full:
                full, best, want, n_edges, n_timed = sec, cand, want_c, n_edges_c, n_timed_c
    layers = st["layers_per_step"]
    parity = {"max_abs": float((gpu_out.cpu() - want).abs().max()), "tol": PARITY_TOL, "n": st["N"],
              "edges_counted_match": bool(n_edges == st["E"]),
              "against": f"oracle/mp_oracle.py at full size (N={st['N']}, E={st['E']}, 8 GGNN layers)"}
    return {"value": round(st["E"] / (full / layers), 1), "unit": "edges/s", "cores": int(best),
            "kind": "port", "host_cpus": os.cpu_count(), "cpu_model": cpu_model(), "seconds_by_threads_8_graph_sample": sweep,
            "full_batch_seconds": round(full, 3), "full_batch_timed_forwards": n_timed,
            "full_batch_seconds_by_threads": full_by_threads,
            "sample_value": round(e_small / (sweep[best] / layers), 1),
            "sample_value_1_thread": round(e_small / (sweep["1"] / layers), 1),
            "sample": "kind \"port\": the torch-CPU fp32 RESTATEMENT of the reference layers (oracle/mp_oracle.py), not the "
                      "reference's own modules -- those cannot be imported on the GPU box (no /root/reference there); in the "
                      "authoring container the two time within 5-25 % of each other (BASELINE.md 3). "
                      f"`value` = E / t_layer of the FULL Graph2Class batch (the GPU line's own inputs: N={st['N']}, "
                      f"E={st['E']}, {layers}-layer GGNN stack forward) at the best thread count (`cores`), 1 warm-up + "
                      "3 timed forwards, median -- also the parity reference; the thread count is the better of the two "
                      f"best of a sweep {{1, 8, 32, all}} on the first 8 of the 48 graphs (N={small['num_nodes']}, E={e_small}; "
                      "`sample_value*`)"}, parity

