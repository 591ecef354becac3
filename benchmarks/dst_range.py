"""The dst-range-sharded cut-edge workloads (north star: "destination-node sharding ... RCCL all-to-all of cut-edge
messages over xGMI"): every layer exchanges halo rows over RCCL.  Secondary entries of the bench line; `sharded_cfg5` is
also reported at N = 1 (world-1 process group, nothing cut) so that the N > 1 figures have their single-GPU counterpart."""
import os
import time

import torch

from benchmarks.common import _log, barrier_sync, max_over_ranks, sum_over_ranks
from benchmarks.varmisuse import cfg4_batch, cfg4_modules

# wall-clock budget of the sharded cut-edge variants at N > 1 (env override: the watchdog test uses a short one)
VARIANT_DEADLINE_S = float(os.environ.get("PTGNN_AMD_BENCH_VARIANT_DEADLINE", "240"))


def _clock_collective(fn, k, w, world, dev):
    for _ in range(w):
        fn()
    barrier_sync(world)
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    barrier_sync(world)
    return max_over_ranks(time.perf_counter() - t0, world, dev) / k


def sharded_cfg5(dev, rank, world, k=5):
    """configs[4] as a weak-scaling dst-range shard: rank p owns 1.25M nodes of ONE power-law graph of
    world x 1.25M nodes and the 12.5M in-edges of its nodes (Zipf-0.8 destinations inside the range, sources
    uniform over ALL ranks' nodes => (world-1)/world of the edges are cut); one GGNN layer, H = M = 256, sum.
    Per step: halo bookkeeping + plan build + halo all-to-all (RCCL) + edge-free table form."""
    from ptgnn_amd import layers as L, sharded, workloads
    N, E, H = 1_250_000, 12_500_000, 256
    if (os.environ.get("PTGNN_AMD_BENCH_SHARE_GPU", "0") not in ("", "0")
            and os.environ.get("PTGNN_AMD_BENCH_FULL_VARIANTS", "0") in ("", "0")):
        # validation mode (every rank on cuda:0 over gloo, whose all-to-all stages 1.3 GB per rank through the host):
        # the same code on a tenth of the shard -- the numbers of such a run mean nothing anyway
        N, E = N // 10, E // 10
    lo = rank * N
    adj = workloads.power_law_graph(N, E, alpha=0.8, seed=1234 + rank)
    g = torch.Generator().manual_seed(99 + rank)
    src = torch.randint(0, world * N, (E,), generator=g, dtype=torch.int64)
    state = {"adj_global": [(src.to(dev), (adj[0][1] + lo).to(dev))], "range": (lo, lo + N),
             "x": workloads.node_states(N, H, seed=7 + rank).to(dev),
             "all_ranges": [(p * N, (p + 1) * N) for p in range(world)]}
    torch.manual_seed(5)
    layer = L.GatedMessagePassingLayer(H, H, 1, "sum").to(dev).eval()

    def step():
        from ptgnn_amd import ops
        ops.clear_plan_cache()
        with torch.no_grad():
            return sharded.layer_forward(layer, state)
    _log("  cfg5 shard: inputs built")
    # the single-GPU counterpart, in the same run: the same layer over this rank's rows with every source folded into the
    # rank's own range (nothing is cut, nothing travels) through the ordinary unsharded forward
    own_adj = [((src % N).to(dev), adj[0][1].to(dev))]

    def step_local():
        from ptgnn_amd import ops
        ops.clear_plan_cache()
        with torch.no_grad():
            return layer(state["x"], own_adj, None, {}, {}, [None])
    dt_local = _clock_collective(step_local, k, 2, world, dev)
    del own_adj
    _log(f"  cfg5 shard: {dt_local * 1e3:.2f} ms/step on one GPU's rows without an exchange")
    dt_exact = _clock_collective(step, k, 2, world, dev)
    _log(f"  cfg5 shard: {dt_exact * 1e3:.2f} ms/step (single block, exact build: one host read-back per minibatch)")
    # steady state of a data loader: learned exchange capacities (sharded.ExchangePlanner), no blocking host read per build
    state["planner"] = sharded.ExchangePlanner()
    reads0 = dict(sharded.HOST_READS)
    dt = _clock_collective(step, k, 2, world, dev)
    reads = {key: sharded.HOST_READS[key] - reads0[key] for key in reads0}
    planner = state.pop("planner")
    _log(f"  cfg5 shard: {dt * 1e3:.2f} ms/step (single block, learned capacities; host reads over {k + 2} builds: {reads})")
    state["overlap"] = True      # two-block mode: own-source block aggregated under the halo all-to-all
    dt2 = _clock_collective(step, k, 2, world, dev)
    _log(f"  cfg5 shard: {dt2 * 1e3:.2f} ms/step (two blocks, overlapped)")
    state["overlap"] = False
    shard = sharded.ShardedGraph.build(state["adj_global"], state["range"], all_ranges=state["all_ranges"])
    y = torch.empty(N, H, device=dev)
    t_x = 0.0 if shard.no_cut else _clock_collective(lambda: shard.exchange(y), k, 2, world, dev)
    halo = sum_over_ranks(shard.n_halo, world, dev)
    return {"workload": f"cfg5 shard x{world}: one power-law graph of {world} x {N / 1e6:.3g}M nodes, {E / 1e6:.3g}M in-edges per GPU, "
                        f"sources uniform over all GPUs ({world - 1}/{world} of the edges cut), 1 GGNN layer H=M=256, sum",
            "ms_per_step": round(dt * 1e3, 3), "edges_per_sec_per_layer": round(E * world / dt, 1),
            "ms_per_step_is": "steady state: learned exchange capacities, no blocking host read per minibatch build",
            "ms_per_step_exact_build": round(dt_exact * 1e3, 3),
            "host_reads_in_timed_builds": reads, "exact_builds": planner.exact_builds, "planner_builds": planner.builds,
            "halo_capacity_rows_this_rank": int(sum(planner.recv_caps or [0])),
            "edges_per_gpu": E, "nodes_per_gpu": N,
            "one_gpu_no_exchange": {"ms_per_step": round(dt_local * 1e3, 3), "edges_per_sec_per_layer": round(E / dt_local, 1),
                                    "note": "the same layer and rows with all sources local (max over ranks): the N = 1 "
                                            "counterpart the sharded figure's efficiency is read against"},
            "ms_per_step_two_block_overlap": round(dt2 * 1e3, 3),
            "halo_rows_all_ranks": int(halo), "halo_bytes_per_layer_all_ranks": int(halo) * H * 4,
            "all_to_all_ms": round(t_x * 1e3, 3), "no_cut": bool(shard.no_cut)}



def sharded_cfg4(dev, rank, world, k=5):
    """configs[3]: the VarMisuse batch (40 graphs x ~2000 nodes, T0 = 10 -> T = 21) through the 8-layer MLP-MP
    stack of varmisuse/train.py:42-74 (hidden 64, max) over `world` GPUs, in BOTH partitions SURVEY.md 8e names:
      * "graph_boundaries": cuts snapped to graph starts (sharded.ranges_on_graph_boundaries) -- a disjoint-union
        batch then has no cut edge, so there is no halo exchange and no bookkeeping: the single-GPU stack on the
        rank's own graphs (what ptgnn's batches allow, and the configuration meant to scale);
      * "through_graphs": ranges balanced by in-edge mass alone, cuts go through graphs -- every layer exchanges halo
        rows over one RCCL all-to-all (`forward_sharded`, edge form over the [own | halo] table), plain and
        two-block (overlapped) mode: the worst case for this batch, kept to measure the exchange path."""
    from ptgnn_amd import ops, sharded, workloads
    H = 64
    mb, adj, n, n2g = cfg4_batch()
    indeg = torch.zeros(n, dtype=torch.int64)
    for _, d_ in adj:
        indeg += torch.bincount(d_, minlength=n)
    mods = cfg4_modules(dev)
    x_all = workloads.node_states(n, H, seed=6)
    out = {}
    for name, ranges, no_cut in (("graph_boundaries", sharded.ranges_on_graph_boundaries(n2g, indeg, world), True),
                                 ("through_graphs", sharded.balanced_node_ranges(indeg, world), False)):
        lo, hi = ranges[rank]
        mine = [(s_[(d_ >= lo) & (d_ < hi)].to(dev), d_[(d_ >= lo) & (d_ < hi)].to(dev)) for s_, d_ in adj]
        e_mine = sum(int(a[0].shape[0]) for a in mine)
        x = x_all[lo:hi].contiguous().to(dev)
        n2g_local = n2g[lo:hi].contiguous().to(dev)
        holder = {"planner": None if no_cut else sharded.ExchangePlanner()}

        def step():
            ops.clear_plan_cache()
            with torch.no_grad():
                shard = sharded.ShardedGraph.build(mine, (lo, hi), all_ranges=ranges, overlap=holder.get("overlap", False),
                                                   assume_no_cut=no_cut, planner=holder["planner"])
                shard.attach_graph_index(n2g_local, mb["num_graphs"])
                holder["shard"] = shard
                return sharded.run_stack(mods, x, shard)
        dts = [_clock_collective(step, k, 2, world, dev) for _ in range(3)]   # the first block also pays one-time set-up
        dt = sorted(dts)[1]                                                    # of the collectives: median, like the other lines
        edges = sum_over_ranks(e_mine, world, dev)
        entry = {"ms_per_forward": round(dt * 1e3, 3), "ms_per_forward_is": "median of 3 blocks",
                 "ms_per_forward_blocks": [round(t * 1e3, 3) for t in dts],
                 "edges_per_sec_per_layer": round(edges / (dt / 8), 1),
                 "edges_per_sec_readme_convention": round(edges / dt, 1),
                 "nodes_per_rank_min_max": [int(min(b_ - a_ for a_, b_ in ranges)), int(max(b_ - a_ for a_, b_ in ranges))]}
        if not no_cut:
            entry["build"] = "learned exchange capacities (sharded.ExchangePlanner): no blocking host read per minibatch"
            entry["exact_builds"], entry["planner_builds"] = holder["planner"].exact_builds, holder["planner"].builds
            holder["overlap"] = True
            entry["ms_per_forward_two_block_overlap"] = round(_clock_collective(step, k, 2, world, dev) * 1e3, 3)
            holder["overlap"] = False
            step()
            shard = holder["shard"]
            halo = sum_over_ranks(shard.n_halo, world, dev)
            y = torch.empty(hi - lo, H, device=dev)
            t_x = 0.0 if shard.no_cut else _clock_collective(lambda: shard.exchange(y), k, 2, world, dev)
            entry.update(halo_rows_all_ranks=int(halo), halo_bytes_per_layer_all_ranks=int(halo) * H * 4,
                         all_to_all_ms_per_layer=round(t_x * 1e3, 3), no_cut=bool(shard.no_cut))
        out[name] = entry
    out["workload"] = (f"cfg4 sharded x{world}: VarMisuse batch N={n}, T=21, 8 MLP-MP layers hidden 64 (+ residuals), "
                       "per-minibatch shard build + plan build + 8 layers per forward")
    return out



def run_variants(result, rank, world, dev):
    """Every rank enters the sharded workloads together; neither a failure nor a hang may cost the primary line (an
    exception on one rank leaves its peers inside a collective: every rank therefore carries a watchdog that prints the
    line measured so far and ends the process).  Fills result["cut_edges_variant"], result["sharded_cfg5"] (top level:
    the north-star split's own edges/s beside its one-GPU counterpart) and result["config"]["dst_range_split"]."""
    import json
    import sys
    import threading
    variants = {}
    result["cut_edges_variant"] = variants
    try:   # which ranks the collective backend really spans (a first multi-GPU run should explain itself)
        import torch.distributed as dist
        mine = torch.tensor([rank, torch.cuda.current_device()], dtype=torch.int64, device=dev)
        seen = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(seen, mine)
        result["rccl_ranks_seen"] = [[int(v) for v in t.tolist()] for t in seen]
        result["collective_backend"] = dist.get_backend()
    except Exception as exc:  # noqa: BLE001
        result["rccl_ranks_seen"] = {"error": f"{type(exc).__name__}: {exc}"}

    def bail_out():
        variants.setdefault("error", f"timed out after {VARIANT_DEADLINE_S} s (a rank failed or a collective hung)")
        _log("sharded variants timed out: emitting the primary line")
        if rank == 0:
            from benchmarks.common import emit_line
            emit_line(json.dumps(result))
        os._exit(0)
    watchdog = threading.Timer(VARIANT_DEADLINE_S, bail_out)
    watchdog.daemon = True
    watchdog.start()
    fault = os.environ.get("PTGNN_AMD_BENCH_FAULT", "")      # test hook: "hang:<rank>" stalls that rank here
    if fault == f"hang:{rank}":
        _log(f"fault injection: rank {rank} stalls before the sharded variants")
        threading.Event().wait()
    legs = (("cfg5_shard", sharded_cfg5),) + ((("cfg4_stack", sharded_cfg4),) if world > 1 else ())
    for key, fn in legs:
        try:
            _log(f"sharded cut-edge variant {key}")
            variants[key] = fn(dev, rank, world)
            torch.cuda.empty_cache()
        except Exception as exc:  # noqa: BLE001
            variants[key] = {"error": f"{type(exc).__name__}: {exc}"}
            _log(f"variant {key} failed on rank {rank}: {exc}")
            if world > 1:
                threading.Event().wait()   # peers are inside a collective: let the watchdog end every rank
            break
    watchdog.cancel()
    # the north-star split (dst-range shards + RCCL halo all-to-all) next to the contract's replica line, where a
    # SCALE record reads it: `value` stays the weak-scaling replica run (one batch per GPU, no collective)
    lift = {"collective_backend": result.get("collective_backend"), "rccl_ranks_seen": result.get("rccl_ranks_seen")}
    c5, c4 = variants.get("cfg5_shard"), variants.get("cfg4_stack")
    if isinstance(c5, dict) and "ms_per_step" in c5:
        lift["cfg5_shard"] = {k: c5[k] for k in ("ms_per_step", "ms_per_step_two_block_overlap", "all_to_all_ms",
                                                 "halo_bytes_per_layer_all_ranks", "edges_per_sec_per_layer",
                                                 "no_cut") if k in c5}
        one = c5["one_gpu_no_exchange"]
        result["sharded_cfg5"] = {
            "n_gpus": world, "scaling": "weak", "edges_per_gpu": c5["edges_per_gpu"], "nodes_per_gpu": c5["nodes_per_gpu"],
            "edges_per_sec_per_layer": c5["edges_per_sec_per_layer"], "ms_per_step": c5["ms_per_step"],
            "ms_per_step_two_block_overlap": c5["ms_per_step_two_block_overlap"], "all_to_all_ms": c5["all_to_all_ms"],
            "halo_bytes_per_layer_all_ranks": c5["halo_bytes_per_layer_all_ranks"], "cut_fraction": round((world - 1) / world, 4),
            "one_gpu_edges_per_sec_per_layer": one["edges_per_sec_per_layer"], "one_gpu_ms_per_step": one["ms_per_step"],
            "vs_n_times_one_gpu": round(c5["edges_per_sec_per_layer"] / (world * one["edges_per_sec_per_layer"]), 4),
            "collective_backend": result.get("collective_backend"),
            "workload": c5["workload"]}
    if isinstance(c4, dict):
        for part in ("graph_boundaries", "through_graphs"):
            if isinstance(c4.get(part), dict):
                lift["cfg4_stack_" + part] = {k: c4[part][k] for k in (
                    "ms_per_forward", "ms_per_forward_two_block_overlap", "all_to_all_ms_per_layer",
                    "halo_bytes_per_layer_all_ranks", "edges_per_sec_per_layer", "no_cut") if k in c4[part]}
    if "error" in variants:
        lift["error"] = variants["error"]
    result["config"]["dst_range_split"] = lift
