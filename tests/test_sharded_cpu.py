"""world_size-2 (and 3) gloo tests of the dst-range sharding host logic (ptgnn_amd/sharded.py):
range partitioning, halo id exchange, source remapping and the per-layer all-to-all of halo rows.
The aggregation itself is checked with the CPU oracle as the *checker* (the product kernels are
GPU-only and covered by the -m gpu tests); what must hold is that the sharded result equals the
unsharded one bit for bit, because every destination row is reduced on one rank in the same order."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _global_graph(n, counts, seed):
    g = torch.Generator().manual_seed(seed)
    adj = []
    for c in counts:
        adj.append((torch.randint(0, n, (c,), generator=g), torch.randint(0, n, (c,), generator=g)))
    x = torch.randn(n, 12, generator=g)
    return adj, x


def _worker(rank, world, port, n, counts, seed, reduce, out_q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import scatter_ref
        from ptgnn_amd import sharded
        adj, x = _global_graph(n, counts, seed)
        indeg = torch.zeros(n, dtype=torch.int64)
        for _, d in adj:
            indeg += torch.bincount(d, minlength=n)
        ranges = sharded.balanced_node_ranges(indeg, world)
        lo, hi = ranges[rank]
        mine = [(s[(d >= lo) & (d < hi)], d[(d >= lo) & (d < hi)]) for s, d in adj]
        shard = sharded.ShardedGraph.build(mine, (lo, hi), build_plan=False)
        assert shard.n_local == hi - lo and shard.plan is None
        # 1) the exchanged table holds exactly the rows of the global matrix it claims to hold
        table = shard.exchange(x[lo:hi].contiguous())
        want_rows = torch.cat([torch.arange(lo, hi), shard.need_ids])
        np.testing.assert_array_equal(table.numpy(), x[want_rows].numpy())
        assert sum(shard.recv_splits) == shard.n_halo and shard.need_ids.numel() == shard.n_halo
        # 2) aggregation over the remapped local adjacency == the global aggregation's own rows
        msgs = torch.cat([table.index_select(0, s) for s, _ in shard.local_adj])
        tgt = torch.cat([d for _, d in shard.local_adj])
        got = scatter_ref.scatter(msgs, tgt, dim=0, dim_size=shard.n_local, reduce=reduce)
        gm = torch.cat([x.index_select(0, s) for s, _ in adj])
        gt = torch.cat([d for _, d in adj])
        want = scatter_ref.scatter(gm, gt, dim=0, dim_size=n, reduce=reduce)[lo:hi]
        np.testing.assert_array_equal(got.numpy(), want.numpy())
        # 3) every edge is owned by exactly one rank
        e_local = torch.tensor([sum(int(s.shape[0]) for s, _ in mine)])
        dist.all_reduce(e_local)
        assert int(e_local) == sum(counts)
        out_q.put((rank, "ok", (lo, hi), shard.n_halo))
    except Exception as e:  # surface the failure in the parent
        import traceback
        out_q.put((rank, "fail", traceback.format_exc(), repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,reduce", [(2, "sum"), (2, "max"), (3, "mean")])
def test_sharded_matches_unsharded(world, reduce):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 300, [1500, 0, 700], 3, reduce, q))
             for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for r in results:
        assert r[1] == "ok", r[2]
    ranges = sorted(r[2] for r in results)
    assert ranges[0][0] == 0 and ranges[-1][1] == 300
    assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    assert all(r[3] > 0 for r in results)   # a random graph always has cut edges


def _worker_nocut(rank, world, port, out_q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ptgnn_amd import sharded
        st = sharded.make_weak_scaling_shard(500, 3000, 8, rank, world, "cpu", seed=5, cut_edges=False)
        sh = sharded.ShardedGraph.build(st["adj_global"], st["range"], build_plan=False,
                                        all_ranges=st["all_ranges"])
        assert sh.no_cut and sh.n_halo == 0 and sum(sh.send_splits) == 0
        lo = st["range"][0]
        assert torch.equal(sh.local_adj[0][0], st["adj_global"][0][0] - lo)
        st2 = sharded.make_weak_scaling_shard(500, 3000, 8, rank, world, "cpu", seed=5, cut_edges=True)
        sh2 = sharded.ShardedGraph.build(st2["adj_global"], st2["range"], build_plan=False)
        assert not sh2.no_cut and sh2.n_halo > 0 and sum(sh2.send_splits) > 0
        out_q.put((rank, "ok"))
    except Exception:
        import traceback
        out_q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_disjoint_union_partition_needs_no_exchange():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_nocut, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for r in results:
        assert r[1] == "ok", r[1]


def test_balanced_ranges_follow_edge_mass():
    from ptgnn_amd import sharded
    deg = torch.zeros(1000, dtype=torch.int64)
    deg[:10] = 10_000                      # hubs at the front
    r = sharded.balanced_node_ranges(deg, 4)
    assert r[0][0] == 0 and r[-1][1] == 1000 and all(a[1] == b[0] for a, b in zip(r, r[1:]))
    assert r[0][1] - r[0][0] < 10          # the first rank gets only a few hub rows
    w = [float((deg[a:b] + 1).sum()) for a, b in r]
    assert max(w) / (sum(w) / 4) < 1.6


def _worker_exchange_grad(rank, world, port, out_q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ptgnn_amd import sharded
        n = 240
        adj, x = _global_graph(n, [900, 300], 11)
        ranges = [(p * n // world, (p + 1) * n // world) for p in range(world)]
        lo, hi = ranges[rank]
        mine = [(s[(d >= lo) & (d < hi)], d[(d >= lo) & (d < hi)]) for s, d in adj]
        shard = sharded.ShardedGraph.build(mine, (lo, hi), build_plan=False, all_ranges=ranges)
        coef = torch.linspace(0.5, 2.0, n).unsqueeze(1) * torch.ones(1, x.shape[1])
        xl = x[lo:hi].clone().requires_grad_(True)
        table = shard.exchange_autograd(xl)
        rows = torch.cat([torch.arange(lo, hi), shard.need_ids])
        ((rank + 1.0) * table * coef[rows]).sum().backward()
        # expected: own use + every peer that lists the row among its halo rows
        weight = torch.full((n,), 0.0)
        weight[lo:hi] += rank + 1.0
        for p, (plo, phi) in enumerate(ranges):
            if p == rank:
                continue
            srcs = torch.cat([s[(d >= plo) & (d < phi)] for s, d in adj])
            need = torch.unique(srcs[(srcs < plo) | (srcs >= phi)])
            need = need[(need >= lo) & (need < hi)]
            weight[need] += p + 1.0
        want = (weight.unsqueeze(1) * coef)[lo:hi]
        np.testing.assert_allclose(xl.grad.numpy(), want.numpy(), rtol=1e-6, atol=1e-6)
        out_q.put((rank, "ok"))
    except Exception:
        import traceback
        out_q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_halo_exchange_backward_is_the_transposed_exchange(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_exchange_grad, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for r in results:
        assert r[1] == "ok", r[1]


# ------------------------------------------------------------------------------------------------
# round 2: the SAME `sharded.run_stack` / `ShardedGraph.build` / `combine_graph_pools` code the GPU path runs,
# driven over gloo with oracle-backed stand-in layers (the product kernels are GPU-only), on a cfg4-shaped stack
# ------------------------------------------------------------------------------------------------
class _OracleShardLayer:
    """Stand-in for a ptgnn_amd message-passing layer: real halo exchange, CPU oracle arithmetic on the local
    table [own | halo] (rows past n_local are halo rows without in-edges and are dropped)."""

    def __init__(self, spec):
        self.spec = spec

    def forward_sharded(self, x_local, shard):
        from oracle import mp_oracle as O
        table = shard.exchange(x_local.contiguous())
        feats = [torch.empty(a[0].shape[0], 0) for a in shard.local_adj]
        fn = O.mlp_mp_layer if self.spec["kind"] == "mlp" else O.ggnn_layer
        return fn(table, shard.local_adj, feats, self.spec)[: shard.n_local]


class _OracleGlobalExchange:
    """Stand-in for GruGlobalStateUpdate.forward_sharded: per-rank partial pools (oracle scatter) combined by the
    product's `combine_graph_pools`, then the oracle GRU cell."""

    def __init__(self, spec):
        self.spec = spec

    def forward_sharded(self, x_local, shard):
        from oracle import mp_oracle as O
        from oracle.scatter_ref import scatter
        from ptgnn_amd import sharded
        idx, G, w = shard.node_to_graph_idx, shard.num_graphs, self.spec
        kind = w["pool"]
        if kind == "weighted_sum":
            weights = torch.sigmoid(O.linear(x_local, w["pool_w"]).squeeze(-1))
            local, kind = scatter(x_local * weights.unsqueeze(-1), idx, dim=0, dim_size=G, reduce="sum"), "sum"
        else:
            local = scatter(x_local, idx, dim=0, dim_size=G, reduce="sum" if kind == "mean" else kind)
        pooled = sharded.combine_graph_pools(local, torch.bincount(idx, minlength=G)[:G], kind, shard.group)
        return O.gru_cell(pooled[idx], x_local, w["w_ih"], w["w_hh"], w["b_ih"], w["b_hh"])


def _cfg4_like(seed):
    """VarMisuse-shaped problem at toy size: 6 graphs, T0 = 10 -> T = 21 edge types, hidden 16; the MLP stack of
    varmisuse/train.py:42-74 (8 MLP-MP layers + residuals) and a GGNN + global-exchange tail (:76-107)."""
    from oracle import mp_oracle as O
    from ptgnn_amd import layers as L, workloads
    H = 16
    mb = workloads.batched_graphs(6, 150, 10, 2.4, refs_per_graph=3, seed=seed)
    n = mb["num_nodes"]
    adj = O.augment_adjacency(mb["adjacency_lists"], n, True, True)
    T = len(adj)
    torch.manual_seed(seed)
    mk = lambda: L.MlpMessagePassingLayer(H, H, H, T, "max").export_weights()            # noqa: E731
    mk2 = lambda: L.MlpMessagePassingLayer(2 * H, H, 2 * H, T, "max").export_weights()    # noqa: E731
    ggnn = L.GatedMessagePassingLayer(H, H, T, "sum").export_weights()
    gru = torch.nn.GRUCell(H, H)
    glob = {"kind": "global_gru", "pool": "weighted_sum", "pool_w": torch.randn(1, H) * 0.3,
            "w_ih": gru.weight_ih.detach(), "w_hh": gru.weight_hh.detach(), "b_ih": gru.bias_ih.detach(),
            "b_hh": gru.bias_hh.detach()}
    glob_max = dict(glob, pool="max")
    specs = [{"kind": "residual_origin", "name": "r1"}, mk(), mk(), mk(), {"kind": "residual_concat", "name": "r1"},
             mk2(), {"kind": "residual_origin", "name": "r2"}, mk(), mk(), {"kind": "residual_mean", "name": "r2"},
             {"kind": "residual_origin", "name": "r3"}, mk(), {"kind": "residual_concat", "name": "r3"}, mk2(),
             ggnn, glob, ggnn, glob_max]
    x = workloads.node_states(n, H, seed=seed + 1)
    return mb, adj, specs, x, H


def _worker_run_stack(rank, world, port, out_q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import mp_oracle as O
        from ptgnn_amd import layers as L, sharded
        mb, adj, specs, x, H = _cfg4_like(61)
        n, n2g = mb["num_nodes"], mb["node_to_graph_idx"]
        want = O.run_layer_stack(x, adj, specs, node_to_graph_idx=n2g)
        indeg = torch.zeros(n, dtype=torch.int64)
        for _, d in adj:
            indeg += torch.bincount(d, minlength=n)
        ranges = sharded.balanced_node_ranges(indeg, world)          # edge-mass cuts: graphs straddle ranks
        lo, hi = ranges[rank]
        assert int(n2g[lo]) == int(n2g[lo - 1]) if rank > 0 else True
        mine = [(s[(d >= lo) & (d < hi)], d[(d >= lo) & (d < hi)]) for s, d in adj]
        shard = sharded.ShardedGraph.build(mine, (lo, hi), build_plan=False)
        shard.attach_graph_index(n2g[lo:hi].contiguous(), mb["num_graphs"])
        # product residual layers (pure node-local glue) + oracle stand-ins for the kernel-backed layers
        mods, res = [], {}
        for sp in specs:
            k = sp["kind"]
            if k in ("mlp", "ggnn"):
                mods.append(_OracleShardLayer(sp))
            elif k == "global_gru":
                mods.append(_OracleGlobalExchange(sp))
            elif k == "residual_origin":
                mods.append(("origin", sp["name"]))
            else:
                dim = H
                r = (L.ConcatResidualLayer if k == "residual_concat" else L.MeanResidualLayer)(dim)
                i = next(j for j, m in enumerate(mods) if isinstance(m, tuple) and m[1] == sp["name"])
                mods[i] = r.pass_through_dummy_layer()
                mods.append(r)
        got = sharded.run_stack(mods, x[lo:hi].contiguous(), shard)
        err = float((got - want[lo:hi]).abs().max())
        assert err <= 1e-5, f"rank {rank}: run_stack vs unsharded oracle: {err:.3e}"
        out_q.put((rank, "ok", shard.n_halo, err))
    except Exception:
        import traceback
        out_q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_run_stack_cfg4_shape_with_global_exchange_over_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_run_stack, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for r in results:
        assert r[1] == "ok", r[1]
    assert all(r[2] > 0 for r in results)       # the cuts go through graphs: every rank exchanges halo rows


def _worker_pools(rank, world, port, out_q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.scatter_ref import scatter
        from ptgnn_amd import sharded
        g = torch.Generator().manual_seed(71)
        n, G, D = 90, 7, 5
        idx, _ = torch.sort(torch.randint(0, G - 1, (n,), generator=g))     # graph G-1 is empty everywhere
        x = torch.randn(n, D, generator=g)
        lo, hi = rank * n // world, (rank + 1) * n // world
        for kind in ("sum", "mean", "max", "min"):
            local = scatter(x[lo:hi], idx[lo:hi], dim=0, dim_size=G, reduce="sum" if kind == "mean" else kind)
            cnt = torch.bincount(idx[lo:hi], minlength=G)
            got = sharded.combine_graph_pools(local, cnt, kind)
            want = scatter(x, idx, dim=0, dim_size=G, reduce=kind)
            np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-6, atol=1e-6)
            assert float(got[G - 1].abs().max()) == 0.0
        # gradient of the differentiable all-reduce: loss = sum over ranks of <c_r, allreduce(x)>
        xl = x[lo:hi].sum(0, keepdim=True).clone().requires_grad_(True)
        y = sharded.all_reduce(xl, "sum")
        (float(rank + 1) * y).sum().backward()
        np.testing.assert_allclose(xl.grad.numpy(), np.full((1, D), sum(range(1, world + 1)), dtype=np.float32))
        out_q.put((rank, "ok"))
    except Exception:
        import traceback
        out_q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_partial_graph_pools_combine_across_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_pools, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for r in results:
        assert r[1] == "ok", r[1]


def _worker_two_blocks(rank, world, port, out_q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import scatter_ref
        from ptgnn_amd import sharded
        n = 300
        adj, x = _global_graph(n, [1500, 0, 700], 13)
        ranges = [(p * n // world, (p + 1) * n // world) for p in range(world)]
        lo, hi = ranges[rank]
        mine = [(s[(d >= lo) & (d < hi)], d[(d >= lo) & (d < hi)]) for s, d in adj]
        shard = sharded.ShardedGraph.build(mine, (lo, hi), build_plan=False, all_ranges=ranges, overlap=True)
        nl = shard.n_local
        assert shard.overlap and shard.plan is None          # CPU tensors: plans are the GPU path's business
        # the two blocks partition this rank's edges by where the source lives, per type, order preserved
        for (ls, ld), (so, do), (sh_, dh) in zip(shard.local_adj, shard.adj_own, shard.adj_halo):
            assert bool((so < nl).all()) and bool((sh_ >= nl).all())
            own_mask = ls < nl
            assert torch.equal(so, ls[own_mask]) and torch.equal(do, ld[own_mask])
            assert torch.equal(sh_, ls[~own_mask]) and torch.equal(dh, ld[~own_mask])
        # async exchange: the own block is aggregated before wait(), the halo block after it
        table = shard.new_table(x.shape[1], x)
        table[:nl] = x[lo:hi]
        work = shard.begin_exchange(table)
        def agg(block, reduce):
            msgs = torch.cat([table.index_select(0, s) for s, _ in block])
            tgt = torch.cat([d for _, d in block])
            return scatter_ref.scatter(msgs, tgt, dim=0, dim_size=nl, reduce=reduce), torch.bincount(tgt, minlength=nl)
        own_sum, deg_o = agg(shard.adj_own, "sum")
        own_max, _ = agg(shard.adj_own, "max")
        work.wait()
        halo_sum, deg_h = agg(shard.adj_halo, "sum")
        halo_max, _ = agg(shard.adj_halo, "max")
        # combine through the product's 2-slot plan (rowptr / col), evaluated with the oracle scatter
        plan_rowptr = lambda deg: torch.cat([torch.zeros(1, dtype=torch.int32), torch.cumsum(deg, 0).to(torch.int32)])
        comb = shard._combine_plan(plan_rowptr(deg_o), plan_rowptr(deg_h))
        E = int(comb.rowptr[-1])
        slot_row = torch.repeat_interleave(torch.arange(nl), (comb.rowptr[1:] - comb.rowptr[:-1]).to(torch.int64))
        gm = torch.cat([x.index_select(0, s) for s, _ in adj])
        gt = torch.cat([d for _, d in adj])
        for reduce, parts in (("sum", (own_sum, halo_sum)), ("max", (own_max, halo_max))):
            stacked = torch.cat(parts)
            got = scatter_ref.scatter(stacked[comb.col[:E].to(torch.int64)], slot_row, dim=0, dim_size=nl, reduce=reduce)
            want = scatter_ref.scatter(gm, gt, dim=0, dim_size=n, reduce=reduce)[lo:hi]
            if reduce == "max":
                np.testing.assert_array_equal(got.numpy(), want.numpy())
            else:
                np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-6, atol=1e-6)
        out_q.put((rank, "ok", shard.n_halo))
    except Exception:
        import traceback
        out_q.put((rank, traceback.format_exc(), 0))
    finally:
        dist.destroy_process_group()


def test_two_block_split_async_exchange_and_combine_plan_over_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_two_blocks, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for r in results:
        assert r[1] == "ok", r[1]
    assert all(r[2] > 0 for r in results)


# ------------------------------------------------------------------------------------------------
# graph-boundary partitions and the group-wide form decision
# ------------------------------------------------------------------------------------------------
def test_ranges_on_graph_boundaries_snap_to_graph_starts_and_cut_no_edge():
    """cuts on graph boundaries (graphneuralnetwork.py:418-423 keeps a graph's ids contiguous) => a
    disjoint-union batch has no cut edge; the cfg4-shaped batch is balanced within one graph's mass."""
    from ptgnn_amd import sharded, workloads
    mb = workloads.batched_graphs(40, 2000, 10, 2.4, seed=21)          # the cfg4 batch of bench.py
    n, n2g = mb["num_nodes"], mb["node_to_graph_idx"]
    adj = list(mb["adjacency_lists"])
    adj = adj + [(d, s) for s, d in adj] + [(torch.arange(n), torch.arange(n))]
    indeg = torch.zeros(n, dtype=torch.int64)
    for _, d in adj:
        indeg += torch.bincount(d, minlength=n)
    for world in (2, 3, 4, 8):
        ranges = sharded.ranges_on_graph_boundaries(n2g, indeg, world)
        assert ranges[0][0] == 0 and ranges[-1][1] == n and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        for lo, hi in ranges[1:]:
            assert lo == n or n2g[lo] != n2g[lo - 1], "a cut inside a graph"
        for lo, hi in ranges:                                           # no edge crosses a rank
            for s, d in adj:
                m = (d >= lo) & (d < hi)
                assert bool(((s[m] >= lo) & (s[m] < hi)).all())
        mass = [float((indeg[lo:hi] + 1).sum()) for lo, hi in ranges]
        biggest_graph = max(float((indeg[n2g == g] + 1).sum()) for g in range(int(n2g.max()) + 1))
        assert max(mass) - min(mass) <= 2 * biggest_graph
    # fewer graphs than ranks: surplus ranks own nothing, nothing breaks
    few = sharded.ranges_on_graph_boundaries(torch.tensor([0, 0, 1, 1, 1]), torch.ones(5, dtype=torch.int64), 4)
    assert few[0][0] == 0 and few[-1][1] == 5 and all(a[1] == b[0] for a, b in zip(few, few[1:]))
    with pytest.raises(ValueError):
        sharded.ranges_on_graph_boundaries(torch.tensor([1, 0]), torch.ones(2, dtype=torch.int64), 2)


def _worker_form(rank, world, port, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ptgnn_amd import layers as L, sharded
        # unbalanced shards around the edge-path threshold: rank 0 owns few nodes with many in-edges, rank 1 many
        # nodes with few -- a per-rank decision (the round-2 code) lands on different sides of E * 1.25 < N * T
        n, T, H, M = 4000, 4, 64, 16                                    # T * M == H: the payload width cannot tell
        ranges = [(0, 400), (400, n)]
        lo, hi = ranges[rank]
        g = torch.Generator().manual_seed(7 + rank)
        per_type = 8000 if rank == 0 else 300
        adj = [(torch.randint(0, n, (per_type,), generator=g), torch.randint(lo, hi, (per_type,), generator=g))
               for _ in range(T)]
        shard = sharded.ShardedGraph.build(adj, (lo, hi), all_ranges=ranges, build_plan=False)
        own_view = L._prefer_edge_path(shard.num_edges, shard.n_local + shard.n_halo, T, H, M)
        group_view = L._prefer_edge_path(*shard.form_sizes(True), T, H, M)
        # a snapped partition: no collective, no read-back, local ids
        adj2 = [(torch.randint(lo, hi, (50,), generator=g), torch.randint(lo, hi, (50,), generator=g))]
        sh2 = sharded.ShardedGraph.build(adj2, (lo, hi), all_ranges=ranges, build_plan=False, assume_no_cut=True)
        ok2 = (sh2.no_cut and sh2.n_halo == 0 and int(sh2.local_adj[0][0].min()) >= 0
               and int(sh2.local_adj[0][0].max()) < hi - lo and sh2.exchange(torch.ones(hi - lo, 3)).shape[0] == hi - lo)
        out_q.put((rank, own_view, group_view, shard.global_stats, ok2))
    finally:
        dist.destroy_process_group()


def test_layer_form_is_one_decision_for_the_group():
    """edge form vs table form fixes what the halo all-to-all carries, so ranks must agree."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_form, args=(r, 2, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p_ in procs:
        p_.join(timeout=60)
    assert res[0][1] != res[1][1], "the test shards are meant to disagree when each rank decides alone"
    assert res[0][2] == res[1][2], "group-wide sizes give one decision"
    assert res[0][3] == res[1][3] and res[0][3][0] == 4 * 8000 + 4 * 300 and res[0][3][1] == 4000
    assert res[0][4] and res[1][4]


# ------------------------------------------------------------------------------------------------
# round 6: learned exchange capacities -- no blocking host read in a steady-state cut-edge build
# ------------------------------------------------------------------------------------------------
def _planner_minibatch(world, n, e, seed, spread):
    """One random graph over `world` equal ranges; `spread` = the fraction of sources drawn from ALL nodes (the others
    stay inside the destination's range): how many halo rows a minibatch needs varies with it."""
    g = torch.Generator().manual_seed(seed)
    per = n // world
    dst = torch.randint(0, n, (e,), generator=g)
    own = dst // per * per + torch.randint(0, per, (e,), generator=g)
    anywhere = torch.randint(0, n, (e,), generator=g)
    src = torch.where(torch.rand(e, generator=g) < spread, anywhere, own)
    half = e // 2
    return [(src[:half], dst[:half]), (src[half:], dst[half:])], torch.randn(n, 6, generator=g)


def _worker_planner(rank, world, port, out_q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import scatter_ref
        from ptgnn_amd import sharded
        from ptgnn_amd._lib import PtgnnAmdError
        n = 120 * world
        per = n // world
        lo, hi = rank * per, (rank + 1) * per
        ranges = [(p * per, (p + 1) * per) for p in range(world)]
        planner = sharded.ExchangePlanner(slack=1.25, granule=8)
        log = []

        def run(seed, spread, e=900):
            adj, x = _planner_minibatch(world, n, e, seed, spread)
            mine = [(s[(d >= lo) & (d < hi)], d[(d >= lo) & (d < hi)]) for s, d in adj]
            before = dict(sharded.HOST_READS)
            shard = sharded.ShardedGraph.build(mine, (lo, hi), build_plan=False, all_ranges=ranges, planner=planner)
            reads = {k: sharded.HOST_READS[k] - before[k] for k in before}
            table = shard.exchange(x[lo:hi].contiguous())
            assert table.shape[0] == shard.n_local + shard.n_halo == per + sum(shard.recv_splits)
            msgs = torch.cat([table.index_select(0, s) for s, _ in shard.local_adj])
            tgt = torch.cat([d for _, d in shard.local_adj])
            got = scatter_ref.scatter(msgs, tgt, dim=0, dim_size=per, reduce="sum")
            gm = torch.cat([x.index_select(0, s) for s, _ in adj])
            gt = torch.cat([d for _, d in adj])
            want = scatter_ref.scatter(gm, gt, dim=0, dim_size=n, reduce="sum")[lo:hi]
            np.testing.assert_array_equal(got.numpy(), want.numpy())       # same rows, same fold order: same bits
            log.append((reads, shard.n_halo))
            return shard

        run(1, 0.5)                                    # first build: exact (one blocking read), capacities learned
        assert log[-1][0] == {"blocking": 1, "late": 0} and planner.exact_builds == 1 and planner.ready()
        caps0 = list(planner.recv_caps)
        assert all(c % 8 == 0 for c in caps0) and caps0[rank] == 0 and sum(caps0) > 0
        for seed in (2, 3, 4):                         # steady state: smaller or equal demand -> NO blocking read
            sh = run(seed, 0.4)
            assert log[-1][0]["blocking"] == 0 and log[-1][1] == sum(planner.recv_caps), log[-1]
            assert sh.recv_splits == planner.recv_caps and not sh.no_cut
        assert planner.exact_builds == 1 and planner.builds == 4 and sharded.HOST_READS["late"] >= 2
        # peers agree on every pair's capacity without ever having talked about it: send_caps[p -> q] == recv_caps[q <- p]
        mine_caps = torch.tensor(planner.recv_caps + planner.send_caps)
        allc = [torch.empty_like(mine_caps) for _ in range(world)]
        dist.all_gather(allc, mine_caps)
        for p in range(world):
            for q in range(world):
                assert int(allc[p][q]) == int(allc[q][world + p]), (p, q)
        # demand beyond the learned capacity: this build still makes no blocking read (its halo is truncated) ...
        adj, x = _planner_minibatch(world, n, 4000, 9, 1.0)
        mine = [(s[(d >= lo) & (d < hi)], d[(d >= lo) & (d < hi)]) for s, d in adj]
        b0 = sharded.HOST_READS["blocking"]
        sharded.ShardedGraph.build(mine, (lo, hi), build_plan=False, all_ranges=ranges, planner=planner)
        assert sharded.HOST_READS["blocking"] == b0
        # ... and the NEXT build raises on every rank (the overflow count is all-reduced), capacities grown
        try:
            run(10, 0.4)
            raise AssertionError("the overflow of the previous minibatch went unnoticed")
        except PtgnnAmdError as exc:
            assert "learned capacity" in str(exc) and planner.last_overflow > 0 and planner.overflows == 1
        assert sum(planner.recv_caps) > sum(caps0)
        run(9, 1.0, e=4000)                            # the minibatch that overflowed now fits: exact results, no blocking read
        assert log[-1][0]["blocking"] == 0
        run(11, 0.4)
        out_q.put((rank, "ok"))
    except Exception:
        import traceback
        out_q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_planner_builds_make_no_blocking_host_read_in_steady_state(world):
    """`ShardedGraph.build(..., planner=ExchangePlanner())` -- after the first (exact) build the halo
    exchange runs on learned per-pair capacities: a cut-edge minibatch is built, exchanged and aggregated with ZERO
    blocking device -> host reads (sharded.HOST_READS), bit-identical to the global aggregation; both ends of every peer
    pair hold the same capacity; a minibatch over capacity is reported by the next build on EVERY rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_planner, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for r in results:
        assert r[1] == "ok", r[1]
