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
