"""Row-range aggregation (ptgnn_amd_gather_reduce_rows_f32) and the aggregation -> GRU pipeline over row ranges
(ops.aggregate_gru): the same kernels on slices of the destination rows, so every result must be BIT-IDENTICAL to the
unsplit launch -- hub rows (chunk-parallel fold + tickets), long rows and the LayerNorm epilogue included."""
import numpy as np
import pytest
import torch

from helpers import empty_feats, to_cuda_adj

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("reduce", ["sum", "max", "mean"])
@pytest.mark.parametrize("graph", ["powerlaw_hubs", "uniform"])
def test_row_range_launches_assemble_the_whole_aggregation_bit_for_bit(graph, reduce):
    from ptgnn_amd import ops, workloads
    N, E, M = 125_000, 1_250_000, 128
    adj = workloads.power_law_graph(N, E, alpha=0.8, seed=7) if graph == "powerlaw_hubs" else workloads.random_graph(N, E, seed=7)
    cadj = to_cuda_adj(adj)
    y = workloads.node_states(N, M, seed=3).cuda()
    ops.clear_plan_cache()
    plan = ops.plan_for(cadj, N)
    if graph == "powerlaw_hubs":
        assert int((plan.rowptr[1:] - plan.rowptr[:-1]).max()) > ops.HUB_THRESHOLD
    whole = ops.gather_reduce(y, plan, M, reduce)
    for bounds in ([0, 64000, N], [0, 31 * 32, 50_016, 50_016, 99_999, N]):
        out = torch.full((N, M), float("nan"), device="cuda")
        for lo, hi in zip(bounds, bounds[1:]):
            ops.gather_reduce(y, plan, M, reduce, out=out, rows=(lo, hi))
        assert torch.equal(out, whole), (graph, reduce, bounds)
    # a range leaves the other rows untouched
    out = torch.full((N, M), 7.0, device="cuda")
    ops.gather_reduce(y, plan, M, reduce, out=out, rows=(1000, 2000))
    assert torch.equal(out[1000:2000], whole[1000:2000])
    assert float(out[:1000].min()) == 7.0 == float(out[2000:].max())
    from ptgnn_amd import PtgnnAmdError
    with pytest.raises(PtgnnAmdError):
        ops.gather_reduce(y, plan, M, reduce, out=out, rows=(10, N + 1))


@pytest.mark.parametrize("form", ["edge", "table"])
@pytest.mark.parametrize("pieces", [2, 3])
def test_pipelined_aggregate_gru_equals_the_unsplit_pair_bit_for_bit(form, pieces, monkeypatch):
    from ptgnn_amd import layers as L, ops, workloads
    mb = workloads.batched_graphs(30, 2500, 4, 2.2, seed=5)
    N, H, T0 = mb["num_nodes"], 128, 4
    adj = mb["adjacency_lists"]
    adj = adj + [(d, s) for s, d in adj] + [(torch.arange(N), torch.arange(N))]
    T = len(adj)
    torch.manual_seed(1)
    layer = L.GatedMessagePassingLayer(H, H, T, "max").cuda().eval()
    x = workloads.node_states(N, H, seed=2).cuda()
    cadj = to_cuda_adj(adj)
    monkeypatch.setattr(L, "EDGE_PATH_BIAS", 1e-9 if form == "table" else 1e9)
    outs = []
    for p in (1, pieces):
        monkeypatch.setattr(ops, "AGG_PIPELINE", p)
        monkeypatch.setattr(ops, "AGG_PIPELINE_MIN_ROWS", 1024)
        ops.clear_plan_cache()
        timer = ops.KernelTimer()
        ops.set_kernel_timer(timer)
        with torch.no_grad():
            for _ in range(3):          # back-to-back layers: the side stream of one call against the next call's main
                y = layer(x, cadj, None, {}, {}, empty_feats(cadj, "cuda"))
        ops.set_kernel_timer(None)
        calls = timer.summary()
        assert calls["gather_reduce"]["calls"] == 3 * p and calls["gru_cell"]["calls"] == 3 * p
        outs.append(y)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])
    assert bool(torch.isfinite(outs[1]).all())


@pytest.mark.parametrize("rows,k,n_out,act,expect", [
    (250_000, 128, 128, None, "k_stream_linear"),        # resident slab (>= 3 units per wave: the streaming dispatch)
    (116_000, 128, 128, None, None),                     # below that the tile kernel runs: GEMM + torch add
    (116_000, 384, 128, None, "k_stream_linear"),        # the GRU backward's d_gh W_hh (K = 3 H): 64-column resident slabs
    (116_000, 768, 128, None, "k_stream_linear_ring"),   # K = 3 x 256 (the Typilus stack's last layer): panel ring
    (116_000, 128, 256, "tanh", "k_stream_linear"),
    (5_000, 100, 36, "relu", None),                      # not a streaming shape: GEMM + torch add
    (33, 64, 32, None, None)])
def test_linear_add_epilogue_equals_the_gemm_followed_by_an_add_bit_for_bit(rows, k, n_out, act, expect):
    from ptgnn_amd import ops
    g = torch.Generator().manual_seed(rows + k)
    x = torch.randn(rows, k, generator=g).cuda()
    w = (torch.randn(n_out, k, generator=g) / k ** 0.5).cuda()
    b = torch.randn(n_out, generator=g).cuda()
    add = torch.randn(rows, n_out, generator=g).cuda()
    for bias in (None, b):
        before = ops.launch_counts()
        got = ops.linear_add(x, w, add, bias, act=act)
        ran = ops.launches_since(before)
        want = ops.linear(x, w, bias, act=act) + add
        assert torch.equal(got, want), (rows, k, n_out, act, bias is not None)
        if expect is not None:
            assert expect in ran, ran
    # strided addend (a column slice) and the ragged last unit
    wide = torch.randn(rows, n_out + 32, generator=g).cuda()
    got = ops.linear_add(x, w, wide[:, 32:], None, act=act)
    assert torch.equal(got, ops.linear(x, w, None, act=act) + wide[:, 32:])
