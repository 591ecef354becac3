"""BASELINE config 5 per-GPU shard at FULL size (1.25 M rows / 12.5 M in-edges / H = 256, Zipf-0.8 destinations) against
the CPU oracle on EVERY row (rounds 2-4 checked a 4 104-row sample): the chunked oracle of
oracle/fullrow.py (`mp_oracle.layer_on_rows`, <= 2 GB resident) for one GGNN and one MLP-MP layer x {sum, max}, and the
aggregation kernel on its own, bit for bit, against the C restatement of torch_scatter's CPU kernel
(oracle/scatter_ref.c) over the message table the GPU itself produced."""
import subprocess
import os

import pytest
import torch

from conftest import ROOT
from helpers import empty_feats, to_cuda_adj

pytestmark = pytest.mark.gpu
N, E, H = 1_250_000, 12_500_000, 256


@pytest.fixture(scope="module")
def shard():
    from ptgnn_amd import workloads
    adj = workloads.power_law_graph(N, E, alpha=0.8, seed=1234)
    x = workloads.node_states(N, H, seed=2)
    return adj, x


@pytest.mark.parametrize("agg", ["sum", "max"])
@pytest.mark.parametrize("kind", ["ggnn", "mlp"])
def test_config5_shard_full_size_every_row_vs_oracle(shard, kind, agg):
    from oracle import fullrow
    from ptgnn_amd import layers as L, ops
    adj, x = shard
    torch.manual_seed(5)
    layer = (L.GatedMessagePassingLayer(H, H, 1, agg) if kind == "ggnn" else L.MlpMessagePassingLayer(H, H, H, 1, agg)).eval()
    spec = layer.export_weights()
    layer = layer.cuda()
    cadj = to_cuda_adj(adj)
    ops.clear_plan_cache()
    before = ops.launch_counts()
    with torch.no_grad():
        got = layer(x.cuda(), cadj, None, {}, {}, empty_feats(cadj, "cuda")).cpu()
    ran = ops.launches_since(before)
    assert any(k.startswith("k_stream") for k in ran), ran          # the HIP GEMMs produced `got`
    del cadj
    torch.cuda.empty_cache()
    res = fullrow.full_row_parity(spec, adj, x, got)
    print(kind, agg, res)
    assert res["rows_checked"] == N and res["edges_checked"] == E
    assert res["ok"], res
    if agg == "max":
        # max aggregation holds the literal 1e-5 on EVERY one of the 1.25 M rows, hub rows (up to 159 k in-edges) included.
        # sum: rows of < 32 in-edges (1.206 M of them) hold it too (`ok` asserts that); above that an un-normalised
        # sum of 10^2..10^5 messages is not 1e-5-accurate in fp32 for anybody -- measured on the GGNN layer: 2 078 rows over the
        # bar, the fp32 ORACLE 2.4e-3 from float64 on the hub rows, the HIP path 1.5e-4 (16x closer); on the MLP-MP layer 3
        # hub rows, oracle 2.2e-5 / ours 1.4e-6 -- so those rows are attributed against float64, bucket by bucket
        # (oracle/fullrow.py)
        assert res["strict_1e-5"], res
    else:
        assert res["max_abs_rows_below_32_in_edges"] <= 1e-5 and res["rows_below_32_in_edges"] > 1_200_000, res


@pytest.mark.parametrize("reduce", ["max", "sum"])
def test_config5_shard_aggregation_every_row_bit_exact_vs_the_c_restatement(shard, reduce):
    from oracle import fullrow
    from ptgnn_amd import ops, workloads
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    adj, _ = shard
    y = workloads.node_states(N, H, seed=9)
    cadj = to_cuda_adj(adj)
    ops.clear_plan_cache()
    plan = ops.plan_for(cadj, N)
    got = ops.gather_reduce(y.cuda(), plan, H, reduce).cpu()
    del cadj, plan
    ops.clear_plan_cache()
    torch.cuda.empty_cache()
    res = fullrow.segment_reduce_all_rows(lambda s, t: y.index_select(0, s), adj, N, H, reduce, got)
    print(reduce, res)
    assert res["rows_checked"] == N
    if reduce == "max":
        assert res["rows_not_bit_identical"] == 0, res
    else:
        # rows up to the hub threshold fold their in-edges in the reference's order (bit for bit); hub rows fold
        # chunk-wise (1024-slot partials combined in chunk order): same numbers, another association
        assert res["rows_not_bit_identical"] == 0 or res["min_in_degree_of_a_mismatch"] > ops.HUB_THRESHOLD, res
