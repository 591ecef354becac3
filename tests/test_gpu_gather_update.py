"""ptgnn_amd_gather_update_f32: aggregation + GELU + LayerNorm + Linear + activation of the MLP-MP layer in ONE launch
(hidden 64).  Bit-identical to ptgnn_amd_gather_reduce_f32 followed by ptgnn_amd_linear_f32 (the same fold order, the
same K order of the MFMA steps), checked against the oracle at the layer level, and asserted to be the kernel that ran."""
import numpy as np
import pytest
import torch

from helpers import empty_feats, to_cuda_adj

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.mark.parametrize("reduce", ["sum", "mean", "max", "min"])
@pytest.mark.parametrize("out_dim,act,bias", [(64, "tanh", True), (128, None, False), (32, "relu", True), (96, "tanh", False)])
def test_gather_update_equals_gather_reduce_then_linear_bit_for_bit(reduce, out_dim, act, bias):
    """(The kernel's tile product runs on v_mfma_f32_16x16x4; the 32x32x2 and 4x4x1 forms that were built and measured
    beside it -- profiles/r05_notes.md 2 -- passed this same test: one K order, one set of bits.)"""
    from ptgnn_amd import ops
    g = torch.Generator().manual_seed(11)
    N, E, M = 50_019, 270_000, 64                     # ragged last tile; some rows without in-edges
    src = torch.randint(0, N, (E,), generator=g)
    dst = torch.randint(0, N - 500, (E,), generator=g)
    dst[:3000] = 17                                   # one long row (3000 in-edges): folds serially here, chunk-wise there
    cadj = to_cuda_adj([(src, dst)])
    msgs = torch.randn(E, M, generator=g).cuda()
    w = (torch.randn(out_dim, M, generator=g) / 8).cuda()
    b = torch.randn(out_dim, generator=g).cuda() if bias else None
    gamma, beta = (1 + 0.1 * torch.randn(M, generator=g)).cuda(), (0.1 * torch.randn(M, generator=g)).cuda()
    ops.clear_plan_cache()
    plan = ops.plan_for(cadj, N)
    for epi in (ops.EPI_GELU_LAYERNORM, ops.EPI_NONE, ops.EPI_LAYERNORM):
        before = ops.launch_counts()
        got = ops.gather_update(msgs, plan, reduce, plan.perm, 0, epi, gamma, beta, 1e-5, w, b, act)
        assert ops.launches_since(before).get("k_gather_update") == 1
        agg = ops.gather_reduce(msgs, plan, M, reduce, epilogue=epi, ln_weight=gamma, ln_bias=beta, type_bits=0, col=plan.perm)
        want = ops.linear(agg, w, b, act=act)
        rows = torch.ones(N, dtype=torch.bool, device="cuda")
        if reduce in ("sum", "mean"):
            rows[17] = False                          # the hub row: another fp32 association (serial vs chunk partials)
            assert float((got[17] - want[17]).abs().max()) <= 1e-5 * max(1.0, float(want[17].abs().max()))
        assert torch.equal(got[rows], want[rows]), (reduce, out_dim, act, epi)
    # a strided destination (the right half of a concat residual's buffer)
    buf = torch.zeros(N, 2 * out_dim, device="cuda")
    ops.gather_update(msgs, plan, reduce, plan.perm, 0, ops.EPI_GELU_LAYERNORM, gamma, beta, 1e-5, w, b, act, out=buf[:, out_dim:])
    agg = ops.gather_reduce(msgs, plan, M, reduce, epilogue=ops.EPI_GELU_LAYERNORM, ln_weight=gamma, ln_bias=beta, type_bits=0,
                            col=plan.perm)
    want = ops.linear(agg, w, b, act=act)
    keep = torch.arange(N, device="cuda") != 17
    assert torch.equal(buf[:, out_dim:][keep], want[keep]) and float(buf[:, :out_dim].abs().max()) == 0.0


def test_gather_update_table_form_with_types_and_refusals():
    from ptgnn_amd import PtgnnAmdError, ops
    g = torch.Generator().manual_seed(3)
    N, T, M = 9_000, 3, 64
    adj = [(torch.randint(0, N, (e,), generator=g), torch.randint(0, N, (e,), generator=g)) for e in (20_000, 7, 31_000)]
    cadj = to_cuda_adj(adj)
    y = torch.randn(N, T * M, generator=g).cuda()               # per-node table: block t = X W_t^T
    w = (torch.randn(64, M, generator=g) / 8).cuda()
    ops.clear_plan_cache()
    plan = ops.plan_for(cadj, N)
    got = ops.gather_update(y, plan, "max", plan.col, plan.type_bits, ops.EPI_GELU, None, None, 1e-5, w, None, "tanh")
    want = ops.linear(ops.gather_reduce(y, plan, M, "max", epilogue=ops.EPI_GELU), w, None, act="tanh")
    assert torch.equal(got, want)
    assert not ops.gather_update_supported(128, 64, plan) and not ops.gather_update_supported(64, 48, plan)
    with pytest.raises(PtgnnAmdError):      # message width 128: not a shape of the fused kernel
        ops.gather_update(torch.randn(N, 128).cuda(), plan, "max", plan.col, plan.type_bits, 0, None, None, 1e-5,
                          torch.randn(64, 128).cuda(), None, None)


@pytest.mark.parametrize("agg", ["sum", "max"])
@pytest.mark.parametrize("use_target", [True, False])
def test_mlp_layer_hidden_64_takes_the_fused_kernel_and_matches_the_oracle(agg, use_target, monkeypatch):
    from oracle import mp_oracle as O
    from ptgnn_amd import layers as L, ops, workloads
    mb = workloads.batched_graphs(6, 1500, 5, 2.4, seed=9)
    N, H = mb["num_nodes"], 64
    adj = mb["adjacency_lists"]
    adj = adj + [(d, s) for s, d in adj] + [(torch.arange(N), torch.arange(N))]
    T = len(adj)
    torch.manual_seed(4)
    layer = L.MlpMessagePassingLayer(H, H, H, T, agg, use_target_state_as_message_input=use_target).eval()
    spec = layer.export_weights()
    x = workloads.node_states(N, H, seed=2)
    with torch.no_grad():
        want = O.mlp_mp_layer(x, adj, [torch.empty(a[0].shape[0], 0) for a in adj], spec)
    layer = layer.cuda()
    cadj = to_cuda_adj(adj)
    outs = {}
    monkeypatch.setattr(ops, "_HUB_SKIP", [0])       # an earlier test's hub plan must not send this one to the unfused pair
    for fused in (True, False):
        monkeypatch.setattr(ops, "GATHER_UPDATE", fused)
        ops.clear_plan_cache()
        before = ops.launch_counts()
        with torch.no_grad():
            outs[fused] = layer(x.cuda(), cadj, None, {}, {}, empty_feats(cadj, "cuda"))
        ran = ops.launches_since(before)
        assert ("k_gather_update" in ran) == fused, ran
    assert torch.equal(outs[True], outs[False])
    np.testing.assert_allclose(outs[True].cpu().numpy(), want.numpy(), rtol=0, atol=TOL)


@pytest.mark.parametrize("hidden", [[64], [32, 64]])
@pytest.mark.parametrize("use_target", [True, False])
def test_deeper_edge_mlps_run_their_first_linear_on_the_grouped_gemm(hidden, use_target, monkeypatch):
    """mlp_hidden_layers > 0 (mlp.py:50-77) without edge features: no index_select / [E, 2H] concat -- the first Linear of
    every edge type is the grouped per-edge GEMM (inference and training), the rest runs on the type's rows.  Checked
    against the layer's own CPU-tensor route (= the reference's arithmetic, tests/test_torch_route_cpu.py): outputs 1e-5,
    every gradient 2e-5 relative."""
    import copy
    from ptgnn_amd import layers as L, ops, workloads
    mb = workloads.batched_graphs(5, 800, 3, 2.5, seed=3)
    N, H = mb["num_nodes"], 64
    adj = mb["adjacency_lists"]
    adj = adj + [(d, s) for s, d in adj] + [(torch.arange(N), torch.arange(N))]
    T = len(adj)
    torch.manual_seed(8)
    cpu_layer = L.MlpMessagePassingLayer(H, H, 64, T, "max", use_target_state_as_message_input=use_target,
                                         mlp_hidden_layers=hidden)
    gpu_layer = copy.deepcopy(cpu_layer).cuda()
    x = workloads.node_states(N, H, seed=4)
    feats = [torch.empty(a[0].shape[0], 0) for a in adj]
    cadj = to_cuda_adj(adj)

    def no_gather(*a, **k):
        raise AssertionError("index_select on the per-edge path")
    # inference
    with torch.no_grad():
        want = cpu_layer.eval()(x, adj, None, {}, {}, feats)
    ops.clear_plan_cache()
    before = ops.launch_counts()
    monkeypatch.setattr(torch.Tensor, "index_select", no_gather)
    with torch.no_grad():
        got = gpu_layer.eval()(x.cuda(), cadj, None, {}, {}, empty_feats(cadj, "cuda"))
    monkeypatch.undo()
    ran = ops.launches_since(before)
    assert "k_stream_edge" in ran or "k_edge_linear" in ran, ran
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=0, atol=TOL)
    # training: output, d x and every parameter gradient
    xc = x.clone().requires_grad_(True)
    yc = cpu_layer.train()(xc, adj, None, {}, {}, feats)
    gout = torch.linspace(-1, 1, yc.numel()).view_as(yc)
    yc.backward(gout)
    xg = x.cuda().requires_grad_(True)
    ops.clear_plan_cache()
    monkeypatch.setattr(torch.Tensor, "index_select", no_gather)
    yg = gpu_layer.train()(xg, cadj, None, {}, {}, empty_feats(cadj, "cuda"))
    monkeypatch.undo()
    yg.backward(gout.cuda())
    np.testing.assert_allclose(yg.detach().cpu().numpy(), yc.detach().numpy(), rtol=0, atol=TOL)
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xc.grad.numpy(), rtol=0, atol=2e-5 * max(1.0, float(xc.grad.abs().max())))
    for (k, pc), (_, pg) in zip(cpu_layer.named_parameters(), gpu_layer.named_parameters()):
        # (5e-5: column sums over ~4 000 rows in two different fp32 orders -- the LayerNorm bias gradient sat at 2.1e-5)
        np.testing.assert_allclose(pg.grad.cpu().numpy(), pc.grad.numpy(), rtol=0,
                                   atol=5e-5 * max(1.0, float(pc.grad.abs().max())), err_msg=k)


def test_a_plan_with_hub_rows_sends_the_following_calls_to_the_unfused_pair(monkeypatch):
    """The fused kernel folds every row serially; a plan that reports hub rows (count read back asynchronously, no host
    synchronisation on the hot path) switches the next GATHER_UPDATE_BACKOFF decisions to gather_reduce + linear."""
    from ptgnn_amd import ops
    monkeypatch.setattr(ops, "_HUB_SKIP", [0])
    monkeypatch.setattr(ops, "_HUB_PENDING", [])
    g = torch.Generator().manual_seed(2)
    N, E = 20_000, 60_000
    src, dst = torch.randint(0, N, (E,), generator=g), torch.randint(0, N, (E,), generator=g)
    plain = ops.build_plan(to_cuda_adj([(src, dst)]), N)
    assert ops.gather_update_supported(64, 64, plain)
    torch.cuda.synchronize()
    assert ops.gather_update_supported(64, 64, plain) and ops._HUB_SKIP[0] == 0
    dst2 = dst.clone()
    dst2[: ops.HUB_THRESHOLD + 500] = 7                       # one hub row
    hubby = ops.build_plan(to_cuda_adj([(src, dst2)]), N)
    assert ops.gather_update_supported(64, 64, hubby)        # the count is not back yet: still fused (exact either way)
    torch.cuda.synchronize()
    fresh = ops.build_plan(to_cuda_adj([(src, dst)]), N)     # a plan whose own count is still unknown ...
    assert not ops.gather_update_supported(64, 64, fresh)    # ... goes by the recent plans: the read-back has arrived, back off
    assert ops._HUB_SKIP[0] == ops.GATHER_UPDATE_BACKOFF - 1
    # what a plan reported about ITSELF decides its later calls, whatever the back-off says: the hub-free plan
    # stays fused, the hub plan never takes the serial fold again -- also after the back-off has run out
    assert plain._has_hubs is False and hubby._has_hubs is True
    assert ops.gather_update_supported(64, 64, plain) and ops._HUB_SKIP[0] == ops.GATHER_UPDATE_BACKOFF - 1
    monkeypatch.setattr(ops, "_HUB_SKIP", [0])
    assert not ops.gather_update_supported(64, 64, hubby) and ops.gather_update_supported(64, 64, plain)
    monkeypatch.setattr(ops, "_HUB_SKIP", [1])
    other = ops.build_plan(to_cuda_adj([(src, dst)]), N)
    assert not ops.gather_update_supported(64, 64, other) and ops.gather_update_supported(64, 64, other)
