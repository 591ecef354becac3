"""GPU parity tests (run with `-m gpu` on an MI355X): the HIP path through the C ABI versus the CPU
oracle on the same seeded inputs, versus the golden fixtures produced by the reference's own
modules, and -- at BASELINE.json config-2 size -- through size-independent properties.

Bars: integer/index outputs bit-exact; max/min aggregation bit-exact; fp32 sums bit-exact where the
fold order is the reference's (segment reduce) and |delta| <= 1e-5 for full layers (north_star).
"""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from helpers import empty_feats, layer_from_spec, stack_from_specs, to_cuda_adj

pytestmark = pytest.mark.gpu

TOL = 1e-5  # BASELINE.json north_star: fp32 node states within 1e-5 of the reference CPU path


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("these tests need the MI355X (run them through gpurun)")
    from ptgnn_amd import _lib
    _lib.load()  # fail loudly if the HIP library is missing


@pytest.fixture
def plan_path():
    """Forces one path of the plan build (ptgnn_amd_set_plan_path) for a test and restores the default."""
    from ptgnn_amd import _lib
    lib = _lib.load()

    def set_path(name):
        _lib.check(lib.ptgnn_amd_set_plan_path({"auto": 0, "wide_records": 1, "lsd": 2}[name]), "set_plan_path")
    yield set_path
    lib.ptgnn_amd_set_plan_path(0)


def ref_csr(adj, num_nodes, transposed=False):
    """numpy restatement of the plan: stable sort of the type-major edge list by destination."""
    T = len(adj)
    tb = int(np.ceil(np.log2(T))) if T > 1 else 0
    src = np.concatenate([a[0].numpy() for a in adj]) if T else np.zeros(0, np.int64)
    dst = np.concatenate([a[1].numpy() for a in adj]) if T else np.zeros(0, np.int64)
    typ = np.concatenate([np.full(a[0].shape[0], t, np.int64) for t, a in enumerate(adj)])
    if transposed:
        src, dst = dst, src
    order = np.argsort(dst, kind="stable")
    rowptr = np.zeros(num_nodes + 1, np.int64)
    np.add.at(rowptr, dst + 1, 1)
    rowptr = np.cumsum(rowptr)
    col = (src[order] << tb) | typ[order]
    return rowptr.astype(np.int32), col.astype(np.int32), order.astype(np.int32), tb


# ------------------------------------------------------------------------------------------------
# plan (integer work: bit exact)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", ["random3", "one_empty_type", "all_empty", "hub", "single", "many_types",
                                  "types_over_table", "rows_over_2p18", "dense_tiny", "one_row_graph",
                                  "tile_multiple", "tiles_over_residency"])
@pytest.mark.parametrize("transposed", [False, True])
@pytest.mark.parametrize("path", ["auto", "wide_records", "lsd"])
def test_csr_build_bit_exact(case, transposed, path, plan_path):
    from ptgnn_amd import ops
    plan_path(path)
    if path != "auto" and case == "tiles_over_residency" and transposed:
        pytest.skip("one orientation of the 3 M-edge case per forced path is enough")
    g = torch.Generator().manual_seed(5)
    n = 1000
    ri = lambda c: torch.randint(0, n, (c,), generator=g, dtype=torch.int64)  # noqa: E731
    z = torch.zeros(0, dtype=torch.int64)
    adj = {
        "random3": [(ri(5000), ri(5000)), (ri(17), ri(17)), (ri(2500), ri(2500))],
        "one_empty_type": [(ri(300), ri(300)), (z, z), (ri(40), ri(40))],
        "all_empty": [(z, z), (z, z)],
        "hub": [(ri(20000), torch.full((20000,), 7, dtype=torch.int64)), (ri(100), ri(100))],
        "single": [(ri(1), ri(1))],
        "many_types": [(ri(50 + 13 * t), ri(50 + 13 * t)) for t in range(23)],
        # > 64 edge types: the flat LSD path with its chunked type tables
        "types_over_table": [(ri(20 + 3 * t), ri(20 + 3 * t)) for t in range(70)],
        # row ids beyond 18 bits: outside the two-level build's envelope
        "rows_over_2p18": [(torch.randint(0, 300_000, (40_000,), generator=g),
                            torch.randint(0, 300_000, (40_000,), generator=g))],
        # 750 edges per row: every bucket of the two-level build spans many 1024-edge chunks
        "dense_tiny": [(torch.randint(0, 40, (30_000,), generator=g), torch.randint(0, 40, (30_000,), generator=g))],
        "one_row_graph": [(torch.zeros(5, dtype=torch.int64), torch.zeros(5, dtype=torch.int64))],
        # exactly two 4096-edge tiles of the scatter pass
        "tile_multiple": [(ri(8192), ri(8192))],
        # 733 tiles: more workgroups than fit on the chip at once, so the look-back of the scatter pass crosses
        # tiles that were not resident together (ticket order keeps it deadlock-free)
        "tiles_over_residency": [(torch.randint(0, 200_000, (c,), generator=g), torch.randint(0, 200_000, (c,), generator=g))
                                 for c in (2_000_000, 1_000_000)],
    }[case]
    n = {"rows_over_2p18": 300_000, "dense_tiny": 40, "one_row_graph": 1, "tiles_over_residency": 200_000}.get(case, n)
    cadj = to_cuda_adj(adj)
    ops.build_plan(cadj, n, transposed=transposed)     # a first build: the second one reuses its control block
    plan = ops.build_plan(cadj, n, transposed=transposed)
    rowptr, col, perm, tb = ref_csr(adj, n, transposed)
    E = len(col)
    assert plan.num_edges == E and plan.type_bits == tb and plan.num_types == len(adj)
    np.testing.assert_array_equal(plan.rowptr.cpu().numpy(), rowptr)
    np.testing.assert_array_equal(plan.col[:E].cpu().numpy(), col)
    np.testing.assert_array_equal(plan.perm[:E].cpu().numpy(), perm)


def _zipf_dst(n, e, alpha, gen):
    """BASELINE config 5's destinations: w_i = (i + 1)^-alpha, inverse-CDF sampling, then a fixed
    random node permutation."""
    w = torch.arange(1, n + 1, dtype=torch.float64).pow(-alpha)
    cdf = torch.cumsum(w, 0)
    cdf /= cdf[-1].clone()
    u = torch.rand(e, generator=gen, dtype=torch.float64)
    ids = torch.searchsorted(cdf, u).clamp_(max=n - 1)
    return torch.randperm(n, generator=gen)[ids]


@pytest.mark.parametrize("case", ["cfg5_scaled_21bit", "rows_22bit_prepass", "edges_over_4m_18bit"])
def test_csr_build_bit_exact_beyond_minibatch_sizes(case):
    """The plan above 4 M edges / 2^18 rows (BASELINE config 5's per-GPU shard is 1.25 M rows / 12.5 M edges), bit
    for bit against numpy's stable argsort, with the hub list checked against the row lengths."""
    from ptgnn_amd import ops
    g = torch.Generator().manual_seed(11)
    n, e, zipf = {"cfg5_scaled_21bit": (1_250_000, 5_300_000, True),      # 21 row bits: 12-byte records, 4096-row buckets
                  "rows_22bit_prepass": (3_000_000, 5_000_000, True),      # 22 row bits: LSD passes (as for anything above 4 M edges)
                  "edges_over_4m_18bit": (200_000, 4_500_000, False)}[case]
    dst = _zipf_dst(n, e, 0.8, g) if zipf else torch.randint(0, n, (e,), generator=g, dtype=torch.int64)
    src = torch.randint(0, n, (e,), generator=g, dtype=torch.int64)
    cut = e // 3
    adj = [(src[:cut], dst[:cut]), (src[cut:], dst[cut:])]
    plan = ops.build_plan(to_cuda_adj(adj), n)
    rowptr, col, perm, tb = ref_csr(adj, n)
    np.testing.assert_array_equal(plan.rowptr.cpu().numpy(), rowptr)
    np.testing.assert_array_equal(plan.col[:e].cpu().numpy(), col)
    np.testing.assert_array_equal(plan.perm[:e].cpu().numpy(), perm)
    deg = np.diff(rowptr.astype(np.int64))
    want = set()
    for row in np.nonzero(deg > ops.HUB_THRESHOLD)[0]:
        for c in range(rowptr[row] // 1024, (rowptr[row + 1] - 1) // 1024 + 1):
            want.add((int(c), int(row)))
    count = int(plan.hub_count.item())
    got = set(map(tuple, plan.hub_entries[:count].cpu().tolist()))
    assert count == len(want) and got == want
    if zipf:
        assert len(want) > 0, "the case is meant to contain hub rows"


@pytest.mark.parametrize("path", ["auto", "wide_records", "lsd"])
def test_backward_plan_mode2_bit_exact(path, plan_path):
    """rows = src * T + type, col = dst (the plan of the message-table gradient): 21 row bits at Graph2Class size."""
    from ptgnn_amd import ops
    plan_path(path)
    g = torch.Generator().manual_seed(3)
    n, T = 60_000, 17
    adj = [(torch.randint(0, n, (c,), generator=g), torch.randint(0, n, (c,), generator=g))
           for c in [40_000 // (t + 1) + 5 for t in range(T)]]
    plan = ops.build_plan(to_cuda_adj(adj), n * T, mode=2)
    src = np.concatenate([a[0].numpy() for a in adj])
    dst = np.concatenate([a[1].numpy() for a in adj])
    typ = np.concatenate([np.full(a[0].shape[0], t, np.int64) for t, a in enumerate(adj)])
    key = src * T + typ
    order = np.argsort(key, kind="stable")
    rowptr = np.zeros(n * T + 1, np.int64)
    np.add.at(rowptr, key + 1, 1)
    E = len(key)
    np.testing.assert_array_equal(plan.rowptr.cpu().numpy(), np.cumsum(rowptr).astype(np.int32))
    np.testing.assert_array_equal(plan.col[:E].cpu().numpy(), dst[order].astype(np.int32))
    np.testing.assert_array_equal(plan.perm[:E].cpu().numpy(), order.astype(np.int32))


def test_csr_build_rejects_cpu_and_int32():
    from ptgnn_amd import PtgnnAmdError, ops
    a = torch.zeros(3, dtype=torch.int64)
    with pytest.raises(PtgnnAmdError):
        ops.build_plan([(a, a)], 4)
    with pytest.raises(PtgnnAmdError):
        ops.build_plan([(a.cuda().int(), a.cuda().int())], 4)


def test_plan_cache_identity():
    from ptgnn_amd import ops
    ops.clear_plan_cache()
    a = torch.arange(10, device="cuda")
    adj = [(a, a.flip(0).contiguous())]
    p1 = ops.plan_for(adj, 10)
    assert ops.plan_for([(adj[0][0], adj[0][1])], 10) is p1          # same tensors -> same plan
    assert ops.plan_for([(a.clone(), adj[0][1])], 10) is not p1      # different tensor object
    adj[0][0].add_(0)                                                 # version bump -> rebuilt
    assert ops.plan_for(adj, 10) is not p1


# ------------------------------------------------------------------------------------------------
# the scatter seam
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("reduce", ["sum", "mean", "max", "min"])
@pytest.mark.parametrize("dim", [1, 3, 24, 64, 128, 200, 256, 512, 640])
def test_scatter_matches_oracle(reduce, dim):
    from oracle import scatter_ref
    from ptgnn_amd.scatter import scatter
    g = torch.Generator().manual_seed(dim)
    E, N = 6000, 701
    src = torch.randn(E, dim, generator=g)
    idx = torch.randint(0, N - 1, (E,), generator=g)
    idx[idx == 13] = 14                       # empty segments: 13 and N-1
    idx[:900] = 5                             # one long segment (many full groups of the fold loop + a tail)
    want = scatter_ref.scatter(src, idx, dim=0, dim_size=N, reduce=reduce)
    got = scatter(src.cuda(), idx.cuda(), dim=0, dim_size=N, reduce=reduce).cpu()
    assert got.dtype == torch.float32 and got.shape == want.shape
    if reduce == "mean":
        np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-6, atol=1e-6)
    else:
        # same fold order as the reference's CPU scatter_add_ / order-independent max,min
        np.testing.assert_array_equal(got.numpy(), want.numpy())
    assert float(got[13].abs().sum()) == 0.0 and float(got[N - 1].abs().sum()) == 0.0


@pytest.mark.parametrize("num_types", [1, 3])
@pytest.mark.parametrize("dim", [32, 64, 128, 256, 512])
def test_fused_table_form_equals_segment_reduce_of_materialised_messages(num_types, dim):
    """Bit for bit: gather + destination term + reduce in one kernel (groups of 8 slots; groups of 4 with a per-slot
    destination term; the term loaded once per row when there is one edge type) == the segment reduce of the
    materialised messages y[src, t] + yd[dst, t] (same adds, same CSR fold order).  Degrees 0 .. ~300 plus one hub
    row, so every group / tail / clamp case of the fold loop occurs."""
    from ptgnn_amd import ops
    g = torch.Generator().manual_seed(num_types * 1000 + dim)
    N, T, M = 3001, num_types, dim
    adj = []
    for t in range(T):
        e = 24000 // T
        src = torch.randint(0, N, (e,), generator=g)
        dst = torch.randint(0, N, (e,), generator=g)
        dst[: e // 8] = torch.randint(0, 40, (e // 8,), generator=g)      # 40 rows of ~75 .. 300 in-edges
        dst[e // 8: e // 8 + (ops.HUB_THRESHOLD + 200) // T + 1] = 77      # one hub row
        dst[dst == 1500] = 1501                                            # an empty row
        adj.append((src.cuda(), dst.cuda()))
    plan = ops.build_plan(adj, N)
    deg = plan.rowptr[1:] - plan.rowptr[:-1]
    assert int(deg.max()) > ops.HUB_THRESHOLD and int(deg[1500]) == 0
    plain_rows = (deg <= ops.HUB_THRESHOLD)          # hub rows fold chunk-wise: sums are not bit-comparable
    y = torch.randn(N, T * M, generator=g).cuda()
    yd = torch.randn(N, T * M, generator=g).cuda()
    msgs = torch.cat([y[s][:, t * M:(t + 1) * M] + yd[d][:, t * M:(t + 1) * M] for t, (s, d) in enumerate(adj)])
    plain = torch.cat([y[s][:, t * M:(t + 1) * M] for t, (s, d) in enumerate(adj)])
    for reduce in ("sum", "mean", "max", "min"):
        for with_dst in (True, False):
            want = ops.segment_reduce(msgs if with_dst else plain, plan, reduce)
            got = ops.gather_reduce(y, plan, M, reduce, ydst=yd if with_dst else None)
            rows = slice(None) if reduce in ("max", "min") else plain_rows
            assert torch.equal(want[rows], got[rows]), (reduce, with_dst)
            if reduce in ("max", "min"):
                w_arg = ops.segment_reduce(msgs if with_dst else plain, plan, reduce, return_arg=True)
                g_arg = ops.gather_reduce(y, plan, M, reduce, ydst=yd if with_dst else None, return_arg=True)
                assert torch.equal(w_arg[0], g_arg[0]) and torch.equal(w_arg[1], g_arg[1]), (reduce, with_dst, "arg")


def test_scatter_known_answers_and_1d():
    from ptgnn_amd.scatter import scatter
    src = torch.tensor([[1.0, -2.0], [3.0, 4.0], [5.0, -6.0], [-7.0, 8.0], [0.5, 0.25]]).cuda()
    idx = torch.tensor([2, 0, 2, 2, 0]).cuda()
    exp = {"sum": [[3.5, 4.25], [0, 0], [-1.0, 0.0], [0, 0]],
           "max": [[3.0, 4.0], [0, 0], [5.0, 8.0], [0, 0]],
           "min": [[0.5, 0.25], [0, 0], [-7.0, -6.0], [0, 0]]}
    for r, e in exp.items():
        np.testing.assert_array_equal(scatter(src, idx, 0, dim_size=4, reduce=r).cpu().numpy(),
                                      np.asarray(e, np.float32))
    out = scatter(src[:, 0].contiguous(), idx, dim=0, reduce="add")  # dim_size from index.max()+1
    np.testing.assert_array_equal(out.cpu().numpy(), np.asarray([3.5, 0.0, -1.0], np.float32))
    out = scatter(torch.zeros(0, 3).cuda(), torch.zeros(0, dtype=torch.int64).cuda(), 0, dim_size=5,
                  reduce="max")
    assert tuple(out.shape) == (5, 3) and float(out.abs().sum()) == 0.0


@pytest.mark.parametrize("shape", [(4000, 1), (4000, 8), (300,)])
def test_scatter_facade_family_matches_oracle(shape):
    """scatter_sum / _mean / _max / _min (values + torch_scatter's arg positions) and the composite
    scatter_softmax / scatter_log_softmax (forward and autograd) vs oracle/scatter_ref.py."""
    from oracle import scatter_ref as R
    from ptgnn_amd import scatter as S
    g = torch.Generator().manual_seed(sum(shape))
    E, segs = shape[0], 57
    src = torch.randn(*shape, generator=g)
    idx = torch.randint(0, segs - 5, (E,), generator=g)          # trailing segments stay empty
    csrc, cidx = src.cuda(), idx.cuda()
    for name in ("scatter_sum", "scatter_add", "scatter_mean"):
        want = getattr(R, name)(src, idx, dim=0, dim_size=segs)
        got = getattr(S, name)(csrc, cidx, dim=0, dim_size=segs)
        np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=0, atol=2e-5)
    for name in ("scatter_max", "scatter_min"):
        wv, wa = getattr(R, name)(src, idx, dim=0, dim_size=segs)
        gv, ga = getattr(S, name)(csrc, cidx, dim=0, dim_size=segs)
        np.testing.assert_array_equal(gv.cpu().numpy(), wv.numpy())
        np.testing.assert_array_equal(ga.cpu().numpy(), wa.numpy())
    gout = torch.randn(*shape, generator=g)
    for name, eps in (("scatter_log_softmax", 0.0), ("scatter_softmax", 1e-12)):
        xo = src.clone().requires_grad_(True)
        want = getattr(R, name)(xo, idx, dim=0, eps=eps, dim_size=segs)
        want.backward(gout)
        xg = csrc.clone().requires_grad_(True)
        got = getattr(S, name)(xg, cidx, dim=0, eps=eps, dim_size=segs)
        got.backward(gout.cuda())
        np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().numpy(), rtol=0, atol=TOL)
        np.testing.assert_allclose(xg.grad.cpu().numpy(), xo.grad.numpy(), rtol=0, atol=2e-5)


@pytest.mark.parametrize("shape", [(3000, 48), (3000, 7), (500,)])
def test_scatter_mul_matches_oracle_forward_and_backward(shape):
    """reduce="mul" of the scatter seam (torch_scatter.scatter_mul; abstractmessagepassing.py:44-50 passes any of
    torch_scatter's reduce names through): product in edge order, empty segments 1, backward
    (grad * out)[index] / src -- vs oracle/scatter_ref.py (restated + KATs in tests/test_oracle_kat.py)."""
    from oracle import scatter_ref as R
    from ptgnn_amd import scatter as S
    g = torch.Generator().manual_seed(sum(shape))
    E, segs = shape[0], 257
    src = 1.0 + 0.25 * torch.randn(*shape, generator=g)            # products of ~12 factors near 1
    idx = torch.randint(0, segs - 4, (E,), generator=g)            # trailing segments stay empty -> 1
    go = torch.randn(segs, *shape[1:], generator=g)
    a = src.clone().requires_grad_(True)
    want = R.scatter(a, idx, dim=0, dim_size=segs, reduce="mul")
    want.backward(go)
    b = src.clone().cuda().requires_grad_(True)
    got = S.scatter(b, idx.cuda(), dim=0, dim_size=segs, reduce="mul")
    got.backward(go.cuda())
    assert torch.equal(got[segs - 4:].cpu(), torch.ones_like(want[segs - 4:]))
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().numpy(), rtol=2e-6, atol=1e-6)
    np.testing.assert_allclose(b.grad.cpu().numpy(), a.grad.numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(S.scatter_mul(b.detach(), idx.cuda(), dim=0, dim_size=segs).cpu().numpy(),
                               want.detach().numpy(), rtol=2e-6, atol=1e-6)


def test_ggnn_layer_with_mul_aggregation_matches_oracle():
    """A GGNN layer whose `message_aggregation_function` is "mul" (legal in the reference: the name goes straight to
    torch_scatter) runs on the HIP GEMMs + the seam's product kernel, inference and training, vs the oracle."""
    from oracle import mp_oracle as O
    from ptgnn_amd import layers as L, ops, workloads
    mb = workloads.batched_graphs(3, 200, 3, 1.2, seed=15)
    N, H = mb["num_nodes"], 32
    adj = O.augment_adjacency(mb["adjacency_lists"], N, True, True)
    T = len(adj)
    torch.manual_seed(4)
    layer = L.GatedMessagePassingLayer(H, H, T, "mul")
    x = workloads.node_states(N, H, seed=9)
    spec = layer.export_weights()
    xo = x.clone().requires_grad_(True)
    want = O.ggnn_layer(xo, adj, [torch.empty(a[0].shape[0], 0) for a in adj], spec)
    gout = workloads.node_states(N, H, seed=10)
    want.backward(gout)
    layer = layer.cuda()
    cadj = to_cuda_adj(adj)
    ops.clear_plan_cache()
    with torch.no_grad():
        got = layer.eval()(x.cuda(), cadj, None, {}, {}, empty_feats(cadj, "cuda"))
    np.testing.assert_allclose(got.cpu().numpy(), want.detach().numpy(), rtol=0, atol=TOL)
    xg = x.cuda().requires_grad_(True)
    layer.train()(xg, cadj, None, {}, {}, empty_feats(cadj, "cuda")).backward(gout.cuda())
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xo.grad.numpy(), rtol=0,
                               atol=2e-5 * max(1.0, float(xo.grad.abs().max())))


@pytest.mark.parametrize("reduce", ["sum", "mean", "max", "min"])
def test_scatter_backward_matches_oracle_autograd(reduce):
    from oracle import scatter_ref
    from ptgnn_amd.scatter import scatter
    g = torch.Generator().manual_seed(11)
    E, N, D = 3000, 257, 48
    src = torch.randn(E, D, generator=g)
    idx = torch.randint(0, N - 3, (E,), generator=g)
    go = torch.randn(N, D, generator=g)
    a = src.clone().requires_grad_(True)
    scatter_ref.scatter(a, idx, dim=0, dim_size=N, reduce=reduce).backward(go)
    b = src.clone().cuda().requires_grad_(True)
    scatter(b, idx.cuda(), dim=0, dim_size=N, reduce=reduce).backward(go.cuda())
    np.testing.assert_allclose(b.grad.cpu().numpy(), a.grad.numpy(), rtol=1e-6, atol=1e-6)


# ------------------------------------------------------------------------------------------------
# dense blocks
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,k,n_out", [(1, 1, 1), (7, 5, 3), (129, 64, 64), (300, 128, 256),
                                          (1000, 130, 70), (257, 256, 384), (4096, 128, 2176)])
@pytest.mark.parametrize("act,bias", [(None, False), ("tanh", True), ("relu", True)])
def test_linear_matches_fp32_reference(rows, k, n_out, act, bias):
    from ptgnn_amd import ops
    g = torch.Generator().manual_seed(rows + k)
    x = torch.randn(rows, k, generator=g)
    w = torch.randn(n_out, k, generator=g) / k ** 0.5
    b = torch.randn(n_out, generator=g) if bias else None
    want = x.double() @ w.double().t()
    if b is not None:
        want = want + b.double()
    want = {"tanh": torch.tanh, "relu": torch.relu, None: lambda v: v}[act](want)
    got = ops.linear(x.cuda(), w.cuda(), b.cuda() if bias else None, act=act).cpu()
    np.testing.assert_allclose(got.numpy(), want.float().numpy(), rtol=0, atol=TOL)


def test_linear_strided_views():
    from ptgnn_amd import ops
    g = torch.Generator().manual_seed(3)
    big = torch.randn(300, 96, generator=g).cuda()
    w = torch.randn(40, 32, generator=g).cuda()
    x = big[:, 32:64]                                   # column slice: ld 96
    out = torch.zeros(300, 100, device="cuda")
    ops.linear(x, w, out=out[:, 10:50])
    want = (x.double().cpu() @ w.double().cpu().t()).float()
    np.testing.assert_allclose(out[:, 10:50].cpu().numpy(), want.numpy(), atol=TOL)
    assert float(out[:, :10].abs().sum()) == 0 and float(out[:, 50:].abs().sum()) == 0


@pytest.mark.parametrize("n,m,h", [(1, 4, 4), (77, 24, 16), (300, 128, 128), (513, 128, 256),
                                   (200, 100, 36)])
def test_gru_cell_matches_oracle(n, m, h):
    from oracle import mp_oracle as O
    from ptgnn_amd import ops
    g = torch.Generator().manual_seed(n)
    a, hh = torch.randn(n, m, generator=g), torch.randn(n, h, generator=g)
    w_ih, w_hh = torch.randn(3 * h, m, generator=g) / m ** 0.5, torch.randn(3 * h, h, generator=g) / h ** 0.5
    b_ih, b_hh = torch.randn(3 * h, generator=g) * 0.1, torch.randn(3 * h, generator=g) * 0.1
    want = O.gru_cell(a, hh, w_ih, w_hh, b_ih, b_hh)
    torch_want = torch._VF.gru_cell(a, hh, w_ih, w_hh, b_ih, b_hh)
    np.testing.assert_allclose(want.numpy(), torch_want.numpy(), atol=2e-6)   # oracle == torch GRUCell
    got = ops.gru_cell(a.cuda(), hh.cuda(), w_ih.cuda(), w_hh.cuda(), b_ih.cuda(), b_hh.cuda()).cpu()
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=0, atol=TOL)


@pytest.mark.parametrize("n,m,h", [(5000, 256, 256), (131, 256, 256), (4097, 384, 256), (3000, 128, 128), (900, 64, 192)])
def test_gru_ring_kernel_is_bit_identical_to_the_other_gru_kernels(n, m, h, monkeypatch):
    """The fused GRU cell whose weight slab does not fit LDS (K = M + H > ~400: BASELINE config 5, H = M = 256)
    streams its weights through a two-panel ring (stream_gemm.hip k_stream_gru_ring).  It accumulates K in the
    library's one fixed order with the streaming kernels' gate math, so at shapes both take it must reproduce the
    slab-resident streaming kernel bit for bit (outputs and training-time gates); the tile kernel differs only by
    its libm gate math (<= 1e-6); all within 1e-5 of float64 (gatedmessagepassing.py:69)."""
    from ptgnn_amd import ops
    g = torch.Generator().manual_seed(n + m)
    a, hh = torch.randn(n, m, generator=g), torch.randn(n, h, generator=g)
    w_ih, w_hh = torch.randn(3 * h, m, generator=g) / m ** 0.5, torch.randn(3 * h, h, generator=g) / h ** 0.5
    b_ih, b_hh = torch.randn(3 * h, generator=g) * 0.1, torch.randn(3 * h, generator=g) * 0.1
    want64 = torch._VF.gru_cell(a.double(), hh.double(), w_ih.double(), w_hh.double(), b_ih.double(), b_hh.double())
    args = [t.cuda() for t in (a, hh, w_ih, w_hh, b_ih, b_hh)]
    monkeypatch.setenv("PTGNN_AMD_GRU_RING", "1")            # also where the resident slab would fit
    ring = ops.gru_cell(*args)
    ring_out, ring_gates = ops.gru_cell_train(*args)
    monkeypatch.setenv("PTGNN_AMD_GRU_RING", "0")            # never the ring: resident slab, or the tile kernel
    other = ops.gru_cell(*args)
    other_out, other_gates = ops.gru_cell_train(*args)
    prev = ops.set_gemm_mode("tile")
    try:
        tile = ops.gru_cell(*args)
    finally:
        ops.set_gemm_mode(prev)
    assert torch.equal(ring, ring_out)
    slab_fits = 96 * (m + h + 4) * 4 + 16 + 8 * 288 * 4 <= 160 * 1024     # stream_gemm.hip stream_gru
    if slab_fits:    # `other` is the slab-resident streaming kernel: same MFMA order, same gate-math routine
        assert torch.equal(ring, other) and torch.equal(ring_out, other_out) and torch.equal(ring_gates, other_gates)
    else:            # `other` is the tile kernel, whose gate math is libm's (the streaming kernels use v_exp / v_rcp)
        assert float((ring - other).abs().max()) <= 1e-6 and float((ring_gates - other_gates).abs().max()) <= 1e-6
    assert float((ring - tile).abs().max()) <= 1e-6
    assert float((ring.cpu().double() - want64).abs().max()) <= TOL


@pytest.mark.parametrize("rows,k,n_out,act", [
    (5000, 384, 128, None),        # the GRU's input-gradient GEMM at H = 128: slab too large, ring by default
    (4099, 128, 128, "tanh"),      # forced onto the ring where the resident slab fits: ragged last unit, activation
    (33000, 256, 256, None),       # two column slabs
    (2048, 512, 128, "relu"),      # the smallest row count the ring takes
])
def test_linear_ring_kernel_is_bit_identical_to_the_other_linear_kernels(rows, k, n_out, act, monkeypatch):
    """Linear layers whose [128, K] weight slab does not fit LDS stream the weights through a two-panel ring
    (stream_gemm.hip k_stream_linear_ring).  K is accumulated in the library's one fixed order, so the ring, the
    slab-resident streaming kernel and the tile kernel give the same bits; float64 within 1e-5 relative to the size of
    the sums."""
    from ptgnn_amd import ops
    g = torch.Generator().manual_seed(rows + k)
    x, w, b = torch.randn(rows, k, generator=g), torch.randn(n_out, k, generator=g) / k ** 0.5, torch.randn(n_out, generator=g)
    want = x.double() @ w.double().t() + b.double()
    want = torch.tanh(want) if act == "tanh" else (torch.relu(want) if act == "relu" else want)
    xc, wc, bc = x.cuda(), w.cuda(), b.cuda()
    monkeypatch.setenv("PTGNN_AMD_LINEAR_RING", "1")
    timer = ops.KernelTimer()
    ring = ops.linear(xc, wc, bc, act=act)
    strided = torch.empty(rows, n_out + 4, device="cuda")[:, 2:2 + n_out]      # rows not 16-byte aligned: dword stores
    ops.linear(xc, wc, bc, act=act, out=strided)
    monkeypatch.setenv("PTGNN_AMD_LINEAR_RING", "0")
    other = ops.linear(xc, wc, bc, act=act)
    prev = ops.set_gemm_mode("tile")
    try:
        tile = ops.linear(xc, wc, bc, act=act)
    finally:
        ops.set_gemm_mode(prev)
    del timer
    assert torch.equal(ring, strided)
    if act == "tanh":      # the tile kernel's tanh is libm's
        assert float((ring - tile).abs().max()) <= 1e-6 and float((ring - other).abs().max()) <= 1e-6
    else:
        assert torch.equal(ring, tile) and torch.equal(ring, other)
    assert float((ring.cpu().double() - want).abs().max()) <= TOL * max(1.0, float(want.abs().max()))


def test_gather_rows():
    from ptgnn_amd import ops
    g = torch.Generator().manual_seed(1)
    for d in (1, 7, 64, 128, 300):
        x = torch.randn(500, d, generator=g)
        idx = torch.randint(0, 500, (1234,), generator=g)
        got = ops.gather_rows(x.cuda(), idx.cuda()).cpu()
        np.testing.assert_array_equal(got.numpy(), x[idx].numpy())


# ------------------------------------------------------------------------------------------------
# layers against golden fixtures of the reference's own modules + against the oracle
# ------------------------------------------------------------------------------------------------
LAYER_CASES = ["ggnn_layer_sum", "ggnn_layer_mean", "ggnn_layer_max", "ggnn_layer_min",
               "mlp_layer_sum_target", "mlp_layer_max_target", "mlp_layer_mean_notarget",
               "mlp_layer_sum_hidden1", "mlp_layer_max_noln_nodense",
               "ggnn_layer_max_w128", "mlp_layer_sum_target_w128"]   # round 4: K % 64 widths (tests/test_gpu_golden_wide.py)


@pytest.mark.parametrize("name", LAYER_CASES)
@pytest.mark.parametrize("path", ["fused", "general"])
def test_layer_matches_reference_golden(name, path):
    from oracle.fixtures import unpack_adj, unpack_specs
    from ptgnn_amd import ops
    g = load_golden(name)
    adj, (spec,) = unpack_adj(g), unpack_specs(g)
    layer = layer_from_spec(spec).cuda().eval()
    x = torch.from_numpy(g["x"]).cuda()
    cadj = to_cuda_adj(adj)
    ops.clear_plan_cache()
    if path == "fused":
        with torch.no_grad():
            y = layer(x, cadj, None, {}, {}, empty_feats(cadj, "cuda"))
    else:  # grad-enabled => per-edge path with the HIP scatter seam
        y = layer(x.requires_grad_(True), cadj, None, {}, {}, empty_feats(cadj, "cuda"))
        assert y.requires_grad
    np.testing.assert_allclose(y.detach().cpu().numpy(), g["y"], rtol=0, atol=TOL)


@pytest.mark.parametrize("name", ["gnn_stack_ggnn_typilus", "gnn_stack_mlp_varmisuse"])
def test_container_matches_reference_golden(name):
    from oracle.fixtures import unpack_adj, unpack_specs
    from ptgnn_amd.gnn import GraphNeuralNetwork
    g = load_golden(name)
    adj, specs = unpack_adj(g), unpack_specs(g)
    net = GraphNeuralNetwork(stack_from_specs(specs), torch.nn.Identity(), introduce_backwards_edges=True,
                             add_self_edges=True).cuda().eval()
    x = torch.from_numpy(g["x"]).cuda()
    n2g = torch.from_numpy(g["node_to_graph_idx"]).cuda()
    refs = {"supernodes": torch.tensor([0, 5, 21, 40, 59]).cuda()}
    ref_g = {"supernodes": torch.tensor([0, 0, 1, 2, 2]).cuda()}
    cadj = to_cuda_adj(adj)
    n_before = len(cadj)
    with torch.no_grad():
        out = net(node_data={"input": x}, adjacency_lists=cadj, edge_feature_data=[],
                  node_to_graph_idx=n2g, reference_node_ids=refs, reference_node_graph_idx=ref_g,
                  num_graphs=3)
        out2 = net(node_data={"input": x}, adjacency_lists=cadj, edge_feature_data=[],
                   node_to_graph_idx=n2g, reference_node_ids=refs, reference_node_graph_idx=ref_g,
                   num_graphs=3)                                   # callable twice (no list mutation)
    assert len(cadj) == n_before
    np.testing.assert_allclose(out.output_node_representations.cpu().numpy(), g["y"], rtol=0, atol=TOL)
    np.testing.assert_array_equal(out.output_node_representations.cpu().numpy(),
                                  out2.output_node_representations.cpu().numpy())   # deterministic
    # index fields are the very same tensor objects, passed through (graphneuralnetwork.py:202-209)
    assert out.node_to_graph_idx is n2g and out.node_idx_references is refs
    assert out.node_graph_idx_reference is ref_g and out.num_graphs == 3
    assert out.reference_nodes_idx is refs and out.input_node_representations is x
    m = net.report_metrics()
    assert m["num_edges"] == 2 * int(g["num_edges"]) and m["num_graphs"] == 6 and m["num_nodes"] == 120
    # the task-head access pattern (graph2class.py:84-89)
    head = out.output_node_representations[out.node_idx_references["supernodes"]]
    np.testing.assert_allclose(head.cpu().numpy(), g["y"][[0, 5, 21, 40, 59]], rtol=0, atol=TOL)


# ------------------------------------------------------------------------------------------------
# BASELINE configs: oracle at sizes it finishes in seconds, properties at full size
# ------------------------------------------------------------------------------------------------
def _mlp_layer(H, M, T, agg, seed):
    from ptgnn_amd import layers as L
    torch.manual_seed(seed)
    return L.MlpMessagePassingLayer(H, H, M, T, agg)


@pytest.mark.parametrize("agg", ["sum", "max", "mean"])
def test_config2_full_size_vs_oracle(agg):
    """BASELINE config 2: N=200k, E=1.1M, 1 MLP-MP layer H=128 -- fp32 vs CPU parity."""
    from oracle import mp_oracle as O
    from ptgnn_amd import ops, workloads
    N, E, H = 200_000, 1_100_000, 128
    adj = workloads.random_graph(N, E)
    x = workloads.node_states(N, H)
    layer = _mlp_layer(H, H, 1, agg, 1234).eval()
    want = O.mlp_mp_layer(x, adj, [torch.empty(E, 0)], layer.export_weights())
    layer = layer.cuda()
    cadj = to_cuda_adj(adj)
    ops.clear_plan_cache()
    with torch.no_grad():
        got = layer(x.cuda(), cadj, None, {}, {}, empty_feats(cadj, "cuda")).cpu()
    err = float((got - want).abs().max())
    assert err <= TOL, f"max |delta| = {err:.3e}"
    assert torch.isfinite(got).all()


def test_config3_graph2class_stack_vs_oracle():
    """BASELINE config 3 shape (T0=8 -> T=17, Typilus GGNN arch, max) at a reduced node count the
    CPU oracle finishes in seconds; the full size is covered by the property test below."""
    from oracle import mp_oracle as O
    from ptgnn_amd import layers as L, workloads
    from ptgnn_amd.gnn import GraphNeuralNetwork
    H = 128
    mb = workloads.batched_graphs(6, 1500, 8, 2.2, seed=7)
    N = mb["num_nodes"]
    torch.manual_seed(3)
    ggnn = L.GatedMessagePassingLayer(H, H, 17, "max")
    r1 = L.ConcatResidualLayer(H)
    last = L.GatedMessagePassingLayer(2 * H, H, 17, "max")
    mods = [r1.pass_through_dummy_layer()] + [ggnn] * 7 + [r1, last]
    specs = ([{"kind": "residual_origin", "name": "r1"}] + [ggnn.export_weights()] * 7
             + [{"kind": "residual_concat", "name": "r1"}, last.export_weights()])
    x = workloads.node_states(N, H, seed=5)
    want, n_edges = O.gnn_forward(x, mb["adjacency_lists"], specs, True, True)
    net = GraphNeuralNetwork(mods, torch.nn.Identity(), True, True).cuda().eval()
    with torch.no_grad():
        out = net(node_data={"input": x.cuda()}, adjacency_lists=to_cuda_adj(mb["adjacency_lists"]),
                  edge_feature_data=[], node_to_graph_idx=mb["node_to_graph_idx"].cuda(),
                  reference_node_ids={k: v.cuda() for k, v in mb["reference_node_ids"].items()},
                  reference_node_graph_idx={k: v.cuda() for k, v in mb["reference_node_graph_idx"].items()},
                  num_graphs=mb["num_graphs"])
    got = out.output_node_representations.cpu()
    assert net.report_metrics()["num_edges"] == n_edges
    err = float((got - want).abs().max())
    assert err <= TOL, f"max |delta| after 8 GGNN layers = {err:.3e}"
    np.testing.assert_array_equal(out.node_idx_references["supernodes"].cpu().numpy(),
                                  mb["reference_node_ids"]["supernodes"].numpy())


def test_config1_ppi_ggnn_full_size_vs_oracle():
    """BASELINE config 1 (the reference's CPU-runnable case): PPI-like batch, 24 graphs x ~2.4k nodes,
    ~14 raw links per node, one raw edge type -> T = 3 with reverse + self edges, 1 GGNN layer H = 64,
    sum -- at FULL size against the CPU oracle, through the container."""
    from oracle import mp_oracle as O
    from ptgnn_amd import layers as L, workloads
    from ptgnn_amd.gnn import GraphNeuralNetwork
    H = 64
    mb = workloads.batched_graphs(24, 2400, 1, 14.0, seed=1234)
    N = mb["num_nodes"]
    torch.manual_seed(1234)
    layer = L.GatedMessagePassingLayer(H, H, 3, "sum")
    x = workloads.node_states(N, H, seed=1234)
    want, n_edges = O.gnn_forward(x, mb["adjacency_lists"], [layer.export_weights()], True, True)
    assert n_edges > 1_500_000
    net = GraphNeuralNetwork([layer], torch.nn.Identity(), True, True).cuda().eval()
    with torch.no_grad():
        out = net(node_data={"input": x.cuda()}, adjacency_lists=to_cuda_adj(mb["adjacency_lists"]),
                  edge_feature_data=[], node_to_graph_idx=mb["node_to_graph_idx"].cuda(),
                  reference_node_ids={}, reference_node_graph_idx={}, num_graphs=mb["num_graphs"])
    assert net.report_metrics()["num_edges"] == n_edges
    err = float((out.output_node_representations.cpu() - want).abs().max())
    assert err <= TOL, f"max |delta| = {err:.3e}"


def test_config4_varmisuse_mlp_stack_vs_oracle():
    """BASELINE config 4 shape: VarMisuse MLP-MP architecture (varmisuse/train.py:42-74: 8 MLP-MP layers,
    hidden 64, max, concat + mean residuals), T0 = 10 raw types -> T = 21, at a node count the CPU oracle
    finishes in seconds."""
    from oracle import mp_oracle as O
    from ptgnn_amd import layers as L, workloads
    from ptgnn_amd.gnn import GraphNeuralNetwork
    H, T = 64, 21
    mb = workloads.batched_graphs(8, 2000, 10, 2.4, seed=21)
    N = mb["num_nodes"]
    torch.manual_seed(4)
    mk = lambda: L.MlpMessagePassingLayer(H, H, H, T, "max", dropout_rate=0.1)          # noqa: E731
    mk2 = lambda: L.MlpMessagePassingLayer(2 * H, H, 2 * H, T, "max", dropout_rate=0.1)  # noqa: E731
    r1, r2 = L.ConcatResidualLayer(H), L.MeanResidualLayer(H)
    r3, r4 = L.ConcatResidualLayer(H), L.MeanResidualLayer(H)
    mods, names = [], []

    def add(m, spec):
        mods.append(m)
        names.append(spec)
    add(r1.pass_through_dummy_layer(), {"kind": "residual_origin", "name": "r1"})
    for _ in range(3):
        m = mk(); add(m, None)
    add(r1, {"kind": "residual_concat", "name": "r1"})
    m = mk2(); add(m, None)
    add(r2.pass_through_dummy_layer(), {"kind": "residual_origin", "name": "r2"})
    for _ in range(2):
        m = mk(); add(m, None)
    add(r2, {"kind": "residual_mean", "name": "r2"})
    add(r3.pass_through_dummy_layer(), {"kind": "residual_origin", "name": "r3"})
    m = mk(); add(m, None)
    add(r3, {"kind": "residual_concat", "name": "r3"})
    m = mk2(); add(m, None)
    specs = [sp if sp is not None else md.export_weights() for md, sp in zip(mods, names)]
    assert sum(1 for sp in specs if sp["kind"] == "mlp") == 8
    x = workloads.node_states(N, H, seed=6)
    want, n_edges = O.gnn_forward(x, mb["adjacency_lists"], specs, True, True)
    net = GraphNeuralNetwork(mods, torch.nn.Identity(), True, True).cuda().eval()
    with torch.no_grad():
        out = net(node_data={"input": x.cuda()}, adjacency_lists=to_cuda_adj(mb["adjacency_lists"]),
                  edge_feature_data=[], node_to_graph_idx=mb["node_to_graph_idx"].cuda(),
                  reference_node_ids={k: v.cuda() for k, v in mb["reference_node_ids"].items()},
                  reference_node_graph_idx={k: v.cuda() for k, v in mb["reference_node_graph_idx"].items()},
                  num_graphs=mb["num_graphs"])
    assert net.report_metrics()["num_edges"] == n_edges
    got = out.output_node_representations.cpu()
    # 8 stacked LayerNorm layers amplify fp32 rounding: attribute the error against a float64 evaluation
    # of the same stack -- the HIP path may be no further from exact arithmetic than the reference's own
    # fp32 arithmetic is (and within the stated 1e-5 per layer of it)
    exact, _ = O.gnn_forward(x.double(), mb["adjacency_lists"], [O.cast_spec(sp, torch.float64) for sp in specs],
                             True, True)
    err_ref = float((want.double() - exact).abs().max())
    err_ours = float((got.double() - exact).abs().max())
    err = float((got - want).abs().max())
    assert err_ours <= max(TOL, 2.0 * err_ref), f"vs fp64: ours {err_ours:.3e}, fp32 oracle {err_ref:.3e}"
    assert err <= 8 * TOL, f"max |delta| after 8 MLP-MP layers = {err:.3e} (ours vs fp64 {err_ours:.3e}, " \
                           f"oracle vs fp64 {err_ref:.3e})"
    print(f"cfg4: ours-vs-oracle {err:.2e}, ours-vs-fp64 {err_ours:.2e}, oracle-vs-fp64 {err_ref:.2e}")
    # the container hands the layer in front of a concat residual the right half of the residual's result to write
    # into (GRU of a GGNN layer, the dense update of an MLP-MP layer): same bits as the module-by-module loop, whose
    # residuals torch.cat -- and no torch.cat inside the container's loop
    from ptgnn_amd import ops
    cadj = to_cuda_adj(mb["adjacency_lists"])
    cadj = cadj + [(d, s_) for s_, d in cadj]
    ar = torch.arange(N, device="cuda")
    cadj.append((ar, ar))
    feats = [None] * len(cadj)
    n2g = mb["node_to_graph_idx"].cuda()
    with torch.no_grad(), L.forward_scope():
        h = x.cuda()
        for m in net.message_passing_layers:
            h = m(h, cadj, n2g, {}, {}, feats)
    cats = []
    real_cat = torch.cat
    try:
        def counting_cat(tensors, *a, **k):     # (weight stacks may be cat-ed; node-state matrices may not)
            if len(tensors) and tensors[0].dim() == 2 and tensors[0].shape[0] == N:
                cats.append(1)
            return real_cat(tensors, *a, **k)
        torch.cat = counting_cat
        ops.clear_plan_cache()
        with torch.no_grad():
            via_container = net.gnn(x.cuda(), cadj, feats, n2g, {}, {})
    finally:
        torch.cat = real_cat
    assert torch.equal(via_container, h) and torch.equal(via_container.cpu(), got)
    assert not cats, "a concat residual fell back to torch.cat inside the container"


@pytest.mark.parametrize("kind", ["ggnn", "mlp"])
@pytest.mark.parametrize("agg", ["sum", "max"])
@pytest.mark.parametrize("gemm_mode", ["stream", "tile"])
def test_config5_scaled_layer_vs_oracle(kind, agg, gemm_mode):
    """BASELINE config 5's own layer configuration -- ONE GGNN / ONE MLP-MP layer at H = M = 256 over a power-law
    graph (Zipf-0.8 destinations through a node permutation, uniform sources) -- scaled to a size the
    CPU oracle finishes in seconds (N = 125 k, E = 1.25 M: ~20 hub rows above the 2048-edge threshold, the largest
    ~24 k in-edges), against oracle/mp_oracle.py (gatedmessagepassing.py:37-69, mlpmessagepassing.py:68-117):
    the K = 512 GRU, the 256 -> 256 pre-transform, hub rows inside a full layer, streaming and tile kernels."""
    from oracle import mp_oracle as O
    from ptgnn_amd import layers as L, ops, workloads
    N, E, H = 125_000, 1_250_000, 256
    adj = workloads.power_law_graph(N, E, alpha=0.8, seed=1234)
    deg = torch.bincount(adj[0][1], minlength=N)
    hubs = deg > ops.HUB_THRESHOLD
    assert int(hubs.sum()) >= 2
    x = workloads.node_states(N, H, seed=2)
    torch.manual_seed(5)
    layer = (L.GatedMessagePassingLayer(H, H, 1, agg) if kind == "ggnn"
             else L.MlpMessagePassingLayer(H, H, H, 1, agg)).eval()
    spec = layer.export_weights()
    feats = [torch.empty(E, 0)]
    fn = O.ggnn_layer if kind == "ggnn" else O.mlp_mp_layer
    with torch.no_grad():
        want = fn(x, adj, feats, spec)
    layer = layer.cuda()
    cadj = to_cuda_adj(adj)
    prev = ops.set_gemm_mode(gemm_mode)
    try:
        ops.clear_plan_cache()
        with torch.no_grad():
            got = layer(x.cuda(), cadj, None, {}, {}, empty_feats(cadj, "cuda")).cpu()
    finally:
        ops.set_gemm_mode(prev)
    err = (got - want).abs()
    if not (kind == "ggnn" and agg == "sum"):
        # max aggregation and the LayerNorm-ed MLP-MP update: 1e-5 on EVERY row, hub rows included (measured 3e-6 ..
        # 7e-6; hub rows fold chunk-wise, not in the reference's serial order -- the update absorbs it)
        assert float(err.max()) <= TOL, f"max |delta| = {float(err.max()):.3e} (hub rows {float(err[hubs].max()):.3e})"
        return
    # GGNN with SUM aggregation feeds an un-normalised sum of up to 26 k messages into the GRU: fp32 itself is not
    # 1e-5-accurate there -- the reference's own fp32 arithmetic (the oracle) sits 5e-5 (rows of 512-2048 in-edges)
    # to 2e-4 (hub rows) from a float64 evaluation of the same layer (scripts/cfg5_parity_diag.py).  So: the literal
    # 1e-5 against the oracle where fp32 supports it (in-degree < 32: 96 % of the rows), and everywhere the HIP
    # path must be as close to float64 as the reference's fp32 is (factor 2 of slack), bucket by bucket.
    low = deg < 32
    assert float(err[low].max()) <= TOL, f"rows with < 32 in-edges: max |delta| = {float(err[low].max()):.3e}"
    with torch.no_grad():
        w64 = fn(x.double(), adj, [f.double() for f in feats], O.cast_spec(spec, torch.float64))
    ours64, ref64 = (got.double() - w64).abs(), (want.double() - w64).abs()
    for lo, hi in ((32, 128), (128, 512), (512, 2049), (2049, 10 ** 9)):
        m = (deg >= lo) & (deg < hi)
        if int(m.sum()):
            ours, ref = float(ours64[m].max()), float(ref64[m].max())
            assert ours <= max(TOL, 2.0 * ref), f"in-degree [{lo}, {hi}): ours-vs-fp64 {ours:.3e}, oracle-fp32-vs-fp64 {ref:.3e}"


def test_config5_powerlaw_shard_full_size_properties():
    """BASELINE config 5 per-GPU shard at FULL size (1.25 M nodes, 12.5 M edges, H = 256, Zipf 0.8
    destinations): no CPU oracle at this size, so size-independent properties --
    (a) checksum of checksums: the column sums of the sum-aggregate equal the column sums of the
        gathered source rows (every edge contributes exactly once, hub rows included);
    (b) mean == sum / max(in-degree, 1) row for row;
    (c) max >= mean elementwise on non-empty rows, and empty rows are exactly 0 for both."""
    from ptgnn_amd import ops, workloads
    N, E, M = 1_250_000, 12_500_000, 256
    adj = workloads.power_law_graph(N, E, alpha=0.8, seed=1234)
    cadj = to_cuda_adj(adj)
    y = workloads.node_states(N, M, seed=2).cuda()
    ops.clear_plan_cache()
    plan = ops.plan_for(cadj, N)
    s = ops.gather_reduce(y, plan, M, "sum")
    src = cadj[0][0]
    want_cols = torch.zeros(M, dtype=torch.float64, device="cuda")
    for lo in range(0, E, 2_000_000):                      # bounded scratch: 2 M x 256 fp32 = 2 GB
        want_cols += y.index_select(0, src[lo: lo + 2_000_000]).sum(0, dtype=torch.float64)
    got_cols = s.sum(0, dtype=torch.float64)
    np.testing.assert_allclose(got_cols.cpu().numpy(), want_cols.cpu().numpy(), rtol=1e-6, atol=1e-1)
    deg = (plan.rowptr[1:] - plan.rowptr[:-1]).to(torch.float32)
    assert int(deg.max()) > ops.HUB_THRESHOLD
    mean = ops.gather_reduce(y, plan, M, "mean")
    ref_mean = s / deg.clamp(min=1).unsqueeze(1)
    assert float((mean - ref_mean).abs().max()) <= 1e-5 * max(1.0, float(ref_mean.abs().max()))
    mx = ops.gather_reduce(y, plan, M, "max")
    nonempty = deg > 0
    assert bool((mx[nonempty] >= mean[nonempty] - 1e-5).all())
    assert float(mx[~nonempty].abs().max()) == 0.0 and float(s[~nonempty].abs().max()) == 0.0


def test_full_size_properties_linearity_and_permutation():
    """Size-independent properties at config-2 size (no CPU oracle needed):
       (a) sum aggregation is linear in the message table: agg(Y1 + Y2) == agg(Y1) + agg(Y2) to fp32
           rounding, and agg(2*Y) == 2*agg(Y) exactly;
       (b) max aggregation is invariant to permuting the edge list (bit exact);
       (c) column sums: sum_v agg_sum(Y)[v] == sum_e Y[src_e] (checksum of checksums, fp64)."""
    from ptgnn_amd import ops, workloads
    N, E, M = 200_000, 1_100_000, 128
    adj = to_cuda_adj(workloads.random_graph(N, E))
    y = workloads.node_states(N, M, seed=9).cuda()
    plan = ops.build_plan(adj, N)
    a1 = ops.gather_reduce(y, plan, M, "sum")
    np.testing.assert_array_equal(ops.gather_reduce(2 * y, plan, M, "sum").cpu().numpy(), (2 * a1).cpu().numpy())
    y2 = workloads.node_states(N, M, seed=10).cuda()
    a12 = ops.gather_reduce(y + y2, plan, M, "sum")
    a2 = ops.gather_reduce(y2, plan, M, "sum")
    assert float((a12 - (a1 + a2)).abs().max()) < 1e-4
    total = a1.double().sum(0).cpu()
    want = y.double().index_select(0, adj[0][0]).sum(0).cpu()
    np.testing.assert_allclose(total.numpy(), want.numpy(), rtol=1e-6, atol=1e-2)  # fp32 row sums
    perm = torch.randperm(E, device="cuda")
    plan_p = ops.build_plan([(adj[0][0][perm], adj[0][1][perm])], N)
    np.testing.assert_array_equal(ops.gather_reduce(y, plan, M, "max").cpu().numpy(),
                                  ops.gather_reduce(y, plan_p, M, "max").cpu().numpy())
    # empty segments are exactly 0 in every mode
    deg = (plan.rowptr[1:] - plan.rowptr[:-1])
    empty = (deg == 0).nonzero().flatten()
    assert empty.numel() > 0
    for r in ("sum", "mean", "max", "min"):
        assert float(ops.gather_reduce(y, plan, M, r)[empty].abs().sum()) == 0.0


# ------------------------------------------------------------------------------------------------
# dst-range sharding on ONE GPU: virtual ranks (collectives replaced by direct copies) so the HIP
# kernels see real halo tables (source rows beyond the local node range)
# ------------------------------------------------------------------------------------------------
def _virtual_shards(adj, n, world):
    from ptgnn_amd import sharded
    indeg = torch.zeros(n, dtype=torch.int64)
    for _, d in adj:
        indeg += torch.bincount(d, minlength=n)
    ranges = sharded.balanced_node_ranges(indeg, world)
    return ranges, _shards_for(adj, ranges)


def _shards_for(adj, ranges):
    from ptgnn_amd import sharded
    shards = []
    for p, (lo, hi) in enumerate(ranges):
        mine = [(s[(d >= lo) & (d < hi)].cuda(), d[(d >= lo) & (d < hi)].cuda()) for s, d in adj]
        shards.append(sharded.ShardedGraph.build_local(mine, ranges, p))
    return shards


def _emulate_exchange(shard, global_rows):
    """What the all-to-all delivers: rows `need_ids` of the GLOBAL row matrix (single-process stand-in)."""
    def exchange_into(table):
        table[shard.n_local:] = global_rows().index_select(0, shard.need_ids)
        return table
    shard.exchange_into = exchange_into
    shard.exchange = lambda rows, sh=shard: sh.exchange_into(
        torch.cat([rows, rows.new_empty(sh.n_halo, rows.shape[1])]))


@pytest.mark.parametrize("kind,agg", [("mlp", "sum"), ("mlp", "max"), ("ggnn", "max"), ("ggnn", "mean")])
def test_sharded_layer_equals_unsharded(kind, agg):
    from ptgnn_amd import layers as L, ops
    n, H, world = 3000, 64, 3
    g = torch.Generator().manual_seed(17)
    T = 1 if kind == "mlp" else 3          # T*M <= H ships message rows; T*M > H ships node states
    adj = [(torch.randint(0, n, (c,), generator=g), torch.randint(0, n, (c,), generator=g))
           for c in ([16000] if T == 1 else [9000, 0, 4000])]
    x = torch.randn(n, H, generator=g).cuda()
    torch.manual_seed(5)
    layer = (L.MlpMessagePassingLayer(H, H, H, T, agg) if kind == "mlp"
             else L.GatedMessagePassingLayer(H, H, T, agg)).cuda().eval()
    ops.clear_plan_cache()
    cadj = to_cuda_adj(adj)
    with torch.no_grad():
        want = layer(x, cadj, None, {}, {}, empty_feats(cadj, "cuda"))
    ranges, shards = _virtual_shards(adj, n, world)
    assert all(s.n_halo > 0 for s in shards)

    def make_exchange(shard):
        def exchange_into(table):
            # what the all-to-all delivers: rows `need_ids` of the GLOBAL row matrix, which here is
            # the concatenation of every virtual rank's own block
            table[shard.n_local:] = shard._global_rows.index_select(0, shard.need_ids)
            return table
        return exchange_into

    # message-table rows depend on the layer; emulate by computing each rank's own block first
    outs = []
    with torch.no_grad():
        for shard in shards:
            lo, hi = shard.lo, shard.hi
            # global matrix of whatever rows get exchanged: message rows if T*M <= H else node states
            if kind == "mlp":
                w = layer._stacked_edge_weights()[: T * H]
                shard._global_rows = ops.linear(x, w) if T * H <= H else x
            else:
                w = layer._stacked_edge_weights()
                shard._global_rows = ops.linear(x, w) if T * H <= H else x
            shard.exchange_into = make_exchange(shard)
            shard.exchange = lambda rows, sh=shard: sh.exchange_into(
                torch.cat([rows, rows.new_empty(sh.n_halo, rows.shape[1])]))
            outs.append(layer.forward_sharded(x[lo:hi].contiguous(), shard))
    got = torch.cat(outs)
    # each destination row is reduced on one rank in the unsharded order -> identical results
    np.testing.assert_array_equal(got.cpu().numpy(), want.cpu().numpy())


# ------------------------------------------------------------------------------------------------
# grouped per-edge GEMM (many sparse edge types)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("use_dst", [False, True])
@pytest.mark.parametrize("H,M", [(32, 64), (128, 128), (64, 200)])
def test_edge_linear_matches_reference_order(use_dst, H, M):
    from ptgnn_amd import ops
    g = torch.Generator().manual_seed(H + M)
    n = 700
    counts = [1000, 0, 129, 1, 128, 513]
    adj = [(torch.randint(0, n, (c,), generator=g), torch.randint(0, n, (c,), generator=g)) for c in counts]
    x = torch.randn(n, H, generator=g)
    K = 2 * H if use_dst else H
    ws = [torch.randn(M, K, generator=g) / K ** 0.5 for _ in counts]
    want = torch.cat([(torch.cat([x[s], x[d]], -1) if use_dst else x[s]).double() @ w.double().t()
                      for (s, d), w in zip(adj, ws)]).float()
    got = ops.edge_linear(x.cuda(), to_cuda_adj(adj), [w.cuda() for w in ws], use_dst).cpu()
    assert got.shape == want.shape
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=0, atol=TOL)


@pytest.mark.parametrize("kind", ["ggnn", "mlp", "mlp_notarget"])
@pytest.mark.parametrize("agg", ["sum", "max"])
def test_edge_path_equals_pretransform_path_and_oracle(kind, agg, monkeypatch):
    from oracle import mp_oracle as O
    from ptgnn_amd import layers as L, ops, workloads
    mb = workloads.batched_graphs(5, 400, 6, 2.2, seed=3)
    N, H = mb["num_nodes"], 64
    adj = O.augment_adjacency(mb["adjacency_lists"], N, True, True)     # T = 13, E ~ 5.4 N
    T = len(adj)
    torch.manual_seed(9)
    if kind == "ggnn":
        layer = L.GatedMessagePassingLayer(H, H, T, agg)
    else:
        layer = L.MlpMessagePassingLayer(H, H, H, T, agg, use_target_state_as_message_input=kind == "mlp")
    x = workloads.node_states(N, H, seed=4)
    feats = [torch.empty(a[0].shape[0], 0) for a in adj]
    want = (O.ggnn_layer if kind == "ggnn" else O.mlp_mp_layer)(x, adj, feats, layer.export_weights())
    layer = layer.cuda().eval()
    cadj = to_cuda_adj(adj)
    outs = {}
    for name, bias in (("edge", 1e-9), ("node", 1e9)):
        monkeypatch.setattr(L, "EDGE_PATH_BIAS", bias)
        ops.clear_plan_cache()
        timer = ops.KernelTimer()
        ops.set_kernel_timer(timer)
        with torch.no_grad():
            outs[name] = layer(x.cuda(), cadj, None, {}, {}, empty_feats(cadj, "cuda")).cpu()
        ops.set_kernel_timer(None)
        used = set(timer.summary())
        assert ("edge_linear" in used) == (name == "edge"), used
    np.testing.assert_allclose(outs["edge"].numpy(), want.numpy(), rtol=0, atol=TOL)
    np.testing.assert_allclose(outs["node"].numpy(), want.numpy(), rtol=0, atol=TOL)


@pytest.mark.parametrize("n,counts,src_pool", [
    (5000, [40000, 0, 7, 12000, 1], 300),          # heavy sharing: 300 distinct sources per type, an empty type
    (70001, [90000, 30000], None),                 # sources over the whole node range (few duplicates)
    (33, [1000], 33),                              # one edge type (type_bits = 0), every node a source many times
])
def test_unique_sources_equal_the_numpy_bookkeeping(n, counts, src_pool, monkeypatch):
    """ptgnn_amd_unique_sources (integer work, bit-exact): per edge type the sorted distinct source ids, for every
    CSR slot the row of its (type, source) pair in the type-major list of pairs, and the device-resident launch table
    of the shared-row edge GEMM (row / unit prefixes; every workgroup of the apportioning owns work)."""
    from ptgnn_amd import ops
    g = torch.Generator().manual_seed(n + len(counts))
    adj = [(torch.randint(0, src_pool or n, (c,), generator=g), torch.randint(0, n, (c,), generator=g)) for c in counts]
    monkeypatch.setattr(ops, "UNIQUE_MIN_EDGES", 0)
    monkeypatch.setattr(ops, "_UNIQ_SKIP", [0])
    ops.clear_plan_cache()
    plan = ops.plan_for(to_cuda_adj(adj), n)
    uq = plan.unique_messages()
    assert uq is not None and plan.unique_messages() is uq
    want = [np.unique(a[0].numpy()) for a in adj]
    off = np.cumsum([0] + [len(w) for w in want])
    assert uq.rows(wait=True) == off[-1]
    np.testing.assert_array_equal(uq.counts.cpu().numpy(), [len(w) for w in want] + [off[-1]])
    for t, (got, _) in enumerate(uq.adjacency()):
        np.testing.assert_array_equal(got.cpu().numpy(), want[t])
    col = plan.col[: plan.num_edges].cpu().numpy().astype(np.int64)
    typ, src = col & ((1 << plan.type_bits) - 1), col >> plan.type_bits
    rows = np.array([off[t] + np.searchsorted(want[t], s_) for t, s_ in zip(typ, src)], dtype=np.int64)
    np.testing.assert_array_equal(uq.slot_row[: plan.num_edges].cpu().numpy().astype(np.int64), rows)
    # the launch table (stream_gemm.h StreamEdgeTable: 3 x 64 pointers, then int64 edge_off[65], int32 unit_off[65],
    # int32 wg_off[65], int32 num_types)
    raw = uq.edge_table.cpu().numpy().tobytes()
    T = len(counts)
    edge_off = np.frombuffer(raw, np.int64, 65, 3 * 64 * 8)[: T + 1]
    unit_off = np.frombuffer(raw, np.int32, 65, 3 * 64 * 8 + 65 * 8)[: T + 1]
    wg_off = np.frombuffer(raw, np.int32, 65, 3 * 64 * 8 + 65 * 8 + 65 * 4)[: T + 1]
    num_types = np.frombuffer(raw, np.int32, 1, 3 * 64 * 8 + 65 * 8 + 2 * 65 * 4)[0]
    assert num_types == T
    np.testing.assert_array_equal(edge_off, off)
    units = np.array([(len(w) + 31) // 32 for w in want])
    np.testing.assert_array_equal(unit_off, np.cumsum([0] + list(units)))
    wgs = np.diff(wg_off)
    assert ((wgs > 0) == (units > 0)).all() and (wgs <= np.maximum(units, 0)).all() and wg_off[-1] <= 256


@pytest.mark.parametrize("mode", ["stream"])   # the shared-row launch exists on the streaming kernels only
@pytest.mark.parametrize("agg", ["sum", "max", "mean"])
def test_ggnn_layer_with_shared_message_rows_gives_the_same_bits(agg, mode, monkeypatch):
    """GGNN inference, edge form: one message row per distinct (edge type, source) pair (GraphPlan.unique_messages)
    instead of one per edge.  A row is the same fmaf chain wherever it is computed and the aggregation folds the
    in-edges in the same CSR order, so the layer output does not change by a bit; it stays within 1e-5 of the
    oracle (gatedmessagepassing.py:37-69)."""
    from oracle import mp_oracle as O
    from ptgnn_amd import layers as L, ops, workloads
    mb = workloads.batched_graphs(6, 700, 5, 3.0, seed=11)
    N, H = mb["num_nodes"], 64
    adj = O.augment_adjacency(mb["adjacency_lists"], N, True, True)
    # make sources repeat inside a type, as the out-edges of one AST / token node do
    adj = [(s - s % 3 if i % 2 == 0 else s, d) for i, (s, d) in enumerate(adj)]
    T = len(adj)
    torch.manual_seed(21)
    layer = L.GatedMessagePassingLayer(H, H, T, agg)
    x = workloads.node_states(N, H, seed=5)
    want = O.ggnn_layer(x, adj, [torch.empty(a[0].shape[0], 0) for a in adj], layer.export_weights())
    layer = layer.cuda().eval()
    cadj = to_cuda_adj(adj)
    monkeypatch.setattr(L, "EDGE_PATH_BIAS", 1e-9)
    monkeypatch.setattr(ops, "UNIQUE_MIN_EDGES", 0)
    monkeypatch.setattr(ops, "_UNIQ_SKIP", [0])
    prev_mode = ops.set_gemm_mode(mode)
    outs, rows = {}, {}
    try:
        for name in ("per_edge", "shared"):
            monkeypatch.setattr(L, "UNIQUE_MESSAGES", name == "shared")
            ops.clear_plan_cache()
            if name == "shared":
                uq = ops.plan_for(cadj, N).unique_messages()
                assert uq is not None and uq.rows(wait=True) < 0.9 * sum(int(a[0].shape[0]) for a in adj)
            timer = ops.KernelTimer()
            ops.set_kernel_timer(timer)
            with torch.no_grad():
                outs[name] = layer(x.cuda(), cadj, None, {}, {}, empty_feats(cadj, "cuda")).cpu()
            ops.set_kernel_timer(None)
            rows[name] = timer.summary()["edge_linear_shared" if name == "shared" else "edge_linear"]["flops"]
    finally:
        ops.set_gemm_mode(prev_mode)
    assert rows["shared"] < 0.9 * rows["per_edge"]
    assert torch.equal(outs["shared"], outs["per_edge"])
    np.testing.assert_allclose(outs["shared"].numpy(), want.numpy(), rtol=0, atol=TOL)


@pytest.mark.parametrize("use_dst", [False, True])
@pytest.mark.parametrize("H,M", [(32, 64), (128, 128), (64, 200), (160, 36)])
def test_edge_weight_grad_matches_fp64(use_dst, H, M):
    """dW_t = d_msg_t^T . [x[src] ; x[dst]] for all types in one call (incl. empty / 1-edge / multi-chunk
    types) against a float64 evaluation; deterministic run to run."""
    from ptgnn_amd import ops
    g = torch.Generator().manual_seed(3 * H + M)
    n = 900
    counts = [5000, 0, 129, 1, 128, 777]
    adj = [(torch.randint(0, n, (c,), generator=g), torch.randint(0, n, (c,), generator=g)) for c in counts]
    x = torch.randn(n, H, generator=g)
    gm = torch.randn(sum(counts), M, generator=g)
    off = np.cumsum([0] + counts)
    want = torch.stack([gm[off[t]:off[t + 1]].double().t()
                        @ (torch.cat([x[s], x[d]], -1) if use_dst else x[s]).double()
                        for t, (s, d) in enumerate(adj)]).float()
    cadj = to_cuda_adj(adj)
    got = ops.edge_weight_grad(x.cuda(), cadj, gm.cuda(), use_dst)
    again = ops.edge_weight_grad(x.cuda(), cadj, gm.cuda(), use_dst)
    assert torch.equal(got, again)
    scale = max(1.0, float(want.abs().max()))
    # relative to the size of the sums (up to 5000-term fp32 dot products)
    assert float((got.cpu() - want).abs().max()) <= 1e-5 * scale


@pytest.mark.parametrize("use_dst", [False, True])
@pytest.mark.parametrize("H,M,counts", [
    (128, 128, [150000, 0, 70001, 3, 40000]),      # 128 x 128 tiles, long unmasked runs + ragged tails, odd row counts
    (256, 128, [60000, 1, 20000]),                 # two k-tiles (four with the target half)
    (96, 96, [30000, 5000]),                       # 32-wide blocks, 3 x 3 (3 x 6) tiles per group
    (64, 64, [17, 2500, 0, 64, 129]),              # 64-wide blocks, ranges shorter than one prefetch block
    (64, 64, [90000, 30001]),                      # with the target half: ONE 128-wide k-tile across the [src ; dst] seam
    (96, 64, [50000, 7]),                          # K = 192 with the target half: the seam inside the middle 64-wide k-tile
])
def test_streaming_weight_grad_shapes_match_fp64(use_dst, H, M, counts):
    """Shapes the streaming weight-gradient kernel takes (widths that are multiples of 32, wgrad_stream.hip): every
    tile shape, ranges long enough for the unmasked loop and short enough for the masked one only, empty and
    one-row types, odd counts (the half-filled last MFMA step).  float64 reference; bit-identical run to run."""
    from ptgnn_amd import ops
    g = torch.Generator().manual_seed(H + 7 * M + len(counts))
    n = 4000
    adj = [(torch.randint(0, n, (c,), generator=g), torch.randint(0, n, (c,), generator=g)) for c in counts]
    x = torch.randn(n, H, generator=g)
    gm = torch.randn(sum(counts), M, generator=g)
    off = np.cumsum([0] + counts)
    want = torch.stack([gm[off[t]:off[t + 1]].double().t()
                        @ (torch.cat([x[s], x[d]], -1) if use_dst else x[s]).double()
                        for t, (s, d) in enumerate(adj)])
    cadj = to_cuda_adj(adj)
    got = ops.edge_weight_grad(x.cuda(), cadj, gm.cuda(), use_dst)
    assert torch.equal(got, ops.edge_weight_grad(x.cuda(), cadj, gm.cuda(), use_dst))
    # fp32 sums of up to 150 k products: the bound scales with sqrt(count) * |terms|
    for t, c in enumerate(counts):
        tol = 1e-5 * max(1.0, float(want[t].abs().max())) + 2e-6 * np.sqrt(max(c, 1))
        assert float((got[t].cpu().double() - want[t]).abs().max()) <= tol, (t, c)


@pytest.mark.parametrize("p", [0.1, 0.5])
def test_edge_linear_hash_dropout_matches_numpy_restatement(p):
    """The in-kernel dropout mask (forward, input-gradient and weight-gradient forms) equals the numpy
    restatement in tests/helpers.py element for element."""
    from helpers import dropout_keep_scale
    from ptgnn_amd import ops
    g = torch.Generator().manual_seed(77)
    n, H, M, seed = 500, 64, 96, 0x1234_5678_9ABC_DEF
    counts = [700, 0, 130, 1]
    E = sum(counts)
    adj = [(torch.randint(0, n, (c,), generator=g), torch.randint(0, n, (c,), generator=g)) for c in counts]
    x = torch.randn(n, H, generator=g)
    ws = [torch.randn(M, H, generator=g) / H ** 0.5 for _ in counts]
    mask = dropout_keep_scale(seed, E, H, p)
    frac = float((mask == 0).float().mean())
    assert abs(frac - p) < 0.02
    xin = torch.cat([x[s] for s, _ in adj]) * mask                       # Dropout(x_src), message order
    off = np.cumsum([0] + counts)
    want_fwd = torch.cat([xin[off[t]:off[t + 1]].double() @ ws[t].double().t() for t in range(len(counts))]).float()
    cadj, cws = to_cuda_adj(adj), [w.cuda() for w in ws]
    got_fwd = ops.edge_linear(x.cuda(), cadj, cws, False, dropout=(1, p, seed)).cpu()
    np.testing.assert_allclose(got_fwd.numpy(), want_fwd.numpy(), rtol=0, atol=TOL)
    # input-gradient form: (d_msg . W_t) * mask over an identity index
    gm = torch.randn(E, M, generator=g)
    want_gin = torch.cat([gm[off[t]:off[t + 1]].double() @ ws[t].double() for t in range(len(counts))]).float() * mask
    ident = torch.arange(E).cuda()
    iadj = [(ident[off[t]:off[t + 1]], ident[off[t]:off[t + 1]]) for t in range(len(counts))]
    got_gin = ops.edge_linear(gm.cuda(), iadj, [w.t().contiguous().cuda() for w in ws], False,
                              dropout=(2, p, seed)).cpu()
    np.testing.assert_allclose(got_gin.numpy(), want_gin.numpy(), rtol=0, atol=TOL)
    # weight-gradient form
    want_gw = torch.stack([gm[off[t]:off[t + 1]].double().t() @ xin[off[t]:off[t + 1]].double()
                           for t in range(len(counts))]).float()
    got_gw = ops.edge_weight_grad(x.cuda(), cadj, gm.cuda(), False, p, seed).cpu()
    np.testing.assert_allclose(got_gw.numpy(), want_gw.numpy(), rtol=0, atol=1e-5 * max(1.0, float(want_gw.abs().max())))


@pytest.mark.parametrize("agg", ["sum", "max"])
def test_ggnn_training_with_per_edge_dropout_matches_oracle_autograd(agg, monkeypatch):
    """The reference's shipped training configuration (dropout on the gathered message input,
    gatedmessagepassing.py:57-61) on the HIP edge path: forward, d x, d W_t and d GRU against the
    oracle's torch-CPU autograd with the SAME (restated) mask."""
    from helpers import dropout_keep_scale
    from oracle import mp_oracle as O
    from ptgnn_amd import layers as L, ops, workloads
    mb = workloads.batched_graphs(3, 150, 3, 2.2, seed=5)
    N, H, M, p, seed = mb["num_nodes"], 32, 64, 0.2, 987654321987
    adj = O.augment_adjacency(mb["adjacency_lists"], N, True, True)
    T = len(adj)
    torch.manual_seed(3)
    layer = L.GatedMessagePassingLayer(H, M, T, agg, dropout_rate=p).train()
    monkeypatch.setattr(L, "_dropout_seed", lambda: seed)
    x = workloads.node_states(N, H, seed=6)
    gout = workloads.node_states(N, H, seed=7)
    spec = layer.export_weights()
    E = sum(int(a[0].shape[0]) for a in adj)
    mask = dropout_keep_scale(seed, E, H, p)

    xo = x.clone().requires_grad_(True)
    ws = [w.clone().requires_grad_(True) for w in spec["edge_w"]]
    gru = [spec[k].clone().requires_grad_(True) for k in ("w_ih", "w_hh", "b_ih", "b_hh")]
    off = np.cumsum([0] + [int(a[0].shape[0]) for a in adj])
    msgs = torch.cat([O.linear(xo[s] * mask[off[t]:off[t + 1]], ws[t]) for t, (s, _) in enumerate(adj)])
    agg_o = O.aggregate_messages(msgs, torch.cat([d for _, d in adj]), N, agg)
    yo = O.gru_cell(agg_o, xo, *gru)
    yo.backward(gout)

    layer = layer.cuda()
    xg = x.cuda().requires_grad_(True)
    cadj = to_cuda_adj(adj)
    ops.clear_plan_cache()
    timer = ops.KernelTimer()
    ops.set_kernel_timer(timer)
    yg = layer(xg, cadj, None, {}, {}, empty_feats(cadj, "cuda"))
    yg.backward(gout.cuda())
    ops.set_kernel_timer(None)
    used = timer.summary()
    assert used["edge_linear"]["calls"] == 2 and used["edge_weight_grad"]["calls"] == 1 \
        and used["segment_spread"]["calls"] == 1, used
    np.testing.assert_allclose(yg.detach().cpu().numpy(), yo.detach().numpy(), rtol=0, atol=TOL)
    sc = max(1.0, float(xo.grad.abs().max()))
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xo.grad.numpy(), rtol=0, atol=2e-5 * sc)
    sd = layer.state_dict(keep_vars=True)
    for t in range(T):
        ours = sd[f"_GatedMessagePassingLayer__edge_message_transformation_layers.{t}.weight"].grad
        sc = max(1.0, float(ws[t].grad.abs().max()))
        np.testing.assert_allclose(ours.cpu().numpy(), ws[t].grad.numpy(), rtol=0, atol=2e-5 * sc)
    ours = sd["_GatedMessagePassingLayer__state_update.weight_ih"].grad
    np.testing.assert_allclose(ours.cpu().numpy(), gru[0].grad.numpy(), rtol=0,
                               atol=2e-5 * max(1.0, float(gru[0].grad.abs().max())))


# ------------------------------------------------------------------------------------------------
# training: fused aggregation forward + backward (HIP kernel both ways) vs oracle autograd on CPU
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("path", ["table", "edge"])
@pytest.mark.parametrize("kind", ["ggnn", "mlp", "mlp_notarget"])
@pytest.mark.parametrize("agg", ["sum", "mean", "max", "min"])
def test_training_gradients_match_oracle_autograd(kind, agg, path, monkeypatch):
    from oracle import mp_oracle as O
    from ptgnn_amd import layers as L, ops, workloads
    monkeypatch.setattr(L, "EDGE_PATH_BIAS", 1e-9 if path == "edge" else 1e9)
    mb = workloads.batched_graphs(3, 150, 3, 2.2, seed=11)
    N, H, M = mb["num_nodes"], 32, (64 if path == "edge" else 48)
    adj = O.augment_adjacency(mb["adjacency_lists"], N, True, True)     # T = 7
    T = len(adj)
    torch.manual_seed(21)
    if kind == "ggnn":
        layer = L.GatedMessagePassingLayer(H, M, T, agg)
    else:
        layer = L.MlpMessagePassingLayer(H, 40, M, T, agg, use_target_state_as_message_input=kind == "mlp")
    layer.train()                                   # dropout p = 0: training-mode fused aggregation
    x = workloads.node_states(N, H, seed=12)
    gout = workloads.node_states(N, layer.output_state_dimension, seed=13)

    # oracle: same math in torch-CPU autograd
    spec = layer.export_weights()

    def req(v):
        if isinstance(v, torch.Tensor) and v.is_floating_point():
            return v.clone().requires_grad_(True)
        if isinstance(v, list):
            return [req(u) for u in v]
        return v
    spec_g = {k: req(v) for k, v in spec.items()}
    xo = x.clone().requires_grad_(True)
    feats = [torch.empty(a[0].shape[0], 0) for a in adj]
    yo = (O.ggnn_layer if kind == "ggnn" else O.mlp_mp_layer)(xo, adj, feats, spec_g)
    yo.backward(gout)

    layer = layer.cuda()
    xg = x.cuda().requires_grad_(True)
    cadj = to_cuda_adj(adj)
    ops.clear_plan_cache()
    timer = ops.KernelTimer()
    ops.set_kernel_timer(timer)
    yg = layer(xg, cadj, None, {}, {}, empty_feats(cadj, "cuda"))
    yg.backward(gout.cuda())
    ops.set_kernel_timer(None)
    calls = timer.summary()
    n_agg = calls["gather_reduce"]["calls"] + calls.get("gather_reduce_masked", {"calls": 0})["calls"]
    assert n_agg >= 2 and calls["csr_build"]["calls"] == 2       # HIP kernel both ways; fwd + bwd plan
    assert ("edge_weight_grad" in calls) == (path == "edge"), calls
    np.testing.assert_allclose(yg.detach().cpu().numpy(), yo.detach().numpy(), rtol=0, atol=TOL)
    scale = max(1.0, float(xo.grad.abs().max()))
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xo.grad.numpy(), rtol=0, atol=2e-5 * scale)
    # parameter gradients
    if kind == "ggnn":
        pairs = [(layer.state_dict(keep_vars=True)[f"_GatedMessagePassingLayer__edge_message_transformation_layers.{t}.weight"],
                  spec_g["edge_w"][t]) for t in range(T)]
        pairs.append((layer.state_dict(keep_vars=True)["_GatedMessagePassingLayer__state_update.weight_ih"], spec_g["w_ih"]))
    else:
        sd = layer.state_dict(keep_vars=True)
        pairs = [(sd[f"_MlpMessagePassingLayer__edge_message_transformation_layers.{t}._MLP__mlp_modules.1.weight"],
                  spec_g["edge_mlp"][t][0]) for t in range(T)]
        pairs.append((sd["_MlpMessagePassingLayer__state_update.1.weight"], spec_g["dense_w"]))
    for ours, ref in pairs:
        s = max(1.0, float(ref.grad.abs().max()))
        np.testing.assert_allclose(ours.grad.cpu().numpy(), ref.grad.numpy(), rtol=0, atol=2e-5 * s)


# ------------------------------------------------------------------------------------------------
# hub rows (power-law graphs, BASELINE config 5 shape scaled to one GPU)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("reduce,dim,with_dst", [("sum", 64, False), ("max", 64, False), ("sum", 256, False),
                                                 ("mean", 128, False), ("sum", 128, True), ("max", 256, True)])
def test_long_rows_launch_keeps_the_serial_fold_order(reduce, dim, with_dst):
    """Plans of >= 2 M edges fold rows of 257 .. 2048 in-edges in a launch of their own (k_long_rows, on the library's
    side stream, 16 slots per round trip) -- in the SAME slot order: sums over rows up to the hub threshold stay
    bit-identical to the CPU scatter, max with its arg exact."""
    from oracle import scatter_ref
    from ptgnn_amd import ops, workloads
    N, E = 150_000, 2_200_000
    adj = workloads.power_law_graph(N, E, alpha=0.9, seed=13)
    src, dst = adj[0]
    deg = torch.bincount(dst, minlength=N)
    assert int(((deg > 256) & (deg <= ops.HUB_THRESHOLD)).sum()) >= 20 and int((deg > ops.HUB_THRESHOLD).sum()) >= 2
    y = workloads.node_states(N, dim, seed=4)
    yd = workloads.node_states(N, dim, seed=6) if with_dst else None     # one edge type: the MLP-MP destination term
    plan = ops.build_plan(to_cuda_adj(adj), N)
    res = ops.gather_reduce(y.cuda(), plan, dim, reduce, ydst=yd.cuda() if with_dst else None,
                            return_arg=reduce == "max")
    got = (res[0] if isinstance(res, tuple) else res).cpu()
    msgs = y[src] + yd[dst] if with_dst else y[src]
    want = scatter_ref.scatter(msgs, dst, dim=0, dim_size=N, reduce=reduce)
    small = deg <= ops.HUB_THRESHOLD
    if reduce == "max":
        np.testing.assert_array_equal(got.numpy(), want.numpy())
        want_arg = scatter_ref.scatter_max(msgs, dst, 0, dim_size=N)[1]
        arg = res[1].cpu().long()
        perm = plan.perm[:E].cpu().long()
        got_edge = torch.where(arg >= 0, perm[arg.clamp(min=0)], torch.full_like(arg, E))
        np.testing.assert_array_equal(got_edge.numpy(), want_arg.numpy())
    elif reduce == "sum":
        np.testing.assert_array_equal(got[small].numpy(), want[small].numpy())
    else:
        np.testing.assert_allclose(got[small].numpy(), want[small].numpy(), rtol=0, atol=1e-6)


@pytest.mark.parametrize("reduce", ["sum", "mean", "max", "min"])
@pytest.mark.parametrize("with_dst", [False, True])
def test_hub_rows_match_oracle(reduce, with_dst):
    """A power-law destination distribution: a few rows with 10^4..10^5 in-edges take the chunked hub
    path.  max/min (and their arg) are exact; sums differ from the serial fold only by fp32 rounding,
    so they are checked against an fp64 evaluation with a tolerance relative to the row's mass."""
    from oracle import scatter_ref
    from ptgnn_amd import ops, workloads
    N, E, M = 20_000, 400_000, 64
    adj = workloads.power_law_graph(N, E, alpha=1.1, seed=3)
    src, dst = adj[0]
    deg = torch.bincount(dst, minlength=N)
    assert int(deg.max()) > 3 * ops.HUB_THRESHOLD and int((deg > ops.HUB_THRESHOLD).sum()) >= 3
    y = workloads.node_states(N, M, seed=4)
    yd = workloads.node_states(N, M, seed=5) if with_dst else None
    plan = ops.build_plan(to_cuda_adj(adj), N)
    assert plan.may_have_hubs()
    res = ops.gather_reduce(y.cuda(), plan, M, reduce, ydst=yd.cuda() if with_dst else None,
                            return_arg=reduce in ("max", "min"))
    got = (res[0] if isinstance(res, tuple) else res).cpu()
    msgs = y[src] + (yd[dst] if with_dst else 0)
    if reduce in ("max", "min"):
        want = scatter_ref.scatter(msgs, dst, dim=0, dim_size=N, reduce=reduce)
        np.testing.assert_array_equal(got.numpy(), want.numpy())
        # arg: the winning CSR slot's edge is the FIRST edge (in message order) attaining the optimum
        fn = scatter_ref.scatter_max if reduce == "max" else scatter_ref.scatter_min
        want_arg = fn(msgs, dst, 0, dim_size=N)[1]
        arg = res[1].cpu().long()
        perm = plan.perm[:E].cpu().long()
        got_edge = torch.where(arg >= 0, perm[arg.clamp(min=0)], torch.full_like(arg, E))
        np.testing.assert_array_equal(got_edge.numpy(), want_arg.numpy())
    else:
        want64 = scatter_ref.scatter(msgs.double(), dst, dim=0, dim_size=N, reduce=reduce)
        mass = scatter_ref.scatter(msgs.double().abs(), dst, dim=0, dim_size=N, reduce=reduce)
        err = (got.double() - want64).abs()
        assert float((err / (1.0 + mass)).max()) < 5e-6      # fp32 accumulation over up to 10^5 terms
        small = deg <= ops.HUB_THRESHOLD                       # non-hub rows: still the reference's order
        want32 = scatter_ref.scatter(msgs, dst, dim=0, dim_size=N, reduce=reduce)
        if reduce == "sum":
            np.testing.assert_array_equal(got[small].numpy(), want32[small].numpy())


def test_hub_path_is_deterministic_and_skipped_when_not_needed():
    from ptgnn_amd import ops, workloads
    N, E, M = 20_000, 400_000, 128
    adj = to_cuda_adj(workloads.power_law_graph(N, E, alpha=1.1, seed=8))
    y = workloads.node_states(N, M, seed=9).cuda()
    plan = ops.build_plan(adj, N)
    a = ops.gather_reduce(y, plan, M, "sum", epilogue=ops.EPI_GELU_LAYERNORM,
                          ln_weight=torch.ones(M, device="cuda"), ln_bias=torch.zeros(M, device="cuda"))
    b = ops.gather_reduce(y, plan, M, "sum", epilogue=ops.EPI_GELU_LAYERNORM,
                          ln_weight=torch.ones(M, device="cuda"), ln_bias=torch.zeros(M, device="cuda"))
    np.testing.assert_array_equal(a.cpu().numpy(), b.cpu().numpy())
    assert torch.isfinite(a).all()
    # a uniform random graph of the same size has no hubs: once the async max-degree read-back has
    # landed the hub launches are skipped
    small = ops.build_plan(to_cuda_adj(workloads.random_graph(100, 1000, seed=2)), 100)
    assert not small.may_have_hubs()            # cannot contain a hub: no chunk workgroups, no workspace
    # the ticket counters are left zero by every launch
    assert int(plan.hub_tickets(M).abs().sum()) == 0


# ------------------------------------------------------------------------------------------------
# VarMisuse GGNN stack with global graph exchange
# ------------------------------------------------------------------------------------------------
def test_varmisuse_ggnn_stack_with_global_exchange_matches_reference_golden():
    from oracle.fixtures import unpack_adj, unpack_specs
    from ptgnn_amd.gnn import GraphNeuralNetwork
    g = load_golden("gnn_stack_ggnn_varmisuse_global")
    adj, specs = unpack_adj(g), unpack_specs(g)
    net = GraphNeuralNetwork(stack_from_specs(specs), torch.nn.Identity(), introduce_backwards_edges=True,
                             add_self_edges=True).cuda().eval()
    x = torch.from_numpy(g["x"]).cuda()
    n2g = torch.from_numpy(g["node_to_graph_idx"]).cuda()
    with torch.no_grad():
        out = net(node_data={"input": x}, adjacency_lists=to_cuda_adj(adj), edge_feature_data=[],
                  node_to_graph_idx=n2g, reference_node_ids={}, reference_node_graph_idx={}, num_graphs=3)
    np.testing.assert_allclose(out.output_node_representations.cpu().numpy(), g["y"], rtol=0, atol=TOL)


@pytest.mark.parametrize("pool", ["sum", "mean", "max", "weighted"])
def test_pooling_over_sorted_index_matches_oracle(pool):
    from oracle import scatter_ref
    from ptgnn_amd import reduceops as R
    g = torch.Generator().manual_seed(2)
    sizes = torch.tensor([5000, 1, 0, 9000, 300])          # includes an empty graph and a long one
    idx = torch.repeat_interleave(torch.arange(5), sizes)
    x = torch.randn(int(sizes.sum()), 64, generator=g)
    e = R.ElementsToSummaryRepresentationInput(x.cuda(), idx.cuda(), 5)
    if pool == "weighted":
        mod = R.WeightedSumVarSizedElementReduce(64).cuda()
        w = mod.state_dict()["_WeightedSumVarSizedElementReduce__weights_layer.weight"].cpu()
        want = scatter_ref.scatter((x * torch.sigmoid(x @ w.t())).double(), idx, dim=0, dim_size=5, reduce="sum")
    else:
        mod = R.SimpleVarSizedElementReduce(pool).cuda()
        want = scatter_ref.scatter(x.double(), idx, dim=0, dim_size=5, reduce=pool)
    with torch.no_grad():
        got = mod(e).cpu()
    if pool == "max":
        np.testing.assert_array_equal(got.numpy(), want.float().numpy())
    else:
        np.testing.assert_allclose(got.double().numpy(), want.numpy(), rtol=2e-5, atol=2e-4)
    assert float(got[2].abs().sum()) == 0.0                # empty graph -> 0


# ------------------------------------------------------------------------------------------------
# differentiable dense blocks (training): Linear / GRUCell on the HIP kernels vs torch-CPU autograd
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,k,n_out", [(5000, 128, 384), (777, 64, 36), (1, 32, 4), (0, 32, 8)])
def test_linear_weight_grad_matches_fp64(rows, k, n_out):
    from ptgnn_amd import ops
    g = torch.Generator().manual_seed(rows + k)
    x, gy = torch.randn(rows, k, generator=g), torch.randn(rows, n_out, generator=g)
    want = (gy.double().t() @ x.double()).float()
    got, got_b = ops.linear_weight_grad(x.cuda(), gy.cuda(), want_bias=True)
    assert torch.equal(got, ops.linear_weight_grad(x.cuda(), gy.cuda()))
    assert float((got.cpu() - want).abs().max()) <= 1e-5 * max(1.0, float(want.abs().max()))
    want_b = gy.double().sum(0).float()
    assert float((got_b.cpu() - want_b).abs().max()) <= 1e-5 * max(1.0, float(want_b.abs().max()))


def test_dense_linear_and_gru_autograd_match_torch_cpu():
    from ptgnn_amd import dense
    torch.manual_seed(5)
    n, m, h = 3000, 64, 96
    cell = torch.nn.GRUCell(m, h)
    lin = torch.nn.Linear(h, 40)
    a, hx, gout = torch.randn(n, m), torch.randn(n, h), torch.randn(n, 40)
    a1, h1 = a.clone().requires_grad_(True), hx.clone().requires_grad_(True)
    lin(cell(a1, h1)).backward(gout)
    want = [a1.grad, h1.grad] + [p.grad.clone() for p in list(cell.parameters()) + list(lin.parameters())]
    cell.zero_grad(); lin.zero_grad()
    cell, lin = cell.cuda(), lin.cuda()
    a2, h2 = a.cuda().requires_grad_(True), hx.cuda().requires_grad_(True)
    y = dense.linear(dense.gru_cell(cell, a2, h2), lin.weight, lin.bias)
    y.backward(gout.cuda())
    got = [a2.grad, h2.grad] + [p.grad for p in list(cell.parameters()) + list(lin.parameters())]
    for g_, w_ in zip(got, want):
        np.testing.assert_allclose(g_.cpu().numpy(), w_.numpy(), rtol=0,
                                   atol=2e-5 * max(1.0, float(w_.abs().max())))


# ------------------------------------------------------------------------------------------------
# device-side minibatch assembly (graphneuralnetwork.py:386-493) -- integer work, bit-exact
# ------------------------------------------------------------------------------------------------
def test_minibatch_builder_matches_reference_golden_bit_exact():
    from helpers import golden_batcher_graphs, replay_minibatches
    from ptgnn_amd.batching import MinibatchBuilder
    g = load_golden("batcher")
    T0, graphs = golden_batcher_graphs(g)
    mbs = replay_minibatches(MinibatchBuilder, T0, graphs, int(g["stop_after"]), lambda b: b.finalize("cuda"))
    assert len(mbs) == int(g["num_minibatches"])
    for bi, mb in enumerate(mbs):
        assert mb["num_graphs"] == int(g[f"mb{bi}.num_graphs"])
        assert mb["node_to_graph_idx"].dtype == torch.int64 and mb["node_to_graph_idx"].is_cuda
        np.testing.assert_array_equal(mb["node_to_graph_idx"].cpu().numpy(), g[f"mb{bi}.node_to_graph_idx"])
        for t in range(T0):
            s, d = mb["adjacency_lists"][t]
            assert s.dtype == torch.int64 and d.dtype == torch.int64
            np.testing.assert_array_equal(s.cpu().numpy(), g[f"mb{bi}.adj.{t}.src"])
            np.testing.assert_array_equal(d.cpu().numpy(), g[f"mb{bi}.adj.{t}.dst"])
        for k in mb["reference_node_ids"]:
            np.testing.assert_array_equal(mb["reference_node_ids"][k].cpu().numpy(), g[f"mb{bi}.ref_ids.{k}"])
            np.testing.assert_array_equal(mb["reference_node_graph_idx"][k].cpu().numpy(),
                                          g[f"mb{bi}.ref_gidx.{k}"])


def test_minibatch_builder_large_batch_equals_oracle_and_feeds_the_layers():
    """Graph2Class-sized batch (48 graphs, T0 = 8): device assembly == the oracle's restatement of the
    reference batcher bit for bit, and its tensors drive a layer directly."""
    from oracle import mp_oracle as O
    from ptgnn_amd import layers as L
    from ptgnn_amd.batching import MinibatchBuilder
    rng = np.random.RandomState(7)
    T0, graphs = 8, []
    for _ in range(48):
        n = int(rng.randint(1500, 3500))
        adj = []
        for t in range(T0):
            e = int(rng.randint(0, 2 * n)) if t != 5 else 0          # one edge type with no edges at all
            adj.append((rng.randint(0, n, e).astype(np.int32), rng.randint(0, n, e).astype(np.int32)))
        graphs.append({"num_nodes": n, "adjacency_lists": adj,
                       "reference_nodes": {"supernodes": rng.randint(0, n, 20).astype(np.int32)}})
    total = sum(gr["num_nodes"] for gr in graphs)
    (want,) = list(O.batch_graphs(graphs, T0, total))         # the node budget is hit by the last graph
    b = MinibatchBuilder(T0, total)
    for i, gr in enumerate(graphs):
        assert b.extend(gr["adjacency_lists"], gr["num_nodes"], gr["reference_nodes"]) == (i < 47)
    mb = b.finalize("cuda")
    assert mb["num_graphs"] == want["num_graphs"] == 48
    np.testing.assert_array_equal(mb["node_to_graph_idx"].cpu().numpy(), want["node_to_graph_idx"].numpy())
    for t in range(T0):
        for side in (0, 1):
            np.testing.assert_array_equal(mb["adjacency_lists"][t][side].cpu().numpy(),
                                          want["adjacency_lists"][t][side].numpy())
    for k in ("reference_node_ids", "reference_node_graph_idx"):
        np.testing.assert_array_equal(mb[k]["supernodes"].cpu().numpy(), want[k]["supernodes"].numpy())
    off = np.concatenate([[0], np.cumsum([gr["num_nodes"] for gr in graphs])])
    N = int(off[-1])
    layer = L.GatedMessagePassingLayer(32, 32, T0, "sum").cuda().eval()
    with torch.no_grad():
        y = layer(torch.randn(N, 32, device="cuda"), mb["adjacency_lists"], mb["node_to_graph_idx"], {}, {},
                  empty_feats(mb["adjacency_lists"], "cuda"))
    assert tuple(y.shape) == (N, 32) and bool(torch.isfinite(y).all())


# ------------------------------------------------------------------------------------------------
# sharded TRAINING (table form + differentiable halo exchange) on the RCCL path at world = 1
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind", ["ggnn", "mlp"])
def test_sharded_training_equals_unsharded_training_world1(kind, monkeypatch):
    import socket
    import torch.distributed as dist
    from ptgnn_amd import layers as L, ops, sharded
    monkeypatch.setattr(L, "EDGE_PATH_BIAS", 1e9)         # both runs take the table form
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        n, H, T = 2000, 64, 3
        g = torch.Generator().manual_seed(23)
        adj = [(torch.randint(0, n, (c,), generator=g).cuda(), torch.randint(0, n, (c,), generator=g).cuda())
               for c in (5000, 0, 2500)]
        x = torch.randn(n, H, generator=g).cuda()
        gout = torch.randn(n, H, generator=g).cuda()
        torch.manual_seed(4)
        layer = (L.GatedMessagePassingLayer(H, H, T, "max") if kind == "ggnn"
                 else L.MlpMessagePassingLayer(H, H, H, T, "sum")).cuda().train()
        grads = []
        for mode in ("plain", "sharded"):
            layer.zero_grad()
            xi = x.clone().requires_grad_(True)
            ops.clear_plan_cache()
            if mode == "plain":
                y = layer(xi, adj, None, {}, {}, empty_feats(adj, "cuda"))
            else:
                shard = sharded.ShardedGraph.build(adj, (0, n))
                y = layer.forward_sharded(xi, shard)
            y.backward(gout)
            grads.append([y.detach(), xi.grad] + [p.grad.clone() for p in layer.parameters()])
        assert torch.equal(grads[0][0], grads[1][0])       # forward: same kernels, same order
        for a, b in zip(*grads):   # the MLP layer's sharded form splits the src / dst GEMMs: fp32 re-association
            sc = max(1.0, float(a.abs().max()))
            assert float((a - b).abs().max()) <= 1e-5 * sc
    finally:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------
# HIP graph capture: the whole layer stack (plan build included) is capturable -- no host sync,
# no allocation outside torch's pools -- and replays bit-identically on new node states
# ------------------------------------------------------------------------------------------------
def test_forward_is_hip_graph_capturable():
    from ptgnn_amd import layers as L, ops, workloads
    from ptgnn_amd.gnn import GraphNeuralNetwork
    mb = workloads.batched_graphs(3, 900, 2, 7.0, seed=9)          # a PPI-sized minibatch (~3k nodes)
    N, H, T = mb["num_nodes"], 64, 5
    torch.manual_seed(1)
    net = GraphNeuralNetwork([L.GatedMessagePassingLayer(H, H, T, "sum"),
                              L.MlpMessagePassingLayer(H, H, H, T, "max")], torch.nn.Identity(),
                             True, True).cuda().eval()
    adj = to_cuda_adj(mb["adjacency_lists"])
    n2g = mb["node_to_graph_idx"].cuda()
    x_static = workloads.node_states(N, H, seed=1).cuda()

    def fwd():
        ops.clear_plan_cache()
        return net(node_data={"input": x_static}, adjacency_lists=adj, edge_feature_data=[],
                   node_to_graph_idx=n2g, reference_node_ids={}, reference_node_graph_idx={},
                   num_graphs=mb["num_graphs"]).output_node_representations

    with torch.no_grad():
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                fwd()                                               # warm-up outside capture
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            y_static = fwd()
        for seed in (2, 3):
            x_new = workloads.node_states(N, H, seed=seed).cuda()
            x_static.copy_(x_new)
            graph.replay()
            got = y_static.clone()
            want = fwd()
            assert torch.equal(got, want)


@pytest.mark.parametrize("pool", ["weighted_sum", "max"])
@pytest.mark.parametrize("sizes", [[700, 1, 1300, 64], [6000, 3, 4500]], ids=["small_graphs", "hub_graphs"])
def test_global_exchange_training_gradients_match_oracle_autograd(pool, sizes):
    """GruGlobalStateUpdate (globalgraphexchange.py:29-64) in training mode: pooling, broadcast-gather and GRU
    cell all on the HIP autograd nodes; gradients w.r.t. the node states and the GRU weights vs the oracle.
    `hub_graphs`: graphs of more than 4096 nodes are hub rows of the sort-free pooling plan
    (`plan_from_sorted_index` + the chunk-parallel hub kernel), forward and backward."""
    from oracle import mp_oracle as O
    from ptgnn_amd import reduceops as R, workloads
    n2g = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
    N, H = int(n2g.shape[0]), 64
    torch.manual_seed(8)
    mod = R.GruGlobalStateUpdate(R.WeightedSumVarSizedElementReduce(H) if pool == "weighted_sum"
                                 else R.SimpleVarSizedElementReduce(pool), H, H).train()
    sd = mod.state_dict()
    spec = {"pool": pool, "w_ih": sd["_GruGlobalStateUpdate__gru_cell.weight_ih"],
            "w_hh": sd["_GruGlobalStateUpdate__gru_cell.weight_hh"],
            "b_ih": sd["_GruGlobalStateUpdate__gru_cell.bias_ih"],
            "b_hh": sd["_GruGlobalStateUpdate__gru_cell.bias_hh"]}
    if pool == "weighted_sum":
        spec["pool_w"] = [v for k, v in sd.items() if k.endswith("weights_layer.weight")][0]
    spec = {k: (v.clone().requires_grad_(True) if isinstance(v, torch.Tensor) else v) for k, v in spec.items()}
    x = workloads.node_states(N, H, seed=3)
    gout = workloads.node_states(N, H, seed=4)
    xo = x.clone().requires_grad_(True)
    yo = O.global_gru_exchange(xo, n2g, spec)
    yo.backward(gout)
    mod = mod.cuda()
    xg = x.cuda().requires_grad_(True)
    yg = mod(xg, [], n2g.cuda(), {}, {}, [])
    yg.backward(gout.cuda())
    if max(sizes) > 4096 or pool == "weighted_sum":
        # a hub row (a graph of thousands of nodes) is folded chunk-wise, not in the reference's serial order, and a
        # 6000-term fp32 sum of O(1) values carries ~1e-3 of rounding in EITHER order before it enters the GRU: attribute
        # against a float64 evaluation -- the HIP path may sit no further from it than the fp32 oracle does (x2).  The
        # weighted-sum pool folds 128-row chunks of every graph (csrc/weighted_pool.hip), whatever its size: same rule
        # (1 300 fp32 rows: 1.2e-5 from the oracle's serial order on one element, round 6)
        x64 = x.double().requires_grad_(True)
        y64 = O.global_gru_exchange(x64, n2g, {k: (v.detach().double() if isinstance(v, torch.Tensor) else v) for k, v in spec.items()})
        y64.backward(gout.double())
        e_ref, e_ours = float((yo.detach().double() - y64.detach()).abs().max()), float((yg.detach().cpu().double() - y64.detach()).abs().max())
        assert e_ours <= max(TOL, 2.0 * e_ref), f"forward vs fp64: ours {e_ours:.3e}, fp32 oracle {e_ref:.3e}"
        g_ref, g_ours = float((xo.grad.double() - x64.grad).abs().max()), float((xg.grad.cpu().double() - x64.grad).abs().max())
        assert g_ours <= max(2e-5 * max(1.0, float(x64.grad.abs().max())), 2.0 * g_ref), f"d x vs fp64: ours {g_ours:.3e}, oracle {g_ref:.3e}"
        return
    np.testing.assert_allclose(yg.detach().cpu().numpy(), yo.detach().numpy(), rtol=0, atol=TOL)
    sc = max(1.0, float(xo.grad.abs().max()))
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xo.grad.numpy(), rtol=0, atol=2e-5 * sc)
    got_w = mod.state_dict(keep_vars=True)["_GruGlobalStateUpdate__gru_cell.weight_ih"].grad
    sc = max(1.0, float(spec["w_ih"].grad.abs().max()))
    np.testing.assert_allclose(got_w.cpu().numpy(), spec["w_ih"].grad.numpy(), rtol=0, atol=2e-5 * sc)


# ------------------------------------------------------------------------------------------------
# backward pinned to the REFERENCE's own gradients (tests/golden/train_*.npz)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("path", ["edge", "table"])
@pytest.mark.parametrize("name", ["train_ggnn_max", "train_ggnn_sum", "train_mlp_sum_target",
                                  "train_mlp_max_notarget", "train_ggnn_max_w64", "train_mlp_sum_target_w64",
                                  "train_ggnn_sum_w128"])
def test_training_gradients_match_reference_golden(name, path, monkeypatch):
    """The HIP training paths (edge form and table form) reproduce the output, d x and every parameter
    gradient the reference's own layer produced under torch autograd on CPU."""
    from oracle.fixtures import unpack_adj, unpack_specs
    from ptgnn_amd import layers as L, ops
    monkeypatch.setattr(L, "EDGE_PATH_BIAS", 1e-9 if path == "edge" else 1e9)
    g = load_golden(name)
    adj, (spec,) = unpack_adj(g), unpack_specs(g)
    layer = layer_from_spec(spec).cuda().train()
    x = torch.from_numpy(g["x"]).cuda().requires_grad_(True)
    cadj = to_cuda_adj(adj)
    ops.clear_plan_cache()
    timer = ops.KernelTimer()
    ops.set_kernel_timer(timer)
    y = layer(x, cadj, None, {}, {}, empty_feats(cadj, "cuda"))
    y.backward(torch.from_numpy(g["gout"]).cuda())
    ops.set_kernel_timer(None)
    assert ("edge_weight_grad" in timer.summary()) == (path == "edge")
    np.testing.assert_allclose(y.detach().cpu().numpy(), g["y"], rtol=0, atol=TOL)
    np.testing.assert_allclose(x.grad.cpu().numpy(), g["g.x"], rtol=0, atol=2e-5 * max(1.0, np.abs(g["g.x"]).max()))
    grads = dict(layer.named_parameters())
    checked = 0
    for key in g.files:
        if key.startswith("g.") and key != "g.x":
            want = g[key]
            got = grads[key[2:]].grad
            assert got is not None, key
            np.testing.assert_allclose(got.cpu().numpy(), want, rtol=0, atol=2e-5 * max(1.0, np.abs(want).max()),
                                       err_msg=key)
            checked += 1
    assert checked >= 8


# ------------------------------------------------------------------------------------------------
# round 2: the benchmarked configs at the size they are benchmarked, per-layer bars, container branches
# ------------------------------------------------------------------------------------------------
def _typilus_ggnn_stack(H, T, seed):
    from ptgnn_amd import layers as L
    torch.manual_seed(seed)
    ggnn = L.GatedMessagePassingLayer(H, H, T, "max")
    r1 = L.ConcatResidualLayer(H)
    last = L.GatedMessagePassingLayer(2 * H, H, T, "max")
    mods = [r1.pass_through_dummy_layer()] + [ggnn] * 7 + [r1, last]
    specs = ([{"kind": "residual_origin", "name": "r1"}] + [ggnn.export_weights()] * 7
             + [{"kind": "residual_concat", "name": "r1"}, last.export_weights()])
    return mods, specs


def _run_container(net, x, mb, **kw):
    return net(node_data={"input": x.cuda()}, adjacency_lists=to_cuda_adj(mb["adjacency_lists"]),
               edge_feature_data=kw.pop("edge_feature_data", []), node_to_graph_idx=mb["node_to_graph_idx"].cuda(),
               reference_node_ids={k: v.cuda() for k, v in mb["reference_node_ids"].items()},
               reference_node_graph_idx={k: v.cuda() for k, v in mb["reference_node_graph_idx"].items()},
               num_graphs=mb["num_graphs"], **kw)


@pytest.mark.parametrize("gemm_mode", ["stream", "tile"])
def test_config3_full_size_vs_oracle(gemm_mode):
    """BASELINE config 3 at the size bench.py measures it: 48 graphs / 115 772 nodes / T = 17 / 625 130 edges,
    the Typilus GGNN stack (8 GGNN layers, hidden 128, max) -- end to end against the CPU oracle, 1e-5, in
    every GEMM mode (the oracle forward takes ~15 s on the host cores)."""
    from oracle import mp_oracle as O
    from ptgnn_amd import ops, workloads
    from ptgnn_amd.gnn import GraphNeuralNetwork
    H = 128
    mb = workloads.batched_graphs(48, 2500, 8, 2.2, seed=1234)
    N = mb["num_nodes"]
    mods, specs = _typilus_ggnn_stack(H, 17, 1234)
    x = workloads.node_states(N, H, seed=5)
    key = "cfg3_full_oracle"
    if key not in _CACHE:
        _CACHE[key] = O.gnn_forward(x, mb["adjacency_lists"], specs, True, True)
    want, n_edges = _CACHE[key]
    net = GraphNeuralNetwork(mods, torch.nn.Identity(), True, True).cuda().eval()
    prev = ops.set_gemm_mode(gemm_mode)
    try:
        with torch.no_grad():
            out = _run_container(net, x, mb)
    finally:
        ops.set_gemm_mode(prev)
    assert N > 110_000 and net.report_metrics()["num_edges"] == n_edges
    err = float((out.output_node_representations.cpu() - want).abs().max())
    print(f"cfg3 full size [{gemm_mode}]: max|delta| vs oracle = {err:.2e}")
    assert err <= TOL, f"[{gemm_mode}] max |delta| after 8 GGNN layers at N={N}: {err:.3e}"
    np.testing.assert_array_equal(out.node_idx_references["supernodes"].cpu().numpy(),
                                  mb["reference_node_ids"]["supernodes"].numpy())


_CACHE = {}


def _varmisuse_mlp_stack(H, T, seed):
    """varmisuse/train.py:42-74 -- [origin r1, MLP x3, r1 (concat), MLP(2H -> H, M = 2H), origin r2, MLP x2,
    r2 (mean), origin r3, MLP, r3 (concat), MLP(2H -> H)]: 8 MLP-MP layers, max, dropout 0.1 (eval)."""
    from ptgnn_amd import layers as L
    torch.manual_seed(seed)
    mk = lambda: L.MlpMessagePassingLayer(H, H, H, T, "max", dropout_rate=0.1)          # noqa: E731
    mk2 = lambda: L.MlpMessagePassingLayer(2 * H, H, 2 * H, T, "max", dropout_rate=0.1)  # noqa: E731
    r1, r2 = L.ConcatResidualLayer(H), L.MeanResidualLayer(H)
    r3 = L.ConcatResidualLayer(H)
    mods, specs = [], []

    def add(m, spec=None):
        mods.append(m)
        specs.append(spec if spec is not None else m.export_weights())
    add(r1.pass_through_dummy_layer(), {"kind": "residual_origin", "name": "r1"})
    for _ in range(3):
        add(mk())
    add(r1, {"kind": "residual_concat", "name": "r1"})
    add(mk2())
    add(r2.pass_through_dummy_layer(), {"kind": "residual_origin", "name": "r2"})
    for _ in range(2):
        add(mk())
    add(r2, {"kind": "residual_mean", "name": "r2"})
    add(r3.pass_through_dummy_layer(), {"kind": "residual_origin", "name": "r3"})
    add(mk())
    add(r3, {"kind": "residual_concat", "name": "r3"})
    add(mk2())
    assert sum(1 for sp in specs if sp["kind"] == "mlp") == 8
    return mods, specs


@pytest.mark.parametrize("gemm_mode", ["stream", "tile"])
def test_config4_varmisuse_full_size_per_layer_and_end_to_end(gemm_mode):
    """BASELINE config 4 at the reference's batch cap (varmisuse/train.py:119: 80 000 nodes): 40 graphs x ~2000
    nodes, T0 = 10 -> T = 21, the 8-layer VarMisuse MLP-MP stack at hidden 64.
      (a) PER LAYER at the stated bar: every MLP-MP layer, fed the ORACLE's input of that layer, is within
          1e-5 of the oracle's output of that layer;
      (b) end to end: 8 stacked LayerNorms amplify fp32 rounding, so the whole-stack delta is attributed
          against a float64 evaluation -- the HIP path may sit no further from exact arithmetic than the
          reference's own fp32 arithmetic does (x2), and the number is printed for DESIGN.md section 6."""
    from oracle import mp_oracle as O
    from ptgnn_amd import ops, workloads
    from ptgnn_amd.gnn import GraphNeuralNetwork
    H, T = 64, 21
    mb = workloads.batched_graphs(40, 2000, 10, 2.4, seed=21)
    N = mb["num_nodes"]
    assert 70_000 < N <= 90_000
    mods, specs = _varmisuse_mlp_stack(H, T, 4)
    x = workloads.node_states(N, H, seed=6)
    key = "cfg4_full_oracle"
    if key not in _CACHE:
        trace = []
        want, n_edges = O.gnn_forward(x, mb["adjacency_lists"], specs, True, True, trace=trace)
        exact, _ = O.gnn_forward(x.double(), mb["adjacency_lists"],
                                 [O.cast_spec(sp, torch.float64) for sp in specs], True, True)
        _CACHE[key] = (want, n_edges, trace, exact)
    want, n_edges, trace, exact = _CACHE[key]
    adj = O.augment_adjacency(mb["adjacency_lists"], N, True, True)
    cadj = to_cuda_adj(adj)
    feats = empty_feats(cadj, "cuda")
    prev = ops.set_gemm_mode(gemm_mode)
    try:
        worst = 0.0
        with torch.no_grad():
            for mod, spec, (x_in, x_out) in zip(mods, specs, trace):
                if spec["kind"] != "mlp":
                    continue
                ops.clear_plan_cache()
                got = mod.cuda().eval()(x_in.cuda(), cadj, None, {}, {}, feats).cpu()
                err = float((got - x_out).abs().max())
                worst = max(worst, err)
                assert err <= TOL, f"[{gemm_mode}] per-layer |delta| = {err:.3e} (layer fed the oracle's input)"
            net = GraphNeuralNetwork(mods, torch.nn.Identity(), True, True).cuda().eval()
            out = _run_container(net, x, mb)
    finally:
        ops.set_gemm_mode(prev)
    assert net.report_metrics()["num_edges"] == n_edges
    got = out.output_node_representations.cpu()
    err_ref = float((want.double() - exact).abs().max())
    err_ours = float((got.double() - exact).abs().max())
    err = float((got - want).abs().max())
    print(f"cfg4 N={N} [{gemm_mode}]: per-layer worst {worst:.2e}; end-to-end ours-vs-oracle {err:.2e}, "
          f"ours-vs-fp64 {err_ours:.2e}, oracle-vs-fp64 {err_ref:.2e}")
    assert err_ours <= max(TOL, 2.0 * err_ref), f"vs fp64: ours {err_ours:.3e}, fp32 oracle {err_ref:.3e}"
    assert err <= err_ref + err_ours + 1e-7


def test_container_edge_dropout_branch_matches_oracle_on_the_kept_edges():
    """graphneuralnetwork.py:105-119: in training mode the container drops edges (and their features) with a
    Bernoulli mask per edge type before the layer loop.  The mask is torch's CUDA generator; replaying it with
    the same seed gives the kept edge lists, on which the oracle must agree."""
    from oracle import mp_oracle as O
    from ptgnn_amd import layers as L, workloads
    from ptgnn_amd.gnn import GraphNeuralNetwork
    H, rate = 64, 0.3
    mb = workloads.batched_graphs(5, 400, 3, 2.5, seed=11)
    N = mb["num_nodes"]
    torch.manual_seed(9)
    l1 = L.GatedMessagePassingLayer(H, H, 7, "sum")
    l2 = L.MlpMessagePassingLayer(H, H, H, 7, "max")
    x = workloads.node_states(N, H, seed=3)
    net = GraphNeuralNetwork([l1, l2], torch.nn.Identity(), True, True, edge_dropout_rate=rate).cuda().train()
    torch.manual_seed(4321)
    with torch.no_grad():
        out = _run_container(net, x, mb)
    # replay: same generator state, same call order (one rand_like per augmented edge type)
    torch.manual_seed(4321)
    adj = to_cuda_adj(O.augment_adjacency(mb["adjacency_lists"], N, True, True))
    kept = []
    for s, d in adj:
        mask = torch.rand_like(s, dtype=torch.float32) > rate
        kept.append((s.masked_select(mask).cpu(), d.masked_select(mask).cpu()))
    assert sum(int(k[0].shape[0]) for k in kept) < sum(int(a[0].shape[0]) for a in adj)
    want = O.run_layer_stack(x, kept, [l1.export_weights(), l2.export_weights()])
    err = float((out.output_node_representations.cpu() - want).abs().max())
    assert err <= TOL, f"edge-dropout branch: max |delta| = {err:.3e}"
    net.eval()
    with torch.no_grad():
        full = _run_container(net, x, mb)
    want_full, _ = O.gnn_forward(x, mb["adjacency_lists"], [l1.export_weights(), l2.export_weights()], True, True)
    assert float((full.output_node_representations.cpu() - want_full).abs().max()) <= TOL   # eval: no dropout


def test_container_return_all_states_and_linear_residual_match_oracle():
    """graphneuralnetwork.py:132-133 (`return_all_states`: concatenation of the input and every module's
    output) and residuallayers.py:99-142 (LinearResidualLayer) through the container."""
    from oracle import mp_oracle as O
    from ptgnn_amd import layers as L, workloads
    from ptgnn_amd.gnn import GraphNeuralNetwork
    H = 64
    mb = workloads.batched_graphs(4, 500, 2, 2.5, seed=12)
    N = mb["num_nodes"]
    torch.manual_seed(10)
    g1 = L.GatedMessagePassingLayer(H, H, 5, "mean")
    m1 = L.MlpMessagePassingLayer(H, 2 * H, H, 5, "sum")
    lin = L.LinearResidualLayer(H, 2 * H, H)
    mods = [lin.pass_through_dummy_layer(), g1, m1, lin]
    w = lin.state_dict()["_LinearResidualLayer__linear_combination.weight"].detach().cpu()
    specs = [{"kind": "residual_origin", "name": "r"}, g1.export_weights(), m1.export_weights(),
             {"kind": "residual_linear", "name": "r", "w": w}]
    x = workloads.node_states(N, H, seed=4)
    trace = []
    want, _ = O.gnn_forward(x, mb["adjacency_lists"], specs, True, True, trace=trace)
    net = GraphNeuralNetwork(mods, torch.nn.Identity(), True, True).cuda().eval()
    assert net.output_node_state_dim == H
    with torch.no_grad():
        out = _run_container(net, x, mb)
        all_states = _run_container(net, x, mb, return_all_states=True)
    assert float((out.output_node_representations.cpu() - want).abs().max()) <= TOL
    want_all = torch.cat([x] + [o for _, o in trace], dim=-1)
    got_all = all_states.output_node_representations.cpu()
    assert got_all.shape == want_all.shape == (N, H + H + H + 2 * H + H)
    assert float((got_all - want_all).abs().max()) <= TOL


def test_container_edge_feature_embedder_branch_matches_oracle():
    """graphneuralnetwork.py:162-186: embedded edge features ride along (reverse edges reuse the forward
    features, self edges get zeros) into GGNN layers built with `edge_feature_dimension` (K = H + F,
    gatedmessagepassing.py:57-61) -- the general per-edge path."""
    from oracle import mp_oracle as O
    from ptgnn_amd import layers as L, workloads
    from ptgnn_amd.gnn import GraphNeuralNetwork
    H, F = 32, 8
    mb = workloads.batched_graphs(3, 300, 2, 2.0, seed=13)
    N = mb["num_nodes"]

    class Embed(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = torch.nn.Linear(5, F, bias=False)

        def forward(self, features):
            return self.lin(features)

    torch.manual_seed(14)
    emb = Embed()
    layer = L.GatedMessagePassingLayer(H, H, 5, "sum", edge_feature_dimension=F)
    gen = torch.Generator().manual_seed(15)
    raw = [torch.randn(int(s.shape[0]), 5, generator=gen) for s, _ in mb["adjacency_lists"]]
    x = workloads.node_states(N, H, seed=5)
    feats = [emb.lin(r).detach() for r in raw]
    want, _ = O.gnn_forward(x, mb["adjacency_lists"], [layer.export_weights()], True, True, edge_features=feats)
    net = GraphNeuralNetwork([layer], torch.nn.Identity(), True, True, edge_feature_embedder=emb).cuda().eval()
    with torch.no_grad():
        out = _run_container(net, x, mb, edge_feature_data=[{"features": r.cuda()} for r in raw])
    err = float((out.output_node_representations.cpu() - want).abs().max())
    assert err <= TOL, f"edge-feature branch: max |delta| = {err:.3e}"


def test_plan_build_flags_out_of_range_ids_without_touching_memory():
    """The reference device-asserts on a bad node id (F.embedding).  Here the plan build clamps it (no
    out-of-bounds access anywhere downstream) and the count surfaces as a PtgnnAmdError."""
    from ptgnn_amd import _lib, ops
    N = 1000
    g = torch.Generator().manual_seed(3)
    src = torch.randint(0, N, (5000,), generator=g)
    dst = torch.randint(0, N, (5000,), generator=g)
    dst[17] = N + 5
    src[99] = -3
    ops.clear_plan_cache()
    ops.check_indices(sync=True)            # clean slate
    plan = ops.build_plan([(src.cuda(), dst.cuda())], N)
    y = torch.randn(N, 64, device="cuda")
    out = ops.gather_reduce(y, plan, 64, "sum")            # runs on the clamped plan: no fault
    msgs = ops.edge_linear(y, [(src.cuda(), dst.cuda())], [torch.randn(64, 64, device="cuda")], False)
    torch.cuda.synchronize()
    assert torch.isfinite(out).all() and torch.isfinite(msgs).all()
    with pytest.raises(_lib.PtgnnAmdError, match="outside"):
        ops.check_indices(sync=True)
    ops.check_indices(sync=True)            # the counter was reset by the raise
    good = ops.build_plan([(src.clamp(0, N - 1).cuda(), dst.clamp(0, N - 1).cuda())], N)
    ops.check_indices(sync=True)
    assert int(good.rowptr[-1]) == 5000


def test_pooling_accepts_an_unsorted_element_to_sample_map():
    """varsizedsummary.py:28-41 is a plain scatter: any element -> sample map must work, sorted or not."""
    from oracle.scatter_ref import scatter
    from ptgnn_amd import reduceops as R
    g = torch.Generator().manual_seed(8)
    idx = torch.randint(0, 7, (500,), generator=g)
    assert bool((idx[1:] < idx[:-1]).any())
    x = torch.randn(500, 32, generator=g)
    for kind in ("sum", "max", "mean"):
        red = R.SimpleVarSizedElementReduce(kind)
        got = red(R.ElementsToSummaryRepresentationInput(x.cuda(), idx.cuda(), 7)).cpu()
        want = scatter(x, idx, dim=0, dim_size=7, reduce=kind)
        assert float((got - want).abs().max()) <= 1e-5, kind
    sidx, _ = torch.sort(idx)
    got = R.SimpleVarSizedElementReduce("sum")(R.ElementsToSummaryRepresentationInput(x.cuda(), sidx.cuda(), 7)).cpu()
    assert float((got - scatter(x, sidx, dim=0, dim_size=7, reduce="sum")).abs().max()) <= 1e-5


# ------------------------------------------------------------------------------------------------
# round 2: sharded layer forms (edge form, global exchange)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind,agg", [("ggnn", "max"), ("ggnn", "sum"), ("mlp", "max"), ("mlp_notarget", "mean")])
def test_sharded_edge_form_equals_unsharded(kind, agg):
    """Many sparse edge types (program-graph shape): the sharded layers take the grouped per-edge GEMM over the
    local table [own | halo] -- same kernels, same per-row order => bit-identical to the unsharded layer."""
    from oracle import mp_oracle as O
    from ptgnn_amd import layers as L, ops, workloads
    H, world = 64, 3
    mb = workloads.batched_graphs(6, 700, 8, 2.2, seed=31)
    n = mb["num_nodes"]
    adj = O.augment_adjacency(mb["adjacency_lists"], n, True, True)      # T = 17
    T = len(adj)
    x = workloads.node_states(n, H, seed=32).cuda()
    torch.manual_seed(33)
    if kind == "ggnn":
        layer = L.GatedMessagePassingLayer(H, H, T, agg)
    else:
        layer = L.MlpMessagePassingLayer(H, H, H, T, agg, use_target_state_as_message_input=kind == "mlp")
    layer = layer.cuda().eval()
    assert L._prefer_edge_path(sum(int(a[0].shape[0]) for a in adj), n, T, H, H)
    ops.clear_plan_cache()
    cadj = to_cuda_adj(adj)
    with torch.no_grad():
        want = layer(x, cadj, None, {}, {}, empty_feats(cadj, "cuda"))
    ranges, shards = _virtual_shards(adj, n, world)          # edge-mass balanced: graphs straddle the cuts
    assert all(sh.n_halo > 0 for sh in shards)
    outs = []
    with torch.no_grad():
        for sh in shards:
            _emulate_exchange(sh, lambda: x)                 # the edge form ships node states
            assert L._prefer_edge_path(sh.plan.num_edges, sh.n_local + sh.n_halo, T, H, H)
            outs.append(layer.forward_sharded(x[sh.lo:sh.hi].contiguous(), sh))
    np.testing.assert_array_equal(torch.cat(outs).cpu().numpy(), want.cpu().numpy())


def test_sharded_varmisuse_ggnn_stack_with_global_exchange_equals_unsharded():
    """varmisuse/train.py:76-107 through sharded.run_stack: GGNN layers (edge form, T = 9), two global-exchange
    layers (weighted-sum and max pooling) and mean residuals, on a partition that follows graph boundaries
    (each graph's pool is then complete on its rank: the cross-rank combine is the world-1 identity here; the
    straddling case is covered by the gloo test).  Bit-identical to the unsharded container."""
    import socket
    import torch.distributed as dist
    from oracle import mp_oracle as O
    from ptgnn_amd import layers as L, ops, reduceops as R, sharded, workloads
    from ptgnn_amd.gnn import GraphNeuralNetwork
    H = 64
    mb = workloads.batched_graphs(9, 500, 4, 2.2, seed=41)
    n, n2g = mb["num_nodes"], mb["node_to_graph_idx"]
    adj = O.augment_adjacency(mb["adjacency_lists"], n, True, True)
    T = len(adj)
    torch.manual_seed(42)
    ggnn = L.GatedMessagePassingLayer(H, H, T, "sum")
    r1, r2 = L.MeanResidualLayer(H), L.MeanResidualLayer(H)
    g1 = R.GruGlobalStateUpdate(R.WeightedSumVarSizedElementReduce(H), H, H)
    g2 = R.GruGlobalStateUpdate(R.SimpleVarSizedElementReduce("max"), H, H)
    mods = [r1.pass_through_dummy_layer(), r2.pass_through_dummy_layer(), ggnn, ggnn, ggnn, g1, ggnn, r1,
            ggnn, ggnn, ggnn, g2, ggnn, r2]
    net = GraphNeuralNetwork(mods, torch.nn.Identity(), True, True).cuda().eval()
    x = workloads.node_states(n, H, seed=43)
    with torch.no_grad():
        want = _run_container(net, x, mb).output_node_representations
    # cuts on graph boundaries: graphs 0-2 | 3-5 | 6-8
    first = torch.searchsorted(n2g, torch.tensor([0, 3, 6, 9])).tolist()
    ranges = [(first[i], first[i + 1]) for i in range(3)]
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        outs = []
        xg = x.cuda()
        with torch.no_grad():
            for sh in _shards_for(adj, ranges):
                assert sh.n_halo == 0                         # whole graphs per rank: nothing to exchange
                _emulate_exchange(sh, lambda: xg)
                sh.attach_graph_index(n2g[sh.lo:sh.hi].cuda(), 9)
                outs.append(sharded.run_stack(mods, xg[sh.lo:sh.hi].contiguous(), sh))
    finally:
        dist.destroy_process_group()
    np.testing.assert_array_equal(torch.cat(outs).cpu().numpy(), want.cpu().numpy())


@pytest.mark.parametrize("kind", ["ggnn", "mlp"])
def test_sharded_edge_form_training_and_dropout_world1(kind):
    """Training over a shard in the edge form (world = 1 RCCL group, so `forward_sharded` == the whole graph):
    with dropout 0 the gradients equal the unsharded layer's; with the shipped per-edge dropout the step runs
    (the reference's dropout_rate configs -- typilus/train.py:44, varmisuse/train.py:80 -- can train sharded)."""
    import socket
    import torch.distributed as dist
    from oracle import mp_oracle as O
    from ptgnn_amd import layers as L, ops, sharded, workloads
    H = 64
    mb = workloads.batched_graphs(4, 500, 8, 2.2, seed=51)
    n = mb["num_nodes"]
    adj = to_cuda_adj(O.augment_adjacency(mb["adjacency_lists"], n, True, True))
    T = len(adj)
    x = workloads.node_states(n, H, seed=52).cuda()
    gout = workloads.node_states(n, H, seed=53).cuda()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        torch.manual_seed(54)
        layer = (L.GatedMessagePassingLayer(H, H, T, "max") if kind == "ggnn"
                 else L.MlpMessagePassingLayer(H, H, H, T, "max")).cuda().train()
        grads = []
        for mode in ("plain", "sharded"):
            layer.zero_grad()
            xi = x.clone().requires_grad_(True)
            ops.clear_plan_cache()
            if mode == "plain":
                y = layer(xi, adj, None, {}, {}, empty_feats(adj, "cuda"))
            else:
                y = layer.forward_sharded(xi, sharded.ShardedGraph.build(adj, (0, n)))
            y.backward(gout)
            grads.append([y.detach(), xi.grad] + [p.grad.clone() for p in layer.parameters()])
        for a, b in zip(*grads):
            assert float((a - b).abs().max()) <= 1e-5 * max(1.0, float(a.abs().max()))
        if kind == "ggnn":
            torch.manual_seed(55)
            drop = L.GatedMessagePassingLayer(H, H, T, "max", dropout_rate=0.1).cuda().train()
            xi = x.clone().requires_grad_(True)
            y = drop.forward_sharded(xi, sharded.ShardedGraph.build(adj, (0, n)))
            y.backward(gout)
            assert torch.isfinite(y).all() and torch.isfinite(xi.grad).all() and float(xi.grad.abs().sum()) > 0
            assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in drop.parameters())
    finally:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------
# round 2: AMP dtypes, odd widths and the general per-edge path all stay on the HIP kernels
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("kind", ["ggnn", "mlp"])
def test_amp_dtypes_are_upcast_like_the_reference_aggregation(kind, dtype):
    """trainer.py:205,221 runs the model under torch.cuda.amp; abstractmessagepassing.py:43-50 up-casts fp16
    messages to fp32 at the scatter.  fp16 / bf16 node states take the fused path in fp32 and come back in the
    caller's dtype: equal to the fp32 oracle on the same (rounded) inputs up to the output rounding."""
    from oracle import mp_oracle as O
    from ptgnn_amd import layers as L, ops, workloads
    H, T = 64, 5
    mb = workloads.batched_graphs(3, 400, 2, 2.5, seed=61)
    n = mb["num_nodes"]
    adj = O.augment_adjacency(mb["adjacency_lists"], n, True, True)
    torch.manual_seed(62)
    layer = (L.GatedMessagePassingLayer(H, H, T, "max") if kind == "ggnn" else L.MlpMessagePassingLayer(H, H, H, T, "sum"))
    x_lo = workloads.node_states(n, H, seed=63).to(dtype)
    feats = [torch.empty(a[0].shape[0], 0) for a in adj]
    fn = O.ggnn_layer if kind == "ggnn" else O.mlp_mp_layer
    want = fn(x_lo.float(), adj, feats, layer.export_weights())
    cadj = to_cuda_adj(adj)
    ops.clear_plan_cache()
    with torch.no_grad():
        got = layer.cuda().eval()(x_lo.cuda(), cadj, None, {}, {}, empty_feats(cadj, "cuda"))
    assert got.dtype == dtype
    eps = 1e-3 if dtype == torch.float16 else 8e-3
    err = float((got.float().cpu() - want).abs().max())
    assert err <= eps * max(1.0, float(want.abs().max())), f"{dtype}: {err:.3e}"
    with torch.autocast("cuda", dtype=dtype), torch.no_grad():      # inside an autocast region too
        got2 = layer(x_lo.cuda(), cadj, None, {}, {}, empty_feats(cadj, "cuda"))
    assert torch.equal(got2, got)


def test_odd_widths_and_general_path_run_on_the_hip_kernels(monkeypatch):
    """Widths that are not multiples of 4, edge features (K = H + F) and hidden edge MLPs: the dense blocks are the
    HIP GEMM (no torch.nn.functional.linear / nn.GRUCell call anywhere), forward and backward, and match the oracle."""
    from oracle import mp_oracle as O
    from ptgnn_amd import dense, layers as L, ops
    calls = []
    real_linear, real_gru = torch.nn.functional.linear, torch.nn.GRUCell.forward
    monkeypatch.setattr(torch.nn.functional, "linear", lambda *a, **k: (calls.append("F.linear"), real_linear(*a, **k))[1])
    monkeypatch.setattr(torch.nn.GRUCell, "forward", lambda self, *a, **k: (calls.append("GRUCell"), real_gru(self, *a, **k))[1])
    g = torch.Generator().manual_seed(71)
    n, H, M, F_, T = 500, 30, 18, 5, 3                   # nothing is a multiple of 4
    adj = [(torch.randint(0, n, (c,), generator=g), torch.randint(0, n, (c,), generator=g)) for c in (900, 0, 400)]
    x = torch.randn(n, H, generator=g)
    feats = [torch.randn(int(a[0].shape[0]), F_, generator=g) for a in adj]
    torch.manual_seed(72)
    ggnn = L.GatedMessagePassingLayer(H, M, T, "sum", edge_feature_dimension=F_)
    mlp = L.MlpMessagePassingLayer(H, 22, M, T, "max", mlp_hidden_layers=1, features_dimension=F_)
    cadj = to_cuda_adj(adj)
    cfeats = [f.cuda() for f in feats]
    for layer, fn in ((ggnn, O.ggnn_layer), (mlp, O.mlp_mp_layer)):
        want = fn(x, adj, feats, layer.export_weights())
        ops.clear_plan_cache()
        with torch.no_grad():
            got = layer.cuda().eval()(x.cuda(), cadj, None, {}, {}, cfeats).cpu()
        assert float((got - want).abs().max()) <= TOL
    # training through the differentiable HIP nodes with odd widths: gradients vs oracle autograd (inference-mode
    # GRU here: the training cell needs widths % 4 == 0 and says so)
    xi = x.cuda().requires_grad_(True)
    w = torch.randn(M, H + F_, generator=g).cuda().requires_grad_(True)
    b = torch.randn(M, generator=g).cuda().requires_grad_(True)
    inp = torch.cat([xi, torch.randn(n, F_, generator=g).cuda()], dim=1)
    y = dense.linear(inp, w, b)
    gy = torch.randn(n, M, generator=g).cuda()
    y.backward(gy)
    inp_c = inp.detach().cpu().double()
    assert float((w.grad.cpu().double() - gy.cpu().double().t() @ inp_c).abs().max()) <= 1e-4
    assert float((b.grad.cpu().double() - gy.cpu().double().sum(0)).abs().max()) <= 1e-4
    assert float((xi.grad.cpu().double() - (gy.cpu().double() @ w.detach().cpu().double())[:, :H]).abs().max()) <= 1e-4
    # the GRU cell in TRAINING with odd widths, and nn.GRUCell(bias=False): the reference accepts both
    # (gatedmessagepassing.py:25); here the two gate GEMMs run on the differentiable HIP Linear + torch's elementwise
    # gate math -- outputs and gradients vs torch's own GRUCell in float64 on the CPU
    for bias in (True, False):
        torch.manual_seed(17)
        cell = torch.nn.GRUCell(M, H, bias=bias)
        ref = torch.nn.GRUCell(M, H, bias=bias).double()
        ref.load_state_dict({k: v.double() for k, v in cell.state_dict().items()})
        a0, h0 = torch.randn(n, M, generator=g), x.clone()
        ar, hr = a0.double().requires_grad_(True), h0.double().requires_grad_(True)
        gout = torch.randn(n, H, generator=g)
        real_gru(ref, ar, hr).backward(gout.double())
        cell = cell.cuda()
        ac, hc = a0.cuda().requires_grad_(True), h0.cuda().requires_grad_(True)
        out = dense.gru_cell(cell, ac, hc)
        out.backward(gout.cuda())
        with torch.no_grad():
            assert float((out.cpu().double() - real_gru(ref, ar, hr)).abs().max()) <= TOL
        assert float((ac.grad.cpu().double() - ar.grad).abs().max()) <= 1e-4
        assert float((hc.grad.cpu().double() - hr.grad).abs().max()) <= 1e-4
        assert float((cell.weight_ih.grad.cpu().double() - ref.weight_ih.grad).abs().max()) <= 1e-3
    assert calls == [], calls


class _DoneWork:
    def wait(self):
        return True


def _emulate_begin_exchange(shard, global_rows):
    def begin_exchange(table):
        table[shard.n_local:] = global_rows().index_select(0, shard.need_ids)
        return _DoneWork()
    shard.begin_exchange = begin_exchange


@pytest.mark.parametrize("case", ["ggnn_edge_max", "ggnn_edge_sum", "ggnn_table_states_max", "mlp_table_rows_sum",
                                  "mlp_table_rows_max", "mlp_edge_max", "mlp_table_states_min"])
def test_sharded_two_block_overlap_mode_equals_unsharded(case, monkeypatch):
    """`ShardedGraph(overlap=True)`: edges split into an own-source and a halo-source block, two partial
    aggregations (the first one would run under the halo all-to-all) and a combine pass with the row epilogue.
    max / min: bit-identical to the unsharded layer; sum: (own partial) + (halo partial) instead of the CSR fold
    order => within 1e-6.  Every way the rows can travel is covered (node states for the edge form and for wide
    tables, message-table rows for narrow ones)."""
    from oracle import mp_oracle as O
    from ptgnn_amd import layers as L, ops, sharded, workloads
    H, world = 64, 3
    kind, form, agg = case.split("_")[0], "_".join(case.split("_")[1:-1]), case.split("_")[-1]
    # the layer form is picked per (sub)graph from its edge / row counts; pin it so that the shard and the whole
    # graph take the SAME form (for the MLP layer with target state the two forms differ by fp32 re-association:
    # one K = 2H chain vs two K = H chains)
    monkeypatch.setattr(L, "EDGE_PATH_BIAS", 0.0 if form == "edge" else 1e9)
    if form == "edge":
        mb = workloads.batched_graphs(6, 700, 8, 2.2, seed=31)
        n = mb["num_nodes"]
        adj = O.augment_adjacency(mb["adjacency_lists"], n, True, True)      # T = 17 -> edge form
    else:
        n = 3000
        g = torch.Generator().manual_seed(17)
        counts = [16000] if form == "table_rows" else [9000, 0, 4000]       # T*M <= H ships message rows
        adj = [(torch.randint(0, n, (c,), generator=g), torch.randint(0, n, (c,), generator=g)) for c in counts]
    T = len(adj)
    x = workloads.node_states(n, H, seed=32).cuda()
    torch.manual_seed(33)
    layer = (L.GatedMessagePassingLayer(H, H, T, agg) if kind == "ggnn" else L.MlpMessagePassingLayer(H, H, H, T, agg))
    layer = layer.cuda().eval()
    ops.clear_plan_cache()
    cadj = to_cuda_adj(adj)
    monkeypatch.setattr(L, "UNIQUE_MESSAGES", False)      # the unsharded layer: one message row per edge ...
    with torch.no_grad():
        want = layer(x, cadj, None, {}, {}, empty_feats(cadj, "cuda"))
    monkeypatch.setattr(L, "UNIQUE_MESSAGES", True)       # ... the shards (GGNN edge form): one per (type, source) pair
    monkeypatch.setattr(ops, "UNIQUE_MIN_EDGES", 0)
    monkeypatch.setattr(ops, "_UNIQ_SKIP", [0])
    indeg = torch.zeros(n, dtype=torch.int64)
    for _, d in adj:
        indeg += torch.bincount(d, minlength=n)
    ranges = sharded.balanced_node_ranges(indeg, world)
    outs = []
    with torch.no_grad():
        for p, (lo, hi) in enumerate(ranges):
            mine = [(s[(d >= lo) & (d < hi)].cuda(), d[(d >= lo) & (d < hi)].cuda()) for s, d in adj]
            sh = sharded.ShardedGraph.build_local(mine, ranges, p, overlap=True)
            assert sh.overlap and sh.n_halo > 0
            assert sh.plan_own.num_edges + sh.plan_halo.num_edges == sh.num_edges and sh.plan_halo.num_edges > 0
            if kind == "mlp" and form == "table_rows":      # message-table rows travel
                w = layer._stacked_edge_weights()[: T * H]
                rows = ops.linear(x, w)
                _emulate_begin_exchange(sh, lambda rows=rows: rows)
            else:
                _emulate_begin_exchange(sh, lambda: x)
            outs.append(layer.forward_sharded(x[lo:hi].contiguous(), sh))
    got = torch.cat(outs)
    if agg in ("max", "min"):
        np.testing.assert_array_equal(got.cpu().numpy(), want.cpu().numpy())
    else:
        assert float((got - want).abs().max()) <= 1e-6 * max(1.0, float(want.abs().max()))


# ------------------------------------------------------------------------------------------------
# the ctypes stub printed there is the documented way into the C ABI -- run it verbatim
# (also the build path WITHOUT a caller-owned control block: the library zeroes one inside the workspace)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("reduce_id,reduce", [(0, "sum"), (2, "max")])
def test_integration_md_ctypes_stub_runs_as_printed(reduce_id, reduce):
    import re
    from oracle import scatter_ref
    from ptgnn_amd import _lib
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "INTEGRATION.md")).read()
    blocks = [b for b in re.findall(r"```python\n(.*?)```", text, flags=re.S) if "def aggregate(" in b]
    assert len(blocks) == 1
    ns = {}
    exec(blocks[0].replace('ctypes.CDLL("libptgnn_amd.so")', f'ctypes.CDLL({_lib.LIB_PATH!r})'), ns)
    g = torch.Generator().manual_seed(77)
    n, counts = 5000, [9000, 0, 4000, 1]
    adj = [(torch.randint(0, n, (c,), generator=g), torch.randint(0, n, (c,), generator=g)) for c in counts]
    msgs = torch.randn(sum(counts), 48, generator=g)
    for _ in range(2):     # twice: the second build must find the in-workspace control block usable again
        got = ns["aggregate"](msgs.cuda(), to_cuda_adj(adj), n, reduce_id)
    want = scatter_ref.scatter(msgs, torch.cat([d for _, d in adj]), dim=0, dim_size=n, reduce=reduce)
    if reduce == "max":
        np.testing.assert_array_equal(got.cpu().numpy(), want.numpy())
    else:
        assert float((got.cpu() - want).abs().max()) <= TOL


# ------------------------------------------------------------------------------------------------
# row epilogue (GELU -> LayerNorm) as a differentiable HIP pair: the training-time twin of the fused
# aggregation epilogue (mlpmessagepassing.py:114-116)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dim", [64, 128, 48, 50, 256, 512, 300, 4])
@pytest.mark.parametrize("flags", ["gelu", "ln", "gelu_ln"])
def test_row_epilogue_forward_and_backward_match_torch_cpu(dim, flags):
    from ptgnn_amd import dense
    g = torch.Generator().manual_seed(dim * 7 + len(flags))
    n = 3001
    x = torch.randn(n, dim, generator=g) * 1.5
    gy = torch.randn(n, dim, generator=g)
    ln_ref = torch.nn.LayerNorm(dim) if "ln" in flags else None
    if ln_ref is not None:
        with torch.no_grad():
            ln_ref.weight.copy_(torch.randn(dim, generator=g))
            ln_ref.bias.copy_(torch.randn(dim, generator=g))
    xr = x.clone().requires_grad_(True)
    yr = torch.nn.functional.gelu(xr) if "gelu" in flags else xr
    if ln_ref is not None:
        yr = ln_ref(yr)
    yr.backward(gy)
    ln = None
    if ln_ref is not None:
        ln = torch.nn.LayerNorm(dim).cuda()
        ln.load_state_dict(ln_ref.state_dict())
    xg = x.cuda().requires_grad_(True)
    y = dense.row_epilogue(xg, "gelu" in flags, ln)
    y.backward(gy.cuda())
    assert float((y.detach().cpu() - yr.detach()).abs().max()) <= TOL
    pairs = [(xg.grad.cpu(), xr.grad)]
    if ln is not None:
        pairs += [(ln.weight.grad.cpu(), ln_ref.weight.grad), (ln.bias.grad.cpu(), ln_ref.bias.grad)]
    for a, b in pairs:       # gradients: the tolerance of the other backward tests (2e-5 x scale)
        assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max()))


def test_row_epilogue_equals_the_fused_inference_epilogue_bitwise_and_handles_edge_shapes():
    from ptgnn_amd import dense, ops
    g = torch.Generator().manual_seed(3)
    n, dim = 2000, 128
    x = torch.randn(n, dim, generator=g).cuda()
    ln = torch.nn.LayerNorm(dim).cuda()
    with torch.no_grad():
        ln.weight.copy_(torch.randn(dim, generator=g))
        ln.bias.copy_(torch.randn(dim, generator=g))
        plan = ops.plan_from_sorted_index(torch.arange(n, device="cuda"), n)      # one slot per row: sum == x
        fused = ops.gather_reduce(x, plan, dim, "sum", epilogue=ops.EPI_GELU | ops.EPI_LAYERNORM,
                                  ln_weight=ln.weight, ln_bias=ln.bias, ln_eps=ln.eps, type_bits=0)
        node = dense.row_epilogue(x, True, ln)
    assert torch.equal(fused, node)
    # no rows: empty output, zero parameter gradients
    x0 = torch.zeros(0, dim, device="cuda", requires_grad=True)
    y0 = dense.row_epilogue(x0, True, ln)
    y0.sum().backward()
    assert y0.shape == (0, dim) and float(ln.weight.grad.abs().max()) == 0.0
    with pytest.raises(Exception):
        dense.row_epilogue(torch.zeros(4, 600, device="cuda"), True, None)        # wider than the kernels tile


# ------------------------------------------------------------------------------------------------
# the round-1 tile kernels (GEMM mode 0: the fallback of every shape the streaming core does not take) on the same bars
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", ["cfg1_ppi", "cfg2_sum", "cfg2_max", "golden_ggnn_max_edge", "golden_ggnn_sum_table",
                                  "golden_mlp_sum_edge", "golden_mlp_max_table", "autograd_ggnn_sum_edge",
                                  "autograd_mlp_mean_table"])
def test_tile_gemm_mode_parity_matrix(case, monkeypatch):
    """ptgnn_amd_set_gemm_mode(0) against the SAME bars as the streaming kernels: BASELINE configs 1 and 2 at full size vs
    the oracle (1e-5), and the training step -- outputs, d x and every parameter gradient -- vs the reference's own
    gradients (tests/golden/train_*.npz) and vs oracle autograd.  (Configs 3, 4 and 5 carry a `tile` parameter in their
    own tests.)  Rounds 2-4 ran this matrix on the opt-in 3 x bf16 split arithmetic, removed in round 5: mode 2 now
    answers EINVAL (asserted below)."""
    from ptgnn_amd import PtgnnAmdError, ops
    with pytest.raises(PtgnnAmdError):
        ops.set_gemm_mode(2)
    with pytest.raises(PtgnnAmdError):
        ops.set_gemm_mode("split")
    prev = ops.set_gemm_mode("tile")
    try:
        if case == "cfg1_ppi":
            test_config1_ppi_ggnn_full_size_vs_oracle()
        elif case.startswith("cfg2_"):
            test_config2_full_size_vs_oracle(case.split("_")[1])
        elif case.startswith("golden_"):
            _, kind, agg, path = case.split("_")
            name = {"ggnn": f"train_ggnn_{agg}", "mlp": "train_mlp_sum_target" if agg == "sum" else "train_mlp_max_notarget"}[kind]
            test_training_gradients_match_reference_golden(name, path, monkeypatch)
        else:
            _, kind, agg, path = case.split("_")
            test_training_gradients_match_oracle_autograd(kind, agg, path, monkeypatch)
    finally:
        ops.set_gemm_mode(prev)


def test_shard_index_counts_global_source_ids_outside_the_id_space():
    """a corrupt GLOBAL source id is clamped by the shard index pass into a valid own / halo row, which the
    plan build's range guard can no longer see; the index pass itself counts it and the count surfaces as a
    PtgnnAmdError (the reference device-asserts in F.embedding, gatedmessagepassing.py:54-56)."""
    from ptgnn_amd import _lib, ops, sharded
    ops.check_indices(sync=True)
    g = torch.Generator().manual_seed(4)
    n = 3000
    ranges = [(0, 1000), (1000, 2000), (2000, 3000)]
    s = torch.randint(0, n, (5000,), generator=g)
    d = torch.randint(1000, 2000, (5000,), generator=g)
    good = sharded.ShardedGraph.build_local([(s.cuda(), d.cuda())], ranges, 1, use_hip_index=True)
    ops.check_indices(sync=True)                                   # a clean shard raises nothing
    s_bad = s.clone()
    s_bad[[3, 77, 4001]] = torch.tensor([n + 5, -2, 1 << 40])
    ls, ld, _, _, stats = ops.shard_index([(s_bad.cuda(), d.cuda())], 1000, 2000, good.bounds, n)
    assert int(ls.max()) < 1000 + int(stats[3 + 1])                # nothing points outside the local [own | halo] table
    with pytest.raises(_lib.PtgnnAmdError, match="3 node id"):
        ops.check_indices(sync=True)
    with pytest.raises(_lib.PtgnnAmdError, match="3 node id"):     # ... and through the sharded build: raised by the plan
        sharded.ShardedGraph.build_local([(s_bad.cuda(), d.cuda())], ranges, 1, use_hip_index=True)   # build's poll or
        ops.check_indices(sync=True)                                                                   # at the latest here
    ops.check_indices(sync=True)                                   # the counter was cleared by the raise
    assert good.n_halo > 0


@pytest.mark.parametrize("case", ["cfg4_like_T21", "one_type_random", "no_remote", "types_over_table"])
def test_shard_index_kernels_equal_the_torch_bookkeeping(case):
    """csrc/shard_index.hip (bitmap mark / compact / remap) against the torch-op chain it replaces (and which the
    gloo CPU tests still run): identical halo id lists, per-owner counts, remapped adjacency and own-source counts,
    for every rank of a 3-way partition."""
    from oracle import mp_oracle as O
    from ptgnn_amd import sharded, workloads
    g = torch.Generator().manual_seed(23)
    if case == "cfg4_like_T21":
        mb = workloads.batched_graphs(6, 900, 10, 2.4, seed=5)
        n = mb["num_nodes"]
        adj = O.augment_adjacency(mb["adjacency_lists"], n, True, True)
    elif case == "one_type_random":
        n = 50_001
        adj = [(torch.randint(0, n, (400_000,), generator=g), torch.randint(0, n, (400_000,), generator=g))]
    elif case == "no_remote":
        n = 3000
        adj = [(torch.randint(0, 1000, (5000,), generator=g) + 1000 * (i % 3), torch.randint(0, 1000, (5000,), generator=g) + 1000 * (i % 3))
               for i in range(3)]
    else:
        n = 2000
        adj = [(torch.randint(0, n, (30 + t,), generator=g), torch.randint(0, n, (30 + t,), generator=g)) for t in range(70)]
    indeg = torch.zeros(n, dtype=torch.int64)
    for _, d in adj:
        indeg += torch.bincount(d, minlength=n)
    ranges = [(0, 1000), (1000, 2000), (2000, 3000)] if case == "no_remote" else sharded.balanced_node_ranges(indeg, 3)
    for rank, (lo, hi) in enumerate(ranges):
        mine = [(s[(d >= lo) & (d < hi)].cuda(), d[(d >= lo) & (d < hi)].cuda()) for s, d in adj]
        a = sharded.ShardedGraph.build_local(mine, ranges, rank, overlap=True, use_hip_index=True)
        b = sharded.ShardedGraph.build_local(mine, ranges, rank, overlap=True, use_hip_index=False)
        assert a.recv_splits == b.recv_splits and a.n_halo == b.n_halo
        assert torch.equal(a.need_ids, b.need_ids)
        for (sa, da), (sb, db) in zip(a.local_adj, b.local_adj):
            assert torch.equal(sa, sb) and torch.equal(da, db)
        for (sa, da), (sb, db) in zip(a.adj_own + a.adj_halo, b.adj_own + b.adj_halo):   # the own-source counts agree
            assert torch.equal(sa, sb) and torch.equal(da, db)
        if case == "no_remote":
            assert a.n_halo == 0
