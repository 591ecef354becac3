"""The host-tensor route (ptgnn_amd/torch_route.py): CPU tensors through the drop-in layers, the container and the
`torch_scatter` facade.  Pinned to the fixtures the REFERENCE's own modules generated (tests/golden/*.npz) -- forward
and gradients -- and, for the facade family, to the oracle restatement of torch_scatter and to known answers.
No GPU, no reference checkout needed (runs on the GPU box's CPU as well)."""
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden
from helpers import empty_feats, layer_from_spec, stack_from_specs

LAYERS = ["ggnn_layer_sum", "ggnn_layer_mean", "ggnn_layer_max", "ggnn_layer_min", "mlp_layer_sum_target",
          "mlp_layer_max_target", "mlp_layer_mean_notarget", "mlp_layer_sum_hidden1", "mlp_layer_max_noln_nodense",
          "ggnn_layer_max_w128", "mlp_layer_sum_target_w128"]


@pytest.mark.parametrize("name", LAYERS)
def test_layer_on_cpu_tensors_matches_reference_golden(name):
    from oracle.fixtures import unpack_adj, unpack_specs
    g = load_golden(name)
    adj, (spec,) = unpack_adj(g), unpack_specs(g)
    layer = layer_from_spec(spec).eval()
    with torch.no_grad():
        y = layer(torch.from_numpy(g["x"]), adj, None, {}, {}, empty_feats(adj, "cpu"))
    np.testing.assert_allclose(y.numpy(), g["y"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("name", ["gnn_stack_ggnn_typilus", "gnn_stack_mlp_varmisuse", "gnn_stack_ggnn_varmisuse_global",
                                  "gnn_stack_ggnn_typilus_w64"])
def test_container_on_cpu_tensors_matches_reference_golden(name):
    from oracle.fixtures import unpack_adj, unpack_specs
    from ptgnn_amd.gnn import GraphNeuralNetwork
    g = load_golden(name)
    adj, specs = unpack_adj(g), unpack_specs(g)
    net = GraphNeuralNetwork(stack_from_specs(specs), torch.nn.Identity(), introduce_backwards_edges=True,
                             add_self_edges=True).eval()
    n2g = torch.from_numpy(g["node_to_graph_idx"])
    with torch.no_grad():
        out = net(node_data={"input": torch.from_numpy(g["x"])}, adjacency_lists=adj, edge_feature_data=[],
                  node_to_graph_idx=n2g, reference_node_ids={}, reference_node_graph_idx={},
                  num_graphs=int(n2g.max()) + 1)
    np.testing.assert_allclose(out.output_node_representations.numpy(), g["y"], rtol=0, atol=2e-6)
    assert out.node_to_graph_idx is n2g
    assert net.report_metrics()["num_nodes"] == g["x"].shape[0]


@pytest.mark.parametrize("name", ["train_ggnn_max", "train_ggnn_sum", "train_mlp_sum_target", "train_mlp_max_notarget",
                                  "train_ggnn_max_w64", "train_mlp_sum_target_w64", "train_ggnn_sum_w128"])
def test_training_on_cpu_tensors_matches_the_reference_gradients(name):
    from oracle.fixtures import unpack_adj, unpack_specs
    g = load_golden(name)
    adj, (spec,) = unpack_adj(g), unpack_specs(g)
    layer = layer_from_spec(spec).train()
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    y = layer(x, adj, None, {}, {}, empty_feats(adj, "cpu"))
    y.backward(torch.from_numpy(g["gout"]))
    np.testing.assert_allclose(y.detach().numpy(), g["y"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(x.grad.numpy(), g["g.x"], rtol=0, atol=2e-6 * max(1.0, np.abs(g["g.x"]).max()))
    grads, checked = dict(layer.named_parameters()), 0
    for key in g.files:
        if key.startswith("g.") and key != "g.x":
            np.testing.assert_allclose(grads[key[2:]].grad.numpy(), g[key], rtol=0,
                                       atol=2e-6 * max(1.0, np.abs(g[key]).max()), err_msg=key)
            checked += 1
    assert checked >= 8


def test_a_gpu_tensor_can_never_enter_the_torch_route():
    """The route is device dispatch, not a fallback: its entry points refuse device tensors (checked with a meta-like
    stand-in: no GPU here), and nothing in ptgnn_amd imports the oracle."""
    from ptgnn_amd import _lib, torch_route

    class FakeCuda:
        is_cuda = True
    with pytest.raises(_lib.PtgnnAmdError):
        torch_route._host_only(FakeCuda())
    src = open(torch_route.__file__).read()
    assert "import oracle" not in src and "from oracle" not in src


@pytest.mark.parametrize("shape", [(40,), (40, 3), (40, 2, 3)])
@pytest.mark.parametrize("dim", [0, -1])
def test_facade_family_on_cpu_tensors_matches_the_oracle_restatement(shape, dim):
    from oracle import scatter_ref as O
    from ptgnn_amd import scatter as S
    g = torch.Generator().manual_seed(3)
    src = torch.randn(*shape, generator=g)
    n = 9
    E = src.shape[dim]
    index = torch.randint(0, n - 2, (E,), generator=g)          # segments n-2, n-1 stay empty
    for reduce in ("sum", "mean", "max", "min", "mul"):
        got = S.scatter(src, index, dim=dim, dim_size=n, reduce=reduce)
        want = O.scatter(src, index, dim=dim, dim_size=n, reduce=reduce)
        np.testing.assert_array_equal(got.numpy(), want.numpy(), err_msg=reduce)
    for ours, theirs in ((S.scatter_max, O.scatter_max), (S.scatter_min, O.scatter_min)):
        (v, a), (wv, wa) = ours(src, index, dim=dim, dim_size=n), theirs(src, index, dim=dim, dim_size=n)
        np.testing.assert_array_equal(v.numpy(), wv.numpy())
        np.testing.assert_array_equal(a.numpy(), wa.numpy())
    for ours, theirs in ((S.scatter_log_softmax, O.scatter_log_softmax), (S.scatter_softmax, O.scatter_softmax)):
        np.testing.assert_allclose(ours(src, index, dim=dim, eps=0.0, dim_size=n).numpy(),
                                   theirs(src, index, dim=dim, eps=0.0, dim_size=n).numpy(), rtol=0, atol=1e-6)


def test_facade_logsumexp_and_std_known_answers_and_gradients():
    from ptgnn_amd import scatter as S
    src = torch.tensor([0.5, -1.0, 2.0, 3.0, 3.0, -4.0], dtype=torch.float64)
    index = torch.tensor([0, 0, 2, 2, 2, 3])
    lse = S.scatter_logsumexp(src, index, dim=0, dim_size=5, eps=0.0)
    want = [torch.logsumexp(src[index == s], 0) if (index == s).any() else torch.tensor(float("-inf")) for s in range(5)]
    np.testing.assert_allclose(lse.numpy(), torch.stack(want).numpy(), rtol=1e-12)
    std = S.scatter_std(src, index, dim=0, dim_size=5)
    for s in (0, 2):
        seg = src[index == s]
        np.testing.assert_allclose(float(std[s]), float((seg.var(unbiased=True) * (len(seg) - 1) / (len(seg) - 1 + 1e-6)).sqrt()),
                                   rtol=1e-12)
    assert float(std[1]) == 0.0 and float(std[3]) == 0.0 and float(std[4]) == 0.0
    # max routes its gradient to the FIRST winner only (torch_scatter's arg), not spread over the tie at 3.0
    x = src.clone().requires_grad_(True)
    S.scatter(x, index, dim=0, dim_size=5, reduce="max").sum().backward()
    assert x.grad.tolist() == [1.0, 0.0, 0.0, 1.0, 0.0, 1.0]
    x = src.clone().requires_grad_(True)
    torch.autograd.gradcheck(lambda t: S.scatter_logsumexp(t, index, dim=0, dim_size=4, eps=0.0)[[0, 2, 3]], (x,))
    torch.autograd.gradcheck(lambda t: S.scatter(t, index, dim=0, dim_size=5, reduce="mul"), (x,))


def test_install_registers_the_facade_as_torch_scatter_in_a_fresh_interpreter():
    """`ptgnn_amd.scatter.install()`: the import forms the reference's sources use (abstractmessagepassing.py:4,
    varmisuse.py:8, varsizedsummary.py:7, grucopydecoder.py:9-10, graphnorm.py:3) resolve to the facade."""
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import ptgnn_amd.scatter as S\n"
        "m = S.install()\n"
        "from torch_scatter import scatter, scatter_add, scatter_sum, scatter_mean, scatter_max, scatter_min, scatter_mul\n"
        "from torch_scatter import scatter_log_softmax, scatter_softmax, scatter_logsumexp, scatter_std\n"
        "from torch_scatter.composite import scatter_log_softmax as a, scatter_logsumexp as b, scatter_softmax as c\n"
        "import torch_scatter.composite\n"
        "import torch, torch_scatter\n"
        "assert torch_scatter is m and scatter is S.scatter and a is S.scatter_log_softmax and b is S.scatter_logsumexp\n"
        "assert S.install() is m\n"
        "v, arg = scatter_max(torch.tensor([1., 5., 2.]), torch.tensor([1, 1, 0]))\n"
        "assert v.tolist() == [2., 5.] and arg.tolist() == [2, 1]\n"
        "print('INSTALL_OK', torch_scatter.__version__)\n" % ROOT)
    proc = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert proc.returncode == 0 and "INSTALL_OK" in proc.stdout, proc.stdout + proc.stderr


def test_host_route_edge_cases_empty_inputs_features_and_empty_edge_types():
    from ptgnn_amd import layers as L, scatter as S
    none = torch.zeros(0, dtype=torch.int64)
    for red, fill in (("sum", 0.0), ("mean", 0.0), ("max", 0.0), ("min", 0.0), ("mul", 1.0)):
        out = S.scatter(torch.zeros(0, 3), none, dim=0, dim_size=4, reduce=red)
        assert tuple(out.shape) == (4, 3) and float(out.min()) == fill == float(out.max())
    v, a = S.scatter_max(torch.zeros(0, 3), none, dim=0, dim_size=4)
    assert tuple(v.shape) == (4, 3) and int(a.max()) == 0          # arg of an empty segment = src.size(dim) = 0
    assert tuple(S.scatter(torch.zeros(0), none, dim=0).shape) == (0,)
    assert bool(torch.isneginf(S.scatter_logsumexp(torch.zeros(0, 2), none, dim=0, dim_size=3)).all())
    assert float(S.scatter_std(torch.zeros(0, 2), none, dim=0, dim_size=3).abs().max()) == 0.0
    x = torch.randn(5, 3, requires_grad=True)
    S.scatter(x[:0], none, dim=0, dim_size=2, reduce="max").sum().backward()
    assert float(x.grad.abs().max()) == 0.0
    # layers built with edge features, one edge type empty (graphneuralnetwork.py:162-186 hands features per type)
    adj = [(torch.tensor([0, 1]), torch.tensor([1, 2])), (none, none), (torch.tensor([3]), torch.tensor([0]))]
    feats = [torch.randn(2, 2), torch.zeros(0, 2), torch.randn(1, 2)]
    g = L.GatedMessagePassingLayer(8, 12, 3, "max", edge_feature_dimension=2).eval()
    m = L.MlpMessagePassingLayer(8, 6, 10, 3, "mean", features_dimension=2).eval()
    with torch.no_grad():
        assert tuple(g(torch.randn(4, 8), adj, None, {}, {}, feats).shape) == (4, 8)
        assert tuple(m(torch.randn(4, 8), adj, None, {}, {}, feats).shape) == (4, 6)


def test_nan_is_the_extremum_of_its_segment_and_the_installed_module_has_a_spec():
    """a NaN in a segment propagates through max / min like torch_scatter's comparison loop (it used to match
    nothing and silently pool to 0), and `importlib.util.find_spec("torch_scatter")` works after `install()`."""
    from ptgnn_amd import scatter as S
    src = torch.tensor([[1.0, 2.0], [float("nan"), 0.5], [3.0, -1.0], [4.0, 4.0]])
    idx = torch.tensor([0, 0, 0, 2])
    v, a = S.scatter_max(src, idx, dim=0, dim_size=3)
    assert bool(torch.isnan(v[0, 0])) and int(a[0, 0]) == 1              # the NaN element is the recorded winner
    assert v[0, 1].item() == 2.0 and int(a[0, 1]) == 0
    assert v[1].tolist() == [0.0, 0.0] and a[1].tolist() == [4, 4]        # empty segment: 0, arg = src.size(dim)
    assert v[2].tolist() == [4.0, 4.0]
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import importlib.util, ptgnn_amd.scatter as S\n"
            "S.install(force=True)\n"
            "spec = importlib.util.find_spec('torch_scatter')\n"
            "assert spec is not None and spec.name == 'torch_scatter'\n"
            "assert importlib.util.find_spec('torch_scatter.composite').name == 'torch_scatter.composite'\n"
            "print('SPEC_OK')\n" % ROOT)
    proc = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert proc.returncode == 0 and "SPEC_OK" in proc.stdout, proc.stdout + proc.stderr
