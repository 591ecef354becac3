"""Executed in a FRESH interpreter by tests/test_dropin_reference_cpu.py (the reference must be importable
before ptgnn_amd.layers is first imported, so that the layers subclass the reference's ABC).

Checks the drop-in claim against the reference's OWN classes, numerically, on the CPU:
  * GraphNeuralNetworkModel(message_passing_layer_creator=<ptgnn_amd layers>) builds the reference's own
    GraphNeuralNetwork container around them (graphneuralnetwork.py:231,249,298-299);
  * every ptgnn_amd layer loads the state_dict of the reference layer of the same constructor arguments
    (strict: identical name-mangled keys and shapes), including mlp_hidden_layers > 0 and the
    global-exchange / residual layers;
  * ptgnn_amd's GnnOutput IS the reference's;
  * the reference's own container (keyword call, graphneuralnetwork.py:122-131), its own batcher and its own
    `WeightedSumVarSizedElementReduce` run AROUND our layers on CPU tensors (ptgnn_amd/torch_route.py: the route
    `typilus/predict.py:25-27` and a CPU trainer take) and reproduce the reference layers' outputs to 1e-6 -- in eval
    mode, and in training mode with dropout (same seed => same masks) including every parameter gradient;
  * a model saved the reference's way (gzip + torch.save of (model, module), abstractneuralmodel.py:155-163) restores
    on "cpu" and predicts.

    --scatter oracle : the reference imports the TEST-SIDE restatement of torch_scatter (oracle/shims.py); the outputs
                       of the REFERENCE layers are written to --save.
    --scatter facade : `ptgnn_amd.scatter.install()` is `torch_scatter` (no oracle scatter in the process); the outputs
                       must equal the ones under --expect (so the facade, through the reference's own call sites,
                       is checked against an independent implementation).
"""
import argparse
import gzip
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--scatter", choices=("oracle", "facade"), default="oracle")
ap.add_argument("--save")
ap.add_argument("--expect")
args = ap.parse_args()

if args.scatter == "facade":
    import ptgnn_amd.scatter
    mod = ptgnn_amd.scatter.install(force=True)          # BEFORE the first `import ptgnn`
    assert sys.modules["torch_scatter"] is mod and sys.modules["torch_scatter.composite"] is mod.composite
from oracle import shims  # noqa: E402

shims.install()                                           # dpu_utils stubs (+ torch_scatter only if not registered yet)
import torch_scatter  # noqa: E402

assert ("ptgnn_amd" in torch_scatter.__version__) == (args.scatter == "facade"), torch_scatter.__version__
from torch_scatter.composite import scatter_log_softmax as _sls  # noqa: E402,F401  (grucopydecoder.py:10's form)
from ptgnn.baseneuralmodel import AbstractNeuralModel  # noqa: E402
from ptgnn.neuralmodels.gnn import GraphData, GraphNeuralNetwork, GraphNeuralNetworkModel  # noqa: E402
from ptgnn.neuralmodels.gnn.messagepassing import (  # noqa: E402
    GatedMessagePassingLayer, GruGlobalStateUpdate, MeanResidualLayer, MlpMessagePassingLayer)
from ptgnn.neuralmodels.gnn.messagepassing.abstractmessagepassing import AbstractMessagePassingLayer  # noqa: E402
from ptgnn.neuralmodels.gnn.messagepassing.residuallayers import ConcatResidualLayer, LinearResidualLayer  # noqa: E402
from ptgnn.neuralmodels.gnn.structs import GnnOutput  # noqa: E402
from ptgnn.neuralmodels.reduceops.varsizedsummary import (  # noqa: E402
    SimpleVarSizedElementReduce, WeightedSumVarSizedElementReduce)

from ptgnn_amd import gnn as G, layers as L, reduceops as R  # noqa: E402

assert issubclass(L.GatedMessagePassingLayer, AbstractMessagePassingLayer)
assert issubclass(L.MlpMessagePassingLayer, AbstractMessagePassingLayer)
assert G.GnnOutput is GnnOutput
H = 16


class _Identity(torch.nn.Module):
    def forward(self, x):
        return x


class _NodeModel(AbstractNeuralModel):
    def initialize_metadata(self): pass
    def update_metadata_from(self, datapoint): pass
    def finalize_metadata(self): pass
    def build_neural_module(self): return _Identity()
    def tensorize(self, datapoint): return int(datapoint)
    def initialize_minibatch(self): return {"ids": []}

    def extend_minibatch_with(self, tensorized_datapoint, partial_minibatch):
        partial_minibatch["ids"].append(tensorized_datapoint)
        return True

    def finalize_minibatch(self, accumulated_minibatch_data, device):
        g = torch.Generator().manual_seed(len(accumulated_minibatch_data["ids"]))
        return {"x": torch.randn(len(accumulated_minibatch_data["ids"]), H, generator=g)}


def ref_layers(n):
    r1, r2 = ConcatResidualLayer(H), MeanResidualLayer(H)
    lin = LinearResidualLayer(H, H, H)
    return [r1.pass_through_dummy_layer(), GatedMessagePassingLayer(H, 24, n, "max", dropout_rate=0.1), r1,
            MlpMessagePassingLayer(2 * H, H, 32, n, "max", mlp_hidden_layers=2, dropout_rate=0.1),
            r2.pass_through_dummy_layer(),
            MlpMessagePassingLayer(H, H, H, n, "sum", use_target_state_as_message_input=False),
            GruGlobalStateUpdate(WeightedSumVarSizedElementReduce(H), H, H), r2,
            lin.pass_through_dummy_layer(),
            GatedMessagePassingLayer(H, H, n, "mean"),
            GruGlobalStateUpdate(SimpleVarSizedElementReduce("max"), H, H), lin]


def our_layers(n):
    r1, r2 = L.ConcatResidualLayer(H), L.MeanResidualLayer(H)
    lin = L.LinearResidualLayer(H, H, H)
    return [r1.pass_through_dummy_layer(), L.GatedMessagePassingLayer(H, 24, n, "max", dropout_rate=0.1), r1,
            L.MlpMessagePassingLayer(2 * H, H, 32, n, "max", mlp_hidden_layers=2, dropout_rate=0.1),
            r2.pass_through_dummy_layer(),
            L.MlpMessagePassingLayer(H, H, H, n, "sum", use_target_state_as_message_input=False),
            # the REFERENCE's pooling module (its scatter_sum call site, varsizedsummary.py:76-81) inside our layer
            R.GruGlobalStateUpdate(WeightedSumVarSizedElementReduce(H), H, H), r2,
            lin.pass_through_dummy_layer(),
            L.GatedMessagePassingLayer(H, H, n, "mean"),
            R.GruGlobalStateUpdate(R.SimpleVarSizedElementReduce("max"), H, H), lin]


def make_model(creator):
    m = GraphNeuralNetworkModel(node_representation_model=_NodeModel(), message_passing_layer_creator=creator,
                                stop_extending_minibatch_after_num_nodes=500, add_self_edges=True)
    rng = np.random.RandomState(3)
    graphs = []
    for g in range(4):
        n = 9 + 3 * g
        edges = {"a": [(int(a), int(b)) for a, b in rng.randint(0, n, (2 * n, 2))],
                 "b": [(int(a), int(b)) for a, b in rng.randint(0, n, (n // 2, 2))],
                 "c": [(0, 1), (0, 1), (2, 2)]}                       # duplicate edge + self loop
        graphs.append(GraphData(node_information=list(range(n)), edges=edges, reference_nodes={"r": [0, n - 1]}))
    m.compute_metadata(iter(graphs), parallelize=False)
    return m, graphs


ref_model, graphs = make_model(ref_layers)
our_model, _ = make_model(our_layers)
torch.manual_seed(11)
ref_net = ref_model.build_neural_module()
our_net = our_model.build_neural_module()
assert type(our_net) is GraphNeuralNetwork                      # the REFERENCE's container, our layers inside
assert our_net.input_node_state_dim == ref_net.input_node_state_dim
assert our_net.output_node_state_dim == ref_net.output_node_state_dim
ref_sd = ref_net.state_dict()
res = our_net.load_state_dict(ref_sd, strict=True)
assert not res.missing_keys and not res.unexpected_keys
our_sd = our_net.state_dict()
assert list(our_sd.keys()) == list(ref_sd.keys())
for k in ref_sd:
    assert our_sd[k].shape == ref_sd[k].shape and torch.equal(our_sd[k], ref_sd[k]), k
n_params = sum(v.numel() for v in ref_sd.values())


def minibatch(model):   # a fresh one per call: the reference's forward appends to `adjacency_lists` in place
    mb = model.initialize_minibatch()
    for g in graphs:
        model.extend_minibatch_with(model.tensorize(g), mb)
    return model.finalize_minibatch(mb, "cpu")


# ---- eval: the reference's container + batcher around our layers == around its own layers
with torch.no_grad():
    want = ref_net.eval()(**minibatch(ref_model))
    got = our_net.eval()(**minibatch(our_model))
N = sum(len(g.node_information) for g in graphs)
assert want.output_node_representations.shape == (N, H) == got.output_node_representations.shape
err_eval = float((got.output_node_representations - want.output_node_representations).abs().max())
assert err_eval <= 1e-6, f"eval: our layers vs the reference's on CPU: {err_eval:.3e}"
for k in want.node_idx_references:
    assert torch.equal(got.node_idx_references[k], want.node_idx_references[k])
assert torch.equal(got.node_to_graph_idx, want.node_to_graph_idx) and got.num_graphs == want.num_graphs

# ---- train: same seed => same dropout masks (the route calls the same modules in the same order); output + gradients
outs = []
for net, model in ((ref_net, ref_model), (our_net, our_model)):
    net.train()
    net.zero_grad()
    torch.manual_seed(5)
    o = net(**minibatch(model)).output_node_representations
    (o * torch.linspace(-1, 1, o.numel()).view_as(o)).sum().backward()
    outs.append((o.detach(), {k: p.grad.clone() for k, p in net.named_parameters()}))
err_train = float((outs[0][0] - outs[1][0]).abs().max())
assert err_train <= 1e-6, f"train forward: {err_train:.3e}"
err_grad = 0.0
for k, g in outs[0][1].items():
    scale = max(1.0, float(g.abs().max()))
    err_grad = max(err_grad, float((g - outs[1][1][k]).abs().max()) / scale)
assert err_grad <= 2e-6, f"parameter gradients: {err_grad:.3e}"
assert set(outs[0][1]) == set(outs[1][1])

# ---- the reference's save / restore-on-cpu / run cycle (abstractneuralmodel.py:155-163; typilus/predict.py:25-27)
with tempfile.TemporaryDirectory() as tmp:
    path = os.path.join(tmp, "model.pkl.gz")
    with gzip.open(path, "wb") as f:
        torch.save((our_model, our_net), f)
    with gzip.open(path, "rb") as f:
        model2, net2 = torch.load(f, map_location="cpu", weights_only=False)
with torch.no_grad():
    again = net2.eval()(**minibatch(model2)).output_node_representations
assert torch.equal(again, got.output_node_representations)

if args.save:
    np.savez(args.save, eval=want.output_node_representations.numpy(), train=outs[0][0].numpy(),
             **{"g_" + k: v.numpy() for k, v in outs[0][1].items()})
if args.expect:
    exp = np.load(args.expect)
    d = float(np.abs(exp["eval"] - got.output_node_representations.numpy()).max())
    assert d <= 1e-6, f"facade as torch_scatter vs the oracle-scatter run: eval {d:.3e}"
    d = float(np.abs(exp["train"] - outs[1][0].numpy()).max())
    assert d <= 1e-6, f"facade as torch_scatter vs the oracle-scatter run: train {d:.3e}"
    for k, v in outs[1][1].items():
        e = exp["g_" + k]
        d = float(np.abs(e - v.numpy()).max()) / max(1.0, float(np.abs(e).max()))
        assert d <= 2e-6, f"facade vs oracle-scatter run: grad {k}: {d:.3e}"
print(f"DROPIN_OK scatter={args.scatter} keys={len(ref_sd)} params={n_params} eval={err_eval:.1e} "
      f"train={err_train:.1e} grad={err_grad:.1e}")
