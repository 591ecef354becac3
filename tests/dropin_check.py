"""Executed in a FRESH interpreter by tests/test_dropin_reference_cpu.py (the reference must be importable
before ptgnn_amd.layers is first imported, so that the layers subclass the reference's ABC).

Checks the drop-in claim against the reference's OWN classes:
  * GraphNeuralNetworkModel(message_passing_layer_creator=<ptgnn_amd layers>) builds the reference's own
    GraphNeuralNetwork container around them (graphneuralnetwork.py:231,249,298-299);
  * every ptgnn_amd layer loads the state_dict of the reference layer of the same constructor arguments
    (strict: identical name-mangled keys and shapes), including mlp_hidden_layers > 0 and the
    global-exchange / residual layers;
  * ptgnn_amd's GnnOutput IS the reference's;
  * on CPU tensors the layers fail loudly (no CPU fallback).
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import shims  # noqa: E402

shims.install()
from ptgnn.baseneuralmodel import AbstractNeuralModel  # noqa: E402
from ptgnn.neuralmodels.gnn import GraphData, GraphNeuralNetwork, GraphNeuralNetworkModel  # noqa: E402
from ptgnn.neuralmodels.gnn.messagepassing import (  # noqa: E402
    GatedMessagePassingLayer, GruGlobalStateUpdate, MeanResidualLayer, MlpMessagePassingLayer)
from ptgnn.neuralmodels.gnn.messagepassing.abstractmessagepassing import AbstractMessagePassingLayer  # noqa: E402
from ptgnn.neuralmodels.gnn.messagepassing.residuallayers import ConcatResidualLayer, LinearResidualLayer  # noqa: E402
from ptgnn.neuralmodels.gnn.structs import GnnOutput  # noqa: E402
from ptgnn.neuralmodels.reduceops.varsizedsummary import (  # noqa: E402
    SimpleVarSizedElementReduce, WeightedSumVarSizedElementReduce)

from ptgnn_amd import _lib, gnn as G, layers as L, reduceops as R  # noqa: E402

assert issubclass(L.GatedMessagePassingLayer, AbstractMessagePassingLayer)
assert issubclass(L.MlpMessagePassingLayer, AbstractMessagePassingLayer)
assert G.GnnOutput is GnnOutput


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
        return {"x": torch.randn(len(accumulated_minibatch_data["ids"]), 16)}


H = 16


def ref_layers(n):
    r1, r2 = ConcatResidualLayer(H), MeanResidualLayer(H)
    lin = LinearResidualLayer(H, H, H)
    return [r1.pass_through_dummy_layer(), GatedMessagePassingLayer(H, 24, n, "max", dropout_rate=0.1), r1,
            MlpMessagePassingLayer(2 * H, H, 32, n, "max", mlp_hidden_layers=2, dropout_rate=0.1),
            r2.pass_through_dummy_layer(),
            MlpMessagePassingLayer(H, H, H, n, "sum", use_target_state_as_message_input=False),
            GruGlobalStateUpdate(WeightedSumVarSizedElementReduce(H), H, H), r2,
            lin.pass_through_dummy_layer(),
            GruGlobalStateUpdate(SimpleVarSizedElementReduce("max"), H, H), lin]


def our_layers(n):
    r1, r2 = L.ConcatResidualLayer(H), L.MeanResidualLayer(H)
    lin = L.LinearResidualLayer(H, H, H)
    return [r1.pass_through_dummy_layer(), L.GatedMessagePassingLayer(H, 24, n, "max", dropout_rate=0.1), r1,
            L.MlpMessagePassingLayer(2 * H, H, 32, n, "max", mlp_hidden_layers=2, dropout_rate=0.1),
            r2.pass_through_dummy_layer(),
            L.MlpMessagePassingLayer(H, H, H, n, "sum", use_target_state_as_message_input=False),
            R.GruGlobalStateUpdate(R.WeightedSumVarSizedElementReduce(H), H, H), r2,
            lin.pass_through_dummy_layer(),
            R.GruGlobalStateUpdate(R.SimpleVarSizedElementReduce("max"), H, H), lin]


def make_model(creator):
    m = GraphNeuralNetworkModel(node_representation_model=_NodeModel(), message_passing_layer_creator=creator,
                                stop_extending_minibatch_after_num_nodes=50, add_self_edges=True)
    graphs = [GraphData(node_information=list(range(5)), edges={"a": [(0, 1), (1, 2)], "b": [(3, 4)]},
                        reference_nodes={"r": [0]}) for _ in range(3)]
    m.compute_metadata(iter(graphs), parallelize=False)
    return m, graphs


ref_model, graphs = make_model(ref_layers)
our_model, _ = make_model(our_layers)
ref_net, our_net = ref_model.build_neural_module(), our_model.build_neural_module()
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

# the reference's own minibatch loop feeds the reference container with our layers; on CPU they must refuse
def minibatch(model):   # a fresh one per call: the reference's forward appends to `adjacency_lists` in place
    mb = model.initialize_minibatch()
    for g in graphs:
        model.extend_minibatch_with(model.tensorize(g), mb)
    return model.finalize_minibatch(mb, "cpu")


try:
    our_net.eval()(**minibatch(our_model))
    raise SystemExit("expected the ptgnn_amd layers to refuse CPU tensors")
except _lib.PtgnnAmdError as exc:
    assert "MI355X" in str(exc)
with torch.no_grad():
    out = ref_net.eval()(**minibatch(ref_model))                # sanity: the reference stack itself runs
assert out.output_node_representations.shape == (15, H)
print(f"DROPIN_OK keys={len(ref_sd)} params={n_params}")
