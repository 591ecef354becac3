"""`GraphNeuralNetwork` container + `GnnOutput`, mirroring
ptgnn/neuralmodels/gnn/graphneuralnetwork.py:28-209 and structs.py:52-76 (same constructor,
keyword-only forward, metrics and NamedTuple fields) so the reference's task heads
(`output_node_representations[node_idx_references[name]]`, graph2class.py:84-89,
varmisuse.py:63-75) consume it unchanged.

MI355X-first differences (behaviour-preserving):
  * the augmented adjacency (forward types, reversed types, self edges -- in that order,
    graphneuralnetwork.py:172-186) is turned into ONE dst-sorted plan per forward that all layers
    share; reversed lists are tuple swaps (no device work) and the identity list is cached per N;
  * the caller's `adjacency_lists` list is NOT mutated (the reference appends to it in place,
    :173,:179, which breaks calling a module twice on the same minibatch -- SURVEY.md App. B).
"""
from typing import Any, Dict, List, NamedTuple, Optional, Tuple

import torch
from torch import nn

from ptgnn_amd import ops
from ptgnn_amd.layers import AbstractMessagePassingLayer, ConcatResidualLayer, forward_scope, set_output_hint

try:
    from ptgnn.neuralmodels.gnn.structs import GnnOutput  # type: ignore
except Exception:
    class GnnOutput(NamedTuple):
        """structs.py:52-76."""
        input_node_representations: torch.Tensor
        output_node_representations: torch.Tensor
        node_to_graph_idx: torch.Tensor
        node_idx_references: Dict[str, torch.Tensor]
        node_graph_idx_reference: Dict[str, torch.Tensor]
        num_graphs: int

        @property
        def reference_nodes_idx(self) -> Dict[str, torch.Tensor]:
            return self.node_idx_references

        @property
        def reference_nodes_graph_idx(self) -> Dict[str, torch.Tensor]:
            return self.node_graph_idx_reference

try:
    from ptgnn.baseneuralmodel import ModuleWithMetrics  # type: ignore
except Exception:
    class ModuleWithMetrics(nn.Module):
        """Minimal mirror of ptgnn/baseneuralmodel/modulewithmetrics.py:28-64."""

        def __init__(self):
            super().__init__()
            self._reset_module_metrics()

        def _reset_module_metrics(self) -> None:
            pass

        def _module_metrics(self) -> Dict[str, Any]:
            return {}

        def report_metrics(self) -> Dict[str, Any]:
            out: Dict[str, Any] = {}
            for m in self.modules():
                if isinstance(m, ModuleWithMetrics):
                    out.update(m._module_metrics())
            return out

        def reset_metrics(self) -> None:
            for m in self.modules():
                if isinstance(m, ModuleWithMetrics):
                    m._reset_module_metrics()

        def train(self, mode: bool = True):
            self.reset_metrics()
            return super().train(mode)


class GraphNeuralNetwork(ModuleWithMetrics):
    """A generic message-passing GNN with discrete edge types (graphneuralnetwork.py:28-209)."""

    def __init__(self, message_passing_layers: List[AbstractMessagePassingLayer],
                 node_embedder: nn.Module, introduce_backwards_edges: bool, add_self_edges: bool,
                 edge_dropout_rate: float = 0.0, edge_feature_embedder: Optional[nn.Module] = None):
        super().__init__()
        self.__message_passing_layers = nn.ModuleList(message_passing_layers)
        self.__node_embedder = node_embedder
        self.__introduce_backwards_edges = introduce_backwards_edges
        self.__add_self_edges = add_self_edges
        assert 0 <= edge_dropout_rate < 1
        self.__edge_dropout_rate = edge_dropout_rate
        self.__edge_feature_embedder = edge_feature_embedder
        self._idents: Optional[torch.Tensor] = None

    @property
    def input_node_state_dim(self) -> int:
        return self.__message_passing_layers[0].input_state_dimension

    @property
    def output_node_state_dim(self) -> int:
        return self.__message_passing_layers[-1].output_state_dimension

    @property
    def message_passing_layers(self):
        return self.__message_passing_layers

    def _reset_module_metrics(self) -> None:
        self.__num_graphs, self.__num_edges, self.__num_nodes = 0, 0, 0

    def _module_metrics(self) -> Dict[str, Any]:
        # A natural synchronisation point (the trainer reads the metrics at the end of an epoch / evaluation,
        # trainer.py:236): surface node ids that were out of range in ANY minibatch since the last check -- with
        # the default PTGNN_AMD_VALIDATE=async a bad id in the last minibatch, or in an inference-only call, would
        # otherwise never be reported (the reference device-asserts at once; "sync" reproduces that per build)
        if torch.cuda.is_available():
            from ptgnn_amd import ops
            ops.check_indices(sync=True)
        return {"num_graphs": int(self.__num_graphs), "num_nodes": int(self.__num_nodes),
                "num_edges": int(self.__num_edges)}

    def _identity_edges(self, num_nodes: int, device) -> torch.Tensor:
        ids = self._idents
        if ids is None or ids.shape[0] != num_nodes or ids.device != device:
            ids = torch.arange(num_nodes, dtype=torch.int64, device=device)
            self._idents = ids
        return ids

    def gnn(self, node_representations: torch.Tensor,
            adjacency_lists: List[Tuple[torch.Tensor, torch.Tensor]],
            edge_feature_embeddings: List[torch.Tensor], node_to_graph_idx: torch.Tensor,
            reference_node_ids: Dict[str, torch.Tensor],
            reference_node_graph_idx: Dict[str, torch.Tensor],
            return_all_states: bool = False) -> torch.Tensor:
        if self.__edge_dropout_rate > 0 and self.training:           # :105-119
            kept_adj, kept_feats = [], []
            for (src, dst), feats in zip(adjacency_lists, edge_feature_embeddings):
                mask = torch.rand_like(src, dtype=torch.float32) > self.__edge_dropout_rate
                kept_adj.append((src.masked_select(mask), dst.masked_select(mask)))
                kept_feats.append(feats[mask])
            adjacency_lists, edge_feature_embeddings = kept_adj, kept_feats

        if node_representations.is_cuda:
            # one sort for the whole stack; layers find it through the identity-keyed plan cache
            ops.plan_for(adjacency_lists, node_representations.shape[0])
        all_states = [node_representations]
        with forward_scope():   # tied layers share their stacked weights within this forward
            layers = list(self.__message_passing_layers)
            for li, mp_layer in enumerate(layers):                    # :122-131
                buf = None
                nxt = layers[li + 1] if li + 1 < len(layers) else None
                if (isinstance(nxt, ConcatResidualLayer) and node_representations.is_cuda and not return_all_states
                        and node_representations.dtype == torch.float32
                        and not (torch.is_grad_enabled() and (node_representations.requires_grad or self.training))):
                    # the layer in front of a concat residual writes into the right half of the residual's result
                    out_dim = mp_layer.output_state_dimension
                    buf = nxt.make_result_buffer(node_representations.shape[0], out_dim, node_representations)
                    set_output_hint(buf[:, buf.shape[1] - out_dim:])
                try:
                    node_representations = mp_layer(
                        node_states=node_representations, adjacency_lists=adjacency_lists,
                        node_to_graph_idx=node_to_graph_idx, reference_node_ids=reference_node_ids,
                        reference_node_graph_idx=reference_node_graph_idx,
                        edge_features=edge_feature_embeddings)
                finally:   # a layer that raises must not leave its hint to an unrelated forward
                    set_output_hint(None)
                if buf is not None and node_representations.data_ptr() == buf[:, buf.shape[1] - out_dim:].data_ptr():
                    node_representations._ptgnn_amd_concat_buffer = buf
                all_states.append(node_representations)
        if return_all_states:
            node_representations = torch.cat(all_states, dim=-1)
        return node_representations

    def forward(self, *, node_data, adjacency_lists: List[Tuple[torch.Tensor, torch.Tensor]],
                edge_feature_data: List, node_to_graph_idx: torch.Tensor,
                reference_node_ids: Dict[str, torch.Tensor],
                reference_node_graph_idx: Dict[str, torch.Tensor], num_graphs, **kwargs) -> GnnOutput:
        initial = self.__node_embedder(**node_data)                   # [N, D]  :160
        device = node_to_graph_idx.device
        if self.__edge_feature_embedder is None:
            feats = [torch.empty(f.shape[0], 0, device=device) for f, _ in adjacency_lists]
        else:
            feats = [self.__edge_feature_embedder(**e) for e in edge_feature_data]

        adj = list(adjacency_lists)                                   # never mutate the caller's list
        if self.__introduce_backwards_edges:                          # :172-174
            adj += [(t, f) for f, t in adj]
            feats += [e for e in feats]
        if self.__add_self_edges:                                     # :176-186
            num_nodes = node_to_graph_idx.shape[0]
            idents = self._identity_edges(num_nodes, device)
            adj.append((idents, idents))
            feats.append(torch.zeros(num_nodes, feats[-1].shape[-1], device=device))

        output = self.gnn(initial, adj, feats, node_to_graph_idx, reference_node_ids,
                          reference_node_graph_idx, **kwargs)
        with torch.no_grad():                                         # :198-201
            self.__num_edges += sum(a[0].shape[0] for a in adj)
            self.__num_graphs += num_graphs
            self.__num_nodes += node_to_graph_idx.shape[0]
        return GnnOutput(input_node_representations=initial, output_node_representations=output,
                         node_to_graph_idx=node_to_graph_idx, node_idx_references=reference_node_ids,
                         node_graph_idx_reference=reference_node_graph_idx, num_graphs=num_graphs)
