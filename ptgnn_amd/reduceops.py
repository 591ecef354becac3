"""Node -> graph pooling ("var-sized element reduce") and the global-exchange layer that uses it:
mirrors of ptgnn/neuralmodels/reduceops/varsizedsummary.py:11-81 and
ptgnn/neuralmodels/gnn/messagepassing/globalgraphexchange.py (same class names, constructor keywords
and name-mangled parameter names), used by the VarMisuse GGNN stack (varmisuse/train.py:76-107).

The pooling is the same segment reduce as message aggregation with `node_to_graph_idx` as the index.
A disjoint-union batch lists the nodes of graph 0, then graph 1, ... (graphneuralnetwork.py:418-423),
so the index is SORTED and the plan needs no sort at all: rowptr = searchsorted, col = identity.
"""
import weakref
from typing import NamedTuple, Union

import torch
from torch import nn

from ptgnn_amd import _lib, dense, ops, torch_route
from ptgnn_amd.layers import AbstractMessagePassingLayer, _check_device, _no_grad_needed
from ptgnn_amd.scatter import gather_rows as gather_rows_autograd, segment_reduce


class ElementsToSummaryRepresentationInput(NamedTuple):
    """varsizedsummary.py:11-18."""
    element_embeddings: torch.Tensor
    element_to_sample_map: torch.Tensor
    num_samples: Union[torch.Tensor, int]


class AbstractVarSizedElementReduce(nn.Module):
    def forward(self, inputs: ElementsToSummaryRepresentationInput) -> torch.Tensor:
        raise NotImplementedError


_NUM_SAMPLES = []   # (weakref(index), version, max + 1) of the most recent index tensors


def _num_samples(index: torch.Tensor) -> int:
    """`index.max() + 1` -- the reference's own host read-back (globalgraphexchange.py:40, once per LAYER there);
    made once per index tensor, i.e. once per minibatch, here."""
    for ref, ver, upper in _NUM_SAMPLES:
        if ref() is index and ver == index._version:
            return upper
    upper = int(index.max()) + 1 if index.numel() else 0
    _NUM_SAMPLES.insert(0, (weakref.ref(index), index._version, upper))
    del _NUM_SAMPLES[4:]
    return upper


def _index_plan(index: torch.Tensor, num_samples: int) -> "ops.GraphPlan":
    """Plan of an element -> sample map.  The reference's reducers are plain torch_scatter calls
    (varsizedsummary.py:35-41,76-81) and accept ANY map, so every map takes the stable plan build -- cached per index
    tensor (`ops.plan_for`), i.e. once per minibatch for all global-exchange layers.  Round 2 tested the map for
    sortedness first to skip the sort for `node_to_graph_idx` (graphneuralnetwork.py:418-423,440-443); that test was
    a host read-back per minibatch, which now costs more than the ~40 us of device time the sort takes."""
    return ops.plan_for([(index, index)], int(num_samples))


def _pool(values: torch.Tensor, index: torch.Tensor, num_samples, reduce: str) -> torch.Tensor:
    if not values.is_cuda:    # device dispatch: CPU tensors take the plain-torch route (ptgnn_amd/torch_route.py)
        return torch_route.segment(values, index, int(num_samples), reduce)    # varsizedsummary.py:35-41: no dtype cast
    plan = _index_plan(index, int(num_samples))
    dt = values.dtype
    return segment_reduce(values.to(torch.float32).contiguous(), plan, reduce).to(dt)


class SimpleVarSizedElementReduce(AbstractVarSizedElementReduce):
    def __init__(self, summarization_type: str):
        super().__init__()
        assert summarization_type in {"sum", "mean", "max", "min"}
        self.__summarization_type = summarization_type

    @property
    def summarization_type(self) -> str:
        return self.__summarization_type

    def forward(self, inputs: ElementsToSummaryRepresentationInput) -> torch.Tensor:
        return _pool(inputs.element_embeddings, inputs.element_to_sample_map, inputs.num_samples,
                     self.__summarization_type)


class _WeightedPool(torch.autograd.Function):
    """sum_{i in sample} sigmoid(x_i . w) x_i as ONE node: forward and backward are the two HIP entry points of
    csrc/weighted_pool.hip (no gemv, no [N, D] product in memory, deterministic weight gradient)."""

    @staticmethod
    def forward(ctx, x, w, index, plan):
        ctx.save_for_backward(x, w, index)
        return ops.weighted_pool(x, w, plan)

    @staticmethod
    def backward(ctx, grad_out):
        x, w, index = ctx.saved_tensors
        gx, gw = ops.weighted_pool_backward(x, w, index, grad_out.contiguous())
        return gx, gw.reshape(w.shape), None, None


class WeightedSumVarSizedElementReduce(AbstractVarSizedElementReduce):
    def __init__(self, representation_size: int):
        super().__init__()
        self.__weights_layer = nn.Linear(representation_size, 1, bias=False)

    @property
    def score_weight(self) -> torch.Tensor:
        """[1, D] weight of the scoring Linear (varsizedsummary.py:71)."""
        return self.__weights_layer.weight

    def forward(self, inputs: ElementsToSummaryRepresentationInput) -> torch.Tensor:
        x, index = inputs.element_embeddings, inputs.element_to_sample_map
        w = self.__weights_layer.weight
        if not x.is_cuda:      # device dispatch: the reference's own operator sequence (varsizedsummary.py:73-81)
            weights = torch.sigmoid(self.__weights_layer(x).squeeze(-1))          # [num_elements]
            return _pool(x * weights.unsqueeze(-1), index, inputs.num_samples, "sum")
        # GPU: score, scaling and segment sum in one HIP pass (csrc/weighted_pool.hip); fp16 / bf16 states (AMP) are
        # pooled in fp32 and cast back, like every aggregation of the package
        plan = _index_plan(index, int(inputs.num_samples))
        dt = x.dtype
        xf, wf = x.to(torch.float32), w.to(torch.float32)
        if _no_grad_needed(xf, wf):
            return ops.weighted_pool(xf, wf, plan).to(dt)
        return _WeightedPool.apply(xf, wf, index, plan).to(dt)


class AbstractGlobalGraphExchange(AbstractMessagePassingLayer):
    """globalgraphexchange.py:13-45: pool node states per graph, broadcast back, update the nodes."""

    def __init__(self, global_graph_representation_module: AbstractVarSizedElementReduce,
                 dropout_rate: float = 0.0):
        super().__init__()
        self.__global_graph_representation_module = global_graph_representation_module
        self.__dropout = nn.Dropout(p=dropout_rate)

    def _update_node_states(self, node_states, global_info_per_node):
        raise NotImplementedError

    @property
    def pooling_module(self) -> AbstractVarSizedElementReduce:
        return self.__global_graph_representation_module

    def forward(self, node_states, adjacency_lists, node_to_graph_idx, reference_node_ids,
                reference_node_graph_idx, edge_features) -> torch.Tensor:
        if not node_states.is_cuda:   # globalgraphexchange.py:37-45 on host tensors
            num_graphs = _num_samples(node_to_graph_idx)
            e = ElementsToSummaryRepresentationInput(node_states, node_to_graph_idx, num_graphs)
            graph_reps = self.__dropout(self.__global_graph_representation_module(e))
            return self._update_node_states(node_states, graph_reps[node_to_graph_idx])
        if node_states.dtype in (torch.float16, torch.bfloat16):   # AMP: fp32 inside, caller's dtype outside
            return self.forward(node_states.float(), adjacency_lists, node_to_graph_idx, reference_node_ids,
                                reference_node_graph_idx, edge_features).to(node_states.dtype)
        num_graphs = _num_samples(node_to_graph_idx)
        e = ElementsToSummaryRepresentationInput(node_states, node_to_graph_idx, num_graphs)
        graph_reps = self.__dropout(self.__global_graph_representation_module(e))
        if graph_reps.dtype != torch.float32:
            per_node = graph_reps[node_to_graph_idx]
        elif _no_grad_needed(graph_reps):
            per_node = ops.gather_rows(graph_reps.contiguous(), node_to_graph_idx)
        else:   # training: HIP row gather whose backward is the HIP segment-sum over the graphs (deterministic)
            plan = _index_plan(node_to_graph_idx, num_graphs)
            per_node = gather_rows_autograd(graph_reps.contiguous(), node_to_graph_idx, plan)
        return self._update_node_states(node_states, per_node)

    def forward_sharded(self, node_states, shard) -> torch.Tensor:
        """The same exchange over a dst-range shard (ptgnn_amd/sharded.py): every rank pools ITS nodes per graph,
        the [num_graphs, D] partial pools are combined with one small all-reduce (a graph may straddle a rank
        boundary), and the broadcast back + node update are local."""
        from ptgnn_amd import sharded
        _check_device(node_states)
        idx, G = shard.node_to_graph_idx, shard.num_graphs
        if idx is None:
            raise _lib.PtgnnAmdError("forward_sharded of a global-exchange layer needs "
                                     "ShardedGraph.attach_graph_index(node_to_graph_idx_local, num_graphs)")
        pool = self.__global_graph_representation_module
        if isinstance(pool, WeightedSumVarSizedElementReduce):
            kind, local = "sum", pool(ElementsToSummaryRepresentationInput(node_states, idx, G))
        elif isinstance(pool, SimpleVarSizedElementReduce):
            kind = pool.summarization_type
            local = _pool(node_states, idx, G, "sum" if kind == "mean" else kind)
        else:
            raise _lib.PtgnnAmdError(f"forward_sharded: cannot combine partial pools of {type(pool).__name__}")
        counts = torch.bincount(idx, minlength=G)[:G]
        graph_reps = sharded.combine_graph_pools(local, counts, kind, shard.group)
        p = self.__dropout.p if self.training else 0.0
        if p > 0 and shard.world > 1:
            # the unsharded layer draws ONE dropout mask per graph representation (globalgraphexchange.py:44-46); every
            # rank holds the same combined representations, so the mask must be the same on every rank too: rank 0 of
            # the group draws it, one small broadcast ([num_graphs, D]) carries it
            import torch.distributed as dist
            keep = (torch.rand_like(graph_reps) >= p).to(graph_reps.dtype) / (1.0 - p)
            dist.broadcast(keep, src=dist.get_global_rank(shard.group, 0) if shard.group is not None else 0,
                           group=shard.group)
            graph_reps = graph_reps * keep
        else:
            graph_reps = self.__dropout(graph_reps)
        if _no_grad_needed(graph_reps):
            per_node = ops.gather_rows(graph_reps.contiguous(), idx)
        else:
            per_node = gather_rows_autograd(graph_reps.contiguous(), idx, _index_plan(idx, G))
        return self._update_node_states(node_states, per_node)


class GruGlobalStateUpdate(AbstractGlobalGraphExchange):
    def __init__(self, global_graph_representation_module: AbstractVarSizedElementReduce,
                 input_state_size: int, summarized_state_size: int, dropout_rate: float = 0.0):
        super().__init__(global_graph_representation_module, dropout_rate)
        self.__input_dim = input_state_size
        self.__summarized_state_size = summarized_state_size
        self.__gru_cell = nn.GRUCell(input_size=summarized_state_size, hidden_size=input_state_size)

    def _update_node_states(self, node_states, global_info_per_node):
        gru = self.__gru_cell
        if not node_states.is_cuda:
            return gru(global_info_per_node, node_states)
        if (node_states.dtype == torch.float32
                and _no_grad_needed(node_states, global_info_per_node, *gru.parameters())):
            return ops.gru_cell(global_info_per_node, node_states, gru.weight_ih, gru.weight_hh,
                                gru.bias_ih, gru.bias_hh)
        return dense.gru_cell(gru, global_info_per_node, node_states)

    @property
    def input_state_dimension(self) -> int:
        return self.__input_dim

    @property
    def output_state_dimension(self) -> int:
        return self.__input_dim

    def export_weights(self) -> dict:
        """Weights in the layout the parity oracle consumes (tests only read this)."""
        gru, pool = self.__gru_cell, self.pooling_module
        spec = {"kind": "global_gru", "w_ih": gru.weight_ih.detach().cpu(), "w_hh": gru.weight_hh.detach().cpu(),
                "b_ih": gru.bias_ih.detach().cpu(), "b_hh": gru.bias_hh.detach().cpu()}
        if isinstance(pool, WeightedSumVarSizedElementReduce):
            spec.update(pool="weighted_sum", pool_w=pool.score_weight.detach().cpu())
        else:
            spec["pool"] = pool.summarization_type
        return spec
