"""Drop-in message-passing layers: same class names, constructor keywords, forward signature,
properties and (name-mangled) parameter names as the reference's
ptgnn/neuralmodels/gnn/messagepassing/{abstractmessagepassing,gatedmessagepassing,
mlpmessagepassing,residuallayers}.py and ptgnn/neuralmodels/mlp.py -- so
``GraphNeuralNetworkModel(message_passing_layer_creator=...)`` (graphneuralnetwork.py:231,298)
accepts them unchanged and ``layer.load_state_dict(reference_layer.state_dict())`` just works.

What is different is HOW a layer runs.  Inference (`torch.no_grad()` or no input requires grad,
dropout inactive, no edge features, single-Linear edge transform) takes the fused MI355X path:

    plan  = dst-sorted CSR of ALL edge types, built once per minibatch          (csr_build.hip)
    Y     = X [W_0; ...; W_{T-1}]^T              one wide fp32-MFMA GEMM        (dense_f32.hip)
    A     = reduce_{in-edges} Y[src, type] (+ Y_dst[v, type]) (+GELU+LayerNorm) (gather_reduce.hip)
    X'    = GRUCell(A, X)  |  tanh(A W^T + b)    fp32-MFMA, fused epilogues     (dense_f32.hip)

using  Linear_t(x_src) == (X W_t^T)[src]  (a bias-free Linear commutes with the row gather) and,
for the MLP layer with target state,  W_t [x_u ; x_v] = W_t^s x_u + W_t^d x_v.

With many sparse edge types the per-node table is replaced by ONE grouped per-edge GEMM over all types
(edge_gemm.hip), picked per minibatch.

Training (grad required) runs on the same kernels: the edge form is one autograd node
(ptgnn_amd/scatter.py `edge_linear`: grouped GEMM forward, split-edge weight-gradient GEMM, input
gradient through the same grouped GEMM + HIP segment-sum) with the GGNN layer's per-edge dropout folded
in as a counter-based hash mask; the table form differentiates through `scatter.gather_reduce` (backward
= the gather-reduce kernel over a backward plan); GRU / Linear blocks are `ptgnn_amd/dense.py` nodes.

Edge features ride the grouped per-edge GEMM as a third K range of its gathered A rows (`ptgnn_amd_edge_linear_feat_f32`;
nothing of the reference's [E, H + F] message input exists in memory) -- inference and, as one autograd node
(`scatter._EdgeLinearFeat`), training; deeper edge MLPs run their first Linear the same way.  Custom aggregation modules,
per-edge dropout together with edge features, biased edge MLPs and widths that are not multiples of 32 take the general
per-edge path: torch only gathers / concatenates rows, every Linear runs on the HIP GEMM (`ptgnn_amd/dense.py`, any width)
and the aggregation on the HIP segment-reduce seam with its autograd rule.  fp16 / bf16 node states (AMP) are up-cast
to fp32 on entry and the result is cast back.  No GPU path runs on a vendor BLAS or falls back to torch.

Device dispatch (round 5): tensors on the CPU take `ptgnn_amd/torch_route.py` (plain torch operators, so that the
reference's `predict.py` -- which restores and runs on "cpu" -- and a CPU `ModelTrainer` work with these layers);
tensors on the GPU always take the HIP library and raise when it is missing.
"""
import contextlib
import os
import threading
from typing import Dict, List, Optional, Tuple, Union

import torch
from torch import nn

from ptgnn_amd import _lib, dense, ops, torch_route
from ptgnn_amd.scatter import (edge_linear as edge_linear_autograd, edge_linear_feat as edge_linear_feat_autograd,
                               gather_reduce as gather_reduce_autograd, segment_reduce)

try:  # inside a ptgnn install the layers ARE ptgnn layers
    from ptgnn.neuralmodels.gnn.messagepassing.abstractmessagepassing import (  # type: ignore
        AbstractMessageAggregation, AbstractMessagePassingLayer)
except Exception:  # standalone (e.g. the GPU box): identical interface
    class AbstractMessagePassingLayer(nn.Module):
        """Interface of abstractmessagepassing.py:8-60."""

        def forward(self, node_states, adjacency_lists, node_to_graph_idx, reference_node_ids,
                    reference_node_graph_idx, edge_features) -> torch.Tensor:
            raise NotImplementedError

        def _aggregate_messages(self, messages, message_targets, num_nodes, aggregation_fn: str):
            from ptgnn_amd.scatter import scatter
            msg_dtype = messages.dtype
            return scatter(messages.to(torch.float32), index=message_targets, dim=0,
                           dim_size=num_nodes, reduce=aggregation_fn).to(msg_dtype)

        @property
        def input_state_dimension(self) -> int:
            raise NotImplementedError

        @property
        def output_state_dimension(self) -> int:
            raise NotImplementedError

    class AbstractMessageAggregation(nn.Module):
        def forward(self, messages, message_targets, num_nodes):
            raise NotImplementedError

        def output_state_size(self, message_input_size: int) -> int:
            raise NotImplementedError

Adj = List[Tuple[torch.Tensor, torch.Tensor]]


# Tensors derived from parameters that may be shared by the layer calls of ONE forward pass (e.g. the
# stacked per-type weights of a tied GGNN layer that the Typilus stack applies seven times): inside a
# `forward_scope()` -- ptgnn_amd.gnn.GraphNeuralNetwork opens one around its layer loop -- they are built
# once, so autograd accumulates one stacked gradient per use instead of T small ones.  Outside a scope
# nothing is cached (the autograd graph of a cached tensor must not outlive its forward).
_SCOPE = threading.local()


@contextlib.contextmanager
def forward_scope():
    outer = getattr(_SCOPE, "cache", None)
    if outer is None:
        _SCOPE.cache = {}
        from ptgnn_amd import dense
        dense.clear_transposed_cache()   # W^T copies of the previous backward: valid for one forward / backward pair
    try:
        yield
    finally:
        _SCOPE.cache = outer


def _scoped(owner, name, make):
    """Per-thread cache keyed on the module OBJECT (kept alive by the key, so an id() can never be reused
    by another module while the entry exists)."""
    cache = getattr(_SCOPE, "cache", None)
    if cache is None or not torch.is_grad_enabled():
        return make()
    key = (owner, name)
    val = cache.get(key)
    if val is None:
        val = cache[key] = make()
    return val


# Output hint: ptgnn_amd.gnn.GraphNeuralNetwork knows which layer feeds a ConcatResidualLayer; it hands that layer the
# right half of the [N, D0 + D1] buffer the residual will return, so the layer's last kernel writes there and the
# residual only has to fill in the left half instead of torch.cat-ing both (inference only; a layer is free to
# ignore the hint -- the residual checks what it actually got).
def set_output_hint(buf: Optional[torch.Tensor]) -> None:
    _SCOPE.out_hint = buf


def _take_output_hint(rows: int, cols: int, like: torch.Tensor, vector_stores: bool = False) -> Optional[torch.Tensor]:
    hint = getattr(_SCOPE, "out_hint", None)
    _SCOPE.out_hint = None
    if hint is None or torch.is_grad_enabled() and like.requires_grad:
        return None
    if tuple(hint.shape) != (rows, cols) or hint.dtype != torch.float32 or hint.device != like.device:
        return None
    if vector_stores and (hint.data_ptr() % 16 != 0 or (rows > 1 and hint.stride(0) % 4 != 0) or hint.stride(1) != 1):
        return None
    return hint


def _no_grad_needed(*tensors) -> bool:
    if not torch.is_grad_enabled():
        return True
    return not any(t is not None and t.requires_grad for t in tensors)


# Per-edge grouped GEMM vs per-node pre-transform: FLOPs are 2*E*K*M vs 2*N*T*K*M, the per-edge rows pay a
# random 512-B gather each; measured crossover on MI355X (profiles/) is near E ~ 0.8 * N * T.
EDGE_PATH_BIAS = 1.25
# GGNN inference, edge form: one message row per distinct (edge type, source) pair instead of one per edge
UNIQUE_MESSAGES = os.environ.get("PTGNN_AMD_UNIQUE_MESSAGES", "1") not in ("", "0")


def _prefer_edge_path(num_edges: int, num_nodes: int, num_types: int, state_dim: int, msg_dim: int) -> bool:
    if state_dim % 32 != 0 or msg_dim % 4 != 0 or num_types < 2:
        return False
    return num_edges * EDGE_PATH_BIAS < num_nodes * num_types


def _edge_messages(table: torch.Tensor, adjacency_lists, plan, weights):
    """GGNN messages of the edge form (inference) and the `col` that maps CSR slots to their rows: one row per distinct
    (edge type, source) pair of the plan where the shared-row launch applies (GraphPlan.unique_messages), else one row
    per edge in reference order (gatedmessagepassing.py:50-61).  The aggregate has the same bits either way."""
    if UNIQUE_MESSAGES and ops.edge_linear_shared_supported(table.shape[1], weights[0].shape[0], len(weights)):
        uniq = plan.unique_messages()
        if uniq is not None:
            return ops.edge_linear_shared(table, uniq, weights), uniq.slot_row
    return ops.edge_linear(table, adjacency_lists, weights, False), plan.perm


def _feat_gemm_ok(node_states, edge_features, state_dim: int, out_dim: int, *params, shapes_only: bool = False) -> bool:
    """Inference with per-edge features (graphneuralnetwork.py:162-186 -> gatedmessagepassing.py:57-61 /
    mlpmessagepassing.py:96-98): the grouped per-edge GEMM reads the feature rows as a third K range of its A operand
    (ptgnn_amd_edge_linear_feat_f32), so the reference's [E, H (+H) + F] input matrix is never built."""
    if node_states.dtype != torch.float32 or state_dim % 32 != 0 or out_dim % 4 != 0:
        return False
    if any(f is None or f.shape[-1] == 0 or f.device != node_states.device or f.dtype != torch.float32
           for f in edge_features):
        return False   # float64 / integer / other-device features: the general path (torch.cat promotes like the reference)
    if len({int(f.shape[-1]) for f in edge_features}) != 1:
        return False
    if shapes_only:     # the caller has a differentiable form (scatter.edge_linear_feat): gradients are no obstacle
        return True
    return (not torch.is_grad_enabled()) or _no_grad_needed(node_states, *edge_features, *params)


def _edge_training_ok(state_dim: int, msg_dim: int) -> bool:
    """The grouped edge GEMM needs its reduction width % 32 == 0: the state width in the forward, the
    message width in the input-gradient GEMM."""
    return state_dim % 32 == 0 and msg_dim % 32 == 0


def _dropout_seed() -> int:
    """A fresh 62-bit seed for the hash dropout mask, drawn from torch's CPU generator (so
    torch.manual_seed makes training runs repeatable) without touching the device."""
    return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())


_AMP_DTYPES = (torch.float16, torch.bfloat16)


def _overlap_reduces():
    from ptgnn_amd import sharded
    return sharded.OVERLAP_REDUCES


def _run_mlp(mlp: "MLP", x: torch.Tensor) -> torch.Tensor:
    """ptgnn/neuralmodels/mlp.py:79-80 with every nn.Linear on the HIP GEMM (general per-edge path)."""
    for m in mlp.modules_in_order:
        x = dense.linear(x, m.weight, m.bias) if isinstance(m, nn.Linear) else m(x)
    return x


def _check_device(node_states: torch.Tensor):
    """The sharded forms (RCCL halo exchange) exist on the GPU only."""
    if not node_states.is_cuda:
        raise _lib.PtgnnAmdError(
            "forward_sharded runs on the MI355X only; node_states is on "
            f"{node_states.device}. (The unsharded `forward` takes CPU tensors through ptgnn_amd.torch_route.)")


def _on_host(node_states: torch.Tensor) -> bool:
    """Device dispatch of the unsharded layers, like a torch operator's: CPU tensors take the plain-torch route
    (ptgnn_amd/torch_route.py: the reference's own `predict.py` restores and runs a model on "cpu",
    typilus/predict.py:25-27; trainer.py:392-395 trains there without a GPU); GPU tensors ALWAYS take
    libptgnn_amd.so and raise when it is missing or fails -- there is no way from a GPU tensor into the torch route."""
    return not node_states.is_cuda


# ------------------------------------------------------------------------------------------------
# MLP (ptgnn/neuralmodels/mlp.py) -- same Sequential index layout => same state_dict keys
# ------------------------------------------------------------------------------------------------
class MLP(nn.Module):
    def __init__(self, input_dimension: int, output_dimension: int,
                 hidden_layers: Union[List[int], int] = 1, use_biases: bool = False,
                 activation: Optional[nn.Module] = nn.ReLU(), dropout_rate: float = 0.0):
        super().__init__()
        if isinstance(hidden_layers, int):
            width = 32 if output_dimension == 1 else output_dimension  # mlp.py:34-43
            sizes = [width] * hidden_layers
        else:
            sizes = list(hidden_layers)
        if len(sizes) > 1:
            assert activation is not None, "Multiple linear layers without an activation"
        mods: List[nn.Module] = []
        d = input_dimension
        for h in sizes:                       # hidden block: Dropout, Linear, (activation)
            lin = nn.Linear(d, h, bias=use_biases)
            nn.init.xavier_uniform_(lin.weight)
            mods += [nn.Dropout(p=dropout_rate), lin] + ([activation] if activation is not None else [])
            d = h
        out = nn.Linear(d, output_dimension, bias=use_biases)   # output block: Dropout, Linear
        nn.init.xavier_uniform_(out.weight)
        mods += [nn.Dropout(p=dropout_rate), out]
        self.__mlp_modules = nn.Sequential(*mods)

    @property
    def linears(self) -> List[nn.Linear]:
        return [m for m in self.__mlp_modules if isinstance(m, nn.Linear)]

    @property
    def modules_in_order(self) -> List[nn.Module]:
        return list(self.__mlp_modules)

    @property
    def is_single_linear(self) -> bool:
        ls = self.linears
        return len(ls) == 1 and ls[0].bias is None

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.__mlp_modules(x)


# ------------------------------------------------------------------------------------------------
# GGNN layer
# ------------------------------------------------------------------------------------------------
class GatedMessagePassingLayer(AbstractMessagePassingLayer):
    """GGNN layer; constructor and semantics of gatedmessagepassing.py:8-77."""

    def __init__(self, state_dimension: int, message_dimension: int, num_edge_types: int,
                 message_aggregation_function: str, dropout_rate: float = 0.0,
                 edge_feature_dimension: int = 0):
        super().__init__()
        self.__edge_message_transformation_layers = nn.ModuleList(
            [nn.Linear(state_dimension + edge_feature_dimension, message_dimension, bias=False)
             for _ in range(num_edge_types)])
        for lin in self.__edge_message_transformation_layers:
            nn.init.xavier_normal_(lin.weight, gain=(1 / num_edge_types) ** 0.5)
        self.__state_update = nn.GRUCell(input_size=message_dimension, hidden_size=state_dimension)
        nn.init.orthogonal_(self.__state_update.weight_hh)
        nn.init.xavier_uniform_(self.__state_update.weight_ih)
        nn.init.normal_(self.__state_update.bias_hh, std=1e-5)
        nn.init.normal_(self.__state_update.bias_ih, std=1e-5)
        self.__state_dimension = state_dimension
        self.__aggregation_fn = message_aggregation_function
        self.__dropout = nn.Dropout(p=dropout_rate)
        self._message_dimension = message_dimension
        self._edge_feature_dimension = edge_feature_dimension

    # -- fused path -------------------------------------------------------------------------
    def _stacked_edge_weights(self) -> torch.Tensor:
        """[T*M, H] = [W_0; ...; W_{T-1}], rebuilt per call: a cache keyed on parameter versions would go
        stale under `p.data` updates (EMA, weight clipping), and the copy is tiny next to the GEMM."""
        ws = [lin.weight.detach() for lin in self.__edge_message_transformation_layers]
        return ws[0] if len(ws) == 1 else torch.cat(ws, dim=0)

    def _table_ok(self, node_states, edge_features) -> bool:
        """Message == row of the per-node table X [W_0; ...]^T: no edge features, no per-edge dropout."""
        if self._edge_feature_dimension != 0 or any(f is not None and f.shape[-1] != 0 for f in edge_features):
            return False
        if self.training and self.__dropout.p > 0:
            return False
        if self.__aggregation_fn not in ops.REDUCE_IDS:   # "mul": the scatter seam's own kernel (general path)
            return False
        return node_states.dtype == torch.float32

    def _fused_ok(self, node_states, edge_features) -> bool:
        if not self._table_ok(node_states, edge_features):
            return False
        return (not torch.is_grad_enabled()) or _no_grad_needed(node_states, *self.parameters())

    def forward(self, node_states: torch.Tensor, adjacency_lists: Adj, node_to_graph_idx,
                reference_node_ids: Dict[str, torch.Tensor],
                reference_node_graph_idx: Dict[str, torch.Tensor],
                edge_features: List[torch.Tensor]) -> torch.Tensor:
        assert len(adjacency_lists) == len(self.__edge_message_transformation_layers)
        if _on_host(node_states):
            return torch_route.ggnn_layer(node_states, adjacency_lists, edge_features,
                                          list(self.__edge_message_transformation_layers), self.__dropout,
                                          self.__state_update, self.__aggregation_fn)
        if node_states.dtype in _AMP_DTYPES:
            # AMP (trainer.py:205,221): the reference computes messages in the autocast dtype and up-casts them
            # to fp32 at the aggregation (abstractmessagepassing.py:43-50).  Here the whole layer runs in fp32 on
            # the HIP kernels -- every intermediate at least as precise as the reference's -- and only the
            # returned states go back to the caller's dtype.
            feats = [f.float() if f is not None and f.dtype in _AMP_DTYPES else f for f in edge_features]
            return self.forward(node_states.float(), adjacency_lists, node_to_graph_idx, reference_node_ids,
                                reference_node_graph_idx, feats).to(node_states.dtype)
        num_nodes = node_states.shape[0]
        plan = ops.plan_for(adjacency_lists, num_nodes)
        gru = self.__state_update
        if self._fused_ok(node_states, edge_features):
            M, T = self._message_dimension, len(adjacency_lists)
            if _prefer_edge_path(plan.num_edges, num_nodes, T, self.__state_dimension, M):
                # many sparse edge types: one grouped per-edge GEMM.  The message W_t x[src] is the same row for
                # every edge of type t that leaves src: the GEMM produces one row per distinct (type, source) pair
                # of the plan (GraphPlan.unique_messages, built once per minibatch on the device) and the
                # aggregation reads it per edge -- same bits, fewer rows.
                weights = [l.weight for l in self.__edge_message_transformation_layers]
                msgs, col = _edge_messages(node_states, adjacency_lists, plan, weights)
                tb = 0
            else:
                msgs, col, tb = ops.linear(node_states, self._stacked_edge_weights()), None, None      # [N, T*M]
            # aggregation -> GRU cell, pipelined over destination-row ranges on large minibatches (ops.aggregate_gru)
            return ops.aggregate_gru(msgs, plan, M, self.__aggregation_fn, node_states, gru.weight_ih, gru.weight_hh,
                                     gru.bias_ih, gru.bias_hh, type_bits=tb, col=col,
                                     out=_take_output_hint(num_nodes, self.__state_dimension, node_states))

        no_feats = self._edge_feature_dimension == 0 and not any(
            f is not None and f.shape[-1] != 0 for f in edge_features)
        p = self.__dropout.p if self.training else 0.0
        M, T = self._message_dimension, len(adjacency_lists)
        if (no_feats and node_states.dtype == torch.float32 and _edge_training_ok(self.__state_dimension, M)
                and (p > 0 or _prefer_edge_path(plan.num_edges, num_nodes, T, self.__state_dimension, M))):
            # training, edge form: grouped per-edge GEMM with the reference's per-edge input dropout
            # folded in (forward + both gradients on HIP), HIP segment reduce, torch GRU cell
            w_stack = _scoped(self, "edge_w", lambda: torch.stack(
                [l.weight for l in self.__edge_message_transformation_layers]))
            msgs = edge_linear_autograd(node_states, plan, w_stack, False, p, _dropout_seed() if p > 0 else 0)
            agg = segment_reduce(msgs, plan, self.__aggregation_fn)
            return dense.gru_cell(gru, agg, node_states)

        if self._table_ok(node_states, edge_features):
            # training without per-edge dropout, few edge types: differentiable HIP GEMM nodes (ptgnn_amd/dense.py)
            # for the dense blocks, the HIP kernel (forward + backward) for the aggregation
            w = torch.cat([l.weight for l in self.__edge_message_transformation_layers], dim=0)
            y = dense.linear(node_states, w)
            agg = gather_reduce_autograd(y, None, plan, self._message_dimension, self.__aggregation_fn)
            return dense.gru_cell(gru, agg, node_states)

        if (not no_feats and p == 0.0 and self.__aggregation_fn in ops.REDUCE_IDS
                and _feat_gemm_ok(node_states, edge_features, self.__state_dimension, M, *self.parameters())):
            # inference with edge features: fused gather of [x[src] | features] inside the grouped GEMM
            msgs = ops.edge_linear(node_states, adjacency_lists,
                                   [l.weight for l in self.__edge_message_transformation_layers], False,
                                   edge_feats=edge_features)
            agg = ops.gather_reduce(msgs, plan, M, self.__aggregation_fn, type_bits=0, col=plan.perm)
            return ops.gru_cell(agg, node_states, gru.weight_ih, gru.weight_hh, gru.bias_ih, gru.bias_hh)

        if (not no_feats and p == 0.0 and self.__aggregation_fn in ops.REDUCE_IDS and _edge_training_ok(self.__state_dimension, M)
                and _feat_gemm_ok(node_states, edge_features, self.__state_dimension, M, shapes_only=True)):
            # training with edge features: the same grouped GEMM as one autograd node (scatter._EdgeLinearFeat) -- no
            # index_select, no [E, H + F] concat; weight / feature / state gradients on the HIP kernels
            msgs = edge_linear_feat_autograd(node_states, plan, [l.weight for l in self.__edge_message_transformation_layers],
                                             False, edge_features)
            agg = segment_reduce(msgs, plan, self.__aggregation_fn)
            return dense.gru_cell(gru, agg, node_states)

        # general per-edge path (per-edge dropout with edge features, odd widths): message order = type-major
        all_messages = []
        for (src, _), feats, lin in zip(adjacency_lists, edge_features,
                                        self.__edge_message_transformation_layers):
            inp = node_states.index_select(0, src)
            if feats is not None and feats.shape[-1] > 0:
                inp = torch.cat([inp, feats.to(inp.dtype)], dim=-1)
            all_messages.append(dense.linear(self.__dropout(inp), lin.weight))     # HIP GEMM, K = H + F
        messages = torch.cat(all_messages, dim=0)
        agg = segment_reduce(messages, plan, self.__aggregation_fn)
        return dense.gru_cell(gru, agg, node_states)

    def forward_sharded(self, node_states: torch.Tensor, shard) -> torch.Tensor:
        """One layer over a dst-range shard (ptgnn_amd/sharded.py): `node_states` are this rank's rows; one
        all-to-all of halo rows, then the same kernels as `forward`.  Form per minibatch, like `forward`:
          * edge form (many sparse edge types, or training with per-edge dropout): the grouped per-edge GEMM
            gathers its A rows from the local table [own | halo] through the remapped adjacency lists;
          * table form: per-node pre-transform; the rows that travel are message-table rows when T*M <= H,
            else node states (pre-transformed after arrival: the weights are replicated)."""
        _check_device(node_states)
        feats = [None] * len(shard.local_adj)
        T = len(shard.local_adj)
        assert T == len(self.__edge_message_transformation_layers)
        if self._edge_feature_dimension != 0 or node_states.dtype != torch.float32:
            raise _lib.PtgnnAmdError("forward_sharded: edge features / non-fp32 states are not supported on a "
                                     "sharded graph")
        if self.__aggregation_fn not in ops.REDUCE_IDS:
            raise _lib.PtgnnAmdError(f"forward_sharded: aggregation {self.__aggregation_fn!r} is not supported on a "
                                     "sharded graph (sum / mean / max / min are)")
        M, H = self._message_dimension, self.__state_dimension
        gru = self.__state_update
        p = self.__dropout.p if self.training else 0.0
        ws = [l.weight for l in self.__edge_message_transformation_layers]
        edge_form = _prefer_edge_path(*shard.form_sizes(T * M > H), T, H, M)   # one decision for the whole group
        if not self._fused_ok(node_states, feats):
            # training.  Edge form: differentiable halo exchange (backward = transposed all-to-all + HIP
            # segment-sum) -> grouped per-edge GEMM node with the hash dropout folded in -> HIP segment reduce
            if _edge_training_ok(H, M) and (p > 0 or edge_form):
                table = shard.exchange_autograd(node_states)
                w_stack = _scoped(self, "edge_w", lambda: torch.stack(ws))
                seed = (_dropout_seed() ^ (0x9E3779B97F4A7C15 * (shard.rank + 1) & (2 ** 62 - 1))) if p > 0 else 0
                msgs = edge_linear_autograd(table, shard.plan, w_stack, False, p, seed)
                agg = segment_reduce(msgs, shard.plan, self.__aggregation_fn)
                return dense.gru_cell(gru, agg, node_states)
            if p > 0:
                raise _lib.PtgnnAmdError("forward_sharded: per-edge dropout needs state and message widths that "
                                         "are multiples of 32")
            w = torch.cat(ws, dim=0)
            if w.shape[0] <= w.shape[1]:
                y = shard.exchange_autograd(dense.linear(node_states, w))
            else:
                y = dense.linear(shard.exchange_autograd(node_states), w)
            agg = gather_reduce_autograd(y, None, shard.plan, M, self.__aggregation_fn)
            return dense.gru_cell(gru, agg, node_states)
        if shard.overlap and self.__aggregation_fn in _overlap_reduces():
            # two-block mode: the own-source block aggregates while the halo rows travel (sharded.py)
            from ptgnn_amd import sharded
            n = shard.n_local
            if edge_form or T * M > H:      # node states travel
                table = shard.new_table(H, node_states)
                table[:n].copy_(node_states)
                work = shard.begin_exchange(table)
                if edge_form:
                    def table_of(block):
                        adj, plan = (shard.adj_own, shard.plan_own) if block == "own" else (shard.adj_halo, shard.plan_halo)
                        return _edge_messages(table, adj, plan, ws) + (0,)
                else:
                    w = self._stacked_edge_weights()
                    y = shard.new_table(T * M, node_states)
                    ops.linear(node_states, w, out=y[:n])

                    def table_of(block):
                        if block == "halo":
                            ops.linear(table[n:], w, out=y[n:])
                        return y, None, None
            else:                           # message-table rows travel
                y = shard.new_table(T * M, node_states)
                ops.linear(node_states, self._stacked_edge_weights(), out=y[:n])
                work = shard.begin_exchange(y)

                def table_of(block):
                    return y, None, None
            agg = sharded.aggregate_two_blocks(shard, work, table_of, M, self.__aggregation_fn)
        elif edge_form:
            table = shard.exchange(node_states)
            msgs, col = _edge_messages(table, shard.local_adj, shard.plan, ws)
            agg = ops.gather_reduce(msgs, shard.plan, M, self.__aggregation_fn, type_bits=0, col=col)
        else:
            w = self._stacked_edge_weights()
            if T * M <= H:   # ship message-table rows: no wider than the state, and no duplicated GEMM work
                y = shard.new_table(T * M, node_states)
                ops.linear(node_states, w, out=y[: shard.n_local])
                shard.exchange_into(y)
            else:            # ship node states, pre-transform own + halo rows locally
                y = ops.linear(shard.exchange(node_states), w)
            agg = ops.gather_reduce(y, shard.plan, M, self.__aggregation_fn)
        return ops.gru_cell(agg, node_states, gru.weight_ih, gru.weight_hh, gru.bias_ih, gru.bias_hh)

    @property
    def input_state_dimension(self) -> int:
        return self.__state_dimension

    @property
    def output_state_dimension(self) -> int:
        return self.__state_dimension

    def export_weights(self) -> Dict:
        """Weights in the layout the parity oracle consumes (tests only read this)."""
        gru = self.__state_update
        return {"kind": "ggnn",
                "edge_w": [l.weight.detach().cpu() for l in self.__edge_message_transformation_layers],
                "w_ih": gru.weight_ih.detach().cpu(), "w_hh": gru.weight_hh.detach().cpu(),
                "b_ih": gru.bias_ih.detach().cpu(), "b_hh": gru.bias_hh.detach().cpu(),
                "agg": self.__aggregation_fn}


# ------------------------------------------------------------------------------------------------
# MLP message passing layer
# ------------------------------------------------------------------------------------------------
class MlpMessagePassingLayer(AbstractMessagePassingLayer):
    """Constructor and semantics of mlpmessagepassing.py:12-125."""

    def __init__(self, input_state_dimension: int, output_state_dimension: int,
                 message_dimension: int, num_edge_types: int,
                 message_aggregation_function: Union[str, AbstractMessageAggregation],
                 message_activation: Optional[nn.Module] = nn.GELU(),
                 use_target_state_as_message_input: bool = True,
                 mlp_hidden_layers: Union[List[int], int] = 0, use_layer_norm: bool = True,
                 use_dense_layer: bool = True, dropout_rate: float = 0.0,
                 dense_activation: Optional[nn.Module] = nn.Tanh(), features_dimension: int = 0):
        super().__init__()
        self.__input_state_dim = input_state_dimension
        self.__use_target_state_as_message_input = use_target_state_as_message_input
        self.__output_state_dim = output_state_dimension
        msg_in = (2 if use_target_state_as_message_input else 1) * input_state_dimension
        self.__edge_message_transformation_layers = nn.ModuleList(
            [MLP(input_dimension=msg_in + features_dimension, output_dimension=message_dimension,
                 hidden_layers=mlp_hidden_layers) for _ in range(num_edge_types)])
        self.__aggregation_fn = message_aggregation_function
        if isinstance(message_aggregation_function, str):
            agg_size = message_dimension
        else:
            agg_size = message_aggregation_function.output_state_size(message_dimension)
        self.__message_activation = message_activation
        upd: List[nn.Module] = []
        ln = dense = act = None
        if use_layer_norm:
            ln = nn.LayerNorm(agg_size)
            upd.append(ln)
        if use_dense_layer:
            dense = nn.Linear(agg_size, output_state_dimension)
            nn.init.xavier_uniform_(dense.weight)
            upd.append(dense)
            if dense_activation is not None:
                act = dense_activation
                upd.append(act)
        drop = nn.Dropout(p=dropout_rate)
        upd.append(drop)
        # registered ONCE, under the reference's name => identical state_dict keys; the aliases below
        # bypass nn.Module registration on purpose
        self.__state_update = nn.Sequential(*upd)
        for name, mod in (("_ln", ln), ("_dense", dense), ("_dense_act", act), ("_dropout", drop)):
            object.__setattr__(self, name, mod)
        self._message_dimension = message_dimension
        self._features_dimension = features_dimension

    # -- fused path -------------------------------------------------------------------------
    def _stacked_edge_weights(self) -> torch.Tensor:
        """[(1|2)*T*M, H]: source halves of every type, then (with target state) the target halves; rebuilt
        per call (see GatedMessagePassingLayer._stacked_edge_weights)."""
        ws = [m.linears[0].weight.detach() for m in self.__edge_message_transformation_layers]
        H = self.__input_state_dim
        parts = [w[:, :H] for w in ws]
        if self.__use_target_state_as_message_input:
            parts += [w[:, H:2 * H] for w in ws]
        return torch.cat(parts, dim=0).contiguous()

    def _table_ok(self, node_states, edge_features) -> bool:
        if not isinstance(self.__aggregation_fn, str) or self.__aggregation_fn not in ops.REDUCE_IDS:
            return False                                     # aggregation modules; "mul": the seam's own kernel
        if self._features_dimension != 0 or any(f is not None and f.shape[-1] != 0 for f in edge_features):
            return False
        if not all(m.is_single_linear for m in self.__edge_message_transformation_layers):
            return False
        return node_states.dtype == torch.float32

    def _fused_ok(self, node_states, edge_features) -> bool:
        if not self._table_ok(node_states, edge_features):
            return False
        return (not torch.is_grad_enabled()) or _no_grad_needed(node_states, *self.parameters())

    def _aggregate_and_update(self, ysrc, ydst, plan, col=None, type_bits=None, two_block=None) -> torch.Tensor:
        """Fused gather/reduce with GELU + LayerNorm folded into the kernel epilogue when the
        layer's activation/normalisation are the stock ones, then the dense update.  `two_block` =
        (shard, work, table_of): the sharded two-block aggregation instead (epilogue in its combine pass)."""
        M = self._message_dimension
        act = self.__message_activation
        gelu_ok = act is None or (isinstance(act, nn.GELU) and getattr(act, "approximate", "none") == "none")
        ln_ok = self._ln is None or (M <= 512 and self._ln.elementwise_affine and self._ln.bias is not None)
        fused = gelu_ok and ln_ok
        epi = {}
        if fused:
            epi = dict(epilogue=(ops.EPI_GELU if act is not None else 0) | (ops.EPI_LAYERNORM if self._ln is not None else 0),
                       ln_weight=self._ln.weight if self._ln is not None else None,
                       ln_bias=self._ln.bias if self._ln is not None else None,
                       ln_eps=self._ln.eps if self._ln is not None else 1e-5)
        if two_block is not None:
            from ptgnn_amd import sharded
            shard, work, table_of = two_block
            agg = sharded.aggregate_two_blocks(shard, work, table_of, M, self.__aggregation_fn, ydst=ydst, **epi)
        elif (fused and ydst is None and self._dense is not None
              and (self._dense_act is None or isinstance(self._dense_act, (nn.Tanh, nn.ReLU)))
              and not (self.training and getattr(self._dropout, "p", 0.0) > 0)
              and _no_grad_needed(ysrc, *self._dense.parameters())
              and ops.gather_update_supported(M, self._dense.out_features, plan)):
            # hidden 64 (the README's default architecture, BASELINE config 4), edge form: aggregation, GELU, LayerNorm,
            # Linear and Tanh in ONE launch -- the [N, M] aggregate never exists in memory (gather_reduce.hip)
            act = "tanh" if isinstance(self._dense_act, nn.Tanh) else ("relu" if isinstance(self._dense_act, nn.ReLU) else None)
            # (the fused launch stores 16-byte vectors: a hint whose rows are not 16-byte aligned -- the right half of a
            # concat buffer whose left half is not a multiple of 4 wide -- is declined, the residual then concatenates)
            hint = _take_output_hint(plan.num_nodes, self._dense.out_features, ysrc, vector_stores=True)
            out = ops.gather_update(ysrc, plan, self.__aggregation_fn, plan.col if col is None else col,
                                    plan.type_bits if type_bits is None else type_bits, epi.get("epilogue", 0),
                                    epi.get("ln_weight"), epi.get("ln_bias"), epi.get("ln_eps", 1e-5),
                                    self._dense.weight, self._dense.bias, act, out=hint)
            return self._dropout(out)
        else:
            agg = ops.gather_reduce(ysrc, plan, M, self.__aggregation_fn, ydst=ydst, col=col, type_bits=type_bits, **epi)
        return self._update(agg, fused)

    def forward_sharded(self, node_states: torch.Tensor, shard) -> torch.Tensor:
        """One layer over a dst-range shard (ptgnn_amd/sharded.py).  Table form: the source term rides the halo
        all-to-all, the destination term W_t^d x_v is purely local.  Edge form (many sparse edge types): the
        grouped per-edge GEMM reads source AND destination rows from the local table [own | halo]
        (destinations are always own rows)."""
        _check_device(node_states)
        feats = [None] * len(shard.local_adj)
        assert len(shard.local_adj) == len(self.__edge_message_transformation_layers)
        T, M, H = len(shard.local_adj), self._message_dimension, self.__input_state_dim
        if not self._table_ok(node_states, feats):
            raise _lib.PtgnnAmdError("forward_sharded needs a string aggregation and single-Linear edge "
                                     "transforms without edge features")
        use_dst = self.__use_target_state_as_message_input
        ws = [m.linears[0].weight for m in self.__edge_message_transformation_layers]
        edge_form = _prefer_edge_path(*shard.form_sizes(T * M > H), T, H, M)   # one decision for the whole group
        if not self._fused_ok(node_states, feats):
            if edge_form and _edge_training_ok(H, M):
                table = shard.exchange_autograd(node_states)
                w_stack = _scoped(self, "edge_w", lambda: torch.stack(ws))
                msgs = edge_linear_autograd(table, shard.plan, w_stack, use_dst)
                return self._update(segment_reduce(msgs, shard.plan, self.__aggregation_fn), False)
            w_src = torch.cat([w[:, :H] for w in ws], dim=0)
            if T * M <= H:
                ysrc = shard.exchange_autograd(dense.linear(node_states, w_src))
            else:
                ysrc = dense.linear(shard.exchange_autograd(node_states), w_src)
            ydst = None
            if use_dst:
                ydst = dense.linear(node_states, torch.cat([w[:, H:2 * H] for w in ws], dim=0))
            agg = gather_reduce_autograd(ysrc, ydst, shard.plan, M, self.__aggregation_fn)
            return self._update(agg, False)
        if shard.overlap and self.__aggregation_fn in _overlap_reduces():
            n = shard.n_local
            w = self._stacked_edge_weights()
            ydst = None
            if edge_form or T * M > H:      # node states travel
                table = shard.new_table(H, node_states)
                table[:n].copy_(node_states)
                work = shard.begin_exchange(table)
                if edge_form:
                    def table_of(block):
                        adj, plan = (shard.adj_own, shard.plan_own) if block == "own" else (shard.adj_halo, shard.plan_halo)
                        return ops.edge_linear(table, adj, ws, use_dst), plan.perm, 0
                else:
                    ysrc = shard.new_table(T * M, node_states)
                    ops.linear(node_states, w[: T * M], out=ysrc[:n])

                    def table_of(block):
                        if block == "halo":
                            ops.linear(table[n:], w[: T * M], out=ysrc[n:])
                        return ysrc, None, None
            else:                           # message-table rows travel
                ysrc = shard.new_table(T * M, node_states)
                ops.linear(node_states, w[: T * M], out=ysrc[:n])
                work = shard.begin_exchange(ysrc)

                def table_of(block):
                    return ysrc, None, None
            if use_dst and not edge_form:
                ydst = ops.linear(node_states, w[T * M:])
            return self._aggregate_and_update(None, ydst, None, two_block=(shard, work, table_of))
        if edge_form:
            table = shard.exchange(node_states)
            msgs = ops.edge_linear(table, shard.local_adj, ws, use_dst)
            return self._aggregate_and_update(msgs, None, shard.plan, col=shard.plan.perm, type_bits=0)
        w = self._stacked_edge_weights()
        w_src = w[: T * M]
        if T * M <= H:
            ysrc = shard.new_table(T * M, node_states)
            ops.linear(node_states, w_src, out=ysrc[: shard.n_local])
            shard.exchange_into(ysrc)
        else:
            ysrc = ops.linear(shard.exchange(node_states), w_src)
        ydst = ops.linear(node_states, w[T * M:]) if use_dst else None
        return self._aggregate_and_update(ysrc, ydst, shard.plan)

    def _update(self, agg: torch.Tensor, fused_epilogue_done: bool) -> torch.Tensor:
        x = agg
        if not fused_epilogue_done:
            act = self.__message_activation
            stock_act = act is None or (isinstance(act, nn.GELU) and getattr(act, "approximate", "none") == "none")
            stock_ln = self._ln is None or (self._ln.elementwise_affine and self._ln.bias is not None
                                            and tuple(self._ln.normalized_shape) == (x.shape[-1],))
            if (stock_act and stock_ln and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2
                    and x.shape[1] <= 512 and (act is not None or self._ln is not None)):
                # training: the same GELU -> LayerNorm arithmetic as the fused inference epilogue, as one HIP
                # autograd node (forward + backward kernels of csrc/row_epilogue.hip)
                x = dense.row_epilogue(x, act is not None, self._ln)
            else:
                if act is not None:
                    x = act(x)
                if self._ln is not None:
                    x = self._ln(x)
        if self._dense is not None:
            if _no_grad_needed(x, *self._dense.parameters()):
                tanh = isinstance(self._dense_act, nn.Tanh)
                hint = None
                if (tanh or self._dense_act is None) and not (self.training and getattr(self._dropout, "p", 0.0) > 0):
                    # nothing follows the GEMM: it may write straight into the right half of a concat residual's result
                    hint = _take_output_hint(x.shape[0], self._dense.out_features, x)
                x = ops.linear(x, self._dense.weight, self._dense.bias, act="tanh" if tanh else None, out=hint)
                if self._dense_act is not None and not tanh:
                    x = self._dense_act(x)
            else:
                act = self._dense_act
                name = "tanh" if isinstance(act, nn.Tanh) else ("relu" if isinstance(act, nn.ReLU) else None)
                if (act is None or name is not None) and isinstance(self._dropout, nn.Dropout):
                    # training: Dropout(act(Linear(x))) as one autograd node (dense._LinearActDropout)
                    fused = dense.linear_act_dropout(x, self._dense.weight, self._dense.bias, name, self._dropout.p,
                                                     self.training)
                    if fused is not None:
                        return fused
                x = dense.linear(x, self._dense.weight, self._dense.bias)
                if act is not None:
                    x = act(x)
        return self._dropout(x)

    def forward(self, node_states: torch.Tensor, adjacency_lists: Adj, node_to_graph_idx,
                reference_node_ids: Dict[str, torch.Tensor],
                reference_node_graph_idx: Dict[str, torch.Tensor],
                edge_features: List[torch.Tensor]) -> torch.Tensor:
        assert len(adjacency_lists) == len(self.__edge_message_transformation_layers), \
            "The number of adjacency lists must be equal to the number of edge types."
        if _on_host(node_states):
            return torch_route.mlp_layer(node_states, adjacency_lists, edge_features,
                                         list(self.__edge_message_transformation_layers),
                                         self.__use_target_state_as_message_input, self.__aggregation_fn,
                                         self.__message_activation, self.__state_update)
        if node_states.dtype in _AMP_DTYPES:   # see GatedMessagePassingLayer.forward
            feats = [f.float() if f is not None and f.dtype in _AMP_DTYPES else f for f in edge_features]
            return self.forward(node_states.float(), adjacency_lists, node_to_graph_idx, reference_node_ids,
                                reference_node_graph_idx, feats).to(node_states.dtype)
        num_nodes = node_states.shape[0]
        T, M = len(adjacency_lists), self._message_dimension

        if self._fused_ok(node_states, edge_features):
            plan = ops.plan_for(adjacency_lists, num_nodes)
            if _prefer_edge_path(plan.num_edges, num_nodes, T, self.__input_state_dim, M):
                msgs = ops.edge_linear(node_states, adjacency_lists,
                                       [m.linears[0].weight for m in self.__edge_message_transformation_layers],
                                       self.__use_target_state_as_message_input)
                return self._aggregate_and_update(msgs, None, plan, col=plan.perm, type_bits=0)
            y = ops.linear(node_states, self._stacked_edge_weights())
            ysrc = y[:, :T * M]
            ydst = y[:, T * M:] if self.__use_target_state_as_message_input else None
            return self._aggregate_and_update(ysrc, ydst, plan)

        if self._table_ok(node_states, edge_features):
            plan = ops.plan_for(adjacency_lists, num_nodes)
            H = self.__input_state_dim
            ws = [m.linears[0].weight for m in self.__edge_message_transformation_layers]
            if _edge_training_ok(H, M) and _prefer_edge_path(plan.num_edges, num_nodes, T, H, M):
                # training, many sparse edge types: grouped per-edge GEMM forward + backward on HIP
                # (the edge MLPs of this layer carry no dropout: mlpmessagepassing.py:39-47)
                w_stack = _scoped(self, "edge_w", lambda: torch.stack(ws))
                msgs = edge_linear_autograd(node_states, plan, w_stack, self.__use_target_state_as_message_input)
                return self._update(segment_reduce(msgs, plan, self.__aggregation_fn), False)
            # training: dense blocks through torch autograd, aggregation fwd + bwd on the HIP kernel
            parts = [w[:, :H] for w in ws]
            if self.__use_target_state_as_message_input:
                parts += [w[:, H:2 * H] for w in ws]
            y = dense.linear(node_states, torch.cat(parts, dim=0))
            ysrc = y[:, :T * M]
            ydst = y[:, T * M:] if self.__use_target_state_as_message_input else None
            agg = gather_reduce_autograd(ysrc, ydst, plan, M, self.__aggregation_fn)
            return self._update(agg, False)

        mlps = list(self.__edge_message_transformation_layers)
        first = [m.linears[0] for m in mlps]
        if (any(f is not None and f.shape[-1] != 0 for f in edge_features) and all(l.bias is None for l in first)
                and isinstance(self.__aggregation_fn, str)
                and _feat_gemm_ok(node_states, edge_features, self.__input_state_dim, first[0].weight.shape[0],
                                  *self.parameters())
                and not (self.training and any(isinstance(m, nn.Dropout) and m.p > 0 for e in mlps
                                               for m in e.modules_in_order))):
            # inference with edge features: the FIRST Linear of every edge MLP as one grouped GEMM that gathers
            # [x[src] | x[dst] | features] itself; the rest of the MLP (activation, further Linears) on its [E_t, .] rows
            plan = ops.plan_for(adjacency_lists, num_nodes)
            hid = ops.edge_linear(node_states, adjacency_lists, [l.weight for l in first],
                                  self.__use_target_state_as_message_input, edge_feats=edge_features)
            if all(m.is_single_linear for m in mlps):
                messages = hid
            else:
                outs, off = [], 0
                for (src, _), edge_mlp in zip(adjacency_lists, mlps):
                    n = int(src.shape[0])
                    mods = edge_mlp.modules_in_order
                    rest = mods[next(i for i, m in enumerate(mods) if isinstance(m, nn.Linear)) + 1:]
                    h = hid[off:off + n]
                    for m in rest:
                        h = dense.linear(h, m.weight, m.bias) if isinstance(m, nn.Linear) else m(h)
                    outs.append(h)
                    off += n
                messages = torch.cat(outs, dim=0)
            if self.__aggregation_fn in ops.REDUCE_IDS and messages.shape[1] % 4 == 0:
                return self._aggregate_and_update(messages, None, plan, col=plan.perm, type_bits=0)
            return self._update(segment_reduce(messages, plan, self.__aggregation_fn), False)

        no_feats = self._features_dimension == 0 and not any(f is not None and f.shape[-1] != 0 for f in edge_features)
        H = self.__input_state_dim
        if ((not no_feats) and isinstance(self.__aggregation_fn, str) and all(l.bias is None for l in first)
                and len({tuple(l.weight.shape) for l in first}) == 1 and _edge_training_ok(H, first[0].weight.shape[0])
                and _feat_gemm_ok(node_states, edge_features, H, first[0].weight.shape[0], shapes_only=True)
                and not any(isinstance(m, nn.Dropout) and m.p > 0 and self.training for e in mlps for m in e.modules_in_order)):
            # training with edge features: the first Linear of every edge MLP as ONE differentiable grouped GEMM that gathers
            # [x[src] | x[dst] | features] itself (scatter._EdgeLinearFeat); the rest of each MLP on the type's rows
            plan = ops.plan_for(adjacency_lists, num_nodes)
            hid = edge_linear_feat_autograd(node_states, plan, [l.weight for l in first],
                                            self.__use_target_state_as_message_input, edge_features)
            outs, off = [], 0
            for (src, _), edge_mlp in zip(adjacency_lists, mlps):
                n = int(src.shape[0])
                mods = edge_mlp.modules_in_order
                h = hid[off:off + n]
                for m in mods[next(i for i, m in enumerate(mods) if isinstance(m, nn.Linear)) + 1:]:
                    h = dense.linear(h, m.weight, m.bias) if isinstance(m, nn.Linear) else m(h)
                outs.append(h)
                off += n
            messages = outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)
            return self._update(segment_reduce(messages, plan, self.__aggregation_fn), False)
        if (no_feats and isinstance(self.__aggregation_fn, str) and node_states.dtype == torch.float32
                and all(l.bias is None for l in first) and len({tuple(l.weight.shape) for l in first}) == 1
                and _edge_training_ok(H, first[0].weight.shape[0])
                and not any(isinstance(m, nn.Dropout) and m.p > 0 and self.training for e in mlps for m in e.modules_in_order)):
            # deeper edge MLPs (mlp_hidden_layers > 0, mlp.py:50-77) without edge features: the FIRST Linear of every edge
            # type is the grouped per-edge GEMM (it gathers [x[src] | x[dst]] itself: no index_select, no [E, 2H] concat --
            # inference and training alike, `_EdgeLinear` carries the gradients); the rest of each MLP runs on the type's
            # [E_t, hidden] rows on the HIP Linear
            plan = ops.plan_for(adjacency_lists, num_nodes)
            grad = not _no_grad_needed(node_states, *self.parameters())
            use_dst = self.__use_target_state_as_message_input
            if grad:
                w_stack = _scoped(self, "edge_w0", lambda: torch.stack([l.weight for l in first]))
                hid = edge_linear_autograd(node_states, plan, w_stack, use_dst)
            else:
                hid = ops.edge_linear(node_states, adjacency_lists, [l.weight for l in first], use_dst)
            out_dim = mlps[0].linears[-1].out_features
            messages = None if grad else torch.empty(hid.shape[0], out_dim, dtype=torch.float32, device=hid.device)
            outs, off = [], 0
            for (src, _), edge_mlp in zip(adjacency_lists, mlps):
                n = int(src.shape[0])
                mods = edge_mlp.modules_in_order
                rest = mods[next(i for i, m in enumerate(mods) if isinstance(m, nn.Linear)) + 1:]
                h = hid[off:off + n]
                last = max((i for i, m in enumerate(rest) if isinstance(m, nn.Linear)), default=-1)
                for i, m in enumerate(rest):
                    if not isinstance(m, nn.Linear):
                        h = m(h)
                    elif grad:
                        h = dense.linear(h, m.weight, m.bias)
                    else:   # the type's last Linear writes its block of the message matrix in place: no concat
                        h = ops.linear(h, m.weight, m.bias, out=messages[off:off + n] if i == last and n > 0 else None)
                if grad:
                    outs.append(h)
                elif last < 0 and n > 0:
                    messages[off:off + n].copy_(h)
                off += n
            if grad:
                messages = torch.cat(outs, dim=0)
            if (not grad) and self.__aggregation_fn in ops.REDUCE_IDS and messages.shape[1] % 4 == 0:
                return self._aggregate_and_update(messages, None, plan, col=plan.perm, type_bits=0)
            return self._update(segment_reduce(messages, plan, self.__aggregation_fn), False)

        # general per-edge path
        all_targets, all_messages = [], []
        for (src, dst), feats, edge_mlp in zip(adjacency_lists, edge_features,
                                               self.__edge_message_transformation_layers):
            all_targets.append(dst)
            inp = node_states.index_select(0, src)
            if self.__use_target_state_as_message_input:
                inp = torch.cat([inp, node_states.index_select(0, dst)], dim=-1)
            if feats is not None and feats.shape[-1] > 0:
                inp = torch.cat([inp, feats.to(inp.dtype)], dim=-1)
            all_messages.append(_run_mlp(edge_mlp, inp))                           # HIP GEMMs
        messages = torch.cat(all_messages, dim=0)
        if isinstance(self.__aggregation_fn, str):
            plan = ops.plan_for(adjacency_lists, num_nodes)
            agg = segment_reduce(messages, plan, self.__aggregation_fn)
        else:
            agg = self.__aggregation_fn(messages=messages, message_targets=torch.cat(all_targets, dim=0),
                                        num_nodes=num_nodes)
        return self._update(agg, False)

    @property
    def input_state_dimension(self) -> int:
        return self.__input_state_dim

    @property
    def output_state_dimension(self) -> int:
        return self.__output_state_dim

    def export_weights(self) -> Dict:
        c = lambda t: None if t is None else t.detach().cpu()  # noqa: E731
        return {"kind": "mlp",
                "edge_mlp": [[l.weight.detach().cpu() for l in m.linears]
                             for m in self.__edge_message_transformation_layers],
                "use_target": self.__use_target_state_as_message_input, "agg": self.__aggregation_fn,
                "gelu": self.__message_activation is not None,
                "ln_w": c(self._ln.weight) if self._ln is not None else None,
                "ln_b": c(self._ln.bias) if self._ln is not None else None,
                "dense_w": c(self._dense.weight) if self._dense is not None else None,
                "dense_b": c(self._dense.bias) if self._dense is not None else None,
                "tanh": self._dense_act is not None}


# ------------------------------------------------------------------------------------------------
# residual layers (residuallayers.py): elementwise glue that every shipped stack uses
# ------------------------------------------------------------------------------------------------
class _ResidualOriginLayer(AbstractMessagePassingLayer):
    def __init__(self, input_dim: int, target_layer):
        super().__init__()
        self.__target_layer = target_layer   # registered like the reference (same state_dict keys)
        self.__input_dim = input_dim

    def forward(self, node_states, adjacency_lists, node_to_graph_idx, reference_node_ids,
                reference_node_graph_idx, edge_features):
        self.__target_layer._original_input = node_states
        return node_states

    @property
    def input_state_dimension(self) -> int:
        return self.__input_dim

    @property
    def output_state_dimension(self) -> int:
        return self.__input_dim


class _ResidualBase(AbstractMessagePassingLayer):
    def __init__(self, input_dim: int):
        super().__init__()
        self._original_input = None
        self._input_dim = input_dim

    def pass_through_dummy_layer(self) -> _ResidualOriginLayer:
        return _ResidualOriginLayer(self._input_dim, target_layer=self)

    def _pop(self) -> torch.Tensor:
        assert self._original_input is not None, "Initial Pass Through Layer was not used."
        x, self._original_input = self._original_input, None
        return x

    @property
    def input_state_dimension(self) -> int:
        return self._input_dim


class MeanResidualLayer(_ResidualBase):
    def forward(self, node_states, adjacency_lists, node_to_graph_idx, reference_node_ids,
                reference_node_graph_idx, edge_features):
        return (self._pop() + node_states) * 0.5   # == stack(..).mean(-1) bit for bit (/2 is exact)

    @property
    def output_state_dimension(self) -> int:
        return self._input_dim


class ConcatResidualLayer(_ResidualBase):
    def forward(self, node_states, adjacency_lists, node_to_graph_idx, reference_node_ids,
                reference_node_graph_idx, edge_features):
        orig = self._pop()
        buf = getattr(node_states, "_ptgnn_amd_concat_buffer", None)
        if (buf is not None and buf.shape[1] == orig.shape[1] + node_states.shape[1] and buf.dtype == orig.dtype
                and node_states.data_ptr() == buf[:, orig.shape[1]:].data_ptr()):
            # the previous layer wrote straight into the right half of the result (see set_output_hint)
            buf[:, : orig.shape[1]].copy_(orig)
            return buf
        return torch.cat((orig, node_states), dim=-1)

    def make_result_buffer(self, num_nodes: int, next_dim: int, like: torch.Tensor) -> torch.Tensor:
        return torch.empty(num_nodes, self._input_dim + next_dim, dtype=like.dtype, device=like.device)

    @property
    def output_state_dimension(self) -> int:
        return 2 * self._input_dim


class LinearResidualLayer(_ResidualBase):
    def __init__(self, state_dimension1: int, state_dimension2: int, target_state_size: int,
                 dropout_rate: float = 0.0):
        super().__init__(state_dimension1)
        self.__input_dim2 = state_dimension2
        self.__linear_combination = nn.Linear(state_dimension1 + state_dimension2, target_state_size,
                                              bias=False)
        self.__dropout = nn.Dropout(p=dropout_rate)

    def forward(self, node_states, adjacency_lists, node_to_graph_idx, reference_node_ids,
                reference_node_graph_idx, edge_features):
        x = torch.cat((self._pop(), node_states), dim=-1)
        lin = self.__linear_combination
        if _on_host(x):
            return self.__dropout(lin(x))
        if x.dtype in _AMP_DTYPES:
            return self.__dropout(dense.linear(x.float(), lin.weight).to(x.dtype))
        if _no_grad_needed(x, lin.weight):
            return self.__dropout(ops.linear(x, lin.weight))
        return self.__dropout(dense.linear(x, lin.weight))

    @property
    def input_state_dimension(self) -> int:
        return self.__input_dim2

    @property
    def output_state_dimension(self) -> int:
        return self.__linear_combination.out_features
