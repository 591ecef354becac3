"""The HOST-TENSOR route of the drop-in layers and of the `torch_scatter` facade: plain torch operators, for tensors
that live on the CPU.

Why it exists.  The reference's own inference entry restores a model and runs it on "cpu"
(ptgnn/implementations/typilus/predict.py:25-27), and its trainer runs on the CPU when no GPU is visible
(ptgnn/baseneuralmodel/trainer.py:392-395).  A checkpoint whose layers are `ptgnn_amd` classes has to survive
both, so the layers dispatch on the DEVICE OF THEIR INPUT exactly like a torch operator does:

    * tensors on the GPU  -> libptgnn_amd.so (HIP kernels).  ALWAYS.  A missing / stale library or an entry
      point that fails raises `PtgnnAmdError`; nothing on that side ever reaches this file (every function here
      starts with `_host_only`, which raises for a device tensor), so a GPU run can not silently become a torch run.
    * tensors on the CPU  -> this file.

This is product code (it ships in `ptgnn_amd/`), written against the reference's semantics
(gatedmessagepassing.py:37-69, mlpmessagepassing.py:68-117, abstractmessagepassing.py:38-50,
globalgraphexchange.py:29-45, residuallayers.py) and torch_scatter 2.0.x's published behaviour; it imports nothing
from `oracle/` (the test-side restatement), and `tests/dropin_check.py` compares it with the reference's own
layers on the reference's own container and batcher.
"""
from typing import List, Optional, Sequence, Tuple

import torch
from torch import nn

from ptgnn_amd import _lib


def _host_only(*tensors) -> None:
    for t in tensors:
        if t is not None and t.is_cuda:
            raise _lib.PtgnnAmdError("ptgnn_amd.torch_route is the CPU-tensor route; a GPU tensor must take the HIP "
                                     "kernels (this is a bug in the caller, not a fallback)")


# ------------------------------------------------------------------------------------------------------------------
# segment reduce with torch_scatter's semantics over rows [E, D] and a 1-D int64 index
# ------------------------------------------------------------------------------------------------------------------
class _SegmentProduct(torch.autograd.Function):
    """reduce="mul": product per segment in element order, empty segments 1; backward as torch_scatter's
    ScatterMul: (grad_out * out)[index] / src with 0 / 0 -> 0."""

    @staticmethod
    def forward(ctx, rows, index, n):
        out = torch.ones(n, rows.shape[1], dtype=rows.dtype).scatter_reduce_(
            0, index.unsqueeze(1).expand_as(rows), rows, "prod", include_self=True)
        ctx.save_for_backward(rows, index, out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        rows, index, out = ctx.saved_tensors
        g = (grad_out * out).index_select(0, index) / rows
        return g.masked_fill(g.isnan(), 0.0), None, None


def _winner_positions(rows: torch.Tensor, index: torch.Tensor, best: torch.Tensor, n: int) -> torch.Tensor:
    """torch_scatter's arg_out: the FIRST element (in element order) that attains the segment's extremum, per column;
    `E` for a segment without elements."""
    E, D = rows.shape
    pos = torch.arange(E, dtype=torch.int64).unsqueeze(1).expand(E, D)
    b = best.index_select(0, index)
    attained = rows == b
    if rows.is_floating_point():     # a NaN in a segment IS its extremum (torch_scatter and the reference propagate it)
        attained = attained | (rows.isnan() & b.isnan())
    cand = torch.where(attained, pos, torch.full_like(pos, E))
    arg = torch.full((n, D), E, dtype=torch.int64)
    return arg.scatter_reduce_(0, index.unsqueeze(1).expand(E, D), cand, reduce="amin", include_self=True)


def segment(rows: torch.Tensor, index: torch.Tensor, n: int, reduce: str, return_arg: bool = False):
    """out[s] = reduce over {rows[e] : index[e] == s}, [n, D]; sums fold in element order (the order the reference's
    CPU scatter_add_ folds them in).  Differentiable with torch_scatter's rules: max / min route the gradient to the
    single recorded winner (not spread over ties like torch's own amax)."""
    _host_only(rows, index)
    E, D = rows.shape
    if reduce in ("sum", "add"):
        out = torch.zeros(n, D, dtype=rows.dtype).index_add_(0, index, rows)
    elif reduce == "mean":
        total = torch.zeros(n, D, dtype=rows.dtype).index_add_(0, index, rows)
        count = torch.bincount(index, minlength=n)[:n].clamp_(min=1)
        out = total / count.to(rows.dtype).unsqueeze(1) if rows.is_floating_point() else \
            total.div(count.unsqueeze(1), rounding_mode="floor")
    elif reduce in ("max", "min"):
        with torch.no_grad():
            best = torch.zeros(n, D, dtype=rows.dtype).scatter_reduce_(
                0, index.unsqueeze(1).expand(E, D), rows.detach(), "amax" if reduce == "max" else "amin",
                include_self=False)
            arg = _winner_positions(rows.detach(), index, best, n)
        if E == 0:
            out = torch.zeros(n, D, dtype=rows.dtype) + rows.sum() * 0
        else:
            picked = rows.gather(0, arg.clamp(max=E - 1))            # gradient reaches the winner only
            out = torch.where(arg < E, picked, torch.zeros((), dtype=rows.dtype))
        if return_arg:
            return out, arg
    elif reduce == "mul":
        out = _SegmentProduct.apply(rows, index, n)
    else:
        raise ValueError(f"unknown aggregation function {reduce!r}")
    if return_arg:
        raise ValueError("arg is defined for max / min only")
    return out


# ------------------------------------------------------------------------------------------------------------------
# torch_scatter-shaped entry points (any `dim`, 1-D or broadcastable index) for host tensors
# ------------------------------------------------------------------------------------------------------------------
def _flatten(src: torch.Tensor, index: torch.Tensor, dim: int):
    """-> (rows [E', D'], index' [E'], restore(out_rows, n) -> tensor of src's layout with size n along dim,
    columns per segment id).  A 1-D index along `dim` keeps one segment id per slice; a wider index (torch_scatter
    broadcasts it over the trailing dimensions) gets one segment id per (index value, column)."""
    dim = dim % src.dim() if src.dim() else 0
    moved = src.movedim(dim, 0)
    tail = tuple(moved.shape[1:])
    width = 1
    for d in tail:
        width *= int(d)
    if index.dim() == 1:
        rows = moved.reshape(moved.shape[0], width)

        def restore(out_rows, n):
            return out_rows.reshape((n,) + tail).movedim(0, dim)
        return rows, index, 1, restore
    idx = index
    for _ in range(idx.dim(), src.dim()):
        idx = idx.unsqueeze(-1)
    idx = idx.expand(src.shape).movedim(dim, 0).reshape(moved.shape[0], width)    # [E, C]
    C = idx.shape[1]
    seg = (idx * C + torch.arange(C, dtype=torch.int64).unsqueeze(0)).reshape(-1)
    rows = moved.reshape(-1, 1)

    def restore(out_rows, n):
        return out_rows.reshape((n,) + tail).movedim(0, dim)
    return rows, seg, C, restore


def _size(index: torch.Tensor, dim_size) -> int:
    if dim_size is not None:
        return int(dim_size)
    return int(index.max()) + 1 if index.numel() else 0


def scatter(src, index, dim: int = -1, out=None, dim_size=None, reduce: str = "sum", return_arg: bool = False):
    _host_only(src, index)
    if out is not None:
        raise _lib.PtgnnAmdError("ptgnn_amd.scatter: the `out=` form is not supported")
    n = _size(index, dim_size)
    rows, seg, C, restore = _flatten(src, index, dim)
    res = segment(rows, seg, n * C, reduce, return_arg=return_arg)
    if not return_arg:
        return restore(res, n)
    vals, arg = res
    if C > 1:   # positions were counted over the flattened (element, column) pairs: back to the element index
        E = src.shape[dim % src.dim()]
        arg = torch.where(arg < rows.shape[0], arg // C, torch.full_like(arg, E))
    return restore(vals, n), restore(arg, n)


def _expand_index(index, src, dim):
    dim = dim % src.dim()
    idx = index
    if idx.dim() == 1:
        shape = [1] * src.dim()
        shape[dim] = idx.shape[0]
        idx = idx.view(shape)
    else:
        for _ in range(idx.dim(), src.dim()):
            idx = idx.unsqueeze(-1)
    return idx.expand(src.shape), dim


def scatter_log_softmax(src, index, dim: int = -1, eps: float = 1e-12, dim_size=None):
    idx, dim = _expand_index(index, src, dim)
    with torch.no_grad():
        shift = scatter(src.detach(), index, dim, None, dim_size, "max").gather(dim, idx)
    rec = src - shift
    total = scatter(rec.exp(), index, dim, None, dim_size, "sum")
    return rec - (total + eps).log().gather(dim, idx)


def scatter_softmax(src, index, dim: int = -1, eps: float = 1e-12, dim_size=None):
    idx, dim = _expand_index(index, src, dim)
    with torch.no_grad():
        shift = scatter(src.detach(), index, dim, None, dim_size, "max").gather(dim, idx)
    e = (src - shift).exp()
    total = scatter(e, index, dim, None, dim_size, "sum")
    return e / (total.gather(dim, idx) + eps)


def scatter_logsumexp(src, index, dim: int = -1, out=None, dim_size=None, eps: float = 1e-12):
    """torch_scatter.composite.scatter_logsumexp (2.0.x): the per-segment maximum is taken over a -inf initialised
    buffer, so a segment without elements answers log(eps) + (-inf) = -inf."""
    if out is not None:
        raise _lib.PtgnnAmdError("ptgnn_amd.scatter: the `out=` form is not supported")
    idx, dim = _expand_index(index, src, dim)
    n = _size(index, dim_size)
    with torch.no_grad():
        vals, arg = scatter(src.detach(), index, dim, None, n, "max", return_arg=True)
        E = src.shape[dim]
        top = torch.where(arg < E, vals, torch.full_like(vals, float("-inf")))
        shift = top.gather(dim, idx)
    rec = src - shift
    rec = rec.masked_fill(rec.isnan(), float("-inf"))
    total = scatter(rec.exp(), index, dim, None, n, "sum")
    return (total + eps).log() + top


def scatter_std(src, index, dim: int = -1, out=None, dim_size=None, unbiased: bool = True):
    """torch_scatter.scatter_std (2.0.x): sqrt(sum (x - mean)^2 / (max(count - 1, 1) + 1e-6))."""
    if out is not None:
        raise _lib.PtgnnAmdError("ptgnn_amd.scatter: the `out=` form is not supported")
    idx, dim = _expand_index(index, src, dim)
    n = _size(index, dim_size)
    count = scatter(torch.ones_like(src), index, dim, None, n, "sum").clamp(min=1)
    mean = scatter(src, index, dim, None, n, "sum") / count
    dev = src - mean.gather(dim, idx)
    ssq = scatter(dev * dev, index, dim, None, n, "sum")
    if unbiased:
        count = (count - 1).clamp(min=1)
    return (ssq / (count + 1e-6)).sqrt()


# ------------------------------------------------------------------------------------------------------------------
# layer bodies
# ------------------------------------------------------------------------------------------------------------------
def aggregate(messages: torch.Tensor, targets: torch.Tensor, num_nodes: int, reduce: str) -> torch.Tensor:
    """abstractmessagepassing.py:38-50: fp32 scatter whatever the message dtype (AMP), result back in that dtype."""
    return segment(messages.to(torch.float32), targets, int(num_nodes), reduce).to(messages.dtype)


def _message_inputs(node_states, adjacency_lists, edge_features, with_target: bool):
    for (src, dst), feats in zip(adjacency_lists, edge_features):
        parts = [node_states.index_select(0, src)]
        if with_target:
            parts.append(node_states.index_select(0, dst))
        if feats is not None:
            parts.append(feats)
        yield parts[0] if len(parts) == 1 else torch.cat(parts, dim=-1)


def ggnn_layer(node_states, adjacency_lists, edge_features, edge_linears: Sequence[nn.Linear], dropout: nn.Module,
               gru: nn.GRUCell, reduce: str) -> torch.Tensor:
    """gatedmessagepassing.py:46-69 on host tensors: per type W_t . Dropout([x_src ; f_e]), type-major messages,
    segment reduce onto the targets, GRUCell(aggregate, state)."""
    _host_only(node_states)
    msgs = [lin(dropout(inp)) for inp, lin in
            zip(_message_inputs(node_states, adjacency_lists, edge_features, False), edge_linears)]
    targets = torch.cat([d for _, d in adjacency_lists])
    agg = aggregate(torch.cat(msgs, dim=0), targets, node_states.shape[0], reduce)
    return gru(agg, node_states)


def mlp_layer(node_states, adjacency_lists, edge_features, edge_mlps: Sequence[nn.Module], with_target: bool,
              aggregation, activation: Optional[nn.Module], state_update: nn.Module) -> torch.Tensor:
    """mlpmessagepassing.py:80-117 on host tensors."""
    _host_only(node_states)
    msgs = [mlp(inp) for inp, mlp in
            zip(_message_inputs(node_states, adjacency_lists, edge_features, with_target), edge_mlps)]
    messages = torch.cat(msgs, dim=0)
    targets = torch.cat([d for _, d in adjacency_lists], dim=0)
    if isinstance(aggregation, str):
        agg = aggregate(messages, targets, node_states.shape[0], aggregation)
    else:
        agg = aggregation(messages=messages, message_targets=targets, num_nodes=node_states.shape[0])
    if activation is not None:
        agg = activation(agg)
    return state_update(agg)
