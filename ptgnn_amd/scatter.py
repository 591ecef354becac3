"""`torch_scatter`-compatible facade on the HIP segment-reduce kernel, plus the autograd seam.

Mirrors the call the reference makes at its single hot-path scatter site
(ptgnn/neuralmodels/gnn/messagepassing/abstractmessagepassing.py:44-50):
``scatter(src, index=targets, dim=0, dim_size=num_nodes, reduce=fn)`` with `src` [E, D] and a 1-D
int64 `index`.  Semantics follow torch_scatter 2.0.x: sum/mean/max/min, empty segments -> 0,
mean divides by max(count, 1); backward: sum -> gather, mean -> gather / count, max/min -> the
gradient flows only to the arg-max/min source element.
"""
from typing import Optional

import torch

from ptgnn_amd import _lib, ops


class _SegmentReduce(torch.autograd.Function):
    """messages [E, D] (type-major edge order) --plan--> [N, D]."""

    @staticmethod
    def forward(ctx, messages, plan, reduce):
        need_arg = reduce in ("max", "min") and messages.requires_grad
        msg = messages if messages.shape[0] > 0 else messages.new_zeros(1, messages.shape[1])
        res = ops.gather_reduce(msg, plan, messages.shape[1], reduce, return_arg=need_arg,
                                type_bits=0, col=plan.perm)
        ctx.plan, ctx.reduce, ctx.num_edges = plan, reduce, messages.shape[0]
        if need_arg:
            out, arg = res
            ctx.save_for_backward(arg)
            ctx.mark_non_differentiable(arg)
            return out
        return res

    @staticmethod
    def backward(ctx, grad_out):
        plan, reduce, E = ctx.plan, ctx.reduce, ctx.num_edges
        plan.wait()
        grad_out = grad_out.contiguous()
        D = grad_out.shape[1]
        if E == 0:
            return grad_out.new_zeros(0, D), None, None
        # slot -> destination row (expand rowptr), slot -> original edge position (perm)
        deg = (plan.rowptr[1:] - plan.rowptr[:-1]).to(torch.int64)
        slot_dst = torch.repeat_interleave(torch.arange(plan.num_nodes, device=grad_out.device), deg,
                                           output_size=E)
        perm = plan.perm[:E].to(torch.int64)
        if reduce in ("sum", "add", "mean"):
            g = grad_out
            if reduce == "mean":
                g = grad_out / deg.clamp(min=1).to(grad_out.dtype).unsqueeze(1)
            grad_slots = ops.gather_rows(g, slot_dst)               # [E, D] in CSR slot order
            grad_msg = torch.empty_like(grad_slots)
            grad_msg[perm] = grad_slots
            return grad_msg, None, None
        (arg,) = ctx.saved_tensors                                  # [N, D] winning slot or -1
        grad_msg = grad_out.new_zeros(E * D)
        valid = arg >= 0
        slot = arg.clamp(min=0).to(torch.int64)
        flat = perm[slot] * D + torch.arange(D, device=grad_out.device).unsqueeze(0)
        grad_msg.index_put_((flat[valid],), grad_out[valid], accumulate=False)
        return grad_msg.view(E, D), None, None


def segment_reduce(messages: torch.Tensor, plan: "ops.GraphPlan", reduce: str) -> torch.Tensor:
    """Differentiable `_aggregate_messages` over a prebuilt plan."""
    if reduce not in ops.REDUCE_IDS:
        raise ValueError(f"unknown aggregation function {reduce!r}")
    return _SegmentReduce.apply(messages, plan, reduce)


def scatter(src: torch.Tensor, index: torch.Tensor, dim: int = -1, out: Optional[torch.Tensor] = None,
            dim_size: Optional[int] = None, reduce: str = "sum") -> torch.Tensor:
    """torch_scatter.scatter for the layout the ptgnn hot path uses: 2-D (or 1-D) `src`, 1-D int64
    `index` along dim 0.  Anything else raises (no silent fallback)."""
    if out is not None:
        raise _lib.PtgnnAmdError("ptgnn_amd.scatter: the `out=` form is not supported")
    squeeze = False
    if src.dim() == 1:
        src, squeeze = src.unsqueeze(1), True
    if dim < 0:
        dim += src.dim() - (1 if squeeze else 0)
    if dim != 0 or src.dim() != 2 or index.dim() != 1 or index.shape[0] != src.shape[0]:
        raise _lib.PtgnnAmdError("ptgnn_amd.scatter supports src [E, D] / [E] with a 1-D index along "
                                 f"dim 0 (got src {tuple(src.shape)}, index {tuple(index.shape)}, dim {dim})")
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() > 0 else 0
    dt = src.dtype
    # torch_scatter semantics: a plan whose "source" column is unused; (index, index) gives dst=index
    plan = ops.build_plan([(index, index)], int(dim_size))
    res = segment_reduce(src.to(torch.float32), plan, reduce).to(dt)
    return res.squeeze(1) if squeeze else res


class _GatherReduce(torch.autograd.Function):
    """Differentiable fused aggregation  out[v] = REDUCE_i ( ysrc[src_i, t_i] (+ ydst[v, t_i]) )  over a
    plan.  Forward and the heavy half of backward are the HIP gather/segment-reduce kernel:

      * sum / mean: d ysrc viewed as [N*T, M] is itself a segment-sum -- row (s, t) adds up the output
        gradients of its out-edges -- i.e. the SAME kernel over the backward plan (rows = src*T + type,
        col = dst), deterministic and atomic-free;
      * max / min : the forward records the winning CSR slot per (node, feature) (torch_scatter's
        arg_out); the gradient is routed to that slot's edge and then segment-summed per (src, type).
    """

    @staticmethod
    def forward(ctx, ysrc, ydst, plan, msg_dim, reduce):
        needs = ysrc.requires_grad or (ydst is not None and ydst.requires_grad)
        want_arg = reduce in ("max", "min") and needs
        res = ops.gather_reduce(ysrc, plan, msg_dim, reduce, ydst=ydst, return_arg=want_arg)
        ctx.plan, ctx.reduce, ctx.msg_dim = plan, reduce, msg_dim
        ctx.src_shape = tuple(ysrc.shape)
        ctx.has_dst = ydst is not None
        if want_arg:
            out, arg = res
            ctx.save_for_backward(arg)
            return out
        ctx.save_for_backward()
        return res

    @staticmethod
    def backward(ctx, grad_out):
        plan, reduce, M = ctx.plan, ctx.reduce, ctx.msg_dim
        plan.wait()
        T, N, E = plan.num_types, plan.num_nodes, plan.num_edges
        g = grad_out.contiguous()
        dev = g.device
        need_src, need_dst = ctx.needs_input_grad[0], ctx.has_dst and ctx.needs_input_grad[1]
        d_src = d_dst = None
        if E == 0:
            if need_src:
                d_src = g.new_zeros(ctx.src_shape)
            if need_dst:
                d_dst = g.new_zeros(N, T * M)
            return d_src, d_dst, None, None, None
        deg = (plan.rowptr[1:] - plan.rowptr[:-1])
        mask = (1 << plan.type_bits) - 1
        if reduce in ("sum", "add", "mean"):
            if reduce == "mean":
                g = g / deg.clamp(min=1).to(g.dtype).unsqueeze(1)
            if need_src:
                bp = plan.backward_plan()
                d_src = ops.gather_reduce(g, bp, M, "sum").view(plan.num_src_rows, T * M)
            if need_dst:
                if T == 1:
                    d_dst = g * deg.to(g.dtype).unsqueeze(1)
                else:  # cnt[v, t] = number of in-edges of type t
                    slot_t = (plan.col[:E] & mask).to(torch.int64)
                    slot_v = torch.repeat_interleave(torch.arange(N, device=dev), deg.to(torch.int64),
                                                     output_size=E)
                    cnt = torch.zeros(N * T, dtype=g.dtype, device=dev)
                    cnt.index_add_(0, slot_v * T + slot_t, torch.ones(E, dtype=g.dtype, device=dev))
                    d_dst = (cnt.view(N, T, 1) * g.view(N, 1, M)).reshape(N, T * M)
        else:
            (arg,) = ctx.saved_tensors                                  # [N, M] winning slot or -1
            valid = arg >= 0
            slot = arg.clamp(min=0).to(torch.int64)
            cols = torch.arange(M, device=dev).unsqueeze(0).expand(N, M)
            if need_src:   # masked segment-sum over the backward plan: no [E, M] gradient is materialised
                bp = plan.backward_plan()
                d_src = ops.gather_reduce_masked(g, arg, bp, plan.forward_slot_of_backward_slot(),
                                                 M).view(plan.num_src_rows, T * M)
            if need_dst:
                t_win = (plan.col[:E] & mask).to(torch.int64)[slot]
                d_dst = g.new_zeros(N, T * M)
                d_dst.scatter_(1, t_win * M + cols, torch.where(valid, g, torch.zeros_like(g)))
        if d_src is not None and tuple(d_src.shape) != ctx.src_shape:
            d_src = d_src.reshape(ctx.src_shape)
        return d_src, d_dst, None, None, None


def gather_reduce(ysrc: torch.Tensor, ydst: Optional[torch.Tensor], plan: "ops.GraphPlan", msg_dim: int,
                  reduce: str) -> torch.Tensor:
    """Differentiable fused aggregation over a message table (see `_GatherReduce`)."""
    if reduce not in ops.REDUCE_IDS:
        raise ValueError(f"unknown aggregation function {reduce!r}")
    return _GatherReduce.apply(ysrc, ydst, plan, msg_dim, reduce)
