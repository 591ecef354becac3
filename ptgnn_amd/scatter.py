"""`torch_scatter`-compatible facade on the HIP segment-reduce kernel, plus the autograd seam.

Mirrors the call the reference makes at its single hot-path scatter site
(ptgnn/neuralmodels/gnn/messagepassing/abstractmessagepassing.py:44-50):
``scatter(src, index=targets, dim=0, dim_size=num_nodes, reduce=fn)`` with `src` [E, D] and a 1-D
int64 `index`.  Semantics follow torch_scatter 2.0.x: sum/mean/max/min, empty segments -> 0,
mean divides by max(count, 1); mul: product in edge order, empty segments -> 1; backward: sum -> gather,
mean -> gather / count, max/min -> the gradient flows only to the arg-max/min source element,
mul -> (grad * out)[index] / src.
"""
from typing import Optional

import torch

from ptgnn_amd import _lib, ops, torch_route


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
        if reduce in ("sum", "add", "mean"):
            g = grad_out
            if reduce == "mean":
                deg = (plan.rowptr[1:] - plan.rowptr[:-1])
                g = grad_out / deg.clamp(min=1).to(grad_out.dtype).unsqueeze(1)
            return ops.segment_spread(g, None, plan), None, None
        (arg,) = ctx.saved_tensors                                  # [N, D] winning slot or -1
        return ops.segment_spread(grad_out, arg, plan), None, None


class _SegmentMul(torch.autograd.Function):
    """reduce="mul" (torch_scatter.scatter_mul): out = product per destination in edge order, empty rows 1;
    backward as torch_scatter's ScatterMul: grad_src = (grad_out * out)[index] / src, NaN (0 / 0) -> 0."""

    @staticmethod
    def forward(ctx, messages, plan):
        out = ops.segment_mul(messages, plan)
        ctx.plan = plan
        ctx.save_for_backward(messages, out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        messages, out = ctx.saved_tensors
        plan = ctx.plan
        E = plan.num_edges
        if E == 0:
            return torch.zeros_like(messages), None
        deg = (plan.rowptr[1:] - plan.rowptr[:-1]).to(torch.int64)
        row_of_slot = torch.repeat_interleave(torch.arange(plan.num_nodes, device=out.device), deg, output_size=E)
        targets = torch.empty(E, dtype=torch.int64, device=out.device)
        targets[plan.perm[:E].to(torch.int64)] = row_of_slot                  # destination of every message row
        g = ops.gather_rows((grad_out * out).contiguous(), targets) / messages
        return g.masked_fill(g.isnan(), 0.0), None


def segment_reduce(messages: torch.Tensor, plan: "ops.GraphPlan", reduce: str) -> torch.Tensor:
    """Differentiable `_aggregate_messages` over a prebuilt plan (torch_scatter's reduce set: sum / add / mean / max /
    min on the fused HIP segment reduce, mul on its own kernel)."""
    if reduce == "mul":
        return _SegmentMul.apply(messages, plan)
    if reduce not in ops.REDUCE_IDS:
        raise ValueError(f"unknown aggregation function {reduce!r}")
    return _SegmentReduce.apply(messages, plan, reduce)


def _prepare(src: torch.Tensor, index: torch.Tensor, dim: int, out, dim_size):
    """Common argument handling of the torch_scatter-shaped entry points: 2-D (or 1-D) `src`, 1-D int64
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
    # torch_scatter semantics: a plan whose "source" column is unused; (index, index) gives dst = index
    plan = ops.build_plan([(index, index)], int(dim_size))
    return src, squeeze, plan


def _any_rank(fn):
    """GPU tensors of any rank with a 1-D index along `dim` (grucopydecoder.py:108 scatter_adds a [I, L, H] tensor along
    dim 0): the reduced dimension is moved to the front, the others flattened into the columns of the [E, D] form the HIP
    kernels take, and the result is given its layout back -- views and reshapes only.  (An index that is itself
    N-D / broadcast is served for CPU tensors only: torch_route._flatten.)"""
    import functools

    @functools.wraps(fn)
    def wrapped(src, index, dim=-1, *args, **kwargs):
        if src.is_cuda and index.dim() == 1 and src.dim() >= 2:
            d = dim % src.dim()
            if src.dim() > 2 or d != 0:
                moved = src.movedim(d, 0)
                rest = tuple(moved.shape[1:])
                res = fn(moved.reshape(moved.shape[0], -1), index, 0, *args, **kwargs)

                def back(t):
                    return t.reshape(t.shape[0], *rest).movedim(0, d)
                return tuple(back(t) for t in res) if isinstance(res, tuple) else back(res)
        return fn(src, index, dim, *args, **kwargs)
    return wrapped


@_any_rank
def scatter(src: torch.Tensor, index: torch.Tensor, dim: int = -1, out: Optional[torch.Tensor] = None,
            dim_size: Optional[int] = None, reduce: str = "sum") -> torch.Tensor:
    """torch_scatter.scatter for the layout the ptgnn hot path uses (abstractmessagepassing.py:44-50,
    pna_aggregation.py:28-45, varsizedsummary.py:35)."""
    if not src.is_cuda:   # device dispatch: host tensors take plain torch operators (ptgnn_amd/torch_route.py)
        return torch_route.scatter(src, index, dim, out, dim_size, reduce)
    src2, squeeze, plan = _prepare(src, index, dim, out, dim_size)
    res = segment_reduce(src2.to(torch.float32), plan, reduce).to(src.dtype)
    return res.squeeze(1) if squeeze else res


def scatter_sum(src, index, dim: int = -1, out=None, dim_size: Optional[int] = None) -> torch.Tensor:
    """torch_scatter.scatter_sum (varsizedsummary.py:59,76,108,174; selfattmessagepassing.py:61)."""
    return scatter(src, index, dim, out, dim_size, "sum")


scatter_add = scatter_sum


def scatter_mul(src, index, dim: int = -1, out=None, dim_size: Optional[int] = None) -> torch.Tensor:
    """torch_scatter.scatter_mul."""
    return scatter(src, index, dim, out, dim_size, "mul")


def scatter_mean(src, index, dim: int = -1, out=None, dim_size: Optional[int] = None) -> torch.Tensor:
    """torch_scatter.scatter_mean (graphnorm.py:36-41)."""
    return scatter(src, index, dim, out, dim_size, "mean")


@_any_rank
def _scatter_minmax(src, index, dim, out, dim_size, reduce):
    if not src.is_cuda:
        return torch_route.scatter(src, index, dim, out, dim_size, reduce, return_arg=True)
    src2, squeeze, plan = _prepare(src, index, dim, out, dim_size)
    x = src2.to(torch.float32)
    res = segment_reduce(x, plan, reduce).to(src.dtype)
    E = x.shape[0]
    with torch.no_grad():   # torch_scatter's arg_out: position of the winner in `src`, E for empty segments
        if E == 0:
            arg = torch.zeros(plan.num_nodes, x.shape[1], dtype=torch.int64, device=x.device)
        else:
            _, slot = ops.gather_reduce(x.detach(), plan, x.shape[1], reduce, return_arg=True, type_bits=0,
                                        col=plan.perm)
            pos = plan.perm[:E].to(torch.int64)[slot.clamp(min=0).to(torch.int64)]
            arg = torch.where(slot >= 0, pos, torch.full_like(pos, E))
    if squeeze:
        res, arg = res.squeeze(1), arg.squeeze(1)
    return res, arg


def scatter_max(src, index, dim: int = -1, out=None, dim_size: Optional[int] = None):
    """torch_scatter.scatter_max -> (values, argmax) (varmisuse.py:86)."""
    return _scatter_minmax(src, index, dim, out, dim_size, "max")


def scatter_min(src, index, dim: int = -1, out=None, dim_size: Optional[int] = None):
    return _scatter_minmax(src, index, dim, out, dim_size, "min")


class _GatherRows(torch.autograd.Function):
    """x[index] with the HIP row gather; backward = the HIP segment-sum over the index's plan."""

    @staticmethod
    def forward(ctx, x, index, plan):
        ctx.plan = plan
        return ops.gather_rows(x, index)

    @staticmethod
    def backward(ctx, g):
        return ops.segment_reduce(g.contiguous(), ctx.plan, "sum"), None, None


def gather_rows(x: torch.Tensor, index: torch.Tensor, plan: "ops.GraphPlan") -> torch.Tensor:
    """Differentiable x[index] on the HIP kernels; `plan` must be the segment plan of `index`
    (rows = x.shape[0]), e.g. `ops.plan_from_sorted_index(node_to_graph_idx, num_graphs)`."""
    return _GatherRows.apply(x, index, plan)


@_any_rank
def scatter_log_softmax(src, index, dim: int = -1, eps: float = 1e-12,
                        dim_size: Optional[int] = None) -> torch.Tensor:
    """torch_scatter.composite.scatter_log_softmax (varsizedsummary.py:57,106,158; varmisuse.py:79;
    grucopydecoder.py:100):  src - max_seg - log(sum_seg exp(src - max_seg) + eps)."""
    if not src.is_cuda:
        return torch_route.scatter_log_softmax(src, index, dim, eps, dim_size)
    src2, squeeze, plan = _prepare(src, index, dim, None, dim_size)
    x = src2.to(torch.float32)
    if x.shape[0] == 0:
        return src.clone()
    with torch.no_grad():   # the shift is a constant of the (shift-invariant) function
        shift = ops.gather_rows(ops.segment_reduce(x.detach(), plan, "max"), index)
    rec = x - shift
    total = segment_reduce(rec.exp(), plan, "sum")
    res = (rec - _GatherRows.apply((total + eps).log(), index, plan)).to(src.dtype)
    return res.squeeze(1) if squeeze else res


@_any_rank
def scatter_softmax(src, index, dim: int = -1, eps: float = 1e-12,
                    dim_size: Optional[int] = None) -> torch.Tensor:
    """torch_scatter.composite.scatter_softmax:  exp(src - max_seg) / (sum_seg exp(src - max_seg) + eps)."""
    if not src.is_cuda:
        return torch_route.scatter_softmax(src, index, dim, eps, dim_size)
    src2, squeeze, plan = _prepare(src, index, dim, None, dim_size)
    x = src2.to(torch.float32)
    if x.shape[0] == 0:
        return src.clone()
    with torch.no_grad():
        shift = ops.gather_rows(ops.segment_reduce(x.detach(), plan, "max"), index)
    e = (x - shift).exp()
    total = segment_reduce(e, plan, "sum")
    res = (e / (_GatherRows.apply(total, index, plan) + eps)).to(src.dtype)
    return res.squeeze(1) if squeeze else res


@_any_rank
def scatter_logsumexp(src, index, dim: int = -1, out=None, dim_size: Optional[int] = None,
                      eps: float = 1e-12) -> torch.Tensor:
    """torch_scatter.composite.scatter_logsumexp (grucopydecoder.py:122,190): log(sum_seg exp(src - max_seg) + eps)
    + max_seg, the maximum taken over a -inf initialised buffer (a segment without elements answers -inf)."""
    if not src.is_cuda:
        return torch_route.scatter_logsumexp(src, index, dim, out, dim_size, eps)
    src2, squeeze, plan = _prepare(src, index, dim, out, dim_size)
    x = src2.to(torch.float32)
    n, D = plan.num_nodes, x.shape[1]
    if x.shape[0] == 0:
        res = torch.full((n, D), float("-inf"), dtype=src.dtype, device=src.device)
        return res.squeeze(1) if squeeze else res
    with torch.no_grad():
        vals, slot = ops.gather_reduce(x.detach(), plan, D, "max", return_arg=True, type_bits=0, col=plan.perm)
        top = torch.where(slot >= 0, vals, torch.full_like(vals, float("-inf")))
        shift = ops.gather_rows(top, index)
    rec = x - shift
    rec = rec.masked_fill(rec.isnan(), float("-inf"))
    total = segment_reduce(rec.exp(), plan, "sum")
    res = ((total + eps).log() + top).to(src.dtype)
    return res.squeeze(1) if squeeze else res


@_any_rank
def scatter_std(src, index, dim: int = -1, out=None, dim_size: Optional[int] = None,
                unbiased: bool = True) -> torch.Tensor:
    """torch_scatter.scatter_std: sqrt(sum_seg (x - mean_seg)^2 / (max(count - 1, 1) + 1e-6)) (count, not count - 1,
    with unbiased=False)."""
    if not src.is_cuda:
        return torch_route.scatter_std(src, index, dim, out, dim_size, unbiased)
    src2, squeeze, plan = _prepare(src, index, dim, out, dim_size)
    x = src2.to(torch.float32)
    n, D = plan.num_nodes, x.shape[1]
    if x.shape[0] == 0:
        res = torch.zeros(n, D, dtype=src.dtype, device=src.device)
        return res.squeeze(1) if squeeze else res
    plan.wait()
    count = (plan.rowptr[1:] - plan.rowptr[:-1]).clamp(min=1).to(torch.float32).unsqueeze(1)
    mean = segment_reduce(x, plan, "sum") / count
    dev = x - _GatherRows.apply(mean.contiguous(), index, plan)
    ssq = segment_reduce(dev * dev, plan, "sum")
    if unbiased:
        count = (count - 1).clamp(min=1)
    res = (ssq / (count + 1e-6)).sqrt().to(src.dtype)
    return res.squeeze(1) if squeeze else res


# ------------------------------------------------------------------------------------------------------------------
# `import torch_scatter` for the untouched reference modules
# ------------------------------------------------------------------------------------------------------------------
FACADE_VERSION = "2.0.6+ptgnn_amd"


def install(force: bool = False):
    """Register this facade as the modules `torch_scatter` and `torch_scatter.composite` in `sys.modules`, so that the
    reference's untouched sources -- `from torch_scatter import scatter` (abstractmessagepassing.py:4),
    `scatter_log_softmax, scatter_max` (varmisuse.py:8), `scatter, scatter_log_softmax, scatter_sum`
    (varsizedsummary.py:7), `scatter_add` + `from torch_scatter.composite import scatter_log_softmax,
    scatter_logsumexp` (grucopydecoder.py:9-10), `scatter_mean` (graphnorm.py:3) -- import the HIP-backed functions
    (GPU tensors) / the plain-torch route (CPU tensors) with NO edit of reference code:

        import ptgnn_amd.scatter; ptgnn_amd.scatter.install()     # before the first `import ptgnn`

    A real `torch_scatter` wheel, when one is importable, is left alone unless `force=True`.  Returns the module."""
    import importlib.util
    import sys
    import types
    have = sys.modules.get("torch_scatter")
    if have is not None and getattr(have, "__version__", "") == FACADE_VERSION:
        return have
    if not force and (have is not None or importlib.util.find_spec("torch_scatter") is not None):
        return sys.modules.get("torch_scatter") or importlib.import_module("torch_scatter")
    top = types.ModuleType("torch_scatter")
    top.__doc__ = "ptgnn_amd.scatter registered as torch_scatter (HIP segment reduce on MI355X; torch on CPU tensors)"
    comp = types.ModuleType("torch_scatter.composite")
    for f in (scatter, scatter_sum, scatter_mul, scatter_mean, scatter_max, scatter_min, scatter_softmax,
              scatter_log_softmax, scatter_logsumexp, scatter_std):
        setattr(top, f.__name__, f)
    top.scatter_add = scatter_add
    for f in (scatter_softmax, scatter_log_softmax, scatter_logsumexp, scatter_std):
        setattr(comp, f.__name__, f)
    top.composite = comp
    top.__version__ = FACADE_VERSION
    top.__path__ = []            # a package: `import torch_scatter.composite` resolves through sys.modules
    # real specs: importlib.util.find_spec("torch_scatter") raises ValueError on a module whose __spec__ is None
    from importlib.machinery import ModuleSpec
    top.__spec__ = ModuleSpec("torch_scatter", None, is_package=True)
    comp.__spec__ = ModuleSpec("torch_scatter.composite", None)
    sys.modules["torch_scatter"] = top
    sys.modules["torch_scatter.composite"] = comp
    return top


class _EdgeLinear(torch.autograd.Function):
    """Per-edge message Linear of ALL edge types as one differentiable op (training form of the fused
    edge path):  msg[off_t + e] = W_t . Dropout([x[src_t[e]] ; x[dst_t[e]]]).

    forward : grouped per-edge GEMM (stream_gemm.hip / edge_gemm.hip); the dropout mask is a hash of (seed, row,
              column), evaluated once per call into ONE BIT per element (`ops.dropout_bitmask`) and kept for the
              backward (shapes outside the streaming kernels re-evaluate the hash inside the tile kernels);
    backward: d W_t  = d msg_t^T . in_t              grouped split-edge GEMM (edge_wgrad.hip)
              d in   = (d msg . W_t) * mask          the same grouped GEMM over an identity index
              d x    = segment-sum of d in over the source rows (+ over the destination rows for the
                       target-state half): the HIP segment reduce over the transposed / forward plan.
    No [E, K] gathered input is kept between forward and backward, and the mask only as bits (E * K / 8 bytes).
    """

    @staticmethod
    def forward(ctx, x, plan, use_dst, dropout_p, dropout_seed, w_stack):
        adj = plan._adj
        drop = (1, dropout_p, dropout_seed) if dropout_p > 0.0 else None
        # the keep mask as one bit per element (E * H / 8 bytes), generated once and shared by the forward, the input
        # gradient and the weight gradient: the hash inside three GEMMs is kernel time (fp32 MFMA shares the VALU lanes)
        bits = None
        if drop is not None and not use_dst and plan.num_edges > 0:
            bits = ops.dropout_bitmask(plan.num_edges, x.shape[1], dropout_p, dropout_seed, x.device)
        msg = ops.edge_linear(x, adj, list(w_stack.unbind(0)), use_dst, dropout=drop, mask_bits=bits)
        ctx.plan, ctx.use_dst, ctx.drop, ctx.bits = plan, use_dst, (dropout_p, dropout_seed), bits
        ctx.save_for_backward(x, w_stack)
        return msg

    @staticmethod
    def backward(ctx, grad_msg):
        x, w_stack = ctx.saved_tensors
        plan, use_dst = ctx.plan, ctx.use_dst
        p, seed = ctx.drop
        adj = plan._adj
        H = x.shape[1]
        gm = grad_msg.contiguous()
        need_x = ctx.needs_input_grad[0]
        d_x = d_w = None
        if ctx.needs_input_grad[5]:
            d_w = ops.edge_weight_grad(x, adj, gm, use_dst, p, seed, mask_bits=ctx.bits)   # [T, M, K], one tensor
        if need_x:
            if plan.num_edges == 0:
                d_x = torch.zeros_like(x)
            else:
                wt = w_stack.detach().transpose(1, 2).contiguous()                             # [T, K, M]
                drop = (2, p, seed) if p > 0.0 else None
                g_in = ops.edge_linear(gm, [(i, i) for i in plan.identity_index()], list(wt.unbind(0)),
                                       False, dropout=drop, mask_bits=ctx.bits)               # [E, K]
                tp = plan.transposed_plan()
                d_x = ops.gather_reduce(g_in[:, :H] if use_dst else g_in, tp, H, "sum", type_bits=0,
                                        col=tp.perm)
                if d_x.shape[0] != x.shape[0]:   # plans over a halo table: rows past the sources are zero
                    d_x = torch.nn.functional.pad(d_x, (0, 0, 0, x.shape[0] - d_x.shape[0]))
                if use_dst:   # destinations are plan rows: the first plan.num_nodes rows of the table
                    d_dst = ops.gather_reduce(g_in[:, H:], plan, H, "sum", type_bits=0, col=plan.perm)
                    if d_dst.shape[0] == d_x.shape[0]:
                        d_x = d_x + d_dst
                    else:
                        d_x[: d_dst.shape[0]] += d_dst
        return d_x, None, None, None, None, d_w


class _EdgeLinearFeat(torch.autograd.Function):
    """msg[off_t + e] = W_t . [x[src_t[e]] ; x[dst_t[e]] (if use_dst) ; feat_t[e]] with per-edge FEATURE rows
    (gatedmessagepassing.py:57-61, mlpmessagepassing.py:90-98 with the features of graphneuralnetwork.py:162-186), training form.

    forward : the grouped per-edge GEMM whose A rows carry the features as a third K range (ptgnn_amd_edge_linear_feat_f32):
              the reference's [E, H (+H) + F] message input is never built;
    backward: the weight's STATE columns and d x exactly as `_EdgeLinear` (split-edge weight-gradient GEMM; grouped GEMM over
              an identity index + segment sums), the weight's F feature columns and d feat per edge type on the dense HIP
              kernels (`linear_weight_grad` / `linear` over the type's [E_t, .] rows -- F is a handful of columns)."""

    @staticmethod
    def forward(ctx, x, plan, use_dst, n_types, *rest):
        ws, feats = list(rest[:n_types]), list(rest[n_types:])
        msg = ops.edge_linear(x, plan._adj, [w.detach() for w in ws], use_dst, edge_feats=[f.detach() for f in feats])
        ctx.plan, ctx.use_dst, ctx.n_types = plan, use_dst, n_types
        ctx.save_for_backward(x, *ws, *feats)
        return msg

    @staticmethod
    def backward(ctx, grad_msg):
        saved = ctx.saved_tensors
        T, plan, use_dst = ctx.n_types, ctx.plan, ctx.use_dst
        x, ws, feats = saved[0], saved[1:1 + T], saved[1 + T:]
        adj = plan._adj
        H = x.shape[1]
        Ks = H * (2 if use_dst else 1)
        gm = grad_msg.contiguous()
        need = ctx.needs_input_grad
        d_x = None
        d_ws = [None] * T
        d_fs = [None] * T
        counts = [int(a[0].shape[0]) for a in adj]
        if any(need[4:4 + T]):
            d_state = ops.edge_weight_grad(x, adj, gm, use_dst)                            # [T, M, Ks]
            off = 0
            for t, (f, n) in enumerate(zip(feats, counts)):
                if need[4 + t]:
                    if n > 0:
                        d_feat_cols = ops.linear_weight_grad(_pad_cols4(f), gm[off:off + n])[:, : f.shape[1]]
                    else:
                        d_feat_cols = gm.new_zeros(gm.shape[1], f.shape[1])
                    d_ws[t] = torch.cat([d_state[t], d_feat_cols], dim=1)
                off += n
        if need[0]:
            if plan.num_edges == 0:
                d_x = torch.zeros_like(x)
            else:
                wt = [w.detach()[:, :Ks].t().contiguous() for w in ws]                     # [Ks, M] per type
                g_in = ops.edge_linear(gm, [(i, i) for i in plan.identity_index()], wt, False)   # [E, Ks]
                tp = plan.transposed_plan()
                d_x = ops.gather_reduce(g_in[:, :H] if use_dst else g_in, tp, H, "sum", type_bits=0, col=tp.perm)
                if d_x.shape[0] != x.shape[0]:
                    d_x = torch.nn.functional.pad(d_x, (0, 0, 0, x.shape[0] - d_x.shape[0]))
                if use_dst:
                    d_dst = ops.gather_reduce(g_in[:, H:], plan, H, "sum", type_bits=0, col=plan.perm)
                    if d_dst.shape[0] == d_x.shape[0]:
                        d_x = d_x + d_dst
                    else:
                        d_x[: d_dst.shape[0]] += d_dst
        off = 0
        for t, (w, f, n) in enumerate(zip(ws, feats, counts)):
            if need[4 + T + t]:
                d_fs[t] = (ops.linear(gm[off:off + n], w.detach()[:, Ks:].t().contiguous()) if n > 0
                           else gm.new_zeros(0, f.shape[1]))
            off += n
        return (d_x, None, None, None, *d_ws, *d_fs)


def _pad_cols4(t: torch.Tensor) -> torch.Tensor:
    extra = (-t.shape[1]) % 4
    return t.contiguous() if extra == 0 else torch.nn.functional.pad(t, (0, extra)).contiguous()


def edge_linear_feat(x: torch.Tensor, plan: "ops.GraphPlan", weights, use_dst: bool, edge_feats) -> torch.Tensor:
    """Differentiable grouped per-edge Linear with per-edge feature rows (see `_EdgeLinearFeat`); `weights[t]` is the
    type's [M, H (+H) + F] nn.Linear weight, `edge_feats[t]` its [E_t, F] fp32 feature rows on x's device."""
    if plan._adj is None:
        raise _lib.PtgnnAmdError("edge_linear_feat: the plan must keep its adjacency lists")
    ws, fs = list(weights), list(edge_feats)
    return _EdgeLinearFeat.apply(x, plan, bool(use_dst), len(ws), *ws, *fs)


def edge_linear(x: torch.Tensor, plan: "ops.GraphPlan", weights, use_dst: bool, dropout_p: float = 0.0,
                dropout_seed: int = 0) -> torch.Tensor:
    """Differentiable grouped per-edge Linear over the plan's adjacency lists (see `_EdgeLinear`).
    `weights`: the per-type nn.Linear weights as a list, or already stacked [T, M, K] (a stack shared by
    the tied layers of a forward makes autograd accumulate ONE [T, M, K] gradient per use instead of T)."""
    if plan._adj is None:
        raise _lib.PtgnnAmdError("edge_linear: the plan must keep its adjacency lists")
    w_stack = weights if isinstance(weights, torch.Tensor) else torch.stack(list(weights))
    if int(float(dropout_p) * 65536.0 + 0.5) == 0:
        # the hash mask keeps an element when 16 hash bits >= round(p * 65536): below p ~ 7.6e-6 that threshold is 0,
        # every element is kept, and "dropout off" (no mask, no 1 / (1 - p) scale) is the one consistent reading for the
        # forward, the input gradient and the weight gradient alike
        dropout_p = 0.0
    return _EdgeLinear.apply(x, plan, bool(use_dst), float(dropout_p), int(dropout_seed), w_stack)


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
