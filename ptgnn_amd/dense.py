"""Differentiable dense blocks of the hot path on the HIP kernels (training form).

The reference's layers end in `nn.GRUCell` (gatedmessagepassing.py:25,69) or `nn.Linear`
(mlpmessagepassing.py:60-63, residuallayers.py:112-116).  In training the forward GEMMs and the input
gradients are ordinary [rows, k] x [k, n] products (fp32 MFMA, `ptgnn_amd_linear_f32`); the WEIGHT
gradients reduce over all `rows` nodes into a tiny [n, k] matrix -- a shape the vendor BLAS runs at
~23 TFLOP/s on MI355X -- so they use the split-row MFMA kernel of edge_wgrad.hip
(`ptgnn_amd_linear_weight_grad_f32`, deterministic two-stage reduction).  The GRU cell's forward is the
same fused kernel inference uses (it additionally emits the gates), its gate-math backward one HBM-bound
HIP kernel.
"""
from typing import Optional

import torch

from ptgnn_amd import _lib, ops


def _kernel_dims_ok(x: torch.Tensor, weight: torch.Tensor) -> bool:
    """fp32 CUDA matrices: everything else is the caller's error (no CPU path) or an AMP dtype the layers
    up-cast before they get here."""
    return x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 2


def _pad4(t: torch.Tensor) -> torch.Tensor:
    """Zero-pad the column count to a multiple of 4 (the split-row weight-gradient kernel reads float4 rows)."""
    extra = (-t.shape[1]) % 4
    return t if extra == 0 and t.stride(0) % 4 == 0 else torch.nn.functional.pad(t, (0, extra)).contiguous()


_WT_CACHE = {}   # id(weight) -> (weakref, version, storage pointer, device, transposed copy)


def _transposed(weight: torch.Tensor) -> torch.Tensor:
    """`weight.t().contiguous()` for the input-gradient GEMMs, kept per (tensor, version, storage, device): a tied layer
    (the Typilus stack applies one GGNN layer seven times) transposes its GRU / Linear weights once per backward pass
    instead of once per use.  An optimizer step bumps the version; `param.data = ...` (module.to() / .cuda() / .float()
    after a backward, EMA weight swaps) does not, but moves the storage or the device, which the key also holds.  An
    in-place edit through `p.data` changes neither, so the cache is additionally dropped at the start of every
    forward (`clear_transposed_cache`, called by `layers.forward_scope`): a copy lives for one forward / backward."""
    import weakref
    key = id(weight)
    hit = _WT_CACHE.get(key)
    if (hit is not None and hit[0]() is weight and hit[1] == weight._version and hit[2] == weight.data_ptr()
            and hit[3] == weight.device):
        return hit[4]
    wt = weight.detach().t().contiguous()
    if len(_WT_CACHE) > 64:
        _WT_CACHE.clear()
    _WT_CACHE[key] = (weakref.ref(weight), weight._version, weight.data_ptr(), weight.device, wt)
    return wt


def clear_transposed_cache() -> None:
    _WT_CACHE.clear()


class _Linear(torch.autograd.Function):
    """y = x W^T (+ b);  d x = g W,  d W = g^T x (split-row kernel),  d b = column sums of g."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return ops.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = g.contiguous()
        d_x = d_w = d_b = None
        if ctx.needs_input_grad[0]:
            d_x = ops.linear(g, _transposed(weight))
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            n_out, k = weight.shape
            res = ops.linear_weight_grad(_pad4(x), _pad4(g), want_bias=want_b)   # bias gradient rides the same pass
            d_w, d_b = res if want_b else (res, None)
            if d_w.shape != weight.shape:     # odd widths were zero-padded to the kernel's float4 rows
                d_w = d_w[:n_out, :k].contiguous()
                d_b = d_b[:n_out].contiguous() if d_b is not None else None
        elif want_b:
            d_b = g.sum(dim=0)
        return d_x, d_w, d_b


class _LinearActDropout(torch.autograd.Function):
    """Dropout(act(x W^T + b)) -- the node update of mlpmessagepassing.py:60-66 -- as ONE autograd node: forward = the
    GEMM with the activation in its epilogue (what inference runs) + torch's dropout kernel (its mask is kept: same
    generator stream as nn.Dropout); backward = one elementwise HIP pass (mask, scale, act') in front of `_Linear`'s
    GEMMs.  Torch ran five kernels around the GEMMs here (tanh, dropout, their two backwards, ...), each a full pass
    over [N, H] -- a fifth of the README architecture's step was such passes."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, p, training):
        y = ops.linear(x, weight, bias, act=act)
        keep = None
        out = y
        if training and p > 0.0:
            out, keep = torch.ops.aten.native_dropout(y, p, True)
        ctx.save_for_backward(x, weight, y, keep)
        ctx.has_bias, ctx.act, ctx.scale = bias is not None, act, (1.0 / (1.0 - p) if keep is not None else 1.0)
        return out

    @staticmethod
    def backward(ctx, g):
        x, weight, y, keep = ctx.saved_tensors
        g = g.contiguous()
        if keep is not None or ctx.act is not None:
            g = ops.act_dropout_backward(g, y, keep, ctx.scale, ctx.act)
        d_x = d_w = d_b = None
        if ctx.needs_input_grad[0]:
            d_x = ops.linear(g, _transposed(weight))
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            res = ops.linear_weight_grad(x, g, want_bias=want_b)
            d_w, d_b = res if want_b else (res, None)
        elif want_b:
            d_b = g.sum(dim=0)
        return d_x, d_w, d_b, None, None, None


def linear_act_dropout(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], act: Optional[str],
                       p: float, training: bool) -> Optional[torch.Tensor]:
    """Dropout(act(Linear(x))) as one autograd node, or None when the shape is not the fused node's (widths that are not
    multiples of 4, non-fp32): the caller then composes `linear` with torch's activation / dropout modules."""
    n_out, k = weight.shape
    if not _kernel_dims_ok(x, weight) or k % 4 != 0 or n_out % 4 != 0 or x.stride(0) % 4 != 0 or p >= 1.0:
        return None
    return _LinearActDropout.apply(x, weight, bias, act, float(p), bool(training))


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Differentiable nn.Linear on the HIP kernels, any widths (odd ones take the kernels' unaligned staging
    path; the weight-gradient kernel sees them zero-padded to float4 rows).  There is no vendor-BLAS or CPU
    route: anything that is not a 2-D fp32 CUDA matrix raises."""
    if not _kernel_dims_ok(x, weight):
        raise _lib.PtgnnAmdError(f"dense.linear needs 2-D float32 CUDA matrices (got x {tuple(x.shape)} {x.dtype} on "
                                 f"{x.device}, weight {weight.dtype}); AMP dtypes are up-cast by the layers")
    return _Linear.apply(x, weight, bias)


class _GruCell(torch.autograd.Function):
    """nn.GRUCell as one autograd node: forward = the fused HIP cell (gate GEMMs + gate math, also emitting
    r, z, n, gh_n); backward = gate-math backward kernel, then the four GEMM halves (input gradients on the
    MFMA GEMM, weight + bias gradients on the split-row kernel)."""

    @staticmethod
    def forward(ctx, a, h, w_ih, w_hh, b_ih, b_hh):
        out, gates = ops.gru_cell_train(a, h, w_ih, w_hh, b_ih, b_hh)
        ctx.save_for_backward(a, h, w_ih, w_hh, gates)
        return out

    @staticmethod
    def backward(ctx, g):
        a, h, w_ih, w_hh, gates = ctx.saved_tensors
        d_gi, d_gh, d_h = ops.gru_gates_backward(g.contiguous(), gates, h)
        need = ctx.needs_input_grad
        d_a = ops.linear(d_gi, _transposed(w_ih)) if need[0] else None
        d_hx = ops.linear_add(d_gh, _transposed(w_hh), d_h) if need[1] else None      # d_h + d_gh W_hh, one launch
        d_wih = d_bih = d_whh = d_bhh = None
        if need[2] or need[4]:
            d_wih, d_bih = ops.linear_weight_grad(a, d_gi, want_bias=True)
        if need[3] or need[5]:
            d_whh, d_bhh = ops.linear_weight_grad(h, d_gh, want_bias=True)
        return d_a, d_hx, d_wih, d_whh, d_bih, d_bhh


def gru_cell(cell: torch.nn.GRUCell, a: torch.Tensor, h: torch.Tensor) -> torch.Tensor:
    """nn.GRUCell on the HIP kernels: the fused inference cell when nothing needs a gradient, else `_GruCell`."""
    if not (_kernel_dims_ok(a, cell.weight_ih) and _kernel_dims_ok(h, cell.weight_hh)):
        raise _lib.PtgnnAmdError(f"dense.gru_cell needs 2-D float32 CUDA matrices (got {a.dtype} / {h.dtype} on "
                                 f"{a.device}); AMP dtypes are up-cast by the layers")
    params = [cell.weight_ih, cell.weight_hh] + ([cell.bias_ih, cell.bias_hh] if cell.bias else [])
    needs_grad = torch.is_grad_enabled() and any(t.requires_grad for t in [a, h] + params)
    if cell.bias and not needs_grad:
        return ops.gru_cell(a, h, cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh)
    if cell.bias and h.shape[1] % 4 == 0 and a.shape[1] % 4 == 0:
        return _GruCell.apply(a, h, cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh)
    # Shapes the fused cell does not tile (nn.GRUCell(bias=False); training with a state / message width that is
    # not a multiple of 4): the reference accepts them (gatedmessagepassing.py:25), so they run as the two gate
    # GEMMs on the differentiable HIP Linear (any width) + torch's elementwise gate math -- same arithmetic, on
    # the GPU, no vendor BLAS.
    hd = h.shape[1]
    gi = linear(a, cell.weight_ih, cell.bias_ih if cell.bias else None)
    gh = linear(h, cell.weight_hh, cell.bias_hh if cell.bias else None)
    r = torch.sigmoid(gi[:, :hd] + gh[:, :hd])
    z = torch.sigmoid(gi[:, hd:2 * hd] + gh[:, hd:2 * hd])
    n = torch.tanh(gi[:, 2 * hd:] + r * gh[:, 2 * hd:])
    return (1.0 - z) * n + z * h


class _RowEpilogue(torch.autograd.Function):
    """y = LayerNorm(GELU(x)) row-wise on the HIP kernels (csrc/row_epilogue.hip); saves only x."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, flags):
        x = x.contiguous()
        ctx.save_for_backward(x, gamma)
        ctx.eps, ctx.flags = eps, flags
        return ops.row_epilogue(x, flags, gamma, beta, eps)

    @staticmethod
    def backward(ctx, gy):
        x, gamma = ctx.saved_tensors
        gx, gg, gb = ops.row_epilogue_backward(x, gy.contiguous(), ctx.flags, gamma, ctx.eps)
        return gx, gg, gb, None, None


def row_epilogue(x: torch.Tensor, gelu: bool, ln: Optional[torch.nn.LayerNorm]) -> torch.Tensor:
    """Differentiable GELU -> LayerNorm of mlpmessagepassing.py:114-116 (either may be absent, not both).  Needs
    fp32 CUDA rows of width <= 512 and an affine LayerNorm with bias -- the caller checks, this raises."""
    flags = (ops.EPI_GELU if gelu else 0) | (ops.EPI_LAYERNORM if ln is not None else 0)
    if flags == 0:
        return x
    if not x.is_cuda or x.dtype != torch.float32 or x.dim() != 2 or x.shape[1] > 512:
        raise _lib.PtgnnAmdError("row_epilogue: expected a float32 CUDA matrix with rows of at most 512 columns")
    if ln is not None:
        return _RowEpilogue.apply(x, ln.weight, ln.bias, float(ln.eps), flags)
    return _RowEpilogue.apply(x, None, None, 1e-5, flags)
