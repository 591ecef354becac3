"""Differentiable dense blocks of the hot path on the HIP kernels (training form).

The reference's layers end in `nn.GRUCell` (gatedmessagepassing.py:25,69) or `nn.Linear`
(mlpmessagepassing.py:60-63, residuallayers.py:112-116).  In training the forward GEMMs and the input
gradients are ordinary [rows, k] x [k, n] products (fp32 MFMA, `ptgnn_amd_linear_f32`); the WEIGHT
gradients reduce over all `rows` nodes into a tiny [n, k] matrix -- a shape the vendor BLAS runs at
~23 TFLOP/s on MI355X -- so they use the split-row MFMA kernel of edge_wgrad.hip
(`ptgnn_amd_linear_weight_grad_f32`, deterministic two-stage reduction).
"""
from typing import Optional

import torch

from ptgnn_amd import ops


def _kernel_dims_ok(x: torch.Tensor, weight: torch.Tensor) -> bool:
    return (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 2
            and x.shape[1] % 4 == 0 and weight.shape[0] % 4 == 0)


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
            d_x = ops.linear(g, weight.detach().t().contiguous())
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            res = ops.linear_weight_grad(x, g, want_bias=want_b)      # bias gradient rides the same pass
            d_w, d_b = res if want_b else (res, None)
        elif want_b:
            d_b = g.sum(dim=0)
        return d_x, d_w, d_b


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Differentiable nn.Linear on the HIP kernels; shapes the kernels do not cover go to torch (still on
    the GPU -- there is no CPU path)."""
    if not _kernel_dims_ok(x, weight):
        return torch.nn.functional.linear(x, weight, bias)
    return _Linear.apply(x, weight, bias)


def gru_cell(cell: torch.nn.GRUCell, a: torch.Tensor, h: torch.Tensor) -> torch.Tensor:
    """Differentiable nn.GRUCell: the two gate GEMMs (forward, input and weight gradients) on the HIP
    kernels, the gate non-linearities through torch's fused pointwise GRU op and its autograd."""
    if not (_kernel_dims_ok(a, cell.weight_ih) and _kernel_dims_ok(h, cell.weight_hh) and cell.bias):
        return cell(a, h)
    # biases go into the gate GEMMs (fused epilogue forward, column sums of the weight-gradient pass
    # backward): the gate math  r, z = sigmoid(gi + gh),  n = tanh(gi_n + r * gh_n)  is unchanged because
    # b_hn sits inside the r * (.) product either way (torch.nn.GRUCell definition)
    gi = _Linear.apply(a, cell.weight_ih, cell.bias_ih)
    gh = _Linear.apply(h, cell.weight_hh, cell.bias_hh)
    return torch.ops.aten._thnn_fused_gru_cell(gi, gh, h, None, None)[0]
