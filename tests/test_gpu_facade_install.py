"""`ptgnn_amd.scatter.install()` on the GPU: the reference's import forms (`from torch_scatter import ...`,
`from torch_scatter.composite import ...`) reach the HIP segment reduce, with the library's launch path asserted, and
the facade additions of round 5 (`scatter_logsumexp`, `scatter_std`) against the host route and float64."""
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_reference_import_forms_reach_the_hip_kernels_through_torch_scatter():
    code = r'''
import sys
sys.path.insert(0, %r)
import torch
import ptgnn_amd.scatter as S
S.install()
from torch_scatter import scatter, scatter_add, scatter_log_softmax, scatter_max, scatter_mean, scatter_sum   # reference forms
from torch_scatter.composite import scatter_log_softmax as sls, scatter_logsumexp
from oracle import scatter_ref as O          # the checker
from ptgnn_amd import ops
g = torch.Generator().manual_seed(0)
E, D, n = 5000, 24, 700
src = torch.randn(E, D, generator=g)
idx = torch.randint(0, n - 3, (E,), generator=g)
timer = ops.KernelTimer(); ops.set_kernel_timer(timer)
for red in ("sum", "mean", "max", "min"):
    got = scatter(src.cuda(), index=idx.cuda(), dim=0, dim_size=n, reduce=red).cpu()
    want = O.scatter(src, idx, 0, None, n, red)
    tol = 0.0 if red in ("sum", "max", "min") else 1e-6
    assert float((got - want).abs().max()) <= tol, red
v, a = scatter_max(src[:, 0].cuda(), index=idx.cuda())
wv, wa = O.scatter_max(src[:, 0], idx)
assert torch.equal(v.cpu(), wv) and torch.equal(a.cpu(), wa)
assert torch.equal(scatter_add(src.cuda(), idx.cuda(), 0, None, n).cpu(), O.scatter_sum(src, idx, 0, None, n))
d = (sls(src[:, 0].cuda(), index=idx.cuda(), dim=0, eps=0).cpu() - O.scatter_log_softmax(src[:, 0], idx, 0, 0.0)).abs().max()
assert float(d) <= 1e-5
ops.set_kernel_timer(None)
names = set(timer.summary())
assert "gather_reduce" in names and "csr_build" in names, names      # the C ABI ran, not torch
print("FACADE_GPU_OK", sorted(names))
''' % ROOT
    proc = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0 and "FACADE_GPU_OK" in proc.stdout, proc.stdout[-2000:] + proc.stderr[-4000:]


@pytest.mark.parametrize("shape", [(3000,), (3000, 12)])
def test_logsumexp_and_std_on_the_gpu_match_the_host_route_and_float64(shape):
    from ptgnn_amd import scatter as S
    g = torch.Generator().manual_seed(5)
    src = torch.randn(*shape, generator=g) * 3
    n = 400
    idx = torch.randint(0, n - 5, (shape[0],), generator=g)
    lse = S.scatter_logsumexp(src.cuda(), idx.cuda(), dim=0, dim_size=n).cpu()
    std = S.scatter_std(src.cuda(), idx.cuda(), dim=0, dim_size=n).cpu()
    lse_h = S.scatter_logsumexp(src.double(), idx, dim=0, dim_size=n)
    std_h = S.scatter_std(src.double(), idx, dim=0, dim_size=n)
    empty = torch.bincount(idx, minlength=n) == 0
    assert bool(torch.isneginf(lse[empty]).all()) and bool(torch.isneginf(lse_h[empty]).all())
    np.testing.assert_allclose(lse[~empty].numpy(), lse_h[~empty].numpy(), rtol=0, atol=1e-5)
    np.testing.assert_allclose(std.numpy(), std_h.numpy(), rtol=0, atol=1e-5)
    # gradients through the HIP autograd nodes vs the host route in float64
    x = src.cuda().requires_grad_(True)
    w = torch.linspace(-1, 1, n).cuda()
    w = w if len(shape) == 1 else w.unsqueeze(1)
    keep = (~empty).cuda()
    (S.scatter_logsumexp(x, idx.cuda(), dim=0, dim_size=n)[keep] * w[keep]).sum().backward()
    xh = src.double().requires_grad_(True)
    (S.scatter_logsumexp(xh, idx, dim=0, dim_size=n)[~empty] * w.cpu().double()[~empty]).sum().backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), xh.grad.numpy(), rtol=0, atol=2e-5)
    x.grad = None
    (S.scatter_std(x, idx.cuda(), dim=0, dim_size=n) * w).sum().backward()
    xh.grad = None
    (S.scatter_std(xh, idx, dim=0, dim_size=n) * w.cpu().double()).sum().backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), xh.grad.numpy(), rtol=0, atol=5e-5)


def test_any_rank_src_on_the_gpu_equals_the_host_route():
    """grucopydecoder.py:100-122 calls scatter_log_softmax / scatter_add / scatter_logsumexp with dim=0 on [I, L] and
    [I, L, H] tensors: the GPU entry points flatten the trailing dimensions into the kernels' [E, D] form and restore the
    layout (only the CPU route did), any `dim`, values and arg positions equal to the host route's."""
    from ptgnn_amd import scatter as S
    g = torch.Generator().manual_seed(11)
    I, L, H, n = 700, 5, 8, 90
    src = torch.randn(I, L, H, generator=g)
    idx = torch.randint(0, n, (I,), generator=g)
    got = S.scatter_add(src.cuda(), index=idx.cuda(), dim=0)
    want = S.scatter_add(src, index=idx, dim=0)
    assert tuple(got.shape) == tuple(want.shape) == (int(idx.max()) + 1, L, H)
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=0, atol=1e-5)
    v, a = S.scatter_max(src.cuda(), idx.cuda(), dim=0, dim_size=n)
    vh, ah = S.scatter_max(src, idx, dim=0, dim_size=n)
    assert torch.equal(v.cpu(), vh) and torch.equal(a.cpu(), ah)
    ls = S.scatter_log_softmax(src[:, :, 0].cuda(), index=idx.cuda(), dim=0, eps=0)
    np.testing.assert_allclose(ls.cpu().numpy(), S.scatter_log_softmax(src[:, :, 0], index=idx, dim=0, eps=0).numpy(),
                               rtol=0, atol=1e-5)
    # the reduced dimension in the middle: index along dim 1 of [L, I, H]
    mid = src.permute(1, 0, 2).contiguous()
    got = S.scatter(mid.cuda(), idx.cuda(), dim=1, dim_size=n, reduce="mean")
    want = S.scatter(mid, idx, dim=1, dim_size=n, reduce="mean")
    assert tuple(got.shape) == (L, n, H)
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=0, atol=1e-5)
    # gradients flow through the reshapes
    x = src.cuda().requires_grad_(True)
    S.scatter_logsumexp(x, idx.cuda(), dim=0, dim_size=n).clamp(min=-50).sum().backward()
    xh = src.double().requires_grad_(True)
    S.scatter_logsumexp(xh, idx, dim=0, dim_size=n).clamp(min=-50).sum().backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), xh.grad.numpy(), rtol=0, atol=2e-5)
