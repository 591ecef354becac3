"""TEST INFRASTRUCTURE ONLY -- never imported by the product package `ptgnn_amd`.

CPU restatement of the *published* algorithm of the third-party library `torch_scatter`
(pinned ``>=2.0.5`` by the reference, setup.py:23; CI at 2.0.6, .github/workflows/tests.yml:17),
which the reference calls at exactly one hot-path site:
ptgnn/neuralmodels/gnn/messagepassing/abstractmessagepassing.py:44-50.
The library is neither vendored under /root/reference nor installable here, so this file
restates torch_scatter/scatter.py + csrc/cpu/scatter_cpu.cpp (2.0.x):

  * a 1-D index is broadcast to src's shape along `dim`
  * sum : zeros(dim_size).scatter_add_(dim, index, src)          (edge order preserved)
  * mean: sum; count = scatter_sum(ones); count[count < 1] = 1; out /= count
  * max/min: reduce; segments that receive no element are 0; arg = winning position,
    src.size(dim) for empty segments; ``scatter(reduce="max")`` returns values only
  * mul : ones(dim_size).scatter_(dim, index, src, reduce="multiply"): the product in edge order; a segment that
    receives no element stays 1 (Reducer<MUL>::init() and no masked_fill afterwards, unlike max/min)
  * scatter_log_softmax: src - max_seg - log(sum_seg exp(src - max_seg) + eps)

PARITY STATUS: the reference's own tests hold no golden vectors for this path
(ptgnn/tests/simplemodel only), so these semantics are pinned by (a) hand-computed
known-answer tests in tests/test_oracle_kat.py, (b) the independent serial C restatement in
oracle/scatter_ref.c, and (c) golden fixtures produced by the reference's own MP modules
running on top of this restatement (tests/golden/make_golden.py).
"""
import torch


def _broadcast(index: torch.Tensor, src: torch.Tensor, dim: int) -> torch.Tensor:
    if dim < 0:
        dim = src.dim() + dim
    if index.dim() == 1:
        for _ in range(0, dim):
            index = index.unsqueeze(0)
    for _ in range(index.dim(), src.dim()):
        index = index.unsqueeze(-1)
    return index.expand(src.size())


def _dim_size(index, dim_size):
    if dim_size is not None:
        return int(dim_size)
    return int(index.max()) + 1 if index.numel() > 0 else 0


def scatter_sum(src, index, dim=-1, out=None, dim_size=None):
    index = _broadcast(index, src, dim)
    if out is None:
        size = list(src.size())
        size[dim] = _dim_size(index, dim_size)
        out = torch.zeros(size, dtype=src.dtype, device=src.device)
    return out.scatter_add_(dim, index, src)


def scatter_add(src, index, dim=-1, out=None, dim_size=None):
    return scatter_sum(src, index, dim, out, dim_size)


def scatter_mean(src, index, dim=-1, out=None, dim_size=None):
    out = scatter_sum(src, index, dim, out, dim_size)
    dim_size = out.size(dim)
    index_dim = dim
    if index_dim < 0:
        index_dim = index_dim + src.dim()
    if index.dim() <= index_dim:
        index_dim = index.dim() - 1
    ones = torch.ones(index.size(), dtype=src.dtype, device=src.device)
    count = scatter_sum(ones, index, index_dim, None, dim_size)
    count[count < 1] = 1
    count = _broadcast(count, out, dim)
    if out.is_floating_point():
        out.true_divide_(count)
    else:
        out.div_(count, rounding_mode="floor")
    return out


def scatter_mul(src, index, dim=-1, out=None, dim_size=None):
    """torch_scatter.scatter_mul (torch_scatter/scatter.py: `torch.ops.torch_scatter.scatter_mul`; csrc/cpu/reducer.h
    MUL: init 1, update `*val *= new_val`), folded in edge order like the serial CPU kernel."""
    index = _broadcast(index, src, dim)
    if out is None:
        size = list(src.size())
        size[dim] = _dim_size(index, dim_size)
        out = torch.ones(size, dtype=src.dtype, device=src.device)
    return out.scatter_reduce_(dim, index, src, reduce="prod", include_self=True)


def _scatter_minmax(src, index, dim, dim_size, is_max):
    """Serial restatement of torch_scatter csrc/cpu/scatter_cpu.cpp for max/min (+arg)."""
    if dim < 0:
        dim = src.dim() + dim
    bindex = _broadcast(index, src, dim)
    size = list(src.size())
    size[dim] = _dim_size(index, dim_size)
    out = torch.zeros(size, dtype=src.dtype, device=src.device)
    red = "amax" if is_max else "amin"
    out.scatter_reduce_(dim, bindex, src, reduce=red, include_self=False)
    # arg: position along `dim` of the winning element; src.size(dim) for empty segments.
    n = src.size(dim)
    pos_shape = [1] * src.dim()
    pos_shape[dim] = n
    pos = torch.arange(n, device=src.device).view(pos_shape).expand_as(src)
    hit = src == out.gather(dim, bindex)
    cand = torch.where(hit, pos, torch.full_like(pos, n))
    arg = torch.full(size, n, dtype=torch.long, device=src.device)
    arg.scatter_reduce_(dim, bindex, cand, reduce="amin", include_self=True)
    return out, arg


def scatter_max(src, index, dim=-1, out=None, dim_size=None):
    assert out is None
    return _scatter_minmax(src, index, dim, dim_size, True)


def scatter_min(src, index, dim=-1, out=None, dim_size=None):
    assert out is None
    return _scatter_minmax(src, index, dim, dim_size, False)


def scatter(src, index, dim=-1, out=None, dim_size=None, reduce="sum"):
    if reduce in ("sum", "add"):
        return scatter_sum(src, index, dim, out, dim_size)
    if reduce == "mean":
        return scatter_mean(src, index, dim, out, dim_size)
    if reduce == "max":
        return scatter_max(src, index, dim, out, dim_size)[0]
    if reduce == "min":
        return scatter_min(src, index, dim, out, dim_size)[0]
    if reduce == "mul":
        return scatter_mul(src, index, dim, out, dim_size)
    raise ValueError(reduce)


def scatter_log_softmax(src, index, dim=-1, eps=1e-12, dim_size=None):
    index_b = _broadcast(index, src, dim)
    max_per = scatter_max(src, index_b, dim=dim, dim_size=dim_size)[0]
    recentered = src - max_per.gather(dim, index_b)
    sum_per = scatter_sum(recentered.exp(), index_b, dim, dim_size=dim_size)
    return recentered - sum_per.add_(eps).log_().gather(dim, index_b)


def scatter_softmax(src, index, dim=-1, eps=1e-12, dim_size=None):
    index_b = _broadcast(index, src, dim)
    max_per = scatter_max(src, index_b, dim=dim, dim_size=dim_size)[0]
    e = (src - max_per.gather(dim, index_b)).exp()
    s = scatter_sum(e, index_b, dim, dim_size=dim_size)
    return e / (s.gather(dim, index_b) + eps)
