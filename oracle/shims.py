"""TEST INFRASTRUCTURE ONLY -- never imported by the product package `ptgnn_amd`.

Import shims that let the *real* reference (microsoft/ptgnn, mounted read-only at
/root/reference) be imported in the authoring container so that golden fixtures can be
generated from the reference's own modules (see tests/golden/make_golden.py).

/root/reference does not exist on the GPU box, so nothing executed there may call
`install()`; fixtures produced here are committed under tests/golden/ and travel instead.

Two third-party dependencies of the reference are absent from the image and cannot be
installed (no network):

* ``torch_scatter`` (pinned ``>=2.0.5`` in reference setup.py:23, CI-tested at 2.0.6,
  .github/workflows/tests.yml:17).  Imported at module top by
  ptgnn/neuralmodels/gnn/messagepassing/abstractmessagepassing.py:4.  We restate its
  *published* algorithm (torch_scatter/scatter.py and csrc/cpu/scatter_cpu.cpp of 2.0.x):
    - index (1-D) is broadcast to src's shape along `dim`
    - sum : zeros(dim_size).scatter_add_(dim, index, src)
    - mean: sum, count = scatter_sum(ones), count.clamp_(min=1), out / count
    - max/min: reduce, segments that receive no element are set to 0; returns (out, arg)
      from scatter_max/scatter_min, values only from scatter(..., reduce="max")
  Because that library is not vendored, hot-path parity is pinned to "reference modules
  executed here + this restatement of torch_scatter".

* ``dpu_utils.utils.iterators`` (BufferedIterator, ThreadedIterator, shuffled_iterator),
  imported by ptgnn/baseneuralmodel/abstractneuralmodel.py:8; pass-through iterators are
  sufficient because the hot path never touches them.
"""
import os
import sys
import types

import torch

REFERENCE_ROOT = os.environ.get("PTGNN_REFERENCE_ROOT", "/root/reference")


from oracle.scatter_ref import (  # noqa: E402
    scatter, scatter_add, scatter_log_softmax, scatter_max, scatter_mean, scatter_min,
    scatter_softmax, scatter_sum,
)


def _make_torch_scatter_module() -> types.ModuleType:
    m = types.ModuleType("torch_scatter")
    for f in (scatter, scatter_sum, scatter_add, scatter_mean, scatter_max, scatter_min,
              scatter_log_softmax, scatter_softmax):
        setattr(m, f.__name__, f)
    comp = types.ModuleType("torch_scatter.composite")
    comp.scatter_log_softmax = scatter_log_softmax
    comp.scatter_softmax = scatter_softmax
    m.composite = comp
    m.__version__ = "2.0.6+restated"
    return m, comp


def _make_dpu_utils_modules():
    dpu = types.ModuleType("dpu_utils")
    utils = types.ModuleType("dpu_utils.utils")
    iters = types.ModuleType("dpu_utils.utils.iterators")

    def ThreadedIterator(original_iterator, max_queue_size=2, enabled=True):
        return iter(original_iterator)

    def BufferedIterator(original_iterator, max_queue_size=3, enabled=True):
        return iter(original_iterator)

    def shuffled_iterator(input_iterator, buffer_size=10000, out_slice_sizes=500):
        return iter(input_iterator)

    iters.ThreadedIterator = ThreadedIterator
    iters.BufferedIterator = BufferedIterator
    iters.shuffled_iterator = shuffled_iterator
    utils.iterators = iters
    # RichPath is only type-annotated on the paths we import
    utils.RichPath = type("RichPath", (), {})
    dpu.utils = utils
    return {"dpu_utils": dpu, "dpu_utils.utils": utils, "dpu_utils.utils.iterators": iters}


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "ptgnn"))


def install() -> None:
    """Make `import ptgnn...` resolve to the real reference with the two stubs in place."""
    if not reference_available():
        raise RuntimeError(
            f"reference checkout not found at {REFERENCE_ROOT}; golden fixtures can only be "
            "regenerated in the authoring container")
    if "torch_scatter" not in sys.modules:
        m, comp = _make_torch_scatter_module()
        sys.modules["torch_scatter"] = m
        sys.modules["torch_scatter.composite"] = comp
    for name, mod in _make_dpu_utils_modules().items():
        sys.modules.setdefault(name, mod)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
