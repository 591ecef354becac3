"""Known-answer tests for the restated third-party scatter (oracle/scatter_ref.py, .c).
Hand-computed on tiny inputs; pins empty-segment -> 0, duplicates, self loops, mean clamp,
arg semantics."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import scatter_ref as sr

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE = os.path.join(os.path.dirname(HERE), "oracle")

SRC = torch.tensor([[1.0, -2.0], [3.0, 4.0], [5.0, -6.0], [-7.0, 8.0], [0.5, 0.25]])
IDX = torch.tensor([2, 0, 2, 2, 0])  # node 1 and node 3 receive nothing; node 2 gets 3 edges
N = 4

EXPECT = {
    "sum": [[3.5, 4.25], [0, 0], [-1.0, 0.0], [0, 0]],
    "mean": [[1.75, 2.125], [0, 0], [-1.0 / 3.0, 0.0], [0, 0]],
    "max": [[3.0, 4.0], [0, 0], [5.0, 8.0], [0, 0]],
    "min": [[0.5, 0.25], [0, 0], [-7.0, -6.0], [0, 0]],
    "mul": [[1.5, 1.0], [1, 1], [-35.0, 96.0], [1, 1]],        # empty segments stay 1 (Reducer<MUL>::init)
}


@pytest.mark.parametrize("reduce", ["sum", "mean", "max", "min", "mul"])
def test_scatter_known_answers(reduce):
    out = sr.scatter(SRC, IDX, dim=0, dim_size=N, reduce=reduce)
    assert out.dtype == torch.float32 and tuple(out.shape) == (N, 2)
    np.testing.assert_allclose(out.numpy(), np.asarray(EXPECT[reduce], np.float32), rtol=0, atol=1e-7)


def test_scatter_add_alias_and_default_dim_size():
    out = sr.scatter(SRC, IDX, dim=0, reduce="add")
    assert tuple(out.shape) == (3, 2)  # index.max()+1
    np.testing.assert_array_equal(out.numpy(), np.asarray(EXPECT["sum"][:3], np.float32))


def test_scatter_max_arg_semantics():
    out, arg = sr.scatter_max(SRC, IDX, dim=0, dim_size=N)
    np.testing.assert_array_equal(arg.numpy(), [[1, 1], [5, 5], [2, 3], [5, 5]])
    out, arg = sr.scatter_min(SRC, IDX, dim=0, dim_size=N)
    np.testing.assert_array_equal(arg.numpy(), [[4, 4], [5, 5], [3, 2], [5, 5]])


def test_scatter_empty_input():
    out = sr.scatter(torch.zeros(0, 3), torch.zeros(0, dtype=torch.int64), dim=0, dim_size=5,
                     reduce="max")
    assert tuple(out.shape) == (5, 3) and float(out.abs().sum()) == 0.0


def test_scatter_all_negative_max_keeps_sign():
    # empty -> 0 must not clobber genuinely negative maxima
    out = sr.scatter(torch.tensor([[-3.0], [-1.0]]), torch.tensor([1, 1]), dim=0, dim_size=2,
                     reduce="max")
    np.testing.assert_array_equal(out.numpy(), [[0.0], [-1.0]])


def test_log_softmax_known_answer():
    src = torch.tensor([0.0, np.log(3.0), 1.0])
    out = sr.scatter_log_softmax(src, torch.tensor([0, 0, 1]), dim=0, eps=0.0)
    np.testing.assert_allclose(out.numpy(), [np.log(0.25), np.log(0.75), 0.0], atol=1e-6)


@pytest.fixture(scope="module")
def clib():
    so = os.path.join(ORACLE, "_build", "libscatter_ref.so")
    subprocess.check_call(["make", "-s", "-C", ORACLE])
    lib = ctypes.CDLL(so)
    lib.ptgnn_oracle_scatter_f32.restype = ctypes.c_int
    lib.ptgnn_oracle_scatter_f32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                             ctypes.c_int64, ctypes.c_int64, ctypes.c_int,
                                             ctypes.c_void_p, ctypes.c_void_p]
    return lib


def c_scatter(lib, src, idx, n, reduce):
    src = np.ascontiguousarray(src, np.float32)
    idx = np.ascontiguousarray(idx, np.int64)
    out = np.empty((n, src.shape[1]), np.float32)
    arg = np.empty((n, src.shape[1]), np.int64)
    rc = lib.ptgnn_oracle_scatter_f32(src.ctypes.data, idx.ctypes.data, src.shape[0], src.shape[1],
                                      n, ["sum", "mean", "max", "min", "mul"].index(reduce),
                                      out.ctypes.data, arg.ctypes.data)
    assert rc == 0
    return out, arg


@pytest.mark.parametrize("reduce", ["sum", "mean", "max", "min", "mul"])
def test_c_restatement_matches_known_answers_and_torch_restatement(clib, reduce):
    out, _ = c_scatter(clib, SRC.numpy(), IDX.numpy(), N, reduce)
    np.testing.assert_allclose(out, np.asarray(EXPECT[reduce], np.float32), rtol=0, atol=1e-7)
    rng = np.random.RandomState(7)
    src = rng.randn(5000, 33).astype(np.float32)
    if reduce == "mul":
        src = (1.0 + 0.2 * src).astype(np.float32)           # products of ~7 factors near 1
    idx = rng.randint(0, 700, size=5000)
    idx[idx == 13] = 14  # guarantee an empty segment
    out, arg = c_scatter(clib, src, idx, 701, reduce)
    ref = sr.scatter(torch.from_numpy(src), torch.from_numpy(idx), dim=0, dim_size=701, reduce=reduce)
    if reduce in ("max", "min"):
        np.testing.assert_array_equal(out, ref.numpy())      # order independent -> bit exact
        fn = sr.scatter_max if reduce == "max" else sr.scatter_min
        np.testing.assert_array_equal(arg, fn(torch.from_numpy(src), torch.from_numpy(idx), 0,
                                              dim_size=701)[1].numpy())
    elif reduce == "sum":
        np.testing.assert_array_equal(out, ref.numpy())      # same edge order -> bit exact
    else:
        np.testing.assert_allclose(out, ref.numpy(), rtol=1e-6, atol=1e-7)
    assert np.all(out[13] == (1 if reduce == "mul" else 0))
