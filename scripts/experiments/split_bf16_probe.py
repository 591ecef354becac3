"""EXPERIMENT driver (GPU box): split-bf16 GEMM vs the library's exact-fp32 MFMA kernel -- time and error vs
float64 on the cfg2 pre-transform shape [200 064, 128] x [128, 256] (see split_bf16_probe.hip)."""
import ctypes
import os
import subprocess
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from ptgnn_amd import ops  # noqa: E402

so = os.path.join(HERE, "libsplit_bf16_probe.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so,
                           os.path.join(HERE, "split_bf16_probe.hip")])
lib = ctypes.CDLL(so)
vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
lib.split_bf16_linear.argtypes = [vp, i64, i32, vp, i32, vp, i32, vp]

rows, K, n_out = 200_064, 128, 256
g = torch.Generator().manual_seed(0)
x = torch.randn(rows, K, generator=g).cuda()
w = (torch.randn(n_out, K, generator=g) / K ** 0.5).cuda()
y = torch.empty(rows, n_out, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
sub = slice(0, 4096)
ref64 = x[sub].double() @ w.double().t()


def run(terms):
    rc = lib.split_bf16_linear(x.data_ptr(), rows, K, w.data_ptr(), n_out, y.data_ptr(), terms, stream)
    assert rc == 0, rc


def clock(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


exact = ops.linear(x, w)
e_fp32 = float((exact[sub].double() - ref64).abs().max())
t_fp32 = clock(lambda: ops.linear(x, w))
flops = 2.0 * rows * K * n_out
print(f"exact fp32 MFMA kernel : {t_fp32:7.1f} us  {flops / t_fp32 / 1e6:6.1f} TFLOP/s  max|err| vs fp64 {e_fp32:.3e}")
for terms in (6, 3, 1):
    run(terms)
    torch.cuda.synchronize()
    err = float((y[sub].double() - ref64).abs().max())
    t = clock(lambda: run(terms))
    print(f"split bf16, {terms} products: {t:7.1f} us  {flops / t / 1e6:6.1f} TFLOP/s (fp32-equivalent)  "
          f"max|err| vs fp64 {err:.3e}  max|diff| vs fp32 kernel {float((y[sub] - exact[sub]).abs().max()):.3e}")
