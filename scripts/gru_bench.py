"""Fused GRU cell timings (HIP events, median): ring / resident-slab / tile kernels at a few shapes."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptgnn_amd import ops  # noqa: E402


def t_med(fn, reps=9):
    for _ in range(2):
        fn()
    evs = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record()
        evs.append((s, e))
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in evs)[reps // 2]


res = {}
for n, m, h in ((1_250_000, 256, 256), (115_772, 128, 128), (115_772, 128, 256)):
    g = torch.Generator().manual_seed(1)
    args = [torch.randn(n, m, generator=g).cuda(), torch.randn(n, h, generator=g).cuda(),
            (torch.randn(3 * h, m, generator=g) / m ** 0.5).cuda(), (torch.randn(3 * h, h, generator=g) / h ** 0.5).cuda(),
            torch.zeros(3 * h).cuda(), torch.zeros(3 * h).cuda()]
    fl = 2.0 * n * 3 * h * (m + h)
    row = {}
    for name, env, mode in (("ring", "1", "stream"), ("default", None, "stream"), ("tile", "0", "tile")):
        if env is None:
            os.environ.pop("PTGNN_AMD_GRU_RING", None)
        else:
            os.environ["PTGNN_AMD_GRU_RING"] = env
        prev = ops.set_gemm_mode(mode)
        ms = t_med(lambda: ops.gru_cell(*args))
        ops.set_gemm_mode(prev)
        row[name] = {"ms": round(ms, 4), "tflops": round(fl / ms / 1e9, 1), "frac": round(fl / ms / 1e9 / 157.3, 3)}
    res[f"n{n}_m{m}_h{h}"] = row
os.environ.pop("PTGNN_AMD_GRU_RING", None)
print(json.dumps(res))
