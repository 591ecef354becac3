"""k_stream_linear_ring against the kernels that otherwise take the shape (HIP events, median).  Run ON THE GPU BOX."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptgnn_amd import ops  # noqa: E402


def t_med(fn, reps=15):
    for _ in range(3):
        fn()
    evs = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record()
        evs.append((s, e))
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in evs)[reps // 2]


g = torch.Generator().manual_seed(5)
res = []
for rows, k, n_out in [(115772, 384, 128), (115772, 256, 128), (115772, 128, 128), (200000, 256, 128),
                       (200000, 128, 128), (1250000, 512, 256), (1250000, 256, 256)]:
    x = torch.randn(rows, k, generator=g).cuda()
    w = (torch.randn(n_out, k, generator=g) / k ** 0.5).cuda()
    row = {"rows": rows, "k": k, "n_out": n_out}
    for name, env in (("default", None), ("ring", "1"), ("no_ring", "0"), ("bn64_resident", "bn64")):
        os.environ.pop("PTGNN_AMD_LINEAR_BN", None)
        if env is None:
            os.environ.pop("PTGNN_AMD_LINEAR_RING", None)
        elif env == "bn64":      # 64-column slabs: the weights stay resident where the 128-column form needs the ring
            os.environ.pop("PTGNN_AMD_LINEAR_RING", None)
            os.environ["PTGNN_AMD_LINEAR_BN"] = "64"
        else:
            os.environ["PTGNN_AMD_LINEAR_RING"] = env
        ms = t_med(lambda: ops.linear(x, w))
        row[name] = {"us": round(ms * 1e3, 1), "frac": round(2.0 * rows * k * n_out / ms / 1e9 / 157.3, 3)}
    os.environ.pop("PTGNN_AMD_LINEAR_BN", None)
    res.append(row)
    print(json.dumps(row), flush=True)
