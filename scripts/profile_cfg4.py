"""configs[3] (VarMisuse MLP-MP stack, hidden 64, T = 21) on ONE GPU, unsharded: the loop rocprofv3 wraps to see
where a cfg4 forward goes (bench.config4; bench.py times the same stack through sharded.run_stack at N > 1).
usage: python scripts/profile_cfg4.py [iters] [mlp|ggnn]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchmarks import varmisuse as bench  # noqa: E402

if __name__ == "__main__":
    r = bench.config4(torch.device("cuda", 0), k=int(sys.argv[1]) if len(sys.argv) > 1 else 20, parity=False,
                      arch=sys.argv[2] if len(sys.argv) > 2 else "mlp")
    print(f"{r['workload']}: {r['ms_per_forward']:.3f} ms per forward ({r['ms_per_forward'] / 8:.3f} ms per MP layer)")
