"""Training-step timing of a Typilus stack on the Graph2Class-style batch (fwd + bwd + Adam) -- a thin driver
around bench.train_cfg3 so the step can be profiled on its own.
  TRAIN_DROPOUTS=0.0,0.1  TRAIN_ARCH=ggnn|mlp  TRAIN_HIDDEN=128  TRAIN_STEPS=10"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchmarks import graph2class as bench  # noqa: E402

dev = torch.device("cuda:0")
arch = os.environ.get("TRAIN_ARCH", "ggnn")
hidden = int(os.environ.get("TRAIN_HIDDEN", "128" if arch == "ggnn" else "64"))
steps = int(os.environ.get("TRAIN_STEPS", "10"))
for dropout in [float(v) for v in os.environ.get("TRAIN_DROPOUTS", "0.0,0.1").split(",")]:
    r = bench.train_cfg3(dev, dropout, steps=steps, warmup=3, arch=arch, H=hidden, forward_too=True)
    print(f"arch={arch} H={hidden} dropout={dropout}: {r['ms_per_train_step']:.2f} ms/train step  "
          f"{r['edges_per_sec_readme_convention'] / 1e6:.1f} M edges/s (README convention; V100 README training "
          f"1.129 M)  forward {r['ms_per_forward']:.2f} ms  peak mem {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB")
