"""configs[0] as the reference batches it (benchmarks/ppi.py: 3 000-node cap, ~14 minibatches): the loop rocprofv3 wraps to see
where a small minibatch goes, kernel by kernel.  usage: python scripts/profile_cfg1.py [passes] [ggnn64|ppi_arch_mlp256]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchmarks import ppi  # noqa: E402

if __name__ == "__main__":
    r = ppi.config1(torch.device("cuda", 0), parity=False, passes=int(sys.argv[1]) if len(sys.argv) > 1 else 5)
    for key in ("ggnn64", "ppi_arch_mlp256"):
        e = r[key]
        print(f"{key}: {e['ms_per_minibatch']:.4f} ms per minibatch ({e['ms_per_layer']:.4f} per layer), "
              f"{e['c_abi_launches_per_layer']} launches per layer, kernels sum {e['device_ms_per_minibatch_sum_of_kernels']:.4f} ms")
