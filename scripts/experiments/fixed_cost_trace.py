"""Kernel durations (rocprofv3 --kernel-trace) of the edge GEMM and the fused GRU against the problem size: the
intercept of t(size) is the part of a launch that no inner loop explains.  Run ON THE GPU BOX under
`rocprofv3 --kernel-trace --stats`; every size runs 12 launches of its own kernel instance (sizes are told apart by
the launch ORDER in the trace)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ptgnn_amd import ops  # noqa: E402

g = torch.Generator().manual_seed(3)
N, T = 116000, 17
x = torch.randn(N, 128, generator=g).cuda()
ws = [(torch.randn(128, 128, generator=g) / 11.3).cuda() for _ in range(T)]
cell = torch.nn.GRUCell(128, 128).cuda()
for per_type in (32, 512, 2048, 8192, 36772):
    adj = [(torch.randint(0, N, (per_type,), generator=g).cuda(), torch.randint(0, N, (per_type,), generator=g).cuda())
           for _ in range(T)]
    for _ in range(12):
        ops.edge_linear(x, adj, ws, False)
    torch.cuda.synchronize()
for n in (2048, 16384, 57886, 115772):
    a, h = torch.randn(n, 128, generator=g).cuda(), torch.randn(n, 128, generator=g).cuda()
    for _ in range(12):
        ops.gru_cell(a, h, cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh)
    torch.cuda.synchronize()
print("done")
