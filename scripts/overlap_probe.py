import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchmarks import synthetic as bench
from ptgnn_amd import ops
dev = torch.device('cuda:0')
st = bench.make_cfg2(dev, 0, 1)
for flag in (False, True, False, True):
    ops.OVERLAP_PLAN_BUILD = flag
    for _ in range(10): bench.step_cfg2(st, 1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): bench.step_cfg2(st, 1)
    torch.cuda.synchronize(); print(flag, round((time.perf_counter()-t0)/50*1e3, 4), 'ms')
