"""Run ON THE GPU BOX with PTGNN_AMD_LIB=<variant .so>: time the streaming GRU / linear / edge kernels (mode 1)."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ptgnn_amd import ops, workloads
def clock(fn, n=20, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize(); ts=[]
    for _ in range(5):
        t0=time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); ts.append((time.perf_counter()-t0)/n)
    return min(ts)*1e6
g=torch.Generator().manual_seed(1)
n,m,h=115772,128,128
a=torch.randn(n,m,generator=g).cuda(); hh=torch.randn(n,h,generator=g).cuda(); cell=torch.nn.GRUCell(m,h).cuda()
x=torch.randn(200000,128,generator=g).cuda(); w=torch.randn(256,128,generator=g).cuda()
mb = workloads.batched_graphs(48, 2500, 8, 2.2, seed=1234); N = mb["num_nodes"]
adj = [(s.cuda(), d.cuda()) for s, d in mb["adjacency_lists"]]; adj = adj + [(d, s) for s, d in adj]
ar = torch.arange(N, device="cuda"); adj.append((ar, ar))
xe = torch.randn(N, 128, generator=g).cuda(); ws = [(torch.randn(128, 128, generator=g) / 11.3).cuda() for _ in adj]
ops.set_gemm_mode(1)
print(os.path.basename(os.environ.get("PTGNN_AMD_LIB","default")),
      f"gru {clock(lambda: ops.gru_cell(a,hh,cell.weight_ih,cell.weight_hh,cell.bias_ih,cell.bias_hh)):.1f} us",
      f"linear256 {clock(lambda: ops.linear(x,w)):.1f} us",
      f"edge {clock(lambda: ops.edge_linear(xe, adj, ws, False)):.1f} us", flush=True)
