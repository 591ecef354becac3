import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ptgnn_amd import ops
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
f=lambda: ops.gru_cell(a,hh,cell.weight_ih,cell.weight_hh,cell.bias_ih,cell.bias_hh)
x=torch.randn(200000,128,generator=g).cuda(); w=torch.randn(256,128,generator=g).cuda()
f2=lambda: ops.linear(x,w)
for mode in (1,2):
    ops.set_gemm_mode(mode)
    for dp in (None,"0","2","6","12","24","40"):
        if dp is None: os.environ.pop("PTGNN_AMD_DEPHASE",None)
        else: os.environ["PTGNN_AMD_DEPHASE"]=dp
        print(f"mode {mode} dephase {dp}: gru {clock(f):.1f} us  linear {clock(f2):.1f} us", flush=True)
