"""Run ON THE GPU BOX: is a GRU row's result independent of where the row sits (slice vs full matrix)?"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ptgnn_amd import ops
g = torch.Generator().manual_seed(1)
for (n, m, h) in ((3000, 64, 64), (3000, 128, 128)):
    a = torch.randn(n, m, generator=g).cuda(); hh = torch.randn(n, h, generator=g).cuda(); cell = torch.nn.GRUCell(m, h).cuda()
    f = lambda a_, h_: ops.gru_cell(a_, h_, cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh)
    for mode in (0, 1, 2):
        ops.set_gemm_mode(mode)
        full, full2 = f(a, hh), f(a, hh)
        print(n, m, h, "mode", mode, "repeat equal:", torch.equal(full, full2))
        for lo, hi in ((0, 1003), (1003, 2107), (2107, 3000), (17, 49)):
            part = f(a[lo:hi].contiguous(), hh[lo:hi].contiguous())
            d = (part != full[lo:hi])
            bad = d.any(1).nonzero().flatten()
            print("   slice", lo, hi, "mismatch elems", int(d.sum()), "rows", int(bad.numel()), "max abs",
                  float((part - full[lo:hi]).abs().max()), "first bad rows", bad[:6].tolist(),
                  "bad cols", d[bad[0]].nonzero().flatten()[:8].tolist() if bad.numel() else None)
