import json, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ptgnn_amd import ops, workloads
os.environ["PTGNN_AMD_EDGE_R64"] = "0"
res = {"lib": os.path.basename(os.environ.get("PTGNN_AMD_LIB", "default"))}
g = torch.Generator().manual_seed(3)
for per_type in (64, 4096, 36772):      # edges per type, T = 17: fixed cost, 1/9 of cfg3, cfg3 size
    N = 116000
    adj = [(torch.randint(0, N, (per_type,), generator=g).cuda(), torch.randint(0, N, (per_type,), generator=g).cuda()) for _ in range(17)]
    for K, M in ((128, 128), (64, 128), (256, 128), (128, 64)):
        x = torch.randn(N, K, generator=g).cuda()
        ws = [(torch.randn(M, K, generator=g) / 11.3).cuda() for _ in adj]
        fn = lambda: ops.edge_linear(x, adj, ws, False)
        for _ in range(20): fn()
        evs = []
        for _ in range(31):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); fn(); e.record(); evs.append((s, e))
        torch.cuda.synchronize()
        t = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
        res[f"E{per_type * 17}_K{K}_M{M}"] = round(t[0], 1)
print(json.dumps(res))
