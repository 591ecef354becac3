import sys, json, torch
sys.path.insert(0, '/root/repo')
from benchmarks import synthetic as bench
dev = torch.device('cuda:0')
for i in range(2):
    r = bench.config5_shard(dev, parity=False)
    print(json.dumps({k: (v if not isinstance(v, dict) else {a: (b if not isinstance(b, dict) else {c: d.get('avg_ms') for c, d in b.items()}) for a, b in v.items() if a in ('ms_per_layer_step', 'kernels')}) for k, v in r.items() if k in ('ggnn_layer', 'mlp_mp_layer')}))
