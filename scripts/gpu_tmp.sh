#!/bin/bash
out=gpurun_out/r03l; mkdir -p $out
timeout 900 python -m pytest tests -x -q -m gpu -k "edge or config3 or config4 or golden or sharded" > $out/focus.log 2>&1; tail -3 $out/focus.log
for e in 0 1; do
PTGNN_AMD_EDGE_TWO_SLABS=$e timeout 600 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('two_slabs=$e', d['ms_per_step'], {k:(v['avg_ms'],v['frac']) for k,v in d['kernels'].items()}, d['repeats']['ms_per_step_median'])"
done
python scripts/profile_cfg4.py 20
PTGNN_AMD_EDGE_TWO_SLABS=0 python scripts/profile_cfg4.py 20
