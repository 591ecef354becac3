#!/bin/bash
# whole GPU suite + bench line (+ optional extra command): scripts/gpu_r04_full.sh <tag>
tag=${1:-r04full}
out=gpurun_out/$tag; mkdir -p $out
cd "$(dirname "$0")/.."
timeout 1800 python -m pytest tests -q -m gpu -x > $out/pytest.log 2>&1
echo "pytest rc=$?" >> $out/pytest.log; tail -12 $out/pytest.log
timeout 700 python bench.py > $out/bench.json 2> $out/bench.err
echo "bench rc=$?"; tail -2 $out/bench.err
python - "$out" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]+'/bench.json').read().strip().split('\n')[-1])
    print('ms_per_step', d['ms_per_step'], 'value', d['value'], 'frac', d['roofline']['frac'], d['roofline']['kernel'])
    print({k:(v.get('avg_ms'), v.get('frac')) for k,v in d['kernels'].items()})
    print('sustained', d.get('sustained'))
    for t in d['graph2class_train']:
        print(t['dropout'], t['ms_per_train_step'], {k:(v['avg_ms'], v['frac']) for k,v in t['kernels_over_4_steps'].items()})
    r=d['readme_default_arch']; print('readme', r['ms_per_train_step'], r.get('ms_per_forward'), {k:(v['avg_ms'], v['frac']) for k,v in r['kernels_over_4_steps'].items()})
    print('cfg4', d['config4'].get('ms_per_step'), d['config4'].get('parity'))
    print('cfg2', d['config2'].get('ms_per_step'), 'cfg5', {k:v for k,v in d['config5_shard'].items() if 'ms' in k})
    print('split', d['split_bf16'].get('ms_per_step'))
except Exception as e:
    print('summary failed', repr(e))
PY
