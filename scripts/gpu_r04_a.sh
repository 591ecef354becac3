#!/bin/bash
# round 4, call A: fence-free edge GEMM + bit-mask dropout -- focus tests, whole suite, training steps, bench line
tag=${1:-r04a}
out=gpurun_out/$tag; mkdir -p $out
cd "$(dirname "$0")/.."
timeout 600 python -m pytest tests/test_gpu_edge_stream.py -x -q -m gpu > $out/focus.log 2>&1
echo "focus rc=$?" >> $out/focus.log; tail -15 $out/focus.log
timeout 1500 python -m pytest tests -q -m gpu -x > $out/pytest.log 2>&1
echo "pytest rc=$?" >> $out/pytest.log; tail -8 $out/pytest.log
TRAIN_STEPS=10 timeout 300 python scripts/train_bench.py > $out/train_ggnn.log 2>&1; tail -3 $out/train_ggnn.log
TRAIN_ARCH=mlp TRAIN_DROPOUTS=0.1 TRAIN_STEPS=10 timeout 300 python scripts/train_bench.py > $out/train_mlp.log 2>&1; tail -2 $out/train_mlp.log
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err
echo "bench rc=$?"; tail -3 $out/bench.err
python - <<'PY'
import json,sys
try:
    d=json.loads(open('gpurun_out/%s/bench.json' % sys.argv[1] if len(sys.argv)>1 else 'gpurun_out/r04a/bench.json').read().strip().split('\n')[-1])
    print('ms_per_step', d['ms_per_step'], 'value', d['value'], 'roofline', d['roofline'])
    print({k:(v.get('avg_ms'), v.get('frac')) for k,v in d['kernels'].items()})
    for t in d['graph2class_train']:
        print(t['dropout'], t['ms_per_train_step'], {k:(v['avg_ms'], v['frac']) for k,v in t['kernels_over_4_steps'].items()})
    r=d['readme_default_arch']; print('readme', r['ms_per_train_step'], r.get('ms_per_forward'))
except Exception as e:
    print('summary failed', e)
PY
