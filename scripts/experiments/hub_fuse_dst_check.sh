cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_h
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_gather_update.py tests/test_gpu_pipeline.py -q -x -k "hub or long_rows or scatter or config2 or fused_table or fuzz or random or layer_matches or pipeline or config1" 2>&1 | tail -2
for i in 1 2; do
timeout 300 python bench.py --workload cfg2 --no-secondary --no-cpu-baseline --no-sharded-variants --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('cfg2', d['ms_per_step'], d['repeats']['ms_per_step_median'], {k:v['avg_ms'] for k,v in d['kernels'].items()})"
done
timeout 300 python scripts/profile_cfg1.py 5 2>/dev/null | tail -2 ) > gpurun_out/r06_h/ab2.log 2>&1
cat gpurun_out/r06_h/ab2.log
