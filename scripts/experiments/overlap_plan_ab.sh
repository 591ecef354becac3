cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_e
for ov in 0 1; do
  for i in 1 2; do
    PTGNN_AMD_OVERLAP_PLAN=$ov timeout 300 python bench.py --workload cfg2 --no-secondary --no-cpu-baseline --no-sharded-variants --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('cfg2 overlap=$ov', d['ms_per_step'], d['repeats']['ms_per_step_median'], {k:v['avg_ms'] for k,v in d['kernels'].items()})"
  done
done >> gpurun_out/r06_e/overlap.log 2>&1
for ov in 0 1; do
  PTGNN_AMD_OVERLAP_PLAN=$ov timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-sharded-variants --no-sustained 2>/dev/null | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('cfg3 overlap=$ov', d['ms_per_step'], d['repeats']['ms_per_step_median'])"
done >> gpurun_out/r06_e/overlap.log 2>&1
cat gpurun_out/r06_e/overlap.log
