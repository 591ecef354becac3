#!/bin/bash
# A/B: the hub list walked inside the main aggregation launch for the destination-term (MLP-MP table form) variants too
# (PTGNN_HUB_FUSE_DST=1 build, scripts/build_variant.sh fusedst gather_reduce.hip -DPTGNN_HUB_FUSE_DST=1)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_h
for lib in "" ptgnn_amd/csrc/libptgnn_amd_fusedst.so; do
  for i in 1 2; do
    PTGNN_AMD_LIB=$lib timeout 300 python bench.py --workload cfg2 --no-secondary --no-cpu-baseline --no-sharded-variants --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('cfg2 lib=[$lib]', d['ms_per_step'], d['repeats']['ms_per_step_median'], {k:v['avg_ms'] for k,v in d['kernels'].items()})"
  done
  PTGNN_AMD_LIB=$lib timeout 300 python scripts/profile_cfg1.py 5 2>/dev/null | tail -2
done > gpurun_out/r06_h/ab.log 2>&1
PTGNN_AMD_LIB=ptgnn_amd/csrc/libptgnn_amd_fusedst.so timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "hub or long_rows or scatter or config2 or fused_table" 2>&1 | tail -2 >> gpurun_out/r06_h/ab.log
cat gpurun_out/r06_h/ab.log
