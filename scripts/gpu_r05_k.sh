#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
out=gpurun_out/r05k; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_gather_update.py tests/test_gpu_edge_features.py tests/test_gpu_two_rank.py -q -m gpu > $out/tests.log 2>&1
echo "tests rc=$?" >> $out/tests.log; tail -5 $out/tests.log
timeout 300 python scripts/profile_cfg4.py 40 2>&1 | tail -1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
