#!/bin/bash
# Round 5, call G: tests touched since the final profile run + a fresh default bench line.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
out=gpurun_out/r05g; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_edge_features.py tests/test_gpu_gather_update.py tests/test_gpu_golden_wide.py tests/test_gpu_pipeline.py -q -m gpu > $out/tests.log 2>&1
echo "tests rc=$?" >> $out/tests.log; tail -6 $out/tests.log
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err
echo "bench rc=$?"; tail -3 $out/bench.err
