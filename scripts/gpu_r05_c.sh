#!/bin/bash
# Round 5, call C: the whole GPU suite on the tree so far + the default bench line.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
out=gpurun_out/r05c; mkdir -p $out
timeout 1800 python -m pytest tests -q -m gpu -x > $out/pytest.log 2>&1
echo "pytest rc=$?" >> $out/pytest.log; tail -12 $out/pytest.log
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err
echo "bench rc=$?"; tail -5 $out/bench.err
