#!/bin/bash
out=gpurun_out/r03g; mkdir -p $out
timeout 900 python -m pytest tests -x -q -m gpu -k "gru_ring or shard_index or split_gemm_mode or watchdog or sharded or two_rank or real_process" > $out/focus.log 2>&1; tail -6 $out/focus.log
timeout 1500 python -m pytest tests -q -m gpu > $out/pytest.log 2>&1; tail -6 $out/pytest.log
timeout 600 python bench.py --sharded-variants --no-cpu-baseline --no-secondary 2> $out/bench_sharded.err | grep '^{' > $out/bench_sharded.json; tail -3 $out/bench_sharded.err
