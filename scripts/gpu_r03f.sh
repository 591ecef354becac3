#!/bin/bash
out=gpurun_out/r03f; mkdir -p $out
timeout 600 python -m pytest tests -x -q -m gpu -k "gru or csr_build_bit_exact_beyond or odd_widths" > $out/focus.log 2>&1; tail -4 $out/focus.log
python scripts/gru_bench.py 2>/dev/null | tail -1 > $out/gru_bench.json; cat $out/gru_bench.json
python scripts/plan_bench.py 2>/dev/null | tail -1 > $out/plan_bench.json; cat $out/plan_bench.json
timeout 1500 python -m pytest tests -q -m gpu -x > $out/pytest.log 2>&1; tail -4 $out/pytest.log
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; tail -12 $out/bench.err
timeout 600 python bench.py --sharded-variants --no-cpu-baseline --no-secondary 2> $out/bench_sharded.err | grep '^{' > $out/bench_sharded.json; tail -3 $out/bench_sharded.err
