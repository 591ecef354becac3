#!/bin/bash
out=gpurun_out/r03p; mkdir -p $out
export OMP_NUM_THREADS=4 PTGNN_AMD_BENCH_BACKEND=gloo PTGNN_AMD_BENCH_SHARE_GPU=1 PTGNN_AMD_BENCH_VARIANT_DEADLINE=60
for side in 0 1; do
if [ $side = 1 ]; then export PTGNN_AMD_SIDE_MIN_EDGES=0; fi
timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 2961$side bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $out/b$side.json 2> $out/b$side.err
echo "side=$side rc=$?"; grep "bench " $out/b$side.err | tail -8
done
