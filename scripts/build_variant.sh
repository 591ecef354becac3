#!/bin/bash
# usage: scripts/build_variant.sh <tag> <source.hip> [-DNAME=VALUE ...]
# A/B builds of ONE translation unit: ptgnn_amd/csrc/libptgnn_amd_<tag>.so = that source compiled with the given
# macros + the other objects of the regular build (python -m ptgnn_amd.build first).  Select it at run time with
# PTGNN_AMD_LIB=ptgnn_amd/csrc/libptgnn_amd_<tag>.so.
set -e
tag=$1; src=$2; shift 2
cd "$(dirname "$0")/../ptgnn_amd/csrc"
base=${src%.*}
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -x hip "$@" -c $src -o ${base}_${tag}.o
objs=""
for o in errors csr_build gather_reduce dense_f32 stream_gemm edge_gemm edge_wgrad wgrad_stream batching row_epilogue shard_index segment_mul weighted_pool; do
  if [ "$o" = "$base" ]; then objs="$objs ${base}_${tag}.o"; else objs="$objs $o.o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o libptgnn_amd_${tag}.so $objs
echo built libptgnn_amd_${tag}.so
