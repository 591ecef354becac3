#!/bin/bash
# round 4, call B: why is the fence-free edge GEMM not faster?  isolated timings of the probe builds + PMC duty
tag=${1:-r04b}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
out=$ROOT/gpurun_out/$tag; mkdir -p $out
cd $ROOT
for lib in "" v2FENCE v2EPI_NQ v2SINK; do
  if [ -n "$lib" ]; then export PTGNN_AMD_LIB=$ROOT/ptgnn_amd/csrc/libptgnn_amd_$lib.so; else unset PTGNN_AMD_LIB; fi
  timeout 120 python scripts/experiments/edge_probe.py 2>&1 | tail -1 | tee -a $out/probe.log
done
unset PTGNN_AMD_LIB
PTGNN_AMD_EDGE_V2=0 timeout 120 python scripts/experiments/edge_probe.py 2>&1 | tail -1 | sed 's/default/v2off/' | tee -a $out/probe.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $out/avail.txt 2>&1
for mode in v2 old; do
  if [ $mode = old ]; then export PTGNN_AMD_EDGE_V2=0; else unset PTGNN_AMD_EDGE_V2; fi
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_$mode -o a -- python $ROOT/scripts/experiments/edge_probe.py > $out/pmc_$mode.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS --output-format csv -d $out/pmc2_$mode -o a -- python $ROOT/scripts/experiments/edge_probe.py > $out/pmc2_$mode.log 2>&1
done
python - <<PY > $out/pmc_summary.txt
import csv, glob, collections
for mode in ("v2", "old"):
    for pre in ("pmc", "pmc2"):
        f = glob.glob("$out/%s_%s/**/*counter_collection.csv" % (pre, mode), recursive=True)
        if not f:
            print(mode, pre, "no csv"); continue
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f[0])):
            kn = r["Kernel_Name"]
            if "k_stream" in kn:
                short = kn.split("::")[-1].split("(")[0]
                agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for kn, d in agg.items():
            print(mode, pre, kn, {c: round(sum(v) / len(v)) for c, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
cat $out/pmc_summary.txt
find $out -name "*.csv" -size +1M -delete; find $out -name "*.db" -delete
