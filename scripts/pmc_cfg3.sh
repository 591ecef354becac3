#!/bin/bash
# Matrix-pipe duty and wait shares of the kernels of the headline step (run ON THE GPU BOX): scripts/pmc_cfg3.sh <tag>
TAG=${1:-r03_cfg3_duty}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -o a -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-sustained > $OUT/a.log 2>&1
python - <<PY > $OUT/summary.txt
import csv, glob, collections
f = glob.glob("$OUT/a/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    kn = r["Kernel_Name"]
    if "ptgnn_amd" in kn:
        short = kn.split("ptgnn_amd::(anonymous namespace)::")[1].split("(")[0] if "(anonymous namespace)::" in kn else kn
        agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("kernel | launches | GUI_ACTIVE/8 (cycles per XCD) | MFMA_BUSY/1024 (cycles per SIMD) | matrix-pipe busy | WAIT_ANY / WAVE_CYCLES")
for kn, d in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("GRBM_GUI_ACTIVE", [0]))):
    n = len(d["GRBM_GUI_ACTIVE"])
    gui = sum(d["GRBM_GUI_ACTIVE"]) / n / 8
    mf = sum(d.get("SQ_VALU_MFMA_BUSY_CYCLES", [0])) / n / 1024
    wait = sum(d.get("SQ_WAIT_ANY", [0])) / max(sum(d.get("SQ_WAVE_CYCLES", [1])), 1)
    print(f"{kn[:60]:60s} | {n:4d} | {gui:10.0f} | {mf:10.0f} | {mf / gui if gui else 0:6.3f} | {wait:6.3f}")
PY
cat $OUT/summary.txt
find $OUT -name "*.csv" -size +1M -delete; find $OUT -name "*.db" -delete
