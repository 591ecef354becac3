#!/usr/bin/env python
"""Summarise a scripts/gpu_profile.sh output directory: per-kernel time from the rocprofv3
kernel-trace stats, and per-launch HBM traffic from the FETCH_SIZE / WRITE_SIZE PMC passes
(FETCH_SIZE doubled for wide coalesced reads per MI355X_MICROARCH.md section HBM -- reported both ways)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def find(root, pattern):
    hits = glob.glob(os.path.join(root, "**", pattern), recursive=True)
    return hits[0] if hits else None


def short(name):
    """`k_linear_tlp<true, 0, 1>` style: function name + template arguments, namespaces and parameters dropped."""
    m = re.search(r"ptgnn_amd::(?:\(anonymous namespace\)::)?(k_\w+(?:<[^>(]*>)?)", name)
    if m:
        return m.group(1)
    m = re.search(r"rocprim::\w+::detail::(\w+)", name)
    if m:
        return "rocprim:" + m.group(1)
    m = re.search(r"at::native::(?:\(anonymous namespace\)::)?(\w+)", name)
    if m:
        return "torch:" + m.group(1)
    return name.split("(")[0][-60:]


def main(root):
    stats = find(os.path.join(root, "trace"), "*kernel_stats.csv")
    print(f"# rocprofv3 summary for {os.path.basename(root)}\n")
    if stats:
        print("## kernel-trace --stats (top 15 by total time)\n")
        print("| kernel | calls | total (us) | avg (us) | % |")
        print("|---|---|---|---|---|")
        rows = list(csv.DictReader(open(stats)))
        for r in rows[:int(os.environ.get("PROF_TOP", "15"))]:
            print(f"| {short(r['Name'])} | {r['Calls']} | {float(r['TotalDurationNs']) / 1e3:.1f} | "
                  f"{float(r['AverageNs']) / 1e3:.2f} | {r['Percentage']} |")
    traffic = defaultdict(lambda: {"launches": 0, "fetch_bytes_x2": 0.0, "write_bytes": 0.0})
    for tag, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        f = find(os.path.join(root, tag), "*counter_collection.csv")
        if not f:
            continue
        agg = defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            k = short(r["Kernel_Name"])
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
            base = traffic[k.split("<")[0]]
            if counter == "FETCH_SIZE":
                base["launches"] += 1
                base["fetch_bytes_x2"] += 2 * 1024 * float(r["Counter_Value"])
            else:
                base["write_bytes"] += 1024 * float(r["Counter_Value"])
        print(f"\n## {counter} per launch (raw counter is in KiB)\n")
        print("| kernel | launches | KiB/launch | MB/launch |" + (" MB/launch x2 (gfx950 wide-read correction) |" if counter == "FETCH_SIZE" else ""))
        print("|---|---|---|---|" + ("---|" if counter == "FETCH_SIZE" else ""))
        for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            mb = v / n * 1024 / 1e6
            extra = f" {2 * mb:.1f} |" if counter == "FETCH_SIZE" else ""
            print(f"| {k} | {n} | {v / n:.0f} | {mb:.1f} |{extra}")
    if traffic:
        out = {k: {"launches": v["launches"],
                   "fetch_bytes_per_launch": round(v["fetch_bytes_x2"] / max(v["launches"], 1)),
                   "write_bytes_per_launch": round(v["write_bytes"] / max(v["launches"], 1)),
                   "hbm_bytes_per_launch": round((v["fetch_bytes_x2"] + v["write_bytes"]) / max(v["launches"], 1))}
               for k, v in traffic.items() if k.startswith("k_")}
        with open(os.path.join(root, "traffic.json"), "w") as f:
            json.dump({"source": os.path.basename(root), "note": "FETCH_SIZE (KiB) x2 wide-read correction for "
                       "gfx950 + WRITE_SIZE (KiB), separate --pmc passes, averaged per launch", "kernels": out},
                      f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1])
