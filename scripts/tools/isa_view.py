#!/usr/bin/env python3
"""Condensed view of one kernel in a hipcc -save-temps .s file: labels, branches, waits, VMEM ops, barriers, with the
MFMA / VALU / DS instruction counts between them.   usage: isa_view.py file.s <substring of mangled kernel name> [--full]"""
import re, sys
path, pat = sys.argv[1], sys.argv[2]
full = "--full" in sys.argv
lines = open(path).read().split("\n")
start = None
for i, l in enumerate(lines):
    if re.match(r"^[A-Za-z_][\w$.]*:", l) and pat in l.split(":")[0]:
        start = i; break
assert start is not None, "kernel not found"
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
body = lines[start:end + 1]
if full:
    print("\n".join(body)); sys.exit(0)
cnt = {"mfma": 0, "valu": 0, "ds": 0, "salu": 0}
def flush():
    if any(cnt.values()):
        print("      ... " + " ".join(f"{k}={v}" for k, v in cnt.items() if v))
    for k in cnt: cnt[k] = 0
for l in body:
    s = l.strip()
    if not s or s.startswith(";") or s.startswith("."):
        if s.startswith(".LBB"): flush(); print(s)
        continue
    op = s.split()[0]
    if re.match(r"\.?LBB", s) or s.endswith(":"):
        flush(); print(s); continue
    if op.startswith("v_mfma"): cnt["mfma"] += 1
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_", "s_waitcnt", "s_barrier", "s_cbranch", "s_branch", "s_endpgm", "s_load", "s_buffer")):
        flush(); print("  " + s.split(";")[0].strip())
    elif op.startswith("ds_"): cnt["ds"] += 1
    elif op.startswith("v_"): cnt["valu"] += 1
    elif op.startswith("s_"): cnt["salu"] += 1
flush()
