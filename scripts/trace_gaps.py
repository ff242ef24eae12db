"""Reads a rocprofv3 --kernel-trace CSV and prints, for the LAST forward of the run, every kernel's duration and the gap to the kernel
before it (us) -- where a many-launch forward spends its time.  usage: python scripts/trace_gaps.py <dir with *_kernel_trace.csv> <launches per forward>"""
import csv
import glob
import sys

d, per = sys.argv[1], int(sys.argv[2])
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
rows = sorted(({"name": r["Kernel_Name"], "s": int(r["Start_Timestamp"]), "e": int(r["End_Timestamp"])} for r in csv.DictReader(open(f))), key=lambda r: r["s"])
rows = [r for r in rows if "planes" not in r["name"].lower() or True]
last = rows[-per:]
tot_k = sum(r["e"] - r["s"] for r in last)
span = last[-1]["e"] - last[0]["s"]
print(f"last forward: {per} launches, span {span / 1e3:.1f} us, kernels {tot_k / 1e3:.1f} us, gaps {(span - tot_k) / 1e3:.1f} us")
prev = None
for r in last:
    gap = (r["s"] - prev) / 1e3 if prev else 0.0
    print(f"  gap {gap:6.2f}  run {(r['e'] - r['s']) / 1e3:7.2f}  {r['name'][:70]}")
    prev = r["e"]
