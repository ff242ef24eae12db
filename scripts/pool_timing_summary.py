"""Development: sums the `pool timing:` lines a run with CRA_POOL_TIMING=1 prints (one per SearchPool::run)."""
import re, sys
w = a = c = pc = sub = 0.0
b = n = 0
for l in open(sys.argv[1]):
    m = re.search(r"wait ([\d.]+) ms, apply ([\d.]+) ms, collect\+submit ([\d.]+) ms \(parallel collect ([\d.]+) \[items: sum ([\d.]+), sum of per-batch max ([\d.]+)\], submit ([\d.]+)\), batches (\d+)", l)
    if m:
        w += float(m[1]); a += float(m[2]); c += float(m[3]); pc += float(m[4]); sub += float(m[7]); b += int(m[8]); n += 1
print(f"runs {n}, batches {b}: wait {w:.1f} ms, apply {a:.1f} ms, collect+submit {c:.1f} ms (parallel collect {pc:.1f}, submit {sub:.1f})")
