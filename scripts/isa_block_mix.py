"""Development: instruction mix of the MFMA-bearing basic blocks of a kernel in a `hipcc -S` listing (what a matrix wave issues per
interval besides its MFMAs: a wave pays a few cycles of issue time for ANY instruction).  usage: isa_block_mix.py file.s kernel-substring"""
import collections, re, sys
lines = open(sys.argv[1]).read().split("\n")
want = sys.argv[2]
start = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and want in l][0]
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
blocks, cur, name = [], [], "entry"
for l in lines[start:end]:
    if re.match(r"^\.LBB\d+_\d+:", l):
        blocks.append((name, cur)); name = l.split(":")[0]; cur = []
    else:
        t = l.strip()
        if t and not t.startswith((";", ".")): cur.append(t)
blocks.append((name, cur))
tot = collections.Counter()
for name, b in blocks:
    n = sum(1 for x in b if x.startswith("v_mfma"))
    if n < 8: continue
    c = collections.Counter()
    for x in b:
        op = x.split()[0]
        k = ("mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else
             "vmem" if op.startswith(("buffer_", "global_")) else "waitcnt" if op == "s_waitcnt" else "nop" if op == "s_nop" else
             "salu" if op.startswith("s_") else "other")
        c[k] += 1
    tot += c
    print(f"{name:12s} {len(b):4d}  {dict(sorted(c.items()))}")
print("all MFMA blocks:", dict(sorted(tot.items())), "non-MFMA per MFMA: %.2f" % ((sum(tot.values()) - tot["mfma"]) / max(1, tot["mfma"])))
