"""Development: scans an AMDGPU assembly listing (hipcc -S --cuda-device-only) for readers of an MFMA result that follow the MFMA too
closely.  The compiler pads its OWN instructions with s_nop; it does not look inside inline asm (;;#ASMSTART ... ;;#ASMEND), so a v_cvt / v_fma
written in asm that consumes an accumulator right behind the MFMA reads the old value on some issues.  Wait states: one per instruction, s_nop N = N + 1, and an MFMA in between
counts its passes (the next instruction of the wave issues only when the matrix pipe has taken it, which for back-to-back MFMAs is when the
previous one has gone through).  usage: isa_mfma_hazards.py file.s [function-substring]"""
import re, sys

# XDL write VGPR -> VALU / LDS / VMEM read, by passes (16, 8, 4) + 2; the 16x16x128 8-bit form is 8 passes (+ 3 = 11: what the
# compiler pads it with), the 32x32x64 one 16.  First match wins.
NEED = {"16x16x128_f8f6f4": 11, "f8f6f4": 18, "32x32": 10, "16x16": 6}
reg = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")


def regs(tok):
    out = set()
    for m in reg.finditer(tok):
        if m.group(1): out |= {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
        else: out.add((m.group(4), int(m.group(5))))
    return out


def scan(lines, name):
    pending = {}          # register -> (wait states still required, mfma line no)
    branch_state = {}     # label -> the pending state the branches to it carry (worst case)
    in_asm = False
    hits = 0
    for no, raw in lines:
        ln = raw.split(";")[0].strip() if not raw.strip().startswith(";") else raw.strip()
        if ln.startswith((";APP", ";;#ASMSTART")): in_asm = True; continue
        if ln.startswith((";NO_APP", ";;#ASMEND")): in_asm = False; continue
        if ln.endswith(":") and not ln.startswith(";"):
            # a label: besides what falls through from the line above, the block is entered by the branches that name it.  Their state
            # is the one recorded at the conditional branch (taken-branch latency not modelled: a few cycles in the code's favour, never
            # against it) -- kept per target label; an unknown entry (a backward branch further down) adds nothing, as before
            for r, v in branch_state.get(ln[:-1], {}).items():
                if r not in pending or pending[r][0] < v[0]: pending[r] = v
            continue
        if not ln or ln.startswith((".", ";")): continue
        op = ln.split()[0]
        args = ln[len(op):]
        parts = [a.strip() for a in args.split(",")]
        ws = 1
        if op == "s_nop": ws = int(parts[0]) + 1
        if op.startswith("v_mfma"):
            need = next(v for k, v in NEED.items() if k in op)
            ws = need - 2
            # srcC == dst accumulate chains are the hardware's business; A / B operands read from pending registers are checked below
            srcs = set().union(*[regs(p) for p in parts[1:3]])
            for r in srcs & pending.keys():
                if pending[r][0] > 0:
                    print(f"{name}:{no}: MFMA operand {r} written by the MFMA at line {pending[r][1]}, {pending[r][0]} wait states short"); hits += 1
            for r in pending: pending[r] = (pending[r][0] - ws, pending[r][1])
            for r in regs(parts[0]): pending[r] = (need, no)
            continue
        if op.startswith(("v_", "ds_", "buffer_", "global_", "flat_")):
            touched = set().union(*[regs(p) for p in parts]) if parts else set()
            for r in touched & pending.keys():
                if pending[r][0] > 0:
                    print(f"{name}:{no}: {'ASM ' if in_asm else ''}{op} touches {r[0]}{r[1]} written by the MFMA at line {pending[r][1]}, {pending[r][0]} wait states short")
                    hits += 1
                    break
            for r in touched: pending.pop(r, None) if op.startswith("v_") and r in regs(parts[0]) else None
        if op.startswith("s_cbranch") or op == "s_branch":
            tgt = parts[0].strip() if parts else ""
            if tgt:                                   # what a taken branch carries to its target (worst case over the branches seen so far);
                st = branch_state.setdefault(tgt, {})  # the branch occupies an issue slot itself
                for r, v in pending.items():
                    if v[0] - 1 > 0 and (r not in st or st[r][0] < v[0] - 1): st[r] = (v[0] - 1, v[1])
        if op in ("s_branch", "s_endpgm", "s_setpc_b64"):
            # nothing falls through an unconditional jump: what the listing prints next is another path's code, reached by a branch of
            # its own (whose taken-branch latency is not modelled: conditional branches keep the linear order, which only SHORTENS
            # real distances)
            pending.clear()
            continue
        for r in list(pending):
            pending[r] = (pending[r][0] - ws, pending[r][1])
            if pending[r][0] <= 0: del pending[r]
    return hits


text = open(sys.argv[1]).read().split("\n")
want = sys.argv[2] if len(sys.argv) > 2 else ""
cur, body, total = None, [], 0
for i, l in enumerate(text, 1):
    m = re.match(r"^(_Z\w+):", l)
    if m: cur, body = m.group(1), []
    elif l.startswith(".Lfunc_end") and cur:
        if want in cur: n = scan(body, cur[:60]); total += n; print(f"{cur[:80]}: {n} short distances")
        cur = None
    elif cur: body.append((i, l))
print("total", total)
