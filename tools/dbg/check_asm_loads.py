#!/usr/bin/env python3
"""Reads a -save-temps ISA listing and checks the hand-issued (inline asm) tile loads of the Lloyd pass kernels: between a
global_load into v[a:b] and the s_waitcnt vmcnt(..) that follows it in the listing no instruction may read or write one of those
registers (hipcc does not know the load is in flight: a register copy there would move stale data; round 6 met exactly that).
Linear scan of the listing - the tile loop is laid out in program order -: a report, not a proof.
usage: check_asm_loads.py kmeans-hip-amdgcn-amd-amdhsa-gfx950.s [kernel-name-substring]"""
import re, sys

path = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "kmeans_pass_mfma_kernelILi1ELi3ELi5ELi4ELb1EE"
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and want in l and l.rstrip().endswith(":") or (l.startswith("_Z") and want in l and ": " in l))
regs_re = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs(text):
    out = set()
    for m in regs_re.finditer(text):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


pending = {}          # register -> line of the load
bad = loads = 0
for i in range(start + 1, len(lines)):
    l = lines[i].split(";")[0].strip()
    if not l or l.endswith(":") or l.startswith("."):
        continue
    if l.startswith("s_endpgm"):
        break
    op = l.split()[0]
    if op.startswith("global_load") and re.search(r", s\[\d+:\d+\]", l):        # the asm form: SGPR base
        dst = l.split(",")[0]
        touched = regs(l.split(",", 1)[1]) & set(pending)                       # its address register must not be in flight either
        for r in regs(dst):
            pending[r] = i + 1
        loads += 1
        if touched:
            bad += 1
            print(f"line {i + 1}: address register in flight: {l}")
        continue
    if op == "s_waitcnt" and "vmcnt" in l:
        n = int(re.search(r"vmcnt\((\d+)\)", l).group(1))
        if n == 0:
            pending.clear()
        continue          # counted waits release the OLDEST loads; this scan keeps them pending (conservative) unless vmcnt(0)
    hit = regs(l) & set(pending)
    if hit and not op.startswith("s_"):
        # a counted wait may already have covered these registers: report with the load's line for a human to judge
        bad += 1
        print(f"line {i + 1}: {l}    <- touches v{sorted(hit)} loaded at line {min(pending[r] for r in hit)} with no vmcnt(0) in between")
print(f"{loads} asm tile loads scanned, {bad} suspicious instructions")
sys.exit(1 if bad else 0)
