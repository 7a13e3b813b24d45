#!/usr/bin/env python3
"""Which VGPRs does a kernel's hot loop only READ (loop invariants held in registers) and which does it carry?

    tools/loop_live.py file.s first_line last_line      (line numbers of the loop body inside file.s)

Reads AMDGPU asm as printed by hipcc -save-temps: the first operand of an instruction is taken as its destination
(stores, ds_write, scratch_store, s_waitcnt and the like have none)."""
import re, sys
lines = open(sys.argv[1]).read().split("\n")[int(sys.argv[2]) - 1:int(sys.argv[3])]
NODST = ("ds_write", "global_store", "scratch_store", "buffer_store", "s_", "v_cmp", ";", ".")
def regs(tok):
    out = []
    for m in re.finditer(r"v\[(\d+):(\d+)\]|v(\d+)", tok):
        if m.group(3) is not None: out.append(int(m.group(3)))
        else: out.extend(range(int(m.group(1)), int(m.group(2)) + 1))
    return out
read_first, written, reads, writes = set(), set(), set(), set()
for ln in lines:
    ln = ln.split(";")[0].strip()
    if not ln or ln.startswith(".") or ln.endswith(":"): continue
    parts = ln.split(None, 1)
    if len(parts) < 2: continue
    op, args = parts
    toks = [t.strip() for t in args.split(",")]
    has_dst = not op.startswith(NODST) or op.startswith("v_cmpx")
    if op.startswith("v_cmp") and toks and toks[0].startswith("v"): has_dst = False
    dst = regs(toks[0]) if has_dst else []
    src = [r for t in (toks[1:] if has_dst else toks) for r in regs(t)]
    if "ds_read_b64_tr_b16" in op or op.startswith("ds_read") or op.startswith("global_load") or op.startswith("scratch_load"):
        dst, src = regs(toks[0]), [r for t in toks[1:] for r in regs(t)]
    for r in src:
        reads.add(r)
        if r not in written: read_first.add(r)
    for r in dst:
        written.add(r); writes.add(r)
inv = sorted(read_first - writes)
carried = sorted(read_first & writes)
print(f"invariant (read, never written in the loop): {len(inv)}: {inv}")
print(f"loop-carried (read before written, written later): {len(carried)}: {carried}")
print(f"temporaries: {len(writes - read_first)}")
