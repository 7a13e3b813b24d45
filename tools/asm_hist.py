#!/usr/bin/env python3
"""Instruction histogram of one kernel of a hipcc -save-temps .s file: asm_hist.py file.s kernel-name-substring [top]."""
import collections, re, sys
lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 50
start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and key in l.split(':')[0] and l.rstrip().split(';')[0].strip().endswith(':'))
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
ops = collections.Counter()
for l in lines[start + 1:end]:
    l = l.strip()
    if not l or l[0] in '.;/' or l.split(';')[0].strip().endswith(':'):
        continue
    ops[l.split()[0]] += 1
for k, v in ops.most_common(top):
    print(f"{v:6d} {k}")
print(sum(ops.values()), "instructions;", sum(v for k, v in ops.items() if k.startswith('v_') and 'mfma' not in k), "VALU;",
      sum(v for k, v in ops.items() if 'mfma' in k), "MFMA;", sum(v for k, v in ops.items() if k.startswith('ds_')), "DS")
