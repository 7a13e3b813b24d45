#!/usr/bin/env python3
"""Print per-kernel register / LDS / spill figures from a hipcc -save-temps .s file."""
import re, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", txt, re.S):
    blk = m.group(0)
    g = lambda k: re.search(r"\." + k + r":\s+(\S+)", blk).group(1)
    print(f"{g('name')[:60]:60s} vgpr={g('vgpr_count'):>4s} agpr={g('agpr_count'):>3s} sgpr={g('sgpr_count'):>3s} "
          f"spill={g('vgpr_spill_count'):>3s} scratch={g('private_segment_fixed_size'):>5s} lds={g('group_segment_fixed_size'):>6s}")
