#!/usr/bin/env python3
"""tools/isa_regs.py FILE.s [SUBSTRING] -- registers, LDS and scratch of every kernel in a hipcc -S listing (.amdhsa metadata)."""
import re
import sys

txt = open(sys.argv[1]).read()
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
    name, body = m.group(1), m.group(2)
    if sub not in name:
        continue
    g = lambda k: (re.search(r"\.amdhsa_" + k + r" (\S+)", body) or [None, "?"])[1]
    print("%-90s vgpr %s (accum offset %s)  sgpr %s  lds %s  scratch %s" % (name[-90:], g("next_free_vgpr"), g("accum_offset"), g("next_free_sgpr"),
                                                                    g("group_segment_fixed_size"), g("private_segment_fixed_size")))
