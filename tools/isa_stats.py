#!/usr/bin/env python3
"""Per-kernel summary of a gfx950 assembly file (hipcc -save-temps): multiply-accumulate counts, registers, scratch, code size."""
import re, sys
s = open(sys.argv[1]).read()
for f in re.split(r'\n\t\.globl\t', s)[1:]:
    name = f.split('\n')[0]
    nv = re.search(r'; NumVgprs: (\d+)', f)
    if not nv:
        continue
    g = lambda pat: (re.search(pat, f) or [None, '?'])[1]
    print('%-70s mad_i64 %5d mad_u64 %5d addc %5d | vgpr %s agpr %s scratch %s occ %s code %s' % (
        name[:70], len(re.findall(r'v_mad_i64_i32', f)), len(re.findall(r'v_mad_u64_u32', f)), len(re.findall(r'v_addc_co_u32', f)),
        nv.group(1), g(r'; NumAgprs: (\d+)'), g(r'; ScratchSize: (\d+)'), g(r'; Occupancy: (\d+)'), g(r'; codeLenInByte = (\d+)')))
