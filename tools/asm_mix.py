#!/usr/bin/env python3
"""Static instruction mix of one kernel in a hipcc .s file (fully unrolled kernels: static = per pass).
usage: tools/asm_mix.py file.s <kernel-name-regex> [top]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
pat = sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
for m in re.finditer(r'^(\S*' + pat + r'\S*):[^\n]*\n(.*?)\n\s*s_endpgm', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    cnt = collections.Counter()
    for line in body.split('\n'):
        line = line.strip()
        if not line or line[0] in ';.' or line.endswith(':'):
            continue
        cnt[line.split()[0]] += 1
    tot = sum(cnt.values())
    nm = cnt.get('v_mfma_f32_32x32x16_f16', 0) + cnt.get('v_mfma_f32_32x32x16_bf16', 0)
    print(f"{name}: {tot} instructions, {nm} MFMAs ({tot / max(nm, 1):.2f} per MFMA)")
    meta = re.search(r'\.agpr_count:\s+(\d+)(?:(?!\.agpr_count).)*?\.name:\s+' + re.escape(name) + r'\n(?:(?!\.agpr_count).)*?\.private_segment_fixed_size:\s+(\d+)(?:(?!\.agpr_count).)*?\.vgpr_count:\s+(\d+)', s, re.S)
    if meta:
        print(f"  agprs {meta.group(1)}, scratch {meta.group(2)} B, vgprs(total) {meta.group(3)}")
    for k, v in cnt.most_common(top):
        print(f"  {k:30s} {v}")
