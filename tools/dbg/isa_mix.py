#!/usr/bin/env python3
"""Instruction mix of the kernels in a hipcc -S device listing whose mangled name contains a pattern.
usage: isa_mix.py file.s pattern [pattern...]"""
import sys
from collections import Counter

lines = open(sys.argv[1]).read().split('\n')
pats = sys.argv[2:]
i = 0
while i < len(lines):
    l = lines[i]
    if l.startswith('_Z') and ':' in l and any(p in l.split(':')[0] for p in pats):
        c = Counter()
        j = i + 1
        while j < len(lines) and 's_endpgm' not in lines[j]:
            t = lines[j].strip()
            if t and not t.startswith((';', '.')) and not t.endswith(':'):
                op = t.split()[0]
                if op.startswith('v_'):
                    c['VALU'] += 1
                elif op.startswith('s_'):
                    c['SALU'] += 1
                else:
                    c[op] += 1
            j += 1
        print(l[:70], dict(c))
        i = j
    i += 1
