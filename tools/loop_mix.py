#!/usr/bin/env python3
"""Instruction mix of the largest loop of one kernel in a hipcc -S listing (development aid).
Usage: loop_mix.py <file.s> <mangled-name-prefix>"""
import re, sys
from collections import Counter
lines = open(sys.argv[1]).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith(sys.argv[2]) and ':' in l)
end = next(i for i in range(start, len(lines)) if '.end_amdhsa_kernel' in lines[i] or lines[i].startswith('.Lfunc_end'))
body = lines[start:end]
labels = {l.strip()[:-1]: i for i, l in enumerate(body) if re.match(r'^\.LBB\d+_\d+:', l.strip())}
loops = []
for i, l in enumerate(body):
    mm = re.search(r's_c?branch\w* (\.LBB\d+_\d+)', l)
    if mm and mm.group(1) in labels and labels[mm.group(1)] < i: loops.append((labels[mm.group(1)], i))
loops.sort(key=lambda s: s[0] - s[1])
for lo, hi in loops[:2]:
    loop = body[lo:hi + 1]
    c = Counter(); vc = Counter()
    for l in loop:
        t = l.strip().split()
        if not t or t[0][0] in '.;': continue
        op = t[0]
        if op.startswith('v_mfma'): c['mfma'] += 1
        elif op.startswith('v_'): c['valu'] += 1; vc[op] += 1
        elif op.startswith('s_waitcnt'): c['waitcnt'] += 1
        elif op.startswith('s_barrier'): c['barrier'] += 1
        elif op.startswith('s_'): c['salu'] += 1
        elif op.startswith('ds_'): c['ds'] += 1
        elif op.startswith(('global_', 'buffer_', 'scratch_')): c[op.split('_')[0]] += 1
        else: c[op] += 1
    print(f"loop of {len(loop)} lines:", dict(c))
    print("  ", vc.most_common(14))
