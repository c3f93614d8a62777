#!/usr/bin/env python3
"""Registers / scratch / occupancy per kernel from `hipcc -Rpass-analysis=kernel-resource-usage` output.
usage: hipcc ... -c x.hip -Rpass-analysis=kernel-resource-usage 2> ru.txt ; python tools/reg_usage.py ru.txt [substring]"""
import re, sys
cur, rows = None, {}
for l in open(sys.argv[1]):
    m = re.search(r'Function Name: (\S+)', l)
    if m:
        cur = m.group(1); rows[cur] = {}
    for k, short in (('VGPRs', 'vgpr'), ('AGPRs', 'agpr'), ('ScratchSize [bytes/lane]', 'scratch'), ('Occupancy [waves/SIMD]', 'occ')):
        m = re.search(r'    ' + re.escape(k) + r': (\d+)', l)
        if m and cur: rows[cur][short] = int(m.group(1))
for n, r in rows.items():
    if len(sys.argv) < 3 or sys.argv[2] in n:
        print(n[:90], r)
