#!/usr/bin/env python3
"""One step of a rocprofv3 --kernel-trace CSV as a listing: start offset, duration and the idle gap in front of every kernel.
usage: python tools/step_listing.py kernel_trace.csv MARKER_SUBSTRING [which]   (the step between the which-th and the next launch of the
marker kernel, counted from the end; default 2 = the last complete step)"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
marker, which = sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 2
ev = []
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    name = re.sub(r"\(.*", "", name).replace("void ", "")
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
ev.sort()
cuts = [i for i, e in enumerate(ev) if marker in e[2]]
a, b = cuts[-which], cuts[-which + 1] if which > 1 else len(ev)
t0, end = ev[a][0], ev[a][0]
print(f"step of {b - a} kernels, span {(ev[b][0] - t0) / 1e3:.1f} us" if b < len(ev) else "last step")
for s, e, n in ev[a:b]:
    gap = (s - end) / 1e3
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {gap:8.1f}  {n[:90]}")
    end = max(end, e)
