#!/usr/bin/env python3
"""Where a step's wall time goes, from a rocprofv3 --kernel-trace CSV (*_kernel_trace.csv): per step the GPU-busy union of all
kernels (streams overlap), the idle gaps between kernels, and which kernels the largest gaps follow.
usage: python tools/timeline.py kernel_trace.csv STEPS [first_kernel_substring]
The trace is cut into steps at every launch of the kernel whose name contains `first_kernel_substring` (default: the first kernel
of the last STEPS repetitions is found by periodicity of the launch sequence)."""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
marker = sys.argv[3] if len(sys.argv) > 3 else None
ev = []
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    name = re.sub(r"\(.*", "", name).replace("void ", "")
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Stream_Id", r.get("Queue_Id", "0"))))
ev.sort()
if marker:
    cuts = [i for i, e in enumerate(ev) if marker in e[2]]
    cuts = cuts[-steps - 1:] if len(cuts) > steps else cuts
else:
    n = len(ev) // (steps + 2)
    cuts = [len(ev) - n * k for k in range(steps, -1, -1)]
tot_span = tot_busy = tot_sum = 0.0
gap_after = defaultdict(float)
nsteps = 0
for a, b in zip(cuts[:-1], cuts[1:]):
    seg = ev[a:b]
    if not seg:
        continue
    nsteps += 1
    t0, t1 = seg[0][0], max(e[1] for e in seg)
    tot_span += ev[b][0] - t0 if b < len(ev) else t1 - t0
    tot_sum += sum(e[1] - e[0] for e in seg)
    cur_end, last = seg[0][1], seg[0][2]
    busy = seg[0][1] - seg[0][0]
    for s, e, nm, _ in seg[1:]:
        if s > cur_end:
            gap_after[last] += s - cur_end
            busy += e - s
            cur_end, last = e, nm
        elif e > cur_end:
            busy += e - cur_end
            cur_end, last = e, nm
    tot_busy += busy
print(f"{nsteps} steps: span {tot_span / nsteps / 1e3:.1f} us/step, GPU busy (union over streams) {tot_busy / nsteps / 1e3:.1f}, "
      f"sum of kernel durations {tot_sum / nsteps / 1e3:.1f}, idle {(tot_span - tot_busy) / nsteps / 1e3:.1f}")
for nm, g in sorted(gap_after.items(), key=lambda kv: -kv[1])[:25]:
    print(f"   idle after {nm[:70]:70s} {g / nsteps / 1e3:8.1f} us/step")
