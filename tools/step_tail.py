#!/usr/bin/env python3
"""The end of a training step from a rocprofv3 --kernel-trace CSV: for one step in the middle of the trace, every kernel of the last
`window` microseconds with its stream, start and end offsets -- which stream finishes last, what the other one is doing meanwhile.
usage: python tools/step_tail.py kernel_trace.csv first_kernel_substring [window_us] [whole]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2]
window = float(sys.argv[3]) if len(sys.argv) > 3 else 1200.0
ev = []
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    name = re.sub(r"\(.*", "", name).replace("void ", "")
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Stream_Id", r.get("Queue_Id", "0"))))
ev.sort()
cuts = [i for i, e in enumerate(ev) if marker in e[2]]
k = len(cuts) // 2
seg = ev[cuts[k]:cuts[k + 1]]
t0 = seg[0][0]
t1 = max(e[1] for e in seg)
print(f"step {k}: {len(seg)} kernels, {(t1 - t0) / 1e3:.1f} us from first start to last end; next step starts {(ev[cuts[k + 1]][0] - t1) / 1e3:.1f} us later")
streams = sorted({e[3] for e in seg})
for s in streams:
    es = [e for e in seg if e[3] == s]
    print(f"  stream {s}: {len(es)} kernels, busy {sum(e[1] - e[0] for e in es) / 1e3:.1f} us, last end at {(max(e[1] for e in es) - t0) / 1e3:.1f} us")
lo = t0 if len(sys.argv) > 4 else t1 - window * 1e3
for s, e, nm, st in seg:
    if e >= lo:
        print(f"  {(s - t0) / 1e3:9.1f} -> {(e - t0) / 1e3:9.1f}  ({(e - s) / 1e3:7.1f} us)  stream {st:>3s}  {nm[:100]}")
