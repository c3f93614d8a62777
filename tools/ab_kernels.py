"""Print the in-step per-kernel times (HIP events of the timed steps) of bench.py JSON lines given as files."""
import json, sys
cols = []
for f in sys.argv[1:]:
    j = json.loads([l for l in open(f) if l.startswith("{")][-1])
    cols.append((f, j["ms_per_step"], {k["kernel"]: k["launch_us"] for k in j.get("roofline_kernels", [])}))
names = list(cols[0][2])
print("%-62s" % "kernel", *["%12s" % c[0].split("/")[-1][:12] for c in cols])
for n in names:
    print("%-62s" % n, *["%12.1f" % c[2].get(n, float("nan")) for c in cols])
print("%-62s" % "ms_per_step", *["%12.3f" % c[1] for c in cols])
