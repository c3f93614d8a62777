"""per-step kernel time table from a rocprofv3 kernel_stats.csv: python tools/kstats.py stats.csv STEPS"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = 0.0
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
    name = re.sub(r"\(.*", "", name).replace("void ", "")[:56]
    t = int(r["TotalDurationNs"]) / steps / 1e3
    tot += t
    print(f"{name:56s} calls/step {int(r['Calls']) / steps:5.1f}  us/step {t:8.1f}  avg {float(r['AverageNs']) / 1e3:7.1f}  max {int(r['MaxNs']) / 1e3:7.1f}")
print("total us/step %.1f" % tot)
