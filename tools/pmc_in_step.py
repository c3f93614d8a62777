"""Condense rocprofv3 --pmc counter_collection CSVs of a bench.py run into per-kernel means.
usage: python tools/pmc_in_step.py OUT.json SKIP file1.csv [file2.csv ...]
Per kernel name and counter: mean over the dispatches of that kernel after the first SKIP (warm-up steps)."""
import csv, json, re, sys
from collections import defaultdict

out, skip, files = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
vals = defaultdict(lambda: defaultdict(list))  # kernel -> counter -> [(dispatch, value)]
dur = defaultdict(dict)
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        name = re.sub(r"^void ", "", name)
        name = re.sub(r"crw::\(anonymous namespace\)::", "", name)
        name = name.split("(")[0]
        vals[name][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        if "Start_Timestamp" in r and "End_Timestamp" in r:  # kernel duration (ns), one value per dispatch
            dur[name][int(r["Dispatch_Id"])] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
res = {}
for k, cs in vals.items():
    res[k] = {}
    for c, lst in cs.items():
        per = defaultdict(float)
        for d, v in lst:  # a counter may be reported in several rows per dispatch (one per instance): sum them
            per[d] += v
        seq = [per[d] for d in sorted(per)][skip:]
        if seq:
            res[k][c] = sum(seq) / len(seq)
            res[k]["dispatches_averaged"] = len(seq)
for k, d in dur.items():
    seq = [d[i] for i in sorted(d)][skip:]
    if seq and k in res:
        res[k]["duration_ns"] = sum(seq) / len(seq)
json.dump(res, open(out, "w"), indent=1)
print("kernels:", len(res))
