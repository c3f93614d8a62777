#!/usr/bin/env python3
"""Mean per LAUNCH, by (kernel, grid size), of the three PMC passes of tools/s2_pmc_cfg5.sh (any command: nothing is assumed about
steps): HBM bytes read / written (FETCH_SIZE / WRITE_SIZE are KB; gfx950 reports half the bytes of wide coalesced reads --
MI355X_MICROARCH.md, HBM -- so read = 2 * 1024 * FETCH_SIZE, write = 1024 * WRITE_SIZE), matrix-pipe occupancy, clock, LDS conflict
share.   usage: python tools/pmc_by_kernel.py DIR OUT.json   (DIR holds FETCH_SIZE.csv, WRITE_SIZE.csv, MFMA.csv)"""
import csv, json, sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]


def load(path):
    disp = defaultdict(lambda: {"c": defaultdict(float)})
    for r in csv.DictReader(open(path)):
        d = disp[int(r["Dispatch_Id"])]
        d["key"] = (r["Kernel_Name"], int(float(r["Grid_Size"])) if r.get("Grid_Size") else 0)
        d["c"][r["Counter_Name"]] += float(r["Counter_Value"])
        if r.get("Start_Timestamp"):
            d["ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    out = defaultdict(lambda: defaultdict(float))
    for d in disp.values():
        o = out[d["key"]]
        o["n"] += 1
        o["us"] += d.get("ns", 0.0) / 1e3
        for k, v in d["c"].items():
            o[k] += v
    return out


fetch, write, mfma = (load(f"{src}/{n}.csv") for n in ("FETCH_SIZE", "WRITE_SIZE", "MFMA"))
res = []
for key in sorted(mfma, key=lambda k: -mfma[k]["us"]):
    m = mfma[key]
    n = m["n"]
    gui = m["GRBM_GUI_ACTIVE"] / 8.0  # the counter sums the 8 XCDs
    e = {"kernel": key[0], "grid_threads": key[1], "launches": int(n), "us_per_launch": round(m["us"] / n, 1)}
    if gui > 0:
        e["effective_clock_GHz"] = round(gui / (m["us"] * 1e3), 3)
        e["mfma_pipe_occupancy"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * gui), 3)
    if m.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_conflict_share"] = round(m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"], 3)
    if key in fetch and key in write and fetch[key]["n"] and write[key]["n"]:
        rd = 2 * 1024 * fetch[key]["FETCH_SIZE"] / fetch[key]["n"]
        wr = 1024 * write[key]["WRITE_SIZE"] / write[key]["n"]
        e["hbm_read_MB"], e["hbm_write_MB"] = round(rd / 1e6, 1), round(wr / 1e6, 1)
        e["hbm_GBps"] = round((rd + wr) / (m["us"] / n * 1e-6) / 1e9, 0) if m["us"] else None
    res.append(e)
json.dump(res, open(dst, "w"), indent=1)
for e in res[:24]:
    print(e)
