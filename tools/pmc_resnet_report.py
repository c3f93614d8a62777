#!/usr/bin/env python3
"""Per-kernel HBM traffic, matrix-pipe occupancy, clock and LDS conflict share INSIDE the Resnet training step, from the three PMC
passes of tools/r03_pmc_resnet.sh (raw CSVs: profiles/r03_pmc_raw/).
usage: python tools/pmc_resnet_report.py profiles/r03_pmc_raw profiles/r03_pmc_resnet.json
Per kernel name: totals PER STEP over its launches (a step launches e.g. rn_conv_kernel<128,32,2,0> 14 times with different
geometries).  FETCH_SIZE / WRITE_SIZE are KB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads
(MI355X_MICROARCH.md, HBM): hbm_read_bytes = 2 * 1024 * FETCH_SIZE, hbm_write_bytes = 1024 * WRITE_SIZE."""
import csv, json, sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
STEPS, WARM = 4, 2  # bench.py --steps 4 --warmup 2 --repeats 1: 6 identical steps, the first WARM are dropped


def per_dispatch(path):
    rows = defaultdict(lambda: {"c": defaultdict(float)})
    for r in csv.DictReader(open(path)):
        d = rows[int(r["Dispatch_Id"])]
        d["name"] = r["Kernel_Name"]
        d["grid"] = int(float(r["Grid_Size"])) if r.get("Grid_Size") else 0
        d["c"][r["Counter_Name"]] += float(r["Counter_Value"])
        if r.get("Start_Timestamp"):
            d["ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return [rows[k] for k in sorted(rows)]


def per_step(path, by_grid=False):
    disp = per_dispatch(path)
    n = len(disp) // (STEPS + WARM)  # dispatches per step (every step launches the same sequence)
    timed = disp[len(disp) - n * STEPS:]
    out = defaultdict(lambda: defaultdict(float))
    for d in timed:
        o = out[(d["name"], d.get("grid", 0)) if by_grid else d["name"]]
        o["launches"] += 1.0 / STEPS
        o["us"] += d.get("ns", 0.0) / 1e3 / STEPS
        for k, v in d["c"].items():
            o[k] += v / STEPS
    return out


fetch, write, mfma = (per_step(f"{src}/{n}.csv") for n in ("FETCH_SIZE", "WRITE_SIZE", "MFMA"))
res = {}
for name in sorted(mfma, key=lambda k: -mfma[k]["us"]):
    m = mfma[name]
    gui = m["GRBM_GUI_ACTIVE"] / 8.0  # the counter sums the 8 XCDs
    e = {"launches_per_step": round(m["launches"], 2), "us_per_step": round(m["us"], 1)}
    if gui > 0:
        e["effective_clock_GHz"] = round(gui / (m["us"] * 1e3), 3)
        e["mfma_pipe_occupancy"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * gui), 3)
    if m.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_conflict_share"] = round(m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"], 3)
    if name in fetch and name in write:
        rd, wr = 2 * 1024 * fetch[name]["FETCH_SIZE"], 1024 * write[name]["WRITE_SIZE"]
        e["hbm_read_MB"], e["hbm_write_MB"] = round(rd / 1e6, 1), round(wr / 1e6, 1)
        e["hbm_GBps"] = round((rd + wr) / (m["us"] * 1e-6) / 1e9, 0) if m["us"] else None
    res[name] = e
# the same per (kernel, grid size in threads): one instance serves several layers; the grid tells them apart (bench.py looks the
# dominant kernel's launch up by it: blocks = patch tiles x column tiles x groups)
fg, wg, mg = (per_step(f"{src}/{n}.csv", True) for n in ("FETCH_SIZE", "WRITE_SIZE", "MFMA"))
by_grid = {}
for (name, grid) in sorted(mg, key=lambda k: -mg[k]["us"]):
    m = mg[(name, grid)]
    if (name, grid) not in fg or (name, grid) not in wg or m["launches"] <= 0:
        continue
    L = m["launches"]
    rd, wr = 2 * 1024 * fg[(name, grid)]["FETCH_SIZE"] / L, 1024 * wg[(name, grid)]["WRITE_SIZE"] / L
    by_grid.setdefault(name, {})[str(grid)] = {"launches_per_step": round(L, 2), "us_per_launch": round(m["us"] / L, 1),
                                              "hbm_read_bytes_per_launch": round(rd), "hbm_write_bytes_per_launch": round(wr)}
tot = {"us_per_step": round(sum(v["us_per_step"] for v in res.values()), 1),
       "hbm_read_MB": round(sum(v.get("hbm_read_MB", 0) for v in res.values()), 1),
       "hbm_write_MB": round(sum(v.get("hbm_write_MB", 0) for v in res.values()), 1)}
json.dump({"_how": __doc__, "step_total": tot, "kernels": res, "by_grid": by_grid}, open(dst, "w"), indent=1)
print(json.dumps(tot))
for k, v in list(res.items())[:24]:
    print(f"{k[:52]:52s}", v)
