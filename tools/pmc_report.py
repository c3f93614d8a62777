#!/usr/bin/env python3
"""Per-kernel HBM traffic and matrix-pipe occupancy INSIDE the training step from the three PMC passes of tools/r02_pmc_run.sh.
usage: python tools/pmc_report.py gpurun_out/r03pmc_cnn profiles/r03   -> profiles/r03_pmc_in_step_traffic.json, profiles/r03_pmc_in_step.json
(round 2: gpurun_out/r02pmc profiles/r02; passes: tools/r03_pmc_cnn.sh / tools/r02_pmc_run.sh)
FETCH_SIZE / WRITE_SIZE are KB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM):
hbm_read_bytes = 2 * 1024 * FETCH_SIZE, hbm_write_bytes = 1024 * WRITE_SIZE.  Algorithmic bytes = the planes a kernel must read /
write once at the default workload (P = 16128 patches of 100 pixels; bf16 hi + lo = 4 B per element, hi only = 2 B)."""
import json, re, sys

src, dst = sys.argv[1], sys.argv[2]
P = 16128
E = P * 100  # plane elements per channel
fetch, write, mfma = (json.load(open(f"{src}/{n}.json")) for n in ("FETCH_SIZE", "WRITE_SIZE", "MFMA"))


def algorithmic(name):
    """-> (kind, cin, cout, read bytes, write bytes) of a hand-written conv kernel, else None."""
    m = re.match(r"conv3x3_kernel<3, (\d+), (\d+), (\d), \d, false(?:, \d+)?>", name)
    if m:
        a, b, mode = int(m.group(1)), int(m.group(2)), int(m.group(3))
        if mode == 0:  # forward cin=a -> cout=b; conv5 keeps only the hi plane of its output
            return "fwd", a, b, E * a * 4, E * b * (2 if (a, b) == (128, 128) else 4)
        # backward-data: input gradient planes with a channels -> b channels (+ the ReLU mask plane of the layer below)
        rd = E * a * (2 if (a, b) == (128, 128) else 4) + (E * b * 2 if b != 32 else 0)  # conv5: dY rebuilt from y5 hi + dgap
        return "bwd", b, a, rd, E * b * 4
    m = re.match(r"conv3x3_wgrad2_kernel<3, (\d+), (\d+), 0>", name) or re.match(r"conv3x3_wgrad_kernel<3, (\d+), (\d+), \d+, \d, false>", name)
    if m:
        ci, co = int(m.group(1)), int(m.group(2))
        dy = E * co * (2 if (ci, co) == (128, 128) else 4)  # conv5: the activation hi plane stands for dY
        nslice = {(128, 128): 128, (64, 128): 256, (32, 64): 256}[(ci, co)]
        return "wgrad", ci, co, dy + E * ci * 4, nslice * co * ci * 9 * 4
    if name.startswith("front_fwd_kernel"):  # training forward also writes the 9968-byte saved record per patch
        return "front_fwd", 1, 32, P * 256 * 4, E * 32 * 4 + P * 9968
    if name.startswith("front_bwd_saved_kernel"):  # patch + saved record + the gradient planes in; per-WG partial weight gradients out
        return "front_bwd", 1, 32, P * 256 * 4 + P * 9968 + E * 32 * 4, 256 * 6840 * 4
    if name.startswith("front_bwd_kernel"):
        return "front_bwd", 1, 32, P * 256 * 4 + E * 32 * 4, 256 * 6840 * 4
    return None


traffic, util = {}, {}
for name in sorted(fetch):
    alg = algorithmic(name)
    if alg is None or "FETCH_SIZE" not in fetch[name] or name not in write:
        continue
    kind, ci, co, ar, aw = alg
    rd, wr = 2 * 1024 * fetch[name]["FETCH_SIZE"], 1024 * write[name]["WRITE_SIZE"]
    traffic[name] = {"kind": kind, "layer_cin": ci, "layer_cout": co, "FETCH_SIZE_KB": fetch[name]["FETCH_SIZE"],
                     "WRITE_SIZE_KB": write[name]["WRITE_SIZE"], "hbm_read_bytes": rd, "hbm_write_bytes": wr, "hbm_bytes": rd + wr,
                     "algorithmic_read_bytes": ar, "read_over_algorithmic": round(rd / ar, 3),
                     "algorithmic_write_bytes": aw, "write_over_algorithmic": round(wr / aw, 3)}
for name, c in sorted(mfma.items()):
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in c or "duration_ns" not in c or algorithmic(name) is None:
        continue
    gui = c["GRBM_GUI_ACTIVE"] / 8.0  # the counter sums the 8 XCDs
    u = {"duration_us": round(c["duration_ns"] / 1e3, 1), "effective_clock_GHz": round(gui / c["duration_ns"], 3),
         "mfma_pipe_occupancy": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * gui), 3)}
    if c.get("SQ_LDS_IDX_ACTIVE"):
        u["lds_conflict_share"] = round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 3)
        u["lds_busy_share"] = round(c["SQ_LDS_IDX_ACTIVE"] / (256 * gui), 3) if False else None
    util[name] = {k: v for k, v in u.items() if v is not None}
how_t = ("tools/r02_pmc_run.sh: rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --kernel-trace -- python3 bench.py --steps 6 "
         "--warmup 2 --no-cpu-baseline --no-probe --no-events ; the kernels INSIDE the training step, mean over the dispatches after the "
         "first 3 of each kernel; " + __doc__.split("FETCH_SIZE / WRITE_SIZE are KB;")[1].strip())
how_u = ("tools/r02_pmc_run.sh: rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -- "
         "python3 bench.py --steps 6 --warmup 2 ... : occupancy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE/8); clock = "
         "GRBM_GUI_ACTIVE/8 / duration; lds_conflict_share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE")
json.dump({"_how": how_t, "kernels": traffic}, open(f"{dst}_pmc_in_step_traffic.json", "w"), indent=1)
json.dump({"_how": how_u, "kernels": util}, open(f"{dst}_pmc_in_step.json", "w"), indent=1)
for n in traffic:
    t, u = traffic[n], util.get(n, {})
    print(f"{n:48s} {t['kind']:9s} read x{t['read_over_algorithmic']:<6} write x{t['write_over_algorithmic']:<6} "
          f"{u.get('duration_us', 0):8.1f} us  pipe {u.get('mfma_pipe_occupancy', 0):.2f} @ {u.get('effective_clock_GHz', 0):.2f} GHz  "
          f"lds conflicts {u.get('lds_conflict_share', 0)}")
