#!/bin/bash
# PMC passes over BASELINE config 5 with the reference's encoder (bench.py --workload labelprop --model 1): FETCH_SIZE and WRITE_SIZE in
# separate passes, a matrix-pipe pass with the clock and the LDS conflict counters (counters with --kernel-trace only).  GPU box, repo root.
# -> gpurun_out/s2pmc/{FETCH_SIZE,WRITE_SIZE,MFMA}.csv + cfg5_pmc.json (tools/pmc_by_kernel.py: mean per launch by kernel and grid)
O=$PWD/gpurun_out/s2pmc; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
pass() {  # name, counters...
  local name=$1; shift
  rm -rf /tmp/pmc_$name
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$name -o p -- python3 $R/bench.py --workload labelprop --model 1 --steps 3 --warmup 1 --no-events > $O/$name.log 2>&1 || { echo "$name pass failed"; tail -3 $O/$name.log; return 1; }
  f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$O/$name.csv" <<'PY'
import csv, sys
rd = csv.DictReader(open(sys.argv[1]))
keep = ["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"]
keep = [k for k in keep if k in rd.fieldnames]
w = csv.DictWriter(open(sys.argv[2], "w"), keep)
w.writeheader()
for r in rd:
    if "crw::" in r["Kernel_Name"]:
        r["Kernel_Name"] = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        w.writerow({k: r[k] for k in keep})
PY
  echo "$name: $(wc -l < $O/$name.csv) rows"
}
pass FETCH_SIZE FETCH_SIZE && pass WRITE_SIZE WRITE_SIZE && pass MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
cd $R && python3 tools/pmc_by_kernel.py gpurun_out/s2pmc gpurun_out/s2pmc/cfg5_pmc.json
echo "pmc done"
