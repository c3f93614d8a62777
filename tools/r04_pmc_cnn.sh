#!/bin/bash
# Round-4 in-step PMC passes over the DEFAULT bench command (CNN encoder): FETCH_SIZE and WRITE_SIZE in separate passes (they cannot
# share one), a matrix-pipe pass with the clock and the LDS conflict counters.  GPU box, repo root.
# -> gpurun_out/r04pmc_cnn/{FETCH_SIZE,WRITE_SIZE,MFMA}.csv (raw rocprofv3 counter CSVs trimmed to the columns used; committed under
#    profiles/r04_pmc_raw/cnn_*.csv) and .json (per-kernel means, tools/pmc_in_step.py) -> tools/pmc_report.py -> profiles/r04_pmc_in_step*.json
O=$PWD/gpurun_out/r04pmc_cnn; mkdir -p $O; R=$PWD
cd /tmp && export TMPDIR=/tmp
pass() {  # name, counters...
  local name=$1; shift
  rm -rf /tmp/pmc_$name
  timeout -k 10 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$name -o p -- python3 $R/bench.py --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --no-probe --no-events > $O/$name.log 2>&1 || { echo "$name pass failed"; tail -3 $O/$name.log; return 1; }
  f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1)
  python3 $R/tools/pmc_in_step.py $O/$name.json 3 $f
  python3 - "$f" "$O/$name.csv" <<'PY'
import csv, sys
rd = csv.DictReader(open(sys.argv[1]))
keep = [k for k in ["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"] if k in rd.fieldnames]
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
echo "pmc done"
