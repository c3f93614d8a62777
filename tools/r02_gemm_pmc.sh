#!/bin/bash
# Matrix-pipe occupancy, clock and LDS conflicts of the bf16 chain GEMM at n = 4096 (PMC pass over tools/probe_gemm.py).
# usage (GPU box, repo root): bash tools/r02_gemm_pmc.sh [library]   -> gpurun_out/r02g/pmc_<kind>.json
O=$PWD/gpurun_out/r02g; mkdir -p $O
export CRW_HIP_LIB=${1:+$PWD/$1}
[ -z "$CRW_HIP_LIB" ] && unset CRW_HIP_LIB
cd /tmp && export TMPDIR=/tmp
for kind in bf16 bf16x3; do
  rm -rf /tmp/pmc_g
  timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_g -o p -- python3 /root/repo/tools/probe_gemm.py $kind 4096 4 8 0 1 > $O/pmc_$kind.log 2>&1 || { echo "pass failed"; tail -3 $O/pmc_$kind.log; exit 1; }
  f=$(find /tmp/pmc_g -name "*counter_collection.csv" | head -1)
  python3 /root/repo/tools/pmc_in_step.py $O/pmc_$kind.json 3 $f
  python3 - <<PY
import json
d=json.load(open("$O/pmc_$kind.json"))
for k,c in d.items():
    if "gemm_pad_bf16" in k and "duration_ns" in c:
        gui=c["GRBM_GUI_ACTIVE"]/8
        print("$kind", k[:60], "%.1f us"%(c["duration_ns"]/1e3), "clock %.2f GHz"%(gui/c["duration_ns"]), "pipe %.3f"%(c["SQ_VALU_MFMA_BUSY_CYCLES"]/(1024*gui)), "lds conflict share %.3f"%(c["SQ_LDS_BANK_CONFLICT"]/max(c["SQ_LDS_IDX_ACTIVE"],1)), "lds active/cycle/CU %.3f"%(c["SQ_LDS_IDX_ACTIVE"]/(256*gui)))
PY
done
