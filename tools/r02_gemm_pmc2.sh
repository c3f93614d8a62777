#!/bin/bash
# L2 behaviour of the bf16 chain GEMM at n = 4096: bytes fetched from beyond L2 (FETCH_SIZE, KB; x2 on gfx950 for wide reads)
# and L2 hits / misses.  GPU box, repo root.
O=$PWD/gpurun_out/r02g; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum"; do
  rm -rf /tmp/pmc_g
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_g -o p -- python3 /root/repo/tools/probe_gemm.py bf16 4096 4 8 0 1 > $O/pmc2.log 2>&1 || { echo "pass failed: $c"; tail -3 $O/pmc2.log; continue; }
  f=$(find /tmp/pmc_g -name "*counter_collection.csv" | head -1)
  python3 /root/repo/tools/pmc_in_step.py $O/pmc2.json 3 $f > /dev/null
  python3 - <<PY
import json
d=json.load(open("$O/pmc2.json"))
for k,c in d.items():
    if "gemm_pad_bf16" in k: print(k[:50], {a:round(b,1) for a,b in c.items()})
PY
done
