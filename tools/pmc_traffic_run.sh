#!/bin/bash
# HBM traffic of the kernels INSIDE the training step: two PMC passes (FETCH_SIZE and WRITE_SIZE cannot share one)
# over the default bench command.  Run on the GPU box from the repo root; result: gpurun_out/pmc_step/*.json
O=$PWD/gpurun_out/pmc_step; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout -k 10 170 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- python3 /root/repo/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-probe --no-events > $O/$c.log 2>&1 || { echo "$c pass failed"; tail -3 $O/$c.log; exit 1; }
  f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1)
  ls -la $f
  python3 /root/repo/tools/pmc_in_step.py $O/$c.json 3 $f
done
