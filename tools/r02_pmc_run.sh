#!/bin/bash
# In-step PMC passes over the default bench command (separate passes: FETCH_SIZE and WRITE_SIZE cannot share one; the matrix-pipe
# pass adds the LDS conflict counters) + the rocprofv3 kernel summary of the same command.  GPU box, repo root.
# -> gpurun_out/r02pmc/{FETCH_SIZE,WRITE_SIZE,MFMA}.json (per-kernel means, tools/pmc_in_step.py), kernel_stats.csv
O=$PWD/gpurun_out/r02pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
pass() {  # name, counters...
  local name=$1; shift
  rm -rf /tmp/pmc_$name
  timeout -k 10 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$name -o p -- python3 /root/repo/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-probe --no-events > $O/$name.log 2>&1 || { echo "$name pass failed"; tail -3 $O/$name.log; return 1; }
  f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1)
  python3 /root/repo/tools/pmc_in_step.py $O/$name.json 3 $f
}
pass FETCH_SIZE FETCH_SIZE && pass WRITE_SIZE WRITE_SIZE && pass MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
rm -rf /tmp/prof
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o run -- python3 /root/repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probe > $O/prof_bench.log 2>&1
find /tmp/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
grep '^{' $O/prof_bench.log > $O/prof_bench_line.json
echo "pmc done"
