#!/bin/bash
# rocprofv3 kernel summaries of the secondary bench workloads (labelprop, dense, shared).  GPU box, repo root.
O=$PWD/gpurun_out/r02p; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for w in labelprop dense shared; do
  rm -rf /tmp/prof_$w
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$w -o run -- python3 /root/repo/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > $O/$w.log 2>&1
  find /tmp/prof_$w -name "*kernel_stats.csv" -exec cp {} $O/${w}_kernel_stats.csv \;
  echo "$w done"
done
