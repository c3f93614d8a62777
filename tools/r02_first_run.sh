#!/bin/bash
# round-2 first pass: GPU parity suite, then rocprofv3 kernel summaries of the walk at the K and D shapes
set -o pipefail
O=$PWD/gpurun_out/r02a; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee $O/status.txt
tail -5 $O/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp
for m in bf16 bf16x3 f32; do
  rm -rf /tmp/prof_$m
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$m -o run -- python3 /root/repo/bench.py --workload chain --modes $m --steps 2 --warmup 1 > $O/chain_$m.log 2>&1
  echo "chain $m rc=$?" | tee -a $O/status.txt
  find /tmp/prof_$m -name "*kernel_stats.csv" -exec cp {} $O/chain_${m}_kernel_stats.csv \;
done
rm -rf /tmp/prof_dense
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dense -o run -- python3 /root/repo/bench.py --workload dense --steps 3 --warmup 1 --no-cpu-baseline --no-probe > $O/dense.log 2>&1
echo "dense rc=$?" | tee -a $O/status.txt
find /tmp/prof_dense -name "*kernel_stats.csv" -exec cp {} $O/dense_kernel_stats.csv \;
