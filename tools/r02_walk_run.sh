#!/bin/bash
# GPU parity suite, then the chain workload (shape K) per arithmetic with rocprofv3 kernel summaries, then default / dense bench lines
set -o pipefail
O=$PWD/gpurun_out/${1:-r02c}; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee $O/status.txt
tail -6 $O/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp
for m in bf16 bf16x3 f32; do
  rm -rf /tmp/prof_$m
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$m -o run -- python3 /root/repo/bench.py --workload chain --modes $m --steps 2 --warmup 1 > $O/chain_$m.log 2>&1
  echo "chain $m rc=$?" | tee -a $O/status.txt
  find /tmp/prof_$m -name "*kernel_stats.csv" -exec cp {} $O/chain_${m}_kernel_stats.csv \;
  grep '^{' $O/chain_$m.log | tail -1 >> $O/lines.jsonl
done
cd /root/repo
timeout -k 10 300 python bench.py --workload dense --steps 3 --warmup 1 --no-cpu-baseline --no-probe 2>/dev/null | grep '^{' >> $O/lines.jsonl; echo "dense rc=$?" | tee -a $O/status.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probe 2>/dev/null | grep '^{' >> $O/lines.jsonl; echo "default rc=$?" | tee -a $O/status.txt
