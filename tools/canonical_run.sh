#!/bin/bash
# The canonical measurement pass of a round, all on ONE box: full GPU parity suite, every bench line, and the
# rocprofv3 kernel summary of the default bench command.  Run on the GPU box from the repo root:
#   bash tools/canonical_run.sh            (outputs under gpurun_out/canon/)
set -o pipefail
O=gpurun_out/${1:-canon}
mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee $O/status.txt
tail -2 $O/pytest_gpu.log
: > $O/lines.jsonl
run() { echo "# bench.py $*" >> $O/lines.jsonl; timeout -k 10 400 python bench.py "$@" 2> $O/err.log | grep '^{' >> $O/lines.jsonl; echo "bench $* rc=$?" | tee -a $O/status.txt; }
run
run --convs bf16
run --convs torch --no-probe --no-cpu-baseline
run --model 1 --no-probe --no-cpu-baseline
run --workload chain --steps 3 --warmup 1
run --workload labelprop --steps 3 --warmup 1
run --workload shared --steps 5 --warmup 2
run --workload dense --steps 3 --warmup 1
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o run -- python3 /root/repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probe > /root/repo/$O/prof_bench.log 2>&1; find /tmp/prof -name "*kernel_stats.csv" -exec cp {} /root/repo/$O/kernel_stats.csv \; )
echo "profile done" | tee -a $O/status.txt
