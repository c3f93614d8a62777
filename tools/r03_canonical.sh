#!/bin/bash
# Round-3 canonical measurement pass, all on ONE box: full GPU parity suite, the bench lines, rocprofv3 kernel summaries of the
# default command and of the Resnet command.  GPU box, repo root: bash tools/r03_canonical.sh   (outputs under gpurun_out/r03canon/)
set -o pipefail
O=gpurun_out/${1:-r03canon}
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee $O/status.txt
tail -2 $O/pytest_gpu.log
: > $O/lines.jsonl
run() { echo "# bench.py $*" >> $O/lines.jsonl; timeout -k 10 500 python bench.py "$@" 2> $O/err.log | grep '^{' >> $O/lines.jsonl; echo "bench $* rc=$?" | tee -a $O/status.txt; }
run
run --model 1
run --convs bf16 --no-cpu-baseline
run --model 1 --convs torch --no-probe --no-cpu-baseline
run --workload labelprop --steps 3 --warmup 1
run --workload train32 --steps 5 --warmup 2 --repeats 3
run --workload chain --steps 3 --warmup 1
run --workload shared --steps 5 --warmup 2
run --workload dense --steps 3 --warmup 1
prof() { # name, bench args...
  local name=$1; shift
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_$name && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o run -- python3 /root/repo/bench.py "$@" > /root/repo/$O/prof_$name.log 2>&1; find /tmp/prof_$name -name "*kernel_stats.csv" -exec cp {} /root/repo/$O/${name}_kernel_stats.csv \; )
  echo "profile $name done" | tee -a $O/status.txt
}
prof default --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --no-probe
prof resnet --model 1 --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --no-probe
# the same step with the side stream off: per-kernel durations of kernels running ALONE (with it on they overlap and do not add up)
export CRW_RN_STREAMS=0
prof resnet_serial --model 1 --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --no-probe
unset CRW_RN_STREAMS
prof labelprop --workload labelprop --steps 3 --warmup 1 --no-events
