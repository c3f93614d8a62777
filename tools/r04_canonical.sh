#!/bin/bash
# Round-4 canonical measurement pass, all on ONE box: full GPU parity suite, the bench lines, rocprofv3 kernel summaries of the
# default command and of the Resnet command (+ its kernel trace for tools/timeline.py).  GPU box, repo root:
#   bash tools/r04_canonical.sh [outdir] [notest]      (outputs under gpurun_out/r04canon/)
set -o pipefail
O=gpurun_out/${1:-r04canon}
mkdir -p $O
R=$PWD
if [ "$2" != "notest" ]; then
  timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee $O/status.txt
  tail -2 $O/pytest_gpu.log
fi
: > $O/lines.jsonl
run() { echo "# bench.py $*" >> $O/lines.jsonl; timeout -k 10 400 python bench.py "$@" 2> $O/err.log | grep '^{' >> $O/lines.jsonl; echo "bench $* rc=$?" | tee -a $O/status.txt; }
run
run --model 1
run --model 1 --convs torch --no-probe --no-cpu-baseline --repeats 3
run --workload labelprop --steps 3 --warmup 1
run --workload labelprop --model 1 --steps 3 --warmup 1
run --workload labelprop --train-steps 300 --steps 3 --warmup 1
run --workload train32 --steps 5 --warmup 2 --repeats 3
run --workload chain --steps 3 --warmup 1
run --workload shared --steps 5 --warmup 2
run --workload dense --steps 3 --warmup 1 --repeats 3
prof() { # name, bench args...
  local name=$1; shift
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_$name && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o run -- python3 $R/bench.py "$@" > $R/$O/prof_$name.log 2>&1; find /tmp/prof_$name -name "*kernel_stats.csv" -exec cp {} $R/$O/${name}_kernel_stats.csv \; ; find /tmp/prof_$name -name "*kernel_trace.csv" -exec cp {} $R/$O/${name}_kernel_trace.csv \; )
  echo "profile $name done" | tee -a $O/status.txt
}
prof default --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --no-probe
prof resnet --model 1 --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --no-probe
python tools/timeline.py $O/resnet_kernel_trace.csv 20 rn_pack_all_kernel > $O/resnet_timeline.txt 2>&1; head -30 $O/resnet_timeline.txt
python tools/timeline.py $O/default_kernel_trace.csv 20 front_fwd_kernel > $O/default_timeline.txt 2>&1; head -12 $O/default_timeline.txt
rm -f $O/*_kernel_trace.csv   # (tens of MB; the timelines above are what is kept)
export CRW_RN_STREAMS=0
prof resnet_serial --model 1 --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --no-probe
rm -f $O/*_kernel_trace.csv
unset CRW_RN_STREAMS
prof labelprop --workload labelprop --steps 3 --warmup 1 --no-events
prof labelprop_resnet --workload labelprop --model 1 --steps 3 --warmup 1 --no-events
rm -f $O/*_kernel_trace.csv
