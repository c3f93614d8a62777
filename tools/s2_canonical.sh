#!/bin/bash
# Second-session canonical pass on ONE box: full GPU parity suite, the bench lines that moved (default, Resnet, cfg5 both encoders),
# rocprofv3 kernel summary + one-step listing of cfg5 with the Resnet encoder.   bash tools/s2_canonical.sh   (outputs: gpurun_out/s2canon/)
set -o pipefail
O=gpurun_out/s2canon
mkdir -p $O
R=$PWD
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee $O/status.txt
tail -2 $O/pytest_gpu.log
: > $O/lines.jsonl
run() { echo "# bench.py $*" >> $O/lines.jsonl; timeout -k 10 400 python bench.py "$@" 2> $O/err.log | grep '^{' >> $O/lines.jsonl; echo "bench $* rc=$?" | tee -a $O/status.txt; }
run
run --model 1
run --workload labelprop --steps 10 --warmup 3
run --workload labelprop --model 1 --steps 10 --warmup 3
run --workload labelprop --train-steps 300 --steps 5 --warmup 2
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_lp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_lp -o run -- python3 $R/bench.py --workload labelprop --model 1 --steps 5 --warmup 2 --no-events > $R/$O/prof_lp.log 2>&1
 find /tmp/prof_lp -name "*kernel_stats.csv" -exec cp {} $R/$O/labelprop_resnet_kernel_stats.csv \;
 find /tmp/prof_lp -name "*kernel_trace.csv" -exec cp {} $R/$O/lp_trace.csv \; )
python tools/step_listing.py $O/lp_trace.csv rn_stem_moments_kernel 3 > $O/lp_step_listing.txt 2>&1
rm -f $O/lp_trace.csv
echo "profile done" | tee -a $O/status.txt
