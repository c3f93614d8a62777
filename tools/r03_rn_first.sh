#!/bin/bash
# round 3: first run of the hand-written Resnet path: tests, step time, per-kernel table
set -o pipefail
O=gpurun_out/r03rn; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_resnet_hip.py -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout -k 10 300 python bench.py --model 1 --no-events --no-probe --no-cpu-baseline > $O/bench_m1.log 2>&1; grep '^{' $O/bench_m1.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('model1 ms/step', d['ms_per_step'], 'cols/s', d['value'], 'loss', d['config']['loss'])"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o rn -- python3 $GRAFT_REPO_ROOT/bench.py --model 1 --no-events --no-probe --no-cpu-baseline --steps 10 --warmup 3 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); echo "stats: $f"; [ -n "$f" ] && head -45 "$f" | cut -c1-200
