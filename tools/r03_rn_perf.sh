#!/bin/bash
# round 3: Resnet path -- parity tests, step time (optionally A/B of env knobs given as arguments "NAME=VAL ..."), per-kernel table
set -o pipefail
O=gpurun_out/r03rn; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_resnet_hip.py -q -m gpu > $O/tests.log 2>&1; tail -1 $O/tests.log
if [ -n "$TESTENV" ]; then  # the parity tests once more under an env knob, e.g. TESTENV=CRW_RN_FUSE_RED=1
  timeout -k 10 600 env $TESTENV python -m pytest tests/test_resnet_hip.py -q -m gpu -k "resnet_hip or resnet_native" > $O/tests_env.log 2>&1; echo "$TESTENV: $(tail -1 $O/tests_env.log)"
fi
run() { timeout -k 10 300 env "$@" python bench.py --model 1 --no-probe --no-cpu-baseline > $O/bench_m1.log 2>&1; grep '^{' $O/bench_m1.log | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$*', 'ms/step %.3f' % d['ms_per_step'], 'loss', d['config']['loss'])
for k in d.get('roofline_kernels', [])[:40]: print('   %-72s x%d %7.1f us  %s %.3f (mfma %.3f exec %.3f hbm %.3f)' % (k['kernel'][:72], k['launches_per_step'], k['launch_us'], k['bound'], k['frac'], k['mfma_frac'], k['mfma_executed_frac'], k['hbm_frac']))
"; }
run CRW_DEFAULT=1
for kv in "$@"; do run $kv; done
if [ -z "$NOPROF" ]; then
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o rn -- python3 $GRAFT_REPO_ROOT/bench.py --model 1 --no-events --no-probe --no-cpu-baseline --steps 10 --warmup 3 --repeats 1 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/kstats.py $O/prof/rn_kernel_stats.csv 13 | head -40
fi
