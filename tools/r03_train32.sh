#!/bin/bash
# round 3: training at BASELINE config 5's geometry (32x32 patches, tiled "map" kernels): bench line + per-kernel table
set -o pipefail
O=gpurun_out/r03t32; mkdir -p $O
timeout -k 10 500 python bench.py --workload train32 --steps 5 --warmup 2 --repeats 3 --cpu-seconds 10 > $O/bench.log 2>&1; grep '^{' $O/bench.log > $O/line.json; python -c "
import json; d=json.load(open('$O/line.json')); print('train32 ms/step %.2f cols/s %.0f' % (d['ms_per_step'], d['value']), d.get('cpu_baseline', {}).get('value'))
for k in d.get('roofline_kernels', [])[:20]: print('   %-70s x%d %8.1f us  alg %.3f exec %.3f' % (k['kernel'][:70], k['launches_per_step'], k['launch_us'], k['frac'], k['mfma_executed_frac']))
"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o t32 -- python3 $GRAFT_REPO_ROOT/bench.py --workload train32 --no-events --no-cpu-baseline --steps 5 --warmup 2 --repeats 1 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/kstats.py $O/prof/t32_kernel_stats.csv 7 > $O/kstats.txt; head -24 $O/kstats.txt
