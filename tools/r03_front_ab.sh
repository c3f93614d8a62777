#!/bin/bash
# A/B of the forward front-end kernels on 512-thread workgroups (default, two per CU) against 1024-thread ones (CRW_FRONT_NT=1024):
# parity tests of the encoder paths, the default bench line and the cfg5 label-propagation line under both.  GPU box, repo root.
set -o pipefail
O=gpurun_out/r03front; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "front or enc or cnn or golden or map or propagate" > $O/tests.log 2>&1; tail -1 $O/tests.log
line() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); ks={k['kernel'][:40]:k['launch_us'] for k in d.get('roofline_kernels',[]) if 'front' in k['kernel']}
        print('$1', 'ms/step %.3f' % d['ms_per_step'], ks)
"; }
for rep in 1 2; do
  for nt in 512 1024; do
    CRW_FRONT_NT=$nt timeout -k 10 300 python bench.py --no-probe --no-cpu-baseline 2>/dev/null | line "default NT=$nt"
  done
done
for nt in 512 1024; do
  CRW_FRONT_NT=$nt timeout -k 10 300 python bench.py --workload labelprop --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | line "labelprop NT=$nt"
done
