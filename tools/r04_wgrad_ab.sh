#!/bin/bash
# A/B on one box: the streamed weight gradient with its dY rows staged TWO k-steps ahead (this tree) against the previous build
# (tools/ubench/libcrw_base.so = the parent commit, one k-step ahead; CRW_HIP_LIB selects it): weight-gradient parity tests, then the
# default bench line both ways, alternating, with the per-kernel in-step event times.  GPU box, repo root.
O=$PWD/gpurun_out/r04w; mkdir -p $O; : > $O/wgrad_ab.log
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "encoder_conv_kernels or full_size_by_replication or wgrad or full_model or trajectory or baseline_shape" 2>&1 | tail -2 | tee -a $O/wgrad_ab.log
line() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); ks={k['kernel'].split(' cin=')[1]:round(k['launch_us'],1) for k in d.get('roofline_kernels',[]) if 'wgrad' in k['kernel']}
        print('$1', 'ms/step %.3f (min %.3f)' % (d['ms_per_step'], d['repeats']['ms_per_step_min']), ks)
"; }
for rep in 1 2 3; do
  CRW_HIP_LIB=$PWD/tools/ubench/libcrw_base.so timeout -k 10 300 python bench.py --no-probe --no-cpu-baseline --repeats 7 2>/dev/null | line "base (1 ahead)" | tee -a $O/wgrad_ab.log
  timeout -k 10 300 python bench.py --no-probe --no-cpu-baseline --repeats 7 2>/dev/null | line "this (2 ahead)" | tee -a $O/wgrad_ab.log
done
