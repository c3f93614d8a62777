#!/bin/bash
# A/B: production library vs tools/ubench/libcrw_exp.so (-DCRW_CONV_OCC6: 6 waves per SIMD for the 32 <-> 64 channel conv kernels)
O=$PWD/gpurun_out/r02c; mkdir -p $O; : > $O/occ.log
timeout -k 10 300 env CRW_HIP_LIB=$PWD/tools/ubench/libcrw_exp.so python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "encoder_conv or full_size or full_model" 2>&1 | tail -2 | tee -a $O/occ.log
for rep in 1 2; do
  for lib in radar-sounder-crw_amd/libcrw_hip.so tools/ubench/libcrw_exp.so; do
    CRW_HIP_LIB=$PWD/$lib timeout -k 10 120 python tools/probe_conv.py 32 64 3 16128 10 2>&1 | grep -v "wgrad\|amdgpu.ids" | sed "s|^|$(basename $lib) |" | tee -a $O/occ.log
    CRW_CONV_NW=8 CRW_HIP_LIB=$PWD/$lib timeout -k 10 120 python tools/probe_conv.py 32 64 3 16128 10 2>&1 | grep "bwd-data" | sed "s|^|$(basename $lib) NW=8 |" | tee -a $O/occ.log
  done
done
for lib in radar-sounder-crw_amd/libcrw_hip.so tools/ubench/libcrw_exp.so radar-sounder-crw_amd/libcrw_hip.so tools/ubench/libcrw_exp.so; do
  CRW_HIP_LIB=$PWD/$lib timeout -k 10 300 python bench.py --no-cpu-baseline --no-probe 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$lib', round(d['ms_per_step'],3), round(d['value']))
for k in d['roofline_kernels']:
    if 'cin=32' in k['kernel'] and 'wgrad' not in k['kernel']: print('   %-62s %8.1f us' % (k['kernel'][:62], k['launch_us']))" | tee -a $O/occ.log
done
