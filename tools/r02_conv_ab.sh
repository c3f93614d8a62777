#!/bin/bash
# A/B of two builds of the conv kernels at the conv3 / conv4 / conv5 layer shapes: tools/ubench/libcrw_base.so (a build of the
# previous commit) vs the production library, then the default bench line.  GPU box, repo root.
O=$PWD/gpurun_out/r02c; mkdir -p $O; : > $O/ab.log
for rep in 1 2; do
  for lib in tools/ubench/libcrw_base.so radar-sounder-crw_amd/libcrw_hip.so; do
    for shape in "32 64" "64 128" "128 128"; do
      CRW_HIP_LIB=$PWD/$lib timeout -k 10 120 python tools/probe_conv.py $shape 3 16128 10 2>&1 | grep -v "wgrad\|amdgpu.ids" | sed "s|^|$(basename $lib) |" | tee -a $O/ab.log
    done
  done
done
for lib in tools/ubench/libcrw_base.so radar-sounder-crw_amd/libcrw_hip.so tools/ubench/libcrw_base.so radar-sounder-crw_amd/libcrw_hip.so; do
  CRW_HIP_LIB=$PWD/$lib timeout -k 10 300 python bench.py --no-cpu-baseline --no-probe 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$lib', round(d['ms_per_step'],3), round(d['value']))
for k in d['roofline_kernels']: print('   %-62s %8.1f us' % (k['kernel'][:62], k['launch_us']))" | tee -a $O/ab.log
done
