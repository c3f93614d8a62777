#!/bin/bash
# A/B of two builds of the bf16 chain GEMM: tools/ubench/libcrw_base.so (the previous commit) vs the production library:
# the four operand layouts at n = 4096, then the chain workload in bf16 mode.  GPU box, repo root.
O=$PWD/gpurun_out/r02g; mkdir -p $O; : > $O/ab.log
timeout -k 10 300 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "gemm_bf16 or bf16_chain" 2>&1 | tail -3 | tee -a $O/ab.log
for lib in tools/ubench/libcrw_base.so radar-sounder-crw_amd/libcrw_hip.so tools/ubench/libcrw_base.so radar-sounder-crw_amd/libcrw_hip.so; do
  for l in "0 0" "0 1" "1 0" "1 1"; do
    CRW_HIP_LIB=$PWD/$lib timeout -k 10 120 python tools/probe_gemm.py bf16 4096 4 20 $l 2>&1 | grep -v amdgpu.ids | sed "s|^|$(basename $lib) |" | tee -a $O/ab.log
  done
done
for lib in tools/ubench/libcrw_base.so radar-sounder-crw_amd/libcrw_hip.so; do
  CRW_HIP_LIB=$PWD/$lib timeout -k 10 300 python bench.py --workload chain --modes bf16 --steps 3 --warmup 1 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); m=d['modes']['bf16']; print('$lib', 'chain bf16 ms/step', round(m['ms_per_step'],2), 'frac', round(m['frac'],3))" | tee -a $O/ab.log
done
