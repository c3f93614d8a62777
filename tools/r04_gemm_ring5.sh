#!/bin/bash
# A/B of the plain-bf16 256 x 256 chain GEMM: ring of five 32-deep half-tiles (default) against the two-stage ring of 64-deep tiles
# (CRW_GEMM_RING5=0), the four operand layouts at n = 4096 (batch 4 and 1) and n = 8192, alternating the settings; then the whole
# walk at [1,32,4096,128] both ways.  GPU box, repo root.
O=$PWD/gpurun_out/r04g; mkdir -p $O; : > $O/ring5.log
for rep in 1 2; do
  for r in 1 0; do
    for l in "0 0" "0 1" "1 0" "1 1"; do
      CRW_GEMM_RING5=$r timeout -k 10 120 python tools/probe_gemm.py bf16 4096 4 20 $l 2>&1 | grep -v amdgpu.ids | sed "s|^|RING5=$r |" | tee -a $O/ring5.log
    done
    CRW_GEMM_RING5=$r timeout -k 10 120 python tools/probe_gemm.py bf16 4096 1 20 0 0 2>&1 | grep -v amdgpu.ids | sed "s|^|RING5=$r |" | tee -a $O/ring5.log
    CRW_GEMM_RING5=$r timeout -k 10 120 python tools/probe_gemm.py bf16 8192 1 10 0 0 2>&1 | grep -v amdgpu.ids | sed "s|^|RING5=$r |" | tee -a $O/ring5.log
  done
done
for rep in 1 2; do
  for r in 1 0; do
    CRW_GEMM_RING5=$r timeout -k 10 200 python bench.py --workload chain --modes bf16 --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('RING5=$r chain bf16', round(d['modes']['bf16']['ms_per_step'],2), 'ms', round(d['modes']['bf16']['frac'],3))" | tee -a $O/ring5.log
  done
done
