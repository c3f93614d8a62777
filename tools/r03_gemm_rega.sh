#!/bin/bash
# A/B of the register-staged A operand (CRW_GEMM_REGA=1) in the plain-bf16 chain GEMM: parity tests under the switch, then the
# four operand layouts at n = 4096 (batch 1 and 4) and n = 8192, alternating the two settings.  GPU box, repo root.
O=$PWD/gpurun_out/r03g; mkdir -p $O; : > $O/rega.log
CRW_GEMM_REGA=1 timeout -k 10 400 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "gemm_bf16 or bf16" 2>&1 | tail -3 | tee -a $O/rega.log || exit 1
for rep in 1 2; do
  for r in 0 1; do
    for l in "0 0" "0 1" "1 0" "1 1"; do
      CRW_GEMM_REGA=$r timeout -k 10 120 python tools/probe_gemm.py bf16 4096 4 20 $l 2>&1 | grep -v amdgpu.ids | sed "s|^|REGA=$r |" | tee -a $O/rega.log
    done
    CRW_GEMM_REGA=$r timeout -k 10 120 python tools/probe_gemm.py bf16 4096 1 20 0 0 2>&1 | grep -v amdgpu.ids | sed "s|^|REGA=$r |" | tee -a $O/rega.log
    CRW_GEMM_REGA=$r timeout -k 10 120 python tools/probe_gemm.py bf16 8192 1 10 0 0 2>&1 | grep -v amdgpu.ids | sed "s|^|REGA=$r |" | tee -a $O/rega.log
  done
done
