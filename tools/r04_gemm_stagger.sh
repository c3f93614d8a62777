#!/bin/bash
# A/B of the 256 x 256 bf16 chain GEMM with the LDS-DMA requests of half the waves moved into the k-tile (CRW_GEMM_STAGGER=1) against the
# default loop: parity subset under the switch, the four operand layouts at n = 4096 (batch 4 and 1) and n = 8192 for plain bf16 and hi/lo
# pairs, alternating the settings; then the whole walk at [1,32,4096,128] both ways.  GPU box, repo root.
O=$PWD/gpurun_out/r04g; mkdir -p $O; : > $O/stagger.log
CRW_GEMM_STAGGER=1 timeout -k 10 400 python -m pytest tests/test_hip_parity.py tests/test_k_shape.py -x -q -m gpu -k "gemm_bf16 or bf16_chain_modes_on_256" 2>&1 | tail -2 | tee -a $O/stagger.log
for rep in 1 2; do
  for r in 1 0; do
    for l in "0 0" "0 1" "1 0" "1 1"; do
      CRW_GEMM_STAGGER=$r timeout -k 10 120 python tools/probe_gemm.py bf16 4096 4 20 $l 2>&1 | grep -v amdgpu.ids | sed "s|^|STAGGER=$r |" | tee -a $O/stagger.log
    done
    CRW_GEMM_STAGGER=$r timeout -k 10 120 python tools/probe_gemm.py bf16 4096 1 20 0 0 2>&1 | grep -v amdgpu.ids | sed "s|^|STAGGER=$r |" | tee -a $O/stagger.log
    CRW_GEMM_STAGGER=$r timeout -k 10 120 python tools/probe_gemm.py bf16 8192 1 10 0 0 2>&1 | grep -v amdgpu.ids | sed "s|^|STAGGER=$r |" | tee -a $O/stagger.log
    CRW_GEMM_STAGGER=$r timeout -k 10 120 python tools/probe_gemm.py bf16x3 4096 4 10 0 0 2>&1 | grep -v amdgpu.ids | sed "s|^|STAGGER=$r |" | tee -a $O/stagger.log
    CRW_GEMM_STAGGER=$r timeout -k 10 120 python tools/probe_gemm.py bf16x3 4096 4 10 0 1 2>&1 | grep -v amdgpu.ids | sed "s|^|STAGGER=$r |" | tee -a $O/stagger.log
  done
done
for rep in 1 2; do
  for r in 1 0; do
    CRW_GEMM_STAGGER=$r timeout -k 10 200 python bench.py --workload chain --modes bf16,bf16x3 --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('STAGGER=$r chain', {k: (round(v['ms_per_step'],2), round(v['frac'],3)) for k, v in d['modes'].items()})" | tee -a $O/stagger.log
  done
done
