#!/bin/bash
# CRW_GEMM_STAGGER = 0 | 1 (requests of the second half of the waves at 1/2 of the k-tile) | 2 (at 1/4): four layouts, plain bf16, n = 4096
O=$PWD/gpurun_out/r04g; mkdir -p $O; : > $O/stagger2.log
for rep in 1 2; do
  for r in 0 1 2; do
    for l in "0 0" "0 1" "1 0" "1 1"; do
      CRW_GEMM_STAGGER=$r timeout -k 10 120 python tools/probe_gemm.py bf16 4096 4 20 $l 2>&1 | grep -v amdgpu.ids | sed "s|^|STAGGER=$r |" | tee -a $O/stagger2.log
    done
  done
done
