#!/bin/bash
# timing-only diagnostic variants of the streamed weight-gradient kernel (STAMPS build: tools/ubench/libcrw_diag.so)
O=$PWD/gpurun_out/r02w; mkdir -p $O
for d in 0 6 0 6; do
  CRW_HIP_LIB=$PWD/tools/ubench/libcrw_diag.so CRW_WGRAD_DIAG=$d timeout -k 10 120 python tools/probe_conv.py 128 128 3 16128 10 2>&1 | grep wgrad | sed "s/^/DIAG=$d /" | tee -a $O/diag.log
done
