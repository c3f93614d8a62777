#!/usr/bin/env python3
"""Micro-benchmark of the chain GEMM kernels (C ABI crw_gemm_f32 / crw_gemm_bf16).
usage: python tools/probe_gemm.py [kind=f32|bf16|bf16x3] [n] [batch] [iters] [ta] [tb]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "radar-sounder-crw_amd")]
import torch
import crw_hip


def run(kind, n, batch, iters, ta=0, tb=0):
    A = torch.rand(batch, n, n, device="cuda")
    B = torch.rand(batch, n, n, device="cuda")
    if kind == "f32":
        C = crw_hip.gemm_f32(A, B, transA=ta, transB=tb)
        call = lambda: crw_hip.gemm_f32(A, B, C, transA=ta, transB=tb)
        mult = 1
    else:
        split = 3 if kind == "bf16x3" else 1
        C, ws = crw_hip.gemm_bf16(A, B, transA=ta, transB=tb, split=split)
        call = lambda: crw_hip.gemm_bf16(A, B, C, transA=ta, transB=tb, split=split, ws=ws, convert=False)
        mult = split
    for _ in range(2):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        call()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    tf = 2.0 * n ** 3 * batch * mult / (ms * 1e-3) / 1e12
    print(f"{kind:7s} n={n:5d} batch={batch:3d} ta={ta} tb={tb}: {ms * 1e3:9.1f} us/launch  {tf:8.1f} TFLOP/s (MFMA flops executed)",
          flush=True)


if __name__ == "__main__":
    a = sys.argv[1:]
    if a:
        run(a[0], int(a[1]), int(a[2]), int(a[3]) if len(a) > 3 else 10, int(a[4]) if len(a) > 4 else 0,
            int(a[5]) if len(a) > 5 else 0)
    else:
        for kind in ("bf16", "bf16x3"):
            for n, batch in ((1024, 16), (2048, 4), (4096, 1), (4096, 3), (8192, 1)):
                run(kind, n, batch, 10)
        for kind in ("bf16", "bf16x3"):
            for ta, tb in ((0, 1), (1, 0), (1, 1)):
                run(kind, 4096, 1, 10, ta, tb)
