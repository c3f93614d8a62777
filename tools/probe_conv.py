#!/usr/bin/env python3
"""Micro-benchmark of one encoder conv layer (forward / backward-data / weight-gradient kernels).
usage: python tools/probe_conv.py [cin cout split P iters]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "radar-sounder-crw_amd")]
import torch
import crw_hip


def run(cin, cout, split, P, iters):
    g = torch.Generator().manual_seed(1)
    mk = lambda c: (torch.randn(P, 100, c, generator=g) * 0.5).cuda()
    xf, dyf = mk(cin), mk(cout)
    xh = xf.bfloat16(); xl = (xf - xh.float()).bfloat16() if split == 3 else None
    dh = dyf.bfloat16(); dl = (dyf - dh.float()).bfloat16() if split == 3 else None
    w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).cuda()
    b = torch.zeros(cout).cuda()
    fh, fl, bh, bl = crw_hip.enc_pack_weights(w, split)
    flops = 2.0 * P * 100 * cin * cout * 9

    def timeit(name, fn):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        print(f"{name:8s} cin={cin:3d} cout={cout:3d} split={split} P={P}: {ms * 1e3:8.1f} us  "
              f"{flops / ms / 1e9:7.1f} TFLOP/s useful  ({flops * (3 if split == 3 else 1) / ms / 1e9:7.1f} MFMA-executed)", flush=True)

    timeit("fwd", lambda: crw_hip.enc_conv3x3(0, split, xh, xl, fh, fl, cout, bias=b))
    timeit("bwd-data", lambda: crw_hip.enc_conv3x3(1, split, dh, dl, bh, bl, cin, mask=xh))
    timeit("wgrad", lambda: crw_hip.enc_wgrad(split, dh, dl, xh, xl))


if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:]]
    if a:
        run(*a)
    else:
        for split in (3, 1):
            for cin, cout in ((128, 128), (64, 128), (32, 64)):
                run(cin, cout, split, 16128, 5)
