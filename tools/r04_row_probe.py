#!/usr/bin/env python3
"""Stand-alone timing of the first BasicBlock's 3x3 products at the bench batch (P = 16128, 5 x 5 x 64 -> 64): forward and
backward-data through crw_rn_conv, alone on the chip (no side stream) -- the harness of profiles/r04_rn_row_experiment.log (run it
under `rocprofv3 --kernel-trace --stats` for kernel durations; the event times below include the host's launch gaps).
usage: python tools/r04_row_probe.py [P] [H] [W]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "radar-sounder-crw_amd"))
import torch
import crw_hip as H

P = int(sys.argv[1]) if len(sys.argv) > 1 else 16128
Hm = int(sys.argv[2]) if len(sys.argv) > 2 else 5
Wm = int(sys.argv[3]) if len(sys.argv) > 3 else 5
H.lib()
g = torch.Generator().manual_seed(3)
x = torch.randn(P, Hm * Wm * 64, generator=g).cuda()
w = (torch.randn(64, 64, 3, 3, generator=g) / 24.0).cuda()
xp = H.rn_split(x, P, Hm * Wm * 64)
wp = H.rn_pack_conv(w)
for mode, name, wq in ((H.RN_FWD, "fwd", wp[:2]), (H.RN_BWD, "bwd-data", wp[2:])):
    for stats in ((True, False) if mode == H.RN_FWD else (False,)):
        for _ in range(3):
            H.rn_conv(mode, P, (Hm, Wm, 64), (Hm, Wm), 64, (3, 3), 1, 1, xp, wq, stats=stats)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            H.rn_conv(mode, P, (Hm, Wm, 64), (Hm, Wm), 64, (3, 3), 1, 1, xp, wq, stats=stats)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        print(f"{name:9s} stats={int(stats)} P={P} {Hm}x{Wm}: {us:7.1f} us/launch", flush=True)
