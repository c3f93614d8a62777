#!/usr/bin/env python3
"""Stand-alone timing of the 64-channel block's weight gradient at the bench batch (P = 16128, 5 x 5 x 64 -> 64) through crw_rn_wgrad,
alone on the chip -- the harness of the second experiment in profiles/r04_rn_row_experiment.log; run under rocprofv3 --kernel-trace
--stats for the kernel durations.  usage: python tools/r04_wrow_probe.py [P] [H] [W]"""
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
d = torch.randn(P, Hm * Wm * 64, generator=g).cuda()
xp = H.rn_split(x, P, Hm * Wm * 64)
dp = H.rn_split(d, P, Hm * Wm * 64)
for _ in range(3):
    H.rn_wgrad(H.RN_FWD, P, (Hm, Wm, 64), (Hm, Wm, 64), (3, 3), 1, 1, xp, dp)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    H.rn_wgrad(H.RN_FWD, P, (Hm, Wm, 64), (Hm, Wm, 64), (3, 3), 1, 1, xp, dp)
e1.record()
torch.cuda.synchronize()
print(f"P={P} {Hm}x{Wm}: {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us per call (kernel + slab sum + host gaps)", flush=True)
