"""Per-phase s_memtime stamps of conv3x3_kernel at every encoder layer shape (needs `make STAMPS=1`).
usage: python tools/stamp_probe_layers.py"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "radar-sounder-crw_amd")]
import torch, crw_hip
lib = crw_hip.lib()
lib.crw_debug_conv_stamps.argtypes = [ctypes.c_void_p]; lib.crw_debug_conv_stamps.restype = None
P = 16128
for split in (3,):
    for cin, cout in ((32, 64), (64, 128), (128, 128)):
        g = torch.Generator().manual_seed(1)
        mk = lambda c: torch.relu(torch.randn(P, 100, c, generator=g) * 0.5).cuda()
        xf, dyf = mk(cin), mk(cout)
        xh, dh = xf.bfloat16(), dyf.bfloat16()
        xl, dl = (xf - xh.float()).bfloat16(), (dyf - dh.float()).bfloat16()
        w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).cuda(); b = torch.zeros(cout).cuda()
        fh, fl, bh, bl = crw_hip.enc_pack_weights(w, split)
        calls = {"fwd": lambda: crw_hip.enc_conv3x3(0, split, xh, xl, fh, fl, cout, bias=b),
                 "bwd": lambda: crw_hip.enc_conv3x3(1, split, dh, dl, bh, bl, cin, mask=xh)}
        for name, fn in calls.items():
            for _ in range(2): fn()
            st = torch.zeros(P, 5, dtype=torch.int64, device="cuda")
            lib.crw_debug_conv_stamps(ctypes.c_void_p(st.data_ptr()))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record()
            torch.cuda.synchronize()
            lib.crw_debug_conv_stamps(None)
            s = st.cpu().double()
            d = s[:, 1:] - s[:, :-1]
            span = (s[:, 4].max() - s[:, 0].min()).item()
            print(f"{name} cin={cin} cout={cout}: ticks/phase [load, kloop, barrier, epilogue] =",
                  [round(x, 1) for x in d.mean(0).tolist()], "WG total", round((s[:, 4] - s[:, 0]).mean().item(), 1),
                  "kernel span", span, "ticks =", round(e0.elapsed_time(e1) * 1e3, 1), "us; WG-lifetimes in flight:",
                  round((s[:, 4] - s[:, 0]).sum().item() / span, 1))
