"""Per-phase cycle counts of conv3x3_wgrad_kernel.  Needs `make -C radar-sounder-crw_amd/csrc STAMPS=1`."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "radar-sounder-crw_amd")]
import torch, crw_hip
lib = crw_hip.lib()
lib.crw_debug_conv_stamps.argtypes = [ctypes.c_void_p]; lib.crw_debug_conv_stamps.restype = None
P = 16128
for split in (3, 1):
    g = torch.Generator().manual_seed(1)
    mk = lambda c: (torch.randn(P, 100, c, generator=g) * 0.5).cuda()
    xf, dyf = mk(128), mk(128)
    xh = xf.bfloat16(); xl = (xf - xh.float()).bfloat16() if split == 3 else None
    dh = dyf.bfloat16(); dl = (dyf - dh.float()).bfloat16() if split == 3 else None
    for _ in range(2): crw_hip.enc_wgrad(split, dh, dl, xh, xl)
    st = torch.zeros(512, 64, dtype=torch.int64, device="cuda")
    lib.crw_debug_conv_stamps(ctypes.c_void_p(st.data_ptr()))
    crw_hip.enc_wgrad(split, dh, dl, xh, xl)
    torch.cuda.synchronize()
    lib.crw_debug_conv_stamps(None)
    m = st[:, :4].double().mean(0) / 126.0
    print("split", split, "cycles per patch [load+sync, bias, k-loop, barrier-wait]:", [round(v, 1) for v in m.tolist()], "total", round(m.sum().item(), 1))
    # phase drift of two co-resident workgroups (linear ids j and j + 256 share a CU): k-loop start offsets
    t = st[:, 4:].cpu()
    for j in (0, 9, 100):
        d = (t[j + 256] - t[j]).tolist()
        per = (t[j, 1:] - t[j, :-1]).double().mean().item()
        print("   wg", j, "vs", j + 256, "period", round(per), "offset of k-loop starts over patches:", d[:6], "...", d[20:24], "...", d[50:54])
