"""Per-phase cycle counts of front_bwd_kernel.  Needs `make -C radar-sounder-crw_amd/csrc STAMPS=1`."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "radar-sounder-crw_amd")]
import torch, crw_hip
lib = crw_hip.lib()
lib.crw_debug_front_stamps.argtypes = [ctypes.c_void_p]; lib.crw_debug_front_stamps.restype = None
P = 16128
names = ["load", "conv1", "pool1", "conv2", "pool2bwd", "wgrad2", "bwddata2", "pool1bwd", "wgrad1", "-"]
for split in (3, 1):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(P, 1, 16, 16, generator=g).cuda()
    w1 = (torch.randn(8, 1, 5, 5, generator=g) * 0.2).cuda(); b1 = torch.zeros(8).cuda()
    w2 = (torch.randn(32, 8, 5, 5, generator=g) * 0.07).cuda(); b2 = torch.zeros(32).cuda()
    dy = torch.randn(P, 100, 32, generator=g).cuda()
    w2p = crw_hip.enc_front_pack(w2, split)
    for _ in range(2):
        crw_hip.enc_front_bwd(split, x, w1, b1, w2p[:2], b2, w2p[2:], dy)
    st = torch.zeros(256, 10, dtype=torch.int64, device="cuda")
    lib.crw_debug_front_stamps(ctypes.c_void_p(st.data_ptr()))
    crw_hip.enc_front_bwd(split, x, w1, b1, w2p[:2], b2, w2p[2:], dy)
    torch.cuda.synchronize()
    lib.crw_debug_front_stamps(None)
    m = st.double().mean(0) / 63.0
    print("split", split, "ticks per patch:", {n: round(v, 1) for n, v in zip(names, m.tolist())}, "total", round(m.sum().item(), 1))
