import os, sys, ctypes
ROOT="/root/repo"
sys.path[:0]=[ROOT, os.path.join(ROOT,"radar-sounder-crw_amd")]
import torch, crw_hip
lib=crw_hip.lib()
# needs a `make -C radar-sounder-crw_amd/csrc STAMPS=1` build
lib.crw_debug_conv_stamps.argtypes=[ctypes.c_void_p]; lib.crw_debug_conv_stamps.restype=None
P=16128
for split in (3,1):
    g=torch.Generator().manual_seed(1)
    xf=(torch.randn(P,100,128,generator=g)*0.5).cuda()
    xh=xf.bfloat16(); xl=(xf-xh.float()).bfloat16() if split==3 else None
    w=(torch.randn(128,128,3,3,generator=g)*0.05).cuda(); b=torch.zeros(128).cuda()
    fh,fl,bh,bl=crw_hip.enc_pack_weights(w,split)
    for _ in range(2): crw_hip.enc_conv3x3(0,split,xh,xl,fh,fl,128,bias=b)
    st=torch.zeros(P,5,dtype=torch.int64,device="cuda")
    lib.crw_debug_conv_stamps(ctypes.c_void_p(st.data_ptr()))
    crw_hip.enc_conv3x3(0,split,xh,xl,fh,fl,128,bias=b)
    torch.cuda.synchronize()
    lib.crw_debug_conv_stamps(None)
    s=st.cpu().double()
    d=(s[:,1:]-s[:,:-1])
    # s_memtime ticks at 100 MHz? report in raw ticks and as fraction
    tot=(s[:,4]-s[:,0])
    print("split",split,"mean ticks per phase [load, kloop, wait-barrier, epilogue]:", [round(x,1) for x in d.mean(0).tolist()], "total", round(tot.mean().item(),1))
    print("   kernel span ticks:", (s[:,4].max()-s[:,0].min()).item())
