"""Staged probe: can the training step be captured in a HIP graph on this stack?  (diagnostic)"""
import os, sys, time, faulthandler
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "radar-sounder-crw_amd")]
import torch
stage = sys.argv[1] if len(sys.argv) > 1 else "torch"
print("stage", stage, flush=True)
if stage == "torch":  # pure torch ops
    lin = torch.nn.Linear(64, 64).cuda()
    x = torch.randn(32, 64, device="cuda")
    opt = torch.optim.Adam(lin.parameters(), lr=1e-3, capturable=True)
    def step():
        opt.zero_grad(set_to_none=True); l = lin(x).square().mean(); l.backward(); opt.step(); return l
else:
    import crw_hip, model as M, encoder as E, dataset as D
    crw_hip.lib()
    ds = D.RGDataset.synthetic(64, 256, 8, (16, 16), (8, 0), seed=11)
    seq = torch.stack([ds[0], ds[8]]).cuda()
    torch.manual_seed(11)
    enc = E.CNN(False)
    if stage == "walk":
        enc.hip_convs = None
    net = M.CRW(enc, 0.01, False).cuda()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, capturable=True)
    def step():
        opt.zero_grad(set_to_none=True); l, _ = net(seq); l.backward(); opt.step(); return l
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): l = step()
torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 20
print("eager ms", eager * 1e3, flush=True)
g = torch.cuda.CUDAGraph()
print("capturing", flush=True)
with torch.cuda.graph(g):
    lg = step()
print("captured", flush=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): g.replay()
torch.cuda.synchronize(); gr = (time.perf_counter() - t0) / 20
print(f"stage {stage}: eager {eager*1e3:.3f} ms, graph {gr*1e3:.3f} ms, loss {l.item():.6f} / {lg.item():.6f}", flush=True)
