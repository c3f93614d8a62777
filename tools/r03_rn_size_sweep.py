"""Native Resnet pass over awkward batch sizes (padding rows, partial tiles, one tile, many tiles) against the PyTorch modules in
fp64 (the yardstick) and in fp32 (PyTorch-ROCm / MIOpen, for comparison).  GPU box: python tools/r03_rn_size_sweep.py"""
import sys, os, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "radar-sounder-crw_amd")]
import torch
import encoder as E

torch.manual_seed(0)
base = E.Resnet(False).cuda()
skip = ("fc0.bias",)  # mathematically zero gradient (a bias in front of a BatchNorm): pure rounding noise on every path


def errs(m, ref, y, yref):
    fe = (y.double() - yref).abs().max().item() / (yref.abs().max().item() + 1e-30)
    worst, wk = 0.0, ""
    for (k, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
        if k in skip:
            continue
        n = q.grad.norm().item()
        e = (p.grad.double() - q.grad).norm().item() / n if n > 0 else 0.0
        if e > worst:
            worst, wk = e, k
    return fe, worst, wk


for P in (2, 3, 63, 127, 128, 129, 255, 256, 257, 1000, 4095):
    x = torch.randn(P, 1, 16, 16).cuda()
    gy = torch.randn(P, 128).cuda()
    a = copy.deepcopy(base)
    b = copy.deepcopy(base); b.hip_convs = None
    c = copy.deepcopy(base).double(); c.hip_convs = None
    ya = a(x); ya.backward(gy)
    yb = b(x); yb.backward(gy)
    yc = c(x.double()); yc.backward(gy.double())
    torch.cuda.synchronize()
    ea, eb = errs(a, c, ya, yc), errs(b, c, yb, yc)
    print(f"P={P:5d}  HIP vs fp64: fwd {ea[0]:.1e} grad {ea[1]:.1e} ({ea[2]})   torch fp32 vs fp64: fwd {eb[0]:.1e} grad {eb[1]:.1e} ({eb[2]})", flush=True)
