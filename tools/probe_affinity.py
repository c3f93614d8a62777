#!/usr/bin/env python3
"""Accuracy (vs fp64) and time of the affinity build and its backward at a large node count.
usage: python tools/probe_affinity.py [N T C iters]   (CRW_AFFINITY_F32=1: the fp32-MFMA kernels)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "radar-sounder-crw_amd")]
import torch
import crw_hip

N, T, C, iters = (int(v) for v in (sys.argv[1:] + ["512", "8", "128", "10"])[:4])
tau = 0.01
g = torch.Generator().manual_seed(3)
emb = torch.randn(1, T, N, C, generator=g).cuda()
A, ehat, norm, stats = crw_hip.affinity_fwd(emb, tau)
e64 = torch.nn.functional.normalize(emb.double(), dim=-1)
A64 = torch.einsum("btnc,btmc->btnm", e64[:, :-1], e64[:, 1:]) / tau
print(f"N={N} T={T} C={C}: max |A - A64| = {(A.double() - A64).abs().max().item():.3e}  (max |A| = {A64.abs().max().item():.1f})")
rmax = A64.max(-1).values
print(f"  row max err {(stats[0].double().view_as(rmax) - rmax).abs().max().item():.3e}", end="")
rsum = torch.exp(A64 - rmax[..., None]).sum(-1)
print(f"  row sum rel err {((stats[1].double().view_as(rsum) - rsum) / rsum).abs().max().item():.3e}")
dA = torch.randn(A.shape, generator=g).cuda() * 1e-3
demb = crw_hip.affinity_bwd(dA, ehat, norm, tau)
e = emb.double().requires_grad_(True)
en = torch.nn.functional.normalize(e, dim=-1)
(torch.einsum("btnc,btmc->btnm", en[:, :-1], en[:, 1:]) / tau * dA.double()).sum().backward()
print(f"  max |demb - demb64| / max |demb64| = {((demb.double() - e.grad).abs().max() / e.grad.abs().max()).item():.3e}")
for name, fn in (("fwd", lambda: crw_hip.affinity_fwd(emb, tau)), ("bwd", lambda: crw_hip.affinity_bwd(dA, ehat, norm, tau))):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"  {name}: {e0.elapsed_time(e1) / iters * 1e3:9.1f} us")
