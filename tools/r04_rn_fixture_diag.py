"""Which side of a free-running comparison moves?  The reference's recorded fp32 CPU step (fixture resnet_train_*) against (a) the HIP
kernels, (b) the same modules on PyTorch-ROCm fp32, (c) the same modules in fp64 -- gradient of a few parameters, relative L2 distance
and cosine between every pair.  usage: python tools/r04_rn_fixture_diag.py [fixture]"""
import copy
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "radar-sounder-crw_amd")):
    sys.path.insert(0, p)
import encoder as crw_encoder  # noqa: E402
from oracle import crw_oracle as orc  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "resnet_train_32x32_B2T4N5"
g = dict(np.load(os.path.join(ROOT, "tests", "golden", name + ".npz")))
torch.manual_seed(int(g["seed"]))
base = crw_encoder.Resnet(False).cuda()
seq = torch.as_tensor(g["seq"]).cuda()
B, T, N, h, w = seq.shape
x = seq.reshape(-1, 1, h, w)
out = {}
stats = {}
for tag, conv, dt in (("hip", "bf16x3", torch.float32), ("torch32", None, torch.float32), ("torch64", None, torch.float64)):
    m = copy.deepcopy(base).to(dt)
    m.hip_convs = conv
    m.train(True)
    emb = m(x.to(dt)).reshape(B, T, N, -1)
    loss, _ = orc.walk_loss_torch(emb.double() if dt == torch.float64 else emb, float(g["tau"]))
    loss.backward()
    out[tag] = {k: p.grad.double().flatten().cpu() for k, p in m.named_parameters()}
    stats[tag] = {k: b.double().cpu() for k, b in m.named_buffers() if b.is_floating_point()}
    print(tag, "loss", loss.item(), "fixture", float(g["loss"]))
out["fixture"] = {k[5:]: torch.as_tensor(v).double().flatten() for k, v in g.items() if k.startswith("grad.")}
tags = list(out)
for k in out["fixture"]:
    print(k)
    for i, a in enumerate(tags):
        for b in tags[i + 1:]:
            u, v = out[a][k], out[b][k]
            print(f"   {a:8s} vs {b:8s}: |u-v|/|v| = {float((u - v).norm() / v.norm()):.3e}   cos = {float(torch.dot(u, v) / (u.norm() * v.norm())):.7f}"
                  f"   max|u-v|/max|v| = {float((u - v).abs().max() / v.abs().max()):.3e}")

print("BatchNorm running statistics after the step, worst relative difference per buffer (vs torch64) and the channel's mean^2 / var there")
for k in stats["torch64"]:
    if not k.endswith("running_var"):
        continue
    ref = stats["torch64"][k]
    mean = stats["torch64"][k.replace("running_var", "running_mean")] / 0.1   # one momentum step from 0
    var = (ref - 0.9) / 0.1                                                    # one momentum step from 1 (unbiased batch variance)
    line = f"   {k:44s}"
    for tag in ("hip", "torch32"):
        d = ((stats[tag][k] - ref).abs() / 0.1 / var.abs().clamp_min(1e-30))
        i = int(d.argmax())
        line += f"  {tag}: {float(d.max()):.2e} (mean^2/var {float(mean[i] ** 2 / var[i]):.1e}, var {float(var[i]):.2e})"
    print(line)
