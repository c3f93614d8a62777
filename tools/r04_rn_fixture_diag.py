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

# conditioning probe: the fp64 modules with the fixture's dLoss/dEmb injected; how much do the gradients move when (a) the injected
# gradient, (b) the input patches carry relative noise of 1e-6 (about what fp32 arithmetic leaves in them)?
demb = torch.as_tensor(g["demb"]).cuda().double()


def grads64(xin, gout):
    m = copy.deepcopy(base).double()
    m.hip_convs = None
    m.train(True)
    m(xin).backward(gout)
    return {k: p.grad.flatten().cpu() for k, p in m.named_parameters()}


gen = torch.Generator(device="cuda").manual_seed(1)
ref64 = grads64(x.double(), demb)
pa = grads64(x.double(), demb * (1 + 1e-6 * torch.randn(demb.shape, generator=gen, device="cuda", dtype=torch.float64)))
pb = grads64(x.double() * (1 + 1e-6 * torch.randn(x.shape, generator=gen, device="cuda", dtype=torch.float64)), demb)
pc = grads64(x.double(), demb.float().double())
print("fp64 modules, fixture's demb injected: relative change of the gradient under 1e-6 relative noise on (a) demb, (b) the patches; (c) demb rounded to fp32")
for k in ("model.fc.weight", "model.layer4.0.bn2.bias", "model.layer2.0.downsample.0.weight", "model.layer1.0.conv1.weight", "model.conv1.weight", "bn0.weight"):
    r = ref64[k]
    print(f"   {k:40s} (a) {float((pa[k] - r).norm() / r.norm()):.2e}  (b) {float((pb[k] - r).norm() / r.norm()):.2e}  (c) {float((pc[k] - r).norm() / r.norm()):.2e}"
          f"   vs fixture {float((out['fixture'][k] - r).norm() / r.norm()) if k in out['fixture'] else float('nan'):.2e}")
# the same injected gradient through the HIP kernels and through PyTorch-ROCm fp32
for tag, conv in (("hip", "bf16x3"), ("torch32", None)):
    m = copy.deepcopy(base)
    m.hip_convs = conv
    m.train(True)
    m(x).backward(demb.float())
    print(f"   {tag} (fp32, demb injected) vs fp64:", {k: f"{float((p.grad.double().flatten().cpu() - ref64[k]).norm() / ref64[k].norm()):.2e}"
                                                       for k, p in m.named_parameters() if k in ("model.fc.weight", "model.layer1.0.conv1.weight", "model.conv1.weight")})
