#!/usr/bin/env python3
"""Golden-vector generator: runs the *reference itself* on CPU and stores inputs + outputs.

Runs ONLY in the build container (needs /root/reference, which never travels to the GPU box).
It writes small ``.npz`` fixtures next to this file; the fixtures hold tensors only (inputs and
the reference's outputs) -- no reference source in any form.

How the reference is made to run here:
  * ``sys.path`` gets ``/root/reference/src`` (the reference installs its modules flat:
    ``model``, ``encoder``, ``utils``, ``dataset``, ``imported.labelprop``).
  * ``ruptures`` and ``torchvision`` are not installed in this image.  ``utils`` imports them at
    module scope, so empty placeholder modules are registered.  ``ruptures`` is only reached
    inside a bare ``try/except`` (src/utils.py:126-132) -> ``change_idx`` is ``None`` here.
    ``torchvision.transforms.Resize((N,1), NEAREST)`` (src/utils.py:139) is restated as
    legacy-nearest interpolation (pure index arithmetic: row floor(i*rows/N), column 0).
  * the reference hard-codes ``'cuda'``; those spots are redirected to CPU:
    ``model.zeros`` (src/model.py:36), ``Tensor.cuda``, ``Tensor.to('cuda')``,
    ``torch.zeros(device='cuda')`` (src/utils.py:90,119,137,141,143; labelprop.py:103).
  * the whole-radargram pipeline (``segment_*`` fixtures) runs the reference's own driver,
    ``scripts/test/test_all.py`` ``main(args)``, imported from /root/reference with its private-data
    factories (``create_dataset`` / ``get_reference`` / ``create_model`` / ``load``) pointed at
    synthetic tensors, plotting and the sklearn reports turned into recorders, and -- because PELT
    is unavailable -- ``propagate`` wrapped so that a chosen change index comes back for chosen
    radargrams (everything else, including the correction / reverse / merge logic, is the reference's).
  * ``model.cross_entropy`` is wrapped with a recorder so every per-cycle transition product
    ``At_k`` (src/model.py:45) is captured -- the scalar loss alone is a weak parity probe.

Usage:  python tests/golden/make_golden.py          (rewrites every fixture)
"""
import os
import sys
import types
import contextlib

sys.dont_write_bytecode = True
REF = os.environ.get("CRW_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as TF


# ----------------------------------------------------------------------------- import plumbing
def _install_placeholders():
    if "ruptures" not in sys.modules:
        sys.modules["ruptures"] = types.ModuleType("ruptures")  # Pelt absent -> AttributeError -> except
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tr = types.ModuleType("torchvision.transforms")

        class InterpolationMode:
            NEAREST = "nearest"

        class Resize:
            def __init__(self, size, interpolation=None):
                self.size = size

            def __call__(self, img):
                out = TF.interpolate(img[None].float(), size=self.size, mode="nearest")[0]
                return out.to(img.dtype)

        fn = types.ModuleType("torchvision.transforms.functional")

        def resize(img, size, interpolation=None):
            return Resize(size)(img)

        fn.resize = resize
        tr.InterpolationMode = InterpolationMode
        tr.Resize = Resize
        tr.functional = fn
        tv.transforms = tr
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.transforms"] = tr
        sys.modules["torchvision.transforms.functional"] = fn


def import_reference():
    _install_placeholders()
    sys.path.insert(0, os.path.join(REF, "src"))
    import model as ref_model  # noqa
    import encoder as ref_encoder  # noqa
    import utils as ref_utils  # noqa
    import dataset as ref_dataset  # noqa
    from imported import labelprop as ref_lp  # noqa

    _zeros = torch.zeros

    def zeros_cpu(*a, **k):
        k.pop("device", None)
        return _zeros(*a, **k)

    ref_model.zeros = zeros_cpu
    return ref_model, ref_encoder, ref_utils, ref_dataset, ref_lp


@contextlib.contextmanager
def cuda_is_cpu():
    """Redirect the reference's hard-coded 'cuda' to CPU for the duration of a call."""
    _zeros, _to, _cuda = torch.zeros, torch.Tensor.to, torch.Tensor.cuda

    def zeros(*a, **k):
        if k.get("device") == "cuda":
            k.pop("device")
        return _zeros(*a, **k)

    def to(self, *a, **k):
        a = tuple("cpu" if (isinstance(x, str) and x == "cuda") else x for x in a)
        if k.get("device") == "cuda":
            k["device"] = "cpu"
        return _to(self, *a, **k)

    torch.zeros, torch.Tensor.to, torch.Tensor.cuda = zeros, to, (lambda self, *a, **k: self)
    try:
        yield
    finally:
        torch.zeros, torch.Tensor.to, torch.Tensor.cuda = _zeros, _to, _cuda


class FixedFeatures(nn.Module):
    """Stub encoder: ignores the patches and returns a fixed [P, C] feature table (a Parameter,
    so the reference's autograd yields dLoss/dEmb)."""

    def __init__(self, table):
        super().__init__()
        self.table = nn.Parameter(table.clone())

    def forward(self, x):
        return self.table


# ----------------------------------------------------------------------------- synthetic inputs
def layered_features(B, T, N, C, noise, gen):
    """Unnormalised features with along-track coherence: node n keeps a base direction that
    drifts slowly with t, plus noise -> informative affinities (loss far from the random value)."""
    base = torch.randn(1, 1, N, C, generator=gen)
    drift = torch.randn(B, 1, N, C, generator=gen) * 0.15
    t = torch.arange(T).view(1, T, 1, 1).float()
    emb = base + drift * t / max(T - 1, 1) + noise * torch.randn(B, T, N, C, generator=gen)
    return (emb * (1.0 + 0.5 * torch.rand(B, T, N, 1, generator=gen))).float()


def layered_radargram(H, W, gen):
    r = torch.arange(H).view(H, 1).float()
    c = torch.arange(W).view(1, W).float()
    return (torch.sin(2 * np.pi * r / 64 + 0.002 * c) + 0.6 * torch.randn(H, W, generator=gen)).float()


def unfold_items(rg, T, h, w, oh, ow, index):
    """Same cut as RGDataset.__getitem__ (src/dataset.py:34-39) done with plain slicing."""
    H, W = rg.shape
    nh = (H - oh) // (h - oh)
    out = torch.empty(T, nh, h, w)
    for t in range(T):
        c0 = (w - ow) * index + t * (w - ow)
        for n in range(nh):
            r0 = n * (h - oh)
            out[t, n] = rg[r0:r0 + h, c0:c0 + w]
    return out


# ----------------------------------------------------------------------------- generators
def run_walk_case(ref_model, name, B, T, N, C, tau, noise, seed):
    gen = torch.Generator().manual_seed(seed)
    emb = layered_features(B, T, N, C, noise, gen)
    enc = FixedFeatures(emb.reshape(B * T * N, C))
    crw = ref_model.CRW(enc, tau, False)
    recorded = []
    orig_ce = ref_model.cross_entropy

    def ce(input, target, *a, **k):
        recorded.append(input.detach().transpose(1, 2).clone())  # input = At^T (src/model.py:45)
        return orig_ce(input, target, *a, **k)

    ref_model.cross_entropy = ce
    try:
        seq = torch.zeros(B, T, N, 2, 2)
        loss, A = crw(seq)
    finally:
        ref_model.cross_entropy = orig_ce
    out = dict(emb=emb.numpy(), tau=np.float32(tau), A=A.detach().numpy())
    if torch.is_tensor(loss):
        loss.backward()
        out["loss"] = loss.detach().numpy()
        out["demb"] = enc.table.grad.reshape(B, T, N, C).numpy()
        out["At"] = torch.stack(recorded, 1).numpy()  # [B, T-2, N, N]
    else:  # T < 3: the reference returns the python number 0/N
        out["loss"] = np.float32(loss)
    np.savez(os.path.join(HERE, name + ".npz"), **out)
    print(f"{name}: loss={float(out['loss']):.7f}  A{tuple(A.shape)}")


def run_cnn_case(ref_model, ref_encoder, name, B, T, N, hw, tau, pos_embed, seed, sd_name):
    torch.manual_seed(seed)
    enc = ref_encoder.CNN(pos_embed)
    sd = {k: v.detach().numpy().copy() for k, v in enc.state_dict().items()}
    np.savez(os.path.join(HERE, sd_name + ".npz"), **sd)
    gen = torch.Generator().manual_seed(seed)
    h, w = hw
    oh = h // 2
    H = N * (h - oh) + oh
    rg = layered_radargram(H, 2 * T * w, gen)
    seq = torch.stack([unfold_items(rg, T, h, w, oh, 0, i * T) for i in range(B)])
    crw = ref_model.CRW(enc, tau, pos_embed)
    recorded = []
    orig_ce = ref_model.cross_entropy

    def ce(input, target, *a, **k):
        recorded.append(input.detach().transpose(1, 2).clone())
        return orig_ce(input, target, *a, **k)

    ref_model.cross_entropy = ce
    feats = {}
    hook = enc.register_forward_hook(lambda m, i, o: feats.__setitem__("emb", o.detach().clone()))
    try:
        with cuda_is_cpu():
            loss, A = crw(seq)
    finally:
        ref_model.cross_entropy = orig_ce
        hook.remove()
    loss.backward()
    out = dict(seq=seq.numpy(), tau=np.float32(tau), pos_embed=np.bool_(pos_embed),
               emb=feats["emb"].reshape(B, T, N, -1).numpy(), A=A.detach().numpy(),
               loss=loss.detach().numpy(), At=torch.stack(recorded, 1).numpy())
    for k, p in enc.named_parameters():
        out["grad." + k] = p.grad.numpy()
    np.savez(os.path.join(HERE, name + ".npz"), **out)
    print(f"{name}: loss={float(loss):.7f} seq{tuple(seq.shape)}")


class PatchFlatten(nn.Module):
    """Stub encoder for inference cases: the 'patch' pixels ARE the feature vector."""

    def forward(self, x):
        return x.flatten(1)


def moving_layer_features(T, N, C, noise, gen):
    """Features of a layered medium whose interfaces undulate along-track: the prototype of node n
    at frame t is interpolated at depth n + 2.5*sin(2*pi*t/17) + 0.08*t, so propagated labels must
    move with the layers (a constant label map would be wrong)."""
    proto = torch.randn(N + 16, C, generator=gen)
    t = torch.arange(T).float()
    depth = torch.arange(N).float()[None, :] + 2.5 * torch.sin(2 * np.pi * t / 17)[:, None] + 0.08 * t[:, None] + 6
    lo = depth.floor().long().clamp(0, N + 14)
    fr = (depth - lo.float()).unsqueeze(-1)
    emb = proto[lo] * (1 - fr) + proto[lo + 1] * fr + noise * torch.randn(T, N, C, generator=gen)
    return (emb * (1.0 + 0.5 * torch.rand(T, N, 1, generator=gen))).float()


def run_labelprop_case(ref_utils, ref_lp, name, T, N, C, M, cfg, use_last, seed, noise=0.35):
    gen = torch.Generator().manual_seed(seed)
    emb = moving_layer_features(T, N, C, noise, gen)  # [T,N,C], natural (unflipped) frame order
    enc = PatchFlatten()
    h = 4
    rows = N * h
    # layered reference segmentation, M horizontal bands with a wiggle
    seg = torch.zeros(rows, 3)
    edges = torch.linspace(0, rows, M + 1)[1:-1]
    for i in range(rows):
        seg[i, :] = float((edges <= i).sum())
    lp = ref_lp.LabelPropVOS_CRW(cfg)
    seq = emb.reshape(T, N, C // 4, 4).clone()
    with cuda_is_cpu():
        pred, xent, change_idx = ref_utils.propagate(seq, seg, enc, lp, M, False, use_last)
    np.savez(os.path.join(HERE, name + ".npz"), emb=emb.numpy(), seg_ref=seg.numpy(),
             nclasses=np.int32(M), cxt_size=np.int32(cfg["CXT_SIZE"]), radius=np.int32(cfg["RADIUS"]),
             temp=np.float32(cfg["TEMP"]), knn=np.int32(cfg["KNN"]), use_last=np.bool_(use_last),
             pred=pred.numpy(), xent=xent.numpy())
    moved = int((pred[:, 1:] != pred[:, :-1]).sum())
    print(f"{name}: pred{tuple(pred.shape)} classes={sorted(set(pred.flatten().tolist()))} "
          f"label changes along-track={moved} change_idx={change_idx}")


def run_labelprop_grid_case(ref_lp, name, T, h, w, C, M, cfg, seed, noise=0.35):
    """The reference's LabelPropVOS_CRW.predict on a 2-D node grid (h x w nodes per frame, src/imported/labelprop.py:67-115 with the
    Euclidean-radius mask of MaskedAttention, maskedatt.py:222-245), driven frame by frame the way src/utils.py:148-160 drives it
    on its N x 1 grids.  Nodes are stored in row-major order: emb [T, h*w, C], seed [h*w], pred [h*w, T]."""
    gen = torch.Generator().manual_seed(seed)
    emb = moving_layer_features(T, h * w, C, noise, gen)                     # [T, N, C]
    ehat = emb / emb.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    seed_lab = (torch.arange(h * w) // w * M // h).float()                    # M horizontal bands
    lp = ref_lp.LabelPropVOS_CRW(cfg)
    as_feat = lambda n: ehat[n].t().reshape(1, C, h, w)
    mask = (seed_lab[None, :] == torch.arange(M)[:, None]).float().reshape(1, M, h, w)
    feats, masks = [as_feat(0)], [mask]
    pred = torch.zeros(h * w, T)
    pred[:, 0] = seed_lab
    with cuda_is_cpu():
        for n in range(1, T):
            mask = lp.predict(feats, masks, as_feat(n))
            feats.append(as_feat(n))
            masks.append(mask)
            pred[:, n] = mask.argmax(1).flatten().float()
    np.savez(os.path.join(HERE, name + ".npz"), emb=emb.numpy(), seed_labels=seed_lab.numpy(), grid=np.int32([h, w]),
             nclasses=np.int32(M), cxt_size=np.int32(cfg["CXT_SIZE"]), radius=np.int32(cfg["RADIUS"]),
             temp=np.float32(cfg["TEMP"]), knn=np.int32(cfg["KNN"]), pred=pred.numpy())
    print(f"{name}: grid {h}x{w} pred{tuple(pred.shape)} classes={sorted(set(pred.flatten().tolist()))} "
          f"label changes along the frames={int((pred[:, 1:] != pred[:, :-1]).sum())}")


def layered_segmentation(rows, cols, M, gen, wiggle=3.0):
    """[rows, cols] class ids: M sub-horizontal bands whose interfaces undulate along-track."""
    c = torch.arange(cols).float()
    out = torch.zeros(rows, cols)
    for k in range(1, M):
        edge = rows * k / M + wiggle * torch.sin(2 * np.pi * c / (37.0 + 11 * k) + k)
        out += (torch.arange(rows).float()[:, None] >= edge[None, :]).float()
    return out


def run_segment_case(ref_dataset, name, dataset_id, nclasses, T, hw, oh, H_rg, n_rg, seg_rows_extra, cfg, use_last,
                     correction, forced_change, seed):
    """scripts/test/test_all.py main(args) of the reference on synthetic data -> forward (+corrected) int8 map that
    the script saves (`predicted_map.pt`, test_all.py:128) and the final map after the reversed-pass merge
    (test_all.py:132-159, what its report is computed on)."""
    import importlib
    import tempfile
    import argparse
    sys.path.insert(0, os.path.join(REF, "scripts", "test"))
    import test_all as ref_main
    importlib.reload(ref_main)
    gen = torch.Generator().manual_seed(seed)
    h, w = hw
    W_rg = n_rg * T * w
    # patch pixels ARE the features (PatchFlatten): a layered medium so that labels must follow the layers
    r = torch.arange(H_rg).float()[:, None]
    c = torch.arange(W_rg).float()[None, :]
    rg = (torch.sin(2 * np.pi * (r + 2.5 * torch.sin(2 * np.pi * c / 61.0)) / 9.0) + 0.5 * torch.cos(0.37 * r + 0.011 * c)
          + 0.25 * torch.randn(H_rg, W_rg, generator=gen)).float()
    N = (H_rg - oh) // (h - oh)
    seg_full = layered_segmentation(H_rg + seg_rows_extra, W_rg, nclasses, gen)
    saved, reports = {}, {}
    tmp = tempfile.mkdtemp()
    torch.save(rg, os.path.join(tmp, "rg.pt"))

    def create_dataset(id, length, dim, overlap, full=False, flip=False):
        return ref_dataset.RGDataset(filepath=os.path.join(tmp, "rg.pt"), length=length, dim=dim, overlap=overlap, flip=flip)

    def get_reference(id, h, w, flip=False, length=None, dim=None, overlap=None):
        return nclasses, seg_full[:h, :].clone()

    calls = {"n": 0}
    orig_propagate = ref_main.propagate

    def propagate(seq, seg_ref, model, lp, ncls, do_pos_embed, use_last):
        pred, xent, change = orig_propagate(seq, seg_ref, model, lp, ncls, do_pos_embed, use_last=use_last)
        i = calls["n"]
        calls["n"] += 1
        if forced_change is not None and i < len(forced_change):
            change = forced_change[i]
        return pred, xent, change

    ref_main.create_dataset = create_dataset
    ref_main.get_reference = get_reference
    ref_main.create_model = lambda id, pos_embed: PatchFlatten()
    ref_main.load = lambda path: {}
    ref_main.plot = lambda **k: None
    ref_main.propagate = propagate
    ref_main.device_count = lambda: 1
    ref_main.classification_report = lambda gt, pred: reports.update(gt=gt.clone(), pred=pred.clone()) or ""
    ref_main.confusion_matrix = lambda gt, pred: ""
    _save = torch.save
    ref_main.torch.save = lambda obj, path: saved.update(map=obj.clone())
    args = argparse.Namespace(model=0, dataset=dataset_id, patch_size=hw, seq_length=T, overlap=(oh, 0),
                              cxt_size=cfg["CXT_SIZE"], radius=cfg["RADIUS"], temp=cfg["TEMP"], knn=cfg["KNN"],
                              model_path="", output_folder=tmp + "/", pos_embed=False, remove_unc=False, flip=False,
                              use_last=use_last, dataset_full=True, correction=correction)
    try:
        with cuda_is_cpu():
            ref_main.main(args)
    finally:
        torch.save = _save
    rows = saved["map"].shape[0]
    final = reports["pred"].reshape(rows, -1)
    np.savez(os.path.join(HERE, name + ".npz"), rg=rg.numpy(), seg=seg_full.numpy(), dataset_id=np.int32(dataset_id),
             nclasses=np.int32(nclasses), T=np.int32(T), patch=np.int32(hw), overlap=np.int32((oh, 0)),
             cxt_size=np.int32(cfg["CXT_SIZE"]), radius=np.int32(cfg["RADIUS"]), temp=np.float32(cfg["TEMP"]),
             knn=np.int32(cfg["KNN"]), use_last=np.bool_(use_last), correction=np.bool_(correction),
             forced_change=np.int32([-1 if f is None else f for f in (forced_change or [])]),
             saved_map=saved["map"].numpy(), final_map=final.numpy().astype(np.int8))
    changed = int((final.to(torch.int8) != saved["map"]).sum())
    print(f"{name}: N={N} map{tuple(saved['map'].shape)} classes={sorted(set(final.flatten().tolist()))} "
          f"pixels changed by the reverse merge={changed}")


def run_resnet_train_case(ref_model, ref_encoder, name, B, T, N, tau, seed, hw=(16, 16), oh=8, eval_after=False):
    """Reference CRW with its default encoder (Resnet, BatchNorm in train mode) on CPU: loss, A, features, a selection of
    parameter gradients (+ the norm of every gradient) and the BatchNorm running statistics after the step.  `hw` / `oh`: patch
    size and vertical overlap (32 x 32 / 24: scripts/test/test_mc1.py:19,21).  eval_after: also the features of the same patches
    from the encoder switched to eval mode after the step (`encoder.train(False)`, scripts/test/test.py:42: BatchNorm on the
    running statistics the step has just updated) and EVERY running statistic."""
    torch.manual_seed(seed)
    enc = ref_encoder.Resnet(False)
    enc.train(True)
    gen = torch.Generator().manual_seed(seed)
    h, w = hw
    rg = layered_radargram(N * (h - oh) + oh, 2 * T * w, gen)
    seq = torch.stack([unfold_items(rg, T, h, w, oh, 0, i * T) for i in range(B)])
    crw = ref_model.CRW(enc, tau, False)
    feats = {}

    def keep(m, i, o):  # the encoder's output and (below) the gradient that reaches it
        feats["emb"] = o.detach().clone()
        o.register_hook(lambda gr: feats.__setitem__("demb", gr.detach().clone()))

    hook = enc.register_forward_hook(keep)
    with cuda_is_cpu():
        loss, A = crw(seq)
    hook.remove()
    loss.backward()
    out = dict(seq=seq.numpy(), tau=np.float32(tau), seed=np.int32(seed), emb=feats["emb"].numpy(), A=A.detach().numpy(),
               loss=loss.detach().numpy(), demb=feats["demb"].numpy())
    keep = ("fc0.weight", "bn0.weight", "model.conv1.weight", "model.layer1.0.conv1.weight", "model.layer2.0.downsample.0.weight",
            "model.layer4.0.bn2.bias", "model.fc.weight", "model.fc.bias")
    names, norms = [], []
    for k, p in enc.named_parameters():
        names.append(k)
        norms.append(float(p.grad.double().norm()))
        if k in keep:
            out["grad." + k] = p.grad.numpy()
    out["grad_names"] = np.array(names)
    out["grad_norms"] = np.array(norms)
    out["bn0.running_mean"] = enc.bn0.running_mean.numpy()
    out["bn0.running_var"] = enc.bn0.running_var.numpy()
    out["model.bn1.running_mean"] = enc.model.bn1.running_mean.numpy()
    if eval_after:
        enc.train(False)
        with torch.no_grad():
            out["emb_eval"] = enc(seq.reshape(-1, 1, h, w)).numpy()
        for k, b in enc.named_buffers():
            if b.is_floating_point():
                out["buffer." + k] = b.numpy()
    np.savez(os.path.join(HERE, name + ".npz"), **out)
    print(f"{name}: loss={float(loss):.7f} seq{tuple(seq.shape)} {len(names)} parameter tensors")


def run_dataset_case(ref_dataset, name, H, W, length, dim, overlap, seed):
    import tempfile
    gen = torch.Generator().manual_seed(seed)
    rg = layered_radargram(H, W, gen)
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "rg.pt")
        torch.save(rg, p)
        ds = ref_dataset.RGDataset(filepath=p, length=length, dim=dim, overlap=overlap, flip=False)
        n = len(ds)
        picks = sorted(set([0, n // 2, n - 1]))
        items = np.stack([ds[i].numpy() for i in picks])
    np.savez(os.path.join(HERE, name + ".npz"), rg=rg.numpy(), length=np.int32(length), dim=np.int32(dim),
             overlap=np.int32(overlap), n_items=np.int32(n), picks=np.int32(picks), items=items)
    print(f"{name}: n_items={n} item{items.shape[1:]}")


def run_resnet_case(ref_encoder, name, seed):
    torch.manual_seed(seed)
    net = ref_encoder.Resnet(False)
    sd = net.state_dict()
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(6, 1, 16, 16, generator=gen)
    net.eval()
    with torch.no_grad():
        y = net(x)
    np.savez(os.path.join(HERE, name + ".npz"), keys=np.array(list(sd.keys())),
             sums=np.array([float(v.double().sum()) for v in sd.values()]), x=x.numpy(), y_eval=y.numpy())
    print(f"{name}: {len(sd)} keys, out{tuple(y.shape)}")


def main():
    torch.set_num_threads(8)
    ref_model, ref_encoder, ref_utils, ref_dataset, ref_lp = import_reference()
    only = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--only=")]  # --only=<fixture name prefix>: just those cases
    if only:
        g = globals()
        for fn in [k for k in g if k.startswith("run_") and k.endswith("_case")]:
            def gated(*a, _f=g[fn], **k):
                name = next(v for v in a if isinstance(v, str))
                if any(name.startswith(o) for o in only):
                    _f(*a, **k)
            g[fn] = gated
    # walk on fixed features: shapes of BASELINE configs 1-3 plus odd/edge cases
    run_walk_case(ref_model, "walk_cfg1_B2T8N7", 2, 8, 7, 128, 0.01, 0.25, 11)
    run_walk_case(ref_model, "walk_odd_B1T4N5", 1, 4, 5, 16, 0.1, 0.5, 12)
    run_walk_case(ref_model, "walk_onecycle_B3T3N6", 3, 3, 6, 8, 0.05, 0.5, 13)
    run_walk_case(ref_model, "walk_T2_nocycle", 1, 2, 4, 8, 0.05, 0.5, 14)
    run_walk_case(ref_model, "walk_cfg2_B1T16N63", 1, 16, 63, 128, 0.01, 0.30, 15)
    run_walk_case(ref_model, "walk_N70_B2T6", 2, 6, 70, 32, 0.07, 0.6, 16)
    run_walk_case(ref_model, "walk_cfg3_B1T32N63", 1, 32, 63, 128, 0.01, 0.30, 17)
    run_walk_case(ref_model, "walk_noise_B2T8N7_tau0p1", 2, 8, 7, 128, 0.1, 3.0, 18)
    # full path with the CNN encoder (weights stored beside)
    run_cnn_case(ref_model, ref_encoder, "cnn_cfg1_B2T8N7", 2, 8, 7, (16, 16), 0.01, False, 11, "cnn_weights_seed11")
    run_cnn_case(ref_model, ref_encoder, "cnn_posembed_B1T4N3", 1, 4, 3, (16, 16), 0.05, True, 21,
                 "cnn_weights_posembed_seed21")
    # label propagation (A.4): truncated context (quirk Q7), full context, reversed
    run_labelprop_case(ref_utils, ref_lp, "labelprop_trunc_T14N10", 14, 10, 16, 3,
                       dict(CXT_SIZE=4, RADIUS=3, TEMP=0.1, KNN=5), False, 31)
    run_labelprop_case(ref_utils, ref_lp, "labelprop_full_T40N48", 40, 48, 128, 4,
                       dict(CXT_SIZE=100, RADIUS=10, TEMP=0.1, KNN=20), False, 32)
    run_labelprop_case(ref_utils, ref_lp, "labelprop_last_T20N24", 20, 24, 64, 5,
                       dict(CXT_SIZE=8, RADIUS=6, TEMP=0.05, KNN=10), True, 33)
    run_labelprop_case(ref_utils, ref_lp, "labelprop_mc1_T100N12", 100, 12, 32, 4,
                       dict(CXT_SIZE=80, RADIUS=10, TEMP=0.01, KNN=10), False, 34)
    # 2-D node grids (never produced by a radargram -- the reference's frames are N x 1 -- but accepted by its predict())
    run_labelprop_grid_case(ref_lp, "labelprop_grid_T10_6x5", 10, 6, 5, 32, 3, dict(CXT_SIZE=3, RADIUS=2, TEMP=0.1, KNN=5), 35)
    run_labelprop_grid_case(ref_lp, "labelprop_grid_T8_4x9", 8, 4, 9, 16, 4, dict(CXT_SIZE=20, RADIUS=3, TEMP=0.05, KNN=7), 36)
    run_resnet_case(ref_encoder, "resnet_seed11", 11)
    run_resnet_train_case(ref_model, ref_encoder, "resnet_train_B2T4N5", 2, 4, 5, 0.05, 11)
    # the reference's cfg5-shaped encoder input: 32 x 32 patches, overlap (24, 0) (scripts/test/test_mc1.py:19,21) -- layer4's map is
    # 2 x 2 there, so the global average pool is no longer the identity; and a non-square size whose stem rows need two column tiles
    run_resnet_train_case(ref_model, ref_encoder, "resnet_train_32x32_B2T4N5", 2, 4, 5, 0.05, 12, hw=(32, 32), oh=24, eval_after=True)
    run_resnet_train_case(ref_model, ref_encoder, "resnet_train_20x27_B2T4N5", 2, 4, 5, 0.05, 13, hw=(20, 27), oh=10)
    # whole-radargram pipeline through the reference's own driver (scripts/test/test_all.py main)
    lpc = dict(CXT_SIZE=4, RADIUS=4, TEMP=0.1, KNN=5)
    run_segment_case(ref_dataset, "segment_ds0_reverse", 0, 4, 8, (8, 8), 4, 52, 3, 2, lpc, True, False, None, 51)
    run_segment_case(ref_dataset, "segment_ds1_reverse", 1, 6, 8, (8, 8), 4, 52, 3, 2, lpc, True, False, None, 52)
    run_segment_case(ref_dataset, "segment_ds3_reverse", 3, 5, 8, (8, 8), 4, 52, 3, 2, lpc, True, False, None, 53)
    run_segment_case(ref_dataset, "segment_ds0_correction", 0, 4, 8, (8, 8), 4, 52, 3, 2, lpc, False, True,
                     [3, None, 7], 54)
    run_segment_case(ref_dataset, "segment_ds3_correction_reverse", 3, 5, 8, (8, 8), 4, 52, 3, 2, lpc, True, True,
                     [None, 5, None], 55)
    # dataset unfold
    run_dataset_case(ref_dataset, "dataset_64x256", 64, 256, 8, (16, 16), (8, 0), 41)
    run_dataset_case(ref_dataset, "dataset_50x200_ow", 50, 200, 5, (12, 10), (4, 2), 42)


if __name__ == "__main__":
    main()
