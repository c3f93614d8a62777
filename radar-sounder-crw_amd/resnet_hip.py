"""``Resnet`` (the reference's default encoder, src/encoder.py:63-89 / :109-155 / :157-272) forward AND backward on the
hand-written HIP kernels of csrc/resnet_gemm.hip + csrc/resnet_bn.hip, through the C ABI (``crw_rn_*``).

Covered: float32 patches of ANY size on the GPU.  Train mode (``module.train()``: what scripts/train.py and most of the reference's
test scripts run -- they never call ``.eval()``): forward and backward, BatchNorm on batch statistics with the running-statistics
update.  Eval mode (``module.eval()`` / ``train(False)``: scripts/test/test.py:42): the forward on the running statistics under
``torch.no_grad()``.  Not covered (PyTorch ops, with a one-time warning): an eval-mode forward that autograd must differentiate,
BatchNorms that are individually frozen or differ in eps / momentum, ``momentum=None``.

Schedule of one training step for 16x16 patches (P patches; every convolution is a matrix product across patches, every
BatchNorm runs on batch statistics exactly like ``nn.BatchNorm2d`` in train mode and updates its running statistics; other patch
sizes: the stem on the gathered products instead of the patch-per-wave kernels, and where layer4's map keeps more than one pixel
-- 32x32 patches: 2x2 -- the average pool + head as one product over that map):

    stem     bn0 statistics from the moments of x; fc0 + bn0 + relu0 + 7x7/2 convolution, a patch per wave (the 3-channel map
             lives in LDS only) -> Z1 [P,81,64] + statistics                                 crw_rn_stem_stats, crw_rn_stem16_fwd, crw_rn_bn_stats_rows
    pool     relu(bn1(Z1)) -> 3x3/2 max-pool -> A1 [P,25,64]                                 crw_rn_bn_pool
    layer1-4 conv3x3 -> bn -> relu -> conv3x3 -> bn (+ 1x1/2 shortcut conv -> bn | identity) -> relu
                                                                                             crw_rn_conv mode 0, crw_rn_bn_stats, crw_rn_bn_apply
    head     global average pool of the 1x1 map (identity) + linear 512 -> 128               crw_rn_conv mode 0 (1x1) + bias

and the mirror image backwards (crw_rn_bn_bwd, crw_rn_conv mode 1, crw_rn_wgrad, crw_rn_pool_bwd, crw_rn_stem16_wgrad,
crw_rn_stem16_bwd).  No PyTorch / MIOpen convolution or batch-norm call is made on this path.
"""
import weakref

import torch

import crw_hip as H


def _out(n, k, s, p):
    return (n + 2 * p - k) // s + 1


def final_map(h, w):
    """layer4's map for h x w patches: 1x1 conv with padding 1, 7x7/2, max-pool 3x3/2, strides 1, 2, 2, 2"""
    hh, ww = _out(_out(h + 2, 7, 2, 3), 3, 2, 1), _out(_out(w + 2, 7, 2, 3), 3, 2, 1)
    for _ in range(3):
        hh, ww = _out(hh, 3, 2, 1), _out(ww, 3, 2, 1)
    return hh, ww


def supported(x, net):
    """What the HIP path covers (module docstring).  Everything the native pass assumes about the module is checked here: the input
    channels fc0 was built for (a mismatch must reach PyTorch's shape error, not an out-of-bounds read of fc0.weight), ONE mode, eps
    and momentum for all 13 BatchNorms (the C ABI takes one of each), a final map the head product can cover."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == net.fc0.in_channels and x.shape[1] in (1, 2)):
        return False
    if net.fc0.weight.dtype != torch.float32:
        return False
    hl, wl = final_map(*x.shape[-2:])
    if hl * wl > 64 or _out(x.shape[-2] + 2, 7, 2, 3) * _out(x.shape[-1] + 2, 7, 2, 3) > 4096:
        return False
    bns = _bn_modules(net, fresh=True)
    if len(bns) != H.RN_NBN or any(m.training != net.training or m.eps != net.bn0.eps or m.momentum != net.bn0.momentum
                                   or not m.track_running_stats or not m.affine for m in bns):
        return False
    if net.training:
        return net.bn0.momentum is not None
    return not torch.is_grad_enabled()  # eval mode: the forward only


def check_batch(x):
    """nn.BatchNorm2d refuses a training batch with ONE value per channel (torch.nn.functional._verify_batch_size): layer4's map is
    1 x 1 at 16x16 patches, so a single patch raises there in the reference -- same error here, before any launch."""
    hl, wl = final_map(*x.shape[-2:])
    if x.shape[0] * hl * wl == 1:
        raise ValueError(f"Expected more than 1 value per channel when training, got input size {torch.Size([1, 512, 1, 1])}")


class _Block:
    """geometry + modules of one BasicBlock"""

    def __init__(self, mod, hin, win):
        self.mod = mod
        self.cin, self.cout = mod.conv1.in_channels, mod.conv1.out_channels
        self.stride = mod.conv1.stride[0]
        self.hin, self.win = hin, win
        self.hout, self.wout = _out(hin, 3, self.stride, 1), _out(win, 3, self.stride, 1)


_BN_MEMO = [None, None]  # (weak reference to net, its BatchNorm modules) of the latest supported() call


def _bn_modules(net, fresh=False):
    """the 13 BatchNorm modules in module order.  supported() walks the module tree (fresh=True) and leaves the list for the forward
    that follows it in the same call of Resnet.forward -- the walk is 0.07 ms of host time, twice per step without this"""
    if not fresh and _BN_MEMO[0] is not None and _BN_MEMO[0]() is net:
        bns = _BN_MEMO[1]
        _BN_MEMO[0] = _BN_MEMO[1] = None
        return bns
    bns = [m for m in net.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    if fresh:
        _BN_MEMO[0], _BN_MEMO[1] = weakref.ref(net), bns
    return bns


class HipResnetNative(torch.autograd.Function):
    """The training path: crw_rn_train_fwd / crw_rn_train_bwd run the schedule above from native code on one workspace (driven
    launch by launch from Python -- ``HipResnetFn`` below, which the tests keep as the readable statement of the schedule -- the
    ~230 launches of a step make the step host-bound)."""

    @staticmethod
    def forward(ctx, x, net, *params):
        check_batch(x)
        x = x.contiguous()
        bns = _bn_modules(net)
        prm = [p.detach().contiguous() for p in params]
        out, ws = H.rn_train_fwd(x, prm, [m.running_mean for m in bns], [m.running_var for m in bns], net.bn0.momentum, net.bn0.eps)
        torch._foreach_add_([m.num_batches_tracked for m in bns], 1)
        ctx.x, ctx.ws, ctx.prm = x, ws, prm
        ctx.prepared = H.rn_grad_views(prm)  # (now, while the host is ahead of the GPU: see there)
        return out

    @staticmethod
    def backward(ctx, dout):
        if ctx.needs_input_grad[0]:
            raise RuntimeError("the HIP Resnet path does not produce a gradient for its input patches (the reference never asks for one)")
        grads = H.rn_train_bwd(dout.contiguous().float(), ctx.x, ctx.prm, ctx.ws, ctx.prepared)
        ctx.ws = ctx.prepared = None
        return (None, None) + tuple(grads)


def nograd_forward(x, net):
    """Train-mode BatchNorm under ``torch.no_grad()`` -- how the reference's test scripts run the encoder (scripts/test/test_mc1.py:
    40-46 never calls .eval(); src/utils.py:108): crw_rn_train_fwd without what only a backward pass would read."""
    check_batch(x)
    bns = _bn_modules(net)
    out, _ = H.rn_train_fwd(x.contiguous(), [p.detach().contiguous() for p in net.parameters()], [m.running_mean for m in bns],
                            [m.running_var for m in bns], net.bn0.momentum, net.bn0.eps, keep=False)
    torch._foreach_add_([m.num_batches_tracked for m in bns], 1)
    return out


def eval_forward(x, net):
    """``net.eval()`` under ``torch.no_grad()``: the same launches with every BatchNorm on its running statistics (crw_rn_eval_fwd)"""
    bns = _bn_modules(net)
    return H.rn_eval_fwd(x.contiguous(), [p.detach().contiguous() for p in net.parameters()], [m.running_mean for m in bns],
                         [m.running_var for m in bns], net.bn0.eps)


class HipResnetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, net, *params):
        check_batch(x)
        x = x.contiguous()
        P, cin, h, w = x.shape
        body = net.model
        mom = net.bn0.momentum
        H0, W0 = h + 2, w + 2
        H1, W1 = _out(H0, 7, 2, 3), _out(W0, 7, 2, 3)
        H2, W2 = _out(H1, 3, 2, 1), _out(W1, 3, 2, 1)
        Hm, Wm = max(H0 + 6, 2 * H1 + 6), max(W0 + 6, 2 * W1 + 6)
        blocks, hh, ww = [], H2, W2
        for i in range(1, 5):
            b = _Block(getattr(body, f"layer{i}")[0], hh, ww)
            blocks.append(b)
            hh, ww = b.hout, b.wout
        hl, wl = hh, ww  # layer4's map: the head averages over it (src/encoder.py:264-266)
        sv = {"geo": (P, cin, h, w, H0, W0, H1, W1, H2, W2, Hm, Wm), "x": x, "blocks": blocks, "head": (hl, wl)}

        # ---- stem: 16 x 16 patches on the patch-per-wave kernels (no map in HBM), other sizes on the gathered product
        if (h, w) == (16, 16):
            wstem = H.rn_pack_stem16(body.conv1.weight)
            stem = H.rn_stem_stats(x, net.fc0, net.bn0, mom)
            Z1, part = H.rn_stem16_fwd(x, stem, wstem[0])
            coef1 = H.rn_bn_stats_rows(part, P * H1 * W1, body.bn1, mom)
            xmap = None
        else:
            wstem = H.rn_pack_stem(body.conv1.weight, h, w)
            xmap, stem = H.rn_stem_fwd(x, net.fc0, net.bn0, Hm, Wm, mom)
            if H.rn_stem_band_ok(h, w):  # like the native pass: the forward product on the band-per-wave kernel (csrc/resnet_stem.hip)
                Z1, part = H.rn_stem_band_fwd(x, stem, H.rn_pack_stem16(body.conv1.weight)[0], H1, W1)
                coef1 = H.rn_bn_stats_rows(part, P * H1 * W1, body.bn1, mom)
            else:
                Z1, part = H.rn_conv(H.RN_STEM_FWD, P, (Hm, Wm, 4), (H1, W1), 64, (7, 7), 2, 3, xmap, wstem[:2], stats=True)
                coef1 = H.rn_bn_stats(part, P, H1 * W1, body.bn1, mom)
        A, amax = H.rn_bn_pool(Z1, coef1, P, H1, W1, 64)
        sv.update(wstem=wstem, xmap=xmap, stem=stem, Z1=Z1, coef1=coef1, amax=amax)

        # ---- residual stages
        recs = []
        for b in blocks:
            m = b.mod
            r = {"Ain": A}
            r["wa"] = H.rn_pack_conv(m.conv1.weight)
            r["wb"] = H.rn_pack_conv(m.conv2.weight)
            npix = b.hout * b.wout
            Za, part = H.rn_conv(H.RN_FWD, P, (b.hin, b.win, b.cin), (b.hout, b.wout), b.cout, (3, 3), b.stride, 1, A, r["wa"][:2],
                                 stats=True)
            r["Za"], r["ca"] = Za, H.rn_bn_stats(part, P, npix, m.bn1, mom)
            r["Aa"] = H.rn_bn_apply(Za, r["ca"], P, npix, b.cout)
            Zb, part = H.rn_conv(H.RN_FWD, P, (b.hout, b.wout, b.cout), (b.hout, b.wout), b.cout, (3, 3), 1, 1, r["Aa"], r["wb"][:2],
                                 stats=True)
            r["Zb"], r["cb"] = Zb, H.rn_bn_stats(part, P, npix, m.bn2, mom)
            if m.downsample is not None:
                r["wd"] = H.rn_pack_conv(m.downsample[0].weight)
                Zd, part = H.rn_conv(H.RN_FWD, P, (b.hin, b.win, b.cin), (b.hout, b.wout), b.cout, (1, 1), b.stride, 0, A, r["wd"][:2],
                                     stats=True)
                r["Zd"], r["cd"] = Zd, H.rn_bn_stats(part, P, npix, m.downsample[1], mom)
                A = H.rn_bn_apply(Zb, r["cb"], P, npix, b.cout, Zd=Zd, coef_d=r["cd"])
            else:
                A = H.rn_bn_apply(Zb, r["cb"], P, npix, b.cout, res=r["Ain"])
            r["Aout"] = A
            recs.append(r)

        # ---- head: global average pool + linear layer as ONE product over layer4's map -- a "convolution" whose kernel covers the
        # map, every tap holding fc.weight / npix (the average pool of a 1x1 map is the identity: the plain linear layer)
        wrep = (body.fc.weight.detach() / (hl * wl))[:, :, None, None].expand(-1, -1, hl, wl).contiguous()
        wfc = H.rn_pack_conv(wrep)
        out, _ = H.rn_conv(H.RN_FWD, P, (hl, wl, 512), (1, 1), body.fc.out_features, (hl, wl), 1, 0, A, wfc[:2], bias=body.fc.bias.detach())
        sv.update(recs=recs, wfc=wfc, Alast=A)
        bns = [m for m in net.modules() if isinstance(m, torch.nn.BatchNorm2d) and m.num_batches_tracked is not None]
        torch._foreach_add_([m.num_batches_tracked for m in bns], 1)
        ctx.sv = sv
        ctx.net = net
        ctx.names = [k for k, _ in net.named_parameters()]
        return out[:P]

    @staticmethod
    def backward(ctx, dout):
        if ctx.needs_input_grad[0]:
            raise RuntimeError("the HIP Resnet path does not produce a gradient for its input patches (the reference never asks for one)")
        sv, net = ctx.sv, ctx.net
        body = net.model
        P, cin, h, w, H0, W0, H1, W1, H2, W2, Hm, Wm = sv["geo"]
        grads = {}
        dout = dout.contiguous().float()
        dO = H.rn_split(dout, P, dout.shape[1])
        nout = dout.shape[1]
        hl, wl = sv["head"]
        dwrep = H.rn_wgrad(H.RN_FWD, P, (hl, wl, 512), (1, 1, nout), (hl, wl), 1, 0, sv["Alast"], dO)  # per pixel of the averaged map
        grads["model.fc.weight"] = dwrep.reshape(nout, 512, hl * wl).sum(-1) / (hl * wl)
        grads["model.fc.bias"] = H.rn_colsum(dout)
        g1, _ = H.rn_conv(H.RN_BWD, P, (1, 1, nout), (hl, wl), 512, (hl, wl), 1, 0, dO, sv["wfc"][2:])
        g2 = None
        for i in (3, 2, 1, 0):
            b, r = sv["blocks"][i], sv["recs"][i]
            m = b.mod
            pre = f"model.layer{i + 1}.0."
            npix = b.hout * b.wout
            has_d = m.downsample is not None
            dzb, dzd, gres, dg, db, dgd, dbd = H.rn_bn_bwd(g1, g2, r["Aout"][0], r["Zb"], r["cb"], P, npix, b.cout,
                                                          Zd=r.get("Zd"), coef_d=r.get("cd"), want_g=not has_d)
            grads[pre + "bn2.weight"], grads[pre + "bn2.bias"] = dg, db
            grads[pre + "conv2.weight"] = H.rn_wgrad(H.RN_FWD, P, (b.hout, b.wout, b.cout), (b.hout, b.wout, b.cout), (3, 3), 1, 1,
                                                     r["Aa"], dzb)
            gA, _ = H.rn_conv(H.RN_BWD, P, (b.hout, b.wout, b.cout), (b.hout, b.wout), b.cout, (3, 3), 1, 1, dzb, r["wb"][2:])
            dza, _, _, dg, db, _, _ = H.rn_bn_bwd(gA, None, r["Aa"][0], r["Za"], r["ca"], P, npix, b.cout)
            grads[pre + "bn1.weight"], grads[pre + "bn1.bias"] = dg, db
            grads[pre + "conv1.weight"] = H.rn_wgrad(H.RN_FWD, P, (b.hin, b.win, b.cin), (b.hout, b.wout, b.cout), (3, 3), b.stride, 1,
                                                     r["Ain"], dza)
            g1, _ = H.rn_conv(H.RN_BWD, P, (b.hout, b.wout, b.cout), (b.hin, b.win), b.cin, (3, 3), b.stride, 1, dza, r["wa"][2:])
            if has_d:
                grads[pre + "downsample.1.weight"], grads[pre + "downsample.1.bias"] = dgd, dbd
                grads[pre + "downsample.0.weight"] = H.rn_wgrad(H.RN_FWD, P, (b.hin, b.win, b.cin), (b.hout, b.wout, b.cout), (1, 1),
                                                                b.stride, 0, r["Ain"], dzd)
                g2, _ = H.rn_conv(H.RN_BWD, P, (b.hout, b.wout, b.cout), (b.hin, b.win), b.cin, (1, 1), b.stride, 0, dzd, r["wd"][2:])
            else:
                g2 = gres
        # max-pool + bn1, stem convolution, stem
        dz1, dg, db = H.rn_pool_bwd(g1, g2, sv["amax"], sv["Z1"], sv["coef1"], P, H1, W1, 64)
        grads["model.bn1.weight"], grads["model.bn1.bias"] = dg, db
        if sv["xmap"] is None:
            grads["model.conv1.weight"] = H.rn_stem16_wgrad(sv["x"], sv["stem"], dz1)
            dw0, db0, dg, db = H.rn_stem16_bwd(sv["x"], sv["stem"], net.fc0.weight.detach(), net.fc0.bias.detach(), sv["wstem"][1], dz1)
        else:
            grads["model.conv1.weight"] = H.rn_wgrad(H.RN_STEM_FWD, P, (Hm, Wm, 4), (H1, W1, 64), (7, 7), 2, 3, sv["xmap"], dz1)
            dX0, _ = H.rn_conv(H.RN_STEM_BWD, P, (H1, W1, 64), (H0, 1), H.rn_stem_cols(w), (7, 7), 2, 3, dz1, sv["wstem"][2:])
            dw0, db0, dg, db = H.rn_stem_bwd(dX0, sv["x"], sv["stem"], net.fc0.weight.detach(), net.fc0.bias.detach())
        grads["fc0.weight"], grads["fc0.bias"], grads["bn0.weight"], grads["bn0.bias"] = dw0, db0, dg, db
        ctx.sv = None
        return (None, None) + tuple(grads[k] for k in ctx.names)
