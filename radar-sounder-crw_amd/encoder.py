"""Patch encoders of the CRW model -- same class surface and ``state_dict`` keys as the reference
(``CNN(pos_embed)``: src/encoder.py:9-57, ``Resnet(pos_embed, pretrained)``: src/encoder.py:63-89)
so checkpoints are interchangeable and, with the same ``torch.manual_seed``, the initial weights
are identical (layers are constructed in the same order, so the RNG stream is consumed alike).

Arithmetic: on an MI355X, for 16x16 patches, the whole conv trunk of ``CNN`` -- the fused front end
conv1-ReLU-pool-conv2-ReLU-pool (csrc/encoder_front.hip), the 3x3 layers conv3/conv4/conv5 (+ReLU)
and the global average pool (csrc/encoder_conv.hip), forward AND backward -- runs in hand-written
HIP kernels (``CNN.hip_convs``: "bf16x3" = hi/lo bf16 operand pairs on the matrix cores, fp32-grade
results, the default; "bf16" = plain bf16 operands; None = PyTorch-ROCm ops).  Only the linear
head stays a PyTorch op.  At other patch sizes the 3x3 trunk runs on the tiled variants of the same kernels (10x10 output
tiles over the feature map): inference (``torch.no_grad``) the whole trunk incl. the front end (``_hip_inference_trunk``);
training likewise, forward and backward (``_HipMapEncoder``).  ``Resnet`` (16x16 patches, train mode) runs forward and backward on
its own hand-written kernels (``resnet_hip``).  CPU tensors use PyTorch ops (a CUDA batch that misses the HIP path warns once).
"""
import warnings

import torch
import torch.nn as nn
import torch.nn.functional as TF

FEATURE_DIM = 128


class _HipEncoder(torch.autograd.Function):
    """The whole conv trunk of ``CNN`` on the HIP kernels: fused front end (conv1-ReLU-pool-conv2-ReLU-
    pool), conv3/conv4/conv5 (+ReLU) and the global average pool.  x [P,cin,16,16] fp32 -> [P,128] fp32.
    Saved for backward: the input patches and the bf16 activation planes of conv2..conv5 outputs
    and the front end's record (pool1 planes + pooling codes), from which its backward kernel runs without recomputation."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, w3, b3, w4, b4, w5, b5, split, bwd_split=None):
        import crw_hip
        x = x.contiguous()
        w2p = crw_hip.enc_front_pack(w2, split)
        packed = [crw_hip.enc_pack_weights(w, split) for w in (w3, w4, w5)]
        ctx.bwd_split = bwd_split or split  # "mixed": forward on hi/lo pairs, backward on the hi planes alone
        # the front end also keeps its pool1 planes and pooling codes (10 KB per patch): its backward kernel then skips the
        # conv1 -> pool1 -> conv2 recomputation
        x3h, x3l, fsaved = crw_hip.enc_front_fwd(split, x, w1, b1, w2p[:2], b2, save=True)
        y3h, y3l, _, _ = crw_hip.enc_conv3x3(0, split, x3h, x3l, packed[0][0], packed[0][1], 64, bias=b3)
        y4h, y4l, _, _ = crw_hip.enc_conv3x3(0, split, y3h, y3l, packed[1][0], packed[1][1], 128, bias=b4)
        y5h, _, _, gap = crw_hip.enc_conv3x3(0, split, y4h, y4l, packed[2][0], packed[2][1], 128, bias=b5, gap=True,
                                             lo_plane=False)  # only the sign of y5 is needed later
        ctx.split = split
        # lo planes are None for plain bf16; everything goes through save_for_backward (in-place weight updates between
        # forward and backward are detected, a second backward without retain_graph raises autograd's own error)
        ctx.save_for_backward(x, w1.detach(), b1.detach(), b2.detach(), *w2p, x3h, x3l, y3h, y3l, y4h, y4l, y5h,
                              *[t for pk in packed for t in (pk[2], pk[3])], fsaved)
        return gap

    @staticmethod
    def backward(ctx, dgap):
        import crw_hip
        s = ctx.bwd_split
        if ctx.needs_input_grad[0]:
            raise RuntimeError("the fused HIP encoder does not produce a gradient for its input patches (the reference never "
                               "asks for one); set CNN.hip_convs = None to differentiate with respect to the input")
        sv = ctx.saved_tensors
        x, w1, b1, b2 = sv[:4]
        w2p = sv[4:8]
        x3h, x3l, y3h, y3l, y4h, y4l, y5h = sv[8:15]
        bwd_w = [(sv[15], sv[16]), (sv[17], sv[18]), (sv[19], sv[20])]
        fsaved = sv[21]
        if s == 1:  # the plain-bf16 kernels read the hi planes only
            x3l = y3l = y4l = None
            bwd_w = [(h, None) for h, _ in bwd_w]
            w2p = (w2p[0], None, w2p[2], None)
        # dY5 = dgap/100 gated by y5 > 0 (ReLU5 + GAP backward) is built inside the two kernels' loaders
        dgap = dgap.contiguous().float()
        dw5, db5 = crw_hip.enc_wgrad(s, y5h, None, y4h, y4l, dgap=dgap)
        d4h, d4l, _, _ = crw_hip.enc_conv3x3(1, s, y5h, None, *bwd_w[2], 128, mask=y4h, dgap=dgap)  # dY4
        dw4, db4 = crw_hip.enc_wgrad(s, d4h, d4l, y3h, y3l)
        d3h, d3l, _, _ = crw_hip.enc_conv3x3(1, s, d4h, d4l, *bwd_w[1], 64, mask=y3h)    # dY3
        dw3, db3 = crw_hip.enc_wgrad(s, d3h, d3l, x3h, x3l)
        _, _, dx3, _ = crw_hip.enc_conv3x3(1, s, d3h, d3l, *bwd_w[0], 32, planes=False, f32=True)  # [P,100,32]
        dw1, db1, dw2, db2 = crw_hip.enc_front_bwd(s, x, w1, b1, w2p[:2], b2, w2p[2:], dx3, saved=fsaved)
        return None, dw1, db1, dw2, db2, dw3, db3, dw4, db4, dw5, db5, None, None

class _HipMapEncoder(torch.autograd.Function):
    """The whole conv trunk of ``CNN`` on patches of ANY size (training at patch sizes other than 16x16, e.g. the 32x32
    patches of BASELINE config 5), forward and backward on the tiled HIP kernels (10x10 output tiles over the feature map):
    front end `crw_enc_front_fwd_map` / `crw_enc_front_bwd_map`, conv3-5 `crw_enc_conv3x3_map` (mode 0 forward, mode 1
    backward-data) / `crw_enc_conv3x3_wgrad_map`, ReLU + average-pool backward `crw_enc_gap_bwd`.
    x [P,cin,h,w] fp32 -> pooled features [P,128]."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, w3, b3, w4, b4, w5, b5, split):
        import crw_hip
        x = x.contiguous()
        P, _, h, w = x.shape
        H, W = h - 6, w - 6
        w2p = crw_hip.enc_front_pack(w2, split)
        packed = [crw_hip.enc_pack_weights(wt, split) for wt in (w3, w4, w5)]
        x3h, x3l = crw_hip.enc_front_fwd_map(split, x, w1, b1, w2p[:2], b2)
        y3h, y3l, _ = crw_hip.enc_conv3x3_map(split, x3h, x3l, packed[0][0], packed[0][1], 64, H, W, bias=b3)
        y4h, y4l, _ = crw_hip.enc_conv3x3_map(split, y3h, y3l, packed[1][0], packed[1][1], 128, H, W, bias=b4)
        y5h, _, gap = crw_hip.enc_conv3x3_map(split, y4h, y4l, packed[2][0], packed[2][1], 128, H, W, bias=b5, gap=True,
                                              lo_plane=False)  # only the sign of y5 is needed later
        ctx.split, ctx.hw = split, (H, W)
        ctx.save_for_backward(x, w1.detach(), b1.detach(), b2.detach(), *w2p, x3h, x3l, y3h, y3l, y4h, y4l, y5h,
                              *[t for pk in packed for t in (pk[2], pk[3])])
        return gap

    @staticmethod
    def backward(ctx, dgap):
        import crw_hip
        if ctx.needs_input_grad[0]:
            raise RuntimeError("the fused HIP encoder does not produce a gradient for its input patches (the reference never "
                               "asks for one); set CNN.hip_convs = None to differentiate with respect to the input")
        s, (H, W) = ctx.split, ctx.hw
        sv = ctx.saved_tensors
        x, w1, b1, b2 = sv[:4]
        w2p = sv[4:8]
        x3h, x3l, y3h, y3l, y4h, y4l, y5h = sv[8:15]
        bw = [(sv[15], sv[16]), (sv[17], sv[18]), (sv[19], sv[20])]
        d5h, d5l = crw_hip.enc_gap_bwd(dgap.contiguous().float(), y5h, s)               # ReLU5 + GAP backward
        dw5, db5 = crw_hip.enc_wgrad_map(s, d5h, d5l, y4h, y4l, H, W)
        d4h, d4l, _ = crw_hip.enc_conv3x3_map(s, d5h, d5l, *bw[2], 128, H, W, mode=1, mask=y4h)
        dw4, db4 = crw_hip.enc_wgrad_map(s, d4h, d4l, y3h, y3l, H, W)
        d3h, d3l, _ = crw_hip.enc_conv3x3_map(s, d4h, d4l, *bw[1], 64, H, W, mode=1, mask=y3h)
        dw3, db3 = crw_hip.enc_wgrad_map(s, d3h, d3l, x3h, x3l, H, W)
        _, _, dx3 = crw_hip.enc_conv3x3_map(s, d3h, d3l, *bw[0], 32, H, W, mode=1, planes=False, f32=True)  # [P, H*W, 32]
        dw1, db1, dw2, db2 = crw_hip.enc_front_bwd_map(s, x, w1, b1, w2p[:2], b2, w2p[2:], dx3)
        return None, dw1, db1, dw2, db2, dw3, db3, dw4, db4, dw5, db5, None


class _HipLinear(torch.autograd.Function):
    """The 128 -> 128 head.  Forward and dX are PyTorch matmuls; the weight gradient dW = dy^T x (M = N = 128, K = P)
    runs on `crw_linear128_wgrad`: hipBLASLt gives that shape 16 workgroups (~100 us at P = 16128), the split over P
    keeps every CU busy (~25 us) and adds the partial matrices in a fixed order."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return TF.linear(x, w, b)

    @staticmethod
    def backward(ctx, dy):
        import crw_hip
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        return dy @ w, crw_hip.linear128_wgrad(dy, x), dy.sum(0)


# (name, out_channels, kernel, followed by 2x2/stride-1 max-pool?)
_CNN_STACK = (("1", 8, 5, True), ("2", 32, 5, True), ("3", 64, 3, False), ("4", 128, 3, False), ("5", 128, 3, False))


def _report(module):
    n = sum(p.numel() for p in module.parameters() if p.requires_grad)
    print(f"Number of trainable parameters: {n}")
    return n


class CNN(nn.Module):
    """Five conv layers (5x5, 5x5, 3x3, 3x3, 3x3; all padding 1) with ReLU, stride-1 max-pools
    after the first two, global average pool and a linear head -> 128-d feature per patch."""
    _warned_fallback = False  # the PyTorch-op path on a GPU warns once per process

    def __init__(self, pos_embed):
        super().__init__()
        cin = 2 if pos_embed else 1
        for name, cout, k, pooled in _CNN_STACK:
            setattr(self, "conv" + name, nn.Conv2d(cin, cout, kernel_size=k, padding=1))
            setattr(self, "relu" + name, nn.ReLU())
            if pooled:
                setattr(self, "pool" + name, nn.MaxPool2d(kernel_size=2, stride=1))
            cin = cout
        self.global_avg_pool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(cin, FEATURE_DIM)
        self.num_params = _report(self)
        # "bf16x3" (default): hi/lo bf16 operand pairs, fp32-grade forward AND backward | "mixed" (opt-in, 16x16 patches): the same
        # forward, backward on the hi planes alone (plain bf16 operands, fp32 accumulate: gradients to ~1e-2) | "bf16": plain bf16
        # operands both ways | None: PyTorch ops.  Only used on an MI355X
        self.hip_convs = "bf16x3"

    def forward(self, x):
        if self.hip_convs and x.is_cuda and x.shape[-2:] == (16, 16) and x.dtype == torch.float32:
            c = [getattr(self, "conv%d" % i) for i in range(1, 6)]
            gap = _HipEncoder.apply(x, c[0].weight, c[0].bias, c[1].weight, c[1].bias, c[2].weight, c[2].bias,
                                    c[3].weight, c[3].bias, c[4].weight, c[4].bias,
                                    3 if self.hip_convs in ("bf16x3", "mixed") else 1, 1 if self.hip_convs == "mixed" else None)
            return self._head(gap)
        if self.hip_convs and x.is_cuda and x.dtype == torch.float32 and min(x.shape[-2:]) >= 7:
            if not torch.is_grad_enabled():
                return self._head(self._hip_inference_trunk(x))
            # training at another patch size: the whole trunk forward AND backward on the tiled HIP kernels
            c = [getattr(self, "conv%d" % i) for i in range(1, 6)]
            gap = _HipMapEncoder.apply(x, c[0].weight, c[0].bias, c[1].weight, c[1].bias, c[2].weight, c[2].bias,
                                       c[3].weight, c[3].bias, c[4].weight, c[4].bias,
                                       3 if self.hip_convs in ("bf16x3", "mixed") else 1)
            return self._head(gap)
        if self.hip_convs and x.is_cuda and not CNN._warned_fallback:
            # not silent: the caller believes it is on the hand-written kernels (set hip_convs = None to choose this path)
            CNN._warned_fallback = True
            warnings.warn(f"CNN.forward: input {tuple(x.shape)} {x.dtype} (grad enabled: {torch.is_grad_enabled()}) is not covered by "
                          "the HIP conv kernels (float32 patches of at least 7x7 on an MI355X): this call "
                          "runs on PyTorch-ROCm / MIOpen convolutions", RuntimeWarning, stacklevel=2)
        for name, _, _, pooled in _CNN_STACK:
            x = getattr(self, "relu" + name)(getattr(self, "conv" + name)(x))
            if pooled:
                x = getattr(self, "pool" + name)(x)
        return self.fc(self.global_avg_pool(x).flatten(1))

    def _head(self, gap):
        if gap.shape[0] % 128 == 0 and gap.shape[0] >= 128 and self.fc.weight.shape == (128, 128) and self.fc.bias is not None:
            return _HipLinear.apply(gap, self.fc.weight, self.fc.bias)
        return self.fc(gap)

    def _hip_inference_trunk(self, x):
        """Inference (no autograd) at patch sizes other than 16x16 -- e.g. the 32x32 patches of BASELINE config 5:
        the whole conv trunk on the tiled HIP kernels -- fused conv1-ReLU-pool-conv2-ReLU-pool (`crw_enc_front_fwd_map`),
        then conv3/conv4/conv5 (+ReLU) and the global average pool (`crw_enc_conv3x3_map`), all on 10x10 output
        tiles over the feature map.
        x [P,cin,h,w] -> pooled features [P,128]."""
        import crw_hip
        split = 3 if self.hip_convs in ("bf16x3", "mixed") else 1
        P, _, h, w = x.shape
        H, W = h - 6, w - 6  # conv3-5 feature map
        w2p = crw_hip.enc_front_pack(self.conv2.weight, split)
        xh, xl = crw_hip.enc_front_fwd_map(split, x, self.conv1.weight, self.conv1.bias, w2p[:2], self.conv2.bias)
        pk = [crw_hip.enc_pack_weights(getattr(self, "conv" + n).weight, split) for n in ("3", "4", "5")]
        y3h, y3l, _ = crw_hip.enc_conv3x3_map(split, xh, xl, pk[0][0], pk[0][1], 64, H, W, bias=self.conv3.bias)
        y4h, y4l, _ = crw_hip.enc_conv3x3_map(split, y3h, y3l, pk[1][0], pk[1][1], 128, H, W, bias=self.conv4.bias)
        _, _, gap = crw_hip.enc_conv3x3_map(split, y4h, y4l, pk[2][0], pk[2][1], 128, H, W, bias=self.conv5.bias,
                                            planes=False, gap=True)
        return gap


class _Residual(nn.Module):
    """conv3x3-BN-ReLU-conv3x3-BN + shortcut, ReLU (ResNet 'basic' block)."""

    def __init__(self, cin, cout, stride, shortcut):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = shortcut

    def forward(self, x):
        y = self.bn2(self.conv2(self.relu(self.bn1(self.conv1(x)))))
        return self.relu(y + (x if self.downsample is None else self.downsample(x)))


class _ResNetBody(nn.Module):
    """7x7/2 stem, 3x3/2 max-pool, four one-block stages (64,128,256,512; strides 1,2,2,2),
    global average pool, linear head."""

    def __init__(self, out_dim):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        cin = 64
        for i, (cout, stride) in enumerate(((64, 1), (128, 2), (256, 2), (512, 2)), start=1):
            shortcut = None
            if stride != 1 or cin != cout:  # built before the block, like the reference (RNG order)
                shortcut = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))
            setattr(self, f"layer{i}", nn.Sequential(_Residual(cin, cout, stride, shortcut)))
            cin = cout
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(cin, out_dim)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


class Resnet(nn.Module):
    """1x1 conv (padding 1) + BN + ReLU lifting 1-2 channels to 3, then the ResNet body.

    On an MI355X, fp32 patches of any size run on the hand-written HIP kernels (``resnet_hip``: every convolution as a matrix-core
    product across patches, BatchNorm statistics from the convolutions' epilogues): in train mode (scripts/train.py, and the test
    scripts that never call ``.eval()``: BatchNorm on batch statistics) forward AND backward; in eval mode (scripts/test/test.py:42
    ``encoder.train(False)``: BatchNorm on the running statistics) the forward under ``torch.no_grad()``.  ``hip_convs = None``
    selects the PyTorch ops, which also serve CPU tensors (host tests).  A CUDA batch that misses the HIP path (``resnet_hip.
    supported``) warns once."""
    _warned_fallback = False

    def __init__(self, pos_embed=True, pretrained=None):  # `pretrained` is ignored by the reference too
        super().__init__()
        self.fc0 = nn.Conv2d(2 if pos_embed else 1, 3, kernel_size=1, padding=1)
        self.bn0 = nn.BatchNorm2d(3)
        self.relu0 = nn.ReLU(inplace=True)
        self.model = _ResNetBody(FEATURE_DIM)
        self.num_params = _report(self)
        # "bf16x3": hi/lo bf16 operand pairs (fp32-grade), the whole pass driven from native code; "stepwise": the same kernels
        # launched one by one from Python (tests); None: PyTorch ops
        self.hip_convs = "bf16x3"

    def forward(self, x):
        if self.hip_convs and x.is_cuda:
            import resnet_hip
            if resnet_hip.supported(x, self):
                if not self.training:
                    return resnet_hip.eval_forward(x, self)
                if not torch.is_grad_enabled() and self.hip_convs != "stepwise":  # train-mode BatchNorm, no backward to follow
                    return resnet_hip.nograd_forward(x, self)
                fn = resnet_hip.HipResnetFn if self.hip_convs == "stepwise" else resnet_hip.HipResnetNative
                return fn.apply(x, self, *self.parameters())
            if not Resnet._warned_fallback:
                Resnet._warned_fallback = True
                warnings.warn(f"Resnet.forward: input {tuple(x.shape)} {x.dtype} (training: {self.training}, grad enabled: "
                              f"{torch.is_grad_enabled()}) is not covered by the HIP kernels (resnet_hip.supported: float32 patches on "
                              "an MI355X, train mode, or eval mode under no_grad, uniform BatchNorms): this call runs on PyTorch-ROCm / "
                              "MIOpen", RuntimeWarning, stacklevel=2)
        return self.model(self.relu0(self.bn0(self.fc0(x))))
