"""Patch encoders of the CRW model -- same class surface and ``state_dict`` keys as the reference
(``CNN(pos_embed)``: src/encoder.py:9-57, ``Resnet(pos_embed, pretrained)``: src/encoder.py:63-89)
so checkpoints are interchangeable and, with the same ``torch.manual_seed``, the initial weights
are identical (layers are constructed in the same order, so the RNG stream is consumed alike).

The arithmetic here is PyTorch-ROCm (MIOpen convolutions): per BASELINE.json's north_star the
encoder is "Python host code on PyTorch-ROCm"; the hand-written HIP kernels start at the encoder
output (model.py -> crw_hip).
"""
import torch
import torch.nn as nn
import torch.nn.functional as TF

FEATURE_DIM = 128

# (name, out_channels, kernel, followed by 2x2/stride-1 max-pool?)
_CNN_STACK = (("1", 8, 5, True), ("2", 32, 5, True), ("3", 64, 3, False), ("4", 128, 3, False), ("5", 128, 3, False))


def _report(module):
    n = sum(p.numel() for p in module.parameters() if p.requires_grad)
    print(f"Number of trainable parameters: {n}")
    return n


class CNN(nn.Module):
    """Five conv layers (5x5, 5x5, 3x3, 3x3, 3x3; all padding 1) with ReLU, stride-1 max-pools
    after the first two, global average pool and a linear head -> 128-d feature per patch."""

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

    def forward(self, x):
        for name, _, _, pooled in _CNN_STACK:
            x = getattr(self, "relu" + name)(getattr(self, "conv" + name)(x))
            if pooled:
                x = getattr(self, "pool" + name)(x)
        return self.fc(self.global_avg_pool(x).flatten(1))


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
    """1x1 conv (padding 1) + BN + ReLU lifting 1-2 channels to 3, then the ResNet body."""

    def __init__(self, pos_embed=True, pretrained=None):  # `pretrained` is ignored by the reference too
        super().__init__()
        self.fc0 = nn.Conv2d(2 if pos_embed else 1, 3, kernel_size=1, padding=1)
        self.bn0 = nn.BatchNorm2d(3)
        self.relu0 = nn.ReLU(inplace=True)
        self.model = _ResNetBody(FEATURE_DIM)
        self.num_params = _report(self)

    def forward(self, x):
        return self.model(self.relu0(self.bn0(self.fc0(x))))
