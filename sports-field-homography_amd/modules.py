"""Parameter containers that reproduce the reference checkpoint (state_dict) layout.

These classes hold *parameters only*.  Their attribute names are the checkpoint
contract of the reference:

* ``DoubleConv.double_conv.{0,3}`` conv3x3 / ``{1,4}`` BatchNorm  (reference: unet/unet_parts.py:7-24)
* ``Down.maxpool_conv.1``                                          (reference: unet/unet_parts.py:27-38)
* ``Up.up`` (ConvTranspose2d 2x2 s2 or Upsample) + ``Up.conv``     (reference: unet/unet_parts.py:41-68)
* ``OutConv.conv``                                                 (reference: unet/unet_parts.py:71-77)
* ``ResNetSTN.{conv0,bn1,layer1..4,reg}``                          (reference: models/resnet.py:143-257)

None of them implements ``forward``: all arithmetic of the hot path runs in the HIP
kernels driven by :mod:`sfh_amd.engine`.  ``torch.nn`` layer objects are used purely
as named parameter/buffer holders so that ``state_dict()`` / ``load_state_dict()`` /
``.to()`` / ``.parameters()`` behave exactly like the reference's modules.
"""
import math

import torch
import torch.nn as nn


class _NoForward(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover - guard
        raise RuntimeError(
            f"{type(self).__name__} is a parameter container; the forward pass runs in the "
            "HIP engine (sfh_amd.engine), not through torch.nn"
        )


class DoubleConv(_NoForward):
    """[conv3x3 pad1 + bias -> BatchNorm2d -> ReLU] x 2 (reference: unet/unet_parts.py:7-24)."""

    def __init__(self, in_channels, out_channels, mid_channels=None):
        super().__init__()
        mid = mid_channels if mid_channels else out_channels
        self.double_conv = nn.Sequential(
            nn.Conv2d(in_channels, mid, kernel_size=3, padding=1),
            nn.BatchNorm2d(mid),
            nn.ReLU(inplace=True),
            nn.Conv2d(mid, out_channels, kernel_size=3, padding=1),
            nn.BatchNorm2d(out_channels),
            nn.ReLU(inplace=True),
        )

    def convs(self):
        s = self.double_conv
        return (s[0], s[1]), (s[3], s[4])


class Down(_NoForward):
    """MaxPool2d(2) then DoubleConv (reference: unet/unet_parts.py:27-38)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.maxpool_conv = nn.Sequential(nn.MaxPool2d(2), DoubleConv(in_channels, out_channels))

    @property
    def block(self):
        return self.maxpool_conv[1]


class Up(_NoForward):
    """2x upsampling (ConvTranspose2d k2 s2, or bilinear) + pad + concat + DoubleConv
    (reference: unet/unet_parts.py:41-68)."""

    def __init__(self, in_channels, out_channels, bilinear=True):
        super().__init__()
        self.bilinear = bool(bilinear)
        if bilinear:
            self.up = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)
            self.conv = DoubleConv(in_channels, out_channels, in_channels // 2)
        else:
            self.up = nn.ConvTranspose2d(in_channels, in_channels // 2, kernel_size=2, stride=2)
            self.conv = DoubleConv(in_channels, out_channels)


class OutConv(_NoForward):
    """1x1 conv + bias (reference: unet/unet_parts.py:71-77)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=1)


class BasicBlock(_NoForward):
    """Two 3x3 convs (no bias) + BN, residual add (reference: models/resnet.py:36-82)."""

    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=1, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride


class Bottleneck(_NoForward):
    """1x1 -> 3x3 (carries the stride) -> 1x1 (x4 channels) + BN, residual add
    (reference: models/resnet.py:85-140, the "ResNet V1.5" placement of the stride)."""

    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, base_width=64):
        super().__init__()
        width = int(planes * (base_width / 64.0))
        self.conv1 = nn.Conv2d(inplanes, width, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, planes * self.expansion, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride


# name -> (block, blocks per stage, width_per_group); reference table models/resnet.py:361-371
# ('resnet52' is the reference's key for its ResNet-152 constructor).  Through the reference's
# factory only resnet18/34/50 are constructible with an explicit in_channels (the deeper and wide
# constructors do not take that argument, models/resnet.py:297-355); they are accepted here.
_RESNET_VARIANTS = {
    "resnet18": (BasicBlock, (2, 2, 2, 2), 64),
    "resnet34": (BasicBlock, (3, 4, 6, 3), 64),
    "resnet50": (Bottleneck, (3, 4, 6, 3), 64),
    "resnet101": (Bottleneck, (3, 4, 23, 3), 64),
    "resnet52": (Bottleneck, (3, 8, 36, 3), 64),
    "resnet152": (Bottleneck, (3, 8, 36, 3), 64),
    "wide_resnet50_2": (Bottleneck, (3, 4, 6, 3), 128),
    "wide_resnet101_2": (Bottleneck, (3, 4, 23, 3), 128),
}
_RESNET_LAYERS = {k: v[1] for k, v in _RESNET_VARIANTS.items()}


class ResNetSTN(_NoForward):
    """ResNet regressor that emits a 3x3 homography (reference: models/resnet.py:143-257).

    BasicBlock and Bottleneck depths run on the HIP path; the grouped-convolution ResNeXt
    variants (models/resnet.py:318-337) do not.
    """

    def __init__(self, name="resnet34", in_channels=4):
        super().__init__()
        if name not in _RESNET_VARIANTS:
            raise NotImplementedError(
                f"resnet_name={name!r}: only {sorted(_RESNET_VARIANTS)} run on the HIP path "
                "(ResNeXt needs grouped convolutions)"
            )
        self.block, layers, self.base_width = _RESNET_VARIANTS[name]
        self.inplanes = 64
        self.conv0 = nn.Conv2d(in_channels, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._stage(64, layers[0], 1)
        self.layer2 = self._stage(128, layers[1], 2)
        self.layer3 = self._stage(256, layers[2], 2)
        self.layer4 = self._stage(512, layers[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.reg = nn.Linear(512 * self.block.expansion, 9)
        self._init_weights()

    def _stage(self, planes, blocks, stride):
        block, kw = self.block, ({} if self.block is BasicBlock else {"base_width": self.base_width})
        down = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            down = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride=stride, bias=False),
                nn.BatchNorm2d(planes * block.expansion),
            )
        mods = [block(self.inplanes, planes, stride, down, **kw)]
        self.inplanes = planes * block.expansion
        mods += [block(self.inplanes, planes, **kw) for _ in range(1, blocks)]
        return nn.Sequential(*mods)

    def _init_weights(self):
        # Same initial state as the reference (models/resnet.py:189-208): He fan-out
        # normal conv weights, unit BN, and an identity homography in ``reg``.
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                fan_out = m.out_channels * m.kernel_size[0] * m.kernel_size[1]
                with torch.no_grad():
                    m.weight.normal_(0.0, math.sqrt(2.0 / fan_out))
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
        with torch.no_grad():
            self.reg.weight.zero_()
            self.reg.bias.copy_(torch.tensor([1, 0, 0, 0, 1, 0, 0, 0, 1], dtype=torch.float32))


def resnet_stn(name, pretrained_path=None, in_channels=4):
    """Factory with the reference's signature (models/resnet.py:373-374)."""
    model = ResNetSTN(name, in_channels)
    if pretrained_path is not None:
        model.load_state_dict(torch.load(pretrained_path, map_location="cpu"), strict=False)
    return model
