"""Per-frame visual encoder (reference: models/vision_net.py:20-147 + torchvision resnet18).

The ResNet-18 trunk + fc conv run on libavsep_gfx950.so as ONE autograd node (models/vision_hip.py), in fp32 or with bf16
conv operands (kernels.set_precision): the full-HIP path of BASELINE.json configs[2].  There is no other backend in the
package; the MIOpen comparison runs of DESIGN.md live in tools/miopen_compare/ (they patch an instance, nothing here
dispatches on an environment variable).
The temporal mean that feeds the fusion is a HIP kernel.
torchvision is not part of this image, so the standard ResNet-18 architecture is restated here
with torchvision's child order, which keeps the reference's ``features.{0,1,4..7}.*`` /
``fc.*`` checkpoint keys.  ``pretrained=True`` (models/__init__.py:63) cannot be honoured
offline: weights are PyTorch's default init unless a checkpoint is loaded.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import kernels as K
from . import vision_hip


class BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return self.relu(y + idt)


def resnet18_features():
    def layer(cin, cout, stride):
        return nn.Sequential(BasicBlock(cin, cout, stride), BasicBlock(cout, cout, 1))
    return nn.Sequential(nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True),
                         nn.MaxPool2d(3, 2, 1), layer(64, 64, 1), layer(64, 128, 2), layer(128, 256, 2),
                         layer(256, 512, 2))


class _TemporalMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, B, T):
        ctx.bt = (B, T)
        return K.temporal_mean(x.contiguous(), B, T)

    @staticmethod
    def backward(ctx, dy):
        B, T = ctx.bt
        return K.temporal_mean_bwd(dy.contiguous(), B, T), None, None


class _VisualBase(nn.Module):
    backend = "hip"

    def _trunk(self, x):
        """fc(features(x)) for x [N,3,H,W]."""
        if self.backend != "hip":
            raise Exception("Unknown vision backend: " + str(self.backend) + " (the package has the HIP trunk only; "
                            "MIOpen comparison runs: tools/miopen_compare)")
        return vision_hip.run(self, x)

    def forward(self, x, pool=True):
        x = self._trunk(x)
        if not pool:
            return x.contiguous()
        if self.pool_type == "avgpool":
            x = F.adaptive_avg_pool2d(x, 1)
        elif self.pool_type == "maxpool":
            x = F.adaptive_max_pool2d(x, 1)
        return x.view(x.size(0), x.size(1))

    def forward_multiframe(self, x, pool=True):
        # vision_net.py:126-147
        B, C, T, H, W = x.shape
        y = self._trunk(x.permute(0, 2, 1, 3, 4).reshape(B * T, C, H, W))
        if not pool:
            if y.is_cuda:
                return _TemporalMean.apply(y, B, T)           # [B,C,h,w]
            return y.contiguous().view(B, T, *y.shape[1:]).mean(1)
        y = y.contiguous().view(B, T, *y.shape[1:]).permute(0, 2, 1, 3, 4)
        if self.pool_type == "avgpool":
            return y.mean(dim=(2, 3, 4))
        return y.amax(dim=(2, 3, 4))


class ResnetFC(_VisualBase):
    def __init__(self, original_resnet=None, fc_dim=64, pool_type="maxpool", conv_size=3):
        super().__init__()
        self.pool_type = pool_type
        self.features = resnet18_features() if original_resnet is None else \
            nn.Sequential(*list(original_resnet.children())[:-2])
        self.fc = nn.Conv2d(512, fc_dim, kernel_size=conv_size, padding=conv_size // 2)


class ResnetDilated(_VisualBase):
    def __init__(self, orig_resnet=None, fc_dim=64, pool_type="maxpool", dilate_scale=16, conv_size=3):
        super().__init__()
        self.pool_type = pool_type
        self.features = resnet18_features() if orig_resnet is None else \
            nn.Sequential(*list(orig_resnet.children())[:-2])
        if dilate_scale == 8:
            self._nostride_dilate(self.features[6], 2)
            self._nostride_dilate(self.features[7], 4)
        elif dilate_scale == 16:
            self._nostride_dilate(self.features[7], 2)
        self.fc = nn.Conv2d(512, fc_dim, kernel_size=conv_size, padding=conv_size // 2)

    @staticmethod
    def _nostride_dilate(layer, dilate):
        # vision_net.py:96-109: remove the stride, dilate the 3x3s (the first one by dilate//2)
        for m in layer.modules():
            if isinstance(m, nn.Conv2d):
                if m.stride == (2, 2):
                    m.stride = (1, 1)
                    if m.kernel_size == (3, 3):
                        m.dilation = (dilate // 2, dilate // 2)
                        m.padding = (dilate // 2, dilate // 2)
                elif m.kernel_size == (3, 3):
                    m.dilation = (dilate, dilate)
                    m.padding = (dilate, dilate)
