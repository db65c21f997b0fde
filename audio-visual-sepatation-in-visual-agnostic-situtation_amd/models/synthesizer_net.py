"""Mask synthesizer (reference: models/synthesizer_net.py:6-70); used by the SoP++ variant only.

forward() of InnerProd/Bias (the K-vector x K basis maps GEMV, HBM bound) is a HIP kernel when no
gradient is required; the differentiable and the per-pixel inference forms use PyTorch-ROCm ops.
"""
import torch
import torch.nn as nn

from .. import lib
from ..lib import call, ptr


class InnerProd(nn.Module):
    def __init__(self, fc_dim):
        super().__init__()
        self.scale = nn.Parameter(torch.ones(fc_dim))
        self.bias = nn.Parameter(torch.zeros(1))

    def _w(self, f):
        return f * self.scale

    def forward(self, feat_img, feat_sound):
        B, C = feat_sound.shape[:2]
        if feat_sound.is_cuda and not (torch.is_grad_enabled() and (
                feat_img.requires_grad or feat_sound.requires_grad or self.bias.requires_grad)):
            z = torch.empty((B, 1) + tuple(feat_sound.shape[2:]), dtype=torch.float32, device=feat_sound.device)
            scale = getattr(self, "scale", None)
            call("avsep_innerprod_fwd", ptr(feat_img.reshape(B, C).contiguous().float()),
                 ptr(feat_sound.contiguous().float()), ptr(scale.detach() if scale is not None else None),
                 ptr(self.bias.detach()), B, C, feat_sound[0, 0].numel(), ptr(z))
            return z
        z = torch.bmm(self._w(feat_img.view(B, 1, C)), feat_sound.reshape(B, C, -1))
        return z.view(B, 1, *feat_sound.shape[2:]) + self.bias

    def forward_nosum(self, feat_img, feat_sound):
        B, C = feat_sound.shape[:2]
        return self._w(feat_img.view(B, C)).view(B, C, 1, 1) * feat_sound + self.bias

    def forward_pixelwise(self, feats_img, feat_sound):
        B, C, HI, WI = feats_img.shape
        _, _, HS, WS = feat_sound.shape
        fi = self._w(feats_img.view(B, C, HI * WI).transpose(1, 2))
        return torch.bmm(fi, feat_sound.view(B, C, HS * WS)).view(B, HI, WI, HS, WS) + self.bias


class Bias(InnerProd):
    def __init__(self):
        nn.Module.__init__(self)
        self.bias = nn.Parameter(torch.zeros(1))

    def _w(self, f):
        return f
