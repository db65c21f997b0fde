"""Mask synthesizer (reference: models/synthesizer_net.py:6-70); used by the SoP++ variant only.

``forward`` — the K-vector x K basis maps GEMV (8.4 MB of basis per sample at K=32: HBM bound) — and its
backward are HIP kernels; so are the inference helpers ``forward_nosum`` (one elementwise pass) and
``forward_pixelwise`` (the visual-feature x audio-feature contraction on the f32 MFMA, avsep_innerprod_pixelwise).
The reference calls the helpers without autograd only ("inference purposes", synthesizer_net.py:28); when a gradient
is required through them the same formulas run as differentiable tensor expressions.
"""
import torch
import torch.nn as nn

from .. import lib
from ..lib import call, ptr


class _InnerProdFn(torch.autograd.Function):
    """z[b,hw] = sum_k img[b,k]*scale[k]*snd[b,k,hw] + bias  (scale may be None for `Bias`)."""

    @staticmethod
    def forward(ctx, img, snd, scale, bias):
        lib.require_gpu(snd)
        B, K = snd.shape[:2]
        HW = snd[0, 0].numel()
        img, snd = img.reshape(B, K).contiguous().float(), snd.contiguous().float()
        z = torch.empty((B, 1) + tuple(snd.shape[2:]), dtype=torch.float32, device=snd.device)
        call("avsep_innerprod_fwd", ptr(img), ptr(snd), ptr(scale), ptr(bias), B, K, HW, ptr(z))
        ctx.save_for_backward(img, snd, scale)
        return z

    @staticmethod
    def backward(ctx, dz):
        img, snd, scale = ctx.saved_tensors
        B, K = snd.shape[:2]
        HW = snd[0, 0].numel()
        dz = dz.contiguous()
        dsnd = torch.empty_like(snd) if ctx.needs_input_grad[1] else None
        r = torch.empty((B, K), dtype=torch.float32, device=snd.device)
        call("avsep_innerprod_bwd", ptr(img), ptr(snd), ptr(scale), ptr(dz), B, K, HW, ptr(dsnd), ptr(r))
        dimg = r * scale if scale is not None else r
        dscale = (img * r).sum(0) if scale is not None else None
        return dimg, dsnd, dscale, dz.sum().reshape(1)


class InnerProd(nn.Module):
    def __init__(self, fc_dim):
        super().__init__()
        self.scale = nn.Parameter(torch.ones(fc_dim))
        self.bias = nn.Parameter(torch.zeros(1))

    def _scale(self):
        return self.scale

    def _w(self, f):
        return f * self.scale

    def forward(self, feat_img, feat_sound):
        return _InnerProdFn.apply(feat_img, feat_sound, self._scale(), self.bias)

    def _wants_grad(self, *ts):
        return torch.is_grad_enabled() and any(t.requires_grad for t in list(ts) + list(self.parameters()))

    def forward_nosum(self, feat_img, feat_sound):
        B, C = feat_sound.shape[:2]
        if self._wants_grad(feat_img, feat_sound):
            return self._w(feat_img.view(B, C)).view(B, C, 1, 1) * feat_sound + self.bias
        lib.require_gpu(feat_sound)
        snd = feat_sound.contiguous().float()
        z = torch.empty_like(snd)
        call("avsep_innerprod_nosum", ptr(feat_img.reshape(B, C).contiguous().float()), ptr(snd), ptr(self._scale()),
             ptr(self.bias), B, C, snd[0, 0].numel(), ptr(z))
        return z

    def forward_pixelwise(self, feats_img, feat_sound):
        B, C, HI, WI = feats_img.shape
        _, _, HS, WS = feat_sound.shape
        if self._wants_grad(feats_img, feat_sound) or C % 2:
            fi = self._w(feats_img.view(B, C, HI * WI).transpose(1, 2))
            return torch.bmm(fi, feat_sound.view(B, C, HS * WS)).view(B, HI, WI, HS, WS) + self.bias
        lib.require_gpu(feat_sound)
        z = torch.empty((B, HI, WI, HS, WS), dtype=torch.float32, device=feat_sound.device)
        call("avsep_innerprod_pixelwise", ptr(feats_img.contiguous().float()), ptr(feat_sound.contiguous().float()),
             ptr(self._scale()), ptr(self.bias), B, C, HI * WI, HS * WS, ptr(z))
        return z


class Bias(InnerProd):
    def __init__(self):
        nn.Module.__init__(self)
        self.bias = nn.Parameter(torch.zeros(1))

    def _scale(self):
        return None

    def _w(self, f):
        return f
