"""SoP++ attention module (reference: SoP++/attention_net.py:8-232): ``get_attmodule(args)`` ->
``AttModel`` ("Base") / ``MatchAtt``.  Parameter free; K-dimensional (K = num_channels, 32 by default)
vectors against a 14x28 map: ~0.03 MMAC per sample, pure latency.

The core both modules share — similarity maps of the pooled audio queries against the mixed visual map, the match
term, the clamp and the attention-weighted context vectors (`att` + `av_infer_forward`, :24-58) — is ONE HIP launch
forward and one backward (csrc/attention.hip, one workgroup per sample); the remaining vector arithmetic (4-element
pools of the [B,K,2,2] queries, the two-permutation cosine matching of [B,2,K] vectors) stays on a handful of tensor ops.
There is no CPU path: the module raises on CPU tensors like every other entry point.
Quirks kept: the attribute the reference calls ``max_pool`` is an AVERAGE pool (:19); the sigmoid kernel divides
by sqrt(K) = sqrt(x.shape[2]) (:33); maps are clamped to [0,1] AFTER the match loss is taken (:49-51).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import lib
from ..lib import call, ptr

_ATT = {"cos": 0, "sig": 1}


def _pool(t):
    return t.mean(dim=(-2, -1))


class _AttInferFn(torch.autograd.Function):
    """(a [B,S,K], mix [B,K,H,W]) -> (ctx [B,S,K], clamped maps [B,S,H,W], match [B]) on csrc/attention.hip."""

    @staticmethod
    def forward(ctx_, a, mix, att):
        B, S, K = a.shape
        H, W = mix.shape[-2:]
        a, mix = a.contiguous().float(), mix.contiguous().float()
        raw = torch.empty((B, S, H, W), dtype=torch.float32, device=a.device)
        ctx = torch.empty((B, S, K), dtype=torch.float32, device=a.device)
        match = torch.empty((B,), dtype=torch.float32, device=a.device)
        call("avsep_attmodel_infer_fwd", ptr(a), ptr(mix), B, S, K, H * W, att, ptr(raw), ptr(ctx), ptr(match))
        ctx_.save_for_backward(a, mix, raw)
        ctx_.att = att
        return ctx, raw.clamp(0, 1), match

    @staticmethod
    def backward(ctx_, dctx, dmaps, dmatch):
        a, mix, raw = ctx_.saved_tensors
        B, S, K = a.shape
        H, W = mix.shape[-2:]
        da, dmix = torch.empty_like(a), torch.empty_like(mix)
        dctx = dctx.contiguous().float() if dctx is not None else torch.zeros_like(a)
        dmaps = dmaps.contiguous().float() if dmaps is not None else None
        dmatch = dmatch.contiguous().float() if dmatch is not None else None
        call("avsep_attmodel_infer_bwd", ptr(a), ptr(mix), ptr(raw), ptr(dctx), ptr(dmaps), ptr(dmatch), B, S, K, H * W,
             ctx_.att, ptr(da), ptr(dmix))
        return da, dmix, None


class _AttBase(nn.Module):
    def __init__(self, **kwargs):
        super().__init__()
        self.att_type = kwargs.get("att_type", "cos")

    def _infer(self, a, mix):
        lib.require_gpu(mix)
        ctx, maps, match = _AttInferFn.apply(a, mix, _ATT[self.att_type])      # S <= 4, K <= 128, H*W <= 4096 (checked by the ABI)
        return ctx, (match.mean().reshape(1), maps)

    def av_infer_forward(self, aud_feats, mix_vis_feats):
        return self._infer(torch.stack([_pool(f) for f in aud_feats], 1), mix_vis_feats)

    def ao_forward(self, aud_feats):
        return torch.stack([_pool(f) for f in aud_feats], 1), None

    @staticmethod
    def _pit(cand, glb):
        both = torch.stack([cand, cand.flip(1)], 1)
        scores = F.cosine_similarity(both, glb[:, None], dim=3).sum(-1)
        srt, idx = torch.sort(scores, dim=1, descending=True)
        match = (-srt[:, 0] + srt[:, 1:].sum(-1)).mean(0).reshape(1)
        return both, idx, match

    def forward(self, aud_feats, mix_vis_feats, sep_vis_feats):
        assert aud_feats is not None
        if mix_vis_feats is None:
            return self.ao_forward(aud_feats)
        if sep_vis_feats is None:
            return self.av_infer_forward(aud_feats, mix_vis_feats)
        return self.av_train_forward(aud_feats, mix_vis_feats, sep_vis_feats)


class AttModel(_AttBase):
    def av_train_forward(self, aud_feats, mix_vis_feats, sep_vis_feats):
        ctx, (reg, maps) = self.av_infer_forward(aud_feats, mix_vis_feats)
        glb = torch.stack([_pool(f) for f in sep_vis_feats], 1)
        both, idx, match = self._pit(ctx, glb)
        ctx = both[torch.arange(ctx.shape[0], device=ctx.device), idx[:, 0]]
        maps = torch.gather(maps, 1, idx[:, :, None, None].expand_as(maps))
        return ctx, (match, reg, maps)


class MatchAtt(_AttBase):
    def av_train_forward(self, aud_feats, mix_vis_feats, sep_vis_feats):
        a = torch.stack([_pool(f) for f in aud_feats], 1)
        glb = torch.stack([_pool(f) for f in sep_vis_feats], 1)
        both, idx, match = self._pit(a, glb)
        a = both[torch.arange(a.shape[0], device=a.device), idx[:, 0]]
        ctx, (_, maps) = self._infer(a, mix_vis_feats)
        return ctx, (match, maps)


def get_attmodule(args):
    if args.fusion_type == "Base":
        return AttModel
    elif args.fusion_type == "MatchAtt":
        return MatchAtt
    assert False
