"""SoP++ attention module (reference: SoP++/attention_net.py:8-232): ``get_attmodule(args)`` ->
``AttModel`` ("Base") / ``MatchAtt``.  Parameter free; K-dimensional (K = num_channels, 32 by default)
vectors against a 14x28 map: ~0.03 MMAC per sample, pure latency.

Round-1 state: these two modules run on PyTorch-ROCm tensor operators on the GPU (like the visual trunk),
not on a dedicated HIP kernel; the U-Net that feeds them and the synthesizer that consumes them are HIP.
Quirks kept: the attribute the reference calls ``max_pool`` is an AVERAGE pool (:19); the sigmoid kernel divides
by sqrt(K) = sqrt(x.shape[2]) (:33); maps are clamped to [0,1] AFTER the match loss is taken (:49-51).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _pool(t):
    return t.mean(dim=(-2, -1))


class _AttBase(nn.Module):
    def __init__(self, **kwargs):
        super().__init__()
        self.att_type = kwargs.get("att_type", "cos")

    def _maps(self, a, v):
        a5, v5 = a[..., None, None], v[:, None]
        if self.att_type == "cos":
            return F.cosine_similarity(a5, v5, dim=2)
        return torch.sigmoid(torch.sum(a5 * v5 / (a.shape[2]) ** 0.5, dim=2))

    def _infer(self, a, mix):
        maps = self._maps(a, mix)
        match = -_pool(maps).sum(-1).mean().reshape(1)
        maps = maps.clamp(0, 1)
        ctx = _pool(mix[:, None] * maps[:, :, None])
        return ctx, (match, maps)

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
