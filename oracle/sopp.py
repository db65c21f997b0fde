"""Oracle for the SoP++ operators: attention module + stage math.  TEST INFRASTRUCTURE ONLY.

Reference: SoP++/attention_net.py:16-232 (AttModel, MatchAtt), SoP++/main.py:94-246
(stage math; that driver does not run as shipped — SURVEY.md §2 note S1 — so only
the operators are pinned, through gen_golden.py).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def _pool(t):
    # attention_net.py:19 — the attribute is named max_pool but is an AVERAGE pool
    return t.mean(dim=(-2, -1))


def _maps(att_type, a, v):
    """a: [B,C,K]; v: [B,K,H,W] -> [B,C,H,W]  (attention_net.py:22-34; sig divides by sqrt(K))."""
    a5 = a[..., None, None]
    v5 = v[:, None]
    if att_type == "cos":
        return F.cosine_similarity(a5, v5, dim=2)
    return torch.sigmoid(torch.sum(a5 * v5 / math.sqrt(a.shape[2]), dim=2))


class AttModule(nn.Module):
    def __init__(self, kind="AttModel", att_type="cos"):
        super().__init__()
        assert kind in ("AttModel", "MatchAtt")
        self.kind, self.att_type = kind, att_type

    def infer(self, a, mix):
        # attention_net.py:36-59
        maps = _maps(self.att_type, a, mix)
        peaks = _pool(maps)                                   # (average, see _pool)
        match = -peaks.sum(-1).mean().reshape(1)
        maps = maps.clamp(0, 1)
        ctx = _pool(mix[:, None] * maps[:, :, None])          # [B,C,K]
        return ctx, (match, maps)

    @staticmethod
    def _pit(cand, glb):
        """cand, glb: [B,C,K]; best permutation of cand against glb by summed cosine."""
        both = torch.stack([cand, cand.flip(1)], 1)           # [B,P,C,K]
        scores = F.cosine_similarity(both, glb[:, None], dim=3).sum(-1)
        srt, idx = torch.sort(scores, dim=1, descending=True)
        match = (-srt[:, 0] + srt[:, 1:].sum(-1)).mean(0).reshape(1)
        return both, idx, match

    def forward(self, aud_feats, mix_vis, sep_vis):
        a = torch.stack([_pool(f) for f in aud_feats], 1)     # [B,C,K]
        if mix_vis is None:
            return a, None                                     # ao_forward :61-75
        if sep_vis is None:
            return self.infer(a, mix_vis)
        glb = torch.stack([_pool(f) for f in sep_vis], 1)
        B = a.shape[0]
        if self.kind == "AttModel":                            # :78-108
            ctx, (reg, maps) = self.infer(a, mix_vis)
            both, idx, match = self._pit(ctx, glb)
            ctx = both[torch.arange(B), idx[:, 0]]
            maps = torch.gather(maps, 1, idx[:, :, None, None].expand_as(maps))
            return ctx, (match, reg, maps)
        both, idx, match = self._pit(a, glb)                   # MatchAtt :181-221
        a = both[torch.arange(B), idx[:, 0]]
        ctx, (_, maps) = self.infer(a, mix_vis)
        return ctx, (match, maps)


# ---------------------------------------------------------------------------
# Stage math of SoP++/main.py:94-246 on the oracle modules (same documented repairs as the product:
# stacked PIT weight in ao_forward; stages 2/3 need AttModel's 3-tuple meta).
# ---------------------------------------------------------------------------
class SopNetWrapper(nn.Module):
    def __init__(self, nets, crit_ao, crit_av):
        super().__init__()
        self.net_sound, self.net_frame, self.net_synthesizer, self.net_pit = nets
        self.crit_ao, self.crit_av = crit_ao, crit_av

    def forward(self, batch, args, use_vis, stage=3):
        from .step import prepare
        from .nets import activate
        mags, mag_mix, log_mag_mix, gt_masks, weight = prepare(batch, args)
        N = args.num_mix
        basis, meta = self.net_sound(log_mag_mix)
        basis = activate(basis, args.sound_activation)
        fw = torch.tensor_split(meta[0], N, dim=1)

        def synth(ctx):
            return [activate(self.net_synthesizer(ctx[:, n, :], basis), args.output_activation) for n in range(N)]
        if not use_vis:
            ctx, _ = self.net_pit(fw, None, None)
            pred = torch.stack(synth(ctx), -1).squeeze(1)
            gt = torch.stack(gt_masks, -1)[:, 0]
            err, perms = self.crit_ao(pred, gt, torch.stack([weight[:, 0]] * N, -1))
            ordered = self.crit_ao.reorder_tensor(pred, perms)
            return err.mean(), {"pred_masks": [ordered[..., i].unsqueeze(1) for i in range(N)]}
        frames = batch["frames"]

        def vis():
            return [activate(self.net_frame.forward_multiframe(f, args.not_pool_vis), args.img_activation) for f in frames]
        if stage == 1:
            ff = vis()
        else:
            with torch.no_grad():
                ff = vis()
            mix = activate(self.net_frame.forward_multiframe(torch.cat(frames, -1), args.not_pool_vis), args.img_activation)
            ctx3, m = self.net_pit(fw, mix, ff)
        gctx = activate(torch.stack(ff, 1).mean(dim=(-2, -1)), args.output_activation)
        if stage == 3:
            pred = synth(activate(ctx3, args.output_activation))
            extra = m[1] + m[0]
        else:
            pred = synth(gctx)
            extra = m[1] if stage == 2 else None
        err = self.crit_av(pred, gt_masks, weight).reshape(1)
        if extra is not None:
            err = err + extra * args.match_weight
        return err, {"pred_masks": pred, "match_loss": extra}
