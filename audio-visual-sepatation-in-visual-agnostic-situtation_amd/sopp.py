"""SoP++ step definition (reference: SoP++/main.py:38-259): basis U-Net + attention module + synthesizer,
three audio-visual training stages and the audio-only PIT stage.  The reference's SoP++ driver does not run
as shipped (SURVEY.md §2 note S1); this module reproduces the *math* of its NetWrapper on the HIP path.

Repairs, both documented in DESIGN.md: ``ao_forward`` hands PitWrapper an un-stacked weight (same defect
as main.py:181) -> per-target stacking as in main.py:103; stages 2/3 read ``meta[1]`` as a regulariser,
which only ``AttModel`` ("Base") provides (MatchAtt returns a 2-tuple) -> stages 2/3 require AttModel.
"""
import torch

from .models import activate
from .models.criterion import PitWrapper
from .net_wrapper import NetWrapper as _StepBase


class NetWrapper(_StepBase):
    def __init__(self, nets, crit_ao, crit_av):
        torch.nn.Module.__init__(self)
        self.net_sound, self.net_frame, self.net_synthesizer, self.net_pit = nets
        self.load_clips = False
        self.crit_ao, self.crit_av = crit_ao, crit_av
        self.stft_plan = None

    # ---- shared pieces -------------------------------------------------------------------------
    def _sound(self, log_mag_mix, args, N):
        feat_basis, meta = self.net_sound(log_mag_mix)
        feat_basis = activate(feat_basis, args.sound_activation)
        return feat_basis, torch.tensor_split(meta[0], N, dim=1)

    def _frames(self, frames, args):
        return [activate(self.net_frame.forward_multiframe(f, args.not_pool_vis), args.img_activation) for f in frames]

    def _synth(self, ctx_feats, feat_basis, args, N):
        return [activate(self.net_synthesizer(ctx_feats[:, n, :], feat_basis), args.output_activation)
                for n in range(N)]

    @staticmethod
    def _global_ctx(feat_frames, args):
        ctx = torch.stack(feat_frames, dim=1).mean(dim=(-2, -1))                     # adaptive_avg_pool3d((None,1,1))
        return activate(ctx, args.output_activation)

    def _mix_vis(self, frames, args):
        concat = torch.cat(list(frames), dim=-1)
        return activate(self.net_frame.forward_multiframe(concat, args.not_pool_vis), args.img_activation)

    @staticmethod
    def _out(pred_masks, gt_masks, mag_mix, mags, weight, match):
        return {"pred_masks": pred_masks, "gt_masks": gt_masks, "mag_mix": mag_mix, "mags": mags, "weight": weight,
                "match_loss": match}

    # ---- SoP++/main.py:94-127 ------------------------------------------------------------------
    def train_av_forward1(self, data, args):
        frames, _, mags, mag_mix, log_mag_mix, gt_masks, weight = data
        N = len(frames)
        feat_basis, _ = self._sound(log_mag_mix, args, N)
        ctx = self._global_ctx(self._frames(frames, args), args)
        pred = self._synth(ctx, feat_basis, args, N)
        err = self.crit_av(pred, gt_masks, weight).reshape(1)
        return err, self._out(pred, gt_masks, mag_mix, mags, weight, torch.zeros_like(err))

    # ---- SoP++/main.py:129-170 -----------------------------------------------------------------
    def train_av_forward2(self, data, args):
        frames, _, mags, mag_mix, log_mag_mix, gt_masks, weight = data
        N = len(frames)
        feat_basis, feat_weights = self._sound(log_mag_mix, args, N)
        with torch.no_grad():
            feat_frames = self._frames(frames, args)
        _, meta = self.net_pit(feat_weights, self._mix_vis(frames, args), feat_frames)
        reg_loss = self._reg(meta)
        pred = self._synth(self._global_ctx(feat_frames, args), feat_basis, args, N)
        err = self.crit_av(pred, gt_masks, weight).reshape(1)
        return err + reg_loss * args.match_weight, self._out(pred, gt_masks, mag_mix, mags, weight, reg_loss.reshape(1))

    # ---- SoP++/main.py:172-213 -----------------------------------------------------------------
    def train_av_forward3(self, data, args):
        frames, _, mags, mag_mix, log_mag_mix, gt_masks, weight = data
        N = len(frames)
        feat_basis, feat_weights = self._sound(log_mag_mix, args, N)
        with torch.no_grad():
            feat_frames = self._frames(frames, args)
        ctx, meta = self.net_pit(feat_weights, self._mix_vis(frames, args), feat_frames)
        ctx = activate(ctx, args.output_activation)
        match_loss, reg_loss = meta[0], self._reg(meta)
        pred = self._synth(ctx, feat_basis, args, N)
        err = self.crit_av(pred, gt_masks, weight).reshape(1)
        return err + (reg_loss + match_loss) * args.match_weight, \
            self._out(pred, gt_masks, mag_mix, mags, weight, (reg_loss + match_loss).reshape(1))

    @staticmethod
    def _reg(meta):
        if len(meta) != 3:
            raise ValueError("stages 2/3 read meta[1] as the regulariser: only AttModel ('Base') provides it")
        return meta[1]

    # ---- SoP++/main.py:215-246 -----------------------------------------------------------------
    def ao_forward(self, data, args):
        mags, mag_mix, log_mag_mix, gt_masks, weight = data
        N = len(mags)
        feat_basis, feat_weights = self._sound(log_mag_mix, args, N)
        ctx, _ = self.net_pit(feat_weights, None, None)
        pred = torch.stack(self._synth(ctx, feat_basis, args, N), dim=-1).squeeze(1)   # B x F x T x C
        gt = torch.stack(gt_masks, dim=-1)[:, 0]
        w = torch.stack([weight[:, 0]] * N, dim=-1)          # repair, see the module docstring
        err, perms = self.crit_ao(pred, gt, w)
        err = torch.mean(err)
        ordered = PitWrapper.reorder_tensor(pred, perms)
        return err, {"pred_masks": [ordered[..., i].unsqueeze(1) for i in range(N)],
                     "gt_masks": [gt[..., i].unsqueeze(1) for i in range(N)],
                     "mag_mix": mag_mix, "mags": mags, "weight": weight}

    def forward(self, batch_data, args, use_vis, stage=3):
        if "mag_mix" not in batch_data:
            self.attach_stft(batch_data, args)
        data = self.prepare(batch_data, args, use_vis)
        if not use_vis:
            return self.ao_forward(data, args)
        return {1: self.train_av_forward1, 2: self.train_av_forward2, 3: self.train_av_forward3}[stage](data, args)
