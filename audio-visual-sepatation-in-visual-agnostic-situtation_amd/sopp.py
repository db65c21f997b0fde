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
from .net_wrapper import FlatSGD, NetWrapper as _StepBase


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


# ---- the three-stage schedule and its optimizer (SoP++/main.py:593-606, 670-688) ------------------------------------------
def stage_of(i, train_steps):
    """SoP++/main.py:674-679: stage 1 while i < train_steps[0], stage 2 on [train_steps[0], train_steps[1]), stage 3 on
    [train_steps[1], train_steps[2]] (both ends included).  Past train_steps[2] the reference leaves `stage` unbound
    and dies with UnboundLocalError; here that is a ValueError that says why."""
    t0, t1, t2 = train_steps
    if i < t0:
        return 1
    if t0 <= i < t1:
        return 2
    if t1 <= i <= t2:
        return 3
    raise ValueError(f"iteration {i} is past the last stage boundary train_steps[2] = {t2}")


def create_optimizer(nets, args, process_group=None, world_size=1):
    """SoP++/main.py:593-606: SGD groups (sound, lr_sound), (synthesizer, lr_synthesizer), (attention module,
    lr_synthesizer) and, unless --fix_vis, (frame features, lr_frame), (frame fc, lr_sound) — as one FlatSGD."""
    net_sound, net_frame, net_synthesizer, net_pit = nets
    groups = [{"params": list(net_sound.parameters()), "lr": args.lr_sound, "name": "sound"},
              {"params": list(net_synthesizer.parameters()), "lr": args.lr_synthesizer, "name": "synthesizer"},
              {"params": list(net_pit.parameters()), "lr": args.lr_synthesizer, "name": "pit"}]
    if not args.fix_vis:
        groups += [{"params": list(net_frame.features.parameters()), "lr": args.lr_frame, "name": "frame_features"},
                   {"params": list(net_frame.fc.parameters()), "lr": args.lr_sound, "name": "frame_fc"}]
    groups = [g for g in groups if any(p.requires_grad for p in g["params"])]     # AttModel / Bias may hold no parameters
    return FlatSGD(groups, momentum=args.beta1, weight_decay=args.weight_decay, process_group=process_group,
                   world_size=world_size)


def train_step_3stage(model, batch, optimizer, use_vis, i, args):
    """SoP++/main.py:670-688: zero_grad -> forward at the stage iteration i belongs to -> err.mean().backward() ->
    optimizer.step(); returns (err.item(), match_loss.item() or None).  torch.optim.SGD skips parameters without a
    gradient; FlatSGD reproduces that through its per-group `no_grad` detection (zero_grad(set_to_none) semantics)."""
    torch.set_grad_enabled(True)
    model.train()
    for p in model.parameters():
        p.grad = None
    stage = stage_of(i, args.train_steps)
    err, outputs = model.forward(batch, args, use_vis, stage)
    err = err.mean()
    err.backward()
    optimizer.step()
    optimizer.zero_grad()
    match_loss = outputs["match_loss"].mean().item() if use_vis else None
    return err.item(), match_loss
