"""Oracle step definition: prepare / forward_av / forward_ao / train_step.

TEST INFRASTRUCTURE ONLY.  Reference: main.py:39-192 (NetWrapper,
forward_avmiximg), :536-569 (create_optimizer, train_step), utils.py:12-26
(warpgrid).
"""
import numpy as np
import torch
import torch.nn.functional as F

from .nets import activate


def warpgrid(bs, HO, WO, warp=True):
    # utils.py:12-26: float64 grid, cast to fp32 at the end
    x = np.linspace(-1, 1, WO)
    y = np.linspace(-1, 1, HO)
    xv, yv = np.meshgrid(x, y)
    if warp:
        gy = (np.power(21, (yv + 1) / 2) - 11) / 10
    else:
        gy = np.log(yv * 10 + 11) / np.log(21) * 2 - 1
    grid = np.zeros((bs, HO, WO, 2))
    grid[..., 0] = xv
    grid[..., 1] = gy
    return grid.astype(np.float32)


def prepare(batch, args):
    """main.py:51-95.  Returns mags(warped), mag_mix, log_mag_mix, gt_masks, weights.
    Like the reference, replaces batch['mags'][n] by the warped tensors."""
    mag_mix = batch["mag_mix"] + 1e-10
    mags = batch["mags"]
    N = args.num_mix
    B, T = mag_mix.size(0), mag_mix.size(3)
    if args.log_freq:
        grid = torch.from_numpy(warpgrid(B, 256, T, warp=True)).to(mag_mix.device)
        mag_mix = F.grid_sample(mag_mix, grid, align_corners=False)
        for n in range(N):
            mags[n] = F.grid_sample(mags[n], grid, align_corners=False)
    if args.weighted_loss:
        weights = torch.clamp(torch.log1p(mag_mix), 1e-3, 10)
    else:
        weights = torch.ones_like(mag_mix)
    gt = []
    for n in range(N):
        if args.binary_mask:
            gt.append((mags[n] > 0.5 * mag_mix).float())
        else:
            gt.append(torch.clamp(mags[n] / mag_mix, 0.0, 5.0))
    log_mag_mix = torch.log(mag_mix).detach()
    return mags, mag_mix, log_mag_mix, gt, weights


class NetWrapper(torch.nn.Module):
    def __init__(self, nets, crit_ao, crit_av):
        super().__init__()
        self.net_sound, self.net_frame = nets
        self.crit_ao = crit_ao
        self.crit_av = crit_av

    def forward_ao(self, data, args):
        # main.py:97-111
        mags, mag_mix, log_mag_mix, gt_masks, weight = data
        feat, *_ = self.net_sound(log_mag_mix, None)
        pred = activate(feat, args.output_activation).permute(0, 2, 3, 1)
        gt = torch.stack(gt_masks, -1)[:, 0]
        S = len(gt_masks)      # main.py:103 hard-codes 2; build-defined generalisation: one weight copy per target
        w2 = torch.stack([weight[:, 0]] * S, -1)
        err, perms = self.crit_ao(pred, gt, w2)
        err = err.mean()
        ordered = self.crit_ao.reorder_tensor(pred, perms)
        return err, {"pred_masks": [ordered[..., i].unsqueeze(1) for i in range(S)],
                     "gt_masks": [gt[..., i].unsqueeze(1) for i in range(S)],
                     "mag_mix": mag_mix, "mags": mags, "weight": w2, "perms": perms}

    def visual(self, frames, args):
        out = []
        for n in range(args.num_mix):
            f = self.net_frame.forward_multiframe(frames[n], pool=args.not_pool_vis)
            out.append(activate(f, args.img_activation))
        return out

    def forward_av(self, data, frames, args):
        # main.py:113-148: two U-Net passes (reversed, then natural visual order)
        mags, mag_mix, log_mag_mix, gt_masks, weight = data
        N = args.num_mix
        feats = self.visual(frames, args)
        total_match = 0
        errs = []
        for order in (slice(None, None, -1), slice(None)):
            feat_sound, meta = self.net_sound(log_mag_mix, feats[order])
            pred = [activate(feat_sound[:, n].unsqueeze(1), args.output_activation) for n in range(N)]
            errs.append(self.crit_av(pred, gt_masks[order], weight).reshape(1))
            total_match = total_match + meta[0]
        err = ((errs[0] + errs[1]) / 2 + args.match_weight * total_match).reshape(1)
        return err, {"pred_masks": pred, "gt_masks": gt_masks, "mag_mix": mag_mix, "mags": mags,
                     "weight": weight, "match_loss": total_match.reshape(1), "att_maps": meta[1],
                     "logits": feat_sound}

    def forward_avmiximg(self, data, frames, args):
        # main.py:162-192 (MixVis)
        mags, mag_mix, log_mag_mix, gt_masks, weight = data
        mix = torch.cat(frames, dim=-1)
        feat_frame = activate(self.net_frame.forward_multiframe(mix, pool=args.not_pool_vis),
                              args.img_activation)
        feat_sound, meta = self.net_sound(log_mag_mix, [feat_frame])
        pred = activate(feat_sound, args.output_activation).permute(0, 2, 3, 1)
        gt = torch.stack(gt_masks, -1)[:, 0]
        # main.py:181 passes the un-stacked [B,1,F,T] weight, which makes PitWrapper's
        # binary_cross_entropy raise a broadcast error: the MixVis branch cannot run as shipped.
        # Build-defined repair: the per-target weight stacking of forward_ao (main.py:103).
        w2 = torch.stack([weight[:, 0]] * 2, -1)
        err, perms = self.crit_ao(pred, gt, w2)
        err = err.mean().reshape(1)
        pred = self.crit_ao.reorder_tensor(pred, perms)
        err = err + meta[0] * args.match_weight
        return err, {"pred_masks": [pred[..., i].unsqueeze(1) for i in range(2)],
                     "gt_masks": [gt[..., i].unsqueeze(1) for i in range(2)],
                     "mag_mix": mag_mix, "mags": mags, "weight": weight,
                     "match_loss": meta[0], "maps": meta[1]}

    def forward(self, batch, args, use_vis, is_share=False):
        data = prepare(batch, args)
        if use_vis:
            if args.fusion_type == "MixVis":
                return self.forward_avmiximg(data, batch["frames"], args)
            return self.forward_av(data, batch["frames"], args)
        return self.forward_ao(data, args)


def create_optimizer(nets, args):
    # main.py:536-547
    net_sound, net_frame = nets
    groups = [{"params": net_sound.parameters(), "lr": args.lr_sound},
              {"params": net_frame.fc.parameters(), "lr": args.lr_sound}]
    if not args.fix_vis:
        groups.append({"params": net_frame.features.parameters(), "lr": args.lr_frame})
    return torch.optim.SGD(groups, momentum=args.beta1, weight_decay=args.weight_decay)


def train_step(model, batch, optimizer, use_vis, args):
    # main.py:557-569
    torch.set_grad_enabled(True)
    model.train()
    model.zero_grad()
    err, outputs = model.forward(batch, args, use_vis)
    err = err.mean()
    err.backward()
    optimizer.step()
    match = outputs["match_loss"].mean().item() if use_vis else None
    return err.item(), match, outputs
