"""Oracle for the inference-time wrapper.  TEST INFRASTRUCTURE ONLY.

Reference: inference.py:29-160 (NetWrapper.prepare_inferdata / forward_ao / forward_av / forward_avmiximg).
The modules it calls are the pinned oracle nets; this file only restates the few lines of glue (single-frame
visual forward, duet = same map twice WITHOUT img_activation, always-on log-frequency warp).
"""
import torch
import torch.nn.functional as F

from .nets import activate
from .step import warpgrid


def forward(nets, audios, frames, args, use_vis=True):
    net_sound, net_frame = nets
    mag_mix, phase_mix = audios
    mag_mix = mag_mix + 1e-10
    B, T = mag_mix.size(0), mag_mix.size(3)
    grid = torch.from_numpy(warpgrid(B, 256, T, warp=True))
    mag_mix = F.grid_sample(mag_mix, grid, align_corners=False)           # inference.py:48-51
    log_mag_mix = torch.log(mag_mix).detach()
    N = args.num_mix
    if not use_vis:                                                          # :55-60
        feat, meta = net_sound(log_mag_mix, None)
        pred = activate(feat, args.output_activation).permute(0, 2, 3, 1)
        return {"pred_masks": [pred[..., i].unsqueeze(1) for i in range(2)], "mag_mix": mag_mix, "maps": meta[1]}
    if args.fusion_type == "MixVis":                                         # :138-160
        mix = torch.cat(frames, dim=-1)
        ff = activate(net_frame.forward_multiframe(mix, pool=args.not_pool_vis), args.img_activation)
        feat, meta = net_sound(log_mag_mix, [ff])
        pred = activate(feat, args.output_activation).permute(0, 2, 3, 1)
        return {"pred_masks": [pred[..., i].unsqueeze(1) for i in range(2)], "mag_mix": mag_mix, "maps": meta[1]}
    frames = [f[:, :, 0] if f.dim() == 5 else f for f in frames]            # :64-66
    if len(frames) == 1:                                                     # duet :69-72
        feats = [net_frame.forward(frames[0], pool=args.not_pool_vis)] * 2
    else:
        feats = [activate(net_frame.forward(frames[n], pool=args.not_pool_vis), args.img_activation) for n in range(N)]
    feat, meta = net_sound(log_mag_mix, feats)
    pred = [activate(feat[:, n].unsqueeze(1), args.output_activation) for n in range(N)]
    return {"pred_masks": pred, "mag_mix": mag_mix, "match_loss": meta[0].reshape(1), "maps": meta[1]}
