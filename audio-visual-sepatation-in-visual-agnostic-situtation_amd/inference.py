"""Inference-time wrapper (reference: inference.py:29-160).

Same surface as the reference's inference ``NetWrapper``: ``forward((mag_mix, phase_mix), frames, args, use_vis)``
-> dict with ``pred_masks`` (list of [B,1,256,T]), ``mag_mix`` (warped), ``phase_mix``, ``maps`` and, for the
audio-visual branch, ``match_loss``.  Differences from the training wrapper that are reproduced:

* frames arrive as single images ([B,3,H,W]; a 5-D clip is cut to its first frame, inference.py:64-66) and go
  through ``net_frame.forward`` (no temporal mean);
* a one-element ``frames`` list is a *duet*: the same visual map feeds both sources, and the reference does NOT
  apply ``img_activation`` on that branch (inference.py:69-72);
* the spectrogram is always log-frequency warped to 256 bins (inference.py:48-51), whatever ``args.log_freq``.

The ``share`` fusion of inference.py:94-122 cannot be built by the reference's own factory (get_fusion_net rejects
it) and is not provided.
"""
import torch

from . import kernels as K
from . import lib
from .models import activate


class NetWrapper(torch.nn.Module):
    def __init__(self, nets):
        super().__init__()
        if len(nets) != 2:
            raise NotImplementedError("the motion branch (net_motion: mmaction + private checkpoint) is out of scope")
        self.net_sound, self.net_frame = nets
        self.load_clips = False

    def prepare_inferdata(self, audios, frames, args):
        mag_mix, phase_mix = audios
        lib.require_gpu(mag_mix)
        mag_mix = K.warp((mag_mix.float() + 1e-10).contiguous(), 256, mag_mix.size(3), 1)
        return frames, mag_mix, torch.log(mag_mix).detach(), phase_mix

    @staticmethod
    def _first_frame(frames):
        if frames[0].dim() == 5:
            for n in range(len(frames)):
                frames[n] = frames[n][:, :, 0]
        return frames

    def forward_ao(self, data, args):
        _, mag_mix, log_mag_mix, phase_mix = data
        feat, meta = self.net_sound(log_mag_mix, None)
        pred = activate(feat, args.output_activation).permute(0, 2, 3, 1)
        return {"pred_masks": [pred[..., i].unsqueeze(1) for i in range(2)], "mag_mix": mag_mix,
                "phase_mix": phase_mix, "maps": meta[1]}

    def forward_av(self, data, args):
        N = args.num_mix
        frames, mag_mix, log_mag_mix, phase_mix = data
        duet = len(frames) == 1
        frames = self._first_frame(frames)
        if duet:
            feats = [self.net_frame.forward(frames[0], pool=args.not_pool_vis)] * 2
        else:
            feats = [activate(self.net_frame.forward(frames[n], pool=args.not_pool_vis), args.img_activation)
                     for n in range(N)]
        feat, meta = self.net_sound(log_mag_mix, feats)
        pred = [activate(feat[:, n].unsqueeze(1), args.output_activation) for n in range(N)]
        return {"pred_masks": pred, "mag_mix": mag_mix, "phase_mix": phase_mix,
                "match_loss": meta[0].reshape(1), "maps": meta[1]}

    def forward_avmiximg(self, data, args):
        # inference.py:138-160 (MixVis: frames concatenated along W, one visual pass over the clip)
        frames, mag_mix, log_mag_mix, phase_mix = data
        mix = torch.cat(frames, dim=-1)
        feat_frame = activate(self.net_frame.forward_multiframe(mix, pool=args.not_pool_vis), args.img_activation)
        self._first_frame(frames)
        feat, meta = self.net_sound(log_mag_mix, [feat_frame])
        pred = activate(feat, args.output_activation).permute(0, 2, 3, 1)
        return {"pred_masks": [pred[..., i].unsqueeze(1) for i in range(2)], "phase_mix": phase_mix,
                "mag_mix": mag_mix, "maps": meta[1]}

    def forward(self, mag_mix, frames, args, use_vis=True):
        data = self.prepare_inferdata(mag_mix, frames, args)
        if args.fusion_type == "share":
            raise Exception("Fusion type undefined!")      # what the reference's own builder raises for it
        if not use_vis:
            return self.forward_ao(data, args)
        if args.fusion_type == "MixVis":
            return self.forward_avmiximg(data, args)
        return self.forward_av(data, args)
