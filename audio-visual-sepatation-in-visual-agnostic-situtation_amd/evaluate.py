"""Evaluation path on the GPU (reference: main.py:197-286 `calc_metrics`):
un-warp the predicted masks (grid_sample on the inverse log-frequency grid, main.py:216-220), threshold
(binary masks), mask x mixture magnitude, iSTFT with the mixture phase (utils.py:101-104), then the
separation ratios against the ground-truth waveforms.

Built: SI-SDR (asteroid/pb_bss_eval definition: projection of the estimate on the reference, no mean
removal), the plain SDR 10*log10(|s|^2 / |s - s_hat|^2), and BSS-eval SDR / SIR / SAR with mir_eval's
semantics (bss_eval.py: 512-tap projections, batched on the device).  The reference runs all of this per
sample on the CPU (librosa + mir_eval); here the waveforms are 3 launches per batch.
"""
import torch

from . import kernels as K
from . import lib
from .lib import call, ptr


def sdr_sums(est, ref):
    """est [R,L'], ref [R,L''] (row-major, first L = min columns used) -> fp64 [R,3] = (<e,r>, <r,r>, <e,e>)."""
    R, L = est.shape[0], min(est.shape[1], ref.shape[1])
    sums = torch.zeros((R, 3), dtype=torch.float64, device=est.device)
    call("avsep_sdr_sums", ptr(est), ptr(ref), R, L, est.stride(0), ref.stride(0), ptr(sums))
    return sums


def si_sdr_from_sums(s, eps=1e-30):
    er, rr, ee = s[:, 0], s[:, 1], s[:, 2]
    proj = er * er / rr.clamp_min(eps)                 # |projection|^2
    return 10.0 * torch.log10(proj.clamp_min(eps) / (ee - proj).clamp_min(eps))


def sdr_from_sums(s, eps=1e-30):
    er, rr, ee = s[:, 0], s[:, 1], s[:, 2]
    return 10.0 * torch.log10(rr.clamp_min(eps) / (rr - 2 * er + ee).clamp_min(eps))


def reconstruct(batch_data, outputs, args, stft_plan=None):
    """Predicted waveforms [N, B, L] from the predicted masks + mixture magnitude/phase (main.py:205-246)."""
    mag_mix, phase_mix = batch_data["mag_mix"], batch_data["phase_mix"]
    lib.require_gpu(mag_mix)
    N, B = args.num_mix, mag_mix.shape[0]
    plan = stft_plan or K.Stft(mag_mix.device, args.stft_frame, args.stft_hop, getattr(args, "stft_pad_mode", "reflect"))
    wavs = []
    for n in range(N):
        m = outputs["pred_masks"][n].detach().float().contiguous()
        if args.log_freq:
            m = K.warp(m, args.stft_frame // 2 + 1, m.shape[3], 0)          # warp=False grid: the un-warp
        if args.binary_mask:
            m = (m > args.mask_thres).float()
        mag = (mag_mix.float() * m)[:, 0].contiguous()
        wavs.append(plan.istft(mag, phase_mix[:, 0].float().contiguous()).clamp_(-1.0, 1.0))
    return torch.stack(wavs, 0)


def calc_metrics(batch_data, outputs, args, stft_plan=None, bss=True):
    """Per-(sample, source) [B,N] 'si_sdr', 'sdr_plain' and — with bss=True — BSS-eval 'sdr', 'sir', 'sar' (what the
    reference's get_metrics reports, main.py:260-266), plus their batch means ('<key>_mean')."""
    from . import bss_eval
    pred = reconstruct(batch_data, outputs, args, stft_plan)                 # [N,B,L]
    N, B, L = pred.shape
    silent = (pred == 0).all(-1)                                             # main.py:248-249
    if bool(silent.any()):
        pred = torch.where(silent[..., None], 0.01 * torch.rand_like(pred), pred)
    gts = torch.stack([a.float() for a in batch_data["audios"][:N]], 0)      # [N,B,audLen]
    s = sdr_sums(pred.reshape(N * B, L), gts.reshape(N * B, -1).contiguous())
    out = {"si_sdr": si_sdr_from_sums(s).view(N, B).t(), "sdr_plain": sdr_from_sums(s).view(N, B).t(), "pred_wavs": pred}
    if bss:
        sdr, sir, sar = bss_eval.bss_eval_sources(gts[..., :L].permute(1, 0, 2), pred.permute(1, 0, 2))
        out.update(sdr=sdr.float(), sir=sir.float(), sar=sar.float())
    else:
        out["sdr"] = out["sdr_plain"]
    for k in [k for k in out if k != "pred_wavs"]:
        out[k + "_mean"] = out[k].mean()
    return out
