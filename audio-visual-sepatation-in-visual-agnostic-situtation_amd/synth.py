"""Seeded synthetic mixtures with the reference loader's batch contract
(dataset/music.py:275-331): per mixture N harmonic-tone waveforms
(6 harmonics, 1/k amplitudes, f0 ~ U(110,1760) Hz at 11025 Hz, attack/decay envelope, a little
noise), each scaled by U(0.5,1.5), clipped to +-1 and divided by N (dataset/base.py:165-169,
music.py:120,127); the mixture is their sum; frames are U(0,1) RGB normalised with the ImageNet
mean/std (dataset/base.py:96-110).  No dataset and no network are needed.
"""
import math

import torch

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def make_waveforms(B, N, aud_len=65535, rate=11025, seed=1234):
    g = torch.Generator().manual_seed(seed)
    t = torch.arange(aud_len, dtype=torch.float64) / rate
    audios = []
    for n in range(N):
        f0 = 110.0 + (1760.0 - 110.0) * torch.rand(B, 1, generator=g, dtype=torch.float64)
        wav = torch.zeros(B, aud_len, dtype=torch.float64)
        for k in range(1, 7):
            wav += torch.sin(2 * math.pi * k * f0 * t[None]) / k
        attack = 0.05 + 0.5 * torch.rand(B, 1, generator=g, dtype=torch.float64)
        decay = 0.5 + 3.0 * torch.rand(B, 1, generator=g, dtype=torch.float64)
        env = torch.clamp(t[None] / attack, max=1.0) * torch.exp(-t[None] / decay)
        wav = 0.4 * wav * env + 1e-3 * torch.randn(B, aud_len, generator=g, dtype=torch.float64)
        scale = 0.5 + torch.rand(B, 1, generator=g, dtype=torch.float64)
        audios.append((torch.clamp(wav * scale, -1.0, 1.0) / N).float())
    mix = torch.stack(audios, 0).sum(0)
    return audios, mix


def make_frames(B, N, T=3, size=224, seed=1234):
    g = torch.Generator().manual_seed(seed + 7)
    mean = torch.tensor(IMAGENET_MEAN).view(1, 3, 1, 1, 1)
    std = torch.tensor(IMAGENET_STD).view(1, 3, 1, 1, 1)
    return [((torch.rand(B, 3, T, size, size, generator=g) - mean) / std) for _ in range(N)]


def make_batch(B, N=2, T=3, size=224, aud_len=65535, seed=1234, device="cpu"):
    """Waveform-level batch: the STFT is left to the consumer (GPU kernel or oracle)."""
    audios, mix = make_waveforms(B, N, aud_len, seed=seed)
    frames = make_frames(B, N, T, size, seed=seed)
    return {"audios": [a.to(device) for a in audios], "audio_mix": mix.to(device),
            "frames": [f.to(device) for f in frames]}
