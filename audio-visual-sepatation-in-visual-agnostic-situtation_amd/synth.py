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


def make_batch_on_device(B, N=2, T=3, size=224, aud_len=65535, seed=1234, device="cuda", rate=11025, informative_frames=True):
    """The same mixtures drawn on the device (its own generator: another stream of random numbers than make_batch, but
    a pure function of `seed`): the training / validation stream of the "SDR on synthetic val" run (bench.py), where a
    fresh batch per step must cost milliseconds.  Also returns the fundamentals `f0` [N,B].

    informative_frames: the reference's premise is that a source's frames identify it.  Uniform-noise frames say nothing
    about the sound, so the audio-visual pass could only learn the permutation-symmetric answer; here a source's frames
    are a vertical grating whose spatial frequency and mean colour encode log f0 (plus a per-frame phase and a little
    noise), ImageNet-normalised like the loader's frames."""
    dev = torch.device(device)
    g = torch.Generator(device=dev).manual_seed(seed)

    def rand(*shape):
        return torch.rand(*shape, generator=g, device=dev, dtype=torch.float32)
    t = torch.arange(aud_len, dtype=torch.float32, device=dev) / rate
    mean = torch.tensor(IMAGENET_MEAN, device=dev).view(1, 3, 1, 1, 1)
    std = torch.tensor(IMAGENET_STD, device=dev).view(1, 3, 1, 1, 1)
    audios, frames, f0s = [], [], []
    for n in range(N):
        f0 = 110.0 + (1760.0 - 110.0) * rand(B, 1)
        wav = torch.zeros(B, aud_len, device=dev)
        phase = 2 * math.pi * (f0.double() * t.double()[None])          # the phase in float64: 1760 Hz x 6 s x 6 harmonics
        for k in range(1, 7):
            wav += torch.sin(k * phase).float() / k
        attack, decay = 0.05 + 0.5 * rand(B, 1), 0.5 + 3.0 * rand(B, 1)
        env = torch.clamp(t[None] / attack, max=1.0) * torch.exp(-t[None] / decay)
        wav = 0.4 * wav * env + 1e-3 * torch.randn(B, aud_len, generator=g, device=dev)
        scale = 0.5 + rand(B, 1)
        audios.append(torch.clamp(wav * scale, -1.0, 1.0) / N)
        f0s.append(f0[:, 0])
        if informative_frames:
            u = (torch.log2(f0 / 110.0) / 4.0).view(B, 1, 1, 1, 1)                      # 0..1 over the four octaves
            xs = torch.arange(size, device=dev, dtype=torch.float32).view(1, 1, 1, 1, size) / size
            ph = 2 * math.pi * rand(B, 1, T, 1, 1)
            base = torch.cat([u, 1.0 - u, 0.5 * torch.ones_like(u)], 1)                  # mean colour
            img = base + 0.25 * torch.sin(2 * math.pi * (4.0 + 28.0 * u) * xs + ph) + 0.1 * (rand(B, 3, T, size, size) - 0.5)
            img = img.expand(B, 3, T, size, size).clamp(0.0, 1.0)
        else:
            img = rand(B, 3, T, size, size)
        frames.append(((img - mean) / std).contiguous())
    mix = torch.stack(audios, 0).sum(0)
    return {"audios": audios, "audio_mix": mix, "frames": frames, "f0": torch.stack(f0s, 0)}
