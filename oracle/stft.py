"""Oracle STFT / iSTFT.  TEST INFRASTRUCTURE ONLY.

The reference computes the STFT with librosa in its DataLoader workers
(dataset/base.py:142-147: ``librosa.stft(audio, n_fft=1022, hop_length=256)``)
and reconstructs with ``librosa.istft(spec, hop_length=256)`` (utils.py:101-104).
librosa is a third-party dependency that is absent from /root/reference and is
not installed; its version is unpinned (no requirements file).  This file
restates librosa's published algorithm:

* stft: ``win_length = n_fft``, periodic Hann window
  (``scipy.signal.get_window('hann', n_fft, fftbins=True)``), ``center=True``
  -> pad n_fft//2 on both sides, frames every ``hop``, one-sided rFFT,
  ``n_frames = 1 + len // hop``; output complex64 ``[1 + n_fft//2, n_frames]``.
  Pad mode is version dependent (``reflect`` before librosa 0.10, ``constant``
  from 0.10); the build documents ``reflect`` as its choice (the reference dates
  from the Python-3.7 / librosa-0.8 era) and supports both.
* istft: windowed inverse rFFT, overlap-add, division by the window-sum-square
  envelope where it exceeds ``tiny(float32)``, trim ``n_fft//2`` on both ends.

Parity for this file is UNPINNED against librosa itself; it is pinned against
numpy.fft (an independent DFT) in tests/test_oracle_stft.py.
"""
import numpy as np


def hann_periodic(n):
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)).astype(np.float64)


def stft(audio, n_fft=1022, hop=256, pad_mode="reflect"):
    audio = np.asarray(audio, dtype=np.float32)
    pad = n_fft // 2
    y = np.pad(audio, (pad, pad), mode=pad_mode)
    n_frames = 1 + (len(y) - n_fft) // hop
    win = hann_periodic(n_fft).astype(np.float32)
    idx = np.arange(n_fft)[:, None] + hop * np.arange(n_frames)[None, :]
    frames = y[idx] * win[:, None]
    return np.fft.rfft(frames, axis=0).astype(np.complex64)


def stft_mag_phase(audio, n_fft=1022, hop=256, pad_mode="reflect"):
    s = stft(audio, n_fft, hop, pad_mode)
    return np.abs(s).astype(np.float32), np.angle(s).astype(np.float32)


def istft(spec, hop=256, length=None):
    n_bins, n_frames = spec.shape
    n_fft = 2 * (n_bins - 1)
    win = hann_periodic(n_fft)
    frames = np.fft.irfft(spec.astype(np.complex128), n=n_fft, axis=0) * win[:, None]
    out_len = n_fft + hop * (n_frames - 1)
    y = np.zeros(out_len)
    wss = np.zeros(out_len)
    for t in range(n_frames):
        y[t * hop:t * hop + n_fft] += frames[:, t]
        wss[t * hop:t * hop + n_fft] += win ** 2
    ok = wss > np.finfo(np.float32).tiny
    y[ok] /= wss[ok]
    y = y[n_fft // 2: out_len - n_fft // 2]
    if length is not None:
        y = y[:length]
    return y.astype(np.float32)


def istft_reconstruction(mag, phase, hop=256):
    # utils.py:101-104
    spec = mag.astype(np.complex64) * np.exp(1j * phase)
    return np.clip(istft(spec, hop), -1.0, 1.0)


def si_sdr(est, ref):
    """asteroid / pb_bss_eval SI-SDR: 10 log10(|a s|^2 / |s_hat - a s|^2), a = <s_hat,s>/<s,s> (no mean removal)."""
    est, ref = np.asarray(est, np.float64), np.asarray(ref, np.float64)
    a = np.dot(est, ref) / np.dot(ref, ref)
    proj = a * ref
    return 10 * np.log10(np.sum(proj ** 2) / np.sum((est - proj) ** 2))


def sdr_plain(est, ref):
    est, ref = np.asarray(est, np.float64), np.asarray(ref, np.float64)
    return 10 * np.log10(np.sum(ref ** 2) / np.sum((ref - est) ** 2))
