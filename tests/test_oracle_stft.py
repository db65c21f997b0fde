"""not-gpu: the STFT/iSTFT restatement (oracle/stft.py) pinned against numpy.fft and against the
librosa conventions it states (librosa itself is absent: parity with it is unpinned)."""
import numpy as np

from oracle import stft as S


def test_shapes_and_dft():
    rng = np.random.default_rng(0)
    wav = rng.standard_normal(65535).astype(np.float32) * 0.1
    spec = S.stft(wav)
    assert spec.shape == (512, 256) and spec.dtype == np.complex64
    # frame 10, bin 37 against a direct O(N) DFT of the windowed frame
    y = np.pad(wav, (511, 511), mode="reflect")
    fr = y[10 * 256:10 * 256 + 1022].astype(np.float64) * S.hann_periodic(1022)
    ref = np.sum(fr * np.exp(-2j * np.pi * 37 * np.arange(1022) / 1022))
    assert abs(spec[37, 10] - ref) < 1e-3 * max(1.0, abs(ref))
    mag, ph = S.stft_mag_phase(wav)
    assert mag.dtype == np.float32 and np.allclose(mag * np.exp(1j * ph), spec, atol=1e-4)


def test_window_is_periodic_hann():
    w = S.hann_periodic(1022)
    assert w[0] == 0 and abs(w[511] - 1) < 1e-12 and abs(w[1] - w[1021]) < 1e-12


def test_pad_modes_differ_only_at_edges():
    rng = np.random.default_rng(1)
    wav = rng.standard_normal(65535).astype(np.float32)
    a, b = S.stft(wav, pad_mode="reflect"), S.stft(wav, pad_mode="constant")
    assert np.allclose(a[:, 2:-2], b[:, 2:-2], atol=1e-4) and not np.allclose(a[:, 0], b[:, 0], atol=1e-3)


def test_istft_round_trip_and_clip():
    rng = np.random.default_rng(2)
    wav = (rng.standard_normal(65535) * 0.2).astype(np.float32)
    mag, ph = S.stft_mag_phase(wav)
    back = S.istft_reconstruction(mag, ph)
    assert back.shape == (65280,) and np.abs(back).max() <= 1.0
    assert np.abs(back[1024:64000] - np.clip(wav, -1, 1)[1024:64000]).max() < 1e-4
