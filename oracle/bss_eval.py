"""Oracle for BSS-eval SDR / SIR / SAR.  TEST INFRASTRUCTURE ONLY.

The reference scores separation with asteroid.metrics.get_metrics(..., metrics_list=['sdr','sir','sar','si_sdr'])
(main.py:260-266), which (through pb_bss_eval) calls mir_eval.separation.bss_eval_sources(reference, estimate,
compute_permutation=False) and averages over the sources.  asteroid / pb_bss_eval / mir_eval are third-party,
unpinned and not installed: this file restates mir_eval's published algorithm (Vincent et al. 2006, BSS_EVAL 3.0
`bss_decomp_mtifilt`: least-squares projection of each estimate on the span of 512 delayed copies of the true
sources) in numpy float64.  Parity against mir_eval itself is UNPINNED; tests pin it on known-answer cases.
"""
import numpy as np

FLEN = 512


def _next_pow2(n):
    return 1 << int(np.ceil(np.log2(n)))


def _project(refs, est, flen=FLEN):
    """Least-squares projection of `est` on the subspace spanned by delayed versions (0..flen-1) of the rows of `refs`."""
    nsrc, nsampl = refs.shape
    refs = np.hstack((refs, np.zeros((nsrc, flen - 1))))
    est = np.hstack((est, np.zeros(flen - 1)))
    n_fft = _next_pow2(nsampl + flen - 1)
    sf = np.fft.rfft(refs, n=n_fft, axis=1)
    sef = np.fft.rfft(est, n=n_fft)
    G = np.zeros((nsrc * flen, nsrc * flen))
    k = np.arange(flen)
    lag = (k[:, None] - k[None, :]) % n_fft                       # toeplitz(c = ss[0], ss[-1], .., r = ss[:flen])
    for i in range(nsrc):
        for j in range(nsrc):
            ss = np.fft.irfft(sf[i] * np.conj(sf[j]), n=n_fft)
            G[i * flen:(i + 1) * flen, j * flen:(j + 1) * flen] = ss[(-lag) % n_fft]
    D = np.zeros(nsrc * flen)
    for i in range(nsrc):
        ssef = np.fft.irfft(sf[i] * np.conj(sef), n=n_fft)
        D[i * flen:(i + 1) * flen] = ssef[(-k) % n_fft]
    try:
        C = np.linalg.solve(G, D)
    except np.linalg.LinAlgError:
        C = np.linalg.lstsq(G, D, rcond=None)[0]
    C = C.reshape(nsrc, flen)
    out = np.zeros(nsampl + flen - 1)
    for i in range(nsrc):
        out += np.convolve(C[i], refs[i])[:nsampl + flen - 1]
    return out


def bss_eval_sources(refs, ests, flen=FLEN):
    """refs, ests: [nsrc, nsampl] (estimate j is scored against reference j: compute_permutation=False).
    Returns (sdr, sir, sar) arrays of length nsrc, in dB."""
    refs, ests = np.asarray(refs, np.float64), np.asarray(ests, np.float64)
    nsrc, nsampl = refs.shape
    sdr, sir, sar = np.zeros(nsrc), np.zeros(nsrc), np.zeros(nsrc)
    for j in range(nsrc):
        s_true = np.hstack((refs[j], np.zeros(flen - 1)))
        p_one = _project(refs[j:j + 1], ests[j], flen)
        p_all = _project(refs, ests[j], flen)
        e_spat = p_one - s_true
        e_interf = p_all - p_one
        e_artif = np.hstack((ests[j], np.zeros(flen - 1))) - p_all
        s_filt = s_true + e_spat
        sdr[j] = 10 * np.log10(np.sum(s_filt ** 2) / np.sum((e_interf + e_artif) ** 2))
        sir[j] = 10 * np.log10(np.sum(s_filt ** 2) / np.sum(e_interf ** 2))
        sar[j] = 10 * np.log10(np.sum((s_filt + e_interf) ** 2) / np.sum(e_artif ** 2))
    return sdr, sir, sar
