"""BSS-eval SDR / SIR / SAR on the GPU.

The reference scores separation with asteroid's `get_metrics(..., ['sdr','sir','sar','si_sdr'])` (main.py:260-266),
i.e. mir_eval.separation.bss_eval_sources(reference, estimate, compute_permutation=False): each estimate is projected
(least squares) on the span of 512 delayed copies of (a) its own true source and (b) all true sources; the three
residuals give SDR, SIR, SAR (Vincent et al. 2006, `bss_decomp_mtifilt`).  mir_eval runs this per sample in numpy
on the CPU and dominates the reference's evaluate(); here the whole batch is one set of batched FFT correlations,
two batched float64 solves (512x512 per source, 512S x 512S for all sources) and FFT filters on the device.
torch.fft / torch.linalg are the numerical library here (eval-only path, not part of the train step).
"""
import torch

FLEN = 512


def _next_pow2(n):
    return 1 << (int(n) - 1).bit_length()


def bss_eval_sources(refs, ests, flen=FLEN):
    """refs, ests: [B, S, L] (estimate j against reference j).  Returns sdr, sir, sar: float64 [B, S] in dB."""
    refs, ests = refs.double(), ests.double()
    B, S, L = refs.shape
    dev = refs.device
    Lp = L + flen - 1
    n = _next_pow2(Lp)
    sf = torch.fft.rfft(refs, n=n)                                         # [B,S,F]   (zero padding implied)
    sef = torch.fft.rfft(ests, n=n)
    k = torch.arange(flen, device=dev)
    lag = (k[None, :] - k[:, None]) % n                                    # T[a,b] = ss[(b-a) mod n]
    ss = torch.fft.irfft(sf[:, :, None] * sf[:, None].conj(), n=n)         # [B,S,S,n]   corr(ref_i, ref_j)
    G = ss[..., lag]                                                       # [B,S,S,flen,flen]
    ssef = torch.fft.irfft(sf[:, None] * sef[:, :, None].conj(), n=n)      # [B,E,S,n]   corr(ref_i, est_e)
    D = ssef[..., (-k) % n]                                                # [B,E,S,flen]

    def project(Gm, Dm, idx):
        """Gm [B,M,M], Dm [B,M,E'] -> filters -> sum_i conv(C_i, ref_i) for the sources in idx: [B,E',Lp]."""
        C, info = torch.linalg.solve_ex(Gm, Dm)                            # [B, len(idx)*flen, E']
        if bool((info != 0).any()):    # singular Gram matrix (e.g. a silent source): least squares, as mir_eval does
            bad = (info != 0).nonzero().flatten().tolist()
            for b in bad:
                C[b] = torch.linalg.lstsq(Gm[b].cpu(), Dm[b].cpu()).solution.to(C.device)
        C = C.view(B, len(idx), flen, -1).permute(0, 3, 1, 2)              # [B,E',S',flen]
        Cf = torch.fft.rfft(C, n=n)                                        # [B,E',S',F]
        return torch.fft.irfft((Cf * sf[:, None, idx]).sum(2), n=n)[..., :Lp]

    # (b) all sources: one 512S x 512S system per sample, both estimates as right-hand sides
    G_all = G.permute(0, 1, 3, 2, 4).reshape(B, S * flen, S * flen)
    D_all = D.reshape(B, S, S * flen).transpose(1, 2)                      # [B, S*flen, E]
    p_all = project(G_all, D_all, list(range(S)))                          # [B,E,Lp]
    # (a) own source only
    p_one = torch.stack([project(G[:, j, j], D[:, j, j].unsqueeze(-1), [j])[:, 0] for j in range(S)], 1)
    pad = lambda t: torch.nn.functional.pad(t, (0, flen - 1))             # noqa: E731
    s_true, est_p = pad(refs), pad(ests)
    e_spat = p_one - s_true
    e_interf = p_all - p_one
    e_artif = est_p - p_all
    s_filt = s_true + e_spat
    en = lambda t: (t * t).sum(-1)                                         # noqa: E731
    sdr = 10 * torch.log10(en(s_filt) / en(e_interf + e_artif))
    sir = 10 * torch.log10(en(s_filt) / en(e_interf))
    sar = 10 * torch.log10(en(s_filt + e_interf) / en(e_artif))
    return sdr, sir, sar
