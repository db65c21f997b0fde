"""BSS-eval SDR / SIR / SAR on the GPU.

The reference scores separation with asteroid's `get_metrics(..., ['sdr','sir','sar','si_sdr'])` (main.py:260-266),
i.e. mir_eval.separation.bss_eval_sources(reference, estimate, compute_permutation=False): each estimate is projected
(least squares) on the span of 512 delayed copies of (a) its own true source and (b) all true sources; the three
residuals give SDR, SIR, SAR (Vincent et al. 2006, `bss_decomp_mtifilt`).  mir_eval runs this per sample in numpy
on the CPU and dominates the reference's evaluate(); here the whole batch goes through three float64 kernels of this
library (csrc/bsseval.hip): direct lagged correlations (the block-Toeplitz Gram matrices and the right-hand sides),
one LU-with-partial-pivoting solve per system (512 x 512 per source, 512 S x 512 S for all sources — numpy.linalg.solve's
algorithm, one workgroup each) and the FIR projections.  Since round 4 no FFT / solver library is involved; what is
left to torch are the residual energies and the logarithms (a few elementwise ops).  Exactly singular Gram matrices (a
silent source) take mir_eval's fallback: minimum-norm least squares, on the host, for that system only.
"""
import torch

from . import lib
from .lib import call, ptr

FLEN = 512


def _gram(R, b, srcs, flen):
    """The Gram matrix of the sources `srcs` of sample b from the lagged correlations (host-side fallback only)."""
    k = torch.arange(flen, device=R.device)
    lag = (k[None, :] - k[:, None]) + flen - 1                              # [a, c] -> c - a + flen - 1
    blocks = [[R[b, i, j][lag] for j in srcs] for i in srcs]
    return torch.cat([torch.cat(row, 1) for row in blocks], 0)


def _solve(R, D, B, S, E, flen, mode):
    """Filters of every system of one kind: mode 0 -> [B, S*flen, E], mode 1 -> [B*S, flen, 1]."""
    L = lib.load()
    M, nsys, nrhs = (S * flen, B, E) if mode == 0 else (flen, B * S, 1)
    nbytes = L.avsep_bss_solve_workspace_bytes(B, S, flen, mode)
    ws = torch.empty((nbytes // 8,), dtype=torch.float64, device=R.device)
    C = torch.empty((nsys, M, nrhs), dtype=torch.float64, device=R.device)
    info = torch.empty((nsys,), dtype=torch.int32, device=R.device)
    call("avsep_bss_solve", ptr(R), ptr(D), B, S, E, flen, mode, ptr(ws), nbytes, ptr(C), ptr(info))
    bad = info.nonzero().flatten().tolist()                                  # (one host sync; the metric is eval-only)
    for s in bad:       # singular Gram matrix (e.g. a silent source): minimum-norm least squares, as mir_eval does
        b = s if mode == 0 else s // S
        srcs = list(range(S)) if mode == 0 else [s % S]
        G = _gram(R, b, srcs, flen).cpu()
        rhs = (D[b].reshape(E, S * flen).t() if mode == 0 else D[b, s % S, s % S].reshape(flen, 1)).cpu()
        # driver gelsd = numpy.linalg.lstsq's (what mir_eval falls back to).  torch's CPU default, gelsy, returned wrong
        # minimum-norm solutions for this exactly rank-deficient system on some calls (errors of 0.2-0.9 in the filters on a
        # 128-thread host, 1e-16 on others: scratch probe of round 5); gelsd / gelss / an eigendecomposition agree to 3e-15
        C[s] = torch.linalg.lstsq(G, rhs.contiguous(), driver="gelsd").solution.to(C.device)
    return C


def bss_eval_sources(refs, ests, flen=FLEN):
    """refs, ests: [B, S, L] (estimate j against reference j).  Returns sdr, sir, sar: float64 [B, S] in dB."""
    lib.require_gpu(refs)
    refs, ests = refs.double().contiguous(), ests.double().contiguous()
    B, S, L = refs.shape
    E = ests.shape[1]
    if E != S:
        raise lib.AvsepError("bss_eval_sources scores estimate j against reference j: as many estimates as references")
    dev = refs.device
    Lp = L + flen - 1
    R = torch.empty((B, S, S, 2 * flen - 1), dtype=torch.float64, device=dev)
    D = torch.empty((B, E, S, flen), dtype=torch.float64, device=dev)
    call("avsep_bss_corr", ptr(refs), ptr(ests), B, S, E, L, flen, ptr(R), ptr(D))
    p_all = torch.empty((B, E, Lp), dtype=torch.float64, device=dev)        # (b) projection on all sources
    p_one = torch.empty((B, S, Lp), dtype=torch.float64, device=dev)        # (a) on the own source only
    call("avsep_bss_project", ptr(refs), ptr(_solve(R, D, B, S, E, flen, 0)), B, S, E, L, flen, 0, ptr(p_all))
    call("avsep_bss_project", ptr(refs), ptr(_solve(R, D, B, S, E, flen, 1)), B, S, E, L, flen, 1, ptr(p_one))
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
