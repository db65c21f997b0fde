"""Oracle mask losses + permutation-invariant wrapper.  TEST INFRASTRUCTURE ONLY.

Reference: models/criterion.py:6-49 (BaseLoss/L1/L2/BCE) and :74-231 (PitWrapper).
"""
from itertools import permutations

import torch
import torch.nn as nn
import torch.nn.functional as F


def _elem(kind, pred, target, weight):
    if kind == "l1":
        return weight * torch.abs(pred - target)
    if kind == "l2":
        return weight * (pred - target) ** 2
    if kind == "bce":
        return F.binary_cross_entropy(pred, target, weight=weight, reduction="none")
    raise Exception("Architecture undefined!")


class MaskLoss(nn.Module):
    """criterion.py:10-25: list -> mean over sources of mean(w*l); tensor -> mean(w*l)."""

    def __init__(self, kind):
        super().__init__()
        self.kind = kind

    def forward(self, preds, targets, weight=None):
        if isinstance(preds, (list, tuple)):
            if weight is None:
                weight = preds[0].new_ones(1)
            errs = [_elem(self.kind, p, t, weight).mean() for p, t in zip(preds, targets)]
            return torch.mean(torch.stack(errs))
        if weight is None:
            weight = preds.new_ones(1)
        return _elem(self.kind, preds, targets, weight).mean()


class PitWrapper(nn.Module):
    """criterion.py:74-231, vectorised.  preds/targets/weights: [B, ..., S].

    loss_mat[b,i,j] = mean(w_i * BCE(pred_j, tgt_i)); the permutation with the
    smallest mean of loss_mat[i, p[i]] wins, first permutation (itertools order)
    kept on ties (strict '>' at criterion.py:133).  perms[b][i] = prediction
    index assigned to target i.
    """

    def __init__(self, kind="bce"):
        super().__init__()
        self.kind = kind

    def loss_matrix(self, preds, targets, weights):
        S = preds.shape[-1]
        p = preds.unsqueeze(-2).expand(*preds.shape[:-1], S, S)       # [..., i, j] = pred_j
        t = targets.unsqueeze(-1).expand(*targets.shape, S)           # [..., i, j] = tgt_i
        w = weights.unsqueeze(-1).expand(*weights.shape, S)
        m = _elem(self.kind, p, t, w)
        return m.mean(dim=tuple(range(1, m.dim() - 2)))               # [B,S,S]

    def forward(self, preds, targets, weights=None):
        if weights is None:
            weights = torch.ones_like(preds)
        mat = self.loss_matrix(preds, targets, weights)
        B, S = mat.shape[0], mat.shape[-1]
        losses, perms = [], []
        for b in range(B):
            best, best_p = None, None
            for p in permutations(range(S)):
                c = mat[b, range(S), p].mean()
                if best is None or best > c:
                    best, best_p = c, p
            losses.append(best)
            perms.append(best_p)
        return torch.stack(losses), perms

    @staticmethod
    def reorder_tensor(tensor, p):
        # criterion.py:180-200: out[b][..., i] = tensor[b][..., p[b][i]]
        out = torch.zeros_like(tensor)
        for b in range(tensor.shape[0]):
            out[b] = tensor[b][..., list(p[b])]
        return out


def build_criterion(arch, use_pit=False):
    # models/__init__.py:121-132: use_pit ignores `arch` and always wraps BCE.
    if arch not in ("bce", "l1", "l2"):
        raise Exception("Architecture undefined!")
    if use_pit:
        return PitWrapper("bce")
    return MaskLoss(arch)
