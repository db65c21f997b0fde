"""Mask losses + permutation-invariant wrapper on the HIP kernels
(reference: models/criterion.py:6-49 BaseLoss/L1/L2/BCE, :74-231 PitWrapper).

One fused kernel (avsep_mask_loss_fwd) applies the output activation, the weighted
element loss and the per-sample reductions; it always produces the full [B,S,S] matrix
sums[b,i,j] = sum w_i * l(pred_j, target_i), of which the plain losses use the diagonal and
PIT every entry.  ``mask_loss`` is the autograd entry; the classes keep the reference's call
signatures.
"""
from itertools import permutations

import torch
import torch.nn as nn

from .. import lib
from ..lib import ACT_NONE, LOSS_BY_NAME, call, ptr


class _MaskLossFn(torch.autograd.Function):
    """x [B,S,FT] (logits or probabilities), gt [S,B,FT], weight [B,FT] or [S,B,FT] or None.
    Returns (pred [B,S,FT], sums [B,S,S] fp64).  d(sums)/dx is applied through `coef`."""

    @staticmethod
    def forward(ctx, x, gt, weight, act, loss):
        lib.require_gpu(x)
        B, S, FT = x.shape
        pred = torch.empty_like(x)
        sums = torch.zeros((B, S, S), dtype=torch.float64, device=x.device)
        wts = 0
        if weight is not None and weight.dim() == 3:
            wts = B * FT
        call("avsep_mask_loss_fwd", ptr(x), ptr(gt), ptr(weight), wts, B, S, FT, act, loss, ptr(pred), ptr(sums))
        ctx.save_for_backward(x, gt, weight)
        ctx.cfg = (act, loss, wts)
        ctx.mark_non_differentiable(pred)
        return pred, sums

    @staticmethod
    def backward(ctx, _dpred, dsums):
        x, gt, weight = ctx.saved_tensors
        act, loss, wts = ctx.cfg
        B, S, FT = x.shape
        coef = dsums.to(torch.float32).contiguous()
        dx = torch.empty_like(x)
        call("avsep_mask_loss_bwd", ptr(x), ptr(gt), ptr(weight), wts, ptr(coef), B, S, FT, act, loss, ptr(dx))
        return dx, None, None, None, None


def mask_loss(x, gt, weight, act, loss):
    """x [B,S,F,T]-like; gt [S,B,...]; weight [B,...] (shared) or [S,B,...] (per target)."""
    B, S = x.shape[:2]
    FT = x[0, 0].numel()
    w = None
    if weight is not None:
        w = weight.reshape(B, FT).contiguous() if weight.numel() == B * FT else weight.reshape(S, B, FT).contiguous()
    pred, sums = _MaskLossFn.apply(x.reshape(B, S, FT).contiguous(), gt.reshape(S, B, FT).contiguous(), w, act,
                                   LOSS_BY_NAME[loss] if isinstance(loss, str) else loss)
    return pred.view_as(x), sums, FT


class BaseLoss(nn.Module):
    kind = "bce"

    def forward(self, preds, targets, weight=None):
        # criterion.py:10-25: list -> mean over sources of mean(w*l); tensor -> mean(w*l)
        if isinstance(preds, (list, tuple)):
            x = torch.cat([p.reshape(p.shape[0], 1, -1) for p in preds], 1)
            gt = torch.stack([t.reshape(t.shape[0], -1) for t in targets], 0)
        elif isinstance(preds, torch.Tensor):
            x = preds.reshape(preds.shape[0], 1, -1)
            gt = targets.reshape(1, targets.shape[0], -1)
        else:
            raise TypeError(type(preds))
        B, S, FT = x.shape
        w = None
        if weight is not None and weight.numel() > 1:
            w = weight.expand_as(preds[0] if isinstance(preds, (list, tuple)) else preds).reshape(B, FT)
        scale = 1.0 if (weight is None or weight.numel() > 1) else weight.reshape(())
        _, sums, _ = mask_loss(x, gt, w, ACT_NONE, self.kind)
        err = torch.diagonal(sums, dim1=1, dim2=2).sum() / (B * S * FT)
        return (err * scale).to(torch.float32)


class L1Loss(BaseLoss):
    kind = "l1"


class L2Loss(BaseLoss):
    kind = "l2"


class BCELoss(BaseLoss):
    kind = "bce"


def best_permutations(loss_mat):
    """criterion.py:111-136: per sample, the permutation p minimising mean_i loss_mat[i, p[i]];
    the first permutation in itertools order wins ties (strict '>').  loss_mat: host [B,S,S]."""
    B, S = loss_mat.shape[:2]
    out = []
    for b in range(B):
        best, best_p = None, None
        for p in permutations(range(S)):
            c = sum(loss_mat[b, i, p[i]] for i in range(S)) / S
            if best is None or best > c:
                best, best_p = c, p
        out.append(best_p)
    return out


class DevicePerms:
    """The winning permutations as a device tensor `idx` [B,S] (idx[b,i] = prediction assigned to target i).  Behaves
    like the reference's list of tuples when a caller looks inside (that, and only that, synchronises)."""

    def __init__(self, idx):
        self.idx = idx
        self._host = None

    def tolist(self):
        if self._host is None:
            self._host = [tuple(r) for r in self.idx.tolist()]
        return self._host

    def __len__(self):
        return self.idx.shape[0]

    def __iter__(self):
        return iter(self.tolist())

    def __getitem__(self, i):
        return self.tolist()[i]

    def __eq__(self, other):
        return self.tolist() == [tuple(q) for q in other]

    def __repr__(self):
        return repr(self.tolist())


_PERM_TABLES = {}


def best_permutations_device(mat):
    """best_permutations without leaving the device: all S! candidate costs in one gather, argmin keeps the FIRST
    minimum = the first permutation in itertools order on ties (criterion.py:133's strict '>')."""
    B, S = mat.shape[:2]
    key = (S, mat.device)
    if key not in _PERM_TABLES:
        _PERM_TABLES[key] = torch.tensor(list(permutations(range(S))), device=mat.device)      # [S!, S]
    table = _PERM_TABLES[key]
    P = table.shape[0]
    cost = torch.gather(mat.detach().unsqueeze(1).expand(B, P, S, S), 3, table[None, :, :, None].expand(B, P, S, 1))
    best = (cost.squeeze(-1).sum(-1) / S).argmin(1)
    return DevicePerms(table[best])


def pit_select(mat):
    """(per-sample PIT loss [B] fp32-castable, DevicePerms) from the mean loss matrix [B,S,S]; no host sync."""
    perms = best_permutations_device(mat)
    loss = torch.gather(mat, 2, perms.idx[:, :, None]).squeeze(-1).mean(-1)
    return loss, perms


class PitWrapper(nn.Module):
    """Permutation-invariant wrapper; preds/targets/weights are [B, ..., S] like the reference.
    ``base_loss`` is kept for signature compatibility (the reference always passes BCE,
    models/__init__.py:130-131)."""

    def __init__(self, base_loss=None, kind="bce", act=ACT_NONE):
        super().__init__()
        self.base_loss, self.kind, self.act = base_loss, kind, act

    def loss_matrix(self, preds, targets, weights):
        """Returns (pred_activated [B,...,S], mean loss matrix [B,S,S])."""
        S = preds.shape[-1]
        x = preds.movedim(-1, 1).contiguous()                      # [B,S,...]
        gt = targets.movedim(-1, 0).contiguous()                   # [S,B,...]
        w = weights.movedim(-1, 0).contiguous() if weights is not None else None
        pred, sums, FT = mask_loss(x, gt, w, self.act, self.kind)
        return pred.movedim(1, -1), sums / FT

    def forward(self, preds, targets, weights=None):
        _, mat = self.loss_matrix(preds, targets, weights)
        loss, perms = pit_select(mat)                              # permutation chosen on the device, no D2H copy
        return loss.to(torch.float32), perms

    @staticmethod
    def reorder_tensor(tensor, p):
        # criterion.py:180-200: out[b][..., i] = tensor[b][..., p[b][i]]
        idx = p.idx if isinstance(p, DevicePerms) else torch.tensor([list(q) for q in p], device=tensor.device)
        shape = [tensor.shape[0]] + [1] * (tensor.dim() - 2) + [tensor.shape[-1]]
        return torch.gather(tensor, -1, idx.view(shape).expand_as(tensor))
