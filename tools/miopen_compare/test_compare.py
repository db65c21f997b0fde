"""GPU tests of the MIOpen comparison tool (run by hand: `python -m pytest tools/miopen_compare -q`; not part of the
product's test suite): the channels-last glue kernels against torch, and the patched trunks against the HIP trunk."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import assert_close, rel_err  # noqa: E402,F401
import backend as B  # noqa: E402


@pytest.fixture
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda", 0)


def _pkg():
    import avsep_amd
    return avsep_amd


@pytest.mark.parametrize("N,C,H,W", [(4, 128, 8, 8), (3, 64, 9, 7), (2, 512, 4, 4), (5, 256, 6, 10), (2, 16, 5, 5),
                                     (40, 64, 56, 56)])
def test_channels_last_bn_pieces(dev, N, C, H, W):
    """csrc/ops_nhwc.hip against torch: two-stage statistics, normalise + residual (with its own affine) + ReLU, its
    backward with the BatchNorm-backward sums, and the folded BatchNorm gradient, on channels-last tensors."""
    K = _pkg().kernels  # noqa: F841
    g = torch.Generator().manual_seed(C + H)
    cl = lambda t: t.to(dev).contiguous(memory_format=torch.channels_last)   # noqa: E731
    y, r, dz = (torch.randn(N, C, H, W, generator=g) for _ in range(3))
    sc, sh, rs, rh = (torch.randn(C, generator=g) for _ in range(4))
    mean, inv = torch.randn(C, generator=g) * 0.2, torch.rand(C, generator=g) + 0.5
    st = torch.full((2 * C,), float("nan"), dtype=torch.float64, device=dev)          # overwritten, not accumulated
    B.K_nhwc_channel_stats(cl(y), st)
    assert_close(st, torch.cat([y.double().sum((0, 2, 3)), (y.double() ** 2).sum((0, 2, 3))]), 1e-5, "stats")
    v = lambda t: t.view(1, -1, 1, 1)   # noqa: E731
    for use_res, use_rs in ((False, False), (True, False), (True, True)):
        pre = y * v(sc) + v(sh)
        if use_res:
            pre = pre + (r * v(rs) + v(rh) if use_rs else r)
        gpre = dz * (pre > 0).float()
        z = B.K_nhwc_affine_act(cl(y), sc.to(dev), sh.to(dev), cl(r) if use_res else None, 1,
                              rs.to(dev) if use_rs else None, rh.to(dev) if use_rs else None)
        assert_close(z, torch.relu(pre), 1e-6, "affine + residual + relu")
        bst = torch.full((2 * C,), float("nan"), dtype=torch.float64, device=dev)
        d = cl(dz).clone()
        B.K_nhwc_affine_act_bwd_(d, cl(y), sc.to(dev), sh.to(dev), cl(r) if use_res else None, mean.to(dev), inv.to(dev), 1,
                               bst, res_scale=rs.to(dev) if use_rs else None, res_shift=rh.to(dev) if use_rs else None)
        assert_close(d, gpre, 1e-6, "masked gradient")
        xhat = (y - v(mean)) * v(inv)
        assert_close(bst, torch.cat([gpre.double().sum((0, 2, 3)), (gpre.double() * xhat.double()).sum((0, 2, 3))]), 1e-5,
                     "BatchNorm-backward sums")
        bst2 = torch.empty((2 * C,), dtype=torch.float64, device=dev)
        d2 = cl(dz).clone()
        B.K_nhwc_affine_act_bwd_(d2, cl(y), None, None, None, mean.to(dev), inv.to(dev), 0, bst2, stats_only=True)
        assert torch.equal(d2, cl(dz))
        assert_close(bst2, torch.cat([dz.double().sum((0, 2, 3)), (dz.double() * xhat.double()).sum((0, 2, 3))]), 1e-5,
                     "statistics-only pass")
        d3 = cl(dz).clone()                                  # residual join: the second gradient is summed on the fly
        B.K_nhwc_affine_act_bwd_(d3, cl(y), sc.to(dev), sh.to(dev), cl(r) if use_res else None, mean.to(dev), inv.to(dev), 1,
                               bst, res_scale=rs.to(dev) if use_rs else None, res_shift=rh.to(dev) if use_rs else None,
                               dz2=cl(r))
        assert_close(d3, (dz + r) * (pre > 0).float(), 1e-6, "masked sum of two gradients")
    pqr = torch.randn(3, C, generator=g)
    out = B.K_nhwc_bn_bwd_apply_(cl(dz).clone(), cl(y), pqr.to(dev))
    assert_close(out, v(pqr[0]) * dz + v(pqr[1]) * y + v(pqr[2]), 1e-6, "folded BatchNorm gradient")
    # second-stage tails: statistics + finalisation, and backward sums + coefficients, against the two-call forms
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    rm2, rv2 = rm.clone(), rv.clone()
    rows = B.K_nhwc_bn_train_stats(cl(y), gamma.to(dev), beta.to(dev), rm, rv, 0.1, 1e-5)
    st2 = K.zeros_stats(C, y.to(dev))
    K.channel_stats(y.to(dev), st2)
    ref_rows = K.bn_finalize(st2, N * H * W, gamma.to(dev), beta.to(dev), rm2, rv2, 0.1, 1e-5, True, y.to(dev))
    assert_close(rows, ref_rows, 1e-5, "statistics + finalisation")
    assert_close(rm, rm2, 1e-6, "running mean")
    assert_close(rv, rv2, 1e-6, "running var")
    d = cl(dz).clone()
    dgamma, dbeta, pq = B.K_nhwc_affine_act_bwd_(d, cl(y), rows[0], rows[1], None, rows[2], rows[3], 1, None,
                                               gamma=gamma.to(dev), coeffs=True)
    bst = K.zeros_stats(C, y.to(dev))
    d_ref = K.affine_act_bwd_(dz.to(dev).clone(), y.to(dev), ref_rows[0], ref_rows[1], None, None, ref_rows[2], ref_rows[3],
                              1, bst)
    rg, rb, rpq = K.bn_bwd_coeffs(bst, N * H * W, gamma.to(dev), ref_rows[2], ref_rows[3])
    assert_close(d, d_ref, 1e-6, "masked gradient (tail form)")
    assert_close(dgamma, rg, 2e-5, "dgamma")
    assert_close(dbeta, rb, 2e-5, "dbeta")
    assert_close(pq, rpq, 2e-5, "pqr")


@pytest.mark.parametrize("N,C,H,W", [(2, 64, 14, 18), (3, 16, 9, 11), (1, 64, 112, 112)])
def test_channels_last_stem_tail(dev, N, C, H, W):
    """maxpool(relu(bn(y))) fused forward (activated map never materialised) and its backward fused with the ReLU mask
    and the train-mode BatchNorm backward, against torch autograd on the CPU."""
    K = _pkg().kernels  # noqa: F841
    g = torch.Generator().manual_seed(H)
    y = torch.randn(N, C, H, W, generator=g) * 1.5 + 0.3
    bn = torch.nn.BatchNorm2d(C)
    bn.weight.data, bn.bias.data = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    yr = y.clone().requires_grad_(True)
    p_ref = F.max_pool2d(torch.relu(bn(yr)), 3, 2, 1)
    cot = torch.randn(p_ref.shape, generator=g)
    (p_ref * cot).sum().backward()
    cl = lambda t: t.to(dev).contiguous(memory_format=torch.channels_last)   # noqa: E731
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    rows = B.K_nhwc_bn_train_stats(cl(y), bn.weight.data.to(dev), bn.bias.data.to(dev), rm, rv, 0.1, 1e-5)
    p, taps = B.K_nhwc_maxpool_bn_relu(cl(y), rows[0], rows[1])
    assert_close(p, p_ref, 2e-6, "pooled activations")
    dgamma, dbeta, dy = B.K_nhwc_maxpool_bn_relu_bwd(cl(cot), taps, cl(y), rows, bn.weight.data.to(dev))
    assert_close(dy, yr.grad, 2e-5, "gradient wrt the conv output (through batch statistics)")
    assert_close(dgamma, bn.weight.grad, 2e-5, "dgamma")
    assert_close(dbeta, bn.bias.grad, 2e-5, "dbeta")


@pytest.mark.parametrize("backend", ["hybrid", "torch"])
def test_patched_trunk_matches_hip_trunk(dev, backend):
    P = _pkg()
    torch.manual_seed(3)
    net = P.models.ResnetDilated(None, fc_dim=16, pool_type="maxpool").to(dev).train()
    x = torch.randn(2, 3, 2, 96, 96, device=dev)
    import copy
    other = B.install(copy.deepcopy(net), backend)
    y0 = net.forward_multiframe(x, pool=False)
    y1 = other.forward_multiframe(x, pool=False)
    assert_close(y1, y0, 2e-4, "features")
    cot = torch.randn_like(y0)
    (y0 * cot).sum().backward()
    (y1 * cot).sum().backward()
    assert_close(other.fc.weight.grad, net.fc.weight.grad, 1e-4, "fc.weight grad")
