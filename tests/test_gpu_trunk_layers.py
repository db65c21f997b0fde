"""-m gpu: per-layer parity of the HIP visual trunk (models/vision_hip.py) against the oracle in float64.

Reference: models/vision_net.py:84-147 (ResnetDilated._nostride_dilate, features, fc) + the torchvision BasicBlock
(restated in oracle/nets.py).  Every stage of the trunk — stem (conv7x7/s2 + BN + ReLU + max-pool), each BasicBlock
geometry the three architectures produce (plain, strided with 1x1 downsample, dilated 1/2/4, dilated with 1x1/s1
downsample) and the fc conv — is run ON ITS OWN: the oracle's float64 input z and output cotangent are injected at
that stage, so no ReLU decision of an upstream layer can differ.  Inside a stage the BatchNorm biases of the oracle
are nudged (identically on both sides) until every ReLU pre-activation is at least MARGIN away from zero and every
max-pool window has a unique winner, i.e. the float32 path provably takes the same branch everywhere.  Then

    output, dL/dz, every parameter gradient (dw, dgamma, dbeta), running statistics:  max|d| / max|ref| <= 1e-4.
"""
import copy

import pytest
import torch
import torch.nn.functional as F

from conftest import assert_close

pytestmark = pytest.mark.gpu
MARGIN = 4e-5
POOL_MARGIN = 1e-5      # fp32 error of the 147-term stem conv is ~1e-6 of its output scale
TOL = 1e-4



def _grads(VH, mod):
    """ParamGrads without a gradient sink: a plain {parameter: gradient} mapping (`.d`), read through __getitem__."""
    class G(VH.ParamGrads):
        def __contains__(self, p):
            return p in self.d

        def __getitem__(self, p):
            return self.d[p]
    return G(mod)

def _pkg():
    import avsep_amd
    return avsep_amd


def _nudge(pre, bias, margin=MARGIN):
    """Shift bias[c] (a BatchNorm beta: the batch statistics do not see it) until no element of the pre-activation
    `pre` [B,C,H,W] (which already contains bias) is within `margin` of zero.  Returns the per-channel shift."""
    C = pre.shape[1]
    v = pre.detach().transpose(0, 1).reshape(C, -1)
    shift = torch.zeros(C, dtype=pre.dtype)
    for k in range(1, 200):
        bad = ((v + shift[:, None]).abs().min(1).values < margin)
        if not bad.any():
            break
        shift[bad] = (7.3 * margin) * ((k + 1) // 2) * (1 if k % 2 else -1)
    assert not ((v + shift[:, None]).abs().min(1).values < margin).any()
    with torch.no_grad():
        bias += shift.to(bias.dtype)
    return shift


def _wide(mod, gen):
    import oracle.nets as ON
    ON.wide_init(mod, gen)
    for m in mod.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1, generator=gen)
            m.running_var.uniform_(0.5, 1.5, generator=gen)


def _pair(arch, dilate, gen):
    """(oracle VisualNet float32 on CPU with wide weights, product net with the same state)."""
    P = _pkg()
    import oracle.nets as ON
    onet = ON.VisualNet(fc_dim=24, pool_type="maxpool", dilate_scale=dilate)
    _wide(onet, gen)
    net = (P.models.ResnetDilated(None, fc_dim=24, dilate_scale=dilate) if arch == "dilated"
           else P.models.ResnetFC(None, fc_dim=24))
    assert list(net.state_dict().keys()) == list(onet.state_dict().keys())
    return onet, net


# (arch, dilate_scale, features index, block index, batch, input size): every BasicBlock geometry of resnet18fc,
# resnet18dilated (scale 16 = the config of record, and scale 8) at its true channel counts and (for the config of
# record) its true spatial size
BLOCKS = [
    ("dilated", 16, 4, 0, 1, 56), ("dilated", 16, 4, 1, 2, 28),      # layer1: 64 -> 64, plain
    ("dilated", 16, 5, 0, 1, 56), ("dilated", 16, 5, 1, 2, 28),      # layer2: 3x3/s2 + 1x1/s2 downsample; plain 128
    ("dilated", 16, 6, 0, 2, 28), ("dilated", 16, 6, 1, 3, 14),      # layer3: 128 -> 256 strided; plain 256 (14x14)
    ("dilated", 16, 7, 0, 3, 14), ("dilated", 16, 7, 1, 2, 14),      # layer4: d1+d2 with 1x1/s1 downsample; d2+d2
    ("dilated", 8, 6, 0, 1, 28), ("dilated", 8, 7, 0, 1, 28), ("dilated", 8, 7, 1, 1, 28),   # scale 8: d1/d2, d2/d4, d4/d4
    ("fc", None, 7, 0, 2, 14), ("fc", None, 7, 1, 4, 7),              # resnet18fc: strided layer4, 7x7 maps
]


@pytest.mark.parametrize("arch,dilate,li,bi,B,H", BLOCKS)
def test_basic_block_layer_parity(dev, arch, dilate, li, bi, B, H):
    P = _pkg()
    from avsep_amd.models import vision_hip as VH
    gen = torch.Generator().manual_seed(100 * li + 10 * bi + (dilate or 0))
    onet, net = _pair(arch, dilate, gen)
    oblk = onet.features[li][bi]
    cin = oblk.conv1.in_channels
    z = F.relu(torch.randn(B, cin, H, H, generator=gen))
    o64 = copy.deepcopy(oblk).double().train()
    stats0 = {k: v.clone() for k, v in o64.state_dict().items() if "running_" in k or "num_batches" in k}
    # branch margins: bn1 -> ReLU, then (bn2 + identity) -> ReLU
    with torch.no_grad():
        pre1 = o64.bn1(o64.conv1(z.double()))
    sh1 = _nudge(pre1, o64.bn1.bias)
    with torch.no_grad():
        a = F.relu(pre1 + sh1.view(1, -1, 1, 1))
        idt = z.double() if o64.downsample is None else o64.downsample(z.double())
        pre2 = o64.bn2(o64.conv2(a)) + idt
    _nudge(pre2, o64.bn2.bias)
    o64.load_state_dict({**o64.state_dict(), **stats0})      # the probes above advanced the running statistics
    sd = {k: (v.float() if v.dtype.is_floating_point else v.clone()) for k, v in o64.state_dict().items()}
    blk = net.features[li][bi]
    blk.load_state_dict(sd)
    blk = blk.to(dev).train()
    zin = z.double().requires_grad_(True)
    out64 = o64(zin)
    cot = torch.randn(out64.shape, generator=gen)
    (out64 * cot.double()).sum().backward()

    R, out = VH.block_forward(blk, z.to(dev), True)
    grads = _grads(VH, blk)
    dz, dz2 = VH.block_backward(R, cot.to(dev).clone(), grads)
    dz = dz + dz2          # (conv branch, identity / downsample branch): the consumer's first pass sums them
    tag = f"{arch}{dilate} features[{li}][{bi}]"
    assert_close(out, out64, TOL, tag + ": output")
    assert_close(dz, zin.grad, TOL, tag + ": dz")
    og = dict(o64.named_parameters())
    for k, p in blk.named_parameters():
        assert p in grads, k
        assert_close(grads[p], og[k].grad, TOL, f"{tag}: grad {k}")
    ob = dict(o64.named_buffers())
    for k, b in blk.named_buffers():
        if b.dtype.is_floating_point:
            assert_close(b, ob[k], TOL, f"{tag}: buffer {k}")
        else:
            assert int(b) == int(ob[k]) == 1, k


# (dilate_scale, features index, batch, input size): both blocks of a layer in a chain, the way trunk_backward runs them in
# fp32, with relu'(bn1(y1)) and its BatchNorm sums in the epilogue of conv2's F(4x4) data gradient (avsep_conv2d_dgrad_act).
# The descriptors are PLANNED for the bench's batch, which takes them to the kernels that carry the epilogue (asserted), and
# run at a batch small enough for the branch margins; the single-block test above runs the same call unplanned, in the
# two-launch form inside the library.
LAYERS = [(16, 4, 1, 56), (16, 5, 1, 56), (16, 6, 2, 28), (16, 7, 3, 14)]
CHAIN_PLAN_SCALE = 96      # the launch decisions of a batch this many times larger (kernels.plan_batch_scale): the bench's


@pytest.mark.parametrize("dilate,li,B,H", LAYERS)
def test_layer_chain_with_fused_activation_gradient(dev, dilate, li, B, H):
    from avsep_amd.models import vision_hip as VH
    K = _pkg().kernels
    gen = torch.Generator().manual_seed(1000 + li)
    onet, net = _pair("dilated", dilate, gen)
    olayer = onet.features[li]
    cin = olayer[0].conv1.in_channels
    z = F.relu(torch.randn(B, cin, H, H, generator=gen))
    o64 = copy.deepcopy(olayer).double().train()
    stats0 = {k: v.clone() for k, v in o64.state_dict().items() if "running_" in k or "num_batches" in k}
    zz = z.double()
    for ob in o64:                                              # branch margins, block after block
        with torch.no_grad():
            pre1 = ob.bn1(ob.conv1(zz))
        sh1 = _nudge(pre1, ob.bn1.bias)
        with torch.no_grad():
            a = F.relu(pre1 + sh1.view(1, -1, 1, 1))
            idt = zz if ob.downsample is None else ob.downsample(zz)
            pre2 = ob.bn2(ob.conv2(a)) + idt
        sh2 = _nudge(pre2, ob.bn2.bias)
        zz = F.relu(pre2 + sh2.view(1, -1, 1, 1))
    o64.load_state_dict({**o64.state_dict(), **stats0})
    sd = {k: (v.float() if v.dtype.is_floating_point else v.clone()) for k, v in o64.state_dict().items()}
    layer = net.features[li]
    layer.load_state_dict(sd)
    layer = layer.to(dev).train()
    zin = z.double().requires_grad_(True)
    out64 = o64(zin)
    cot = torch.randn(out64.shape, generator=gen)
    (out64 * cot.double()).sum().backward()

    K.plan_batch_scale = CHAIN_PLAN_SCALE
    try:
        R0, z1 = VH.block_forward(layer[0], z.to(dev), True)
        R1, out = VH.block_forward(layer[1], z1, True)
        assert R0["cv2"].dgrad_act_fused() and R1["cv2"].dgrad_act_fused()
        grads = _grads(VH, layer)
        g, g2 = VH.block_backward(R1, cot.to(dev).clone(), grads)
        dz, dz2 = VH.block_backward(R0, g, grads, g2)
        dz = dz + dz2
    finally:
        K.plan_batch_scale = 1
    tag = f"dilated{dilate} features[{li}]"
    assert_close(out, out64, TOL, tag + ": output")
    assert_close(dz, zin.grad, TOL, tag + ": dz")
    og = dict(o64.named_parameters())
    for k, p in layer.named_parameters():
        assert p in grads, k
        assert_close(grads[p], og[k].grad, TOL, f"{tag}: grad {k}")


def _pool_margin(act):
    """smallest gap between the winner and the runner-up over all 3x3/s2/p1 windows whose winner is positive."""
    B, C, H, W = act.shape
    win = F.unfold(F.pad(act, (1, 1, 1, 1), value=float("-inf")).reshape(B * C, 1, H + 2, W + 2), 3, stride=2)
    top = win.topk(2, dim=1).values
    gap = (top[:, 0] - top[:, 1])[top[:, 0] > 0]
    return gap.min().item() if gap.numel() else 1.0


@pytest.mark.parametrize("B,H,seed0", [(2, 64, 507), (1, 224, 537)])     # seeds known to have clear margins
def test_stem_layer_parity(dev, B, H, seed0):
    """conv7x7/s2 + BN + ReLU + MaxPool(3,2,1) with its backward (dw0, dgamma, dbeta), vision_net.py:111-117 children 0-3."""
    from avsep_amd.models import vision_hip as VH
    for seed in range(40):
        gen = torch.Generator().manual_seed(seed0 + seed)
        onet, net = _pair("dilated", 16, gen)
        x = torch.randn(B, 3, H, H, generator=gen)
        o64 = copy.deepcopy(onet.features[:4]).double().train()
        rm, rv = o64[1].running_mean.clone(), o64[1].running_var.clone()
        with torch.no_grad():
            pre = o64[1](o64[0](x.double()))
        sh = _nudge(pre, o64[1].bias)
        o64[1].running_mean.copy_(rm); o64[1].running_var.copy_(rv); o64[1].num_batches_tracked.zero_()
        if _pool_margin(F.relu(pre + sh.view(1, -1, 1, 1))) > POOL_MARGIN:
            break
    else:
        pytest.fail("no seed with a unique max-pool winner everywhere")
    f = net.features
    f[0].load_state_dict({k: v.float() for k, v in o64[0].state_dict().items()})
    f[1].load_state_dict({k: (v.float() if v.dtype.is_floating_point else v) for k, v in o64[1].state_dict().items()})
    f = f.to(dev).train()
    out64 = o64(x.double())
    cot = torch.randn(out64.shape, generator=gen)
    (out64 * cot.double()).sum().backward()
    S, z = VH.stem_forward(f, x.to(dev), True)
    grads = _grads(VH, f)
    VH.stem_backward(f, S, cot.to(dev).clone(), grads)
    assert_close(z, out64, TOL, "stem output")
    assert_close(grads[f[0].weight], o64[0].weight.grad, TOL, "stem dw")
    assert_close(grads[f[1].weight], o64[1].weight.grad, TOL, "stem dgamma")
    assert_close(grads[f[1].bias], o64[1].bias.grad, TOL, "stem dbeta")
    assert_close(f[1].running_mean, o64[1].running_mean, TOL, "stem running_mean")
    assert_close(f[1].running_var, o64[1].running_var, TOL, "stem running_var")


@pytest.mark.parametrize("B,H,fc_dim", [(3, 14, 256), (2, 7, 24)])
def test_fc_conv_layer_parity(dev, B, H, fc_dim):
    """the 3x3 fc conv with bias on the last feature map (vision_net.py:82,121): output, dz, dw, dbias."""
    from avsep_amd.models import vision_hip as VH
    gen = torch.Generator().manual_seed(9)
    fc = torch.nn.Conv2d(512, fc_dim, 3, padding=1)
    _wide(fc, gen)
    z = F.relu(torch.randn(B, 512, H, H, generator=gen))
    o64 = copy.deepcopy(fc).double()
    zin = z.double().requires_grad_(True)
    out64 = o64(zin)
    cot = torch.randn(out64.shape, generator=gen)
    (out64 * cot.double()).sum().backward()
    fc = fc.to(dev)
    cvf, out = VH.fc_forward(fc, z.to(dev))
    grads = _grads(VH, fc)
    dz = VH.fc_backward(fc, cvf, cot.to(dev), grads)
    assert_close(out, out64, TOL, "fc output")
    assert_close(dz, zin.grad, TOL, "fc dz")
    assert_close(grads[fc.weight], o64.weight.grad, TOL, "fc dw")
    assert_close(grads[fc.bias], o64.bias.grad, TOL, "fc dbias")
