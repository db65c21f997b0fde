"""ResNet-18 (dilated) frame trunk + fc conv as ONE autograd node over libavsep_gfx950.so.

Reference: models/vision_net.py:62-147 (ResnetDilated / ResnetFC: torchvision resnet18 children[:-2],
``_nostride_dilate``, the fc conv) — the per-frame visual conv stack of the AV step.

Launch sequence per BasicBlock (z = block input, a materialised post-ReLU tensor):

    y1 = conv3x3(z)                         BN1 batch statistics in the conv epilogue
    y2 = conv3x3(relu(bn1(y1)))             BN1 + ReLU folded into conv2's operand gather; BN2 statistics in the epilogue
    yd = conv1x1(z)                         only with a downsample branch; BNd statistics in the epilogue
    z' = relu(bn2(y2) + (bnd(yd) | z))      one elementwise pass (avsep_affine_act)

The stem is conv7x7/s2 with BN statistics in the epilogue, then max-pool 3x3/s2 reading relu(bn(y0)) on the fly.
Backward mirrors it with the BatchNorm backward folded as dy = p*dz + q*y + r (see audio_net.py).  The torch
modules in ``net.features`` / ``net.fc`` stay the parameter and running-statistics holders, so checkpoints and the
optimizer see the same tensors whichever backend runs.
"""
import torch

from .. import kernels as K
from .. import lib
from ..lib import ACT_NONE, ACT_RELU
from .audio_net import ParamGrads, _bn_back, _bn_run


def _geom(conv):
    k, s, p, d = conv.kernel_size, conv.stride, conv.padding, conv.dilation
    if k[0] != k[1] or s[0] != s[1] or p[0] != p[1] or d[0] != d[1]:
        raise lib.AvsepError("the HIP visual trunk needs square conv geometry")
    return k[0], s[0], p[0], d[0]


def _w(conv):
    """OIHW-dense weight (the flat optimizer may hold it channels-last for the MIOpen backend)."""
    return conv.weight.detach().contiguous()


def _conv(src, conv, aff=None):
    k, s, p, d = _geom(conv)
    if aff is None:
        return K.Conv(src, conv.out_channels, k, s, p, d)
    return K.Conv(src, conv.out_channels, k, s, p, d, sc0=aff[0], sh0=aff[1], act0=ACT_RELU)


def _conv_bn(src, conv, bn, training, aff=None):
    cv = _conv(src, conv, aff)
    st = K.zeros_stats(conv.out_channels, src) if training else None
    y = cv.fwd(cv.pack(_w(conv), 0), None, st)
    return cv, y, _bn_run(bn, st, K.per_channel(y), training, src)


def blocks_of(features):
    return [b for i in (4, 5, 6, 7) for b in features[i]]


def param_list(net):
    return list(net.features.parameters()) + list(net.fc.parameters())


_S2D_INDEX = {}


def _stem_s2d_weight(w):
    """w [Co,C,7,7] -> w' [Co,16,4,4] with w'[co][(dy*2+dx)*C + c][a][b] = w[co][c][2a+dy-1][2b+dx-1] (zero outside the 7x7
    window, zero pad channels): the stride-1 form of the stem over kernels.space_to_depth2(x)."""
    Co, Cc = w.shape[:2]
    key = (Cc, w.device)
    if key not in _S2D_INDEX:
        idx = torch.zeros((16, 4, 4), dtype=torch.long)
        ok = torch.zeros((16, 4, 4), dtype=torch.bool)
        for q in range(4 * Cc):
            c, dy, dx = q % Cc, (q // Cc) >> 1, (q // Cc) & 1
            for a in range(4):
                for b in range(4):
                    kh, kw = 2 * a + dy - 1, 2 * b + dx - 1
                    if 0 <= kh <= 6 and 0 <= kw <= 6:
                        idx[q, a, b], ok[q, a, b] = (c * 7 + kh) * 7 + kw, True
        _S2D_INDEX[key] = (idx.to(w.device).reshape(-1), ok.to(w.device).reshape(1, 16, 4, 4))
    idx, ok = _S2D_INDEX[key]
    return (w.reshape(Co, -1)[:, idx].reshape(Co, 16, 4, 4) * ok).contiguous()


def _stem_s2d_weight_grad(dw2, Cc):
    """The adjoint of _stem_s2d_weight: dw' [Co,16,4,4] (gradient of the stride-1 form) -> dw [Co,C,7,7].  Every 7x7 tap sits at
    exactly one (phase channel, a, b) position, so this is a gather."""
    Co = dw2.shape[0]
    key = ("inv", Cc, dw2.device)
    if key not in _S2D_INDEX:
        inv = torch.zeros((Cc, 7, 7), dtype=torch.long)
        for q in range(4 * Cc):
            c, dy, dx = q % Cc, (q // Cc) >> 1, (q // Cc) & 1
            for a in range(4):
                for b in range(4):
                    kh, kw = 2 * a + dy - 1, 2 * b + dx - 1
                    if 0 <= kh <= 6 and 0 <= kw <= 6:
                        inv[c, kh, kw] = (q * 4 + a) * 4 + b
        _S2D_INDEX[key] = inv.to(dw2.device).reshape(-1)
    return dw2.reshape(Co, -1)[:, _S2D_INDEX[key]].reshape(Co, Cc, 7, 7)


def _stem_is_s2d(conv, x):
    return (conv.kernel_size == (7, 7) and conv.stride == (2, 2) and conv.padding == (3, 3) and conv.dilation == (1, 1)
            and conv.in_channels <= 4 and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0)


def stem_forward(f, x, training):
    """conv7x7/s2 (+BN statistics) -> max-pool 3x3/s2 over relu(bn(y0)) read on the fly.  The forward conv runs in its
    stride-1 form over the space-to-depth frames (16-channel K-tiles on the halo-patch kernels); the weight gradient
    keeps the 7x7/s2 descriptor over the frames themselves."""
    S = {"x": x}
    if _stem_is_s2d(f[0], x):
        S["cv0"] = _conv(x, f[0])                       # descriptor of the original conv: the weight gradient uses it
        cs = K.Conv(K.space_to_depth2(x), f[0].out_channels, 4, 1, 0)
        S["cs"] = cs
        st = K.zeros_stats(f[0].out_channels, x) if training else None
        S["y0"] = cs.fwd(cs.pack(_stem_s2d_weight(_w(f[0])), 0), None, st)
        S["bn0"] = _bn_run(f[1], st, K.per_channel(S["y0"]), training, x)
    else:
        S["cv0"], S["y0"], S["bn0"] = _conv_bn(x, f[0], f[1], training)
    z, S["idx"] = K.maxpool3x3s2(S["y0"], S["bn0"][0], S["bn0"][1], ACT_RELU)
    return S, z


def block_forward(blk, z, training):
    """One BasicBlock on a materialised post-ReLU input z -> (saved record, z')."""
    R = {"mod": blk, "z": z}
    R["cv1"], R["y1"], R["bn1"] = _conv_bn(z, blk.conv1, blk.bn1, training)
    if K.get_precision() == "f32":
        # fp32: relu(bn1(y1)) is written out once (one elementwise pass) and conv2 reads it as a plain tensor.  Folding the
        # affine + ReLU into conv2's staging costs vector instructions in its forward AND its weight gradient, and the f32
        # MFMA does not overlap them (SQ: 8-11 VALU per MFMA against 4.5-5.5 for the plain-input instantiations of the
        # Winograd kernels): 109.9 -> 109.0 ms per step.  With bf16 operands the fold is free (47.2 vs 47.7 ms): kept there.
        # (Round 5, with the F(4x4) kernels: folding costs them +9-11 % per call and saves this pass — same-box A/B of the
        # whole step 78.1 / 79.4 ms materialised against 78.6 / 78.7 folded: a wash, the materialised form stays.)
        a1 = K.affine_act(R["y1"], R["bn1"][0], R["bn1"][1], None, ACT_RELU)
        R["cv2"], R["y2"], R["bn2"] = _conv_bn(a1, blk.conv2, blk.bn2, training)
    else:
        R["cv2"], R["y2"], R["bn2"] = _conv_bn(R["y1"], blk.conv2, blk.bn2, training, aff=R["bn1"])
    if blk.downsample is not None:
        R["cvd"], R["yd"], R["bnd"] = _conv_bn(z, blk.downsample[0], blk.downsample[1], training)
        out = K.affine_act(R["y2"], R["bn2"][0], R["bn2"][1], R["yd"], ACT_RELU, R["bnd"][0], R["bnd"][1])
    else:
        out = K.affine_act(R["y2"], R["bn2"][0], R["bn2"][1], R["z"], ACT_RELU)
    return R, out


def fc_forward(fc, z):
    cvf = _conv(z, fc)
    bias = fc.bias.detach() if fc.bias is not None else None
    return cvf, cvf.fwd(cvf.pack(_w(fc), 0), bias, None, out_b16=False)     # consumed by torch ops (temporal mean, activate)


def trunk_forward(net, x, training):
    S, z = stem_forward(net.features, x, training)
    S["blocks"] = []
    for blk in blocks_of(net.features):
        R, z = block_forward(blk, z, training)
        S["blocks"].append(R)
    S["cvf"], out = fc_forward(net.fc, z)
    return S, out


def _relu_bn_back(grads, g, y, bnrow, bn_mod, res=None, rs=None, rh=None, g2=None):
    """g <- relu'(bn(y) [+ residual]) * (g [+ g2]) (in place when g already has y's storage format); returns (g, the folded
    BatchNorm-backward coefficients of bn(y))."""
    bst = K.zeros_stats(K.channels(y), y)
    g = K.affine_act_bwd_(g, y, bnrow[0], bnrow[1], res, None, bnrow[2], bnrow[3], ACT_RELU, bst, res_scale=rs, res_shift=rh,
                          dz2=g2)
    return g, _bn_back(grads, bn_mod, bnrow, bst, K.per_channel(y))


def fc_backward(fc, cvf, dout, grads):
    grads.wgrad(cvf, fc.weight, dout, fc.bias)
    return cvf.dgrad(cvf.pack(_w(fc), 1), dout)


def block_backward(R, g, grads, g2=None):
    """dL/dz' = g (+ g2), consumed in place -> dL/dz as ONE tensor (downsample blocks) or as the pair (conv branch,
    identity branch): the pair is summed by the consumer's first elementwise pass instead of by a launch of its own.
    Parameter gradients are accumulated into `grads`.

    fp32: the ReLU' / BatchNorm-sum pass over conv2's data gradient runs in the EPILOGUE of that kernel
    (avsep_conv2d_dgrad_act): dL/da1 is never written unmasked.  The block TAIL relu'(bn2(y2) + residual) stays a pass of its
    own: in the next block's conv1 epilogue its three extra operand reads cost more than the pass (one workgroup per CU,
    nothing overlaps the epilogue's memory latency): +4-8 % per call on the 28x28 / 14x14 maps against -1 % for this one
    (`tools/conv_bench.py dgrad --act-epilogue bn1|tail`)."""
    blk = R["mod"]
    ds = blk.downsample is not None
    bnd = R.get("bnd")
    g, pqr2 = _relu_bn_back(grads, g, R["y2"], R["bn2"], blk.bn2, res=R["yd"] if ds else R["z"],
                            rs=bnd[0] if ds else None, rh=bnd[1] if ds else None, g2=g2)   # g = dL/d(pre-ReLU sum)
    dy2 = K.bn_bwd_apply_(g, R["y2"], pqr2, fresh=True)
    grads.wgrad(R["cv2"], blk.conv2.weight, dy2)
    if K.get_precision() == "f32" and not K.is_b16(R["y1"]):
        bn1 = R["bn1"]
        bst1 = K.zeros_stats(K.channels(R["y1"]), R["y1"])
        da = R["cv2"].dgrad_act(R["cv2"].pack(_w(blk.conv2), 1), dy2, R["y1"], bn1[0], bn1[1], bn1[2], bn1[3], ACT_RELU, bst1)
        pqr1 = _bn_back(grads, blk.bn1, bn1, bst1, K.per_channel(R["y1"]))
        del dy2
    else:
        da = R["cv2"].dgrad(R["cv2"].pack(_w(blk.conv2), 1), dy2)
        del dy2
        da, pqr1 = _relu_bn_back(grads, da, R["y1"], R["bn1"], blk.bn1)
    da = K.bn_bwd_apply_(da, R["y1"], pqr1)                         # da = dL/dy1
    grads.wgrad(R["cv1"], blk.conv1.weight, da)
    dz = R["cv1"].dgrad(R["cv1"].pack(_w(blk.conv1), 1), da)
    del da
    if ds:
        bst = K.zeros_stats(K.channels(g), g)                       # BNd statistics of g (values of g unchanged)
        g = K.affine_act_bwd_(g, R["yd"], None, None, None, None, bnd[2], bnd[3], ACT_NONE, bst, stats_only=True)
        pqrd = _bn_back(grads, blk.downsample[1], bnd, bst, K.per_channel(g))
        g = K.bn_bwd_apply_(g, R["yd"], pqrd)                       # g = dL/dyd
        grads.wgrad(R["cvd"], blk.downsample[0].weight, g)
        return dz, R["cvd"].dgrad(R["cvd"].pack(_w(blk.downsample[0]), 1), g)
    return dz, g


def stem_backward(f, S, g, grads, g2=None):
    y0, bn0 = S["y0"], S["bn0"]
    g = g.contiguous()
    if g2 is not None and not K.is_b16(y0):
        g = K.to_f32(g).add_(K.to_f32(g2))
        g2 = None                                                   # B16 images: the two kernels below add g2 on the fly
    bst = K.zeros_stats(K.channels(y0), y0)
    K.maxpool_bn_relu_bwd_stats(g, S["idx"], y0, bn0, bst, g2)      # sums over the pooled grid: dz is never written
    pqr0 = _bn_back(grads, f[1], bn0, bst, K.per_channel(y0))
    cs = S.get("cs")
    if cs is not None and K.is_b16(y0) and cs.io_formats(2)[0] == K.FMT_B16:
        # bf16 mode: the weight gradient in the conv's stride-1 form over the space-to-depth frames (one 16-channel B16 block:
        # csrc/wgrad_b16.hip wgradb_ci16_kernel), mapped back to the 7x7 taps by a gather; dy0 stays a B16 image
        dy0 = K.maxpool_bn_relu_bwd_apply(g, S["idx"], y0, bn0, pqr0, g2)
        dw2, _ = cs.wgrad(dy0)
        grads.add(f[0].weight, _stem_s2d_weight_grad(dw2, f[0].in_channels))
        return
    # pool backward + ReLU mask + BatchNorm backward; fp32 out: the 7x7/s2 weight gradient over 3 channels is an fp32 kernel
    dy0 = K.maxpool_bn_relu_bwd_apply(g, S["idx"], y0, bn0, pqr0, g2, out_f32=True)
    grads.wgrad(S["cv0"], f[0].weight, dy0)                         # the frames need no gradient


def trunk_backward(net, S, dout, grads):
    g = fc_backward(net.fc, S["cvf"], dout, grads)                  # dL/dz of the last block
    g2 = None
    for R in reversed(S["blocks"]):
        g, g2 = block_backward(R, g, grads, g2)
    stem_backward(net.features, S, g, grads, g2)


class _ResnetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, x, *params):
        S, out = trunk_forward(net, x, net.training)
        ctx.S, ctx.net, ctx.training = S, net, net.training
        return out

    @staticmethod
    def backward(ctx, dout):
        if not ctx.training:
            raise lib.AvsepError("backward through the visual trunk needs train mode (batch statistics)")
        grads = ParamGrads(ctx.net)
        trunk_backward(ctx.net, ctx.S, dout.contiguous(), grads)
        ctx.S = None
        return (None, None, *grads.finish(param_list(ctx.net), "frame"))


def run(net, x):
    """fc(features(x)) for frames x [N,3,H,W] on the HIP kernels."""
    lib.require_gpu(x)
    return _ResnetFn.apply(net, x.contiguous().float(), *param_list(net))
