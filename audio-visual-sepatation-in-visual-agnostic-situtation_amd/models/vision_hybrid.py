"""ResNet-18 (dilated) frame trunk + fc conv, "hybrid" backend: convolutions on MIOpen's NHWC fp32 implicit-GEMM
kernels (through aten::convolution / convolution_backward on channels-last tensors), everything between them —
train-mode BatchNorm statistics, normalise + residual + ReLU, and their backward with the BatchNorm gradient folded
as dy = p*dz + q*y + r — on this library's channels-last kernels (csrc/ops_nhwc.hip).

Reference: models/vision_net.py:62-147 + torchvision BasicBlock.  Same launch plan as models/vision_hip.py except
that relu(bn1(y1)) is materialised (MIOpen cannot fold it into its operand load).  Compared with the plain
PyTorch-ROCm module graph ("torch" backend) this removes one full read+write pass per BatchNorm in each direction
and the separate add / ReLU / threshold kernels of every BasicBlock tail; ONE autograd node for the whole trunk.
"""
import torch

from .. import kernels as K
from .. import lib
from ..lib import ACT_NONE, ACT_RELU
from .audio_net import BN_EPS, BN_MOMENTUM, _bn_run


def _acc(grads, p, g):      # comparison-only backend: plain dict of gradients handed back to autograd
    if g is not None:
        grads[p] = grads[p] + g if p in grads else g
from .vision_hip import blocks_of, param_list

aten = torch.ops.aten


def _conv(x, conv):
    return aten.convolution(x, conv.weight, conv.bias, conv.stride, conv.padding, conv.dilation, False, [0, 0], 1)


def _conv_back(grads, g, x, conv, need_dx=True):
    bias = [conv.out_channels] if conv.bias is not None else None
    dx, dw, db = aten.convolution_backward(g, x, conv.weight, bias, conv.stride, conv.padding, conv.dilation, False,
                                           [0, 0], 1, [need_dx, True, conv.bias is not None])
    _acc(grads, conv.weight, dw)
    if conv.bias is not None:
        _acc(grads, conv.bias, db)
    return dx


def _conv_bn(x, conv, bn, training):
    y = _conv(x, conv)
    if not training:
        return y, _bn_run(bn, None, y.numel() // y.shape[1], False, y)
    rows = K.nhwc_bn_train_stats(y, bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, BN_MOMENTUM,
                                 BN_EPS, bn.num_batches_tracked)   # statistics + finalisation + counter: two launches
    return y, rows


def trunk_forward(net, x, training):
    f = net.features
    S = {"x": x}
    S["y0"], S["bn0"] = _conv_bn(x, f[0], f[1], training)
    z, S["taps"] = K.nhwc_maxpool_bn_relu(S["y0"], S["bn0"][0], S["bn0"][1])   # relu(bn(y0)) is never materialised
    S["blocks"] = []
    for blk in blocks_of(f):
        R = {"mod": blk, "z": z}
        R["y1"], R["bn1"] = _conv_bn(z, blk.conv1, blk.bn1, training)
        R["a1"] = K.nhwc_affine_act(R["y1"], R["bn1"][0], R["bn1"][1], None, ACT_RELU)
        R["y2"], R["bn2"] = _conv_bn(R["a1"], blk.conv2, blk.bn2, training)
        if blk.downsample is not None:
            R["yd"], R["bnd"] = _conv_bn(z, blk.downsample[0], blk.downsample[1], training)
            z = K.nhwc_affine_act(R["y2"], R["bn2"][0], R["bn2"][1], R["yd"], ACT_RELU, R["bnd"][0], R["bnd"][1])
        else:
            z = K.nhwc_affine_act(R["y2"], R["bn2"][0], R["bn2"][1], R["z"], ACT_RELU)
        S["blocks"].append(R)
    S["zf"] = z
    return S, _conv(z, net.fc)


def _relu_bn_back(grads, g, y, bnrow, bn_mod, res=None, rs=None, rh=None, g2=None, act=ACT_RELU, stats_only=False):
    """g <- act'(.) * (g [+ g2]) in place (g2: the other incoming gradient of a residual join, summed on the fly);
    accumulates the BatchNorm parameter gradients of bn(y) and returns its folded-gradient coefficients (p, q, r)."""
    dgamma, dbeta, pqr = K.nhwc_affine_act_bwd_(g, y, bnrow[0] if not stats_only else None,
                                                bnrow[1] if not stats_only else None, res, bnrow[2], bnrow[3], act, None,
                                                res_scale=rs, res_shift=rh, dz2=g2, stats_only=stats_only,
                                                gamma=bn_mod.weight.detach(), coeffs=True)
    _acc(grads, bn_mod.weight, dgamma)
    _acc(grads, bn_mod.bias, dbeta)
    return pqr


def trunk_backward(net, S, dout, grads):
    f = net.features
    g, g2 = _conv_back(grads, dout, S["zf"], net.fc), None           # dL/dz of the last block (+ its second branch)
    for R in reversed(S["blocks"]):
        blk = R["mod"]
        ds = blk.downsample is not None
        bnd = R.get("bnd")
        pqr2 = _relu_bn_back(grads, g, R["y2"], R["bn2"], blk.bn2, res=R["yd"] if ds else R["z"],
                             rs=bnd[0] if ds else None, rh=bnd[1] if ds else None, g2=g2)   # g = dL/d(pre-ReLU sum)
        dy2 = K.nhwc_bn_bwd_apply_(g, R["y2"], pqr2, out=torch.empty_like(g))
        da = _conv_back(grads, dy2, R["a1"], blk.conv2)
        del dy2
        pqr1 = _relu_bn_back(grads, da, R["y1"], R["bn1"], blk.bn1)
        K.nhwc_bn_bwd_apply_(da, R["y1"], pqr1)                         # da = dL/dy1
        dz = _conv_back(grads, da, R["z"], blk.conv1)
        del da
        if ds:
            pqrd = _relu_bn_back(grads, g, R["yd"], bnd, blk.downsample[1], act=ACT_NONE, stats_only=True)   # g unchanged
            K.nhwc_bn_bwd_apply_(g, R["yd"], pqrd)                      # g = dL/dyd
            g2 = _conv_back(grads, g, R["z"], blk.downsample[0])
        else:
            g2 = g                                                      # identity branch
        g = dz                                                          # the join (g + g2) is summed by the next consumer
    g.add_(g2)
    dgamma, dbeta, dy0 = K.nhwc_maxpool_bn_relu_bwd(g, S["taps"], S["y0"], S["bn0"], f[1].weight.detach())
    _acc(grads, f[1].weight, dgamma)
    _acc(grads, f[1].bias, dbeta)
    _conv_back(grads, dy0, S["x"], f[0], need_dx=False)                 # the frames need no gradient


class _ResnetHybridFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, x, *params):
        S, out = trunk_forward(net, x, net.training)
        ctx.S, ctx.net, ctx.training = S, net, net.training
        return out

    @staticmethod
    def backward(ctx, dout):
        if not ctx.training:
            raise lib.AvsepError("backward through the visual trunk needs train mode (batch statistics)")
        grads = {}
        trunk_backward(ctx.net, ctx.S, dout.contiguous(memory_format=torch.channels_last), grads)
        ctx.S = None
        return (None, None, *[grads.get(p) for p in param_list(ctx.net)])


def run(net, x):
    """fc(features(x)) for frames x [N,3,H,W]: MIOpen convolutions + channels-last HIP BatchNorm/ReLU/residual."""
    lib.require_gpu(x)
    return _ResnetHybridFn.apply(net, x.float().contiguous(memory_format=torch.channels_last), *param_list(net))
