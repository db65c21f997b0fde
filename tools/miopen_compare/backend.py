"""COMPARISON TOOL, not product code: the package's visual trunk is models/vision_hip.py and nothing else.

ResNet-18 (dilated) frame trunk + fc conv, "hybrid" backend: convolutions on MIOpen's NHWC fp32 implicit-GEMM
kernels (through aten::convolution / convolution_backward on channels-last tensors), everything between them —
train-mode BatchNorm statistics, normalise + residual + ReLU, and their backward with the BatchNorm gradient folded
as dy = p*dz + q*y + r — on the channels-last kernels of this tool (ops_nhwc.hip -> libavsep_nhwc_gfx950.so).

Reference: models/vision_net.py:62-147 + torchvision BasicBlock.  Same launch plan as models/vision_hip.py except
that relu(bn1(y1)) is materialised (MIOpen cannot fold it into its operand load).  Compared with the plain
PyTorch-ROCm module graph ("torch" backend) this removes one full read+write pass per BatchNorm in each direction
and the separate add / ReLU / threshold kernels of every BasicBlock tail; ONE autograd node for the whole trunk.
"""
import ctypes as C
import os
import subprocess
import sys
import types

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
import avsep_amd as P                                     # noqa: E402
from avsep_amd import kernels as K, lib                   # noqa: E402
from avsep_amd.lib import ACT_NONE, ACT_RELU              # noqa: E402
from avsep_amd.models.audio_net import BN_EPS, BN_MOMENTUM, _bn_run       # noqa: E402
from avsep_amd.models.vision_hip import blocks_of, param_list             # noqa: E402

LIB_PATH = os.path.join(_HERE, "libavsep_nhwc_gfx950.so")
_P, _I, _F, _Z = C.c_void_p, C.c_int32, C.c_float, C.c_size_t
SIGNATURES = {
    "avsep_nhwc_stats_workspace_bytes": (C.c_size_t, [C.c_int64, _I]),
    "avsep_nhwc_channel_stats": (C.c_int, [_P, C.c_int64, _I, _P, _P, _Z, _P]),
    "avsep_nhwc_affine_act": (C.c_int, [_P, _P, _P, _P, _P, _P, _I, C.c_int64, _I, _P, _P]),
    "avsep_nhwc_bn_train_stats": (C.c_int, [_P, C.c_int64, _I, _P, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P, _Z, _P]),
    "avsep_nhwc_affine_act_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, C.c_int64, _I, _P, _P, _P, _P, _P,
                                            _P, _P, _Z, _P]),
    "avsep_nhwc_bn_bwd_apply": (C.c_int, [_P, _P, _P, C.c_int64, _I, _P, _P]),
    "avsep_nhwc_maxpool_bn_relu_fwd": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "avsep_nhwc_maxpool_bn_relu_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _Z, _P]),
}
_lib = None


def build():
    subprocess.run(["make", "-C", _HERE], check=True)


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def _call(name, *args):
    rc = getattr(_load(), name)(*args, lib.stream())
    if rc != 0:
        raise lib.AvsepError(f"{name} failed ({rc})")


def ptr_cl(t):
    """Base pointer of a dense channels-last ([N,H,W,C] storage) tensor, for the avsep_nhwc_* entry points."""
    if t is None:
        return None
    assert t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last), "expected a dense channels_last tensor"
    return t.data_ptr()


# ---- channels-last ([N,C,H,W] tensors with torch.channels_last strides = dense [N*H*W, C]) ----------------------
def _cl(t):
    """(M, C) of a dense channels-last 4-D tensor."""
    ptr_cl(t)
    return t.numel() // t.shape[1], t.shape[1]


def _nhwc_ws(M, Cc, like):
    nbytes = _load().avsep_nhwc_stats_workspace_bytes(M, Cc)
    return torch.empty((nbytes // 4,), dtype=torch.float32, device=like.device), nbytes


def K_nhwc_channel_stats(x, stats):
    M, Cc = _cl(x)
    ws, nbytes = _nhwc_ws(M, Cc, x)
    _call("avsep_nhwc_channel_stats", ptr_cl(x), M, Cc, lib.ptr(stats), lib.ptr(ws), nbytes)


def K_nhwc_bn_train_stats(x, gamma, beta, rmean, rvar, momentum, eps, num_batches_tracked=None):
    """Train-mode BatchNorm2d statistics of a channels-last tensor + finalisation: rows (scale, shift, mean, invstd);
    updates the running statistics and (if given) the int64 num_batches_tracked buffer."""
    M, Cc = _cl(x)
    ws, nbytes = _nhwc_ws(M, Cc, x)
    out = K._f32((4, Cc), x)
    _call("avsep_nhwc_bn_train_stats", ptr_cl(x), M, Cc, lib.ptr(gamma), lib.ptr(beta), lib.ptr(rmean), lib.ptr(rvar),
         lib.ptr(num_batches_tracked), float(momentum), float(eps), lib.ptr(out[0]), lib.ptr(out[1]), lib.ptr(out[2]), lib.ptr(out[3]), lib.ptr(ws), nbytes)
    return out


def K_nhwc_affine_act(y, scale, shift, residual, act, res_scale=None, res_shift=None):
    M, Cc = _cl(y)
    z = torch.empty_like(y)                       # preserves the channels_last strides
    _call("avsep_nhwc_affine_act", ptr_cl(y), lib.ptr(scale), lib.ptr(shift), ptr_cl(residual), lib.ptr(res_scale),
         lib.ptr(res_shift), act, M, Cc, ptr_cl(z))
    return z


def K_nhwc_affine_act_bwd_(dz, y, scale, shift, residual, mean, invstd, act, bstats, res_scale=None, res_shift=None,
                         stats_only=False, dz2=None, gamma=None, coeffs=False):
    """dz <- act'(scale*y+shift [+res]) * (dz [+ dz2]) in place (or statistics only); writes bstats.  With `coeffs` the
    second stage also produces (dgamma, dbeta, pqr) of bn(y) with weight `gamma`, which are returned."""
    M, Cc = _cl(y)
    ws, nbytes = _nhwc_ws(M, Cc, y) if (bstats is not None or coeffs) else (None, 0)
    dgamma = dbeta = pqr = None
    if coeffs:
        dgamma, dbeta, pqr = K._f32((Cc,), y), K._f32((Cc,), y), K._f32((3, Cc), y)
    _call("avsep_nhwc_affine_act_bwd", ptr_cl(dz), ptr_cl(dz2), ptr_cl(y), lib.ptr(scale), lib.ptr(shift), ptr_cl(residual),
         lib.ptr(res_scale), lib.ptr(res_shift), lib.ptr(mean), lib.ptr(invstd), act, M, Cc, None if stats_only else ptr_cl(dz),
         lib.ptr(bstats), lib.ptr(gamma), lib.ptr(dgamma), lib.ptr(dbeta), lib.ptr(pqr), lib.ptr(ws), nbytes)
    return (dgamma, dbeta, pqr) if coeffs else dz


def K_nhwc_bn_bwd_apply_(dz, y, pqr, out=None):
    M, Cc = _cl(y)
    dst = dz if out is None else out
    _call("avsep_nhwc_bn_bwd_apply", ptr_cl(dz), ptr_cl(y), lib.ptr(pqr), M, Cc, ptr_cl(dst))
    return dst


def K_nhwc_maxpool_bn_relu(y, scale, shift):
    """MaxPool2d(3,2,1) of relu(scale*y+shift), channels-last, activated map never materialised -> (pooled, taps)."""
    M, Cc = _cl(y)
    N, _, H, W = y.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out = torch.empty((N, Cc, Ho, Wo), dtype=torch.float32, device=y.device).contiguous(memory_format=torch.channels_last)
    taps = torch.empty((N * Ho * Wo * (Cc // 4),), dtype=torch.int32, device=y.device)
    _call("avsep_nhwc_maxpool_bn_relu_fwd", ptr_cl(y), lib.ptr(scale), lib.ptr(shift), N, H, W, Cc, ptr_cl(out), lib.ptr(taps))
    return out, taps


def K_nhwc_maxpool_bn_relu_bwd(g, taps, y, bnrow, gamma):
    """Backward of nhwc_maxpool_bn_relu + the stem BatchNorm: returns (dgamma, dbeta, dy)."""
    M, Cc = _cl(y)
    _cl(g)
    N, _, H, W = y.shape
    ws, nbytes = _nhwc_ws(M, Cc, y)
    dgamma, dbeta, pqr = K._f32((Cc,), y), K._f32((Cc,), y), K._f32((3, Cc), y)
    args = (ptr_cl(g), lib.ptr(taps), ptr_cl(y), lib.ptr(bnrow[0]), lib.ptr(bnrow[1]), lib.ptr(bnrow[2]), lib.ptr(bnrow[3]), lib.ptr(gamma),
            N, H, W, Cc, lib.ptr(dgamma), lib.ptr(dbeta), lib.ptr(pqr))
    _call("avsep_nhwc_maxpool_bn_relu_bwd", *args, None, lib.ptr(ws), nbytes)
    dy = torch.empty_like(y)
    _call("avsep_nhwc_maxpool_bn_relu_bwd", *args, ptr_cl(dy), lib.ptr(ws), nbytes)
    return dgamma, dbeta, dy




def _acc(grads, p, g):      # comparison-only backend: plain dict of gradients handed back to autograd
    if g is not None:
        grads[p] = grads[p] + g if p in grads else g

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
    rows = K_nhwc_bn_train_stats(y, bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, BN_MOMENTUM,
                                 BN_EPS, bn.num_batches_tracked)   # statistics + finalisation + counter: two launches
    return y, rows


def trunk_forward(net, x, training):
    f = net.features
    S = {"x": x}
    S["y0"], S["bn0"] = _conv_bn(x, f[0], f[1], training)
    z, S["taps"] = K_nhwc_maxpool_bn_relu(S["y0"], S["bn0"][0], S["bn0"][1])   # relu(bn(y0)) is never materialised
    S["blocks"] = []
    for blk in blocks_of(f):
        R = {"mod": blk, "z": z}
        R["y1"], R["bn1"] = _conv_bn(z, blk.conv1, blk.bn1, training)
        R["a1"] = K_nhwc_affine_act(R["y1"], R["bn1"][0], R["bn1"][1], None, ACT_RELU)
        R["y2"], R["bn2"] = _conv_bn(R["a1"], blk.conv2, blk.bn2, training)
        if blk.downsample is not None:
            R["yd"], R["bnd"] = _conv_bn(z, blk.downsample[0], blk.downsample[1], training)
            z = K_nhwc_affine_act(R["y2"], R["bn2"][0], R["bn2"][1], R["yd"], ACT_RELU, R["bnd"][0], R["bnd"][1])
        else:
            z = K_nhwc_affine_act(R["y2"], R["bn2"][0], R["bn2"][1], R["z"], ACT_RELU)
        S["blocks"].append(R)
    S["zf"] = z
    return S, _conv(z, net.fc)


def _relu_bn_back(grads, g, y, bnrow, bn_mod, res=None, rs=None, rh=None, g2=None, act=ACT_RELU, stats_only=False):
    """g <- act'(.) * (g [+ g2]) in place (g2: the other incoming gradient of a residual join, summed on the fly);
    accumulates the BatchNorm parameter gradients of bn(y) and returns its folded-gradient coefficients (p, q, r)."""
    dgamma, dbeta, pqr = K_nhwc_affine_act_bwd_(g, y, bnrow[0] if not stats_only else None,
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
        dy2 = K_nhwc_bn_bwd_apply_(g, R["y2"], pqr2, out=torch.empty_like(g))
        da = _conv_back(grads, dy2, R["a1"], blk.conv2)
        del dy2
        pqr1 = _relu_bn_back(grads, da, R["y1"], R["bn1"], blk.bn1)
        K_nhwc_bn_bwd_apply_(da, R["y1"], pqr1)                         # da = dL/dy1
        dz = _conv_back(grads, da, R["z"], blk.conv1)
        del da
        if ds:
            pqrd = _relu_bn_back(grads, g, R["yd"], bnd, blk.downsample[1], act=ACT_NONE, stats_only=True)   # g unchanged
            K_nhwc_bn_bwd_apply_(g, R["yd"], pqrd)                      # g = dL/dyd
            g2 = _conv_back(grads, g, R["z"], blk.downsample[0])
        else:
            g2 = g                                                      # identity branch
        g = dz                                                          # the join (g + g2) is summed by the next consumer
    g.add_(g2)
    dgamma, dbeta, dy0 = K_nhwc_maxpool_bn_relu_bwd(g, S["taps"], S["y0"], S["bn0"], f[1].weight.detach())
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



def install(net, backend):
    """Route `net`'s trunk (a ResnetFC / ResnetDilated of the package) through MIOpen for a comparison run:
    "hybrid" = MIOpen NHWC convolutions + this tool's channels-last glue kernels, "torch" = the plain module graph."""
    if backend == "hybrid":
        net._trunk = types.MethodType(lambda self, x: run(self, x), net)
    elif backend == "torch":
        net._trunk = types.MethodType(
            lambda self, x: self.fc(self.features(x.contiguous(memory_format=torch.channels_last))), net)
    else:
        raise ValueError(backend)
    net.backend = backend
    # note: the package's FlatSGD keeps conv weights OIHW; MIOpen's NHWC kernels re-lay them out per call (round 3 kept
    # them OHWI inside the flat buffers for this comparison: 533 mixtures/s at batch 64; round 4 measures 531 without it)
    return net
