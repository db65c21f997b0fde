"""Audio U-Net on the HIP kernels (drop-in for the reference's models/audio_net.py:10-203).

Same constructor, same ``forward(x, v) -> (feat, (match_loss, att_maps))`` contract and the
same ``state_dict`` key names (``bn0.*``, ``unet_block[.mid_forward]^k.down_forward.{0|1}.weight``,
``...down_forward.2.*``, ``...up_forward.2.*``, ``...up_forward.3.*``).  The whole network is ONE
autograd node: forward and backward are explicit launch sequences over libavsep_gfx950.so
(implicit-GEMM convolutions with BatchNorm/LeakyReLU/ReLU/concat folded into their operand
gathers and BN statistics into their epilogues), not a graph of torch operators.

Reference quirks reproduced: the skip tensor is the LeakyReLU'd block
input (in-place aliasing, audio_net.py:64,119-122) so the decoder sees ReLU(z); there is no
BatchNorm after the innermost down conv nor around the outermost block; the fusion sits between
the innermost down and up convs.
"""
import os

import torch
import torch.nn as nn

from .. import kernels as K
from .. import lib
from ..lib import ACT_LRELU02, ACT_NONE, ACT_RELU
from . import fusion_net

BN_EPS, BN_MOMENTUM = 1e-5, 0.1


class _Slots(nn.Module):
    def put(self, idx, mod):
        self.add_module(str(idx), mod)
        return mod

    def at(self, idx):
        return getattr(self, str(idx))


class Conv2dParams(nn.Module):
    """Parameter holder (class name contains 'Conv' so the reference-style weights_init applies)."""

    def __init__(self, cin, cout, k, bias):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        self.bias = nn.Parameter(torch.empty(cout)) if bias else None
        self.kernel_size, self.in_channels, self.out_channels = (k, k), cin, cout
        # nn.Conv2d default init (kaiming_uniform a=sqrt(5)); ModelBuilder.weights_init overrides it
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)
        if bias:
            bound = 1.0 / (cin * k * k) ** 0.5
            nn.init.uniform_(self.bias, -bound, bound)


class BatchNorm2dParams(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self.num_features = c


class _Level(nn.Module):
    def __init__(self, outer_nc, inner_nc, in_nc, up_in_nc, kind, child):
        super().__init__()
        self.kind = kind
        self.down_forward = _Slots()
        if child is not None:
            self.mid_forward = child
        self.up_forward = _Slots()
        self.down_conv = self.down_forward.put(0 if kind == "outer" else 1, Conv2dParams(in_nc, inner_nc, 4, False))
        self.down_bn = self.down_forward.put(2, BatchNorm2dParams(inner_nc)) if kind == "mid" else None
        self.up_conv = self.up_forward.put(2, Conv2dParams(up_in_nc, outer_nc, 3, kind == "outer"))
        self.up_bn = self.up_forward.put(3, BatchNorm2dParams(outer_nc)) if kind != "outer" else None

    # the aliases above are plain attributes, not extra registrations
    def __setattr__(self, name, value):
        if name in ("down_conv", "down_bn", "up_conv", "up_bn"):
            object.__setattr__(self, name, value)
        else:
            super().__setattr__(name, value)


class Unet(nn.Module):
    def __init__(self, fc_dim=64, num_downs=5, ngf=64, use_dropout=False, fusion_type="con_motion",
                 att_type="cos", fuse_upsample=False, extra_size=None):
        super().__init__()
        if use_dropout:
            raise NotImplementedError("use_dropout is never enabled by the reference builders")
        self.extra_size = extra_size
        self.fusion_type, self.att_type = fusion_type, att_type
        self.fuse_upsample = fuse_upsample
        self.fuse_head = True    # decoder head (<= 4 output channels) on the fused kernels of csrc/head.hip
        if extra_size is None:
            self.fusion = fusion_net.get_fusion_net(fusion_type)(att_type=att_type)
            self.fusion.num_src = fc_dim if 2 < fc_dim <= 4 else 2    # one output channel per source (num_channels == num_mix)
            lvl = _Level(ngf * 8, ngf * 8, ngf * 8, ngf * 16, "inner", None)
        else:
            # SoP++ variant (SoP++/audio_net.py:151-198): the bottleneck conv also emits 2*extra_size
            # per-source "weight" channels, split off before the decoder; no fusion
            self.fusion = None
            lvl = _Level(ngf * 8, ngf * 8 + 2 * extra_size, ngf * 8, ngf * 8, "inner", None)
        for _ in range(num_downs - 5):
            lvl = _Level(ngf * 8, ngf * 8, ngf * 8, ngf * 16, "mid", lvl)
        lvl = _Level(ngf * 4, ngf * 8, ngf * 4, ngf * 16, "mid", lvl)
        lvl = _Level(ngf * 2, ngf * 4, ngf * 2, ngf * 8, "mid", lvl)
        lvl = _Level(ngf, ngf * 2, ngf, ngf * 4, "mid", lvl)
        lvl = _Level(fc_dim, ngf, 1, ngf * 2, "outer", lvl)
        self.bn0 = BatchNorm2dParams(1)
        self.unet_block = lvl
        self.ao_draws = None  # tests may pin the audio-only random swap

    # the fusion module is parameter free and must not add state_dict keys
    def __setattr__(self, name, value):
        if name == "fusion":
            object.__setattr__(self, name, value)
        else:
            super().__setattr__(name, value)

    def levels(self):
        out, l = [], self.unet_block
        while True:
            out.append(l)
            if l.kind == "inner":
                return out
            l = l.mid_forward

    def param_list(self):
        """Flat, fixed order of every trainable tensor (the argument order of the autograd node)."""
        ps = [self.bn0.weight, self.bn0.bias]
        for l in self.levels():
            ps.append(l.down_conv.weight)
            if l.down_bn is not None:
                ps += [l.down_bn.weight, l.down_bn.bias]
            ps.append(l.up_conv.weight)
            if l.up_conv.bias is not None:
                ps.append(l.up_conv.bias)
            if l.up_bn is not None:
                ps += [l.up_bn.weight, l.up_bn.bias]
        return ps

    def forward_pair(self, x, v_a, v_b, before_decode=None):
        """Two audio-visual passes on the same input with different visual inputs (the AV step of main.py:113-148),
        sharing the encoder.  Returns ((feat_a, meta_a), (feat_b, meta_b)) exactly like two forward() calls.
        `before_decode`: called once the encoder has been issued and before anything reads the visual inputs (they may
        still be in flight on other streams: NetWrapper.forward)."""
        lib.require_gpu(x)
        if self.extra_size is not None or len(v_a) != len(v_b) or not 2 <= len(v_a) <= 4:
            if before_decode is not None:
                before_decode()
            return self.forward(x, v_a), self.forward(x, v_b)
        if before_decode is not None and not all(t.is_contiguous() and t.dtype == torch.float32 for t in (*v_a, *v_b)):
            before_decode()
            before_decode = None
        vs = [t.contiguous().float() for t in (*v_a, *v_b)]
        object.__setattr__(self, "_before_decode", before_decode)
        fa, ma, aa, fb, mb, ab = _UnetPairFn.apply(self, len(v_a), x.contiguous().float(), *vs, *self.param_list())
        return (fa, (ma, aa)), (fb, (mb, ab))

    def forward(self, x, v=None):
        lib.require_gpu(x)
        B = x.shape[0]
        if self.extra_size is not None:
            feat, extra, _ = _UnetFn.apply(self, None, 0, x.contiguous().float(), *self.param_list())
            return feat, (extra,)
        draws = None
        if v is None:
            draws = self.ao_draws if self.ao_draws is not None else self.fusion.draw(B)
            vs = []
        else:
            vs = [t.contiguous().float() for t in v]
        out = _UnetFn.apply(self, draws, len(vs), x.contiguous().float(), *vs, *self.param_list())
        feat, match_loss, att_maps = out
        if v is None:
            return feat, (None, None)
        return feat, (match_loss, att_maps)


def _bn_run(bn, stats, count, training, like, repeat=1):
    """scale/shift/mean/invstd rows for one BatchNorm; updates the running buffers in training
    (`repeat` times: a shared encoder stands for `repeat` identical forward passes of the reference)."""
    return K.bn_finalize(stats, count, bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var,
                         BN_MOMENTUM, BN_EPS, training, like, bn.num_batches_tracked, repeat)


class ParamGrads:
    """Parameter gradients of one backward pass of an autograd node.

    Without a gradient sink: a dict of tensors that the node returns to autograd (AccumulateGrad then adds each into
    ``p.grad``).  With the step's ``FlatSGD`` attached to the network (``net._grad_sink``, set by create_optimizer): the
    FIRST contribution of a step to a parameter is written by the producing kernel straight into the parameter's zeroed
    flat ``.grad`` view (``dest``), later contributions of the same node are added there, and autograd receives None for
    it — no temporary, no accumulate launch per parameter (124 of the 204 ATen adds of a step were those)."""

    def __init__(self, net):
        self.sink = getattr(net, "_grad_sink", None)
        self.d, self.placed, self.second = {}, {}, {}     # dicts: tensors hash by identity

    def dest(self, p):
        """Tensor a kernel may WRITE the gradient of `p` into, or None (then hand the result to add())."""
        if self.sink is None or p is None or p in self.d or p in self.second:
            return None
        if p in self.placed:             # second contribution of THIS node (the second decoder pass of an AV step)
            v = self.sink.scratch_dest(p)
            if v is not None:
                self.second[p] = v
            return v
        v = self.sink.grad_dest(p)
        if v is None:                    # another node of this step was first: a scratch view, folded in by finish()
            v = self.sink.scratch_dest(p)
            if v is not None:
                self.second[p] = v
        if v is not None:
            self.placed[p] = v
        return v

    def add(self, p, g, wrote=None):
        """Contribution `g` to p's gradient; `wrote` = the destination the kernel has already written it into."""
        if g is None:
            return
        if wrote is not None and g is wrote:
            return
        if p in self.second:
            self.second[p].add_(g)               # the view handed out for it (the scratch buffer is per stream)
        elif p in self.placed:
            self.placed[p].add_(g)
        elif p in self.d:
            self.d[p] = self.d[p] + g
        else:
            v = self.dest(p)
            if v is not None:
                v.copy_(g)
            else:
                self.d[p] = g

    def wgrad(self, cv, p, dy, bias_p=None):
        """Weight (and bias) gradient of conv `cv`, written in place when this is the parameter's first contribution."""
        out = self.dest(p)
        ob = self.dest(bias_p) if bias_p is not None else None
        dw, db = cv.wgrad(dy, want_bias=bias_p is not None, out=out, out_bias=ob)
        self.add(p, dw, out)
        if bias_p is not None:
            self.add(bias_p, db, ob)

    def fold_second(self):
        """Add the scratch contributions placed so far into the gradient buffer NOW, on the current stream — the stream the
        second decoder pass wrote them on when the two passes of an AV step run on forked streams."""
        if self.sink is not None and self.second:
            self.sink.fold_scratch(list(self.second))
            for p in self.second:
                self.placed[p] = self.sink._slot[p]      # any later contribution of this node adds into the gradient view
            self.second = {}

    def finish(self, params, group=None):
        """The tuple autograd gets (None for the parameters already in their flat views)."""
        out = [self.d.get(p) for p in params]
        if self.sink is not None:
            if self.second:
                self.sink.fold_scratch(list(self.second))
            self.sink.node_finished(group, [p for p in params if p in self.d])
        return out


def _bn_back(grads, bn_mod, bnrow, bstats, count):
    dg, db = grads.dest(bn_mod.weight), grads.dest(bn_mod.bias)
    dgamma, dbeta, pqr = K.bn_bwd_coeffs(bstats, count, bn_mod.weight.detach(), bnrow[2], bnrow[3], dg, db)
    grads.add(bn_mod.weight, dgamma, dg)
    grads.add(bn_mod.bias, dbeta, db)
    return pqr


def _encode(net, x, training, repeat=1):
    """bn0 + the down convs (conv k4 s2 p1; the previous level's BatchNorm + LeakyReLU are folded into the gather)."""
    lv = net.levels()
    B, _, H, W = x.shape
    E = {"x": x, "lv": lv}
    st0 = K.zeros_stats(1, x)
    if training:
        K.channel_stats(x, st0)
    E["bn0"] = _bn_run(net.bn0, st0, B * H * W, training, x, repeat)     # audio_net.py:37,41
    src, aff, act = x, E["bn0"], ACT_NONE
    E["dconv"], E["yd"], E["dbn"] = [], [], []
    for l in lv:
        w = l.down_conv.weight.detach()
        cv = K.Conv(src, w.shape[0], 4, 2, 1, sc0=aff[0] if aff is not None else None,
                    sh0=aff[1] if aff is not None else None, act0=act)
        st = K.zeros_stats(w.shape[0], x) if (l.down_bn is not None and training) else None
        y = cv.fwd(cv.pack(w, 0), None, st)
        bn = _bn_run(l.down_bn, st, K.per_channel(y), training, x, repeat) if l.down_bn is not None else None
        E["dconv"].append(cv); E["yd"].append(y); E["dbn"].append(bn)
        src, aff, act = y, bn, ACT_LRELU02
    return E


def _decode(net, E, vs, draws, training):
    """Bottleneck fusion (or the SoP++ split) + the up path: ReLU + bilinear x2 + conv k3 p1 over concat(skip, inner)."""
    lv, x = E["lv"], E["x"]
    L = len(lv)
    ybot = K.to_f32(E["yd"][-1])             # the fusion kernels and the SoP++ split read fp32
    extra = fus = feat_vec = None
    if net.extra_size is None:
        fus = net.fusion.run_forward(ybot, vs, draws)
        feat_vec = fus["feat"]                               # [B, D] broadcast vectors
    else:
        e2 = 2 * net.extra_size
        extra, ybot = ybot[:, :e2].contiguous(), ybot[:, e2:].contiguous()
    D = {"fus": fus, "ybot": ybot, "extra": extra, "vs": vs,
         "uconv": [None] * L, "yu": [None] * L, "ubn": [None] * L, "cat": [None] * L}
    for i in range(L - 1, -1, -1):
        l = lv[i]
        w = l.up_conv.weight.detach()
        if i == L - 1:
            cat = K.Cat(feat_vec, ybot, bcast0=True) if feat_vec is not None else K.Cat(ybot, None)
            fused = False
        else:
            dbn, ubn = E["dbn"][i], D["ubn"][i + 1]
            cat = K.Cat(E["yd"][i], D["yu"][i + 1], sc0=dbn[0] if dbn is not None else None,
                        sh0=dbn[1] if dbn is not None else None, sc1=ubn[0], sh1=ubn[1])
            fused = net.fuse_upsample
        cv = None
        if i < L - 1:   # upsample folded into the conv's operand staging: nothing materialised
            cv = K.Conv(cat.keep[0], w.shape[0], 3, 1, 1, x1=cat.keep[1], sc0=cat.keep[2], sh0=cat.keep[3],
                        act0=ACT_RELU, sc1=cat.keep[4], sh1=cat.keep[5], act1=ACT_RELU, up2x=True)
            cv.head = cv.head_applicable() and net.fuse_head     # few-output-channel head: always worth fusing
            fused = fused or cv.head
        if not fused:
            cv = K.Conv(cat.fwd(), w.shape[0], 3, 1, 1)
            cv.head = False
        st = K.zeros_stats(w.shape[0], x) if (l.up_bn is not None and training) else None
        bias = l.up_conv.bias.detach() if l.up_conv.bias is not None else None
        # the level below the fused head feeds fp32 kernels only (head forward / gradients, BatchNorm backward on fp32 dz)
        y = cv.fwd(cv.pack(w, 0), bias, st, out_b16=False if (i == 1 and net.fuse_head) else None)
        bn = _bn_run(l.up_bn, st, K.per_channel(y), training, x) if l.up_bn is not None else None
        D["uconv"][i], D["yu"][i], D["ubn"][i], D["cat"][i] = cv, y, bn, cat
    return D


def _decode_bwd(net, E, D, dlogits, dsecond, has_vis, grads, Gd, order=None):
    """Decoder + fusion backward, outermost -> innermost.  Accumulates parameter gradients into `grads` and the
    skip-path gradients into Gd[i] (gradient reaching BN(yd[i]), already ReLU-masked); returns (dbot, dvs).
    `order` = ("record" | "wait", events): the two passes of an AV step on forked streams — the first records an event
    after it has written Gd[i], the second waits for it before accumulating into the same tensor."""
    lv, x = E["lv"], E["x"]

    def gd_sync(i, when):
        if order is None:
            return
        mode, evs = order
        if mode == "record" and when == "after":
            evs[i] = torch.cuda.Event()
            evs[i].record()
        elif mode == "wait" and when == "before" and evs[i] is not None:
            torch.cuda.current_stream().wait_event(evs[i])
    L = len(lv)
    g = dlogits.contiguous()
    dfeat = dbot = None
    for i in range(L):
        l, cv, cat = lv[i], D["uconv"][i], D["cat"][i]
        w = l.up_conv.weight.detach()
        grads.wgrad(cv, l.up_conv.weight, g, l.up_conv.bias)
        if cv.head:                                        # fused head: straight to the low-res sources
            ubn = D["ubn"][i + 1]
            bst = K.zeros_stats(ubn.shape[1], x)
            gd_sync(i, "before")
            Gd[i], dz = cv.dgrad_up2x(w, g, mean1=ubn[2], invstd1=ubn[3], bstats1=bst, g0_acc=Gd[i])
            gd_sync(i, "after")
            yu = D["yu"][i + 1]
            # the consumers of g (level i + 1's weight / data gradient) are bf16 kernels in bf16 mode: write the B16 image directly
            g = K.bn_bwd_apply_(dz, yu, _bn_back(grads, lv[i + 1].up_bn, ubn, bst, K.per_channel(yu)),
                                to_b16_out=K.want_b16(K.channels(yu)))
            continue
        dU = cv.dgrad(cv.pack(w, 1), g, out_b16=cat.b16)   # wrt the (virtual) upsampled input, in the format cat.bwd reads
        if i == L - 1:
            if net.extra_size is None:
                dfeat, dbot = cat.bwd(dU)
            else:
                dxb, _ = cat.bwd(dU)
                dbot = torch.cat([dsecond.to(dxb.dtype).contiguous(), dxb], 1)
        else:
            ubn = D["ubn"][i + 1]
            bst = K.zeros_stats(ubn.shape[1], x)
            gd_sync(i, "before")
            Gd[i], dz = cat.bwd(dU, mean1=ubn[2], invstd1=ubn[3], bstats1=bst, g0_acc=Gd[i])
            gd_sync(i, "after")
            yu = D["yu"][i + 1]
            pqr = _bn_back(grads, lv[i + 1].up_bn, ubn, bst, K.per_channel(yu))
            g = K.bn_bwd_apply_(dz, yu, pqr)
        del dU
    dvs = []
    if net.extra_size is None:   # adds the fusion's gradient wrt the bottleneck into dbot
        dvs = net.fusion.run_backward(D["ybot"], D["vs"], D["fus"], dfeat, dbot, None, dsecond if has_vis else None)
    return dbot, dvs


def _encode_bwd(net, E, dbot, Gd, grads):
    """Encoder backward, innermost -> outermost (LeakyReLU' + skip-gradient add + BN backward between the convs)."""
    lv, x = E["lv"], E["x"]
    g = dbot
    for i in range(len(lv) - 1, -1, -1):
        l, cv = lv[i], E["dconv"][i]
        w = l.down_conv.weight.detach()
        grads.wgrad(cv, l.down_conv.weight, g)
        # wrt act(BN(prev)) (or BN0(x) for i == 0), in the storage format of the tensor whose mask / statistics come next
        dS = cv.dgrad(cv.pack(w, 1), g, out_b16=K.is_b16(E["yd"][i - 1]) if i > 0 else False)
        if i > 0:
            yprev, bn = E["yd"][i - 1], E["dbn"][i - 1]
            bst = K.zeros_stats(K.channels(yprev), x) if bn is not None else None
            dS = K.affine_act_bwd_(dS, yprev, bn[0] if bn is not None else None, bn[1] if bn is not None else None,
                                   None, Gd[i - 1], bn[2] if bn is not None else None,
                                   bn[3] if bn is not None else None, ACT_LRELU02, bst)
            g = K.bn_bwd_apply_(dS, yprev, _bn_back(grads, lv[i - 1].down_bn, bn, bst, K.per_channel(yprev))) \
                if bn is not None else dS
        else:
            bn0 = E["bn0"]
            bst = K.zeros_stats(1, x)
            K.affine_act_bwd_(dS, x, None, None, None, None, bn0[2], bn0[3], ACT_NONE, bst)
            _bn_back(grads, net.bn0, bn0, bst, x.numel())


def _outputs(net, D, x):
    logits = D["yu"][0]
    if net.extra_size is not None:
        return logits, D["extra"], x.new_zeros(())
    if D["vs"]:
        return logits, D["fus"]["match_part"].mean(), D["fus"]["att_maps"]
    return logits, x.new_zeros(()), x.new_zeros(())


class _UnetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, draws, nv, x, *rest):
        vs, training = list(rest[:nv]), net.training
        _release_deferred(net)
        E = _encode(net, x, training)
        D = _decode(net, E, vs, draws, training)
        ctx.E, ctx.D, ctx.net, ctx.nv, ctx.training = E, D, net, nv, training
        out = _outputs(net, D, x)
        if net.extra_size is None:
            ctx.mark_non_differentiable(out[2])
        return out

    @staticmethod
    def backward(ctx, dlogits, dsecond, _datt):
        # dsecond is the cotangent of the second output: the match loss, or the SoP++ `extra` channels
        if not ctx.training:
            raise lib.AvsepError("backward through the U-Net needs train mode (batch statistics)")
        net, E, D = ctx.net, ctx.E, ctx.D
        grads, Gd = ParamGrads(net), [None] * len(E["lv"])
        dbot, dvs = _decode_bwd(net, E, D, dlogits, dsecond, bool(ctx.nv), grads, Gd)
        _encode_bwd(net, E, dbot, Gd, grads)
        ctx.E = ctx.D = None
        return (None, None, None, None, *dvs, *grads.finish(net.param_list(), "sound"))


_FORK_PAIR = os.environ.get("AVSEP_FORK_PAIR", "1") != "0"
_ENC_SIDE = os.environ.get("AVSEP_ENC_SIDE", "1") != "0"


def _release_deferred(net):
    """Drop the tensors a previous backward left to its side stream (the encoder's backward runs there while the caller's
    stream has moved on): first order the current stream behind that stream, so that memory handed back to the current
    stream's pool is not reused under kernels that still read it."""
    if net.__dict__.get("_pair_deferred") is not None:
        s = net.__dict__.get("_pair_side")
        if s is not None:
            torch.cuda.current_stream().wait_stream(s)
        object.__setattr__(net, "_pair_deferred", None)


def _pair_stream(net, x):
    """The side stream of the second decoder pass of an AV step, or None (CPU tensors, AVSEP_FORK_PAIR=0)."""
    if not (x.is_cuda and getattr(net, "fork_pair", _FORK_PAIR)):
        return None
    s = net.__dict__.get("_pair_side")
    if s is None:
        s = torch.cuda.Stream()
        object.__setattr__(net, "_pair_side", s)
    return s


class _UnetPairFn(torch.autograd.Function):
    """The two U-Net passes of an audio-visual step (main.py:128-141: reversed, then natural visual order) as ONE
    node: both passes see the same input, so the encoder (and its batch statistics) is identical — it runs once
    forward, and once backward on the SUM of the two passes' bottleneck / skip gradients (the backward is linear
    in them).  The encoder's BatchNorm running statistics still receive two momentum updates, like two passes."""

    @staticmethod
    def forward(ctx, net, nv, x, *rest):
        training = net.training
        _release_deferred(net)
        E = _encode(net, x, training, repeat=2)
        hook = net.__dict__.pop("_before_decode", None)
        if hook is not None:
            hook()                                   # the visual inputs' streams are joined here, behind the encoder
        side = _pair_stream(net, x)
        if side is None:
            Da = _decode(net, E, list(rest[:nv]), None, training)
            Db = _decode(net, E, list(rest[nv:2 * nv]), None, training)
        else:
            # The two decoder passes only share what they read (encoder activations, weights) and the BatchNorm running
            # statistics: the second is issued on its own stream, kernels.fork_streams orders the statistics updates.
            main, fork = torch.cuda.current_stream(), torch.cuda.Event()
            fork.record()
            with K.fork_streams():
                Da = _decode(net, E, list(rest[:nv]), None, training)
                side.wait_event(fork)
                with torch.cuda.stream(side):
                    Db = _decode(net, E, list(rest[nv:2 * nv]), None, training)
                main.wait_stream(side)
        ctx.E, ctx.Da, ctx.Db, ctx.net, ctx.training = E, Da, Db, net, training
        oa, ob = _outputs(net, Da, x), _outputs(net, Db, x)
        if side is not None:
            for t in ob:
                t.record_stream(torch.cuda.current_stream())
        ctx.mark_non_differentiable(oa[2], ob[2])
        return (*oa, *ob)

    @staticmethod
    def backward(ctx, dla, dma, _a, dlb, dmb, _b):
        if not ctx.training:
            raise lib.AvsepError("backward through the U-Net needs train mode (batch statistics)")
        net, E = ctx.net, ctx.E
        grads, Gd = ParamGrads(net), [None] * len(E["lv"])
        side = _pair_stream(net, E["x"]) if grads.sink is not None else None
        if side is None:
            dbot_a, dva = _decode_bwd(net, E, ctx.Da, dla, dma, True, grads, Gd)
            dbot_b, dvb = _decode_bwd(net, E, ctx.Db, dlb, dmb, True, grads, Gd)
        else:
            # pass A on this stream, pass B on the side stream: B's parameter gradients go to the side stream's scratch buffer
            # (ParamGrads: the second contribution of a node) and are folded in there, after A's direct writes; the
            # accumulation into the shared skip gradients Gd[i] is ordered level by level.
            main, fork, evs = torch.cuda.current_stream(), torch.cuda.Event(), [None] * len(E["lv"])
            fork.record()
            dbot_a, dva = _decode_bwd(net, E, ctx.Da, dla, dma, True, grads, Gd, order=("record", evs))
            done_a = torch.cuda.Event()
            done_a.record()
            # (a backward pass ACCUMULATED onto an earlier one has put A's gradients into scratch views that B adds into:
            # then B waits for all of A)
            side.wait_event(done_a if grads.second else fork)
            with torch.cuda.stream(side):
                dbot_b, dvb = _decode_bwd(net, E, ctx.Db, dlb, dmb, True, grads, Gd, order=("wait", evs))
                done_b = torch.cuda.Event()
                done_b.record()
                side.wait_event(done_a)
                grads.fold_second()
            if not getattr(net, "encoder_bwd_on_side", _ENC_SIDE):
                main.wait_stream(side)
                for t in [dbot_b, *dvb]:
                    if t is not None:
                        t.record_stream(main)
                _encode_bwd(net, E, dbot_a.add_(dbot_b), Gd, grads)
                ctx.E = ctx.Da = ctx.Db = None
                return (None, None, None, *dva, *dvb, *grads.finish(net.param_list(), "sound"))
            with torch.cuda.stream(side):
                # The encoder's backward continues HERE, on the side stream, and this stream returns to autograd with the
                # gradients of the visual inputs: the visual trunk's nodes start while the encoder's gradients are computed
                # (they need nothing from it).  Its parameter gradients are covered by the node's event (FlatSGD.node_finished,
                # recorded on the side stream), which the optimizer step and the all-reduces wait for.
                _encode_bwd(net, E, dbot_b.add_(dbot_a), Gd, grads)
                pg = grads.finish(net.param_list(), "sound")
            main.wait_event(done_b)
            for t in dvb:
                if t is not None:
                    t.record_stream(main)
            if any(g is not None for g in pg):
                main.wait_stream(side)               # a gradient handed back to autograd: it must be complete on this stream
            # what the side stream is still reading stays alive until the streams have met again: at the end of this backward
            # pass (FlatSGD._end_of_backward), or at the latest before the next forward
            object.__setattr__(net, "_pair_deferred", (E, ctx.Da, ctx.Db, Gd, dbot_a, dbot_b))
            grads.sink.at_end_of_backward(lambda: _release_deferred(net))
            ctx.E = ctx.Da = ctx.Db = None
            return (None, None, None, *dva, *dvb, *pg)
        _encode_bwd(net, E, dbot_a.add_(dbot_b), Gd, grads)
        ctx.E = ctx.Da = ctx.Db = None
        return (None, None, None, *dva, *dvb, *grads.finish(net.param_list(), "sound"))
