"""Oracle networks: audio U-Net, bottleneck fusion, visual encoder, synthesizer.

Plain PyTorch (CPU, fp32) restatement; TEST INFRASTRUCTURE ONLY (see package
docstring).  Module/parameter names reproduce the reference's ``state_dict``
keys so the same weights load into the reference, the oracle and the HIP path.
"""
import itertools
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class _Slots(nn.Module):
    """Numbered child container used only to reproduce nn.Sequential key names."""

    def put(self, idx, mod):
        self.add_module(str(idx), mod)
        return mod

    def at(self, idx):
        return getattr(self, str(idx))


# ----------------------------------------------------------------------------
# Bottleneck fusion  (reference: models/fusion_net.py)
# ----------------------------------------------------------------------------
def _attend(kind, a, v, scale_dim):
    """a: [..., Dc] audio vectors broadcastable against v: [..., Dc, H, W].

    cos -> F.cosine_similarity (fusion_net.py:27-29); sig -> sigmoid of the
    scaled dot product (fusion_net.py:31-32, divisor sqrt(shape[scale_dim])).
    """
    a = a[..., None, None]
    if kind == "cos":
        return F.cosine_similarity(a, v, dim=-3)
    if kind == "sig":
        return torch.sigmoid(torch.sum(a * v / math.sqrt(scale_dim), dim=-3))
    raise ValueError(kind)


def ao_permute_n(x, draws, C):
    """BUILD-DEFINED generalisation of the audio-only swap to C > 2 sources (no reference counterpart: the reference
    hard-codes two blocks, fusion_net.py:93-104).  draws: long[B], index of a permutation of the C audio blocks in
    itertools order; slot c receives block perm[c]; channels beyond C*(D//C) of the tile are zero (remainder rule of
    `Fusion._coloc_n`)."""
    B, D, Fq, T = x.shape
    Dc = D // C
    g = torch.amax(x, dim=(2, 3))[:, :C * Dc].view(B, C, Dc)
    table = torch.tensor(list(itertools.permutations(range(C))))
    sel = torch.gather(g, 1, table[draws.long()][:, :, None].expand(B, C, Dc))
    tiles = torch.cat([sel.reshape(B, C * Dc), x.new_zeros(B, D - C * Dc)], 1)
    return torch.cat([tiles.reshape(B, D, 1, 1).expand(B, D, Fq, T), x], 1)


def ao_swap(x, draws):
    """fusion_net.py:93-104 (identical in CoLoc/CoLoc_Sel/MixVis).

    draws: bool[B] = (torch.rand(B) > 0.5).  one_hot(draw) is [1,0] for 0 and
    [0,1] for 1, and is used as a *gather index*: draw 0 -> (block1, block0),
    draw 1 -> (block0, block1).  If every draw is 0 the one_hot is [B,1] wide
    and gather then fills only slot 0 with index 1... the reference then
    broadcasts a [B,1,...] tensor, so both slots receive block 1.
    """
    B, D, Fq, T = x.shape
    g = torch.amax(x, dim=(2, 3)).view(B, 2, D // 2)
    draws = draws.to(torch.long)
    if int(draws.max()) == 0:
        sel = g[:, 1:2].expand(B, 2, D // 2)
    else:
        first = torch.where(draws.bool()[:, None], g[:, 0], g[:, 1])
        second = torch.where(draws.bool()[:, None], g[:, 1], g[:, 0])
        sel = torch.stack([first, second], 1)
    tiles = sel.reshape(B, D, 1, 1).expand(B, D, Fq, T)
    return torch.cat([tiles, x], 1)


class Fusion(nn.Module):
    """hidsep (=CoLoc), CoLoc_Sel and MixVis (fusion_net.py:20-311). Parameter free."""

    def __init__(self, fusion_type="hidsep", att_type="cos"):
        super().__init__()
        if fusion_type not in ("hidsep", "CoLoc_Sel", "MixVis"):
            raise AssertionError(fusion_type)  # fusion_net.py:17-18
        self.fusion_type = fusion_type
        self.att_type = att_type
        self.ao_draws = None  # tests may pin the AO random draw

    num_src = 2     # audio blocks of the audio-only path (the AV path reads it off len(v_ls))

    def forward(self, x, v_ls):
        if v_ls is None:
            B = x.shape[0]
            if self.num_src > 2:
                draws = self.ao_draws if self.ao_draws is not None else \
                    torch.randint(0, math.factorial(self.num_src), (B,))
                return ao_permute_n(x, draws, self.num_src), (None, None)
            draws = self.ao_draws if self.ao_draws is not None else (torch.rand(B) > 0.5)
            return ao_swap(x, draws), (None, None)
        if self.fusion_type == "MixVis":
            return self._mixvis(x, v_ls)
        if len(v_ls) > 2:
            if self.fusion_type != "hidsep":
                raise NotImplementedError("only CoLoc (hidsep) is generalised beyond two sources")
            return self._coloc_n(x, v_ls)
        return self._coloc(x, v_ls, select=(self.fusion_type == "CoLoc_Sel"))

    def _coloc_n(self, x, v_ls):
        """BUILD-DEFINED generalisation of CoLoc (fusion_net.py:35-72) to C = len(v_ls) > 2 sources — BASELINE.json
        configs[4]; the reference hard-codes C = P = 2 (`x_t = stack((x_p1, x_p2))`) and has no counterpart.  Rules:
          * Dc = D // C channels per audio block; the blocks are the FIRST C*Dc pooled channels, the D - C*Dc remainder
            channels take no part in the matching and their tile channels are zero, so the fused tensor keeps the
            reference's 2*D channels (the U-Net's parameter shapes do not depend on C);
          * all C! permutations in itertools order: maps[b,p,c] = att(a[perm_p[c]], v_c); scores[b,p] = sum_c max maps;
          * best = FIRST maximum (torch.sort on two entries gives the same winner absent ties);
            match_loss = mean_b(-score_best + sum of the other scores); att_maps = maps[b, best];
          * f_c = max_{h,w}(v_c * att_maps_c), tiled.
        With C = 2 this is `_coloc(select=False)` (asserted by tests/test_oracle_golden.py against the reference goldens)."""
        B, D, Fq, T = x.shape
        C = len(v_ls)
        Dc = D // C
        a = torch.amax(x, dim=(2, 3))[:, :C * Dc].view(B, C, Dc)
        table = torch.tensor(list(itertools.permutations(range(C))))            # [P,C]
        perms = a[:, table]                                                       # [B,P,C,Dc]
        v = torch.stack(v_ls, 1)                                                  # [B,C,Dc,H,W]
        maps = _attend(self.att_type, perms, v[:, None], Dc)                      # [B,P,C,H,W]
        scores = torch.amax(maps, dim=(3, 4)).sum(-1)                             # [B,P]
        best = scores.argmax(1)                                                   # first maximum
        srt_best = scores.gather(1, best[:, None])[:, 0]
        match_loss = (-srt_best + (scores.sum(1) - srt_best)).mean(0)
        att = maps[torch.arange(B), best]                                         # [B,C,H,W]
        f = torch.amax(v * att[:, :, None], dim=(3, 4))                           # [B,C,Dc]
        tiles = torch.cat([f.reshape(B, C * Dc), x.new_zeros(B, D - C * Dc)], 1)
        return torch.cat([tiles.reshape(B, D, 1, 1).expand(B, D, Fq, T), x], 1), (match_loss, att)

    def _coloc(self, x, v_ls, select):
        # fusion_net.py:35-72 (CoLoc) / 127-190 (CoLoc_Sel); C = P = 2.
        B, D, Fq, T = x.shape
        Dc = D // 2
        a = torch.amax(x, dim=(2, 3)).view(B, 2, Dc)            # audio blocks
        perms = torch.stack([a, a.flip(1)], 1)                   # [B,P,C,Dc]
        v = torch.stack(v_ls, 1)                                 # [B,C,Dc,H,W]
        # the reference divides by sqrt(x_t.shape[3]) = sqrt(Dc)
        maps = _attend(self.att_type, perms, v[:, None], Dc)     # [B,P,C,H,W]
        per_c = torch.amax(maps, dim=(3, 4))                     # [B,P,C]
        scores = per_c.sum(-1)                                   # [B,P]
        srt, idx = torch.sort(scores, dim=1, descending=True)
        match_loss = (-srt[:, 0] + srt[:, 1:].sum(-1)).mean(0)
        best = idx[:, 0]
        att = maps[torch.arange(B), best]                        # [B,C,H,W]
        if not select:
            f = torch.amax(v * att[:, :, None], dim=(3, 4))      # [B,C,Dc]
        else:
            H, W = att.shape[-2:]
            where = att.reshape(B, 2, H * W).argmax(-1)          # [B,C]
            vf = v.reshape(B, 2, Dc, H * W)
            f = torch.gather(vf, 3, where[:, :, None, None].expand(B, 2, Dc, 1))[..., 0]
        tiles = f.reshape(B, D, 1, 1).expand(B, D, Fq, T)
        return torch.cat([tiles, x], 1), (match_loss, att)

    def _mixvis(self, x, v_ls):
        # fusion_net.py:248-285
        assert len(v_ls) == 1
        v = v_ls[0]                                              # [B,Dc,H,W2]
        B, D, Fq, T = x.shape
        Dc = D // 2
        a = torch.amax(x, dim=(2, 3)).view(B, 2, Dc)
        # here the reference divides by sqrt(x.shape[2]) with x = [B,C,Dc,1,1] -> sqrt(Dc)
        maps = _attend(self.att_type, a, v[:, None], Dc)         # [B,C,H,W2]
        flat = maps.reshape(B, 2, -1)
        size = flat.shape[-1]
        where = flat.argmax(-1)
        vf = v.reshape(B, 1, Dc, size).expand(B, 2, Dc, size)
        sel = torch.gather(vf, 3, where[:, :, None, None].expand(B, 2, Dc, 1))[..., 0]
        neg_peaks = -flat.amax(-1)                               # [B,C]
        match_loss = neg_peaks.sum(-1).mean().reshape(1) \
            + flat.sum(-1).sum(-1).mean(-1).reshape(1) / size
        match_loss = match_loss + F.cosine_similarity(sel[:, 0], sel[:, 1], dim=1).mean().reshape(1)
        tiles = sel.reshape(B, D, 1, 1).expand(B, D, Fq, T)
        return torch.cat([tiles, x], 1), (match_loss, maps)


# ----------------------------------------------------------------------------
# Audio U-Net  (reference: models/audio_net.py:10-203)
# ----------------------------------------------------------------------------
class _Level(nn.Module):
    """One nesting level.  Holds parameters under the reference's key names:
    down_forward.{0|1}=conv, down_forward.2=BN, up_forward.2=conv, up_forward.3=BN."""

    def __init__(self, outer_nc, inner_nc, in_nc, up_in_nc, kind, child, fusion=None):
        super().__init__()
        self.kind = kind  # 'outer' | 'mid' | 'inner'
        # registration order follows the reference (down_forward, mid_forward, up_forward[, fusion])
        self.down_forward = _Slots()
        if child is not None:
            self.mid_forward = child
        self.up_forward = _Slots()
        conv_idx = 0 if kind == "outer" else 1
        self.down_conv_idx, self.up_conv_idx = conv_idx, 2
        self.down_forward.put(conv_idx, nn.Conv2d(in_nc, inner_nc, 4, 2, 1, bias=False))
        if kind == "mid":
            self.down_forward.put(2, nn.BatchNorm2d(inner_nc))
        self.up_forward.put(2, nn.Conv2d(up_in_nc, outer_nc, 3, 1, 1, bias=(kind == "outer")))
        if kind != "outer":
            self.up_forward.put(3, nn.BatchNorm2d(outer_nc))
        if fusion is not None:
            self.fusion = fusion

    @property
    def down_conv(self):
        return self.down_forward.at(self.down_conv_idx)

    @property
    def down_bn(self):
        return self.down_forward.at(2) if self.kind == "mid" else None

    @property
    def up_conv(self):
        return self.up_forward.at(2)

    @property
    def up_bn(self):
        return self.up_forward.at(3) if self.kind != "outer" else None


class Unet(nn.Module):
    def __init__(self, fc_dim=64, num_downs=5, ngf=64, fusion_type="hidsep", att_type="cos",
                 extra_size=None):
        super().__init__()
        # innermost first, as the reference builds it (audio_net.py:19-35)
        self.extra_size = extra_size
        if extra_size is None:
            fusion = Fusion(fusion_type, att_type)
            fusion.num_src = fc_dim if 2 < fc_dim <= 4 else 2      # one output channel per source (num_channels == num_mix)
            lvl = _Level(ngf * 8, ngf * 8, ngf * 8, ngf * 16, "inner", None, fusion)
        else:  # SoP++ variant (SoP++/audio_net.py:151-198): no fusion, wider bottleneck conv
            lvl = _Level(ngf * 8, ngf * 8 + 2 * extra_size, ngf * 8, ngf * 8, "inner", None, None)
        for _ in range(num_downs - 5):
            lvl = _Level(ngf * 8, ngf * 8, ngf * 8, ngf * 16, "mid", lvl)
        lvl = _Level(ngf * 4, ngf * 8, ngf * 4, ngf * 16, "mid", lvl)
        lvl = _Level(ngf * 2, ngf * 4, ngf * 2, ngf * 8, "mid", lvl)
        lvl = _Level(ngf, ngf * 2, ngf, ngf * 4, "mid", lvl)
        lvl = _Level(fc_dim, ngf, 1, ngf * 2, "outer", lvl)
        self.bn0 = nn.BatchNorm2d(1)
        self.unet_block = lvl

    def levels(self):
        out, l = [], self.unet_block
        while True:
            out.append(l)
            if l.kind == "inner":
                return out
            l = l.mid_forward

    @staticmethod
    def _up(l, h):
        h = F.interpolate(F.relu(h), scale_factor=2, mode="bilinear", align_corners=True)
        h = l.up_conv(h)
        return l.up_bn(h) if l.up_bn is not None else h

    def forward(self, x, v=None):
        h = self.bn0(x)
        lv = self.levels()
        skips = []
        for l in lv:
            if l.kind != "outer":
                # in-place LeakyReLU aliases the skip tensor (audio_net.py:64,119-122,197-203)
                h = F.leaky_relu(h, 0.2)
                skips.append(h)
            h = l.down_conv(h)
            if l.down_bn is not None:
                h = l.down_bn(h)
        extra = None
        if self.extra_size is None:
            h, meta = lv[-1].fusion(h, v)
        else:
            extra, h = torch.split(h, [2 * self.extra_size, h.shape[1] - 2 * self.extra_size], 1)
            meta = (extra,)
        for l in reversed(lv):
            h = self._up(l, h)
            if l.kind != "outer":
                h = torch.cat([skips.pop(), h], 1)
        return h, meta


# ----------------------------------------------------------------------------
# Visual encoder  (reference: models/vision_net.py; torchvision resnet18 restated)
# ----------------------------------------------------------------------------
class BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=False)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(
                nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return self.relu(y + idt)


def resnet18_trunk():
    """Children 0..7 of torchvision.models.resnet18 (conv1,bn1,relu,maxpool,layer1..4);
    the standard published architecture (He et al. 2016), restated — torchvision is not
    installed here, parity at this boundary is unpinned."""
    def layer(cin, cout, stride):
        return nn.Sequential(BasicBlock(cin, cout, stride), BasicBlock(cout, cout, 1))
    return nn.Sequential(
        nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=False),
        nn.MaxPool2d(3, 2, 1),
        layer(64, 64, 1), layer(64, 128, 2), layer(128, 256, 2), layer(256, 512, 2))


def _dilate(mod, dilate):
    # vision_net.py:96-109
    for m in mod.modules():
        if isinstance(m, nn.Conv2d):
            if m.stride == (2, 2):
                m.stride = (1, 1)
                if m.kernel_size == (3, 3):
                    m.dilation = (dilate // 2, dilate // 2)
                    m.padding = (dilate // 2, dilate // 2)
            elif m.kernel_size == (3, 3):
                m.dilation = (dilate, dilate)
                m.padding = (dilate, dilate)


class VisualNet(nn.Module):
    """ResnetDilated (vision_net.py:71-147) when dilate_scale in (8,16); ResnetFC (:20-68) when None."""

    def __init__(self, fc_dim=64, pool_type="maxpool", dilate_scale=16, conv_size=3):
        super().__init__()
        self.pool_type = pool_type
        self.features = resnet18_trunk()
        if dilate_scale == 8:
            _dilate(self.features[6], 2)
            _dilate(self.features[7], 4)
        elif dilate_scale == 16:
            _dilate(self.features[7], 2)
        self.fc = nn.Conv2d(512, fc_dim, conv_size, padding=conv_size // 2)

    def forward(self, x, pool=True):
        x = self.fc(self.features(x))
        if not pool:
            return x
        x = F.adaptive_avg_pool2d(x, 1) if self.pool_type == "avgpool" else F.adaptive_max_pool2d(x, 1)
        return x.view(x.size(0), x.size(1))

    def forward_multiframe(self, x, pool=True):
        B, C, T, H, W = x.shape
        y = self.fc(self.features(x.permute(0, 2, 1, 3, 4).reshape(B * T, C, H, W)))
        _, C2, h, w = y.shape
        y = y.view(B, T, C2, h, w).permute(0, 2, 1, 3, 4)
        if not pool:
            return y.mean(2)
        if self.pool_type == "avgpool":
            return y.mean(dim=(2, 3, 4))
        return y.amax(dim=(2, 3, 4))


# ----------------------------------------------------------------------------
# Synthesizer  (reference: models/synthesizer_net.py)
# ----------------------------------------------------------------------------
class InnerProd(nn.Module):
    def __init__(self, fc_dim):
        super().__init__()
        self.scale = nn.Parameter(torch.ones(fc_dim))
        self.bias = nn.Parameter(torch.zeros(1))

    def _w(self, f):
        return f * self.scale

    def forward(self, feat_img, feat_sound):          # synthesizer_net.py:12-19
        B, C = feat_sound.shape[:2]
        z = torch.einsum("bc,bcn->bn", self._w(feat_img.view(B, C)), feat_sound.reshape(B, C, -1))
        return z.view(B, 1, *feat_sound.shape[2:]) + self.bias

    def forward_nosum(self, feat_img, feat_sound):    # :21-26
        B, C = feat_sound.shape[:2]
        return self._w(feat_img.view(B, C)).view(B, C, 1, 1) * feat_sound + self.bias

    def forward_pixelwise(self, feats_img, feat_sound):  # :29-38
        B, C, HI, WI = feats_img.shape
        _, _, HS, WS = feat_sound.shape
        fi = self._w(feats_img.view(B, C, HI * WI).transpose(1, 2))
        z = torch.bmm(fi, feat_sound.view(B, C, HS * WS)).view(B, HI, WI, HS, WS)
        return z + self.bias


class Bias(InnerProd):
    def __init__(self):
        nn.Module.__init__(self)
        self.bias = nn.Parameter(torch.zeros(1))

    def _w(self, f):
        return f


# ----------------------------------------------------------------------------
# Builder helpers (reference: models/__init__.py:16-92)
# ----------------------------------------------------------------------------
def activate(x, activation):
    if activation == "sigmoid":
        return torch.sigmoid(x)
    if activation == "softmax":
        return F.softmax(x, dim=1)
    if activation == "relu":
        return F.relu(x)
    if activation == "tanh":
        return torch.tanh(x)
    if activation == "no":
        return x
    raise Exception("Unkown activation!")


def weights_init(m):
    name = m.__class__.__name__
    if name.find("Conv") != -1:
        m.weight.data.normal_(0.0, 0.001)
    elif name.find("BatchNorm") != -1:
        m.weight.data.normal_(1.0, 0.02)
        m.bias.data.fill_(0)
    elif name.find("Linear") != -1:
        m.weight.data.normal_(0.0, 0.0001)


def wide_init(net, gen, gain=1.0):
    """Kaiming-scaled weights so logits span several units (SURVEY.md §4 caveat:
    the reference init gives masks ~0.5 and makes parity vacuous)."""
    for m in net.modules():
        if isinstance(m, nn.Conv2d):
            fan_in = m.in_channels * m.kernel_size[0] * m.kernel_size[1]
            m.weight.data = torch.randn(m.weight.shape, generator=gen) * (gain * math.sqrt(2.0 / fan_in))
            if m.bias is not None:
                m.bias.data = torch.randn(m.bias.shape, generator=gen) * 0.1
        elif isinstance(m, nn.BatchNorm2d):
            m.weight.data = 1.0 + 0.2 * torch.randn(m.weight.shape, generator=gen)
            m.bias.data = 0.1 * torch.randn(m.bias.shape, generator=gen)


def build_sound(arch="unet5", fc_dim=64, fusion_type="hidsep", att_type="cos", extra_size=None):
    downs = {"unet5": 5, "unet6": 6, "unet7": 7}
    if arch not in downs:
        raise Exception("Architecture undefined!")
    net = Unet(fc_dim=fc_dim, num_downs=downs[arch], fusion_type=fusion_type, att_type=att_type,
               extra_size=extra_size)
    net.apply(weights_init)
    return net


def build_frame(arch="resnet18dilated", fc_dim=64, pool_type="avgpool"):
    if arch == "resnet18fc":
        return VisualNet(fc_dim, pool_type, dilate_scale=None)
    if arch == "resnet18dilated":
        return VisualNet(fc_dim, pool_type, dilate_scale=16)
    raise Exception("Architecture undefined!")
