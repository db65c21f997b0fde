"""Tensor-level wrappers over the C ABI (one Python function per entry point of include/avsep.h).

Only shape bookkeeping and output allocation happen here; all arithmetic is in
libavsep_gfx950.so.  Tensors are dense fp32 NCHW on the current cuda device.
"""
import ctypes as C
import os

import torch

from . import lib
from .lib import ConvDesc, CatDesc, call, ptr


def _f32(shape, like):
    return torch.empty(shape, dtype=torch.float32, device=like.device)


# ---- storage formats (include/avsep.h AVSEP_FMT_*) ------------------------------------------------------------------------
# F32: dense fp32 [N,C,H,W].  B16: torch.bfloat16 of shape [N, C/16, H, W, 16] — the channel-blocked image the bf16 kernels
# stage with one 16-byte load per (position, 8-channel half).  A tensor's dtype IS its format tag.
FMT_F32, FMT_B16 = 0, 1


def is_b16(t):
    return t is not None and t.dtype == torch.bfloat16


def fmt_of(t):
    return FMT_B16 if is_b16(t) else FMT_F32


def dims(t):
    """(N, C, H, W) of an activation tensor in either format."""
    if is_b16(t):
        N, CB, H, W, _ = t.shape
        return N, CB * 16, H, W
    return tuple(t.shape)


def channels(t):
    return t.shape[1] * 16 if is_b16(t) else t.shape[1]


def per_channel(t):
    """Elements per channel (the BatchNorm count N*H*W)."""
    return t.numel() // channels(t)


def _b16(shape_nchw, like):
    N, Cc, H, W = shape_nchw
    return torch.empty((N, Cc // 16, H, W, 16), dtype=torch.bfloat16, device=like.device)


def empty_as(t):
    return torch.empty_like(t)


def _twin_hit(t):
    """The remembered other-format twin of `t` if it is current.  A twin built on another stream (two passes that share a
    tensor run on forked streams) is waited for, and its memory is kept until this stream is done with it."""
    twin = getattr(t, "_avsep_twin", None)
    if twin is None or twin[1] != t._version:
        return None
    if twin[2] is not None:
        cur = torch.cuda.current_stream(t.device)
        if cur != twin[3]:
            cur.wait_event(twin[2])
            twin[0].record_stream(cur)
    return twin[0]


def _twin_set(t, out):
    ev = st = None
    if t.is_cuda:
        ev, st = torch.cuda.Event(), torch.cuda.current_stream(t.device)
        ev.record()
    t._avsep_twin = (out, t._version, ev, st)
    out._avsep_twin = (t, out._version, ev, st)


def to_b16(t):
    """B16 image of an fp32 NCHW tensor (one HBM pass).  The converted twin is remembered on the tensor object, so an operand
    that several kernels of a step need in the other format (the forward and the weight gradient of a conv) is converted once."""
    if t is None or is_b16(t):
        return t
    twin = _twin_hit(t)
    if twin is not None:
        return twin
    N, Cc, H, W = t.shape
    out = _b16((N, Cc, H, W), t)
    call("avsep_f32_to_b16", ptr(t), N, Cc, H * W, ptr(out))
    _twin_set(t, out)
    return out


def to_f32(t):
    if t is None or not is_b16(t):
        return t
    twin = _twin_hit(t)
    if twin is not None:
        return twin
    N, Cc, H, W = dims(t)
    out = _f32((N, Cc, H, W), t)
    call("avsep_b16_to_f32", ptr(t), N, Cc, H * W, ptr(out))
    _twin_set(t, out)
    return out


def as_fmt(t, fmt):
    return to_b16(t) if fmt == FMT_B16 else to_f32(t)


def b16_ok(c):
    """May a [N,c,H,W] tensor be kept as a B16 image?"""
    return c % 16 == 0


# Activations between the bf16 kernels are kept as B16 images when the arithmetic mode is bf16 (set_precision).  The module
# attribute exists for A/B measurements from a script (False: fp32 NCHW tensors, converted at every bf16 kernel's door).
b16_activations = True


def want_b16(c=16):
    return _precision == 1 and b16_activations and b16_ok(c)


# bf16 mode: weight gradients of the <= 8x8 maps (the deep U-Net levels) run over ONE grid image of the whole batch instead of
# one mostly empty 256-pixel chunk per image (csrc/b16.hip: avsep_b16_grid_pack)
grid_small_maps = os.environ.get("AVSEP_GRID_SMALL_MAPS", "1") != "0"


def grid_pack(x, n, c, h, w, gx, gy, py, px, sc=None, sh=None, act=0):
    """B16 grid image [1, c/16, gy*py, gx*px, 16] of the n small maps of x (fp32 NCHW or B16), act(affine(.)) applied, zeros
    everywhere else."""
    out = _b16((1, c, gy * py, gx * px), x)
    call("avsep_b16_grid_pack", ptr(x), fmt_of(x), n, c, h, w, gx, py, px, gy * py, gx * px, ptr(sc), ptr(sh), act, ptr(out))
    return out


# Operand precision of the convolutions: "f32" (exact f32 MFMA, the reference's arithmetic) or "bf16" (operands rounded
# to bf16 while they are staged, fp32 accumulation / BatchNorm statistics / outputs / master weights: BASELINE.json
# configs[2]).  Geometries without a bf16 kernel run in f32 either way.
PREC_BY_NAME = {"f32": 0, "fp32": 0, "bf16": 1}
_precision = PREC_BY_NAME[os.environ.get("AVSEP_PRECISION", "f32")]


def set_precision(name):
    global _precision
    _precision = PREC_BY_NAME[name]


def get_precision():
    return "bf16" if _precision else "f32"


# Kernel families the convolutions must NOT use (avsep_conv_desc.algo, a bit set of lib.ALGO_NO values): A/B runs and the
# gradient-error attribution (tools/grad_attribution.py).  The library itself reads no environment variable; this host
# layer takes the initial mask from AVSEP_ALGO_NO="winograd,flat,..." (or the older per-family AVSEP_NO_<FAMILY>=1 names) once,
# at import, and every descriptor built afterwards carries the module's current mask.
def _algo_from_env():
    mask = 0
    for name in os.environ.get("AVSEP_ALGO_NO", "").replace(" ", "").split(","):
        if name:
            mask |= lib.ALGO_NO[name.lower()]
    for name, bit in lib.ALGO_NO.items():
        if os.environ.get("AVSEP_NO_" + name.upper()) is not None:
            mask |= bit
    return mask


algo_mask = _algo_from_env()
conv_tune = 0      # avsep_conv_desc.tune (tools/conv_bench.py only)


def set_algo_mask(*names):
    """set_algo_mask("winograd", ...) — forbid those kernel families from now on; set_algo_mask() clears the mask."""
    global algo_mask
    algo_mask = 0
    for n in names:
        algo_mask |= lib.ALGO_NO[n]
    return algo_mask


# Batch the launch heuristics are planned for, as a multiple of the real batch (avsep_conv_desc.plan_n): the parity tests
# set 8 so that a batch-8 step takes, layer by layer, the kernel instantiations of the batch-64 step bench.py times.
plan_batch_scale = 1


_pack_cache = None


class pack_scope:
    """`with pack_scope():` — packed weight images are cached for the duration (Conv.pack).  train_step_async wraps
    forward + backward, the optimizer step comes after; an in-place update of a weight inside a scope changes the tensor's
    version counter, which is part of the cache key, so the image is rebuilt instead of going stale."""

    def __enter__(self):
        global _pack_cache
        self.prev, _pack_cache = _pack_cache, ({} if _pack_cache is None else _pack_cache)
        return self

    def __exit__(self, *exc):
        global _pack_cache
        _pack_cache = self.prev
        return False


def out_size(h, k, s, p, d):
    return (h + 2 * p - d * (k - 1) - 1) // s + 1


class Conv:
    """Geometry + virtual-input description of one convolution call (avsep_conv_desc).  Operands may be fp32 NCHW or B16
    images; each call asks the library which format its kernel stages (avsep_conv_io_formats) and converts at the door
    if the tensor it was given is in the other one."""

    def __init__(self, x0, cout, k, stride, pad, dil=1, x1=None, sc0=None, sh0=None, act0=0,
                 sc1=None, sh1=None, act1=0, up2x=False, prec=None):
        lib.require_gpu(x0)
        if x1 is not None:                       # two-source descriptors exist for the fp32 kernels only
            x0, x1 = to_f32(x0), to_f32(x1)
        N, C0, Hs, Ws = dims(x0)
        C1 = x1.shape[1] if x1 is not None else 0
        H, W = (2 * Hs, 2 * Ws) if up2x else (Hs, Ws)
        kh, kw = (k, k) if isinstance(k, int) else k
        self.N, self.Cin, self.H, self.W, self.Cout = N, C0 + C1, H, W, cout
        self.KH, self.KW = kh, kw
        self.Ho, self.Wo = out_size(H, kh, stride, pad, dil), out_size(W, kw, stride, pad, dil)
        self.keep = [x0, x1, sc0, sh0, sc1, sh1]
        d = ConvDesc()
        d.N, d.Cin, d.H, d.W, d.Cout, d.Ho, d.Wo = N, C0 + C1, H, W, cout, self.Ho, self.Wo
        d.KH, d.KW, d.stride, d.pad, d.dil = kh, kw, stride, pad, dil
        d.C0, d.act0, d.act1, d.up2x = C0, act0, act1, int(up2x)
        d.prec = _precision if prec is None else PREC_BY_NAME[prec]
        d.plan_n = N * plan_batch_scale if plan_batch_scale != 1 else 0
        d.algo, d.tune = algo_mask, conv_tune
        d.x0, d.x1 = ptr(x0), ptr(x1)
        d.xfmt = fmt_of(x0)
        d.scale0, d.shift0, d.scale1, d.shift1 = ptr(sc0), ptr(sh0), ptr(sc1), ptr(sh1)
        self.d = d
        self.ref = C.byref(d)
        self.like = x0
        self._grid_xg = None

    def io_formats(self, mode):
        """(format the inputs of this call must have, may the output be B16) — mode 0 fwd, 1 dgrad, 2 wgrad."""
        a, b = C.c_int32(0), C.c_int32(0)
        rc = lib.load().avsep_conv_io_formats(self.ref, mode, C.byref(a), C.byref(b))
        if rc != 0:
            raise lib.AvsepError(f"avsep_conv_io_formats failed ({rc})")
        return a.value, bool(b.value)

    def _x_as(self, fmt):
        """Hand the kernel x0 in the format it stages."""
        if self.d.xfmt != fmt:
            x = as_fmt(self.keep[0], fmt)
            self.keep.append(x)
            self.d.x0, self.d.xfmt = ptr(x), fmt

    def _out(self, shape_nchw, want, allowed):
        b = bool(want) and allowed and b16_ok(shape_nchw[1])
        return (_b16(shape_nchw, self.like) if b else _f32(shape_nchw, self.like)), (FMT_B16 if b else FMT_F32)

    def pack(self, w, mode):
        """Operand image of `w` for this call (mode 0 forward, 1 data gradient).  Inside a `pack_scope()` — one train step,
        during which the weights do not change — the image of a (weight, geometry, mode) is built once and shared: the
        second decoder pass of an AV step and the visual trunk's second source used to repack every weight (56 of the
        114 pack launches of a step)."""
        g = self._grid_geometry(0) if mode == 0 else None
        d = self._grid_desc(g) if g is not None else self.d      # a grid forward multiplies by the N = 1 conv's weight image
        ref = C.byref(d)
        key = None
        if _pack_cache is not None:
            key = (w.data_ptr(), w._version, mode, d.N, d.Cin, d.H, d.W, d.Cout, d.KH, d.KW, d.stride, d.pad, d.dil, d.C0, d.up2x, d.prec,
                   d.plan_n, bool(d.scale0), bool(d.scale1), d.act0, d.act1, d.algo, d.tune)
            hit = _pack_cache.get(key)
            if hit is not None:
                if hit[2] is not None and torch.cuda.current_stream() != hit[3]:
                    cur = torch.cuda.current_stream()
                    cur.wait_event(hit[2])           # packed on another stream (fork_streams): order behind it ...
                    hit[0].record_stream(cur)        # ... and keep its memory from being reused before this stream is done
                return hit[0]
        n = lib.load().avsep_conv_packed_floats(ref, mode)
        out = _f32((n,), w)
        call("avsep_conv_pack_weights", ref, ptr(w), ptr(out), mode)
        if key is not None:
            ev = None
            if out.is_cuda:                  # streams may share the cache (fork_streams, and the backward of such passes):
                ev = torch.cuda.Event()      # remember where and when the image was built
                ev.record()
            # holding `w` keeps a temporary weight tensor's address from being reused
            _pack_cache[key] = (out, w, ev, torch.cuda.current_stream() if ev is not None else None)
        return out

    def _grid_geometry(self, mode=2):
        """(images per grid row, grid rows, input pitch, output pitch) when this call runs over a grid image of the batch
        (bf16 mode, one source, maps of at most 8x8, avsep_b16_grid_pack): the weight gradient (mode 2) of 3x3 / pad 1 and
        4x4 / stride 2 / pad 1, the forward (mode 0) of 4x4 / stride 2 / pad 1.  None otherwise."""
        if mode not in (0, 2):
            return None
        key = (mode, grid_small_maps, b16_activations)
        memo = self.__dict__.setdefault("_grid_memo", {})
        hit = memo.get(key, 0)
        if hit != 0:
            return hit
        memo[key] = g = self._grid_geometry_of(mode)
        return g

    def _grid_geometry_of(self, mode):
        d = self.d
        if not (grid_small_maps and d.prec == 1 and b16_activations and self.keep[1] is None and not d.up2x and d.dil == 1 and
                self.N >= 2 and b16_ok(self.Cin) and b16_ok(self.Cout) and max(self.H, self.W) <= 8):
            return None
        geo = (self.KH, self.KW, d.stride, d.pad)
        if geo == (3, 3, 1, 1) and mode == 2:
            pin = pout = (self.H + 1, self.W + 1)
        elif geo == (4, 4, 2, 1) and self.H % 2 == 0 and self.W % 2 == 0:
            pin, pout = (self.H + 2, self.W + 2), (self.H // 2 + 1, self.W // 2 + 1)
        else:
            return None
        gx = 1
        while gx * gx < self.N:
            gx += 1
        g = (gx, (self.N + gx - 1) // gx, pin, pout)
        # only where the grid image is large enough for the bf16 kernel of that mode (two 2x2 maps are not)
        name = lib.load().avsep_conv_kernel_name(C.byref(self._grid_desc(g)), mode, 0).decode()
        return g if name == ("wgradb_kernel" if mode == 2 else "convbf_kernel") else None

    def _grid_x(self, g):
        """The grid image of this call's activated input, shared by its forward and its weight gradient."""
        x0, _, sc0, sh0 = self.keep[:4]
        hit = self._grid_xg
        if hit is not None and hit[0] is x0 and hit[1] == x0._version:
            return hit[2]
        gx, gy, (pyi, pxi), _ = g
        xg = grid_pack(x0, self.N, self.Cin, self.H, self.W, gx, gy, pyi, pxi, sc0, sh0, self.d.act0)
        self._grid_xg = (x0, x0._version, xg)
        return xg

    def _grid_inner(self, g):
        gx, gy, _, (pyo, pxo) = g
        d = self.d
        inner = Conv(self._grid_x(g), self.Cout, (self.KH, self.KW), d.stride, d.pad, d.dil, prec="bf16")
        if (inner.Ho, inner.Wo) != (gy * pyo, gx * pxo):
            raise lib.AvsepError("grid image geometry")
        return inner

    def _grid_desc(self, g):
        """Descriptor of the N = 1 convolution over the grid images (geometry only: for the dispatch queries)."""
        gx, gy, (pyi, pxi), (pyo, pxo) = g
        d = ConvDesc.from_buffer_copy(self.d)
        d.N, d.H, d.W, d.Ho, d.Wo, d.plan_n = 1, gy * pyi, gx * pxi, gy * pyo, gx * pxo, 0
        d.scale0 = d.shift0 = None
        d.act0, d.xfmt = 0, FMT_B16
        return d

    def _ws(self, query):
        nbytes = getattr(lib.load(), query)(self.ref)
        ws = torch.empty((nbytes // 4,), dtype=torch.float32, device=self.like.device) if nbytes else None
        return ws, nbytes

    def fwd(self, w_packed, bias=None, stats=None, out_b16=None):
        """`out_b16`: ask for the output as a B16 image (default: whenever the arithmetic mode keeps B16 activations);
        granted when this call's kernel can write one, else the result is fp32 NCHW."""
        g = self._grid_geometry(0)
        if g is not None:                            # small maps: one conv over the grid image of the batch, then the real positions
            gx, gy, _, (pyo, pxo) = g
            yg = self._grid_inner(g).fwd(w_packed, bias, None, out_b16=False)
            y, self.d.yfmt = self._out((self.N, self.Cout, self.Ho, self.Wo), want_b16(self.Cout) if out_b16 is None else out_b16, True)
            call("avsep_grid_unpack", ptr(yg), self.N, self.Cout, self.Ho, self.Wo, gx, pyo, pxo, gy * pyo, gx * pxo, ptr(y),
                 self.d.yfmt, ptr(stats))
            return y
        need, allowed = self.io_formats(0)
        self._x_as(need)
        y, self.d.yfmt = self._out((self.N, self.Cout, self.Ho, self.Wo), want_b16(self.Cout) if out_b16 is None else out_b16,
                                   allowed)
        ws, nbytes = self._ws("avsep_conv2d_fwd_workspace_bytes")
        call("avsep_conv2d_fwd", self.ref, ptr(w_packed), ptr(bias), ptr(y), ptr(stats), ptr(ws), nbytes)
        return y

    def dgrad(self, w_packed_d, dy, out_b16=None):
        need, allowed = self.io_formats(1)
        dy = as_fmt(dy, need)
        self.d.dyfmt = need
        dx, self.d.dxfmt = self._out((self.N, self.Cin, self.H, self.W), want_b16(self.Cin) if out_b16 is None else out_b16,
                                     allowed)
        ws, nbytes = self._ws("avsep_conv2d_dgrad_workspace_bytes")
        call("avsep_conv2d_dgrad", self.ref, ptr(w_packed_d), ptr(dy), ptr(dx), ptr(ws), nbytes)
        return dx

    def dgrad_act_fused(self):
        """True when dgrad_act runs as ONE kernel (the activation gradient in the data-gradient kernel's epilogue)."""
        self.d.dxfmt = FMT_F32
        return bool(lib.load().avsep_conv2d_dgrad_act_fused(self.ref))

    def dgrad_act(self, w_packed_d, dy, y, scale, shift, mean, invstd, act, bstats, residual=None, res_scale=None,
                  res_shift=None, dz2=None, add=None):
        """act'(scale*y + shift [+ res_scale*residual + res_shift]) * (dgrad(dy) [+ dz2]) [+ add] and its BatchNorm-backward
        sums: avsep_conv2d_dgrad_act (fp32 tensors; the data gradient is never written unmasked)."""
        need, _ = self.io_formats(1)
        dy = as_fmt(dy, need)
        self.d.dyfmt = need
        self.d.dxfmt = FMT_F32
        dx = _f32((self.N, self.Cin, self.H, self.W), self.like)
        for t in (y, residual, dz2, add):
            if t is not None and (is_b16(t) or tuple(t.shape) != tuple(dx.shape) or not t.is_contiguous()):
                raise lib.AvsepError("dgrad_act operands must be dense fp32 tensors of dx's shape")
        e = lib.ActBwd(ptr(y), ptr(scale), ptr(shift), ptr(residual), ptr(res_scale), ptr(res_shift), ptr(dz2), ptr(add),
                       ptr(mean), ptr(invstd), ptr(bstats), int(act))
        ws, nbytes = self._ws("avsep_conv2d_dgrad_workspace_bytes")
        call("avsep_conv2d_dgrad_act", self.ref, ptr(w_packed_d), ptr(dy), C.byref(e), ptr(dx), ptr(ws), nbytes)
        return dx

    def kernel_name(self, mode, with_stats=True):
        """Kernel family the library dispatches this call to (mode: "fwd" | "dgrad" | "wgrad")."""
        ref = self.ref
        g = self._grid_geometry({"fwd": 0, "wgrad": 2}.get(mode, 1))
        if g is not None:                            # the call runs on the grid image of the batch
            ref = C.byref(self._grid_desc(g))
        return lib.load().avsep_conv_kernel_name(ref, {"fwd": 0, "dgrad": 1, "wgrad": 2}[mode], int(with_stats)).decode()

    def kernel_variant(self, mode, with_stats=True, plan_n=None):
        """Family + the grid-size dependent launch decisions (tile shape, workgroup size, split-K) of this call; with
        `plan_n` the answer for a descriptor planned for that batch instead."""
        d = self.d
        if plan_n is not None:
            d = ConvDesc.from_buffer_copy(self.d)
            d.plan_n = plan_n
        buf = C.create_string_buffer(128)
        rc = lib.load().avsep_conv_kernel_variant(C.byref(d), {"fwd": 0, "dgrad": 1, "wgrad": 2}[mode], int(with_stats), buf, 128)
        if rc != 0:
            raise lib.AvsepError(f"avsep_conv_kernel_variant failed ({rc})")
        return buf.value.decode()

    def head_applicable(self):
        """True when this (up2x, Cout <= 4) conv takes the fused decoder-head kernels (csrc/head.hip)."""
        return bool(lib.load().avsep_conv2d_head_applicable(self.ref))

    def dgrad_up2x(self, w, dy, mean1=None, invstd1=None, bstats1=None, g0_acc=None):
        """Gradient wrt the two LOW-RES sources of an up2x conv (dgrad + transposed bilinear + ReLU mask fused)."""
        x0, x1 = self.keep[0], self.keep[1]
        g0 = g0_acc if g0_acc is not None else torch.empty_like(x0)
        g1 = torch.empty_like(x1) if x1 is not None else None
        ws, nbytes = self._ws("avsep_conv2d_dgrad_up2x_workspace_bytes")
        call("avsep_conv2d_dgrad_up2x", self.ref, ptr(w), ptr(to_f32(dy)), ptr(g0), ptr(g1), ptr(mean1), ptr(invstd1),
             ptr(bstats1), int(g0_acc is not None), ptr(ws), nbytes)
        return g0, g1

    def wgrad(self, dy, want_bias=False, out=None, out_bias=None):
        """dw (and dbias); `out` / `out_bias`: dense destinations to write into (e.g. a parameter's flat .grad view)."""
        shape = (self.Cout, self.Cin, self.KH, self.KW)
        if out is not None and (tuple(out.shape) != shape or not out.is_contiguous() or out.dtype != torch.float32):
            raise lib.AvsepError("wgrad destination must be a dense fp32 OIHW tensor")
        g = self._grid_geometry(2)
        if g is not None:
            gx, gy, _, (pyo, pxo) = g
            dyg = grid_pack(dy, self.N, self.Cout, self.Ho, self.Wo, gx, gy, pyo, pxo)
            return self._grid_inner(g).wgrad(dyg, want_bias, out, out_bias)
        need, _ = self.io_formats(2)
        self._x_as(need)
        dy = as_fmt(dy, need)
        self.d.dyfmt = need
        dw = out if out is not None else _f32(shape, self.like)
        db = (out_bias if out_bias is not None else _f32((self.Cout,), self.like)) if want_bias else None
        nbytes = lib.load().avsep_conv2d_wgrad_workspace_bytes(self.ref)
        ws = torch.empty((max(nbytes, 4) // 4,), dtype=torch.float32, device=self.like.device)
        call("avsep_conv2d_wgrad", self.ref, ptr(dy), ptr(dw), ptr(db), ptr(ws), nbytes)
        return dw, db


class _StatsArena:
    """Zeroed fp64 statistics buffers (BatchNorm sums, 2 x C doubles each) carved out of ONE device array that a single
    memset clears, instead of one fill launch per buffer (120 per train step).  Every consumer of a buffer (bn_finalize /
    bn_bwd_coeffs) is launched right after its producers on the same stream, so when the bump pointer wraps, the memset
    that re-clears the array is ordered after all of them; a slice handed out is never live across a wrap as long as no
    more than `capacity` doubles are requested between its allocation and its last use (a step uses ~40 K of the 2 M)."""

    def __init__(self, device, capacity=1 << 21):
        self.buf = torch.zeros((capacity,), dtype=torch.float64, device=device)
        self.cap, self.off, self.stream = capacity, 0, torch.cuda.current_stream(device)

    def take(self, n):
        n = (n + 15) // 16 * 16                        # 128-byte granules: atomics of two buffers never share a line
        if n > self.cap // 4 or torch.cuda.current_stream(self.buf.device) != self.stream:
            return None
        if self.off + n > self.cap:
            self.buf.zero_()
            self.off = 0
        out = self.buf[self.off:self.off + n]
        self.off += n
        return out


_arenas = {}


def zeros_stats(c, like):
    """[2*c] zeroed doubles on like.device (see _StatsArena)."""
    dev = like.device
    key = (dev, torch.cuda.current_stream(dev))        # one arena per stream: its re-clearing memset is ordered on that stream
    arena = _arenas.get(key)
    if arena is None:
        arena = _arenas[key] = _StatsArena(dev)
    out = arena.take(2 * c)
    if out is None:
        return torch.zeros((2 * c,), dtype=torch.float64, device=dev)
    return out[:2 * c]


def channel_stats(x, stats):
    x = to_f32(x)
    N, Cc = x.shape[:2]
    call("avsep_channel_stats", ptr(x), N, Cc, x.numel() // (N * Cc), ptr(stats))


class fork_streams:
    """`with fork_streams():` — independent passes of one network (the visual trunk over source 0, source 1, ...) are being
    issued on different HIP streams.  What they share is ordered with events instead of by the stream: a packed weight image
    is waited for by the streams that did not build it (Conv.pack), and the running-statistics update of a BatchNorm layer
    waits for the previous update of the same buffers — the passes update them in the order they were issued, exactly as
    if they had run back to back (vision_net.py:126-147 is called once per source, main.py:117-121)."""

    def __enter__(self):
        global _order
        self.prev, _order = _order, ({} if _order is None else _order)
        return self

    def __exit__(self, *exc):
        global _order
        _order = self.prev
        return False


_order = None      # fork_streams: running_mean.data_ptr() -> (event of the last update, its stream)


def bn_finalize(stats, count, gamma, beta, rmean, rvar, momentum, eps, training, like, num_batches_tracked=None, updates=1):
    Cc = gamma.numel()
    out = _f32((4, Cc), like)  # scale, shift, mean, invstd
    ordered = _order is not None and training and rmean is not None
    if ordered:
        last = _order.get(rmean.data_ptr())
        if last is not None and last[1] != torch.cuda.current_stream():
            torch.cuda.current_stream().wait_event(last[0])
    call("avsep_bn_finalize", ptr(stats), float(count), ptr(gamma), ptr(beta), ptr(rmean), ptr(rvar),
         float(momentum), float(eps), Cc, int(training), ptr(out[0]), ptr(out[1]), ptr(out[2]), ptr(out[3]),
         ptr(num_batches_tracked), int(updates))
    if ordered:
        ev = torch.cuda.Event()
        ev.record()
        _order[rmean.data_ptr()] = (ev, torch.cuda.current_stream())
    return out


def bn_bwd_coeffs(bstats, count, gamma, mean, invstd, dgamma=None, dbeta=None):
    """(dgamma, dbeta, pqr); `dgamma` / `dbeta`: destinations to write into instead of fresh tensors."""
    Cc = gamma.numel()
    dgamma = dgamma if dgamma is not None else _f32((Cc,), gamma)
    dbeta = dbeta if dbeta is not None else _f32((Cc,), gamma)
    pqr = _f32((3, Cc), gamma)
    call("avsep_bn_bwd_coeffs", ptr(bstats), float(count), ptr(gamma), ptr(mean), ptr(invstd), Cc,
         ptr(dgamma), ptr(dbeta), ptr(pqr))
    return dgamma, dbeta, pqr


def _drop_twin(t):
    """`t` is about to be overwritten through a raw pointer: forget its converted twin (to_b16 / to_f32 cache)."""
    if t is not None and hasattr(t, "_avsep_twin"):
        other = t._avsep_twin[0]
        if hasattr(other, "_avsep_twin") and other._avsep_twin[0] is t:
            del other._avsep_twin
        del t._avsep_twin


def _same_fmt(main, *others):
    """The elementwise kernels take all their tensor operands in ONE format: that of `main`."""
    f = fmt_of(main)
    return [as_fmt(o, f) for o in others]


def bn_bwd_apply_(dz, y, pqr, out=None, fresh=False, to_b16_out=False):
    """dy = p*dz + q*y + r per channel, in y's storage format; in place on dz unless `out` is given or `fresh` asks for a
    new tensor.  `to_b16_out` (fp32 dz and y, C % 16 == 0): write the result as a B16 image in the same pass — its consumers
    are bf16 kernels.  Returns the result tensor."""
    N, Cc, H, W = dims(y)
    if to_b16_out and not is_b16(y) and not is_b16(dz) and b16_ok(Cc):
        dst = _b16((N, Cc, H, W), y)
        call("avsep_bn_bwd_apply_to_b16", ptr(dz), ptr(y), ptr(pqr), N, Cc, H * W, ptr(dst))
        return dst
    (dz,) = _same_fmt(y, dz)
    dst = torch.empty_like(y) if fresh else (dz if out is None else out)
    _drop_twin(dst)
    if is_b16(y):
        call("avsep_b16_bn_bwd_apply", ptr(dz), ptr(y), ptr(pqr), N, Cc, H * W, ptr(dst))
    else:
        call("avsep_bn_bwd_apply", ptr(dz), ptr(y), ptr(pqr), N, Cc, H * W, ptr(dst))
    return dst


def affine_act(y, scale, shift, residual, act, res_scale=None, res_shift=None):
    N, Cc, H, W = dims(y)
    z = torch.empty_like(y)
    (residual,) = _same_fmt(y, residual)
    if is_b16(y):
        call("avsep_b16_affine_act", ptr(y), ptr(scale), ptr(shift), ptr(residual), ptr(res_scale), ptr(res_shift), act, N, Cc,
             H * W, ptr(z))
    else:
        call("avsep_affine_act", ptr(y), ptr(scale), ptr(shift), ptr(residual), ptr(res_scale), ptr(res_shift), act, N, Cc,
             H * W, ptr(z))
    return z


def affine_act_bwd_(dz, y, scale, shift, residual, add, mean, invstd, act, bstats, res_scale=None, res_shift=None,
                    out=None, dz2=None, stats_only=False):
    """dz <- act'(scale*y+shift[+res]) * (dz [+ dz2]) (+ add) (in place unless `out`); accumulates bstats.  All tensor
    operands are brought to the format of `y`; the (possibly converted) result tensor is returned."""
    N, Cc, H, W = dims(y)
    dz, dz2, residual, add = _same_fmt(y, dz, dz2, residual, add)
    dst = dz if out is None else out
    if stats_only and is_b16(y):       # the B16 kernel can skip the write (dz itself is the result: no activation, no add)
        call("avsep_b16_affine_act_bwd", ptr(dz), ptr(dz2), ptr(y), ptr(scale), ptr(shift), ptr(residual), ptr(res_scale),
             ptr(res_shift), ptr(add), ptr(mean), ptr(invstd), act, N, Cc, H * W, None, ptr(bstats))
        return dz
    _drop_twin(dst)
    if is_b16(y):
        call("avsep_b16_affine_act_bwd", ptr(dz), ptr(dz2), ptr(y), ptr(scale), ptr(shift), ptr(residual), ptr(res_scale),
             ptr(res_shift), ptr(add), ptr(mean), ptr(invstd), act, N, Cc, H * W, ptr(dst), ptr(bstats))
    else:
        call("avsep_affine_act_bwd", ptr(dz), ptr(dz2), ptr(y), ptr(scale), ptr(shift), ptr(residual), ptr(res_scale), ptr(res_shift),
             ptr(add), ptr(mean), ptr(invstd), act, N, Cc, H * W, ptr(dst), ptr(bstats))
    return dst


class Cat:
    """relu(affine(cat(x0, x1))) -> bilinear x2 (the U-Net decoder's glue).  Two B16 sources (no broadcast vector) run the
    B16 kernels and produce / consume B16 images; anything else runs the fp32 kernels on fp32 copies."""

    def __init__(self, x0, x1, sc0=None, sh0=None, sc1=None, sh1=None, bcast0=False, hw=None):
        self.b16 = (not bcast0 and x1 is not None and (is_b16(x0) or is_b16(x1)) and b16_ok(channels(x0))
                    and b16_ok(channels(x1)))
        if self.b16:
            x0, x1 = to_b16(x0), to_b16(x1)
        else:
            x0, x1 = to_f32(x0), to_f32(x1)
        N, C0 = (x0.shape[0], x0.shape[1]) if bcast0 else dims(x0)[:2]
        C1 = channels(x1) if x1 is not None else 0
        H, W = hw if hw is not None else (dims(x1)[2:] if x1 is not None else dims(x0)[2:])
        self.shape = (N, C0, C1, H, W)
        self.keep = (x0, x1, sc0, sh0, sc1, sh1)
        d = CatDesc()
        d.N, d.C0, d.C1, d.H, d.W, d.bcast0, d.bcast1 = N, C0, C1, H, W, int(bcast0), 0
        d.x0, d.x1 = ptr(x0), ptr(x1)
        d.scale0, d.shift0, d.scale1, d.shift1 = ptr(sc0), ptr(sh0), ptr(sc1), ptr(sh1)
        self.d, self.ref, self.like, self.bcast0 = d, C.byref(d), (x1 if x1 is not None else x0), bcast0

    def fwd(self):
        N, C0, C1, H, W = self.shape
        x0, x1, sc0, sh0, sc1, sh1 = self.keep
        if self.b16:
            out = _b16((N, C0 + C1, 2 * H, 2 * W), self.like)
            call("avsep_b16_relu_up2x_fwd", ptr(x0), ptr(x1), ptr(sc0), ptr(sh0), ptr(sc1), ptr(sh1), N, C0, C1, H, W, ptr(out))
            return out
        out = _f32((N, C0 + C1, 2 * H, 2 * W), self.like)
        call("avsep_relu_up2x_fwd", self.ref, ptr(out))
        return out

    def bwd(self, dout, mean1=None, invstd1=None, bstats1=None, g0_acc=None):
        """g0_acc: an existing source-0 gradient to accumulate into (shared-encoder AV step; must be in this Cat's format)."""
        N, C0, C1, H, W = self.shape
        x0, x1, sc0, sh0, sc1, sh1 = self.keep
        if self.b16:
            dout = to_b16(dout)
            g0 = g0_acc if g0_acc is not None else _b16((N, C0, H, W), self.like)
            assert is_b16(g0), "accumulating into an fp32 skip gradient from a B16 decoder level"
            _drop_twin(g0)
            g1 = _b16((N, C1, H, W), self.like)
            call("avsep_b16_relu_up2x_bwd", ptr(x0), ptr(x1), ptr(sc0), ptr(sh0), ptr(sc1), ptr(sh1), N, C0, C1, H, W, ptr(dout),
                 ptr(g0), ptr(g1), ptr(mean1), ptr(invstd1), ptr(bstats1), int(g0_acc is not None))
            return g0, g1
        dout = to_f32(dout)
        g0 = g0_acc if g0_acc is not None else _f32((N, C0) if self.bcast0 else (N, C0, H, W), self.like)
        assert not is_b16(g0)
        _drop_twin(g0)
        g1 = _f32((N, C1, H, W), self.like) if C1 else None
        call("avsep_relu_up2x_bwd", self.ref, ptr(dout), ptr(g0), ptr(g1), ptr(mean1), ptr(invstd1),
             ptr(bstats1), int(g0_acc is not None))
        return g0, g1


def prepare(mag_mix, mags, log_freq, weighted, binary, fout=256):
    """mag_mix [B,1,F,T]; mags [S,B,1,F,T] (one buffer).  Returns mix_w, mags_w, log, weight, gt."""
    lib.require_gpu(mag_mix)
    S, B, _, Fin, T = mags.shape
    Fo = fout if log_freq else Fin
    mix_w, logm, weight = (_f32((B, 1, Fo, T), mag_mix) for _ in range(3))
    mags_w, gt = _f32((S, B, 1, Fo, T), mag_mix), _f32((S, B, 1, Fo, T), mag_mix)
    call("avsep_prepare", ptr(mag_mix), ptr(mags), S, B, Fin, T, Fo, int(bool(log_freq)), int(bool(weighted)),
         int(bool(binary)), ptr(mix_w), ptr(mags_w), ptr(logm), ptr(weight), ptr(gt))
    return mix_w, mags_w, logm, weight, gt


def warp(x, hout, wout, warp_flag):
    B, Cc, Hin, Win = x.shape
    y = _f32((B, Cc, hout, wout), x)
    call("avsep_warp", ptr(x), B * Cc, Hin, Win, hout, wout, int(warp_flag), ptr(y))
    return y


def sgd_momentum_(p, g, buf, lr, momentum, weight_decay, grad_scale, first):
    call("avsep_sgd_momentum", ptr(p), ptr(g), ptr(buf), p.numel(), float(lr), float(momentum),
         float(weight_decay), float(grad_scale), int(first))


def temporal_mean(x, B, T):
    chw = x.numel() // (B * T)
    y = _f32((B,) + tuple(x.shape[1:]), x)
    call("avsep_temporal_mean_fwd", ptr(x), B, T, chw, ptr(y))
    return y


def temporal_mean_bwd(dy, B, T):
    chw = dy.numel() // B
    dx = _f32((B * T,) + tuple(dy.shape[1:]), dy)
    call("avsep_temporal_mean_bwd", ptr(dy), B, T, chw, ptr(dx))
    return dx


def maxpool3x3s2(x, scale=None, shift=None, act=0):
    """MaxPool2d(3,2,1) of act(scale*x+shift) (the stem's BN+ReLU folded into the pooling read).  B16 in -> B16 out with a
    one-byte winning-tap image; fp32 in -> fp32 out with int32 flat indices."""
    N, Cc, H, W = dims(x)
    Ho, Wo = out_size(H, 3, 2, 1, 1), out_size(W, 3, 2, 1, 1)
    if is_b16(x):
        y = _b16((N, Cc, Ho, Wo), x)
        idx = torch.empty((N, Cc // 16, Ho, Wo, 16), dtype=torch.uint8, device=x.device)
        call("avsep_b16_maxpool3x3s2_fwd", ptr(x), ptr(scale), ptr(shift), act, N, Cc, H, W, ptr(y), ptr(idx))
        return y, idx
    y = _f32((N, Cc, Ho, Wo), x)
    idx = torch.empty((N, Cc, Ho, Wo), dtype=torch.int32, device=x.device)
    call("avsep_maxpool3x3s2_fwd", ptr(x), ptr(scale), ptr(shift), act, Cc, N * Cc, H, W, ptr(y), ptr(idx))
    return y, idx


def maxpool3x3s2_bwd(dy, idx, H, W):
    N, Cc = dy.shape[:2]
    dx = _f32((N, Cc, H, W), dy)
    call("avsep_maxpool3x3s2_bwd", ptr(dy), ptr(idx), N * Cc, H, W, ptr(dx))
    return dx


def space_to_depth2(x, cp=16, b16=None):
    """[N,C,H,W] -> [N,cp,H/2+3,W/2+3]: the 2x2 phases of x as channels, zero border (2 before, 1 after): see avsep.h.
    With `b16` (default: the arithmetic mode keeps B16 activations) the result is written as a one-block B16 image."""
    N, Cc, H, W = x.shape
    if (want_b16(cp) if b16 is None else b16) and cp == 16:
        xs = _b16((N, 16, H // 2 + 3, W // 2 + 3), x)
        call("avsep_b16_space_to_depth2", ptr(x), N, Cc, H, W, ptr(xs))
        return xs
    xs = _f32((N, cp, H // 2 + 3, W // 2 + 3), x)
    call("avsep_space_to_depth2", ptr(x), N, Cc, H, W, cp, ptr(xs))
    return xs


def maxpool_bn_relu_bwd_stats(g, idx, y, bnrow, bstats, g2=None):
    """BatchNorm-backward sums of the stem tail taken over the pooled grid (see avsep.h); bstats is accumulated into.
    `g2`: a second gradient of the pooled map, added on the fly (B16 images only; fp32 callers add it beforehand)."""
    N, Cc, H, W = dims(y)
    if is_b16(y):
        g, g2 = to_b16(g), to_b16(g2)
        call("avsep_b16_maxpool_bn_relu_bwd", ptr(g), ptr(g2), ptr(idx), ptr(y), ptr(bnrow[0]), ptr(bnrow[1]), ptr(bnrow[2]),
             ptr(bnrow[3]), None, N, Cc, H, W, ptr(bstats), None, 0)
        return
    assert g2 is None
    call("avsep_maxpool_bn_relu_bwd_stats", ptr(to_f32(g)), ptr(idx), ptr(y), ptr(bnrow[0]), ptr(bnrow[1]), ptr(bnrow[2]), ptr(bnrow[3]),
         N, Cc, H, W, ptr(bstats))


def maxpool_bn_relu_bwd_apply(g, idx, y, bnrow, pqr, g2=None, out_f32=False):
    """dL/d(raw stem conv output) from dL/d(pooled): max-pool backward + ReLU mask + folded BatchNorm backward in one pass.
    `out_f32` (B16 inputs): write the result as fp32 NCHW (its only consumer, the stem's weight gradient, is an fp32 kernel)."""
    N, Cc, H, W = dims(y)
    if is_b16(y):
        g, g2 = to_b16(g), to_b16(g2)
        dy = _f32((N, Cc, H, W), y) if out_f32 else torch.empty_like(y)
        call("avsep_b16_maxpool_bn_relu_bwd", ptr(g), ptr(g2), ptr(idx), ptr(y), ptr(bnrow[0]), ptr(bnrow[1]), None, None, ptr(pqr),
             N, Cc, H, W, None, ptr(dy), int(out_f32))
        return dy
    dy = torch.empty_like(y)
    assert g2 is None
    call("avsep_maxpool_bn_relu_bwd_apply", ptr(to_f32(g)), ptr(idx), ptr(y), ptr(bnrow[0]), ptr(bnrow[1]), ptr(pqr), N, Cc, H, W, ptr(dy))
    return dy


class Stft:
    """librosa-style STFT/iSTFT plan (bases built once on the device)."""

    def __init__(self, device, n_fft=1022, hop=256, pad_mode="reflect"):
        self.n_fft, self.hop, self.reflect = n_fft, hop, int(pad_mode == "reflect")
        L = lib.load()
        with torch.cuda.device(device):
            self.fwd_basis = torch.empty((L.avsep_stft_basis_floats(n_fft, 0),), dtype=torch.float32, device=device)
            self.inv_basis = torch.empty((L.avsep_stft_basis_floats(n_fft, 1),), dtype=torch.float32, device=device)
            call("avsep_stft_basis", n_fft, ptr(self.fwd_basis), ptr(self.inv_basis))

    def stft(self, wav, want_phase=True):
        """wav [R,L] -> mag, phase [R, n_fft/2+1, 1+L//hop]."""
        R, Ln = wav.shape
        bins, frames = self.n_fft // 2 + 1, 1 + Ln // self.hop
        nbytes = lib.load().avsep_stft_workspace_bytes(R, Ln, self.n_fft, self.hop)
        ws = torch.empty((nbytes // 4,), dtype=torch.float32, device=wav.device)
        mag = _f32((R, bins, frames), wav)
        phase = _f32((R, bins, frames), wav) if want_phase else None
        call("avsep_stft_mag", ptr(wav), R, Ln, self.n_fft, self.hop, self.reflect, ptr(self.fwd_basis), ptr(mag),
             ptr(phase), ptr(ws), nbytes)
        return mag, phase

    def istft(self, mag, phase):
        R, bins, frames = mag.shape
        out_len = self.hop * (frames - 1)
        nbytes = lib.load().avsep_istft_workspace_bytes(R, self.n_fft, frames)
        ws = torch.empty((nbytes // 4,), dtype=torch.float32, device=mag.device)
        wav = _f32((R, out_len), mag)
        call("avsep_istft", ptr(mag), ptr(phase), R, self.n_fft, self.hop, frames, ptr(self.inv_basis), ptr(wav),
             out_len, ptr(ws), nbytes)
        return wav
