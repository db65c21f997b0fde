"""-m gpu: every HIP entry point against plain PyTorch fp32 on the CPU (conv/BN/upsample are
floating-point kernels, so the torch reference is kept beside the oracle; tolerances are stated
per test: fp32 accumulation-order noise only)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import assert_close

pytestmark = pytest.mark.gpu


def _pkg():
    import avsep_amd
    return avsep_amd


CONV_CASES = [
    # N, Cin, H, W, Cout, k, stride, pad, dil
    (2, 1, 32, 32, 64, 4, 2, 1, 1),        # U-Net d1 shape family
    (2, 64, 16, 16, 128, 4, 2, 1, 1),
    (3, 96, 8, 8, 40, 3, 1, 1, 1),         # ragged channel counts
    (2, 64, 12, 40, 136, 3, 1, 1, 1),      # 3x3 halo-patch path, 4x32 tiles, ragged H/W/Cout
    (2, 36, 19, 20, 48, 3, 1, 1, 1),       # 3x3 halo-patch path, 8x16 tiles, 64-row M tile
    (2, 160, 10, 6, 130, 3, 1, 1, 1),      # > one 128 tile in M, odd spatial
    (2, 3, 30, 30, 64, 7, 2, 3, 1),        # resnet stem
    (2, 32, 14, 14, 48, 3, 1, 2, 2),       # dilated
    (2, 32, 15, 15, 48, 3, 2, 1, 1),       # 3x3 stride 2, odd size
    (2, 32, 14, 14, 64, 1, 2, 0, 1),       # 1x1 stride 2 downsample
    (4, 256, 4, 4, 256, 4, 2, 1, 1),       # deep, tiny spatial
    (8, 512, 4, 4, 512, 4, 2, 1, 1),       # deep level: few tiles, K = 8192 -> split-K slabs + combine (bias, stats)
    (4, 640, 8, 8, 96, 3, 1, 1, 1),        # split-K on the 3x3 im2col path (W < 16)
    (1, 20, 12, 16, 2, 3, 1, 1, 1),        # Cout = 2 (last U-Net conv) -> direct small-Cout kernels
    (2, 37, 21, 144, 2, 3, 1, 1, 1),       # same path: ragged H, W > one 128 tile, Cin not a multiple of the chunk
    (2, 16, 10, 10, 3, 3, 1, 1, 1),        # W % 16 != 0 -> falls back to the MFMA path
    (3, 32, 14, 14, 40, 3, 1, 2, 2),       # dilated 3x3 (ResNet layer4) on the halo-patch kernel, 14x14 map
    (2, 64, 14, 14, 136, 3, 1, 2, 2),      # same, 128-row weight-gradient tiles, rows only 8-byte aligned
    (2, 64, 18, 22, 72, 3, 1, 1, 1),       # W % 4 == 2 undilated weight gradient
    (2, 36, 20, 40, 136, 3, 1, 2, 2),      # dilated, 4x32 tiles
    (3, 32, 14, 14, 48, 3, 1, 1, 1),       # 14x14 map, undilated
    (2, 1, 32, 48, 24, 4, 2, 1, 1),        # first U-Net conv (Cin = 1): blocked small-Cin data gradient
    (2, 3, 20, 16, 10, 4, 2, 1, 1),        # same kernel, Cin = 3, ragged thread blocks
    (2, 8, 64, 64, 40, 4, 2, 1, 1),        # 4x4 s2 on the halo-patch kernel: 4x32 tiles, 64-row M tiles; dgrad: 4 parity classes
    (2, 34, 40, 72, 36, 4, 2, 1, 1),       # same, ragged tiles (Ho = 20, Wo = 36), Cin = 34 (17 channel pairs)
    (24, 6, 64, 64, 136, 4, 2, 1, 1),      # 128-row M tiles (>= 384 workgroups), second M tile ragged
    (96, 4, 32, 32, 136, 4, 2, 1, 1),      # 8x16 tiles x 128 rows
    (3, 32, 34, 32, 64, 4, 2, 1, 1),       # 8x16 tiles x 64 rows, Ho = 17
    # flat-pixel tiles (conv_flat.hip): 128 consecutive pixels of (n, h, w) per tile, tiles cross image boundaries
    (5, 32, 7, 7, 40, 3, 1, 1, 1),         # 7x7 maps (resnet18fc layer4): up to 4 images per tile
    (3, 16, 28, 28, 40, 3, 1, 1, 1),       # 28x28
    (2, 16, 28, 28, 40, 3, 1, 2, 2),       # 28x28 dilated (dilate_scale 8, layer3)
    (2, 8, 56, 56, 72, 3, 1, 1, 1),        # 56x56, two 64-row M tiles
    (2, 16, 20, 14, 40, 3, 1, 1, 1),       # H > W
    (130, 8, 14, 14, 136, 3, 1, 1, 1),     # 128-row M tiles (>= 384 workgroups), ragged second M tile, last pixel tile partial
    (130, 8, 14, 14, 136, 3, 1, 2, 2),     # same, dilated
    (8, 4, 56, 56, 136, 3, 1, 1, 1),       # 128-row M tiles at 56x56
    # conv_misc.hip: the trunk's 3x3/s2 and 1x1 convs on the halo-patch kernel
    (3, 32, 56, 56, 48, 3, 2, 1, 1),       # 3x3 / stride 2: forward on the stride-2 patch, dgrad = 1/2/2/4-tap parity classes
    (2, 64, 30, 28, 144, 3, 2, 1, 1),      # same, 128-row dgrad tiles? (Cin = 64 -> 64-row), 14-wide outputs, H != W
    (24, 16, 56, 56, 144, 3, 2, 1, 1),     # forward on 128-row M tiles (>= 384 workgroups); dgrad (Cin < 32) on the im2col kernel
    (3, 64, 14, 14, 144, 1, 1, 0, 1),      # 1x1 (layer4.0 downsample)
    (2, 32, 28, 28, 48, 1, 2, 0, 1),       # 1x1 / stride 2: dgrad zero-fills the skipped pixels
    (40, 128, 28, 28, 256, 1, 2, 0, 1),    # 1x1 / stride 2 at the layer3.0 downsample shape, 128-row tiles both ways
    # wgrad_smallci.hip: weight gradient over <= 4 input channels (ResNet stem 7x7/s2, U-Net first conv 4x4/s2): family asserted
    (3, 3, 32, 64, 72, 7, 2, 3, 1),        # stem geometry, two 64-channel tiles (second ragged), 5 N tiles of 147 columns
    (2, 3, 224, 224, 64, 7, 2, 3, 1),      # the stem itself
    (3, 1, 64, 32, 40, 4, 2, 1, 1),        # Cin = 1: one N tile, half of the waves idle
    # conv_wino.hip: Winograd F(2x2, 3x3) forward + data gradient (needs >= 128 workgroups, conv_wino.hip wn_applicable,
    # hence the batch sizes); the kernel family of these rows is ASSERTED (EXPECT_FAMILY below)
    (64, 48, 32, 32, 80, 3, 1, 1, 1),      # one 8x8-tile group per workgroup (U-Net decoder maps), ragged second M tile
    (22, 64, 56, 56, 72, 3, 1, 1, 1),      # four 4x4-tile groups (56x56: 7x7 groups per image)
    (80, 48, 28, 28, 72, 3, 1, 1, 1),      # two 2x16-tile groups (28x28)
    (300, 48, 14, 14, 56, 3, 1, 1, 1),     # eight 1x8-tile groups (14x14), last workgroup partial
    (260, 48, 14, 14, 56, 3, 1, 2, 2),     # dilation 2: the four 7x7 parity sub-images, odd sub-image size
    (80, 48, 28, 28, 56, 3, 1, 2, 2),      # dilation 2 at 28x28: 14x14 sub-images on the 1x8-tile groups
    (72, 48, 20, 36, 72, 3, 1, 1, 1),      # ragged group grid (H, W not multiples of the group)
    # wgrad_wino.hip: Winograd weight gradient (Cin % 64 == 0, >= 192 workgroups); forward / dgrad as above
    (64, 64, 32, 32, 80, 3, 1, 1, 1),      # 4x16-pixel chunks, ragged second output-channel tile
    (300, 128, 14, 14, 56, 3, 1, 1, 1),    # two 2x16-pixel groups per chunk (14x14), two input-channel tiles
    (260, 128, 14, 14, 56, 3, 1, 2, 2),    # dilation 2: 7x7 parity sub-images on 8x8-pixel chunks, strided dY loads
    (40, 64, 20, 36, 72, 3, 1, 1, 1),      # ragged group grid
    # wgrad4d_kernel (wgrad_wino.hip): 4x4 / stride 2 weight gradient on the same skeleton (Cin % 64 == 0, >= 128 workgroups)
    (64, 64, 32, 32, 80, 4, 2, 1, 1),      # 4x16-pixel chunks, ragged second output-channel tile
    (64, 64, 24, 24, 72, 4, 2, 1, 1),      # 8x8-pixel chunks
    (72, 64, 20, 36, 72, 4, 2, 1, 1),      # ragged group grid
    # conv_wino4.hip: Winograd F(4x4, 3x3) forward + data gradient (H, W multiples of 4 and >= 16, >= 192 workgroups of
    # 64 channels x 32 tiles at the planned batch); kernel family ASSERTED
    (40, 72, 32, 64, 80, 3, 1, 1, 1),      # one 4x8-tile group per workgroup (16x32 pixels), 9 / 10 K-tiles, ragged second M tile
    (200, 72, 16, 16, 136, 3, 1, 1, 1),    # two 4x4-tile groups (16x16 maps: two images per workgroup), three M tiles
    (16, 72, 56, 56, 72, 3, 1, 1, 1),      # eight 2x2-tile groups (56x56: 7x7 groups per image)
    (72, 72, 20, 36, 72, 3, 1, 1, 1),      # ragged group grid, last workgroup partial
    (400, 72, 14, 14, 72, 3, 1, 1, 1),     # 14x14 (ResNet layer3): partial last tile row / column, masked scalar stores
    (72, 72, 18, 30, 72, 3, 1, 1, 1),      # H, W = 2 mod 4 on the 2x2-tile groups
    (260, 72, 14, 14, 72, 3, 1, 2, 2),     # dilation 2: the four 7x7 parity sub-images, one 2x2-tile group each (ResNet layer4)
    (80, 72, 28, 28, 72, 3, 1, 2, 2),      # dilation 2 at 28x28: 14x14 sub-images on the 4x4-tile groups
    # wgrad_wino4.hip: Winograd F(4x4, 3x3) weight gradient (Cin % 32 == 0, 8x8 regions >= 70 % full, >= 128 workgroups)
    (48, 96, 32, 48, 80, 3, 1, 1, 1),      # three input-channel blocks, ragged second output-channel tile, 14 regions per split
    (26, 64, 24, 40, 72, 3, 1, 1, 1),      # 3 x 5 regions per image, 7 regions per split (splits cross images), odd step counts
    # ... the masked form: maps the 8x8 regions do not divide, and dilation 2 (parity sub-maps)
    (200, 64, 14, 14, 72, 3, 1, 1, 1),     # 14x14 (ResNet layer3): 2 x 2 regions, the second 6 pixels wide, partial dY tiles
    (40, 64, 28, 28, 72, 3, 1, 1, 1),      # 28x28 (ResNet layer2): the fourth region row / column is half a region
    (260, 64, 14, 14, 72, 3, 1, 2, 2),     # dilation 2 (ResNet layer4): one 7x7 region per parity sub-map, pixels 2 apart
    (72, 64, 20, 36, 72, 3, 1, 1, 1),      # H != W, both ragged (3 x 5 regions per image)
    (36, 64, 24, 32, 72, 3, 1, 2, 2),      # dilation 2 over 12x16 sub-maps: two region rows, the second half valid
    (40, 64, 28, 28, 72, 3, 1, 2, 2),      # dilation 2 at 28x28: 2 x 4 paired-parity regions per sub-map row parity, both ragged
    (60, 64, 20, 29, 72, 3, 1, 1, 1),      # odd W: dY element by element
]


def _rows_after(marker_case, count):
    i = CONV_CASES.index(marker_case)
    return CONV_CASES[i:i + count]


# kernel family a row exists to exercise: a moved dispatch threshold must fail the test, not silently fall back to the
# direct-form kernel (which passes the same numeric bound)
EXPECT_FAMILY = {}
EXPECT_MASK = {}          # kernel families a row forbids (kernels.set_algo_mask): the F(2x2) rows keep F(4x4) out of their way
for _c in _rows_after((64, 48, 32, 32, 80, 3, 1, 1, 1), 7):
    EXPECT_FAMILY[_c] = {"fwd": "wino_kernel", "dgrad": "wino_kernel"}
    EXPECT_MASK[_c] = ("winograd4",)
for _c in _rows_after((64, 64, 32, 32, 80, 3, 1, 1, 1), 4):
    EXPECT_FAMILY[_c] = {"fwd": "wino_kernel", "dgrad": "wino_kernel", "wgrad": "winow_kernel"}
    EXPECT_MASK[_c] = ("winograd4",)
for _c in _rows_after((40, 72, 32, 64, 80, 3, 1, 1, 1), 8):
    EXPECT_FAMILY[_c] = {"fwd": "wino4_kernel", "dgrad": "wino4_kernel"}
EXPECT_FAMILY[(48, 96, 32, 48, 80, 3, 1, 1, 1)] = {"fwd": "wino4_kernel", "dgrad": "wino4_kernel", "wgrad": "winow4_kernel"}
EXPECT_FAMILY[(26, 64, 24, 40, 72, 3, 1, 1, 1)] = {"wgrad": "winow4_kernel"}
for _c in _rows_after((200, 64, 14, 14, 72, 3, 1, 1, 1), 7):
    EXPECT_FAMILY[_c] = {"wgrad": "winow4_kernel"}
for _c in _rows_after((64, 64, 32, 32, 80, 4, 2, 1, 1), 3):
    EXPECT_FAMILY[_c] = {"wgrad": "wgrad4d_kernel"}
for _c in _rows_after((3, 3, 32, 64, 72, 7, 2, 3, 1), 3) + [(2, 1, 32, 48, 24, 4, 2, 1, 1)]:
    EXPECT_FAMILY[_c] = {"wgrad": "smallci_wgrad_kernel"}


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(dev, case):
    K = _pkg().kernels
    K.set_algo_mask(*EXPECT_MASK.get(case, ()))
    try:
        _conv_case(dev, K, case)
    finally:
        K.set_algo_mask()


def _conv_case(dev, K, case):
    N, Cin, H, W, Cout, k, s, p, d = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    b = torch.randn(Cout, generator=g)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, b, s, p, d)
    dy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(dy)
    xd, wd, bd, dyd = x.to(dev), w.to(dev), b.to(dev), dy.to(dev)
    cv = K.Conv(xd, Cout, k, s, p, d)
    for mode, fam in EXPECT_FAMILY.get(case, {}).items():
        assert cv.kernel_name(mode) == fam, (case, mode, cv.kernel_variant(mode))
    st = K.zeros_stats(Cout, xd)
    y = cv.fwd(cv.pack(wd, 0), bd, st)
    assert_close(y, y_ref, 2e-5, "fwd")
    assert_close(cv.fwd(cv.pack(wd, 0), bd, None), y_ref, 2e-5, "fwd without stats (direct path when Cout <= 4)")
    st_ref = torch.cat([y_ref.double().sum((0, 2, 3)), (y_ref.double() ** 2).sum((0, 2, 3))])
    assert_close(st, st_ref, 1e-5, "stats")
    dx = cv.dgrad(cv.pack(wd, 1), dyd)
    assert_close(dx, xr.grad, 2e-5, "dgrad")
    dw, db = cv.wgrad(dyd, want_bias=True)
    assert_close(dw, wr.grad, 2e-5, "wgrad")
    assert_close(db, dy.sum((0, 2, 3)), 2e-5, "dbias")


def _bf16(t):
    return t.to(torch.bfloat16).to(torch.float64)


BF16_CASES = [
    # N, Cin, H, W, Cout, k, stride, pad, dil, affine+act (0 none, 1 ReLU, 2 LeakyReLU)
    (2, 32, 12, 40, 144, 3, 1, 1, 1, 1),     # 4x32 tiles x 128 rows (+ ragged second M tile), ragged H / W
    (2, 32, 12, 40, 136, 3, 1, 1, 1, 1),     # Cout % 16 != 0: the data gradient has no bf16 kernel and runs in exact f32
    (2, 48, 19, 20, 48, 3, 1, 1, 1, 0),      # 8x16 tiles x 64 rows, three K-tiles
    (70, 16, 16, 64, 144, 3, 1, 1, 1, 1),    # 8x32 tiles, 512-thread workgroups (>= 512 of them)
    (130, 16, 32, 16, 144, 3, 1, 1, 1, 0),   # 16x16 tiles, 512 threads, 128 rows
    (2, 32, 20, 40, 144, 3, 1, 2, 2, 1),     # dilated
    (3, 32, 14, 14, 48, 3, 1, 1, 1, 1),      # flat 14x14, 128-pixel tiles crossing images
    (3, 32, 14, 14, 48, 3, 1, 2, 2, 0),      # flat 14x14 dilated
    (5, 32, 7, 7, 48, 3, 1, 1, 1, 1),        # flat 7x7
    (700, 16, 14, 14, 144, 3, 1, 1, 1, 1),   # flat 14x14, 256-pixel tiles, 128 rows, last tile partial
    (2, 32, 64, 64, 48, 4, 2, 1, 1, 2),      # 4x4 / stride 2 (U-Net encoder): forward + 4 parity-class data gradients
    (2, 48, 40, 72, 144, 4, 2, 1, 1, 2),     # same, ragged tiles, three M tiles of 64 rows
    (70, 16, 64, 64, 80, 4, 2, 1, 1, 0),     # same, 512-thread workgroups
    (3, 48, 10, 72, 160, 3, 1, 1, 1, 1),     # weight gradient: 4x32 pixel tiles, ragged rows / columns / channel blocks
    (4, 64, 18, 22, 40, 3, 1, 1, 1, 2),      # W % 4 == 2 (float2 staging), 32-wide tiles
    (6, 32, 14, 14, 144, 3, 1, 2, 2, 1),     # dilated 14x14 (ResNet layer4): 16-wide tiles, rows 8-byte aligned
    (2, 32, 12, 40, 48, 3, 1, 2, 2, 0),      # dilated, 32-wide 2-row tiles
    (9, 64, 16, 16, 128, 3, 1, 1, 1, 0),     # several pixel tiles per slab, 16-wide
    (8, 64, 8, 8, 144, 3, 1, 1, 1, 0),       # flat 8x8 (U-Net u6 shape family), weight gradient with half-empty k-steps
    (4, 512, 4, 4, 160, 3, 1, 1, 1, 1),      # flat 4x4, K = 4608: split-K slabs + combine (bias, stats) / slab reduce
    (6, 256, 8, 8, 144, 3, 1, 1, 1, 0),      # flat 8x8, two K splits
    (4, 32, 16, 16, 48, 4, 2, 1, 1, 2),      # 4x4 / stride 2 with 8-wide outputs (U-Net d5)
    (3, 64, 24, 64, 160, 4, 2, 1, 1, 2),     # 4x4 / stride 2 weight gradient: 2x32 output tiles, ragged rows, two M tiles
    (5, 32, 32, 32, 48, 4, 2, 1, 1, 1),      # same, 4x16 output tiles (16-wide outputs)
    (2, 48, 20, 72, 40, 4, 2, 1, 1, 0),      # same, ragged channel blocks (48 = 32 + 16) and columns (Wo = 36)
    (3, 32, 56, 56, 48, 3, 2, 1, 1, 1),      # 3x3 / stride 2 (ResNet layer2.0 conv1): forward, 1/2/2/4-tap parity-class dgrad, wgrad
    (2, 64, 30, 28, 144, 3, 2, 1, 1, 0),     # same, 14-wide outputs (layer3.0): forward + dgrad in bf16, weight gradient in f32
    (3, 64, 14, 14, 144, 1, 1, 0, 1, 1),     # 1x1 (layer4.0 downsample): forward + dgrad
    (2, 32, 28, 28, 48, 1, 2, 0, 1, 0),      # 1x1 / stride 2 (layer2.0 / layer3.0 downsample): dgrad zero-fills the skipped pixels
    (7, 64, 24, 40, 64, 3, 1, 1, 1, 1),      # resident-weights persistent form (ResNet layer1): 64 -> 64, folded affine + ReLU, ragged tiles
    (40, 64, 16, 64, 48, 3, 1, 1, 1, 0),     # same, raw input, 48 output rows
    (150, 64, 24, 40, 64, 3, 1, 1, 1, 1),    # same, 900 ragged tiles on 256 persistent workgroups: several tiles per workgroup
    (6, 16, 27, 43, 64, 4, 1, 0, 1, 0),      # the stem's 4x4 / stride-1 form over 16 space-to-depth channels (resident weights; Cin = 16 wgrad)
]


@pytest.mark.parametrize("case", BF16_CASES)
def test_conv_bf16_operands(dev, case):
    """avsep_conv_desc.prec = AVSEP_PREC_BF16.  The bf16 kernels stage B16 images (bf16, [N][C/16][H][W][16]): the input is
    rounded to bf16 when it is stored, the folded affine + activation are applied in fp32 to the stored value (one fmaf)
    and the result is rounded to bf16 again on its way into LDS; weights (and dY) are rounded once.  The kernel must equal a
    float64 convolution of exactly those rounded operands up to fp32 accumulation order; the BatchNorm statistics are those
    of the fp32 result.  A call without a bf16 kernel runs in exact f32 on the unrounded operands.  Outputs requested as
    B16 images must be the bf16 rounding of the fp32 outputs, bit for bit.  Against the unrounded float32 convolution the
    difference is the operand rounding itself (2^-9 relative per rounding): reported, bounded loosely."""
    K = _pkg().kernels
    N, Cin, H, W, Cout, k, s, p, d, aff = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    b = torch.randn(Cout, generator=g)
    sc, sh = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3

    def activated(x_):
        if not aff:
            return x_
        # the kernel folds the affine with ONE rounding (fmaf): float64 product + sum rounded once to float32 is the same
        # value; a separately rounded product would differ in the last float32 bit and flip bf16 roundings
        v_ = (x_.double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)).float()
        return F.relu(v_) if aff == 1 else torch.where(v_ > 0, v_, 0.2 * v_)
    v = activated(x)                                       # exact-f32 path
    v16 = activated(x.to(torch.bfloat16).float())          # bf16 kernels: the stored image is rounded first
    t = lambda z: z.to(dev)
    cv = K.Conv(t(x), Cout, k, s, p, d, sc0=t(sc) if aff else None, sh0=t(sh) if aff else None, act0=aff, prec="bf16")
    bf_fwd, bf_dgrad, bf_wgrad = (cv.kernel_name(m) in ("convbf_kernel", "wgradb_kernel") for m in ("fwd", "dgrad", "wgrad"))
    assert bf_fwd, "every BF16_CASES geometry has a bf16 forward kernel"
    vr, wr = _bf16(v16).requires_grad_(True), _bf16(w).requires_grad_(True)
    y_ref = F.conv2d(vr, wr, b.double(), s, p, d)
    dy = torch.randn(y_ref.shape, generator=g)
    # the data gradient rounds dY and the weights; the reference for it is the conv-transpose of the rounded dY
    dx_ref, dw_ref = torch.autograd.grad(y_ref, (vr, wr), _bf16(dy))
    st = K.zeros_stats(Cout, t(x))
    y = cv.fwd(cv.pack(t(w), 0), t(b), st)
    assert y.dtype == torch.float32
    assert_close(y, y_ref, 2e-5, "fwd vs rounded operands")
    st_ref = torch.cat([y_ref.sum((0, 2, 3)), (y_ref ** 2).sum((0, 2, 3))])
    assert_close(st, st_ref, 1e-5, "stats")
    need, b16_out = cv.io_formats(0)
    assert need == K.FMT_B16
    if b16_out and Cout % 16 == 0:                         # the same call writing a B16 image: bf16(fp32 result), exactly
        y16 = cv.fwd(cv.pack(t(w), 0), t(b), None, out_b16=True)
        assert K.is_b16(y16) and K.dims(y16) == tuple(y.shape)
        assert torch.equal(K.to_f32(y16), y.to(torch.bfloat16).float()), "B16 forward output != bf16(fp32 output)"
    dx = cv.dgrad(cv.pack(t(w), 1), t(dy))
    if bf_dgrad:
        assert Cout % 16 == 0
        assert_close(dx, dx_ref, 2e-5, "dgrad vs rounded operands")
        if cv.io_formats(1)[1] and Cin % 16 == 0:
            dx16 = cv.dgrad(cv.pack(t(w), 1), K.to_b16(t(dy)), out_b16=True)
            assert K.is_b16(dx16)
            assert torch.equal(K.to_f32(dx16), dx.to(torch.bfloat16).float()), "B16 data gradient != bf16(fp32 data gradient)"
    else:   # no bf16 data-gradient kernel for these channel counts: exact f32 arithmetic on the unrounded operands
        v32 = v.clone().requires_grad_(True)
        assert_close(dx, torch.autograd.grad(F.conv2d(v32, w, b, s, p, d), v32, dy)[0], 2e-5, "dgrad (f32 fallback)")
    v32, w32 = v.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y32 = F.conv2d(v32, w32, b, s, p, d)
    assert_close(y, y32, 3e-2, "fwd vs unrounded fp32 (bf16 operand rounding)")
    # weight gradient: bf16 kernel (rounded dY and rounded activated input) where one exists, else exact f32
    dw, db = cv.wgrad(t(dy), want_bias=True)
    from conftest import rel_err
    e_bf, e_32 = rel_err(dw, dw_ref), rel_err(dw, torch.autograd.grad(y32, w32, dy)[0])
    assert (e_bf if bf_wgrad else e_32) <= 2e-5, (bf_wgrad, e_bf, e_32)
    # the bias gradient of a bf16 weight-gradient call is summed from the B16 image of dY
    assert_close(db, (_bf16(dy) if bf_wgrad else dy.double()).sum((0, 2, 3)), 2e-5, "dbias")


@pytest.mark.parametrize("case", [
    # N, Cin, H, W, Cout, k, stride, pad, affine (0 none, 2 BatchNorm + LeakyReLU)
    (5, 32, 4, 4, 48, 3, 1, 1, 0), (7, 64, 8, 8, 32, 3, 1, 1, 0), (3, 16, 4, 8, 16, 3, 1, 1, 2),
    (6, 32, 8, 8, 32, 4, 2, 1, 2), (10, 48, 4, 4, 64, 4, 2, 1, 0), (40, 16, 2, 2, 16, 3, 1, 1, 0), (9, 16, 8, 4, 32, 4, 2, 1, 2),
])
def test_small_map_weight_gradient_over_grid_image(dev, case):
    """bf16 mode, maps of at most 8x8 (the deep U-Net levels, audio_net.py:64-69,75-76): the weight gradient runs over ONE grid
    image of the batch (avsep_b16_grid_pack: images side by side, zero separators = every image's padding) on wgradb_kernel.
    Against float64 on the bf16-rounded operands; the grid image itself is checked position by position."""
    K = _pkg().kernels
    N, Cin, H, W, Cout, k, s, p, aff = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, Cin, H, W, generator=g)
    sc, sh = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
    t = lambda z: z.to(dev)
    x16 = x.to(torch.bfloat16).float()
    v = x16
    if aff:
        v = (x16.double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)).float()
        v = torch.where(v > 0, v, 0.2 * v)
    for xin in (K.to_b16(t(x)), t(x)):                    # B16 and fp32 inputs (an fp32 input is rounded as a B16 image would be)
        cv = K.Conv(xin, Cout, k, s, p, sc0=t(sc) if aff else None, sh0=t(sh) if aff else None, act0=aff, prec="bf16")
        geo = cv._grid_geometry()
        assert geo is not None and cv.kernel_name("wgrad") == "wgradb_kernel"
        gx, gy, (pyi, pxi), (pyo, pxo) = geo
        xg = K.to_f32(K.grid_pack(xin, N, Cin, H, W, gx, gy, pyi, pxi, t(sc) if aff else None, t(sh) if aff else None, aff)).cpu()
        ref = torch.zeros(1, Cin, gy * pyi, gx * pxi)
        for n in range(N):
            ref[0, :, (n // gx) * pyi:(n // gx) * pyi + H, (n % gx) * pxi:(n % gx) * pxi + W] = v[n].to(torch.bfloat16).float()
        assert torch.equal(xg, ref), "grid image"
        dy = torch.randn(N, Cout, cv.Ho, cv.Wo, generator=g)
        wz = torch.zeros(Cout, Cin, k, k, dtype=torch.float64, requires_grad=True)
        bz = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
        F.conv2d(_bf16(v), wz, bz, s, p).backward(_bf16(dy))
        dw, db = cv.wgrad(K.to_b16(t(dy)) if K.is_b16(xin) else t(dy), want_bias=True)
        assert_close(dw, wz.grad, 2e-5, "weight gradient over the grid image")
        assert_close(db, bz.grad, 2e-5, "bias gradient over the grid image")
    if k == 4:      # the stride-2 forward (U-Net encoder d5..d7) also runs over the grid image: real positions + statistics
        assert cv._grid_geometry(0) is not None and cv.kernel_name("fwd") == "convbf_kernel"
        w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
        b = torch.randn(Cout, generator=g)
        y_ref = F.conv2d(_bf16(v), _bf16(w), b.double(), s, p)
        st = K.zeros_stats(Cout, t(x))
        y = cv.fwd(cv.pack(t(w), 0), t(b), st, out_b16=False)
        assert y.dtype == torch.float32 and tuple(y.shape) == tuple(y_ref.shape)
        assert_close(y, y_ref, 2e-5, "forward over the grid image")
        assert_close(st, torch.cat([y_ref.sum((0, 2, 3)), (y_ref ** 2).sum((0, 2, 3))]), 1e-5, "statistics of the real positions")
        y16 = cv.fwd(cv.pack(t(w), 0), t(b), None, out_b16=True)
        assert K.is_b16(y16) and torch.equal(K.to_f32(y16), y.to(torch.bfloat16).float())
    K.grid_small_maps = False
    try:
        assert cv._grid_geometry() is None and cv._grid_geometry(0) is None
        if k == 4:
            assert_close(cv.fwd(cv.pack(t(w), 0), t(b), None, out_b16=False), y, 2e-2, "grid forward vs per-image forward")
        dw0, _ = cv.wgrad(t(dy))                          # the per-image path on the same operands (exact f32 or bf16 kernels)
    finally:
        K.grid_small_maps = True
    assert_close(dw0, dw, 2e-2, "grid path vs per-image path (operand rounding apart)")
    tiny = K.Conv(t(x[:2, :, :2, :2].contiguous()), Cout, 3, 1, 1, prec="bf16")
    assert tiny._grid_geometry() is None                  # two 2x2 maps: no grid image is worth a bf16 weight-gradient launch


def test_stem_weight_gradient_over_space_to_depth_b16(dev):
    """The ResNet stem's weight gradient in bf16 mode (vision_net.py:84-89 conv1): the 7x7/s2 conv runs as a 4x4/s1 conv over
    the space-to-depth frames (ONE 16-channel B16 block); csrc/wgrad_b16.hip wgradb_ci16_kernel makes (tap, ci) the GEMM's N
    dimension.  Against float64 on the bf16-rounded operands, in the 4x4 form and mapped back to the 7x7 taps."""
    K = _pkg().kernels
    from avsep_amd.models import vision_hip as VH
    g = torch.Generator().manual_seed(31)
    N, H, W, Co = 5, 64, 96, 64
    x = torch.randn(N, 3, H, W, generator=g)
    xs = K.space_to_depth2(x.to(dev), b16=True)
    cs = K.Conv(xs, Co, 4, 1, 0, prec="bf16")
    assert cs.kernel_name("wgrad") == "wgradb_kernel" and cs.io_formats(2)[0] == K.FMT_B16
    dy = torch.randn(N, Co, H // 2, W // 2, generator=g)
    dw2, _ = cs.wgrad(K.to_b16(dy.to(dev)))
    xs64 = K.to_f32(xs).double().cpu().requires_grad_(False)
    w2 = torch.zeros(Co, 16, 4, 4, dtype=torch.float64, requires_grad=True)
    F.conv2d(xs64, w2).backward(_bf16(dy))
    assert_close(dw2, w2.grad, 2e-5, "stem weight gradient, 4x4 form over the space-to-depth frames")
    w7 = torch.zeros(Co, 3, 7, 7, dtype=torch.float64, requires_grad=True)
    F.conv2d(_bf16(x), w7, None, 2, 3).backward(_bf16(dy))
    assert_close(VH._stem_s2d_weight_grad(dw2, 3), w7.grad, 2e-5, "mapped back to the 7x7 / stride 2 taps")


def test_b16_conversions_and_elementwise(dev):
    """csrc/b16.hip against torch on the unpacked values: f32 <-> B16 round trip (B16 = bf16 [N][C/16][H][W][16]); the
    BasicBlock tail, its backward with the BatchNorm-backward sums, the folded BatchNorm backward; results are the bf16
    rounding of the fp32 formula evaluated on the bf16-rounded inputs."""
    K = _pkg().kernels
    g = torch.Generator().manual_seed(5)
    N, C, H, W = 3, 48, 9, 14
    r16 = lambda z: z.to(torch.bfloat16).float()            # noqa: E731
    x = torch.randn(N, C, H, W, generator=g)
    X = K.to_b16(x.to(dev))
    assert X.dtype == torch.bfloat16 and tuple(X.shape) == (N, C // 16, H, W, 16) and K.dims(X) == (N, C, H, W)
    assert torch.equal(X.cpu().float(), r16(x).view(N, C // 16, 16, H, W).permute(0, 1, 3, 4, 2)), "blocked layout"
    back = K.to_f32(K.to_b16(x.to(dev)).clone())
    assert torch.equal(back.cpu(), r16(x)), "f32 -> B16 -> f32"
    y, res, dz, dz2, add = (torch.randn(N, C, H, W, generator=g) for _ in range(5))
    sc, sh, rs, rh = (torch.randn(C, generator=g) for _ in range(4))
    mean, inv = torch.randn(C, generator=g) * 0.2, torch.rand(C, generator=g) + 0.5
    B = lambda z: K.to_b16(z.to(dev)).clone()               # noqa: E731
    v4 = lambda z: z.view(1, -1, 1, 1)                      # noqa: E731
    for act, neg in ((1, 0.0), (2, 0.2), (0, 1.0)):
        for use_res in (False, True):
            pre = r16(y) * v4(sc) + v4(sh) + ((r16(res) * v4(rs) + v4(rh)) if use_res else 0)
            z_ref = torch.where(pre > 0, pre, neg * pre)
            z = K.affine_act(B(y), sc.to(dev), sh.to(dev), B(res) if use_res else None, act,
                             rs.to(dev) if use_res else None, rh.to(dev) if use_res else None)
            assert K.is_b16(z)
            assert_close(K.to_f32(z), r16(z_ref), 4e-3, f"b16 affine_act act={act}")
            gref = torch.where(pre > 0, 1.0, neg) * (r16(dz) + r16(dz2)) + r16(add)
            bst = K.zeros_stats(C, X)
            out = K.affine_act_bwd_(B(dz), B(y), sc.to(dev), sh.to(dev), B(res) if use_res else None, B(add), mean.to(dev),
                                    inv.to(dev), act, bst, res_scale=rs.to(dev) if use_res else None,
                                    res_shift=rh.to(dev) if use_res else None, dz2=B(dz2))
            assert_close(K.to_f32(out), r16(gref), 4e-3, f"b16 affine_act_bwd act={act}")
            xhat = (r16(y) - v4(mean)) * v4(inv)
            st_ref = torch.cat([gref.double().sum((0, 2, 3)), (gref.double() * xhat.double()).sum((0, 2, 3))])
            assert_close(bst, st_ref, 1e-4, "b16 BatchNorm-backward sums (taken on the fp32 values before rounding)")
    pqr = torch.randn(3, C, generator=g)
    fused = K.bn_bwd_apply_(dz.to(dev), y.to(dev), pqr.to(dev), to_b16_out=True)       # fp32 in, B16 out in one pass
    assert K.is_b16(fused)
    assert_close(K.to_f32(fused), r16(v4(pqr[0]) * dz + v4(pqr[1]) * y + v4(pqr[2])), 4e-3, "fp32 bn_bwd_apply written as B16")
    o = K.bn_bwd_apply_(B(dz), B(y), pqr.to(dev), fresh=True)
    assert_close(K.to_f32(o), r16(v4(pqr[0]) * r16(dz) + v4(pqr[1]) * r16(y) + v4(pqr[2])), 4e-3, "b16 bn_bwd_apply")
    bst = K.zeros_stats(C, X)
    same = B(dz)
    ret = K.affine_act_bwd_(same, B(y), None, None, None, None, mean.to(dev), inv.to(dev), 0, bst, stats_only=True)
    assert ret is same
    st_ref = torch.cat([r16(dz).double().sum((0, 2, 3)), (r16(dz).double() * ((r16(y) - v4(mean)) * v4(inv)).double()).sum((0, 2, 3))])
    assert_close(bst, st_ref, 1e-4, "statistics-only pass")


@pytest.mark.parametrize("N,C0,C1,H,W", [(2, 32, 16, 9, 20), (3, 16, 48, 16, 16), (1, 16, 16, 33, 7)])
def test_b16_relu_up2x_and_stem_tail(dev, N, C0, C1, H, W):
    """The U-Net decoder glue and the stem tail on B16 images against the fp32 kernels of the same library run on the
    bf16-rounded inputs: up2x(relu(affine(cat))) and its adjoint (both source gradients, accumulation, BatchNorm-backward
    sums), MaxPool(3,2,1) over relu(bn(y)) with byte tap indices and the fused pool / ReLU / BatchNorm backward."""
    K = _pkg().kernels
    g = torch.Generator().manual_seed(N * 7 + H)
    r16 = lambda z: z.to(torch.bfloat16).float()            # noqa: E731
    x0, x1 = torch.randn(N, C0, H, W, generator=g), torch.randn(N, C1, H, W, generator=g)
    sc0, sh0 = torch.rand(C0, generator=g) + 0.5, torch.randn(C0, generator=g) * 0.3
    sc1, sh1 = torch.rand(C1, generator=g) + 0.5, torch.randn(C1, generator=g) * 0.3
    d = lambda z: z.to(dev)                                 # noqa: E731
    cat16 = K.Cat(K.to_b16(d(x0)).clone(), K.to_b16(d(x1)).clone(), sc0=d(sc0), sh0=d(sh0), sc1=d(sc1), sh1=d(sh1))
    cat32 = K.Cat(d(r16(x0)), d(r16(x1)), sc0=d(sc0), sh0=d(sh0), sc1=d(sc1), sh1=d(sh1))
    assert cat16.b16 and not cat32.b16
    U16, U32 = cat16.fwd(), cat32.fwd()
    assert K.is_b16(U16)
    assert_close(K.to_f32(U16), U32, 4e-3, "b16 relu_up2x forward")
    dU = torch.randn(N, C0 + C1, 2 * H, 2 * W, generator=g)
    mean1, inv1 = d(torch.randn(C1, generator=g) * 0.1), d(torch.rand(C1, generator=g) + 0.5)
    b16s, b32s = K.zeros_stats(C1, d(x0)), K.zeros_stats(C1, d(x0))
    g0, g1 = cat16.bwd(K.to_b16(d(dU)).clone(), mean1=mean1, invstd1=inv1, bstats1=b16s)
    h0, h1 = cat32.bwd(d(r16(dU)), mean1=mean1, invstd1=inv1, bstats1=b32s)
    assert_close(K.to_f32(g0), h0, 4e-3, "b16 relu_up2x backward g0")
    assert_close(K.to_f32(g1), h1, 4e-3, "b16 relu_up2x backward g1")
    assert_close(b16s, b32s, 1e-4, "b16 relu_up2x backward sums")
    base = torch.randn(N, C0, H, W, generator=g)
    acc, _ = cat16.bwd(K.to_b16(d(dU)).clone(), g0_acc=K.to_b16(d(base)).clone())
    assert_close(K.to_f32(acc), r16(base).to(dev) + h0, 8e-3, "b16 relu_up2x backward accumulate")
    # stem tail
    C = C0
    y = torch.randn(N, C, H, W, generator=g)
    rows = torch.stack([torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3, torch.randn(C, generator=g) * 0.1,
                        torch.rand(C, generator=g) + 0.5]).to(dev)
    Y16 = K.to_b16(d(y)).clone()
    z16, idx16 = K.maxpool3x3s2(Y16, rows[0], rows[1], 1)
    z32, idx32 = K.maxpool3x3s2(d(r16(y)), rows[0], rows[1], 1)
    assert idx16.dtype == torch.uint8
    assert_close(K.to_f32(z16), z32, 4e-3, "b16 maxpool forward")
    Ho, Wo = z32.shape[2:]
    gp, gp2 = torch.randn(N, C, Ho, Wo, generator=g), torch.randn(N, C, Ho, Wo, generator=g)
    gsum = (r16(gp) + r16(gp2)).to(dev)
    s16, s32 = K.zeros_stats(C, d(y)), K.zeros_stats(C, d(y))
    K.maxpool_bn_relu_bwd_stats(K.to_b16(d(gp)).clone(), idx16, Y16, rows, s16, g2=K.to_b16(d(gp2)).clone())
    K.maxpool_bn_relu_bwd_stats(gsum, idx32, d(r16(y)), rows, s32)
    assert_close(s16, s32, 1e-4, "b16 stem-tail backward sums")
    pqr = torch.randn(3, C, generator=g).to(dev)
    for out_f32 in (False, True):
        dy16 = K.maxpool_bn_relu_bwd_apply(K.to_b16(d(gp)).clone(), idx16, Y16, rows, pqr, g2=K.to_b16(d(gp2)).clone(), out_f32=out_f32)
        dy32 = K.maxpool_bn_relu_bwd_apply(gsum, idx32, d(r16(y)), rows, pqr)
        assert (dy16.dtype == torch.float32) == out_f32
        assert_close(K.to_f32(dy16), dy32, 4e-3 if not out_f32 else 2e-5, f"b16 stem-tail backward apply (fp32 out {out_f32})")
    # space-to-depth straight into a one-block B16 image
    fr = torch.randn(2, 3, 12, 20, generator=g)
    s2d16, s2d32 = K.space_to_depth2(d(fr), b16=True), K.space_to_depth2(d(fr), b16=False)
    assert K.is_b16(s2d16) and torch.equal(K.to_f32(s2d16), s2d32.to(torch.bfloat16).float())


@pytest.mark.parametrize("N,family,wfamily", [(32, "wino_kernel", "winow_kernel"), (64, "wino4_kernel", "winow4_kernel")])
def test_conv_virtual_input_winograd(dev, N, family, wfamily):
    """the same folded two-source input (C0 == C1) through the Winograd kernels' staging (conv_wino.hip / wgrad_wino.hip with
    the F(4x4) kernels masked out, conv_wino4.hip / wgrad_wino4.hip otherwise), with the BatchNorm sums of the result."""
    K = _pkg().kernels
    K.set_algo_mask(*(("winograd4",) if family == "wino_kernel" else ()))
    try:
        _virtual_input_winograd_case(dev, K, N, family, wfamily)
    finally:
        K.set_algo_mask()


def _virtual_input_winograd_case(dev, K, N, family, wfamily):
    g = torch.Generator().manual_seed(11)
    C0, C1, Cout, H, W = 64, 64, 72, 32, 32
    x0, x1 = torch.randn(N, C0, H, W, generator=g), torch.randn(N, C1, H, W, generator=g)
    sc0, sh0 = torch.rand(C0, generator=g) + 0.5, torch.randn(C0, generator=g)
    sc1, sh1 = torch.rand(C1, generator=g) + 0.5, torch.randn(C1, generator=g)
    w = torch.randn(Cout, C0 + C1, 3, 3, generator=g) * 0.05
    a0 = F.leaky_relu(x0 * sc0.view(1, -1, 1, 1) + sh0.view(1, -1, 1, 1), 0.2)
    a1 = F.relu(x1 * sc1.view(1, -1, 1, 1) + sh1.view(1, -1, 1, 1))
    wr = w.clone().requires_grad_(True)
    y_ref = F.conv2d(torch.cat([a0, a1], 1), wr, None, 1, 1)
    dy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(dy)
    y_ref = y_ref.detach()
    t = lambda z: z.to(dev)
    cv = K.Conv(t(x0), Cout, 3, 1, 1, x1=t(x1), sc0=t(sc0), sh0=t(sh0), act0=2, sc1=t(sc1), sh1=t(sh1), act1=1)
    assert cv.kernel_name("fwd", True) == family and cv.kernel_name("wgrad") == wfamily
    assert_close(cv.wgrad(t(dy))[0], wr.grad, 2e-5, "wgrad")
    st = K.zeros_stats(Cout, cv.like)
    assert_close(cv.fwd(cv.pack(t(w), 0), None, st), y_ref, 2e-5, "fwd")
    st_ref = torch.cat([y_ref.double().sum((0, 2, 3)), (y_ref.double() ** 2).sum((0, 2, 3))])
    assert_close(st, st_ref, 1e-5, "stats")
    # one source, activation only (identity affine rows)
    x01 = torch.cat([x0, x1], 1)
    # 4x4 / stride 2 over a folded BatchNorm + LeakyReLU input: weight gradient on wgrad4d_kernel
    sc, sh = torch.cat([sc0, sc1]), torch.cat([sh0, sh1])
    w4 = (torch.randn(Cout, C0 + C1, 4, 4, generator=g) * 0.05).requires_grad_(True)
    y4 = F.conv2d(F.leaky_relu(x01 * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1), 0.2), w4, None, 2, 1)
    dy4 = torch.randn(y4.shape, generator=g)
    y4.backward(dy4)
    cv4 = K.Conv(t(x01), Cout, 4, 2, 1, sc0=t(sc), sh0=t(sh), act0=2)
    assert cv4.kernel_name("wgrad") == "wgrad4d_kernel"
    assert_close(cv4.wgrad(t(dy4))[0], w4.grad, 2e-5, "wgrad 4x4/s2")
    cv1 = K.Conv(t(x01), Cout, 3, 1, 1, act0=1)
    assert cv1.kernel_name("fwd", False) == family
    assert_close(cv1.fwd(cv1.pack(t(w), 0)), F.conv2d(F.relu(x01), w, None, 1, 1), 2e-5, "fwd relu")


@pytest.mark.parametrize("family", ["wino_kernel", "wino4_kernel"])
def test_conv_dilated_folded_input_winograd(dev, family):
    """dilation 2 over a folded BatchNorm + ReLU input (the ResNet layer3/4 form): the parity-sub-image instantiations of the
    Winograd forward, data-gradient (F(2x2) and F(4x4)) and weight-gradient kernels with the affine staging path, and the
    BatchNorm sums."""
    K = _pkg().kernels
    K.set_algo_mask(*(("winograd4",) if family == "wino_kernel" else ()))
    try:
        _dilated_folded_case(dev, K, family)
    finally:
        K.set_algo_mask()


def _dilated_folded_case(dev, K, family):
    g = torch.Generator().manual_seed(12)
    N, Cin, Cout, H, W = 260, 128, 72, 14, 14
    x = torch.randn(N, Cin, H, W, generator=g)
    sc, sh = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05).requires_grad_(True)
    v = F.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).requires_grad_(True)
    y_ref = F.conv2d(v, w, None, 1, 2, 2)
    dy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(dy)
    t = lambda z: z.to(dev)
    cv = K.Conv(t(x), Cout, 3, 1, 2, 2, sc0=t(sc), sh0=t(sh), act0=1)
    assert (cv.kernel_name("fwd", True), cv.kernel_name("dgrad"), cv.kernel_name("wgrad")) == (family, family, "winow_kernel" if family == "wino_kernel" else "winow4_kernel")
    st = K.zeros_stats(Cout, cv.like)
    yd = y_ref.detach()
    assert_close(cv.fwd(cv.pack(t(w.detach()), 0), None, st), yd, 2e-5, "fwd")
    assert_close(st, torch.cat([yd.double().sum((0, 2, 3)), (yd.double() ** 2).sum((0, 2, 3))]), 1e-5, "stats")
    assert_close(cv.dgrad(cv.pack(t(w.detach()), 1), t(dy)), v.grad, 2e-5, "dgrad")
    assert_close(cv.wgrad(t(dy))[0], w.grad, 2e-5, "wgrad")


@pytest.mark.parametrize("H,W", [(12, 16), (10, 14), (9, 11), (64, 112)])   # pixel quads / pixel pairs / single pixels / the stem's width
def test_stem_tail_backward_apply(dev, H, W):
    """max-pool 3x3/s2 backward + ReLU mask + folded BatchNorm backward in one pass over the stem's conv output
    (vision_net.py:111-117 children 1-3, backward): the three kernel forms of avsep_maxpool_bn_relu_bwd_apply."""
    K = _pkg().kernels
    g = torch.Generator().manual_seed(41)
    N, C = 3, 5
    y = torch.randn(N, C, H, W, generator=g)
    sc, sh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    a = F.relu(y * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    z_ref, ind = F.max_pool2d(a, 3, 2, 1, return_indices=True)
    t = lambda v: v.to(dev)
    rows = torch.stack([sc, sh, torch.zeros(C), torch.ones(C)]).to(dev)
    z, idx = K.maxpool3x3s2(t(y), rows[0], rows[1], 1)
    assert_close(z, z_ref, 1e-6, "maxpool forward")
    assert torch.equal(idx.cpu().long()[z_ref > 0], ind[z_ref > 0])          # (an all-zero window has no unique winner)
    gp = torch.randn(z_ref.shape, generator=g)
    pqr = torch.randn(3, C, generator=g)
    da = torch.zeros(N, C, H * W).scatter_add_(2, ind.reshape(N, C, -1), (gp * (z_ref > 0)).reshape(N, C, -1)).reshape(N, C, H, W)
    v = lambda k: pqr[k].view(1, -1, 1, 1)
    dy_ref = v(0) * (da * (a > 0)) + v(1) * y + v(2)
    dy = K.maxpool_bn_relu_bwd_apply(t(gp), idx, t(y), rows, t(pqr))
    assert_close(dy, dy_ref, 1e-6, "stem-tail backward apply")


@pytest.mark.parametrize("N,C,H,W,dil,fused", [(40, 72, 56, 56, 1, True),     # 16-byte stores (W % 4 == 0)
                                                  (400, 72, 14, 14, 1, True),    # masked scalar stores, partial tiles
                                                  (260, 72, 14, 14, 2, True),    # parity sub-images (stride-2 stores)
                                                  (72, 72, 20, 36, 1, True),     # ragged group grid
                                                  (2, 24, 10, 12, 1, False)])    # a family without the epilogue: two launches
def test_conv_dgrad_through_activation(dev, N, C, H, W, dil, fused):
    """avsep_conv2d_dgrad_act: act'(sc*y + sh [+ rs*res + rh]) * (dgrad(dy) [+ dz2]) [+ add] and the BatchNorm-backward sums,
    in the F(4x4) data-gradient kernel's epilogue (`fused`) or as the two launches it stands for — the ReLU'/BatchNorm-sum
    passes of a ResNet BasicBlock (vision_net.py:84-109; torchvision BasicBlock.forward)."""
    K = _pkg().kernels
    g = torch.Generator().manual_seed(31)
    Cout = 72
    w = torch.randn(Cout, C, 3, 3, generator=g) * 0.05
    dy = torch.randn(N, Cout, H, W, generator=g)
    x = torch.zeros(N, C, H, W).requires_grad_(True)
    F.conv2d(x, w, None, 1, dil, dil).backward(dy)
    dxr = x.grad.double()
    t = lambda z: None if z is None else z.to(dev)
    cv = K.Conv(t(x.detach()), Cout, 3, 1, dil, dil)
    assert cv.dgrad_act_fused() == fused
    assert cv.kernel_name("dgrad") == ("wino4_kernel" if fused else cv.kernel_name("dgrad"))
    wp = cv.pack(t(w), 1)
    sc, sh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    rs, rh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    mean, invstd = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    res, dz2, add = (torch.randn(N, C, H, W, generator=g) for _ in range(3))
    v = lambda r: r.view(1, -1, 1, 1).double()
    # pre-activations at least 1e-2 away from zero, then y from them: both sides take the same branch everywhere
    pre = torch.randn(N, C, H, W, generator=g).double()
    pre = torch.where(pre.abs() < 1e-2, torch.full_like(pre, 1e-2), pre)
    for act, slope, use_res, use_rs, use_dz2, use_add in ((1, 0.0, True, True, True, False), (1, 0.0, True, False, True, True),
                                                          (2, 0.2, False, False, False, False), (0, 1.0, False, False, False, True)):
        rterm = (v(rs) * res.double() + v(rh) if use_rs else res.double()) if use_res else 0.0
        y = ((pre - v(sh) - rterm) / v(sc)).float()
        gfac = torch.where(pre > 0, torch.ones_like(pre), torch.full_like(pre, slope))
        gref = gfac * (dxr + (dz2.double() if use_dz2 else 0.0)) + (add.double() if use_add else 0.0)
        sref = torch.cat([gref.sum((0, 2, 3)), (gref * (y.double() - v(mean)) * v(invstd)).sum((0, 2, 3))])
        bst = K.zeros_stats(C, cv.like)
        out = cv.dgrad_act(wp, t(dy), t(y), t(sc), t(sh), t(mean), t(invstd), act, bst, residual=t(res) if use_res else None,
                           res_scale=t(rs) if use_rs else None, res_shift=t(rh) if use_rs else None,
                           dz2=t(dz2) if use_dz2 else None, add=t(add) if use_add else None)
        tag = f"act {act} res {use_res}/{use_rs} dz2 {use_dz2} add {use_add}"
        assert_close(out, gref, 2e-5, "dgrad_act " + tag)
        assert_close(bst, sref, 2e-5, "dgrad_act sums " + tag)
        # the two launches the call stands for
        bst2 = K.zeros_stats(C, cv.like)
        two = K.affine_act_bwd_(cv.dgrad(wp, t(dy), out_b16=False), t(y), t(sc), t(sh), t(res) if use_res else None,
                                t(add) if use_add else None, t(mean), t(invstd), act, bst2, res_scale=t(rs) if use_rs else None,
                                res_shift=t(rh) if use_rs else None, dz2=t(dz2) if use_dz2 else None)
        assert_close(out, two, 2e-6, "dgrad_act against dgrad + affine_act_bwd " + tag)
        assert_close(bst, bst2, 2e-6, "dgrad_act sums against dgrad + affine_act_bwd " + tag)


@pytest.mark.parametrize("up2x,H,W", [(False, 9, 7), (True, 9, 7), (False, 10, 36), (True, 10, 18), (True, 6, 10)])
def test_conv_virtual_input(dev, up2x, H, W):
    """two-source concat + per-channel affine + LeakyReLU/ReLU (+ bilinear x2) folded into the gather;
    the larger sizes go through the 3x3 halo-patch kernel, the small ones through the im2col gather."""
    K = _pkg().kernels
    g = torch.Generator().manual_seed(5)
    N, C0, C1, Cout = 2, 24, 40, 36
    x0, x1 = torch.randn(N, C0, H, W, generator=g), torch.randn(N, C1, H, W, generator=g)
    sc0, sh0 = torch.rand(C0, generator=g) + 0.5, torch.randn(C0, generator=g)
    sc1, sh1 = torch.rand(C1, generator=g) + 0.5, torch.randn(C1, generator=g)
    w = torch.randn(Cout, C0 + C1, 3, 3, generator=g) * 0.05
    a0 = F.leaky_relu(x0 * sc0.view(1, -1, 1, 1) + sh0.view(1, -1, 1, 1), 0.2)
    a1 = F.relu(x1 * sc1.view(1, -1, 1, 1) + sh1.view(1, -1, 1, 1))
    v = torch.cat([a0, a1], 1)
    if up2x:
        v = F.interpolate(v, scale_factor=2, mode="bilinear", align_corners=True)
    v = v.requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    y_ref = F.conv2d(v, wr, None, 1, 1)
    dy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(dy)
    t = lambda z: z.to(dev)
    cv = K.Conv(t(x0), Cout, 3, 1, 1, x1=t(x1), sc0=t(sc0), sh0=t(sh0), act0=2, sc1=t(sc1), sh1=t(sh1), act1=1,
                up2x=up2x)
    y = cv.fwd(cv.pack(t(w), 0))
    assert_close(y, y_ref, 2e-5, "fwd")
    assert_close(cv.dgrad(cv.pack(t(w), 1), t(dy)), v.grad, 2e-5, "dgrad")
    assert_close(cv.wgrad(t(dy))[0], wr.grad, 2e-5, "wgrad")


def test_bn_pieces(dev):
    K = _pkg().kernels
    g = torch.Generator().manual_seed(6)
    N, C, H, W = 3, 10, 12, 9
    y = torch.randn(N, C, H, W, generator=g) * 2 + 1
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    bn = torch.nn.BatchNorm2d(C)
    bn.weight.data, bn.bias.data = gamma.clone(), beta.clone()
    yr = y.clone().requires_grad_(True)
    z_ref = bn(yr)
    dz = torch.randn(z_ref.shape, generator=g)
    z_ref.backward(dz)
    yd = y.to(dev)
    st = K.zeros_stats(C, yd)
    K.channel_stats(yd, st)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    rows = K.bn_finalize(st, N * H * W, gamma.to(dev), beta.to(dev), rm, rv, 0.1, 1e-5, True, yd)
    z = K.affine_act(yd, rows[0], rows[1], None, 0)
    assert_close(z, z_ref, 1e-5, "bn fwd")
    assert_close(rm, bn.running_mean, 1e-5, "running_mean")
    assert_close(rv, bn.running_var, 1e-5, "running_var")
    dzd = dz.to(dev).clone()
    bst = K.zeros_stats(C, yd)
    K.affine_act_bwd_(dzd, yd, rows[0], rows[1], None, None, rows[2], rows[3], 0, bst)
    dgamma, dbeta, pqr = K.bn_bwd_coeffs(bst, N * H * W, gamma.to(dev), rows[2], rows[3])
    dy = K.bn_bwd_apply_(dzd, yd, pqr)
    assert_close(dgamma, bn.weight.grad, 2e-5, "dgamma")
    assert_close(dbeta, bn.bias.grad, 2e-5, "dbeta")
    assert_close(dy, yr.grad, 5e-5, "bn dx")


@pytest.mark.parametrize("bcast,H,W", [(False, 4, 3), (True, 4, 3), (False, 20, 12), (False, 5, 4), (False, 13, 8),
                                       (False, 16, 32), (False, 9, 128), (False, 64, 64)])
def test_relu_up2x(dev, bcast, H, W):
    """W = 4..128 powers of two take the static-pair kernels (pair forward, row-sweep backward), the rest the
    gather / LDS-tiled kernels; ragged H covers partial row chunks."""
    K = _pkg().kernels
    g = torch.Generator().manual_seed(7)
    N, C0, C1 = 2, 6, 5
    x0 = torch.randn(N, C0, generator=g) if bcast else torch.randn(N, C0, H, W, generator=g)
    x1 = torch.randn(N, C1, H, W, generator=g)
    sc1, sh1 = torch.rand(C1, generator=g) + 0.5, torch.randn(C1, generator=g)
    a0 = (x0.view(N, C0, 1, 1).expand(N, C0, H, W) if bcast else x0).clone().requires_grad_(True)
    a1 = (x1 * sc1.view(1, -1, 1, 1) + sh1.view(1, -1, 1, 1)).requires_grad_(True)
    out_ref = F.interpolate(F.relu(torch.cat([a0, a1], 1)), scale_factor=2, mode="bilinear", align_corners=True)
    dout = torch.randn(out_ref.shape, generator=g)
    out_ref.backward(dout)
    t = lambda z: z.to(dev)
    cat = K.Cat(t(x0), t(x1), sc1=t(sc1), sh1=t(sh1), bcast0=bcast)
    # the source index r*o is a float product: at o ~ 255 its rounding (ulp 7.6e-6) shows up as ~3e-6 of the output
    assert_close(cat.fwd(), out_ref, 5e-6, "fwd")
    mean1, invstd1 = torch.randn(C1, generator=g), torch.rand(C1, generator=g) + 0.5
    bst = K.zeros_stats(C1, t(x1))
    g0, g1 = cat.bwd(t(dout), mean1=t(mean1), invstd1=t(invstd1), bstats1=bst)
    assert_close(g0, a0.grad.sum((2, 3)) if bcast else a0.grad, 1e-5, "g0")
    assert_close(g1, a1.grad, 1e-5, "g1")
    if not bcast:   # accumulate into an existing source-0 gradient
        base = torch.randn(N, C0, H, W, generator=g)
        acc, _ = cat.bwd(t(dout), g0_acc=t(base).clone())
        assert_close(acc, base + a0.grad, 1e-5, "g0 accumulate")
    xhat = (x1 - mean1.view(1, -1, 1, 1)) * invstd1.view(1, -1, 1, 1)
    ref = torch.cat([a1.grad.double().sum((0, 2, 3)), (a1.grad.double() * xhat).sum((0, 2, 3))])
    assert_close(bst, ref, 1e-5, "bstats")


def test_relu_up2x_many_planes(dev):
    """More (image, channel) planes than one grid dimension holds (the batch-64 step's [64, 1024, 16, 16] level has 65536):
    the static-pair forward and the row-sweep backward spread the planes over grid (y, z)."""
    K = _pkg().kernels
    g = torch.Generator().manual_seed(8)
    N, C0, C1, H, W = 3, 6000, 5001, 16, 4                      # 33003 planes > 32768 per grid.y slice, ragged last z slice
    x0, x1 = torch.randn(N, C0, H, W, generator=g), torch.randn(N, C1, H, W, generator=g)
    a0, a1 = x0.clone().requires_grad_(True), x1.clone().requires_grad_(True)
    out_ref = F.interpolate(F.relu(torch.cat([a0, a1], 1)), scale_factor=2, mode="bilinear", align_corners=True)
    dout = torch.randn(out_ref.shape, generator=g)
    out_ref.backward(dout)
    cat = K.Cat(x0.to(dev), x1.to(dev))
    assert_close(cat.fwd(), out_ref, 5e-6, "fwd")
    g0, g1 = cat.bwd(dout.to(dev))
    assert_close(g0, a0.grad, 1e-5, "g0")
    assert_close(g1, a1.grad, 1e-5, "g1")


def test_prepare_golden(dev, golden):
    K = _pkg().kernels
    G = golden("prepare")
    mags = torch.stack([G["mags0"], G["mags1"]], 0).to(dev)
    for tag, kw in [("bin_w", dict(binary=1, weighted=1, log_freq=1)), ("ratio_now", dict(binary=0, weighted=0, log_freq=1)),
                    ("nolog", dict(binary=1, weighted=1, log_freq=0))]:
        mix_w, mags_w, logm, weight, gt = K.prepare(G["mag_mix"].to(dev), mags, kw["log_freq"], kw["weighted"], kw["binary"])
        assert_close(mix_w, G[f"{tag}.mag_mix"], 1e-5, tag + " mag_mix")
        assert_close(logm, G[f"{tag}.log_mag_mix"], 1e-5, tag + " log")
        assert_close(weight, G[f"{tag}.weights"], 1e-5, tag + " weight")
        for n in range(2):
            assert_close(mags_w[n], G[f"{tag}.mags{n}"], 1e-5, tag + " mags")
            ref = G[f"{tag}.gt_masks{n}"]
            if kw["binary"]:   # a mask bit may flip where mags == 0.5*mix to the last ulp
                frac = (gt[n].cpu() != ref).float().mean().item()
                assert frac <= 1e-4, f"{tag} gt{n}: {frac} of the mask bits differ"
            else:
                assert_close(gt[n], ref, 1e-4, tag + " gt ratio")
    # unwarp (main.py:216-220) against torch grid_sample on the reference grid
    import numpy as np
    x = torch.rand(2, 1, 256, 24)
    ref = F.grid_sample(x, torch.from_numpy(_warpgrid(2, 512, 24, False)), align_corners=False)
    assert_close(K.warp(x.to(dev), 512, 24, 0), ref, 1e-5, "unwarp")


def _warpgrid(bs, HO, WO, warp):
    import numpy as np
    x, y = np.linspace(-1, 1, WO), np.linspace(-1, 1, HO)
    xv, yv = np.meshgrid(x, y)
    gy = (np.power(21, (yv + 1) / 2) - 11) / 10 if warp else np.log(yv * 10 + 11) / np.log(21) * 2 - 1
    grid = np.zeros((bs, HO, WO, 2))
    grid[..., 0], grid[..., 1] = xv, gy
    return grid.astype(np.float32)


def test_mask_loss_and_pit_golden(dev, golden):
    M = _pkg().models
    G = golden("criterion")
    t = lambda z: z.to(dev)
    preds, tg, w = [t(G["p0"]), t(G["p1"])], [t(G["t0"]), t(G["t1"])], t(G["w"])
    for kind, cls in (("bce", M.BCELoss), ("l1", M.L1Loss), ("l2", M.L2Loss)):
        assert_close(cls()(preds, tg, w), G[f"{kind}.list"], 1e-5, kind + " list")
        assert_close(cls()(preds[0], tg[0]), G[f"{kind}.tensor_now"], 1e-5, kind + " tensor")
    pit = M.PitWrapper(F.binary_cross_entropy)
    loss, perms = pit(t(G["pit.P"]), t(G["pit.T"]), t(G["pit.W"]))
    assert_close(loss, G["pit.loss"], 1e-5, "pit loss")
    assert [tuple(p) for p in perms] == [tuple(p) for p in G["pit.perms"].tolist()]
    assert torch.equal(pit.reorder_tensor(t(G["pit.P"]), perms).cpu(), G["pit.reordered"])
    # gradient of the fused sigmoid+BCE path against autograd
    g = torch.Generator().manual_seed(3)
    logits = torch.randn(2, 2, 6, 5, generator=g) * 3
    gt = (torch.rand(2, 2, 1, 6, 5, generator=g) > 0.5).float()
    wt = torch.rand(2, 1, 6, 5, generator=g)
    lr = logits.clone().requires_grad_(True)
    ref = sum(F.binary_cross_entropy(torch.sigmoid(lr[:, n:n + 1]), gt[n], weight=wt) for n in range(2)) / 2
    ref.backward()
    from avsep_amd.models.criterion import mask_loss
    ld = logits.to(dev).requires_grad_(True)
    pred, sums, FT = mask_loss(ld, gt.to(dev), wt.to(dev), 3, "bce")
    err = torch.diagonal(sums, dim1=1, dim2=2).sum() / (2 * 2 * FT)
    err.backward()
    assert_close(err, ref, 1e-5, "fused bce")
    assert_close(ld.grad, lr.grad, 1e-5, "fused bce grad")
    assert_close(pred, torch.sigmoid(logits), 1e-6, "pred")


def test_sgd_matches_torch(dev):
    K = _pkg().kernels
    g = torch.Generator().manual_seed(8)
    p = torch.randn(1000, generator=g)
    ref = p.clone().requires_grad_(True)
    opt = torch.optim.SGD([ref], lr=0.01, momentum=0.9, weight_decay=1e-4)
    pd, buf = p.to(dev), torch.zeros(1000, device=dev)
    for it in range(3):
        gr = torch.randn(1000, generator=g)
        ref.grad = gr.clone()
        opt.step()
        K.sgd_momentum_(pd, gr.to(dev), buf, 0.01, 0.9, 1e-4, 1.0, it == 0)
    assert_close(pd, ref, 1e-6, "sgd")


def test_stft_against_numpy(dev):
    import numpy as np
    from oracle import stft as OS
    K = _pkg().kernels
    g = torch.Generator().manual_seed(9)
    wav = torch.randn(3, 65535, generator=g) * 0.3
    for mode in ("reflect", "constant"):
        plan = K.Stft(dev, 1022, 256, mode)
        mag, phase = plan.stft(wav.to(dev))
        assert mag.shape == (3, 512, 256)
        for r in range(3):
            m_ref, p_ref = OS.stft_mag_phase(wav[r].numpy(), 1022, 256, mode)
            # tolerance: fp32 DFT-as-GEMM vs numpy's pocketfft in fp64-cast-to-c64, relative to the peak magnitude
            assert_close(mag[r], torch.from_numpy(m_ref), 2e-5, "stft mag " + mode)
            spec = mag[r].cpu() * torch.exp(1j * phase[r].cpu())
            spec_ref = torch.from_numpy(m_ref * np.exp(1j * p_ref))
            assert (spec - spec_ref).abs().max() / spec_ref.abs().max() < 5e-5
        back = plan.istft(mag, phase)
        ref = torch.from_numpy(np.stack([OS.istft(OS.stft(wav[r].numpy(), 1022, 256, mode)) for r in range(3)]))
        assert_close(back, ref, 5e-5, "istft " + mode)
    # round trip: interior samples are reconstructed
    assert (back.cpu()[:, 1024:64000] - wav[:, 1024:64000]).abs().max() < 1e-3


@pytest.mark.parametrize("n_fft,hop,L,R", [(510, 128, 9000, 5), (254, 128, 8000, 2), (1022, 256, 12000, 1)])
def test_stft_other_geometries(dev, n_fft, hop, L, R):
    """n_fft within (3, 4] hops takes the 1x4-conv form on the halo-patch kernel (ragged row / frame tiles here),
    anything else the strided 1-D conv through the im2col kernel."""
    from oracle import stft as OS
    K = _pkg().kernels
    wav = torch.randn(R, L, generator=torch.Generator().manual_seed(L)) * 0.3
    plan = K.Stft(dev, n_fft, hop, "reflect")
    mag, phase = plan.stft(wav.to(dev))
    assert mag.shape == (R, n_fft // 2 + 1, 1 + L // hop)
    for r in range(R):
        m_ref, p_ref = OS.stft_mag_phase(wav[r].numpy(), n_fft, hop, "reflect")
        assert_close(mag[r], torch.from_numpy(m_ref), 2e-5, "stft mag")
        spec = mag[r].cpu() * torch.exp(1j * phase[r].cpu())
        spec_ref = torch.from_numpy(m_ref * np.exp(1j * p_ref))
        assert (spec - spec_ref).abs().max() / spec_ref.abs().max() < 5e-5


def test_pool_and_mean(dev):
    K = _pkg().kernels
    g = torch.Generator().manual_seed(10)
    x = torch.randn(3, 5, 11, 14, generator=g)
    xr = x.clone().requires_grad_(True)
    y_ref = F.max_pool2d(xr, 3, 2, 1)
    dy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(dy)
    y, idx = K.maxpool3x3s2(x.to(dev))
    assert_close(y, y_ref, 0, "maxpool")
    assert_close(K.maxpool3x3s2_bwd(dy.to(dev), idx, 11, 14), xr.grad, 1e-6, "maxpool bwd")
    f = torch.randn(6, 4, 3, 3, generator=g)
    assert_close(K.temporal_mean(f.to(dev), 2, 3), f.view(2, 3, 4, 3, 3).mean(1), 1e-6, "tmean")
    d = torch.randn(2, 4, 3, 3, generator=g)
    assert_close(K.temporal_mean_bwd(d.to(dev), 2, 3), (d / 3).repeat_interleave(3, 0), 1e-6, "tmean bwd")


@pytest.mark.parametrize("N,C0,C1,Hl,Wl,Cout,affine", [(2, 8, 8, 32, 32, 2, True), (2, 5, 3, 24, 20, 2, True),
                                                        (1, 4, 4, 128, 128, 2, False), (2, 6, 2, 16, 64, 1, True),
                                                        (2, 3, 5, 8, 16, 3, True), (1, 4, 4, 12, 8, 4, False),
                                                        (2, 8, 0, 16, 16, 2, True)])
def test_fused_decoder_head_matches_unfused(dev, N, C0, C1, Hl, Wl, Cout, affine):
    """csrc/head.hip (fwd / wgrad / dgrad straight to the low-res sources) against the unfused launch sequence
    relu_up2x_fwd -> conv3x3 -> (dgrad -> relu_up2x_bwd), and against torch on the CPU for the forward."""
    import avsep_amd  # noqa: F401
    from avsep_amd import kernels as K
    from avsep_amd.lib import ACT_RELU
    g = torch.Generator().manual_seed(N * 100 + C0 * 10 + Cout)
    x0 = torch.randn(N, C0, Hl, Wl, generator=g)
    x1 = torch.randn(N, C1, Hl, Wl, generator=g) if C1 else None
    w = torch.randn(Cout, C0 + C1, 3, 3, generator=g) * 0.2
    b = torch.randn(Cout, generator=g)
    dy = torch.randn(N, Cout, 2 * Hl, 2 * Wl, generator=g)
    sc0 = sh0 = sc1 = sh1 = None
    if affine:
        sc0, sh0 = torch.rand(C0, generator=g) + 0.5, torch.randn(C0, generator=g) * 0.3
        if C1:
            sc1, sh1 = torch.rand(C1, generator=g) + 0.5, torch.randn(C1, generator=g) * 0.3
    d = lambda t: None if t is None else t.to(dev)   # noqa: E731
    X0, X1, W_, B_, DY = d(x0), d(x1), d(w), d(b), d(dy)
    S0, H0, S1, H1 = d(sc0), d(sh0), d(sc1), d(sh1)
    cv = K.Conv(X0, Cout, 3, 1, 1, x1=X1, sc0=S0, sh0=H0, act0=ACT_RELU, sc1=S1, sh1=H1, act1=ACT_RELU, up2x=True)
    assert cv.head_applicable()
    y = cv.fwd(cv.pack(W_, 0), B_, None)
    # CPU reference of the forward
    a0 = x0 * sc0.view(1, -1, 1, 1) + sh0.view(1, -1, 1, 1) if affine else x0
    parts = [a0]
    if C1:
        parts.append(x1 * sc1.view(1, -1, 1, 1) + sh1.view(1, -1, 1, 1) if affine else x1)
    U = torch.nn.functional.interpolate(torch.relu(torch.cat(parts, 1)), scale_factor=2, mode="bilinear", align_corners=True)
    assert_close(y, torch.nn.functional.conv2d(U, w, b, padding=1), 2e-5, "fused head forward vs torch")
    # unfused sequence on the GPU
    cat = K.Cat(X0, X1, sc0=S0, sh0=H0, sc1=S1, sh1=H1)
    Um = cat.fwd()
    cu = K.Conv(Um, Cout, 3, 1, 1)
    assert_close(y, cu.fwd(cu.pack(W_, 0), B_, None), 2e-5, "forward vs unfused")
    dw, db = cv.wgrad(DY, want_bias=True)
    dwu, dbu = cu.wgrad(DY, want_bias=True)
    assert_close(dw, dwu, 5e-5, "dw")
    assert_close(db, dbu, 5e-5, "db")
    dU = cu.dgrad(cu.pack(W_, 1), DY)
    mean1 = invstd1 = None
    bst = bstu = None
    if C1:
        mean1, invstd1 = d(torch.randn(C1, generator=g) * 0.1), d(torch.rand(C1, generator=g) + 0.5)
        bst, bstu = K.zeros_stats(C1, X0), K.zeros_stats(C1, X0)
    g0u, g1u = cat.bwd(dU, mean1=mean1, invstd1=invstd1, bstats1=bstu)
    g0, g1 = cv.dgrad_up2x(W_, DY, mean1=mean1, invstd1=invstd1, bstats1=bst)
    assert_close(g0, g0u, 5e-5, "g0")
    if C1:
        assert_close(g1, g1u, 5e-5, "g1")
        assert_close(bst.float(), bstu.float(), 1e-4, "BatchNorm-backward sums")
    # accumulation into an existing source-0 gradient (second pass of the shared-encoder AV step)
    base = torch.randn(N, C0, Hl, Wl, generator=g).to(dev)
    acc, _ = cv.dgrad_up2x(W_, DY, g0_acc=base.clone())
    assert_close(acc, base + g0u, 5e-5, "g0 accumulate")



def test_decoder_head_true_channels_vs_torch_autograd(dev):
    """The decoder head at its TRUE operand widths (C0 = C1 = 64: the 128-channel contraction of audio_net.py:72-76 that the
    batch-64 step launches), forward, weight gradient and the data gradients wrt both low-res sources against torch-CPU
    autograd of  conv3x3(upsample_x2(relu(affine(cat)))) — the other head rows stop at 16 channels."""
    import avsep_amd  # noqa: F401
    from avsep_amd import kernels as K
    from avsep_amd.lib import ACT_RELU
    g = torch.Generator().manual_seed(4242)
    N, C0, C1, Hl, Wl, Cout = 2, 64, 64, 64, 32, 2
    x0 = torch.randn(N, C0, Hl, Wl, generator=g).requires_grad_(True)
    x1 = torch.randn(N, C1, Hl, Wl, generator=g).requires_grad_(True)
    w = (torch.randn(Cout, C0 + C1, 3, 3, generator=g) * 0.05).requires_grad_(True)
    b = torch.randn(Cout, generator=g).requires_grad_(True)
    dy = torch.randn(N, Cout, 2 * Hl, 2 * Wl, generator=g)
    sc0, sh0 = torch.rand(C0, generator=g) + 0.5, torch.randn(C0, generator=g) * 0.3
    sc1, sh1 = torch.rand(C1, generator=g) + 0.5, torch.randn(C1, generator=g) * 0.3
    a0 = x0 * sc0.view(1, -1, 1, 1) + sh0.view(1, -1, 1, 1)
    a1 = x1 * sc1.view(1, -1, 1, 1) + sh1.view(1, -1, 1, 1)
    a0.retain_grad(); a1.retain_grad()
    U = F.interpolate(torch.relu(torch.cat([a0, a1], 1)), scale_factor=2, mode="bilinear", align_corners=True)
    y_ref = F.conv2d(U, w, b, padding=1)
    y_ref.backward(dy)
    d = lambda t: t.detach().to(dev)   # noqa: E731
    cv = K.Conv(d(x0), Cout, 3, 1, 1, x1=d(x1), sc0=d(sc0), sh0=d(sh0), act0=ACT_RELU, sc1=d(sc1), sh1=d(sh1), act1=ACT_RELU,
                up2x=True)
    assert cv.head_applicable()
    for mode, fam in (("fwd", "head_fwd_kernel"), ("dgrad", "head_dgrad_kernel"), ("wgrad", "head_wgrad_kernel")):
        assert cv.kernel_name(mode, False) == fam, (mode, cv.kernel_name(mode, False))
    y = cv.fwd(cv.pack(d(w), 0), d(b), None)
    assert_close(y, y_ref, 2e-5, "head forward")
    dw, db = cv.wgrad(d(dy), want_bias=True)
    assert_close(dw, w.grad, 2e-5, "head dw")
    assert_close(db, b.grad, 2e-5, "head db")
    # the kernel returns the gradient wrt the activated-and-affine'd sources' PRE-ReLU value (dL/d a_i): a0.grad / a1.grad
    mean1, invstd1 = d(torch.randn(C1, generator=g) * 0.1), d(torch.rand(C1, generator=g) + 0.5)
    bst = K.zeros_stats(C1, d(x0))
    g0, g1 = cv.dgrad_up2x(d(w), d(dy), mean1=mean1, invstd1=invstd1, bstats1=bst)
    assert_close(g0, a0.grad, 2e-5, "head g0")
    assert_close(g1, a1.grad, 2e-5, "head g1")
    xh = (x1.detach().double() - mean1.cpu().double().view(1, -1, 1, 1)) * invstd1.cpu().double().view(1, -1, 1, 1)
    gd = a1.grad.double()
    st_ref = torch.cat([gd.sum((0, 2, 3)), (gd * xh).sum((0, 2, 3))])
    assert_close(bst, st_ref, 1e-4, "head BatchNorm-backward sums")


def test_down_conv_virtual_input(dev):
    """4x4 s2 halo-patch forward with the folded BatchNorm affine + LeakyReLU(0.2) of the previous encoder level
    (audio_net.py:57-58) and the BatchNorm statistics epilogue; dgrad / wgrad of the same descriptor."""
    K = _pkg().kernels
    g = torch.Generator().manual_seed(12)
    N, Cin, H, W, Cout = 3, 16, 48, 64, 48
    x = torch.randn(N, Cin, H, W, generator=g)
    sc, sh = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.5
    w = torch.randn(Cout, Cin, 4, 4, generator=g) * 0.06
    v = F.leaky_relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1), 0.2).requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    y_ref = F.conv2d(v, wr, None, 2, 1)
    dy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(dy)
    t = lambda z: z.to(dev)   # noqa: E731
    cv = K.Conv(t(x), Cout, 4, 2, 1, sc0=t(sc), sh0=t(sh), act0=2)
    st = K.zeros_stats(Cout, t(x))
    y = cv.fwd(cv.pack(t(w), 0), None, st)
    assert_close(y, y_ref, 2e-5, "fwd")
    st_ref = torch.cat([y_ref.detach().double().sum((0, 2, 3)), (y_ref.detach().double() ** 2).sum((0, 2, 3))])
    assert_close(st, st_ref, 1e-5, "stats")
    assert_close(cv.dgrad(cv.pack(t(w), 1), t(dy)), v.grad, 2e-5, "dgrad (wrt the activated input)")
    assert_close(cv.wgrad(t(dy))[0], wr.grad, 2e-5, "wgrad")


def test_resnet_glue_ops(dev):
    """The two elementwise pieces of the HIP visual trunk: max-pool 3x3/s2/p1 reading relu(scale*x+shift) on the fly
    (stem) and the BasicBlock tail relu(bn2(y2) + bn_d(y_d)) with its backward (+ BatchNorm-backward sums)."""
    K = _pkg().kernels
    g = torch.Generator().manual_seed(21)
    N, C, H, W = 2, 6, 11, 14
    x = torch.randn(N, C, H, W, generator=g)
    sc, sh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.5
    t = lambda z: z.to(dev)   # noqa: E731
    v = lambda z: z.view(1, -1, 1, 1)   # noqa: E731
    a = torch.relu(x * v(sc) + v(sh)).requires_grad_(True)
    p_ref = F.max_pool2d(a, 3, 2, 1)
    dp = torch.randn(p_ref.shape, generator=g)
    p_ref.backward(dp)
    p, idx = K.maxpool3x3s2(t(x), t(sc), t(sh), 1)
    assert_close(p, p_ref, 1e-6, "max-pool of relu(affine(x))")
    assert_close(K.maxpool3x3s2_bwd(t(dp), idx, H, W), a.grad, 1e-6, "max-pool backward (wrt the activated input)")
    # W % 8 == 0, even H: the four-outputs-per-thread kernel (the stem's 112x112 maps).  ReLU zeros tie inside most
    # windows: the gradient check pins "first maximum in scan order wins" (what F.max_pool2d does)
    x8 = torch.randn(3, 5, 12, 24, generator=g)
    sc8, sh8 = torch.rand(5, generator=g) + 0.5, torch.randn(5, generator=g) * 0.5 - 0.8
    a8 = torch.relu(x8 * v(sc8) + v(sh8)).requires_grad_(True)
    p8_ref = F.max_pool2d(a8, 3, 2, 1)
    dp8 = torch.randn(p8_ref.shape, generator=g)
    p8_ref.backward(dp8)
    p8, idx8 = K.maxpool3x3s2(t(x8), t(sc8), t(sh8), 1)
    assert_close(p8, p8_ref, 1e-6, "max-pool (vectorised kernel)")
    assert_close(K.maxpool3x3s2_bwd(t(dp8), idx8, 12, 24), a8.grad, 1e-6, "max-pool backward after the vectorised forward")
    # block tail
    y, r, dz = (torch.randn(N, C, H, W, generator=g) for _ in range(3))
    rs, rh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.5
    mean, inv = torch.randn(C, generator=g) * 0.2, torch.rand(C, generator=g) + 0.5
    for use_rs in (False, True):
        pre = y * v(sc) + v(sh) + (r * v(rs) + v(rh) if use_rs else r)
        z = K.affine_act(t(y), t(sc), t(sh), t(r), 1, t(rs) if use_rs else None, t(rh) if use_rs else None)
        assert_close(z, torch.relu(pre), 1e-6, "bn2 + residual + relu")
        gref = dz * (pre > 0).float()
        bst = K.zeros_stats(C, t(y))
        d = K.affine_act_bwd_(t(dz).clone(), t(y), t(sc), t(sh), t(r), None, t(mean), t(inv), 1, bst,
                              res_scale=t(rs) if use_rs else None, res_shift=t(rh) if use_rs else None)
        assert_close(d, gref, 1e-6, "masked gradient")
        xhat = (y - v(mean)) * v(inv)
        assert_close(bst, torch.cat([gref.double().sum((0, 2, 3)), (gref.double() * xhat.double()).sum((0, 2, 3))]), 1e-5,
                     "BatchNorm-backward sums")


def test_bss_eval_vs_oracle(dev):
    """BSS-eval SDR/SIR/SAR (mir_eval semantics, 512-tap projections) on the GPU against the numpy oracle, plus the
    known answers: +0.1 x other source -> SIR = 20 dB; +1 % noise -> SDR = SAR = 40 dB; a 12-tap filtered copy of the
    source still scores high SDR (the metric is invariant to short filters; the plain SDR of the same signal is < 3 dB)."""
    from oracle import bss_eval as OB
    from avsep_amd import bss_eval as PB
    rs = np.random.RandomState(0)
    L = 6000
    s = rs.randn(2, 2, L)
    h = np.zeros(13); h[7], h[12] = 0.8, 0.3
    ests = np.stack([np.stack([s[0, 0] + 0.1 * s[0, 1], s[0, 1] + 0.01 * rs.randn(L)]),
                     np.stack([np.convolve(s[1, 0], h)[:L], 0.5 * s[1, 1] + 0.3 * s[1, 0] + 0.05 * rs.randn(L)])])
    sdr, sir, sar = PB.bss_eval_sources(torch.from_numpy(s).to(dev), torch.from_numpy(ests).to(dev))
    for b in range(2):
        o = OB.bss_eval_sources(s[b], ests[b])
        for got, ref, name in zip((sdr, sir, sar), o, ("sdr", "sir", "sar")):
            for j in range(2):
                if ref[j] < 100:                          # (> 100 dB = numerically zero residual: only its size is checked)
                    assert abs(got[b, j].item() - ref[j]) < 1e-3, (name, b, j, got[b, j].item(), ref[j])
                else:
                    assert got[b, j].item() > 100
    # (the 1024-tap projection absorbs 1024/6000 of the white noise: +0.8 dB on the noise-limited ratios)
    assert abs(sir[0, 0].item() - 20.0) < 1.0 and abs(sdr[0, 1].item() - 40.0) < 1.0 and abs(sar[0, 1].item() - 40.0) < 1.0
    plain = 10 * np.log10((s[1, 0] ** 2).sum() / ((s[1, 0] - ests[1, 0]) ** 2).sum())
    assert sdr[1, 0].item() > 25 and plain < 3


@pytest.mark.parametrize("B,S,L,flen", [(3, 2, 20000, 512), (2, 3, 9000, 512), (2, 2, 3000, 64), (1, 1, 5000, 512)])
def test_bss_eval_kernels_vs_oracle(dev, B, S, L, flen):
    """csrc/bsseval.hip against the numpy restatement of mir_eval (oracle/bss_eval.py) on correlated mixtures: more samples,
    three sources (a 1536 x 1536 system per sample), a short filter, a single source; and the pieces on their own — the
    lagged correlations against numpy.correlate, the solved filters against numpy.linalg.solve on the same Gram matrix."""
    from oracle import bss_eval as OB
    from avsep_amd import bss_eval as PB
    P = _pkg()
    rs = np.random.RandomState(B * 100 + S * 10 + flen)
    s = rs.randn(B, S, L)
    s[:, :, 1:] += 0.6 * s[:, :, :-1]                                     # coloured sources: an ill-conditioned Gram matrix
    mixm = np.eye(S) + 0.2 * rs.randn(S, S)
    ests = np.einsum("es,bsl->bel", mixm, s) + 0.03 * rs.randn(B, S, L)
    sdr, sir, sar = PB.bss_eval_sources(torch.from_numpy(s).to(dev), torch.from_numpy(ests).to(dev), flen)
    for b in range(B):
        o = OB.bss_eval_sources(s[b], ests[b], flen)
        for got, ref, name in zip((sdr, sir, sar), o, ("sdr", "sir", "sar")):
            for j in range(S):
                if np.isfinite(ref[j]) and ref[j] < 100:
                    assert abs(got[b, j].item() - ref[j]) < 1e-3, (name, b, j, got[b, j].item(), ref[j])
    # the pieces: correlations and one solved system
    K = P.kernels
    R = torch.empty((B, S, S, 2 * flen - 1), dtype=torch.float64, device=dev)
    D = torch.empty((B, S, S, flen), dtype=torch.float64, device=dev)
    rt, et = torch.from_numpy(s).to(dev), torch.from_numpy(ests).to(dev)
    P.lib.call("avsep_bss_corr", K.ptr(rt), K.ptr(et), B, S, S, L, flen, K.ptr(R), K.ptr(D))
    i, j, e = 0, S - 1, S - 1
    full = np.correlate(s[0, i], s[0, j], "full")                         # full[L - 1 + tau] = sum_t a[t + tau] * b[t]
    assert_close(R[0, i, j], torch.from_numpy(full[L - 1 - (flen - 1):L - 1 + flen].copy()), 1e-11, "lagged correlations")
    fe = np.correlate(ests[0, e], s[0, i], "full")                        # fe[L - 1 + k] = sum_t est[t + k] * ref[t] = sum_t ref[t - k] est[t]
    assert_close(D[0, e, i], torch.from_numpy(fe[L - 1:L - 1 + flen].copy()), 1e-11, "right-hand sides")
    C = PB._solve(R, D, B, S, S, flen, 0)
    G = PB._gram(R, 0, list(range(S)), flen).cpu().numpy()
    rhs = D[0].reshape(S, S * flen).t().cpu().numpy()
    assert_close(C[0], torch.from_numpy(np.linalg.solve(G, rhs)), 1e-6, "filters vs numpy.linalg.solve on the same Gram matrix")


def test_bss_eval_silent_source_takes_the_least_squares_fallback(dev):
    """A silent reference makes the Gram matrix exactly singular: mir_eval catches numpy's LinAlgError and solves by least
    squares; the solve kernel reports the zero pivot (info) and the host side does the same for that system only."""
    from oracle import bss_eval as OB
    from avsep_amd import bss_eval as PB
    rs = np.random.RandomState(5)
    L = 4000
    s = rs.randn(2, 2, L)
    s[1, 1] = 0.0                                                          # sample 1: source 1 is silent
    ests = s + 0.1 * rs.randn(2, 2, L)
    sdr, sir, sar = PB.bss_eval_sources(torch.from_numpy(s).to(dev), torch.from_numpy(ests).to(dev))
    for b in range(2):
        with np.errstate(divide="ignore", invalid="ignore"):
            o = OB.bss_eval_sources(s[b], ests[b])
        for got, ref in zip((sdr, sir, sar), o):
            for j in range(2):
                if np.isfinite(ref[j]) and abs(ref[j]) < 100:
                    assert abs(got[b, j].item() - ref[j]) < 1e-3, (b, j, got[b, j].item(), ref[j])


@pytest.mark.parametrize("att", ["sig", "cos"])
@pytest.mark.parametrize("B,S,K,H,W", [(3, 2, 32, 14, 28), (2, 3, 8, 5, 7), (1, 4, 128, 20, 40),
                                       (1, 4, 128, 64, 64)])     # the limit shape of the ABI: 67 KB / 151 KB of dynamic LDS
def test_attmodel_core_kernel(dev, att, B, S, K, H, W):
    """csrc/attention.hip (SoP++/attention_net.py:24-58: similarity maps, match term, clamp, context vectors) against the
    same formula in float64 torch with autograd: values and the gradients wrt the audio queries and the visual map,
    with cotangents on all three outputs.  The cos maps are scaled into (-1.5, 1.5) by the test so that the clamp's
    inactive branch is exercised too."""
    from avsep_amd.models.attention_net import _AttInferFn, _ATT
    g = torch.Generator().manual_seed(B * 100 + K)
    a = torch.randn(B, S, K, generator=g) * (1.0 if att == "sig" else 1.0)
    mix = torch.randn(B, K, H, W, generator=g)
    a64, m64 = a.double().requires_grad_(True), mix.double().requires_grad_(True)
    a5, v5 = a64[..., None, None], m64[:, None]
    if att == "cos":
        maps = F.cosine_similarity(a5, v5, dim=2)
    else:
        maps = torch.sigmoid(torch.sum(a5 * v5 / K ** 0.5, dim=2))
    match = -maps.mean(dim=(-2, -1)).sum(-1)
    mc = maps.clamp(0, 1)
    ctx = (m64[:, None] * mc[:, :, None]).mean(dim=(-2, -1))
    c_ctx, c_maps, c_match = (torch.randn(t.shape, generator=g) for t in (ctx, mc, match))
    ((ctx * c_ctx.double()).sum() + (mc * c_maps.double()).sum() + (match * c_match.double()).sum()).backward()
    ad, md = a.to(dev).requires_grad_(True), mix.to(dev).requires_grad_(True)
    ctx_d, maps_d, match_d = _AttInferFn.apply(ad, md, _ATT[att])
    ((ctx_d * c_ctx.to(dev)).sum() + (maps_d * c_maps.to(dev)).sum() + (match_d * c_match.to(dev)).sum()).backward()
    assert_close(ctx_d, ctx, 2e-5, "ctx"); assert_close(maps_d, mc, 2e-5, "maps"); assert_close(match_d, match, 2e-5, "match")
    assert_close(ad.grad, a64.grad, 5e-5, "d queries")
    assert_close(md.grad, m64.grad, 5e-5, "d visual map")
