// bf16-operand / fp32-accumulate convolutions (avsep_conv_desc.prec == AVSEP_PREC_BF16): host side of halo_bf16.h.
//   3x3 / stride 1 / pad = dil (dil 1, 2): forward, and data gradient through flipped + transposed weights
//     — the U-Net decoder convs (models/audio_net.py:75-76,85-87,96-98,180-182) and the ResNet BasicBlock convs
//       (models/vision_net.py:84-92 over torchvision resnet18);
//   4x4 / stride 2 / pad 1: forward (U-Net encoder, audio_net.py:57-58,170-171) and its data gradient as four 2x2-tap
//     parity classes (see conv3x3.hip);
//   3x3 / stride 2 / pad 1 forward (ResNet layer2.0 / layer3.0 conv1).
// Packed weight image (bf16), written once per optimizer step by bf_pack_kernel and copied tile by tile into LDS by
// LDS-DMA:  [K-tile kt (16 channels)][tap][m (padded to 128)][16 channels], the two 8-channel halves of a row swapped
// when bit 3 of m is set (the LDS bank swizzle of halo_bf16.h, applied at pack time because the DMA cannot permute).
#include <stdlib.h>

#include "halo_bf16.h"
#include "res_bf16.h"

// mode 0: forward  M = Cout, k-channel = ci, value w[m][c][tap]
// mode 1: dgrad    M = Cin,  k-channel = co, value w[c][m][flip(tap)]           (KH x KW taps, stride 1)
// mode 2: dgrad of a 4x4/s2 conv, 4 parity classes x 2x2 taps: class cls at image offset cls * (Kc/16)*4*ld rows;
//         tap (th, tw) of class (ph, pw) -> (kh, kw) = (ph ? 2-2*th : 3-2*th, pw ? 2-2*tw : 3-2*tw)
// mode 3: dgrad of a 3x3/s2/p1 conv: class (ph, pw) has (ph ? 2 : 1) x (pw ? 2 : 1) taps (1, 2, 2, 4 = 9 in all, the
//         class images follow each other); tap th -> kh = ph ? 2-2*th : 1 reads dY row a + th for input row 2a + ph
__global__ void bf_pack_kernel(const float* __restrict__ w, unsigned* __restrict__ out, int Cout, int Cin, int KH, int KW,
                               int ld, int ktiles, int mode) {
  int NT = mode == 2 ? 4 : KH * KW;
  const long long rows = (long long)(mode == 2 ? 4 : 1) * ktiles * NT * ld;      // 32-byte rows (mode 3: 9 taps in all)
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;          // one thread per (row, stored half)
  if (i >= rows * 2) return;
  long long row = i >> 1;
  const int hs = (int)(i & 1);
  int cls = 0;
  if (mode == 3) {                                                               // classes of 1, 2, 2, 4 taps
    const long long unit = (long long)ktiles * ld;
    if (row < unit) { cls = 0; NT = 1; }
    else if (row < 3 * unit) { cls = 1; NT = 2; row -= unit; }
    else if (row < 5 * unit) { cls = 2; NT = 2; row -= 3 * unit; }
    else { cls = 3; NT = 4; row -= 5 * unit; }
  }
  const int m = (int)(row % ld);
  long long q = row / ld;
  const int tap = (int)(q % NT); q /= NT;
  const int kt = (int)(q % ktiles);
  if (mode == 2) cls = (int)(q / ktiles);
  const int h = hs ^ ((m >> 3) & 1);                                             // logical half stored in slot hs
  const int KHW = KH * KW;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = kt * 16 + 8 * h + j;
    float x = 0.f;
    if (mode == 0) {
      if (m < Cout && c < Cin) x = w[((long long)m * Cin + c) * KHW + tap];
    } else if (mode == 1) {
      if (m < Cin && c < Cout) x = w[((long long)c * Cin + m) * KHW + (KHW - 1 - tap)];
    } else if (mode == 2) {
      const int ph = cls >> 1, pw = cls & 1, th = tap >> 1, tw = tap & 1;
      const int kh = ph ? 2 - 2 * th : 3 - 2 * th, kw = pw ? 2 - 2 * tw : 3 - 2 * tw;
      if (m < Cin && c < Cout) x = w[((long long)c * Cin + m) * 16 + kh * 4 + kw];
    } else {
      const int ph = cls >> 1, pw = cls & 1, ntw = pw ? 2 : 1, th = tap / ntw, tw = tap % ntw;
      const int kh = ph ? 2 - 2 * th : 1, kw = pw ? 2 - 2 * tw : 1;
      if (m < Cin && c < Cout) x = w[((long long)c * Cin + m) * 9 + kh * 3 + kw];
    }
    v[j] = x;
  }
  u32x4 o = {bf_pack2(v[0], v[1]), bf_pack2(v[2], v[3]), bf_pack2(v[4], v[5]), bf_pack2(v[6], v[7])};
  reinterpret_cast<u32x4*>(out)[i] = o;
}

static inline bool bf_enabled(const avsep_conv_desc* d) { return !(d->algo & AVSEP_ALGO_NO_BF16_KERNELS); }

// geometry classes served by convbf_kernel; mode 0 forward, 1 data gradient
static int bf_class(const avsep_conv_desc* d) {
  if (d->up2x || d->C0 != d->Cin) return 0;
  if (d->KH == 3 && d->KW == 3 && d->stride == 1 && (d->dil == 1 || d->dil == 2) && d->pad == d->dil) return 3;
  if (d->KH == 4 && d->KW == 4 && d->stride == 2 && d->pad == 1 && d->dil == 1) return 4;
  if (d->KH == 3 && d->KW == 3 && d->stride == 2 && d->pad == 1 && d->dil == 1) return 5;      // ResNet layer2.0 / layer3.0 conv1
  if (d->KH == 1 && d->KW == 1 && (d->stride == 1 || d->stride == 2) && d->pad == 0) return 1;  // ResNet downsample convs
  if (d->KH == 4 && d->KW == 4 && d->stride == 1 && d->pad == 0 && d->dil == 1) return 6;      // stem after space-to-depth
  return 0;
}
// flat-pixel tiles: the small square maps of the visual trunk (14x14, 7x7) and of the deep U-Net levels (8x8, 4x4)
static int bf_flat_w(int H, int W, int dil) {
  if (H < W) return 0;
  if (dil == 1 && (W == 14 || W == 7 || W == 8 || W == 4)) return W;
  if (dil == 2 && W == 14) return W;
  return 0;
}
static bool bf_flat(const avsep_conv_desc* d) { return bf_flat_w(d->H, d->W, d->dil) > 0; }

// split-K over the 16-channel K-tiles for layers whose output grid cannot fill the chip (deep U-Net levels: 4x4 / 8x8
// maps with K = 9216): partial slabs in the workspace + the fp32 combine of conv.hip
struct BfSplit { int splits, kts; };
static BfSplit bf_split_plan(long long wgs, int nK) {
  BfSplit p{1, 0};
  if (wgs >= 384 || nK < 16) return p;
  int s = cdiv(768, wgs), maxs = nK / 8;
  if (s > maxs) s = maxs;
  if (s > 32) s = 32;
  if (s < 2) return p;
  p.kts = cdiv(nK, s);
  p.splits = cdiv(nK, p.kts);
  if (p.splits < 2) { p.splits = 1; p.kts = 0; }
  return p;
}
// tile decisions of the flat launches, shared by the launch, the workspace query and avsep_conv_kernel_variant.
// Measured on one box (round 4, `tools/conv_bench.py --prec bf16`): 256 -> 256 @ 14x14 (294 pixel tiles of 256) ran 30 % faster
// on 64-row tiles x 512 threads than on 128-row tiles x 256 threads (0.086 -> 0.060 ms forward and data gradient): when
// 128-row tiles cannot fill two rounds of 512-thread workgroups, halve the rows instead of the pixels.
struct BfFlat { bool m64, big; int gm; };
static BfFlat bf_flat_plan(int M, long long planP) {
  BfFlat f;
  const long long t256 = cdiv(planP, 256);
  f.m64 = M <= 64 || (long long)cdiv(M, 128) * t256 < 512;
  f.gm = cdiv(M, f.m64 ? 64 : 128);
  f.big = (long long)f.gm * t256 >= 512;
  return f;
}
// workgroups of the unsplit flat launch (mirrors bf_launch_flat)
static long long bf_flat_wgs(int M, long long P) {
  const BfFlat f = bf_flat_plan(M, P);
  return (long long)f.gm * cdiv(P, f.big ? 256 : 128);
}

size_t bf_workspace_bytes(const avsep_conv_desc* d, int mode);
bool bf_applicable(const avsep_conv_desc* d, int mode) {
  if (d->prec != AVSEP_PREC_BF16 || !bf_enabled(d)) return false;
  const int cls = bf_class(d);
  if (!cls) return false;
  const int kc = mode == 0 ? d->Cin : d->Cout, m = mode == 0 ? d->Cout : d->Cin;
  if (kc % BF_CK != 0 || m < 32) return false;
  if (mode == 0 && d->scale0 && d->Cin > BF_AFF_MAX) return false;
  // 32-bit 16-byte-unit offsets inside the staged B16 image, int pixel indices; grid dimension
  const long long in_elems = mode == 0 ? (long long)d->N * d->Cin * d->H * d->W : (long long)d->N * d->Cout * d->Ho * d->Wo;
  if (in_elems >= (1LL << 33) || (long long)d->N * d->H * d->W >= (1LL << 31) || d->N > 65535) return false;
  if (cls == 3) return bf_flat(d) || (d->W >= 16 && d->H >= 4);
  if (cls == 6) return mode == 0 && d->Wo >= 16 && d->Ho >= 4;
  if (cls == 1) return d->Wo >= 8 && d->Ho >= 4 && (d->stride == 1 || mode == 0 || ((d->H & 1) == 0 && (d->W & 1) == 0));
  if (mode == 0) return d->Wo >= 8 && d->Ho >= 4;               // 8-wide outputs (U-Net d5) use half of a 16-wide tile
  return d->Wo >= 8 && d->Ho >= 4 && (d->H & 1) == 0 && (d->W & 1) == 0;
}

// may the output (y for mode 0, dx for mode 1) of this call be written as a B16 image?  Not through split-K slabs (their
// combine writes fp32), and the output channels must come in whole 16-channel blocks
bool bf_out_b16(const avsep_conv_desc* d, int mode) {
  const int M = mode == 0 ? d->Cout : d->Cin;
  if (M % 16 != 0) return false;
  return bf_workspace_bytes(d, mode) == 0;
}

size_t bf_workspace_bytes(const avsep_conv_desc* d, int mode) {
  if (bf_class(d) != 3 || !bf_flat(d)) return 0;
  const int M = mode == 0 ? d->Cout : d->Cin, kc = mode == 0 ? d->Cin : d->Cout;
  const long long P = (long long)d->N * d->H * d->W;
  BfSplit sp = bf_split_plan(bf_flat_wgs(M, (long long)plan_batch(d) * d->H * d->W), kc / BF_CK);
  return sp.splits > 1 ? (size_t)sp.splits * M * P * sizeof(float) : 0;
}

size_t bf_packed_floats(const avsep_conv_desc* d, int mode) {
  const int kc = mode == 0 ? d->Cin : d->Cout, ld = roundup(mode == 0 ? d->Cout : d->Cin, 128);
  const int NT = (bf_class(d) == 4 && mode == 1) ? 16 : d->KH * d->KW;           // 4x4/s2 dgrad: 4 classes x 4 taps; 3x3/s2: 1+2+2+4
  return (size_t)(kc / BF_CK) * NT * ld * 8;                                      // 32-byte rows = 8 floats
}

int bf_pack(const avsep_conv_desc* d, const float* w, float* packed, int mode, hipStream_t st) {
  const int kc = mode == 0 ? d->Cin : d->Cout, ld = roundup(mode == 0 ? d->Cout : d->Cin, 128);
  const int pmode = (mode == 1 && bf_class(d) == 4) ? 2 : (mode == 1 && bf_class(d) == 5) ? 3 : mode;
  const long long halves = (long long)bf_packed_floats(d, mode) / 4;
  hipLaunchKernelGGL(bf_pack_kernel, dim3(cdiv(halves, 256)), dim3(256), 0, st, w, reinterpret_cast<unsigned*>(packed),
                     d->Cout, d->Cin, d->KH, d->KW, ld, kc / BF_CK, pmode);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

// tile decisions shared by the launches below and avsep_conv_kernel_variant
struct BfRect { bool wide, m64, big; int gm; };
// nK = 16-channel K-tiles of the call.  Round-4 measurements behind the two rules (one box, conv_bench): (1) short-K calls
// (<= 8 K-tiles: the data gradients of u2 / u3, 256 <- 64 @ 128x128 and 512 <- 128 @ 64x64) gain 7 % on 64-row tiles — twice
// the workgroups hide each other's prologue and epilogue — while long-K calls lose 5 %; (2) 1024 -> 512 @ 16x16 forward
// (256 workgroups of 256 pixels) runs 0.189 -> 0.140 ms on the 512-thread tiles: one full round of them is enough.
static BfRect bf_rect_plan(int M, int Ho, int Wo, int taps, long long planN, int nK) {
  BfRect r;
  r.wide = Wo >= 32;
  // 16-tap weight tiles of 128 rows (2 x 64 KB) would not fit beside the stride-2 patch: 64-row tiles there
  // short reductions (K-tiles x taps <= 72): 64-row tiles when there are several M tiles anyway (M >= 256) or the taps are few
  // (the 1 / 2 / 4-tap parity classes of the stride-2 data gradients: 128 <- 256 @ 28x28 0.090 -> 0.077 ms); a single
  // 128-row tile stays (64 -> 128 3x3/s2 forward: 0.066 on 64 rows, 0.056 on 128)
  r.m64 = M <= 64 || taps > 9 || (nK * taps <= 72 && (M >= 256 || taps <= 4));
  r.gm = cdiv(M, r.m64 ? 64 : 128);
  const long long wg256 = (long long)r.gm * cdiv(Wo, r.wide ? 32 : 16) * cdiv(Ho, r.wide ? 8 : 16) * planN;
  // 512-thread tiles from one full round; a 16-row tile also pays on 9..15-row maps (128 -> 256 3x3/s2 @ 14x14: 0.075 -> 0.048 ms)
  r.big = wg256 >= 256 && Ho > 8;
  return r;
}
static bool bf_res_ok(int taps_h, int taps_w, int Kc, int M, int W, int dil, bool flat);
void bf_variant(const avsep_conv_desc* d, int mode, char* buf, size_t cap) {
  const int cls = bf_class(d);
  const int M = mode == 0 ? d->Cout : d->Cin, kc = mode == 0 ? d->Cin : d->Cout;
  const long long pn = plan_batch(d);
  if (cls == 3 && bf_flat(d)) {
    const long long planP = pn * d->H * d->W;
    const BfFlat f = bf_flat_plan(M, planP);
    snprintf(buf, cap, "flat%d,%dx%d,split%d", bf_flat_w(d->H, d->W, d->dil), f.m64 ? 64 : 128, f.big ? 256 : 128,
             bf_split_plan(bf_flat_wgs(M, planP), kc / BF_CK).splits);
    return;
  }
  if ((cls == 3 || (cls == 6 && mode == 0)) && bf_res_ok(d->KH, d->KW, kc, M, mode == 0 ? d->Wo : d->W, d->dil, false)) {
    snprintf(buf, cap, "res8x32,64x256");
    return;
  }
  int Ho = d->Ho, Wo = d->Wo, taps = d->KH * d->KW;
  if (mode == 1 && cls == 3) { Ho = d->H; Wo = d->W; }
  else if (mode == 1 && cls == 1) taps = 1;
  else if (mode == 1) { Ho = d->H / 2; Wo = d->W / 2; taps = 4; }     // stride-2 data gradient: the (2x2-tap) parity classes
  const BfRect r = bf_rect_plan(M, Ho, Wo, taps, pn, kc / BF_CK);
  snprintf(buf, cap, "%s,%dx%d", r.wide ? (r.big ? "8x32" : "4x32") : (r.big ? "16x16" : "8x16"), r.m64 ? 64 : 128, r.big ? 256 : 128);
}

// ---- resident-weights persistent form (res_bf16.h): short reductions with <= 64 output rows --------------------------------
// 3x3 / s1 / dil 1 with exactly 64 reduction channels on maps >= 32 wide (ResNet layer1, forward and data gradient), and the
// stem's 4x4 / s1 form over the 16-channel space-to-depth frames
static bool bf_res_ok(int taps_h, int taps_w, int Kc, int M, int W, int dil, bool flat) {
  if (flat || dil != 1 || M > 64 || W < 32) return false;
  return (taps_h == 3 && taps_w == 3 && Kc == 64) || (taps_h == 4 && taps_w == 4 && Kc == 16);
}
template <int KH_, int KW_, int NKT>
static int bf_launch_res(C3Args& a, hipStream_t st) {
  a.gridM = 1;
  a.tilesX = cdiv(a.Wo, 32);
  a.tilesY = cdiv(a.Ho, 8);
  const long long T = (long long)a.N * a.tilesX * a.tilesY;
  dim3 grid((unsigned)(T < cu_count() ? T : cu_count()));
  const bool raw = a.sc0 == nullptr && a.act0 == AVSEP_ACT_NONE;
  if (raw) hipLaunchKernelGGL((convbf_res_kernel<8, 32, KH_, KW_, NKT, true>), grid, dim3(512), 0, st, a);
  else hipLaunchKernelGGL((convbf_res_kernel<8, 32, KH_, KW_, NKT, false>), grid, dim3(512), 0, st, a);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

// ---- launch: tile choice ------------------------------------------------------------------------------------------
// KH_/KW_/S_/DIL_ select the instantiation; a.Ho/Wo = output tile space.  256-pixel tiles x 128 rows (512 threads)
// when that still yields >= 2 workgroups per CU-slot, else 128-pixel tiles; 64-row tiles for Cout <= 64.
template <int KH_, int KW_, int S_, int DIL_>
static int bf_launch_rect(C3Args& a, hipStream_t st) {
  constexpr bool ONLY64 = KH_ * KW_ > 9;
  const BfRect r = bf_rect_plan(a.Cout, a.Ho, a.Wo, KH_ * KW_, c3_plan_n(a), a.Cin / BF_CK);
  const bool wide = r.wide, m64 = r.m64, big = r.big;
  a.gridM = r.gm;
  a.tilesX = cdiv(a.Wo, wide ? 32 : 16);
  a.tilesY = cdiv(a.Ho, wide ? (big ? 8 : 4) : (big ? 16 : 8));
  dim3 grid((unsigned)((long long)a.gridM * a.tilesX * a.tilesY * a.N));
  const bool raw = a.sc0 == nullptr && a.act0 == AVSEP_ACT_NONE;
#define BF_L(TH_, TW_, BM_, NWN_)                                                                                               \
  do {                                                                                                                          \
    if (raw) hipLaunchKernelGGL((convbf_kernel<TH_, TW_, BM_, KH_, KW_, S_, DIL_, 0, NWN_, true>), grid, dim3(128 * NWN_), 0, st, a); \
    else hipLaunchKernelGGL((convbf_kernel<TH_, TW_, BM_, KH_, KW_, S_, DIL_, 0, NWN_, false>), grid, dim3(128 * NWN_), 0, st, a); \
  } while (0)
  if (m64) {
    if (big) { if (wide) BF_L(8, 32, 64, 4); else BF_L(16, 16, 64, 4); }
    else { if (wide) BF_L(4, 32, 64, 2); else BF_L(8, 16, 64, 2); }
  } else if constexpr (!ONLY64) {
    if (big) { if (wide) BF_L(8, 32, 128, 4); else BF_L(16, 16, 128, 4); }
    else { if (wide) BF_L(4, 32, 128, 2); else BF_L(8, 16, 128, 2); }
  }
#undef BF_L
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

template <int FW_, int DIL_>
static int bf_launch_flat(C3Args& a, int splits, hipStream_t st) {
  const long long P = (long long)a.N * a.H * a.W, planP = c3_plan_n(a) * a.H * a.W;
  const BfFlat f = bf_flat_plan(a.Cout, planP);
  const bool m64 = f.m64, big = f.big;
  a.gridM = f.gm;
  a.tilesX = cdiv(P, big ? 256 : 128);
  a.tilesY = 1;
  dim3 grid((unsigned)((long long)a.gridM * a.tilesX), splits);
  const bool raw = a.sc0 == nullptr && a.act0 == AVSEP_ACT_NONE;
#define BF_F(BM_, NWN_)                                                                                                    \
  do {                                                                                                                     \
    if (raw) hipLaunchKernelGGL((convbf_kernel<0, 0, BM_, 3, 3, 1, DIL_, FW_, NWN_, true>), grid, dim3(128 * NWN_), 0, st, a); \
    else hipLaunchKernelGGL((convbf_kernel<0, 0, BM_, 3, 3, 1, DIL_, FW_, NWN_, false>), grid, dim3(128 * NWN_), 0, st, a); \
  } while (0)
  if (big) { if (m64) BF_F(64, 4); else BF_F(128, 4); }
  else { if (m64) BF_F(64, 2); else BF_F(128, 2); }
#undef BF_F
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

// 3x3 / stride 1; `ws` (may be null): split-K slabs, then *splits_out > 1 and the caller combines
static int bf3_launch(C3Args& a, int dil, void* ws, size_t ws_bytes, int* splits_out, hipStream_t st) {
  a.Ho = a.H; a.Wo = a.W; a.padh = a.padw = dil; a.os = 1; a.ooh = a.oow = 0; a.OHs = a.H; a.OWs = a.W;
  *splits_out = 1;
  const int fw = bf_flat_w(a.H, a.W, dil);
  if (fw) {
    const long long P = (long long)a.N * a.H * a.W;
    BfSplit sp = bf_split_plan(bf_flat_wgs(a.Cout, c3_plan_n(a) * a.H * a.W), a.Cin / BF_CK);
    if (sp.splits > 1 && ws && ws_bytes >= (size_t)sp.splits * a.Cout * P * sizeof(float)) {
      a.kts = sp.kts; a.slab = (long long)a.Cout * P; a.out = (float*)ws;
      *splits_out = sp.splits;
    } else {
      sp.splits = 1;
    }
    if (fw == 14) return dil == 1 ? bf_launch_flat<14, 1>(a, sp.splits, st) : bf_launch_flat<14, 2>(a, sp.splits, st);
    if (fw == 7) return bf_launch_flat<7, 1>(a, sp.splits, st);
    if (fw == 8) return bf_launch_flat<8, 1>(a, sp.splits, st);
    return bf_launch_flat<4, 1>(a, sp.splits, st);
  }
  if (bf_res_ok(3, 3, a.Cin, a.Cout, a.W, dil, false)) return bf_launch_res<3, 3, 4>(a, st);
  return dil == 1 ? bf_launch_rect<3, 3, 1, 1>(a, st) : bf_launch_rect<3, 3, 1, 2>(a, st);
}
int splitk_combine(const float* ws, long long slab, int S, const avsep_conv_desc* d, const float* bias, float* y, double* stats,
                   hipStream_t st);                                                   // conv.hip
int reduce_slabs(const float* ws, float* out, long long n, int S, hipStream_t st);   // conv.hip

int bf_fwd(const avsep_conv_desc* d, const float* wp, const float* bias, float* y, double* stats, void* ws, size_t ws_bytes,
           hipStream_t st) {
  C3Args a{};
  a.N = d->N; a.planN = d->plan_n; a.Cin = d->Cin; a.H = d->H; a.W = d->W; a.Cout = d->Cout;
  a.C0 = d->Cin; a.C1 = 0; a.act0 = d->act0; a.Hs = d->H; a.Ws = d->W;
  a.x0 = d->x0; a.sc0 = d->scale0; a.sh0 = d->shift0;
  a.wp = wp; a.wp_ld = roundup(d->Cout, 128); a.out = y; a.bias = bias; a.stats = stats;
  if (d->xfmt != AVSEP_FMT_B16) return AVSEP_ERR_ARG;               // the bf16 kernels stage B16 images only
  a.out16 = d->yfmt == AVSEP_FMT_B16;
  if (a.out16 && (d->Cout % 16 != 0 || !bf_out_b16(d, 0))) return AVSEP_ERR_ARG;
  if (bf_class(d) == 3) {
    int splits = 1;
    int rc = bf3_launch(a, d->dil, ws, ws_bytes, &splits, st);
    if (rc || splits == 1) return rc;
    return splitk_combine((const float*)ws, a.slab, splits, d, bias, y, stats, st);
  }
  a.Ho = d->Ho; a.Wo = d->Wo; a.padh = a.padw = d->pad; a.os = 1; a.ooh = a.oow = 0; a.OHs = d->Ho; a.OWs = d->Wo;
  switch (bf_class(d)) {
    case 4: return bf_launch_rect<4, 4, 2, 1>(a, st);
    case 5: return bf_launch_rect<3, 3, 2, 1>(a, st);
    case 6: return bf_res_ok(4, 4, a.Cin, a.Cout, a.Wo, 1, false) ? bf_launch_res<4, 4, 1>(a, st) : bf_launch_rect<4, 4, 1, 1>(a, st);
    default: return d->stride == 1 ? bf_launch_rect<1, 1, 1, 1>(a, st) : bf_launch_rect<1, 1, 2, 1>(a, st);
  }
}

int bf_dgrad(const avsep_conv_desc* d, const float* wp, const float* dy, float* dx, void* ws, size_t ws_bytes, hipStream_t st) {
  if (d->dyfmt != AVSEP_FMT_B16) return AVSEP_ERR_ARG;
  const int out16 = d->dxfmt == AVSEP_FMT_B16;
  if (out16 && (d->Cin % 16 != 0 || !bf_out_b16(d, 1))) return AVSEP_ERR_ARG;
  if (bf_class(d) == 3) {
    C3Args a{};
    a.N = d->N; a.planN = d->plan_n; a.Cin = d->Cout; a.H = d->H; a.W = d->W; a.Cout = d->Cin;
    a.C0 = d->Cout; a.C1 = 0; a.Hs = d->H; a.Ws = d->W;
    a.x0 = dy; a.wp = wp; a.wp_ld = roundup(d->Cin, 128); a.out = dx; a.out16 = out16;
    int splits = 1;
    int rc = bf3_launch(a, d->dil, ws, ws_bytes, &splits, st);
    if (rc || splits == 1) return rc;
    return reduce_slabs((const float*)ws, dx, a.slab, splits, st);
  }
  const int ld = roundup(d->Cin, 128);
  const int bcls = bf_class(d);
  if (bcls == 1) {            // 1x1: dX at the sampled positions = W^T dY; stride 2 leaves the other positions zero
    C3Args a{};
    a.N = d->N; a.planN = d->plan_n; a.Cin = d->Cout; a.H = d->Ho; a.W = d->Wo; a.Cout = d->Cin;
    a.C0 = d->Cout; a.C1 = 0; a.Hs = d->Ho; a.Ws = d->Wo;
    a.x0 = dy; a.wp = wp; a.wp_ld = ld; a.out = dx; a.out16 = out16;
    a.Ho = d->Ho; a.Wo = d->Wo; a.padh = a.padw = 0; a.os = d->stride; a.ooh = a.oow = 0; a.OHs = d->H; a.OWs = d->W;
    if (d->stride == 2 && hipMemsetAsync(dx, 0, (size_t)d->N * d->Cin * d->H * d->W * (out16 ? 2 : sizeof(float)), st) != hipSuccess)
      return AVSEP_ERR_LAUNCH;
    return bf_launch_rect<1, 1, 1, 1>(a, st);
  }
  const size_t unit_floats = (size_t)(d->Cout / BF_CK) * ld * 8;        // one tap of one class image
  size_t off = 0;
  for (int cls = 0; cls < 4; ++cls) {
    const int ph = cls >> 1, pw = cls & 1;
    C3Args a{};
    a.N = d->N; a.planN = d->plan_n; a.Cin = d->Cout; a.H = d->Ho; a.W = d->Wo; a.Cout = d->Cin;     // the conv runs over dY
    a.C0 = d->Cout; a.C1 = 0; a.Hs = d->Ho; a.Ws = d->Wo;
    a.x0 = dy; a.wp = wp + off; a.wp_ld = ld; a.out = dx; a.out16 = out16;
    a.Ho = d->H / 2; a.Wo = d->W / 2;
    a.os = 2; a.ooh = ph; a.oow = pw; a.OHs = d->H; a.OWs = d->W;
    int rc;
    if (bcls == 4) {          // 4x4/s2: every class is a 2x2-tap conv, pad 1 | 0 for parity 0 | 1
      a.padh = ph ? 0 : 1; a.padw = pw ? 0 : 1;
      rc = bf_launch_rect<2, 2, 1, 1>(a, st);
      off += 4 * unit_floats;
    } else {                  // 3x3/s2: (ph ? 2 : 1) x (pw ? 2 : 1) taps over dY rows a .. a + 1, no padding
      a.padh = a.padw = 0;
      rc = ph ? (pw ? bf_launch_rect<2, 2, 1, 1>(a, st) : bf_launch_rect<2, 1, 1, 1>(a, st))
              : (pw ? bf_launch_rect<1, 2, 1, 1>(a, st) : bf_launch_rect<1, 1, 1, 1>(a, st));
      off += (size_t)(ph ? 2 : 1) * (pw ? 2 : 1) * unit_floats;
    }
    if (rc) return rc;
  }
  return AVSEP_OK;
}
