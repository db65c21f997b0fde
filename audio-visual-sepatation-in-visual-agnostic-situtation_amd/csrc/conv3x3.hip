// 3x3 / stride 1 / pad 1 convolution on the f32 MFMA with an LDS halo patch ("v2" path).
// These are the U-Net decoder convs (audio_net.py:75-76,85-87,96-98,180-182): 77 % of the U-Net's MACs.
//
// Instead of gathering an im2col operand (every input element fetched and transformed 9 times, with
// per-element address arithmetic that out-weighs the 64-cycle f32 MFMA), a workgroup stages, per chunk of
// CK input channels, the (TH+2)x(TW+2) input halo patch of its TH x TW output tile ONCE — the folded
// BatchNorm affine, ReLU, the two-source skip concat and the optional bilinear x2 upsample are applied
// while staging — and the MFMA loop reads both operands from LDS with immediate offsets:
//   K order is (channel pair, tap, channel parity); lane half (lane>>5) selects the parity, so the B
//   operand address is  lane_base(pixel, parity) + const(pair, tap)  — zero VALU per operand.
// The same kernel computes the data gradient (a 3x3/s1/p1 conv of dY with flipped, transposed weights):
// only the weight packing differs.  D[co][pix], pixels on the lanes -> coalesced NCHW stores.
#include <stdlib.h>

#include "halo_kernel.h"

// weight packing for this path.  Rows r = ((cpair*9 + tap)*2 + parity), column = output channel.
//   mode 0 (forward): in-channel ci = 2*cpair+parity, value w[co][ci][tap],          column co
//   mode 1 (dgrad)  : "in"-channel co = 2*cpair+parity, value w[co][ci][8 - tap],    column ci
__global__ void c3_pack_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin, int rows, int ld,
                               int mode) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)rows * ld) return;
  int row = (int)(i / ld), col = (int)(i % ld);
  int parity = row & 1, q = row >> 1, tap = q % 9, ch = 2 * (q / 9) + parity;
  float v = 0.f;
  if (mode == 0) {
    if (ch < Cin && col < Cout) v = w[((long long)col * Cin + ch) * 9 + tap];
  } else {
    if (ch < Cout && col < Cin) v = w[((long long)ch * Cin + col) * 9 + (8 - tap)];
  }
  out[i] = v;
}

// ---------------------------------------------------------------------------
// host side (called from conv.hip)
// ---------------------------------------------------------------------------
// forward: needs Cin % 4 == 0 and the source split on a chunk boundary; dgrad: Cout % 4 == 0
// conv_flat.hip: flat-pixel tiles for the small square maps of the visual trunk
int c3_flat_width(int H, int W, int dil);
int c3_flat_launch(C3Args& a, int dil, hipStream_t st);
static bool c3_flat_enabled(int algo) { return !(algo & AVSEP_ALGO_NO_FLAT); }
static bool c3_flat(const avsep_conv_desc* d) {
  return !d->up2x && d->C0 == d->Cin && c3_flat_width(d->H, d->W, d->dil) > 0 && (long long)d->N * d->H * d->W < 0x7fffffffLL &&
         c3_flat_enabled(d->algo);
}

bool c3_applicable(const avsep_conv_desc* d, int mode) {
  if (!(d->KH == 3 && d->KW == 3 && d->stride == 1 && (d->dil == 1 || d->dil == 2) && d->pad == d->dil)) return false;
  if (d->dil == 2 && d->up2x) return false;
  if (!c3_flat(d) && (d->W < 12 || d->H < 4 || d->N > 65535)) return false;   // 7x7 maps only through the flat tiles
  if (mode == 0) return d->Cin % C3_CK == 0 && d->C0 % C3_CK == 0 && d->Cout > 4;
  return d->Cout % C3_CK == 0 && d->Cin >= 32;
}
size_t c3_packed_floats(const avsep_conv_desc* d, int mode) {
  int rows = (mode == 0 ? d->Cin : d->Cout) * 9, ld = roundup(mode == 0 ? d->Cout : d->Cin, 128);
  return (size_t)rows * ld;
}
int c3_pack(const avsep_conv_desc* d, const float* w, float* packed, int mode, hipStream_t st) {
  int rows = (mode == 0 ? d->Cin : d->Cout) * 9, ld = roundup(mode == 0 ? d->Cout : d->Cin, 128);
  long long total = (long long)rows * ld;
  hipLaunchKernelGGL(c3_pack_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, w, packed, d->Cout, d->Cin, rows, ld, mode);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

static int c3_launch(C3Args& a, hipStream_t st, int dil = 1) {
  a.Ho = a.H; a.Wo = a.W; a.padh = a.padw = dil; a.os = 1; a.ooh = a.oow = 0; a.OHs = a.H; a.OWs = a.W;
  if (!a.up2x && a.C1 == 0 && c3_flat_width(a.H, a.W, dil) > 0 && c3_flat_enabled(a.algo))
    return c3_flat_launch(a, dil, st);
  const bool wide = a.W >= 32;
  a.tilesX = cdiv(a.W, wide ? 32 : 16);
  a.tilesY = cdiv(a.H, wide ? 4 : 8);
  // 64-row tiles when the GEMM M dimension is small, when 128-row tiles would leave most CUs with one workgroup, or when
  // their count quantises badly over the 256 CUs (384 workgroups = 1.5 per CU runs at 75 %: the ResNet 256-channel
  // 14x14 layers)
  const long long wg128 = (long long)cdiv(a.Cout, 128) * a.tilesX * a.tilesY * c3_plan_n(a);
  const bool narrow = c3_narrow_rule(a.Cout, wg128, true);
  a.gridM = cdiv(a.Cout, narrow ? 64 : 128);
  dim3 grid((unsigned)((long long)a.gridM * a.tilesX * a.tilesY * a.N));
#define C3_LAUNCH(TH_, TW_, BM_)                                                                               \
  do {                                                                                                       \
    if (a.up2x) hipLaunchKernelGGL((conv3x3_kernel<TH_, TW_, BM_, true>), grid, dim3(256), 0, st, a);         \
    else if (dil == 2) hipLaunchKernelGGL((conv3x3_kernel<TH_, TW_, BM_, false, 3, 1, 2>), grid, dim3(256), 0, st, a); \
    else hipLaunchKernelGGL((conv3x3_kernel<TH_, TW_, BM_, false>), grid, dim3(256), 0, st, a);              \
  } while (0)
  if (wide && !narrow) C3_LAUNCH(4, 32, 128);
  else if (wide) C3_LAUNCH(4, 32, 64);
  else if (!narrow) C3_LAUNCH(8, 16, 128);
  else C3_LAUNCH(8, 16, 64);
#undef C3_LAUNCH
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

// tile decision of the halo-patch launches above as text (avsep_conv_kernel_variant); M = GEMM rows, Ho x Wo = tile space
void c3_variant_text(int M, int Ho, int Wo, long long planN, bool flat, bool quantise, char* buf, size_t cap) {
  if (flat) {
    const long long wg128 = (long long)cdiv(M, 128) * cdiv(planN * Ho * Wo, 128);
    snprintf(buf, cap, "flat%d,BM%d", Wo, c3_narrow_rule(M, wg128, true) ? 64 : 128);
    return;
  }
  const bool wide = Wo >= 32;
  const long long wg128 = (long long)cdiv(M, 128) * cdiv(Wo, wide ? 32 : 16) * cdiv(Ho, wide ? 4 : 8) * planN;
  snprintf(buf, cap, "%s,BM%d", wide ? "4x32" : "8x16", c3_narrow_rule(M, wg128, quantise) ? 64 : 128);
}
void c3_variant(const avsep_conv_desc* d, int mode, char* buf, size_t cap) {
  const bool flat = !d->up2x && d->C0 == d->Cin && c3_flat_width(d->H, d->W, d->dil) > 0 && c3_flat_enabled(d->algo);
  c3_variant_text(mode == 0 ? d->Cout : d->Cin, d->H, d->W, plan_batch(d), flat, true, buf, cap);
}
void c4_variant(const avsep_conv_desc* d, int mode, char* buf, size_t cap) {
  if (mode == 0) c3_variant_text(d->Cout, d->Ho, d->Wo, plan_batch(d), false, false, buf, cap);
  else c3_variant_text(d->Cin, d->H / 2, d->W / 2, plan_batch(d), false, false, buf, cap);     // each of the 4 parity classes
}

int c3_fwd(const avsep_conv_desc* d, const float* wp, const float* bias, float* y, double* stats, hipStream_t st) {
  C3Args a{};
  a.N = d->N; a.planN = d->plan_n; a.algo = d->algo; a.Cin = d->Cin; a.H = d->H; a.W = d->W; a.Cout = d->Cout;
  a.C0 = d->C0; a.C1 = d->Cin - d->C0; a.act0 = d->act0; a.act1 = d->act1; a.up2x = d->up2x;
  a.Hs = d->up2x ? d->H / 2 : d->H; a.Ws = d->up2x ? d->W / 2 : d->W;
  a.rh = (d->up2x && d->H > 1) ? (float)(a.Hs - 1) / (float)(d->H - 1) : 0.f;
  a.rw = (d->up2x && d->W > 1) ? (float)(a.Ws - 1) / (float)(d->W - 1) : 0.f;
  a.x0 = d->x0; a.x1 = d->x1; a.sc0 = d->scale0; a.sh0 = d->shift0; a.sc1 = d->scale1; a.sh1 = d->shift1;
  a.wp = wp; a.wp_ld = roundup(d->Cout, 128); a.out = y; a.bias = bias; a.stats = stats;
  return c3_launch(a, st, d->dil);
}

// dX[N,Cin,H,W] = conv3x3(dY[N,Cout,H,W], flipped/transposed weights)
int c3_dgrad(const avsep_conv_desc* d, const float* wp, const float* dy, float* dx, hipStream_t st) {
  C3Args a{};
  a.N = d->N; a.planN = d->plan_n; a.algo = d->algo; a.Cin = d->Cout; a.H = d->H; a.W = d->W; a.Cout = d->Cin;
  a.C0 = d->Cout; a.C1 = 0; a.Hs = d->H; a.Ws = d->W;
  a.x0 = dy; a.wp = wp; a.wp_ld = roundup(d->Cin, 128); a.out = dx;
  return c3_launch(a, st, d->dil);      // the flipped-weight identity holds for any dilation with pad == dil
}

// ===========================================================================
// 4x4 / stride 2 / pad 1 (the U-Net encoder's down convs, audio_net.py:57-58,170-171) on the same kernel.
//   forward : KS = 4, S = 2, one channel pair per K-tile (32 K rows), de-interleaved patch columns.
//   dgrad   : the transposed conv splits into the 4 parity classes (ph, pw) of the input pixel; class (ph, pw)
//             is a 2x2-tap stride-1 conv over dY (taps kh = 3-2*th | 2-2*th, pad = 1 | 0 for ph = 0 | 1; same
//             for columns) whose outputs are stored at (2a+ph, 2b+pw): 4 launches of the KS = 2 instantiation,
//             no MFMA work on structural zeros.
// ===========================================================================
// packed operand rows r = ((cpair*NT + tap)*2 + parity), column = GEMM M index
//   mode 0 (forward, NT = 16): in-channel 2*cpair+parity, w[co][ci][tap],                       column co
//   mode 1 (dgrad,  NT = 4, class cls = 2*ph+pw at row offset cls*Cout*4): "in"-channel = co,   column ci,
//           tap (th, tw) -> (kh, kw) = (ph ? 2-2*th : 3-2*th, pw ? 2-2*tw : 3-2*tw)
__global__ void c4_pack_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin, int rows, int ld,
                               int mode) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)rows * ld) return;
  int row = (int)(i / ld), col = (int)(i % ld);
  float v = 0.f;
  if (mode == 0) {
    int parity = row & 1, q = row >> 1, tap = q % 16, ch = 2 * (q / 16) + parity;
    if (ch < Cin && col < Cout) v = w[((long long)col * Cin + ch) * 16 + tap];
  } else {
    int per = Cout * 4, cls = row / per, rr = row % per;
    int parity = rr & 1, q = rr >> 1, tap = q % 4, ch = 2 * (q / 4) + parity;
    int ph = cls >> 1, pw = cls & 1, th = tap >> 1, tw = tap & 1;
    int kh = ph ? 2 - 2 * th : 3 - 2 * th, kw = pw ? 2 - 2 * tw : 3 - 2 * tw;
    if (ch < Cout && col < Cin) v = w[((long long)ch * Cin + col) * 16 + kh * 4 + kw];
  }
  out[i] = v;
}

// forward: Cin, C0 even, Wo >= 16; dgrad: Cout % 8 == 0, even H and W (all four classes have the same extent)
bool c4_applicable(const avsep_conv_desc* d, int mode) {
  if (!(d->KH == 4 && d->KW == 4 && d->stride == 2 && d->pad == 1 && d->dil == 1) || d->up2x) return false;
  if (d->Wo < 16 || d->Ho < 4 || d->N > 65535) return false;
  if (mode == 0) return d->Cin % 2 == 0 && d->C0 % 2 == 0 && d->Cout >= 32;
  return d->Cout % 8 == 0 && d->Cin >= 32 && (d->H & 1) == 0 && (d->W & 1) == 0;
}
size_t c4_packed_floats(const avsep_conv_desc* d, int mode) {
  int rows = (mode == 0 ? d->Cin : d->Cout) * 16, ld = roundup(mode == 0 ? d->Cout : d->Cin, 128);
  return (size_t)rows * ld;
}
int c4_pack(const avsep_conv_desc* d, const float* w, float* packed, int mode, hipStream_t st) {
  int rows = (mode == 0 ? d->Cin : d->Cout) * 16, ld = roundup(mode == 0 ? d->Cout : d->Cin, 128);
  long long total = (long long)rows * ld;
  hipLaunchKernelGGL(c4_pack_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, w, packed, d->Cout, d->Cin, rows, ld, mode);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

// tile choice shared by the forward and the dgrad classes; KS_/S_/CK_ select the instantiation
template <int KS_, int S_, int CK_>
static int c4_launch(C3Args& a, hipStream_t st) {
  const bool wide = a.Wo >= 32;
  a.tilesX = cdiv(a.Wo, wide ? 32 : 16);
  a.tilesY = cdiv(a.Ho, wide ? 4 : 8);
  const bool narrow = c3_narrow_rule(a.Cout, (long long)cdiv(a.Cout, 128) * a.tilesX * a.tilesY * c3_plan_n(a), false);
  a.gridM = cdiv(a.Cout, narrow ? 64 : 128);
  dim3 grid((unsigned)((long long)a.gridM * a.tilesX * a.tilesY * a.N));
  if (wide && !narrow) hipLaunchKernelGGL((conv3x3_kernel<4, 32, 128, false, KS_, S_, 1, CK_>), grid, dim3(256), 0, st, a);
  else if (wide) hipLaunchKernelGGL((conv3x3_kernel<4, 32, 64, false, KS_, S_, 1, CK_>), grid, dim3(256), 0, st, a);
  else if (!narrow) hipLaunchKernelGGL((conv3x3_kernel<8, 16, 128, false, KS_, S_, 1, CK_>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((conv3x3_kernel<8, 16, 64, false, KS_, S_, 1, CK_>), grid, dim3(256), 0, st, a);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

int c4_fwd(const avsep_conv_desc* d, const float* wp, const float* bias, float* y, double* stats, hipStream_t st) {
  C3Args a{};
  a.N = d->N; a.planN = d->plan_n; a.algo = d->algo; a.Cin = d->Cin; a.H = d->H; a.W = d->W; a.Cout = d->Cout;
  a.C0 = d->C0; a.C1 = d->Cin - d->C0; a.act0 = d->act0; a.act1 = d->act1; a.up2x = 0;
  a.Hs = d->H; a.Ws = d->W;
  a.x0 = d->x0; a.x1 = d->x1; a.sc0 = d->scale0; a.sh0 = d->shift0; a.sc1 = d->scale1; a.sh1 = d->shift1;
  a.wp = wp; a.wp_ld = roundup(d->Cout, 128); a.out = y; a.bias = bias; a.stats = stats;
  a.Ho = d->Ho; a.Wo = d->Wo; a.padh = a.padw = 1; a.os = 1; a.ooh = a.oow = 0; a.OHs = d->Ho; a.OWs = d->Wo;
  return c4_launch<4, 2, 2>(a, st);
}

// dX[N,Cin,H,W] from dY[N,Cout,Ho,Wo]: four 2x2-tap stride-1 convs over dY, one per input-pixel parity class
int c4_dgrad(const avsep_conv_desc* d, const float* wp, const float* dy, float* dx, hipStream_t st) {
  const int ld = roundup(d->Cin, 128);
  for (int cls = 0; cls < 4; ++cls) {
    const int ph = cls >> 1, pw = cls & 1;
    C3Args a{};
    a.N = d->N; a.planN = d->plan_n; a.algo = d->algo; a.Cin = d->Cout; a.H = d->Ho; a.W = d->Wo; a.Cout = d->Cin;     // the conv runs over dY
    a.C0 = d->Cout; a.C1 = 0; a.Hs = d->Ho; a.Ws = d->Wo;
    a.x0 = dy; a.wp = wp + (size_t)cls * d->Cout * 4 * ld; a.wp_ld = ld; a.out = dx;
    a.Ho = d->H / 2; a.Wo = d->W / 2; a.padh = ph ? 0 : 1; a.padw = pw ? 0 : 1;
    a.os = 2; a.ooh = ph; a.oow = pw; a.OHs = d->H; a.OWs = d->W;
    int rc = c4_launch<2, 1, 8>(a, st);     // 8 channels per K-tile: 16 k-steps per staged patch (4 left the MFMA pipe 47-63 % busy)
    if (rc) return rc;
  }
  return AVSEP_OK;
}

// ===========================================================================
// STFT as a 1 x 4 convolution (stft.hip): with the padded waveforms stored hop-transposed, XT[c][row][j] =
// padded[row][hop*j + c], frame f of row r is  sum_{j<4, c<hop} basis[m][hop*j + c] * XT[c][r][f + j]  — a
// KH = 1, KW = 4, stride-1 conv with Cin = hop channels over an [R x (frames+3)] image, i.e. the halo-patch kernel
// instead of the im2col gather (46 -> ~110 TFLOP/s on the [1024 x 1022] x [1022 x 24576] DFT of a batch).
// out[m][r][f], m = bin | bins + bin.
// ===========================================================================
int c1x4_stft_fwd(const float* xt, const float* wp, float* out, int R, int NH, int hop, int cout, int frames,
                  hipStream_t st) {
  C3Args a{};
  a.N = 1; a.Cin = hop; a.H = R; a.W = NH; a.Cout = cout;
  a.C0 = hop; a.C1 = 0; a.Hs = R; a.Ws = NH;
  a.x0 = xt; a.wp = wp; a.wp_ld = roundup(cout, 128); a.out = out;
  a.Ho = R; a.Wo = frames; a.padh = a.padw = 0; a.os = 1; a.ooh = a.oow = 0; a.OHs = R; a.OWs = frames;
  a.tilesX = cdiv(frames, 32); a.tilesY = cdiv(R, 4); a.gridM = cdiv(cout, 128);
  dim3 grid((unsigned)((long long)a.gridM * a.tilesX * a.tilesY));
  hipLaunchKernelGGL((conv3x3_kernel<4, 32, 128, false, 1, 1, 1, 4, 1, 4>), grid, dim3(256), 0, st, a);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

// ===========================================================================
// Weight gradient of the 3x3/p1 conv (stride 1, or stride 2: ResNet layer2.0 / layer3.0 conv1) with the same LDS halo patch.
//   dW[co][ci][tap] = sum_pix dY[co][pix] * X[ci][S*pix + tap]
// GEMM: M = co (128 per workgroup), N = (tap, ci) with 32 input channels on the lanes and the 9 taps as
// 9 MFMA column tiles, K = pixels of a TH x TW tile (MFMA lane half = pixel parity).  8 waves: wave w owns
// output rows [32(w&3), +32) x taps {0..4} (w<4) or {5..8} (w>=4) — waves w and w+4 share a SIMD, so every
// SIMD carries 9 MFMAs per k-step; <=80 accumulator registers leave room for 2 workgroups (4 waves/SIMD).
// Per k-step 1 A read + 4-5 B reads; every B address is  lane*PS + parity + const(tap, pixel pair).
// A workgroup sweeps `tiles_per_split` pixel tiles and writes one partial slab; a reduce kernel sums the
// slabs (deterministic, no atomics).
// ===========================================================================
struct W3Args {
  int N, Cin, H, W, Cout, Ho, Wo;     // H x W: input map, Ho x Wo: dY (equal at stride 1)
  int C0, C1, act0, act1, up2x, Hs, Ws;
  float rh, rw;
  const float *x0, *x1, *sc0, *sh0, *sc1, *sh1;
  const float* dy;
  float* out;
  int tilesX, tilesY, gridM, gridC, tiles_per_split;
  int tapmajor;   // 1: `out` is a slab buffer [split][tap][Cout][Cin] (lanes = ci -> 128-byte stores); 0: OIHW
};

constexpr int W3_CC = 32;   // input channels per workgroup (one MFMA column tile per tap)

__device__ __forceinline__ float w3_src(const W3Args& a, int n, int c, int hs, int ws) {
  float v;
  if (c < a.C0) {
    v = a.x0[(((long long)n * a.C0 + c) * a.Hs + hs) * a.Ws + ws];
    if (a.sc0) v = fmaf(v, a.sc0[c], a.sh0[c]);
    v = act_apply(v, a.act0);
  } else {
    int c1 = c - a.C0;
    v = a.x1[(((long long)n * a.C1 + c1) * a.Hs + hs) * a.Ws + ws];
    if (a.sc1) v = fmaf(v, a.sc1[c1], a.sh1[c1]);
    v = act_apply(v, a.act1);
  }
  return v;
}

// one pixel tile of the wgrad GEMM for a wave that owns the NTAP consecutive taps starting at TAP0.  TAP0 is a template
// parameter so that every B address is ONE per-lane base register + an immediate (with a run-time tap0 each tap
// needs its own address register and a v_add per read: measured 1.35 VALU instructions per MFMA in this loop, and the
// f32 MFMA shares the vector ALUs).
template <int TH, int TW, int PW, int MAXT, int TAP0, int NTAP, int DIL = 1, int S = 1>
__device__ __forceinline__ void w3_mfma_tile(const float* Ap, const float* Bp, f32x16 (&acc)[MAXT]) {
#pragma unroll 1
  for (int py = 0; py < TH; ++py) {
    const float* ar = Ap + py * TW;
    const float* br = Bp + py * S * PW;
    float av[2], bv[2][NTAP];                               // operands of the next k-step are read one step ahead
    av[0] = ar[0];
#pragma unroll
    for (int j = 0; j < NTAP; ++j) bv[0][j] = br[(((TAP0 + j) / 3) * PW + (TAP0 + j) % 3) * DIL];   // (+ S * px below)
#pragma unroll 4
    for (int px = 0; px < TW; px += 2) {
      const int cur = (px >> 1) & 1;
      if (px + 2 < TW) {
        av[cur ^ 1] = ar[px + 2];
#pragma unroll
        for (int j = 0; j < NTAP; ++j) bv[cur ^ 1][j] = br[(((TAP0 + j) / 3) * PW + (TAP0 + j) % 3) * DIL + (px + 2) * S];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NTAP; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur], bv[cur][j], acc[j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// DIL: dilation (pad = DIL); A2: rows of dY are only 8-byte aligned (W % 4 == 2, e.g. the 14x14 ResNet maps), the A tile
// is then loaded as pairs of float2.  The U-Net instantiations are <.., 1, false>.
template <int TH, int TW, bool UP2X, int BM, int DIL = 1, bool A2 = false, int S = 1>
__global__ __launch_bounds__(512, 2) void wgrad3x3_kernel(W3Args a) {
  static_assert(!(UP2X && DIL != 1), "the fused upsample path is undilated");
  static_assert(S == 1 || (DIL == 1 && !UP2X), "stride 2: plain 3x3 / pad 1");
  constexpr int NPIX = TH * TW, PH = (TH - 1) * S + 2 * DIL + 1, PW = (TW - 1) * S + 2 * DIL + 1;
  constexpr int PS = (PH * PW) | 1;                       // odd per-channel stride: lanes = channels -> no bank conflicts
  constexpr int LDA = NPIX + 1;
  constexpr int NT = 512;
  constexpr int NP = W3_CC * PH * PW, PE = (NP + NT - 1) / NT;
  constexpr int NA4 = BM * NPIX / 4, AE = NA4 / NT;
  static_assert(NA4 % NT == 0, "A tile must split evenly");
  // ONE array, the patch first: ds_read offsets are 16-bit immediates, and the 4-5 B reads per k-step (patch) must
  // stay below 64 KB so that they need no address arithmetic (the A tile behind it is read once per k-step)
  __shared__ float smem[2 * W3_CC * PS + 2 * BM * LDA];
  float (*Ps)[W3_CC * PS] = reinterpret_cast<float (*)[W3_CC * PS]>(smem);
  float (*As)[BM * LDA] = reinterpret_cast<float (*)[BM * LDA]>(smem + 2 * W3_CC * PS);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lk = lane >> 5;
  // 1-D grid, XCD-aware (round 5): the workgroups of one K-split read the same pixels and share an L2
  const int t_ = xcd_remap(blockIdx.x, gridDim.x), per_split_ = a.gridM * a.gridC;
  const int split = t_ / per_split_, lin_ = t_ % per_split_;
  const int mt = lin_ % a.gridM, ct = lin_ / a.gridM;
  const int m0 = mt * BM, c0 = ct * W3_CC;
  const int tiles_img = a.tilesX * a.tilesY, tiles_all = tiles_img * a.N;
  const int t_begin = split * a.tiles_per_split, t_end = min(tiles_all, t_begin + a.tiles_per_split);
  const long long HW = (long long)a.Ho * a.Wo;            // dY plane

  // BM=128: 4 row blocks x tap groups {0-4},{5-8} (waves w and w+4 share a SIMD: 9 MFMAs per SIMD and k-step);
  // BM=64: 2 row blocks x tap groups {0-2},{3,4},{5,6},{7,8}
  constexpr int RB = BM / 32, MAXT = (BM == 128) ? 5 : 3;
  const int wrow = wave % RB, tg = __builtin_amdgcn_readfirstlane(wave / RB);
  const int tap0 = BM == 128 ? (tg ? 5 : 0) : (tg == 0 ? 0 : 1 + 2 * tg);
  const int ntap = BM == 128 ? (tg ? 4 : 5) : (tg == 0 ? 3 : 2);
  f32x16 acc[MAXT];
#pragma unroll
  for (int j = 0; j < MAXT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // Staging is software-pipelined through registers (issue(t+1) -> MFMAs(t) -> finish(t+1)) and double-buffered
  // in LDS.  The f32 MFMA shares the vector ALUs, so the per-element decode is hoisted out of the tile loop:
  // per tile only (n, h0, w0) change.
  const bool first_src = c0 < a.C0;                        // block-uniform (C0 % 32 == 0 is checked on the host)
  const float* xsrc = first_src ? a.x0 : a.x1;
  const int Csrc = first_src ? a.C0 : a.C1, csrc0 = first_src ? c0 : c0 - a.C0;
  const float* scs = first_src ? a.sc0 : a.sc1;
  const float* shs = first_src ? a.sh0 : a.sh1;
  const int act_s = first_src ? a.act0 : a.act1;
  const float slope_s = act_slope(act_s);   // branch-free activation in the register-pipelined staging
  const long long sHW = (long long)a.Hs * a.Ws;
  // per staged element ONE packed word: LDS slot (bits 0-15) | tile row (16-19) | tile column (20-25) | valid (31).
  // (separate registers for these cost ~26 VGPRs and pushed the 128-row instantiations into scratch)
  static_assert(BM * LDA < 65536 && W3_CC * PS < 65536 && TH < 16 && PH < 16 && PW < 64, "packed staging word");
  int a_goff[AE];
  unsigned a_pk[AE];
#pragma unroll
  for (int e = 0; e < AE; ++e) {
    int idx = tid + NT * e;
    int q = idx % (TW / 4), r = (idx / (TW / 4)) % TH, co = idx / (NPIX / 4);
    a_goff[e] = min(m0 + co, a.Cout - 1) * (int)HW + r * a.Wo + 4 * q;   // Cout*H*W < 2^31 is checked on the host
    a_pk[e] = (unsigned)(co * LDA + r * TW + 4 * q) | (unsigned)r << 16 | (unsigned)(4 * q) << 20 |
              (m0 + co < a.Cout ? 0x80000000u : 0u);
  }
  int p_goff[PE];
  unsigned p_pk[PE];
  float p_sc[PE], p_sh[PE];
#pragma unroll
  for (int e = 0; e < PE; ++e) {
    int idx = min(tid + NT * e, NP - 1);
    int cc = idx / (PH * PW), r = (idx % (PH * PW)) / PW, col = idx % PW;
    const bool chok = (PE * NT == NP || tid + NT * e < NP) && c0 + cc < a.Cin;
    int cs = min(csrc0 + cc, Csrc - 1);
    p_goff[e] = cs * (int)sHW + (r - DIL) * a.Ws + (col - DIL);
    p_pk[e] = (unsigned)(cc * PS + r * PW + col) | (unsigned)r << 16 | (unsigned)col << 20 | (chok ? 0x80000000u : 0u);
    p_sc[e] = scs ? scs[cs] : 1.f;
    p_sh[e] = scs ? shs[cs] : 0.f;
  }
  f32x4 areg[AE];
  float praw[PE];
  unsigned amask = 0, pmask = 0;
  auto issue = [&](int t) __attribute__((always_inline)) {
    const int n = t / tiles_img, tt = t % tiles_img, h0 = (tt / a.tilesX) * TH, w0 = (tt % a.tilesX) * TW;
    const float* dyb = a.dy + (long long)n * a.Cout * HW + (long long)h0 * a.Wo + w0;
    const float* xb = xsrc + (long long)n * Csrc * sHW + (long long)(h0 * S) * a.Ws + w0 * S;
    amask = pmask = 0;
#pragma unroll
    for (int e = 0; e < AE; ++e) {
      unsigned pk = a_pk[e];
      asm volatile("" : "+v"(pk));          // opaque: keeps the decode inside the tile loop (LICM would re-expand it into registers)
      const int a_r = (pk >> 16) & 15, a_c = (pk >> 20) & 63;
      const bool ok = (pk >> 31) && h0 + a_r < a.Ho && w0 + a_c < a.Wo;
      if constexpr (!A2) {
        areg[e] = *reinterpret_cast<const f32x4*>(ok ? dyb + a_goff[e] : a.dy);
      } else {                 // the second pair of the quad may lie past the row end
        const bool ok2 = ok && w0 + a_c + 2 < a.Wo;
        const float2 lo = *reinterpret_cast<const float2*>(ok ? dyb + a_goff[e] : a.dy);
        const float2 hi = *reinterpret_cast<const float2*>(ok2 ? dyb + a_goff[e] + 2 : a.dy);
        areg[e] = f32x4{lo.x, lo.y, ok2 ? hi.x : 0.f, ok2 ? hi.y : 0.f};
      }
      amask |= (unsigned)ok << e;
    }
#pragma unroll
    for (int e = 0; e < PE; ++e) {
      unsigned pk = p_pk[e];
      asm volatile("" : "+v"(pk));
      const int p_r = (int)((pk >> 16) & 15) - DIL, p_col = (int)((pk >> 20) & 63) - DIL;
      const bool ok = (pk >> 31) && (unsigned)(h0 * S + p_r) < (unsigned)a.H && (unsigned)(w0 * S + p_col) < (unsigned)a.W;
      praw[e] = *(ok ? xb + p_goff[e] : xsrc);
      pmask |= (unsigned)ok << e;
    }
  };
  auto finish = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < AE; ++e) {
      const bool ok = (amask >> e) & 1u;
      unsigned pk = a_pk[e];
      asm volatile("" : "+v"(pk));
      float* d = &As[buf][pk & 0xffffu];
      d[0] = ok ? areg[e].x : 0.f; d[1] = ok ? areg[e].y : 0.f; d[2] = ok ? areg[e].z : 0.f; d[3] = ok ? areg[e].w : 0.f;
    }
#pragma unroll
    for (int e = 0; e < PE; ++e) {
      float v = act_by_slope(fmaf(praw[e], p_sc[e], p_sh[e]), slope_s);
      unsigned pk = p_pk[e];
      asm volatile("" : "+v"(pk));
      if (PE * NT == NP || tid + NT * e < NP) Ps[buf][pk & 0xffffu] = ((pmask >> e) & 1u) ? v : 0.f;
    }
  };
  // bilinear-upsampled virtual input: staged directly (4 corner loads per element do not fit the register pipeline)
  auto stage_up2x = [&](int t, int buf) __attribute__((always_inline)) {
    const int n = t / tiles_img, tt = t % tiles_img, h0 = (tt / a.tilesX) * TH, w0 = (tt % a.tilesX) * TW;
#pragma unroll 3
    for (int e = 0; e < PE; ++e) {
      int idx = tid + NT * e;
      float v = 0.f;
      if (PE * NT == NP || idx < NP) {
        int cc = idx / (PH * PW), r = (idx % (PH * PW)) / PW, col = idx % PW;
        int gh = h0 - 1 + r, gw = w0 - 1 + col, c = c0 + cc;
        if (c < a.Cin && (unsigned)gh < (unsigned)a.H && (unsigned)gw < (unsigned)a.W) {
          float fh = a.rh * (float)gh, fw = a.rw * (float)gw;
          int hh0 = (int)fh, ww0 = (int)fw;
          int hh1 = hh0 + (hh0 < a.Hs - 1), ww1 = ww0 + (ww0 < a.Ws - 1);
          float lh = fh - (float)hh0, lw = fw - (float)ww0;
          float v00 = w3_src(a, n, c, hh0, ww0), v01 = w3_src(a, n, c, hh0, ww1);
          float v10 = w3_src(a, n, c, hh1, ww0), v11 = w3_src(a, n, c, hh1, ww1);
          v = (1.f - lh) * ((1.f - lw) * v00 + lw * v01) + lh * ((1.f - lw) * v10 + lw * v11);
        }
        Ps[buf][(idx / (PH * PW)) * PS + idx % (PH * PW)] = v;
      }
    }
  };

  if (t_begin < t_end) {
    issue(t_begin);
    finish(0);
    if constexpr (UP2X) stage_up2x(t_begin, 0);
  }
  __syncthreads();
  for (int t = t_begin; t < t_end; ++t) {
    const int buf = (t - t_begin) & 1;
    if (t + 1 < t_end) issue(t + 1);
    const float* Ap = As[buf] + (wrow * 32 + li) * LDA + lk;
    const float* Bp = Ps[buf] + li * PS + lk * S;
    // one straight-line MFMA stream per tap group (wave-uniform switch; compile-time taps -> immediate LDS offsets)
    if constexpr (BM == 128) {
      if (tg == 0) w3_mfma_tile<TH, TW, PW, MAXT, 0, 5, DIL, S>(Ap, Bp, acc);
      else w3_mfma_tile<TH, TW, PW, MAXT, 5, 4, DIL, S>(Ap, Bp, acc);
    } else {
      if (tg == 0) w3_mfma_tile<TH, TW, PW, MAXT, 0, 3, DIL, S>(Ap, Bp, acc);
      else if (tg == 1) w3_mfma_tile<TH, TW, PW, MAXT, 3, 2, DIL, S>(Ap, Bp, acc);
      else if (tg == 2) w3_mfma_tile<TH, TW, PW, MAXT, 5, 2, DIL, S>(Ap, Bp, acc);
      else w3_mfma_tile<TH, TW, PW, MAXT, 7, 2, DIL, S>(Ap, Bp, acc);
    }
    if (t + 1 < t_end) {
      finish(buf ^ 1);
      if constexpr (UP2X) stage_up2x(t + 1, buf ^ 1);
    }
    __syncthreads();
  }
  // epilogue: row = output channel, MFMA column = input channel (lane), column tile = tap
  const int ci = c0 + li;
#pragma unroll
  for (int j = 0; j < MAXT; ++j)
    if (j < ntap) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int co = m0 + wrow * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (co < a.Cout && ci < a.Cin) {
          // OIHW puts the 32 lanes (ci) 36 bytes apart: every 4-byte store its own memory transaction (measured
          // 6.6x write amplification on the slabs).  Slabs are therefore tap-major; w3_reduce_kernel transposes.
          if (a.tapmajor) a.out[(((long long)split * 9 + tap0 + j) * a.Cout + co) * a.Cin + ci] = acc[j][r];
          else a.out[((long long)co * a.Cin + ci) * 9 + tap0 + j] = acc[j][r];
        }
      }
    }
}

bool w3_applicable(const avsep_conv_desc* d) {
  if (d->KH == 3 && d->KW == 3 && d->stride == 2 && d->dil == 1 && d->pad == 1)       // ResNet layer2.0 / layer3.0 conv1
    return !d->up2x && d->C0 == d->Cin && (d->H & 1) == 0 && (d->W & 1) == 0 && d->Wo >= 12 && (d->Wo & 1) == 0 && d->Cout > 4 &&
           d->Cin >= 32 && d->N <= 65535 && (long long)(d->Cout > d->Cin ? d->Cout : d->Cin) * d->H * d->W < 0x7fffffffLL;
  if (!(d->KH == 3 && d->KW == 3 && d->stride == 1 && (d->dil == 1 || d->dil == 2) && d->pad == d->dil)) return false;
  if (d->up2x && (d->dil != 1 || (d->W & 3))) return false;
  return d->W >= 12 && d->H >= 2 && (d->W & 1) == 0 && d->Cout > 4 && d->Cin >= 32 && d->N <= 65535 &&
         (d->C0 == d->Cin || d->C0 % W3_CC == 0) &&
         (long long)(d->Cout > d->Cin ? d->Cout : d->Cin) * d->H * d->W < 0x7fffffffLL;
}

struct W3Plan { int tilesX, tilesY, gridM, gridC, splits, tps; bool wide; };
static W3Plan w3_plan(const avsep_conv_desc* d) {
  W3Plan p;
  // tiles are tiles of dY.  Wide (2 x 32) tiles: not for the dilated patch (it would not leave room for two workgroups per
  // CU); stride 2 takes 2 x 16 tiles (5 x 33 patch)
  p.wide = d->Wo >= 32 && d->dil == 1 && d->stride == 1;
  p.tilesX = cdiv(d->Wo, p.wide ? 32 : 16);
  p.tilesY = cdiv(d->Ho, (p.wide || d->stride == 2) ? 2 : 4);
  p.gridM = cdiv(d->Cout, d->Cout <= 64 ? 64 : 128);
  p.gridC = cdiv(d->Cin, W3_CC);
  long long tiles = (long long)p.tilesX * p.tilesY * d->N;
  int want = cdiv(768, p.gridM * p.gridC);                 // ~3 workgroups per CU in flight
  long long maxs = tiles / 4 > 0 ? tiles / 4 : 1;           // at least 4 pixel tiles per slab
  int splits = (int)(want < maxs ? want : maxs);
  if (splits < 1) splits = 1;
  p.tps = (int)((tiles + splits - 1) / splits);
  p.splits = (int)((tiles + p.tps - 1) / p.tps);
  return p;
}
size_t w3_workspace_floats(const avsep_conv_desc* d) {
  W3Plan p = w3_plan(d);
  return p.splits > 1 ? (size_t)p.splits * d->Cout * d->Cin * 9 : 0;
}
// dw[cc][tap] = sum_z ws[z][tap][cc], cc = co*Cin + ci: coalesced plane reads, 36 contiguous bytes written per (co, ci).
// Block = 4 split groups x 64 values of cc: a thread sums every 4th slab, the four partial sums meet in LDS in a fixed
// order (deterministic).  One thread per cc walking all slabs alone left the 64-channel layers (P = 4096: 16 blocks, hundreds
// of slabs) latency-bound: 0.82 ms per fp32 step over 44 launches.
__global__ __launch_bounds__(256) void w3_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out, long long P, int S) {
  __shared__ float part[3][9][64];
  const int sg = threadIdx.x >> 6, l = threadIdx.x & 63;
  const long long cc = (long long)blockIdx.x * 64 + l;
  const bool live = cc < P;
  float s[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) s[t] = 0.f;
  if (live) {
#pragma unroll 2
    for (int z = sg; z < S; z += 4)
#pragma unroll
      for (int t = 0; t < 9; ++t) s[t] += ws[((long long)z * 9 + t) * P + cc];
  }
  if (sg > 0) {
#pragma unroll
    for (int t = 0; t < 9; ++t) part[sg - 1][t][l] = s[t];
  }
  __syncthreads();
  if (sg == 0 && live) {
#pragma unroll
    for (int t = 0; t < 9; ++t) out[cc * 9 + t] = (s[t] + part[0][t][l]) + (part[1][t][l] + part[2][t][l]);
  }
}
int w3_reduce(const float* ws, float* dw, long long P, int splits, hipStream_t st) {
  hipLaunchKernelGGL(w3_reduce_kernel, dim3(cdiv(P, 64)), dim3(256), 0, st, ws, dw, P, splits);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}
int w3_wgrad(const avsep_conv_desc* d, const float* dy, float* dw, float* ws, hipStream_t st) {
  W3Plan p = w3_plan(d);
  W3Args a{};
  a.N = d->N; a.Cin = d->Cin; a.H = d->H; a.W = d->W; a.Cout = d->Cout; a.Ho = d->Ho; a.Wo = d->Wo;
  a.C0 = d->C0; a.C1 = d->Cin - d->C0; a.act0 = d->act0; a.act1 = d->act1; a.up2x = d->up2x;
  a.Hs = d->up2x ? d->H / 2 : d->H; a.Ws = d->up2x ? d->W / 2 : d->W;
  a.rh = (d->up2x && d->H > 1) ? (float)(a.Hs - 1) / (float)(d->H - 1) : 0.f;
  a.rw = (d->up2x && d->W > 1) ? (float)(a.Ws - 1) / (float)(d->W - 1) : 0.f;
  a.x0 = d->x0; a.x1 = d->x1; a.sc0 = d->scale0; a.sh0 = d->shift0; a.sc1 = d->scale1; a.sh1 = d->shift1;
  a.dy = dy; a.out = p.splits > 1 ? ws : dw; a.tapmajor = p.splits > 1;
  a.tilesX = p.tilesX; a.tilesY = p.tilesY; a.gridM = p.gridM; a.gridC = p.gridC; a.tiles_per_split = p.tps;
  dim3 grid(p.gridM * p.gridC * p.splits);
#define W3_LAUNCH(TH_, TW_, UP_, DIL_, A2_)                                                                            \
  do {                                                                                                                 \
    if (d->Cout <= 64) hipLaunchKernelGGL((wgrad3x3_kernel<TH_, TW_, UP_, 64, DIL_, A2_>), grid, dim3(512), 0, st, a);  \
    else hipLaunchKernelGGL((wgrad3x3_kernel<TH_, TW_, UP_, 128, DIL_, A2_>), grid, dim3(512), 0, st, a);              \
  } while (0)
  const bool a2 = (d->Wo & 3) != 0;
#define W3_LAUNCH_S2(A2_)                                                                                                \
  do {                                                                                                                   \
    if (d->Cout <= 64) hipLaunchKernelGGL((wgrad3x3_kernel<2, 16, false, 64, 1, A2_, 2>), grid, dim3(512), 0, st, a);     \
    else hipLaunchKernelGGL((wgrad3x3_kernel<2, 16, false, 128, 1, A2_, 2>), grid, dim3(512), 0, st, a);                  \
  } while (0)
  if (d->stride == 2) {
    if (a2) W3_LAUNCH_S2(true); else W3_LAUNCH_S2(false);
  } else if (d->up2x) {
    if (p.wide) W3_LAUNCH(2, 32, true, 1, false);
    else W3_LAUNCH(4, 16, true, 1, false);
  } else if (d->dil == 1 && !a2) {                // the U-Net's instantiations
    if (p.wide) W3_LAUNCH(2, 32, false, 1, false);
    else W3_LAUNCH(4, 16, false, 1, false);
  } else if (d->dil == 1) {
    if (p.wide) W3_LAUNCH(2, 32, false, 1, true);
    else W3_LAUNCH(4, 16, false, 1, true);
  } else if (!a2) {
    W3_LAUNCH(4, 16, false, 2, false);
  } else {
    W3_LAUNCH(4, 16, false, 2, true);
  }
#undef W3_LAUNCH
#undef W3_LAUNCH_S2
  AVSEP_LAUNCH_CHECK();
  if (p.splits > 1) {
    long long P = (long long)d->Cout * d->Cin;
    hipLaunchKernelGGL(w3_reduce_kernel, dim3(cdiv(P, 64)), dim3(256), 0, st, ws, dw, P, p.splits);
    AVSEP_LAUNCH_CHECK();
  }
  return AVSEP_OK;
}
