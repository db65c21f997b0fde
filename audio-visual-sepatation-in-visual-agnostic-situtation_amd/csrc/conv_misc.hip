// fp32 halo-patch instantiations for the remaining convolution classes of the visual trunk (torchvision resnet18 behind
// models/vision_net.py:84-92), which used to run on the im2col kernel at 20-48 TFLOP/s:
//   3x3 / stride 2 / pad 1 (layer2.0 / layer3.0 conv1): forward on the stride-2 patch (de-interleaved columns, like the
//     4x4/s2 encoder convs); data gradient as four parity classes of the input pixel — class (ph, pw) is a
//     (ph ? 2 : 1) x (pw ? 2 : 1)-tap stride-1 conv over dY without padding, stored at (2a + ph, 2b + pw);
//   1x1 / stride 1 | 2 (the downsample convs): forward; data gradient = the 1x1 conv with transposed weights, for
//     stride 2 stored at the sampled pixels of a zero-filled dX.
// Weight gradients of these classes stay on the im2col kernel.
#include <stdlib.h>

#include "halo_kernel.h"

static inline bool cm_enabled(const avsep_conv_desc* d) { return !(d->algo & AVSEP_ALGO_NO_MISC_PATCH); }
static int cm_class(const avsep_conv_desc* d) {
  if (d->up2x || d->dil != 1) return 0;
  if (d->KH == 3 && d->KW == 3 && d->stride == 2 && d->pad == 1) return 5;
  if (d->KH == 1 && d->KW == 1 && (d->stride == 1 || d->stride == 2) && d->pad == 0) return 1;
  if (d->KH == 4 && d->KW == 4 && d->stride == 1 && d->pad == 0) return 6;      // the stem after space-to-depth (forward only)
  return 0;
}
bool cm_applicable(const avsep_conv_desc* d, int mode) {
  const int cls = cm_class(d);
  if (!cls || !cm_enabled(d) || d->N > 65535 || d->Wo < 12 || d->Ho < 4) return false;
  if (cls == 6 && mode != 0) return false;
  if (mode == 0) {
    if (cls == 5 || cls == 6) return d->Cin % 2 == 0 && d->C0 % 2 == 0 && d->Cout >= 32;
    return d->Cin % 16 == 0 && d->C0 % 16 == 0 && d->Cout >= 32;
  }
  if (d->Cin < 32) return false;
  if (cls == 5) return d->Cout % 16 == 0 && (d->H & 1) == 0 && (d->W & 1) == 0;
  return d->Cout % 16 == 0 && (d->stride == 1 || ((d->H & 1) == 0 && (d->W & 1) == 0));
}

// packed operand rows r = ((cpair*NT + tap)*2 + parity), column = GEMM M index (ld = roundup(M, 128))
//   mode 0: forward, NT = KH*KW taps, in-channel 2*cpair+parity, w[co][ci][tap], column co
//   mode 1: 1x1 data gradient: "in"-channel co, column ci
//   mode 3: 3x3/s2 data gradient: the class images (1, 2, 2, 4 taps) follow each other; "in"-channel co, column ci,
//           tap (th, tw) of class (ph, pw) -> (kh, kw) = (ph ? 2-2*th : 1, pw ? 2-2*tw : 1)
__global__ void cm_pack_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin, int KHW, int ld, int mode) {
  const long long kc = mode == 0 ? Cin : Cout;
  const long long total = kc * KHW * ld;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  long long row = i / ld;
  const int col = (int)(i % ld);
  float v = 0.f;
  if (mode != 3) {
    const int parity = (int)(row & 1), tap = (int)((row >> 1) % KHW), ch = 2 * (int)((row >> 1) / KHW) + parity;
    if (mode == 0) { if (ch < Cin && col < Cout) v = w[((long long)col * Cin + ch) * KHW + tap]; }
    else { if (ch < Cout && col < Cin) v = w[((long long)ch * Cin + col) * KHW + (KHW - 1 - tap)]; }
  } else {
    int cls, NT;
    const long long unit = (long long)Cout;                      // rows of one tap of a class image
    if (row < unit) { cls = 0; NT = 1; }
    else if (row < 3 * unit) { cls = 1; NT = 2; row -= unit; }
    else if (row < 5 * unit) { cls = 2; NT = 2; row -= 3 * unit; }
    else { cls = 3; NT = 4; row -= 5 * unit; }
    const int parity = (int)(row & 1), tap = (int)((row >> 1) % NT), ch = 2 * (int)((row >> 1) / NT) + parity;
    const int ph = cls >> 1, pw = cls & 1, ntw = pw ? 2 : 1, th = tap / ntw, tw = tap % ntw;
    const int kh = ph ? 2 - 2 * th : 1, kw = pw ? 2 - 2 * tw : 1;
    if (ch < Cout && col < Cin) v = w[((long long)ch * Cin + col) * 9 + kh * 3 + kw];
  }
  out[i] = v;
}
size_t cm_packed_floats(const avsep_conv_desc* d, int mode) {
  return (size_t)(mode == 0 ? d->Cin : d->Cout) * d->KH * d->KW * roundup(mode == 0 ? d->Cout : d->Cin, 128);
}
int cm_pack(const avsep_conv_desc* d, const float* w, float* packed, int mode, hipStream_t st) {
  const int ld = roundup(mode == 0 ? d->Cout : d->Cin, 128);
  const long long total = (long long)cm_packed_floats(d, mode);
  const int pmode = (mode == 1 && cm_class(d) == 5) ? 3 : mode;
  hipLaunchKernelGGL(cm_pack_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, w, packed, d->Cout, d->Cin, d->KH * d->KW, ld, pmode);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

template <int KH_, int KW_, int S_, int CK_>
static int cm_launch(C3Args& a, hipStream_t st) {
  const bool wide = a.Wo >= 32;
  a.tilesX = cdiv(a.Wo, wide ? 32 : 16);
  a.tilesY = cdiv(a.Ho, wide ? 4 : 8);
  const bool narrow = c3_narrow_rule(a.Cout, (long long)cdiv(a.Cout, 128) * a.tilesX * a.tilesY * c3_plan_n(a), false);
  a.gridM = cdiv(a.Cout, narrow ? 64 : 128);
  dim3 grid((unsigned)((long long)a.gridM * a.tilesX * a.tilesY * a.N));
  if (wide && !narrow) hipLaunchKernelGGL((conv3x3_kernel<4, 32, 128, false, 3, S_, 1, CK_, KH_, KW_>), grid, dim3(256), 0, st, a);
  else if (wide) hipLaunchKernelGGL((conv3x3_kernel<4, 32, 64, false, 3, S_, 1, CK_, KH_, KW_>), grid, dim3(256), 0, st, a);
  else if (!narrow) hipLaunchKernelGGL((conv3x3_kernel<8, 16, 128, false, 3, S_, 1, CK_, KH_, KW_>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((conv3x3_kernel<8, 16, 64, false, 3, S_, 1, CK_, KH_, KW_>), grid, dim3(256), 0, st, a);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

void c3_variant_text(int M, int Ho, int Wo, long long planN, bool flat, bool quantise, char* buf, size_t cap);   // conv3x3.hip
void cm_variant(const avsep_conv_desc* d, int mode, char* buf, size_t cap) {
  if (mode == 0) c3_variant_text(d->Cout, d->Ho, d->Wo, plan_batch(d), false, false, buf, cap);
  else if (cm_class(d) == 1) c3_variant_text(d->Cin, d->Ho, d->Wo, plan_batch(d), false, false, buf, cap);
  else c3_variant_text(d->Cin, d->H / 2, d->W / 2, plan_batch(d), false, false, buf, cap);
}

int cm_fwd(const avsep_conv_desc* d, const float* wp, const float* bias, float* y, double* stats, hipStream_t st) {
  C3Args a{};
  a.N = d->N; a.planN = d->plan_n; a.Cin = d->Cin; a.H = d->H; a.W = d->W; a.Cout = d->Cout;
  a.C0 = d->C0; a.C1 = d->Cin - d->C0; a.act0 = d->act0; a.act1 = d->act1; a.up2x = 0;
  a.Hs = d->H; a.Ws = d->W;
  a.x0 = d->x0; a.x1 = d->x1; a.sc0 = d->scale0; a.sh0 = d->shift0; a.sc1 = d->scale1; a.sh1 = d->shift1;
  a.wp = wp; a.wp_ld = roundup(d->Cout, 128); a.out = y; a.bias = bias; a.stats = stats;
  a.Ho = d->Ho; a.Wo = d->Wo; a.padh = a.padw = d->pad; a.os = 1; a.ooh = a.oow = 0; a.OHs = d->Ho; a.OWs = d->Wo;
  if (cm_class(d) == 5) return cm_launch<3, 3, 2, 2>(a, st);
  if (cm_class(d) == 6) return cm_launch<4, 4, 1, 2>(a, st);
  return d->stride == 1 ? cm_launch<1, 1, 1, 16>(a, st) : cm_launch<1, 1, 2, 16>(a, st);
}

int cm_dgrad(const avsep_conv_desc* d, const float* wp, const float* dy, float* dx, hipStream_t st) {
  const int ld = roundup(d->Cin, 128);
  C3Args a{};
  a.N = d->N; a.planN = d->plan_n; a.Cin = d->Cout; a.H = d->Ho; a.W = d->Wo; a.Cout = d->Cin;       // the conv runs over dY
  a.C0 = d->Cout; a.C1 = 0; a.Hs = d->Ho; a.Ws = d->Wo;
  a.x0 = dy; a.wp_ld = ld; a.out = dx; a.padh = a.padw = 0; a.OHs = d->H; a.OWs = d->W;
  if (cm_class(d) == 1) {
    a.wp = wp; a.Ho = d->Ho; a.Wo = d->Wo; a.os = d->stride; a.ooh = a.oow = 0;
    if (d->stride == 2 && hipMemsetAsync(dx, 0, (size_t)d->N * d->Cin * d->H * d->W * sizeof(float), st) != hipSuccess)
      return AVSEP_ERR_LAUNCH;
    return cm_launch<1, 1, 1, 16>(a, st);
  }
  size_t off = 0;
  for (int cls = 0; cls < 4; ++cls) {
    const int ph = cls >> 1, pw = cls & 1;
    a.wp = wp + off; a.Ho = d->H / 2; a.Wo = d->W / 2; a.os = 2; a.ooh = ph; a.oow = pw;
    int rc = ph ? (pw ? cm_launch<2, 2, 1, 8>(a, st) : cm_launch<2, 1, 1, 8>(a, st))
                : (pw ? cm_launch<1, 2, 1, 8>(a, st) : cm_launch<1, 1, 1, 16>(a, st));
    if (rc) return rc;
    off += (size_t)(ph ? 2 : 1) * (pw ? 2 : 1) * d->Cout * ld;
  }
  return AVSEP_OK;
}
