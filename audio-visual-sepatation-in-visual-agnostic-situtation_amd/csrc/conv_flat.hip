// Flat-pixel instantiations of the halo-patch kernel (halo_kernel.h, FW > 0) for the small square maps of the visual
// trunk: torchvision ResNet-18 BasicBlock 3x3 convs at 56x56 / 28x28 / 14x14 (/ 7x7 for resnet18fc), dilation 1 or 2
// (models/vision_net.py:96-109 `_nostride_dilate`), forward and data gradient (flipped weights, conv3x3.hip).
// A rectangular 8x16 tile covers a 14x14 (and a 28x28) map at 77 %, a 4x32 tile a 56x56 map at 87.5 %; the flat tile's
// 128 MFMA columns are 128 consecutive pixels of (n, h, w), so only the last tile of the batch is partial.
#include "halo_kernel.h"

// FW if (W, dil) has a flat instantiation, else 0.  Geometry requirements are checked by the caller (c3_applicable).
int c3_flat_width(int H, int W, int dil) {
  // measured on the trunk (B = 96 frames): 14x14 maps gain (256 ch: 67 -> 86 TFLOP/s forward, 69 -> 89 data gradient;
  // 256 -> 512: 72 -> 91 data gradient), 28x28 and 56x56 maps lose 2-6 % to the larger patch (their rectangular
  // tiles are already 87.5 % full) and keep the rectangular tiles
  if (H < W) return 0;
  if (dil == 1 && (W == 14 || W == 7)) return W;
  if (dil == 2 && W == 14) return W;
  return 0;
}

template <int FW, int DIL>
static void flat_launch_w(const C3Args& a, bool narrow, dim3 grid, hipStream_t st) {
  if (narrow) hipLaunchKernelGGL((conv3x3_kernel<0, 0, 64, false, 3, 1, DIL, 4, 3, 3, FW>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((conv3x3_kernel<0, 0, 128, false, 3, 1, DIL, 4, 3, 3, FW>), grid, dim3(256), 0, st, a);
}

// a: filled like c3_launch's (3x3, stride 1, pad = dil, single source, no upsample); sets the tiling fields
int c3_flat_launch(C3Args& a, int dil, hipStream_t st) {
  const long long P = (long long)a.N * a.H * a.W;
  if (P > 0x7fffffffLL || a.C1 != 0 || a.up2x) return AVSEP_ERR_ARG;
  a.tilesX = cdiv(P, 128);
  a.tilesY = 1;
  const long long wg128 = (long long)cdiv(a.Cout, 128) * cdiv(c3_plan_n(a) * a.H * a.W, 128);
  const bool narrow = c3_narrow_rule(a.Cout, wg128, true);
  a.gridM = cdiv(a.Cout, narrow ? 64 : 128);
  dim3 grid((unsigned)((long long)a.gridM * a.tilesX));
  const int fw = c3_flat_width(a.H, a.W, dil);
  if (dil == 1) {
    if (fw == 14) flat_launch_w<14, 1>(a, narrow, grid, st);
    else if (fw == 7) flat_launch_w<7, 1>(a, narrow, grid, st);
    else return AVSEP_ERR_ARG;
  } else {
    if (fw == 14) flat_launch_w<14, 2>(a, narrow, grid, st);
    else return AVSEP_ERR_ARG;
  }
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}
