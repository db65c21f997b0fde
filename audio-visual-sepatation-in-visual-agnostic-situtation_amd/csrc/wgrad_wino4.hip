// Weight gradient of the 3x3 / stride 1 / pad 1 convolutions in the Winograd F(4x4, 3x3) form, fp32:
//     dW = sum over 4x4 tiles of  G^T [ (A dY A^T) (.) (B^T d B) ] G          (the transpose of conv_wino4.hip's forward)
// dY = the 4x4 cotangent tile, d = the 6x6 input tile around it: 36 products per tile and (co, ci) where the direct form
// has 144 and the F(2x2) form of wgrad_wino.hip 64.  Per transform position xi this is a [Cout x tiles] x [tiles x Cin]
// GEMM whose K dimension is the TILE index: one MFMA k-step consumes two tiles, and both operands are transformed in the
// kernel.  A 512-thread workgroup owns 64 output x 32 input channels (36 x 2 accumulator tiles, 9 per wave = 144
// registers: wave (cb, q) has positions 9q .. 9q+8 of output-channel block cb) and sweeps the K-tiles (8 tiles = one
// 8x16-pixel region, or two 8x8 regions) of its K-split in two phases per K-tile:
//   T  every thread transforms one (output channel, tile) of dY — 16 values straight from global memory (four 16-byte row
//      loads issued a phase earlier), A y A^T in 80 vector instructions — into Yh[xi][co][tile], and one (input channel,
//      tile, half) of the input patch — B^T d B as in conv_wino4.hip, 72 instructions — into V[xi][tile][ci];
//   M  36 MFMAs per wave, both fragments conflict-free ds_read_b32; behind them the stores of the NEXT K-tile's raw input
//      patch into LDS (folded BatchNorm affine + activation + two-source concat on the way), the global loads of the
//      patch after that and the dY rows of the next K-tile.
// The f32 MFMA does not overlap vector instructions, so separating the transform phase from the MFMA phase costs no
// matrix-pipe time that an interleaved form would not also pay; it keeps ONE copy of Yh / V in LDS (83 + 37 KB), which
// is what makes 8-tile K-tiles fit.  The input patch is staged position-major: its 128 interior pixels per channel are
// 8 pieces per thread that can never leave the image (region sizes divide the map), the halo ring is 4-5 pieces whose
// validity (region on an image border) is one AND + compare + select per K-tile.
// Epilogue: M goes through LDS in four passes (as conv_wino4.hip), every thread applies G^T M G to one (co, ci) and
// writes its 9 taps into a tap-major partial slab [split][tap][Cout][Cin]; conv3x3.hip's w3_reduce sums the slabs.
#include <stdlib.h>
#include <type_traits>

#include "common.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int X4_BCO = 64, X4_BCI = 32;         // channels per workgroup
constexpr int X4_THREADS = 512;
constexpr int X4_KPS_MAX = 128;                 // K-tiles of one split (their origin records live in LDS; host-checked)
constexpr int X4_YS = 9;                        // Yh[xi][co][tile]: 8 tiles padded to 9 words (32 lanes = 32 channels hit 32 banks)
constexpr int X4_Y_FLOATS = 36 * X4_BCO * X4_YS;
constexpr int X4_V_FLOATS = 36 * 8 * X4_BCI;

struct X4Args {
  int N, C0, C1, Cin, H, W, Cout;
  int ryn, rxn;                                 // regions per image along y / x
  int nkt, kps;                                 // K-tiles in total / per split
  int gridM, gridC, act0, act1;
  const float *x0, *x1, *sc0, *sh0, *sc1, *sh1;
  const float* dy;
  float* out;                                   // slabs [split][tap][Cout][Cin]
};

// G groups (regions) of RH x RW tiles per K-tile (G*RH*RW = 8); PW = LDS row stride of a region's patch, GS = stride
// between the regions, PS = stride between channels (a multiple of 4 with PS mod 64 = 12 or 52: the b128 row reads of
// 16 lanes = 16 channels then cover the 64 banks); RAW: no affine and no activation on the input.
template <int G, int RH, int RW, int PW, int GS, int PS, bool RAW>
__global__ __launch_bounds__(X4_THREADS) void winow4_kernel(X4Args a) {
  static_assert(G * RH * RW == 8 && (G == 1 || G == 2), "8 tiles per K-tile");
  constexpr int NT = X4_THREADS, BCO = X4_BCO, BCI = X4_BCI;
  constexpr int RHP = 4 * RH, RWP = 4 * RW;                       // region in pixels
  constexpr int PHG = RHP + 2, PCG = RWP + 2;                     // its patch
  static_assert(PW % 4 == 0 && PW >= PCG && GS % 4 == 0 && GS >= PHG * PW && PS % 4 == 0 && PS >= G * GS, "patch strides");
  static_assert(G * RHP * RWP == 128, "128 interior pixels per channel");
  constexpr int NIS = BCI * 128 / NT;                             // interior pieces per thread: 8
  constexpr int NHG = 2 * PCG + 2 * (PHG - 2), NH = G * NHG;      // halo ring positions per region / per channel
  constexpr int NHS = (BCI * NH + NT - 1) / NT;                   // halo pieces per thread
  constexpr int P_FLOATS = BCI * PS;
  static_assert(X4_Y_FLOATS >= 36 * 16 * 32, "epilogue exchange fits the Yh buffer");
  __shared__ __attribute__((aligned(16))) float smem[X4_Y_FLOATS + X4_V_FLOATS + P_FLOATS];
  __shared__ int gtab[X4_KPS_MAX][G][4];        // per K-tile and region: {byte offset of its origin in x, in dy, border bits, 0}
  __shared__ f32x2 aff[BCI];
  float* const Ys = smem;
  float* const Vs = smem + X4_Y_FLOATS;
  float* const Ps = Vs + X4_V_FLOATS;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 31, lk = lane >> 5;
  // 1-D grid, XCD-aware: the gridM * gridC workgroups of one K-split read the same pixels (dY is shared by the gridC input-
  // channel blocks, the input patch by the gridM output-channel blocks); under round-robin dispatch they would land on eight
  // different XCDs and every L2 would fetch those pixels again — consecutive LOGICAL ids share an XCD instead
  const int t_ = xcd_remap(blockIdx.x, gridDim.x), per_split = a.gridM * a.gridC;
  const int split = t_ / per_split, lin_ = t_ % per_split;
  const int mt = lin_ % a.gridM, ct = lin_ / a.gridM;
  const int m0 = mt * BCO, c0 = ct * BCI;
  const int HW = a.H * a.W;
  const bool src1 = c0 >= a.C0;                 // the 32-channel input block lies in one source
  const float* const xs = src1 ? a.x1 : a.x0;
  const int Cs = src1 ? a.C1 : a.C0, cs0 = src1 ? c0 - a.C0 : c0;
  const float slope = act_slope(src1 ? a.act1 : a.act0);
  const int kt0 = split * a.kps, nk = min(a.kps, a.nkt - kt0);

  // ---- origin records of the split's K-tiles ----
  {
    const int per = a.ryn * a.rxn;
    for (int e = tid; e < nk * G; e += NT) {
      const int k = e / G, g = e % G, reg = (kt0 + k) * G + g;
      const int img = reg / per, ry = (reg % per) / a.rxn, rx = reg % a.rxn;
      const int y0 = ry * RHP, x0 = rx * RWP;
      // byte offsets of pixel (y0, x0): x from channel cs0 of image img (the resource base is shifted by one row + one
      // pixel, so the patch origin (y0-1, x0-1) has this very offset), dy from channel m0
      gtab[k][g][0] = (int)(4u * (unsigned)((img * Cs + cs0) * HW + y0 * a.W + x0));
      gtab[k][g][1] = (int)(4u * (unsigned)((img * a.Cout + m0) * HW + y0 * a.W + x0));
      gtab[k][g][2] = ((y0 == 0) | ((y0 + RHP == a.H) << 1) | ((x0 == 0) << 2) | ((x0 + RWP == a.W) << 3)) << (4 * g);
      gtab[k][g][3] = 0;
    }
    if constexpr (!RAW) {
      if (tid < BCI) {
        const float* sc = src1 ? a.sc1 : a.sc0;
        const float* sh = src1 ? a.sh1 : a.sh0;
        aff[tid] = f32x2{sc ? sc[cs0 + tid] : 1.f, sc ? sh[cs0 + tid] : 0.f};
      }
    }
  }
  __syncthreads();

  // ---- input patch pieces (position-major).  Interior: element e = tid + 512 sl -> channel (tid >> 7) + 4 sl, pixel
  //      tid & 127 of the 128 interior pixels: ONE lane offset and ONE LDS word, the slot is a scalar / immediate step. ----
  unsigned i_off;
  int i_lds;
  const int wg_i = G == 2 ? (wave & 1) : 0;     // the region of this wave's interior pixels (pixel bit 6)
  {
    const int ci = tid >> 7, pos = tid & 127;
    const int g = pos / (RHP * RWP), iy = (pos % (RHP * RWP)) / RWP, ix = pos % RWP;
    // relative to the patch origin (y0-1, x0-1): pixel (iy+1, ix+1)
    i_off = 4u * (unsigned)(ci * HW + (iy + 1) * a.W + ix + 1);
    i_lds = ci * PS + g * GS + (iy + 1) * PW + ix + 1;
  }
  // Halo ring: element e = tid + 512 sl -> channel e / NH, ring position e % NH.
  unsigned h_off[NHS], h_cls[NHS];
  int h_lds[NHS], h_ci[NHS];
  bool h_g1[NHS];
#pragma unroll
  for (int sl = 0; sl < NHS; ++sl) {
    const int e = tid + NT * sl, ec = min(e, BCI * NH - 1);
    const int ci = ec / NH, h = ec % NH, g = h / NHG, hh = h % NHG;
    int r, col;
    if (hh < PCG) { r = 0; col = hh; }
    else if (hh < 2 * PCG) { r = PHG - 1; col = hh - PCG; }
    else { const int k = hh - 2 * PCG; r = 1 + k % (PHG - 2); col = (k / (PHG - 2)) ? PCG - 1 : 0; }
    h_off[sl] = 4u * (unsigned)(ci * HW + r * a.W + col);
    h_lds[sl] = e < BCI * NH ? ci * PS + g * GS + r * PW + col : -1;
    h_cls[sl] = e < BCI * NH ? (unsigned)((r == 0) | ((r == PHG - 1) << 1) | ((col == 0) << 2) | ((col == PCG - 1) << 3)) << (4 * g) : 0x100u;
    h_g1[sl] = g == 1;
    h_ci[sl] = ci;
  }
  // resource of the input shifted back by one row + one pixel: every offset above is >= 0; the bytes in front of the tensor
  // are only ever addressed by pieces that are outside their image, and those carry the offset 0xffffffff
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(xs - (a.W + 1)), 0, 0xfffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, 0xfffffff0, 0x00020000);

  float iraw[NIS], hraw[NHS];
  bool hinv[NHS];
  // the origin records of a K-tile, read ONCE per phase into scalar registers (x: both regions' origins and border bits)
  unsigned kx0 = 0, kx1 = 0, kborder = 0, kyo = 0;
  auto rec = [&](int k, int g, int f) __attribute__((always_inline)) { return (unsigned)__builtin_amdgcn_readfirstlane(gtab[k][g][f]); };
  auto x_records = [&](int k) __attribute__((always_inline)) {
    kx0 = rec(k, 0, 0);
    kx1 = G == 2 ? rec(k, 1, 0) : kx0;
    kborder = rec(k, 0, 2) | (G == 2 ? rec(k, 1, 2) : 0u) | 0x100u;
  };
  auto g_interior = [&](int sl) __attribute__((always_inline)) {
    const unsigned so = (wg_i ? kx1 : kx0) + (unsigned)sl * 16u * (unsigned)HW;                 // + 4 channels per slot
    iraw[sl] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, (int)i_off, (int)so, 0));
  };
  auto g_halo = [&](int sl) __attribute__((always_inline)) {
    const bool inv = (h_cls[sl] & kborder) != 0u;
    unsigned off = h_off[sl];
    unsigned so = kx0;
    if constexpr (G == 2) {                     // the ring positions of a wave belong to both regions: origin in the lane offset
      off += h_g1[sl] ? kx1 : kx0;
      so = 0u;
    }
    off = inv ? 0xffffffffu : off;
    hraw[sl] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, (int)off, (int)so, 0));
    hinv[sl] = inv;
  };
  auto s_interior = [&](int sl) __attribute__((always_inline)) {
    float v = iraw[sl];
    if constexpr (!RAW) {
      const f32x2 sa = aff[(tid >> 7) + 4 * sl];
      v = act_by_slope(fmaf(v, sa[0], sa[1]), slope);
    }
    Ps[i_lds + sl * 4 * PS] = v;
  };
  auto s_halo = [&](int sl) __attribute__((always_inline)) {
    float v = hraw[sl];
    if constexpr (!RAW) {
      const f32x2 sa = aff[h_ci[sl]];
      v = act_by_slope(fmaf(v, sa[0], sa[1]), slope);
      v = hinv[sl] ? 0.f : v;                   // act(affine(0)) != 0: padding stays zero
    }
    if (h_lds[sl] >= 0) Ps[h_lds[sl]] = v;
  };

  // ---- dY item: output channel (tid >> 2) & 63, tile (tid & 3) + 4 * (tid >> 8) of the K-tile (G = 2: the region of a
  //      wave's tiles is uniform: waves 0-3 region 0, waves 4-7 region 1) ----
  const int y_co = (tid >> 2) & 63, y_tile = (tid & 3) + 4 * ((tid >> 8) & 1);
  const int wg_y = G == 2 ? (wave >> 2) : 0;
  unsigned y_off;
  {
    const int tg = y_tile % (RH * RW), ty = tg / RW, tx = tg % RW;
    y_off = 4u * (unsigned)(y_co * HW + 4 * ty * a.W + 4 * tx);
  }
  const bool y_ok = m0 + y_co < a.Cout;
  const int y_dst = y_co * X4_YS + y_tile;      // Yh[xi][co][tile]: + xi * 64 * 9
  f32x4 dyr[4];
  auto y_issue = [&](int k) __attribute__((always_inline)) {
    const unsigned so = kyo = rec(k, wg_y, 1);
    const unsigned off = y_ok ? y_off : 0xffffffffu;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      dyr[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dy, (int)off, (int)(so + (unsigned)(r * 4) * (unsigned)a.W), 0));
  };
  // the six outputs of A y (y = four values): y0, e+o, e-o, p+2r, p-2r, y3 — 8 instructions
#define X4_A6(Y0, Y1, Y2, Y3, O0, O1, O2, O3, O4, O5)                        \
  do {                                                                       \
    const float e__ = Y0 + Y2, o__ = Y1 + Y3;                                \
    const float p__ = fmaf(4.f, Y2, Y0), r__ = fmaf(4.f, Y3, Y1);            \
    O0 = Y0;                                                                 \
    O1 = e__ + o__;                                                          \
    O2 = e__ - o__;                                                          \
    O3 = fmaf(2.f, r__, p__);                                                \
    O4 = fmaf(-2.f, r__, p__);                                               \
    O5 = Y3;                                                                 \
  } while (0)
  auto y_transform = [&]() __attribute__((always_inline)) {
    float t[6][4];
#pragma unroll
    for (int b = 0; b < 4; ++b)
      X4_A6(dyr[0][b], dyr[1][b], dyr[2][b], dyr[3][b], t[0][b], t[1][b], t[2][b], t[3][b], t[4][b], t[5][b]);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      float o0, o1, o2, o3, o4, o5;
      X4_A6(t[i][0], t[i][1], t[i][2], t[i][3], o0, o1, o2, o3, o4, o5);
      float* yp = Ys + y_dst + (i * 6) * (BCO * X4_YS);
      yp[0 * BCO * X4_YS] = o0;
      yp[1 * BCO * X4_YS] = o1;
      yp[2 * BCO * X4_YS] = o2;
      yp[3 * BCO * X4_YS] = o3;
      yp[4 * BCO * X4_YS] = o4;
      yp[5 * BCO * X4_YS] = o5;
    }
  };
#undef X4_A6

  // ---- V item: input channel li, tile 2 * (wave >> 1) + lk, half = wave & 1 (rows 0-2 / 3-5 of B^T d B) ----
  const int thalf = wave & 1;
  int v_src, v_dst;
  {
    const int tile = 2 * (wave >> 1) + lk, g = tile / (RH * RW), tg = tile % (RH * RW), ty = tg / RW, tx = tg % RW;
    v_src = li * PS + g * GS + (4 * ty + thalf) * PW + 4 * tx;
    v_dst = (18 * thalf * 8 + tile) * BCI + li;               // V[18 half + m][tile][ci]: + m * 256
  }
  auto v_transform = [&](auto half_) __attribute__((always_inline)) {
    constexpr int HALF = decltype(half_)::value;
    f32x4 ra[5];
    f32x2 rb[5];
#pragma unroll
    for (int r = 0; r < 5; ++r) {
      ra[r] = *reinterpret_cast<const f32x4*>(Ps + v_src + r * PW);
      rb[r] = *reinterpret_cast<const f32x2*>(Ps + v_src + r * PW + 4);
    }
    float tt[3][6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      float R[5];
#pragma unroll
      for (int r = 0; r < 5; ++r) R[r] = c < 4 ? ra[r][c & 3] : rb[r][c & 1];
      if constexpr (HALF == 0) {
        const float aa = fmaf(-4.f, R[2], R[4]), bb = fmaf(-4.f, R[1], R[3]);
        tt[0][c] = fmaf(4.f, R[0], fmaf(-5.f, R[2], R[4]));
        tt[1][c] = aa + bb;
        tt[2][c] = aa - bb;
      } else {
        const float cc_ = R[3] - R[1], ee = R[2] - R[0];
        tt[0][c] = fmaf(2.f, ee, cc_);
        tt[1][c] = fmaf(-2.f, ee, cc_);
        tt[2][c] = fmaf(4.f, R[0], fmaf(-5.f, R[2], R[4]));
      }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float X0 = tt[i][0], X1 = tt[i][1], X2 = tt[i][2], X3 = tt[i][3], X4 = tt[i][4], X5 = tt[i][5];
      const float a_ = fmaf(-4.f, X2, X4), b_ = fmaf(-4.f, X1, X3), c_ = X4 - X2, e_ = X3 - X1;
      float* vp = Vs + v_dst + i * 6 * (8 * BCI);
      vp[0 * 8 * BCI] = fmaf(4.f, X0, fmaf(-5.f, X2, X4));
      vp[1 * 8 * BCI] = a_ + b_;
      vp[2 * 8 * BCI] = a_ - b_;
      vp[3 * 8 * BCI] = fmaf(2.f, e_, c_);
      vp[4 * 8 * BCI] = fmaf(-2.f, e_, c_);
      vp[5 * 8 * BCI] = fmaf(4.f, X1, fmaf(-5.f, X3, X5));
    }
  };

  // ---- MFMA role: wave (cb = wave & 1, q = wave >> 1): positions 9q .. 9q+8 of output-channel block cb ----
  const int wcb = wave & 1, wq = wave >> 1;
  const int a_lane = (9 * wq * BCO + wcb * 32 + li) * X4_YS + lk;         // Yh[9q + e][cb*32 + li][2s + lk]
  const int b_lane = (9 * wq * 8 + lk) * BCI + li;                        // V[9q + e][2s + lk][li]
  f32x16 acc[9];
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 9; ++e) acc[e] = zero16;

  auto lds_barrier = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // LDS only: the global loads in flight are register loads
  };

  // ---- prologue: patch of K-tile 0 in LDS, patch of K-tile 1 and dY of K-tile 0 in flight ----
  x_records(0);
#pragma unroll
  for (int sl = 0; sl < NIS; ++sl) g_interior(sl);
#pragma unroll
  for (int sl = 0; sl < NHS; ++sl) g_halo(sl);
  y_issue(0);
#pragma unroll
  for (int sl = 0; sl < NIS; ++sl) s_interior(sl);
#pragma unroll
  for (int sl = 0; sl < NHS; ++sl) s_halo(sl);
  {
    x_records(min(1, nk - 1));
#pragma unroll
    for (int sl = 0; sl < NIS; ++sl) g_interior(sl);
#pragma unroll
    for (int sl = 0; sl < NHS; ++sl) g_halo(sl);
  }
  lds_barrier();

  constexpr int RBK = 4;                        // operand reads run this many MFMAs ahead
  for (int k = 0; k < nk; ++k) {
    // -------- T phase --------
    y_transform();
    if (thalf) v_transform(std::integral_constant<int, 1>{});
    else v_transform(std::integral_constant<int, 0>{});
    lds_barrier();
    // -------- M phase: 36 MFMAs; behind them S(k+1), G(k+2) and the dY rows of k+1 --------
    const int k1 = min(k + 1, nk - 1), k2 = min(k + 2, nk - 1);
    x_records(k2);
    y_issue(k1);
    float av[36], bv[36];
#pragma unroll
    for (int m = 0; m < RBK; ++m) {
      av[m] = Ys[a_lane + (m % 9) * (BCO * X4_YS) + 2 * (m / 9)];
      bv[m] = Vs[b_lane + ((m % 9) * 8 + 2 * (m / 9)) * BCI];
    }
#pragma unroll
    for (int m = 0; m < 36; ++m) {
      const int e = m % 9;
      acc[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[m], acc[e], 0, 0, 0);
      if (m + RBK < 36) {
        av[m + RBK] = Ys[a_lane + ((m + RBK) % 9) * (BCO * X4_YS) + 2 * ((m + RBK) / 9)];
        bv[m + RBK] = Vs[b_lane + (((m + RBK) % 9) * 8 + 2 * ((m + RBK) / 9)) * BCI];
      }
      if (m < NIS) {
        s_interior(m);
        g_interior(m);
      } else if (m < NIS + NHS) {
        s_halo(m - NIS);
        g_halo(m - NIS);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    lds_barrier();
  }
  __syncthreads();

  // ---- epilogue: dW = G^T M G, four passes (channel block cbp, accumulator rows 8h .. 8h+7 = 16 output channels) ----
  float* const Zs = Ys;                         // [xi 36][co 16][ci 32]
  const int ec = tid >> 5, eci = tid & 31;
#define X4_GT3(M0, M1, M2, M3, M4, M5, O0, O1, O2)                                   \
  do {                                                                               \
    const float s12__ = M1 + M2, s34__ = M3 + M4;                                    \
    O0 = fmaf(0.25f, M0, fmaf(-1.f / 6.f, s12__, (1.f / 24.f) * s34__));             \
    O1 = fmaf(1.f / 6.f, M2 - M1, (1.f / 12.f) * (M3 - M4));                         \
    O2 = fmaf(1.f / 6.f, s34__ - s12__, M5);                                         \
  } while (0)
#pragma unroll
  for (int cbp = 0; cbp < 2; ++cbp) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (wcb == cbp) {
#pragma unroll
        for (int e = 0; e < 9; ++e)
#pragma unroll
          for (int rr = 0; rr < 8; ++rr)
            Zs[((9 * wq + e) * 16 + (rr & 3) + 8 * (rr >> 2) + 4 * lk) * 32 + li] = acc[e][8 * h + rr];
      }
      __syncthreads();
      {
        const int co = m0 + cbp * 32 + 16 * h + ec, ci = c0 + eci;
        const float* z = Zs + ec * 32 + eci;
        float r_[3][6];                          // G^T M: rows
#pragma unroll
        for (int j = 0; j < 6; ++j)
          X4_GT3(z[(0 * 6 + j) * 512], z[(1 * 6 + j) * 512], z[(2 * 6 + j) * 512], z[(3 * 6 + j) * 512], z[(4 * 6 + j) * 512],
                 z[(5 * 6 + j) * 512], r_[0][j], r_[1][j], r_[2][j]);
        if (co < a.Cout && ci < a.Cin) {
          float* o = a.out + ((long long)split * 9 * a.Cout + co) * a.Cin + ci;
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            float w0, w1, w2;
            X4_GT3(r_[i][0], r_[i][1], r_[i][2], r_[i][3], r_[i][4], r_[i][5], w0, w1, w2);
            o[(long long)(3 * i + 0) * a.Cout * a.Cin] = w0;
            o[(long long)(3 * i + 1) * a.Cout * a.Cin] = w1;
            o[(long long)(3 * i + 2) * a.Cout * a.Cin] = w2;
          }
        }
      }
      __syncthreads();
    }
  }
#undef X4_GT3
}

// ---------------------------------------------------------------------------
// host side (called from conv.hip)
// ---------------------------------------------------------------------------
int w3_reduce(const float* ws, float* dw, long long P, int splits, hipStream_t st);      // conv3x3.hip

struct X4Plan { int cfg, ryn, rxn, nkt, kps, splits, gridM, gridC; };
// cfg 0: one 8x16-pixel region per K-tile (maps that tile by 8 x 16); cfg 1: two 8x8 regions (maps that tile by 8 x 8)
static bool x4_plan(const avsep_conv_desc* d, X4Plan* out) {
  X4Plan p{};
  if (!(d->H & 7) && !(d->W & 15)) p.cfg = 0;
  else if (!(d->H & 7) && !(d->W & 7)) p.cfg = 1;
  else return false;
  const int rhp = 8, rwp = p.cfg == 0 ? 16 : 8, g = p.cfg == 0 ? 1 : 2;
  p.ryn = d->H / rhp;
  p.rxn = d->W / rwp;
  const long long nreg = (long long)d->N * p.ryn * p.rxn;
  if (nreg % g) return false;
  p.nkt = (int)(nreg / g);
  p.gridM = cdiv(d->Cout, X4_BCO);
  p.gridC = cdiv(d->Cin, X4_BCI);
  // one workgroup per CU (512 threads x 256 registers, ~150 KB of LDS): ONE round of workgroups, as in wgrad_wino.hip
  int want = cu_count() / (p.gridM * p.gridC);
  if (want < 1) want = 1;
  p.kps = cdiv(p.nkt, want);
  if (p.kps > X4_KPS_MAX) p.kps = X4_KPS_MAX;
  if (p.kps < 1) p.kps = 1;
  p.splits = cdiv(p.nkt, p.kps);
  *out = p;
  return true;
}

bool x4_applicable(const avsep_conv_desc* d) {
  if ((d->algo & (AVSEP_ALGO_NO_WINOGRAD | AVSEP_ALGO_NO_WINOGRAD_WGRAD | AVSEP_ALGO_NO_WINOGRAD4)) || d->prec != AVSEP_PREC_F32) return false;
  if (!(d->KH == 3 && d->KW == 3 && d->stride == 1 && d->dil == 1 && d->pad == 1) || d->up2x) return false;
  const int C1 = d->Cin - d->C0;
  if (d->Cin % X4_BCI || d->C0 % X4_BCI || d->Cout < 48 || (C1 != 0 && C1 != d->C0)) return false;
  if ((long long)d->N * (d->C0 > d->Cout ? d->C0 : d->Cout) * d->H * d->W >= 0x3fffffffLL) return false;   // 32-bit BYTE offsets
  const avsep_conv_desc e = plan_desc(d);
  X4Plan p;
  if (!x4_plan(&e, &p)) return false;
  X4Plan q;
  if (!x4_plan(d, &q)) return false;
  return (long long)p.gridM * p.gridC * p.splits >= 128 && p.nkt >= 8;
}
size_t x4_workspace_floats(const avsep_conv_desc* d) {
  X4Plan p;
  if (!x4_plan(d, &p)) return 0;
  return (size_t)p.splits * 9 * d->Cout * d->Cin;
}
void x4_variant(const avsep_conv_desc* d, char* buf, size_t cap) {
  X4Plan p;
  const avsep_conv_desc e = plan_desc(d);
  if (!x4_plan(&e, &p)) { snprintf(buf, cap, "?"); return; }
  snprintf(buf, cap, "%s,split%d", p.cfg == 0 ? "1x8x16" : "2x8x8", p.splits);
}

int x4_wgrad(const avsep_conv_desc* d, const float* dy, float* dw, float* ws, hipStream_t st) {
  X4Plan p;
  if (!x4_plan(d, &p)) return AVSEP_ERR_ARG;
  X4Args a{};
  a.N = d->N; a.C0 = d->C0; a.C1 = d->Cin - d->C0; a.Cin = d->Cin; a.H = d->H; a.W = d->W; a.Cout = d->Cout;
  a.ryn = p.ryn; a.rxn = p.rxn; a.nkt = p.nkt; a.kps = p.kps; a.gridM = p.gridM; a.gridC = p.gridC;
  a.act0 = d->act0; a.act1 = d->act1;
  a.x0 = d->x0; a.x1 = d->x1; a.sc0 = d->scale0; a.sh0 = d->shift0; a.sc1 = d->scale1; a.sh1 = d->shift1;
  a.dy = dy; a.out = ws;
  const bool raw = !d->scale0 && !d->scale1 && d->act0 == AVSEP_ACT_NONE && (a.C1 == 0 || d->act1 == AVSEP_ACT_NONE);
  dim3 grid((unsigned)(p.gridM * p.gridC * p.splits));
  // <G, RH, RW, PW, GS, PS>
  if (p.cfg == 0) {
    if (raw) hipLaunchKernelGGL((winow4_kernel<1, 2, 4, 20, 200, 204, true>), grid, dim3(X4_THREADS), 0, st, a);
    else hipLaunchKernelGGL((winow4_kernel<1, 2, 4, 20, 200, 204, false>), grid, dim3(X4_THREADS), 0, st, a);
  } else {
    if (raw) hipLaunchKernelGGL((winow4_kernel<2, 2, 2, 12, 120, 244, true>), grid, dim3(X4_THREADS), 0, st, a);
    else hipLaunchKernelGGL((winow4_kernel<2, 2, 2, 12, 120, 244, false>), grid, dim3(X4_THREADS), 0, st, a);
  }
  AVSEP_LAUNCH_CHECK();
  return w3_reduce(ws, dw, (long long)d->Cout * d->Cin, p.splits, st);
}
