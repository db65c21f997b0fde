// Weight gradient of the 3x3 / stride 1 / pad 1 convolutions in the Winograd F(4x4, 3x3) form, fp32:
//     dW = sum over 4x4 tiles of  G^T [ (A dY A^T) (.) (B^T d B) ] G          (the transpose of conv_wino4.hip's forward)
// dY = the 4x4 cotangent tile, d = the 6x6 input tile around it: 36 products per tile and (co, ci) where the direct form
// has 144 and the F(2x2) form of wgrad_wino.hip 64.  Per transform position xi this is a [Cout x tiles] x [tiles x Cin]
// GEMM whose K dimension is the TILE index: one MFMA k-step consumes two tiles, and both operands are transformed in the
// kernel.  A 512-thread workgroup owns 64 output x 32 input channels (36 x 2 accumulator tiles, 9 per wave = 144
// registers: wave (cb, q) has positions 9q .. 9q+8 of output-channel block cb) and sweeps the 8x8-pixel regions (2x2
// tiles = two MFMA k-steps) of its K-split, ONE region per step and one LDS-only barrier per step:
//   M  18 MFMAs per wave on the transformed operands of region j, both fragments conflict-free ds_read_b32;
//   T  behind them, the transforms of region j+1 into the other copy of the operands: waves 0-3 take one (output channel,
//      tile) of dY each — 16 values straight from global memory (four 16-byte row loads issued a step earlier), A y A^T in
//      80 vector instructions, 36 stores into Yh4[xi][tile][co] — waves 4-7 one (input channel, tile, half) of the input
//      patch — B^T d B as in conv_wino4.hip, 10 row reads, 72 instructions, 18 stores into V4[xi][tile][ci];
//   S  the raw input patch of region j+2 registers -> LDS (folded BatchNorm affine + activation + two-source concat);
//   G  the patch of region j+3 and (waves 0-3) the dY rows of region j+2 global -> registers.
// Everything in LDS is double buffered (Yh4 41.5 KB + V4 18.4 KB + patch 15.9 KB, twice = 151.5 KB).  The patch is staged
// position-major: its 64 interior pixels per channel are 4 pieces per thread that can never leave the image (8x8 regions
// divide the map), the halo ring is 3 pieces whose validity (region on an image border) is one AND + compare + select per
// region; the regions' origin records (offset | border bits) sit in LDS and are read into scalar registers once per step.
// (First form of this kernel, measured and replaced: 8-tile K-tiles in two phases, transform | MFMA, with ONE copy of the
// operands — its transform phase is bound by the LDS store path, 110 KB per K-tile at 64-85 B/clk: 0.51 of the matrix pipe
// busy; the pipelined form is 3-15 % faster per call)
// Epilogue: M goes through LDS in four passes (as conv_wino4.hip), every thread applies G^T M G to one (co, ci) and
// writes its 9 taps into a tap-major partial slab [split][tap][Cout][Cin]; conv3x3.hip's w3_reduce sums the slabs.
// The grid is 1-D and XCD-aware: the gridM * gridC workgroups of one K-split read the same pixels and share an L2.
#include <stdlib.h>
#include <type_traits>

#include "common.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int X4_BCO = 64, X4_BCI = 32;         // channels per workgroup
constexpr int X4_THREADS = 512;

struct X4Args {
  int N, C0, C1, Cin, H, W, Cout;
  int ryn, rxn;                                 // 8x8 regions per image (per parity sub-map) along y / x
  int dil, Hs, Ws;                              // GEN: dilation (1 | 2) and the size of a parity sub-map (H / dil, W / dil)
  int nkt, kps;                                 // regions in total / per split
  int gridM, gridC, act0, act1;
  const float *x0, *x1, *sc0, *sh0, *sc1, *sh1;
  const float* dy;
  float* out;                                   // slabs [split][tap][Cout][Cin]
};

constexpr int X4P_YCO = 72;                     // Yh4[xi][tile][co]: 64 channels padded to 72 words (the (tile, co) lanes of a store hit 32 banks)
constexpr int X4P_Y_FLOATS = 36 * 4 * X4P_YCO;  // 10368
constexpr int X4P_V_FLOATS = 36 * 4 * X4_BCI;   // 4608
constexpr int X4P_PW = 12, X4P_PS = 124;        // patch 10 x 10 in rows of 12; channel stride 124 (= -4 mod 64: b128 row reads of 16 channels cover 64 banks)
constexpr int X4P_P_FLOATS = X4_BCI * X4P_PS;   // 3968
constexpr int X4P_KPS_MAX = 512;                // regions of one split (origin records in LDS)

// GEN: maps the 8x8 regions do not divide (28x28, 14x14) and dilation 2 (= four parity sub-maps of H/2 x W/2, each a
// dilation-1 problem on pixels 2 apart).  A region's record carries a third word, the VALID rows | columns of its 10x10
// patch as bit masks; every patch piece and every dY element tests its own (row bit | column bit) against it and loads
// through the out-of-range offset (= 0) when it is outside the map.  GEN 1: dilation 1 and W even — dY in 8-byte pairs
// (a pair is inside or outside together); GEN 2: odd W, dY element by element; GEN 3: dilation 2 and W even, the
// two x parities of a sub-map row TOGETHER in one region: its four tiles are (tile row ty, x parity px), the patch is 10
// sub-rows x 12 CONSECUTIVE columns of the map (the two parities' 6-column patches interleaved, de-interleaved by the LDS
// store addresses) and a lane pair (px = 0 | 1) loads the 8 consecutive dY values of a tile row as two 8-byte pairs each
// and swaps the odd / even ones through the DPP crossbar — half the cache-line lookups of GEN 2, which bound it, and patch
// rows that are contiguous in memory.
template <bool RAW, int GEN>
__global__ __launch_bounds__(X4_THREADS) void winow4_kernel(X4Args a) {
  constexpr int NT = X4_THREADS, BCO = X4_BCO, BCI = X4_BCI, PW = X4P_PW, PS = X4P_PS, YCO = X4P_YCO;
  // interior pieces per thread: 4; halo ring: 36 positions per channel, 3 pieces per thread.  GEN 3: the 120 positions of
  // 4 channels on threads 0-479, 8 pieces per thread (channel 4 sl + tid / 120), all of them masked
  constexpr int NIS = GEN == 3 ? 8 : BCI * 64 / NT;
  constexpr int NH = 36, NHS = GEN == 3 ? 0 : (BCI * NH + NT - 1) / NT, NHA = NHS ? NHS : 1;
  constexpr int ISL = GEN == 3 ? 4 : 8;         // channels between the interior pieces of a thread
  constexpr int BUF = X4P_Y_FLOATS + X4P_V_FLOATS + X4P_P_FLOATS;
  static_assert(2 * BUF >= 36 * 16 * 32, "epilogue exchange fits");
  __shared__ __attribute__((aligned(16))) float smem[2 * BUF];
  __shared__ unsigned gtab[X4P_KPS_MAX][GEN ? 3 : 2];   // per region: {byte offset of its origin in x | border bits, byte offset in dy, GEN: valid rows | columns << 10}
  __shared__ f32x2 aff[BCI];
  auto Yb = [&](int b) __attribute__((always_inline)) { return smem + b * BUF; };
  auto Vb = [&](int b) __attribute__((always_inline)) { return smem + b * BUF + X4P_Y_FLOATS; };
  auto Pb = [&](int b) __attribute__((always_inline)) { return smem + b * BUF + X4P_Y_FLOATS + X4P_V_FLOATS; };

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 31, lk = lane >> 5;
  const int t_ = xcd_remap(blockIdx.x, gridDim.x), per_split = a.gridM * a.gridC;
  const int split = t_ / per_split, lin_ = t_ % per_split;
  const int mt = lin_ % a.gridM, ct = lin_ / a.gridM;
  const int m0 = mt * BCO, c0 = ct * BCI;
  const int HW = a.H * a.W;
  const bool src1 = c0 >= a.C0;
  const float* const xs = src1 ? a.x1 : a.x0;
  const int Cs = src1 ? a.C1 : a.C0, cs0 = src1 ? c0 - a.C0 : c0;
  const float slope = act_slope(src1 ? a.act1 : a.act0);
  const int kt0 = split * a.kps, nk = min(a.kps, a.nkt - kt0);

  const int D = GEN ? a.dil : 1;
  {
    const int per = a.ryn * a.rxn;
    if constexpr (GEN == 3) {
      const int pper = 2 * per;
      for (int k = tid; k < nk; k += NT) {
        const int reg = kt0 + k, img = reg / pper, rem = reg % pper, py = rem / per;
        const int ry = (rem % per) / a.rxn, rx = rem % a.rxn;
        const int y0 = ry * 8, x0 = rx * 8, pix = (2 * y0 + py) * a.W + x0;
        const int vr9 = min(9, a.Hs - y0), qmax = min(11, a.W - x0 + 1);   // last valid patch row / column (column q = map column x0 - 2 + q)
        const unsigned rm = ((2u << vr9) - 1u) & ~(y0 == 0 ? 1u : 0u), cm = ((2u << qmax) - 1u) & ~(x0 == 0 ? 3u : 0u);
        gtab[k][0] = 4u * (unsigned)((img * Cs + cs0) * HW + pix);
        gtab[k][1] = 4u * (unsigned)((img * a.Cout + m0) * HW + pix);
        gtab[k][2] = rm | (cm << 10);
      }
    } else if constexpr (GEN) {
      const int pper = D * D * per;
      for (int k = tid; k < nk; k += NT) {
        const int reg = kt0 + k, img = reg / pper, rem = reg % pper, par = rem / per;
        const int ry = (rem % per) / a.rxn, rx = rem % a.rxn, py = par / D, px = par % D;
        const int y0 = ry * 8, x0 = rx * 8, pix = (D * y0 + py) * a.W + D * x0 + px;
        const int vr9 = min(9, a.Hs - y0), vc9 = min(9, a.Ws - x0);       // last valid patch row / column
        const unsigned rm = ((2u << vr9) - 1u) & ~(y0 == 0 ? 1u : 0u), cm = ((2u << vc9) - 1u) & ~(x0 == 0 ? 1u : 0u);
        gtab[k][0] = 4u * (unsigned)((img * Cs + cs0) * HW + pix);
        gtab[k][1] = 4u * (unsigned)((img * a.Cout + m0) * HW + pix);
        gtab[k][2] = rm | (cm << 10);
      }
    } else
    for (int k = tid; k < nk; k += NT) {
      const int reg = kt0 + k, img = reg / per, ry = (reg % per) / a.rxn, rx = reg % a.rxn;
      const int y0 = ry * 8, x0 = rx * 8;
      // (offsets are multiples of 32 bytes: the border bits ride in the low four)
      gtab[k][0] = (4u * (unsigned)((img * Cs + cs0) * HW + y0 * a.W + x0)) |
                   (unsigned)((y0 == 0) | ((y0 + 8 == a.H) << 1) | ((x0 == 0) << 2) | ((x0 + 8 == a.W) << 3));
      gtab[k][1] = 4u * (unsigned)((img * a.Cout + m0) * HW + y0 * a.W + x0);
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

  // ---- patch pieces: interior element e = tid + 512 sl -> channel (tid >> 6) + 8 sl, pixel tid & 63 (one lane offset, one
  //      LDS word); halo ring element e -> channel e / 36, ring position e % 36 ----
  unsigned i_off, i_pm;                         // (GEN) i_pm / h_cls: the piece's row bit | column bit << 10
  int i_lds, i_ci = tid >> 6;
  if constexpr (GEN == 3) {
    const int ci = tid / 120, rq = tid % 120, r = rq / 12, q = rq % 12, px = q & 1, sc = q >> 1;
    const bool on = tid < 480;
    i_ci = on ? ci : 3;
    i_off = on ? 4u * (unsigned)(ci * HW + 2 * r * a.W + q) : 0xffffffffu;
    i_pm = on ? (1u << r) | (1u << (10 + q)) : 1u << 22;
    // parity 0: sub-columns 0-5 at words 0-5 of the row; parity 1: 0-3 at words 8-11, 4-5 at words 6-7 (16-byte row reads);
    // the idle threads write the 4 pad words behind a channel's patch
    i_lds = on ? ci * PS + r * PW + (px ? (sc < 4 ? 8 + sc : 2 + sc) : sc) : (tid & 3) * PS + 120 + ((tid >> 2) & 3);
  } else {
    const int ci = tid >> 6, pos = tid & 63, iy = pos >> 3, ix = pos & 7;
    i_off = 4u * (unsigned)(ci * HW + D * ((iy + 1) * a.W + ix + 1));
    i_lds = ci * PS + (iy + 1) * PW + ix + 1;
    i_pm = (1u << (iy + 1)) | (1u << (10 + ix + 1));
  }
  unsigned h_off[NHA], h_cls[NHA];
  int h_lds[NHA], h_ci[NHA];
#pragma unroll
  for (int sl = 0; sl < NHS; ++sl) {
    const int e = tid + NT * sl, ec = min(e, BCI * NH - 1);
    const int ci = ec / NH, hh = ec % NH;
    int r, col;
    if (hh < 10) { r = 0; col = hh; }
    else if (hh < 20) { r = 9; col = hh - 10; }
    else { const int k = hh - 20; r = 1 + (k & 7); col = (k >> 3) ? 9 : 0; }
    h_off[sl] = 4u * (unsigned)(ci * HW + D * (r * a.W + col));
    h_lds[sl] = e < BCI * NH ? ci * PS + r * PW + col : -1;
    if constexpr (GEN) h_cls[sl] = e < BCI * NH ? (1u << r) | (1u << (10 + col)) : 1u << 20;
    else h_cls[sl] = e < BCI * NH ? (unsigned)((r == 0) | ((r == 9) << 1) | ((col == 0) << 2) | ((col == 9) << 3)) : 0x10u;
    h_ci[sl] = ci;
  }
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(xs - (GEN == 3 ? 2 * a.W + 2 : D * (a.W + 1))), 0, 0xfffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, 0xfffffff0, 0x00020000);

  float iraw[NIS], hraw[NHA];
  bool hinv[NHA], iinv[NIS];
  auto g_interior = [&](unsigned xrec, unsigned srec, int sl) __attribute__((always_inline)) {
    if constexpr (GEN) {
      const bool inv = (i_pm & srec) != i_pm;
      const unsigned off = inv ? 0xffffffffu : i_off;
      iraw[sl] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, (int)off, (int)(xrec + (unsigned)sl * (4u * ISL) * (unsigned)HW), 0));
      iinv[sl] = inv;
    } else {
      const unsigned so = (xrec & ~31u) + (unsigned)sl * 32u * (unsigned)HW;                    // + 8 channels per slot
      iraw[sl] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, (int)i_off, (int)so, 0));
    }
  };
  auto g_halo = [&](unsigned xrec, unsigned srec, int sl) __attribute__((always_inline)) {
    const bool inv = GEN ? (h_cls[sl] & srec) != h_cls[sl] : (h_cls[sl] & ((xrec & 15u) | 0x10u)) != 0u;
    const unsigned off = inv ? 0xffffffffu : h_off[sl];
    hraw[sl] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, (int)off, (int)(GEN ? xrec : xrec & ~31u), 0));
    hinv[sl] = inv;
  };
  auto s_interior = [&](int b, int sl) __attribute__((always_inline)) {
    float v = iraw[sl];
    if constexpr (!RAW) {
      const f32x2 sa = aff[i_ci + ISL * sl];
      v = act_by_slope(fmaf(v, sa[0], sa[1]), slope);
      if constexpr (GEN) v = iinv[sl] ? 0.f : v;
    }
    Pb(b)[i_lds + sl * ISL * PS] = v;
  };
  auto s_halo = [&](int b, int sl) __attribute__((always_inline)) {
    float v = hraw[sl];
    if constexpr (!RAW) {
      const f32x2 sa = aff[h_ci[sl]];
      v = act_by_slope(fmaf(v, sa[0], sa[1]), slope);
      v = hinv[sl] ? 0.f : v;
    }
    if (h_lds[sl] >= 0) Pb(b)[h_lds[sl]] = v;
  };
  auto xrec_of = [&](int k) __attribute__((always_inline)) { return (unsigned)__builtin_amdgcn_readfirstlane((int)gtab[k][0]); };
  auto yrec_of = [&](int k) __attribute__((always_inline)) { return (unsigned)__builtin_amdgcn_readfirstlane((int)gtab[k][1]); };
  auto srec_of = [&](int k) __attribute__((always_inline)) { return GEN ? (unsigned)__builtin_amdgcn_readfirstlane((int)gtab[k][GEN ? 2 : 0]) : 0u; };

  // ---- dY role (waves 0-3): output channel tid >> 2, tile tid & 3 = (ty, tx) ----
  const bool yrole = wave < 4;
  const int y_co = (tid >> 2) & 63, y_tile = tid & 3;
  const unsigned y_off = (m0 + y_co < a.Cout) ? 4u * (unsigned)(y_co * HW + (GEN == 3 ? 8 * (y_tile >> 1) * a.W + 4 * (y_tile & 1)
                                                                                      : D * (4 * (y_tile >> 1) * a.W + 4 * (y_tile & 1)))) : 0xffffffffu;
  // (GEN) the tile's rows / columns in the region's masks (GEN 3: the lane's 4 consecutive map columns, patch columns 2 + 4 px ...)
  const int y_rsh = 4 * (y_tile >> 1) + 1, y_csh = 10 + 4 * (y_tile & 1) + (GEN == 3 ? 2 : 1);
  const bool y_px = y_tile & 1;
  const int y_dst = y_tile * YCO + y_co;        // Yh4[xi][tile][co]: + xi * 4 * 72
  f32x4 dyr[4];
  auto y_issue = [&](unsigned yrec, unsigned srec) __attribute__((always_inline)) {
    if constexpr (GEN == 1 || GEN == 3) {
      // pair (r, p) = columns 2p, 2p + 1 of row r: valid when row bit r and column bit 2p of the tile are
      const unsigned rsel = (srec >> y_rsh) & 15u, csel = (srec >> y_csh) & 15u, psel = (csel & 1u) | ((csel >> 1) & 2u);
      unsigned m8 = 0u;
#pragma unroll
      for (int r = 0; r < 4; ++r) m8 |= (0u - ((rsel >> r) & 1u)) & (psel << (2 * r));
      const unsigned nm = ~m8;
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
          const unsigned t = (unsigned)((int)(nm << (31 - (2 * r + pp))) >> 31);               // all ones when invalid
          const f32x2 v2 = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_dy, (int)(y_off | t),
                                                                                          (int)(yrec + 4u * (unsigned)((GEN == 3 ? 2 : 1) * r * a.W + 2 * pp)), 0));
          dyr[r][2 * pp] = v2[0];
          dyr[r][2 * pp + 1] = v2[1];
        }
    } else if constexpr (GEN == 2) {
      // element (r, c) is valid when row bit r and column bit c of the tile are: nm = ~(16 bits, row-major)
      const unsigned rsel = (srec >> y_rsh) & 15u, csel = (srec >> y_csh) & 15u;
      unsigned m16 = 0u;
#pragma unroll
      for (int r = 0; r < 4; ++r) m16 |= (0u - ((rsel >> r) & 1u)) & (csel << (4 * r));
      const unsigned nm = ~m16;
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const unsigned t = (unsigned)((int)(nm << (31 - (4 * r + c))) >> 31);            // all ones when invalid
          dyr[r][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_dy, (int)(y_off | t),
                                                                                     (int)(yrec + 4u * (unsigned)D * (unsigned)(r * a.W + c)), 0));
        }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        dyr[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dy, (int)y_off, (int)(yrec + (unsigned)(r * 4) * (unsigned)a.W), 0));
    }
  };
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
  float yt[6][4];
  // GEN 3: the lane pair holds map columns 0-3 | 4-7 of the tile row; parity 0 wants columns 0 2 4 6, parity 1 columns 1 3 5 7
  auto y_swap = [&]() __attribute__((always_inline)) {
    if constexpr (GEN == 3) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float k0 = y_px ? dyr[r][1] : dyr[r][0], k1 = y_px ? dyr[r][3] : dyr[r][2];
        const float g0 = y_px ? dyr[r][0] : dyr[r][1], g1 = y_px ? dyr[r][2] : dyr[r][3];
        const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, g0), 0xB1, 0xf, 0xf, false));
        const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, g1), 0xB1, 0xf, 0xf, false));
        dyr[r] = f32x4{y_px ? r0 : k0, y_px ? r1 : k1, y_px ? k0 : r0, y_px ? k1 : r1};
      }
    }
  };
  auto y_stage1 = [&](int b_) __attribute__((always_inline)) {              // column b_ of dY: 8 instructions
    X4_A6(dyr[0][b_], dyr[1][b_], dyr[2][b_], dyr[3][b_], yt[0][b_], yt[1][b_], yt[2][b_], yt[3][b_], yt[4][b_], yt[5][b_]);
  };
  auto y_stage2 = [&](int b, int i) __attribute__((always_inline)) {         // row i: 8 instructions + 6 stores
    float o0, o1, o2, o3, o4, o5;
    X4_A6(yt[i][0], yt[i][1], yt[i][2], yt[i][3], o0, o1, o2, o3, o4, o5);
    float* yp = Yb(b) + y_dst + (i * 6) * (4 * YCO);
    yp[0 * 4 * YCO] = o0;
    yp[1 * 4 * YCO] = o1;
    yp[2 * 4 * YCO] = o2;
    yp[3 * 4 * YCO] = o3;
    yp[4 * 4 * YCO] = o4;
    yp[5 * 4 * YCO] = o5;
  };
#undef X4_A6

  // ---- input role (waves 4-7): input channel li, tile 2 * ((wave - 4) >> 1) + lk, half = wave & 1 ----
  const int thalf = wave & 1;
  int v_src, v_dst, v_hi = 4;                   // (v_hi: sub-columns 4-5 of the tile's patch rows, relative to v_src)
  if constexpr (GEN == 3) v_hi = lk ? -2 : 4;
  {
    const int tile = 2 * ((wave >> 1) & 1) + lk, ty = tile >> 1, tx = tile & 1;
    v_src = li * PS + (4 * ty + thalf) * PW + (GEN == 3 ? 8 : 4) * tx;
    v_dst = (18 * thalf * 4 + tile) * BCI + li;               // V4[18 half + m][tile][ci]: + m * 128
  }
  f32x4 ra[5];
  f32x2 rb[5];
  float tt[3][6];
  auto v_read = [&](int b, int k) __attribute__((always_inline)) {          // k = 0..9: row k/2, columns 0-3 / 4-5
    const float* pp = Pb(b) + v_src + (k >> 1) * PW;
    if (k & 1) rb[k >> 1] = *reinterpret_cast<const f32x2*>(pp + v_hi);
    else ra[k >> 1] = *reinterpret_cast<const f32x4*>(pp);
  };
  auto v_stage1 = [&](auto half_, int c) __attribute__((always_inline)) {
    constexpr int HALF = decltype(half_)::value;
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
  };
  auto v_stage2 = [&](int b, int i) __attribute__((always_inline)) {
    const float X0 = tt[i][0], X1 = tt[i][1], X2 = tt[i][2], X3 = tt[i][3], X4 = tt[i][4], X5 = tt[i][5];
    const float a_ = fmaf(-4.f, X2, X4), b_ = fmaf(-4.f, X1, X3), c_ = X4 - X2, e_ = X3 - X1;
    float* vp = Vb(b) + v_dst + i * 6 * (4 * BCI);
    vp[0 * 4 * BCI] = fmaf(4.f, X0, fmaf(-5.f, X2, X4));
    vp[1 * 4 * BCI] = a_ + b_;
    vp[2 * 4 * BCI] = a_ - b_;
    vp[3 * 4 * BCI] = fmaf(2.f, e_, c_);
    vp[4 * 4 * BCI] = fmaf(-2.f, e_, c_);
    vp[5 * 4 * BCI] = fmaf(4.f, X1, fmaf(-5.f, X3, X5));
  };

  // ---- MFMA role: wave (cb = wave & 1, q = wave >> 1) ----
  const int wcb = wave & 1, wq = wave >> 1;
  const int a_lane = (9 * wq * 4 + lk) * YCO + wcb * 32 + li;             // Yh4[9q + e][2s + lk][cb*32 + li]
  const int b_lane = (9 * wq * 4 + lk) * BCI + li;                        // V4[9q + e][2s + lk][li]
  f32x16 acc[9];
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 9; ++e) acc[e] = zero16;
  auto lds_barrier = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
  constexpr int RBK = 4;

  // One region: M(j) on buffers `buf`; T(j+1) into the other Yh4 / V4 from dyr / the other patch buffer; S(j+2) into patch
  // buffer `buf`, G(j+3) into the piece registers; (dY role) the rows of region j+2 once stage 1 has consumed those of j+1.
  auto step = [&](int j, auto buf_, auto role_) __attribute__((always_inline)) {
    constexpr int buf = decltype(buf_)::value;
    constexpr int ROLE = decltype(role_)::value;            // 0: dY transform, 1 / 2: input transform half 0 / 1
    const unsigned xr3 = xrec_of(min(j + 3, nk - 1)), sr3 = srec_of(min(j + 3, nk - 1));
    const unsigned yr2 = ROLE == 0 ? yrec_of(min(j + 2, nk - 1)) : 0u, sy2 = ROLE == 0 ? srec_of(min(j + 2, nk - 1)) : 0u;
    const float* Yr = Yb(buf) + a_lane;
    const float* Vr = Vb(buf) + b_lane;
    float av[18], bv[18];
#pragma unroll
    for (int m = 0; m < RBK; ++m) {
      av[m] = Yr[((m % 9) * 4 + 2 * (m / 9)) * YCO];
      bv[m] = Vr[((m % 9) * 4 + 2 * (m / 9)) * BCI];
    }
#pragma unroll
    for (int m = 0; m < 18; ++m) {
      const int e = m % 9;
      acc[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[m], acc[e], 0, 0, 0);
      if (m + RBK < 18) {
        av[m + RBK] = Yr[(((m + RBK) % 9) * 4 + 2 * ((m + RBK) / 9)) * YCO];
        bv[m + RBK] = Vr[(((m + RBK) % 9) * 4 + 2 * ((m + RBK) / 9)) * BCI];
      }
      if (m < NIS) {
        s_interior(buf, m);
        g_interior(xr3, sr3, m);
      } else if (m < NIS + NHS) {
        s_halo(buf, m - NIS);
        g_halo(xr3, sr3, m - NIS);
      }
      if constexpr (ROLE == 0) {
        if (m == 0) y_swap();
        if (m < 4) y_stage1(m);
        if (m == 4) y_issue(yr2, sy2);
        if (m >= 5 && m < 17 && ((m - 5) & 1) == 0) y_stage2(buf ^ 1, (m - 5) >> 1);
      } else {
        if (m < 5) { v_read(buf ^ 1, 2 * m); v_read(buf ^ 1, 2 * m + 1); }
        if (m >= 5 && m < 11) v_stage1(std::integral_constant<int, ROLE - 1>{}, m - 5);
        if (m >= 11 && m < 17 && ((m - 11) & 1) == 0) v_stage2(buf ^ 1, (m - 11) >> 1);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    lds_barrier();
  };

  auto run = [&](auto role_) __attribute__((always_inline)) {
    constexpr int ROLE = decltype(role_)::value;
    // prologue: patches of regions 0 and 1 -> LDS, pieces of region 2 and the dY rows of region 0 (then 1) in flight, T(0)
    {
      const unsigned x0r = xrec_of(0), x1r = xrec_of(min(1, nk - 1)), x2r = xrec_of(min(2, nk - 1));
      const unsigned s0r = srec_of(0), s1r = srec_of(min(1, nk - 1)), s2r = srec_of(min(2, nk - 1));
#pragma unroll
      for (int sl = 0; sl < NIS; ++sl) g_interior(x0r, s0r, sl);
#pragma unroll
      for (int sl = 0; sl < NHS; ++sl) g_halo(x0r, s0r, sl);
      if constexpr (ROLE == 0) y_issue(yrec_of(0), s0r);
#pragma unroll
      for (int sl = 0; sl < NIS; ++sl) { s_interior(0, sl); g_interior(x1r, s1r, sl); }
#pragma unroll
      for (int sl = 0; sl < NHS; ++sl) { s_halo(0, sl); g_halo(x1r, s1r, sl); }
#pragma unroll
      for (int sl = 0; sl < NIS; ++sl) { s_interior(1, sl); g_interior(x2r, s2r, sl); }
#pragma unroll
      for (int sl = 0; sl < NHS; ++sl) { s_halo(1, sl); g_halo(x2r, s2r, sl); }
    }
    lds_barrier();
    if constexpr (ROLE == 0) {
      y_swap();
#pragma unroll
      for (int b_ = 0; b_ < 4; ++b_) y_stage1(b_);
      y_issue(yrec_of(min(1, nk - 1)), srec_of(min(1, nk - 1)));
#pragma unroll
      for (int i = 0; i < 6; ++i) y_stage2(0, i);
    } else {
#pragma unroll
      for (int k = 0; k < 10; ++k) v_read(0, k);
#pragma unroll
      for (int c = 0; c < 6; ++c) v_stage1(std::integral_constant<int, ROLE - 1>{}, c);
#pragma unroll
      for (int i = 0; i < 3; ++i) v_stage2(0, i);
    }
    lds_barrier();
    for (int j = 0; j < nk; j += 2) {
      step(j, std::integral_constant<int, 0>{}, role_);
      if (j + 1 < nk) step(j + 1, std::integral_constant<int, 1>{}, role_);
    }
  };
  if (yrole) run(std::integral_constant<int, 0>{});
  else if (thalf) run(std::integral_constant<int, 2>{});
  else run(std::integral_constant<int, 1>{});
  __syncthreads();

  // ---- epilogue: as winow4_kernel ----
  float* const Zs = smem;                       // [xi 36][co 16][ci 32]
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
        float r_[3][6];
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

struct X4Plan { int ryn, rxn, nkt, kps, splits, gridM, gridC, dil, Hs, Ws, gen; };
constexpr double X4_MIN_FILL = 0.7;             // pixels of the map / pixels of its 8x8 regions: below it the F(2x2) form wins
static bool x4_plan(const avsep_conv_desc* d, X4Plan* out) {
  X4Plan p{};
  p.dil = d->dil;
  if (p.dil != 1 && p.dil != 2) return false;
  if (d->H % p.dil || d->W % p.dil) return false;
  p.Hs = d->H / p.dil;
  p.Ws = d->W / p.dil;
  p.ryn = cdiv(p.Hs, 8);
  p.gen = (d->W & 1) ? 2 : p.dil == 2 ? 3 : ((p.Hs & 7) || (p.Ws & 7)) ? 1 : 0;
  // (gen 3: a region is 8 sub-map rows x 8 MAP columns, both x parities)
  p.rxn = cdiv(p.gen == 3 ? d->W : p.Ws, 8);
  const int nsub = p.gen == 3 ? 2 : p.dil * p.dil;
  if ((double)d->H * d->W < X4_MIN_FILL * 64.0 * p.ryn * p.rxn * nsub) return false;
  p.nkt = d->N * nsub * p.ryn * p.rxn;
  p.gridM = cdiv(d->Cout, X4_BCO);
  p.gridC = cdiv(d->Cin, X4_BCI);
  // one workgroup per CU (512 threads x 256 registers, 156 KB of LDS): ONE round of workgroups, as in wgrad_wino.hip
  int want = cu_count() / (p.gridM * p.gridC);
  if (want < 1) want = 1;
  p.kps = cdiv(p.nkt, want);
  if (p.kps > X4P_KPS_MAX) p.kps = X4P_KPS_MAX;
  if (p.kps < 1) p.kps = 1;
  p.splits = cdiv(p.nkt, p.kps);
  *out = p;
  return true;
}

bool x4_applicable(const avsep_conv_desc* d) {
  if ((d->algo & (AVSEP_ALGO_NO_WINOGRAD | AVSEP_ALGO_NO_WINOGRAD_WGRAD | AVSEP_ALGO_NO_WINOGRAD4)) || d->prec != AVSEP_PREC_F32) return false;
  if (!(d->KH == 3 && d->KW == 3 && d->stride == 1 && (d->dil == 1 || d->dil == 2) && d->pad == d->dil) || d->up2x) return false;
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
  snprintf(buf, cap, "8x8%s,split%d", p.gen == 3 ? "p,dil2" : p.gen == 2 ? "e" : p.gen == 1 ? "g" : "", p.splits);
}

int x4_wgrad(const avsep_conv_desc* d, const float* dy, float* dw, float* ws, hipStream_t st) {
  X4Plan p;
  if (!x4_plan(d, &p)) return AVSEP_ERR_ARG;
  X4Args a{};
  a.N = d->N; a.C0 = d->C0; a.C1 = d->Cin - d->C0; a.Cin = d->Cin; a.H = d->H; a.W = d->W; a.Cout = d->Cout;
  a.ryn = p.ryn; a.rxn = p.rxn; a.dil = p.dil; a.Hs = p.Hs; a.Ws = p.Ws; a.nkt = p.nkt; a.kps = p.kps; a.gridM = p.gridM; a.gridC = p.gridC;
  a.act0 = d->act0; a.act1 = d->act1;
  a.x0 = d->x0; a.x1 = d->x1; a.sc0 = d->scale0; a.sh0 = d->shift0; a.sc1 = d->scale1; a.sh1 = d->shift1;
  a.dy = dy; a.out = ws;
  const bool raw = !d->scale0 && !d->scale1 && d->act0 == AVSEP_ACT_NONE && (a.C1 == 0 || d->act1 == AVSEP_ACT_NONE);
  dim3 grid((unsigned)(p.gridM * p.gridC * p.splits));
  if (p.gen == 3) {
    if (raw) hipLaunchKernelGGL((winow4_kernel<true, 3>), grid, dim3(X4_THREADS), 0, st, a);
    else hipLaunchKernelGGL((winow4_kernel<false, 3>), grid, dim3(X4_THREADS), 0, st, a);
  } else if (p.gen == 2) {
    if (raw) hipLaunchKernelGGL((winow4_kernel<true, 2>), grid, dim3(X4_THREADS), 0, st, a);
    else hipLaunchKernelGGL((winow4_kernel<false, 2>), grid, dim3(X4_THREADS), 0, st, a);
  } else if (p.gen == 1) {
    if (raw) hipLaunchKernelGGL((winow4_kernel<true, 1>), grid, dim3(X4_THREADS), 0, st, a);
    else hipLaunchKernelGGL((winow4_kernel<false, 1>), grid, dim3(X4_THREADS), 0, st, a);
  } else {
    if (raw) hipLaunchKernelGGL((winow4_kernel<true, 0>), grid, dim3(X4_THREADS), 0, st, a);
    else hipLaunchKernelGGL((winow4_kernel<false, 0>), grid, dim3(X4_THREADS), 0, st, a);
  }
  AVSEP_LAUNCH_CHECK();
  return w3_reduce(ws, dw, (long long)d->Cout * d->Cin, p.splits, st);
}
