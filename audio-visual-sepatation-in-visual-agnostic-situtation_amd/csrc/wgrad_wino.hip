// Weight gradient of the 3x3 / stride 1 / 'same' convolutions (dilation 1 or 2) in the Winograd form, fp32:
//     dW[kh][kw] = sum over 2x2 tiles of  G^T [ (A dY A^T) (.) (B^T d B) ] G        (the transpose of F(2x2, 3x3))
// dY = the 2x2 cotangent tile, d = the 4x4 input tile around it, B^T as in conv_wino.hip,
//     A = [[1,0],[1,1],[1,-1],[0,-1]],   G^T = [[1,.5,.5,0],[0,.5,-.5,0],[0,.5,.5,1]]:
// 16 products per tile and (co, ci) instead of the 36 of the direct form (conv3x3.hip's wgrad3x3_kernel: 48 ms of the
// 153 ms fp32 step).  Per transform position xi = (i, j) this is a [Cout x tiles] x [tiles x Cin] GEMM whose K
// dimension is the tile index, so one MFMA k-step consumes two tiles (lane half = tile parity) and BOTH operands are
// transformed from LDS patches on their way into the MFMA:
//   * a workgroup (8 waves, two per SIMD) owns 64 output x 64 input channels and sweeps the chunks (16 tiles each) of
//     its K-split; wave (i, cib) owns transform row i for input-channel block cib and both output-channel blocks:
//     4 x 2 accumulator tiles = 128 registers;
//   * per k-step a lane reads the two dY rows of its tile (2 x ds_read_b64 per output-channel block) and the two input
//     rows that B^T's row i combines (4 x ds_read_b64), 20 VALU make 8 + 4 fragments, 8 MFMAs consume them; lanes are
//     CHANNELS here, so the per-channel LDS strides are 2 * odd words: 32 lanes x 8 bytes hit 64 distinct banks;
//   * signs and the 1/2 factors of A and G are moved out of the loop (row 3 of A is used as (0, +1); the output
//     transform applies s = (1, 1, 1, -1) and G^T once per workgroup);
//   * staging follows conv_wino.hip: global loads of chunk c+2 and LDS stores of chunk c+1 hang behind individual MFMAs
//     of chunk c, one barrier per chunk; the folded BatchNorm affine + activation of the input (and the two-source
//     concat: a 64-channel block lies in one source) are applied on the way in;
//   * the workgroup writes one tap-major partial slab [split][tap][Cout][Cin]; conv3x3.hip's w3_reduce_kernel sums the
//     slabs (deterministic, no atomics).
// Dilation 2: the four parity sub-images are independent undilated problems that add into the same dW; a chunk takes
// the same 4x8 sub-pixel region of BOTH column parities of one row-parity class, so its loads stay 8-byte pairs of
// adjacent full-resolution pixels.
#include <stdlib.h>
#include <type_traits>

#include "common.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int WW_THREADS = 512, WW_B = 64;      // 64 x 64 channels per workgroup
constexpr int WW_GT_MAX = 512;                  // group records of one K-split kept in LDS (host-checked)

struct WwArgs {
  int N, C0, C1, Cin, H, W, Cout;
  int Hq, Wq, gyn, gxn, ngroups, nchunks, cps;  // (sub-)image geometry, tile groups, chunks in total / per split
  int gridM, gridC, act0, act1;
  const float *x0, *x1, *sc0, *sh0, *sc1, *sh1;
  const float* dy;
  float* out;                                   // slabs [split][tap][Cout][Cin]
};

// G groups of GH x GW tiles per chunk (G*GH*GW = 16 tiles = 8 k-steps); SUB: dilation 2 (parity sub-images); RAW: the
// input needs no affine and no activation.
template <int G, int GH, int GW, bool SUB, bool RAW>
__global__ __launch_bounds__(WW_THREADS) void winow_kernel(WwArgs a) {
  static_assert(G * GH * GW == 16 && (GW % 2) == 0, "16 tiles per chunk, tile pairs inside a tile row");
  constexpr int NT = WW_THREADS, B = WW_B;
  // The input patch of a group is staged in even-aligned COLUMN PAIRS starting two pixels left of the group (pair k =
  // pixels x0-2+2k, +1): one 8-byte global load, one mask and one offset per pair (W and x0 are even, so a pair is
  // inside or outside the image together); in LDS pair k sits at columns 2k+1, 2k+2 of a (2*GW+6)-wide row so that the
  // columns a tile reads (x0+2tx-1 .. +2) start at the even column 2tx+2.
  constexpr int PHG = 2 * GH + 2, KP = GW + 2, PCG = 2 * GW + 6, GE = PHG * PCG, XCH = G * GE;
  constexpr int PSX = ((XCH / 2) | 1) * 2;                   // per-channel stride of the input patch: 2 * odd words
  constexpr int GD = 4 * GH * GW, PSD = 66;                  // dY: 64 pixels per channel per chunk, stride 2 * 33
  constexpr int PP = PHG * KP, NXG = B * PP;                 // column pairs per channel / per group
  // Dilation 2 (SUB): the G = 2 groups of a chunk are the two COLUMN parities of one region of one row-parity class, so an
  // 8-byte load of two adjacent full-resolution pixels feeds one element of each group (stride-2 scalar loads otherwise).
  static_assert(!SUB || G == 2, "dilation 2: the two groups of a chunk are the column parities");
  constexpr int SG = SUB ? 1 : G;                            // groups as the STAGING code sees them
  constexpr int PEG = (NXG + NT - 1) / NT, PEX = SG * PEG;   // input pieces per thread: group-major, so the group of a piece is static
  constexpr int PDG = (SUB ? B * GD : B * GD / 2) / NT;      // dY pieces per thread per group: one float2 each
  constexpr int PED = SG * PDG;
  static_assert((SUB ? B * GD : B * GD / 2) % NT == 0, "whole dY pieces");
  constexpr int NPIECE = PEX + PED;
  static_assert(NPIECE <= 32, "valid bits of the pieces fit one register");
  constexpr int X_FLOATS = B * PSX, D_FLOATS = B * PSD;
  static_assert(X_FLOATS < (1 << 14) && 2 * X_FLOATS >= 8 * 16 * 64, "epilogue exchange fits the input buffers");
  __shared__ __attribute__((aligned(16))) float Xs[2][X_FLOATS];
  __shared__ __attribute__((aligned(16))) float Ds[2][D_FLOATS];
  __shared__ __attribute__((aligned(16))) int gtab[WW_GT_MAX][4];     // {image, pixel offset of the group origin, y0 << 16 | x0, valid}
  __shared__ f32x2 aff[B];                     // (scale, shift) of the 64 input channels

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lk = lane >> 5;
  const int wi = wave & 3, wcb = wave >> 2;
  // 1-D grid, XCD-aware (round 5): the gridM * gridC workgroups of one K-split read the same pixels; consecutive LOGICAL ids
  // share an XCD, so those pixels are fetched into one L2 instead of eight (wgrad_wino4.hip: -10 % on the 256 -> 64 layer)
  const int t_ = xcd_remap(blockIdx.x, gridDim.x), per_split_ = a.gridM * a.gridC;
  const int split = t_ / per_split_, lin_ = t_ % per_split_;
  const int mt = lin_ % a.gridM, ct = lin_ / a.gridM;
  const int m0 = mt * B, c0 = ct * B;
  const int iHW = a.H * a.W;
  // the 64-channel input block lies in one source
  const bool src1 = c0 >= a.C0;
  const float* const xs = src1 ? a.x1 : a.x0;
  const int Cs = src1 ? a.C1 : a.C0, cs0 = src1 ? c0 - a.C0 : c0;
  const float slope = act_slope(src1 ? a.act1 : a.act0);
  const int ch0 = split * a.cps, nc = min(a.cps, a.nchunks - ch0);     // chunks [ch0, ch0 + nc) of this split

  // ---- group records of the whole split: origin offsets (input and dY), patch origin, validity ----
  const int per = a.gyn * a.gxn;
  for (int e = tid; e < nc * SG; e += NT) {
    const int gid = ch0 * SG + e, gidc = min(gid, a.ngroups - 1);
    const int img = gidc / per, gy = (gidc % per) / a.gxn, gx = gidc % a.gxn;     // SUB: img = 2 * n + row parity
    const int y0 = gy * 2 * GH, x0 = gx * 2 * GW;
    const int n = SUB ? img >> 1 : img, ph = SUB ? img & 1 : 0;
    // element offset of (sub-)pixel (y0, x0) [column parity 0] inside channel 0 of image n, WITHOUT the channel term
    const int pix = SUB ? (2 * y0 + ph) * a.W + 2 * x0 : y0 * a.W + x0;
    gtab[e][0] = n;
    gtab[e][1] = pix;
    gtab[e][2] = (y0 << 16) | x0;
    gtab[e][3] = gid < a.ngroups;
  }
  if constexpr (!RAW) {
    if (tid < B) {
      const float* sc = src1 ? a.sc1 : a.sc0;
      const float* sh = src1 ? a.sh1 : a.sh0;
      aff[tid] = f32x2{sc ? sc[cs0 + tid] : 1.f, sc ? sh[cs0 + tid] : 0.f};
    }
  }
  __syncthreads();

  // ---- per-thread piece constants (the same for every chunk; only the group origins change) ----
  // input piece (g, e): channel ci, patch row r, column pair k.  xk = LDS slot of the pair's first column | r << 14 | k << 18 |
  // ci << 24; xrel = element offset from the patch origin (y0-1, x0-2).
  constexpr int PSTEP = SUB ? 2 : 1;
  unsigned xk[PEX];
  int xrel[PEX];
#pragma unroll
  for (int p = 0; p < PEX; ++p) {
    const int g = p / PEG, e = p % PEG;
    const int idx = min(tid + NT * e, NXG - 1);
    const int ci = idx / PP, rem = idx % PP, r = rem / KP, k = rem % KP;
    xk[p] = (unsigned)(ci * PSX + g * GE + r * PCG + 2 * k + 1) | ((unsigned)r << 14) | ((unsigned)k << 18) | ((unsigned)ci << 24);
    xrel[p] = ci * iHW + (r * a.W + 2 * k) * PSTEP;          // SUB: full-resolution row 2r, column 4k from the patch origin
  }
  // dY piece (g, e): channel co, row, column (pairs when !SUB) of the group.  dk = LDS slot | row << 14 | col << 18 | co << 24;
  // drel = element offset from the group origin (y0, x0)
  unsigned dk[PED];
  int drel[PED];
#pragma unroll
  for (int p = 0; p < PED; ++p) {
    const int g = p / PDG, e = p % PDG;
    const int idx = tid + NT * e;
    const int co = SUB ? idx / GD : idx / (GD / 2), rem = SUB ? idx % GD : (idx % (GD / 2)) * 2;    // (sub-)pixel index inside the group
    const int row = rem / (2 * GW), col = rem % (2 * GW);
    dk[p] = (unsigned)(co * PSD + g * GD + rem) | ((unsigned)row << 14) | ((unsigned)col << 18) | ((unsigned)co << 24);
    drel[p] = co * iHW + (row * a.W + col) * PSTEP;
  }
  f32x2 xraw[PEX], xraw2[SUB ? PEX : 1];   // SUB: (column parity 0, 1) at sub-column x, and at x + 1
  f32x2 draw[PED];                          // SUB: (column parity 0, 1) of one sub-pixel
  unsigned okbits = 0, okhi = 0;     // piece valid; SUB: second pixel of an input pair valid (odd sub-image widths)

  // group records of the chunk being loaded, wave-uniform (scalar registers)
  int g_n[SG], g_pix[SG], g_y0[SG], g_x0[SG], g_ok[SG];
  auto load_groups = [&](int c) __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < SG; ++g) {
      const int4 gt = *reinterpret_cast<const int4*>(gtab[c * SG + g]);
      g_n[g] = __builtin_amdgcn_readfirstlane(gt.x);
      g_pix[g] = __builtin_amdgcn_readfirstlane(gt.y);
      g_y0[g] = __builtin_amdgcn_readfirstlane(gt.z >> 16);
      g_x0[g] = __builtin_amdgcn_readfirstlane(gt.z & 0xffff);
      g_ok[g] = __builtin_amdgcn_readfirstlane(gt.w);
    }
  };
  // piece pc of the chunk whose groups are loaded: BUFFER load into registers — scalar resource, scalar group offset, the
  // piece's constant byte offset in a vector register; a masked piece gets the offset 0xffffffff, which no buffer covers,
  // and loads as 0 (no 64-bit vector address arithmetic, and for RAW inputs no valid bits and no selects at the store:
  // the f32 MFMA does not overlap vector instructions).  The input resource starts (W + 2) * PSTEP elements BEFORE the
  // source so that the patch origin (y0-1, x0-2) of every group is a non-negative scalar offset; nothing is read there.
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(xs - (a.W + 2) * PSTEP), 0, 0xfffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, 0xfffffff0, 0x00020000);
  auto ld2 = [](__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) __attribute__((always_inline)) {
    return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0));
  };
  auto issue_piece = [&](int pc) __attribute__((always_inline)) {
    if (pc < PEX) {
      const int g = pc / PEG, e = pc % PEG;
      const int r = (xk[pc] >> 14) & 15, k = (xk[pc] >> 18) & 31;
      const int y = g_y0[g] - 1 + r, x = g_x0[g] - 2 + 2 * k;
      const bool ok = (PEG * NT == NXG || tid + NT * e < NXG) && g_ok[g] && (unsigned)y < (unsigned)a.Hq && (unsigned)x < (unsigned)a.Wq;
      const unsigned soff = 4u * (unsigned)((g_n[g] * Cs + cs0) * iHW + g_pix[g]);      // N * C * H * W < 2^30: host check
      const unsigned off = ok ? 4u * (unsigned)xrel[pc] : 0xffffffffu;
      if constexpr (SUB) {
        const bool ok1 = ok && x + 1 < a.Wq;
        xraw[pc] = ld2(rs_x, off, soff);
        xraw2[pc] = ld2(rs_x, ok1 ? off + 8 : 0xffffffffu, soff);
        if constexpr (!RAW) okhi = (okhi & ~(1u << pc)) | ((unsigned)ok1 << pc);
      } else {
        xraw[pc] = ld2(rs_x, off, soff);
      }
      if constexpr (!RAW) okbits = (okbits & ~(1u << pc)) | ((unsigned)ok << pc);
    } else if (pc < NPIECE) {
      const int p = pc - PEX, g = p / PDG;
      const int row = (dk[p] >> 14) & 15, col = (dk[p] >> 18) & 31, co = (int)(dk[p] >> 24);
      const bool ok = g_ok[g] && g_y0[g] + row < a.Hq && g_x0[g] + col < a.Wq && m0 + co < a.Cout;   // W, col even: a pair is in or out together
      const unsigned soff = 4u * (unsigned)((g_n[g] * a.Cout + m0) * iHW + g_pix[g]);
      draw[p] = ld2(rs_d, ok ? 4u * (unsigned)drel[p] : 0xffffffffu, soff);
    }
  };
  auto affine_act2 = [&](f32x2 v, int ci) __attribute__((always_inline)) {
    const f32x2 ss = aff[ci];
    v = __builtin_elementwise_fma(v, f32x2{ss[0], ss[0]}, f32x2{ss[1], ss[1]});       // v_pk_fma_f32 / v_pk_mul_f32
    const f32x2 w2 = v * f32x2{slope, slope};
    return f32x2{fmaxf(v[0], w2[0]), fmaxf(v[1], w2[1])};
  };
  auto finish_piece = [&](int buf, int pc) __attribute__((always_inline)) {
    if (pc < PEX) {
      f32x2 v = xraw[pc];
      if constexpr (!RAW) v = affine_act2(v, xk[pc] >> 24);
      const bool ok = RAW || ((okbits >> pc) & 1u);            // RAW: a masked piece already loaded as 0
      const bool ok1 = RAW || (SUB ? (bool)((okhi >> pc) & 1u) : ok);
      if (PEG * NT == NXG || tid + NT * (pc % PEG) < NXG) {
        float* q = &Xs[buf][xk[pc] & 0x3fffu];
        if constexpr (SUB) {                  // v = both column parities at sub-column x, v2 at x + 1: one element of each group each
          f32x2 v2 = xraw2[pc];
          if constexpr (!RAW) v2 = affine_act2(v2, xk[pc] >> 24);
          q[0] = ok ? v[0] : 0.f;
          q[1] = ok1 ? v2[0] : 0.f;
          q[GE] = ok ? v[1] : 0.f;
          q[GE + 1] = ok1 ? v2[1] : 0.f;
        } else {
          q[0] = ok ? v[0] : 0.f;
          q[1] = ok1 ? v[1] : 0.f;
        }
      }
    } else if (pc < NPIECE) {
      const int p = pc - PEX;                 // (a masked dY piece loaded as 0)
      if constexpr (SUB) {
        Ds[buf][dk[p] & 0x3fffu] = draw[p][0];
        Ds[buf][(dk[p] & 0x3fffu) + GD] = draw[p][1];
      } else {
        *reinterpret_cast<f32x2*>(&Ds[buf][dk[p] & 0x3fffu]) = draw[p];
      }
    }
  };

  f32x16 acc[4][2];         // [xi column j][output-channel block]
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][cb][r] = 0.f;

  // row i of B^T on the input:  t = d[rA] + sgn * d[rB]   ((0,2,-) (1,2,+) (2,1,-) (1,3,-))
  const int rA = wi == 0 ? 0 : (wi == 2 ? 2 : 1), rB = wi == 2 ? 1 : (wi == 3 ? 3 : 2);
  const float sgn = wi == 1 ? 1.f : -1.f;
  // row i of A on dY (row 3 taken as (0, +1): the sign moves to the output transform):  t = al * dY[0] + be * dY[1]
  const float al = wi == 3 ? 0.f : 1.f, be = wi == 0 ? 0.f : (wi == 2 ? -1.f : 1.f);
  const int lbx = (wcb * 32 + li) * PSX + 2 * lk;            // + rA/rB * PCG + tile(ks)
  const int lbd = li * PSD + 2 * lk;                         // + cob * 32 * PSD + tile(ks)

  float u[2][2][4], v[2][4];           // [slot][output-channel block][j], [slot][j]
  f32x2 d0[2], d1[2], xa[2], xb[2];    // dY rows 0 / 1 per output-channel block; input rows A / B as two column pairs
  float ta[2][2], tb[4];
  auto tile_x = [&](int ks) { const int t = 2 * ks, g = t / (GH * GW), ty = (t % (GH * GW)) / GW, tx = t % GW; return g * GE + 2 * ty * PCG + 2 * tx + 2; };
  auto tile_d = [&](int ks) { const int t = 2 * ks, g = t / (GH * GW), ty = (t % (GH * GW)) / GW, tx = t % GW; return g * GD + 2 * ty * 2 * GW + 2 * tx; };
  auto read_d = [&](int buf, int ks, int cob) __attribute__((always_inline)) {
    const float* p = &Ds[buf][lbd + cob * 32 * PSD + tile_d(ks)];
    d0[cob] = *reinterpret_cast<const f32x2*>(p);
    d1[cob] = *reinterpret_cast<const f32x2*>(p + 2 * GW);
  };
  auto read_x = [&](int buf, int ks, int which) __attribute__((always_inline)) {
    const float* p = &Xs[buf][lbx + (which ? rB : rA) * PCG + tile_x(ks)];
    f32x2* dst = which ? xb : xa;
    dst[0] = *reinterpret_cast<const f32x2*>(p);
    dst[1] = *reinterpret_cast<const f32x2*>(p + 2);
  };
  auto xform_d = [&](int slot, int cob, int st) __attribute__((always_inline)) {
    if (st == 0) { ta[cob][0] = fmaf(be, d1[cob][0], al * d0[cob][0]); ta[cob][1] = fmaf(be, d1[cob][1], al * d0[cob][1]); }
    if (st == 1) {
      u[slot][cob][0] = ta[cob][0];
      u[slot][cob][1] = ta[cob][0] + ta[cob][1];
      u[slot][cob][2] = ta[cob][0] - ta[cob][1];
      u[slot][cob][3] = ta[cob][1];
    }
  };
  auto xform_x = [&](int slot, int st) __attribute__((always_inline)) {
    if (st == 0) {
      tb[0] = fmaf(sgn, xb[0][0], xa[0][0]); tb[1] = fmaf(sgn, xb[0][1], xa[0][1]);
      tb[2] = fmaf(sgn, xb[1][0], xa[1][0]); tb[3] = fmaf(sgn, xb[1][1], xa[1][1]);
    }
    if (st == 1) { v[slot][0] = tb[0] - tb[2]; v[slot][1] = tb[1] + tb[2]; v[slot][2] = tb[2] - tb[1]; v[slot][3] = tb[1] - tb[3]; }
  };

  // ---- prologue: chunk 0 -> buffer 0, loads of chunk 1 in flight, operands of k-step 0 ----
  load_groups(0);
#pragma unroll
  for (int pc = 0; pc < NPIECE; ++pc) issue_piece(pc);
#pragma unroll
  for (int pc = 0; pc < NPIECE; ++pc) finish_piece(0, pc);
  __syncthreads();
  load_groups(min(1, nc - 1));
#pragma unroll
  for (int pc = 0; pc < NPIECE; ++pc) issue_piece(pc);
  read_d(0, 0, 0); read_d(0, 0, 1); read_x(0, 0, 0); read_x(0, 0, 1);
#pragma unroll
  for (int cob = 0; cob < 2; ++cob) { xform_d(0, cob, 0); xform_d(0, cob, 1); }
  xform_x(0, 0); xform_x(0, 1);

  // ---- main loop: 8 k-steps (blocks of 8 MFMAs) per chunk; everything else rides behind individual MFMAs:
  //   MFMA 0-3 : the next block's four operand read pairs          MFMA 3-7 : its transforms (20 VALU)
  //   blocks 4-5 : LDS stores of chunk c+1 (loaded six blocks earlier); barrier after block 6 (the last block that
  //   reads this chunk's buffers); block 7 prefetches from the other buffer; blocks 6-7 : global loads of chunk c+2.
  // Staging is unconditional (chunk indices clamped): a branch around it would fork the accumulator state.
  // (two chunks per trip, LDS buffer index a compile-time constant: every LDS address of the loop is base + immediate)
  auto chunk = [&](int c, auto buf_) __attribute__((always_inline)) {
    constexpr int buf = decltype(buf_)::value;
    const int c2 = min(c + 2, nc - 1);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int cur = ks & 1, nxt = cur ^ 1;
      const int nbuf = ks < 7 ? buf : buf ^ 1, nks = ks < 7 ? ks + 1 : 0;
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const int q = m >> 1, cb = m & 1;
        acc[q][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[cur][cb][q], v[cur][q], acc[q][cb], 0, 0, 0);
        if (m == 0) read_d(nbuf, nks, 0);
        if (m == 1) read_d(nbuf, nks, 1);
        if (m == 2) read_x(nbuf, nks, 0);
        if (m == 3) { read_x(nbuf, nks, 1); xform_d(nxt, 0, 0); }
        if (m == 4) { xform_d(nxt, 0, 1); xform_d(nxt, 1, 0); }
        if (m == 5) xform_d(nxt, 1, 1);
        if (m == 6) xform_x(nxt, 0);
        if (m == 7) xform_x(nxt, 1);
        if (ks == 4 || ks == 5) {
#pragma unroll
          for (int pc = (ks - 4) * 8 + m; pc < NPIECE; pc += 16) finish_piece(buf ^ 1, pc);
        }
        if (ks == 6 || ks == 7) {
          if (ks == 6 && m == 0) load_groups(c2);
#pragma unroll
          for (int pc = (ks - 6) * 8 + m; pc < NPIECE; pc += 16) issue_piece(pc);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (ks == 6) __syncthreads();
    }
  };
  for (int c = 0; c < nc; c += 2) {
    chunk(c, std::integral_constant<int, 0>{});
    if (c + 1 < nc) chunk(c + 1, std::integral_constant<int, 1>{});
  }
  __syncthreads();          // the last block's operand prefetch has read LDS: drain before the epilogue reuses it

  // ---- epilogue: dW = G^T (s s^T . M) G.  Columns (j -> kw) in registers; rows (i -> kh) through LDS, one (output-channel
  //      block, kw) plane set at a time: 8 waves x 16 x 64 floats = 32 KB.  Wave (i < 3, cib) then owns tap row kh = i.
  //      C/D map: ci = lane & 31, co row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5). ----
  float* const Zs = &Xs[0][0];                // [wave][r][lane]
  const int ci = c0 + wcb * 32 + li;
#pragma unroll
  for (int cob = 0; cob < 2; ++cob) {
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float m0_ = acc[0][cob][r], m1 = acc[1][cob][r], m2 = acc[2][cob][r], m3 = acc[3][cob][r];
        const float z = kw == 0 ? m0_ + 0.5f * (m1 + m2) : (kw == 1 ? 0.5f * (m1 - m2) : 0.5f * (m1 + m2) - m3);
        Zs[(wave * 16 + r) * 64 + lane] = z;
      }
      __syncthreads();
      if (wi < 3) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float* z = Zs + ((wcb * 4) * 16 + r) * 64 + lane;      // + i * 1024
          const float z1 = z[1024], z2 = z[2048];
          const float w_ = wi == 0 ? z[0] + 0.5f * (z1 + z2) : (wi == 1 ? 0.5f * (z1 - z2) : 0.5f * (z1 + z2) - z[3072]);
          const int co = m0 + cob * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
          if (co < a.Cout && ci < a.Cin) a.out[(((long long)split * 9 + wi * 3 + kw) * a.Cout + co) * a.Cin + ci] = w_;
        }
      }
      __syncthreads();
    }
  }
}

// ===========================================================================
// 4x4 / stride 2 / pad 1 weight gradient (the U-Net encoder's down convs, audio_net.py:57-58,170-171) on the same skeleton.
// An output pixel (oy, ox) reads the input window rows 2oy-1 .. 2oy+2, columns 2ox-1 .. 2ox+2 — exactly the 4x4 input tile
// of Winograd tile (oy, ox) — and dW[kh][kw] accumulates dY[oy][ox] * d[kh][kw]: the 16 taps ARE the 16 accumulator
// positions, with identity transforms.  Same staging of the input patch, same wave layout (wave (kh, ci block) owns the
// four taps of row kh for both output-channel blocks), no VALU in the operand path (one ds_read_b32 per dY fragment, two
// ds_read_b64 per input row), no exchange in the epilogue.  K-step = two output pixels, a chunk = 16 of them.
// ===========================================================================
struct W4dArgs {
  int N, Cin, H, W, Cout, Ho, Wo;
  int gyn, gxn, ngroups, nchunks, cps;
  int gridM, gridC, act0;
  const float *x0, *sc0, *sh0;
  const float* dy;
  float* out;                                   // slabs [split][16 taps][Cout][Cin]
};

template <int G, int GH, int GW, bool RAW>
__global__ __launch_bounds__(WW_THREADS) void wgrad4d_kernel(W4dArgs a) {
  static_assert(G * GH * GW == 16 && (GW % 2) == 0, "16 output pixels per chunk, pixel pairs inside a row");
  constexpr int NT = WW_THREADS, B = WW_B;
  constexpr int PHG = 2 * GH + 2, KP = GW + 2, PCG = 2 * GW + 6, GE = PHG * PCG, XCH = G * GE;
  constexpr int PSX = ((XCH / 2) | 1) * 2;
  constexpr int GD = GH * GW, PSD = 17;                      // dY: 16 output pixels per channel per chunk, odd stride (32 banks)
  constexpr int PP = PHG * KP, NXG = B * PP;
  constexpr int PEG = (NXG + NT - 1) / NT, PEX = G * PEG;
  static_assert(B * GD / 2 * G == NT, "one dY pair per thread and chunk");
  constexpr int NPIECE = PEX + 1;
  static_assert(NPIECE <= 32, "valid bits of the pieces fit one register");
  constexpr int X_FLOATS = B * PSX, D_FLOATS = B * PSD;
  static_assert(X_FLOATS < (1 << 14), "LDS slot field");
  __shared__ __attribute__((aligned(16))) float Xs[2][X_FLOATS];
  __shared__ __attribute__((aligned(16))) float Ds[2][D_FLOATS];
  __shared__ __attribute__((aligned(16))) int gtab[WW_GT_MAX][4];
  __shared__ f32x2 aff[B];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lk = lane >> 5;
  const int wi = wave & 3, wcb = wave >> 2;
  // 1-D grid, XCD-aware (round 5): the gridM * gridC workgroups of one K-split read the same pixels; consecutive LOGICAL ids
  // share an XCD, so those pixels are fetched into one L2 instead of eight (wgrad_wino4.hip: -10 % on the 256 -> 64 layer)
  const int t_ = xcd_remap(blockIdx.x, gridDim.x), per_split_ = a.gridM * a.gridC;
  const int split = t_ / per_split_, lin_ = t_ % per_split_;
  const int mt = lin_ % a.gridM, ct = lin_ / a.gridM;
  const int m0 = mt * B, c0 = ct * B;
  const long long HW = (long long)a.H * a.W, HWo = (long long)a.Ho * a.Wo;
  const int iHW = a.H * a.W, iHWo = a.Ho * a.Wo;
  const float slope = act_slope(a.act0);
  const int ch0 = split * a.cps, nc = min(a.cps, a.nchunks - ch0);

  const int per = a.gyn * a.gxn;
  for (int e = tid; e < nc * G; e += NT) {
    const int gid = ch0 * G + e, gidc = min(gid, a.ngroups - 1);
    const int img = gidc / per, gy = (gidc % per) / a.gxn, gx = gidc % a.gxn;
    const int y0 = gy * 2 * GH, x0 = gx * 2 * GW;             // input-pixel origin of the group (= 2 * output origin)
    gtab[e][0] = img;
    gtab[e][1] = y0 * a.W + x0;
    gtab[e][2] = (y0 << 16) | x0;
    gtab[e][3] = gid < a.ngroups;
  }
  if constexpr (!RAW) {
    if (tid < B) aff[tid] = f32x2{a.sc0 ? a.sc0[c0 + tid] : 1.f, a.sc0 ? a.sh0[c0 + tid] : 0.f};
  }
  __syncthreads();

  unsigned xk[PEX];
  int xrel[PEX];
#pragma unroll
  for (int p = 0; p < PEX; ++p) {
    const int g = p / PEG, e = p % PEG;
    const int idx = min(tid + NT * e, NXG - 1);
    const int ci = idx / PP, rem = idx % PP, r = rem / KP, k = rem % KP;
    xk[p] = (unsigned)(ci * PSX + g * GE + r * PCG + 2 * k + 1) | ((unsigned)r << 14) | ((unsigned)k << 18) | ((unsigned)ci << 24);
    xrel[p] = ci * iHW + r * a.W + 2 * k;
  }
  // the thread's dY pair: group dg, channel dco, output row drow, output column pair dcol of the group
  constexpr int DPG = B * GD / 2;                             // pairs per group
  const int dg = tid / DPG, dco = (tid % DPG) / (GD / 2), drem = (tid % (GD / 2)) * 2, drow = drem / GW, dcol = drem % GW;
  const int dslot = dco * PSD + dg * GD + drem, drel = dco * iHWo + drow * a.Wo + dcol;
  f32x2 xraw[PEX], draw;
  unsigned okbits = 0;

  int g_n[G], g_pix[G], g_y0[G], g_x0[G], g_ok[G];
  auto load_groups = [&](int c) __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int4 gt = *reinterpret_cast<const int4*>(gtab[c * G + g]);
      g_n[g] = __builtin_amdgcn_readfirstlane(gt.x);
      g_pix[g] = __builtin_amdgcn_readfirstlane(gt.y);
      g_y0[g] = __builtin_amdgcn_readfirstlane(gt.z >> 16);
      g_x0[g] = __builtin_amdgcn_readfirstlane(gt.z & 0xffff);
      g_ok[g] = __builtin_amdgcn_readfirstlane(gt.w);
    }
  };
  // buffer loads as in winow_kernel: scalar resource and group offset, the piece's constant byte offset in a vector register,
  // 0xffffffff (outside every buffer: loads as 0) for a masked piece — no 64-bit vector adds, no valid bits / selects
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x0 - (a.W + 2)), 0, 0xfffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, 0xfffffff0, 0x00020000);
  auto ld2 = [](__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) __attribute__((always_inline)) {
    return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0));
  };
  auto issue_piece = [&](int pc) __attribute__((always_inline)) {
    if (pc < PEX) {
      const int g = pc / PEG, e = pc % PEG;
      const int r = (xk[pc] >> 14) & 15, k = (xk[pc] >> 18) & 31;
      const int y = g_y0[g] - 1 + r, x = g_x0[g] - 2 + 2 * k;
      const bool ok = (PEG * NT == NXG || tid + NT * e < NXG) && g_ok[g] && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
      const unsigned soff = 4u * (unsigned)((g_n[g] * a.Cin + c0) * (int)HW + g_pix[g]);     // N * C * H * W < 2^30: host check
      xraw[pc] = ld2(rs_x, ok ? 4u * (unsigned)xrel[pc] : 0xffffffffu, soff);
      if constexpr (!RAW) okbits = (okbits & ~(1u << pc)) | ((unsigned)ok << pc);
    } else if (pc < NPIECE) {
      // the group index of the thread's dY pair is a lane value when G > 1: select among the scalar records
      int n = g_n[0], y0 = g_y0[0], x0 = g_x0[0], gok = g_ok[0];
#pragma unroll
      for (int g = 1; g < G; ++g)
        if (dg == g) { n = g_n[g]; y0 = g_y0[g]; x0 = g_x0[g]; gok = g_ok[g]; }
      const int oy = (y0 >> 1) + drow, ox = (x0 >> 1) + dcol;
      const bool ok = gok && oy < a.Ho && ox < a.Wo && m0 + dco < a.Cout;       // Wo, dcol even: a pair is in or out together
      const unsigned off = (unsigned)((long long)(n * a.Cout + m0) * HWo) + (unsigned)(__mul24(y0 >> 1, a.Wo) + (x0 >> 1) + drel);
      draw = ld2(rs_d, ok ? 4u * off : 0xffffffffu, 0u);
    }
  };
  auto finish_piece = [&](int buf, int pc) __attribute__((always_inline)) {
    if (pc < PEX) {
      f32x2 v = xraw[pc];
      if constexpr (!RAW) {
        const f32x2 ss = aff[xk[pc] >> 24];
        v = __builtin_elementwise_fma(v, f32x2{ss[0], ss[0]}, f32x2{ss[1], ss[1]});
        const f32x2 w2 = v * f32x2{slope, slope};
        v = f32x2{fmaxf(v[0], w2[0]), fmaxf(v[1], w2[1])};
        const bool ok = (okbits >> pc) & 1u;                     // (a RAW piece outside the image already loaded as 0)
        v = f32x2{ok ? v[0] : 0.f, ok ? v[1] : 0.f};
      }
      if (PEG * NT == NXG || tid + NT * (pc % PEG) < NXG) {
        float* q = &Xs[buf][xk[pc] & 0x3fffu];
        q[0] = v[0];
        q[1] = v[1];
      }
    } else if (pc < NPIECE) {
      Ds[buf][dslot] = draw[0];
      Ds[buf][dslot + 1] = draw[1];
    }
  };

  f32x16 acc[4][2];         // [kw][output-channel block]
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][cb][r] = 0.f;

  const int lbx = (wcb * 32 + li) * PSX + 2 * lk + wi * PCG;   // input row kh = wave row of the pixel's window
  const int lbd = li * PSD + lk;
  float u[2][2];
  f32x2 v[2][2];
  auto tile_x = [&](int ks) { const int t = 2 * ks, g = t / (GH * GW), ty = (t % (GH * GW)) / GW, tx = t % GW; return g * GE + 2 * ty * PCG + 2 * tx + 2; };
  auto read_ops = [&](int buf, int ks, int slot, int part) __attribute__((always_inline)) {
    if (part == 0) {
      u[slot][0] = Ds[buf][lbd + 2 * ks];
      u[slot][1] = Ds[buf][lbd + 32 * PSD + 2 * ks];
    } else {
      const float* p = &Xs[buf][lbx + tile_x(ks)];
      v[slot][0] = *reinterpret_cast<const f32x2*>(p);
      v[slot][1] = *reinterpret_cast<const f32x2*>(p + 2);
    }
  };

  load_groups(0);
#pragma unroll
  for (int pc = 0; pc < NPIECE; ++pc) issue_piece(pc);
#pragma unroll
  for (int pc = 0; pc < NPIECE; ++pc) finish_piece(0, pc);
  __syncthreads();
  load_groups(min(1, nc - 1));
#pragma unroll
  for (int pc = 0; pc < NPIECE; ++pc) issue_piece(pc);
  read_ops(0, 0, 0, 0);
  read_ops(0, 0, 0, 1);

  for (int c = 0; c < nc; ++c) {
    const int buf = c & 1;
    const int c2 = min(c + 2, nc - 1);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int cur = ks & 1, nxt = cur ^ 1;
      const int nbuf = ks < 7 ? buf : buf ^ 1, nks = ks < 7 ? ks + 1 : 0;
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const int q = m >> 1, cb = m & 1;
        acc[q][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[cur][cb], v[cur][q >> 1][q & 1], acc[q][cb], 0, 0, 0);
        if (m == 0) read_ops(nbuf, nks, nxt, 0);
        if (m == 1) read_ops(nbuf, nks, nxt, 1);
        if (ks == 4 || ks == 5) {
#pragma unroll
          for (int pc = (ks - 4) * 8 + m; pc < NPIECE; pc += 16) finish_piece(buf ^ 1, pc);
        }
        if (ks == 6 || ks == 7) {
          if (ks == 6 && m == 0) load_groups(c2);
#pragma unroll
          for (int pc = (ks - 6) * 8 + m; pc < NPIECE; pc += 16) issue_piece(pc);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (ks == 6) __syncthreads();
    }
  }

  // epilogue: tap (kh = wave row, kw = q) straight to the slab; ci = lane & 31, co row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  const int ci = c0 + wcb * 32 + li;
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int cob = 0; cob < 2; ++cob)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = m0 + cob * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (co < a.Cout && ci < a.Cin) a.out[(((long long)split * 16 + wi * 4 + q) * a.Cout + co) * a.Cin + ci] = acc[q][cob][r];
      }
}

// ---------------------------------------------------------------------------
// host side (called from conv.hip)
// ---------------------------------------------------------------------------
int w3_reduce(const float* ws, float* dw, long long P, int splits, hipStream_t st);      // conv3x3.hip

struct WwCfg { int gh, gw, g; };     // pixels per group, groups per chunk
static const WwCfg WW_CFGS[4] = {{4, 16, 1}, {8, 8, 1}, {2, 16, 2}, {4, 8, 1}};     // [3]: dilation 2 (4x8 sub-pixels x both column parities)
static int ww_cfg(int Hq, int Wq, int tune) {
  const int force = (tune >> 4) & 15;                   // avsep_conv_desc.tune: measurement tools force a group shape
  if (force >= 1 && force <= 3) return force - 1;
  int best = 0;
  double be = 0.0;
  for (int c = 0; c < 3; ++c) {
    const WwCfg& k = WW_CFGS[c];
    const double e = (double)Hq * Wq / ((double)roundup(Hq, k.gh) * roundup(Wq, k.gw));
    if (e > be + 0.03) { be = e; best = c; }
  }
  return best;
}
struct WwPlan { int Hq, Wq, cfg, gyn, gxn, ngroups, nchunks, gridM, gridC, splits, cps; };
static WwPlan ww_plan(const avsep_conv_desc* d) {
  WwPlan p{};
  const bool sub = d->dil == 2;
  p.Hq = sub ? (d->H + 1) / 2 : d->H;
  p.Wq = sub ? (d->W + 1) / 2 : d->W;
  p.cfg = sub ? 3 : ww_cfg(p.Hq, p.Wq, d->tune);
  const WwCfg& k = WW_CFGS[p.cfg];
  p.gyn = cdiv(p.Hq, k.gh);
  p.gxn = cdiv(p.Wq, k.gw);
  p.ngroups = d->N * (sub ? 2 : 1) * p.gyn * p.gxn;      // dilation 2: one staging group = a region of a ROW-parity class, both column parities
  p.nchunks = cdiv(p.ngroups, k.g);
  p.gridM = cdiv(d->Cout, WW_B);
  p.gridC = cdiv(d->Cin, WW_B);
  // one workgroup per CU (512 threads x ~230 registers) and no overlap between consecutive workgroups of a CU: ONE round
  // of workgroups is the fastest plan at every layer shape of the step (measured 256 / 384 / 512 / 768 / 1024 / 1536
  // workgroups: 186 / 144 / 186 / 182 / 180 / 173 TFLOP/s at 1024 -> 256 @ 32x32, 121 / 91 / 101 / 83 / 71 / 63 at
  // 64 -> 64 @ 56x56) — fewer epilogues and slabs, no partial last round
  const int tw = (d->tune >> 8) & 0xffff;                 // avsep_conv_desc.tune: measurement tools set the target workgroup count
  int want = (tw ? tw : cu_count()) / (p.gridM * p.gridC);
  const int maxs = p.nchunks / 8 > 0 ? p.nchunks / 8 : 1;    // at least 8 chunks per split
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  p.cps = cdiv(p.nchunks, want);
  if (p.cps * k.g > WW_GT_MAX) p.cps = WW_GT_MAX / k.g;
  p.splits = cdiv(p.nchunks, p.cps);
  return p;
}

bool ww_applicable(const avsep_conv_desc* d) {
  const bool off = (d->algo & (AVSEP_ALGO_NO_WINOGRAD | AVSEP_ALGO_NO_WINOGRAD_WGRAD)) != 0;
  if (off || d->prec != AVSEP_PREC_F32) return false;
  if (!(d->KH == 3 && d->KW == 3 && d->stride == 1 && (d->dil == 1 || d->dil == 2) && d->pad == d->dil) || d->up2x) return false;
  if ((d->H & 1) || (d->W & 1) || d->H < (d->dil == 1 ? 8 : 14) || d->W < (d->dil == 1 ? 8 : 14)) return false;
  const int C1 = d->Cin - d->C0;
  if (d->Cin % WW_B || d->C0 % WW_B || d->Cout < 48 || (C1 != 0 && C1 != d->C0)) return false;
  if (d->H >= 32768 || d->W >= 32768 || (long long)d->H * d->W >= (1 << 24)) return false;   // 24-bit offset arithmetic
  if ((long long)d->N * (d->C0 > d->Cout ? d->C0 : d->Cout) * d->H * d->W >= 0x3fffffffLL) return false;   // 32-bit BYTE offsets of the buffer loads
  const avsep_conv_desc e = plan_desc(d);
  const WwPlan p = ww_plan(&e);
  return (long long)p.gridM * p.gridC * p.splits >= 128 && p.nchunks >= 8;
}
size_t ww_workspace_floats(const avsep_conv_desc* d) {
  const WwPlan p = ww_plan(d);
  return (size_t)p.splits * 9 * d->Cout * d->Cin;
}

template <bool SUB, bool RAW>
static void ww_launch_cfg(const WwArgs& a, int cfg, dim3 grid, hipStream_t st) {
  if constexpr (SUB) {
    hipLaunchKernelGGL((winow_kernel<2, 2, 4, true, RAW>), grid, dim3(WW_THREADS), 0, st, a);
    return;
  }
  switch (cfg) {
    case 0: hipLaunchKernelGGL((winow_kernel<1, 2, 8, false, RAW>), grid, dim3(WW_THREADS), 0, st, a); break;
    case 1: hipLaunchKernelGGL((winow_kernel<1, 4, 4, false, RAW>), grid, dim3(WW_THREADS), 0, st, a); break;
    default: hipLaunchKernelGGL((winow_kernel<2, 1, 8, false, RAW>), grid, dim3(WW_THREADS), 0, st, a); break;
  }
}

int ww_wgrad(const avsep_conv_desc* d, const float* dy, float* dw, float* ws, hipStream_t st) {
  const WwPlan p = ww_plan(d);
  WwArgs a{};
  a.N = d->N; a.C0 = d->C0; a.C1 = d->Cin - d->C0; a.Cin = d->Cin; a.H = d->H; a.W = d->W; a.Cout = d->Cout;
  a.Hq = p.Hq; a.Wq = p.Wq; a.gyn = p.gyn; a.gxn = p.gxn; a.ngroups = p.ngroups; a.nchunks = p.nchunks; a.cps = p.cps;
  a.gridM = p.gridM; a.gridC = p.gridC; a.act0 = d->act0; a.act1 = d->act1;
  a.x0 = d->x0; a.x1 = d->x1; a.sc0 = d->scale0; a.sh0 = d->shift0; a.sc1 = d->scale1; a.sh1 = d->shift1;
  a.dy = dy; a.out = ws;
  const bool raw = !d->scale0 && !d->scale1 && d->act0 == AVSEP_ACT_NONE && (a.C1 == 0 || d->act1 == AVSEP_ACT_NONE);
  dim3 grid((unsigned)(p.gridM * p.gridC * p.splits));
  const bool sub = d->dil == 2;
  if (sub && raw) ww_launch_cfg<true, true>(a, p.cfg, grid, st);
  else if (sub) ww_launch_cfg<true, false>(a, p.cfg, grid, st);
  else if (raw) ww_launch_cfg<false, true>(a, p.cfg, grid, st);
  else ww_launch_cfg<false, false>(a, p.cfg, grid, st);
  AVSEP_LAUNCH_CHECK();
  return w3_reduce(ws, dw, (long long)d->Cout * d->Cin, p.splits, st);
}

// ---- 4x4 / stride 2 ----
// dw[cc][tap] = sum_z ws[z][tap][cc], cc = co*Cin + ci (16 taps)
__global__ void w4_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out, long long P, int S) {
  const long long cc = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (cc >= P) return;
  float s[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) s[t] = 0.f;
#pragma unroll 2
  for (int z = 0; z < S; ++z)
#pragma unroll
    for (int t = 0; t < 16; ++t) s[t] += ws[((long long)z * 16 + t) * P + cc];
#pragma unroll
  for (int t = 0; t < 4; ++t)
    reinterpret_cast<float4*>(out + cc * 16)[t] = float4{s[4 * t], s[4 * t + 1], s[4 * t + 2], s[4 * t + 3]};
}
static int w4_reduce(const float* ws, float* dw, long long P, int splits, hipStream_t st) {
  hipLaunchKernelGGL(w4_reduce_kernel, dim3(cdiv(P, 256)), dim3(256), 0, st, ws, dw, P, splits);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

static WwPlan w4d_plan(const avsep_conv_desc* d) {
  avsep_conv_desc e = *d;                       // the chunk geometry of the 3x3 plan on the INPUT grid (tile = output pixel)
  e.dil = 1;
  return ww_plan(&e);
}
bool w4d_applicable(const avsep_conv_desc* d) {
  const bool off = (d->algo & (AVSEP_ALGO_NO_WINOGRAD | AVSEP_ALGO_NO_WINOGRAD_WGRAD)) != 0;
  // (bf16 descriptors too: conv.hip asks the bf16 kernel first, and the maps it does not take — 8 / 4 wide at the deep
  // U-Net levels — are better off here in fp32 than on the im2col kernel: 0.31 against 0.60 ms at 512 -> 512 @ 16x16)
  if (off) return false;
  if (!(d->KH == 4 && d->KW == 4 && d->stride == 2 && d->pad == 1 && d->dil == 1) || d->up2x || d->C0 != d->Cin) return false;
  if ((d->H & 3) || (d->W & 3) || d->H < 8 || d->W < 8 || d->Ho * 2 != d->H || d->Wo * 2 != d->W) return false;
  if (d->Cin % WW_B || d->Cout < 48 || d->H >= 32768 || d->W >= 32768 || (long long)d->H * d->W >= (1 << 24)) return false;
  if ((long long)d->N * (d->Cin > d->Cout ? d->Cin : d->Cout) * d->H * d->W >= 0x3fffffffLL) return false;
  const avsep_conv_desc e = plan_desc(d);
  const WwPlan p = w4d_plan(&e);
  return p.cfg < 2 && (long long)p.gridM * p.gridC * p.splits >= 128 && p.nchunks >= 8;
}
size_t w4d_workspace_floats(const avsep_conv_desc* d) {
  const WwPlan p = w4d_plan(d);
  return (size_t)p.splits * 16 * d->Cout * d->Cin;
}
int w4d_wgrad(const avsep_conv_desc* d, const float* dy, float* dw, float* ws, hipStream_t st) {
  const WwPlan p = w4d_plan(d);
  W4dArgs a{};
  a.N = d->N; a.Cin = d->Cin; a.H = d->H; a.W = d->W; a.Cout = d->Cout; a.Ho = d->Ho; a.Wo = d->Wo;
  a.gyn = p.gyn; a.gxn = p.gxn; a.ngroups = p.ngroups; a.nchunks = p.nchunks; a.cps = p.cps;
  a.gridM = p.gridM; a.gridC = p.gridC; a.act0 = d->act0;
  a.x0 = d->x0; a.sc0 = d->scale0; a.sh0 = d->shift0; a.dy = dy; a.out = ws;
  const bool raw = !d->scale0 && d->act0 == AVSEP_ACT_NONE;
  dim3 grid((unsigned)(p.gridM * p.gridC * p.splits));
#define W4D_LAUNCH(G_, GH_, GW_)                                                                               \
  do {                                                                                                       \
    if (raw) hipLaunchKernelGGL((wgrad4d_kernel<G_, GH_, GW_, true>), grid, dim3(WW_THREADS), 0, st, a);      \
    else hipLaunchKernelGGL((wgrad4d_kernel<G_, GH_, GW_, false>), grid, dim3(WW_THREADS), 0, st, a);        \
  } while (0)
  if (p.cfg == 0) W4D_LAUNCH(1, 2, 8);
  else if (p.cfg == 1) W4D_LAUNCH(1, 4, 4);
  else return AVSEP_ERR_ARG;          // 2x16-pixel groups only win on heights that are not multiples of 4 (excluded above)
#undef W4D_LAUNCH
  AVSEP_LAUNCH_CHECK();
  return w4_reduce(ws, dw, (long long)d->Cout * d->Cin, p.splits, st);
}
