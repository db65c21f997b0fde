// 3x3 / stride 1 / pad 1 convolution in the Winograd F(4x4, 3x3) form on the f32 MFMA:
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A        per 4x4 output tile: 36 products where the direct form has 144
// i.e. 36 independent [Cout x Cin] x [Cin x tiles] GEMMs at 1/4 of the direct form's MFMA work (F(2x2,3x3) in
// conv_wino.hip: 16 products per 2x2 tile = 4/9).  These are the big 3x3 layers of the step whose maps tile by 4: the
// U-Net decoder convs u2-u5 (audio_net.py:75-76,85-87,96-98) and ResNet layer1 / layer2 (vision_net.py:96-109), forward
// and data gradient (the same conv over dY with flipped, transposed weights: only the weight transform differs).
//
// The transform of F(4x4) is too wide for the one-transform-row-per-wave scheme of conv_wino.hip (six rows on four SIMDs,
// and 36 accumulator positions x 64 channels x 32 tiles = 1152 registers per lane have to be spread evenly), so the input
// transform is DECOUPLED from the MFMA waves through LDS.  One 512-thread workgroup owns 64 output channels x 32 tiles
// (512 output pixels) and loops over K-tiles of 8 input channels; per K-tile and workgroup:
//   G  global -> registers: the raw halo patch of the tiles (position-major: a thread owns <= 2 patch positions and loads
//      them for the 8 channels: one lane offset per position, the channel is a scalar offset; outside the image the
//      offset is 0xffffffff, which loads as 0);
//   S  registers -> LDS patch (folded BatchNorm affine + activation + two-source concat applied on the way; padding
//      positions of an affine input are zeroed once and their stores go to a dead word);
//   T  LDS patch -> V = B^T d B -> LDS: thread (channel, tile, half) reads 5 rows x 6 columns of its tile's 6x6 window,
//      computes three of the six transform rows in 72 vector instructions (1D transforms factored as
//      a = d4 - 4 d2, b = d3 - 4 d1, a +- b, ...: 12 instead of 18 per 6 outputs) and writes 18 of the 36 positions
//      into V[xi][channel][tile];
//   M  36 MFMAs per wave: wave (cb, q) owns the 9 positions xi = 9q .. 9q+8 of channel block cb: 9 accumulator tiles
//      (144 registers), A fragments (the transformed weights U) straight from global memory in MFMA lane order — each
//      (xi, cb) fragment is used by exactly one wave, so staging them in LDS would only add traffic — and B fragments
//      as conflict-free ds_read_b32 of V.
// The four stages of consecutive K-tiles overlap (M(i) | T(i+1) | S(i+2) | G(i+3)) with ONE barrier per K-tile; every
// non-MFMA instruction is hung behind an MFMA (the f32 MFMA does not overlap vector instructions, but LDS and memory
// instructions issue in its shadow).  The barrier waits for LDS only (s_waitcnt lgkmcnt(0); s_barrier): the global loads
// in flight are register loads.
// Epilogue: Y = A^T M A.  The 36 positions of an output tile sit in four waves, so M goes through LDS in four passes
// (channel block x accumulator-row half: 36 x 16 x 32 floats = the two V buffers), then every thread transforms one
// (channel, tile): 100 vector instructions, four 16-byte row stores, BatchNorm sums.  The data gradient can be taken through
// the activation in front of the conv's input there (EPI instantiations, W4Epi below: avsep_conv2d_dgrad_act).
#include <stdlib.h>
#include <type_traits>

#include "common.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int W4_CK = 8;              // input channels per K-tile (host-side packing constant)
constexpr int W4_BM = 64;             // output channels per workgroup
constexpr int W4_NT = 32;             // 4x4 output tiles per workgroup
constexpr int W4_THREADS = 512;
constexpr int W4_AFF_MAX = 2048;      // channels of the folded affine rows kept in LDS (host-checked)
constexpr int W4_U_FLOATS = W4_CK * W4_BM * 36;     // transformed weights of one (K-tile, M-tile): 73,728 bytes
constexpr int W4_V_FLOATS = 36 * W4_CK * W4_NT;     // transformed input of one K-tile

// Data gradient through the activation of the tensor the convolution read (avsep_conv2d_dgrad_act): the epilogue is
// avsep_affine_act_bwd on the values it holds in registers,
//     g = act'(sc*y + sh [+ rs*res + rh]) * (dx [+ dz2]) [+ add],   stats += (sum g, sum g * (y - mean) * invstd),
// with y, res, dz2, add laid out like dx.  y == nullptr: the plain data gradient.
struct W4Epi {
  const float *y, *sc, *sh, *res, *rs, *rh, *dz2, *add, *mean, *inv;
  float slope;
};

struct W4Args {
  int N, C0, C1, Cin, H, W, Cout;   // Cin = C0 + C1
  int Hq, Wq;                       // the (sub-)image the tiles live in: H x W, or H/2 x W/2 per parity class (dilation 2)
  int gyn, gxn, ngroups;            // tile groups per (sub-)image along y / x, and in total
  int gridM, act0, act1;
  const float *x0, *x1, *sc0, *sh0, *sc1, *sh1;
  const float* up;                  // [K-tile][M-tile][cb][q][chunk 9][lk 2][li 32][4]
  float* out;
  const float* bias;
  double* stats;                    // forward: (sum y, sum y^2) per output channel; with e.y: the sums of the line above
  W4Epi e;
};

// U = G g G^T (6x6) of every (input channel, output channel) pair, in the order the MFMA waves load it: wave (cb, q),
// chunk c, lane (lk, li) reads the four values idx = 4c .. 4c+3 of its K-tile with idx = s*9 + e:
//   k-step s (input channel kt*8 + 2s + lk), position xi = 9q + e, output channel mt*64 + cb*32 + li.
//   mode 0 (forward): g = w[co][ci][.][.]                         mode 1 (dgrad): g = w[ch][col][2-kh][2-kw] ("in" = co)
__global__ void wino4_pack_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin, int gridM, int nK,
                                  int mode) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)nK * gridM * (W4_CK * W4_BM)) return;
  const int col = (int)(i % W4_BM), chl = (int)((i / W4_BM) % W4_CK);
  const int mt = (int)((i / (W4_CK * W4_BM)) % gridM), kt = (int)(i / ((long long)W4_CK * W4_BM * gridM));
  const int ch = kt * W4_CK + chl, co = mt * W4_BM + col;
  double g[9];
  const bool ok = mode == 0 ? (ch < Cin && co < Cout) : (ch < Cout && co < Cin);
#pragma unroll
  for (int k = 0; k < 9; ++k)
    g[k] = !ok ? 0.0 : (double)(mode == 0 ? w[((long long)co * Cin + ch) * 9 + k] : w[((long long)ch * Cin + co) * 9 + (8 - k)]);
  // G = [[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]]; in double, rounded once
  double t[6][3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const double g0 = g[c], g1 = g[3 + c], g2 = g[6 + c];
    t[0][c] = g0 / 4.0;
    t[1][c] = -(g0 + g1 + g2) / 6.0;
    t[2][c] = -(g0 - g1 + g2) / 6.0;
    t[3][c] = g0 / 24.0 + g1 / 12.0 + g2 / 6.0;
    t[4][c] = g0 / 24.0 - g1 / 12.0 + g2 / 6.0;
    t[5][c] = g2;
  }
  const int s = chl >> 1, lk = chl & 1, cb = col >> 5, li = col & 31;
  float* o = out + ((long long)kt * gridM + mt) * W4_U_FLOATS;
#pragma unroll
  for (int ti = 0; ti < 6; ++ti) {
    const double g0 = t[ti][0], g1 = t[ti][1], g2 = t[ti][2];
    const double u[6] = {g0 / 4.0, -(g0 + g1 + g2) / 6.0, -(g0 - g1 + g2) / 6.0, g0 / 24.0 + g1 / 12.0 + g2 / 6.0,
                         g0 / 24.0 - g1 / 12.0 + g2 / 6.0, g2};
#pragma unroll
    for (int tj = 0; tj < 6; ++tj) {
      const int xi = ti * 6 + tj, q = xi / 9, e = xi % 9, idx = s * 9 + e, c = idx >> 2, jj = idx & 3;
      o[((((cb * 4 + q) * 9 + c) * 2 + lk) * 32 + li) * 4 + jj] = (float)u[tj];
    }
  }
}

// The six outputs of the 1D input transform B^T x (x = six values): 12 vector instructions.
#define W4_BT6(X0, X1, X2, X3, X4, X5, O0, O1, O2, O3, O4, O5)              \
  do {                                                                      \
    const float a__ = fmaf(-4.f, X2, X4), b__ = fmaf(-4.f, X1, X3);         \
    const float c__ = X4 - X2, e__ = X3 - X1;                               \
    O0 = fmaf(4.f, X0, fmaf(-5.f, X2, X4));                                 \
    O1 = a__ + b__;                                                         \
    O2 = a__ - b__;                                                         \
    O3 = fmaf(2.f, e__, c__);                                               \
    O4 = fmaf(-2.f, e__, c__);                                              \
    O5 = fmaf(4.f, X1, fmaf(-5.f, X3, X5));                                 \
  } while (0)
// The four outputs of the 1D output transform A^T m (m = six values): 10 vector instructions.
#define W4_AT4(M0, M1, M2, M3, M4, M5, O0, O1, O2, O3)                      \
  do {                                                                      \
    const float s1__ = M1 + M2, d1__ = M1 - M2, s3__ = M3 + M4, d3__ = M3 - M4; \
    O0 = M0 + s1__ + s3__;                                                  \
    O1 = fmaf(2.f, d3__, d1__);                                             \
    O2 = fmaf(4.f, s3__, s1__);                                             \
    O3 = fmaf(8.f, d3__, d1__) + M5;                                        \
  } while (0)

// G groups of GH x GW tiles (G*GH*GW = 32); PW = LDS row stride of a group's patch (multiple of 4: the transform reads
// rows as b128 + b64), GS = LDS stride between groups; RAW: no affine and no activation on the staged tensor; SUB: the
// groups tile the four parity sub-images of a dilation-2 conv (a dilated 'same' conv is four independent undilated convs
// over the pixels of equal row / column parity); FULL: H and W are multiples of 4 (whole tiles, 16-byte row stores) —
// otherwise the last tile row / column of a (sub-)image is partial: its inputs load as zeros, its outputs are masked.
// GX: extra words in front of every second group and twice as many in front of every fourth (g*GS + GX*(g>>1) + 2*GX*(g>>2)):
// with eight 2x2-tile groups no uniform group stride spreads the row reads of a 16-lane b128 access over all 64 banks (4-way
// conflicts measured with a uniform stride of 128 words).
// EPI: the instantiations that carry the activation-gradient epilogue (W4Epi; data gradients only, so RAW)
template <int G, int GH, int GW, int PW, int GS, int GX, bool RAW, bool SUB, bool FULL, bool EPI = false>
__global__ __launch_bounds__(W4_THREADS) void wino4_kernel(W4Args a) {
  static_assert(!(SUB && FULL), "sub-image stores are strided");
  auto goff = [](int g) __attribute__((always_inline)) { return g * GS + GX * (g >> 1) + 2 * GX * (g >> 2); };
  static_assert(G * GH * GW == W4_NT, "32 tiles per workgroup");
  constexpr int CK = W4_CK, BM = W4_BM, NT = W4_THREADS;
  constexpr int PHG = 4 * GH + 2, PCG = 4 * GW + 2;                 // patch of one group (valid elements)
  static_assert(PW % 4 == 0 && PW > PCG && GS % 4 == 0 && GX % 4 == 0 && GS >= PHG * PW, "patch strides");
  constexpr int PS = G * GS + GX * ((G - 1) >> 1) + 2 * GX * ((G - 1) >> 2);   // floats per channel (= goff(G - 1) + GS)
  constexpr int P_FLOATS = CK * PS, V_FLOATS = W4_V_FLOATS;
  constexpr int NPOS = G * PHG * PCG, NSLOT = (NPOS + NT - 1) / NT;  // patch positions, positions per thread
  constexpr int DEAD = PCG;                                          // a word of row 0 that no transform reads (PW > PCG)
  static_assert(2 * V_FLOATS >= 36 * 16 * 32, "epilogue exchange fits the V buffers");
  static_assert(2 * P_FLOATS >= 2 * BM, "statistics partials fit the patch buffers");
  __shared__ __attribute__((aligned(16))) float smem[2 * P_FLOATS + 2 * V_FLOATS + (RAW ? 0 : 2 * W4_AFF_MAX)];
  float* const Pb = smem;
  float* const Vb = smem + 2 * P_FLOATS;
  float* const aff = Vb + 2 * V_FLOATS;          // [channel][scale, shift]

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 31, lk = lane >> 5;
  const int wcb = wave & 1, wq = wave >> 1;      // MFMA role: channel block, position group
  const int thalf = wave & 1, tp = wave >> 1;    // transform role: half (rows 0-2 / 3-5), channel pair
  int t = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = t % a.gridM, pt = t / a.gridM, m0 = mt * BM;
  const int per = a.gyn * a.gxn;
  const long long HW = (long long)a.H * a.W;

  // ---- group table: image / origin of the G tile groups of this workgroup ----
  __shared__ int gtab[G][4];                     // {image, y0, x0, valid}
  if (tid < G) {
    const int gid = pt * G + tid, gidc = min(gid, a.ngroups - 1);
    const int img = gidc / per, gy = (gidc % per) / a.gxn, gx = gidc % a.gxn;
    gtab[tid][0] = img;
    gtab[tid][1] = gy * 4 * GH;
    gtab[tid][2] = gx * 4 * GW;
    gtab[tid][3] = gid < a.ngroups;
  }
  if constexpr (!RAW) {                          // folded BatchNorm rows of both sources -> LDS once (identity where absent)
    for (int c = tid; c < a.Cin; c += NT) {
      const bool s0 = c < a.C0;
      const float* sc = s0 ? a.sc0 : a.sc1;
      const float* sh = s0 ? a.sh0 : a.sh1;
      const int cs = s0 ? c : c - a.C0;
      aff[2 * c] = sc ? sc[cs] : 1.f;
      aff[2 * c + 1] = sc ? sh[cs] : 0.f;
    }
  }
  __syncthreads();

  // ---- patch loader state: per owned position one byte offset (channel 0 of its image) and one LDS word ----
  unsigned p_off[NSLOT];
  int p_lds[NSLOT];
#pragma unroll
  for (int sl = 0; sl < NSLOT; ++sl) {
    const int pos = tid + NT * sl, posc = min(pos, NPOS - 1);
    const int g = posc / (PHG * PCG), r = (posc % (PHG * PCG)) / PCG, col = posc % PCG;
    const int img = gtab[g][0], y = gtab[g][1] - 1 + r, x = gtab[g][2] - 1 + col;
    const bool valid = pos < NPOS;
    const bool inimg = valid && gtab[g][3] && (unsigned)y < (unsigned)a.Hq && (unsigned)x < (unsigned)a.Wq;
    int n = img, fy = y, fx = x;
    if constexpr (SUB) {
      n = img >> 2;
      fy = 2 * y + ((img >> 1) & 1);
      fx = 2 * x + (img & 1);
    }
    // BYTE offset from the source's base (the channel offset is scalar): element offsets < 2^30, host check
    p_off[sl] = inimg ? 4u * (unsigned)((long long)n * a.C0 * HW + (long long)fy * a.W + fx) : 0xffffffffu;   // C1 == C0 when there is a source 1
    const int word = goff(g) + r * PW + col;
    if constexpr (RAW) {
      p_lds[sl] = valid ? word : DEAD;           // an element outside the image loads as 0: stored like any other
    } else {
      p_lds[sl] = inimg ? word : DEAD;           // act(affine(0)) != 0: padding is zeroed here, once, for both buffers
      if (valid && !inimg) {
#pragma unroll
        for (int cc = 0; cc < CK; ++cc) {
          Pb[cc * PS + word] = 0.f;
          Pb[P_FLOATS + cc * PS + word] = 0.f;
        }
      }
    }
  }
  const int kt_switch = a.C1 > 0 ? a.C0 / CK : 0x7fffffff;      // first K-tile of source 1
  const float slope0 = act_slope(a.act0), slope1 = act_slope(a.act1);
  const int nK = a.Cin / CK;

  const __amdgpu_buffer_rsrc_t rs_x0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.x0, 0, 0xfffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x1 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x1 ? a.x1 : a.x0), 0, 0xfffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_up = __builtin_amdgcn_make_buffer_rsrc((void*)a.up, 0, 0xfffffff0, 0x00020000);

  // ---- stage G / S: NSLOT * CK pieces per K-tile ----
  constexpr int NPC = NSLOT * CK;
  float praw[NPC];
  auto g_issue_to = [&](float* dst, int kt_, int pc) __attribute__((always_inline)) {
    const int kt = __builtin_amdgcn_readfirstlane(kt_);
    const int sl = pc / CK, cc = pc % CK;
    const bool src1 = kt >= kt_switch;
    const unsigned soff = (unsigned)((src1 ? kt - kt_switch : kt) * CK + cc) * 4u * (unsigned)HW;
    dst[pc] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(src1 ? rs_x1 : rs_x0, (int)p_off[sl], (int)soff, 0));
  };
  auto g_issue = [&](int kt_, int pc) __attribute__((always_inline)) { g_issue_to(praw, kt_, pc); };
  auto s_store_from = [&](const float* src, int kt, int buf, int pc) __attribute__((always_inline)) {
    const int sl = pc / CK, cc = pc % CK;
    float v = src[pc];
    if constexpr (!RAW) {
      const f32x2 sa = *reinterpret_cast<const f32x2*>(aff + 2 * (kt * CK + cc));
      v = act_by_slope(fmaf(v, sa[0], sa[1]), kt >= kt_switch ? slope1 : slope0);
    }
    (Pb + buf * P_FLOATS + cc * PS)[p_lds[sl]] = v;
  };
  auto s_store = [&](int kt, int buf, int pc) __attribute__((always_inline)) { s_store_from(praw, kt, buf, pc); };

  // ---- stage T: thread (half = wave & 1, channel = 2 * (wave >> 1) + lk, tile = li) ----
  int t_src, t_dst;
  {
    const int g = li / (GH * GW), ty = (li % (GH * GW)) / GW, tx = li % GW;
    t_src = (2 * tp + lk) * PS + goff(g) + (4 * ty + thalf) * PW + 4 * tx;
    t_dst = (18 * thalf * CK + 2 * tp + lk) * W4_NT + li;               // V[(18 half + m)][channel][tile], m * 256 apart
  }
  f32x4 ra[5];
  f32x2 rb[5];
  float tt[3][6];
  auto t_read = [&](int buf, int k) __attribute__((always_inline)) {    // k = 0..9: row k/2, columns 0-3 / 4-5
    const float* pp = Pb + buf * P_FLOATS + t_src + (k >> 1) * PW;
    if (k & 1) rb[k >> 1] = *reinterpret_cast<const f32x2*>(pp + 4);
    else ra[k >> 1] = *reinterpret_cast<const f32x4*>(pp);
  };
  auto t_stage1 = [&](auto half_, int c) __attribute__((always_inline)) {    // column c of the three rows: 6 instructions
    constexpr int HALF = decltype(half_)::value;
    float R[5];
#pragma unroll
    for (int r = 0; r < 5; ++r) R[r] = c < 4 ? ra[r][c & 3] : rb[r][c & 1];
    if constexpr (HALF == 0) {        // rows 0-2 of B^T from d0..d4
      const float aa = fmaf(-4.f, R[2], R[4]), bb = fmaf(-4.f, R[1], R[3]);
      tt[0][c] = fmaf(4.f, R[0], fmaf(-5.f, R[2], R[4]));
      tt[1][c] = aa + bb;
      tt[2][c] = aa - bb;
    } else {                          // rows 3-5 from d1..d5 (R[r] = d[r+1])
      const float cc_ = R[3] - R[1], ee = R[2] - R[0];
      tt[0][c] = fmaf(2.f, ee, cc_);
      tt[1][c] = fmaf(-2.f, ee, cc_);
      tt[2][c] = fmaf(4.f, R[0], fmaf(-5.f, R[2], R[4]));
    }
  };
  auto t_stage2 = [&](int buf, int i) __attribute__((always_inline)) {  // row i: 12 instructions + 6 stores
    float v0, v1, v2, v3, v4, v5;
    W4_BT6(tt[i][0], tt[i][1], tt[i][2], tt[i][3], tt[i][4], tt[i][5], v0, v1, v2, v3, v4, v5);
    float* vp = Vb + buf * V_FLOATS + t_dst + i * 6 * (CK * W4_NT);
    vp[0 * CK * W4_NT] = v0;
    vp[1 * CK * W4_NT] = v1;
    vp[2 * CK * W4_NT] = v2;
    vp[3 * CK * W4_NT] = v3;
    vp[4 * CK * W4_NT] = v4;
    vp[5 * CK * W4_NT] = v5;
  };

  // ---- stage M operands ----
  f32x4 areg[9];                                  // idx = s * 9 + e -> areg[idx >> 2][idx & 3]
  const unsigned a_lane = (unsigned)(((wcb * 4 + wq) * 9 * 2 + lk) * 32 + li) * 16u;      // byte offset of chunk 0
  auto a_issue = [&](int kt_, int c) __attribute__((always_inline)) {
    const int kt = __builtin_amdgcn_readfirstlane(kt_);
    const unsigned soff = (unsigned)(kt * a.gridM + mt) * (unsigned)(W4_U_FLOATS * 4);    // < 2^32: host check
    areg[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_up, (int)(a_lane + (unsigned)c * 1024u), (int)soff, 0));
  };
  const int b_lane = (9 * wq * CK + lk) * W4_NT + li;                  // V[9q + e][2s + lk][li]
  f32x16 acc[9];
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  auto lds_barrier = [&]() __attribute__((always_inline)) {
    // LDS only: the global loads in flight target registers and must stay in flight (a __syncthreads() would drain them)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };

  // One K-tile: M(kt) on buffer `buf`, T(kt+1) into the other V buffer, S(kt+2) + G(kt+3) on the patch buffer `buf`.
  // Slot m (behind MFMA m = s * 9 + e):   B read of MFMA m + RB;   every 4th slot the A chunk that just retired, for kt+1;
  //   slots 0..NPC-1: S piece, then G piece into the same register;   T: reads in slots 0-9, stage 1 in 10-21 (one column per
  //   two slots), stage 2 in 22-33 (one row per four slots).
  constexpr int RB = 4;
  auto ktile = [&](int kt, auto buf_, auto first_, auto half_) __attribute__((always_inline)) {
    constexpr int buf = decltype(buf_)::value;
    constexpr bool FIRST = decltype(first_)::value;
    const int kt1 = min(kt + 1, nK - 1), kt2 = min(kt + 2, nK - 1), kt3 = min(kt + 3, nK - 1);
    const float* Vr = Vb + buf * V_FLOATS + b_lane;
    float bv[36];
#pragma unroll
    for (int m = 0; m < RB; ++m) bv[m] = Vr[((m % 9) * CK + 2 * (m / 9)) * W4_NT];
#pragma unroll
    for (int m = 0; m < 36; ++m) {
      const int s = m / 9, e = m % 9;
      const float av = areg[m >> 2][m & 3];
      if (FIRST && s == 0) acc[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[m], zero16, 0, 0, 0);
      else acc[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[m], acc[e], 0, 0, 0);
      if (m + RB < 36) bv[m + RB] = Vr[(((m + RB) % 9) * CK + 2 * ((m + RB) / 9)) * W4_NT];
      if ((m & 3) == 3) a_issue(kt1, m >> 2);
      if (m < NPC) {
        s_store(kt2, buf, m);
        g_issue(kt3, m);
      }
      if (m < 10) t_read(buf ^ 1, m);
      if (m >= 10 && m < 22 && ((m - 10) & 1) == 0) t_stage1(half_, (m - 10) >> 1);
      if (m >= 22 && m < 34 && ((m - 22) & 3) == 0) t_stage2(buf ^ 1, (m - 22) >> 2);
      __builtin_amdgcn_sched_barrier(0);
    }
    lds_barrier();
  };
  static_assert(NPC <= 36, "one S / G piece per MFMA slot");

  auto run = [&](auto half_) __attribute__((always_inline)) {
    // prologue: G(0) G(1) S(0) S(1) | T(0) | then the loop's first trip needs V(0), P(1), praw = G(2), areg = A(0).
    // The patches of K-tiles 0 and 1 are loaded together (the second set of registers is free: no accumulator lives yet):
    // one memory round trip less in front of the first MFMA.
    {
      float praw1[NPC];
#pragma unroll
      for (int pc = 0; pc < NPC; ++pc) g_issue_to(praw, 0, pc);
#pragma unroll
      for (int pc = 0; pc < NPC; ++pc) g_issue_to(praw1, min(1, nK - 1), pc);
#pragma unroll
      for (int c = 0; c < 9; ++c) a_issue(0, c);
#pragma unroll
      for (int pc = 0; pc < NPC; ++pc) s_store_from(praw, 0, 0, pc);
#pragma unroll
      for (int pc = 0; pc < NPC; ++pc) g_issue_to(praw, min(2, nK - 1), pc);
#pragma unroll
      for (int pc = 0; pc < NPC; ++pc) s_store_from(praw1, min(1, nK - 1), 1, pc);
    }
    lds_barrier();
#pragma unroll
    for (int k = 0; k < 10; ++k) t_read(0, k);
#pragma unroll
    for (int c = 0; c < 6; ++c) t_stage1(half_, c);
#pragma unroll
    for (int i = 0; i < 3; ++i) t_stage2(0, i);
    lds_barrier();
    ktile(0, std::integral_constant<int, 0>{}, std::true_type{}, half_);
    if (nK > 1) ktile(1, std::integral_constant<int, 1>{}, std::false_type{}, half_);
    for (int kt = 2; kt < nK; kt += 2) {
      ktile(kt, std::integral_constant<int, 0>{}, std::false_type{}, half_);
      if (kt + 1 < nK) ktile(kt + 1, std::integral_constant<int, 1>{}, std::false_type{}, half_);
    }
  };
  if (thalf) run(std::integral_constant<int, 1>{});
  else run(std::integral_constant<int, 0>{});
  __syncthreads();          // everything staged past the last K-tile has landed: LDS is free for the epilogue

  // ---- epilogue: Y = A^T M A, four passes (channel block cbp, accumulator rows 8h .. 8h+7 = 16 channels) ----
  //      C/D map of an accumulator tile: tile = lane & 31, channel row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
  float* const Zs = Vb;                      // [xi 36][channel 16][tile 32]
  float* const s_sum = Pb;                   // [BM]
  float* const s_sq = Pb + BM;
  const int ec = tid >> 5, etile = tid & 31; // output role: channel within the pass, tile
  const int eg = etile / (GH * GW), ety = (etile % (GH * GW)) / GW, etx = etile % GW;
  const int img = gtab[eg][0], oy = gtab[eg][1] + 4 * ety, ox = gtab[eg][2] + 4 * etx;
  const bool tok = gtab[eg][3] && oy < a.Hq && ox < a.Wq;        // FULL: a started tile is whole
  long long obase;
  if constexpr (SUB) obase = (long long)(img >> 2) * a.Cout * HW + (long long)(2 * oy + ((img >> 1) & 1)) * a.W + 2 * ox + (img & 1);
  else obase = (long long)img * a.Cout * HW + (long long)oy * a.W + ox;
  constexpr int XS = SUB ? 2 : 1;
  const bool want_stats = a.stats != nullptr;
  constexpr bool epi = EPI;
#pragma unroll
  for (int cbp = 0; cbp < 2; ++cbp) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      // (EPI) y at this thread's 4x4 outputs: in flight across the exchange below (the other operands are loaded where they
      // are used: 16 more registers each, and the accumulators are still live)
      f32x4 pyv[4];
      if constexpr (EPI) {
        const int prow = m0 + cbp * 32 + 16 * h + ec;
        const float* py = a.e.y + obase + (long long)prow * HW;
        const bool pok = prow < a.Cout && tok;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if constexpr (FULL) {
            pyv[i] = pok ? *reinterpret_cast<const f32x4*>(py + (long long)i * a.W) : f32x4{0.f, 0.f, 0.f, 0.f};
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              pyv[i][j] = (pok && oy + i < a.Hq && ox + j < a.Wq) ? py[(long long)(XS * i) * a.W + XS * j] : 0.f;
          }
        }
      }
      if (wcb == cbp) {
#pragma unroll
        for (int e = 0; e < 9; ++e)
#pragma unroll
          for (int rr = 0; rr < 8; ++rr)
            Zs[((9 * wq + e) * 16 + (rr & 3) + 8 * (rr >> 2) + 4 * lk) * 32 + li] = acc[e][8 * h + rr];
      }
      lds_barrier();           // LDS only: the 16-byte stores of the previous pass stay in flight (a __syncthreads() waits for them)
      {
        const int lrow = cbp * 32 + 16 * h + ec, row = m0 + lrow;
        const bool rok = row < a.Cout;
        const float* z = Zs + ec * 32 + etile;
        float r_[4][6];                      // A^T M: rows
#pragma unroll
        for (int j = 0; j < 6; ++j)
          W4_AT4(z[(0 * 6 + j) * 512], z[(1 * 6 + j) * 512], z[(2 * 6 + j) * 512], z[(3 * 6 + j) * 512], z[(4 * 6 + j) * 512],
                 z[(5 * 6 + j) * 512], r_[0][j], r_[1][j], r_[2][j], r_[3][j]);
        const float bias = (a.bias && rok) ? a.bias[row] : 0.f;
        const long long ooff = obase + (long long)row * HW;
        float* o = a.out + ooff;
        float s = 0.f, q = 0.f;
        float esc = 1.f, esh = 0.f, ers = 1.f, erh = 0.f, emu = 0.f, eis = 1.f;
        if (epi && rok) {
          if (a.e.sc) { esc = a.e.sc[row]; esh = a.e.sh[row]; }
          if (a.e.rs) { ers = a.e.rs[row]; erh = a.e.rh[row]; }
          if (a.e.mean) { emu = a.e.mean[row]; eis = a.e.inv[row]; }
        }
        // one element through the activation gradient; returns g and adds to the two sums
        auto through = [&](float v, float yk, float rk, float d2, float ak) __attribute__((always_inline)) {
          float pre = fmaf(yk, esc, esh);
          if (a.e.res) pre += fmaf(rk, ers, erh);
          float g = (pre > 0.f ? 1.f : a.e.slope) * (a.e.dz2 ? v + d2 : v);
          if (a.e.add) g += ak;
          s += g;
          q += g * (yk - emu) * eis;
          return g;
        };
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          f32x4 y;
          W4_AT4(r_[i][0], r_[i][1], r_[i][2], r_[i][3], r_[i][4], r_[i][5], y[0], y[1], y[2], y[3]);
          y += bias;
          if constexpr (FULL) {
            if (rok && tok) {
              if (epi) {
                const long long eo = ooff + (long long)i * a.W;
                const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
                const f32x4 yv = pyv[i];
                const f32x4 rv = a.e.res ? *reinterpret_cast<const f32x4*>(a.e.res + eo) : zero4;
                const f32x4 dv = a.e.dz2 ? *reinterpret_cast<const f32x4*>(a.e.dz2 + eo) : zero4;
                const f32x4 av = a.e.add ? *reinterpret_cast<const f32x4*>(a.e.add + eo) : zero4;
#pragma unroll
                for (int j = 0; j < 4; ++j) y[j] = through(y[j], yv[j], rv[j], dv[j], av[j]);
                *reinterpret_cast<f32x4*>(o + (long long)i * a.W) = y;
              } else {
                *reinterpret_cast<f32x4*>(o + (long long)i * a.W) = y;
                s += (y[0] + y[1]) + (y[2] + y[3]);
                q += (y[0] * y[0] + y[1] * y[1]) + (y[2] * y[2] + y[3] * y[3]);
              }
            }
          } else {
            if (rok && tok && oy + i < a.Hq) {
              float* orow = o + (long long)(XS * i) * a.W;
#pragma unroll
              for (int j = 0; j < 4; ++j)
                if (ox + j < a.Wq) {
                  if (epi) {
                    const long long eo = ooff + (long long)(XS * i) * a.W + XS * j;
                    orow[XS * j] = through(y[j], pyv[i][j], a.e.res ? a.e.res[eo] : 0.f, a.e.dz2 ? a.e.dz2[eo] : 0.f,
                                           a.e.add ? a.e.add[eo] : 0.f);
                  } else {
                    orow[XS * j] = y[j];
                    s += y[j];
                    q += y[j] * y[j];
                  }
                }
            }
          }
        }
        if (want_stats) {
          s = half_sum_hi(s);
          q = half_sum_hi(q);
          if (li == 31) {
            s_sum[lrow] = s;
            s_sq[lrow] = q;
          }
        }
      }
      lds_barrier();
    }
  }
  if (want_stats) {
    for (int rr = tid; rr < BM; rr += NT) {
      const int row = m0 + rr;
      if (row < a.Cout) {
        atomicAdd(&a.stats[row], (double)s_sum[rr]);
        atomicAdd(&a.stats[a.Cout + row], (double)s_sq[rr]);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// host side (called from conv.hip)
// ---------------------------------------------------------------------------
struct W4Cfg { int gh, gw, g; };      // pixels per group, groups per workgroup
static const W4Cfg W4_CFGS[3] = {{16, 32, 1}, {16, 16, 2}, {8, 8, 8}};

// the group shape that wastes the fewest tiles on an H x W map (ties: the larger group, i.e. the smaller halo)
static int w4_cfg(int H, int W) {
  int best = 0;
  double be = 0.0;
  for (int c = 0; c < 3; ++c) {
    const W4Cfg& k = W4_CFGS[c];
    const double e = (double)H * W / ((double)roundup(H, k.gh) * roundup(W, k.gw));
    if (e > be + 0.03) { be = e; best = c; }
  }
  return best;
}

struct W4Plan { int Hq, Wq, cfg, gyn, gxn, ngroups, ptiles, gridM; };
static W4Plan w4_plan(const avsep_conv_desc* d, int mode) {
  W4Plan p{};
  const bool sub = d->dil == 2;
  p.Hq = sub ? d->H / 2 : d->H;
  p.Wq = sub ? d->W / 2 : d->W;
  p.cfg = w4_cfg(p.Hq, p.Wq);
  const W4Cfg& k = W4_CFGS[p.cfg];
  p.gyn = cdiv(p.Hq, k.gh);
  p.gxn = cdiv(p.Wq, k.gw);
  p.ngroups = d->N * (sub ? 4 : 1) * p.gyn * p.gxn;
  p.ptiles = cdiv(p.ngroups, k.g);
  p.gridM = cdiv(mode == 0 ? d->Cout : d->Cin, W4_BM);
  return p;
}

bool w4_applicable(const avsep_conv_desc* d, int mode) {
  if ((d->algo & (AVSEP_ALGO_NO_WINOGRAD | AVSEP_ALGO_NO_WINOGRAD4)) || d->prec != AVSEP_PREC_F32) return false;
  if (!(d->KH == 3 && d->KW == 3 && d->stride == 1 && (d->dil == 1 || d->dil == 2) && d->pad == d->dil) || d->up2x) return false;
  if ((d->H & 1) || (d->W & 1) || d->H < 14 || d->W < 14) return false;          // even maps (dilation 2: equal parity sub-images)
  {   // tile fill: F(4x4) pays 36 products per 4x4 tile, F(2x2) 16 per 2x2 — worth it from ~60 % of whole tiles
    const int hq = d->dil == 2 ? d->H / 2 : d->H, wq = d->dil == 2 ? d->W / 2 : d->W;
    if ((double)hq * wq < 0.6 * (double)roundup(hq, 4) * roundup(wq, 4)) return false;
  }
  const int cin = mode == 0 ? d->Cin : d->Cout, cout = mode == 0 ? d->Cout : d->Cin;
  if (cin % W4_CK || cin < 32 || cin > W4_AFF_MAX || cout < 48) return false;
  if (mode == 0) {
    const int C1 = d->Cin - d->C0;
    if (d->C0 % W4_CK || (C1 != 0 && C1 != d->C0)) return false;
  }
  if ((long long)d->N * (mode == 0 ? d->C0 : d->Cout) * d->H * d->W >= 0x3fffffffLL) return false;   // 32-bit BYTE offsets
  if ((long long)(cin / W4_CK) * cdiv(cout, W4_BM) * W4_U_FLOATS * 4 >= 0xffffffffLL) return false;
  const avsep_conv_desc e = plan_desc(d);
  const W4Plan p = w4_plan(&e, mode);
  return (long long)p.ptiles * p.gridM >= 192;      // at least 3/4 of the CUs busy (below that F(2x2) with its 256-pixel tiles fills better)
}
size_t w4_packed_floats(const avsep_conv_desc* d, int mode) {
  const int cin = mode == 0 ? d->Cin : d->Cout, cout = mode == 0 ? d->Cout : d->Cin;
  return (size_t)(cin / W4_CK) * cdiv(cout, W4_BM) * W4_U_FLOATS;
}
int w4_pack(const avsep_conv_desc* d, const float* w, float* packed, int mode, hipStream_t st) {
  const int cin = mode == 0 ? d->Cin : d->Cout, cout = mode == 0 ? d->Cout : d->Cin;
  const int nK = cin / W4_CK, gridM = cdiv(cout, W4_BM);
  const long long total = (long long)nK * gridM * (W4_CK * W4_BM);
  hipLaunchKernelGGL(wino4_pack_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, w, packed, d->Cout, d->Cin, gridM, nK, mode);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}
void w4_variant(const avsep_conv_desc* d, int mode, char* buf, size_t cap) {
  const W4Cfg& k = W4_CFGS[w4_plan(d, mode).cfg];
  snprintf(buf, cap, "%dx%dx%d%s", k.g, k.gh, k.gw, d->dil == 2 ? ",sub" : "");
}

template <bool RAW, bool SUB, bool FULL, bool EPI = false>
static void w4_launch_cfg(const W4Args& a, int cfg, dim3 grid, hipStream_t st) {
  // <G, GH, GW, PW, GS>: row strides chosen so that the b128 row reads of the transform spread over the 64 banks
  switch (cfg) {
    case 0: hipLaunchKernelGGL((wino4_kernel<1, 4, 8, 40, 18 * 40, 0, RAW, SUB, FULL, EPI>), grid, dim3(W4_THREADS), 0, st, a); break;
    case 1: hipLaunchKernelGGL((wino4_kernel<2, 4, 4, 20, 384, 0, RAW, SUB, FULL, EPI>), grid, dim3(W4_THREADS), 0, st, a); break;
    default: hipLaunchKernelGGL((wino4_kernel<8, 2, 2, 12, 120, 16, RAW, SUB, FULL, EPI>), grid, dim3(W4_THREADS), 0, st, a); break;
  }
}

static int w4_launch(W4Args& a, const avsep_conv_desc* d, int mode, bool raw, hipStream_t st) {
  const W4Plan p = w4_plan(d, mode);               // (the group shape depends on the map only, not on the batch)
  a.Hq = p.Hq; a.Wq = p.Wq; a.gyn = p.gyn; a.gxn = p.gxn; a.ngroups = p.ngroups; a.gridM = p.gridM;
  dim3 grid((unsigned)((long long)p.ptiles * p.gridM));
  const bool sub = d->dil == 2, full = !sub && !(d->H & 3) && !(d->W & 3);
  if (a.e.y) {                                     // (data gradient: raw input)
    if (sub) w4_launch_cfg<true, true, false, true>(a, p.cfg, grid, st);
    else if (full) w4_launch_cfg<true, false, true, true>(a, p.cfg, grid, st);
    else w4_launch_cfg<true, false, false, true>(a, p.cfg, grid, st);
  } else if (sub) {
    if (raw) w4_launch_cfg<true, true, false>(a, p.cfg, grid, st);
    else w4_launch_cfg<false, true, false>(a, p.cfg, grid, st);
  } else if (full) {
    if (raw) w4_launch_cfg<true, false, true>(a, p.cfg, grid, st);
    else w4_launch_cfg<false, false, true>(a, p.cfg, grid, st);
  } else {
    if (raw) w4_launch_cfg<true, false, false>(a, p.cfg, grid, st);
    else w4_launch_cfg<false, false, false>(a, p.cfg, grid, st);
  }
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

int w4_fwd(const avsep_conv_desc* d, const float* up, const float* bias, float* y, double* stats, hipStream_t st) {
  W4Args a{};
  a.N = d->N; a.C0 = d->C0; a.C1 = d->Cin - d->C0; a.Cin = d->Cin; a.H = d->H; a.W = d->W; a.Cout = d->Cout;
  a.act0 = d->act0; a.act1 = d->act1;
  a.x0 = d->x0; a.x1 = d->x1; a.sc0 = d->scale0; a.sh0 = d->shift0; a.sc1 = d->scale1; a.sh1 = d->shift1;
  a.up = up; a.out = y; a.bias = bias; a.stats = stats;
  const bool raw = !d->scale0 && !d->scale1 && d->act0 == AVSEP_ACT_NONE && (a.C1 == 0 || d->act1 == AVSEP_ACT_NONE);
  return w4_launch(a, d, 0, raw, st);
}

// dX[N,Cin,H,W] = conv3x3(dY[N,Cout,H,W], flipped / transposed weights)
int w4_dgrad(const avsep_conv_desc* d, const float* up, const float* dy, float* dx, const avsep_act_bwd* e, hipStream_t st) {
  W4Args a{};
  a.N = d->N; a.C0 = d->Cout; a.C1 = 0; a.Cin = d->Cout; a.H = d->H; a.W = d->W; a.Cout = d->Cin;
  a.x0 = dy; a.up = up; a.out = dx;
  if (e) {
    a.e = W4Epi{e->y, e->scale, e->shift, e->residual, e->res_scale, e->res_shift, e->dz2, e->add, e->mean, e->invstd,
                act_slope_host(e->act)};
    a.stats = e->bstats;
  }
  return w4_launch(a, d, 1, true, st);
}
