// 3x3 / stride 1 / 'same' convolution (dilation 1 or 2) in the Winograd F(2x2, 3x3) form on the f32 MFMA:
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A          per 2x2 output tile, 16 products instead of 36
// i.e. 16 independent [Cout x Cin] x [Cin x tiles] GEMMs, one per transform position xi = (i, j), at 4/9 of the
// direct form's MFMA work.  These are the U-Net decoder convs (audio_net.py:75-76,85-87,96-98,180-182) and the
// BasicBlock / dilated layer3-4 convs of the visual trunk (vision_net.py:96-109), forward and data gradient (the data
// gradient is the same conv over dY with flipped, transposed weights — only the weight transform differs).
//
// One workgroup (4 waves) owns 64 output channels x 64 tiles (256 output pixels) and loops over K-tiles of 8 input
// channels; everything between HBM and the MFMA operands happens in LDS / registers:
//   * the input halo patch of the tiles is staged ONCE per K-tile exactly as in halo_kernel.h (folded BatchNorm
//     affine + ReLU/LeakyReLU + the two-source skip concat applied on the way in, zero padding written explicitly);
//   * the transformed weights U[xi][ci][co] come from a pre-transformed image (wino_pack_kernel), one contiguous 32 KB
//     block per (K-tile, M-tile): eight float4 loads and ds_write_b128 per thread, no address arithmetic;
//   * wave w owns transform ROW i = w: a lane reads the two patch rows that row i combines (B^T has two non-zeros per
//     row) as four ds_read_b64, and 8 VALU adds give the B fragments of xi = (w, 0..3) for one (channel, tile):
//     16 MFMAs (4 xi x 2 channel blocks x 2 tile blocks) per 16 VALU and 16 LDS reads;
//   * the 4 x 2 x 2 accumulator tiles (256 registers) stay in the matrix domain until the end; the epilogue applies
//     the column half of A^T . A in registers, exchanges the row half between the waves through LDS (64 KB, the dead
//     operand buffers) and stores float2 pairs, with the BatchNorm sums of the fp32 result as in halo_kernel.h.
// The 64 tiles of a workgroup are G groups of GH x GW tiles; a group is a rectangle of one image (or, for dilation 2,
// of one of its four parity sub-images: a dilated 'same' conv is four independent undilated convs over the pixels of
// equal row/column parity), so 56x56 / 28x28 / 14x14 / 7x7 maps are covered without the 23 % waste a 16x16-pixel
// rectangle has on them.  Patch row / group strides are padded so that the 32 lanes of a ds_read_b64 hit 64 distinct
// banks.
#include <stdlib.h>
#include <type_traits>

#include "common.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int WN_CK = 8;            // input channels per K-tile (host-side packing constant)
constexpr int WN_BM = 64;           // output channels per workgroup
constexpr int WN_AFF_MAX = 2048;    // channels of the folded affine rows kept in LDS (host-checked)
constexpr int WN_A_FLOATS = WN_CK * 16 * WN_BM;   // one K-tile of one M-tile of the transformed weights

struct WArgs {
  int N, C0, C1, Cin, H, W, Cout;   // full-image geometry; Cin = C0 + C1
  int Hq, Wq;                       // the (sub-)image the tiles live in: H x W, or ceil(H/2) x ceil(W/2) per parity class
  int gyn, gxn, ngroups;            // tile groups per (sub-)image along y / x, and in total
  int gridM, act0, act1;
  const float *x0, *x1, *sc0, *sh0, *sc1, *sh1;
  const float* up;                  // [K-tile][M-tile][(channel pair, xi, parity) = 128 rows][64]
  float* out;
  const float* bias;
  double* stats;
};

// rows r = (cpair*16 + xi)*2 + parity of K-tile kt: input channel kt*8 + 2*cpair + parity, xi = 4*i + j
//   mode 0 (forward): g = w[co][ci][.][.]                         column co
//   mode 1 (dgrad)  : g = w[ch][col][2-kh][2-kw] ("in" = co)      column ci
__global__ void wino_pack_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin, int gridM, int nK,
                                 int mode) {
  // one thread per (input channel, output channel): 9 contiguous taps in, U = G g G^T (16 values) out; consecutive
  // threads are consecutive output channels, so every one of the 16 stores is coalesced
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)nK * gridM * (WN_CK * WN_BM)) return;
  const int col = (int)(i % WN_BM), parity = (int)((i / WN_BM) & 1), cpl = (int)((i / (2 * WN_BM)) % (WN_CK / 2));
  const int mt = (int)((i / (WN_CK * WN_BM)) % gridM), kt = (int)(i / ((long long)WN_CK * WN_BM * gridM));
  const int ch = kt * WN_CK + 2 * cpl + parity, co = mt * WN_BM + col;
  float g[9];
  const bool ok = mode == 0 ? (ch < Cin && co < Cout) : (ch < Cout && co < Cin);
#pragma unroll
  for (int k = 0; k < 9; ++k)
    g[k] = !ok ? 0.f : (mode == 0 ? w[((long long)co * Cin + ch) * 9 + k] : w[((long long)ch * Cin + co) * 9 + (8 - k)]);
  // G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]: rows of G g (4 x 3), then columns
  float t[4][3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float s02 = 0.5f * (g[c] + g[6 + c]), h1 = 0.5f * g[3 + c];
    t[0][c] = g[c];
    t[1][c] = s02 + h1;
    t[2][c] = s02 - h1;
    t[3][c] = g[6 + c];
  }
  float* o = out + ((long long)kt * gridM + mt) * WN_A_FLOATS + (cpl * 32 + parity) * WN_BM + col;
#pragma unroll
  for (int ti = 0; ti < 4; ++ti) {
    const float s02 = 0.5f * (t[ti][0] + t[ti][2]), h1 = 0.5f * t[ti][1];
    o[(ti * 4 + 0) * 2 * WN_BM] = t[ti][0];
    o[(ti * 4 + 1) * 2 * WN_BM] = s02 + h1;
    o[(ti * 4 + 2) * 2 * WN_BM] = s02 - h1;
    o[(ti * 4 + 3) * 2 * WN_BM] = t[ti][2];
  }
}

// G groups of GH x GW tiles (G*GH*GW = 64); PWG = LDS row stride of a group's patch, GS = LDS stride between groups
// (both padded for the bank map); SUB: the groups tile the four parity sub-images of a dilation-2 conv; RAW: no affine
// and no activation on the staged tensor (every data gradient).
//
// 512 threads = 8 waves, TWO per SIMD: wave (i, tb) owns transform row i of tile block tb (32 tiles) for both channel
// blocks — 4 x 2 accumulator tiles = 128 registers, so that two waves fit a SIMD.  That is what keeps the matrix pipe
// fed: a wave issues in order and the f32 MFMA holds its issue slot, so with ONE wave per SIMD (the first version of
// this kernel: 4 waves x 256 accumulator registers) every ds_read / VALU / ds_write between two MFMAs was dead time
// for the pipe (measured 49-58 % MFMA busy against 84 % for halo_kernel.h, which runs two workgroups per CU).
constexpr int WN_THREADS = 512;
template <int G, int GH, int GW, int PWG, int GS, bool SUB, bool RAW>
__global__ __launch_bounds__(WN_THREADS) void wino_kernel(WArgs a) {
  static_assert(G * GH * GW == 64, "64 tiles per workgroup");
  constexpr int CK = WN_CK, BM = WN_BM, NT = WN_THREADS;
  constexpr int PHG = 2 * GH + 2, PCG = 2 * GW + 2, GE = PHG * PCG;   // patch of one group (valid elements)
  static_assert(PWG >= PCG && PWG % 2 == 0 && GS >= PHG * PWG && GS % 2 == 0, "patch strides");
  constexpr int PS = G * GS;                                          // floats per channel
  constexpr int NPATCH = CK * G * GE, PE = (NPATCH + NT - 1) / NT;
  constexpr int A_FLOATS = WN_A_FLOATS, AE = A_FLOATS / 4 / NT;       // float4 pieces per thread
  constexpr int P_FLOATS = CK * PS;
  static_assert(2 * A_FLOATS >= 4 * 2 * 2 * 16 * 64, "epilogue exchange fits the weight buffers");
  static_assert(2 * P_FLOATS >= 8 * BM, "statistics partials fit the patch buffers");
  __shared__ __attribute__((aligned(16))) float smem[2 * P_FLOATS + 2 * A_FLOATS + (RAW ? 0 : 2 * WN_AFF_MAX)];
  float* const Pb = smem;                       // patch first: its immediate offsets stay small
  float* const Ab = smem + 2 * P_FLOATS;
  float* const aff_sc = Ab + 2 * A_FLOATS;
  float* const aff_sh = aff_sc + WN_AFF_MAX;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lk = lane >> 5;
  const int wi = wave & 3, wtb = wave >> 2;     // transform row, tile block
  int t = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = t % a.gridM, pt = t / a.gridM, m0 = mt * BM;
  const int per = a.gyn * a.gxn;
  const long long HW = (long long)a.H * a.W;

  // ---- group table: image / origin of the G tile groups of this workgroup (one thread each), read back from LDS ----
  __shared__ int gtab[G][4];                  // {(sub-)image, y0, x0, valid}
  if (tid < G) {
    const int gid = pt * G + tid, gidc = min(gid, a.ngroups - 1);
    const int img = gidc / per, gy = (gidc % per) / a.gxn, gx = gidc % a.gxn;
    gtab[tid][0] = img;
    gtab[tid][1] = gy * 2 * GH;
    gtab[tid][2] = gx * 2 * GW;
    gtab[tid][3] = gid < a.ngroups;
  }
  __syncthreads();

  // ---- patch loader state: one 32-bit element offset + one packed (LDS slot | channel | valid) word per element ----
  unsigned p_off[PE], p_pk[PE];
  int p_lds[PE];
#pragma unroll
  for (int e = 0; e < PE; ++e) {
    const int idx = min(tid + NT * e, NPATCH - 1);
    const int cc = idx / (G * GE), rem = idx % (G * GE), g = rem / GE, r = (rem % GE) / PCG, col = rem % PCG;
    const int img = gtab[g][0], y = gtab[g][1] - 1 + r, x = gtab[g][2] - 1 + col;
    const bool ok = (PE * NT == NPATCH || tid + NT * e < NPATCH) && gtab[g][3] && (unsigned)y < (unsigned)a.Hq &&
                    (unsigned)x < (unsigned)a.Wq;
    const int yc = min(max(y, 0), a.Hq - 1), xc = min(max(x, 0), a.Wq - 1);
    int n = img, fy = yc, fx = xc;
    if constexpr (SUB) {
      n = img >> 2;
      fy = 2 * yc + ((img >> 1) & 1);
      fx = 2 * xc + (img & 1);
    }
    // BYTE offset from the source's base (the K-tile offset is scalar): element offsets < 2^30, host check
    p_off[e] = ok ? 4u * (unsigned)(((long long)n * a.C0 + cc) * HW + (long long)fy * a.W + fx) : 0xffffffffu;   // C1 == C0 when there is a source 1
    p_pk[e] = (unsigned)(cc * PS + g * GS + r * PWG + col) | ((unsigned)cc << 20) | ((unsigned)ok << 24);
    p_lds[e] = cc * PS + g * GS + r * PWG + col;       // (its own register: the store address is then base + immediate)
  }
  if constexpr (!RAW) {                       // folded BatchNorm rows of both sources -> LDS once (identity where absent)
    for (int c = tid; c < a.Cin; c += NT) {
      const bool s0 = c < a.C0;
      const float* sc = s0 ? a.sc0 : a.sc1;
      const float* sh = s0 ? a.sh0 : a.sh1;
      const int cs = s0 ? c : c - a.C0;
      aff_sc[c] = sc ? sc[cs] : 1.f;
      aff_sh[c] = sc ? sh[cs] : 0.f;
    }
    __syncthreads();
  }
  const int kt_switch = a.C1 > 0 ? a.C0 / CK : 0x7fffffff;      // first K-tile of source 1
  const float slope0 = act_slope(a.act0), slope1 = act_slope(a.act1);
  float praw[PE];
  f32x4 areg[AE];
  unsigned w_off[AE];                         // byte offsets of this thread's pieces inside a weight tile
#pragma unroll
  for (int pc = 0; pc < AE; ++pc) w_off[pc] = (unsigned)(tid * 4 + pc * (NT * 4)) * 4u;

  // The staged tile of a K-tile moves in AE + PE independent pieces (one global load, later one LDS store, each) so that
  // the MFMA loop can hang them behind individual MFMAs.  The weight tile is one contiguous 32 KB block per (K-tile,
  // M-tile).  (An LDS-DMA for it would cost an s_waitcnt vmcnt(0) — a full memory round trip — in front of the first
  // ds_read of every K-tile: the compiler orders every LDS read behind an outstanding LDS-DMA it cannot tell apart.)
  constexpr int NPIECE = AE + PE;
  // Global loads are BUFFER loads: scalar resource (source 0 / source 1 / the transformed weights), scalar K-tile offset, one
  // constant 32-bit lane offset per piece — no 64-bit vector address arithmetic in the loop (plain global loads spent
  // 8 v_lshl_add_u64 + 4 add / addc per K-tile on it, and the f32 MFMA does not overlap vector instructions).  An element
  // outside the image has the offset 0xffffffff, which no buffer covers: it loads as 0 (the zero padding of a RAW input).
  const __amdgpu_buffer_rsrc_t rs_x0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.x0, 0, 0xfffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x1 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x1 ? a.x1 : a.x0), 0, 0xfffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_up = __builtin_amdgcn_make_buffer_rsrc((void*)a.up, 0, 0xfffffff0, 0x00020000);
  auto issue_piece = [&](int kt_, int pc) __attribute__((always_inline)) {
    const int kt = __builtin_amdgcn_readfirstlane(kt_);
    if (pc < AE) {
      const unsigned soff = (unsigned)(kt * a.gridM + mt) * (unsigned)(A_FLOATS * 4);        // < 2^32: host check
      areg[pc] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_up, (int)w_off[pc], (int)soff, 0));
    } else if (pc < NPIECE) {
      const bool src1 = kt >= kt_switch;
      const unsigned soff = (unsigned)(src1 ? kt - kt_switch : kt) * (unsigned)(CK * 4) * (unsigned)HW;
      praw[pc - AE] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(src1 ? rs_x1 : rs_x0, (int)p_off[pc - AE], (int)soff, 0));
    }
  };
  auto finish_piece = [&](int kt, int buf, int pc) __attribute__((always_inline)) {
    if (pc < AE) {
      *reinterpret_cast<f32x4*>(Ab + buf * A_FLOATS + (pc * NT + tid) * 4) = areg[pc];
    } else if (pc < NPIECE) {
      const int e = pc - AE;
      float v = praw[e];
      if constexpr (!RAW) {
        const float slope = kt >= kt_switch ? slope1 : slope0;
        const int c = kt * CK + ((p_pk[e] >> 20) & 15);
        v = act_by_slope(fmaf(v, aff_sc[c], aff_sh[c]), slope);
      }
      if constexpr (!RAW) v = ((p_pk[e] >> 24) & 1u) ? v : 0.f;        // (a RAW element outside the image already loaded as 0)
      if (PE * NT == NPATCH || tid + NT * e < NPATCH) (Pb + buf * P_FLOATS)[RAW ? p_lds[e] : (int)(p_pk[e] & 0xfffffu)] = v;   // (non-RAW: registers)
    }
  };

  f32x16 acc[4][2];         // [xi column j][channel block]; NOT zeroed: the first block's eight MFMAs take C = 0 (256 v_mov less)

  // transform row i of B^T:  t = d[rA] + sgn * d[rB]   ((0,2,-) (1,2,+) (2,1,-) (1,3,-))
  const int rA = wi == 0 ? 0 : (wi == 2 ? 2 : 1), rB = wi == 2 ? 1 : (wi == 3 ? 3 : 2);
  const float sgn = wi == 1 ? 1.f : -1.f;
  int lbA, lbB;
  {
    const int tile = wtb * 32 + li, g = tile / (GH * GW), ty = (tile % (GH * GW)) / GW, tx = tile % GW;
    const int lb = lk * PS + g * GS + 2 * ty * PWG + 2 * tx;
    lbA = lb + rA * PWG;
    lbB = lb + rB * PWG;
  }
  const int a_lane = wi * 512 + lk * 64 + li;       // A row ((kp*16 + 4*i + q)*2 + lk), column cb*32 + li

  // ---- main loop.  One BLOCK = the 8 MFMAs of one channel pair; every other instruction of the loop is hung behind
  // one of those MFMAs (sched_barrier after each):
  //   MFMA 0-1 : ds_read of the next block's two patch row pairs            MFMA 2-5 : its eight weight fragments
  //   MFMA 4-7 : the 8 VALU of the next block's transform (two per MFMA)
  //   block 2  : + the LDS stores of the NEXT K-tile (its global loads were issued three blocks earlier), then the
  //              one barrier of the K-tile;   block 3 : + the global loads of the K-tile after that, and its operand
  //              reads already come from the other buffer, so no latency is exposed behind the barrier.
  // Staging is unconditional (tile indices clamped; the tail re-stages the last tile into a dead buffer): a branch
  // around it would fork the accumulator state.
  const int nK = a.Cin / CK;
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float av[2][4][2], v[2][4], tt[4];
  f32x2 bA[2], bB[2];                       // column pairs of the two patch rows of the block being transformed
  auto read_b = [&](int buf, int kp, int which) __attribute__((always_inline)) {
    const float* pp = Pb + buf * P_FLOATS + kp * 2 * PS + (which ? lbB : lbA);
    f32x2* dst = which ? bB : bA;
    dst[0] = *reinterpret_cast<const f32x2*>(pp);
    dst[1] = *reinterpret_cast<const f32x2*>(pp + 2);
  };
  auto read_a = [&](int buf, int kp, int slot, int q) __attribute__((always_inline)) {
    const float* Ak = Ab + buf * A_FLOATS + a_lane + kp * 2048 + q * 128;
    av[slot][q][0] = Ak[0];
    av[slot][q][1] = Ak[32];
  };
  auto transform = [&](int slot, int st) __attribute__((always_inline)) {       // st = 0..3: two VALU each
    if (st == 0) { tt[0] = fmaf(sgn, bB[0][0], bA[0][0]); tt[1] = fmaf(sgn, bB[0][1], bA[0][1]); }
    if (st == 1) { tt[2] = fmaf(sgn, bB[1][0], bA[1][0]); tt[3] = fmaf(sgn, bB[1][1], bA[1][1]); }
    if (st == 2) { v[slot][0] = tt[0] - tt[2]; v[slot][1] = tt[1] + tt[2]; }
    if (st == 3) { v[slot][2] = tt[2] - tt[1]; v[slot][3] = tt[1] - tt[3]; }
  };

#pragma unroll
  for (int pc = 0; pc < NPIECE; ++pc) issue_piece(0, pc);
#pragma unroll
  for (int pc = 0; pc < NPIECE; ++pc) finish_piece(0, 0, pc);
  __syncthreads();
#pragma unroll
  for (int pc = 0; pc < NPIECE; ++pc) issue_piece(min(1, nK - 1), pc);
  read_b(0, 0, 0);
  read_b(0, 0, 1);
#pragma unroll
  for (int q = 0; q < 4; ++q) read_a(0, 0, 0, q);
#pragma unroll
  for (int st = 0; st < 4; ++st) transform(0, st);

  // Two K-tiles per trip with the LDS buffer index a compile-time constant: with `buf = kt & 1` at run time every LDS
  // access of the loop paid a vector add for `base + buf * size` (about 26 of the 88 vector instructions per 32 MFMAs, and
  // the f32 MFMA does not overlap them).
  auto ktile = [&](int kt, auto buf_, auto first_) __attribute__((always_inline)) {
    constexpr int buf = decltype(buf_)::value;
    constexpr bool FIRST = decltype(first_)::value;            // the very first K-tile: its first block initialises the accumulators
    const int kt1 = min(kt + 1, nK - 1), kt2 = min(kt + 2, nK - 1);
#pragma unroll
    for (int kp = 0; kp < CK / 2; ++kp) {
      const int cur = kp & 1, nxt = cur ^ 1;
      const int nbuf = kp + 1 < CK / 2 ? buf : buf ^ 1, nkp = kp + 1 < CK / 2 ? kp + 1 : 0;
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const int q = m >> 1, cb = m & 1;
        if (FIRST && kp == 0) acc[q][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][q][cb], v[cur][q], zero16, 0, 0, 0);
        else acc[q][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][q][cb], v[cur][q], acc[q][cb], 0, 0, 0);
        if (m < 2) read_b(nbuf, nkp, m);
        if (m >= 2 && m < 6) read_a(nbuf, nkp, nxt, m - 2);
        if (m >= 4) transform(nxt, m - 4);
        if (kp == CK / 2 - 2) {
#pragma unroll
          for (int pc = m; pc < NPIECE; pc += 8) finish_piece(kt1, buf ^ 1, pc);
        }
        if (kp == CK / 2 - 1) {
#pragma unroll
          for (int pc = m; pc < NPIECE; pc += 8) issue_piece(kt2, pc);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (kp == CK / 2 - 2) __syncthreads();
    }
  };
  ktile(0, std::integral_constant<int, 0>{}, std::true_type{});
  if (nK > 1) ktile(1, std::integral_constant<int, 1>{}, std::false_type{});
  for (int kt = 2; kt < nK; kt += 2) {
    ktile(kt, std::integral_constant<int, 0>{}, std::false_type{});
    if (kt + 1 < nK) ktile(kt + 1, std::integral_constant<int, 1>{}, std::false_type{});
  }
  __syncthreads();          // the last block's operand prefetch has read LDS: drain before the epilogue reuses it

  // ---- epilogue: Y = A^T M A.  Columns (j) in registers, rows (i) through LDS, one channel block at a time (64 KB, the
  //      dead weight buffers); wave (i, tb) then owns output row parity i&1 of its tile block for the accumulator rows
  //      r = 8*(i>>1) .. +7.  C/D map: tile = lane&31, channel row = (r&3) + 8*(r>>2) + 4*(lane>>5). ----
  float* const Zs = Ab;                       // [i][b][tile block][r][lane]
  float* const s_sum = Pb;                    // [slot = (i&1)*2 + tb][BM]
  float* const s_sq = Pb + 4 * BM;
  const int ea = wi & 1, r0 = 8 * (wi >> 1), slot = ea * 2 + wtb;
  const int tile = wtb * 32 + li, eg = tile / (GH * GW), ety = (tile % (GH * GW)) / GW, etx = tile % GW;
  const int img = gtab[eg][0], oy = gtab[eg][1] + 2 * ety + ea, ox = gtab[eg][2] + 2 * etx;
  const bool ok0 = gtab[eg][3] && oy < a.Hq && ox < a.Wq, ok1 = ok0 && ox + 1 < a.Wq;
  long long obase;
  if constexpr (SUB) obase = (long long)(img >> 2) * a.Cout * HW + (long long)(2 * oy + ((img >> 1) & 1)) * a.W + 2 * ox + (img & 1);
  else obase = (long long)img * a.Cout * HW + (long long)oy * a.W + ox;
  constexpr int XS = SUB ? 2 : 1;
  const bool want_stats = a.stats != nullptr;
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float m0_ = acc[0][cb][r], m1 = acc[1][cb][r], m2 = acc[2][cb][r], m3 = acc[3][cb][r];
      Zs[(((wi * 2 + 0) * 2 + wtb) * 16 + r) * 64 + lane] = m0_ + m1 + m2;
      Zs[(((wi * 2 + 1) * 2 + wtb) * 16 + r) * 64 + lane] = m1 - m2 - m3;
    }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int r = r0 + rr;
      const int lrow = cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk, row = m0 + lrow;
      const bool rok = row < a.Cout;
      const float bias = (a.bias && rok) ? a.bias[row] : 0.f;
      float y[2];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const float* z = Zs + ((b * 2 + wtb) * 16 + r) * 64 + lane;       // + i * (2*2*16*64)
        const float z1 = z[1 * 4096], z2 = z[2 * 4096];
        y[b] = (ea == 0 ? z[0] + z1 + z2 : z1 - z2 - z[3 * 4096]) + bias;
      }
      float* o = a.out + obase + (long long)row * HW;
      float s = 0.f, q = 0.f;
      if (rok && ok1) {
        if constexpr (SUB) { o[0] = y[0]; o[XS] = y[1]; }
        else *reinterpret_cast<f32x2*>(o) = f32x2{y[0], y[1]};
        s = y[0] + y[1];
        q = y[0] * y[0] + y[1] * y[1];
      } else if (rok && ok0) {
        o[0] = y[0];
        s = y[0];
        q = y[0] * y[0];
      }
      if (want_stats) {
        s = half_sum_hi(s);
        q = half_sum_hi(q);
        if (li == 31) {
          s_sum[slot * BM + lrow] = s;
          s_sq[slot * BM + lrow] = q;
        }
      }
    }
    __syncthreads();
  }
  if (want_stats) {
    for (int rr = tid; rr < BM; rr += NT) {
      const int row = m0 + rr;
      if (row < a.Cout) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) { s += s_sum[w * BM + rr]; q += s_sq[w * BM + rr]; }
        atomicAdd(&a.stats[row], (double)s);
        atomicAdd(&a.stats[a.Cout + row], (double)q);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// host side (called from conv.hip)
// ---------------------------------------------------------------------------
struct WnCfg { int gh, gw, g; };     // pixels per group, groups per workgroup
static const WnCfg WN_CFGS[4] = {{16, 16, 1}, {4, 32, 2}, {8, 8, 4}, {2, 16, 8}};

// the group shape that wastes the fewest tiles on an Hq x Wq map (ties: the shape with the smaller halo overhead)
static int wn_cfg(int Hq, int Wq, int tune) {
  if ((tune & 15) >= 1 && (tune & 15) <= 4) return (tune & 15) - 1;      // avsep_conv_desc.tune: measurement tools force a group shape
  int best = 0;
  double be = 0.0;
  for (int c = 0; c < 4; ++c) {
    const WnCfg& k = WN_CFGS[c];
    const double e = (double)Hq * Wq / ((double)roundup(Hq, k.gh) * roundup(Wq, k.gw));
    if (e > be + 0.03) { be = e; best = c; }
  }
  return best;
}

struct WnPlan { int Hq, Wq, cfg, gyn, gxn, ngroups, ptiles, gridM; };
static WnPlan wn_plan(const avsep_conv_desc* d, int mode) {
  WnPlan p{};
  const bool sub = d->dil == 2;
  p.Hq = sub ? (d->H + 1) / 2 : d->H;
  p.Wq = sub ? (d->W + 1) / 2 : d->W;
  p.cfg = wn_cfg(p.Hq, p.Wq, d->tune);
  const WnCfg& k = WN_CFGS[p.cfg];
  p.gyn = cdiv(p.Hq, k.gh);
  p.gxn = cdiv(p.Wq, k.gw);
  p.ngroups = d->N * (sub ? 4 : 1) * p.gyn * p.gxn;
  p.ptiles = cdiv(p.ngroups, k.g);
  p.gridM = cdiv(mode == 0 ? d->Cout : d->Cin, WN_BM);
  return p;
}

bool wn_applicable(const avsep_conv_desc* d, int mode) {
  if ((d->algo & AVSEP_ALGO_NO_WINOGRAD) || d->prec != AVSEP_PREC_F32) return false;
  if (!(d->KH == 3 && d->KW == 3 && d->stride == 1 && (d->dil == 1 || d->dil == 2) && d->pad == d->dil) || d->up2x) return false;
  if ((d->H & 1) || (d->W & 1) || d->H < (d->dil == 1 ? 8 : 14) || d->W < (d->dil == 1 ? 8 : 14)) return false;   // float2 stores; tiny maps stay on split-K
  const int cin = mode == 0 ? d->Cin : d->Cout, cout = mode == 0 ? d->Cout : d->Cin;
  if (cin % WN_CK || cin < 32 || cin > WN_AFF_MAX || cout < 48) return false;
  if (mode == 0) {
    const int C1 = d->Cin - d->C0;
    if (d->C0 % WN_CK || (C1 != 0 && C1 != d->C0)) return false;
  }
  if ((long long)d->N * (mode == 0 ? d->C0 : d->Cout) * d->H * d->W >= 0x3fffffffLL) return false;   // 32-bit BYTE offsets
  const avsep_conv_desc e = plan_desc(d);
  const WnPlan p = wn_plan(&e, mode);
  return (long long)p.ptiles * p.gridM >= 128;      // at least half of the CUs busy (below that the split-K im2col path wins)
}
size_t wn_packed_floats(const avsep_conv_desc* d, int mode) {
  const int cin = mode == 0 ? d->Cin : d->Cout, cout = mode == 0 ? d->Cout : d->Cin;
  return (size_t)(cin / WN_CK) * cdiv(cout, WN_BM) * WN_A_FLOATS;
}
int wn_pack(const avsep_conv_desc* d, const float* w, float* packed, int mode, hipStream_t st) {
  const int cin = mode == 0 ? d->Cin : d->Cout, cout = mode == 0 ? d->Cout : d->Cin;
  const int nK = cin / WN_CK, gridM = cdiv(cout, WN_BM);
  const long long total = (long long)nK * gridM * (WN_CK * WN_BM);
  hipLaunchKernelGGL(wino_pack_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, w, packed, d->Cout, d->Cin, gridM, nK, mode);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

template <bool SUB, bool RAW>
static void wn_launch_cfg(const WArgs& a, int cfg, dim3 grid, hipStream_t st) {
  // <G, GH, GW, PWG, GS>: strides chosen so that the 32 tiles of a tile block read 64 distinct banks (see header)
  switch (cfg) {
    case 0: hipLaunchKernelGGL((wino_kernel<1, 8, 8, 24, 18 * 24, SUB, RAW>), grid, dim3(WN_THREADS), 0, st, a); break;
    case 1: hipLaunchKernelGGL((wino_kernel<2, 2, 16, 48, 6 * 48, SUB, RAW>), grid, dim3(WN_THREADS), 0, st, a); break;
    case 2: hipLaunchKernelGGL((wino_kernel<4, 4, 4, 12, 160, SUB, RAW>), grid, dim3(WN_THREADS), 0, st, a); break;
    default: hipLaunchKernelGGL((wino_kernel<8, 1, 8, 18, 80, SUB, RAW>), grid, dim3(WN_THREADS), 0, st, a); break;
  }
}

static int wn_launch(WArgs& a, const avsep_conv_desc* d, int mode, bool raw, hipStream_t st) {
  const WnPlan p = wn_plan(d, mode);
  a.Hq = p.Hq; a.Wq = p.Wq; a.gyn = p.gyn; a.gxn = p.gxn; a.ngroups = p.ngroups; a.gridM = p.gridM;
  dim3 grid((unsigned)((long long)p.ptiles * p.gridM));
  const bool sub = d->dil == 2;
  if (sub && raw) wn_launch_cfg<true, true>(a, p.cfg, grid, st);
  else if (sub) wn_launch_cfg<true, false>(a, p.cfg, grid, st);
  else if (raw) wn_launch_cfg<false, true>(a, p.cfg, grid, st);
  else wn_launch_cfg<false, false>(a, p.cfg, grid, st);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

int wn_fwd(const avsep_conv_desc* d, const float* up, const float* bias, float* y, double* stats, hipStream_t st) {
  WArgs a{};
  a.N = d->N; a.C0 = d->C0; a.C1 = d->Cin - d->C0; a.Cin = d->Cin; a.H = d->H; a.W = d->W; a.Cout = d->Cout;
  a.act0 = d->act0; a.act1 = d->act1;
  a.x0 = d->x0; a.x1 = d->x1; a.sc0 = d->scale0; a.sh0 = d->shift0; a.sc1 = d->scale1; a.sh1 = d->shift1;
  a.up = up; a.out = y; a.bias = bias; a.stats = stats;
  const bool raw = !d->scale0 && !d->scale1 && d->act0 == AVSEP_ACT_NONE && (a.C1 == 0 || d->act1 == AVSEP_ACT_NONE);
  return wn_launch(a, d, 0, raw, st);
}

// dX[N,Cin,H,W] = conv3x3(dY[N,Cout,H,W], flipped / transposed weights); the identity holds for any dilation with pad == dil
int wn_dgrad(const avsep_conv_desc* d, const float* up, const float* dy, float* dx, hipStream_t st) {
  WArgs a{};
  a.N = d->N; a.C0 = d->Cout; a.C1 = 0; a.Cin = d->Cout; a.H = d->H; a.W = d->W; a.Cout = d->Cin;
  a.x0 = dy; a.up = up; a.out = dx;
  return wn_launch(a, d, 1, true, st);
}
