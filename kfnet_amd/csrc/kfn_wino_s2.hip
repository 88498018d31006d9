// kfn_wino_s2.hip -- 3x3 STRIDE-2 SAME convolution (even H, W) with 25 instead of 36 multiplies per 2x2 outputs.
//
// Reference: tf.layers.conv2d(kernel 3, strides 2, 'same') in cnn_wrapper/network.py:116-135 -- SCoordNet's
// conv2a / conv3a / conv4a (cnn_wrapper/SCoordNet.py:12-27).  For an even input TF pads one row/column AFTER the
// image only:  y[i,j] = bias + sum_{a,b in 0..2} w[a,b] x[2i+a, 2j+b],  x = 0 at row H / column W.
//
// Polyphase + minimal filtering.  Split x into its four parity phases; the 3x3 stride-2 filter is the sum of
//   a 2x2 filter on the (even,even) phase   taps w[0,0] w[0,2] w[2,0] w[2,2]
//   a 2x1 filter on the (even,odd)  phase   taps w[0,1] w[2,1]
//   a 1x2 filter on the (odd,even)  phase   taps w[1,0] w[1,2]
//   a 1x1 filter on the (odd,odd)   phase   tap  w[1,1]
// all of stride 1.  F(2,2) (y0 = m1 + m2, y1 = m2 - m3 with m1 = (d0-d1) g0, m2 = d1 (g0+g1), m3 = (d1-d2) g1)
// computes two outputs of a 2-tap filter with 3 multiplies; nested, a 2x2 output tile costs 9 + 6 + 6 + 4 = 25
// products per input channel instead of 36 -- 25 "positions", each a [tiles x Cin] x [Cin x Cout] GEMM.
//
// Accumulator folding.  With A^T = [[1,1,0],[0,1,-1]] a position feeds the 2x2 outputs with coefficients in
// {0,+1,-1}, and after folding the sign into the (pre-transformed) weights only NINE distinct patterns remain:
//   D00 D01 D10 D11   one output each              (the 4 corner positions of the 2x2 part, the outer positions
//                                                   of the two 1-D parts, the 1x1 part: 16 positions)
//   R0 R1             both outputs of a row        (4 positions)      C0 C1   both outputs of a column (4)
//   Z                 all four                     (1)
//   Y00 = D00 + R0 + C0 + Z,  Y01 = D01 + R0 + C1 + Z,  Y10 = D10 + R1 + C0 + Z,  Y11 = D11 + R1 + C1 + Z.
// Positions with the same pattern accumulate into the SAME 32x32 accumulator: 9 accumulators = 144 AGPRs for
// 25 MFMA streams (the stride-1 F(2x2,3x3) kernels need 16).  The two 1-D parts use the same transformed weight
// for both outputs of their untransformed direction: 16 distinct weight fragments per 8 channels, the same B
// traffic as kfn_wino3.hip for 25/16 of its MFMAs.
//
// Kernel structure: kfn_wino3.hip's.  A workgroup of four waves owns 8 x 4 output tiles (16 x 8 output pixels,
// 33 x 17 input pixels) x 128 output channels; wave w consumes every position for channels n0 + 32 w .. and
// PRODUCES tile row w: lanes 0-31 gather/transform the (even,even) and (odd,odd) phases (13 pixels of the 5x5
// patch), lanes 32-63 the two mixed phases (12 pixels), 4 lanes x 16 B per pixel = the 16 channels of a
// super-step.  Super-step = 2 chunks of 8 channels = 200 MFMAs per wave, one s_barrier each; V lives in 4 LDS
// chunk buffers of [26 slots][2 k-halves][32 tiles][4 floats] (117 KiB).
#include "kfn_common.h"
#include <type_traits>
#include <cstdlib>

#ifndef KFN_WINO_DEFAULT_N_FAST
#define KFN_WINO_DEFAULT_N_FAST 1   // measured (kfn_conv_desc.wino_order M_FAST / N_FAST): 3.8 vs 4.8 GB fetched per launch, same time
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr unsigned OOBV = 0x80000000u;
constexpr int BW = 8, BH = 4;           // tile block: 8 (x) by 4 (y) output tiles = 32 MFMA rows
constexpr int NWAVE = 4;
constexpr int NT = 32 * NWAVE;          // output channels per workgroup
constexpr int NPOS = 25;                // products per tile and channel
constexpr int NSLOT = 26;               // LDS slots per chunk: 13 per producer half-wave
constexpr int NFRAG = 16;               // distinct transformed-weight fragments per chunk
constexpr int NACC = 9;
// LDS layout of one chunk of V per operand precision H16 (0: fp32 fragments of 16 B, 1: fp16 fragments of 8 B); the
// pads spread the lanes of one producer store group (8 lanes x 16 B / 16 lanes x 8 B) over all banks, cf. kfn_wino3.hip
template <bool H16> struct VLayoutS2 {
  static constexpr int FRAG = H16 ? 8 : 16;
  static constexpr int VHALF = H16 ? 256 + 32 : 512 + 64;
  static constexpr int VPOS = 2 * VHALF;
  static constexpr int VBUF = NSLOT * VPOS + (H16 ? 64 : 32);
  static constexpr int LDS = 4 * VBUF > 65536 ? 4 * VBUF : 65536;   // the epilogue stages 4 x 16 KiB
};
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
constexpr int NVBUF = 4;                // two super-steps of 2 chunks
constexpr int SS_CH = 16;               // input channels per super-step

// accumulators
constexpr int D00 = 0, D01 = 1, D10 = 2, D11 = 3, R0 = 4, R1 = 5, C0 = 6, C1 = 7, ZZ = 8;

// The 25 positions in consumption order: 5 groups of 5 positions with 5 different accumulators each (the MFMAs of
// a group run k-step-major, so an accumulator is touched every 5th MFMA).  Per position: LDS slot, weight
// fragment, accumulator.  Slots: 0-8 = 2x2 part [xi][nu] (3 xi + nu), 9-12 = 1x1 part [m][n], 13-18 = (even,odd)
// part [xi][n], 19-24 = (odd,even) part [m][nu].  Fragments: 0-8 = 2x2 part, 9-11 = (even,odd) [xi], 12-14 =
// (odd,even) [nu], 15 = the centre tap.
constexpr int POS_SLOT[NPOS] = {0, 2, 6, 8, 20,   13, 14, 17, 18, 23,   19, 21, 22, 24, 15,   9, 10, 11, 12, 16,   4, 1, 7, 3, 5};
constexpr int POS_FRAG[NPOS] = {0, 2, 6, 8, 13,   9, 9, 11, 11, 13,     12, 14, 12, 14, 10,   15, 15, 15, 15, 10,  4, 1, 7, 3, 5};
constexpr int POS_ACC[NPOS] = {D00, D01, D10, D11, R0,   D00, D01, D10, D11, R1,   D00, D01, D10, D11, C0,
                               D00, D01, D10, D11, C1,   ZZ, R0, R1, C0, C1};
// is position p the last user of its weight fragment inside a chunk?  (then the next chunk's fragment is loaded)
constexpr bool last_user(int p) {
  for (int q = p + 1; q < NPOS; ++q)
    if (POS_FRAG[q] == POS_FRAG[p]) return false;
  return true;
}

struct WinoS2Args {
  const float* x;
  const float* u2;    // [Cin/8][16][cout_pad][8]
  const float* bias;
  float* y;
  int N, H, W, Cin, ldx;
  int Cout, cout_pad, ldy;
  int Ho, Wo;         // output image
  int Th, Tw;         // output tiles per image
  int vrows;          // N * Th
  int bw;
  int tiles_m, tiles_n;
  int relu;
  int n_group;         // workgroup order: n_group channel groups of a tile block adjacent (1 = M fastest, tiles_n = all)
  int wide_store;
  unsigned long long x_bytes;
  unsigned long long y_bytes;
  unsigned u_bytes;
  // split-K (wino_s2b_kernel only; kfn_conv2d_winograd_s2_splitk): k_split copies of the tile grid, copy s accumulates
  // super-steps [s * ss_per_split, ..) and writes RAW partial sums into plane s of the workspace (cf. kfn_wino4.hip)
  int k_split, ss_per_split;
  unsigned long long y_split_bytes;
};

template <int I, int N, class F>
__device__ __forceinline__ void sfor_impl(F& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    sfor_impl<I + 1, N>(f);
  }
}
template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
  sfor_impl<0, N>(f);
}

__device__ __forceinline__ f32x4 bload(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0));
}

__device__ __forceinline__ int xcd_remap_s2(int b, int nwg) {
  int xcd = b & 7;
  int q = nwg >> 3, r = nwg & 7;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (b >> 3);
}

__device__ __forceinline__ f32x2 pk_add(const f32x2& a, const f32x2& b) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// H16: BASELINE config 5's fp16-operand convolutions -- the input transform runs in fp32 on the fp32 activations
// and is rounded to fp16 when it is stored to LDS, the weight fragments arrive as fp16, one v_mfma_f32_32x32x8_f16
// per (position, 8-channel chunk) replaces four v_mfma_f32_32x32x2_f32; accumulation and epilogue stay fp32.
template <bool H16>
__global__ __launch_bounds__(64 * NWAVE, 1) void wino_s2_kernel(WinoS2Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem_s2[];   // [NVBUF][VBUF]
  constexpr int VHALF = VLayoutS2<H16>::VHALF, VPOS = VLayoutS2<H16>::VPOS, VBUF = VLayoutS2<H16>::VBUF, FRAG = VLayoutS2<H16>::FRAG;
  using frag_t = std::conditional_t<H16, f32x2, f32x4>;
  constexpr int NVG = H16 ? 3 : 2;      // V fragment groups in flight

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int nwg = p.tiles_m * p.tiles_n;
  const int tile = xcd_remap_s2(blockIdx.x, nwg);
  // Workgroups that run side by side on an XCD share its L2.  n_fast: the Cout/128 channel groups of one tile
  // block are neighbours (the input crosses HBM once, every group's weights are live at once); else M fastest
  // (one channel group's weights stay hot, the input is fetched once per channel group).
  const int per = p.tiles_m * p.n_group;        // (n_group divides tiles_n)
  const int gset = tile / per, rem_ = tile - gset * per;
  const int tm = rem_ / p.n_group;
  const int tn = gset * p.n_group + (rem_ - tm * p.n_group);
  const int cb = tm % p.bw, rb = tm / p.bw;
  const int n0 = tn * NT + wave * 32;

  // ---- block geometry (uniform): virtual tile rows run over the batch image after image ----------
  const int vr0 = rb * BH;
  const int img0 = vr0 / p.Th;
  const int ty0 = vr0 - img0 * p.Th;
  const int brk = (p.Th - ty0 < BH) ? (p.Th - ty0) : BH;   // tile rows >= brk belong to image img0 + 1

  const unsigned long long a_base = (unsigned long long)img0 * p.H * p.W * p.ldx * 4ull;
  const unsigned long long a_rest = p.x_bytes - a_base;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(p.x)) + a_base, 0,
      (int)(a_rest < 0x7fffffffull ? a_rest : 0x7fffffffull), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsU =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.u2), 0, p.u_bytes, 0x00020000);

  // ---- this lane as a PRODUCER ------------------------------------------------------------------------
  // wave w owns tile row w; lane = (phase set ps, tile column tc, channel quad q of the super-step's 16 channels):
  // 4 neighbouring lanes read 64 contiguous bytes of a pixel.  Set 0 gathers patch pixels (u, v) of the
  // (even,even) phase [3x3, slot 3m+n = pixel (2m, 2n)] and of the (odd,odd) phase [slot 9+2m+n = (2m+1, 2n+1)];
  // set 1 the (even,odd) phase [slot 2m+n = (2m, 2n+1)] and the (odd,even) phase [slot 6+3m+n = (2m+1, 2n)].
  const int ps = lane >> 5, tc = (lane >> 2) & 7, pq = lane & 3;
  unsigned goff[13];     // byte offset of this lane's i-th patch pixel at its channel quad, or OOB (-> 0)
  {
    const int tr = wave;
    const int img_rel = tr < brk ? 0 : 1;
    const int ty = tr < brk ? ty0 + tr : tr - brk;
    const int tx = cb * BW + tc;
    const bool tile_ok = (vr0 + tr < p.vrows) && tx < p.Tw;
#pragma unroll
    for (int i = 0; i < 13; ++i) {
      const int ua = i < 9 ? 2 * (i / 3) : 2 * ((i - 9) >> 1) + 1, va = i < 9 ? 2 * (i % 3) : 2 * ((i - 9) & 1) + 1;
      const int ub = i < 6 ? 2 * (i >> 1) : 2 * ((i - 6) / 3) + 1, vb = i < 6 ? 2 * (i & 1) + 1 : 2 * ((i - 6) % 3);
      const int u = ps ? ub : ua, v = ps ? vb : va;
      const int yy = 4 * ty + u, xx = 4 * tx + v;
      const bool ok = tile_ok && yy < p.H && xx < p.W && (i < 12 || ps == 0);
      goff[i] = ok ? (unsigned)((((img_rel * p.H + yy) * p.W + xx) * p.ldx + pq * 4) * 4) : OOBV;
    }
  }
  // LDS addresses: slot s of chunk c at c*VBUF + s*VPOS + half*VHALF + tile*16
  const int v_lane = (lane >> 5) * VHALF + (lane & 31) * FRAG;                                          // consumer
  const int v_st = (pq >> 1) * VBUF + ps * 13 * VPOS + (pq & 1) * VHALF + wave * (8 * FRAG) + tc * FRAG;   // producer
  const int n_chunks = p.Cin / 8;
  const int n_super = n_chunks / 2;
  const int s_last = n_super - 1;

  // ---- this lane as a CONSUMER ---------------------------------------------------------------
  const int li = lane & 31, lh = lane >> 5;
  const unsigned voff_b = (unsigned)(((n0 + li) * 8 + lh * 4) * (H16 ? 2 : 4));
  const unsigned b_step = (unsigned)p.cout_pad * (H16 ? 16u : 32u);     // bytes between fragments
  const int q_last = n_chunks * NFRAG - 1;

  const int n = n0 + li;
  const bool n_ok = n < p.Cout;
  const float bv = (p.bias != nullptr && n_ok) ? p.bias[n] : 0.f;
  f32x16 acc[NACC];
#pragma unroll
  for (int g = 0; g < NACC; ++g)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[g][e] = (g <= D11) ? bv : 0.f;   // the bias rides in the four one-output accumulators

  f32x2 pv[26];          // producer: 13 patch pixels x 4 channels
  frag_t bq[NFRAG];      // weight fragments of the current chunk (reloaded for the next one after their last use)
  frag_t vq[NVG][5];     // V fragments: the group in flight and the next one (fp16: the next two)

  auto p_gather = [&](auto ic, int ss) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    const int sc = ss < s_last ? ss : s_last;
    const f32x4 q = bload(rsA, goff[i], (unsigned)(sc * (SS_CH * 4)));
    pv[2 * i] = q.xy;
    pv[2 * i + 1] = q.zw;
  };
  // The transforms of the two half-waves differ.  ONE asm block that narrows EXEC to each half in turn: a C++
  // `if (ps == 0) ... else ...` here is control flow in the middle of the MFMA stream, around which hipcc shuffles
  // the accumulators between AGPRs and VGPRs (224 v_accvgpr moves per super-step).  Lanes 0-31: B^T e B on the 3x3
  // (even,even) patch e[m][n] = pv[3m+n] (rows, then columns; the (odd,odd) pixels pass through).  Lanes 32-63:
  // (even,odd) f[m][n] = pv[2m+n] along m, (odd,even) g[m][n] = pv[6+3m+n] along n.  pv[i] = channel pairs
  // (2i, 2i+1) of pixel i; every line is (d0,d1,d2) -> (d0-d1, d1, d1-d2).  EXEC is all ones before and after.
  auto p_transform = [&]() __attribute__((always_inline)) {
    asm volatile(
      "s_mov_b32 exec_hi, 0\n\t"
      "v_pk_add_f32 %0, %0, %2 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %4, %2, %4 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %1, %1, %3 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %5, %3, %5 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %6, %6, %8 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %10, %8, %10 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %7, %7, %9 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %11, %9, %11 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %12, %12, %14 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %16, %14, %16 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %13, %13, %15 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %17, %15, %17 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, %6 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %12, %6, %12 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %1, %1, %7 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %13, %7, %13 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %2, %2, %8 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %14, %8, %14 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %3, %3, %9 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %15, %9, %15 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %4, %4, %10 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %16, %10, %16 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %5, %5, %11 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %17, %11, %17 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "s_mov_b32 exec_lo, 0\n\t"
      "s_mov_b32 exec_hi, -1\n\t"
      "v_pk_add_f32 %0, %0, %4 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %8, %4, %8 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %1, %1, %5 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %9, %5, %9 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %2, %2, %6 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %10, %6, %10 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %3, %3, %7 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %11, %7, %11 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %12, %12, %14 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %16, %14, %16 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %13, %13, %15 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %17, %15, %17 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %18, %18, %20 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %22, %20, %22 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %19, %19, %21 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %23, %21, %23 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "s_mov_b32 exec_lo, -1\n\t"
      : "+v"(pv[0]), "+v"(pv[1]), "+v"(pv[2]), "+v"(pv[3]), "+v"(pv[4]), "+v"(pv[5]), "+v"(pv[6]), "+v"(pv[7]), "+v"(pv[8]), "+v"(pv[9]), "+v"(pv[10]), "+v"(pv[11]), "+v"(pv[12]), "+v"(pv[13]), "+v"(pv[14]), "+v"(pv[15]), "+v"(pv[16]), "+v"(pv[17]), "+v"(pv[18]), "+v"(pv[19]), "+v"(pv[20]), "+v"(pv[21]), "+v"(pv[22]), "+v"(pv[23]), "+v"(pv[24]), "+v"(pv[25]));
  };
  auto p_store = [&](auto gc, int ss) __attribute__((always_inline)) {
    constexpr int g = decltype(gc)::value;
    const f32x4 q = {pv[2 * g].x, pv[2 * g].y, pv[2 * g + 1].x, pv[2 * g + 1].y};
    if constexpr (H16) {
      const f16x4 h = {(_Float16)q.x, (_Float16)q.y, (_Float16)q.z, (_Float16)q.w};   // RNE
      *reinterpret_cast<f16x4*>(smem_s2 + (ss & 1) * (2 * VBUF) + v_st + g * VPOS) = h;
    } else {
      *reinterpret_cast<f32x4*>(smem_s2 + (ss & 1) * (2 * VBUF) + v_st + g * VPOS) = q;
    }
  };
  auto b_load = [&](auto fc, int ch) __attribute__((always_inline)) {
    constexpr int f = decltype(fc)::value;
    const int qi = ch * NFRAG + f;
    const int qc = qi < q_last ? qi : q_last;
    if constexpr (H16) bq[f] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsU, voff_b, (unsigned)qc * b_step, 0));
    else bq[f] = bload(rsU, voff_b, (unsigned)qc * b_step);
  };
  // V fragment of position pp of chunk ch into half `hb` of the double buffer
  auto v_read = [&](auto pc, auto hb, int ch) __attribute__((always_inline)) {
    constexpr int pp = decltype(pc)::value;
    constexpr int slot = POS_SLOT[pp];
    vq[decltype(hb)::value][pp % 5] =
        *reinterpret_cast<const frag_t*>(smem_s2 + (ch & (NVBUF - 1)) * VBUF + slot * VPOS + v_lane);
  };

  // ---- prologue: every wave produces its tile row of super-step 0 -------------------------------
  sfor<13>([&](auto ic) { p_gather(ic, 0); });
  sfor<NFRAG>([&](auto fc) { b_load(fc, 0); });
  p_transform();
  sfor<13>([&](auto gc) { p_store(gc, 0); });
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // One super-step = chunks 2ks, 2ks+1: 2 x 100 MFMA slots (fp16: 2 x 25).  Producer work for super-step ks+1: 13
  // gathers behind the first slots, the transform burst in the second chunk, the 13 stores after it; then the
  // barrier.
  constexpr int SPC = H16 ? 25 : 100;                       // MFMA slots per chunk
  constexpr int GSTEP = H16 ? 2 : 8, XSLOT = H16 ? 36 : 150, SSLOT = H16 ? 37 : 160, SSTEP = H16 ? 1 : 2;
  for (int ks = 0; ks < n_super; ++ks) {
    const int nxt = ks + 1;
    // the first group(s) of the first chunk: nothing of this super-step could be read before the barrier.  10 groups
    // per super-step; fp32 rotates 2 fragment buffers (back at 0 here), fp16 3 (restarted at 0 here).
    sfor<5>([&](auto pc) { v_read(pc, std::integral_constant<int, 0>{}, 2 * ks); });
    if constexpr (H16) sfor<5>([&](auto pc) { v_read(std::integral_constant<int, 5 + decltype(pc)::value>{}, std::integral_constant<int, 1>{}, 2 * ks); });
    sfor<2>([&](auto cc_) {
      constexpr int cc = decltype(cc_)::value;
      const int ch = 2 * ks + cc;
      sfor<SPC>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int G = H16 ? j / 5 : j / 20, t = H16 ? 3 : (j % 20) / 5, k = j % 5;
        constexpr int pp = G * 5 + k;
        constexpr int ia = POS_ACC[pp], ifr = POS_FRAG[pp];
        constexpr int gi = cc * 5 + G;            // group index inside the super-step
        constexpr int hb = gi % NVG;              // fragment buffer this group reads
        if constexpr (H16)
          acc[ia] = __builtin_amdgcn_mfma_f32_32x32x8f16(__builtin_bit_cast(f16x4, vq[hb][k]), __builtin_bit_cast(f16x4, bq[ifr]), acc[ia], 0, 0, 0);
        else
          acc[ia] = __builtin_amdgcn_mfma_f32_32x32x2f32(vq[hb][k][t], bq[ifr][t], acc[ia], 0, 0, 0);
        // the next chunk's fragment, once this chunk is done with the registers
        if constexpr (t == 3 && last_user(pp)) b_load(std::integral_constant<int, ifr>{}, ch + 1);
        // V of a later group (same chunk, or the second chunk of this super-step): fp32 the next group during
        // k-step 1, fp16 the group after next right behind the MFMA that frees the buffer
        if constexpr (H16) {
          constexpr int tg = gi + 2;              // target group inside the super-step
          if constexpr (tg < 10) v_read(std::integral_constant<int, (tg % 5) * 5 + k>{}, std::integral_constant<int, tg % NVG>{}, 2 * ks + tg / 5);
        } else if constexpr (t == 1) {
          if constexpr (G < 4) v_read(std::integral_constant<int, (G + 1) * 5 + k>{}, std::integral_constant<int, hb ^ 1>{}, ch);
          else if constexpr (cc == 0) v_read(std::integral_constant<int, k>{}, std::integral_constant<int, hb ^ 1>{}, ch + 1);
        }
        constexpr int sj = cc * SPC + j;   // slot inside the super-step
        if constexpr (sj < 13 * GSTEP && sj % GSTEP == 0) p_gather(std::integral_constant<int, sj / GSTEP>{}, nxt);
        if constexpr (sj == XSLOT) p_transform();
        if constexpr (sj >= SSLOT && sj < SSLOT + 13 * SSTEP && (sj - SSLOT) % SSTEP == 0) p_store(std::integral_constant<int, (sj - SSLOT) / SSTEP>{}, nxt);
        __builtin_amdgcn_sched_barrier(0);
      });
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }

  // ---- epilogue: 9 accumulators -> 2x2 outputs, transposed through LDS, 16-byte stores (cf. kfn_wino3.hip) ----
  const bool relu = p.relu != 0;
  const unsigned long long y_base = (unsigned long long)img0 * p.Ho * p.Wo * p.ldy * 4ull;
  const unsigned long long y_rest = p.y_bytes - y_base;
  const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<char*>(p.y) + y_base, 0, (int)(y_rest < 0x7fffffffull ? y_rest : 0x7fffffffull), 0x00020000);
  const int txl = cb * BW + 4 * lh;
  const unsigned voff_y = (unsigned)((2 * txl * p.ldy + n) * 4);
  const int pix_bytes = p.ldy * 4;
  auto out_transform = [&](auto&& put) __attribute__((always_inline)) {
#pragma unroll
    for (int ep = 0; ep < 8; ++ep) {
      const int e0 = 2 * ep;
      auto pr = [&](int a) __attribute__((always_inline)) { return f32x2{acc[a][e0], acc[a][e0 + 1]}; };
      const f32x2 z0 = pk_add(pr(R0), pr(ZZ)), z1 = pk_add(pr(R1), pr(ZZ));
      const f32x2 c0 = pr(C0), c1 = pr(C1);
      const f32x2 o0 = pk_add(pk_add(pr(D00), z0), c0), o1 = pk_add(pk_add(pr(D01), z0), c1);
      const f32x2 o2 = pk_add(pk_add(pr(D10), z1), c0), o3 = pk_add(pk_add(pr(D11), z1), c1);
      const int trow = e0 >> 2, ec = e0 & 3;
      put(o0.x, trow, ec, 0, 0); put(o1.x, trow, ec, 0, 1); put(o2.x, trow, ec, 1, 0); put(o3.x, trow, ec, 1, 1);
      put(o0.y, trow, ec + 1, 0, 0); put(o1.y, trow, ec + 1, 0, 1); put(o2.y, trow, ec + 1, 1, 0); put(o3.y, trow, ec + 1, 1, 1);
    }
  };
  if (p.wide_store) {
    char* const stg = smem_s2 + wave * 16384;
    const int st_w = lh * 1024 + li * 4;   // block pixel (oy, ox) = (2 trow + a, 8 lh + 2 ec + b) -> row oy*16 + ox
    out_transform([&](float v, int trow, int ec, int a, int b) __attribute__((always_inline)) {
      *reinterpret_cast<float*>(stg + st_w + ((2 * trow + a) * 16 + 2 * ec + b) * 128) = v;
    });
    const int oxl = lane >> 3, nq = lane & 7;            // store lane: pixel column oxl (+8), channel quad nq
    const unsigned voff_q = (unsigned)((oxl * p.ldy + n0 + nq * 4) * 4);
    const bool q_ok = n0 + nq * 4 < p.Cout;
    const int ox0 = 2 * cb * BW;
    const unsigned voff_h[2] = {(q_ok && ox0 + oxl < p.Wo) ? voff_q : OOBV, (q_ok && ox0 + 8 + oxl < p.Wo) ? voff_q : OOBV};
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      f32x4 v = *reinterpret_cast<const f32x4*>(stg + i * 1024 + lane * 16);
      if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      const int trow = i >> 2, a = (i >> 1) & 1, hx = i & 1;
      const int img_rel = trow < brk ? 0 : 1;
      const int ty = trow < brk ? ty0 + trow : trow - brk;
      const int oy = 2 * ty + a;
      const bool row_ok = vr0 + trow < p.vrows && oy < p.Ho;        // uniform
      const unsigned soff = (unsigned)(((img_rel * p.Ho + oy) * p.Wo + ox0 + 8 * hx) * pix_bytes);
      kfn::buffer_store_b128<KFN_NT_STORE_AUX>(v, rsY, row_ok ? voff_h[hx] : OOBV, soff);
    }
  } else {
    // Cout or the row pitch not a multiple of 4 floats: one dword per store
    out_transform([&](float v, int trow, int ec, int a, int b) __attribute__((always_inline)) {
      const int img_rel = trow < brk ? 0 : 1;
      const int ty = trow < brk ? ty0 + trow : trow - brk;
      const int oy = 2 * ty + a;
      v = relu ? fmaxf(v, 0.f) : v;
      const unsigned soff = (unsigned)(((img_rel * p.Ho + oy) * p.Wo + 2 * ec + b) * pix_bytes);
      const int tx = txl + ec;
      const bool ok = n_ok && tx < p.Tw && 2 * tx + b < p.Wo && vr0 + trow < p.vrows && oy < p.Ho;
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsY, ok ? voff_y : OOBV, soff, 0);
    });
  }
}


// =====================================================================================================================
// The EIGHT-WAVE form (two waves per SIMD), fp32 operands: wino_s2b_kernel.  Same tile block (8 x 4 tiles x 128 channels), same
// positions / accumulator folding; on v_mfma_f32_16x16x4_f32 a wave owns 25 positions x 32 tiles x SIXTEEN channels = 9
// accumulators x 2 tile halves x 4 registers = 72 -- two waves per SIMD fit, one wave's loads / transform / LDS traffic issue
// under the other's MFMAs (what this bought the F(4x4,3x3) kernel: kfn_wino4.hip, wino4b_kernel).
//   * V of a chunk: [26 slots][4 k][16 rows][2 tile halves][2 k-steps] floats (+16: the two chunks of a super-step 16 banks
//     apart): lane (row r, k) of the A operand reads ONE ds_read_b128 per (chunk, position); input channel of (k, k-step s) =
//     2 k + s.  Rows are stored at r ^ k (conflict-free reads, cf. wino4b_kernel).
//   * producer: every wave; wave w = tile row w & 3 x tile columns 4 (w >> 2) .. + 3; lane = (phase set, tile column, channel
//     PAIR of the super-step's 16): 13 loads of 8 bytes, 8 lanes on 64 contiguous bytes of a pixel; the transform is 12 / 8
//     packed subtractions per half-wave (one register per pixel).
//   * weights per PAIR of fragments (graph.pack_winograd_s2_kernel_b: [Cin/8][8 pairs][cout_pad][4 k][2 fragments][2 k-steps]):
//     one 16-byte load per lane, chunk and pair.
//   * epilogue: lane (channel, k) holds tiles 16 th + 4 k + e; per wave an 8 KiB staging image [8 rows][16 px][16 ch], 16-byte
//     stores of 64-byte runs.
constexpr int SB_VPOS = 256;                        // floats per slot
constexpr int SB_VBUF = NSLOT * SB_VPOS + 16;       // floats per chunk buffer
constexpr int SB_LDS_V = NVBUF * SB_VBUF * 4;       // 106 752 B
constexpr int SB_LDS = SB_LDS_V > 65536 ? SB_LDS_V : 65536;   // the epilogue stages 8 x 8 KiB
constexpr int pair_last_user_pos(int q) {           // the last position of a chunk that uses fragment 2q or 2q + 1
  int last = -1;
  for (int pp = 0; pp < NPOS; ++pp)
    if (POS_FRAG[pp] / 2 == q) last = pp;
  return last;
}

__global__ __launch_bounds__(512, 1) void wino_s2b_kernel(WinoS2Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem_s2[];
  float* const smf = reinterpret_cast<float*>(smem_s2);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int nwg = p.tiles_m * p.tiles_n;
  const int split = p.k_split > 1 ? (int)(blockIdx.x / (unsigned)nwg) : 0;      // K split slowest
  const int tile = xcd_remap_s2((int)blockIdx.x - split * nwg, nwg);
  const int per = p.tiles_m * p.n_group;
  const int gset = tile / per, rem_ = tile - gset * per;
  const int tm = rem_ / p.n_group;
  const int tn = gset * p.n_group + (rem_ - tm * p.n_group);
  const int cb = tm % p.bw, rb = tm / p.bw;
  const int n0 = tn * NT + wave * 16;

  const int vr0 = rb * BH;
  const int img0 = vr0 / p.Th;
  const int ty0 = vr0 - img0 * p.Th;
  const int brk = (p.Th - ty0 < BH) ? (p.Th - ty0) : BH;
  const unsigned long long a_base = (unsigned long long)img0 * p.H * p.W * p.ldx * 4ull;
  const unsigned long long a_rest = p.x_bytes - a_base;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(p.x)) + a_base, 0,
      (int)(a_rest < 0x7fffffffull ? a_rest : 0x7fffffffull), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsU =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.u2), 0, p.u_bytes, 0x00020000);

  // ---- PRODUCER: tile (row tr = wave & 3, column tc = 4 (wave >> 2) + ((lane >> 3) & 3)), phase set ps = lane >> 5, channel pair
  // pq8 = lane & 7 of the super-step's 16 (chunk pq8 >> 2, k = pq8 & 3) ----
  const int ps = lane >> 5, pq8 = lane & 7;
  const int ptr_ = wave & 3, ptc = 4 * (wave >> 2) + ((lane >> 3) & 3);
  unsigned goff[13];
  {
    const int img_rel = ptr_ < brk ? 0 : 1;
    const int ty = ptr_ < brk ? ty0 + ptr_ : ptr_ - brk;
    const int tx = cb * BW + ptc;
    const bool tile_ok = (vr0 + ptr_ < p.vrows) && tx < p.Tw;
#pragma unroll
    for (int i = 0; i < 13; ++i) {
      const int ua = i < 9 ? 2 * (i / 3) : 2 * ((i - 9) >> 1) + 1, va = i < 9 ? 2 * (i % 3) : 2 * ((i - 9) & 1) + 1;
      const int ub = i < 6 ? 2 * (i >> 1) : 2 * ((i - 6) / 3) + 1, vb = i < 6 ? 2 * (i & 1) + 1 : 2 * ((i - 6) % 3);
      const int u = ps ? ub : ua, v = ps ? vb : va;
      const int yy = 4 * ty + u, xx = 4 * tx + v;
      const bool ok = tile_ok && yy < p.H && xx < p.W && (i < 12 || ps == 0);
      goff[i] = ok ? (unsigned)((((img_rel * p.H + yy) * p.W + xx) * p.ldx + pq8 * 2) * 4) : OOBV;
    }
  }
  const int pt = 8 * ptr_ + ptc, pk = pq8 & 3;
  const int v_st = (pq8 >> 2) * SB_VBUF + ps * 13 * SB_VPOS + pk * 64 + (((pt & 15) ^ pk) * 4) + (pt >> 4) * 2;   // floats
  const int n_chunks = p.Cin / 8;
  const int ks0 = split * p.ss_per_split;                                   // first super-step of this split (0 without split-K)
  const int n_super = p.k_split > 1 ? ((n_chunks / 2 - ks0) < p.ss_per_split ? (n_chunks / 2 - ks0) : p.ss_per_split) : n_chunks / 2;
  const int s_last = n_super - 1;
  const int c_base = 2 * ks0;

  // ---- CONSUMER ----
  const int rl = lane & 15, kl = lane >> 4;
  const int v_lane = kl * 64 + ((rl ^ kl) * 4);                         // floats; + slot * SB_VPOS
  const unsigned voff_b = (unsigned)(((n0 + rl) * 16 + kl * 4) * 4);
  const unsigned b_step = (unsigned)p.cout_pad * 64u;                   // bytes between fragment PAIRS
  const int q_last = n_chunks * (NFRAG / 2) - 1;
  const int n = n0 + rl;
  const bool n_ok = n < p.Cout;
  const float bv = (p.bias != nullptr && n_ok) ? p.bias[n] : 0.f;
  f32x4 acc[NACC][2];
#pragma unroll
  for (int g = 0; g < NACC; ++g)
#pragma unroll
    for (int th = 0; th < 2; ++th) {
      const float v0 = (g <= D11) ? bv : 0.f;       // the bias rides in the four one-output accumulators
      acc[g][th] = f32x4{v0, v0, v0, v0};
    }

  f32x2 pv[13];          // producer: 13 patch pixels x 2 channels
  f32x4 bq[NFRAG / 2];   // weight fragment pairs of the current chunk
  f32x4 vq[2][5];        // V fragments: the group in flight and the next one

  auto p_gather = [&](auto ic, int ss) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    const int sc = ss < s_last ? ss : s_last;
    pv[i] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsA, goff[i], (unsigned)((ks0 + sc) * (SS_CH * 4)), 0));
  };
  // (d0, d1, d2) -> (d0 - d1, d1, d1 - d2) along the transformed axes; the two half-waves differ (see wino_s2_kernel): lanes
  // 0-31 the 3x3 (even,even) patch pv[3m+n] rows then columns, lanes 32-63 (even,odd) pv[2m+n] along m and (odd,even)
  // pv[6+3m+n] along n.  One asm block narrowing EXEC, no control flow in the MFMA stream.
  auto p_transform = [&]() __attribute__((always_inline)) {
    asm volatile(
      "s_mov_b32 exec_hi, 0\n\t"
      "v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %2, %1, %2 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %3, %3, %4 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %5, %4, %5 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %6, %6, %7 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %8, %7, %8 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, %3 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %6, %3, %6 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %1, %1, %4 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %7, %4, %7 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %2, %2, %5 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %8, %5, %8 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "s_mov_b32 exec_lo, 0\n\t"
      "s_mov_b32 exec_hi, -1\n\t"
      "v_pk_add_f32 %0, %0, %2 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %4, %2, %4 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %1, %1, %3 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %5, %3, %5 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %6, %6, %7 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %8, %7, %8 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %9, %9, %10 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %11, %10, %11 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "s_mov_b32 exec_lo, -1\n\t"
      : "+v"(pv[0]), "+v"(pv[1]), "+v"(pv[2]), "+v"(pv[3]), "+v"(pv[4]), "+v"(pv[5]), "+v"(pv[6]), "+v"(pv[7]), "+v"(pv[8]),
        "+v"(pv[9]), "+v"(pv[10]), "+v"(pv[11]), "+v"(pv[12]));
  };
  auto p_store = [&](auto gc, int ss) __attribute__((always_inline)) {
    constexpr int g = decltype(gc)::value;
    *reinterpret_cast<f32x2*>(smf + (ss & 1) * (2 * SB_VBUF) + v_st + g * SB_VPOS) = pv[g];
  };
  auto b_load = [&](auto qc_, int ch) __attribute__((always_inline)) {     // fragment pair q of chunk ch
    constexpr int q = decltype(qc_)::value;
    const int qi = (c_base + ch) * (NFRAG / 2) + q;
    const int qc = qi < q_last ? qi : q_last;
    bq[q] = bload(rsU, voff_b, (unsigned)qc * b_step);
  };
  auto v_read = [&](auto pc, auto hb, int ch) __attribute__((always_inline)) {
    constexpr int pp = decltype(pc)::value;
    constexpr int slot = POS_SLOT[pp];
    vq[decltype(hb)::value][pp % 5] = *reinterpret_cast<const f32x4*>(smf + (ch & (NVBUF - 1)) * SB_VBUF + slot * SB_VPOS + v_lane);
  };

  // ---- prologue ----
  sfor<13>([&](auto ic) { p_gather(ic, 0); });
  sfor<NFRAG / 2>([&](auto qc_) { b_load(qc_, 0); });
  p_transform();
  sfor<13>([&](auto gc) { p_store(gc, 0); });
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // One super-step = chunks 2ks, 2ks+1: 2 x 100 MFMA slots (5 groups of 5 positions x 2 k-steps x 2 tile halves).
  constexpr int SPC = 100;
  constexpr int GSTEP = 8, XSLOT = 150, SSLOT = 160, SSTEP = 2;
  for (int ks = 0; ks < n_super; ++ks) {
    const int nxt = ks + 1;
    sfor<5>([&](auto pc) { v_read(pc, std::integral_constant<int, 0>{}, 2 * ks); });
    sfor<2>([&](auto cc_) {
      constexpr int cc = decltype(cc_)::value;
      const int ch = 2 * ks + cc;
      sfor<SPC>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int G = j / 20, t = (j % 20) / 5, k = j % 5;          // t = 2 s + th: (s, th) = (0,0) (0,1) (1,0) (1,1)
        constexpr int sk = t / 2, th = t % 2;
        constexpr int pp = G * 5 + k;
        constexpr int ia = POS_ACC[pp], ifr = POS_FRAG[pp];
        constexpr int gi = cc * 5 + G;
        constexpr int hb = gi % 2;
        acc[ia][th] = __builtin_amdgcn_mfma_f32_16x16x4f32(vq[hb][k][2 * th + sk], bq[ifr / 2][2 * (ifr % 2) + sk], acc[ia][th], 0, 0, 0);
        // the next chunk's fragment pair, once this chunk is done with both of its fragments
        if constexpr (t == 3 && pair_last_user_pos(ifr / 2) == pp) b_load(std::integral_constant<int, ifr / 2>{}, ch + 1);
        // V of the next group during the second quarter of this one
        if constexpr (t == 1) {
          if constexpr (G < 4) v_read(std::integral_constant<int, (G + 1) * 5 + k>{}, std::integral_constant<int, hb ^ 1>{}, ch);
          else if constexpr (cc == 0) v_read(std::integral_constant<int, k>{}, std::integral_constant<int, hb ^ 1>{}, ch + 1);
        }
        constexpr int sj = cc * SPC + j;
        if constexpr (sj < 13 * GSTEP && sj % GSTEP == 0) p_gather(std::integral_constant<int, sj / GSTEP>{}, nxt);
        if constexpr (sj == XSLOT) p_transform();
        if constexpr (sj >= SSLOT && sj < SSLOT + 13 * SSTEP && (sj - SSLOT) % SSTEP == 0) p_store(std::integral_constant<int, (sj - SSLOT) / SSTEP>{}, nxt);
        __builtin_amdgcn_sched_barrier(0);
      });
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }

  // ---- epilogue: 9 accumulators -> 2x2 outputs.  Lane (channel n0 + rl, k = kl), accumulator half th, element e: tile
  // 16 th + 4 kl + e = (tile row 2 th + (kl >> 1), tile column 4 (kl & 1) + e). ----
  const bool relu = p.relu != 0;
  const unsigned long long y_base = (unsigned long long)img0 * p.Ho * p.Wo * p.ldy * 4ull;
  const unsigned long long y_rest = p.y_bytes - y_base;
  const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<char*>(p.y) + y_base + (unsigned long long)split * p.y_split_bytes, 0,
      (int)(y_rest < 0x7fffffffull ? y_rest : 0x7fffffffull), 0x00020000);
  auto out_transform = [&](auto&& put) __attribute__((always_inline)) {
#pragma unroll
    for (int th = 0; th < 2; ++th)
#pragma unroll
      for (int ep = 0; ep < 2; ++ep) {
        const int e0 = 2 * ep;
        auto pr = [&](int a) __attribute__((always_inline)) { return f32x2{acc[a][th][e0], acc[a][th][e0 + 1]}; };
        const f32x2 z0 = pk_add(pr(R0), pr(ZZ)), z1 = pk_add(pr(R1), pr(ZZ));
        const f32x2 c0 = pr(C0), c1 = pr(C1);
        const f32x2 o0 = pk_add(pk_add(pr(D00), z0), c0), o1 = pk_add(pk_add(pr(D01), z0), c1);
        const f32x2 o2 = pk_add(pk_add(pr(D10), z1), c0), o3 = pk_add(pk_add(pr(D11), z1), c1);
        const int trow = 2 * th + (kl >> 1), ec = 4 * (kl & 1) + e0;      // tile row, tile column inside the block
        put(o0.x, trow, ec, 0, 0); put(o1.x, trow, ec, 0, 1); put(o2.x, trow, ec, 1, 0); put(o3.x, trow, ec, 1, 1);
        put(o0.y, trow, ec + 1, 0, 0); put(o1.y, trow, ec + 1, 0, 1); put(o2.y, trow, ec + 1, 1, 0); put(o3.y, trow, ec + 1, 1, 1);
      }
  };
  if (p.wide_store) {
    // every wave is behind the loop's last barrier: V is dead.  Per wave [8 block rows][16 px][16 ch] floats = 8 KiB.
    float* const stg = smf + wave * 2048;
    out_transform([&](float v, int trow, int ec, int a, int b) __attribute__((always_inline)) {
      stg[((2 * trow + a) * 16 + 2 * ec + b) * 16 + rl] = v;
    });
    __builtin_amdgcn_wave_barrier();
    const int ox = lane >> 2, nq = lane & 3;             // store lane: block pixel column ox, channel quad nq
    const int ox0 = 2 * cb * BW;
    const bool q_ok = n0 + nq * 4 < p.Cout && ox0 + ox < p.Wo;
    const unsigned voff_q = q_ok ? (unsigned)(((ox0 + ox) * p.ldy + n0 + nq * 4) * 4) : OOBV;
    const int pix_bytes = p.ldy * 4;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      f32x4 v = *reinterpret_cast<const f32x4*>(stg + (i * 16 + ox) * 16 + nq * 4);
      if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      const int trow = i >> 1, a = i & 1;
      const int img_rel = trow < brk ? 0 : 1;
      const int ty = trow < brk ? ty0 + trow : trow - brk;
      const int oy = 2 * ty + a;
      const bool row_ok = vr0 + trow < p.vrows && oy < p.Ho;        // uniform
      const unsigned soff = (unsigned)(((img_rel * p.Ho + oy) * p.Wo) * pix_bytes);
      kfn::buffer_store_b128<KFN_NT_STORE_AUX>(v, rsY, row_ok ? voff_q : OOBV, soff);
    }
  } else {
    const int pix_bytes = p.ldy * 4;
    out_transform([&](float v, int trow, int ec, int a, int b) __attribute__((always_inline)) {
      const int img_rel = trow < brk ? 0 : 1;
      const int ty = trow < brk ? ty0 + trow : trow - brk;
      const int oy = 2 * ty + a;
      v = relu ? fmaxf(v, 0.f) : v;
      const int tx = cb * BW + ec;
      const unsigned soff = (unsigned)(((img_rel * p.Ho + oy) * p.Wo + 2 * tx + b) * pix_bytes);
      const bool ok = n_ok && tx < p.Tw && 2 * tx + b < p.Wo && vr0 + trow < p.vrows && oy < p.Ho;
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsY, ok ? (unsigned)(n * 4) : OOBV, soff, 0);
    });
  }
}

}  // namespace

// Can the polyphase kernel take this layer?  (host-side routing; no device access)
int kfn::wino_s2_lds_bytes(int wino_form, int operand_dtype) {
  if (wino_form == KFN_WINO_FORM_S2_F42) return operand_dtype == KFN_OPERAND_F32 ? kfn::wino_s2c_lds_bytes() : -1;
  if (wino_form == KFN_WINO_FORM_S2_EIGHT_WAVE) return operand_dtype == KFN_OPERAND_F32 ? SB_LDS : -1;
  if (wino_form != KFN_WINO_FORM_AUTO) return -1;
  return operand_dtype == KFN_OPERAND_F16 ? VLayoutS2<true>::LDS : VLayoutS2<false>::LDS;
}

extern "C" int kfn_winograd_s2_supported(const kfn_conv_desc* d) {
  kfn_conv_desc d_full;
  if (!d || kfn::conv_desc_in(d, &d_full, "kfn_winograd_s2_supported", true) != KFN_OK) return 0;
  if ((d_full.x_layout != KFN_LAYOUT_NHWC || d_full.y_layout != KFN_LAYOUT_NHWC) && d_full.wino_form != KFN_WINO_FORM_S2_F42) return 0;
  d = &d_full;
  if (d->wino_form == KFN_WINO_FORM_S2_F42) return kfn::wino_s2c_supported(d);
  if (d->x_dtype != KFN_ACT_F32 || d->y_dtype != KFN_ACT_F32) return 0;
  if (d->kh != 3 || d->kw != 3 || d->stride != 2 || d->transposed) return 0;
  if (d->H <= 0 || d->W <= 0 || (d->H & 1) || (d->W & 1)) return 0;     // 'same' pads after the image only
  if (d->Cin <= 0 || d->Cin % SS_CH != 0) return 0;
  if ((d->H / 2 + 1) / 2 < BH) return 0;   // a 4-row tile block may straddle at most two images
  if (d->epilogue != KFN_EPI_NONE || (d->operand_dtype != KFN_OPERAND_F32 && d->operand_dtype != KFN_OPERAND_F16)) return 0;
  if (d->cout_pad % 32 != 0) return 0;
  return 1;
}

namespace {
// `d` normalised; y / ldy = where the kernel writes (the output tensor, or plane 0 of the split-K workspace); k_split > 1:
// the eight-wave kernel on k_split copies of the tile grid, raw partial sums (the caller passes bias = nullptr, relu = 0)
int s2_launch(const kfn_conv_desc* d, const float* x, const void* u2_packed, const float* bias, float* y, int ldy, int relu,
              int k_split, void* stream) {
  KFN_REQUIRE(d->x_dtype == KFN_ACT_F32 && d->y_dtype == KFN_ACT_F32,
              "kfn_conv2d_winograd_s2: fp32 activations in memory only (x_dtype / y_dtype = KFN_ACT_F16 is implemented by kfn_conv2d_nhwc)");
  KFN_REQUIRE(d->kh == 3 && d->kw == 3 && d->stride == 2 && !d->transposed,
              "kfn_conv2d_winograd_s2: only 3x3 stride-2 SAME convolutions");
  KFN_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0, "kfn_conv2d_winograd_s2: bad shape %dx%dx%d", d->N, d->H, d->W);
  if ((d->H & 1) || (d->W & 1))
    return kfn::fail(KFN_ERR_UNSUPPORTED, "kfn_conv2d_winograd_s2: H=%d W=%d must be even", d->H, d->W);
  if (d->Cin <= 0 || d->Cin % SS_CH != 0)
    return kfn::fail(KFN_ERR_UNSUPPORTED, "kfn_conv2d_winograd_s2: Cin=%d must be a multiple of %d", d->Cin, SS_CH);
  if ((d->H / 2 + 1) / 2 < BH)
    return kfn::fail(KFN_ERR_UNSUPPORTED, "kfn_conv2d_winograd_s2: H=%d is below %d rows", d->H, 4 * BH - 2);
  KFN_REQUIRE(d->ldx >= d->Cin && d->ldx % 4 == 0 && d->Cout > 0 && ldy >= d->Cout &&
                  d->cout_pad >= d->Cout && d->cout_pad % 32 == 0,
              "kfn_conv2d_winograd_s2: bad strides / channel counts");
  KFN_REQUIRE(d->epilogue == KFN_EPI_NONE && (d->operand_dtype == KFN_OPERAND_F32 || d->operand_dtype == KFN_OPERAND_F16),
              "kfn_conv2d_winograd_s2: fp32 or fp16 operands, no fused head epilogue");
  const bool h16 = d->operand_dtype == KFN_OPERAND_F16;
  KFN_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(u2_packed)) & 15) == 0,
              "kfn_conv2d_winograd_s2: buffers must be 16-byte aligned");
  const long img_b = (long)d->H * d->W * d->ldx * 4L;
  const long out_b = (long)(d->H / 2) * (d->W / 2) * ldy * 4L;
  KFN_REQUIRE(2 * img_b < (1L << 31) && 2 * out_b < (1L << 31) && 16L * d->cout_pad * d->Cin * 4L < (1L << 31),
              "kfn_conv2d_winograd_s2: image or kernel beyond 2 GiB of 32-bit offsets");
  WinoS2Args a;
  a.x = x; a.u2 = static_cast<const float*>(u2_packed); a.bias = bias; a.y = y;
  a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.ldx = d->ldx;
  a.Cout = d->Cout; a.cout_pad = d->cout_pad; a.ldy = ldy;
  a.Ho = d->H / 2; a.Wo = d->W / 2;
  a.Th = (a.Ho + 1) / 2; a.Tw = (a.Wo + 1) / 2;
  const long vrows = (long)d->N * a.Th;
  a.vrows = (int)vrows;
  a.bw = kfn::ceil_div(a.Tw, BW);
  const long tiles_m = (long)a.bw * kfn::ceil_div(a.vrows, BH);
  a.tiles_n = kfn::ceil_div(d->cout_pad, NT);
  KFN_REQUIRE(tiles_m * a.tiles_n < (1L << 31), "kfn_conv2d_winograd_s2: grid too large");
  a.tiles_m = (int)tiles_m;
  a.relu = relu;
  {
    // AUTO: two channel groups of a tile block adjacent (round 4, profiles/r04_wino4_microbench.log: conv4a 7.19 ms against
    // 7.34 with all eight adjacent and 7.29 with the tile blocks fastest; conv3a / conv2a within 1 % of each other)
    int ng = a.tiles_n % 2 == 0 ? 2 : (KFN_WINO_DEFAULT_N_FAST ? a.tiles_n : 1);
    if (d->wino_order == KFN_WINO_ORDER_N_FAST) ng = a.tiles_n;
    else if (d->wino_order == KFN_WINO_ORDER_M_FAST) ng = 1;
    else if (d->wino_order >= KFN_WINO_ORDER_GROUPS(1)) ng = d->wino_order - KFN_WINO_ORDER_GROUPS(0);
    if (ng < 1 || ng > a.tiles_n || a.tiles_n % ng != 0) ng = KFN_WINO_DEFAULT_N_FAST ? a.tiles_n : 1;
    a.n_group = ng;
  }
  a.wide_store = (d->Cout % 4 == 0 && ldy % 4 == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0) ? 1 : 0;
  const long in_pix = (long)d->N * d->H * d->W, out_pix = (long)d->N * a.Ho * a.Wo;
  a.x_bytes = (unsigned long long)(((in_pix - 1) * d->ldx + d->Cin) * 4L);
  a.y_bytes = (unsigned long long)(((out_pix - 1) * ldy + d->Cout) * 4L);
  a.u_bytes = (unsigned)(16L * d->cout_pad * d->Cin * (h16 ? 2L : 4L));
  a.k_split = 1;
  a.ss_per_split = d->Cin / SS_CH;
  a.y_split_bytes = 0;
  if (k_split > 1) {
    const int n_super = d->Cin / SS_CH;
    a.k_split = k_split;
    a.ss_per_split = kfn::ceil_div(n_super, k_split);
    KFN_REQUIRE((long)(k_split - 1) * a.ss_per_split < n_super, "kfn_conv2d_winograd_s2_splitk: k_split=%d leaves an empty split of %d super-steps",
                k_split, n_super);
    a.y_split_bytes = (unsigned long long)out_pix * d->Cout * 4ull;
    KFN_REQUIRE((long)a.tiles_m * a.tiles_n * k_split < (1L << 31), "kfn_conv2d_winograd_s2_splitk: grid too large");
  }
  const dim3 grid((unsigned)((long)a.tiles_m * a.tiles_n * a.k_split)), block(64 * NWAVE);
  if (d->wino_form == KFN_WINO_FORM_S2_EIGHT_WAVE || k_split > 1) {
    // the eight-wave form: fp32 operands only, weights packed per pair of fragments (graph.pack_winograd_s2_kernel_b)
    if (h16) return kfn::fail(KFN_ERR_UNSUPPORTED, "kfn_conv2d_winograd_s2: the eight-wave form takes fp32 operands only");
    static std::atomic<uint64_t> attr_done_b{0};
    int rc = kfn::set_max_dynamic_lds(reinterpret_cast<const void*>(wino_s2b_kernel), SB_LDS, attr_done_b);
    if (rc != KFN_OK) return rc;
    hipLaunchKernelGGL(wino_s2b_kernel, grid, dim3(512), SB_LDS, (hipStream_t)stream, a);
    KFN_LAUNCH_CHECK("wino_s2b_kernel");
    return KFN_OK;
  }
  if (h16) {
    static std::atomic<uint64_t> attr_done16{0};
    int rc = kfn::set_max_dynamic_lds(reinterpret_cast<const void*>(wino_s2_kernel<true>), VLayoutS2<true>::LDS, attr_done16);
    if (rc != KFN_OK) return rc;
    hipLaunchKernelGGL(wino_s2_kernel<true>, grid, block, VLayoutS2<true>::LDS, (hipStream_t)stream, a);
  } else {
    static std::atomic<uint64_t> attr_done{0};
    int rc = kfn::set_max_dynamic_lds(reinterpret_cast<const void*>(wino_s2_kernel<false>), VLayoutS2<false>::LDS, attr_done);
    if (rc != KFN_OK) return rc;
    hipLaunchKernelGGL(wino_s2_kernel<false>, grid, block, VLayoutS2<false>::LDS, (hipStream_t)stream, a);
  }
  KFN_LAUNCH_CHECK("wino_s2_kernel");
  return KFN_OK;
}
}  // namespace

extern "C" int kfn_conv2d_winograd_s2(const kfn_conv_desc* d, const float* x, const void* u2_packed, const float* bias,
                                      float* y, void* stream) {
  KFN_REQUIRE(d && x && u2_packed && y, "kfn_conv2d_winograd_s2: null argument");
  KFN_CONV_DESC_IN_LAYOUTS(d, "kfn_conv2d_winograd_s2");
  if (d->wino_form == KFN_WINO_FORM_S2_F42) return kfn::launch_wino_s2c(d, x, u2_packed, bias, y, stream);   // kfn_wino_s2c.hip
  if (d->x_layout != KFN_LAYOUT_NHWC || d->y_layout != KFN_LAYOUT_NHWC)
    return kfn::fail(KFN_ERR_UNSUPPORTED, "kfn_conv2d_winograd_s2: only the F(4,2) form (KFN_WINO_FORM_S2_F42) takes KFN_LAYOUT_C16 activations");
  return s2_launch(d, x, u2_packed, bias, y, d->ldy, d->relu, 1, stream);
}

// ---- split-K form of the eight-wave kernel (round 5; cf. kfn_conv2d_winograd_f43_splitk): BASELINE configs[1]'s conv4a launches
// 320 workgroups at batch 1 = two rounds on 256 CUs, the second a quarter full.
extern "C" int kfn_winograd_s2_splitk_workspace_bytes(const kfn_conv_desc* d, int k_split, size_t* bytes) {
  KFN_REQUIRE(d && bytes, "kfn_winograd_s2_splitk_workspace_bytes: null argument");
  KFN_CONV_DESC_IN(d, "kfn_winograd_s2_splitk_workspace_bytes");
  KFN_REQUIRE(k_split >= 1 && d->N > 0 && d->H > 1 && d->W > 1 && d->Cout > 0, "kfn_winograd_s2_splitk_workspace_bytes: bad argument");
  *bytes = k_split > 1 ? (size_t)k_split * d->N * (d->H / 2) * (d->W / 2) * d->Cout * sizeof(float) : 0;
  return KFN_OK;
}

extern "C" int kfn_conv2d_winograd_s2_splitk(const kfn_conv_desc* d, const float* x, const void* u2b_packed, const float* bias,
                                             float* y, float* workspace, int k_split, void* stream) {
  KFN_REQUIRE(d && x && u2b_packed && y, "kfn_conv2d_winograd_s2_splitk: null argument");
  KFN_CONV_DESC_IN(d, "kfn_conv2d_winograd_s2_splitk");
  KFN_REQUIRE((d->wino_form == KFN_WINO_FORM_AUTO || d->wino_form == KFN_WINO_FORM_S2_EIGHT_WAVE) && d->operand_dtype == KFN_OPERAND_F32,
              "kfn_conv2d_winograd_s2_splitk: the eight-wave form only (fp32 operands, weights packed per pair of fragments)");
  const int n_super = d->Cin > 0 ? d->Cin / SS_CH : 0;
  KFN_REQUIRE(k_split >= 1 && k_split <= (n_super > 0 ? n_super : 1), "kfn_conv2d_winograd_s2_splitk: k_split=%d outside 1..%d (Cin/16)",
              k_split, n_super);
  kfn_conv_desc d8 = *d;
  d8.wino_form = KFN_WINO_FORM_S2_EIGHT_WAVE;
  if (k_split == 1) return s2_launch(&d8, x, u2b_packed, bias, y, d->ldy, d->relu, 1, stream);
  KFN_REQUIRE(workspace != nullptr && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0 && d->Cout % 4 == 0 && d->ldy >= d->Cout &&
                  d->ldy % 4 == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0,
              "kfn_conv2d_winograd_s2_splitk: k_split > 1 needs a 16-byte aligned workspace of kfn_winograd_s2_splitk_workspace_bytes(), "
              "Cout %% 4 == 0 and a 16-byte aligned output with ldy %% 4 == 0");
  int rc = s2_launch(&d8, x, u2b_packed, nullptr, workspace, d->Cout, 0, k_split, stream);
  if (rc != KFN_OK) return rc;
  const long pixels = (long)d->N * (d->H / 2) * (d->W / 2);
  return kfn::launch_splitk_reduce(workspace, k_split, pixels, bias, d->relu, y, d->Cout, d->ldy, (hipStream_t)stream);
}
