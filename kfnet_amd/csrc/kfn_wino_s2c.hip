// kfn_wino_s2c.hip -- 3x3 STRIDE-2 SAME convolution with 81 instead of 144 multiplies per 4x4 outputs: polyphase + F(4,2).
//
// Reference: tf.layers.conv2d(kernel 3, strides 2, 'same') in cnn_wrapper/network.py:116-135 -- SCoordNet's conv2a / conv3a /
// conv4a (cnn_wrapper/SCoordNet.py:23,25,27).  For an even input TF pads one row / column AFTER the image only:
//   y[i,j] = bias + sum_{a,b in 0..2} w[a,b] x[2i+a, 2j+b],   x = 0 at row H / column W.
// kfn_wino_s2.hip evaluates the four polyphase filters (2x2 on the (even,even) phase, 2x1 on (even,odd), 1x2 on (odd,even),
// 1x1 on (odd,odd), all stride 1) under F(2,2): 25 products per 2x2 outputs = 100 per 16.  That kernel is MFMA-bound (MfmaUtil
// 87 %), so only fewer products help: here the same four filters run under F(4,2) on 4x4 OUTPUT tiles (9x9 input patches):
//   (even,even) 5x5 pixels, 2x2 taps : F(4,2) x F(4,2)   25 products     B^T e B
//   (even,odd)  5x4 pixels, 2x1 taps : F(4,2) along y    20 products     B^T f C^T
//   (odd,even)  4x5 pixels, 1x2 taps : F(4,2) along x    20 products     C g B
//   (odd,odd)   4x4 pixels, 1 tap    :                   16 products     C h C^T
// = 81 per 16 outputs (direct 144, F(2,2) 100).  F(4,2) with the points {0, 1, -1, 2, inf}:
//   B^T = [[2,-1,-2,1,0],[0,2,1,-1,0],[0,-2,3,-1,0],[0,-1,0,1,0],[0,2,-1,-2,1]]     G = [[1/2,0],[1/2,1/2],[1/6,-1/6],[1/6,1/3],[0,1]]
//   A^T = [[1,1,1,1,0],[0,1,-1,2,0],[0,1,1,4,0],[0,1,-1,8,1]]
// Accumulator folding.  All 81 products of a tile accumulate into the 25 accumulators M[xi][nu] of the 2-D F(4,2) form, and the
// output transform is the plain Y = A^T M A: along an axis with ONE tap the four products p_0..p_3 must come out of A^T unchanged,
// i.e. they enter M at the indices {0,1,2,4} as m = C p with A^T[:, {0,1,2,4}] C = I:
//   C = [[1,0,-1,0],[0,1/2,1/2,0],[0,-1/2,1/2,0],[0,-1,0,1]]        (m_3 = 0)
// -- and C, being linear, is applied to the DATA (4 values -> 4 values, 5 operations), so a one-tap axis costs one weight fragment
// for all four positions.  36 distinct weight fragments per 16 input channels (25 + 5 + 5 + 1) for 81 positions.
// (tests/test_oracle_kat.py::test_polyphase_f42_stride2_identity checks the algebra with the packer's fragments on the CPU.)
//
// Kernel.  A workgroup of EIGHT waves (two per SIMD) owns 4 x 4 tiles (16 x 16 output pixels, 33 x 33 input pixels) x 128 output
// channels.  CONSUMER: every wave, 16 output channels x all 16 tiles on v_mfma_f32_16x16x4_f32 (A = V from LDS, B = weights from
// memory): 25 accumulators x 4 registers; per super-step of 16 input channels 81 positions x 4 k-steps = 324 MFMAs in 19 groups of
// 4-5 positions with distinct accumulators (k-step major inside a group).  PRODUCER: wave w transforms ONE phase for 8 tiles (lane =
// tile x channel pair; 8 lanes read 64 contiguous bytes of a pixel): waves 0,1 (even,even) 25 pixels, 2,3 (even,odd) 20, 4,5 (odd,odd)
// 16, 6,7 (odd,even) 20 -- a SIMD's two waves gather 41 / 40 pixels together; the four code paths differ by wave, never inside one.
// LDS.  V of a super-step is 81 slots x [4 k][16 tiles][4 k-steps] floats = 81 KiB: two full buffers exceed the 160 KiB.  The
// slots are laid out in CONSUMPTION order; the 43 consumed in the first half of a super-step are single-buffered, the other 38
// double-buffered: a barrier in the middle of the super-step (behind position 43) frees the first 43 slots for the next super-step's
// values, which the producers store in the second half anyway (gather in the first half, transform, store).  119 KiB.
#include "kfn_common.h"
#include <type_traits>

// Timing-experiment builds (tools/mb/build_s2c.sh; results are WRONG on purpose): bit 0 no mid barrier, 1 no gathers, 2 no
// transform, 3 no V stores, 4 no weight loads in the loop, 5 no V reads in the loop, 6 no output stores, 7 no epilogue at all
#ifndef KFN_S2C_EXP
#define KFN_S2C_EXP 0
#endif
// 1: the gathers of super-step s+2 are issued in super-step s right behind the V stores of s+1 (the patch registers are free
// from there on): more than a super-step between a gather and the transform that consumes it.  0: gathered at the start of s+1.
#ifndef KFN_S2C_LATE_GATHER
#define KFN_S2C_LATE_GATHER 1
#endif
// 1: launches of at least two workgroups per CU run the persistent form (wino_s2c_pkernel, below); 0: one workgroup per tile block
#ifndef KFN_S2C_PERSIST
#define KFN_S2C_PERSIST 1
#endif
#ifndef KFN_S2C_PERSIST_MAX_SS
#define KFN_S2C_PERSIST_MAX_SS 64         // ... of layers with at most this many super-steps
#endif
#ifndef KFN_S2C_XSLOT_B
#define KFN_S2C_XSLOT_B 24
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr unsigned OOBV = 0x80000000u;
constexpr int NT = 128;                 // output channels per workgroup
constexpr int NPOS = 81, NFRAG = 36, NACC = 25, NGROUP = 19;
constexpr int SS_CH = 16;               // input channels per super-step
constexpr int NS = 43;                  // positions (= slots in consumption order) of the first half: single-buffered
constexpr int ND = NPOS - NS;           // 38 double-buffered
constexpr int SLOT_F = 256;             // floats per slot
constexpr int LDS_V = (NS + 2 * ND) * SLOT_F * 4;          // 121 856 B
constexpr int STG_ROW = 4 * 16 * 16 + 16;                  // floats per tile row of the per-wave output image (+64 B skew)
constexpr int LDS_OUT = 8 * 4 * STG_ROW * 4;               // 133 120 B
constexpr int LDS_BYTES = LDS_V > LDS_OUT ? LDS_V : LDS_OUT;
constexpr int JMID = 4 * NS;            // MFMA slot of the mid barrier (172)

// groups in consumption order: round r = EE_r (5 positions) EO_r (4) OE_r (4) OO_r (4, r < 4)
constexpr int G_NPOS[NGROUP] = {5, 4, 4, 4, 5, 4, 4, 4, 5, 4, 4, 4, 5, 4, 4, 4, 5, 4, 4};
constexpr int G_POS0[NGROUP] = {0, 5, 9, 13, 17, 22, 26, 30, 34, 39, 43, 47, 51, 56, 60, 64, 68, 73, 77};
// per position (consumption order): weight register (0-4 the (even,even) fragment of column nu, 5 (even,odd), 6 (odd,even), 7 the
// centre tap), accumulator 5 xi + nu, fragment in memory
constexpr int POS_BREG[NPOS] = {0, 1, 2, 3, 4, 5, 5, 5, 5, 6, 6, 6, 6, 7, 7, 7, 7, 0, 1, 2, 3, 4, 5, 5, 5, 5, 6, 6, 6, 6, 7, 7, 7, 7, 0, 1, 2, 3, 4, 5, 5,
                               5, 5, 6, 6, 6, 6, 7, 7, 7, 7, 0, 1, 2, 3, 4, 5, 5, 5, 5, 6, 6, 6, 6, 7, 7, 7, 7, 0, 1, 2, 3, 4, 5, 5, 5, 5, 6, 6, 6, 6};
constexpr int POS_ACC[NPOS] = {0, 1, 2, 3, 4, 0, 1, 2, 4, 0, 5, 10, 20, 0, 1, 2, 4, 5, 6, 7, 8, 9, 5, 6, 7, 9, 1, 6, 11, 21, 5, 6, 7, 9, 10, 11, 12, 13, 14,
                              10, 11, 12, 14, 2, 7, 12, 22, 10, 11, 12, 14, 15, 16, 17, 18, 19, 15, 16, 17, 19, 3, 8, 13, 23, 20, 21, 22, 24, 20, 21, 22,
                              23, 24, 20, 21, 22, 24, 4, 9, 14, 24};
constexpr int POS_FRAG[NPOS] = {0, 1, 2, 3, 4, 25, 25, 25, 25, 30, 30, 30, 30, 35, 35, 35, 35, 5, 6, 7, 8, 9, 26, 26, 26, 26, 31, 31, 31, 31, 35, 35, 35,
                               35, 10, 11, 12, 13, 14, 27, 27, 27, 27, 32, 32, 32, 32, 35, 35, 35, 35, 15, 16, 17, 18, 19, 28, 28, 28, 28, 33, 33, 33,
                               33, 35, 35, 35, 35, 20, 21, 22, 23, 24, 29, 29, 29, 29, 34, 34, 34, 34};
// LDS index (= consumption rank) of transform slot s: slots 0-24 (even,even) [xi][nu], 25-44 (even,odd) [xi][nu'], 45-64 (odd,even)
// [xi'][nu], 65-80 (odd,odd) [xi'][nu'] (primed indices run over the accumulator indices {0,1,2,4})
constexpr int SLOT_IDX[NPOS] = {0, 1, 2, 3, 4, 17, 18, 19, 20, 21, 34, 35, 36, 37, 38, 51, 52, 53, 54, 55, 68, 69, 70, 71, 72, 5, 6, 7, 8, 22, 23, 24, 25,
                               39, 40, 41, 42, 56, 57, 58, 59, 73, 74, 75, 76, 9, 26, 43, 60, 77, 10, 27, 44, 61, 78, 11, 28, 45, 62, 79, 12, 29, 46,
                               63, 80, 13, 14, 15, 16, 30, 31, 32, 33, 47, 48, 49, 50, 64, 65, 66, 67};
constexpr int group_of_pos(int pp) {
  int g = 0;
  while (g + 1 < NGROUP && G_POS0[g + 1] <= pp) ++g;
  return g;
}
// is position pp the last of the super-step that uses its weight register's current fragment?
constexpr bool frag_last_user(int pp) {
  for (int q = pp + 1; q < NPOS; ++q)
    if (POS_FRAG[q] == POS_FRAG[pp]) return false;
  return true;
}
// the fragment that follows POS_FRAG[pp] in register POS_BREG[pp]: (fragment, 0) in this super-step or (fragment, 1) in the next
constexpr int next_frag(int pp) {
  for (int q = pp + 1; q < NPOS; ++q)
    if (POS_BREG[q] == POS_BREG[pp] && POS_FRAG[q] != POS_FRAG[pp]) return POS_FRAG[q];
  for (int q = 0; q < NPOS; ++q)
    if (POS_BREG[q] == POS_BREG[pp]) return POS_FRAG[q] + 64;     // + 64: of the NEXT super-step
  return -1;
}
// phases: pixel counts, first transform slot
constexpr int PART_NPX[4] = {25, 20, 20, 16};      // A (even,even)  B (even,odd)  C (odd,even)  D (odd,odd)
constexpr int PART_SLOT0[4] = {0, 25, 45, 65};

struct S2cArgs {
  const float* x;
  const float* u;     // [Cin/16][36][cout_pad][16]
  const float* bias;
  float* y;
  int N, H, W, Cin, ldx;
  int Cout, cout_pad, ldy;
  int Ho, Wo;
  int Th, Tw;         // output tiles (4x4) per image
  int vrows;          // N * Th
  int bw;             // tile-block columns
  int tiles_m, tiles_n;
  int relu;
  int n_group;
  unsigned long long x_bytes, y_bytes;
  unsigned u_bytes;
  // byte strides of the activation layouts (kfn_conv_desc.x_layout / y_layout): element (n, h, w, c) at n * img + (h * W + w) * pix +
  // (c >> 4) * cb + (c & 15) * 4 -- NHWC: pix = ld * 4, cb = 64; KFN_LAYOUT_C16: pix = 64, cb = H * W * 64
  unsigned x_pix, x_cb, x_img;
  unsigned y_pix, y_cb, y_img;
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

__device__ __forceinline__ int xcd_remap_c(int b, int nwg) {
  int xcd = b & 7;
  int q = nwg >> 3, r = nwg & 7;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (b >> 3);
}

// (d0..d4) -> B^T d for F(4,2), points {0,1,-1,2,inf}: 9 operations on a channel pair (hipcc emits v_pk_add_f32 / v_pk_fma_f32; the
// same transform on scalar instructions, which MI355X_MICROARCH.md prices lower beside MFMAs of ONE wave per SIMD, measured 0.5 %
// slower here with two waves per SIMD)
__device__ __forceinline__ void bt5(f32x2& d0, f32x2& d1, f32x2& d2, f32x2& d3, f32x2& d4) {
  const f32x2 t = d3 - d1;
  const f32x2 q = d2 - d1;
  const f32x2 a = d0 - d2;
  const f32x2 b = d4 - d2;
  const f32x2 s = d1 + d2;
  d0 = a * 2.0f + t;
  d1 = s - t;
  d2 = q * 3.0f - t;
  d3 = t;
  d4 = b - t * 2.0f;
}
// (e0..e3) -> C e: the one-tap axis, into the accumulator indices {0,1,2,4}: 5 operations
__device__ __forceinline__ void ct4(f32x2& e0, f32x2& e1, f32x2& e2, f32x2& e3) {
  const f32x2 h = e2 * 0.5f;
  const f32x2 m0 = e0 - e2;
  const f32x2 m3 = e3 - e1;
  const f32x2 m1 = e1 * 0.5f + h;
  const f32x2 m2 = h - e1 * 0.5f;
  e0 = m0; e1 = m1; e2 = m2; e3 = m3;
}

__global__ __launch_bounds__(512, 1) void wino_s2c_kernel(S2cArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem_c[];
  float* const smf = reinterpret_cast<float*>(smem_c);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int nwg = p.tiles_m * p.tiles_n;
  const int tile = xcd_remap_c((int)blockIdx.x, nwg);
  const int per = p.tiles_m * p.n_group;
  const int gset = tile / per, rem_ = tile - gset * per;
  const int tm = rem_ / p.n_group;
  const int tn = gset * p.n_group + (rem_ - tm * p.n_group);
  const int cb = tm % p.bw, rb = tm / p.bw;
  const int n0 = tn * NT + wave * 16;

  // block geometry (uniform): virtual tile rows run over the batch image after image
  const int vr0 = rb * 4;
  const int img0 = vr0 / p.Th;
  const int ty0 = vr0 - img0 * p.Th;
  const int brk = (p.Th - ty0 < 4) ? (p.Th - ty0) : 4;        // tile rows >= brk belong to image img0 + 1
  const unsigned long long a_base = (unsigned long long)img0 * p.x_img;
  const unsigned long long a_rest = p.x_bytes - a_base;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(p.x)) + a_base, 0,
      (int)(a_rest < 0x7fffffffull ? a_rest : 0x7fffffffull), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsU = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.u), 0, p.u_bytes, 0x00020000);

  // ---- PRODUCER: phase `part` of tile t = 8 (wave & 1) + (lane >> 3), channel pair pq8 = lane & 7 of the super-step's 16 ----
  const int part = wave < 2 ? 0 : wave < 4 ? 1 : wave < 6 ? 3 : 2;          // (uniform)  SIMD i runs waves i and i + 4
  const int pt = 8 * (wave & 1) + (lane >> 3), pq8 = lane & 7;
  const int ptr_ = pt >> 2, ptc = pt & 3;
  // H and W are multiples of 8: a tile's 9x9 patch leaves the image only with its row 8 (last tile row) / column 8 (last tile
  // column); four per-lane bases by (row 8?, column 8?), OOB (-> 0) where the pixel does not exist
  unsigned gbase[2][2];
  {
    const int img_rel = ptr_ < brk ? 0 : 1;
    const int ty = ptr_ < brk ? ty0 + ptr_ : ptr_ - brk;
    const int tx = cb * 4 + ptc;
    const bool tile_ok = (vr0 + ptr_ < p.vrows) && tx < p.Tw;
    const unsigned base = (unsigned)img_rel * p.x_img + (unsigned)(8 * ty * p.W + 8 * tx) * p.x_pix + (unsigned)(pq8 * 8);
    const bool r8 = 8 * ty + 8 < p.H, c8 = 8 * tx + 8 < p.W;
    gbase[0][0] = tile_ok ? base : OOBV;
    gbase[1][0] = tile_ok && r8 ? base : OOBV;
    gbase[0][1] = tile_ok && c8 ? base : OOBV;
    gbase[1][1] = tile_ok && r8 && c8 ? base : OOBV;
  }
  const unsigned row_b = (unsigned)p.W * p.x_pix, pix_b = p.x_pix;
  // slot layout [4 k][16 rows][4 k-steps] floats: channel 2 pq8 + e = 4 k + s, k = pq8 >> 1, s = 2 (pq8 & 1) + e; tile t is stored
  // in row t ^ 2k.  Reads: ds_read_b128 at (lane's k, row) -- per hardware lane group 16 lanes on 256 different bytes of a 256-byte
  // window (MI355X_MICROARCH.md, LDS: 4 x 16 lanes, 64 banks).  Stores: ds_write_b64 is served in groups of 16 CONTIGUOUS lanes (two
  // tiles x 8 channel pairs) on 32 banks: the XOR puts the four k of a group on four different 32-byte chunks of the 128-byte window.
  const int pk = pq8 >> 1, ph = pq8 & 1;
  const int v_st = (pk * 16 + (pt ^ (2 * pk))) * 4 + 2 * ph;                                 // floats inside a slot
  const int n_super = p.Cin / SS_CH;

  // ---- CONSUMER: lane (tile row r of the A operand / channel n0 + r of the B operand, k) ----
  const int rl = lane & 15, kl = lane >> 4;
  const int v_rd = (kl * 16 + (rl ^ (2 * kl))) * 4;                                        // floats inside a slot
  // (channels past cout_pad -- a channel block of 128 that the packed weights fill only partly -- re-read the last packed channel:
  //  their results are never stored, and no lane reaches beyond the last fragment)
  const int nb = n0 + rl < p.cout_pad ? n0 + rl : p.cout_pad - 1;
  const unsigned voff_b = (unsigned)((nb * 16 + kl * 4) * 4);
  const unsigned b_step = (unsigned)p.cout_pad * 64u;                                       // bytes between fragments
  const int n = n0 + rl;
  const bool n_ok = n < p.Cout;
  const float bv = (p.bias != nullptr && n_ok) ? p.bias[n] : 0.f;

  f32x4 acc[NACC];
#pragma unroll
  for (int g = 0; g < NACC; ++g) {
    const float v0 = g == 6 ? bv : 0.f;        // A^T e_1 = (1,1,1,1): M[1][1] = b gives every output + b
    acc[g] = f32x4{v0, v0, v0, v0};
  }
  f32x2 pv[25];        // producer: up to 25 patch pixels x 2 channels
  f32x4 bq[8];         // weight fragments: 5 (even,even) columns, (even,odd), (odd,even), centre
  f32x4 vq[2][5];      // V fragments (4 k-steps): [group parity][position of the group]

  // register r <- fragment index fq (over all super-steps).  The callers clamp the SUPER-STEP of a prefetch to the last one (once
  // per super-step, scalar): a prefetch past the last super-step re-reads valid memory (nobody consumes it).  (Not needed for
  // safety: on gfx950 the range check of a raw descriptor counts voffset + soffset -- tools/mb/srd_probe.hip -- so the unclamped
  // prefetch would come back as zeros; it was written before that was measured and costs two scalar instructions per super-step.)
  auto b_load = [&](auto rc, int fq) __attribute__((always_inline)) {
    constexpr int r = decltype(rc)::value;
    bq[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsU, voff_b, (unsigned)fq * b_step, 0));
  };

  auto run = [&](auto part_c) __attribute__((always_inline)) {
    constexpr int PART = decltype(part_c)::value;
    constexpr int NPX = PART_NPX[PART];
    constexpr int PW = (PART == 0 || PART == 2) ? 5 : 4;                 // pixels per patch row of this phase
    constexpr int PH = NPX / PW;
    // patch pixel i = (m, n) = (i / PW, i % PW) -> patch coordinates (u, v)
    auto p_gather = [&](auto ic, int ss) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      constexpr int m = i / PW, nn = i % PW;
      constexpr int u = 2 * m + ((PART == 2 || PART == 3) ? 1 : 0), v = 2 * nn + ((PART == 1 || PART == 3) ? 1 : 0);
      // (ss is clamped by the caller, see b_load)
      const unsigned so = (unsigned)u * row_b + (unsigned)v * pix_b + (unsigned)ss * p.x_cb;
      pv[i] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsA, gbase[u == 8][v == 8], so, 0));
    };
    // the transform, one 1-D line per call (NLINE lines): first the lines along n (patch rows), then along m
    constexpr int NLINE = PH + PW;
    auto p_line = [&](auto lc) __attribute__((always_inline)) {
      constexpr int l = decltype(lc)::value;
      if constexpr (l < PH) {                 // row m = l: along n
        constexpr int m = l;
        if constexpr (PW == 5) bt5(pv[m * 5], pv[m * 5 + 1], pv[m * 5 + 2], pv[m * 5 + 3], pv[m * 5 + 4]);
        else ct4(pv[m * 4], pv[m * 4 + 1], pv[m * 4 + 2], pv[m * 4 + 3]);
      } else {                                // column n = l - PH: along m
        constexpr int nn = l - PH;
        if constexpr (PH == 5) bt5(pv[nn], pv[PW + nn], pv[2 * PW + nn], pv[3 * PW + nn], pv[4 * PW + nn]);
        else ct4(pv[nn], pv[PW + nn], pv[2 * PW + nn], pv[3 * PW + nn]);
      }
    };
    // (the double-buffered region lies beyond the 64 KiB a DS instruction's offset field reaches: its lane addresses are formed ONCE
    // per super-step -- stD / rdD -- and every access is base + immediate)
    auto p_store = [&](auto ic, float* stD) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      constexpr int idx = SLOT_IDX[PART_SLOT0[PART] + i];
      if constexpr (idx < NS) *reinterpret_cast<f32x2*>(smf + idx * SLOT_F + v_st) = pv[i];
      else *reinterpret_cast<f32x2*>(stD + (idx - NS) * SLOT_F) = pv[i];
    };
    // V fragment of position pp
    auto v_read = [&](auto pc, const float* rdD) __attribute__((always_inline)) {
      constexpr int pp = decltype(pc)::value;
      constexpr int g = group_of_pos(pp), k = pp - G_POS0[g];
      if constexpr (pp < NS) vq[g & 1][k] = *reinterpret_cast<const f32x4*>(smf + pp * SLOT_F + v_rd);
      else vq[g & 1][k] = *reinterpret_cast<const f32x4*>(rdD + (pp - NS) * SLOT_F);
    };

    // ---- prologue: super-step 0 ----
    sfor<NPX>([&](auto ic) { p_gather(ic, 0); });
    sfor<8>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
      constexpr int f0 = r < 5 ? r : r == 5 ? 25 : r == 6 ? 30 : 35;
      b_load(rc, f0);
    });
    sfor<NLINE>([&](auto lc) { p_line(lc); });
    sfor<NPX>([&](auto ic) { p_store(ic, smf + NS * SLOT_F + v_st); });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const int s_last = n_super - 1;
    if constexpr (KFN_S2C_LATE_GATHER) sfor<NPX>([&](auto ic) { p_gather(ic, s_last < 1 ? s_last : 1); });

    // producer timetable inside a super-step (MFMA slots): gathers from slot 0 every GSTEP, transform lines from XSLOT every
    // XSTEP, [mid barrier at JMID], stores from SSLOT every SSTEP
    // (the two waves of a SIMD -- w and w + 4: phases A+D, B+C -- run their transform bursts at different times)
    constexpr int GSTEP = 4, XSLOT = (KFN_S2C_LATE_GATHER && PART >= 2) ? KFN_S2C_XSLOT_B : 112, XSTEP = 4, SSLOT = JMID + 4 + (PART >= 2 ? 2 : 0), SSTEP = 4;
    static_assert((KFN_S2C_LATE_GATHER || GSTEP * 25 <= XSLOT) && XSLOT + XSTEP * 10 <= JMID && SSLOT + 2 + SSTEP * 25 <= 4 * NPOS, "producer timetable");
    for (int ks = 0; ks < n_super; ++ks) {
      const int nxt = ks + 1;
      const int ks1 = nxt < s_last ? nxt : s_last, ks2 = nxt + 1 < s_last ? nxt + 1 : s_last;      // clamped look-ahead super-steps
      const float* const rdD = smf + (NS + (ks & 1) * ND) * SLOT_F + v_rd;
      float* const stD = smf + (NS + (nxt & 1) * ND) * SLOT_F + v_st;
      // the first group: nothing of this super-step could be read before the barrier
      sfor<5>([&](auto kc) { v_read(kc, rdD); });
      sfor<4 * NPOS>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int g = group_of_pos(j / 4);                 // (positions before G_POS0[g] account for 4 slots each)
        constexpr int np = G_NPOS[g], j0 = 4 * G_POS0[g];
        constexpr int kst = (j - j0) / np, k = (j - j0) % np;  // k-step, position inside the group
        constexpr int pp = G_POS0[g] + k;
        if constexpr (j == JMID && !(KFN_S2C_EXP & 1)) {
          // every wave has read the 43 single-buffered slots: they may take the next super-step's values
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
        }
        acc[POS_ACC[pp]] = __builtin_amdgcn_mfma_f32_16x16x4f32(vq[g & 1][k][kst], bq[POS_BREG[pp]][kst], acc[POS_ACC[pp]], 0, 0, 0);
        // V of the NEXT group (the other register set) during this group's k-step 1
        if constexpr (kst == 1 && g + 1 < NGROUP && !(KFN_S2C_EXP & 32)) {
          constexpr int np1 = G_NPOS[g + 1];
          if constexpr (k < np1) v_read(std::integral_constant<int, G_POS0[g + 1] + k>{}, rdD);
          if constexpr (k == np - 1 && np1 > np) v_read(std::integral_constant<int, G_POS0[g + 1] + np>{}, rdD);
        }
        // weights: the register's next fragment once this one has had its last k-step
        if constexpr (kst == 3 && frag_last_user(pp) && !(KFN_S2C_EXP & 16)) {
          constexpr int nf = next_frag(pp);
          b_load(std::integral_constant<int, POS_BREG[pp]>{}, (nf >= 64 ? ks1 : ks) * NFRAG + (nf & 63));
        }
        if constexpr (!KFN_S2C_LATE_GATHER && j < NPX * GSTEP && j % GSTEP == 0 && !(KFN_S2C_EXP & 2)) p_gather(std::integral_constant<int, j / GSTEP>{}, ks1);
        if constexpr (KFN_S2C_LATE_GATHER && j >= SSLOT + 2 && j < SSLOT + 2 + NPX * SSTEP && (j - SSLOT - 2) % SSTEP == 0 && !(KFN_S2C_EXP & 2))
          p_gather(std::integral_constant<int, (j - SSLOT - 2) / SSTEP>{}, ks2);
        if constexpr (j >= XSLOT && j < XSLOT + NLINE * XSTEP && (j - XSLOT) % XSTEP == 0 && !(KFN_S2C_EXP & 4)) p_line(std::integral_constant<int, (j - XSLOT) / XSTEP>{});
        if constexpr (j >= SSLOT && j < SSLOT + NPX * SSTEP && (j - SSLOT) % SSTEP == 0 && !(KFN_S2C_EXP & 8)) p_store(std::integral_constant<int, (j - SSLOT) / SSTEP>{}, stD);
        __builtin_amdgcn_sched_barrier(0);
      });
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
  };
  if (part == 0) run(std::integral_constant<int, 0>{});
  else if (part == 1) run(std::integral_constant<int, 1>{});
  else if (part == 2) run(std::integral_constant<int, 2>{});
  else run(std::integral_constant<int, 3>{});

  // ---- epilogue: Y = A^T M A per tile.  Lane (channel n0 + rl, k = kl) holds, in element e of every accumulator, tile 4 kl + e =
  // (tile row kl, tile column e).  Per wave an output image [4 tile rows][4 rows][16 px][16 ch] (+64 B skew per tile row: the four
  // k groups of a store hit different banks), then 16-byte stores of 64-byte runs. ----
  const bool relu = p.relu != 0;
  const unsigned long long y_base = (unsigned long long)img0 * p.y_img;
  const unsigned long long y_rest = p.y_bytes - y_base;
  const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<char*>(p.y) + y_base, 0, (int)(y_rest < 0x7fffffffull ? y_rest : 0x7fffffffull), 0x00020000);
  float* const stg = smf + wave * (4 * STG_ROW);
  // 1-D output transform: (m0..m4) -> (y0..y3)
  auto at4 = [](float m0, float m1, float m2, float m3, float m4, float& y0, float& y1, float& y2, float& y3) __attribute__((always_inline)) {
    const float s = m1 + m2, d = m1 - m2;
    y0 = (m0 + s) + m3;
    y1 = d + 2.0f * m3;
    y2 = s + 4.0f * m3;
    y3 = (d + 8.0f * m3) + m4;
  };
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float t[4][5];       // rows: A^T M
#pragma unroll
    for (int nu = 0; nu < 5; ++nu)
      at4(acc[nu][e], acc[5 + nu][e], acc[10 + nu][e], acc[15 + nu][e], acc[20 + nu][e], t[0][nu], t[1][nu], t[2][nu], t[3][nu]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float y0, y1, y2, y3;
      at4(t[i][0], t[i][1], t[i][2], t[i][3], t[i][4], y0, y1, y2, y3);
      float* const row = stg + kl * STG_ROW + (i * 16 + 4 * e) * 16 + rl;
      row[0] = y0; row[16] = y1; row[32] = y2; row[48] = y3;
    }
  }
  __builtin_amdgcn_wave_barrier();
  const int ox = lane >> 2, nq = lane & 3;             // store lane: block pixel column ox, channel quad nq
  const int ox0 = 16 * cb;
  const bool q_ok = n0 + nq * 4 < p.Cout && ox0 + ox < p.Wo;
  const unsigned voff_q = q_ok ? (unsigned)(ox0 + ox) * p.y_pix + (unsigned)(n0 >> 4) * p.y_cb + (unsigned)(nq * 16) : OOBV;   // (n0 is a multiple of 16)
  const unsigned pix_bytes = p.y_pix;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int trow = i >> 2, a = i & 3;
    f32x4 v = *reinterpret_cast<const f32x4*>(stg + trow * STG_ROW + (a * 16 + ox) * 16 + nq * 4);
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    const int img_rel = trow < brk ? 0 : 1;
    const int ty = trow < brk ? ty0 + trow : trow - brk;
    const int oy = 4 * ty + a;
    const bool row_ok = vr0 + trow < p.vrows && oy < p.Ho;        // uniform
    const unsigned soff = (unsigned)img_rel * p.y_img + (unsigned)(oy * p.Wo) * pix_bytes;
    if (!(KFN_S2C_EXP & 64)) kfn::buffer_store_b128<KFN_NT_STORE_AUX>(v, rsY, row_ok ? voff_q : OOBV, soff);
  }
}


// =====================================================================================================================
// PERSISTENT form (round 6, second half).  Measured on the kernel above (tools/mb_s2.py, MB_LAYERS with Cin = 64 .. 512 on conv2a's
// shape): the time per workgroup is 9.8 us per super-step + an intercept of 18-24 us -- launch, the exposed first gathers and
// transform, the epilogue with every CU of the chip storing at the same moment -- i.e. 30-40 % of conv2a (4 super-steps), 11 % of
// conv3a, 5 % of conv4a.  Here a workgroup walks a SEQUENCE of tile blocks (its XCD's share, strided by the XCD's workgroups) and
// the producers never stop: the gathers of super-step s + 2, the weight prefetches of s + 1 and the V stores of s + 1 simply run
// on into the NEXT block's first super-steps (the V slots do not care which block they serve), so a block boundary costs the
// consumers their output transform and stores -- through a small per-wave staging area BESIDE the V buffers (one tile column per
// pass: 4.25 KiB per wave) -- and nothing else: no launch, no prologue, no barrier of its own.
// This lane's index from the hardware, opaque to the optimiser: block-level code of the persistent kernel derives everything it
// needs per lane from it on the spot.  (Hoisted out of the block loop those values do not fit beside the super-step's 248
// registers; hipcc spills them, and every reload sits in the vector-memory queue BEHIND the gathers and weight prefetches that
// are in flight across the block boundary -- waiting for it drains them.)
__device__ __forceinline__ int lane_now() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}
constexpr int PSTG = 4 * (16 * 16 + 16);                      // floats per wave: [4 tile rows][16 px + skew][16 ch]
constexpr int LDS_P = LDS_V + 8 * PSTG * 4;                   // 156 672 B

__global__ __launch_bounds__(512, 1) void wino_s2c_pkernel(S2cArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem_c[];
  float* const smf = reinterpret_cast<float*>(smem_c);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // this workgroup's tiles: XCD x = blockIdx & 7 owns the tiles [xbase, xend) (as xcd_remap_c deals them); its workgroup j =
  // blockIdx >> 3 takes xbase + j, + wpx, + 2 wpx, ... (wpx = workgroups of the grid on that XCD): at any moment the XCD's
  // workgroups are on neighbouring tiles and, with n_group channel groups adjacent, mostly on the same weights
  const int nwg = p.tiles_m * p.tiles_n;
  const int xcd = (int)blockIdx.x & 7;
  const int q8 = nwg >> 3, r8 = nwg & 7;
  const int xbase = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int xend = xbase + q8 + (xcd < r8 ? 1 : 0);
  const int wpx = ((int)gridDim.x >> 3) + (xcd < ((int)gridDim.x & 7) ? 1 : 0);
  int L = xbase + ((int)blockIdx.x >> 3);
  if (L >= xend) return;                                       // (uniform; cannot happen with gridDim <= nwg)
  const int per = p.tiles_m * p.n_group;
  const int n_super = p.Cin / SS_CH;
  const int s_last = n_super - 1;

  struct Blk { int cb, vr0, img0, ty0, brk, tn; };
  auto blk_of = [&](int tile) __attribute__((always_inline)) {
    Blk b;
    const int gset = tile / per, rem_ = tile - gset * per;
    const int tm = rem_ / p.n_group;
    b.tn = gset * p.n_group + (rem_ - tm * p.n_group);
    b.cb = tm % p.bw;
    b.vr0 = (tm / p.bw) * 4;
    b.img0 = b.vr0 / p.Th;
    b.ty0 = b.vr0 - b.img0 * p.Th;
    b.brk = (p.Th - b.ty0 < 4) ? (p.Th - b.ty0) : 4;
    return b;
  };
  const __amdgpu_buffer_rsrc_t rsU = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.u), 0, p.u_bytes, 0x00020000);

  // ---- PRODUCER (see the kernel above): phase `part`, tile pt, channel pair pq8; its geometry follows the block being GATHERED ----
  const int part = wave < 2 ? 0 : wave < 4 ? 1 : wave < 6 ? 3 : 2;
  const int pt = 8 * (wave & 1) + (lane >> 3), pq8 = lane & 7;
  const int ptr_ = pt >> 2, ptc = pt & 3;
  unsigned gbase[2][2];
  __amdgpu_buffer_rsrc_t rsA;
  auto set_producer = [&](const Blk& b) __attribute__((always_inline)) {
    const int lv = lane_now();
    const int pt = 8 * (wave & 1) + (lv >> 3), pq8 = lv & 7;
    const int ptr_ = pt >> 2, ptc = pt & 3;
    const unsigned long long a_base = (unsigned long long)b.img0 * p.x_img;
    const unsigned long long a_rest = p.x_bytes - a_base;
    rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.x)) + a_base, 0,
                                            (int)(a_rest < 0x7fffffffull ? a_rest : 0x7fffffffull), 0x00020000);
    const int img_rel = ptr_ < b.brk ? 0 : 1;
    const int ty = ptr_ < b.brk ? b.ty0 + ptr_ : ptr_ - b.brk;
    const int tx = b.cb * 4 + ptc;
    const bool tile_ok = (b.vr0 + ptr_ < p.vrows) && tx < p.Tw;
    const unsigned base = (unsigned)img_rel * p.x_img + (unsigned)(8 * ty * p.W + 8 * tx) * p.x_pix + (unsigned)(pq8 * 8);
    const bool r8_ = 8 * ty + 8 < p.H, c8_ = 8 * tx + 8 < p.W;
    gbase[0][0] = tile_ok ? base : OOBV;
    gbase[1][0] = tile_ok && r8_ ? base : OOBV;
    gbase[0][1] = tile_ok && c8_ ? base : OOBV;
    gbase[1][1] = tile_ok && r8_ && c8_ ? base : OOBV;
  };
  const unsigned row_b = (unsigned)p.W * p.x_pix, pix_b = p.x_pix;
  const int pk = pq8 >> 1, ph = pq8 & 1;
  const int v_st = (pk * 16 + (pt ^ (2 * pk))) * 4 + 2 * ph;

  // ---- CONSUMER ----
  const int rl = lane & 15, kl = lane >> 4;
  const int v_rd = (kl * 16 + (rl ^ (2 * kl))) * 4;
  const unsigned b_step = (unsigned)p.cout_pad * 64u;
  auto voff_of = [&](int tn) __attribute__((always_inline)) {       // this lane's offset inside a weight fragment of channel group tn
    const int lv = lane_now();
    const int rl = lv & 15, kl = lv >> 4;
    const int nn = tn * NT + wave * 16 + rl;
    const int nb = nn < p.cout_pad ? nn : p.cout_pad - 1;
    return (unsigned)((nb * 16 + kl * 4) * 4);
  };
  f32x4 acc[NACC];
  f32x2 pv[25];
  f32x4 bq[8];
  f32x4 vq[2][5];
  auto acc_init = [&](int tn) __attribute__((always_inline)) {
    const int n = tn * NT + wave * 16 + (lane_now() & 15);
    const float bv = (p.bias != nullptr && n < p.Cout) ? p.bias[n] : 0.f;
#pragma unroll
    for (int g = 0; g < NACC; ++g) {
      const float v0 = g == 6 ? bv : 0.f;
      acc[g] = f32x4{v0, v0, v0, v0};
    }
  };
  auto b_load = [&](auto rc, int fq, unsigned voff) __attribute__((always_inline)) {
    constexpr int r = decltype(rc)::value;
    bq[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsU, voff, (unsigned)fq * b_step, 0));
  };
  auto p_gather = [&](auto part_c, auto ic, int ss) __attribute__((always_inline)) {
    constexpr int PART = decltype(part_c)::value, i = decltype(ic)::value;
    constexpr int PW = (PART == 0 || PART == 2) ? 5 : 4;
    constexpr int m = i / PW, nn = i % PW;
    constexpr int u = 2 * m + ((PART == 2 || PART == 3) ? 1 : 0), v = 2 * nn + ((PART == 1 || PART == 3) ? 1 : 0);
    const unsigned so = (unsigned)u * row_b + (unsigned)v * pix_b + (unsigned)ss * p.x_cb;
    pv[i] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsA, gbase[u == 8][v == 8], so, 0));
  };
  auto p_line = [&](auto part_c, auto lc) __attribute__((always_inline)) {
    constexpr int PART = decltype(part_c)::value, l = decltype(lc)::value;
    constexpr int PW = (PART == 0 || PART == 2) ? 5 : 4, PH = PART_NPX[PART] / PW;
    if constexpr (l < PH) {
      constexpr int m = l;
      if constexpr (PW == 5) bt5(pv[m * 5], pv[m * 5 + 1], pv[m * 5 + 2], pv[m * 5 + 3], pv[m * 5 + 4]);
      else ct4(pv[m * 4], pv[m * 4 + 1], pv[m * 4 + 2], pv[m * 4 + 3]);
    } else {
      constexpr int nn = l - PH;
      if constexpr (PH == 5) bt5(pv[nn], pv[PW + nn], pv[2 * PW + nn], pv[3 * PW + nn], pv[4 * PW + nn]);
      else ct4(pv[nn], pv[PW + nn], pv[2 * PW + nn], pv[3 * PW + nn]);
    }
  };
  auto p_store = [&](auto part_c, auto ic, float* stD) __attribute__((always_inline)) {
    constexpr int PART = decltype(part_c)::value, i = decltype(ic)::value;
    constexpr int idx = SLOT_IDX[PART_SLOT0[PART] + i];
    if constexpr (idx < NS) *reinterpret_cast<f32x2*>(smf + idx * SLOT_F + v_st) = pv[i];
    else *reinterpret_cast<f32x2*>(stD + (idx - NS) * SLOT_F) = pv[i];
  };
  auto v_read = [&](auto pc, const float* rdD) __attribute__((always_inline)) {
    constexpr int pp = decltype(pc)::value;
    constexpr int g = group_of_pos(pp), k = pp - G_POS0[g];
    if constexpr (pp < NS) vq[g & 1][k] = *reinterpret_cast<const f32x4*>(smf + pp * SLOT_F + v_rd);
    else vq[g & 1][k] = *reinterpret_cast<const f32x4*>(rdD + (pp - NS) * SLOT_F);
  };

  // ---- state of the block loop (all uniform but the lane offsets) ----
  Blk cur = blk_of(L), nxt_b = cur;
  bool has_next = L + wpx < xend;
  if (has_next) nxt_b = blk_of(L + wpx);
  int gs = 0;                                      // super-steps done by this workgroup: parity of the double-buffered V region
  set_producer(cur);
  acc_init(cur.tn);

  // ---- prologue of the FIRST block only ----
  auto pro = [&](auto part_c) __attribute__((always_inline)) {
    constexpr int PART = decltype(part_c)::value;
    constexpr int NPX = PART_NPX[PART];
    constexpr int NLINE = NPX / ((PART == 0 || PART == 2) ? 5 : 4) + ((PART == 0 || PART == 2) ? 5 : 4);
    sfor<NPX>([&](auto ic) { p_gather(part_c, ic, 0); });
    sfor<8>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
      constexpr int f0 = r < 5 ? r : r == 5 ? 25 : r == 6 ? 30 : 35;
      b_load(rc, f0, voff_of(cur.tn));
    });
    sfor<NLINE>([&](auto lc) { p_line(part_c, lc); });
    sfor<NPX>([&](auto ic) { p_store(part_c, ic, smf + NS * SLOT_F + v_st); });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    sfor<NPX>([&](auto ic) { p_gather(part_c, ic, 1); });          // (n_super >= 2 in this form)
  };
  // ---- the super-steps of ONE block; the producers' look-ahead runs on into the next block ----
  auto kl_ = [&](auto part_c) __attribute__((always_inline)) {
    constexpr int PART = decltype(part_c)::value;
    constexpr int NPX = PART_NPX[PART];
    constexpr int NLINE = NPX / ((PART == 0 || PART == 2) ? 5 : 4) + ((PART == 0 || PART == 2) ? 5 : 4);
    constexpr int XSLOT = PART >= 2 ? KFN_S2C_XSLOT_B : 112, XSTEP = 4, SSLOT = JMID + 4 + (PART >= 2 ? 2 : 0), SSTEP = 4;
    static_assert(XSLOT + XSTEP * 10 <= JMID && SSLOT + 2 + SSTEP * 25 <= 4 * NPOS, "producer timetable");
    for (int ks = 0; ks < n_super; ++ks, ++gs) {
      // look-ahead targets: the weights of the next super-step, the gathers of the one after -- of THIS block, of the NEXT block's
      // first super-steps, or (last block) a harmless re-read of the last super-step
      const bool w_wrap = ks + 1 >= n_super, g_wrap = ks + 2 >= n_super;
      const int ss_w = !w_wrap ? ks + 1 : (has_next ? 0 : s_last);
      const unsigned voff_b = voff_of(cur.tn);
      const unsigned voff_w = (w_wrap && has_next) ? voff_of(nxt_b.tn) : voff_b;
      if (ks == n_super - 2 && has_next) set_producer(nxt_b);      // every gather of this block has been issued
      const int ss_g = !g_wrap ? ks + 2 : (has_next ? ks + 2 - n_super : s_last);
      const float* const rdD = smf + (NS + (gs & 1) * ND) * SLOT_F + v_rd;
      float* const stD = smf + (NS + ((gs + 1) & 1) * ND) * SLOT_F + v_st;
      sfor<5>([&](auto kc) { v_read(kc, rdD); });
      sfor<4 * NPOS>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int g = group_of_pos(j / 4);
        constexpr int np = G_NPOS[g], j0 = 4 * G_POS0[g];
        constexpr int kst = (j - j0) / np, k = (j - j0) % np;
        constexpr int pp = G_POS0[g] + k;
        if constexpr (j == JMID) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
        }
        acc[POS_ACC[pp]] = __builtin_amdgcn_mfma_f32_16x16x4f32(vq[g & 1][k][kst], bq[POS_BREG[pp]][kst], acc[POS_ACC[pp]], 0, 0, 0);
        if constexpr (kst == 1 && g + 1 < NGROUP) {
          constexpr int np1 = G_NPOS[g + 1];
          if constexpr (k < np1) v_read(std::integral_constant<int, G_POS0[g + 1] + k>{}, rdD);
          if constexpr (k == np - 1 && np1 > np) v_read(std::integral_constant<int, G_POS0[g + 1] + np>{}, rdD);
        }
        if constexpr (kst == 3 && frag_last_user(pp)) {
          constexpr int nf = next_frag(pp);
          if constexpr (nf >= 64) b_load(std::integral_constant<int, POS_BREG[pp]>{}, ss_w * NFRAG + (nf & 63), voff_w);
          else b_load(std::integral_constant<int, POS_BREG[pp]>{}, ks * NFRAG + nf, voff_b);
        }
        if constexpr (j >= SSLOT + 2 && j < SSLOT + 2 + NPX * SSTEP && (j - SSLOT - 2) % SSTEP == 0)
          p_gather(part_c, std::integral_constant<int, (j - SSLOT - 2) / SSTEP>{}, ss_g);
        if constexpr (j >= XSLOT && j < XSLOT + NLINE * XSTEP && (j - XSLOT) % XSTEP == 0) p_line(part_c, std::integral_constant<int, (j - XSLOT) / XSTEP>{});
        if constexpr (j >= SSLOT && j < SSLOT + NPX * SSTEP && (j - SSLOT) % SSTEP == 0) p_store(part_c, std::integral_constant<int, (j - SSLOT) / SSTEP>{}, stD);
        __builtin_amdgcn_sched_barrier(0);
      });
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
  };

  const bool relu = p.relu != 0;
  float* const stg = smf + LDS_V / 4 + wave * PSTG;
  const unsigned pix_bytes = p.y_pix;
  auto at4 = [](float m0, float m1, float m2, float m3, float m4, float& y0, float& y1, float& y2, float& y3) __attribute__((always_inline)) {
    const float s = m1 + m2, d = m1 - m2;
    y0 = (m0 + s) + m3;
    y1 = d + 2.0f * m3;
    y2 = s + 4.0f * m3;
    y3 = (d + 8.0f * m3) + m4;
  };
  // The whole walk -- prologue, block loop, epilogues -- sits INSIDE each phase's branch: with the block loop outside the four
  // branches every patch / weight register live across a block boundary had to agree at the branch merges, and hipcc spilled 139
  // VGPRs around them (scratch traffic inside the super-steps).  The epilogue's code is therefore instantiated four times (cold).
  auto walk = [&](auto part_c) __attribute__((always_inline)) {
  pro(part_c);
  for (;;) {
    kl_(part_c);

    // ---- epilogue of block `cur`: Y = A^T M A per tile; lane (channel n0 + rl, k = kl), element e = tile (row kl, column e).
    // One pass per tile COLUMN e through the wave's own staging area [4 tile rows][16 px (i, j)][16 ch]; store lane = (pixel, channel
    // quad): one instruction writes 4 rows x 4 px x 64 bytes of one tile row. ----
    {
      const int lv = lane_now();
      const int rl = lv & 15, kl = lv >> 4;
      const int n0 = cur.tn * NT + wave * 16;
      const unsigned long long y_base = (unsigned long long)cur.img0 * p.y_img;
      const unsigned long long y_rest = p.y_bytes - y_base;
      const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(
          reinterpret_cast<char*>(p.y) + y_base, 0, (int)(y_rest < 0x7fffffffull ? y_rest : 0x7fffffffull), 0x00020000);
      const int spx = lv >> 2, nq = lv & 3;                  // store lane: pixel (i, j) = (spx >> 2, spx & 3) of a tile, channel quad
      const bool q_ok = n0 + nq * 4 < p.Cout;
      const unsigned voff_q = q_ok ? (unsigned)((spx >> 2) * p.Wo + (spx & 3)) * p.y_pix + (unsigned)(n0 >> 4) * p.y_cb + (unsigned)(nq * 16) : OOBV;
#pragma unroll
      for (int e = 0; e < ((KFN_S2C_EXP & 128) ? 0 : 4); ++e) {
        float t[4][5];
#pragma unroll
        for (int nu = 0; nu < 5; ++nu)
          at4(acc[nu][e], acc[5 + nu][e], acc[10 + nu][e], acc[15 + nu][e], acc[20 + nu][e], t[0][nu], t[1][nu], t[2][nu], t[3][nu]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float y0, y1, y2, y3;
          at4(t[i][0], t[i][1], t[i][2], t[i][3], t[i][4], y0, y1, y2, y3);
          float* const row = stg + kl * (16 * 16 + 16) + (i * 4) * 16 + rl;
          row[0] = y0; row[16] = y1; row[32] = y2; row[48] = y3;
        }
        __builtin_amdgcn_wave_barrier();
        const int tx = cur.cb * 4 + e;
        const bool col_ok = tx < p.Tw;                           // (uniform)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          f32x4 v = *reinterpret_cast<const f32x4*>(stg + kk * (16 * 16 + 16) + spx * 16 + nq * 4);
          if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
          const int img_rel = kk < cur.brk ? 0 : 1;
          const int ty = kk < cur.brk ? cur.ty0 + kk : kk - cur.brk;
          const bool row_ok = col_ok && cur.vr0 + kk < p.vrows;  // (uniform; Ho is a multiple of 4: whole tiles)
          const unsigned soff = (unsigned)img_rel * p.y_img + (unsigned)(4 * ty * p.Wo + 4 * tx) * pix_bytes;
          if (!(KFN_S2C_EXP & 64)) kfn::buffer_store_b128<KFN_NT_STORE_AUX>(v, rsY, row_ok ? voff_q : OOBV, soff);
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    if (!has_next) break;
    // ---- next block: the consumers' side follows the producers, who are already there ----
    L += wpx;
    cur = nxt_b;
    has_next = L + wpx < xend;
    if (has_next) nxt_b = blk_of(L + wpx);
    acc_init(cur.tn);
  }
  };
  if (part == 0) walk(std::integral_constant<int, 0>{});
  else if (part == 1) walk(std::integral_constant<int, 1>{});
  else if (part == 2) walk(std::integral_constant<int, 2>{});
  else walk(std::integral_constant<int, 3>{});
}

}  // namespace

int kfn::wino_s2c_lds_bytes() { return KFN_S2C_PERSIST ? (LDS_P > LDS_BYTES ? LDS_P : LDS_BYTES) : LDS_BYTES; }

// pointer-free routing check of the F(4,2) form (kfn_winograd_s2_supported answers it for wino_form = KFN_WINO_FORM_S2_F42)
int kfn::wino_s2c_supported(const kfn_conv_desc* d) {
  if (d->x_dtype != KFN_ACT_F32 || d->y_dtype != KFN_ACT_F32 || d->operand_dtype != KFN_OPERAND_F32) return 0;
  if (d->kh != 3 || d->kw != 3 || d->stride != 2 || d->transposed || d->epilogue != KFN_EPI_NONE) return 0;
  if (d->H <= 0 || d->W <= 0 || (d->H & 7) || (d->W & 7)) return 0;          // whole 4x4 output tiles; pad after the image only
  if (d->H / 8 < 4) return 0;                                                 // a 4-row tile block straddles at most two images
  if (d->Cin <= 0 || d->Cin % SS_CH != 0 || d->cout_pad % 32 != 0) return 0;
  if (d->Cout % 4 != 0 || d->ldy % 4 != 0 || d->ldx % 2 != 0) return 0;      // 16-byte stores, 8-byte gathers
  if (d->x_layout == KFN_LAYOUT_C16 && d->ldx != d->Cin) return 0;            // channel-blocked tensors are dense
  if (d->y_layout == KFN_LAYOUT_C16 && (d->Cout % 16 != 0 || d->ldy != d->Cout)) return 0;
  return 1;
}

int kfn::launch_wino_s2c(const kfn_conv_desc* d, const float* x, const void* u_packed, const float* bias, float* y, void* stream) {
  if (!wino_s2c_supported(d))
    return kfn::fail(KFN_ERR_UNSUPPORTED, "kfn_conv2d_winograd_s2 (F(4,2) form): needs fp32, 3x3 stride 2, H and W multiples of 8 with H >= 32, "
                     "Cin %% 16 == 0, Cout %% 4 == 0, ldy %% 4 == 0, KFN_LAYOUT_C16 tensors dense with C %% 16 == 0 (got H=%d W=%d Cin=%d ldx=%d Cout=%d ldy=%d, "
                     "x_layout %d, y_layout %d)", d->H, d->W, d->Cin, d->ldx, d->Cout, d->ldy, d->x_layout, d->y_layout);
  KFN_REQUIRE(d->N > 0 && d->ldx >= d->Cin && d->ldy >= d->Cout && d->cout_pad >= d->Cout, "kfn_conv2d_winograd_s2 (F(4,2) form): bad strides / channel counts");
  KFN_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(u_packed) | reinterpret_cast<uintptr_t>(y)) & 15) == 0,
              "kfn_conv2d_winograd_s2 (F(4,2) form): buffers must be 16-byte aligned");
  const long img_b = (long)d->H * d->W * d->ldx * 4L;
  const long out_b = (long)(d->H / 2) * (d->W / 2) * d->ldy * 4L;
  KFN_REQUIRE(2 * img_b < (1L << 31) && 2 * out_b < (1L << 31) && (long)NFRAG * d->cout_pad * d->Cin * 4L < (1L << 31),
              "kfn_conv2d_winograd_s2 (F(4,2) form): image or kernel beyond 2 GiB of 32-bit offsets");
  S2cArgs a;
  a.x = x; a.u = static_cast<const float*>(u_packed); a.bias = bias; a.y = y;
  a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.ldx = d->ldx;
  a.Cout = d->Cout; a.cout_pad = d->cout_pad; a.ldy = d->ldy;
  a.Ho = d->H / 2; a.Wo = d->W / 2;
  a.Th = a.Ho / 4; a.Tw = a.Wo / 4;
  a.vrows = d->N * a.Th;
  a.bw = kfn::ceil_div(a.Tw, 4);
  const long tiles_m = (long)a.bw * kfn::ceil_div(a.vrows, 4);
  a.tiles_n = kfn::ceil_div(d->cout_pad, NT);
  KFN_REQUIRE(tiles_m * a.tiles_n < (1L << 31), "kfn_conv2d_winograd_s2 (F(4,2) form): grid too large");
  a.tiles_m = (int)tiles_m;
  a.relu = d->relu;
  {
    int ng = a.tiles_n % 2 == 0 ? 2 : a.tiles_n;
    if (d->wino_order == KFN_WINO_ORDER_N_FAST) ng = a.tiles_n;
    else if (d->wino_order == KFN_WINO_ORDER_M_FAST) ng = 1;
    else if (d->wino_order >= KFN_WINO_ORDER_GROUPS(1)) ng = d->wino_order - KFN_WINO_ORDER_GROUPS(0);
    if (ng < 1 || ng > a.tiles_n || a.tiles_n % ng != 0) ng = a.tiles_n;
    a.n_group = ng;
  }
  const long in_pix = (long)d->N * d->H * d->W, out_pix = (long)d->N * a.Ho * a.Wo;
  a.x_bytes = (unsigned long long)(((in_pix - 1) * d->ldx + d->Cin) * 4L);
  a.y_bytes = (unsigned long long)(((out_pix - 1) * d->ldy + d->Cout) * 4L);
  a.u_bytes = (unsigned)((long)NFRAG * d->cout_pad * d->Cin * 4L);
  const bool xb = d->x_layout == KFN_LAYOUT_C16, yb = d->y_layout == KFN_LAYOUT_C16;
  a.x_img = (unsigned)img_b; a.x_pix = xb ? 64u : (unsigned)d->ldx * 4u; a.x_cb = xb ? (unsigned)(d->H * d->W) * 64u : 64u;
  a.y_img = (unsigned)out_b; a.y_pix = yb ? 64u : (unsigned)d->ldy * 4u; a.y_cb = yb ? (unsigned)(a.Ho * a.Wo) * 64u : 64u;
  const long nwg = (long)a.tiles_m * a.tiles_n;
#if KFN_S2C_PERSIST
  // the persistent form: one workgroup per CU walking its share of the tile blocks (needs two super-steps of look-ahead inside a
  // block; launches that do not even fill the chip once keep one workgroup per block)
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  // MEASURED (batch 20, same box, two runs each): conv2a (4 super-steps) 2.66 -> 2.53 ms, conv3a (16) 4.01 -> 3.93, conv4a (32) 3.81 ->
  // 3.81.  What a block still costs beyond its super-steps (15 of conv2a's 54 us) is the issue of its 131 KB of output stores (a
  // timing build without them: -8 % on conv2a in either form) and the consumers' epilogue, which persistence cannot hide.  (The
  // first build spilled 31 block-level VGPRs whose reloads queued behind the prefetches in flight: lane_now().)
  if (d->Cin / SS_CH >= 2 && d->Cin / SS_CH <= KFN_S2C_PERSIST_MAX_SS && nwg >= 2L * n_cu) {
    static std::atomic<uint64_t> attr_done_p{0};
    int rcp = kfn::set_max_dynamic_lds(reinterpret_cast<const void*>(wino_s2c_pkernel), LDS_P, attr_done_p);
    if (rcp != KFN_OK) return rcp;
    hipLaunchKernelGGL(wino_s2c_pkernel, dim3((unsigned)n_cu), dim3(512), LDS_P, (hipStream_t)stream, a);
    KFN_LAUNCH_CHECK("wino_s2c_pkernel");
    return KFN_OK;
  }
#endif
  static std::atomic<uint64_t> attr_done{0};
  int rc = kfn::set_max_dynamic_lds(reinterpret_cast<const void*>(wino_s2c_kernel), LDS_BYTES, attr_done);
  if (rc != KFN_OK) return rc;
  hipLaunchKernelGGL(wino_s2c_kernel, dim3((unsigned)nwg), dim3(512), LDS_BYTES, (hipStream_t)stream, a);
  KFN_LAUNCH_CHECK("wino_s2c_kernel");
  return KFN_OK;
}
