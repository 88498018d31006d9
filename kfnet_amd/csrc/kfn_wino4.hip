// kfn_wino4.hip -- single-kernel Winograd F(4x4,3x3) for the 3x3 stride-1 SAME layers with Cin >= 512
// (tf.layers.conv2d behind Network.conv, cnn_wrapper/network.py:116-135; SCoordNet conv3b / conv4b / conv5,
// cnn_wrapper/SCoordNet.py:26-30): 36 products per 4x4 outputs instead of 16 per 2x2 -- 2.25 multiplies per output
// instead of 4 (F(2x2,3x3), kfn_wino3.hip) or 9 (direct).  fp32 throughout; interpolation points {0, +-1, +-2}:
//
//   B^T = [ 4  0 -5  0  1  0 ]   G = [ 1/4    0     0  ]   A^T = [ 1  1  1  1  1  0 ]
//         [ 0 -4 -4  1  1  0 ]       [-1/6  -1/6  -1/6 ]         [ 0  1 -1  2 -2  0 ]
//         [ 0  4 -4 -1  1  0 ]       [-1/6   1/6  -1/6 ]         [ 0  1  1  4  4  0 ]
//         [ 0 -2 -1  2  1  0 ]       [ 1/24  1/12  1/6 ]         [ 0  1 -1  8 -8  1 ]
//         [ 0  2 -1 -2  1  0 ]       [ 1/24 -1/12  1/6 ]
//         [ 0  4  0 -5  0  1 ]       [ 0     0     1   ]
//
//   Y = A^T [ (G g G^T) o (B^T d B) ] A.  U = G g G^T is prepared on the host in fp64 (graph.pack_winograd_f43_kernel).
//   Error budget on the full 12-layer SCoordNet (tools/experiments/f43_error_budget.py): coordinates 1.4e-6 max-abs
//   against the fp64 convolution (F(2x2,3x3): 5e-7; the parity target is 1e-4).
//
// Structure = kfn_wino3.hip's, re-cut for 36 positions:
//   * a workgroup of FOUR waves (one per SIMD) owns a block of 4 x 8 tiles (16 x 32 output pixels = the 32 rows of a
//     32x32 MFMA) x 64 output channels.  Wave (wx, wc) accumulates the 18 positions (xi, nu), xi in {3 wx .. 3 wx + 2},
//     for the 32 channels of column block wc: 18 accumulators of 32x32 = 288 registers.
//   * V = B^T d B of a SUPER-STEP (16 input channels = 2 chunks of 8) is produced once per workgroup and shared through
//     LDS: wave w owns tile COLUMN w of the block, lane = (tile row, channel pair) gathers its tile's 6x6 patch for 2
//     channels (36 loads of 8 bytes; 8 neighbouring lanes read 64 contiguous bytes of a pixel; patch columns are
//     wave-uniform scalar offsets, columns outside the image read through a zero-length descriptor, rows outside carry
//     an offset that fails the range check: zero padding costs no instruction), transforms it in registers (144 packed
//     FMAs as one burst) and stores 36 x 8 bytes.  4 chunk buffers of [36 positions][2 k-halves][32 tiles][4 floats]
//     (padded: the producer's stores and the consumer's fragment reads are bank-conflict free) = 153 KiB; the buffers
//     written during super-step k are read during k+1: one barrier per super-step (144 MFMAs = 9.2 K cycles).
//   * B fragments as in kfn_wino3.hip: U re-packed [Cin/8][36][cout_pad][8] (one 1 KiB run per (chunk, position, 32
//     channels)), L2 -> registers through a ring one chunk deep.
//   * Output transform: the nu pass is local to a wave (all six nu of its three xi); the xi pass splits into the wave's
//     half and its partner's (same column block, other xi half).  Through an LDS image [32 tiles][16 pixels][64 channels]
//     the partners exchange: each writes the two output rows the OTHER finishes, barrier, each adds what it received to
//     its own partial for the two rows it finishes (a + b in one fixed order per element: deterministic); the image then
//     leaves as 16-byte stores, 16 lanes per 256 contiguous bytes of a pixel.
//   * 18 accumulators = 288 registers exceed the 256 of the accumulation file: 16 live there (builtin MFMAs), two in
//     VGPRs (MFMAs written by hand with VGPR destinations -- hipcc gives all MFMAs of a function one register class).
//
// Measured (profiles/r04_wino4_microbench.log, batch 32): 1.43-1.48x wino3_kernel on conv2b / 3b / 4b / 5 / 6, 1.3x on the
// 64-channel layers; 113-116 executed TFLOP/s = 0.72-0.74 of the fp32 MFMA peak (wino3: 0.88).  Where the rest goes
// (timing builds, conv4b): patch loads 12 % (7 % issue even when they hit L1 -- 36 loads per 144 MFMAs and wave --, 5 %
// memory), the transform burst 6 %, weight-fragment latency 2 %, prologue + epilogue ~5 %.
#include "kfn_common.h"
#include <type_traits>
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr unsigned ROW_POISON = 0x80000000u;   // a row offset that fails the range check (two images stay below 2 GiB)
constexpr int BW4 = 4, BH4 = 8;            // tile block: 4 (x) by 8 (y) tiles of 4x4 output pixels
constexpr int NPOS = 36, WPOS = 18;        // positions in all / per wave
constexpr int CPS = 2;                     // chunks of 8 input channels per super-step
constexpr int VHALF = 32 * 16 + 32;        // one k-half of a position: [32 tiles][4 floats] + pad (8 dwords: see below)
constexpr int VPOS = 2 * VHALF;
constexpr int VBUF = NPOS * VPOS + 16;     // one chunk; the pad puts the two chunks of a super-step 4 dwords apart mod 32
constexpr int LDS_V = 2 * CPS * VBUF;      // 156 736 B
constexpr int LDS_OUT = 32 * 16 * 64 * 4;  // 131 072 B: the output image of the epilogue (overlays the V buffers)
#ifndef KFN_W4_NB
#define KFN_W4_NB 9
#endif
constexpr int NB = KFN_W4_NB;              // B ring: positions ahead (must divide 36; 9 = half a chunk = 2.3 K cycles)
constexpr int NVR = 6;                     // V fragment ring (must divide 36)
constexpr int SPC = 72;                    // MFMA slots per chunk: 18 positions x 4 k-steps
// producer schedule inside the 144 slots of a super-step
#ifndef KFN_W4_GSTEP
#define KFN_W4_GSTEP 2      // one patch load every GSTEP slots, from slot 0
#define KFN_W4_XSLOT 100    // the transform burst
#define KFN_W4_SSLOT 106    // first V store, then one per slot
#endif
// Line touches (see `touch` below): slot of the first of the three touch loads of a super-step; < 0 = none.  OFF: they move
// the stall, they do not remove it (measured: conv4b 6.78 ms with, 6.38 without -- profiles/r04_wino4_microbench.log).
#ifndef KFN_W4_TSLOT
#define KFN_W4_TSLOT (-1)
#endif
// The transform as 12 one-dimensional passes of 2 x 6 packed instructions, each half in its own MFMA gap (1), instead of one
// burst of 144 at XSLOT (0): row pass r (patch row r: its six loads went out at slots 12 r .. 12 r + 10) in slots RS0 + 8 r
// and + 1, column pass c in slots CS0 + 6 c and + 1, its six V stores (positions 6 xi + c) in the six slots after it.
// MEASURED (profiles/r04_wino4_microbench.log, r4t): no gain -- conv2b 6.99 -> 7.11 ms, conv3b 6.41 -> 6.48, conv4b 6.26 ->
// 6.16, conv5 / conv1b equal: a packed fp32 instruction costs the MFMA stream the same wherever it stands.  XDIST = 2 (the
// same in 48 quarters of six PLAIN v_fma / v_add / v_sub, -DKFN_W4_RS0=62 -DKFN_W4_CS0=106): 7.57 / 7.02 / 6.72 ms, worse.  OFF.
#ifndef KFN_W4_XDIST
#define KFN_W4_XDIST 0
#define KFN_W4_RS0 60
#define KFN_W4_CS0 102
#endif
static_assert(KFN_W4_XDIST != 1 || (KFN_W4_RS0 + 8 * 5 + 1 < KFN_W4_CS0 && KFN_W4_CS0 + 6 * 5 + 2 + 6 <= 144 && KFN_W4_RS0 >= 12 * 0 + 10),
              "distributed transform schedule");
static_assert(KFN_W4_XDIST != 2 || (KFN_W4_RS0 + 8 * 5 + 3 < KFN_W4_CS0 && KFN_W4_CS0 + 4 * 5 + 4 + 6 <= 144 && KFN_W4_RS0 >= 10),
              "distributed plain-instruction transform schedule");
// ... and in how many pieces each of the three touch loads is issued (1, 2 or 4: 64 / 32 / 16 live lanes per piece), one
// piece every KFN_W4_TSTEP slots: spreads the misses over the super-step
#ifndef KFN_W4_TPIECES
#define KFN_W4_TPIECES 1
#define KFN_W4_TSTEP 1
#endif
// timing experiments only (wrong results on purpose; tools/mb/build_w4.sh): bit 0 no transform, 1 no patch loads,
// 2 no V stores, 3 no B loads in the main loop, 4 every patch load of the main loop re-reads super-step 0 (L1/L2-hot
// activations), 5 every B load re-reads chunk 0 (L2-hot weights)
#ifndef KFN_W4_DBG
#define KFN_W4_DBG 0
#endif
// which form kfn_conv2d_winograd_f43 launches when kfn_conv_desc.wino_form is AUTO: 0 = four waves (wino4_kernel), 1 = eight
#ifndef KFN_W4_DEFAULT_EIGHT_WAVE
#define KFN_W4_DEFAULT_EIGHT_WAVE 0
#endif
static_assert(36 * KFN_W4_GSTEP <= KFN_W4_XSLOT && KFN_W4_XSLOT < KFN_W4_SSLOT && KFN_W4_SSLOT + 36 <= CPS * SPC, "producer schedule");
static_assert(36 % NB == 0 && 36 % NVR == 0 && LDS_OUT <= LDS_V, "ring slots are compile-time constants per super-step");

struct Wino4Args {
  const float* x;
  const float* u4;    // [Cin/8][36][cout_pad][8]
  const float* bias;
  float* y;
  int N, H, W, Cin, ldx;
  int Cout, cout_pad, ldy;
  int Th, Tw;         // tiles per image (4x4 output pixels each)
  int vrows;          // N * Th: tile rows of the whole batch, image after image
  int bw;             // ceil(Tw / 4) column blocks
  int tiles_m, tiles_n;
  int relu;
  int n_group;        // workgroup order: n_group channel groups (of 64) of one tile block are neighbours, blocks next, group sets slowest
  unsigned long long x_bytes;
  unsigned long long y_bytes;
  unsigned u_bytes;
  // wino4b_kernel only -- byte strides of the activation layouts (kfn_conv_desc.x_layout / y_layout): element (n, h, w, c) lies at
  // n * img + (h * W + w) * pix + (c >> 4) * cb + (c & 15) * 4.  NHWC: pix = ld * 4, cb = 64; KFN_LAYOUT_C16 (per image
  // [C/16][H][W][16]): pix = 64, cb = H * W * 64.  The image stride is the same in both.
  unsigned x_pix, x_cb, x_img;
  unsigned y_pix, y_cb, y_img;
  // split-K (wino4b_kernel only; kfn_conv2d_winograd_f43_splitk): the grid is k_split copies of the tile grid, copy s
  // accumulates super-steps [s * ss_per_split, (s + 1) * ss_per_split) and writes its RAW partial sums (no bias, no ReLU)
  // into plane s of the workspace (y / ldy / y_bytes describe ONE plane, y_split_bytes the distance between planes)
  int k_split, ss_per_split;
  unsigned long long y_split_bytes;
#ifdef KFN_WINO4_PROF
  unsigned long long* prof;   // tools/mb/wino4_prof.hip: [block][wave][8] phase stamps, then [block][wave][11] timeline of one super-step
#endif
};

#ifdef KFN_WINO4_PROF
// (every lane stores the same value to the same address: a lane-0 branch would be divergent control flow, after which
//  hipcc wraps the gathers' uniform descriptors in waterfall loops -- the measurement trap of CHANGELOG round 2)
#define KFN_STAMP4(i) (p.prof[((size_t)blockIdx.x * 4 + wave) * 8 + (i)] = __builtin_readcyclecounter())
#ifndef KFN_W4_TL_KS
#define KFN_W4_TL_KS 1      // the super-step whose 16-slot timeline is kept
#endif
#else
#define KFN_STAMP4(i) do { } while (0)
#endif

template <int I, int N, class F>
__device__ __forceinline__ void sfor4_impl(F& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    sfor4_impl<I + 1, N>(f);
  }
}
template <int N, class F>
__device__ __forceinline__ void sfor4(F&& f) {
  sfor4_impl<0, N>(f);
}

__device__ __forceinline__ int xcd_remap4(int b, int nwg) {
  int xcd = b & 7;
  int q = nwg >> 3, r = nwg & 7;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (b >> 3);
}

// packed fp32 helpers (two channels per register pair).  Constants ride in SGPR pairs.
__device__ __forceinline__ f32x2 pk_add4(const f32x2& a, const f32x2& b) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ f32x2 pk_sub4(const f32x2& a, const f32x2& b) {   // a - b
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ f32x2 pk_fma4(const f32x2& a, const f32x2& k, const f32x2& c) {   // a * k + c, k wave-uniform
  f32x2 r;
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(k), "v"(c));
  return r;
}
__device__ __forceinline__ f32x2 pk_mul4(const f32x2& a, const f32x2& k) {
  f32x2 r;
  asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "s"(k));
  return r;
}

struct BtConst {
  f32x2 p4, m4, m5, p2, m2;
};
// one 1-D pass of B^T on six register pairs, in place:
//   t0 = 4 d0 - 5 d2 + d4          t1 = (d4 - 4 d2) + (d3 - 4 d1)     t2 = (d4 - 4 d2) - (d3 - 4 d1)
//   t3 = (d4 - d2) + 2 (d3 - d1)   t4 = (d4 - d2) - 2 (d3 - d1)       t5 = 4 d1 - 5 d3 + d5
__device__ __forceinline__ void bt6(f32x2& d0, f32x2& d1, f32x2& d2, f32x2& d3, f32x2& d4, f32x2& d5, const BtConst& k) {
  const f32x2 a = pk_fma4(d2, k.m4, d4);
  const f32x2 b = pk_fma4(d1, k.m4, d3);
  const f32x2 c = pk_sub4(d4, d2);
  const f32x2 e = pk_sub4(d3, d1);
  const f32x2 u = pk_fma4(d2, k.m5, d4);
  const f32x2 v = pk_fma4(d3, k.m5, d5);
  d0 = pk_fma4(d0, k.p4, u);
  d5 = pk_fma4(d1, k.p4, v);
  d1 = pk_add4(a, b);
  d2 = pk_sub4(a, b);
  d3 = pk_fma4(e, k.p2, c);
  d4 = pk_fma4(e, k.m2, c);
}
// the same pass in two halves of six instructions (the temporaries live across one MFMA slot)
struct BtTmp {
  f32x2 a, b, c, e, u, v;
};
__device__ __forceinline__ void bt6_a(const f32x2& d1, const f32x2& d2, const f32x2& d3, const f32x2& d4, const f32x2& d5,
                                      const BtConst& k, BtTmp& t) {
  t.a = pk_fma4(d2, k.m4, d4);
  t.b = pk_fma4(d1, k.m4, d3);
  t.c = pk_sub4(d4, d2);
  t.e = pk_sub4(d3, d1);
  t.u = pk_fma4(d2, k.m5, d4);
  t.v = pk_fma4(d3, k.m5, d5);
}
__device__ __forceinline__ void bt6_b(f32x2& d0, f32x2& d1, f32x2& d2, f32x2& d3, f32x2& d4, f32x2& d5, const BtConst& k,
                                      const BtTmp& t) {
  d0 = pk_fma4(d0, k.p4, t.u);
  d5 = pk_fma4(d1, k.p4, t.v);
  d1 = pk_add4(t.a, t.b);
  d2 = pk_sub4(t.a, t.b);
  d3 = pk_fma4(t.e, k.p2, t.c);
  d4 = pk_fma4(t.e, k.m2, t.c);
}
// ... and in four quarters of six PLAIN fp32 instructions (component Z = 0 | 1 of every pair; inline asm so that the
// compiler does not re-pack them): KFN_W4_XDIST == 2
__device__ __forceinline__ float pl_fma(float a, float k, float c) {
  float r;
  asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(k), "v"(c));
  return r;
}
__device__ __forceinline__ float pl_add(float a, float b) {
  float r;
  asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float pl_sub(float a, float b) {
  float r;
  asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
template <int Z>
__device__ __forceinline__ void bt6_qa(const f32x2& d1, const f32x2& d2, const f32x2& d3, const f32x2& d4, const f32x2& d5,
                                       const BtConst& k, BtTmp& t) {
  t.a[Z] = pl_fma(d2[Z], k.m4.x, d4[Z]);
  t.b[Z] = pl_fma(d1[Z], k.m4.x, d3[Z]);
  t.c[Z] = pl_sub(d4[Z], d2[Z]);
  t.e[Z] = pl_sub(d3[Z], d1[Z]);
  t.u[Z] = pl_fma(d2[Z], k.m5.x, d4[Z]);
  t.v[Z] = pl_fma(d3[Z], k.m5.x, d5[Z]);
}
template <int Z>
__device__ __forceinline__ void bt6_qb(f32x2& d0, f32x2& d1, f32x2& d2, f32x2& d3, f32x2& d4, f32x2& d5, const BtConst& k,
                                       const BtTmp& t) {
  d0[Z] = pl_fma(d0[Z], k.p4.x, t.u[Z]);
  d5[Z] = pl_fma(d1[Z], k.p4.x, t.v[Z]);
  d1[Z] = pl_add(t.a[Z], t.b[Z]);
  d2[Z] = pl_sub(t.a[Z], t.b[Z]);
  d3[Z] = pl_fma(t.e[Z], k.p2.x, t.c[Z]);
  d4[Z] = pl_fma(t.e[Z], k.m2.x, t.c[Z]);
}
__device__ __forceinline__ void bt_d_b6(f32x2 (&v)[36], const BtConst& k) {   // v[6 r + c] -> v[6 xi + nu]
#pragma unroll
  for (int r = 0; r < 6; ++r) bt6(v[6 * r + 0], v[6 * r + 1], v[6 * r + 2], v[6 * r + 3], v[6 * r + 4], v[6 * r + 5], k);
#pragma unroll
  for (int c = 0; c < 6; ++c) bt6(v[c], v[6 + c], v[12 + c], v[18 + c], v[24 + c], v[30 + c], k);
}

__global__ __launch_bounds__(256, 1) void wino4_kernel(Wino4Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem4[];   // [4][VBUF], later the output image

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wx = wave & 1, wc = wave >> 1;      // xi half, 32-channel column block
  KFN_STAMP4(0);
  const int nwg = p.tiles_m * p.tiles_n;
  const int tile = xcd_remap4(blockIdx.x, nwg);
  // Workgroups that run side by side on an XCD share its L2: with n_group channel groups of a tile block adjacent the
  // input block crosses the fabric tiles_n / n_group times (instead of tiles_n) and a weight slice is shared by
  // 32 / n_group CUs (instead of 32).  n_group divides tiles_n (the launcher sees to it).
  const int per = p.tiles_m * p.n_group;
  const int gset = tile / per, rem = tile - gset * per;
  const int tm = rem / p.n_group;
  const int tn = gset * p.n_group + (rem - tm * p.n_group);
  const int cb = tm % p.bw, rb = tm / p.bw;
  const int nbase = tn * 64;                    // the workgroup's 64 output channels
  const int n0 = nbase + wc * 32;               // this wave's 32

  // ---- block geometry (uniform) ----------------------------------------------------------
  const int vr0 = rb * BH4;
  const int img0 = vr0 / p.Th;
  const int ty0 = vr0 - img0 * p.Th;
  const int brk = (p.Th - ty0 < BH4) ? (p.Th - ty0) : BH4;   // tile rows >= brk belong to image img0 + 1 (Th >= 8)

  const unsigned long long a_base = (unsigned long long)img0 * p.H * p.W * p.ldx * 4ull;
  const unsigned long long a_rest = p.x_bytes - a_base;
  const unsigned long long two_img = 2ull * p.H * p.W * p.ldx * 4ull;
  const __amdgpu_buffer_rsrc_t rsU =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.u4), 0, p.u_bytes, 0x00020000);

  // ---- this lane as a PRODUCER: tile (tr = lane >> 3, tc = wave), channels 2 cp, 2 cp + 1 of the super-step's 16 ------
  // The wave owns tile COLUMN `wave` of the block: the six patch columns are wave-uniform (scalar offsets; a column left or
  // right of the image reads through a descriptor with num_records = 0), the rows are per lane (offset, or a mark that
  // fails the range check).  No vector ALU work per load: every VALU instruction is paid in MFMA time here.
  const int cp = lane & 7, ptr = lane >> 3;
  unsigned roff[6];                     // per lane: byte offset of patch row r at this lane's channel pair, or ROW_POISON
  unsigned coff[6];                     // uniform: byte offset of patch column c
  bool cok[6];                          // uniform: column inside the image (and the tile column exists)
  {
    const int img_rel = ptr < brk ? 0 : 1;
    const int ty = ptr < brk ? ty0 + ptr : ptr - brk;
    const int tx = cb * BW4 + wave;
    const bool row_tile_ok = (vr0 + ptr < p.vrows);
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const int yy = 4 * ty - 1 + r;
      roff[r] = (row_tile_ok && (unsigned)yy < (unsigned)p.H)
                    ? (unsigned)((img_rel * p.H + yy) * p.W) * (unsigned)(p.ldx * 4) + (unsigned)(cp * 8) : ROW_POISON;
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const int xx = 4 * tx - 1 + c;
      cok[c] = (tx < p.Tw) && ((unsigned)xx < (unsigned)p.W);
      coff[c] = cok[c] ? (unsigned)(xx * p.ldx * 4) : 0u;
    }
  }
  const int x_records = (int)(a_rest < two_img ? a_rest : two_img);
  // ---- line touches -------------------------------------------------------------------------------------
  // A super-step reads 64 bytes (16 channels) of every patch pixel: every OTHER super-step opens a new 128-byte line of
  // every pixel, and with 288 line misses per wave in flight the patch loads back up the vector-memory path -- the odd
  // super-steps ran 13.2 K cycles against 10.3 K for the even ones (tools/mb/wino4_prof.hip).  So the line is opened one
  // super-step EARLY by three loads per wave in which every lane touches a different pixel of the wave's footprint
  // (its tile column: 6 patch columns x the 32 core rows of the block = 192 pixels; the two halo rows are left to the
  // patch loads): 3 instructions instead of 36 raise the same misses, and when the patch loads come they hit the L2.
  // RESULT (cycle stamps, conv4b): the odd super-steps do drop to 10.6 K -- and the even ones, which now carry the 768
  // misses of the four waves in three instructions each, rise to 13.5 K: the stall sits in the CU's miss handling
  // (~140 lines must be in flight per CU at HBM latency to feed this loop), not in who raises the misses.  Kept as an
  // experiment switch (KFN_W4_TSLOT >= 0), off by default.
  unsigned tq[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int q = lane + 64 * k;              // pixel q of the footprint: row q / 6 of the 32 core rows, patch column q % 6
    const int rr = q / 6, fc = q - 6 * rr;
    const int tr = rr >> 2;
    const int img_rel = tr < brk ? 0 : 1;
    const int ty = tr < brk ? ty0 + tr : tr - brk;
    const int yy = 4 * ty + (rr & 3);
    const int xx = 4 * (cb * BW4 + wave) - 1 + fc;
    const bool ok = (vr0 + tr < p.vrows) && (yy < p.H) && ((unsigned)xx < (unsigned)p.W);
    tq[k] = ok ? (unsigned)((img_rel * p.H + yy) * p.W + xx) * (unsigned)(p.ldx * 4) : ROW_POISON;
  }
  unsigned touched = 0, tv[3] = {0u, 0u, 0u};   // tv: the touch loads in flight (consumed one super-step later: no wait)
  // V store address of this lane inside a chunk buffer: chunk cp >> 2, k-half (cp >> 1) & 1, pair cp & 1, tile 4 ptr + wave
  // (a ds_write_b64 group of 16 lanes = 8 channel pairs x 2 tile rows: dwords {0,2} + {0,8} + {0,4} + {0,16}: 32 banks once)
  const int v_st = (cp >> 2) * VBUF + ((cp >> 1) & 1) * VHALF + (ptr * 4 + wave) * 16 + (cp & 1) * 8;
  const int n_chunks = p.Cin / 8;
  const int n_super = n_chunks / CPS;
  const int s_last = n_super - 1;

  // ---- this lane as a CONSUMER ---------------------------------------------------------------
  const int li = lane & 31, lh = lane >> 5;
  const int v_lane = (WPOS * wx) * VPOS + lh * VHALF + li * 16;          // position 18 wx + l: + l * VPOS
  const unsigned voff_b = (unsigned)(((n0 + li) * 8 + lh * 4) * 4);
  const unsigned b_step = (unsigned)p.cout_pad * 32u;                    // bytes between consecutive (chunk, position) fragments
  const int q_last = n_chunks * NPOS - 1;

  f32x16 acc[WPOS];
#pragma unroll
  for (int l = 0; l < WPOS; ++l)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[l][e] = 0.f;

  f32x2 pv[36];        // producer: raw 6x6 patch -> V, two channels
  f32x4 bq[NB];        // B ring
  f32x4 vq[NVR];       // V fragment ring
  BtConst kc;
  BtTmp bt_tmp;        // (distributed transform: the six temporaries of a pass between its two slots)
  kc.p4 = f32x2{4.f, 4.f}; kc.m4 = f32x2{-4.f, -4.f}; kc.m5 = f32x2{-5.f, -5.f}; kc.p2 = f32x2{2.f, 2.f}; kc.m2 = f32x2{-2.f, -2.f};

  auto p_gather = [&](auto ic, int ss) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    constexpr int r = i / 6, c = i % 6;
    const int sc = (KFN_W4_DBG & 16) ? 0 : (ss < s_last ? ss : s_last);     // past the end: re-read the last super-step (nobody consumes it)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.x)) + a_base, 0, cok[c] ? x_records : 0, 0x00020000);
    pv[i] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, roff[r], coff[c] + (unsigned)(sc * 64), 0));
  };
  auto p_store = [&](auto gc, int ss) __attribute__((always_inline)) {
    constexpr int g = decltype(gc)::value;
    *reinterpret_cast<f32x2*>(smem4 + (ss & 1) * (CPS * VBUF) + v_st + g * VPOS) = pv[g];
  };
  // touch the line that super-step `ss` will read (only when it opens a new one: ss even; else a zero-length descriptor
  // makes the load a no-op -- no control flow in the MFMA stream).  The value is kept alive, never used.
  auto touch = [&](auto kc_, int ss) __attribute__((always_inline)) {
    constexpr int piece = decltype(kc_)::value;           // piece = k * TPIECES + part
    constexpr int k = piece / KFN_W4_TPIECES, part = piece % KFN_W4_TPIECES;
    const bool live = ((ss & 1) == 0) && ss <= s_last;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.x)) + a_base, 0, live ? x_records : 0, 0x00020000);
    touched |= tv[k];      // the previous touch through this register: arrived long ago
    const unsigned off = (KFN_W4_TPIECES == 1 || (lane * KFN_W4_TPIECES) / 64 == part) ? tq[k] : ROW_POISON;
    tv[k] = __builtin_amdgcn_raw_buffer_load_b32(rs, off, (unsigned)(ss * 64), 0);
  };
  // this wave's fragment (chunk ch, local position l) = global fragment ch * 36 + 18 wx + l, into ring slot `sl`
  auto b_load = [&](auto sl_, int ch, int l) __attribute__((always_inline)) {
    constexpr int sl = decltype(sl_)::value;
    const int q = ((KFN_W4_DBG & 32) ? 0 : ch) * NPOS + WPOS * wx + l;
    const int qc = q < q_last ? q : q_last;
    bq[sl] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsU, voff_b, (unsigned)qc * b_step, 0));
  };
  auto v_read = [&](auto sl_, int ch, int l) __attribute__((always_inline)) {
    constexpr int sl = decltype(sl_)::value;
    vq[sl] = *reinterpret_cast<const f32x4*>(smem4 + (ch & (2 * CPS - 1)) * VBUF + l * VPOS + v_lane);
  };

  KFN_STAMP4(1);
  // ---- prologue: the workgroup produces super-step 0; every wave fills its B ring --------------------------
  sfor4<36>([&](auto ic) { p_gather(ic, 0); });
  sfor4<NB>([&](auto sc) { b_load(sc, 0, decltype(sc)::value); });
  bt_d_b6(pv, kc);
  sfor4<36>([&](auto gc) { p_store(gc, 0); });
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  KFN_STAMP4(2);
#ifdef KFN_WINO4_PROF
  unsigned long long tl[11];
#endif
  // ---- main loop: one super-step = 2 chunks x 72 MFMA slots ------------------------------------------------
  // slot j of a chunk -> (position l, k-step t): four positions interleaved (groups 0..3), then the last two
  for (int ks = 0; ks < n_super; ++ks) {
    const int c0 = ks * CPS;
    sfor4<NVR>([&](auto gc) { v_read(gc, c0, decltype(gc)::value); });
    sfor4<CPS>([&](auto cc_) {
      constexpr int cc = decltype(cc_)::value;
      const int ch = c0 + cc;
      sfor4<SPC>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int l = j < 64 ? (j >> 4) * 4 + (j & 3) : 16 + ((j - 64) & 1);
        constexpr int t = j < 64 ? (j >> 2) & 3 : (j - 64) >> 1;
        constexpr int qs = cc * WPOS + l;            // fragment index inside the super-step (36 per wave)
        constexpr int sb = qs % NB, sv = qs % NVR;
        // 18 accumulators are 288 registers, the accumulation file holds 256: hipcc gives every MFMA of a function the
        // AGPR form, and accumulators 16 / 17 would be copied in and out around each of their MFMAs (496 v_accvgpr
        // moves per super-step).  Their MFMAs are therefore written with VGPR destinations by hand; between two
        // MFMAs on the same accumulator there is always at least one other MFMA (64 cycles), so the hardware's
        // accumulator forwarding rules are met without software wait states.
        if constexpr (l < 16) {
          acc[l] = __builtin_amdgcn_mfma_f32_32x32x2f32(vq[sv][t], bq[sb][t], acc[l], 0, 0, 0);
        } else {
          const float av = vq[sv][t], bvv = bq[sb][t];
          asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[l]) : "v"(av), "v"(bvv));
        }
        if constexpr (t == 3) {
          // the slots this position held are free: the B fragment one ring ahead, the V fragment NVR positions ahead
          // (same chunk, or the next chunk of THIS super-step -- the next super-step's V is behind the barrier)
          if constexpr (!(KFN_W4_DBG & 8)) {
            if constexpr (l + NB < WPOS) b_load(std::integral_constant<int, sb>{}, ch, l + NB);
            else b_load(std::integral_constant<int, sb>{}, ch + 1, l + NB - WPOS);
          }
          if constexpr (l + NVR < WPOS) v_read(std::integral_constant<int, sv>{}, ch, l + NVR);
          else if constexpr (cc < CPS - 1) v_read(std::integral_constant<int, sv>{}, ch + 1, l + NVR - WPOS);
        }
        constexpr int sj = cc * SPC + j;
#ifdef KFN_WINO4_PROF
        if constexpr (sj % 16 == 0)
          if (ks == KFN_W4_TL_KS) tl[sj / 16] = __builtin_readcyclecounter();
#endif
        if constexpr (!(KFN_W4_DBG & 2) && sj < 36 * KFN_W4_GSTEP && sj % KFN_W4_GSTEP == 0)
          p_gather(std::integral_constant<int, sj / KFN_W4_GSTEP>{}, ks + 1);
        if constexpr (KFN_W4_TSLOT >= 0 && sj >= KFN_W4_TSLOT && sj < KFN_W4_TSLOT + 3 * KFN_W4_TPIECES * KFN_W4_TSTEP &&
                      (sj - KFN_W4_TSLOT) % KFN_W4_TSTEP == 0)
          touch(std::integral_constant<int, (sj >= KFN_W4_TSLOT) ? (sj - KFN_W4_TSLOT) / KFN_W4_TSTEP : 0>{}, ks + 2);
        if constexpr (KFN_W4_XDIST == 2) {
          // quarters: row pass r in slots RS0 + 8 r + q, column pass c in slots CS0 + 4 c + q, q = 0..3 = (half a, Z 0), (half a, Z 1),
          // (half b, Z 0), (half b, Z 1); the six stores of column c in the six slots behind its pass
          if constexpr (!(KFN_W4_DBG & 1)) {
            constexpr int rr = (sj - KFN_W4_RS0) / 8, rq = (sj - KFN_W4_RS0) % 8;
            if constexpr (sj >= KFN_W4_RS0 && rr < 6 && rq < 4) {
              if constexpr (rq == 0) bt6_qa<0>(pv[6 * rr + 1], pv[6 * rr + 2], pv[6 * rr + 3], pv[6 * rr + 4], pv[6 * rr + 5], kc, bt_tmp);
              if constexpr (rq == 1) bt6_qa<1>(pv[6 * rr + 1], pv[6 * rr + 2], pv[6 * rr + 3], pv[6 * rr + 4], pv[6 * rr + 5], kc, bt_tmp);
              if constexpr (rq == 2) bt6_qb<0>(pv[6 * rr], pv[6 * rr + 1], pv[6 * rr + 2], pv[6 * rr + 3], pv[6 * rr + 4], pv[6 * rr + 5], kc, bt_tmp);
              if constexpr (rq == 3) bt6_qb<1>(pv[6 * rr], pv[6 * rr + 1], pv[6 * rr + 2], pv[6 * rr + 3], pv[6 * rr + 4], pv[6 * rr + 5], kc, bt_tmp);
            }
            constexpr int c4 = (sj - KFN_W4_CS0) / 4, cq = (sj - KFN_W4_CS0) % 4;
            if constexpr (sj >= KFN_W4_CS0 && c4 < 6) {
              if constexpr (cq == 0) bt6_qa<0>(pv[6 + c4], pv[12 + c4], pv[18 + c4], pv[24 + c4], pv[30 + c4], kc, bt_tmp);
              if constexpr (cq == 1) bt6_qa<1>(pv[6 + c4], pv[12 + c4], pv[18 + c4], pv[24 + c4], pv[30 + c4], kc, bt_tmp);
              if constexpr (cq == 2) bt6_qb<0>(pv[c4], pv[6 + c4], pv[12 + c4], pv[18 + c4], pv[24 + c4], pv[30 + c4], kc, bt_tmp);
              if constexpr (cq == 3) bt6_qb<1>(pv[c4], pv[6 + c4], pv[12 + c4], pv[18 + c4], pv[24 + c4], pv[30 + c4], kc, bt_tmp);
            }
          }
          if constexpr (!(KFN_W4_DBG & 4) && sj >= KFN_W4_CS0 + 4) {
            // store k of column nu = k / 6 (position 6 (k % 6) + nu) in slot CS0 + 4 nu + 4 + (k % 6)
            sfor4<6>([&](auto nuc) {
              constexpr int nu = decltype(nuc)::value;
              constexpr int xi = sj - (KFN_W4_CS0 + 4 * nu + 4);
              if constexpr (xi >= 0 && xi < 6) p_store(std::integral_constant<int, 6 * xi + nu>{}, ks + 1);
            });
          }
        } else if constexpr (KFN_W4_XDIST == 1) {
          if constexpr (!(KFN_W4_DBG & 1)) {
            constexpr int rr = (sj - KFN_W4_RS0) / 8, rh = (sj - KFN_W4_RS0) % 8;
            if constexpr (sj >= KFN_W4_RS0 && rr < 6 && rh == 0)
              bt6_a(pv[6 * rr + 1], pv[6 * rr + 2], pv[6 * rr + 3], pv[6 * rr + 4], pv[6 * rr + 5], kc, bt_tmp);
            if constexpr (sj >= KFN_W4_RS0 && rr < 6 && rh == 1)
              bt6_b(pv[6 * rr], pv[6 * rr + 1], pv[6 * rr + 2], pv[6 * rr + 3], pv[6 * rr + 4], pv[6 * rr + 5], kc, bt_tmp);
            constexpr int cc6 = (sj - KFN_W4_CS0) / 6, ch6 = (sj - KFN_W4_CS0) % 6;
            if constexpr (sj >= KFN_W4_CS0 && cc6 < 6 && ch6 == 0)
              bt6_a(pv[6 + cc6], pv[12 + cc6], pv[18 + cc6], pv[24 + cc6], pv[30 + cc6], kc, bt_tmp);
            if constexpr (sj >= KFN_W4_CS0 && cc6 < 6 && ch6 == 1)
              bt6_b(pv[cc6], pv[6 + cc6], pv[12 + cc6], pv[18 + cc6], pv[24 + cc6], pv[30 + cc6], kc, bt_tmp);
          }
          // store k = 6 nu + xi (position 6 xi + nu) in slot CS0 + 2 + k: the six of column pass nu right behind it
          if constexpr (!(KFN_W4_DBG & 4) && sj >= KFN_W4_CS0 + 2 && sj < KFN_W4_CS0 + 2 + 36) {
            constexpr int k = sj - (KFN_W4_CS0 + 2);
            p_store(std::integral_constant<int, 6 * (k % 6) + k / 6>{}, ks + 1);
          }
        } else {
          if constexpr (!(KFN_W4_DBG & 1) && sj == KFN_W4_XSLOT) bt_d_b6(pv, kc);
          if constexpr (!(KFN_W4_DBG & 4) && sj >= KFN_W4_SSLOT && sj < KFN_W4_SSLOT + 36)
            p_store(std::integral_constant<int, sj - KFN_W4_SSLOT>{}, ks + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    });
#ifdef KFN_WINO4_PROF
    if (ks == KFN_W4_TL_KS) tl[9] = __builtin_readcyclecounter();
#endif
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#ifdef KFN_WINO4_PROF
    if (ks == KFN_W4_TL_KS) tl[10] = __builtin_readcyclecounter();
#endif
  }
#ifdef KFN_WINO4_PROF
#pragma unroll
  for (int i = 0; i < 11; ++i) p.prof[((size_t)gridDim.x * 4) * 8 + ((size_t)blockIdx.x * 4 + wave) * 11 + i] = tl[i];
#endif
  KFN_STAMP4(3);
  asm volatile("" ::"v"(touched), "v"(tv[0]), "v"(tv[1]), "v"(tv[2]));   // (the touch loads must not be optimised away)

  // ---- epilogue -------------------------------------------------------------------------------------------
  // Partial output transform of this wave's 18 positions, reduced with the partner wave (same column block, other xi half)
  // through the output image [32 tiles][16 px][64 ch] in LDS (every wave is behind the loop's last barrier: V is dead).
  //   pass 1: the rows i of the 4x4 outputs that the PARTNER finishes (wave wx = 0 finishes i in {0,1}, wx = 1 i in {2,3})
  //           are written to the image;  barrier;
  //   pass 2: the rows this wave finishes: own partial + the partner's from the image -> back to the same addresses.
  // (LDS atomics -- ds_add_f32 of both partials into a zeroed image -- were the first version: 220 K cycles per
  //  workgroup, they serialise per lane.)  tile m = (e & 3) + 8 (e >> 2) + 4 lh, channel 32 wc + li:
  //  float index (m * 16 + 4 i + j) * 64 + 32 wc + li
  {
    const f32x2 k2 = {2.f, 2.f}, k4 = {4.f, 4.f}, k8 = {8.f, 8.f};
    float* const img = reinterpret_cast<float*>(smem4) + (lh * 4 * 1024 + wc * 32 + li);
    // rows i0, i0 + 1 of the partial outputs of e-pair e0 for this wave's xi half (LOWER: xi in {0,1,2})
    auto rows = [&](auto lower_half, auto e0c, auto i0c, f32x2 (&P)[2][4]) __attribute__((always_inline)) {
      constexpr bool LOWER = decltype(lower_half)::value;
      constexpr int e0 = decltype(e0c)::value, i0 = decltype(i0c)::value;
      f32x2 R[3][4];     // nu pass: R[a][j] = sum_nu M[a][nu] A^T[j][nu]
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        f32x2 M[6];
#pragma unroll
        for (int nu = 0; nu < 6; ++nu) M[nu] = f32x2{acc[6 * a + nu][e0], acc[6 * a + nu][e0 + 1]};
        const f32x2 s1 = pk_add4(M[1], M[2]), d1 = pk_sub4(M[1], M[2]);
        const f32x2 s2 = pk_add4(M[3], M[4]), d2 = pk_sub4(M[3], M[4]);
        R[a][0] = pk_add4(pk_add4(M[0], s1), s2);
        R[a][1] = pk_fma4(d2, k2, d1);
        R[a][2] = pk_fma4(s2, k4, s1);
        R[a][3] = pk_add4(pk_fma4(d2, k8, d1), M[5]);
      }
      // xi pass over this wave's half: P[i][j] = sum_a A^T[i][xi_a] R[a][j]
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (LOWER) {   // A^T columns 0..2: (1,0,0,0), (1,1,1,1), (1,-1,1,-1)
          if constexpr (i0 == 0) {
            P[0][j] = pk_add4(R[0][j], pk_add4(R[1][j], R[2][j]));
            P[1][j] = pk_sub4(R[1][j], R[2][j]);
          } else {
            P[0][j] = pk_add4(R[1][j], R[2][j]);
            P[1][j] = pk_sub4(R[1][j], R[2][j]);
          }
        } else {                 // A^T columns 3..5: (1,2,4,8), (1,-2,4,-8), (0,0,0,1)
          if constexpr (i0 == 0) {
            P[0][j] = pk_add4(R[0][j], R[1][j]);
            P[1][j] = pk_mul4(pk_sub4(R[0][j], R[1][j]), k2);
          } else {
            P[0][j] = pk_mul4(pk_add4(R[0][j], R[1][j]), k4);
            P[1][j] = pk_fma4(pk_sub4(R[0][j], R[1][j]), k8, R[2][j]);
          }
        }
      }
    };
    auto reduce = [&](auto lower_half) __attribute__((always_inline)) {
      constexpr bool LOWER = decltype(lower_half)::value;
      constexpr int I_MINE = LOWER ? 0 : 2, I_THEIRS = LOWER ? 2 : 0;
      sfor4<8>([&](auto epc) {
        constexpr int e0 = 2 * decltype(epc)::value;
        f32x2 P[2][4];
        rows(lower_half, std::integral_constant<int, e0>{}, std::integral_constant<int, I_THEIRS>{}, P);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int m16 = (((e0 + h) & 3) + 8 * ((e0 + h) >> 2)) * 16;
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) img[(m16 + (I_THEIRS + i) * 4 + j) * 64] = h == 0 ? P[i][j].x : P[i][j].y;
        }
      });
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      KFN_STAMP4(4);
      sfor4<8>([&](auto epc) {
        constexpr int e0 = 2 * decltype(epc)::value;
        f32x2 P[2][4];
        rows(lower_half, std::integral_constant<int, e0>{}, std::integral_constant<int, I_MINE>{}, P);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int m16 = (((e0 + h) & 3) + 8 * ((e0 + h) >> 2)) * 16;
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float* const q = img + (m16 + (I_MINE + i) * 4 + j) * 64;
              *q = *q + (h == 0 ? P[i][j].x : P[i][j].y);
            }
        }
      });
    };
    if (wx == 0) reduce(std::true_type{});
    else reduce(std::false_type{});
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  KFN_STAMP4(5);
  // (3) the image leaves: iteration `it` = tile it, pixel row i = wave, column j = lane >> 4, channel quad lane & 15
  {
    const bool relu = p.relu != 0;
    const unsigned long long y_base = (unsigned long long)img0 * p.H * p.W * p.ldy * 4ull;
    const unsigned long long y_rest = p.y_bytes - y_base;
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(p.y) + y_base, 0, (int)(y_rest < 0x7fffffffull ? y_rest : 0x7fffffffull), 0x00020000);
    const int j = lane >> 4, nq = nbase + 4 * (lane & 15);
    const bool q_ok = nq < p.Cout;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (p.bias != nullptr && q_ok) bv = *reinterpret_cast<const f32x4*>(p.bias + nq);
    const unsigned voff = (unsigned)((j * p.ldy + nq) * 4);
    const int pix_bytes = p.ldy * 4;
#pragma unroll 4
    for (int it = 0; it < 32; ++it) {
      f32x4 v = *reinterpret_cast<const f32x4*>(smem4 + it * 4096 + wave * 1024 + lane * 16);
      v += bv;
      if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      const int tr = it >> 2, tcc = it & 3;
      const int img_rel = tr < brk ? 0 : 1;
      const int ty = tr < brk ? ty0 + tr : tr - brk;
      const int tx = cb * BW4 + tcc;
      const int oy = 4 * ty + wave;
      const bool row_ok = (vr0 + tr < p.vrows) && (tx < p.Tw) && (oy < p.H);     // uniform
      const bool ok = row_ok && q_ok && (4 * tx + j < p.W);
      const unsigned soff = (unsigned)(((img_rel * p.H + oy) * p.W + 4 * tx) * pix_bytes);
      kfn::buffer_store_b128<KFN_NT_STORE_AUX>(v, rsY, ok ? voff : ROW_POISON, row_ok ? soff : 0u);
    }
  }
  KFN_STAMP4(6);
#ifdef KFN_WINO4_PROF
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  KFN_STAMP4(7);
#endif
}


// =====================================================================================================================
// The EIGHT-WAVE form (two waves per SIMD): wino4b_kernel.
// One wave per SIMD hides nothing of its own loads / transform / LDS traffic (timing builds above: 0.91 -> 0.72 of the MFMA
// peak is instruction issue of the producer work).  288 accumulator registers per wave forbid a second wave on 32x32x2
// tiles; on v_mfma_f32_16x16x4_f32 (same FLOP rate, 8 passes) a wave can own 18 positions x 32 tiles x SIXTEEN channels =
// 36 accumulators of 4 registers = 144: eight waves = position half wx (the output reduction stays two-way) x channel
// quarter wq.  Same tile block (4 x 8 tiles x 64 channels), same weight packing, same LDS budget:
//   * V of a chunk: [36 positions][4 k][16 rows][2 tile halves][2 k-steps] floats -- lane (row r, k) of the A operand reads
//     ONE ds_read_b128 per (chunk, position) = both tile halves x both k-steps; input channel of (k, k-step s) = 2 k + s,
//     so the B operand's pair is 8 contiguous bytes of the packed weights.  Rows are stored at r ^ k: the reads stay
//     conflict-free (a read group holds rows {0-3, 12-15} of one k and {4-11} of the next) and the producer's 4-byte stores
//     spread over 16 banks instead of 4.
//   * producer: all eight waves, one CHANNEL per lane (36 patch registers instead of 72, plain instead of packed
//     transform arithmetic): wave w = tile column w & 3, tile rows 4 (w >> 2) .. + 3, lane = (tile row, channel of the
//     super-step's 16) -- 16 lanes read 64 contiguous bytes of a pixel.
//   * weights re-packed per PAIR of positions (graph.pack_winograd_f43_kernel_b): one 16-byte load per lane, chunk and pair.
//   * epilogue: lane (channel, k) holds tiles 16 th + 4 k + e; the partner exchange and the 16-byte image reads as above, the
//     image tile pitch 1028 floats (the four k groups of a store on different banks).
constexpr int B_VPOS = 256;                       // floats per position
constexpr int B_VBUF = NPOS * B_VPOS;             // floats per chunk buffer
constexpr int B_LDS_V = 2 * CPS * B_VBUF * 4;     // 147 456 B
constexpr int B_TILE = 1028;                      // floats per tile of the output image
constexpr int B_LDS_OUT = 32 * B_TILE * 4;        // 131 584 B
constexpr int B_LDS = B_LDS_V > B_LDS_OUT ? B_LDS_V : B_LDS_OUT;
#ifndef KFN_W4B_NB
#define KFN_W4B_NB 6
#endif
constexpr int NBB = KFN_W4B_NB;                   // B ring in position PAIRS (16 bytes per lane and pair)
#ifndef KFN_W4B_NVB
#define KFN_W4B_NVB 2
#endif
constexpr int NVB = KFN_W4B_NVB;                  // V ring (16 bytes per lane and fragment)
static_assert(18 % NBB == 0 && 36 % NVB == 0 && NBB <= WPOS / 2 && NVB <= WPOS, "ring slots are compile-time constants per super-step");
#ifndef KFN_W4B_GSTEP
#define KFN_W4B_GSTEP 2
#define KFN_W4B_XSLOT 100
#define KFN_W4B_SSLOT 106
#endif
static_assert(36 * KFN_W4B_GSTEP <= KFN_W4B_XSLOT && KFN_W4B_XSLOT < KFN_W4B_SSLOT && KFN_W4B_SSLOT + 36 <= CPS * SPC,
              "producer schedule (eight-wave form)");
// timing experiments only (wrong results on purpose): bit 0 no transform, 1 no patch loads, 2 no V stores, 3 no B loads in the loop,
// 4 no output stores (round 6: 4.5 of a workgroup's 16 us of fixed cost; delaying the first round's workgroups by up to 31 x 0.25 / 1 us so
// that the rounds do not store at the same moment changed nothing: the cost is per CU, not a chip-wide burst)
#ifndef KFN_W4B_DBG
#define KFN_W4B_DBG 0
#endif
#ifndef KFN_W4B_STAGGER
#define KFN_W4B_STAGGER 0
#endif
#ifndef KFN_W4B_GPS
#define KFN_W4B_GPS 2     // staggered form: patch loads per slot (36 loads in the first 18 slots of the half)
#define KFN_W4B_HX 56     // ... the transform's slot inside the half
#define KFN_W4B_SPS 3     // ... V stores per slot (36 stores in the 12 slots behind the transform)
#endif
#ifndef KFN_W4B_XDIST
#define KFN_W4B_XDIST 0
#endif
#ifndef KFN_W4B_PRIO
#define KFN_W4B_PRIO 0
#endif
#ifndef KFN_W4B_XOFF
#define KFN_W4B_XOFF 22   // stagger form 2: the late waves' transform at XSLOT + XOFF, their 36 stores two per slot behind it
#endif

// one 1-D pass of B^T on six values, in place (bt6 above, one channel)
__device__ __forceinline__ void bt6s(float& d0, float& d1, float& d2, float& d3, float& d4, float& d5) {
  const float a = __builtin_fmaf(d2, -4.f, d4);
  const float b = __builtin_fmaf(d1, -4.f, d3);
  const float c = d4 - d2;
  const float e = d3 - d1;
  const float u = __builtin_fmaf(d2, -5.f, d4);
  const float v = __builtin_fmaf(d3, -5.f, d5);
  d0 = __builtin_fmaf(d0, 4.f, u);
  d5 = __builtin_fmaf(d1, 4.f, v);
  d1 = a + b;
  d2 = a - b;
  d3 = __builtin_fmaf(e, 2.f, c);
  d4 = __builtin_fmaf(e, -2.f, c);
}
__device__ __forceinline__ void bt_d_b6s(float (&v)[36]) {   // v[6 r + c] -> v[6 xi + nu]
#pragma unroll
  for (int r = 0; r < 6; ++r) bt6s(v[6 * r + 0], v[6 * r + 1], v[6 * r + 2], v[6 * r + 3], v[6 * r + 4], v[6 * r + 5]);
#pragma unroll
  for (int c = 0; c < 6; ++c) bt6s(v[c], v[6 + c], v[12 + c], v[18 + c], v[24 + c], v[30 + c]);
}

// The same transform on ONE channel per lane with PACKED instructions (round 5): 72 v_pk instead of 144 plain ones.
// The patch sits in 18 register pairs, pair (c, m) = rows (2m, 2m + 1) of column c.
//   pass 1, along r, six instructions per column: with P0 = (d0,d1), P1 = (d2,d3), P2 = (d4,d5)
//       (t0,t5) = 4 P0 - 5 P1 + P2                         two fma: the two outer outputs are ONE formula on shifted inputs
//       (a, c)  = d4 - (4,1) d2     (b, e) = d3 - (4,1) d1  one fma each: broadcast halves (op_sel), constant pair (-4,-1)
//       (t1,t3) = (a,c) + (1,2)(b,e)    (t2,t4) = (a,c) - (1,2)(b,e)
//     leaves pairs over the TRANSFORMED row index: (xi 0, 5), (1, 3), (2, 4) -- still one column per pair, so
//   pass 2, along c, is the plain formula (bt6 above, 12 instructions) on three rows of pairs.
// Same products and sums as bt6s in the same roundings (fma(x, 1, y) = x + y, fma(x, -1, y) = y - x); the two passes run in
// the other order than bt_d_b6s, so single results differ from that form in the last bit.
#ifndef KFN_W4B_PACKED
#define KFN_W4B_PACKED 1
#endif
struct BtConstP {
  BtConst k;            // pass 2
  f32x2 m41, p12, m12;  // (-4,-1), (1,2), (-1,-2)
};
__device__ __forceinline__ void bt6r(f32x2& P0, f32x2& P1, f32x2& P2, const BtConstP& k) {
  f32x2 t, ac, be;
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(t) : "v"(P1), "s"(k.k.m5), "v"(P2));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "=v"(ac) : "v"(P1), "s"(k.m41), "v"(P2));   // lo halves: d2, d4
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[1,1,1]" : "=v"(be) : "v"(P0), "s"(k.m41), "v"(P1));   // hi halves: d1, d3
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(P0) : "v"(P0), "s"(k.k.p4), "v"(t));
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(P1) : "v"(be), "s"(k.p12), "v"(ac));
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(P2) : "v"(be), "s"(k.m12), "v"(ac));
}
__device__ __forceinline__ void bt_d_b6p(f32x2 (&pp)[18], const BtConstP& k) {
#pragma unroll
  for (int c = 0; c < 6; ++c) bt6r(pp[3 * c], pp[3 * c + 1], pp[3 * c + 2], k);
#pragma unroll
  for (int m = 0; m < 3; ++m) bt6(pp[m], pp[3 + m], pp[6 + m], pp[9 + m], pp[12 + m], pp[15 + m], k.k);
}
// patch element (r, c) in the pairs, and transformed position (xi, nu) after bt_d_b6p
#define KFN_PP_IN(pp, r, c) (pp)[3 * (c) + (r) / 2][(r) & 1]
#define KFN_PP_OUT(pp, xi, nu) (pp)[3 * (nu) + ((xi) == 0 || (xi) == 5 ? 0 : ((xi) == 1 || (xi) == 3 ? 1 : 2))][((xi) == 5 || (xi) == 3 || (xi) == 4) ? 1 : 0]

// KFN_W4B_PAIR (round 5, experiment): TWO channels and HALF the patch rows per producer lane.  Lane = (row half rh, tile row
// of 4, channel pair of 8): a lane loads rows 3 rh .. 3 rh + 2 of its tile's patch for two adjacent channels -- 18 loads of 8
// bytes instead of 36 of 4 (8 lanes on 64 contiguous bytes of a pixel; an instruction touches 4 tiles x 2 rows) --, runs the
// pass along c on its three rows (packed over the channel pair), trades columns with its partner lane (lane ^ 32, the other
// row half of the same tile and pair) through 18 v_permlane32_swap -- lower lanes end with all six rows of columns 0..2, upper
// lanes of columns 3..5 --, runs the pass along r on its three columns and stores 18 positions x 8 bytes (the pair = the two
// k-steps of one k: adjacent floats of V).  Per wave and super-step: 18 + 18 vector-memory instructions instead of 36 + 18, 18
// ds_write_b64 instead of 36 ds_write_b32, 72 v_pk + 18 swaps.  Same products and sums as bt_d_b6s, in its order.
// MEASURED (profiles/r05_wino4b_pair_ab.log, batch 32, same box, two runs): correct on the first run (all F(4x4) and border
// tests) and 3-4 % SLOWER on every layer (conv4b 5.80 -> 6.00 ms, conv2b 6.41 -> 6.65): a third fewer vector-memory INSTRUCTIONS
// buy nothing when each of them still touches eight separate 64-byte half-lines -- what the gather costs is L1 line
// transactions (4 per old instruction, 8 per new one: the same 144 per wave and super-step), not instruction issue.  OFF.
#ifndef KFN_W4B_PAIR
#define KFN_W4B_PAIR 0
#endif
// cache-policy bits of the patch loads (buffer instruction aux: 1 = sc0, 2 = nt, 3 = both): A/B only.  MEASURED
// (profiles/r05_wino4b_patch_aux_ab.log): sc0 = the same time, nt (alone or with sc0) 17-25 % SLOWER -- the patches of neighbouring
// tiles overlap (6x6 on a 4x4 pitch) and the four tile rows of a wave follow each other: those re-reads are L1 hits that nt gives up.
#ifndef KFN_W4B_PATCH_AUX
#define KFN_W4B_PATCH_AUX 0
#endif
constexpr int P_NG = KFN_W4B_PAIR ? 18 : 36;      // producer: loads / V stores per lane and super-step
constexpr int P_NS = KFN_W4B_PAIR ? 18 : 36;
__device__ __forceinline__ void bt_d_b6q(f32x2 (&q)[3][6], const BtConst& k) {
#pragma unroll
  for (int r = 0; r < 3; ++r) bt6(q[r][0], q[r][1], q[r][2], q[r][3], q[r][4], q[r][5], k);
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int j = 0; j < 3; ++j) {       // upper half of q[r][j] <-> lower half of q[r][j + 3]
      // (scalars first: this clang's __builtin_bit_cast of a vector ELEMENT -- v.y, v[1] -- reads element 0)
      const float ax = q[r][j].x, ay = q[r][j].y, bx = q[r][j + 3].x, by = q[r][j + 3].y;
      const auto sx = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, ax), __builtin_bit_cast(unsigned, bx), false, false);
      const auto sy = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, ay), __builtin_bit_cast(unsigned, by), false, false);
      const unsigned x0 = sx[0], x1 = sx[1], y0 = sy[0], y1 = sy[1];
      q[r][j] = f32x2{__builtin_bit_cast(float, x0), __builtin_bit_cast(float, y0)};
      q[r][j + 3] = f32x2{__builtin_bit_cast(float, x1), __builtin_bit_cast(float, y1)};
    }
#pragma unroll
  for (int j = 0; j < 3; ++j) bt6(q[0][j], q[1][j], q[2][j], q[0][j + 3], q[1][j + 3], q[2][j + 3], k);
}

__global__ __launch_bounds__(512, 1) void wino4b_kernel(Wino4Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem4[];   // [4][B_VBUF] floats, later the output image
  float* const smf = reinterpret_cast<float*>(smem4);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wx = wave & 1, wq = wave >> 1;      // position half, 16-channel quarter
  const int nwg = p.tiles_m * p.tiles_n;
  const int split = p.k_split > 1 ? (int)(blockIdx.x / (unsigned)nwg) : 0;   // K split slowest: a split's workgroups keep the tile order
  const int tile = xcd_remap4((int)blockIdx.x - split * nwg, nwg);
  const int per = p.tiles_m * p.n_group;
  const int gset = tile / per, rem = tile - gset * per;
  const int tm = rem / p.n_group;
  const int tn = gset * p.n_group + (rem - tm * p.n_group);
  const int cb = tm % p.bw, rb = tm / p.bw;
  const int nbase = tn * 64;
  const int n0 = nbase + wq * 16;

  const int vr0 = rb * BH4;
  const int img0 = vr0 / p.Th;
  const int ty0 = vr0 - img0 * p.Th;
  const int brk = (p.Th - ty0 < BH4) ? (p.Th - ty0) : BH4;
  const unsigned long long a_base = (unsigned long long)img0 * p.x_img;
  const unsigned long long a_rest = p.x_bytes - a_base;
  const unsigned long long two_img = 2ull * p.x_img;
  const __amdgpu_buffer_rsrc_t rsU =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.u4), 0, p.u_bytes, 0x00020000);

  // ---- PRODUCER: tile (row ptr = 4 (wave >> 2) + (lane >> 4), column tc = wave & 3), channel c16 = lane & 15 of the super-step's
  // 16 (chunk pch = c16 >> 3): 16 lanes read 64 contiguous bytes of a pixel, a load instruction touches four rows (the 8-channel
  // x 8-row form touched eight 32-byte pieces: twice the L1 transactions of the four-wave kernel) ---------
#if KFN_W4B_PAIR
  const int rh = lane >> 5, cp = lane & 7;
  const int pch = cp >> 2, kq = cp & 3;          // the pair's chunk of the super-step and its k (channels 2 cp, 2 cp + 1 = k-steps 0, 1)
  const int tc = wave & 3, ptr = 4 * (wave >> 2) + ((lane >> 3) & 3);
  unsigned roff[3];                              // patch rows 3 rh + 0 .. 2
#else
  const int c16 = lane & 15, c8 = c16 & 7, pch = c16 >> 3;
  const int tc = wave & 3, ptr = 4 * (wave >> 2) + (lane >> 4);
  unsigned roff[6];
#endif
  unsigned coff[6];
  bool cok[6];
  {
    const int img_rel = ptr < brk ? 0 : 1;
    const int ty = ptr < brk ? ty0 + ptr : ptr - brk;
    const int tx = cb * BW4 + tc;
    const bool row_tile_ok = (vr0 + ptr < p.vrows);
#if KFN_W4B_PAIR
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int yy = 4 * ty - 1 + 3 * rh + r;
      roff[r] = (row_tile_ok && (unsigned)yy < (unsigned)p.H)
                    ? (unsigned)img_rel * p.x_img + (unsigned)(yy * p.W) * p.x_pix + (unsigned)(cp * 8) : ROW_POISON;
    }
#else
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const int yy = 4 * ty - 1 + r;
      roff[r] = (row_tile_ok && (unsigned)yy < (unsigned)p.H)
                    ? (unsigned)img_rel * p.x_img + (unsigned)(yy * p.W) * p.x_pix + (unsigned)(c16 * 4) : ROW_POISON;
    }
#endif
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const int xx = 4 * tx - 1 + c;
      cok[c] = (tx < p.Tw) && ((unsigned)xx < (unsigned)p.W);
      coff[c] = cok[c] ? (unsigned)xx * p.x_pix : 0u;
    }
  }
  const int x_records = (int)(a_rest < two_img ? a_rest : two_img);
  // V store address (floats) inside a super-step's buffer pair: chunk pch, k = c8 >> 1, k-step = c8 & 1, tile t = 4 ptr + tc
  const int pt = 4 * ptr + tc;
#if KFN_W4B_PAIR
  // (both k-steps of k = kq in one 8-byte store; this lane's positions are nu = j + 3 rh: the row half picks the column half)
  const int v_st = pch * B_VBUF + kq * 64 + (((pt & 15) ^ kq) * 4) + (pt >> 4) * 2 + 3 * rh * B_VPOS;
#else
  const int v_st = pch * B_VBUF + (c8 >> 1) * 64 + (((pt & 15) ^ (c8 >> 1)) * 4) + (pt >> 4) * 2 + (c8 & 1);
#endif
  const int n_chunks = p.Cin / 8;
  const int ks0 = split * p.ss_per_split;                                   // first super-step of this split (0 without split-K)
  const int n_super = p.k_split > 1 ? ((n_chunks / CPS - ks0) < p.ss_per_split ? (n_chunks / CPS - ks0) : p.ss_per_split)
                                    : n_chunks / CPS;
  const int s_last = n_super - 1;
  const int c_base = ks0 * CPS;                                             // ... its first chunk

  // ---- CONSUMER: A row rl = lane & 15 (tiles rl, 16 + rl), k = kl = lane >> 4; B column nl = lane & 15 (channel n0 + nl) -------
  const int rl = lane & 15, kl = lane >> 4;
  const int v_lane = (WPOS * wx) * B_VPOS + kl * 64 + ((rl ^ kl) * 4);                 // floats; + l * B_VPOS
  // B: U4b [Cin/8][18 position pairs][cout_pad][4 k][2 positions][2 k-steps] (graph.pack_winograd_f43_kernel_b): one
  // 16-byte load per (chunk, pair of positions) and lane
  const unsigned voff_b = (unsigned)(((n0 + rl) * 16 + kl * 4) * 4);
  const unsigned b_step = (unsigned)p.cout_pad * 64u;
  const int q_last = n_chunks * (NPOS / 2) - 1;

  f32x4 acc[WPOS][2];
#pragma unroll
  for (int l = 0; l < WPOS; ++l)
#pragma unroll
    for (int th = 0; th < 2; ++th) acc[l][th] = f32x4{0.f, 0.f, 0.f, 0.f};

#if KFN_W4B_PAIR
  f32x2 pq[3][6];
  static_assert(KFN_W4B_PACKED && !KFN_W4B_STAGGER, "the pair producer shares the packed constants; no staggered schedule");
#endif
#if KFN_W4B_PACKED
  f32x2 pp[18];
  const BtConstP kp = {{{4.f, 4.f}, {-4.f, -4.f}, {-5.f, -5.f}, {2.f, 2.f}, {-2.f, -2.f}}, {-4.f, -1.f}, {1.f, 2.f}, {-1.f, -2.f}};
  static_assert(!KFN_W4B_XDIST, "the distributed transform exists in the plain form only");
#else
  float pv[36];
#endif
  f32x4 bq[NBB];       // ring of position PAIRS
  f32x4 vq[NVB];

  auto p_gather = [&](auto ic, int ss) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    constexpr int r = i / 6, c = i % 6;
    const int sc = ss < s_last ? ss : s_last;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.x)) + a_base, 0, cok[c] ? x_records : 0, 0x00020000);
#if KFN_W4B_PAIR
    pq[r][c] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, roff[r], coff[c] + (unsigned)(ks0 + sc) * p.x_cb, 0));
#else
    const float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, roff[r], coff[c] + (unsigned)(ks0 + sc) * p.x_cb, KFN_W4B_PATCH_AUX));
#if KFN_W4B_PACKED
    KFN_PP_IN(pp, r, c) = v;
#else
    pv[i] = v;
#endif
#endif
  };
  auto p_store = [&](auto gc, int ss) __attribute__((always_inline)) {
    constexpr int g = decltype(gc)::value;
#if KFN_W4B_PAIR
    constexpr int xi = g / 3, j = g % 3;      // position (xi, nu = j + 3 rh): xi < 3 sits in pq[xi][j], the others in pq[xi - 3][j + 3]
    *reinterpret_cast<f32x2*>(smf + (ss & 1) * (CPS * B_VBUF) + v_st + (6 * xi + j) * B_VPOS) = xi < 3 ? pq[xi % 3][j] : pq[xi % 3][j + 3];
#elif KFN_W4B_PACKED
    smf[(ss & 1) * (CPS * B_VBUF) + v_st + g * B_VPOS] = KFN_PP_OUT(pp, g / 6, g % 6);
#else
    smf[(ss & 1) * (CPS * B_VBUF) + v_st + g * B_VPOS] = pv[g];
#endif
  };
  auto b_load = [&](auto sl_, int ch, int pr) __attribute__((always_inline)) {      // pair pr (0..8) of this wave's positions
    constexpr int sl = decltype(sl_)::value;
    const int q = (c_base + ch) * (NPOS / 2) + (WPOS / 2) * wx + pr;
    const int qc = q < q_last ? q : q_last;
    bq[sl] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsU, voff_b, (unsigned)qc * b_step, 0));
  };
  auto v_read = [&](auto sl_, int ch, int l) __attribute__((always_inline)) {
    constexpr int sl = decltype(sl_)::value;
    vq[sl] = *reinterpret_cast<const f32x4*>(smf + (ch & (2 * CPS - 1)) * B_VBUF + l * B_VPOS + v_lane);
  };

  // ---- prologue ----------------------------------------------------------------------------------------------------
  sfor4<P_NG>([&](auto ic) { p_gather(ic, 0); });
  sfor4<NBB>([&](auto sc) { b_load(sc, 0, decltype(sc)::value); });
#if KFN_W4B_PAIR
  bt_d_b6q(pq, kp.k);
#elif KFN_W4B_PACKED
  bt_d_b6p(pp, kp);
#else
  bt_d_b6s(pv);
#endif
  sfor4<P_NS>([&](auto gc) { p_store(gc, 0); });
  // The two waves of a SIMD (w and w + 4) do their producer work in different HALVES of a super-step (KFN_W4B_STAGGER): in
  // lockstep both would stand in the same transform burst / load group at the same time and the MFMA pipe would idle
  // (timing builds: the transform alone cost 9.5 % that way).  Waves 0-3 gather, transform and store in slots 0..71, waves
  // 4-7 in slots 72..143 -- nothing of the patch lives across the barrier (a half-super-step SHIFT of the late waves did:
  // 36 more live registers, spills in the loop, 108 -> 87 TFLOP/s).
  const bool late = KFN_W4B_STAGGER && wave >= 4;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // ---- main loop: one super-step = 2 chunks x 72 MFMA slots (9 pairs of positions x 2 k-steps x 2 positions x 2 tile halves) ----
  auto super_step = [&](auto latec, int ks) __attribute__((always_inline)) {
    constexpr bool LATE = decltype(latec)::value;
    // producer slots of this wave inside the 144 (staggered: inside its half): KFN_W4B_GPS gathers per slot from G0, the
    // transform at XS, KFN_W4B_SPS stores per slot from S0
    // KFN_W4B_STAGGER: 0 = every wave on one schedule; 1 = waves 0-3 produce in slots 0..71, waves 4-7 in 72..143 (measured
    // worse: the bursts get denser); 2 = the same loads, the late waves' transform burst and stores KFN_W4B_XOFF slots later
    constexpr int G0 = KFN_W4B_STAGGER == 1 ? (LATE ? 72 : 0) : 0;
    constexpr int XS = KFN_W4B_STAGGER == 1 ? G0 + KFN_W4B_HX : KFN_W4B_XSLOT + ((KFN_W4B_STAGGER == 2 && LATE) ? KFN_W4B_XOFF : 0);
    constexpr int S0 = KFN_W4B_STAGGER == 1 ? G0 + KFN_W4B_HX + 2 : (KFN_W4B_STAGGER == 2 && LATE) ? XS + 2 : KFN_W4B_SSLOT;
    constexpr int GPS = KFN_W4B_STAGGER == 1 ? KFN_W4B_GPS : 1, GST = KFN_W4B_STAGGER == 1 ? 1 : KFN_W4B_GSTEP;
    constexpr int SPS = KFN_W4B_STAGGER == 1 ? KFN_W4B_SPS : (KFN_W4B_STAGGER == 2 && LATE) ? 2 : 1;
    static_assert(G0 + GST * (35 / GPS) < XS && XS < S0 && S0 + (35 / SPS) < (KFN_W4B_STAGGER == 1 ? G0 + 72 : 144), "producer schedule");
    const int c0 = ks * CPS;
    sfor4<NVB>([&](auto gc) { v_read(gc, c0, decltype(gc)::value); });
    sfor4<CPS>([&](auto cc_) {
      constexpr int cc = decltype(cc_)::value;
      const int ch = c0 + cc;
      sfor4<SPC>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int pr = j / 8, sk = (j % 8) / 4, lp = ((j % 8) / 2) % 2, th = j % 2;
        constexpr int l = 2 * pr + lp;
        constexpr int qs = cc * WPOS + l;
        constexpr int qp = cc * (WPOS / 2) + pr;       // pair index inside the super-step (18 per wave)
        constexpr int sb = qp % NBB, sv = qs % NVB;
        acc[l][th] = __builtin_amdgcn_mfma_f32_16x16x4f32(vq[sv][2 * th + sk], bq[sb][2 * lp + sk], acc[l][th], 0, 0, 0);
        if constexpr (sk == 1 && th == 1 && lp == 1 && !(KFN_W4B_DBG & 8)) {   // the pair's last MFMA: its B slot is free
          if constexpr (pr + NBB < WPOS / 2) b_load(std::integral_constant<int, sb>{}, ch, pr + NBB);
          else b_load(std::integral_constant<int, sb>{}, ch + 1, pr + NBB - WPOS / 2);
        }
        if constexpr (sk == 1 && th == 1) {      // the last MFMA of fragment (ch, l): its V slot is free
          if constexpr (l + NVB < WPOS) v_read(std::integral_constant<int, sv>{}, ch, l + NVB);
          else if constexpr (cc < CPS - 1) v_read(std::integral_constant<int, sv>{}, ch + 1, l + NVB - WPOS);
        }
        constexpr int sj = cc * SPC + j;
        if constexpr (!(KFN_W4B_DBG & 2) && sj >= G0 && (sj - G0) % GST == 0 && (sj - G0) / GST * GPS < P_NG) {
          sfor4<GPS>([&](auto uc) {
            constexpr int gi = (sj - G0) / GST * GPS + decltype(uc)::value;
            if constexpr (gi < P_NG) p_gather(std::integral_constant<int, gi>{}, ks + 1);
          });
        }
#if !KFN_W4B_PACKED
        if constexpr (KFN_W4B_XDIST) {
          // the transform as 12 passes in 12 different slots (two waves per SIMD: a pass of 12 instructions can sit under the
          // partner's MFMAs, a burst of 144 on both waves at once cannot): row pass r (its loads went out by slot 12 r + 10) at
          // slot 42 + 12 r, column pass c at 104 + 2 c, store k = 6 nu + xi (position 6 xi + nu) at slot 106 + k
          if constexpr (!(KFN_W4B_DBG & 1)) {
            if constexpr (sj >= 42 && (sj - 42) % 12 == 0 && (sj - 42) / 12 < 6) {
              constexpr int r = (sj - 42) / 12;
              bt6s(pv[6 * r], pv[6 * r + 1], pv[6 * r + 2], pv[6 * r + 3], pv[6 * r + 4], pv[6 * r + 5]);
            }
            if constexpr (sj >= 104 && (sj - 104) % 2 == 0 && (sj - 104) / 2 < 6) {
              constexpr int c = (sj - 104) / 2;
              bt6s(pv[c], pv[6 + c], pv[12 + c], pv[18 + c], pv[24 + c], pv[30 + c]);
            }
          }
          if constexpr (!(KFN_W4B_DBG & 4) && sj >= 106 && sj < 142) {
            constexpr int k = sj - 106;
            p_store(std::integral_constant<int, 6 * (k % 6) + k / 6>{}, ks + 1);
          }
        } else
#endif
        {
#if KFN_W4B_PAIR
          if constexpr (!(KFN_W4B_DBG & 1) && sj == XS) bt_d_b6q(pq, kp.k);
#elif KFN_W4B_PACKED
          if constexpr (!(KFN_W4B_DBG & 1) && sj == XS) bt_d_b6p(pp, kp);
#else
          if constexpr (!(KFN_W4B_DBG & 1) && sj == XS) bt_d_b6s(pv);
#endif
          if constexpr (!(KFN_W4B_DBG & 4) && sj >= S0 && (sj - S0) * SPS < P_NS) {
            sfor4<SPS>([&](auto uc) {
              constexpr int gi = (sj - S0) * SPS + decltype(uc)::value;
              if constexpr (gi < P_NS) p_store(std::integral_constant<int, gi>{}, ks + 1);
            });
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
#if KFN_W4B_PRIO
  // static priority for the second-dispatched half: at equal priority the issue arbitration goes by age and waves 4-7 lose every
  // contested slot (MI355X_MICROARCH: two waves per SIMD, item 4)
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
  if (late) {
    for (int ks = 0; ks < n_super; ++ks) super_step(std::true_type{}, ks);
  } else {
    for (int ks = 0; ks < n_super; ++ks) super_step(std::false_type{}, ks);
  }
#if KFN_W4B_PRIO
  __builtin_amdgcn_s_setprio(0);
#endif

  // ---- epilogue: partial output transform of this wave's 18 positions, exchange with the partner (other xi half, same channel
  // quarter = wave ^ 1) through the image [32 tiles][B_TILE], as in wino4_kernel.  Lane (nl, kl): tiles 16 th + 4 kl + e. ----
  {
    const f32x2 k2 = {2.f, 2.f}, k4 = {4.f, 4.f}, k8 = {8.f, 8.f};
    float* const img = smf + (4 * kl) * B_TILE + wq * 16 + rl;
    auto rows = [&](auto lower_half, auto thc, auto e0c, auto i0c, f32x2 (&P)[2][4]) __attribute__((always_inline)) {
      constexpr bool LOWER = decltype(lower_half)::value;
      constexpr int th = decltype(thc)::value, e0 = decltype(e0c)::value, i0 = decltype(i0c)::value;
      f32x2 R[3][4];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        f32x2 M[6];
#pragma unroll
        for (int nu = 0; nu < 6; ++nu) M[nu] = f32x2{acc[6 * a + nu][th][e0], acc[6 * a + nu][th][e0 + 1]};
        const f32x2 s1 = pk_add4(M[1], M[2]), d1 = pk_sub4(M[1], M[2]);
        const f32x2 s2 = pk_add4(M[3], M[4]), d2 = pk_sub4(M[3], M[4]);
        R[a][0] = pk_add4(pk_add4(M[0], s1), s2);
        R[a][1] = pk_fma4(d2, k2, d1);
        R[a][2] = pk_fma4(s2, k4, s1);
        R[a][3] = pk_add4(pk_fma4(d2, k8, d1), M[5]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (LOWER) {
          if constexpr (i0 == 0) {
            P[0][j] = pk_add4(R[0][j], pk_add4(R[1][j], R[2][j]));
            P[1][j] = pk_sub4(R[1][j], R[2][j]);
          } else {
            P[0][j] = pk_add4(R[1][j], R[2][j]);
            P[1][j] = pk_sub4(R[1][j], R[2][j]);
          }
        } else {
          if constexpr (i0 == 0) {
            P[0][j] = pk_add4(R[0][j], R[1][j]);
            P[1][j] = pk_mul4(pk_sub4(R[0][j], R[1][j]), k2);
          } else {
            P[0][j] = pk_mul4(pk_add4(R[0][j], R[1][j]), k4);
            P[1][j] = pk_fma4(pk_sub4(R[0][j], R[1][j]), k8, R[2][j]);
          }
        }
      }
    };
    auto reduce = [&](auto lower_half) __attribute__((always_inline)) {
      constexpr bool LOWER = decltype(lower_half)::value;
      constexpr int I_MINE = LOWER ? 0 : 2, I_THEIRS = LOWER ? 2 : 0;
      sfor4<4>([&](auto qc) {
        constexpr int th = decltype(qc)::value / 2, e0 = 2 * (decltype(qc)::value % 2);
        f32x2 P[2][4];
        rows(lower_half, std::integral_constant<int, th>{}, std::integral_constant<int, e0>{}, std::integral_constant<int, I_THEIRS>{}, P);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
              img[(16 * th + e0 + h) * B_TILE + ((I_THEIRS + i) * 4 + j) * 64] = h == 0 ? P[i][j].x : P[i][j].y;
      });
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      sfor4<4>([&](auto qc) {
        constexpr int th = decltype(qc)::value / 2, e0 = 2 * (decltype(qc)::value % 2);
        f32x2 P[2][4];
        rows(lower_half, std::integral_constant<int, th>{}, std::integral_constant<int, e0>{}, std::integral_constant<int, I_MINE>{}, P);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float* const q = img + (16 * th + e0 + h) * B_TILE + ((I_MINE + i) * 4 + j) * 64;
              *q = *q + (h == 0 ? P[i][j].x : P[i][j].y);
            }
      });
    };
    if (wx == 0) reduce(std::true_type{});
    else reduce(std::false_type{});
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  // the image leaves: iteration `it` = tiles 2 it, 2 it + 1 (wave >> 2), pixel row i = wave & 3, column j = lane >> 4, channel quad lane & 15
  {
    const bool relu = p.relu != 0;
    const unsigned long long y_base = (KFN_W4B_DBG & 64) ? 0ull : (unsigned long long)img0 * p.y_img;
    const unsigned long long y_rest = p.y_bytes - y_base;
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(p.y) + y_base + (unsigned long long)split * p.y_split_bytes, 0,
        (int)(y_rest < 0x7fffffffull ? y_rest : 0x7fffffffull), 0x00020000);
    const int j = lane >> 4, nq = nbase + 4 * (lane & 15);
    const int pi = wave & 3, tsel = wave >> 2;
    const bool q_ok = nq < p.Cout;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (p.bias != nullptr && q_ok) bv = *reinterpret_cast<const f32x4*>(p.bias + nq);
    const unsigned voff = (unsigned)j * p.y_pix + (unsigned)(nq >> 4) * p.y_cb + (unsigned)((nq & 15) * 4);
    const unsigned pix_bytes = p.y_pix;
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
      const int t = 2 * it + tsel;
      f32x4 v = *reinterpret_cast<const f32x4*>(smf + t * B_TILE + (4 * pi + j) * 64 + 4 * (lane & 15));
      v += bv;
      if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      const int tr = t >> 2, tcc = t & 3;
      const int img_rel = tr < brk ? 0 : 1;
      const int ty = tr < brk ? ty0 + tr : tr - brk;
      const int tx = cb * BW4 + tcc;
      const int oy = 4 * ty + pi;
      const bool row_ok = (vr0 + tr < p.vrows) && (tx < p.Tw) && (oy < p.H);     // uniform
      const bool ok = row_ok && q_ok && (4 * tx + j < p.W);
      unsigned soff = (unsigned)img_rel * p.y_img + (unsigned)(oy * p.W + 4 * tx) * pix_bytes;
      if (KFN_W4B_DBG & 64) soff = (unsigned)((4 * tr + pi) * p.W + 4 * tcc + 16 * (blockIdx.x & 255)) * pix_bytes;
      // (timing bits: 4 no output stores; 5 only every other one; 6 every workgroup of a CU-sized group writes the same 131 KB)
      if ((KFN_W4B_DBG & 32) && (it & 1)) continue;
      if (!(KFN_W4B_DBG & 16)) kfn::buffer_store_b128<KFN_NT_STORE_AUX>(v, rsY, ok ? voff : ROW_POISON, row_ok ? soff : 0u);
    }
  }
}

// y[pix][c] = relu(bias[c] + sum_s ws[s][pix][c]), s = 0 .. k_split-1 in THAT order (one fixed order: deterministic, and the
// same for every launch geometry).  ws planes are dense [pixels][Cout]; one float4 per thread.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const f32x4* __restrict__ ws, int k_split, long plane_quads,
                                                                  const float* __restrict__ bias, int relu, float* __restrict__ y,
                                                                  int Cout, int ldy, long pixels) {
  const int cq = Cout >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < plane_quads; i += (long)gridDim.x * 256) {
    const long pix = i / cq;
    const int c = (int)(i - pix * cq) * 4;
    f32x4 acc = __builtin_nontemporal_load(ws + i);
    for (int s2 = 1; s2 < k_split; ++s2) acc += __builtin_nontemporal_load(ws + (long)s2 * plane_quads + i);
    if (bias != nullptr) acc += *reinterpret_cast<const f32x4*>(bias + c);
    if (relu) { acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f); }
    *reinterpret_cast<f32x4*>(y + pix * ldy + c) = acc;
  }
}

}  // namespace

#ifdef KFN_WINO4_PROF
unsigned long long* g_wino4_prof = nullptr;
#endif

int kfn::wino_f43_lds_bytes(int wino_form) {
  if (wino_form == KFN_WINO_FORM_F43_FOUR_WAVE) return LDS_V;
  if (wino_form == KFN_WINO_FORM_F43_EIGHT_WAVE) return B_LDS;
  return -1;
}

// Dynamic LDS per workgroup of the kernel a Winograd entry point launches for `desc` (stride 1: kfn_conv2d_winograd_f43
// when wino_form names one of its two forms, else kfn_conv2d_winograd_fused's routing; stride 2: kfn_conv2d_winograd_s2),
// so that a host can route around kernels a device's LDS cannot hold (kfn_device_info's lds_bytes_per_cu) instead of
// failing in the first launch.  No device access.
extern "C" int kfn_winograd_lds_bytes(const kfn_conv_desc* d, int* bytes) {
  KFN_REQUIRE(d && bytes, "kfn_winograd_lds_bytes: null argument");
  KFN_CONV_DESC_IN_LAYOUTS(d, "kfn_winograd_lds_bytes");
  KFN_REQUIRE(d->kh == 3 && d->kw == 3 && (d->stride == 1 || d->stride == 2) && !d->transposed,
              "kfn_winograd_lds_bytes: the Winograd kernels take 3x3 stride-1 / stride-2 convolutions only");
  int b;
  if (d->stride == 2) b = kfn::wino_s2_lds_bytes(d->wino_form, d->operand_dtype);
  else if (d->wino_form == KFN_WINO_FORM_F43_FOUR_WAVE || d->wino_form == KFN_WINO_FORM_F43_EIGHT_WAVE) b = kfn::wino_f43_lds_bytes(d->wino_form);
  else b = kfn::wino_fused_lds_bytes(d);
  if (b < 0) return kfn::fail(KFN_ERR_UNSUPPORTED, "kfn_winograd_lds_bytes: no kernel for wino_form %d at stride %d", d->wino_form, d->stride);
  *bytes = b;
  return KFN_OK;
}

// Can the F(4x4,3x3) kernel take this layer?  (host-side routing; no device access.  Mirrors every check of the launcher
// that does not need the pointers: what remains there is the 16-byte alignment of y / u4_packed / bias and 8-byte of x.)
extern "C" int kfn_winograd_f43_supported(const kfn_conv_desc* d) {
  kfn_conv_desc d_full;
  if (!d || kfn::conv_desc_in(d, &d_full, "kfn_winograd_f43_supported", true) != KFN_OK) return 0;
  d = &d_full;
  if (d->kh != 3 || d->kw != 3 || d->stride != 1 || d->transposed) return 0;
  if (d->x_layout != KFN_LAYOUT_NHWC || d->y_layout != KFN_LAYOUT_NHWC) {       // channel-blocked: the eight-wave kernel, dense tensors
    if (d->wino_form == KFN_WINO_FORM_F43_FOUR_WAVE || (d->wino_form == 0 && !KFN_W4_DEFAULT_EIGHT_WAVE)) return 0;
    if (d->x_layout == KFN_LAYOUT_C16 && d->ldx != d->Cin) return 0;
    if (d->y_layout == KFN_LAYOUT_C16 && (d->Cout % 16 != 0 || d->ldy != d->Cout)) return 0;
  }
  if (d->x_dtype != KFN_ACT_F32 || d->y_dtype != KFN_ACT_F32 || d->operand_dtype != KFN_OPERAND_F32) return 0;
  if (d->Cin <= 0 || d->Cin % 16 != 0) return 0;
  if ((d->H + 3) / 4 < BH4) return 0;               // an 8-row tile block may straddle at most two images
  if (d->epilogue != KFN_EPI_NONE) return 0;
  if (d->N <= 0 || d->W <= 0 || d->Cout <= 0) return 0;
  if (d->cout_pad % 32 != 0 || d->cout_pad < d->Cout || d->Cout % 4 != 0 || d->ldy % 4 != 0 || d->ldx % 2 != 0) return 0;
  if (d->ldx < d->Cin || d->ldy < d->Cout) return 0;
  if (2L * d->H * d->W * d->ldx * 4L >= (1L << 30)) return 0;            // two images of the input below 1 GiB
  if (2L * d->H * d->W * d->ldy * 4L >= (1L << 31)) return 0;            // ... of the output below 2 GiB
  if (36L * d->cout_pad * d->Cin * 4L >= (1L << 31)) return 0;           // the transformed kernel below 2 GiB
  if ((long)d->N * ((d->H + 3) / 4) >= (1L << 30)) return 0;
  return 1;
}

namespace {

// Argument checks + geometry shared by kfn_conv2d_winograd_f43 and its split-K form.  `y` / `ldy` = where the kernel writes
// (the output tensor, or one dense plane of the split-K workspace).
int wino4_setup(const kfn_conv_desc* d, const float* x, const float* u4_packed, const float* bias, float* y, int ldy,
                const char* who, Wino4Args* out) {
  KFN_REQUIRE(d->kh == 3 && d->kw == 3 && d->stride == 1 && !d->transposed, "%s: only 3x3 stride-1 SAME convolutions", who);
  KFN_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0, "%s: bad shape %dx%dx%d", who, d->N, d->H, d->W);
  KFN_REQUIRE(d->x_dtype == KFN_ACT_F32 && d->y_dtype == KFN_ACT_F32 && d->operand_dtype == KFN_OPERAND_F32 &&
                  d->epilogue == KFN_EPI_NONE,
              "%s: fp32 operands and activations, no fused head epilogue", who);
  if (d->Cin <= 0 || d->Cin % 16 != 0)
    return kfn::fail(KFN_ERR_UNSUPPORTED, "%s: Cin=%d must be a multiple of 16", who, d->Cin);
  if ((d->H + 3) / 4 < BH4)
    return kfn::fail(KFN_ERR_UNSUPPORTED, "%s: H=%d is below %d rows", who, d->H, 4 * BH4 - 3);
  KFN_REQUIRE(d->ldx >= d->Cin && d->ldx % 2 == 0 && d->Cout > 0 && ldy >= d->Cout && d->cout_pad >= d->Cout &&
                  d->cout_pad % 32 == 0,
              "%s: bad strides / channel counts", who);
  if (d->Cout % 4 != 0 || ldy % 4 != 0 || (reinterpret_cast<uintptr_t>(y) & 15) != 0)
    return kfn::fail(KFN_ERR_UNSUPPORTED, "%s: Cout and ldy must be multiples of 4 and y 16-byte aligned", who);
  KFN_REQUIRE(((reinterpret_cast<uintptr_t>(x) & 7) | (reinterpret_cast<uintptr_t>(u4_packed) & 15) |
               (bias ? (reinterpret_cast<uintptr_t>(bias) & 15) : 0)) == 0,
              "%s: x must be 8-byte, u4_packed and bias 16-byte aligned", who);
  const long img_b = (long)d->H * d->W * d->ldx * 4L;
  const long out_b = (long)d->H * d->W * ldy * 4L;
  if (2 * img_b >= (1L << 30) || 2 * out_b >= (1L << 31) || 36L * d->cout_pad * d->Cin * 4L >= (1L << 31))
    return kfn::fail(KFN_ERR_UNSUPPORTED, "%s: image or kernel beyond the 32-bit offsets of this form", who);
  Wino4Args a;
  a.x = x; a.u4 = u4_packed; a.bias = bias; a.y = y;
  a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.ldx = d->ldx;
  a.Cout = d->Cout; a.cout_pad = d->cout_pad; a.ldy = ldy;
  a.Th = (d->H + 3) / 4; a.Tw = (d->W + 3) / 4;
  const long vrows = (long)d->N * a.Th;
  KFN_REQUIRE(vrows < (1L << 30), "%s: N*ceil(H/4) = %ld tile rows exceed 32-bit addressing", who, vrows);
  a.vrows = (int)vrows;
  a.bw = kfn::ceil_div(a.Tw, BW4);
  const long tiles_m = (long)a.bw * kfn::ceil_div(a.vrows, BH4);
  a.tiles_n = kfn::ceil_div(d->cout_pad, 64);
  KFN_REQUIRE(tiles_m * a.tiles_n < (1L << 31), "%s: grid too large", who);
  a.tiles_m = (int)tiles_m;
  a.relu = d->relu;
  // workgroup order: M fastest (1 group), all channel groups of a block adjacent (N fastest), or KFN_WINO_ORDER_GROUPS(n).
  // AUTO: four groups (two when the layer has fewer) -- the input block then crosses the fabric a quarter as often and a
  // weight slice is still shared by 8 CUs of the XCD: conv4b 6.28 -> 6.07 ms, conv5 3.26 -> 3.19, conv3b 6.46 -> 6.36,
  // conv2b equal (same log); 8 groups and N fastest lose again (the weight slices no longer share an L2).
  int ng = 1;
  if (d->wino_order == KFN_WINO_ORDER_AUTO) ng = a.tiles_n % 4 == 0 ? 4 : (a.tiles_n % 2 == 0 ? 2 : 1);
  else if (d->wino_order == KFN_WINO_ORDER_N_FAST) ng = a.tiles_n;
  else if (d->wino_order >= KFN_WINO_ORDER_GROUPS(1)) ng = d->wino_order - KFN_WINO_ORDER_GROUPS(0);
  if (ng < 1 || ng > a.tiles_n || a.tiles_n % ng != 0) ng = 1;
  a.n_group = ng;
  const long in_pix = (long)d->N * d->H * d->W;
  a.x_bytes = (unsigned long long)(((in_pix - 1) * d->ldx + d->Cin) * 4L);
  a.y_bytes = (unsigned long long)(((in_pix - 1) * ldy + d->Cout) * 4L);
  // the B ring prefetches whole 1 KiB fragments of 32 output channels: the last column block of a 32-but-not-64-multiple
  // cout_pad reads 32 channels past the matrix -- the range check returns zeros for them (their accumulators are never stored)
  a.u_bytes = (unsigned)(36L * d->cout_pad * d->Cin * 4L);
  const bool xb = d->x_layout == KFN_LAYOUT_C16, yb = d->y_layout == KFN_LAYOUT_C16;
  if (xb && d->ldx != d->Cin)
    return kfn::fail(KFN_ERR_UNSUPPORTED, "%s: a KFN_LAYOUT_C16 input is dense (ldx=%d, Cin=%d)", who, d->ldx, d->Cin);
  if (yb && (d->Cout % 16 != 0 || ldy != d->Cout))
    return kfn::fail(KFN_ERR_UNSUPPORTED, "%s: a KFN_LAYOUT_C16 output is dense and Cout %% 16 == 0 (ldy=%d, Cout=%d)", who, ldy, d->Cout);
  a.x_img = (unsigned)img_b; a.x_pix = xb ? 64u : (unsigned)d->ldx * 4u; a.x_cb = xb ? (unsigned)(d->H * d->W) * 64u : 64u;
  a.y_img = (unsigned)out_b; a.y_pix = yb ? 64u : (unsigned)ldy * 4u; a.y_cb = yb ? (unsigned)(d->H * d->W) * 64u : 64u;
  a.k_split = 1;
  a.ss_per_split = d->Cin / (8 * CPS);
  a.y_split_bytes = 0;
#ifdef KFN_WINO4_PROF
  a.prof = g_wino4_prof;
#endif
  *out = a;
  return KFN_OK;
}

int wino4b_launch(const Wino4Args& a, hipStream_t stream) {
  static std::atomic<uint64_t> attr_done_b{0};
  int rcb = kfn::set_max_dynamic_lds(reinterpret_cast<const void*>(wino4b_kernel), B_LDS, attr_done_b);
  if (rcb != KFN_OK) return rcb;
  hipLaunchKernelGGL(wino4b_kernel, dim3((unsigned)((long)a.tiles_m * a.tiles_n * a.k_split)), dim3(512), B_LDS, stream, a);
  KFN_LAUNCH_CHECK("wino4b_kernel");
  return KFN_OK;
}

}  // namespace

extern "C" int kfn_conv2d_winograd_f43(const kfn_conv_desc* d, const float* x, const float* u4_packed, const float* bias,
                                       float* y, void* stream) {
  KFN_REQUIRE(d && x && u4_packed && y, "kfn_conv2d_winograd_f43: null argument");
  KFN_CONV_DESC_IN_LAYOUTS(d, "kfn_conv2d_winograd_f43");
  Wino4Args a;
  int rc = wino4_setup(d, x, u4_packed, bias, y, d->ldy, "kfn_conv2d_winograd_f43", &a);
  if (rc != KFN_OK) return rc;
  // kfn_conv_desc.wino_form: KFN_WINO_FORM_F43_FOUR_WAVE / _EIGHT_WAVE pick the kernel (A/B measurements); AUTO = the default below
  const bool eight = d->wino_form == KFN_WINO_FORM_F43_EIGHT_WAVE ||
                     (d->wino_form != KFN_WINO_FORM_F43_FOUR_WAVE && KFN_W4_DEFAULT_EIGHT_WAVE);
  if (!eight && (d->x_layout != KFN_LAYOUT_NHWC || d->y_layout != KFN_LAYOUT_NHWC))
    return kfn::fail(KFN_ERR_UNSUPPORTED, "kfn_conv2d_winograd_f43: the four-wave form reads and writes NHWC only");
  if (eight) return wino4b_launch(a, (hipStream_t)stream);
  static std::atomic<uint64_t> attr_done{0};
  rc = kfn::set_max_dynamic_lds(reinterpret_cast<const void*>(wino4_kernel), LDS_V, attr_done);
  if (rc != KFN_OK) return rc;
  hipLaunchKernelGGL(wino4_kernel, dim3((unsigned)(a.tiles_m * a.tiles_n)), dim3(256), LDS_V, (hipStream_t)stream, a);
  KFN_LAUNCH_CHECK("wino4_kernel");
  return KFN_OK;
}

// ---- split-K form (round 5: BASELINE configs[1], one 480x640 frame): the layers whose launch has fewer workgroups than
// the chip has CUs (conv4b 160, conv5 80, conv6 40 at batch 1) split their input channels over k_split copies of the tile
// grid.  Copy s writes raw partial sums into plane s of `workspace` ([k_split][N*H*W][Cout] floats, dense); a second kernel
// adds the planes in the order 0, 1, .., k_split-1, the bias, the ReLU and writes y: two launches, no atomics, bit-stable.
extern "C" int kfn_winograd_f43_splitk_workspace_bytes(const kfn_conv_desc* d, int k_split, size_t* bytes) {
  KFN_REQUIRE(d && bytes, "kfn_winograd_f43_splitk_workspace_bytes: null argument");
  KFN_CONV_DESC_IN(d, "kfn_winograd_f43_splitk_workspace_bytes");
  KFN_REQUIRE(k_split >= 1 && d->N > 0 && d->H > 0 && d->W > 0 && d->Cout > 0, "kfn_winograd_f43_splitk_workspace_bytes: bad argument");
  *bytes = k_split > 1 ? (size_t)k_split * d->N * d->H * d->W * d->Cout * sizeof(float) : 0;
  return KFN_OK;
}

extern "C" int kfn_conv2d_winograd_f43_splitk(const kfn_conv_desc* d, const float* x, const float* u4b_packed, const float* bias,
                                              float* y, float* workspace, int k_split, void* stream) {
  KFN_REQUIRE(d && x && u4b_packed && y, "kfn_conv2d_winograd_f43_splitk: null argument");
  KFN_CONV_DESC_IN(d, "kfn_conv2d_winograd_f43_splitk");
  KFN_REQUIRE(d->wino_form == KFN_WINO_FORM_AUTO || d->wino_form == KFN_WINO_FORM_F43_EIGHT_WAVE,
              "kfn_conv2d_winograd_f43_splitk: the eight-wave form only (weights packed per pair of positions)");
  const int n_super = d->Cin > 0 ? d->Cin / (8 * CPS) : 0;
  KFN_REQUIRE(k_split >= 1 && k_split <= (n_super > 0 ? n_super : 1), "kfn_conv2d_winograd_f43_splitk: k_split=%d outside 1..%d (Cin/16)",
              k_split, n_super);
  Wino4Args a;
  if (k_split == 1) {
    int rc = wino4_setup(d, x, u4b_packed, bias, y, d->ldy, "kfn_conv2d_winograd_f43_splitk", &a);
    if (rc != KFN_OK) return rc;
    return wino4b_launch(a, (hipStream_t)stream);
  }
  KFN_REQUIRE(workspace != nullptr && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0 && d->ldy >= d->Cout && d->ldy % 4 == 0 &&
                  (reinterpret_cast<uintptr_t>(y) & 15) == 0,
              "kfn_conv2d_winograd_f43_splitk: k_split > 1 needs a 16-byte aligned workspace of kfn_winograd_f43_splitk_workspace_bytes() "
              "and a 16-byte aligned output with ldy %% 4 == 0");
  int rc = wino4_setup(d, x, u4b_packed, nullptr, workspace, d->Cout, "kfn_conv2d_winograd_f43_splitk", &a);
  if (rc != KFN_OK) return rc;
  const long pixels = (long)d->N * d->H * d->W;
  a.relu = 0;
  a.k_split = k_split;
  a.ss_per_split = kfn::ceil_div(n_super, k_split);
  KFN_REQUIRE((long)(k_split - 1) * a.ss_per_split < n_super, "kfn_conv2d_winograd_f43_splitk: k_split=%d leaves an empty split of %d super-steps",
              k_split, n_super);
  a.y_split_bytes = (unsigned long long)pixels * d->Cout * 4ull;
  KFN_REQUIRE((long)a.tiles_m * a.tiles_n * k_split < (1L << 31), "kfn_conv2d_winograd_f43_splitk: grid too large");
  rc = wino4b_launch(a, (hipStream_t)stream);
  if (rc != KFN_OK) return rc;
  return kfn::launch_splitk_reduce(workspace, k_split, pixels, bias, d->relu, y, d->Cout, d->ldy, (hipStream_t)stream);
}

int kfn::launch_splitk_reduce(const float* ws, int k_split, long pixels, const float* bias, int relu, float* y, int Cout, int ldy,
                              hipStream_t stream) {
  const long quads = pixels * (Cout / 4);
  long blocks = (quads + 255) / 256;
  if (blocks > 256L * 8) blocks = 256L * 8;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, stream,
                     reinterpret_cast<const f32x4*>(ws), k_split, quads, bias, relu, y, Cout, ldy, pixels);
  KFN_LAUNCH_CHECK("splitk_reduce_kernel");
  return KFN_OK;
}
