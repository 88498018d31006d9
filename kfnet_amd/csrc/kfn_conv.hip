// kfn_conv.hip -- fp32 implicit-GEMM convolution on v_mfma_f32_32x32x2_f32 (gfx950).
//
// Replaces tf.layers.conv2d / conv2d_transpose behind Network.conv / Network.deconv
// (cnn_wrapper/network.py:116-135, 418-437).  M = N*Ho*Wo output pixels, N = Cout,
// K = taps*Cin.  Both operands are staged through LDS "K-contiguous":
//   A tile [BM][BK]  = im2col rows gathered on the fly from the NHWC activations
//   B tile [BN][BK]  = rows of the pre-packed weight matrix w_packed[Cout][K]
// LDS rows are unpadded; the 16-byte quad q of row r is stored at quad q ^ sw(r)
// (sw(r) = (r>>1)&7 for BK=32, (r>>2)&3 for BK=16), which makes both the ds_write_b128
// staging stores and the ds_read_b128 fragment reads (lane (i,h) reads row i, floats
// [8c+4h, 8c+4h+4)) bank-conflict free, and keeps a 160x128x32 double-buffered tile at
// 72 KiB so TWO workgroups (2 waves per SIMD) fit in the 160 KiB LDS of a CU: one
// wave's staging/barrier gaps are filled by the other wave's MFMAs.
// Each 8-wide k-chunk feeds four 32x32x2 MFMAs per (mi,ni) tile: MFMA step t takes
// k = 8c + 4h + t from lane half h -- A and B use the same assignment, so the k
// permutation is harmless.  Global loads are raw buffer loads (hardware range check
// gives the zero padding for free: invalid taps use an out-of-range offset); loads of
// stage s+1 are issued before the MFMAs of stage s (register-staged double buffer, one
// barrier per stage).
#include "kfn_common.h"
#include <type_traits>
#include <cstdlib>

// fp16-activation kernels: pin the issue order of the main loop (one sched_barrier per MFMA slot).  Without it hipcc
// re-clusters the stage -- 8 MFMAs, then ALL fragment reads of the next chunk in one burst with the LDS stores and
// global loads behind them, then a wait for those reads in front of the next MFMA -- which exposes one LDS round trip
// per chunk (tools/mb/build_hot.sh builds the A/B library with -DKFN_F16_PIN=0).
#ifndef KFN_F16_PIN
#define KFN_F16_PIN 1
#endif
// position of the stage barrier inside the last k-chunk of a stage: after J / KFN_F16_BAR_DIV of its J MFMAs (fp16-activation
// kernels; everything else keeps J / 2).  The fragment reads of the next stage follow the barrier and are covered by the rest.
// Same-box A/B on the eight-wave 256x256 tile (profiles/r04_c5_layer_microbench.log, conv2b / 3b / 4b / 5, TFLOP/s):
// J/2 with both operands on LDS-DMA 1090 / 1153 / 1188 / 1201, J/4: 1123 / 1180 / 1197 / 1215.
#ifndef KFN_F16_BAR_DIV
#define KFN_F16_BAR_DIV 4
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned OOB = 0x80000000u;  // voffset that always fails the buffer range check

// Exact unsigned division by an invariant (Granlund-Montgomery, branch-free form):
// 5 VALU instructions instead of the ~30 of the emulated 32-bit divide.  The row -> pixel
// decode runs once per tile and thread, which is a visible share of a short-K tile.
struct FastDiv {
  unsigned d, mul, sh;
};
inline FastDiv make_fastdiv(int d) {
  FastDiv f{(unsigned)d, 0u, 0u};
  if (d > 1) {
    int L = 0;
    while ((1u << L) < (unsigned)d) ++L;   // ceil(log2 d), d < 2^31
    f.mul = (unsigned)((((unsigned long long)1 << 32) * (((unsigned long long)1 << L) - (unsigned)d)) / (unsigned)d + 1ull);
    f.sh = (unsigned)(L - 1);
  }
  return f;
}
__device__ __forceinline__ int fdiv(int n, const FastDiv& f) {
  const unsigned un = (unsigned)n;
  const unsigned t = __umulhi(f.mul, un);
  const unsigned q = (t + ((un - t) >> 1)) >> f.sh;
  return (int)(f.d == 1u ? un : q);
}

struct ConvArgs {
  const float* x;
  const float* x2;  // MODE_CVOL: f2 (x is f1)
  const float* w;
  const float* bias;
  float* y;
  int N, H, W, Cin, ldx;
  int Ho, Wo, Cout, cout_pad, ldy;
  int kh, kw, stride, pad_t, pad_l;
  unsigned kw_inv;   // ceil(65536 / kw): tap / kw == (tap * kw_inv) >> 16 for tap < 32 (a scalar integer divide is ~20 SALU instructions per stage)
  int relu, epilogue;
  int M, Ktot;
  int tiles_m, tiles_n;
  unsigned long long x_bytes;
  unsigned w_bytes;
  int rot_mode;  // K-chunk rotation: 0 off, 1 per (m,n) tile, 2 per (m,n,group)
  unsigned w_lo_bytes;  // f16x3: byte offset of the lo weight matrix behind the hi one
  float out_scale;      // f16x3: 2^-k undoing the weight pre-scale
  FastDiv fd_cls, fd_img, fd_row;  // row index -> (parity class,) image, row, column (set by launch_cfg)
};

__device__ __forceinline__ int xcd_remap(int b, int nwg) {
  // Blocks are dispatched round-robin over the 8 XCDs (b % 8).  Give every XCD a
  // contiguous run of logical tiles so that tiles sharing A rows / B columns share an L2.
  int xcd = b & 7;
  int q = nwg >> 3, r = nwg & 7;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (b >> 3);
}

__device__ __forceinline__ f32x4 buf_load(__amdgpu_buffer_rsrc_t rsrc, unsigned voff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0));
}
// voffset (per lane, range-checked) + soffset (wave-uniform, NOT range-checked): the uniform part of an
// address costs a scalar add instead of a VALU add -- and VALU instructions are paid in MFMA time on the
// fp32 matrix pipe (tools/mb/mfma_fill.hip: ~5 cycles each beside a 64-cycle MFMA, never hidden).
__device__ __forceinline__ f32x4 buf_load_s(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0));
}

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>)
template <int I, int N, class F>
__device__ __forceinline__ void static_for_impl(F& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for_impl<I + 1, N>(f);
  }
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl<0, N>(f);
}

template <int BK>
__device__ __forceinline__ int swz(int r) {
  return BK == 32 ? ((r >> 1) & 7) : ((r >> 2) & 3);
}

// MODE_CONV: tf.layers.conv2d; MODE_DECONV: conv2d_transpose (stride 2);
// MODE_WINO: the 16 GEMMs of Winograd F(2x2,3x3) for 3x3 stride-1 SAME convs -- group
// g = (xi,nu) = tile % 16, A rows are 2x2 output tiles whose B^T d B input transform is
// evaluated on the fly in the loader (4 signed source pixels per element), B = the
// pre-transformed weights U_g = (G g G^T)[xi][nu], output = M_g [tiles][Cout] workspace.
// MODE_CVOL: OFlowNet conv0 (3x3 on the 8x8 window grid) with the local cost volume
// V[p,i,j,:] = f2[p] - f1[p + (i-4, j-4)] (KFNet/KFNet.py:343-359) generated in the loader:
// the 39 MB/frame volume is never materialised, f1/f2 (614 KB each) stay L2-resident.
constexpr int MODE_CONV = 0, MODE_DECONV = 1, MODE_WINO = 2, MODE_CVOL = 3;

// F16 (conv / transposed conv only): operands are rounded to fp16 while they are staged
// (activations stay fp32 in HBM, weights are pre-packed fp16), products accumulate in fp32 on
// v_mfma_f32_32x32x16_f16 -- 16x the fp32 MFMA rate, for BASELINE config 5 ("fp16 convs").
// An LDS row is still 64 bytes = four 16-byte quads, now holding 32 halfs: BK (the fp32
// k-step, must be 16) covers KCH = 32 channels per stage.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// PREC 2 ("f16x3"): every fp32 operand x is split while staged into hi = f16(x) and
// lo = f16(x - hi) (22 significand bits together); the product is evaluated as
// hi*hi + hi*lo + lo*hi on the fp16 MFMA with fp32 accumulation -- three MFMAs at 16x the
// fp32 rate instead of eight fp32 MFMAs per 16-wide k-step.  Weights are pre-split (and
// pre-scaled by 2^10 so that their lo parts are fp16 normals; the epilogue multiplies the
// exact power of two back).  Dropped lo*lo term and fp16 rounding of lo: ~2^-21 relative
// per product, i.e. fp32-class accuracy.
constexpr int PREC_F32 = 0, PREC_F16 = 1, PREC_F16X3 = 2;
// PREC_F32_N16: fp32 with 16-column sub-tiles on v_mfma_f32_16x16x4_f32 (same FLOP rate as
// 32x32x2).  For the 16-channel layers of OFlowNet (conv6, upconv0) a 32-wide tile wastes
// half of every MFMA; here a 32-row block is two 16-row MFMAs per 4 k, the two k-chunks of
// a stage become the two row halves, and everything else (staging, schedule) is unchanged.
constexpr int PREC_F32_N16 = 3;
// PREC_F16 with fp16 ACTIVATIONS in memory (BASELINE config 5 end to end, forward convs only):
//   _X  the input tensor holds halfs: one 16-byte load per LDS quad (8 channels), no conversion while staging;
//   _Y  the output tensor holds halfs: the accumulators (+ bias, ReLU) are rounded once (RNE), the tile is
//       transposed through the idle operand LDS and leaves as 16-byte runs of 8 channels;
//   _XY both.  k-step 16 or 32 (= 32 / 64 channels per stage).
constexpr int PREC_F16_X = 4, PREC_F16_Y = 5, PREC_F16_XY = 6;
// PREC_F16_XY_BDMA: as _XY, with the WEIGHT tile going global -> LDS directly (`buffer_load_dwordx4 ... lds`): no
// staging registers and no ds_write_b128 for two thirds of the operand bytes (timing builds without LDS stores run
// 25-30 % faster: tools/mb/build_hot.sh).  A lane's 16 bytes land at wave base + lane*16, so the XOR swizzle of the
// LDS rows moves to the SOURCE address (lane (row, slot q) fetches logical quad q ^ sw(row)); three B buffers: the
// transfer for stage s+2 is issued in stage s, behind the LDS stores of A -- loads return in order, so the wait the
// compiler places in front of the next stage's A stores also covers it, one stage before its barrier.
constexpr int PREC_F16_XY_BDMA = 7;
// PREC_F16_XY_DMA: BOTH operand tiles global -> LDS directly, three buffers each: no staging registers, no ds_write at
// all.  Nothing in registers orders the transfers any more, so the schedule is explicit: the transfers for stage s+2
// are issued right AFTER the barrier of stage s (their buffers were last read before the barrier of stage s-1), every
// wave waits `vmcnt(0)` right BEFORE the barrier of stage s+1 -- a full stage later -- and the first fragment reads of
// stage s+2 follow that barrier.  Activation rows outside the image carry an out-of-range offset: the transfer writes
// zeros for them (checked on hardware, tools/mb/dma_probe.hip), the same free SAME padding as on the register path.
constexpr int PREC_F16_XY_DMA = 8;

template <int TM, int TN, int WM, int WN, int BK, int MODE, int PREC = PREC_F32>
__global__ __launch_bounds__(64 * WM * WN, (TM * TN > 8 ? 1 : 2)) void conv_mfma_kernel(ConvArgs p) {
  // (a 128x128 wave tile = 16 accumulators = 256 registers: one wave per SIMD, the full 512-register budget)
  constexpr bool ADMA = (PREC == PREC_F16_XY_DMA);                      // activations through LDS-DMA too
  constexpr bool BDMA = (PREC == PREC_F16_XY_BDMA) || ADMA;
  constexpr bool X16 = (PREC == PREC_F16_X || PREC == PREC_F16_XY || BDMA);   // activations read as halfs
  constexpr bool Y16 = (PREC == PREC_F16_Y || PREC == PREC_F16_XY || BDMA);   // activations written as halfs
  static_assert(!BDMA || BK == 16, "LDS-DMA weights: k-step 16");
  constexpr bool F16 = (PREC == PREC_F16 || PREC == PREC_F16X3 || X16 || Y16);   // operands live in LDS as halfs
  constexpr unsigned XB = X16 ? 2u : 4u;                               // bytes per input element
  // fp16-activation kernels walk the K dimension TAP-INNERMOST: all live taps of one channel chunk, then the next
  // chunk.  A tile's rows re-read the same ~(BM + halo) pixels for every tap; with the taps innermost that
  // working set is (pixels x KCH channels) -- a few KB, L1/L2 resident -- instead of (pixels x Cin), which at
  // Cin >= 512 falls out of the 4 MiB L2 of an XCD between two taps (64 resident workgroups x 180 KB).
  // Their weights are packed CHUNK-MAJOR for it, [K/32][cout_pad][32 halfs] (graph.pack_conv_kernel_chunked): the
  // B tile of a stage is one (k-step 32: two) contiguous run of BN x 64 bytes, every fetched line fully used.
  // The fp32 kernels keep the tap-outermost order (and with it their summation order) and the [Cout][K] layout.
  constexpr bool TAP_INNER = X16 || Y16;
  constexpr bool N16 = (PREC == PREC_F32_N16);
  constexpr int CB = N16 ? 16 : 32;          // columns per MFMA sub-tile
  static_assert(!N16 || BK == 16, "the 16-column variant maps the two k-chunks of a 16-deep stage to row halves");
  constexpr bool X3 = (PREC == PREC_F16X3);  // ... as separate hi and lo tiles
  constexpr int NPART = X3 ? 2 : 1;
  constexpr bool TRANSPOSED = (MODE == MODE_DECONV);
  constexpr bool WINO = (MODE == MODE_WINO);
  constexpr bool CVOL = (MODE == MODE_CVOL);
  static_assert(!F16 || ((BK == 16 || ((X16 || Y16) && BK == 32)) && (MODE == MODE_CONV || MODE == MODE_DECONV)),
                "F16: conv/deconv at BK=16 (fp16 activations: also 32)");
  static_assert(!(X16 || Y16) || MODE == MODE_CONV, "fp16 activations: forward convolutions only");
  constexpr int NSRC = WINO ? 4 : ((CVOL || (F16 && !X16)) ? 2 : 1);  // global loads per A quad
  constexpr int QCH = F16 ? 8 : 4;                           // channels per 16-byte LDS quad
  constexpr int KCH = F16 ? 2 * BK : BK;                     // channels per stage
  constexpr int BM = 32 * TM * WM;
  constexpr int BN = CB * TN * WN;
  constexpr int NT = 64 * WM * WN;
  constexpr int QPR = BK / 4;       // float4 quads per tile row
  constexpr int RPP = NT / QPR;     // tile rows covered per pass of the whole block
  constexpr int AP = (BM + RPP - 1) / RPP;
  constexpr int BP = (BN + RPP - 1) / RPP;
  constexpr int A_PART = BM * BK, B_PART = BN * BK;   // one (hi or lo) tile, in floats
  constexpr int A_ELEMS = A_PART * NPART;
  constexpr int B_ELEMS = B_PART * NPART;
  constexpr int NCH = BK / 8;       // 8-wide k-chunks per stage

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                 // [2][BM][BK] swizzled
  float* Bs = smem + (ADMA ? 3 : 2) * A_ELEMS;   // [2][BN][BK] swizzled ([3] with LDS-DMA; then A has three buffers as well)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN;
  const int wn = wave % WN;

  const int nwg = p.tiles_m * p.tiles_n * (WINO ? 16 : 1);
  int tile = xcd_remap(blockIdx.x, nwg);
  int grp = 0;
  int tn, tm;
  if (WINO) {
    // Order inside an XCD's run: blocks of 8 M-tiles; within a block the (group, n-tile)
    // combination is the slow index and the M-tile the fast one.  The ~64 workgroups resident
    // on an XCD are then 8 M-tiles x 8 combinations: 8 workgroups share each U slice and 8
    // share each M-tile's source pixels through the L2 (measured: all-combinations-of-one-
    // M-tile-together 19.7-19.8 ms of GEMM phase per batch, blocks of 4 / 8 / 16: 19.53 /
    // 19.48 / 19.96; one group at a time over all M-tiles: 3-15 % slower).
    constexpr int MBk = 8;
    const int NC = 16 * p.tiles_n;
    const int blk = tile / (MBk * NC);
    const int first = blk * MBk;
    const int size = (p.tiles_m - first < MBk) ? p.tiles_m - first : MBk;
    const int r = tile - blk * MBk * NC;
    const int combo = r / size;
    tm = first + (r - combo * size);
    grp = combo & 15;
    tn = combo >> 4;
  } else {
    tn = tile % p.tiles_n;
    tm = tile / p.tiles_n;
  }
  const int m0 = tm * BM;
  const int n0 = tn * BN;

  const int q = tid % QPR;
  const int r0 = tid / QPR;

  // The A descriptor is re-based at the first image this tile touches, so 32-bit byte
  // offsets only have to span the few images of ONE tile (activations may exceed 2 GiB).
  const int HoWo = p.Ho * p.Wo;
  const int n_first = TRANSPOSED ? 0 : fdiv(CVOL ? (m0 >> 6) : m0, p.fd_img);
  const unsigned long long a_base = (unsigned long long)n_first * p.H * p.W * p.ldx * (unsigned long long)XB;
  const unsigned long long a_rest = p.x_bytes - a_base;
  // The range check of a buffer load looks at the per-lane offset only (the scalar offset is excluded), and
  // the per-row offset of the input pixel (iy0, ix0) may lie up to (pad_t rows + pad_l pixels) BEFORE the
  // image (5 rows + 5 pixels for the cost-volume window): the descriptor is based that far below the tile's
  // first image and every row offset shifted up by the same amount, so that row offsets are non-negative
  // and the tap / channel offset can ride in the scalar operand.
  const unsigned a_shift = (TRANSPOSED || WINO) ? 0u
                           : (unsigned)(((CVOL ? 5 : p.pad_t) * p.W + (CVOL ? 5 : p.pad_l)) * p.ldx) * XB;
  char* const a_ptr = const_cast<char*>(reinterpret_cast<const char*>(p.x)) + a_base;
  const unsigned long long a_span = a_rest + a_shift;
  const int a_records = (int)(a_span < 0x7fffffffull ? a_span : 0x7fffffffull);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(a_ptr - a_shift, 0, a_records, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsA2 = __builtin_amdgcn_make_buffer_rsrc(  // CVOL: f2 (same shape as f1)
      const_cast<char*>(reinterpret_cast<const char*>(CVOL ? p.x2 : p.x)) + a_base, 0,
      (int)(a_rest < 0x7fffffffull ? a_rest : 0x7fffffffull), 0x00020000);
  float* const b_ptr = const_cast<float*>(p.w) + (size_t)grp * p.cout_pad * p.Ktot;
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(b_ptr, 0, p.w_bytes, 0x00020000);

  // ---- per-thread im2col row state -------------------------------------------------
  // conv:        a_off = byte offset of input pixel (iy0, ix0) (may be "negative" = wrapped;
  //              adding the tap offset brings valid taps back in range), a_msk = valid taps
  // transposed:  a_off = n_img*H*W, coordinates kept in a_y/a_x, offset computed per tap
  unsigned a_off[AP];
  unsigned a_msk[AP];
  int a_y[AP], a_x[AP];
  unsigned a_w4[WINO ? AP : 1][4];  // WINO: byte offsets of the 4 signed source pixels (or OOB)
  unsigned a_off2[CVOL ? AP : 1];   // CVOL: byte offset of f2[p]
  unsigned a_msk2[CVOL ? AP : 1];   // CVOL: taps whose shifted f1 pixel is inside the image
  // B^T rows of F(2x2,3x3): V[xi] = sa*d[ra] + sb*d[rb]
  const int w_xi = grp >> 2, w_nu = grp & 3;
  const int w_ra = (w_xi == 0) ? 0 : 1, w_rb = (w_xi == 3) ? 3 : 2;
  const int w_ca = (w_nu == 0) ? 0 : 1, w_cb = (w_nu == 3) ? 3 : 2;
  const float w_sra = (w_xi == 2) ? -1.f : 1.f, w_srb = (w_xi == 0 || w_xi == 3) ? -1.f : 1.f;
  const float w_sca = (w_nu == 2) ? -1.f : 1.f, w_scb = (w_nu == 0 || w_nu == 3) ? -1.f : 1.f;
  const float w_s00 = w_sra * w_sca, w_s01 = w_sra * w_scb, w_s10 = w_srb * w_sca, w_s11 = w_srb * w_scb;
  typedef float f32x2s __attribute__((ext_vector_type(2)));
  const f32x2s w_p00 = {w_s00, w_s00}, w_p01 = {w_s01, w_s01}, w_p10 = {w_s10, w_s10}, w_p11 = {w_s11, w_s11};
#pragma unroll
  for (int i = 0; i < AP; ++i) {
    const int r = r0 + i * RPP;
    const int m = m0 + r;
    a_off[i] = 0;
    a_msk[i] = 0;
    a_y[i] = a_x[i] = 0;
    if (r < BM && m < p.M) {
      int n_img, oy, ox;
      if (CVOL) {
        // row = (pixel pp, window cell (wi, wj)); conv0 slides over the 8x8 window grid
        const int pp = m >> 6, pos = m & 63;
        const int HW = p.H * p.W;
        const int n_abs = fdiv(pp, p.fd_img);
        const int rem = pp - n_abs * HW;
        n_img = n_abs - n_first;
        oy = fdiv(rem, p.fd_row);  // pixel coordinates of pp in the feature map
        ox = rem - oy * p.W;
        const int wi = pos >> 3, wj = pos & 7;
        unsigned m1 = 0, m2 = 0;
        for (int t = 0; t < 9; ++t) {
          const int ky = t / 3, kx = t - ky * 3;
          const int ci = wi + ky - 1, cj = wj + kx - 1;          // window cell read by this tap
          if ((unsigned)ci < 8u && (unsigned)cj < 8u) {          // else conv0's SAME zero padding
            m1 |= 1u << t;
            const int sy = oy + ci - 4, sx = ox + cj - 4;        // translate(): f1 shifted, 0 outside
            if ((unsigned)sy < (unsigned)p.H && (unsigned)sx < (unsigned)p.W) m2 |= 1u << t;
          }
        }
        a_off[i] = (unsigned)(((n_img * p.H + oy + wi - 5) * p.W + (ox + wj - 5)) * p.ldx + q * 4) * 4u + a_shift;
        a_off2[CVOL ? i : 0] = (unsigned)((n_img * HW + rem) * p.ldx + q * 4) * 4u;
        a_msk[i] = m1;
        a_msk2[CVOL ? i : 0] = m2;
        continue;
      }
      if (TRANSPOSED) {
        // parity-class-major row order: m = cls*(N*H*W) + (n, i, j); output pixel
        // (2i + (cls>>1), 2j + (cls&1)).  All rows of a tile (bar 3 seams) then share a
        // parity class and hence the same 1/2/2/4 live taps -- the rest are skipped.
        const int NHW = p.N * p.H * p.W;
        const int cls = fdiv(m, p.fd_cls);
        const int idx = m - cls * NHW;
        n_img = fdiv(idx, p.fd_img);
        const int rem = idx - n_img * (p.H * p.W);
        const int ii = fdiv(rem, p.fd_row);
        oy = 2 * ii + (cls >> 1);
        ox = 2 * (rem - ii * p.W) + (cls & 1);
      } else {
        const int n_abs = fdiv(m, p.fd_img);
        const int rem = m - n_abs * HoWo;
        n_img = n_abs - n_first;
        oy = fdiv(rem, p.fd_row);
        ox = rem - oy * p.Wo;
      }
      unsigned msk = 0;
      if (WINO) {
        // (oy, ox) is the 2x2 output tile; its 4x4 input patch starts at (2*oy-1, 2*ox-1)
        const int ys[2] = {2 * oy - 1 + w_ra, 2 * oy - 1 + w_rb};
        const int xs[2] = {2 * ox - 1 + w_ca, 2 * ox - 1 + w_cb};
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const bool ok = ((unsigned)ys[a] < (unsigned)p.H) && ((unsigned)xs[b] < (unsigned)p.W);
            a_w4[WINO ? i : 0][a * 2 + b] =
                ok ? (unsigned)(((n_img * p.H + ys[a]) * p.W + xs[b]) * p.ldx + q * 4) * 4u : OOB;
          }
        msk = 1u;
      } else if (TRANSPOSED) {
        const int by = oy + p.pad_t, bx = ox + p.pad_l;
        a_y[i] = by;
        a_x[i] = bx;
        a_off[i] = (unsigned)(n_img * p.H * p.W);
        // tap (ky,kx) is live iff by-ky and bx-kx are even, >= 0 and inside the input
        unsigned colmask = 0;
        for (int kx = 0; kx < p.kw; ++kx) {
          const int tx = bx - kx;
          if (tx >= 0 && !(tx & 1) && (tx >> 1) < p.W) colmask |= 1u << kx;
        }
        for (int ky = 0; ky < p.kh; ++ky) {
          const int ty = by - ky;
          if (ty >= 0 && !(ty & 1) && (ty >> 1) < p.H) msk |= colmask << (ky * p.kw);
        }
      } else {
        const int iy0 = oy * p.stride - p.pad_t, ix0 = ox * p.stride - p.pad_l;
        // (LDS-DMA: slot q of row r receives logical quad q ^ sw(r) -- the swizzle is applied at the source)
        a_off[i] = (unsigned)(((n_img * p.H + iy0) * p.W + ix0) * p.ldx + (ADMA ? ((q ^ swz<BK>(r)) & (QPR - 1)) : q) * QCH) * XB + a_shift;
        // valid taps = [ky_lo, ky_hi) x [kx_lo, kx_hi)
        const int ky_lo = iy0 < 0 ? -iy0 : 0, ky_hi = (p.H - iy0 < p.kh) ? p.H - iy0 : p.kh;
        const int kx_lo = ix0 < 0 ? -ix0 : 0, kx_hi = (p.W - ix0 < p.kw) ? p.W - ix0 : p.kw;
        const unsigned colmask = (kx_hi > kx_lo) ? (((1u << (kx_hi - kx_lo)) - 1u) << kx_lo) : 0u;
        for (int ky = ky_lo; ky < ky_hi; ++ky) msk |= colmask << (ky * p.kw);
      }
      a_msk[i] = msk;
    }
  }
  unsigned b_off[BP];
#pragma unroll
  for (int i = 0; i < BP; ++i) {
    const int r = r0 + i * RPP;
    const int n = n0 + r;
    if constexpr (BDMA)        // LDS slot q of row r receives logical quad q ^ sw(r) (the swizzle is applied at the source)
      b_off[i] = (r < BN && n < p.cout_pad) ? (unsigned)(n * 64 + ((q ^ swz<BK>(r)) & 3) * 16) : OOB;
    else if constexpr (TAP_INNER)   // chunk-major weights: 64-byte rows of a [cout_pad][32] block, the k-step-32 quads 4..7 in the next block
      b_off[i] = (r < BN && n < p.cout_pad) ? (unsigned)(((q >> 2) * p.cout_pad + n) * 64 + (q & 3) * 16) : OOB;
    else
      b_off[i] = (r < BN && n < p.cout_pad) ? (unsigned)(n * p.Ktot + q * QCH) * (F16 ? 2u : 4u) : OOB;
  }

  // ---- which taps touch at least one in-range input pixel of this tile? -----------
  // (zero-padding taps of whole tiles are skipped: exact, they only add +0.)
  unsigned tapmask = 1u;   // Winograd GEMMs: one "tap"
  if constexpr (!WINO) {
    // OR over the wave by shuffles, over the waves through a small static LDS array (one barrier)
    __shared__ unsigned red[NT / 64];
    unsigned mine = 0;
#pragma unroll
    for (int i = 0; i < AP; ++i) mine |= a_msk[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine |= (unsigned)__shfl_xor((int)mine, o);
    if (lane == 0) red[wave] = mine;
    __syncthreads();
    tapmask = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) tapmask |= red[w];
  }

  f32x16 acc[TM][TN];
  f32x4 acc4[N16 ? TM : 1][N16 ? TN : 1][2];   // N16: two 16-row halves per 32-row block
#pragma unroll
  for (int mi = 0; mi < TM; ++mi)
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;
      if constexpr (N16) {
        acc4[mi][ni][0] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc4[mi][ni][1] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }

  f32x4 ga[AP * NSRC], gb[BP * NPART];

  // ---- stage iterator of the LOAD stream (compute only counts stages) ---------------
  const int kchunks = p.Cin / KCH;
  const int n_stages = __builtin_popcount(tapmask) * kchunks;
  int ld_tap = tapmask ? __builtin_ctz(tapmask) : 32;
  // Concurrent workgroups walk the Cin chunks in ROTATED order: with NHWC the pixel stride
  // is Cin*4 bytes (4 KiB at Cin = 1024), so workgroups in lockstep would all read the same
  // byte range modulo the pixel stride and pile onto the same L2 channels/sets.
  const int rot = ((p.rot_mode & 3) == 0 || (p.rot_mode & 3) == 3) ? 0 : ((tm * 7 + tn * 3 + ((p.rot_mode & 3) == 2 ? grp * 5 : 0)) % kchunks);
  int ld_ci = 0;
  int ld_c0 = rot * KCH;
  // Is the load stream's current tap inside the image for row slot i?  Evaluated when the tap CHANGES (once
  // per Cin/KCH stages) and carried as a lane mask, so that a load costs one v_cndmask instead of a
  // shift + and + compare + select per stage.
  bool a_ok[AP], a_ok2[CVOL ? AP : 1];
  auto refresh_ok = [&]() {
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      a_ok[i] = (ld_tap < 32) && ((a_msk[i] >> (ld_tap & 31)) & 1u);
      if (CVOL) a_ok2[CVOL ? i : 0] = (ld_tap < 32) && ((a_msk2[CVOL ? i : 0] >> (ld_tap & 31)) & 1u);
    }
  };
  refresh_ok();
  const int first_tap = ld_tap;
  auto advance = [&]() {
    if constexpr (TAP_INNER) {
      const unsigned rest = (ld_tap < 31) ? (tapmask & ~((2u << ld_tap) - 1u)) : 0u;
      if (rest) {
        ld_tap = __builtin_ctz(rest);
      } else {
        ++ld_ci;
        ld_c0 += KCH;
        ld_tap = (ld_ci >= kchunks) ? 32 : first_tap;
      }
      refresh_ok();
      return;
    }
    ++ld_ci;
    ld_c0 += KCH;
    if (ld_c0 >= p.Cin) ld_c0 = 0;
    if (ld_ci >= kchunks) {
      ld_ci = 0;
      ld_c0 = rot * KCH;
      const unsigned rest = (ld_tap < 31) ? (tapmask & ~((2u << ld_tap) - 1u)) : 0u;
      ld_tap = rest ? __builtin_ctz(rest) : 32;
      if (!WINO && !TRANSPOSED) refresh_ok();
    }
  };

  // Issue the global loads of the stage the iterator points at (all-OOB = zeros once the
  // iterator has run off the end: keeps the loop body branch-free).
  auto load_one = [&](int k, bool live, unsigned adelta, unsigned bdelta, int ky, int kx, int tap) {
#if defined(KFN_CONV_HOT) && (KFN_CONV_HOT & 8)   // timing experiment only: no global loads in the main loop
    if (TAP_INNER && ld_ci > 0) return;
#endif
    if (WINO && k < AP * NSRC) {
      // per-lane: the source pixel's offset (or OOB for zero padding); uniform: channel offset; a stream
      // that has run off the end reads through a descriptor with num_records = 0 (scalar select)
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a_ptr, 0, live ? a_records : 0, 0x00020000);
      ga[k] = buf_load_s(rs, a_w4[WINO ? k / NSRC : 0][k % NSRC], (unsigned)ld_c0 * 4u);
    } else if (CVOL && k < AP * NSRC) {
      const int i = k / NSRC;
      if (k % NSRC == 0) {  // f2[p]: same for every tap that lies inside the window
        ga[k] = buf_load_s(rsA2, a_ok[i] ? a_off2[CVOL ? i : 0] : OOB, (unsigned)ld_c0 * 4u);
      } else {              // shifted f1
        ga[k] = buf_load_s(rsA, a_ok2[CVOL ? i : 0] ? a_off[i] : OOB, adelta);
      }
    } else if (X16 && k < AP * NSRC) {   // fp16 activations: the quad's 8 channels are one 16-byte load
#if defined(KFN_CONV_HOT) && (KFN_CONV_HOT & 1)   // timing experiment only (tools/mb/build_hot.sh): every A load hits a 64 KiB window
      ga[k] = buf_load_s(rsA, a_shift + (a_off[k] & 0xFFF0u), 0u);   // inside the tensor: the descriptor is based a_shift below it
#else
      ga[k] = buf_load_s(rsA, a_ok[k] ? a_off[k] : OOB, adelta);
#endif
    } else if (F16 && k < AP * NSRC) {   // two consecutive float4 = the 8 channels of one fp16 quad
      const int i = k / NSRC;
      if (TRANSPOSED) {
        const int ty = a_y[i] - ky, tx = a_x[i] - kx;
        const unsigned pix = a_off[i] + (unsigned)((ty >> 1) * p.W + (tx >> 1));
        unsigned vo = (pix * (unsigned)p.ldx + (unsigned)(ld_c0 + q * QCH)) * 4u;
        vo = (live && ((a_msk[i] >> tap) & 1u)) ? vo + (unsigned)(k % NSRC) * 16u : OOB;
        ga[k] = buf_load(rsA, vo);
      } else {
        ga[k] = buf_load_s(rsA, a_ok[i] ? a_off[i] : OOB, adelta + (unsigned)(k % NSRC) * 16u);
      }
    } else if (!F16 && k < AP) {
      const int i = k;
      if (TRANSPOSED) {
        const int ty = a_y[i] - ky, tx = a_x[i] - kx;
        const unsigned pix = a_off[i] + (unsigned)((ty >> 1) * p.W + (tx >> 1));
        unsigned vo = (pix * (unsigned)p.ldx + (unsigned)(ld_c0 + q * QCH)) * 4u;
        vo = (live && ((a_msk[i] >> tap) & 1u)) ? vo : OOB;
        ga[i] = buf_load(rsA, vo);
      } else {
        ga[i] = buf_load_s(rsA, a_ok[i] ? a_off[i] : OOB, adelta);
      }
    } else {
      const int kk = k - AP * NSRC;
      const int i = kk / NPART;
      // f16x3: the packed weights are [hi | lo], lo starts w_lo_bytes after hi
      const unsigned part_off = (X3 && (kk % NPART)) ? p.w_lo_bytes : 0u;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(b_ptr, 0, live ? (int)p.w_bytes : 0, 0x00020000);
#if defined(KFN_CONV_HOT) && (KFN_CONV_HOT & 2)   // timing experiment only: every B load re-reads the first K chunk
      gb[kk] = buf_load_s(rs, b_off[i], part_off);
#else
      gb[kk] = buf_load_s(rs, b_off[i], (F16 ? bdelta >> 1 : bdelta) + part_off);
#endif
    }
  };

  // swizzled LDS store position of this thread's quad: sw(r) is the same for all passes
  // because RPP is a multiple of 16 rows.
  const int wr_off = r0 * BK + ((q ^ swz<BK>(r0)) * 4);
  auto store_one = [&](int k, int buf) {
#if defined(KFN_CONV_HOT) && (KFN_CONV_HOT & 4)   // timing experiment only: no LDS stores in the main loop (stale operands)
    if (TAP_INNER && n_stages > 2 && buf >= 0) return;
#endif
    if (k < AP) {
      const int i = k;
      f32x4 v;
      if (WINO) {  // B^T d B for this (xi,nu): four signed source pixels, as packed fp32 math
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x4 g0 = ga[i * NSRC + 0], g1 = ga[i * NSRC + (WINO ? 1 : 0)], g2 = ga[i * NSRC + (WINO ? 2 : 0)],
                    g3 = ga[i * NSRC + (WINO ? 3 : 0)];
        // (the compiler scalarises <2 x float> fma; the signs are wave-uniform register pairs)
        f32x2 lo, hi;
        asm("v_pk_mul_f32 %0, %1, %2" : "=v"(lo) : "s"(w_p00), "v"(g0.xy));
        asm("v_pk_mul_f32 %0, %1, %2" : "=v"(hi) : "s"(w_p00), "v"(g0.zw));
        asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(lo) : "s"(w_p01), "v"(g1.xy));
        asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(hi) : "s"(w_p01), "v"(g1.zw));
        asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(lo) : "s"(w_p10), "v"(g2.xy));
        asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(hi) : "s"(w_p10), "v"(g2.zw));
        asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(lo) : "s"(w_p11), "v"(g3.xy));
        asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(hi) : "s"(w_p11), "v"(g3.zw));
        v = f32x4{lo.x, lo.y, hi.x, hi.y};
      } else if (CVOL) {
        v = ga[i * NSRC] - ga[i * NSRC + (CVOL ? 1 : 0)];   // diff_feat = feat_map2 - shift(feat_map1)
      } else if (X16) {
        v = ga[i];
      } else if (F16) {
        const f32x4 c0 = ga[i * NSRC], c1 = ga[i * NSRC + (F16 ? 1 : 0)];   // channels 0-3, 4-7
        const f16x8 h = {(_Float16)c0.x, (_Float16)c0.y, (_Float16)c0.z, (_Float16)c0.w,
                         (_Float16)c1.x, (_Float16)c1.y, (_Float16)c1.z, (_Float16)c1.w};   // RNE
        v = __builtin_bit_cast(f32x4, h);
        if (X3 && (AP * RPP == BM || r0 + i * RPP < BM)) {
          const f16x8 l = {(_Float16)(c0.x - (float)h[0]), (_Float16)(c0.y - (float)h[1]),
                           (_Float16)(c0.z - (float)h[2]), (_Float16)(c0.w - (float)h[3]),
                           (_Float16)(c1.x - (float)h[4]), (_Float16)(c1.y - (float)h[5]),
                           (_Float16)(c1.z - (float)h[6]), (_Float16)(c1.w - (float)h[7])};
          *reinterpret_cast<f32x4*>(As + buf * A_ELEMS + A_PART + wr_off + i * RPP * BK) =
              __builtin_bit_cast(f32x4, l);
        }
      } else {
        v = ga[i];
      }
      if (AP * RPP == BM || r0 + i * RPP < BM)
        *reinterpret_cast<f32x4*>(As + buf * A_ELEMS + wr_off + i * RPP * BK) = v;
    } else {
      const int kk = k - AP;
      const int i = kk / NPART;
      if (BP * RPP == BN || r0 + i * RPP < BN)
        *reinterpret_cast<f32x4*>(Bs + buf * B_ELEMS + (kk % NPART) * B_PART + wr_off + i * RPP * BK) = gb[kk];
    }
  };

  // fragment read positions: row (lane&31) of each 32-row sub-tile, logical quad 2c+h
  const int li = lane & 31, lh = lane >> 5;
  int rdq[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) rdq[c] = (((2 * c + lh) ^ swz<BK>(li)) * 4);
  const int a_rd = (wm * TM * 32 + li) * BK;
  const int b_rd = (wn * TN * 32 + li) * BK;
  // N16: lane (i = lane & 15, kq = lane >> 4) reads one quad of row i of a 16-row half.
  // ds_read_b128 is serviced in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... ;
  // with the staging swizzle (r>>2)&3 the k-quad order 0,3,1,2 over kq makes every group
  // touch 16 distinct 16-byte slots (A and B use the same order, so the sum is unchanged).
  const int l16 = lane & 15, kq16 = (0x9C >> (2 * (lane >> 4))) & 3;

  // double-buffered fragments: [A hi (TM) | B hi (TN)] and, for f16x3, [A lo | B lo] behind them
  f32x4 fr[2][(TM + TN) * NPART];
  int bcur = 0;   // BDMA: B buffer of the current stage (s % 3)
  auto read_one = [&](int k, int slot, int buf, int c, int bbuf) {
    const int part = k / (TM + TN), kk = k % (TM + TN);
    if constexpr (N16) {
      // chunk c = row half c of every 32-row block; the B fragment is the same for both halves
      const int row = (kk < TM) ? (wm * TM + kk) * 32 + 16 * c + l16 : (wn * TN + (kk - TM)) * 16 + l16;
      const float* base = (kk < TM) ? As + buf * A_ELEMS : Bs + buf * B_ELEMS;
      (void)bbuf;
      fr[slot][k] = *reinterpret_cast<const f32x4*>(base + row * BK + ((kq16 ^ swz<BK>(row)) * 4));
    } else if (kk < TM)
      fr[slot][k] = *reinterpret_cast<const f32x4*>(As + (ADMA ? bbuf : buf) * A_ELEMS + part * A_PART + a_rd + kk * 32 * BK + rdq[c]);
    else
      fr[slot][k] = *reinterpret_cast<const f32x4*>(Bs + (BDMA ? bbuf : buf) * B_ELEMS + part * B_PART + b_rd + (kk - TM) * 32 * BK + rdq[c]);
  };

  constexpr int NLD = (ADMA ? 0 : AP * NSRC) + (BDMA ? 0 : BP * NPART);  // global loads per stage (into registers)
  constexpr int NST = (ADMA ? 0 : AP) + (BDMA ? 0 : BP * NPART);         // LDS store ops per stage (an f16x3 A op writes hi and lo)
  constexpr int NDMA = BDMA ? BP : 0;                       // weight-tile transfers global -> LDS per stage
  constexpr int NADMA = ADMA ? AP : 0;                      // activation-tile transfers
  // BDMA: transfer i of this wave covers rows 64 i + 16 wave .. +15 of the B tile, lane L -> byte L*16 of that 1 KiB run
  typedef __attribute__((address_space(3))) void* lds_ptr_t [[maybe_unused]];   // (used in the device pass only)
  const unsigned lds_b0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)Bs + (unsigned)(wave * 16 * BK * 4));
  auto dma_one = [&](int i, bool live, unsigned bdelta, int bbuf) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(b_ptr, 0, live ? (int)p.w_bytes : 0, 0x00020000);
    const unsigned dst = lds_b0 + (unsigned)(bbuf * B_ELEMS * 4) + (unsigned)(i * RPP * BK * 4);
#if defined(__HIP_DEVICE_COMPILE__)   // (the host pass cannot type-check this target builtin: it would silently drop every kernel stub)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(uintptr_t)dst, 16, b_off[i], bdelta >> 1, 0, 0);
#else
    (void)rs; (void)dst;
#endif
  };
  const unsigned lds_a0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)As + (unsigned)(wave * 16 * BK * 4));
  auto dma_a_one = [&](int i, unsigned adelta, int abuf) {
    const unsigned dst = lds_a0 + (unsigned)(abuf * A_ELEMS * 4) + (unsigned)(i * RPP * BK * 4);
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(uintptr_t)dst, 16, a_ok[i] ? a_off[i] : OOB, adelta, 0, 0);
#else
    (void)dst; (void)adelta;
#endif
  };
  constexpr int NFR = (TM + TN) * NPART;       // fragment reads per chunk
  constexpr int NTERM = X3 ? 3 : 1;
  constexpr int J = (F16 ? NTERM : 4) * TM * TN;   // MFMAs per chunk

  if (n_stages > 0) {
    // ---- prologue: stage 0 -> LDS buffer 0, stage 1 -> registers ----------------------
    {
      const int ky = (int)(((unsigned)ld_tap * p.kw_inv) >> 16), kx = ld_tap - ky * p.kw;
      const unsigned adelta = (unsigned)((ky * p.W + kx) * p.ldx + ld_c0) * XB;
      const unsigned bdelta = TAP_INNER ? (unsigned)((ld_tap * (p.Cin >> 5) + (ld_c0 >> 5)) * p.cout_pad) * 128u
                                        : (unsigned)(ld_tap * p.Cin + ld_c0) * 4u;
#pragma unroll
      for (int k = 0; k < NLD; ++k) load_one(k, true, adelta, bdelta, ky, kx, ld_tap);
#pragma unroll
      for (int i = 0; i < NDMA; ++i) dma_one(i, true, bdelta, 0);
#pragma unroll
      for (int i = 0; i < NADMA; ++i) dma_a_one(i, adelta, 0);
      advance();
#pragma unroll
      for (int k = 0; k < NST; ++k) store_one(k, 0);
    }
    {
      const bool live = ld_tap < 32;
      const int tp = live ? ld_tap : 0;
      const int ky = (int)(((unsigned)tp * p.kw_inv) >> 16), kx = tp - ky * p.kw;
      const unsigned adelta = (unsigned)((ky * p.W + kx) * p.ldx + ld_c0) * XB;
      const unsigned bdelta = TAP_INNER ? (unsigned)((tp * (p.Cin >> 5) + (ld_c0 >> 5)) * p.cout_pad) * 128u
                                        : (unsigned)(tp * p.Cin + ld_c0) * 4u;
#pragma unroll
      for (int k = 0; k < NLD; ++k) load_one(k, live, adelta, bdelta, ky, kx, tp);
#pragma unroll
      for (int i = 0; i < NDMA; ++i) dma_one(i, live, bdelta, 1);
#pragma unroll
      for (int i = 0; i < NADMA; ++i) dma_a_one(i, adelta, 1);
      advance();
    }
    if constexpr (BDMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // both weight tiles have landed
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NFR; ++k) read_one(k, 0, 0, 0, 0);

    // ---- main loop: one iteration == one BK-deep stage ---------------------------------
    // Program order inside an iteration (everything but the MFMAs is slotted BETWEEN
    // MFMAs, which leave ~64 cycles of issue slack each):
    //   chunk 0       : ds_write regs(stage s+1) -> buf^1,  fragment reads of chunk 1
    //   chunk 1       : buffer loads of stage s+2 -> regs,  fragment reads of chunk 2
    //   chunk 2..N-2  : fragment reads of the next chunk
    //   chunk N-1     : first half of the MFMAs, lgkmcnt(0) + barrier, fragment reads of
    //                   chunk 0 of stage s+1 from buf^1, second half of the MFMAs
    // (BK = 16 has two chunks: stores in chunk 0, loads + barrier in chunk 1.)
    for (int s = 0; s < n_stages; ++s) {
      const int buf = s & 1;
      const bool live = ld_tap < 32;
      const int tp = live ? ld_tap : 0;
      const int ky = (int)(((unsigned)tp * p.kw_inv) >> 16), kx = tp - ky * p.kw;
      const unsigned adelta = (unsigned)((ky * p.W + kx) * p.ldx + ld_c0) * XB;
      const unsigned bdelta = TAP_INNER ? (unsigned)((tp * (p.Cin >> 5) + (ld_c0 >> 5)) * p.cout_pad) * 128u
                                        : (unsigned)(tp * p.Cin + ld_c0) * 4u;
      static_for<NCH>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        constexpr int slot = c & 1;
        constexpr bool last = (c == NCH - 1);
        // chunk that carries the global loads (issuing them in chunk 0 right behind the stores
        // was measured slightly slower)
        constexpr int LOADC = (NCH > 2) ? 1 : NCH - 1;
        // side ops of this chunk, in issue order
        constexpr int n_rd = last ? 0 : NFR;
        constexpr int n_st = (c == 0) ? NST : 0;
        constexpr int n_dma = (c == 0 && !ADMA) ? NDMA : 0;     // behind the A stores of the same chunk (see PREC_F16_XY_BDMA)
        constexpr int n_ld = (c == LOADC) ? NLD : 0;
        constexpr int n_side = n_rd + n_st + n_dma + n_ld;
        constexpr int JB = (TAP_INNER ? J / KFN_F16_BAR_DIV : J / 2) > 0 ? (TAP_INNER ? J / KFN_F16_BAR_DIV : J / 2) : 1;   // MFMAs of the last chunk in front of the barrier
        constexpr int jspan = last ? JB : J;  // in the last chunk side ops ride in front of the barrier
        static_for<J>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          constexpr int t = j / (TM * TN), rem = j % (TM * TN), mi = rem / TN, ni = rem % TN;
          if constexpr (last && j == (J > 1 ? JB : 0)) {
            if constexpr (ADMA) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // this wave's transfers for stage s+1 have landed
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            static_for<NFR>([&](auto kc) { read_one(decltype(kc)::value, slot ^ 1, buf ^ 1, 0, bcur == 2 ? 0 : bcur + 1); });
            if constexpr (ADMA) {
              // stage s+2 -> the buffers stage s-1 used (every read of them came before the PREVIOUS barrier)
              static_for<NDMA>([&](auto kc) { dma_one(decltype(kc)::value, live, bdelta, bcur == 0 ? 2 : bcur - 1); });
              static_for<NADMA>([&](auto kc) { dma_a_one(decltype(kc)::value, adelta, bcur == 0 ? 2 : bcur - 1); });
            }
          }
          if constexpr (F16) {
            // t = term: f16x3 adds the two cross terms first, the hi*hi term last
            constexpr int LO = TM + TN;
            constexpr int ia = (X3 && t == 0) ? LO + mi : mi;
            constexpr int ib = (X3 && t == 1) ? LO + TM + ni : TM + ni;
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fr[slot][ia]),
                                                                 __builtin_bit_cast(f16x8, fr[slot][ib]),
                                                                 acc[mi][ni], 0, 0, 0);
          } else if constexpr (N16)
            acc4[mi][ni][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(fr[slot][mi][t], fr[slot][TM + ni][t], acc4[mi][ni][c], 0, 0, 0);
          else
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fr[slot][mi][t], fr[slot][TM + ni][t], acc[mi][ni], 0, 0, 0);
          // side ops k with floor(k*jspan/n_side) == j ride behind MFMA j
          constexpr int kb = n_side ? (j * n_side + jspan - 1) / jspan : 0;
          constexpr int ke0 = n_side ? ((j + 1) * n_side + jspan - 1) / jspan : 0;
          constexpr int ke = ke0 < n_side ? ke0 : n_side;
          static_for<(ke > kb ? ke - kb : 0)>([&](auto kc) {
            constexpr int k = kb + decltype(kc)::value;
            if constexpr (k < n_rd) read_one(k, slot ^ 1, buf, c + 1, bcur);
            else if constexpr (k < n_rd + n_st) store_one(k - n_rd, buf ^ 1);
            else if constexpr (k < n_rd + n_st + n_dma) dma_one(k - n_rd - n_st, live, bdelta, bcur == 0 ? 2 : bcur - 1);
            else load_one(k - n_rd - n_st - n_dma, live, adelta, bdelta, ky, kx, tp);
          });
          if constexpr (KFN_F16_PIN != 0 && TAP_INNER) __builtin_amdgcn_sched_barrier(0);
        });
      });
      advance();
      if constexpr (BDMA) bcur = (bcur == 2) ? 0 : bcur + 1;
    }
  }

  // transposed: rows are in parity-class order, translate to output pixel indices via LDS
  int* out_pix = reinterpret_cast<int*>(smem);
  if (TRANSPOSED) {
    __syncthreads();  // everyone is done with the operand buffers
    for (int r = tid; r < BM; r += NT) {
      const int m = m0 + r;
      int op = -1;
      if (m < p.M) {
        const int NHW = p.N * p.H * p.W;
        const int cls = fdiv(m, p.fd_cls);
        const int idx = m - cls * NHW;
        const int n_img = fdiv(idx, p.fd_img);
        const int rem = idx - n_img * (p.H * p.W);
        const int ii = fdiv(rem, p.fd_row);
        op = (n_img * p.Ho + 2 * ii + (cls >> 1)) * p.Wo + 2 * (rem - ii * p.W) + (cls & 1);
      }
      out_pix[r] = op;
    }
    __syncthreads();
  }

  // ---- epilogue: bias, ReLU, fused head ops, store ------------------------------------
  // C/D layout of the 32x32 MFMA: col = lane & 31, row = (e & 3) + 8*(e >> 2) + 4*(lane >> 5).
  // The fused head epilogues exist only in the 32-column instantiations (the host forces one
  // of them); everything per element is branch-free: ReLU is a max + select, the store is
  // a buffer store (lane offset + wave-uniform row offset) whose range check drops the rows
  // past M of the last tile and the columns past Cout.  The output descriptor is re-based per tile (the
  // Winograd workspace [tile][16][Cout] exceeds 4 GiB).
  constexpr bool HEAD_EPI = (MODE == MODE_CONV) && TN == 1 && WN == 1 && !N16;
  const bool relu = p.relu != 0;
  const int rowmul = WINO ? 16 : 1;
  const unsigned row_bytes = (unsigned)(rowmul * p.ldy) * 4u;   // byte distance of consecutive GEMM rows
  const int rows_left = p.M - m0;
  const unsigned long long y_off = ((unsigned long long)m0 * rowmul + (WINO ? grp : 0)) * (unsigned long long)p.ldy * 4ull;
  const unsigned long long y_span = (unsigned long long)(rows_left < BM ? rows_left : BM) * row_bytes;
  const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<char*>(p.y) + (TRANSPOSED ? 0ull : y_off), 0,
      TRANSPOSED ? 0 : (int)(y_span - (WINO ? (unsigned long long)grp * p.ldy * 4ull : 0ull)), 0x00020000);
  auto epilogue = [&](auto epi_c) {
    constexpr int EPI = decltype(epi_c)::value;
    if constexpr (N16) {
      // C/D layout of the 16x16 MFMA: col = lane & 15, row = 4*(lane >> 4) + e
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) {
        const int n = n0 + (wn * TN + ni) * 16 + l16;
        const bool n_ok = n < p.Cout;
        const float bv = (p.bias != nullptr && n_ok) ? p.bias[n] : 0.f;
        const unsigned lane_off = n_ok ? (unsigned)(wm * TM * 32 + 4 * (lane >> 4)) * row_bytes + (unsigned)n * 4u : OOB;
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
          for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int rloc = mi * 32 + 16 * c + e;
              float v = acc4[mi][ni][c][e] + bv;
              v = relu ? fmaxf(v, 0.f) : v;
              if constexpr (TRANSPOSED) {
                const int op = out_pix[wm * TM * 32 + rloc + 4 * (lane >> 4)];
                if (n_ok && op >= 0) p.y[(size_t)op * p.ldy + n] = v;
              } else {
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsY,
                                                      lane_off + (unsigned)rloc * row_bytes, 0, 0);
              }
            }
      }
      return;
    }
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
      const int n = n0 + (wn * TN + ni) * 32 + li;
      const bool n_ok = n < p.Cout;
      const float bv = (p.bias != nullptr && n_ok) ? p.bias[n] : 0.f;
      const unsigned lane_off = n_ok ? (unsigned)(wm * TM * 32 + 4 * lh) * row_bytes + (unsigned)n * 4u : OOB;
#pragma unroll
      for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int rloc = mi * 32 + (e & 3) + 8 * (e >> 2);   // + wm*TM*32 + 4*lh = row within the tile
          float v = (X3 ? acc[mi][ni][e] * p.out_scale : acc[mi][ni][e]) + bv;
          v = relu ? fmaxf(v, 0.f) : v;
          if constexpr (EPI == KFN_EPI_L2NORM) {
            float ss = n_ok ? v * v : 0.f;
            ss += __shfl_xor(ss, 16);
            ss += __shfl_xor(ss, 8);
            ss += __shfl_xor(ss, 4);
            ss += __shfl_xor(ss, 2);
            ss += __shfl_xor(ss, 1);
            v = v / sqrtf(fmaxf(ss, 1e-12f));
          } else if constexpr (EPI == KFN_EPI_EXP_CH3) {
            if (n == 3) v = expf(v);
          } else if constexpr (EPI == KFN_EPI_EXP_1E2) {
            v = expf(v) * 1e-2f;
          }
          if constexpr (TRANSPOSED) {
            const int op = out_pix[wm * TM * 32 + rloc + 4 * lh];
            if (n_ok && op >= 0) p.y[(size_t)op * p.ldy + n] = v;
          } else {
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsY,
                                                  lane_off + (unsigned)rloc * row_bytes, 0, 0);
          }
        }
      }
    }
  };
  if constexpr (Y16) {
    // fp16 output: bias + ReLU in fp32, ONE rounding to half (RNE), then the BM x BN tile is transposed through
    // the operand LDS (idle now) so that it leaves as 16-byte runs of 8 channels -- 8 lanes cover 128 contiguous
    // bytes of a pixel -- instead of one 2-byte store per accumulator element (global stores are issue-bound).
    // Lanes (n, n+1) of a 32x32 block hold neighbouring channels of the same 16 rows: the even lane takes row
    // e of both, the odd lane row e+1 of both (one DPP quad_perm exchange per row pair), each packs its pair and
    // writes one dword; rows are BN halfs = BN/2 dwords apart, unpadded (the b128 read-back is conflict-free).
    constexpr int RD = BN / 2;           // dwords per tile row
    constexpr int CPR = BN / 8;          // 16-byte chunks per tile row
    static_assert((BM * CPR) % NT == 0, "tile chunks must divide evenly over the workgroup");
    unsigned* Ct = reinterpret_cast<unsigned*>(smem);
    // (LDS-DMA weights: the transfers issued for the two stages past the end write zeros into B buffers this tile
    //  is about to overlay -- they must have landed first)
    if constexpr (BDMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                     // every wave is done reading the operand buffers
    const bool odd = (li & 1) != 0;
    const bool relu16 = p.relu != 0;
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
      const int n = n0 + (wn * TN + ni) * 32 + li;
      const float bv = (p.bias != nullptr && n < p.Cout) ? p.bias[n] : 0.f;
      const int cdw = ((wn * TN + ni) * 32 + (li & ~1)) >> 1;
#pragma unroll
      for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
          float v0 = acc[mi][ni][e] + bv, v1 = acc[mi][ni][e + 1] + bv;
          v0 = relu16 ? fmaxf(v0, 0.f) : v0;
          v1 = relu16 ? fmaxf(v1, 0.f) : v1;
          const float send = odd ? v0 : v1;
          const float recv = __builtin_bit_cast(
              float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0xB1 /* quad_perm [1,0,3,2] */,
                                                 0xF, 0xF, false));
          typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
          const f16x2 pk = {(_Float16)(odd ? recv : v0), (_Float16)(odd ? v1 : recv)};
          const int row = (wm * TM + mi) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh + (odd ? 1 : 0);
          Ct[row * RD + cdw] = __builtin_bit_cast(unsigned, pk);
        }
      }
    }
    __syncthreads();
    const unsigned row_b16 = (unsigned)p.ldy * 2u;
    const int rows16 = p.M - m0;
    const unsigned long long span16 = (unsigned long long)(rows16 < BM ? rows16 : BM) * row_b16;
    const __amdgpu_buffer_rsrc_t rsY16 = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(p.y) + (unsigned long long)m0 * row_b16, 0, (int)span16, 0x00020000);
#pragma unroll
    for (int j = 0; j < (BM * CPR) / NT; ++j) {
      const int id = tid + j * NT;
      const int row = id / CPR, ch = id % CPR;
      const u32x4 v = *reinterpret_cast<const u32x4*>(Ct + row * RD + ch * 4);
      const int n = n0 + ch * 8;
      // rows past M fail the range check; chunks past Cout (a partial last column tile) are dropped here
      const unsigned voff = (n < p.Cout) ? (unsigned)row * row_b16 + (unsigned)n * 2u : OOB;
      __builtin_amdgcn_raw_buffer_store_b128(v, rsY16, voff, 0, KFN_NT_STORE_AUX);
    }
    return;
  }
  if constexpr (HEAD_EPI) {
    switch (p.epilogue) {
      case KFN_EPI_L2NORM: epilogue(std::integral_constant<int, KFN_EPI_L2NORM>{}); break;
      case KFN_EPI_EXP_CH3: epilogue(std::integral_constant<int, KFN_EPI_EXP_CH3>{}); break;
      case KFN_EPI_EXP_1E2: epilogue(std::integral_constant<int, KFN_EPI_EXP_1E2>{}); break;
      default: epilogue(std::integral_constant<int, KFN_EPI_NONE>{}); break;
    }
  } else {
    epilogue(std::integral_constant<int, KFN_EPI_NONE>{});
  }
}

struct TileCfg {
  int cfg, bm, bn;
  double prior;   // measured MFMA utilisation of the instantiation on large problems (direct)
  double wprior;  // same for the Winograd GEMMs (k-step 16: 128x128 keeps 3 workgroups per CU)
};
const TileCfg kCfgs[] = {{KFN_CFG_160x128, 160, 128, 0.87, 0.70}, {KFN_CFG_128x128, 128, 128, 0.885, 0.77},
                         {KFN_CFG_192x64, 192, 64, 0.78, 0.66},   {KFN_CFG_128x64, 128, 64, 0.76, 0.66},
                         {KFN_CFG_256x32, 256, 32, 0.70, 0.45},   {KFN_CFG_128x32, 128, 32, 0.60, 0.45},
                         {KFN_CFG_64x64, 64, 64, 0.60, 0.60},     {KFN_CFG_160x256, 160, 256, 0.0, 0.0},
                         {KFN_CFG_128x256, 128, 256, 0.0, 0.82},
                         {KFN_CFG_256x16, 256, 16, 0.50, 0.0},    {KFN_CFG_128x16, 128, 16, 0.45, 0.0},
                         {KFN_CFG_256x64, 256, 64, 0.0, 0.0},     {KFN_CFG_256x256, 256, 256, 0.0, 0.0},
                         {KFN_CFG_256x256_W8, 256, 256, 0.0, 0.0}, {KFN_CFG_512x64, 512, 64, 0.0, 0.0}};

const TileCfg* find_cfg(int cfg) {
  for (const TileCfg& c : kCfgs)
    if (c.cfg == cfg) return &c;
  return nullptr;
}

template <int TM, int TN, int WM, int WN, int BK, int MODE, int F16 = PREC_F32>
int launch_cfg(const ConvArgs& a0, hipStream_t stream) {
  constexpr int BM = 32 * TM * WM, BN = (F16 == PREC_F32_N16 ? 16 : 32) * TN * WN, NT = 64 * WM * WN;
  constexpr size_t smem_ops = (size_t)((F16 == PREC_F16_XY_DMA ? 3 : 2) * BM + ((F16 == PREC_F16_XY_BDMA || F16 == PREC_F16_XY_DMA) ? 3 : 2) * BN) * BK * sizeof(float) * (F16 == PREC_F16X3 ? 2 : 1);
  constexpr size_t smem_epi = (F16 == PREC_F16_Y || F16 == PREC_F16_XY || F16 == PREC_F16_XY_BDMA || F16 == PREC_F16_XY_DMA) ? (size_t)BM * BN * 2 : 0;   // fp16 output tile
  constexpr size_t smem = smem_ops > smem_epi ? smem_ops : smem_epi;
  ConvArgs a = a0;
  a.tiles_m = kfn::ceil_div(a.M, BM);
  a.tiles_n = kfn::ceil_div(a.Cout, BN);
  a.fd_cls = make_fastdiv(MODE == MODE_DECONV ? a.N * a.H * a.W : 1);
  a.fd_img = make_fastdiv((MODE == MODE_DECONV || MODE == MODE_CVOL) ? a.H * a.W : a.Ho * a.Wo);
  a.fd_row = make_fastdiv((MODE == MODE_DECONV || MODE == MODE_CVOL) ? a.W : a.Wo);
  auto kern = conv_mfma_kernel<TM, TN, WM, WN, BK, MODE, F16>;
  static std::atomic<uint64_t> attr_done{0};   // per kernel instantiation: bit per device
  {
    int rc = kfn::set_max_dynamic_lds(reinterpret_cast<const void*>(kern), (int)smem, attr_done);
    if (rc != KFN_OK) return rc;
  }
  dim3 grid(a.tiles_m * a.tiles_n * (MODE == MODE_WINO ? 16 : 1)), block(NT);
  hipLaunchKernelGGL(kern, grid, block, smem, stream, a);
  KFN_LAUNCH_CHECK("conv_mfma_kernel");
  return KFN_OK;
}

template <int BK, int TR, int F16 = PREC_F32>
int dispatch_cfg(int cfg, const ConvArgs& a, hipStream_t s) {
  switch (cfg) {
    case KFN_CFG_160x128: return launch_cfg<5, 1, 1, 4, BK, TR, F16>(a, s);
    case KFN_CFG_128x128: return launch_cfg<2, 2, 2, 2, BK, TR, F16>(a, s);
    case KFN_CFG_128x64: return launch_cfg<2, 1, 2, 2, BK, TR, F16>(a, s);
    case KFN_CFG_128x32: return launch_cfg<1, 1, 4, 1, BK, TR, F16>(a, s);
    case KFN_CFG_64x64: return launch_cfg<1, 1, 2, 2, BK, TR, F16>(a, s);
    case KFN_CFG_256x32: return launch_cfg<2, 1, 4, 1, BK, TR, F16>(a, s);
    case KFN_CFG_192x64: return launch_cfg<3, 1, 2, 2, BK, TR, F16>(a, s);
    case KFN_CFG_160x256: return launch_cfg<5, 1, 1, 8, BK, TR, F16>(a, s);
    case KFN_CFG_128x256: return launch_cfg<2, 4, 2, 2, BK, TR, F16>(a, s);
    case KFN_CFG_256x16:
    case KFN_CFG_128x16:
      // 16-column tiles: fp32 operands, k-step 16, direct and transposed convolutions
      if constexpr (F16 == PREC_F32 && BK == 16 && (TR == MODE_CONV || TR == MODE_DECONV)) {
        return cfg == KFN_CFG_256x16 ? launch_cfg<2, 1, 4, 1, 16, TR, PREC_F32_N16>(a, s)
                                     : launch_cfg<1, 1, 4, 1, 16, TR, PREC_F32_N16>(a, s);
      } else {
        return kfn::fail(KFN_ERR_ARG, "kfn_conv2d_nhwc: the 16-column tiles need fp32 operands and k-step 16");
      }
    default: return kfn::fail(KFN_ERR_ARG, "kfn_conv2d_nhwc: unknown config %d", cfg);
  }
}

// fp16 ACTIVATIONS (kfn_conv_desc.x_dtype / y_dtype = KFN_ACT_F16): the wide-tile instantiations only.
template <int BK, int PREC>
int dispatch_f16io(int cfg, const ConvArgs& a, hipStream_t s) {
  switch (cfg) {
    case KFN_CFG_128x128: return launch_cfg<2, 2, 2, 2, BK, MODE_CONV, PREC>(a, s);
    case KFN_CFG_128x256: return launch_cfg<2, 4, 2, 2, BK, MODE_CONV, PREC>(a, s);
    case KFN_CFG_192x64: return launch_cfg<3, 1, 2, 2, BK, MODE_CONV, PREC>(a, s);
    case KFN_CFG_128x64: return launch_cfg<2, 1, 2, 2, BK, MODE_CONV, PREC>(a, s);
    case KFN_CFG_256x64: return launch_cfg<2, 2, 4, 1, BK, MODE_CONV, PREC>(a, s);
    case KFN_CFG_256x256: return launch_cfg<4, 4, 2, 2, BK, MODE_CONV, PREC>(a, s);
    case KFN_CFG_256x256_W8: return launch_cfg<4, 2, 2, 4, BK, MODE_CONV, PREC>(a, s);
    case KFN_CFG_512x64: return launch_cfg<4, 2, 4, 1, BK, MODE_CONV, PREC>(a, s);
    default: return kfn::fail(KFN_ERR_ARG, "kfn_conv2d_nhwc: config %d has no fp16-activation instantiation", cfg);
  }
}
// Tile and k-step of an fp16-activation layer (measured at config 5's shapes, tools/mb_f16.py,
// profiles/r03_c5_layer_microbench.log): with chunk-major weights the 128x256 tile at k-step 16 (48 KiB of LDS, two
// workgroups per CU) wins on every layer with >= 256 output channels (1000-1040 TFLOP/s; 128x128 900-940; k-step
// 32 -- 96 KiB, one workgroup per CU -- 790-870); 64-channel layers (conv1b) take 256x64, four waves side by side in M
// (576 vs 524 TFLOP/s on 128x64).  desc->config / desc->k_step override either.
void f16io_plan(const kfn_conv_desc* d, int* cfg, int* bk) {
  int c = d->config, k = d->k_step;
  // (round 4: the eight-wave 256x256 tile -- half the weight transfers per MFMA, one workgroup of two waves per SIMD --
  //  beats 128x256 on every >= 256-channel layer of config 5: conv4b 1144 vs 1085, conv3b 1112 vs 1069, conv5 1141 vs
  //  1077, conv6 1104 vs 1049, conv2b 1032 vs 1024, conv3a 1000 vs 986 TFLOP/s; profiles/r04_c5_layer_microbench.log)
  if (c == KFN_CFG_AUTO) c = d->Cout >= 256 ? KFN_CFG_256x256_W8 : (d->Cout >= 128 ? KFN_CFG_128x128 : KFN_CFG_256x64);
  if (k == 0 || d->Cin % 64 != 0) k = 16;
  *cfg = c;
  *bk = k;
}
int f16io_config(const kfn_conv_desc* d, int M) {
  (void)M;
  int c, k;
  f16io_plan(d, &c, &k);
  return c;
}
int f16io_bk(const kfn_conv_desc* d) {
  int c, k;
  f16io_plan(d, &c, &k);
  return k;
}

// Tile choice: maximise (useful MFMA work) / (CU-rounds * tile work) over the CUs.
int auto_config(int M, int Cout, int num_cu, bool wino = false, bool n16 = false) {
  double best = -1.0;
  int best_cfg = KFN_CFG_128x32;
  for (const TileCfg& c : kCfgs) {
    if (c.bn == 16 && (!n16 || Cout > 16)) continue;   // 16-column tiles: fp32 direct/transposed convs with <= 16 channels
    const long groups = wino ? 16 : 1;  // the 16 Winograd GEMMs are one launch
    long tiles = (long)kfn::ceil_div(M, c.bm) * kfn::ceil_div(Cout, c.bn) * groups;
    long rounds = (tiles + num_cu - 1) / num_cu;
    double eff = ((double)M * Cout * groups) / ((double)rounds * num_cu * c.bm * c.bn) * (wino ? c.wprior : c.prior);
    if (eff > best) {
      best = eff;
      best_cfg = c.cfg;
    }
  }
  return best_cfg;
}

int g_num_cu = 0;

// k-step per mode.  BK = 16 halves the LDS tile (36 KiB at 160x128) so THREE workgroups fit a
// CU (VGPR-limited to 3 waves/SIMD): measured +3 % (160x128), +16 % (192x64), +33 % (256x32)
// over BK = 32 on the direct kernel.  The Winograd GEMMs need the 128x128 tile for that
// (166 VGPRs; the 160x128 one has 205): 128x128x16 beats 160x128x32 by 1-7 %.
// (k-step 32 is instantiated for the fp16-activation kernels only.)
int pick_bk(int cin, int mode) {
  (void)cin; (void)mode;
  return 16;
}

// K-chunk rotation between concurrent workgroups (ConvArgs::rot_mode 1 / 2): measured without effect on
// MI355X (326.4 / 326.5 / 326.6 frames/s for 0 / 1 / 2); off.
constexpr int kRotMode = 0;

int num_cu() {
  if (g_num_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
        prop.multiProcessorCount > 0)
      g_num_cu = prop.multiProcessorCount;
    else
      g_num_cu = 256;  // MI355X
  }
  return g_num_cu;
}

int validate(const kfn_conv_desc* d) {
  KFN_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0, "kfn_conv2d_nhwc: bad shape %dx%dx%d", d->N, d->H, d->W);
  KFN_REQUIRE(d->Cin > 0 && d->Cin % 16 == 0, "kfn_conv2d_nhwc: Cin=%d must be a multiple of 16", d->Cin);
  KFN_REQUIRE(d->ldx >= d->Cin && d->ldx % 4 == 0, "kfn_conv2d_nhwc: bad ldx=%d", d->ldx);
  KFN_REQUIRE(d->Cout > 0 && d->ldy >= d->Cout, "kfn_conv2d_nhwc: bad Cout=%d ldy=%d", d->Cout, d->ldy);
  KFN_REQUIRE(d->cout_pad >= d->Cout && d->cout_pad % 32 == 0, "kfn_conv2d_nhwc: bad cout_pad=%d", d->cout_pad);
  KFN_REQUIRE(d->kh > 0 && d->kw > 0 && d->kh * d->kw <= 32, "kfn_conv2d_nhwc: bad kernel %dx%d", d->kh, d->kw);
  KFN_REQUIRE(d->stride == 1 || d->stride == 2, "kfn_conv2d_nhwc: stride %d unsupported", d->stride);
  KFN_REQUIRE(!d->transposed || d->stride == 2, "kfn_conv2d_nhwc: transposed conv needs stride 2");
  KFN_REQUIRE(d->epilogue != KFN_EPI_L2NORM || d->Cout == 32, "kfn_conv2d_nhwc: L2NORM epilogue needs Cout == 32");
  return KFN_OK;
}

void out_shape(const kfn_conv_desc* d, int* Ho, int* Wo, int* pad_t, int* pad_l) {
  if (d->transposed) {
    // padding of the forward SAME conv (s*H -> H) whose input-gradient this is
    int o;
    *Ho = d->H * d->stride;
    *Wo = d->W * d->stride;
    kfn::same_pad(*Ho, d->kh, d->stride, &o, pad_t);
    kfn::same_pad(*Wo, d->kw, d->stride, &o, pad_l);
  } else {
    kfn::same_pad(d->H, d->kh, d->stride, Ho, pad_t);
    kfn::same_pad(d->W, d->kw, d->stride, Wo, pad_l);
  }
}

int pick_config(const kfn_conv_desc* d, int M) {
  // the 16-column tiles exist for fp32 operands at k-step 16 without fused head epilogue
  const bool n16_ok = d->operand_dtype == KFN_OPERAND_F32 && d->epilogue == KFN_EPI_NONE &&
                      pick_bk(d->Cin, d->transposed ? MODE_DECONV : MODE_CONV) == 16;
  int cfg = d->config;
  if (cfg == KFN_CFG_AUTO) cfg = auto_config(M, d->Cout, num_cu(), false, n16_ok);
  // the fused head epilogues are only compiled into the 32-column tiles (L2NORM needs BN == 32)
  if (d->epilogue != KFN_EPI_NONE && cfg != KFN_CFG_256x32) cfg = KFN_CFG_128x32;
  return cfg;
}

// fp16 activations in, fp32 out (the heads under an fp16 scope): four tiles are instantiated; any other choice of
// the heuristic maps to the nearest of them instead of failing at launch.
int x16_config(int cfg, int Cout) {
  switch (cfg) {
    case KFN_CFG_256x32: case KFN_CFG_128x32: case KFN_CFG_128x64: case KFN_CFG_128x128: return cfg;
    default: return Cout > 64 ? KFN_CFG_128x128 : (Cout > 32 ? KFN_CFG_128x64 : KFN_CFG_128x32);
  }
}

}  // namespace

extern "C" int kfn_conv2d_out_shape(const kfn_conv_desc* d, int* Ho, int* Wo) {
  KFN_REQUIRE(d && Ho && Wo, "kfn_conv2d_out_shape: null argument");
  KFN_CONV_DESC_IN(d, "kfn_conv2d_out_shape");
  int pt, pl;
  out_shape(d, Ho, Wo, &pt, &pl);
  return KFN_OK;
}

extern "C" int kfn_conv2d_plan(const kfn_conv_desc* d, int* config, int* bk, int* tiles) {
  KFN_REQUIRE(d && config && bk && tiles, "kfn_conv2d_plan: null argument");
  KFN_CONV_DESC_IN(d, "kfn_conv2d_plan");
  int rc = validate(d);
  if (rc != KFN_OK) return rc;
  int Ho, Wo, pt, pl;
  out_shape(d, &Ho, &Wo, &pt, &pl);
  const int M = d->N * Ho * Wo;
  const bool y16 = d->y_dtype == KFN_ACT_F16;
  *config = y16 ? f16io_config(d, M) : pick_config(d, M);
  if (!y16 && d->x_dtype == KFN_ACT_F16 && d->config == KFN_CFG_AUTO) *config = x16_config(*config, d->Cout);
  const TileCfg* c = find_cfg(*config);
  KFN_REQUIRE(c, "kfn_conv2d_plan: unknown config %d", *config);
  *bk = y16 ? f16io_bk(d) : pick_bk(d->Cin, d->transposed ? MODE_DECONV : MODE_CONV);
  *tiles = kfn::ceil_div(M, c->bm) * kfn::ceil_div(d->Cout, c->bn);
  return KFN_OK;
}

extern "C" int kfn_conv2d_nhwc(const kfn_conv_desc* d, const float* x, const float* w_packed,
                               const float* bias, float* y, void* stream) {
  KFN_REQUIRE(d && x && w_packed && y, "kfn_conv2d_nhwc: null argument");
  KFN_CONV_DESC_IN(d, "kfn_conv2d_nhwc");
  int rc = validate(d);
  if (rc != KFN_OK) return rc;
  KFN_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(w_packed) & 15) == 0,
              "kfn_conv2d_nhwc: x / w_packed must be 16-byte aligned");

  ConvArgs a;
  a.x = x; a.x2 = nullptr; a.w = w_packed; a.bias = bias; a.y = y;
  a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.ldx = d->ldx;
  a.Cout = d->Cout; a.cout_pad = d->cout_pad; a.ldy = d->ldy;
  a.kh = d->kh; a.kw = d->kw; a.stride = d->stride;
  a.kw_inv = (65536u + (unsigned)d->kw - 1u) / (unsigned)d->kw;
  a.relu = d->relu; a.epilogue = d->epilogue;
  out_shape(d, &a.Ho, &a.Wo, &a.pad_t, &a.pad_l);
  const long M = (long)d->N * a.Ho * a.Wo;
  const long in_pix = (long)d->N * d->H * d->W;
  const bool x16 = d->x_dtype == KFN_ACT_F16, y16 = d->y_dtype == KFN_ACT_F16;
  KFN_REQUIRE((d->x_dtype == KFN_ACT_F32 || x16) && (d->y_dtype == KFN_ACT_F32 || y16),
              "kfn_conv2d_nhwc: unknown activation dtype %d / %d", d->x_dtype, d->y_dtype);
  if (x16 || y16) {
    KFN_REQUIRE(d->operand_dtype == KFN_OPERAND_F16 && !d->transposed,
                "kfn_conv2d_nhwc: fp16 activations need operand_dtype KFN_OPERAND_F16 and a forward convolution");
    KFN_REQUIRE(!x16 || d->ldx % 8 == 0, "kfn_conv2d_nhwc: fp16 input needs ldx %% 8 == 0 (ldx=%d)", d->ldx);
    KFN_REQUIRE(!y16 || (d->ldy % 8 == 0 && d->Cout % 8 == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0 &&
                         d->epilogue == KFN_EPI_NONE),
                "kfn_conv2d_nhwc: fp16 output needs Cout %% 8 == 0, ldy %% 8 == 0, a 16-byte aligned y and no head epilogue");
  }
  KFN_REQUIRE(d->k_step == 0 || d->k_step == 16 || d->k_step == 32, "kfn_conv2d_nhwc: k_step must be 0, 16 or 32");
  const long xb = x16 ? 2L : 4L;
  const long x_bytes = ((in_pix - 1) * d->ldx + d->Cin) * xb;
  a.Ktot = d->kh * d->kw * d->Cin;
  const bool x3 = d->operand_dtype == KFN_OPERAND_F16X3;
  const bool f16 = d->operand_dtype == KFN_OPERAND_F16 || x3;
  KFN_REQUIRE(d->operand_dtype == KFN_OPERAND_F32 || f16, "kfn_conv2d_nhwc: unknown operand_dtype %d", d->operand_dtype);
  KFN_REQUIRE(!f16 || d->Cin % 32 == 0, "kfn_conv2d_nhwc: fp16 operands need Cin %% 32 == 0 (Cin=%d)", d->Cin);
  const long w_bytes = (long)d->cout_pad * a.Ktot * (f16 ? 2L : 4L) * (x3 ? 2L : 1L);
  a.w_lo_bytes = x3 ? (unsigned)((long)d->cout_pad * a.Ktot * 2L) : 0u;
  a.out_scale = x3 ? (1.0f / 1024.0f) : 1.0f;
  // 32-bit byte offsets (+ the OOB marker 2^31): weights below 2 GiB, and the images one
  // 160-row tile can touch below 2 GiB (the A descriptor is re-based per tile).
  KFN_REQUIRE(!d->transposed || x_bytes < (1L << 31), "kfn_conv2d_nhwc: transposed conv input above 2 GiB");
  const long img_bytes = (long)d->H * d->W * d->ldx * xb;
  const long imgs_per_tile = 160 / ((long)a.Ho * a.Wo) + 2;
  KFN_REQUIRE(M < (1L << 31) && w_bytes < (1L << 31) && img_bytes * imgs_per_tile < (1L << 31),
              "kfn_conv2d_nhwc: tensor too large for 32-bit buffer addressing (image %ld B, w %ld B)", img_bytes,
              w_bytes);
  a.M = (int)M;
  a.x_bytes = (unsigned long long)x_bytes;
  a.w_bytes = (unsigned)w_bytes;
  a.tiles_m = a.tiles_n = 0;
  a.rot_mode = kRotMode;

  const int cfg = pick_config(d, a.M);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (x3) {
    KFN_REQUIRE(!d->transposed, "kfn_conv2d_nhwc: f16x3 operands are implemented for forward convolutions only");
    return dispatch_cfg<16, MODE_CONV, PREC_F16X3>(cfg, a, s);
  }
  if (x16 || y16) {
    // fp16 activations end to end (BASELINE config 5): wide tiles, k-step 32 (64 channels per stage) when Cin allows
    if (y16) {
      const int c16 = f16io_config(d, a.M);
      KFN_REQUIRE(d->weights_path >= KFN_WEIGHTS_AUTO && d->weights_path <= KFN_OPERANDS_LDS_DMA,
                  "kfn_conv2d_nhwc: unknown weights_path %d", d->weights_path);
      // weights global -> LDS directly: +6-8 % on the 128x256 tile (966-1026 -> 1032-1089 TFLOP/s, profiles/
      // r03_c5_layer_microbench.log), neutral on 128x128; AUTO takes it where it pays
      const bool dma = d->weights_path == KFN_WEIGHTS_LDS_DMA ||
                       (d->weights_path == KFN_WEIGHTS_AUTO &&
                        (c16 == KFN_CFG_128x256 || c16 == KFN_CFG_256x256 || c16 == KFN_CFG_256x256_W8));
      // AUTO on the eight-wave tile: both operand tiles global -> LDS directly (+1-3 % over weights only; on the four-wave
      // tiles the activation tile through registers stays ahead)
      if (x16 && f16io_bk(d) == 16 &&
          (d->weights_path == KFN_OPERANDS_LDS_DMA || (d->weights_path == KFN_WEIGHTS_AUTO && c16 == KFN_CFG_256x256_W8))) {
        switch (c16) {
          case KFN_CFG_128x256: return launch_cfg<2, 4, 2, 2, 16, MODE_CONV, PREC_F16_XY_DMA>(a, s);
          case KFN_CFG_128x128: return launch_cfg<2, 2, 2, 2, 16, MODE_CONV, PREC_F16_XY_DMA>(a, s);
          case KFN_CFG_256x64: return launch_cfg<2, 2, 4, 1, 16, MODE_CONV, PREC_F16_XY_DMA>(a, s);
          case KFN_CFG_256x256: return launch_cfg<4, 4, 2, 2, 16, MODE_CONV, PREC_F16_XY_DMA>(a, s);
          case KFN_CFG_256x256_W8: return launch_cfg<4, 2, 2, 4, 16, MODE_CONV, PREC_F16_XY_DMA>(a, s);
          case KFN_CFG_512x64: return launch_cfg<4, 2, 4, 1, 16, MODE_CONV, PREC_F16_XY_DMA>(a, s);
          default: return kfn::fail(KFN_ERR_ARG, "kfn_conv2d_nhwc: config %d has no LDS-DMA instantiation", c16);
        }
      }
      if (x16 && f16io_bk(d) == 16 && dma) {
        switch (c16) {
          case KFN_CFG_128x256: return launch_cfg<2, 4, 2, 2, 16, MODE_CONV, PREC_F16_XY_BDMA>(a, s);
          case KFN_CFG_128x128: return launch_cfg<2, 2, 2, 2, 16, MODE_CONV, PREC_F16_XY_BDMA>(a, s);
          case KFN_CFG_256x256: return launch_cfg<4, 4, 2, 2, 16, MODE_CONV, PREC_F16_XY_BDMA>(a, s);
          case KFN_CFG_256x256_W8: return launch_cfg<4, 2, 2, 4, 16, MODE_CONV, PREC_F16_XY_BDMA>(a, s);
          case KFN_CFG_512x64: return launch_cfg<4, 2, 4, 1, 16, MODE_CONV, PREC_F16_XY_BDMA>(a, s);
          default: return kfn::fail(KFN_ERR_ARG, "kfn_conv2d_nhwc: config %d has no LDS-DMA instantiation", c16);
        }
      }
      if (f16io_bk(d) == 32)
        return x16 ? dispatch_f16io<32, PREC_F16_XY>(c16, a, s) : dispatch_f16io<32, PREC_F16_Y>(c16, a, s);
      return x16 ? dispatch_f16io<16, PREC_F16_XY>(c16, a, s) : dispatch_f16io<16, PREC_F16_Y>(c16, a, s);
    }
    // fp16 in, fp32 out (the heads: 'prediction' with its exp epilogue): the narrow tiles
    switch (d->config == KFN_CFG_AUTO ? x16_config(cfg, d->Cout) : cfg) {
      case KFN_CFG_256x32: return launch_cfg<2, 1, 4, 1, 16, MODE_CONV, PREC_F16_X>(a, s);
      case KFN_CFG_128x32: return launch_cfg<1, 1, 4, 1, 16, MODE_CONV, PREC_F16_X>(a, s);
      case KFN_CFG_128x64: return launch_cfg<2, 1, 2, 2, 16, MODE_CONV, PREC_F16_X>(a, s);
      case KFN_CFG_128x128: return launch_cfg<2, 2, 2, 2, 16, MODE_CONV, PREC_F16_X>(a, s);
      default: return kfn::fail(KFN_ERR_ARG, "kfn_conv2d_nhwc: config %d has no fp16-input instantiation", cfg);
    }
  }
  if (f16) {
    if (d->transposed) return dispatch_cfg<16, MODE_DECONV, PREC_F16>(cfg, a, s);
    return dispatch_cfg<16, MODE_CONV, PREC_F16>(cfg, a, s);
  }
  if (d->transposed) return dispatch_cfg<16, MODE_DECONV>(cfg, a, s);
  return dispatch_cfg<16, MODE_CONV>(cfg, a, s);
}

// ------------------------------------------------------------------------------------
// Winograd F(2x2,3x3) path for 3x3 stride-1 SAME convs: 16 GEMMs (MODE_WINO above) into a
// [tiles][16][Cout] workspace, then the A^T M A output transform + bias + ReLU.
// ------------------------------------------------------------------------------------
namespace {

__global__ __launch_bounds__(256) void wino_output_kernel(const float* __restrict__ ws,
                                                          const float* __restrict__ bias,
                                                          float* __restrict__ y, int Ho, int Wo, int Th,
                                                          int Tw, int Cout, int ldy, int relu, long Mt) {
  const int C4 = Cout >> 2;
  const long total = Mt * C4;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(idx % C4);
    const long t = idx / C4;
    f32x4 m[16];
#pragma unroll
    for (int g = 0; g < 16; ++g)
      m[g] = *reinterpret_cast<const f32x4*>(ws + ((size_t)t * 16 + g) * Cout + c4 * 4);
    // A^T = [[1,1,1,0],[0,1,-1,-1]]
    f32x4 r0[4], r1[4];
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) {
      r0[nu] = (m[0 * 4 + nu] + m[1 * 4 + nu]) + m[2 * 4 + nu];
      r1[nu] = (m[1 * 4 + nu] - m[2 * 4 + nu]) - m[3 * 4 + nu];
    }
    f32x4 o[4];
    o[0] = (r0[0] + r0[1]) + r0[2];
    o[1] = (r0[1] - r0[2]) - r0[3];
    o[2] = (r1[0] + r1[1]) + r1[2];
    o[3] = (r1[1] - r1[2]) - r1[3];
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (bias) bv = *reinterpret_cast<const f32x4*>(bias + c4 * 4);
    const int tx = (int)(t % Tw);
    const long t2 = t / Tw;
    const int ty = (int)(t2 % Th);
    const long n = t2 / Th;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int oy = 2 * ty + a, ox = 2 * tx + b;
        if (oy < Ho && ox < Wo) {
          f32x4 v = o[a * 2 + b] + bv;
          if (relu) {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
          }
          *reinterpret_cast<f32x4*>(y + ((size_t)(n * Ho + oy) * Wo + ox) * ldy + c4 * 4) = v;
        }
      }
  }
}

int wino_validate(const kfn_conv_desc* d) {
  int rc = validate(d);
  if (rc != KFN_OK) return rc;
  KFN_REQUIRE(d->kh == 3 && d->kw == 3 && d->stride == 1 && !d->transposed,
              "kfn_conv2d_winograd: only 3x3 stride-1 SAME convolutions");
  KFN_REQUIRE(d->Cout % 4 == 0 && d->ldy % 4 == 0, "kfn_conv2d_winograd: Cout and ldy must be multiples of 4");
  KFN_REQUIRE(d->epilogue == KFN_EPI_NONE, "kfn_conv2d_winograd: fused head epilogues are not supported");
  KFN_REQUIRE(d->x_dtype == KFN_ACT_F32 && d->y_dtype == KFN_ACT_F32 && d->operand_dtype == KFN_OPERAND_F32,
              "kfn_conv2d_winograd: fp32 operands and fp32 activations in memory only");
  return KFN_OK;
}

}  // namespace

extern "C" int kfn_winograd_workspace_bytes(const kfn_conv_desc* d, size_t* bytes) {
  KFN_REQUIRE(d && bytes, "kfn_winograd_workspace_bytes: null argument");
  KFN_CONV_DESC_IN(d, "kfn_winograd_workspace_bytes");
  int rc = wino_validate(d);
  if (rc != KFN_OK) return rc;
  const size_t Mt = (size_t)d->N * ((d->H + 1) / 2) * ((d->W + 1) / 2);
  *bytes = 16 * Mt * (size_t)d->Cout * sizeof(float);
  return KFN_OK;
}

extern "C" int kfn_winograd_plan(const kfn_conv_desc* d, int* config, int* bk, int* tiles) {
  KFN_REQUIRE(d && config && bk && tiles, "kfn_winograd_plan: null argument");
  KFN_CONV_DESC_IN(d, "kfn_winograd_plan");
  int rc = wino_validate(d);
  if (rc != KFN_OK) return rc;
  const int Mt = d->N * ((d->H + 1) / 2) * ((d->W + 1) / 2);
  *config = d->config == KFN_CFG_AUTO ? auto_config(Mt, d->Cout, num_cu(), true) : d->config;
  const TileCfg* c = find_cfg(*config);
  KFN_REQUIRE(c, "kfn_winograd_plan: unknown config %d", *config);
  *bk = pick_bk(d->Cin, MODE_WINO);
  *tiles = 16 * kfn::ceil_div(Mt, c->bm) * kfn::ceil_div(d->Cout, c->bn);
  return KFN_OK;
}

extern "C" int kfn_conv2d_winograd(const kfn_conv_desc* d, const float* x, const float* u_packed,
                                   const float* bias, float* y, float* workspace, int phases,
                                   void* stream) {
  KFN_REQUIRE(d && x && u_packed && y && workspace, "kfn_conv2d_winograd: null argument");
  KFN_CONV_DESC_IN(d, "kfn_conv2d_winograd");
  int rc = wino_validate(d);
  if (rc != KFN_OK) return rc;
  KFN_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(u_packed) |
                reinterpret_cast<uintptr_t>(workspace) | reinterpret_cast<uintptr_t>(y)) & 15) == 0,
              "kfn_conv2d_winograd: buffers must be 16-byte aligned");
  const int Th = (d->H + 1) / 2, Tw = (d->W + 1) / 2;
  const long Mt = (long)d->N * Th * Tw;
  ConvArgs a;
  a.x = x; a.x2 = nullptr; a.w = u_packed; a.bias = nullptr; a.y = workspace;
  a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.ldx = d->ldx;
  a.Cout = d->Cout; a.cout_pad = d->cout_pad; a.ldy = d->Cout;
  a.kh = 1; a.kw = 1; a.stride = 1; a.pad_t = 1; a.pad_l = 1;
  a.kw_inv = 65536u;
  a.relu = 0; a.epilogue = KFN_EPI_NONE;
  a.Ho = Th; a.Wo = Tw;  // row index space of the GEMMs = 2x2 output tiles
  a.Ktot = d->Cin;
  const long in_pix = (long)d->N * d->H * d->W;
  const long x_bytes = ((in_pix - 1) * d->ldx + d->Cin) * 4L;
  const long w_bytes = (long)d->cout_pad * a.Ktot * 4L;  // one group
  const long img_bytes = (long)d->H * d->W * d->ldx * 4L;
  const long imgs_per_tile = 160 / ((long)Th * Tw) + 2;
  KFN_REQUIRE(Mt < (1L << 31) && w_bytes < (1L << 31) && img_bytes * imgs_per_tile < (1L << 31),
              "kfn_conv2d_winograd: tensor too large for 32-bit buffer addressing");
  a.M = (int)Mt;
  a.x_bytes = (unsigned long long)x_bytes;
  a.w_bytes = (unsigned)w_bytes;
  a.tiles_m = a.tiles_n = 0;
  a.rot_mode = kRotMode;
  a.w_lo_bytes = 0;
  a.out_scale = 1.0f;
  int cfg = d->config;
  if (cfg == KFN_CFG_AUTO) cfg = auto_config(a.M, d->Cout, num_cu(), true);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  KFN_REQUIRE(phases >= 1 && phases <= 3, "kfn_conv2d_winograd: phases must be 1 (GEMMs), 2 (output) or 3");
  if (phases & 1) {
    rc = dispatch_cfg<16, MODE_WINO>(cfg, a, s);
    if (rc != KFN_OK) return rc;
  }
  if (!(phases & 2)) return KFN_OK;
  int Ho, Wo, pt, pl;
  out_shape(d, &Ho, &Wo, &pt, &pl);
  const long total = Mt * (d->Cout / 4);
  long blocks = (total + 255) / 256;
  if (blocks > 256L * 64) blocks = 256L * 64;
  hipLaunchKernelGGL(wino_output_kernel, dim3((unsigned)blocks), dim3(256), 0, s, workspace, bias, y, Ho, Wo, Th,
                     Tw, d->Cout, d->ldy, d->relu, Mt);
  KFN_LAUNCH_CHECK("wino_output_kernel");
  return KFN_OK;
}


// ------------------------------------------------------------------------------------
// KFNet.BuildCoordVolume (KFNet/KFNet.py:343-359,372) fused into OFlowNet's conv0
// (cnn_wrapper/OFlowNet.py:19): the 8x8 local cost volume is generated in the MFMA
// kernel's loader (MODE_CVOL) and never written to HBM.
// ------------------------------------------------------------------------------------
extern "C" int kfn_cost_volume_conv(const float* f1, const float* f2, const float* w_packed,
                                    const float* bias, float* y, int N, int H, int W, int C, int Cout,
                                    int cout_pad, int ldy, int relu, int config, void* stream) {
  KFN_REQUIRE(f1 && f2 && w_packed && y, "kfn_cost_volume_conv: null argument");
  KFN_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 16 == 0, "kfn_cost_volume_conv: bad shape N=%d H=%d W=%d C=%d", N, H, W, C);
  KFN_REQUIRE(Cout > 0 && ldy >= Cout && cout_pad >= Cout && cout_pad % 32 == 0, "kfn_cost_volume_conv: bad Cout/ldy");
  KFN_REQUIRE(((reinterpret_cast<uintptr_t>(f1) | reinterpret_cast<uintptr_t>(f2) |
                reinterpret_cast<uintptr_t>(w_packed)) & 15) == 0, "kfn_cost_volume_conv: misaligned buffer");
  ConvArgs a;
  a.x = f1; a.x2 = f2; a.w = w_packed; a.bias = bias; a.y = y;
  a.N = N; a.H = H; a.W = W; a.Cin = C; a.ldx = C;
  a.Cout = Cout; a.cout_pad = cout_pad; a.ldy = ldy;
  a.kh = 3; a.kw = 3; a.stride = 1; a.pad_t = 1; a.pad_l = 1;
  a.kw_inv = 21846u;
  a.relu = relu; a.epilogue = KFN_EPI_NONE;
  a.Ho = 8; a.Wo = 8;
  const long M = (long)N * H * W * 64;
  const long x_bytes = (long)N * H * W * C * 4L;
  a.Ktot = 9 * C;
  const long w_bytes = (long)cout_pad * a.Ktot * 4L;
  const long img_bytes = (long)H * W * C * 4L;
  KFN_REQUIRE(M < (1L << 31) && w_bytes < (1L << 31) && 3 * img_bytes < (1L << 31),
              "kfn_cost_volume_conv: tensor too large for 32-bit buffer addressing");
  a.M = (int)M;
  a.x_bytes = (unsigned long long)x_bytes;
  a.w_bytes = (unsigned)w_bytes;
  a.tiles_m = a.tiles_n = 0;
  a.rot_mode = 0;
  a.w_lo_bytes = 0;
  a.out_scale = 1.0f;
  int cfg = config;
  if (cfg == KFN_CFG_AUTO) cfg = auto_config(a.M, Cout, num_cu());
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  return dispatch_cfg<16, MODE_CVOL>(cfg, a, s);
}
