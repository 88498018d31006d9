// kfn_wino3.hip -- single-kernel Winograd F(2x2,3x3), four waves sharing ONE input transform.
//
// Same algorithm and data layouts as kfn_wino2.hip (kfn_conv2d_winograd_fused routes here when the layer has
// >= 128 output channels and Cin % 32 == 0).  wino2_kernel pays, per wave of 32 tiles x 32 output channels:
// one input transform (64 packed adds per 64 MFMAs = ~9 % of the fp32 matrix pipe, which VALU work is never
// hidden behind), one read of the raw patch per 32 output channels (the input crosses HBM Cout/32 times) and
// its own prologue round trip.  Here a workgroup of FOUR waves (one per SIMD) owns the same block of 8 x 4
// tiles x 128 output channels (wave w: channels n0 + 32 w ...):
//
//   * V = B^T d B of an 8-channel chunk is computed ONCE, by one wave, and shared through LDS
//     ([16 positions][32 tiles][8 ch] = 16 KiB per chunk; the consumer's fragment address is the producer's
//     store address, so both sides are conflict-free 1 KiB runs).  The producing wave gathers its tiles' raw
//     4x4 patches straight from global memory (16 range-checked loads, zero padding baked into the per-lane
//     offsets), transforms them in registers (64 packed adds) and stores 16 quads.
//   * Work is balanced over SUPER-STEPS of 4 chunks (32 input channels): in super-step k every wave runs the
//     MFMAs of chunks 4k..4k+3 for its 32 channels AND produces chunk 4(k+1)+w for everybody -- so every wave
//     carries exactly one transform per 256 MFMAs (~2.5 %) and one raw-buffer barrier per super-step
//     (16 K cycles) is all the synchronisation there is.  8 V buffers (128 KiB) make that legal: the buffers
//     written during super-step k are the ones read during k-1.
//   * The input crosses HBM Cout/128 times instead of Cout/32, and the prologue is shared.
//   * B fragments as in wino2 (U re-packed [Cin/8][16][Cout][8], 1 KiB contiguous per fragment, L2 -> registers),
//     but the ring is 16 deep (one whole chunk ahead): the V registers of wino2 are gone.
#include "kfn_common.h"
#include <type_traits>
#include <cstdlib>

#ifndef KFN_WINO_DEFAULT_N_FAST
#define KFN_WINO_DEFAULT_N_FAST 0   // measured (kfn_conv_desc.wino_order M_FAST / N_FAST): 2.9 vs 4.8 GB fetched per launch, conv4b 4.59 vs 4.72 ms
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr unsigned OOBV = 0x80000000u;
constexpr int BW = 8, BH = 4;           // tile block: 8 (x) by 4 (y) tiles = 32 MFMA rows
// NW = waves per workgroup = 32-channel column blocks (128 or 64 output channels per workgroup).  NW = 2 is the
// form for layers with <= 64 output channels (SCoordNet conv1b): the two waves share the transform of an 8 x 4
// tile block, a super-step is NW chunks, every wave produces TWO tile rows of 16 channels, and two workgroups
// (72 KiB of V buffers each) share a CU -- still one wave per SIMD.
// One chunk of V in LDS: [16 positions][2 k-halves][32 tiles][4 floats], padded so that the PRODUCER's stores
// (8 consecutive lanes = the 8 channel quads of one tile = 4 chunks x 2 halves) fall on 8 different 16-byte bank
// groups: half stride = 512 + 64, chunk stride = 16 positions + 16 bytes.
// LDS layout of one chunk of V per operand precision H16 (0: fp32 fragments of 16 B, 1: fp16 fragments of 8 B).
// The pads put the 8 (fp32) / 16 (fp16) lanes of one producer store group on different banks.
template <bool H16> struct VLayout {
  static constexpr int FRAG = H16 ? 8 : 16;                 // bytes a lane reads per position: 4 k-values
  static constexpr int VHALF = H16 ? 256 + 16 : 512 + 64;   // [32 tiles][FRAG] + pad
  static constexpr int VPOS = 2 * VHALF;                    // two k-halves
  static constexpr int VBUF = 16 * VPOS + (H16 ? 32 : 16);  // 16 positions + pad
};
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
constexpr int NBR = 16;                 // B ring = one chunk of positions ahead
constexpr int NVR = 8;                  // V fragment ring (positions ahead inside a super-step)
// producer schedule inside a super-step of 256 MFMA slots (tools/mb/wino3_prof.hip sweeps these)
#ifndef KFN_W3_GSTEP
#define KFN_W3_GSTEP 8      // one raw-patch gather every GSTEP slots, from slot 0
#define KFN_W3_XSLOT 175    // the transform burst
#define KFN_W3_SSLOT 192    // first V store
#define KFN_W3_SSTEP 2      // one V store every SSTEP slots
#endif
#ifndef KFN_W3P_GSTEP       // the two-wave form: 128 slots per super-step
#define KFN_W3P_GSTEP 4
#define KFN_W3P_XSLOT 96
#define KFN_W3P_SSLOT 104
#define KFN_W3P_SSTEP 1
#endif

struct Wino3Args {
  const float* x;
  const float* u2;    // [Cin/8][16][cout_pad][8]
  const float* bias;
  float* y;
  int N, H, W, Cin, ldx;
  int Cout, cout_pad, ldy;
  int Th, Tw;
  int vrows;
  int bw;
  int tiles_m, tiles_n;
  int relu;
  int n_fast;          // workgroup order: channel groups of a tile block adjacent (1) or M fastest (0)
  int wide_store;     // Cout, ldy multiples of 4 and y 16-byte aligned: 16-byte stores of the transposed block
  unsigned long long x_bytes;
  unsigned long long y_bytes;
  unsigned u_bytes;
#ifdef KFN_WINO3_PROF
  unsigned long long* prof;   // tools/mb/wino3_prof.hip: [block][wave][8] cycle stamps
#endif
};

#ifdef KFN_WINO3_PROF
// Every lane stores the same counter value to the same address: an `if (lane == 0)` here is divergent control flow,
// after which hipcc no longer trusts the gather descriptors to be uniform and wraps each buffer_load in a
// waterfall loop (4 v_readfirstlane + compare + branch) -- +12 % on the main loop, in the profiled build only.
#define KFN_STAMP(i) (p.prof[((size_t)blockIdx.x * NW + wave) * 8 + (i)] = __builtin_readcyclecounter())
#else
#define KFN_STAMP(i) do { } while (0)
#endif

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

__device__ __forceinline__ int xcd_remap3(int b, int nwg) {
  int xcd = b & 7;
  int q = nwg >> 3, r = nwg & 7;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (b >> 3);
}

__device__ __forceinline__ void pk_sub_ip(f32x2& a, const f32x2& b) {
  asm("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(a) : "v"(b));
}
__device__ __forceinline__ void pk_rsub_ip(f32x2& a, const f32x2& b) {
  asm("v_pk_add_f32 %0, %1, %0 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(a) : "v"(b));
}
__device__ __forceinline__ f32x2 pk_add(const f32x2& a, const f32x2& b) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ f32x2 pk_sub(const f32x2& a, const f32x2& b) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// (d0,d1,d2,d3) -> (d0-d2, d1+d2, d2-d1, d1-d3): one 1-D pass of B^T, in place
__device__ __forceinline__ void bt_pass(f32x2& d0, f32x2& d1, f32x2& d2, f32x2& d3) {
  pk_sub_ip(d0, d2);
  pk_rsub_ip(d3, d1);
  const f32x2 s = pk_add(d1, d2);
  pk_sub_ip(d2, d1);
  d1 = s;
}
__device__ __forceinline__ void bt_d_b(f32x2 (&v)[32]) {   // v[2*(4*r + c) + half] -> v[2*(4*xi + nu) + half]
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int h = 0; h < 2; ++h) bt_pass(v[2 * (4 * r + 0) + h], v[2 * (4 * r + 1) + h], v[2 * (4 * r + 2) + h], v[2 * (4 * r + 3) + h]);
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int h = 0; h < 2; ++h) bt_pass(v[2 * (0 + c) + h], v[2 * (4 + c) + h], v[2 * (8 + c) + h], v[2 * (12 + c) + h]);
}

// H16: BASELINE config 5's fp16-operand convolutions -- V is rounded to fp16 when it is stored to LDS (the input
// transform itself runs in fp32 on the fp32 activations), U arrives as fp16, one v_mfma_f32_32x32x8_f16 per
// (position, 8-channel chunk) replaces four v_mfma_f32_32x32x2_f32; accumulation, bias, output transform fp32.
template <bool H16, int NW>
__device__ __forceinline__ void wino3_body(const Wino3Args& p) {
  static_assert(NW == 4 || (NW == 2 && !H16), "four waves, or the two-wave fp32 form");
  constexpr int CPS = NW;           // chunks (of 8 input channels) per super-step
  constexpr int NVBUF = 2 * CPS;    // chunk buffers: two super-steps
  constexpr int NT = 32 * NW;       // output channels per workgroup
  extern __shared__ __attribute__((aligned(16))) char smem3[];   // [NVBUF][VBUF]
  constexpr int VHALF = VLayout<H16>::VHALF, VPOS = VLayout<H16>::VPOS, VBUF = VLayout<H16>::VBUF, FRAG = VLayout<H16>::FRAG;
  using frag_t = std::conditional_t<H16, f32x2, f32x4>;   // one (position, chunk) fragment of a lane

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform: keep it in an SGPR
  KFN_STAMP(0);
#ifdef KFN_WINO3_PROF
  p.prof[((size_t)blockIdx.x * NW + wave) * 8 + 6] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID
  p.prof[((size_t)blockIdx.x * NW + wave) * 8 + 7] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));  // XCC_ID
#endif
  const int nwg = p.tiles_m * p.tiles_n;
  const int tile = xcd_remap3(blockIdx.x, nwg);
  // Workgroups that run side by side on an XCD share its L2.  n_fast: the Cout/128 channel groups of one tile
  // block are neighbours (the input crosses HBM once, every group's weights are live at once); else M fastest
  // (one channel group's weights stay hot, the input is fetched once per channel group).
  const int tm = p.n_fast ? tile / p.tiles_n : tile % p.tiles_m;
  const int tn = p.n_fast ? tile % p.tiles_n : tile / p.tiles_m;
  const int cb = tm % p.bw, rb = tm / p.bw;
  const int n0 = tn * NT + wave * 32;       // this wave's 32 output channels

  // ---- block geometry (uniform) ----------------------------------------------------------
  const int vr0 = rb * BH;
  const int img0 = vr0 / p.Th;
  const int ty0 = vr0 - img0 * p.Th;
  const int brk = (p.Th - ty0 < BH) ? (p.Th - ty0) : BH;   // tile rows >= brk belong to image img0 + 1

  const unsigned long long a_base = (unsigned long long)img0 * p.H * p.W * p.ldx * 4ull;
  const __amdgpu_buffer_rsrc_t rsU =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.u2), 0, p.u_bytes, 0x00020000);

  // ---- this lane as a PRODUCER: wave w owns tile row w of the block for ALL 32 channels of a super-step ----
  // lane = (tile column tc, chunk cq, k-half h): 8 CONSECUTIVE lanes read the 128 contiguous bytes of a pixel's 32
  // channels (the texture addresser only merges neighbouring lanes: lanes of one quad on four different pixels
  // are four requests), and a lane's 16 loads are the 4x4 patch of tile (w, tc) for channels 8 cq + 4 h ..+3.
  // NW = 2: lane = (tile row of the wave's two rs, tile column tc, chunk cq of 2, k-half h), tc's bit 1 in lane bit 2
  // so that the 8 lanes of a ds_write_b128 group -- (tc bit 1, cq, h) -- store to 8 different 16-byte bank groups;
  // 4 neighbouring lanes read the 64 contiguous bytes of a pixel's 16 channels.
  const int ph = lane & 1;
  const int cq = NW == 4 ? (lane >> 1) & 3 : (lane >> 1) & 1;
  const int tc = NW == 4 ? lane >> 3 : ((lane >> 2) & 1) * 2 + ((lane >> 3) & 1) + ((lane >> 4) & 1) * 4;
  const int ptr = NW == 4 ? wave : 2 * wave + (lane >> 5);   // tile row this lane produces (uniform when NW = 4)
  // NW = 4: a patch pixel's address splits into a wave-uniform row part (tile row = wave): one descriptor per patch
  // row, based at the row's first pixel and exactly one image row long (zero long if the row is above/below the image
  // or the tile row does not exist), and a per-lane column part.  Columns left of the image give a negative =
  // huge unsigned vector offset, columns right of it an offset past the row: both fail the range check and
  // read as zero.
  // NW = 2: the tile row differs between the wave's halves, so ONE descriptor covers the block's (at most two) images
  // and every lane carries the 16 offsets of its patch, poisoned (>= 1 GiB, beyond any two images the launcher
  // admits) where the row or the column falls outside the image.
  unsigned gcol[NW == 4 ? 4 : 1];                   // NW = 4: byte offset of patch column c at this lane's channel quad
  __amdgpu_buffer_rsrc_t rs_row[NW == 4 ? 4 : 1];   // NW = 4: uniform row descriptors; NW = 2: [0] = the block's images
  unsigned goff[NW == 4 ? 1 : 16];                  // NW = 2: per-lane offsets of the 4x4 patch
  {
    const int tr = ptr;
    const int img_rel = tr < brk ? 0 : 1;
    const int ty = tr < brk ? ty0 + tr : tr - brk;
    const int tx = cb * BW + tc;
    const bool tile_ok = (vr0 + tr < p.vrows);
    const int row_bytes = ((p.W - 1) * p.ldx + p.Cin) * 4;
    if constexpr (NW == 4) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int yy = 2 * ty - 1 + r;
        const bool ok = tile_ok && (unsigned)yy < (unsigned)p.H;
        const unsigned long long off =
            ok ? (unsigned long long)((img_rel * p.H + yy) * p.W) * (unsigned long long)(p.ldx * 4) : 0ull;
        rs_row[r] = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(p.x)) + a_base + off, 0, ok ? row_bytes : 0, 0x00020000);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) gcol[c] = (unsigned)(((2 * tx - 1 + c) * p.ldx + cq * 8 + ph * 4) * 4);
    } else {
      const unsigned long long rest = p.x_bytes - a_base;
      const unsigned long long two = 2ull * p.H * p.W * p.ldx * 4ull;
      rs_row[0] = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.x)) + a_base, 0,
                                                    (int)(rest < two ? rest : two), 0x00020000);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int yy = 2 * ty - 1 + r;
        const bool rok = tile_ok && (unsigned)yy < (unsigned)p.H;
        const unsigned roff = rok ? (unsigned)((img_rel * p.H + yy) * p.W) * (unsigned)(p.ldx * 4) : 0x80000000u;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int xx = 2 * tx - 1 + c;
          const unsigned coff = (unsigned)xx < (unsigned)p.W ? (unsigned)((xx * p.ldx + cq * 8 + ph * 4) * 4) : 0x40000000u;
          goff[4 * r + c] = roff + coff;
        }
      }
      (void)row_bytes;
    }
  }
  // consumer fragment of position g: g*VPOS + half*VHALF + tile*16; the producer lane stores into chunk cq
  const int v_lane = (lane >> 5) * VHALF + (lane & 31) * FRAG;
  const int v_st = cq * VBUF + ph * VHALF + ptr * (8 * FRAG) + tc * FRAG;
  const int n_chunks = p.Cin / 8;
  const int n_super = n_chunks / CPS;
  const int s_last = n_super - 1;

  // ---- this lane as a CONSUMER ---------------------------------------------------------------
  const int li = lane & 31, lh = lane >> 5;
  const unsigned voff_b = (unsigned)(((n0 + li) * 8 + lh * 4) * (H16 ? 2 : 4));
  const unsigned b_step = (unsigned)p.cout_pad * (H16 ? 16u : 32u);
  const int q_last = n_chunks * 16 - 1;

  const int n = n0 + li;
  const bool n_ok = n < p.Cout;
  const float bv = (p.bias != nullptr && n_ok) ? p.bias[n] : 0.f;
  f32x16 acc[16];
#pragma unroll
  for (int g = 0; g < 16; ++g)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[g][e] = (g == 5) ? bv : 0.f;   // bias rides in position (1,1), cf. kfn_wino2.hip

  // At fp16 MFMA rates a super-step is ~2 K cycles -- less than one HBM round trip and barely one L2 round trip --
  // so the fp16 instantiation looks further ahead: raw patches are gathered TWO super-steps ahead into a second
  // register buffer (transformed and stored one super-step later, when they have long arrived), and the B ring is
  // two chunks deep.  (fp16 fragments are half the registers, which pays for both.)
  constexpr int NPB = H16 ? 2 : 1;            // raw-patch register buffers
  constexpr int NB = H16 ? 2 * NBR : NBR;     // B ring depth in fragments
  f32x2 pv[NPB][32];   // producer: raw patch -> V of the tile row this wave produces
  frag_t bq[NB];       // B ring
  frag_t vq[NVR];      // V fragment ring

  // producer steps for super-step `ss` (clamped past the end: chunks nobody will read), register buffer `pb`
  auto p_gather = [&](auto ic, int ss, auto pb_) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value, pb = decltype(pb_)::value;
    constexpr int r = i >> 2, c = i & 3;
    const int sc = ss < s_last ? ss : s_last;
    f32x4 q;
    if constexpr (NW == 4) q = bload(rs_row[r], gcol[c], (unsigned)(sc * 128));
    else q = bload(rs_row[0], goff[i], (unsigned)(sc * 64));
    pv[pb][2 * i] = q.xy;
    pv[pb][2 * i + 1] = q.zw;
  };
  auto p_store = [&](auto gc, int ss, auto pb_) __attribute__((always_inline)) {
    constexpr int g = decltype(gc)::value, pb = decltype(pb_)::value;
    if constexpr (H16) {
      const f16x4 q = {(_Float16)pv[pb][2 * g].x, (_Float16)pv[pb][2 * g].y, (_Float16)pv[pb][2 * g + 1].x,
                       (_Float16)pv[pb][2 * g + 1].y};   // RNE
      *reinterpret_cast<f16x4*>(smem3 + (ss & 1) * (CPS * VBUF) + v_st + g * VPOS) = q;
    } else {
      const f32x4 q = {pv[pb][2 * g].x, pv[pb][2 * g].y, pv[pb][2 * g + 1].x, pv[pb][2 * g + 1].y};
      *reinterpret_cast<f32x4*>(smem3 + (ss & 1) * (CPS * VBUF) + v_st + g * VPOS) = q;
    }
  };
  // fragment qidx = chunk*16 + position into ring slot `sl`
  auto b_load = [&](auto sl_, int qidx) __attribute__((always_inline)) {
    constexpr int sl = decltype(sl_)::value;
    const int qc = qidx < q_last ? qidx : q_last;
    if constexpr (H16) bq[sl] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsU, voff_b, (unsigned)qc * b_step, 0));
    else bq[sl] = bload(rsU, voff_b, (unsigned)qc * b_step);
  };
  auto v_read = [&](auto gc, int ch) __attribute__((always_inline)) {
    constexpr int g = decltype(gc)::value;
    vq[g % NVR] = *reinterpret_cast<const frag_t*>(smem3 + (ch & (NVBUF - 1)) * VBUF + g * VPOS + v_lane);
  };

  KFN_STAMP(1);
  // ---- prologue: every wave produces its tile row of super-step 0 -------------------------------
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  sfor<16>([&](auto ic) { p_gather(ic, 0, I0{}); });
  sfor<NB>([&](auto gc) { b_load(gc, decltype(gc)::value); });
  bt_d_b(pv[0]);
  sfor<16>([&](auto gc) { p_store(gc, 0, I0{}); });
  if constexpr (H16) sfor<16>([&](auto ic) { p_gather(ic, 1, I1{}); });   // super-step 1: transformed during super-step 0
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  KFN_STAMP(2);
  // One super-step: the MFMAs of chunks c0 .. c0+3 (slot j = 4 interleaved positions x 4 k-steps), and as a
  // producer: gather chunk c0+4+wave behind the first MFMAs of chunk c0, transform it as ONE burst in the
  // middle of chunk c0+2, store it behind the MFMAs of chunk c0+3; then the barrier.
#ifdef KFN_W3_TL
#ifndef KFN_W3_TL_KS
#define KFN_W3_TL_KS (-1)   // the super-step whose timeline is kept (-1: the last one)
#endif
  unsigned long long tl[17];
#endif
  // MFMA slots per chunk and the producer's schedule inside the 4-chunk super-step.  fp32: gather super-step ks+1
  // early, transform late, store.  fp16: transform + store super-step ks+1 (gathered during ks-1) first, then gather
  // ks+2 into the buffer that just became free.
  constexpr int SPC = H16 ? 16 : 64;
  constexpr int GSLOT = H16 ? 32 : 0, GSTEP = H16 ? 2 : (NW == 4 ? KFN_W3_GSTEP : KFN_W3P_GSTEP);
  constexpr int XSLOT = H16 ? 8 : (NW == 4 ? KFN_W3_XSLOT : KFN_W3P_XSLOT);
  constexpr int SSLOT = H16 ? 12 : (NW == 4 ? KFN_W3_SSLOT : KFN_W3P_SSLOT), SSTEP = H16 ? 1 : (NW == 4 ? KFN_W3_SSTEP : KFN_W3P_SSTEP);
  static_assert(GSLOT + 16 * GSTEP <= CPS * SPC && SSLOT + 16 * SSTEP <= CPS * SPC && XSLOT < SSLOT, "producer schedule inside the super-step");
  auto super_step = [&](int ks, auto par_) __attribute__((always_inline)) {
    constexpr int par = decltype(par_)::value;        // fp16: ks & 1 = the buffer this super-step gathers into
    using GB = std::integral_constant<int, par>;               // gather buffer
    using XB = std::integral_constant<int, H16 ? par ^ 1 : 0>;  // transform / store buffer
    const int c0 = ks * CPS;
    const int g_ss = H16 ? ks + 2 : ks + 1;   // super-step being gathered
    const int x_ss = ks + 1;                  // super-step being transformed and stored
    // the V fragments of the first NVR positions (nothing of this super-step could be read before the barrier)
    sfor<NVR>([&](auto gc) { v_read(gc, c0); });
    sfor<CPS>([&](auto cc_) {
      constexpr int cc = decltype(cc_)::value;
      const int ch = c0 + cc;
      const int qbase = ch * 16;
      sfor<SPC>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        // fp32: 4 interleaved positions x 4 k-steps of 32x32x2; fp16: one 32x32x8 per position
        constexpr int g = H16 ? j : (j >> 4) * 4 + (j & 3), t = H16 ? 3 : (j >> 2) & 3;
        constexpr int sl = (cc * 16 + g) % NB;     // B ring slot of fragment ch*16 + g (c0 is a multiple of 4)
#ifdef KFN_W3_TL
        if constexpr (!H16 && (cc * 64 + j) % 16 == 0)
          if (KFN_W3_TL_KS < 0 || ks == KFN_W3_TL_KS) tl[(cc * 64 + j) / 16] = __builtin_readcyclecounter();
#endif
        if constexpr (H16)
          acc[g] = __builtin_amdgcn_mfma_f32_32x32x8f16(__builtin_bit_cast(f16x4, vq[g % NVR]), __builtin_bit_cast(f16x4, bq[sl]), acc[g], 0, 0, 0);
        else
          acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(vq[g % NVR][t], bq[sl][t], acc[g], 0, 0, 0);
        if constexpr (t == 3) {
          b_load(std::integral_constant<int, sl>{}, qbase + g + NB);
          // V fragment NVR positions ahead: same chunk, or the next chunk of THIS super-step
          if constexpr (g + NVR < 16) v_read(std::integral_constant<int, g + NVR>{}, ch);
          else if constexpr (cc < CPS - 1) v_read(std::integral_constant<int, g + NVR - 16>{}, ch + 1);
        }
        // producer work, spread thin: the four waves of a CU share one texture addresser and one LDS port, and a
        // wave whose vector-memory instruction cannot issue stalls its MFMAs behind it
        constexpr int sj = cc * SPC + j;   // slot inside the super-step
        if constexpr (sj >= GSLOT && sj < GSLOT + 16 * GSTEP && (sj - GSLOT) % GSTEP == 0)
          p_gather(std::integral_constant<int, (sj - GSLOT) / GSTEP>{}, g_ss, GB{});
        if constexpr (sj == XSLOT) bt_d_b(pv[XB::value]);
        if constexpr (sj >= SSLOT && sj < SSLOT + 16 * SSTEP && (sj - SSLOT) % SSTEP == 0)
          p_store(std::integral_constant<int, (sj - SSLOT) / SSTEP>{}, x_ss, XB{});
        __builtin_amdgcn_sched_barrier(0);
      });
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#ifdef KFN_W3_TL
    if (KFN_W3_TL_KS < 0 || ks == KFN_W3_TL_KS) tl[16] = __builtin_readcyclecounter();
#endif
  };
  if constexpr (H16) {
    for (int ks = 0; ks < n_super; ks += 2) {   // n_super is even (Cin % 64 == 0): two super-steps per trip, one per buffer
      super_step(ks, I0{});
      super_step(ks + 1, I1{});
    }
  } else {
    for (int ks = 0; ks < n_super; ++ks) super_step(ks, I0{});
  }

#ifdef KFN_W3_TL
#pragma unroll
  for (int i = 0; i < 17; ++i) p.prof[((size_t)gridDim.x * NW) * 8 + ((size_t)blockIdx.x * NW + wave) * 17 + i] = tl[i];
#endif
  KFN_STAMP(3);
  // ---- epilogue (per wave, as kfn_wino2.hip) -----------------------------------------------------
  const bool relu = p.relu != 0;
  const unsigned long long y_base = (unsigned long long)img0 * p.H * p.W * p.ldy * 4ull;
  const unsigned long long y_rest = p.y_bytes - y_base;
  const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<char*>(p.y) + y_base, 0, (int)(y_rest < 0x7fffffffull ? y_rest : 0x7fffffffull), 0x00020000);
  const int txl = cb * BW + 4 * lh;
  const unsigned voff_y = (unsigned)((2 * txl * p.ldy + n) * 4);
  const int pix_bytes = p.ldy * 4;
  // Output transform (16 positions -> 2x2 pixels, packed adds), then the stores.  Global stores are ISSUE bound
  // (~60 cycles per store instruction per CU whatever its width), so 64 dword stores per wave cost ~16 K cycles.
  // Instead every wave transposes its 128 pixels x 32 channels through its own 16 KiB of the (now idle) V
  // buffers -- [pixel][32 channels], written by ds_write_b32 (32 lanes = one 128-byte pixel row), read back as
  // 1 KiB runs -- and issues 16 stores of 16 bytes per lane: 8 lanes cover the 128 contiguous bytes of a pixel.
  auto out_transform = [&](auto&& put) __attribute__((always_inline)) {
#pragma unroll
    for (int ep = 0; ep < 8; ++ep) {
      const int e0 = 2 * ep;
      f32x2 r0[4], r1[4];
#pragma unroll
      for (int nu = 0; nu < 4; ++nu) {
        const f32x2 m0 = {acc[0 + nu][e0], acc[0 + nu][e0 + 1]}, m1 = {acc[4 + nu][e0], acc[4 + nu][e0 + 1]};
        const f32x2 m2 = {acc[8 + nu][e0], acc[8 + nu][e0 + 1]}, m3 = {acc[12 + nu][e0], acc[12 + nu][e0 + 1]};
        r0[nu] = pk_add(pk_add(m0, m1), m2);
        r1[nu] = pk_sub(pk_sub(m1, m2), m3);
      }
      const f32x2 o0 = pk_add(pk_add(r0[0], r0[1]), r0[2]), o1 = pk_sub(pk_sub(r0[1], r0[2]), r0[3]);
      const f32x2 o2 = pk_add(pk_add(r1[0], r1[1]), r1[2]), o3 = pk_sub(pk_sub(r1[1], r1[2]), r1[3]);
      const int trow = e0 >> 2, ec = e0 & 3;
      put(o0.x, trow, ec, 0, 0); put(o1.x, trow, ec, 0, 1); put(o2.x, trow, ec, 1, 0); put(o3.x, trow, ec, 1, 1);
      put(o0.y, trow, ec + 1, 0, 0); put(o1.y, trow, ec + 1, 0, 1); put(o2.y, trow, ec + 1, 1, 0); put(o3.y, trow, ec + 1, 1, 1);
    }
  };
  if (p.wide_store) {
    char* const stg = smem3 + wave * (H16 ? 16384 : VBUF);   // 16 KiB per wave inside the idle V buffers
    const int st_w = lh * 1024 + li * 4;   // block pixel (oy, ox) = (2 trow + a, 8 lh + 2 ec + b) -> row oy*16 + ox
    out_transform([&](float v, int trow, int ec, int a, int b) __attribute__((always_inline)) {
      *reinterpret_cast<float*>(stg + st_w + ((2 * trow + a) * 16 + 2 * ec + b) * 128) = v;
    });
#ifdef KFN_W3_EPI
    KFN_STAMP(6);
#endif
    const int oxl = lane >> 3, nq = lane & 7;            // store lane: pixel column oxl (+8), channel quad nq
    const unsigned voff_q = (unsigned)((oxl * p.ldy + n0 + nq * 4) * 4);
    const bool q_ok = n0 + nq * 4 < p.Cout;
    const int ox0 = 2 * cb * BW;
    const unsigned voff_h[2] = {(q_ok && ox0 + oxl < p.W) ? voff_q : OOBV, (q_ok && ox0 + 8 + oxl < p.W) ? voff_q : OOBV};
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      f32x4 v = *reinterpret_cast<const f32x4*>(stg + i * 1024 + lane * 16);
      if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      const int trow = i >> 2, a = (i >> 1) & 1, hx = i & 1;
      const int img_rel = trow < brk ? 0 : 1;
      const int ty = trow < brk ? ty0 + trow : trow - brk;
      const int oy = 2 * ty + a;
      const bool row_ok = vr0 + trow < p.vrows && oy < p.H;        // uniform
      const unsigned soff = (unsigned)(((img_rel * p.H + oy) * p.W + ox0 + 8 * hx) * pix_bytes);
      kfn::buffer_store_b128<KFN_NT_STORE_AUX>(v, rsY, row_ok ? voff_h[hx] : OOBV, soff);
    }
  } else {
    // Cout or the row pitch not a multiple of 4 floats: one dword per store
    out_transform([&](float v, int trow, int ec, int a, int b) __attribute__((always_inline)) {
      const int img_rel = trow < brk ? 0 : 1;
      const int ty = trow < brk ? ty0 + trow : trow - brk;
      const int oy = 2 * ty + a;
      v = relu ? fmaxf(v, 0.f) : v;
      const unsigned soff = (unsigned)(((img_rel * p.H + oy) * p.W + 2 * ec + b) * pix_bytes);
      const int tx = txl + ec;
      const bool ok = n_ok && tx < p.Tw && 2 * tx + b < p.W && vr0 + trow < p.vrows && oy < p.H;
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsY, ok ? voff_y : OOBV, soff, 0);
    });
  }
  KFN_STAMP(4);
#ifdef KFN_WINO3_PROF
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  KFN_STAMP(5);
#endif
}

// the four-wave form (128 output channels per workgroup) and the two-wave form (33 .. 64 output channels in all)
template <bool H16>
__global__ __launch_bounds__(256, 1) void wino3_kernel(Wino3Args p) { wino3_body<H16, 4>(p); }
__global__ __launch_bounds__(128, 1) void wino3_pair_kernel(Wino3Args p) { wino3_body<false, 2>(p); }

}  // namespace

#ifdef KFN_WINO3_PROF
unsigned long long* g_wino3_prof = nullptr;
#endif

namespace kfn {

// (the routing rule of kfn_conv2d_winograd_fused, kfn_wino2.hip, without its byte-range conditions)
int wino_fused_lds_bytes(const kfn_conv_desc* d) {
  const bool h16 = d->operand_dtype == KFN_OPERAND_F16;
  const bool one = d->wino_form == KFN_WINO_FORM_ONE_WAVE;
  if ((!one || h16) && d->Cout >= 128 && d->Cin % 32 == 0) return 8 * (h16 ? VLayout<true>::VBUF : VLayout<false>::VBUF);
  if (!one && !h16 && d->cout_pad == 64 && d->Cin % 16 == 0) return 4 * VLayout<false>::VBUF;
  return h16 ? -1 : WINO2_LDS_BYTES;
}

// Launch of the 4-wave form (Cin % 32 == 0, Cout >= 128) or the two-wave form (cout_pad == 64, Cin % 16 == 0) for
// a descriptor kfn_conv2d_winograd_fused has already validated.  Called from kfn_wino2.hip.
int launch_wino3(const kfn_conv_desc* d, const float* x, const void* u2_packed, const float* bias, float* y,
                 hipStream_t stream) {
  const bool h16 = d->operand_dtype == KFN_OPERAND_F16;
  Wino3Args a;
  a.x = x; a.u2 = static_cast<const float*>(u2_packed); a.bias = bias; a.y = y;
  a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.ldx = d->ldx;
  a.Cout = d->Cout; a.cout_pad = d->cout_pad; a.ldy = d->ldy;
  a.Th = (d->H + 1) / 2; a.Tw = (d->W + 1) / 2;
  const long vrows = (long)d->N * a.Th;
  const long in_pix = (long)d->N * d->H * d->W;
  // (the caller checked the per-image and weight byte ranges; the batch-wide row count is checked here so that a
  // very large N cannot overflow vrows and the img0 / ty0 arithmetic derived from it)
  KFN_REQUIRE(vrows < (1L << 30), "kfn_conv2d_winograd_fused: N*ceil(H/2) = %ld tile rows exceed 32-bit addressing", vrows);
  a.vrows = (int)vrows;
  a.bw = ceil_div(a.Tw, BW);
  const long tiles_m = (long)a.bw * ceil_div(a.vrows, BH);
  const bool pair = !h16 && d->cout_pad == 64;          // the two-wave form: all of a layer's 33..64 output channels in one workgroup
  a.tiles_n = ceil_div(d->cout_pad, pair ? 64 : 128);
  KFN_REQUIRE(tiles_m * a.tiles_n < (1L << 31), "kfn_conv2d_winograd_fused: grid too large");
  a.tiles_m = (int)tiles_m;
  a.relu = d->relu;
  a.n_fast = d->wino_order == KFN_WINO_ORDER_N_FAST ? 1 : (d->wino_order == KFN_WINO_ORDER_M_FAST ? 0 : KFN_WINO_DEFAULT_N_FAST);
  a.wide_store = (d->Cout % 4 == 0 && d->ldy % 4 == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0) ? 1 : 0;
  a.x_bytes = (unsigned long long)(((in_pix - 1) * d->ldx + d->Cin) * 4L);
  a.y_bytes = (unsigned long long)(((in_pix - 1) * d->ldy + d->Cout) * 4L);
  a.u_bytes = (unsigned)(16L * d->cout_pad * d->Cin * (h16 ? 2L : 4L));
#ifdef KFN_WINO3_PROF
  a.prof = g_wino3_prof;
#endif
  const dim3 grid((unsigned)(a.tiles_m * a.tiles_n));
  if (pair) {
    // per-lane patch offsets carry their out-of-image marks in bits 30 and 31: two images must stay below 1 GiB
    KFN_REQUIRE(2L * d->H * d->W * d->ldx * 4L < (1L << 30), "kfn_conv2d_winograd_fused: image beyond 512 MiB in the two-wave form");
    constexpr int lds = 4 * VLayout<false>::VBUF;
    static std::atomic<uint64_t> attr_done2{0};
    int rc = set_max_dynamic_lds(reinterpret_cast<const void*>(wino3_pair_kernel), lds, attr_done2);
    if (rc != KFN_OK) return rc;
    hipLaunchKernelGGL(wino3_pair_kernel, grid, dim3(128), lds, stream, a);
  } else if (h16) {
    constexpr int lds = 8 * VLayout<true>::VBUF;
    static std::atomic<uint64_t> attr_done16{0};
    int rc = set_max_dynamic_lds(reinterpret_cast<const void*>(wino3_kernel<true>), lds, attr_done16);
    if (rc != KFN_OK) return rc;
    hipLaunchKernelGGL(wino3_kernel<true>, grid, dim3(256), lds, stream, a);
  } else {
    constexpr int lds = 8 * VLayout<false>::VBUF;
    static std::atomic<uint64_t> attr_done{0};
    int rc = set_max_dynamic_lds(reinterpret_cast<const void*>(wino3_kernel<false>), lds, attr_done);
    if (rc != KFN_OK) return rc;
    hipLaunchKernelGGL(wino3_kernel<false>, grid, dim3(256), lds, stream, a);
  }
  KFN_LAUNCH_CHECK(pair ? "wino3_pair_kernel" : "wino3_kernel");
  return KFN_OK;
}

}  // namespace kfn
