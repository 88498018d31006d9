// kfn_wino2.hip -- single-kernel Winograd F(2x2,3x3) convolution (3x3, stride 1, SAME), fp32.
//
// Replaces the (16 GEMMs -> [tiles][16][Cout] workspace -> output transform) pair of
// kfn_conv2d_winograd for tf.layers.conv2d behind Network.conv (cnn_wrapper/network.py:116-135)
// on the wide stride-1 layers of SCoordNet (cnn_wrapper/SCoordNet.py:22-30).
//
// One WAVEFRONT is one workgroup and owns a block of 8 x 4 Winograd tiles (= 16 x 8 output
// pixels = the 32 rows of a 32x32 MFMA) x 32 output channels x ALL 16 transform positions:
// 16 accumulators of 32x32 (256 registers per lane, one wave per SIMD, four per CU).  The
// inverse transform Y = A^T M A therefore happens in registers in the epilogue: there is no
// workspace and no second kernel, and every input pixel is fetched once per workgroup.
//
//   A operand: the block's RAW input patch (18 x 10 pixels, +2 rows when the block straddles
//       two images of the batch) is staged global -> registers -> LDS once per 16-channel stage,
//       each lane copying contiguous 16-byte quads (64 B per pixel).  A lane (tile i, k-half h)
//       reads its tile's 4x4 patch back (16 ds_read_b128), evaluates B^T d B in registers
//       (32 float4 adds) and feeds the 16 results straight to the MFMAs -- the transformed
//       tensor V never exists in memory, on chip or off.
//   B operand: the pre-transformed weights, re-packed [Cin/8][16][Cout][8] so that the
//       fragment of one (k-chunk, position) for 32 channels is ONE contiguous 1 KiB read,
//       go global/L2 -> registers directly through an 8-deep register ring; the four waves of
//       a CU walk the same slice (tiles are ordered M-fastest, so one XCD's L2 holds the slice).
//   No barrier anywhere: a wave synchronises only with itself (ds_write -> ds_read order).
//
// Pipeline per 8-channel chunk (64 MFMAs): the reads + transform of the NEXT chunk, the B loads
// 8 positions ahead and half a stage of staging loads + LDS writes (for the stage after next) are
// slotted behind the MFMAs of the current chunk.
#include "kfn_common.h"
#ifndef KFN_WINO2_DBG
#define KFN_WINO2_DBG 0
#endif
#include <type_traits>
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned OOBV = 0x80000000u;  // voffset that always fails the buffer range check
constexpr int BW = 8, BH = 4;           // tile block of a wave: 8 (x) by 4 (y) tiles = 32 MFMA rows
constexpr int RW = 2 * BW + 2;          // 18 region pixels per row
constexpr int RH = 2 * BH + 4;          // 12 region rows: 10, +2 when the block straddles two images
constexpr int KS = 16;                  // input channels per stage
constexpr int QPP = KS / 4;             // 16-byte quads per pixel and stage
// LDS image of a stage: [RH rows][QPP quads][RW = 18 px] x 16 bytes.  Staging: one load per region row
// for its first 16 pixels (lane = (px, quad)) plus two loads for the columns 16, 17 of all rows; every load
// is "wave-uniform row offset + per-lane column offset": two address registers in all.  Pixel-fastest
// order: the 16 lanes of a ds_read_b128 group read tiles 2 pixels = 32 bytes apart, i.e. 8 distinct
// 16-byte slots of the 256-byte bank window (2-way conflict; [px][quad] order would be 128 bytes apart
// = 8-way).  Every fragment address is one per-lane register + a compile-time immediate.
constexpr int QPITCH = RW * 16;         // 288 bytes between the quads of a row
constexpr int RPITCH = QPP * QPITCH;    // 1152 bytes between region rows
constexpr int NMAIN = RH;               // 12
constexpr int NLOAD = NMAIN + 2;        // 14 staging loads per lane and stage
constexpr int NHALF = NLOAD / 2;        // staged in two halves of 7 (28 staging registers)
constexpr int BUF_BYTES = 16384;        // >= RH * RPITCH, a power of two: the other buffer is address ^ BUF_BYTES
constexpr int NB = 8;                   // depth of the B register ring (transform positions ahead); must divide 16

static_assert(BUF_BYTES >= RH * RPITCH + 32 * 16 && (BUF_BYTES & (BUF_BYTES - 1)) == 0, "buffer toggle by XOR");
static_assert(2 * BUF_BYTES == kfn::WINO2_LDS_BYTES, "kfn_winograd_lds_bytes() quotes this size");
static_assert(16 % NB == 0, "the ring slot of a position is (position % 16) % NB");

struct Wino2Args {
  const float* x;
  const float* u2;    // [Cin/8][16][cout_pad][8]
  const float* bias;
  float* y;
  int N, H, W, Cin, ldx;
  int Cout, cout_pad, ldy;
  int Th, Tw;         // tiles per image
  int vrows;          // N * Th: tile rows of the whole batch, enumerated image after image
  int bw;             // ceil(Tw / BW) column blocks
  int tiles_m, tiles_n;
  int relu;
  int wide_store;     // Cout, ldy multiples of 4 and y 16-byte aligned: LDS-transposed 16-byte stores
  int dbg;            // timing experiments only (KFN_WINO2_DBG): 1 = every A row read from row 0, 2 = every B fragment = fragment 0
  unsigned long long x_bytes;
  unsigned long long y_bytes;
  unsigned u_bytes;
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

__device__ __forceinline__ int xcd_remap2(int b, int nwg) {
  // blocks are dispatched round-robin over the 8 XCDs: give every XCD a contiguous run of tiles
  int xcd = b & 7;
  int q = nwg >> 3, r = nwg & 7;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (b >> 3);
}

// Every VALU instruction costs MFMA time here: v_mfma_f32_32x32x2_f32 runs at the fp32 VECTOR rate and a
// co-issued VALU op from the same wave is not hidden behind it (tools/mb/mfma_fill.hip: 64 cycles per MFMA
// alone, +14 for one v_add_f32 between two MFMAs, +4.5 per further one; ~5.3 each when issued as one burst;
// v_pk_add_f32 ~6.5 for two floats).  The transform is therefore written with packed adds (the compiler
// only packs the additions, hence the inline asm for a - b) and issued as ONE burst per chunk.
typedef float f32x2 __attribute__((ext_vector_type(2)));
// a -= b ; a = b - a ; r = a + b on register pairs (in place: no copies, no extra registers)
__device__ __forceinline__ void pk_sub_ip(f32x2& a, const f32x2& b) {
  asm("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(a) : "v"(b));
}
__device__ __forceinline__ void pk_rsub_ip(f32x2& a, const f32x2& b) {   // a = b - a
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

// One 1-D pass of B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]] on four register pairs, in place:
// (d0,d1,d2,d3) -> (d0-d2, d1+d2, d2-d1, d1-d3)
__device__ __forceinline__ void bt_pass(f32x2& d0, f32x2& d1, f32x2& d2, f32x2& d3) {
  pk_sub_ip(d0, d2);               // d0 = d0 - d2
  pk_rsub_ip(d3, d1);              // d3 = d1 - d3
  const f32x2 s = pk_add(d1, d2);  // d1 + d2
  pk_sub_ip(d2, d1);               // d2 = d2 - d1
  d1 = s;
}
// B^T d B in place on the 16 raw quads of a tile's 4x4 patch, held as 32 register pairs
// v[2*(4*r + c) + half]: first along the columns of every patch row (index nu), then along the rows
// (index xi); v[2*(4*xi + nu) + half] on return.
__device__ __forceinline__ void bt_d_b(f32x2 (&v)[32]) {
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int h = 0; h < 2; ++h) bt_pass(v[2 * (4 * r + 0) + h], v[2 * (4 * r + 1) + h], v[2 * (4 * r + 2) + h], v[2 * (4 * r + 3) + h]);
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int h = 0; h < 2; ++h) bt_pass(v[2 * (0 + c) + h], v[2 * (4 + c) + h], v[2 * (8 + c) + h], v[2 * (12 + c) + h]);
}

template <bool WIDE>   // WIDE: LDS-transposed 16-byte output stores (Cout, ldy multiples of 4, y 16-byte aligned)
__global__ __launch_bounds__(64, 1) void wino2_kernel(Wino2Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem2[];   // [2][BUF_BYTES] raw patches

  const int lane = threadIdx.x;
  const int nwg = p.tiles_m * p.tiles_n;
  const int tile = xcd_remap2(blockIdx.x, nwg);
  const int tm = tile % p.tiles_m;          // M fastest: concurrent workgroups of an XCD share the U slice
  const int tn = tile / p.tiles_m;
  const int cb = tm % p.bw, rb = tm / p.bw;
  const int n0 = tn * 32;

  // ---- block geometry (wave-uniform) ---------------------------------------------------
  // tile rows vr0 .. vr0+3 of the batch; rows >= brk (if any) belong to the next image
  const int vr0 = rb * BH;
  const int img0 = vr0 / p.Th;
  const int ty0 = vr0 - img0 * p.Th;
  const int brk = (p.Th - ty0 < BH) ? (p.Th - ty0) : BH;
  const int first_rows = 2 * brk + 2;       // region rows that show image img0

  const unsigned long long a_base = (unsigned long long)img0 * p.H * p.W * p.ldx * 4ull;
  const unsigned long long a_rest = p.x_bytes - a_base;
  char* const a_ptr = const_cast<char*>(reinterpret_cast<const char*>(p.x)) + a_base;
  const int a_records = (int)(a_rest < 0x7fffffffull ? a_rest : 0x7fffffffull);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(a_ptr, 0, a_records, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsU =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.u2), 0, p.u_bytes, 0x00020000);

  // ---- staging addresses ---------------------------------------------------------------------
  // main load k = region row k: lane = (px = lane >> 2, quad = lane & 3), address = per-lane column
  // offset (VGPR, out-of-image columns baked in as OOB) + wave-uniform row/channel offset (SGPR).
  // A region row that lies outside its image (zero padding) or is unused is read through a descriptor
  // whose num_records is 0 (the hardware then returns 0): one s_cselect on a descriptor word -- scalar
  // work, free beside the MFMAs, where a v_cndmask per load is not.
  // extra load j: lane -> (row = 8*j + lane / 8, px = 16 + (lane >> 2 & 1), quad = lane & 3).
  const int row_stride = p.W * p.ldx * 4;                               // bytes per image row
  const int base_first = (2 * ty0 - 1) * row_stride;                    // region row 0 (image img0)
  const int base_second = (p.H - 1 - first_rows) * row_stride;          // + q*row_stride for rows behind the seam
  auto row_ok = [&](int q) __attribute__((always_inline)) {
    const bool first = q < first_rows;
    const int yy = first ? 2 * ty0 - 1 + q : q - first_rows - 1;
    const int imo = first ? 0 : 1;
    return (q < RH) && (brk < BH || q < 2 * BH + 2) && (img0 + imo < p.N) && ((unsigned)yy < (unsigned)p.H);
  };
  unsigned rowmask = 0;
#pragma unroll
  for (int q = 0; q < RH; ++q) rowmask |= row_ok(q) ? (1u << q) : 0u;
  const int x_main = 2 * BW * cb - 1 + (lane >> 2);
  const unsigned voff_main = ((unsigned)x_main < (unsigned)p.W) ? (unsigned)((x_main * p.ldx + (lane & 3) * 4) * 4) : OOBV;
  unsigned voff_extra[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int q = 8 * j + (lane >> 3);     // 8 quads (2 px x 4) per row, 8 rows per load
    const int x = 2 * BW * cb - 1 + 16 + ((lane >> 2) & 1);
    const bool first = q < first_rows;
    const int yy = first ? 2 * ty0 - 1 + q : q - first_rows - 1;
    voff_extra[j] = (row_ok(q) && (unsigned)x < (unsigned)p.W)
                        ? (unsigned)((((first ? 0 : p.H) + yy) * p.W + x) * p.ldx * 4 + (lane & 3) * 16) : OOBV;
  }
  // LDS position of this lane's quad: main load k -> row k, extra load j -> row 8*j + lane/8, px 16 + ...
  const int wr_main = (lane & 3) * QPITCH + (lane >> 2) * 16;
  const int wr_extra = (lane >> 3) * RPITCH + (lane & 3) * QPITCH + (16 + ((lane >> 2) & 1)) * 16;
  // second extra load: rows 8..11 in lanes 0..31; lanes 32..63 (rows 12..15 do not exist, their loads
  // return 0) write into the unused tail of the buffer instead
  const int wr_extra1 = lane < 32 ? wr_extra + 8 * RPITCH : RH * RPITCH + (lane - 32) * 16;

  // ---- fragment addressing ----------------------------------------------------------------
  const int li = lane & 31, lh = lane >> 5;
  const int tr = li >> 3, tc = li & 7;
  // patch pixel (r, c) of tile (tr, tc), k-quad 2*chunk + lh: region row 2*tr + r (+2 behind the image
  // seam), column 2*tc + c  ->  rd_lo + r*RPITCH + c*16 + chunk*2*QPITCH
  const int rq0 = 2 * tr + (tr >= brk ? 2 : 0);
  const int rd_lo = rq0 * RPITCH + lh * QPITCH + 2 * tc * 16;
  const unsigned voff_b = (unsigned)(((n0 + li) * 8 + lh * 4) * 4);
  const unsigned b_step = (unsigned)p.cout_pad * 32u;    // bytes between consecutive (k-chunk, position) slices
  const int n_stages = p.Cin / KS;
  const int q_last = n_stages * 2 * 16 - 1;

  // The bias rides in the accumulator of position (xi,nu) = (1,1): A^T[a][1] = A[1][b] = 1 for all four
  // outputs of a tile, so Y = A^T M A receives it exactly once per output -- no epilogue add.
  const int n = n0 + li;
  const bool n_ok = n < p.Cout;
  const float bv = (p.bias != nullptr && n_ok) ? p.bias[n] : 0.f;
  f32x16 acc[16];
#pragma unroll
  for (int g = 0; g < 16; ++g)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[g][e] = (g == 5) ? bv : 0.f;

  f32x4 ra[NHALF];    // staging registers (half a stage of this lane's quads)
  f32x2 va[32], vb[32];   // V of the current / next chunk as register pairs (v[2*g + half])
  f32x4 bq[NB];

  // staging load k (0..13) of stage s into ra[k % 7].  Loads past the last stage re-read the last stage
  // (never used) instead of being masked: scalar clamp, no VALU.
  auto a_load = [&](auto kc, int s) __attribute__((always_inline)) {
    constexpr int k = decltype(kc)::value;
    const int sc = s < n_stages ? s : n_stages - 1;
    if constexpr (k < NMAIN) {
      const bool ok = (rowmask >> k) & 1u;
      const int soff = (p.dbg & 1) ? sc * (KS * 4) : (k < first_rows ? base_first : base_second) + k * row_stride + sc * (KS * 4);
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a_ptr, 0, ok ? a_records : 0, 0x00020000);
      ra[k % NHALF] = bload(rs, voff_main, ok ? (unsigned)soff : 0u);
    } else {
      ra[k % NHALF] = bload(rsA, voff_extra[k - NMAIN], (unsigned)(sc * (KS * 4)));
    }
  };
  // LDS addresses are "per-lane register + immediate".  The registers exist twice, for the buffer of the
  // current stage (index 0) and for the other one (index 1), and are toggled (^ BUF_BYTES) once per stage:
  // 8 VALU per 128 MFMAs instead of one address add per access.
  int wr_main_b[2] = {wr_main, wr_main ^ BUF_BYTES}, wr_extra_b[2] = {wr_extra, wr_extra ^ BUF_BYTES};
  int wr_extra1_b[2] = {wr_extra1, wr_extra1 ^ BUF_BYTES};
  int rd_lo_b[2] = {rd_lo, rd_lo ^ BUF_BYTES};
  auto a_write = [&](auto kc, auto buf_c) __attribute__((always_inline)) {
    constexpr int k = decltype(kc)::value;
    constexpr int b = decltype(buf_c)::value;   // 0 = buffer of the current stage, 1 = the other one
    if constexpr (k < NMAIN) *reinterpret_cast<f32x4*>(smem2 + wr_main_b[b] + k * RPITCH) = ra[k % NHALF];
    else if constexpr (k == NMAIN) *reinterpret_cast<f32x4*>(smem2 + wr_extra_b[b]) = ra[k % NHALF];
    else *reinterpret_cast<f32x4*>(smem2 + wr_extra1_b[b]) = ra[k % NHALF];
  };
  auto b_load = [&](auto gc, int qidx) __attribute__((always_inline)) {
    constexpr int g = decltype(gc)::value;
    const int qc = qidx < q_last ? qidx : q_last;     // past the end: re-read the last slice (never used)
    bq[g % NB] = bload(rsU, voff_b, (p.dbg & 2) ? 0u : (unsigned)qc * b_step);
  };
  // patch read (r, c) of chunk `chunk` from buffer `buf` into v[4*r + c]
  auto v_read = [&](auto ic, f32x2 (&v)[32], auto buf_c, auto chunk_c) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value, r = i >> 2, c = i & 3;
    constexpr int b = decltype(buf_c)::value, chunk = decltype(chunk_c)::value;
    const f32x4 q = *reinterpret_cast<const f32x4*>(smem2 + rd_lo_b[b] + (r * RPITCH + c * 16 + chunk * 2 * QPITCH));
    v[2 * i] = q.xy;
    v[2 * i + 1] = q.zw;
  };

  // ---- prologue ------------------------------------------------------------------------------
  // stage 0 -> buffer 0 (both halves), first half of stage 1 -> buffer 1, B ring, V of (stage 0, chunk 0).
  // All 21 staging loads are issued at once -- the V registers are free here and serve as landing
  // space -- so the prologue costs ONE memory round trip.
  {
    f32x4 l0[NLOAD], l1[NHALF];   // landing space (the V registers are not live yet)
    auto land = [&](auto kc, int s, f32x4& dst) __attribute__((always_inline)) {
      constexpr int k = decltype(kc)::value;
      a_load(kc, s);
      dst = ra[k % NHALF];
    };
    auto put = [&](auto kc, auto buf_c, const f32x4& src) __attribute__((always_inline)) {
      constexpr int k = decltype(kc)::value;
      ra[k % NHALF] = src;
      a_write(kc, buf_c);
    };
    sfor<NLOAD>([&](auto kc) { land(kc, 0, l0[decltype(kc)::value]); });
    sfor<NHALF>([&](auto kc) { land(kc, 1, l1[decltype(kc)::value]); });
    sfor<NB>([&](auto gc) { b_load(gc, decltype(gc)::value); });
    sfor<NLOAD>([&](auto kc) { put(kc, std::integral_constant<int, 0>{}, l0[decltype(kc)::value]); });
    sfor<NHALF>([&](auto kc) { put(kc, std::integral_constant<int, 1>{}, l1[decltype(kc)::value]); });
  }
  sfor<16>([&](auto ic) { v_read(ic, va, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}); });
  bt_d_b(va);

  // One chunk = 16 positions x 4 MFMAs (slot j).  Behind the MFMAs: the B load NB positions ahead (after
  // the last MFMA of a position), the 16 patch reads of the NEXT chunk (one per slot 8..23, free), 7 staging
  // loads (slots 0..6) and their LDS writes (slots 56..62), and behind slot 47 the transform of the next
  // chunk as ONE burst of 64 packed adds.
  auto chunk = [&](auto first_c, f32x2 (&vcur)[32], f32x2 (&vnext)[32], int s) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(first_c)::value;   // chunk 0 of the stage
    using cur_t = std::integral_constant<int, 0>;      // address set of this stage's LDS buffer
    using nxt_t = std::integral_constant<int, 1>;
    const int qbase = (s * 2 + (FIRST ? 0 : 1)) * 16;
    sfor<64>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      // four positions are interleaved (k-step t outer, position inner): consecutive MFMAs never share an accumulator
      constexpr int g = (j >> 4) * 4 + (j & 3), t = (j >> 2) & 3;
      acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(vcur[2 * g + (t >> 1)][t & 1], bq[g % NB][t], acc[g], 0, 0, 0);
      if constexpr (t == 3) b_load(std::integral_constant<int, g>{}, qbase + g + NB);
      // chunk 0 prefetches chunk 1 of the same stage, chunk 1 prefetches chunk 0 of the next stage
      constexpr int RD0 = 8;
      if constexpr (j >= RD0 && j < RD0 + 16) {
        if constexpr (FIRST) v_read(std::integral_constant<int, j - RD0>{}, vnext, cur_t{}, std::integral_constant<int, 1>{});
        else v_read(std::integral_constant<int, j - RD0>{}, vnext, nxt_t{}, std::integral_constant<int, 0>{});
      }
      if constexpr (j == 47) bt_d_b(vnext);
      // staging: chunk 0 of stage s carries the SECOND half of stage s+1 (-> other buffer), chunk 1 the
      // FIRST half of stage s+2 (-> this stage's buffer, whose reads are over)
      constexpr int LD0 = 0, WR0 = 56;
      if constexpr (j >= LD0 && j < LD0 + NHALF) {
        constexpr int k = (FIRST ? NHALF : 0) + (j - LD0);
        a_load(std::integral_constant<int, k>{}, FIRST ? s + 1 : s + 2);
      }
      if constexpr (j >= WR0 && j < WR0 + NHALF) {
        constexpr int k = (FIRST ? NHALF : 0) + (j - WR0);
        if constexpr (FIRST) a_write(std::integral_constant<int, k>{}, nxt_t{});
        else a_write(std::integral_constant<int, k>{}, cur_t{});
      }
      // pin the slot: the compiler's own list scheduler otherwise sinks every load next to its use
      // (register pressure) and the prefetch distances collapse into vmcnt(0) stalls
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  for (int s = 0; s < n_stages; ++s) {
    chunk(std::true_type{}, va, vb, s);
    chunk(std::false_type{}, vb, va, s);
    // the other buffer becomes the current one
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      wr_main_b[b] ^= BUF_BYTES; wr_extra_b[b] ^= BUF_BYTES; wr_extra1_b[b] ^= BUF_BYTES; rd_lo_b[b] ^= BUF_BYTES;
    }
  }

  // ---- epilogue: Y = A^T M A per (tile, channel), ReLU, store ----------------------------------
  // C/D layout of the 32x32 MFMA: col = lane & 31 (channel), row = (e & 3) + 8*(e >> 2) + 4*(lane >> 5)
  // = tile (e >> 2, (e & 3) + 4*lh) of the 4 x 8 block, i.e. the tile ROW is uniform per e.  Element
  // pairs (e, e+1) are transformed together with packed adds; stores are buffer stores: per-lane column
  // offset + wave-uniform row offset, 32 channels of a pixel = one 128-byte run.
  const bool relu = p.relu != 0;
  const unsigned long long y_base = (unsigned long long)img0 * p.H * p.W * p.ldy * 4ull;
  const unsigned long long y_rest = p.y_bytes - y_base;
  const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<char*>(p.y) + y_base, 0, (int)(y_rest < 0x7fffffffull ? y_rest : 0x7fffffffull), 0x00020000);
  const int txl = cb * BW + 4 * lh;                    // tile column of this lane for e & 3 == 0
  const unsigned voff_y = (unsigned)((2 * txl * p.ldy + n) * 4);
  const int pix_bytes = p.ldy * 4;
  // fast path: the whole 8x4 block lies inside the tensor (no per-lane masks)
  const bool full = (vr0 + BH <= p.vrows) && (cb * BW + BW <= p.Tw) && (p.W == 2 * p.Tw) && (p.H == 2 * p.Th) &&
                    (n0 + 32 <= p.Cout);
  auto emit = [&](float v, int trow, int ec, int a, int b) __attribute__((always_inline)) {
    const int img_rel = trow < brk ? 0 : 1;
    const int ty = trow < brk ? ty0 + trow : trow - brk;
    const int oy = 2 * ty + a;
    v = relu ? fmaxf(v, 0.f) : v;
    const unsigned soff = (unsigned)(((img_rel * p.H + oy) * p.W + 2 * ec + b) * pix_bytes);
    if (full) {
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsY, voff_y, soff, 0);
    } else if (vr0 + trow < p.vrows && oy < p.H) {
      const int tx = txl + ec;
      const bool ok = n_ok && tx < p.Tw && 2 * tx + b < p.W;
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsY, ok ? voff_y : OOBV, soff, 0);
    }
  };
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
  if constexpr (WIDE) {
    // Global stores are ISSUE bound (~60 cycles per store instruction per CU whatever its width): 128 dword stores
    // per wave cost more than a third of a 64-channel layer's whole K loop.  As in kfn_wino3.hip the wave
    // transposes its 128 pixels x 32 channels through 16 KiB of the (now idle) staging buffers -- [pixel][32
    // channels], written by ds_write_b32, read back as 1 KiB runs -- and issues 16 stores of 16 bytes per lane: 8
    // lanes cover the 128 contiguous bytes of a pixel.  (LDS operations of one wave complete in order: the stray
    // prefetch reads of the last chunk cannot be overtaken by these writes.)
    char* const stg = smem2;
    const int st_w = lh * 1024 + li * 4;   // block pixel (oy, ox) = (2 trow + a, 8 lh + 2 ec + b) -> row oy*16 + ox
    out_transform([&](float v, int trow, int ec, int a, int b) __attribute__((always_inline)) {
      *reinterpret_cast<float*>(stg + st_w + ((2 * trow + a) * 16 + 2 * ec + b) * 128) = v;
    });
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
      const bool row_in = vr0 + trow < p.vrows && oy < p.H;        // uniform
      const unsigned soff = (unsigned)(((img_rel * p.H + oy) * p.W + ox0 + 8 * hx) * pix_bytes);
      kfn::buffer_store_b128<KFN_NT_STORE_AUX>(v, rsY, row_in ? voff_h[hx] : OOBV, soff);
    }
    return;
  } else {
    out_transform(emit);
  }
}

}  // namespace

namespace kfn {
int launch_wino3(const kfn_conv_desc* d, const float* x, const void* u2_packed, const float* bias, float* y,
                 hipStream_t stream);   // kfn_wino3.hip: four waves share one input transform (fp32 or fp16 operands)
}

// Can the single-kernel path take this layer?  (host-side routing; no device access)
extern "C" int kfn_winograd_fused_supported(const kfn_conv_desc* d) {
  kfn_conv_desc d_full;
  if (!d || kfn::conv_desc_in(d, &d_full, "kfn_winograd_fused_supported") != KFN_OK) return 0;
  d = &d_full;
  if (d->x_dtype != KFN_ACT_F32 || d->y_dtype != KFN_ACT_F32) return 0;
  if (d->kh != 3 || d->kw != 3 || d->stride != 1 || d->transposed) return 0;
  if (d->Cin <= 0 || d->Cin % KS != 0) return 0;
  if ((d->H + 1) / 2 < BH) return 0;   // a 4-row tile block may straddle at most two images
  if (d->epilogue != KFN_EPI_NONE) return 0;
  if (d->cout_pad % 32 != 0) return 0;
  // fp16 operands (BASELINE config 5): the four-wave form only
  if (d->operand_dtype == KFN_OPERAND_F16) return d->Cout >= 128 && d->Cin % 64 == 0;   // two super-steps per loop trip
  if (d->operand_dtype != KFN_OPERAND_F32) return 0;
  return 1;
}

extern "C" int kfn_conv2d_winograd_fused(const kfn_conv_desc* d, const float* x, const void* u2_packed,
                                         const float* bias, float* y, void* stream) {
  KFN_REQUIRE(d && x && u2_packed && y, "kfn_conv2d_winograd_fused: null argument");
  KFN_CONV_DESC_IN(d, "kfn_conv2d_winograd_fused");
  KFN_REQUIRE(d->x_dtype == KFN_ACT_F32 && d->y_dtype == KFN_ACT_F32,
              "kfn_conv2d_winograd_fused: fp32 activations in memory only (x_dtype / y_dtype = KFN_ACT_F16 is implemented by kfn_conv2d_nhwc)");
  KFN_REQUIRE(d->kh == 3 && d->kw == 3 && d->stride == 1 && !d->transposed,
              "kfn_conv2d_winograd_fused: only 3x3 stride-1 SAME convolutions");
  KFN_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0, "kfn_conv2d_winograd_fused: bad shape %dx%dx%d", d->N, d->H, d->W);
  if (d->Cin <= 0 || d->Cin % KS != 0)
    return kfn::fail(KFN_ERR_UNSUPPORTED, "kfn_conv2d_winograd_fused: Cin=%d must be a multiple of %d", d->Cin, KS);
  if ((d->H + 1) / 2 < BH)
    return kfn::fail(KFN_ERR_UNSUPPORTED, "kfn_conv2d_winograd_fused: H=%d is below %d rows", d->H, 2 * BH - 1);
  KFN_REQUIRE(d->ldx >= d->Cin && d->ldx % 4 == 0 && d->Cout > 0 && d->ldy >= d->Cout &&
                  d->cout_pad >= d->Cout && d->cout_pad % 32 == 0,
              "kfn_conv2d_winograd_fused: bad strides / channel counts");
  KFN_REQUIRE(d->epilogue == KFN_EPI_NONE && (d->operand_dtype == KFN_OPERAND_F32 || d->operand_dtype == KFN_OPERAND_F16),
              "kfn_conv2d_winograd_fused: fp32 or fp16 operands, no fused head epilogue");
  const bool h16 = d->operand_dtype == KFN_OPERAND_F16;
  if (h16 && !(d->Cout >= 128 && d->Cin % 64 == 0))
    return kfn::fail(KFN_ERR_UNSUPPORTED, "kfn_conv2d_winograd_fused: fp16 operands need Cout >= 128 and Cin %% 64 == 0");
  KFN_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(u2_packed)) & 15) == 0,
              "kfn_conv2d_winograd_fused: buffers must be 16-byte aligned");
  {
    // >= 128 output channels and whole 32-channel super-steps: the 4-wave form (one transform per 128 output
    // channels, input read Cout/128 times); kfn_conv_desc.wino_form = KFN_WINO_FORM_ONE_WAVE forces the one-wave form for A/B measurements
    const int form = d->wino_form == KFN_WINO_FORM_ONE_WAVE ? 2 : 0;
    const long img_b = (long)d->H * d->W * d->ldx * 4L;
    if ((form != 2 || h16) && d->Cout >= 128 && d->Cin % 32 == 0 && 2 * img_b < (1L << 31) &&
        2L * d->H * d->W * d->ldy * 4L < (1L << 31) && 16L * d->cout_pad * d->Cin * 4L < (1L << 31))
      return kfn::launch_wino3(d, x, u2_packed, bias, y, (hipStream_t)stream);
    // 33 .. 64 output channels (conv1b): the two-wave form of the same kernel -- one transform and one read of the
    // input for all 64 channels, two workgroups per CU
    if (form != 2 && !h16 && d->cout_pad == 64 && d->Cin % 16 == 0 && 2 * img_b < (1L << 30) &&
        2L * d->H * d->W * d->ldy * 4L < (1L << 31) && 16L * d->cout_pad * d->Cin * 4L < (1L << 31))
      return kfn::launch_wino3(d, x, u2_packed, bias, y, (hipStream_t)stream);
    if (h16) return kfn::fail(KFN_ERR_UNSUPPORTED, "kfn_conv2d_winograd_fused: image or kernel beyond 2 GiB of 32-bit offsets");
  }
  Wino2Args a;
  a.x = x; a.u2 = static_cast<const float*>(u2_packed); a.bias = bias; a.y = y;
  a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.ldx = d->ldx;
  a.Cout = d->Cout; a.cout_pad = d->cout_pad; a.ldy = d->ldy;
  a.Th = (d->H + 1) / 2; a.Tw = (d->W + 1) / 2;
  const long vrows = (long)d->N * a.Th;
  const long in_pix = (long)d->N * d->H * d->W;
  const long x_bytes = ((in_pix - 1) * d->ldx + d->Cin) * 4L;
  const long u_bytes = 16L * d->cout_pad * d->Cin * 4L;
  const long img_bytes = (long)d->H * d->W * d->ldx * 4L;
  KFN_REQUIRE(vrows < (1L << 30) && u_bytes < (1L << 31) && 2 * img_bytes < (1L << 31) &&
                  2L * d->H * d->W * d->ldy * 4L < (1L << 31),
              "kfn_conv2d_winograd_fused: tensor too large for 32-bit buffer addressing");
  a.vrows = (int)vrows;
  a.bw = kfn::ceil_div(a.Tw, BW);
  const long tiles_m = (long)a.bw * kfn::ceil_div(a.vrows, BH);
  a.tiles_n = d->cout_pad / 32;
  KFN_REQUIRE(tiles_m * a.tiles_n < (1L << 31), "kfn_conv2d_winograd_fused: grid too large");
  a.tiles_m = (int)tiles_m;
  a.relu = d->relu;
  a.wide_store = (d->Cout % 4 == 0 && d->ldy % 4 == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0) ? 1 : 0;
#ifdef KFN_WINO2_NO_WIDE   // A/B builds only (tools/mb/build_hot.sh): the dword-store epilogue on aligned outputs too
  a.wide_store = 0;
#endif
  a.dbg = KFN_WINO2_DBG;   // build-time timing hooks (-DKFN_WINO2_DBG=1: hot A, 2: hot B); 0 in the product build
  a.x_bytes = (unsigned long long)x_bytes;
  a.y_bytes = (unsigned long long)(((in_pix - 1) * d->ldy + d->Cout) * 4L);
  a.u_bytes = (unsigned)u_bytes;
  if (a.wide_store)
    hipLaunchKernelGGL(wino2_kernel<true>, dim3((unsigned)(a.tiles_m * a.tiles_n)), dim3(64), 2 * BUF_BYTES,
                       (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(wino2_kernel<false>, dim3((unsigned)(a.tiles_m * a.tiles_n)), dim3(64), 2 * BUF_BYTES,
                       (hipStream_t)stream, a);
  KFN_LAUNCH_CHECK("wino2_kernel");
  return KFN_OK;
}
