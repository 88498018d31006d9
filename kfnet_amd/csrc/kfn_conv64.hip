// kfn_conv64.hip -- 3x3 stride-1 SAME convolution 64 -> 64 channels on fp16 activations (BASELINE config 5: SCoordNet's
// conv1b at full resolution, cnn_wrapper/SCoordNet.py:20 through tf.layers.conv2d, cnn_wrapper/network.py:116-135).
//
// Why its own kernel.  With 64 output channels the implicit-GEMM tile of kfn_conv.hip reuses an activation fragment for
// two MFMAs only: both operands come through the LDS at one KB per v_mfma_f32_32x32x16_f16, the K loop is 36 steps long,
// and the layer sat at 0.23 of the fp16 MFMA peak (`SQ_WAIT_INST_LDS` 170 M cycles, round 3).  Here
//   * the WEIGHTS never touch the LDS: 9 taps x 64 x 32 halfs = 36 A fragments = 144 VGPRs per wave, loaded once per
//     workgroup (wave = (channel half hc, pixel half hp) of a 64-channel x SW-pixel row strip);
//   * an INPUT ROW is read from the LDS once for the THREE output rows it feeds: the accumulators of rows r+1, r, r-1
//     (3 x PXB x 16 registers, the whole accumulation file) take the products with the kernel rows dy = 0, 1, 2 of the
//     same B fragment back to back -- 12 PXB fragment reads for 36 PXB MFMAs per row and wave;
//   * the workgroup walks down its strip: per step one new input row (SW + 2 pixels x 128 B, zero outside the image by the
//     buffer range check) goes global -> registers -> LDS under the MFMAs of the current one, one finished output row
//     leaves through an LDS transpose as whole 128-byte pixel lines.  One s_barrier per row.
// MFMA roles: D[channel][pixel] = A[channel][k] B[k][pixel] -- a lane of D holds 4 consecutive channels of one pixel four
// times over, which the transpose writes as 8-byte pieces of the NHWC line.
//
// LDS: input rows [2][(SW + 2) px][144 B], output tiles [2][SW px][144 B]; the 16-byte pad per pixel puts the 16 lanes of
// every ds_read_b128 group on 16 different slots (9 px mod 16 is a bijection).  SW = 64 PXB, PXB = 2 | 3 (960 = 5 x 192).
#include "kfn_common.h"
#include <type_traits>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int PS = 144;                  // LDS bytes per pixel
constexpr unsigned OOB = 0x80000000u;    // a voffset that fails every range check: the load returns 0, the store is dropped

struct C64Args {
  const char* x;        // fp16 [N][H][W][ldx]
  char* y;              // fp16 [N][H][W][ldy]
  const f16x8* w;       // [2 channel halves][36 fragments][64 lanes] x 8 halfs (graph.pack_conv64_rows_kernel)
  const float* bias;    // [64] or null
  int N, H, W, ldx, ldy, relu;
  int rows_per_chunk, chunks, strips;
  unsigned x_img_bytes, y_img_bytes;
};

template <int I, int N, class F>
__device__ __forceinline__ void sfor64_impl(F& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    sfor64_impl<I + 1, N>(f);
  }
}
template <int N, class F>
__device__ __forceinline__ void sfor64(F&& f) {
  sfor64_impl<0, N>(f);
}

template <int PXB, bool RELU>
__global__ __launch_bounds__(256, 1) void conv64_rows_kernel(C64Args p) {
  constexpr int SW = 64 * PXB;
  constexpr int ROWB = (SW + 2) * PS;
  constexpr int OUTB = SW * PS;
  constexpr int IN_ITEMS = (SW + 2) * 8;                 // 16-byte pieces of an input row
  constexpr int NI = (IN_ITEMS + 255) / 256;
  constexpr int NO = SW * 8 / 256;                       // 16-byte pieces of an output row per thread
  constexpr int NG = 3 * PXB;                            // fragment groups per row: (pixel block, dx) x 4 channel chunks
  extern __shared__ __attribute__((aligned(16))) char smem64[];   // [2][ROWB] | [2][OUTB]
  char* const ring = smem64;
  char* const otile = smem64 + 2 * ROWB;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hc = wave & 1, hp = wave >> 1;
  const int j = lane & 31, h = lane >> 5;

  // work item -> (image, strip, chunk of rows); chunks of an image strip are neighbours (they share halo rows in the L2)
  const int chunk = blockIdx.x % p.chunks;
  const int t0 = blockIdx.x / p.chunks;
  const int strip = t0 % p.strips;
  const int n = t0 / p.strips;
  const int r0 = chunk * p.rows_per_chunk;
  const int r1 = (r0 + p.rows_per_chunk < p.H) ? r0 + p.rows_per_chunk : p.H;
  const int x0 = strip * SW;

  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(p.x) + (size_t)n * p.x_img_bytes, 0, (int)p.x_img_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsY =
      __builtin_amdgcn_make_buffer_rsrc(p.y + (size_t)n * p.y_img_bytes, 0, (int)p.y_img_bytes, 0x00020000);

  // ---- weights: 36 A fragments of this wave's channel half, for the whole workgroup's life -------------------------
  f16x8 wr[36];
#pragma unroll
  for (int f = 0; f < 36; ++f) wr[f] = p.w[(hc * 36 + f) * 64 + lane];

  // ---- this thread's pieces of an input row: piece = tid + 256 i -> (LDS pixel px = piece / 8, slot = piece % 8) ---------
  unsigned in_col[NI];       // byte offset inside an image row, or OOB (left / right of the image, past the row's pieces)
  int in_lds[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int piece = tid + 256 * i;
    const int px = piece >> 3, slot = piece & 7;
    const int xx = x0 - 1 + px;
    const bool ok = piece < IN_ITEMS && xx >= 0 && xx < p.W;
    in_col[i] = ok ? (unsigned)(xx * p.ldx * 2 + slot * 16) : OOB;
    in_lds[i] = px * PS + slot * 16;                              // (pieces past the row are not stored: row_store)
  }
  const unsigned row_bytes_x = (unsigned)(p.W * p.ldx * 2);
  u32x4 ld[NI];
  auto row_load = [&](int ri) __attribute__((always_inline)) {       // input row ri -> registers (zeros above / below the image)
    const bool row_ok = ri >= 0 && ri < p.H;                          // uniform
    const unsigned soff = row_ok ? (unsigned)ri * row_bytes_x : 0u;
#pragma unroll
    for (int i = 0; i < NI; ++i)
      ld[i] = __builtin_amdgcn_raw_buffer_load_b128(rsX, (row_ok && (tid + 256 * i) < IN_ITEMS) ? in_col[i] : OOB, soff, 0);
  };
  auto row_store = [&](int b) __attribute__((always_inline)) {       // registers -> ring[b]
#pragma unroll
    for (int i = 0; i < NI; ++i)
      if (NI * 256 == IN_ITEMS || tid + 256 * i < IN_ITEMS) *reinterpret_cast<u32x4*>(ring + b * ROWB + in_lds[i]) = ld[i];
  };

  // ---- consumer addresses ---------------------------------------------------------------------------------------------
  // B fragment (pixel block pb, tap column dx, channel chunk c): lane (j, h) reads 8 halfs of LDS pixel 32 (PXB hp + pb) + j + dx
  const int b_base = (32 * PXB * hp + j) * PS + h * 16;
  // D element e = 4 g + i -> out tile pixel 32 (PXB hp + pb) + j, channel 32 hc + 8 g + 4 h + i
  const int o_base = (32 * PXB * hp + j) * PS + (32 * hc + 4 * h) * 2;
  // output pieces: piece = tid + 256 i -> pixel piece / 8, slot piece % 8
  unsigned out_col[NO];
  int out_lds[NO];
#pragma unroll
  for (int i = 0; i < NO; ++i) {
    const int piece = tid + 256 * i;
    const int px = piece >> 3, slot = piece & 7;
    out_col[i] = (x0 + px < p.W) ? (unsigned)((x0 + px) * p.ldy * 2 + slot * 16) : OOB;
    out_lds[i] = px * PS + slot * 16;
  }
  const unsigned row_bytes_y = (unsigned)(p.W * p.ldy * 2);

  f32x16 acc[3][PXB];
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int pb = 0; pb < PXB; ++pb)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[s][pb][e] = 0.f;
  // an output row's accumulators START from the bias (element e = 4 g + i is channel 32 hc + 8 g + 4 h + i): no bias adds later
  f32x16 bias_c;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f32x4 b4 = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + 32 * hc + 8 * g + 4 * h) : f32x4{0.f, 0.f, 0.f, 0.f};
    bias_c[4 * g] = b4.x; bias_c[4 * g + 1] = b4.y; bias_c[4 * g + 2] = b4.z; bias_c[4 * g + 3] = b4.w;
  }

  // ---- prologue: input row r0 - 1 ---------------------------------------------------------------------------------------
  row_load(r0 - 1);
  row_store(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  int cur = 0;
  // One step = input row ri, in ring[cur].  Accumulator sets by phase ph = (ri - (r0 - 1)) % 3: `fresh` = output row ri + 1
  // (kernel row 0, starts from the bias), `mid` = row ri (kernel row 1), `done` = row ri - 1 (kernel row 2: pixel block pb of it
  // is complete after fragment group 3 pb + 2).  One wave per SIMD: everything that is not an MFMA is placed into the MFMA
  // stream a few instructions per gap (a v_mfma_f32_32x32x16_f16 covers ~5 issue slots):
  //   groups 0 / 1      the output row finished in the PREVIOUS step leaves: out tile [cur ^ 1] -> registers -> global
  //   groups 3pb+3, +4  the done row's pixel block pb: ReLU, one RNE rounding, 8-byte pieces into out tile [cur]
  //   tail              the last pixel block's pieces, the next input row registers -> ring[cur ^ 1], the step's barrier
  // No control flow inside: rows that must not be stored get an out-of-range store offset, loads past the chunk are harmless.
  u32x4 ov[NO];
  auto tile_piece = [&](auto donec, auto pbc, auto gc) __attribute__((always_inline)) {
    constexpr int DONE = decltype(donec)::value, pb = decltype(pbc)::value, g = decltype(gc)::value;
    f16x4 hv = {(_Float16)acc[DONE][pb][4 * g], (_Float16)acc[DONE][pb][4 * g + 1], (_Float16)acc[DONE][pb][4 * g + 2],
                (_Float16)acc[DONE][pb][4 * g + 3]};   // RNE; max(round(v), 0) == round(max(v, 0))
    if constexpr (RELU) hv = __builtin_elementwise_max(hv, f16x4{(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f});
    *reinterpret_cast<f16x4*>(otile + cur * OUTB + o_base + 32 * pb * PS + 16 * g) = hv;
  };
  auto step = [&](auto phc, int ri) __attribute__((always_inline)) {
    constexpr int ph = decltype(phc)::value;
    constexpr int FRESH = ph, MID = (ph + 2) % 3, DONE = (ph + 1) % 3;
    row_load(ri + 1);                               // the next input row, under this row's MFMAs
    const char* const rb = ring + cur * ROWB;
    const char* const prev_tile = otile + (cur ^ 1) * OUTB;
    const bool prev_emit = ri - 2 >= r0;            // the row finished in the previous step (ri - 2) belongs to this chunk
    const unsigned prev_soff = prev_emit ? (unsigned)(ri - 2) * row_bytes_y : 0u;
    f16x8 bf[2][4];
    auto frag_reads = [&](auto gc, f16x8 (&dst)[4]) __attribute__((always_inline)) {
      constexpr int g = decltype(gc)::value;
      constexpr int pb = g / 3, dx = g % 3;
#pragma unroll
      for (int c = 0; c < 4; ++c)
        dst[c] = *reinterpret_cast<const f16x8*>(rb + b_base + (32 * pb + dx) * PS + c * 32);
    };
    frag_reads(std::integral_constant<int, 0>{}, bf[0]);
    __builtin_amdgcn_sched_barrier(0);
    sfor64<NG>([&](auto gc) {
      constexpr int g = decltype(gc)::value;
      constexpr int pb = g / 3, dx = g % 3;
      if constexpr (g + 1 < NG) frag_reads(std::integral_constant<int, g + 1>{}, bf[(g + 1) & 1]);
      if constexpr (g == 0) {
#pragma unroll
        for (int i = 0; i < NO; ++i) ov[i] = *reinterpret_cast<const u32x4*>(prev_tile + out_lds[i]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f16x8 b = bf[g & 1][c];
        if (dx == 0 && c == 0)
          acc[FRESH][pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[(0 * 3 + dx) * 4 + c], b, bias_c, 0, 0, 0);
        else
          acc[FRESH][pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[(0 * 3 + dx) * 4 + c], b, acc[FRESH][pb], 0, 0, 0);
        acc[MID][pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[(1 * 3 + dx) * 4 + c], b, acc[MID][pb], 0, 0, 0);
        acc[DONE][pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[(2 * 3 + dx) * 4 + c], b, acc[DONE][pb], 0, 0, 0);
        // ---- the fillers of this gap ----
        if constexpr (g == 1) {                     // the previous row's pieces leave (NO = 2 PXB <= 8 stores over 4 gaps)
#pragma unroll
          for (int i = c; i < NO; i += 4)
            kfn::buffer_store_b128<KFN_NT_STORE_AUX>(ov[i], rsY, prev_emit ? out_col[i] : OOB, prev_soff);
        }
        if constexpr (g >= 3 && (g - 3) % 3 < 2) {  // pixel block (g - 3) / 3 of the done row: pieces 2 * ((g - 3) % 3) + c / 2 ...
          constexpr int epb = (g - 3) / 3;
          if (c % 2 == 0) {
            if constexpr ((g - 3) % 3 == 0) {
              if (c == 0) tile_piece(std::integral_constant<int, DONE>{}, std::integral_constant<int, epb>{}, std::integral_constant<int, 0>{});
              else tile_piece(std::integral_constant<int, DONE>{}, std::integral_constant<int, epb>{}, std::integral_constant<int, 1>{});
            } else {
              if (c == 0) tile_piece(std::integral_constant<int, DONE>{}, std::integral_constant<int, epb>{}, std::integral_constant<int, 2>{});
              else tile_piece(std::integral_constant<int, DONE>{}, std::integral_constant<int, epb>{}, std::integral_constant<int, 3>{});
            }
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    // tail: the last pixel block of the done row, the next input row into the ring, the barrier
    sfor64<4>([&](auto gc) {
      tile_piece(std::integral_constant<int, DONE>{}, std::integral_constant<int, PXB - 1>{}, gc);
    });
    row_store(cur ^ 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's LDS stores are done before the barrier releases the readers
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    cur ^= 1;
  };

  int last = r0 - 1;
  for (int ri = r0 - 1; ri <= r1; ri += 3) {
    step(std::integral_constant<int, 0>{}, ri);
    last = ri;
    if (ri + 1 <= r1) { step(std::integral_constant<int, 1>{}, ri + 1); last = ri + 1; }
    if (ri + 2 <= r1) { step(std::integral_constant<int, 2>{}, ri + 2); last = ri + 2; }
  }
  // the row finished by the last step (r1 - 1, in out tile [cur ^ 1] after the flip) leaves
  {
    const char* const prev_tile = otile + (cur ^ 1) * OUTB;
    const unsigned soff = (unsigned)(last - 1) * row_bytes_y;
#pragma unroll
    for (int i = 0; i < NO; ++i) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(prev_tile + out_lds[i]);
      kfn::buffer_store_b128<KFN_NT_STORE_AUX>(v, rsY, (last - 1 >= r0) ? out_col[i] : OOB, soff);
    }
  }
}

template <int PXB, bool RELU>
int launch64(const C64Args& a, int grid, hipStream_t stream) {
  constexpr int SW = 64 * PXB;
  constexpr int LDS = 2 * (SW + 2) * PS + 2 * SW * PS;
  static std::atomic<uint64_t> attr_done{0};
  int rc = kfn::set_max_dynamic_lds(reinterpret_cast<const void*>(conv64_rows_kernel<PXB, RELU>), LDS, attr_done);
  if (rc != KFN_OK) return rc;
  hipLaunchKernelGGL((conv64_rows_kernel<PXB, RELU>), dim3(grid), dim3(256), LDS, stream, a);
  KFN_LAUNCH_CHECK("conv64_rows_kernel");
  return KFN_OK;
}

// strip width: the one that pads the image width least (ties: the wider strip -- fewer halo columns)
int pick_pxb(int W) {
  const int w2 = kfn::ceil_div(W, 128) * 128, w3 = kfn::ceil_div(W, 192) * 192;
  return w3 <= w2 ? 3 : 2;
}

}  // namespace

// Can kfn_conv3x3_c64_f16 take this layer?  (host-side routing; no device access)
extern "C" int kfn_conv3x3_c64_f16_supported(const kfn_conv_desc* d) {
  kfn_conv_desc d_full;
  if (!d || kfn::conv_desc_in(d, &d_full, "kfn_conv3x3_c64_f16_supported") != KFN_OK) return 0;
  d = &d_full;
  if (d->kh != 3 || d->kw != 3 || d->stride != 1 || d->transposed || d->epilogue != KFN_EPI_NONE) return 0;
  if (d->x_dtype != KFN_ACT_F16 || d->y_dtype != KFN_ACT_F16 || d->operand_dtype != KFN_OPERAND_F16) return 0;
  if (d->Cin != 64 || d->Cout != 64) return 0;
  if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->ldx < 64 || d->ldy < 64 || d->ldx % 8 != 0 || d->ldy % 8 != 0) return 0;
  if ((long)d->H * d->W * d->ldx * 2L >= (1L << 31) || (long)d->H * d->W * d->ldy * 2L >= (1L << 31)) return 0;
  return 1;
}

extern "C" int kfn_conv3x3_c64_f16(const kfn_conv_desc* d, const void* x, const void* w_packed, const float* bias, void* y,
                                   void* stream) {
  KFN_REQUIRE(d && x && w_packed && y, "kfn_conv3x3_c64_f16: null argument");
  KFN_CONV_DESC_IN(d, "kfn_conv3x3_c64_f16");
  if (kfn_conv3x3_c64_f16_supported(d) != 1)
    return kfn::fail(KFN_ERR_UNSUPPORTED,
                     "kfn_conv3x3_c64_f16: only 3x3 stride-1 SAME, 64 -> 64 channels, fp16 operands and fp16 activations in and "
                     "out (x_dtype = y_dtype = KFN_ACT_F16), ldx / ldy multiples of 8, an image below 2 GiB, no fused epilogue");
  KFN_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(w_packed) |
                (bias ? reinterpret_cast<uintptr_t>(bias) : 0)) & 15) == 0,
              "kfn_conv3x3_c64_f16: x, y, w_packed and bias must be 16-byte aligned");
  C64Args a;
  a.x = static_cast<const char*>(x);
  a.y = static_cast<char*>(y);
  a.w = static_cast<const f16x8*>(w_packed);
  a.bias = bias;
  a.N = d->N; a.H = d->H; a.W = d->W; a.ldx = d->ldx; a.ldy = d->ldy; a.relu = d->relu;
  a.x_img_bytes = (unsigned)((long)d->H * d->W * d->ldx * 2L);
  a.y_img_bytes = (unsigned)((long)d->H * d->W * d->ldy * 2L);
  const int pxb = pick_pxb(d->W);
  a.strips = kfn::ceil_div(d->W, 64 * pxb);
  // Rows per workgroup.  A chunk of R output rows costs R + 2 row steps (two halo rows) plus the weight fetch and the
  // prologue row (~ one more step), and the 256 CUs hold one workgroup each: take the chunk count that maximises
  // [useful steps / paid steps] x [fill of the last round of workgroups].
  const long strips_total = (long)d->N * a.strips;
  int best_rows = d->H;
  double best_eff = -1.0;
  for (int c = 1; c <= d->H; ++c) {
    const int R = kfn::ceil_div(d->H, c);
    if (R < 4 && c > 1) break;
    const long g = strips_total * kfn::ceil_div(d->H, R);
    const double eff = ((double)R / (R + 3)) * ((double)g / (double)(((g + 255) / 256) * 256));
    if (eff > best_eff) { best_eff = eff; best_rows = R; }
  }
  a.rows_per_chunk = best_rows;
  a.chunks = kfn::ceil_div(d->H, a.rows_per_chunk);
  const long grid = strips_total * a.chunks;
  KFN_REQUIRE(grid < (1L << 31), "kfn_conv3x3_c64_f16: grid too large");
  hipStream_t st = (hipStream_t)stream;
  if (pxb == 3) return a.relu ? launch64<3, true>(a, (int)grid, st) : launch64<3, false>(a, (int)grid, st);
  return a.relu ? launch64<2, true>(a, (int)grid, st) : launch64<2, false>(a, (int)grid, st);
}
