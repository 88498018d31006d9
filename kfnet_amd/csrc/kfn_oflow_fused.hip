// kfn_oflow_fused.hip -- the two ends of OFlowNet that touch the 8x8 window grid, window-resident:
//
//   kfn_oflow_head   conv0 (by linearity: T_k[p] - G_k[p + cell - 4], ReLU; KFNet/KFNet.py:343-359,372 +
//                    cnn_wrapper/OFlowNet.py:19) -> conv1a (3x3, stride 2, 32 -> 32, ReLU; OFlowNet.py:20)
//   kfn_oflow_tail2  upconv0 (conv2d_transpose 3x3 stride 2, 32 -> 16, ReLU; OFlowNet.py:37) ++ conv0 (recomputed)
//                    -> conv6 (3x3, 48 -> 16, ReLU) -> prediction (3x3, 16 -> 1) -> softmax over the 64 cells ->
//                    soft-argmax flow (OFlowNet.py:38-47, KFNet/KFNet.py:381-385)
//
// Launch by launch (round 2) conv0's output [P,8,8,32] -- 1.26 GB per 32-frame batch -- was written by
// kfn_cost_volume_gather, read by conv1a and read again, inside concat0, by the tail; upconv0 ran on the generic
// kernel with half-empty 16-column tiles and scattered its 16 channels into concat0.  Here ONE WAVE owns a window:
// conv0's cells are evaluated where they are needed from the per-pixel maps T [N,H,W,9*32] and Gp [N,H+4,W+4,9*32]
// (the class convolutions of the factored cost volume, see kfn_cost_volume_gather in include/kfnet_hip.h) straight
// into an LDS image, neighbouring windows share 7/8 of their G reads through the L2; upconv0 is 72 MFMAs on the 4x4
// conv5 patch (one 16x16x32 product per (output parity class, live tap)); nothing but conv1a's [P,4,4,32] and the
// flow leaves the chip.  All MFMAs are v_mfma_f32_16x16x4_f32 (exact fp32; kfn_oflow_tail2_f16: 16x16x16 f16 for config 5); every layer's weights live in registers
// for the whole kernel (conv1a 144, upconv0 72, conv6 108 per lane).
//
// LDS images (per wave; zero borders written once, interiors rewritten per window):
//   tail2: concat0 10x10 cells x 56 floats (cell (cy+1, cx+1); channels [0,16) upconv0, [16,48) conv0),
//          conv6 10 rows x (10 cells x 16 + 4 pad), conv5 patch 5x5 x 36 (zero row 0 / column 0: the transposed conv's i-1, j-1 taps).
//          An M-block of conv6 is 8 rows x 2 columns of the window: with 56 floats per cell the A-fragment
//          ds_read_b128s are bank-conflict free (the 2x8 blocks of kfn_oflow_tail.hip cannot be).
//   head:  conv0 9x9 cells x 36 floats (TF SAME for stride 2 on an even size pads bottom / right only).
#include "kfn_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr unsigned OOBV = 0x80000000u;   // voffset that always fails the buffer range check -> reads 0

constexpr int C0 = 32;            // conv0 / conv1a / conv5 channels
constexpr int CU = 16;            // upconv0 channels
constexpr int C2 = 16;            // conv6 channels
constexpr int C9 = 9 * C0;        // channels of the T / G maps
constexpr int LD1 = 56;           // floats per cell: concat0 image
constexpr int LD2 = C2;           // conv6 image: 16 floats per cell, rows of 10 cells + 4 floats.  The prediction conv reads it
constexpr int ROW2 = 10 * LD2 + 4; // with lane = window cell, a ds_read_b128 per (tap, channel quad): with this pitch the 16 lanes of
                                   // every read group hit 16 different 16-byte slots (20-float cells in rows of 200: 24 double hits
                                   // per read, 50 M conflict cycles per launch -- round 3's PMC)
constexpr int LD3 = 36;           // conv5 patch image
constexpr int LDA = 36;           // head: conv0 image
constexpr int T1_BYTES = 100 * LD1 * 4;
constexpr int T2_BYTES = 10 * ROW2 * 4;
constexpr int T3_BYTES = 25 * LD3 * 4;
constexpr int TAIL_WAVE_BYTES = T1_BYTES + T2_BYTES + T3_BYTES;
constexpr int HEAD_WAVE_BYTES = 81 * LDA * 4;

__device__ __forceinline__ f32x4 buf_load(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0));
}

// four fp32 values of an LDS image -> the four halfs of a v_mfma_f32_16x16x16_f16 operand (round to nearest even)
__device__ __forceinline__ f16x4 to_h4(const f32x4& v) {
  return f16x4{(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
}

struct FusedArgs {
  const float* T;      // [N,H,W,9*32]
  const float* Gp;     // [N,H+4,W+4,9*32]
  int N, H, W, P;
  unsigned t_bytes, g_bytes;
  int relu0;           // conv0's ReLU (OFlowNet.py:19)
};

// conv0 of one window, this lane's share: item k = it*64 + lane = (cell = k / 8, channel quad q = k % 8).
// Border class of a cell (first / interior / last per axis) picks which of the 9 class kernels applies;
// the G term is read at the cell's shifted pixel in the map extended by 2 and is 0 outside it.
struct Conv0Stage {
  unsigned t_off[8];   // byte offset inside T[p] (wave-uniform part added per window)
  int g_dy[8], g_dx[8];
  unsigned g_ch[8];
  __device__ __forceinline__ void init(int lane) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int k = it * 64 + lane, cell = k >> 3, q = k & 7;
      const int ci = cell >> 3, cj = cell & 7;
      const int cls = ((ci == 0) ? 0 : (ci == 7 ? 2 : 1)) * 3 + ((cj == 0) ? 0 : (cj == 7 ? 2 : 1));
      t_off[it] = (unsigned)(cls * C0 + q * 4) * 4u;
      g_ch[it] = t_off[it];
      g_dy[it] = ci - 2;
      g_dx[it] = cj - 2;
    }
  }
};

__device__ __forceinline__ void conv0_issue(const FusedArgs& a, const Conv0Stage& st, __amdgpu_buffer_rsrc_t rsT,
                                            __amdgpu_buffer_rsrc_t rsG, int p_lane, f32x4 (&tv)[8], f32x4 (&gv)[8]) {
  // p -> (n, yy, x): wave-uniform (readfirstlane keeps the divisions on the scalar unit)
  const int p = __builtin_amdgcn_readfirstlane(p_lane);
  const int x = p % a.W;
  const int t2 = p / a.W;
  const int yy = t2 % a.H;
  const int n = t2 / a.H;
  const int Hp = a.H + 4, Wp = a.W + 4;
  const unsigned t_base = (unsigned)p * (unsigned)(C9 * 4);
  const unsigned g_img = (unsigned)n * (unsigned)(Hp * Wp) * (unsigned)(C9 * 4);
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    tv[it] = buf_load(rsT, st.t_off[it], t_base);
    const int gy = yy + st.g_dy[it], gx = x + st.g_dx[it];
    const bool ok = (unsigned)gy < (unsigned)Hp && (unsigned)gx < (unsigned)Wp;
    const unsigned voff = ok ? (unsigned)(gy * Wp + gx) * (unsigned)(C9 * 4) + st.g_ch[it] : OOBV;
    gv[it] = buf_load(rsG, voff, g_img);
  }
}

__device__ __forceinline__ f32x4 conv0_value(f32x4 t, f32x4 g, bool relu) {
  f32x4 v = t - g;
  if (relu) {
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
  }
  return v;
}

// ------------------------------------------------------------------------------------------------
// head: conv0 -> conv1a
// ------------------------------------------------------------------------------------------------
// H16 (kfn_oflow_head_f16, BASELINE config 5): conv1a on v_mfma_f32_16x16x16_f16 -- the conv0 image stays fp32, a lane's
// four consecutive channels are rounded to halfs where the MFMA reads them; 36 MFMAs per window instead of 144.
template <bool H16>
__global__ __launch_bounds__(256, 2) void oflow_head_kernel(FusedArgs a, const void* __restrict__ w1p_,
                                                            const float* __restrict__ b1, float* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) char smem_of[];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  char* const tA = smem_of + wv * HEAD_WAVE_BYTES;
  for (int i = lane; i < HEAD_WAVE_BYTES / 16; i += 64) reinterpret_cast<f32x4*>(tA)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  __builtin_amdgcn_wave_barrier();   // the zero fill is ordered before the first interior stores (same wave; pins the compiler)

  // conv1a weights: fragment t = (tap*8 + j)*2 + nb of lane (n = l%16, kq = l/16) = w[tap][kq*8 + j][nb*16 + n]
  // H16: fragment t = (tap*2 + s)*2 + nb, four halfs j = w[tap][kq*8 + 4s + j][nb*16 + n]   (graph.pack_oflow_head_kernel_f16)
  float wreg[H16 ? 1 : 144];
  f16x4 wh[H16 ? 36 : 1];
  if constexpr (H16) {
#pragma unroll
    for (int t = 0; t < 36; ++t) wh[t] = static_cast<const f16x4*>(w1p_)[t * 64 + lane];
  } else {
#pragma unroll
    for (int t = 0; t < 144; ++t) wreg[t] = static_cast<const float*>(w1p_)[t * 64 + lane];
  }
  const int li = lane & 15, kq = lane >> 4;
  const float bias0 = b1 ? b1[li] : 0.f, bias1 = b1 ? b1[16 + li] : 0.f;

  Conv0Stage st;
  st.init(lane);
  int st_off[8];
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int k = it * 64 + lane, cell = k >> 3, q = k & 7;
    st_off[it] = ((cell >> 3) * 9 + (cell & 7)) * (LDA * 4) + q * 16;
  }
  // A fragments: row r = output cell (oi, oj) = (r>>2, r&3); tap (ky,kx) reads conv0 cell (2 oi + ky, 2 oj + kx)
  const int a_base = ((2 * (li >> 2)) * 9 + 2 * (li & 3)) * (LDA * 4) + kq * 32;

  const __amdgpu_buffer_rsrc_t rsT = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.T), 0, (int)a.t_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Gp), 0, (int)a.g_bytes, 0x00020000);
  const bool relu0 = a.relu0 != 0;

  f32x4 tv[8], gv[8];
  int p = blockIdx.x * 4 + wv;
  const int pstride = gridDim.x * 4;
  if (p < a.P) conv0_issue(a, st, rsT, rsG, p, tv, gv);
  for (; p < a.P; p += pstride) {
#pragma unroll
    for (int it = 0; it < 8; ++it) *reinterpret_cast<f32x4*>(tA + st_off[it]) = conv0_value(tv[it], gv[it], relu0);
    const int pn = p + pstride;
    if (pn < a.P) conv0_issue(a, st, rsT, rsG, pn, tv, gv);   // wave-uniform branch
    __builtin_amdgcn_wave_barrier();

    f32x4 acc[2] = {f32x4{bias0, bias0, bias0, bias0}, f32x4{bias1, bias1, bias1, bias1}};
    if constexpr (H16) {
      // all 18 A reads first (the MFMAs are 4 passes each: read-wait-multiply per tap would be LDS latency, 36 times),
      // four independent accumulators (nb, s), summed at the end
      f32x4 av[9][2];
#pragma unroll
      for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int sq = 0; sq < 2; ++sq)
          av[tap][sq] = *reinterpret_cast<const f32x4*>(tA + a_base + ((tap / 3) * 9 + (tap % 3)) * (LDA * 4) + sq * 16);
      __builtin_amdgcn_sched_barrier(0);
      f32x4 acc2[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const f16x4 h0 = to_h4(av[tap][0]), h1 = to_h4(av[tap][1]);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          acc[nb] = __builtin_amdgcn_mfma_f32_16x16x16f16(h0, wh[(tap * 2 + 0) * 2 + nb], acc[nb], 0, 0, 0);
          acc2[nb] = __builtin_amdgcn_mfma_f32_16x16x16f16(h1, wh[(tap * 2 + 1) * 2 + nb], acc2[nb], 0, 0, 0);
        }
      }
      acc[0] += acc2[0];
      acc[1] += acc2[1];
    } else {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int toff = ((tap / 3) * 9 + (tap % 3)) * (LDA * 4);
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(tA + a_base + toff);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(tA + a_base + toff + 16);
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
            acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(j < 4 ? a0[j & 3] : a1[j & 3], wreg[(tap * 8 + j) * 2 + nb],
                                                           acc[nb], 0, 0, 0);
      }
    }
    // accumulator element e of lane (n, kq) = output cell (kq, e), channel nb*16 + n; ReLU (OFlowNet.py:20)
    float* yp = y + (size_t)p * (16 * C0);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int e = 0; e < 4; ++e) yp[(kq * 4 + e) * C0 + nb * 16 + li] = fmaxf(acc[nb][e], 0.f);
    __builtin_amdgcn_wave_barrier();
  }
}

// ------------------------------------------------------------------------------------------------
// tail: upconv0 ++ conv0 -> conv6 -> prediction -> softmax -> soft-argmax
// ------------------------------------------------------------------------------------------------
// H16 (BASELINE config 5, "fp16 convs"): upconv0 and conv6 on v_mfma_f32_16x16x16_f16 -- the LDS images stay fp32 (conv0's
// T - G, the ReLUs, 'prediction', softmax and the soft-argmax are fp32 arithmetic as before), a lane's four consecutive
// channels of an A read are rounded to halfs where they are consumed, the weights arrive as halfs: 27 + 18 operand
// registers pairs instead of 108 + 72 floats, 108 + 18 MFMAs per window instead of 432 + 72 (at 16x the rate).
template <bool H16>
__global__ __launch_bounds__(256, 1) void oflow_tail2_kernel(FusedArgs a, const float* __restrict__ x5,
                                                             const void* __restrict__ wup_, const float* __restrict__ bu,
                                                             const void* __restrict__ w6p_, const float* __restrict__ b6,
                                                             const float* __restrict__ wp, const float* __restrict__ bp,
                                                             float* __restrict__ flow, float* __restrict__ logits_out) {
  extern __shared__ __attribute__((aligned(16))) char smem_of[];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  char* const t1 = smem_of + wv * TAIL_WAVE_BYTES;
  char* const t2 = t1 + T1_BYTES;
  char* const t3 = t2 + T2_BYTES;
  for (int i = lane; i < TAIL_WAVE_BYTES / 16; i += 64) reinterpret_cast<f32x4*>(t1)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  __builtin_amdgcn_wave_barrier();   // the zero fill is ordered before the first interior stores (same wave; pins the compiler)

  // conv6: fragment t = tap*12 + j of lane (n, kq) = w6[tap][kq*12 + j][n]   (graph.pack_oflow_tail_kernel)
  // H16: fragment t = tap*3 + s, four halfs j = w6[tap][kq*12 + 4s + j][n]    (graph.pack_oflow_tail_kernel_f16)
  float w6r[H16 ? 1 : 108];
  f16x4 w6h[H16 ? 27 : 1];
  if constexpr (H16) {
#pragma unroll
    for (int t = 0; t < 27; ++t) w6h[t] = static_cast<const f16x4*>(w6p_)[t * 64 + lane];
  } else {
#pragma unroll
    for (int t = 0; t < 108; ++t) w6r[t] = static_cast<const float*>(w6p_)[t * 64 + lane];
  }
  // upconv0: fragment t = tap*8 + j = wu[ky][kx][n][kq*8 + j]                  (graph.pack_oflow_upconv_kernel)
  // H16: fragment t = tap*2 + s, four halfs j = wu[ky][kx][n][kq*8 + 4s + j]  (graph.pack_oflow_upconv_kernel_f16)
  float wur[H16 ? 1 : 72];
  f16x4 wuh[H16 ? 18 : 1];
  if constexpr (H16) {
#pragma unroll
    for (int t = 0; t < 18; ++t) wuh[t] = static_cast<const f16x4*>(wup_)[t * 64 + lane];
  } else {
#pragma unroll
    for (int t = 0; t < 72; ++t) wur[t] = static_cast<const float*>(wup_)[t * 64 + lane];
  }
  const int li = lane & 15, kq = lane >> 4;
  const float bias6 = b6 ? b6[li] : 0.f;
  const float biasu = bu ? bu[li] : 0.f;
  const float bias_p = bp ? bp[0] : 0.f;

  Conv0Stage st;
  st.init(lane);
  int st_off[8];   // conv0 item -> concat0 image, channels 16..47 of padded cell (cy+1, cx+1)
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int k = it * 64 + lane, cell = k >> 3, q = k & 7;
    st_off[it] = (((cell >> 3) + 1) * 10 + (cell & 7) + 1) * (LD1 * 4) + (CU + q * 4) * 4;
  }
  int s5_off[2];   // conv5 patch item k = it*64 + lane -> (cell = k/8 of the 4x4 patch, quad k%8), padded cell (i+1, j+1)
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int k = it * 64 + lane, cell = k >> 3, q = k & 7;
    s5_off[it] = (((cell >> 2) + 1) * 5 + (cell & 3) + 1) * (LD3 * 4) + q * 16;
  }
  // upconv0 A fragments: row r = input cell (i, j) = (r>>2, r&3); shift (di,dj) reads padded cell (i+1-di, j+1-dj)
  const int u_base = (((li >> 2) + 1) * 5 + (li & 3) + 1) * (LD3 * 4) + kq * 32;
  // upconv0 result: element e of lane (n, kq), class (pa,pb) = output cell (2 kq + pa, 2 e + pb), channel n
  const int uo_base = ((2 * kq + 1) * 10 + 1) * (LD1 * 4) + li * 4;
  // conv6 A fragments: M-block mb = window columns 2mb, 2mb+1; row r = cell (r>>1, 2mb + (r&1))
  const int a_base = ((li >> 1) * 10 + (li & 1)) * (LD1 * 4) + kq * 48;
  // conv6 result: element e of block mb = cell (2 kq + (e>>1), 2 mb + (e&1)), channel n
  const int o_base = ((2 * kq + 1) * ROW2 + LD2) * 4 + li * 4;
  // prediction: lane = window cell (ci, cj)
  const int ci = lane >> 3, cj = lane & 7;
  const int p_base = (ci * ROW2 + cj * LD2) * 4;

  const __amdgpu_buffer_rsrc_t rsT = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.T), 0, (int)a.t_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Gp), 0, (int)a.g_bytes, 0x00020000);
  const bool relu0 = a.relu0 != 0;

  f32x4 tv[8], gv[8], v5[2];
  int p = blockIdx.x * 4 + wv;
  const int pstride = gridDim.x * 4;
  if (p < a.P) {
    conv0_issue(a, st, rsT, rsG, p, tv, gv);
    const f32x4* src = reinterpret_cast<const f32x4*>(x5 + (size_t)p * (16 * C0));
    v5[0] = src[lane];
    v5[1] = src[64 + lane];
  }
  for (; p < a.P; p += pstride) {
    // ---- this window into LDS, the next one into registers -----------------------------------
#pragma unroll
    for (int it = 0; it < 8; ++it) *reinterpret_cast<f32x4*>(t1 + st_off[it]) = conv0_value(tv[it], gv[it], relu0);
    *reinterpret_cast<f32x4*>(t3 + s5_off[0]) = v5[0];
    *reinterpret_cast<f32x4*>(t3 + s5_off[1]) = v5[1];
    const int pn = p + pstride;
    if (pn < a.P) {   // wave-uniform
      conv0_issue(a, st, rsT, rsG, pn, tv, gv);
      const f32x4* src = reinterpret_cast<const f32x4*>(x5 + (size_t)pn * (16 * C0));
      v5[0] = src[lane];
      v5[1] = src[64 + lane];
    }
    __builtin_amdgcn_wave_barrier();

    // ---- upconv0: y[2i+pa, 2j+pb] = b + sum over the live taps of class (pa,pb) of x[i-di, j-dj] . w[ky][kx] --------
    // (TF conv2d_transpose 3x3 stride 2 SAME 4 -> 8, pad_t = pad_l = 0: even rows take ky = 0 (input row i) and
    //  ky = 2 (row i-1), odd rows ky = 1 (row i); the same per column -- 4 + 2 + 2 + 1 = 9 (class, tap) products)
    f32x4 ua[2][2][2];   // [di][dj][k quad]
#pragma unroll
    for (int di = 0; di < 2; ++di)
#pragma unroll
      for (int dj = 0; dj < 2; ++dj)
#pragma unroll
        for (int s = 0; s < 2; ++s)
          ua[di][dj][s] = *reinterpret_cast<const f32x4*>(t3 + u_base - (di * 5 + dj) * (LD3 * 4) + s * 16);
    if constexpr (H16) __builtin_amdgcn_sched_barrier(0);   // the eight reads as one group (4-pass MFMAs hide no LDS latency)
    f32x4 ud[2][2];
#pragma unroll
    for (int pa = 0; pa < 2; ++pa)
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) ud[pa][pb] = f32x4{biasu, biasu, biasu, biasu};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int pa = (ky == 1) ? 1 : 0, pb = (kx == 1) ? 1 : 0;
        const int di = (ky == 2) ? 1 : 0, dj = (kx == 2) ? 1 : 0;
        if constexpr (H16) {
#pragma unroll
          for (int s = 0; s < 2; ++s)
            ud[pa][pb] = __builtin_amdgcn_mfma_f32_16x16x16f16(to_h4(ua[di][dj][s]), wuh[(ky * 3 + kx) * 2 + s], ud[pa][pb], 0, 0, 0);
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j)
            ud[pa][pb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[di][dj][j >> 2][j & 3], wur[(ky * 3 + kx) * 8 + j],
                                                              ud[pa][pb], 0, 0, 0);
        }
      }
#pragma unroll
    for (int pa = 0; pa < 2; ++pa)
#pragma unroll
      for (int pb = 0; pb < 2; ++pb)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          *reinterpret_cast<float*>(t1 + uo_base + (pa * 10 + 2 * e + pb) * (LD1 * 4)) = fmaxf(ud[pa][pb][e], 0.f);
    __builtin_amdgcn_wave_barrier();

    // ---- conv6: 4 M-blocks x 9 taps x 12 k-steps of 16x16x4, two M-blocks interleaved ----------
    f32x4 acc[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) acc[mb] = f32x4{bias6, bias6, bias6, bias6};
    if constexpr (H16) {
      // (four independent accumulators per tap: a 16x16x16 MFMA is 4 passes, its result is needed again 3 MFMAs later)
      // and the 12 A reads of tap + 1 are issued as one group before the 12 MFMAs of tap: left to itself the compiler
      // emits read - wait - MFMA triples, 108 LDS latencies per window
      f32x4 af[2][4][3];
      auto a_reads = [&](int tap, f32x4 (&dst)[4][3]) __attribute__((always_inline)) {
        const int toff = ((tap / 3) * 10 + (tap % 3)) * (LD1 * 4);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
          for (int s = 0; s < 3; ++s)
            dst[mb][s] = *reinterpret_cast<const f32x4*>(t1 + a_base + mb * (2 * LD1 * 4) + toff + s * 16);
      };
      a_reads(0, af[0]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        if (tap < 8) a_reads(tap + 1, af[(tap + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int sq = 0; sq < 3; ++sq)
#pragma unroll
          for (int mb = 0; mb < 4; ++mb)
            acc[mb] = __builtin_amdgcn_mfma_f32_16x16x16f16(to_h4(af[tap & 1][mb][sq]), w6h[tap * 3 + sq], acc[mb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int mp = 0; mp < (H16 ? 0 : 2); ++mp) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int toff = ((tap / 3) * 10 + (tap % 3)) * (LD1 * 4);
        f32x4 af[2][3];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int s = 0; s < 3; ++s)
            af[m][s] = *reinterpret_cast<const f32x4*>(t1 + a_base + (mp * 2 + m) * (2 * LD1 * 4) + toff + s * 16);
#pragma unroll
        for (int j = 0; j < 12; ++j)
#pragma unroll
          for (int m = 0; m < 2; ++m)
            acc[mp * 2 + m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m][j >> 2][j & 3], w6r[tap * 12 + j], acc[mp * 2 + m], 0, 0, 0);
      }
    }
    // ---- ReLU, conv6 image -------------------------------------------------------------------
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        *reinterpret_cast<float*>(t2 + o_base + ((e >> 1) * ROW2 + (2 * mb + (e & 1)) * LD2) * 4) = fmaxf(acc[mb][e], 0.f);
    __builtin_amdgcn_wave_barrier();

    // ---- prediction conv: one window cell per lane, wave-uniform weights -------------------------
    // The 36 reads go out as one group (one wave per SIMD: nothing else hides an LDS latency per read -- 36 of them were
    // 3.6 K cycles per window), then 144 FMAs in four independent chains (one per channel quad).
    f32x4 xv[9][C2 / 4];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int q = 0; q < C2 / 4; ++q)
        xv[tap][q] = *reinterpret_cast<const f32x4*>(t2 + p_base + ((tap / 3) * ROW2 + (tap % 3) * LD2) * 4 + q * 16);
    __builtin_amdgcn_sched_barrier(0);
    float lq[C2 / 4];
#pragma unroll
    for (int q = 0; q < C2 / 4; ++q) lq[q] = 0.f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      // (wave-uniform scalar loads; hipcc hoists the 144 of them out of the window loop and spills the SGPRs to VGPR
      //  lanes: 156 v_readlane per window.  Per-lane copies end up in AGPRs and cost a v_accvgpr_read each: no better.)
      const float* wt = wp + tap * C2;
#pragma unroll
      for (int q = 0; q < C2 / 4; ++q) {
        lq[q] = fmaf(xv[tap][q].x, wt[q * 4 + 0], lq[q]);
        lq[q] = fmaf(xv[tap][q].y, wt[q * 4 + 1], lq[q]);
        lq[q] = fmaf(xv[tap][q].z, wt[q * 4 + 2], lq[q]);
        lq[q] = fmaf(xv[tap][q].w, wt[q * 4 + 3], lq[q]);
      }
    }
    const float lg = ((lq[0] + lq[1]) + (lq[2] + lq[3])) + bias_p;
    if (logits_out) logits_out[(size_t)p * 64 + lane] = lg;
    // ---- softmax over the 64 cells + soft-argmax (DPP reductions: kfn_common.h) -----------
    const float mx = kfn::wave_max_dpp(lg);
    const float ex = expf(lg - mx);
    const float se = kfn::wave_sum_dpp(ex);
    const float pr = ex / se;
    const float sx = kfn::wave_sum_dpp(pr * (float)(cj - 4));
    const float sy = kfn::wave_sum_dpp(pr * (float)(ci - 4));
    if (lane == 0) {
      flow[(size_t)p * 2 + 0] = sx;
      flow[(size_t)p * 2 + 1] = sy;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

int fill_args(FusedArgs& a, const float* T, const float* Gp, int N, int H, int W, int relu0, const char* who) {
  KFN_REQUIRE(T && Gp, "%s: null argument", who);
  KFN_REQUIRE(N > 0 && H > 0 && W > 0, "%s: bad shape N=%d H=%d W=%d", who, N, H, W);
  const long P = (long)N * H * W;
  const long t_bytes = P * C9 * 4L, g_bytes = (long)N * (H + 4) * (W + 4) * C9 * 4L;
  KFN_REQUIRE(P < (1L << 31) && t_bytes < (1L << 31) && g_bytes < (1L << 31),
              "%s: T / G maps beyond 2 GiB of 32-bit buffer offsets (N=%d)", who, N);
  KFN_REQUIRE(((reinterpret_cast<uintptr_t>(T) | reinterpret_cast<uintptr_t>(Gp)) & 15) == 0, "%s: misaligned map", who);
  a.T = T; a.Gp = Gp; a.N = N; a.H = H; a.W = W; a.P = (int)P;
  a.t_bytes = (unsigned)t_bytes; a.g_bytes = (unsigned)g_bytes; a.relu0 = relu0;
  return KFN_OK;
}

}  // namespace

namespace {
template <bool H16>
int launch_head(const char* who, const float* T, const float* Gp, int N, int H, int W, int relu0, const void* w1_packed,
                const float* b1, float* y, void* stream) {
  KFN_REQUIRE(w1_packed && y, "%s: null argument", who);
  KFN_REQUIRE((reinterpret_cast<uintptr_t>(w1_packed) & 7) == 0, "%s: packed weights must be 8-byte aligned", who);
  FusedArgs a;
  int rc = fill_args(a, T, Gp, N, H, W, relu0, who);
  if (rc != KFN_OK) return rc;
  static std::atomic<uint64_t> attr_done{0};   // (one per instantiation)
  rc = kfn::set_max_dynamic_lds(reinterpret_cast<const void*>(oflow_head_kernel<H16>), 4 * HEAD_WAVE_BYTES, attr_done);
  if (rc != KFN_OK) return rc;
  int blocks = kfn::ceil_div(a.P, 4);
  if (blocks > 512) blocks = 512;      // two workgroups of four waves per CU, each wave walks its windows
  hipLaunchKernelGGL(oflow_head_kernel<H16>, dim3(blocks), dim3(256), 4 * HEAD_WAVE_BYTES, (hipStream_t)stream, a,
                     w1_packed, b1, y);
  KFN_LAUNCH_CHECK("oflow_head_kernel");
  return KFN_OK;
}
}  // namespace

extern "C" int kfn_oflow_head(const float* T, const float* Gp, int N, int H, int W, int relu0, const float* w1_packed,
                              const float* b1, float* y, void* stream) {
  return launch_head<false>("kfn_oflow_head", T, Gp, N, H, W, relu0, w1_packed, b1, y, stream);
}

extern "C" int kfn_oflow_head_f16(const float* T, const float* Gp, int N, int H, int W, int relu0, const void* w1_packed_f16,
                                  const float* b1, float* y, void* stream) {
  return launch_head<true>("kfn_oflow_head_f16", T, Gp, N, H, W, relu0, w1_packed_f16, b1, y, stream);
}

namespace {
template <bool H16>
int launch_tail2(const char* who, const float* T, const float* Gp, int N, int H, int W, int relu0, const float* x5,
                 const void* wu_packed, const float* bu, const void* w6_packed, const float* b6, const float* wp,
                 const float* bp, float* flow_xy, float* opt_logits, void* stream) {
  KFN_REQUIRE(x5 && wu_packed && w6_packed && wp && flow_xy, "%s: null argument", who);
  KFN_REQUIRE((reinterpret_cast<uintptr_t>(x5) & 15) == 0, "%s: x5 must be 16-byte aligned", who);
  KFN_REQUIRE(((reinterpret_cast<uintptr_t>(wu_packed) | reinterpret_cast<uintptr_t>(w6_packed)) & 7) == 0,
              "%s: packed weights must be 8-byte aligned", who);
  FusedArgs a;
  int rc = fill_args(a, T, Gp, N, H, W, relu0, who);
  if (rc != KFN_OK) return rc;
  static std::atomic<uint64_t> attr_done{0};   // (one per instantiation)
  rc = kfn::set_max_dynamic_lds(reinterpret_cast<const void*>(oflow_tail2_kernel<H16>), 4 * TAIL_WAVE_BYTES, attr_done);
  if (rc != KFN_OK) return rc;
  int blocks = kfn::ceil_div(a.P, 4);
  if (blocks > 256) blocks = 256;      // one workgroup of four waves per CU
  hipLaunchKernelGGL(oflow_tail2_kernel<H16>, dim3(blocks), dim3(256), 4 * TAIL_WAVE_BYTES, (hipStream_t)stream, a, x5,
                     wu_packed, bu, w6_packed, b6, wp, bp, flow_xy, opt_logits);
  KFN_LAUNCH_CHECK("oflow_tail2_kernel");
  return KFN_OK;
}
}  // namespace

extern "C" int kfn_oflow_tail2(const float* T, const float* Gp, int N, int H, int W, int relu0, const float* x5,
                               const float* wu_packed, const float* bu, const float* w6_packed, const float* b6,
                               const float* wp, const float* bp, float* flow_xy, float* opt_logits, void* stream) {
  return launch_tail2<false>("kfn_oflow_tail2", T, Gp, N, H, W, relu0, x5, wu_packed, bu, w6_packed, b6, wp, bp, flow_xy,
                             opt_logits, stream);
}

extern "C" int kfn_oflow_tail2_f16(const float* T, const float* Gp, int N, int H, int W, int relu0, const float* x5,
                                   const void* wu_packed_f16, const float* bu, const void* w6_packed_f16, const float* b6,
                                   const float* wp, const float* bp, float* flow_xy, float* opt_logits, void* stream) {
  return launch_tail2<true>("kfn_oflow_tail2_f16", T, Gp, N, H, W, relu0, x5, wu_packed_f16, bu, w6_packed_f16, b6, wp, bp,
                            flow_xy, opt_logits, stream);
}
