// kfn_stream_ops.hip -- the HBM-bound kernels of the path: uint8 image ingest + first
// convolutions (Cin = 3), local cost volume, softmax/soft-argmax flow head, channel copy.
#include "kfn_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4s __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------
// uint8 image -> (x-128)*0.00625 -> 3x3 SAME conv (Cin = 3) + bias + ReLU, two heads.
// SCoordNet.preprocess+conv1a (cnn_wrapper/SCoordNet.py:20-21,34-37) and the feature
// tower's preprocess+feat1 (KFNet/KFNet.py:317-320) read the image ONCE.
// Block = 64x4 output pixels; the 66x6x3 preprocessed halo lives in LDS (zero padding is
// applied in the preprocessed domain, as TF pads the already-normalised tensor).
// ------------------------------------------------------------------------------------
constexpr int FT_W = 64, FT_H = 4;
constexpr int FT_TW3 = (FT_W + 2) * 3;

// One head of the first layer.  LP = C/4 lanes cooperate on one pixel: lane s owns output
// channels [4s, 4s+4) (27 float4 weights in registers), so a pixel's C floats leave the
// wave as one contiguous LP*16-byte run and the 64/LP pixels a wave handles per round are
// x-adjacent: fully coalesced stores (the layer is store-bound: 98 MB/frame written).
// OUT16: the head's output tensor holds IEEE halfs (BASELINE config 5, fp16 activations): the four channels of a
// lane are rounded once (RNE) and leave as one 8-byte store, a pixel's C halfs as one contiguous LP*8-byte run.
template <int LP, bool OUT16 = false>
__device__ __forceinline__ void first_head(const float* tile, const float* __restrict__ w,
                                           const float* __restrict__ b, void* __restrict__ yv,
                                           int n, int H, int W, int x0, int y0, int tid) {
  constexpr int C = LP * 4;
  constexpr int SLOTS = 256 / LP;            // pixels in flight per block round
  constexpr int ROUNDS = (FT_W * FT_H) / SLOTS;
  const int s = tid % LP;
  const int ps = tid / LP;
  f32x4 wk[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) wk[k] = *reinterpret_cast<const f32x4*>(w + k * C + s * 4);
  f32x4 bias = {0.f, 0.f, 0.f, 0.f};
  if (b) bias = *reinterpret_cast<const f32x4*>(b + s * 4);
#pragma unroll 2
  for (int it = 0; it < ROUNDS; ++it) {
    const int pi = it * SLOTS + ps;
    const int ty = pi / FT_W, tx = pi - ty * FT_W;
    const int gx = x0 + tx, gy = y0 + ty;
    f32x4 acc = bias;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const float* row = tile + (ty + ky) * FT_TW3 + tx * 3;
#pragma unroll
      for (int j = 0; j < 9; ++j) {
        const float xv = row[j];
        acc += xv * wk[ky * 9 + j];
      }
    }
    if (gx < W && gy < H) {
      f32x4 v = {fmaxf(acc.x, 0.f), fmaxf(acc.y, 0.f), fmaxf(acc.z, 0.f), fmaxf(acc.w, 0.f)};
      const size_t e = (((size_t)n * H + gy) * W + gx) * C + s * 4;
      if constexpr (OUT16) {
        typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
        const f16x4 h = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
        *reinterpret_cast<f16x4*>(static_cast<_Float16*>(yv) + e) = h;
      } else {
        *reinterpret_cast<f32x4*>(static_cast<float*>(yv) + e) = v;
      }
    }
  }
}

template <int LP1, int LP2, bool OUT16_1 = false>
__global__ __launch_bounds__(256) void first_conv_kernel(
    const uint8_t* __restrict__ img, int N, int H, int W,
    const float* __restrict__ w1, const float* __restrict__ b1, void* __restrict__ y1,
    const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ y2) {
  __shared__ float tile[(FT_H + 2) * FT_TW3];
  const int tid = threadIdx.x;
  const int x0 = blockIdx.x * FT_W, y0 = blockIdx.y * FT_H, n = blockIdx.z;
  const uint8_t* src = img + (size_t)n * H * W * 3;
  for (int i = tid; i < (FT_H + 2) * FT_TW3; i += 256) {
    int yy = i / FT_TW3, rem = i - yy * FT_TW3;
    int xx = rem / 3;
    int gy = y0 + yy - 1, gx = x0 + xx - 1;
    float v = 0.f;
    if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
      v = ((float)src[((size_t)gy * W + x0 - 1) * 3 + rem] - 128.0f) * 0.00625f;
    tile[i] = v;
  }
  __syncthreads();
  first_head<LP1, OUT16_1>(tile, w1, b1, y1, n, H, W, x0, y0, tid);
  if constexpr (LP2 > 0) first_head<LP2>(tile, w2, b2, y2, n, H, W, x0, y0, tid);
}

// ------------------------------------------------------------------------------------
// The same layer, LANE = PIXEL (round 3).  In the kernel above C/4 lanes share a pixel and all of them read the
// same 27 inputs from LDS: per pixel 4.2 ds_read2_b32 wave-instructions and 17 v_pk_fma_f32 -- the LDS pipe and the
// VALU are both at ~17 cycles per pixel and CU, and the layer runs at 3.5 TB/s although it only has to stream its
// output.  Here a wave owns one 64-pixel row of the 64x4 tile: a lane reads its 27 inputs ONCE (27 conflict-free
// ds_read_b32 per 64 pixels), the weights are wave-uniform and come from scalar registers, 80 accumulators per lane;
// the results are transposed through a per-wave LDS staging buffer so that they leave as 16-byte non-temporal
// stores in which 8 (fp32 half of 32 channels, or all 64 fp16 channels) or 4 (16-channel head) neighbouring lanes
// cover one pixel's contiguous 128- or 64-byte run.  The buffer holds 32 pixels x 32 dwords (the lanes of one
// wave half stage, all 64 lanes store) or 64 pixels x 16 dwords: 5 KiB per wave, 25 KiB per workgroup, so six
// workgroups (24 waves, the 80-VGPR limit) share a CU -- with a 64 x 32-dword buffer there were three, and the
// store queue drained between their epilogues.  The halo is filled with
// aligned dword loads of the uint8 rows (4 values per load; the byte-wise form remains for images whose rows are
// not a multiple of 4 bytes).
// ------------------------------------------------------------------------------------
constexpr int PX_SP = 36;                       // staging stride (dwords) of a 32-dword pixel: conflict-free ds_write_b128
constexpr int PX_SP16 = 20;                     // ... of a 16-dword pixel
constexpr int PX_STAGE = 64 * PX_SP16;          // per wave: 32 pixels x 32 dwords or 64 pixels x 16 dwords (5 KiB)
static_assert(PX_STAGE >= 32 * PX_SP, "both pass shapes fit");
constexpr int PX_LDS_BYTES = ((FT_H + 2) * FT_TW3 + 4 * PX_STAGE) * 4;

__device__ __constant__ float kZeroBias[64] = {0.f};

template <int C>
__device__ __forceinline__ void px_accumulate(const float* tile_row, const float* __restrict__ w, const float* __restrict__ b,
                                              float (&acc)[C]) {
  const float* bb = b ? b : kZeroBias;            // ONE uniform select (a test per channel becomes a branch per channel)
#pragma unroll
  for (int c = 0; c < C; ++c) acc[c] = bb[c];
  // A REAL loop over the 27 taps (C FMAs per trip): fully unrolled, the scheduler hoists the scalar weight loads
  // of many taps to the top and spills ~800 scalar registers into VGPR lanes.
  int off = 0, j = 0;
#pragma unroll 1
  for (int t = 0; t < 27; ++t) {
    const float xv = tile_row[off];
    const float* wk = w + t * C;                     // wave-uniform: scalar loads
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = fmaf(xv, wk[c], acc[c]);
    ++off;
    if (++j == 9) {
      j = 0;
      off += FT_TW3 - 9;
    }
  }
}

// 32 fp32 channels (NQ = 8 quads per pixel) or 16 (NQ = 4) of this wave's 64 pixels: LDS transpose, then
// 64/NQ pixels per store instruction.  `cstride` = channels per pixel in memory, `c0` = first channel.
template <class V>
__device__ __forceinline__ void px_st(V* p, const V& v) {
  __builtin_nontemporal_store(v, p);          // written once, read by the next layer after 3 GB of other stores
}

template <int NQ>
__device__ __forceinline__ void px_store_f32(float* stage, const float* vals, float* __restrict__ y, size_t row_elem0,
                                             int cstride, int c0, int lane, int valid_px) {
#pragma unroll
  for (int q = 0; q < NQ; ++q)
    *reinterpret_cast<f32x4*>(stage + lane * PX_SP16 + q * 4) = f32x4{vals[q * 4], vals[q * 4 + 1], vals[q * 4 + 2], vals[q * 4 + 3]};
  __builtin_amdgcn_wave_barrier();
  static_assert(NQ * 4 <= 16, "a pass stages at most 16 dwords per pixel");
  constexpr int PPI = 64 / NQ;                  // pixels per store instruction
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const int px = i * PPI + lane / NQ, q = lane % NQ;
    const f32x4 v = *reinterpret_cast<const f32x4*>(stage + px * PX_SP16 + q * 4);
    if (px < valid_px) px_st(reinterpret_cast<f32x4*>(y + row_elem0 + (size_t)px * cstride + c0 + q * 4), v);
  }
  __builtin_amdgcn_wave_barrier();
}

// 32 channels of 32 pixels (half `ph` of the wave): the half's lanes stage, all 64 lanes store 128-byte runs
__device__ __forceinline__ void px_store_half(float* stage, const float* vals, float* __restrict__ y, size_t row_elem0,
                                              int cstride, int c0, int lane, int valid_px, int ph) {
  if ((lane >> 5) == ph) {
#pragma unroll
    for (int q = 0; q < 8; ++q)
      *reinterpret_cast<f32x4*>(stage + (lane & 31) * PX_SP + q * 4) = f32x4{vals[q * 4], vals[q * 4 + 1], vals[q * 4 + 2], vals[q * 4 + 3]};
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int pl = i * 8 + lane / 8, q = lane % 8, px = ph * 32 + pl;
    const f32x4 v = *reinterpret_cast<const f32x4*>(stage + pl * PX_SP + q * 4);
    if (px < valid_px) px_st(reinterpret_cast<f32x4*>(y + row_elem0 + (size_t)px * cstride + c0 + q * 4), v);
  }
  __builtin_amdgcn_wave_barrier();
}

template <int C2, bool OUT16_1>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8))) void first_conv_px_kernel(
    const uint8_t* __restrict__ img, int N, int H, int W, unsigned img_bytes,
    const float* __restrict__ w1, const float* __restrict__ b1, void* __restrict__ y1,
    const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ y2) {
  constexpr int C1 = 64;
  extern __shared__ __attribute__((aligned(16))) float px_smem[];
  float* tile = px_smem;                                   // [(FT_H+2)][FT_TW3] preprocessed halo
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  float* stage = px_smem + (FT_H + 2) * FT_TW3 + wv * PX_STAGE;
  const int x0 = blockIdx.x * FT_W, y0 = blockIdx.y * FT_H, n = blockIdx.z;
  // ---- halo: aligned dwords of the uint8 rows, 4 values each -----------------------------------
  {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(img), 0, (int)img_bytes, 0x00020000);
    constexpr int DPR = (FT_TW3 + 3) / 4 + 1;              // dwords that cover a 198-byte run at any alignment
    for (int i = tid; i < (FT_H + 2) * DPR; i += 256) {
      const int yy = i / DPR, j = i - yy * DPR;
      const int gy = y0 + yy - 1;
      const long s = ((long)(n * H + gy) * W + x0 - 1) * 3;             // byte index of tile[yy][0]
      const long a = (s & ~3L) + 4L * j;
      const bool row_ok = (unsigned)gy < (unsigned)H;
      const unsigned word = row_ok && a >= 0 ? __builtin_amdgcn_raw_buffer_load_b32(rs, (unsigned)a, 0, 0) : 0u;
#pragma unroll
      for (int bb = 0; bb < 4; ++bb) {
        const int idx = (int)(a + bb - s);
        if (idx >= 0 && idx < FT_TW3) {
          const int gx = x0 - 1 + idx / 3;
          float v = 0.f;
          if (row_ok && (unsigned)gx < (unsigned)W) v = ((float)((word >> (8 * bb)) & 255u) - 128.0f) * 0.00625f;
          tile[yy * FT_TW3 + idx] = v;
        }
      }
    }
  }
  __syncthreads();
  const int gy = y0 + wv;                                  // this wave's image row
  if (gy >= H) return;                                     // (no block-level barrier below)
  const int valid_px = (W - x0 < 64) ? W - x0 : 64;
  const float* tile_row = tile + wv * FT_TW3 + lane * 3;
  const size_t pix0 = ((size_t)n * H + gy) * W + x0;
  {
    float acc[C1];
    px_accumulate<C1>(tile_row, w1, b1, acc);
#pragma unroll
    for (int c = 0; c < C1; ++c) acc[c] = fmaxf(acc[c], 0.f);
    if constexpr (OUT16_1) {
      // 64 halfs = 32 dwords per pixel, in two passes of 32 channels (16 dwords): 4 lanes per pixel and pass
      typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
      _Float16* yh = static_cast<_Float16*>(y1);
      // a pixel's 64 halfs are one 128-byte run: the lanes of half `ph` stage their pixels, all 64 lanes store
#pragma unroll
      for (int ph = 0; ph < 2; ++ph) {
        if ((lane >> 5) == ph) {
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            u32x4s pk;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const f16x2 h = {(_Float16)acc[q * 8 + 2 * e], (_Float16)acc[q * 8 + 2 * e + 1]};
              pk[e] = __builtin_bit_cast(unsigned, h);
            }
            *reinterpret_cast<u32x4s*>(stage + (lane & 31) * PX_SP + q * 4) = pk;
          }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int pl = i * 8 + lane / 8, q = lane % 8, px = ph * 32 + pl;
          const u32x4s v = *reinterpret_cast<const u32x4s*>(stage + pl * PX_SP + q * 4);
          if (px < valid_px) px_st(reinterpret_cast<u32x4s*>(yh + (pix0 + px) * C1 + q * 8), v);
        }
        __builtin_amdgcn_wave_barrier();
      }
    } else {
      float* yf = static_cast<float*>(y1);
      // four passes (pixel half x channel half) through a 5 KiB staging buffer: six workgroups per CU stay resident
#pragma unroll
      for (int cp = 0; cp < 4; ++cp) px_store_half(stage, acc + 32 * (cp & 1), yf, pix0 * C1, C1, 32 * (cp & 1), lane, valid_px, cp >> 1);
    }
  }
  if constexpr (C2 > 0) {
    static_assert(C2 == 16, "second head: 16 channels");
    float acc[C2];
    px_accumulate<C2>(tile_row, w2, b2, acc);
#pragma unroll
    for (int c = 0; c < C2; ++c) acc[c] = fmaxf(acc[c], 0.f);
    px_store_f32<4>(stage, acc, y2, pix0 * C2, C2, 0, lane, valid_px);
  }
}

// ------------------------------------------------------------------------------------
// KFNet.BuildCoordVolume (KFNet/KFNet.py:343-359) + reshape (:372), materialised form.
// One thread per float4 of the output: vol[p][i][j][c4] = f2[p][c4] - f1[shifted][c4].
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cost_volume_kernel(const float* __restrict__ f1,
                                                          const float* __restrict__ f2,
                                                          float* __restrict__ vol, int N, int H,
                                                          int W, int C, int window) {
  const int C4 = C >> 2;
  const long total = (long)N * H * W * window * window * C4;
  const int half = window >> 1;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    int c4 = (int)(idx % C4);
    long t = idx / C4;
    int j = (int)(t % window); t /= window;
    int i = (int)(t % window); t /= window;  // t = pixel index over N*H*W
    int x = (int)(t % W);
    long t2 = t / W;
    int y = (int)(t2 % H);
    int n = (int)(t2 / H);
    f32x4 a = *reinterpret_cast<const f32x4*>(f2 + t * C + c4 * 4);
    int sy = y + i - half, sx = x + j - half;
    f32x4 b = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W)
      b = *reinterpret_cast<const f32x4*>(f1 + (((long)n * H + sy) * W + sx) * C + c4 * 4);
    *reinterpret_cast<f32x4*>(vol + idx * 4) = a - b;
  }
}

// ------------------------------------------------------------------------------------
// Factored cost volume + conv0 (KFNet/KFNet.py:343-359,372 + cnn_wrapper/OFlowNet.py:19).
// conv0 is linear and every volume entry is f2[p] - f1[p + cell - 4], so for window cell
// (ci,cj) of pixel p
//     conv0(V)[p,ci,cj] = b + S_k f2[p] - G_k(p + (ci-4, cj-4)),   k = class(ci,cj),
// where S_k = sum of the 3x3 taps that stay inside the 8x8 window for that cell (conv0's own
// SAME padding: 3 row classes x 3 column classes) and G_k = 3x3 SAME convolution of the
// zero-extended f1 with the same taps.  T_k = b + S_k f2 (a 1x1 conv) and G_k (a 3x3 conv on
// the feature MAP, 9*C channels) are computed once per pixel instead of once per cell; this
// kernel gathers, subtracts and applies the ReLU: 64x less MFMA work than the per-cell conv.
//   T  [N,H,W,9*C]           class-major channels
//   Gp [N,H+4,W+4,9*C]       G on the map extended by 2 (conv of f1 zero-padded by 2)
//   y  [(N*H*W),8,8,C]       pixel stride ldy floats
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pad_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                        int N, int H, int W, int C, int pad) {
  const int C4 = C >> 2, Hp = H + 2 * pad, Wp = W + 2 * pad;
  const long total = (long)N * Hp * Wp * C4;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(idx % C4);
    long t = idx / C4;
    const int b = (int)(t % Wp); t /= Wp;
    const int a = (int)(t % Hp);
    const int n = (int)(t / Hp);
    const int sy = a - pad, sx = b - pad;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W)
      v = *reinterpret_cast<const f32x4*>(x + (((long)n * H + sy) * W + sx) * C + c4 * 4);
    *reinterpret_cast<f32x4*>(y + idx * 4) = v;
  }
}

__global__ __launch_bounds__(256) void cost_volume_gather_kernel(const float* __restrict__ T,
                                                                  const float* __restrict__ Gp,
                                                                  float* __restrict__ y, int N, int H, int W,
                                                                  int C, int ldy, int relu) {
  // one workgroup per pixel (grid-stride): the pixel decode is wave-uniform (scalar unit), a
  // thread handles (cell, channel quad) items; consecutive lanes = consecutive quads of a cell
  const int C4 = C >> 2, Hp = H + 4, Wp = W + 4, C9 = 9 * C;
  const int items = 64 * C4;
  const int P = N * H * W;
  for (int p = blockIdx.x; p < P; p += gridDim.x) {
    const int x = p % W;
    const int t2 = p / W;
    const int yy = t2 % H;
    const int n = t2 / H;
    const float* Tp = T + (size_t)p * C9;
    const float* Gn = Gp + (size_t)n * Hp * Wp * C9;
    float* yp = y + (size_t)p * 64 * ldy;
    for (int item = threadIdx.x; item < items; item += 256) {
      const int cell = item / C4, c4 = item - cell * C4;
      const int ci = cell >> 3, cj = cell & 7;
      const int cls = ((ci == 0) ? 0 : (ci == 7 ? 2 : 1)) * 3 + ((cj == 0) ? 0 : (cj == 7 ? 2 : 1));
      f32x4 v = *reinterpret_cast<const f32x4*>(Tp + cls * C + c4 * 4);
      const int a = yy + ci - 2, b = x + cj - 2;   // position in the map extended by 2
      if ((unsigned)a < (unsigned)Hp && (unsigned)b < (unsigned)Wp)
        v -= *reinterpret_cast<const f32x4*>(Gn + ((size_t)a * Wp + b) * C9 + cls * C + c4 * 4);
      if (relu) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      }
      *reinterpret_cast<f32x4*>(yp + cell * ldy + c4 * 4) = v;
    }
  }
}

// ------------------------------------------------------------------------------------
// softmax over the window cells (OFlowNet.py:45-47) + soft-argmax flow
// (KFNet/KFNet.py:381-385).  One 64-lane wavefront per pixel == one lane per cell of the
// 8x8 window; max / sum / weighted sums by wave shuffles.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__global__ __launch_bounds__(256) void flow_softargmax_kernel(const float* __restrict__ logits,
                                                              float* __restrict__ flow,
                                                              float* __restrict__ prob, int P,
                                                              int window) {
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int area = window * window;
  const int half = window >> 1;
  for (int p = blockIdx.x * 4 + wv; p < P; p += gridDim.x * 4) {
    float mx = -INFINITY;
    for (int k = lane; k < area; k += 64) mx = fmaxf(mx, logits[(size_t)p * area + k]);
    mx = wave_max(mx);
    float se = 0.f, sx = 0.f, sy = 0.f;
    for (int k = lane; k < area; k += 64) {
      float e = expf(logits[(size_t)p * area + k] - mx);
      se += e;
    }
    se = wave_sum(se);
    for (int k = lane; k < area; k += 64) {
      float pr = expf(logits[(size_t)p * area + k] - mx) / se;
      if (prob) prob[(size_t)p * area + k] = pr;
      int i = k / window, j = k - i * window;
      sx += pr * (float)(j - half);
      sy += pr * (float)(i - half);
    }
    sx = wave_sum(sx);
    sy = wave_sum(sy);
    if (lane == 0) {
      flow[(size_t)p * 2 + 0] = sx;
      flow[(size_t)p * 2 + 1] = sy;
    }
  }
}

// ------------------------------------------------------------------------------------
// OFlowNet 'prediction' conv (3x3, C -> 1, no ReLU; OFlowNet.py:41) fused with the softmax
// over the 64 window cells (OFlowNet.py:45-47) and the soft-argmax flow (KFNet.py:381-385).
// One wavefront per pixel, one lane per window cell (i,j): the pixel's 8x8xC activation
// tile is staged in LDS with a zero border (the conv's SAME padding), each lane does its
// 9*C MACs with wave-uniform weights, then max/sum/weighted-sum by wave shuffles.  The 64
// logits never go to HBM.
// ------------------------------------------------------------------------------------
constexpr int FH_LD = 36;  // LDS floats per tile cell (C <= 32, +4 pad against bank conflicts)

__global__ __launch_bounds__(256) void flow_head_kernel(const float* __restrict__ x,
                                                        const float* __restrict__ w,
                                                        const float* __restrict__ bias,
                                                        float* __restrict__ flow,
                                                        float* __restrict__ logits_out, int P, int C) {
  __shared__ __attribute__((aligned(16))) float tiles[4][10 * 10 * FH_LD];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  float* tile = tiles[wv];
  const int C4 = C >> 2;
  // zero border once (interior is overwritten per pixel)
  for (int i = lane; i < 100 * FH_LD; i += 64) tile[i] = 0.f;
  const int ci = lane >> 3, cj = lane & 7;
  const float b0 = bias ? bias[0] : 0.f;
  for (int p = blockIdx.x * 4 + wv; p < P; p += gridDim.x * 4) {
    const float* src = x + (size_t)p * 64 * C;
    // 64 cells x C floats, coalesced float4 loads; cell (a,b) -> tile[(a+1)*10 + (b+1)]
    for (int k = lane; k < 64 * C4; k += 64) {
      const int cell = k / C4, q = k - cell * C4;
      const f32x4 v = *reinterpret_cast<const f32x4*>(src + cell * C + q * 4);
      *reinterpret_cast<f32x4*>(tile + (((cell >> 3) + 1) * 10 + (cell & 7) + 1) * FH_LD + q * 4) = v;
    }
    __builtin_amdgcn_wave_barrier();
    float acc = b0;
    for (int ky = 0; ky < 3; ++ky)
      for (int kx = 0; kx < 3; ++kx) {
        const float* cellp = tile + ((ci + ky) * 10 + cj + kx) * FH_LD;
        const float* wp = w + (ky * 3 + kx) * C;
        for (int q = 0; q < C4; ++q) {
          const f32x4 xv = *reinterpret_cast<const f32x4*>(cellp + q * 4);
          acc = fmaf(xv.x, wp[q * 4 + 0], acc);
          acc = fmaf(xv.y, wp[q * 4 + 1], acc);
          acc = fmaf(xv.z, wp[q * 4 + 2], acc);
          acc = fmaf(xv.w, wp[q * 4 + 3], acc);
        }
      }
    if (logits_out) logits_out[(size_t)p * 64 + lane] = acc;
    const float mx = wave_max(acc);
    const float e = expf(acc - mx);
    const float se = wave_sum(e);
    const float pr = e / se;
    const float sx = wave_sum(pr * (float)(cj - 4));
    const float sy = wave_sum(pr * (float)(ci - 4));
    if (lane == 0) {
      flow[(size_t)p * 2 + 0] = sx;
      flow[(size_t)p * 2 + 1] = sy;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

__global__ __launch_bounds__(256) void copy_channels_kernel(const float* __restrict__ src, int ld_src,
                                                            float* __restrict__ dst, int ld_dst,
                                                            int P, int C) {
  const long total = (long)P * C;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    long p = idx / C;
    int c = (int)(idx - p * C);
    dst[p * ld_dst + c] = src[p * ld_src + c];
  }
}

}  // namespace

extern "C" int kfn_first_conv_u8_ex(const uint8_t* img, int N, int H, int W, const float* w1, const float* b1,
                                    void* y1, int C1, int y1_dtype, const float* w2, const float* b2, float* y2,
                                    int C2, void* stream);

extern "C" int kfn_first_conv_u8(const uint8_t* img, int N, int H, int W, const float* w1,
                                 const float* b1, float* y1, int C1, const float* w2,
                                 const float* b2, float* y2, int C2, void* stream) {
  return kfn_first_conv_u8_ex(img, N, H, W, w1, b1, y1, C1, KFN_ACT_F32, w2, b2, y2, C2, stream);
}

extern "C" int kfn_first_conv_u8_ex(const uint8_t* img, int N, int H, int W, const float* w1, const float* b1,
                                    void* y1, int C1, int y1_dtype, const float* w2, const float* b2, float* y2,
                                    int C2, void* stream) {
  KFN_REQUIRE(img && w1 && y1, "kfn_first_conv_u8: null argument");
  KFN_REQUIRE(y1_dtype == KFN_ACT_F32 || y1_dtype == KFN_ACT_F16, "kfn_first_conv_u8: unknown output dtype %d", y1_dtype);
  KFN_REQUIRE(N > 0 && H > 0 && W > 0, "kfn_first_conv_u8: bad shape");
  KFN_REQUIRE(C2 == 0 || (w2 && y2), "kfn_first_conv_u8: second head needs w2/y2");
  dim3 grid(kfn::ceil_div(W, FT_W), kfn::ceil_div(H, FT_H), N), block(256);
  hipStream_t s = (hipStream_t)stream;
#define KFN_FIRST(L1, L2)                                                                     \
  hipLaunchKernelGGL((first_conv_kernel<L1, L2>), grid, block, 0, s, img, N, H, W, w1, b1, y1, w2, b2, y2)
  // lane-per-pixel form for the 64(+16)-channel layers whose rows are whole dwords (every size the path uses)
  const long img_total = (long)N * H * W * 3;
  if (C1 == 64 && (C2 == 16 || C2 == 0) && (W * 3) % 4 == 0 && (reinterpret_cast<uintptr_t>(img) & 3) == 0 &&
      img_total < (1L << 31) && (reinterpret_cast<uintptr_t>(y1) & 15) == 0 && (C2 == 0 || (reinterpret_cast<uintptr_t>(y2) & 15) == 0)) {
    static std::atomic<uint64_t> attr_done[4] = {};
    const bool h16 = y1_dtype == KFN_ACT_F16;
    const int which = (C2 == 16 ? 0 : 1) + (h16 ? 2 : 0);
#define KFN_FIRST_PX(C2V, O16)                                                                                        \
    do {                                                                                                               \
      int rc = kfn::set_max_dynamic_lds(reinterpret_cast<const void*>(first_conv_px_kernel<C2V, O16>), PX_LDS_BYTES,   \
                                        attr_done[which]);                                                             \
      if (rc != KFN_OK) return rc;                                                                                     \
      hipLaunchKernelGGL((first_conv_px_kernel<C2V, O16>), grid, block, PX_LDS_BYTES, s, img, N, H, W,                 \
                         (unsigned)img_total, w1, b1, y1, w2, b2, y2);                                                 \
    } while (0)
    if (C2 == 16 && !h16) KFN_FIRST_PX(16, false);
    else if (C2 == 16) KFN_FIRST_PX(16, true);
    else if (!h16) KFN_FIRST_PX(0, false);
    else KFN_FIRST_PX(0, true);
#undef KFN_FIRST_PX
    KFN_LAUNCH_CHECK("first_conv_px_kernel");
    return KFN_OK;
  }
  if (y1_dtype == KFN_ACT_F16) {
    // fp16 activations (BASELINE config 5): the wide head (SCoordNet conv1a) writes halfs, a second head stays fp32
    if (C1 == 64 && C2 == 16)
      hipLaunchKernelGGL((first_conv_kernel<16, 4, true>), grid, block, 0, s, img, N, H, W, w1, b1, y1, w2, b2, y2);
    else if (C1 == 64 && C2 == 0)
      hipLaunchKernelGGL((first_conv_kernel<16, 0, true>), grid, block, 0, s, img, N, H, W, w1, b1, y1, w2, b2, y2);
    else
      return kfn::fail(KFN_ERR_UNSUPPORTED, "kfn_first_conv_u8: fp16 output is instantiated for head widths 64(+16) only");
  } else if (C1 == 64 && C2 == 16) KFN_FIRST(16, 4);
  else if (C1 == 64 && C2 == 0) KFN_FIRST(16, 0);
  else if (C1 == 16 && C2 == 0) KFN_FIRST(4, 0);
  else if (C1 == 32 && C2 == 0) KFN_FIRST(8, 0);
  else if (C1 == 16 && C2 == 64) KFN_FIRST(4, 16);
  else if (C1 == 32 && C2 == 16) KFN_FIRST(8, 4);
  else
    return kfn::fail(KFN_ERR_UNSUPPORTED, "kfn_first_conv_u8: head widths (%d,%d) not instantiated "
                     "(have 64+16, 16+64, 32+16, 64, 32, 16)", C1, C2);
#undef KFN_FIRST
  KFN_LAUNCH_CHECK("first_conv_kernel");
  return KFN_OK;
}

extern "C" int kfn_cost_volume(const float* f1, const float* f2, float* vol, int N, int H, int W,
                               int C, int window, void* stream) {
  KFN_REQUIRE(f1 && f2 && vol, "kfn_cost_volume: null argument");
  KFN_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && window > 0 && window % 2 == 0,
              "kfn_cost_volume: bad shape N=%d H=%d W=%d C=%d window=%d", N, H, W, C, window);
  long total = (long)N * H * W * window * window * (C / 4);
  long blocks = (total + 255) / 256;
  if (blocks > 256L * 64) blocks = 256L * 64;
  hipLaunchKernelGGL(cost_volume_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     f1, f2, vol, N, H, W, C, window);
  KFN_LAUNCH_CHECK("cost_volume_kernel");
  return KFN_OK;
}

extern "C" int kfn_pad_nhwc(const float* x, float* y, int N, int H, int W, int C, int pad, void* stream) {
  KFN_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && pad >= 0, "kfn_pad_nhwc: bad argument");
  KFN_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0, "kfn_pad_nhwc: misaligned buffer");
  long total = (long)N * (H + 2 * pad) * (W + 2 * pad) * (C / 4);
  long blocks = (total + 255) / 256;
  if (blocks > 256L * 32) blocks = 256L * 32;
  hipLaunchKernelGGL(pad_nhwc_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, y, N, H, W, C, pad);
  KFN_LAUNCH_CHECK("pad_nhwc_kernel");
  return KFN_OK;
}

extern "C" int kfn_cost_volume_gather(const float* T, const float* Gp, float* y, int N, int H, int W, int C,
                                      int ldy, int relu, void* stream) {
  KFN_REQUIRE(T && Gp && y, "kfn_cost_volume_gather: null argument");
  KFN_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && ldy >= C && ldy % 4 == 0,
              "kfn_cost_volume_gather: bad shape N=%d H=%d W=%d C=%d ldy=%d", N, H, W, C, ldy);
  KFN_REQUIRE(((reinterpret_cast<uintptr_t>(T) | reinterpret_cast<uintptr_t>(Gp) | reinterpret_cast<uintptr_t>(y)) & 15) == 0,
              "kfn_cost_volume_gather: misaligned buffer");
  KFN_REQUIRE((long)N * H * W < (1L << 31), "kfn_cost_volume_gather: too many pixels");
  long blocks = (long)N * H * W;
  if (blocks > 256L * 256) blocks = 256L * 256;
  hipLaunchKernelGGL(cost_volume_gather_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, T, Gp,
                     y, N, H, W, C, ldy, relu);
  KFN_LAUNCH_CHECK("cost_volume_gather_kernel");
  return KFN_OK;
}

extern "C" int kfn_flow_softargmax(const float* logits, float* flow_xy, float* prob, int P,
                                   int window, void* stream) {
  KFN_REQUIRE(logits && flow_xy, "kfn_flow_softargmax: null argument");
  KFN_REQUIRE(P > 0 && window > 0 && window % 2 == 0, "kfn_flow_softargmax: bad shape");
  int blocks = kfn::ceil_div(P, 4);
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(flow_softargmax_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, logits,
                     flow_xy, prob, P, window);
  KFN_LAUNCH_CHECK("flow_softargmax_kernel");
  return KFN_OK;
}

extern "C" int kfn_flow_head(const float* x, const float* w, const float* bias, float* flow_xy,
                             float* opt_logits, int P, int C, void* stream) {
  KFN_REQUIRE(x && w && flow_xy, "kfn_flow_head: null argument");
  KFN_REQUIRE(P > 0 && C > 0 && C % 4 == 0 && C <= 32, "kfn_flow_head: bad shape P=%d C=%d (C%%4==0, C<=32)", P, C);
  KFN_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0, "kfn_flow_head: x must be 16-byte aligned");
  int blocks = kfn::ceil_div(P, 4);
  if (blocks > 256 * 8) blocks = 256 * 8;
  hipLaunchKernelGGL(flow_head_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, w, bias, flow_xy,
                     opt_logits, P, C);
  KFN_LAUNCH_CHECK("flow_head_kernel");
  return KFN_OK;
}

extern "C" int kfn_copy_channels(const float* src, int ld_src, float* dst, int ld_dst, int P, int C,
                                 void* stream) {
  KFN_REQUIRE(src && dst && P > 0 && C > 0 && ld_src >= C && ld_dst >= C, "kfn_copy_channels: bad argument");
  long total = (long)P * C;
  long blocks = (total + 255) / 256;
  if (blocks > 256L * 32) blocks = 256L * 32;
  hipLaunchKernelGGL(copy_channels_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src,
                     ld_src, dst, ld_dst, P, C);
  KFN_LAUNCH_CHECK("copy_channels_kernel");
  return KFN_OK;
}
