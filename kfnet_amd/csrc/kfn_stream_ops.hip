// kfn_stream_ops.hip -- the HBM-bound kernels of the path: uint8 image ingest + first
// convolutions (Cin = 3), local cost volume, softmax/soft-argmax flow head, channel copy.
#include "kfn_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------
// uint8 image -> (x-128)*0.00625 -> 3x3 SAME conv (Cin = 3) + bias + ReLU, two heads.
// SCoordNet.preprocess+conv1a (cnn_wrapper/SCoordNet.py:20-21,34-37) and the feature
// tower's preprocess+feat1 (KFNet/KFNet.py:317-320) read the image ONCE.
// Block = 64x4 output pixels; the 66x6x3 preprocessed halo lives in LDS (zero padding is
// applied in the preprocessed domain, as TF pads the already-normalised tensor).
// Each thread owns one pixel: its 27 inputs sit in registers, weights are wave-uniform
// (scalar loads), outputs are produced 16 channels at a time.
// ------------------------------------------------------------------------------------
constexpr int FT_W = 64, FT_H = 4;

__global__ __launch_bounds__(256) void first_conv_kernel(
    const uint8_t* __restrict__ img, int N, int H, int W,
    const float* __restrict__ w1, const float* __restrict__ b1, float* __restrict__ y1, int C1,
    const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ y2, int C2) {
  __shared__ float tile[(FT_H + 2) * (FT_W + 2) * 3];
  const int tid = threadIdx.x;
  const int x0 = blockIdx.x * FT_W, y0 = blockIdx.y * FT_H, n = blockIdx.z;
  const uint8_t* src = img + (size_t)n * H * W * 3;
  constexpr int TW3 = (FT_W + 2) * 3;
  for (int i = tid; i < (FT_H + 2) * TW3; i += 256) {
    int yy = i / TW3, rem = i - yy * TW3;
    int xx = rem / 3, c = rem - xx * 3;
    int gy = y0 + yy - 1, gx = x0 + xx - 1;
    float v = 0.f;
    if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
      v = ((float)src[((size_t)gy * W + gx) * 3 + c] - 128.0f) * 0.00625f;
    tile[i] = v;
  }
  __syncthreads();
  const int tx = tid & (FT_W - 1), ty = tid / FT_W;
  const int gx = x0 + tx, gy = y0 + ty;
  float xin[27];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int c = 0; c < 3; ++c) xin[(ky * 3 + kx) * 3 + c] = tile[(ty + ky) * TW3 + (tx + kx) * 3 + c];
  if (gx >= W || gy >= H) return;
  const size_t pix = ((size_t)n * H + gy) * W + gx;

  auto head = [&](const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ y, int C) {
    for (int g = 0; g < C; g += 16) {
      float acc[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) acc[c] = b ? b[g + c] : 0.f;
#pragma unroll
      for (int k = 0; k < 27; ++k)
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[c] = fmaf(xin[k], w[k * C + g + c], acc[c]);
      float* dst = y + pix * C + g;
#pragma unroll
      for (int c = 0; c < 16; c += 4) {
        f32x4 v = {fmaxf(acc[c], 0.f), fmaxf(acc[c + 1], 0.f), fmaxf(acc[c + 2], 0.f), fmaxf(acc[c + 3], 0.f)};
        *reinterpret_cast<f32x4*>(dst + c) = v;
      }
    }
  };
  head(w1, b1, y1, C1);
  if (C2 > 0) head(w2, b2, y2, C2);
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

extern "C" int kfn_first_conv_u8(const uint8_t* img, int N, int H, int W, const float* w1,
                                 const float* b1, float* y1, int C1, const float* w2,
                                 const float* b2, float* y2, int C2, void* stream) {
  KFN_REQUIRE(img && w1 && y1, "kfn_first_conv_u8: null argument");
  KFN_REQUIRE(N > 0 && H > 0 && W > 0, "kfn_first_conv_u8: bad shape");
  KFN_REQUIRE(C1 > 0 && C1 % 16 == 0 && C2 >= 0 && C2 % 16 == 0,
              "kfn_first_conv_u8: C1=%d C2=%d must be multiples of 16", C1, C2);
  KFN_REQUIRE(C2 == 0 || (w2 && y2), "kfn_first_conv_u8: second head needs w2/y2");
  dim3 grid(kfn::ceil_div(W, FT_W), kfn::ceil_div(H, FT_H), N), block(256);
  hipLaunchKernelGGL(first_conv_kernel, grid, block, 0, (hipStream_t)stream, img, N, H, W, w1, b1,
                     y1, C1, w2, b2, y2, C2);
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
