// kfn_util_ops.hip -- the reference's graph-level helpers as stand-alone launches.
//
// On eval.py's path these three are fused into kalman_scan_kernel (kfn_kalman.hip: fuse_pixel); a script that calls them at
// Python level (KFNet/eval.py:57-59 calls ApplyTransform itself) gets them here with the same arithmetic:
//   kfn_apply_transform   KFNet/util.py:12-40   x' = (T [x;1])[0:3], no perspective divide
//   kfn_pixel_map         KFNet/util.py:42-63   map[b,y,x] = (x, y)  (optionally ((x-u)/fx, (y-v)/fy))
//   kfn_bilinear_sampler  tools/util.py:3-94    clamped corners AND weights from the clamped corners, add_n order
// Built with -ffp-contract=off (kfnet_amd/build.py): products and sums are rounded one by one like TF's elementwise ops.
#include "kfn_common.h"

namespace {

__global__ void apply_transform_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ Tm, long t_stride,
                                       long px_per_batch, float* __restrict__ y, int ldy, long P) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const float* M = Tm + (p / px_per_batch) * t_stride;      // [3 or 4][4] row-major, rows 0-2 used
  const float a = x[p * ldx], b = x[p * ldx + 1], c = x[p * ldx + 2];
  y[p * ldy + 0] = ((M[0] * a + M[1] * b) + M[2] * c) + M[3];
  y[p * ldy + 1] = ((M[4] * a + M[5] * b) + M[6] * c) + M[7];
  y[p * ldy + 2] = ((M[8] * a + M[9] * b) + M[10] * c) + M[11];
}

__global__ void pixel_map_kernel(float* __restrict__ y, int ldy, int H, int W, long P, int normalize, float u, float v,
                                 float fx, float fy) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const long q = p % ((long)H * W);
  float mx = (float)(q % W), my = (float)(q / W);
  if (normalize) {
    mx = (mx - u) / fx;
    my = (my - v) / fy;
  }
  y[p * ldy] = mx;
  y[p * ldy + 1] = my;
}

// one thread per (output pixel, channel)
__global__ void bilinear_sampler_kernel(const float* __restrict__ img, int ldi, int Hs, int Ws, int C,
                                        const float* __restrict__ co, int ldc, float* __restrict__ out, int ldo,
                                        long px_per_batch, long P) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P * C) return;
  const long p = i / C;
  const int c = (int)(i - p * C);
  const long b = p / px_per_batch;
  const float px = co[p * ldc], py = co[p * ldc + 1];
  const float xmax = (float)(Ws - 1), ymax = (float)(Hs - 1);
  const float x0 = floorf(px), x1 = x0 + 1.0f;
  const float y0 = floorf(py), y1 = y0 + 1.0f;
  const float x0s = fminf(fmaxf(x0, 0.f), xmax), x1s = fminf(fmaxf(x1, 0.f), xmax);
  const float y0s = fminf(fmaxf(y0, 0.f), ymax), y1s = fminf(fmaxf(y1, 0.f), ymax);
  const float wx0 = x1s - px, wx1 = px - x0s;
  const float wy0 = y1s - py, wy1 = py - y0s;
  const long base = b * Hs * Ws;
  const long ix0 = (long)x0s, ix1 = (long)x1s, iy0 = (long)y0s, iy1 = (long)y1s;
  const float im00 = img[(base + iy0 * Ws + ix0) * ldi + c];
  const float im01 = img[(base + iy1 * Ws + ix0) * ldi + c];
  const float im10 = img[(base + iy0 * Ws + ix1) * ldi + c];
  const float im11 = img[(base + iy1 * Ws + ix1) * ldi + c];
  const float w00 = wx0 * wy0, w01 = wx0 * wy1, w10 = wx1 * wy0, w11 = wx1 * wy1;
  out[p * ldo + c] = ((w00 * im00 + w01 * im01) + w10 * im10) + w11 * im11;   // tf.add_n order
}

}  // namespace

extern "C" int kfn_apply_transform(const float* coords, int ld_in, const float* transform, int per_batch, int B, int H, int W,
                                   float* out, int ld_out, void* stream) {
  KFN_REQUIRE(coords && transform && out && B > 0 && H > 0 && W > 0 && ld_in >= 3 && ld_out >= 3, "kfn_apply_transform: bad argument");
  const long P = (long)B * H * W;
  hipLaunchKernelGGL(apply_transform_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, coords, ld_in,
                     transform, per_batch ? 16L : 0L, (long)H * W, out, ld_out, P);
  return kfn::check_hip(hipGetLastError(), "kfn_apply_transform launch");
}

extern "C" int kfn_pixel_map(float* out, int ld_out, int B, int H, int W, int normalize, float u, float v, float focal_x,
                             float focal_y, void* stream) {
  KFN_REQUIRE(out && B > 0 && H > 0 && W > 0 && ld_out >= 2, "kfn_pixel_map: bad argument");
  KFN_REQUIRE(!normalize || (focal_x != 0.f && focal_y != 0.f), "kfn_pixel_map: normalize needs non-zero focal lengths");
  const long P = (long)B * H * W;
  hipLaunchKernelGGL(pixel_map_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, ld_out, H, W, P,
                     normalize, u, v, focal_x, focal_y);
  return kfn::check_hip(hipGetLastError(), "kfn_pixel_map launch");
}

extern "C" int kfn_bilinear_sampler(const float* imgs, int ld_img, int B, int Hs, int Ws, int C, const float* coords, int ld_coords,
                                    int Ht, int Wt, float* out, int ld_out, void* stream) {
  KFN_REQUIRE(imgs && coords && out && B > 0 && Hs > 0 && Ws > 0 && C > 0 && Ht > 0 && Wt > 0, "kfn_bilinear_sampler: bad argument");
  KFN_REQUIRE(ld_img >= C && ld_out >= C && ld_coords >= 2, "kfn_bilinear_sampler: pixel strides smaller than the channel counts");
  const long P = (long)B * Ht * Wt;
  const long n = P * C;
  hipLaunchKernelGGL(bilinear_sampler_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, imgs, ld_img, Hs, Ws,
                     C, coords, ld_coords, out, ld_out, (long)Ht * Wt, P);
  return kfn::check_hip(hipGetLastError(), "kfn_bilinear_sampler launch");
}
