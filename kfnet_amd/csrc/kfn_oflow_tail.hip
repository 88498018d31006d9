// kfn_oflow_tail.hip -- the tail of OFlowNet for one pixel's 8x8 window in one launch:
//   conv6 (3x3, 48 -> 16, ReLU) on concat0 = [upconv0 | conv0]      cnn_wrapper/OFlowNet.py:36-40
//   prediction (3x3, 16 -> 1, linear)                                cnn_wrapper/OFlowNet.py:41
//   softmax over the 64 window cells                                 cnn_wrapper/OFlowNet.py:45-47
//   soft-argmax flow                                                 KFNet/KFNet.py:381-385
//
// As separate launches conv6 ran on the generic implicit-GEMM kernel with 16 output channels: every input element
// went through the L1 nine times (8 GB fetched for 1 GB of input, 0.5 of the MFMA peak) and its 334 MB output made
// a round trip through HBM to the flow head.  Here ONE WAVE owns a window: the 8x8x48 patch is staged once into an
// LDS image with a zero border (the conv's SAME padding; 10x10 cells of 48+4 floats), the nine taps' A fragments
// are 16-byte LDS reads from it, conv6's 27 KiB of weights live in 108 registers per lane for the whole kernel
// (v_mfma_f32_16x16x4_f32: lane (i = l%16, kq = l/16) supplies A[i][kq] and B[kq][i]; K is ordered so that a lane's
// 12 k-steps of a tap are 12 CONSECUTIVE channels kq*12 .. kq*12+11 -> three ds_read_b128 feed 12 MFMAs).  conv6's
// output goes to a second LDS image, the prediction conv is 144 FMAs per lane (lane = window cell, wave-uniform
// weights), then max / sum / weighted sums by wave shuffles.  The next window's patch is fetched into registers
// while the MFMAs of the current one run.  No VALU work inside the MFMA phase.
#include "kfn_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int C1 = 48;            // channels of concat0
constexpr int C2 = 16;            // channels of conv6
constexpr int LD1 = C1 + 4;       // floats per cell of the padded input image
constexpr int LD2 = C2 + 4;       // floats per cell of the padded conv6 image
constexpr int T1_BYTES = 100 * LD1 * 4;
constexpr int T2_BYTES = 100 * LD2 * 4;
constexpr int WAVE_BYTES = T1_BYTES + T2_BYTES;

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

__global__ __launch_bounds__(256, 1) void oflow_tail_kernel(const float* __restrict__ x, const float* __restrict__ w6p,
                                                            const float* __restrict__ b6, const float* __restrict__ wp,
                                                            const float* __restrict__ bp, float* __restrict__ flow,
                                                            float* __restrict__ logits_out, int P) {
  extern __shared__ __attribute__((aligned(16))) char smem_ot[];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  char* const t1 = smem_ot + wv * WAVE_BYTES;
  char* const t2 = t1 + T1_BYTES;
  // zero both images once: the borders stay zero, the interiors are rewritten per window
  for (int i = lane; i < WAVE_BYTES / 16; i += 64) reinterpret_cast<f32x4*>(t1)[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // conv6 weights: fragment t = tap*12 + j of this lane = w6[tap][kq*12 + j][n]  (host-packed [108][64])
  float wreg[108];
#pragma unroll
  for (int t = 0; t < 108; ++t) wreg[t] = w6p[t * 64 + lane];
  const int li = lane & 15, kq = lane >> 4;
  const float bias6 = b6 ? b6[li] : 0.f;      // accumulator column = lane & 15
  const float bias_p = bp ? bp[0] : 0.f;

  // staging: float4 k = it*64 + lane of the window (cell = k / 12, quad = k % 12) -> padded cell (cy+1, cx+1)
  int st_off[12];
#pragma unroll
  for (int it = 0; it < 12; ++it) {
    const int k = it * 64 + lane, cell = k / 12, q = k - cell * 12;
    st_off[it] = (((cell >> 3) + 1) * 10 + (cell & 7) + 1) * (LD1 * 4) + q * 16;
  }
  // A fragments: row i of M-block mb is window cell mb*16 + i; tap (ky,kx) reads padded cell (cy+ky, cx+kx)
  const int a_base = ((li >> 3) * 10 + (li & 7)) * (LD1 * 4) + kq * 48;
  // conv6 output: accumulator element e of M-block mb is cell mb*16 + 4*kq + e, channel li
  const int o_base = (((4 * kq) >> 3) + 1) * 10 * (LD2 * 4) + (((4 * kq) & 7) + 1) * (LD2 * 4) + li * 4;
  // prediction: lane = window cell (ci, cj)
  const int ci = lane >> 3, cj = lane & 7;
  const int p_base = (ci * 10 + cj) * (LD2 * 4);

  f32x4 stg[12];
  int p = blockIdx.x * 4 + wv;
  const int pstride = gridDim.x * 4;
  if (p < P) {
    const f32x4* src = reinterpret_cast<const f32x4*>(x + (size_t)p * 64 * C1);
#pragma unroll
    for (int it = 0; it < 12; ++it) stg[it] = src[it * 64 + lane];
  }
  for (; p < P; p += pstride) {
    // ---- the window into LDS, the next one into registers -------------------------------------
#pragma unroll
    for (int it = 0; it < 12; ++it) *reinterpret_cast<f32x4*>(t1 + st_off[it]) = stg[it];
    const int pn = p + pstride;
    if (pn < P) {   // wave-uniform
      const f32x4* src = reinterpret_cast<const f32x4*>(x + (size_t)pn * 64 * C1);
#pragma unroll
      for (int it = 0; it < 12; ++it) stg[it] = src[it * 64 + lane];
    }
    __builtin_amdgcn_wave_barrier();

    // ---- conv6: 4 M-blocks x 9 taps x 12 k-steps of 16x16x4, two M-blocks interleaved ----------
    f32x4 acc[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) acc[mb] = f32x4{bias6, bias6, bias6, bias6};
#pragma unroll
    for (int mp = 0; mp < 2; ++mp) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int toff = ((tap / 3) * 10 + (tap % 3)) * (LD1 * 4);
        f32x4 a[2][3];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int s = 0; s < 3; ++s)
            a[m][s] = *reinterpret_cast<const f32x4*>(t1 + a_base + (mp * 2 + m) * (2 * 10 * LD1 * 4) + toff + s * 16);
#pragma unroll
        for (int j = 0; j < 12; ++j)
#pragma unroll
          for (int m = 0; m < 2; ++m)
            acc[mp * 2 + m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][j >> 2][j & 3], wreg[tap * 12 + j], acc[mp * 2 + m], 0, 0, 0);
      }
    }
    // ---- ReLU, conv6 image -------------------------------------------------------------------
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        // cell = mb*16 + 4*kq + e: row += 2*mb, column += e (4*kq is 0 or 4 mod 8: e never wraps the row)
        *reinterpret_cast<float*>(t2 + o_base + (mb * 2 * 10 + e) * (LD2 * 4)) = fmaxf(acc[mb][e], 0.f);
      }
    __builtin_amdgcn_wave_barrier();

    // ---- prediction conv: one window cell per lane, wave-uniform weights -------------------------
    float lg = bias_p;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const char* cellp = t2 + p_base + ((tap / 3) * 10 + (tap % 3)) * (LD2 * 4);
      const float* wt = wp + tap * C2;
#pragma unroll
      for (int q = 0; q < C2 / 4; ++q) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(cellp + q * 16);
        lg = fmaf(xv.x, wt[q * 4 + 0], lg);
        lg = fmaf(xv.y, wt[q * 4 + 1], lg);
        lg = fmaf(xv.z, wt[q * 4 + 2], lg);
        lg = fmaf(xv.w, wt[q * 4 + 3], lg);
      }
    }
    if (logits_out) logits_out[(size_t)p * 64 + lane] = lg;
    // ---- softmax over the 64 cells + soft-argmax (same operation order as flow_head_kernel) -----------
    const float mx = wave_max(lg);
    const float ex = expf(lg - mx);
    const float se = wave_sum(ex);
    const float pr = ex / se;
    const float sx = wave_sum(pr * (float)(cj - 4));
    const float sy = wave_sum(pr * (float)(ci - 4));
    if (lane == 0) {
      flow[(size_t)p * 2 + 0] = sx;
      flow[(size_t)p * 2 + 1] = sy;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace

extern "C" int kfn_oflow_tail(const float* x, const float* w6_packed, const float* b6, const float* wp, const float* bp,
                              float* flow_xy, float* opt_logits, int P, int c_in, int c_mid, void* stream) {
  KFN_REQUIRE(x && w6_packed && wp && flow_xy, "kfn_oflow_tail: null argument");
  KFN_REQUIRE(P > 0, "kfn_oflow_tail: bad window count %d", P);
  if (c_in != C1 || c_mid != C2)
    return kfn::fail(KFN_ERR_UNSUPPORTED, "kfn_oflow_tail: only the %d -> %d -> 1 tail of OFlowNet (got %d -> %d)", C1, C2,
                     c_in, c_mid);
  KFN_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0, "kfn_oflow_tail: x must be 16-byte aligned");
  static std::atomic<uint64_t> attr_done{0};
  {
    int rc = kfn::set_max_dynamic_lds(reinterpret_cast<const void*>(oflow_tail_kernel), 4 * WAVE_BYTES, attr_done);
    if (rc != KFN_OK) return rc;
  }
  int blocks = kfn::ceil_div(P, 4);
  if (blocks > 256) blocks = 256;      // one workgroup of four waves per CU, each wave walks its windows
  hipLaunchKernelGGL(oflow_tail_kernel, dim3(blocks), dim3(256), 4 * WAVE_BYTES, (hipStream_t)stream, x, w6_packed, b6, wp,
                     bp, flow_xy, opt_logits, P);
  KFN_LAUNCH_CHECK("oflow_tail_kernel");
  return KFN_OK;
}
