// kfn_metrics.hip -- the per-frame evaluation numbers of KFNet/eval.py on the device.
//
// eval.py prints for every step the three CoordLossWithUncertainty(downsample=True) results
// (KFNet/KFNet.py:192-232, wired in KF_fusion, KFNet/train.py:252-257), the median distance errors
// (KFNet/eval.py:17-29,106-111) and the share of NIS values inside (0.0157, 2.706) (eval.py:10-15).
// One workgroup per frame reduces everything that is a SUM or a COUNT over the h x w grid from the
// buffers the scan already produced (measurement, raw prediction, raw KF estimate, emitted record,
// NIS) and the nearest-down-sampled label maps; the per-pixel distance maps are written out for the
// host, which only takes the medians (np.median of the positive entries, eval.py:28-29).
//
// Quirks of the reference kept on purpose:
//   * loss_map = min(3 log(sigma) + d2 / (2 sigma^2), -2)          (KFNet.py:215-216: capped from ABOVE)
//   * the ground truth fed to the losses is the PAIR batch [2,h,w,.] while the prediction is [1,h,w,.]:
//     both label maps are compared with the same prediction and valid_pixel = sum(mask_a) + sum(mask_b) + 1
//   * accuracy = (valid_pixel - count_nonzero(max(mask * d2 - thr^2, 0))) / valid_pixel
//   * the distance errors use the SECOND label of the pair and, on a reset step, the measurement in place
//     of the prediction; the KF distance uses the emitted (NIS-gated, reset-overridden) coordinates.
#include "kfn_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MT = 256;      // threads per frame
constexpr int NSTAT = 9;     // reduced quantities

struct MetricArgs {
  const f32x4* meas;     // [T,HW] (z, sigma_z) raw
  const f32x4* temp;     // [T,HW] (x^-, sigma^-) raw, as the graph computes it
  const f32x4* kf;       // [T,HW] (x, sigma) raw KF estimate, ungated
  const f32x4* rec;      // [T,HW] emitted record (T.x, 1/sigma)
  const float* nis;      // [T,HW,3]
  const f32x4* labels;   // [L,HW] (gt xyz, mask) already nearest-down-sampled (src = floor(dst*8))
  const int* pair;       // [T,2] label indices (a, b) of the step's frame pair
  const unsigned char* reset;  // [T] 1 = the host re-initialised the filter on this step
  float* stats;          // [T,16]
  float* dist;           // [T,3,HW] cm, 0 where masked
  int T, HW;
  float thr2, min_unc;
  int has_transform;
  float M[12];
};

__device__ __forceinline__ void xform(const MetricArgs& a, const f32x4& v, float* o) {
  if (a.has_transform) {   // ApplyTransform (KFNet/util.py:12-40), same operation order as the scan
    o[0] = ((a.M[0] * v.x + a.M[1] * v.y) + a.M[2] * v.z) + a.M[3];
    o[1] = ((a.M[4] * v.x + a.M[5] * v.y) + a.M[6] * v.z) + a.M[7];
    o[2] = ((a.M[8] * v.x + a.M[9] * v.y) + a.M[10] * v.z) + a.M[11];
  } else {
    o[0] = v.x; o[1] = v.y; o[2] = v.z;
  }
}

__device__ __forceinline__ float d2(const float* c, const f32x4& g) {
  const float dx = c[0] - g.x, dy = c[1] - g.y, dz = c[2] - g.z;
  return (dx * dx + dy * dy) + dz * dz;
}

// CoordLossWithUncertainty terms of one pixel against one label map: adds the masked loss and the
// "inaccurate" indicator
__device__ __forceinline__ void loss_terms(const MetricArgs& a, const float* c, float sigma, const f32x4& g,
                                           float& loss_sum, float& inacc) {
  const float m = (g.w == 1.0f) ? 1.0f : 0.0f;             // tf.cast(tf.equal(mask, 1.0))
  const float dd = d2(c, g);
  const float u = fmaxf(sigma, a.min_unc);
  float lm = 3.0f * logf(u) + dd / (2.0f * (u * u));
  lm = fminf(lm, -2.0f);
  loss_sum += m * lm;
  inacc += (fmaxf(m * dd - a.thr2, 0.0f) != 0.0f) ? 1.0f : 0.0f;
}

__global__ __launch_bounds__(MT) void eval_metrics_kernel(MetricArgs a) {
  __shared__ float red[MT / 64][NSTAT];
  const int t = blockIdx.x;
  const int tid = threadIdx.x;
  const size_t off = (size_t)t * a.HW;
  const f32x4* la = a.labels + (size_t)a.pair[2 * t] * a.HW;
  const f32x4* lb = a.labels + (size_t)a.pair[2 * t + 1] * a.HW;
  const bool reset = a.reset[t] != 0;
  float s[NSTAT];
#pragma unroll
  for (int i = 0; i < NSTAT; ++i) s[i] = 0.f;
  for (int p = tid; p < a.HW; p += MT) {
    const f32x4 z = a.meas[off + p], tp = a.temp[off + p], kf = a.kf[off + p], rc = a.rec[off + p];
    const f32x4 ga = la[p], gb = lb[p];
    float cz[3], ct[3], ck[3];
    xform(a, z, cz);
    xform(a, tp, ct);
    xform(a, kf, ck);
    // losses / accuracies against BOTH labels of the pair (broadcast quirk)
    loss_terms(a, cz, z.w, ga, s[0], s[3]);
    loss_terms(a, cz, z.w, gb, s[0], s[3]);
    loss_terms(a, ct, tp.w, ga, s[1], s[4]);
    loss_terms(a, ct, tp.w, gb, s[1], s[4]);
    loss_terms(a, ck, kf.w, ga, s[2], s[5]);
    loss_terms(a, ck, kf.w, gb, s[2], s[5]);
    s[6] += ((ga.w == 1.0f) ? 1.0f : 0.0f) + ((gb.w == 1.0f) ? 1.0f : 0.0f);
    // NIS band (eval.py:10-15) over the 3 channels
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v = a.nis[(off + p) * 3 + c];
      s[7] += (v > 0.0f) ? 1.0f : 0.0f;
      s[8] += (v > 0.0157f && v < 2.706f) ? 1.0f : 0.0f;
    }
    // distance maps against the second label (eval.py:106-108), raw mask value as multiplier (eval.py:26)
    const float* ctd = reset ? cz : ct;                      // eval.py:98: temp output := measurement on a reset
    const float crec[3] = {rc.x, rc.y, rc.z};               // already transformed / gated / reset-overridden
    a.dist[((size_t)t * 3 + 0) * a.HW + p] = sqrtf(d2(cz, gb)) * gb.w * 100.0f;
    a.dist[((size_t)t * 3 + 1) * a.HW + p] = sqrtf(d2(ctd, gb)) * gb.w * 100.0f;
    a.dist[((size_t)t * 3 + 2) * a.HW + p] = sqrtf(d2(crec, gb)) * gb.w * 100.0f;
  }
  // deterministic block reduction: wave shuffles, then the 4 wave partials in wave order
#pragma unroll
  for (int i = 0; i < NSTAT; ++i) {
    float v = s[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((tid & 63) == 0) red[tid >> 6][i] = v;
  }
  __syncthreads();
  if (tid < 16) {
    float v = 0.f;
    if (tid < NSTAT) {
      v = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
      if (tid == 6) v += 1.0f;                               // valid_pixel = reduce_sum(mask) + 1.
    }
    a.stats[(size_t)t * 16 + tid] = v;
  }
}

}  // namespace

extern "C" int kfn_eval_metrics(const float* meas, const float* temp, const float* kf_raw, const float* records,
                                const float* nis, const float* labels, const int32_t* label_pair,
                                const uint8_t* reset_flags, const float* transform12, int T, int HW,
                                float dist_threshold, float min_uncertainty, float* stats, float* dist_maps,
                                void* stream) {
  KFN_REQUIRE(meas && temp && kf_raw && records && nis && labels && label_pair && reset_flags && stats && dist_maps,
              "kfn_eval_metrics: null argument");
  KFN_REQUIRE(T > 0 && HW > 0, "kfn_eval_metrics: bad shape T=%d HW=%d", T, HW);
  KFN_REQUIRE(((reinterpret_cast<uintptr_t>(meas) | reinterpret_cast<uintptr_t>(temp) | reinterpret_cast<uintptr_t>(kf_raw) |
                reinterpret_cast<uintptr_t>(records) | reinterpret_cast<uintptr_t>(labels)) & 15) == 0,
              "kfn_eval_metrics: misaligned buffer");
  MetricArgs a;
  a.meas = reinterpret_cast<const f32x4*>(meas);
  a.temp = reinterpret_cast<const f32x4*>(temp);
  a.kf = reinterpret_cast<const f32x4*>(kf_raw);
  a.rec = reinterpret_cast<const f32x4*>(records);
  a.nis = nis;
  a.labels = reinterpret_cast<const f32x4*>(labels);
  a.pair = label_pair;
  a.reset = reset_flags;
  a.stats = stats;
  a.dist = dist_maps;
  a.T = T; a.HW = HW;
  a.thr2 = (float)((double)dist_threshold * (double)dist_threshold);   // the reference squares in Python (double)
  a.min_unc = min_uncertainty;
  a.has_transform = transform12 != nullptr;
  for (int i = 0; i < 12; ++i) a.M[i] = transform12 ? transform12[i] : 0.f;
  hipLaunchKernelGGL(eval_metrics_kernel, dim3((unsigned)T), dim3(MT), 0, (hipStream_t)stream, a);
  KFN_LAUNCH_CHECK("eval_metrics_kernel");
  return KFN_OK;
}
