// kfn_kalman.hip -- the recurrent part of KFNet as one persistent scan kernel.
//
// Replaces, per frame and pixel (compiled with -ffp-contract=off so every product/sum is
// rounded exactly like the reference's unfused TF elementwise ops):
//   KFNet.BuildOFlowNet tail   KFNet/KFNet.py:386-401  (pixel_map, 2x bilinear_sampler,
//                                                       variance clamp + propagate)
//   tools.util.bilinear_sampler tools/util.py:36-93    (clamped corners + clamped weights)
//   KFNet.BuildKFCoord         KFNet/KFNet.py:148-162
//   KFNet.GetNIS               KFNet/KFNet.py:164-184
//   eval.py loop body          KFNet/eval.py:87-126    (reset, NIS output gate, state
//                                                       feedback, ApplyTransform, 1/sigma)
//
// One workgroup (768 threads = 12 wavefronts at 60x80) per sequence.  The [H*W] x (x,y,z,sigma)
// state lives in LDS as float4 (76.8 KB at 60x80) for the whole scan, so the 4-tap warp
// gather never touches HBM; per frame the kernel streams 28 B/px of inputs (flow 8,
// sigma_trans 4, measurement 16) and 16 B/px of records from/to HBM, and the inputs of
// frame t+1 are already in flight while frame t is being fused (register prefetch).
// The state is double-buffered in LDS where two copies fit (one barrier per frame), single-buffered otherwise.
//
// Round 5 (VERDICT r4 Next #8) -- ROLLING prefetch: a pixel's input registers are dead the moment the pixel is fused, so
// the loads of the SAME pixel of frame t+1 are issued into them right there: no second register set (35 input registers
// per thread instead of 98), the 21 loads a thread keeps in flight are spread over the frame instead of bursting at
// its start, and the workgroup can be 1024 threads (16 waves, 128-VGPR budget) where the burst form needed 768 x 7 pixels at
// 168.  Records leave as non-temporal 16-byte stores and the once-read inputs arrive as non-temporal loads (neither is
// touched again by this launch; the [h,w,4] state never leaves LDS).
#include "kfn_common.h"

// 1 = the scan kernel's buffer descriptors are forced into SGPRs (v_readfirstlane), 0 = as hipcc makes them (waterfall loops)
#ifndef KFN_SCAN_UNIFORM_SRD
#define KFN_SCAN_UNIFORM_SRD 1
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));


struct KalmanArgs {
  const f32x2* flow;
  const float* sigma_t;
  const f32x4* meas;
  f32x4* state;
  f32x4* rec;
  f32x4* opt_temp;
  float* opt_nis;
  f32x4* opt_kf;      // raw (untransformed, ungated) KF estimate per frame, or null
  int raw_on_reset;   // debug outputs of a reset frame = what the reference GRAPH computes there
  kfn_kalman_desc d;
};

struct PixIn {
  f32x2 flow;
  float st;
  f32x4 z;
};

// streamed-once data: the nt bit keeps it from displacing what other kernels of the step left in L2 / MALL
template <bool NT, typename V>
__device__ __forceinline__ V ld_stream(const V* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}
template <bool NT, typename V>
__device__ __forceinline__ void st_stream(V* p, const V& v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// One pixel of one frame: process model (warp + variance propagation), Kalman update, NIS,
// output record.  `st` is the previous state of this sequence (LDS in the scan kernel,
// global memory in the per-frame kernel); returns the new state of pixel p.
// DBG = false compiles the optional outputs (opt_temp / opt_nis / opt_kf) out: besides the stores themselves this removes
// every VARIABLE number of vector-memory operations between a prefetch load and its use -- with an optional store in a
// branch the compiler's s_waitcnt insertion must assume the branch not taken and degrades every wait to vmcnt(0), which
// serialises the rolling prefetch (seen in the listing: vmcnt(0) in front of every pixel).  `valid`: this thread's pixel
// exists (p < HW); invalid lanes compute on a clamped duplicate and store nothing.
// (x, y) = the pixel's grid position (p = y W + x); rec_out != nullptr: the record is handed back instead of stored
// (the scan kernel stores it through a buffer descriptor with scalar frame / slot offsets).
// sqrt and quotient, CORRECTLY ROUNDED like sqrtf / '/' at -ffp-contract=off, for operands in the normal range: the refinement steps
// hipcc emits for the IEEE forms (v_sqrt + the two one-ulp neighbours tested by their residuals; v_rcp + two Newton steps on the
// reciprocal and the quotient + the final residual fma) WITHOUT the denormal pre-scaling around them (v_div_scale x 2 / v_div_fmas;
// the 2^32 scaling, its undo and the zero / infinity class test of the root): 9 + 10 instead of 16 + 12 instructions and none of
// the VCC hazards' s_nops.  The scan is VALU-bound (154 instructions per pixel, profiles/r06_kalman_scan_pmc.json); variances and
// sigmas of this model lie in [1e-10, 1e4].  Bit-identical to the IEEE forms there (tools/mb/kalman_mb: every variant against
// the IEEE build; tests/test_gpu_ops.py::test_kalman_lean_arithmetic_is_correctly_rounded); zero, infinity and NaN come out as
// IEEE's (v_div_fixup stays; sqrt(0) = 0); operands below 2^-96 may differ in the last place.
__device__ __forceinline__ float sqrt_rn_normal(float x) {
  const float s = __builtin_amdgcn_sqrtf(x);
  const float sd = __builtin_bit_cast(float, __builtin_bit_cast(int, s) - 1);
  const float su = __builtin_bit_cast(float, __builtin_bit_cast(int, s) + 1);
  const float rd = __builtin_fmaf(-sd, s, x);
  const float ru = __builtin_fmaf(-su, s, x);
  float r = rd <= 0.f ? sd : s;
  r = ru > 0.f ? su : r;
  return r;
}
__device__ __forceinline__ float div_rn_normal(float a, float b) {
  const float r0 = __builtin_amdgcn_rcpf(b);
  const float nb = -b;
  const float f0 = __builtin_fmaf(nb, r0, 1.0f);
  const float f1 = __builtin_fmaf(f0, r0, r0);
  const float m = a * f1;
  const float f2 = __builtin_fmaf(nb, m, a);
  const float f3 = __builtin_fmaf(f2, f1, m);
  const float f4 = __builtin_fmaf(nb, f3, a);
  const float q = __builtin_fmaf(f4, f1, f3);
  return __builtin_amdgcn_div_fixupf(q, b, a);
}
#ifndef KFN_KALMAN_LEAN
#define KFN_KALMAN_LEAN 1
#endif
template <bool LEAN>
__device__ __forceinline__ float k_sqrt(float x) {
  if constexpr (LEAN) return sqrt_rn_normal(x);
  else return sqrtf(x);
}
template <bool LEAN>
__device__ __forceinline__ float k_div(float a, float b) {
  if constexpr (LEAN) return div_rn_normal(a, b);
  else return a / b;
}

template <bool NT = false, bool DBG = true, bool LEAN = (KFN_KALMAN_LEAN != 0)>
__device__ __forceinline__ f32x4 fuse_pixel(const KalmanArgs& a_in, const f32x4* st, const PixIn& in, int p, int x, int y,
                                            size_t off, bool reset, int W, float xmax, float ymax,
                                            float eps2, bool want_nis, bool valid = true, f32x4* rec_out = nullptr) {
  KalmanArgs a = a_in;
  if constexpr (!DBG) { a.opt_temp = nullptr; a.opt_nis = nullptr; a.opt_kf = nullptr; }
  const f32x4 z = in.z;  // (zx, zy, zz, sigma_z)
  f32x4 outv;                // record before transform: (x, y, z, sigma)
  f32x4 nv;
  // eval.py runs the graph on EVERY step -- also on a reset step, where temp / KF / NIS are computed
  // from whatever the state variables hold and only the host overrides outputs and feedback afterwards
  // (KFNet/eval.py:94-101).  With raw_on_reset the debug outputs (temp, NIS, raw KF: what the losses of
  // eval.py:113-118 are computed from) follow the graph; the state and the record follow the host.
  const bool graph_path = !reset || (a.raw_on_reset && (a.opt_temp || a.opt_nis || a.opt_kf));
  if (reset && !graph_path && valid) {
    if (a.opt_temp) a.opt_temp[off + p] = z;
    if (a.opt_kf) a.opt_kf[off + p] = z;
    if (a.opt_nis) {
      a.opt_nis[(off + p) * 3 + 0] = 0.f;
      a.opt_nis[(off + p) * 3 + 1] = 0.f;
      a.opt_nis[(off + p) * 3 + 2] = 0.f;
    }
  }
  if (graph_path) {
    // pixel_map = GetPixelMap + flow (KFNet.py:386, util.py:42-63: (x, y))
    const float px = (float)x + in.flow.x;
    const float py = (float)y + in.flow.y;
    // bilinear_sampler (tools/util.py:36-93)
    const float x0 = floorf(px), x1 = x0 + 1.0f;
    const float y0 = floorf(py), y1 = y0 + 1.0f;
    const float x0s = fminf(fmaxf(x0, 0.f), xmax), x1s = fminf(fmaxf(x1, 0.f), xmax);
    const float y0s = fminf(fmaxf(y0, 0.f), ymax), y1s = fminf(fmaxf(y1, 0.f), ymax);
    const float wx0 = x1s - px, wx1 = px - x0s;
    const float wy0 = y1s - py, wy1 = py - y0s;
    const int ix0 = (int)x0s, ix1 = (int)x1s, iy0 = (int)y0s, iy1 = (int)y1s;
    const f32x4 im00 = st[iy0 * W + ix0];
    const f32x4 im01 = st[iy1 * W + ix0];
    const f32x4 im10 = st[iy0 * W + ix1];
    const f32x4 im11 = st[iy1 * W + ix1];
    const float w00 = wx0 * wy0, w01 = wx0 * wy1, w10 = wx1 * wy0, w11 = wx1 * wy1;
    const f32x4 g = ((w00 * im00 + w01 * im01) + w10 * im10) + w11 * im11;  // add_n order
    // variance propagation (KFNet.py:393-401)
    const float last_var = fmaxf(g.w * g.w, eps2);
    const float trans_var = fmaxf(in.st * in.st, eps2);
    const float temp_unc = k_sqrt<LEAN>(trans_var + last_var);
    // BuildKFCoord (KFNet.py:148-162) -- note last_variance = square(sqrt(.))
    const float lv = temp_unc * temp_unc;
    const float mv = z.w * z.w;
    const float K = k_div<LEAN>(lv, lv + mv);
    const float om = fmaxf(1.0f - K, 0.0f);
    f32x4 kf;
    kf.x = om * g.x + K * z.x;
    kf.y = om * g.y + K * z.y;
    kf.z = om * g.z + K * z.z;
    kf.w = k_sqrt<LEAN>(om * lv);
    nv = kf;
    outv = kf;  // eval.py:103-104: the raw KF state (nv) is what is fed back
    if (a.opt_kf && valid) a.opt_kf[off + p] = kf;
    if (want_nis) {
      // GetNIS (KFNet.py:164-184)
      const float iu = k_sqrt<LEAN>(temp_unc * temp_unc + z.w * z.w);
      const float iv = iu * iu;
      const float d0 = z.x - g.x, d1 = z.y - g.y, d2 = z.z - g.z;
      const float n0 = k_div<LEAN>(d0 * d0, iv), n1 = k_div<LEAN>(d1 * d1, iv), n2 = k_div<LEAN>(d2 * d2, iv);
      if (a.d.nis_gate > 0.f && ((n0 + n1) + n2) > a.d.nis_gate) {
        // eval.py:87-92: gated OUTPUT takes the measurement coords, keeps KF sigma
        outv.x = z.x; outv.y = z.y; outv.z = z.z;
      }
      if (a.opt_nis && valid) {
        a.opt_nis[(off + p) * 3 + 0] = n0;
        a.opt_nis[(off + p) * 3 + 1] = n1;
        a.opt_nis[(off + p) * 3 + 2] = n2;
      }
    }
    if (a.opt_temp && valid) {
      f32x4 tv = {g.x, g.y, g.z, temp_unc};
      a.opt_temp[off + p] = tv;
    }
  }
  if (reset) {
    // eval.py:94-101: state := measurement, outputs := measurement
    nv = z;
    outv = z;
  }
  // ApplyTransform (util.py:12-40) + 1/sigma (eval.py:123)
  f32x4 r;
  if (a.d.has_transform) {
    const float* M = a.d.transform;
    r.x = ((M[0] * outv.x + M[1] * outv.y) + M[2] * outv.z) + M[3];
    r.y = ((M[4] * outv.x + M[5] * outv.y) + M[6] * outv.z) + M[7];
    r.z = ((M[8] * outv.x + M[9] * outv.y) + M[10] * outv.z) + M[11];
  } else {
    r.x = outv.x; r.y = outv.y; r.z = outv.z;
  }
  r.w = k_div<LEAN>(1.0f, outv.w);
  if (rec_out != nullptr) *rec_out = r;
  else if (valid) st_stream<NT>(a.rec + off + p, r);
  return nv;
}

// DBL: the state is double-buffered in LDS (2 x 76.8 KB at 60x80): frame t gathers from
// buffer t&1 and writes the fused state straight into the other one -- one barrier per
// frame and no per-thread copy of the new state.  Grids whose two copies exceed the
// 160 KB LDS use the single-buffer form (fuse into registers, barrier, write back, barrier).
// D = depth of the input ring (1 <= D <= PPT, PPT % D == 0): slot k % D holds the inputs of pixel slot k; the moment pixel k
// is fused its registers are re-loaded with pixel k + D of the same frame, or pixel k + D - PPT of the NEXT frame (inputs do
// not depend on the state, so the ring rolls straight across the frame barrier).  D = PPT is the full rolling prefetch (a
// whole frame of inputs in flight per thread); the single-buffer forms of the larger grids, which also hold PPT new states
// in registers across their mid-frame barrier, take a short ring.  NT: non-temporal input loads / record stores.
// PTR: how the streams are addressed.  true = per-pixel 64-bit pointers (clamped index, predicated stores): the compiler
// keeps them live across the frame loop (eight VGPRs per slot), which the double-buffered 60x80 form can afford; it was the
// fastest form there while the descriptors still sat in waterfall loops (0.662 / 0.672 of 8 TB/s at T = 64 / 256 against
// 0.633 / 0.645), and with scalar descriptors the two are level (S = 256, same box: 0.664 / 0.630 against 0.661 / 0.629-0.647),
// so the production 60x80 form stays as measured in the profiles.  false = buffer descriptors with scalar frame / slot
// offsets (below): no per-slot registers, which is what lets the single-buffer forms of the larger grids run without spills
// (68x120, S = 256: 0.720 against 0.43 for rounds 1-4's form).
template <int KT, int PPT, bool DBL, int D, bool NT, bool DBG, bool PTR = false, bool LEAN = (KFN_KALMAN_LEAN != 0)>
__global__ __launch_bounds__(KT) void kalman_scan_kernel(KalmanArgs a) {
  static_assert(D >= 1 && D <= PPT && PPT % D == 0, "ring slots are compile-time constants");
  extern __shared__ __attribute__((aligned(16))) float smem_k[];
  f32x4* st_base = reinterpret_cast<f32x4*>(smem_k);
  const int tid = threadIdx.x;
  const int s = blockIdx.x;
  const int H = a.d.H, W = a.d.W, HW = H * W, T = a.d.T;
  const float eps2 = a.d.min_uncertainty * a.d.min_uncertainty;
  const float xmax = (float)(W - 1), ymax = (float)(H - 1);
  const bool want_nis = (DBG && a.opt_nis != nullptr) || (a.d.nis_gate > 0.f);

  // state -> LDS
  for (int p = tid; p < HW; p += KT) st_base[p] = a.state[(size_t)s * HW + p];

  // ---- addressing: ONE buffer descriptor per stream, based at this sequence and exactly its T frames long.  Pixel slot k
  // of frame t sits at  voffset = tid * size  (per thread)  +  soffset = (t HW + k KT) * size  (wave-uniform: scalar
  // registers and scalar arithmetic).  The per-pixel 64-bit pointers of the first version cost eight loop-invariant VGPRs
  // per slot (the compiler hoists them out of the frame loop): 56-224 spilled registers in the single-buffer forms.  The
  // hardware range-checks voffset + soffset against num_records (tools/mb/srd_probe.hip): a thread whose slot lies past the grid
  // (tid + k KT >= HW) gets a voffset beyond every sequence -- its loads return zeros, its stores are dropped -- so every
  // load and store is unconditional, the number of vector-memory operations between a load and its use is a compile-time
  // constant and the s_waitcnt in front of the use leaves the younger loads in flight.  (The look-ahead's frame index is
  // clamped to T - 1: it re-reads the last frame instead of zeros.  T HW 16 < 2^31 -- checked by the launcher -- keeps the
  // poison 0x08000000 elements beyond num_records for all three element sizes.)
  const size_t seq_px = (size_t)s * T * HW;
  const unsigned aux = NT ? 2u : 0u;          // the non-temporal bit: streamed once, not re-read by this launch
  // (hipcc does not prove these descriptors wave-uniform -- the sequence offset reaches them through the divergent
  //  state-copy loop above -- and would wrap every buffer instruction in a waterfall loop: four v_readfirstlane, two
  //  compares, a branch.  The base pointers go through v_readfirstlane instead: 72 instead of 88 VGPRs, a quarter fewer
  //  instructions; same box, S = 256: 68x120 0.651 -> 0.720 of 8 TB/s, config 5's S = 4 x T = 64 scan 0.711 -> 0.601 ms,
  //  60x80 descriptor form 0.613 -> 0.661 (profiles/r05_kalman_srd_ab.log).  The first build of this returned WRONG records
  //  in the single-buffer forms: that was the 16-byte-store hazard of kfn_common.h buffer_store_b128 -- without the
  //  waterfall loop nothing separated the record store from the next write of its data registers -- and with the padded
  //  store every form is bit-identical to rounds 1-4's kernel again.  KFN_SCAN_UNIFORM_SRD=0 builds the waterfall form.)
#if KFN_SCAN_UNIFORM_SRD
  auto uniform = [](auto* q) {       // the pointer's two halves through v_readfirstlane: the descriptor is built in SGPRs
    const uintptr_t v = reinterpret_cast<uintptr_t>(q);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<decltype(q)>((uintptr_t)lo | ((uintptr_t)hi << 32));
  };
#else
  auto uniform = [](auto* q) { return q; };
#endif
  const __amdgpu_buffer_rsrc_t rsF = __builtin_amdgcn_make_buffer_rsrc(uniform(const_cast<f32x2*>(a.flow + seq_px)), 0, T * HW * 8, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc(uniform(const_cast<float*>(a.sigma_t + seq_px)), 0, T * HW * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsM = __builtin_amdgcn_make_buffer_rsrc(uniform(const_cast<f32x4*>(a.meas + seq_px)), 0, T * HW * 16, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(uniform(a.rec + seq_px), 0, T * HW * 16, 0x00020000);
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  constexpr unsigned POISON = 0x08000000u;    // element index: x 4 / 8 / 16 >= num_records of the three element sizes
  PixIn ring[D];
  // `tv` = this thread's index, opaque per frame (see below); slot k's element index for the range check
  auto slot_index = [&](int tv, int k) { return (tv + k * KT < HW) ? (unsigned)tv : POISON; };
  auto load_pixel = [&](int tv, int t, int k, PixIn& dst) __attribute__((always_inline)) {
    const int tc = t < T - 1 ? t : T - 1;
    if constexpr (PTR) {
      // threads past the grid re-read the last pixel (and store nothing): every load unconditional here too
      const size_t q = seq_px + (size_t)tc * HW + min(tid + k * KT, HW - 1);
      dst.flow = ld_stream<NT>(a.flow + q);
      dst.st = ld_stream<NT>(a.sigma_t + q);
      dst.z = ld_stream<NT>(a.meas + q);
      return;
    }
    const unsigned e = (unsigned)(tc * HW + k * KT);             // uniform element offset of the slot
    const unsigned vi = slot_index(tv, k);
    dst.flow = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsF, vi * 8u, e * 8u, aux));
    dst.st = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsS, vi * 4u, e * 4u, aux));
    dst.z = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsM, vi * 16u, e * 16u, aux));
  };
  // grid position of slot k: (y0, x0) of pixel `tid` plus the wave-uniform (k KT) / W, (k KT) % W with one carry
  const int y0 = tid / W, x0 = tid - y0 * W;
#pragma unroll
  for (int k = 0; k < D; ++k) load_pixel(tid, 0, k, ring[k]);
  __syncthreads();

  for (int t = 0; t < T; ++t) {
    const int gi = a.d.t0 + t;
    const bool reset = a.d.reset_period > 0 && (gi % a.d.reset_period) == 0;
    const size_t off = seq_px + (size_t)t * HW;
    const f32x4* st = DBL ? st_base + (t & 1) * HW : st_base;
    f32x4* st_new = DBL ? st_base + ((t + 1) & 1) * HW : st_base;
    f32x4 newst[DBL ? 1 : PPT];
    // (opaque per frame: keeps the per-slot positions / offsets / predicates from being hoisted out of the frame loop
    //  into several live registers per slot)
    int x0f = x0, y0f = y0, tv = tid;
    asm volatile("" : "+v"(x0f), "+v"(y0f), "+v"(tv));
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      const int p = tv + k * KT;
      const bool valid = p < HW;
      const int qk = (k * KT) / W, rk = (k * KT) - qk * W;     // uniform
      int xk = x0f + rk, yk = y0f + qk;
      if (xk >= W) { xk -= W; yk += 1; }
      f32x4 rec, nv;
      if constexpr (PTR) {
        const int pc = min(tid + k * KT, HW - 1);
        const int yc = pc / W, xc = pc - yc * W;
        nv = fuse_pixel<NT, DBG, LEAN>(a, st, ring[k % D], pc, xc, yc, off, reset, W, xmax, ymax, eps2, want_nis, valid);
      } else {
        nv = fuse_pixel<NT, DBG, LEAN>(a, st, ring[k % D], p, xk, yk, off, reset, W, xmax, ymax, eps2, want_nis, valid, &rec);
        kfn::buffer_store_b128<NT ? 2 : 0>(rec, rsR, slot_index(tv, k) * 16u,
                                           (unsigned)(t * HW + k * KT) * 16u);               // (threads past the grid: dropped)
      }
      if (DBL) { if (valid) st_new[p] = nv; } else newst[DBL ? 0 : k] = nv;
      // this slot's inputs are consumed: fetch the pixel that uses the slot next
      if (k + D < PPT) load_pixel(tv, t, k + D, ring[k % D]);
      else load_pixel(tv, t + 1, k + D - PPT, ring[k % D]);
      // keep the unrolled pixels sequential: interleaving them only multiplies live temporaries (measured with pairs and
      // with all five interleaved: -0.3 ... -1 %)
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!DBL) {
      __syncthreads();  // every gather of frame t done
#pragma unroll
      for (int k = 0; k < PPT; ++k) {
        int p = tid + k * KT;
        if (p < HW) st_base[p] = newst[DBL ? 0 : k];
      }
    }
    __syncthreads();  // new state visible, old buffer free
  }
  const f32x4* st_fin = DBL ? st_base + (T & 1) * HW : st_base;
  for (int p = tid; p < HW; p += KT) a.state[(size_t)s * HW + p] = st_fin[p];
}

// Grids whose state does not fit the LDS (more than 10 240 pixels): one launch per frame over
// all sequences, the state ping-pongs between two global buffers (`prev` is only read, `next`
// only written, so the 4-tap gather needs no synchronisation inside a launch).
__global__ __launch_bounds__(256) void kalman_step_kernel(KalmanArgs a, const f32x4* __restrict__ prev,
                                                          f32x4* __restrict__ next, int t) {
  const int H = a.d.H, W = a.d.W, HW = H * W;
  const int p = blockIdx.x * 256 + threadIdx.x;
  const int s = blockIdx.y;
  if (p >= HW) return;
  const size_t off = ((size_t)s * a.d.T + t) * HW;
  PixIn in;
  in.flow = a.flow[off + p];
  in.st = a.sigma_t[off + p];
  in.z = a.meas[off + p];
  const int gi = a.d.t0 + t;
  const bool reset = a.d.reset_period > 0 && (gi % a.d.reset_period) == 0;
  const float eps2 = a.d.min_uncertainty * a.d.min_uncertainty;
  const bool want_nis = (a.opt_nis != nullptr) || (a.d.nis_gate > 0.f);
  const int py = p / W, px = p - py * W;
  next[(size_t)s * HW + p] = fuse_pixel(a, prev + (size_t)s * HW, in, p, px, py, off, reset, W, (float)(W - 1),
                                        (float)(H - 1), eps2, want_nis);
}

// KFNet.BuildKFCoord alone (KFNet/KFNet.py:148-162), optional GetNIS (:164-184):
// 32 B read + 16 B written per pixel, pure HBM streaming.
__device__ __forceinline__ f32x4 kf_update(const f32x4 l, const f32x4 z) {
  const float lv = l.w * l.w;
  const float mv = z.w * z.w;
  const float K = lv / (lv + mv);
  const float om = fmaxf(1.0f - K, 0.0f);
  f32x4 kf;
  kf.x = om * l.x + K * z.x;
  kf.y = om * l.y + K * z.y;
  kf.z = om * l.z + K * z.z;
  kf.w = sqrtf(om * lv);
  return kf;
}
__device__ __forceinline__ void kf_nis(const f32x4 l, const f32x4 z, float* nis) {
  const float iu = sqrtf(l.w * l.w + z.w * z.w);
  const float iv = iu * iu;
  const float d0 = z.x - l.x, d1 = z.y - l.y, d2 = z.z - l.z;
  nis[0] = (d0 * d0) / iv;
  nis[1] = (d1 * d1) / iv;
  nis[2] = (d2 * d2) / iv;
}

// BLOCK threads, U pixels per thread and trip (pixel p + j * BLOCK: every load / store of a wave is one contiguous 1 KiB
// run); all 2 U loads of a trip are issued before the first result is needed, so a thread keeps 32 U bytes in flight
// (rounds 1-4: U = 1, one float4 pair per trip, 0.55 of 8 TB/s).  NT: the operands are read once and the result is
// not re-read by this kernel -- non-temporal loads and stores.
template <int BLOCK, int U, bool NT>
__global__ __launch_bounds__(BLOCK) void kalman_fuse_kernel(const f32x4* __restrict__ pred,
                                                            const f32x4* __restrict__ meas,
                                                            f32x4* __restrict__ out,
                                                            float* __restrict__ nis, long P) {
  const long stride = (long)gridDim.x * (BLOCK * U);
  for (long base = (long)blockIdx.x * (BLOCK * U) + threadIdx.x; base < P; base += stride) {
    f32x4 l[U], z[U];
    if (base + (long)(U - 1) * BLOCK < P) {            // a whole trip (every lane of it in range): no per-load predicate
#pragma unroll
      for (int j = 0; j < U; ++j) l[j] = ld_stream<NT>(pred + base + (long)j * BLOCK);
#pragma unroll
      for (int j = 0; j < U; ++j) z[j] = ld_stream<NT>(meas + base + (long)j * BLOCK);
#pragma unroll
      for (int j = 0; j < U; ++j) {
        st_stream<NT>(out + base + (long)j * BLOCK, kf_update(l[j], z[j]));
        if (nis) kf_nis(l[j], z[j], nis + (base + (long)j * BLOCK) * 3);
      }
    } else {
      for (int j = 0; j < U; ++j) {
        const long p = base + (long)j * BLOCK;
        if (p >= P) break;
        const f32x4 lj = pred[p], zj = meas[p];
        out[p] = kf_update(lj, zj);
        if (nis) kf_nis(lj, zj, nis + p * 3);
      }
    }
  }
}

// KFNet.GetKFCoord2 (KFNet/KFNet.py:487-502): the same gain, but the posterior variance in the symmetric
// ("Joseph") form  (1-K)^2 P^- + K^2 R  instead of BuildKFCoord's  max(1-K,0) P^-.  Not on eval.py's path (the
// recursive graph uses BuildKFCoord); kept for the reference's API.  Operation order as in the reference.
__global__ __launch_bounds__(256) void kalman_fuse2_kernel(const f32x4* __restrict__ pred,
                                                           const f32x4* __restrict__ meas,
                                                           f32x4* __restrict__ out, long P) {
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (long)gridDim.x * blockDim.x) {
    const f32x4 l = pred[p], z = meas[p];
    const float mv = z.w * z.w;              // measure_variance2
    const float tv = l.w * l.w;              // variance_12 = temp_variance2
    const float K = tv / (tv + mv);
    const float om = fmaxf(1.0f - K, 0.0f);
    const float omr = 1.0f - K;
    f32x4 kf;
    kf.x = om * l.x + K * z.x;
    kf.y = om * l.y + K * z.y;
    kf.z = om * l.z + K * z.z;
    kf.w = sqrtf((omr * omr) * tv + (K * K) * mv);
    out[p] = kf;
  }
}

// The form launched for grids whose two state copies fit the LDS (up to 5 120 pixels: 60x80 and below).  Build-time
// switches of the A/B builds (tools/mb/kalman_mb.hip instantiates every combination in one binary).
#ifndef KFN_SCAN_KT
#define KFN_SCAN_KT 1024
#define KFN_SCAN_PPT 5
#endif
#ifndef KFN_SCAN_DEPTH
#define KFN_SCAN_DEPTH KFN_SCAN_PPT
#endif
#ifndef KFN_SCAN_BIG_DEPTH
#define KFN_SCAN_BIG_DEPTH 2
#endif
#ifndef KFN_SCAN_NT
#define KFN_SCAN_NT 1
#endif
// fuse kernel: 256 threads x 4 pixels, non-temporal, ONE trip per thread (measured, tools/mb/kalman_mb, P = 78.6 M / 314.6 M
// pixels: rounds 1-4's 256 x 1 grid-stride form 0.623 / 0.622 of 8 TB/s; 256 x 4 nt capped at 4096 blocks 0.773 / 0.703;
// 512 x 4 nt 0.738 / 0.680; 256 x 4 nt one trip 0.786 / 0.750)
#ifndef KFN_FUSE_BLOCK
#define KFN_FUSE_BLOCK 256
#define KFN_FUSE_U 4
#define KFN_FUSE_NT 1
#endif

template <int KT, int PPT, bool DBL, int D, bool NT, bool DBG, bool PTR = false, bool LEAN = (KFN_KALMAN_LEAN != 0)>
int launch_scan_dbg(const KalmanArgs& a, hipStream_t stream) {
  const size_t smem = (size_t)a.d.H * a.d.W * sizeof(f32x4) * (DBL ? 2 : 1);
  auto kern = kalman_scan_kernel<KT, PPT, DBL, D, NT, DBG, PTR, LEAN>;
  static std::atomic<uint64_t> attr_done{0};   // per instantiation: bit per device
  {
    int rc = kfn::set_max_dynamic_lds(reinterpret_cast<const void*>(kern), 160 * 1024, attr_done);
    if (rc != KFN_OK) return rc;
  }
  hipLaunchKernelGGL(kern, dim3(a.d.S), dim3(KT), smem, stream, a);
  KFN_LAUNCH_CHECK("kalman_scan_kernel");
  return KFN_OK;
}

// the production instantiation has no optional outputs compiled in; a call that asks for temp / NIS / raw-KF maps
// (eval metrics, debug) gets the instantiation that has them -- DD = its ring depth (the three optional stores per pixel
// cost registers: a full ring would spill)
template <int KT, int PPT, bool DBL, int D, bool NT, int KT_DBG = KT, int PPT_DBG = PPT, int DD = 1>
int launch_scan(const KalmanArgs& a, hipStream_t stream) {
  if (a.opt_temp || a.opt_nis || a.opt_kf) return launch_scan_dbg<KT_DBG, PPT_DBG, DBL, DD, NT, true, false>(a, stream);
  return launch_scan_dbg<KT, PPT, DBL, D, NT, false, /*PTR=*/DBL>(a, stream);
}

// self-check of the scan's sqrt / quotient (kfn_kalman_arith_probe): out[4 i .. 4 i + 3] = sqrt_rn_normal(a), sqrtf(a),
// div_rn_normal(a, b), a / b
__global__ __launch_bounds__(256) void kalman_arith_probe_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                 f32x4* __restrict__ out, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float x = a[i], y = b[i];
  f32x4 r;
  r.x = sqrt_rn_normal(x);
  r.y = sqrtf(x);
  r.z = div_rn_normal(x, y);
  r.w = x / y;
  out[i] = r;
}

}  // namespace

extern "C" int kfn_kalman_scan_scratch_bytes(const kfn_kalman_desc* d, size_t* bytes) {
  KFN_REQUIRE(d && bytes, "kfn_kalman_scan_scratch_bytes: null argument");
  KFN_REQUIRE(d->S > 0 && d->H > 1 && d->W > 1, "kfn_kalman_scan_scratch_bytes: bad shape S=%d H=%d W=%d", d->S, d->H, d->W);
  const size_t hw = (size_t)d->H * d->W;
  *bytes = hw * 16 > 160 * 1024 ? (size_t)d->S * hw * 16 : 0;   // a second copy of the state, only when it cannot live in LDS
  return KFN_OK;
}

extern "C" int kfn_kalman_scan_ex(const kfn_kalman_desc* d, const float* flow_xy,
                                  const float* sigma_trans, const float* meas, float* state,
                                  float* records, float* opt_temp, float* opt_nis, float* opt_kf,
                                  int raw_on_reset, void* scratch_buf, void* stream) {
  KFN_REQUIRE(d && flow_xy && sigma_trans && meas && state && records, "kfn_kalman_scan: null argument");
  KFN_REQUIRE(d->S > 0 && d->T > 0 && d->H > 1 && d->W > 1, "kfn_kalman_scan: bad shape S=%d T=%d H=%d W=%d",
              d->S, d->T, d->H, d->W);
  const int HW = d->H * d->W;
  KFN_REQUIRE(((reinterpret_cast<uintptr_t>(flow_xy) & 7) | (reinterpret_cast<uintptr_t>(meas) & 15) |
               (reinterpret_cast<uintptr_t>(state) & 15) | (reinterpret_cast<uintptr_t>(records) & 15) |
               (reinterpret_cast<uintptr_t>(opt_temp) & 15) | (reinterpret_cast<uintptr_t>(opt_kf) & 15)) == 0,
              "kfn_kalman_scan: misaligned buffer");
  KalmanArgs a;
  a.flow = reinterpret_cast<const f32x2*>(flow_xy);
  a.sigma_t = sigma_trans;
  a.meas = reinterpret_cast<const f32x4*>(meas);
  a.state = reinterpret_cast<f32x4*>(state);
  a.rec = reinterpret_cast<f32x4*>(records);
  a.opt_temp = reinterpret_cast<f32x4*>(opt_temp);
  a.opt_nis = opt_nis;
  a.opt_kf = reinterpret_cast<f32x4*>(opt_kf);
  a.raw_on_reset = raw_on_reset;
  a.d = *d;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if ((size_t)HW * 16 > 160 * 1024) {
    // state larger than the LDS: per-frame launches, state ping-pong between `state` and the CALLER's scratch copy
    // (kfn_kalman_scan_scratch_bytes): nothing is allocated here, the T launches are stream-ordered and capturable.
    // (A single cooperative launch with a grid barrier per frame was the alternative: ~4 us per barrier against ~1.5 us
    //  per dependent launch boundary on this part -- MI355X_MICROARCH.md's price list -- so the launches stay.)
    const size_t bytes = (size_t)d->S * HW * sizeof(f32x4);
    KFN_REQUIRE(scratch_buf != nullptr && (reinterpret_cast<uintptr_t>(scratch_buf) & 15) == 0,
                "kfn_kalman_scan: a %dx%d grid does not fit the LDS: pass a 16-byte aligned scratch buffer of "
                "kfn_kalman_scan_scratch_bytes() = %zu bytes", d->H, d->W, bytes);
    f32x4* scratch = reinterpret_cast<f32x4*>(scratch_buf);
    f32x4* buf[2] = {a.state, scratch};
    const dim3 grid((unsigned)((HW + 255) / 256), (unsigned)d->S);
    int rc = KFN_OK;
    for (int t = 0; t < d->T && rc == KFN_OK; ++t) {
      hipLaunchKernelGGL(kalman_step_kernel, grid, dim3(256), 0, s, a, buf[t & 1], buf[(t + 1) & 1], t);
      rc = kfn::check_hip(hipGetLastError(), "kalman_step_kernel");
    }
    if (rc == KFN_OK && (d->T & 1))
      rc = kfn::check_hip(hipMemcpyAsync(a.state, scratch, bytes, hipMemcpyDeviceToDevice, s), "state copy-back");
    return rc;
  }
  KFN_REQUIRE((long)d->T * HW * 16L < (1L << 31),
              "kfn_kalman_scan: T * H * W = %ld pixel-frames per sequence exceed the 32-bit offsets of one launch (at most %ld frames "
              "of this grid): scan the sequence in chunks", (long)d->T * HW, ((1L << 31) - 1) / (16L * HW));
  // 768 threads (12 wavefronts, 170-VGPR budget) x 7 pixels cover the 60x80 grid without
  // register spills; larger grids fall back to 1024 threads and the single-buffer form.
  const bool dbl = (size_t)HW * 32 <= 160 * 1024;  // two LDS copies of the state fit
  constexpr bool NT = KFN_SCAN_NT != 0;
  if (dbl && HW <= KFN_SCAN_KT * KFN_SCAN_PPT)
    return launch_scan<KFN_SCAN_KT, KFN_SCAN_PPT, true, KFN_SCAN_DEPTH, NT, 768, 7, 1>(a, s);
  // single LDS copy (config 5's 68x120 = 8160-pixel grid takes the second line): the new states of a frame wait in
  // registers for the mid-frame barrier -- 1024 threads keep that to PPT <= 10 float4 per thread beside a two-deep input
  // ring.  Measured (tools/mb/kalman_mb, 68x120, S = 256 x T = 32 / S = 4 x T = 64): rounds 1-4's 512 x 16 with the loads at
  // the frame start 0.867 / 1.011 ms; 1024 x 8 with D = 2 0.793 / 0.914; D = 4 0.796 / 0.968; 768 x 11 D = 1 0.787 / 1.064.
  if (HW <= 1024 * 6) return launch_scan<1024, 6, false, 2, NT, 512, 12, 1>(a, s);
  if (HW <= 1024 * 8) return launch_scan<1024, 8, false, KFN_SCAN_BIG_DEPTH, NT, 512, 16, 1>(a, s);
  return launch_scan<1024, 10, false, 1, NT, 512, 20, 1>(a, s);
}

extern "C" int kfn_kalman_scan(const kfn_kalman_desc* d, const float* flow_xy,
                               const float* sigma_trans, const float* meas, float* state,
                               float* records, float* opt_temp, float* opt_nis, void* scratch, void* stream) {
  return kfn_kalman_scan_ex(d, flow_xy, sigma_trans, meas, state, records, opt_temp, opt_nis, nullptr, 0, scratch, stream);
}

extern "C" int kfn_kalman_fuse(const float* pred, const float* meas, float* out, float* opt_nis,
                               long P, void* stream) {
  KFN_REQUIRE(pred && meas && out && P > 0, "kfn_kalman_fuse: bad argument");
  KFN_REQUIRE(((reinterpret_cast<uintptr_t>(pred) | reinterpret_cast<uintptr_t>(meas) |
                reinterpret_cast<uintptr_t>(out)) & 15) == 0, "kfn_kalman_fuse: misaligned buffer");
  constexpr int PER = KFN_FUSE_BLOCK * KFN_FUSE_U;
  const long blocks = (P + PER - 1) / PER;      // one trip per thread (the kernel's loop runs once)
  KFN_REQUIRE(blocks < (1L << 31), "kfn_kalman_fuse: P=%ld too large for one launch", P);
  hipLaunchKernelGGL((kalman_fuse_kernel<KFN_FUSE_BLOCK, KFN_FUSE_U, KFN_FUSE_NT != 0>), dim3((unsigned)blocks), dim3(KFN_FUSE_BLOCK), 0,
                     (hipStream_t)stream, reinterpret_cast<const f32x4*>(pred), reinterpret_cast<const f32x4*>(meas),
                     reinterpret_cast<f32x4*>(out), opt_nis, P);
  KFN_LAUNCH_CHECK("kalman_fuse_kernel");
  return KFN_OK;
}

extern "C" int kfn_kalman_fuse2(const float* pred, const float* meas, float* out, long P, void* stream) {
  KFN_REQUIRE(pred && meas && out && P > 0, "kfn_kalman_fuse2: bad argument");
  KFN_REQUIRE(((reinterpret_cast<uintptr_t>(pred) | reinterpret_cast<uintptr_t>(meas) |
                reinterpret_cast<uintptr_t>(out)) & 15) == 0, "kfn_kalman_fuse2: misaligned buffer");
  long blocks = (P + 255) / 256;
  if (blocks > 256L * 16) blocks = 256L * 16;
  hipLaunchKernelGGL(kalman_fuse2_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const f32x4*>(pred), reinterpret_cast<const f32x4*>(meas),
                     reinterpret_cast<f32x4*>(out), P);
  KFN_LAUNCH_CHECK("kalman_fuse2_kernel");
  return KFN_OK;
}

extern "C" int kfn_kalman_arith_probe(const float* a, const float* b, float* out, long n, void* stream) {
  KFN_REQUIRE(a && b && out && n > 0 && (n + 255) / 256 < (1L << 31), "kfn_kalman_arith_probe: bad argument");
  KFN_REQUIRE((reinterpret_cast<uintptr_t>(out) & 15) == 0, "kfn_kalman_arith_probe: out must be 16-byte aligned");
  hipLaunchKernelGGL(kalman_arith_probe_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, b,
                     reinterpret_cast<f32x4*>(out), n);
  KFN_LAUNCH_CHECK("kalman_arith_probe_kernel");
  return KFN_OK;
}
