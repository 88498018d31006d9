// kfn_conv.hip -- fp32 implicit-GEMM convolution on v_mfma_f32_32x32x2_f32 (gfx950).
//
// Replaces tf.layers.conv2d / conv2d_transpose behind Network.conv / Network.deconv
// (cnn_wrapper/network.py:116-135, 418-437).  M = N*Ho*Wo output pixels, N = Cout,
// K = taps*Cin.  Both operands are staged through LDS "K-contiguous":
//   A tile [BM][BK]  = im2col rows gathered on the fly from the NHWC activations
//   B tile [BN][BK]  = rows of the pre-packed weight matrix w_packed[Cout][K]
// with a 4-float row pad so that the ds_read_b128 fragment reads (lane (i,h) reads
// row i, floats [8c+4h, 8c+4h+4)) are bank-conflict free (row stride 36 or 20 dwords:
// 9 resp. 5 are odd, so 16 rows that are distinct mod 16 cover all 64 banks).
// Each 8-wide k-chunk feeds four 32x32x2 MFMAs per (mi,ni) tile: MFMA step t takes
// k = 8c + 4h + t from lane half h -- A and B use the same assignment, so the k
// permutation is harmless.  Global loads for stage s+1 are issued before the MFMAs of
// stage s (register-staged double buffer, one barrier per stage).
#include "kfn_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvArgs {
  const float* x;
  const float* w;
  const float* bias;
  float* y;
  int N, H, W, Cin, ldx;
  int Ho, Wo, Cout, cout_pad, ldy;
  int kh, kw, stride, pad_t, pad_l;
  int transposed, relu, epilogue;
  int M, Ktot;
  int tiles_m, tiles_n;
};

__device__ __forceinline__ int xcd_remap(int b, int nwg) {
  // Blocks are dispatched round-robin over the 8 XCDs (b % 8).  Give every XCD a
  // contiguous run of logical tiles so that tiles sharing A rows / B columns share an L2.
  int xcd = b & 7;
  int q = nwg >> 3, r = nwg & 7;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (b >> 3);
}

template <int TM, int TN, int WM, int WN, int BK>
__global__ __launch_bounds__(64 * WM * WN) void conv_mfma_kernel(ConvArgs p) {
  constexpr int BM = 32 * TM * WM;
  constexpr int BN = 32 * TN * WN;
  constexpr int NT = 64 * WM * WN;
  constexpr int LDK = BK + 4;
  constexpr int QPR = BK / 4;       // float4 quads per tile row
  constexpr int RPP = NT / QPR;     // tile rows covered per pass of the whole block
  constexpr int AP = (BM + RPP - 1) / RPP;
  constexpr int BP = (BN + RPP - 1) / RPP;
  constexpr int A_ELEMS = BM * LDK;
  constexpr int B_ELEMS = BN * LDK;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                 // [2][BM][LDK]
  float* Bs = smem + 2 * A_ELEMS;   // [2][BN][LDK]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN;
  const int wn = wave % WN;

  const int nwg = p.tiles_m * p.tiles_n;
  const int tile = xcd_remap(blockIdx.x, nwg);
  const int tn = tile % p.tiles_n;
  const int tm = tile / p.tiles_n;
  const int m0 = tm * BM;
  const int n0 = tn * BN;

  const int q = tid % QPR;
  const int r0 = tid / QPR;

  // ---- per-thread im2col row state -------------------------------------------------
  int a_pix0[AP];  // n_img*H*W
  int a_iy0[AP];
  int a_ix0[AP];
  const int HoWo = p.Ho * p.Wo;
#pragma unroll
  for (int i = 0; i < AP; ++i) {
    int r = r0 + i * RPP;
    int m = m0 + r;
    if (r < BM && m < p.M) {
      int n_img = m / HoWo;
      int rem = m - n_img * HoWo;
      int oy = rem / p.Wo;
      int ox = rem - oy * p.Wo;
      a_pix0[i] = n_img * p.H * p.W;
      if (p.transposed) {
        a_iy0[i] = oy + p.pad_t;
        a_ix0[i] = ox + p.pad_l;
      } else {
        a_iy0[i] = oy * p.stride - p.pad_t;
        a_ix0[i] = ox * p.stride - p.pad_l;
      }
    } else {
      a_pix0[i] = 0;
      a_iy0[i] = -(1 << 20);
      a_ix0[i] = -(1 << 20);
    }
  }

  auto tap_pixel = [&](int i, int ky, int kx, int& pix) -> bool {
    if (p.transposed) {
      int ty = a_iy0[i] - ky, tx = a_ix0[i] - kx;
      bool ok = (ty >= 0) && (tx >= 0) && (((ty | tx) & 1) == 0) && ((ty >> 1) < p.H) &&
                ((tx >> 1) < p.W);
      pix = a_pix0[i] + (ty >> 1) * p.W + (tx >> 1);
      return ok;
    } else {
      int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
      bool ok = ((unsigned)iy < (unsigned)p.H) && ((unsigned)ix < (unsigned)p.W);
      pix = a_pix0[i] + iy * p.W + ix;
      return ok;
    }
  };

  // ---- which taps touch at least one in-range input pixel of this tile? -----------
  // (zero-padding taps of whole tiles are skipped: exact, they only add +0.)
  const int ntaps = p.kh * p.kw;
  unsigned tapmask = 0;
  {
    unsigned mine = 0;
    for (int t = 0; t < ntaps; ++t) {
      int ky = t / p.kw, kx = t - ky * p.kw;
      bool any = false;
#pragma unroll
      for (int i = 0; i < AP; ++i) {
        int pix;
        any |= tap_pixel(i, ky, kx, pix);
      }
      if (any) mine |= (1u << t);
    }
    // block-wide OR through LDS (smem is free before the main loop)
    unsigned* red = reinterpret_cast<unsigned*>(smem);
    if (tid == 0) red[0] = 0;
    __syncthreads();
    if (mine) atomicOr(red, mine);
    __syncthreads();
    tapmask = red[0];
    __syncthreads();
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int mi = 0; mi < TM; ++mi)
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  f32x4 ga[AP], gb[BP];

  auto load_stage = [&](int tap, int c0) {
    int ky = tap / p.kw, kx = tap - ky * p.kw;
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      int pix;
      bool ok = tap_pixel(i, ky, kx, pix);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ok) v = *reinterpret_cast<const f32x4*>(p.x + (size_t)pix * p.ldx + c0 + q * 4);
      ga[i] = v;
    }
    const int kbase = tap * p.Cin + c0 + q * 4;
#pragma unroll
    for (int i = 0; i < BP; ++i) {
      int r = r0 + i * RPP;
      int n = n0 + r;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (r < BN && n < p.cout_pad)
        v = *reinterpret_cast<const f32x4*>(p.w + (size_t)n * p.Ktot + kbase);
      gb[i] = v;
    }
  };

  auto store_stage = [&](int buf) {
    float* a = As + buf * A_ELEMS;
    float* b = Bs + buf * B_ELEMS;
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      int r = r0 + i * RPP;
      if (AP * RPP == BM || r < BM) *reinterpret_cast<f32x4*>(a + r * LDK + q * 4) = ga[i];
    }
#pragma unroll
    for (int i = 0; i < BP; ++i) {
      int r = r0 + i * RPP;
      if (BP * RPP == BN || r < BN) *reinterpret_cast<f32x4*>(b + r * LDK + q * 4) = gb[i];
    }
  };

  auto compute_stage = [&](int buf) {
    const float* a = As + buf * A_ELEMS + (wm * TM * 32 + (lane & 31)) * LDK + (lane >> 5) * 4;
    const float* b = Bs + buf * B_ELEMS + (wn * TN * 32 + (lane & 31)) * LDK + (lane >> 5) * 4;
#pragma unroll
    for (int c = 0; c < BK / 8; ++c) {
      f32x4 af[TM], bf[TN];
#pragma unroll
      for (int mi = 0; mi < TM; ++mi)
        af[mi] = *reinterpret_cast<const f32x4*>(a + mi * 32 * LDK + c * 8);
#pragma unroll
      for (int ni = 0; ni < TN; ++ni)
        bf[ni] = *reinterpret_cast<const f32x4*>(b + ni * 32 * LDK + c * 8);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
          for (int ni = 0; ni < TN; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi][t], bf[ni][t], acc[mi][ni], 0, 0, 0);
    }
  };

  auto next_tap = [&](int t) {
    ++t;
    while (t < ntaps && !((tapmask >> t) & 1u)) ++t;
    return t;
  };

  int tap = next_tap(-1);
  if (tap < ntaps) {
    int c0 = 0;
    load_stage(tap, c0);
    store_stage(0);
    __syncthreads();
    int buf = 0;
    while (true) {
      int c1 = c0 + BK, tap1 = tap;
      if (c1 >= p.Cin) {
        c1 = 0;
        tap1 = next_tap(tap);
      }
      const bool has_next = tap1 < ntaps;
      if (has_next) load_stage(tap1, c1);
      compute_stage(buf);
      if (!has_next) break;
      store_stage(buf ^ 1);
      __syncthreads();
      buf ^= 1;
      tap = tap1;
      c0 = c1;
    }
  }

  // ---- epilogue: bias, ReLU, fused head ops, store ------------------------------------
  // C/D layout of the 32x32 MFMA: col = lane & 31, row = (e & 3) + 8*(e >> 2) + 4*(lane >> 5).
  const int col_l = lane & 31;
  const int rowh = 4 * (lane >> 5);
#pragma unroll
  for (int ni = 0; ni < TN; ++ni) {
    const int n = n0 + (wn * TN + ni) * 32 + col_l;
    const bool n_ok = n < p.Cout;
    const float bv = (p.bias != nullptr && n_ok) ? p.bias[n] : 0.f;
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + (wm * TM + mi) * 32 + (e & 3) + 8 * (e >> 2) + rowh;
        float v = acc[mi][ni][e] + bv;
        if (p.relu) v = fmaxf(v, 0.f);
        if (p.epilogue == KFN_EPI_L2NORM) {
          float ss = n_ok ? v * v : 0.f;
          ss += __shfl_xor(ss, 16);
          ss += __shfl_xor(ss, 8);
          ss += __shfl_xor(ss, 4);
          ss += __shfl_xor(ss, 2);
          ss += __shfl_xor(ss, 1);
          v = v / sqrtf(fmaxf(ss, 1e-12f));
        } else if (p.epilogue == KFN_EPI_EXP_CH3) {
          if (n == 3) v = expf(v);
        } else if (p.epilogue == KFN_EPI_EXP_1E2) {
          v = expf(v) * 1e-2f;
        }
        if (n_ok && m < p.M) p.y[(size_t)m * p.ldy + n] = v;
      }
    }
  }
}

template <int TM, int TN, int WM, int WN, int BK>
int launch_cfg(const ConvArgs& a0, hipStream_t stream) {
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, NT = 64 * WM * WN;
  constexpr int LDK = BK + 4;
  constexpr size_t smem = (size_t)2 * (BM + BN) * LDK * sizeof(float);
  ConvArgs a = a0;
  a.tiles_m = kfn::ceil_div(a.M, BM);
  a.tiles_n = kfn::ceil_div(a.Cout, BN);
  auto kern = conv_mfma_kernel<TM, TN, WM, WN, BK>;
  static bool attr_done = false;  // benign race: idempotent attribute
  if (!attr_done) {
    KFN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    attr_done = true;
  }
  dim3 grid(a.tiles_m * a.tiles_n), block(NT);
  hipLaunchKernelGGL(kern, grid, block, smem, stream, a);
  KFN_LAUNCH_CHECK("conv_mfma_kernel");
  return KFN_OK;
}

template <int BK>
int dispatch_cfg(int cfg, const ConvArgs& a, hipStream_t s) {
  switch (cfg) {
    case KFN_CFG_160x128: return launch_cfg<5, 1, 1, 4, BK>(a, s);
    case KFN_CFG_128x128: return launch_cfg<2, 2, 2, 2, BK>(a, s);
    case KFN_CFG_128x64: return launch_cfg<2, 1, 2, 2, BK>(a, s);
    case KFN_CFG_128x32: return launch_cfg<1, 1, 4, 1, BK>(a, s);
    case KFN_CFG_64x64: return launch_cfg<1, 1, 2, 2, BK>(a, s);
    default: return kfn::fail(KFN_ERR_ARG, "kfn_conv2d_nhwc: unknown config %d", cfg);
  }
}

// Tile choice: maximise (useful MFMA work) / (CU-rounds * tile work) for 256 CUs.
int auto_config(const ConvArgs& a, int num_cu) {
  struct Cand { int cfg, bm, bn; };
  const Cand cands[] = {{KFN_CFG_160x128, 160, 128}, {KFN_CFG_128x128, 128, 128},
                        {KFN_CFG_128x64, 128, 64},   {KFN_CFG_128x32, 128, 32},
                        {KFN_CFG_64x64, 64, 64}};
  double best = -1.0;
  int best_cfg = KFN_CFG_128x32;
  for (const Cand& c : cands) {
    long tiles = (long)kfn::ceil_div(a.M, c.bm) * kfn::ceil_div(a.Cout, c.bn);
    long rounds = (tiles + num_cu - 1) / num_cu;
    double eff = ((double)a.M * a.Cout) / ((double)rounds * num_cu * c.bm * c.bn);
    // larger tiles amortise LDS traffic / barriers better: small bonus
    eff *= (c.bm * c.bn >= 160 * 128) ? 1.00 : (c.bm * c.bn >= 128 * 128) ? 0.97
           : (c.bm * c.bn >= 128 * 64) ? 0.92 : 0.85;
    if (eff > best) {
      best = eff;
      best_cfg = c.cfg;
    }
  }
  return best_cfg;
}

int g_num_cu = 0;

}  // namespace

extern "C" int kfn_conv2d_out_shape(const kfn_conv_desc* d, int* Ho, int* Wo) {
  KFN_REQUIRE(d && Ho && Wo, "kfn_conv2d_out_shape: null argument");
  if (d->transposed) {
    *Ho = d->H * d->stride;
    *Wo = d->W * d->stride;
  } else {
    int pb;
    kfn::same_pad(d->H, d->kh, d->stride, Ho, &pb);
    kfn::same_pad(d->W, d->kw, d->stride, Wo, &pb);
  }
  return KFN_OK;
}

extern "C" int kfn_conv2d_nhwc(const kfn_conv_desc* d, const float* x, const float* w_packed,
                               const float* bias, float* y, void* stream) {
  KFN_REQUIRE(d && x && w_packed && y, "kfn_conv2d_nhwc: null argument");
  KFN_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0, "kfn_conv2d_nhwc: bad shape %dx%dx%d", d->N, d->H, d->W);
  KFN_REQUIRE(d->Cin > 0 && d->Cin % 16 == 0, "kfn_conv2d_nhwc: Cin=%d must be a multiple of 16", d->Cin);
  KFN_REQUIRE(d->ldx >= d->Cin && d->ldx % 4 == 0, "kfn_conv2d_nhwc: bad ldx=%d", d->ldx);
  KFN_REQUIRE(d->Cout > 0 && d->ldy >= d->Cout, "kfn_conv2d_nhwc: bad Cout=%d ldy=%d", d->Cout, d->ldy);
  KFN_REQUIRE(d->cout_pad >= d->Cout && d->cout_pad % 32 == 0, "kfn_conv2d_nhwc: bad cout_pad=%d", d->cout_pad);
  KFN_REQUIRE(d->kh > 0 && d->kw > 0 && d->kh * d->kw <= 32, "kfn_conv2d_nhwc: bad kernel %dx%d", d->kh, d->kw);
  KFN_REQUIRE(d->stride == 1 || d->stride == 2, "kfn_conv2d_nhwc: stride %d unsupported", d->stride);
  KFN_REQUIRE(!d->transposed || d->stride == 2, "kfn_conv2d_nhwc: transposed conv needs stride 2");
  KFN_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(w_packed) & 15) == 0,
              "kfn_conv2d_nhwc: x / w_packed must be 16-byte aligned");
  KFN_REQUIRE(d->epilogue != KFN_EPI_L2NORM || d->Cout == 32, "kfn_conv2d_nhwc: L2NORM epilogue needs Cout == 32");

  ConvArgs a;
  a.x = x; a.w = w_packed; a.bias = bias; a.y = y;
  a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.ldx = d->ldx;
  a.Cout = d->Cout; a.cout_pad = d->cout_pad; a.ldy = d->ldy;
  a.kh = d->kh; a.kw = d->kw; a.stride = d->stride;
  a.transposed = d->transposed; a.relu = d->relu; a.epilogue = d->epilogue;
  if (d->transposed) {
    // padding of the forward SAME conv (s*H -> H) whose input-gradient this is
    int o;
    a.Ho = d->H * d->stride;
    a.Wo = d->W * d->stride;
    kfn::same_pad(a.Ho, d->kh, d->stride, &o, &a.pad_t);
    kfn::same_pad(a.Wo, d->kw, d->stride, &o, &a.pad_l);
  } else {
    kfn::same_pad(d->H, d->kh, d->stride, &a.Ho, &a.pad_t);
    kfn::same_pad(d->W, d->kw, d->stride, &a.Wo, &a.pad_l);
  }
  long M = (long)d->N * a.Ho * a.Wo;
  KFN_REQUIRE(M < (1L << 31) && (long)d->N * d->H * d->W < (1L << 31), "kfn_conv2d_nhwc: tensor too large");
  a.M = (int)M;
  a.Ktot = d->kh * d->kw * d->Cin;
  a.tiles_m = a.tiles_n = 0;

  if (g_num_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    KFN_HIP(hipGetDevice(&dev));
    KFN_HIP(hipGetDeviceProperties(&prop, dev));
    g_num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  int cfg = d->config;
  if (cfg == KFN_CFG_AUTO) cfg = auto_config(a, g_num_cu);
  if (d->epilogue == KFN_EPI_L2NORM) cfg = KFN_CFG_128x32;  // one 32-lane half == all channels
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (d->Cin % 32 == 0) return dispatch_cfg<32>(cfg, a, s);
  return dispatch_cfg<16>(cfg, a, s);
}
