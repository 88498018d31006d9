// kfn_wino16.hip -- fully fused Winograd F(2x2,3x3) convolution (3x3, stride 1, SAME).
//
// One workgroup (4 wavefronts, one per SIMD) owns 64 output tiles (2x2 pixels each) x 64
// output channels and ALL 16 transform positions (xi,nu): each wavefront accumulates
// 16 x (32 tiles x 32 channels) in 256 accumulator registers per lane, so the inverse
// transform Y = A^T M A is evaluated in registers in the epilogue and the
// [tiles][16][Cout] workspace of the two-kernel path (kfn_conv2d_winograd) never exists;
// every source pixel is fetched once per workgroup (not once per (xi,nu) group).
//
// Per k-step of 8 input channels:
//   waves 0-1: each thread loads the 4x4 raw patch of one (tile, channel quad) -- 16 range-
//              checked buffer loads (zero padding for free) --, evaluates B^T d B (32 float4
//              adds) and stores the 16 transformed quads to LDS V[g][tile][8];
//   waves 2-3: each thread copies 16 quads of the pre-transformed weights U[g][cout][8];
//   all waves: 16 groups x 4 MFMAs (v_mfma_f32_32x32x2_f32; lane (i,h) feeds k = 4h..4h+3).
// Global loads run two stages ahead in two alternating register sets; the transform + LDS
// stores of stage s+1 and the loads of stage s+2 are slotted between the MFMAs of stage s.
#include "kfn_common.h"
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned OOB16 = 0x80000000u;
constexpr int WT = 64;          // wino tiles per workgroup
constexpr int WC = 64;          // output channels per workgroup
constexpr int WK = 8;           // input channels per stage
constexpr int V_ELEMS = 16 * WT * WK;   // floats per V buffer
constexpr int U_ELEMS = 16 * WC * WK;

struct Wino16Args {
  const float* x;
  const float* u;     // [16][cout_pad][Cin]
  const float* bias;
  float* y;
  int N, H, W, Cin, ldx;
  int Cout, cout_pad, ldy;
  int Th, Tw, Mt;
  int relu;
  int tiles_m, tiles_n;
  unsigned long long x_bytes;
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

__device__ __forceinline__ f32x4 bload(__amdgpu_buffer_rsrc_t rsrc, unsigned voff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0));
}

__device__ __forceinline__ int xcd_remap16(int b, int nwg) {
  int xcd = b & 7;
  int q = nwg >> 3, r = nwg & 7;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (b >> 3);
}

__global__ __launch_bounds__(256, 1) void wino16_kernel(Wino16Args p) {
  extern __shared__ __attribute__((aligned(16))) float smem16[];
  float* Vs = smem16;                 // [2][16][WT][WK]
  float* Us = smem16 + 2 * V_ELEMS;   // [2][16][WC][WK]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;       // MFMA role: 2 (tiles) x 2 (channels)
  const bool a_loader = wave < 2;                // staging role

  // logical tile: m fastest, so that one XCD's contiguous run mostly shares one U slice
  const int nwg = p.tiles_m * p.tiles_n;
  const int tile = xcd_remap16(blockIdx.x, nwg);
  const int tm = tile % p.tiles_m;
  const int tn = tile / p.tiles_m;
  const int m0 = tm * WT;
  const int n0 = tn * WC;

  // ---- staging state ---------------------------------------------------------------
  // A loader: unit = (tile row, channel quad); B loader: unit = (cout row, channel quad)
  const int unit = tid & 127;
  const int urow = unit >> 1, uquad = unit & 1;
  const int ThTw = p.Th * p.Tw;
  const int n_first = m0 / ThTw;
  const unsigned long long a_base = (unsigned long long)n_first * p.H * p.W * p.ldx * 4ull;
  const unsigned long long a_rest = p.x_bytes - a_base;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(p.x)) + a_base, 0,
      (int)(a_rest < 0x7fffffffull ? a_rest : 0x7fffffffull), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsU =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.u), 0, p.u_bytes, 0x00020000);

  unsigned a_off = 0;      // byte offset of patch pixel (0,0), channel quad uquad
  unsigned a_valid = 0;    // bit (r*4+c): patch pixel inside the image
  unsigned row_stride = (unsigned)(p.W * p.ldx) * 4u, px_stride = (unsigned)p.ldx * 4u;
  unsigned b_off = OOB16;  // byte offset of U[0][n0+urow][uquad*4]
  const unsigned g_stride = (unsigned)(p.cout_pad * p.Cin) * 4u;
  if (a_loader) {
    const int m = m0 + urow;
    if (m < p.Mt) {
      const int n_abs = m / ThTw;
      const int rem = m - n_abs * ThTw;
      const int ty = rem / p.Tw, tx = rem - ty * p.Tw;
      const int y0 = 2 * ty - 1, x0 = 2 * tx - 1;
      a_off = (unsigned)((((n_abs - n_first) * p.H + y0) * p.W + x0) * p.ldx + uquad * 4) * 4u;
      for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c)
          if ((unsigned)(y0 + r) < (unsigned)p.H && (unsigned)(x0 + c) < (unsigned)p.W) a_valid |= 1u << (r * 4 + c);
    }
  } else {
    const int co = n0 + urow;
    if (co < p.cout_pad) b_off = (unsigned)(co * p.Cin + uquad * 4) * 4u;
  }

  // LDS store position (floats) of this unit's quad within group 0; groups are WT*WK (or WC*WK) apart
  const int st_off = urow * WK + ((uquad ^ ((urow >> 3) & 1)) * 4);

  // fragment read positions
  const int li = lane & 31, lh = lane >> 5;
  const int rd_q = ((lh ^ ((li >> 3) & 1)) * 4);
  const int v_rd = (wm * 32 + li) * WK + rd_q;
  const int u_rd = (wn * 32 + li) * WK + rd_q;

  f32x16 acc[16];
#pragma unroll
  for (int g = 0; g < 16; ++g)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[g][e] = 0.f;

  const int n_stages = p.Cin / WK;
  f32x4 ra[16];   // staging registers

  auto issue_loads = [&](f32x4 (&r)[16], int s) __attribute__((always_inline)) {
    const bool live = s < n_stages;
    const unsigned c0b = (unsigned)(s * WK) * 4u;
    if (a_loader) {
      sfor<16>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        constexpr int rr = k >> 2, cc = k & 3;
        const bool ok = live && ((a_valid >> k) & 1u);
        r[k] = bload(rsA, ok ? a_off + (unsigned)rr * row_stride + (unsigned)cc * px_stride + c0b : OOB16);
      });
    } else {
      sfor<16>([&](auto kc) {
        constexpr int g = decltype(kc)::value;
        r[g] = bload(rsU, (live && b_off != OOB16) ? b_off + (unsigned)g * g_stride + c0b : OOB16);
      });
    }
  };

  // B^T d B on the 16 raw quads (in place), then 16 LDS stores
  auto transform_store = [&](f32x4 (&r)[16], int buf) __attribute__((always_inline)) {
    if (a_loader) {
      // columns: t[r][0] = d0 - d2, t[r][1] = d1 + d2, t[r][2] = d2 - d1, t[r][3] = d1 - d3
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const f32x4 d0 = r[rr * 4 + 0], d1 = r[rr * 4 + 1], d2 = r[rr * 4 + 2], d3 = r[rr * 4 + 3];
        r[rr * 4 + 0] = d0 - d2;
        r[rr * 4 + 1] = d1 + d2;
        r[rr * 4 + 2] = d2 - d1;
        r[rr * 4 + 3] = d1 - d3;
      }
      // rows: V[0] = t0 - t2, V[1] = t1 + t2, V[2] = t2 - t1, V[3] = t1 - t3
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        const f32x4 t0 = r[0 * 4 + cc], t1 = r[1 * 4 + cc], t2 = r[2 * 4 + cc], t3 = r[3 * 4 + cc];
        r[0 * 4 + cc] = t0 - t2;
        r[1 * 4 + cc] = t1 + t2;
        r[2 * 4 + cc] = t2 - t1;
        r[3 * 4 + cc] = t1 - t3;
      }
      float* dst = Vs + buf * V_ELEMS + st_off;
#pragma unroll
      for (int g = 0; g < 16; ++g) *reinterpret_cast<f32x4*>(dst + g * WT * WK) = r[g];
    } else {
      float* dst = Us + buf * U_ELEMS + st_off;
#pragma unroll
      for (int g = 0; g < 16; ++g) *reinterpret_cast<f32x4*>(dst + g * WC * WK) = r[g];
    }
  };

  auto compute = [&](int buf) __attribute__((always_inline)) {
    const float* v = Vs + buf * V_ELEMS + v_rd;
    const float* u = Us + buf * U_ELEMS + u_rd;
    f32x4 fa = *reinterpret_cast<const f32x4*>(v);
    f32x4 fb = *reinterpret_cast<const f32x4*>(u);
    sfor<16>([&](auto gc) {
      constexpr int g = decltype(gc)::value;
      f32x4 na = fa, nb = fb;
      if constexpr (g + 1 < 16) {
        na = *reinterpret_cast<const f32x4*>(v + (g + 1) * WT * WK);
        nb = *reinterpret_cast<const f32x4*>(u + (g + 1) * WC * WK);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
        acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t], fb[t], acc[g], 0, 0, 0);
      fa = na;
      fb = nb;
      // keep the fragment prefetch exactly one group deep (hoisting all 32 reads costs 128 VGPRs)
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  // ---- pipeline ----------------------------------------------------------------------
  // single staging set: loads of stage s+1 are issued before the MFMAs of stage s and are
  // transformed + stored behind them (one barrier per stage)
  issue_loads(ra, 0);
  transform_store(ra, 0);
  __syncthreads();
  for (int s = 0; s < n_stages; ++s) {
    const int buf = s & 1;
    issue_loads(ra, s + 1);
    compute(buf);
    transform_store(ra, buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: Y = A^T M A per (tile, channel), bias, ReLU, store ---------------------
  const int n = n0 + wn * 32 + li;
  const bool n_ok = n < p.Cout;
  const float bv = (p.bias != nullptr && n_ok) ? p.bias[n] : 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int m = m0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
    float r0[4], r1[4];
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) {
      r0[nu] = (acc[0 * 4 + nu][e] + acc[1 * 4 + nu][e]) + acc[2 * 4 + nu][e];
      r1[nu] = (acc[1 * 4 + nu][e] - acc[2 * 4 + nu][e]) - acc[3 * 4 + nu][e];
    }
    float o[4];
    o[0] = (r0[0] + r0[1]) + r0[2];
    o[1] = (r0[1] - r0[2]) - r0[3];
    o[2] = (r1[0] + r1[1]) + r1[2];
    o[3] = (r1[1] - r1[2]) - r1[3];
    if (n_ok && m < p.Mt) {
      const int n_img = m / ThTw;
      const int rem = m - n_img * ThTw;
      const int ty = rem / p.Tw, tx = rem - ty * p.Tw;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int oy = 2 * ty + a, ox = 2 * tx + b;
          if (oy < p.H && ox < p.W) {
            float v = o[a * 2 + b] + bv;
            if (p.relu) v = fmaxf(v, 0.f);
            p.y[((size_t)(n_img * p.H + oy) * p.W + ox) * p.ldy + n] = v;
          }
        }
    }
  }
}

}  // namespace

extern "C" int kfn_conv2d_winograd_fused(const kfn_conv_desc* d, const float* x, const float* u_packed,
                                         const float* bias, float* y, void* stream) {
  KFN_REQUIRE(d && x && u_packed && y, "kfn_conv2d_winograd_fused: null argument");
  KFN_REQUIRE(d->kh == 3 && d->kw == 3 && d->stride == 1 && !d->transposed,
              "kfn_conv2d_winograd_fused: only 3x3 stride-1 SAME convolutions");
  KFN_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cin % 8 == 0,
              "kfn_conv2d_winograd_fused: bad shape (Cin %% 8 == 0 required)");
  KFN_REQUIRE(d->ldx >= d->Cin && d->ldx % 4 == 0 && d->Cout > 0 && d->ldy >= d->Cout && d->cout_pad >= d->Cout,
              "kfn_conv2d_winograd_fused: bad strides / channel counts");
  KFN_REQUIRE(d->epilogue == KFN_EPI_NONE && d->operand_dtype == KFN_OPERAND_F32,
              "kfn_conv2d_winograd_fused: fp32, no fused head epilogue");
  KFN_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(u_packed)) & 15) == 0,
              "kfn_conv2d_winograd_fused: buffers must be 16-byte aligned");
  Wino16Args a;
  a.x = x; a.u = u_packed; a.bias = bias; a.y = y;
  a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.ldx = d->ldx;
  a.Cout = d->Cout; a.cout_pad = d->cout_pad; a.ldy = d->ldy;
  a.Th = (d->H + 1) / 2; a.Tw = (d->W + 1) / 2;
  const long Mt = (long)d->N * a.Th * a.Tw;
  const long in_pix = (long)d->N * d->H * d->W;
  const long x_bytes = ((in_pix - 1) * d->ldx + d->Cin) * 4L;
  const long u_bytes = 16L * d->cout_pad * d->Cin * 4L;
  const long img_bytes = (long)d->H * d->W * d->ldx * 4L;
  const long imgs_per_tile = WT / ((long)a.Th * a.Tw) + 2;
  KFN_REQUIRE(Mt < (1L << 31) && u_bytes < (1L << 31) && img_bytes * imgs_per_tile < (1L << 31),
              "kfn_conv2d_winograd_fused: tensor too large for 32-bit buffer addressing");
  a.Mt = (int)Mt;
  a.relu = d->relu;
  a.tiles_m = kfn::ceil_div(a.Mt, WT);
  a.tiles_n = kfn::ceil_div(d->Cout, WC);
  a.x_bytes = (unsigned long long)x_bytes;
  a.u_bytes = (unsigned)u_bytes;
  constexpr size_t smem = (size_t)2 * (V_ELEMS + U_ELEMS) * sizeof(float);
  static std::atomic<uint64_t> attr_done{0};
  {
    int rc = kfn::set_max_dynamic_lds(reinterpret_cast<const void*>(wino16_kernel), (int)smem, attr_done);
    if (rc != KFN_OK) return rc;
  }
  hipLaunchKernelGGL(wino16_kernel, dim3(a.tiles_m * a.tiles_n), dim3(256), smem, (hipStream_t)stream, a);
  KFN_LAUNCH_CHECK("wino16_kernel");
  return KFN_OK;
}
