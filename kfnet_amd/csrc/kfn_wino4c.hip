// kfn_wino4c.hip -- Winograd F(4x4,3x3) in the shape of kfn_wino_s2c.hip: persistent workgroups, every wave ALL 36 positions.
//
// Reference: tf.layers.conv2d(kernel 3, strides 1, 'same') in cnn_wrapper/network.py:116-135 -- SCoordNet's conv2b / conv3b / conv4b /
// conv5 / conv6 (cnn_wrapper/SCoordNet.py:24,26,28-30).  Same algorithm, transform matrices and transformed weights as
// kfn_wino4.hip (points {0, +-1, +-2, inf}); what differs is the cut.  tools/mb_w4_kdep.py measured wino4b_kernel at 4.8 us per
// super-step + 16 us per WORKGROUP (launch, exposed prologue, the partner exchange of its epilogue -- a wave there holds half the
// positions of 32 tiles x 16 channels --, 131 KB of output stores): 15 % of that kernel's time at the bench batch.  Here
//   * a workgroup (eight waves, two per SIMD) owns 4 x 4 tiles (16 x 16 output pixels) x 128 output channels; wave w consumes ALL
//     36 positions of the 16 tiles for channels n0 + 16 w: 36 accumulators x 4 registers, no exchange in the epilogue;
//   * V of a super-step (16 input channels) is 36 slots x [4 k][16 tiles ^ k][4 k-steps] floats = 36 KiB, double-buffered 72 KiB:
//     room for a per-wave staging area BESIDE it, so the workgroup can be PERSISTENT -- it walks its XCD's share of the tile
//     blocks, and the producers' look-ahead (gathers two super-steps ahead, weight ring, V stores) runs on into the next block;
//   * producer = kfn_wino4.hip's (one channel per lane, packed transform): wave w gathers / transforms tile column w & 3 for the
//     super-steps of parity w >> 2 -- a wave alternates between a super-step of 36 gathers and one of 72 packed transform
//     instructions + 36 stores: half the producer work per MFMA of wino4b_kernel (its block has 64 channels), at twice the
//     weight stream (36 fragments per 144 MFMAs and wave);
//   * consumer order: pairs of positions, k-step major (a0 b0 a1 b1 ..): an accumulator every second MFMA (32-cycle issue, 40-cycle
//     dependent latency), V ring of two pairs, weight ring of 6 fragments refilled right behind a fragment's last MFMA.
// Launches of fewer than two workgroups per CU, layers with fewer than 128 output channels or an odd number of super-steps stay
// with wino4b_kernel (the graph routes; kfn_conv_desc.wino_form = KFN_WINO_FORM_F43_PERSISTENT, weights pack_winograd_f43_kernel_c).
#include "kfn_common.h"
#include <type_traits>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr unsigned POISON = 0x80000000u;
constexpr int NT = 128;                  // output channels per workgroup
constexpr int NPOS = 36, SS_CH = 16;
constexpr int SLOT_F = 256;              // floats per slot
constexpr int VBUF = NPOS * SLOT_F;      // floats per super-step buffer
constexpr int LDS_V = 2 * VBUF * 4;      // 73 728 B
constexpr int PSTG = 4 * (16 * 16 + 16); // floats per wave: [4 tile rows][16 px + skew][16 ch]
constexpr int LDS_C = LDS_V + 8 * PSTG * 4;   // 108 544 B
constexpr int NBQ = 6;                   // weight ring (fragments of 16 bytes per lane); divides 36: a position keeps its register
static_assert(NPOS % NBQ == 0, "the ring must not rotate from super-step to super-step");

struct W4cArgs {
  const float* x;
  const float* u;     // [Cin/16][36][cout_pad][16]
  const float* bias;
  float* y;
  int N, H, W, Cin, ldx;
  int Cout, cout_pad, ldy;
  int Th, Tw;         // tiles (4x4 output pixels) per image
  int vrows;          // N * Th
  int bw;             // tile-block columns
  int tiles_m, tiles_n;
  int relu;
  int n_group;
  unsigned long long x_bytes, y_bytes;
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

// this lane's index from the hardware, opaque to the optimiser (see kfn_wino_s2c.hip: block-level values derived on the spot)
__device__ __forceinline__ int lane_now() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

// ---- the input transform of kfn_wino4.hip (one channel per lane, packed over row pairs) ----
__device__ __forceinline__ f32x2 pk_add(const f32x2& a, const f32x2& b) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ f32x2 pk_sub(const f32x2& a, const f32x2& b) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ f32x2 pk_fma(const f32x2& a, const f32x2& k, const f32x2& c) {   // a * k + c, k wave-uniform
  f32x2 r;
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(k), "v"(c));
  return r;
}
struct BtK {
  f32x2 p4, m4, m5, p2, m2;   // pass along c
  f32x2 m41, p12, m12;        // pass along r: (-4,-1), (1,2), (-1,-2)
};
//   t0 = 4 d0 - 5 d2 + d4          t1 = (d4 - 4 d2) + (d3 - 4 d1)     t2 = (d4 - 4 d2) - (d3 - 4 d1)
//   t3 = (d4 - d2) + 2 (d3 - d1)   t4 = (d4 - d2) - 2 (d3 - d1)       t5 = 4 d1 - 5 d3 + d5
__device__ __forceinline__ void bt6(f32x2& d0, f32x2& d1, f32x2& d2, f32x2& d3, f32x2& d4, f32x2& d5, const BtK& k) {
  const f32x2 a = pk_fma(d2, k.m4, d4);
  const f32x2 b = pk_fma(d1, k.m4, d3);
  const f32x2 c = pk_sub(d4, d2);
  const f32x2 e = pk_sub(d3, d1);
  const f32x2 u = pk_fma(d2, k.m5, d4);
  const f32x2 v = pk_fma(d3, k.m5, d5);
  d0 = pk_fma(d0, k.p4, u);
  d5 = pk_fma(d1, k.p4, v);
  d1 = pk_add(a, b);
  d2 = pk_sub(a, b);
  d3 = pk_fma(e, k.p2, c);
  d4 = pk_fma(e, k.m2, c);
}
// rows paired in registers: P0 = (d0,d1), P1 = (d2,d3), P2 = (d4,d5) of one column -> pairs over the transformed row index
// (0,5), (1,3), (2,4): six instructions (kfn_wino4.hip: bt6r)
__device__ __forceinline__ void bt6r(f32x2& P0, f32x2& P1, f32x2& P2, const BtK& k) {
  f32x2 t, ac, be;
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(t) : "v"(P1), "s"(k.m5), "v"(P2));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "=v"(ac) : "v"(P1), "s"(k.m41), "v"(P2));   // lo halves: d2, d4
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[1,1,1]" : "=v"(be) : "v"(P0), "s"(k.m41), "v"(P1));   // hi halves: d1, d3
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(P0) : "v"(P0), "s"(k.p4), "v"(t));
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(P1) : "v"(be), "s"(k.p12), "v"(ac));
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(P2) : "v"(be), "s"(k.m12), "v"(ac));
}
// patch element (r, c) in the 18 pairs, and transformed position (xi, nu) after the two passes
#define W4C_IN(pp, r, c) (pp)[3 * (c) + (r) / 2][(r) & 1]
#define W4C_OUT(pp, xi, nu) (pp)[3 * (nu) + ((xi) == 0 || (xi) == 5 ? 0 : ((xi) == 1 || (xi) == 3 ? 1 : 2))][((xi) == 5 || (xi) == 3 || (xi) == 4) ? 1 : 0]

__global__ __launch_bounds__(512, 1) void wino4c_kernel(W4cArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem_4c[];
  float* const smf = reinterpret_cast<float*>(smem_4c);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // this workgroup's tile blocks (kfn_wino_s2c.hip: wino_s2c_pkernel): XCD x = blockIdx & 7 owns [xbase, xend), its workgroup
  // j = blockIdx >> 3 takes xbase + j, + wpx, ...
  const int nwg = p.tiles_m * p.tiles_n;
  const int xcd = (int)blockIdx.x & 7;
  const int q8 = nwg >> 3, r8 = nwg & 7;
  const int xbase = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int xend = xbase + q8 + (xcd < r8 ? 1 : 0);
  const int wpx = ((int)gridDim.x >> 3) + (xcd < ((int)gridDim.x & 7) ? 1 : 0);
  int L = xbase + ((int)blockIdx.x >> 3);
  if (L >= xend) return;
  const int per = p.tiles_m * p.n_group;
  const int n_super = p.Cin / SS_CH;       // even (the launcher checks)
  const int s_last = n_super - 1;

  struct Blk { int cb, vr0, img0, ty0, brk, tn; };
  auto blk_of = [&](int tile) __attribute__((always_inline)) {
    Blk b;
    const int gset = tile / per, rem_ = tile - gset * per;
    const int tm = rem_ / p.n_group;
    b.tn = gset * p.n_group + (rem_ - tm * p.n_group);
    b.cb = tm % p.bw;
    b.vr0 = (tm / p.bw) * 4;
    b.img0 = b.vr0 / p.Th;
    b.ty0 = b.vr0 - b.img0 * p.Th;
    b.brk = (p.Th - b.ty0 < 4) ? (p.Th - b.ty0) : 4;
    return b;
  };
  const __amdgpu_buffer_rsrc_t rsU = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.u), 0, p.u_bytes, 0x00020000);

  // ---- PRODUCER: tile column tc = wave & 3, super-steps of parity hp = wave >> 2; lane = (tile row lane >> 4, channel lane & 15):
  // 16 lanes read 64 contiguous bytes of a pixel, a load instruction touches four rows.  Geometry of the block being GATHERED:
  // per lane the byte offset of its six patch rows (or POISON), per wave the six columns (offset + "exists", uniform). ----
  const int tc = wave & 3, hp = wave >> 2;
  unsigned roff[6];
  unsigned cbase = 0, cmask = 0;            // (uniform) byte offset of patch column 0 (wraps for column -1), bit c = column c exists
  unsigned long long a_base = 0;
  int x_records = 0;
  auto set_producer = [&](const Blk& b) __attribute__((always_inline)) {
    const int lv = lane_now();
    const int ptr = lv >> 4, c16 = lv & 15;
    a_base = (unsigned long long)b.img0 * p.H * p.W * p.ldx * 4ull;
    const unsigned long long a_rest = p.x_bytes - a_base;
    const unsigned long long two_img = 2ull * p.H * p.W * p.ldx * 4ull;
    x_records = (int)(a_rest < two_img ? a_rest : two_img);
    const int img_rel = ptr < b.brk ? 0 : 1;
    const int ty = ptr < b.brk ? b.ty0 + ptr : ptr - b.brk;
    const int tx = b.cb * 4 + tc;
    const bool row_tile_ok = (b.vr0 + ptr < p.vrows);
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const int yy = 4 * ty - 1 + r;
      roff[r] = (row_tile_ok && (unsigned)yy < (unsigned)p.H)
                    ? (unsigned)((img_rel * p.H + yy) * p.W) * (unsigned)(p.ldx * 4) + (unsigned)(c16 * 4) : POISON;
    }
    cbase = (unsigned)((4 * tx - 1) * p.ldx * 4);
    cmask = 0;
#pragma unroll
    for (int c = 0; c < 6; ++c)
      cmask |= ((tx < p.Tw) && ((unsigned)(4 * tx - 1 + c) < (unsigned)p.W)) ? (1u << c) : 0u;
  };
  const unsigned pix_b = (unsigned)(p.ldx * 4);
  // V store address (floats inside a slot): channel c16 = 4 k + s, tile t = 4 (lane >> 4) + tc stored in row t ^ k
  const int v_st = ((lane & 15) >> 2) * 64 + (((4 * (lane >> 4) + tc) ^ ((lane & 15) >> 2)) * 4) + (lane & 3);
  // ---- CONSUMER: lane (tile row rl of the A operand / channel n0 + rl of the B operand, k = kl) ----
  const int v_rd = ((lane >> 4) * 16 + ((lane & 15) ^ (lane >> 4))) * 4;
  const unsigned b_step = (unsigned)p.cout_pad * 64u;
  auto voff_of = [&](int tn) __attribute__((always_inline)) {
    const int lv = lane_now();
    const int nn = tn * NT + wave * 16 + (lv & 15);
    const int nb = nn < p.cout_pad ? nn : p.cout_pad - 1;
    return (unsigned)((nb * 16 + (lv >> 4) * 4) * 4);
  };

  f32x4 acc[NPOS];
  f32x2 pp[18];        // one channel's 6x6 patch, rows paired
  f32x4 bq[NBQ];
  f32x4 vq[2][2];
  const BtK kk = {{4.f, 4.f}, {-4.f, -4.f}, {-5.f, -5.f}, {2.f, 2.f}, {-2.f, -2.f}, {-4.f, -1.f}, {1.f, 2.f}, {-1.f, -2.f}};
  auto acc_init = [&](int tn) __attribute__((always_inline)) {
    const int n = tn * NT + wave * 16 + (lane_now() & 15);
    const float bv = (p.bias != nullptr && n < p.Cout) ? p.bias[n] : 0.f;
#pragma unroll
    for (int g = 0; g < NPOS; ++g) {
      const float v0 = g == 7 ? bv : 0.f;      // A^T e_1 = (1,1,1,1): M[1][1] = b gives every output + b
      acc[g] = f32x4{v0, v0, v0, v0};
    }
  };
  auto b_load = [&](auto rc, int fq, unsigned voff) __attribute__((always_inline)) {
    constexpr int r = decltype(rc)::value;
    bq[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsU, voff, (unsigned)fq * b_step, 0));
  };
  auto p_gather = [&](auto ic, int ss) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    constexpr int r = i / 6, c = i % 6;
    const bool ok = (cmask >> c) & 1u;                             // (uniform: a column outside the image reads through an empty descriptor)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.x)) + a_base, 0, ok ? x_records : 0, 0x00020000);
    W4C_IN(pp, r, c) = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, roff[r], ok ? cbase + (unsigned)c * pix_b + (unsigned)(ss * 64) : 0u, 0));
  };
  auto p_transform = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < 6; ++c) bt6r(pp[3 * c], pp[3 * c + 1], pp[3 * c + 2], kk);
#pragma unroll
    for (int m = 0; m < 3; ++m) bt6(pp[m], pp[3 + m], pp[6 + m], pp[9 + m], pp[12 + m], pp[15 + m], kk);
  };
  auto p_store = [&](auto gc, float* stB) __attribute__((always_inline)) {      // position g = 6 xi + nu
    constexpr int g = decltype(gc)::value;
    stB[g * SLOT_F] = W4C_OUT(pp, g / 6, g % 6);
  };
  auto v_read = [&](auto pc, const float* rdB) __attribute__((always_inline)) {
    constexpr int pq = decltype(pc)::value;                        // position; pair pq / 2 -> ring half (pq / 2) & 1
    vq[(pq >> 1) & 1][pq & 1] = *reinterpret_cast<const f32x4*>(rdB + pq * SLOT_F);
  };

  // ---- block-loop state ----
  Blk cur = blk_of(L), nxt_b = cur;
  bool has_next = L + wpx < xend;
  if (has_next) nxt_b = blk_of(L + wpx);
  set_producer(cur);
  acc_init(cur.tn);

  // ---- one super-step of the consumer stream + this wave's producer role in it.  ROLE 0: gather (for local super-step ss_g, of the
  // block the producer geometry points at); ROLE 1: transform the patch gathered in the previous super-step and store it into
  // buffer `stB` (= V of the next super-step).  Weight look-ahead: fragment (position q + NBQ) behind position q's last MFMA --
  // of this super-step, or of (ss_w, voff_w) = the next one (possibly the next block's first). ----
  auto super_step = [&](auto role_c, int ks, int ss_g, int ss_w, unsigned voff_b, unsigned voff_w, const float* rdB, float* stB)
      __attribute__((always_inline)) {
    constexpr int ROLE = decltype(role_c)::value;
    sfor<2>([&](auto kc) { v_read(kc, rdB); });
    sfor<4 * NPOS>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      constexpr int m = j / 8, kst = (j % 8) / 2, k = j % 2;      // pair, k-step, position of the pair
      constexpr int pq = 2 * m + k;
      acc[pq] = __builtin_amdgcn_mfma_f32_16x16x4f32(vq[m & 1][k][kst], bq[pq % NBQ][kst], acc[pq], 0, 0, 0);
      // V of the next pair (the other ring half) during this pair's first k-step
      if constexpr (kst == 0 && m + 1 < NPOS / 2) v_read(std::integral_constant<int, 2 * (m + 1) + k>{}, rdB);
      // weights: position pq + NBQ takes this fragment's register behind its last k-step
      if constexpr (kst == 3) {
        if constexpr (pq + NBQ < NPOS) b_load(std::integral_constant<int, pq % NBQ>{}, ks * NPOS + pq + NBQ, voff_b);
        else b_load(std::integral_constant<int, pq % NBQ>{}, ss_w * NPOS + pq + NBQ - NPOS, voff_w);
      }
      if constexpr (ROLE == 0) {
        if constexpr (j >= 4 && j < 4 + 36 * 3 && (j - 4) % 3 == 0) p_gather(std::integral_constant<int, (j - 4) / 3>{}, ss_g);
      } else {
        if constexpr (j == 6) p_transform();
        if constexpr (j >= 24 && j < 24 + 36 * 3 && (j - 24) % 3 == 0) p_store(std::integral_constant<int, (j - 24) / 3>{}, stB);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  const bool relu = p.relu != 0;
  float* const stg = smf + LDS_V / 4 + wave * PSTG;
  const int pix_bytes = p.ldy * 4;
  // 1-D output transform A^T (4 x 6): (m0..m5) -> (y0..y3)
  auto at6 = [](float m0, float m1, float m2, float m3, float m4, float m5, float& y0, float& y1, float& y2, float& y3) __attribute__((always_inline)) {
    const float s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
    y0 = (m0 + s1) + s2;
    y1 = d1 + 2.0f * d2;
    y2 = s1 + 4.0f * s2;
    y3 = (d1 + 8.0f * d2) + m5;
  };

  // The whole walk sits inside each producer group's branch (kfn_wino_s2c.hip: with the block loop outside the branches hipcc
  // spills what is live across their merges).  Group hp produces V of the super-steps of parity hp:
  //   hp = 0: even super-steps GATHER (for ks + 2), odd ones TRANSFORM + STORE (V of ks + 1);  hp = 1: the other way round.
  auto walk = [&](auto hp_c) __attribute__((always_inline)) {
    constexpr int HP = decltype(hp_c)::value;
    // prologue of the first block: V(0) by group 0, the gathers of V(1) by group 1, the first NBQ weight fragments by everybody
    {
      const unsigned voff0 = voff_of(cur.tn);
      sfor<NBQ>([&](auto rc) { b_load(rc, decltype(rc)::value, voff0); });
      if constexpr (HP == 0) {
        sfor<36>([&](auto ic) { p_gather(ic, 0); });
        p_transform();
        sfor<36>([&](auto gc) { p_store(gc, smf + v_st); });
      } else {
        sfor<36>([&](auto ic) { p_gather(ic, 1); });
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    for (;;) {
      for (int ks = 0; ks < n_super; ks += 2) {
        const unsigned voff_b = voff_of(cur.tn);
        const unsigned voff_n = has_next ? voff_of(nxt_b.tn) : voff_b;
        const float* const rd0 = smf + v_rd;                 // V(ks): buffer 0 (n_super is even: parity of ks = parity of the buffer)
        const float* const rd1 = smf + VBUF + v_rd;
        float* const st0 = smf + v_st;
        float* const st1 = smf + VBUF + v_st;
        const bool last_pair = ks + 2 >= n_super;
        // ---- even super-step ks: weights of ks + 1 (same block); group 0 gathers ks + 2 (or the next block's 0), group 1
        // transforms + stores V(ks + 1) ----
        if constexpr (HP == 0) {
          if (last_pair && has_next) set_producer(nxt_b);    // every gather of this block has been issued
          const int ss_g = !last_pair ? ks + 2 : (has_next ? 0 : s_last);
          super_step(std::integral_constant<int, 0>{}, ks, ss_g, ks + 1, voff_b, voff_b, rd0, st1);
        } else {
          super_step(std::integral_constant<int, 1>{}, ks, 0, ks + 1, voff_b, voff_b, rd0, st1);
        }
        // ---- odd super-step ks + 1: weights of ks + 2 (or the next block's 0); group 0 transforms + stores V(ks + 2), group 1
        // gathers ks + 3 (or the next block's 1) ----
        const int ss_w = !last_pair ? ks + 2 : (has_next ? 0 : s_last);
        const unsigned voff_w = last_pair ? voff_n : voff_b;
        if constexpr (HP == 0) {
          super_step(std::integral_constant<int, 1>{}, ks + 1, 0, ss_w, voff_b, voff_w, rd1, st0);
        } else {
          if (last_pair && has_next) set_producer(nxt_b);
          const int ss_g = !last_pair ? ks + 3 : (has_next ? 1 : s_last);
          super_step(std::integral_constant<int, 0>{}, ks + 1, ss_g, ss_w, voff_b, voff_w, rd1, st0);
        }
      }
      // ---- epilogue of block `cur`: Y = A^T M A per tile; lane (channel n0 + rl, k = kl), element e = tile (row kl, column e);
      // one pass per tile COLUMN through the wave's staging area [4 tile rows][16 px][16 ch] (kfn_wino_s2c.hip) ----
      {
        const int lv = lane_now();
        const int rl = lv & 15, kl = lv >> 4;
        const int n0 = cur.tn * NT + wave * 16;
        const unsigned long long y_base = (unsigned long long)cur.img0 * p.H * p.W * p.ldy * 4ull;
        const unsigned long long y_rest = p.y_bytes - y_base;
        const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<char*>(p.y) + y_base, 0, (int)(y_rest < 0x7fffffffull ? y_rest : 0x7fffffffull), 0x00020000);
        const int spx = lv >> 2, nq = lv & 3;                    // store lane: pixel (i, j) = (spx >> 2, spx & 3) of a tile, channel quad
        const int pi = spx >> 2, pj = spx & 3;
        const bool q_ok = n0 + nq * 4 < p.Cout;
        const unsigned voff_q = (unsigned)(((pi * p.W + pj) * p.ldy + n0 + nq * 4) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t[4][6];
#pragma unroll
          for (int nu = 0; nu < 6; ++nu)
            at6(acc[nu][e], acc[6 + nu][e], acc[12 + nu][e], acc[18 + nu][e], acc[24 + nu][e], acc[30 + nu][e], t[0][nu], t[1][nu], t[2][nu], t[3][nu]);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float y0, y1, y2, y3;
            at6(t[i][0], t[i][1], t[i][2], t[i][3], t[i][4], t[i][5], y0, y1, y2, y3);
            float* const row = stg + kl * (16 * 16 + 16) + (i * 4) * 16 + rl;
            row[0] = y0; row[16] = y1; row[32] = y2; row[48] = y3;
          }
          __builtin_amdgcn_wave_barrier();
          const int tx = cur.cb * 4 + e;
          const bool col_ok = q_ok && tx < p.Tw && 4 * tx + pj < p.W;      // (per lane: the image may end inside a tile)
#pragma unroll
          for (int k4 = 0; k4 < 4; ++k4) {
            f32x4 v = *reinterpret_cast<const f32x4*>(stg + k4 * (16 * 16 + 16) + spx * 16 + nq * 4);
            if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            const int img_rel = k4 < cur.brk ? 0 : 1;
            const int ty = k4 < cur.brk ? cur.ty0 + k4 : k4 - cur.brk;
            const bool ok = col_ok && cur.vr0 + k4 < p.vrows && 4 * ty + pi < p.H;
            const unsigned soff = (unsigned)(((img_rel * p.H + 4 * ty) * p.W + 4 * tx) * pix_bytes);
            kfn::buffer_store_b128<KFN_NT_STORE_AUX>(v, rsY, ok ? voff_q : POISON, ok ? soff : 0u);
          }
          __builtin_amdgcn_wave_barrier();
        }
      }
      if (!has_next) break;
      L += wpx;
      cur = nxt_b;
      has_next = L + wpx < xend;
      if (has_next) nxt_b = blk_of(L + wpx);
      acc_init(cur.tn);
    }
  };
  if (hp == 0) walk(std::integral_constant<int, 0>{});
  else walk(std::integral_constant<int, 1>{});
}

}  // namespace

int kfn::wino4c_lds_bytes() { return LDS_C; }

// pointer-free conditions of the persistent F(4x4,3x3) form (kfn_winograd_f43_supported answers them for wino_form =
// KFN_WINO_FORM_F43_PERSISTENT on top of the common ones)
int kfn::wino4c_supported(const kfn_conv_desc* d) {
  if ((d->Cin / SS_CH) % 2 != 0 || d->Cin < 2 * SS_CH) return 0;       // super-steps come in (gather, transform) pairs
  if (d->Cout < NT || d->cout_pad % 32 != 0) return 0;                 // eight waves x 16 channels
  if ((d->H + 3) / 4 < 4) return 0;                                    // a 4-row tile block straddles at most two images
  return 1;
}

int kfn::launch_wino4c(const kfn_conv_desc* d, const float* x, const float* u_packed, const float* bias, float* y, void* stream) {
  // (the common checks -- dtypes, strides, alignment, 32-bit ranges -- were made by kfn_conv2d_winograd_f43's set-up)
  if (!wino4c_supported(d))
    return kfn::fail(KFN_ERR_UNSUPPORTED, "kfn_conv2d_winograd_f43 (persistent form): needs Cout >= 128, an even number >= 2 of 16-channel "
                     "super-steps and H >= 13 (got Cin=%d Cout=%d H=%d)", d->Cin, d->Cout, d->H);
  W4cArgs a;
  a.x = x; a.u = u_packed; a.bias = bias; a.y = y;
  a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.ldx = d->ldx;
  a.Cout = d->Cout; a.cout_pad = d->cout_pad; a.ldy = d->ldy;
  a.Th = (d->H + 3) / 4; a.Tw = (d->W + 3) / 4;
  a.vrows = d->N * a.Th;
  a.bw = kfn::ceil_div(a.Tw, 4);
  const long tiles_m = (long)a.bw * kfn::ceil_div(a.vrows, 4);
  a.tiles_n = kfn::ceil_div(d->cout_pad, NT);
  KFN_REQUIRE(tiles_m * a.tiles_n < (1L << 31), "kfn_conv2d_winograd_f43 (persistent form): grid too large");
  a.tiles_m = (int)tiles_m;
  a.relu = d->relu;
  {
    int ng = a.tiles_n % 2 == 0 ? 2 : 1;
    if (d->wino_order == KFN_WINO_ORDER_N_FAST) ng = a.tiles_n;
    else if (d->wino_order == KFN_WINO_ORDER_M_FAST) ng = 1;
    else if (d->wino_order >= KFN_WINO_ORDER_GROUPS(1)) ng = d->wino_order - KFN_WINO_ORDER_GROUPS(0);
    if (ng < 1 || ng > a.tiles_n || a.tiles_n % ng != 0) ng = 1;
    a.n_group = ng;
  }
  const long in_pix = (long)d->N * d->H * d->W;
  a.x_bytes = (unsigned long long)(((in_pix - 1) * d->ldx + d->Cin) * 4L);
  a.y_bytes = (unsigned long long)(((in_pix - 1) * d->ldy + d->Cout) * 4L);
  a.u_bytes = (unsigned)(36L * d->cout_pad * d->Cin * 4L);
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  const long nwg = tiles_m * a.tiles_n;
  const unsigned grid = (unsigned)(nwg < n_cu ? nwg : n_cu);
  static std::atomic<uint64_t> attr_done{0};
  int rc = kfn::set_max_dynamic_lds(reinterpret_cast<const void*>(wino4c_kernel), LDS_C, attr_done);
  if (rc != KFN_OK) return rc;
  hipLaunchKernelGGL(wino4c_kernel, dim3(grid), dim3(512), LDS_C, (hipStream_t)stream, a);
  KFN_LAUNCH_CHECK("wino4c_kernel");
  return KFN_OK;
}
