// tools/mb/kalman_mb.hip -- A/B micro-benchmark of the Kalman kernels (csrc/kfn_kalman.hip) on the GPU box: every variant is
// an instantiation of the SAME templates the library ships, timed with HIP events on identical buffers, results compared
// bit for bit with the first variant.  Build (cross-compiles without a GPU):
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/mb/kalman_mb.hip kfnet_amd/csrc/kfn_runtime.hip -o tools/mb/kalman_mb
//     tools/mb/kalman_mb [S=256] [T=64]
#include "../../kfnet_amd/csrc/kfn_kalman.hip"

#include <vector>
#include <string>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

namespace {
// ---- the scan kernel as rounds 1-4 shipped it (burst prefetch of frame t+1 into a second register set, conditional loads),
// kept here as the in-run baseline of the A/B ----
// DBL: the state is double-buffered in LDS (2 x 76.8 KB at 60x80): frame t gathers from
// buffer t&1 and writes the fused state straight into the other one -- one barrier per
// frame and no per-thread copy of the new state.  Grids whose two copies exceed the
// 160 KB LDS use the single-buffer form (fuse into registers, barrier, write back, barrier).
template <int KT, int PPT, bool DBL, bool PREFETCH>
__global__ __launch_bounds__(KT) void kalman_scan_kernel_r4(KalmanArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem_k[];
  f32x4* st_base = reinterpret_cast<f32x4*>(smem_k);
  const int tid = threadIdx.x;
  const int s = blockIdx.x;
  const int H = a.d.H, W = a.d.W, HW = H * W, T = a.d.T;
  const float eps2 = a.d.min_uncertainty * a.d.min_uncertainty;
  const float xmax = (float)(W - 1), ymax = (float)(H - 1);
  const bool want_nis = (a.opt_nis != nullptr) || (a.d.nis_gate > 0.f);

  // state -> LDS
  for (int p = tid; p < HW; p += KT) st_base[p] = a.state[(size_t)s * HW + p];

  const size_t seq_off = (size_t)s * T * HW;
  PixIn cur[PPT], nxt[PREFETCH ? PPT : 1];
  auto load_inputs = [&](int t, auto& dst) {
    const size_t off = seq_off + (size_t)t * HW;
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      int p = tid + k * KT;
      if (p < HW) {
        dst[k].flow = a.flow[off + p];
        dst[k].st = a.sigma_t[off + p];
        dst[k].z = a.meas[off + p];
      }
    }
  };
  if (PREFETCH) load_inputs(0, cur);
  __syncthreads();

  for (int t = 0; t < T; ++t) {
    if constexpr (PREFETCH) {
      if (t + 1 < T) load_inputs(t + 1, nxt);
    } else {
      load_inputs(t, cur);
    }
    const int gi = a.d.t0 + t;
    const bool reset = a.d.reset_period > 0 && (gi % a.d.reset_period) == 0;
    const size_t off = seq_off + (size_t)t * HW;
    const f32x4* st = DBL ? st_base + (t & 1) * HW : st_base;
    f32x4* st_new = DBL ? st_base + ((t + 1) & 1) * HW : st_base;
    f32x4 newst[DBL ? 1 : PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      const int p = tid + k * KT;
      if (p < HW) {
        const f32x4 nv = fuse_pixel<false, true, false>(a, st, cur[k], p, p % W, p / W, off, reset, W, xmax, ymax, eps2, want_nis);
        if (DBL) st_new[p] = nv; else newst[DBL ? 0 : k] = nv;
      }
      // keep the unrolled pixels sequential: interleaving them only multiplies live
      // temporaries (the 128-VGPR budget of a 1024-thread workgroup is tight)
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
    if constexpr (PREFETCH) {
#pragma unroll
      for (int k = 0; k < PPT; ++k) cur[k] = nxt[k];
    }
    __syncthreads();  // new state visible, old buffer free
  }
  const f32x4* st_fin = DBL ? st_base + (T & 1) * HW : st_base;
  for (int p = tid; p < HW; p += KT) a.state[(size_t)s * HW + p] = st_fin[p];
}

template <int KT, int PPT, bool DBL, bool PREFETCH>
int launch_scan_r4(const KalmanArgs& a, hipStream_t stream) {
  const size_t smem = (size_t)a.d.H * a.d.W * sizeof(f32x4) * (DBL ? 2 : 1);
  auto kern = kalman_scan_kernel_r4<KT, PPT, DBL, PREFETCH>;
  static std::atomic<uint64_t> attr_done{0};
  int rc = kfn::set_max_dynamic_lds(reinterpret_cast<const void*>(kern), 160 * 1024, attr_done);
  if (rc != KFN_OK) return rc;
  hipLaunchKernelGGL(kern, dim3(a.d.S), dim3(KT), smem, stream, a);
  return KFN_OK;
}

struct Variant { std::string name; int (*launch)(const KalmanArgs&, hipStream_t); };

template <int BLOCK, int U, bool NT>
int launch_fuse(const f32x4* pred, const f32x4* meas, f32x4* out, long P, hipStream_t st) {
  constexpr int PER = BLOCK * U;
  long blocks = (P + PER - 1) / PER;
  if (blocks > 256L * 16) blocks = 256L * 16;
  hipLaunchKernelGGL((kalman_fuse_kernel<BLOCK, U, NT>), dim3((unsigned)blocks), dim3(BLOCK), 0, st, pred, meas, out, (float*)nullptr, P);
  return 0;
}
template <int BLOCK, int U, bool NT>
int launch_fuse_full(const f32x4* pred, const f32x4* meas, f32x4* out, long P, hipStream_t st) {   // one trip per thread: no grid-stride loop
  constexpr int PER = BLOCK * U;
  long blocks = (P + PER - 1) / PER;
  hipLaunchKernelGGL((kalman_fuse_kernel<BLOCK, U, NT>), dim3((unsigned)blocks), dim3(BLOCK), 0, st, pred, meas, out, (float*)nullptr, P);
  return 0;
}
struct FuseVariant { std::string name; int (*launch)(const f32x4*, const f32x4*, f32x4*, long, hipStream_t); };
}  // namespace

int main(int argc, char** argv) {
  const int S = argc > 1 ? atoi(argv[1]) : 256, T = argc > 2 ? atoi(argv[2]) : 64;
  const int H = argc > 3 ? atoi(argv[3]) : 60, W = argc > 4 ? atoi(argv[4]) : 80;
  const size_t HW = (size_t)H * W, N = (size_t)S * T * HW;
  std::vector<float> h_flow(N * 2), h_sig(N), h_meas(N * 4), h_state((size_t)S * HW * 4);
  unsigned rs = 12345u;
  auto rnd = [&]() { rs = rs * 1664525u + 1013904223u; return (float)((rs >> 8) & 0xFFFF) / 65536.0f; };
  for (auto& v : h_flow) v = (rnd() - 0.5f) * 6.0f;
  for (auto& v : h_sig) v = rnd() * 0.05f + 0.001f;
  for (size_t i = 0; i < N; ++i) { for (int c = 0; c < 3; ++c) h_meas[i * 4 + c] = rnd() * 2.f - 1.f; h_meas[i * 4 + 3] = rnd() * 0.3f + 0.05f; }
  for (size_t i = 0; i < (size_t)S * HW; ++i) { for (int c = 0; c < 3; ++c) h_state[i * 4 + c] = rnd(); h_state[i * 4 + 3] = rnd() * 0.3f + 0.05f; }
  float *d_flow, *d_sig, *d_meas, *d_state, *d_state0, *d_rec, *d_ref;
  CK(hipMalloc(&d_flow, N * 8)); CK(hipMalloc(&d_sig, N * 4)); CK(hipMalloc(&d_meas, N * 16));
  CK(hipMalloc(&d_state, S * HW * 16)); CK(hipMalloc(&d_state0, S * HW * 16)); CK(hipMalloc(&d_rec, N * 16)); CK(hipMalloc(&d_ref, N * 16));
  CK(hipMemcpy(d_flow, h_flow.data(), N * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_sig, h_sig.data(), N * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_meas, h_meas.data(), N * 16, hipMemcpyHostToDevice)); CK(hipMemcpy(d_state0, h_state.data(), S * HW * 16, hipMemcpyHostToDevice));
  KalmanArgs a{};
  a.flow = (const f32x2*)d_flow; a.sigma_t = d_sig; a.meas = (const f32x4*)d_meas; a.state = (f32x4*)d_state; a.rec = (f32x4*)d_rec;
  a.d.S = S; a.d.T = T; a.d.H = H; a.d.W = W; a.d.t0 = 1; a.d.reset_period = 500; a.d.min_uncertainty = 1e-5f; a.d.nis_gate = 0.f;
  a.d.has_transform = 1;
  const float M[12] = {1, 0, 0, 0.1f, 0, 1, 0, 0.2f, 0, 0, 1, 0.3f};
  for (int i = 0; i < 12; ++i) a.d.transform[i] = M[i];
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double bytes_hbm = (double)N * 44.0 + 2.0 * S * HW * 16.0;

  std::vector<Variant> vs;
  if (HW * 32 <= 160 * 1024) {
    vs.push_back({"r4: 768x7 burst prefetch (rounds 1-4), IEEE sqrt / div", launch_scan_r4<768, 7, true, true>});
    vs.push_back({"768x7  D=7 nt (rolling)", launch_scan_dbg<768, 7, true, 7, true, false>});
    vs.push_back({"768x7  D=1 plain (~ rounds 1-4 without the burst prefetch)", launch_scan_dbg<768, 7, true, 1, false, false>});
    vs.push_back({"768x7  D=7 plain", launch_scan_dbg<768, 7, true, 7, false, false>});
    vs.push_back({"1024x5 D=5 nt, descriptors, IEEE", launch_scan_dbg<1024, 5, true, 5, true, false, false, false>});
    vs.push_back({"1024x5 D=5 plain", launch_scan_dbg<1024, 5, true, 5, false, false>});
    vs.push_back({"1024x5 D=5 nt, POINTER addressing, IEEE sqrt / div", launch_scan_dbg<1024, 5, true, 5, true, false, true, false>});
    vs.push_back({"1024x5 D=5 nt, POINTER addressing, lean (default)", launch_scan_dbg<1024, 5, true, 5, true, false, true, true>});
    vs.push_back({"1024x5 D=5 nt, descriptors, lean", launch_scan_dbg<1024, 5, true, 5, true, false, false, true>});
    vs.push_back({"768x7  D=7 nt, pointer addressing", launch_scan_dbg<768, 7, true, 7, true, false, true>});
    vs.push_back({"1024x5 D=1 nt", launch_scan_dbg<1024, 5, true, 1, true, false>});
    vs.push_back({"512x10 D=10 nt", launch_scan_dbg<512, 10, true, 10, true, false>});
    vs.push_back({"512x10 D=5 nt", launch_scan_dbg<512, 10, true, 5, true, false>});
    vs.push_back({"768x7  D=1 nt debug-outputs build (null pointers)", launch_scan_dbg<768, 7, true, 1, true, true>});
  } else {
    vs.push_back({"r4: 512x16 single-buffer, loads at frame start (rounds 1-4)", launch_scan_r4<512, 16, false, false>});
    vs.push_back({"512x16 D=1 plain single-buffer", launch_scan_dbg<512, 16, false, 1, false, false>});
    vs.push_back({"512x16 D=2 nt", launch_scan_dbg<512, 16, false, 2, true, false>});
    vs.push_back({"512x16 D=4 nt", launch_scan_dbg<512, 16, false, 4, true, false>});
    vs.push_back({"768x11 D=1 nt", launch_scan_dbg<768, 11, false, 1, true, false>});
    vs.push_back({"1024x8 D=2 nt", launch_scan_dbg<1024, 8, false, 2, true, false>});
    vs.push_back({"1024x8 D=4 nt", launch_scan_dbg<1024, 8, false, 4, true, false>});
  }
  printf("# kalman scan: S=%d T=%d %dx%d, %.3f GB crossing HBM per launch (44 B/px + the state once)\n", S, T, H, W, bytes_hbm / 1e9);
  bool first = true;
  for (auto& v : vs) {
    CK(hipMemcpy(d_state, d_state0, S * HW * 16, hipMemcpyDeviceToDevice));
    CK(hipMemset(d_rec, 0xFF, N * 16));
    if (v.launch(a, st) != 0) { printf("%-60s launch failed: %s\n", v.name.c_str(), kfn_last_error()); continue; }
    CK(hipStreamSynchronize(st));
    int same = -1;
    if (first) { CK(hipMemcpy(d_ref, d_rec, N * 16, hipMemcpyDeviceToDevice)); first = false; }
    else {
      std::vector<float> x(1 << 20), y(1 << 20);      // compare 4 MiB at the start, middle and end
      same = 1;
      for (size_t o : {(size_t)0, (N * 4 / 2) & ~(size_t)3, N * 4 - (1 << 20)}) {
        CK(hipMemcpy(x.data(), d_rec + o, 4 << 20, hipMemcpyDeviceToHost)); CK(hipMemcpy(y.data(), d_ref + o, 4 << 20, hipMemcpyDeviceToHost));
        if (memcmp(x.data(), y.data(), 4 << 20) != 0) same = 0;
      }
    }
    float best = 1e9f, sum = 0;
    const int reps = 7;
    for (int r = 0; r < reps; ++r) {
      CK(hipEventRecord(e0, st)); v.launch(a, st); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best; sum += ms;
    }
    printf("%-60s avg %.4f ms  best %.4f ms  %.0f GB/s  %.3f of 8 TB/s  %s\n", v.name.c_str(), sum / reps, best, bytes_hbm / (sum / reps * 1e-3) / 1e9,
           bytes_hbm / (sum / reps * 1e-3) / 8e12, same < 0 ? "(reference)" : (same ? "bit-identical" : "DIFFERS"));
  }

  // ---- BuildKFCoord alone: 48 B/px ----
  const long P = (long)N;
  std::vector<FuseVariant> fv = {
      {"fuse 256x1 plain, 4096 blocks (rounds 1-4)", launch_fuse<256, 1, false>},
      {"fuse 256x1 nt", launch_fuse<256, 1, true>},
      {"fuse 256x4 plain", launch_fuse<256, 4, false>},
      {"fuse 256x4 nt", launch_fuse<256, 4, true>},
      {"fuse 512x4 nt (default)", launch_fuse<512, 4, true>},
      {"fuse 512x2 nt", launch_fuse<512, 2, true>},
      {"fuse 1024x2 nt", launch_fuse<1024, 2, true>},
      {"fuse 256x8 nt", launch_fuse<256, 8, true>},
      {"fuse 256x4 nt, one trip per thread", launch_fuse_full<256, 4, true>},
      {"fuse 512x4 nt, one trip per thread", launch_fuse_full<512, 4, true>},
  };
  printf("# kalman fuse: P=%ld px, %.3f GB per launch (48 B/px)\n", P, P * 48.0 / 1e9);
  first = true;
  for (auto& v : fv) {
    CK(hipMemset(d_rec, 0xFF, N * 16));
    v.launch((const f32x4*)d_ref, (const f32x4*)d_meas, (f32x4*)d_rec, P, st);
    CK(hipStreamSynchronize(st));
    static std::vector<float> keep;
    std::vector<float> x(1 << 20);
    CK(hipMemcpy(x.data(), d_rec + N * 4 - (1 << 20), 4 << 20, hipMemcpyDeviceToHost));
    int same = -1;
    if (first) { keep = x; first = false; } else same = memcmp(x.data(), keep.data(), 4 << 20) == 0;
    float sum = 0; const int reps = 7;
    for (int r = 0; r < reps; ++r) {
      CK(hipEventRecord(e0, st)); v.launch((const f32x4*)d_ref, (const f32x4*)d_meas, (f32x4*)d_rec, P, st); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); sum += ms;
    }
    printf("%-60s avg %.4f ms  %.0f GB/s  %.3f of 8 TB/s  %s\n", v.name.c_str(), sum / reps, P * 48.0 / (sum / reps * 1e-3) / 1e9,
           P * 48.0 / (sum / reps * 1e-3) / 8e12, same < 0 ? "(reference)" : (same ? "bit-identical" : "DIFFERS"));
  }
  return 0;
}
