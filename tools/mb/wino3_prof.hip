// Cycle-stamp profile of wino3_kernel (phases of one workgroup, gaps between workgroups on a CU).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mb/wino3_prof.hip -o tools/mb/wino3_prof
//   tools/mb/wino3_prof N H W Cin Cout
#define KFN_WINO3_PROF 1
#include "../../kfnet_amd/csrc/kfn_wino3.hip"
#include <vector>
#include <map>
#include <algorithm>
#include <cstdlib>

namespace kfn {
char* err_buf() { static thread_local char b[512]; return b; }
int fail(int code, const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr);
  return code;
}
}  // namespace kfn

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 16, H = argc > 2 ? atoi(argv[2]) : 240, W = argc > 3 ? atoi(argv[3]) : 320;
  const int Cin = argc > 4 ? atoi(argv[4]) : 256, Cout = argc > 5 ? atoi(argv[5]) : 256;
  const int xpad = argc > 6 ? atoi(argv[6]) : 0, ypad = argc > 7 ? atoi(argv[7]) : 0;   // extra floats of pixel pitch
  kfn_conv_desc d;
  memset(&d, 0, sizeof d);
  d.struct_size = (int)sizeof d;
  d.N = N; d.H = H; d.W = W; d.Cin = Cin; d.ldx = Cin + xpad; d.Cout = Cout; d.cout_pad = (Cout + 31) / 32 * 32; d.ldy = Cout + ypad;
  d.kh = d.kw = 3; d.stride = 1; d.relu = 1;
  const size_t xb = (size_t)N * H * W * d.ldx * 4, yb = (size_t)N * H * W * d.ldy * 4, ub = (size_t)16 * d.cout_pad * Cin * 4;
  float *x, *y, *u, *b;
  hipMalloc(&x, xb); hipMalloc(&y, yb); hipMalloc(&u, ub); hipMalloc(&b, d.cout_pad * 4);
  hipMemset(x, 0, xb); hipMemset(u, 0, ub); hipMemset(b, 0, d.cout_pad * 4);
  const int Th = (H + 1) / 2, Tw = (W + 1) / 2;
  const long nblk = (long)((Tw + 7) / 8) * ((N * Th + 3) / 4) * ((d.cout_pad + 127) / 128);
  const int NWV = d.cout_pad == 64 ? 2 : 4;   // waves per workgroup: the two-wave form takes 33..64 output channels
  hipMalloc(&g_wino3_prof, nblk * 4 * 25 * 8);
  hipMemset(g_wino3_prof, 0, nblk * 4 * 25 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) {
    hipEventRecord(e0, 0);
    int rc = kfn::launch_wino3(&d, x, u, b, y, 0);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("launch rc=%d  %.3f ms  (%ld blocks)\n", rc, ms, nblk);
  }
  std::vector<unsigned long long> h(nblk * NWV * 8);
  hipMemcpy(h.data(), g_wino3_prof, nblk * NWV * 8 * 8, hipMemcpyDeviceToHost);
  // per-phase means over waves
  double ph[5] = {0, 0, 0, 0, 0};
  for (long i = 0; i < nblk * NWV; ++i)
    for (int k = 0; k < 5; ++k) ph[k] += (double)(h[i * 8 + k + 1] - h[i * 8 + k]);
  const char* nm[5] = {"setup (geometry, acc init)", "prologue (gather, B ring, transform, barrier)", "main loop",
                       "epilogue issue", "store drain"};
  double tot = 0;
  for (int k = 0; k < 5; ++k) tot += ph[k];
  for (int k = 0; k < 5; ++k) printf("%-48s %10.0f cycles  %5.1f %%\n", nm[k], ph[k] / (nblk * NWV), 100 * ph[k] / tot);
#ifdef KFN_W3_EPI
  {
    double a = 0, b = 0;
    for (long i = 0; i < nblk * NWV; ++i) { a += (double)(h[i * 8 + 6] - h[i * 8 + 3]); b += (double)(h[i * 8 + 4] - h[i * 8 + 6]); }
    printf("epilogue: output transform + LDS writes %.0f cycles, LDS reads + stores %.0f cycles\n", a / (nblk * NWV), b / (nblk * NWV));
  }
#endif
  const int chunks = Cin / 8;
  {
    // distribution of the main-loop time over workgroups (wave 0), and its mean by block column
    std::vector<double> ml(nblk);
    const int bw = (Tw + 7) / 8;
    const long tiles_m = (long)bw * ((N * Th + 3) / 4);
    std::vector<double> by_cb(bw, 0.0); std::vector<long> n_cb(bw, 0);
    for (long bi = 0; bi < nblk; ++bi) {
      ml[bi] = (double)(h[(bi * NWV) * 8 + 3] - h[(bi * NWV) * 8 + 2]) / chunks;
      const int nwg = (int)nblk; const int xcd = bi & 7; const int q = nwg >> 3, r = nwg & 7;
      const long base = (xcd < r) ? (long)xcd * (q + 1) : (long)r * (q + 1) + (long)(xcd - r) * q;
      const long tile = base + (bi >> 3);
      const int cb = (int)((tile % tiles_m) % bw);
      by_cb[cb] += ml[bi]; n_cb[cb]++;
    }
    std::vector<double> srt(ml); std::sort(srt.begin(), srt.end());
    printf("main loop/chunk over workgroups: min %.0f p10 %.0f median %.0f p90 %.0f max %.0f\n", srt[0], srt[nblk / 10],
           srt[nblk / 2], srt[nblk * 9 / 10], srt[nblk - 1]);
    printf("mean by block column:");
    for (int c = 0; c < bw; ++c) printf(" %.0f", by_cb[c] / (n_cb[c] ? n_cb[c] : 1));
    printf("\n");
  }
  printf("main loop per chunk: %.0f cycles (64 MFMAs = 4096 at one per 64)\n", ph[2] / (nblk * NWV) / chunks);
  // gaps between consecutive workgroups on the same SIMD (wave 0 of each block keyed by xcc/se/cu/simd)
  std::map<unsigned long long, std::vector<std::pair<unsigned long long, unsigned long long>>> per;
  for (long bi = 0; bi < nblk; ++bi)
    for (int w = 0; w < NWV; ++w) {
      const unsigned long long* r = &h[(bi * NWV + w) * 8];
      const unsigned hw = (unsigned)r[6], xcc = (unsigned)r[7] & 0xf;
      const unsigned long long key = ((unsigned long long)xcc << 32) | (hw & 0xff30);   // SIMD, CU, SH, SE
      per[key].push_back({r[0], r[5]});
    }
  double gap = 0, busy = 0; long ng = 0;
  for (auto& kv : per) {
    auto& v = kv.second;
    std::sort(v.begin(), v.end());
    for (size_t i = 0; i < v.size(); ++i) busy += (double)(v[i].second - v[i].first);
    for (size_t i = 1; i < v.size(); ++i) { gap += (double)((long long)(v[i].first - v[i - 1].second)); ++ng; }
  }
  printf("distinct (xcc, hw_id) slots: %zu; mean gap between a wave's end and the next wave's start there: %.0f cycles\n",
         per.size(), ng ? gap / ng : 0.0);
#ifdef KFN_W3_TL
  {
    std::vector<unsigned long long> tl(nblk * NWV * 17);
    hipMemcpy(tl.data(), g_wino3_prof + nblk * NWV * 8, nblk * NWV * 17 * 8, hipMemcpyDeviceToHost);
    double seg[16] = {0};
    const int nseg = NWV * 4;   // a super-step is NWV chunks of 64 slots
    for (long i = 0; i < nblk * NWV; ++i)
      for (int k = 0; k < nseg; ++k) seg[k] += (double)(tl[i * 17 + (k + 1 < nseg ? k + 1 : 16)] - tl[i * 17 + k]);
    printf("last super-step, cycles per 16-slot segment (1024 = MFMA bound):");
    for (int k = 0; k < nseg; ++k) printf(" %.0f", seg[k] / (nblk * NWV));
    printf("\n");
  }
#endif
  printf("per block total %.0f cycles; MFMA-only would be %d\n", tot / (nblk * NWV), chunks * 4096);
  return 0;
}
