// Cycle-stamp profile of wino4_kernel (phases of one workgroup + the 16-slot timeline of one super-step).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mb/wino4_prof.hip -o tools/mb/wino4_prof
//   tools/mb/wino4_prof N H W Cin Cout        (all-zero operands: the clock holds its maximum)
#define KFN_WINO4_PROF 1
#include "../../kfnet_amd/csrc/kfn_wino4.hip"
#include <vector>
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
  const int N = argc > 1 ? atoi(argv[1]) : 16, H = argc > 2 ? atoi(argv[2]) : 60, W = argc > 3 ? atoi(argv[3]) : 80;
  const int Cin = argc > 4 ? atoi(argv[4]) : 1024, Cout = argc > 5 ? atoi(argv[5]) : 1024;
  kfn_conv_desc d = KFN_CONV_DESC_INIT;
  d.N = N; d.H = H; d.W = W; d.Cin = Cin; d.ldx = Cin; d.Cout = Cout; d.cout_pad = (Cout + 31) / 32 * 32; d.ldy = Cout;
  d.kh = d.kw = 3; d.stride = 1; d.relu = 1;
  const size_t xb = (size_t)N * H * W * Cin * 4, yb = (size_t)N * H * W * Cout * 4, ub = (size_t)36 * d.cout_pad * Cin * 4;
  float *x, *y, *u, *b;
  hipMalloc(&x, xb); hipMalloc(&y, yb); hipMalloc(&u, ub); hipMalloc(&b, d.cout_pad * 4);
  hipMemset(x, 0, xb); hipMemset(u, 0, ub); hipMemset(b, 0, d.cout_pad * 4);
  const int Th = (H + 3) / 4, Tw = (W + 3) / 4;
  const long nblk = (long)((Tw + 3) / 4) * ((N * Th + 7) / 8) * ((d.cout_pad + 63) / 64);
  const size_t pw = (size_t)nblk * 4 * (8 + 11);
  hipMalloc(&g_wino4_prof, pw * 8);
  hipMemset(g_wino4_prof, 0, pw * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) {
    hipEventRecord(e0, 0);
    int rc = kfn_conv2d_winograd_f43(&d, x, u, b, y, 0);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("launch rc=%d  %.3f ms  (%ld workgroups, %d super-steps each)\n", rc, ms, nblk, Cin / 16);
  }
  std::vector<unsigned long long> h(pw);
  hipMemcpy(h.data(), g_wino4_prof, pw * 8, hipMemcpyDeviceToHost);
  const long nw = nblk * 4;
  const char* nm[7] = {"set-up (geometry, offsets, accumulator init)", "prologue (gather, B ring, transform, stores, barrier)",
                       "main loop", "epilogue pass 1 (partner's rows -> LDS) + barrier", "epilogue pass 2 (own rows: add, write back) + barrier",
                       "image -> global (LDS reads + 32 stores per lane, issue)", "store drain"};
  double ph[7] = {0}, tot = 0;
  for (long i = 0; i < nw; ++i)
    for (int k = 0; k < 7; ++k) ph[k] += (double)(h[i * 8 + k + 1] - h[i * 8 + k]);
  for (int k = 0; k < 7; ++k) tot += ph[k];
  for (int k = 0; k < 7; ++k) printf("%-64s %10.0f cycles  %5.1f %%\n", nm[k], ph[k] / nw, 100 * ph[k] / tot);
  const int ns = Cin / 16;
  printf("per workgroup %.0f cycles; main loop per super-step %.0f (144 MFMAs = 9216 at one per 64 cycles); MFMA-only total %d\n",
         tot / nw, ph[2] / nw / ns, ns * 9216);
  const unsigned long long* tl = h.data() + (size_t)nw * 8;
  double seg[10] = {0};
  for (long i = 0; i < nw; ++i)
    for (int k = 0; k < 10; ++k) seg[k] += (double)(tl[i * 11 + k + 1] - tl[i * 11 + k]);
  printf("super-step %d, cycles per 16-slot segment (1024 = MFMA bound; gathers in segments 0-4, transform burst in 6, V stores 6-8):\n ", KFN_W4_TL_KS);
  for (int k = 0; k < 9; ++k) printf(" %.0f", seg[k] / nw);
  printf("   | wait for LDS + barrier %.0f\n", seg[9] / nw);
  return 0;
}
