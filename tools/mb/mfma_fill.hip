// Microbenchmark: how many filler instructions hide in the shadow of one v_mfma_f32_32x32x2_f32 when a
// SIMD holds ONE wave?  Prints cycles per MFMA for F independent VALU / SALU fillers between MFMAs.
//   hipcc --offload-arch=gfx950 -O3 tools/mb/mfma_fill.hip -o tools/mb/mfma_fill && tools/mb/mfma_fill
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int F, int KIND>
__global__ __launch_bounds__(64, 1) void k(float* out, int iters, long long* cyc) {
  extern __shared__ float lds[];
  f32x16 acc[4];
  for (int g = 0; g < 4; ++g) for (int e = 0; e < 16; ++e) acc[g][e] = 0.f;
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  float x0 = a, x1 = b, x2 = a + b, x3 = a - b;
  int s0 = iters;
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  f4 d4 = {a, b, a, b};
  f2 p0 = {a, b}, p1 = {b, a}, p2 = {1.f, 2.f};
  int ldsaddr = threadIdx.x * 16;
  lds[threadIdx.x] = a;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j & 3], 0, 0, 0);
#pragma unroll
      for (int f = 0; f < (KIND == 5 ? (j % 8 == 7 ? 8 * F : 0) : F); ++f) {
        if (KIND == 0 || KIND == 5) {
          if ((f & 3) == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x0) : "v"(b));
          if ((f & 3) == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x1) : "v"(b));
          if ((f & 3) == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x2) : "v"(b));
          if ((f & 3) == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x3) : "v"(b));
        } else if (KIND == 1) {
          asm volatile("s_add_i32 %0, %0, 1" : "+s"(s0));
        } else if (KIND == 2) {
          asm volatile("ds_read_b128 %0, %1" : "=v"(d4) : "v"(ldsaddr));
        } else if (KIND == 3) {
          asm volatile("ds_write_b128 %0, %1" : : "v"(ldsaddr), "v"(d4) : "memory");
        } else if (KIND == 4) {
          if ((f & 1) == 0) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p0) : "v"(p2));
          else asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p1) : "v"(p2));
        } else if (KIND == 6) {
          d4 = __builtin_nontemporal_load(reinterpret_cast<const f4*>(out) + ((it * 16 + j) * F + f) % 1024 * 64 + threadIdx.x);
        }
      }
      if (KIND == 2 || KIND == 6) { if (j == 15) asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)" ::: "memory"); }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = x0 + x1 + x2 + x3 + s0 + d4.x + d4.y + d4.z + d4.w + p0.x + p0.y + p1.x + p1.y;
  for (int g = 0; g < 4; ++g) for (int e = 0; e < 16; ++e) s += acc[g][e];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int F, int KIND>
void run(const char* kind, float* out, long long* cyc) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<F, KIND>), dim3(1024), dim3(64), 36 * 1024, 0, out, iters, cyc);   // warm
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<F, KIND>), dim3(1024), dim3(64), 36 * 1024, 0, out, iters, cyc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double n = 16.0 * iters;
  printf("%s fillers/MFMA = %2d: %.1f ns/MFMA wall (x2.4 = %.1f cyc), s_memtime/readcyclecounter %.1f ticks/MFMA, %.1f TF chip\n", kind, F,
         ms * 1e6 / n, ms * 1e6 / n * 2.4, (double)c / n, 1024.0 * n * 4096 * 64 / 64 / (ms * 1e-3) / 1e12);
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 1024 * 64 * 16 + 1024 * 64 * 4); hipMalloc(&cyc, 8);
  run<0, 0>("VALU", out, cyc); run<1, 0>("VALU", out, cyc); run<2, 0>("VALU", out, cyc); run<4, 0>("VALU", out, cyc);
  run<6, 0>("VALU", out, cyc); run<8, 0>("VALU", out, cyc); run<12, 0>("VALU", out, cyc); run<16, 0>("VALU", out, cyc);
  run<8, 1>("SALU", out, cyc);
  run<1, 2>("ds_read_b128", out, cyc); run<2, 2>("ds_read_b128", out, cyc); run<4, 2>("ds_read_b128", out, cyc);
  run<1, 3>("ds_write_b128", out, cyc); run<2, 3>("ds_write_b128", out, cyc);
  run<1, 4>("v_pk_add_f32", out, cyc); run<2, 4>("v_pk_add_f32", out, cyc); run<4, 4>("v_pk_add_f32", out, cyc); run<8, 4>("v_pk_add_f32", out, cyc);
  run<1, 5>("VALU burst 8 per 8 MFMA (avg fillers)", out, cyc); run<2, 5>("VALU burst 16 per 8 MFMA (avg)", out, cyc); run<4, 5>("VALU burst 32 per 8 MFMA (avg)", out, cyc);
  run<1, 6>("global_load_dwordx4", out, cyc); run<2, 6>("global_load_dwordx4", out, cyc);
  return 0;
}
