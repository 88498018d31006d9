// Probe of `buffer_load_dwordx4 ... lds` (LDS-DMA) on gfx950: where do the lanes' 16 bytes land, and what does a lane
// whose offset fails the buffer range check write?   hipcc --offload-arch=gfx950 -O2 tools/mb/dma_probe.hip -o tools/mb/dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__global__ void k(const float* g, float* out, int nbytes) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  for (int i = threadIdx.x; i < 1024; i += 64) smem[i] = -7.f;     // stale marker
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g), 0, nbytes, 0x00020000);
  const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem);
  // lane L reads global quad (63 - L) (reversed), lanes 8..15 are out of range
  unsigned voff = (63u - threadIdx.x) * 16u;
  if (threadIdx.x >= 8 && threadIdx.x < 16) voff = 0x80000000u;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(uintptr_t)base, 16, voff, 0u, 0, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(uintptr_t)(base + 2048u), 16, threadIdx.x * 16u, 1024u, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 64) out[i] = smem[i];
}
int main() {
  std::vector<float> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = (float)i;
  float *g, *o;
  hipMalloc(&g, 4096); hipMalloc(&o, 4096);
  hipMemcpy(g, h.data(), 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, g, o, 4096);
  std::vector<float> r(1024);
  hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
  printf("first DMA (reversed source, lanes 8..15 out of range): LDS quads 0..19 hold source quad:");
  for (int q = 0; q < 20; ++q) printf(" %g", r[q * 4] / 4.f);
  printf("\nsecond DMA (soffset 1024, LDS base +2048): LDS floats 512..519:");
  for (int i = 512; i < 520; ++i) printf(" %g", r[i]);
  printf("\nuntouched LDS float 300: %g\n", r[300]);
  return 0;
}
