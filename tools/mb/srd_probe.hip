// Does a raw buffer descriptor's range check count the SCALAR offset?  records = 256 bytes over a 4 KiB buffer of 1, 2, 3, ...:
// lane i loads a dword at voffset = 4 i with soffset = 0 / 128 / 512 and with the same distance in the VGPR instead.
//   hipcc --offload-arch=gfx950 -O2 -o tools/mb/srd_probe tools/mb/srd_probe.hip && tools/mb/srd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(const int* buf, int* out) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(buf), 0, 256, 0x00020000);
  const unsigned v = threadIdx.x * 4u;
  out[threadIdx.x + 0 * 64] = __builtin_amdgcn_raw_buffer_load_b32(rs, v, 0u, 0);
  out[threadIdx.x + 1 * 64] = __builtin_amdgcn_raw_buffer_load_b32(rs, v, 128u, 0);
  out[threadIdx.x + 2 * 64] = __builtin_amdgcn_raw_buffer_load_b32(rs, v, 512u, 0);
  out[threadIdx.x + 3 * 64] = __builtin_amdgcn_raw_buffer_load_b32(rs, v + 128u, 0u, 0);
  out[threadIdx.x + 4 * 64] = __builtin_amdgcn_raw_buffer_load_b32(rs, v + 512u, 0u, 0);
}
int main() {
  int h[1024], *d, *o, r[320];
  for (int i = 0; i < 1024; ++i) h[i] = i + 1;
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(r));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o);
  hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  const char* name[5] = {"soffset 0", "soffset 128", "soffset 512", "voffset + 128", "voffset + 512"};
  for (int k = 0; k < 5; ++k) {
    printf("%-14s lanes 0, 31, 32, 63: %d %d %d %d\n", name[k], r[k * 64], r[k * 64 + 31], r[k * 64 + 32], r[k * 64 + 63]);
  }
  return 0;
}
