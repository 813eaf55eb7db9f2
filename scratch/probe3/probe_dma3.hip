// Where does `buffer_load_dwordx3 ... lds` put lane l's 12 bytes?  Dumps the first 1.5 KiB of LDS after one wave-load.
#include <hip/hip_runtime.h>
#include <cstdio>
#define LDSP(p) ((__attribute__((address_space(3))) void*)(p))
__global__ void k(const unsigned* g, unsigned* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  for (int i = threadIdx.x; i < 512; i += 64) ((unsigned*)smem)[i] = 0xDEAD0000u + i;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(g), 0, 0x7FFFFFFF, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDSP(smem), 12, threadIdx.x * 12, 0, 0, 0);
  __builtin_amdgcn_s_waitcnt(0x0f70);
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 64) out[i] = ((unsigned*)smem)[i];
}
int main() {
  unsigned *g, *o, h[512], src[512];
  for (int i = 0; i < 512; ++i) src[i] = i;
  hipMalloc(&g, 2048); hipMalloc(&o, 2048);
  hipMemcpy(g, src, 2048, hipMemcpyHostToDevice);
  k<<<1, 64, 4096>>>(g, o);
  hipMemcpy(h, o, 2048, hipMemcpyDeviceToHost);
  for (int i = 0; i < 272; ++i) printf("%08x%c", h[i], (i % 8 == 7) ? '\n' : ' ');
  return 0;
}
