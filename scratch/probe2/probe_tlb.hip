// Probe: cost of touching R distinct regions at kernel start (address translation), stride S between regions.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <int R>
__global__ __launch_bounds__(256) void touch(const uint32_t* __restrict__ base, uint32_t* out, size_t stride_words, size_t blk_words) {
  const uint32_t* p = base + blockIdx.x * blk_words + threadIdx.x * 4;
  u32x4 v[R];
#pragma unroll
  for (int r = 0; r < R; ++r) v[r] = *reinterpret_cast<const u32x4*>(p + r * stride_words);
  u32x4 acc = v[0];
#pragma unroll
  for (int r = 1; r < R; ++r) acc ^= v[r];
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x1234567u) out[threadIdx.x] = acc.x;
}
int main() {
  const size_t bytes = size_t(3) << 30;
  uint32_t *buf, *out; CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 4096)); CK(hipMemset(buf, 1, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipStream_t st; CK(hipStreamCreate(&st));
  const size_t strides[] = {4096, 65536, 2u << 20, 32u << 20};
  for (size_t S : strides) {
    for (int R : {1, 2, 4, 8, 16}) {
      // 40 launches, each at a fresh base (64 MB apart) so nothing is cached/translated from the previous one
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
      for (int i = 0; i < 40; ++i) {
        const uint32_t* b = buf + (size_t(i) * (48u << 20) + (i & 1) * 8192) / 4;
        const size_t sw = S / 4, bw = 1024;   // each block reads 4 KB per region; 256 blocks = 1 MB per region
        switch (R) {
          case 1: hipLaunchKernelGGL(touch<1>, dim3(256), dim3(256), 0, st, b, out, sw, bw); break;
          case 2: hipLaunchKernelGGL(touch<2>, dim3(256), dim3(256), 0, st, b, out, sw, bw); break;
          case 4: hipLaunchKernelGGL(touch<4>, dim3(256), dim3(256), 0, st, b, out, sw, bw); break;
          case 8: hipLaunchKernelGGL(touch<8>, dim3(256), dim3(256), 0, st, b, out, sw, bw); break;
          default: hipLaunchKernelGGL(touch<16>, dim3(256), dim3(256), 0, st, b, out, sw, bw); break;
        }
      }
      CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
      CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("stride %8zu KB  regions %2d : %6.2f us per launch\n", S >> 10, R, ms * 1e3 / 40);
      hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
  }
  return 0;
}
