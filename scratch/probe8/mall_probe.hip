// Probe: does a SPARSE touch (one 4-byte load per S bytes) of a weight range bring it into the Infinity Cache, so that a later launch streams it faster?
// Standalone: hipcc --offload-arch=gfx950 -O3 mall_probe.hip -o mall_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <bool NT, class T> __device__ inline T ld(const T* p) { if constexpr (NT) return __builtin_nontemporal_load(p); else return *p; }

template <bool NT>
__global__ __launch_bounds__(1024) void stream_kernel(const u32x4* __restrict__ p, size_t n16, unsigned* out) {
  unsigned acc = 0;
  const size_t stride = static_cast<size_t>(gridDim.x) * 1024;
  size_t i = static_cast<size_t>(blockIdx.x) * 1024 + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    const u32x4 a = ld<NT>(p + i), b = ld<NT>(p + i + stride), c = ld<NT>(p + i + 2 * stride), d = ld<NT>(p + i + 3 * stride);
    acc += a.x ^ b.y ^ c.z ^ d.w;
  }
  for (; i < n16; i += stride) acc += ld<NT>(p + i).x;
  if (acc == 0x12345678u) out[0] = acc;
}
// one 4-byte load per `step` bytes
__global__ __launch_bounds__(1024) void touch_kernel(const unsigned* __restrict__ p, size_t bytes, int step, unsigned* out) {
  unsigned acc = 0;
  const size_t n = bytes / step, stride = static_cast<size_t>(gridDim.x) * 1024;
  for (size_t i = static_cast<size_t>(blockIdx.x) * 1024 + threadIdx.x; i < n; i += stride) acc += p[i * (step / 4)];
  if (acc == 0x12345678u) out[0] = acc;
}

int main() {
  const size_t WB = 128ull << 20, JB = 1024ull << 20;
  void *w, *junk; unsigned* out;
  CK(hipMalloc(&w, WB)); CK(hipMalloc(&junk, JB)); CK(hipMalloc(&out, 64));
  CK(hipMemset(w, 1, WB)); CK(hipMemset(junk, 2, JB));
  hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
  auto med = [](std::vector<float>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  const int steps[] = {0, -1, 4096, 256, 128, 64, 32};   // 0: cold, -1: warm (stream twice)
  for (int fin = 0; fin < 2; ++fin)
  for (int step : steps) {
    std::vector<float> ts, tt;
    for (int rep = 0; rep < 15; ++rep) {
      stream_kernel<true><<<256, 1024>>>(static_cast<const u32x4*>(junk), JB / 16, out);          // flush the caches
      CK(hipEventRecord(e0));
      if (step > 0) touch_kernel<<<256, 1024>>>(static_cast<const unsigned*>(w), WB, step, out);
      if (step < 0) stream_kernel<false><<<256, 1024>>>(static_cast<const u32x4*>(w), WB / 16, out);
      CK(hipEventRecord(e1));
      if (fin) stream_kernel<true><<<256, 1024>>>(static_cast<const u32x4*>(w), WB / 16, out); else stream_kernel<false><<<256, 1024>>>(static_cast<const u32x4*>(w), WB / 16, out);
      CK(hipEventRecord(e2));
      CK(hipEventSynchronize(e2));
      float a, b; CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&b, e1, e2));
      tt.push_back(a * 1e3f); ts.push_back(b * 1e3f);
    }
    const float s = med(ts), t = med(tt);
    printf("final %s | prepare %-22s : prepare %7.2f us   then stream 128 MiB %7.2f us = %6.2f TB/s\n",
           fin ? "nt     " : "default",
           step == 0 ? "nothing (cold)" : step < 0 ? "full stream (warm)" : (std::string("touch 4 B per ") + std::to_string(step) + " B").c_str(), t, s, WB / s * 1e-6);
  }
  return 0;
}
