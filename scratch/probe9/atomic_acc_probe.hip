// Probe (review item 6): how long does a [32 x 4096] accumulate from 8 XCDs take with memory-side atomics?
// 256 workgroups x 1024 threads; XCD g's 32 workgroups together add one value to EVERY element of the [32 x 4096] accumulator, so every element
// receives 8 adds (one per XCD).  Variants: 8-byte fixed-point adds (the decode block's accumulator), 4-byte float adds, plain stores of the same bytes
// into 8 private slices (what the batched path does today: fp32 slices summed by the next launch), and an empty launch.
// hipcc --offload-arch=gfx950 -O3 atomic_acc_probe.hip -o atomic_acc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int M = 32, D = 4096, N = M * D;

template <int MODE>
__global__ __launch_bounds__(1024) void acc_kernel(unsigned long long* acc64, float* acc32, float* slices) {
  const int g = static_cast<int>(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u);
  const int r = static_cast<int>(blockIdx.x) >> 3;                      // member 0..31 of the XCD
  // element e = (r * 1024 + tid) * 4 + i: consecutive lanes -> consecutive 32-byte (u64) / 16-byte (f32) pieces
  const int e0 = (r * 1024 + threadIdx.x) * 4;
  if constexpr (MODE == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) __hip_atomic_fetch_add(acc64 + e0 + i, static_cast<unsigned long long>(g + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else if constexpr (MODE == 1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) __hip_atomic_fetch_add(acc32 + e0 + i, static_cast<float>(g + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else if constexpr (MODE == 2) {
    *reinterpret_cast<float4*>(slices + static_cast<size_t>(g) * N + e0) = make_float4(g + 1.f, g + 1.f, g + 1.f, g + 1.f);
  }
}

int main() {
  unsigned long long* a64; float *a32, *sl;
  CK(hipMalloc(&a64, N * 8)); CK(hipMalloc(&a32, N * 4)); CK(hipMalloc(&sl, static_cast<size_t>(8) * N * 4));
  CK(hipMemset(a64, 0, N * 8)); CK(hipMemset(a32, 0, N * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* names[4] = {"8-byte fixed-point atomic adds (8 x 1 MiB)", "4-byte float atomic adds (8 x 512 KiB)", "plain stores into 8 fp32 slices (8 x 512 KiB)", "empty launch"};
  float base = 0.f;
  for (int mode : {3, 0, 1, 2, 3}) {
    std::vector<float> ts;
    for (int rep = 0; rep < 12; ++rep) {
      CK(hipEventRecord(e0));
      for (int k = 0; k < 50; ++k) {
        if (mode == 0) acc_kernel<0><<<256, 1024>>>(a64, a32, sl);
        else if (mode == 1) acc_kernel<1><<<256, 1024>>>(a64, a32, sl);
        else if (mode == 2) acc_kernel<2><<<256, 1024>>>(a64, a32, sl);
        else acc_kernel<3><<<256, 1024>>>(a64, a32, sl);
      }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      ts.push_back(ms * 1e3f / 50);
    }
    std::sort(ts.begin(), ts.end());
    const float t = ts[ts.size() / 2];
    if (mode == 3) base = t;
    printf("%-48s : %6.2f us per launch back to back  (%+.2f us over the empty launch)\n", names[mode], t, t - base);
  }
  // check: every element received 8 adds per launch
  std::vector<unsigned long long> h(N); CK(hipMemcpy(h.data(), a64, N * 8, hipMemcpyDeviceToHost));
  unsigned long long want = 36ull * 12 * 50; bool ok = true; for (int i = 0; i < N; ++i) ok &= h[i] == want;
  printf("fixed-point sums %s (every element = %llu)\n", ok ? "exact" : "WRONG", want);
  return 0;
}
