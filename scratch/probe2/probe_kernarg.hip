// Probe: does kernarg preloading (-mllvm -amdgpu-kernarg-preload-count=N) shorten a dependent launch chain?
// Kernel: every workgroup loads x[0..] through a pointer argument and writes y (the shape of a GEMV prologue start).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(const float* __restrict__ x, float* __restrict__ y, int n, float s) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] = x[i] * s + 1.0f;
}
int main() {
  const int n = 256 * 256;
  float *a, *b; hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMemset(a, 0, n * 4);
  hipStream_t st; hipStreamCreate(&st);
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  for (int i = 0; i < 500; ++i) { hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, st, a, b, n, 1.0f); hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, st, b, a, n, 1.0f); }
  hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipGraphLaunch(ge, st); hipStreamSynchronize(st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int r = 0; r < 3; ++r) {
    hipEventRecord(e0, st); hipGraphLaunch(ge, st); hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("dependent chain of 1000 tiny kernels: %.3f us per kernel\n", ms);
  }
  return 0;
}
