// Which touch granularity makes a later nt stream of the same bytes hit the Infinity Cache?
// touch(S): one dword per S bytes (S = 16: dwordx4 per lane, full lines); stream: nt dwordx4 loads of everything.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void touch(const unsigned* p, size_t bytes, int S, unsigned* sink) {
  unsigned acc = 0;
  const size_t n = bytes / S, gt = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gt) {
    if (S == 16) { u32x4 v = *(const u32x4*)((const char*)p + i * 16); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    else acc ^= *(const unsigned*)((const char*)p + i * S);
  }
  if (acc == 0x12345678u) *sink = acc;
}
__global__ void stream(const unsigned* p, size_t bytes, unsigned* sink) {
  unsigned acc = 0;
  const size_t n = bytes / 16, gt = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gt) {
    u32x4 v = __builtin_nontemporal_load((const u32x4*)p + i);
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) *sink = acc;
}
int main() {
  const size_t B = 33u << 20; const int NB = 24;
  std::vector<unsigned*> bufs(NB); unsigned* sink;
  for (auto& b : bufs) { hipMalloc(&b, B); hipMemset(b, 1, B); }
  hipMalloc(&sink, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](int S, bool do_touch, bool do_stream) {
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
      hipDeviceSynchronize(); hipEventRecord(e0);
      for (int i = 0; i < NB; ++i) {
        if (do_touch) touch<<<1024, 256>>>(bufs[i], B, S, sink);
        if (do_stream) stream<<<2048, 256>>>(bufs[i], B, sink);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
    }
    return best * 1e3f / NB;
  };
  printf("stream cold: %.2f us per 33 MB (%.2f TB/s)\n", run(16, false, true), 33.0 * 1.048576 / run(16, false, true));
  for (int S : {16, 32, 64, 128, 256}) {
    float t = run(S, true, false), b = run(S, true, true);
    printf("touch stride %3d: touch %.2f us, touch+stream %.2f us -> stream after touch %.2f us\n", S, t, b, b - t);
  }
  return 0;
}
