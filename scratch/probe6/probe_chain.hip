// Round-5 probe: can ONE launch carry a chain of dependent 256-workgroup stages?
//   hipcc --offload-arch=gfx950 -O3 -o bin/probe_chain probe_chain.hip
// Grid = STAGES x 256 workgroups of 1024 threads, 100 KB of dynamic LDS (one workgroup per CU).  Stage s = blockIdx / 256 waits until
// all 256 workgroups of stage s - 1 have signalled (one device-scope counter per stage), "works" for W us, signals.
// Questions: (1) are workgroups dispatched in blockIdx order, each stage filling the CUs the previous stage frees (no deadlock)?
// (2) does blockIdx % 8 stay the XCD label in every stage?  (3) what does the link cost: last signal of stage s -> first workgroup of
// stage s + 1 released, against the gap between two dependent launches on one stream and between two nodes of a replayed hipGraph?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef unsigned long long u64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef __attribute__((address_space(1))) unsigned gu32;
#define RLX __ATOMIC_RELAXED
#define AGENT __HIP_MEMORY_SCOPE_AGENT
__device__ __forceinline__ u64 now() { return __builtin_amdgcn_s_memrealtime(); }     // 100 MHz
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xF; }
__device__ __forceinline__ unsigned hw_id() { return __builtin_amdgcn_s_getreg((31 << 11) | 4); }
constexpr int NWG = 256, NT = 1024;
constexpr size_t BIG_LDS = 100 * 1024;

struct Rec { u64 t_start, t_ready, t_end; unsigned xcc, hw; };

// mode 0: chain inside one launch (stage from blockIdx); mode 1: one stage per launch (stage passed in, no wait)
__global__ __launch_bounds__(NT) void chain(unsigned* counters, Rec* rec, int work_ticks, int stage_arg, int mode, int epoch) {
  extern __shared__ unsigned char smem[];
  const int stage = mode == 0 ? static_cast<int>(blockIdx.x) / NWG : stage_arg;
  const int wg = static_cast<int>(blockIdx.x) % NWG;
  const u64 t0 = now();
  u64 t1 = t0;
  if (threadIdx.x == 0) smem[0] = 1;
  if (mode == 0 && stage > 0) {
    if (threadIdx.x == 0) {
      gu32* c = reinterpret_cast<gu32*>(reinterpret_cast<uintptr_t>(counters)) + (stage - 1) * 16;
      unsigned spins = 0;
      while (__hip_atomic_load(c, RLX, AGENT) != static_cast<unsigned>(NWG * epoch)) {
        if (++spins > (1u << 24)) break;
        __builtin_amdgcn_s_sleep(1);
      }
      t1 = now();
    }
    __syncthreads();
  }
  // "work": W + jitter ticks
  const u64 until = now() + work_ticks + (wg % 7) * 20;
  while (now() < until) __builtin_amdgcn_s_sleep(2);
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    atomicAdd(counters + stage * 16, 1u);
    Rec r;
    r.t_start = t0; r.t_ready = t1; r.t_end = now(); r.xcc = xcc_id(); r.hw = hw_id();
    rec[stage * NWG + wg] = r;
  }
}

static void report(const char* name, const std::vector<Rec>& r, int stages) {
  printf("== %s\n", name);
  u64 base = ~0ull;
  for (auto& x : r) base = std::min(base, x.t_start);
  bool inorder = true, bij = true;
  for (int s = 0; s < stages; ++s) {
    u64 smin = ~0ull, smax = 0, rmin = ~0ull, rmax = 0, emin = ~0ull, emax = 0;
    int cnt[8][8] = {};
    for (int w = 0; w < NWG; ++w) {
      const Rec& x = r[s * NWG + w];
      smin = std::min(smin, x.t_start); smax = std::max(smax, x.t_start);
      rmin = std::min(rmin, x.t_ready); rmax = std::max(rmax, x.t_ready);
      emin = std::min(emin, x.t_end); emax = std::max(emax, x.t_end);
      cnt[w % 8][x.xcc & 7]++;
    }
    for (int a = 0; a < 8; ++a) { int nz = 0; for (int b = 0; b < 8; ++b) nz += cnt[a][b] != 0; if (nz != 1) bij = false; }
    double prev_end = 0;
    if (s > 0) { u64 pe = 0; for (int w = 0; w < NWG; ++w) pe = std::max(pe, r[(s - 1) * NWG + w].t_end); prev_end = (pe - base) / 100.0;
                 u64 ps = 0; for (int w = 0; w < NWG; ++w) ps = std::max(ps, r[(s - 1) * NWG + w].t_start); if (smin < ps) {} }
    printf("  stage %2d: start %7.2f..%7.2f  released %7.2f..%7.2f  end %7.2f..%7.2f us", s, (smin - base) / 100.0, (smax - base) / 100.0,
           (rmin - base) / 100.0, (rmax - base) / 100.0, (emin - base) / 100.0, (emax - base) / 100.0);
    if (s > 0) printf("   link: prev last end -> first released %5.2f, -> last released %5.2f", (rmin - base) / 100.0 - prev_end, (rmax - base) / 100.0 - prev_end);
    printf("\n");
  }
  (void)inorder;
  printf("  blockIdx %% 8 <-> XCC_ID bijection in every stage: %s\n", bij ? "yes" : "NO");
}

int main() {
  const int STAGES = 8;
  unsigned* counters; Rec* rec;
  CK(hipMalloc(&counters, 64 * 16 * 4)); CK(hipMalloc(&rec, sizeof(Rec) * STAGES * NWG));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(chain), hipFuncAttributeMaxDynamicSharedMemorySize, BIG_LDS));
  std::vector<Rec> h(STAGES * NWG);
  hipStream_t st; CK(hipStreamCreate(&st));
  for (int work : {800, 300}) {
    printf("#### work per stage %.1f us (+ up to 1.2 us of jitter)\n", work / 100.0);
    // --- A: one launch, STAGES x 256 workgroups
    CK(hipMemsetAsync(counters, 0, 64 * 16 * 4, st));
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemsetAsync(counters, 0, 64 * 16 * 4, st));
      hipLaunchKernelGGL(chain, dim3(STAGES * NWG), dim3(NT), BIG_LDS, st, counters, rec, work, 0, 0, 1);
      CK(hipStreamSynchronize(st));
    }
    CK(hipMemcpy(h.data(), rec, sizeof(Rec) * STAGES * NWG, hipMemcpyDeviceToHost));
    report("A: one launch, stages chained through device-scope counters", h, STAGES);
    // --- B: STAGES dependent launches on one stream
    for (int rep = 0; rep < 3; ++rep) {
      for (int s = 0; s < STAGES; ++s) hipLaunchKernelGGL(chain, dim3(NWG), dim3(NT), BIG_LDS, st, counters, rec, work, s, 1, 1);
      CK(hipStreamSynchronize(st));
    }
    CK(hipMemcpy(h.data(), rec, sizeof(Rec) * STAGES * NWG, hipMemcpyDeviceToHost));
    report("B: one launch per stage, same stream (eager)", h, STAGES);
    // --- C: the same as a replayed graph
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int s = 0; s < STAGES; ++s) hipLaunchKernelGGL(chain, dim3(NWG), dim3(NT), BIG_LDS, st, counters, rec, work, s, 1, 1);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; ++rep) { CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st)); }
    CK(hipMemcpy(h.data(), rec, sizeof(Rec) * STAGES * NWG, hipMemcpyDeviceToHost));
    report("C: one launch per stage, hipGraph replay", h, STAGES);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}
