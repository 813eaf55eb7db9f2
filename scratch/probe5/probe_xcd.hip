// Round-5 step 0: what does an XCD-local hand-off cost, and where do workgroups land?
//   hipcc --offload-arch=gfx950 -O3 -o bin/probe_xcd probe_xcd.hip
// Sections (all grids: 256 workgroups x 1024 threads, 100 KB of dynamic LDS = one workgroup per CU):
//   P1 census     : (XCC_ID, CU) of every workgroup over 64 replayed launches -- is blockIdx % 8 a stable XCD label, one workgroup per CU?
//   P2 barrier    : 32-workgroup barrier of the group blockIdx % 8, 64 rounds inside one launch, three forms, idle / beside a weight stream
//   P3 chain      : the two hand-offs of the fused attention launch as data-tagged 8-byte granules:
//                   28 producers x 14 granules -> 4 consumers (192 granules each) -> 64 granules each -> 32 consumers (256 granules each)
//   P4 atomics    : 8 groups adding 4096 partial sums into ONE [4096] accumulator: i64 / f32, 8-lane and 64-lane forms
//   P5 prologue   : every workgroup reads X KB that the previous launch wrote (atomics or plain stores) before it can start
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef __attribute__((address_space(1))) unsigned gu32;
typedef __attribute__((address_space(1))) u64 gu64;
#define RLX __ATOMIC_RELAXED
#define AGENT __HIP_MEMORY_SCOPE_AGENT
#define WGSCOPE __HIP_MEMORY_SCOPE_WORKGROUP

__device__ __forceinline__ u64 now() { return __builtin_amdgcn_s_memrealtime(); }     // 100 MHz
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xF; }
__device__ __forceinline__ unsigned hw_id() { return __builtin_amdgcn_s_getreg((31 << 11) | 4); }

constexpr int NWG = 256, NT = 1024, GROUPS = 8, GSZ = 32;
constexpr size_t BIG_LDS = 100 * 1024;

// ------------------------------------------------------------------------------------------------ P1
__global__ __launch_bounds__(NT) void census(unsigned* out, int launch) {
  extern __shared__ unsigned char smem[];
  if (threadIdx.x == 0) {
    smem[0] = 1;
    out[(launch * NWG + blockIdx.x) * 2] = xcc_id();
    out[(launch * NWG + blockIdx.x) * 2 + 1] = hw_id();
  }
}

// ------------------------------------------------------------------------------------------------ background stream
// waves 1.. of every workgroup pull `rows` 1-KB rows each (nt) while wave 0 does the protocol under test
__device__ __forceinline__ unsigned stream_rows(const uint32_t* w, int rows, size_t stride_rows, int gw) {
  const int lane = threadIdx.x & 63;
  unsigned acc = 0;
  for (int r = 0; r < rows; r += 4) {
    u32x4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      v[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(w + (static_cast<size_t>(gw) + static_cast<size_t>(r + i) * stride_rows) * 256 + lane * 4));
#pragma unroll
    for (int i = 0; i < 4; ++i) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
  }
  return acc;
}

// ------------------------------------------------------------------------------------------------ P2
// form 0: agent-scope atomic add on the group's counter + sc1 poll; form 1: workgroup-scope atomic (no sc bits) + sc1 poll;
// form 2: one flag word per member (sc1 store of the round), a 32-lane sc1 poll of the group's 128-byte line
template <int FORM, int LOAD>
__global__ __launch_bounds__(NT) void barrier_probe(unsigned* state, u64* times, unsigned* err, const uint32_t* w, int rounds, unsigned* sink) {
  extern __shared__ unsigned char smem[];
  const int g = blockIdx.x % GROUPS, r = blockIdx.x / GROUPS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  gu32* ctr = (gu32*)(state + g * 64);                 // one 256-byte block per group
  gu32* flags = (gu32*)(state + 1024 + g * 64);
  if (wave == 0) {
    const u64 t0 = now();
    for (int it = 1; it <= rounds; ++it) {
      if (FORM == 0) { if (lane == 0) __hip_atomic_fetch_add(ctr, 1u, RLX, AGENT); }
      else if (FORM == 1) { if (lane == 0) __hip_atomic_fetch_add(ctr, 1u, RLX, WGSCOPE); }
      else { if (lane == 0) __hip_atomic_store(flags + r, static_cast<unsigned>(it), RLX, AGENT); }
      unsigned spins = 0;
      for (;;) {
        bool ok;
        if (FORM == 2) { const unsigned v = __hip_atomic_load(flags + (lane & 31), RLX, AGENT); ok = v >= static_cast<unsigned>(it); }
        else { const unsigned v = __hip_atomic_load(ctr, RLX, AGENT); ok = v >= static_cast<unsigned>(it * GSZ); }
        if (__all(ok)) break;
        if (++spins > (1u << 22)) { if (lane == 0) err[0] = 1; break; }
      }
    }
    const u64 t1 = now();
    if (lane == 0) times[blockIdx.x] = t1 - t0;
  } else if (LOAD) {
    const unsigned a = stream_rows(w, LOAD, NWG * 15, blockIdx.x * 15 + wave - 1);
    if (a == 0x1234567u) sink[threadIdx.x] = a;
  }
  if (threadIdx.x == 0) smem[0] = 1;
}

// ------------------------------------------------------------------------------------------------ P3
// granule = {tag << 32 | value}; buffers double-buffered by round parity.  Roles inside group g (r = blockIdx / 8):
//   r < 4: attention head r: waits for 192 phase-1 granules (its q: 64, k: 64, v: 64), publishes 64;  r >= 4: phase-1 producer of
//   14 granules (wave w < 14, one granule each -- as the row-pair totals of the real kernel); everyone: waits for the 256 phase-2 granules.
// FLAGS = 1: payload as sc1 16-byte stores + drained counter add; consumers poll the counter, then sc1-load the payload.
template <int FLAGS, int LOAD>
__global__ __launch_bounds__(NT) void chain_probe(u64* gran, unsigned* state, u64* times, unsigned* err, const uint32_t* w, int rounds, unsigned* sink) {
  extern __shared__ unsigned char smem[];
  unsigned* lds = reinterpret_cast<unsigned*>(smem);
  const int g = FLAGS == 2 ? static_cast<int>(xcc_id() & 7) : blockIdx.x % GROUPS, r = blockIdx.x / GROUPS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // per group and parity: 448 phase-1 granules (q 4 x 64 | k 64 | v 64 -> producers write 28 x 14 = 392 of them: enough for the probe) + 256 phase-2
  gu64* base = (gu64*)(gran) + static_cast<size_t>(g) * 2 * 1024;
  gu32* c1 = (gu32*)(state + g * 64), *c2 = (gu32*)(state + g * 64 + 32);
  u64 t_h1 = 0, t_h2 = 0;
  const u64 t0 = now();
  unsigned bad = 0;
  for (int it = 1; it <= rounds; ++it) {
    gu64* p1 = base + (it & 1) * 1024, *p2 = p1 + 512;
    const unsigned tag = static_cast<unsigned>(it);
    if (r >= 4) {                                            // ---- phase-1 producer
      if (wave < 14 && lane == 0) {
        const int idx = (r - 4) * 14 + wave;                  // 0 .. 391
        if (FLAGS == 2) __hip_atomic_store(p1 + idx, (static_cast<u64>(tag) << 32) | (tag * 1000u + idx), RLX, WGSCOPE);   // sc0: stays in this XCD's L2
        else if (FLAGS) {
          __hip_atomic_store((gu32*)(p1) + idx, tag * 1000u + idx, RLX, AGENT);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __hip_atomic_fetch_add(c1, 1u, RLX, AGENT);
        } else __hip_atomic_store(p1 + idx, (static_cast<u64>(tag) << 32) | (tag * 1000u + idx), RLX, AGENT);
      }
    } else {                                                 // ---- attention head r: 192 granules by waves 0..2, then 64 by wave 0
      if (wave < 3) {
        const u64 ta = now();
        const int idx = wave == 0 ? r * 64 + lane : (wave == 1 ? 256 + lane : 320 + lane);
        const int idc = idx < 392 ? idx : 391;
        unsigned spins = 0, val = 0;
        for (;;) {
          bool ok;
          if (FLAGS == 1) { ok = __hip_atomic_load(c1, RLX, AGENT) >= static_cast<unsigned>(it * 392); }
          else { const u64 x = __hip_atomic_load(p1 + idc, RLX, AGENT); val = static_cast<unsigned>(x); ok = (x >> 32) == tag; }
          if (__all(ok)) break;
          if (++spins > (1u << 22)) { if (lane == 0) err[0] = 2; break; }
        }
        if (FLAGS == 1) val = __hip_atomic_load((gu32*)(p1) + idc, RLX, AGENT);
        if (val != tag * 1000u + idc) bad++;
        lds[wave * 64 + lane] = val;
        if (wave == 0) t_h1 += now() - ta;
      }
      __syncthreads();
      if (wave == 0) {
        const unsigned v = lds[lane] + lds[64 + lane] + lds[128 + lane];
        if (FLAGS == 2) __hip_atomic_store(p2 + r * 64 + lane, (static_cast<u64>(tag) << 32) | v, RLX, WGSCOPE);
        else if (FLAGS) {
          __hip_atomic_store((gu32*)(p2) + r * 64 + lane, v, RLX, AGENT);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (lane == 0) __hip_atomic_fetch_add(c2, 1u, RLX, AGENT);
        } else __hip_atomic_store(p2 + r * 64 + lane, (static_cast<u64>(tag) << 32) | v, RLX, AGENT);
      }
    }
    // ---- everyone: the 256 phase-2 granules (waves 0..3)
    if (wave < 4) {
      const u64 ta = now();
      unsigned spins = 0, val = 0;
      for (;;) {
        bool ok;
        if (FLAGS == 1) ok = __hip_atomic_load(c2, RLX, AGENT) >= static_cast<unsigned>(it * 4);
        else { const u64 x = __hip_atomic_load(p2 + wave * 64 + lane, RLX, AGENT); val = static_cast<unsigned>(x); ok = (x >> 32) == tag; }
        if (__all(ok)) break;
        if (++spins > (1u << 22)) { if (lane == 0) err[0] = 3; break; }
      }
      if (FLAGS == 1) val = __hip_atomic_load((gu32*)(p2) + wave * 64 + lane, RLX, AGENT);
      if (FLAGS != 1 && val != 3u * tag * 1000u + static_cast<unsigned>((wave * 64 + lane) + (256 + lane) + (320 + lane < 392 ? 320 + lane : 391))) bad++;
      lds[256 + wave * 64 + lane] = val;
      if (wave == 0) t_h2 += now() - ta;
    } else if (LOAD && it <= 8) {
      const unsigned a = stream_rows(w, LOAD, NWG * 12 * 8, ((it - 1) * NWG + blockIdx.x) * 12 + wave - 4);
      if (a == 0x1234567u) sink[threadIdx.x] = a;
    }
    __syncthreads();
  }
  const u64 t1 = now();
  if (threadIdx.x == 0) { times[blockIdx.x * 4] = t1 - t0; times[blockIdx.x * 4 + 1] = t_h1; times[blockIdx.x * 4 + 2] = t_h2; }
  if (bad) err[1] = bad;
}

// ------------------------------------------------------------------------------------------------ P4
// every group adds the same 4096 rows; member r owns rows [128 r, 128 r + 128).  FORM 0: i64, 8 active lanes per wave (8 rows per wave);
// 1: i64, 64 lanes of waves 0/1; 2: f32 64 lanes; 3: f32 8 lanes per wave
template <int FORM>
__global__ __launch_bounds__(NT) void atomics_probe(u64* acc64, float* acc32, int rounds) {
  extern __shared__ unsigned char smem[];
  const int r = blockIdx.x / GROUPS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) smem[0] = 1;
  for (int it = 0; it < rounds; ++it) {
    if (FORM == 0) { if ((lane & 7) == 7) __hip_atomic_fetch_add(acc64 + r * 128 + wave * 8 + (lane >> 3), 3ull, RLX, AGENT); }
    else if (FORM == 1) { if (wave < 2) __hip_atomic_fetch_add(acc64 + r * 128 + wave * 64 + lane, 3ull, RLX, AGENT); }
    else if (FORM == 2) { if (wave < 2) __hip_atomic_fetch_add(acc32 + r * 128 + wave * 64 + lane, 1.0f, RLX, AGENT); }
    else { if ((lane & 7) == 7) __hip_atomic_fetch_add(acc32 + r * 128 + wave * 8 + (lane >> 3), 1.0f, RLX, AGENT); }
  }
}

// ------------------------------------------------------------------------------------------------ P5
// writer: dirties `kb` KB the way the producer would (MODE 0 plain stores by 256 workgroups, 1: i64 atomics); reader: every workgroup
// loads all `kb` KB (16 B per lane, as many loads in flight as it takes), reduces, barrier, leaves.
template <int MODE>
__global__ __launch_bounds__(NT) void p5_writer(u64* buf, int kb) {
  const int n64 = kb * 128;
  for (int i = blockIdx.x * NT + threadIdx.x; i < n64; i += NWG * NT) {
    if (MODE == 0) buf[i] = i * 3ull;
    else __hip_atomic_fetch_add(buf + i, 3ull, RLX, AGENT);
  }
}
template <int NLOADS>
__global__ __launch_bounds__(NT) void p5_reader(const u32x4* buf, unsigned* sink) {
  extern __shared__ unsigned char smem[];
  unsigned* lds = reinterpret_cast<unsigned*>(smem);
  u32x4 v[NLOADS > 0 ? NLOADS : 1];
#pragma unroll
  for (int i = 0; i < NLOADS; ++i) v[i] = buf[i * NT + threadIdx.x];
  unsigned a = 0;
#pragma unroll
  for (int i = 0; i < NLOADS; ++i) a ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
  lds[threadIdx.x] = a;
  __syncthreads();
  if (lds[(threadIdx.x * 7) % NT] == 0x1234567u) sink[threadIdx.x] = a;
}

// ------------------------------------------------------------------------------------------------ host
static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

template <typename F>
static double time_graph(hipStream_t st, int nl, F&& enqueue) {
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int i = 0; i < nl; ++i) enqueue(i);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    best = std::min(best, ms);
  }
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  return best * 1e3 / (5.0 * nl);
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  unsigned *state, *err, *sink, *cens; u64 *times, *gran, *acc64; float* acc32; uint32_t* pool;
  const size_t pool_bytes = size_t(1024) << 20;
  CK(hipMalloc(&state, 16384)); CK(hipMalloc(&err, 64)); CK(hipMalloc(&sink, 8192)); CK(hipMalloc(&cens, 64 * NWG * 8));
  CK(hipMalloc(&times, NWG * 4 * 8)); CK(hipMalloc(&gran, GROUPS * 2 * 1024 * 8)); CK(hipMalloc(&acc64, 4096 * 8 * 8)); CK(hipMalloc(&acc32, 4096 * 4));
  CK(hipMalloc(&pool, pool_bytes)); CK(hipMemset(pool, 1, pool_bytes));
  CK(hipMemset(err, 0, 64));
  hipFuncSetAttribute(reinterpret_cast<const void*>(census), hipFuncAttributeMaxDynamicSharedMemorySize, BIG_LDS);

  // ---- P1
  {
    CK(hipMemset(cens, 0xFF, 64 * NWG * 8));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 64; ++i) hipLaunchKernelGGL(census, dim3(NWG), dim3(NT), BIG_LDS, st, cens, i);
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    std::vector<unsigned> h(64 * NWG * 2);
    CK(hipMemcpy(h.data(), cens, h.size() * 4, hipMemcpyDeviceToHost));
    int rr_ok = 0, one_per_cu = 0, same_base = 0; unsigned base0 = h[0];
    int hist[8][8] = {};
    for (int l = 0; l < 64; ++l) {
      bool rr = true; int cnt[8] = {};
      std::vector<unsigned> ids;
      for (int b = 0; b < NWG; ++b) {
        const unsigned x = h[(l * NWG + b) * 2], id = h[(l * NWG + b) * 2 + 1];
        if (x != ((h[l * NWG * 2] + b) & 7)) rr = false;
        cnt[x & 7]++;
        hist[b & 7][x & 7]++;
        // HW_ID: cu_id bits 11:8, sh 12, se 15:13 ; plus the XCD
        ids.push_back((x << 16) | ((id >> 8) & 0xFF));
      }
      std::sort(ids.begin(), ids.end());
      const bool uniq = std::unique(ids.begin(), ids.end()) == ids.end();
      bool even = true; for (int x = 0; x < 8; ++x) even &= cnt[x] == 32;
      rr_ok += rr && even; one_per_cu += uniq; same_base += h[l * NWG * 2] == base0;
    }
    printf("P1 census, 64 replayed launches of 256 x 1024 threads (100 KB LDS):\n");
    printf("   launches with XCC_ID(b) == (XCC_ID(0) + b) %% 8 for all b and 32 workgroups per XCD: %d / 64\n", rr_ok);
    printf("   launches with 256 distinct (XCD, SE, CU) placements (one workgroup per CU):          %d / 64\n", one_per_cu);
    printf("   launches whose block 0 sits on the same XCD as in launch 0 (XCC %u):                  %d / 64\n", base0, same_base);
    printf("   histogram [blockIdx %% 8][XCC_ID] over all launches:\n");
    for (int a = 0; a < 8; ++a) { printf("     b%%8=%d:", a); for (int x = 0; x < 8; ++x) printf(" %5d", hist[a][x]); printf("\n"); }
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }

  auto report_times = [&](const char* name, int rounds, int cols, int col) {
    std::vector<u64> h(NWG * cols);
    CK(hipMemcpy(h.data(), times, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> v;
    for (int b = 0; b < NWG; ++b) v.push_back(h[b * cols + col] * 0.01 / rounds);
    std::sort(v.begin(), v.end());
    printf("   %-44s per round: median %.3f us, min %.3f, max %.3f\n", name, v[NWG / 2], v[0], v[NWG - 1]);
  };
  auto check_err = [&](const char* what) {
    unsigned ev[2]; CK(hipMemcpy(ev, err, 8, hipMemcpyDeviceToHost));
    if (ev[0] || ev[1]) printf("   !! %s: timeout code %u, bad payloads %u\n", what, ev[0], ev[1]);
    CK(hipMemset(err, 0, 64));
  };

  // ---- P2
  printf("P2 barrier of the 32 workgroups with equal blockIdx %% 8 (wave 0 of each; 64 rounds inside one launch):\n");
  {
    const int rounds = 64;
#define RUN_BAR(FORM, LOAD, NAME) do { \
      hipFuncSetAttribute(reinterpret_cast<const void*>(barrier_probe<FORM, LOAD>), hipFuncAttributeMaxDynamicSharedMemorySize, BIG_LDS); \
      for (int rep = 0; rep < 2; ++rep) { CK(hipMemsetAsync(state, 0, 16384, st)); \
        hipLaunchKernelGGL((barrier_probe<FORM, LOAD>), dim3(NWG), dim3(NT), BIG_LDS, st, state, times, err, pool, rounds, sink); } \
      CK(hipStreamSynchronize(st)); report_times(NAME, rounds, 1, 0); check_err(NAME); } while (0)
    RUN_BAR(0, 0, "agent atomic add + sc1 poll, idle");
    RUN_BAR(1, 0, "workgroup-scope atomic add + sc1 poll, idle");
    RUN_BAR(2, 0, "flag word per member, 32-lane sc1 poll, idle");
    RUN_BAR(0, 64, "agent atomic add, beside 15 x 64 KB / CU stream");
    RUN_BAR(2, 64, "flag words, beside 15 x 64 KB / CU stream");
  }

  // ---- P3
  printf("P3 two chained hand-offs per round (28 x 14 granules -> 4 heads x 192; 4 x 64 granules -> 32 x 256), 64 rounds in one launch:\n");
  {
    const int rounds = 64;
#define RUN_CHAIN(FLAGS, LOAD, NAME) do { \
      hipFuncSetAttribute(reinterpret_cast<const void*>(chain_probe<FLAGS, LOAD>), hipFuncAttributeMaxDynamicSharedMemorySize, BIG_LDS); \
      for (int rep = 0; rep < 2; ++rep) { CK(hipMemsetAsync(state, 0, 16384, st)); CK(hipMemsetAsync(gran, 0, GROUPS * 2 * 1024 * 8, st)); \
        hipLaunchKernelGGL((chain_probe<FLAGS, LOAD>), dim3(NWG), dim3(NT), BIG_LDS, st, gran, state, times, err, pool, rounds, sink); } \
      CK(hipStreamSynchronize(st)); report_times(NAME " (whole round)", rounds, 4, 0); \
      { std::vector<u64> h(NWG * 4); CK(hipMemcpy(h.data(), times, h.size() * 8, hipMemcpyDeviceToHost)); \
        std::vector<double> a, b; for (int x = 0; x < NWG; ++x) { if (x / 8 < 4) a.push_back(h[x * 4 + 1] * 0.01 / rounds); b.push_back(h[x * 4 + 2] * 0.01 / rounds); } \
        printf("      wait inside hand-off 1 (heads): median %.3f us; wait inside hand-off 2 (all): median %.3f us\n", med(a), med(b)); } \
      check_err(NAME); } while (0)
    RUN_CHAIN(0, 0, "granules, idle");
    RUN_CHAIN(1, 0, "sc1 payload + drained counter, idle");
    RUN_CHAIN(0, 16, "granules, 12 waves/CU streaming 16 KB each in rounds 1-8");
    RUN_CHAIN(2, 0, "XCD-local granules (sc0 stores, sc1 loads, group = XCC_ID), idle");
    RUN_CHAIN(2, 16, "XCD-local granules, 12 waves/CU streaming 16 KB each in rounds 1-8");
  }

  // ---- P4
  printf("P4 eight groups adding 4096 partial sums into one accumulator (100 rounds per launch, 20 launches per graph):\n");
  {
    const int rounds = 100;
#define RUN_ATOM(FORM, NAME) do { \
      hipFuncSetAttribute(reinterpret_cast<const void*>(atomics_probe<FORM>), hipFuncAttributeMaxDynamicSharedMemorySize, BIG_LDS); \
      const double us = time_graph(st, 20, [&](int) { hipLaunchKernelGGL((atomics_probe<FORM>), dim3(NWG), dim3(NT), BIG_LDS, st, acc64, acc32, rounds); }); \
      const double us1 = time_graph(st, 20, [&](int) { hipLaunchKernelGGL((atomics_probe<FORM>), dim3(NWG), dim3(NT), BIG_LDS, st, acc64, acc32, 1); }); \
      printf("   %-40s %.3f us per round of 32768 adds (launch with 1 round: %.2f us, with 100: %.2f us)\n", NAME, (us - us1) / (rounds - 1), us1, us); } while (0)
    CK(hipMemset(acc64, 0, 4096 * 8)); CK(hipMemset(acc32, 0, 4096 * 4));
    RUN_ATOM(0, "i64, 8 lanes per wave (8 rows per wave)");
    RUN_ATOM(1, "i64, 64 lanes of 2 waves");
    RUN_ATOM(2, "f32, 64 lanes of 2 waves");
    RUN_ATOM(3, "f32, 8 lanes per wave");
  }

  // ---- P5
  printf("P5 (writer launch, reader launch) pairs: every one of 256 workgroups loads X KB the writer has just written, barrier, exit:\n");
  {
    u64* buf; CK(hipMalloc(&buf, 256 * 1024)); CK(hipMemset(buf, 0, 256 * 1024));
#define RUN_P5(MODE, NL, NAME) do { \
      hipFuncSetAttribute(reinterpret_cast<const void*>(p5_reader<NL>), hipFuncAttributeMaxDynamicSharedMemorySize, BIG_LDS); \
      const double us = time_graph(st, 20, [&](int) { hipLaunchKernelGGL((p5_writer<MODE>), dim3(NWG), dim3(NT), 0, st, buf, NL * 16); \
                                                     hipLaunchKernelGGL((p5_reader<NL>), dim3(NWG), dim3(NT), BIG_LDS, st, reinterpret_cast<const u32x4*>(buf), sink); }); \
      printf("   %-52s %.2f us per pair\n", NAME, us); } while (0)
    RUN_P5(0, 0, "plain writer of 0 KB, reader loads nothing");
    RUN_P5(0, 1, "plain stores, 16 KB");
    RUN_P5(0, 2, "plain stores, 32 KB");
    RUN_P5(0, 3, "plain stores, 48 KB");
    RUN_P5(0, 8, "plain stores, 128 KB");
    RUN_P5(1, 1, "i64 atomics, 16 KB");
    RUN_P5(1, 2, "i64 atomics, 32 KB");
    RUN_P5(1, 3, "i64 atomics, 48 KB");
  }
  printf("done\n");
  return 0;
}
