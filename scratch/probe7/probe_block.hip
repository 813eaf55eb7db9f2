// Round-6 step 0 (i): the chip-wide release a one-launch decoder block needs between o_proj's partial sums and the MLP's prologue,
// with 256 resident 1024-thread workgroups, idle and beside an LDS-DMA weight stream.
//   hipcc --offload-arch=gfx950 -O3 -o bin/probe_block probe_block.hip
//
// One round = one block's release chain (all hand-offs as in the product kernels: XCD-local hops are sc0 granule stores + sc1 polls with the
// group taken from XCC_ID; cross-XCD words are agent-scope stores / loads; the sums are agent-scope 64-bit atomic adds):
//   every workgroup : 128 atomic adds into acc[r * 128 ..] (o_proj's fixed-point sums: all 8 XCDs add to the same 4096 words), drained;
//                     barrier; ARRIVE = one granule in the XCD's arrival line
//   XCD leader (r=0): wave 0 polls the 32 arrival granules -> stores flag[g] (agent scope) -> polls the 8 flags -> the whole workgroup loads
//                     the 32 KB of sums (sc1), x and the norm weight, reduces the statistic, normalises, PUBLISHES 2048 granules
//                     (two bf16 each) XCD-locally
//   every workgroup : thread t polls granules 2 t, 2 t + 1 (its 4-value image item), checks the payload, barrier
// Stamps (s_memrealtime, 100 MHz, chip-wide clock) per workgroup and round; the host reports medians over rounds.
// Backgrounds: 0 idle; 1 waves 2-15 of every workgroup issue `kb` KB of LDS-DMA (nt) at the START of the round (a stream still in flight
// when the chain runs); 2 the same issued right after ARRIVE (down_proj's words requested while the release is pending).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned long long u64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef __attribute__((address_space(1))) unsigned gu32;
typedef __attribute__((address_space(1))) u64 gu64;
#define RLX __ATOMIC_RELAXED
#define AGENT __HIP_MEMORY_SCOPE_AGENT
#define WGSCOPE __HIP_MEMORY_SCOPE_WORKGROUP
__device__ __forceinline__ u64 now() { return __builtin_amdgcn_s_memrealtime(); }
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7; }

constexpr int NWG = 256, NT = 1024, NW = 16, GROUPS = 8, GSZ = 32, D = 4096;
constexpr size_t LDS_BYTES = 156 * 1024;
constexpr int NST = 8;                       // stamps per workgroup and round
constexpr unsigned SPIN = 1u << 22;

struct Args {
  u64* acc;            // [4][D] fixed-point sums, slot = round % 4
  u64* arrive;         // [8][32] granules
  unsigned* flag;      // [8][64] one line per XCD
  u64* xg;             // [8][2048] granules
  const unsigned short* x;     // [D]
  const unsigned short* w;     // [D]
  const unsigned char* pool;   // weight pool for the background stream
  u64* stamps;         // [rounds][NWG][NST]
  unsigned* err;
  int rounds, kb;
  size_t pool_stride;  // bytes between rounds' regions
};

__device__ __forceinline__ float bf16_to_f32(unsigned short v) { return __uint_as_float(static_cast<unsigned>(v) << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
  unsigned u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return static_cast<unsigned short>(u >> 16);
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

template <int BG>
__global__ __launch_bounds__(NT) void release_chain(const Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* red = reinterpret_cast<float*>(smem);                 // [16]
  unsigned* xl = reinterpret_cast<unsigned*>(smem + 256);      // [2048] received x'
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = static_cast<int>(xcc_id()), r = static_cast<int>(blockIdx.x) >> 3;
  gu64* arr = (gu64*)(a.arrive) + g * GSZ;
  gu32* flg = (gu32*)(a.flag);
  gu64* xg = (gu64*)(a.xg) + static_cast<size_t>(g) * 2048;
  const __amdgpu_buffer_rsrc_t pd = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.pool), 0, 0x7FFFFFFF, 0x00020000);
  unsigned bad = 0;
  auto dma = [&](int it) {
    // waves 2..15: pieces n = wave - 2, + 14, ... of this workgroup's kb KB
    if (wave >= 2) {
      for (int n = wave - 2; n < a.kb; n += 14)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(pd, (__attribute__((address_space(3))) void*)(smem + 16 * 1024 + n * 1024), 16, lane * 16,
                                                 static_cast<int>((it & 15) * a.pool_stride + (static_cast<size_t>(blockIdx.x) * a.kb + n) * 1024), 0, 2);
    }
  };
  for (int it = 1; it <= a.rounds; ++it) {
    const unsigned tag = static_cast<unsigned>(it);
    u64* st = a.stamps + (static_cast<size_t>(it - 1) * NWG + blockIdx.x) * NST;
    u64* acc = a.acc + (it & 3) * D;
    if (tid == 0) st[0] = now();
    if (BG == 1) dma(it);
    // ---- o_proj-like sums
    if (tid < 128) {
      __hip_atomic_fetch_add((gu64*)(acc) + r * 128 + tid, 1ull, RLX, AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (wave == 0) {
      if (lane == 0) {
        st[1] = now();
        __hip_atomic_store(arr + r, (static_cast<u64>(tag) << 32) | static_cast<unsigned>(r), RLX, WGSCOPE);
      }
    }
    if (BG == 2) dma(it);
    if (r == 0) {
      // ---- leader
      if (wave == 0) {
        for (unsigned spins = 0;;) {
          const u64 v = __hip_atomic_load(arr + (lane & 31), RLX, AGENT);
          if (__all(static_cast<unsigned>(v >> 32) == tag)) break;
          if (++spins > SPIN) { if (lane == 0) a.err[0] = 1; break; }
          __builtin_amdgcn_s_sleep(1);
        }
        if (lane == 0) { st[2] = now(); __hip_atomic_store(flg + g * 64, tag, RLX, AGENT); }
        for (unsigned spins = 0;;) {
          const unsigned v = __hip_atomic_load(flg + (lane & 7) * 64, RLX, AGENT);
          if (__all(v == tag)) break;
          if (++spins > SPIN) { if (lane == 0) a.err[0] = 2; break; }
          __builtin_amdgcn_s_sleep(1);
        }
        if (lane == 0) st[3] = now();
      }
      __syncthreads();
      // the sums: 4 x u64 per thread with sc1 loads (the adds were memory-side atomics: nothing of them is in this XCD's L2)
      const __amdgpu_buffer_rsrc_t ad = __builtin_amdgcn_make_buffer_rsrc(acc, 0, D * 8, 0x00020000);
      const u32x4 s0 = __builtin_amdgcn_raw_buffer_load_b128(ad, tid * 32, 0, 16);        // aux bit 4: sc1
      const u32x4 s1 = __builtin_amdgcn_raw_buffer_load_b128(ad, tid * 32 + 16, 0, 16);
      const u32x2 xv = *reinterpret_cast<const u32x2*>(a.x + 4 * tid);
      const u32x2 wv = *reinterpret_cast<const u32x2*>(a.w + 4 * tid);
      const u64 sv[4] = {(static_cast<u64>(s0.y) << 32) | s0.x, (static_cast<u64>(s0.w) << 32) | s0.z,
                         (static_cast<u64>(s1.y) << 32) | s1.x, (static_cast<u64>(s1.w) << 32) | s1.z};
      const u64 want = 8ull * static_cast<u64>((it + 3) >> 2);
      bool okv = sv[0] == want && sv[1] == want && sv[2] == want && sv[3] == want;
      if (!okv) bad++;
      float xf[4] = {bf16_to_f32(static_cast<unsigned short>(xv.x)), bf16_to_f32(static_cast<unsigned short>(xv.x >> 16)),
                     bf16_to_f32(static_cast<unsigned short>(xv.y)), bf16_to_f32(static_cast<unsigned short>(xv.y >> 16))};
      float ssq = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) { xf[i] = bf16_to_f32(f32_to_bf16(xf[i] + static_cast<float>(static_cast<long long>(sv[i])) * (1.0f / 1024.f))); ssq = fmaf(xf[i], xf[i], ssq); }
      ssq = wave_sum(ssq);
      if (lane == 0) red[wave] = ssq;
      if (tid == 0) st[4] = now();
      __syncthreads();
      float tot = 0.f;
#pragma unroll
      for (int i = 0; i < NW; ++i) tot += red[i];
      const float rstd = rsqrtf(tot / D + 1e-5f);
      const float wf[4] = {bf16_to_f32(static_cast<unsigned short>(wv.x)), bf16_to_f32(static_cast<unsigned short>(wv.x >> 16)),
                           bf16_to_f32(static_cast<unsigned short>(wv.y)), bf16_to_f32(static_cast<unsigned short>(wv.y >> 16))};
      unsigned short yo[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) yo[i] = f32_to_bf16(xf[i] * rstd * wf[i]);
      // payload check word: the low 16 bits carry y, the probe's consumers verify (tid, ok) through the second granule's high half
      const unsigned p0 = static_cast<unsigned>(yo[0]) | (static_cast<unsigned>(yo[1]) << 16);
      const unsigned p1 = okv ? (static_cast<unsigned>(yo[2]) | (static_cast<unsigned>(yo[3]) << 16)) : 0xDEADBEEFu;
      __hip_atomic_store(xg + 2 * tid, (static_cast<u64>(tag) << 32) | p0, RLX, WGSCOPE);
      __hip_atomic_store(xg + 2 * tid + 1, (static_cast<u64>(tag) << 32) | p1, RLX, WGSCOPE);
      if (tid == 0) st[5] = now();
    }
    // ---- everyone: the item's two granules
    {
      u64 xa, xb;
      for (unsigned spins = 0;;) {
        xa = __hip_atomic_load(xg + 2 * tid, RLX, AGENT);
        xb = __hip_atomic_load(xg + 2 * tid + 1, RLX, AGENT);
        if (__all(static_cast<unsigned>(xa >> 32) == tag && static_cast<unsigned>(xb >> 32) == tag)) break;
        if (++spins > SPIN) { if (lane == 0) a.err[0] = 3; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      if (static_cast<unsigned>(xb) == 0xDEADBEEFu) bad++;
      xl[2 * tid] = static_cast<unsigned>(xa);
      xl[2 * tid + 1] = static_cast<unsigned>(xb);
    }
    if (tid == 0) st[6] = now();
    if (BG != 0 && wave >= 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) st[7] = now();
  }
  if (bad) atomicAdd(a.err + 1, bad);
}

// ---- LDS-DMA fill alone: every CU pulls kb KB (16 waves x pieces of 1 KB), issue -> landed
__global__ __launch_bounds__(NT) void fill_probe(const unsigned char* pool, u64* times, int kb, int nt, size_t off) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const __amdgpu_buffer_rsrc_t pd = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(pool), 0, 0x7FFFFFFF, 0x00020000);
  __syncthreads();
  const u64 t0 = now();
  if (nt) {
    for (int n = wave; n < kb; n += NW)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(pd, (__attribute__((address_space(3))) void*)(smem + n * 1024), 16, lane * 16,
                                               static_cast<int>(off + (static_cast<size_t>(blockIdx.x) * kb + n) * 1024), 0, 2);
  } else {
    for (int n = wave; n < kb; n += NW)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(pd, (__attribute__((address_space(3))) void*)(smem + n * 1024), 16, lane * 16,
                                               static_cast<int>(off + (static_cast<size_t>(blockIdx.x) * kb + n) * 1024), 0, 0);
  }
  const u64 t1 = now();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const u64 t2 = now();
  if (tid == 0) { times[blockIdx.x * 4] = t0; times[blockIdx.x * 4 + 1] = t1; times[blockIdx.x * 4 + 2] = t2; }
}

static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }
static double pct(std::vector<double> v, double p) { std::sort(v.begin(), v.end()); return v[std::min(v.size() - 1, static_cast<size_t>(p * v.size()))]; }

int main() {
  hipStream_t stream; CK(hipStreamCreate(&stream));
  const int rounds = 48;
  Args a{};
  unsigned char* pool; const size_t pool_bytes = size_t(1024) << 20;
  CK(hipMalloc(&pool, pool_bytes)); CK(hipMemset(pool, 1, pool_bytes));
  CK(hipMalloc(&a.acc, 4 * D * 8)); CK(hipMalloc(&a.arrive, 8 * 32 * 8)); CK(hipMalloc(&a.flag, 8 * 64 * 4)); CK(hipMalloc(&a.xg, 8 * 2048 * 8));
  unsigned short *x, *w; CK(hipMalloc(&x, D * 2)); CK(hipMalloc(&w, D * 2));
  { std::vector<unsigned short> h(D); for (int i = 0; i < D; ++i) h[i] = 0x3F80 + (i % 7); CK(hipMemcpy(x, h.data(), D * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(w, h.data(), D * 2, hipMemcpyHostToDevice)); }
  a.x = x; a.w = w; a.pool = pool; a.pool_stride = size_t(60) << 20;
  CK(hipMalloc(&a.stamps, size_t(rounds) * NWG * NST * 8)); CK(hipMalloc(&a.err, 64));
  a.rounds = rounds;
  auto run = [&](int bg, int kb, const char* name) {
    a.kb = kb;
    CK(hipMemset(a.acc, 0, 4 * D * 8)); CK(hipMemset(a.arrive, 0, 8 * 32 * 8)); CK(hipMemset(a.flag, 0, 8 * 64 * 4)); CK(hipMemset(a.xg, 0, 8 * 2048 * 8));
    CK(hipMemset(a.err, 0, 64)); CK(hipMemset(a.stamps, 0, size_t(rounds) * NWG * NST * 8));
#define LAUNCH(BG) do { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(release_chain<BG>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES)); \
    hipLaunchKernelGGL((release_chain<BG>), dim3(NWG), dim3(NT), LDS_BYTES, stream, a); } while (0)
    if (bg == 0) LAUNCH(0); else if (bg == 1) LAUNCH(1); else LAUNCH(2);
    CK(hipStreamSynchronize(stream));
    unsigned ev[2]; CK(hipMemcpy(ev, a.err, 8, hipMemcpyDeviceToHost));
    std::vector<u64> h(size_t(rounds) * NWG * NST);
    CK(hipMemcpy(h.data(), a.stamps, h.size() * 8, hipMemcpyDeviceToHost));
    // per round: last arrival = max st[1]; leader stages relative to it; member received: median / max of st[6] - last arrival
    std::vector<double> L32, L8, Lsum, Lpub, Mmed, Mmax, Rnd, Skew;
    for (int it = 8; it < rounds; ++it) {
      const u64* p = h.data() + size_t(it) * NWG * NST;
      u64 first_arr = ~0ull, last_arr = 0, t32 = 0, t8 = 0, tsum = 0, tpub = 0, mx = 0, st0 = ~0ull, e7 = 0;
      std::vector<double> m;
      for (int b = 0; b < NWG; ++b) { last_arr = std::max(last_arr, p[b * NST + 1]); first_arr = std::min(first_arr, p[b * NST + 1]); st0 = std::min(st0, p[b * NST]); e7 = std::max(e7, p[b * NST + 7]); }
      for (int b = 0; b < NWG; ++b) {
        if ((b >> 3) == 0) { t32 = std::max(t32, p[b * NST + 2]); t8 = std::max(t8, p[b * NST + 3]); tsum = std::max(tsum, p[b * NST + 4]); tpub = std::max(tpub, p[b * NST + 5]); }
        m.push_back((double(p[b * NST + 6]) - double(last_arr)) * 0.01);
        mx = std::max(mx, p[b * NST + 6]);
      }
      L32.push_back((double(t32) - double(last_arr)) * 0.01); L8.push_back((double(t8) - double(last_arr)) * 0.01);
      Lsum.push_back((double(tsum) - double(last_arr)) * 0.01); Lpub.push_back((double(tpub) - double(last_arr)) * 0.01);
      Mmed.push_back(med(m)); Mmax.push_back((double(mx) - double(last_arr)) * 0.01);
      Rnd.push_back((double(e7) - double(st0)) * 0.01); Skew.push_back((double(last_arr) - double(first_arr)) * 0.01);
    }
    printf("%-58s | arrival skew %.2f | after the LAST arrival: leaders hold 32 arrivals %.2f, 8 flags %.2f, sums+stat %.2f, published %.2f; members received: median %.2f (p90 of rounds %.2f), last %.2f (p90 %.2f) | round %.2f us%s\n",
           name, med(Skew), med(L32), med(L8), med(Lsum), med(Lpub), med(Mmed), pct(Mmed, 0.9), med(Mmax), pct(Mmax, 0.9), med(Rnd),
           (ev[0] || ev[1]) ? "  !! ERR" : "");
    if (ev[0] || ev[1]) printf("   !! timeout code %u, bad payloads %u\n", ev[0], ev[1]);
  };
  printf("release chain, %d rounds in one launch (first 8 dropped), 256 x 1024 threads, %zu KB LDS; times in us after the round's LAST arrival\n", rounds, LDS_BYTES / 1024);
  for (int rep = 0; rep < 2; ++rep) {
    run(0, 0, "idle");
    run(1, 56, "56 KB / CU LDS-DMA (nt) issued at round start");
    run(1, 112, "112 KB / CU LDS-DMA (nt) issued at round start");
    run(2, 56, "56 KB / CU LDS-DMA (nt) issued behind ARRIVE");
    run(2, 112, "112 KB / CU LDS-DMA (nt) issued behind ARRIVE");
  }
  // ---- fill alone
  {
    u64* times; CK(hipMalloc(&times, NWG * 4 * 8));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(fill_probe), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    printf("LDS-DMA fill alone (every CU, 16 waves, 1 KB pieces), issue start -> all landed (chip-wide: first start -> last landed):\n");
    const int kbs[4] = {28, 56, 112, 140};
    for (int nt = 0; nt < 2; ++nt)
      for (int k = 0; k < 4; ++k) {
        std::vector<double> per, chip, iss;
        for (int rep = 0; rep < 12; ++rep) {
          hipLaunchKernelGGL(fill_probe, dim3(NWG), dim3(NT), LDS_BYTES, stream, pool, times, kbs[k], nt, size_t(rep % 12) * (size_t(60) << 20));
          CK(hipStreamSynchronize(stream));
          std::vector<u64> h(NWG * 4); CK(hipMemcpy(h.data(), times, h.size() * 8, hipMemcpyDeviceToHost));
          u64 s = ~0ull, e = 0; std::vector<double> v, vi;
          for (int b = 0; b < NWG; ++b) { s = std::min(s, h[b * 4]); e = std::max(e, h[b * 4 + 2]); v.push_back((h[b * 4 + 2] - h[b * 4]) * 0.01); vi.push_back((h[b * 4 + 1] - h[b * 4]) * 0.01); }
          if (rep >= 2) { per.push_back(med(v)); chip.push_back((e - s) * 0.01); iss.push_back(med(vi)); }
        }
        printf("   %3d KB / CU, %s: issue %.2f us, per CU %.2f us (%.1f GB/s), chip %.2f us (%.2f TB/s)\n", kbs[k], nt ? "nt     " : "default", med(iss), med(per),
               kbs[k] * 1.024 / med(per), med(chip), kbs[k] * 1024.0 * NWG / med(chip) * 1e-6);
      }
  }
  printf("done\n");
  return 0;
}
