// One-shot peer-to-peer all-reduce(sum) of a short activation vector over the GPUs of one node (tensor-parallel decode:
// the exchange after o_proj and after down_proj, gbx-lm_amd/tp.py) -- build-side extension, the reference has no TP.
//
// RCCL's ring all-reduce is built for bandwidth; an 8 KiB vector at batch 1 is pure latency (64 calls per token).  Here every rank
// PUSHES its vector into an inbox inside every peer's memory (one hop over xGMI, posted writes) and then sums what arrived in ITS
// OWN inbox, polling local memory only:
//   * the unit is an 8-byte granule {tag = call epoch, payload = two 16-bit values}, written by ONE system-scope store and read by
//     ONE system-scope load: the data is its own flag, no fence, no separate ready word (cdna_hip_programming.md Guideline 16, R2);
//   * sums run in rank order 0 .. world-1 in fp32 with ONE rounding to the activation dtype, so every rank ends up with the same
//     bits (RCCL's ring rounds after every hop and in a rank-dependent order);
//   * inboxes are double-buffered by epoch parity: a rank can only be one call ahead of a peer (it needs the peer's granules of call
//     e to finish call e), so call e + 1 never overwrites granules of call e - 1 that a peer may still be reading ... of call e - 1
//     is finished by everyone once anyone has seen all of call e;
//   * the epoch lives in device memory and is advanced by the launch itself (last workgroup out), so a captured HIP graph replays it.
// Inboxes are exported / imported with hipIpc*; the spin is bounded: a launch that gives up raises the device-visible error word
// (gbx_p2p_status, tp.P2PAllReduce.check -- called at every synchronisation point of bench.py and the tests) and returns; it never
// traps and never waits endlessly.
#include "gbx_device.h"
#include "gbx_env.h"
#include "kernels.h"
#include <cstring>

namespace gbx {

constexpr int P2P_MAX_WORLD = 8;
constexpr int P2P_BLOCKS = 8;

struct P2PArgs {
  unsigned short* x;                       // [n] in / out
  unsigned long long* inbox[P2P_MAX_WORLD];   // rank p's inbox as mapped in this process (own included)
  unsigned* state;                         // [0] epoch of the last finished call, [1] workgroups done in this call
  int n, rank, world, cap;                 // cap = granules per (parity, source) slab
  unsigned spin_limit;                     // polls of one granule before the launch gives up (2^26 with s_sleep 8: tens of seconds)
};

typedef __attribute__((address_space(1))) unsigned long long p2p_g64;

template <bool BF16>
__global__ __launch_bounds__(256) void p2p_allreduce_kernel(const P2PArgs a) {
  const unsigned e = __hip_atomic_load(a.state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
  const int ng = a.n / 2;
  const size_t slab = static_cast<size_t>(a.cap);
  const size_t par = (e & 1u) * static_cast<size_t>(a.world) * slab;
  const unsigned* xw = reinterpret_cast<const unsigned*>(a.x);
  // ---- push: my granules into every peer's inbox[parity][my rank] ----
  for (int i = blockIdx.x * 256 + threadIdx.x; i < ng; i += gridDim.x * 256) {
    const unsigned long long g = (static_cast<unsigned long long>(e) << 32) | xw[i];
    for (int p = 0; p < a.world; ++p) {
      if (p == a.rank) continue;
      __hip_atomic_store((p2p_g64*)(a.inbox[p] + par + static_cast<size_t>(a.rank) * slab + i), g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  // ---- pull: sum what arrived in my own inbox, in rank order ----
  for (int i = blockIdx.x * 256 + threadIdx.x; i < ng; i += gridDim.x * 256) {
    float s0 = 0.f, s1 = 0.f;
    for (int s = 0; s < a.world; ++s) {
      unsigned w;
      if (s == a.rank) w = xw[i];
      else {
        p2p_g64* src = (p2p_g64*)(a.inbox[a.rank] + par + static_cast<size_t>(s) * slab + i);
        unsigned long long g;
        unsigned spins = 0;
        for (;;) {
          g = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          if (static_cast<unsigned>(g >> 32) == e) break;
          // a peer that never arrived (tens of seconds: rank skew of a graph instantiate or a slow first launch is NOT that):
          // raise the device-visible error word (gbx_p2p_status) and leave with whatever is there -- never hang the queue,
          // never take the whole context down with a trap
          if (++spins > a.spin_limit) { __hip_atomic_store(a.state + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
          __builtin_amdgcn_s_sleep(8);
        }
        w = static_cast<unsigned>(g);
      }
      s0 += act_lo<BF16>(w);
      s1 += act_hi<BF16>(w);
    }
    reinterpret_cast<unsigned*>(a.x)[i] = static_cast<unsigned>(f32_to_act<BF16>(s0)) | (static_cast<unsigned>(f32_to_act<BF16>(s1)) << 16);
  }
  // ---- last workgroup out advances the epoch (stream order makes the next call see it) ----
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned done = __hip_atomic_fetch_add(a.state + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (done == gridDim.x - 1) {
      __hip_atomic_store(a.state + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.state, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

hipError_t launch_p2p_allreduce(void* x, int n, int rank, int world, void* const* inbox, int cap, void* state, bool bf16, hipStream_t s) {
  if (world < 2 || world > P2P_MAX_WORLD || rank < 0 || rank >= world || n <= 0 || (n & 1) || n / 2 > cap) return hipErrorInvalidValue;
  P2PArgs a{};
  a.x = static_cast<unsigned short*>(x);
  for (int p = 0; p < world; ++p) a.inbox[p] = static_cast<unsigned long long*>(inbox[p]);
  a.state = static_cast<unsigned*>(state);
  a.n = n; a.rank = rank; a.world = world; a.cap = cap;
  static const int spin_log2 = env_switch("GBX_P2P_SPIN_LOG2");
  a.spin_limit = 1u << (spin_log2 < 4 ? 4 : (spin_log2 > 30 ? 30 : spin_log2));
  const int blocks = std::min(P2P_BLOCKS, (n / 2 + 255) / 256);
  if (bf16) hipLaunchKernelGGL((p2p_allreduce_kernel<true>), dim3(blocks), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((p2p_allreduce_kernel<false>), dim3(blocks), dim3(256), 0, s, a);
  return hipGetLastError();
}

}  // namespace gbx
