// One launch for [RoPE + KV-cache append + single-token attention + o_proj] of a decoder block at batch 1 (round 3):
// gbx_lm/models/qllama.py:78-96 -- `queries/keys = rope(...)`, `cache.update_and_fetch`, `scaled_dot_product_attention`,
// `self.o_proj(output)` -- where o_proj is the QuantizedLinear call of gbx_lm/models/quantized_linear_gba.py:195-203, M = 1.
//
// Why: in the 5-launch block the attention launch (3.9 us) and the o_proj launch (3.5 us) are both pure latency chains with a
// 1.3-1.4 us boundary each (profiles/r03_layer_timeline_before.txt); o_proj's 4.7 MB are 0.8 us of stream.  o_proj is a
// contraction over heads, so it splits over KV groups:   y = sum_g  Wo[:, cols of group g] . attn_out[group g].
// Grid [n_kv groups] x [D / 128 row blocks] (8 x 32 = 256 workgroups = one per CU for Llama-3-8B), workgroup (g, r):
//   1. requests its slice of Wo -- 128 rows x (REP * HD = 512 values = 128 B at 2 bits: one cache line per row) -- first;
//   2. r < REP: it is the PRODUCER of query head g * REP + r: all 16 waves sweep that head's cached positions (64 per step),
//      the 128 outputs (rounded to the activation dtype like the SDPA output) become their piece of the pre-shifted fp16 image
//      of gemv_impl.h and leave as 72 eight-byte {tag, value} granules, each ONE write-through (sc1) store: the data is the
//      flag (cdna_hip_programming.md Guideline 16, R2);
//   3. one wave re-reads the group's 288 granules (sc1 loads) until every tag is set, parks the values in LDS, and all waves
//      multiply the Wo slice (same in-place subnormal dot, scale / zero per group);
//   4. the 128 partial row sums are added into acc[D] as 64-bit FIXED-POINT integers (2^-32 units; integer adds are
//      associative, so the result does not depend on the arrival order -- fp32 atomics would make the token stream
//      non-reproducible).  The consumer (the gate|up launch, GemvArgs::x_acc) forms h + round(acc) in its prologue.
// First version (measured, kept out): every workgroup computed its group's four heads itself (no hand-off): 10.3 us per
// launch -- four heads of vector-ALU sweep per CU (2.5 us) and 32 workgroups x 4 heads hammering the same K / V lines of one L2.
// Forward progress: consumers wait only on workgroups with a LOWER linear block index (r < REP), which the dispatcher starts
// first; the spin is bounded (trap).  Tags are cleared by the launch that clears acc (GemvArgs::zero_buf covers both), so a
// constant tag serves every replay.
#include "gemv_impl.h"
#include "kernels.h"

namespace gbx {

struct AttnOprojArgs {
  const unsigned short* q;      // [n_heads * HD]   projection outputs, un-rotated
  const unsigned short* k;      // [n_kv * HD]
  const unsigned short* v;      // [n_kv * HD]
  unsigned short* kcache;       // [n_kv, capacity, HD]
  unsigned short* vcache;
  const float* inv_freq;        // [HD / 2]
  const long long* pos;         // device scalar
  const uint32_t* qweight;      // o_proj [D, n_heads * HD * bits / 32]
  const unsigned short* scales; // [D, n_heads * HD / group]
  const unsigned short* zeros;
  long long* acc;               // [D] fixed point (2^-32), zero on entry (the down_proj launch of the previous block clears it)
  unsigned long long* gran;     // [n_kv * 4 * 72] {tag, value} granules of the hand-off, zero on entry (cleared with acc)
  int n_heads, n_kv, capacity, D;
  int log2g;
  float rope_scale, sm_scale;
  int traditional;
#ifdef GBX_STAMPS
  unsigned long long* stamps;
#endif
};

constexpr float ACC_SCALE = 4294967296.f;       // 2^32

typedef __attribute__((address_space(1))) unsigned long long gu64_t;
constexpr int AO_GRANULES_PER_HEAD = 72;       // 64 image dwords (128 values) + 4 sub-slots x (sum, 2^(24-E))

template <int BITS, bool BF16>
__global__ __launch_bounds__(1024) void attn_oproj_kernel(const AttnOprojArgs a) {
  constexpr int HD = 128, REP = 4, KS = REP * HD;             // contraction slice of one KV group
  constexpr int RB = 128;                                     // o_proj rows per workgroup
  using F = Fmt<BITS>;
  using L = Lds<BITS>;
  constexpr int PAIRS = L::PAIRS, SSTR = L::SSTR, NSUB = F::NSUB, UP = F::UNIT_PAIRS;
  constexpr int NSLOT = KS / F::VPL;                          // 8 (2 bits) / 16 (3, 4 bits) slots per row slice
  constexpr int SPH = NSLOT / REP;                            // slots per head: 2 / 4
  constexpr int RPL = 64 / NSLOT;                             // rows per wave load
  constexpr int NL = RB / (16 * RPL);                         // loads per wave: 1 / 2
  constexpr int LP = HD / 8, PPW = 64 / LP, NG = 16;         // sweep geometry: 16 lanes per position, 4 positions per wave step; one merged state per wave
  constexpr int GPH = AO_GRANULES_PER_HEAD;
  constexpr float NEG = -1e30f;
  using WT = typename SlotWords<BITS>::type;

  __shared__ float q_s[HD], kn_s[HD], vn_s[HD];
  __shared__ float gm[NG], gl[NG];
  __shared__ float gacc[NG][HD + 8];
  __shared__ float o_s[HD];
  __shared__ __attribute__((aligned(16))) unsigned img_s[NSLOT * SSTR];
  __shared__ float2 xs_s[NSLOT * NSUB];

  const int g = blockIdx.x, rblk = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef GBX_STAMPS
#define AO_STAMP(i) do { if (lane == 0 && a.stamps) a.stamps[(static_cast<size_t>(blockIdx.y * gridDim.x + blockIdx.x) * 16 + wave) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define AO_STAMP(i) do {} while (0)
#endif
  AO_STAMP(0);

  // ---- 1. this wave's rows of the Wo slice (non-temporal: streamed once), their scale / zero -- requested first by the
  //         consumer workgroups, after the K / V prefetch by the producers (whose chain is the launch's critical path) ----
  const int Kfull = a.n_heads * HD;
  const int W = Kfull / 32 * BITS;                            // words per packed row
  const int G = Kfull >> a.log2g;
  const int slot = lane % NSLOT;
  WT wreg[NL];
  unsigned short sreg[NL], zreg[NL];
  int rowof[NL];
  auto issue_wo = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int row = rblk * RB + (wave * NL + i) * RPL + lane / NSLOT;
      rowof[i] = row;
      const int rc = min(row, a.D - 1);
      const uint32_t* src = a.qweight + static_cast<size_t>(rc) * W + (g * NSLOT + slot) * F::SLOT_WORDS;
      if constexpr (BITS == 3) {
        wreg[i].x = __builtin_nontemporal_load(src);
        wreg[i].y = __builtin_nontemporal_load(src + 1);
        wreg[i].z = __builtin_nontemporal_load(src + 2);
      } else {
        wreg[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src));
      }
      const int grp = (g * KS + slot * F::VPL) >> a.log2g;
      sreg[i] = a.scales[static_cast<size_t>(rc) * G + grp];
      zreg[i] = a.zeros[static_cast<size_t>(rc) * G + grp];
    }
  };
  if (rblk >= REP) issue_wo();
  AO_STAMP(8);
  gu64_t* const gran = (gu64_t*)(a.gran) + static_cast<size_t>(g) * (REP * GPH);   // explicit global address space: never a flat access

  // ---- 2. producer workgroups: the attention of query head g * REP + rblk, all 16 waves on one head -------------------
  if (rblk < REP) {
    const int h = g * REP + rblk;
    constexpr int UB0 = 2;                                    // the first 128 cached positions are requested before anything else
    const int d0 = (lane % LP) * 8, sub = lane / LP;
    const unsigned short* kbase = a.kcache + static_cast<size_t>(g) * a.capacity * HD;
    const unsigned short* vbase = a.vcache + static_cast<size_t>(g) * a.capacity * HD;
    u32x4 k0[UB0], v0[UB0];
#pragma unroll
    for (int u = 0; u < UB0; ++u) {
      const int tc = min(wave * PPW + u * 16 * PPW + sub, a.capacity - 1);
      k0[u] = *reinterpret_cast<const u32x4*>(kbase + static_cast<size_t>(tc) * HD + d0);
      v0[u] = *reinterpret_cast<const u32x4*>(vbase + static_cast<size_t>(tc) * HD + d0);
    }
    // operands of RoPE / the value copy: requested now, used once the position counter has arrived
    const bool isk = tid >= HD / 2;
    const int ri = isk ? tid - HD / 2 : tid;                  // rotation pair index (tid < HD)
    const int i1 = a.traditional ? 2 * ri : ri;
    const int i2 = a.traditional ? 2 * ri + 1 : ri + HD / 2;
    unsigned short rx1 = 0, rx2 = 0, vb = 0;
    float frq = 0.f;
    if (tid < HD) {
      const unsigned short* src = isk ? a.k + static_cast<size_t>(g) * HD : a.q + static_cast<size_t>(h) * HD;
      rx1 = src[i1];
      rx2 = src[i2];
      frq = a.inv_freq[ri];
    } else if (tid < 2 * HD) {
      vb = a.v[static_cast<size_t>(g) * HD + (tid - HD)];
    }
    AO_STAMP(9);
    const int pos = static_cast<int>(*a.pos);
    AO_STAMP(1);
    // RoPE: threads 0..63 rotate q, 64..127 rotate k, 128..255 copy v; the producer of the group's first head appends to the cache
    if (tid < HD) {
      const float x1 = act_to_f32<BF16>(rx1), x2 = act_to_f32<BF16>(rx2);
      const float rev = static_cast<float>(pos) * a.rope_scale * frq * 0.15915494309189535f;
      const float fr = rev - floorf(rev);
      const float sn = __builtin_amdgcn_sinf(fr), cs = __builtin_amdgcn_cosf(fr);
      const unsigned short r1 = f32_to_act<BF16>(x1 * cs - x2 * sn), r2 = f32_to_act<BF16>(x1 * sn + x2 * cs);
      if (isk) {
        kn_s[i1] = act_to_f32<BF16>(r1);
        kn_s[i2] = act_to_f32<BF16>(r2);
        if (rblk == 0) {
          unsigned short* dst = a.kcache + (static_cast<size_t>(g) * a.capacity + pos) * HD;
          dst[i1] = r1;
          dst[i2] = r2;
        }
      } else {
        q_s[i1] = act_to_f32<BF16>(r1) * a.sm_scale;
        q_s[i2] = act_to_f32<BF16>(r2) * a.sm_scale;
      }
    } else if (tid < 2 * HD) {
      const int j = tid - HD;
      vn_s[j] = act_to_f32<BF16>(vb);
      if (rblk == 0) a.vcache[(static_cast<size_t>(g) * a.capacity + pos) * HD + j] = vb;
    }
    __syncthreads();
    AO_STAMP(2);
    issue_wo();                               // off the producer's critical path: needed only after the hand-off

    float qr[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) qr[j] = q_s[d0 + j];
    float m = NEG, l = 0.f, acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    auto consume = [&](const u32x4& kq, const u32x4& vq, bool valid) {
      const unsigned kw[4] = {kq.x, kq.y, kq.z, kq.w};
      const unsigned vw[4] = {vq.x, vq.y, vq.z, vq.w};
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s = fmaf(qr[2 * j], act_lo<BF16>(kw[j]), s);
        s = fmaf(qr[2 * j + 1], act_hi<BF16>(kw[j]), s);
      }
      s += dpp_f32<0xB1>(s);
      s += dpp_f32<0x4E>(s);
      s += dpp_f32<0x141>(s);
      s += dpp_f32<0x140>(s);
      const float mn = valid ? fmaxf(m, s) : m;
      const float corr = __expf(m - mn);
      const float p = valid ? __expf(s - mn) : 0.f;
      l = l * corr + p;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[2 * j] = acc[2 * j] * corr + p * act_lo<BF16>(vw[j]);
        acc[2 * j + 1] = acc[2 * j + 1] * corr + p * act_hi<BF16>(vw[j]);
      }
      m = mn;
    };
#pragma unroll
    for (int u = 0; u < UB0; ++u) consume(k0[u], v0[u], wave * PPW + u * 16 * PPW + sub < pos);
    constexpr int UB = 2;
    for (int t0 = UB0 * 16 * PPW + wave * PPW; t0 < pos; t0 += 16 * PPW * UB) {
      u32x4 kv4[UB], vv4[UB];
      bool valid[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int t = t0 + u * 16 * PPW + sub;
        valid[u] = t < pos;
        const int tc = valid[u] ? t : pos - 1;
        kv4[u] = *reinterpret_cast<const u32x4*>(kbase + static_cast<size_t>(tc) * HD + d0);
        vv4[u] = *reinterpret_cast<const u32x4*>(vbase + static_cast<size_t>(tc) * HD + d0);
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) consume(kv4[u], vv4[u], valid[u]);
    }
    if (wave == 0 && sub == 0) {            // the new token itself (its k, v are in LDS)
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) s = fmaf(qr[j], kn_s[d0 + j], s);
      s += dpp_f32<0xB1>(s);
      s += dpp_f32<0x4E>(s);
      s += dpp_f32<0x141>(s);
      s += dpp_f32<0x140>(s);
      const float mn = fmaxf(m, s);
      const float corr = __expf(m - mn), p = __expf(s - mn);
      l = l * corr + p;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = acc[j] * corr + p * vn_s[d0 + j];
      m = mn;
    }
    // the four position sub-groups of the wave (lanes l, l ^ 16, l ^ 32, l ^ 48 hold the same 8 features) merge their
    // online-softmax states in registers first: 16 states reach LDS instead of 64 (a 64-term merge loop took 3.6 us)
#pragma unroll
    for (int x = 16; x <= 32; x <<= 1) {
      const float m2 = __shfl_xor(m, x), l2 = __shfl_xor(l, x);
      const float mn = fmaxf(m, m2);
      const float c1 = __expf(m - mn), c2 = __expf(m2 - mn);
      l = l * c1 + l2 * c2;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = acc[j] * c1 + __shfl_xor(acc[j], x) * c2;
      m = mn;
    }
    const int gi = wave;
    AO_STAMP(3);
    if (lane == 0) { gm[gi] = m; gl[gi] = l; }
    if (sub == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) gacc[gi][d0 + j] = acc[j];
    }
    __syncthreads();
    AO_STAMP(4);
    if (tid < HD) {                         // merge the NG partial states; one output feature per thread
      float M = NEG;
#pragma unroll
      for (int i = 0; i < NG; ++i) M = fmaxf(M, gm[i]);
      float num = 0.f, den = 0.f;
#pragma unroll
      for (int i = 0; i < NG; ++i) {
        const float w = __expf(gm[i] - M);
        num = fmaf(w, gacc[i][tid], num);
        den = fmaf(w, gl[i], den);
      }
      o_s[tid] = round_bf16_or_f16(num / den, BF16);          // the SDPA output is an activation-dtype tensor
      AO_STAMP(10);
    }
    __syncthreads();
    AO_STAMP(11);
    // the head's 128 outputs -> pre-shifted fp16 pairs (the image of gemv_impl.h, section 3b) -> granules
    {
      constexpr int PPU = (BITS == 3) ? UP : 2;               // pairs per item
      constexpr int TPU = UP / PPU;
      constexpr int NIT = (HD / F::UNIT_VALS) * TPU;          // 32 items (4 at 3 bits)
      constexpr int GL = (BITS == 3) ? 1 : 16 / PPU;
      constexpr int UPS = F::VPL / F::UNIT_VALS;
      if (tid < 64) {
        const int it = min(tid, NIT - 1);
        const int u = it / TPU, t = it % TPU;
        float xa[PPU], xb[PPU];
        unsigned amax = 0;
        float usum = 0.f;
#pragma unroll
        for (int i = 0; i < PPU; ++i) {
          const int ka = (BITS == 3) ? F::ka(i) : PPU * t + i;
          const int kb = (BITS == 3) ? F::kb(i) : UP + PPU * t + i;
          xa[i] = o_s[u * F::UNIT_VALS + ka];
          xb[i] = o_s[u * F::UNIT_VALS + kb];
          usum += xa[i] + xb[i];
          amax = max(amax, max(__float_as_uint(xa[i]) & 0x7FFFFFFFu, __float_as_uint(xb[i]) & 0x7FFFFFFFu));
        }
        float ssum = usum;
        if constexpr (GL >= 2) { ssum += dpp_f32<0xB1>(ssum); amax = max(amax, dpp_u32<0xB1>(amax)); }
        if constexpr (GL >= 4) { ssum += dpp_f32<0x4E>(ssum); amax = max(amax, dpp_u32<0x4E>(amax)); }
        if constexpr (GL >= 8) { ssum += dpp_f32<0x141>(ssum); amax = max(amax, dpp_u32<0x141>(amax)); }
        int E = 141 - static_cast<int>(amax >> 23);
        E = min(max(E, -100), 110);
        if (tid < NIT) {
          // granule index of image dword (slot sl of the head, pair p): sl * PAIRS + p; of sub-slot j's (sum, factor): 64 + 2 j + {0, 1}
          const int sl = u / UPS, uin = u % UPS;
          gu64_t* dst = gran + rblk * GPH + sl * PAIRS + uin * UP + PPU * t;
#pragma unroll
          for (int i = 0; i < PPU; ++i) {
            int ps;
            if constexpr (BITS == 3) ps = F::pos(i);
            else if constexpr (BITS == 2) { const int pi = PPU * t + i; ps = 2 * (pi < 5 ? pi : pi - 5); }
            else { ps = ((PPU * t + i) & 1) ? 4 : 0; }
            const float sc = exp2i(E - ps);
            h2 v;
            v.x = static_cast<_Float16>(xa[i] * sc);
            v.y = static_cast<_Float16>(xb[i] * sc);
            __hip_atomic_store(dst + i, (1ull << 32) | as_u32(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          if ((it % GL) == 0) {
            gu64_t* ds = gran + rblk * GPH + 64 + 2 * (it / GL);
            __hip_atomic_store(ds, (1ull << 32) | __float_as_uint(ssum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(ds + 1, (1ull << 32) | __float_as_uint(exp2i(24 - E)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
    }
    AO_STAMP(5);
  }

  // ---- 3. every workgroup: gather the group's 4 x 72 granules (one wave), park them in LDS ---------------------------
  if (wave == 0) {
    constexpr int NGR = REP * GPH;                            // 288
    constexpr int PASS = (NGR + 63) / 64;                     // 5 granules per lane
    unsigned val[PASS];
    for (unsigned spins = 0;; ++spins) {
      bool ok = true;
#pragma unroll
      for (int kx = 0; kx < PASS; ++kx) {
        const int idx = min(lane + 64 * kx, NGR - 1);
        const unsigned long long x = __hip_atomic_load(gran + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        val[kx] = static_cast<unsigned>(x);
        ok &= (x >> 32) == 1ull;
      }
      if (__all(ok)) break;
      if (spins > (1u << 18)) __builtin_trap();               // a producer that never ran: fail loudly, never hang the queue
      __builtin_amdgcn_s_sleep(1);
    }
    AO_STAMP(12);
#pragma unroll
    for (int kx = 0; kx < PASS; ++kx) {
      const int idx = lane + 64 * kx;
      if (idx < NGR) {
        const int hq = idx / GPH, r = idx % GPH;
        if (r < 64) img_s[(hq * SPH + r / PAIRS) * SSTR + r % PAIRS] = val[kx];
        else reinterpret_cast<unsigned*>(xs_s)[(hq * 4 + (r - 64) / 2) * 2 + ((r - 64) & 1)] = val[kx];
      }
    }
  }
  __syncthreads();
  AO_STAMP(6);

  // ---- 4. the Wo slice: lane = (row, slot) ---------------------------------------------------------------------------
  h2 xv[PAIRS];
  {
    const u32x4* src = reinterpret_cast<const u32x4*>(img_s + slot * SSTR);
#pragma unroll
    for (int q = 0; q < PAIRS / 4; ++q) {
      const u32x4 v = src[q];
      xv[4 * q + 0] = as_h2(v.x); xv[4 * q + 1] = as_h2(v.y); xv[4 * q + 2] = as_h2(v.z); xv[4 * q + 3] = as_h2(v.w);
    }
  }
  float xs[NSUB], cf[NSUB];
#pragma unroll
  for (int s = 0; s < NSUB; ++s) { const float2 t = xs_s[slot * NSUB + s]; xs[s] = t.x; cf[s] = t.y; }
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    float d[NSUB];
    dot_slot<BITS>(wreg[i], xv, d);
    const float sc = act_to_f32<BF16>(sreg[i]), ze = act_to_f32<BF16>(zreg[i]);
    float part;
    if constexpr (NSUB == 2) {              // both halves of a 64-value slot share one group (group size >= 64)
      const float t = fmaf(cf[1], d[1], cf[0] * d[0]);
      part = fmaf(ze, xs[0] + xs[1], sc * t);
    } else {
      part = fmaf(ze, xs[0], sc * cf[0] * d[0]);
    }
    part += dpp_f32<0xB1>(part);
    part += dpp_f32<0x4E>(part);
    part += dpp_f32<0x141>(part);
    if constexpr (NSLOT == 16) part += dpp_f32<0x140>(part);
    if (slot == 0 && rowof[i] < a.D) {
      const long long fx = __float2ll_rn(part * ACC_SCALE);
      atomicAdd(reinterpret_cast<unsigned long long*>(a.acc + rowof[i]), static_cast<unsigned long long>(fx));
    }
  }
  AO_STAMP(7);
}

bool attn_oproj_supported(int n_heads, int n_kv, int head_dim, int D, int bits, int group_size, int capacity) {
  if (head_dim != 128 || n_kv < 1 || n_heads != 4 * n_kv) return false;
  if (D % 128 != 0 || D / 128 < 4 || (bits != 2 && bits != 3 && bits != 4)) return false;
  if (group_size != 64 && group_size != 128) return false;
  return capacity >= 1;
}

hipError_t launch_attn_oproj(const void* q, const void* k, const void* v, void* kcache, void* vcache, const float* inv_freq,
                             const long long* pos, const uint32_t* qweight, const void* scales, const void* zeros, long long* acc,
                             int n_heads, int n_kv, int head_dim, int capacity, int D, int bits, int log2g, float rope_scale,
                             float sm_scale, int traditional, bool bf16, hipStream_t s) {
  if (!attn_oproj_supported(n_heads, n_kv, head_dim, D, bits, 1 << log2g, capacity)) return hipErrorInvalidValue;
  AttnOprojArgs a{static_cast<const unsigned short*>(q), static_cast<const unsigned short*>(k), static_cast<const unsigned short*>(v),
                  static_cast<unsigned short*>(kcache), static_cast<unsigned short*>(vcache), inv_freq, pos, qweight,
                  static_cast<const unsigned short*>(scales), static_cast<const unsigned short*>(zeros), acc,
                  reinterpret_cast<unsigned long long*>(acc + D), n_heads, n_kv, capacity,
                  D, log2g, rope_scale, sm_scale, traditional};
#ifdef GBX_STAMPS
  a.stamps = stamp_region();
#endif
  dim3 grid(n_kv, D / 128), block(1024);
#define GBX_AO_LAUNCH(B)                                                                          \
  do {                                                                                            \
    if (bf16) hipLaunchKernelGGL((attn_oproj_kernel<B, true>), grid, block, 0, s, a);             \
    else hipLaunchKernelGGL((attn_oproj_kernel<B, false>), grid, block, 0, s, a);                 \
  } while (0)
  if (bits == 2) GBX_AO_LAUNCH(2);
  else if (bits == 3) GBX_AO_LAUNCH(3);
  else GBX_AO_LAUNCH(4);
#undef GBX_AO_LAUNCH
  return hipGetLastError();
}

}  // namespace gbx
