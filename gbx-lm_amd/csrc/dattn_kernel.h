// dattn_kernel: see dattn_impl.h for the role, the mapping and the hand-off form.  One TU per q|k|v bit width (dattn_b{2,3,4}.hip).
#pragma once
#include "dattn_impl.h"
#include "kernels.h"
#include <type_traits>

namespace gbx {

typedef __attribute__((address_space(1))) unsigned long long da_gu64;
typedef __attribute__((address_space(1))) unsigned da_gu32;

// LDS map (bytes).  More than half of a CU's 160 KB is requested so that a CU holds exactly one workgroup (the 256 workgroups then cover the 256 CUs).
constexpr int DA_LDS_IMG1 = 0;                     // phase-1 activation image: Lds<BQ>::bytes(K <= 4096) <= 11.3 KB
constexpr int DA_LDS_RED = 12 * 1024;              // 16 floats: RMSNorm partials
constexpr int DA_LDS_P3 = 12 * 1024 + 256;         // phase 3: 2 KB of raw attention output, then its operand image (<= 5.2 KB)
constexpr int DA_LDS_ATT = 24 * 1024;              // attention scratch: q | k | v words, rotated q / k / v, up to 64 softmax states (37.6 KB)
constexpr int DA_LDS_DUMP = 100 * 1024;          // cooperative sweep, attention workgroups: 16 KB that the L2 prefetch's LDS-DMA overwrites (never read)
#ifdef GBX_STAMPS
constexpr int DA_LDS_BYTES = 156 * 1024;         // diagnostic build: room for the step-0 probe's LDS-DMA stream
#else
constexpr int DA_LDS_BYTES = 116 * 1024;         // (the cooperative sweep's 128 partial states end at 98.1 KB)
#endif
#ifndef DA_SWEEP4_MAX
#define DA_SWEEP4_MAX 160      // contexts up to here are swept by 4 waves (one per SIMD), up to 320 by 8, longer ones by 16
#endif

#ifdef GBX_STAMPS
// diagnostic build: stamps are kept in LDS (behind the product's region) and flushed at the end of the launch -- a global store per
// stamp would sit in the vector-memory queue and lengthen every s_waitcnt vmcnt that follows it
#define DA_STAMP(i) do { if (lane == 0) reinterpret_cast<unsigned long long*>(smem + DA_LDS_BYTES)[wave * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define DA_STAMP_EXTRA 2048
#else
#define DA_STAMP_EXTRA 0
#define DA_STAMP(i) do {} while (0)
#endif

// ---- one item of the activation image (dgemv_impl.h, step 3): PPU pairs of one unit -> pre-shifted fp16 pairs + the sub-slot's
//      {sum, 2^(24 - E)}.  xw / ww: the item's raw 16-bit activations (and norm weights) as packed words, first the ka halves, then kb.
template <int BITS, bool BF16, int PPU>
__device__ __forceinline__ void da_image_item(const unsigned (&xw)[PPU], const unsigned (&ww)[PPU], bool norm, float rstd, int it, int nitems,
                                              h2* xp_lds, float2* xs_lds) {
  using F = Fmt<BITS>;
  using L = Lds<BITS>;
  constexpr int UP = F::UNIT_PAIRS, TPU = UP / PPU, UPS = F::VPL / F::UNIT_VALS, SSTR = L::SSTR;
  constexpr int GL = (BITS == 3) ? 1 : 16 / PPU;
  const int u = it / TPU, t = it % TPU;
  float xa[PPU], xb[PPU];
  unsigned amax = 0;
  float usum = 0.f;
#pragma unroll
  for (int i = 0; i < PPU; ++i) {
    int wa, ha, wb, hb;
    if constexpr (BITS == 3) { wa = F::ka(i) >> 1; ha = F::ka(i) & 1; wb = F::kb(i) >> 1; hb = F::kb(i) & 1; }
    else { wa = i >> 1; ha = i & 1; wb = (PPU >> 1) + (i >> 1); hb = i & 1; }
    float va = ha ? act_hi<BF16>(xw[wa]) : act_lo<BF16>(xw[wa]);
    float vb = hb ? act_hi<BF16>(xw[wb]) : act_lo<BF16>(xw[wb]);
    if (norm) {
      va = round_bf16_or_f16(va * rstd * (ha ? act_hi<BF16>(ww[wa]) : act_lo<BF16>(ww[wa])), BF16);
      vb = round_bf16_or_f16(vb * rstd * (hb ? act_hi<BF16>(ww[wb]) : act_lo<BF16>(ww[wb])), BF16);
    }
    xa[i] = va;
    xb[i] = vb;
    usum += va + vb;
    amax = max(amax, max(__float_as_uint(va) & 0x7FFFFFFFu, __float_as_uint(vb) & 0x7FFFFFFFu));
  }
  float ssum = usum;
  if constexpr (GL >= 2) { ssum += dpp_f32<0xB1>(ssum); amax = max(amax, dpp_u32<0xB1>(amax)); }
  if constexpr (GL >= 4) { ssum += dpp_f32<0x4E>(ssum); amax = max(amax, dpp_u32<0x4E>(amax)); }
  if constexpr (GL >= 8) { ssum += dpp_f32<0x141>(ssum); amax = max(amax, dpp_u32<0x141>(amax)); }
  int E = 141 - static_cast<int>(amax >> 23);
  E = min(max(E, -100), 110);
  if (it < nitems) {
    const int slot = u / UPS, uin = u % UPS;
    unsigned* dst = reinterpret_cast<unsigned*>(xp_lds + static_cast<size_t>(slot) * SSTR + uin * UP + PPU * t);
    unsigned img[PPU];
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
      img[i] = as_u32(v);
    }
    if constexpr (PPU == 2) *reinterpret_cast<u32x2*>(dst) = u32x2{img[0], img[1]};
    else {
#pragma unroll
      for (int i = 0; i < PPU; i += 4) *reinterpret_cast<u32x4*>(dst + i) = u32x4{img[i], img[i + 1], img[i + 2], img[i + 3]};
    }
    if ((it % GL) == 0) xs_lds[it / GL] = make_float2(ssum, exp2i(24 - E));
  }
}

// the item's raw words from a 16-bit vector (global or LDS)
template <int BITS, int PPU>
__device__ __forceinline__ void da_load_item(const unsigned short* base, int it, unsigned (&out)[PPU]) {
  using F = Fmt<BITS>;
  constexpr int UP = F::UNIT_PAIRS, TPU = UP / PPU;
  const int u = it / TPU, t = it % TPU;
  if constexpr (BITS == 3) {
    const u32x4* src = reinterpret_cast<const u32x4*>(base + static_cast<size_t>(u) * F::UNIT_VALS);
#pragma unroll
    for (int q = 0; q < 4; ++q) { const u32x4 v = src[q]; out[4 * q] = v.x; out[4 * q + 1] = v.y; out[4 * q + 2] = v.z; out[4 * q + 3] = v.w; }
  } else {
    static_assert(BITS == 3 || PPU == 2, "two pairs per thread");
    const unsigned short* pa = base + static_cast<size_t>(u) * F::UNIT_VALS + PPU * t;
    out[0] = *reinterpret_cast<const unsigned*>(pa);
    out[1] = *reinterpret_cast<const unsigned*>(pa + UP);
  }
}

// phase 3, one lane = one slot of one o_proj row: s * sum(q x) + z * sum(x) over the slot (dgemv's consume, one row)
template <int BO, bool BF16>
__device__ __forceinline__ float da_o_slot(const u32x4& w, unsigned sc_bits, unsigned ze_bits, const h2* xp_lds, const float2* xs_lds, int slot) {
  using F = Fmt<BO>;
  using L = Lds<BO>;
  constexpr int PAIRS = L::PAIRS, NSUB = F::NSUB;
  h2 xv[PAIRS];
  const u32x4* src = reinterpret_cast<const u32x4*>(xp_lds + static_cast<size_t>(slot) * L::SSTR);
#pragma unroll
  for (int q = 0; q < PAIRS / 4; ++q) {
    const u32x4 v = src[q];
    xv[4 * q + 0] = as_h2(v.x); xv[4 * q + 1] = as_h2(v.y); xv[4 * q + 2] = as_h2(v.z); xv[4 * q + 3] = as_h2(v.w);
  }
  typename SlotWords<BO>::type ws;
  if constexpr (BO == 3) ws = u32x3{w.x, w.y, w.z}; else ws = w;
  float d[NSUB];
  dot_slot<BO>(ws, xv, d);
  const float sc = act_to_f32<BF16>(static_cast<unsigned short>(sc_bits)), ze = act_to_f32<BF16>(static_cast<unsigned short>(ze_bits));
  if constexpr (NSUB == 2) {
    const float2 t0 = xs_lds[slot * 2], t1 = xs_lds[slot * 2 + 1];
    const float t = fmaf(t1.y, d[1], t0.y * d[0]);
    return fmaf(ze, t0.x + t1.x, sc * t);
  } else {
    const float2 t0 = xs_lds[slot];
    return fmaf(ze, t0.x, (sc * t0.y) * d[0]);
  }
}

typedef float da_f32x2 __attribute__((ext_vector_type(2)));
template <bool BF16> __device__ __forceinline__ da_f32x2 da_unpack2(unsigned w) { return da_f32x2{act_lo<BF16>(w), act_hi<BF16>(w)}; }
// q . k over the 16 lanes that share one cached position (8 features each); kf: the lane's 8 features of k as four fp32 pairs
__device__ __forceinline__ float da_dot16(const da_f32x2 (&kf)[4], const da_f32x2 (&q2)[4]) {
  da_f32x2 t = kf[0] * q2[0];
  t = __builtin_elementwise_fma(kf[1], q2[1], t);
  t = __builtin_elementwise_fma(kf[2], q2[2], t);
  t = __builtin_elementwise_fma(kf[3], q2[3], t);
  float s = t.x + t.y;
  s += dpp_f32<0xB1>(s);
  s += dpp_f32<0x4E>(s);
  s += dpp_f32<0x141>(s);
  s += dpp_f32<0x140>(s);
  return s;
}
__device__ __forceinline__ void da_axpy8(const da_f32x2 (&vf)[4], float p, da_f32x2 (&acc2)[4]) {
  const da_f32x2 p2 = da_f32x2{p, p};
#pragma unroll
  for (int c = 0; c < 4; ++c) acc2[c] = __builtin_elementwise_fma(p2, vf[c], acc2[c]);
}
template <bool BF16> __device__ __forceinline__ void da_unpack8(const u32x4& w, da_f32x2 (&o)[4]) {
  o[0] = da_unpack2<BF16>(w.x); o[1] = da_unpack2<BF16>(w.y); o[2] = da_unpack2<BF16>(w.z); o[3] = da_unpack2<BF16>(w.w);
}

typedef int da_v8i __attribute__((ext_vector_type(8)));
typedef float da_v4f __attribute__((ext_vector_type(4)));

// MX = 0: vector-ALU dots (any width).  MX = 2: the q|k|v rows (2 bits, group 128, K = 4096: two 2048-value spans) and the o_proj slice
// (2 bits, group 128, rep = 4: four rows x four groups per operand) on the FP4 x FP8 matrix cores -- dgemv_mx.hip's arithmetic: the
// producers' consume is vector-ALU bound (14 two-row batches per CU: the four wave groups of a SIMD publish 0.33 us apart) and so is
// the o_proj slice (128 rows x 512 columns per CU: 0.64 us of v_and + v_dot2c).
template <int BQ, int P, bool BF16, int MX = 0>
__global__ __launch_bounds__(DA_NT) void dattn_kernel(const DattnArgs a) {
  static_assert(MX == 0 || (MX == 2 && BQ == 2 && P == 1 && BF16), "matrix-core form: 2-bit rows of 4096 values, bf16");
  using F = Fmt<BQ>;
  using L = Lds<BQ>;
  using WT = typename SlotWords<BQ>::type;
  constexpr int NT = DA_NT, NW = DA_NW, R = 2, HD = DA_HD;
  constexpr int PAIRS = L::PAIRS, SSTR = L::SSTR, NSUB = F::NSUB;
  constexpr unsigned KILL = 0x80000000u;
  constexpr int RS_FLAGS = 0x00020000, RS_ALL = 0x7FFFFFFF;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef GBX_STAMPS
  if (lane < 16) reinterpret_cast<unsigned long long*>(smem + DA_LDS_BYTES)[wave * 16 + lane] = 0ull;
#endif
  DA_STAMP(0);
  const int g = static_cast<int>(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u);     // HW_REG_XCC_ID: this XCD = this kv group
  const int r = static_cast<int>(blockIdx.x) >> 3;                                       // member of the group
  const int rep = a.rep, D = a.D;
  da_gu32* gen_p = reinterpret_cast<da_gu32*>(reinterpret_cast<uintptr_t>(a.state)) + g * 16;
  da_gu64* hand1 = reinterpret_cast<da_gu64*>(reinterpret_cast<uintptr_t>(a.hand)) + static_cast<size_t>(g) * DA_HAND_STRIDE;
  da_gu64* hand2 = hand1 + 1024;
  auto rsrc = [](const void* ptr, int bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, bytes, RS_FLAGS); };
  // a hand-off that never arrives: status word for the host, and the accumulator's non-finite marker so that the stream turns NaN in-band
  auto timeout = [&](unsigned code) {
    if (lane == 0) {
      a.state[8 * 16] = code;
      __hip_atomic_fetch_add(reinterpret_cast<da_gu64*>(reinterpret_cast<uintptr_t>(a.acc)) + a.D, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  // one wave polls 64 granules until every tag matches; returns the lane's value
  auto sweep = [&](da_gu64* p, unsigned tag, unsigned code) -> unsigned {
    unsigned long long x;
    for (unsigned spins = 0;;) {
      x = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);           // sc1: past this CU's L1, served by the XCD's L2
      if (__all(static_cast<unsigned>(x >> 32) == tag)) break;
      if (++spins > a.spin_limit) { timeout(code); break; }
      __builtin_amdgcn_s_sleep(1);
    }
    return static_cast<unsigned>(x);
  };
  // the same for two granules per lane (both in flight together)
  auto sweep2 = [&](da_gu64* pa, da_gu64* pb, unsigned tag, unsigned code, unsigned& va, unsigned& vb) {
    unsigned long long x, y;
    for (unsigned spins = 0;;) {
      x = __hip_atomic_load(pa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      y = __hip_atomic_load(pb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__all(static_cast<unsigned>(x >> 32) == tag && static_cast<unsigned>(y >> 32) == tag)) break;
      if (++spins > a.spin_limit) { timeout(code); break; }
      __builtin_amdgcn_s_sleep(1);
    }
    va = static_cast<unsigned>(x);
    vb = static_cast<unsigned>(y);
  };

  if (a.zero_buf) {                                          // (kernel-argument uniform)
    for (int i = static_cast<int>(blockIdx.x) * NT + tid; i < a.zero_n; i += static_cast<int>(gridDim.x) * NT) a.zero_buf[i] = 0ull;
  }
  // Long contexts: ONE CU pulls a head's K / V at ~33 GB/s (0.5 KB per position: 160 positions hide behind phase 1, 650 take 10 us),
  // so beyond coop_min positions (GBX_DATTN_COOP_MIN) the group's PRODUCERS sweep the context in chunks once their rows are published -- every chunk's
  // K / V is read once for all `rep` heads -- and leave one partial softmax state per (producer, head); the head's workgroup only
  // rotates / appends the new token and merges.  (The group's hand-offs stay XCD-local: flags are granules, states are sc0 stores.)
  const int cpos = static_cast<int>(*a.pos);
  const bool coop = cpos > a.coop_min && (rep == 3 || rep == 4);     // (two head pairs x 8 position sets; LDS sized for 4 heads)
  const int cd0 = (lane & 15) * 8, csub = lane >> 4;          // 16 lanes per cached position, 8 features per lane
  // ---------------- o_proj: this workgroup's rows, this group's columns -- requested early, consumed in phase 3 ----------------
  const int obits = a.o_bits;
  const int ovpl_l2 = obits == 2 ? 6 : 5;
  const int KC = rep * HD;                                  // input columns of this group
  const int KO = KC * DA_GROUPS;                            // o_proj input dims
  const int spr = KC >> ovpl_l2;                            // slots per row slice
  const int lpr_l2 = spr <= 8 ? 3 : (spr <= 16 ? 4 : 5);    // lanes per row (power of two >= spr)
  const int rpp = 64 >> lpr_l2;                             // rows per wave and pass
  const int RPW = D / DA_MEMBERS;                           // rows of this workgroup
  const int W4o = (KO >> 3) * obits;                        // bytes per packed o_proj row
  const int slotb = obits == 3 ? 12 : 16;
  const int Go = KO >> a.o_log2g;
  const int oslot = lane & ((1 << lpr_l2) - 1);
  u32x4 ow[DA_OP];
  unsigned osc[DA_OP], oze[DA_OP];
  // matrix-core form: operand lane (nn, kg) holds 8 bytes of (row nn / 4 of the operand's four, group nn % 4): two operands = 8 rows per wave
  const int mnn = lane & 15, mkg = lane >> 4;
  u32x2 owm[2];
  auto issue_o = [&]() {
    if constexpr (MX != 0) {
      const __amdgpu_buffer_rsrc_t od = rsrc(a.o_qw, D * W4o);
      const __amdgpu_buffer_rsrc_t sd = rsrc(a.o_s, RS_ALL), zd = rsrc(a.o_z, RS_ALL);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = r * RPW + wave * 8 + 4 * j + (mnn >> 2);
        owm[j] = __builtin_amdgcn_raw_buffer_load_b64(od, static_cast<unsigned>(row * W4o + (g * 4 + (mnn & 3)) * 32 + mkg * 8), 0, 0);
        const unsigned so = static_cast<unsigned>((row * Go + g * 4 + (mnn & 3)) * 2);
        osc[j] = __builtin_amdgcn_raw_buffer_load_b16(sd, so, 0, 0);
        oze[j] = __builtin_amdgcn_raw_buffer_load_b16(zd, so, 0, 0);
      }
      return;
    }
    const __amdgpu_buffer_rsrc_t od = rsrc(a.o_qw, D * W4o);
    const __amdgpu_buffer_rsrc_t sd = rsrc(a.o_s, RS_ALL), zd = rsrc(a.o_z, RS_ALL);
#pragma unroll
    for (int p = 0; p < DA_OP; ++p) {
      const int rl = (p * NW + wave) * rpp + (lane >> lpr_l2);
      const bool live = rl < RPW && oslot < spr;
      const int row = r * RPW + (live ? rl : 0);
      const unsigned kill = live ? 0u : KILL;
      ow[p] = __builtin_amdgcn_raw_buffer_load_b128(od, static_cast<unsigned>(row * W4o + (g * spr + oslot) * slotb) | kill, 0, 0);
      const unsigned so = static_cast<unsigned>((row * Go + ((g * KC + (oslot << ovpl_l2)) >> a.o_log2g)) * 2) | kill;
      osc[p] = __builtin_amdgcn_raw_buffer_load_b16(sd, so, 0, 0);
      oze[p] = __builtin_amdgcn_raw_buffer_load_b16(zd, so, 0, 0);
    }
  };

  unsigned tag;
  if (r >= rep) {
    if constexpr (MX != 0) {
    // ================= phase 1 on the matrix cores (dgemv_mx.hip's arithmetic; 2 bits, group 128, K = 4096 = two spans) =================
    constexpr int SP = 2, NBLK = SP * 64;
    unsigned char* img = smem + DA_LDS_IMG1;                       // [span][limb][piece][64 lanes][16 B]
    float* bsum = reinterpret_cast<float*>(smem + DA_LDS_IMG1 + SP * 4096);
    unsigned char* scl = reinterpret_cast<unsigned char*>(bsum + NBLK);
    float* red_lds = reinterpret_cast<float*>(smem + DA_LDS_RED);
    const int K = D, W4 = K / 4, G = K >> 7;
    const int nb = (rep + 2) * (HD / 2);
    // batch lb = wave * producers + producer: EVERY producer owns a batch (nb >= producers), so every member of the group publishes at
    // least one granule that the attention workgroups wait for -- the argument the generation advance at the end of the launch rests on
    const int lb = wave * (DA_MEMBERS - rep) + (r - rep);
    const bool has = lb < nb;
    const int qrows = rep * HD;
    const int lr = 2 * (has ? lb : 0);
    const int grow = lr < qrows ? g * qrows + lr
                                : (lr < qrows + HD ? DA_GROUPS * qrows + g * HD + (lr - qrows)
                                                   : DA_GROUPS * qrows + DA_GROUPS * HD + g * HD + (lr - qrows - HD));
    const int gb = grow >> 1;
    const __amdgpu_buffer_rsrc_t qd = rsrc(a.qkv_qw, RS_ALL), szd = rsrc(a.qkv_sz, RS_ALL);
    const unsigned killw = has ? 0u : KILL;
    const unsigned voff_w = mnn * 32 + mkg * 8, voff_s = mnn * 8;
    constexpr int NIT = SP * 512, PW = 8, PT = PW * 64, IPT = NIT / PT;
    const bool pro = wave < PW;
    const bool carried = a.ssq_in != nullptr;
    u32x2 xr[IPT], wr[IPT];
    u32x4 pq[4] = {u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}};
    if (pro) {
      u32x4 acv[IPT][2];
      const __amdgpu_buffer_rsrc_t ad = rsrc(a.x_acc ? static_cast<const void*>(a.x_acc) : static_cast<const void*>(a.x), RS_ALL);
      const unsigned ak = a.x_acc ? 0u : KILL;
#pragma unroll
      for (int j = 0; j < IPT; ++j) {
        const int it = tid + PT * j;
        acv[j][0] = __builtin_amdgcn_raw_buffer_load_b128(ad, static_cast<unsigned>(it * 32) | ak, 0, 0);
        acv[j][1] = __builtin_amdgcn_raw_buffer_load_b128(ad, static_cast<unsigned>(it * 32 + 16) | ak, 0, 0);
      }
      const unsigned apoison = __builtin_amdgcn_raw_buffer_load_b32(ad, static_cast<unsigned>(K * 8) | ak, 0, 0);
#pragma unroll
      for (int j = 0; j < IPT; ++j) {
        const int it = tid + PT * j;
        xr[j] = *reinterpret_cast<const u32x2*>(a.x + 4 * it);
        wr[j] = *reinterpret_cast<const u32x2*>(a.norm_w + 4 * it);
      }
      {
        const __amdgpu_buffer_rsrc_t pd = rsrc(carried ? static_cast<const void*>(a.ssq_in) : static_cast<const void*>(a.x), RS_ALL);
        const unsigned pk = carried ? 0u : KILL;
#pragma unroll
        for (int i = 0; i < 4; ++i) pq[i] = __builtin_amdgcn_raw_buffer_load_b128(pd, static_cast<unsigned>(lane * 16 + 1024 * i) | pk, 0, 0);
      }
      DA_STAMP(1);
      if (a.x_acc) {
        const bool poison = apoison != 0u;
        const bool writer = blockIdx.x == static_cast<unsigned>(rep * DA_GROUPS) && a.x_out != nullptr;
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
          unsigned wv[2];
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const u32x4 av = acv[j][q];
            const float o0 = round_bf16_or_f16(fx32_to_f32(av.x, av.y), BF16);
            const float o1 = round_bf16_or_f16(fx32_to_f32(av.z, av.w), BF16);
            const unsigned xw = q == 0 ? xr[j].x : xr[j].y;
            float x0 = act_lo<BF16>(xw) + o0, x1 = act_hi<BF16>(xw) + o1;
            if (poison) { x0 = __uint_as_float(0x7FC00000u); x1 = x0; }
            wv[q] = static_cast<unsigned>(f32_to_act<BF16>(x0)) | (static_cast<unsigned>(f32_to_act<BF16>(x1)) << 16);
          }
          xr[j] = u32x2{wv[0], wv[1]};
          if (writer) *reinterpret_cast<u32x2*>(a.x_out + 4 * (tid + PT * j)) = xr[j];
        }
      }
      if (!carried) {
        float ssq = 0.f;
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
          const unsigned w2[2] = {xr[j].x, xr[j].y};
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const float lo = act_lo<BF16>(w2[q]), hi = act_hi<BF16>(w2[q]);
            ssq = fmaf(hi, hi, fmaf(lo, lo, ssq));
          }
        }
        ssq = wave_sum(ssq);
        if (lane == 0) red_lds[wave] = ssq;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();                                           // barrier A: x, the norm weight and the statistic's partials have landed
    const unsigned gen = __hip_atomic_load(gen_p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    DA_STAMP(2);
    if (pro) {
      float tot;
      if (carried) {
        float sp = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          sp += (__uint_as_float(pq[i].x) + __uint_as_float(pq[i].y)) + (__uint_as_float(pq[i].z) + __uint_as_float(pq[i].w));
        tot = wave_sum(sp);
      } else {
        tot = 0.f;
#pragma unroll
        for (int i = 0; i < PW; ++i) tot += red_lds[i];
      }
      const float rstd = rsqrtf(tot / static_cast<float>(K) + a.eps);
#pragma unroll
      for (int j = 0; j < IPT; ++j) {
        const int it = tid + PT * j;
        float v[4] = {act_lo<BF16>(xr[j].x), act_hi<BF16>(xr[j].x), act_lo<BF16>(xr[j].y), act_hi<BF16>(xr[j].y)};
        v[0] = round_bf16_or_f16(v[0] * rstd * act_lo<BF16>(wr[j].x), BF16);
        v[1] = round_bf16_or_f16(v[1] * rstd * act_hi<BF16>(wr[j].x), BF16);
        v[2] = round_bf16_or_f16(v[2] * rstd * act_lo<BF16>(wr[j].y), BF16);
        v[3] = round_bf16_or_f16(v[3] * rstd * act_hi<BF16>(wr[j].y), BF16);
        float sum = (v[0] + v[1]) + (v[2] + v[3]);
        unsigned amax = max(max(__float_as_uint(v[0]) & 0x7FFFFFFFu, __float_as_uint(v[1]) & 0x7FFFFFFFu),
                            max(__float_as_uint(v[2]) & 0x7FFFFFFFu, __float_as_uint(v[3]) & 0x7FFFFFFFu));
        sum += dpp_f32<0xB1>(sum); amax = max(amax, dpp_u32<0xB1>(amax));
        sum += dpp_f32<0x4E>(sum); amax = max(amax, dpp_u32<0x4E>(amax));
        sum += dpp_f32<0x141>(sum); amax = max(amax, dpp_u32<0x141>(amax));
        if (amax >= 0x7F800000u) sum = __uint_as_float(0x7FC00000u);     // Inf / NaN in the block: NaN through the zero-point term of every row
        int ea = static_cast<int>(amax >> 23);
        ea = min(max(ea, 20), 240);
        const float up = exp2i(134 - ea);
        const int b = it >> 3, u = it & 7;
        const int h = b >> 6, m = (b & 63) >> 2, kb = b & 3;
        const int ga = 2 * (kb & 1) + (u >> 2);
        unsigned char* dst = img + (h * 4 + (kb >> 1)) * 1024 + (m + 16 * ga) * 16 + 2 * (u & 3);
        const float r0 = v[0] * up, r1 = v[1] * up, r2 = v[2] * up, r3 = v[3] * up;
        const int pe = __builtin_amdgcn_cvt_pk_fp8_f32(r0, r2, 0, false);
        const int po = __builtin_amdgcn_cvt_pk_fp8_f32(r1, r3, 0, false);
        *reinterpret_cast<unsigned short*>(dst) = static_cast<unsigned short>(pe);
        *reinterpret_cast<unsigned short*>(dst + 8) = static_cast<unsigned short>(po);
        const int le = __builtin_amdgcn_cvt_pk_fp8_f32(r0 - __builtin_amdgcn_cvt_f32_fp8(pe, 0), r2 - __builtin_amdgcn_cvt_f32_fp8(pe, 1), 0, false);
        const int lo = __builtin_amdgcn_cvt_pk_fp8_f32(r1 - __builtin_amdgcn_cvt_f32_fp8(po, 0), r3 - __builtin_amdgcn_cvt_f32_fp8(po, 1), 0, false);
        *reinterpret_cast<unsigned short*>(dst + 2048) = static_cast<unsigned short>(le);
        *reinterpret_cast<unsigned short*>(dst + 2048 + 8) = static_cast<unsigned short>(lo);
        if (u == 0) { bsum[h * 64 + m + 16 * kb] = sum; scl[h * 64 + m + 16 * kb] = static_cast<unsigned char>(ea - 7); }
      }
    }
    // this wave's two rows (one packed row per operand: 16 groups of 128 values per span)
    u32x2 w[2][SP], sz[SP];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int h = 0; h < SP; ++h) w[i][h] = __builtin_amdgcn_raw_buffer_load_b64(qd, (voff_w + h * 512) | killw, (grow + i) * W4, 2);
    }
#pragma unroll
    for (int h = 0; h < SP; ++h) sz[h] = __builtin_amdgcn_raw_buffer_load_b64(szd, (voff_s + h * 128) | killw, gb * (G * 8), 0);
    __syncthreads();                                           // barrier B: image written
    DA_STAMP(3);
    tag = gen + 1u;
    if (tag == 0u) tag = 1u;
    if (has) {
      da_v8i areg[SP][2];
      int sa = 0;
      float xsg[SP];
      const bool diag = mkg == (mnn >> 2);
#pragma unroll
      for (int h = 0; h < SP; ++h) {
#pragma unroll
        for (int l = 0; l < 2; ++l) {
          const u32x4 p0 = *reinterpret_cast<const u32x4*>(img + (h * 4 + 0) * 1024 + l * 2048 + lane * 16);
          const u32x4 p1 = *reinterpret_cast<const u32x4*>(img + (h * 4 + 1) * 1024 + l * 2048 + lane * 16);
          areg[h][l][0] = p0.x; areg[h][l][1] = p0.y; areg[h][l][2] = p0.z; areg[h][l][3] = p0.w;
          areg[h][l][4] = p1.x; areg[h][l][5] = p1.y; areg[h][l][6] = p1.z; areg[h][l][7] = p1.w;
        }
        sa |= static_cast<int>(scl[h * 64 + lane]) << (8 * h);
        const float gs = (bsum[h * 64 + mnn] + bsum[h * 64 + mnn + 16]) + (bsum[h * 64 + mnn + 32] + bsum[h * 64 + mnn + 48]);
        xsg[h] = diag ? gs : 0.f;
      }
      float dm[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) dm[q] = (diag && (lane & 3) == q) ? 1.f : 0.f;
      float acc[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float yv = 0.f;
        auto span = [&](auto H) {
          constexpr int h = decltype(H)::value;
          da_v8i bv;
          const unsigned w0 = w[i][h].x, w1 = w[i][h].y;
          bv[0] = w0 & 0x33333333u; bv[1] = (w0 >> 2) & 0x33333333u; bv[2] = w1 & 0x33333333u; bv[3] = (w1 >> 2) & 0x33333333u;
          bv[4] = 0; bv[5] = 0; bv[6] = 0; bv[7] = 0;
          da_v4f z = {0.f, 0.f, 0.f, 0.f};
          z = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(areg[h][0], bv, z, 0, 4, h, sa, 0, 127);
          z = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(areg[h][1], bv, z, 0, 4, h, sa, 0, 127);
          const float dsel = fmaf(z[0], dm[0], fmaf(z[1], dm[1], fmaf(z[2], dm[2], z[3] * dm[3])));
          const unsigned szw = i == 0 ? sz[h].x : sz[h].y;
          const float sc = sz_scale<BF16>(szw), ze = sz_zero<BF16>(szw);
          yv += fmaf(ze, xsg[h], (sc + sc) * dsel);
        };
        span(std::integral_constant<int, 0>{});
        span(std::integral_constant<int, 1>{});
        acc[i] = yv;
      }
      const float v0 = wave_sum2(acc[0], acc[1]);
      const float t0 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v0), 31));
      const float t1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v0), 63));
      const unsigned val = static_cast<unsigned>(f32_to_act<BF16>(t0)) | (static_cast<unsigned>(f32_to_act<BF16>(t1)) << 16);
      if (lane == 0) __hip_atomic_store(hand1 + lb, (static_cast<unsigned long long>(tag) << 32) | val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    DA_STAMP(4);
    } else {
    // =========================================== phase 1: the group's q | k | v rows ===========================================
    h2* xp_lds = reinterpret_cast<h2*>(smem + DA_LDS_IMG1);
    const int K = D;
    const int nslots = K / F::VPL;
    float2* xs_lds = reinterpret_cast<float2*>(smem + DA_LDS_IMG1 + static_cast<size_t>(nslots) * SSTR * 4);
    float* red_lds = reinterpret_cast<float*>(smem + DA_LDS_RED);
    const int W4 = K / 32 * BQ * 4;
    const int G = K >> a.qkv_log2g;
    const int gshift = a.qkv_log2g - 5;
    // this wave's batch: local rows (2 lb, 2 lb + 1) of the group -> rows of the concatenated matrix
    const int nb = (rep + 2) * (HD / 2);
    // batch lb = wave * producers + producer: EVERY producer owns a batch (nb >= producers), so every member of the group publishes at
    // least one granule that the attention workgroups wait for -- the argument the generation advance at the end of the launch rests on
    const int lb = wave * (DA_MEMBERS - rep) + (r - rep);
    const bool has = lb < nb;
    const int qrows = rep * HD;
    const int lr = 2 * (has ? lb : 0);
    const int grow = lr < qrows ? g * qrows + lr
                                : (lr < qrows + HD ? DA_GROUPS * qrows + g * HD + (lr - qrows)
                                                   : DA_GROUPS * qrows + DA_GROUPS * HD + g * HD + (lr - qrows - HD));
    const int gb = grow >> 1;
    const __amdgpu_buffer_rsrc_t qd = rsrc(a.qkv_qw, RS_ALL), szd = rsrc(a.qkv_sz, RS_ALL);
    unsigned voff_w[P], voff_s[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int slot = p * 64 + lane;
      const unsigned dead = (slot < nslots && has) ? 0u : KILL;
      voff_w[p] = static_cast<unsigned>(slot * (F::SLOT_WORDS * 4)) | dead;
      voff_s[p] = static_cast<unsigned>(((slot * NSUB) >> gshift) * (R * 4)) | dead;
    }
    // ---- 1. activations and norm weights first, then the carried statistic
    constexpr int UP = F::UNIT_PAIRS;
    constexpr int PPU = (BQ == 3) ? UP : 2;
    constexpr int TPU = UP / PPU;
    // The prologue runs on waves 0-7: the 16 waves of the workgroup start ~0.35 us apart and everything up to the image barrier
    // used to wait for the LAST wave's activation loads; the early half carries two items per thread, the late half only joins
    // barrier A ("x has landed": the cue to request rows without delaying x) and barrier B (image written).
    constexpr int PW = 8, PT = PW * 64;
    constexpr int IPT = (P * 64 * (F::VPL / 2) + PT * PPU - 1) / (PT * PPU);
    const int nitems = (K / F::UNIT_VALS) * TPU;
    const bool pro = wave < PW;
    unsigned xr[IPT][PPU], wr[IPT][PPU];
    u32x4 pq[4] = {u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}};
    const bool carried = a.ssq_in != nullptr;
    // 3 bits with x_acc (round 6): an image item is a whole 32-value unit, so only K / 32 = 128 threads own items -- and each of them used to
    // pull 256 bytes of fixed-point sums through 16 dependent-latency loads (the fused attention launch behind a fused MLP launch: 14.2 us
    // against 11.5 behind a down launch that hands over a finished stream).  STAGED: all 512 prologue threads form x' for 8 values each (four
    // loads of sums), leave it in LDS and reduce the statistic; behind barrier A the item threads read their 32 values from LDS.
    const bool staged = BQ == 3 && a.x_acc != nullptr;
    unsigned* xtmp = reinterpret_cast<unsigned*>(smem + DA_LDS_ATT);          // [K / 2] words of x' (the attention scratch is idle in a producer workgroup now)
    if constexpr (BQ == 3) {
      if (pro && staged) {
        const int nch = K >> 3;                                                // 8-value chunks
        const bool live = tid < nch;
        const int c = live ? tid : 0;
        const __amdgpu_buffer_rsrc_t ad = rsrc(a.x_acc, RS_ALL);
        u32x4 sv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) sv[q] = __builtin_amdgcn_raw_buffer_load_b128(ad, static_cast<unsigned>((c * 8 + 2 * q) * 8), 0, 0);
        const unsigned apoison = __builtin_amdgcn_raw_buffer_load_b32(ad, static_cast<unsigned>(K * 8), 0, 0);
        const u32x4 xv = *reinterpret_cast<const u32x4*>(a.x + 8 * c);
#pragma unroll
        for (int j = 0; j < IPT; ++j) da_load_item<BQ, PPU>(a.norm_w, min(tid + PT * j, nitems - 1), wr[j]);
        DA_STAMP(1);
        const bool poison = apoison != 0u;
        const unsigned xw[4] = {xv.x, xv.y, xv.z, xv.w};
        unsigned ow[4];
        float ssq = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float o0 = round_bf16_or_f16(fx32_to_f32(sv[q].x, sv[q].y), BF16);
          const float o1 = round_bf16_or_f16(fx32_to_f32(sv[q].z, sv[q].w), BF16);
          float x0 = act_lo<BF16>(xw[q]) + o0, x1 = act_hi<BF16>(xw[q]) + o1;
          if (poison) { x0 = __uint_as_float(0x7FC00000u); x1 = x0; }
          ow[q] = static_cast<unsigned>(f32_to_act<BF16>(x0)) | (static_cast<unsigned>(f32_to_act<BF16>(x1)) << 16);
          const float lo = act_lo<BF16>(ow[q]), hi = act_hi<BF16>(ow[q]);
          ssq += live ? fmaf(hi, hi, lo * lo) : 0.f;
        }
        if (live) {
          *reinterpret_cast<u32x4*>(xtmp + 4 * c) = u32x4{ow[0], ow[1], ow[2], ow[3]};
          if (blockIdx.x == static_cast<unsigned>(rep * DA_GROUPS) && a.x_out != nullptr) *reinterpret_cast<u32x4*>(a.x_out + 8 * c) = u32x4{ow[0], ow[1], ow[2], ow[3]};
        }
        ssq = wave_sum(ssq);
        if (lane == 0) red_lds[wave] = ssq;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    if (pro && !staged) {
      // the previous block's down_proj sums (x_acc launches) first: the longest round trip of the prologue; without x_acc the
      // offsets are out of range and nothing is fetched
      u32x4 acv[IPT][PPU];
      unsigned apoison;
      {
        const __amdgpu_buffer_rsrc_t ad = rsrc(a.x_acc ? static_cast<const void*>(a.x_acc) : static_cast<const void*>(a.x), RS_ALL);
        const unsigned ak = a.x_acc ? 0u : KILL;
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
          const int it = min(tid + PT * j, nitems - 1);
          const int u = it / TPU, t = it % TPU;
#pragma unroll
          for (int q = 0; q < PPU; ++q) {
            int idx;
            if constexpr (BQ == 3) idx = u * F::UNIT_VALS + 2 * q;
            else idx = u * F::UNIT_VALS + PPU * t + (q < PPU / 2 ? 2 * q : UP + 2 * (q - PPU / 2));
            acv[j][q] = __builtin_amdgcn_raw_buffer_load_b128(ad, static_cast<unsigned>(idx * 8) | ak, 0, 0);
          }
        }
        apoison = __builtin_amdgcn_raw_buffer_load_b32(ad, static_cast<unsigned>(K * 8) | ak, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < IPT; ++j) {
        const int it = min(tid + PT * j, nitems - 1);
        da_load_item<BQ, PPU>(a.x, it, xr[j]);
        da_load_item<BQ, PPU>(a.norm_w, it, wr[j]);
      }
      if (a.x_acc) {                                           // x <- round(x + round(acc * 2^-32)); ONE producer workgroup writes it out
        const bool poison = apoison != 0u;
        const bool writer = blockIdx.x == static_cast<unsigned>(rep * DA_GROUPS) && a.x_out != nullptr;
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
          const int it = min(tid + PT * j, nitems - 1);
          const int u = it / TPU, t = it % TPU;
          const bool live = tid + PT * j < nitems;
#pragma unroll
          for (int q = 0; q < PPU; ++q) {
            int idx;
            if constexpr (BQ == 3) idx = u * F::UNIT_VALS + 2 * q;
            else idx = u * F::UNIT_VALS + PPU * t + (q < PPU / 2 ? 2 * q : UP + 2 * (q - PPU / 2));
            const u32x4 av = acv[j][q];
            const float o0 = round_bf16_or_f16(fx32_to_f32(av.x, av.y), BF16);
            const float o1 = round_bf16_or_f16(fx32_to_f32(av.z, av.w), BF16);
            float x0 = act_lo<BF16>(xr[j][q]) + o0, x1 = act_hi<BF16>(xr[j][q]) + o1;
            if (poison) { x0 = __uint_as_float(0x7FC00000u); x1 = x0; }
            const unsigned wv = static_cast<unsigned>(f32_to_act<BF16>(x0)) | (static_cast<unsigned>(f32_to_act<BF16>(x1)) << 16);
            xr[j][q] = wv;
            if (writer && live) *reinterpret_cast<unsigned*>(a.x_out + idx) = wv;
          }
        }
      }
      const __amdgpu_buffer_rsrc_t pd = rsrc(carried ? static_cast<const void*>(a.ssq_in) : static_cast<const void*>(a.x), RS_ALL);
      const unsigned pk = carried ? 0u : KILL;
#pragma unroll
      for (int i = 0; i < 4; ++i) pq[i] = __builtin_amdgcn_raw_buffer_load_b128(pd, static_cast<unsigned>(lane * 16 + 1024 * i) | pk, 0, 0);
      DA_STAMP(1);
      if (!carried) {
        float ssq = 0.f;
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
          const bool live = tid + PT * j < nitems;
#pragma unroll
          for (int q = 0; q < PPU; ++q) {
            const float lo = act_lo<BF16>(xr[j][q]), hi = act_hi<BF16>(xr[j][q]);
            ssq += live ? fmaf(hi, hi, lo * lo) : 0.f;
          }
        }
        ssq = wave_sum(ssq);
        if (lane == 0) red_lds[wave] = ssq;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();                                           // barrier A: x, the norm weight and the statistic's partials have landed
    const unsigned gen = __hip_atomic_load(gen_p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // first needed at the publish
    DA_STAMP(2);
    if (pro) {
      float tot;
      if (carried) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          s += (__uint_as_float(pq[i].x) + __uint_as_float(pq[i].y)) + (__uint_as_float(pq[i].z) + __uint_as_float(pq[i].w));
        tot = wave_sum(s);
      } else {
        tot = 0.f;
#pragma unroll
        for (int i = 0; i < PW; ++i) tot += red_lds[i];
      }
      const float rstd = rsqrtf(tot / static_cast<float>(K) + a.eps);
      if constexpr (BQ == 3) {
        if (staged) {                                            // this thread's unit(s) of x' from LDS: word q = pair q of the unit
#pragma unroll
          for (int j = 0; j < IPT; ++j) {
            const int it = min(tid + PT * j, nitems - 1);
#pragma unroll
            for (int q = 0; q < PPU; ++q) xr[j][q] = xtmp[it * PPU + q];
          }
        }
      }
      // ---- 3. image
#pragma unroll
      for (int j = 0; j < IPT; ++j)
        if ((tid & ~63) + PT * j < nitems) da_image_item<BQ, BF16, PPU>(xr[j], wr[j], true, rstd, tid + PT * j, nitems, xp_lds, xs_lds);
    }
    // ---- 4. this wave's two rows, then the o_proj words behind them
    WT w[R][P];
    unsigned sz[P][R];
    auto request_rows = [&]() {
#pragma unroll
      for (int i = 0; i < R; ++i) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
          if constexpr (BQ == 3) w[i][p] = __builtin_amdgcn_raw_buffer_load_b96(qd, voff_w[p], (grow + i) * W4, 2);
          else w[i][p] = __builtin_amdgcn_raw_buffer_load_b128(qd, voff_w[p], (grow + i) * W4, 2);
        }
      }
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(szd, voff_s[p], gb * (G * R * 4), 0);
        sz[p][0] = v.x; sz[p][1] = v.y;
      }
    };
#ifndef DA_EXP_AFTER
#define DA_EXP_AFTER 0           // experiment: the image-writing waves request their rows BEHIND barrier B (two lane passes: 48-64 KB per CU in front of it)
#endif
    constexpr bool AFTER = DA_EXP_AFTER != 0 && P == 2;
    if (!AFTER || !pro) request_rows();
    __syncthreads();                                           // barrier B: image written
    DA_STAMP(3);
    if (AFTER && pro) request_rows();
    tag = gen + 1u;
    if (tag == 0u) tag = 1u;
    if (has) {
      h2 xv[P][PAIRS];
      float xs[P][NSUB], cf[P][NSUB];
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int slot = p * 64 + lane;
        const bool live = slot < nslots;
        const int sl = min(slot, nslots - 1);
        const u32x4* src = reinterpret_cast<const u32x4*>(xp_lds + static_cast<size_t>(sl) * SSTR);
#pragma unroll
        for (int q = 0; q < PAIRS / 4; ++q) {
          const u32x4 v = src[q];
          xv[p][4 * q + 0] = as_h2(live ? v.x : 0u);
          xv[p][4 * q + 1] = as_h2(live ? v.y : 0u);
          xv[p][4 * q + 2] = as_h2(live ? v.z : 0u);
          xv[p][4 * q + 3] = as_h2(live ? v.w : 0u);
        }
#pragma unroll
        for (int s = 0; s < NSUB; ++s) {
          const float2 t = xs_lds[sl * NSUB + s];
          xs[p][s] = live ? t.x : 0.f;
          cf[p][s] = live ? t.y : 0.f;
        }
      }
      float acc[R] = {0.f, 0.f};
#pragma unroll
      for (int p = 0; p < P; ++p) {
        float d[R][NSUB];
        WT wp[R];
#pragma unroll
        for (int i = 0; i < R; ++i) wp[i] = w[i][p];
        dot_slot_rows<BQ, R>(wp, xv[p], d);
#pragma unroll
        for (int i = 0; i < R; ++i) {
          const float sc = sz_scale<BF16>(sz[p][i]), ze = sz_zero<BF16>(sz[p][i]);
          if constexpr (NSUB == 2) {
            const float t = fmaf(cf[p][1], d[i][1], cf[p][0] * d[i][0]);
            acc[i] = fmaf(sc, t, acc[i]);
            acc[i] = fmaf(ze, xs[p][0] + xs[p][1], acc[i]);
          } else {
            const float s2 = sc * cf[p][0];
            acc[i] = fmaf(s2, d[i][0], acc[i]);
            acc[i] = fmaf(ze, xs[p][0], acc[i]);
          }
        }
      }
      const float v0 = wave_sum2(acc[0], acc[1]);          // row 0 in lane 31, row 1 in lane 63
      const float t0 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v0), 31));
      const float t1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v0), 63));
      const unsigned val = static_cast<unsigned>(f32_to_act<BF16>(t0)) | (static_cast<unsigned>(f32_to_act<BF16>(t1)) << 16);
      // sc0 store: write-through to this XCD's L2, where the group's attention members poll it
      if (lane == 0) __hip_atomic_store(hand1 + lb, (static_cast<unsigned long long>(tag) << 32) | val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    DA_STAMP(4);
    }
    // o_proj words: needed ~4 us from here, requested off the critical path.  (Loads return in order: under the cooperative sweep the
    // chunk's K / V rows and the q poll go first, or they would queue behind these 32 KB per CU.)
    if (!coop) issue_o();
#ifdef GBX_STAMPS
    if (a.probe_src && (a.probe_mode & 16) && !coop) __syncthreads();      // every wave of this workgroup has published: its granule stores are not behind the stream
    if (a.probe_src && (a.probe_mode & 1) && !coop && !((a.probe_mode & 2) && wave < 2)) {
      const __amdgpu_buffer_rsrc_t pd = rsrc(a.probe_src, RS_ALL);
      const int w0 = (a.probe_mode & 2) ? wave - 2 : wave, nwv = (a.probe_mode & 2) ? NW - 2 : NW;
      for (int n = w0; n < a.probe_kb_p; n += nwv)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(pd, (__attribute__((address_space(3))) void*)(smem + 24 * 1024 + n * 1024), 16, lane * 16,
                                                 static_cast<int>((static_cast<size_t>(blockIdx.x) * 128 + n) * 1024), 0, 2);
    }
#endif
    if (coop) {
      // ---------------- cooperative sweep: this producer's chunk of the context, all `rep` heads of the group ----------------
      // A wave owns a PAIR of heads (the K / V unpacking is shared) and one of 8 "position sets"; round u of set s covers positions
      // p0 + (u * 8 + s) * 4 + [0, 4), 16 lanes each.  RB rounds of K / V are requested NOW -- behind phase 1, whose weight stream already
      // saturates this CU's fill path; the rows were pulled into this XCD's L2 by the group's idle attention workgroups meanwhile -- and
      // re-requested as a ring; the softmax is two-pass per block of RB rounds (one rescale per block).
      constexpr int RB = 4;
      constexpr float NEG = -1e30f;
      const int pi = r - rep, np = DA_MEMBERS - rep;
      const int CH = (((cpos + np - 1) / np) + 3) & ~3;
      const int p0 = pi * CH, p1 = min(cpos, p0 + CH);
      constexpr int nps = 8, ppr = nps * 4, nst = nps * 4;      // position sets; positions per round; states per head
      const int cpair = wave >> 3, cps = wave & 7;              // this wave's head pair and set
      const int ja = 2 * cpair, jb = min(ja + 1, rep - 1);      // (three heads: the second pair computes its one head twice)
      const int nr = p1 > p0 ? (p1 - p0 + ppr - 1) / ppr : 0;   // rounds (workgroup-uniform)
      float* q_s = reinterpret_cast<float*>(smem + DA_LDS_ATT);          // [4][128] rotated, scaled q of the group's heads
      float* gm = q_s + 4 * HD;                                          // [rep * nst <= 128] running maxima
      float* gl = gm + 128;                                              // denominators
      float* gw = gl + 128;                                              // merge weights
      float (*gacc)[HD + 8] = reinterpret_cast<float (*)[HD + 8]>(gw + 128);       // [128][136]
      const __amdgpu_buffer_rsrc_t kd = rsrc(a.kcache + static_cast<size_t>(g) * a.capacity * HD, a.capacity * HD * 2);
      const __amdgpu_buffer_rsrc_t vd = rsrc(a.vcache + static_cast<size_t>(g) * a.capacity * HD, a.capacity * HD * 2);
      const int tl0 = p0 + cps * 4 + csub;                      // this lane's position in round 0
      u32x4 kc[RB], vc[RB];
      auto issue_ring = [&]() {
#pragma unroll
        for (int u = 0; u < RB; ++u) {
          const int t = tl0 + u * ppr;
          const unsigned off = static_cast<unsigned>(t * (HD * 2) + cd0 * 2) | (t < p1 ? 0u : KILL);
          kc[u] = __builtin_amdgcn_raw_buffer_load_b128(kd, off, 0, 0);
          vc[u] = __builtin_amdgcn_raw_buffer_load_b128(vd, off, 0, 0);
        }
      };
      // The LAST `rep` waves fetch the rotated q (one head each).  Loads return in order, so a polling wave asks for nothing before its poll -- its
      // own rows would hold the first poll back by their latency; the last waves own the position sets that run out of positions first.
      const int qh = wave - (NW - rep);
      if (qh < 0) { issue_ring(); issue_o(); }
      if (qh >= 0) {                                            // head qh of the group: its rotated q pairs
        const int i = lane;
        const int i1 = a.traditional ? 2 * i : i;
        const int i2 = a.traditional ? 2 * i + 1 : i + HD / 2;
        // (the head's own workgroup has rotated q 0.5 us after the last q row is out -- this workgroup's poll starts behind its own rows --
        //  and publishes the rotated pairs as 64 more granules)
        const unsigned rw = sweep(hand1 + DA_HAND_QROT + qh * (HD / 2) + lane, tag, 4u);
        const unsigned short r1 = static_cast<unsigned short>(rw & 0xFFFFu), r2 = static_cast<unsigned short>(rw >> 16);
        q_s[qh * HD + i1] = act_to_f32<BF16>(r1) * a.sm_scale;
        q_s[qh * HD + i2] = act_to_f32<BF16>(r2) * a.sm_scale;
        issue_ring();
        issue_o();
      }
      __syncthreads();
      DA_STAMP(12);
      // position sets that hold a cached position at all (a short chunk leaves whole waves without work: they skip the sweep, and the
      // merge below never reads their states)
      const int nvs = nr > 1 ? nps : (nr == 1 ? (p1 - p0 + 3) >> 2 : 0);
      if (cps < nvs) {
        da_f32x2 qa[4], qb[4], acca[4], accb[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          qa[c] = *reinterpret_cast<const da_f32x2*>(q_s + ja * HD + cd0 + 2 * c);
          qb[c] = *reinterpret_cast<const da_f32x2*>(q_s + jb * HD + cd0 + 2 * c);
          acca[c] = da_f32x2{0.f, 0.f};
          accb[c] = da_f32x2{0.f, 0.f};
        }
        float ma = NEG, la = 0.f, mb = NEG, lb = 0.f;
        for (int ub = 0; ub < nr; ub += RB) {
          float sa[RB], sb[RB];
#pragma unroll
          for (int u = 0; u < RB; ++u) {
            sa[u] = NEG;
            sb[u] = NEG;
            if (ub + u < nr) {                                  // (workgroup-uniform)
              const int t = tl0 + (ub + u) * ppr;
              da_f32x2 kf[4];
              da_unpack8<BF16>(kc[u], kf);
              const float s0 = da_dot16(kf, qa), s1 = da_dot16(kf, qb);
              if (t < p1) { sa[u] = s0; sb[u] = s1; }
              if (ub + u + RB < nr) {                           // this slot's next round
                const int tn = t + RB * ppr;
                kc[u] = __builtin_amdgcn_raw_buffer_load_b128(kd, static_cast<unsigned>(tn * (HD * 2) + cd0 * 2) | (tn < p1 ? 0u : KILL), 0, 0);
              }
            }
          }
          float na = ma, nb = mb;
#pragma unroll
          for (int u = 0; u < RB; ++u) { na = fmaxf(na, sa[u]); nb = fmaxf(nb, sb[u]); }
          if (ub > 0) {                                         // one rescale per block (nothing to rescale in the first)
            const float ca = __expf(ma - na), cb = __expf(mb - nb);
            la *= ca;
            lb *= cb;
#pragma unroll
            for (int c = 0; c < 4; ++c) { acca[c] = acca[c] * da_f32x2{ca, ca}; accb[c] = accb[c] * da_f32x2{cb, cb}; }
          }
          ma = na;
          mb = nb;
#pragma unroll
          for (int u = 0; u < RB; ++u) {
            if (ub + u < nr) {
              const float pa = sa[u] > 0.5f * NEG ? __expf(sa[u] - ma) : 0.f;
              const float pb = sb[u] > 0.5f * NEG ? __expf(sb[u] - mb) : 0.f;
              da_f32x2 vf[4];
              da_unpack8<BF16>(vc[u], vf);
              la += pa;
              lb += pb;
              da_axpy8(vf, pa, acca);
              da_axpy8(vf, pb, accb);
              if (ub + u + RB < nr) {
                const int tn = tl0 + (ub + u + RB) * ppr;
                vc[u] = __builtin_amdgcn_raw_buffer_load_b128(vd, static_cast<unsigned>(tn * (HD * 2) + cd0 * 2) | (tn < p1 ? 0u : KILL), 0, 0);
              }
            }
          }
        }
        const int sxa = ja * nst + cps * 4 + csub, sxb = jb * nst + cps * 4 + csub;
        if ((lane & 15) == 0) { gm[sxa] = ma; gl[sxa] = la; }
#pragma unroll
        for (int c = 0; c < 4; ++c) *reinterpret_cast<da_f32x2*>(&gacc[sxa][cd0 + 2 * c]) = acca[c];
        if (jb != ja) {
          if ((lane & 15) == 0) { gm[sxb] = mb; gl[sxb] = lb; }
#pragma unroll
          for (int c = 0; c < 4; ++c) *reinterpret_cast<da_f32x2*>(&gacc[sxb][cd0 + 2 * c]) = accb[c];
        }
      }
      __syncthreads();
      DA_STAMP(13);
      // merge the head's nvs * 4 states: wave j weighs them (lane = state) and sends {max, denominator}; then one (head, feature) per thread
      da_gu64* pp0 = reinterpret_cast<da_gu64*>(reinterpret_cast<uintptr_t>(a.part)) + static_cast<size_t>(g * DA_MEMBERS + pi) * 4 * DA_PART_STRIDE;
      const unsigned long long tg = static_cast<unsigned long long>(tag) << 32;
      if (wave < rep) {
        const bool live = lane < nvs * 4;
        const float mv = live ? gm[wave * nst + lane] : NEG;
        const float M = key2f(wave_max_u32(f2key(mv)));
        const float wgt = live ? __expf(mv - M) : 0.f;
        const float den = wave_sum(live ? wgt * gl[wave * nst + lane] : 0.f);
        if (live) gw[wave * nst + lane] = wgt;
        if (lane < 2) __hip_atomic_store(pp0 + wave * DA_PART_STRIDE + lane, tg | __float_as_uint(lane ? den : M), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      __syncthreads();
      DA_STAMP(15);
      {
        const int j = tid >> 7, f = tid & 127;
        if (j < rep) {
          const int s0 = j * nst;
          float num = 0.f;
          for (int c = 0; c < nvs * 4; c += 4) {
            float tw[4], ta[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { tw[i] = gw[s0 + c + i]; ta[i] = gacc[s0 + c + i][f]; }
#pragma unroll
            for (int i = 0; i < 4; ++i) num = fmaf(tw[i], ta[i], num);
          }
          // every value leaves as its own {tag, value} granule (sc0: stays in this XCD's L2): no flag, no ordering between the stores
          __hip_atomic_store(pp0 + j * DA_PART_STRIDE + 2 + f, tg | __float_as_uint(num), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
      DA_STAMP(14);
    }
  } else {
    // =========================================== phase 2: attention of head g * rep + r ===========================================
    unsigned* raw32 = reinterpret_cast<unsigned*>(smem + DA_LDS_ATT);                    // q | k | v as they arrive: 3 x 64 words
    float* q_s = reinterpret_cast<float*>(smem + DA_LDS_ATT + 768);
    float* kn_s = q_s + HD;
    float* vn_s = kn_s + HD;
    float* gm = vn_s + HD;
    float* gl = gm + 64;
    float (*gacc)[HD + 8] = reinterpret_cast<float (*)[HD + 8]>(gl + 64);
    constexpr float NEG = -1e30f;
    // 16 lanes per position, 4 positions per wave and round.  UB0 rounds of K / V live in registers beside the early o_proj words: 10 where
    // the build fits 128 registers, 8 in the 3-bit and the fp16 kernels -- with 10 the compiler parked an o_proj word quad in SCRATCH
    // (a wait for the load right at its issue, 5 MB of scratch writes per launch: profiles/r06_pmc_dattn.txt)
    constexpr int LP = HD / 8, PPW = 64 / LP, UB0 = (BQ == 3 || !BF16) ? 8 : 10;
    const int h = g * rep + r;
    const int d0 = (lane % LP) * 8, sub = lane / LP;
    // The sweep is vector-ALU work once K / V sit in registers, and every sweeping wave pays a fixed cost (q, state hand-over), so
    // as FEW waves sweep as can hold the context in UB0 rounds of registers: 4 (one per SIMD) up to 160 positions, 8 up to 320,
    // 16 beyond (then an online loop takes what the registers do not hold).  Round u of sweeping wave w covers positions
    // (u * nws + w) * 4 + sub.  Everything is requested NOW, while the producers are busy (~4 us before q arrives).
    const int pos = static_cast<int>(*a.pos);
    const int nws = pos <= (DA_SWEEP4_MAX < UB0 * 4 * PPW ? DA_SWEEP4_MAX : UB0 * 4 * PPW) ? 4 : (pos <= UB0 * 8 * PPW ? 8 : 16);
    const int ppr = nws * PPW;
    const bool sweeper = wave < nws && !coop;                  // (cooperative sweep: the producers own the cached positions)
    const __amdgpu_buffer_rsrc_t kd = rsrc(a.kcache + static_cast<size_t>(g) * a.capacity * HD, a.capacity * HD * 2);
    const __amdgpu_buffer_rsrc_t vd = rsrc(a.vcache + static_cast<size_t>(g) * a.capacity * HD, a.capacity * HD * 2);
    const unsigned short* kbase = a.kcache + static_cast<size_t>(g) * a.capacity * HD;
    const unsigned short* vbase = a.vcache + static_cast<size_t>(g) * a.capacity * HD;
    u32x4 k0[UB0], v0[UB0];
#pragma unroll
    for (int u = 0; u < UB0; ++u) {
      const int t = (u * nws + wave) * PPW + sub;
      const unsigned off = static_cast<unsigned>(t * (HD * 2) + d0 * 2) | (t < pos ? 0u : KILL);
      k0[u] = u32x4{0, 0, 0, 0};
      v0[u] = u32x4{0, 0, 0, 0};
      if (sweeper && u * ppr < pos) {                          // (wave-uniform) rounds wholly beyond the context issue nothing
        k0[u] = __builtin_amdgcn_raw_buffer_load_b128(kd, off, 0, 0);
        v0[u] = __builtin_amdgcn_raw_buffer_load_b128(vd, off, 0, 0);
      }
    }
    if (coop) {
      // Cooperative sweep: this workgroup is idle until q arrives, and its CU's fill path is free while the producers' is saturated by
      // the weight stream -- so the group's `rep` attention workgroups pull the cached K / V rows (the first 1024 positions) into the
      // XCD's L2 now, and the producers read their chunks from there once phase 1 is over.  LDS-DMA into a dump area: no registers, and
      // nothing for the compiler to discard.  4-row piece n goes to workgroup n % rep, wave (n / rep) % 16.
      const int npc = (min(pos, 1024) + 3) >> 2;
      for (int n = r + rep * wave; n < npc; n += rep * NW) {
        const int t = min(4 * n + csub, pos - 1);
        const unsigned off = static_cast<unsigned>(t * (HD * 2) + cd0 * 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(kd, (__attribute__((address_space(3))) void*)(smem + DA_LDS_DUMP + wave * 1024), 16, off, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(vd, (__attribute__((address_space(3))) void*)(smem + DA_LDS_DUMP + wave * 1024), 16, off, 0, 0, 0);
      }
    }
    issue_o();
#ifdef GBX_STAMPS
    if (a.probe_src && (a.probe_mode & 8) && !coop) {          // attention workgroups, early: behind the K / V prefetch and the o words
      const __amdgpu_buffer_rsrc_t pd = rsrc(a.probe_src, RS_ALL);
      for (int n = wave; n < a.probe_kb_a0; n += NW)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(pd, (__attribute__((address_space(3))) void*)(smem + 64 * 1024 + n * 1024), 16, lane * 16,
                                                 static_cast<int>((static_cast<size_t>(blockIdx.x) * 128 + n) * 1024), 0, 2);
    }
#endif
    const unsigned gen = __hip_atomic_load(gen_p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float inv_f = a.inv_freq[lane];                      // RoPE frequency of this lane's pair (waves 0 and 1 rotate q and k)
    // the rotation depends on the position only: ready before q / k arrive
    const float rope_rev = static_cast<float>(pos) * a.rope_scale * inv_f * 0.15915494309189535f;
    const float rope_fr = rope_rev - floorf(rope_rev);
    const float sn = __builtin_amdgcn_sinf(rope_fr), cs = __builtin_amdgcn_cosf(rope_fr);
    DA_STAMP(1);
    tag = gen + 1u;
    if (tag == 0u) tag = 1u;
    // ---- hand-off 1: this head's q (64 granules), the group's k and v -- polled by the threads that rotate / store them: wave 0
    //      the q pairs, wave 1 the k pairs (element e lives in granule e / 2, half e % 2), wave 2 two values of v per lane
    if (wave < 2) {
      const bool isk = wave == 1;
      const int i = lane;
      const int i1 = a.traditional ? 2 * i : i;
      const int i2 = a.traditional ? 2 * i + 1 : i + HD / 2;
      da_gu64* base = hand1 + (isk ? rep : r) * (HD / 2);
      unsigned w1, w2;
      sweep2(base + (i1 >> 1), base + (i2 >> 1), tag, 1u, w1, w2);
      DA_STAMP(5);
      const float x1 = act_to_f32<BF16>(static_cast<unsigned short>((i1 & 1) ? (w1 >> 16) : (w1 & 0xFFFFu)));
      const float x2 = act_to_f32<BF16>(static_cast<unsigned short>((i2 & 1) ? (w2 >> 16) : (w2 & 0xFFFFu)));
      const unsigned short r1 = f32_to_act<BF16>(x1 * cs - x2 * sn), r2 = f32_to_act<BF16>(x1 * sn + x2 * cs);
      if (isk) {
        kn_s[i1] = act_to_f32<BF16>(r1);
        kn_s[i2] = act_to_f32<BF16>(r2);
        if (r == 0) {
          unsigned short* dst = a.kcache + (static_cast<size_t>(g) * a.capacity + pos) * HD;
          dst[i1] = r1;
          dst[i2] = r2;
        }
      } else {
        q_s[i1] = act_to_f32<BF16>(r1) * a.sm_scale;
        q_s[i2] = act_to_f32<BF16>(r2) * a.sm_scale;
        if (coop)                                               // the group's producers sweep with it
          __hip_atomic_store(hand1 + DA_HAND_QROT + r * (HD / 2) + lane, (static_cast<unsigned long long>(tag) << 32) | r1 | (static_cast<unsigned>(r2) << 16),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    } else if (wave == 2) {
      const unsigned vw = sweep(hand1 + (rep + 1) * (HD / 2) + lane, tag, 1u);
      vn_s[2 * lane] = act_lo<BF16>(vw);
      vn_s[2 * lane + 1] = act_hi<BF16>(vw);
      if (r == 0) *reinterpret_cast<unsigned*>(a.vcache + (static_cast<size_t>(g) * a.capacity + pos) * HD + 2 * lane) = vw;
    }
    __syncthreads();
    if (coop) {
      // ---- cooperative sweep: this workgroup owns the new token and the merge of the producers' partial states of its head ----
      // thread (f = tid % 128, pg = tid / 128) polls feature f of producers pg * 4 + [0, 4) -- 12 granules in flight -- and merges them;
      // the eight partial results meet in LDS.
      const int np = DA_MEMBERS - rep;
      const float snew = wave_sum(fmaf(q_s[2 * lane], kn_s[2 * lane], q_s[2 * lane + 1] * kn_s[2 * lane + 1]));     // q . k of the new token
      const int f = tid & (HD - 1), pg = tid >> 7;
      da_gu64* pb = reinterpret_cast<da_gu64*>(reinterpret_cast<uintptr_t>(a.part)) + (static_cast<size_t>(g * DA_MEMBERS) * 4 + r) * DA_PART_STRIDE;
      // lane l also polls ONE of the 8 {max, denominator} granules of its four producers (l % 4: producer, (l / 4) % 2: which); they are
      // wave-uniform per producer and move to scalar registers once the tags match
      da_gu64* pmd = pb + static_cast<size_t>(min(pg * 4 + (lane & 3), np - 1)) * (4 * DA_PART_STRIDE) + ((lane >> 2) & 1);
      unsigned long long gq[4], gs;
      for (unsigned spins = 0;;) {
        gs = __hip_atomic_load(pmd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          gq[i] = __hip_atomic_load(pb + static_cast<size_t>(min(pg * 4 + i, np - 1)) * (4 * DA_PART_STRIDE) + 2 + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool ok = static_cast<unsigned>(gs >> 32) == tag;
#pragma unroll
        for (int i = 0; i < 4; ++i) ok = ok && static_cast<unsigned>(gq[i] >> 32) == tag;
        if (__all(ok)) break;
        if (++spins > a.spin_limit) { timeout(5u); break; }
        __builtin_amdgcn_s_sleep(2);
      }
      DA_STAMP(8);
      {
        float pm[4], pd[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          pm[i] = __uint_as_float(__builtin_amdgcn_readlane(static_cast<unsigned>(gs), i));
          pd[i] = __uint_as_float(__builtin_amdgcn_readlane(static_cast<unsigned>(gs), 4 + i));
        }
        float M = NEG;
#pragma unroll
        for (int i = 0; i < 4; ++i) if (pg * 4 + i < np) M = fmaxf(M, pm[i]);
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float wgt = pg * 4 + i < np ? __expf(pm[i] - M) : 0.f;
          num = fmaf(wgt, __uint_as_float(static_cast<unsigned>(gq[i])), num);
          den = fmaf(wgt, pd[i], den);
        }
        if (f == 0) { gm[pg] = M; gl[pg] = den; }
        gacc[pg][f] = num;
      }
      __syncthreads();
      if (tid < HD) {
        float M = snew;
#pragma unroll
        for (int i = 0; i < 8; ++i) M = fmaxf(M, gm[i]);
        const float wn = __expf(snew - M);
        float num = wn * vn_s[tid], den = wn;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float wgt = __expf(gm[i] - M);
          num = fmaf(wgt, gacc[i][tid], num);
          den = fmaf(wgt, gl[i], den);
        }
        const unsigned ob = f32_to_act<BF16>(num / den);
        if (a.attn_out) a.attn_out[static_cast<size_t>(h) * HD + tid] = static_cast<unsigned short>(ob);
        const unsigned nb2 = dpp_u32<0xB1>(ob);
        if ((lane & 1) == 0)
          __hip_atomic_store(hand2 + r * (HD / 2) + (tid >> 1), (static_cast<unsigned long long>(tag) << 32) | ob | (nb2 << 16),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    } else {
    // ---- cached positions [0, pos): 16 lanes per position, online softmax per lane group (attn_decode_kernel, 16-wave form)
    // Packed fp32 arithmetic (v_pk_fma_f32: two lanes of work per issue slot -- this sweep is vector-ALU bound once the K / V rows
    // sit in registers) and a two-pass softmax over the prefetched rounds: all scores first, ONE running-maximum update, then the
    // exponentials and the value sums without a per-position rescale.
    DA_STAMP(12);
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    auto unpack2 = [](unsigned w) { return f32x2{act_lo<BF16>(w), act_hi<BF16>(w)}; };
    f32x2 q2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) q2[j] = f32x2{q_s[d0 + 2 * j], q_s[d0 + 2 * j + 1]};
    float m = NEG, l = 0.f;
    f32x2 acc2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc2[j] = f32x2{0.f, 0.f};
    auto score = [&](const u32x4& kq) {
      f32x2 t = unpack2(kq.x) * q2[0];
      t = __builtin_elementwise_fma(unpack2(kq.y), q2[1], t);
      t = __builtin_elementwise_fma(unpack2(kq.z), q2[2], t);
      t = __builtin_elementwise_fma(unpack2(kq.w), q2[3], t);
      float s = t.x + t.y;
      s += dpp_f32<0xB1>(s);
      s += dpp_f32<0x4E>(s);
      s += dpp_f32<0x141>(s);
      s += dpp_f32<0x140>(s);
      return s;
    };
    auto add_values = [&](const u32x4& vq, float p) {
      const f32x2 p2 = f32x2{p, p};
      acc2[0] = __builtin_elementwise_fma(p2, unpack2(vq.x), acc2[0]);
      acc2[1] = __builtin_elementwise_fma(p2, unpack2(vq.y), acc2[1]);
      acc2[2] = __builtin_elementwise_fma(p2, unpack2(vq.z), acc2[2]);
      acc2[3] = __builtin_elementwise_fma(p2, unpack2(vq.w), acc2[3]);
    };
    auto rescale = [&](float corr) {
      const f32x2 c2 = f32x2{corr, corr};
#pragma unroll
      for (int j = 0; j < 4; ++j) acc2[j] = acc2[j] * c2;
    };
#ifdef GBX_STAMPS     // diagnostic build only: when has every prefetched K / V row landed?
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DA_STAMP(15);
#endif
    if (sweeper) {
      {
        // Two straight-line halves of UB0 / 2 rounds each (the second only when the context reaches it): inside a half the rounds
        // are independent chains the scheduler interleaves -- a branch per round made every round its own basic block and the
        // single wave of a SIMD waited out each dot -> DPP -> max chain in turn.
        constexpr int HB = UB0 / 2;
        float sc[UB0];
        const bool second = HB * ppr < pos;                                                   // (wave-uniform)
        auto scores = [&](int u0) {
#pragma unroll
          for (int u = u0; u < u0 + HB; ++u) {
            const float s = score(k0[u]);
            sc[u] = ((u * nws + wave) * PPW + sub < pos) ? s : NEG;
          }
#pragma unroll
          for (int u = u0; u < u0 + HB; ++u) m = fmaxf(m, sc[u]);
        };
        auto values = [&](int u0) {
          float p[HB];
#pragma unroll
          for (int u = 0; u < HB; ++u) p[u] = sc[u0 + u] > 0.5f * NEG ? __expf(sc[u0 + u] - m) : 0.f;
#pragma unroll
          for (int u = 0; u < HB; ++u) { l += p[u]; add_values(v0[u0 + u], p[u]); }
        };
        scores(0);
        if (second) scores(HB);
        DA_STAMP(13);
        values(0);
        if (second) values(HB);
      }
      DA_STAMP(14);
      auto consume = [&](const u32x4& kq, const u32x4& vq, bool valid) {                     // online form: positions beyond the prefetch
        const float s = score(kq);
        const float mn = valid ? fmaxf(m, s) : m;
        const float corr = __expf(m - mn);
        const float p = valid ? __expf(s - mn) : 0.f;
        l = l * corr + p;
        rescale(corr);
        add_values(vq, p);
        m = mn;
      };
      constexpr int UB = 2;
      for (int t0 = UB0 * ppr + wave * PPW; t0 < pos; t0 += ppr * UB) {
        u32x4 kv4[UB], vv4[UB];
        bool valid[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          const int t = t0 + u * ppr + sub;
          valid[u] = t < pos;
          const int tc = valid[u] ? t : pos - 1;
          kv4[u] = *reinterpret_cast<const u32x4*>(kbase + static_cast<size_t>(tc) * HD + d0);
          vv4[u] = *reinterpret_cast<const u32x4*>(vbase + static_cast<size_t>(tc) * HD + d0);
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) consume(kv4[u], vv4[u], valid[u]);
      }
      if (wave == 0 && sub == 0) {                             // the new token itself (its k, v are in LDS)
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) s = fmaf(q2[j].x, kn_s[d0 + 2 * j], fmaf(q2[j].y, kn_s[d0 + 2 * j + 1], s));
        s += dpp_f32<0xB1>(s);
        s += dpp_f32<0x4E>(s);
        s += dpp_f32<0x141>(s);
        s += dpp_f32<0x140>(s);
        const float mn = fmaxf(m, s);
        const float corr = __expf(m - mn), p = __expf(s - mn);
        l = l * corr + p;
        rescale(corr);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc2[j] = __builtin_elementwise_fma(f32x2{p, p}, f32x2{vn_s[d0 + 2 * j], vn_s[d0 + 2 * j + 1]}, acc2[j]);
        m = mn;
      }
      // one softmax state per (sweeping wave, lane group)
      const int sidx = wave * PPW + sub;
      if ((lane % LP) == 0) { gm[sidx] = m; gl[sidx] = l; }
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x2*>(&gacc[sidx][d0 + 2 * j]) = acc2[j];
    }
    DA_STAMP(8);
    __syncthreads();
    if (tid < HD) {
      const int ns = nws * PPW;                              // 16, 32 or 64 states: chunks of 16, every chunk's LDS reads in flight together
      float M = NEG;
      for (int c = 0; c < ns; c += 16) {
        float t[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) t[i] = gm[c + i];
#pragma unroll
        for (int i = 0; i < 16; ++i) M = fmaxf(M, t[i]);
      }
      float num = 0.f, den = 0.f;
      for (int c = 0; c < ns; c += 16) {
        float tm[16], tl[16], ta[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) { tm[i] = gm[c + i]; tl[i] = gl[c + i]; ta[i] = gacc[c + i][tid]; }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float wgt = __expf(tm[i] - M);
          num = fmaf(wgt, ta[i], num);
          den = fmaf(wgt, tl[i], den);
        }
      }
      const unsigned ob = f32_to_act<BF16>(num / den);
      if (a.attn_out) a.attn_out[static_cast<size_t>(h) * HD + tid] = static_cast<unsigned short>(ob);
      const unsigned nb2 = dpp_u32<0xB1>(ob);                // the odd neighbour's value on the even lane
      if ((lane & 1) == 0)
        __hip_atomic_store(hand2 + r * (HD / 2) + (tid >> 1), (static_cast<unsigned long long>(tag) << 32) | ob | (nb2 << 16),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    }
    DA_STAMP(9);
#ifdef GBX_STAMPS
    if (a.probe_src && (a.probe_mode & 4) && !coop) {          // attention workgroups, behind their output granules
      const __amdgpu_buffer_rsrc_t pd = rsrc(a.probe_src, RS_ALL);
      for (int n = wave; n < a.probe_kb_a; n += NW)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(pd, (__attribute__((address_space(3))) void*)(smem + 64 * 1024 + (n % 90) * 1024), 16, lane * 16,
                                                 static_cast<int>((static_cast<size_t>(blockIdx.x) * 128 + 32 + n) * 1024), 0, 2);
    }
#endif
  }

  // =========================================== phase 3: o_proj over this group's columns ===========================================
  unsigned* raw3 = reinterpret_cast<unsigned*>(smem + DA_LDS_P3);
  h2* xp3 = reinterpret_cast<h2*>(smem + DA_LDS_P3 + 2048);
  const unsigned short* raw3h = reinterpret_cast<const unsigned short*>(raw3);
  auto phase3 = [&](auto BO_) {
    constexpr int BO = decltype(BO_)::value;
    using FO = Fmt<BO>;
    using LO = Lds<BO>;
    constexpr int PPU3 = (BO == 3) ? FO::UNIT_PAIRS : 2;
    constexpr int TPU3 = FO::UNIT_PAIRS / PPU3;
    float2* xs3 = reinterpret_cast<float2*>(reinterpret_cast<unsigned char*>(xp3) + static_cast<size_t>(spr) * LO::SSTR * 4);
    const int nitems = (KC / FO::UNIT_VALS) * TPU3;
    if constexpr (BO == 3) {
      // 3 bits: an image item is a whole 32-value unit -- the attention output goes through LDS as raw values first
      if (wave < rep) raw3[wave * 64 + lane] = sweep(hand2 + wave * (HD / 2) + lane, tag, 2u);
      __syncthreads();
      DA_STAMP(10);
      if ((tid & ~63) < nitems) {
        unsigned xw[PPU3];
        da_load_item<BO, PPU3>(raw3h, min(tid, nitems - 1), xw);
        da_image_item<BO, BF16, PPU3>(xw, xw, false, 1.f, tid, nitems, xp3, xs3);
      }
    } else {
      // 2 / 4 bits: an image item is two value pairs = two granules (unit u, pair t: granules u * UP + t and + UP / 2): the lanes
      // that poll them write the image directly -- no staging of raw values, one barrier less
      constexpr int UPG = FO::UNIT_PAIRS;                     // granules per unit
      if ((tid & ~63) < nitems) {
        const int it = min(tid, nitems - 1);
        const int u = it / TPU3, t = it % TPU3;
        unsigned xw[PPU3];
        sweep2(hand2 + u * UPG + t, hand2 + u * UPG + UPG / 2 + t, tag, 2u, xw[0], xw[1]);
        DA_STAMP(10);
        da_image_item<BO, BF16, PPU3>(xw, xw, false, 1.f, tid, nitems, xp3, xs3);
      }
    }
    __syncthreads();
    DA_STAMP(11);
#pragma unroll
    for (int p = 0; p < DA_OP; ++p) {
      const int rl = (p * NW + wave) * rpp + (lane >> lpr_l2);
      if ((p * NW + wave) * rpp >= RPW) break;                // wave-uniform
      const bool live = rl < RPW && oslot < spr;
      float v = da_o_slot<BO, BF16>(ow[p], osc[p], oze[p], xp3, xs3, live ? oslot : 0);
      v = live ? v : 0.f;
      v += dpp_f32<0xB1>(v);
      v += dpp_f32<0x4E>(v);
      v += dpp_f32<0x141>(v);
      if (lpr_l2 >= 4) v += dpp_f32<0x140>(v);
      if (lpr_l2 >= 5) v += dpp_f32<0x142, 0xA>(v);
      if (oslot == (1 << lpr_l2) - 1 && rl < RPW) {
        const int row = r * RPW + rl;
        da_gu64* accp = reinterpret_cast<da_gu64*>(reinterpret_cast<uintptr_t>(a.acc));
        if (fabsf(v) < 2147483648.f) {
          const long long fx = __float2ll_rn(v * 4294967296.f);
          __hip_atomic_fetch_add(accp + row, static_cast<unsigned long long>(fx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          __hip_atomic_fetch_add(accp + D, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // NaN / Inf / out of range: the consumer makes x NaN
        }
      }
    }
  };
  if constexpr (MX != 0) {
    // ---- the o_proj slice on the matrix cores: 4 rows x 4 groups per operand, the attention output as two E4M3 limbs per value ----
    unsigned char* img3 = smem + DA_LDS_P3;                        // [limb][piece][64 lanes][16 B]; operand row m holds group m % 4
    float* bsum3 = reinterpret_cast<float*>(img3 + 4096);          // [m + 16 kb]
    unsigned char* scl3 = reinterpret_cast<unsigned char*>(bsum3 + 64);
    if (wave < 2) {                                                // 128 items of 4 values = two granules each
      const int it = tid;
      unsigned w0, w1;
      sweep2(hand2 + 2 * it, hand2 + 2 * it + 1, tag, 2u, w0, w1);
      DA_STAMP(10);
      const float v[4] = {act_lo<BF16>(w0), act_hi<BF16>(w0), act_lo<BF16>(w1), act_hi<BF16>(w1)};
      float sum = (v[0] + v[1]) + (v[2] + v[3]);
      unsigned amax = max(max(__float_as_uint(v[0]) & 0x7FFFFFFFu, __float_as_uint(v[1]) & 0x7FFFFFFFu),
                          max(__float_as_uint(v[2]) & 0x7FFFFFFFu, __float_as_uint(v[3]) & 0x7FFFFFFFu));
      sum += dpp_f32<0xB1>(sum); amax = max(amax, dpp_u32<0xB1>(amax));
      sum += dpp_f32<0x4E>(sum); amax = max(amax, dpp_u32<0x4E>(amax));
      sum += dpp_f32<0x141>(sum); amax = max(amax, dpp_u32<0x141>(amax));
      if (amax >= 0x7F800000u) sum = __uint_as_float(0x7FC00000u);
      int ea = static_cast<int>(amax >> 23);
      ea = min(max(ea, 20), 240);
      const float up = exp2i(134 - ea);
      const int b = it >> 3, u = it & 7;
      const int gi = b >> 2, kb = b & 3;
      const int ga = 2 * (kb & 1) + (u >> 2);
      const float r0 = v[0] * up, r1 = v[1] * up, r2 = v[2] * up, r3 = v[3] * up;
      const int pe = __builtin_amdgcn_cvt_pk_fp8_f32(r0, r2, 0, false);
      const int po = __builtin_amdgcn_cvt_pk_fp8_f32(r1, r3, 0, false);
      const int le = __builtin_amdgcn_cvt_pk_fp8_f32(r0 - __builtin_amdgcn_cvt_f32_fp8(pe, 0), r2 - __builtin_amdgcn_cvt_f32_fp8(pe, 1), 0, false);
      const int lo = __builtin_amdgcn_cvt_pk_fp8_f32(r1 - __builtin_amdgcn_cvt_f32_fp8(po, 0), r3 - __builtin_amdgcn_cvt_f32_fp8(po, 1), 0, false);
#pragma unroll
      for (int c = 0; c < 4; ++c) {                                // the four operand rows that carry this group
        const int m = gi + 4 * c;
        unsigned char* dst = img3 + (kb >> 1) * 1024 + (m + 16 * ga) * 16 + 2 * (u & 3);
        *reinterpret_cast<unsigned short*>(dst) = static_cast<unsigned short>(pe);
        *reinterpret_cast<unsigned short*>(dst + 8) = static_cast<unsigned short>(po);
        *reinterpret_cast<unsigned short*>(dst + 2048) = static_cast<unsigned short>(le);
        *reinterpret_cast<unsigned short*>(dst + 2048 + 8) = static_cast<unsigned short>(lo);
        if (u == 0) { bsum3[m + 16 * kb] = sum; scl3[m + 16 * kb] = static_cast<unsigned char>(ea - 7); }
      }
    }
    __syncthreads();
    DA_STAMP(11);
    da_v8i a3[2];
#pragma unroll
    for (int l = 0; l < 2; ++l) {
      const u32x4 p0 = *reinterpret_cast<const u32x4*>(img3 + 0 * 1024 + l * 2048 + lane * 16);
      const u32x4 p1 = *reinterpret_cast<const u32x4*>(img3 + 1 * 1024 + l * 2048 + lane * 16);
      a3[l][0] = p0.x; a3[l][1] = p0.y; a3[l][2] = p0.z; a3[l][3] = p0.w;
      a3[l][4] = p1.x; a3[l][5] = p1.y; a3[l][6] = p1.z; a3[l][7] = p1.w;
    }
    const int sa3 = static_cast<int>(scl3[lane]);
    const bool diag = mkg == (mnn >> 2);
    const float gs3 = (bsum3[mnn] + bsum3[mnn + 16]) + (bsum3[mnn + 32] + bsum3[mnn + 48]);
    const float xsg3 = diag ? gs3 : 0.f;
    float dm[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) dm[q] = (diag && (lane & 3) == q) ? 1.f : 0.f;
    da_gu64* accp = reinterpret_cast<da_gu64*>(reinterpret_cast<uintptr_t>(a.acc));
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      da_v8i bv;
      const unsigned w0 = owm[j].x, w1 = owm[j].y;
      bv[0] = w0 & 0x33333333u; bv[1] = (w0 >> 2) & 0x33333333u; bv[2] = w1 & 0x33333333u; bv[3] = (w1 >> 2) & 0x33333333u;
      bv[4] = 0; bv[5] = 0; bv[6] = 0; bv[7] = 0;
      da_v4f z = {0.f, 0.f, 0.f, 0.f};
      z = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a3[0], bv, z, 0, 4, 0, sa3, 0, 127);
      z = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a3[1], bv, z, 0, 4, 0, sa3, 0, 127);
      const float dsel = fmaf(z[0], dm[0], fmaf(z[1], dm[1], fmaf(z[2], dm[2], z[3] * dm[3])));
      const float sc = act_to_f32<BF16>(static_cast<unsigned short>(osc[j])), ze = act_to_f32<BF16>(static_cast<unsigned short>(oze[j]));
      float v = fmaf(ze, xsg3, (sc + sc) * dsel);                  // (row nn / 4, group nn % 4) on the diagonal lanes 20 (nn / 4) + nn % 4, 0 elsewhere
      v += dpp_f32<0xB1>(v);
      v += dpp_f32<0x4E>(v);                                       // the row's four groups: an aligned quad
      if (diag && (mnn & 3) == 0) {
        const int row = r * RPW + wave * 8 + 4 * j + (mnn >> 2);
        if (fabsf(v) < 2147483648.f) {
          const long long fx = __float2ll_rn(v * 4294967296.f);
          __hip_atomic_fetch_add(accp + row, static_cast<unsigned long long>(fx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          __hip_atomic_fetch_add(accp + D, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
  } else {
    if (obits == 2) phase3(std::integral_constant<int, 2>{});
    else if (obits == 3) phase3(std::integral_constant<int, 3>{});
    else phase3(std::integral_constant<int, 4>{});
  }
  // member 0 advances the group's generation: every member has read it -- member 0 passes phase 3 only after every attention workgroup's
  // output granules, an attention workgroup publishes only after every producer's granules, every producer owns at least one batch
  // (interleaved batch order above) and reads the generation before it publishes
  if (r == 0 && tid == 0) __hip_atomic_store(gen_p, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  DA_STAMP(6);
#ifdef GBX_STAMPS
  if (lane < 16 && a.stamps) a.stamps[(static_cast<size_t>(blockIdx.x) * DA_NW + wave) * 16 + lane] = reinterpret_cast<unsigned long long*>(smem + DA_LDS_BYTES)[wave * 16 + lane];
#endif
}

bool dattn_mx_wanted(const DattnArgs& a, bool bf16);      // dattn_common.hip: GBX_DATTN_MX and the shapes the matrix-core form covers
// the matrix-core instantiation exists for <2, 1> only (specialised in dattn_b2.hip)
template <int BQ, int P_> struct DattnMx {
  static constexpr bool has = false;
  static hipError_t launch(const DattnArgs&, hipStream_t) { return hipErrorInvalidValue; }
};
#define GBX_DEFINE_LAUNCH_DATTN_MX(BQ, P_)                                                                                   \
  template <> struct DattnMx<BQ, P_> {                                                                                       \
    static constexpr bool has = true;                                                                                        \
    static hipError_t launch(const DattnArgs& a, hipStream_t stream) {                                                       \
      static DeviceOnce once;                                                                                                \
      if (once.pending()) {                                                                                                  \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(dattn_kernel<BQ, P_, true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, DA_LDS_BYTES + DA_STAMP_EXTRA); \
        if (e != hipSuccess) return e;                                                                                       \
        once.done();                                                                                                         \
      }                                                                                                                      \
      hipLaunchKernelGGL((dattn_kernel<BQ, P_, true, 2>), dim3(DA_GROUPS * DA_MEMBERS), dim3(DA_NT), DA_LDS_BYTES + DA_STAMP_EXTRA, stream, a);       \
      return hipGetLastError();                                                                                              \
    }                                                                                                                        \
  };
#define GBX_DEFINE_LAUNCH_DATTN(BQ, P_)                                                                                      \
  template <> hipError_t launch_dattn_bits<BQ>(const DattnArgs& a, const DattnPlan& pl, bool bf16, hipStream_t stream) {     \
    static DeviceOnce once;                                                                                                  \
    if (pl.P != P_) return hipErrorInvalidValue;                                                                             \
    if (DattnMx<BQ, P_>::has && dattn_mx_wanted(a, bf16)) return DattnMx<BQ, P_>::launch(a, stream);                         \
    if (once.pending()) {                                                                                                    \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(dattn_kernel<BQ, P_, true>), hipFuncAttributeMaxDynamicSharedMemorySize, DA_LDS_BYTES + DA_STAMP_EXTRA); \
      if (e != hipSuccess) return e;                                                                                         \
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(dattn_kernel<BQ, P_, false>), hipFuncAttributeMaxDynamicSharedMemorySize, DA_LDS_BYTES + DA_STAMP_EXTRA);           \
      if (e != hipSuccess) return e;                                                                                         \
      once.done();                                                                                                           \
    }                                                                                                                        \
    if (bf16) hipLaunchKernelGGL((dattn_kernel<BQ, P_, true>), dim3(DA_GROUPS * DA_MEMBERS), dim3(DA_NT), DA_LDS_BYTES + DA_STAMP_EXTRA, stream, a);   \
    else hipLaunchKernelGGL((dattn_kernel<BQ, P_, false>), dim3(DA_GROUPS * DA_MEMBERS), dim3(DA_NT), DA_LDS_BYTES + DA_STAMP_EXTRA, stream, a);       \
    return hipGetLastError();                                                                                                \
  }

}  // namespace gbx
