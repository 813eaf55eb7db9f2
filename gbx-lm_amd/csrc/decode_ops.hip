// Callers of the QuantizedLinear hot path inside one decode step, as gfx950 kernels (batch 1, one new token):
//   * attn_decode : RoPE(q, k) + KV-cache append + softmax(q k^T) v for one layer  (gbx_lm/models/qllama.py:76-95)
//   * dense_gemv  : unquantised lm_head / tied embed_tokens.as_linear, optional fused final RMSNorm
//                   (gbx_lm/models/qllama.py:174,194-198)
//   * rmsnorm     : standalone RMSNorm for shapes the GEMV prologue does not fuse
// All read the decode position from DEVICE memory so one captured HIP graph serves every token.
#include "gbx_device.h"
#include "gbx_env.h"
#include "kernels.h"
#include <cstdlib>

namespace gbx {

// ------------------------------------------------------------------------------------------------ attention (decode)
struct AttnArgs {
  const unsigned short* q;      // [n_heads * HD]    (projection outputs, un-rotated)
  const unsigned short* k;      // [n_kv * HD]
  const unsigned short* v;      // [n_kv * HD]
  unsigned short* kcache;       // [n_kv, capacity, HD]
  unsigned short* vcache;       // [n_kv, capacity, HD]
  const float* inv_freq;        // [HD/2] (rope dims == HD)
  const long long* pos;         // device scalar: index of the new token
  unsigned short* out;          // [n_heads * HD]
  int n_heads, n_kv, capacity;
  float rope_scale, sm_scale;
  int traditional;
  // grouped kernel only: optional per-head RMSNorm of q and k before RoPE (Qwen3's q_norm / k_norm, weights [HD])
  const unsigned short* qn_w;
  const unsigned short* kn_w;
  float norm_eps;
  // grouped kernel only: optional biases of the q / k / v projections (Qwen2), added here as the second rounding the
  // projection's `+ bias` would have done (all three or none)
  const unsigned short* q_bias;
  const unsigned short* k_bias;
  const unsigned short* v_bias;
  // grouped kernel only: long contexts split the cached positions over gridDim.z workgroups per (kv group, sequence);
  // split s writes (num[HD], max, den) per head to ws[((b * n_heads + head) * splits + s) * (HD + 2)] and a second launch
  // merges them (caller-owned scratch)
  float* ws;
  // grouped kernel only (batched decode): q | k | v arrive as the fp32 slices of ONE k-split GEMM over the concatenated projection,
  // qkv_ws[slice][batch][qkv_n] with qkv_n = (n_heads + 2 n_kv) * HD (columns q | k | v); a value is round(ws[0] + ws[1] + ...) exactly
  // as the GEMM's summing launch would have written it (batch_finish_kernel).  q / k / v above are then unused (null).
  const float* qkv_ws;
  int qkv_slices;
  long long qkv_mn;             // floats per slice = batch * qkv_n
#ifdef GBX_STAMPS
  unsigned long long* stamps;   // diagnostic build only
#endif
};
#ifdef GBX_STAMPS
#define ATT_STAMP(i) do { if ((threadIdx.x & 63) == 0 && a.stamps) a.stamps[(static_cast<size_t>(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define ATT_STAMP(i) do {} while (0)
#endif

// NW waves per head.  NW = 4: every (wave, lane group) keeps its own softmax state and the 4 * PPW states meet in LDS.
// NW = 16 (round 4, head_dim 128): the sweep over the cached positions is the kernel's longest phase (1.2 of 4.1 us at context
// 128-190: one wave per SIMD walking 8 dependent rounds of load -> dot -> exp -> accumulate), so sixteen waves take two rounds each;
// the four lane groups of a wave merge their states in registers (v_permlane32_swap / v_permlane16_swap) and 16 states meet in LDS,
// as many as before.
#ifndef GBX_ATTN_XCD_ORDER
#define GBX_ATTN_XCD_ORDER 1
#endif
template <bool BF16, int HD, int NW = 4>
__global__ __launch_bounds__(NW * 64) void attn_decode_kernel(AttnArgs a) {
  constexpr bool XCD_ORDER = GBX_ATTN_XCD_ORDER != 0;
  constexpr int LP = HD / 8;          // lanes that share one cached position (8 features = one 16-byte load each)
  constexpr int PPW = 64 / LP;        // positions per wave per iteration
  constexpr bool WMERGE = NW > 4;     // lane groups of a wave merged in registers
  static_assert(!WMERGE || HD == 128, "in-wave merge: four lane groups of 16");
  constexpr int NG = WMERGE ? NW : NW * PPW;         // partial softmax states that meet in LDS
  constexpr float NEG = -1e30f;
  __shared__ float q_s[HD], kn_s[HD], vn_s[HD];
  __shared__ float gm[NG], gl[NG];
  __shared__ float gacc[NG][HD + 8];

  // XCD-aware head order (speed only, never correctness): workgroups b and b + 8 share an XCD under the round-robin placement, so
  // slot b % 8 takes the CONSECUTIVE heads [slot * H/8, (slot + 1) * H/8) -- at 32 heads / 8 kv heads the four query heads of one kv
  // group -- and three of the four workgroups find their K / V head in the XCD's L2 instead of pulling it from the Infinity Cache.
  int h = blockIdx.x;
  if (XCD_ORDER && (a.n_heads & 7) == 0) h = (static_cast<int>(blockIdx.x) & 7) * (a.n_heads >> 3) + (static_cast<int>(blockIdx.x) >> 3);
  const int rep = a.n_heads / a.n_kv;
  const int kvh = h / rep;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  ATT_STAMP(0);
  {                                   // synchronous batched decode: sequence blockIdx.y, every sequence at *pos
    const size_t b = blockIdx.y;
    a.q += b * a.n_heads * HD; a.out += b * a.n_heads * HD;
    a.k += b * a.n_kv * HD; a.v += b * a.n_kv * HD;
    a.kcache += b * a.n_kv * a.capacity * HD; a.vcache += b * a.n_kv * a.capacity * HD;
  }
  // ---- the first UB0 * 4 * PPW cached positions are requested before anything else: their addresses depend neither on
  //      the position counter (clamped to the buffer; masked when consumed) nor on q, so the cache latency overlaps the
  //      q/k/v loads, RoPE and the barrier instead of following them (one dependent memory round trip less) ----
  constexpr int LP0 = HD / 8, PPW0 = 64 / LP0, UB0 = NW == 4 ? 8 : 128 / (NW * PPW0);
  const int d0 = (lane % LP0) * 8, sub = lane / LP0;
  const unsigned short* kbase = a.kcache + static_cast<size_t>(kvh) * a.capacity * HD;
  const unsigned short* vbase = a.vcache + static_cast<size_t>(kvh) * a.capacity * HD;
  u32x4 k0[UB0], v0[UB0];
#pragma unroll
  for (int u = 0; u < UB0; ++u) {
    const int tc = min(wave * PPW0 + u * NW * PPW0 + sub, a.capacity - 1);
    k0[u] = *reinterpret_cast<const u32x4*>(kbase + static_cast<size_t>(tc) * HD + d0);
    v0[u] = *reinterpret_cast<const u32x4*>(vbase + static_cast<size_t>(tc) * HD + d0);
  }
  const int pos = static_cast<int>(*a.pos);
  ATT_STAMP(1);

  // ---- RoPE on this head's q and on its kv head's k (fp32 math, one rounding to the activation dtype like
  //      mx.fast.rope); the first q head of a kv group appends k, v to the cache ----
  if (tid < HD) {
    const bool isk = tid >= HD / 2;
    const int i = isk ? tid - HD / 2 : tid;
    const unsigned short* src = isk ? a.k + static_cast<size_t>(kvh) * HD : a.q + static_cast<size_t>(h) * HD;
    const int i1 = a.traditional ? 2 * i : i;
    const int i2 = a.traditional ? 2 * i + 1 : i + HD / 2;
    const float x1 = act_to_f32<BF16>(src[i1]), x2 = act_to_f32<BF16>(src[i2]);
    // angle in revolutions, reduced to [0,1): v_sin/v_cos take revolutions and are accurate to ~1e-6 there
    const float rev = static_cast<float>(pos) * a.rope_scale * a.inv_freq[i] * 0.15915494309189535f;
    const float fr = rev - floorf(rev);
    const float sn = __builtin_amdgcn_sinf(fr), cs = __builtin_amdgcn_cosf(fr);
    const unsigned short r1 = f32_to_act<BF16>(x1 * cs - x2 * sn), r2 = f32_to_act<BF16>(x1 * sn + x2 * cs);
    if (isk) {
      kn_s[i1] = act_to_f32<BF16>(r1);
      kn_s[i2] = act_to_f32<BF16>(r2);
      if (h % rep == 0) {
        unsigned short* dst = a.kcache + (static_cast<size_t>(kvh) * a.capacity + pos) * HD;
        dst[i1] = r1;
        dst[i2] = r2;
      }
    } else {
      q_s[i1] = act_to_f32<BF16>(r1) * a.sm_scale;
      q_s[i2] = act_to_f32<BF16>(r2) * a.sm_scale;
    }
  } else if (tid < 2 * HD) {
    const int j = tid - HD;
    const unsigned short vb = a.v[static_cast<size_t>(kvh) * HD + j];
    vn_s[j] = act_to_f32<BF16>(vb);
    if (h % rep == 0) a.vcache[(static_cast<size_t>(kvh) * a.capacity + pos) * HD + j] = vb;
  }
  __syncthreads();
  ATT_STAMP(2);

  // ---- cached positions [0, pos): LP lanes per position, online softmax per (wave, sub) group ----
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
    // sum over the LP lanes of this position (LP = 16: one DPP row; LP = 8: half a row)
    s += dpp_f32<0xB1>(s);
    s += dpp_f32<0x4E>(s);
    s += dpp_f32<0x141>(s);
    if constexpr (LP == 16) s += dpp_f32<0x140>(s);
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
  for (int u = 0; u < UB0; ++u) consume(k0[u], v0[u], wave * PPW + u * NW * PPW + sub < pos);
  constexpr int UB = NW == 4 ? 4 : 2;   // iterations whose K/V loads are in flight together
  for (int t0 = UB0 * NW * PPW + wave * PPW; t0 < pos; t0 += NW * PPW * UB) {
    u32x4 kv4[UB], vv4[UB];
    bool valid[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int t = t0 + u * NW * PPW + sub;
      valid[u] = t < pos;
      const int tc = valid[u] ? t : pos - 1;
      kv4[u] = *reinterpret_cast<const u32x4*>(kbase + static_cast<size_t>(tc) * HD + d0);
      vv4[u] = *reinterpret_cast<const u32x4*>(vbase + static_cast<size_t>(tc) * HD + d0);
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) consume(kv4[u], vv4[u], valid[u]);
  }
  // ---- the new token itself (its k, v are in LDS, not yet visible through the cache to other blocks) ----
  if (wave == 0 && sub == 0) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s = fmaf(qr[j], kn_s[d0 + j], s);
    s += dpp_f32<0xB1>(s);
    s += dpp_f32<0x4E>(s);
    s += dpp_f32<0x141>(s);
    if constexpr (LP == 16) s += dpp_f32<0x140>(s);
    const float mn = fmaxf(m, s);
    const float corr = __expf(m - mn), p = __expf(s - mn);
    l = l * corr + p;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = acc[j] * corr + p * vn_s[d0 + j];
    m = mn;
  }
  if constexpr (WMERGE) {
    // the wave's four lane groups (rows of 16 lanes, each holding the same 8 features per lane) -> one state, in registers
    auto merge = [&](float pm, float pl, const float (&pa)[8]) {
      const float mn = fmaxf(m, pm);
      const float c1 = __expf(m - mn), c2 = __expf(pm - mn);
      l = l * c1 + pl * c2;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = acc[j] * c1 + pa[j] * c2;
      m = mn;
    };
    {   // partner lane ^ 32
      const bool hi = lane >= 32;
      auto part = [&](float v) {
        auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        const unsigned r0 = r[0], r1 = r[1];
        return __uint_as_float(hi ? r0 : r1);
      };
      float pa[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) pa[j] = part(acc[j]);
      const float pm = part(m), pl = part(l);
      merge(pm, pl, pa);
    }
    {   // partner lane ^ 16
      const bool odd = (lane >> 4) & 1;
      auto part = [&](float v) {
        auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        const unsigned r0 = r[0], r1 = r[1];
        return __uint_as_float(odd ? r0 : r1);
      };
      float pa[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) pa[j] = part(acc[j]);
      const float pm = part(m), pl = part(l);
      merge(pm, pl, pa);
    }
  }
  const int g = WMERGE ? wave : wave * PPW + sub;
  ATT_STAMP(3);
  if (WMERGE ? lane == 0 : lane % LP == 0) { gm[g] = m; gl[g] = l; }
  if (!WMERGE || sub == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) gacc[g][d0 + j] = acc[j];
  }
  __syncthreads();
  ATT_STAMP(4);
  // ---- merge the NG partial states; one output feature per thread ----
  if (tid < HD) {
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
    a.out[static_cast<size_t>(h) * HD + tid] = f32_to_act<BF16>(num / den);
  }
  ATT_STAMP(6);
}

// (Round 2, measured and not kept: the sweep below on the matrix cores -- S^T = mfma(K rows from the cache in operand layout,
// Q^T), O^T += mfma(V^T by transposed LDS reads of a wave-private V tile, P^T), the REP heads as REP of the 16 columns of a
// 16x16x32 tile, 472 GPU tests green -- B = 32: 2.60 -> 2.56 ms/step at context 128, 3.37 -> 3.32 at 1024; batch 1 at context
// 2048: 648 -> 640 tok/s.  The kernel is a latency chain (projection loads -> RoPE -> barrier -> K / V round trip -> merge),
// not vector-ALU work: the arithmetic it removed was not on the critical path.)
// Batched decode variant: one workgroup per (kv head, sequence) serves all REP query heads of the group, so every cached
// K / V row is read ONCE per group instead of once per query head (at B = 32 the per-head kernel moved 4x the cache
// bytes: 67 MB per layer at context 128).  Same arithmetic and layout as attn_decode_kernel.
// NW waves per workgroup (4, or 8: round 4 -- at B = 32 / context 128-190 the four-wave form is a chain of 8 dependent sweep rounds with
// four heads' arithmetic in each and one wave per SIMD: 9.4 us per layer; eight waves take half the rounds each, two per SIMD).
#ifndef GQA_UB0_WIDE
#define GQA_UB0_WIDE 2
#endif
template <bool BF16, int HD, int REP, bool STATIC, int NW = 4>
__global__ __launch_bounds__(NW * 64) void attn_decode_gqa_kernel(AttnArgs a) {
  constexpr int LP = HD / 8, PPW = 64 / LP, NG = NW * PPW, NT = NW * 64;
  constexpr float NEG = -1e30f;
  __shared__ float q_s[REP][HD], kn_s[HD], vn_s[HD];
  __shared__ float gm[NG][REP], gl[NG][REP];
  __shared__ float gacc[NG][REP][HD + 8];

  // STATIC (family-agnostic graph decode, models/base.py): q is already rotated, the new token's k, v are already in
  // the cache at *pos -- plain softmax(q K^T) V over positions 0..*pos.  A workgroup then serves REP of the group's
  // rep = n_heads / n_kv query heads (blockIdx.x = kv head * parts + part).
  const int rep = a.n_heads / a.n_kv, parts = rep / REP;
  const int kvh = blockIdx.x / parts, hq0 = kvh * rep + (blockIdx.x % parts) * REP;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  {
    const size_t b = blockIdx.y;
    a.out += b * a.n_heads * HD;
    if (a.q) { a.q += b * a.n_heads * HD; a.k += b * a.n_kv * HD; a.v += b * a.n_kv * HD; }
    a.kcache += b * a.n_kv * a.capacity * HD; a.vcache += b * a.n_kv * a.capacity * HD;
  }
  constexpr int UB0 = NW == 4 ? 4 : GQA_UB0_WIDE;      // 64 speculative positions either way (round 5, slices requested first: 96 -> 9.5 us, 160 -> 11.1 us, 64 -> 8.6 us per launch at B = 32)
  const int d0 = (lane % LP) * 8, sub = lane / LP;
  const unsigned short* kbase = a.kcache + static_cast<size_t>(kvh) * a.capacity * HD;
  const unsigned short* vbase = a.vcache + static_cast<size_t>(kvh) * a.capacity * HD;
  // Slice form (batched decode): loads return in order, and RoPE needs this thread's 2 x 8 slice values (12 KB per workgroup), not the
  // 64 KB of speculative cache rows -- so the slices are requested FIRST (one (head, pair) item per thread when the workgroup is wide
  // enough; the v values by the first HD threads) and summed when RoPE gets to them.
  constexpr int NIT = (REP + 1) * (HD / 2);
  const bool pre = !STATIC && a.qkv_ws != nullptr && NT >= NIT && NT >= HD;
  float ps1[8], ps2[8], psv[8];
  if (pre) {
    const float* wsrow0 = a.qkv_ws + static_cast<size_t>(blockIdx.y) * ((a.n_heads + 2 * a.n_kv) * HD);
    const int pp = min(tid, NIT - 1);
    const int hh = pp / (HD / 2), i = pp % (HD / 2);
    const int col = hh == REP ? a.n_heads * HD + kvh * HD : (hq0 + hh) * HD;
    const int i1 = a.traditional ? 2 * i : i, i2 = a.traditional ? 2 * i + 1 : i + HD / 2;
    const float* pv = wsrow0 + (a.n_heads + a.n_kv) * HD + kvh * HD + min(tid, HD - 1);
#pragma unroll
    for (int sl = 0; sl < 8; ++sl) {
      const size_t so = static_cast<size_t>(min(sl, a.qkv_slices - 1)) * a.qkv_mn;
      ps1[sl] = wsrow0[col + i1 + so];
      ps2[sl] = wsrow0[col + i2 + so];
      psv[sl] = pv[so];
    }
  }
  u32x4 k0[UB0], v0[UB0];
#pragma unroll
  for (int u = 0; u < UB0; ++u) {                   // first 4 * 4 * PPW positions requested before *pos is known
    const int tc = min(wave * PPW + u * NW * PPW + sub, a.capacity - 1);
    k0[u] = *reinterpret_cast<const u32x4*>(kbase + static_cast<size_t>(tc) * HD + d0);
    v0[u] = *reinterpret_cast<const u32x4*>(vbase + static_cast<size_t>(tc) * HD + d0);
  }
  const int pos = static_cast<int>(*a.pos);
  const int nctx = STATIC ? pos + 1 : pos;          // cached positions to visit
  // this workgroup's slice of them (all of them unless the launch splits the context)
  const int nsplit = a.ws ? static_cast<int>(gridDim.z) : 1, sp = a.ws ? static_cast<int>(blockIdx.z) : 0;
  const int per = nsplit > 1 ? (((nctx + nsplit - 1) / nsplit + NW * PPW - 1) / (NW * PPW)) * (NW * PPW) : nctx;
  const int tb = sp * per, te = min(nctx, tb + per);
  const bool last_split = sp == nsplit - 1;

  if constexpr (STATIC) {
    for (int p = tid; p < REP * HD; p += NT) q_s[p / HD][p % HD] = act_to_f32<BF16>(a.q[static_cast<size_t>(hq0) * HD + p]) * a.sm_scale;
  } else {
  __shared__ float ssq_s[REP + 1];
  // projection output (+ bias, rounded to the activation dtype like the projection's own `+ bias`)
  // (`col`: first column of this head inside the concatenated q | k | v row, for the slice form)
  const int qkv_n = (a.n_heads + 2 * a.n_kv) * HD;
  const float* wsrow = a.qkv_ws ? a.qkv_ws + static_cast<size_t>(blockIdx.y) * qkv_n : nullptr;
  auto sum8 = [&](const float (&t)[8]) {              // v = ws[0]; v += ws[s], s = 1..: the summing launch's order, one rounding
    float v = t[0];
#pragma unroll
    for (int sl = 1; sl < 8; ++sl)
      if (sl < a.qkv_slices) v += t[sl];
    return act_to_f32<BF16>(f32_to_act<BF16>(v));
  };
  // which: 0 = no preloaded values, 1 / 2 = this thread's first / second value of its pair, 3 = its v value
  auto ldb = [&](int col, const unsigned short* src, const unsigned short* bias, int i, int which = 0) {
    float x;
    if (wsrow) {                                    // every slice requested before the first add
      if (which == 1) x = sum8(ps1);
      else if (which == 2) x = sum8(ps2);
      else if (which == 3) x = sum8(psv);
      else {
        const float* p = wsrow + col + i;
        float t[8];
#pragma unroll
        for (int sl = 0; sl < 8; ++sl) t[sl] = p[static_cast<size_t>(min(sl, a.qkv_slices - 1)) * a.qkv_mn];
        x = sum8(t);
      }
    } else {
      x = act_to_f32<BF16>(src[i]);
    }
    return bias ? act_to_f32<BF16>(f32_to_act<BF16>(x + act_to_f32<BF16>(bias[i]))) : x;
  };
  const int col_k = a.n_heads * HD + kvh * HD, col_v = (a.n_heads + a.n_kv) * HD + kvh * HD;
  const bool normed = a.qn_w != nullptr;
  const bool writer = blockIdx.x % parts == 0 && blockIdx.z == 0;   // one workgroup of the kv group appends k, v
  if (normed) {                                     // sum of squares per head (q heads of this part, then the k head)
    if (tid <= REP) ssq_s[tid] = 0.f;
    __syncthreads();
    for (int p = tid; p < (REP + 1) * (HD / 2); p += NT) {
      const int hh = p / (HD / 2), i = p % (HD / 2);
      const unsigned short* src = hh == REP ? a.k + static_cast<size_t>(kvh) * HD : a.q + (static_cast<size_t>(hq0) + hh) * HD;
      const unsigned short* bsrc = !a.q_bias ? nullptr : hh == REP ? a.k_bias + static_cast<size_t>(kvh) * HD : a.q_bias + (static_cast<size_t>(hq0) + hh) * HD;
      const int col = hh == REP ? col_k : (hq0 + hh) * HD;
      const float x1 = ldb(col, src, bsrc, 2 * i), x2 = ldb(col, src, bsrc, 2 * i + 1);
      atomicAdd(&ssq_s[hh], x1 * x1 + x2 * x2);
    }
    __syncthreads();
  }
  // RoPE on the REP query heads and the kv head's k; append k, v to the cache
  for (int p = tid; p < (REP + 1) * (HD / 2); p += NT) {
    const int hh = p / (HD / 2), i = p % (HD / 2);   // hh == REP: the k head
    const bool isk = hh == REP;
    const unsigned short* src = isk ? a.k + static_cast<size_t>(kvh) * HD : a.q + (static_cast<size_t>(hq0) + hh) * HD;
    const int i1 = a.traditional ? 2 * i : i;
    const int i2 = a.traditional ? 2 * i + 1 : i + HD / 2;
    const unsigned short* bsrc = !a.q_bias ? nullptr : isk ? a.k_bias + static_cast<size_t>(kvh) * HD : a.q_bias + (static_cast<size_t>(hq0) + hh) * HD;
    const int col = isk ? col_k : (hq0 + hh) * HD;
    float x1 = ldb(col, src, bsrc, i1, pre ? 1 : 0), x2 = ldb(col, src, bsrc, i2, pre ? 2 : 0);     // (pre: one item per thread, p == tid)
    if (normed) {                                   // RMSNorm output rounded to the activation dtype, like the separate op
      const float rstd = rsqrtf(ssq_s[hh] / static_cast<float>(HD) + a.norm_eps);
      const unsigned short* w = isk ? a.kn_w : a.qn_w;
      x1 = act_to_f32<BF16>(f32_to_act<BF16>(x1 * rstd * act_to_f32<BF16>(w[i1])));
      x2 = act_to_f32<BF16>(f32_to_act<BF16>(x2 * rstd * act_to_f32<BF16>(w[i2])));
    }
    const float rev = static_cast<float>(pos) * a.rope_scale * a.inv_freq[i] * 0.15915494309189535f;
    const float fr = rev - floorf(rev);
    const float sn = __builtin_amdgcn_sinf(fr), cs = __builtin_amdgcn_cosf(fr);
    const unsigned short r1 = f32_to_act<BF16>(x1 * cs - x2 * sn), r2 = f32_to_act<BF16>(x1 * sn + x2 * cs);
    if (isk) {
      kn_s[i1] = act_to_f32<BF16>(r1);
      kn_s[i2] = act_to_f32<BF16>(r2);
      if (writer) {
        unsigned short* dst = a.kcache + (static_cast<size_t>(kvh) * a.capacity + pos) * HD;
        dst[i1] = r1;
        dst[i2] = r2;
      }
    } else {
      q_s[hh][i1] = act_to_f32<BF16>(r1) * a.sm_scale;
      q_s[hh][i2] = act_to_f32<BF16>(r2) * a.sm_scale;
    }
  }
  if (tid < HD) {
    const unsigned short vb = f32_to_act<BF16>(ldb(col_v, a.v + static_cast<size_t>(kvh) * HD, a.v_bias ? a.v_bias + static_cast<size_t>(kvh) * HD : nullptr, tid, pre ? 3 : 0));
    vn_s[tid] = act_to_f32<BF16>(vb);
    if (writer) a.vcache[(static_cast<size_t>(kvh) * a.capacity + pos) * HD + tid] = vb;
  }
  }
  __syncthreads();

  float qr[REP][8], m[REP], l[REP], acc[REP][8];
#pragma unroll
  for (int r = 0; r < REP; ++r) {
    m[r] = NEG; l[r] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { qr[r][j] = q_s[r][d0 + j]; acc[r][j] = 0.f; }
  }
  auto consume = [&](const u32x4& kq, const u32x4& vq, bool valid) {
    const unsigned kw[4] = {kq.x, kq.y, kq.z, kq.w};
    const unsigned vw[4] = {vq.x, vq.y, vq.z, vq.w};
    float kf[8], vf[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      kf[2 * j] = act_lo<BF16>(kw[j]); kf[2 * j + 1] = act_hi<BF16>(kw[j]);
      vf[2 * j] = act_lo<BF16>(vw[j]); vf[2 * j + 1] = act_hi<BF16>(vw[j]);
    }
#pragma unroll
    for (int r = 0; r < REP; ++r) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) s = fmaf(qr[r][j], kf[j], s);
      s += dpp_f32<0xB1>(s);
      s += dpp_f32<0x4E>(s);
      s += dpp_f32<0x141>(s);
      if constexpr (LP == 16) s += dpp_f32<0x140>(s);
      const float mn = valid ? fmaxf(m[r], s) : m[r];
      const float corr = __expf(m[r] - mn);
      const float p = valid ? __expf(s - mn) : 0.f;
      l[r] = l[r] * corr + p;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[r][j] = acc[r][j] * corr + p * vf[j];
      m[r] = mn;
    }
  };
#pragma unroll
  // (the speculative first positions belong to split 0; the other splits start their own slice below)
  for (int u = 0; u < UB0; ++u) consume(k0[u], v0[u], sp == 0 && wave * PPW + u * NW * PPW + sub < te);
  // Main loop: UB positions per lane group as ONE block of the online softmax -- UB independent dot / DPP chains, one
  // rescale of the accumulators per block instead of per position (per position the chain dot -> DPP -> max -> exp ->
  // acc * corr ran serially UB times with a single wave or two per SIMD to hide it: 1.2 TB/s of cache at long contexts).
  constexpr int UB = 4;
  for (int t0 = (sp == 0 ? UB0 * NW * PPW : tb) + wave * PPW; t0 < te; t0 += NW * PPW * UB) {
    u32x4 kv4[UB], vv4[UB];
    bool valid[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int t = t0 + u * NW * PPW + sub;
      valid[u] = t < te;
      const int tc = valid[u] ? t : max(te - 1, 0);
      kv4[u] = *reinterpret_cast<const u32x4*>(kbase + static_cast<size_t>(tc) * HD + d0);
      vv4[u] = *reinterpret_cast<const u32x4*>(vbase + static_cast<size_t>(tc) * HD + d0);
    }
    float sc[REP][UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const unsigned kw[4] = {kv4[u].x, kv4[u].y, kv4[u].z, kv4[u].w};
#pragma unroll
      for (int r = 0; r < REP; ++r) {
        float d = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) d = fmaf(qr[r][2 * j], act_lo<BF16>(kw[j]), fmaf(qr[r][2 * j + 1], act_hi<BF16>(kw[j]), d));
        d += dpp_f32<0xB1>(d);
        d += dpp_f32<0x4E>(d);
        d += dpp_f32<0x141>(d);
        if constexpr (LP == 16) d += dpp_f32<0x140>(d);
        sc[r][u] = valid[u] ? d : NEG;
      }
    }
#pragma unroll
    for (int r = 0; r < REP; ++r) {
      float mn = m[r];
#pragma unroll
      for (int u = 0; u < UB; ++u) mn = fmaxf(mn, sc[r][u]);
      const float corr = __expf(m[r] - mn);
      float pw[UB], ps = 0.f;
#pragma unroll
      for (int u = 0; u < UB; ++u) { pw[u] = valid[u] ? __expf(sc[r][u] - mn) : 0.f; ps += pw[u]; }
      l[r] = l[r] * corr + ps;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[r][j] *= corr;
      m[r] = mn;
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const unsigned vw[4] = {vv4[u].x, vv4[u].y, vv4[u].z, vv4[u].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[r][2 * j] = fmaf(pw[u], act_lo<BF16>(vw[j]), acc[r][2 * j]);
          acc[r][2 * j + 1] = fmaf(pw[u], act_hi<BF16>(vw[j]), acc[r][2 * j + 1]);
        }
      }
    }
  }
  // the new token itself (k, v in LDS)
  if (!STATIC && last_split && wave == 0 && sub == 0) {
#pragma unroll
    for (int r = 0; r < REP; ++r) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) s = fmaf(qr[r][j], kn_s[d0 + j], s);
      s += dpp_f32<0xB1>(s);
      s += dpp_f32<0x4E>(s);
      s += dpp_f32<0x141>(s);
      if constexpr (LP == 16) s += dpp_f32<0x140>(s);
      const float mn = fmaxf(m[r], s);
      const float corr = __expf(m[r] - mn), p = __expf(s - mn);
      l[r] = l[r] * corr + p;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[r][j] = acc[r][j] * corr + p * vn_s[d0 + j];
      m[r] = mn;
    }
  }
  const int g = wave * PPW + sub;
#pragma unroll
  for (int r = 0; r < REP; ++r) {
    if (lane % LP == 0) { gm[g][r] = m[r]; gl[g][r] = l[r]; }
#pragma unroll
    for (int j = 0; j < 8; ++j) gacc[g][r][d0 + j] = acc[r][j];
  }
  __syncthreads();
  for (int o = tid; o < REP * HD; o += NT) {
    const int r = o / HD, f = o % HD;
    float M = NEG;
#pragma unroll
    for (int i = 0; i < NG; ++i) M = fmaxf(M, gm[i][r]);
    float num = 0.f, den = 0.f;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      const float w = __expf(gm[i][r] - M);
      num = fmaf(w, gacc[i][r][f], num);
      den = fmaf(w, gl[i][r], den);
    }
    if (nsplit > 1) {                               // partial of this slice: merged by attn_merge_kernel
      float* wsp = a.ws + ((static_cast<size_t>(blockIdx.y) * a.n_heads + hq0 + r) * nsplit + sp) * (HD + 2);
      wsp[f] = num;
      if (f == 0) { wsp[HD] = M; wsp[HD + 1] = den; }
    } else {
      a.out[(static_cast<size_t>(hq0) + r) * HD + f] = f32_to_act<BF16>(num / den);
    }
  }
}

// second launch of a split-context attention: out[b, head, :] = sum_s e^(M_s - M) num_s / sum_s e^(M_s - M) den_s.
// The (max, den) pairs of the <= 64 slices are read by the first wave in parallel and the weights shared through LDS; the
// numerator loop then has independent, coalesced loads (the first version walked the slices with dependent scalar-style
// loads: 9.5 us per launch, as long as the attention itself).
template <bool BF16, int HD>
__global__ __launch_bounds__(HD) void attn_merge_kernel(const float* __restrict__ ws, unsigned short* __restrict__ out, int n_heads,
                                                        int nsplit) {
  __shared__ float w_s[64];
  __shared__ float den_s;
  const size_t bh = static_cast<size_t>(blockIdx.y) * n_heads + blockIdx.x;
  const float* p = ws + bh * nsplit * (HD + 2);
  const int f = threadIdx.x;
  if (f < 64) {
    const float ml = f < nsplit ? p[f * (HD + 2) + HD] : -1e30f;
    const float dl = f < nsplit ? p[f * (HD + 2) + HD + 1] : 0.f;
    const float M = key2f(wave_max_u32(f2key(ml)));
    const float w = f < nsplit ? __expf(ml - M) : 0.f;
    w_s[f] = w;
    const float den = wave_sum(w * dl);
    if (f == 0) den_s = den;
  }
  __syncthreads();
  float num0 = 0.f, num1 = 0.f;
  int s = 0;
  for (; s + 1 < nsplit; s += 2) {
    num0 = fmaf(w_s[s], p[s * (HD + 2) + f], num0);
    num1 = fmaf(w_s[s + 1], p[(s + 1) * (HD + 2) + f], num1);
  }
  if (s < nsplit) num0 = fmaf(w_s[s], p[s * (HD + 2) + f], num0);
  out[bh * HD + f] = f32_to_act<BF16>((num0 + num1) / den_s);
}

template <int HD, int REP>
static hipError_t launch_attn_gqa(const AttnArgs& a, int batch, bool bf16, hipStream_t s, int splits = 1) {
  dim3 grid(a.n_kv * (a.n_heads / a.n_kv / REP), batch, a.ws ? splits : 1), block(256);
  static const bool wide = env_switch("GBX_ATTN_GQA_WIDE") != 0;
  if constexpr (HD == 128 && REP == 4) {
    if (wide && splits <= 1 && !a.ws) {               // eight waves per (kv head, sequence): unsplit contexts of batched decode
      if (bf16) hipLaunchKernelGGL((attn_decode_gqa_kernel<true, HD, REP, false, 8>), grid, dim3(512), 0, s, a);
      else hipLaunchKernelGGL((attn_decode_gqa_kernel<false, HD, REP, false, 8>), grid, dim3(512), 0, s, a);
      return hipGetLastError();
    }
  }
  if (bf16) hipLaunchKernelGGL((attn_decode_gqa_kernel<true, HD, REP, false>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((attn_decode_gqa_kernel<false, HD, REP, false>), grid, block, 0, s, a);
  if (a.ws && splits > 1) {
    if (bf16) hipLaunchKernelGGL((attn_merge_kernel<true, HD>), dim3(a.n_heads, batch), dim3(HD), 0, s, a.ws, a.out, a.n_heads, splits);
    else hipLaunchKernelGGL((attn_merge_kernel<false, HD>), dim3(a.n_heads, batch), dim3(HD), 0, s, a.ws, a.out, a.n_heads, splits);
  }
  return hipGetLastError();
}

template <int HD, int REP>
static hipError_t launch_attn_static_cfg(const AttnArgs& a, int batch, bool bf16, hipStream_t s) {
  dim3 grid(a.n_kv * (a.n_heads / a.n_kv / REP), batch), block(256);
  if (bf16) hipLaunchKernelGGL((attn_decode_gqa_kernel<true, HD, REP, true>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((attn_decode_gqa_kernel<false, HD, REP, true>), grid, block, 0, s, a);
  return hipGetLastError();
}

// softmax(q K^T * sm_scale) V over cache positions 0..*pos for one (already rotated) query token per sequence
hipError_t launch_attn_static(const void* q, const void* kcache, const void* vcache, const long long* pos, void* out, int batch,
                              int n_heads, int n_kv, int head_dim, int capacity, float sm_scale, bool bf16, hipStream_t s) {
  AttnArgs a{static_cast<const unsigned short*>(q), nullptr, nullptr,
             static_cast<unsigned short*>(const_cast<void*>(kcache)), static_cast<unsigned short*>(const_cast<void*>(vcache)),
             nullptr, pos, static_cast<unsigned short*>(out), n_heads, n_kv, capacity, 1.f, sm_scale, 0, nullptr, nullptr, 0.f, nullptr, nullptr, nullptr,
             nullptr};
  const int rep = n_heads / n_kv;
  const int R = rep % 4 == 0 ? 4 : rep % 3 == 0 ? 3 : rep % 2 == 0 ? 2 : 1;
  if (head_dim == 128) {
    switch (R) {
      case 4: return launch_attn_static_cfg<128, 4>(a, batch, bf16, s);
      case 3: return launch_attn_static_cfg<128, 3>(a, batch, bf16, s);
      case 2: return launch_attn_static_cfg<128, 2>(a, batch, bf16, s);
      default: return launch_attn_static_cfg<128, 1>(a, batch, bf16, s);
    }
  }
  if (head_dim == 64) {
    switch (R) {
      case 4: return launch_attn_static_cfg<64, 4>(a, batch, bf16, s);
      case 3: return launch_attn_static_cfg<64, 3>(a, batch, bf16, s);
      case 2: return launch_attn_static_cfg<64, 2>(a, batch, bf16, s);
      default: return launch_attn_static_cfg<64, 1>(a, batch, bf16, s);
    }
  }
  return hipErrorInvalidValue;
}

// Context splits of the grouped kernel (needs caller scratch): enough workgroups to fill the chip once the cache is long,
// 128+ positions per slice (measured: 1024 -> 646, 2048 -> 645 tok/s against 629 / 614 with 256).  The per-head batch-1 kernel gives a 2048-token context 32 workgroups that each re-read their
// kv head (measured: 456 tok/s at context 2048 against 773 at 128).
// Query heads per workgroup of a split launch: 2 (120 VGPRs, two workgroups per CU cover each other's dot / DPP / exp
// chains; K/V read once per pair) -- but 1 while the cache is short and the batch small, where the extra workgroups matter
// more than the re-read (context 900: 689 vs 654 tok/s; from 2048 on the pair wins: 650 vs 631, 563 vs 499 at 8192).
static int split_heads(int batch, int n_heads, int rep, int capacity) {
  if (rep % 2) return 1;
  return (capacity <= 1536 && batch * n_heads <= 64) ? 1 : 2;
}

int attn_splits(int batch, int n_heads, int n_kv, int capacity) {
  const int rep = n_heads / n_kv;
  static const int force_r = gbx::env_switch("GBX_ATTN_SPLIT_R");   // tuning aid
  const int R = force_r ? force_r : split_heads(batch, n_heads, rep, capacity);
  const int wgs = n_kv * (rep / R) * batch;
  static const int min_cap = gbx::env_switch("GBX_ATTN_SPLIT_MIN");   // tuning aid
  if (capacity < min_cap) return 1;                    // short caches: the un-split kernels win (649 vs 774 tok/s at context 128)
  static const int div = gbx::env_switch("GBX_ATTN_SPLIT_DIV");     // tuning aid
  int s = std::min(std::min(512 / std::max(wgs, 1), capacity / div), 32);
  return s < 2 ? 1 : s;
}

size_t attn_workspace_bytes(int batch, int n_heads, int n_kv, int head_dim, int capacity) {
  const int s = attn_splits(batch, n_heads, n_kv, capacity);
  return s > 1 ? static_cast<size_t>(batch) * n_heads * s * (head_dim + 2) * sizeof(float) : 0;
}

hipError_t launch_attn_decode(const void* q, const void* k, const void* v, void* kcache, void* vcache, const float* inv_freq,
                              const long long* pos, void* out, int batch, int n_heads, int n_kv, int head_dim, int capacity,
                              float rope_scale, float sm_scale, int traditional, bool bf16, hipStream_t s,
                              const void* q_norm_w, const void* k_norm_w, float norm_eps, const void* q_bias, const void* k_bias,
                              const void* v_bias, float* ws, const float* qkv_ws, int qkv_slices) {
  AttnArgs a{static_cast<const unsigned short*>(q), static_cast<const unsigned short*>(k), static_cast<const unsigned short*>(v),
             static_cast<unsigned short*>(kcache), static_cast<unsigned short*>(vcache), inv_freq, pos,
             static_cast<unsigned short*>(out), n_heads, n_kv, capacity, rope_scale, sm_scale, traditional,
             static_cast<const unsigned short*>(q_norm_w), static_cast<const unsigned short*>(k_norm_w), norm_eps,
             static_cast<const unsigned short*>(q_bias), static_cast<const unsigned short*>(k_bias),
             static_cast<const unsigned short*>(v_bias), nullptr};
#ifdef GBX_STAMPS
  a.stamps = stamp_region();
#endif
  const int rep = n_heads / n_kv;
  const int splits = ws ? attn_splits(batch, n_heads, n_kv, capacity) : 1;
  if (splits > 1) a.ws = ws;
  if (qkv_ws) {
    a.qkv_ws = qkv_ws; a.qkv_slices = qkv_slices;
    a.qkv_mn = static_cast<long long>(batch) * (n_heads + 2 * n_kv) * head_dim;
  }
  static const bool force_grouped = (gbx::env_switch("GBX_ATTN_GROUPED") != 0);    // tuning aid: grouped kernel at batch 1 too
  if (q_norm_w || q_bias || qkv_ws || (batch > 1 && rep > 1) || (force_grouped && rep > 1) || splits > 1) {   // grouped kernel: batched decode, or q/k norms (any batch)
    // 2 heads per workgroup when the context is split: 120 VGPRs -> two workgroups per CU, whose waves cover each other's
    // dependent dot / DPP / exp chains (with 4 heads, 166 VGPRs, one wave per SIMD ran the cache at 1.2 TB/s)
    static const int force_r = gbx::env_switch("GBX_ATTN_SPLIT_R");
    const int R = splits > 1 ? (force_r ? force_r : split_heads(batch, n_heads, rep, capacity)) : rep % 4 == 0 ? 4 : rep % 3 == 0 ? 3 : rep % 2 == 0 ? 2 : 1;
    if (head_dim == 128) {
      switch (R) {
        case 4: return launch_attn_gqa<128, 4>(a, batch, bf16, s, splits);
        case 3: return launch_attn_gqa<128, 3>(a, batch, bf16, s, splits);
        case 2: return launch_attn_gqa<128, 2>(a, batch, bf16, s, splits);
        default: return launch_attn_gqa<128, 1>(a, batch, bf16, s, splits);
      }
    }
    if (head_dim == 64) {
      switch (R) {
        case 4: return launch_attn_gqa<64, 4>(a, batch, bf16, s, splits);
        case 3: return launch_attn_gqa<64, 3>(a, batch, bf16, s, splits);
        case 2: return launch_attn_gqa<64, 2>(a, batch, bf16, s, splits);
        default: return launch_attn_gqa<64, 1>(a, batch, bf16, s, splits);
      }
    }
    return hipErrorInvalidValue;
  }
  dim3 grid(n_heads, batch), block(256);
  static const bool wide = env_switch("GBX_ATTN_WIDE") != 0;
  if (head_dim == 128 && wide) {
    if (bf16) hipLaunchKernelGGL((attn_decode_kernel<true, 128, 16>), grid, dim3(1024), 0, s, a);
    else hipLaunchKernelGGL((attn_decode_kernel<false, 128, 16>), grid, dim3(1024), 0, s, a);
  } else if (head_dim == 128) {
    if (bf16) hipLaunchKernelGGL((attn_decode_kernel<true, 128>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((attn_decode_kernel<false, 128>), grid, block, 0, s, a);
  } else if (head_dim == 64) {
    if (bf16) hipLaunchKernelGGL((attn_decode_kernel<true, 64>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((attn_decode_kernel<false, 64>), grid, block, 0, s, a);
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ dense GEMV (lm_head)
// y[n] = sum_k W[n,k] * x'[k], W row-major 16-bit, x' = x or RMSNorm(x).  One wave per row, lane l owns the 8 features
// [p*512 + 8l, +8) of pass p (one 16-byte non-temporal load per pass); activations stay in VGPRs as packed pairs and
// meet the weights in v_dot2_f32_bf16 / v_dot2_f32_f16 (fp32 accumulate).  HBM-bound: 2 bytes per weight.
struct DenseArgs {
  const unsigned short* x;       // [K]
  const unsigned short* w;       // [N, K]
  unsigned short* y;             // [N]
  const unsigned short* norm_w;  // [K] or null
  float eps;
  int N, K, rows_per_wave;
  unsigned long long* best;      // optional [GBX_ARGMAX_SLOTS] packed (value, index) maxima: fused greedy argmax
  int null_rows;              // measurement only (GBX_CHAIN_FLOOR=2): every row read is row 0 / 1
  int index_offset;           // vocabulary shard: global index of row 0 (argmax keys carry global indices)
};

// Greedy sampling fused into the head: every workgroup folds its rows' rounded logits into one of 64 sharded 64-bit
// keys (orderable value bits << 32 | ~index, so the LOWEST index wins ties like argmax does); the next token's embed
// kernel reduces the 64 keys.  No separate 128K-wide reduction launch, no host round trip.
constexpr int GBX_ARGMAX_SLOTS = 64;
__device__ __forceinline__ unsigned long long argmax_key(float v, int idx) {
  unsigned u = __float_as_uint(v);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return (static_cast<unsigned long long>(u) << 32) | static_cast<unsigned long long>(0xFFFFFFFFu - static_cast<unsigned>(idx));
}

template <bool BF16>
__device__ __forceinline__ float dot2_act(unsigned wbits, unsigned xbits, float acc) {
  if constexpr (BF16) {
    typedef short s2 __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(s2, wbits), __builtin_bit_cast(s2, xbits), acc, false);
  } else {
    return __builtin_amdgcn_fdot2(as_h2(wbits), as_h2(xbits), acc, false);
  }
}

template <bool BF16, int P>
__global__ __launch_bounds__(256) void dense_gemv_kernel(const DenseArgs a) {
  constexpr int D = 2;                  // rows in flight per wave (ring)
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = a.K, N = a.N;
  const int nchunks = K / 8;            // 16-byte chunks per row
  const int rpw = a.rows_per_wave;
  const int row_begin = (blockIdx.x * 4 + wave) * rpw;

  // activations (and norm weights) for this lane's chunks
  u32x4 xv[P];
  const unsigned short* nw = a.norm_w ? a.norm_w : a.x;
  u32x4 wv[P];
  float ssq = 0.f;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int c = p * 64 + lane;
    const int cc = min(c, nchunks - 1);
    xv[p] = *reinterpret_cast<const u32x4*>(a.x + static_cast<size_t>(cc) * 8);
    wv[p] = *reinterpret_cast<const u32x4*>(nw + static_cast<size_t>(cc) * 8);
  }
  auto issue = [&](u32x4 (&buf)[P], int row) {
    const unsigned short* wr = a.w + static_cast<size_t>(a.null_rows ? (row & 1) : min(row, N - 1)) * K;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int cc = min(p * 64 + lane, nchunks - 1);
      buf[p] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wr + static_cast<size_t>(cc) * 8));
    }
  };
  u32x4 ring[D][P];
#pragma unroll
  for (int d = 0; d < D; ++d) issue(ring[d], row_begin + d);

  if (a.norm_w) {      // fused final RMSNorm: every workgroup recomputes the statistics of the (tiny) vector
    if (wave == 0) {
      for (int c = lane; c < nchunks; c += 64) {
        const u32x4 t = *reinterpret_cast<const u32x4*>(a.x + static_cast<size_t>(c) * 8);
        const unsigned tw[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float lo = act_lo<BF16>(tw[j]), hi = act_hi<BF16>(tw[j]);
          ssq += lo * lo + hi * hi;
        }
      }
      ssq = wave_sum(ssq);
      if (lane == 0) red[0] = ssq;
    }
    __syncthreads();
    const float rstd = rsqrtf(red[0] / static_cast<float>(K) + a.eps);
#pragma unroll
    for (int p = 0; p < P; ++p) {
      unsigned xw[4] = {xv[p].x, xv[p].y, xv[p].z, xv[p].w};
      const unsigned ww[4] = {wv[p].x, wv[p].y, wv[p].z, wv[p].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned lo = f32_to_act<BF16>(act_lo<BF16>(xw[j]) * rstd * act_lo<BF16>(ww[j]));
        const unsigned hi = f32_to_act<BF16>(act_hi<BF16>(xw[j]) * rstd * act_hi<BF16>(ww[j]));
        xw[j] = lo | (hi << 16);
      }
      xv[p] = u32x4{xw[0], xw[1], xw[2], xw[3]};
    }
  }
#pragma unroll
  for (int p = 0; p < P; ++p)      // chunks past the end of the row contribute nothing
    if (p * 64 + lane >= nchunks) xv[p] = u32x4{0, 0, 0, 0};

  unsigned long long wbest = 0;
  auto consume = [&](const u32x4 (&buf)[P], int row) {
    float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      acc0 = dot2_act<BF16>(buf[p].x, xv[p].x, acc0);
      acc1 = dot2_act<BF16>(buf[p].y, xv[p].y, acc1);
      acc0 = dot2_act<BF16>(buf[p].z, xv[p].z, acc0);
      acc1 = dot2_act<BF16>(buf[p].w, xv[p].w, acc1);
    }
    const float tot = wave_sum(acc0 + acc1);
    if (lane == 0 && row < N) {
      const unsigned short o = f32_to_act<BF16>(tot);
      a.y[row] = o;
      const unsigned long long key = argmax_key(act_to_f32<BF16>(o), row + a.index_offset);
      wbest = key > wbest ? key : wbest;
    }
  };
  const int steps = rpw / D;
  int base = row_begin;
  for (int st = 1; st < steps; ++st) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      consume(ring[d], base + d);
      issue(ring[d], base + D + d);
    }
    base += D;
  }
#pragma unroll
  for (int d = 0; d < D; ++d) consume(ring[d], base + d);
  if (a.best) {                                  // one atomic per workgroup, spread over the sharded slots
    __shared__ unsigned long long bw[4];
    if (lane == 0) bw[wave] = wbest;
    __syncthreads();
    if (tid == 0) {
      unsigned long long b = bw[0];
#pragma unroll
      for (int i = 1; i < 4; ++i) b = bw[i] > b ? bw[i] : b;
      atomicMax(a.best + (blockIdx.x % GBX_ARGMAX_SLOTS), b);
    }
  }
}

// Start of a decode step on the device: reduce the previous step's argmax keys to the next token id, publish it,
// advance the position, clear the keys, and copy that token's embedding row into the residual stream
// (embed_tokens lookup, gbx_lm/models/qllama.py:164).  `mode` 0: token comes from *tok_io (teacher forcing / sampled on
// host), 1: token comes from the keys (greedy chain, no host involvement).
template <bool BF16>
__global__ __launch_bounds__(256) void embed_next_kernel(unsigned long long* best, long long* tok_io, long long* pos,
                                                         const unsigned short* __restrict__ table, unsigned short* __restrict__ h,
                                                         int D, int vocab, int mode, int advance, int nkeys, unsigned long long* also_clear) {
  __shared__ long long tok_s;
  const int tid = threadIdx.x;
  if (tid < 64) {
    // nkeys = 64 (one GPU) or 64 * world (vocabulary-sharded head: every rank's keys, gathered): lane-wise maximum first
    unsigned long long k = 0ull;
    for (int i = tid; i < nkeys; i += 64) { const unsigned long long v = best[i]; k = v > k ? v : k; best[i] = 0ull; }
    if (also_clear) also_clear[tid] = 0ull;
    // 64-lane max of a 64-bit key via two 32-bit stages (hi word, then lo word among the winners)
    const unsigned hi = static_cast<unsigned>(k >> 32);
    const unsigned mhi = wave_max_u32(hi);
    const unsigned lo = (hi == mhi) ? static_cast<unsigned>(k & 0xFFFFFFFFu) : 0u;
    const unsigned mlo = wave_max_u32(lo);
    if (tid == 0) {
      long long t = (mode == 1) ? static_cast<long long>(0xFFFFFFFFu - mlo) : *tok_io;
      t = t < 0 ? 0 : (t >= vocab ? vocab - 1 : t);
      tok_s = t;
      *tok_io = t;
      if (advance) *pos = *pos + 1;
    }
  }
  __syncthreads();
  const long long t = tok_s;
  const u32x4* src = reinterpret_cast<const u32x4*>(table + static_cast<size_t>(t) * D);
  u32x4* dst = reinterpret_cast<u32x4*>(h);
  for (int i = tid; i < D / 8; i += 256) dst[i] = src[i];
}

hipError_t launch_embed_next(void* best, void* tok_io, void* pos, const void* table, void* h, int D, int vocab, int mode,
                             int advance, bool bf16, hipStream_t s, int nkeys, void* also_clear) {
  auto B = static_cast<unsigned long long*>(best);
  auto T = static_cast<long long*>(tok_io);
  auto P = static_cast<long long*>(pos);
  auto E = static_cast<const unsigned short*>(table);
  auto H = static_cast<unsigned short*>(h);
  auto C = static_cast<unsigned long long*>(also_clear);
  if (bf16) hipLaunchKernelGGL((embed_next_kernel<true>), dim3(1), dim3(256), 0, s, B, T, P, E, H, D, vocab, mode, advance, nkeys, C);
  else hipLaunchKernelGGL((embed_next_kernel<false>), dim3(1), dim3(256), 0, s, B, T, P, E, H, D, vocab, mode, advance, nkeys, C);
  return hipGetLastError();
}

hipError_t launch_dense_gemv(const void* x, const void* w, void* y, const void* norm_w, float eps, int N, int K, void* best,
                             bool bf16, hipStream_t s, int index_offset) {
  DenseArgs a{static_cast<const unsigned short*>(x), static_cast<const unsigned short*>(w), static_cast<unsigned short*>(y),
              static_cast<const unsigned short*>(norm_w), eps, N, K, 0, static_cast<unsigned long long*>(best)};
  const int owners = 256 * 8 * 4;                       // waves when every workgroup is resident (8 per CU)
  int rpw = (N + owners - 1) / owners;
  rpw = ((rpw + 1) / 2) * 2;
  if (rpw < 2) rpw = 2;
  a.rows_per_wave = rpw;
  const int blocks = (N + 4 * rpw - 1) / (4 * rpw);
  static const int chain_floor = gbx::env_switch("GBX_CHAIN_FLOOR");      // measurement only: see gemv_impl.h
  if (chain_floor == 1) a.rows_per_wave = 2;
  a.null_rows = chain_floor == 2 ? 1 : 0;
  a.index_offset = index_offset;
  const int P = (K / 8 + 63) / 64;
  dim3 grid(blocks), block(256);
#define GBX_DENSE(PP)                                                                                  \
  if (bf16) hipLaunchKernelGGL((dense_gemv_kernel<true, PP>), grid, block, 0, s, a);                   \
  else hipLaunchKernelGGL((dense_gemv_kernel<false, PP>), grid, block, 0, s, a);
  if (P <= 1) { GBX_DENSE(1) }
  else if (P <= 2) { GBX_DENSE(2) }
  else if (P <= 4) { GBX_DENSE(4) }
  else if (P <= 6) { GBX_DENSE(6) }
  else if (P <= 8) { GBX_DENSE(8) }
  else if (P <= 16) { GBX_DENSE(16) }
  else return hipErrorInvalidValue;
#undef GBX_DENSE
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ logprobs + greedy token
// The tail of a generate_step (gbx_lm/utils.py:305-306,285): logprobs = logits - logsumexp(logits) and, for the default
// sampler, token = argmax(logprobs) -- two launches over the vocabulary instead of torch's reductions (~100 us per token
// at V = 128 K): (1) per-workgroup (max, sum exp) partials, (2) every workgroup folds the partials into the fp32
// logsumexp (one partial per thread, reduced through LDS), rounds it to the activation dtype like `.astype(logits.dtype)`, writes
// its slice of logprobs (activation-dtype subtraction) and bids its slice's best (value, lowest index) into one 64-bit key; the
// workgroup that finishes LAST turns the key into the int64 token and -- for generate_step, which sends the token id and the decode
// runner's status words to the host after every step -- packs (token, status words) into one buffer: one copy instead of three.
constexpr int LP_WGS = 128;

template <bool BF16>
__global__ __launch_bounds__(256) void logprob_partials_kernel(const unsigned short* __restrict__ logits, float2* __restrict__ part,
                                                               unsigned long long* __restrict__ key, int V) {
  __shared__ float red[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (blockIdx.x == 0 && tid == 0) { key[0] = 0ull; key[1] = 0ull; }          // the bid and the finished-workgroup count of launch 2
  const int per = (V + LP_WGS - 1) / LP_WGS, lo = blockIdx.x * per, hi = min(lo + per, V);
  float m = -3.0e38f;
  for (int i = lo + tid; i < hi; i += 256) m = fmaxf(m, act_to_f32<BF16>(logits[i]));
  m = key2f(wave_max_u32(f2key(m)));
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
  for (int i = lo + tid; i < hi; i += 256) sum += __expf(act_to_f32<BF16>(logits[i]) - m);
  sum = wave_sum(sum);
  if (lane == 0) red[4 + wave] = sum;
  __syncthreads();
  if (tid == 0) part[blockIdx.x] = float2{m, (red[4] + red[5]) + (red[6] + red[7])};
}

struct LogprobTail {
  long long* token;                    // int64 [1]
  long long* packed;                   // optional int64 [4]: token, status words 0..2 (0 where absent)
  const unsigned* status[3];           // optional device words (the decode runner's fused launches)
};

template <bool BF16>
__global__ __launch_bounds__(256) void logprob_write_kernel(const unsigned short* __restrict__ logits, const float2* __restrict__ part,
                                                            unsigned short* __restrict__ logprobs, unsigned long long* __restrict__ key,
                                                            int V, const LogprobTail tail) {
  __shared__ unsigned long long best[4];
  __shared__ float redm[4], reds[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // fold the partials: thread g < 128 holds pair g (every workgroup repeats it).  The ORDER of the sum differs from a serial fold by
  // rounding only; every workgroup folds in the same order, so all of them subtract the same logsumexp.
  static_assert(LP_WGS <= 256, "one partial per thread");
  const float2 pr = tid < LP_WGS ? part[tid] : float2{-3.0e38f, 0.f};
  float M = key2f(wave_max_u32(f2key(pr.x)));
  if (lane == 0) redm[wave] = M;
  __syncthreads();
  M = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));
  float S = wave_sum(tid < LP_WGS ? pr.y * __expf(pr.x - M) : 0.f);
  if (lane == 0) reds[wave] = S;
  __syncthreads();
  S = (reds[0] + reds[1]) + (reds[2] + reds[3]);
  const float lse = act_to_f32<BF16>(f32_to_act<BF16>(M + __logf(S)));
  const int per = (V + LP_WGS - 1) / LP_WGS, lo = blockIdx.x * per, hi = min(lo + per, V);
  unsigned long long k = 0ull;
  for (int i = lo + tid; i < hi; i += 256) {
    const unsigned short o = f32_to_act<BF16>(act_to_f32<BF16>(logits[i]) - lse);
    logprobs[i] = o;
    const unsigned long long c = (static_cast<unsigned long long>(f2key(act_to_f32<BF16>(o))) << 32) | static_cast<unsigned>(~i);
    k = c > k ? c : k;                                // larger value first, then the smaller index
  }
  // wave max of a 64-bit key: high word first, then the low word among the lanes that hold that high word
  const unsigned hi32 = wave_max_u32(static_cast<unsigned>(k >> 32));
  const unsigned lo32 = wave_max_u32(static_cast<unsigned>(k >> 32) == hi32 ? static_cast<unsigned>(k) : 0u);
  if (lane == 0) best[wave] = (static_cast<unsigned long long>(hi32) << 32) | lo32;
  __syncthreads();
  if (tid == 0) {
    unsigned long long b = best[0];
    for (int w = 1; w < 4; ++w) b = best[w] > b ? best[w] : b;
    if (hi > lo) atomicMax(key, b);
    // (both atomics are performed at the device's memory side, in program order of this thread: the workgroup whose count comes back
    //  LP_WGS - 1 reads a key that holds every bid)
    __threadfence();
    if (atomicAdd(key + 1, 1ull) == static_cast<unsigned long long>(LP_WGS - 1)) {
      const unsigned long long kk = __hip_atomic_load(key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const long long t = static_cast<long long>(~static_cast<unsigned>(kk));
      *tail.token = t;
      if (tail.packed) {
        tail.packed[0] = t;
#pragma unroll
        for (int q = 0; q < 3; ++q) tail.packed[1 + q] = tail.status[q] ? static_cast<long long>(*tail.status[q]) : 0ll;
      }
    }
  }
}

hipError_t launch_logprobs_argmax(const void* logits, void* logprobs, void* partials, unsigned long long* key, long long* token,
                                  int V, bool bf16, hipStream_t s, long long* packed, const unsigned* const* status) {
  auto L = static_cast<const unsigned short*>(logits);
  auto O = static_cast<unsigned short*>(logprobs);
  auto P = static_cast<float2*>(partials);
  LogprobTail tail{token, packed, {status ? status[0] : nullptr, status ? status[1] : nullptr, status ? status[2] : nullptr}};
  if (bf16) {
    hipLaunchKernelGGL((logprob_partials_kernel<true>), dim3(LP_WGS), dim3(256), 0, s, L, P, key, V);
    hipLaunchKernelGGL((logprob_write_kernel<true>), dim3(LP_WGS), dim3(256), 0, s, L, P, O, key, V, tail);
  } else {
    hipLaunchKernelGGL((logprob_partials_kernel<false>), dim3(LP_WGS), dim3(256), 0, s, L, P, key, V);
    hipLaunchKernelGGL((logprob_write_kernel<false>), dim3(LP_WGS), dim3(256), 0, s, L, P, O, key, V, tail);
  }
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ RMSNorm (standalone)
template <bool BF16>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const unsigned short* __restrict__ x, const unsigned short* __restrict__ w,
                                                      unsigned short* __restrict__ y, int K, float eps) {
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned short* xr = x + static_cast<size_t>(blockIdx.x) * K;
  unsigned short* yr = y + static_cast<size_t>(blockIdx.x) * K;
  float ssq = 0.f;
  for (int i = tid; i < K; i += 256) { const float v = act_to_f32<BF16>(xr[i]); ssq += v * v; }
  ssq = wave_sum(ssq);
  if (lane == 0) red[wave] = ssq;
  __syncthreads();
  const float rstd = rsqrtf(((red[0] + red[1]) + (red[2] + red[3])) / static_cast<float>(K) + eps);
  for (int i = tid; i < K; i += 256) yr[i] = f32_to_act<BF16>(act_to_f32<BF16>(xr[i]) * rstd * act_to_f32<BF16>(w[i]));
}

typedef float f32x4s __attribute__((ext_vector_type(4)));

// Residual add + RMSNorm in one pass (batched decode glue): h <- round(h + d) (the activation-dtype add the model graph
// does, qllama.py:138-140), y = RMSNorm(h) * w.  h is updated in place.
// Rows up to 8192 values stay in registers between the two passes: 16-byte loads, one barrier.
// ADD == 2: the addend is the sum of `slices` fp32 partial tiles ws[s][M][K] of a k-split GEMM, rounded once to the
// activation dtype exactly like the summing launch it replaces (batch_finish_kernel: v = ws[0]; v += ws[s], s = 1..).
template <bool BF16, int ADD, int NT = 256>
__global__ __launch_bounds__(NT) void rmsnorm_vec_kernel(unsigned short* __restrict__ h, const unsigned short* __restrict__ d,
                                                          const unsigned short* __restrict__ w, unsigned short* __restrict__ y,
                                                          int K, float eps, const float* __restrict__ ws = nullptr, int slices = 0,
                                                          long long mn = 0) {
  __shared__ float red[NT / 64];
  constexpr int ITERS = 1024 / NT;                  // 16-byte pieces per thread (K <= 8192)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t row = static_cast<size_t>(blockIdx.x) * K;
  const int n8 = K >> 3;                            // 16-byte pieces per row (K % 8 == 0, K <= 8192: at most 4 per thread)
  u32x4 v[ITERS];
  float ssq = 0.f;
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int p = it * NT + tid;
    if (p < n8) {
      u32x4 hv = reinterpret_cast<const u32x4*>(h + row)[p];
      if constexpr (ADD != 0) {
        u32x4 dv;
        if constexpr (ADD == 2) {
          // every slice of a batch of 8 is REQUESTED before the first add (a loop of load / add exposes one memory latency
          // per slice: 5.7 us for the 7 slices of a down projection); the adds keep the order v = ws[0]; v += ws[s], s = 1..
          const float* src = ws + row + static_cast<size_t>(p) * 8;
          f32x4s a0 = *reinterpret_cast<const f32x4s*>(src), a1 = *reinterpret_cast<const f32x4s*>(src + 4);
          for (int sl = 1; sl < slices; sl += 8) {
            f32x4s t0[8], t1[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const float* sp = src + static_cast<size_t>(min(sl + j, slices - 1)) * mn;     // clamped: no branch around a load
              t0[j] = *reinterpret_cast<const f32x4s*>(sp);
              t1[j] = *reinterpret_cast<const f32x4s*>(sp + 4);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
              if (sl + j < slices) { a0 += t0[j]; a1 += t1[j]; }
          }
          dv = u32x4{static_cast<unsigned>(f32_to_act<BF16>(a0.x)) | (static_cast<unsigned>(f32_to_act<BF16>(a0.y)) << 16),
                     static_cast<unsigned>(f32_to_act<BF16>(a0.z)) | (static_cast<unsigned>(f32_to_act<BF16>(a0.w)) << 16),
                     static_cast<unsigned>(f32_to_act<BF16>(a1.x)) | (static_cast<unsigned>(f32_to_act<BF16>(a1.y)) << 16),
                     static_cast<unsigned>(f32_to_act<BF16>(a1.z)) | (static_cast<unsigned>(f32_to_act<BF16>(a1.w)) << 16)};
        } else {
          dv = reinterpret_cast<const u32x4*>(d + row)[p];
        }
        const unsigned hw[4] = {hv.x, hv.y, hv.z, hv.w}, dw[4] = {dv.x, dv.y, dv.z, dv.w};
        unsigned o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          o[j] = static_cast<unsigned>(f32_to_act<BF16>(act_lo<BF16>(hw[j]) + act_lo<BF16>(dw[j]))) |
                 (static_cast<unsigned>(f32_to_act<BF16>(act_hi<BF16>(hw[j]) + act_hi<BF16>(dw[j]))) << 16);
        hv = u32x4{o[0], o[1], o[2], o[3]};
        reinterpret_cast<u32x4*>(h + row)[p] = hv;
      }
      v[it] = hv;
      const unsigned hw[4] = {hv.x, hv.y, hv.z, hv.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float a0 = act_lo<BF16>(hw[j]), a1 = act_hi<BF16>(hw[j]); ssq = fmaf(a0, a0, fmaf(a1, a1, ssq)); }
    }
  }
  u32x4 wq[ITERS];                                      // the norm weights are requested before the reduction, not after it
#pragma unroll
  for (int it = 0; it < ITERS; ++it) wq[it] = reinterpret_cast<const u32x4*>(w)[min(it * NT + tid, n8 - 1)];
  ssq = wave_sum(ssq);
  if (lane == 63) red[wave] = ssq;
  __syncthreads();
  float tot = (red[0] + red[1]) + (red[2] + red[3]);
  if constexpr (NT == 512) tot += (red[4] + red[5]) + (red[6] + red[7]);
  const float rstd = rsqrtf(tot / static_cast<float>(K) + eps);
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int p = it * NT + tid;
    if (p < n8) {
      const u32x4 wv = wq[it];
      const unsigned hw[4] = {v[it].x, v[it].y, v[it].z, v[it].w}, ww[4] = {wv.x, wv.y, wv.z, wv.w};
      unsigned o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        o[j] = static_cast<unsigned>(f32_to_act<BF16>(act_lo<BF16>(hw[j]) * rstd * act_lo<BF16>(ww[j]))) |
               (static_cast<unsigned>(f32_to_act<BF16>(act_hi<BF16>(hw[j]) * rstd * act_hi<BF16>(ww[j]))) << 16);
      reinterpret_cast<u32x4*>(y + row)[p] = u32x4{o[0], o[1], o[2], o[3]};
    }
  }
}

template <bool BF16>
__global__ __launch_bounds__(256) void add_rmsnorm_kernel(unsigned short* __restrict__ h, const unsigned short* __restrict__ d,
                                                          const unsigned short* __restrict__ w, unsigned short* __restrict__ y,
                                                          int K, float eps) {
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned short* hr = h + static_cast<size_t>(blockIdx.x) * K;
  const unsigned short* dr = d + static_cast<size_t>(blockIdx.x) * K;
  unsigned short* yr = y + static_cast<size_t>(blockIdx.x) * K;
  float ssq = 0.f;
  for (int i = tid; i < K; i += 256) {
    const unsigned short sb = f32_to_act<BF16>(act_to_f32<BF16>(hr[i]) + act_to_f32<BF16>(dr[i]));
    hr[i] = sb;
    const float v = act_to_f32<BF16>(sb);
    ssq += v * v;
  }
  ssq = wave_sum(ssq);
  if (lane == 63) red[wave] = ssq;
  __syncthreads();
  const float rstd = rsqrtf(((red[0] + red[1]) + (red[2] + red[3])) / static_cast<float>(K) + eps);
  for (int i = tid; i < K; i += 256) yr[i] = f32_to_act<BF16>(act_to_f32<BF16>(hr[i]) * rstd * act_to_f32<BF16>(w[i]));
}

static bool vec_ok(const void* a, const void* b, const void* c, const void* d, int K) {
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  return K % 8 == 0 && K <= 8192 && al(a) && al(b) && al(c) && al(d);
}

hipError_t launch_add_rmsnorm(void* h, const void* d, const void* w, void* y, int M, int K, float eps, bool bf16, hipStream_t s) {
  auto H = static_cast<unsigned short*>(h);
  auto D = static_cast<const unsigned short*>(d);
  auto W = static_cast<const unsigned short*>(w);
  auto Y = static_cast<unsigned short*>(y);
  if (vec_ok(h, d, w, y, K)) {
    if (bf16) hipLaunchKernelGGL((rmsnorm_vec_kernel<true, 1>), dim3(M), dim3(256), 0, s, H, D, W, Y, K, eps);
    else hipLaunchKernelGGL((rmsnorm_vec_kernel<false, 1>), dim3(M), dim3(256), 0, s, H, D, W, Y, K, eps);
    return hipGetLastError();
  }
  if (bf16) hipLaunchKernelGGL((add_rmsnorm_kernel<true>), dim3(M), dim3(256), 0, s, H, D, W, Y, K, eps);
  else hipLaunchKernelGGL((add_rmsnorm_kernel<false>), dim3(M), dim3(256), 0, s, H, D, W, Y, K, eps);
  return hipGetLastError();
}

// h += round(sum of the fp32 slices of a k-split GEMM), y = RMSNorm(h) * w: the summing launch of the down projection folded
// into the next block's add + norm (K % 8 == 0, K <= 8192, 16-byte aligned rows: the callers' decode shapes)
hipError_t launch_add_rmsnorm_slices(void* h, const float* ws, int slices, const void* w, void* y, int M, int K, float eps, bool bf16,
                                     hipStream_t s) {
  auto H = static_cast<unsigned short*>(h);
  auto W = static_cast<const unsigned short*>(w);
  auto Y = static_cast<unsigned short*>(y);
  const long long mn = static_cast<long long>(M) * K;
  // 512 threads (one 16-byte piece of a <= 4096-value row each): the launch is a chain of dependent loads per thread, so the
  // fewer pieces a thread walks the shorter it is (B = 32 decode: 4.9 -> see DESIGN 3.4)
  static const bool wide = env_switch("GBX_NORM_WIDE") != 0;      // default 0: measured 2.60 -> 2.57 ms per B = 32 step (noise level), and the
  if (wide && K > 2048) {                                         // 256-thread form is bit-identical to gbx_add_rmsnorm on summed slices
    if (bf16) hipLaunchKernelGGL((rmsnorm_vec_kernel<true, 2, 512>), dim3(M), dim3(512), 0, s, H, nullptr, W, Y, K, eps, ws, slices, mn);
    else hipLaunchKernelGGL((rmsnorm_vec_kernel<false, 2, 512>), dim3(M), dim3(512), 0, s, H, nullptr, W, Y, K, eps, ws, slices, mn);
    return hipGetLastError();
  }
  if (bf16) hipLaunchKernelGGL((rmsnorm_vec_kernel<true, 2>), dim3(M), dim3(256), 0, s, H, nullptr, W, Y, K, eps, ws, slices, mn);
  else hipLaunchKernelGGL((rmsnorm_vec_kernel<false, 2>), dim3(M), dim3(256), 0, s, H, nullptr, W, Y, K, eps, ws, slices, mn);
  return hipGetLastError();
}

// The same launch spread over the whole chip (B = 32 decode: the one-workgroup-per-row form above pulls 8 slices x 16 KB = 128 KB
// through ONE CU per row -- 4.9 us at that CU's fill rate -- and 224 of the 256 CUs idle).  A row is cut into K / 512 segments of 512
// values, one single-wave workgroup each (8 values per lane); the row's statistic is the sum of the segments' partials, exchanged
// through the L2 of ONE XCD: the workgroups of a row are placed on the same XCD by construction (group = HW_REG_XCC_ID, member =
// blockIdx / 8, row = group + 8 * (member / segments): the fused attention launch's scheme, dattn_impl.h), so the partials leave as
// workgroup-scope {tag, value} granule stores (write-through to that L2) and are polled with agent-scope loads (served by that L2).
// A placement that is not a bijection leaves a row short of a segment and the launch raises its status word (bounded polls) instead
// of returning a wrong row.  Tags come from a generation word per (group, row slot) that segment 0 advances when it has the row's
// partials (every segment has read the word by then).  h is bit-identical to the launch above (same slice order, same roundings); the
// statistic is summed in a different order (segment partials), so y may differ from it in the last place.
constexpr int SN_STATUS_WORD = 256 * 16;                  // u32 index: gen words [256][16], then the status word
constexpr int SN_HAND_BYTES = (256 * 16 + 16) * 4;        // granules [256 row slots][8 segments] start here
constexpr unsigned SN_SPIN_LIMIT = 1u << 22;
size_t slices_norm_state_bytes() { return static_cast<size_t>(SN_HAND_BYTES) + 256 * 8 * 8; }
bool slices_norm_xcd_supported(int M, int K) {
  if (!xcd_layout_ok()) return false;          // a row's segments meet in ONE XCD's L2: 8 XCDs, round-robin placement (xcd_layout.hip)
  if (K % 512 || K / 512 > 8 || K < 512) return false;
  const int S = K / 512;
  return M >= 1 && M <= 8 * (32 / S);
}
typedef __attribute__((address_space(1))) unsigned long long sn_gu64;
typedef __attribute__((address_space(1))) unsigned sn_gu32;

template <bool BF16>
__global__ __launch_bounds__(64) void slices_norm_xcd_kernel(unsigned short* __restrict__ h, const float* __restrict__ ws, int slices, long long mn,
                                                             const unsigned short* __restrict__ w, unsigned short* __restrict__ y, int M, int K,
                                                             float eps, unsigned* state) {
  const int lane = threadIdx.x;
  const int g = static_cast<int>(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u);       // HW_REG_XCC_ID
  const int j = static_cast<int>(blockIdx.x) >> 3;
  const int S = K >> 9;
  const int rs = j / S, seg = j - rs * S;
  const int row = g + 8 * rs;
  if (row >= M || rs >= 32 / S) return;                       // (whole rows: every segment of a row takes the same branch)
  sn_gu32* gen_p = reinterpret_cast<sn_gu32*>(reinterpret_cast<uintptr_t>(state)) + (g * 32 + rs) * 16;
  sn_gu64* hand = reinterpret_cast<sn_gu64*>(reinterpret_cast<uintptr_t>(state) + SN_HAND_BYTES) + (g * 32 + rs) * 8;
  const unsigned gen = __hip_atomic_load(gen_p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const size_t off = static_cast<size_t>(row) * K + static_cast<size_t>(seg * 64 + lane) * 8;
  // every slice is requested before the first add; the adds keep the order v = ws[0]; v += ws[s], s = 1.. (batch_finish_kernel)
  const float* src = ws + off;
  f32x4s a0 = *reinterpret_cast<const f32x4s*>(src), a1 = *reinterpret_cast<const f32x4s*>(src + 4);
  const u32x4 hv0 = *reinterpret_cast<const u32x4*>(h + off);
  const u32x4 wv = *reinterpret_cast<const u32x4*>(w + static_cast<size_t>(seg * 64 + lane) * 8);
  for (int sl = 1; sl < slices; sl += 8) {
    f32x4s t0[8], t1[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float* sp = src + static_cast<size_t>(min(sl + i, slices - 1)) * mn;
      t0[i] = *reinterpret_cast<const f32x4s*>(sp);
      t1[i] = *reinterpret_cast<const f32x4s*>(sp + 4);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (sl + i < slices) { a0 += t0[i]; a1 += t1[i]; }
  }
  const unsigned dw[4] = {static_cast<unsigned>(f32_to_act<BF16>(a0.x)) | (static_cast<unsigned>(f32_to_act<BF16>(a0.y)) << 16),
                          static_cast<unsigned>(f32_to_act<BF16>(a0.z)) | (static_cast<unsigned>(f32_to_act<BF16>(a0.w)) << 16),
                          static_cast<unsigned>(f32_to_act<BF16>(a1.x)) | (static_cast<unsigned>(f32_to_act<BF16>(a1.y)) << 16),
                          static_cast<unsigned>(f32_to_act<BF16>(a1.z)) | (static_cast<unsigned>(f32_to_act<BF16>(a1.w)) << 16)};
  const unsigned hw[4] = {hv0.x, hv0.y, hv0.z, hv0.w};
  unsigned o[4];
  float ssq = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    o[i] = static_cast<unsigned>(f32_to_act<BF16>(act_lo<BF16>(hw[i]) + act_lo<BF16>(dw[i]))) |
           (static_cast<unsigned>(f32_to_act<BF16>(act_hi<BF16>(hw[i]) + act_hi<BF16>(dw[i]))) << 16);
    const float v0 = act_lo<BF16>(o[i]), v1 = act_hi<BF16>(o[i]);
    ssq = fmaf(v0, v0, fmaf(v1, v1, ssq));
  }
  *reinterpret_cast<u32x4*>(h + off) = u32x4{o[0], o[1], o[2], o[3]};
  ssq = wave_sum(ssq);                                        // (wave-uniform)
  unsigned tag = gen + 1u;
  if (tag == 0u) tag = 1u;
  if (lane == 0)
    __hip_atomic_store(hand + seg, (static_cast<unsigned long long>(tag) << 32) | __float_as_uint(ssq), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  unsigned long long gv;
  for (unsigned spins = 0;;) {
    gv = __hip_atomic_load(hand + min(lane, S - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__all(static_cast<unsigned>(gv >> 32) == tag)) break;
    if (++spins > SN_SPIN_LIMIT) {
      if (lane == 0) atomicOr(state + SN_STATUS_WORD, 1u);
      break;
    }
    __builtin_amdgcn_s_sleep(1);
  }
  float tot = 0.f;
  for (int i = 0; i < S; ++i) tot += __uint_as_float(static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(static_cast<unsigned>(gv)), i)));
  const float rstd = rsqrtf(tot / static_cast<float>(K) + eps);
  const unsigned ww[4] = {wv.x, wv.y, wv.z, wv.w};
  unsigned yo[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
    yo[i] = static_cast<unsigned>(f32_to_act<BF16>(act_lo<BF16>(o[i]) * rstd * act_lo<BF16>(ww[i]))) |
            (static_cast<unsigned>(f32_to_act<BF16>(act_hi<BF16>(o[i]) * rstd * act_hi<BF16>(ww[i]))) << 16);
  *reinterpret_cast<u32x4*>(y + off) = u32x4{yo[0], yo[1], yo[2], yo[3]};
  if (seg == 0 && lane == 0) __hip_atomic_store(gen_p, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

hipError_t launch_add_rmsnorm_slices_xcd(void* h, const float* ws, int slices, const void* w, void* y, int M, int K, float eps, bool bf16,
                                         void* state, hipStream_t s) {
  auto H = static_cast<unsigned short*>(h);
  auto W = static_cast<const unsigned short*>(w);
  auto Y = static_cast<unsigned short*>(y);
  const long long mn = static_cast<long long>(M) * K;
  if (bf16) hipLaunchKernelGGL((slices_norm_xcd_kernel<true>), dim3(256), dim3(64), 0, s, H, ws, slices, mn, W, Y, M, K, eps, static_cast<unsigned*>(state));
  else hipLaunchKernelGGL((slices_norm_xcd_kernel<false>), dim3(256), dim3(64), 0, s, H, ws, slices, mn, W, Y, M, K, eps, static_cast<unsigned*>(state));
  return hipGetLastError();
}

// SwiGLU glue for batched decode: y = round(round(silu(g)) * u), the two roundings of `nn.silu(gate) * up` in the
// activation dtype (qllama.py:115).  8 values (16 bytes) per thread.
template <bool BF16>
__global__ __launch_bounds__(256) void swiglu_kernel(const unsigned short* __restrict__ g, const unsigned short* __restrict__ u,
                                                     unsigned short* __restrict__ y, long long n8) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n8; i += static_cast<long long>(gridDim.x) * 256) {
    const u32x4 gv = reinterpret_cast<const u32x4*>(g)[i], uv = reinterpret_cast<const u32x4*>(u)[i];
    const unsigned gw[4] = {gv.x, gv.y, gv.z, gv.w}, uw[4] = {uv.x, uv.y, uv.z, uv.w};
    unsigned o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float g0 = act_lo<BF16>(gw[j]), g1 = act_hi<BF16>(gw[j]);
      const float s0 = act_to_f32<BF16>(f32_to_act<BF16>(g0 / (1.f + __expf(-g0))));
      const float s1 = act_to_f32<BF16>(f32_to_act<BF16>(g1 / (1.f + __expf(-g1))));
      o[j] = static_cast<unsigned>(f32_to_act<BF16>(s0 * act_lo<BF16>(uw[j]))) |
             (static_cast<unsigned>(f32_to_act<BF16>(s1 * act_hi<BF16>(uw[j]))) << 16);
    }
    reinterpret_cast<u32x4*>(y)[i] = u32x4{o[0], o[1], o[2], o[3]};
  }
}

hipError_t launch_swiglu(const void* g, const void* u, void* y, long long n, bool bf16, hipStream_t s) {
  const long long n8 = n / 8;
  const unsigned blocks = static_cast<unsigned>(std::min<long long>((n8 + 255) / 256, 4096));
  auto G = static_cast<const unsigned short*>(g);
  auto U = static_cast<const unsigned short*>(u);
  auto Y = static_cast<unsigned short*>(y);
  if (bf16) hipLaunchKernelGGL((swiglu_kernel<true>), dim3(blocks), dim3(256), 0, s, G, U, Y, n8);
  else hipLaunchKernelGGL((swiglu_kernel<false>), dim3(blocks), dim3(256), 0, s, G, U, Y, n8);
  return hipGetLastError();
}

// short rows (per-head q / k norms, K = head_dim): one wavefront per row, four rows per workgroup
template <bool BF16>
__global__ __launch_bounds__(256) void rmsnorm_rows_kernel(const unsigned short* __restrict__ x, const unsigned short* __restrict__ w,
                                                           unsigned short* __restrict__ y, long long M, int K, float eps) {
  const int lane = threadIdx.x & 63;
  const long long row = static_cast<long long>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const unsigned short* xr = x + row * K;
  unsigned short* yr = y + row * K;
  float ssq = 0.f;
  for (int i = lane; i < K; i += 64) { const float v = act_to_f32<BF16>(xr[i]); ssq += v * v; }
  ssq = wave_sum(ssq);
  ssq = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(ssq), 63));
  const float rstd = rsqrtf(ssq / static_cast<float>(K) + eps);
  for (int i = lane; i < K; i += 64) yr[i] = f32_to_act<BF16>(act_to_f32<BF16>(xr[i]) * rstd * act_to_f32<BF16>(w[i]));
}

hipError_t launch_rmsnorm(const void* x, const void* w, void* y, int M, int K, float eps, bool bf16, hipStream_t s) {
  auto X = static_cast<const unsigned short*>(x);
  auto W = static_cast<const unsigned short*>(w);
  auto Y = static_cast<unsigned short*>(y);
  if (K <= 512) {
    const unsigned blocks = static_cast<unsigned>((static_cast<long long>(M) + 3) / 4);
    if (bf16) hipLaunchKernelGGL((rmsnorm_rows_kernel<true>), dim3(blocks), dim3(256), 0, s, X, W, Y, static_cast<long long>(M), K, eps);
    else hipLaunchKernelGGL((rmsnorm_rows_kernel<false>), dim3(blocks), dim3(256), 0, s, X, W, Y, static_cast<long long>(M), K, eps);
    return hipGetLastError();
  }
  if (vec_ok(x, x, w, y, K)) {                      // one pass, rows in registers
    auto Xm = const_cast<unsigned short*>(X);
    if (bf16) hipLaunchKernelGGL((rmsnorm_vec_kernel<true, 0>), dim3(M), dim3(256), 0, s, Xm, X, W, Y, K, eps);
    else hipLaunchKernelGGL((rmsnorm_vec_kernel<false, 0>), dim3(M), dim3(256), 0, s, Xm, X, W, Y, K, eps);
    return hipGetLastError();
  }
  if (bf16) hipLaunchKernelGGL((rmsnorm_kernel<true>), dim3(M), dim3(256), 0, s, X, W, Y, K, eps);
  else hipLaunchKernelGGL((rmsnorm_kernel<false>), dim3(M), dim3(256), 0, s, X, W, Y, K, eps);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ RoPE on [rows, L, D]
// mx.fast.rope semantics for any family (gbx_lm/models/rope_utils.py, su_rope.py, qdeepseek_v3.py:111-125): element pair i of
// the first `dims` features of row (r, t) is rotated by angle (offset + t) * scale * inv_freq[i]; `traditional` pairs
// (2i, 2i+1), else (i, i + dims/2); input scaled by `mscale` first (su / YaRN); fp32 math, one rounding.  The offset is
// an immediate or, for graph-replayed decode, read from device memory.  Features past `dims` are copied.  x is a
// strided [B, H, L, D] view (element strides sb, sh, sl; unit feature stride), y is contiguous and must not alias x.
// one rotation, the same explicit fma form in both kernels below (the two still disagree by an fp32 ulp on rare elements --
// 1 in ~2000 after the 16-bit rounding -- which the test allows)
__device__ __forceinline__ void rope_rot(float x1, float x2, float sn, float cs, float& o1, float& o2) {
  o1 = __fmaf_rn(x1, cs, -__fmul_rn(x2, sn));
  o2 = __fmaf_rn(x1, sn, __fmul_rn(x2, cs));
}

template <bool BF16>
__global__ __launch_bounds__(256) void rope_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ y,
                                                   const float* __restrict__ inv_freq, const long long* __restrict__ posp,
                                                   long long offset, long long nrows, int H, int L, int D, int dims,
                                                   long long sb, long long sh, long long sl, float scale, float mscale,
                                                   int traditional) {
  const long long base = posp ? *posp : offset;
  const int half = dims / 2;
  const long long total = nrows * half;
  for (long long e = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; e < total;
       e += static_cast<long long>(gridDim.x) * blockDim.x) {
    const long long row = e / half;                     // row = (b * H + h) * L + t
    const int i = static_cast<int>(e % half), t = static_cast<int>(row % L);
    const long long bh = row / L;
    const unsigned short* xr = x + (bh / H) * sb + (bh % H) * sh + t * sl;
    const int i1 = traditional ? 2 * i : i, i2 = traditional ? 2 * i + 1 : i + half;
    const float x1 = act_to_f32<BF16>(xr[i1]) * mscale, x2 = act_to_f32<BF16>(xr[i2]) * mscale;
    const float ang = static_cast<float>(base + t) * scale * inv_freq[i];
    float sn, cs;
    sincosf(ang, &sn, &cs);
    float o1, o2;
    rope_rot(x1, x2, sn, cs, o1, o2);
    y[row * D + i1] = f32_to_act<BF16>(o1);
    y[row * D + i2] = f32_to_act<BF16>(o2);
  }
  if (dims < D) {
    const int tail = D - dims;
    for (long long e = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; e < nrows * tail;
         e += static_cast<long long>(gridDim.x) * blockDim.x) {
      const long long row = e / tail, bh = row / L;
      const int j = dims + static_cast<int>(e % tail);
      y[row * D + j] = x[(bh / H) * sb + (bh % H) * sh + (row % L) * sl + j];
    }
  }
}

// Prompt-sized calls (many rows of one position share the angles): one workgroup per token (b, t); the dims / 2 (sin, cos)
// pairs of the position are computed ONCE (64 threads, same sincosf as above) and applied to all H heads with 16-byte loads and
// stores -- the generic kernel above computes them per (head, pair) with 2-byte accesses (100 us for a 2048 x 8 x 32 x 128
// chunk, 2.7 TB/s).  dims == D, D % 16 == 0 (non-traditional: 8-element chunk c pairs with chunk c + D / 16), D <= 256.
template <bool BF16>
__global__ __launch_bounds__(256) void rope_tokens_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ y,
                                                          const float* __restrict__ inv_freq, const long long* __restrict__ posp,
                                                          long long offset, int H, int L, int D, long long sb, long long sh,
                                                          long long sl, float scale, float mscale, int traditional,
                                                          long long yrows = 0, int yt0 = 0, const unsigned short* __restrict__ v = nullptr,
                                                          unsigned short* __restrict__ vy = nullptr, long long vsb = 0,
                                                          long long vsh = 0, long long vsl = 0) {
  // yrows > 0: y is a KV-cache buffer [B, H, yrows, D] written at rows yt0 + t (else the contiguous [B, H, L, D]); v / vy: the value
  // rows of the same tokens copied into their cache in the same launch (cache.update_and_fetch of a prompt chunk)
  __shared__ float sn_s[128], cs_s[128];
  const int t = blockIdx.x, b = blockIdx.y, half = D / 2;
  const long long yr_rows = yrows ? yrows : L;
  const long long base = posp ? *posp : offset;
  if (threadIdx.x < half) {
    const float ang = static_cast<float>(base + t) * scale * inv_freq[threadIdx.x];
    float sn, cs;
    sincosf(ang, &sn, &cs);
    sn_s[threadIdx.x] = sn;
    cs_s[threadIdx.x] = cs;
  }
  __syncthreads();
  const int cph = D / 16;                                 // chunk pairs per head
  for (int w = threadIdx.x; w < H * cph; w += 256) {
    const int h = w / cph, c = w % cph;
    const unsigned short* xr = x + b * sb + h * sh + t * sl;
    unsigned short* yr = y + ((static_cast<long long>(b) * H + h) * yr_rows + yt0 + t) * D;
    if (v) {                                              // the value chunk pair of this work item
      const unsigned short* vr = v + b * vsb + h * vsh + t * vsl;
      unsigned short* vd = vy + ((static_cast<long long>(b) * H + h) * yr_rows + yt0 + t) * D;
      *reinterpret_cast<u32x4*>(vd + c * 8) = *reinterpret_cast<const u32x4*>(vr + c * 8);
      *reinterpret_cast<u32x4*>(vd + half + c * 8) = *reinterpret_cast<const u32x4*>(vr + half + c * 8);
    }
    if (traditional) {                                    // pairs (2i, 2i+1): both 16-byte chunks of this work item are self-contained
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int e0 = (2 * c + q) * 8;
        const u32x4 v = *reinterpret_cast<const u32x4*>(xr + e0);
        const unsigned wv[4] = {v.x, v.y, v.z, v.w};
        unsigned o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int i = e0 / 2 + j;
          const float x1 = act_lo<BF16>(wv[j]) * mscale, x2 = act_hi<BF16>(wv[j]) * mscale;
          float o1, o2;
          rope_rot(x1, x2, sn_s[i], cs_s[i], o1, o2);
          o[j] = static_cast<unsigned>(f32_to_act<BF16>(o1)) | (static_cast<unsigned>(f32_to_act<BF16>(o2)) << 16);
        }
        *reinterpret_cast<u32x4*>(yr + e0) = u32x4{o[0], o[1], o[2], o[3]};
      }
    } else {
      const u32x4 va = *reinterpret_cast<const u32x4*>(xr + c * 8), vb = *reinterpret_cast<const u32x4*>(xr + half + c * 8);
      const unsigned wa[4] = {va.x, va.y, va.z, va.w}, wb[4] = {vb.x, vb.y, vb.z, vb.w};
      unsigned oa[4], ob[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = c * 8 + 2 * j;
        const float a0 = act_lo<BF16>(wa[j]) * mscale, a1 = act_hi<BF16>(wa[j]) * mscale;
        const float b0 = act_lo<BF16>(wb[j]) * mscale, b1 = act_hi<BF16>(wb[j]) * mscale;
        float p0, q0, p1, q1;
        rope_rot(a0, b0, sn_s[i], cs_s[i], p0, q0);
        rope_rot(a1, b1, sn_s[i + 1], cs_s[i + 1], p1, q1);
        oa[j] = static_cast<unsigned>(f32_to_act<BF16>(p0)) | (static_cast<unsigned>(f32_to_act<BF16>(p1)) << 16);
        ob[j] = static_cast<unsigned>(f32_to_act<BF16>(q0)) | (static_cast<unsigned>(f32_to_act<BF16>(q1)) << 16);
      }
      *reinterpret_cast<u32x4*>(yr + c * 8) = u32x4{oa[0], oa[1], oa[2], oa[3]};
      *reinterpret_cast<u32x4*>(yr + half + c * 8) = u32x4{ob[0], ob[1], ob[2], ob[3]};
    }
  }
}

hipError_t launch_rope(const void* x, void* y, const float* inv_freq, const long long* posp, long long offset, long long B,
                       int H, int L, int D, int dims, long long sb, long long sh, long long sl, float scale, float mscale,
                       int traditional, bool bf16, hipStream_t s) {
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (L >= 16 && dims == D && D % 16 == 0 && D <= 256 && B <= 65535 && al16(x) && al16(y) && sb % 8 == 0 && sh % 8 == 0 && sl % 8 == 0) {
    dim3 grid(L, static_cast<unsigned>(B));
    auto X = static_cast<const unsigned short*>(x);
    auto Y = static_cast<unsigned short*>(y);
    if (bf16) hipLaunchKernelGGL((rope_tokens_kernel<true>), grid, dim3(256), 0, s, X, Y, inv_freq, posp, offset, H, L, D, sb, sh, sl, scale, mscale, traditional);
    else hipLaunchKernelGGL((rope_tokens_kernel<false>), grid, dim3(256), 0, s, X, Y, inv_freq, posp, offset, H, L, D, sb, sh, sl, scale, mscale, traditional);
    return hipGetLastError();
  }
  const long long nrows = B * H * L, total = nrows * (dims / 2);
  const unsigned blocks = static_cast<unsigned>(std::min<long long>((total + 255) / 256, 16384));
  auto X = static_cast<const unsigned short*>(x);
  auto Y = static_cast<unsigned short*>(y);
  if (bf16) hipLaunchKernelGGL((rope_kernel<true>), dim3(blocks), dim3(256), 0, s, X, Y, inv_freq, posp, offset, nrows, H, L, D, dims, sb, sh, sl, scale, mscale, traditional);
  else hipLaunchKernelGGL((rope_kernel<false>), dim3(blocks), dim3(256), 0, s, X, Y, inv_freq, posp, offset, nrows, H, L, D, dims, sb, sh, sl, scale, mscale, traditional);
  return hipGetLastError();
}

// RoPE(k) of a prompt chunk written straight into the key cache + the value rows copied into theirs: rope(k, offset) followed by
// cache.update_and_fetch(k, v) (qllama.py:88-91, cache.py:222-244) as ONE launch.  k / v: strided [B, H, L, D] views of the
// projection outputs; caches [B, H, capacity, D]; rows t0 .. t0 + L - 1 are written, angles from position pos0 + t.
hipError_t launch_rope_kv_write(const void* k, const void* v, void* kcache, void* vcache, const float* inv_freq, long long pos0,
                                long long B, int H, int L, int D, long long ksb, long long ksh, long long ksl, long long vsb,
                                long long vsh, long long vsl, int capacity, int t0, float scale, int traditional, bool bf16,
                                hipStream_t s) {
  dim3 grid(L, static_cast<unsigned>(B));
  auto K = static_cast<const unsigned short*>(k);
  auto V = static_cast<const unsigned short*>(v);
  auto KC = static_cast<unsigned short*>(kcache);
  auto VC = static_cast<unsigned short*>(vcache);
  if (bf16) hipLaunchKernelGGL((rope_tokens_kernel<true>), grid, dim3(256), 0, s, K, KC, inv_freq, nullptr, pos0, H, L, D, ksb, ksh, ksl, scale, 1.f,
                               traditional, static_cast<long long>(capacity), t0, V, VC, vsb, vsh, vsl);
  else hipLaunchKernelGGL((rope_tokens_kernel<false>), grid, dim3(256), 0, s, K, KC, inv_freq, nullptr, pos0, H, L, D, ksb, ksh, ksl, scale, 1.f,
                          traditional, static_cast<long long>(capacity), t0, V, VC, vsb, vsh, vsl);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ batched RoPE + cache append
// One launch for a synchronous batched decode step (every sequence at position *pos): q is rotated in place, k is rotated
// into kcache[b, kvh, pos, :], v is copied into vcache[b, kvh, pos, :].  Replaces rope(q), rope(k) and
// cache.update_and_fetch of qllama.py:84-91 for L = 1 (fp32 math, one rounding, like mx.fast.rope).
template <bool BF16>
__global__ __launch_bounds__(128) void rope_append_kernel(unsigned short* __restrict__ q, const unsigned short* __restrict__ k,
                                                          const unsigned short* __restrict__ v, unsigned short* __restrict__ kcache,
                                                          unsigned short* __restrict__ vcache, const float* __restrict__ inv_freq,
                                                          const long long* __restrict__ posp, int n_heads, int n_kv, int hd,
                                                          int capacity, float rope_scale, int traditional) {
  const int b = blockIdx.y, hh = blockIdx.x;            // hh < n_heads: query head; else kv head hh - n_heads
  const int pos = static_cast<int>(*posp);
  const bool isk = hh >= n_heads;
  const int kvh = hh - n_heads;
  const unsigned short* src = isk ? k + (static_cast<size_t>(b) * n_kv + kvh) * hd : q + (static_cast<size_t>(b) * n_heads + hh) * hd;
  unsigned short* dst = isk ? kcache + ((static_cast<size_t>(b) * n_kv + kvh) * capacity + pos) * hd
                            : q + (static_cast<size_t>(b) * n_heads + hh) * hd;
  for (int i = threadIdx.x; i < hd / 2; i += blockDim.x) {
    const int i1 = traditional ? 2 * i : i, i2 = traditional ? 2 * i + 1 : i + hd / 2;
    const float x1 = act_to_f32<BF16>(src[i1]), x2 = act_to_f32<BF16>(src[i2]);
    const float rev = static_cast<float>(pos) * rope_scale * inv_freq[i] * 0.15915494309189535f;
    const float fr = rev - floorf(rev);
    const float sn = __builtin_amdgcn_sinf(fr), cs = __builtin_amdgcn_cosf(fr);
    dst[i1] = f32_to_act<BF16>(x1 * cs - x2 * sn);
    dst[i2] = f32_to_act<BF16>(x1 * sn + x2 * cs);
  }
  if (isk) {
    const unsigned short* vs = v + (static_cast<size_t>(b) * n_kv + kvh) * hd;
    unsigned short* vd = vcache + ((static_cast<size_t>(b) * n_kv + kvh) * capacity + pos) * hd;
    for (int i = threadIdx.x; i < hd; i += blockDim.x) vd[i] = vs[i];
  }
}

hipError_t launch_rope_append(void* q, const void* k, const void* v, void* kcache, void* vcache, const float* inv_freq,
                              const long long* pos, int batch, int n_heads, int n_kv, int hd, int capacity, float rope_scale,
                              int traditional, bool bf16, hipStream_t s) {
  dim3 grid(n_heads + n_kv, batch), block(128);
  auto Q = static_cast<unsigned short*>(q);
  auto K = static_cast<const unsigned short*>(k);
  auto V = static_cast<const unsigned short*>(v);
  auto KC = static_cast<unsigned short*>(kcache);
  auto VC = static_cast<unsigned short*>(vcache);
  if (bf16) hipLaunchKernelGGL((rope_append_kernel<true>), grid, block, 0, s, Q, K, V, KC, VC, inv_freq, pos, n_heads, n_kv, hd, capacity, rope_scale, traditional);
  else hipLaunchKernelGGL((rope_append_kernel<false>), grid, block, 0, s, Q, K, V, KC, VC, inv_freq, pos, n_heads, n_kv, hd, capacity, rope_scale, traditional);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ EMINF logits optimiser
// Entropy minimisation at inference (gbx_lm/infer_opt.py:30-77): normalised-gradient descent on the entropy of the
// next-token distribution, with accept / reject and step-size decay -- up to 3 + 15 r/(1+r) iterations of five reductions
// over the vocabulary, each of which the reference (and a torch restatement) syncs to the host for.  Here the whole
// optimisation is ONE launch of one 1024-thread workgroup: the logits (0.5 MB at V = 128K, L2-resident) are swept four
// times per iteration, every decision is taken on the device.
//   entropy(x) = -sum softmax(x) * log(softmax(x) + 1e-10)
//   grad = p * (sum(p * (logp + 1)) - (logp + 1));  x' = x - alpha * grad / |grad|;  accept if entropy(x') < entropy(x),
//   else alpha *= 0.65 (stop below 1e-4); stop once entropy < max(threshold, 0.1 * entropy0).
// (Caching the probabilities between the passes was measured no faster: one CU's 64 B/clk path to L2 is the bound then.)
struct EminfArgs {
  const void* logits;       // [V] fp32 (in_mode 2) or act dtype
  float* cur;               // [V] work / result: best logits
  float* nxt;               // [V] scratch
  float* out_entropy;       // [2]: best entropy, initial entropy
  int V, in_mode;           // in_mode: 0 f16, 1 bf16, 2 f32
  float alpha, threshold;
  int num_steps;            // < 0: the reference's allocation from the initial entropy
};

__device__ __forceinline__ float block_sum(float v, float* red, int tid) {
  v = wave_sum(v);
  __syncthreads();                          // previous use of red[] is over
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) t += red[i];
  return t;
}
__device__ __forceinline__ float block_max(float v, float* red, int tid) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d));
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  float t = red[0];
#pragma unroll
  for (int i = 1; i < 16; ++i) t = fmaxf(t, red[i]);
  return t;
}

__global__ __launch_bounds__(1024) void eminf_kernel(const EminfArgs a) {
  __shared__ float red[16];
  const int tid = threadIdx.x, V = a.V;
  float* cur = a.cur;
  float* nxt = a.nxt;
  // load + (max, sum exp) + entropy of the input   (every thread re-reads only elements it wrote itself)
  float m0 = -3.0e38f;
  for (int i = tid; i < V; i += 1024) {
    float v;
    if (a.in_mode == 2) v = static_cast<const float*>(a.logits)[i];
    else if (a.in_mode == 1) v = act_to_f32<true>(static_cast<const unsigned short*>(a.logits)[i]);
    else v = act_to_f32<false>(static_cast<const unsigned short*>(a.logits)[i]);
    cur[i] = v;
    m0 = fmaxf(m0, v);
  }
  float mx = block_max(m0, red, tid);
  float z0 = 0.f;
  for (int i = tid; i < V; i += 1024) z0 += __expf(cur[i] - mx);
  float logz = __logf(block_sum(z0, red, tid));
  float h = 0.f;
  for (int i = tid; i < V; i += 1024) {
    const float p = __expf(cur[i] - mx - logz);
    h -= p * __logf(p + 1e-10f);
  }
  float hc = block_sum(h, red, tid);
  const float h0 = hc;
  float best_h = hc;
  const float target = fmaxf(a.threshold, 0.1f * h0);
  int steps = a.num_steps;
  if (steps < 0) {
    const float r = fmaxf(0.f, h0 - a.threshold) / a.threshold;
    steps = static_cast<int>(3.f + 15.f * (r / (1.f + r)));
  }
  float alpha = a.alpha;
  for (int st = 0; st < steps; ++st) {
    if (hc < target) break;
    // S = sum p (logp + 1); |grad|^2 = sum (p (S - (logp + 1)))^2
    float s = 0.f;
    for (int i = tid; i < V; i += 1024) {
      const float lp = cur[i] - mx - logz;
      s += __expf(lp) * (lp + 1.f);
    }
    const float S = block_sum(s, red, tid);
    float g2 = 0.f;
    for (int i = tid; i < V; i += 1024) {
      const float lp = cur[i] - mx - logz;
      const float g = __expf(lp) * (S - (lp + 1.f));
      g2 += g * g;
    }
    const float gn = sqrtf(block_sum(g2, red, tid));
    if (!(gn > 0.f)) break;
    const float step = alpha / gn;
    // update pass with an online (max, sum exp) of the new iterate, then ONE more pass for its entropy
    float m2 = -3.0e38f, z2 = 0.f;
    for (int i = tid; i < V; i += 1024) {
      const float lp = cur[i] - mx - logz;
      const float v = cur[i] - step * (__expf(lp) * (S - (lp + 1.f)));
      nxt[i] = v;
      if (v > m2) { z2 = z2 * __expf(m2 - v) + 1.f; m2 = v; }
      else z2 += __expf(v - m2);
    }
    const float mx2 = block_max(m2, red, tid);
    const float logz2 = __logf(block_sum(z2 * __expf(m2 - mx2), red, tid));
    float hsum = 0.f;
    for (int i = tid; i < V; i += 1024) {
      const float p = __expf(nxt[i] - mx2 - logz2);
      hsum -= p * __logf(p + 1e-10f);
    }
    const float hn = block_sum(hsum, red, tid);
    if (hn < hc) {                              // accept: the new iterate becomes current (and best: entropy only falls)
      float* t = cur; cur = nxt; nxt = t;
      mx = mx2; logz = logz2; hc = hn;
      if (hn < best_h) best_h = hn;
    } else {
      alpha *= 0.65f;
      if (alpha < 1e-4f) break;
    }
  }
  if (cur != a.cur) {                           // result always lands in a.cur
    for (int i = tid; i < V; i += 1024) a.cur[i] = cur[i];
  }
  if (tid == 0) { a.out_entropy[0] = best_h; a.out_entropy[1] = h0; }
}

hipError_t launch_eminf(const void* logits, float* cur, float* nxt, float* out_entropy, int V, int in_mode, float alpha,
                        float threshold, int num_steps, hipStream_t s) {
  EminfArgs a{logits, cur, nxt, out_entropy, V, in_mode, alpha, threshold, num_steps};
  hipLaunchKernelGGL(eminf_kernel, dim3(1), dim3(1024), 0, s, a);
  return hipGetLastError();
}

}  // namespace gbx
