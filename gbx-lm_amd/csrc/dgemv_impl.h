// Decode-block GEMV ("dgemv"): the M == 1 launches of a decoder block, y[n] = sum_k x[k] * (s[n,k/g] * q[n,k] + z[n,k/g]),
// with the block's fusions (RMSNorm prologue, residual epilogue, SwiGLU epilogue).
//
// Replaces the M == 1 case of mx.quantized_matmul(..., transpose=True) (gbx_lm/models/quantized_linear_gba.py:195-203) at its
// seven call sites per block (gbx_lm/models/qllama.py:76, 115, 137-141).  Same arithmetic as gemv_kernel (gemv_impl.h: in-place
// field masks as fp16 subnormals, v_dot2, per-sub-slot block exponent, s*sum(q x) + z*sum(x)), bit-identical to it for the same
// workgroup width; what is different is everything AROUND the per-weight work, which is what a launch over rows of a few
// thousand values is bound by (profiles/r03_layer_timeline_after.txt, r03_chain_floor.txt):
//   * 104 bytes of kernel arguments read by one group of scalar loads (gemv_kernel: 280 bytes fetched in five dependent
//     round trips, ~0.7 us before the first activation load was issued);
//   * R = 2 or 4 rows per batch: one 16-byte (scale|zero) load for four rows, one 4-byte residual load, descriptors built
//     once, a dozen scalar instructions per batch (gemv_kernel: ~200);
//   * the epilogue runs on the two lanes that hold the row totals (no readlane, no single-lane scalar epilogue);
//   * no steady-state loop for the decode shapes: a wave owns at most two batches, both in flight before it consumes
//     (deeper launches fall through a 2-deep ring); a wave without a batch issues its loads with the out-of-range bit
//     set in the address, so they cost no memory traffic and vmcnt stays exactly countable;
//   * the RMSNorm statistic can come from the PRODUCER of x: the o_proj / down_proj launches leave one partial sum of
//     squares per wave (ssq_out, plain stores, fixed slots), the following norm'd launch adds 1024 floats in every
//     wave instead of a block reduction behind a barrier that waits for the slowest wave's x.
#pragma once
#include "gbx_device.h"
#include "gemv_fmt.h"
#include <hip/hip_runtime.h>

namespace gbx {

constexpr int DG_SSQ_SLOTS = 1024;      // floats a partial-statistic buffer holds (a producer writes one per wave / workgroup, the rest stay 0)

struct DgArgs {
  const unsigned short* x;          // [K] activations (act dtype)
  const uint32_t* qw;               // [N, K*bits/32]; SwiGLU: gate
  const uint32_t* qw2;              // SwiGLU: up (same shape); otherwise unused
  const uint32_t* sz;               // packed (scale | zero << 16) companion in batch order: [nbatch][K/g][R]
  unsigned short* y;                // [N]
  const unsigned short* residual;   // [N] (EPI 1)
  const unsigned short* norm_w;     // [K] RMSNorm weight or null
  const float* ssq_in;              // [DG_SSQ_SLOTS] partial sums of x^2 left by the producer of x, or null (computed here)
  float* ssq_out;                   // partial sums of y^2 of this launch (one per wave; KS == 4: one per workgroup), or null
  // round 5 (the fused attention launch, dattn_impl.h, leaves o_proj's partial sums in a 2^-32 fixed-point accumulator):
  const long long* x_acc;           // [K + 1] or null: x <- round(x + round(x_acc * 2^-32)) in the prologue (word K != 0: x <- NaN)
  unsigned short* x_out;            // with x_acc: workgroup 0 writes the new x here (the residual stream the next launch adds to)
  unsigned long long* zero_buf;     // or null: zero_n 8-byte words this launch clears (the accumulator, once its consumer has run)
  int zero_n;
  int N;                            // output rows (SwiGLU: rows of ONE of the two matrices)
  int K;
  int log2g;                        // 6, 7 (2-bit) / 5, 6, 7 (3-, 4-bit): one (scale, zero) per lane and pass
  int nbatch;                       // batches of R rows (SwiGLU: R/2 gate + R/2 up rows)
  int bpo;                          // rounds: a row owner takes batch j*owners + (its index) in round j < bpo - 1 ...
  int apw;                          // ... and, in the last round, owners [0, apw) of a workgroup take batch (bpo-1)*owners + block*apw + owner
  float eps;
  int null_rows;                    // measurement only (GBX_CHAIN_FLOOR=2): every batch re-reads batch 0's rows
  int long_prio;                    // 1: owners with a batch in the last round run at s_setprio 1 (uneven launches: gate|up)
#ifdef GBX_STAMPS
  unsigned long long* stamps;       // diagnostic build only: [workgroup][wave][16] s_memrealtime stamps
#endif
};

template <int BITS, int P, int R>
struct DgBatch {
  typename SlotWords<BITS>::type w[R][P];
  unsigned sz[P][R];
  unsigned res;
};

// EPI: 0 plain store, 1 y = residual + round(W x), 2 SwiGLU (rows [0, R/2) of a batch from qw = gate, [R/2, R) from qw2 = up).
// KS : 1 a wave owns whole rows; 4 the four waves of a row group own a quarter of K each (partial sums meet in LDS).
// NW : waves per workgroup.  R: rows per batch.
// D  : 1 = every row owner has exactly one batch (bpo == 1: the short launches -- no second register buffer, no second issue);
//      2 = two register buffers, a batch is requested when its buffer has been consumed (batch 1: right behind the prologue's barrier).
// LATE: where batch 0 is requested.  false: right behind the activation loads, ahead of the prologue (launches without a norm:
//      o_proj, down_proj); true: only once the image has been written (launches with the RMSNorm prologue: q|k|v, gate|up).
// Measured (scratch/dg_order.sh, one launch kind replayed back to back from a graph, 2-bit 8B shapes, us per launch):
//                                   q|k|v   o      gate|up  down
//   batch 0 ahead of the prologue    5.3    3.7    10.7     6.6       (R = 4 / 4 / 2 / 2)
//   batch 0 behind the image (LATE)  4.7    3.9     9.9     6.6
//   batches 0 AND 1 ahead            5.6    3.7    11.4     7.6
//   batch 1 once x has landed        5.5    3.6    11.1     7.3
// Every weight byte requested ahead of x delays x and the norm weight -- all workgroups read the same 8-16 KB through the L2
// channels the stream uses -- and the statistic's barrier then waits for the slowest wave's x; the launches without a norm
// have no such barrier and gain from the early request.  (An experiment build that requested NO weights at all ran q|k|v in
// 4.5, o in 3.4 and down in 6.7 us: these launches are bound by prologue and consumption latency, not by the stream.)
template <int BITS, int P, int KS, int NW, int R, int EPI, int D, bool LATE, bool BF16>
__global__ __launch_bounds__(NW * 64) void dgemv_kernel(const DgArgs a) {
  static_assert(D >= 1 && D <= 4, "register buffers");
  static_assert(!LATE || KS == 1, "the late-request form goes with the RMSNorm prologue");
  static_assert(R == 2 || R == 4, "rows per batch");
  static_assert(KS == 1 || KS == 4, "K split");
  static_assert(KS == 1 || EPI != 2, "SwiGLU launches own whole rows");
  using F = Fmt<BITS>;
  using L = Lds<BITS>;
  using WT = typename SlotWords<BITS>::type;
  constexpr int NT = NW * 64;
  constexpr int NWO = KS == 1 ? NW : NW / 4;               // row owners per workgroup
  constexpr int PAIRS = L::PAIRS, SSTR = L::SSTR, NSUB = F::NSUB;
  constexpr int UPS = F::VPL / F::UNIT_VALS;
  constexpr bool SWI = EPI == 2;
  constexpr int RO = SWI ? R / 2 : R;                      // output rows per batch
  constexpr unsigned KILL = 0x80000000u;                   // buffer offset beyond num_records: the load returns 0 and fetches nothing

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int K = a.K;
  const int nslots = K / F::VPL;
  h2* xp_lds = reinterpret_cast<h2*>(smem);
  float2* xs_lds = reinterpret_cast<float2*>(smem + static_cast<size_t>(nslots) * SSTR * 4);
  float* red_lds = reinterpret_cast<float*>(xs_lds + nslots * NSUB);     // NW floats
  float* part_lds = red_lds + NW;                                         // KS == 4: [NWO][bpo][R][4]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef GBX_STAMPS
#define DG_STAMP(i) do { if (lane == 0 && a.stamps) a.stamps[(static_cast<size_t>(blockIdx.x) * NW + wave) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define DG_STAMP(i) do {} while (0)
#endif
  DG_STAMP(0);
  const int W4 = K / 32 * BITS * 4;                        // bytes per packed row
  const int G = K >> a.log2g;
  const int gshift = a.log2g - 5;
  const int kq = KS == 1 ? 0 : (wave & 3), ow = KS == 1 ? wave : (wave >> 2);
  const int owners = static_cast<int>(gridDim.x) * NWO;
  const int bpo = a.bpo, nbatch = a.nbatch;
  auto batch_of = [&](int j) {                             // >= nbatch: this owner has no batch in round j
    if (j < bpo - 1) return j * owners + static_cast<int>(blockIdx.x) * NWO + ow;
    if (j == bpo - 1 && ow < a.apw) return j * owners + static_cast<int>(blockIdx.x) * a.apw + ow;
    return nbatch;
  };

  constexpr int RS_FLAGS = 0x00020000, RS_ALL = 0x7FFFFFFF;
  auto rsrc = [](const void* ptr) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, RS_ALL, RS_FLAGS); };
  const __amdgpu_buffer_rsrc_t qd = rsrc(a.qw), qd2 = rsrc(SWI ? a.qw2 : a.qw), szd = rsrc(a.sz);
  const __amdgpu_buffer_rsrc_t resd = rsrc(EPI == 1 ? static_cast<const void*>(a.residual) : static_cast<const void*>(a.y));

  // per-lane byte offsets that never change: packed words / (scale|zero) group of each pass (lanes past the row: killed)
  unsigned voff_w[P], voff_s[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int slot = p * 64 * KS + kq * 64 + lane;
    const unsigned dead = slot < nslots ? 0u : KILL;
    voff_w[p] = static_cast<unsigned>(slot * (F::SLOT_WORDS * 4)) | dead;
    voff_s[p] = static_cast<unsigned>(((slot * NSUB) >> gshift) * (R * 4)) | dead;
  }
  const unsigned voff_r = (lane & 32) ? R : 0;             // residual: lanes 0-31 take the batch's first R/2 rows, 32-63 the others

  // rows [i0, i1) of batch bi; the batch's small loads ((scale|zero) words, residual) go with its last row
  auto issue_part = [&](DgBatch<BITS, P, R>& b, int bi, int i0, int i1) {
    const bool live = bi < nbatch;
    const int bc = (live && !a.null_rows) ? bi : 0;
    const unsigned kill = live ? 0u : KILL;
#pragma unroll
    for (int i = 0; i < R; ++i) {
      if (i < i0 || i >= i1) continue;
      const int row = SWI ? bc * RO + (i % RO) : bc * R + i;
      const int woff = row * W4;
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const __amdgpu_buffer_rsrc_t d = (SWI && i >= RO) ? qd2 : qd;
        if constexpr (BITS == 3) b.w[i][p] = __builtin_amdgcn_raw_buffer_load_b96(d, voff_w[p] | kill, woff, 2);    // nt: streamed once
        else b.w[i][p] = __builtin_amdgcn_raw_buffer_load_b128(d, voff_w[p] | kill, woff, 2);
      }
    }
    if (i1 < R) return;
    const int soff = (live ? bi : 0) * (G * R * 4);
#pragma unroll
    for (int p = 0; p < P; ++p) {
      if constexpr (R == 4) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(szd, voff_s[p] | kill, soff, 0);
        b.sz[p][0] = v.x; b.sz[p][1] = v.y; b.sz[p][2] = v.z; b.sz[p][3] = v.w;
      } else {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(szd, voff_s[p] | kill, soff, 0);
        b.sz[p][0] = v.x; b.sz[p][1] = v.y;
      }
    }
    if constexpr (EPI == 1 && KS == 1) {
      if constexpr (R == 4) b.res = __builtin_amdgcn_raw_buffer_load_b32(resd, voff_r | kill, (live ? bi : 0) * (R * 2), 0);
      else b.res = __builtin_amdgcn_raw_buffer_load_b16(resd, voff_r | kill, (live ? bi : 0) * (R * 2), 0);
    }
  };
  auto issue = [&](DgBatch<BITS, P, R>& b, int bi) { issue_part(b, bi, 0, R); };

  // ---------------- 1. activation loads first (vmcnt retires in order), then the statistic partials, then batch 0 ------------
  // Prologue geometry as in gemv_kernel: a thread converts PPU pairs of one unit; sub-slot statistics meet through DPP.
  constexpr int UP = F::UNIT_PAIRS;
  constexpr int PAIRS_MAX = P * KS * 64 * (F::VPL / 2);
  constexpr int PPQ = (PAIRS_MAX + NT - 1) / NT;
  constexpr int PP = (BITS == 3) ? UP : (PPQ <= 2 ? 2 : (PPQ <= 4 ? 4 : UP));
  constexpr int PPU = PP < UP ? PP : UP;
  constexpr int TPU = UP / PPU;
  constexpr int UPT = (PAIRS_MAX + NT * PPU - 1) / (NT * PPU);
  constexpr int GL = (BITS == 3) ? 1 : 16 / PPU;
  constexpr int XW = PPU;
  const int nunits = K / F::UNIT_VALS;
  const int nitems = nunits * TPU;
  const bool norm = (KS == 1) && a.norm_w != nullptr;
  const unsigned short* __restrict__ xg = a.x;
  const unsigned short* __restrict__ wg = norm ? a.norm_w : xg;
  constexpr int WPT = (KS == 1) ? UPT : 1;
  unsigned xr[UPT][XW], wr[WPT][XW];
  // (the base pointer is global memory, or -- x_acc launches -- the LDS copy of x': `auto` keeps the address space)
  auto load_item = [&](auto base, int it, unsigned* out) {
    const int u = it / TPU, t = it % TPU;
    if constexpr (BITS == 3) {
      auto src = base + static_cast<size_t>(u) * F::UNIT_VALS;
#pragma unroll
      for (int q = 0; q < 4; ++q) { const u32x4 v = *(const u32x4*)(src + 8 * q); out[4 * q] = v.x; out[4 * q + 1] = v.y; out[4 * q + 2] = v.z; out[4 * q + 3] = v.w; }
    } else {
      auto pa = base + static_cast<size_t>(u) * F::UNIT_VALS + PPU * t;
      auto pb = pa + UP;
      if constexpr (PPU == 2) {
        out[0] = *(const unsigned*)(pa);
        out[1] = *(const unsigned*)(pb);
      } else if constexpr (PPU == 4) {
        const u32x2 va = *(const u32x2*)(pa), vb = *(const u32x2*)(pb);
        out[0] = va.x; out[1] = va.y; out[2] = vb.x; out[3] = vb.y;
      } else {
        const u32x4 va = *(const u32x4*)(pa), vb = *(const u32x4*)(pb);
        out[0] = va.x; out[1] = va.y; out[2] = va.z; out[3] = va.w; out[4] = vb.x; out[5] = vb.y; out[6] = vb.z; out[7] = vb.w;
      }
    }
  };
  if constexpr (KS == 1) {
#pragma unroll
    for (int j = 0; j < UPT; ++j) load_item(wg, min(tid + NT * j, nitems - 1), wr[j]);
  }
  if (a.x_acc) {                                             // (kernel-argument uniform; launches with the RMSNorm prologue only)
    // x' = round(x + round(acc * 2^-32)): o_proj's output rounded once, the residual add rounded once (gbx_lm/models/qllama.py:96,138),
    // formed ONCE per workgroup by all its threads (a pair of values each) and parked in LDS, where the item loads below find it;
    // workgroup 0 also writes it out as the new residual stream.  acc[K] != 0 (a non-finite partial sum upstream): x' = NaN.
    unsigned* xl = reinterpret_cast<unsigned*>(red_lds + NW);
    const unsigned* x32 = reinterpret_cast<const unsigned*>(a.x);
    const bool poison = a.x_acc[K] != 0;
    constexpr int XU = 4;
    float ssq_acc = 0.f;                                     // the statistic of x' rides on the same barrier as x' itself
    for (int base = 0; base < K / 2; base += NT * XU) {
      unsigned hv[XU];
      u32x4 av[XU];
#pragma unroll
      for (int i = 0; i < XU; ++i) {
        const int wi = min(base + NT * i + tid, K / 2 - 1);
        hv[i] = x32[wi];
        av[i] = *reinterpret_cast<const u32x4*>(a.x_acc + 2 * wi);
      }
#pragma unroll
      for (int i = 0; i < XU; ++i) {
        const int wi = base + NT * i + tid;
        const float o0 = round_bf16_or_f16(fx32_to_f32(av[i].x, av[i].y), BF16);
        const float o1 = round_bf16_or_f16(fx32_to_f32(av[i].z, av[i].w), BF16);
        float x0 = act_lo<BF16>(hv[i]) + o0, x1 = act_hi<BF16>(hv[i]) + o1;
        if (poison) { x0 = __uint_as_float(0x7FC00000u); x1 = x0; }
        const unsigned wv = static_cast<unsigned>(f32_to_act<BF16>(x0)) | (static_cast<unsigned>(f32_to_act<BF16>(x1)) << 16);
        if (wi < K / 2) {
          xl[wi] = wv;
          if (blockIdx.x == 0 && a.x_out) reinterpret_cast<unsigned*>(a.x_out)[wi] = wv;
          const float lo = act_lo<BF16>(wv), hi = act_hi<BF16>(wv);
          ssq_acc = fmaf(hi, hi, fmaf(lo, lo, ssq_acc));
        }
      }
    }
    ssq_acc = wave_sum(ssq_acc);
    if (lane == 0) red_lds[wave] = ssq_acc;
    __syncthreads();
    const __attribute__((address_space(3))) unsigned short* xs3 = (const __attribute__((address_space(3))) unsigned short*)(xl);
#pragma unroll
    for (int j = 0; j < UPT; ++j) load_item(xs3, min(tid + NT * j, nitems - 1), xr[j]);
  } else {
#pragma unroll
    for (int j = 0; j < UPT; ++j) load_item(xg, min(tid + NT * j, nitems - 1), xr[j]);
  }
  // partial sums of x^2 from the producer (every wave adds all of them itself: no barrier on this path)
  u32x4 pq[4] = {u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}};
  const bool carried = norm && a.ssq_in != nullptr;
  if constexpr (KS == 1) {
    const __amdgpu_buffer_rsrc_t pd = rsrc(carried ? static_cast<const void*>(a.ssq_in) : static_cast<const void*>(a.x));
    const unsigned pk = carried ? 0u : KILL;
#pragma unroll
    for (int i = 0; i < 4; ++i) pq[i] = __builtin_amdgcn_raw_buffer_load_b128(pd, static_cast<unsigned>(lane * 16 + 1024 * i) | pk, 0, 0);
  }

  if (a.zero_buf) {                                          // (kernel-argument uniform)
    for (int i = static_cast<int>(blockIdx.x) * NT + tid; i < a.zero_n; i += static_cast<int>(gridDim.x) * NT) a.zero_buf[i] = 0ull;
  }
  DgBatch<BITS, P, R> ring[D];
#ifndef DG_EXP_EARLY_LATE
#define DG_EXP_EARLY_LATE 0      // experiment: rows of batch 0 the LATE form still requests ahead of the prologue
#endif
#ifndef DG_EXP_EARLY_EARLY
#define DG_EXP_EARLY_EARLY R     // experiment: rows of batch 0 the early form requests ahead of the prologue
#endif
  constexpr int E0 = LATE ? DG_EXP_EARLY_LATE : DG_EXP_EARLY_EARLY;
  if constexpr (E0 > 0) issue_part(ring[0], batch_of(0), 0, E0);
  DG_STAMP(1);
#ifdef GBX_STAMPS
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"(E0 * P + (E0 == R ? P + ((EPI == 1 && KS == 1) ? 1 : 0) : 0)) : "memory");   // x, norm weight, partials have landed
  DG_STAMP(8);
#endif

  // ---------------- 2. RMSNorm statistic ------------------------------------------------------------------------------------
  float rstd = 1.f;
  if (norm) {
    float tot;
    if (carried) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        s += (__uint_as_float(pq[i].x) + __uint_as_float(pq[i].y)) + (__uint_as_float(pq[i].z) + __uint_as_float(pq[i].w));
      tot = wave_sum(s);
    } else if (a.x_acc) {                                      // (reduced with x' above: one barrier for both)
      tot = 0.f;
#pragma unroll
      for (int i = 0; i < NW; ++i) tot += red_lds[i];
    } else {
      float ssq = 0.f;
#pragma unroll
      for (int j = 0; j < UPT; ++j) {
        const bool live = (tid + NT * j) < nitems;
#pragma unroll
        for (int q = 0; q < XW; ++q) {
          const float lo = act_lo<BF16>(xr[j][q]), hi = act_hi<BF16>(xr[j][q]);
          ssq += live ? fmaf(hi, hi, lo * lo) : 0.f;
        }
      }
      ssq = wave_sum(ssq);
      if (lane == 0) red_lds[wave] = ssq;
      __syncthreads();
      tot = 0.f;
#pragma unroll
      for (int i = 0; i < NW; ++i) tot += red_lds[i];
    }
    rstd = rsqrtf(tot / static_cast<float>(K) + a.eps);
  }
  DG_STAMP(9);

  // ---------------- 3. activations -> pre-shifted fp16 pairs, one block exponent per 32-value sub-slot ------------------------
#pragma unroll
  for (int j = 0; j < UPT; ++j) {
    if ((tid & ~63) + NT * j < nitems) {
      const int it = tid + NT * j;
      const int u = it / TPU, t = it % TPU;
      float xa[PPU], xb[PPU];
      unsigned amax = 0;
      float usum = 0.f;
#pragma unroll
      for (int i = 0; i < PPU; ++i) {
        int wa, ha, wb, hb;
        if constexpr (BITS == 3) { wa = F::ka(i) >> 1; ha = F::ka(i) & 1; wb = F::kb(i) >> 1; hb = F::kb(i) & 1; }
        else { wa = i >> 1; ha = i & 1; wb = (PPU >> 1) + (i >> 1); hb = i & 1; }
        float va = ha ? act_hi<BF16>(xr[j][wa]) : act_lo<BF16>(xr[j][wa]);
        float vb = hb ? act_hi<BF16>(xr[j][wb]) : act_lo<BF16>(xr[j][wb]);
        if constexpr (KS == 1) {
          if (norm) {
            va = round_bf16_or_f16(va * rstd * (ha ? act_hi<BF16>(wr[j][wa]) : act_lo<BF16>(wr[j][wa])), BF16);
            vb = round_bf16_or_f16(vb * rstd * (hb ? act_hi<BF16>(wr[j][wb]) : act_lo<BF16>(wr[j][wb])), BF16);
          }
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
  }
  DG_STAMP(10);
#ifndef DG_EXP_AFTER
#define DG_EXP_AFTER 0           // experiment: 3- / 4-bit LATE launches request batch 0 BEHIND the image barrier (the request itself can wait for queue space)
#endif
  constexpr bool AFTER = LATE && BITS >= 3 && DG_EXP_AFTER != 0;
  if constexpr (E0 < R && !AFTER) issue_part(ring[0], batch_of(0), E0, R);
  __syncthreads();
  DG_STAMP(2);
  if constexpr (E0 < R && AFTER) issue_part(ring[0], batch_of(0), E0, R);
#pragma unroll
  for (int d = 1; d < D; ++d) issue(ring[d], batch_of(d));

  // a wave that owns no batch at all is done (it was only here for the prologue); KS == 4 and ssq_out launches meet again at a
  // barrier / store below, so their idle waves stay
  if constexpr (KS == 1) {
    if (batch_of(0) >= nbatch && !a.ssq_out) return;
  }
  // ---------------- 4. lane-resident activations ------------------------------------------------------------------------------
  h2 xv[P][PAIRS];
  float xs[P][NSUB], cf[P][NSUB];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int slot = p * 64 * KS + kq * 64 + lane;
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

  if (a.long_prio && ow < a.apw) __builtin_amdgcn_s_setprio(1);     // wave-uniform
  DG_STAMP(3);
  // ---------------- 5. consume one batch: R row totals, then the epilogue on the lanes that hold them -------------------------
  float sq = 0.f;                                           // this lane's share of sum(y^2) (lanes 31 / 63 only)
  auto consume = [&](const DgBatch<BITS, P, R>& cur, int j) {
    const int b = batch_of(j);
    if (b >= nbatch) return;                                // wave-uniform; no loads behind this branch
    float acc[R];
#pragma unroll
    for (int i = 0; i < R; ++i) acc[i] = 0.f;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      float d[R][NSUB];
      WT wp[R];
#pragma unroll
      for (int i = 0; i < R; ++i) wp[i] = cur.w[i][p];
      dot_slot_rows<BITS, R>(wp, xv[p], d);
#pragma unroll
      for (int i = 0; i < R; ++i) {
        const float sc = sz_scale<BF16>(cur.sz[p][i]), ze = sz_zero<BF16>(cur.sz[p][i]);
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
    // totals: lane 31 holds the first operand's, lane 63 the second operand's (wave_sum2)
    float v0, v1 = 0.f;
    if constexpr (R == 2) v0 = wave_sum2(acc[0], acc[1]);
    else if constexpr (SWI) { v0 = wave_sum2(acc[0], acc[1]); v1 = wave_sum2(acc[2], acc[3]); }     // (g0 | g1), (u0 | u1)
    else { v0 = wave_sum2(acc[0], acc[2]); v1 = wave_sum2(acc[1], acc[3]); }                       // rows (0 | 2), (1 | 3)
    if constexpr (KS == 4) {
      if ((lane & 31) == 31) {
        float* pp = part_lds + (static_cast<size_t>(ow * bpo + j) * R + (lane >> 5) * (R / 2)) * 4 + kq;   // [owner][round][row][quarter]
        pp[0] = v0;
        if constexpr (R == 4) pp[4] = v1;
      }
      return;
    }
    if constexpr (SWI) {
      if constexpr (R == 2) {
        const float t0 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v0), 31));
        const float t1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v0), 63));
        const float g = round_bf16_or_f16(t0, BF16), up = round_bf16_or_f16(t1, BF16);
        const float sl = round_bf16_or_f16(g / (1.f + __expf(-g)), BF16);
        if (lane == 0) a.y[b] = f32_to_act<BF16>(sl * up);
      } else if ((lane & 31) == 31) {     // lane 31: pair 0 of the batch, lane 63: pair 1
        const float g = round_bf16_or_f16(v0, BF16), up = round_bf16_or_f16(v1, BF16);
        const float sl = round_bf16_or_f16(g / (1.f + __expf(-g)), BF16);
        a.y[b * RO + (lane >> 5)] = f32_to_act<BF16>(sl * up);
      }
    } else if ((lane & 31) == 31) {
      if constexpr (R == 2) {             // lane 31: row 2b, lane 63: row 2b + 1
        float v = v0;
        if constexpr (EPI == 1) v = act_to_f32<BF16>(static_cast<unsigned short>(cur.res & 0xFFFFu)) + round_bf16_or_f16(v, BF16);
        const unsigned short o = f32_to_act<BF16>(v);
        a.y[b * R + (lane >> 5)] = o;
        const float f = act_to_f32<BF16>(o);
        sq = fmaf(f, f, sq);
      } else {                            // lane 31: rows 4b, 4b + 1; lane 63: rows 4b + 2, 4b + 3 -- one 4-byte store each
        float p0 = v0, p1 = v1;
        if constexpr (EPI == 1) {
          p0 = act_lo<BF16>(cur.res) + round_bf16_or_f16(p0, BF16);
          p1 = act_hi<BF16>(cur.res) + round_bf16_or_f16(p1, BF16);
        }
        const unsigned o0 = f32_to_act<BF16>(p0), o1 = f32_to_act<BF16>(p1);
        *reinterpret_cast<unsigned*>(a.y + static_cast<size_t>(b) * R + (lane >> 5) * 2) = o0 | (o1 << 16);
        const float f0 = act_to_f32<BF16>(static_cast<unsigned short>(o0)), f1 = act_to_f32<BF16>(static_cast<unsigned short>(o1));
        sq = fmaf(f1, f1, fmaf(f0, f0, sq));
      }
    }
  };

  // ---------------- 6. one batch (D == 1), or the D-deep ring: the decode shapes have at most D batches per owner and skip the loop -----
#ifdef GBX_STAMPS     // when has batch 0 landed?  (diagnostic build only: the explicit wait is not in the product kernel)
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"((D - 1) * (R * P + P + ((EPI == 1 && KS == 1) ? 1 : 0))) : "memory");
  DG_STAMP(11);
#endif
  if constexpr (D == 1) {
    consume(ring[0], 0);
    DG_STAMP(4);
  } else {
    int base = 0;
    for (; base + D < bpo; base += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        consume(ring[d], base + d);
        issue(ring[d], batch_of(base + D + d));
      }
    }
#pragma unroll
    for (int d = 0; d < D; ++d) {
      consume(ring[d], base + d);
      if (d == 0) DG_STAMP(4);
      if (d == 1) DG_STAMP(5);
    }
  }
  DG_STAMP(6);

  if constexpr (KS == 4) {
    __syncthreads();
    // one thread per output row: quarter sums in the order of gemv_kernel, residual, store
    for (int e = tid; e < NWO * bpo * R; e += NT) {
      const int o = e / (bpo * R), jr = e % (bpo * R), j = jr / R, i = jr % R;
      int b;
      if (j < bpo - 1) b = j * owners + static_cast<int>(blockIdx.x) * NWO + o;
      else b = (o < a.apw) ? j * owners + static_cast<int>(blockIdx.x) * a.apw + o : nbatch;
      if (b < nbatch) {
        const float4 p = *reinterpret_cast<const float4*>(part_lds + static_cast<size_t>(e) * 4);
        float v = (p.x + p.y) + (p.z + p.w);
        const int row = b * R + i;
        if constexpr (EPI == 1) v = act_to_f32<BF16>(a.residual[row]) + round_bf16_or_f16(v, BF16);
        const unsigned short ov = f32_to_act<BF16>(v);
        a.y[row] = ov;
        const float f = act_to_f32<BF16>(ov);
        sq = fmaf(f, f, sq);
      }
    }
  }
  // partial sums of y^2, one per WAVE that stored rows (plain store, no barrier, fixed slot): the consumer adds all DG_SSQ_SLOTS
  if (a.ssq_out) {                                          // wave-uniform
    if constexpr (KS == 1) {
      const float ws = wave_sum(sq);
      if (lane == 0) a.ssq_out[blockIdx.x * NW + wave] = ws;
    } else if (wave == 0) {                                 // the row threads of the loop above all sit in wave 0 (host: NWO * bpo * R <= 64)
      const float ws = wave_sum(sq);
      if (lane == 0) a.ssq_out[blockIdx.x] = ws;
    }
  }
}

// ---- host side -------------------------------------------------------------------------------------------------------------
// A plan fixes the geometry of one launch: which instantiation, how many workgroups, how its batches are dealt to the row owners.
struct DgPlan {
  int P, KS, NW, R, D, grid, bpo, apw, nbatch;
  int ssq_n;        // partial sums of y^2 an ssq_out launch writes (0: this geometry cannot)
  size_t shmem;
};
// Geometry for (bits, N, K, group, epilogue); false when the launch is outside what dgemv covers (the caller keeps gemv_kernel).
// rows_per_batch: 0 = choose, else 2 / 4 (what the caller packed the (scale|zero) companion for).
bool dg_plan(int bits, int N, int K, int group, int epi, bool norm, int rows_per_batch, DgPlan* pl);

template <int BITS>
hipError_t launch_dgemv(const DgArgs& a, const DgPlan& pl, int epi, bool bf16, hipStream_t stream);

bool dg_late_default();     // GBX_DG_LATE (A/B switch, default 1)
// dgemv_mx.hip: the same launches with the dot products on the FP4 x FP8 matrix cores (2 bits, group 128, bf16, K = 2048 SP)
bool dg_mx_wanted(const DgArgs& a, const DgPlan& pl, int epi, bool bf16);
hipError_t launch_dgemv_mx(const DgArgs& a, const DgPlan& pl, int epi, hipStream_t stream);
hipError_t launch_dg_pack(const void* s0, const void* z0, const void* s1, const void* z1, uint32_t* out, int N, int G, int R, hipStream_t s);

#define GBX_DG_CASE_D(BITS, P_, KS_, NW_, R_, EPI_, D_)                                                                                  \
  if (pl.P == P_ && pl.KS == KS_ && pl.NW == NW_ && pl.R == R_ && epi == EPI_ && pl.D == D_) {                                  \
    constexpr bool L_ = (KS_ == 1) && (EPI_ != 1);     /* the norm'd launches are the plain / SwiGLU ones of a block */                          \
    if (late && L_) {                                                                                                                         \
      if (bf16) hipLaunchKernelGGL((dgemv_kernel<BITS, P_, KS_, NW_, R_, EPI_, D_, L_, true>), grid, dim3(NW_ * 64), pl.shmem, stream, args); \
      else hipLaunchKernelGGL((dgemv_kernel<BITS, P_, KS_, NW_, R_, EPI_, D_, L_, false>), grid, dim3(NW_ * 64), pl.shmem, stream, args);     \
    } else {                                                                                                                                  \
      if (bf16) hipLaunchKernelGGL((dgemv_kernel<BITS, P_, KS_, NW_, R_, EPI_, D_, false, true>), grid, dim3(NW_ * 64), pl.shmem, stream, args);   \
      else hipLaunchKernelGGL((dgemv_kernel<BITS, P_, KS_, NW_, R_, EPI_, D_, false, false>), grid, dim3(NW_ * 64), pl.shmem, stream, args);       \
    }                                                                                                                                         \
    return hipGetLastError();                                                                                                                 \
  }
#define GBX_DG_CASE(BITS, P_, KS_, NW_, R_, EPI_) GBX_DG_CASE_D(BITS, P_, KS_, NW_, R_, EPI_, 1) GBX_DG_CASE_D(BITS, P_, KS_, NW_, R_, EPI_, 2)
// 4-deep ring: the one-pass, 16-wave, two-row launches only (gate|up of the 2-bit layers: 3.5 batches per wave, all in flight)
#define GBX_DG_CASE_D4(BITS) GBX_DG_CASE_D(BITS, 1, 1, 16, 2, 0, 4) GBX_DG_CASE_D(BITS, 1, 1, 16, 2, 1, 4) GBX_DG_CASE_D(BITS, 1, 1, 16, 2, 2, 4) \
  GBX_DG_CASE_D(BITS, 1, 1, 16, 2, 0, 3) GBX_DG_CASE_D(BITS, 1, 1, 16, 2, 1, 3) GBX_DG_CASE_D(BITS, 1, 1, 16, 2, 2, 3)
#define GBX_DG_EPIS_KS1(BITS, P_, NW_, R_) GBX_DG_CASE(BITS, P_, 1, NW_, R_, 0) GBX_DG_CASE(BITS, P_, 1, NW_, R_, 1) GBX_DG_CASE(BITS, P_, 1, NW_, R_, 2)
#define GBX_DG_EPIS_KS4(BITS, P_, NW_, R_) GBX_DG_CASE(BITS, P_, 4, NW_, R_, 0) GBX_DG_CASE(BITS, P_, 4, NW_, R_, 1)

// one TU per bit width (they compile in parallel)
#define GBX_DEFINE_LAUNCH_DGEMV(BITS)                                                                          \
  template <> hipError_t launch_dgemv<BITS>(const DgArgs& a, const DgPlan& pl, int epi, bool bf16, hipStream_t stream) { \
    DgArgs args = a;                                                                                           \
    args.nbatch = pl.nbatch; args.bpo = pl.bpo; args.apw = pl.apw;                                             \
    const bool late = a.norm_w != nullptr && dg_late_default();                                                \
    const dim3 grid(pl.grid);                                                                                  \
    GBX_DG_CASE_D4(BITS)                                                                                       \
    GBX_DG_EPIS_KS1(BITS, 1, 4, 2) GBX_DG_EPIS_KS1(BITS, 1, 8, 2) GBX_DG_EPIS_KS1(BITS, 1, 16, 2)              \
    GBX_DG_EPIS_KS1(BITS, 1, 4, 4) GBX_DG_EPIS_KS1(BITS, 1, 8, 4) GBX_DG_EPIS_KS1(BITS, 1, 16, 4)              \
    GBX_DG_EPIS_KS1(BITS, 2, 4, 2) GBX_DG_EPIS_KS1(BITS, 2, 8, 2) GBX_DG_EPIS_KS1(BITS, 2, 16, 2)              \
    GBX_DG_EPIS_KS4(BITS, 1, 8, 2) GBX_DG_EPIS_KS4(BITS, 1, 16, 2)                                             \
    GBX_DG_EPIS_KS4(BITS, 1, 8, 4) GBX_DG_EPIS_KS4(BITS, 1, 16, 4)                                             \
    GBX_DG_EPIS_KS4(BITS, 2, 8, 2) GBX_DG_EPIS_KS4(BITS, 2, 16, 2)                                             \
    return hipErrorInvalidValue;                                                                               \
  }

}  // namespace gbx
