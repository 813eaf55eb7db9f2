// Decode GEMV with the activation prologue OFF the weight stream's critical path (round 3).
//
// Same arithmetic, same row -> wave mapping and same epilogues as gemv_kernel (gemv_impl.h; role: the M == 1 case of
// mx.quantized_matmul at gbx_lm/models/quantized_linear_gba.py:195-203 inside the decode block, qllama.py:76-115), but the
// waves of a workgroup have two roles:
//   * PW "prologue" waves fetch x (and the RMSNorm weight) as their FIRST memory instructions, reduce the statistics
//     among themselves (LDS words + spin, no workgroup barrier), write the pre-shifted fp16 image to LDS and raise a flag;
//   * every other wave requests ALL of its row batches at once (D batches of 2 rows in flight per wave) and only then
//     waits for the flag.
// In gemv_kernel every wave sat in the prologue (1.5-2.3 us: x round trip, statistics, image, two barriers) with ONE batch
// in flight, so a quarter of the launch streamed 7 of 33 MB; requesting more ahead of the prologue back-pressured the
// in-order vector-memory path and delayed x itself ("tried and rejected").  With the roles split the x
// loads are at the head of the CU's queue and the loader waves may block at issue for as long as they like.
// Stamps of the 5-launch layer that motivated it: profiles/r03_layer_timeline_before.txt.
//
// Restrictions (the decode runner's launches): packed scale|zero companion (GemvArgs::sz), one activation row, no
// expert gather, group size 64 / 128.  Everything else stays on gemv_kernel.
#pragma once
#include "gemv_impl.h"

namespace gbx {

// Flag words live in LDS and are accessed through an EXPLICIT local-address-space pointer: a volatile access through a generic
// pointer is not address-space-inferred by hipcc, becomes a flat_load and drags an `s_waitcnt vmcnt(0)` into the spin loop --
// i.e. the loader waves would wait for every prefetched row before they look at the flag (seen in the ISA, measured: 691 tok/s).
typedef __attribute__((address_space(3))) volatile unsigned lds_flag_t;
__device__ __forceinline__ lds_flag_t* lds_flag_ptr(void* generic) { return (lds_flag_t*)generic; }
// spin until the first `n` flag words are all non-zero (wave-uniform result)
__device__ __forceinline__ void lds_wait_flags(lds_flag_t* f, int n) {
  for (;;) {
    unsigned all = 1;
    for (int i = 0; i < n; ++i) all &= f[i];
    if (__builtin_amdgcn_readfirstlane(all) != 0) break;
    __builtin_amdgcn_s_sleep(1);
  }
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void lds_raise_flag(lds_flag_t* f) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's LDS writes have landed
  *f = 1u;
}

// D  = batches in flight per owner = ring depth; batches_per_owner is a multiple of D (== D for the decode shapes)
// PW = prologue waves (waves 0 .. PW-1)
template <int BITS, int P, int KS, int D, bool BF16, int NW, int PW>
__global__ __launch_bounds__(NW * 64) void gemv_ov_kernel(const GemvArgs a) {
  constexpr int NT = NW * 64, PT = PW * 64;
  constexpr int GPB = NW / 4;
  using F = Fmt<BITS>;
  using L = Lds<BITS>;
  constexpr int PAIRS = L::PAIRS, SSTR = L::SSTR, NSUB = F::NSUB;
  constexpr int R = GEMV_R;
  constexpr int UPS = F::VPL / F::UNIT_VALS;
  constexpr int UPT = (P * KS * 64 * UPS + PT - 1) / PT;     // units per prologue thread
  constexpr int UW = F::UNIT_VALS / 2;
  constexpr int UPSUB = (32 / F::UNIT_VALS) ? (32 / F::UNIT_VALS) : 1;
  static_assert(PW >= 1 && PW <= 8 && PW < NW, "prologue waves");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int K = a.K, N = a.N;
  const int nslots = K / F::VPL;
  h2* xp_lds = reinterpret_cast<h2*>(smem);
  float2* xs_lds = reinterpret_cast<float2*>(smem + static_cast<size_t>(nslots) * SSTR * 4);
  float* red_lds = reinterpret_cast<float*>(xs_lds + nslots * NSUB);     // PW floats used
  lds_flag_t* flags = lds_flag_ptr(red_lds + NW);   // [0, 8): statistics posted, [8, 16): image written, [16, 24): x landed
  float* part_lds = red_lds + NW + 24;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid < 24) flags[tid] = 0u;
  __syncthreads();                                  // the only barrier ahead of the stream: nothing is in flight yet

  const int W = K / 32 * BITS;
  const int G = K >> a.log2g;
  const unsigned short* __restrict__ xg = static_cast<const unsigned short*>(a.x);
  const int gshift = a.log2g - 5;
  const bool swiglu = a.swiglu != 0;
  const int rowstep = swiglu ? 1 : R;
  const int bpo = a.batches_per_owner;
  const int kq = wave & 3, rg = wave >> 2;
  const int owner = KS == 1 ? (blockIdx.x * NW + wave) : (blockIdx.x * GPB + rg);
  const int owners = static_cast<int>(gridDim.x) * (KS == 1 ? NW : GPB);
  auto batch_of = [&](int j) { return owner + j * owners; };
  const int slot0 = (KS == 1 ? 0 : kq * 64) + lane;
  constexpr int PSTRIDE = 64 * KS;
  const int n0 = a.seg[0].N, n01 = n0 + a.seg[1].N;

  constexpr int RS_FLAGS = 0x00020000, RS_ALL = 0x7FFFFFFF;
  auto rsrc = [](const void* ptr) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, RS_ALL, RS_FLAGS); };
  const uint32_t* const q0 = a.seg[0].qweight; const uint32_t* const q1 = a.seg[1].qweight; const uint32_t* const q2 = a.seg[2].qweight;
  unsigned short* const y0 = a.seg[0].y; unsigned short* const y1 = a.seg[1].y; unsigned short* const y2 = a.seg[2].y;
  const int n1 = a.seg[1].N, n2 = a.seg[2].N;
  auto locate = [&](int r0, int i, int& sg, int& row) {
    if (swiglu) { sg = i; row = min(r0, N - 1); }
    else {
      const int r = min(r0 + i, N - 1);
      sg = (r >= n0) + (r >= n01);
      row = r - (sg == 0 ? 0 : (sg == 1 ? n0 : n01));
    }
  };
#ifdef GBX_STAMPS
#define GBX_OV_STAMP(i) do { if (lane == 0 && a.stamps) a.stamps[(static_cast<size_t>(blockIdx.x) * NW + wave) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define GBX_OV_STAMP(i) do {} while (0)
#endif
  const __amdgpu_buffer_rsrc_t resd = rsrc(a.residual ? static_cast<const void*>(a.residual) : static_cast<const void*>(y0));
  const __amdgpu_buffer_rsrc_t pd = rsrc(a.sz);
  int voff_w[P], voff_s[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int slot = min(p * PSTRIDE + slot0, nslots - 1);
    voff_w[p] = slot * (F::SLOT_WORDS * 4);
    voff_s[p] = ((slot * NSUB) >> gshift) * 8;
  }
  // one batch: R packed rows (non-temporal: streamed once), both rows' (scale | zero) pairs, both residuals
  auto issue = [&](RowBatch<BITS, P>& b, int r0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < R; ++i) {
      int sgi, row;
      locate(r0, i, sgi, row);
      const __amdgpu_buffer_rsrc_t qd = rsrc(sgi == 0 ? q0 : (sgi == 1 ? q1 : q2));
      const int woff = row * (W * 4);
#pragma unroll
      for (int p = 0; p < P; ++p) {
        if constexpr (BITS == 3) b.w[i][p] = __builtin_amdgcn_raw_buffer_load_b96(qd, voff_w[p], woff, 2);
        else b.w[i][p] = __builtin_amdgcn_raw_buffer_load_b128(qd, voff_w[p], woff, 2);
      }
    }
    const int rc = min(r0, N - rowstep);
    const int poff = (swiglu ? rc : (rc >> 1)) * (G * 8);
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const u32x2 v0 = __builtin_amdgcn_raw_buffer_load_b64(pd, voff_s[p], poff, 0);
      b.szp[p][0][0] = v0.x; b.szp[p][0][1] = v0.y;
    }
    b.res2 = __builtin_amdgcn_raw_buffer_load_b32(resd, 0, min(rc, n0 - 2) * 2, 0);
  };
  RowBatch<BITS, P> ring[D];
  GBX_OV_STAMP(0);
#ifdef GBX_STAMPS
  if (lane == 0 && a.stamps)
    a.stamps[(static_cast<size_t>(blockIdx.x) * NW + wave) * 16 + 7] =
        (static_cast<unsigned long long>(__builtin_amdgcn_s_getreg((31 << 11) | 20)) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 4);
#endif

  auto finish = [&](int r0, float t0, float t1, unsigned short res0, unsigned short res1) __attribute__((always_inline)) {
    if (swiglu) {
      if (r0 < N) {
        const float g = round_bf16_or_f16(t0, BF16), up = round_bf16_or_f16(t1, BF16);
        const float sl = round_bf16_or_f16(g / (1.f + __expf(-g)), BF16);
        y0[r0] = f32_to_act<BF16>(sl * up);
      }
      return;
    }
    const float tv[2] = {t0, t1};
    const unsigned short rv[2] = {res0, res1};
#pragma unroll
    for (int i = 0; i < R; ++i) {
      if (r0 + i < N) {
        int sgi, row;
        locate(r0, i, sgi, row);
        float v = tv[i];
        if (a.residual) v = act_to_f32<BF16>(rv[i]) + round_bf16_or_f16(v, BF16);
        unsigned short* ysel = sgi == 0 ? y0 : (sgi == 1 ? y1 : y2);
        ysel[row] = f32_to_act<BF16>(v);
      }
    }
  };

  // ---- everything after a wave's requests are out: wait for the image, keep the lane's activations, stream ----
  auto rest = [&]() __attribute__((always_inline)) {
    lds_wait_flags(flags + 8, PW);
    GBX_OV_STAMP(2);
    h2 xv[P][PAIRS];
    float xs[P][NSUB], cf[P][NSUB];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int slot = p * PSTRIDE + slot0;
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
    auto consume = [&](const RowBatch<BITS, P>& cur, int j) __attribute__((always_inline)) {
      const int b = batch_of(j);
      float acc[R];
#pragma unroll
      for (int i = 0; i < R; ++i) {
        acc[i] = 0.f;
#pragma unroll
        for (int p = 0; p < P; ++p) {
          float d[NSUB];
          dot_slot<BITS>(cur.w[i][p], xv[p], d);
          const float sc0 = sz_scale<BF16>(cur.szp[p][0][i]), ze0 = sz_zero<BF16>(cur.szp[p][0][i]);
          if constexpr (NSUB == 2) {
            const float t = fmaf(cf[p][1], d[1], cf[p][0] * d[0]);
            acc[i] = fmaf(sc0, t, acc[i]);
            acc[i] = fmaf(ze0, xs[p][0] + xs[p][1], acc[i]);
          } else {
            const float sc = sc0 * cf[p][0];
            acc[i] = fmaf(sc, d[0], acc[i]);
            acc[i] = fmaf(ze0, xs[p][0], acc[i]);
          }
        }
      }
      const float tot = wave_sum2(acc[0], acc[1]);
      const float t0 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(tot), 31));
      const float t1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(tot), 63));
      if constexpr (KS == 1) {
        if (lane == 0) finish(b * rowstep, t0, t1, static_cast<unsigned short>(cur.res2 & 0xFFFFu), static_cast<unsigned short>(cur.res2 >> 16));
      } else {
        if (lane == 0) {
          float* pp = part_lds + static_cast<size_t>(rg * bpo + j) * (R * 4) + kq;
          pp[0] = t0;
          pp[4] = t1;
        }
      }
    };
    GBX_OV_STAMP(3);
    const int steps = bpo / D;
    int base = 0;
    for (int st = 1; st < steps; ++st) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        consume(ring[d], base + d);
        issue(ring[d], batch_of(base + D + d) * rowstep);
      }
      base += D;
    }
    GBX_OV_STAMP(4);
#pragma unroll
    for (int d = 0; d < D; ++d) {
      consume(ring[d], base + d);
      if (d == 0) GBX_OV_STAMP(5);
    }
    GBX_OV_STAMP(6);
    if constexpr (KS == 4) {
      __syncthreads();
      for (int bl = tid; bl < GPB * bpo; bl += NT) {
        const float4 p0 = *reinterpret_cast<const float4*>(part_lds + static_cast<size_t>(bl) * (R * 4));
        const float4 p1 = *reinterpret_cast<const float4*>(part_lds + static_cast<size_t>(bl) * (R * 4) + 4);
        const int r0 = ((blockIdx.x * GPB + bl / bpo) + (bl % bpo) * owners) * rowstep;
        unsigned short res0 = 0, res1 = 0;
        if (a.residual && !swiglu) {
          if (r0 < N) res0 = a.residual[r0];
          if (r0 + 1 < N) res1 = a.residual[r0 + 1];
        }
        finish(r0, (p0.x + p0.y) + (p0.z + p0.w), (p1.x + p1.y) + (p1.z + p1.w), res0, res1);
      }
    }
  };

  if (wave < PW) {
    // ================= prologue wave: x first, one batch, the image, then the rest of its batches =================
    const int nunits = K / F::UNIT_VALS;
    const bool norm = (KS == 1) && a.norm_w != nullptr;
    const unsigned short* __restrict__ wg = norm ? a.norm_w : xg;
    constexpr int WPT = (KS == 1) ? UPT : 1;
    unsigned xr[UPT][UW], wr[WPT][UW];
#pragma unroll
    for (int j = 0; j < UPT; ++j) {
      const int u = min(tid + PT * j, nunits - 1);
      const unsigned* src = reinterpret_cast<const unsigned*>(xg + static_cast<size_t>(u) * F::UNIT_VALS);
      const unsigned* wsrc = reinterpret_cast<const unsigned*>(wg + static_cast<size_t>(u) * F::UNIT_VALS);
#pragma unroll
      for (int q = 0; q < UW; q += 4) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(src + q);
        xr[j][q] = v.x; xr[j][q + 1] = v.y; xr[j][q + 2] = v.z; xr[j][q + 3] = v.w;
        if constexpr (KS == 1) {
          const u32x4 t = *reinterpret_cast<const u32x4*>(wsrc + q);
          wr[j][q] = t.x; wr[j][q + 1] = t.y; wr[j][q + 2] = t.z; wr[j][q + 3] = t.w;
        }
      }
    }
    // x is in registers: only now may the loader waves of this workgroup flood the memory pipeline (a burst ahead of x delays
    // x itself -- measured: image ready at 4.9 us instead of 3.1 -- here and in every workgroup that starts later)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) flags[16 + wave] = 1u;
    issue(ring[0], batch_of(0) * rowstep);
    GBX_OV_STAMP(1);
    float rstd = 1.f;
    if (norm) {
      float ssq = 0.f;
#pragma unroll
      for (int j = 0; j < UPT; ++j) {
        const bool live = (tid + PT * j) < nunits;
#pragma unroll
        for (int q = 0; q < UW; ++q) {
          const float lo = act_lo<BF16>(xr[j][q]), hi = act_hi<BF16>(xr[j][q]);
          ssq += live ? (lo * lo + hi * hi) : 0.f;
        }
      }
      ssq = wave_sum(ssq);
      if (lane == 0) {
        red_lds[wave] = ssq;
        lds_raise_flag(flags + wave);
      }
      lds_wait_flags(flags, PW);
      float tot = 0.f;
#pragma unroll
      for (int i = 0; i < PW; ++i) tot += red_lds[i];
      rstd = rsqrtf(tot / static_cast<float>(K) + a.eps);
    }
#pragma unroll
    for (int j = 0; j < UPT; ++j) {
      const int u = tid + PT * j;
      float xf[F::UNIT_VALS];
      unsigned amax = 0;
#pragma unroll
      for (int q = 0; q < UW; ++q) {
        float lo = act_lo<BF16>(xr[j][q]), hi = act_hi<BF16>(xr[j][q]);
        if constexpr (KS == 1) {
          if (norm) {
            lo = round_bf16_or_f16(lo * rstd * act_lo<BF16>(wr[j][q]), BF16);
            hi = round_bf16_or_f16(hi * rstd * act_hi<BF16>(wr[j][q]), BF16);
          }
        }
        xf[2 * q] = lo;
        xf[2 * q + 1] = hi;
        amax = max(amax, max(__float_as_uint(lo) & 0x7FFFFFFFu, __float_as_uint(hi) & 0x7FFFFFFFu));
      }
      float usum = 0.f;
#pragma unroll
      for (int q = 0; q < F::UNIT_VALS; ++q) usum += xf[q];
      float ssum;
      if constexpr (BITS == 2) { ssum = pair_sum(usum); amax = max(amax, dpp_u32<0xB1>(amax)); }
      else if constexpr (BITS == 4) {
        ssum = quad_sum(usum);
        amax = max(amax, dpp_u32<0xB1>(amax));
        amax = max(amax, dpp_u32<0x4E>(amax));
      } else ssum = usum;
      int E = 141 - static_cast<int>(amax >> 23);
      E = min(max(E, -100), 110);
      if (u < nunits) {
        const int slot = u / UPS, uin = u % UPS;
        h2* dst = xp_lds + static_cast<size_t>(slot) * SSTR + uin * F::UNIT_PAIRS;
#pragma unroll
        for (int i = 0; i < F::UNIT_PAIRS; ++i) {
          const float sc = exp2i(E - F::pos(i));
          h2 v;
          v.x = static_cast<_Float16>(xf[F::ka(i)] * sc);
          v.y = static_cast<_Float16>(xf[F::kb(i)] * sc);
          dst[i] = v;
        }
        if ((u % UPSUB) == 0) xs_lds[u / UPSUB] = make_float2(ssum, exp2i(24 - E));
      }
    }
    if (lane == 0) lds_raise_flag(flags + 8 + wave);
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int d = 1; d < D; ++d) issue(ring[d], batch_of(d) * rowstep);
    rest();
  } else {
    // ================= loader wave: every batch of the first ring round at once =================
    issue(ring[0], batch_of(0) * rowstep);
    if constexpr (D > 1) {
      lds_wait_flags(flags + 16, PW);
#pragma unroll
      for (int d = 1; d < D; ++d) issue(ring[d], batch_of(d) * rowstep);
    }
    GBX_OV_STAMP(1);
    rest();
  }
}

template <int BITS, int P, int KS, int NW>
static size_t gemv_ov_shmem(int K, int bpo) {
  size_t b = Lds<BITS>::bytes(K) + (NW + 24) * sizeof(float) + 16;
  if (KS == 4) b += static_cast<size_t>(NW / 4) * bpo * GEMV_R * 4 * sizeof(float);
  return b;
}

template <int BITS, int P, int KS, int D, int NW, int PW>
static hipError_t launch_gemv_ov_d(const GemvArgs& a, bool bf16, int bpo, int nbatches, hipStream_t stream) {
  GemvArgs args = a;
  args.batches_per_owner = bpo;
  const size_t shmem = gemv_ov_shmem<BITS, P, KS, NW>(a.K, bpo);
  const int batches_per_block = (KS == 1 ? NW : NW / 4) * bpo;
  const int blocks = (nbatches + batches_per_block - 1) / batches_per_block;
  dim3 grid(blocks), block(NW * 64);
  if (bf16) hipLaunchKernelGGL((gemv_ov_kernel<BITS, P, KS, D, true, NW, PW>), grid, block, shmem, stream, args);
  else hipLaunchKernelGGL((gemv_ov_kernel<BITS, P, KS, D, false, NW, PW>), grid, block, shmem, stream, args);
  return hipGetLastError();
}

// ring depth for a geometry: every batch of an owner in flight when that is at most 4 (2 at two passes per row), else a
// 2-deep ring that streams like gemv_kernel
template <int BITS, int P, int KS, int NW>
static hipError_t launch_gemv_ov_nw(const GemvArgs& a, bool bf16, int nbatches, hipStream_t stream) {
  constexpr int PW = 4;
  const int opb = (KS == 1) ? NW : NW / 4;
  const int owners = GEMV_CUS * opb;                    // one workgroup per CU
  int bpo = (nbatches + owners - 1) / owners;
  constexpr int DMAX = (P == 1) ? 4 : 2;
  if (bpo <= 1) return launch_gemv_ov_d<BITS, P, KS, 1, NW, PW>(a, bf16, 1, nbatches, stream);
  if (bpo <= 2) return launch_gemv_ov_d<BITS, P, KS, 2, NW, PW>(a, bf16, 2, nbatches, stream);
  if constexpr (DMAX >= 4) {
    if (bpo <= 4) return launch_gemv_ov_d<BITS, P, KS, 4, NW, PW>(a, bf16, 4, nbatches, stream);
  }
  bpo = ((bpo + 1) / 2) * 2;
  if (KS == 4 && bpo > GEMV_MAX_BATCHES_PER_GROUP_KS4) bpo = GEMV_MAX_BATCHES_PER_GROUP_KS4;
  return launch_gemv_ov_d<BITS, P, KS, 2, NW, PW>(a, bf16, bpo, nbatches, stream);
}

// workgroup width: the widest of 16 / 12 / 8 waves that still gives (nearly) every CU a workgroup
template <int BITS, int P, int KS>
static hipError_t launch_gemv_ov_p(const GemvArgs& a, bool bf16, hipStream_t stream) {
  const int rowstep = a.swiglu ? 1 : GEMV_R;
  const int nbatches = (a.N + rowstep - 1) / rowstep;
  const int per_wave = KS == 1 ? 1 : 4;                 // waves that share one batch
  const int waves = nbatches * per_wave;                // waves needed at one batch each
  if constexpr (P == 1) {                               // two passes per row: 16 waves x 2 batches in flight do not fit 128 VGPRs
    if (waves >= GEMV_CUS * 14) return launch_gemv_ov_nw<BITS, P, KS, 16>(a, bf16, nbatches, stream);
  }
  if (waves >= GEMV_CUS * 10) return launch_gemv_ov_nw<BITS, P, KS, 12>(a, bf16, nbatches, stream);
  return launch_gemv_ov_nw<BITS, P, KS, 8>(a, bf16, nbatches, stream);
}

// true when the launch fits this kernel (the caller falls back to gemv_kernel otherwise)
inline bool gemv_ov_eligible(const GemvArgs& a, int M) {
  static const bool off = [] { const char* e = getenv("GBX_GEMV_OV"); return e && e[0] == '0'; }();
  if (off || M != 1 || !a.sz || a.expert_idx || a.bias || a.log2g < 6) return false;
  const int rowstep = a.swiglu ? 1 : GEMV_R;
  const int nbatches = (a.N + rowstep - 1) / rowstep;
  return nbatches >= 1024;                              // tiny matrices: the 4-wave geometry of gemv_kernel
}

template <int BITS>
static hipError_t dispatch_gemv_ov(const GemvArgs& a, bool bf16, hipStream_t s) {
  const int nslots = a.K / Fmt<BITS>::VPL;
  if (nslots <= 64) return launch_gemv_ov_p<BITS, 1, 1>(a, bf16, s);
  if (nslots <= 128) return launch_gemv_ov_p<BITS, 2, 1>(a, bf16, s);
  if (nslots <= 256) return launch_gemv_ov_p<BITS, 1, 4>(a, bf16, s);
  if (nslots <= 512) return launch_gemv_ov_p<BITS, 2, 4>(a, bf16, s);
  return hipErrorInvalidValue;
}

#undef GBX_DEFINE_LAUNCH_GEMV
#define GBX_DEFINE_LAUNCH_GEMV(BITS)                                                              \
  template <> hipError_t launch_gemv<BITS>(const GemvArgs& a, int M, bool bf16, hipStream_t s) { \
    if (gemv_ov_eligible(a, M)) return dispatch_gemv_ov<BITS>(a, bf16, s);                        \
    return dispatch_gemv<BITS>(a, M, bf16, s);                                                    \
  }

}  // namespace gbx
