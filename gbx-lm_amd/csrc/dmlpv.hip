// The MLP half of a decoder block at M == 1 as ONE launch for 3- and 4-bit rows ("dmlpv", round 6): dmlp.hip's structure --
//
//   x' = round(h + round(o_proj sums))  ->  RMSNorm(x')  ->  gate | up rows of ONE eighth of the intermediate vector + SwiGLU
//      ->  XCD-local hand-off of those I / 8 activations  ->  down_proj over exactly that eighth of its input columns
//      ->  fixed-point partial sums into acc_out[D]
//
// -- on the vector-ALU arithmetic of the decode GEMV (gemv_fmt.h: in-place field masks as fp16 subnormals, v_dot2, per-sub-slot block
// exponent, s * sum(q x) + z * sum(x)), which covers every width and group size; dmlp.hip's FP4 x FP8 matrix-core form only exists for 2-bit
// fields.  Replaces the gate|up and down launches of the three-launch block:
//   mx.quantized_matmul at M == 1 (gbx_lm/models/quantized_linear_gba.py:195-203) at the call sites of gbx_lm/models/qllama.py:115
//   (down_proj(silu(gate_proj(x)) * up_proj(x))) with the post-attention RMSNorm (:139) and the residual adds (:138, :140) as in dmlp.hip.
// What is different from dmlp.hip besides the arithmetic: down_proj's packed words do not fit registers at 4 bits (8 rows x 16 bytes per
// lane on top of the activation image), so they stream into LDS by LDS-DMA (<= 112 KB per CU, requested behind each wave's last gate / up
// rows, no register held) and are consumed from there, one lane = one 32-value slot of one row (dattn's o_proj form).
// Covered: 3 or 4 bits (gate, up, down alike), groups 32 / 64 / 128, bf16; D = 4096, I = 14336 (Llama-3-8B) and D = 3072, I = 8192
// (Llama-3.2-3B): I / 8 a multiple of 128 with I / 256 even in [32, 64], D / 512 in {6, 8}.
#include "dattn_kernel.h"
#include "gbx_env.h"

namespace gbx {

typedef __attribute__((address_space(1))) unsigned long long dv_gu64;
typedef __attribute__((address_space(1))) unsigned dv_gu32;

constexpr int DV_LDS_IMG1 = 0;                     // gate|up activation image: Lds<B>::bytes(4096) <= 11.3 KB
constexpr int DV_LDS_RED = 12 * 1024;              // 16 floats
constexpr int DV_LDS_P3 = 12 * 1024 + 256;         // down_proj: 4 KB of raw activations (3 bits), then its image (<= 5.2 KB)
constexpr int DV_LDS_DW = 24 * 1024;               // down_proj's packed words: D / 32 rows x (I / 8 values) <= 112 KB
constexpr int DV_LDS_BYTES = 138 * 1024;

template <int B, int P, bool BF16>
__global__ __launch_bounds__(DA_NT) void dmlpv_kernel(const DmlpArgs a) {
  using F = Fmt<B>;
  using L = Lds<B>;
  using WT = typename SlotWords<B>::type;
  constexpr int NT = DA_NT, NW = DA_NW, R = 2;
  constexpr int PAIRS = L::PAIRS, SSTR = L::SSTR, NSUB = F::NSUB;
  static_assert(NSUB == 1, "3- and 4-bit slots");
  constexpr unsigned KILL = 0x80000000u;
  constexpr int RS_FLAGS = 0x00020000, RS_ALL = 0x7FFFFFFF;
  constexpr int SLOTB = B == 3 ? 12 : 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int K = a.K;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef GBX_STAMPS
#define DV_STAMP(i) do { if (lane == 0 && a.stamps) a.stamps[(static_cast<size_t>(blockIdx.x) * NW + wave) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define DV_STAMP(i) do {} while (0)
#endif
  DV_STAMP(0);
  const int g = static_cast<int>(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u);     // HW_REG_XCC_ID: this XCD = this eighth of the intermediate vector
  const int r = static_cast<int>(blockIdx.x) >> 3;
  dv_gu32* gen_p = reinterpret_cast<dv_gu32*>(reinterpret_cast<uintptr_t>(a.state)) + g * 16;
  dv_gu64* hand = reinterpret_cast<dv_gu64*>(reinterpret_cast<uintptr_t>(a.hand)) + static_cast<size_t>(g) * DA_HAND_STRIDE;
  auto rsrc = [](const void* ptr) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, RS_ALL, RS_FLAGS); };
  auto timeout = [&](unsigned code) {
    if (lane == 0) {
      a.state[8 * 16] = code;
      __hip_atomic_fetch_add(reinterpret_cast<dv_gu64*>(reinterpret_cast<uintptr_t>(a.acc_out)) + a.D, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  auto sweep = [&](dv_gu64* p, unsigned tag, unsigned code) -> unsigned {
    unsigned long long x;
    for (unsigned spins = 0;;) {
      x = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__all(static_cast<unsigned>(x >> 32) == tag)) break;
      if (++spins > a.spin_limit) { timeout(code); break; }
      __builtin_amdgcn_s_sleep(1);
    }
    return static_cast<unsigned>(x);
  };
  auto sweep2 = [&](dv_gu64* pa, dv_gu64* pb, unsigned tag, unsigned code, unsigned& va, unsigned& vb) {
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

  const int nslots = K / F::VPL;                                // 128 (K = 4096) or 96 (K = 3072)
  h2* xp_lds = reinterpret_cast<h2*>(smem + DV_LDS_IMG1);
  float2* xs_lds = reinterpret_cast<float2*>(smem + DV_LDS_IMG1 + static_cast<size_t>(nslots) * SSTR * 4);
  float* red_lds = reinterpret_cast<float*>(smem + DV_LDS_RED);
  const int W4 = K / 32 * B * 4;                                // bytes per packed gate / up row
  const int G = K >> a.gu_log2g;
  const int gshift = a.gu_log2g - 5;
  const int IX = a.I >> 3;                                      // activations of this XCD
  // this wave's gate | up batches: activations [n0, n0 + nloc) of the XCD, consecutive so that two of them share a granule
  const int nloc = wave < a.a4 ? 4 : 2;
  const int nx0 = r * a.bpw + (wave < a.a4 ? 4 * wave : 4 * a.a4 + 2 * (wave - a.a4));
  const int n0 = g * IX + nx0;
  const __amdgpu_buffer_rsrc_t qd = rsrc(a.gate_qw), qd2 = rsrc(a.up_qw), szd = rsrc(a.gu_sz);
  unsigned voff_w[P], voff_s[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int slot = p * 64 + lane;
    const unsigned dead = slot < nslots ? 0u : KILL;
    voff_w[p] = static_cast<unsigned>(slot * SLOTB) | dead;
    voff_s[p] = static_cast<unsigned>((slot >> gshift) * (R * 4)) | dead;
  }
  struct Batch { WT w[R][P]; unsigned sz[P][R]; };
  auto issue = [&](Batch& b, int j) {
    const unsigned kill = j < nloc ? 0u : KILL;
    const int row = n0 + (j < nloc ? j : 0);
#pragma unroll
    for (int p = 0; p < P; ++p) {
      if constexpr (B == 3) {
        b.w[0][p] = __builtin_amdgcn_raw_buffer_load_b96(qd, voff_w[p] | kill, row * W4, 2);
        b.w[1][p] = __builtin_amdgcn_raw_buffer_load_b96(qd2, voff_w[p] | kill, row * W4, 2);
      } else {
        b.w[0][p] = __builtin_amdgcn_raw_buffer_load_b128(qd, voff_w[p] | kill, row * W4, 2);
        b.w[1][p] = __builtin_amdgcn_raw_buffer_load_b128(qd2, voff_w[p] | kill, row * W4, 2);
      }
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(szd, voff_s[p] | kill, row * (G * R * 4), 0);
      b.sz[p][0] = v.x; b.sz[p][1] = v.y;
    }
  };

  // ---------------- 1. prologue on waves 0-7 (dattn_kernel.h, vector form): x' from the fixed-point sums, statistic, fp16 image ----------------
  constexpr int UP = F::UNIT_PAIRS;
  constexpr int PPU = (B == 3) ? UP : 2;
  constexpr int TPU = UP / PPU;
  constexpr int PW = 8, PT = PW * 64;
  constexpr int IPT = (P * 64 * (F::VPL / 2) + PT * PPU - 1) / (PT * PPU);
  const int nitems = (K / F::UNIT_VALS) * TPU;
  const bool pro = wave < PW;
  unsigned xr[IPT][PPU], wr[IPT][PPU];
  if (a.zero_buf) {
    for (int i = static_cast<int>(blockIdx.x) * NT + tid; i < a.zero_n; i += static_cast<int>(gridDim.x) * NT) a.zero_buf[i] = 0ull;
  }
  if (pro) {
    u32x4 acv[IPT][PPU];
    const __amdgpu_buffer_rsrc_t ad = rsrc(a.x_acc);
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
      const int it = min(tid + PT * j, nitems - 1);
      const int u = it / TPU, t = it % TPU;
#pragma unroll
      for (int q = 0; q < PPU; ++q) {
        int idx;
        if constexpr (B == 3) idx = u * F::UNIT_VALS + 2 * q;
        else idx = u * F::UNIT_VALS + PPU * t + (q < PPU / 2 ? 2 * q : UP + 2 * (q - PPU / 2));
        acv[j][q] = __builtin_amdgcn_raw_buffer_load_b128(ad, static_cast<unsigned>(idx * 8), 0, 0);
      }
    }
    const unsigned apoison = __builtin_amdgcn_raw_buffer_load_b32(ad, static_cast<unsigned>(K * 8), 0, 0);
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
      const int it = min(tid + PT * j, nitems - 1);
      da_load_item<B, PPU>(a.x, it, xr[j]);
      da_load_item<B, PPU>(a.norm_w, it, wr[j]);
    }
    DV_STAMP(1);
    const bool poison = apoison != 0u;
    float ssq = 0.f;
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
      const int it = min(tid + PT * j, nitems - 1);
      const int u = it / TPU, t = it % TPU;
      const bool live = tid + PT * j < nitems;
#pragma unroll
      for (int q = 0; q < PPU; ++q) {
        int idx;
        if constexpr (B == 3) idx = u * F::UNIT_VALS + 2 * q;
        else idx = u * F::UNIT_VALS + PPU * t + (q < PPU / 2 ? 2 * q : UP + 2 * (q - PPU / 2));
        const u32x4 av = acv[j][q];
        const float o0 = round_bf16_or_f16(fx32_to_f32(av.x, av.y), BF16);
        const float o1 = round_bf16_or_f16(fx32_to_f32(av.z, av.w), BF16);
        float x0 = act_lo<BF16>(xr[j][q]) + o0, x1 = act_hi<BF16>(xr[j][q]) + o1;
        if (poison) { x0 = __uint_as_float(0x7FC00000u); x1 = x0; }
        const unsigned wv = static_cast<unsigned>(f32_to_act<BF16>(x0)) | (static_cast<unsigned>(f32_to_act<BF16>(x1)) << 16);
        xr[j][q] = wv;
        if (blockIdx.x == 0 && live && a.x_out) *reinterpret_cast<unsigned*>(a.x_out + idx) = wv;
        const float lo = act_lo<BF16>(wv), hi = act_hi<BF16>(wv);
        ssq += live ? fmaf(hi, hi, lo * lo) : 0.f;
      }
    }
    ssq = wave_sum(ssq);
    if (lane == 0) red_lds[wave] = ssq;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();                                               // barrier A: the activations have landed (cue for the late waves' first rows)
  const unsigned gen = __hip_atomic_load(gen_p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  DV_STAMP(9);
  const int RPWG = a.D / DA_MEMBERS;                             // down_proj rows of this workgroup
  const int rows_w = RPWG / NW;                                  // rows of this wave (8 or 6)
  const int W4d = a.I / 32 * B * 4;                              // bytes per packed down row
  const int nsl = IX >> 5;                                       // 32-value slots of this XCD's slice (56 or 32)
  const int SLB = nsl * SLOTB;                                   // bytes of a row's slice in memory
  const int SLL = nsl * 16;                                      // ... and in LDS: `buffer_load_dwordx3 ... lds` lands lane l's 12 bytes at l * 16 (scratch/probe3/probe_dma3.hip)
  unsigned char* dwl = smem + DV_LDS_DW;
  const __amdgpu_buffer_rsrc_t dd = rsrc(a.down_qw), dsd = rsrc(a.down_sz);
  if (pro) {
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < PW; ++i) tot += red_lds[i];
    const float rstd = rsqrtf(tot / static_cast<float>(K) + a.eps);
#pragma unroll
    for (int j = 0; j < IPT; ++j)
      if ((tid & ~63) + PT * j < nitems) da_image_item<B, BF16, PPU>(xr[j], wr[j], true, rstd, tid + PT * j, nitems, xp_lds, xs_lds);
  }
  DV_STAMP(10);
  Batch ring[2];
#ifndef DV_ISSUE_BEHIND_B
  issue(ring[0], 0);
#endif
  __syncthreads();                                               // barrier B: image written
  DV_STAMP(2);
#ifdef DV_ISSUE_BEHIND_B
  issue(ring[0], 0);
#endif
  issue(ring[1], 1);
  unsigned tag = gen + 1u;
  if (tag == 0u) tag = 1u;

  // ---------------- 2. gate | up (dgemv's consume, one (gate, up) row pair per batch), SwiGLU, two activations per granule ----------------
  unsigned short actv[4] = {0, 0, 0, 0};
  {
    h2 xv[P][PAIRS];
    float xs[P], cf[P];
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
      const float2 t = xs_lds[sl];
      xs[p] = live ? t.x : 0.f;
      cf[p] = live ? t.y : 0.f;
    }
    auto consume = [&](const Batch& cur, int j) {
      if (j >= nloc) return;                                     // wave-uniform
      float acc[R] = {0.f, 0.f};
#pragma unroll
      for (int p = 0; p < P; ++p) {
        float d[R][NSUB];
        WT wp[R];
#pragma unroll
        for (int i = 0; i < R; ++i) wp[i] = cur.w[i][p];
        dot_slot_rows<B, R>(wp, xv[p], d);
#pragma unroll
        for (int i = 0; i < R; ++i) {
          const float sc = sz_scale<BF16>(cur.sz[p][i]), ze = sz_zero<BF16>(cur.sz[p][i]);
          const float s2 = sc * cf[p];
          acc[i] = fmaf(s2, d[i][0], acc[i]);
          acc[i] = fmaf(ze, xs[p], acc[i]);
        }
      }
      const float v0 = wave_sum2(acc[0], acc[1]);              // gate in lane 31, up in lane 63
      const float t0 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v0), 31));
      const float t1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v0), 63));
      const float gt = round_bf16_or_f16(t0, BF16), up = round_bf16_or_f16(t1, BF16);
      const float sl = round_bf16_or_f16(gt / (1.f + __expf(-gt)), BF16);
      actv[j] = f32_to_act<BF16>(sl * up);
    };
    auto publish = [&](int q) {                                  // activations nx0 + 2 q, + 1 -> granule (nx0 >> 1) + q (sc0: stays in this XCD's L2)
      if (2 * q >= nloc) return;
      const unsigned val = static_cast<unsigned>(actv[2 * q]) | (static_cast<unsigned>(actv[2 * q + 1]) << 16);
      if (lane == 0) {
        __hip_atomic_store(hand + (nx0 >> 1) + q, (static_cast<unsigned long long>(tag) << 32) | val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (a.act_out) *reinterpret_cast<unsigned*>(a.act_out + n0 + 2 * q) = val;
      }
    };
    consume(ring[0], 0);
    issue(ring[0], 2);
    consume(ring[1], 1);
    publish(0);
    issue(ring[1], 3);
    // ---- down_proj's words of this wave's rows, this XCD's columns: LDS-DMA behind the last gate / up rows (no register held: at 4 bits
    //      they are 8 rows x 16 bytes per lane on top of the image); one row slice per instruction, lanes = its 32-value slots.
    //      (Measured and not kept: all of them requested by the eight idle waves right behind barrier A, while the image is built -- the
    //      first gate / up rows then arrive behind 112 KB per CU: 779-780 vs 786-788 tok/s on 8B S40.)
    if (lane < nsl) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (i >= rows_w) break;
        const int rl = wave * rows_w + i;
        auto dst = (__attribute__((address_space(3))) void*)(dwl + rl * SLL);
        if constexpr (B == 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(dd, dst, 12, lane * 12, (r * RPWG + rl) * W4d + g * SLB, 0, 2);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(dd, dst, 16, lane * 16, (r * RPWG + rl) * W4d + g * SLB, 0, 2);
      }
    }
    DV_STAMP(3);
    consume(ring[0], 2);
    consume(ring[1], 3);
    publish(1);
  }
  DV_STAMP(4);
  // (scale | zero) of the lane's slot of each of the wave's rows: one dword per row (gbx_dgemv_pack of down_proj, two rows per batch)
  const int Gd = a.I >> a.dn_log2g;
  const int ngx = IX >> a.dn_log2g;                              // groups of this XCD's slice
  unsigned dsz[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = r * RPWG + wave * rows_w + (i < rows_w ? i : 0);
    const int gi = g * ngx + ((min(lane, nsl - 1) * 32) >> a.dn_log2g);
    dsz[i] = __builtin_amdgcn_raw_buffer_load_b32(dsd, static_cast<unsigned>((((row >> 1) * Gd + gi) * 2 + (row & 1)) * 4) | (i < rows_w ? 0u : KILL), 0, 0);
  }
  // ---------------- 3. hand-off: the XCD's activations -> down_proj's fp16 image (dattn_kernel.h, phase 3) ----------------
  unsigned* raw3 = reinterpret_cast<unsigned*>(smem + DV_LDS_P3);
  h2* xp3 = reinterpret_cast<h2*>(smem + DV_LDS_P3 + 4096);
  float2* xs3 = reinterpret_cast<float2*>(reinterpret_cast<unsigned char*>(xp3) + static_cast<size_t>(nsl) * SSTR * 4);
  const int nit3 = (IX / F::UNIT_VALS) * TPU;
  if constexpr (B == 3) {
    // 3 bits: an image item is a whole 32-value unit -- the activations go through LDS as raw values first (IX / 2 granules)
    const int ngr = IX >> 1;
    if (wave * 64 < ngr) raw3[wave * 64 + lane] = sweep(hand + min(wave * 64 + lane, ngr - 1), tag, 3u);
    __syncthreads();
    DV_STAMP(5);
    if ((tid & ~63) < nit3) {
      unsigned xw[PPU];
      da_load_item<B, PPU>(reinterpret_cast<const unsigned short*>(raw3), min(tid, nit3 - 1), xw);
      da_image_item<B, BF16, PPU>(xw, xw, false, 1.f, tid, nit3, xp3, xs3);
    }
  } else {
    // 4 bits: an image item is two value pairs = two granules (unit u, pair t: granules u * UP + t and + UP / 2)
    if ((tid & ~63) < nit3) {
      const int it = min(tid, nit3 - 1);
      const int u = it / TPU, t = it % TPU;
      unsigned xw[PPU];
      sweep2(hand + u * UP + t, hand + u * UP + UP / 2 + t, tag, 3u, xw[0], xw[1]);
      DV_STAMP(5);
      da_image_item<B, BF16, PPU>(xw, xw, false, 1.f, tid, nit3, xp3, xs3);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // this wave's down_proj rows have landed in LDS
  __syncthreads();
  DV_STAMP(6);
  // ---------------- 4. down_proj over this XCD's columns: one lane = one slot of one row ----------------
  // The lane's slot of the activation image is the same for every row of the wave: read ONCE (da_o_slot re-reads 64 bytes per lane and row:
  // 448 KB of LDS traffic per CU, 5.2 us for this phase measured); two rows per reduction.
  {
    dv_gu64* accp = reinterpret_cast<dv_gu64*>(reinterpret_cast<uintptr_t>(a.acc_out));
    const int sl = min(lane, nsl - 1);
    h2 xv3[PAIRS];
    {
      const u32x4* src = reinterpret_cast<const u32x4*>(xp3 + static_cast<size_t>(sl) * SSTR);
#pragma unroll
      for (int q = 0; q < PAIRS / 4; ++q) {
        const u32x4 v = src[q];
        xv3[4 * q + 0] = as_h2(v.x); xv3[4 * q + 1] = as_h2(v.y); xv3[4 * q + 2] = as_h2(v.z); xv3[4 * q + 3] = as_h2(v.w);
      }
    }
    const float2 t3 = xs3[sl];
    const bool livel = lane < nsl;
    float yv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      yv[i] = 0.f;
      if (i >= rows_w) continue;
      const int rl = wave * rows_w + i;
      const u32x4 w = *reinterpret_cast<const u32x4*>(dwl + rl * SLL + sl * 16);      // (3 bits: .x .y .z)
      WT ws;
      if constexpr (B == 3) ws = u32x3{w.x, w.y, w.z}; else ws = w;
      float d[NSUB];
      dot_slot<B>(ws, xv3, d);
      const float sc = act_to_f32<BF16>(static_cast<unsigned short>(dsz[i] & 0xFFFFu)), ze = act_to_f32<BF16>(static_cast<unsigned short>(dsz[i] >> 16));
      const float v = fmaf(ze, t3.x, (sc * t3.y) * d[0]);
      yv[i] = livel ? v : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (2 * q >= rows_w) break;
      const float v = wave_sum2(yv[2 * q], yv[2 * q + 1]);       // row 2 q in lane 31, row 2 q + 1 in lane 63
      if ((lane & 31) == 31) {
        const int row = r * RPWG + wave * rows_w + 2 * q + (lane >> 5);
        if (fabsf(v) < 2147483648.f) {
          const long long fx = __float2ll_rn(v * 4294967296.f);
          __hip_atomic_fetch_add(accp + row, static_cast<unsigned long long>(fx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          __hip_atomic_fetch_add(accp + a.D, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
  }
  if (r == 0 && tid == 0) __hip_atomic_store(gen_p, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  DV_STAMP(7);
}

bool dmlpv_plan(int D, int I, int bits_gu, int group_gu, int bits_dn, int group_dn, bool bf16, int* bpw, int* a4) {
  if (!xcd_layout_ok()) return false;
  if (!bf16 || bits_gu != bits_dn || (bits_gu != 3 && bits_gu != 4)) return false;
  auto okg = [](int g) { return g == 32 || g == 64 || g == 128; };
  if (!okg(group_gu) || !okg(group_dn)) return false;
  if (D != 4096 && D != 3072) return false;                                 // two lane passes of 32-value slots; D / 32 rows per workgroup, D / 512 per wave
  if (I <= 0 || (I % (8 * 128)) != 0) return false;
  const int IX = I / 8;
  if (IX > 2048 || IX < 512) return false;
  if ((IX % DA_MEMBERS) != 0) return false;
  const int b = IX / DA_MEMBERS;                                            // batches per workgroup: 4 a4 + 2 (16 - a4)
  if (b < 32 || b > 64 || (b & 1)) return false;
  const int rows_w = D / (DA_MEMBERS * DA_NW);
  if ((D % (DA_MEMBERS * DA_NW)) != 0 || rows_w > 8) return false;
  const int slb = (IX / 32) * 16;
  if (static_cast<long long>(D / DA_MEMBERS) * slb > DV_LDS_BYTES - DV_LDS_DW) return false;
  if (static_cast<long long>(I) * (D / 32 * bits_gu * 4) >= 0x7FFFFFFFll || static_cast<long long>(D) * (I / 32 * bits_dn * 4) >= 0x7FFFFFFFll) return false;
  *bpw = b;
  *a4 = (b - 32) / 2;
  return true;
}

hipError_t launch_dmlpv(const DmlpArgs& a, int bits, hipStream_t stream) {
  static DeviceOnce once;
  if (once.pending()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(dmlpv_kernel<3, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, DV_LDS_BYTES);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(dmlpv_kernel<4, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, DV_LDS_BYTES);
    if (e != hipSuccess) return e;
    once.done();
  }
  if (bits == 3) hipLaunchKernelGGL((dmlpv_kernel<3, 2, true>), dim3(DA_GROUPS * DA_MEMBERS), dim3(DA_NT), DV_LDS_BYTES, stream, a);
  else hipLaunchKernelGGL((dmlpv_kernel<4, 2, true>), dim3(DA_GROUPS * DA_MEMBERS), dim3(DA_NT), DV_LDS_BYTES, stream, a);
  return hipGetLastError();
}

}  // namespace gbx
