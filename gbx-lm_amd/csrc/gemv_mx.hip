// Batch-1 decode GEMV for 2-bit GBA weights on the block-scaled matrix cores of gfx950 (round 3):
//   y[n] = sum_k x[k] * (s[n, k/128] * q[n, k] + z[n, k/128])          -- the M == 1 case of mx.quantized_matmul at
// gbx_lm/models/quantized_linear_gba.py:195-203, same fusions as gemv_kernel (RMSNorm prologue, residual / SwiGLU epilogue,
// x_acc / zero_buf of the four-launch block).
//
// Why: gemv_kernel is bound by the vector ALUs -- one v_and + one v_dot2c per PAIR of weights, ~270 cycles per packed row and
// wave; with a quarter of that work (timing experiment, wrong results) the token went 1308 -> 1179 us.  Here the multiply-
// accumulate is v_mfma_scale_f32_16x16x128_f8f6f4:
//   * B operand = the packed weights AS THEY ARE: a 2-bit field masked in place inside its nibble is the FP4 (E2M1) number
//     q / 2 (codes 0..3 = 0, 0.5, 1, 1.5: measured, scratch/probe_mx.hip), so one packed dword (16 fields) becomes two operand
//     registers with 3 VALU instructions (w & 0x33333333, (w >> 2) & 0x33333333) -- 0.19 per weight instead of ~1.06.  No
//     re-packing of the checkpoint: a wave loads 16 rows x 128 contiguous bytes per step straight from [N, K/16] words.
//   * A operand = the activations as FOUR FP8 (E4M3) limbs with one E8M0 block exponent per 128-value quantisation group
//     (x * 2^-e = L0 + L1 + L2 + L3, each limb the FP8 rounding of what the previous ones left: 16 significant bits below the
//     group maximum, every bf16 / fp16 value within 2^-8 of it exact), built once per workgroup in LDS.
//   * the 16 rows of the A tile are 4 group slots x 4 limbs and a lane group's k-block only feeds its own slot (the other
//     rows of that k-block are zero), so ONE accumulator tile collects the sums of four quantisation groups for 16 weight rows:
//     after 4 MFMAs (16 rows x 512 values) lane (n, g) holds the four limb sums of (row n, group g), adds them and applies
//     s and z in fp32 -- ~2 flush instructions per MFMA.
// Operand slot map of the instruction for A = fp8, B = fp4 (measured, scratch/probe_mx2.hip): B lane group gb, dwords 0-1
// <-> A lane group 2 (gb & 1), bytes 16 (gb >> 1) + 0..15; dwords 2-3 <-> A lane group 2 (gb & 1) + 1, same bytes.
#include "gemv_impl.h"
#include "kernels.h"

namespace gbx {

typedef int v8i_t __attribute__((ext_vector_type(8)));
typedef float v4f_t __attribute__((ext_vector_type(4)));

constexpr int MX_NW = 8;                       // waves per workgroup
constexpr int MX_SETV = 512;                   // values of a row one 4-MFMA set covers (4 quantisation groups of 128)

// ---- LDS flags (explicit local address space: a volatile access through a generic pointer becomes a flat load that drags an
// s_waitcnt vmcnt(0) into every spin -- scratch/negative/gemv_ov.h) ----
typedef __attribute__((address_space(3))) volatile unsigned mx_flag_t;
__device__ __forceinline__ void mx_wait_all(mx_flag_t* f, int n) {      // until the first n words are all non-zero
  for (;;) {
    unsigned all = 1;
    for (int i = 0; i < n; ++i) all &= f[i];
    if (__builtin_amdgcn_readfirstlane(all) != 0) break;
    __builtin_amdgcn_s_sleep(1);
  }
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void mx_raise(mx_flag_t* f) {                 // by one lane, after this wave's LDS writes
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  *f = 1u;
}

constexpr int MX_NL = 2;                       // loader waves (the last two of the workgroup)
constexpr int MX_NC = MX_NW - MX_NL;           // consumer waves: activation prologue, matrix-core work, epilogue
constexpr int MX_MAXSLOT = 8;                  // (tile, k part) slots per workgroup, each with its own LDS region

// LDS: image [G][t = 4][limb = 4][32 bytes], per group {sum(x) fp32}, the E8M0 bytes, norm partials, flags, slot partial rows, a
// zero block ...
__host__ __device__ inline size_t mx_fixed_bytes(int K) {
  const int G = K / 128;
  return ((static_cast<size_t>(G) * 512 + static_cast<size_t>(G) * 4 + ((G + 63) & ~63) + MX_NW * 4 + 64 * 4 + MX_MAXSLOT * 16 * 4 + 64) + 15) & ~static_cast<size_t>(15);
}
// ... then per slot the 16 packed-row segments (spp * 128 bytes each) of its tile, filled by LDS-DMA.  ONE instruction moves
// 1 KiB = R = 8 / spp rows x their whole segment, the rows' 16-byte pieces interleaved (lane L = piece L / R of row L % R): whole
// cache lines on the memory side; on the LDS side a block is 1024 + 16 R bytes, which puts the 16 rows that a quarter-wave
// reads (one row per lane, same piece) into 16 distinct bank quads.
__host__ __device__ inline int mx_block_bytes(int spp) { return 1024 + 16 * (8 / spp); }
__host__ __device__ inline size_t mx_slot_bytes(int spp) { return static_cast<size_t>(2 * spp) * mx_block_bytes(spp); }   // 16 / R blocks
__host__ __device__ inline size_t mx_lds_bytes(int K, int spp, int nslots) { return mx_fixed_bytes(K) + nslots * mx_slot_bytes(spp); }

// SPP = sets (of 4 MFMAs = 512 values per row) per slot: 8, 4, 2 or 1.  Workgroup = MX_NC consumer waves + MX_NL loader waves.
// Loader waves request every slot of the workgroup (a wave issues in order: one that is stuck in the CU's vector-memory queue
// cannot convert x, so the waves that do the prologue never issue a row request -- first versions: prologue done at 6-8 us) and
// raise a slot's flag when its rows have landed; consumer wave c owns slots c, c + MX_NC.
template <int SPP, bool BF16, bool XACC>
__global__ __launch_bounds__(MX_NW * 64) void gemv_mx_kernel(const GemvArgs a, const int tpw, const int kp) {
  constexpr int NTC = MX_NC * 64;                                         // consumer threads
  constexpr int R = 8 / SPP, NBLK = 16 / R, BST = 1024 + 16 * R;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int K = a.K, N = a.N;
  const int G = K >> 7;
  unsigned char* img = smem;                                              // [G][4][4][32]
  float* xsum = reinterpret_cast<float*>(smem + static_cast<size_t>(G) * 512);
  unsigned char* scl = reinterpret_cast<unsigned char*>(xsum + G);        // [G] E8M0 bytes
  float* red = reinterpret_cast<float*>(smem + static_cast<size_t>(G) * 512 + G * 4 + ((G + 63) & ~63));
  mx_flag_t* flags = (mx_flag_t*)(red + MX_NW);                           // [0,8) statistics, [8,16) image, [16,24) slot landed, [24,32) slot partial ready
  float* part = red + MX_NW + 64;                                         // [MX_MAXSLOT][16]
  unsigned* zero_blk = reinterpret_cast<unsigned*>(part + MX_MAXSLOT * 16);   // 32 zero bytes (+ pad)
  unsigned char* const slots_lds = smem + mx_fixed_bytes(K);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool swiglu = a.swiglu != 0;
  const int n0 = a.seg[0].N, n01 = n0 + a.seg[1].N;
#ifdef GBX_STAMPS
#define MX_STAMP(i) do { if (lane == 0 && a.stamps) a.stamps[(static_cast<size_t>(blockIdx.x) * MX_NW + wave) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define MX_STAMP(i) do {} while (0)
#endif
  MX_STAMP(0);
  if (tid < 64) flags[tid] = 0u;
  if (tid >= 64 && tid < 80) zero_blk[tid - 64] = 0u;
  __syncthreads();                                                        // the only hardware barrier: nothing is in flight yet

  const int ntiles = swiglu ? (N + 7) / 8 : (N + 15) / 16;
  const int nslot = tpw * kp;                                             // slots of this workgroup: slot = (tile slot) * kp + k part
  const int W = K / 16;                                                   // words per packed row
  const int nsets = K / MX_SETV;
  const uint32_t* const q0p = a.seg[0].qweight; const uint32_t* const q1p = a.seg[1].qweight; const uint32_t* const q2p = a.seg[2].qweight;
  unsigned short* const y0p = a.seg[0].y; unsigned short* const y1p = a.seg[1].y; unsigned short* const y2p = a.seg[2].y;
  // tile of a slot: wave-uniform scalars (the tile lies inside ONE segment: every segment has a multiple of 16 rows)
  auto slot_tile = [&](int sl, int& tile, int& kpart, int& tsg, int& trow0) __attribute__((always_inline)) {
    tile = __builtin_amdgcn_readfirstlane(blockIdx.x * tpw + sl / kp);
    kpart = __builtin_amdgcn_readfirstlane(sl % kp);
    const int tc = min(tile, ntiles - 1);
    const int tr0 = swiglu ? tc * 8 : tc * 16;
    tsg = swiglu ? 0 : (tr0 >= n0) + (tr0 >= n01);
    trow0 = tr0 - (tsg == 0 ? 0 : (tsg == 1 ? n0 : n01));
  };

  if (wave >= MX_NC) {
    // ======================= loader wave: slots l, l + MX_NL, ... -- request, then publish in landing order ===============
    const int l0 = wave - MX_NC;
    auto issue_slot = [&](int sl) __attribute__((always_inline)) {
      int tile, kpart, tsg, trow0;
      slot_tile(sl, tile, kpart, tsg, trow0);
      const uint32_t* const tq = tsg == 0 ? q0p : (tsg == 1 ? q1p : q2p);
      unsigned char* dst = slots_lds + static_cast<size_t>(sl) * (NBLK * BST);
#pragma unroll
      for (int b = 0; b < NBLK; ++b) {                                    // rows b R .. b R + R - 1 (same matrix: R divides 8)
        const int r1 = b * R;
        const uint32_t* base = swiglu ? (r1 < 8 ? q0p : q1p) : tq;
        const int row = swiglu ? trow0 + (r1 & 7) : trow0 + r1;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(base), 0, 0x7FFFFFFF, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + b * BST), 16,
                                                 (lane % R) * (W * 4) + (lane / R) * 16, row * (W * 4) + kpart * (SPP * 128), 0, 2);
      }
    };
    int prev = -1;
    for (int sl = l0; sl < nslot; sl += MX_NL) {
      issue_slot(sl);
      if (prev >= 0) {
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NBLK) : "memory");    // everything but the slot just requested has landed
        if (lane == 0) flags[16 + prev] = 1u;
      }
      prev = sl;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (prev >= 0 && lane == 0) flags[16 + prev] = 1u;
    MX_STAMP(6);
    return;
  }

  // ======================= consumer waves ===============================================================================
  // this wave's first slot: scale | zero words for the flush are requested right away
  const int nrow = lane & 15, gb = lane >> 4;
  MX_STAMP(1);
  if (a.zero_buf) {
    for (int i = blockIdx.x * NTC + tid; i < a.zero_n; i += static_cast<int>(gridDim.x) * NTC) a.zero_buf[i] = 0;
  }
  // ---------------- 1. activations -> four FP8 limbs + one E8M0 exponent per 128-value group, in LDS ----------------
  const unsigned short* __restrict__ xg = static_cast<const unsigned short*>(a.x);
  const bool norm = a.norm_w != nullptr;
  const unsigned short* __restrict__ wg = norm ? a.norm_w : xg;
  // one thread = HALF a packed dword's worth of values (8)
  const int nhu = K / 8;
  constexpr int UPT_MAX = 6;                                              // K <= 16384 on 384 threads
  const int upt = (nhu + NTC - 1) / NTC;
  unsigned xr[UPT_MAX][4], wr[UPT_MAX][4];
#pragma unroll
  for (int j = 0; j < UPT_MAX; ++j) {
    if (j < upt) {
      const int hu = min(tid + NTC * j, nhu - 1);
      const u32x4 v0 = *reinterpret_cast<const u32x4*>(xg + static_cast<size_t>(hu) * 8);
      xr[j][0] = v0.x; xr[j][1] = v0.y; xr[j][2] = v0.z; xr[j][3] = v0.w;
      if (norm) {
        const u32x4 t0 = *reinterpret_cast<const u32x4*>(wg + static_cast<size_t>(hu) * 8);
        wr[j][0] = t0.x; wr[j][1] = t0.y; wr[j][2] = t0.z; wr[j][3] = t0.w;
      }
      if constexpr (XACC) {      // x <- round(x + round(acc * 2^-32)) (gemv_impl.h, section 3.0); workgroup 0 writes the new stream
        const u32x4* ap = reinterpret_cast<const u32x4*>(a.x_acc + static_cast<size_t>(hu) * 8);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const u32x4 v = ap[q];
          const long long a0 = static_cast<long long>((static_cast<unsigned long long>(v.y) << 32) | v.x);
          const long long a1 = static_cast<long long>((static_cast<unsigned long long>(v.w) << 32) | v.z);
          const float r0 = round_bf16_or_f16(static_cast<float>(a0) * (1.f / 4294967296.f), BF16);
          const float r1 = round_bf16_or_f16(static_cast<float>(a1) * (1.f / 4294967296.f), BF16);
          const unsigned lo = f32_to_act<BF16>(act_lo<BF16>(xr[j][q]) + r0), hi = f32_to_act<BF16>(act_hi<BF16>(xr[j][q]) + r1);
          xr[j][q] = lo | (hi << 16);
        }
        if (blockIdx.x == 0 && a.x_out && tid + NTC * j < nhu)
          *reinterpret_cast<u32x4*>(a.x_out + static_cast<size_t>(hu) * 8) = u32x4{xr[j][0], xr[j][1], xr[j][2], xr[j][3]};
      }
    }
  }
  float rstd = 1.f;
  if (norm) {
    float ssq = 0.f;
#pragma unroll
    for (int j = 0; j < UPT_MAX; ++j) {
      if (j < upt) {
        const bool live = (tid + NTC * j) < nhu;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float lo = act_lo<BF16>(xr[j][q]), hi = act_hi<BF16>(xr[j][q]);
          ssq += live ? fmaf(hi, hi, lo * lo) : 0.f;   // explicit: -ffp-contract=on may pick either product for the fma, per instantiation
        }
      }
    }
    ssq = wave_sum(ssq);
    if (lane == 0) { red[wave] = ssq; mx_raise(flags + wave); }
    mx_wait_all(flags, MX_NC);
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < MX_NC; ++i) tot += red[i];
    rstd = rsqrtf(tot / static_cast<float>(K) + a.eps);
  }
  MX_STAMP(9);
#pragma unroll
  for (int j = 0; j < UPT_MAX; ++j) {
    if (j < upt && (tid & ~63) + NTC * j < nhu) {                          // wave-uniform: idle waves skip the arithmetic
      const int hu = tid + NTC * j;
      float r[8];
      unsigned amax = 0;
      float gsum = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float lo = act_lo<BF16>(xr[j][q]), hi = act_hi<BF16>(xr[j][q]);
        if (norm) {
          lo = round_bf16_or_f16(lo * rstd * act_lo<BF16>(wr[j][q]), BF16);
          hi = round_bf16_or_f16(hi * rstd * act_hi<BF16>(wr[j][q]), BF16);
        }
        r[2 * q] = lo;
        r[2 * q + 1] = hi;
        gsum += lo + hi;
        amax = max(amax, max(__float_as_uint(lo) & 0x7FFFFFFFu, __float_as_uint(hi) & 0x7FFFFFFFu));
      }
      // the 16 items (lanes) of a 128-value group: sum and maximum (items past the end hold copies of the last item and only
      // ever meet each other: K is a multiple of 128 and 384 = 3 * 128 keeps groups inside one wave)
      gsum += dpp_f32<0xB1>(gsum); amax = max(amax, dpp_u32<0xB1>(amax));
      gsum += dpp_f32<0x4E>(gsum); amax = max(amax, dpp_u32<0x4E>(amax));
      gsum += dpp_f32<0x141>(gsum); amax = max(amax, dpp_u32<0x141>(amax));
      gsum += dpp_f32<0x140>(gsum); amax = max(amax, dpp_u32<0x140>(amax));
      // x' = x * 2^(134 - ea): the group maximum lands in [128, 256) (E4M3 holds up to 448); the instruction multiplies by 2^(scale - 127)
      int ea = static_cast<int>(amax >> 23);
      ea = min(max(ea, 20), 240);
      const float up = exp2i(134 - ea);
      if (hu < nhu) {
        const int u = hu >> 1, hh = hu & 1;
        const int g = u >> 3, du = u & 7;
        // the 16 bytes of one limb of a packed dword's values: fields in the order (0, 2, ..., 14, 1, 3, ..., 15) = the nibble
        // order of the two operand registers the dword turns into; this item owns fields 8 hh .. 8 hh + 7: one dword of the even
        // fields (byte 4 hh) and one of the odd fields (byte 8 + 4 hh)
        unsigned char* dst = img + (static_cast<size_t>(g) * 4 + (du & 3)) * 128 + (du >> 2) * 16 + hh * 4;
#pragma unroll
        for (int q = 0; q < 8; ++q) r[q] *= up;
#pragma unroll
        for (int l = 0; l < 4; ++l) {
          int pe = __builtin_amdgcn_cvt_pk_fp8_f32(r[0], r[2], 0, false);
          pe = __builtin_amdgcn_cvt_pk_fp8_f32(r[4], r[6], pe, true);
          int po = __builtin_amdgcn_cvt_pk_fp8_f32(r[1], r[3], 0, false);
          po = __builtin_amdgcn_cvt_pk_fp8_f32(r[5], r[7], po, true);
          *reinterpret_cast<unsigned*>(dst + l * 32) = static_cast<unsigned>(pe);
          *reinterpret_cast<unsigned*>(dst + l * 32 + 8) = static_cast<unsigned>(po);
          if (l < 3) {
            r[0] -= __builtin_amdgcn_cvt_f32_fp8(pe, 0); r[2] -= __builtin_amdgcn_cvt_f32_fp8(pe, 1);
            r[4] -= __builtin_amdgcn_cvt_f32_fp8(pe, 2); r[6] -= __builtin_amdgcn_cvt_f32_fp8(pe, 3);
            r[1] -= __builtin_amdgcn_cvt_f32_fp8(po, 0); r[3] -= __builtin_amdgcn_cvt_f32_fp8(po, 1);
            r[5] -= __builtin_amdgcn_cvt_f32_fp8(po, 2); r[7] -= __builtin_amdgcn_cvt_f32_fp8(po, 3);
          }
        }
        if ((hu & 15) == 0) { xsum[g] = gsum; scl[g] = static_cast<unsigned char>(ea - 7); }
      }
    }
  }
  MX_STAMP(10);
  if (lane == 0) mx_raise(flags + 8 + wave);
  mx_wait_all(flags + 8, MX_NC);
  MX_STAMP(2);

  // ---------------- 2. this wave's slots: SPP sets of 4 MFMAs each ----------------
  // A operand of lane (m = lane & 15 -> slot m >> 2, limb m & 3; lane group ga): limb bytes of group 4S + ga when the slot
  // is the lane group's own, zeros otherwise (a 32-byte zero block read with stride 0)
  const int m = lane & 15, ga = lane >> 4;
  const bool mine = (m >> 2) == ga;
  const unsigned a_base = mine ? static_cast<unsigned>(ga * 512 + (m & 3) * 32) : static_cast<unsigned>(reinterpret_cast<unsigned char*>(zero_blk) - smem);
  const unsigned a_step_t = mine ? 128u : 0u, a_step_s = mine ? 2048u : 0u;
  const int scale_one = 127;
  // this lane's 16-byte pieces of row nrow inside a slot region: block nrow / R, row nrow % R of the interleave; piece
  // 4 (gb & 1) + (gb >> 1) of each 128-byte set chunk, and the one 32 bytes on
  const int pc = 4 * (gb & 1) + (gb >> 1);
  const unsigned w_off = (nrow / R) * BST + (nrow % R) * 16 + pc * (16 * R);
  // phase A: the matrix-core work of this wave's (at most two) slots; phase B: k-part sums and epilogues.  (Two phases: the
  // wave that owns part 0 of a tile may also own one of its later parts -- it must not wait for itself.)
  float ys[2] = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int sl = wave + i * MX_NC;
    if (sl < nslot) {
      int tile, kpart, tsg, trow0;
      slot_tile(sl, tile, kpart, tsg, trow0);
      const int set0 = kpart * SPP;
      // row of the packed companion this lane flushes: [N][G][2] (gate, up) pairs for SwiGLU, else rows of a pair interleaved
      const int tc = min(tile, ntiles - 1);
      const int szrow = swiglu ? (tc * 8 + (nrow & 7)) * 2 + (nrow >> 3) : tc * 16 + nrow;
      unsigned szr[SPP];
#pragma unroll
      for (int s = 0; s < SPP; ++s) {
        const int S = min(set0 + s, nsets - 1);
        szr[s] = a.sz[(static_cast<size_t>(szrow >> 1) * G + 4 * S + gb) * 2 + (szrow & 1)];
      }
      mx_wait_all(flags + 16 + sl, 1);                                    // the slot's rows have landed (loader wave)
      MX_STAMP(3);
      const unsigned char* const wrow = slots_lds + static_cast<size_t>(sl) * (NBLK * BST) + w_off;
      float yacc = 0.f;
#pragma unroll
      for (int s = 0; s < SPP; ++s) {
        const int S = min(set0 + s, nsets - 1);
        v4f_t c = {0.f, 0.f, 0.f, 0.f};
        // The E8M0 scale of A's hardware k-block kb comes from lane (row m, lane group kb) -- NOT from the lane that holds the
        // bytes: a lane group's 32 bytes straddle the k-blocks (ga >> 1) and (ga >> 1) + 2 (measured: one-hot x,
        // scratch/dbg_mx3.py).  Row m only has data in its own slot's lane group, so all four lanes of row m carry the exponent
        // of slot m >> 2.
        const int sa = scl[4 * S + (m >> 2)];
        const u32x4 wAv = *reinterpret_cast<const u32x4*>(wrow + s * (128 * R));
        const u32x4 wBv = *reinterpret_cast<const u32x4*>(wrow + s * (128 * R) + 32 * R);
        const unsigned wa[4] = {wAv.x, wAv.y, wAv.z, wAv.w};
        const unsigned wb[4] = {wBv.x, wBv.y, wBv.z, wBv.w};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const u32x4* ap = reinterpret_cast<const u32x4*>(smem + a_base + S * a_step_s + t * a_step_t);
          const u32x4 a0 = ap[0], a1 = ap[1];
          v8i_t av, bv;
          av[0] = a0.x; av[1] = a0.y; av[2] = a0.z; av[3] = a0.w; av[4] = a1.x; av[5] = a1.y; av[6] = a1.z; av[7] = a1.w;
          bv[0] = wa[t] & 0x33333333u; bv[1] = (wa[t] >> 2) & 0x33333333u; bv[2] = wb[t] & 0x33333333u; bv[3] = (wb[t] >> 2) & 0x33333333u;
          bv[4] = 0; bv[5] = 0; bv[6] = 0; bv[7] = 0;
          c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 0 /*A: fp8 e4m3*/, 4 /*B: fp4*/, 0, sa, 0, scale_one);
        }
        // lane (n, gc = ga): limb sums of (row n, group 4S + gc); the FP4 codes are q / 2
        const float tsum = (c[0] + c[1]) + (c[2] + c[3]);
        const float sc = sz_scale<BF16>(szr[s]), ze = sz_zero<BF16>(szr[s]);
        yacc += fmaf(ze, xsum[4 * S + ga], (sc + sc) * tsum);
      }
      MX_STAMP(4);
      // the four lane groups hold the four group slots of the same 16 rows
      yacc += __shfl_xor(yacc, 16);
      yacc += __shfl_xor(yacc, 32);
      if (kp > 1 && kpart != 0) {         // k parts 1.. of a tile park their rows in LDS
        if (lane < 16) part[sl * 16 + lane] = yacc;
        if (lane == 0) mx_raise(flags + 24 + sl);
      }
      ys[i] = yacc;
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int sl = wave + i * MX_NC;
    if (sl < nslot) {
      int tile, kpart, tsg, trow0;
      slot_tile(sl, tile, kpart, tsg, trow0);
      if (kpart != 0) continue;
      float yacc = ys[i];
      if (kp > 1) {                       // part 0 adds the others in part order
        mx_wait_all(flags + 24 + sl + 1, kp - 1);
        if (lane < 16) for (int p = 1; p < kp; ++p) yacc += part[(sl + p) * 16 + lane];
      }
      // ---------------- 3. epilogue: lanes 0..15 hold the 16 row totals ----------------
      if (tile < ntiles) {
        if (swiglu) {
          const float up = __shfl(yacc, (lane & 7) + 8);
          const int r = tile * 8 + lane;
          if (lane < 8 && r < N) {
            const float g = round_bf16_or_f16(yacc, BF16), u = round_bf16_or_f16(up, BF16);
            const float sl2 = round_bf16_or_f16(g / (1.f + __expf(-g)), BF16);
            y0p[r] = f32_to_act<BF16>(sl2 * u);
          }
        } else if (lane < 16) {
          unsigned short* const ty = tsg == 0 ? y0p : (tsg == 1 ? y1p : y2p);
          const int row = trow0 + lane;
          float v = yacc;
          if (a.residual) v = act_to_f32<BF16>(a.residual[row]) + round_bf16_or_f16(v, BF16);
          ty[row] = f32_to_act<BF16>(v);
        }
      }
    }
  }
  MX_STAMP(6);
}

// ---- launch: slots (tile, k part) per workgroup so that every CU gets (about) one workgroup --------------------------------
bool gemv_mx_eligible(const GemvArgs& a, int M) {
  // Opt-in (GBX_GEMV_MX=1): numerically on par with gemv_kernel (tests/test_mx_gpu.py), but not faster yet -- with the matrix
  // cores doing the arithmetic a gate|up launch is bound by its cold 33 MB stream (9.2 us with the arithmetic removed, 11.2
  // with it, 11.2 for gemv_kernel): DESIGN.md 3.1c.
  static const bool on = [] { const char* e = getenv("GBX_GEMV_MX"); return e && e[0] == '1'; }();
  if (!on || M != 1 || !a.sz || a.expert_idx || a.bias || a.log2g != 7) return false;
  if (a.K % MX_SETV != 0 || a.K > 16384) return false;
  if (a.norm_w && a.K > 8192) return false;
  for (int i = 0; i < a.nseg; ++i) if (a.seg[i].N % 16 != 0) return false;
  if (a.swiglu && (a.seg[0].N % 8 != 0)) return false;
  if (a.x_acc && a.K > 8192) return false;
  return true;
}

// tiles per workgroup, k parts per tile and sets per slot: at most MX_MAXSLOT slots of at most 8 sets, the k parts dividing the
// sets into 1, 2, 4 or 8 sets each; about one workgroup per CU
static bool mx_plan(const GemvArgs& a, int& tpw, int& kp, int& spp, int& blocks) {
  const int ntiles = a.swiglu ? (a.N + 7) / 8 : (a.N + 15) / 16;
  const int nsets = a.K / MX_SETV;
  tpw = ntiles / GEMV_CUS;
  tpw = tpw < 1 ? 1 : (tpw > MX_MAXSLOT ? MX_MAXSLOT : tpw);
  kp = 0;
  for (int c = MX_MAXSLOT / tpw; c >= 1; --c) {       // the most parts that fit: more waves per tile, smaller slots
    if (nsets % c) continue;
    const int sp = nsets / c;
    if (sp == 1 || sp == 2 || sp == 4 || sp == 8) { kp = c; break; }
  }
  if (!kp) return false;
  spp = nsets / kp;
  blocks = (ntiles + tpw - 1) / tpw;
  return mx_lds_bytes(a.K, spp, tpw * kp) <= 160 * 1024;
}

template <int SPP>
static hipError_t launch_mx_spp(const GemvArgs& a, bool bf16, int blocks, int tpw, int kp, hipStream_t s) {
  const size_t shmem = mx_lds_bytes(a.K, SPP, tpw * kp);
  dim3 grid(blocks), block(MX_NW * 64);
  static DeviceOnce attr_done[2][2];
  auto set_attr = [&](auto kern, DeviceOnce& once) {
    if (once.pending()) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); once.done(); }
  };
  if (a.x_acc) {
    if (bf16) { set_attr(gemv_mx_kernel<SPP, true, true>, attr_done[1][1]); hipLaunchKernelGGL((gemv_mx_kernel<SPP, true, true>), grid, block, shmem, s, a, tpw, kp); }
    else { set_attr(gemv_mx_kernel<SPP, false, true>, attr_done[0][1]); hipLaunchKernelGGL((gemv_mx_kernel<SPP, false, true>), grid, block, shmem, s, a, tpw, kp); }
  } else {
    if (bf16) { set_attr(gemv_mx_kernel<SPP, true, false>, attr_done[1][0]); hipLaunchKernelGGL((gemv_mx_kernel<SPP, true, false>), grid, block, shmem, s, a, tpw, kp); }
    else { set_attr(gemv_mx_kernel<SPP, false, false>, attr_done[0][0]); hipLaunchKernelGGL((gemv_mx_kernel<SPP, false, false>), grid, block, shmem, s, a, tpw, kp); }
  }
  return hipGetLastError();
}

hipError_t launch_gemv_mx(const GemvArgs& a, bool bf16, hipStream_t s) {
  int tpw, kp, spp, blocks;
  if (!mx_plan(a, tpw, kp, spp, blocks)) return hipErrorInvalidValue;
  switch (spp) {
    case 1: return launch_mx_spp<1>(a, bf16, blocks, tpw, kp, s);
    case 2: return launch_mx_spp<2>(a, bf16, blocks, tpw, kp, s);
    case 4: return launch_mx_spp<4>(a, bf16, blocks, tpw, kp, s);
    default: return launch_mx_spp<8>(a, bf16, blocks, tpw, kp, s);
  }
}

bool gemv_mx_shape_ok(const GemvArgs& a) {
  int tpw, kp, spp, blocks;
  return mx_plan(a, tpw, kp, spp, blocks);
}

}  // namespace gbx
