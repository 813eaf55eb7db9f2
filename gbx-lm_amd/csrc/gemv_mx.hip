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

// workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt, i.e. would wait for every row segment the
// LDS-DMA requests have in flight
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

typedef int v8i_t __attribute__((ext_vector_type(8)));
typedef float v4f_t __attribute__((ext_vector_type(4)));

constexpr int MX_NW = 8;                       // waves per workgroup
constexpr int MX_SETV = 512;                   // values of a row one 4-MFMA set covers (4 quantisation groups of 128)

// LDS: image [G][t = 4][limb = 4][32 bytes], then per group {sum(x) fp32}, the E8M0 bytes, norm partials, k-part partial rows,
// a zero block ...
__host__ __device__ inline size_t mx_fixed_bytes(int K) {
  const int G = K / 128;
  return ((static_cast<size_t>(G) * 512 + static_cast<size_t>(G) * 4 + ((G + 63) & ~63) + MX_NW * 4 + MX_NW * 16 * 4 + 64) + 15) & ~static_cast<size_t>(15);
}
// ... then per wave the 16 packed-row segments of its tile, filled by LDS-DMA:
//   spp == 8 (a whole 1 KiB row per wave): two K halves x 8 blocks of 1024 + 32 bytes; a block is ONE 64-lane DMA instruction
//     that carries 512 bytes of two rows (r, r + 4), their 16-byte pieces interleaved -- whole cache lines on the memory side,
//     bank-conflict-free operand reads on the LDS side, and the first half's MFMAs run while the second half is in flight;
//   else: 16 rows of spp * 128 bytes + 16 bytes of padding, one (partly masked) instruction per row.
constexpr int MX_BST = 1024 + 32;
__host__ __device__ inline size_t mx_wave_bytes(int spp) { return spp == 8 ? 16 * MX_BST : 16 * (spp * 128 + 16); }
__host__ __device__ inline size_t mx_lds_bytes(int K, int spp) { return mx_fixed_bytes(K) + MX_NW * mx_wave_bytes(spp); }

// SPP = sets (of 4 MFMAs = 512 values per row) per wave
template <int SPP, bool BF16, bool XACC>
__global__ __launch_bounds__(MX_NW * 64) void gemv_mx_kernel(const GemvArgs a, const int tpw, const int kp) {
  constexpr int NT = MX_NW * 64;
  constexpr bool HALVES = SPP == 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int K = a.K, N = a.N;
  const int G = K >> 7;
  unsigned char* img = smem;                                              // [G][4][4][32]
  float* xsum = reinterpret_cast<float*>(smem + static_cast<size_t>(G) * 512);
  unsigned char* scl = reinterpret_cast<unsigned char*>(xsum + G);        // [G] E8M0 bytes
  float* red = reinterpret_cast<float*>(smem + static_cast<size_t>(G) * 512 + G * 4 + ((G + 63) & ~63));
  float* part = red + MX_NW;                                              // [MX_NW][16]
  unsigned* zero_blk = reinterpret_cast<unsigned*>(part + MX_NW * 16);    // 32 zero bytes (+ pad)

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

  // ---------------- 0. this wave's tile and k part (wave-uniform scalars) ----------------
  // (integer division by a run-time value goes through the vector ALU: pin the results back into SGPRs, or every descriptor
  // built from them turns into a waterfall loop)
  const int tile = __builtin_amdgcn_readfirstlane(blockIdx.x * tpw + wave / kp);   // 16 weight rows (swiglu: 8 gate + 8 up)
  const int kpart = __builtin_amdgcn_readfirstlane(wave % kp);
  const int ntiles = swiglu ? (N + 7) / 8 : (N + 15) / 16;
  const bool wlive = wave < tpw * kp && tile < ntiles;
  const int nrow = lane & 15, gb = lane >> 4;
  // The tile lies inside ONE segment (every segment has a multiple of 16 rows): segment, first local row and the pointers are
  // wave-uniform scalars.  (Per-lane selects among the kernel-argument pointers make hipcc spill the argument struct to
  // scratch and fetch every descriptor from there behind a full vmcnt drain.)
  const int tclamp = min(tile, ntiles - 1);
  const int tr0 = swiglu ? tclamp * 8 : tclamp * 16;                      // first global row of the tile
  const int tsg = swiglu ? 0 : (tr0 >= n0) + (tr0 >= n01);
  const int trow0 = tr0 - (tsg == 0 ? 0 : (tsg == 1 ? n0 : n01));         // first row inside its matrix
  const uint32_t* const q0p = a.seg[0].qweight; const uint32_t* const q1p = a.seg[1].qweight; const uint32_t* const q2p = a.seg[2].qweight;
  unsigned short* const y0p = a.seg[0].y; unsigned short* const y1p = a.seg[1].y; unsigned short* const y2p = a.seg[2].y;
  const uint32_t* const tq = tsg == 0 ? q0p : (tsg == 1 ? q1p : q2p);
  unsigned short* const ty = tsg == 0 ? y0p : (tsg == 1 ? y1p : y2p);
  // row of the packed companion this lane flushes: [N][G][2] (gate, up) pairs for SwiGLU, else rows of a pair interleaved
  const int szrow = swiglu ? (tr0 + (nrow & 7)) * 2 + (nrow >> 3) : tr0 + nrow;
  const int W = K / 16;                                                   // words per packed row
  const int set0 = kpart * SPP;
  const int nsets = K / MX_SETV;

  // Packed rows reach the CU by LDS-DMA in whole cache lines (loading the operand layout straight into registers -- lane =
  // (row, k-block), 16 rows per quarter-wave -- made every quarter-wave touch 16 lines: 14 GB/s per CU, gate|up 11.7 us).
  constexpr int RSTRIDE = SPP * 128 + 16;
  unsigned char* const wreg = smem + mx_fixed_bytes(K) + static_cast<size_t>(wave) * mx_wave_bytes(SPP);
  unsigned szr[SPP];
  auto dma = [&](const uint32_t* base, int soff, int voff, unsigned char* dst, bool on) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(base), 0, 0x7FFFFFFF, 0x00020000);
    if (on) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, voff, soff, 0, 2);
  };
  auto issue_rows = [&](int h) __attribute__((always_inline)) {         // HALVES: K half h of all 16 rows; else h == 0: everything
    if constexpr (HALVES) {
#pragma unroll
      for (int b = 0; b < 8; ++b) {                                       // rows (r1, r1 + 4) of the same matrix
        const int r1 = b < 4 ? b : b + 4;
        const uint32_t* base = swiglu ? (b < 4 ? q0p : q1p) : tq;
        const int row = swiglu ? trow0 + (r1 & 7) : trow0 + r1;
        dma(base, row * (W * 4) + set0 * 128 + h * 512, (lane & 1) * (W * 16) + (lane >> 1) * 16, wreg + (h * 8 + b) * MX_BST, true);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const uint32_t* base = swiglu ? (i < 8 ? q0p : q1p) : tq;
        const int row = swiglu ? trow0 + (i & 7) : trow0 + i;
        dma(base, row * (W * 4) + set0 * 128, lane * 16, wreg + i * RSTRIDE, lane < SPP * 8);
      }
    }
  };
  auto issue_sz = [&]() __attribute__((always_inline)) {                // scale | zero of (row, group 4S + gc) for the flush: gc = gb
#pragma unroll
    for (int s = 0; s < SPP; ++s) {
      const int S = min(set0 + s, nsets - 1);
      szr[s] = a.sz[(static_cast<size_t>(szrow >> 1) * G + 4 * S + gb) * 2 + (szrow & 1)];
    }
  };
  MX_STAMP(1);

  // ---------------- 1. activations -> four FP8 limbs + one E8M0 exponent per 128-value group, in LDS ----------------
  if (tid < 16) zero_blk[tid] = 0u;
  if (a.zero_buf) {
    for (int i = blockIdx.x * NT + tid; i < a.zero_n; i += static_cast<int>(gridDim.x) * NT) a.zero_buf[i] = 0;
  }
  const unsigned short* __restrict__ xg = static_cast<const unsigned short*>(a.x);
  const bool norm = a.norm_w != nullptr;
  const unsigned short* __restrict__ wg = norm ? a.norm_w : xg;
  // one thread = HALF a packed dword's worth of values (8): at K = 4096 every one of the 512 threads has exactly one item
  const int nhu = K / 8;
  constexpr int UPT_MAX = 4;                                              // K <= 16384
  const int upt = (nhu + NT - 1) / NT;
  unsigned xr[UPT_MAX][4], wr[UPT_MAX][4];
#pragma unroll
  for (int j = 0; j < UPT_MAX; ++j) {
    if (j < upt) {
      const int hu = min(tid + NT * j, nhu - 1);
      const u32x4 v0 = *reinterpret_cast<const u32x4*>(xg + static_cast<size_t>(hu) * 8);
      const u32x4 t0 = *reinterpret_cast<const u32x4*>(wg + static_cast<size_t>(hu) * 8);
      xr[j][0] = v0.x; xr[j][1] = v0.y; xr[j][2] = v0.z; xr[j][3] = v0.w;
      wr[j][0] = t0.x; wr[j][1] = t0.y; wr[j][2] = t0.z; wr[j][3] = t0.w;
    }
  }
  long long xacc[XACC ? UPT_MAX : 1][XACC ? 8 : 1];
  if constexpr (XACC) {
#pragma unroll
    for (int j = 0; j < UPT_MAX; ++j) {
      if (j < upt) {
        const int hu = min(tid + NT * j, nhu - 1);
        const u32x4* ap = reinterpret_cast<const u32x4*>(a.x_acc + static_cast<size_t>(hu) * 8);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const u32x4 v = ap[q];
          xacc[j][2 * q] = static_cast<long long>((static_cast<unsigned long long>(v.y) << 32) | v.x);
          xacc[j][2 * q + 1] = static_cast<long long>((static_cast<unsigned long long>(v.w) << 32) | v.z);
        }
      }
    }
  }
  // Request order (a wave issues in order, the CU's vector-memory queue is shared): x first, then every row of the tile.
  issue_rows(0);
  issue_sz();
  if constexpr (HALVES) issue_rows(1);
  MX_STAMP(8);
  // x is needed now: its loads are older than every row request, so this wait leaves all of those in flight
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"((HALVES ? 16 : 16) + SPP) : "memory");
  if constexpr (XACC) {      // x <- round(x + round(acc * 2^-32)) (gemv_impl.h, section 3.0); workgroup 0 writes the new stream
#pragma unroll
    for (int j = 0; j < UPT_MAX; ++j) {
      if (j < upt) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float r0 = round_bf16_or_f16(static_cast<float>(xacc[j][2 * q]) * (1.f / 4294967296.f), BF16);
          const float r1 = round_bf16_or_f16(static_cast<float>(xacc[j][2 * q + 1]) * (1.f / 4294967296.f), BF16);
          const unsigned lo = f32_to_act<BF16>(act_lo<BF16>(xr[j][q]) + r0), hi = f32_to_act<BF16>(act_hi<BF16>(xr[j][q]) + r1);
          xr[j][q] = lo | (hi << 16);
        }
        const int hu = tid + NT * j;
        if (blockIdx.x == 0 && a.x_out && hu < nhu)
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
        const bool live = (tid + NT * j) < nhu;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float lo = act_lo<BF16>(xr[j][q]), hi = act_hi<BF16>(xr[j][q]);
          ssq += live ? (lo * lo + hi * hi) : 0.f;
        }
      }
    }
    ssq = wave_sum(ssq);
    if (lane == 0) red[wave] = ssq;
    lds_barrier();
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < MX_NW; ++i) tot += red[i];
    rstd = rsqrtf(tot / static_cast<float>(K) + a.eps);
  }
  MX_STAMP(9);
#pragma unroll
  for (int j = 0; j < UPT_MAX; ++j) {
    if (j < upt && (tid & ~63) + NT * j < nhu) {                           // wave-uniform: idle waves skip the arithmetic
      const int hu = tid + NT * j;
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
      // ever meet each other: K is a multiple of 128)
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
  lds_barrier();
  MX_STAMP(2);

  // ---------------- 2. the tile: SPP sets of 4 MFMAs ----------------
  // A operand of lane (m = lane & 15 -> slot m >> 2, limb m & 3; lane group ga): limb bytes of group 4S + ga when the slot
  // is the lane group's own, zeros otherwise (a 32-byte zero block read with stride 0)
  const int m = lane & 15, ga = lane >> 4;
  const bool mine = (m >> 2) == ga;
  const unsigned a_base = mine ? static_cast<unsigned>(ga * 512 + (m & 3) * 32) : static_cast<unsigned>(reinterpret_cast<unsigned char*>(zero_blk) - smem);
  const unsigned a_step_t = mine ? 128u : 0u, a_step_s = mine ? 2048u : 0u;
  float yacc = 0.f;
  const int scale_one = 127;
  // this lane's 16-byte pieces of row nrow: piece 4 (gb & 1) + (gb >> 1) of each 128-byte set chunk, and the one 32 bytes on
  const int pc = 4 * (gb & 1) + (gb >> 1);
  const unsigned char* const wrow = HALVES ? wreg + ((nrow & 3) + 4 * (nrow >> 3)) * MX_BST + ((nrow >> 2) & 1) * 16 + pc * 32
                                           : wreg + nrow * RSTRIDE + pc * 16;
  auto do_set = [&](int s) __attribute__((always_inline)) {
    const int S = min(set0 + s, nsets - 1);
    const bool slive = set0 + s < nsets;
    v4f_t c = {0.f, 0.f, 0.f, 0.f};
    // The E8M0 scale of A's hardware k-block kb comes from lane (row m, lane group kb) -- NOT from the lane that holds the
    // bytes: a lane group's 32 bytes straddle the k-blocks (ga >> 1) and (ga >> 1) + 2 (measured: one-hot x, scratch/dbg_mx3.py).
    // Row m only has data in its own slot's lane group, so all four lanes of row m carry the exponent of slot m >> 2.
    const int sa = scl[4 * S + (m >> 2)];
    u32x4 wAv, wBv;
    if constexpr (HALVES) {
      const unsigned char* p = wrow + (s >> 2) * (8 * MX_BST) + (s & 3) * 256;       // half, then 8 pieces x 32 bytes per set
      wAv = *reinterpret_cast<const u32x4*>(p);
      wBv = *reinterpret_cast<const u32x4*>(p + 64);
    } else {
      wAv = *reinterpret_cast<const u32x4*>(wrow + s * 128);
      wBv = *reinterpret_cast<const u32x4*>(wrow + s * 128 + 32);
    }
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
    const float contrib = fmaf(ze, xsum[4 * S + ga], (sc + sc) * tsum);
    yacc += slive ? contrib : 0.f;
  };
#ifdef GBX_MX_NOCOMPUTE
  if (a.K < 0)
#endif
  {
    if constexpr (HALVES) {
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // the first K half of every row has landed; the second is in flight
#pragma unroll
      for (int s = 0; s < 4; ++s) do_set(s);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int s = 4; s < 8; ++s) do_set(s);
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every row segment of this wave has landed in its LDS region
#pragma unroll
      for (int s = 0; s < SPP; ++s) do_set(s);
    }
  }
  MX_STAMP(4);
  // the four lane groups hold the four group slots of the same 16 rows
  yacc += __shfl_xor(yacc, 16);
  yacc += __shfl_xor(yacc, 32);
  if (kp > 1) {                                                           // k parts of one tile meet in LDS, summed in part order
    if (lane < 16) part[wave * 16 + lane] = yacc;
    lds_barrier();
    if (kpart == 0 && lane < 16) {
      float t = yacc;
      for (int p = 1; p < kp; ++p) t += part[(wave + p) * 16 + lane];
      yacc = t;
    }
  }
  // ---------------- 3. epilogue: lanes 0..15 of the tile's first wave hold the 16 row totals ----------------
  if (wlive && kpart == 0) {
    if (swiglu) {
      const float up = __shfl(yacc, (lane & 7) + 8);
      const int r = tile * 8 + lane;
      if (lane < 8 && r < N) {
        const float g = round_bf16_or_f16(yacc, BF16), u = round_bf16_or_f16(up, BF16);
        const float sl = round_bf16_or_f16(g / (1.f + __expf(-g)), BF16);
        y0p[r] = f32_to_act<BF16>(sl * u);
      }
    } else if (lane < 16) {
      const int row = trow0 + lane;
      float v = yacc;
      if (a.residual) v = act_to_f32<BF16>(a.residual[row]) + round_bf16_or_f16(v, BF16);
      ty[row] = f32_to_act<BF16>(v);
    }
  }
  MX_STAMP(6);
}

// ---- launch: tiles per workgroup and k parts per tile so that every CU gets (about) one workgroup --------------------------
bool gemv_mx_eligible(const GemvArgs& a, int M) {
  static const bool off = [] { const char* e = getenv("GBX_GEMV_MX"); return e && e[0] == '0'; }();
  if (off || M != 1 || !a.sz || a.expert_idx || a.bias || a.log2g != 7) return false;
  if (a.K % MX_SETV != 0 || a.K > 16384) return false;
  if (a.norm_w && a.K > 8192) return false;
  for (int i = 0; i < a.nseg; ++i) if (a.seg[i].N % 16 != 0) return false;
  if (a.swiglu && (a.seg[0].N % 8 != 0)) return false;
  if (a.x_acc && a.K > 8192) return false;
  return true;
}

template <int SPP>
static hipError_t launch_mx_spp(const GemvArgs& a, bool bf16, int blocks, int tpw, int kp, hipStream_t s) {
  const size_t shmem = mx_lds_bytes(a.K, SPP);
  dim3 grid(blocks), block(MX_NW * 64);
  static bool attr_done[2][2] = {{false, false}, {false, false}};
  auto set_attr = [&](auto kern, bool& done) {
    if (!done) { hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }
    done = true;
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

// tiles per workgroup (about one workgroup per CU) and k parts per tile (as many waves per tile as the workgroup has left)
static bool mx_plan(const GemvArgs& a, int& tpw, int& kp, int& spp, int& blocks) {
  const int ntiles = a.swiglu ? (a.N + 7) / 8 : (a.N + 15) / 16;
  const int nsets = a.K / MX_SETV;
  tpw = (ntiles + GEMV_CUS / 2) / GEMV_CUS;
  tpw = tpw < 1 ? 1 : (tpw > MX_NW ? MX_NW : tpw);
  kp = 0;
  for (int c = MX_NW / tpw; c >= 1; --c) {
    if (nsets % c) continue;
    const int sp = nsets / c;
    if (sp == 1 || sp == 2 || sp == 4 || sp == 7 || sp == 8) { kp = c; break; }
  }
  if (!kp) return false;
  spp = nsets / kp;
  blocks = (ntiles + tpw - 1) / tpw;
  return mx_lds_bytes(a.K, spp) <= 160 * 1024;
}

hipError_t launch_gemv_mx(const GemvArgs& a, bool bf16, hipStream_t s) {
  int tpw, kp, spp, blocks;
  if (!mx_plan(a, tpw, kp, spp, blocks)) return hipErrorInvalidValue;
  switch (spp) {
    case 1: return launch_mx_spp<1>(a, bf16, blocks, tpw, kp, s);
    case 2: return launch_mx_spp<2>(a, bf16, blocks, tpw, kp, s);
    case 4: return launch_mx_spp<4>(a, bf16, blocks, tpw, kp, s);
    case 7: return launch_mx_spp<7>(a, bf16, blocks, tpw, kp, s);
    default: return launch_mx_spp<8>(a, bf16, blocks, tpw, kp, s);
  }
}

bool gemv_mx_shape_ok(const GemvArgs& a) {
  int tpw, kp, spp, blocks;
  return mx_plan(a, tpw, kp, spp, blocks);
}

}  // namespace gbx
