// Packed-row format helpers shared by the decode GEMV kernels (gemv_impl.h, dgemv_impl.h): slot geometry per bit width,
// the in-place field masks (fp16 subnormals) and the v_dot2 unit products, and the LDS geometry of the activation image.
// Role: the per-weight arithmetic of mx.quantized_matmul at M == 1 (gbx_lm/models/quantized_linear_gba.py:195-203).
#pragma once
#include "gbx_device.h"

namespace gbx {
// Per bit width: slot geometry and the pair table.  Pair i of a unit holds fields (ka(i), kb(i)) sitting at bit
// pos(i) of the low / high 16-bit half of the register the dot routine masks.  (constexpr functions, not static
// arrays: a static constexpr array odr-used in device code has no device-side storage.)
template <int BITS> struct Fmt;
template <> struct Fmt<2> {
  static constexpr int SLOT_WORDS = 4, VPL = 64, NSUB = 2, UNIT_VALS = 16, UNIT_PAIRS = 8;
  __host__ __device__ static constexpr int ka(int i) { return i; }
  __host__ __device__ static constexpr int kb(int i) { return 8 + i; }
  __host__ __device__ static constexpr int pos(int i) { return i < 5 ? 2 * i : 2 * (i - 5); }
};
template <> struct Fmt<4> {
  static constexpr int SLOT_WORDS = 4, VPL = 32, NSUB = 1, UNIT_VALS = 8, UNIT_PAIRS = 4;
  __host__ __device__ static constexpr int ka(int i) { return i; }
  __host__ __device__ static constexpr int kb(int i) { return 4 + i; }
  __host__ __device__ static constexpr int pos(int i) { return (i & 1) ? 4 : 0; }
};
template <> struct Fmt<3> {
  static constexpr int SLOT_WORDS = 3, VPL = 32, NSUB = 1, UNIT_VALS = 32, UNIT_PAIRS = 16;
  __host__ __device__ static constexpr int ka(int i) {
    constexpr int t[16] = {0, 1, 2, 3, 4, 6, 7, 8, 9, 11, 12, 13, 14, 15, 5, 10};
    return t[i];
  }
  __host__ __device__ static constexpr int kb(int i) { return ka(i) + 16; }
  __host__ __device__ static constexpr int pos(int i) {
    constexpr int t[16] = {0, 3, 6, 0, 3, 2, 5, 0, 3, 1, 4, 7, 0, 3, 0, 0};
    return t[i];
  }
};

__device__ __forceinline__ float dot2h(unsigned wbits, h2 xv, float acc) {
  return __builtin_amdgcn_fdot2(as_h2(wbits), xv, acc, false);
}

// ---- per-unit dot products: masks select fields in place (fp16 subnormals) ---------------------
// 2-bit: one word = 16 fields; pairs (f_j, f_{8+j}).
__device__ __forceinline__ float dot_unit_b2(unsigned w, const h2* xp, float acc) {
#ifdef GBX_EXP_SKIPDOT   // timing experiment only (wrong results): a quarter of the vector-ALU work per packed word
  acc = dot2h(w & 0x00030003u, xp[0], acc);
  return dot2h(w & 0x000C000Cu, xp[1], acc);
#endif
  acc = dot2h(w & 0x00030003u, xp[0], acc);
  acc = dot2h(w & 0x000C000Cu, xp[1], acc);
  acc = dot2h(w & 0x00300030u, xp[2], acc);
  acc = dot2h(w & 0x00C000C0u, xp[3], acc);
  acc = dot2h(w & 0x03000300u, xp[4], acc);
  const unsigned t = w >> 10;
  acc = dot2h(t & 0x00030003u, xp[5], acc);
  acc = dot2h(t & 0x000C000Cu, xp[6], acc);
  acc = dot2h(t & 0x00300030u, xp[7], acc);
  return acc;
}
// 4-bit: one word = 8 fields; pairs (f_j, f_{4+j}).
__device__ __forceinline__ float dot_unit_b4(unsigned w, const h2* xp, float acc) {
  acc = dot2h(w & 0x000F000Fu, xp[0], acc);
  acc = dot2h(w & 0x00F000F0u, xp[1], acc);
  const unsigned t = w >> 8;
  acc = dot2h(t & 0x000F000Fu, xp[2], acc);
  acc = dot2h(t & 0x00F000F0u, xp[3], acc);
  return acc;
}
// 3-bit: three words = 32 fields = six 16-bit halves H0..H5; (H0,H3), (H1,H4), (H2,H5) hold whole
// fields at equal in-half offsets; fields 5,10,21,26 straddle halves and are realigned by funnel shifts.
__device__ __forceinline__ float dot_unit_b3(unsigned w0, unsigned w1, unsigned w2, const h2* xp, float acc) {
  const unsigned A = (w0 & 0x0000FFFFu) | (w1 & 0xFFFF0000u);   // H0 | H3 : fields (k, 16+k), k = 0..4
  acc = dot2h(A & 0x00070007u, xp[0], acc);
  acc = dot2h(A & 0x00380038u, xp[1], acc);
  acc = dot2h(A & 0x01C001C0u, xp[2], acc);
  const unsigned A9 = A >> 9;
  acc = dot2h(A9 & 0x00070007u, xp[3], acc);
  acc = dot2h(A9 & 0x00380038u, xp[4], acc);
  const unsigned B = __builtin_amdgcn_alignbit(w2, w0, 16);     // H1 | H4 : fields (6+i, 22+i), i = 0..3
  acc = dot2h(B & 0x001C001Cu, xp[5], acc);
  acc = dot2h(B & 0x00E000E0u, xp[6], acc);
  const unsigned B8 = B >> 8;
  acc = dot2h(B8 & 0x00070007u, xp[7], acc);
  acc = dot2h(B8 & 0x00380038u, xp[8], acc);
  const unsigned C = (w1 & 0x0000FFFFu) | (w2 & 0xFFFF0000u);   // H2 | H5 : fields (11+i, 27+i), i = 0..4
  acc = dot2h(C & 0x000E000Eu, xp[9], acc);
  acc = dot2h(C & 0x00700070u, xp[10], acc);
  acc = dot2h(C & 0x03800380u, xp[11], acc);
  const unsigned C10 = C >> 10;
  acc = dot2h(C10 & 0x00070007u, xp[12], acc);
  acc = dot2h(C10 & 0x00380038u, xp[13], acc);
  // straddlers: (f5, f21) and (f10, f26)
  const unsigned S1 = ((w0 >> 15) & 0x0000FFFFu) | (__builtin_amdgcn_alignbit(w2, w1, 15) & 0xFFFF0000u);
  acc = dot2h(S1 & 0x00070007u, xp[14], acc);
  const unsigned S2 = (__builtin_amdgcn_alignbit(w1, w0, 30) & 0x0000FFFFu) | ((w2 << 2) & 0xFFFF0000u);
  acc = dot2h(S2 & 0x00070007u, xp[15], acc);
  return acc;
}

template <int BITS> struct SlotWords;
template <> struct SlotWords<2> { using type = u32x4; };
template <> struct SlotWords<4> { using type = u32x4; };
template <> struct SlotWords<3> { using type = u32x3; };

// dot of one slot -> per-sub-slot integer sums (scaled by 2^(E-24))
template <int BITS>
__device__ __forceinline__ void dot_slot(const typename SlotWords<BITS>::type& w, const h2* xp, float* d) {
  if constexpr (BITS == 2) {
    d[0] = dot_unit_b2(w.y, xp + 8, dot_unit_b2(w.x, xp, 0.f));
    d[1] = dot_unit_b2(w.w, xp + 24, dot_unit_b2(w.z, xp + 16, 0.f));
  } else if constexpr (BITS == 4) {
    float a = dot_unit_b4(w.x, xp, 0.f);
    float b = dot_unit_b4(w.y, xp + 4, 0.f);
    a = dot_unit_b4(w.z, xp + 8, a);
    b = dot_unit_b4(w.w, xp + 12, b);
    d[0] = a + b;
  } else {
    d[0] = dot_unit_b3(w.x, w.y, w.z, xp, 0.f);
  }
}

// ---- the same unit products for R rows at once, the rows' (and sub-slots') dependent v_dot2c chains INTERLEAVED ------------------
// Every chain accumulates in exactly the order of dot_slot (bit-identical per row); what changes is the instruction order: a wave
// that is alone on its SIMD (the short decode launches) issues one of 2R independent chains per slot instead of waiting ~7 cycles
// for the previous v_dot2c of a single chain (measured: 4 rows of 4096 2-bit values 0.88 -> see DESIGN 3.1d).
template <int R>
__device__ __forceinline__ void dot_words_b2_rows(const unsigned (&wa)[R], const unsigned (&wb)[R], const h2* xa, const h2* xb,
                                                  float (&da)[R], float (&db)[R]) {
  unsigned ta[R], tb[R];
#pragma unroll
  for (int r = 0; r < R; ++r) { ta[r] = wa[r] >> 10; tb[r] = wb[r] >> 10; }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const unsigned m = 0x00030003u << (2 * (i < 5 ? i : i - 5));
#pragma unroll
    for (int r = 0; r < R; ++r) {
      da[r] = dot2h((i < 5 ? wa[r] : ta[r]) & m, xa[i], da[r]);
      db[r] = dot2h((i < 5 ? wb[r] : tb[r]) & m, xb[i], db[r]);
    }
  }
}
template <int R>
__device__ __forceinline__ void dot_words_b4_rows(const unsigned (&wa)[R], const unsigned (&wb)[R], const h2* xa, const h2* xb,
                                                  float (&da)[R], float (&db)[R]) {
  unsigned ta[R], tb[R];
#pragma unroll
  for (int r = 0; r < R; ++r) { ta[r] = wa[r] >> 8; tb[r] = wb[r] >> 8; }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned m = (i & 1) ? 0x00F000F0u : 0x000F000Fu;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      da[r] = dot2h((i < 2 ? wa[r] : ta[r]) & m, xa[i], da[r]);
      db[r] = dot2h((i < 2 ? wb[r] : tb[r]) & m, xb[i], db[r]);
    }
  }
}
// d[r][s]: per-row, per-sub-slot sums of one slot of R rows
template <int BITS, int R>
__device__ __forceinline__ void dot_slot_rows(const typename SlotWords<BITS>::type (&w)[R], const h2* xp, float (&d)[R][Fmt<BITS>::NSUB]) {
  if constexpr (BITS == 2) {
    unsigned wx[R], wy[R], wz[R], ww[R];
    float d0[R], d1[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { wx[r] = w[r].x; wy[r] = w[r].y; wz[r] = w[r].z; ww[r] = w[r].w; d0[r] = 0.f; d1[r] = 0.f; }
    dot_words_b2_rows<R>(wx, wz, xp, xp + 16, d0, d1);          // sub-slot 0: words x, y; sub-slot 1: words z, w
    dot_words_b2_rows<R>(wy, ww, xp + 8, xp + 24, d0, d1);
#pragma unroll
    for (int r = 0; r < R; ++r) { d[r][0] = d0[r]; d[r][1] = d1[r]; }
  } else if constexpr (BITS == 4) {
    unsigned wx[R], wy[R], wz[R], ww[R];
    float a[R], b[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { wx[r] = w[r].x; wy[r] = w[r].y; wz[r] = w[r].z; ww[r] = w[r].w; a[r] = 0.f; b[r] = 0.f; }
    dot_words_b4_rows<R>(wx, wy, xp, xp + 4, a, b);             // chains a: words x, z; b: words y, w (dot_slot's order)
    dot_words_b4_rows<R>(wz, ww, xp + 8, xp + 12, a, b);
#pragma unroll
    for (int r = 0; r < R; ++r) d[r][0] = a[r] + b[r];
  } else {
    // 3 bits: one chain of 16 per row; the rows interleave at the granularity the compiler's scheduler gives them
#pragma unroll
    for (int r = 0; r < R; ++r) d[r][0] = dot_unit_b3(w[r].x, w[r].y, w[r].z, xp, 0.f);
  }
}

// LDS geometry: slot s keeps its VPL/2 fp16 pairs at h2 index s*SSTR (16 B of padding per slot keeps the
// one-time ds_read_b128 of 64 lanes at distinct bank quads); then one {sum(x), 2^(24-E)} pair per 32-value sub-slot.
template <int BITS> struct Lds {
  static constexpr int PAIRS = Fmt<BITS>::VPL / 2;
  static constexpr int SSTR = PAIRS + 4;
  __host__ __device__ static constexpr size_t bytes(int K) {
    const int nslots = K / Fmt<BITS>::VPL;
    return static_cast<size_t>(nslots) * SSTR * 4 + static_cast<size_t>(nslots) * Fmt<BITS>::NSUB * 8 + 16;
  }
};

__device__ __forceinline__ float round_bf16_or_f16(float v, bool bf16) {
  return bf16 ? bf16_to_f32(f32_to_act<true>(v)) : f16_to_f32(f32_to_act<false>(v));
}

}  // namespace gbx
