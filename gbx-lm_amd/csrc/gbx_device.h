// Device-side helpers shared by every gfx950 kernel of the GBA QuantizedLinear path.
// CDNA4 only: 64-wide wavefronts, DPP row ops, v_dot2_f32_f16.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gbx {

using u32x2 = unsigned __attribute__((ext_vector_type(2)));
using u32x3 = unsigned __attribute__((ext_vector_type(3)));
using u32x4 = unsigned __attribute__((ext_vector_type(4)));
using h2 = _Float16 __attribute__((ext_vector_type(2)));

constexpr int WAVE = 64;

// NOTE (hipcc / ROCm 7.2): __builtin_bit_cast applied directly to an ext-vector ELEMENT expression (v.y) reads
// element 0.  Always go through a by-value scalar parameter.
__device__ __forceinline__ h2 as_h2(unsigned u) { return __builtin_bit_cast(h2, u); }
__device__ __forceinline__ unsigned as_u32(h2 v) { return __builtin_bit_cast(unsigned, v); }

// ---- 16-bit activation formats ------------------------------------------------------------
__device__ __forceinline__ float bf16_to_f32(unsigned short b) {
  return __uint_as_float(static_cast<unsigned>(b) << 16);
}
__device__ __forceinline__ float f16_to_f32(unsigned short b) {
  return static_cast<float>(__builtin_bit_cast(_Float16, b));
}
template <bool BF16>
__device__ __forceinline__ float act_to_f32(unsigned short b) {
  if constexpr (BF16) return bf16_to_f32(b); else return f16_to_f32(b);
}
// One rounding (RNE) from fp32 to the activation dtype; a plain cast keeps NaN a NaN on gfx950
// (v_cvt_pk_bf16_f32), unlike integer bias-and-shift.
template <bool BF16>
__device__ __forceinline__ unsigned short f32_to_act(float f) {
  if constexpr (BF16) {
    __bf16 v = static_cast<__bf16>(f);
    return __builtin_bit_cast(unsigned short, v);
  } else {
    _Float16 v = static_cast<_Float16>(f);
    return __builtin_bit_cast(unsigned short, v);
  }
}
// low / high 16-bit halves of a packed pair of activations
template <bool BF16>
__device__ __forceinline__ float act_lo(unsigned w) { return act_to_f32<BF16>(static_cast<unsigned short>(w & 0xFFFFu)); }
template <bool BF16>
__device__ __forceinline__ float act_hi(unsigned w) { return act_to_f32<BF16>(static_cast<unsigned short>(w >> 16)); }

// ---- DPP wavefront reductions -------------------------------------------------------------
// dpp_ctrl encodings (GCN3+/CDNA): quad_perm = 8-bit selector, row_half_mirror 0x141,
// row_mirror 0x140, row_bcast15 0x142, row_bcast31 0x143.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_f32(float v) {
  int r = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false);
  return __builtin_bit_cast(float, r);
}
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ unsigned dpp_u32(unsigned v) {
  return static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), CTRL, ROW_MASK, 0xF, false));
}

// Sum over the 64 lanes; the result is returned wave-uniform (read from lane 63).
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_f32<0xB1>(v);         // quad_perm [1,0,3,2]
  v += dpp_f32<0x4E>(v);         // quad_perm [2,3,0,1]
  v += dpp_f32<0x141>(v);        // row_half_mirror : 8-lane sums
  v += dpp_f32<0x140>(v);        // row_mirror      : 16-lane (row) sums in every lane of the row
  v += dpp_f32<0x142, 0xA>(v);   // row_bcast15 into rows 1,3
  v += dpp_f32<0x143, 0xC>(v);   // row_bcast31 into rows 2,3 -> lane 63 holds the total
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// order-preserving map float -> uint (so that the DPP integer max serves floats of either sign) and back
__device__ __forceinline__ unsigned f2key(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// Sums of TWO accumulators at once: v_permlane32_swap folds the wave in half (lanes 0-31 carry `a`, lanes 32-63 carry
// `b`), four DPP steps finish the 16-lane rows, row_bcast15 joins the two rows of each half.
// Result: a's total in lanes 16..31, b's total in lanes 48..63 (callers read lane 31 / lane 63).
__device__ __forceinline__ float wave_sum2(float a, float b) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  const unsigned rx = r[0], ry = r[1];
  float v = __uint_as_float(rx) + __uint_as_float(ry);
  v += dpp_f32<0xB1>(v);
  v += dpp_f32<0x4E>(v);
  v += dpp_f32<0x141>(v);
  v += dpp_f32<0x140>(v);
  v += dpp_f32<0x142, 0xA>(v);
  return v;
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
  v = max(v, dpp_u32<0xB1>(v));
  v = max(v, dpp_u32<0x4E>(v));
  v = max(v, dpp_u32<0x141>(v));
  v = max(v, dpp_u32<0x140>(v));
  v = max(v, dpp_u32<0x142, 0xA>(v));
  v = max(v, dpp_u32<0x143, 0xC>(v));
  return static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(v), 63));
}
// sums inside aligned groups of 2 / 4 lanes (every lane gets its group's sum)
__device__ __forceinline__ float pair_sum(float v) { return v + dpp_f32<0xB1>(v); }
__device__ __forceinline__ float quad_sum(float v) { v += dpp_f32<0xB1>(v); return v + dpp_f32<0x4E>(v); }

// 2^-32 fixed point (two's complement int64 as lo / hi words) -> fp32, one rounding: hi * 2^32 + lo is exact in fp64 below 2^53 and
// fp64 adds / converts are single instructions on CDNA (the compiler's int64 -> float sequence is ~20)
__device__ __forceinline__ float fx32_to_f32(unsigned lo, unsigned hi) {
  const double d = __builtin_fma(static_cast<double>(static_cast<int>(hi)), 4294967296.0, static_cast<double>(lo));
  return static_cast<float>(d) * 2.3283064365386963e-10f;
}

// exact 2^e as fp32, e in [-126, 127]
__device__ __forceinline__ float exp2i(int e) { return __uint_as_float(static_cast<unsigned>(127 + e) << 23); }

// ---- generic field extraction, used by the debug / fallback kernels ---
template <int BITS>
__device__ __forceinline__ unsigned field_of_row(const uint32_t* __restrict__ row, int k) {
  const int o = k * BITS, wi = o >> 5, s = o & 31;
  constexpr unsigned mask = (1u << BITS) - 1u;
  unsigned q = (row[wi] >> s) & mask;
  if (BITS == 3 && s + BITS > 32) q |= (row[wi + 1] << (32 - s)) & mask;
  return q;
}

// packed (scale | zero << 16) dword -> fp32
template <bool BF16> __device__ __forceinline__ float sz_scale(unsigned v) {
  if constexpr (BF16) return __uint_as_float(v << 16);
  else return act_to_f32<false>(static_cast<unsigned short>(v & 0xFFFFu));
}
template <bool BF16> __device__ __forceinline__ float sz_zero(unsigned v) {
  if constexpr (BF16) return __uint_as_float(v & 0xFFFF0000u);
  else return act_to_f32<false>(static_cast<unsigned short>(v >> 16));
}


}  // namespace gbx
