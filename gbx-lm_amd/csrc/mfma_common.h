// Shared pieces of the MFMA dequant-GEMM kernels (mfma_gemm.hip: prefill tiles; mfma_skinny.hip: M <= 64, split-K).
#pragma once
#include "gbx_device.h"

namespace gbx {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct MfmaArgs {
  const unsigned short* x;      // [M, K]
  const uint32_t* qweight;      // [N, K*bits/32]
  const unsigned short* scales; // [N, K/g]
  const unsigned short* zeros;  // [N, K/g]
  const unsigned short* bias;   // [N] or null
  unsigned short* y;            // [M, N]
  int M, N, K, log2g;
  // split-K kernel only: up to two more matrices that share x, K, bits and group size (q/k/v, gate/up) in ONE launch;
  // N is then the sum of the row counts, the segments are tiled in order (each row count a multiple of the tile)
  int nseg;                     // 0 or 1: single matrix
  int n_seg[3];                 // rows of segment 0..2 (segment 0 described by the fields above)
  const uint32_t* qweight_x[2];
  const unsigned short* scales_x[2];
  const unsigned short* zeros_x[2];
  const unsigned short* bias_x[2];
  unsigned short* y_x[2];
  // batch kernel only: k split ACROSS workgroups (gridDim.y = ksplit slices of the k range); slice s writes its fp32
  // partial tile to ws[s][M][N] (caller-owned workspace) and a second launch sums the slices
  float* ws;
  int ksplit;
  // batch kernel only: nseg == 2 (gate_proj, up_proj of one block), every workgroup takes rows [n0, n0 + R/2) of BOTH and
  // writes y[m, n] = silu(gate[m, n]) * up[m, n] (the reference's op-by-op roundings, qllama.py:115) to y [M, n_seg[0]]
  int swiglu;
  // batch kernel only, with ws: leave the fp32 slices in ws and skip the summing launch (the consumer folds the sum in:
  // gbx_add_rmsnorm_slices)
  int partials_only;
  // prefill kernel, BITS == 16 instantiation: the weights already dequantised to the activation dtype, [N, K] row-major
  // (one gbx_dequantize pass per call into the caller's workspace, shared by all M / 256 row tiles instead of being redone
  // by every one of them)
  const unsigned short* wdense;
#ifdef GBX_PSTAMPS
  long long* stamps;            // diagnostic build only: [workgroup][wave][4] s_memtime totals
#endif
};

// mfma_batch.hip: 2 <= M <= 32, 2/4 bits, group 128 (no-dequant matrix-core kernel); false -> use the split-K kernel
bool qgemm_batch_supported(int M, int K, int bits, int log2g);
hipError_t launch_qgemm_batch(const MfmaArgs& a, int bits, bool bf16, hipStream_t s);
// second launch of a k-split GEMM: y = round(sum of the fp32 slices ws[slices][M][N]) (+ bias as a second rounding)
hipError_t launch_splitk_finish(const float* ws, const unsigned short* bias, unsigned short* y, int slices, int M, int N, bool bf16,
                                hipStream_t s);
// same for the prefill tiling at short prompts (mfma_gemm.hip)
size_t qgemm_prefill_workspace_bytes(int M, int N, int K);
// fp32 workspace bytes with which launch_qgemm_batch splits k across workgroups (0: it would not)
size_t qgemm_batch_workspace_bytes(int M, int N, int K, int bits, int log2g);
// k slices launch_qgemm_batch writes with a workspace of that size (1: it does not split)
int qgemm_batch_slices(int M, int N, int K, int bits, int log2g);

constexpr int KC = 128;                 // k-values per chunk (one LDS stage of activations)
constexpr int XROW = KC * 2 + 16;       // LDS bytes per activation row: 256 B + 16 B pad (conflict-free ds_read_b128)

template <bool BF16>
__device__ __forceinline__ f32x16 mfma(const u32x4& a, const u32x4& b, const f32x16& c) {
  if constexpr (BF16) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
// two fp32 -> one packed pair of activations, ONE v_cvt_pk_{bf16,f16}_f32 (round to nearest even)
template <bool BF16>
__device__ __forceinline__ unsigned pack2(float a, float b) {
  f32x2v v; v.x = a; v.y = b;
  if constexpr (BF16) return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2v));
  else return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2v));
}

// Octet o (compile-time) of a chunk's packed words: the 8 consecutive fields 8o .. 8o+7 of the row, as a bit window
// starting at bit 0 (4 bits: the whole word; 2 bits: a 16-bit half; 3 bits: 24 bits realigned across words).
template <int BITS>
__device__ __forceinline__ unsigned octet(const unsigned* wv, int o) {
  if constexpr (BITS == 4) return wv[o];
  else if constexpr (BITS == 2) return (o & 1) ? (wv[o >> 1] >> 16) : wv[o >> 1];
  else {
    constexpr int LAST = KC * 3 / 32 - 1;
    const int b = 24 * o, w = b >> 5;
    return __builtin_amdgcn_alignbit(wv[w + 1 <= LAST ? w + 1 : LAST], wv[w], b & 31);
  }
}
// window -> the 8 integer fields (natural k order) as fp32
template <int BITS>
__device__ __forceinline__ void decode8(unsigned win, float* q) {
  if constexpr (BITS == 4) {
    const unsigned e = win & 0x0F0F0F0Fu, o = (win >> 4) & 0x0F0F0F0Fu;  // even / odd nibbles as bytes
    q[0] = static_cast<float>(e & 0xFFu);         q[1] = static_cast<float>(o & 0xFFu);
    q[2] = static_cast<float>((e >> 8) & 0xFFu);  q[3] = static_cast<float>((o >> 8) & 0xFFu);
    q[4] = static_cast<float>((e >> 16) & 0xFFu); q[5] = static_cast<float>((o >> 16) & 0xFFu);
    q[6] = static_cast<float>(e >> 24);           q[7] = static_cast<float>(o >> 24);
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) q[j] = static_cast<float>((win >> (BITS * j)) & ((1u << BITS) - 1u));
  }
}
// The 8 integer fields of weight row n for k-step t of a 32x32x16 MFMA (16 values per step), lane half h.
template <int BITS>
__device__ __forceinline__ void fields8(const unsigned* wv, int t, int h, float* q) {
  const unsigned w0 = octet<BITS>(wv, 2 * t), w1 = octet<BITS>(wv, 2 * t + 1);
  decode8<BITS>(h ? w1 : w0, q);
}
// Same for a 16x16x32 MFMA (32 values per step): lane quarter kb = 0..3.
template <int BITS>
__device__ __forceinline__ void fields8q(const unsigned* wv, int t, int kb, float* q) {
  const unsigned w0 = octet<BITS>(wv, 4 * t), w1 = octet<BITS>(wv, 4 * t + 1);
  const unsigned w2 = octet<BITS>(wv, 4 * t + 2), w3 = octet<BITS>(wv, 4 * t + 3);
  decode8<BITS>(kb == 0 ? w0 : kb == 1 ? w1 : kb == 2 ? w2 : w3, q);
}

typedef float f32x4v __attribute__((ext_vector_type(4)));
template <bool BF16>
__device__ __forceinline__ f32x4v mfma16(const u32x4& a, const u32x4& b, const f32x4v& c) {
  if constexpr (BF16) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

}  // namespace gbx
