// Batched GBA QuantizedLinear (M > 1): y[M,N] = x[M,K] . dequant(W[N,K])^T on the matrix cores (gfx950 MFMA).
//
// Replaces mx.quantized_matmul(..., transpose=True) for prefill / batched decode
// (gbx_lm/models/quantized_linear_gba.py:195-203 when x is [B, L, K] with B*L > 1; e.g. the chunked prefill at
// gbx_lm/utils.py:312-319 and the batched eval at gbx_lm/evaluate.py:144-185).
//
// The dequantised tile is a genuine dense contraction here, so it runs on v_mfma_f32_32x32x16_{bf16,f16}:
//   * B operand = the packed integers themselves, turned into exact 16-bit floats by OR-ing the field into a magic
//     exponent (bf16: 0x4300|q = 128+q, fp16: 0x6400|q = 1024+q) -- two fields per v_and_or, no int->float convert.
//     Lane (n = l&31, h = l>>5) owns 8 k-values of weight row n per k-step, which is ONE packed word at 4 bits and half
//     a word at 2 bits, so the fragment is built from the lane's own words with no cross-lane traffic.
//   * A operand = the activations, staged once per K-chunk in LDS in the same (permuted) k order the mask trick
//     produces, read back as one ds_read_b128 per fragment.
//   * per quantisation group the exact integer partial product P = sum_k x*(MAGIC+q) is folded into the fp32 total as
//     s*P + (z - MAGIC*s)*sum_k(x): the algebraic form of w = s*q + z, identical arithmetic to the decode GEMV.
// MFMA-bound for large M (2*M*N*K flops), HBM-bound for small M (weights read once per 128-column tile).
#include "gbx_device.h"
#include "kernels.h"

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
};

constexpr int KC = 128;                 // k-values per chunk (one LDS stage of activations)
constexpr int XROW = KC * 2 + 16;       // LDS bytes per activation row: 256 B + 16 B pad (conflict-free ds_read_b128)

template <bool BF16>
__device__ __forceinline__ f32x16 mfma(const u32x4& a, const u32x4& b, const f32x16& c) {
  if constexpr (BF16) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// position inside a 16-value k-step at which activation k-offset `o` must sit so that lane half h's ds_read_b128 (8
// consecutive positions starting at 8h) pairs with the weight fragment built below
template <int BITS>
__host__ __device__ constexpr int xpos(int o) {
  if (BITS == 4) {            // per 8-group: fragment order [f0,f4,f1,f5,f2,f6,f3,f7]
    const int g8 = o / 8, i = o % 8;
    return g8 * 8 + (i < 4 ? 2 * i : 2 * (i - 4) + 1);
  } else if (BITS == 2) {     // h = 0: [f0,f8,f1,f9,f2,f10,f3,f11], h = 1: [f4,f12,f5,f13,f6,f14,f7,f15]
    const int hi = (o >= 8), i = o % 8;          // hi: second element of its pair
    const int h = (i >= 4), j = i % 4;
    return h * 8 + 2 * j + hi;
  } else {
    return o;                 // 3-bit: natural order
  }
}

// B fragment (8 weights of row n for k-step t, lane half h) as four packed pairs of exact 16-bit floats MAGIC+q
template <int BITS, bool BF16>
__device__ __forceinline__ u32x4 wfrag(const unsigned* wv, int t, int h) {
  constexpr unsigned MG = BF16 ? 0x43004300u : 0x64006400u;
  u32x4 f;
  if constexpr (BITS == 4) {
    const unsigned w = h ? wv[2 * t + 1] : wv[2 * t];
    f.x = (w & 0x000F000Fu) | MG;
    f.y = ((w >> 4) & 0x000F000Fu) | MG;
    f.z = ((w >> 8) & 0x000F000Fu) | MG;
    f.w = ((w >> 12) & 0x000F000Fu) | MG;
  } else if constexpr (BITS == 2) {
    const unsigned w = wv[t] >> (8 * h);
    f.x = (w & 0x00030003u) | MG;
    f.y = ((w >> 2) & 0x00030003u) | MG;
    f.z = ((w >> 4) & 0x00030003u) | MG;
    f.w = ((w >> 6) & 0x00030003u) | MG;
  } else {
    // 24 bits starting at bit 48t + 24h of the chunk's 12-word bit-stream; both halves use compile-time word
    // indices (a lane-dependent register index would spill the words to scratch), then select
    const int b0 = 48 * t, b1 = 48 * t + 24;
    const unsigned win0 = __builtin_amdgcn_alignbit(wv[(b0 >> 5) + 1 < 12 ? (b0 >> 5) + 1 : 11], wv[b0 >> 5], b0 & 31);
    const unsigned win1 = __builtin_amdgcn_alignbit(wv[(b1 >> 5) + 1 < 12 ? (b1 >> 5) + 1 : 11], wv[b1 >> 5], b1 & 31);
    const unsigned win = h ? win1 : win0;
    f.x = ((win & 7u) | (((win >> 3) & 7u) << 16)) | MG;
    f.y = (((win >> 6) & 7u) | (((win >> 9) & 7u) << 16)) | MG;
    f.z = (((win >> 12) & 7u) | (((win >> 15) & 7u) << 16)) | MG;
    f.w = (((win >> 18) & 7u) | (((win >> 21) & 7u) << 16)) | MG;
  }
  return f;
}

// MT = 32-row activation tiles per workgroup (BM = 32*MT); 4 waves, each owns 32 output columns (BN = 128).
template <int BITS, int MT, bool BF16>
__global__ __launch_bounds__(256) void qgemm_mfma_kernel(const MfmaArgs a) {
  constexpr int BM = 32 * MT;
  constexpr int WPC = KC * BITS / 32;              // packed words per row per chunk (8 / 12 / 16)
  constexpr float MAGIC = BF16 ? 128.f : 1024.f;
  __shared__ __attribute__((aligned(16))) unsigned char xs_lds[BM * XROW];
  __shared__ float xsum_lds[BM][4];                // sum of x over each 32-value sub-group of the chunk

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int K = a.K, N = a.N, M = a.M;
  const int W = K / 32 * BITS, G = K >> a.log2g;
  const int m0 = blockIdx.y * BM;
  const int n = blockIdx.x * 128 + wave * 32 + r;  // this lane's output column = weight row
  const int nc = min(n, N - 1);
  const uint32_t* wrow = a.qweight + static_cast<size_t>(nc) * W;
  const int kpg = 1 << a.log2g;                    // k-values per group (32 / 64 / 128)
  const int steps_per_group = kpg / 16;

  f32x16 tot[MT], P[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) { tot[i] = f32x16{0}; P[i] = f32x16{0}; }

  for (int kc = 0; kc < K; kc += KC) {
    // ---- packed words of this lane's row for the chunk (both lane halves of a row read the same 32-64 bytes) ----
    unsigned wv[WPC];
    const uint32_t* wsrc = wrow + (kc / 32) * BITS;
#pragma unroll
    for (int q = 0; q < WPC; q += 4) {
      const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wsrc + q));
      wv[q] = v.x; wv[q + 1] = v.y; wv[q + 2] = v.z; wv[q + 3] = v.w;
    }
    // ---- stage the activation chunk [BM x 128] into LDS in fragment order, with per-32 sums ----
    __syncthreads();                               // previous chunk's fragments have been read
#pragma unroll
    for (int pz = tid; pz < BM * 8; pz += 256) {   // one piece = 16 consecutive k of one row
      const int m = pz >> 3, p16 = pz & 7;
      const int mg = min(m0 + m, M - 1);
      const unsigned* src = reinterpret_cast<const unsigned*>(a.x + static_cast<size_t>(mg) * K + kc + p16 * 16);
      const u32x4 v0 = *reinterpret_cast<const u32x4*>(src);
      const u32x4 v1 = *reinterpret_cast<const u32x4*>(src + 4);
      const unsigned raw[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
      unsigned short e[16], o[16];
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        e[2 * j] = static_cast<unsigned short>(raw[j] & 0xFFFFu);
        e[2 * j + 1] = static_cast<unsigned short>(raw[j] >> 16);
        s += act_lo<BF16>(raw[j]) + act_hi<BF16>(raw[j]);
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) o[xpos<BITS>(j)] = e[j];
      unsigned* dst = reinterpret_cast<unsigned*>(xs_lds + m * XROW + p16 * 32);
      u32x4 w0, w1;
      w0.x = o[0] | (static_cast<unsigned>(o[1]) << 16); w0.y = o[2] | (static_cast<unsigned>(o[3]) << 16);
      w0.z = o[4] | (static_cast<unsigned>(o[5]) << 16); w0.w = o[6] | (static_cast<unsigned>(o[7]) << 16);
      w1.x = o[8] | (static_cast<unsigned>(o[9]) << 16); w1.y = o[10] | (static_cast<unsigned>(o[11]) << 16);
      w1.z = o[12] | (static_cast<unsigned>(o[13]) << 16); w1.w = o[14] | (static_cast<unsigned>(o[15]) << 16);
      *reinterpret_cast<u32x4*>(dst) = w0;
      *reinterpret_cast<u32x4*>(dst + 4) = w1;
      const float s2 = pair_sum(s);                // pieces 2i, 2i+1 = one 32-value sub-group (adjacent lanes)
      if ((p16 & 1) == 0) xsum_lds[m][p16 >> 1] = s2;
    }
    __syncthreads();

    // ---- 8 k-steps of 16; flush the integer partial product at every quantisation-group boundary ----
#pragma unroll
    for (int t = 0; t < KC / 16; ++t) {
      const u32x4 bf = wfrag<BITS, BF16>(wv, t, h);
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const u32x4 af = *reinterpret_cast<const u32x4*>(xs_lds + (i * 32 + r) * XROW + (2 * t + h) * 16);
        P[i] = mfma<BF16>(af, bf, P[i]);
      }
      if (((t + 1) % steps_per_group) == 0) {      // group (or sub-group) complete
        const int gi = (kc + 16 * t) >> a.log2g;
        const float s = act_to_f32<BF16>(a.scales[static_cast<size_t>(nc) * G + gi]);
        const float z = act_to_f32<BF16>(a.zeros[static_cast<size_t>(nc) * G + gi]) - MAGIC * s;
        const int sub_lo = ((16 * (t + 1) - kpg) >> 5), sub_n = kpg >> 5;   // 32-value sub-sums that make up the group
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int row = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;       // C/D layout of the 32x32 tile
            float xsum = 0.f;
            for (int u = 0; u < sub_n; ++u) xsum += xsum_lds[row][sub_lo + u];
            tot[i][e] = fmaf(s, P[i][e], fmaf(z, xsum, tot[i][e]));
            P[i][e] = 0.f;
          }
        }
      }
    }
  }

  // ---- epilogue: one rounding to the activation dtype (+ bias) ----
  if (n < N) {
    const float b = a.bias ? act_to_f32<BF16>(a.bias[n]) : 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (m < M) {
          float v = tot[i][e];
          if (a.bias) v = act_to_f32<BF16>(f32_to_act<BF16>(v)) + b;      // separate `x + bias` rounding like the reference
          a.y[static_cast<size_t>(m) * N + n] = f32_to_act<BF16>(v);
        }
      }
    }
  }
}

template <int BITS, int MT>
static void launch_mt(const MfmaArgs& a, bool bf16, hipStream_t s) {
  dim3 grid((a.N + 127) / 128, (a.M + 32 * MT - 1) / (32 * MT)), block(256);
  if (bf16) hipLaunchKernelGGL((qgemm_mfma_kernel<BITS, MT, true>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((qgemm_mfma_kernel<BITS, MT, false>), grid, block, 0, s, a);
}

template <int BITS>
static void launch_bits(const MfmaArgs& a, bool bf16, hipStream_t s) {
  if (a.M <= 32) launch_mt<BITS, 1>(a, bf16, s);
  else launch_mt<BITS, 2>(a, bf16, s);      // MT = 4 spills with this register layout; revisit with a leaner flush
}

hipError_t launch_qgemm_mfma(const void* x, const uint32_t* qw, const void* sc, const void* ze, const void* bias, void* y,
                             int M, int N, int K, int bits, int log2g, bool bf16, hipStream_t s) {
  MfmaArgs a{static_cast<const unsigned short*>(x), qw, static_cast<const unsigned short*>(sc),
             static_cast<const unsigned short*>(ze), static_cast<const unsigned short*>(bias),
             static_cast<unsigned short*>(y), M, N, K, log2g};
  switch (bits) {
    case 2: launch_bits<2>(a, bf16, s); break;
    case 3: launch_bits<3>(a, bf16, s); break;
    case 4: launch_bits<4>(a, bf16, s); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace gbx
