// Batch-1 decode GEMV for GBA packed weights on gfx950:  y[n] = sum_k x[k] * (s[n,k/g]*q[n,k] + z[n,k/g]).
//
// Replaces the M == 1 case of mx.quantized_matmul(..., transpose=True) called at
// gbx_lm/models/quantized_linear_gba.py:195-203 (224 calls per Llama-3-8B token).
//
// Design (HBM-read bound; see DESIGN.md "gemv"):
//   * one wavefront owns whole output rows; lane l owns 16-byte (12-byte at 3 bits) "slots"
//     l, l+64, ... of the packed row, so every wave-load is one contiguous 1 KiB (768 B) burst of
//     the reference's own [N, K*bits/32] layout -- no repacking, no LDS round trip for weights,
//     non-temporal (streamed once).
//   * the activation vector is converted ONCE per workgroup into "pre-shifted" fp16 pairs and kept
//     in VGPRs for all rows of the wave.  Weights are never shifted into place: a 2/3/4-bit field
//     masked in situ inside a 16-bit half is an fp16 SUBNORMAL of value q*2^(pos-24); the matching
//     activation is pre-multiplied by 2^(E-pos), so v_dot2_f32_f16 accumulates 2^(E-24)*sum(q*x)
//     exactly (integers times bf16/fp16 values, fp32 accumulate).  Cost: ~1.1 (2/4-bit) and ~1.4
//     (3-bit) VALU lane-ops per weight.  E is a per-call block exponent so that any bf16 input fits fp16.
//   * per 32-value sub-slot the lane applies its group's scale/zero in the algebraic form
//     s*sum(q*x) + z*sum(x), then one DPP wave reduction per row.
#pragma once
#include "gbx_device.h"

namespace gbx {

struct GemvArgs {
  const void* x;            // [M, K] activations (act dtype)
  const uint32_t* qweight;  // [N, K*bits/32]
  const void* scales;       // [N, K/g] (act dtype)
  const void* zeros;        // [N, K/g]
  const void* bias;         // [N] or null
  void* y;                  // [M, N]
  int N, K;
  int log2g;                // 5,6,7
  int rows_per_wave;
};

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

// LDS geometry: slot s keeps its VPL/2 fp16 pairs at h2 index s*SSTR (16 B of padding per slot keeps the
// one-time ds_read_b128 of 64 lanes at distinct bank quads).
template <int BITS> struct Lds {
  static constexpr int PAIRS = Fmt<BITS>::VPL / 2;
  static constexpr int SSTR = PAIRS + 4;
  __host__ __device__ static constexpr size_t bytes(int K) {
    const int nslots = K / Fmt<BITS>::VPL;
    return static_cast<size_t>(nslots) * SSTR * 4 + static_cast<size_t>(nslots) * Fmt<BITS>::NSUB * 4 + 64;
  }
};

// P = number of 64-slot passes one lane makes over a row (K <= P*64*VPL); R = rows in flight per wave.
template <int BITS, int P, bool BF16>
__global__ __launch_bounds__(256) void gemv_kernel(const GemvArgs a) {
  using F = Fmt<BITS>;
  using L = Lds<BITS>;
  using WT = typename SlotWords<BITS>::type;
  constexpr int PAIRS = L::PAIRS, SSTR = L::SSTR, NSUB = F::NSUB;
  constexpr int R = (P == 1) ? 4 : (P <= 4 ? 2 : 1);
  constexpr int UPS = F::VPL / F::UNIT_VALS;                 // units per slot
  constexpr int UPT = (P * 64 * UPS + 255) / 256;            // units per thread in the prologue
  constexpr int UW = F::UNIT_VALS / 2;                       // 32-bit words of activations per unit

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int K = a.K, N = a.N;
  const int nslots = K / F::VPL;
  h2* xp_lds = reinterpret_cast<h2*>(smem);
  float* xs_lds = reinterpret_cast<float*>(smem + static_cast<size_t>(nslots) * SSTR * 4);
  unsigned* red_lds = reinterpret_cast<unsigned*>(xs_lds + nslots * NSUB);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int W = K / 32 * BITS;                    // words per packed row
  const int G = K >> a.log2g;                     // groups per row
  const int m = blockIdx.y;                       // activation row (M > 1 re-streams the weights)
  const unsigned short* __restrict__ xg = static_cast<const unsigned short*>(a.x) + static_cast<size_t>(m) * K;
  const unsigned short* __restrict__ sg = static_cast<const unsigned short*>(a.scales);
  const unsigned short* __restrict__ zg = static_cast<const unsigned short*>(a.zeros);

  // ---------------- prologue: activations -> pre-shifted fp16 pairs + per-sub-slot sums ----------
  const int nunits = K / F::UNIT_VALS;
  unsigned xr[UPT][UW];
  unsigned amax = 0;
#pragma unroll
  for (int j = 0; j < UPT; ++j) {
    const int u = tid + 256 * j;
    if (u < nunits) {
      const unsigned* src = reinterpret_cast<const unsigned*>(xg + static_cast<size_t>(u) * F::UNIT_VALS);
#pragma unroll
      for (int q = 0; q < UW; q += 4) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(src + q);
        xr[j][q] = v.x; xr[j][q + 1] = v.y; xr[j][q + 2] = v.z; xr[j][q + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int q = 0; q < UW; ++q) xr[j][q] = 0;
    }
#pragma unroll
    for (int q = 0; q < UW; ++q) {
      amax = max(amax, __float_as_uint(act_lo<BF16>(xr[j][q])) & 0x7FFFFFFFu);
      amax = max(amax, __float_as_uint(act_hi<BF16>(xr[j][q])) & 0x7FFFFFFFu);
    }
  }
  amax = wave_max_u32(amax);
  if (lane == 0) red_lds[wave] = amax;
  __syncthreads();
  amax = max(max(red_lds[0], red_lds[1]), max(red_lds[2], red_lds[3]));
  // block exponent: largest |x| lands in [2^14, 2^15) as fp16
  int E = 141 - static_cast<int>(amax >> 23);
  E = min(max(E, -100), 110);
  const float cfix = exp2i(24 - E);               // undoes 2^(E-24) after the integer dot

#pragma unroll
  for (int j = 0; j < UPT; ++j) {
    const int u = tid + 256 * j;
    float xf[F::UNIT_VALS];
#pragma unroll
    for (int q = 0; q < UW; ++q) { xf[2 * q] = act_lo<BF16>(xr[j][q]); xf[2 * q + 1] = act_hi<BF16>(xr[j][q]); }
    float usum = 0.f;
#pragma unroll
    for (int q = 0; q < F::UNIT_VALS; ++q) usum += xf[q];
    // every lane takes part in the DPP sums (inactive units contribute 0)
    float ssum;
    if constexpr (BITS == 2) ssum = pair_sum(usum);
    else if constexpr (BITS == 4) ssum = quad_sum(usum);
    else ssum = usum;
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
      constexpr int UPSUB = 32 / F::UNIT_VALS == 0 ? 1 : 32 / F::UNIT_VALS;   // units per 32-value sub-slot
      if ((u % UPSUB) == 0) xs_lds[u / UPSUB] = ssum;
    }
  }
  __syncthreads();

  // ---------------- lane-resident activations ----------------------------------------------------
  h2 xv[P][PAIRS];
  float xs[P][NSUB];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int slot = p * 64 + lane;
    if (slot < nslots) {
      const u32x4* src = reinterpret_cast<const u32x4*>(xp_lds + static_cast<size_t>(slot) * SSTR);
#pragma unroll
      for (int q = 0; q < PAIRS / 4; ++q) {
        const u32x4 v = src[q];
        xv[p][4 * q + 0] = as_h2(v.x);
        xv[p][4 * q + 1] = as_h2(v.y);
        xv[p][4 * q + 2] = as_h2(v.z);
        xv[p][4 * q + 3] = as_h2(v.w);
      }
#pragma unroll
      for (int s = 0; s < NSUB; ++s) xs[p][s] = xs_lds[slot * NSUB + s];
    } else {
#pragma unroll
      for (int q = 0; q < PAIRS; ++q) xv[p][q] = h2{0, 0};
#pragma unroll
      for (int s = 0; s < NSUB; ++s) xs[p][s] = 0.f;
    }
  }

  // ---------------- stream the rows ---------------------------------------------------------------
  const int rpw = a.rows_per_wave;
  const int row_begin = (blockIdx.x * 4 + wave) * rpw;
  const int row_end = min(row_begin + rpw, N);
  const int gshift = a.log2g - 5;                 // 32-value sub-slot index -> group index
  const bool split = (NSUB == 2) && (gshift == 0);

  for (int r0 = row_begin; r0 < row_end; r0 += R) {
    WT w[R][P];
    unsigned short sraw[R][P][NSUB], zraw[R][P][NSUB];
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int row = min(r0 + i, N - 1);
      const uint32_t* wrow = a.qweight + static_cast<size_t>(row) * W;
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int slot = p * 64 + lane;
        if (slot < nslots) {
          w[i][p] = __builtin_nontemporal_load(reinterpret_cast<const WT*>(wrow + static_cast<size_t>(slot) * F::SLOT_WORDS));
          const int ss = slot * NSUB;
          const size_t gi = static_cast<size_t>(row) * G + (ss >> gshift);
          sraw[i][p][0] = sg[gi];
          zraw[i][p][0] = zg[gi];
          if constexpr (NSUB == 2) {
            if (split) { sraw[i][p][1] = sg[gi + 1]; zraw[i][p][1] = zg[gi + 1]; }
            else { sraw[i][p][1] = sraw[i][p][0]; zraw[i][p][1] = zraw[i][p][0]; }
          }
        } else {
          w[i][p] = WT{};
#pragma unroll
          for (int s = 0; s < NSUB; ++s) { sraw[i][p][s] = 0; zraw[i][p][s] = 0; }
        }
      }
    }
    float acc[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
      acc[i] = 0.f;
#pragma unroll
      for (int p = 0; p < P; ++p) {
        float d[NSUB];
        dot_slot<BITS>(w[i][p], xv[p], d);
#pragma unroll
        for (int s = 0; s < NSUB; ++s) {
          const float sc = act_to_f32<BF16>(sraw[i][p][s]) * cfix;
          const float ze = act_to_f32<BF16>(zraw[i][p][s]);
          acc[i] = fmaf(sc, d[s], acc[i]);
          acc[i] = fmaf(ze, xs[p][s], acc[i]);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const float tot = wave_sum(acc[i]);
      const int row = r0 + i;
      if (lane == 0 && row < row_end) {
        float v = tot;
        if (a.bias) v += act_to_f32<BF16>(static_cast<const unsigned short*>(a.bias)[row]);
        static_cast<unsigned short*>(a.y)[static_cast<size_t>(m) * N + row] = f32_to_act<BF16>(v);
      }
    }
  }
}

// Host-side launcher for one bit width (defined per TU so the three widths compile in parallel).
template <int BITS>
hipError_t launch_gemv(const GemvArgs& a, int M, bool bf16, hipStream_t stream);

template <int BITS, int P>
static hipError_t launch_gemv_p(const GemvArgs& a, int M, bool bf16, hipStream_t stream) {
  GemvArgs args = a;
  // rows per wave: aim for >= ~2 workgroups per CU (256 CUs) while keeping a few rows per wave so the
  // lane-resident activations are amortised.
  constexpr int R = (P == 1) ? 4 : (P <= 4 ? 2 : 1);
  int rpw = R;
  while (rpw < 64 && (a.N + 4 * rpw * 2 - 1) / (4 * rpw * 2) >= 1024) rpw *= 2;
  args.rows_per_wave = rpw;
  const int blocks = (a.N + 4 * rpw - 1) / (4 * rpw);
  const size_t shmem = Lds<BITS>::bytes(a.K);
  dim3 grid(blocks, M), block(256);
  if (bf16) hipLaunchKernelGGL((gemv_kernel<BITS, P, true>), grid, block, shmem, stream, args);
  else hipLaunchKernelGGL((gemv_kernel<BITS, P, false>), grid, block, shmem, stream, args);
  return hipGetLastError();
}

// max K the lane-resident fast path covers for a bit width (beyond -> generic kernel)
template <int BITS> constexpr int gemv_max_k() { return (BITS == 2 ? 4 : 8) * 64 * Fmt<BITS>::VPL; }

template <int BITS>
static hipError_t dispatch_gemv(const GemvArgs& a, int M, bool bf16, hipStream_t s) {
  const int nslots = a.K / Fmt<BITS>::VPL;
  if (nslots <= 64) return launch_gemv_p<BITS, 1>(a, M, bf16, s);
  if (nslots <= 128) return launch_gemv_p<BITS, 2>(a, M, bf16, s);
  if (nslots <= 256) return launch_gemv_p<BITS, 4>(a, M, bf16, s);
  if constexpr (BITS != 2) {
    if (nslots <= 512) return launch_gemv_p<BITS, 8>(a, M, bf16, s);
  }
  return hipErrorInvalidValue;
}

#define GBX_DEFINE_LAUNCH_GEMV(BITS)                                                              \
  template <> hipError_t launch_gemv<BITS>(const GemvArgs& a, int M, bool bf16, hipStream_t s) { \
    return dispatch_gemv<BITS>(a, M, bf16, s);                                                    \
  }

}  // namespace gbx
