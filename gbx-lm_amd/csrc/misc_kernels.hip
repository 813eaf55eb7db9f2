// Support kernels of the GBA QuantizedLinear path: kernel-side unpack dump, dequantize
// (mx.dequantize role, gbx_lm/tuner/lora.py:62-68) and a shape-generic qlinear used for shapes
// the streaming kernels do not cover (K not a multiple of 128, very long rows).
#include "gbx_device.h"
#include "kernels.h"

namespace gbx {

// ---- unpack: q[n,k] --------------------------------------------------------------------------
template <int BITS>
__global__ __launch_bounds__(256) void unpack_kernel(const uint32_t* __restrict__ qw, int32_t* __restrict__ q,
                                                     int N, int K) {
  const int W = K / 32 * BITS;
  const size_t total = static_cast<size_t>(N) * K;
  for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < total;
       i += static_cast<size_t>(gridDim.x) * blockDim.x) {
    const int n = static_cast<int>(i / K), k = static_cast<int>(i % K);
    q[i] = static_cast<int32_t>(field_of_row<BITS>(qw + static_cast<size_t>(n) * W, k));
  }
}

hipError_t launch_unpack(const uint32_t* qw, int32_t* q, int N, int K, int bits, hipStream_t s) {
  const size_t total = static_cast<size_t>(N) * K;
  const int blocks = static_cast<int>(std::min<size_t>((total + 255) / 256, 4096));
  switch (bits) {
    case 2: hipLaunchKernelGGL(unpack_kernel<2>, dim3(blocks), dim3(256), 0, s, qw, q, N, K); break;
    case 3: hipLaunchKernelGGL(unpack_kernel<3>, dim3(blocks), dim3(256), 0, s, qw, q, N, K); break;
    case 4: hipLaunchKernelGGL(unpack_kernel<4>, dim3(blocks), dim3(256), 0, s, qw, q, N, K); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// ---- dequantize: w = s*q + z -----------------------------------------------------------------
// fp32 output is bit-exact against the oracle: one multiply and one add, never contracted to an fma.
template <int BITS, bool SZ_BF16, int OUT>
__global__ __launch_bounds__(256) void dequant_kernel(const uint32_t* __restrict__ qw, const unsigned short* __restrict__ sc,
                                                      const unsigned short* __restrict__ ze, void* __restrict__ w,
                                                      int N, int K, int log2g) {
  const int W = K / 32 * BITS, G = K >> log2g;
  const size_t total = static_cast<size_t>(N) * K;
  for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < total;
       i += static_cast<size_t>(gridDim.x) * blockDim.x) {
    const int n = static_cast<int>(i / K), k = static_cast<int>(i % K);
    const float q = static_cast<float>(field_of_row<BITS>(qw + static_cast<size_t>(n) * W, k));
    const float s = act_to_f32<SZ_BF16>(sc[static_cast<size_t>(n) * G + (k >> log2g)]);
    const float z = act_to_f32<SZ_BF16>(ze[static_cast<size_t>(n) * G + (k >> log2g)]);
    const float v = __fadd_rn(__fmul_rn(q, s), z);
    if constexpr (OUT == 2) static_cast<float*>(w)[i] = v;
    else if constexpr (OUT == 1) static_cast<unsigned short*>(w)[i] = f32_to_act<true>(v);
    else static_cast<unsigned short*>(w)[i] = f32_to_act<false>(v);
  }
}

template <int BITS, bool SZ_BF16>
static void launch_dequant_o(const uint32_t* qw, const void* sc, const void* ze, void* w, int N, int K, int log2g,
                             int out, int blocks, hipStream_t s) {
  auto S = static_cast<const unsigned short*>(sc);
  auto Z = static_cast<const unsigned short*>(ze);
  if (out == 2) hipLaunchKernelGGL((dequant_kernel<BITS, SZ_BF16, 2>), dim3(blocks), dim3(256), 0, s, qw, S, Z, w, N, K, log2g);
  else if (out == 1) hipLaunchKernelGGL((dequant_kernel<BITS, SZ_BF16, 1>), dim3(blocks), dim3(256), 0, s, qw, S, Z, w, N, K, log2g);
  else hipLaunchKernelGGL((dequant_kernel<BITS, SZ_BF16, 0>), dim3(blocks), dim3(256), 0, s, qw, S, Z, w, N, K, log2g);
}

hipError_t launch_dequant(const uint32_t* qw, const void* sc, const void* ze, void* w, int N, int K, int bits,
                          int log2g, bool sz_bf16, int out, hipStream_t s) {
  const size_t total = static_cast<size_t>(N) * K;
  const int blocks = static_cast<int>(std::min<size_t>((total + 255) / 256, 8192));
#define GBX_DQ(B)                                                                         \
  if (sz_bf16) launch_dequant_o<B, true>(qw, sc, ze, w, N, K, log2g, out, blocks, s);   \
  else launch_dequant_o<B, false>(qw, sc, ze, w, N, K, log2g, out, blocks, s);
  switch (bits) {
    case 2: GBX_DQ(2) break;
    case 3: GBX_DQ(3) break;
    case 4: GBX_DQ(4) break;
    default: return hipErrorInvalidValue;
  }
#undef GBX_DQ
  return hipGetLastError();
}

// ---- dequantize, one 32-value unit (BITS packed words -> 64 bytes of 16-bit weights) per thread ------------------------------
// Same arithmetic and rounding as dequant_kernel (q * s, + z, one rounding to the activation dtype); used by the long-prompt
// path, which dequantises a whole matrix once per call (mfma_gemm.hip: launch_qgemm_dense): 16-byte loads and stores instead
// of one element per thread (4096 x 4096: 31 -> ~10 us).
template <int BITS, bool BF16>
__global__ __launch_bounds__(256) void dequant_units_kernel(const uint32_t* __restrict__ qw, const unsigned short* __restrict__ sc,
                                                            const unsigned short* __restrict__ ze, unsigned short* __restrict__ w,
                                                            long long units, int upr, int log2g, int blk, int which) {
  for (long long u = blockIdx.x * 256LL + threadIdx.x; u < units; u += static_cast<long long>(gridDim.x) * 256) {
    const long long n = u / upr;
    const int ur = static_cast<int>(u - n * upr);
    // blk > 0: two matrices interleaved in blocks of blk rows (this one is matrix `which`): gate | up of a fused SwiGLU GEMM
    const long long no = blk ? (n / blk) * (2 * blk) + which * blk + n % blk : n;
    const uint32_t* src = qw + u * BITS;
    unsigned wv[BITS];
#pragma unroll
    for (int i = 0; i < BITS; ++i) wv[i] = src[i];
    const int g = (ur * 32) >> log2g, G = (upr * 32) >> log2g;
    const float s = act_to_f32<BF16>(sc[n * G + g]), z = act_to_f32<BF16>(ze[n * G + g]);
    unsigned o[16];
#pragma unroll
    for (int j = 0; j < 32; j += 2) {
      unsigned short h[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int bit = (j + t) * BITS, wi = bit >> 5, sh = bit & 31;
        unsigned q = (wv[wi] >> sh) & ((1u << BITS) - 1u);
        if (BITS == 3 && sh + BITS > 32) q |= (wv[wi + 1 < BITS ? wi + 1 : wi] << (32 - sh)) & 7u;
        h[t] = f32_to_act<BF16>(__fadd_rn(__fmul_rn(static_cast<float>(q), s), z));
      }
      o[j >> 1] = static_cast<unsigned>(h[0]) | (static_cast<unsigned>(h[1]) << 16);
    }
    u32x4* dst = reinterpret_cast<u32x4*>(w + (no * upr + ur) * 32);
#pragma unroll
    for (int i = 0; i < 4; ++i) dst[i] = u32x4{o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]};
  }
}

hipError_t launch_dequant_units(const uint32_t* qw, const void* sc, const void* ze, void* w, int N, int K, int bits, int log2g,
                                bool bf16, hipStream_t s, int blk, int which) {
  const long long units = static_cast<long long>(N) * (K / 32);
  const int blocks = static_cast<int>(std::min<long long>((units + 255) / 256, 16384));
  auto S = static_cast<const unsigned short*>(sc);
  auto Z = static_cast<const unsigned short*>(ze);
  auto Wd = static_cast<unsigned short*>(w);
#define GBX_DU(B)                                                                                                        \
  if (bf16) hipLaunchKernelGGL((dequant_units_kernel<B, true>), dim3(blocks), dim3(256), 0, s, qw, S, Z, Wd, units, K / 32, log2g, blk, which); \
  else hipLaunchKernelGGL((dequant_units_kernel<B, false>), dim3(blocks), dim3(256), 0, s, qw, S, Z, Wd, units, K / 32, log2g, blk, which);
  switch (bits) {
    case 2: GBX_DU(2) break;
    case 3: GBX_DU(3) break;
    case 4: GBX_DU(4) break;
    default: return hipErrorInvalidValue;
  }
#undef GBX_DU
  return hipGetLastError();
}

// ---- packed scale|zero companion (load time) ---------------------------------------------------
// out[p][g][i]: i-th row of pair p; pair = rows (2p, 2p+1) of one tensor, or row p of tensor 0 / tensor 1.
__global__ __launch_bounds__(256) void pack_sz_kernel(const unsigned short* __restrict__ s0, const unsigned short* __restrict__ z0,
                                                      const unsigned short* __restrict__ s1, const unsigned short* __restrict__ z1,
                                                      uint32_t* __restrict__ out, int npairs, int G) {
  const size_t total = static_cast<size_t>(npairs) * G * 2;
  for (size_t t = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; t < total;
       t += static_cast<size_t>(gridDim.x) * blockDim.x) {
    const int i = static_cast<int>(t & 1);
    const size_t pg = t >> 1;
    const int g = static_cast<int>(pg % G);
    const size_t p = pg / G;
    const unsigned short* sc = s1 ? (i ? s1 : s0) : s0;
    const unsigned short* ze = s1 ? (i ? z1 : z0) : z0;
    const size_t row = s1 ? p : 2 * p + i;
    out[t] = static_cast<uint32_t>(sc[row * G + g]) | (static_cast<uint32_t>(ze[row * G + g]) << 16);
  }
}

hipError_t launch_pack_sz(const void* s0, const void* z0, const void* s1, const void* z1, uint32_t* out, int N, int G,
                          hipStream_t s) {
  const int npairs = s1 ? N : N / 2;
  const size_t total = static_cast<size_t>(npairs) * G * 2;
  const int blocks = static_cast<int>(std::min<size_t>((total + 255) / 256, 4096));
  hipLaunchKernelGGL(pack_sz_kernel, dim3(blocks), dim3(256), 0, s, static_cast<const unsigned short*>(s0),
                     static_cast<const unsigned short*>(z0), static_cast<const unsigned short*>(s1),
                     static_cast<const unsigned short*>(z1), out, npairs, G);
  return hipGetLastError();
}

// ---- 3-bit rows as (2-bit plane, 1-bit plane) per 32 values (load time; round 6) ----------------------------------------------------------
// The loader's 3-bit row is a bit stream: field k at bit 3 k, fields straddle words -- the reason every 3-bit kernel is vector-ALU bound.
// Per 32 values (three words in, three words out, SAME place in the row): words 0 / 1 = the low two bits of values 0..15 / 16..31 as 2-bit
// fields in value order (exactly a 2-bit row's words), word 2 = the high bits, ordered so that (h >> i) & 0x11111111 is the FP4 operand
// dword i of the matrix-core kernels: bit 4 n + 0 / 1 / 2 / 3 = high bit of value 2 n / 2 n + 1 / 16 + 2 n / 17 + 2 n.  q = lo + 4 hi.
__global__ __launch_bounds__(256) void repack_planes3_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t nblocks) {
  for (size_t b = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; b < nblocks; b += static_cast<size_t>(gridDim.x) * blockDim.x) {
    const unsigned long long w01 = static_cast<unsigned long long>(in[3 * b]) | (static_cast<unsigned long long>(in[3 * b + 1]) << 32);
    const unsigned long long w12 = static_cast<unsigned long long>(in[3 * b + 1]) | (static_cast<unsigned long long>(in[3 * b + 2]) << 32);
    uint32_t lo0 = 0, lo1 = 0, hi = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const int bit = 3 * i;
      const unsigned v = bit < 32 ? static_cast<unsigned>(w01 >> bit) & 7u : static_cast<unsigned>(w12 >> (bit - 32)) & 7u;
      if (i < 16) lo0 |= (v & 3u) << (2 * i); else lo1 |= (v & 3u) << (2 * (i - 16));
      const int n = (i & 15) >> 1, slot = (i & 1) + 2 * (i >> 4);
      hi |= (v >> 2) << (4 * n + slot);
    }
    out[3 * b] = lo0; out[3 * b + 1] = lo1; out[3 * b + 2] = hi;
  }
}

hipError_t launch_repack_planes3(const uint32_t* in, uint32_t* out, size_t nblocks, hipStream_t s) {
  const int blocks = static_cast<int>(std::min<size_t>((nblocks + 255) / 256, 8192));
  hipLaunchKernelGGL(repack_planes3_kernel, dim3(blocks), dim3(256), 0, s, in, out, nblocks);
  return hipGetLastError();
}

// ---- shape-generic qlinear (any K % 32 == 0, any M): one wave per (row n, activation row m) ----
// Lanes stride over 32-value chunks (always a whole number of words); fp32 math throughout.
template <int BITS, bool BF16>
__global__ __launch_bounds__(256) void qlinear_generic_kernel(const unsigned short* __restrict__ x, const uint32_t* __restrict__ qw,
                                                              const unsigned short* __restrict__ sc, const unsigned short* __restrict__ ze,
                                                              const unsigned short* __restrict__ bias, unsigned short* __restrict__ y,
                                                              int N, int K, int log2g) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.x * 4 + wave;
  const int m = blockIdx.y;
  if (n >= N) return;
  const int W = K / 32 * BITS, G = K >> log2g;
  const uint32_t* row = qw + static_cast<size_t>(n) * W;
  const unsigned short* xr = x + static_cast<size_t>(m) * K;
  float acc = 0.f;
  for (int c = lane; c < K / 32; c += 64) {
    uint32_t wv[BITS + 1];
#pragma unroll
    for (int i = 0; i < BITS; ++i) wv[i] = row[c * BITS + i];
    wv[BITS] = 0;
    float qx = 0.f, sx = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const float xv = act_to_f32<BF16>(xr[c * 32 + j]);
      qx = fmaf(static_cast<float>(field_of_row<BITS>(wv, j)), xv, qx);
      sx += xv;
    }
    const int g = (c * 32) >> log2g;
    acc = fmaf(act_to_f32<BF16>(sc[static_cast<size_t>(n) * G + g]), qx, acc);
    acc = fmaf(act_to_f32<BF16>(ze[static_cast<size_t>(n) * G + g]), sx, acc);
  }
  const float tot = wave_sum(acc);
  if (lane == 0) {
    float v = tot;
    if (bias) v += act_to_f32<BF16>(bias[n]);
    y[static_cast<size_t>(m) * N + n] = f32_to_act<BF16>(v);
  }
}

hipError_t launch_qlinear_generic(const void* x, const uint32_t* qw, const void* sc, const void* ze, const void* bias,
                                  void* y, int M, int N, int K, int bits, int log2g, bool bf16, hipStream_t s) {
  dim3 grid((N + 3) / 4, M), block(256);
  auto X = static_cast<const unsigned short*>(x);
  auto S = static_cast<const unsigned short*>(sc);
  auto Z = static_cast<const unsigned short*>(ze);
  auto B = static_cast<const unsigned short*>(bias);
  auto Y = static_cast<unsigned short*>(y);
#define GBX_GEN(BITS)                                                                                              \
  if (bf16) hipLaunchKernelGGL((qlinear_generic_kernel<BITS, true>), grid, block, 0, s, X, qw, S, Z, B, Y, N, K, log2g); \
  else hipLaunchKernelGGL((qlinear_generic_kernel<BITS, false>), grid, block, 0, s, X, qw, S, Z, B, Y, N, K, log2g);
  switch (bits) {
    case 2: GBX_GEN(2) break;
    case 3: GBX_GEN(3) break;
    case 4: GBX_GEN(4) break;
    default: return hipErrorInvalidValue;
  }
#undef GBX_GEN
  return hipGetLastError();
}

}  // namespace gbx
