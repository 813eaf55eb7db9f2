// Batched GBA QuantizedLinear (M > 1): y[M,N] = x[M,K] . dequant(W[N,K])^T on the matrix cores (gfx950 MFMA).
//
// Replaces mx.quantized_matmul(..., transpose=True) for prefill / batched decode
// (gbx_lm/models/quantized_linear_gba.py:195-203 when x is [B, L, K] with B*L > 1; e.g. the chunked prefill at
// gbx_lm/utils.py:312-319 and the batched eval at gbx_lm/evaluate.py:144-185).
//
// Here the dequantised tile is a genuine dense contraction, so it runs on v_mfma_f32_32x32x16_{bf16,f16}:
//   * B operand: lane (n = l&31, h = l>>5) owns 8 consecutive k of weight row n per k-step -- ONE packed word at 4 bits,
//     half a word at 2 bits, 24 bits at 3 bits -- so a fragment is built from the lane's own registers: fields -> fp32
//     (v_cvt_f32_ubyte*), w = s*q + z in fp32, one rounding to the activation dtype (v_cvt_pk).  That is the arithmetic
//     of a dequantise-then-matmul (what MLX's qmm does too); the exact-integer algebraic form is kept for the M = 1 GEMV.
//   * A operand: activations staged per 128-wide K chunk in LDS (natural order, rows padded by 16 B so ds_read_b128 is
//     conflict-free), double-buffered; next chunk's activations and packed words are in flight during the MFMAs.
//   * wave tile (32*MT) x (32*NT), workgroup = WM x (4/WM) waves.
// MFMA-bound for large M (2*M*N*K flop), HBM-bound for small M (weights read once per column tile).
#include "mfma_common.h"
#include "kernels.h"

namespace gbx {

// WM waves along M, 4/WM along N; each wave owns a (32*MT) x (32*NT) output tile.
template <int BITS, int WM, int MT, int NT, bool BF16>
__global__ __launch_bounds__(256) void qgemm_mfma_kernel(const MfmaArgs a) {
  constexpr int WN = 4 / WM;
  constexpr int BM = 32 * MT * WM, BN = 32 * NT * WN;
  constexpr int WPC = KC * BITS / 32;               // packed words per row per chunk (8 / 12 / 16)
  constexpr int XPT = BM * KC * 2 / 16 / 256;       // 16-byte activation pieces per thread per chunk
  static_assert(XPT >= 1, "tile too small for 256 threads");
  __shared__ __attribute__((aligned(16))) unsigned char xs_lds[2][BM * XROW];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;
  const int K = a.K, N = a.N, M = a.M;
  const int W = K / 32 * BITS, G = K >> a.log2g;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int gpc = KC >> a.log2g;                     // groups per chunk: 1, 2 or 4
  const int kshift = a.log2g - 4;                    // k-step index -> group-in-chunk index

  int nrow[NT];
  const uint32_t* wrow[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    nrow[j] = min(n0 + (wn * NT + j) * 32 + r, N - 1);
    wrow[j] = a.qweight + static_cast<size_t>(nrow[j]) * W;
  }

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x16{0};

  // two register sets for the packed words / scales, used alternately (the chunk loop is unrolled by two so that no
  // set is ever copied)
  unsigned wA[NT][WPC], wB[NT][WPC];
  unsigned short sA[NT][4], zA[NT][4], sB[NT][4], zB[NT][4];
  u32x4 xstage[XPT];

  auto load_w = [&](unsigned (&wv)[NT][WPC], unsigned short (&sv)[NT][4], unsigned short (&zv)[NT][4], int kc) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const uint32_t* src = wrow[j] + (kc / 32) * BITS;
#pragma unroll
      for (int q = 0; q < WPC; q += 4) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(src + q);
        wv[j][q] = v.x; wv[j][q + 1] = v.y; wv[j][q + 2] = v.z; wv[j][q + 3] = v.w;
      }
      const size_t g0 = static_cast<size_t>(nrow[j]) * G + (kc >> a.log2g);
#pragma unroll
      for (int u = 0; u < 4; ++u) {                  // unconditional (clamped) so the waitcnt pass can count
        const size_t gi = g0 + min(u, gpc - 1);
        sv[j][u] = a.scales[gi];
        zv[j][u] = a.zeros[gi];
      }
    }
  };
  // piece p of this thread: row = (tid + 256*p) / 16, 16-byte column = (tid + 256*p) % 16
  auto load_x = [&](int kc) {
#pragma unroll
    for (int p = 0; p < XPT; ++p) {
      const int idx = tid + 256 * p, row = idx >> 4, col = idx & 15;
      const int mg = min(m0 + row, M - 1);
      xstage[p] = *reinterpret_cast<const u32x4*>(a.x + static_cast<size_t>(mg) * K + kc + col * 8);
    }
  };
  auto store_x = [&](int buf) {
#pragma unroll
    for (int p = 0; p < XPT; ++p) {
      const int idx = tid + 256 * p, row = idx >> 4, col = idx & 15;
      *reinterpret_cast<u32x4*>(xs_lds[buf] + row * XROW + col * 16) = xstage[p];
    }
  };
  // all MFMAs of one chunk: B fragments dequantised from registers, A fragments from LDS buffer `buf`
  auto compute = [&](const unsigned (&wv)[NT][WPC], const unsigned short (&sv)[NT][4], const unsigned short (&zv)[NT][4], int buf) {
    float sf[NT][4], zf[NT][4];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int u = 0; u < 4; ++u) { sf[j][u] = act_to_f32<BF16>(sv[j][u]); zf[j][u] = act_to_f32<BF16>(zv[j][u]); }
#pragma unroll
    for (int t = 0; t < KC / 16; ++t) {
      u32x4 bf[NT];
      const int gi = t >> kshift;                    // group of this k-step inside the chunk (wave-uniform)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        float q[8];
        fields8<BITS>(wv[j], t, h, q);
        const float s = gi == 0 ? sf[j][0] : gi == 1 ? sf[j][1] : gi == 2 ? sf[j][2] : sf[j][3];
        const float z = gi == 0 ? zf[j][0] : gi == 1 ? zf[j][1] : gi == 2 ? zf[j][2] : zf[j][3];
        bf[j].x = pack2<BF16>(fmaf(s, q[0], z), fmaf(s, q[1], z));
        bf[j].y = pack2<BF16>(fmaf(s, q[2], z), fmaf(s, q[3], z));
        bf[j].z = pack2<BF16>(fmaf(s, q[4], z), fmaf(s, q[5], z));
        bf[j].w = pack2<BF16>(fmaf(s, q[6], z), fmaf(s, q[7], z));
      }
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const u32x4 af = *reinterpret_cast<const u32x4*>(xs_lds[buf] + ((wm * MT + i) * 32 + r) * XROW + (2 * t + h) * 16);
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = mfma<BF16>(af, bf[j], acc[i][j]);
      }
    }
  };

  load_x(0);
  load_w(wA, sA, zA, 0);
  store_x(0);
  __syncthreads();

  const int nchunks = K / KC;
  for (int c = 0; c < nchunks; c += 2) {
    // even chunk: set A, LDS buffer 0; prefetch chunk c+1 into set B / buffer 1
    int kn = min((c + 1) * KC, K - KC);              // clamped: a last prefetch re-reads the final chunk harmlessly
    load_x(kn);
    load_w(wB, sB, zB, kn);
    compute(wA, sA, zA, 0);
    store_x(1);
    __syncthreads();
    if (c + 1 >= nchunks) break;
    // odd chunk: set B, LDS buffer 1; prefetch chunk c+2 into set A / buffer 0
    kn = min((c + 2) * KC, K - KC);
    load_x(kn);
    load_w(wA, sA, zA, kn);
    compute(wB, sB, zB, 1);
    store_x(0);
    __syncthreads();
  }

  // ---- epilogue: one rounding to the activation dtype (+ bias as a second rounding, like the reference's x + bias) ----
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + (wn * NT + j) * 32 + r;
    if (n >= N) continue;
    const float b = a.bias ? act_to_f32<BF16>(a.bias[n]) : 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + (wm * MT + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;   // C/D layout of the 32x32 tile
        if (m < M) {
          float v = acc[i][j][e];
          if (a.bias) v = act_to_f32<BF16>(f32_to_act<BF16>(v)) + b;
          a.y[static_cast<size_t>(m) * N + n] = f32_to_act<BF16>(v);
        }
      }
    }
  }
}

template <int BITS, int WM, int MT, int NT>
static void launch_cfg(const MfmaArgs& a, bool bf16, hipStream_t s) {
  constexpr int BM = 32 * MT * WM, BN = 32 * NT * (4 / WM);
  dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM), block(256);
  if (bf16) hipLaunchKernelGGL((qgemm_mfma_kernel<BITS, WM, MT, NT, true>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((qgemm_mfma_kernel<BITS, WM, MT, NT, false>), grid, block, 0, s, a);
}

template <int BITS>
static void launch_bits(const MfmaArgs& a, bool bf16, hipStream_t s) {
  if (a.M <= 32) launch_cfg<BITS, 1, 1, 1>(a, bf16, s);        // 32 x 128 tile: batched decode, weights streamed once
  else if (a.M <= 64) launch_cfg<BITS, 1, 2, 1>(a, bf16, s);   // 64 x 128
  else launch_cfg<BITS, 2, 2, 2>(a, bf16, s);                  // 128 x 128: prefill
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
