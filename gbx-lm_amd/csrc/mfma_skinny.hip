// Batched decode / short prompts (2 <= M <= 64): y[M,N] = x[M,K] . dequant(W[N,K])^T, weight-stream bound.
//
// Same arithmetic as mfma_gemm.hip (fields -> fp32, w = s*q + z, one rounding to the activation dtype, MFMA 32x32x16,
// fp32 accumulate), different parallelisation: at these M the packed weights are read once and the kernel is bound by
// how many lanes stream them, so a workgroup owns ONLY 32 weight rows and splits K over its KW waves (each wave owns a
// contiguous run of 128-value chunks); the KW partial 32x32 tiles meet in LDS.  N/32 workgroups x KW waves keep the
// whole chip streaming even for N = 4096 (the earlier 128-column tiling put 32 workgroups on 256 CUs).
//   * B operand: lane (n = l&31, h = l>>5) dequantises 8 consecutive k of weight row n from its own packed words.
//   * A operand: lane (m = l&31, h) loads its 8 consecutive k of activation row m straight from global memory (x is
//     at most 64 rows: L2-resident), one chunk ahead in registers.
// Replaces mx.quantized_matmul for x of shape [B, 1, K] with 2 <= B <= 64 (gbx_lm/models/quantized_linear_gba.py:195-203).
#include "mfma_common.h"
#include "gbx_env.h"
#include "kernels.h"

namespace gbx {

// NCH = chunks of packed words a wave holds in registers at once: at these M nothing hides memory latency but bytes in
// flight, so a wave requests a whole window of NCH chunks (its entire K range when K <= 4096) before the first MFMA.
// What bounds the kernel otherwise is the texture-address unit: every load instruction costs one cycle per distinct
// cache line it touches, so (a) a window's scales / zeros come as ONE wide load each when a chunk is one group
// (G128; the four 2-byte loads per chunk and tensor of the generic path were half of all line requests), and (b) the
// activation chunk is fetched coalesced (4 rows x 256 B per instruction) and turned into A fragments through a
// wave-private LDS slab instead of 32-line gathers straight from global memory.
// TN = weight rows per workgroup = MFMA tile edge: 32 (v_mfma 32x32x16) while N/32 workgroups fill the chip, else 16
// (v_mfma 16x16x32: twice the workgroups for the narrow projections, N = 4096 -> 256 x KW waves).
template <int BITS, int TN, int MT, int KW, int NCH, bool G128, bool BF16>
__global__ __launch_bounds__(KW * 64) void qgemm_skinny_kernel(const MfmaArgs a) {
  constexpr int WPC = KC * BITS / 32;               // packed words per row per chunk (8 / 12 / 16)
  constexpr int KSTEP = TN == 32 ? 16 : 32;         // k-values per MFMA
  constexpr int TS = KC / KSTEP;                    // k-steps per chunk
  constexpr int ROWS = MT * TN;                     // activation rows of the workgroup
  constexpr int AE = TN == 32 ? 16 : 4;             // accumulator elements per lane and tile
  constexpr int XP = ROWS * 16 / 64;                // 16-byte activation pieces per lane per chunk
  constexpr int SLAB = ROWS * XROW;                 // bytes of one wave's activation slab
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // KW slabs; later float red[KW][MT][16][64]
  float* red = reinterpret_cast<float*>(smem);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & (TN - 1), h = lane / TN;     // weight row / activation row inside the tile; k-octet inside a step
  const int K = a.K, N = a.N, M = a.M;
  const int W = K / 32 * BITS, G = K >> a.log2g;
  const int n0 = blockIdx.x * TN, m0 = blockIdx.y * ROWS;
  const int gpc = KC >> a.log2g;                    // groups per chunk: 1, 2 or 4
  const int nchunks = K / KC;                       // >= NCH (launcher)
  const int c0 = wave * nchunks / KW, c1 = (wave + 1) * nchunks / KW;   // this wave's chunks
  unsigned char* slab = smem + wave * SLAB;

  // segment of this tile (wave-uniform): a tile never straddles two matrices (row counts are multiples of the tile)
  const uint32_t* qw = a.qweight;
  const unsigned short* sc = a.scales;
  const unsigned short* ze = a.zeros;
  const unsigned short* bs = a.bias;
  unsigned short* yo = a.y;
  int nl0 = n0, NL = N;                             // tile start and row count inside the segment
  if (a.nseg > 1) {
    const int sg = (n0 >= a.n_seg[0]) + (n0 >= a.n_seg[0] + a.n_seg[1]);
    NL = a.n_seg[sg];
    if (sg > 0) {
      nl0 = n0 - a.n_seg[0] - (sg > 1 ? a.n_seg[1] : 0);
      qw = a.qweight_x[sg - 1]; sc = a.scales_x[sg - 1]; ze = a.zeros_x[sg - 1]; bs = a.bias_x[sg - 1]; yo = a.y_x[sg - 1];
    }
  }
  const int nrow = min(nl0 + r, NL - 1);
  const uint32_t* wrow = qw + static_cast<size_t>(nrow) * W;
  const unsigned short* srow = sc + static_cast<size_t>(nrow) * G;
  const unsigned short* zrow = ze + static_cast<size_t>(nrow) * G;
  // coalesced activation fetch: piece p = it*64 + lane -> row p/16, 16-byte column p%16
  const unsigned short* xsrc[XP];
  int xdst[XP];
#pragma unroll
  for (int it = 0; it < XP; ++it) {
    const int p = it * 64 + lane, row = p >> 4, col = p & 15;
    xsrc[it] = a.x + static_cast<size_t>(min(m0 + row, M - 1)) * K + col * 8;
    xdst[it] = row * XROW + col * 16;
  }

  f32x16 acc32[TN == 32 ? MT : 1];
  f32x4v acc16[TN == 16 ? MT : 1];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    if constexpr (TN == 32) acc32[i] = f32x16{0};
    else acc16[i] = f32x4v{0};
  }

  unsigned wv[NCH][WPC];
  unsigned short sv[NCH][4], zv[NCH][4];
  unsigned sw[NCH / 2], zw[NCH / 2];                // G128: the window's NCH scales / zeros, two per dword
  constexpr int XD = (XP <= 8) ? NCH : 1;          // activation chunks held in registers: the whole window when it fits
  u32x4 xs[XD][XP];

  auto load_x = [&](u32x4 (&xv)[XP], int c) {
    const int kc = min(c, nchunks - 1) * KC;        // clamped: a prefetch past the end re-reads the final chunk harmlessly
#pragma unroll
    for (int it = 0; it < XP; ++it) xv[it] = *reinterpret_cast<const u32x4*>(xsrc[it] + kc);
  };
  auto store_x = [&](const u32x4 (&xv)[XP]) {
#pragma unroll
    for (int it = 0; it < XP; ++it) *reinterpret_cast<u32x4*>(slab + xdst[it]) = xv[it];
  };
  auto load_w = [&](int cw) {                       // window of chunks cw .. cw+NCH-1 (cw <= nchunks - NCH)
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const uint32_t* src = wrow + ((cw + j) * KC / 32) * BITS;
#pragma unroll
      for (int q = 0; q < WPC; q += 4) {
        const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src + q));
        wv[j][q] = v.x; wv[j][q + 1] = v.y; wv[j][q + 2] = v.z; wv[j][q + 3] = v.w;
      }
    }
    if constexpr (G128) {                           // chunk == group: NCH consecutive 16-bit values per tensor
      static_assert(NCH == 4, "wide scale/zero load is written for a 4-chunk window");
      const u32x2 s2 = *reinterpret_cast<const u32x2*>(srow + cw);     // cw % 4 == 0 is not guaranteed: 2-byte aligned
      const u32x2 z2 = *reinterpret_cast<const u32x2*>(zrow + cw);
      sw[0] = s2.x; sw[1] = s2.y; zw[0] = z2.x; zw[1] = z2.y;
    } else {
#pragma unroll
      for (int j = 0; j < NCH; ++j) {
        const int g0 = ((cw + j) * KC) >> a.log2g;
#pragma unroll
        for (int u = 0; u < 4; ++u) {               // unconditional (clamped) so the waitcnt pass can count
          sv[j][u] = srow[g0 + min(u, gpc - 1)];
          zv[j][u] = zrow[g0 + min(u, gpc - 1)];
        }
      }
    }
  };
  auto compute = [&](int j) {
    float sf[4], zf[4];
    if constexpr (G128) {
      const unsigned short sb = static_cast<unsigned short>((j & 1) ? (sw[j >> 1] >> 16) : (sw[j >> 1] & 0xFFFFu));
      const unsigned short zb = static_cast<unsigned short>((j & 1) ? (zw[j >> 1] >> 16) : (zw[j >> 1] & 0xFFFFu));
      sf[0] = sf[1] = sf[2] = sf[3] = act_to_f32<BF16>(sb);
      zf[0] = zf[1] = zf[2] = zf[3] = act_to_f32<BF16>(zb);
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) { sf[u] = act_to_f32<BF16>(sv[j][u]); zf[u] = act_to_f32<BF16>(zv[j][u]); }
    }
#pragma unroll
    for (int t = 0; t < TS; ++t) {
      const int gi = G128 ? 0 : ((t * KSTEP) >> a.log2g);   // group of this k-step inside the chunk (wave-uniform)
      float q[8];
      if constexpr (TN == 32) fields8<BITS>(wv[j], t, h, q);
      else fields8q<BITS>(wv[j], t, h, q);
      const float s = gi == 0 ? sf[0] : gi == 1 ? sf[1] : gi == 2 ? sf[2] : sf[3];
      const float z = gi == 0 ? zf[0] : gi == 1 ? zf[1] : gi == 2 ? zf[2] : zf[3];
      u32x4 bf;
      bf.x = pack2<BF16>(fmaf(s, q[0], z), fmaf(s, q[1], z));
      bf.y = pack2<BF16>(fmaf(s, q[2], z), fmaf(s, q[3], z));
      bf.z = pack2<BF16>(fmaf(s, q[4], z), fmaf(s, q[5], z));
      bf.w = pack2<BF16>(fmaf(s, q[6], z), fmaf(s, q[7], z));
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const u32x4 af = *reinterpret_cast<const u32x4*>(slab + (i * TN + r) * XROW + ((KSTEP / 8) * t + h) * 16);
        if constexpr (TN == 32) acc32[i] = mfma<BF16>(af, bf, acc32[i]);
        else acc16[i] = mfma16<BF16>(af, bf, acc16[i]);
      }
    }
  };

  // Windows never run past the row: the last one is shifted back and skips the chunks an earlier window already did.
  for (int cb = c0; cb < c1; cb += NCH) {
    const int cw = min(cb, nchunks - NCH);
    load_w(cw);
#pragma unroll
    for (int d = 0; d < XD; ++d) load_x(xs[d], cw + d);
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      store_x(xs[j % XD]);                          // chunk cw+j: registers -> this wave's slab (previous reads are done)
      if constexpr (XD < NCH) load_x(xs[j % XD], cw + j + XD);
      if (cw + j >= cb && cw + j < c1) compute(j);
    }
  }
  __syncthreads();                                  // the slabs become the reduction buffer

  // ---- K-split reduction through LDS, then one rounding to the activation dtype (+ bias as a second rounding) ----
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int e = 0; e < AE; ++e) {
      float v;
      if constexpr (TN == 32) v = acc32[i][e]; else v = acc16[i][e];
      red[((wave * MT + i) * AE + e) * 64 + lane] = v;
    }
  __syncthreads();
  constexpr int TOTAL = MT * AE * 64;               // output elements of the workgroup
  for (int idx = tid; idx < TOTAL; idx += KW * 64) {
    const int ie = idx >> 6, ln = idx & 63;         // ie = i*AE + e
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < KW; ++w) v += red[(w * MT * AE + ie) * 64 + ln];
    const int i = ie / AE, e = ie % AE;
    const int n = nl0 + (ln & (TN - 1));
    // C/D layouts: 32x32: row = (e&3) + 8*(e>>2) + 4*(lane>>5); 16x16: row = 4*(lane>>4) + e
    const int m = m0 + i * TN + (TN == 32 ? (e & 3) + 8 * (e >> 2) + 4 * (ln >> 5) : 4 * (ln >> 4) + e);
    if (n < NL && m < M) {
      if (bs) v = act_to_f32<BF16>(f32_to_act<BF16>(v)) + act_to_f32<BF16>(bs[n]);
      yo[static_cast<size_t>(m) * NL + n] = f32_to_act<BF16>(v);
    }
  }
}

template <int BITS, int TN, int MT, int KW, bool G128>
static hipError_t launch_skinny_cfg(const MfmaArgs& a, bool bf16, hipStream_t s) {
  constexpr int NCH = 4, ROWS = MT * TN, AE = TN == 32 ? 16 : 4;
  dim3 grid((a.N + TN - 1) / TN, (a.M + ROWS - 1) / ROWS), block(KW * 64);
  const size_t red_bytes = static_cast<size_t>(KW) * MT * AE * 64 * sizeof(float);
  const size_t slab_bytes = static_cast<size_t>(KW) * ROWS * XROW;
  const size_t shmem = red_bytes > slab_bytes ? red_bytes : slab_bytes;
  if (shmem > 48 * 1024) {        // opt in to the large LDS carve-out once per instantiation
    static DeviceOnce done[2];
    if (done[bf16].pending()) {
      hipError_t e = bf16 ? hipFuncSetAttribute(reinterpret_cast<const void*>(&qgemm_skinny_kernel<BITS, TN, MT, KW, NCH, G128, true>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(shmem))
                          : hipFuncSetAttribute(reinterpret_cast<const void*>(&qgemm_skinny_kernel<BITS, TN, MT, KW, NCH, G128, false>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(shmem));
      if (e != hipSuccess) return e;
      done[bf16].done();
    }
  }
  if (bf16) hipLaunchKernelGGL((qgemm_skinny_kernel<BITS, TN, MT, KW, NCH, G128, true>), grid, block, shmem, s, a);
  else hipLaunchKernelGGL((qgemm_skinny_kernel<BITS, TN, MT, KW, NCH, G128, false>), grid, block, shmem, s, a);
  return hipGetLastError();
}

template <int BITS, int TN, int MT>
static hipError_t launch_skinny_mt(const MfmaArgs& a, bool bf16, hipStream_t s) {
  // 8-way K split (a wave holds its whole K range in one window at K = 4096), 4-way for short rows
  const int nchunks = a.K / KC;
  // wide scale/zero loads need every window to start at an even chunk (4-byte aligned pair of 16-bit values)
  const bool g128 = a.log2g == 7 && (nchunks % 16) == 0;
  if (nchunks >= 32) return g128 ? launch_skinny_cfg<BITS, TN, MT, 8, true>(a, bf16, s) : launch_skinny_cfg<BITS, TN, MT, 8, false>(a, bf16, s);
  return g128 ? launch_skinny_cfg<BITS, TN, MT, 4, true>(a, bf16, s) : launch_skinny_cfg<BITS, TN, MT, 4, false>(a, bf16, s);
}

template <int BITS>
static hipError_t launch_skinny_bits(const MfmaArgs& a, bool bf16, hipStream_t s) {
  if (a.N <= 4096) {              // narrow projection: 16-row tiles put up to 256 workgroups (one per CU) on the chip
    if (a.M <= 16) return launch_skinny_mt<BITS, 16, 1>(a, bf16, s);
    if (a.M <= 32) return launch_skinny_mt<BITS, 16, 2>(a, bf16, s);
    return launch_skinny_mt<BITS, 16, 4>(a, bf16, s);
  }
  return a.M <= 32 ? launch_skinny_mt<BITS, 32, 1>(a, bf16, s) : launch_skinny_mt<BITS, 32, 2>(a, bf16, s);
}

static bool use_batch_kernel(const MfmaArgs& a, int bits) {
  static const bool off = (gbx::env_switch("GBX_NO_BATCH_KERNEL") != 0);   // diagnostic: time the split-K kernel instead
  // 33..64 rows: two passes pay off only where the wide (128-row) workgroups apply (gate|up: 61 -> 40 us at M = 64;
  // the narrow projections are faster in one pass of the dequantising kernel)
  if (a.M > 32 && a.N < 128 * 192) return false;
  return !off && a.M <= 64 && qgemm_batch_supported(a.M <= 32 ? a.M : 32, a.K, bits, a.log2g);
}

// The no-dequant kernel takes 32 activation rows; 33..64 rows run as two launches (the second pass finds the packed
// weights in L2 / Infinity Cache), which still beats one pass of the dequantising kernel.
static hipError_t launch_batch_rows(const MfmaArgs& a, int bits, bool bf16, hipStream_t s) {
  if (a.M <= 32) return launch_qgemm_batch(a, bits, bf16, s);
  MfmaArgs lo = a;
  lo.M = 32;
  hipError_t e = launch_qgemm_batch(lo, bits, bf16, s);
  if (e != hipSuccess) return e;
  MfmaArgs hi = a;
  hi.M = a.M - 32;
  hi.x = a.x + static_cast<size_t>(32) * a.K;
  const bool multi = a.nseg > 1;          // (SwiGLU launches write ONE [M, n_seg[0]] output: the same offset rule)
  hi.y = a.y + static_cast<size_t>(32) * (multi ? a.n_seg[0] : a.N);
  for (int i = 1; i < a.nseg; ++i) hi.y_x[i - 1] = a.y_x[i - 1] + static_cast<size_t>(32) * a.n_seg[i];
  return launch_qgemm_batch(hi, bits, bf16, s);
}

hipError_t launch_qgemm_skinny_multi(const void* x, int nseg, const uint32_t* const* qw, const void* const* sc, const void* const* ze,
                                     const void* const* bias, void* const* y, const int* n, int M, int K, int bits, int log2g,
                                     bool bf16, hipStream_t s) {
  MfmaArgs a{};
  a.x = static_cast<const unsigned short*>(x);
  a.qweight = qw[0]; a.scales = static_cast<const unsigned short*>(sc[0]); a.zeros = static_cast<const unsigned short*>(ze[0]);
  a.bias = static_cast<const unsigned short*>(bias[0]); a.y = static_cast<unsigned short*>(y[0]);
  a.M = M; a.K = K; a.log2g = log2g; a.nseg = nseg; a.N = 0;
  for (int i = 0; i < nseg; ++i) { a.n_seg[i] = n[i]; a.N += n[i]; }
  for (int i = 1; i < nseg; ++i) {
    a.qweight_x[i - 1] = qw[i]; a.scales_x[i - 1] = static_cast<const unsigned short*>(sc[i]);
    a.zeros_x[i - 1] = static_cast<const unsigned short*>(ze[i]); a.bias_x[i - 1] = static_cast<const unsigned short*>(bias[i]);
    a.y_x[i - 1] = static_cast<unsigned short*>(y[i]);
  }
  if (use_batch_kernel(a, bits)) {
    const hipError_t e = launch_batch_rows(a, bits, bf16, s);
    if (e != hipErrorInvalidValue) return e;        // segment rows the wide workgroups cannot tile: split-K kernel
  }
  switch (bits) {
    case 2: return launch_skinny_bits<2>(a, bf16, s);
    case 3: return launch_skinny_bits<3>(a, bf16, s);
    case 4: return launch_skinny_bits<4>(a, bf16, s);
    default: return hipErrorInvalidValue;
  }
}

// gate_proj | up_proj + SwiGLU in one launch of the no-dequant batch kernel (2 <= M <= 64); hipErrorInvalidValue when the
// shape is outside it (the caller then runs the three ops)
hipError_t launch_qgemm_swiglu(const void* x, const uint32_t* gq, const void* gs, const void* gz, const uint32_t* uq, const void* us,
                               const void* uz, void* y, int M, int N, int K, int bits, int log2g, bool bf16, hipStream_t s) {
  MfmaArgs a{};
  a.x = static_cast<const unsigned short*>(x);
  a.qweight = gq; a.scales = static_cast<const unsigned short*>(gs); a.zeros = static_cast<const unsigned short*>(gz);
  a.qweight_x[0] = uq; a.scales_x[0] = static_cast<const unsigned short*>(us); a.zeros_x[0] = static_cast<const unsigned short*>(uz);
  a.y = static_cast<unsigned short*>(y);
  a.M = M; a.K = K; a.log2g = log2g; a.nseg = 2; a.n_seg[0] = a.n_seg[1] = N; a.N = 2 * N; a.swiglu = 1;
  if (M < 2 || M > 64 || !qgemm_batch_supported(M <= 32 ? M : 32, K, bits, log2g)) return hipErrorInvalidValue;
  return launch_batch_rows(a, bits, bf16, s);
}

// k-split batch GEMM that stops at the fp32 slices ws[slices][M][N] (2 <= M <= 32; hipErrorInvalidValue when this shape does
// not split): the summing launch is folded into the consumer (gbx_add_rmsnorm_slices)
hipError_t launch_qgemm_partials(const void* x, const uint32_t* qw, const void* sc, const void* ze, float* ws, int M, int N, int K,
                                 int bits, int log2g, bool bf16, hipStream_t s) {
  MfmaArgs a{};
  a.x = static_cast<const unsigned short*>(x); a.qweight = qw; a.scales = static_cast<const unsigned short*>(sc);
  a.zeros = static_cast<const unsigned short*>(ze); a.M = M; a.N = N; a.K = K; a.log2g = log2g;
  a.ws = ws; a.partials_only = 1;
  if (M < 2 || M > 32 || qgemm_batch_slices(M, N, K, bits, log2g) <= 1) return hipErrorInvalidValue;
  return launch_qgemm_batch(a, bits, bf16, s);
}

hipError_t launch_qgemm_skinny(const void* x, const uint32_t* qw, const void* sc, const void* ze, const void* bias, void* y,
                               int M, int N, int K, int bits, int log2g, bool bf16, hipStream_t s, float* ws) {
  MfmaArgs a{};
  a.x = static_cast<const unsigned short*>(x); a.qweight = qw; a.scales = static_cast<const unsigned short*>(sc);
  a.zeros = static_cast<const unsigned short*>(ze); a.bias = static_cast<const unsigned short*>(bias);
  a.y = static_cast<unsigned short*>(y); a.M = M; a.N = N; a.K = K; a.log2g = log2g;
  a.ws = M <= 32 ? ws : nullptr;                    // caller-owned fp32 workspace: k split across workgroups (mfma_batch.hip)
  if (use_batch_kernel(a, bits)) {
    const hipError_t e = launch_batch_rows(a, bits, bf16, s);
    if (e != hipErrorInvalidValue) return e;        // segment rows the wide workgroups cannot tile: split-K kernel
  }
  switch (bits) {
    case 2: return launch_skinny_bits<2>(a, bf16, s);
    case 3: return launch_skinny_bits<3>(a, bf16, s);
    case 4: return launch_skinny_bits<4>(a, bf16, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace gbx
