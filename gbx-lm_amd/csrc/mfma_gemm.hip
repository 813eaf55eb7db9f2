// Prefill GBA QuantizedLinear (M > 64): y[M,N] = x[M,K] . dequant(W[N,K])^T on the matrix cores (gfx950 MFMA).
//
// Replaces mx.quantized_matmul(..., transpose=True) for prompts / batched evaluation
// (gbx_lm/models/quantized_linear_gba.py:195-203 when x is [B, L, K] with B*L > 64; e.g. the chunked prefill at
// gbx_lm/utils.py:312-319 and the batched eval at gbx_lm/evaluate.py:144-185).
//
// Here the dequantised tile is a genuine dense contraction, so it runs on v_mfma_f32_32x32x16_{bf16,f16}.  What limits a
// dequant-GEMM is the VALU work of turning packed fields into 16-bit operands, so every weight is dequantised ONCE per
// workgroup and shared through LDS by all of its (64*MT) activation rows:
//   * workgroup tile (64*MT) x 128, K chunks of 64, 4 waves as 2 (M) x 2 (N); a wave owns a (32*MT) x 64 tile =
//     MT x 2 accumulators of 32x32.
//   * B tile: 128 weight rows x 64 k.  Thread (row = tid/2, half = tid%2) owns 32 consecutive values of one row per chunk
//     -- 2 / 3 / 4 packed words, always inside one group -- and dequantises them as fields -> fp32, w = s*q + z (fp32 fma),
//     ONE rounding to the activation dtype (v_cvt_pk), then four 16-byte LDS stores.  That is the arithmetic of a
//     dequantise-then-matmul (what MLX's qmm does); the exact-integer algebraic form is kept for the M = 1 GEMV.
//   * A tile: (64*MT) activation rows x 64 k, fetched coalesced (8 rows x 128 B per wave instruction) into registers
//     while the previous chunk is on the matrix cores, then stored to LDS.
//   * LDS rows are 128 B + 16 B pad, so the 16-byte fragment reads (lane = row, 8 consecutive k) are conflict-free;
//     two buffers, one barrier per chunk.
// MFMA-bound (2*M*N*K flop); the weights are re-read from L2/HBM once per (64*MT)-row tile.
#include "mfma_common.h"
#include "gbx_env.h"
#include "kernels.h"
#include <cstdlib>

namespace gbx {

constexpr int PKC = 64;                  // k-values per chunk
constexpr int PROW = PKC * 2 + 16;       // LDS bytes per tile row (activations and dequantised weights alike)

// WV waves per workgroup as 2 (M) x WV/2 (N); a wave owns a (32*MT) x (32*NT) tile.  WV = 8 puts two waves on every
// SIMD, so one wave's staging (global loads: 64 B/clk per CU, LDS stores, dequantise VALU) issues under the other's MFMAs
// -- with one wave per SIMD those issue costs add to the MFMA time (stamps: 1.2k cycles of load issue + 3.3k of body per
// 2k-cycle chunk of MFMAs).
template <int BITS, int MT, int NT, int WV, bool BF16>
__global__ __launch_bounds__(WV * 64) void qgemm_prefill_kernel(const MfmaArgs a) {
  constexpr int NTH = WV * 64, WN = WV / 2;
  constexpr int BM = 64 * MT;                     // activation rows per workgroup
  constexpr int PBN = 32 * NT * WN;               // weight rows (output columns) per workgroup
  constexpr bool DENSE = BITS == 16;              // weights pre-dequantised (a.wdense): the B tile is staged like the A tile
  constexpr int WB = DENSE ? 1 : BITS;            // packed words per 32-value unit
  constexpr int NU = PBN * 2 / NTH;               // 32-value units of the B tile per thread and chunk
  constexpr int AP = BM * 8 / NTH;                // 16-byte activation pieces per thread per chunk
  constexpr int BP = DENSE ? PBN * 8 / NTH : 4;   // dense: 16-byte weight pieces per thread per chunk
  static_assert(NU >= 1 && AP >= 4 && BP % 4 == 0, "tile too small for the thread count");
  constexpr int ABYTES = BM * PROW, BBYTES = PBN * PROW, STAGE = ABYTES + BBYTES;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // 2 stages of [A tile | B tile]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;
  const int K = a.K, N = a.N, M = a.M;
  const int W = K / 32 * BITS, G = K >> a.log2g;
  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (each with its own L2), so workgroup b runs on
  // XCD b % 8.  Give every XCD a CONTIGUOUS run of tiles in (M-tile major, N-tile minor) order: the ~32 workgroups an XCD
  // holds at a time then share one or two activation tiles, which are fetched into that L2 once and hit by the others
  // (with the plain 2-D grid the column tiles of one activation tile were spread over all eight L2s: 8x the misses, and
  // the activation requests -- 32 KB per chunk and workgroup -- queued behind the per-CU miss capacity).
  const int tiles_n = (N + PBN - 1) / PBN, tiles_m = (M + BM - 1) / BM;
  const int tiles = tiles_m * tiles_n;
  const int per_xcd = (tiles + 7) / 8;
  const int tile = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (tile >= tiles) return;
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * PBN;
  // k split across workgroups (blockIdx.y; short prompts leave most CUs without a tile): this workgroup's chunk range
  const int ksplit = a.ws ? a.ksplit : 1;
  const int nchunks = K / PKC / ksplit;
  const int kbase = blockIdx.y * nchunks * PKC;

  // B loader: unit u = tid*NU + j -> row u/2 of the tile, 32-value half u%2 of the chunk (NU = 2: the whole row chunk)
  const int brow = (tid * NU) >> 1, bhalf = (tid * NU) & 1;
  const int bn = min(n0 + brow, N - 1);
  const uint32_t* wsrc = a.qweight + static_cast<size_t>(bn) * W + bhalf * WB;
  const unsigned short* bsrc[BP];                 // dense: piece p = tid + NTH*i -> weight row p/8, 16-byte column p%8
  int bdstd[BP];
#pragma unroll
  for (int i = 0; i < BP; ++i) {
    const int p = tid + NTH * i, row = p >> 3, col = p & 7;
    bsrc[i] = (DENSE ? a.wdense : a.x) + static_cast<size_t>(min(n0 + row, N - 1)) * K + col * 8;
    bdstd[i] = ABYTES + row * PROW + col * 16;
  }
  const unsigned short* ssrc = a.scales + static_cast<size_t>(bn) * G;
  const unsigned short* zsrc = a.zeros + static_cast<size_t>(bn) * G;
  const int bdst = ABYTES + brow * PROW + bhalf * 64;
  // A loader: piece p = tid + NTH*i -> row p/8, 16-byte column p%8
  const unsigned short* asrc[AP];
  int adst[AP];
#pragma unroll
  for (int i = 0; i < AP; ++i) {
    const int p = tid + NTH * i, row = p >> 3, col = p & 7;
    asrc[i] = a.x + static_cast<size_t>(min(m0 + row, M - 1)) * K + col * 8;
    adst[i] = row * PROW + col * 16;
  }

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x16{0};

  // Two register sets: chunk c+2 is requested while chunk c is on the matrix cores and chunk c+1 moves from registers
  // to LDS, so a global load has a whole iteration (>= 1000 MFMA cycles) before its first use.
  u32x4 areg[2][AP];
  u32x4 breg[2][DENSE ? BP : 1];
  unsigned wreg[2][NU * WB];
  unsigned short sreg[2][NU], zreg[2][NU];

  // global -> registers, in four parts so that the requests of a chunk spread over the k-steps of the chunk before it
  // (the vector-memory path moves 64 B/clk per CU: a burst of 10-14 requests per thread at the top of a chunk holds
  // every wave for ~1k cycles).  Part 0 carries the scales / zeros and packed words, every part a quarter of the
  // activation pieces.  Clamped: late prefetches re-read the final chunk.
  auto load_part = [&](int set, int c, int t) {
    const int kc = kbase + min(c, nchunks - 1) * PKC;
    if constexpr (DENSE) {
#pragma unroll
      for (int i = t * (BP / 4); i < (t + 1) * (BP / 4); ++i) breg[set][i] = *reinterpret_cast<const u32x4*>(bsrc[i] + kc);
    } else
    if (t == 0) {
#pragma unroll
      for (int j = 0; j < NU; ++j) {
        const int g = (kc + (bhalf + j) * 32) >> a.log2g;
        sreg[set][j] = ssrc[g];
        zreg[set][j] = zsrc[g];
      }
      const uint32_t* src = wsrc + (kc / 32) * WB;
      if constexpr (NU * WB % 4 == 0) {
#pragma unroll
        for (int q = 0; q < NU * WB; q += 4) {
          const u32x4 v = *reinterpret_cast<const u32x4*>(src + q);
          wreg[set][q] = v.x; wreg[set][q + 1] = v.y; wreg[set][q + 2] = v.z; wreg[set][q + 3] = v.w;
        }
      } else if constexpr (NU * WB % 2 == 0) {
#pragma unroll
        for (int q = 0; q < NU * WB; q += 2) {
          const u32x2 v = *reinterpret_cast<const u32x2*>(src + q);
          wreg[set][q] = v.x; wreg[set][q + 1] = v.y;
        }
      } else {
#pragma unroll
        for (int q = 0; q < NU * WB; ++q) wreg[set][q] = src[q];
      }
    }
#pragma unroll
    for (int i = t * (AP / 4); i < (t + 1) * (AP / 4); ++i) areg[set][i] = *reinterpret_cast<const u32x4*>(asrc[i] + kc);
  };
  auto load_chunk = [&](int set, int c) {
#pragma unroll
    for (int t = 0; t < 4; ++t) load_part(set, c, t);
  };
  // quarter t (0..3) of: registers -> LDS stage, dequantising the weights on the way
  auto store_quarter = [&](int set, int stage, int t) {
    unsigned char* base = lds + stage * STAGE;
#pragma unroll
    for (int i = t * (AP / 4); i < (t + 1) * (AP / 4); ++i) *reinterpret_cast<u32x4*>(base + adst[i]) = areg[set][i];
    if constexpr (DENSE) {
#pragma unroll
      for (int i = t * (BP / 4); i < (t + 1) * (BP / 4); ++i) *reinterpret_cast<u32x4*>(base + bdstd[i]) = breg[set][i];
    } else
#pragma unroll
    for (int j = 0; j < NU; ++j) {                // octet t of every unit this thread owns
      const unsigned* wu = wreg[set] + j * WB;
      unsigned win;
      if constexpr (BITS == 4) win = wu[t];
      else if constexpr (BITS == 2) win = (t & 1) ? (wu[t >> 1] >> 16) : wu[t >> 1];
      else {
        const int b = 24 * t, w = b >> 5;         // 96-bit period: octet t starts at bit 24*t
        win = __builtin_amdgcn_alignbit(wu[w + 1 <= 2 ? w + 1 : 2], wu[w], b & 31);
      }
      float q[8];
      decode8<DENSE ? 4 : BITS>(win, q);
      const float s = act_to_f32<BF16>(sreg[set][j]), z = act_to_f32<BF16>(zreg[set][j]);
      u32x4 v;
      v.x = pack2<BF16>(fmaf(s, q[0], z), fmaf(s, q[1], z));
      v.y = pack2<BF16>(fmaf(s, q[2], z), fmaf(s, q[3], z));
      v.z = pack2<BF16>(fmaf(s, q[4], z), fmaf(s, q[5], z));
      v.w = pack2<BF16>(fmaf(s, q[6], z), fmaf(s, q[7], z));
      *reinterpret_cast<u32x4*>(base + bdst + j * 64 + t * 16) = v;
    }
  };
  // Fragments are read one k-step ahead of the MFMAs that use them (two register sets), so the LDS latency of step t+1
  // hides under the 2*MT MFMAs of step t; only the first step of a chunk (right after the barrier) waits for LDS.
  u32x4 fa[2][MT], fb[2][NT];
  auto read_frags = [&](int fs, int stage, int t) {
    const unsigned char* abase = lds + stage * STAGE + (wm * 32 * MT + r) * PROW + h * 16;
    const unsigned char* bbase = lds + stage * STAGE + ABYTES + (wn * 32 * NT + r) * PROW + h * 16;
#pragma unroll
    for (int j = 0; j < NT; ++j) fb[fs][j] = *reinterpret_cast<const u32x4*>(bbase + j * 32 * PROW + t * 32);
#pragma unroll
    for (int i = 0; i < MT; ++i) fa[fs][i] = *reinterpret_cast<const u32x4*>(abase + i * 32 * PROW + t * 32);
  };
  auto mfma_step = [&](int fs) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = mfma<BF16>(fa[fs][i], fb[fs][j], acc[i][j]);
  };
  // one chunk: LDS stage `stage` on the matrix cores while register set `set` (chunk c+1) moves into the other stage and
  // the set just emptied quarter by quarter is refilled with chunk c+2
  auto chunk = [&](int stage, int set, int cnext) {
    read_frags(0, stage, 0);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (t < 3) read_frags((t + 1) & 1, stage, t + 1);
      mfma_step(t & 1);
      store_quarter(set, stage ^ 1, t);
      load_part(set ^ 1, cnext, t);
    }
  };
  static_assert(AP % 4 == 0, "activation pieces are stored in four quarters");

  load_chunk(0, 0);
  load_chunk(1, 1);
#pragma unroll
  for (int t = 0; t < 4; ++t) store_quarter(0, 0, t);
  __syncthreads();
#ifdef GBX_PSTAMPS
  long long t_bar = 0, t_issue = 0, t_body = 0;
  const long long t_start = __builtin_amdgcn_s_memtime();
#define PST(var, code) { const long long _t0 = __builtin_amdgcn_s_memtime(); code; var += __builtin_amdgcn_s_memtime() - _t0; }
#else
#define PST(var, code) { code; }
#endif
  for (int c = 0; c < nchunks; c += 2) {
    // even chunk c: LDS stage 0 on the matrix cores; registers set 1 (chunk c+1) -> stage 1; request chunk c+2 into set 0
    PST(t_body, chunk(0, 1, c + 2); __builtin_amdgcn_sched_barrier(0));
    PST(t_bar, __syncthreads());
    // odd chunk c+1 (K % 128 == 0: the chunk count is even): stage 1; set 0 (chunk c+2) -> stage 0; request chunk c+3
    PST(t_body, chunk(1, 0, c + 3); __builtin_amdgcn_sched_barrier(0));
    PST(t_bar, __syncthreads());
  }
#ifdef GBX_PSTAMPS
  if (lane == 0 && a.stamps) {
    long long* o = a.stamps + (static_cast<size_t>(blockIdx.x) * 4 + wave) * 4;
    o[0] = __builtin_amdgcn_s_memtime() - t_start; o[1] = t_issue; o[2] = t_body; o[3] = t_bar;
  }
#endif

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
          if (ksplit > 1) {                         // fp32 partial of this k slice; launch_splitk_finish sums and rounds
            a.ws[(static_cast<size_t>(blockIdx.y) * M + m) * N + n] = v;
          } else {
            if (a.bias) v = act_to_f32<BF16>(f32_to_act<BF16>(v)) + b;
            a.y[static_cast<size_t>(m) * N + n] = f32_to_act<BF16>(v);
          }
        }
      }
    }
  }
}

template <int BITS, int MT, int NT, int WV>
static hipError_t launch_prefill_cfg(const MfmaArgs& a, bool bf16, hipStream_t s) {
  constexpr int BM = 64 * MT, BN = 32 * NT * (WV / 2);
  const int tiles = ((a.N + BN - 1) / BN) * ((a.M + BM - 1) / BM);
  dim3 grid(8 * ((tiles + 7) / 8), a.ws ? a.ksplit : 1), block(WV * 64);   // x: the kernel maps workgroup -> tile XCD-aware
  const size_t shmem = 2 * static_cast<size_t>(BM + BN) * PROW;
  if (shmem > 48 * 1024) {        // opt in to the large LDS carve-out once per instantiation
    static DeviceOnce done[2];
    if (done[bf16].pending()) {
      hipError_t e = bf16 ? hipFuncSetAttribute(reinterpret_cast<const void*>(&qgemm_prefill_kernel<BITS, MT, NT, WV, true>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(shmem))
                          : hipFuncSetAttribute(reinterpret_cast<const void*>(&qgemm_prefill_kernel<BITS, MT, NT, WV, false>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(shmem));
      if (e != hipSuccess) return e;
      done[bf16].done();
    }
  }
  if (bf16) hipLaunchKernelGGL((qgemm_prefill_kernel<BITS, MT, NT, WV, true>), grid, block, shmem, s, a);
  else hipLaunchKernelGGL((qgemm_prefill_kernel<BITS, MT, NT, WV, false>), grid, block, shmem, s, a);
  return hipGetLastError();
}

// Short prompts (65..~512 rows) give the 128 x 128 tiling only N/128 x M/128 workgroups, each walking all of K alone
// (48 us for 4096 x 4096 at any M <= 512, 160 us for the down projection): with caller scratch the k range is split
// over up to 8 workgroups per tile and a second launch sums the fp32 slices (mfma_batch.hip: launch_splitk_finish).
static int prefill_ksplit(int M, int N, int K) {
  const long tiles = static_cast<long>((M + 127) / 128) * ((N + 127) / 128);
  if (M > 2048 || tiles > 128 || (N & 3)) return 1;   // (measured: beyond 128 tiles the split and its summing launch lose)
  int best = 1;
  for (int s = 2; s <= 8; s *= 2)
    if ((K / PKC) % (2 * s) == 0 && tiles * s <= 512) best = s;
  return best;
}

size_t qgemm_prefill_workspace_bytes(int M, int N, int K) {
  const int s = prefill_ksplit(M, N, K);
  return s > 1 ? static_cast<size_t>(s) * M * N * sizeof(float) : 0;
}

// Which workgroup tile a prefill-sized call (M > 64) runs on: 1 = 128 x 128, 2 = 256 x 128, 3 = 256 x 256 (8 waves),
// 5 = 128 x 128 with k split over workgroups + summing launch (caller scratch).  The biggest tile that still gives every
// CU a workgroup: LDS bytes per flop fall with the tile.  GBX_PREFILL_TILE=1..4 forces one (tuning aid).
int prefill_tile_id(int M, int N, int K, bool has_ws) {
  if (has_ws && prefill_ksplit(M, N, K) > 1) return 5;
  static const int force = gbx::env_switch("GBX_PREFILL_TILE");
  const long t256 = static_cast<long>((M + 255) / 256) * ((N + 255) / 256);
  const long t128 = static_cast<long>((M + 255) / 256) * ((N + 127) / 128);
  if (force == 4) return 4;
  if (force == 3 || (force == 0 && t256 >= 256)) return 3;
  if (force == 2 || (force == 0 && t128 >= 128 && M > 128)) return 2;
  return 1;
}

template <int BITS>
static hipError_t launch_prefill_bits(const MfmaArgs& a0, bool bf16, hipStream_t s) {
  MfmaArgs a = a0;
  a.ksplit = a.ws ? prefill_ksplit(a.M, a.N, a.K) : 1;
  if (a.ksplit <= 1) a.ws = nullptr;
  if (a.ws) {
    hipError_t e = launch_prefill_cfg<BITS, 2, 2, 4>(a, bf16, s);                                        // 128 x 128 x ksplit
    if (e != hipSuccess) return e;
    return launch_splitk_finish(a.ws, a.bias, a.y, a.ksplit, a.M, a.N, bf16, s);
  }
  switch (prefill_tile_id(a.M, a.N, a.K, false)) {
    case 4: return launch_prefill_cfg<BITS, 4, 4, 4>(a, bf16, s);     // 256 x 256, 4 waves (tuning aid only)
    case 3: return launch_prefill_cfg<BITS, 4, 2, 8>(a, bf16, s);     // 256 x 256, 8 waves
    case 2: return launch_prefill_cfg<BITS, 4, 2, 4>(a, bf16, s);     // 256 x 128
    default: return launch_prefill_cfg<BITS, 2, 2, 4>(a, bf16, s);    // 128 x 128
  }
}

// Long prompts (M >= GBX_DENSE_MIN_M rows): dequantise ONCE per call into the caller's workspace (fields -> fp32, w = s*q + z,
// one rounding: the arithmetic of the fused kernel's staging pass, shared by all M / 256 row tiles instead of being redone by
// each of them), then the same tile kernel with the weight tile staged like the activation tile (no VALU work per element).
hipError_t launch_qgemm_dense(const void* x, const uint32_t* qw, const void* sc, const void* ze, const void* bias, void* y, int M,
                              int N, int K, int bits, int log2g, bool bf16, hipStream_t s, void* wdense, float* part) {
  hipError_t e = launch_dequant_units(qw, sc, ze, wdense, N, K, bits, log2g, bf16, s);
  if (e != hipSuccess) return e;
  static const bool no8p = (gbx::env_switch("GBX_DENSE_NO_8PHASE") != 0);      // tuning aid: the register-staged tiles on dense weights
  if (!no8p && gemm_dense_supported(M, N, K)) return launch_gemm_dense(x, wdense, bias, y, M, N, K, bf16, s, false, part);   // mfma_dense.hip
  MfmaArgs a{};
  a.x = static_cast<const unsigned short*>(x); a.wdense = static_cast<const unsigned short*>(wdense);
  a.qweight = qw; a.scales = static_cast<const unsigned short*>(sc); a.zeros = static_cast<const unsigned short*>(ze);
  a.bias = static_cast<const unsigned short*>(bias); a.y = static_cast<unsigned short*>(y);
  a.M = M; a.N = N; a.K = K; a.log2g = log2g;
  return launch_prefill_bits<16>(a, bf16, s);
}

hipError_t launch_qgemm_dense_swiglu(const void* x, const uint32_t* gq, const void* gs, const void* gz, const uint32_t* uq, const void* us,
                                     const void* uz, void* y, int M, int N, int K, int bits, int log2g, bool bf16, hipStream_t s,
                                     void* wdense) {
  if (N % 128 || !gemm_dense_supported(M, 2 * N, K)) return hipErrorInvalidValue;
  hipError_t e = launch_dequant_units(gq, gs, gz, wdense, N, K, bits, log2g, bf16, s, 128, 0);
  if (e != hipSuccess) return e;
  e = launch_dequant_units(uq, us, uz, wdense, N, K, bits, log2g, bf16, s, 128, 1);
  if (e != hipSuccess) return e;
  return launch_gemm_dense(x, wdense, nullptr, y, M, 2 * N, K, bf16, s, true);
}

hipError_t launch_qgemm_mfma(const void* x, const uint32_t* qw, const void* sc, const void* ze, const void* bias, void* y,
                             int M, int N, int K, int bits, int log2g, bool bf16, hipStream_t s, float* ws) {
  MfmaArgs a{};
  a.ws = ws;
  a.x = static_cast<const unsigned short*>(x); a.qweight = qw; a.scales = static_cast<const unsigned short*>(sc);
  a.zeros = static_cast<const unsigned short*>(ze); a.bias = static_cast<const unsigned short*>(bias);
  a.y = static_cast<unsigned short*>(y); a.M = M; a.N = N; a.K = K; a.log2g = log2g;
#ifdef GBX_PSTAMPS
  a.stamps = reinterpret_cast<long long*>(strtoull(getenv("GBX_STAMP_PTR") ? getenv("GBX_STAMP_PTR") : "0", nullptr, 0));
#endif
  switch (bits) {
    case 2: return launch_prefill_bits<2>(a, bf16, s);
    case 3: return launch_prefill_bits<3>(a, bf16, s);
    case 4: return launch_prefill_bits<4>(a, bf16, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace gbx
