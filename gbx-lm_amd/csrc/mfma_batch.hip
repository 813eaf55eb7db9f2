// Batched decode (2 <= M <= 32 rows of activations), 2-, 3- and 4-bit at group 128, 2-bit also at groups 64 / 32: y[M,N] = x[M,K] . dequant(W[N,K])^T without
// dequantising.  Replaces mx.quantized_matmul for x of shape [B, 1, K] (gbx_lm/models/quantized_linear_gba.py:195-203).
//
// At these M the packed weights are read once and mfma_skinny.hip was bound by everything except HBM: ~4 VALU
// instructions per weight element to build bf16 weights, the activation tile re-read from L2 by every wave (8 bytes of x
// per byte of 2-bit weight at M = 32), and one window of loads per wave with nothing overlapping it.  Here:
//   * the integer fields go to the matrix cores as they are.  A 2- or 4-bit field ORed under the exponent of 128.0
//     (bf16) / 1024.0 (fp16) IS the number 128+q / 1024+q, and two fields 16 bits apart in a packed word make one MFMA
//     operand dword with a shift and one v_and_or: 1 VALU instruction per element instead of 4.  Per group
//     (= 128-value chunk) the MFMA accumulates t[m,n] = sum_k x[m,k] (C + q[n,k]) exactly as an fp32 sum of exact
//     products, and the affine map is applied to the 32x32 partial tile once per group:
//         acc[m,n] += s[n,g] * t[m,n] + (z[n,g] - C s[n,g]) * xsum[m,g],   xsum[m,g] = sum_{k in g} x[m,k]
//     (w = s q + z is never rounded to the activation dtype, so this is closer to the fp64 oracle than the dequantising
//     kernels; same tolerance in the tests).
//   * the k order inside a chunk is free as long as both MFMA operands agree, so lane-group h of a wave owns the h-th
//     1/H of the chunk: a lane loads ONLY the packed words it consumes (16 bytes per chunk at 2 bits), and the
//     activation tile is stored in LDS in the pair order the fields come out of a word ((i, i+8) at 2 bits, (i, i+4) at
//     4 bits) by the loader threads, which also produce xsum with v_dot2.
//   * a workgroup is 8 compute waves = NG row groups x KG k-ranges, plus 4 loader waves.  The NG waves of a k-range
//     share ONE activation slab, so x is read from L2 once per NG*TNW weight rows (128 rows for the wide projections:
//     2 bytes of x per byte of weight).  The loader waves stage step s+1 (8 chunks across the workgroup: load, pair
//     order, xsum, LDS) while the compute waves run step s with the packed words of step s+1 in flight; when every
//     wave did both, the load-issue phase and the MFMA phase alternated (in-kernel stamps: 2.3k + 2.8k cycles of
//     staging and barrier wait per 3.8k cycles of MFMA work).  The KG partial tiles meet in LDS at the end; one
//     rounding to the activation dtype.
// Measured at M = 32, 2 bits (scratch/bench_skinny32.py, us, split-K kernel -> this one): gate|up 28672x4096 45.2 -> 19.6,
// down 4096x14336 33.6 -> 25.6, q|k|v 6144x4096 11.8 -> 10.7, o 4096x4096 9.6 -> 9.9.  What bounds it now is the L1 fill
// rate per CU: every workgroup still pulls the whole 256 KB activation tile through its vector-memory path (57 MB
// chip-wide beside 29 MB of weights for gate|up); sharing x further needs a k split ACROSS workgroups, i.e. a
// caller-owned fp32 workspace in the ABI (next round).
// Also tried: holding the loader waves back one barrier so that the compute waves' 6 loads per step are not queued behind
// the loaders' 64 (19.6 -> 23.4 us: the phase then starts late for the loaders).  What the numbers fit is a fill-rate
// limit per CU of ~24 KB/us from L2 or HBM alike (256 CUs x 24 GB/s = the 6.2 TB/s copy ceiling of the chip): a workgroup
// here pulls 256 KB of x + 128 KB of weights = 16 us.
// Round 2, two more measured negatives (M = 32, 2 bits, scratch/bench_skinny32.py):
//   * TWO steps of activations in flight per loader thread (second register set, 168 VGPRs + 15 spilled dwords):
//     28672x4096 22.0 -> 27.4 us, 4096x14336 18.0 -> 20.5 us -- more bulk requests queued in the CU's in-order vector-memory
//     path only delay the weight loads the compute waves wait for;
//   * 256-row workgroups (NG = 8, KG = 1: x staged once per 256 weight rows, half the activation traffic per CU) with the k
//     range split over two workgroups and the fp32 slices summed by the second launch: 28672x4096 21.0 -> 29.4 us (the GEMM
//     launch alone ~25 us): with 2 steps per workgroup the ramp / drain of the staging pipeline outweighs the bytes saved.
// In-kernel stamps of this kernel (compute waves, 4096-value rows = 4 steps): a step is ~8k cycles of which the MFMAs are
// ~2.5k; ISSUING the next step's 6 weight loads takes 3-5k (they queue behind the loaders' 64 requests in the in-order path).
// -DBX_STAMPS: diagnostic build with s_memtime phase stamps (scratch/stamps_batch.py).
#include "mfma_common.h"
#include "gbx_env.h"
#include "kernels.h"

namespace gbx {

bool qgemm_batch_supported(int M, int K, int bits, int log2g);
#ifdef BX_WITH_V2                                   // experiment builds only: scratch/negative/mfma_batch2.hip linked in (scratch/build_v2.sh)
bool qgemm_batch2_ok(const MfmaArgs& a, int rows);
hipError_t launch_qgemm_batch2(const MfmaArgs& a, int bits, int rows, bool bf16, hipStream_t s);
#endif

template <bool BF16> __device__ __forceinline__ float xdot2(unsigned v, float acc) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  if constexpr (BF16) return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, v), __builtin_bit_cast(bf2, 0x3F803F80u), acc, false);
  else return __builtin_amdgcn_fdot2(as_h2(v), as_h2(0x3C003C00u), acc, false);
}

#ifdef BX_STAMPS
__device__ long long* bx_stamp_ptr;                 // diagnostic build only (scratch/stamps_batch.py)
#endif

template <int BITS, int TNW, int NG, int KG, int GRP, bool BF16>
__global__ __launch_bounds__(768) void qgemm_batch_kernel(const MfmaArgs a) {
  static_assert(NG * KG == 8 && BITS >= 2 && BITS <= 4 && (TNW == 16 || TNW == 32), "unsupported shape");
  static_assert(GRP == 128 || (BITS == 2 && (GRP == 64 || GRP == 32)), "groups of 64 / 32 are built for 2 bits");
  // Groups smaller than the chunk (IL): an MFMA step must stay inside one group, so the k assignment is interleaved --
  // step t covers the 16 * H consecutive k-values [16 H t, 16 H (t+1)) with lane group h on unit t*H + h -- instead of
  // lane group h owning the h-th 1/H of the chunk; every lane then loads the chunk's 8 words and shifts by its half.
  // NF group flushes per chunk, SPG MFMA steps per group.
  constexpr bool IL = GRP < 128;
  constexpr int NF = 128 / GRP;
  constexpr int NCHW = 8 / KG;                      // chunks per step of one k-range (the workgroup does 8 per step)
  constexpr int H = 64 / TNW;                       // lane groups of a wave (k split inside a chunk)
  constexpr int UPL = 16 / H;                       // 16-byte activation units per lane and chunk = MFMA steps per chunk
  constexpr int WL = IL ? 8 : UPL * BITS / 4;       // packed words per lane and chunk
  constexpr int SPG = UPL / NF;                     // MFMA steps per group (>= 1: TNW = 16 at GRP = 32 -> 1)
  constexpr int WPC = KC * BITS / 32;               // packed words per row and chunk
  constexpr int MT = 32 / TNW;                      // activation row tiles (32 rows)
  constexpr int AE = TNW == 32 ? 16 : 4;            // accumulator elements per lane and tile
  constexpr int CHUNK = 32 * XROW;                  // slab bytes of one chunk (32 activation rows)
  constexpr int BUF = 8 * CHUNK;                    // one step of the whole workgroup
  constexpr unsigned MAGIC = BF16 ? 0x43004300u : 0x64006400u;   // 128.0 | 1024.0 twice
  constexpr float CF = BF16 ? 128.f : 1024.f;
  constexpr unsigned FMASK = BITS == 2 ? 0x00030003u : BITS == 3 ? 0x00070007u : 0x000F000Fu;
  // (f & FMASK) | MAGIC as ONE instruction: v_and_or_b32 is a VOP3 encoding, which on gfx9 takes no literal and one scalar operand, so left to
  // itself the compiler emits v_and_b32 + v_or_b32 with a 32-bit literal each (round 4: 11.6 -> 7.6 vector instructions per MFMA in the
  // compute waves).  The mask lives in an SGPR, the exponent image in a VGPR.
  unsigned fmask_s = FMASK, magic_v = MAGIC;
  asm volatile("" : "+s"(fmask_s));
  asm volatile("" : "+v"(magic_v));
  auto fld = [&](unsigned f) {
#ifdef BX_PLAIN_FIELDS                              // A/B build (scratch/build_exp.sh): what the compiler makes of the C expression
    return (f & FMASK) | MAGIC;
#else
    unsigned d;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(d) : "v"(f), "s"(fmask_s), "v"(magic_v));
    return d;
#endif
  };
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [2][8 chunks][32 rows][XROW] | xsum [2][8][NF][32]
  float* xsum = reinterpret_cast<float*>(smem + 2 * BUF);
  float* red = reinterpret_cast<float*>(smem);                            // after the main loop: [8 waves][MT*AE][64]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ng = wave % NG, kg = wave / NG;
  const int r = lane & (TNW - 1), h = lane / TNW;
  const int K = a.K, M = a.M;
  const int W = K / 32 * BITS, G = K / GRP;
  // this workgroup's slice of the k range (the whole row unless the launch splits k across workgroups)
  const int ksplit = a.ws ? a.ksplit : 1;
  const int nchunks = K / KC / ksplit, cbase = blockIdx.y * nchunks;
  const int steps = nchunks / 8, cpk = nchunks / KG;

  // segment of this workgroup's rows (uniform): a workgroup never straddles two matrices (launcher)
  const int nwg = blockIdx.x * (NG * TNW);
  const uint32_t* qw = a.qweight;
  const unsigned short* sc = a.scales;
  const unsigned short* ze = a.zeros;
  const unsigned short* bs = a.bias;
  unsigned short* yo = a.y;
  int nl0 = nwg, NL = a.N;
  const bool swiglu = a.swiglu != 0;               // (uniform) row groups [0, NG/2) = gate rows, [NG/2, NG) = the same rows of up
  int ngl = ng;                                     // row group inside the workgroup's row range of ONE matrix
  if (swiglu) {
    nl0 = blockIdx.x * (NG * TNW / 2);
    NL = a.n_seg[0];
    if (ng >= NG / 2) { qw = a.qweight_x[0]; sc = a.scales_x[0]; ze = a.zeros_x[0]; ngl = ng - NG / 2; }
  } else if (a.nseg > 1) {
    const int sg = (nwg >= a.n_seg[0]) + (nwg >= a.n_seg[0] + a.n_seg[1]);
    NL = a.n_seg[sg];
    if (sg > 0) {
      nl0 = nwg - a.n_seg[0] - (sg > 1 ? a.n_seg[1] : 0);
      qw = a.qweight_x[sg - 1]; sc = a.scales_x[sg - 1]; ze = a.zeros_x[sg - 1]; bs = a.bias_x[sg - 1]; yo = a.y_x[sg - 1];
    }
  }
  const int nrow = min(nl0 + ngl * TNW + r, NL - 1);
  const uint32_t* wrow = qw + static_cast<size_t>(nrow) * W + (IL ? 0 : h * WL) + static_cast<size_t>(cbase + kg * cpk) * WPC;
  const unsigned short* srow = sc + static_cast<size_t>(nrow) * G + (cbase + kg * cpk) * NF;
  const unsigned short* zrow = ze + static_cast<size_t>(nrow) * G + (cbase + kg * cpk) * NF;

  // loader role (waves 8..11): per step the workgroup stages 8 chunks x 32 rows x 256 B; loader thread -> 8 pieces of
  // 32 B (16 k-values), one per chunk slot
  const bool loader = wave >= 8;
  const int lrow = (tid & 255) >> 3, lpc = tid & 7;
  const unsigned short* xrow = a.x + static_cast<size_t>(min(lrow, M - 1)) * K + lpc * 16;
  const int xdst0 = lrow * XROW + lpc * 32;         // slab byte offset inside a chunk slot
  // M <= 16: only the first 16 activation rows are staged (loader waves 10, 11 idle, half the x traffic -- the term that
  // bounds this kernel) and the second 16-row tile is skipped; rows 16..31 of a 32x32 tile then multiply stale LDS
  // contents whose results are never stored
  const bool half_rows = M <= 16;
  const bool lact = !half_rows || lrow < 16;
  auto xoff = [&](int cs) { return (cbase + (cs / NCHW) * cpk + (cs % NCHW)) * KC; };   // element offset of slot cs at step 0

  f32x16 acc32[TNW == 32 ? 1 : 1];
  f32x4v acc16[TNW == 16 ? MT : 1];
  acc32[0] = f32x16{0};
#pragma unroll
  for (int i = 0; i < (TNW == 16 ? MT : 1); ++i) acc16[i] = f32x4v{0};

  unsigned wq[2][NCHW][WL];                        // packed words: current step, next step
  constexpr int SR = NCHW * NF >= 2 ? NCHW * NF / 2 : 1;   // dwords holding the step's NCHW * NF scales (zeros) of the lane's row
  unsigned scq[2][SR], zeq[2][SR];
  u32x4 xs[8][2];                                   // loader: activations of the step being staged
#ifdef BX_KM                                        // -DBX_KM=<bits>: knock-out builds (scratch/build_exp.sh), timing only: 1 no x loads, 2 no staging,
                                                    // 4 no operand build / MFMA / flush, 8 no weight loads, 16 x loads from one hot 2 KB, 32 loaders fetch the weights
#pragma unroll
  for (int i = 0; i < 8; ++i) { xs[i][0] = u32x4{0, 0, 0, 0}; xs[i][1] = u32x4{0, 0, 0, 0}; }
#pragma unroll
  for (int j = 0; j < NCHW; ++j)
#pragma unroll
    for (int q = 0; q < WL; ++q) wq[1][j][q] = 0x55555555u;
#pragma unroll
  for (int q = 0; q < SR; ++q) { scq[1][q] = 0x3C003C00u; zeq[1][q] = 0; }
#endif

  auto load_x = [&](int step) {
    if (!lact) return;
#if defined(BX_KM) && (BX_KM & 3)
    return;                                         // knock-out experiment: no activation loads
#endif
#pragma unroll
    for (int it = 0; it < 8; ++it) {
#if defined(BX_KM) && (BX_KM & 16)                   // timing only: every load from the same 2 KB (row 0, first 8 chunks)
      const unsigned short* p = a.x + lpc * 16 + (it & 7) * KC;
      xs[it][0] = *reinterpret_cast<const u32x4*>(p);
      xs[it][1] = *reinterpret_cast<const u32x4*>(p + 8);
#else
      const unsigned short* p = xrow + xoff(it) + step * (NCHW * KC);
      xs[it][0] = *reinterpret_cast<const u32x4*>(p);
      xs[it][1] = *reinterpret_cast<const u32x4*>(p + 8);
#endif
    }
  };
#if defined(BX_KM) && (BX_KM & 32)
  // experiment: the LOADER waves fetch the step's packed words (8 lanes per 128-byte line, 8 rows per instruction) and drop them;
  // the compute waves issue no loads at all -- what a kernel whose compute waves never touch the vector-memory path would cost
  u32x4 wl[8];
  auto load_wl = [&](int step) {
    const int lt = tid & 255;
    constexpr int LPR = NCHW * KG * WPC * 4 / 128;          // 128-byte lines per row and step
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int idx = it * 256 + lt;                          // piece of 16 B: [row][line][8 pieces]
      const int row = idx / (8 * LPR), pc = idx % (8 * LPR);
      if (row < NG * TNW) {
        const uint32_t* src = a.qweight + static_cast<size_t>(min(nwg + row, a.N - 1)) * W + static_cast<size_t>(cbase) * WPC + step * (8 * WPC) + pc * 4;
        wl[it] = *reinterpret_cast<const u32x4*>(src);
      }
    }
  };
  auto sink_wl = [&]() {
#pragma unroll
    for (int it = 0; it < 8; ++it) asm volatile("" :: "v"(wl[it]));
  };
#endif
  auto load_w = [&](unsigned (&wn)[NCHW][WL], unsigned (&sc_n)[SR], unsigned (&ze_n)[SR], int step) {
#if defined(BX_KM) && (BX_KM & 8)
    return;                                         // knock-out experiment: no weight loads
#endif
#pragma unroll
    for (int j = 0; j < NCHW; ++j) {
      const uint32_t* src = wrow + static_cast<size_t>(step * NCHW + j) * WPC;
      if constexpr (WL == 6) {                      // 3 bits, 32-row tiles: 24 bytes per lane and chunk (8-byte aligned)
#pragma unroll
        for (int q = 0; q < 6; q += 2) {
          const u32x2 v = *reinterpret_cast<const u32x2*>(src + q);
          wn[j][q] = v.x; wn[j][q + 1] = v.y;
        }
      } else if constexpr (WL == 3) {               // 3 bits, 16-row tiles: 12 bytes
        wn[j][0] = src[0]; wn[j][1] = src[1]; wn[j][2] = src[2];
      } else if constexpr (WL >= 4) {
#pragma unroll
        for (int q = 0; q < WL; q += 4) {
          const u32x4 v = *reinterpret_cast<const u32x4*>(src + q);
          wn[j][q] = v.x; wn[j][q + 1] = v.y; wn[j][q + 2] = v.z; wn[j][q + 3] = v.w;
        }
      } else {
        const u32x2 v = *reinterpret_cast<const u32x2*>(src);
        wn[j][0] = v.x; wn[j][1] = v.y;
      }
    }
    // the step's NCHW * NF consecutive scales / zeros of the row as wide loads (a load costs a cycle per distinct line)
    const unsigned short* sp = srow + step * (NCHW * NF);
    const unsigned short* zp = zrow + step * (NCHW * NF);
    if constexpr (SR >= 4) {
#pragma unroll
      for (int q = 0; q < SR; q += 4) {
        const u32x4 s4 = *reinterpret_cast<const u32x4*>(sp + 2 * q), z4 = *reinterpret_cast<const u32x4*>(zp + 2 * q);
        sc_n[q] = s4.x; sc_n[q + 1] = s4.y; sc_n[q + 2] = s4.z; sc_n[q + 3] = s4.w;
        ze_n[q] = z4.x; ze_n[q + 1] = z4.y; ze_n[q + 2] = z4.z; ze_n[q + 3] = z4.w;
      }
    } else if constexpr (SR == 2) {
      const u32x2 s2 = *reinterpret_cast<const u32x2*>(sp), z2 = *reinterpret_cast<const u32x2*>(zp);
      sc_n[0] = s2.x; sc_n[1] = s2.y; ze_n[0] = z2.x; ze_n[1] = z2.y;
    } else if constexpr (NCHW * NF == 2) {
      sc_n[0] = *reinterpret_cast<const unsigned*>(sp); ze_n[0] = *reinterpret_cast<const unsigned*>(zp);
    } else {
      sc_n[0] = sp[0]; ze_n[0] = zp[0];
    }
  };
  // registers -> pair order of the packed fields -> slab; xsum of the piece, reduced over the 8 lanes of a row chunk
  auto store_x = [&](int buf) {
    if (!lact) return;
#if defined(BX_KM) && (BX_KM & 2)
    return;                                         // knock-out experiment: no staging at all
#endif
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const u32x4 lo = xs[it][0], hi = xs[it][1];
      u32x4 u0, u1;
      if constexpr (BITS == 2) {                    // word of 16 fields: dword j of unit b = (x[4b+j], x[4b+j+8])
        u0.x = __builtin_amdgcn_perm(hi.x, lo.x, 0x05040100u); u0.y = __builtin_amdgcn_perm(hi.x, lo.x, 0x07060302u);
        u0.z = __builtin_amdgcn_perm(hi.y, lo.y, 0x05040100u); u0.w = __builtin_amdgcn_perm(hi.y, lo.y, 0x07060302u);
        u1.x = __builtin_amdgcn_perm(hi.z, lo.z, 0x05040100u); u1.y = __builtin_amdgcn_perm(hi.z, lo.z, 0x07060302u);
        u1.z = __builtin_amdgcn_perm(hi.w, lo.w, 0x05040100u); u1.w = __builtin_amdgcn_perm(hi.w, lo.w, 0x07060302u);
      } else if constexpr (BITS == 3) {             // fields leave a 24-bit window in adjacent pairs: natural order
        u0 = lo; u1 = hi;
      } else {                                      // word of 8 fields: dword j of its unit = (x[j], x[j+4])
        u0.x = __builtin_amdgcn_perm(lo.z, lo.x, 0x05040100u); u0.y = __builtin_amdgcn_perm(lo.z, lo.x, 0x07060302u);
        u0.z = __builtin_amdgcn_perm(lo.w, lo.y, 0x05040100u); u0.w = __builtin_amdgcn_perm(lo.w, lo.y, 0x07060302u);
        u1.x = __builtin_amdgcn_perm(hi.z, hi.x, 0x05040100u); u1.y = __builtin_amdgcn_perm(hi.z, hi.x, 0x07060302u);
        u1.z = __builtin_amdgcn_perm(hi.w, hi.y, 0x05040100u); u1.w = __builtin_amdgcn_perm(hi.w, hi.y, 0x07060302u);
      }
      unsigned char* dst = smem + buf * BUF + it * CHUNK + xdst0;
      *reinterpret_cast<u32x4*>(dst) = u0;
      *reinterpret_cast<u32x4*>(dst + 16) = u1;
      float sm = xdot2<BF16>(lo.x, 0.f);
      sm = xdot2<BF16>(lo.y, sm); sm = xdot2<BF16>(lo.z, sm); sm = xdot2<BF16>(lo.w, sm);
      sm = xdot2<BF16>(hi.x, sm); sm = xdot2<BF16>(hi.y, sm); sm = xdot2<BF16>(hi.z, sm); sm = xdot2<BF16>(hi.w, sm);
      sm += dpp_f32<0xB1>(sm);                      // 2 pieces = 32 k
      if constexpr (NF <= 2) sm += dpp_f32<0x4E>(sm);   // 4 pieces = 64 k
      if constexpr (NF == 1) sm += dpp_f32<0x141>(sm);  // 8 pieces: the row chunk
      if (lpc % (8 / NF) == 0) xsum[((buf * 8 + it) * NF + lpc / (8 / NF)) * 32 + lrow] = sm;
    }
  };
  auto compute = [&](const unsigned (&wc)[NCHW][WL], const unsigned (&sc_c)[SR], const unsigned (&ze_c)[SR], int buf) {
#if defined(BX_KM) && (BX_KM & 4)
    {                                               // knock-out experiment: loads stay, no MFMA / flush work
#pragma unroll
      for (int j = 0; j < NCHW; ++j)
#pragma unroll
        for (int q = 0; q < WL; ++q) asm volatile("" :: "v"(wc[j][q]));
      asm volatile("" :: "v"(sc_c[0]), "v"(ze_c[0]));
      return;
    }
#endif
#pragma unroll
    for (int j = 0; j < NCHW; ++j) {
      const int cs = kg * NCHW + j;
      const unsigned char* slab = smem + buf * BUF + cs * CHUNK;
      f32x16 g32;
      f32x4v g16[TNW == 16 ? MT : 1];
#pragma unroll
      for (int t = 0; t < UPL; ++t) {
        u32x4 bf;
        if constexpr (IL) {                         // 2 bits: unit t*H + h = half (h & 1) of word t (H = 2) / 2t + (h >> 1) (H = 4)
          unsigned w;
          if constexpr (H == 2) w = wc[j][t] >> (8 * h);
          else w = ((h >> 1) ? wc[j][2 * t + 1] : wc[j][2 * t]) >> (8 * (h & 1));
          bf.x = fld(w); bf.y = fld(w >> 2); bf.z = fld(w >> 4); bf.w = fld(w >> 6);
        } else if constexpr (BITS == 2) {
          const unsigned w = wc[j][t >> 1];
          const int sh = 8 * (t & 1);
          bf.x = fld(w >> sh); bf.y = fld(w >> (sh + 2)); bf.z = fld(w >> (sh + 4)); bf.w = fld(w >> (sh + 6));
        } else if constexpr (BITS == 3) {
          // unit t = the 24-bit window at bit 24 t of the lane's words; fields (2j, 2j+1) of it sit 3 bits apart: the low
          // field stays, the window shifted 13 bits up puts its neighbour on bit 16 of the same dword
          const int b0 = 24 * t, wi = b0 >> 5, sh = b0 & 31;       // constants after unrolling
          const unsigned win = sh == 0 ? wc[j][wi] : __builtin_amdgcn_alignbit(wc[j][wi + 1 < WL ? wi + 1 : WL - 1], wc[j][wi], sh);
          const unsigned p0 = win, p1 = win >> 6, p2 = win >> 12, p3 = win >> 18;
          bf.x = fld((p0 & 7u) | (p0 << 13)); bf.y = fld((p1 & 7u) | (p1 << 13));
          bf.z = fld((p2 & 7u) | (p2 << 13)); bf.w = fld((p3 & 7u) | (p3 << 13));
        } else {
          const unsigned w = wc[j][t];
          bf.x = fld(w); bf.y = fld(w >> 4); bf.z = fld(w >> 8); bf.w = fld(w >> 12);
        }
        const int aoff = (IL ? t * H + h : h * UPL + t) * 16;
        const bool first = t % SPG == 0;            // first step of a group: the accumulator tile restarts from zero
        if constexpr (TNW == 32) {
          const u32x4 af = *reinterpret_cast<const u32x4*>(slab + r * XROW + aoff);
          g32 = mfma<BF16>(af, bf, first ? f32x16{0} : g32);
        } else {
#pragma unroll
          for (int i = 0; i < MT; ++i) {
            if (i > 0 && half_rows) continue;       // (uniform)
            const u32x4 af = *reinterpret_cast<const u32x4*>(slab + (i * TNW + r) * XROW + aoff);
            g16[i] = mfma16<BF16>(af, bf, first ? f32x4v{0} : g16[i]);
          }
        }
        if (t % SPG == SPG - 1) {                   // group flush: acc += s * t + (z - C s) * xsum[row]
          const int gi = t / SPG, si = j * NF + gi;  // group inside the chunk; its scale / zero inside the step
          const unsigned sw = sc_c[si >> 1], zw = ze_c[si >> 1];
          const float s = act_to_f32<BF16>(static_cast<unsigned short>((si & 1) ? (sw >> 16) : (sw & 0xFFFFu)));
          const float zz = act_to_f32<BF16>(static_cast<unsigned short>((si & 1) ? (zw >> 16) : (zw & 0xFFFFu))) - CF * s;
          const float* xsm = xsum + ((buf * 8 + cs) * NF + gi) * 32;
          if constexpr (TNW == 32) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
              const f32x4v xv = *reinterpret_cast<const f32x4v*>(xsm + 8 * jj + 4 * (lane >> 5));
#pragma unroll
              for (int e = 0; e < 4; ++e) acc32[0][4 * jj + e] = fmaf(s, g32[4 * jj + e], fmaf(zz, xv[e], acc32[0][4 * jj + e]));
            }
          } else {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
              const f32x4v xv = *reinterpret_cast<const f32x4v*>(xsm + i * 16 + 4 * (lane >> 4));
#pragma unroll
              for (int e = 0; e < 4; ++e) acc16[i][e] = fmaf(s, g16[i][e], fmaf(zz, xv[e], acc16[i][e]));
            }
          }
        }
      }
    }
  };

  // Every workgroup walks its k range from a different starting step: at any moment the chip then reads all eight
  // 128-byte pieces of the 1 KB rows instead of the same piece of every row (HBM channel camping).
  const int rot = blockIdx.x % steps;
  auto kstep = [&](int i) { const int t = i + rot; return t >= steps ? t - steps : t; };
#ifdef BX_STAMPS
  long long st[24];
  int sti = 0;
#define STAMP() { if (sti < 24) st[sti++] = __builtin_amdgcn_s_memtime(); }
#else
#define STAMP()
#endif
  // Producer / consumer split: while the 8 compute waves run step s out of slab[s & 1], the 4 loader waves turn the
  // activations of step s+1 into slab[(s+1) & 1] (and have step s+2 in flight); one barrier per step.  With every wave
  // doing both, the load-issue phase (TA-bound, all waves at once) and the MFMA phase alternated instead of overlapping.
  // (two separate loops, not one loop with a role branch inside: the register allocator would keep both roles' state
  // live across the shared back edge.  Both execute steps+1 barriers.)
  STAMP();
  if (loader) {
    load_x(kstep(0));
#if defined(BX_KM) && (BX_KM & 32)
    load_wl(kstep(0));
#endif
    store_x(0);
#if defined(BX_KM) && (BX_KM & 32)
    sink_wl();
    if (steps > 1) load_wl(kstep(1));
#endif
    if (steps > 1) load_x(kstep(1));
    STAMP();
    __syncthreads();
    for (int step = 0; step < steps; ++step) {
#ifdef BX_STAMPS
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      STAMP();                                      // (stamps build) landed
      if (step + 1 < steps) {
        store_x((step & 1) ^ 1);                    // step+1: registers -> slab (+ xsum)
#ifdef BX_STAMPS
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#endif
        STAMP();                                    // stored
#if defined(BX_KM) && (BX_KM & 32)
        sink_wl();
        if (step + 2 < steps) load_wl(kstep(step + 2));
#endif
        if (step + 2 < steps) load_x(kstep(step + 2));
      } else { STAMP(); }
      STAMP();                                      // issued
      __syncthreads();                              // slab[buf ^ 1] complete, slab[buf] readers done
      STAMP();
    }
  } else {
    load_w(wq[1], scq[1], zeq[1], kstep(0));
    STAMP();
    __syncthreads();
    for (int step = 0; step < steps; ++step) {
#ifdef BX_STAMPS
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      STAMP();
#pragma unroll
      for (int j = 0; j < NCHW; ++j)
#pragma unroll
        for (int q = 0; q < WL; ++q) wq[0][j][q] = wq[1][j][q];
#pragma unroll
      for (int q = 0; q < SR; ++q) { scq[0][q] = scq[1][q]; zeq[0][q] = zeq[1][q]; }
#ifndef BX_PLAIN_FIELDS
      // fld() is inline assembly, and the compiler's s_waitcnt insertion does not cover a register whose FIRST consumer is an asm
      // operand (found on scratch/negative/mfma_batch2.hip: an unshifted word was read before its load had landed).  The only loads
      // outstanding here are this step's words, wanted now: wait for them by hand and tie the registers to the wait.
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < NCHW; ++j)
#pragma unroll
        for (int q = 0; q < WL; ++q) asm volatile("" : "+v"(wq[0][j][q]));
#endif
      if (step + 1 < steps) load_w(wq[1], scq[1], zeq[1], kstep(step + 1));
      STAMP();
      compute(wq[0], scq[0], zeq[0], step & 1);
#ifdef BX_STAMPS
      __builtin_amdgcn_sched_barrier(0);
#endif
      STAMP();
      __syncthreads();
      STAMP();
    }
  }
  __syncthreads();                                  // the slabs become the reduction buffer

  // ---- k-range reduction through LDS, one rounding to the activation dtype (+ bias as a second rounding) ----
  constexpr int TE = MT * AE;                       // accumulator elements per lane (16 / 8)
  if (!loader)
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int e = 0; e < AE; ++e) {
      float v;
      if constexpr (TNW == 32) v = acc32[0][e]; else v = acc16[i][e];
      red[((kg * NG + ng) * TE + i * AE + e) * 64 + lane] = v;
    }
  __syncthreads();
  if (swiglu) {
    if constexpr (NG >= 2) {
      for (int idx = tid; idx < (NG / 2) * TE * 64; idx += 768) {
        const int g = idx / (TE * 64), ie = (idx / 64) % TE, ln = idx & 63;
        float gt = 0.f, up = 0.f;
#pragma unroll
        for (int w = 0; w < KG; ++w) {
          gt += red[((w * NG + g) * TE + ie) * 64 + ln];
          up += red[((w * NG + g + NG / 2) * TE + ie) * 64 + ln];
        }
        const int i = ie / AE, e = ie % AE;
        const int n = nl0 + g * TNW + (ln & (TNW - 1));
        const int m = i * TNW + (TNW == 32 ? (e & 3) + 8 * (e >> 2) + 4 * (ln >> 5) : 4 * (ln >> 4) + e);
        if (n < NL && m < M) {       // activation-dtype gate, up, silu(gate), product: the unfused graph's four roundings
          const float g0 = act_to_f32<BF16>(f32_to_act<BF16>(gt)), u0 = act_to_f32<BF16>(f32_to_act<BF16>(up));
          const float sl = act_to_f32<BF16>(f32_to_act<BF16>(g0 / (1.f + __expf(-g0))));
          a.y[static_cast<size_t>(m) * NL + n] = f32_to_act<BF16>(sl * u0);
        }
      }
    }
    return;
  }
  for (int idx = tid; idx < NG * TE * 64; idx += 768) {
    const int g = idx / (TE * 64), ie = (idx / 64) % TE, ln = idx & 63;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < KG; ++w) v += red[((w * NG + g) * TE + ie) * 64 + ln];
    const int i = ie / AE, e = ie % AE;
    const int n = nl0 + g * TNW + (ln & (TNW - 1));
    // C/D layouts: 32x32: row = (e&3) + 8*(e>>2) + 4*(lane>>5); 16x16: row = 4*(lane>>4) + e
    const int m = i * TNW + (TNW == 32 ? (e & 3) + 8 * (e >> 2) + 4 * (ln >> 5) : 4 * (ln >> 4) + e);
    if (n < NL && m < M) {
      if (ksplit > 1) {                             // fp32 partial of this k slice (single-matrix launches only)
        a.ws[(static_cast<size_t>(blockIdx.y) * M + m) * NL + n] = v;
      } else {
        if (bs) v = act_to_f32<BF16>(f32_to_act<BF16>(v)) + act_to_f32<BF16>(bs[n]);
        yo[static_cast<size_t>(m) * NL + n] = f32_to_act<BF16>(v);
      }
    }
  }
#ifdef BX_STAMPS
  STAMP();
  if (lane == 0 && bx_stamp_ptr) {
    long long* o = bx_stamp_ptr + (static_cast<size_t>(blockIdx.x) * 12 + wave) * 24;
#pragma unroll
    for (int i = 0; i < 24; ++i) o[i] = i < sti ? st[i] : 0;
  }
#endif
}

template <int BITS, int TNW, int NG, int KG, int GRP>
static hipError_t launch_batch_grp(const MfmaArgs& a, bool bf16, hipStream_t s);

template <int BITS, int TNW, int NG, int KG>
static hipError_t launch_batch_cfg(const MfmaArgs& a, bool bf16, hipStream_t s) {
  if constexpr (BITS == 2) {
    if (a.log2g == 6) return launch_batch_grp<BITS, TNW, NG, KG, 64>(a, bf16, s);
    if (a.log2g == 5) return launch_batch_grp<BITS, TNW, NG, KG, 32>(a, bf16, s);
  }
  return launch_batch_grp<BITS, TNW, NG, KG, 128>(a, bf16, s);
}

// second launch of a k-split GEMM: y = round(sum of the slices) (+ bias as a second rounding); 4 outputs per thread
template <bool BF16>
__global__ __launch_bounds__(256) void batch_finish_kernel(const float* __restrict__ ws, const unsigned short* __restrict__ bias,
                                                           unsigned short* __restrict__ y, int slices, long long mn, int N) {
  const long long i = (blockIdx.x * 256LL + threadIdx.x) * 4;
  if (i >= mn) return;
  f32x4v v = *reinterpret_cast<const f32x4v*>(ws + i);
  for (int s = 1; s < slices; ++s) {
    const f32x4v u = *reinterpret_cast<const f32x4v*>(ws + s * mn + i);
    v += u;
  }
  unsigned short o[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float t = v[j];
    if (bias) t = act_to_f32<BF16>(f32_to_act<BF16>(t)) + act_to_f32<BF16>(bias[(i + j) % N]);
    o[j] = f32_to_act<BF16>(t);
  }
  *reinterpret_cast<u32x2*>(y + i) = u32x2{static_cast<unsigned>(o[0]) | (static_cast<unsigned>(o[1]) << 16),
                                          static_cast<unsigned>(o[2]) | (static_cast<unsigned>(o[3]) << 16)};
}

template <int BITS, int TNW, int NG, int KG, int GRP>
static hipError_t launch_batch_grp(const MfmaArgs& a, bool bf16, hipStream_t s) {
  constexpr int ROWS = NG * TNW;
  constexpr int SHMEM = 2 * 8 * 32 * XROW + 2 * 8 * (128 / GRP) * 32 * 4;
  static DeviceOnce done[2];
  if (done[bf16].pending()) {
    hipError_t e = bf16 ? hipFuncSetAttribute(reinterpret_cast<const void*>(&qgemm_batch_kernel<BITS, TNW, NG, KG, GRP, true>), hipFuncAttributeMaxDynamicSharedMemorySize, SHMEM)
                        : hipFuncSetAttribute(reinterpret_cast<const void*>(&qgemm_batch_kernel<BITS, TNW, NG, KG, GRP, false>), hipFuncAttributeMaxDynamicSharedMemorySize, SHMEM);
    if (e != hipSuccess) return e;
    done[bf16].done();
  }
  dim3 grid(a.swiglu ? (a.n_seg[0] + ROWS / 2 - 1) / (ROWS / 2) : (a.N + ROWS - 1) / ROWS, a.ws ? a.ksplit : 1), block(768);
#ifdef BX_STAMPS
  long long* sp = reinterpret_cast<long long*>(strtoull(getenv("GBX_STAMP_PTR") ? getenv("GBX_STAMP_PTR") : "0", nullptr, 0));
  (void)hipMemcpyToSymbol(HIP_SYMBOL(bx_stamp_ptr), &sp, sizeof(sp));
#endif
  if (bf16) hipLaunchKernelGGL((qgemm_batch_kernel<BITS, TNW, NG, KG, GRP, true>), grid, block, SHMEM, s, a);
  else hipLaunchKernelGGL((qgemm_batch_kernel<BITS, TNW, NG, KG, GRP, false>), grid, block, SHMEM, s, a);
  return hipGetLastError();
}

// k split across workgroups (needs the caller's fp32 workspace): worth it where the rows are long and few -- the down
// projection, N = 4096 x K = 14336 -- because every workgroup stages the activations of its whole k range: 16-row
// workgroups pull 256 x 917 KB of x through the CUs, 128-row workgroups x 7 slices pull 224 x 131 KB.
static int batch_ksplit(int N, int K, int nseg, bool partials = false) {
  static const int min_k = gbx::env_switch("GBX_BATCH_KSPLIT_MINK");   // tuning aid
  // with the summing launch folded into the consumer (partials) the split also pays on the short rows of o_proj: every
  // workgroup then stages 1/s of the activation tile -- the traffic that bounds this kernel
  if (nseg > 1 || N % 4 || N > 8192 || K < (partials ? 2048 : min_k)) return 1;
  const int tiles = (N + 127) / 128, nchunks = K / KC;
  int best = 1;
  for (int s = 2; s <= 16; ++s)
    if (nchunks % (8 * s) == 0 && tiles * s <= 256) best = s;
  return best;
}

size_t qgemm_batch_workspace_bytes(int M, int N, int K, int bits, int log2g) {
  if (log2g != 7 || !qgemm_batch_supported(M < 32 ? M : 32, K, bits, log2g)) return 0;   // (the split runs the 128-row shape)
  const int s = batch_ksplit(N, K, 1);
  return s > 1 ? static_cast<size_t>(s) * M * N * sizeof(float) : 0;
}

int qgemm_batch_slices(int M, int N, int K, int bits, int log2g) {
  if (log2g != 7 || !qgemm_batch_supported(M < 32 ? M : 32, K, bits, log2g)) return 1;
  return batch_ksplit(N, K, 1, true);
}

// workgroup rows: as many as still give ~200 workgroups (256 CUs, one workgroup each), and never across a segment
template <int BITS>
static hipError_t launch_batch_bits(const MfmaArgs& a0, bool bf16, hipStream_t s) {
  MfmaArgs a = a0;
  a.ksplit = (a.ws && a.log2g == 7) ? batch_ksplit(a.N, a.K, a.nseg, a.partials_only != 0) : 1;
  if (a.ksplit <= 1) a.ws = nullptr;
  if (a.ws) {                                       // 128-row workgroups x ksplit slices, then the summing launch
    // (few tiles even after the split -- o_proj: 32 x 4 -- : 64-row workgroups put one on every CU)
    const bool narrow = a.partials_only && ((a.N + 127) / 128) * a.ksplit <= 128 && a.N % 64 == 0;
#ifdef BX_WITH_V2
    hipError_t e = qgemm_batch2_ok(a, narrow ? 64 : 128) ? launch_qgemm_batch2(a, BITS, narrow ? 64 : 128, bf16, s)
                   : narrow ? launch_batch_cfg<BITS, 32, 2, 4>(a, bf16, s) : launch_batch_cfg<BITS, 32, 4, 2>(a, bf16, s);
#else
    hipError_t e = narrow ? launch_batch_cfg<BITS, 32, 2, 4>(a, bf16, s) : launch_batch_cfg<BITS, 32, 4, 2>(a, bf16, s);
#endif
    if (e != hipSuccess || a.partials_only) return e;
    return launch_splitk_finish(a.ws, a.bias, a.y, a.ksplit, a.M, a.N, bf16, s);
  }
  int rows = a.N >= 128 * 192 ? 128 : a.N >= 64 * 192 ? 64 : a.N >= 32 * 160 ? 32 : 16;
  static const int forced = gbx::env_switch("GBX_BATCH_ROWS");
  if (forced) rows = forced;
  if (a.log2g < 7 && rows == 128) rows = 64;        // small groups: twice the packed words per lane -- the 128-row shape spills
  if (a.swiglu) {                                   // half of a workgroup's rows come from each matrix: rows / 2 must tile n_seg[0]
    if (a.nseg != 2 || a.n_seg[0] != a.n_seg[1] || a.ws) return hipErrorInvalidValue;
    if (rows < 32) rows = 32;
    while (rows >= 32 && a.n_seg[0] % (rows / 2)) rows >>= 1;
    if (rows < 32) return hipErrorInvalidValue;
  } else
  for (int i = 0; i < (a.nseg > 1 ? a.nseg : 0); ++i)
    while (a.n_seg[i] % rows) rows >>= 1;
  if (rows < 16) return hipErrorInvalidValue;
#ifdef BX_WITH_V2
  if (qgemm_batch2_ok(a, rows)) return launch_qgemm_batch2(a, BITS, rows, bf16, s);
#endif
  switch (rows) {
    case 128: return launch_batch_cfg<BITS, 32, 4, 2>(a, bf16, s);
    case 64: return launch_batch_cfg<BITS, 32, 2, 4>(a, bf16, s);
    case 32: return launch_batch_cfg<BITS, 16, 2, 4>(a, bf16, s);
    default: return launch_batch_cfg<BITS, 16, 1, 8>(a, bf16, s);
  }
}

hipError_t launch_splitk_finish(const float* ws, const unsigned short* bias, unsigned short* y, int slices, int M, int N, bool bf16,
                                hipStream_t s) {
  const long long mn = static_cast<long long>(M) * N;       // N % 4 == 0 (callers)
  const unsigned blocks = static_cast<unsigned>((mn / 4 + 255) / 256);
  if (bf16) hipLaunchKernelGGL((batch_finish_kernel<true>), dim3(blocks), dim3(256), 0, s, ws, bias, y, slices, mn, N);
  else hipLaunchKernelGGL((batch_finish_kernel<false>), dim3(blocks), dim3(256), 0, s, ws, bias, y, slices, mn, N);
  return hipGetLastError();
}

bool qgemm_batch_supported(int M, int K, int bits, int log2g) {
  return M >= 1 && M <= 32 && bits >= 2 && bits <= 4 && (log2g == 7 || (bits == 2 && (log2g == 6 || log2g == 5))) && K % 1024 == 0;
}

hipError_t launch_qgemm_batch(const MfmaArgs& a, int bits, bool bf16, hipStream_t s) {
  return bits == 2 ? launch_batch_bits<2>(a, bf16, s) : bits == 3 ? launch_batch_bits<3>(a, bf16, s) : launch_batch_bits<4>(a, bf16, s);
}

}  // namespace gbx
