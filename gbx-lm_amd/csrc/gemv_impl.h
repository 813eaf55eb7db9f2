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
#include "gbx_env.h"
#include "gemv_fmt.h"
#include "kernels.h"
#include <cstdlib>

namespace gbx {

// One packed matrix of a launch.  A launch streams up to three matrices that share the activation vector, K, bits
// and group size (q/k/v, or gate/up), so the seven projections of a decoder block take four launches.
struct GemvSeg {
  const uint32_t* qweight;        // [N, K*bits/32]
  const unsigned short* scales;   // [N, K/g] (act dtype)
  const unsigned short* zeros;    // [N, K/g]
  unsigned short* y;              // [M, N]
  int N;
};

struct GemvArgs {
  const void* x;            // [M, K] activations (act dtype)
  GemvSeg seg[3];
  int nseg;
  int N;                    // total output rows over the segments (SwiGLU mode: rows of ONE segment)
  int K;
  int log2g;                // 5,6,7
  int batches_per_owner;    // multiple of the ring depth D
  const void* bias;         // [N] or null (single-segment launches; added by a second tiny kernel)
  // ---- fused decode-block options (all optional) ----
  const unsigned short* norm_w;    // RMSNorm weight [K]: x <- bf16(x * rsqrt(mean(x^2) + eps) * w)  (qllama.py:138-139)
  float eps;
  const unsigned short* residual;  // [N]: y <- residual + bf16(result)                              (qllama.py:138,140)
  int swiglu;                      // seg[0] = gate, seg[1] = up: y0[n] = silu(gate[n]) * up[n]       (qllama.py:115)
  // optional load-time companion of (scales, zeros) for the whole launch: one dword (scale bits | zero bits << 16) per
  // (row, group), the two rows of a batch interleaved -- [N/2][K/g][2] over the concatenated rows of the segments
  // (pairs (2p, 2p+1); every segment has an even N), or [N][K/g][2] = (gate row p, up row p) for a SwiGLU launch --
  // so a lane fetches both rows' scale and zero with ONE 8-byte load (gbx_pack_scales_zeros)
  const uint32_t* sz;
  // ---- expert-indexed launch (QuantizedSwitchLinear / mx.gather_qmm, switch_layers.py:78-92): blockIdx.y = (token, slot)
  // pair p; its activation row is x[p / x_div], its matrix is expert expert_idx[p] of the stacked [E, N, ...] tensors
  const int* expert_idx;           // [pairs] or null
  int x_div;                       // pairs that share one activation row (k for up/gate, 1 for down); >= 1
  int num_experts;
  long long e_stride_w, e_stride_g;   // elements between experts: N*W words, N*G scales
  // ---- activation vector formed in the prologue (launches behind gbx_attn_oproj_decode): x <- round(x + round(x_acc * 2^-32)),
  // x_acc = 64-bit fixed-point sums of the o_proj slices; workgroup 0 also writes the new x to x_out (the residual stream
  // of the block, qllama.py:138: h = x + r) -- never x itself, which the other workgroups are still reading
  const long long* x_acc;          // [K] or null
  unsigned short* x_out;           // [K] or null
  // ---- buffer this launch clears for a later one (the accumulator above, once its consumer has run) ----
  long long* zero_buf;
  int zero_n;
  // Uneven split (KS == 1 fat geometry): waves [0, long_waves) of every workgroup own batches_per_owner batches, the others one
  // less, so that a row count that is not a multiple of (resident workgroups x waves x 2) still puts one workgroup on EVERY CU
  // with equal work per SIMD (waves w, w+4, w+8, w+12 share a SIMD: long_waves is a multiple of 4).  0: every wave owns
  // batches_per_owner batches.
  int long_waves;
  int null_rows;                   // measurement only (GBX_CHAIN_FLOOR=2): every batch re-reads rows 0 / 1 -- the whole launch sequence with no weight stream
#ifdef GBX_STAMPS
  unsigned long long* stamps;      // diagnostic build only: [workgroup][wave][16] s_memrealtime stamps
#endif
};

// One batch = R rows as they sit in registers between issue and use.
constexpr int GEMV_R = 2;
template <int BITS, int P>
struct RowBatch {
  typename SlotWords<BITS>::type w[GEMV_R][P];
  unsigned short s[GEMV_R][P][2], z[GEMV_R][P][2];
  unsigned short res[GEMV_R];
  unsigned szp[P][2][GEMV_R];   // PK: [pass][sub-slot group][row] packed scale|zero dwords
  unsigned res2;            // PK: residual of both rows in one dword
};

// P  = 64-slot passes one lane makes over its part of a row.
// KS = wavefronts that share one row (1: a wave owns whole rows; 4: the workgroup's four waves each own a quarter of
//      K and their partial sums meet in LDS once, after the last row) -- long rows (down_proj) stay at P = 1 and
//      a small register footprint this way.
// D  = batches in flight per wave (ring of register buffers).  Every row owner gets a multiple of D batches and every
//      issue is unconditional, so hipcc can count vmcnt exactly: activations are waited for with the whole first
//      ring still in flight, and batch b+D is requested the moment batch b has been consumed.
// NW = wavefronts per workgroup.  The activation vector is converted ONCE per workgroup (through LDS), so fat
// workgroups (16 waves = one per CU-slot) make that prologue a small fraction of the row work even when every wave
// owns only one or two rows.
#ifndef GBX_FAT_NW_P1
#define GBX_FAT_NW_P1 16        // (experiment builds: -DGBX_FAT_NW_P1=8 -- two 8-wave workgroups per CU for gate|up: 799 -> 765 tok/s)
#endif
template <int P> struct WavesPerBlock { static constexpr int value = (P == 1) ? GBX_FAT_NW_P1 : 8; };   // "fat" geometry

// SPLIT (2-bit only): group_size 32, i.e. the two 32-value halves of a lane's slot carry different scales.
// MODE 1 (PK): scales/zeros come from the packed companion GemvArgs::sz (N of every segment even).
// MODE 2 (GATHER): expert-indexed launch (GemvArgs::expert_idx).  Compile-time so that the plain decode launches carry
// neither the expert-offset arithmetic nor its kernel arguments' loads (measured: 1.5-2 % of a token when done at run time).
// MODE 3 (PK + XACC): packed companion and the activation vector comes as residual stream + fixed-point accumulator
// (GemvArgs::x_acc): the launch that follows gbx_attn_oproj_decode.
// MODE 5: PK with exactly one plain / residual segment, fixed at compile time.
template <int BITS, int P, int KS, int D, bool BF16, bool SPLIT, int NW, int MODE>
__global__ __launch_bounds__(NW * 64) void gemv_kernel(const GemvArgs a) {
  constexpr bool PK = MODE == 1 || MODE == 3 || MODE == 5, GATHER = MODE == 2, XACC = MODE == 3;
  // MODE 5 (PK, ONE plain or residual segment known at compile time: o_proj, down_proj of the decode block): row location,
  // descriptor selects and the epilogue's runtime branches are ~200 scalar and ~20 vector instructions per 2-row batch otherwise,
  // and launches whose rows are a few thousand values long are bound by exactly that: 775 -> 794 tok/s (A/B, GBX_GEMV_CEPI).
  // (The same for the SwiGLU pair -- MODE 4 in the experiment -- measured slightly slower than the runtime form: 772 tok/s; not kept.)
  constexpr bool CSWI = false, CONE = MODE == 5;
  constexpr int NT = NW * 64;
  constexpr int GPB = NW / 4;                                 // KS == 4: row groups (of four K-quarter waves) per workgroup
  using F = Fmt<BITS>;
  using L = Lds<BITS>;
  using WT = typename SlotWords<BITS>::type;
  constexpr int PAIRS = L::PAIRS, SSTR = L::SSTR, NSUB = F::NSUB;
  constexpr int R = GEMV_R;
  constexpr int UPS = F::VPL / F::UNIT_VALS;                 // units per slot

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int K = a.K, N = a.N;
  const int nslots = K / F::VPL;
  h2* xp_lds = reinterpret_cast<h2*>(smem);
  float2* xs_lds = reinterpret_cast<float2*>(smem + static_cast<size_t>(nslots) * SSTR * 4);
  float* red_lds = reinterpret_cast<float*>(xs_lds + nslots * NSUB);    // NW floats (RMSNorm block reduction)
  float* part_lds = red_lds + NW;                                        // KS == 4: [batches of the block][R][4]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: row/segment math stays on the SALU
  const int W = K / 32 * BITS;                    // words per packed row
  const int G = K >> a.log2g;                     // groups per row
  const int m = blockIdx.y;                       // activation row (M > 1 re-streams the weights)
  int xrow = m;
  long long eo = 0;       // expert of this pair (wave-uniform scalar load; clamped so that a bad index cannot leave the tensors)
  if constexpr (GATHER) {
    xrow = m / a.x_div;
    eo = static_cast<long long>(min(max(a.expert_idx[m], 0), a.num_experts - 1));
  }
  const unsigned short* __restrict__ xg = static_cast<const unsigned short*>(a.x) + static_cast<size_t>(xrow) * K;
  const int gshift = a.log2g - 5;                 // 32-value sub-slot index -> group index
  const bool swiglu = CSWI ? true : (CONE ? false : a.swiglu != 0);
  const int rowstep = swiglu ? 1 : R;             // output rows covered by one batch
  const int bpo = a.batches_per_owner;
  const int kq = wave & 3, rg = wave >> 2;        // KS == 4: K-quarter of this wave, row group inside the workgroup
  // Owner o takes batches o, o + owners, o + 2*owners, ...: at any moment the whole chip streams ONE contiguous
  // region of the matrix (all HBM channels busy), instead of 4096 regions a fixed stride apart (channel hot spots
  // and a multi-microsecond straggler tail).
  const int owner = KS == 1 ? (blockIdx.x * NW + wave) : (blockIdx.x * GPB + rg);
  const int owners = static_cast<int>(gridDim.x) * (KS == 1 ? NW : GPB);
  // (uneven split: the last round belongs to the long waves only, packed densely; a short wave's last ring slot points past the
  // end -- its loads clamp to the last row, an L2 hit, and it is never consumed)
  const int long_waves = KS == 1 ? a.long_waves : 0;
  const int my_n = (long_waves && wave >= long_waves) ? bpo - 1 : bpo;
  auto batch_of = [&](int j) {
    if (KS == 1 && long_waves && j == bpo - 1)
      return wave < long_waves ? (bpo - 1) * owners + static_cast<int>(blockIdx.x) * long_waves + wave : (N + rowstep - 1) / rowstep;
    return owner + j * owners;
  };
  const int slot0 = (KS == 1 ? 0 : kq * 64) + lane;        // this lane's slot in pass 0
  constexpr int PSTRIDE = 64 * KS;                          // slot stride between passes
  const int n0 = a.seg[0].N, n01 = n0 + a.seg[1].N;

  // One buffer descriptor per segment tensor, held in SGPRs; a row is then addressed as
  //   descriptor (s_cselect by segment) + SGPR byte offset of the row + a per-lane VGPR offset that never changes,
  // so streaming a row costs no vector address arithmetic at all.  (Indexing the kernel-argument struct with a
  // computed index instead makes hipcc fetch it with VECTOR loads and drain vmcnt on every row.)
  constexpr int RS_FLAGS = 0x00020000, RS_ALL = 0x7FFFFFFF;
  auto rsrc = [](const void* ptr) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, RS_ALL, RS_FLAGS); };
  const uint32_t* const q0 = a.seg[0].qweight + (GATHER ? eo * a.e_stride_w : 0); const uint32_t* const q1 = a.seg[1].qweight + (GATHER ? eo * a.e_stride_w : 0); const uint32_t* const q2 = a.seg[2].qweight;
  const unsigned short* const s0 = a.seg[0].scales + (GATHER ? eo * a.e_stride_g : 0); const unsigned short* const s1 = a.seg[1].scales + (GATHER ? eo * a.e_stride_g : 0); const unsigned short* const s2 = a.seg[2].scales;
  const unsigned short* const z0 = a.seg[0].zeros + (GATHER ? eo * a.e_stride_g : 0); const unsigned short* const z1 = a.seg[1].zeros + (GATHER ? eo * a.e_stride_g : 0); const unsigned short* const z2 = a.seg[2].zeros;
  unsigned short* const y0 = a.seg[0].y; unsigned short* const y1 = a.seg[1].y; unsigned short* const y2 = a.seg[2].y;
  const int n1 = a.seg[1].N, n2 = a.seg[2].N;
  // row of batch element i -> (segment, local row); wave-uniform
  auto locate = [&](int r0, int i, int& sg, int& row) {
    if (swiglu) { sg = i; row = min(r0, N - 1); }
    else if (CONE) { sg = 0; row = min(r0 + i, N - 1); }
    else {
      const int r = min(r0 + i, N - 1);
      sg = (r >= n0) + (r >= n01);
      row = r - (sg == 0 ? 0 : (sg == 1 ? n0 : n01));
    }
  };
  const bool null_rows = a.null_rows != 0;
#ifdef GBX_STAMPS
#define GBX_STAMP(i) do { if (lane == 0 && a.stamps) a.stamps[(static_cast<size_t>(blockIdx.x) * NW + wave) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define GBX_STAMP(i) do {} while (0)
#endif
  // ---------------- 0. per-lane offsets and the small loads of a batch (scale|zero pairs, residual) ----------------
  // (Requesting the first batch's small loads AHEAD of the activation loads was measured 0.3-0.4 us slower per launch:
  // everything queued in front of x delays the prologue.)
  // Branch-free on purpose: a load in one arm and a register write in the other makes hipcc drain vmcnt at the join
  // (write-after-pending-load), which serialises the whole stream.  Out-of-range lanes / rows re-read the last valid
  // slot / row and are nulled through their zero activations or the store guard.
  const __amdgpu_buffer_rsrc_t resd = rsrc(a.residual ? static_cast<const void*>(a.residual) : static_cast<const void*>(y0));
  const __amdgpu_buffer_rsrc_t pd = rsrc(PK ? static_cast<const void*>(a.sz) : static_cast<const void*>(y0));
  int voff_w[P], voff_s[P];                        // per-lane byte offsets inside a row: packed words / scales
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int slot = min(p * PSTRIDE + slot0, nslots - 1);
    voff_w[p] = slot * (F::SLOT_WORDS * 4);
    voff_s[p] = ((slot * NSUB) >> gshift) * (PK ? 8 : 2);
  }
  auto issue_small = [&](RowBatch<BITS, P>& b, int r0) {     // PK: both rows' (scale|zero) and residual
    if constexpr (PK) {
      const int rc = min(r0, N - rowstep);                  // batches past the end re-read the last pair
      const int poff = (swiglu ? rc : (rc >> 1)) * (G * 8);
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const u32x2 v0 = __builtin_amdgcn_raw_buffer_load_b64(pd, voff_s[p], poff, 0);
        b.szp[p][0][0] = v0.x; b.szp[p][0][1] = v0.y;
        if constexpr (SPLIT) {
          const u32x2 v1 = __builtin_amdgcn_raw_buffer_load_b64(pd, voff_s[p] + 8, poff, 0);
          b.szp[p][1][0] = v1.x; b.szp[p][1][1] = v1.y;
        }
      }
      b.res2 = __builtin_amdgcn_raw_buffer_load_b32(resd, 0, (m * n0 + min(rc, n0 - 2)) * 2, 0);
    }
  };
  RowBatch<BITS, P> ring[D];
  GBX_STAMP(0);
#ifdef GBX_STAMPS
  if (lane == 0 && a.stamps)   // where the wave runs: XCC id << 32 | HW_ID (wave, simd, cu, sh, se)
    a.stamps[(static_cast<size_t>(blockIdx.x) * NW + wave) * 16 + 7] =
        (static_cast<unsigned long long>(__builtin_amdgcn_s_getreg((31 << 11) | 20)) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 4);
#endif
  // ---------------- 1. activation (and norm weight) loads first: vmcnt retires in order -------------------------
  // Prologue geometry (round 3): the conversion of x into the pre-shifted image is redundant per workgroup and bound by
  // the CU's vector ALUs (~10 instructions per value; stamps: 0.5-0.9 us with one 16-value unit per thread and three
  // quarters of a fat workgroup's threads idle), so a thread takes PP PAIRS of one unit -- as few as keeps every thread
  // of the workgroup busy (2 pairs = 4 values in the 16-wave geometry at K = 4096) -- and the per-32-value statistics meet
  // through DPP inside the 16 / PP neighbouring lanes.  3 bits keeps whole units (its pairs are not contiguous).
  constexpr int UP = F::UNIT_PAIRS;
  constexpr int PAIRS_MAX = P * KS * 64 * (F::VPL / 2);
  constexpr int PPQ = (PAIRS_MAX + NT - 1) / NT;
  constexpr int PP = (BITS == 3) ? UP : (PPQ <= 2 ? 2 : (PPQ <= 4 ? 4 : UP));     // pairs per thread and unit (<= UP: 4-bit units have 4)
  static_assert(PP <= UP || BITS == 3, "pairs per thread");
  constexpr int PPU = PP < UP ? PP : UP;
  constexpr int TPU = UP / PPU;                               // threads per unit
  constexpr int UPT = (PAIRS_MAX + NT * PPU - 1) / (NT * PPU);   // (unit, part) items per thread
  constexpr int GL = (BITS == 3) ? 1 : 16 / PPU;              // lanes that share one 32-value sub-slot
  constexpr int XW = PPU;                                     // 32-bit words of activations per item (2 * PPU values)
  const int nunits = K / F::UNIT_VALS;
  const int nitems = nunits * TPU;
  // fused RMSNorm only where a wave owns whole rows (KS == 1: K = hidden size); the launcher rejects it otherwise
  const bool norm = (KS == 1) && a.norm_w != nullptr;
  const unsigned short* __restrict__ wg = norm ? a.norm_w : xg;       // dummy source keeps the loads unconditional
  constexpr int WPT = (KS == 1) ? UPT : 1;
  unsigned xr[UPT][XW], wr[WPT][XW];
  auto load_item = [&](const unsigned short* base, int it, unsigned* out) {
    const int u = it / TPU, t = it % TPU;
    if constexpr (BITS == 3) {                                // the whole 32-value unit: four 16-byte loads
      const u32x4* src = reinterpret_cast<const u32x4*>(base + static_cast<size_t>(u) * F::UNIT_VALS);
#pragma unroll
      for (int q = 0; q < 4; ++q) { const u32x4 v = src[q]; out[4 * q] = v.x; out[4 * q + 1] = v.y; out[4 * q + 2] = v.z; out[4 * q + 3] = v.w; }
    } else {                                                  // values [PPU t, +PPU) and [UP + PPU t, +PPU) of the unit
      const unsigned short* pa = base + static_cast<size_t>(u) * F::UNIT_VALS + PPU * t;
      const unsigned short* pb = pa + UP;
      if constexpr (PPU == 2) {
        out[0] = *reinterpret_cast<const unsigned*>(pa);
        out[1] = *reinterpret_cast<const unsigned*>(pb);
      } else if constexpr (PPU == 4) {
        const u32x2 va = *reinterpret_cast<const u32x2*>(pa), vb = *reinterpret_cast<const u32x2*>(pb);
        out[0] = va.x; out[1] = va.y; out[2] = vb.x; out[3] = vb.y;
      } else {
        const u32x4 va = *reinterpret_cast<const u32x4*>(pa), vb = *reinterpret_cast<const u32x4*>(pb);
        out[0] = va.x; out[1] = va.y; out[2] = va.z; out[3] = va.w; out[4] = vb.x; out[5] = vb.y; out[6] = vb.z; out[7] = vb.w;
      }
    }
  };
#pragma unroll
  for (int j = 0; j < UPT; ++j) {
    const int it = min(tid + NT * j, nitems - 1);
    load_item(xg, it, xr[j]);
    if constexpr (KS == 1) load_item(wg, it, wr[j]);
  }
  // XACC: the accumulator entries of the same values (two per 16-byte load), in the order of the item's registers
  long long xacc[XACC ? UPT : 1][XACC ? 2 * XW : 1];
  if constexpr (XACC) {
#pragma unroll
    for (int j = 0; j < UPT; ++j) {
      const int it = min(tid + NT * j, nitems - 1);
      const int u = it / TPU, t = it % TPU;
#pragma unroll
      for (int q = 0; q < XW; ++q) {          // word q of the item holds values (2q, 2q + 1) of its a-part / b-part (or of the unit at 3 bits)
        int k0;
        if constexpr (BITS == 3) k0 = u * F::UNIT_VALS + 2 * q;
        else k0 = u * F::UNIT_VALS + (q < PPU / 2 ? PPU * t + 2 * q : UP + PPU * t + 2 * (q - PPU / 2));
        const u32x4 v = *reinterpret_cast<const u32x4*>(a.x_acc + k0);
        xacc[j][2 * q] = static_cast<long long>((static_cast<unsigned long long>(v.y) << 32) | v.x);
        xacc[j][2 * q + 1] = static_cast<long long>((static_cast<unsigned long long>(v.w) << 32) | v.z);
      }
    }
  }
  if (a.zero_buf) {                          // wave-uniform; a handful of 8-byte stores per thread at most
    for (int i = blockIdx.x * NT + tid; i < a.zero_n; i += static_cast<int>(gridDim.x) * NT) a.zero_buf[i] = 0;
  }

  // ---------------- 2. the first batch of packed rows goes in flight before any activation math ---------------
  auto issue_rows = [&](RowBatch<BITS, P>& b, int r0) {
#pragma unroll
    for (int i = 0; i < R; ++i) {
      int sgi, row;
      locate(r0, i, sgi, row);
      // pointer selects stay on the SALU; only descriptor words 0-1 differ between segments
      const __amdgpu_buffer_rsrc_t qd = rsrc(sgi == 0 ? q0 : (sgi == 1 ? q1 : q2));
      const __amdgpu_buffer_rsrc_t sd = rsrc(sgi == 0 ? s0 : (sgi == 1 ? s1 : s2));
      const __amdgpu_buffer_rsrc_t zd = rsrc(sgi == 0 ? z0 : (sgi == 1 ? z1 : z2));
      const int woff = null_rows ? (row & 1) * (W * 4) : row * (W * 4), goff = row * (G * 2);
#pragma unroll
      for (int p = 0; p < P; ++p) {
        if constexpr (BITS == 3) b.w[i][p] = __builtin_amdgcn_raw_buffer_load_b96(qd, voff_w[p], woff, 2);   // nt
        else b.w[i][p] = __builtin_amdgcn_raw_buffer_load_b128(qd, voff_w[p], woff, 2);
        if constexpr (!PK) {
          b.s[i][p][0] = __builtin_amdgcn_raw_buffer_load_b16(sd, voff_s[p], goff, 0);
          b.z[i][p][0] = __builtin_amdgcn_raw_buffer_load_b16(zd, voff_s[p], goff, 0);
          if constexpr (SPLIT) {       // second 32-value half has its own group only when g == 32 (2-bit)
            b.s[i][p][1] = __builtin_amdgcn_raw_buffer_load_b16(sd, voff_s[p] + 2, goff, 0);
            b.z[i][p][1] = __builtin_amdgcn_raw_buffer_load_b16(zd, voff_s[p] + 2, goff, 0);
          }
        }
      }
      if constexpr (!PK) b.res[i] = __builtin_amdgcn_raw_buffer_load_b16(resd, 0, (m * n0 + min(row, n0 - 1)) * 2, 0);
    }
  };
  // Only the FIRST batch is requested ahead of the activation prologue: a wave issues in order, and a deep burst of
  // row loads here would hold the wave (TA back-pressure) for microseconds before it could even start converting x --
  // the whole chip would then stream first and compute afterwards instead of overlapping the two.
  auto issue = [&](RowBatch<BITS, P>& b, int r0) { issue_rows(b, r0); issue_small(b, r0); };
  issue(ring[0], batch_of(0) * rowstep);

  GBX_STAMP(1);
  // ---------------- 3.0 XACC: x <- round(x + round(acc * 2^-32)) (the o_proj epilogue `h + r` of qllama.py:138) ------
  if constexpr (XACC) {
#pragma unroll
    for (int j = 0; j < UPT; ++j) {
#pragma unroll
      for (int q = 0; q < XW; ++q) {
        const float r0 = round_bf16_or_f16(static_cast<float>(xacc[j][2 * q]) * (1.f / 4294967296.f), BF16);
        const float r1 = round_bf16_or_f16(static_cast<float>(xacc[j][2 * q + 1]) * (1.f / 4294967296.f), BF16);
        const unsigned lo = f32_to_act<BF16>(act_lo<BF16>(xr[j][q]) + r0), hi = f32_to_act<BF16>(act_hi<BF16>(xr[j][q]) + r1);
        xr[j][q] = lo | (hi << 16);
      }
      const int it = tid + NT * j;
      if (blockIdx.x == 0 && a.x_out && it < nitems) {        // the new residual stream, once
        const int u = it / TPU, t = it % TPU;
        if constexpr (BITS == 3) {
          u32x4* dst = reinterpret_cast<u32x4*>(a.x_out + static_cast<size_t>(u) * F::UNIT_VALS);
#pragma unroll
          for (int q = 0; q < 4; ++q) dst[q] = u32x4{xr[j][4 * q], xr[j][4 * q + 1], xr[j][4 * q + 2], xr[j][4 * q + 3]};
        } else {
          unsigned* da = reinterpret_cast<unsigned*>(a.x_out + static_cast<size_t>(u) * F::UNIT_VALS + PPU * t);
          unsigned* db = reinterpret_cast<unsigned*>(a.x_out + static_cast<size_t>(u) * F::UNIT_VALS + UP + PPU * t);
#pragma unroll
          for (int q = 0; q < PPU / 2; ++q) { da[q] = xr[j][q]; db[q] = xr[j][PPU / 2 + q]; }
        }
      }
    }
  }

  // ---------------- 3a. optional fused RMSNorm: block-wide mean of squares ---------------------------------------
  float rstd = 1.f;
#ifdef GBX_STAMPS
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"(R * P + (PK ? P + 1 : 0)) : "memory");   // x (and w) have landed (diagnostic build only)
  GBX_STAMP(8);
#endif
  if (norm) {
    float ssq = 0.f;
#pragma unroll
    for (int j = 0; j < UPT; ++j) {
      const bool live = (tid + NT * j) < nitems;
#pragma unroll
      for (int q = 0; q < XW; ++q) {
        const float lo = act_lo<BF16>(xr[j][q]), hi = act_hi<BF16>(xr[j][q]);
        ssq += live ? fmaf(hi, hi, lo * lo) : 0.f;   // explicit: -ffp-contract=on may pick either product for the fma, per instantiation
      }
    }
    ssq = wave_sum(ssq);
    if (lane == 0) red_lds[wave] = ssq;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) tot += red_lds[i];
    rstd = rsqrtf(tot / static_cast<float>(K) + a.eps);
  }
  GBX_STAMP(9);

  // ---------------- 3b. activations -> pre-shifted fp16 pairs, one block exponent per 32-value sub-slot ----------
#pragma unroll
  for (int j = 0; j < UPT; ++j) {
    // waves whose items all lie past the end skip the arithmetic (wave-uniform): in a fat workgroup they would only compete
    // with the live waves for the vector ALUs
    if ((tid & ~63) + NT * j < nitems) {
      const int it = tid + NT * j;
      const int u = it / TPU, t = it % TPU;
      // value of pair i: low half xa[i] (field ka), high half xb[i] (field kb)
      float xa[PPU], xb[PPU];
      unsigned amax = 0;
      float usum = 0.f;
#pragma unroll
      for (int i = 0; i < PPU; ++i) {
        int wa, ha, wb, hb;                                     // (word, half) of the two values inside the item's registers
        if constexpr (BITS == 3) { wa = F::ka(i) >> 1; ha = F::ka(i) & 1; wb = F::kb(i) >> 1; hb = F::kb(i) & 1; }
        else { wa = i >> 1; ha = i & 1; wb = (PPU >> 1) + (i >> 1); hb = i & 1; }
        float va = ha ? act_hi<BF16>(xr[j][wa]) : act_lo<BF16>(xr[j][wa]);
        float vb = hb ? act_hi<BF16>(xr[j][wb]) : act_lo<BF16>(xr[j][wb]);
        if constexpr (KS == 1) {
          if (norm) {   // one rounding to the activation dtype, like the reference's RMSNorm output
            va = round_bf16_or_f16(va * rstd * (ha ? act_hi<BF16>(wr[j][wa]) : act_lo<BF16>(wr[j][wa])), BF16);
            vb = round_bf16_or_f16(vb * rstd * (hb ? act_hi<BF16>(wr[j][wb]) : act_lo<BF16>(wr[j][wb])), BF16);
          }
        }
        xa[i] = va;
        xb[i] = vb;
        usum += va + vb;
        amax = max(amax, max(__float_as_uint(va) & 0x7FFFFFFFu, __float_as_uint(vb) & 0x7FFFFFFFu));
      }
      // sub-slot statistics over the GL lanes that share it (every lane of a live wave takes part; items past the end hold
      // copies of the last item and only ever meet each other or are never stored)
      float ssum = usum;
      if constexpr (GL >= 2) { ssum += dpp_f32<0xB1>(ssum); amax = max(amax, dpp_u32<0xB1>(amax)); }
      if constexpr (GL >= 4) { ssum += dpp_f32<0x4E>(ssum); amax = max(amax, dpp_u32<0x4E>(amax)); }
      if constexpr (GL >= 8) { ssum += dpp_f32<0x141>(ssum); amax = max(amax, dpp_u32<0x141>(amax)); }
      // largest |x| of the sub-slot lands in [2^14, 2^15) as fp16; products with q*2^(pos-24) stay exact
      int E = 141 - static_cast<int>(amax >> 23);
      E = min(max(E, -100), 110);
      if (it < nitems) {
        const int slot = u / UPS, uin = u % UPS;
        unsigned* dst = reinterpret_cast<unsigned*>(xp_lds + static_cast<size_t>(slot) * SSTR + uin * UP + PPU * t);
        unsigned img[PPU];
#pragma unroll
        for (int i = 0; i < PPU; ++i) {
          int ps;                                               // bit position of pair (PPU t + i) of the unit
          if constexpr (BITS == 3) ps = F::pos(i);
          else if constexpr (BITS == 2) { const int pi = PPU * t + i; ps = 2 * (pi < 5 ? pi : pi - 5); }
          else { ps = ((PPU * t + i) & 1) ? 4 : 0; }
          const float sc = exp2i(E - ps);
          h2 v;
          v.x = static_cast<_Float16>(xa[i] * sc);
          v.y = static_cast<_Float16>(xb[i] * sc);
          img[i] = as_u32(v);
        }
        if constexpr (PPU == 2) *reinterpret_cast<u32x2*>(dst) = u32x2{img[0], img[1]};
        else {
#pragma unroll
          for (int i = 0; i < PPU; i += 4) *reinterpret_cast<u32x4*>(dst + i) = u32x4{img[i], img[i + 1], img[i + 2], img[i + 3]};
        }
        if ((it % GL) == 0) xs_lds[it / GL] = make_float2(ssum, exp2i(24 - E));
      }
    }
  }
  GBX_STAMP(10);
  __syncthreads();

  GBX_STAMP(2);
  // ---------------- 4. lane-resident activations (zeros for lanes past the end of the row) ------------------------
  h2 xv[P][PAIRS];
  float xs[P][NSUB], cf[P][NSUB];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int slot = p * PSTRIDE + slot0;
    const bool live = slot < nslots;
    const int sl = min(slot, nslots - 1);
    const u32x4* src = reinterpret_cast<const u32x4*>(xp_lds + static_cast<size_t>(sl) * SSTR);
#pragma unroll
    for (int q = 0; q < PAIRS / 4; ++q) {
      const u32x4 v = src[q];
      xv[p][4 * q + 0] = as_h2(live ? v.x : 0u);
      xv[p][4 * q + 1] = as_h2(live ? v.y : 0u);
      xv[p][4 * q + 2] = as_h2(live ? v.z : 0u);
      xv[p][4 * q + 3] = as_h2(live ? v.w : 0u);
    }
#pragma unroll
    for (int s = 0; s < NSUB; ++s) {
      const float2 t = xs_lds[sl * NSUB + s];
      xs[p][s] = live ? t.x : 0.f;
      cf[p][s] = live ? t.y : 0.f;
    }
  }

  // ---------------- 5. epilogue of one batch: the two row totals are wave-uniform here ---------------------------
  auto finish = [&](int r0, float t0, float t1, unsigned short res0, unsigned short res1) {
    if (swiglu) {
      if (r0 < N) {     // bf16 gate, bf16 up, bf16 silu(gate), bf16 product: the reference's op-by-op roundings
        const float g = round_bf16_or_f16(t0, BF16), up = round_bf16_or_f16(t1, BF16);
        const float sl = round_bf16_or_f16(g / (1.f + __expf(-g)), BF16);
        y0[static_cast<size_t>(m) * n0 + r0] = f32_to_act<BF16>(sl * up);
      }
      return;
    }
    const float tv[2] = {t0, t1};
    const unsigned short rv[2] = {res0, res1};
#pragma unroll
    for (int i = 0; i < R; ++i) {
      if (r0 + i < N) {
        int sgi, row;
        locate(r0, i, sgi, row);
        float v = tv[i];
        if (a.residual) v = act_to_f32<BF16>(rv[i]) + round_bf16_or_f16(v, BF16);
        unsigned short* ysel = sgi == 0 ? y0 : (sgi == 1 ? y1 : y2);
        const int nsel = sgi == 0 ? n0 : (sgi == 1 ? n1 : n2);
        ysel[static_cast<size_t>(m) * nsel + row] = f32_to_act<BF16>(v);
      }
    }
  };

  // ---------------- 6. stream the rows: consume ring slot d, refill it D batches ahead ----------------------------
  auto consume = [&](const RowBatch<BITS, P>& cur, int j) {
    const int b = batch_of(j);
    float acc[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
      acc[i] = 0.f;
#pragma unroll
      for (int p = 0; p < P; ++p) {
        float d[NSUB];
        dot_slot<BITS>(cur.w[i][p], xv[p], d);
        auto scale_of = [&](int s) { return PK ? sz_scale<BF16>(cur.szp[p][s][i]) : act_to_f32<BF16>(cur.s[i][p][s]); };
        auto zero_of = [&](int s) { return PK ? sz_zero<BF16>(cur.szp[p][s][i]) : act_to_f32<BF16>(cur.z[i][p][s]); };
        if constexpr (NSUB == 2 && !SPLIT) {     // both halves share one (scale, zero): s*(c0*d0 + c1*d1) + z*(xs0 + xs1)
          const float t = fmaf(cf[p][1], d[1], cf[p][0] * d[0]);
          acc[i] = fmaf(scale_of(0), t, acc[i]);
          acc[i] = fmaf(zero_of(0), xs[p][0] + xs[p][1], acc[i]);
        } else {
#pragma unroll
          for (int s = 0; s < NSUB; ++s) {
            const float sc = scale_of(s) * cf[p][s];
            const float ze = zero_of(s);
            acc[i] = fmaf(sc, d[s], acc[i]);
            acc[i] = fmaf(ze, xs[p][s], acc[i]);
          }
        }
      }
    }
    const float tot = wave_sum2(acc[0], acc[1]);            // batch row 0 total in lane 31, row 1 total in lane 63
    const float t0 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(tot), 31));
    const float t1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(tot), 63));
    if constexpr (KS == 1) {
      // (measured, round 3: parking the totals in lane j and running ONE vector epilogue for all of the wave's batches at the end
      // is slower -- 776 -> 764 tok/s: it moves the epilogues from under the stream onto the launch's tail)
      if (lane == 0) {
        if constexpr (PK) finish(b * rowstep, t0, t1, static_cast<unsigned short>(cur.res2 & 0xFFFFu), static_cast<unsigned short>(cur.res2 >> 16));
        else finish(b * rowstep, t0, t1, cur.res[0], cur.res[1]);
      }
    } else {
      if (lane == 0) {
        float* pp = part_lds + static_cast<size_t>(rg * bpo + j) * (R * 4) + kq;
        pp[0] = t0;
        pp[4] = t1;
      }
    }
  };
#pragma unroll
  for (int d = 1; d < D; ++d) issue(ring[d], batch_of(d) * rowstep);
  GBX_STAMP(3);
  const int steps = bpo / D;
  int base = 0;
  for (int st = 1; st < steps; ++st) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      consume(ring[d], base + d);
      issue(ring[d], batch_of(base + D + d) * rowstep);
    }
    base += D;
  }
  GBX_STAMP(4);
#pragma unroll
  for (int d = 0; d < D; ++d) {
    if (base + d < my_n) consume(ring[d], base + d);      // (wave-uniform; no loads behind this branch)
    if (d == 0) GBX_STAMP(5);
  }
  GBX_STAMP(6);

  if constexpr (KS == 4) {
    __syncthreads();
    for (int bl = tid; bl < GPB * bpo; bl += NT) {
      const float4 p0 = *reinterpret_cast<const float4*>(part_lds + static_cast<size_t>(bl) * (R * 4));
      const float4 p1 = *reinterpret_cast<const float4*>(part_lds + static_cast<size_t>(bl) * (R * 4) + 4);
      const int r0 = ((blockIdx.x * GPB + bl / bpo) + (bl % bpo) * owners) * rowstep;
      unsigned short res0 = 0, res1 = 0;
      if (a.residual && !swiglu) {
        if (r0 < N) res0 = a.residual[static_cast<size_t>(m) * n0 + r0];
        if (r0 + 1 < N) res1 = a.residual[static_cast<size_t>(m) * n0 + r0 + 1];
      }
      finish(r0, (p0.x + p0.y) + (p0.z + p0.w), (p1.x + p1.y) + (p1.z + p1.w), res0, res1);
    }
  }
}

// y[m, n] += bias[n] (second rounding, like the reference's separate `x + self.bias`, quantized_linear_gba.py:204-205)
template <bool BF16>
__global__ __launch_bounds__(256) void add_bias_kernel(unsigned short* __restrict__ y, const unsigned short* __restrict__ bias,
                                                       int N, size_t total, const int* __restrict__ expert_idx, int num_experts) {
  for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < total;
       i += static_cast<size_t>(gridDim.x) * blockDim.x) {
    const size_t e = expert_idx ? static_cast<size_t>(min(max(expert_idx[i / N], 0), num_experts - 1)) : 0;   // bias[indices] (switch_layers.py:90-91)
    y[i] = f32_to_act<BF16>(act_to_f32<BF16>(y[i]) + act_to_f32<BF16>(bias[e * N + i % N]));
  }
}

// Host-side launcher for one bit width (defined per TU so the three widths compile in parallel).
template <int BITS>
hipError_t launch_gemv(const GemvArgs& a, int M, bool bf16, hipStream_t stream);

constexpr int GEMV_MAX_BATCHES_PER_GROUP_KS4 = 256;
constexpr int GEMV_CUS = 256;                  // MI355X

// Workgroups of this kernel variant that are resident at once on the whole chip (asked once, then cached).
template <int BITS, int P, int KS, int D, bool SPLIT, int NW, int PK>
static int resident_blocks(bool bf16, size_t shmem) {
  static int cached[2] = {0, 0};
  int& c = cached[bf16 ? 1 : 0];
  if (c == 0) {
    constexpr int NT = NW * 64;
    int per_cu = 0;
    hipError_t e = bf16 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gemv_kernel<BITS, P, KS, D, true, SPLIT, NW, PK>, NT, shmem)
                        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gemv_kernel<BITS, P, KS, D, false, SPLIT, NW, PK>, NT, shmem);
    if (e != hipSuccess || per_cu < 1) per_cu = 1;
    if (per_cu * NW > 32) per_cu = 32 / NW;
    c = per_cu * GEMV_CUS;
  }
  return c;
}

template <int BITS, int P, int KS, int NW>
static size_t gemv_shmem(int K, int bpo) {
  size_t b = Lds<BITS>::bytes(K) + NW * sizeof(float) + 16;
  if (KS == 4) b += static_cast<size_t>(NW / 4) * bpo * GEMV_R * 4 * sizeof(float);
  return b;
}

template <int BITS, int P, int KS, int D, bool SPLIT, int NW, int PK>
static hipError_t launch_gemv_d(const GemvArgs& a, int M, bool bf16, int bpo, int nbatches, hipStream_t stream) {
  GemvArgs args = a;
  args.batches_per_owner = bpo;
  const size_t shmem = gemv_shmem<BITS, P, KS, NW>(a.K, bpo);
  const int batches_per_block = (KS == 1 ? NW : NW / 4) * bpo;
  const int blocks = (nbatches + batches_per_block - 1) / batches_per_block;
  // GBX_CHAIN_FLOOR (measurement only, wrong results): 1 = the same grid, every wave stops after ONE ring round; 2 = every batch
  // re-reads rows 0 / 1 (same instruction stream, no weight stream from HBM) -- what is left of
  // a decode step is its dependency chain (launch boundaries, activation prologues, reductions) with next to no weight stream
  static const int chain_floor = gbx::env_switch("GBX_CHAIN_FLOOR");
  if (chain_floor == 1) { args.batches_per_owner = D; args.long_waves = 0; }
  if (chain_floor == 2) args.null_rows = 1;
  dim3 grid(blocks, M), block(NW * 64);
  if (bf16) hipLaunchKernelGGL((gemv_kernel<BITS, P, KS, D, true, SPLIT, NW, PK>), grid, block, shmem, stream, args);
  else hipLaunchKernelGGL((gemv_kernel<BITS, P, KS, D, false, SPLIT, NW, PK>), grid, block, shmem, stream, args);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess && a.bias) {
    const size_t total = static_cast<size_t>(M) * a.N;
    const int bb = static_cast<int>(std::min<size_t>((total + 255) / 256, 2048));
    auto B = static_cast<const unsigned short*>(a.bias);
    if (bf16) hipLaunchKernelGGL((add_bias_kernel<true>), dim3(bb), dim3(256), 0, stream, a.seg[0].y, B, a.N, total, a.expert_idx, a.num_experts);
    else hipLaunchKernelGGL((add_bias_kernel<false>), dim3(bb), dim3(256), 0, stream, a.seg[0].y, B, a.N, total, a.expert_idx, a.num_experts);
    e = hipGetLastError();
  }
  return e;
}

// One round of co-resident workgroups covers all rows: pick the shallowest ring whose resident owners can hold the
// batches, else split the batches evenly over the resident owners of the deepest ring (multiples of D per owner).
template <int BITS, int P, int KS, bool SPLIT, int NW, int PK>
static hipError_t launch_gemv_nw(const GemvArgs& a, int M, bool bf16, int nbatches, hipStream_t stream) {
  const int opb = (KS == 1) ? NW : NW / 4;                 // row owners per workgroup
  const int own1 = resident_blocks<BITS, P, KS, 1, SPLIT, NW, PK>(bf16, gemv_shmem<BITS, P, KS, NW>(a.K, 1)) * opb;
  if (nbatches <= own1) return launch_gemv_d<BITS, P, KS, 1, SPLIT, NW, PK>(a, M, bf16, 1, nbatches, stream);
  const int own2 = resident_blocks<BITS, P, KS, 2, SPLIT, NW, PK>(bf16, gemv_shmem<BITS, P, KS, NW>(a.K, 2)) * opb;
  int bpo = (nbatches + own2 - 1) / own2;
  bpo = ((bpo + 1) / 2) * 2;
  if (KS == 4 && bpo > GEMV_MAX_BATCHES_PER_GROUP_KS4) bpo = GEMV_MAX_BATCHES_PER_GROUP_KS4;
  if constexpr (KS == 1 && NW % 4 == 0) {
    // Uneven split: when the even split above leaves resident slots empty (gate|up of an 8B layer: 14336 batches = 3.5 per wave of
    // 256 x 16 -> 224 workgroups x 4 batches, 32 CUs idle in a launch that is bound by the vector ALUs of each CU), give
    // every resident workgroup `lw` waves with bpo batches and NW - lw with bpo - 1.
    const int res = own2 / opb;                             // resident workgroups
    static const bool uneven = (gbx::env_switch("GBX_GEMV_UNEVEN") != 0);
    if (uneven && nbatches % res == 0 && (nbatches + NW * bpo - 1) / (NW * bpo) < res) {
      const int lw = nbatches / res - NW * (bpo - 1);       // res * (lw * bpo + (NW - lw) * (bpo - 1)) == nbatches
      if (lw > 0 && lw < NW && lw % 4 == 0) {
        GemvArgs u = a;
        u.long_waves = lw;
        return launch_gemv_d<BITS, P, KS, 2, SPLIT, NW, PK>(u, M, bf16, bpo, res * NW * bpo, stream);
      }
    }
  }
  return launch_gemv_d<BITS, P, KS, 2, SPLIT, NW, PK>(a, M, bf16, bpo, nbatches, stream);
}

// Geometry: large matrices launch as fat workgroups (16 waves at one pass per row, 8 at two: the activation prologue is
// amortised over a whole CU and, since round 3, spread over all of its threads); small ones (a few rows per wave: q/k/v, o)
// as 4-wave workgroups (several per CU: every CU gets a prologue-light workgroup early).
// (measured, round 3, with the per-thread prologue items: 8 / 12 / 16-wave workgroups for q/k/v and o: 753 / 758 / 748 tok/s vs 764)
template <int BITS, int P, int KS, bool SPLIT, int MODE>
static hipError_t launch_gemv_small(const GemvArgs& a, int M, bool bf16, int nbatches, hipStream_t stream) {
  return launch_gemv_nw<BITS, P, KS, SPLIT, 4, MODE>(a, M, bf16, nbatches, stream);
}
template <int BITS, int P, int KS, bool SPLIT>
static hipError_t launch_gemv_p(const GemvArgs& a, int M, bool bf16, hipStream_t stream) {
  const int rowstep = a.swiglu ? 1 : GEMV_R;
  const int nbatches = (a.N + rowstep - 1) / rowstep;
#ifdef GBX_EXPERIMENTAL      // libgbx_experimental.so only (gbx_exp_qlinear_fused_forward)
  if (a.sz && a.x_acc) {   // ... whose activations are residual stream + fixed-point accumulator (behind gbx_attn_oproj_decode)
    if constexpr (KS == 1 && !SPLIT) {
      if (nbatches <= 4096) return launch_gemv_small<BITS, P, KS, SPLIT, 3>(a, M, bf16, nbatches, stream);
      return launch_gemv_nw<BITS, P, KS, SPLIT, WavesPerBlock<P>::value, 3>(a, M, bf16, nbatches, stream);
    }
    return hipErrorInvalidValue;
  }
#else
  if (a.x_acc) return hipErrorInvalidValue;
#endif
  if (a.sz) {              // packed scale/zero companion of the launch (decode runner)
    static const bool cepi = (gbx::env_switch("GBX_GEMV_CEPI") != 0);   // A/B switch
    if (cepi && !a.swiglu && a.nseg == 1) {
      if (KS == 1 && nbatches <= 4096) return launch_gemv_small<BITS, P, KS, SPLIT, 5>(a, M, bf16, nbatches, stream);
      return launch_gemv_nw<BITS, P, KS, SPLIT, WavesPerBlock<P>::value, 5>(a, M, bf16, nbatches, stream);
    }
    if (KS == 1 && nbatches <= 4096) return launch_gemv_small<BITS, P, KS, SPLIT, 1>(a, M, bf16, nbatches, stream);
    // (measured, round 3: 14-wave workgroups on all 256 CUs instead of 16 waves on 224: 12.8 vs 11.2 us for gate|up; 12 waves: 14.3)
    return launch_gemv_nw<BITS, P, KS, SPLIT, WavesPerBlock<P>::value, 1>(a, M, bf16, nbatches, stream);
  }
  if (a.expert_idx) {      // expert-indexed pairs (QuantizedSwitchLinear)
    if (KS == 1 && nbatches <= 4096) return launch_gemv_small<BITS, P, KS, SPLIT, 2>(a, M, bf16, nbatches, stream);
    return launch_gemv_nw<BITS, P, KS, SPLIT, WavesPerBlock<P>::value, 2>(a, M, bf16, nbatches, stream);
  }
  if (KS == 1 && nbatches <= 4096) return launch_gemv_small<BITS, P, KS, SPLIT, 0>(a, M, bf16, nbatches, stream);
  return launch_gemv_nw<BITS, P, KS, SPLIT, WavesPerBlock<P>::value, 0>(a, M, bf16, nbatches, stream);
}

// max K the lane-resident fast path covers for a bit width (beyond -> generic kernel)
template <int BITS> constexpr int gemv_max_k() { return 512 * Fmt<BITS>::VPL; }

template <int BITS, bool SPLIT>
static hipError_t dispatch_gemv_s(const GemvArgs& a, int M, bool bf16, hipStream_t s) {
  const int nslots = a.K / Fmt<BITS>::VPL;
  if (nslots <= 64) return launch_gemv_p<BITS, 1, 1, SPLIT>(a, M, bf16, s);
  if (nslots <= 128) return launch_gemv_p<BITS, 2, 1, SPLIT>(a, M, bf16, s);
  if (nslots <= 256) return launch_gemv_p<BITS, 1, 4, SPLIT>(a, M, bf16, s);
  if (nslots <= 512) return launch_gemv_p<BITS, 2, 4, SPLIT>(a, M, bf16, s);
  return hipErrorInvalidValue;
}

template <int BITS>
static hipError_t dispatch_gemv(const GemvArgs& a, int M, bool bf16, hipStream_t s) {
  if constexpr (BITS == 2) {
    if (a.log2g == 5) return dispatch_gemv_s<2, true>(a, M, bf16, s);
  }
  return dispatch_gemv_s<BITS, false>(a, M, bf16, s);
}

#ifdef GBX_EXPERIMENTAL      // the scaled-MFMA GEMVs are opt-in kernels of libgbx_experimental.so (GBX_GEMV_MXD / GBX_GEMV_MX)
#define GBX_DEFINE_LAUNCH_GEMV(BITS)                                                              \
  template <> hipError_t launch_gemv<BITS>(const GemvArgs& a, int M, bool bf16, hipStream_t s) { \
    if (BITS == 2 && gemv_mxd_eligible(a, M, bf16)) return launch_gemv_mxd(a, s);                 \
    if (BITS == 2 && gemv_mx_eligible(a, M) && gemv_mx_shape_ok(a)) return launch_gemv_mx(a, bf16, s); \
    return dispatch_gemv<BITS>(a, M, bf16, s);                                                    \
  }
#else
#define GBX_DEFINE_LAUNCH_GEMV(BITS)                                                              \
  template <> hipError_t launch_gemv<BITS>(const GemvArgs& a, int M, bool bf16, hipStream_t s) { \
    return dispatch_gemv<BITS>(a, M, bf16, s);                                                    \
  }
#endif

}  // namespace gbx
