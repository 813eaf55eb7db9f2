// Causal self-attention over a prompt chunk (L > 1 query positions against S = offset + L cached keys) on the matrix cores:
// softmax(q k^T * scale + causal mask) v without materialising the L x S scores.  Replaces the
// mx.fast.scaled_dot_product_attention call of the prefill (gbx_lm/models/base.py:101-122 via qllama.py:84-95; mask of
// base.py:27-62: query i of the chunk sees keys 0 .. offset + i), grouped-query heads read the shared K / V head directly
// (no repeated copies).  The caller of the QuantizedLinear path, not the path itself: what it has to be is not slower than
// the library it displaces (torch SDPA: 250 TFLOP/s at B = 8, T = 4096 with an explicit mask tensor).
//
// Structure: one workgroup = 4 waves = 128 query rows of one (sequence, head); a wave owns 32 query rows.  K / V tiles of 64
// keys go global -> registers -> LDS (two buffers each, K two tiles ahead, V one), both row-major; the V^T operand is
// gathered by ds_read_b64_tr_b16 (hardware transposed read).  One barrier per key tile; the S^T MFMAs of tile t+1 are issued
// before the softmax of tile t, so the vector ALU and the matrix pipe of a wave overlap.
// Both products are computed "swapped" so that a lane owns ONE query and the softmax is lane-local:
//   S^T[key, q]  = mfma(K rows, Q rows)   -> lane (q = lane & 31) holds 16 keys per 32-key sub-tile (the other 16 in lane ^ 32)
//   O^T[d, q]   += mfma(V^T rows, P^T)    -> the P^T operand of step t is exactly the lane's own p[8t .. 8t+7] (no shuffle),
//                                            the accumulator column is the lane's query: rescaling by alpha is per lane
// Online softmax in base 2 (scores scaled by scale * log2 e), fp32 statistics, P rounded to the activation dtype for P V
// (as every fused attention does), one division at the end.
// Measured variants (cfg4 shape, TFLOP/s of causal flops): V transposed by 2-byte LDS stores, 1 workgroup / CU 174; transposed
// reads + 2 workgroups / CU 547; one barrier per tile + bulk fragment reads (this file) 590-611; forced to 1 workgroup / CU 375;
// S^T of tile t+1 issued before the softmax of tile t 519, softmax per 32-key sub-tile with the other sub-tile's MFMAs issued
// first 588 (a wave issues in order: a dependent MFMA chain ahead of the VALU block holds it back until the chain has ISSUED, so
// only an instruction-level interleave would overlap them); V pitch without the bank-spreading pad 563.
// That interleave was then built and measured (round 2, not kept): TWO 32-query tiles per wave at one wave per SIMD (493
// registers), the S^T MFMAs of tile B issued one per ~12 VALU instructions of tile A's exponentials, P V of A between B's
// exponentials, every K / V fragment feeding two MFMAs, 11 tests green: 405-410 TFLOP/s.  PMC: MFMA pipe busy 19 %, VALU 28 % --
// with ONE wave per SIMD every LDS / barrier / dependency bubble is exposed (half the SIMD time is neither), and the register
// allocator parks the 128 accumulator registers in AGPRs with a copy to VGPRs per key tile; moving the accumulations to inline
// asm with AGPR constraints removed those copies, not the bubbles (and forfeits the compiler's MFMA hazard handling).
// Two waves per SIMD from two independent workgroups -- this kernel -- remain the better trade in HIP source.
#include "mfma_common.h"
#include "kernels.h"

namespace gbx {

namespace {

struct PrefillAttnArgs {
  const unsigned short* q;     // [B, Hq, L, D] through strides (elements): qs_b, qs_h, qs_l; D contiguous
  const unsigned short* k;     // [B, Hkv, S.., D]: ks_b, ks_h, row stride D
  const unsigned short* v;
  unsigned short* out;         // [B, L, Hq * D] contiguous
  long long qs_b, qs_h, qs_l, ks_b, ks_h, vs_b, vs_h;
  int B, Hq, Hkv, L, S, offset;
  float scale_log2;            // sm_scale * log2(e)
};

typedef short s16x4 __attribute__((ext_vector_type(4)));
// ds_read_b64_tr_b16: per 16-lane group a 4-row x 16-column block of 16-bit elements, delivered column-major (lane i of the
// group: column i, rows 0..3); lane 4q+p of the group supplies the address of row q, columns 4p .. 4p+3.  EXEC must be full.
__device__ __forceinline__ u32x2 lds_read_tr(const unsigned char* p) {
  return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (s16x4 __attribute__((address_space(3)))*)(__attribute__((address_space(3))) unsigned char*)p));
}

template <bool BF16, int HD>
__global__ __launch_bounds__(256, 2) void attn_prefill_kernel(const PrefillAttnArgs a) {
  constexpr int KT = 64;                       // keys per tile
  constexpr int KROW = HD * 2 + 16;            // K tile row bytes (+16: conflict-free 16-byte fragment reads)
  constexpr int VROW = HD * 2 + 64;            // V tile row bytes: pitch = 16 dwords mod 64, the 4 rows of a transposed block
                                               // (16 banks each per 32-lane half) land on disjoint banks
  constexpr int KS = HD / 16;                  // k-steps of q k^T
  constexpr int DT = HD / 32;                  // 32-row tiles of O^T
  constexpr int PPT = KT * HD / 8 / 256;       // 16-byte pieces of a K (or V) tile per thread
  constexpr int KBUF = KT * KROW, VBUF = KT * VROW;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * KBUF + 2 * VBUF];   // K ping | K pong | V ping | V pong
  unsigned char* kl = lds;
  unsigned char* vl = lds + 2 * KBUF;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int head = blockIdx.y, b = blockIdx.z, kvh = head / (a.Hq / a.Hkv);
  const int qb = (gridDim.x - 1 - blockIdx.x) * 128;                     // longest key sweeps are dispatched first
  const int qi = qb + wave * 32 + r;                                     // this lane's query (both halves of the wave)
  const int L = a.L, S = a.S;
  const unsigned short* kg = a.k + b * a.ks_b + kvh * a.ks_h;
  const unsigned short* vg = a.v + b * a.vs_b + kvh * a.vs_h;

  // Q fragments (second MFMA operand: column = query, 8 d-values per half and k-step), kept for the whole key sweep
  u32x4 qf[KS];
  {
    const unsigned short* qp = a.q + b * a.qs_b + head * a.qs_h + static_cast<long long>(min(qi, L - 1)) * a.qs_l + h * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const u32x4*>(qp + ks * 16);
  }
  f32x16 o[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i) o[i] = f32x16{0};
  // running maximum in RAW score units (the scale is folded into the exponent's fma), partial sum of this lane's 32 keys/tile
  float m_run = -INFINITY, l_run = 0.f;
  const float c = a.scale_log2;

  // keys this workgroup needs: 0 .. offset + (last query of the block), at most S
  const int kmax = min(S, a.offset + min(qb + 128, L));
  const int ntiles = (kmax + KT - 1) / KT;
  u32x4 kr[PPT], vr[PPT];
  // K / V tiles by buffer loads: one instruction per 16 bytes, the tile offset in an SGPR, nothing recomputed per tile on the
  // vector ALU; rows past S (last tile, and the tiles requested past the sweep) read as zeros -- masked by the key limit
  const unsigned nrec = static_cast<unsigned>(min(static_cast<long long>(S) * HD * 2, 0xFFFFFFFFLL));
  const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(kg), 0, nrec, 0x00020000);
  const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(vg), 0, nrec, 0x00020000);
  const int voff = tid * 16;                                              // piece p = tid + 256 i of a tile: byte offset 16 p
  auto load_k = [&](int t) {
#pragma unroll
    for (int i = 0; i < PPT; ++i) kr[i] = __builtin_amdgcn_raw_buffer_load_b128(krs, voff + i * 4096, t * (KT * HD * 2), 0);
  };
  auto load_v = [&](int t) {
#pragma unroll
    for (int i = 0; i < PPT; ++i) vr[i] = __builtin_amdgcn_raw_buffer_load_b128(vrs, voff + i * 4096, t * (KT * HD * 2), 0);
  };
  auto store_k = [&](int buf) {
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const int p = tid + 256 * i;
      *reinterpret_cast<u32x4*>(kl + buf * KBUF + (p / (HD / 8)) * KROW + (p % (HD / 8)) * 16) = kr[i];
    }
  };
  auto store_v = [&](int buf) {
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const int p = tid + 256 * i;
      *reinterpret_cast<u32x4*>(vl + buf * VBUF + (p / (HD / 8)) * VROW + (p % (HD / 8)) * 16) = vr[i];
    }
  };
  // transposed-read base of this lane: row 4h + q, columns 16 * (second 16-lane group of the half) + 4p
  const unsigned char* vbase = vl + (4 * h + ((lane >> 2) & 3)) * VROW + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
  const unsigned char* kbase = kl + r * KROW + h * 16;

  // ONE barrier per key tile: at the top of iteration t (after the barrier) tile t is in buffer t & 1 and tile t+1 is in
  // registers (requested an iteration ago); it is parked in the other buffer -- whose readers all passed the barrier -- and
  // tile t+2 is requested.  LDS fragment reads are issued in bulk ahead of the MFMAs that consume them (the compiler's own
  // order was read-two / wait / multiply, every LDS latency exposed): the K fragments of the second sub-tile refill the
  // registers of the first as its MFMAs retire, the first half of the V^T fragments is requested BEFORE the softmax.
  if (ntiles > 0) {
    load_k(0); load_v(0);
    store_k(0); store_v(0);
    load_k(1); load_v(1);
  }
  for (int t = 0; t < ntiles; ++t) {
    __syncthreads();
    store_k((t + 1) & 1);
    store_v((t + 1) & 1);
    load_k(t + 2);
    load_v(t + 2);

    // ---- S^T = K Q^T for the two 32-key sub-tiles
    f32x16 st[2];
    st[0] = f32x16{0};
    st[1] = f32x16{0};
    const unsigned char* kb = kbase + (t & 1) * KBUF;
    u32x4 fr[KS];                                                         // 8 fragments in flight (K, then V^T)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) fr[ks] = *reinterpret_cast<const u32x4*>(kb + ks * 32);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      st[0] = mfma<BF16>(fr[ks], qf[ks], st[0]);
      fr[ks] = *reinterpret_cast<const u32x4*>(kb + 32 * KROW + ks * 32);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) st[1] = mfma<BF16>(fr[ks], qf[ks], st[1]);
    __builtin_amdgcn_sched_barrier(0);
    // V^T fragments of d-tiles 0 .. DT/2-1 (keys 32u + 16tt + 4h + {0..3} and the same + 8): in flight under the softmax
    const unsigned char* vb = vbase + (t & 1) * VBUF;
    auto vfrag = [&](int dt, int j) {                                     // j = 2u + tt
      const u32x2 v0 = lds_read_tr(vb + ((j >> 1) * 32 + (j & 1) * 16) * VROW + dt * 64);
      const u32x2 v1 = lds_read_tr(vb + ((j >> 1) * 32 + (j & 1) * 16 + 8) * VROW + dt * 64);
      return u32x4{v0.x, v0.y, v1.x, v1.y};
    };
    constexpr int NF = KS;                                                // fragments held: (DT / 2) d-tiles x 4 = 8 at HD 128
#pragma unroll
    for (int i = 0; i < NF; ++i) fr[i] = vfrag(i >> 2, i & 3);
    __builtin_amdgcn_sched_barrier(0);

    // ---- causal / length mask on the tiles that reach past some query's last key (wave-uniform test)
    if ((t + 1) * KT - 1 > min(a.offset + qb + wave * 32, S - 1)) {
      const int klim = min(a.offset + qi, S - 1) - t * KT - 4 * h;
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int e = 0; e < 16; ++e) st[u][e] = (u * 32 + (e & 3) + 8 * (e >> 2)) <= klim ? st[u][e] : -INFINITY;
    }
    float mx = st[0][0];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int e = 0; e < 16; ++e) mx = fmaxf(mx, st[u][e]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));                                   // the other 32 keys of this query
    // ---- rescale only when some query's maximum grew by more than 2^8 in exponent units (P <= 256 otherwise: exact
    //      softmax either way, the factor cancels in the final division)
    if (__builtin_amdgcn_ballot_w64((mx - m_run) * c > 8.f) != 0) {
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);    // (m_new is finite: key 0 is visible to every query)
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[dt][e] *= alpha;
    }
    const float mc = -m_run * c;
    float ls = 0.f;
    u32x4 pf[4];                                                          // P^T operands: index 2u + tt
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float p[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) { p[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[u][e], c, mc)); ls += p[e]; }
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
        pf[2 * u + tt] = u32x4{pack2<BF16>(p[8 * tt], p[8 * tt + 1]), pack2<BF16>(p[8 * tt + 2], p[8 * tt + 3]),
                               pack2<BF16>(p[8 * tt + 4], p[8 * tt + 5]), pack2<BF16>(p[8 * tt + 6], p[8 * tt + 7])};
    }
    l_run += ls;
    __builtin_amdgcn_sched_barrier(0);
    // ---- O^T += V^T P^T: the held fragments first, each slot refilled with the fragment NF positions later as it retires
#pragma unroll
    for (int i = 0; i < DT * 4; ++i) {
      o[i >> 2] = mfma<BF16>(fr[i % NF], pf[i & 3], o[i >> 2]);
      if (i + NF < DT * 4) fr[i % NF] = vfrag((i + NF) >> 2, i & 3);
    }
  }

  // ---- normalise and store: lane (q, h) holds d = 32 dt + (e & 3) + 8 (e >> 2) + 4 h of its query
  const float lt = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.f / lt;
  if (qi < L) {
    unsigned short* op = a.out + (static_cast<long long>(b) * L + qi) * (a.Hq * HD) + head * HD;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = dt * 32 + 8 * g + 4 * h;
        *reinterpret_cast<u32x2*>(op + d) = u32x2{pack2<BF16>(o[dt][4 * g] * inv, o[dt][4 * g + 1] * inv),
                                                pack2<BF16>(o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv)};
      }
  }
}

}  // namespace

hipError_t launch_attn_prefill(const void* q, const void* k, const void* v, void* out, int B, int Hq, int Hkv, int L, int S, int D,
                               long long qs_b, long long qs_h, long long qs_l, long long ks_b, long long ks_h, long long vs_b,
                               long long vs_h, int offset, float sm_scale, bool bf16, hipStream_t s) {
  PrefillAttnArgs a{static_cast<const unsigned short*>(q), static_cast<const unsigned short*>(k), static_cast<const unsigned short*>(v),
                    static_cast<unsigned short*>(out), qs_b, qs_h, qs_l, ks_b, ks_h, vs_b, vs_h, B, Hq, Hkv, L, S, offset,
                    sm_scale * 1.4426950408889634f};
  dim3 grid((L + 127) / 128, Hq, B), block(256);
  if (D == 128) {
    if (bf16) hipLaunchKernelGGL((attn_prefill_kernel<true, 128>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((attn_prefill_kernel<false, 128>), grid, block, 0, s, a);
  } else if (D == 64) {
    if (bf16) hipLaunchKernelGGL((attn_prefill_kernel<true, 64>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((attn_prefill_kernel<false, 64>), grid, block, 0, s, a);
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace gbx
