// The MLP half of a decoder block at M == 1 as ONE launch whose hand-off never leaves an XCD ("dmlp", round 5):
//
//   x' = round(h + round(o_proj sums))  ->  RMSNorm(x')  ->  gate | up rows of ONE eighth of the intermediate vector + SwiGLU
//      ->  XCD-local hand-off of those I / 8 activations  ->  down_proj over exactly that eighth of its input columns
//      ->  fixed-point partial sums into acc_out[D]
//
// replaces two launches of the three-launch block (gate|up + SwiGLU, down + residual):
//   mx.quantized_matmul at M == 1 (gbx_lm/models/quantized_linear_gba.py:195-203) at the call sites of gbx_lm/models/qllama.py:115
//   (down_proj(silu(gate_proj(x)) * up_proj(x))) with the post-attention RMSNorm (:139) in front and the residual adds (:138, :140)
//   folded into the prologue of this launch and of the next block's fused attention launch (dattn_impl.h, x_acc).
//
// Why: down_proj is the block's third all-to-all link.  K-split by XCD it is not: XCD g produces act[g I/8, (g+1) I/8) (the gate / up
// rows of those outputs) and consumes exactly that range as down_proj's input columns, so the only exchange is inside the XCD --
// 0.45 us as data-tagged granules (profiles/r05_xcd_barrier.txt) instead of a launch boundary (~1.5 us) plus the down launch's
// own activation prologue (~2.3 us before its first row is consumed).  Its weights are requested behind each wave's last gate / up
// rows and sit in registers when the activations arrive.  The eight partial sums per output row meet in a 2^-32 fixed-point
// accumulator (integer adds: order-independent, bit-reproducible), which the NEXT launch's prologue turns into the residual stream.
//
// Arithmetic: dgemv_mx.hip's (FP4 x FP8 matrix cores: a 2-bit field masked in place is the E2M1 number q / 2, the activations are two
// E4M3 limbs with one E8M0 exponent per 32 values, the diagonal of the 16 x 16 result holds 16 per-group sums) for BOTH matrix
// products.  2 bits, group 128, bf16, K = 4096 (two 2048-value spans), I / 8 <= 2048 a multiple of 128, D % 512 == 0.
// Round 6, BITS = 4 (same shapes): a 4-bit field q = lo + 4 hi is TWO 2-bit fields in one nibble, so the same two masks that split a 2-bit
// word into its even and odd fields -- w & 0x33333333, (w >> 2) & 0x33333333 -- split a 4-bit word into the E2M1 operands lo / 2 and hi / 2 of
// the SAME 8 values; the second pair of MFMAs carries hi with a B-side block scale of 2^2: sum x (lo / 2 + 2 hi) = dot / 2 as before.  No
// repacking, a natural-order activation image (4 consecutive values = 4 consecutive bytes), twice the MFMAs, a quarter of the vector
// instructions of the vector form (dmlpv.hip).  down_proj's 112 KB of words per CU do not fit registers: LDS-DMA, two rows behind each
// gate | up batch.  Everything else (other widths / groups / shapes) stays on the three-launch block or dmlpv.
#include "dattn_impl.h"
#include "gbx_env.h"
#include "kernels.h"
#include <type_traits>

namespace gbx {

typedef int dml_v8i __attribute__((ext_vector_type(8)));
typedef float dml_v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned long long dml_gu64;
typedef __attribute__((address_space(1))) unsigned dml_gu32;

constexpr int DM_LDS_BYTES = 84 * 1024;    // one workgroup per CU
constexpr int DM_LDS_DW = 32 * 1024;       // BITS = 4: down_proj's words, [wave][8 rows][64 lanes][16 B]
constexpr int DM_LDS_BYTES4 = DM_LDS_DW + 128 * 1024;

template <int SP, int BITS>
__global__ __launch_bounds__(DA_NT) void dmlp_kernel(const DmlpArgs a) {
  static_assert(BITS == 2 || BITS == 3 || BITS == 4, "2-bit fields; 4-bit fields as (lo, hi) pairs of 2-bit fields; 3-bit rows REPACKED as a 2-bit and a 1-bit plane");
  // a lane's 32 values of one (row, k-block).  BITS = 3: the rows must be in the plane layout of gbx_repack_planes (misc_kernels.hip) -- words
  // 0 / 1 a 2-bit row's words (the low two bits), word 2 the high bits with (h >> i) & 0x11111111 = operand dword i
  using WT = typename std::conditional<BITS == 4, u32x4, typename std::conditional<BITS == 3, u32x3, u32x2>::type>::type;
  constexpr int LB = 4 * BITS;                                   // bytes of those 32 values
  constexpr bool BF16 = true;
  constexpr int NW = DA_NW, NT = DA_NT;
  constexpr int NBLK = SP * 64;
  constexpr unsigned KILL = 0x80000000u;
  constexpr int RS_FLAGS = 0x00020000, RS_ALL = 0x7FFFFFFF;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* img = smem;                                     // gate|up operand: [span][limb][piece][64 lanes][16 B]
  float* bsum = reinterpret_cast<float*>(smem + SP * 4096);
  unsigned char* scl = reinterpret_cast<unsigned char*>(bsum + NBLK);
  float* red_lds = reinterpret_cast<float*>(scl + NBLK);         // [8] RMSNorm partials
  unsigned char* img2 = smem + 16 * 1024;                        // down operand: one span
  float* bsum2 = reinterpret_cast<float*>(img2 + 4096);
  unsigned char* scl2 = reinterpret_cast<unsigned char*>(bsum2 + 64);

  const int K = a.K;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef GBX_STAMPS
#define DML_STAMP(i) do { if (lane == 0 && a.stamps) a.stamps[(static_cast<size_t>(blockIdx.x) * NW + wave) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define DML_STAMP(i) do {} while (0)
#endif
  DML_STAMP(0);
  const int g = static_cast<int>(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u);     // HW_REG_XCC_ID: this XCD = this eighth of the intermediate vector
  const int r = static_cast<int>(blockIdx.x) >> 3;
  dml_gu32* gen_p = reinterpret_cast<dml_gu32*>(reinterpret_cast<uintptr_t>(a.state)) + g * 16;
  dml_gu64* hand = reinterpret_cast<dml_gu64*>(reinterpret_cast<uintptr_t>(a.hand)) + static_cast<size_t>(g) * DA_HAND_STRIDE;
  auto rsrc = [](const void* ptr) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, RS_ALL, RS_FLAGS); };
  const int W4 = K * BITS / 8;                                   // bytes per packed gate / up row
  const int G = K >> 7;
  const int IX = a.I >> 3;                                       // activations of this XCD
  const int nn = lane & 15, kg = lane >> 4;
  const unsigned voff_w = nn * (4 * LB) + kg * LB;               // this lane's bytes inside a span (2048 values = 64 LB bytes)
  const unsigned voff_s = nn * 8;
  // this wave's gate | up batches: activations [n0, n0 + nloc) of the XCD, consecutive so that two of them share a granule
  const int nloc = wave < a.a4 ? 4 : 2;
  const int nx0 = r * a.bpw + (wave < a.a4 ? 4 * wave : 4 * a.a4 + 2 * (wave - a.a4));
  const int n0 = g * IX + nx0;
  const __amdgpu_buffer_rsrc_t qd = rsrc(a.gate_qw), qd2 = rsrc(a.up_qw), szd = rsrc(a.gu_sz);
  // down_proj: this wave's rows, this XCD's columns
  const int RPWG = a.D / DA_MEMBERS;                             // rows of this workgroup
  const int rows_w = RPWG / NW;                                  // rows of this wave (8)
  const int W4d = a.I * BITS / 8;                                // bytes per packed down row
  const int Gd = a.I >> 7;
  const int ng = IX >> 7;                                        // 128-value groups of this XCD's slice (<= 16)
  const __amdgpu_buffer_rsrc_t dd = rsrc(a.down_qw), dsd = rsrc(a.down_sz);
  const unsigned killd = nn < ng ? 0u : KILL;
  const int drow0 = r * RPWG + wave * rows_w;
  constexpr bool DWL = BITS != 2;                                // down_proj's words through LDS (84 / 112 KB per CU do not wait well in registers)
  unsigned char* dwl = smem + DM_LDS_DW + wave * 8192;           // DWL: [row][lane][16 B] (a 12-byte lane lands at lane * 16 too)
  // DWL: two rows of down_proj's words by LDS-DMA (no registers; a killed lane's bytes arrive as zeros)
  auto dma_down = [&](int i0) {
    if constexpr (DWL) {
#pragma unroll
      for (int i = i0; i < i0 + 2; ++i) {
        if (i < rows_w) {
          const unsigned vo = (voff_w + static_cast<unsigned>(g * (IX * BITS / 8))) | killd;
          if constexpr (BITS == 4)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(dd, (__attribute__((address_space(3))) void*)(dwl + i * 1024), 16, vo, (drow0 + i) * W4d, 0, 2);
          else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(dd, (__attribute__((address_space(3))) void*)(dwl + i * 1024), 12, vo, (drow0 + i) * W4d, 0, 2);
        }
      }
    }
  };
  struct Batch { WT w[2][SP]; u32x2 sz[SP]; };
  auto ldw = [](const __amdgpu_buffer_rsrc_t d, unsigned voff, int soff) -> WT {
    if constexpr (BITS == 4) return __builtin_amdgcn_raw_buffer_load_b128(d, voff, soff, 2);
    else if constexpr (BITS == 3) return __builtin_amdgcn_raw_buffer_load_b96(d, voff, soff, 2);
    else return __builtin_amdgcn_raw_buffer_load_b64(d, voff, soff, 2);
  };
  // four consecutive values of a 32-value block (item u of 8) as two E4M3 limbs into an operand image: where they go depends on which
  // operand dword / nibble carries the weight they meet (2 bits: even fields of a word in one dword, odd fields in the next; 4 bits: natural order)
  auto put_item = [](unsigned char* piece_lane, int u, float r0, float r1, float r2, float r3) {
    if constexpr (BITS == 4) {
      int ph = __builtin_amdgcn_cvt_pk_fp8_f32(r0, r1, 0, false);
      ph = __builtin_amdgcn_cvt_pk_fp8_f32(r2, r3, ph, true);
      int pl = __builtin_amdgcn_cvt_pk_fp8_f32(r0 - __builtin_amdgcn_cvt_f32_fp8(ph, 0), r1 - __builtin_amdgcn_cvt_f32_fp8(ph, 1), 0, false);
      pl = __builtin_amdgcn_cvt_pk_fp8_f32(r2 - __builtin_amdgcn_cvt_f32_fp8(ph, 2), r3 - __builtin_amdgcn_cvt_f32_fp8(ph, 3), pl, true);
      *reinterpret_cast<int*>(piece_lane + 4 * (u & 3)) = ph;
      *reinterpret_cast<int*>(piece_lane + 2048 + 4 * (u & 3)) = pl;
    } else {
      unsigned char* dst = piece_lane + 2 * (u & 3);
      const int pe = __builtin_amdgcn_cvt_pk_fp8_f32(r0, r2, 0, false);
      const int po = __builtin_amdgcn_cvt_pk_fp8_f32(r1, r3, 0, false);
      *reinterpret_cast<unsigned short*>(dst) = static_cast<unsigned short>(pe);
      *reinterpret_cast<unsigned short*>(dst + 8) = static_cast<unsigned short>(po);
      const int le = __builtin_amdgcn_cvt_pk_fp8_f32(r0 - __builtin_amdgcn_cvt_f32_fp8(pe, 0), r2 - __builtin_amdgcn_cvt_f32_fp8(pe, 1), 0, false);
      const int lo = __builtin_amdgcn_cvt_pk_fp8_f32(r1 - __builtin_amdgcn_cvt_f32_fp8(po, 0), r3 - __builtin_amdgcn_cvt_f32_fp8(po, 1), 0, false);
      *reinterpret_cast<unsigned short*>(dst + 2048) = static_cast<unsigned short>(le);
      *reinterpret_cast<unsigned short*>(dst + 2048 + 8) = static_cast<unsigned short>(lo);
    }
  };
  // one (row, span) on the matrix cores: the result's diagonal holds sum x q / 2 per group
  auto mx_row = [](const dml_v8i& a_hi, const dml_v8i& a_lo, const WT& w, int opsel, int sa) -> dml_v4f {
    dml_v4f z = {0.f, 0.f, 0.f, 0.f};
    if constexpr (BITS == 4) {
      dml_v8i lo, hi;
      lo[0] = w.x & 0x33333333u; lo[1] = w.y & 0x33333333u; lo[2] = w.z & 0x33333333u; lo[3] = w.w & 0x33333333u;
      hi[0] = (w.x >> 2) & 0x33333333u; hi[1] = (w.y >> 2) & 0x33333333u; hi[2] = (w.z >> 2) & 0x33333333u; hi[3] = (w.w >> 2) & 0x33333333u;
      lo[4] = 0; lo[5] = 0; lo[6] = 0; lo[7] = 0; hi[4] = 0; hi[5] = 0; hi[6] = 0; hi[7] = 0;
      if (opsel == 0) {
        z = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a_hi, lo, z, 0, 4, 0, sa, 0, 127);
        z = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a_lo, lo, z, 0, 4, 0, sa, 0, 127);
        z = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a_hi, hi, z, 0, 4, 0, sa, 0, 129);
        z = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a_lo, hi, z, 0, 4, 0, sa, 0, 129);
      } else {
        z = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a_hi, lo, z, 0, 4, 1, sa, 0, 127);
        z = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a_lo, lo, z, 0, 4, 1, sa, 0, 127);
        z = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a_hi, hi, z, 0, 4, 1, sa, 0, 129);
        z = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a_lo, hi, z, 0, 4, 1, sa, 0, 129);
      }
    } else {
      dml_v8i bv;
      bv[0] = w.x & 0x33333333u; bv[1] = (w.x >> 2) & 0x33333333u; bv[2] = w.y & 0x33333333u; bv[3] = (w.y >> 2) & 0x33333333u;
      bv[4] = 0; bv[5] = 0; bv[6] = 0; bv[7] = 0;
      if (opsel == 0) {
        z = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a_hi, bv, z, 0, 4, 0, sa, 0, 127);
        z = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a_lo, bv, z, 0, 4, 0, sa, 0, 127);
      } else {
        z = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a_hi, bv, z, 0, 4, 1, sa, 0, 127);
        z = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a_lo, bv, z, 0, 4, 1, sa, 0, 127);
      }
      if constexpr (BITS == 3) {                                 // the 1-bit plane: E2M1 code 1 = 0.5, block scale 2^2: 2 x bit = (4 x bit) / 2
        dml_v8i hb;
        hb[0] = w.z & 0x11111111u; hb[1] = (w.z >> 1) & 0x11111111u; hb[2] = (w.z >> 2) & 0x11111111u; hb[3] = (w.z >> 3) & 0x11111111u;
        hb[4] = 0; hb[5] = 0; hb[6] = 0; hb[7] = 0;
        if (opsel == 0) {
          z = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a_hi, hb, z, 0, 4, 0, sa, 0, 129);
          z = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a_lo, hb, z, 0, 4, 0, sa, 0, 129);
        } else {
          z = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a_hi, hb, z, 0, 4, 1, sa, 0, 129);
          z = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a_lo, hb, z, 0, 4, 1, sa, 0, 129);
        }
      }
    }
    return z;
  };
  // K < SP * 2048 (Llama-3.2-3B: 3072): groups beyond K exist in the operand layout only -- their lanes fetch nothing (zeros), their image is zero
  unsigned gkill[SP];
#pragma unroll
  for (int h = 0; h < SP; ++h) gkill[h] = (h * 16 + nn) < G ? 0u : KILL;
  auto issue = [&](Batch& b, int j) {
    const unsigned kill = j < nloc ? 0u : KILL;
    const int row = n0 + (j < nloc ? j : 0);
    const int woff = row * W4;
#pragma unroll
    for (int h = 0; h < SP; ++h) b.w[0][h] = ldw(qd, (voff_w + h * (64 * LB)) | kill | gkill[h], woff);
#pragma unroll
    for (int h = 0; h < SP; ++h) b.w[1][h] = ldw(qd2, (voff_w + h * (64 * LB)) | kill | gkill[h], woff);
#pragma unroll
    for (int h = 0; h < SP; ++h) b.sz[h] = __builtin_amdgcn_raw_buffer_load_b64(szd, (voff_s + h * 128) | kill | gkill[h], row * (G * 8), 0);
  };

  // ---------------- 1. prologue on waves 0-7 (dgemv_mx.hip): x' from the fixed-point sums, statistic, limb image ----------------
  constexpr int NIT = SP * 512, PW = 8, PT = PW * 64, IPT = (NIT + PT - 1) / PT;
  const int nit = K >> 2;                                        // live items (4 values each); items nit .. NIT - 1 are written as zeros
  const bool pro = wave < PW;
  u32x2 xr[IPT], wr[IPT];
  if (a.zero_buf) {
    for (int i = static_cast<int>(blockIdx.x) * NT + tid; i < a.zero_n; i += static_cast<int>(gridDim.x) * NT) a.zero_buf[i] = 0ull;
  }
  if (pro) {
    u32x4 acv[IPT][2];
    const __amdgpu_buffer_rsrc_t ad = rsrc(a.x_acc);
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
      const int it = min(tid + PT * j, nit - 1);
      acv[j][0] = __builtin_amdgcn_raw_buffer_load_b128(ad, static_cast<unsigned>(it * 32), 0, 0);
      acv[j][1] = __builtin_amdgcn_raw_buffer_load_b128(ad, static_cast<unsigned>(it * 32 + 16), 0, 0);
    }
    const unsigned apoison = __builtin_amdgcn_raw_buffer_load_b32(ad, static_cast<unsigned>(K * 8), 0, 0);
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
      const int it = min(tid + PT * j, nit - 1);
      xr[j] = *reinterpret_cast<const u32x2*>(a.x + 4 * it);
      wr[j] = *reinterpret_cast<const u32x2*>(a.norm_w + 4 * it);
    }
    DML_STAMP(1);
    const bool poison = apoison != 0u;
    float ssq = 0.f;
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
      const int it = min(tid + PT * j, nit - 1);
      const bool live = (tid + PT * j) < nit;
      unsigned wv[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const u32x4 av = acv[j][q];
        const float o0 = round_bf16_or_f16(fx32_to_f32(av.x, av.y), BF16);
        const float o1 = round_bf16_or_f16(fx32_to_f32(av.z, av.w), BF16);
        const unsigned xw = q == 0 ? xr[j].x : xr[j].y;
        float x0 = act_lo<BF16>(xw) + o0, x1 = act_hi<BF16>(xw) + o1;
        if (poison) { x0 = __uint_as_float(0x7FC00000u); x1 = x0; }
        wv[q] = static_cast<unsigned>(f32_to_act<BF16>(x0)) | (static_cast<unsigned>(f32_to_act<BF16>(x1)) << 16);
        const float lo = act_lo<BF16>(wv[q]), hi = act_hi<BF16>(wv[q]);
        ssq += live ? fmaf(hi, hi, lo * lo) : 0.f;
      }
      xr[j] = live ? u32x2{wv[0], wv[1]} : u32x2{0u, 0u};
      if (!live) wr[j] = u32x2{0u, 0u};
      if (blockIdx.x == 0 && live && a.x_out) *reinterpret_cast<u32x2*>(a.x_out + 4 * it) = xr[j];
    }
    ssq = wave_sum(ssq);
    if (lane == 0) red_lds[wave] = ssq;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  else {
    // the late waves clear down_proj's operand image meanwhile: its rows / scale bytes of groups beyond this XCD's slice are never
    // written below, and a stale NaN pattern there would reach every lane's diagonal pick as NaN * 0
    const int i = tid - PT;                                      // 0 .. 511
    if (i < 256) *reinterpret_cast<u32x4*>(img2 + i * 16) = u32x4{0, 0, 0, 0};
    else if (i < 256 + 20) *reinterpret_cast<u32x4*>(img2 + 4096 + (i - 256) * 16) = u32x4{0, 0, 0, 0};      // bsum2 (256 B) + scl2 (64 B)
  }
  __syncthreads();                                               // barrier A: the activations have landed (cue for the late waves' first rows)
  const unsigned gen = __hip_atomic_load(gen_p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  DML_STAMP(9);
  if (pro) {
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < PW; ++i) tot += red_lds[i];
    const float rstd = rsqrtf(tot / static_cast<float>(K) + a.eps);
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
      if ((tid & ~63) + PT * j < NIT) {
        const int it = tid + PT * j;
        float v[4] = {act_lo<BF16>(xr[j].x), act_hi<BF16>(xr[j].x), act_lo<BF16>(xr[j].y), act_hi<BF16>(xr[j].y)};
        v[0] = round_bf16_or_f16(v[0] * rstd * act_lo<BF16>(wr[j].x), BF16);
        v[1] = round_bf16_or_f16(v[1] * rstd * act_hi<BF16>(wr[j].x), BF16);
        v[2] = round_bf16_or_f16(v[2] * rstd * act_lo<BF16>(wr[j].y), BF16);
        v[3] = round_bf16_or_f16(v[3] * rstd * act_hi<BF16>(wr[j].y), BF16);
        float sum = (v[0] + v[1]) + (v[2] + v[3]);
        unsigned amax = max(max(__float_as_uint(v[0]) & 0x7FFFFFFFu, __float_as_uint(v[1]) & 0x7FFFFFFFu),
                            max(__float_as_uint(v[2]) & 0x7FFFFFFFu, __float_as_uint(v[3]) & 0x7FFFFFFFu));
        sum += dpp_f32<0xB1>(sum); amax = max(amax, dpp_u32<0xB1>(amax));
        sum += dpp_f32<0x4E>(sum); amax = max(amax, dpp_u32<0x4E>(amax));
        sum += dpp_f32<0x141>(sum); amax = max(amax, dpp_u32<0x141>(amax));
        if (amax >= 0x7F800000u) sum = __uint_as_float(0x7FC00000u);   // Inf / NaN in the block: NaN through the zero-point term of every row
        int ea = static_cast<int>(amax >> 23);
        ea = min(max(ea, 20), 240);
        const float up = exp2i(134 - ea);
        if (it < NIT) {
          const int b = it >> 3, u = it & 7;
          const int h = b >> 6, m = (b & 63) >> 2, kb = b & 3;
          const int ga = 2 * (kb & 1) + (u >> 2);
          put_item(img + (h * 4 + (kb >> 1)) * 1024 + (m + 16 * ga) * 16, u, v[0] * up, v[1] * up, v[2] * up, v[3] * up);
          if (u == 0) { bsum[h * 64 + m + 16 * kb] = sum; scl[h * 64 + m + 16 * kb] = static_cast<unsigned char>(ea - 7); }
        }
      }
    }
  }
  DML_STAMP(10);
  Batch ring[2];
  // 4 bits: a batch is 64 KB per CU, more than a CU keeps in flight (~50 KB): requested in front of barrier B by all 16 waves, the eight
  // image-writing waves waited to ISSUE and barrier B with them (3.8 instead of 2.9 us).  They ask behind the barrier instead.
  if (BITS == 2 || !pro) issue(ring[0], 0);
  __syncthreads();                                               // barrier B: image written
  DML_STAMP(2);
  if (BITS != 2 && pro) issue(ring[0], 0);
  issue(ring[1], 1);
  unsigned tag = gen + 1u;
  if (tag == 0u) tag = 1u;

  // ---------------- 2. gate | up on the matrix cores (dgemv_mx.hip), SwiGLU, two activations per granule ----------------
  const bool diag = kg == (nn >> 2);
  float dm[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) dm[q] = (diag && (lane & 3) == q) ? 1.f : 0.f;
  unsigned short actv[4] = {0, 0, 0, 0};
  {
    dml_v8i areg[SP][2];
    int sa = 0;
    float xsg[SP];
#pragma unroll
    for (int h = 0; h < SP; ++h) {
#pragma unroll
      for (int l = 0; l < 2; ++l) {
        const u32x4 p0 = *reinterpret_cast<const u32x4*>(img + (h * 4 + 0) * 1024 + l * 2048 + lane * 16);
        const u32x4 p1 = *reinterpret_cast<const u32x4*>(img + (h * 4 + 1) * 1024 + l * 2048 + lane * 16);
        areg[h][l][0] = p0.x; areg[h][l][1] = p0.y; areg[h][l][2] = p0.z; areg[h][l][3] = p0.w;
        areg[h][l][4] = p1.x; areg[h][l][5] = p1.y; areg[h][l][6] = p1.z; areg[h][l][7] = p1.w;
      }
      sa |= static_cast<int>(scl[h * 64 + lane]) << (8 * h);
      const float gs = (bsum[h * 64 + nn] + bsum[h * 64 + nn + 16]) + (bsum[h * 64 + nn + 32] + bsum[h * 64 + nn + 48]);
      xsg[h] = diag ? gs : 0.f;
    }
    auto consume = [&](const Batch& cur, int j) {
      if (j >= nloc) return;                                     // wave-uniform
      dml_v4f c[2][SP];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        static_assert(SP <= 2, "the scale operand's byte select is 0 / 1");
        c[i][0] = mx_row(areg[0][0], areg[0][1], cur.w[i][0], 0, sa);
        if constexpr (SP > 1) c[i][1] = mx_row(areg[1][0], areg[1][1], cur.w[i][1], 1, sa);
      }
      float acc[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float yv = 0.f;
#pragma unroll
        for (int h = 0; h < SP; ++h) {
          const float dsel = fmaf(c[i][h][0], dm[0], fmaf(c[i][h][1], dm[1], fmaf(c[i][h][2], dm[2], c[i][h][3] * dm[3])));
          const unsigned szw = i == 0 ? cur.sz[h].x : cur.sz[h].y;
          const float sc = sz_scale<BF16>(szw), ze = sz_zero<BF16>(szw);
          yv += fmaf(ze, xsg[h], (sc + sc) * dsel);
        }
        acc[i] = yv;
      }
      const float v0 = wave_sum2(acc[0], acc[1]);
      const float t0 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v0), 31));
      const float t1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v0), 63));
      const float gt = round_bf16_or_f16(t0, BF16), up = round_bf16_or_f16(t1, BF16);
      const float sl = round_bf16_or_f16(gt / (1.f + __expf(-gt)), BF16);
      actv[j] = f32_to_act<BF16>(sl * up);
    };
    auto publish = [&](int q) {                                  // activations nx0 + 2 q, + 1 -> granule (nx0 >> 1) + q (sc0: stays in this XCD's L2)
      if (2 * q >= nloc) return;
      const unsigned val = static_cast<unsigned>(actv[2 * q]) | (static_cast<unsigned>(actv[2 * q + 1]) << 16);
      if (lane == 0) {
        __hip_atomic_store(hand + (nx0 >> 1) + q, (static_cast<unsigned long long>(tag) << 32) | val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (a.act_out) *reinterpret_cast<unsigned*>(a.act_out + n0 + 2 * q) = val;
      }
    };
    consume(ring[0], 0);
    issue(ring[0], 2);
    dma_down(0);
    consume(ring[1], 1);
    publish(0);
    issue(ring[1], 3);
    dma_down(2);
    DML_STAMP(3);
    // ---- down_proj words of this wave's 8 rows, this XCD's columns: requested behind the last gate / up rows, consumed in 4.
    consume(ring[0], 2);
    dma_down(4);
    consume(ring[1], 3);
    dma_down(6);
    publish(1);
  }
  DML_STAMP(4);
  // (requested here rather than between the consumes: the matrix-core operands above keep ~90 VGPRs live)
  WT dw[!DWL ? 8 : 1];
  u32x2 dsz[4];
  {
    const int row0 = drow0;
    if constexpr (!DWL) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        dw[i] = ldw(dd, (voff_w + static_cast<unsigned>(g * (IX * BITS / 8))) | (i < rows_w ? killd : KILL), (row0 + (i < rows_w ? i : 0)) * W4d);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)      // rows (row0 + 2 q, + 1): one (scale | zero) pair each for group g * ng + nn
      dsz[q] = __builtin_amdgcn_raw_buffer_load_b64(dsd, static_cast<unsigned>((g * ng + nn) * 8) | (2 * q < rows_w ? killd : KILL),
                                                    ((row0 >> 1) + (2 * q < rows_w ? q : 0)) * (Gd * 8), 0);
  }
  // ---------------- 3. hand-off: the XCD's activations, polled by the lanes that convert them (two granules = one 4-value item) ----------------
  const int nit2 = IX >> 2;
  if ((tid & ~63) < nit2) {
    const int it = min(tid, nit2 - 1);
    unsigned long long xa, xb;
    for (unsigned spins = 0;;) {
      xa = __hip_atomic_load(hand + 2 * it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      xb = __hip_atomic_load(hand + 2 * it + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__all(static_cast<unsigned>(xa >> 32) == tag && static_cast<unsigned>(xb >> 32) == tag)) break;
      if (++spins > a.spin_limit) {                               // status word + the accumulator's non-finite marker: the stream turns NaN in-band
        if (lane == 0) {
          a.state[8 * 16] = 3u;
          __hip_atomic_fetch_add(reinterpret_cast<dml_gu64*>(reinterpret_cast<uintptr_t>(a.acc_out)) + a.D, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    DML_STAMP(5);
    const unsigned w0 = static_cast<unsigned>(xa), w1 = static_cast<unsigned>(xb);
    const float v[4] = {act_lo<BF16>(w0), act_hi<BF16>(w0), act_lo<BF16>(w1), act_hi<BF16>(w1)};
    float sum = (v[0] + v[1]) + (v[2] + v[3]);
    unsigned amax = max(max(__float_as_uint(v[0]) & 0x7FFFFFFFu, __float_as_uint(v[1]) & 0x7FFFFFFFu),
                        max(__float_as_uint(v[2]) & 0x7FFFFFFFu, __float_as_uint(v[3]) & 0x7FFFFFFFu));
    sum += dpp_f32<0xB1>(sum); amax = max(amax, dpp_u32<0xB1>(amax));
    sum += dpp_f32<0x4E>(sum); amax = max(amax, dpp_u32<0x4E>(amax));
    sum += dpp_f32<0x141>(sum); amax = max(amax, dpp_u32<0x141>(amax));
    if (amax >= 0x7F800000u) sum = __uint_as_float(0x7FC00000u);     // Inf / NaN in the block: NaN through the zero-point term of every row
    int ea = static_cast<int>(amax >> 23);
    ea = min(max(ea, 20), 240);
    const float up = exp2i(134 - ea);
    if (tid < nit2) {
      const int b = it >> 3, u = it & 7;
      const int m = b >> 2, kb = b & 3;
      const int ga = 2 * (kb & 1) + (u >> 2);
      put_item(img2 + (kb >> 1) * 1024 + (m + 16 * ga) * 16, u, v[0] * up, v[1] * up, v[2] * up, v[3] * up);
      if (u == 0) { bsum2[m + 16 * kb] = sum; scl2[m + 16 * kb] = static_cast<unsigned char>(ea - 7); }
    }
  }
  // (groups beyond this XCD's slice -- operand rows / columns >= ng -- hold whatever LDS held: their products stay in result rows
  //  and columns >= ng, which no lane selects below)
  __syncthreads();
  DML_STAMP(6);
  // ---------------- 4. down_proj over this XCD's columns: one packed row slice per scaled-MFMA operand, 2 limbs ----------------
  {
    dml_v8i a2[2];
#pragma unroll
    for (int l = 0; l < 2; ++l) {
      const u32x4 p0 = *reinterpret_cast<const u32x4*>(img2 + 0 * 1024 + l * 2048 + lane * 16);
      const u32x4 p1 = *reinterpret_cast<const u32x4*>(img2 + 1 * 1024 + l * 2048 + lane * 16);
      a2[l][0] = p0.x; a2[l][1] = p0.y; a2[l][2] = p0.z; a2[l][3] = p0.w;
      a2[l][4] = p1.x; a2[l][5] = p1.y; a2[l][6] = p1.z; a2[l][7] = p1.w;
    }
    const bool lg = nn < ng;                                     // this lane's group exists in the slice
    const int sa2 = static_cast<int>(scl2[lane]);
    const float gs2 = (bsum2[nn] + bsum2[nn + 16]) + (bsum2[nn + 32] + bsum2[nn + 48]);
    const float xsg2 = (diag && lg) ? gs2 : 0.f;
    float yv[8];
    if constexpr (DWL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's own rows have landed in ITS part of LDS
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      WT wv;
      if constexpr (BITS == 4) wv = *reinterpret_cast<const u32x4*>(dwl + i * 1024 + lane * 16);
      else if constexpr (BITS == 3) { const u32x4 t4 = *reinterpret_cast<const u32x4*>(dwl + i * 1024 + lane * 16); wv = u32x3{t4.x, t4.y, t4.z}; }
      else wv = dw[i];
      const dml_v4f z = mx_row(a2[0], a2[1], wv, 0, sa2);
      const float dsel = fmaf(z[0], dm[0], fmaf(z[1], dm[1], fmaf(z[2], dm[2], z[3] * dm[3])));
      const unsigned szw = (i & 1) ? dsz[i >> 1].y : dsz[i >> 1].x;
      const float sc = sz_scale<BF16>(szw), ze = sz_zero<BF16>(szw);
      yv[i] = lg ? fmaf(ze, xsg2, (sc + sc) * dsel) : 0.f;
    }
    dml_gu64* accp = reinterpret_cast<dml_gu64*>(reinterpret_cast<uintptr_t>(a.acc_out));
    const int row0 = drow0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (2 * q >= rows_w) break;
      const float v = wave_sum2(yv[2 * q], yv[2 * q + 1]);     // row 2 q in lane 31, row 2 q + 1 in lane 63
      if ((lane & 31) == 31) {
        const int row = row0 + 2 * q + (lane >> 5);
        if (fabsf(v) < 2147483648.f) {
          const long long fx = __float2ll_rn(v * 4294967296.f);
          __hip_atomic_fetch_add(accp + row, static_cast<unsigned long long>(fx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          __hip_atomic_fetch_add(accp + a.D, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
  }
  if (r == 0 && tid == 0) __hip_atomic_store(gen_p, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  DML_STAMP(7);
}

// ---- the last block's down_proj sums -> the residual stream the output head reads; every accumulator of the token is left zero ----
template <bool BF16>
__global__ __launch_bounds__(1024) void acc_finish_kernel(const unsigned short* __restrict__ x, unsigned long long* __restrict__ acc,
                                                          unsigned short* __restrict__ out, int D, unsigned long long* z0,
                                                          unsigned long long* z1, unsigned long long* z2) {
  const bool poison = acc[D] != 0ull;
  __syncthreads();                                               // (one workgroup: every thread has read the marker before it is cleared)
  for (int i = threadIdx.x; i <= D; i += blockDim.x) {
    if (i < D) {
      const unsigned long long av = acc[i];
      const float o = round_bf16_or_f16(fx32_to_f32(static_cast<unsigned>(av), static_cast<unsigned>(av >> 32)), BF16);
      float v = act_to_f32<BF16>(x[i]) + o;
      if (poison) v = __uint_as_float(0x7FC00000u);
      out[i] = f32_to_act<BF16>(v);
    }
    acc[i] = 0ull;
    if (z0) z0[i] = 0ull;
    if (z1) z1[i] = 0ull;
    if (z2) z2[i] = 0ull;
  }
}

bool dmlp_plan(int D, int I, int bits_gu, int group_gu, int bits_dn, int group_dn, bool bf16, int* bpw, int* a4) {
  if (!xcd_layout_ok()) return false;                                       // 8 XCDs x 32 CUs, all available, round-robin placement (xcd_layout.hip)
  // 2 bits, or (round 6) 4 bits as pairs of 2-bit fields (GBX_DMLP_MX4=0: the 4-bit launch stays on the vector form, dmlpv.hip)
  static const bool mx4 = env_switch("GBX_DMLP_MX4") != 0;
  if (!bf16 || bits_gu != bits_dn || !(bits_gu == 2 || bits_gu == 3 || (bits_gu == 4 && mx4)) || group_gu != 128 || group_dn != 128) return false;
  // (3 bits: only for rows in the plane layout -- gbx_dmlp.weight_layout, checked by the caller of this plan)
  if (D > 4096 || D <= 2048 || (D % 1024) != 0) return false;               // two 2048-value spans, the second one full or half (4096: 8B; 3072: 3.2-3B)
  if (I <= 0 || (I % (8 * 128)) != 0) return false;
  const int IX = I / 8;
  if (IX > 2048 || IX < 256) return false;                                  // one span of down's operand; at least one polling wave
  if ((IX % DA_MEMBERS) != 0) return false;
  const int b = IX / DA_MEMBERS;                                            // batches per workgroup
  if (b < 32 || b > 64 || (b & 1)) return false;                            // 4 a4 + 2 (16 - a4) = b
  if ((D % (DA_MEMBERS * DA_NW)) != 0 || D / (DA_MEMBERS * DA_NW) > 8 || ((D / (DA_MEMBERS * DA_NW)) & 1)) return false;
  if (static_cast<long long>(I) * (D / 8 * bits_gu) >= 0x7FFFFFFFll || static_cast<long long>(D) * (I / 8 * bits_dn) >= 0x7FFFFFFFll) return false;
  *bpw = b;
  *a4 = (b - 32) / 2;
  return true;
}

hipError_t launch_dmlp(const DmlpArgs& a, int bits, hipStream_t stream) {
  static DeviceOnce once;
  if (once.pending()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(dmlp_kernel<2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, DM_LDS_BYTES);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(dmlp_kernel<2, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, DM_LDS_BYTES4);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(dmlp_kernel<2, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, DM_LDS_BYTES4);
    if (e != hipSuccess) return e;
    once.done();
  }
  if (a.K > 4096 || a.K <= 2048 || (a.K % 1024) != 0 || bits < 2 || bits > 4) return hipErrorInvalidValue;
  if (bits == 3) hipLaunchKernelGGL((dmlp_kernel<2, 3>), dim3(DA_GROUPS * DA_MEMBERS), dim3(DA_NT), DM_LDS_BYTES4, stream, a);
  else if (bits == 4) hipLaunchKernelGGL((dmlp_kernel<2, 4>), dim3(DA_GROUPS * DA_MEMBERS), dim3(DA_NT), DM_LDS_BYTES4, stream, a);
  else hipLaunchKernelGGL((dmlp_kernel<2, 2>), dim3(DA_GROUPS * DA_MEMBERS), dim3(DA_NT), DM_LDS_BYTES, stream, a);
  return hipGetLastError();
}

hipError_t launch_acc_finish(const void* x, void* acc, void* out, int D, bool bf16, void* z0, void* z1, void* z2, hipStream_t stream) {
  if (bf16)
    hipLaunchKernelGGL(acc_finish_kernel<true>, dim3(1), dim3(1024), 0, stream, static_cast<const unsigned short*>(x),
                       static_cast<unsigned long long*>(acc), static_cast<unsigned short*>(out), D, static_cast<unsigned long long*>(z0),
                       static_cast<unsigned long long*>(z1), static_cast<unsigned long long*>(z2));
  else
    hipLaunchKernelGGL(acc_finish_kernel<false>, dim3(1), dim3(1024), 0, stream, static_cast<const unsigned short*>(x),
                       static_cast<unsigned long long*>(acc), static_cast<unsigned short*>(out), D, static_cast<unsigned long long*>(z0),
                       static_cast<unsigned long long*>(z1), static_cast<unsigned long long*>(z2));
  return hipGetLastError();
}

}  // namespace gbx
