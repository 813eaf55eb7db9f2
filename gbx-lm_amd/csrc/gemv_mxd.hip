// Decode GEMV (M = 1), 2 bits, group 128, bf16, K a multiple of 2048: the dot products of gemv_kernel on the FP4 x FP8 matrix cores,
// with gemv_kernel's launch structure kept as it is (a wave owns whole rows, packed rows go global -> VGPR as contiguous bursts, one
// activation prologue per workgroup, 2-deep ring of row batches, the same fusions).  Round 3, opt-out GBX_GEMV_MXD=0.
//
// Why: the decode launches are bound by their heads and by the vector-ALU work of consuming the packed fields (one v_and + one
// half-rate v_dot2c per PAIR of weights: ~170 VALU instructions per 2-row batch and wave at K = 4096; with the weight stream removed
// the launches are no faster, profiles/r03_launch_heads.txt).  gemv_mx.hip moved that work to v_mfma_scale_f32_16x16x128_f8f6f4 but
// fed it 16 ROWS per operand, which needs an LDS detour for the weights and loader waves -- slower.  Here ONE row feeds the
// instruction: its B operand is 16 consecutive 128-value segments of the same packed row (= 16 quantisation groups, 512 contiguous
// bytes, lane (n, kg) loads the 8 bytes at 32 n + 8 kg: still one contiguous burst per wave-load), its A operand the 16 matching
// segments of the activation vector, and only the DIAGONAL of the 16 x 16 result is used: D[n][n] = sum over group n of x . q.  15/16
// of the multiply-adds are wasted, on a pipe that idles otherwise: 4 MFMAs (2 spans x 2 limbs, ~32 cycles each) replace ~128 VALU
// instructions per row.
//   * B = the packed words as they are: a 2-bit field masked in place in its nibble is the FP4 (E2M1) number q / 2 (gemv_mx.hip):
//     (w & 0x33333333, (w >> 2) & 0x33333333) per word, 6 VALU instructions per 64 weights.
//   * A = the activations as TWO FP8 (E4M3) limbs, hi = fp8(x'), lo = fp8(x' - hi), x' = x * 2^-e with one E8M0 exponent per
//     32-value block (the instruction's own block scale): a bf16 value (8 significant bits) splits EXACTLY into 4 + 4 bits; values
//     more than 2^10 below their block's maximum lose low bits (absolute error <= 2^-17 of the block maximum).  fp16 activations
//     (11 bits) would need a third limb: they stay on gemv_kernel.
//   * operand slot map (measured, scratch/probe_mx2.hip; gemv_mx.hip runs on it): A lane (m, ga) byte j pairs with B lane
//     (n, gb = (ga >> 1) + 2 (j >> 4)), dword 2 (ga & 1) + ((j & 15) >> 3), nibble j & 7; the E8M0 scale of k-block gb of row m is
//     read from A lane (m, gb).  With B lane (n, gb) holding values 32 gb .. 32 gb + 31 of segment n as (even fields of word 0, odd
//     fields of word 0, even / odd of word 1), byte j of A lane (m, ga) is value 32 gb + 16 (ga & 1) + 2 (j & 7) + ((j & 15) >> 3)
//     of segment m, and the scale blocks are the natural 32-value blocks.
//   * flush: the 16 lanes that hold a diagonal element ((lane >> 4) == (lane & 15) >> 2, element lane & 3) apply the group's
//     scale and zero in fp32 exactly like every other kernel: y += 2 s D[n][n] + z sum_{k in group} x; the wave reduction and the
//     epilogues (residual, SwiGLU, up to three matrices) are gemv_kernel's.
#include "gemv_impl.h"
#include "kernels.h"
#include <type_traits>

namespace gbx {

typedef int mxd_v8i __attribute__((ext_vector_type(8)));
typedef float mxd_v4f __attribute__((ext_vector_type(4)));

template <int SP>
struct MxdBatch {
  u32x2 w[GEMV_R][SP];      // 64 packed fields per (row, span) and lane
  u32x2 sz[SP];             // (scale | zero << 16) of group 16 h + (lane & 15): .x row 0, .y row 1 of the batch
  unsigned res2;            // residual of both rows
};

template <int SP>
__host__ __device__ constexpr size_t mxd_lds_bytes(int nw) {
  return static_cast<size_t>(SP) * 4096 + static_cast<size_t>(SP) * 64 * 4 + static_cast<size_t>(SP) * 64 + static_cast<size_t>(nw) * 4 + 64;
}

// SP = 2048-value spans per row (K = 2048 SP), NW = waves per workgroup, D = row batches in flight per wave.
template <int SP, int NW, int D>
__global__ __launch_bounds__(NW * 64) void gemv_mxd_kernel(const GemvArgs a) {
  constexpr bool BF16 = true;
  constexpr int NT = NW * 64, R = GEMV_R;
  constexpr int NBLK = SP * 64;                                  // 32-value blocks of the activation vector
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* img = smem;                                     // [span][limb][piece][64 lanes][16 B]
  float* bsum = reinterpret_cast<float*>(smem + SP * 4096);      // [span][m + 16 kb]: sum of the block's values
  unsigned char* scl = reinterpret_cast<unsigned char*>(bsum + NBLK);   // [span][m + 16 kb]: E8M0 byte
  float* red = reinterpret_cast<float*>(scl + NBLK);             // [NW] RMSNorm partials

  const int K = a.K, N = a.N;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int W = K / 16, G = K / 128;
  const bool swiglu = a.swiglu != 0;
  const int rowstep = swiglu ? 1 : R;
  const int bpo = a.batches_per_owner;
  const int owner = blockIdx.x * NW + wave, owners = static_cast<int>(gridDim.x) * NW;
  const int long_waves = a.long_waves;
  const int my_n = (long_waves && wave >= long_waves) ? bpo - 1 : bpo;
  auto batch_of = [&](int j) {
    if (long_waves && j == bpo - 1)
      return wave < long_waves ? (bpo - 1) * owners + static_cast<int>(blockIdx.x) * long_waves + wave : (N + rowstep - 1) / rowstep;
    return owner + j * owners;
  };
  const int n0 = a.seg[0].N, n01 = n0 + a.seg[1].N;
  constexpr int RS_FLAGS = 0x00020000, RS_ALL = 0x7FFFFFFF;
  auto rsrc = [](const void* ptr) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, RS_ALL, RS_FLAGS); };
  const uint32_t* const q0 = a.seg[0].qweight; const uint32_t* const q1 = a.seg[1].qweight; const uint32_t* const q2 = a.seg[2].qweight;
  unsigned short* const y0 = a.seg[0].y; unsigned short* const y1 = a.seg[1].y; unsigned short* const y2 = a.seg[2].y;
  const int n1 = a.seg[1].N, n2 = a.seg[2].N;
  auto locate = [&](int r0, int i, int& sg, int& row) {          // row of batch element i -> (segment, local row); wave-uniform
    if (swiglu) { sg = i; row = min(r0, N - 1); }
    else {
      const int r = min(r0 + i, N - 1);
      sg = (r >= n0) + (r >= n01);
      row = r - (sg == 0 ? 0 : (sg == 1 ? n0 : n01));
    }
  };
  const bool null_rows = a.null_rows != 0;
  const __amdgpu_buffer_rsrc_t resd = rsrc(a.residual ? static_cast<const void*>(a.residual) : static_cast<const void*>(y0));
  const __amdgpu_buffer_rsrc_t pd = rsrc(a.sz);
  const int nn = lane & 15, kg = lane >> 4;
  const int voff_w = nn * 32 + kg * 8;                            // this lane's 8 bytes inside a 512-byte span
  const int voff_s = nn * 8;                                      // its group's (row 0, row 1) pair inside a span's 16 groups

  auto issue = [&](MxdBatch<SP>& b, int r0) {
#pragma unroll
    for (int i = 0; i < R; ++i) {
      int sgi, row;
      locate(r0, i, sgi, row);
      const __amdgpu_buffer_rsrc_t qd = rsrc(sgi == 0 ? q0 : (sgi == 1 ? q1 : q2));
      const int woff = null_rows ? (row & 1) * (W * 4) : row * (W * 4);
#pragma unroll
      for (int h = 0; h < SP; ++h) b.w[i][h] = __builtin_amdgcn_raw_buffer_load_b64(qd, voff_w + h * 512, woff, 2);   // nt
    }
    const int rc = min(r0, N - rowstep);                          // batches past the end re-read the last pair
    const int poff = (swiglu ? rc : (rc >> 1)) * (G * 8);
#pragma unroll
    for (int h = 0; h < SP; ++h) b.sz[h] = __builtin_amdgcn_raw_buffer_load_b64(pd, voff_s + h * 128, poff, 0);
    b.res2 = __builtin_amdgcn_raw_buffer_load_b32(resd, 0, min(rc, n0 - 2) * 2, 0);
  };

  // ---------------- 1. activation (and norm weight) loads first; item = 4 consecutive values ----------------
  const unsigned short* __restrict__ xg = static_cast<const unsigned short*>(a.x);
  const bool norm = a.norm_w != nullptr;
  const unsigned short* __restrict__ wg = norm ? a.norm_w : xg;
  constexpr int NIT = SP * 512;                                   // items of 4 values
  constexpr int IPT = (NIT + NT - 1) / NT;
  u32x2 xr[IPT], wr[IPT];
#pragma unroll
  for (int j = 0; j < IPT; ++j) {
    const int it = min(tid + NT * j, NIT - 1);
    xr[j] = *reinterpret_cast<const u32x2*>(xg + 4 * it);
    wr[j] = *reinterpret_cast<const u32x2*>(wg + 4 * it);
  }
  if (a.zero_buf) {
    for (int i = blockIdx.x * NT + tid; i < a.zero_n; i += static_cast<int>(gridDim.x) * NT) a.zero_buf[i] = 0;
  }
  // ---------------- 2. the first batch of packed rows goes in flight before any activation math ----------------
  MxdBatch<SP> ring[D];
  issue(ring[0], batch_of(0) * rowstep);

  // ---------------- 3. optional RMSNorm statistics, then the limb image ----------------
  float rstd = 1.f;
  if (norm) {
    float ssq = 0.f;
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
      const bool live = (tid + NT * j) < NIT;
      const unsigned w2[2] = {xr[j].x, xr[j].y};
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const float lo = act_lo<BF16>(w2[q]), hi = act_hi<BF16>(w2[q]);
        ssq += live ? fmaf(hi, hi, lo * lo) : 0.f;
      }
    }
    ssq = wave_sum(ssq);
    if (lane == 0) red[wave] = ssq;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) tot += red[i];
    rstd = rsqrtf(tot / static_cast<float>(K) + a.eps);
  }
#pragma unroll
  for (int j = 0; j < IPT; ++j) {
    if ((tid & ~63) + NT * j < NIT) {                             // (wave-uniform) idle waves skip the arithmetic
      const int it = tid + NT * j;
      float v[4] = {act_lo<BF16>(xr[j].x), act_hi<BF16>(xr[j].x), act_lo<BF16>(xr[j].y), act_hi<BF16>(xr[j].y)};
      if (norm) {                                                 // one rounding to the activation dtype, like the reference's RMSNorm output
        v[0] = round_bf16_or_f16(v[0] * rstd * act_lo<BF16>(wr[j].x), BF16);
        v[1] = round_bf16_or_f16(v[1] * rstd * act_hi<BF16>(wr[j].x), BF16);
        v[2] = round_bf16_or_f16(v[2] * rstd * act_lo<BF16>(wr[j].y), BF16);
        v[3] = round_bf16_or_f16(v[3] * rstd * act_hi<BF16>(wr[j].y), BF16);
      }
      float sum = (v[0] + v[1]) + (v[2] + v[3]);
      unsigned amax = max(max(__float_as_uint(v[0]) & 0x7FFFFFFFu, __float_as_uint(v[1]) & 0x7FFFFFFFu),
                          max(__float_as_uint(v[2]) & 0x7FFFFFFFu, __float_as_uint(v[3]) & 0x7FFFFFFFu));
      // the 8 items (lanes) of a 32-value block
      sum += dpp_f32<0xB1>(sum); amax = max(amax, dpp_u32<0xB1>(amax));
      sum += dpp_f32<0x4E>(sum); amax = max(amax, dpp_u32<0x4E>(amax));
      sum += dpp_f32<0x141>(sum); amax = max(amax, dpp_u32<0x141>(amax));
      // x' = x * 2^(134 - ea): the block maximum lands in [128, 256) (E4M3 holds up to 448); the instruction multiplies by 2^(scale - 127)
      int ea = static_cast<int>(amax >> 23);
      ea = min(max(ea, 20), 240);
      const float up = exp2i(134 - ea);
      if (it < NIT) {
        const int b = it >> 3, u = it & 7;
        const int h = b >> 6, m = (b & 63) >> 2, kb = b & 3;
        const int ga = 2 * (kb & 1) + (u >> 2);
        unsigned char* dst = img + (h * 4 + (kb >> 1)) * 1024 + (m + 16 * ga) * 16 + 2 * (u & 3);   // limb 0; limb 1 is 2048 bytes on
        const float r0 = v[0] * up, r1 = v[1] * up, r2 = v[2] * up, r3 = v[3] * up;
        const int pe = __builtin_amdgcn_cvt_pk_fp8_f32(r0, r2, 0, false);        // even fields: bytes 2 (u & 3), + 1
        const int po = __builtin_amdgcn_cvt_pk_fp8_f32(r1, r3, 0, false);        // odd fields:  bytes 8 + 2 (u & 3), + 1
        *reinterpret_cast<unsigned short*>(dst) = static_cast<unsigned short>(pe);
        *reinterpret_cast<unsigned short*>(dst + 8) = static_cast<unsigned short>(po);
        const int le = __builtin_amdgcn_cvt_pk_fp8_f32(r0 - __builtin_amdgcn_cvt_f32_fp8(pe, 0), r2 - __builtin_amdgcn_cvt_f32_fp8(pe, 1), 0, false);
        const int lo = __builtin_amdgcn_cvt_pk_fp8_f32(r1 - __builtin_amdgcn_cvt_f32_fp8(po, 0), r3 - __builtin_amdgcn_cvt_f32_fp8(po, 1), 0, false);
        *reinterpret_cast<unsigned short*>(dst + 2048) = static_cast<unsigned short>(le);
        *reinterpret_cast<unsigned short*>(dst + 2048 + 8) = static_cast<unsigned short>(lo);
        if (u == 0) { bsum[h * 64 + m + 16 * kb] = sum; scl[h * 64 + m + 16 * kb] = static_cast<unsigned char>(ea - 7); }
      }
    }
  }
  __syncthreads();

  // ---------------- 4. lane-resident operands: A limbs, block scales, group sums (diagonal lanes only) ----------------
  mxd_v8i areg[SP][2];
  int sa = 0;
  float xsg[SP];
  const bool diag = kg == (nn >> 2);
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
  // diagonal pick as arithmetic (a nested select here became exec-mask branches with the MFMA result latency exposed per span):
  // dm[q] = 1 on the lane that holds D[n][n] in accumulator register q, 0 elsewhere
  float dm[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) dm[q] = (diag && (lane & 3) == q) ? 1.f : 0.f;

  // ---------------- 5. epilogue of one batch (gemv_kernel's): the two row totals are wave-uniform here ----------------
  auto finish = [&](int r0, float t0, float t1, unsigned short res0, unsigned short res1) {
    if (swiglu) {
      if (r0 < N) {     // bf16 gate, bf16 up, bf16 silu(gate), bf16 product: the reference's op-by-op roundings
        const float g = round_bf16_or_f16(t0, BF16), up = round_bf16_or_f16(t1, BF16);
        const float sl = round_bf16_or_f16(g / (1.f + __expf(-g)), BF16);
        y0[r0] = f32_to_act<BF16>(sl * up);
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
        ysel[row] = f32_to_act<BF16>(v);
      }
    }
  };

  // ---------------- 6. one batch: 2 rows x SP spans x 2 limbs on the matrix cores, all issued before the first flush ----------------
  auto consume = [&](const MxdBatch<SP>& cur, int j) {
    mxd_v4f c[R][SP];
#pragma unroll
    for (int i = 0; i < R; ++i) {
      auto span = [&](auto H) {
        constexpr int h = decltype(H)::value;
        mxd_v8i bv;
        const unsigned w0 = cur.w[i][h].x, w1 = cur.w[i][h].y;
        bv[0] = w0 & 0x33333333u; bv[1] = (w0 >> 2) & 0x33333333u; bv[2] = w1 & 0x33333333u; bv[3] = (w1 >> 2) & 0x33333333u;
        bv[4] = 0; bv[5] = 0; bv[6] = 0; bv[7] = 0;
        mxd_v4f z = {0.f, 0.f, 0.f, 0.f};
        z = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(areg[h][0], bv, z, 0 /*A: fp8 e4m3*/, 4 /*B: fp4*/, h, sa, 0, 127);
        c[i][h] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(areg[h][1], bv, z, 0, 4, h, sa, 0, 127);
      };
      span(std::integral_constant<int, 0>{});
      if constexpr (SP > 1) span(std::integral_constant<int, 1>{});
      if constexpr (SP > 2) span(std::integral_constant<int, 2>{});
      if constexpr (SP > 3) span(std::integral_constant<int, 3>{});
    }
    float acc[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
      float yv = 0.f;
#pragma unroll
      for (int h = 0; h < SP; ++h) {
        const float dsel = fmaf(c[i][h][0], dm[0], fmaf(c[i][h][1], dm[1], fmaf(c[i][h][2], dm[2], c[i][h][3] * dm[3])));
        const unsigned szw = i == 0 ? cur.sz[h].x : cur.sz[h].y;
        const float sc = sz_scale<BF16>(szw), ze = sz_zero<BF16>(szw);
        yv += fmaf(ze, xsg[h], (sc + sc) * dsel);                                 // the FP4 codes are q / 2; xsg, dsel are 0 off the diagonal
      }
      acc[i] = yv;
    }
    const float tot = wave_sum2(acc[0], acc[1]);              // batch row 0 total in lane 31, row 1 total in lane 63
    const float t0 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(tot), 31));
    const float t1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(tot), 63));
    if (lane == 0) finish(batch_of(j) * rowstep, t0, t1, static_cast<unsigned short>(cur.res2 & 0xFFFFu), static_cast<unsigned short>(cur.res2 >> 16));
  };
#pragma unroll
  for (int d = 1; d < D; ++d) issue(ring[d], batch_of(d) * rowstep);
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
#pragma unroll
  for (int d = 0; d < D; ++d) {
    if (base + d < my_n) consume(ring[d], base + d);
  }
}

template <int SP, int NW, int D>
static int mxd_resident_blocks() {
  static int cached = 0;
  if (cached == 0) {
    int per_cu = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gemv_mxd_kernel<SP, NW, D>, NW * 64, mxd_lds_bytes<SP>(NW));
    if (e != hipSuccess || per_cu < 1) per_cu = 1;
    if (per_cu * NW > 32) per_cu = 32 / NW;
    cached = per_cu * GEMV_CUS;
  }
  return cached;
}

template <int SP, int NW, int D>
static hipError_t mxd_launch_d(const GemvArgs& a0, int bpo, int nbatches, int long_waves, hipStream_t s) {
  GemvArgs a = a0;
  a.batches_per_owner = bpo;
  a.long_waves = long_waves;
  static const int chain_floor = [] { const char* e = getenv("GBX_CHAIN_FLOOR"); return e ? atoi(e) : 0; }();
  if (chain_floor == 2) a.null_rows = 1;
  const int blocks = (nbatches + NW * bpo - 1) / (NW * bpo);
  hipLaunchKernelGGL((gemv_mxd_kernel<SP, NW, D>), dim3(blocks), dim3(NW * 64), mxd_lds_bytes<SP>(NW), s, a);
  return hipGetLastError();
}

// Geometry = gemv_kernel's (launch_gemv_nw): one round of co-resident workgroups, ring depth 1 when every wave gets one batch,
// the uneven split when the even one leaves CUs idle.
template <int SP, int NW>
static hipError_t mxd_launch_nw(const GemvArgs& a, int nbatches, hipStream_t s) {
  const int own1 = mxd_resident_blocks<SP, NW, 1>() * NW;
  if (nbatches <= own1) return mxd_launch_d<SP, NW, 1>(a, 1, nbatches, 0, s);
  const int res = mxd_resident_blocks<SP, NW, 2>();
  const int own2 = res * NW;
  int bpo = (nbatches + own2 - 1) / own2;
  bpo = ((bpo + 1) / 2) * 2;
  static const int deep = [] { const char* e = getenv("GBX_MXD_DEEP"); return e ? atoi(e) : 0; }();   // experiment: whole wave's rows in flight
  if (nbatches % res == 0 && (nbatches + NW * bpo - 1) / (NW * bpo) < res) {
    const int lw = nbatches / res - NW * (bpo - 1);
    if (lw > 0 && lw < NW && lw % 4 == 0) {
      if (deep && bpo == 4) return mxd_launch_d<SP, NW, 4>(a, bpo, res * NW * bpo, lw, s);
      return mxd_launch_d<SP, NW, 2>(a, bpo, res * NW * bpo, lw, s);
    }
  }
  if (deep && bpo == 4) return mxd_launch_d<SP, NW, 4>(a, bpo, nbatches, 0, s);
  return mxd_launch_d<SP, NW, 2>(a, bpo, nbatches, 0, s);
}

bool gemv_mxd_eligible(const GemvArgs& a, int M, bool bf16) {
  // GBX_GEMV_MXD: 1 = every eligible launch, 2 = only the fat geometry (more than 4096 row batches: gate|up), unset / 0 = off
  static const int on = [] { const char* e = getenv("GBX_GEMV_MXD"); return e ? atoi(e) : 0; }();
  if (on == 2) {
    const int rowstep = a.swiglu ? 1 : GEMV_R;
    if ((a.N + rowstep - 1) / rowstep <= 4096) return false;
  }
  if (!on || M != 1 || !bf16 || a.log2g != 7 || !a.sz || a.x_acc || a.expert_idx || a.bias) return false;
  if (a.K != 2048 && a.K != 4096) return false;
  for (int i = 0; i < a.nseg; ++i) if (a.seg[i].N & 1) return false;
  return true;
}

hipError_t launch_gemv_mxd(const GemvArgs& a, hipStream_t s) {
  const int rowstep = a.swiglu ? 1 : GEMV_R;
  const int nbatches = (a.N + rowstep - 1) / rowstep;
  const bool small = nbatches <= 4096;
  if (a.K == 2048) return small ? mxd_launch_nw<1, 4>(a, nbatches, s) : mxd_launch_nw<1, 16>(a, nbatches, s);
  return small ? mxd_launch_nw<2, 4>(a, nbatches, s) : mxd_launch_nw<2, 16>(a, nbatches, s);
}

}  // namespace gbx
