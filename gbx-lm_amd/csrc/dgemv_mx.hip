// Decode-block GEMV with the dot products on the FP4 x FP8 matrix cores (2 bits, group 128, bf16, K = 2048 SP, SP <= 4).
//
// Same role and launch structure as dgemv_kernel (dgemv_impl.h: mx.quantized_matmul at M == 1, gbx_lm/models/quantized_linear_gba.py:195-203;
// a wave owns whole rows, compact arguments, two-row batches, vector epilogue, carried RMSNorm statistic, weights requested behind
// the activation image), but the per-weight arithmetic is gemv_mxd.hip's (round 3, experimental library): ONE packed row is the B
// operand of v_mfma_scale_f32_16x16x128_f8f6f4 -- 16 consecutive 128-value groups as the 16 columns, a 2-bit field masked in place
// in its nibble is the FP4 (E2M1) number q / 2 -- the activation vector is the A operand as two FP8 (E4M3) limbs with one E8M0
// exponent per 32 values (the block maximum is scaled into [128, 256): an element >= 2^-9 of its block's maximum is carried EXACTLY by
// the two limbs, a smaller one with an absolute error <= 2^-17 of the block maximum -- tests/test_dgemv_gpu.py pins both; a NaN / Inf
// activation turns its block's sum into NaN, which reaches every row through the zero-point term), and the DIAGONAL of the 16 x 16
// result holds the 16 per-group sums of x . q.
// Why now: with the launch overhead cut (dgemv), the widest launch (gate|up: 28 rows per SIMD) is bound by its vector-ALU work --
// one v_and + one half-rate v_dot2c per pair of weights, ~3.4 k VALU instructions per SIMD = 4-5 us of a 9.1 us launch, the same
// with the weight stream removed (profiles/r04_chain_floor.txt) -- and 4 MFMAs + ~12 VALU instructions per row replace ~70.
// Operand slot map (measured in round 3, scratch/probe_mx2.hip): A lane (m, ga) byte j pairs with B lane (n, gb = (ga >> 1) + 2 (j >> 4)),
// dword 2 (ga & 1) + ((j & 15) >> 3), nibble j & 7; the E8M0 scale of k-block gb of row m is read from A lane (m, gb).
#include "dgemv_impl.h"
#include "gbx_env.h"
#include <type_traits>

namespace gbx {

typedef int dmx_v8i __attribute__((ext_vector_type(8)));
typedef float dmx_v4f __attribute__((ext_vector_type(4)));

template <int SP>
struct DmxBatch {
  u32x2 w[2][SP];          // 64 packed fields per (row, span) and lane
  u32x2 sz[SP];            // (scale | zero << 16) of group 16 h + (lane & 15): .x row 0, .y row 1 of the batch
  unsigned res;
};

template <int SP>
__host__ __device__ constexpr size_t dmx_lds_bytes(int nw) {
  return static_cast<size_t>(SP) * 4096 + static_cast<size_t>(SP) * 64 * 4 + static_cast<size_t>(SP) * 64 + static_cast<size_t>(nw) * 4 + 64;
}

// SP = 2048-value spans per row; NW waves per workgroup; EPI as dgemv_kernel (0 plain, 1 residual, 2 SwiGLU: row 0 of a batch from
// qw = gate, row 1 from qw2 = up); LATE: batch 0 requested behind the activation image.  Two rows per batch, two register buffers.
template <int SP, int NW, int EPI, bool LATE>
__global__ __launch_bounds__(NW * 64) void dgemv_mx_kernel(const DgArgs a) {
  constexpr bool BF16 = true, SWI = EPI == 2;
  constexpr int NT = NW * 64, R = 2;
  constexpr int RO = SWI ? 1 : 2;
  constexpr int NBLK = SP * 64;                                  // 32-value blocks of the activation vector
  constexpr unsigned KILL = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* img = smem;                                     // [span][limb][piece][64 lanes][16 B]
  float* bsum = reinterpret_cast<float*>(smem + SP * 4096);      // [span][m + 16 kb]: sum of the block's values
  unsigned char* scl = reinterpret_cast<unsigned char*>(bsum + NBLK);   // [span][m + 16 kb]: E8M0 byte
  float* red_lds = reinterpret_cast<float*>(scl + NBLK);         // [NW] RMSNorm partials

  const int K = a.K;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef GBX_STAMPS
#define DMX_STAMP(i) do { if (lane == 0 && a.stamps) a.stamps[(static_cast<size_t>(blockIdx.x) * NW + wave) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define DMX_STAMP(i) do {} while (0)
#endif
  DMX_STAMP(0);
  const int W4 = K / 4;                                          // bytes per packed row (2 bits)
  const int G = K >> 7;
  const int ow = wave;
  const int owners = static_cast<int>(gridDim.x) * NW;
  const int bpo = a.bpo, nbatch = a.nbatch;
  auto batch_of = [&](int j) {
    if (j < bpo - 1) return j * owners + static_cast<int>(blockIdx.x) * NW + ow;
    if (j == bpo - 1 && ow < a.apw) return j * owners + static_cast<int>(blockIdx.x) * a.apw + ow;
    return nbatch;
  };
  constexpr int RS_FLAGS = 0x00020000, RS_ALL = 0x7FFFFFFF;
  auto rsrc = [](const void* ptr) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, RS_ALL, RS_FLAGS); };
  const __amdgpu_buffer_rsrc_t qd = rsrc(a.qw), qd2 = rsrc(SWI ? a.qw2 : a.qw), szd = rsrc(a.sz);
  const __amdgpu_buffer_rsrc_t resd = rsrc(EPI == 1 ? static_cast<const void*>(a.residual) : static_cast<const void*>(a.y));
  const int nn = lane & 15, kg = lane >> 4;
  const unsigned voff_w = nn * 32 + kg * 8;                      // this lane's 8 bytes inside a 512-byte span
  const unsigned voff_s = nn * 8;                                // its group's (row 0, row 1) pair inside a span's 16 groups
  const unsigned voff_r = (lane & 32) ? 2 : 0;

  auto issue = [&](DmxBatch<SP>& b, int bi) {
    const bool live = bi < nbatch;
    const int bc = (live && !a.null_rows) ? bi : 0;
    const unsigned kill = live ? 0u : KILL;
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int row = SWI ? bc : bc * R + i;
      const int woff = row * W4;
      const __amdgpu_buffer_rsrc_t d = (SWI && i == 1) ? qd2 : qd;
#pragma unroll
      for (int h = 0; h < SP; ++h) b.w[i][h] = __builtin_amdgcn_raw_buffer_load_b64(d, (voff_w + h * 512) | kill, woff, 2);   // nt
    }
    const int soff = (live ? bi : 0) * (G * 8);
#pragma unroll
    for (int h = 0; h < SP; ++h) b.sz[h] = __builtin_amdgcn_raw_buffer_load_b64(szd, (voff_s + h * 128) | kill, soff, 0);
    if constexpr (EPI == 1) b.res = __builtin_amdgcn_raw_buffer_load_b16(resd, voff_r | kill, (live ? bi : 0) * 4, 0);
  };

  // ---------------- 1. activation (and norm weight) loads first; item = 4 consecutive values ----------------
  const unsigned short* __restrict__ xg = a.x;
  const bool norm = a.norm_w != nullptr;
  const unsigned short* __restrict__ wg = norm ? a.norm_w : xg;
  constexpr int NIT = SP * 512;                                   // items of 4 values
  // Round 5: in a 16-wave workgroup the LATE (RMSNorm) prologue runs on waves 0-7 only.  The waves of a 1024-thread workgroup start
  // ~0.35 us apart (profiles/r05_layer_timeline.txt) and everything up to the image barrier used to wait for the last one's loads;
  // the early half now carries two items per thread, the late half only joins barrier A (= "x has landed": its cue to request
  // batch 0 without delaying x) and barrier B (image written).
  constexpr int PW = (NW == 16 && LATE) ? 8 : NW;
  constexpr int PT = PW * 64;
  constexpr int IPT = (NIT + PT - 1) / PT;
  const bool pro = PW == NW ? true : wave < PW;                   // wave-uniform
  u32x2 xr[IPT], wr[IPT];
  u32x4 acv[IPT][2];
  unsigned apoison = 0;
  u32x4 pq[4];
  const bool carried = norm && a.ssq_in != nullptr;
  if (a.zero_buf) {
    for (int i = static_cast<int>(blockIdx.x) * NT + tid; i < a.zero_n; i += static_cast<int>(gridDim.x) * NT) a.zero_buf[i] = 0ull;
  }
  DmxBatch<SP> ring[2];
  if (pro) {
    // the fixed-point o_proj sums of an x_acc launch first (the longest of the prologue's round trips; unconditional issue: without
    // x_acc the offsets are out of range and nothing is fetched)
    {
      const __amdgpu_buffer_rsrc_t ad = rsrc(a.x_acc ? static_cast<const void*>(a.x_acc) : static_cast<const void*>(a.x));
      const unsigned ak = a.x_acc ? 0u : KILL;
#pragma unroll
      for (int j = 0; j < IPT; ++j) {
        const int it = min(tid + PT * j, NIT - 1);
        acv[j][0] = __builtin_amdgcn_raw_buffer_load_b128(ad, static_cast<unsigned>(it * 32) | ak, 0, 0);
        acv[j][1] = __builtin_amdgcn_raw_buffer_load_b128(ad, static_cast<unsigned>(it * 32 + 16) | ak, 0, 0);
      }
      apoison = __builtin_amdgcn_raw_buffer_load_b32(ad, static_cast<unsigned>(K * 8) | ak, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
      const int it = min(tid + PT * j, NIT - 1);
      xr[j] = *reinterpret_cast<const u32x2*>(xg + 4 * it);
      wr[j] = *reinterpret_cast<const u32x2*>(wg + 4 * it);
    }
    {
      const __amdgpu_buffer_rsrc_t pd = rsrc(carried ? static_cast<const void*>(a.ssq_in) : static_cast<const void*>(a.x));
      const unsigned pk = carried ? 0u : KILL;
#pragma unroll
      for (int i = 0; i < 4; ++i) pq[i] = __builtin_amdgcn_raw_buffer_load_b128(pd, static_cast<unsigned>(lane * 16 + 1024 * i) | pk, 0, 0);
    }
    if constexpr (!LATE) issue(ring[0], batch_of(0));             // (PW == NW here: every wave is a prologue wave)
    DMX_STAMP(1);
    if (a.x_acc) {
      // x <- round(x + round(acc * 2^-32)) (dgemv_impl.h: the fused attention launch's o_proj sums; workgroup 0 writes the new stream)
      const bool poison = apoison != 0u;
#pragma unroll
      for (int j = 0; j < IPT; ++j) {
        const int it = min(tid + PT * j, NIT - 1);
        const bool live = (tid + PT * j) < NIT;
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
        }
        xr[j] = u32x2{wv[0], wv[1]};
        if (blockIdx.x == 0 && live && a.x_out) *reinterpret_cast<u32x2*>(a.x_out + 4 * it) = xr[j];
      }
    }
    DMX_STAMP(8);
    if (norm && !carried) {
      float ssq = 0.f;
#pragma unroll
      for (int j = 0; j < IPT; ++j) {
        const bool live = (tid + PT * j) < NIT;
        const unsigned w2[2] = {xr[j].x, xr[j].y};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const float lo = act_lo<BF16>(w2[q]), hi = act_hi<BF16>(w2[q]);
          ssq += live ? fmaf(hi, hi, lo * lo) : 0.f;
        }
      }
      ssq = wave_sum(ssq);
      if (lane == 0) red_lds[wave] = ssq;
    }
    if constexpr (PW != NW) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // barrier A below says: the activations have landed
  }
  // ---------------- 2. RMSNorm statistic (carried from the producer of x, or reduced here); barrier A ----------------
  if (PW != NW || (norm && !carried)) __syncthreads();
  float rstd = 1.f;
  if (pro && norm) {
    float tot;
    if (carried) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        s += (__uint_as_float(pq[i].x) + __uint_as_float(pq[i].y)) + (__uint_as_float(pq[i].z) + __uint_as_float(pq[i].w));
      tot = wave_sum(s);
    } else {
      tot = 0.f;
#pragma unroll
      for (int i = 0; i < PW; ++i) tot += red_lds[i];
    }
    rstd = rsqrtf(tot / static_cast<float>(K) + a.eps);
  }
  DMX_STAMP(9);
  // ---------------- 3. the limb image: x' = x 2^-e per 32-value block, hi = fp8(x'), lo = fp8(x' - hi) ----------------
#pragma unroll
  for (int j = 0; j < IPT; ++j) {
    if (pro && (tid & ~63) + PT * j < NIT) {                      // (wave-uniform) idle waves skip the arithmetic
      const int it = tid + PT * j;
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
      if (amax >= 0x7F800000u) sum = __uint_as_float(0x7FC00000u);   // Inf / NaN in the block: NaN through the zero-point term of every row (documented: DESIGN 3.1d)
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
  DMX_STAMP(10);
  if constexpr (LATE) issue(ring[0], batch_of(0));
  DMX_STAMP(11);
  __syncthreads();
  DMX_STAMP(2);
  issue(ring[1], batch_of(1));
  if (batch_of(0) >= nbatch && !a.ssq_out) return;               // a wave without any batch was only here for the prologue

  // ---------------- 4. lane-resident operands: A limbs, block scales, group sums (diagonal lanes only) ----------------
  dmx_v8i areg[SP][2];
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
  // dm[q] = 1 on the lane that holds D[n][n] in accumulator register q, 0 elsewhere (arithmetic pick: no exec-mask branches)
  float dm[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) dm[q] = (diag && (lane & 3) == q) ? 1.f : 0.f;

  // ---------------- 5. one batch: 2 rows x SP spans x 2 limbs on the matrix cores, then dgemv's vector epilogue ----------------
  float sq = 0.f;
  auto consume = [&](const DmxBatch<SP>& cur, int j) {
    const int b = batch_of(j);
    if (b >= nbatch) return;
    dmx_v4f c[R][SP];
#pragma unroll
    for (int i = 0; i < R; ++i) {
      auto span = [&](auto H) {
        constexpr int h = decltype(H)::value;
        dmx_v8i bv;
        const unsigned w0 = cur.w[i][h].x, w1 = cur.w[i][h].y;
        bv[0] = w0 & 0x33333333u; bv[1] = (w0 >> 2) & 0x33333333u; bv[2] = w1 & 0x33333333u; bv[3] = (w1 >> 2) & 0x33333333u;
        bv[4] = 0; bv[5] = 0; bv[6] = 0; bv[7] = 0;
        dmx_v4f z = {0.f, 0.f, 0.f, 0.f};
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
        yv += fmaf(ze, xsg[h], (sc + sc) * dsel);                 // the FP4 codes are q / 2; xsg, dsel are 0 off the diagonal
      }
      acc[i] = yv;
    }
    const float v0 = wave_sum2(acc[0], acc[1]);                  // row 0 total in lane 31, row 1 total in lane 63
    if constexpr (SWI) {
      const float t0 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v0), 31));
      const float t1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v0), 63));
      const float g = round_bf16_or_f16(t0, BF16), up = round_bf16_or_f16(t1, BF16);
      const float sl = round_bf16_or_f16(g / (1.f + __expf(-g)), BF16);
      if (lane == 0) a.y[b] = f32_to_act<BF16>(sl * up);
    } else if ((lane & 31) == 31) {
      float v = v0;
      if constexpr (EPI == 1) v = act_to_f32<BF16>(static_cast<unsigned short>(cur.res & 0xFFFFu)) + round_bf16_or_f16(v, BF16);
      const unsigned short o = f32_to_act<BF16>(v);
      a.y[b * R + (lane >> 5)] = o;
      const float f = act_to_f32<BF16>(o);
      sq = fmaf(f, f, sq);
    }
  };
  int base = 0;
  for (; base + 2 < bpo; base += 2) {
    consume(ring[0], base);
    issue(ring[0], batch_of(base + 2));
    consume(ring[1], base + 1);
    issue(ring[1], batch_of(base + 3));
  }
  consume(ring[0], base);
  DMX_STAMP(4);
  consume(ring[1], base + 1);
  DMX_STAMP(6);
  if (a.ssq_out) {
    const float ws = wave_sum(sq);
    if (lane == 0) a.ssq_out[blockIdx.x * NW + wave] = ws;
  }
}

// Which launches run here (GBX_DG_MX: 0 none, 1 the SwiGLU launches = gate|up, 2 every eligible launch)
bool dg_mx_wanted(const DgArgs& a, const DgPlan& pl, int epi, bool bf16) {
  static const int mode = env_switch("GBX_DG_MX");
  // (K = 8192 -- four spans of operands in registers -- builds with > 100 spilled registers at 16 waves per CU: scratch/kernel_resources.py;
  //  those launches stay on the vector form)
  if (!mode || !bf16 || pl.KS != 1 || pl.R != 2 || a.log2g != 7 || (a.K % 2048) != 0 || a.K > 6144) return false;
  if (mode == 1 && epi != 2) return false;
  const int sp = a.K / 2048;
  return sp <= 2 ? (pl.NW == 16 || pl.NW == 8 || pl.NW == 4) : pl.NW == 16;
}

hipError_t launch_dgemv_mx(const DgArgs& a, const DgPlan& pl, int epi, hipStream_t stream) {
  DgArgs args = a;
  args.nbatch = pl.nbatch; args.bpo = pl.bpo; args.apw = pl.apw;
  const bool late = a.norm_w != nullptr && dg_late_default();
  const dim3 grid(pl.grid);
  const int sp = a.K / 2048;
#define DMX_CASE(SP_, NW_, EPI_)                                                                                                     \
  if (sp == SP_ && pl.NW == NW_ && epi == EPI_) {                                                                                    \
    const size_t shm = dmx_lds_bytes<SP_>(NW_);                                                                                      \
    if (late && EPI_ != 1) hipLaunchKernelGGL((dgemv_mx_kernel<SP_, NW_, EPI_, (EPI_ != 1)>), grid, dim3(NW_ * 64), shm, stream, args); \
    else hipLaunchKernelGGL((dgemv_mx_kernel<SP_, NW_, EPI_, false>), grid, dim3(NW_ * 64), shm, stream, args);                       \
    return hipGetLastError();                                                                                                        \
  }
#define DMX_EPIS(SP_, NW_) DMX_CASE(SP_, NW_, 0) DMX_CASE(SP_, NW_, 1) DMX_CASE(SP_, NW_, 2)
  DMX_EPIS(1, 4) DMX_EPIS(1, 8) DMX_EPIS(1, 16)
  DMX_EPIS(2, 4) DMX_EPIS(2, 8) DMX_EPIS(2, 16)
  DMX_EPIS(3, 16)
  return hipErrorInvalidValue;
}

}  // namespace gbx
