// Decode GEMV (M = 1) on the integer matrix cores: y[n] = sum_k x[k] * (s[n,k/g] * q[n,k] + z[n,k/g]), 2-bit, g = 128.
//
// Same operator as gemv_impl.h (gbx_lm/models/quantized_linear_gba.py:195-203 with one activation row); different
// arithmetic engine.  The VALU path spends ~1.5 lane-cycles per weight on "mask + v_dot2c"; at 2 bits that is the
// same order as the HBM time of the packed row, and the two overlap only partially.  Here the multiply-accumulate is
// v_mfma_i32_16x16x64_i8:
//   * A operand (weights): a 2-bit field masked IN PLACE inside its byte is already a valid int8 (q at bits 0-1,
//     4q at bits 2-3), so one packed dword becomes the four A registers of one MFMA with 5 VALU ops
//     (w & 0x03030303, w & 0x0C0C0C0C, the same on w >> 4) -- 0.31 ops per weight instead of 1.5.
//   * B operand (activation): x of one 128-value group is turned into a block-fixed-point integer X = rint(x * 2^E)
//     (|X| < 2^24: every bf16/fp16 value within 2^-16 of the group maximum is represented exactly) and split into four
//     balanced base-128 limbs = four int8 COLUMNS of B.  The 16 columns of a tile are 4 groups x 4 limbs; a lane's
//     K-block belongs to exactly one group, lanes of the other groups' columns hold zeros, so one accumulator tile
//     collects four groups' exact integer sums  D[row][4*group + limb] = sum_k q[row,k] * limb(X[k])  at once.
//     (Positions whose weight byte is 4q carry limb(X), the others limb(4X): every column sums 4*q*X.)
//   * every second K-step (4 groups) the tile is flushed: acc[row] += float(D) * (128^limb * 2^-(E+2)) * s[row,group]
//     and, on the limb-0 lanes, + z[row,group] * sum(x over the group).  fp32 from here on, like the VALU path.
//   * the packed weights are re-tiled ONCE at load time (gbx_prepack_tiles) so that a wave-load is still one
//     contiguous 1 KiB burst: block (tile of 16 rows, K-step of 256 values) holds, for lane l, the 16 bytes of row
//     l%16 at K-block l/16.  scales/zeros come as the packed (s | z << 16) dwords of gbx_pack_scales_zeros re-tiled to
//     [tile][group][16 rows].
// Work split: a workgroup = RG row groups x KS waves; wave (rg, kq) owns K-steps [4kq, 4kq+4) of the tiles of its row
// group (its B registers: 64 VGPRs, loaded once), partial row sums meet in LDS once per workgroup.
#include "gbx_device.h"
#include "kernels.h"

namespace gbx {

typedef int i32x4 __attribute__((ext_vector_type(4)));

struct GemvMfmaArgs {
  const unsigned short* x;        // [K]
  const uint32_t* wt;             // tiled packed weights: [ntiles][nsteps][64 lanes][4 dwords]
  const uint32_t* szt;            // tiled scale|zero dwords: [ntiles][G][16]
  unsigned short* y;              // [N]
  int N, K, ntiles, rg_per_block, ks;
  // ---- fused decode-block options (same meaning as GemvArgs) ----
  const unsigned short* norm_w;   // RMSNorm weight [K]: x <- act(x * rsqrt(mean(x^2) + eps) * w)
  float eps;
  const unsigned short* residual; // [N]: y <- residual + act(result)
  int swiglu;                     // tiles hold 8 gate rows then the 8 up rows of the same outputs: y[n] = silu(gate[n]) * up[n]
};

template <bool BF16> __device__ __forceinline__ float round_act(float v) { return act_to_f32<BF16>(f32_to_act<BF16>(v)); }

// SPW = K-steps (of 256 values) per wave: 4 (64 B-operand VGPRs) for long rows / many tiles, 2 for the narrow projections
// (twice the waves per tile: half the serial work per wave where a row group owns only one or two tiles).
template <bool BF16, int GM_SPW>
__global__ __launch_bounds__(1024) void gemv_mfma_kernel(const GemvMfmaArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int K = a.K, N = a.N;
  const int nsteps = K / 256, G = K / 128;
  const int KS = a.ks, RG = a.rg_per_block;
  // LDS: B images [nsteps][4 m][4 kb][4 limbs][16 B] | group table float2[G] | partial sums
  unsigned char* bimg = smem;
  float2* gtab = reinterpret_cast<float2*>(smem + static_cast<size_t>(nsteps) * 1024);
  float* part = reinterpret_cast<float*>(gtab + G);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rg = wave / KS, kq = wave % KS;
  const int nthreads = blockDim.x;
  const int nrg = static_cast<int>(gridDim.x) * RG;          // row groups on the chip
  const int rgid = blockIdx.x * RG + rg;
  const int tiles_per_rg = (a.ntiles + nrg - 1) / nrg;

  // ---------------- first tile's weights go in flight before the activation prologue -----------------------------
  const __amdgpu_buffer_rsrc_t wd = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(a.wt), 0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t sd = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(a.szt), 0, 0x7FFFFFFF, 0x00020000);
  const int c = lane & 15, rq = lane >> 4;
  u32x4 w[2][GM_SPW], sz[2][GM_SPW / 2];
  auto issue = [&](int set, int j) {
    const int tile = min(rgid + j * nrg, a.ntiles - 1);      // clamped: surplus batches re-read the last tile, never stored
    const int woff = (tile * nsteps + kq * GM_SPW) * 1024;
#pragma unroll
    for (int s = 0; s < GM_SPW; ++s) w[set][s] = __builtin_amdgcn_raw_buffer_load_b128(wd, lane * 16 + s * 1024, woff, 2);
#pragma unroll
    for (int f = 0; f < GM_SPW / 2; ++f) {
      const int g = (kq * GM_SPW + 2 * f) * 2 + (c >> 2);
      sz[set][f] = __builtin_amdgcn_raw_buffer_load_b128(sd, (g * 16 + 4 * rq) * 4, tile * G * 64, 0);
    }
  };
  issue(0, 0);

  // ---------------- activation prologue: x -> block-fixed-point limbs in MFMA B-operand order ---------------------
  // item = (k16, th): the 16 consecutive values k = 16*k16 .. +15 (step k16/16, K-block (k16/4)%4, MFMA k16%4), of which
  // this thread converts those at positions 4j + t, t in {2th, 2th+1}: the bytes of B-image dwords t of all four limbs
  float rstd = 1.f;
  const bool norm = a.norm_w != nullptr;
  if (norm) {                                     // block-wide mean of squares (fp32), like the VALU kernel's prologue
    float ssq = 0.f;
    for (int item = tid; item < K / 8; item += nthreads) {
      const unsigned* xwp = reinterpret_cast<const unsigned*>(a.x) + (item >> 1) * 8 + (item & 1);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float lo = act_lo<BF16>(xwp[2 * j]), hi = act_hi<BF16>(xwp[2 * j]);
        ssq += lo * lo + hi * hi;
      }
    }
    ssq = wave_sum(ssq);
    float* red = part;                            // partial-sum area is free until the tiles are consumed
    if (lane == 0) red[wave] = ssq;
    __syncthreads();
    float tot = 0.f;
    for (int i = 0; i < nthreads / 64; ++i) tot += red[i];
    rstd = rsqrtf(tot / static_cast<float>(K) + a.eps);
  }
  for (int item = tid; item < K / 8; item += nthreads) {
    const int k16 = item >> 1, th = item & 1;
    // values 4j + 2th (low halves of words 2j + th) and 4j + 2th + 1 (their high halves); the second read of x (after
    // the statistics pass) is served by the L1
    const unsigned* xwp = reinterpret_cast<const unsigned*>(a.x) + k16 * 8 + th;
    const unsigned* nwp = reinterpret_cast<const unsigned*>(norm ? a.norm_w : a.x) + k16 * 8 + th;
    float xe[4], xo[4];
    unsigned amax = 0;
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned wv = xwp[2 * j];
      xe[j] = act_lo<BF16>(wv);
      xo[j] = act_hi<BF16>(wv);
      if (norm) {                                 // one rounding to the activation dtype, like the reference's RMSNorm output
        const unsigned nv = nwp[2 * j];
        xe[j] = round_act<BF16>(xe[j] * rstd * act_lo<BF16>(nv));
        xo[j] = round_act<BF16>(xo[j] * rstd * act_hi<BF16>(nv));
      }
      amax = max(amax, max(__float_as_uint(xe[j]) & 0x7FFFFFFFu, __float_as_uint(xo[j]) & 0x7FFFFFFFu));
      sum += xe[j] + xo[j];
    }
    // group = 16 consecutive items (128 values): max exponent and sum over them
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
      amax = max(amax, static_cast<unsigned>(__shfl_xor(static_cast<int>(amax), d)));
      sum += __shfl_xor(sum, d);
    }
    // X = rint(x * 2^E) with |X| < 2^24:  E = 23 - floor(log2(max|x|)) = 150 - biased exponent of the maximum
    int E = 150 - static_cast<int>(amax >> 23);
    E = min(max(E, -100), 120);
    if ((item & 15) == 0) gtab[item >> 4] = make_float2(exp2i(-E - 2), sum);
    // even t (weight byte = q) carries limbs of 4X, odd t (weight byte = 4q) limbs of X.  Balanced base-128 digits
    // through a biased copy: Yb = Y + 64*(1 + 128 + 128^2); digit_i = ((Yb >> 7i) & 127) - 64, top = Yb >> 21.
    const float sce = exp2i(E + 2), sco = exp2i(E);
    unsigned le[4] = {0, 0, 0, 0}, lo[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int Ye = __float2int_rn(xe[j] * sce) + 0x102040, Yo = __float2int_rn(xo[j] * sco) + 0x102040;
#pragma unroll
      for (int l = 0; l < 4; ++l) {
        const int de = l < 3 ? ((Ye >> (7 * l)) & 127) - 64 : (Ye >> 21);
        const int dq = l < 3 ? ((Yo >> (7 * l)) & 127) - 64 : (Yo >> 21);
        le[l] |= static_cast<unsigned>(de & 0xFF) << (8 * j);
        lo[l] |= static_cast<unsigned>(dq & 0xFF) << (8 * j);
      }
    }
    const int st = k16 >> 4, kb = (k16 >> 2) & 3, m = k16 & 3;
    unsigned char* dst = bimg + (((st * 4 + m) * 4 + kb) * 4) * 16 + th * 8;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      u32x2 v; v.x = le[l]; v.y = lo[l];
      *reinterpret_cast<u32x2*>(dst + l * 16) = v;
    }
  }
  __syncthreads();

  // ---------------- this wave's B registers: lane (column c, K-block kb = lane/16) ---------------------------------
  const int kb = lane >> 4;
  i32x4 breg[GM_SPW][4];
  float cw[GM_SPW / 2], zx[GM_SPW / 2];
#pragma unroll
  for (int s = 0; s < GM_SPW; ++s) {
    const int st = kq * GM_SPW + s;
    const bool mine = (c >> 2) == ((s & 1) * 2 + (kb >> 1));        // this lane's K-block lies in its column's group
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(bimg + ((((st * 4 + m) * 4 + kb) * 4) + (c & 3)) * 16);
      breg[s][m] = mine ? __builtin_bit_cast(i32x4, v) : i32x4{0, 0, 0, 0};
    }
  }
#pragma unroll
  for (int f = 0; f < GM_SPW / 2; ++f) {
    const float2 t = gtab[(kq * GM_SPW + 2 * f) * 2 + (c >> 2)];
    const float lw = (c & 3) == 0 ? 1.f : (c & 3) == 1 ? 128.f : (c & 3) == 2 ? 16384.f : 2097152.f;
    cw[f] = t.x * lw;
    zx[f] = (c & 3) == 0 ? t.y : 0.f;
  }

  // ---------------- stream the tiles ----------------------------------------------------------------------------
  auto consume = [&](int set, int j) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int f = 0; f < GM_SPW / 2; ++f) {
      i32x4 d = {0, 0, 0, 0};
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int s = 2 * f + s2;
        const unsigned wv[4] = {w[set][s].x, w[set][s].y, w[set][s].z, w[set][s].w};
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const unsigned hi = wv[m] >> 4;
          i32x4 av;
          av.x = static_cast<int>(wv[m] & 0x03030303u);
          av.y = static_cast<int>(wv[m] & 0x0C0C0C0Cu);
          av.z = static_cast<int>(hi & 0x03030303u);
          av.w = static_cast<int>(hi & 0x0C0C0C0Cu);
          d = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, breg[s][m], d, 0, 0, 0);
        }
      }
      const unsigned szv[4] = {sz[set][f].x, sz[set][f].y, sz[set][f].z, sz[set][f].w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float df = static_cast<float>(d[i]) * cw[f];
        acc[i] = fmaf(df, sz_scale<BF16>(szv[i]), acc[i]);
        acc[i] = fmaf(zx[f], sz_zero<BF16>(szv[i]), acc[i]);
      }
    }
    // sum over the 16 columns (one DPP row); every lane of the row ends with the total
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v = acc[i];
      v += dpp_f32<0xB1>(v);
      v += dpp_f32<0x4E>(v);
      v += dpp_f32<0x141>(v);
      v += dpp_f32<0x140>(v);
      acc[i] = v;
    }
    if (c == 0) {
      float4 o; o.x = acc[0]; o.y = acc[1]; o.z = acc[2]; o.w = acc[3];
      *reinterpret_cast<float4*>(part + ((static_cast<size_t>(rg) * tiles_per_rg + j) * KS + kq) * 16 + 4 * rq) = o;
    }
  };
  int j = 0;
  for (; j + 1 < tiles_per_rg; j += 2) {          // pairs: no conditional loads (surplus requests are clamped re-reads)
    issue(1, j + 1);
    consume(0, j);
    issue(0, j + 2);
    consume(1, j + 1);
  }
  if (j < tiles_per_rg) consume(0, j);            // odd tail: set 0 holds it already
  __syncthreads();

  // ---------------- cross-slice sums, epilogue (residual / SwiGLU), store -----------------------------------------
  const int rows_out = a.swiglu ? 8 : 16;         // output rows per tile
  for (int idx = tid; idx < RG * tiles_per_rg * rows_out; idx += nthreads) {
    const int rr = idx % rows_out, tj = idx / rows_out;   // tj = rg' * tiles_per_rg + j
    const int rgl = tj / tiles_per_rg, jj = tj % tiles_per_rg;
    const int tile = (blockIdx.x * RG + rgl) + jj * nrg;
    const int n = tile * rows_out + rr;
    if (tile < a.ntiles && n < N) {
      float v = 0.f, u = 0.f;
      for (int q = 0; q < KS; ++q) {
        v += part[(static_cast<size_t>(tj) * KS + q) * 16 + rr];
        if (a.swiglu) u += part[(static_cast<size_t>(tj) * KS + q) * 16 + rr + 8];
      }
      if (a.swiglu) {         // act gate, act up, act silu(gate), act product: the reference's op-by-op roundings
        const float g = round_act<BF16>(v), up = round_act<BF16>(u);
        const float sl = round_act<BF16>(g / (1.f + __expf(-g)));
        a.y[n] = f32_to_act<BF16>(sl * up);
      } else {
        if (a.residual) v = act_to_f32<BF16>(a.residual[n]) + round_act<BF16>(v);
        a.y[n] = f32_to_act<BF16>(v);
      }
    }
  }
}

// ---- load-time re-tiling ---------------------------------------------------------------------------------------
// wt[tile][step][lane][d] = row(tile, lane%16)[step*16 + (lane/16)*4 + d]   (2-bit: 16 dwords per row and step), where
// row(tile, r) = row tile*16 + r of the matrix, or -- SwiGLU pair -- row tile*8 + r of gate (r < 8) / tile*8 + r - 8 of up.
__global__ __launch_bounds__(256) void prepack_tiles_kernel(const uint32_t* __restrict__ qw, const uint32_t* __restrict__ qw2,
                                                            uint32_t* __restrict__ wt, int N, int W, int nsteps, size_t total) {
  for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < total;
       i += static_cast<size_t>(gridDim.x) * blockDim.x) {
    const int d = static_cast<int>(i & 3), lane = static_cast<int>((i >> 2) & 63);
    const size_t blk = i >> 8;
    const int step = static_cast<int>(blk % nsteps);
    const size_t tile = blk / nsteps;
    const int r = lane & 15;
    const uint32_t* src = qw2 ? (r < 8 ? qw : qw2) : qw;
    const size_t row = qw2 ? tile * 8 + (r & 7) : tile * 16 + r;
    wt[i] = row < static_cast<size_t>(N) ? src[row * W + step * 16 + (lane >> 4) * 4 + d] : 0u;
  }
}
// szt[tile][g][r] = scale bits | zero bits << 16 of row(tile, r), group g
__global__ __launch_bounds__(256) void prepack_sz_tiles_kernel(const unsigned short* __restrict__ sc, const unsigned short* __restrict__ ze,
                                                               const unsigned short* __restrict__ sc2, const unsigned short* __restrict__ ze2,
                                                               uint32_t* __restrict__ szt, int N, int G, size_t total) {
  for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < total;
       i += static_cast<size_t>(gridDim.x) * blockDim.x) {
    const int r = static_cast<int>(i & 15);
    const size_t tg = i >> 4;
    const int g = static_cast<int>(tg % G);
    const size_t tile = tg / G;
    const unsigned short* s = sc2 ? (r < 8 ? sc : sc2) : sc;
    const unsigned short* z = sc2 ? (r < 8 ? ze : ze2) : ze;
    const size_t row = sc2 ? tile * 8 + (r & 7) : tile * 16 + r;
    szt[i] = row < static_cast<size_t>(N) ? (static_cast<uint32_t>(s[row * G + g]) | (static_cast<uint32_t>(z[row * G + g]) << 16)) : 0u;
  }
}

hipError_t launch_prepack_tiles(const uint32_t* qw, const void* sc, const void* ze, const uint32_t* qw2, const void* sc2,
                                const void* ze2, uint32_t* wt, uint32_t* szt, int N, int K, hipStream_t s) {
  const int ntiles = qw2 ? (N + 7) / 8 : (N + 15) / 16, nsteps = K / 256, G = K / 128, W = K / 16;
  const size_t tw = static_cast<size_t>(ntiles) * nsteps * 256, ts = static_cast<size_t>(ntiles) * G * 16;
  hipLaunchKernelGGL(prepack_tiles_kernel, dim3(static_cast<unsigned>(std::min<size_t>((tw + 255) / 256, 8192))), dim3(256), 0, s,
                     qw, qw2, wt, N, W, nsteps, tw);
  hipLaunchKernelGGL(prepack_sz_tiles_kernel, dim3(static_cast<unsigned>(std::min<size_t>((ts + 255) / 256, 4096))), dim3(256), 0, s,
                     static_cast<const unsigned short*>(sc), static_cast<const unsigned short*>(ze),
                     static_cast<const unsigned short*>(sc2), static_cast<const unsigned short*>(ze2), szt, N, G, ts);
  return hipGetLastError();
}

template <int SPW>
static hipError_t launch_gemv_mfma_spw(const GemvMfmaArgs& a0, bool bf16, hipStream_t s) {
  GemvMfmaArgs a = a0;
  const int nsteps = a.K / 256, G = a.K / 128;
  const int KS = nsteps / SPW;
  // row groups per workgroup: up to ~8 waves, but never fewer workgroups than CUs while tiles last
  int RG = std::max(1, 8 / KS);
  while (RG > 1 && (a.ntiles + RG - 1) / RG < 256) --RG;
  const int blocks = std::min((a.ntiles + RG - 1) / RG, 256);
  const int nrg = blocks * RG;
  const int tprg = (a.ntiles + nrg - 1) / nrg;
  a.rg_per_block = RG;
  a.ks = KS;
  const size_t shmem = static_cast<size_t>(nsteps) * 1024 + static_cast<size_t>(G) * 8 + static_cast<size_t>(RG) * tprg * KS * 16 * 4 + 64;
  static DeviceOnce attr[2];
  if (attr[bf16].pending()) {
    hipError_t e = bf16 ? hipFuncSetAttribute(reinterpret_cast<const void*>(&gemv_mfma_kernel<true, SPW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
                        : hipFuncSetAttribute(reinterpret_cast<const void*>(&gemv_mfma_kernel<false, SPW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr[bf16].done();
  }
  dim3 grid(blocks), block(KS * RG * 64);
  if (bf16) hipLaunchKernelGGL((gemv_mfma_kernel<true, SPW>), grid, block, shmem, s, a);
  else hipLaunchKernelGGL((gemv_mfma_kernel<false, SPW>), grid, block, shmem, s, a);
  return hipGetLastError();
}

hipError_t launch_gemv_mfma(const void* x, const uint32_t* wt, const uint32_t* szt, void* y, int N, int K, const void* norm_w,
                            float eps, const void* residual, int swiglu, bool bf16, hipStream_t s) {
  const int nsteps = K / 256;
  if (K % 1024 != 0 || nsteps > 64) return hipErrorInvalidValue;
  const int ntiles = swiglu ? (N + 7) / 8 : (N + 15) / 16;
  GemvMfmaArgs a{static_cast<const unsigned short*>(x), wt, szt, static_cast<unsigned short*>(y), N, K, ntiles, 1, 1,
                 static_cast<const unsigned short*>(norm_w), eps, static_cast<const unsigned short*>(residual), swiglu};
  // serial work of the busiest row group ~ tiles per row group x steps per wave, for 8-wave workgroups on 256 CUs
  auto cost = [&](int spw) {
    const int KS = nsteps / spw;
    int RG = std::max(1, 8 / KS);
    while (RG > 1 && (ntiles + RG - 1) / RG < 256) --RG;
    const int nrg = std::min((ntiles + RG - 1) / RG, 256) * RG;
    return ((ntiles + nrg - 1) / nrg) * spw;
  };
  if (nsteps <= 32 && nsteps % 2 == 0 && cost(2) < cost(4)) return launch_gemv_mfma_spw<2>(a, bf16, s);
  if (nsteps % 4 != 0) return hipErrorInvalidValue;
  return launch_gemv_mfma_spw<4>(a, bf16, s);
}

}  // namespace gbx
