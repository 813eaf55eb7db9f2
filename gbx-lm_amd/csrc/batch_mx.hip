// EXPERIMENTAL (libgbx_experimental.so): the batched-decode GEMM (M <= 32 rows of activations, 2 bits, group 128, bf16) with its dot
// products on the FP4 x FP8 matrix cores -- the operand the round-3 and round-4 reviews asked to see measured at M = 32.
// Replaces mx.quantized_matmul for x of shape [B, 1, K] (gbx_lm/models/quantized_linear_gba.py:195-203), like mfma_batch.hip.
//
// Arithmetic (dgemv_mx.hip's, here with 32 REAL activation rows instead of 16 groups of one row): a packed 2-bit field masked in place
// in its nibble is the FP4 (E2M1) number q / 2; the activation tile is two FP8 (E4M3) limbs with one E8M0 exponent per 32 values (the
// block maximum scaled into [128, 256): exact for elements >= 2^-9 of the block maximum, absolute error <= 2^-17 of it below); per
// (16 weight rows, 128-value group) FOUR v_mfma_scale_f32_16x16x128_f8f6f4 (2 row tiles x 2 limbs) give t[m, n] = sum_k x[m, k] q[n, k] / 2
// and the affine map is applied per group:  acc[m, n] += 2 s[n, g] t[m, n] + z[n, g] xsum[m, g].
// The activation image is a function of x alone: written ONCE (bmx_image_kernel; in a fused design by the producer of x), not by
// every workgroup.  Image of group g, row tile mt, limb l: two 1 KB pieces [piece][64 lanes][16 B]; A lane (m, ga) byte j holds
//   k = 32 ((ga >> 1) + 2 (j >> 4)) + 16 (ga & 1) + 2 (j & 7) + ((j & 15) >> 3)          (dgemv_mx.hip's operand slot map with the B
// lane (n, gb) masking its two packed words in place: dwords w0 & M, (w0 >> 2) & M, w1 & M, (w1 >> 2) & M).
#include "gbx_device.h"
#include "kernels.h"
#include <hip/hip_runtime.h>
#include <cstdlib>

namespace gbx {

typedef int bmx_v8i __attribute__((ext_vector_type(8)));
typedef float bmx_v4f __attribute__((ext_vector_type(4)));

constexpr int BMX_GROUP_IMG = 8192;          // bytes of one group's image: [mt 2][limb 2][piece 2][64][16]
constexpr int BMX_GROUP_AUX = 256;           // + 128 scale bytes [mt][64] + 32 group sums (fp32)

// one thread per (row m < 32, group g): 128 values -> 4 blocks of 32
__global__ __launch_bounds__(256) void bmx_image_kernel(const unsigned short* __restrict__ x, int M, int K, unsigned char* __restrict__ img,
                                                        unsigned char* __restrict__ aux) {
  const int G = K >> 7;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= 32 * G) return;
  const int m = idx & 31, g = idx >> 5;
  const int mt = m >> 4, mm = m & 15;
  unsigned char* gi = img + static_cast<size_t>(g) * BMX_GROUP_IMG;
  unsigned char* ga_ = aux + static_cast<size_t>(g) * BMX_GROUP_AUX;
  float gsum = 0.f;
  for (int kb = 0; kb < 4; ++kb) {
    float v[32];
    unsigned amax = 0;
    for (int r = 0; r < 32; ++r) {
      const float f = m < M ? act_to_f32<true>(x[static_cast<size_t>(m) * K + g * 128 + kb * 32 + r]) : 0.f;
      v[r] = f;
      gsum += f;
      amax = max(amax, __float_as_uint(f) & 0x7FFFFFFFu);
    }
    int ea = static_cast<int>(amax >> 23);
    ea = min(max(ea, 20), 240);
    const float up = exp2i(134 - ea);
    for (int r = 0; r < 32; r += 2) {
      const float r0 = v[r] * up, r1 = v[r + 1] * up;
      const int ph = __builtin_amdgcn_cvt_pk_fp8_f32(r0, r1, 0, false);
      const int pl = __builtin_amdgcn_cvt_pk_fp8_f32(r0 - __builtin_amdgcn_cvt_f32_fp8(ph, 0), r1 - __builtin_amdgcn_cvt_f32_fp8(ph, 1), 0, false);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int rr = r + e, r16 = rr & 15;
        const int ga = 2 * (kb & 1) + (rr >> 4);
        const int j = 16 * (kb >> 1) + 8 * (r16 & 1) + (r16 >> 1);
        const int lane = mm + 16 * ga;
        const size_t off = static_cast<size_t>(mt * 2) * 2048 + (j >> 4) * 1024 + lane * 16 + (j & 15);
        gi[off] = static_cast<unsigned char>((ph >> (8 * e)) & 0xFF);                 // limb 0
        gi[off + 2048] = static_cast<unsigned char>((pl >> (8 * e)) & 0xFF);          // limb 1
      }
    }
    ga_[mt * 64 + mm + 16 * kb] = static_cast<unsigned char>(ea - 7);              // A lane (m, gb = kb)
  }
  reinterpret_cast<float*>(ga_ + 128)[m] = gsum;
}

// Workgroup = 8 waves = 4 row pairs (32 weight rows each: two 16-row B tiles sharing every A fragment) x 2 k halves (even / odd
// groups of a pair); 128 weight rows per workgroup, groups [ks * ng, (ks + 1) * ng) of the k range (blockIdx.y = ks: narrow matrices
// split k across workgroups and leave fp32 slices, like gbx_qlinear_forward_partials).
// EVERYTHING a step needs -- the image of a group pair (16 KB + 512 B of scales / sums) and the pair's packed words of the 128 rows
// (8 KB) -- arrives by LDS-DMA into a ring of BMX_NS stages, three stages in flight: no vector registers on the load path, waits are
// counted by hand (3 DMA instructions per wave and stage, 4 for wave 0), one raw s_barrier per step.  The packed words are stored with
// an XOR swizzle applied to the SOURCE address (slot s of row r holds piece s ^ ((r >> 1) & 3)) so that the B-operand reads are
// conflict-free although the DMA's LDS image is lane-linear.  Scales | zeros of the workgroup's rows and groups sit in LDS once.
constexpr int BMX_NS = 4;
constexpr int BMX_ST_AUX = 16384, BMX_ST_W = 16896, BMX_STAGE = 25088;      // image 2 x 8192 | aux 2 x 256 | words 128 x 64
constexpr int BMX_SZ_OFF = BMX_NS * BMX_STAGE;                               // [128 rows][ng <= 32] u32 (scale | zero << 16)
constexpr int BMX_LDS = BMX_SZ_OFF + 128 * 32 * 8;     // [128 rows][ng <= 32] {2 s, z} as fp32

#define BMX_LDSP(p) ((__attribute__((address_space(3))) void*)(p))

struct BmxArgs {
  const unsigned char* img;     // [G][8192]
  const unsigned char* aux;     // [G][256]
  const uint32_t* qw;           // [N][K / 16]
  const unsigned short* sc;     // [N][G]
  const unsigned short* ze;
  unsigned short* y;            // [M][N] (one k range)
  float* ws;                    // [ksplit][M][N] (k split across workgroups)
  int N, K, M, ng;              // ng: groups per workgroup (even, <= 32)
};

__global__ __launch_bounds__(512) void qgemm_batch_mx_kernel(const BmxArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int RS_FLAGS = 0x00020000;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int np = wave & 3, kh = wave >> 2;                      // row pair, k half
  const int nn = lane & 15, gb = lane >> 4;
  const int K = a.K, G = K >> 7, W4 = K >> 2;                   // bytes per packed row
  const int n0 = blockIdx.x * 128;
  const int ng = a.ng, gbeg = blockIdx.y * ng, npairs = ng >> 1;
  const __amdgpu_buffer_rsrc_t idesc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.img), 0, G * BMX_GROUP_IMG, RS_FLAGS);
  const __amdgpu_buffer_rsrc_t adesc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.aux), 0, G * BMX_GROUP_AUX, RS_FLAGS);
  const __amdgpu_buffer_rsrc_t wdesc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(a.qw), 0, 0x7FFFFFFF, RS_FLAGS);
  const int wr = wave * 16 + (lane >> 2);                        // row of the workgroup this lane fetches packed words for
  const unsigned woff = static_cast<unsigned>((n0 + wr) * W4 + (((lane & 3) ^ ((wr >> 1) & 3)) << 4));
  auto issue = [&](int p) {                                      // stage p -> slot p % BMX_NS
    unsigned char* st = smem + (p % BMX_NS) * BMX_STAGE;
    const int g2 = gbeg + 2 * p;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(idesc, BMX_LDSP(st + wave * 1024), 16, static_cast<unsigned>(wave * 1024 + lane * 16), g2 * BMX_GROUP_IMG, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(idesc, BMX_LDSP(st + 8192 + wave * 1024), 16, static_cast<unsigned>(8192 + wave * 1024 + lane * 16), g2 * BMX_GROUP_IMG, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wdesc, BMX_LDSP(st + BMX_ST_W + wave * 1024), 16, woff, g2 * 32, 0, 0);
    if (wave == 0 && lane < 32)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(adesc, BMX_LDSP(st + BMX_ST_AUX), 16, static_cast<unsigned>(lane * 16), g2 * BMX_GROUP_AUX, 0, 0);
  };
  // wait until this wave's DMA of the OLDEST stage in flight has landed, `ahead` younger stages may stay in flight
  auto wait_stage = [&](int ahead) {
    if (wave == 0) {
      if (ahead >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      if (ahead >= 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if (ahead == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  };

  // scales | zeros of the 128 rows x ng groups -> LDS (ordinary loads, landed before the first DMA is requested: no mixed queue)
  float2* szl = reinterpret_cast<float2*>(smem + BMX_SZ_OFF);
  {                                                              // one 16-byte piece (8 groups) of scales and of zeros per thread -> {2 s, z} fp32
    const int cpr = ng >> 3;                                     // pieces per row (ng % 8 == 0)
    if (tid < 128 * cpr) {
      const int row = tid / cpr, c = tid - row * cpr;
      const size_t o = static_cast<size_t>(n0 + row) * G + gbeg + c * 8;
      const u32x4 sv = *reinterpret_cast<const u32x4*>(a.sc + o), zv = *reinterpret_cast<const u32x4*>(a.ze + o);
      float2* d = szl + row * ng + c * 8;
      const unsigned sw[4] = {sv.x, sv.y, sv.z, sv.w}, zw[4] = {zv.x, zv.y, zv.z, zv.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        d[2 * q] = make_float2(2.f * act_lo<true>(sw[q]), act_lo<true>(zw[q]));
        d[2 * q + 1] = make_float2(2.f * act_hi<true>(sw[q]), act_hi<true>(zw[q]));
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  issue(0);
  if (npairs > 1) issue(1);
  if (npairs > 2) issue(2);

  bmx_v4f acc[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) acc[t][mt] = bmx_v4f{0.f, 0.f, 0.f, 0.f};

  // The fragments of step p + 1 are read from LDS BEFORE the matrix instructions of step p are issued (two register sets): with two
  // waves per SIMD nothing else hides the LDS latency and the per-group vector work behind the 8 x 64 cycles of a step's MFMAs.
  struct Frag {
    bmx_v8i af[2][2];
    int sa[2];
    float xs[2][4];
    u32x2 wv[2];
    float2 sz[2];
  };
  auto load_frags = [&](int p, Frag& f) {
    const unsigned char* st = smem + (p % BMX_NS) * BMX_STAGE;
    const unsigned char* gi = st + kh * BMX_GROUP_IMG;
    const unsigned char* ga = st + BMX_ST_AUX + kh * BMX_GROUP_AUX;
    const int gl = 2 * p + kh;                                   // this wave's group of the pair, local index
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
      for (int l = 0; l < 2; ++l) {
        const u32x4 p0 = *reinterpret_cast<const u32x4*>(gi + mt * 4096 + l * 2048 + lane * 16);
        const u32x4 p1 = *reinterpret_cast<const u32x4*>(gi + mt * 4096 + l * 2048 + 1024 + lane * 16);
        f.af[mt][l][0] = p0.x; f.af[mt][l][1] = p0.y; f.af[mt][l][2] = p0.z; f.af[mt][l][3] = p0.w;
        f.af[mt][l][4] = p1.x; f.af[mt][l][5] = p1.y; f.af[mt][l][6] = p1.z; f.af[mt][l][7] = p1.w;
      }
      f.sa[mt] = static_cast<int>(ga[mt * 64 + lane]);
      const u32x4 xv = *reinterpret_cast<const u32x4*>(ga + 128 + (mt * 16 + 4 * gb) * 4);
      f.xs[mt][0] = __uint_as_float(xv.x); f.xs[mt][1] = __uint_as_float(xv.y); f.xs[mt][2] = __uint_as_float(xv.z); f.xs[mt][3] = __uint_as_float(xv.w);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int r = np * 32 + 16 * t + nn;                       // row of the workgroup
      f.wv[t] = *reinterpret_cast<const u32x2*>(st + BMX_ST_W + r * 64 + (((kh * 2 + (gb >> 1)) ^ ((r >> 1) & 3)) << 4) + (gb & 1) * 8);
      f.sz[t] = szl[r * ng + gl];
    }
  };
  auto compute = [&](const Frag& f) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      bmx_v8i bv;
      bv[0] = f.wv[t].x & 0x33333333u; bv[1] = (f.wv[t].x >> 2) & 0x33333333u; bv[2] = f.wv[t].y & 0x33333333u; bv[3] = (f.wv[t].y >> 2) & 0x33333333u;
      bv[4] = 0; bv[5] = 0; bv[6] = 0; bv[7] = 0;
      const float s2 = f.sz[t].x, zz = f.sz[t].y;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        bmx_v4f c = {0.f, 0.f, 0.f, 0.f};
        c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(f.af[mt][0], bv, c, 0 /*A: fp8 e4m3*/, 4 /*B: fp4*/, 0, f.sa[mt], 0, 127);
        c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(f.af[mt][1], bv, c, 0, 4, 0, f.sa[mt], 0, 127);
        // D[row = activation row 4 gb + i][col = weight row nn]: the column's scale / zero, the row's group sum
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[t][mt][i] = fmaf(zz, f.xs[mt][i], fmaf(s2, c[i], acc[t][mt][i]));
      }
    }
  };
  // top of step p: stage p + 1 has landed everywhere and every wave holds stage p in registers -> slot p % NS is free for stage p + NS
  auto advance = [&](int p) {
    if (p + 1 < npairs) wait_stage(min(npairs - 2 - p, 2));
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (p + BMX_NS < npairs) issue(p + BMX_NS);
  };
  Frag f0, f1;
  wait_stage(min(npairs - 1, 2));
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if (3 < npairs) issue(3);
  load_frags(0, f0);
  for (int p = 0; p < npairs; p += 2) {
    advance(p);
    if (p + 1 < npairs) load_frags(p + 1, f1);
    compute(f0);
    if (p + 1 < npairs) {
      advance(p + 1);
      if (p + 2 < npairs) load_frags(p + 2, f0);
      compute(f1);
    }
  }
  // the two k halves meet in LDS: half 1 writes, half 0 adds and stores
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  float* red = reinterpret_cast<float*>(smem);                  // [4 row pairs][16 values][64 lanes]
  if (kh == 1) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i) red[(np * 16 + (t * 2 + mt) * 4 + i) * 64 + lane] = acc[t][mt][i];
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if (kh == 0) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float v = acc[t][mt][i] + red[(np * 16 + (t * 2 + mt) * 4 + i) * 64 + lane];
          const int m = mt * 16 + 4 * gb + i;
          const size_t o = static_cast<size_t>(m) * a.N + n0 + np * 32 + 16 * t + nn;
          if (m < a.M) {
            if (a.ws) a.ws[static_cast<size_t>(blockIdx.y) * a.M * a.N + o] = v;
            else a.y[o] = f32_to_act<true>(v);
          }
        }
  }
}

size_t batch_mx_image_bytes(int K) { return static_cast<size_t>(K >> 7) * BMX_GROUP_IMG; }
size_t batch_mx_aux_bytes(int K) { return static_cast<size_t>(K >> 7) * BMX_GROUP_AUX; }

hipError_t launch_batch_mx_image(const void* x, int M, int K, void* img, void* aux, hipStream_t s) {
  const int n = 32 * (K >> 7);
  hipLaunchKernelGGL(bmx_image_kernel, dim3((n + 255) / 256), dim3(256), 0, s, static_cast<const unsigned short*>(x), M, K,
                     static_cast<unsigned char*>(img), static_cast<unsigned char*>(aux));
  return hipGetLastError();
}

hipError_t launch_batch_mx(const void* img, const void* aux, const uint32_t* qw, const void* sc, const void* ze, void* y, float* ws, int ksplit,
                           int M, int N, int K, hipStream_t s) {
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(qgemm_batch_mx_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, BMX_LDS);
    if (e != hipSuccess) return e;
    attr = true;
  }
  BmxArgs a;
  a.img = static_cast<const unsigned char*>(img); a.aux = static_cast<const unsigned char*>(aux);
  a.qw = qw; a.sc = static_cast<const unsigned short*>(sc); a.ze = static_cast<const unsigned short*>(ze);
  a.y = static_cast<unsigned short*>(y); a.ws = ksplit > 1 ? ws : nullptr; a.N = N; a.K = K; a.M = M; a.ng = (K >> 7) / ksplit;
  hipLaunchKernelGGL(qgemm_batch_mx_kernel, dim3(N / 128, ksplit), dim3(512), BMX_LDS, s, a);
  return hipGetLastError();
}

// k ranges of a launch: 1 when the rows alone fill the chip, else the largest split that keeps an even number (<= 32) of groups per workgroup
int batch_mx_ksplit(int N, int K) {
  const int G = K >> 7;
  // candidates: groups per workgroup a multiple of 8, at most 32; the smallest split that gives the chip >= 200 workgroups (or the largest allowed)
  int best = 0;
  for (int c = 1; c <= 16; ++c) {
    if (G % c) continue;
    const int ng = G / c;
    if (ng > 32 || (ng % 8)) continue;
    best = c;
    if ((N / 128) * c >= 200) break;
  }
  return best;
}

}  // namespace gbx
