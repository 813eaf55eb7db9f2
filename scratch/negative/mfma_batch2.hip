// Batched decode GEMM (2 <= M <= 32 rows), second pipeline (round 4): the arithmetic of mfma_batch.hip -- integer fields under an
// exponent image straight into the matrix cores, the affine map applied once per group -- with every byte moved by the LOADER waves.
//
// What the first pipeline was bound by (in-kernel stamps, profiles/r04_batch_stamps.txt; knock-out builds, profiles/r04_batch_knockouts.txt):
// its compute waves fetched their own packed words, 16 bytes per lane from 32 different rows per instruction.  ISSUING the six loads
// of a step took a compute wave 6-9 thousand cycles (the CU's vector-memory path takes such an instruction at about two cycles per
// lane and is shared with the loaders' 64 KB of activations), its MFMA work 3.2-4.1 thousand: the loaders sat at the barrier for
// 60-80 % of a step and the launch time was the SUM of activation staging, weight stream and arithmetic instead of their maximum.
// Here:
//   * the loader waves fetch the packed words too -- eight lanes per 128-byte line, so a wave instruction covers 8 lines instead of
//     32 -- and the (scale, zero) pairs, and park them in LDS beside the activation slab; the compute waves read everything from LDS
//     and issue no vector-memory instruction between the prologue and the epilogue;
//   * a step is 4 chunks (512 k-values) instead of 8, so two register sets per loader thread fit (x: 2 x 32, words: 2 x 16-32
//     VGPRs): the loads of step s+2 are ISSUED before the registers of step s+1 are turned into the slab, one full step of lead
//     instead of the tail of one;
//   * step s covers the contiguous k range [512 s, 512 s + 512); k-range kg of the workgroup takes chunks [kg NCHW, (kg+1) NCHW)
//     of every step (any partition of k is as good as another for the partial sums), so the loaders' requests are whole lines.
// LDS per workgroup: activation slabs 2 x 34 KB, packed words 2 x ROWS x (step bytes + 16) (37 KB at 2 bits / 128 rows, 70 KB
// at 4 bits), (scale, zero) 2 x 2 KB, xsum 2 x 512 B.
// Group 128 only (the smaller groups of 2-bit checkpoints stay on mfma_batch.hip); k per workgroup a multiple of 512.
#include "mfma_common.h"
#include "gbx_env.h"
#include "kernels.h"

namespace gbx {

template <bool BF16> __device__ __forceinline__ float xdot2b(unsigned v, float acc) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  if constexpr (BF16) return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, v), __builtin_bit_cast(bf2, 0x3F803F80u), acc, false);
  else return __builtin_amdgcn_fdot2(as_h2(v), as_h2(0x3C003C00u), acc, false);
}

#ifdef BX_STAMPS
__device__ long long* bx2_stamp_ptr;                // diagnostic build only (scratch/stamps_batch.py)
#endif

template <int BITS, int TNW, int NG, int KG, bool BF16>
__global__ __launch_bounds__(768) void qgemm_batch2_kernel(const MfmaArgs a) {
  static_assert(NG * KG == 8 && KG <= 4 && BITS >= 2 && BITS <= 4 && (TNW == 16 || TNW == 32), "unsupported shape");
  constexpr int SCH = 4;                            // chunks per step
  constexpr int NCHW = SCH / KG;                    // chunks per step of one k-range
  constexpr int H = 64 / TNW;                       // lane groups of a wave (k split inside a chunk)
  constexpr int UPL = 16 / H;                       // MFMA steps per chunk
  constexpr int WL = UPL * BITS / 4;                // packed words per lane and chunk
  constexpr int WPC = KC * BITS / 32;               // packed words per row and chunk
  constexpr int MT = 32 / TNW;                      // activation row tiles (32 rows)
  constexpr int AE = TNW == 32 ? 16 : 4;            // accumulator elements per lane and tile
  constexpr int ROWS = NG * TNW;                    // weight rows per workgroup
  constexpr int CHUNK = 32 * XROW;                  // slab bytes of one chunk (32 activation rows)
  constexpr int XBUF = SCH * CHUNK;
  constexpr int SB = SCH * WPC * 4;                 // packed bytes per row and step: 128 / 192 / 256
#ifdef BX2_WROW
  constexpr int WROW = BX2_WROW;
#else
  constexpr int WROW = SB + 16;
#endif                                              // LDS row stride of the packed words (conflict-free 8 / 16-byte reads)
  constexpr int WBUF = ROWS * WROW;
  constexpr int SZBUF = ROWS * 16;                  // per row: 4 scales | 4 zeros of the step
  constexpr int XSB = SCH * 32 * 4;                 // xsum [chunk][32 rows]
  constexpr int PPR = SB / 16;                      // 16-byte pieces per row and step
  constexpr int WPT = (ROWS * PPR + 255) / 256;     // pieces per loader thread and step
  constexpr unsigned MAGIC = BF16 ? 0x43004300u : 0x64006400u;   // 128.0 | 1024.0 twice
  constexpr float CF = BF16 ? 128.f : 1024.f;
  constexpr unsigned FMASK = BITS == 2 ? 0x00030003u : BITS == 3 ? 0x00070007u : 0x000F000Fu;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const xb = smem;                   // [2][XBUF]
  unsigned char* const wb = smem + 2 * XBUF;        // [2][WBUF]
  unsigned char* const zb = wb + 2 * WBUF;          // [2][SZBUF]
  float* const xsum = reinterpret_cast<float*>(zb + 2 * SZBUF);   // [2][SCH][32]
  float* const red = reinterpret_cast<float*>(smem);              // after the main loop: [8 waves][MT*AE][64]

  // (f & FMASK) | MAGIC as one VOP3 instruction (mfma_batch.hip)
  unsigned fmask_s = FMASK, magic_v = MAGIC;
  asm volatile("" : "+s"(fmask_s));
  asm volatile("" : "+v"(magic_v));
  auto fld = [&](unsigned f) {
    unsigned d;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(d) : "v"(f), "s"(fmask_s), "v"(magic_v));
    return d;
  };

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ng = wave % NG, kg = wave / NG;
  const int r = lane & (TNW - 1), h = lane / TNW;
  const int K = a.K, M = a.M;
  const int W = K / 32 * BITS, G = K / KC;
  const int ksplit = a.ws ? a.ksplit : 1;
  const int nchunks = K / KC / ksplit, cbase = blockIdx.y * nchunks;
  const int steps = nchunks / SCH;

  // segment of this workgroup's rows (uniform): a workgroup never straddles two matrices, except the SwiGLU pair by design
  const int nwg = blockIdx.x * ROWS;
  const uint32_t* qw = a.qweight;
  const unsigned short* sc = a.scales;
  const unsigned short* ze = a.zeros;
  const unsigned short* bs = a.bias;
  unsigned short* yo = a.y;
  int nl0 = nwg, NL = a.N;
  const bool swiglu = a.swiglu != 0;               // (uniform) rows [0, ROWS/2) = gate rows, [ROWS/2, ROWS) = the same rows of up
  if (swiglu) {
    nl0 = blockIdx.x * (ROWS / 2);
    NL = a.n_seg[0];
  } else if (a.nseg > 1) {
    const int sg = (nwg >= a.n_seg[0]) + (nwg >= a.n_seg[0] + a.n_seg[1]);
    NL = a.n_seg[sg];
    if (sg > 0) {
      nl0 = nwg - a.n_seg[0] - (sg > 1 ? a.n_seg[1] : 0);
      qw = a.qweight_x[sg - 1]; sc = a.scales_x[sg - 1]; ze = a.zeros_x[sg - 1]; bs = a.bias_x[sg - 1]; yo = a.y_x[sg - 1];
    }
  }
  // weight row `row` of the workgroup (0 .. ROWS-1) -> matrix and row inside it
  auto row_src = [&](int row, const uint32_t*& q, const unsigned short*& s_, const unsigned short*& z_) {
    int rl = row;
    q = qw; s_ = sc; z_ = ze;
    if (swiglu && row >= ROWS / 2) { q = a.qweight_x[0]; s_ = a.scales_x[0]; z_ = a.zeros_x[0]; rl = row - ROWS / 2; }
    return min(nl0 + rl, NL - 1);
  };

  const bool loader = wave >= 8;
  const int lt = tid & 255;
  const bool half_rows = M <= 16;                   // M <= 16: only the first 16 activation rows are staged (loader waves 10, 11 stage none)

  // Every workgroup walks its k range from a different starting step (HBM channel camping, mfma_batch.hip)
  const int rot = blockIdx.x % steps;
  auto kstep = [&](int i) { const int t = i + rot; return t >= steps ? t - steps : t; };
#ifdef BX_STAMPS
  long long st[24];
  int sti = 0;
#define STAMP2() { if (sti < 24) st[sti++] = __builtin_amdgcn_s_memtime(); }
#else
#define STAMP2()
#endif
  STAMP2();

  f32x16 acc32 = f32x16{0};
  f32x4v acc16[TNW == 16 ? MT : 1];
#pragma unroll
  for (int i = 0; i < (TNW == 16 ? MT : 1); ++i) acc16[i] = f32x4v{0};

#ifdef BX2_SPLIT
  // Experiment: the two streams on DIFFERENT waves (a wave's loads retire in order: words from HBM ahead of activations from L2 in one
  // wave's queue make the activations wait for HBM).  Waves 8, 9: activations only (two row pieces per thread); waves 10, 11: packed
  // words + (scale, zero) only.  ROWS = 128, 2 bits (WPT = 4) as built by -DBX2_ONE.
  if (loader && lt < 128) {
    const int lrow = lt >> 3, lpc = lt & 7;                       // rows lrow and lrow + 16
    const unsigned short* xrow0 = a.x + static_cast<size_t>(min(lrow, M - 1)) * K + lpc * 16 + static_cast<size_t>(cbase) * KC;
    const unsigned short* xrow1 = a.x + static_cast<size_t>(min(lrow + 16, M - 1)) * K + lpc * 16 + static_cast<size_t>(cbase) * KC;
    const bool act1 = !half_rows;
    u32x4 xs[2][SCH][2][2];
    auto issue_x = [&](int set, int ks) {
#pragma unroll
      for (int it = 0; it < SCH; ++it) {
        const unsigned short* p0 = xrow0 + (ks * SCH + it) * KC;
        const unsigned short* p1 = xrow1 + (ks * SCH + it) * KC;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(xs[set][it][0][0]) : "v"(p0) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(xs[set][it][0][1]) : "v"(p0) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(xs[set][it][1][0]) : "v"(p1) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(xs[set][it][1][1]) : "v"(p1) : "memory");
      }
    };
    auto landed_x = [&](int set) {
      asm volatile("s_waitcnt vmcnt(16)"
                   : "+v"(xs[set][0][0][0]), "+v"(xs[set][0][0][1]), "+v"(xs[set][0][1][0]), "+v"(xs[set][0][1][1]),
                     "+v"(xs[set][1][0][0]), "+v"(xs[set][1][0][1]), "+v"(xs[set][1][1][0]), "+v"(xs[set][1][1][1]),
                     "+v"(xs[set][2][0][0]), "+v"(xs[set][2][0][1]), "+v"(xs[set][2][1][0]), "+v"(xs[set][2][1][1]),
                     "+v"(xs[set][3][0][0]), "+v"(xs[set][3][0][1]), "+v"(xs[set][3][1][0]), "+v"(xs[set][3][1][1])
                   :: "memory");
    };
    auto park_x = [&](int set, int buf) {
#pragma unroll
      for (int it = 0; it < SCH; ++it)
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          if (rr == 1 && !act1) continue;
          const u32x4 lo = xs[set][it][rr][0], hi = xs[set][it][rr][1];
          u32x4 u0, u1;
          u0.x = __builtin_amdgcn_perm(hi.x, lo.x, 0x05040100u); u0.y = __builtin_amdgcn_perm(hi.x, lo.x, 0x07060302u);
          u0.z = __builtin_amdgcn_perm(hi.y, lo.y, 0x05040100u); u0.w = __builtin_amdgcn_perm(hi.y, lo.y, 0x07060302u);
          u1.x = __builtin_amdgcn_perm(hi.z, lo.z, 0x05040100u); u1.y = __builtin_amdgcn_perm(hi.z, lo.z, 0x07060302u);
          u1.z = __builtin_amdgcn_perm(hi.w, lo.w, 0x05040100u); u1.w = __builtin_amdgcn_perm(hi.w, lo.w, 0x07060302u);
          unsigned char* dst = xb + buf * XBUF + it * CHUNK + (lrow + 16 * rr) * XROW + lpc * 32;
          *reinterpret_cast<u32x4*>(dst) = u0;
          *reinterpret_cast<u32x4*>(dst + 16) = u1;
          float sm = xdot2b<BF16>(lo.x, 0.f);
          sm = xdot2b<BF16>(lo.y, sm); sm = xdot2b<BF16>(lo.z, sm); sm = xdot2b<BF16>(lo.w, sm);
          sm = xdot2b<BF16>(hi.x, sm); sm = xdot2b<BF16>(hi.y, sm); sm = xdot2b<BF16>(hi.z, sm); sm = xdot2b<BF16>(hi.w, sm);
          sm += dpp_f32<0xB1>(sm); sm += dpp_f32<0x4E>(sm); sm += dpp_f32<0x141>(sm);
          if (lpc == 0) xsum[(buf * SCH + it) * 32 + lrow + 16 * rr] = sm;
        }
    };
    issue_x(0, kstep(0));
    issue_x(1, kstep(min(1, steps - 1)));
    landed_x(0);
    park_x(0, 0);
    __syncthreads();
    for (int step = 0; step < steps; step += 2) {
      issue_x(0, kstep(min(step + 2, steps - 1)));
      landed_x(1);
      park_x(1, 1);
      __syncthreads();
      if (step + 1 >= steps) break;
      issue_x(1, kstep(min(step + 3, steps - 1)));
      landed_x(0);
      park_x(0, 0);
      __syncthreads();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else if (loader) {
    const int wt = lt - 128;
    constexpr int WP2 = (ROWS * PPR + 127) / 128;                  // pieces per thread and step (8)
    static_assert(WP2 == 8 && ROWS * 2 == 256, "built for 2 bits, 128 rows");
    const uint32_t* wsrc[WP2];
    int wdst[WP2];
#pragma unroll
    for (int p = 0; p < WP2; ++p) {
      const int idx = p * 128 + wt;
      const int row = min(idx / PPR, ROWS - 1), pc = idx % PPR;
      const uint32_t* q; const unsigned short* s_; const unsigned short* z_;
      const int nrow = row_src(row, q, s_, z_);
      wsrc[p] = q + static_cast<size_t>(nrow) * W + static_cast<size_t>(cbase) * WPC + pc * 4;
      wdst[p] = row * WROW + pc * 16;
    }
    const unsigned short* zsrc[2];
    int zdst[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int ent = e * 128 + wt, row = ent >> 1, kind = ent & 1;
      const uint32_t* q; const unsigned short* s_; const unsigned short* z_;
      const int nrow = row_src(row, q, s_, z_);
      zsrc[e] = (kind ? z_ : s_) + static_cast<size_t>(nrow) * G + cbase;
      zdst[e] = row * 16 + kind * 8;
    }
    u32x4 wl[2][WP2];
    u32x2 zl[2][2];
    auto issue_w = [&](int set, int ks) {
#pragma unroll
      for (int p = 0; p < WP2; ++p) {
        const uint32_t* src = wsrc[p] + static_cast<size_t>(ks) * (SCH * WPC);
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(wl[set][p]) : "v"(src) : "memory");
      }
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const unsigned short* zp = zsrc[e] + ks * SCH;
        asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(zl[set][e]) : "v"(zp) : "memory");
      }
    };
    auto landed_w = [&](int set) {
      asm volatile("s_waitcnt vmcnt(10)"
                   : "+v"(wl[set][0]), "+v"(wl[set][1]), "+v"(wl[set][2]), "+v"(wl[set][3]), "+v"(wl[set][4]), "+v"(wl[set][5]),
                     "+v"(wl[set][6]), "+v"(wl[set][7]), "+v"(zl[set][0]), "+v"(zl[set][1])
                   :: "memory");
    };
    auto park_w = [&](int set, int buf) {
#pragma unroll
      for (int p = 0; p < WP2; ++p) *reinterpret_cast<u32x4*>(wb + buf * WBUF + wdst[p]) = wl[set][p];
#pragma unroll
      for (int e = 0; e < 2; ++e) *reinterpret_cast<u32x2*>(zb + buf * SZBUF + zdst[e]) = zl[set][e];
    };
    issue_w(0, kstep(0));
    issue_w(1, kstep(min(1, steps - 1)));
    landed_w(0);
    park_w(0, 0);
    __syncthreads();
    for (int step = 0; step < steps; step += 2) {
      issue_w(0, kstep(min(step + 2, steps - 1)));
      landed_w(1);
      park_w(1, 1);
      __syncthreads();
      if (step + 1 >= steps) break;
      issue_w(1, kstep(min(step + 3, steps - 1)));
      landed_w(0);
      park_w(0, 0);
      __syncthreads();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else
#endif
  if (loader) {
    // ---------------- loader waves: global -> registers (two sets) -> LDS ----------------
    const int lrow = lt >> 3, lpc = lt & 7;
    const bool lact = !half_rows || lrow < 16;      // (wave-uniform) parks activations; every loader wave REQUESTS them (straight-line loads)
    const unsigned short* xrow = a.x + static_cast<size_t>(min(lrow, M - 1)) * K + lpc * 16 + static_cast<size_t>(cbase) * KC;
    const int xdst0 = lrow * XROW + lpc * 32;
    // packed words: piece p of the thread = 16 bytes [pc * 16, +16) of row `row`'s step segment
    const uint32_t* wsrc[WPT];
    int wdst[WPT];
#pragma unroll
    for (int p = 0; p < WPT; ++p) {
      const int idx = p * 256 + lt;
      const int row = min(idx / PPR, ROWS - 1), pc = idx % PPR;
      const uint32_t* q; const unsigned short* s_; const unsigned short* z_;
      const int nrow = row_src(row, q, s_, z_);
      wsrc[p] = q + static_cast<size_t>(nrow) * W + static_cast<size_t>(cbase) * WPC + pc * 4;
      wdst[p] = idx < ROWS * PPR ? row * WROW + pc * 16 : -1;
    }
    // (scale, zero): thread -> (row, kind), 8 bytes = the step's 4 groups
    const unsigned short* zsrc;
    int zdst;
    {
      const int row = min(lt >> 1, ROWS - 1), kind = lt & 1;
      const uint32_t* q; const unsigned short* s_; const unsigned short* z_;
      const int nrow = row_src(row, q, s_, z_);
      zsrc = (kind ? z_ : s_) + static_cast<size_t>(nrow) * G + cbase;
      zdst = lt < ROWS * 2 ? row * 16 + kind * 8 : -1;
    }
    u32x4 xs[2][SCH][2];
    u32x4 wl[2][WPT];
    u32x2 zl[2];
#ifdef BX_KM
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
      for (int it = 0; it < SCH; ++it) { xs[q][it][0] = u32x4{0, 0, 0, 0}; xs[q][it][1] = u32x4{0, 0, 0, 0}; }
#pragma unroll
      for (int p = 0; p < WPT; ++p) wl[q][p] = u32x4{0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u};
      zl[q] = u32x2{0x3C003C00u, 0};
    }
#endif
    // -DBX_KM knock-out builds (timing only): 8 no weight loads, 1 no activation loads
    // The loads are inline assembly and the waits are written by hand (NL loads per set, retired in order): the compiler's own
    // s_waitcnt insertion loses count across the loop's back edge and waits for the NEW set before it lets the old one be used,
    // which leaves one step in flight instead of two.  Every register of a set is tied to the wait that covers it.
    auto issue_w = [&](int set, int ks) {
#if !(defined(BX_KM) && (BX_KM & 8))
#pragma unroll
      for (int p = 0; p < WPT; ++p) {
        const uint32_t* src = wsrc[p] + static_cast<size_t>(ks) * (SCH * WPC);
#ifdef BX2_PLAIN_LOADS
        wl[set][p] = *reinterpret_cast<const u32x4*>(src);
#else
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(wl[set][p]) : "v"(src) : "memory");
#endif
      }
      const unsigned short* zp = zsrc + ks * SCH;
#ifdef BX2_PLAIN_LOADS
      zl[set] = *reinterpret_cast<const u32x2*>(zp);
#else
      asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(zl[set]) : "v"(zp) : "memory");
#endif
#endif
    };
    auto issue_x = [&](int set, int ks, int it) {
#if !(defined(BX_KM) && (BX_KM & 1))
      const unsigned short* p = xrow + (ks * SCH + it) * KC;
#ifdef BX2_PLAIN_LOADS
      xs[set][it][0] = *reinterpret_cast<const u32x4*>(p);
      xs[set][it][1] = *reinterpret_cast<const u32x4*>(p + 8);
#else
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(xs[set][it][0]) : "v"(p) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(xs[set][it][1]) : "v"(p) : "memory");
#endif
#endif
    };
    constexpr int NL = WPT + 1 + 2 * SCH;           // loads per set
    auto landed = [&](int set) {                     // wait until only the NL loads of the other set are outstanding; set `set` is then valid
#ifdef BX2_PLAIN_LOADS
      return;
#endif
#ifdef BX2_WAIT_ALL
      constexpr int Y = 0;
#else
      constexpr int Y = NL;
#endif
      static_assert(SCH == 4, "operand list below");
      asm volatile("s_waitcnt vmcnt(%9)"
                   : "+v"(xs[set][0][0]), "+v"(xs[set][0][1]), "+v"(xs[set][1][0]), "+v"(xs[set][1][1]), "+v"(xs[set][2][0]), "+v"(xs[set][2][1]),
                     "+v"(xs[set][3][0]), "+v"(xs[set][3][1]), "+v"(zl[set])
                   : "n"(Y) : "memory");
      // (one statement per register: a variable listed twice in one statement is copied BEFORE it -- a copy of a register still in flight)
#pragma unroll
      for (int p = 0; p < WPT; ++p) asm volatile("" : "+v"(wl[set][p]) :: "memory");
    };
    auto park_w = [&](int set, int buf) {            // registers -> LDS buffers `buf`
#pragma unroll
      for (int p = 0; p < WPT; ++p)
        if (wdst[p] >= 0) *reinterpret_cast<u32x4*>(wb + buf * WBUF + wdst[p]) = wl[set][p];
      if (zdst >= 0) *reinterpret_cast<u32x2*>(zb + buf * SZBUF + zdst) = zl[set];
    };
    auto park_x = [&](int set, int buf, int it) {
      const u32x4 lo = xs[set][it][0], hi = xs[set][it][1];
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
      unsigned char* dst = xb + buf * XBUF + it * CHUNK + xdst0;
      *reinterpret_cast<u32x4*>(dst) = u0;
      *reinterpret_cast<u32x4*>(dst + 16) = u1;
      float sm = xdot2b<BF16>(lo.x, 0.f);
      sm = xdot2b<BF16>(lo.y, sm); sm = xdot2b<BF16>(lo.z, sm); sm = xdot2b<BF16>(lo.w, sm);
      sm = xdot2b<BF16>(hi.x, sm); sm = xdot2b<BF16>(hi.y, sm); sm = xdot2b<BF16>(hi.z, sm); sm = xdot2b<BF16>(hi.w, sm);
      sm += dpp_f32<0xB1>(sm);                      // 2 pieces = 32 k
      sm += dpp_f32<0x4E>(sm);                      // 4 pieces = 64 k
      sm += dpp_f32<0x141>(sm);                     // 8 pieces: the row chunk
      if (lpc == 0) xsum[(buf * SCH + it) * 32 + lrow] = sm;
    };
    // one turn of the loader: request step `nstep` into register set `ns` and turn set `os` (requested a turn ago) into LDS
    // buffers `buf`.  The requests are interleaved with the parking work chunk by chunk: the CU's vector-memory path takes a
    // request only when it has room, and a loader that issues all of them first sits blocked for half the step with its
    // parking work still ahead of it.
    // Requests and parking are unconditional (straight-line code lets the compiler count its s_waitcnt vmcnt exactly; a branch around a
    // load makes it wait for everything): past the end of the k range a turn re-requests the last step and parks into a buffer
    // nobody reads any more.
    auto turn = [&](int ns, int nstep, int os, int buf, bool do_park = true) {
      const int ks = kstep(min(nstep, steps - 1));
      issue_w(ns, ks);
#pragma unroll
      for (int it = 0; it < SCH; ++it) issue_x(ns, ks, it);
      landed(os);
#ifdef BX2_COND_PARK
      if (!do_park) return;
#endif
      park_w(os, buf);
      if (lact) {
#pragma unroll
        for (int it = 0; it < SCH; ++it) park_x(os, buf, it);
      }
    };
    {                                               // prologue: steps 0 and 1 requested, step 0 parked
      const int k0 = kstep(0), k1 = kstep(min(1, steps - 1));
      issue_w(0, k0);
#pragma unroll
      for (int it = 0; it < SCH; ++it) issue_x(0, k0, it);
      issue_w(1, k1);
#pragma unroll
      for (int it = 0; it < SCH; ++it) issue_x(1, k1, it);
      landed(0);
      park_w(0, 0);
      if (lact) {
#pragma unroll
        for (int it = 0; it < SCH; ++it) park_x(0, 0, it);
      }
    }
    STAMP2();
    __syncthreads();
    for (int step = 0; step < steps; step += 2) {
      // even step: set 0 is free (its step was parked before the last barrier), set 1 holds step + 1
      STAMP2();
      turn(0, step + 2, 1, 1, step + 1 < steps);
      STAMP2(); STAMP2();
      __syncthreads();
      STAMP2();
      if (step + 1 >= steps) break;
      STAMP2();
      turn(1, step + 3, 0, 0, step + 2 < steps);
      STAMP2(); STAMP2();
      __syncthreads();
      STAMP2();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the requests past the end land in registers the epilogue is about to reuse
  } else {
    // ---------------- compute waves: LDS -> matrix cores ----------------
    const int wrow = (ng * TNW + r) * WROW + h * (WL * 4);
    const int zrow = (ng * TNW + r) * 16;
    STAMP2();
    __syncthreads();
    for (int step = 0; step < steps; ++step) {
      const int buf = step & 1;
      STAMP2();
      unsigned wc[NCHW][WL];
#pragma unroll
      for (int j = 0; j < NCHW; ++j) {
        const unsigned char* src = wb + buf * WBUF + wrow + (kg * NCHW + j) * (WPC * 4);
        if constexpr (WL % 4 == 0) {
#pragma unroll
          for (int q = 0; q < WL; q += 4) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(src + q * 4);
            wc[j][q] = v.x; wc[j][q + 1] = v.y; wc[j][q + 2] = v.z; wc[j][q + 3] = v.w;
          }
        } else if constexpr (WL % 2 == 0) {
#pragma unroll
          for (int q = 0; q < WL; q += 2) {
            const u32x2 v = *reinterpret_cast<const u32x2*>(src + q * 4);
            wc[j][q] = v.x; wc[j][q + 1] = v.y;
          }
        } else {
#pragma unroll
          for (int q = 0; q < WL; ++q) wc[j][q] = *reinterpret_cast<const unsigned*>(src + q * 4);
        }
      }
      const u32x4 szv = *reinterpret_cast<const u32x4*>(zb + buf * SZBUF + zrow);      // s01 s23 z01 z23
      const unsigned szw[4] = {szv.x, szv.y, szv.z, szv.w};
      // The field builder below is inline assembly, and the compiler's s_waitcnt insertion does not cover a register whose FIRST
      // consumer is an asm operand (measured: word 1 of the first chunk, used unshifted, was read before the ds_read had landed --
      // wrong sums on some launches).  The packed words are waited for by hand and tied to the wait.
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < NCHW; ++j)
#pragma unroll
        for (int q = 0; q < WL; ++q) asm volatile("" : "+v"(wc[j][q]));
      STAMP2();
#pragma unroll
      for (int j = 0; j < NCHW; ++j) {
        const int cs = kg * NCHW + j;
        const unsigned char* slab = xb + buf * XBUF + cs * CHUNK;
        f32x16 g32;
        f32x4v g16[TNW == 16 ? MT : 1];
#pragma unroll
        for (int t = 0; t < UPL; ++t) {
          u32x4 bf;
          if constexpr (BITS == 2) {
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
          const int aoff = (h * UPL + t) * 16;
          const bool first = t == 0;                  // first step of a group: the accumulator tile restarts from zero
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
        }
        // group flush: acc += s * t + (z - C s) * xsum[row]
        const unsigned sw = szw[cs >> 1], zw = szw[2 + (cs >> 1)];
        const float s = act_to_f32<BF16>(static_cast<unsigned short>((cs & 1) ? (sw >> 16) : (sw & 0xFFFFu)));
        const float zz = act_to_f32<BF16>(static_cast<unsigned short>((cs & 1) ? (zw >> 16) : (zw & 0xFFFFu))) - CF * s;
        const float* xsm = xsum + (buf * SCH + cs) * 32;
        if constexpr (TNW == 32) {
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const f32x4v xv = *reinterpret_cast<const f32x4v*>(xsm + 8 * jj + 4 * (lane >> 5));
#pragma unroll
            for (int e = 0; e < 4; ++e) acc32[4 * jj + e] = fmaf(s, g32[4 * jj + e], fmaf(zz, xv[e], acc32[4 * jj + e]));
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
#ifdef BX_STAMPS
      __builtin_amdgcn_sched_barrier(0);
#endif
      STAMP2();
      __syncthreads();
      STAMP2();
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
      if constexpr (TNW == 32) v = acc32[e]; else v = acc16[i][e];
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
  } else {
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
  }
#ifdef BX_STAMPS
  STAMP2();
  if (lane == 0 && bx2_stamp_ptr) {
    long long* o = bx2_stamp_ptr + (static_cast<size_t>(blockIdx.x) * 12 + wave) * 24;
#pragma unroll
    for (int i = 0; i < 24; ++i) o[i] = i < sti ? st[i] : 0;
  }
#endif
}

template <int BITS, int TNW, int NG, int KG>
static hipError_t launch_batch2_cfg(const MfmaArgs& a, bool bf16, hipStream_t s) {
  constexpr int ROWS = NG * TNW;
  constexpr int SB = 4 * (KC * BITS / 32) * 4;
#ifdef BX2_WROW
  constexpr int SHMEM = 2 * 4 * 32 * XROW + 2 * ROWS * BX2_WROW + 2 * ROWS * 16 + 2 * 4 * 32 * 4;
#else
  constexpr int SHMEM = 2 * 4 * 32 * XROW + 2 * ROWS * (SB + 16) + 2 * ROWS * 16 + 2 * 4 * 32 * 4;
#endif
  static_assert(SHMEM <= 160 * 1024 && SHMEM >= 8 * (32 / TNW) * (TNW == 32 ? 16 : 4) * 64 * 4, "LDS budget");
  static DeviceOnce done[2];
  if (done[bf16].pending()) {
#ifdef BX2_ONE
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&qgemm_batch2_kernel<BITS, TNW, NG, KG, true>), hipFuncAttributeMaxDynamicSharedMemorySize, SHMEM);
#else
    hipError_t e = bf16 ? hipFuncSetAttribute(reinterpret_cast<const void*>(&qgemm_batch2_kernel<BITS, TNW, NG, KG, true>), hipFuncAttributeMaxDynamicSharedMemorySize, SHMEM)
                        : hipFuncSetAttribute(reinterpret_cast<const void*>(&qgemm_batch2_kernel<BITS, TNW, NG, KG, false>), hipFuncAttributeMaxDynamicSharedMemorySize, SHMEM);
#endif
    if (e != hipSuccess) return e;
    done[bf16].done();
  }
  dim3 grid(a.swiglu ? (a.n_seg[0] + ROWS / 2 - 1) / (ROWS / 2) : (a.N + ROWS - 1) / ROWS, a.ws ? a.ksplit : 1), block(768);
#ifdef BX_STAMPS
  long long* sp = reinterpret_cast<long long*>(strtoull(getenv("GBX_STAMP_PTR") ? getenv("GBX_STAMP_PTR") : "0", nullptr, 0));
  (void)hipMemcpyToSymbol(HIP_SYMBOL(bx2_stamp_ptr), &sp, sizeof(sp));
#endif
#ifdef BX2_ONE
  hipLaunchKernelGGL((qgemm_batch2_kernel<BITS, TNW, NG, KG, true>), grid, block, SHMEM, s, a);
#else
  if (bf16) hipLaunchKernelGGL((qgemm_batch2_kernel<BITS, TNW, NG, KG, true>), grid, block, SHMEM, s, a);
  else hipLaunchKernelGGL((qgemm_batch2_kernel<BITS, TNW, NG, KG, false>), grid, block, SHMEM, s, a);
#endif
  return hipGetLastError();
}

// rows per workgroup 128 / 64 / 32 -> (TNW, NG, KG) as in mfma_batch.hip; false: the shape is not built here (16-row workgroups,
// groups under 128, a k range per workgroup that is not a multiple of 512)
bool qgemm_batch2_ok(const MfmaArgs& a, int rows) {
  static const bool on = !(getenv("GBX_BATCH_V2") && getenv("GBX_BATCH_V2")[0] == '0');
  const int ksplit = a.ws ? a.ksplit : 1;
#ifdef BX2_ONE
  if (rows != 128) return false;
#endif
  return on && a.log2g == 7 && rows >= 32 && (a.K / ksplit) % 512 == 0 && a.K % ksplit == 0;
}

template <int BITS>
static hipError_t launch_batch2_bits(const MfmaArgs& a, int rows, bool bf16, hipStream_t s) {
#ifdef BX2_ONE                                      // experiment builds: one shape (seconds instead of minutes to compile)
  if (BITS != 2 || rows != 128 || !bf16) return hipErrorInvalidValue;
  if constexpr (BITS == 2) return launch_batch2_cfg<2, 32, 4, 2>(a, true, s);
#else
  switch (rows) {
    case 128: return launch_batch2_cfg<BITS, 32, 4, 2>(a, bf16, s);
    case 64: return launch_batch2_cfg<BITS, 32, 2, 4>(a, bf16, s);
    default: return launch_batch2_cfg<BITS, 16, 2, 4>(a, bf16, s);
  }
#endif
  return hipErrorInvalidValue;
}

hipError_t launch_qgemm_batch2(const MfmaArgs& a, int bits, int rows, bool bf16, hipStream_t s) {
  return bits == 2 ? launch_batch2_bits<2>(a, rows, bf16, s) : bits == 3 ? launch_batch2_bits<3>(a, rows, bf16, s) : launch_batch2_bits<4>(a, rows, bf16, s);
}

}  // namespace gbx
