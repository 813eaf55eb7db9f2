// Dense y[M,N] = x[M,K] . w[N,K]^T on the matrix cores for LONG prompts: the second half of the "dequantise once per call"
// prefill path (mfma_gemm.hip: launch_qgemm_dense).  The weights arrive already dequantised to the activation dtype
// (gbx_dequantize arithmetic: fields -> fp32, w = s*q + z, one rounding), so this is a plain 16-bit GEMM and the kernel can
// spend everything on keeping the matrix pipe fed:
//   * workgroup tile 256 x 256, K tiles of 64, 8 waves as 2 (M) x 4 (N), a wave owns 128 x 64 = 8 x 4 accumulators of
//     v_mfma_f32_16x16x32 (128 VGPRs) -- the 256^2 "8-phase" structure of cdna_hip_programming.md section 5;
//   * both operands go HBM/L2 -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`, no VGPR staging, no ds_write pass), four
//     16 KB half-tiles per K tile (A rows 0-127 / 128-255, B rows 0-127 / 128-255), two K tiles of LDS (128 KB);
//     the LDS image is lane-linear, an XOR swizzle of the 16-byte column with (row >> 1) & 7 is applied to the SOURCE column
//     of every lane: a 16 x 32 fragment read (lane = row l & 15, k block l >> 4) then hits 16 different bank quads in each of
//     ds_read_b128's four lane groups (PMC: the guide's st_16x32 swizzle left 49 % of the LDS cycles as bank conflicts here);
//   * four phases per K tile, one C quadrant (16 MFMAs) each, fragments of the whole K tile are in registers after phase 2,
//     so the tile's LDS slots are restaged from phase 3 on: the four half-tiles of tile t+2 are requested in phases 3, 4 of
//     tile t and 1, 2 of tile t+1 and waited for ONCE per K tile with a counted vmcnt(4) (never 0 in the loop);
//   * the two wave groups (waves 0-3 / 4-7: one wave of each on every SIMD) run one barrier interval apart, so one group's
//     MFMA segment covers the other's ds_read / DMA-issue segment.
// Ordering rules this schedule obeys (MI355X_MICROARCH.md, cdna_hip_programming.md section 5): a staged buffer is read one
// phase after the counted wait + barrier that retire it (RAW); a buffer is restaged only after a barrier that follows the
// lgkmcnt(0) retiring its last ds_read in every wave (WAR); raw s_barrier, never __syncthreads() (which would drain vmcnt).
#include "mfma_common.h"
#include "gbx_env.h"
#include "kernels.h"

namespace gbx {

namespace {

constexpr int DT = 256;            // tile edge (M and N)
constexpr int DK = 64;             // K tile
constexpr int HALF = 128 * DK * 2; // bytes of one half-tile (128 rows x 64 k)
constexpr int TILE_LDS = 4 * HALF; // A-lo | A-hi | B-lo | B-hi
constexpr int DENSE_LDS = 8 * 128 * 144;   // 2 K tiles (131072 B) during the loop; 8 wave-private 128 x 144 B output tiles in the epilogue

#define DLDSP(p) ((__attribute__((address_space(3))) void*)(p))
__device__ __forceinline__ constexpr int wc_vm(int n) { return (n & 0xF) | ((n >> 4) << 14) | (0x7 << 4) | (0xF << 8); }
__device__ __forceinline__ void bar() { asm volatile("s_barrier" ::: "memory"); }

struct DenseArgs2 {
  const unsigned short* x;     // [M, K]
  const unsigned short* w;     // [N, K]
  const unsigned short* bias;  // [N] or null
  unsigned short* y;           // [M, N] ([M, N / 2] with swiglu)
  int M, N, K;
  int swiglu;                  // w rows alternate in blocks of 128: gate rows | up rows; y = silu(gate) * up
  // few-tile shapes: the k range split over `ksplit` workgroups per tile, slice s writes its fp32 partial tile to part[s][M][N]
  // (caller scratch) and a second launch sums the slices, adds the bias and rounds (launch_splitk_finish)
  float* part;
  int ksplit;
};

template <bool BF16>
__global__ __launch_bounds__(512) void gemm_dense_kernel(const DenseArgs2 a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wcn = wave & 3;          // wave row (M half) / wave column (64 of the 256 N)
  const int M = a.M, N = a.N, K = a.K;
  const int ksplit = a.part ? a.ksplit : 1;
  const int nk = K / DK / ksplit;                    // K tiles of this workgroup's slice

  // XCD-aware tile order (as mfma_gemm.hip): every XCD walks a contiguous run of (tile, k slice) units, M-tile major
  const int tiles_n = (N + DT - 1) / DT, tiles_m = (M + DT - 1) / DT, tiles = tiles_m * tiles_n * ksplit;
  const int per_xcd = (tiles + 7) / 8;
  const int unit = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (unit >= tiles) return;
  const int tile = unit / ksplit, kslice = unit % ksplit;
  const int m0 = (tile / tiles_n) * DT, n0 = (tile % tiles_n) * DT;
  const int kbase = kslice * nk * (DK * 2);          // byte offset of the slice inside a row

  // ---- DMA side: piece p = tid + 512 i (i = 0, 1) of a half-tile = (row p >> 3, 16-byte column p & 7); LDS linear, the
  // swizzle on the source column: slot (row, c) holds logical column c ^ ((row >> 1) & 7)
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.x), 0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.w), 0, 0x7FFFFFFF, 0x00020000);
  int voa[2][2], vob[2][2];                          // [half][i] byte offsets of this lane's pieces at k = 0
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int p = tid + 512 * i, row = p >> 3, c = (p & 7) ^ ((row >> 1) & 7);
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      voa[hf][i] = (min(m0 + hf * 128 + row, M - 1) * K + c * 8) * 2;
      vob[hf][i] = (min(n0 + hf * 128 + row, N - 1) * K + c * 8) * 2;
    }
  }
  const int dma_lds = wave * 1024;                   // + 8192 i: this wave's 1 KiB run inside a half-tile slot
  // half-tile h (0 A-lo, 1 A-hi, 2 B-lo, 3 B-hi) of K tile kt -> LDS buffer kt & 1
  auto stage = [&](int kt, int h) {
    const int ktc = min(kt, nk - 1);                 // past the end: re-request the last tile (keeps the vmcnt arithmetic uniform)
    unsigned char* dst = lds + (kt & 1) * TILE_LDS + h * HALF + dma_lds;
    const int koff = kbase + ktc * (DK * 2);
    if (h < 2) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, DLDSP(dst), 16, voa[h][0], koff, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, DLDSP(dst + 8192), 16, voa[h][1], koff, 0, 0);
    } else {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, DLDSP(dst), 16, vob[h - 2][0], koff, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, DLDSP(dst + 8192), 16, vob[h - 2][1], koff, 0, 0);
    }
  };

  // ---- fragment side: lane = (row l & 15, k block l >> 4) of a 16 x 32 operand; swizzled column
  const int fr = lane & 15, fk = lane >> 4;
  const int sw = (fr >> 1) & 7;
  const int fa_off = wr * HALF + fr * 128;                          // + rt * 2048 + ((ks * 4 + fk) ^ sw) * 16
  const int fb_off = (2 + (wcn >> 1)) * HALF + ((wcn & 1) * 64 + fr) * 128;
  const int c0 = (fk ^ sw) * 16, c1 = ((4 + fk) ^ sw) * 16;         // k-step 0 / 1 column bytes
  u32x4 fa[8][2], fb[4][2];
  auto read_a = [&](const unsigned char* buf, int half) {           // row tiles 4 half .. 4 half + 3
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      fa[half * 4 + rt][0] = *reinterpret_cast<const u32x4*>(buf + fa_off + (half * 4 + rt) * 2048 + c0);
      fa[half * 4 + rt][1] = *reinterpret_cast<const u32x4*>(buf + fa_off + (half * 4 + rt) * 2048 + c1);
    }
  };
  auto read_b = [&](const unsigned char* buf, int half) {           // column tiles 2 half, 2 half + 1
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      fb[half * 2 + ct][0] = *reinterpret_cast<const u32x4*>(buf + fb_off + (half * 2 + ct) * 2048 + c0);
      fb[half * 2 + ct][1] = *reinterpret_cast<const u32x4*>(buf + fb_off + (half * 2 + ct) * 2048 + c1);
    }
  };
  f32x4v acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4v{0};
  auto quadrant = [&](int ah, int bh) {
#ifndef DENSE_NOPRIO
    __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        acc[ah * 4 + rt][bh * 2 + ct] = mfma16<BF16>(fa[ah * 4 + rt][0], fb[bh * 2 + ct][0], acc[ah * 4 + rt][bh * 2 + ct]);
        acc[ah * 4 + rt][bh * 2 + ct] = mfma16<BF16>(fa[ah * 4 + rt][1], fb[bh * 2 + ct][1], acc[ah * 4 + rt][bh * 2 + ct]);
      }
#ifndef DENSE_NOPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
  };

  // ---- prologue: K tile 0 whole, the first two half-tiles of K tile 1
#pragma unroll
  for (int h = 0; h < 4; ++h) stage(0, h);
  stage(1, 0);
  stage(1, 1);
  __builtin_amdgcn_s_waitcnt(wc_vm(4));              // K tile 0 has landed (this wave's share); tile 1's two may fly
  bar();
  if (wr == 1) bar();                                // the second wave group runs one barrier interval behind the first

  for (int t = 0; t < nk; ++t) {
    const unsigned char* buf = lds + (t & 1) * TILE_LDS;
    // phase 1: B0 + A0 -> quadrant (A0, B0); request half-tile 2 of K tile t+1
    read_b(buf, 0);
    __builtin_amdgcn_sched_barrier(0);
    read_a(buf, 0);
    stage(t + 1, 2);
    bar();
    quadrant(0, 0);
    bar();
    // phase 2: B1 + A1 (the K tile is in registers afterwards) -> quadrant (A0, B1); request half-tile 3 of K tile t+1
    read_b(buf, 1);
    __builtin_amdgcn_sched_barrier(0);
    read_a(buf, 1);
    stage(t + 1, 3);
    __builtin_amdgcn_s_waitcnt(0xC07F);              // lgkmcnt(0): every fragment read of this buffer retired before the barrier
    bar();
    quadrant(0, 1);
    bar();
    // phase 3: quadrant (A1, B1); buffer t & 1 is free: request half-tile 0 of K tile t+2 into it
    stage(t + 2, 0);
    bar();
    quadrant(1, 1);
    bar();
    // phase 4: quadrant (A1, B0); request half-tile 1 of K tile t+2; K tile t+1 must have landed: all but the 4 youngest DMAs
    stage(t + 2, 1);
    __builtin_amdgcn_s_waitcnt(wc_vm(4));
    bar();
    quadrant(1, 0);
    bar();
  }
  if (wr == 0) bar();                                // barrier counts of the two groups match again
  __builtin_amdgcn_s_waitcnt(wc_vm(0));              // no LDS-DMA may outlive the workgroup

  // ---- epilogue: one rounding to the activation dtype (+ bias as a second rounding, like the reference's x + bias).
  // A lane holds 4 consecutive ROWS of one column per 16x16 tile: stored from the registers that is 128 two-byte stores per
  // lane in 32-byte segments (measured: ~30 us per workgroup, a quarter of the K = 4096 tile).  The wave's 128 x 64 tile
  // goes through its own LDS region instead (rows of 128 B + 16 B pad) and leaves as 16-byte row-contiguous stores.
  bar();                                             // every wave is past its last fragment read / DMA wait: LDS is free
  if (ksplit > 1) {
    // fp32 partial tile of this k slice, through the wave's own LDS region in two passes of 64 rows (rows of 256 B + 16 B pad)
    // so that it leaves as 16-byte row-contiguous stores
    constexpr int PROW = 64 * 4 + 16;
    unsigned char* pt = lds + wave * (64 * PROW);
    float* dst0 = a.part + (static_cast<size_t>(kslice) * M) * N;
    const int nbp = n0 + wcn * 64, mbp = m0 + wr * 128;
    const bool full = (nbp + 64 <= N) && (N % 4 == 0);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            *reinterpret_cast<float*>(pt + (rt * 16 + 4 * fk + e) * PROW + (ct * 16 + fr) * 4) = acc[half * 4 + rt][ct][e];
      __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int p = it * 64 + lane, row = p >> 4, c = p & 15;    // 16-byte piece c (4 floats) of row `row`
        const int m = mbp + half * 64 + row;
        if (m >= M) continue;
        const u32x4 v = *reinterpret_cast<const u32x4*>(pt + row * PROW + c * 16);
        float* dst = dst0 + static_cast<size_t>(m) * N + nbp + c * 4;
        if (full) {
          *reinterpret_cast<u32x4*>(dst) = v;
        } else {
          const unsigned wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (nbp + c * 4 + q < N) dst[q] = __builtin_bit_cast(float, wv[q]);
        }
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);              // this pass's reads retired before the next pass overwrites the region
    }
    return;
  }
  constexpr int EROW = 144;
  unsigned char* et = lds + wave * (128 * EROW);
  const int nb = n0 + wcn * 64, mb = m0 + wr * 128;
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    const int n = min(nb + ct * 16 + fr, N - 1);
    const float b = a.bias ? act_to_f32<BF16>(a.bias[n]) : 0.f;
#pragma unroll
    for (int rt = 0; rt < 8; ++rt)
#pragma unroll
      for (int e = 0; e < 4; ++e) {                  // C/D layout of the 16x16 tile: row = 4 (lane >> 4) + e, column = lane & 15
        float v = acc[rt][ct][e];
        if (a.bias) v = act_to_f32<BF16>(f32_to_act<BF16>(v)) + b;
        *reinterpret_cast<unsigned short*>(et + (rt * 16 + 4 * fk + e) * EROW + (ct * 16 + fr) * 2) = f32_to_act<BF16>(v);
      }
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);                // the wave's own LDS writes (no other wave touches this region)
  if (a.swiglu) {
    // Tile columns 0-127 are gate rows j, columns 128-255 the up rows of the SAME j (waves wcn and wcn + 2 of one wave row):
    // y[m, n0 / 2 + j] = round(round(silu(gate)) * up) on the already rounded GEMM outputs -- the reference's op-by-op
    // roundings (qllama.py:115), what gbx_swiglu computes from the two tensors this epilogue never writes.
    bar();
    const int NO = N >> 1;
    const unsigned char* gt = lds + (wr * 4 + (wcn & 1)) * (128 * EROW);
    const unsigned char* ut = lds + (wr * 4 + 2 + (wcn & 1)) * (128 * EROW);
    const int ob = (n0 >> 1) + (wcn & 1) * 64;       // output column of this wave's piece 0
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int p = it * 64 + lane, row = (wcn >> 1) * 64 + (p >> 3), c = p & 7;
      const int m = mb + row;
      if (m >= M) continue;
      const u32x4 gv = *reinterpret_cast<const u32x4*>(gt + row * EROW + c * 16);
      const u32x4 uv = *reinterpret_cast<const u32x4*>(ut + row * EROW + c * 16);
      const unsigned gw[4] = {gv.x, gv.y, gv.z, gv.w}, uw[4] = {uv.x, uv.y, uv.z, uv.w};
      unsigned o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float g0 = act_lo<BF16>(gw[j]), g1 = act_hi<BF16>(gw[j]);
        const float s0 = act_to_f32<BF16>(f32_to_act<BF16>(g0 / (1.f + __expf(-g0))));
        const float s1 = act_to_f32<BF16>(f32_to_act<BF16>(g1 / (1.f + __expf(-g1))));
        o[j] = static_cast<unsigned>(f32_to_act<BF16>(s0 * act_lo<BF16>(uw[j]))) |
               (static_cast<unsigned>(f32_to_act<BF16>(s1 * act_hi<BF16>(uw[j]))) << 16);
      }
      *reinterpret_cast<u32x4*>(a.y + static_cast<size_t>(m) * NO + ob + c * 8) = u32x4{o[0], o[1], o[2], o[3]};
    }
    return;
  }
  const bool full_n = (nb + 64 <= N) && (N % 8 == 0);
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int p = it * 64 + lane, row = p >> 3, c = p & 7;       // 16-byte piece c of row `row`
    const int m = mb + row;
    if (m >= M) continue;
    const u32x4 v = *reinterpret_cast<const u32x4*>(et + row * EROW + c * 16);
    unsigned short* dst = a.y + static_cast<size_t>(m) * N + nb + c * 8;
    if (full_n) {
      *reinterpret_cast<u32x4*>(dst) = v;
    } else {
      const unsigned wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (nb + c * 8 + q < N) dst[q] = static_cast<unsigned short>(wv[q >> 1] >> (16 * (q & 1)));
    }
  }
}


// ---- M <= 32 (the output head of a decode batch: 32 x 128256 x 4096 = 1.05 GB of weights per launch) -------------------------
// The 256 x 256 tile above spends 8x the matrix-core work on padding rows at M = 32 (239 us, 4.4 TB/s: MFMA bound).  Here the
// launch is a stream: a workgroup (NWV waves) owns 32 NWV weight rows over the whole k range, a step is 128 k-values -- the
// weight tile and the 8 KB activation tile are requested as whole 256-byte row segments (16 lanes per row, 8 fully used lines
// per wave instruction), parked in ONE LDS image (row stride 272 B: conflict-free ds_read_b128 of 32x32x16 fragments) and
// consumed by 8 MFMAs per wave; the next step's requests are issued before the MFMAs.  Every workgroup pulls the whole 256 KB
// of x through its CU beside its weight rows, and the CU's fill path carries the SUM: 128-row workgroups (3 per CU)
// add 1 MB of x to 4.1 MB of weights per CU and stream at 4.5 TB/s; 256-row workgroups (8 waves, 78 KB of LDS, two per CU, 501
// workgroups = one round) halve that.  Nothing else is close to a limit (8 MFMAs = 256 cycles against ~4000 cycles of fill per step).
constexpr int SK = 128;                 // k-values per step
constexpr int SROW = SK * 2 + 16;       // LDS row stride
constexpr int skinny_lds(int nwv) { return (32 * nwv + 32) * SROW; }

template <bool BF16, int NWV>
__global__ __launch_bounds__(NWV * 64) void dense_batch_kernel(const DenseArgs2 a) {
  constexpr int ROWS = 32 * NWV, NT = 64 * NWV, WIT = ROWS * 16 / NT, XIT = 32 * 16 / NT, RPI = NT / 16;   // pieces per thread, rows per iteration
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* const wl_s = lds;                    // [ROWS][SROW]
  unsigned char* const xl_s = lds + ROWS * SROW;      // [32 rows][SROW]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int M = a.M, N = a.N, K = a.K;
  const int n0 = blockIdx.x * ROWS;
  const int steps = K / SK;
  // loader mapping: piece idx = it * NT + tid -> (row = idx >> 4, 16-byte piece pc = idx & 15)
  const int prow = tid >> 4, pc = tid & 15;
  const unsigned short* wsrc = a.w + static_cast<size_t>(min(n0 + prow, N - 1)) * K + pc * 8;
  const size_t wrowi = static_cast<size_t>(RPI) * K;  // it -> RPI rows further
  const int pdst = prow * SROW + pc * 16;
  // two register sets: the requests of step s+2 go out BEFORE the registers of step s+1 are parked.  The loads are inline assembly
  // with hand-counted waits (NL per set, retired in order): the compiler's own s_waitcnt insertion loses count across the back edge
  // and waits for the newer set too.  Requests are unconditional -- past the end the last step again.
  static_assert(WIT == 8 && XIT == 1, "operand lists below are written for 8 waves");
  const int rot = static_cast<int>(blockIdx.x % static_cast<unsigned>(steps));
  constexpr int NL = WIT + XIT;
  u32x4 wv[2][WIT], xv[2][XIT];
  auto request = [&](int set, int step) {
    // every workgroup walks k from a different starting step: rows are K * 2 bytes apart, so the same k offset of every row is the
    // same few HBM channels (the camping mfma_batch.hip avoids the same way)
    int ks = min(step, steps - 1) + rot;
    ks = ks >= steps ? ks - steps : ks;
    const int k0 = ks * SK;
#pragma unroll
    for (int it = 0; it < WIT; ++it) {
      const int rr = n0 + prow + RPI * it;
      const unsigned short* src = (rr < N ? wsrc + it * wrowi : wsrc) + k0;
      asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(wv[set][it]) : "v"(src) : "memory");     // streamed once
    }
    const unsigned short* xsrc = a.x + static_cast<size_t>(min(prow, M - 1)) * K + pc * 8 + k0;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(xv[set][0]) : "v"(xsrc) : "memory");
  };
  auto landed = [&](int set) {                       // only the NL requests of the other set may still be outstanding
    asm volatile("s_waitcnt vmcnt(%9)"
                 : "+v"(wv[set][0]), "+v"(wv[set][1]), "+v"(wv[set][2]), "+v"(wv[set][3]), "+v"(wv[set][4]), "+v"(wv[set][5]),
                   "+v"(wv[set][6]), "+v"(wv[set][7]), "+v"(xv[set][0])
                 : "n"(NL) : "memory");
  };
  auto park = [&](int set) {
#pragma unroll
    for (int it = 0; it < WIT; ++it) *reinterpret_cast<u32x4*>(wl_s + pdst + it * RPI * SROW) = wv[set][it];
#pragma unroll
    for (int it = 0; it < XIT; ++it) *reinterpret_cast<u32x4*>(xl_s + pdst + it * RPI * SROW) = xv[set][it];
  };
  const int r = lane & 31, h = lane >> 5;
  const unsigned char* wfrag = wl_s + (wave * 32 + r) * SROW + h * 16;
  const unsigned char* xfrag = xl_s + r * SROW + h * 16;
  f32x16 acc = f32x16{0};
  auto compute = [&]() {
#pragma unroll
    for (int t = 0; t < SK / 16; ++t) {
      const u32x4 bf = *reinterpret_cast<const u32x4*>(wfrag + t * 32);
      const u32x4 af = *reinterpret_cast<const u32x4*>(xfrag + t * 32);
      acc = mfma<BF16>(af, bf, acc);
    }
  };
  request(0, 0);
  request(1, 1);
  for (int step = 0; step < steps; step += 2) {
    __syncthreads();                                  // the previous step's fragment reads are done
    landed(0);
    park(0);
    request(0, step + 2);
    __syncthreads();
    compute();
    if (step + 1 >= steps) break;
    __syncthreads();
    landed(1);
    park(1);
    request(1, step + 3);
    __syncthreads();
    compute();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the requests past the end land in registers the epilogue reuses
  // one rounding to the activation dtype (+ bias as a second rounding, like the reference's x + bias)
  const int n = n0 + wave * 32 + r;
  if (n < N) {
    const float b = a.bias ? act_to_f32<BF16>(a.bias[n]) : 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int m = (e & 3) + 8 * (e >> 2) + 4 * h;
      if (m < M) {
        float v = acc[e];
        if (a.bias) v = act_to_f32<BF16>(f32_to_act<BF16>(v)) + b;
        a.y[static_cast<size_t>(m) * N + n] = f32_to_act<BF16>(v);
      }
    }
  }
}

template <int NWV>
hipError_t launch_dense_batch_w(const DenseArgs2& a, bool bf16, hipStream_t s) {
  constexpr int LDSB = skinny_lds(NWV);
  static DeviceOnce once;
  if (once.pending()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_batch_kernel<true, NWV>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSB);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_batch_kernel<false, NWV>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSB);
    if (e != hipSuccess) return e;
    once.done();
  }
  dim3 grid((a.N + 32 * NWV - 1) / (32 * NWV)), block(64 * NWV);
  if (bf16) hipLaunchKernelGGL((dense_batch_kernel<true, NWV>), grid, block, LDSB, s, a);
  else hipLaunchKernelGGL((dense_batch_kernel<false, NWV>), grid, block, LDSB, s, a);
  return hipGetLastError();
}
hipError_t launch_dense_batch(const DenseArgs2& a, bool bf16, hipStream_t s) {
  return launch_dense_batch_w<8>(a, bf16, s);
}

}  // namespace

// K % 64 == 0; x and w 16-byte aligned rows (K % 8 == 0).  Byte offsets inside the kernel are 32-bit: M * K and N * K * 2 < 2^31.
bool gemm_dense_supported(int M, int N, int K) {
  return K % DK == 0 && K >= DK && static_cast<long long>(M) * K * 2 < (1LL << 31) && static_cast<long long>(N) * K * 2 < (1LL << 31);
}

// k slices for a shape whose 256 x 256 tiles leave most CUs idle: as many as still give every slice >= 8 K tiles and at most one
// round of workgroups (1: no split)
int gemm_dense_ksplit(int M, int N, int K) {
  const int tiles = ((N + DT - 1) / DT) * ((M + DT - 1) / DT), nk = K / DK;
  if (tiles >= 176) return 1;
  int best = 1;
  for (int sp = 2; sp <= 8; ++sp)
    if (nk % sp == 0 && nk / sp >= 8 && tiles * sp <= 256) best = sp;
  return best;
}

size_t gemm_dense_workspace_bytes(int M, int N, int K) {
  const int sp = gemm_dense_ksplit(M, N, K);
  return sp > 1 ? static_cast<size_t>(sp) * M * N * sizeof(float) : 0;
}

hipError_t launch_gemm_dense(const void* x, const void* w, const void* bias, void* y, int M, int N, int K, bool bf16, hipStream_t s,
                             bool swiglu, float* part) {
  if (swiglu && (N % 256 || bias)) return hipErrorInvalidValue;
  static const bool skinny = env_switch("GBX_DENSE_SKINNY") != 0;
  if (skinny && !swiglu && M <= 32 && K % SK == 0 && N >= 128 * 256) {        // a stream of weight rows: one pass, no padding rows
    DenseArgs2 a{static_cast<const unsigned short*>(x), static_cast<const unsigned short*>(w), static_cast<const unsigned short*>(bias),
                 static_cast<unsigned short*>(y), M, N, K, 0, nullptr, 1};
    return launch_dense_batch(a, bf16, s);
  }
  const int ksplit = (part && !swiglu && N % 4 == 0) ? gemm_dense_ksplit(M, N, K) : 1;
  static DeviceOnce once;
  if (once.pending()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_dense_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, DENSE_LDS);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_dense_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, DENSE_LDS);
    if (e != hipSuccess) return e;
    once.done();
  }
  DenseArgs2 a{static_cast<const unsigned short*>(x), static_cast<const unsigned short*>(w), static_cast<const unsigned short*>(bias),
               static_cast<unsigned short*>(y), M, N, K, swiglu ? 1 : 0, ksplit > 1 ? part : nullptr, ksplit};
  const int tiles = ((N + DT - 1) / DT) * ((M + DT - 1) / DT) * ksplit;
  dim3 grid(8 * ((tiles + 7) / 8)), block(512);
  if (bf16) hipLaunchKernelGGL((gemm_dense_kernel<true>), grid, block, DENSE_LDS, s, a);
  else hipLaunchKernelGGL((gemm_dense_kernel<false>), grid, block, DENSE_LDS, s, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || ksplit <= 1) return e;
  return launch_splitk_finish(part, static_cast<const unsigned short*>(bias), static_cast<unsigned short*>(y), ksplit, M, N, bf16, s);
}

}  // namespace gbx
