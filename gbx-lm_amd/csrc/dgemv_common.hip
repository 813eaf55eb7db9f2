// Host-side geometry of the decode-block GEMV (dgemv_impl.h) and the load-time packing of its (scale | zero) companion.
#include "dgemv_impl.h"
#include "gbx_env.h"
#include <cstdlib>

namespace gbx {


bool dg_late_default() {
  static const bool v = env_switch("GBX_DG_LATE") != 0;
  return v;
}

bool dg_plan(int bits, int N, int K, int group, int epi, bool norm, int rows_per_batch, DgPlan* pl) {
  if (bits < 2 || bits > 4 || N <= 0 || K <= 0 || (K % 128) != 0) return false;
  const int vpl = bits == 2 ? 64 : 32;
  if (group != 32 && group != 64 && group != 128) return false;
  if (group < vpl) return false;                     // one (scale, zero) per lane and pass (2-bit group 32: gemv_kernel's SPLIT form)
  const int nslots = K / vpl;
  int P, KS;
  if (nslots <= 64) { P = 1; KS = 1; }
  else if (nslots <= 128) { P = 2; KS = 1; }
  else if (nslots <= 256) { P = 1; KS = 4; }
  else if (nslots <= 512) { P = 2; KS = 4; }
  else return false;
  if (epi == 2 && KS != 1) return false;
  if (static_cast<long long>(N) * (K / 32 * bits) * 4 >= 0x7FFFFFFFll) return false;   // row offsets are 32-bit scalars
  // rows per batch (measured, scratch/dg_sweep.sh): 4 for the plain / residual launches without a norm whose rows are one lane
  // pass long (o_proj: 3.6 vs 4.2 us), 2 everywhere else (q|k|v 4.7 vs 5.0, gate|up 9.9 vs 10.6, down 6.6 vs 7.5); two passes
  // per row keep two-row batches anyway (register budget of 16-wave workgroups)
  static const int force_r = env_switch("GBX_DG_R");
  int R = rows_per_batch ? rows_per_batch : (force_r ? force_r : ((P == 1 && KS == 1 && epi != 2 && !norm) ? 4 : 2));
  if (R != 2 && R != 4) return false;
  if (P == 2 && R == 4) { if (rows_per_batch) return false; R = 2; }
  auto divides = [&](int r) { return N % (epi == 2 ? r / 2 : r) == 0; };
  if (!divides(R)) { if (rows_per_batch || !divides(2)) return false; R = 2; }
  const int RO = epi == 2 ? R / 2 : R;
  const int nbatch = N / RO;
  constexpr int CUS = 256;
  static const int min_nw = env_switch("GBX_DG_MIN_NW");
  int NW, grid, bpo, apw;
  // two-pass rows (3 / 4 bits at K = 4096: twice the bytes per row): GBX_DG_P2_NW workgroup width of the KS == 1 launches
  static const int p2_nw = env_switch("GBX_DG_P2_NW");
  if (KS == 1) {
    if (nbatch <= CUS * 16) {
      bpo = 1;
      apw = (nbatch + CUS - 1) / CUS;
      NW = apw <= 4 ? 4 : (apw <= 8 ? 8 : 16);
      if (NW < min_nw) NW = min_nw <= 8 ? (min_nw <= 4 ? 4 : 8) : 16;
      if (P == 2 && p2_nw && p2_nw < NW) { NW = p2_nw; apw = NW; }          // several narrower workgroups per CU
      grid = (nbatch + apw - 1) / apw;
    } else {
      static const int mr_nw = env_switch("GBX_DG_MR_NW");
      NW = (P == 2 && p2_nw) ? p2_nw : ((P == 1 && (mr_nw == 8 || mr_nw == 4)) ? mr_nw : 16);
      const int wgs = CUS * (16 / NW);                                       // 16 resident waves per CU either way
      const int slots = CUS * 16;
      bpo = (nbatch + slots - 1) / slots;
      const int rem = nbatch - (bpo - 1) * slots;
      apw = (rem + wgs - 1) / wgs;
      grid = wgs;
    }
  } else {
    // row groups of four waves; 16-wave workgroups (4 owners) unless the launch is small
    if (nbatch <= CUS * 4) {
      bpo = 1;
      apw = (nbatch + CUS - 1) / CUS;
      NW = apw <= 2 ? 8 : 16;
      if (min_nw > 8) NW = 16;
      grid = (nbatch + apw - 1) / apw;
    } else {
      NW = 16;
      const int slots = CUS * 4;
      bpo = (nbatch + slots - 1) / slots;
      const int rem = nbatch - (bpo - 1) * slots;
      apw = (rem + CUS - 1) / CUS;
      grid = CUS;
    }
  }
  size_t lds = (bits == 2 ? Lds<2>::bytes(K) : (bits == 3 ? Lds<3>::bytes(K) : Lds<4>::bytes(K))) + NW * sizeof(float) + 16;
  if (KS == 4) lds += static_cast<size_t>(NW / 4) * bpo * R * 4 * sizeof(float);
  if (KS == 1 && norm) lds += static_cast<size_t>(K) * 2;              // x' of an x_acc launch (dgemv_impl.h)
  if (lds > 160 * 1024) return false;
  // partial sums of y^2 the launch can leave behind: one per wave (KS 1), one per workgroup when its row threads fit one wave (KS 4)
  int ssq_n = KS == 1 ? grid * NW : ((NW / 4) * bpo * R <= 64 ? grid : 0);
  if (ssq_n > DG_SSQ_SLOTS) ssq_n = 0;
  // register buffers: one batch per owner -> 1; else 2, or 4 where that puts ALL of an owner's batches in flight (GBX_DG_DEPTH)
  static const int depth = env_switch("GBX_DG_DEPTH");
  int D = bpo == 1 ? 1 : 2;
  if ((depth == 4 || depth == 3) && bpo > 2 && bpo <= 4 && P == 1 && KS == 1 && NW == 16 && R == 2) D = depth;
  *pl = DgPlan{P, KS, NW, R, D, grid, bpo, apw, nbatch, ssq_n, lds};
  return true;
}

// out[b][g][r] = scale | zero << 16 of row (b, r) and group g.  Rows of batch b: b*R + r of the one matrix, or (two matrices,
// SwiGLU) r < R/2: row b*R/2 + r of the first (gate), else row b*R/2 + r - R/2 of the second (up).
__global__ __launch_bounds__(256) void dg_pack_kernel(const unsigned short* __restrict__ s0, const unsigned short* __restrict__ z0,
                                                      const unsigned short* __restrict__ s1, const unsigned short* __restrict__ z1,
                                                      uint32_t* __restrict__ out, int nbatch, int G, int R) {
  const size_t total = static_cast<size_t>(nbatch) * G * R;
  for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < total; i += static_cast<size_t>(gridDim.x) * blockDim.x) {
    const int r = static_cast<int>(i % R);
    const int g = static_cast<int>((i / R) % G);
    const size_t b = i / (static_cast<size_t>(R) * G);
    const unsigned short *s = s0, *z = z0;
    size_t row;
    if (s1) {
      const int h = R / 2;
      row = b * h + (r % h);
      if (r >= h) { s = s1; z = z1; }
    } else row = b * R + r;
    out[i] = static_cast<uint32_t>(s[row * G + g]) | (static_cast<uint32_t>(z[row * G + g]) << 16);
  }
}

hipError_t launch_dg_pack(const void* s0, const void* z0, const void* s1, const void* z1, uint32_t* out, int N, int G, int R, hipStream_t s) {
  const int nbatch = s1 ? N / (R / 2) : N / R;
  const size_t total = static_cast<size_t>(nbatch) * G * R;
  const int blocks = static_cast<int>(std::min<size_t>((total + 255) / 256, 4096));
  hipLaunchKernelGGL(dg_pack_kernel, dim3(blocks), dim3(256), 0, s, static_cast<const unsigned short*>(s0), static_cast<const unsigned short*>(z0),
                     static_cast<const unsigned short*>(s1), static_cast<const unsigned short*>(z1), out, nbatch, G, R);
  return hipGetLastError();
}

}  // namespace gbx
