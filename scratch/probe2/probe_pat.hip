// Probe: (1) HBM streaming rate of three lane->address patterns; (2) i8 MFMA positional layout + issue rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7FFFFFFF, 0x00020000);
}
// rows of ROWB bytes; tile = 16 rows. PAT 0: wave reads whole rows (64 lanes x 16 B = 1 KB), rows interleaved over waves.
// PAT 1: wave = K-slice of 64 B over 16 rows (16 waves cover the 1 KB row).  PAT 2: wave = 256-B slice (4 loads), 4 waves/row tile.
template <int PAT, int DEPTH>
__global__ __launch_bounds__(1024) void stream_kernel(const uint32_t* __restrict__ w, uint32_t* __restrict__ out, int nrows, int rowb) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const __amdgpu_buffer_rsrc_t d = rsrc(w);
  u32x4 acc = {0, 0, 0, 0};
  if (PAT == 0) {
    const int owner = blockIdx.x * 16 + wave, owners = gridDim.x * 16;
    const int voff = lane * 16;
    const int per = nrows / owners;
    for (int j = 0; j < per; j += DEPTH) {
      u32x4 v[DEPTH];
#pragma unroll
      for (int t = 0; t < DEPTH; ++t) v[t] = __builtin_amdgcn_raw_buffer_load_b128(d, voff, (owner + (j + t) * owners) * rowb, 2);
#pragma unroll
      for (int t = 0; t < DEPTH; ++t) acc ^= v[t];
    }
  } else if (PAT == 1) {
    const int ntiles = nrows / 16, per = ntiles / gridDim.x;
    const int voff = (lane & 15) * rowb + wave * 64 + (lane >> 4) * 16;
    for (int j = 0; j < per; j += DEPTH) {
      u32x4 v[DEPTH];
#pragma unroll
      for (int t = 0; t < DEPTH; ++t) v[t] = __builtin_amdgcn_raw_buffer_load_b128(d, voff, (blockIdx.x + (j + t) * gridDim.x) * 16 * rowb, 2);
#pragma unroll
      for (int t = 0; t < DEPTH; ++t) acc ^= v[t];
    }
  } else {
    const int rg = wave >> 2, kq = wave & 3;
    const int ntiles = nrows / 16, owners = gridDim.x * 4, owner = blockIdx.x * 4 + rg, per = ntiles / owners;
    const int voff = (lane & 15) * rowb + kq * 256 + (lane >> 4) * 16;
    for (int j = 0; j < per; j += DEPTH / 4) {
      u32x4 v[DEPTH];
#pragma unroll
      for (int t = 0; t < DEPTH; ++t) v[t] = __builtin_amdgcn_raw_buffer_load_b128(d, voff + (t & 3) * 64, (owner + (j + t / 4) * owners) * 16 * rowb, 2);
#pragma unroll
      for (int t = 0; t < DEPTH; ++t) acc ^= v[t];
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[threadIdx.x] = acc.x;
}

// positional check: D[r][c] = sum over (kb, j) of A_lane(r,kb).byte[j] * B_lane(c,kb).byte[j] ?
__global__ void mfma_layout_kernel(const int8_t* a /*[16][4][16]*/, const int8_t* b /*[16][4][16]*/, int* dout /*[16][16]*/) {
  const int lane = threadIdx.x, r = lane & 15, kb = lane >> 4;
  const i32x4 av = *reinterpret_cast<const i32x4*>(a + (r * 4 + kb) * 16);
  const i32x4 bv = *reinterpret_cast<const i32x4*>(b + (r * 4 + kb) * 16);
  i32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, bv, c, 0, 0, 0);
  for (int i = 0; i < 4; ++i) dout[(kb * 4 + i) * 16 + r] = c[i];   // row = (lane>>4)*4+i, col = lane&15
}

// issue-rate probe: per iteration UNP valu ops + 1 mfma, 'waves' waves per SIMD
template <int NV>
__global__ __launch_bounds__(1024) void mfma_rate_kernel(uint32_t* out, int iters, uint32_t seed) {
  uint32_t w = seed + threadIdx.x;
  i32x4 bv = {1, 2, 3, 4};
  i32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    i32x4 a0, a1;
    a0[0] = w & 0x03030303u; a0[1] = w & 0x0C0C0C0Cu; uint32_t t = w >> 4; a0[2] = t & 0x03030303u; a0[3] = t & 0x0C0C0C0Cu;
    uint32_t w2 = w * 3 + it;
    a1[0] = w2 & 0x03030303u; a1[1] = w2 & 0x0C0C0C0Cu; t = w2 >> 4; a1[2] = t & 0x03030303u; a1[3] = t & 0x0C0C0C0Cu;
    if (NV > 0) { c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, bv, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, bv, c1, 0, 0, 0); }
    else { c0 += a0; c1 += a1; }
    w = w2 ^ 0x9E3779B9u;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c0[1] + c0[2] + c0[3] + c1[0] + c1[1] + c1[2] + c1[3];
}

int main() {
  const int rowb = 1024;
  const size_t nrows = 16 * 256 * 36;   // 147456 rows x 1 KB = 151 MB
  uint32_t *w, *out;
  CK(hipMalloc(&w, nrows * rowb * 2)); CK(hipMalloc(&out, 1 << 22));
  CK(hipMemset(w, 1, nrows * rowb * 2));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](auto launch, const char* name, double bytes) {
    for (int i = 0; i < 3; ++i) launch(i & 1);
    hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) launch(i & 1);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %8.2f us  %8.1f GB/s\n", name, ms * 1e3 / reps, bytes * reps / ms / 1e6);
  };
  const double bytes = double(nrows) * rowb;
  auto base = [&](int alt) { return w + (alt ? nrows * rowb / 4 : 0); };
  timeit([&](int alt) { hipLaunchKernelGGL((stream_kernel<0, 4>), dim3(256), dim3(1024), 0, 0, base(alt), out, (int)nrows, rowb); }, "PAT0 rows 1KB/wave-load depth4", bytes);
  timeit([&](int alt) { hipLaunchKernelGGL((stream_kernel<0, 9>), dim3(256), dim3(1024), 0, 0, base(alt), out, (int)nrows, rowb); }, "PAT0 depth9", bytes);
  timeit([&](int alt) { hipLaunchKernelGGL((stream_kernel<1, 4>), dim3(256), dim3(1024), 0, 0, base(alt), out, (int)nrows, rowb); }, "PAT1 16rows x 64B depth4", bytes);
  timeit([&](int alt) { hipLaunchKernelGGL((stream_kernel<1, 9>), dim3(256), dim3(1024), 0, 0, base(alt), out, (int)nrows, rowb); }, "PAT1 depth9", bytes);
  timeit([&](int alt) { hipLaunchKernelGGL((stream_kernel<2, 4>), dim3(256), dim3(1024), 0, 0, base(alt), out, (int)nrows, rowb); }, "PAT2 16rows x 256B(4 loads) depth4", bytes);
  timeit([&](int alt) { hipLaunchKernelGGL((stream_kernel<2, 12>), dim3(256), dim3(1024), 0, 0, base(alt), out, (int)nrows, rowb); }, "PAT2 depth12", bytes);
  // small: 32768 rows (33 MB) like gate/up
  const int nsm = 28672;
  timeit([&](int alt) { hipLaunchKernelGGL((stream_kernel<0, 7>), dim3(256), dim3(1024), 0, 0, base(alt) + (alt ? 0 : nrows * rowb / 8), out, nsm, rowb); }, "PAT0 28672 rows depth7 (1 round)", double(nsm) * rowb);
  timeit([&](int alt) { hipLaunchKernelGGL((stream_kernel<1, 7>), dim3(256), dim3(1024), 0, 0, base(alt) + (alt ? 0 : nrows * rowb / 8), out, nsm, rowb); }, "PAT1 28672 rows depth7 (all in flight)", double(nsm) * rowb);

  // layout
  std::vector<int8_t> ha(1024), hb(1024); std::vector<int> hd(256), ref(256, 0);
  for (int i = 0; i < 1024; ++i) { ha[i] = (i * 7 + 3) % 13; hb[i] = ((i * 5 + 1) % 255) - 127; }
  for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c) { int s = 0; for (int kb = 0; kb < 4; ++kb) for (int j = 0; j < 16; ++j) s += int(ha[(r * 4 + kb) * 16 + j]) * int(hb[(c * 4 + kb) * 16 + j]); ref[r * 16 + c] = s; }
  int8_t *da, *db; int* dd; CK(hipMalloc(&da, 1024)); CK(hipMalloc(&db, 1024)); CK(hipMalloc(&dd, 1024));
  CK(hipMemcpy(da, ha.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), 1024, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(mfma_layout_kernel, dim3(1), dim3(64), 0, 0, da, db, dd);
  CK(hipMemcpy(hd.data(), dd, 1024, hipMemcpyDeviceToHost));
  int bad = 0; for (int i = 0; i < 256; ++i) bad += hd[i] != ref[i];
  printf("i8 mfma positional layout: %d mismatches of 256 (d[0]=%d ref %d, d[17]=%d ref %d)\n", bad, hd[0], ref[0], hd[17], ref[17]);

  // rate
  for (int wpb : {256, 512, 1024}) {
    const int iters = 4096;
    for (int nv = 1; nv >= 0; --nv) {
      for (int i = 0; i < 2; ++i) { if (nv) hipLaunchKernelGGL((mfma_rate_kernel<1>), dim3(256), dim3(wpb), 0, 0, out, iters, 1u); else hipLaunchKernelGGL((mfma_rate_kernel<0>), dim3(256), dim3(wpb), 0, 0, out, iters, 1u); }
      hipEventRecord(e0);
      if (nv) hipLaunchKernelGGL((mfma_rate_kernel<1>), dim3(256), dim3(wpb), 0, 0, out, iters, 1u); else hipLaunchKernelGGL((mfma_rate_kernel<0>), dim3(256), dim3(wpb), 0, 0, out, iters, 1u);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double wps = wpb / 256.0;   // waves per SIMD
      printf("rate %s: %d waves/SIMD: %.1f ns per (2 unpack + 2 mfma) iteration per wave -> %.1f cycles@2.4GHz per SIMD-iteration\n", nv ? "unpack+mfma" : "unpack only", (int)wps, ms * 1e6 / iters, ms * 1e6 / iters * 2.4 / wps);
    }
  }
  return 0;
}
