// Issue rate of the VALU ops a dequantise step can be built from (8 independent chains, 2 waves/SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __bf16 b2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(const unsigned* in, unsigned* out, int iters) {
  unsigned w[8]; float f[8]; f2 p[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) { w[c] = in[threadIdx.x + c]; f[c] = __uint_as_float(w[c] & 0x3FFFFFFF); p[c] = f2{f[c], f[c] + 1.f}; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        if (MODE == 0) f[c] += static_cast<float>(w[c] & 0xFFu);                          // v_cvt_f32_ubyte0 + add
        else if (MODE == 1) { f2 v{f[c], p[c].x}; w[c] += __builtin_bit_cast(unsigned, __builtin_convertvector(v, b2)); }  // cvt_pk_bf16 + add
        else if (MODE == 2) p[c] = __builtin_elementwise_fma(p[c], f2{1.0001f, 0.9999f}, f2{f[c], f[c]});   // v_pk_fma_f32
        else if (MODE == 3) w[c] = ((w[c] >> 3) & 0x00780000u) | 0x3F800000u;            // shift + and_or
        else if (MODE == 4) w[c] = __builtin_amdgcn_ubfe(w[c], 5, 4) + w[c];             // bfe + add
        else if (MODE == 5) w[c] = __builtin_amdgcn_perm(w[c], w[(c + 1) & 7], 0x07060302u);   // v_perm_b32
        else if (MODE == 6) f[c] = fmaf(f[c], 1.0001f, 0.5f);                             // v_fma_f32
        else if (MODE == 7) f[c] += static_cast<float>(w[c]);                             // v_cvt_f32_u32 + add
        else if (MODE == 8) w[c] = (w[c] & 0x0F0F0F0Fu) + 1u;                             // and + add
      }
    }
  }
  unsigned s = 0;
#pragma unroll
  for (int c = 0; c < 8; ++c) s += w[c] + __float_as_uint(f[c]) + __float_as_uint(p[c].x + p[c].y);
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, int nops, const unsigned* in, unsigned* out) {
  const int iters = 2000, grid = 256 * 2;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<MODE><<<grid, 256>>>(in, out, iters); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); k<MODE><<<grid, 256>>>(in, out, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  // per SIMD: 2 waves, each iters*4*8 groups of nops instructions
  const double cyc = ms * 1e-3 * 2.4e9 / (2.0 * iters * 32);
  printf("%-34s %6.2f cycles per group of %d instr per wave  (%.2f per instr)\n", name, cyc, nops, cyc / nops);
}
int main() {
  unsigned *in, *out; (void)hipMalloc(&in, 8192); (void)hipMemset(in, 0x11, 8192); (void)hipMalloc(&out, 512 * 256 * 4);
  run<0>("cvt_f32_ubyte0 + add_f32", 2, in, out);
  run<7>("cvt_f32_u32 + add_f32", 2, in, out);
  run<1>("cvt_pk_bf16_f32 + add_u32", 2, in, out);
  run<2>("pk_fma_f32", 1, in, out);
  run<3>("lshr + and_or", 2, in, out);
  run<4>("bfe + add_u32", 2, in, out);
  run<5>("perm_b32", 1, in, out);
  run<6>("fma_f32", 1, in, out);
  run<8>("and + add_u32", 2, in, out);
  return 0;
}
