// Issue rate of single VALU instructions, via inline asm (8 independent destinations, 2 waves/SIMD, 1 WG of 256 thr x2 per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#define BODY(ASM) \
  for (int it = 0; it < iters; ++it) { \
    _Pragma("unroll") for (int u = 0; u < 8; ++u) { \
      asm volatile(ASM : "+v"(d[0]) : "v"(a), "v"(b)); asm volatile(ASM : "+v"(d[1]) : "v"(a), "v"(b)); \
      asm volatile(ASM : "+v"(d[2]) : "v"(a), "v"(b)); asm volatile(ASM : "+v"(d[3]) : "v"(a), "v"(b)); \
      asm volatile(ASM : "+v"(d[4]) : "v"(a), "v"(b)); asm volatile(ASM : "+v"(d[5]) : "v"(a), "v"(b)); \
      asm volatile(ASM : "+v"(d[6]) : "v"(a), "v"(b)); asm volatile(ASM : "+v"(d[7]) : "v"(a), "v"(b)); } }
template <int MODE>
__global__ __launch_bounds__(256) void k(const unsigned* in, unsigned* out, int iters) {
  unsigned d[8]; unsigned a = in[threadIdx.x], b = in[threadIdx.x + 256];
  for (int c = 0; c < 8; ++c) d[c] = in[threadIdx.x + c];
  if (MODE == 0) BODY("v_cvt_f32_ubyte0 %0, %1")
  else if (MODE == 1) BODY("v_cvt_f32_ubyte2 %0, %1")
  else if (MODE == 2) BODY("v_cvt_f32_u32 %0, %1")
  else if (MODE == 3) BODY("v_cvt_pk_bf16_f32 %0, %1, %2")
  else if (MODE == 4) BODY("v_and_or_b32 %0, %1, %2, %0")
  else if (MODE == 5) BODY("v_bfe_u32 %0, %1, 4, 4")
  else if (MODE == 6) BODY("v_perm_b32 %0, %1, %2, %0")
  else if (MODE == 7) BODY("v_fma_f32 %0, %1, %2, %0")
  else if (MODE == 8) BODY("v_and_b32 %0, %1, %2")
  else if (MODE == 9) BODY("v_lshl_or_b32 %0, %1, 19, %2")
  else if (MODE == 10) BODY("v_cvt_f32_f16 %0, %1")
  else if (MODE == 11) BODY("v_lshrrev_b32 %0, 4, %1")
  else if (MODE == 12) BODY("v_cvt_pk_f16_f32 %0, %1, %2")
  unsigned s = 0;
  for (int c = 0; c < 8; ++c) s += d[c];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, const unsigned* in, unsigned* out) {
  const int iters = 2000, grid = 256 * 2;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<MODE><<<grid, 256>>>(in, out, iters); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); k<MODE><<<grid, 256>>>(in, out, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double cyc = ms * 1e-3 * 2.4e9 / (2.0 * iters * 64);      // per SIMD: 2 waves x iters*64 instr
  printf("%-22s %6.2f cycles per wave-instruction (2 waves/SIMD, at 2.4 GHz)\n", name, cyc);
}
// 64-bit operand variants
template <int MODE>
__global__ __launch_bounds__(256) void k2(const unsigned* in, unsigned* out, int iters) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 d[8]; f2 a{1.0001f, 0.9999f}, b{0.5f, 0.25f};
  for (int c = 0; c < 8; ++c) d[c] = f2{(float)in[threadIdx.x + c], 1.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int c = 0; c < 8; ++c) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(d[c]) : "v"(a), "v"(b));
    }
  }
  float s = 0; for (int c = 0; c < 8; ++c) s += d[c].x + d[c].y;
  out[blockIdx.x * 256 + threadIdx.x] = __float_as_uint(s);
}
int main() {
  unsigned *in, *out; (void)hipMalloc(&in, 8192); (void)hipMemset(in, 0x11, 8192); (void)hipMalloc(&out, 512 * 256 * 4);
  run<0>("v_cvt_f32_ubyte0", in, out); run<1>("v_cvt_f32_ubyte2", in, out); run<2>("v_cvt_f32_u32", in, out);
  run<3>("v_cvt_pk_bf16_f32", in, out); run<12>("v_cvt_pk_f16_f32", in, out); run<10>("v_cvt_f32_f16", in, out);
  run<4>("v_and_or_b32", in, out); run<5>("v_bfe_u32", in, out);
  run<6>("v_perm_b32", in, out); run<7>("v_fma_f32", in, out); run<8>("v_and_b32", in, out); run<9>("v_lshl_or_b32", in, out);
  run<11>("v_lshrrev_b32", in, out);
  {
    const int iters = 2000, grid = 512; hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k2<0><<<grid, 256>>>(in, out, iters); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); k2<0><<<grid, 256>>>(in, out, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-22s %6.2f cycles per wave-instruction\n", "v_pk_fma_f32", ms * 1e-3 * 2.4e9 / (2.0 * iters * 64));
  }
  return 0;
}
