// Probe 2: full k-slot map of v_mfma_scale_f32_16x16x128_f8f6f4, A = fp8 e4m3 (8 dwords/lane), B = fp4 (which dwords/nibbles are used).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void k_mfma(const unsigned* A, const unsigned* B, float* C) {
  const int l = threadIdx.x;
  v8i a, b;
  for (int i = 0; i < 8; ++i) { a[i] = A[l * 8 + i]; b[i] = B[l * 8 + i]; }
  v4f c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 4, 0, 127, 0, 127);
  for (int i = 0; i < 4; ++i) C[l * 4 + i] = c[i];
}
static float e4m3(unsigned char c) { int s = c >> 7, e = (c >> 3) & 15, m = c & 7; float v = e ? ldexpf(1.f + m / 8.f, e - 7) : ldexpf(m / 8.f, -6); return s ? -v : v; }
int main() {
  unsigned hA[64 * 8], hB[64 * 8]; float hC[64 * 4];
  unsigned *dA, *dB; float* dC; hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC);
  // A row 0 = lanes 0, 16, 32, 48; slot s = group * 32 + byte -> code: s < 126 ? s + 1 : 0x80 | (s - 125)
  memset(hA, 0, sizeof hA);
  unsigned char code[128];
  for (int s = 0; s < 128; ++s) { code[s] = s < 126 ? s + 1 : (0x80 | (s - 125)); const int g = s / 32, b = s % 32; hA[(16 * g) * 8 + b / 4] |= (unsigned)code[s] << (8 * (b % 4)); }
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice);
  for (int g = 0; g < 4; ++g) {
    printf("B lane %2d (col 0, lane group %d): nibble (dword, idx) -> A slot (group, byte)\n", 16 * g, g);
    for (int v = 0; v < 8; ++v) {
      printf("  dword %d:", v);
      for (int i = 0; i < 8; ++i) {
        memset(hB, 0, sizeof hB); hB[(16 * g) * 8 + v] = 0x2u << (4 * i);
        hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, dA, dB, dC); hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
        const float got = hC[0]; int slot = -1;
        for (int s = 0; s < 128; ++s) if (e4m3(code[s]) == got) slot = s;
        if (got == 0.f) printf("   --  "); else printf(" %d:%-3d", slot / 32, slot % 32);
      }
      printf("\n");
    }
  }
  return 0;
}
