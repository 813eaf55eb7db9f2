// Host-side launcher declarations shared between the kernel TUs and the C-ABI (capi.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>

namespace gbx {

struct GemvArgs;

template <int BITS>
hipError_t launch_gemv(const GemvArgs& a, int M, bool bf16, hipStream_t stream);

hipError_t launch_unpack(const uint32_t* qw, int32_t* q, int N, int K, int bits, hipStream_t s);
hipError_t launch_dequant(const uint32_t* qw, const void* sc, const void* ze, void* w, int N, int K, int bits,
                          int log2g, bool sz_bf16, int out, hipStream_t s);
hipError_t launch_qlinear_generic(const void* x, const uint32_t* qw, const void* sc, const void* ze, const void* bias,
                                  void* y, int M, int N, int K, int bits, int log2g, bool bf16, hipStream_t s);

}  // namespace gbx
