// extern "C" entry points declared in include/gbx_qlinear.h: argument validation (the reference's own
// asserts, gbx_lm/models/quantized_linear_gba.py:61-62,272 and gbx_lm/utils.py:819-821), kernel dispatch,
// thread-local error text.  No allocation, no synchronisation: every call is graph-capturable.
#include "gbx_qlinear.h"
#include "gemv_impl.h"
#include "kernels.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int hip_fail(hipError_t e, const char* what) {
  return fail(GBX_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}

int log2_group(int g) { return g == 32 ? 5 : g == 64 ? 6 : g == 128 ? 7 : -1; }

bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

int check_shape(int64_t M, int32_t N, int32_t K, int32_t bits, int32_t group) {
  if (bits != 2 && bits != 3 && bits != 4) return fail(GBX_ERR_INVALID, "bits must be 2, 3 or 4 (got %d)", bits);
  if (log2_group(group) < 0)
    return fail(GBX_ERR_INVALID, "The group size value (%d) must be 32, 64 or 128.", group);
  if (N <= 0 || K <= 0 || M < 0) return fail(GBX_ERR_INVALID, "non-positive dimension M=%lld N=%d K=%d", (long long)M, N, K);
  if (K % 32) return fail(GBX_ERR_INVALID, "input_dims (%d) must be a multiple of 32", K);
  if (K % group) return fail(GBX_ERR_INVALID, "input_dims (%d) must be a multiple of group_size (%d)", K, group);
  if (M > 65535) return fail(GBX_ERR_INVALID, "M=%lld exceeds the launch limit of this entry point", (long long)M);
  return GBX_OK;
}

}  // namespace

extern "C" {

int gbx_version(void) { return 1000; }

const char* gbx_last_error(void) { return g_err; }

int gbx_qlinear_check(int64_t M, int32_t N, int32_t K, int32_t bits, int32_t group_size) {
  return check_shape(M, N, K, bits, group_size);
}

size_t gbx_qlinear_workspace_bytes(int64_t, int32_t, int32_t, int32_t, int32_t) { return 0; }

int gbx_qlinear_forward(const void* x, const uint32_t* qweight, const void* scales, const void* zeros,
                        const void* bias, void* y, int64_t M, int32_t N, int32_t K, int32_t bits,
                        int32_t group_size, int32_t act_dtype, void* hip_stream) {
  int rc = check_shape(M, N, K, bits, group_size);
  if (rc) return rc;
  if (act_dtype != GBX_F16 && act_dtype != GBX_BF16) return fail(GBX_ERR_INVALID, "act_dtype must be GBX_F16 or GBX_BF16");
  if (M == 0) return GBX_OK;   /* empty batch: nothing to enqueue (x / y may legitimately be null) */
  if (!x || !qweight || !scales || !zeros || !y) return fail(GBX_ERR_INVALID, "null device pointer");
  if (!aligned(qweight, 16) || !aligned(x, 16))
    return fail(GBX_ERR_INVALID, "x and qweight must be 16-byte aligned");
  if (!aligned(scales, 2) || !aligned(zeros, 2) || !aligned(y, 2) || (bias && !aligned(bias, 2)))
    return fail(GBX_ERR_INVALID, "scales/zeros/bias/y must be 2-byte aligned");
  hipStream_t stream = static_cast<hipStream_t>(hip_stream);
  const bool bf16 = act_dtype == GBX_BF16;
  const int log2g = log2_group(group_size);

  static const bool force_generic = getenv("GBX_FORCE_GENERIC") != nullptr;
  const int max_k = bits == 2 ? gbx::gemv_max_k<2>() : bits == 3 ? gbx::gemv_max_k<3>() : gbx::gemv_max_k<4>();
  hipError_t e;
  if (!force_generic && (K % 128) == 0 && K <= max_k) {
    gbx::GemvArgs a{x, qweight, scales, zeros, bias, y, N, K, log2g, 0};
    switch (bits) {
      case 2: e = gbx::launch_gemv<2>(a, static_cast<int>(M), bf16, stream); break;
      case 3: e = gbx::launch_gemv<3>(a, static_cast<int>(M), bf16, stream); break;
      default: e = gbx::launch_gemv<4>(a, static_cast<int>(M), bf16, stream); break;
    }
  } else {
    e = gbx::launch_qlinear_generic(x, qweight, scales, zeros, bias, y, static_cast<int>(M), N, K, bits, log2g, bf16, stream);
  }
  if (e != hipSuccess) return hip_fail(e, "gbx_qlinear_forward launch");
  return GBX_OK;
}

int gbx_unpack(const uint32_t* qweight, int32_t* q, int32_t N, int32_t K, int32_t bits, void* hip_stream) {
  int rc = check_shape(1, N, K, bits, 32);
  if (rc) return rc;
  if (!qweight || !q) return fail(GBX_ERR_INVALID, "null device pointer");
  hipError_t e = gbx::launch_unpack(qweight, q, N, K, bits, static_cast<hipStream_t>(hip_stream));
  if (e != hipSuccess) return hip_fail(e, "gbx_unpack launch");
  return GBX_OK;
}

int gbx_dequantize(const uint32_t* qweight, const void* scales, const void* zeros, void* w, int32_t N, int32_t K,
                   int32_t bits, int32_t group_size, int32_t sz_dtype, int32_t out_dtype, void* hip_stream) {
  int rc = check_shape(1, N, K, bits, group_size);
  if (rc) return rc;
  if (!qweight || !scales || !zeros || !w) return fail(GBX_ERR_INVALID, "null device pointer");
  if (sz_dtype != GBX_F16 && sz_dtype != GBX_BF16) return fail(GBX_ERR_INVALID, "sz_dtype must be GBX_F16 or GBX_BF16");
  if (out_dtype < 0 || out_dtype > GBX_F32) return fail(GBX_ERR_INVALID, "out_dtype must be GBX_F16, GBX_BF16 or GBX_F32");
  hipError_t e = gbx::launch_dequant(qweight, scales, zeros, w, N, K, bits, log2_group(group_size), sz_dtype == GBX_BF16,
                                     out_dtype, static_cast<hipStream_t>(hip_stream));
  if (e != hipSuccess) return hip_fail(e, "gbx_dequantize launch");
  return GBX_OK;
}

}  // extern "C"
