// Argument validation and thread-local error text shared by the C-ABI translation units (capi.hip, capi_exp.hip).  Everything is
// file-local (anonymous namespace): each shared library keeps its own error buffer.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#ifndef GBX_OK
#error "include gbx_qlinear.h first (error codes)"
#endif
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
  if (M > (1LL << 40)) return fail(GBX_ERR_INVALID, "M=%lld is not a plausible row count", (long long)M);
  return GBX_OK;
}

}  // namespace
