// extern "C" entry points declared in include/gbx_qlinear.h: argument validation (the reference's own
// asserts, gbx_lm/models/quantized_linear_gba.py:61-62,272 and gbx_lm/utils.py:819-821), kernel dispatch,
// thread-local error text.  No allocation, no synchronisation: every call is graph-capturable.
#include "gbx_qlinear.h"
#include "gbx_env.h"
#include "gemv_impl.h"
#include "dgemv_impl.h"
#include "dattn_impl.h"
#include "kernels.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "capi_util.h"

#ifdef GBX_STAMPS
namespace gbx {
static unsigned long long g_stamp_launch = 0;
unsigned long long* stamp_region() {
  const char* p = getenv("GBX_STAMP_PTR");
  if (!p) return nullptr;
  unsigned long long* base = reinterpret_cast<unsigned long long*>(strtoull(p, nullptr, 0));
  const char* n = getenv("GBX_STAMP_SLOTS");
  const unsigned long long slots = n ? strtoull(n, nullptr, 0) : 1;
  return base + (g_stamp_launch++ % slots) * STAMP_REGION_WORDS;
}
}  // namespace gbx
extern "C" void gbx_debug_stamp_reset(void) { gbx::g_stamp_launch = 0; }
#endif

extern "C" {

/* 1302: gbx_dmlp grows weight_layout; + gbx_repack_planes, gbx_dmlp_weight_layout (3-bit rows as a 2-bit and a 1-bit plane for the matrix cores).  1301: + gbx_logprobs_argmax_packed (an addition; gbx_logprobs_argmax is two launches instead of three).  1300: gbx_acc_residual grows act_dtype; + gbx_xcd_layout_check, gbx_dblock_*; a timed-out hand-off poisons its accumulator.  1202: + gbx_add_rmsnorm_slices_xcd, gbx_norm_state_bytes (additions).  1201: + gbx_dattn.x_acc / x_out / zero_buf, gbx_dmlp_*, gbx_acc_residual.  1200: gbx_dgemv grows (x_acc, x_out, zero_buf, zero_n) + gbx_dattn_* + gbx_dgemv_kernel_id.  1101: + gbx_attn_decode_slices (an addition: nothing else changed).  1100: gbx_dgemv_* entry points; gbx_qlinear_fused (x_acc, x_out, zero_buf, zero_n since 1001) -- a consumer built against
 * an older header must not pass its shorter struct to this library: check gbx_version() >= the header it compiled with. */
int gbx_version(void) { return 1302; }

const char* gbx_last_error(void) { return g_err; }

int gbx_qlinear_check(int64_t M, int32_t N, int32_t K, int32_t bits, int32_t group_size) {
  return check_shape(M, N, K, bits, group_size);
}

size_t gbx_qlinear_workspace_bytes(int64_t M, int32_t N, int32_t K, int32_t bits, int32_t group_size) {
  if (check_shape(M, N, K, bits, group_size) != GBX_OK || M < 2) return 0;
  if (M <= 32) return gbx::qgemm_batch_workspace_bytes(static_cast<int>(M), N, K, bits, log2_group(group_size));
  if (gbx::dense_path_wanted(M, N, K)) {            /* the weights, dequantised once per call (+ fp32 slices of a k-split dense GEMM) */
    const size_t img = (static_cast<size_t>(N) * K * 2 + 255) & ~static_cast<size_t>(255);
    const size_t part = M <= 61440 ? gbx::gemm_dense_workspace_bytes(static_cast<int>(M), N, K) : 0;
    return part ? img + part : static_cast<size_t>(N) * K * 2;
  }
  if (M > 64 && (K % 128) == 0) return gbx::qgemm_prefill_workspace_bytes(static_cast<int>(M), N, K);
  return 0;
}

int gbx_qlinear_kernel_id(int64_t M, int32_t N, int32_t K, int32_t bits, int32_t group_size, int32_t has_workspace) {
  if (check_shape(M, N, K, bits, group_size) != GBX_OK) return -1;
  static const bool force_generic = (gbx::env_switch("GBX_FORCE_GENERIC") != 0);
  const int max_k = bits == 2 ? gbx::gemv_max_k<2>() : bits == 3 ? gbx::gemv_max_k<3>() : gbx::gemv_max_k<4>();
  if (M == 0) return 0;
  if (!force_generic && M > 1 && M <= 64 && (K % 128) == 0 && K >= 512) return 20;               /* batched-decode kernels */
  if (!force_generic && gbx::dense_path_wanted(M, N, K) && has_workspace) return 30 + gbx::prefill_tile_id(static_cast<int>(M), N, K, false);
  if (!force_generic && M > 1 && (K % 128) == 0) return 10 + gbx::prefill_tile_id(static_cast<int>(M), N, K, has_workspace != 0);
  if (!force_generic && (K % 128) == 0 && K <= max_k) return 1;                                   /* decode GEMV */
  return 2;                                                                                       /* generic kernel */
}

int gbx_qlinear_forward(const void* x, const uint32_t* qweight, const void* scales, const void* zeros,
                        const void* bias, void* y, int64_t M, int32_t N, int32_t K, int32_t bits,
                        int32_t group_size, int32_t act_dtype, void* hip_stream) {
  return gbx_qlinear_forward_ws(x, qweight, scales, zeros, bias, y, M, N, K, bits, group_size, act_dtype, nullptr, 0, hip_stream);
}

int gbx_qlinear_forward_ws(const void* x, const uint32_t* qweight, const void* scales, const void* zeros,
                           const void* bias, void* y, int64_t M, int32_t N, int32_t K, int32_t bits,
                           int32_t group_size, int32_t act_dtype, void* workspace, size_t workspace_bytes, void* hip_stream) {
  int rc = check_shape(M, N, K, bits, group_size);
  if (rc) return rc;
  if (act_dtype != GBX_F16 && act_dtype != GBX_BF16) return fail(GBX_ERR_INVALID, "act_dtype must be GBX_F16 or GBX_BF16");
  if (M == 0) return GBX_OK;   /* empty batch: nothing to enqueue (x / y may legitimately be null) */
  if (!x || !qweight || !scales || !zeros || !y) return fail(GBX_ERR_INVALID, "null device pointer");
  if (!aligned(qweight, 16) || !aligned(x, 16))
    return fail(GBX_ERR_INVALID, "x and qweight must be 16-byte aligned");
  if (!aligned(scales, 2) || !aligned(zeros, 2) || !aligned(y, 2) || (bias && !aligned(bias, 2)))
    return fail(GBX_ERR_INVALID, "scales/zeros/bias/y must be 2-byte aligned");
  // The reference op has no row limit; a launch has (grid.y <= 65535 rows on the row-per-block kernels, 32-bit tile
  // arithmetic on the matrix-core ones): longer inputs go in slabs of 61440 rows (a multiple of every row tile).
  constexpr int64_t SLAB = 61440;
  if (M > SLAB) {
    if ((K % 8) != 0) return fail(GBX_ERR_UNSUPPORTED, "M > %lld needs K %% 8 == 0 (slab alignment)", (long long)SLAB);
    for (int64_t m0 = 0; m0 < M; m0 += SLAB) {
      const int64_t mm = M - m0 < SLAB ? M - m0 : SLAB;
      rc = gbx_qlinear_forward_ws(static_cast<const unsigned short*>(x) + m0 * K, qweight, scales, zeros, bias,
                                  static_cast<unsigned short*>(y) + m0 * N, mm, N, K, bits, group_size, act_dtype,
                                  workspace, workspace_bytes, hip_stream);
      if (rc) return rc;
    }
    return GBX_OK;
  }
  hipStream_t stream = static_cast<hipStream_t>(hip_stream);
  const bool bf16 = act_dtype == GBX_BF16;
  const int log2g = log2_group(group_size);

  static const bool force_generic = (gbx::env_switch("GBX_FORCE_GENERIC") != 0);
  const int max_k = bits == 2 ? gbx::gemv_max_k<2>() : bits == 3 ? gbx::gemv_max_k<3>() : gbx::gemv_max_k<4>();
  hipError_t e;
  // dense path: the image always fits when the caller sized the workspace with gbx_qlinear_workspace_bytes; a few-tile shape takes
  // it only together with the scratch of its k split (the un-split dense GEMM would leave most CUs idle)
  float* dense_part = nullptr;
  bool use_dense = !force_generic && gbx::dense_path_wanted(M, N, K) && workspace && aligned(workspace, 16) &&
                   workspace_bytes >= static_cast<size_t>(N) * K * 2;
  if (use_dense) {
    const size_t img = (static_cast<size_t>(N) * K * 2 + 255) & ~static_cast<size_t>(255);
    const size_t part = gbx::gemm_dense_workspace_bytes(static_cast<int>(M), N, K);
    if (part && workspace_bytes >= img + part) dense_part = reinterpret_cast<float*>(static_cast<char*>(workspace) + img);
    const long long tiles = ((M + 255) / 256) * ((N + 255) / 256);
    if (!dense_part && M < gbx::GBX_DENSE_MIN_M && tiles < 176) use_dense = false;
  }
  if (!force_generic && M > 1 && M <= 64 && (K % 128) == 0 && K >= 512) {
    // a sufficient, 16-byte aligned workspace lets the batched-decode kernel split k across workgroups
    float* ws = nullptr;
    const size_t need = M <= 32 ? gbx::qgemm_batch_workspace_bytes(static_cast<int>(M), N, K, bits, log2g) : 0;
    if (workspace && need && workspace_bytes >= need && aligned(workspace, 16)) ws = static_cast<float*>(workspace);
    e = gbx::launch_qgemm_skinny(x, qweight, scales, zeros, bias, y, static_cast<int>(M), N, K, bits, log2g, bf16, stream, ws);
  } else if (use_dense) {
    e = gbx::launch_qgemm_dense(x, qweight, scales, zeros, bias, y, static_cast<int>(M), N, K, bits, log2g, bf16, stream, workspace, dense_part);
  } else if (!force_generic && M > 1 && (K % 128) == 0) {
    float* ws = nullptr;                            // short prompts: k split across workgroups when the caller brought scratch
    const size_t need = gbx::qgemm_prefill_workspace_bytes(static_cast<int>(M), N, K);
    if (workspace && need && workspace_bytes >= need && aligned(workspace, 16)) ws = static_cast<float*>(workspace);
    e = gbx::launch_qgemm_mfma(x, qweight, scales, zeros, bias, y, static_cast<int>(M), N, K, bits, log2g, bf16, stream, ws);
  } else if (!force_generic && (K % 128) == 0 && K <= max_k) {
    gbx::GemvArgs a{};
    a.x = x;
    a.seg[0] = gbx::GemvSeg{qweight, static_cast<const unsigned short*>(scales), static_cast<const unsigned short*>(zeros),
                            static_cast<unsigned short*>(y), N};
    a.nseg = 1; a.N = N; a.K = K; a.log2g = log2g; a.bias = bias;
#ifdef GBX_STAMPS
    a.stamps = gbx::stamp_region();
#endif
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

int gbx_qlinear_fused_forward(const gbx_qlinear_fused* d, void* hip_stream) {
  if (!d) return fail(GBX_ERR_INVALID, "null descriptor");
  if (d->nseg < 1 || d->nseg > 3) return fail(GBX_ERR_INVALID, "nseg must be 1..3 (got %d)", d->nseg);
  if (d->act_dtype != GBX_F16 && d->act_dtype != GBX_BF16) return fail(GBX_ERR_INVALID, "act_dtype must be GBX_F16 or GBX_BF16");
  int ntot = 0;
  for (int i = 0; i < d->nseg; ++i) {
    int rc = check_shape(1, d->N[i], d->K, d->bits, d->group_size);
    if (rc) return rc;
    if (!d->qweight[i] || !d->scales[i] || !d->zeros[i]) return fail(GBX_ERR_INVALID, "null device pointer in segment %d", i);
    if (!aligned(d->qweight[i], 16)) return fail(GBX_ERR_INVALID, "qweight must be 16-byte aligned");
    ntot += d->N[i];
  }
  if (!d->x || !aligned(d->x, 16) || !d->y[0]) return fail(GBX_ERR_INVALID, "x / y missing or x not 16-byte aligned");
  if (d->swiglu && (d->nseg != 2 || d->N[0] != d->N[1])) return fail(GBX_ERR_INVALID, "swiglu needs two equal-sized segments");
  if (!d->swiglu) for (int i = 0; i < d->nseg; ++i) if (!d->y[i]) return fail(GBX_ERR_INVALID, "null y in segment %d", i);
  if (d->residual && (d->nseg != 1 || d->swiglu)) return fail(GBX_ERR_INVALID, "residual needs a single plain segment");
  if (d->norm_weight && !aligned(d->norm_weight, 16)) return fail(GBX_ERR_INVALID, "norm_weight must be 16-byte aligned");
  const int K = d->K, bits = d->bits;
  const int vpl = bits == 2 ? 64 : 32;
  const int nslots = K / vpl;
  if ((K % 128) != 0 || nslots > 512) return fail(GBX_ERR_UNSUPPORTED, "fused path needs K %% 128 == 0 and K <= %d (K = %d)", 512 * vpl, K);
  if (d->norm_weight && nslots > 128)
    return fail(GBX_ERR_UNSUPPORTED, "fused RMSNorm needs K <= %d at %d bits; call gbx_rmsnorm first", 128 * vpl, bits);
  gbx::GemvArgs a{};
  a.x = d->x;
  if (d->sz && !aligned(d->sz, 8)) return fail(GBX_ERR_INVALID, "sz must be 8-byte aligned");
  for (int i = 0; i < d->nseg; ++i) {
    if (d->sz && (d->N[i] & 1)) return fail(GBX_ERR_INVALID, "packed scales/zeros need an even row count (segment %d has %d)", i, d->N[i]);
    a.seg[i] = gbx::GemvSeg{d->qweight[i], static_cast<const unsigned short*>(d->scales[i]),
                            static_cast<const unsigned short*>(d->zeros[i]), static_cast<unsigned short*>(d->y[i]), d->N[i]};
  }
  a.sz = d->sz;
  for (int i = d->nseg; i < 3; ++i) a.seg[i] = a.seg[0], a.seg[i].N = 0;     /* valid pointers, zero rows */
  a.nseg = d->nseg;
  a.N = d->swiglu ? d->N[0] : ntot;
  a.K = K;
  a.log2g = log2_group(d->group_size);
  a.norm_w = static_cast<const unsigned short*>(d->norm_weight);
  a.eps = d->norm_eps;
  a.residual = static_cast<const unsigned short*>(d->residual);
  a.swiglu = d->swiglu ? 1 : 0;
#ifdef GBX_STAMPS
  a.stamps = gbx::stamp_region();
#endif
  hipStream_t stream = static_cast<hipStream_t>(hip_stream);
  const bool bf16 = d->act_dtype == GBX_BF16;
  hipError_t e;
  switch (bits) {
    case 2: e = gbx::launch_gemv<2>(a, 1, bf16, stream); break;
    case 3: e = gbx::launch_gemv<3>(a, 1, bf16, stream); break;
    default: e = gbx::launch_gemv<4>(a, 1, bf16, stream); break;
  }
  if (e != hipSuccess) return hip_fail(e, "gbx_qlinear_fused_forward launch");
  return GBX_OK;
}



int gbx_pack_scales_zeros(const void* scales0, const void* zeros0, const void* scales1, const void* zeros1,
                          uint32_t* out, int32_t N, int32_t G, void* hip_stream) {
  if (N < 0 || G <= 0) return fail(GBX_ERR_INVALID, "bad shape N=%d G=%d", N, G);
  if (N == 0) return GBX_OK;
  if (!scales0 || !zeros0 || !out) return fail(GBX_ERR_INVALID, "null device pointer");
  if ((scales1 == nullptr) != (zeros1 == nullptr)) return fail(GBX_ERR_INVALID, "scales1 and zeros1 go together");
  if (!scales1 && (N & 1)) return fail(GBX_ERR_INVALID, "row-pair packing needs an even N (got %d)", N);
  hipError_t e = gbx::launch_pack_sz(scales0, zeros0, scales1, zeros1, out, N, G, static_cast<hipStream_t>(hip_stream));
  if (e != hipSuccess) return hip_fail(e, "gbx_pack_scales_zeros launch");
  return GBX_OK;
}

int gbx_qlinear_forward_multi(const void* x, int32_t nseg, const uint32_t* const* qweight, const void* const* scales,
                              const void* const* zeros, const void* const* bias, void* const* y, const int32_t* N,
                              int64_t M, int32_t K, int32_t bits, int32_t group_size, int32_t act_dtype, void* hip_stream) {
  if (nseg < 1 || nseg > 3 || !qweight || !scales || !zeros || !bias || !y || !N) return fail(GBX_ERR_INVALID, "nseg must be 1..3 and the tables non-null");
  if (act_dtype != GBX_F16 && act_dtype != GBX_BF16) return fail(GBX_ERR_INVALID, "act_dtype must be GBX_F16 or GBX_BF16");
  if (M < 2 || M > 64 || K < 512 || (K % 128) != 0)
    return fail(GBX_ERR_UNSUPPORTED, "multi-matrix launch covers 2 <= M <= 64, K >= 512, K %% 128 == 0 (M=%lld K=%d)", (long long)M, K);
  int n[3] = {0, 0, 0};
  for (int i = 0; i < nseg; ++i) {
    int rc = check_shape(M, N[i], K, bits, group_size);
    if (rc) return rc;
    if (N[i] % 32) return fail(GBX_ERR_UNSUPPORTED, "every N must be a multiple of 32 (segment %d has %d)", i, N[i]);
    if (!qweight[i] || !scales[i] || !zeros[i] || !y[i]) return fail(GBX_ERR_INVALID, "null device pointer in segment %d", i);
    if (!aligned(qweight[i], 16)) return fail(GBX_ERR_INVALID, "qweight must be 16-byte aligned");
    n[i] = N[i];
  }
  if (!x || !aligned(x, 16)) return fail(GBX_ERR_INVALID, "x missing or not 16-byte aligned");
  hipError_t e = gbx::launch_qgemm_skinny_multi(x, nseg, qweight, scales, zeros, bias, y, n, static_cast<int>(M), K, bits,
                                                log2_group(group_size), act_dtype == GBX_BF16, static_cast<hipStream_t>(hip_stream));
  if (e != hipSuccess) return hip_fail(e, "gbx_qlinear_forward_multi launch");
  return GBX_OK;
}

int gbx_qlinear_swiglu_forward(const void* x, const uint32_t* gate_qweight, const void* gate_scales, const void* gate_zeros,
                               const uint32_t* up_qweight, const void* up_scales, const void* up_zeros, void* y, int64_t M,
                               int32_t N, int32_t K, int32_t bits, int32_t group_size, int32_t act_dtype, void* hip_stream) {
  int rc = check_shape(M, N, K, bits, group_size);
  if (rc) return rc;
  if (act_dtype != GBX_F16 && act_dtype != GBX_BF16) return fail(GBX_ERR_INVALID, "act_dtype must be GBX_F16 or GBX_BF16");
  if (!x || !gate_qweight || !gate_scales || !gate_zeros || !up_qweight || !up_scales || !up_zeros || !y) return fail(GBX_ERR_INVALID, "null device pointer");
  if (!aligned(x, 16) || !aligned(gate_qweight, 16) || !aligned(up_qweight, 16)) return fail(GBX_ERR_INVALID, "x and qweight must be 16-byte aligned");
  if (M < 2 || M > 64 || (N % 16) || (K % 1024))
    return fail(GBX_ERR_UNSUPPORTED, "fused SwiGLU GEMM covers 2 <= M <= 64, N %% 16 == 0, K %% 1024 == 0 (M=%lld N=%d K=%d)", (long long)M, N, K);
  hipError_t e = gbx::launch_qgemm_swiglu(x, gate_qweight, gate_scales, gate_zeros, up_qweight, up_scales, up_zeros, y, static_cast<int>(M), N, K,
                                          bits, log2_group(group_size), act_dtype == GBX_BF16, static_cast<hipStream_t>(hip_stream));
  if (e == hipErrorInvalidValue) return fail(GBX_ERR_UNSUPPORTED, "fused SwiGLU GEMM: (bits %d, group %d, N %d) is outside the batch kernel", bits, group_size, N);
  if (e != hipSuccess) return hip_fail(e, "gbx_qlinear_swiglu_forward launch");
  return GBX_OK;
}

size_t gbx_qlinear_swiglu_workspace_bytes(int64_t M, int32_t N, int32_t K) {
  // (the SwiGLU epilogue needs whole tiles: no k split here, so the few-tile shapes stay with the unfused ops)
  if (M < gbx::GBX_DENSE_MIN_M && (M < 1024 || ((M + 255) / 256) * ((2LL * N + 255) / 256) < 176)) return 0;
  if (N <= 0 || K <= 0 || (N % 128) || (K % 128) || !gbx::gemm_dense_supported(static_cast<int>(std::min<int64_t>(M, 61440)), 2 * N, K)) return 0;
  return static_cast<size_t>(2) * N * K * 2;
}

int gbx_qlinear_swiglu_forward_ws(const void* x, const uint32_t* gate_qweight, const void* gate_scales, const void* gate_zeros,
                                  const uint32_t* up_qweight, const void* up_scales, const void* up_zeros, void* y, int64_t M,
                                  int32_t N, int32_t K, int32_t bits, int32_t group_size, int32_t act_dtype, void* workspace,
                                  size_t workspace_bytes, void* hip_stream) {
  const size_t need = gbx_qlinear_swiglu_workspace_bytes(M, N, K);
  if (need == 0 || !workspace || workspace_bytes < need)       // decode-sized batches (and callers without scratch): the one-launch batch kernel
    return gbx_qlinear_swiglu_forward(x, gate_qweight, gate_scales, gate_zeros, up_qweight, up_scales, up_zeros, y, M, N, K, bits, group_size,
                                      act_dtype, hip_stream);
  int rc = check_shape(M, N, K, bits, group_size);
  if (rc) return rc;
  if (act_dtype != GBX_F16 && act_dtype != GBX_BF16) return fail(GBX_ERR_INVALID, "act_dtype must be GBX_F16 or GBX_BF16");
  if (!x || !gate_qweight || !gate_scales || !gate_zeros || !up_qweight || !up_scales || !up_zeros || !y) return fail(GBX_ERR_INVALID, "null device pointer");
  if (!aligned(x, 16) || !aligned(gate_qweight, 16) || !aligned(up_qweight, 16) || !aligned(workspace, 16) || !aligned(y, 16))
    return fail(GBX_ERR_INVALID, "x, qweight, y and workspace must be 16-byte aligned");
  const size_t esz = 2;
  for (int64_t m0 = 0; m0 < M; m0 += 61440) {                  // 32-bit byte offsets inside the kernel: rows in slabs
    const int mm = static_cast<int>(std::min<int64_t>(M - m0, 61440));
    hipError_t e = gbx::launch_qgemm_dense_swiglu(static_cast<const char*>(x) + m0 * K * esz, gate_qweight, gate_scales, gate_zeros, up_qweight,
                                                  up_scales, up_zeros, static_cast<char*>(y) + m0 * N * esz, mm, N, K, bits,
                                                  log2_group(group_size), act_dtype == GBX_BF16, static_cast<hipStream_t>(hip_stream), workspace);
    if (e == hipErrorInvalidValue) return fail(GBX_ERR_UNSUPPORTED, "dense SwiGLU GEMM: shape outside the kernel (N %d, K %d)", N, K);
    if (e != hipSuccess) return hip_fail(e, "gbx_qlinear_swiglu_forward_ws launch");
  }
  return GBX_OK;
}

int gbx_gather_qlinear_forward(const void* x, const uint32_t* qweight, const void* scales, const void* zeros,
                               const void* bias, const int32_t* indices, void* y, int32_t pairs, int32_t x_div,
                               int32_t num_experts, int32_t N, int32_t K, int32_t bits, int32_t group_size,
                               int32_t act_dtype, void* hip_stream) {
  int rc = check_shape(1, N, K, bits, group_size);
  if (rc) return rc;
  if (act_dtype != GBX_F16 && act_dtype != GBX_BF16) return fail(GBX_ERR_INVALID, "act_dtype must be GBX_F16 or GBX_BF16");
  if (pairs < 0 || pairs > 65535 || x_div < 1 || num_experts < 1)
    return fail(GBX_ERR_INVALID, "bad gather shape: pairs=%d x_div=%d num_experts=%d", pairs, x_div, num_experts);
  if (pairs == 0) return GBX_OK;
  if (!x || !qweight || !scales || !zeros || !indices || !y) return fail(GBX_ERR_INVALID, "null device pointer");
  if (!aligned(qweight, 16) || !aligned(x, 16)) return fail(GBX_ERR_INVALID, "x and qweight must be 16-byte aligned");
  const int max_k = bits == 2 ? gbx::gemv_max_k<2>() : bits == 3 ? gbx::gemv_max_k<3>() : gbx::gemv_max_k<4>();
  if ((K % 128) != 0 || K > max_k)
    return fail(GBX_ERR_UNSUPPORTED, "gather path needs K %% 128 == 0 and K <= %d at %d bits (K = %d)", max_k, bits, K);
  if ((static_cast<long long>(N) * (K / 32 * bits) * 4) % 16 != 0) return fail(GBX_ERR_UNSUPPORTED, "expert stride must keep 16-byte alignment");
  gbx::GemvArgs a{};
  a.x = x;
  a.seg[0] = gbx::GemvSeg{qweight, static_cast<const unsigned short*>(scales), static_cast<const unsigned short*>(zeros),
                          static_cast<unsigned short*>(y), N};
  a.seg[1] = a.seg[0]; a.seg[1].N = 0; a.seg[2] = a.seg[1];
  a.nseg = 1; a.N = N; a.K = K; a.log2g = log2_group(group_size); a.bias = bias;
  a.expert_idx = indices; a.x_div = x_div; a.num_experts = num_experts;
  a.e_stride_w = static_cast<long long>(N) * (K / 32 * bits);
  a.e_stride_g = static_cast<long long>(N) * (K / group_size);
  hipStream_t stream = static_cast<hipStream_t>(hip_stream);
  const bool bf16 = act_dtype == GBX_BF16;
  hipError_t e;
  switch (bits) {
    case 2: e = gbx::launch_gemv<2>(a, pairs, bf16, stream); break;
    case 3: e = gbx::launch_gemv<3>(a, pairs, bf16, stream); break;
    default: e = gbx::launch_gemv<4>(a, pairs, bf16, stream); break;
  }
  if (e != hipSuccess) return hip_fail(e, "gbx_gather_qlinear_forward launch");
  return GBX_OK;
}

int gbx_gather_swiglu_forward(const void* x, const uint32_t* gate_qweight, const void* gate_scales, const void* gate_zeros,
                              const uint32_t* up_qweight, const void* up_scales, const void* up_zeros, const int32_t* indices,
                              void* y, int32_t pairs, int32_t x_div, int32_t num_experts, int32_t N, int32_t K, int32_t bits,
                              int32_t group_size, int32_t act_dtype, void* hip_stream) {
  int rc = check_shape(1, N, K, bits, group_size);
  if (rc) return rc;
  if (act_dtype != GBX_F16 && act_dtype != GBX_BF16) return fail(GBX_ERR_INVALID, "act_dtype must be GBX_F16 or GBX_BF16");
  if (pairs < 0 || pairs > 65535 || x_div < 1 || num_experts < 1)
    return fail(GBX_ERR_INVALID, "bad gather shape: pairs=%d x_div=%d num_experts=%d", pairs, x_div, num_experts);
  if (pairs == 0) return GBX_OK;
  if (!x || !gate_qweight || !gate_scales || !gate_zeros || !up_qweight || !up_scales || !up_zeros || !indices || !y)
    return fail(GBX_ERR_INVALID, "null device pointer");
  if (!aligned(gate_qweight, 16) || !aligned(up_qweight, 16) || !aligned(x, 16)) return fail(GBX_ERR_INVALID, "x and qweight must be 16-byte aligned");
  const int max_k = bits == 2 ? gbx::gemv_max_k<2>() : bits == 3 ? gbx::gemv_max_k<3>() : gbx::gemv_max_k<4>();
  if ((K % 128) != 0 || K > max_k)
    return fail(GBX_ERR_UNSUPPORTED, "gather path needs K %% 128 == 0 and K <= %d at %d bits (K = %d)", max_k, bits, K);
  if ((static_cast<long long>(N) * (K / 32 * bits) * 4) % 16 != 0) return fail(GBX_ERR_UNSUPPORTED, "expert stride must keep 16-byte alignment");
  gbx::GemvArgs a{};
  a.x = x;
  a.seg[0] = gbx::GemvSeg{gate_qweight, static_cast<const unsigned short*>(gate_scales), static_cast<const unsigned short*>(gate_zeros),
                          static_cast<unsigned short*>(y), N};
  a.seg[1] = gbx::GemvSeg{up_qweight, static_cast<const unsigned short*>(up_scales), static_cast<const unsigned short*>(up_zeros),
                          static_cast<unsigned short*>(y), N};
  a.seg[2] = a.seg[0]; a.seg[2].N = 0;
  a.nseg = 2; a.N = N; a.K = K; a.log2g = log2_group(group_size); a.swiglu = 1;
  a.expert_idx = indices; a.x_div = x_div; a.num_experts = num_experts;
  a.e_stride_w = static_cast<long long>(N) * (K / 32 * bits);
  a.e_stride_g = static_cast<long long>(N) * (K / group_size);
  hipStream_t stream = static_cast<hipStream_t>(hip_stream);
  const bool bf16 = act_dtype == GBX_BF16;
  hipError_t e;
  switch (bits) {
    case 2: e = gbx::launch_gemv<2>(a, pairs, bf16, stream); break;
    case 3: e = gbx::launch_gemv<3>(a, pairs, bf16, stream); break;
    default: e = gbx::launch_gemv<4>(a, pairs, bf16, stream); break;
  }
  if (e != hipSuccess) return hip_fail(e, "gbx_gather_swiglu_forward launch");
  return GBX_OK;
}




int gbx_attn_decode(const void* q, const void* k, const void* v, void* kcache, void* vcache, const float* inv_freq,
                    const int64_t* pos, void* out, int32_t n_heads, int32_t n_kv_heads, int32_t head_dim,
                    int32_t capacity, float rope_scale, float sm_scale, int32_t traditional, int32_t act_dtype,
                    void* hip_stream) {
  return gbx_attn_decode_batched(q, k, v, kcache, vcache, inv_freq, pos, out, 1, n_heads, n_kv_heads, head_dim, capacity,
                                 rope_scale, sm_scale, traditional, act_dtype, hip_stream);
}



int gbx_attn_decode_batched(const void* q, const void* k, const void* v, void* kcache, void* vcache, const float* inv_freq,
                            const int64_t* pos, void* out, int32_t batch, int32_t n_heads, int32_t n_kv_heads,
                            int32_t head_dim, int32_t capacity, float rope_scale, float sm_scale, int32_t traditional,
                            int32_t act_dtype, void* hip_stream) {
  return gbx_attn_decode_normed(q, k, v, kcache, vcache, inv_freq, pos, out, batch, n_heads, n_kv_heads, head_dim, capacity,
                                rope_scale, sm_scale, traditional, nullptr, nullptr, 0.f, nullptr, nullptr, nullptr, act_dtype,
                                hip_stream);
}

int gbx_attn_decode_normed(const void* q, const void* k, const void* v, void* kcache, void* vcache, const float* inv_freq,
                           const int64_t* pos, void* out, int32_t batch, int32_t n_heads, int32_t n_kv_heads,
                           int32_t head_dim, int32_t capacity, float rope_scale, float sm_scale, int32_t traditional,
                           const void* q_norm_weight, const void* k_norm_weight, float norm_eps, const void* q_bias,
                           const void* k_bias, const void* v_bias, int32_t act_dtype, void* hip_stream) {
  return gbx_attn_decode_ws(q, k, v, kcache, vcache, inv_freq, pos, out, batch, n_heads, n_kv_heads, head_dim, capacity, rope_scale,
                            sm_scale, traditional, q_norm_weight, k_norm_weight, norm_eps, q_bias, k_bias, v_bias, act_dtype, nullptr, 0,
                            hip_stream);
}

size_t gbx_attn_workspace_bytes(int32_t batch, int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int32_t capacity) {
  if (batch <= 0 || n_heads <= 0 || n_kv_heads <= 0 || n_heads % n_kv_heads || capacity <= 0 || (head_dim != 64 && head_dim != 128)) return 0;
  return gbx::attn_workspace_bytes(batch, n_heads, n_kv_heads, head_dim, capacity);
}

int gbx_attn_decode_ws(const void* q, const void* k, const void* v, void* kcache, void* vcache, const float* inv_freq,
                       const int64_t* pos, void* out, int32_t batch, int32_t n_heads, int32_t n_kv_heads,
                       int32_t head_dim, int32_t capacity, float rope_scale, float sm_scale, int32_t traditional,
                       const void* q_norm_weight, const void* k_norm_weight, float norm_eps, const void* q_bias,
                       const void* k_bias, const void* v_bias, int32_t act_dtype, void* workspace, size_t workspace_bytes,
                       void* hip_stream) {
  if ((q_norm_weight == nullptr) != (k_norm_weight == nullptr)) return fail(GBX_ERR_INVALID, "q and k norm weights come together");
  if ((q_bias == nullptr) != (k_bias == nullptr) || (q_bias == nullptr) != (v_bias == nullptr))
    return fail(GBX_ERR_INVALID, "q, k and v biases come together");
  if (batch == 0) return GBX_OK;
  if (batch < 0 || batch > 65535) return fail(GBX_ERR_INVALID, "batch must be in [0, 65535] (got %d)", batch);
  if (!q || !k || !v || !kcache || !vcache || !inv_freq || !pos || !out) return fail(GBX_ERR_INVALID, "null device pointer");
  if (head_dim != 64 && head_dim != 128) return fail(GBX_ERR_UNSUPPORTED, "head_dim must be 64 or 128 (got %d)", head_dim);
  if (n_heads <= 0 || n_kv_heads <= 0 || n_heads % n_kv_heads) return fail(GBX_ERR_INVALID, "n_heads must be a multiple of n_kv_heads");
  if (capacity <= 0) return fail(GBX_ERR_INVALID, "capacity must be positive");
  if (act_dtype != GBX_F16 && act_dtype != GBX_BF16) return fail(GBX_ERR_INVALID, "act_dtype must be GBX_F16 or GBX_BF16");
  if (!aligned(kcache, 16) || !aligned(vcache, 16)) return fail(GBX_ERR_INVALID, "KV cache must be 16-byte aligned");
  float* ws = nullptr;                              // enough caller scratch: long contexts are split over workgroups
  {
    const size_t need = gbx::attn_workspace_bytes(batch, n_heads, n_kv_heads, head_dim, capacity);
    if (workspace && need && workspace_bytes >= need && aligned(workspace, 16)) ws = static_cast<float*>(workspace);
  }
  hipError_t e = gbx::launch_attn_decode(q, k, v, kcache, vcache, inv_freq, reinterpret_cast<const long long*>(pos), out,
                                         batch, n_heads, n_kv_heads, head_dim, capacity, rope_scale, sm_scale, traditional,
                                         act_dtype == GBX_BF16, static_cast<hipStream_t>(hip_stream), q_norm_weight, k_norm_weight,
                                         norm_eps, q_bias, k_bias, v_bias, ws);
  if (e != hipSuccess) return hip_fail(e, "gbx_attn_decode launch");
  return GBX_OK;
}

int gbx_attn_decode_slices(const float* qkv_slices_ws, int32_t slices, void* kcache, void* vcache, const float* inv_freq,
                           const int64_t* pos, void* out, int32_t batch, int32_t n_heads, int32_t n_kv_heads,
                           int32_t head_dim, int32_t capacity, float rope_scale, float sm_scale, int32_t traditional,
                           const void* q_norm_weight, const void* k_norm_weight, float norm_eps, int32_t act_dtype, void* workspace,
                           size_t workspace_bytes, void* hip_stream) {
  if ((q_norm_weight == nullptr) != (k_norm_weight == nullptr)) return fail(GBX_ERR_INVALID, "q and k norm weights come together");
  if (batch == 0) return GBX_OK;
  if (batch < 0 || batch > 65535) return fail(GBX_ERR_INVALID, "batch must be in [0, 65535] (got %d)", batch);
  if (!qkv_slices_ws || !kcache || !vcache || !inv_freq || !pos || !out) return fail(GBX_ERR_INVALID, "null device pointer");
  if (slices < 1 || slices > 8) return fail(GBX_ERR_UNSUPPORTED, "1 <= slices <= 8 (got %d)", slices);
  if (head_dim != 64 && head_dim != 128) return fail(GBX_ERR_UNSUPPORTED, "head_dim must be 64 or 128 (got %d)", head_dim);
  if (n_heads <= 0 || n_kv_heads <= 0 || n_heads % n_kv_heads) return fail(GBX_ERR_INVALID, "n_heads must be a multiple of n_kv_heads");
  if (capacity <= 0) return fail(GBX_ERR_INVALID, "capacity must be positive");
  if (act_dtype != GBX_F16 && act_dtype != GBX_BF16) return fail(GBX_ERR_INVALID, "act_dtype must be GBX_F16 or GBX_BF16");
  if (!aligned(kcache, 16) || !aligned(vcache, 16) || !aligned(qkv_slices_ws, 4)) return fail(GBX_ERR_INVALID, "misaligned pointer");
  float* ws = nullptr;
  {
    const size_t need = gbx::attn_workspace_bytes(batch, n_heads, n_kv_heads, head_dim, capacity);
    if (workspace && need && workspace_bytes >= need && aligned(workspace, 16)) ws = static_cast<float*>(workspace);
  }
  hipError_t e = gbx::launch_attn_decode(nullptr, nullptr, nullptr, kcache, vcache, inv_freq, reinterpret_cast<const long long*>(pos), out,
                                         batch, n_heads, n_kv_heads, head_dim, capacity, rope_scale, sm_scale, traditional,
                                         act_dtype == GBX_BF16, static_cast<hipStream_t>(hip_stream), q_norm_weight, k_norm_weight,
                                         norm_eps, nullptr, nullptr, nullptr, ws, qkv_slices_ws, slices);
  if (e != hipSuccess) return hip_fail(e, "gbx_attn_decode_slices launch");
  return GBX_OK;
}

int gbx_embed_next_token(void* argmax_keys, int64_t* token_io, int64_t* pos, const void* embed_table, void* h,
                         int32_t D, int32_t vocab, int32_t from_keys, int32_t advance_pos, int32_t act_dtype, void* hip_stream) {
  if (!argmax_keys || !token_io || !pos || !embed_table || !h) return fail(GBX_ERR_INVALID, "null device pointer");
  if (D <= 0 || (D % 8) || vocab <= 0) return fail(GBX_ERR_INVALID, "need D %% 8 == 0 and vocab > 0");
  if (!aligned(embed_table, 16) || !aligned(h, 16) || !aligned(argmax_keys, 8)) return fail(GBX_ERR_INVALID, "misaligned pointer");
  if (act_dtype != GBX_F16 && act_dtype != GBX_BF16) return fail(GBX_ERR_INVALID, "act_dtype must be GBX_F16 or GBX_BF16");
  hipError_t e = gbx::launch_embed_next(argmax_keys, token_io, pos, embed_table, h, D, vocab, from_keys ? 1 : 0,
                                        advance_pos ? 1 : 0, act_dtype == GBX_BF16, static_cast<hipStream_t>(hip_stream));
  if (e != hipSuccess) return hip_fail(e, "gbx_embed_next_token launch");
  return GBX_OK;
}

/* ---- one-shot P2P all-reduce: inbox management (hipIpc) and the collective ---- */
size_t gbx_p2p_inbox_bytes(int32_t world, int32_t max_elems) {
  if (world < 2 || world > 8 || max_elems <= 0) return 0;
  return static_cast<size_t>(2) * world * ((max_elems + 1) / 2) * 8 + 64;       /* [parity][source][granule] + the state words */
}

int gbx_p2p_inbox_create(int32_t world, int32_t max_elems, void** inbox_out, void* ipc_handle_out) {
  const size_t bytes = gbx_p2p_inbox_bytes(world, max_elems);
  if (!bytes || !inbox_out || !ipc_handle_out) return fail(GBX_ERR_INVALID, "bad world / max_elems / output pointers");
  void* p = nullptr;
  /* Peers write this memory over xGMI while a kernel of THIS GPU polls it: that needs fine-grained (uncached-at-L2) memory.  A
   * coarse-grained hipMalloc would let the local L2 serve stale lines to the poll -- a hang or a stale payload that no single-GPU
   * test can show -- so there is no fallback: the caller keeps RCCL (tp.enable_p2p).  GBX_P2P_FORCE_COARSE=1: test hook. */
  hipError_t e = gbx::env_switch("GBX_P2P_FORCE_COARSE") ? hipErrorNotSupported : hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(GBX_ERR_UNSUPPORTED, "gbx_p2p_inbox_create: fine-grained device memory is not available (%s); the one-shot P2P all-reduce "
                "needs it -- use the RCCL all-reduce", hipGetErrorString(e));
  }
  e = hipMemset(p, 0, bytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  hipIpcMemHandle_t h;
  if (e == hipSuccess) e = hipIpcGetMemHandle(&h, p);
  if (e != hipSuccess) { (void)hipFree(p); return hip_fail(e, "gbx_p2p_inbox_create (memset / hipIpcGetMemHandle)"); }
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
  memcpy(ipc_handle_out, &h, sizeof(h));
  *inbox_out = p;
  return GBX_OK;
}

int gbx_p2p_inbox_open(const void* ipc_handle, void** mapped_out) {
  if (!ipc_handle || !mapped_out) return fail(GBX_ERR_INVALID, "null pointer");
  hipIpcMemHandle_t h;
  memcpy(&h, ipc_handle, sizeof(h));
  void* p = nullptr;
  hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) return hip_fail(e, "hipIpcOpenMemHandle");
  *mapped_out = p;
  return GBX_OK;
}

int gbx_p2p_inbox_close(void* mapped) { return mapped && hipIpcCloseMemHandle(mapped) != hipSuccess ? fail(GBX_ERR_HIP, "hipIpcCloseMemHandle failed") : GBX_OK; }
int gbx_p2p_inbox_destroy(void* inbox) { return inbox && hipFree(inbox) != hipSuccess ? fail(GBX_ERR_HIP, "hipFree failed") : GBX_OK; }

int gbx_p2p_status(const void* inbox, int32_t world, int32_t max_elems, int32_t* timed_out) {
  const size_t bytes = gbx_p2p_inbox_bytes(world, max_elems);
  if (!inbox || !bytes || !timed_out) return fail(GBX_ERR_INVALID, "bad inbox / world / max_elems / output pointer");
  unsigned flag = 0;        /* state word 2, behind the slabs: set by a launch whose wait for a peer ran into its bound */
  hipError_t e = hipMemcpy(&flag, static_cast<const char*>(inbox) + bytes - 64 + 8, sizeof(flag), hipMemcpyDeviceToHost);
  if (e != hipSuccess) return hip_fail(e, "gbx_p2p_status copy");
  *timed_out = flag ? 1 : 0;
  return GBX_OK;
}

int gbx_p2p_allreduce(void* x, int32_t n, int32_t rank, int32_t world, void* const* inboxes, int32_t max_elems, int32_t act_dtype,
                      void* hip_stream) {
  if (!x || !inboxes) return fail(GBX_ERR_INVALID, "null pointer");
  if (act_dtype != GBX_F16 && act_dtype != GBX_BF16) return fail(GBX_ERR_INVALID, "act_dtype must be GBX_F16 or GBX_BF16");
  if (world < 2 || world > 8 || rank < 0 || rank >= world) return fail(GBX_ERR_INVALID, "bad rank / world (%d / %d)", rank, world);
  if (n <= 0 || (n & 1) || n > max_elems || !aligned(x, 4)) return fail(GBX_ERR_INVALID, "n must be even, <= max_elems (%d), x 4-byte aligned (n = %d)", max_elems, n);
  for (int p = 0; p < world; ++p) if (!inboxes[p] || !aligned(inboxes[p], 8)) return fail(GBX_ERR_INVALID, "inbox %d missing or misaligned", p);
  const int cap = (max_elems + 1) / 2;
  void* state = static_cast<char*>(inboxes[rank]) + static_cast<size_t>(2) * world * cap * 8;     /* own inbox: the two state words behind the slabs */
  hipError_t e = gbx::launch_p2p_allreduce(x, n, rank, world, inboxes, cap, state, act_dtype == GBX_BF16, static_cast<hipStream_t>(hip_stream));
  if (e != hipSuccess) return hip_fail(e, "gbx_p2p_allreduce launch");
  return GBX_OK;
}

int gbx_embed_next_token_sharded(void* gathered_keys, int32_t nkeys, void* local_keys, int64_t* token_io, int64_t* pos,
                                 const void* embed_table, void* h, int32_t D, int32_t vocab, int32_t advance_pos, int32_t act_dtype,
                                 void* hip_stream) {
  if (!gathered_keys || !token_io || !pos || !embed_table || !h) return fail(GBX_ERR_INVALID, "null device pointer");
  if (nkeys < GBX_ARGMAX_SLOTS || (nkeys % GBX_ARGMAX_SLOTS)) return fail(GBX_ERR_INVALID, "nkeys must be a multiple of %d (got %d)", GBX_ARGMAX_SLOTS, nkeys);
  if (D <= 0 || (D % 8) || vocab <= 0) return fail(GBX_ERR_INVALID, "need D %% 8 == 0 and vocab > 0");
  if (!aligned(embed_table, 16) || !aligned(h, 16) || !aligned(gathered_keys, 8) || (local_keys && !aligned(local_keys, 8)))
    return fail(GBX_ERR_INVALID, "misaligned pointer");
  if (act_dtype != GBX_F16 && act_dtype != GBX_BF16) return fail(GBX_ERR_INVALID, "act_dtype must be GBX_F16 or GBX_BF16");
  hipError_t e = gbx::launch_embed_next(gathered_keys, token_io, pos, embed_table, h, D, vocab, 1, advance_pos ? 1 : 0,
                                        act_dtype == GBX_BF16, static_cast<hipStream_t>(hip_stream), nkeys, local_keys);
  if (e != hipSuccess) return hip_fail(e, "gbx_embed_next_token_sharded launch");
  return GBX_OK;
}

int gbx_dense_gemv_argmax_shard(const void* x, const void* w, void* y, const void* norm_weight, float norm_eps, int32_t N, int32_t K,
                                void* argmax_keys, int32_t index_offset, int32_t act_dtype, void* hip_stream) {
  if (!x || !w || !y) return fail(GBX_ERR_INVALID, "null device pointer");
  if (N <= 0 || K <= 0 || (K % 8) || index_offset < 0) return fail(GBX_ERR_INVALID, "need N > 0, K %% 8 == 0, index_offset >= 0 (N=%d K=%d)", N, K);
  if (K > 8192) return fail(GBX_ERR_UNSUPPORTED, "K = %d exceeds the dense GEMV limit of 8192", K);
  if (!aligned(x, 16) || !aligned(w, 16) || (norm_weight && !aligned(norm_weight, 16)))
    return fail(GBX_ERR_INVALID, "x, w and norm_weight must be 16-byte aligned");
  if (act_dtype != GBX_F16 && act_dtype != GBX_BF16) return fail(GBX_ERR_INVALID, "act_dtype must be GBX_F16 or GBX_BF16");
  if (argmax_keys && !aligned(argmax_keys, 8)) return fail(GBX_ERR_INVALID, "argmax_keys must be 8-byte aligned");
  hipError_t e = gbx::launch_dense_gemv(x, w, y, norm_weight, norm_eps, N, K, argmax_keys, act_dtype == GBX_BF16,
                                        static_cast<hipStream_t>(hip_stream), index_offset);
  if (e != hipSuccess) return hip_fail(e, "gbx_dense_gemv_argmax_shard launch");
  return GBX_OK;
}

int gbx_dense_gemv(const void* x, const void* w, void* y, const void* norm_weight, float norm_eps, int32_t N, int32_t K,
                   int32_t act_dtype, void* hip_stream) {
  return gbx_dense_gemv_argmax(x, w, y, norm_weight, norm_eps, N, K, nullptr, act_dtype, hip_stream);
}

int gbx_dense_gemv_argmax(const void* x, const void* w, void* y, const void* norm_weight, float norm_eps, int32_t N, int32_t K,
                          void* argmax_keys, int32_t act_dtype, void* hip_stream) {
  if (!x || !w || !y) return fail(GBX_ERR_INVALID, "null device pointer");
  if (N <= 0 || K <= 0 || (K % 8)) return fail(GBX_ERR_INVALID, "need N > 0 and K %% 8 == 0 (N=%d K=%d)", N, K);
  if (K > 8192) return fail(GBX_ERR_UNSUPPORTED, "K = %d exceeds the dense GEMV limit of 8192", K);
  if (!aligned(x, 16) || !aligned(w, 16) || (norm_weight && !aligned(norm_weight, 16)))
    return fail(GBX_ERR_INVALID, "x, w and norm_weight must be 16-byte aligned");
  if (act_dtype != GBX_F16 && act_dtype != GBX_BF16) return fail(GBX_ERR_INVALID, "act_dtype must be GBX_F16 or GBX_BF16");
  if (argmax_keys && !aligned(argmax_keys, 8)) return fail(GBX_ERR_INVALID, "argmax_keys must be 8-byte aligned");
  hipError_t e = gbx::launch_dense_gemv(x, w, y, norm_weight, norm_eps, N, K, argmax_keys, act_dtype == GBX_BF16,
                                        static_cast<hipStream_t>(hip_stream));
  if (e != hipSuccess) return hip_fail(e, "gbx_dense_gemv launch");
  return GBX_OK;
}

int gbx_rope(const void* x, void* y, const float* inv_freq, const int64_t* pos, int64_t offset, int64_t B, int32_t H, int32_t L,
             int32_t D, int32_t dims, int64_t stride_b, int64_t stride_h, int64_t stride_l, float scale, float mscale,
             int32_t traditional, int32_t act_dtype, void* hip_stream) {
  if (B == 0 || H == 0 || L == 0) return GBX_OK;
  if (!x || !y || !inv_freq) return fail(GBX_ERR_INVALID, "null device pointer");
  if (x == y) return fail(GBX_ERR_INVALID, "gbx_rope is out of place");
  if (B < 0 || H < 0 || L < 0 || D <= 0 || dims <= 0 || (dims & 1) || dims > D)
    return fail(GBX_ERR_INVALID, "bad shape B=%lld H=%d L=%d D=%d dims=%d", (long long)B, H, L, D, dims);
  if (act_dtype != GBX_F16 && act_dtype != GBX_BF16) return fail(GBX_ERR_INVALID, "act_dtype must be GBX_F16 or GBX_BF16");
  hipError_t e = gbx::launch_rope(x, y, inv_freq, reinterpret_cast<const long long*>(pos), offset, B, H, L, D, dims, stride_b,
                                  stride_h, stride_l, scale, mscale, traditional, act_dtype == GBX_BF16,
                                  static_cast<hipStream_t>(hip_stream));
  if (e != hipSuccess) return hip_fail(e, "gbx_rope launch");
  return GBX_OK;
}

int gbx_rope_kv_write(const void* k, const void* v, void* kcache, void* vcache, const float* inv_freq, int64_t pos0, int64_t B,
                      int32_t H, int32_t L, int32_t D, int64_t k_stride_b, int64_t k_stride_h, int64_t k_stride_l, int64_t v_stride_b,
                      int64_t v_stride_h, int64_t v_stride_l, int32_t capacity, int32_t t0, float scale, int32_t traditional,
                      int32_t act_dtype, void* hip_stream) {
  if (B == 0 || H == 0 || L == 0) return GBX_OK;
  if (!k || !v || !kcache || !vcache || !inv_freq) return fail(GBX_ERR_INVALID, "null device pointer");
  if (B < 0 || B > 65535 || H <= 0 || L < 0 || t0 < 0 || capacity < t0 + L)
    return fail(GBX_ERR_INVALID, "bad shape B=%lld H=%d L=%d capacity=%d t0=%d", (long long)B, H, L, capacity, t0);
  if (D % 16 || D <= 0 || D > 256) return fail(GBX_ERR_UNSUPPORTED, "head_dim must be a multiple of 16 up to 256 (got %d)", D);
  if (act_dtype != GBX_F16 && act_dtype != GBX_BF16) return fail(GBX_ERR_INVALID, "act_dtype must be GBX_F16 or GBX_BF16");
  if (!aligned(k, 16) || !aligned(v, 16) || !aligned(kcache, 16) || !aligned(vcache, 16) ||
      ((k_stride_b | k_stride_h | k_stride_l | v_stride_b | v_stride_h | v_stride_l) & 7))
    return fail(GBX_ERR_INVALID, "k / v / caches must be 16-byte aligned with strides that are multiples of 8 elements");
  hipError_t e = gbx::launch_rope_kv_write(k, v, kcache, vcache, inv_freq, pos0, B, H, L, D, k_stride_b, k_stride_h, k_stride_l,
                                           v_stride_b, v_stride_h, v_stride_l, capacity, t0, scale, traditional, act_dtype == GBX_BF16,
                                           static_cast<hipStream_t>(hip_stream));
  if (e != hipSuccess) return hip_fail(e, "gbx_rope_kv_write launch");
  return GBX_OK;
}

int gbx_rope_append(void* q, const void* k, const void* v, void* kcache, void* vcache, const float* inv_freq,
                    const int64_t* pos, int32_t batch, int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int32_t capacity,
                    float rope_scale, int32_t traditional, int32_t act_dtype, void* hip_stream) {
  if (batch == 0) return GBX_OK;
  if (!q || !k || !v || !kcache || !vcache || !inv_freq || !pos) return fail(GBX_ERR_INVALID, "null device pointer");
  if (batch < 0 || batch > 65535 || n_heads <= 0 || n_kv_heads <= 0 || head_dim <= 0 || (head_dim & 1) || capacity <= 0)
    return fail(GBX_ERR_INVALID, "bad shape batch=%d heads=%d/%d head_dim=%d capacity=%d", batch, n_heads, n_kv_heads, head_dim, capacity);
  if (act_dtype != GBX_F16 && act_dtype != GBX_BF16) return fail(GBX_ERR_INVALID, "act_dtype must be GBX_F16 or GBX_BF16");
  hipError_t e = gbx::launch_rope_append(q, k, v, kcache, vcache, inv_freq, reinterpret_cast<const long long*>(pos), batch, n_heads,
                                         n_kv_heads, head_dim, capacity, rope_scale, traditional, act_dtype == GBX_BF16,
                                         static_cast<hipStream_t>(hip_stream));
  if (e != hipSuccess) return hip_fail(e, "gbx_rope_append launch");
  return GBX_OK;
}

int gbx_eminf_optimize(const void* logits, int32_t logits_dtype, float* best_logits, float* scratch, float* out_entropy,
                       int32_t V, float alpha, float threshold, int32_t num_steps, void* hip_stream) {
  if (V <= 0) return fail(GBX_ERR_INVALID, "bad vocabulary size %d", V);
  if (!logits || !best_logits || !scratch || !out_entropy) return fail(GBX_ERR_INVALID, "null device pointer");
  if (logits_dtype != GBX_F16 && logits_dtype != GBX_BF16 && logits_dtype != GBX_F32) return fail(GBX_ERR_INVALID, "logits_dtype must be GBX_F16, GBX_BF16 or GBX_F32");
  if (!(threshold > 0.f)) return fail(GBX_ERR_INVALID, "threshold must be positive");
  hipError_t e = gbx::launch_eminf(logits, best_logits, scratch, out_entropy, V, logits_dtype, alpha, threshold, num_steps,
                                   static_cast<hipStream_t>(hip_stream));
  if (e != hipSuccess) return hip_fail(e, "gbx_eminf_optimize launch");
  return GBX_OK;
}

int gbx_rmsnorm(const void* x, const void* w, void* y, int64_t M, int32_t K, float eps, int32_t act_dtype, void* hip_stream) {
  if (M == 0) return GBX_OK;
  if (!x || !w || !y) return fail(GBX_ERR_INVALID, "null device pointer");
  if (M < 0 || M > 65535 * 32768LL || K <= 0) return fail(GBX_ERR_INVALID, "bad shape M=%lld K=%d", (long long)M, K);
  if (act_dtype != GBX_F16 && act_dtype != GBX_BF16) return fail(GBX_ERR_INVALID, "act_dtype must be GBX_F16 or GBX_BF16");
  hipError_t e = gbx::launch_rmsnorm(x, w, y, static_cast<int>(M), K, eps, act_dtype == GBX_BF16, static_cast<hipStream_t>(hip_stream));
  if (e != hipSuccess) return hip_fail(e, "gbx_rmsnorm launch");
  return GBX_OK;
}

int gbx_attn_cached(const void* q, const void* kcache, const void* vcache, const int64_t* pos, void* out, int32_t batch,
                    int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int32_t capacity, float sm_scale, int32_t act_dtype,
                    void* hip_stream) {
  if (batch == 0) return GBX_OK;
  if (!q || !kcache || !vcache || !pos || !out) return fail(GBX_ERR_INVALID, "null device pointer");
  if (batch < 0 || batch > 65535) return fail(GBX_ERR_INVALID, "batch must be in [0, 65535] (got %d)", batch);
  if (head_dim != 64 && head_dim != 128) return fail(GBX_ERR_UNSUPPORTED, "head_dim must be 64 or 128 (got %d)", head_dim);
  if (n_heads <= 0 || n_kv_heads <= 0 || n_heads % n_kv_heads) return fail(GBX_ERR_INVALID, "n_heads must be a multiple of n_kv_heads");
  if (capacity <= 0) return fail(GBX_ERR_INVALID, "capacity must be positive");
  if (act_dtype != GBX_F16 && act_dtype != GBX_BF16) return fail(GBX_ERR_INVALID, "act_dtype must be GBX_F16 or GBX_BF16");
  if (!aligned(kcache, 16) || !aligned(vcache, 16)) return fail(GBX_ERR_INVALID, "KV cache must be 16-byte aligned");
  hipError_t e = gbx::launch_attn_static(q, kcache, vcache, reinterpret_cast<const long long*>(pos), out, batch, n_heads, n_kv_heads,
                                         head_dim, capacity, sm_scale, act_dtype == GBX_BF16, static_cast<hipStream_t>(hip_stream));
  if (e != hipSuccess) return hip_fail(e, "gbx_attn_cached launch");
  return GBX_OK;
}

int gbx_attn_prefill(const void* q, const void* k, const void* v, void* out, int32_t batch, int32_t n_heads, int32_t n_kv_heads,
                     int32_t L, int32_t S, int32_t head_dim, int64_t q_stride_b, int64_t q_stride_h, int64_t q_stride_l,
                     int64_t k_stride_b, int64_t k_stride_h, int64_t v_stride_b, int64_t v_stride_h, int32_t offset, float sm_scale,
                     int32_t act_dtype, void* hip_stream) {
  if (batch == 0 || L == 0) return GBX_OK;
  if (!q || !k || !v || !out) return fail(GBX_ERR_INVALID, "null device pointer");
  if (batch < 0 || batch > 65535) return fail(GBX_ERR_INVALID, "batch must be in [0, 65535] (got %d)", batch);
  if (head_dim != 64 && head_dim != 128) return fail(GBX_ERR_UNSUPPORTED, "head_dim must be 64 or 128 (got %d)", head_dim);
  if (n_heads <= 0 || n_heads > 65535 || n_kv_heads <= 0 || n_heads % n_kv_heads)
    return fail(GBX_ERR_INVALID, "n_heads must be a multiple of n_kv_heads");
  if (L < 0 || offset < 0 || S < offset + L) return fail(GBX_ERR_INVALID, "need S >= offset + L (L %d, S %d, offset %d)", L, S, offset);
  if (act_dtype != GBX_F16 && act_dtype != GBX_BF16) return fail(GBX_ERR_INVALID, "act_dtype must be GBX_F16 or GBX_BF16");
  if (!aligned(q, 16) || !aligned(k, 16) || !aligned(v, 16) || !aligned(out, 8)) return fail(GBX_ERR_INVALID, "q / k / v must be 16-byte aligned");
  if ((q_stride_b | q_stride_h | q_stride_l | k_stride_b | k_stride_h | v_stride_b | v_stride_h) & 7)
    return fail(GBX_ERR_INVALID, "strides must be multiples of 8 elements");
  hipError_t e = gbx::launch_attn_prefill(q, k, v, out, batch, n_heads, n_kv_heads, L, S, head_dim, q_stride_b, q_stride_h, q_stride_l,
                                          k_stride_b, k_stride_h, v_stride_b, v_stride_h, offset, sm_scale, act_dtype == GBX_BF16,
                                          static_cast<hipStream_t>(hip_stream));
  if (e != hipSuccess) return hip_fail(e, "gbx_attn_prefill launch");
  return GBX_OK;
}

int gbx_moe_route(const void* logits, int32_t* indices, void* scores, int64_t tokens, int32_t num_experts, int32_t top_k,
                  int32_t mode, int32_t renormalize, int32_t act_dtype, void* hip_stream) {
  if (tokens == 0) return GBX_OK;
  if (!logits || !indices || !scores) return fail(GBX_ERR_INVALID, "null device pointer");
  if (tokens < 0 || tokens > 2147483647LL) return fail(GBX_ERR_INVALID, "bad token count %lld", (long long)tokens);
  if (num_experts <= 0 || num_experts > 1024) return fail(GBX_ERR_UNSUPPORTED, "num_experts must be in [1, 1024] (got %d)", num_experts);
  if (top_k <= 0 || top_k > 8 || top_k > num_experts) return fail(GBX_ERR_UNSUPPORTED, "top_k must be in [1, min(8, num_experts)] (got %d)", top_k);
  if (mode != 0 && mode != 1) return fail(GBX_ERR_INVALID, "mode must be 0 (softmax then top-k) or 1 (top-k then softmax)");
  if (act_dtype != GBX_F16 && act_dtype != GBX_BF16) return fail(GBX_ERR_INVALID, "act_dtype must be GBX_F16 or GBX_BF16");
  hipError_t e = gbx::launch_moe_route(logits, indices, scores, static_cast<int>(tokens), num_experts, top_k, mode, renormalize ? 1 : 0,
                                       act_dtype == GBX_BF16, static_cast<hipStream_t>(hip_stream));
  if (e != hipSuccess) return hip_fail(e, "gbx_moe_route launch");
  return GBX_OK;
}

int gbx_moe_combine(const void* y, const void* scores, void* out, int64_t tokens, int32_t top_k, int32_t D, int32_t act_dtype,
                    void* hip_stream) {
  return gbx_moe_combine_residual(y, scores, nullptr, out, tokens, top_k, D, act_dtype, hip_stream);
}

int gbx_moe_combine_residual(const void* y, const void* scores, const void* residual, void* out, int64_t tokens, int32_t top_k,
                             int32_t D, int32_t act_dtype, void* hip_stream) {
  if (tokens == 0) return GBX_OK;
  if (!y || !scores || !out) return fail(GBX_ERR_INVALID, "null device pointer");
  if (tokens < 0 || tokens > 65535 || top_k <= 0 || D <= 0 || (D & 1)) return fail(GBX_ERR_INVALID, "bad shape tokens=%lld top_k=%d D=%d", (long long)tokens, top_k, D);
  if (act_dtype != GBX_F16 && act_dtype != GBX_BF16) return fail(GBX_ERR_INVALID, "act_dtype must be GBX_F16 or GBX_BF16");
  hipError_t e = gbx::launch_moe_combine(y, scores, out, static_cast<int>(tokens), top_k, D, act_dtype == GBX_BF16, static_cast<hipStream_t>(hip_stream), residual);
  if (e != hipSuccess) return hip_fail(e, "gbx_moe_combine launch");
  return GBX_OK;
}

int gbx_kv_append(const void* k, const void* v, void* kcache, void* vcache, const int64_t* pos, int64_t rows, int32_t capacity,
                  int32_t k_dim, int32_t v_dim, int32_t act_dtype, void* hip_stream) {
  if (rows == 0) return GBX_OK;
  if (!k || !v || !kcache || !vcache || !pos) return fail(GBX_ERR_INVALID, "null device pointer");
  if (rows < 0 || rows > 2147483647LL || capacity <= 0 || k_dim <= 0 || v_dim <= 0)
    return fail(GBX_ERR_INVALID, "bad shape rows=%lld capacity=%d k_dim=%d v_dim=%d", (long long)rows, capacity, k_dim, v_dim);
  if (act_dtype != GBX_F16 && act_dtype != GBX_BF16) return fail(GBX_ERR_INVALID, "act_dtype must be GBX_F16 or GBX_BF16");
  hipError_t e = gbx::launch_kv_append(k, v, kcache, vcache, reinterpret_cast<const long long*>(pos), static_cast<int>(rows), capacity,
                                       k_dim, v_dim, static_cast<hipStream_t>(hip_stream));
  if (e != hipSuccess) return hip_fail(e, "gbx_kv_append launch");
  return GBX_OK;
}

size_t gbx_logprobs_workspace_bytes(void) { return 128 * 8 + 16; }

int gbx_logprobs_argmax(const void* logits, void* logprobs, int64_t* token, void* workspace, int32_t vocab, int32_t act_dtype,
                        void* hip_stream) {
  if (!logits || !logprobs || !token || !workspace) return fail(GBX_ERR_INVALID, "null device pointer");
  if (vocab <= 0) return fail(GBX_ERR_INVALID, "vocab must be positive (got %d)", vocab);
  if (!aligned(workspace, 8)) return fail(GBX_ERR_INVALID, "workspace must be 8-byte aligned");
  if (act_dtype != GBX_F16 && act_dtype != GBX_BF16) return fail(GBX_ERR_INVALID, "act_dtype must be GBX_F16 or GBX_BF16");
  char* ws = static_cast<char*>(workspace);
  hipError_t e = gbx::launch_logprobs_argmax(logits, logprobs, ws, reinterpret_cast<unsigned long long*>(ws + 128 * 8),
                                             reinterpret_cast<long long*>(token), vocab, act_dtype == GBX_BF16,
                                             static_cast<hipStream_t>(hip_stream));
  if (e != hipSuccess) return hip_fail(e, "gbx_logprobs_argmax launch");
  return GBX_OK;
}

int gbx_logprobs_argmax_packed(const void* logits, void* logprobs, int64_t* token, void* workspace, int32_t vocab, int32_t act_dtype,
                               const uint32_t* status0, const uint32_t* status1, const uint32_t* status2, int64_t* packed,
                               void* hip_stream) {
  if (!logits || !logprobs || !token || !workspace || !packed) return fail(GBX_ERR_INVALID, "null device pointer");
  if (vocab <= 0) return fail(GBX_ERR_INVALID, "vocab must be positive (got %d)", vocab);
  if (!aligned(workspace, 8) || !aligned(packed, 8)) return fail(GBX_ERR_INVALID, "workspace and packed must be 8-byte aligned");
  if (act_dtype != GBX_F16 && act_dtype != GBX_BF16) return fail(GBX_ERR_INVALID, "act_dtype must be GBX_F16 or GBX_BF16");
  char* ws = static_cast<char*>(workspace);
  const unsigned* st[3] = {status0, status1, status2};
  hipError_t e = gbx::launch_logprobs_argmax(logits, logprobs, ws, reinterpret_cast<unsigned long long*>(ws + 128 * 8),
                                             reinterpret_cast<long long*>(token), vocab, act_dtype == GBX_BF16,
                                             static_cast<hipStream_t>(hip_stream), reinterpret_cast<long long*>(packed), st);
  if (e != hipSuccess) return hip_fail(e, "gbx_logprobs_argmax_packed launch");
  return GBX_OK;
}

int gbx_add_rmsnorm(void* h, const void* delta, const void* w, void* y, int64_t M, int32_t K, float eps, int32_t act_dtype,
                    void* hip_stream) {
  if (M == 0) return GBX_OK;
  if (!h || !delta || !w || !y) return fail(GBX_ERR_INVALID, "null device pointer");
  if (M < 0 || M > 2147483647LL || K <= 0) return fail(GBX_ERR_INVALID, "bad shape M=%lld K=%d", (long long)M, K);
  if (act_dtype != GBX_F16 && act_dtype != GBX_BF16) return fail(GBX_ERR_INVALID, "act_dtype must be GBX_F16 or GBX_BF16");
  hipError_t e = gbx::launch_add_rmsnorm(h, delta, w, y, static_cast<int>(M), K, eps, act_dtype == GBX_BF16, static_cast<hipStream_t>(hip_stream));
  if (e != hipSuccess) return hip_fail(e, "gbx_add_rmsnorm launch");
  return GBX_OK;
}

size_t gbx_qlinear_partials_bytes(int64_t M, int32_t N, int32_t K, int32_t bits, int32_t group_size) {
  if (check_shape(M, N, K, bits, group_size) != GBX_OK || M < 2 || M > 32 || (K % 128) != 0 || K < 512) return 0;
  const int slices = gbx::qgemm_batch_slices(static_cast<int>(M), N, K, bits, log2_group(group_size));
  return slices > 1 ? static_cast<size_t>(slices) * M * N * sizeof(float) : 0;
}

int gbx_qlinear_forward_partials(const void* x, const uint32_t* qweight, const void* scales, const void* zeros, void* workspace,
                                 size_t workspace_bytes, int32_t* slices_out, int64_t M, int32_t N, int32_t K, int32_t bits,
                                 int32_t group_size, int32_t act_dtype, void* hip_stream) {
  int rc = check_shape(M, N, K, bits, group_size);
  if (rc) return rc;
  if (act_dtype != GBX_F16 && act_dtype != GBX_BF16) return fail(GBX_ERR_INVALID, "act_dtype must be GBX_F16 or GBX_BF16");
  if (!x || !qweight || !scales || !zeros || !workspace || !slices_out) return fail(GBX_ERR_INVALID, "null pointer");
  if (!aligned(x, 16) || !aligned(qweight, 16) || !aligned(workspace, 16)) return fail(GBX_ERR_INVALID, "x, qweight and workspace must be 16-byte aligned");
  const int log2g = log2_group(group_size);
  const int slices = (M >= 2 && M <= 32 && (K % 128) == 0 && K >= 512) ? gbx::qgemm_batch_slices(static_cast<int>(M), N, K, bits, log2g) : 1;
  if (slices <= 1) return fail(GBX_ERR_UNSUPPORTED, "this shape does not split k across workgroups (M=%lld N=%d K=%d): use gbx_qlinear_forward", (long long)M, N, K);
  const size_t need = static_cast<size_t>(slices) * M * N * sizeof(float);      /* (not gbx_qlinear_workspace_bytes: short rows split here only) */
  if (workspace_bytes < need) return fail(GBX_ERR_INVALID, "workspace too small: %zu < %zu", workspace_bytes, need);
  hipError_t e = gbx::launch_qgemm_partials(x, qweight, scales, zeros, static_cast<float*>(workspace), static_cast<int>(M), N, K, bits, log2g,
                                            act_dtype == GBX_BF16, static_cast<hipStream_t>(hip_stream));
  if (e != hipSuccess) return hip_fail(e, "gbx_qlinear_forward_partials launch");
  *slices_out = slices;
  return GBX_OK;
}

int gbx_add_rmsnorm_slices(void* h, const float* slices_ws, int32_t slices, const void* w, void* y, int64_t M, int32_t K, float eps,
                           int32_t act_dtype, void* hip_stream) {
  if (M == 0) return GBX_OK;
  if (!h || !slices_ws || !w || !y) return fail(GBX_ERR_INVALID, "null device pointer");
  if (M < 0 || M > 65535 || K <= 0 || slices < 1) return fail(GBX_ERR_INVALID, "bad shape M=%lld K=%d slices=%d", (long long)M, K, slices);
  if ((K % 8) || K > 8192 || !aligned(h, 16) || !aligned(slices_ws, 16) || !aligned(w, 16) || !aligned(y, 16))
    return fail(GBX_ERR_UNSUPPORTED, "gbx_add_rmsnorm_slices needs K %% 8 == 0, K <= 8192 and 16-byte aligned tensors (K = %d)", K);
  if (act_dtype != GBX_F16 && act_dtype != GBX_BF16) return fail(GBX_ERR_INVALID, "act_dtype must be GBX_F16 or GBX_BF16");
  hipError_t e = gbx::launch_add_rmsnorm_slices(h, slices_ws, slices, w, y, static_cast<int>(M), K, eps, act_dtype == GBX_BF16, static_cast<hipStream_t>(hip_stream));
  if (e != hipSuccess) return hip_fail(e, "gbx_add_rmsnorm_slices launch");
  return GBX_OK;
}

size_t gbx_norm_state_bytes(void) { return gbx::slices_norm_state_bytes(); }

int gbx_add_rmsnorm_slices_xcd(void* h, const float* slices_ws, int32_t slices, const void* w, void* y, int64_t M, int32_t K, float eps,
                               int32_t act_dtype, void* state, void* hip_stream) {
  if (M == 0) return GBX_OK;
  if (!h || !slices_ws || !w || !y || !state) return fail(GBX_ERR_INVALID, "null device pointer");
  if (M < 0 || M > 65535 || K <= 0 || slices < 1) return fail(GBX_ERR_INVALID, "bad shape M=%lld K=%d slices=%d", (long long)M, K, slices);
  if (act_dtype != GBX_F16 && act_dtype != GBX_BF16) return fail(GBX_ERR_INVALID, "act_dtype must be GBX_F16 or GBX_BF16");
  if (!gbx::slices_norm_xcd_supported(static_cast<int>(M), K) || !aligned(h, 16) || !aligned(slices_ws, 16) || !aligned(w, 16) || !aligned(y, 16) ||
      !aligned(state, 64))
    return fail(GBX_ERR_UNSUPPORTED, "gbx_add_rmsnorm_slices_xcd needs K %% 512 == 0, K <= 4096, M <= 8 * (32 / (K / 512)) and aligned tensors (M = %lld, K = %d)",
                (long long)M, K);
  hipError_t e = gbx::launch_add_rmsnorm_slices_xcd(h, slices_ws, slices, w, y, static_cast<int>(M), K, eps, act_dtype == GBX_BF16, state,
                                                    static_cast<hipStream_t>(hip_stream));
  if (e != hipSuccess) return hip_fail(e, "gbx_add_rmsnorm_slices_xcd launch");
  return GBX_OK;
}

size_t gbx_dense_workspace_bytes(int64_t M, int32_t N, int32_t K) {
  if (M <= 0 || M > 2147483647LL || N <= 0 || K <= 0 || (N % 4) || !gbx::gemm_dense_supported(static_cast<int>(M), N, K)) return 0;
  return gbx::gemm_dense_workspace_bytes(static_cast<int>(M), N, K);
}

int gbx_dense_forward_ws(const void* x, const void* w, const void* bias, void* y, int64_t M, int32_t N, int32_t K, int32_t act_dtype,
                         void* workspace, size_t workspace_bytes, void* hip_stream) {
  if (act_dtype != GBX_F16 && act_dtype != GBX_BF16) return fail(GBX_ERR_INVALID, "act_dtype must be GBX_F16 or GBX_BF16");
  if (M < 0 || N <= 0 || K <= 0) return fail(GBX_ERR_INVALID, "bad shape M=%lld N=%d K=%d", (long long)M, N, K);
  if (M == 0) return GBX_OK;
  if (!x || !w || !y) return fail(GBX_ERR_INVALID, "null device pointer");
  if (!aligned(x, 16) || !aligned(w, 16) || !aligned(y, (N % 8) == 0 ? 16 : 2))
    return fail(GBX_ERR_INVALID, "x and w must be 16-byte aligned (and y too when N %% 8 == 0: the epilogue stores 16 bytes at a time)");
  if (M > 2147483647LL || !gbx::gemm_dense_supported(static_cast<int>(M), N, K))
    return fail(GBX_ERR_UNSUPPORTED, "dense matrix-core GEMM needs K %% 64 == 0 and M*K, N*K < 2^30 elements (M=%lld N=%d K=%d)", (long long)M, N, K);
  const size_t need = gbx_dense_workspace_bytes(M, N, K);
  float* part = (need && workspace && workspace_bytes >= need && aligned(workspace, 16) && aligned(y, 8)) ? static_cast<float*>(workspace) : nullptr;
  hipError_t e = gbx::launch_gemm_dense(x, w, bias, y, static_cast<int>(M), N, K, act_dtype == GBX_BF16, static_cast<hipStream_t>(hip_stream),
                                        false, part);
  if (e != hipSuccess) return hip_fail(e, "gbx_dense_forward_ws launch");
  return GBX_OK;
}

int gbx_dense_forward(const void* x, const void* w, const void* bias, void* y, int64_t M, int32_t N, int32_t K, int32_t act_dtype,
                      void* hip_stream) {
  if (act_dtype != GBX_F16 && act_dtype != GBX_BF16) return fail(GBX_ERR_INVALID, "act_dtype must be GBX_F16 or GBX_BF16");
  if (M < 0 || N <= 0 || K <= 0) return fail(GBX_ERR_INVALID, "bad shape M=%lld N=%d K=%d", (long long)M, N, K);
  if (M == 0) return GBX_OK;
  if (!x || !w || !y) return fail(GBX_ERR_INVALID, "null device pointer");
  if (!aligned(x, 16) || !aligned(w, 16) || !aligned(y, (N % 8) == 0 ? 16 : 2))
    return fail(GBX_ERR_INVALID, "x and w must be 16-byte aligned (and y too when N %% 8 == 0: the epilogue stores 16 bytes at a time)");
  if (M > 2147483647LL || !gbx::gemm_dense_supported(static_cast<int>(M), N, K))
    return fail(GBX_ERR_UNSUPPORTED, "dense matrix-core GEMM needs K %% 64 == 0 and M*K, N*K < 2^30 elements (M=%lld N=%d K=%d)", (long long)M, N, K);
  hipError_t e = gbx::launch_gemm_dense(x, w, bias, y, static_cast<int>(M), N, K, act_dtype == GBX_BF16, static_cast<hipStream_t>(hip_stream));
  if (e != hipSuccess) return hip_fail(e, "gbx_dense_forward launch");
  return GBX_OK;
}

int gbx_dense_swiglu_forward(const void* x, const void* w_gate_up, void* y, int64_t M, int32_t N, int32_t K, int32_t act_dtype,
                             void* hip_stream) {
  if (act_dtype != GBX_F16 && act_dtype != GBX_BF16) return fail(GBX_ERR_INVALID, "act_dtype must be GBX_F16 or GBX_BF16");
  if (M < 0 || N <= 0 || K <= 0) return fail(GBX_ERR_INVALID, "bad shape M=%lld N=%d K=%d", (long long)M, N, K);
  if (M == 0) return GBX_OK;
  if (!x || !w_gate_up || !y) return fail(GBX_ERR_INVALID, "null device pointer");
  if (!aligned(x, 16) || !aligned(w_gate_up, 16) || !aligned(y, 16)) return fail(GBX_ERR_INVALID, "x, w and y must be 16-byte aligned");
  if ((N % 128) || M > 2147483647LL || N > (1 << 29) || !gbx::gemm_dense_supported(static_cast<int>(M), 2 * N, K))
    return fail(GBX_ERR_UNSUPPORTED, "dense SwiGLU GEMM needs N %% 128 == 0, K %% 64 == 0 and M*K, 2*N*K < 2^30 elements (M=%lld N=%d K=%d)",
                (long long)M, N, K);
  hipError_t e = gbx::launch_gemm_dense(x, w_gate_up, nullptr, y, static_cast<int>(M), 2 * N, K, act_dtype == GBX_BF16,
                                        static_cast<hipStream_t>(hip_stream), true);
  if (e != hipSuccess) return hip_fail(e, "gbx_dense_swiglu_forward launch");
  return GBX_OK;
}

int gbx_swiglu(const void* gate, const void* up, void* y, int64_t n, int32_t act_dtype, void* hip_stream) {
  if (n == 0) return GBX_OK;
  if (!gate || !up || !y) return fail(GBX_ERR_INVALID, "null device pointer");
  if (n < 0 || (n % 8)) return fail(GBX_ERR_INVALID, "element count must be a non-negative multiple of 8 (got %lld)", (long long)n);
  if (!aligned(gate, 16) || !aligned(up, 16) || !aligned(y, 16)) return fail(GBX_ERR_INVALID, "gate, up and y must be 16-byte aligned");
  if (act_dtype != GBX_F16 && act_dtype != GBX_BF16) return fail(GBX_ERR_INVALID, "act_dtype must be GBX_F16 or GBX_BF16");
  hipError_t e = gbx::launch_swiglu(gate, up, y, n, act_dtype == GBX_BF16, static_cast<hipStream_t>(hip_stream));
  if (e != hipSuccess) return hip_fail(e, "gbx_swiglu launch");
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
  hipError_t e;
  if (out_dtype == sz_dtype && aligned(w, 16) && aligned(qweight, 4))      // 16-bit image: a 32-value unit per thread, 16-byte stores (same arithmetic)
    e = gbx::launch_dequant_units(qweight, scales, zeros, w, N, K, bits, log2_group(group_size), sz_dtype == GBX_BF16,
                                  static_cast<hipStream_t>(hip_stream));
  else
    e = gbx::launch_dequant(qweight, scales, zeros, w, N, K, bits, log2_group(group_size), sz_dtype == GBX_BF16,
                            out_dtype, static_cast<hipStream_t>(hip_stream));
  if (e != hipSuccess) return hip_fail(e, "gbx_dequantize launch");
  return GBX_OK;
}


/* ---- decode-block GEMV (dgemv_impl.h) ---- */
int gbx_dgemv_plan(int32_t N, int32_t K, int32_t bits, int32_t group_size, int32_t epilogue, int32_t with_norm,
                   int32_t* rows_per_batch, int32_t* ssq_partials) {
  int rc = check_shape(1, N, K, bits, group_size);
  if (rc) return rc;
  if (epilogue < GBX_DG_PLAIN || epilogue > GBX_DG_SWIGLU) return fail(GBX_ERR_INVALID, "epilogue must be GBX_DG_PLAIN / RESIDUAL / SWIGLU (got %d)", epilogue);
  gbx::DgPlan pl;
  if (!gbx::dg_plan(bits, N, K, group_size, epilogue, with_norm != 0, 0, &pl))
    return fail(GBX_ERR_UNSUPPORTED, "dgemv does not cover N=%d K=%d bits=%d group=%d epilogue=%d (use gbx_qlinear_fused_forward)", N, K, bits, group_size, epilogue);
  if (rows_per_batch) *rows_per_batch = pl.R;
  if (ssq_partials) *ssq_partials = epilogue == GBX_DG_SWIGLU ? 0 : pl.ssq_n;
  return GBX_OK;
}

int gbx_dgemv_pack(const void* scales0, const void* zeros0, const void* scales1, const void* zeros1, uint32_t* out,
                   int32_t N, int32_t G, int32_t rows_per_batch, void* hip_stream) {
  if (N <= 0 || G <= 0) return fail(GBX_ERR_INVALID, "bad shape N=%d G=%d", N, G);
  if (rows_per_batch != 2 && rows_per_batch != 4) return fail(GBX_ERR_INVALID, "rows_per_batch must be 2 or 4 (got %d)", rows_per_batch);
  if (!scales0 || !zeros0 || !out) return fail(GBX_ERR_INVALID, "null device pointer");
  if ((scales1 == nullptr) != (zeros1 == nullptr)) return fail(GBX_ERR_INVALID, "scales1 and zeros1 go together");
  const int ro = scales1 ? rows_per_batch / 2 : rows_per_batch;
  if (N % ro) return fail(GBX_ERR_INVALID, "N (%d) must be a multiple of %d", N, ro);
  if (!aligned(out, 16)) return fail(GBX_ERR_INVALID, "out must be 16-byte aligned");
  hipError_t e = gbx::launch_dg_pack(scales0, zeros0, scales1, zeros1, out, N, G, rows_per_batch, static_cast<hipStream_t>(hip_stream));
  if (e != hipSuccess) return hip_fail(e, "gbx_dgemv_pack launch");
  return GBX_OK;
}

static int dgemv_prepare(const gbx_dgemv* d, gbx::DgArgs* out, gbx::DgPlan* plo) {
  if (!d) return fail(GBX_ERR_INVALID, "null descriptor");
  if (d->struct_size != sizeof(gbx_dgemv)) return fail(GBX_ERR_INVALID, "gbx_dgemv.struct_size is %u, this library expects %zu (header / library mismatch)", d->struct_size, sizeof(gbx_dgemv));
  if (d->act_dtype != GBX_F16 && d->act_dtype != GBX_BF16) return fail(GBX_ERR_INVALID, "act_dtype must be GBX_F16 or GBX_BF16");
  if (d->epilogue < GBX_DG_PLAIN || d->epilogue > GBX_DG_SWIGLU) return fail(GBX_ERR_INVALID, "epilogue must be GBX_DG_PLAIN / RESIDUAL / SWIGLU (got %d)", d->epilogue);
  int rc = check_shape(1, d->N, d->K, d->bits, d->group_size);
  if (rc) return rc;
  if (!d->x || !d->qweight || !d->sz || !d->y) return fail(GBX_ERR_INVALID, "null device pointer (x / qweight / sz / y)");
  if (!aligned(d->x, 16) || !aligned(d->qweight, 16) || !aligned(d->sz, 16) || !aligned(d->y, 4))
    return fail(GBX_ERR_INVALID, "x / qweight / sz must be 16-byte aligned, y 4-byte aligned");
  if (d->epilogue == GBX_DG_SWIGLU && (!d->qweight_up || !aligned(d->qweight_up, 16))) return fail(GBX_ERR_INVALID, "SWIGLU needs qweight_up (16-byte aligned)");
  if (d->epilogue == GBX_DG_RESIDUAL && !d->residual) return fail(GBX_ERR_INVALID, "RESIDUAL needs residual");
  if (d->norm_weight && !aligned(d->norm_weight, 16)) return fail(GBX_ERR_INVALID, "norm_weight must be 16-byte aligned");
  if (d->ssq_in && (!d->norm_weight || !aligned(d->ssq_in, 16))) return fail(GBX_ERR_INVALID, "ssq_in goes with norm_weight and must be 16-byte aligned");
  if (d->ssq_out && d->epilogue == GBX_DG_SWIGLU) return fail(GBX_ERR_INVALID, "ssq_out is for PLAIN / RESIDUAL launches");
  if (d->x_acc && (!d->norm_weight || d->ssq_in || !aligned(d->x_acc, 16)))
    return fail(GBX_ERR_INVALID, "x_acc goes with norm_weight, without ssq_in (the statistic of the NEW x is reduced in the launch), 16-byte aligned");
  if (d->x_out && (!d->x_acc || !aligned(d->x_out, 8))) return fail(GBX_ERR_INVALID, "x_out goes with x_acc and must be 8-byte aligned");
  if (d->x_out && (d->x_out == d->x || d->x_out == d->y)) return fail(GBX_ERR_INVALID, "x_out must not alias x or y (every workgroup reads x)");
  if (d->zero_buf && (d->zero_n <= 0 || !aligned(d->zero_buf, 8))) return fail(GBX_ERR_INVALID, "zero_buf needs zero_n > 0 and 8-byte alignment");
  if (d->zero_buf && d->x_acc && reinterpret_cast<const void*>(d->zero_buf) == reinterpret_cast<const void*>(d->x_acc))
    return fail(GBX_ERR_INVALID, "a launch cannot clear the accumulator it reads");
  gbx::DgPlan pl;
  if (!gbx::dg_plan(d->bits, d->N, d->K, d->group_size, d->epilogue, d->norm_weight != nullptr, d->rows_per_batch, &pl))
    return fail(GBX_ERR_UNSUPPORTED, "dgemv does not cover N=%d K=%d bits=%d group=%d epilogue=%d rows_per_batch=%d", d->N, d->K, d->bits, d->group_size, d->epilogue, d->rows_per_batch);
  if (d->norm_weight && pl.KS != 1) return fail(GBX_ERR_UNSUPPORTED, "fused RMSNorm needs K <= 128 lane slots (K = %d at %d bits); call gbx_rmsnorm first", d->K, d->bits);
  if (d->ssq_out && pl.ssq_n == 0) return fail(GBX_ERR_UNSUPPORTED, "ssq_out: this launch geometry cannot leave its partial statistic in %d slots (gbx_dgemv_plan reports 0)", GBX_DG_SSQ_SLOTS);
  static_assert(GBX_DG_SSQ_SLOTS == gbx::DG_SSQ_SLOTS, "public constant out of sync");
  gbx::DgArgs a{};
  a.x = static_cast<const unsigned short*>(d->x);
  a.qw = d->qweight;
  a.qw2 = d->epilogue == GBX_DG_SWIGLU ? d->qweight_up : d->qweight;
  a.sz = d->sz;
  a.y = static_cast<unsigned short*>(d->y);
  a.residual = static_cast<const unsigned short*>(d->residual);
  a.norm_w = static_cast<const unsigned short*>(d->norm_weight);
  a.ssq_in = d->ssq_in;
  a.ssq_out = d->ssq_out;
  a.x_acc = reinterpret_cast<const long long*>(d->x_acc);
  a.x_out = static_cast<unsigned short*>(d->x_out);
  a.zero_buf = reinterpret_cast<unsigned long long*>(d->zero_buf);
  a.zero_n = d->zero_buf ? d->zero_n : 0;
  a.N = d->N; a.K = d->K; a.log2g = log2_group(d->group_size);
  a.eps = d->norm_eps;
  static const int chain_floor = gbx::env_switch("GBX_CHAIN_FLOOR");
  a.null_rows = chain_floor == 2 ? 1 : 0;
  static const int long_prio = gbx::env_switch("GBX_DG_LONG_PRIO");
  a.long_prio = (long_prio && pl.bpo > 1 && pl.apw < (pl.KS == 1 ? pl.NW : pl.NW / 4)) ? 1 : 0;
  *out = a;
  *plo = pl;
  return GBX_OK;
}

int gbx_dgemv_kernel_id(const gbx_dgemv* d, int32_t* kernel, int32_t* waves_per_workgroup, int32_t* rows_per_batch) {
  gbx::DgArgs a;
  gbx::DgPlan pl;
  int rc = dgemv_prepare(d, &a, &pl);
  if (rc) return rc;
  const bool mx = d->bits == 2 && gbx::dg_mx_wanted(a, pl, d->epilogue, d->act_dtype == GBX_BF16);
  if (kernel) *kernel = mx ? 1 : 0;
  if (waves_per_workgroup) *waves_per_workgroup = pl.NW;
  if (rows_per_batch) *rows_per_batch = pl.R;
  return GBX_OK;
}

int gbx_dgemv_forward(const gbx_dgemv* d, void* hip_stream) {
  gbx::DgArgs a;
  gbx::DgPlan pl;
  int rc = dgemv_prepare(d, &a, &pl);
  if (rc) return rc;
#ifdef GBX_STAMPS
  a.stamps = gbx::stamp_region();
#endif
  hipStream_t stream = static_cast<hipStream_t>(hip_stream);
  const bool bf16 = d->act_dtype == GBX_BF16;
  hipError_t e;
  if (d->bits == 2 && gbx::dg_mx_wanted(a, pl, d->epilogue, bf16)) {
    e = gbx::launch_dgemv_mx(a, pl, d->epilogue, stream);
    if (e != hipSuccess) return hip_fail(e, "gbx_dgemv_forward (matrix-core form) launch");
    return GBX_OK;
  }
  switch (d->bits) {
    case 2: e = gbx::launch_dgemv<2>(a, pl, d->epilogue, bf16, stream); break;
    case 3: e = gbx::launch_dgemv<3>(a, pl, d->epilogue, bf16, stream); break;
    default: e = gbx::launch_dgemv<4>(a, pl, d->epilogue, bf16, stream); break;
  }
  if (e != hipSuccess) return hip_fail(e, "gbx_dgemv_forward launch");
  return GBX_OK;
}

/* ---- the device check of every launch with XCD-local hand-offs (xcd_layout.hip): static checks + a census launch, once per device ---- */
static int xcd_layout_gate(const char* who) {
  hipError_t e = gbx::xcd_census();
  if (e != hipSuccess) return hip_fail(e, who);
  const char* why = gbx::xcd_layout_reason();
  if (why[0]) return fail(GBX_ERR_UNSUPPORTED, "%s: this device does not take the launches with XCD-local hand-offs: %s", who, why);
  return GBX_OK;
}

int gbx_xcd_layout_check(void) { return xcd_layout_gate("gbx_xcd_layout_check"); }

/* ---- fused attention launch (dattn_impl.h) ---- */
int gbx_dattn_plan(int32_t D, int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int32_t qkv_bits, int32_t qkv_group_size,
                   int32_t o_bits, int32_t o_group_size) {
  gbx::DattnPlan pl;
  if (int rc = xcd_layout_gate("gbx_dattn_plan")) return rc;
  if (!gbx::dattn_plan(D, n_heads, n_kv_heads, head_dim, qkv_bits, qkv_group_size, o_bits, o_group_size, &pl))
    return fail(GBX_ERR_UNSUPPORTED, "the fused attention launch does not cover D=%d heads=%d/%d head_dim=%d q|k|v %d bits g%d, o %d bits g%d",
                D, n_heads, n_kv_heads, head_dim, qkv_bits, qkv_group_size, o_bits, o_group_size);
  return GBX_OK;
}

size_t gbx_dattn_state_bytes(void) {
  return static_cast<size_t>(gbx::DA_GROUPS) * gbx::DA_HAND_STRIDE * 8 + static_cast<size_t>(gbx::DA_STATE_WORDS) * 4 +
         static_cast<size_t>(gbx::DA_PART_GRANULES) * 8;
}

int gbx_dattn_forward(const gbx_dattn* d, void* hip_stream) {
  if (!d) return fail(GBX_ERR_INVALID, "null descriptor");
  if (d->struct_size != sizeof(gbx_dattn)) return fail(GBX_ERR_INVALID, "gbx_dattn.struct_size is %u, this library expects %zu (header / library mismatch)", d->struct_size, sizeof(gbx_dattn));
  if (d->act_dtype != GBX_F16 && d->act_dtype != GBX_BF16) return fail(GBX_ERR_INVALID, "act_dtype must be GBX_F16 or GBX_BF16");
  if (log2_group(d->qkv_group_size) < 0 || log2_group(d->o_group_size) < 0) return fail(GBX_ERR_INVALID, "The group size value must be 32, 64 or 128.");
  gbx::DattnPlan pl;
  if (!gbx::dattn_plan(d->D, d->n_heads, 8, 128, d->qkv_bits, d->qkv_group_size, d->o_bits, d->o_group_size, &pl))
    return fail(GBX_ERR_UNSUPPORTED, "the fused attention launch does not cover D=%d heads=%d q|k|v %d bits g%d, o %d bits g%d (gbx_dattn_plan)",
                d->D, d->n_heads, d->qkv_bits, d->qkv_group_size, d->o_bits, d->o_group_size);
  if (!d->x || !d->norm_weight || !d->qkv_qweight || !d->qkv_sz || !d->kcache || !d->vcache || !d->inv_freq || !d->pos || !d->o_qweight ||
      !d->o_scales || !d->o_zeros || !d->acc || !d->state)
    return fail(GBX_ERR_INVALID, "null device pointer");
  if (!aligned(d->x, 16) || !aligned(d->norm_weight, 16) || !aligned(d->qkv_qweight, 16) || !aligned(d->qkv_sz, 16) || !aligned(d->kcache, 16) ||
      !aligned(d->vcache, 16) || !aligned(d->o_qweight, 16) || !aligned(d->acc, 16) || !aligned(d->state, 64) || (d->ssq_in && !aligned(d->ssq_in, 16)))
    return fail(GBX_ERR_INVALID, "x / norm_weight / qweights / sz / caches / acc / ssq_in must be 16-byte aligned, state 64-byte aligned");
  if (d->capacity <= 0) return fail(GBX_ERR_INVALID, "capacity must be positive");
  gbx::DattnArgs a{};
  a.x = static_cast<const unsigned short*>(d->x);
  a.norm_w = static_cast<const unsigned short*>(d->norm_weight);
  a.ssq_in = d->ssq_in;
  a.qkv_qw = d->qkv_qweight;
  a.qkv_sz = d->qkv_sz;
  a.kcache = static_cast<unsigned short*>(d->kcache);
  a.vcache = static_cast<unsigned short*>(d->vcache);
  a.inv_freq = d->inv_freq;
  a.pos = reinterpret_cast<const long long*>(d->pos);
  a.o_qw = d->o_qweight;
  a.o_s = static_cast<const unsigned short*>(d->o_scales);
  a.o_z = static_cast<const unsigned short*>(d->o_zeros);
  a.acc = reinterpret_cast<unsigned long long*>(d->acc);
  a.hand = static_cast<unsigned long long*>(d->state);
  a.state = reinterpret_cast<unsigned*>(static_cast<unsigned char*>(d->state) + static_cast<size_t>(gbx::DA_GROUPS) * gbx::DA_HAND_STRIDE * 8);
  a.attn_out = static_cast<unsigned short*>(d->attn_out);
  a.part = reinterpret_cast<unsigned long long*>(static_cast<unsigned char*>(d->state) + static_cast<size_t>(gbx::DA_GROUPS) * gbx::DA_HAND_STRIDE * 8 +
                                    static_cast<size_t>(gbx::DA_STATE_WORDS) * 4);
  {
    static const int coop_min = gbx::env_switch("GBX_DATTN_COOP_MIN");
    a.coop_min = coop_min;
    static const int spin_log2 = gbx::env_switch("GBX_DATTN_SPIN_LOG2");
    a.spin_limit = 1u << (spin_log2 < 0 ? 0 : (spin_log2 > 30 ? 30 : spin_log2));
  }
  if (d->x_acc && (d->ssq_in || !aligned(d->x_acc, 16))) return fail(GBX_ERR_INVALID, "x_acc goes without ssq_in and must be 16-byte aligned");
  if (d->x_out && (!d->x_acc || d->x_out == d->x || !aligned(d->x_out, 4))) return fail(GBX_ERR_INVALID, "x_out goes with x_acc and must not alias x");
  if (d->zero_buf && (d->zero_n <= 0 || !aligned(d->zero_buf, 8) || reinterpret_cast<const void*>(d->zero_buf) == reinterpret_cast<const void*>(d->x_acc) ||
                      reinterpret_cast<const void*>(d->zero_buf) == reinterpret_cast<const void*>(d->acc)))
    return fail(GBX_ERR_INVALID, "zero_buf needs zero_n > 0, 8-byte alignment, and must be neither x_acc nor acc of this launch");
  a.x_acc = reinterpret_cast<const long long*>(d->x_acc);
  a.x_out = static_cast<unsigned short*>(d->x_out);
  a.zero_buf = reinterpret_cast<unsigned long long*>(d->zero_buf);
  a.zero_n = d->zero_buf ? d->zero_n : 0;
  a.D = d->D; a.rep = d->n_heads / 8; a.capacity = d->capacity; a.bpw = pl.bpw;
  a.qkv_log2g = log2_group(d->qkv_group_size); a.o_bits = d->o_bits; a.o_log2g = log2_group(d->o_group_size);
  a.traditional = d->traditional;
  a.eps = d->norm_eps; a.rope_scale = d->rope_scale; a.sm_scale = d->sm_scale;
#ifdef GBX_STAMPS
  a.stamps = gbx::stamp_region();
  {
    static const char* ps = getenv("GBX_PROBE_SRC");
    static unsigned long long launch = 0;
    if (ps) {
      // a fresh 32 MB region of the pool per launch (24 regions: nothing of it is still in the L2s or the Infinity Cache)
      a.probe_src = reinterpret_cast<const unsigned char*>(strtoull(ps, nullptr, 0)) + (launch++ % 24) * (static_cast<size_t>(32) << 20);
      auto geti = [](const char* n, int dflt) { const char* v = getenv(n); return v ? atoi(v) : dflt; };
      a.probe_mode = geti("GBX_PROBE_MODE", 1);
      a.probe_kb_p = geti("GBX_PROBE_KB_P", 112);
      a.probe_kb_a = geti("GBX_PROBE_KB_A", 90);
      a.probe_kb_a0 = geti("GBX_PROBE_KB_A0", 32);
    }
  }
#endif
  hipError_t e = gbx::launch_dattn(a, pl, d->qkv_bits, d->act_dtype == GBX_BF16, static_cast<hipStream_t>(hip_stream));
  if (e != hipSuccess) return hip_fail(e, "gbx_dattn_forward launch");
  return GBX_OK;
}

int gbx_dmlp_plan(int32_t D, int32_t I, int32_t gate_up_bits, int32_t gate_up_group_size, int32_t down_bits, int32_t down_group_size,
                  int32_t act_dtype) {
  int bpw, a4;
  if (int rc = xcd_layout_gate("gbx_dmlp_plan")) return rc;
  if (!gbx::dmlp_plan(D, I, gate_up_bits, gate_up_group_size, down_bits, down_group_size, act_dtype == GBX_BF16, &bpw, &a4) &&
      !gbx::dmlpv_plan(D, I, gate_up_bits, gate_up_group_size, down_bits, down_group_size, act_dtype == GBX_BF16, &bpw, &a4))
    return fail(GBX_ERR_UNSUPPORTED, "the fused MLP launch does not cover D=%d I=%d gate|up %d bits g%d, down %d bits g%d, dtype %d", D, I,
                gate_up_bits, gate_up_group_size, down_bits, down_group_size, act_dtype);
  return GBX_OK;
}

int gbx_repack_planes(const uint32_t* qweight, uint32_t* out, int64_t N, int32_t K, int32_t bits, void* hip_stream) {
  if (bits != 3) return fail(GBX_ERR_UNSUPPORTED, "gbx_repack_planes: 3-bit rows (got %d bits)", bits);
  if (N < 0 || K <= 0 || (K % 32) != 0) return fail(GBX_ERR_INVALID, "bad shape N=%lld K=%d", (long long)N, K);
  if (N == 0) return GBX_OK;
  if (!qweight || !out || !aligned(qweight, 4) || !aligned(out, 4)) return fail(GBX_ERR_INVALID, "null or misaligned device pointer");
  if (qweight == out) return fail(GBX_ERR_INVALID, "out must not alias qweight (the reference tensor stays what it was)");
  hipError_t e = gbx::launch_repack_planes3(qweight, out, static_cast<size_t>(N) * (K / 32), static_cast<hipStream_t>(hip_stream));
  if (e != hipSuccess) return hip_fail(e, "gbx_repack_planes launch");
  return GBX_OK;
}

// 3-bit rows take the matrix-core launch in the plane layout (GBX_DMLP_MX3, default 1), else the vector form in the reference layout
static bool dmlp_wants_planes(int D, int I, int bg, int gg, int bd, int gd, bool bf16) {
  static const bool mx3 = gbx::env_switch("GBX_DMLP_MX3") != 0;
  int bpw, a4;
  return mx3 && bg == 3 && bd == 3 && gbx::dmlp_plan(D, I, bg, gg, bd, gd, bf16, &bpw, &a4);
}

int gbx_dmlp_weight_layout(int32_t D, int32_t I, int32_t gate_up_bits, int32_t gate_up_group_size, int32_t down_bits, int32_t down_group_size,
                           int32_t act_dtype) {
  if (int rc = gbx_dmlp_plan(D, I, gate_up_bits, gate_up_group_size, down_bits, down_group_size, act_dtype)) return rc < 0 ? rc : -rc;
  return dmlp_wants_planes(D, I, gate_up_bits, gate_up_group_size, down_bits, down_group_size, act_dtype == GBX_BF16) ? GBX_LAYOUT_PLANES
                                                                                                                       : GBX_LAYOUT_REFERENCE;
}

int gbx_dmlp_forward(const gbx_dmlp* d, void* hip_stream) {
  if (!d) return fail(GBX_ERR_INVALID, "null descriptor");
  if (d->struct_size != sizeof(gbx_dmlp)) return fail(GBX_ERR_INVALID, "gbx_dmlp.struct_size is %u, this library expects %zu (header / library mismatch)", d->struct_size, sizeof(gbx_dmlp));
  const int bg = d->gate_up_bits ? d->gate_up_bits : 2, gg = d->gate_up_group_size ? d->gate_up_group_size : 128;
  const int bd = d->down_bits ? d->down_bits : 2, gd = d->down_group_size ? d->down_group_size : 128;
  int bpw, a4;
  if (d->weight_layout != GBX_LAYOUT_REFERENCE && d->weight_layout != GBX_LAYOUT_PLANES) return fail(GBX_ERR_INVALID, "gbx_dmlp.weight_layout is %d", d->weight_layout);
  const bool planes = d->weight_layout == GBX_LAYOUT_PLANES;
  if (planes && !(bg == 3 && bd == 3)) return fail(GBX_ERR_INVALID, "GBX_LAYOUT_PLANES is the layout of 3-bit rows (gbx_repack_planes)");
  // 3-bit rows: the matrix-core launch reads planes only, the vector launch the reference layout only
  const bool mx = (bg != 3 || planes) && gbx::dmlp_plan(d->D, d->I, bg, gg, bd, gd, d->act_dtype == GBX_BF16, &bpw, &a4);
  if (planes && !mx) return fail(GBX_ERR_UNSUPPORTED, "no matrix-core fused MLP launch for these 3-bit shapes: pass the reference layout (gbx_dmlp_weight_layout)");
  if (!mx && !gbx::dmlpv_plan(d->D, d->I, bg, gg, bd, gd, d->act_dtype == GBX_BF16, &bpw, &a4))
    return fail(GBX_ERR_UNSUPPORTED, "the fused MLP launch does not cover D=%d I=%d gate|up %d bits g%d, down %d bits g%d, dtype %d (gbx_dmlp_plan)", d->D, d->I,
                bg, gg, bd, gd, d->act_dtype);
  if (!d->x || !d->x_acc || !d->x_out || !d->norm_weight || !d->gate_qweight || !d->up_qweight || !d->gate_up_sz || !d->down_qweight || !d->down_sz ||
      !d->acc_out || !d->state)
    return fail(GBX_ERR_INVALID, "null device pointer");
  if (!aligned(d->x, 16) || !aligned(d->x_acc, 16) || !aligned(d->x_out, 8) || !aligned(d->norm_weight, 16) || !aligned(d->gate_qweight, 16) ||
      !aligned(d->up_qweight, 16) || !aligned(d->gate_up_sz, 16) || !aligned(d->down_qweight, 16) || !aligned(d->down_sz, 16) ||
      !aligned(d->acc_out, 16) || !aligned(d->state, 64))
    return fail(GBX_ERR_INVALID, "pointers must be 16-byte aligned (state: 64)");
  if (d->x_out == d->x) return fail(GBX_ERR_INVALID, "x_out must not alias x (every workgroup reads x)");
  if (reinterpret_cast<const void*>(d->acc_out) == reinterpret_cast<const void*>(d->x_acc)) return fail(GBX_ERR_INVALID, "acc_out must not be x_acc");
  if (d->zero_buf && (d->zero_n <= 0 || !aligned(d->zero_buf, 8) || reinterpret_cast<const void*>(d->zero_buf) == reinterpret_cast<const void*>(d->x_acc) ||
                      reinterpret_cast<const void*>(d->zero_buf) == reinterpret_cast<const void*>(d->acc_out)))
    return fail(GBX_ERR_INVALID, "zero_buf needs zero_n > 0, 8-byte alignment, and must be neither x_acc nor acc_out of this launch");
  gbx::DmlpArgs a{};
  a.x = static_cast<const unsigned short*>(d->x);
  a.x_acc = reinterpret_cast<const long long*>(d->x_acc);
  a.x_out = static_cast<unsigned short*>(d->x_out);
  a.norm_w = static_cast<const unsigned short*>(d->norm_weight);
  a.gate_qw = d->gate_qweight; a.up_qw = d->up_qweight; a.gu_sz = d->gate_up_sz;
  a.down_qw = d->down_qweight; a.down_sz = d->down_sz;
  a.acc_out = reinterpret_cast<unsigned long long*>(d->acc_out);
  a.zero_buf = reinterpret_cast<unsigned long long*>(d->zero_buf);
  a.zero_n = d->zero_buf ? d->zero_n : 0;
  a.hand = static_cast<unsigned long long*>(d->state);
  a.state = reinterpret_cast<unsigned*>(static_cast<unsigned char*>(d->state) + static_cast<size_t>(gbx::DA_GROUPS) * gbx::DA_HAND_STRIDE * 8);
  a.act_out = static_cast<unsigned short*>(d->act_out);
  a.K = d->D; a.I = d->I; a.D = d->D; a.bpw = bpw; a.a4 = a4;
  a.gu_log2g = log2_group(gg); a.dn_log2g = log2_group(gd);
  {
    static const int spin_log2 = gbx::env_switch("GBX_DATTN_SPIN_LOG2");
    a.spin_limit = 1u << (spin_log2 < 0 ? 0 : (spin_log2 > 30 ? 30 : spin_log2));
  }
  a.eps = d->norm_eps;
#ifdef GBX_STAMPS
  a.stamps = gbx::stamp_region();
#endif
  hipError_t e = mx ? gbx::launch_dmlp(a, bg, static_cast<hipStream_t>(hip_stream)) : gbx::launch_dmlpv(a, bg, static_cast<hipStream_t>(hip_stream));
  if (e != hipSuccess) return hip_fail(e, "gbx_dmlp_forward launch");
  return GBX_OK;
}

int gbx_acc_residual(const void* x, int64_t* acc, void* out, int32_t D, int32_t act_dtype, int64_t* zero0, int64_t* zero1, int64_t* zero2, void* hip_stream) {
  if (!x || !acc || !out || D <= 0) return fail(GBX_ERR_INVALID, "null pointer or bad D");
  if (act_dtype != GBX_F16 && act_dtype != GBX_BF16) return fail(GBX_ERR_INVALID, "act_dtype must be GBX_F16 or GBX_BF16");
  hipError_t e = gbx::launch_acc_finish(x, acc, out, D, act_dtype == GBX_BF16, zero0, zero1, zero2, static_cast<hipStream_t>(hip_stream));
  if (e != hipSuccess) return hip_fail(e, "gbx_acc_residual launch");
  return GBX_OK;
}

int gbx_dblock_plan(int32_t D, int32_t I, int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int32_t qkv_bits, int32_t qkv_group_size,
                    int32_t o_bits, int32_t o_group_size, int32_t gate_up_bits, int32_t gate_up_group_size, int32_t down_bits,
                    int32_t down_group_size, int32_t act_dtype) {
  if (int rc = xcd_layout_gate("gbx_dblock_plan")) return rc;
  if (!gbx::dblock_plan(D, I, n_heads, n_kv_heads, head_dim, qkv_bits, qkv_group_size, o_bits, o_group_size, gate_up_bits, gate_up_group_size,
                        down_bits, down_group_size, act_dtype == GBX_BF16))
    return fail(GBX_ERR_UNSUPPORTED, "the one-launch block covers 2 bits, group 128, bf16, D = 4096, 32 / 8 heads of 128, I = 14336 (got D=%d I=%d heads=%d/%d hd=%d "
                "bits %d/%d/%d/%d groups %d/%d/%d/%d dtype %d)", D, I, n_heads, n_kv_heads, head_dim, qkv_bits, o_bits, gate_up_bits, down_bits,
                qkv_group_size, o_group_size, gate_up_group_size, down_group_size, act_dtype);
  return GBX_OK;
}

size_t gbx_dblock_state_bytes(void) { return gbx::dblock_state_bytes(); }
int32_t gbx_dblock_max_context(void) { return gbx::dblock_max_ctx(); }

int gbx_dblock_forward(const gbx_dblock* d, void* hip_stream) {
  if (!d) return fail(GBX_ERR_INVALID, "null descriptor");
  if (d->struct_size != sizeof(gbx_dblock)) return fail(GBX_ERR_INVALID, "gbx_dblock.struct_size is %u, this library expects %zu (header / library mismatch)", d->struct_size, sizeof(gbx_dblock));
  if (d->act_dtype != GBX_BF16) return fail(GBX_ERR_UNSUPPORTED, "the one-launch block is bf16");
  if (!gbx::xcd_layout_ok()) return fail(GBX_ERR_UNSUPPORTED, "gbx_dblock_forward: %s (gbx_dblock_plan)", gbx::xcd_layout_reason());
  if (!d->x || !d->norm_weight || !d->qkv_qweight || !d->qkv_sz || !d->kcache || !d->vcache || !d->inv_freq || !d->pos || !d->o_qweight || !d->o_scales ||
      !d->o_zeros || !d->acc_o || !d->norm2_weight || !d->gate_qweight || !d->up_qweight || !d->gate_up_sz || !d->down_qweight || !d->down_sz || !d->acc_d ||
      !d->x2_out || !d->state)
    return fail(GBX_ERR_INVALID, "null device pointer");
  const void* al16[] = {d->x, d->norm_weight, d->qkv_qweight, d->qkv_sz, d->kcache, d->vcache, d->o_qweight, d->acc_o, d->norm2_weight, d->gate_qweight,
                        d->up_qweight, d->gate_up_sz, d->down_qweight, d->down_sz, d->acc_d, d->x2_out, d->ssq_in, d->x_acc, d->x_out};
  for (const void* p : al16)
    if (p && !aligned(p, 16)) return fail(GBX_ERR_INVALID, "every tensor of gbx_dblock must be 16-byte aligned");
  if (!aligned(d->state, 64)) return fail(GBX_ERR_INVALID, "state must be 64-byte aligned");
  if (d->capacity <= 0) return fail(GBX_ERR_INVALID, "capacity must be positive");
  if (d->x_acc && d->ssq_in) return fail(GBX_ERR_INVALID, "x_acc goes without ssq_in");
  if (d->x_acc && (!d->x_out || d->x_out == d->x)) return fail(GBX_ERR_INVALID, "x_acc needs x_out (the leaders read the stream back), not aliasing x");
  if (!d->x_acc && d->x_out) return fail(GBX_ERR_INVALID, "x_out goes with x_acc");
  // (x2_out MAY be x: every read of x sits in front of a workgroup's ARRIVE, the one write of x2_out behind the chip-wide release)
  if (d->x2_out == d->x_out) return fail(GBX_ERR_INVALID, "x2_out must not alias x_out (the leaders read x_out back while one of them writes x2_out)");
  if (reinterpret_cast<const void*>(d->acc_o) == reinterpret_cast<const void*>(d->acc_d) || reinterpret_cast<const void*>(d->x_acc) == reinterpret_cast<const void*>(d->acc_o) ||
      reinterpret_cast<const void*>(d->x_acc) == reinterpret_cast<const void*>(d->acc_d))
    return fail(GBX_ERR_INVALID, "acc_o, acc_d and x_acc are three different accumulators");
  const uint64_t* zb[2] = {d->zero_buf0, d->zero_buf1};
  const int32_t zn[2] = {d->zero_n0, d->zero_n1};
  for (int i = 0; i < 2; ++i)
    // (acc_d itself MAY be a zero_buf: it is cleared with write-through stores before this workgroup's ARRIVE and added to only behind the
    //  chip-wide release, which waits for every workgroup's ARRIVE -- the two-launch block clears it in its first launch the same way)
    if (zb[i] && (zn[i] <= 0 || !aligned(zb[i], 8) || reinterpret_cast<const void*>(zb[i]) == reinterpret_cast<const void*>(d->x_acc) ||
                  reinterpret_cast<const void*>(zb[i]) == reinterpret_cast<const void*>(d->acc_o)))
      return fail(GBX_ERR_INVALID, "a zero_buf needs zero_n > 0, 8-byte alignment, and must be neither x_acc nor acc_o of this launch");
  gbx::DblockArgs a{};
  a.x = static_cast<const unsigned short*>(d->x);
  a.norm_w = static_cast<const unsigned short*>(d->norm_weight);
  a.ssq_in = d->ssq_in;
  a.qkv_qw = d->qkv_qweight; a.qkv_sz = d->qkv_sz;
  a.kcache = static_cast<unsigned short*>(d->kcache); a.vcache = static_cast<unsigned short*>(d->vcache);
  a.inv_freq = d->inv_freq;
  a.pos = reinterpret_cast<const long long*>(d->pos);
  a.o_qw = d->o_qweight; a.o_s = static_cast<const unsigned short*>(d->o_scales); a.o_z = static_cast<const unsigned short*>(d->o_zeros);
  a.acc_o = reinterpret_cast<unsigned long long*>(d->acc_o);
  a.x_acc = reinterpret_cast<const long long*>(d->x_acc);
  a.x_out = static_cast<unsigned short*>(d->x_out);
  a.attn_out = static_cast<unsigned short*>(d->attn_out);
  a.norm2_w = static_cast<const unsigned short*>(d->norm2_weight);
  a.gate_qw = d->gate_qweight; a.up_qw = d->up_qweight; a.gu_sz = d->gate_up_sz; a.down_qw = d->down_qweight; a.down_sz = d->down_sz;
  a.acc_d = reinterpret_cast<unsigned long long*>(d->acc_d);
  a.x2_out = static_cast<unsigned short*>(d->x2_out);
  a.act_out = static_cast<unsigned short*>(d->act_out);
  a.hand = static_cast<unsigned long long*>(d->state);
  a.state = reinterpret_cast<unsigned*>(static_cast<unsigned char*>(d->state) + gbx::dblock_status_offset() - 8 * 16 * 4);
  a.zero0 = reinterpret_cast<unsigned long long*>(d->zero_buf0); a.zero_n0 = d->zero_buf0 ? d->zero_n0 : 0;
  a.zero1 = reinterpret_cast<unsigned long long*>(d->zero_buf1); a.zero_n1 = d->zero_buf1 ? d->zero_n1 : 0;
  a.capacity = d->capacity; a.traditional = d->traditional; a.max_ctx = gbx::dblock_max_ctx();
  {
    static const int spin_log2 = gbx::env_switch("GBX_DATTN_SPIN_LOG2");
    a.spin_limit = 1u << (spin_log2 < 0 ? 0 : (spin_log2 > 30 ? 30 : spin_log2));
  }
  a.eps1 = d->norm_eps; a.eps2 = d->norm2_eps; a.rope_scale = d->rope_scale; a.sm_scale = d->sm_scale;
#ifdef GBX_STAMPS
  a.stamps = gbx::stamp_region();
#endif
  hipError_t e = gbx::launch_dblock(a, static_cast<hipStream_t>(hip_stream));
  if (e != hipSuccess) return hip_fail(e, "gbx_dblock_forward launch");
  return GBX_OK;
}

int gbx_dblock_status(const void* state, int32_t* code) {
  if (!state || !code) return fail(GBX_ERR_INVALID, "null pointer");
  unsigned v = 0;
  hipError_t e = hipMemcpy(&v, static_cast<const unsigned char*>(state) + gbx::dblock_status_offset(), 4, hipMemcpyDeviceToHost);
  if (e != hipSuccess) return hip_fail(e, "gbx_dblock_status");
  *code = static_cast<int32_t>(v);
  return GBX_OK;
}

int gbx_dattn_status(const void* state, int32_t* code) {
  if (!state || !code) return fail(GBX_ERR_INVALID, "null pointer");
  unsigned v = 0;
  const unsigned char* p = static_cast<const unsigned char*>(state) + static_cast<size_t>(gbx::DA_GROUPS) * gbx::DA_HAND_STRIDE * 8 + 8 * 16 * 4;
  hipError_t e = hipMemcpy(&v, p, 4, hipMemcpyDeviceToHost);
  if (e != hipSuccess) return hip_fail(e, "gbx_dattn_status");
  *code = static_cast<int32_t>(v);
  return GBX_OK;
}

}  // extern "C"
