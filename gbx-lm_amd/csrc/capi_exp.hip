// extern "C" entry points of libgbx_experimental.so (include/gbx_experimental.h): the measured-negative / opt-in kernels of rounds
// 2-3, kept as tested entry points OUTSIDE the product ABI -- the streamed multi-op GEMV engine (gemv_stream.hip), the integer
// matrix-core GEMV on re-tiled weights (gemv_mfma.hip), the FP4 x FP8 scaled-MFMA GEMVs (gemv_mx.hip, gemv_mxd.hip), the merged
// attention + o_proj launch (attn_oproj.hip) and the fused launch whose activation vector is residual stream + fixed-point
// accumulator (gemv_kernel MODE 3).  Built with -DGBX_EXPERIMENTAL -Dgbx=gbxx (own namespace: no symbol shared with the product
// library).  Same conventions as capi.hip: no allocation, no synchronisation, graph-capturable.
#include "gbx_qlinear.h"
#include "gbx_experimental.h"
#include "gemv_impl.h"
#include "gemv_stream.h"
#include "kernels.h"
#include "capi_util.h"

#ifdef GBX_STAMPS
namespace gbx {
unsigned long long* stamp_region() { return nullptr; }
}  // namespace gbx
#endif

extern "C" {

static int check_tiled(int32_t N, int32_t K, int32_t bits, int32_t group_size) {
  int rc = check_shape(1, N, K, bits, group_size);
  if (rc) return rc;
  if (bits != 2 || group_size != 128 || (K % 1024) != 0 || K > 16384)
    return fail(GBX_ERR_UNSUPPORTED, "tiled decode GEMV covers 2-bit, group_size 128, K %% 1024 == 0, K <= 16384 (got bits=%d g=%d K=%d)", bits, group_size, K);
  return GBX_OK;
}

int gbx_exp_version(void) { return 1100; }
const char* gbx_exp_last_error(void) { return g_err; }

int gbx_exp_qlinear_fused_forward(const gbx_exp_qlinear_fused* d, void* hip_stream) {
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
  if (d->x_acc) {
    if (!d->sz || nslots > 128 || d->group_size < 64) return fail(GBX_ERR_UNSUPPORTED, "x_acc needs the packed companion (sz), one pass per wave over K (K <= %d) and group_size >= 64", 128 * vpl);
    if (!aligned(d->x_acc, 16) || (d->x_out && !aligned(d->x_out, 16))) return fail(GBX_ERR_INVALID, "x_acc / x_out must be 16-byte aligned");
    if (d->x_out == d->x) return fail(GBX_ERR_INVALID, "x_out must not alias x (the other workgroups are still reading it)");
  } else if (d->x_out) return fail(GBX_ERR_INVALID, "x_out without x_acc");
  if (d->zero_buf && (d->zero_n < 0 || !aligned(d->zero_buf, 8))) return fail(GBX_ERR_INVALID, "bad zero_buf / zero_n");
  a.x_acc = reinterpret_cast<const long long*>(d->x_acc);
  a.x_out = static_cast<unsigned short*>(d->x_out);
  a.zero_buf = reinterpret_cast<long long*>(d->zero_buf);
  a.zero_n = d->zero_buf ? d->zero_n : 0;
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
  if (e != hipSuccess) return hip_fail(e, "gbx_exp_qlinear_fused_forward launch");
  return GBX_OK;
}

int gbx_stream_encode(const gbx_exp_qlinear_fused* descs, int32_t n, void* out, size_t out_bytes) {
  static_assert(sizeof(gbx::StOp) == GBX_STREAM_OP_BYTES, "public op size out of sync with the device format");
  if (!descs || !out || n < 1) return fail(GBX_ERR_INVALID, "null descriptor table / output or n < 1");
  if (out_bytes < static_cast<size_t>(n) * sizeof(gbx::StOp)) return fail(GBX_ERR_INVALID, "output buffer too small: %zu < %zu", out_bytes, static_cast<size_t>(n) * sizeof(gbx::StOp));
  gbx::StOp* ops = static_cast<gbx::StOp*>(out);
  for (int k = 0; k < n; ++k) {
    const gbx_exp_qlinear_fused* d = descs + k;
    if (d->nseg < 1 || d->nseg > 3) return fail(GBX_ERR_INVALID, "op %d: nseg must be 1..3 (got %d)", k, d->nseg);
    if (d->x_acc || d->x_out || d->zero_buf) return fail(GBX_ERR_UNSUPPORTED, "op %d: x_acc / x_out / zero_buf run on gbx_qlinear_fused_forward", k);
    if (d->act_dtype != GBX_F16 && d->act_dtype != GBX_BF16) return fail(GBX_ERR_INVALID, "op %d: act_dtype must be GBX_F16 or GBX_BF16", k);
    int ntot = 0;
    for (int i = 0; i < d->nseg; ++i) {
      int rc = check_shape(1, d->N[i], d->K, d->bits, d->group_size);
      if (rc) return rc;
      if (!d->qweight[i] || !aligned(d->qweight[i], 16)) return fail(GBX_ERR_INVALID, "op %d: qweight of segment %d missing or not 16-byte aligned", k, i);
      if (d->N[i] & 1) return fail(GBX_ERR_UNSUPPORTED, "op %d: the streamed path needs even row counts (segment %d has %d)", k, i, d->N[i]);
      ntot += d->N[i];
    }
    if (!d->sz || !aligned(d->sz, 16)) return fail(GBX_ERR_UNSUPPORTED, "op %d: the streamed path needs the packed scale|zero companion (gbx_pack_scales_zeros), 16-byte aligned", k);
    if ((d->K / d->group_size) & 1) return fail(GBX_ERR_UNSUPPORTED, "op %d: the streamed path needs an even number of groups per row (K / group_size = %d)", k, d->K / d->group_size);
    if (!d->x || !aligned(d->x, 16) || !d->y[0]) return fail(GBX_ERR_INVALID, "op %d: x / y missing or x not 16-byte aligned", k);
    if (d->swiglu && (d->nseg != 2 || d->N[0] != d->N[1])) return fail(GBX_ERR_INVALID, "op %d: swiglu needs two equal-sized segments", k);
    if (!d->swiglu) for (int i = 0; i < d->nseg; ++i) if (!d->y[i]) return fail(GBX_ERR_INVALID, "op %d: null y in segment %d", k, i);
    if (d->residual && (d->nseg != 1 || d->swiglu)) return fail(GBX_ERR_INVALID, "op %d: residual needs a single plain segment", k);
    if (d->norm_weight && !aligned(d->norm_weight, 16)) return fail(GBX_ERR_INVALID, "op %d: norm_weight must be 16-byte aligned", k);
    const int K = d->K, bits = d->bits, vpl = bits == 2 ? 64 : 32;
    const int nslots = K / vpl;
    if ((K % 128) != 0 || nslots > 512) return fail(GBX_ERR_UNSUPPORTED, "op %d: needs K %% 128 == 0 and K <= %d (K = %d)", k, 512 * vpl, K);
    if (bits == 2 && d->group_size == 32) return fail(GBX_ERR_UNSUPPORTED, "op %d: 2-bit group_size 32 runs on gbx_qlinear_fused_forward", k);
    if ((d->norm_weight || d->swiglu) && nslots > 128) return fail(GBX_ERR_UNSUPPORTED, "op %d: fused RMSNorm / SwiGLU need K <= %d at %d bits", k, 128 * vpl, bits);
    gbx::StOp o{};
    o.type = gbx::ST_GEMV;
    o.bits = bits;
    o.ks = nslots <= 128 ? 1 : 4;
    o.P = (nslots <= 64 || (nslots > 128 && nslots <= 256)) ? 1 : 2;
    o.K = K;
    o.gshift = log2_group(d->group_size) - 5;
    o.nseg = d->nseg;
    o.flags = (d->swiglu ? gbx::ST_F_SWIGLU : 0) | (d->norm_weight ? gbx::ST_F_NORM : 0) | (d->residual ? gbx::ST_F_RESID : 0);
    o.n0 = d->N[0];
    o.n01 = d->N[0] + (d->nseg > 1 ? d->N[1] : 0);
    o.N = d->swiglu ? d->N[0] : ntot;
    o.nbatch = d->swiglu ? d->N[0] : ntot / 2;
    o.W4 = K / 32 * bits * 4;
    o.G8 = (K / d->group_size) * 8;
    o.nslots = nslots;
    const int owners = 1 * (o.ks == 1 ? gbx::ST_NW : gbx::ST_NW / 4);   /* per CU; the launch has >= 1 CU ... */
    (void)owners;
    for (int i = 0; i < 3; ++i) {
      const int s = i < d->nseg ? i : 0;
      o.qw[i] = d->qweight[s];
      o.y[i] = static_cast<unsigned short*>(d->y[s]);
    }
    o.sz = d->sz;
    o.x = static_cast<const unsigned short*>(d->x);
    o.norm_w = static_cast<const unsigned short*>(d->norm_weight);
    o.resid = static_cast<const unsigned short*>(d->residual);
    o.eps = d->norm_eps;
    ops[k] = o;
  }
  return GBX_OK;
}

int gbx_stream_run(const void* device_program, const void* host_program, int32_t begin, int32_t end, int32_t act_dtype, void* hip_stream) {
  if (!device_program || !aligned(device_program, 16)) return fail(GBX_ERR_INVALID, "device program missing or not 16-byte aligned");
  if (!host_program) return fail(GBX_ERR_INVALID, "host copy of the program missing (its first op of the range rides in the kernel arguments)");
  if (begin < 0 || end <= begin) return fail(GBX_ERR_INVALID, "empty op range [%d, %d)", begin, end);
  if (act_dtype != GBX_F16 && act_dtype != GBX_BF16) return fail(GBX_ERR_INVALID, "act_dtype must be GBX_F16 or GBX_BF16");
  hipError_t e = gbx::launch_stream(device_program, static_cast<const gbx::StOp*>(host_program) + begin, begin, end, act_dtype == GBX_BF16, static_cast<hipStream_t>(hip_stream));
  if (e != hipSuccess) return hip_fail(e, "gbx_stream_run launch");
  return GBX_OK;
}

static int check_batch_mx(int64_t M, int32_t N, int32_t K) {
  if (M < 1 || M > 32 || N <= 0 || K <= 0) return fail(GBX_ERR_INVALID, "bad shape M=%lld N=%d K=%d", (long long)M, N, K);
  if ((N % 128) || (K % 1024) || K > 16384)
    return fail(GBX_ERR_UNSUPPORTED, "the FP4 x FP8 batch GEMM covers N %% 128 == 0, K %% 1024 == 0, K <= 16384 (2 bits, group 128, bf16; got N=%d K=%d)", N, K);
  return GBX_OK;
}

size_t gbx_exp_batch_mx_image_bytes(int32_t K) { return (K > 0 && K % 128 == 0) ? gbx::batch_mx_image_bytes(K) : 0; }
size_t gbx_exp_batch_mx_aux_bytes(int32_t K) { return (K > 0 && K % 128 == 0) ? gbx::batch_mx_aux_bytes(K) : 0; }

int gbx_exp_batch_mx_image(const void* x, int64_t M, int32_t K, void* image, void* aux, void* hip_stream) {
  int rc = check_batch_mx(M, 128, K);
  if (rc) return rc;
  if (!x || !image || !aux || !aligned(image, 16) || !aligned(aux, 16)) return fail(GBX_ERR_INVALID, "null or misaligned device pointer");
  hipError_t e = gbx::launch_batch_mx_image(x, static_cast<int>(M), K, image, aux, static_cast<hipStream_t>(hip_stream));
  if (e != hipSuccess) return hip_fail(e, "gbx_exp_batch_mx_image launch");
  return GBX_OK;
}

int gbx_exp_batch_mx_slices(int32_t N, int32_t K) {
  if (N <= 0 || K <= 0 || (N % 128) || (K % 1024)) return 0;
  return gbx::batch_mx_ksplit(N, K);
}

int gbx_exp_batch_mx_forward(const void* image, const void* aux, const uint32_t* qweight, const void* scales, const void* zeros, void* y,
                             float* slices_ws, int64_t M, int32_t N, int32_t K, void* hip_stream) {
  int rc = check_batch_mx(M, N, K);
  if (rc) return rc;
  if (!image || !aux || !qweight || !scales || !zeros) return fail(GBX_ERR_INVALID, "null device pointer");
  if (!aligned(image, 16) || !aligned(aux, 16) || !aligned(qweight, 16)) return fail(GBX_ERR_INVALID, "image, aux and qweight must be 16-byte aligned");
  const int ks = gbx::batch_mx_ksplit(N, K);
  if (ks < 1) return fail(GBX_ERR_UNSUPPORTED, "no k split of K = %d leaves an even number (<= 32) of groups per workgroup", K);
  if (ks > 1 ? !slices_ws : !y) return fail(GBX_ERR_INVALID, "this shape %s", ks > 1 ? "splits k: slices_ws [gbx_exp_batch_mx_slices][M][N] fp32 needed" : "needs y");
  hipError_t e = gbx::launch_batch_mx(image, aux, qweight, scales, zeros, y, slices_ws, ks, static_cast<int>(M), N, K, static_cast<hipStream_t>(hip_stream));
  if (e != hipSuccess) return hip_fail(e, "gbx_exp_batch_mx_forward launch");
  return GBX_OK;
}

int gbx_qlinear_prepack_tiles(const uint32_t* qweight, const void* scales, const void* zeros,
                              const uint32_t* qweight_up, const void* scales_up, const void* zeros_up,
                              uint32_t* wt, uint32_t* szt, int32_t N, int32_t K, int32_t bits, int32_t group_size,
                              void* hip_stream) {
  int rc = check_tiled(N, K, bits, group_size);
  if (rc) return rc;
  if (N == 0) return GBX_OK;
  if (!qweight || !scales || !zeros || !wt || !szt) return fail(GBX_ERR_INVALID, "null device pointer");
  if ((qweight_up != nullptr) != (scales_up != nullptr) || (qweight_up != nullptr) != (zeros_up != nullptr))
    return fail(GBX_ERR_INVALID, "qweight_up, scales_up and zeros_up go together");
  if (!aligned(wt, 16) || !aligned(szt, 16)) return fail(GBX_ERR_INVALID, "tiled buffers must be 16-byte aligned");
  hipError_t e = gbx::launch_prepack_tiles(qweight, scales, zeros, qweight_up, scales_up, zeros_up, wt, szt, N, K,
                                           static_cast<hipStream_t>(hip_stream));
  if (e != hipSuccess) return hip_fail(e, "gbx_qlinear_prepack_tiles launch");
  return GBX_OK;
}

int gbx_gemv_tiled(const void* x, const uint32_t* wt, const uint32_t* szt, void* y, int32_t N, int32_t K, int32_t bits,
                   int32_t group_size, const void* norm_weight, float norm_eps, const void* residual, int32_t swiglu,
                   int32_t act_dtype, void* hip_stream) {
  int rc = check_tiled(N, K, bits, group_size);
  if (rc) return rc;
  if (act_dtype != GBX_F16 && act_dtype != GBX_BF16) return fail(GBX_ERR_INVALID, "act_dtype must be GBX_F16 or GBX_BF16");
  if (N == 0) return GBX_OK;
  if (!x || !wt || !szt || !y) return fail(GBX_ERR_INVALID, "null device pointer");
  if (!aligned(x, 16) || !aligned(wt, 16) || !aligned(szt, 16)) return fail(GBX_ERR_INVALID, "x and the tiled buffers must be 16-byte aligned");
  if (norm_weight && !aligned(norm_weight, 4)) return fail(GBX_ERR_INVALID, "norm_weight must be 4-byte aligned");
  if (swiglu && residual) return fail(GBX_ERR_INVALID, "swiglu and residual exclude each other");
  hipError_t e = gbx::launch_gemv_mfma(x, wt, szt, y, N, K, norm_weight, norm_eps, residual, swiglu ? 1 : 0,
                                       act_dtype == GBX_BF16, static_cast<hipStream_t>(hip_stream));
  if (e != hipSuccess) return hip_fail(e, "gbx_gemv_tiled launch");
  return GBX_OK;
}

int gbx_attn_oproj_supported(int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int32_t D, int32_t bits, int32_t group_size) {
  return gbx::attn_oproj_supported(n_heads, n_kv_heads, head_dim, D, bits, group_size, 1) ? 1 : 0;
}

int gbx_attn_oproj_decode(const void* q, const void* k, const void* v, void* kcache, void* vcache, const float* inv_freq,
                          const int64_t* pos, const uint32_t* o_qweight, const void* o_scales, const void* o_zeros,
                          int64_t* acc, int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int32_t capacity, int32_t D,
                          int32_t bits, int32_t group_size, float rope_scale, float sm_scale, int32_t traditional,
                          int32_t act_dtype, void* hip_stream) {
  if (!q || !k || !v || !kcache || !vcache || !inv_freq || !pos || !o_qweight || !o_scales || !o_zeros || !acc)
    return fail(GBX_ERR_INVALID, "null device pointer");
  if (act_dtype != GBX_F16 && act_dtype != GBX_BF16) return fail(GBX_ERR_INVALID, "act_dtype must be GBX_F16 or GBX_BF16");
  if (capacity < 1) return fail(GBX_ERR_INVALID, "capacity must be positive");
  if (!gbx::attn_oproj_supported(n_heads, n_kv_heads, head_dim, D, bits, group_size, capacity))
    return fail(GBX_ERR_UNSUPPORTED, "attention + o_proj launch needs head_dim 128, n_heads == 4 n_kv_heads, D %% 128 == 0, bits 2/3/4, group 64/128 "
                "(got heads %d/%d x %d, D %d, %d bits, group %d)", n_heads, n_kv_heads, head_dim, D, bits, group_size);
  if (!aligned(o_qweight, 16) || !aligned(kcache, 16) || !aligned(vcache, 16) || !aligned(acc, 8))
    return fail(GBX_ERR_INVALID, "o_qweight / caches must be 16-byte aligned, acc 8-byte aligned");
  hipError_t e = gbx::launch_attn_oproj(q, k, v, kcache, vcache, inv_freq, reinterpret_cast<const long long*>(pos), o_qweight, o_scales,
                                        o_zeros, reinterpret_cast<long long*>(acc), n_heads, n_kv_heads, head_dim, capacity, D, bits,
                                        log2_group(group_size), rope_scale, sm_scale, traditional, act_dtype == GBX_BF16,
                                        static_cast<hipStream_t>(hip_stream));
  if (e != hipSuccess) return hip_fail(e, "gbx_attn_oproj_decode launch");
  return GBX_OK;
}

}  // extern "C"
