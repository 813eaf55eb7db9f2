// Host-side launcher declarations shared between the kernel TUs and the C-ABI (capi.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <atomic>

namespace gbx {

// hipFuncSetAttribute (the opt-in to more than 64 KB of dynamic LDS) is a PER-DEVICE setting: a process that drives several GPUs
// must make it once on each, and two host threads may race to be first.  `pending()` is true until `done()` has been called on the
// calling thread's current device; setting the attribute twice is harmless, launching before it is set is not.
struct DeviceOnce {
  std::atomic<unsigned long long> seen{0};
  unsigned long long bit() const { int d = 0; (void)hipGetDevice(&d); return 1ull << (d & 63); }
  bool pending() const { return (seen.load(std::memory_order_acquire) & bit()) == 0; }
  void done() { seen.fetch_or(bit(), std::memory_order_release); }
};

struct GemvArgs;

#ifdef GBX_STAMPS
// diagnostic build only: the next per-launch region of the s_memrealtime stamp buffer (GBX_STAMP_PTR / GBX_STAMP_SLOTS), capi.hip
unsigned long long* stamp_region();
constexpr int STAMP_REGION_WORDS = 4096 * 16;
#endif

template <int BITS>
hipError_t launch_gemv(const GemvArgs& a, int M, bool bf16, hipStream_t stream);
// gemv_mx.hip: the 2-bit decode GEMV on the block-scaled matrix cores (FP4 weights in place x FP8 activation limbs)
bool gemv_mx_eligible(const GemvArgs& a, int M);
bool gemv_mx_shape_ok(const GemvArgs& a);
hipError_t launch_gemv_mx(const GemvArgs& a, bool bf16, hipStream_t s);
// gemv_mxd.hip: the same launches with ONE packed row per scaled-MFMA operand (16 consecutive groups, diagonal of the result):
// 2 bits, group 128, bf16, K = 2048 / 4096, packed scale|zero companion
bool gemv_mxd_eligible(const GemvArgs& a, int M, bool bf16);
hipError_t launch_gemv_mxd(const GemvArgs& a, hipStream_t s);

hipError_t launch_unpack(const uint32_t* qw, int32_t* q, int N, int K, int bits, hipStream_t s);
hipError_t launch_dequant(const uint32_t* qw, const void* sc, const void* ze, void* w, int N, int K, int bits,
                          int log2g, bool sz_bf16, int out, hipStream_t s);
hipError_t launch_dequant_units(const uint32_t* qw, const void* sc, const void* ze, void* w, int N, int K, int bits, int log2g,
                                bool bf16, hipStream_t s, int blk = 0, int which = 0);
hipError_t launch_qlinear_generic(const void* x, const uint32_t* qw, const void* sc, const void* ze, const void* bias,
                                  void* y, int M, int N, int K, int bits, int log2g, bool bf16, hipStream_t s);

hipError_t launch_qgemm_mfma(const void* x, const uint32_t* qw, const void* sc, const void* ze, const void* bias, void* y,
                             int M, int N, int K, int bits, int log2g, bool bf16, hipStream_t s, float* ws = nullptr);
size_t qgemm_batch_workspace_bytes(int M, int N, int K, int bits, int log2g);
bool gemm_dense_supported(int M, int N, int K);
hipError_t launch_gemm_dense(const void* x, const void* w, const void* bias, void* y, int M, int N, int K, bool bf16, hipStream_t s,
                             bool swiglu = false, float* part = nullptr);
// k split of the dense GEMM for few-tile shapes (fp32 slices in caller scratch + the summing launch): slices (1: none) and scratch bytes
int gemm_dense_ksplit(int M, int N, int K);
size_t gemm_dense_workspace_bytes(int M, int N, int K);
// long-prompt SwiGLU: gate and up dequantised into ONE dense matrix (blocks of 128 rows alternating), one GEMM over its 2 N rows
// whose epilogue writes silu(gate) * up [M, N]; wdense: 2 * N * K * 2 bytes
hipError_t launch_qgemm_dense_swiglu(const void* x, const uint32_t* gq, const void* gs, const void* gz, const uint32_t* uq, const void* us,
                                     const void* uz, void* y, int M, int N, int K, int bits, int log2g, bool bf16, hipStream_t s,
                                     void* wdense);
constexpr int GBX_DENSE_MIN_M = 4096;
// Prompt-sized calls with caller scratch: dequantise once + the 256 x 256 dense GEMM wherever its tiles fill the chip --
// always from 4096 rows, from 1024 rows when the 256 x 256 tiles make at least ~0.7 of a round over the 256 CUs (measured,
// 4 bits: M 3072 x N 4096: 161 -> 96 us, M 1024 x N 14336: 172 -> 96, M 2048 x N 14336: 268 -> 190; below that the fused tiles
// with their finer shapes / k split win, e.g. M 2048 x N 4096: 89 vs 96)
// ... and, with its k range split over workgroups (gemm_dense_ksplit, fp32 slices + a summing launch), the few-tile shapes of a
// 1024..4095-row chunk too (M 1024 x N 4096: 83 -> 45 us, K 14336: 256 -> 104)
int gemm_dense_ksplit(int M, int N, int K);
inline bool dense_path_wanted(long long M, int N, int K) {
  if ((K % 128) != 0) return false;
  if (M >= GBX_DENSE_MIN_M) return true;
  if (M < 1024) return false;
  return ((M + 255) / 256) * ((N + 255) / 256) >= 176 || (N % 4 == 0 && gemm_dense_ksplit(static_cast<int>(M), N, K) > 1);
}
hipError_t launch_rope_kv_write(const void* k, const void* v, void* kcache, void* vcache, const float* inv_freq, long long pos0,
                                long long B, int H, int L, int D, long long ksb, long long ksh, long long ksl, long long vsb,
                                long long vsh, long long vsl, int capacity, int t0, float scale, int traditional, bool bf16,
                                hipStream_t s);
// attn_prefill.hip: causal attention of L query rows at positions offset .. offset + L - 1 against S keys (strides in elements)
hipError_t launch_attn_prefill(const void* q, const void* k, const void* v, void* out, int B, int Hq, int Hkv, int L, int S, int D,
                               long long qs_b, long long qs_h, long long qs_l, long long ks_b, long long ks_h, long long vs_b,
                               long long vs_h, int offset, float sm_scale, bool bf16, hipStream_t s);
hipError_t launch_qgemm_dense(const void* x, const uint32_t* qw, const void* sc, const void* ze, const void* bias, void* y, int M,
                              int N, int K, int bits, int log2g, bool bf16, hipStream_t s, void* wdense, float* part = nullptr);
size_t qgemm_prefill_workspace_bytes(int M, int N, int K);
int prefill_tile_id(int M, int N, int K, bool has_ws);
hipError_t launch_qgemm_skinny(const void* x, const uint32_t* qw, const void* sc, const void* ze, const void* bias, void* y,
                               int M, int N, int K, int bits, int log2g, bool bf16, hipStream_t s, float* ws = nullptr);
hipError_t launch_qgemm_skinny_multi(const void* x, int nseg, const uint32_t* const* qw, const void* const* sc, const void* const* ze,
                                     const void* const* bias, void* const* y, const int* n, int M, int K, int bits, int log2g,
                                     bool bf16, hipStream_t s);
hipError_t launch_qgemm_swiglu(const void* x, const uint32_t* gq, const void* gs, const void* gz, const uint32_t* uq, const void* us,
                               const void* uz, void* y, int M, int N, int K, int bits, int log2g, bool bf16, hipStream_t s);
hipError_t launch_qgemm_partials(const void* x, const uint32_t* qw, const void* sc, const void* ze, float* ws, int M, int N, int K,
                                 int bits, int log2g, bool bf16, hipStream_t s);
int qgemm_batch_slices(int M, int N, int K, int bits, int log2g);
hipError_t launch_add_rmsnorm_slices(void* h, const float* ws, int slices, const void* w, void* y, int M, int K, float eps, bool bf16, hipStream_t s);
// the same spread over the chip, the row statistic exchanged inside one XCD's L2 (decode_ops.hip); state: slices_norm_state_bytes() zeroed bytes
size_t slices_norm_state_bytes();
bool slices_norm_xcd_supported(int M, int K);
// xcd_layout.hip: what the launches with XCD-local hand-offs require of the device (256 CUs in one partition, no CU mask, consecutive
// workgroups dealt round-robin over the 8 XCDs -- the last one MEASURED by a census launch).  reason: "" = fine.
const char* xcd_layout_reason();
bool xcd_layout_ok();
hipError_t xcd_census();          // synchronous, once per device; never call it inside a stream capture
hipError_t launch_add_rmsnorm_slices_xcd(void* h, const float* ws, int slices, const void* w, void* y, int M, int K, float eps, bool bf16,
                                         void* state, hipStream_t s);
hipError_t launch_prepack_tiles(const uint32_t* qw, const void* sc, const void* ze, const uint32_t* qw2, const void* sc2,
                                const void* ze2, uint32_t* wt, uint32_t* szt, int N, int K, hipStream_t s);
hipError_t launch_gemv_mfma(const void* x, const uint32_t* wt, const uint32_t* szt, void* y, int N, int K, const void* norm_w,
                            float eps, const void* residual, int swiglu, bool bf16, hipStream_t s);
hipError_t launch_attn_decode(const void* q, const void* k, const void* v, void* kcache, void* vcache, const float* inv_freq,
                              const long long* pos, void* out, int batch, int n_heads, int n_kv, int head_dim, int capacity,
                              float rope_scale, float sm_scale, int traditional, bool bf16, hipStream_t s,
                              const void* q_norm_w = nullptr, const void* k_norm_w = nullptr, float norm_eps = 0.f,
                              const void* q_bias = nullptr, const void* k_bias = nullptr, const void* v_bias = nullptr,
                              float* ws = nullptr, const float* qkv_ws = nullptr, int qkv_slices = 0);
int attn_splits(int batch, int n_heads, int n_kv, int capacity);
// p2p_allreduce.hip: one-shot push all-reduce of a short 16-bit vector over peer-mapped inboxes
hipError_t launch_p2p_allreduce(void* x, int n, int rank, int world, void* const* inbox, int cap, void* state, bool bf16, hipStream_t s);
// attn_oproj.hip: attention + o_proj slices of one decode token in one launch (short caches)
bool attn_oproj_supported(int n_heads, int n_kv, int head_dim, int D, int bits, int group_size, int capacity);
hipError_t launch_attn_oproj(const void* q, const void* k, const void* v, void* kcache, void* vcache, const float* inv_freq,
                             const long long* pos, const uint32_t* qweight, const void* scales, const void* zeros, long long* acc,
                             int n_heads, int n_kv, int head_dim, int capacity, int D, int bits, int log2g, float rope_scale,
                             float sm_scale, int traditional, bool bf16, hipStream_t s);
size_t attn_workspace_bytes(int batch, int n_heads, int n_kv, int head_dim, int capacity);
hipError_t launch_dense_gemv(const void* x, const void* w, void* y, const void* norm_w, float eps, int N, int K, void* best,
                             bool bf16, hipStream_t s, int index_offset = 0);
hipError_t launch_embed_next(void* best, void* tok_io, void* pos, const void* table, void* h, int D, int vocab, int mode,
                             int advance, bool bf16, hipStream_t s, int nkeys = 64, void* also_clear = nullptr);
hipError_t launch_repack_planes3(const uint32_t* in, uint32_t* out, size_t nblocks, hipStream_t s);
hipError_t launch_pack_sz(const void* s0, const void* z0, const void* s1, const void* z1, uint32_t* out, int N, int G,
                          hipStream_t s);
hipError_t launch_rope(const void* x, void* y, const float* inv_freq, const long long* posp, long long offset, long long B,
                       int H, int L, int D, int dims, long long sb, long long sh, long long sl, float scale, float mscale,
                       int traditional, bool bf16, hipStream_t s);
hipError_t launch_rope_append(void* q, const void* k, const void* v, void* kcache, void* vcache, const float* inv_freq,
                              const long long* pos, int batch, int n_heads, int n_kv, int hd, int capacity, float rope_scale,
                              int traditional, bool bf16, hipStream_t s);
hipError_t launch_eminf(const void* logits, float* cur, float* nxt, float* out_entropy, int V, int in_mode, float alpha,
                        float threshold, int num_steps, hipStream_t s);
hipError_t launch_attn_static(const void* q, const void* kcache, const void* vcache, const long long* pos, void* out, int batch,
                              int n_heads, int n_kv, int head_dim, int capacity, float sm_scale, bool bf16, hipStream_t s);
hipError_t launch_moe_route(const void* logits, int* inds, void* scores, int T, int E, int k, int mode, int renorm, bool bf16,
                            hipStream_t s);
hipError_t launch_moe_combine(const void* y, const void* scores, void* out, int T, int k, int D, bool bf16, hipStream_t s,
                              const void* residual = nullptr);
hipError_t launch_kv_append(const void* k, const void* v, void* kc, void* vc, const long long* pos, int rows, int cap, int dk, int dv,
                            hipStream_t s);
hipError_t launch_logprobs_argmax(const void* logits, void* logprobs, void* partials, unsigned long long* key, long long* token,
                                  int V, bool bf16, hipStream_t s, long long* packed = nullptr, const unsigned* const* status = nullptr);
hipError_t launch_add_rmsnorm(void* h, const void* d, const void* w, void* y, int M, int K, float eps, bool bf16, hipStream_t s);
hipError_t launch_swiglu(const void* g, const void* u, void* y, long long n, bool bf16, hipStream_t s);
hipError_t launch_stream(const void* dev_prog, const void* host_first_op, int op_begin, int op_end, bool bf16, hipStream_t s);
hipError_t launch_rmsnorm(const void* x, const void* w, void* y, int M, int K, float eps, bool bf16, hipStream_t s);

// batch_mx.hip (experimental library): the M <= 32 batch GEMM on the FP4 x FP8 matrix cores (2 bits, group 128, bf16)
size_t batch_mx_image_bytes(int K);
size_t batch_mx_aux_bytes(int K);
hipError_t launch_batch_mx_image(const void* x, int M, int K, void* img, void* aux, hipStream_t s);
hipError_t launch_batch_mx(const void* img, const void* aux, const uint32_t* qw, const void* sc, const void* ze, void* y, float* ws, int ksplit,
                           int M, int N, int K, hipStream_t s);
int batch_mx_ksplit(int N, int K);

}  // namespace gbx
