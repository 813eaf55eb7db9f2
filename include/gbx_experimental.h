/* C-ABI of libgbx_experimental.so: the opt-in / measured-negative kernels of the GBA QuantizedLinear decode path on gfx950.
 *
 * NOT what a gbx-lm maintainer binds (that is include/gbx_qlinear.h): these are the lab-notebook kernels of rounds 2-3, kept
 * as tested entry points because their measurements are cited in DESIGN.md -- the streamed multi-op GEMV engine, the integer
 * matrix-core GEMV on re-tiled weights, the FP4 x FP8 scaled-MFMA GEMVs, the merged attention + o_proj launch.  Every one of them
 * replaces the same reference call as its product counterpart, mx.quantized_matmul at M == 1
 * (gbx_lm/models/quantized_linear_gba.py:195-203; call sites gbx_lm/models/qllama.py:76,96,115), with the same result contract.
 * Same conventions as gbx_qlinear.h (plain pointers, caller-owned memory, stream-ordered, error codes + gbx_exp_last_error).
 */
#ifndef GBX_EXPERIMENTAL_H
#define GBX_EXPERIMENTAL_H
#include "gbx_qlinear.h"
#ifdef __cplusplus
extern "C" {
#endif

int gbx_exp_version(void);
const char* gbx_exp_last_error(void);

/* gbx_qlinear_fused plus the fields of the launch that follows gbx_attn_oproj_decode.  gbx_exp_qlinear_fused_forward is
 * gbx_qlinear_fused_forward with them, and with the opt-in kernels selected by the environment of the process:
 *   GBX_GEMV_MXD=1 (every eligible 2-bit launch) / =2 (gate|up only): one packed row per scaled-MFMA operand (gemv_mxd.hip);
 *   GBX_GEMV_MX=1: loader / consumer waves + LDS-DMA row staging on the scaled MFMA (gemv_mx.hip). */
typedef struct gbx_exp_qlinear_fused {
  const void* x;
  int32_t K, bits, group_size, act_dtype;
  int32_t nseg;
  const uint32_t* qweight[3];
  const void* scales[3];
  const void* zeros[3];
  void* y[3];
  int32_t N[3];
  const void* norm_weight;
  float norm_eps;
  const void* residual;
  int32_t swiglu;
  const uint32_t* sz;
  /* optional (needs sz, K <= one row pass, group_size >= 64): the activation vector is formed in the launch's prologue as
   * x <- round(x + round(x_acc * 2^-32)) -- x_acc = the int64 fixed-point o_proj sums gbx_attn_oproj_decode leaves behind, i.e. the
   * residual add `h = x + r` of qllama.py:138 -- and, with x_out != NULL (never x itself), the new x is also written there once. */
  const int64_t* x_acc;
  void* x_out;
  /* optional: zero_n int64 words at zero_buf are cleared by this launch (the accumulator, once its consumer has run) */
  int64_t* zero_buf;
  int32_t zero_n;
} gbx_exp_qlinear_fused;
int gbx_exp_qlinear_fused_forward(const gbx_exp_qlinear_fused* desc, void* hip_stream);

/* Load-time re-tiling for the matrix-core decode GEMV (2-bit, group_size 128, K % 1024 == 0): 16-row tiles, K-steps of
 * 256 values, so that a wavefront load is one contiguous 1 KiB block that IS the int8 MFMA operand layout, plus the
 * (scale | zero << 16) dwords as [tile][group][16 rows].  With a second matrix (qweight_up != NULL: gate_proj / up_proj
 * of one SwiGLU launch, both [N, K]) a tile holds 8 gate rows followed by the 8 up rows of the same outputs.
 * Sizes in uint32: wt = T*16 * K/16, szt = T*16 * K/128 with T = ceil(N/16) (single) or ceil(N/8) (pair); rows past N
 * are zero-filled.  The originals stay valid for the other entry points. */
int gbx_qlinear_prepack_tiles(const uint32_t* qweight, const void* scales, const void* zeros,
                              const uint32_t* qweight_up, const void* scales_up, const void* zeros_up,
                              uint32_t* wt, uint32_t* szt, int32_t N, int32_t K, int32_t bits, int32_t group_size,
                              void* hip_stream);

/* Round 5: the batched-decode GEMM (M <= 32 activation rows, 2 bits, group 128, bf16) on the FP4 x FP8 matrix cores -- the operand the
 * round-3 / round-4 reviews asked to see measured at M = 32 (product path: gbx_qlinear_forward at 2 <= M <= 32, mfma_batch.hip; reference op
 * mx.quantized_matmul at [B, 1, K], gbx_lm/models/quantized_linear_gba.py:195-203).  The activation tile is converted ONCE into two E4M3
 * limbs + one E8M0 exponent per 32 values + per-group sums (gbx_exp_batch_mx_image: image_bytes(K) + aux_bytes(K) of caller memory; rows
 * beyond M are zero); gbx_exp_batch_mx_forward then computes y[M, N] = x . dequant(W)^T with four scaled MFMAs per (16 weight rows,
 * group).  N % 128 == 0, K % 1024 == 0.  Tolerance: the two-limb bound of dgemv_mx.hip. */
size_t gbx_exp_batch_mx_image_bytes(int32_t K);
size_t gbx_exp_batch_mx_aux_bytes(int32_t K);
int gbx_exp_batch_mx_image(const void* x, int64_t M, int32_t K, void* image, void* aux, void* hip_stream);
/* k ranges the launch splits this shape into (1: y [M, N] is written; > 1: fp32 slices slices_ws [ranges][M][N] for the consumer to sum,
 * like gbx_qlinear_forward_partials; 0: shape not covered) */
int gbx_exp_batch_mx_slices(int32_t N, int32_t K);
int gbx_exp_batch_mx_forward(const void* image, const void* aux, const uint32_t* qweight, const void* scales, const void* zeros, void* y,
                             float* slices_ws, int64_t M, int32_t N, int32_t K, void* hip_stream);

/* y[N] = W x' for ONE activation row from the tiled form above; same result contract and the same optional fusions as
 * gbx_qlinear_fused_forward: x' = RMSNorm(x) * norm_weight (norm_weight != NULL), y <- residual + round(result)
 * (residual != NULL), or -- pair tiles, swiglu != 0 -- y[n] = silu(gate[n]) * up[n] with the reference's op-by-op
 * roundings.  Exact integer sums per group on the int8 matrix cores, fp32 scale/zero application. */
int gbx_gemv_tiled(const void* x, const uint32_t* wt, const uint32_t* szt, void* y, int32_t N, int32_t K, int32_t bits,
                   int32_t group_size, const void* norm_weight, float norm_eps, const void* residual, int32_t swiglu,
                   int32_t act_dtype, void* hip_stream);


/* gbx_attn_decode and the o_proj QuantizedLinear call behind it (qllama.py:78-96; quantized_linear_gba.py:195-203 at M = 1)
 * in ONE launch, for short caches: workgroup (kv group g, row block r) computes the attention of the group's query heads
 * and multiplies its slice Wo[128 r .. 128 r + 127, columns of group g]; the partial row sums are ADDED into acc[D] as
 * 64-bit fixed-point integers in units of 2^-32 (integer adds: the result does not depend on the arrival order).  The
 * workgroups r < 4 of a group each compute ONE query head (all 16 waves) and hand its 128 outputs to the group's other
 * workgroups inside the launch as 72 eight-byte {tag, value} words written through to memory; consumers poll them
 * (bounded spin; they only ever wait for workgroups with a lower block index).
 * acc: int64[D + n_kv_heads * GBX_ATTN_OPROJ_HANDOFF_WORDS]: the first D words receive the sums, the rest is the hand-off
 * area; ALL of it must be zero on entry (a gbx_qlinear_fused launch with zero_buf = acc, zero_n = its length clears it
 * again once the consumer -- a gbx_qlinear_fused launch with x_acc = acc -- has run).  Shapes: head_dim 128,
 * n_heads == 4 * n_kv_heads, D % 128 == 0, D >= 512, o_proj [D, n_heads * head_dim] at 2 / 3 / 4 bits with group_size 64 or
 * 128 (gbx_attn_oproj_supported says so).  The context is not split over workgroups: callers switch to gbx_attn_decode_ws +
 * gbx_qlinear_fused_forward once the cache is long (the decode runner: capacity >= 1024). */
#define GBX_ATTN_OPROJ_HANDOFF_WORDS 288
int gbx_attn_oproj_supported(int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int32_t D, int32_t bits, int32_t group_size);
int gbx_attn_oproj_decode(const void* q, const void* k, const void* v, void* kcache, void* vcache, const float* inv_freq,
                          const int64_t* pos, const uint32_t* o_qweight, const void* o_scales, const void* o_zeros,
                          int64_t* acc, int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int32_t capacity, int32_t D,
                          int32_t bits, int32_t group_size, float rope_scale, float sm_scale, int32_t traditional,
                          int32_t act_dtype, void* hip_stream);

/* ---- streamed decode GEMV / persistent decode engine (gbx-lm_amd/csrc/gemv_stream.hip) -------------------------------
 * A PROGRAM is a device-resident array of encoded ops; one launch walks ops [begin, end) with one 8-wave workgroup per
 * CU.  Every op is one gbx_qlinear_fused group (q|k|v, o, gate|up + SwiGLU, down: the QuantizedLinear calls of
 * gbx_lm/models/qllama.py:76,96,115 at M = 1) with the same arithmetic, roundings and results as
 * gbx_qlinear_fused_forward (bit-identical without the fused RMSNorm, whose fp32 sum order differs), but the packed
 * rows reach the CU through a per-wave LDS ring filled by LDS-DMA that runs ahead of the activation prologue and, when
 * a launch covers several ops, ahead of the next op's dependency (weights never depend on activations).
 * Requirements per op: desc->sz present (gbx_pack_scales_zeros), group_size 64 or 128 (32 at 3/4 bits), K % 128 == 0,
 * K / (64 values at 2 bits | 32 at 3/4 bits) <= 512, RMSNorm / SwiGLU only where that quotient is <= 128.
 * gbx_stream_encode validates n descriptors and writes n * GBX_STREAM_OP_BYTES bytes of device-format ops into HOST
 * memory `out`; the caller uploads them (the library never allocates) and keeps the host copy: gbx_stream_run enqueues one
 * launch and passes op `begin` by value in the kernel arguments (no cold fetch of the program before the first load).
 * Also needed per op: K / group_size even, sz 16-byte aligned. */
#define GBX_STREAM_OP_BYTES 160
int gbx_stream_encode(const gbx_exp_qlinear_fused* descs, int32_t n, void* out, size_t out_bytes);
int gbx_stream_run(const void* device_program, const void* host_program, int32_t begin, int32_t end, int32_t act_dtype,
                   void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* GBX_EXPERIMENTAL_H */
