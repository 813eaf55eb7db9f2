/* gbx_qlinear.h -- C-ABI of the MI355X-native GBA QuantizedLinear hot path.
 *
 * Drop-in boundary for the ONE operator call the reference makes on this path:
 *
 *     mx.quantized_matmul(x, qweight, scales=scales, biases=zeros, transpose=True,
 *                         group_size=group_size, bits=bits)  [+ bias]
 *                                   -- gbx_lm/models/quantized_linear_gba.py:195-205
 *
 * Conventions (all entry points)
 *   - plain pointers and sizes only; no torch / HIP C++ types in signatures.
 *   - every pointer is a DEVICE pointer owned by the caller; the library never
 *     allocates, frees or retains device memory (workspace is caller-provided).
 *   - all work is enqueued on `hip_stream` (a hipStream_t passed as void*; NULL =
 *     the null stream); no host synchronisation, so every call is legal inside
 *     hipStreamBeginCapture / hipGraph capture.
 *   - return value: GBX_OK or a GBX_ERR_* code; gbx_last_error() returns a
 *     thread-local message for the last failing call on this thread.
 *   - dtype codes: GBX_F16 = IEEE half, GBX_BF16 = bfloat16.  x, scales, zeros,
 *     bias and y share `act_dtype` (the loader casts scales/zeros to the
 *     activation dtype, cf. gbx_lm/utils.py:841-843 which casts them to bf16).
 *
 * Packed layout (identical to the reference tensors, no re-packing required):
 *   qweight  uint32 [N, K*bits/32] row-major; field k of row n = `bits` bits at bit
 *            offset k*bits of the row's little-endian bit-stream
 *                                   -- quantized_linear_gba.py:61,64-67; SURVEY.md App. A
 *   scales, zeros  [N, K/group_size] -- quantized_linear_gba.py:62,110-117
 *   w[n,k] = scales[n,k/g] * q[n,k] + zeros[n,k/g]   ("+": quantized_linear_gba.py:151-155)
 *   bits in {2,3,4}; group_size in {32,64,128} (quantized_linear_gba.py:272); K % 32 == 0.
 */
#ifndef GBX_QLINEAR_H
#define GBX_QLINEAR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GBX_OK 0
#define GBX_ERR_INVALID 1     /* shape / bits / group / alignment violation  -> ValueError   */
#define GBX_ERR_HIP 2         /* HIP runtime error                           -> RuntimeError */
#define GBX_ERR_UNSUPPORTED 3 /* valid but not implemented                   -> NotImplementedError */

#define GBX_F16 0
#define GBX_BF16 1

/* Library ABI version (major*1000 + minor). */
int gbx_version(void);

/* Thread-local message of the last failing call ("" if none). Never NULL. */
const char* gbx_last_error(void);

/* Validates (N, K, bits, group_size) exactly as the reference does at
 * quantized_linear_gba.py:61-62,272 and utils.py:819-821. GBX_OK or GBX_ERR_INVALID. */
int gbx_qlinear_check(int64_t M, int32_t N, int32_t K, int32_t bits, int32_t group_size);

/* y[M,N] = x[M,K] @ dequant(qweight, scales, zeros)^T (+ bias[N] if bias != NULL).
 * Replaces mx.quantized_matmul(..., transpose=True) + bias add
 * (quantized_linear_gba.py:195-205).  fp32 accumulation, output rounded once to act_dtype.
 *   M == 1           : wavefront-reduction GEMV (HBM-bound decode path)
 *   M  > 1           : batched path
 * x row stride = K elements, y row stride = N elements (contiguous). */
int gbx_qlinear_forward(const void* x, const uint32_t* qweight, const void* scales,
                        const void* zeros, const void* bias, void* y,
                        int64_t M, int32_t N, int32_t K, int32_t bits, int32_t group_size,
                        int32_t act_dtype, void* hip_stream);

/* gbx_qlinear_forward with a caller-owned scratch buffer of at least gbx_qlinear_workspace_bytes(M, N, K, bits, group_size)
 * bytes (16-byte aligned device memory, contents undefined before and after, never retained).  With it the
 * batched-decode kernel (2 <= M <= 32) splits long rows across workgroups -- the down projection of a B = 32 decode
 * step: 25 -> 13 us -- and sums the fp32 slices in a second launch; without it (or where the size query returns 0)
 * the call is identical to gbx_qlinear_forward.  Results differ from it only in the fp32 association of the k sum. */
int gbx_qlinear_forward_ws(const void* x, const uint32_t* qweight, const void* scales, const void* zeros,
                           const void* bias, void* y, int64_t M, int32_t N, int32_t K, int32_t bits,
                           int32_t group_size, int32_t act_dtype, void* workspace, size_t workspace_bytes, void* hip_stream);

/* Bytes of caller-provided scratch with which gbx_qlinear_forward_ws runs this shape faster (0: no use for it; today
 * non-zero for decode batches 2 <= M <= 32 on long-row projections -- group 128, N <= 8192, K >= 8192 -- and for short
 * prompts, 64 < M <= 2048, whose 128 x 128 tiling would leave most of the chip idle). */
size_t gbx_qlinear_workspace_bytes(int64_t M, int32_t N, int32_t K, int32_t bits, int32_t group_size);

/* q[N,K] int32 = integer fields of qweight (kernel-side unpack dump; the bit-exact half of
 * the parity contract).  No reference call site: test/debug surface only. */
int gbx_unpack(const uint32_t* qweight, int32_t* q, int32_t N, int32_t K, int32_t bits,
               void* hip_stream);

/* w[N,K] = scales*q + zeros.  Role of mx.dequantize as used at gbx_lm/tuner/lora.py:62-68 and
 * gbx_lm/tuner/utils.py:214-220.  out_dtype: GBX_F16 / GBX_BF16, or 2 for fp32 (bit-exact
 * against the oracle: one fp32 multiply and one fp32 add, no contraction). */
#define GBX_F32 2
int gbx_dequantize(const uint32_t* qweight, const void* scales, const void* zeros, void* w,
                   int32_t N, int32_t K, int32_t bits, int32_t group_size, int32_t sz_dtype,
                   int32_t out_dtype, void* hip_stream);

/* ------------------------------------------------------------------------------------------------
 * Fused decode-block entry points (batch 1, one new token).  Same conventions as above.  They replace the
 * reference's per-op call sequence inside one transformer block so that a decoded token is ~5 launches per
 * layer instead of ~40:   gbx_lm/models/qllama.py:68-96 (Attention), :114-115 (MLP), :131-141 (block).
 * ------------------------------------------------------------------------------------------------ */

/* Up to three packed matrices that share the activation vector x[K], bits and group_size, streamed by ONE
 * launch (q/k/v or gate/up), with optional prologue / epilogue fusions:
 *   norm_weight != NULL : x <- RMSNorm(x) * norm_weight, rounded once to act_dtype   (qllama.py:138-139)
 *   residual    != NULL : y[0][n] <- residual[n] + round(result[n])   (nseg == 1)     (qllama.py:138,140)
 *   swiglu      != 0    : nseg == 2, y[0][n] <- silu(seg0[n]) * seg1[n]               (qllama.py:115)
 * Unused segment slots must be zeroed. */
typedef struct gbx_qlinear_fused {
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
  /* optional: packed (scale | zero << 16) companion of the whole launch, built once at load time by
   * gbx_pack_scales_zeros (segments concatenated in order, every N even; gate/up interleaved for swiglu) --
   * one 8-byte load per lane and row pair instead of four 2-byte loads. */
  const uint32_t* sz;
} gbx_qlinear_fused;

int gbx_qlinear_fused_forward(const gbx_qlinear_fused* desc, void* hip_stream);

/* ------------------------------------------------------------------------------------------------
 * Decode-block GEMV (round 4): ONE packed matrix (or the gate / up pair) against ONE activation vector, the launch the
 * graph-replayed batch-1 decode runs 4x per layer.  Same arithmetic and result contract as gbx_qlinear_fused_forward
 * (bit-identical for equal workgroup widths); what differs is the launch overhead around it (104 bytes of kernel
 * arguments, 2 / 4 rows per batch with one (scale | zero) load, vector epilogue, no steady-state loop) and an RMSNorm
 * statistic that can be CARRIED between launches: a launch with ssq_out leaves one partial sum of y^2 per workgroup, the
 * next launch with ssq_in adds them instead of reducing x^2 itself.
 *   mx.quantized_matmul at M == 1: gbx_lm/models/quantized_linear_gba.py:195-203; call sites qllama.py:76,115,137-141.
 * Covered: bits 2/3/4, group 64/128 (3/4 bits also 32), K % 128 == 0, K <= 512 lane slots, N a multiple of
 * rows_per_batch (epilogue SWIGLU: of rows_per_batch / 2), whole matrix < 2 GB.  gbx_dgemv_plan answers whether a shape is
 * covered and how the companion must be packed; everything else stays on gbx_qlinear_fused_forward.
 * ------------------------------------------------------------------------------------------------ */
#define GBX_DG_PLAIN 0
#define GBX_DG_RESIDUAL 1    /* y[n] <- residual[n] + round(result[n])                 (qllama.py:138,140) */
#define GBX_DG_SWIGLU 2      /* y[n] <- silu(qweight . x)[n] * (qweight_up . x)[n]      (qllama.py:115) */
#define GBX_DG_SSQ_SLOTS 1024 /* floats of a partial-statistic buffer; zero-filled by the caller once */

typedef struct gbx_dgemv {
  uint32_t struct_size;        /* sizeof(gbx_dgemv) of the caller's header: a mismatch is GBX_ERR_INVALID */
  int32_t epilogue;            /* GBX_DG_* */
  const void* x;               /* [K], act_dtype, 16-byte aligned */
  const uint32_t* qweight;     /* [N, K*bits/32] (SWIGLU: gate_proj) */
  const uint32_t* qweight_up;  /* SWIGLU: up_proj, same shape; else NULL */
  const uint32_t* sz;          /* gbx_dgemv_pack output for this launch's rows_per_batch */
  void* y;                     /* [N] */
  const void* residual;        /* RESIDUAL: [N] (may alias y) */
  const void* norm_weight;     /* optional RMSNorm weight [K]: x <- RMSNorm(x) * w, rounded once (K <= 128 lane slots) */
  const float* ssq_in;         /* optional with norm_weight: GBX_DG_SSQ_SLOTS partial sums of x^2 (a producer's ssq_out) */
  float* ssq_out;              /* optional (PLAIN / RESIDUAL): GBX_DG_SSQ_SLOTS floats; this launch writes its first
                                  `ssq_partials` (gbx_dgemv_plan) entries, the rest must be zero */
  int32_t N, K, bits, group_size, act_dtype;
  int32_t rows_per_batch;      /* what sz was packed for (gbx_dgemv_plan) */
  float norm_eps;
  /* round 5, with gbx_dattn_forward (below): o_proj's output arrives as 2^-32 fixed-point partial sums instead of a vector */
  const int64_t* x_acc;        /* optional, with norm_weight and without ssq_in: [K + 1]; the prologue forms
                                  x <- round(x + round(x_acc * 2^-32)) (qllama.py:96,138: o_proj's rounding, then the residual
                                  add's); x_acc[K] != 0 (a non-finite partial sum) makes every x NaN */
  void* x_out;                 /* optional with x_acc: [K], the new x (the residual stream), written once */
  uint64_t* zero_buf;          /* optional: zero_n 8-byte words this launch clears (the accumulator, after its consumer) */
  int32_t zero_n;
} gbx_dgemv;

/* rows_per_batch (2 / 4) the library prefers for this launch (with_norm: it will carry norm_weight) and the number of partial
 * sums of y^2 it writes when given ssq_out (0: this launch cannot carry the statistic); GBX_ERR_UNSUPPORTED when dgemv does not
 * cover the shape.  gbx_dgemv_forward accepts either packing (rows_per_batch 2 or 4) where the rows divide. */
int gbx_dgemv_plan(int32_t N, int32_t K, int32_t bits, int32_t group_size, int32_t epilogue, int32_t with_norm,
                   int32_t* rows_per_batch, int32_t* ssq_partials);
/* out[b][g][r] = scale | zero << 16 (16-bit patterns verbatim; gbx_lm/utils.py:841-843 is the cast that made them) of row r
 * of batch b: row b*R + r of (scales0, zeros0), or -- scales1 != NULL, SWIGLU -- r < R/2: row b*R/2 + r of (scales0, zeros0) =
 * gate, else row b*R/2 + r - R/2 of (scales1, zeros1) = up.  out = uint32[N * G] (SWIGLU: [2 * N * G]), G = K / group_size. */
int gbx_dgemv_pack(const void* scales0, const void* zeros0, const void* scales1, const void* zeros1, uint32_t* out,
                   int32_t N, int32_t G, int32_t rows_per_batch, void* hip_stream);
int gbx_dgemv_forward(const gbx_dgemv* desc, void* hip_stream);
/* The kernel gbx_dgemv_forward launches for this descriptor, for profiles and bench labels: 0 = dgemv_kernel (vector dot),
 * 1 = dgemv_mx_kernel (FP4 x FP8 matrix cores); workgroup width and rows per batch through the optional out parameters.
 * GBX_ERR_UNSUPPORTED as gbx_dgemv_forward would report it. */
int gbx_dgemv_kernel_id(const gbx_dgemv* desc, int32_t* kernel, int32_t* waves_per_workgroup, int32_t* rows_per_batch);

/* ------------------------------------------------------------------------------------------------
 * Fused attention launch of a decoder block at M == 1 (round 5): input RMSNorm -> q | k | v projections -> RoPE + KV-cache
 * append -> attention -> o_proj, ONE launch per block instead of three (qllama.py:137 norm, :76 q/k/v, :78-95 attention,
 * :96 o_proj).  One kv head per XCD (n_kv_heads == 8, head_dim 128): the two hand-offs inside the launch (q|k|v -> attention ->
 * o_proj) stay inside one XCD's L2; o_proj is K-split by kv group and its partial sums meet in `acc` as 2^-32 fixed-point
 * integers (order-independent, bit-reproducible).  The consumer is the block's gate|up launch (gbx_dgemv.x_acc / x_out), the
 * down_proj launch clears acc (gbx_dgemv.zero_buf).  Caches longer than the launch's single sweep stay on gbx_attn_decode_ws.
 * ------------------------------------------------------------------------------------------------ */
typedef struct gbx_dattn {
  uint32_t struct_size;        /* sizeof(gbx_dattn) */
  int32_t act_dtype;
  const void* x;               /* [D] residual stream */
  const void* norm_weight;     /* [D] input_layernorm weight */
  const float* ssq_in;         /* optional: GBX_DG_SSQ_SLOTS partial sums of x^2 left by the launch that wrote x */
  const uint32_t* qkv_qweight; /* [(n_heads + 2 * 8) * 128, D * qkv_bits / 32]: q_proj | k_proj | v_proj rows concatenated */
  const uint32_t* qkv_sz;      /* gbx_dgemv_pack of that matrix, rows_per_batch 2 */
  void* kcache;                /* [8, capacity, 128] (reference layout, models/cache.py:222-244, batch 1) */
  void* vcache;
  const float* inv_freq;       /* [64] RoPE inverse frequencies */
  const int64_t* pos;          /* device scalar: index of the new token (cached positions [0, pos)) */
  const uint32_t* o_qweight;   /* [D, n_heads * 128 * o_bits / 32] */
  const void* o_scales;        /* [D, n_heads * 128 / o_group_size], act_dtype */
  const void* o_zeros;
  int64_t* acc;                /* [D + 1], zero on entry; receives o_proj . attention as 2^-32 fixed point (word D: non-finite marker) */
  void* state;                 /* gbx_dattn_state_bytes() bytes, zero-filled ONCE by the caller, private to one stream */
  void* attn_out;              /* optional [n_heads * 128]: the attention output before o_proj (tests) */
  /* after a block whose MLP ran as gbx_dmlp_forward the stream arrives as (x, fixed-point down_proj sums): */
  const int64_t* x_acc;        /* optional [D + 1], without ssq_in: the launch normalises round(x + round(x_acc * 2^-32)) ... */
  void* x_out;                 /* ... and writes that vector here once (must not alias x) */
  uint64_t* zero_buf;          /* optional: zero_n 8-byte words this launch clears (an accumulator whose consumer has run) */
  int32_t zero_n;
  int32_t D, n_heads, capacity;
  int32_t qkv_bits, qkv_group_size, o_bits, o_group_size;
  int32_t traditional;
  float norm_eps, rope_scale, sm_scale;
} gbx_dattn;

/* GBX_OK when the fused launch covers the shape (8 kv heads, head_dim 128, D <= 4096, n_heads a multiple of 8 up to 64, ...),
 * GBX_ERR_UNSUPPORTED otherwise (the caller keeps gbx_dgemv_forward + gbx_attn_decode + gbx_dgemv_forward). */
int gbx_dattn_plan(int32_t D, int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int32_t qkv_bits, int32_t qkv_group_size,
                   int32_t o_bits, int32_t o_group_size);
size_t gbx_dattn_state_bytes(void);
int gbx_dattn_forward(const gbx_dattn* desc, void* hip_stream);
/* Synchronous read of the launch status word: *code = 0, or the hand-off that timed out (1: q|k|v granules, 2: attention output,
 * 3: the activations of gbx_dmlp_forward / gbx_dblock_forward, 4: the rotated q of the cooperative sweep, 5: the partial softmax
 * states of the cooperative sweep, 6: gbx_dblock_forward's arrival line, 7: its leader flags, 8: its x' granules, 9: a context
 * beyond what gbx_dblock_forward sweeps).  A launch that gives up ALSO raises the non-finite marker of the accumulator it adds to,
 * so the residual stream -- and with it the logits and the token's logprob -- turn NaN in-band; the caller must still fail or fall
 * back (gbx-lm_amd/decode.py rebuilds its runner on the five-launch block and replays the step).  Off the timed path (it
 * synchronises the device).  Serves the state buffer of any of the fused launches. */
int gbx_dattn_status(const void* state, int32_t* code);

/* ------------------------------------------------------------------------------------------------
 * Fused MLP launch of a decoder block at M == 1 (round 5): residual add of o_proj's fixed-point sums -> post-attention RMSNorm ->
 * gate_proj | up_proj + SwiGLU -> down_proj, ONE launch (qllama.py:138-140, :115).  XCD g of the chip produces the activations
 * [g I/8, (g+1) I/8) and consumes exactly that range as down_proj's input columns, so the one exchange inside the launch stays in
 * that XCD's L2; the eight partial sums per output row meet in acc_out as 2^-32 fixed-point integers.  The NEXT launch forms the
 * residual stream: gbx_dattn.x_acc / x_out of the next block, or gbx_acc_residual after the last one.
 * Covered: 2 bits, group 128, bf16, D = 4096, I a multiple of 1024 with I / 256 even in [32, 64] (Llama-3-8B: 14336) on the FP4 x FP8
 * matrix cores; 3 or 4 bits (gate, up and down alike), groups 32 / 64 / 128, bf16, D = 4096 or 3072 (Llama-3.2-3B: I = 8192) on the
 * vector-ALU arithmetic of the decode GEMV, down_proj's words streamed into LDS (round 6).
 * ------------------------------------------------------------------------------------------------ */
typedef struct gbx_dmlp {
  uint32_t struct_size;        /* sizeof(gbx_dmlp) */
  int32_t act_dtype;           /* GBX_BF16 */
  const void* x;               /* [D] the stream before o_proj's output is added */
  const int64_t* x_acc;        /* [D + 1] o_proj sums (gbx_dattn.acc) */
  void* x_out;                 /* [D] round(x + round(x_acc * 2^-32)), written once: the stream the next launch adds down_proj to */
  const void* norm_weight;     /* [D] post_attention_layernorm weight */
  const uint32_t* gate_qweight;/* [I, D / 16] */
  const uint32_t* up_qweight;  /* [I, D / 16] */
  const uint32_t* gate_up_sz;  /* gbx_dgemv_pack(gate, up) with rows_per_batch 2 */
  const uint32_t* down_qweight;/* [D, I / 16] */
  const uint32_t* down_sz;     /* gbx_dgemv_pack(down) with rows_per_batch 2 */
  int64_t* acc_out;            /* [D + 1], zero on entry: down_proj sums, 2^-32 fixed point (word D: non-finite marker) */
  uint64_t* zero_buf;          /* optional: zero_n 8-byte words this launch clears */
  void* state;                 /* gbx_dattn_state_bytes() bytes of its OWN (not the attention launch's), zero-filled once */
  void* act_out;               /* optional [I]: the SwiGLU output (tests) */
  int32_t D, I, zero_n;
  float norm_eps;
  int32_t gate_up_bits, gate_up_group_size, down_bits, down_group_size;   /* since 1300; 0 = 2 bits / group 128 */
  int32_t weight_layout;       /* since 1302: GBX_LAYOUT_REFERENCE, or GBX_LAYOUT_PLANES when gate / up / down_qweight are gbx_repack_planes outputs */
  int32_t reserved0;           /* 0 */
} gbx_dmlp;
/* 3-bit rows for the matrix cores (since 1302).  The loader's 3-bit row is a bit stream whose fields straddle words; gbx_repack_planes
 * rewrites every 32 values (three words, in place in the row) as a 2-bit plane (two words: exactly a 2-bit row's words, the low two bits)
 * and a 1-bit plane (one word, bit 4 n + 0 / 1 / 2 / 3 = the high bit of value 2 n / 2 n + 1 / 16 + 2 n / 17 + 2 n): q = lo + 4 hi, same
 * bytes, same shape [N, K * 3 / 32].  A load-time companion like gbx_dgemv_pack; the reference tensor stays what it was.
 * gbx_dmlp_weight_layout: the layout gbx_dmlp_forward wants for these shapes (negative: no fused MLP launch covers them). */
#define GBX_LAYOUT_REFERENCE 0
#define GBX_LAYOUT_PLANES 1
int gbx_repack_planes(const uint32_t* qweight, uint32_t* out, int64_t N, int32_t K, int32_t bits, void* hip_stream);
int gbx_dmlp_weight_layout(int32_t D, int32_t I, int32_t gate_up_bits, int32_t gate_up_group_size, int32_t down_bits, int32_t down_group_size,
                           int32_t act_dtype);
int gbx_dmlp_plan(int32_t D, int32_t I, int32_t gate_up_bits, int32_t gate_up_group_size, int32_t down_bits, int32_t down_group_size,
                  int32_t act_dtype);
int gbx_dmlp_forward(const gbx_dmlp* desc, void* hip_stream);
/* out[i] <- round(x[i] + round(acc[i] * 2^-32)) for i < D (acc[D] != 0: NaN), then acc and up to three more [D + 1] accumulators
 * are cleared: the step between the last block's gbx_dmlp_forward and the output head.  act_dtype: GBX_F16 / GBX_BF16 (since 1300). */
int gbx_acc_residual(const void* x, int64_t* acc, void* out, int32_t D, int32_t act_dtype, int64_t* zero0, int64_t* zero1, int64_t* zero2,
                     void* hip_stream);
/* ------------------------------------------------------------------------------------------------
 * One WHOLE decoder block at M == 1 as one launch (round 6): gbx_dattn_forward followed by gbx_dmlp_forward, fused -- input RMSNorm ->
 * q | k | v -> RoPE + cache append -> attention -> o_proj -> residual -> post-attention RMSNorm -> gate | up + SwiGLU -> down_proj
 * (qllama.py:131-141 with :68-96 and :114-115; seven mx.quantized_matmul calls, quantized_linear_gba.py:195-203).  The MLP's packed
 * rows stream into LDS while the attention chain runs; o_proj's sums and the MLP's prologue meet in ONE chip-wide release inside the
 * launch (per-XCD arrival line -> 8 leader flags -> each XCD's leader forms x' = round(h + round(o)), its statistic and RMSNorm(x')
 * once and publishes it XCD-locally).  Same arithmetic, same roundings, same accumulators as the two launches it replaces:
 * o_proj's sums land in acc_o, down_proj's in acc_d, the NEXT launch (gbx_dblock / gbx_dattn x_acc, or gbx_acc_residual) forms the stream.
 * Covered: 2 bits, group 128, bf16, D = 4096, 32 query / 8 kv heads of 128, I = 14336 (Llama-3-8B), at most gbx_dblock_max_context()
 * cached positions (beyond it the caller replays gbx_dattn_forward + gbx_dmlp_forward, whose cooperative sweep covers long contexts).
 * ------------------------------------------------------------------------------------------------ */
typedef struct gbx_dblock {
  uint32_t struct_size;        /* sizeof(gbx_dblock) */
  int32_t act_dtype;           /* GBX_BF16 */
  /* attention half: as gbx_dattn */
  const void* x;               /* [D] residual stream entering the block */
  const void* norm_weight;     /* [D] input_layernorm weight */
  const float* ssq_in;         /* optional carried statistic of x (not with x_acc) */
  const uint32_t* qkv_qweight; /* q | k | v rows concatenated */
  const uint32_t* qkv_sz;      /* gbx_dgemv_pack of that matrix, rows_per_batch 2 */
  void* kcache;                /* [8, capacity, 128] */
  void* vcache;
  const float* inv_freq;       /* [64] */
  const int64_t* pos;          /* device scalar */
  const uint32_t* o_qweight;
  const void* o_scales;
  const void* o_zeros;
  int64_t* acc_o;              /* [D + 1], zero on entry: o_proj sums */
  const int64_t* x_acc;        /* optional [D + 1]: the previous block's down_proj sums */
  void* x_out;                 /* with x_acc: [D] round(x + round(x_acc * 2^-32)), written once */
  void* attn_out;              /* optional [4096] (tests) */
  /* MLP half: as gbx_dmlp */
  const void* norm2_weight;    /* [D] post_attention_layernorm weight */
  const uint32_t* gate_qweight;
  const uint32_t* up_qweight;
  const uint32_t* gate_up_sz;  /* gbx_dgemv_pack(gate, up), rows_per_batch 2 */
  const uint32_t* down_qweight;
  const uint32_t* down_sz;     /* gbx_dgemv_pack(down), rows_per_batch 2 */
  int64_t* acc_d;              /* [D + 1]: down_proj sums; zero on entry unless it is one of the zero_bufs */
  void* x2_out;                /* [D]: the stream after the attention residual (the next launch adds acc_d to it); may be x, not x_out */
  void* act_out;               /* optional [I] (tests) */
  uint64_t* zero_buf0;         /* optional: zero_n0 8-byte words this launch clears (not x_acc, not acc_o; acc_d itself is allowed) */
  uint64_t* zero_buf1;         /* optional: zero_n1 words */
  void* state;                 /* gbx_dblock_state_bytes() bytes, 64-byte aligned, zero-filled ONCE, private to one stream */
  int32_t zero_n0, zero_n1;
  int32_t capacity, traditional;
  float norm_eps, norm2_eps, rope_scale, sm_scale;
} gbx_dblock;
int gbx_dblock_plan(int32_t D, int32_t I, int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int32_t qkv_bits, int32_t qkv_group_size,
                    int32_t o_bits, int32_t o_group_size, int32_t gate_up_bits, int32_t gate_up_group_size, int32_t down_bits,
                    int32_t down_group_size, int32_t act_dtype);
size_t gbx_dblock_state_bytes(void);
int32_t gbx_dblock_max_context(void);
int gbx_dblock_forward(const gbx_dblock* desc, void* hip_stream);
/* gbx_dattn_status for a gbx_dblock state buffer (its status word sits behind a larger hand-off area). */
int gbx_dblock_status(const void* state, int32_t* code);

/* The device check behind every launch with XCD-local hand-offs (gbx_dattn_*, gbx_dmlp_*, gbx_dblock_*, gbx_add_rmsnorm_slices_xcd):
 * 256 CUs in one partition, no CU mask on the process, and -- measured once per device by a census launch of the same geometry --
 * consecutive workgroups on 8 different XCDs, one workgroup per CU.  GBX_OK, or GBX_ERR_UNSUPPORTED with the reason in
 * gbx_last_error() (the *_plan functions then refuse too and the caller keeps the five-launch block).  Synchronous: not inside a
 * stream capture.  (since 1300) */
int gbx_xcd_layout_check(void);

/* Load-time companion of the loader's scale/zero cast (gbx_lm/utils.py:841-843): out[p][g][i] = scale bits | zero bits << 16
 * (16-bit act-dtype patterns, copied verbatim) for the two rows a GEMV batch streams together:
 *   scales1 == NULL : rows (2p, 2p+1) of (scales0, zeros0), out = uint32[N/2][G][2], N even;
 *   scales1 != NULL : row p of (scales0, zeros0) and row p of (scales1, zeros1) -- gate_proj / up_proj of a SwiGLU
 *                     launch --, out = uint32[N][G][2].
 * G = K / group_size. */
int gbx_pack_scales_zeros(const void* scales0, const void* zeros0, const void* scales1, const void* zeros1,
                          uint32_t* out, int32_t N, int32_t G, void* hip_stream);

/* Up to three QuantizedLinear calls that share the activation rows, K, bits and group size (q/k/v, gate/up of one block:
 * qllama.py:76, :115) in ONE launch of the split-K matrix-core kernel, for batched decode: 2 <= M <= 64, K >= 512,
 * K % 128 == 0, every N[i] a multiple of 32.  y[i] is [M, N[i]]; bias[i] may be NULL.  Same arithmetic and result
 * contract as gbx_qlinear_forward per matrix. */
int gbx_qlinear_forward_multi(const void* x, int32_t nseg, const uint32_t* const* qweight, const void* const* scales,
                              const void* const* zeros, const void* const* bias, void* const* y, const int32_t* N,
                              int64_t M, int32_t K, int32_t bits, int32_t group_size, int32_t act_dtype, void* hip_stream);

/* gate_proj, up_proj and SwiGLU of one decoder block for a decode-sized batch in ONE launch of the no-dequant matrix-core
 * kernel: y[m, n] = silu(x . gate^T)[m, n] * (x . up^T)[m, n], x [M, K], both matrices [N, K] of one (bits, group_size),
 * y [M, N]; the four activation-dtype roundings of the unfused graph (gbx_lm/models/qllama.py:115: gate, up, silu, product).
 * 2 <= M <= 64, K % 1024 == 0, N % 16 == 0, (bits, group_size) as the batch kernel takes them (2/3/4 bits at 128, 2 bits
 * also at 64 / 32); GBX_ERR_UNSUPPORTED otherwise (run gbx_qlinear_forward_multi + gbx_swiglu). */
int gbx_qlinear_swiglu_forward(const void* x, const uint32_t* gate_qweight, const void* gate_scales, const void* gate_zeros,
                               const uint32_t* up_qweight, const void* up_scales, const void* up_zeros, void* y, int64_t M,
                               int32_t N, int32_t K, int32_t bits, int32_t group_size, int32_t act_dtype, void* hip_stream);

/* The same op with caller scratch for long prompts: M >= 4096 rows (or >= 1024 rows whose 256 x 256 tiles fill the chip) and a workspace of gbx_qlinear_swiglu_workspace_bytes
 * (2 * N * K * 2 bytes; 0 = this shape has no scratch path) dequantise gate and up ONCE into one dense matrix and run one
 * matrix-core GEMM whose epilogue applies the SwiGLU (same roundings: each GEMM output to the activation dtype, silu rounded,
 * product rounded); without enough scratch, or for smaller M, it forwards to gbx_qlinear_swiglu_forward. */
size_t gbx_qlinear_swiglu_workspace_bytes(int64_t M, int32_t N, int32_t K);
int gbx_qlinear_swiglu_forward_ws(const void* x, const uint32_t* gate_qweight, const void* gate_scales, const void* gate_zeros,
                                  const uint32_t* up_qweight, const void* up_scales, const void* up_zeros, void* y, int64_t M,
                                  int32_t N, int32_t K, int32_t bits, int32_t group_size, int32_t act_dtype, void* workspace,
                                  size_t workspace_bytes, void* hip_stream);

/* The k-split batched-decode GEMM without its summing launch: writes `*slices_out` fp32 partial tiles
 * workspace[s][M][N] (s < *slices_out; size from gbx_qlinear_partials_bytes) and no y.  For the long-row projection of
 * a decoder block (down_proj, 2 <= M <= 32) whose consumer -- the next block's residual add + RMSNorm -- folds the sum in
 * (gbx_add_rmsnorm_slices): one launch and one pass over [M, N] less per layer.  GBX_ERR_UNSUPPORTED when this shape does
 * not split (use gbx_qlinear_forward); no bias. */
size_t gbx_qlinear_partials_bytes(int64_t M, int32_t N, int32_t K, int32_t bits, int32_t group_size);   /* 0: this shape does not split */
int gbx_qlinear_forward_partials(const void* x, const uint32_t* qweight, const void* scales, const void* zeros, void* workspace,
                                 size_t workspace_bytes, int32_t* slices_out, int64_t M, int32_t N, int32_t K, int32_t bits,
                                 int32_t group_size, int32_t act_dtype, void* hip_stream);

/* h[M,K] += round(sum_s slices_ws[s][M][K]) (the one rounding the summing launch would have applied), y = RMSNorm(h) * w.
 * K % 8 == 0, K <= 8192, 16-byte aligned tensors.  gbx_add_rmsnorm with the k-split GEMM's slices as the addend. */
int gbx_add_rmsnorm_slices(void* h, const float* slices_ws, int32_t slices, const void* w, void* y, int64_t M, int32_t K, float eps,
                           int32_t act_dtype, void* hip_stream);

/* gbx_add_rmsnorm_slices spread over the whole chip: a row is cut into K / 512 segments (one single-wave workgroup each) whose
 * partial sums of squares meet in the L2 of one XCD (the row's workgroups share an XCD by construction; {tag, value} granules, bounded
 * polls).  The one-workgroup-per-row form pulls slices * 4 * K bytes through one CU per row; at M = 32, K = 4096, 8 slices this form
 * measures [see DESIGN 3.8] against 4.9 us.  h is bit-identical to gbx_add_rmsnorm_slices; y may differ in the last place (the statistic
 * is summed per segment).  K % 512 == 0, K <= 4096, M <= 8 * (32 / (K / 512)), else GBX_ERR_UNSUPPORTED.
 * state: gbx_norm_state_bytes() bytes, 64-byte aligned, zeroed ONCE by the caller and then owned by these launches (generation tags:
 * one buffer serves every launch of a stream).  The 32-bit word at byte GBX_NORM_STATUS_OFFSET of state is 0, or 1 after a launch
 * whose polls ran out (placement anomaly): the caller reads it outside the timed path.
 * Replaces the same reference lines as gbx_add_rmsnorm_slices (gbx_lm/models/qllama.py:137-141: residual add + RMSNorm of the block). */
size_t gbx_norm_state_bytes(void);
int gbx_add_rmsnorm_slices_xcd(void* h, const float* slices_ws, int32_t slices, const void* w, void* y, int64_t M, int32_t K, float eps,
                               int32_t act_dtype, void* state, void* hip_stream);
#define GBX_NORM_STATUS_OFFSET 16384

/* y[M,N] = x[M,K] . w[N,K]^T (+ bias[N] as a second rounding) for UNQUANTISED 16-bit weights on the matrix cores
 * (csrc/mfma_dense.hip: 256 x 256 tiles, LDS-DMA staging): the output head of a batch of decode rows (`nn.Linear lm_head` /
 * `embed_tokens.as_linear`, gbx_lm/models/qllama.py:194-198, M = batch: the weight matrix is streamed once for all rows)
 * and the GEMM stage of the dequantise-once prefill path.  K % 64 == 0; x, w 16-byte aligned. */
int gbx_dense_forward(const void* x, const void* w, const void* bias, void* y, int64_t M, int32_t N, int32_t K, int32_t act_dtype,
                      void* hip_stream);

/* The same GEMM with caller scratch: shapes whose 256 x 256 tiles would leave most compute units idle (a 1024..4095-row prompt
 * chunk against a 4096-row matrix) split the k range over workgroups, fp32 partial tiles go to the workspace
 * (gbx_dense_workspace_bytes; 0 = this shape does not split) and a second launch sums them in slice order, adds the bias and rounds once. */
size_t gbx_dense_workspace_bytes(int64_t M, int32_t N, int32_t K);
int gbx_dense_forward_ws(const void* x, const void* w, const void* bias, void* y, int64_t M, int32_t N, int32_t K, int32_t act_dtype,
                         void* workspace, size_t workspace_bytes, void* hip_stream);

/* The same GEMM over TWO matrices in one dense image with the SwiGLU in its epilogue: w_gate_up [2 * N, K] holds gate rows and up
 * rows in alternating blocks of 128 (rows [256 t, 256 t + 128) = gate rows [128 t, +128), the next 128 = the same up rows -- the
 * image gbx_qlinear_swiglu_forward_ws leaves in its workspace, which a caller may keep and reuse); y [M, N] = silu(x . gate^T) *
 * (x . up^T) with each GEMM output, the silu and the product rounded to the activation dtype (qllama.py:115). */
int gbx_dense_swiglu_forward(const void* x, const void* w_gate_up, void* y, int64_t M, int32_t N, int32_t K, int32_t act_dtype,
                             void* hip_stream);

/* Expert-indexed QuantizedLinear for a few (token, expert-slot) pairs -- the decode case of
 * mx.gather_qmm(x, qweight[E,N,W], scales[E,N,G], zeros[E,N,G], rhs_indices=indices, transpose=True) + bias[indices]
 * (QuantizedSwitchLinear.__call__, gbx_lm/models/switch_layers.py:78-92):
 *   y[p, :] = W[indices[p]] . x[p / x_div, :] (+ bias[indices[p], :]),  p = 0 .. pairs-1,
 * x_div = pairs sharing one activation row (num_experts_per_tok when SwitchGLU feeds up/gate_proj, 1 for down_proj).
 * indices: device int32[pairs] (out-of-range values are clamped); pairs <= 65535.  Each pair streams its expert's
 * rows once through the decode GEMV; for many tokens sort by expert and call gbx_qlinear_forward per expert. */
int gbx_gather_qlinear_forward(const void* x, const uint32_t* qweight, const void* scales, const void* zeros,
                               const void* bias, const int32_t* indices, void* y, int32_t pairs, int32_t x_div,
                               int32_t num_experts, int32_t N, int32_t K, int32_t bits, int32_t group_size,
                               int32_t act_dtype, void* hip_stream);

/* gate and up projection of SwitchGLU in one expert-indexed launch: y[p, :] = silu(x_p . gate[e_p]^T) * (x_p . up[e_p]^T)
 * with e_p = indices[p], both stacked tensors [num_experts, N, ...] of one (bits, group_size), roundings as the three
 * activation-dtype ops of gbx_lm/models/switch_layers.py:194-199 (up_proj, gate_proj, activation).  Other arguments as
 * gbx_gather_qlinear_forward. */
int gbx_gather_swiglu_forward(const void* x, const uint32_t* gate_qweight, const void* gate_scales, const void* gate_zeros,
                              const uint32_t* up_qweight, const void* up_scales, const void* up_zeros, const int32_t* indices,
                              void* y, int32_t pairs, int32_t x_div, int32_t num_experts, int32_t N, int32_t K, int32_t bits,
                              int32_t group_size, int32_t act_dtype, void* hip_stream);

/* One layer of single-token attention: RoPE on q and k at position *pos, append k, v to the caller's KV cache
 * ([n_kv, capacity, head_dim], the layout of gbx_lm/models/cache.py:222-244 with B = 1), softmax(q k^T * sm_scale) v
 * over positions 0..*pos.  Replaces qllama.py:78-95 (reshape/rope/cache.update_and_fetch/SDPA).  `pos` is a DEVICE
 * int64 so that one captured graph serves every token.  head_dim in {64, 128}; rope over the full head_dim with
 * inverse frequencies inv_freq[head_dim/2] (fp32; already Llama-3-scaled if the model asks for it,
 * gbx_lm/models/rope_utils.py:70-122). */
int gbx_attn_decode(const void* q, const void* k, const void* v, void* kcache, void* vcache,
                    const float* inv_freq, const int64_t* pos, void* out,
                    int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int32_t capacity,
                    float rope_scale, float sm_scale, int32_t traditional, int32_t act_dtype, void* hip_stream);

/* y[N] = W[N,K] x'[K] for an UNQUANTISED 16-bit weight (lm_head / tied embed_tokens.as_linear,
 * qllama.py:194-198), x' = x or RMSNorm(x)*norm_weight (the final model.norm, qllama.py:174) when
 * norm_weight != NULL.  K % 8 == 0, K <= 8192. */
int gbx_dense_gemv(const void* x, const void* w, void* y, const void* norm_weight, float norm_eps,
                   int32_t N, int32_t K, int32_t act_dtype, void* hip_stream);

/* Same, plus greedy sampling fused into the head (the reference's `sampler = argmax(logprobs)`, gbx_lm/utils.py:285,
 * applied to the rounded logits; logsumexp does not change the argmax): every workgroup folds its rows into one of
 * GBX_ARGMAX_SLOTS (= 64) 64-bit keys in `argmax_keys` (uint64[64], must be zero before the call; lowest index wins
 * ties).  gbx_embed_next_token consumes and clears them. */
#define GBX_ARGMAX_SLOTS 64
int gbx_dense_gemv_argmax(const void* x, const void* w, void* y, const void* norm_weight, float norm_eps,
                          int32_t N, int32_t K, void* argmax_keys, int32_t act_dtype, void* hip_stream);

/* Start of the next decode step, entirely on the device: token <- argmax keys (from_keys != 0) or *token_io;
 * *token_io <- token; keys <- 0; *pos += 1 (advance_pos != 0); h[D] <- embed_table[token, :]
 * (embed_tokens lookup, gbx_lm/models/qllama.py:164; feeding y back as the next input, gbx_lm/utils.py:325-338).
 * With from_keys the greedy loop needs no host round trip between tokens. */
int gbx_embed_next_token(void* argmax_keys, int64_t* token_io, int64_t* pos, const void* embed_table, void* h,
                         int32_t D, int32_t vocab, int32_t from_keys, int32_t advance_pos, int32_t act_dtype,
                         void* hip_stream);

/* One-shot peer-to-peer all-reduce(sum) of a short 16-bit vector x[n] (n even, <= max_elems) over the `world` GPUs (2..8) of one
 * node -- the exchange after o_proj / down_proj of tensor-parallel decode (build-side extension).  Every rank owns an INBOX
 * (gbx_p2p_inbox_create: device memory + its 64-byte hipIpc handle), maps every peer's inbox (gbx_p2p_inbox_open on the handles the
 * ranks exchanged out of band) and passes the `world` mapped pointers, own inbox at index `rank`.  gbx_p2p_allreduce pushes x into
 * the peers' inboxes as 8-byte {epoch, payload} granules and sums what arrived in its own inbox in rank order, fp32, one rounding:
 * the same bits on every rank.  Stream-ordered, no host synchronisation, graph-capturable (the epoch lives in the inbox).  All
 * ranks must make the same sequence of calls with the same n.
 * The inbox MUST be fine-grained device memory (a running kernel polls what peers write over xGMI): gbx_p2p_inbox_create returns
 * GBX_ERR_UNSUPPORTED when the runtime cannot provide it -- there is no coarse-grained fallback -- and the caller keeps its RCCL
 * all-reduce.  A launch whose wait for a peer exceeds its bound (tens of seconds) raises an error word (gbx_p2p_status) and
 * returns garbage instead of hanging or trapping. */
size_t gbx_p2p_inbox_bytes(int32_t world, int32_t max_elems);
int gbx_p2p_inbox_create(int32_t world, int32_t max_elems, void** inbox_out, void* ipc_handle_out /* 64 bytes */);
int gbx_p2p_inbox_open(const void* ipc_handle, void** mapped_out);
int gbx_p2p_inbox_close(void* mapped);
int gbx_p2p_inbox_destroy(void* inbox);
int gbx_p2p_allreduce(void* x, int32_t n, int32_t rank, int32_t world, void* const* inboxes, int32_t max_elems, int32_t act_dtype,
                      void* hip_stream);
/* Host-side check (synchronous 4-byte copy): *timed_out = 1 when a gbx_p2p_allreduce launch on this inbox gave up waiting for a
 * peer (its result is then garbage); the flag stays set.  Call it outside the timed path (end of a generation, teardown). */
int gbx_p2p_status(const void* inbox, int32_t world, int32_t max_elems, int32_t* timed_out);

/* Vocabulary-sharded output head (tensor parallel, build-side extension: the reference has no TP): rank r holds rows
 * [index_offset, index_offset + N) of the head.  gbx_dense_gemv_argmax_shard is gbx_dense_gemv_argmax whose keys carry GLOBAL
 * row indices; after the ranks have exchanged their GBX_ARGMAX_SLOTS keys (one all-gather of 512 bytes per rank),
 * gbx_embed_next_token_sharded reduces the nkeys = GBX_ARGMAX_SLOTS * world gathered keys to the next token exactly like
 * gbx_embed_next_token(from_keys = 1) -- same token on every rank, lowest index wins ties -- clears them and (local_keys != NULL)
 * the rank's own GBX_ARGMAX_SLOTS keys. */
int gbx_dense_gemv_argmax_shard(const void* x, const void* w, void* y, const void* norm_weight, float norm_eps, int32_t N, int32_t K,
                                void* argmax_keys, int32_t index_offset, int32_t act_dtype, void* hip_stream);
int gbx_embed_next_token_sharded(void* gathered_keys, int32_t nkeys, void* local_keys, int64_t* token_io, int64_t* pos,
                                 const void* embed_table, void* h, int32_t D, int32_t vocab, int32_t advance_pos, int32_t act_dtype,
                                 void* hip_stream);

/* gbx_attn_decode for `batch` sequences that all stand at position *pos (synchronous batched decode): q / out
 * [batch, n_heads * head_dim], k / v [batch, n_kv * head_dim], caches [batch, n_kv, capacity, head_dim]. */
int gbx_attn_decode_batched(const void* q, const void* k, const void* v, void* kcache, void* vcache, const float* inv_freq,
                            const int64_t* pos, void* out, int32_t batch, int32_t n_heads, int32_t n_kv_heads,
                            int32_t head_dim, int32_t capacity, float rope_scale, float sm_scale, int32_t traditional,
                            int32_t act_dtype, void* hip_stream);

/* gbx_attn_decode_batched with an optional per-head RMSNorm of q and k before RoPE (both weights [head_dim], or both
 * NULL): Qwen3's q_norm / k_norm (gbx_lm/models/qqwen3.py:  self.q_norm(queries...), self.k_norm(keys...)), and with the
 * optional biases of the q / k / v projections (all three [n_heads|n_kv * head_dim], or all NULL: Qwen2,
 * gbx_lm/models/qqwen2.py:44-47), added with the rounding of the projection's own `+ bias` -- the whole attention block
 * between the bias-free q|k|v matmul and o_proj in one launch, any batch >= 1. */
int gbx_attn_decode_normed(const void* q, const void* k, const void* v, void* kcache, void* vcache, const float* inv_freq,
                           const int64_t* pos, void* out, int32_t batch, int32_t n_heads, int32_t n_kv_heads,
                           int32_t head_dim, int32_t capacity, float rope_scale, float sm_scale, int32_t traditional,
                           const void* q_norm_weight, const void* k_norm_weight, float norm_eps, const void* q_bias,
                           const void* k_bias, const void* v_bias, int32_t act_dtype, void* hip_stream);

/* gbx_attn_decode_normed with caller-owned scratch of gbx_attn_workspace_bytes(batch, n_heads, n_kv_heads, head_dim,
 * capacity) bytes (16-byte aligned; 0 = no use for it): once the cache is long the cached positions of a (kv group,
 * sequence) are split over several workgroups (flash-decoding) and a second launch merges their (max, sum, numerator)
 * partials -- the per-head kernel reads a 2048-token context with 32 workgroups. */
size_t gbx_attn_workspace_bytes(int32_t batch, int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int32_t capacity);
int gbx_attn_decode_ws(const void* q, const void* k, const void* v, void* kcache, void* vcache, const float* inv_freq,
                       const int64_t* pos, void* out, int32_t batch, int32_t n_heads, int32_t n_kv_heads,
                       int32_t head_dim, int32_t capacity, float rope_scale, float sm_scale, int32_t traditional,
                       const void* q_norm_weight, const void* k_norm_weight, float norm_eps, const void* q_bias,
                       const void* k_bias, const void* v_bias, int32_t act_dtype, void* workspace, size_t workspace_bytes,
                       void* hip_stream);

/* gbx_attn_decode_ws whose q | k | v arrive as the fp32 slices of ONE k-split batch GEMM over the row-concatenated projection
 * (gbx_qlinear_forward_partials on [q_proj; k_proj; v_proj], N = (n_heads + 2 n_kv_heads) * head_dim): qkv_slices_ws[slices][batch][N],
 * 1 <= slices <= 8.  Every value is round(ws[0] + ws[1] + ...) to the activation dtype, the rounding and order of the GEMM's own summing
 * launch -- the results are bit-identical to gbx_attn_decode_ws on the summed q, k, v -- so the batched decode step of
 * gbx_lm/models/qllama.py:76-95 ([B, 1, K] through q_proj / k_proj / v_proj, RoPE, cache update, SDPA) costs one GEMM launch that
 * stages a quarter of the activations per workgroup and no summing launch.  Bias-free projections only. */
int gbx_attn_decode_slices(const float* qkv_slices_ws, int32_t slices, void* kcache, void* vcache, const float* inv_freq,
                           const int64_t* pos, void* out, int32_t batch, int32_t n_heads, int32_t n_kv_heads,
                           int32_t head_dim, int32_t capacity, float rope_scale, float sm_scale, int32_t traditional,
                           const void* q_norm_weight, const void* k_norm_weight, float norm_eps, int32_t act_dtype, void* workspace,
                           size_t workspace_bytes, void* hip_stream);

/* Single-token attention over a cache that already holds the new token: out[b, h, :] = softmax(q[b, h, :] . K[b, kv(h), 0..*pos, :]^T
 * * sm_scale) V[b, kv(h), 0..*pos, :] with q [batch, n_heads, head_dim] already rotated, caches [batch, n_kv, capacity, head_dim],
 * `pos` a DEVICE int64.  mx.fast.scaled_dot_product_attention as every model graph calls it for L = 1 (models/base.py:65-92),
 * grouped-query heads served from one read of K / V. */
int gbx_attn_cached(const void* q, const void* kcache, const void* vcache, const int64_t* pos, void* out, int32_t batch,
                    int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int32_t capacity, float sm_scale, int32_t act_dtype,
                    void* hip_stream);

/* Causal attention of a prompt chunk: L query rows (positions offset .. offset + L - 1) of every head against the S >= offset + L
 * cached keys; query i sees keys 0 .. offset + i.  out[b, i, h, :] = softmax(q[b, h, i, :] . K[b, kv(h), :, :]^T * sm_scale + mask) V,
 * written as [batch, L, n_heads * head_dim] (the layout o_proj consumes).  q / k / v are addressed through element strides
 * (head_dim contiguous, key rows head_dim apart): q[b, h, i] at q + b * q_stride_b + h * q_stride_h + i * q_stride_l,
 * K[b, g, j] at k + b * k_stride_b + g * k_stride_h + j * head_dim, V likewise.  Replaces the L > 1 call of
 * mx.fast.scaled_dot_product_attention with the mask of create_causal_mask (gbx_lm/models/base.py:27-62, :101-122); grouped-query
 * heads read the shared K / V head.  head_dim 64 or 128; no sliding window. */
int gbx_attn_prefill(const void* q, const void* k, const void* v, void* out, int32_t batch, int32_t n_heads, int32_t n_kv_heads,
                     int32_t L, int32_t S, int32_t head_dim, int64_t q_stride_b, int64_t q_stride_h, int64_t q_stride_l,
                     int64_t k_stride_b, int64_t k_stride_h, int64_t v_stride_b, int64_t v_stride_h, int32_t offset, float sm_scale,
                     int32_t act_dtype, void* hip_stream);

/* Expert routing of `tokens` rows of router logits[tokens, num_experts] on the device: indices[tokens, top_k] (int32, best
 * first, ties to the lowest expert) and mixing weights scores[tokens, top_k] in the activation dtype.
 * mode 0: softmax over all experts (fp32, rounded), top_k largest, divided by their sum if `renormalize`
 *         (gbx_lm/models/qqwen3_moe.py:126-137); mode 1: top_k of the logits, softmax over those (qmixtral.py).
 * num_experts <= 1024, top_k <= 8. */
int gbx_moe_route(const void* logits, int32_t* indices, void* scores, int64_t tokens, int32_t num_experts, int32_t top_k,
                  int32_t mode, int32_t renormalize, int32_t act_dtype, void* hip_stream);

/* out[t, :] = sum_s scores[t, s] * y[t, s, :] with y [tokens, top_k, D]: the weighted sum after SwitchGLU
 * (qqwen3_moe.py:139-141); products rounded to the activation dtype, fp32 sum, one rounding.  D even. */
int gbx_moe_combine(const void* y, const void* scores, void* out, int64_t tokens, int32_t top_k, int32_t D, int32_t act_dtype,
                    void* hip_stream);

/* gbx_moe_combine with the block's residual add fused: out[t, :] = residual[t, :] + (sum_s scores * y) -- `h + self.mlp(...)`
 * of the decoder layer (qqwen3_moe.py:165-170), the sum rounded first, then the add (two roundings, like the two ops). */
int gbx_moe_combine_residual(const void* y, const void* scores, const void* residual, void* out, int64_t tokens, int32_t top_k,
                             int32_t D, int32_t act_dtype, void* hip_stream);

/* kcache[r, *pos, :] = k[r, :], vcache[r, *pos, :] = v[r, :] for rows = batch * kv heads: the single-token
 * cache.update_and_fetch of gbx_lm/models/cache.py:222-244 at a DEVICE-resident position (16-bit elements; k_dim and v_dim
 * may differ: DeepSeek's latent attention). */
int gbx_kv_append(const void* k, const void* v, void* kcache, void* vcache, const int64_t* pos, int64_t rows, int32_t capacity,
                  int32_t k_dim, int32_t v_dim, int32_t act_dtype, void* hip_stream);

/* The tail of one generate_step (gbx_lm/utils.py:305-306 and the default sampler, :285): logprobs[vocab] = logits -
 * logsumexp(logits) (fp32 logsumexp rounded to the activation dtype, activation-dtype subtraction) and *token =
 * argmax(logprobs) (lowest index on ties) -- two small launches, nothing leaves the device.  `workspace`: caller-owned
 * scratch of gbx_logprobs_workspace_bytes() bytes (8-byte aligned). */
size_t gbx_logprobs_workspace_bytes(void);
int gbx_logprobs_argmax(const void* logits, void* logprobs, int64_t* token, void* workspace, int32_t vocab, int32_t act_dtype,
                        void* hip_stream);
/* The same, and -- for a host loop that reads the token id after every step (generate_step, utils.py:323-338) -- packed[0] = token,
 * packed[1..3] = the device words status0..2 (0 for a NULL pointer) as they are when the launch ends: the token id and the decode
 * runner's hand-off status words (gbx_dattn_status / gbx_dblock_status words) leave the device in ONE copy.  (since 1301) */
int gbx_logprobs_argmax_packed(const void* logits, void* logprobs, int64_t* token, void* workspace, int32_t vocab, int32_t act_dtype,
                               const uint32_t* status0, const uint32_t* status1, const uint32_t* status2, int64_t* packed,
                               void* hip_stream);

/* Residual add + RMSNorm in one launch: h[M, K] <- h + delta (one rounding to the activation dtype, the add of
 * qllama.py:138-140), y = RMSNorm(h) * w.  h is updated in place; y may not alias h. */
int gbx_add_rmsnorm(void* h, const void* delta, const void* w, void* y, int64_t M, int32_t K, float eps, int32_t act_dtype,
                    void* hip_stream);

/* y = silu(gate) * up over n elements (n % 8 == 0), rounded like the two activation-dtype ops of qllama.py:115. */
int gbx_swiglu(const void* gate, const void* up, void* y, int64_t n, int32_t act_dtype, void* hip_stream);

/* Rotary embedding of the strided view x[B, H, L, D] (element strides stride_b/h/l, unit feature stride) into the
 * contiguous y[B, H, L, D] (out of place): pair i of the first `dims` features at time step t turns by
 * (offset + t) * scale * inv_freq[i]; pairs are (2i, 2i+1) if traditional else (i, i + dims/2); x is scaled by mscale first
 * (su / YaRN variants); features past dims are copied.  `pos` != NULL: the offset is read from device memory
 * (graph-replayed decode).  mx.fast.rope as the model graphs call it (qllama.py:88-89, qphi3.py:99-103,
 * qdeepseek_v3.py:113-125). */
int gbx_rope(const void* x, void* y, const float* inv_freq, const int64_t* pos, int64_t offset, int64_t B, int32_t H, int32_t L,
             int32_t D, int32_t dims, int64_t stride_b, int64_t stride_h, int64_t stride_l, float scale, float mscale,
             int32_t traditional, int32_t act_dtype, void* hip_stream);

/* RoPE of the keys of a prompt chunk written straight into the key cache, the values copied into theirs -- ``rope(k, offset)``
 * followed by ``cache.update_and_fetch(k, v)`` (qllama.py:88-91, models/cache.py:222-244) for L > 1 tokens in ONE launch.
 * k / v: [B, H, L, D] addressed through element strides (D contiguous: the transposed view of a projection output); caches
 * [B, H, capacity, D] contiguous; rows t0 .. t0 + L - 1 are written, row t rotated by (pos0 + t) * scale * inv_freq[i] with the
 * arithmetic of gbx_rope (full-width RoPE, pairs (i, i + D/2) or, traditional, (2i, 2i+1)). */
int gbx_rope_kv_write(const void* k, const void* v, void* kcache, void* vcache, const float* inv_freq, int64_t pos0, int64_t B,
                      int32_t H, int32_t L, int32_t D, int64_t k_stride_b, int64_t k_stride_h, int64_t k_stride_l, int64_t v_stride_b,
                      int64_t v_stride_h, int64_t v_stride_l, int32_t capacity, int32_t t0, float scale, int32_t traditional,
                      int32_t act_dtype, void* hip_stream);

/* Synchronous batched decode step, every sequence at position *pos (device int64): q[batch, n_heads*head_dim] is rotated
 * in place, k[batch, n_kv*head_dim] is rotated into kcache[b, kvh, *pos, :], v copied into vcache[b, kvh, *pos, :]
 * (caches [batch, n_kv, capacity, head_dim]).  rope(q), rope(k), cache.update_and_fetch of qllama.py:84-91 for one new
 * token per sequence; rope over the full head_dim like gbx_attn_decode. */
int gbx_rope_append(void* q, const void* k, const void* v, void* kcache, void* vcache, const float* inv_freq,
                    const int64_t* pos, int32_t batch, int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int32_t capacity,
                    float rope_scale, int32_t traditional, int32_t act_dtype, void* hip_stream);

/* EMINF test-time logits optimiser (gbx_lm/infer_opt.py:30-77, the second caller of the decode path, :79-142): normalised
 * gradient descent on the entropy of softmax(logits) with accept / reject and step decay (alpha *= 0.65, stop below 1e-4),
 * target entropy max(threshold, 0.1 * H0), num_steps < 0 -> int(3 + 15 r/(1+r)), r = max(0, H0 - threshold)/threshold.
 * One launch, every decision on the device (the reference takes 4-5 host-synced scalars per iteration).
 * logits [V] of logits_dtype (GBX_F16 / GBX_BF16 / GBX_F32); best_logits, scratch: float[V]; out_entropy: float[2] =
 * {best entropy, initial entropy}. */
int gbx_eminf_optimize(const void* logits, int32_t logits_dtype, float* best_logits, float* scratch, float* out_entropy,
                       int32_t V, float alpha, float threshold, int32_t num_steps, void* hip_stream);

/* y[M,K] = RMSNorm(x[M,K]) * w[K] (fp32 statistics, one rounding).  mx.nn.RMSNorm as used at qllama.py:125-128,155. */
int gbx_rmsnorm(const void* x, const void* w, void* y, int64_t M, int32_t K, float eps, int32_t act_dtype,
                void* hip_stream);

/* Which kernel family a gbx_qlinear_forward[_ws] call of this shape dispatches to (introspection for tests and profiles):
 * 0 nothing (M = 0), 1 decode GEMV, 2 generic kernel, 20 batched-decode matrix-core kernels (2 <= M <= 64),
 * 11 / 12 / 13 prefill tile 128 x 128 / 256 x 128 / 256 x 256 (8 waves), 15 prefill with k split over workgroups
 * (has_workspace != 0 and the shape qualifies), 31 / 32 / 33 the same tiles on weights dequantised once per call into the
 * workspace (M >= 4096, or M >= 1024 with at least 176 tiles of 256 x 256, and has_workspace != 0); -1 invalid shape. */
int gbx_qlinear_kernel_id(int64_t M, int32_t N, int32_t K, int32_t bits, int32_t group_size, int32_t has_workspace);

#ifdef __cplusplus
}
#endif
#endif /* GBX_QLINEAR_H */
