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

/* Bytes of caller-provided scratch some shapes of gbx_qlinear_forward may use in later
 * versions; the current kernels need none and this returns 0. */
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

#ifdef __cplusplus
}
#endif
#endif /* GBX_QLINEAR_H */
