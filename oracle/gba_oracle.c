/* CPU oracle (plain C) for the GBA QuantizedLinear hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Checker, never product: linked/loaded only by tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg.  PARITY UNPINNED for the in-row bit order (the reference
 * delegates to mx.quantized_matmul, third-party mlx>=0.25.0, absent here; see
 * oracle/gba_oracle.py header and SURVEY.md section 8c / Appendix A).
 *
 * Restates:
 *   unpack      SURVEY.md Appendix A formula (the MLX affine layout consumed at
 *               gbx_lm/models/quantized_linear_gba.py:195-203)
 *   dequant     w = q*scale + zeros            quantized_linear_gba.py:151-155 ("+", not "-")
 *   forward     y = x @ w^T (+ bias)           quantized_linear_gba.py:195-205
 *   shapes      qweight [N, K/32*bits] u32, scales/zeros [N, K/g]   quantized_linear_gba.py:61-62
 *
 * scales / zeros / x / bias arrive as 16-bit patterns (bf16 or IEEE fp16) exactly as the
 * loader leaves them (gbx_lm/utils.py:841-843 casts scales/zeros to bf16).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#define GBX_F16 0
#define GBX_BF16 1

static inline float bf16_to_f32(uint16_t h) {
    uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f;
}

static inline float f16_to_f32(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1Fu, man = h & 0x3FFu, u;
    if (exp == 0) {
        if (man == 0) { u = sign; }
        else { /* subnormal */
            float f = (float)man * 5.9604644775390625e-8f; /* 2^-24 */
            memcpy(&u, &f, 4); u |= sign;
        }
    } else if (exp == 31) { u = sign | 0x7F800000u | (man << 13); }
    else { u = sign | ((exp + 112u) << 23) | (man << 13); }
    float f; memcpy(&f, &u, 4); return f;
}

static inline float h_to_f32(uint16_t h, int dt) { return dt == GBX_BF16 ? bf16_to_f32(h) : f16_to_f32(h); }

/* field k of one packed row */
static inline uint32_t field(const uint32_t* row, int64_t k, int bits) {
    int64_t o = k * bits; int64_t wi = o >> 5; int s = (int)(o & 31);
    uint32_t mask = (1u << bits) - 1u;
    uint32_t q = (row[wi] >> s) & mask;
    if (s + bits > 32) q |= (row[wi + 1] << (32 - s)) & mask;   /* only b = 3 */
    return q;
}

int gba_oracle_check(int64_t n, int64_t k, int bits, int group) {
    if (bits != 2 && bits != 3 && bits != 4) return 1;
    if (group != 32 && group != 64 && group != 128) return 2;   /* quantized_linear_gba.py:272 */
    if (k <= 0 || n <= 0 || (k % 32) || (k % group)) return 3;
    return 0;
}

/* q[n*K + k] = field k of row n */
int gba_oracle_unpack(const uint32_t* qweight, int64_t n, int64_t k, int bits, int32_t* q) {
    if (bits != 2 && bits != 3 && bits != 4) return 1;
    int64_t w = k * bits / 32;
    for (int64_t r = 0; r < n; ++r)
        for (int64_t c = 0; c < k; ++c) q[r * k + c] = (int32_t)field(qweight + r * w, c, bits);
    return 0;
}

/* w[n*K + k] = scales[n, k/g] * q + zeros[n, k/g]   (fp32) */
int gba_oracle_dequant(const uint32_t* qweight, const uint16_t* scales, const uint16_t* zeros,
                       int64_t n, int64_t k, int bits, int group, int sz_dtype, float* wout) {
    int rc = gba_oracle_check(n, k, bits, group); if (rc) return rc;
    int64_t w = k * bits / 32, g = k / group;
    for (int64_t r = 0; r < n; ++r)
        for (int64_t c = 0; c < k; ++c) {
            float s = h_to_f32(scales[r * g + c / group], sz_dtype);
            float z = h_to_f32(zeros[r * g + c / group], sz_dtype);
            wout[r * k + c] = (float)field(qweight + r * w, c, bits) * s + z;
        }
    return 0;
}

/* y[m*N + n] = sum_k x[m,k] * w[n,k] (+ bias[n]); dequant in fp32, accumulate in fp32
 * (acc64 = 0) or fp64 (acc64 = 1, the y_ref64 tolerance anchor).  Output is double either way.
 * Rows of the weight are dequantised one at a time (no [N,K] temporary) so the full-size
 * layers fit the CPU baseline leg; OpenMP over output rows when compiled with -fopenmp. */
int gba_oracle_qlinear(const uint16_t* x, const uint32_t* qweight, const uint16_t* scales,
                       const uint16_t* zeros, const uint16_t* bias, double* y,
                       int64_t m, int64_t n, int64_t k, int bits, int group,
                       int act_dtype, int sz_dtype, int acc64) {
    int rc = gba_oracle_check(n, k, bits, group); if (rc) return rc;
    int64_t w = k * bits / 32, g = k / group;
    float* xf = (float*)malloc(sizeof(float) * (size_t)(m * k));
    if (!xf) return 4;
    for (int64_t i = 0; i < m * k; ++i) xf[i] = h_to_f32(x[i], act_dtype);
    int fail = 0;
#pragma omp parallel
    {
        float* wr = (float*)malloc(sizeof(float) * (size_t)k);
        if (!wr) { fail = 1; }
        else {
#pragma omp for schedule(static)
            for (int64_t r = 0; r < n; ++r) {
                const uint32_t* row = qweight + r * w;
                for (int64_t gi = 0; gi < g; ++gi) {          /* one (scale, zero) pair per group of `group` fields */
                    const float s = h_to_f32(scales[r * g + gi], sz_dtype);
                    const float z = h_to_f32(zeros[r * g + gi], sz_dtype);
                    for (int64_t c = gi * group; c < (gi + 1) * group; ++c)
                        wr[c] = (float)field(row, c, bits) * s + z;
                }
                float b = bias ? h_to_f32(bias[r], act_dtype) : 0.f;
                for (int64_t i = 0; i < m; ++i) {
                    const float* xr = xf + i * k;
                    if (acc64) {
                        double a = 0.0;
                        for (int64_t c = 0; c < k; ++c) a += (double)xr[c] * (double)wr[c];
                        y[i * n + r] = a + (double)b;
                    } else {
                        float a = 0.f;
                        for (int64_t c = 0; c < k; ++c) a += xr[c] * wr[c];
                        y[i * n + r] = (double)(a + b);
                    }
                }
            }
            free(wr);
        }
    }
    free(xf);
    return fail ? 4 : 0;
}

int gba_oracle_version(void) { return 1; }
