// Mixture-of-experts glue of a decode step as single launches (callers of QuantizedSwitchLinear / mx.gather_qmm):
//   * moe_route   : router logits -> the k selected experts and their mixing weights
//                   mode 0 (gbx_lm/models/qqwen3_moe.py:126-137): softmax over all experts (fp32, rounded to the activation
//                   dtype), k largest, optional renormalisation by their sum;
//                   mode 1 (gbx_lm/models/qmixtral.py:  top-k of the logits, softmax over those k).
//   * moe_combine : y[t, :] = sum_s scores[t, s] * y_s[t, s, :]  (the weighted sum after SwitchGLU, qqwen3_moe.py:139-141),
//                   products rounded to the activation dtype like the elementwise multiply, fp32 sum, one rounding.
// One wavefront per token for the router (E <= 1024 experts); the decisions never leave the device.
#include "gbx_device.h"
#include "kernels.h"

namespace gbx {

__device__ __forceinline__ float wave_max_f32(float v) {
  v = fmaxf(v, dpp_f32<0xB1>(v));
  v = fmaxf(v, dpp_f32<0x4E>(v));
  v = fmaxf(v, dpp_f32<0x141>(v));
  v = fmaxf(v, dpp_f32<0x140>(v));
  v = fmaxf(v, __shfl_xor(v, 16));
  v = fmaxf(v, __shfl_xor(v, 32));
  return v;
}

template <bool BF16>
__global__ __launch_bounds__(64) void moe_route_kernel(const unsigned short* __restrict__ logits, int* __restrict__ inds,
                                                       unsigned short* __restrict__ scores, int E, int k, int mode, int renorm) {
  constexpr int VPL = 16;                           // experts per lane: E <= 1024
  const int lane = threadIdx.x;
  const unsigned short* row = logits + static_cast<size_t>(blockIdx.x) * E;
  float v[VPL];
  float mx = -3.0e38f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int e = lane + 64 * i;
    v[i] = e < E ? act_to_f32<BF16>(row[e]) : -3.0e38f;
    mx = fmaxf(mx, v[i]);
  }
  if (mode == 0) {                                  // probabilities over all experts, rounded to the activation dtype
    mx = wave_max_f32(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      v[i] = lane + 64 * i < E ? __expf(v[i] - mx) : 0.f;
      sum += v[i];
    }
    sum = wave_sum(sum);
#pragma unroll
    for (int i = 0; i < VPL; ++i) v[i] = lane + 64 * i < E ? act_to_f32<BF16>(f32_to_act<BF16>(v[i] / sum)) : -3.0e38f;
  }
  // k rounds of wave-wide argmax (ties: lowest expert index)
  float top[8];
  int topi[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    if (s < k) {
      float best = -3.0e38f;
      int bi = 0x7fffffff;
#pragma unroll
      for (int i = 0; i < VPL; ++i)
        if (v[i] > best) { best = v[i]; bi = lane + 64 * i; }
      const float wbest = wave_max_f32(best);
      unsigned cand = best == wbest ? ~static_cast<unsigned>(bi) : 0u;     // larger key = smaller index
      cand = wave_max_u32(cand);
      const int win = static_cast<int>(~cand);
      top[s] = wbest;
      topi[s] = win;
#pragma unroll
      for (int i = 0; i < VPL; ++i)
        if (lane + 64 * i == win) v[i] = -3.0e38f;
    }
  }
  if (lane == 0) {
    float w[8];
    if (mode == 0) {
      float sum = 0.f;
      for (int s = 0; s < k; ++s) sum += top[s];
      const float sr = act_to_f32<BF16>(f32_to_act<BF16>(sum));           // the activation-dtype `scores.sum()`
      for (int s = 0; s < k; ++s) w[s] = renorm ? top[s] / sr : top[s];
    } else {
      float sum = 0.f;
      for (int s = 0; s < k; ++s) { w[s] = __expf(top[s] - top[0]); sum += w[s]; }
      for (int s = 0; s < k; ++s) w[s] /= sum;
    }
    for (int s = 0; s < k; ++s) {
      inds[static_cast<size_t>(blockIdx.x) * k + s] = topi[s];
      scores[static_cast<size_t>(blockIdx.x) * k + s] = f32_to_act<BF16>(w[s]);
    }
  }
}

hipError_t launch_moe_route(const void* logits, int* inds, void* scores, int T, int E, int k, int mode, int renorm, bool bf16,
                            hipStream_t s) {
  auto L = static_cast<const unsigned short*>(logits);
  auto S = static_cast<unsigned short*>(scores);
  if (bf16) hipLaunchKernelGGL((moe_route_kernel<true>), dim3(T), dim3(64), 0, s, L, inds, S, E, k, mode, renorm);
  else hipLaunchKernelGGL((moe_route_kernel<false>), dim3(T), dim3(64), 0, s, L, inds, S, E, k, mode, renorm);
  return hipGetLastError();
}

template <bool BF16>
__global__ __launch_bounds__(256) void moe_combine_kernel(const unsigned short* __restrict__ y, const unsigned short* __restrict__ scores,
                                                          unsigned short* __restrict__ out, int k, int D) {
  const size_t t = blockIdx.y;
  const int d = (blockIdx.x * 256 + threadIdx.x) * 2;
  if (d >= D) return;
  float a0 = 0.f, a1 = 0.f;
  for (int s = 0; s < k; ++s) {
    const float w = act_to_f32<BF16>(scores[t * k + s]);
    const unsigned v = *reinterpret_cast<const unsigned*>(y + (t * k + s) * D + d);
    a0 += act_to_f32<BF16>(f32_to_act<BF16>(w * act_lo<BF16>(v)));
    a1 += act_to_f32<BF16>(f32_to_act<BF16>(w * act_hi<BF16>(v)));
  }
  *reinterpret_cast<unsigned*>(out + t * D + d) =
      static_cast<unsigned>(f32_to_act<BF16>(a0)) | (static_cast<unsigned>(f32_to_act<BF16>(a1)) << 16);
}

hipError_t launch_moe_combine(const void* y, const void* scores, void* out, int T, int k, int D, bool bf16, hipStream_t s) {
  auto Y = static_cast<const unsigned short*>(y);
  auto S = static_cast<const unsigned short*>(scores);
  auto O = static_cast<unsigned short*>(out);
  dim3 grid((D / 2 + 255) / 256, T), block(256);
  if (bf16) hipLaunchKernelGGL((moe_combine_kernel<true>), grid, block, 0, s, Y, S, O, k, D);
  else hipLaunchKernelGGL((moe_combine_kernel<false>), grid, block, 0, s, Y, S, O, k, D);
  return hipGetLastError();
}

}  // namespace gbx
