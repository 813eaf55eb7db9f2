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

template <bool BF16, int VPL>
__global__ __launch_bounds__(64) void moe_route_kernel(const unsigned short* __restrict__ logits, int* __restrict__ inds,
                                                       unsigned short* __restrict__ scores, int E, int k, int mode, int renorm) {
  const int lane = threadIdx.x;                     // expert e = lane + 64 i, i < VPL (E <= 64 VPL)
  const unsigned short* row = logits + static_cast<size_t>(blockIdx.x) * E;
  constexpr float NEG = -3.0e38f;
  float v[VPL];
  float mx = NEG;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int e = lane + 64 * i;
    v[i] = e < E ? act_to_f32<BF16>(row[e]) : NEG;
    mx = fmaxf(mx, v[i]);
  }
  if (mode == 0) {                                  // probabilities over all experts, rounded to the activation dtype
    mx = key2f(wave_max_u32(f2key(mx)));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      v[i] = lane + 64 * i < E ? __expf(v[i] - mx) : 0.f;
      sum += v[i];
    }
    sum = wave_sum(sum);
#pragma unroll
    for (int i = 0; i < VPL; ++i) v[i] = lane + 64 * i < E ? act_to_f32<BF16>(f32_to_act<BF16>(v[i] / sum)) : NEG;
  }
  // k rounds of wave-wide argmax (ties: lowest expert index)
  float top[8];
  int topi[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    if (s < k) {
      float best = NEG;
      int bi = 0x7fffffff;
#pragma unroll
      for (int i = 0; i < VPL; ++i)
        if (v[i] > best) { best = v[i]; bi = lane + 64 * i; }
      const unsigned kb = f2key(best);
      const unsigned wk = wave_max_u32(kb);
      const unsigned cand = wave_max_u32(kb == wk ? ~static_cast<unsigned>(bi) : 0u);     // larger key = smaller index
      const int win = static_cast<int>(~cand);
      top[s] = key2f(wk);
      topi[s] = win;
#pragma unroll
      for (int i = 0; i < VPL; ++i)
        if (lane + 64 * i == win) v[i] = NEG;
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

template <int VPL>
static void launch_route_vpl(const unsigned short* L, int* inds, unsigned short* S, int T, int E, int k, int mode, int renorm,
                             bool bf16, hipStream_t s) {
  if (bf16) hipLaunchKernelGGL((moe_route_kernel<true, VPL>), dim3(T), dim3(64), 0, s, L, inds, S, E, k, mode, renorm);
  else hipLaunchKernelGGL((moe_route_kernel<false, VPL>), dim3(T), dim3(64), 0, s, L, inds, S, E, k, mode, renorm);
}

hipError_t launch_moe_route(const void* logits, int* inds, void* scores, int T, int E, int k, int mode, int renorm, bool bf16,
                            hipStream_t s) {
  auto L = static_cast<const unsigned short*>(logits);
  auto S = static_cast<unsigned short*>(scores);
  if (E <= 64) launch_route_vpl<1>(L, inds, S, T, E, k, mode, renorm, bf16, s);
  else if (E <= 128) launch_route_vpl<2>(L, inds, S, T, E, k, mode, renorm, bf16, s);
  else if (E <= 256) launch_route_vpl<4>(L, inds, S, T, E, k, mode, renorm, bf16, s);
  else launch_route_vpl<16>(L, inds, S, T, E, k, mode, renorm, bf16, s);
  return hipGetLastError();
}

template <bool BF16>
__global__ __launch_bounds__(256) void moe_combine_kernel(const unsigned short* __restrict__ y, const unsigned short* __restrict__ scores,
                                                          unsigned short* __restrict__ out, int k, int D,
                                                          const unsigned short* __restrict__ residual) {
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
  if (residual) {                                   // h + mlp(...): the block's residual add as a second rounding
    const unsigned r = *reinterpret_cast<const unsigned*>(residual + t * D + d);
    a0 = act_lo<BF16>(r) + act_to_f32<BF16>(f32_to_act<BF16>(a0));
    a1 = act_hi<BF16>(r) + act_to_f32<BF16>(f32_to_act<BF16>(a1));
  }
  *reinterpret_cast<unsigned*>(out + t * D + d) =
      static_cast<unsigned>(f32_to_act<BF16>(a0)) | (static_cast<unsigned>(f32_to_act<BF16>(a1)) << 16);
}

hipError_t launch_moe_combine(const void* y, const void* scores, void* out, int T, int k, int D, bool bf16, hipStream_t s,
                              const void* residual) {
  auto Y = static_cast<const unsigned short*>(y);
  auto S = static_cast<const unsigned short*>(scores);
  auto O = static_cast<unsigned short*>(out);
  auto R = static_cast<const unsigned short*>(residual);
  dim3 grid((D / 2 + 255) / 256, T), block(256);
  if (bf16) hipLaunchKernelGGL((moe_combine_kernel<true>), grid, block, 0, s, Y, S, O, k, D, R);
  else hipLaunchKernelGGL((moe_combine_kernel<false>), grid, block, 0, s, Y, S, O, k, D, R);
  return hipGetLastError();
}

// One new token per sequence into both caches at the device-resident position (the static-mode
// cache.update_and_fetch of gbx_lm/models/cache.py:222-244): rows = batch * kv heads.
__global__ __launch_bounds__(256) void kv_append_kernel(const unsigned short* __restrict__ k, const unsigned short* __restrict__ v,
                                                        unsigned short* __restrict__ kc, unsigned short* __restrict__ vc,
                                                        const long long* __restrict__ posp, int cap, int dk, int dv) {
  const long long pos = *posp;
  const size_t r = blockIdx.x;
  for (int i = threadIdx.x; i < dk; i += 256) kc[(r * cap + pos) * dk + i] = k[r * dk + i];
  for (int i = threadIdx.x; i < dv; i += 256) vc[(r * cap + pos) * dv + i] = v[r * dv + i];
}

hipError_t launch_kv_append(const void* k, const void* v, void* kc, void* vc, const long long* pos, int rows, int cap, int dk, int dv,
                            hipStream_t s) {
  hipLaunchKernelGGL(kv_append_kernel, dim3(rows), dim3(256), 0, s, static_cast<const unsigned short*>(k),
                     static_cast<const unsigned short*>(v), static_cast<unsigned short*>(kc), static_cast<unsigned short*>(vc), pos,
                     cap, dk, dv);
  return hipGetLastError();
}

}  // namespace gbx
