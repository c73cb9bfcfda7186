// Gate prologue shared by the GLA decode and prefill kernels: turns the raw
// q/k/g projections of one (token, head) channel into the (q*scale, k, exp-ready g)
// triple the recurrence consumes.
#pragma once

#include "common.h"

namespace hymeta {

struct GateParams {
  int mode;          // enum hymeta_gate_mode
  float scale;       // applied to q
  float normalizer;  // HYMETA_GATE_7B
  float clamp;       // HYMETA_GATE_70B_CLAMP
};

// bf16 * bf16 scale as the single-token kernel of the reference forms it
// (vllm_hymeta/models/my_fused_recurrent.py:59: a tensor-times-scalar product is computed in the
// tensor's dtype, so the scalar is cast to bf16 first and the product is rounded to bf16; the
// widening to fp32 at :75 comes after).  ``scale_bf16`` is the scale already rounded.
__device__ __forceinline__ float scaled_query_bf16(float q, float scale_bf16) { return bf16_round(q * scale_bf16); }

// Decode-step form.  q_raw, k_raw: the bf16 projections widened to fp32; g_raw: the gate input
// widened to fp32 (ignored by the 70B modes); lb: lower bound of this channel; gp.scale: the
// scale rounded to bf16.  Outputs: qs = feature-mapped q times scale, rounded to bf16 (as fp32);
// kb = k rounded to bf16 (as fp32); g = log-decay (fp32, <= 0).
__device__ __forceinline__ void apply_gate(const GateParams& gp, float q_raw, float k_raw,
                                           float g_raw, float lb, float& qs, float& kb,
                                           float& g) {
  switch (gp.mode) {
    case HYMETA_GATE_7B: {
      qs = scaled_query_bf16(fmaxf(q_raw, 0.f), gp.scale);
      kb = fmaxf(k_raw, 0.f);
      // logsigmoid is evaluated on a bf16 tensor and divided in bf16.
      g = bf16_round(bf16_round(logsigmoidf(g_raw)) / gp.normalizer);
      break;
    }
    case HYMETA_GATE_70B_CLAMP: {
      qs = scaled_query_bf16(fmaxf(q_raw, 0.f), gp.scale);
      float kt = fminf(sigmoidf(k_raw), gp.clamp);
      g = __logf(1.f - kt);
      kb = bf16_round(kt);
      break;
    }
    case HYMETA_GATE_70B_LOWER_BOUND: {
      qs = scaled_query_bf16(fmaxf(q_raw, 0.f), gp.scale);
      float kt = sigmoidf(k_raw);
      float gl = 1.f - (1.f - lb) * kt;
      g = __logf(gl);
      kb = bf16_round(1.f - gl);
      break;
    }
    default: {
      qs = scaled_query_bf16(q_raw, gp.scale);
      kb = k_raw;
      g = g_raw;
      break;
    }
  }
}

// log(1 + e) for e in [0, 1]: series below 1/16 (error < e^5/5), v_log_f32 above; both well
// inside half a bf16 ulp of the result, which is what the callers round to.
__device__ __forceinline__ float log_normal(float x) {   // natural log of a positive normal number
  return __builtin_amdgcn_logf(x) * 0.6931471805599453f;
}
__device__ __forceinline__ float log1p_unit(float e) {
  const float series = e * (1.f + e * (-0.5f + e * (1.f / 3.f - 0.25f * e)));
  const float big = log_normal(1.f + e);
  // bitwise blend: keeps the two short paths straight-line instead of a divergent branch
  const uint32_t m = e < 0.0625f ? 0xffffffffu : 0u;
  return __uint_as_float((__float_as_uint(series) & m) | (__float_as_uint(big) & ~m));
}

// Branch-free, compile-time-mode form of apply_gate for the prefill producers, where the gate
// math is the VALU budget: same formulas, except that q * scale stays in fp32 (the chunked kernel
// of flash-linear-attention that the reference calls for prefill is third party: its rounding
// points are not known here); the sigmoid uses the hardware reciprocal (1 ulp) and
// the 7B normaliser is applied as a multiplication by 1/normalizer (exact for powers of two,
// 16 in every published configuration).
template <int GATE>
__device__ __forceinline__ void apply_gate_fast(const GateParams& gp, float inv_normalizer, float q_raw, float k_raw,
                                                float g_raw, float lb, float& qs, float& kb, float& g) {
  if constexpr (GATE == HYMETA_GATE_7B) {
    qs = fmaxf(q_raw, 0.f) * gp.scale;
    kb = fmaxf(k_raw, 0.f);
    const float ls = fminf(g_raw, 0.f) - log1p_unit(__expf(-fabsf(g_raw)));
    g = bf16_round(bf16_round(ls) * inv_normalizer);
  } else if constexpr (GATE == HYMETA_GATE_70B_CLAMP) {
    qs = fmaxf(q_raw, 0.f) * gp.scale;
    const float kt = fminf(__builtin_amdgcn_rcpf(1.f + __expf(-k_raw)), gp.clamp);
    g = log_normal(1.f - kt);
    kb = bf16_round(kt);
  } else if constexpr (GATE == HYMETA_GATE_70B_LOWER_BOUND) {
    qs = fmaxf(q_raw, 0.f) * gp.scale;
    const float kt = __builtin_amdgcn_rcpf(1.f + __expf(-k_raw));
    const float gl = 1.f - (1.f - lb) * kt;
    g = log_normal(gl);
    kb = bf16_round(1.f - gl);
  } else {
    qs = q_raw * gp.scale;
    kb = k_raw;
    g = g_raw;
  }
}

}  // namespace hymeta
