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

// q_raw, k_raw: the bf16 projections widened to fp32; g_raw: the gate input
// widened to fp32 (ignored by the 70B modes); lb: lower bound of this channel.
// Outputs: qs = feature-mapped q times scale (fp32); kb = k rounded to bf16
// (as fp32); g = log-decay (fp32, <= 0).
__device__ __forceinline__ void apply_gate(const GateParams& gp, float q_raw, float k_raw,
                                           float g_raw, float lb, float& qs, float& kb,
                                           float& g) {
  switch (gp.mode) {
    case HYMETA_GATE_7B: {
      qs = fmaxf(q_raw, 0.f) * gp.scale;
      kb = fmaxf(k_raw, 0.f);
      // logsigmoid is evaluated on a bf16 tensor and divided in bf16.
      g = bf16_round(bf16_round(logsigmoidf(g_raw)) / gp.normalizer);
      break;
    }
    case HYMETA_GATE_70B_CLAMP: {
      qs = fmaxf(q_raw, 0.f) * gp.scale;
      float kt = fminf(sigmoidf(k_raw), gp.clamp);
      g = __logf(1.f - kt);
      kb = bf16_round(kt);
      break;
    }
    case HYMETA_GATE_70B_LOWER_BOUND: {
      qs = fmaxf(q_raw, 0.f) * gp.scale;
      float kt = sigmoidf(k_raw);
      float gl = 1.f - (1.f - lb) * kt;
      g = __logf(gl);
      kb = bf16_round(1.f - gl);
      break;
    }
    default: {
      qs = q_raw * gp.scale;
      kb = k_raw;
      g = g_raw;
      break;
    }
  }
}

}  // namespace hymeta
