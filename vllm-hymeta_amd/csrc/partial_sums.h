// Reading a K-sliced linear's result straight from its partial sums (slabs [slice][M][N] of fp32, or int32 for W8A8): the
// slices are added in slice order, W8A8 scales and the bias applied, and the sum rounded to bf16 -- the bits the linear's
// own reduce launch would have written.  Used by the consumers that take the place of that launch.
#pragma once

#include "common.h"

namespace hymeta {

// 8 consecutive columns of row t (col a multiple of 8), as bf16-rounded floats.  Four slices' loads are in flight at a
// time (a loop of dependent load-add pairs would wait out one cache latency per slice).
template <bool I8>
__device__ __forceinline__ void partial_row8(const hymeta_gemm_partials& p, int64_t t, int col, float (&x)[8]) {
  const int64_t slab = static_cast<int64_t>(p.M) * p.N;
  const uint16_t* bias = static_cast<const uint16_t*>(p.bias);
  float b[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (bias != nullptr) {
    const u32x4 bw = ld16(bias + col);
#pragma unroll
    for (int i = 0; i < 4; ++i) b[2 * i] = lo_bf(bw[i]), b[2 * i + 1] = hi_bf(bw[i]);
  }
  if constexpr (I8) {
    const int32_t* src = static_cast<const int32_t*>(p.slabs) + t * p.N + col;
    int acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int sp = 0; sp < p.splits; sp += 4) {
      u32x4 lo[4], hi[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t off = static_cast<int64_t>(min(sp + i, p.splits - 1)) * slab;
        lo[i] = ld16(src + off), hi[i] = ld16(src + off + 4);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (sp + i < p.splits) {
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] += static_cast<int>(lo[i][e]), acc[4 + e] += static_cast<int>(hi[i][e]);
        }
    }
    const float xs = p.x_scale[t];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = bf16_round(w8a8_scaled(acc[e], xs, p.w_scale[col + e], b[e]));
  } else {
    const float* src = static_cast<const float*>(p.slabs) + t * p.N + col;
    f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
    for (int sp = 0; sp < p.splits; sp += 4) {
      f32x4 l4[4], h4[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t off = static_cast<int64_t>(min(sp + i, p.splits - 1)) * slab;
        l4[i] = *reinterpret_cast<const f32x4*>(src + off), h4[i] = *reinterpret_cast<const f32x4*>(src + off + 4);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (sp + i < p.splits) lo += l4[i], hi += h4[i];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) x[e] = bf16_round(lo[e] + b[e]), x[4 + e] = bf16_round(hi[e] + b[4 + e]);
  }
}

// 4 consecutive columns of row t (fp32 slabs only; col a multiple of 4), as bf16-rounded floats.
__device__ __forceinline__ f32x4 partial_row4(const hymeta_gemm_partials& p, int64_t t, int col) {
  const int64_t slab = static_cast<int64_t>(p.M) * p.N;
  const float* src = static_cast<const float*>(p.slabs) + t * p.N + col;
  f32x4 sum = {0.f, 0.f, 0.f, 0.f};
  for (int sp = 0; sp < p.splits; sp += 4) {
    f32x4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const f32x4*>(src + static_cast<int64_t>(min(sp + i, p.splits - 1)) * slab);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (sp + i < p.splits) sum += v[i];
  }
  if (p.bias != nullptr) {
    const u32x2 bw = *reinterpret_cast<const u32x2*>(static_cast<const uint16_t*>(p.bias) + col);
    sum += f32x4{lo_bf(bw[0]), hi_bf(bw[0]), lo_bf(bw[1]), hi_bf(bw[1])};
  }
  return f32x4{bf16_round(sum[0]), bf16_round(sum[1]), bf16_round(sum[2]), bf16_round(sum[3])};
}

// One column of row t (fp32 slabs only), as a bf16-rounded float.
__device__ __forceinline__ float partial_elem(const hymeta_gemm_partials& p, int64_t t, int col) {
  const int64_t slab = static_cast<int64_t>(p.M) * p.N;
  const float* src = static_cast<const float*>(p.slabs) + t * p.N + col;
  float sum = 0.f;
  for (int sp = 0; sp < p.splits; sp += 4) {
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = src[static_cast<int64_t>(min(sp + i, p.splits - 1)) * slab];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (sp + i < p.splits) sum += v[i];
  }
  if (p.bias != nullptr) sum += bf2f(static_cast<const uint16_t*>(p.bias)[col]);
  return bf16_round(sum);
}

}  // namespace hymeta
