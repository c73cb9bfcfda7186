// Reading a K-sliced linear's result straight from its partial sums (slabs [slice][M][N] of fp32, or int32 for W8A8): the
// slices are added in slice order, W8A8 scales and the bias applied, and the sum rounded to bf16 -- the bits the linear's
// own reduce launch would have written.  Used by the consumers that take the place of that launch.
#pragma once

#include <type_traits>

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

// partial_row8 for two column groups of the same row at once, every load of a batch of four slices of both in flight
// together (two calls in a row are not interleaved by the compiler: the second's loads wait behind the first's adds).
template <bool I8>
__device__ __forceinline__ void partial_row8_x2(const hymeta_gemm_partials& p, int64_t t, int col_a, int col_b, float (&xa)[8], float (&xb)[8]) {
  const int64_t slab = static_cast<int64_t>(p.M) * p.N;
  const uint16_t* bias = static_cast<const uint16_t*>(p.bias);
  float ba[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, bb[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  u32x4 bwa = {0, 0, 0, 0}, bwb = {0, 0, 0, 0};
  if (bias != nullptr) bwa = ld16(bias + col_a), bwb = ld16(bias + col_b);
  const char* src_a = static_cast<const char*>(p.slabs) + (t * p.N + col_a) * 4;
  const char* src_b = static_cast<const char*>(p.slabs) + (t * p.N + col_b) * 4;
  typedef typename std::conditional<I8, i32x4, f32x4>::type vec_t;
  vec_t lo_a = {0, 0, 0, 0}, hi_a = {0, 0, 0, 0}, lo_b = {0, 0, 0, 0}, hi_b = {0, 0, 0, 0};
  for (int sp = 0; sp < p.splits; sp += 4) {
    vec_t la[4], ha[4], lb[4], hb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t off = static_cast<int64_t>(min(sp + i, p.splits - 1)) * slab * 4;
      la[i] = *reinterpret_cast<const vec_t*>(src_a + off), ha[i] = *reinterpret_cast<const vec_t*>(src_a + off + 16);
      lb[i] = *reinterpret_cast<const vec_t*>(src_b + off), hb[i] = *reinterpret_cast<const vec_t*>(src_b + off + 16);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (sp + i < p.splits) lo_a += la[i], hi_a += ha[i], lo_b += lb[i], hi_b += hb[i];
  }
  if (bias != nullptr) {
#pragma unroll
    for (int i = 0; i < 4; ++i) ba[2 * i] = lo_bf(bwa[i]), ba[2 * i + 1] = hi_bf(bwa[i]), bb[2 * i] = lo_bf(bwb[i]), bb[2 * i + 1] = hi_bf(bwb[i]);
  }
  if constexpr (I8) {
    const float xs = p.x_scale[t];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      xa[e] = bf16_round(w8a8_scaled(lo_a[e], xs, p.w_scale[col_a + e], ba[e]));
      xa[4 + e] = bf16_round(w8a8_scaled(hi_a[e], xs, p.w_scale[col_a + 4 + e], ba[4 + e]));
      xb[e] = bf16_round(w8a8_scaled(lo_b[e], xs, p.w_scale[col_b + e], bb[e]));
      xb[4 + e] = bf16_round(w8a8_scaled(hi_b[e], xs, p.w_scale[col_b + 4 + e], bb[4 + e]));
    }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      xa[e] = bf16_round(lo_a[e] + ba[e]), xa[4 + e] = bf16_round(hi_a[e] + ba[4 + e]);
      xb[e] = bf16_round(lo_b[e] + bb[e]), xb[4 + e] = bf16_round(hi_b[e] + bb[4 + e]);
    }
  }
}

// 4 consecutive columns of row t (col a multiple of 4), as bf16-rounded floats.
template <bool I8>
__device__ __forceinline__ f32x4 partial_row4(const hymeta_gemm_partials& p, int64_t t, int col) {
  const int64_t slab = static_cast<int64_t>(p.M) * p.N;
  typedef typename std::conditional<I8, i32x4, f32x4>::type vec_t;
  const char* src = static_cast<const char*>(p.slabs) + (t * p.N + col) * 4;
  vec_t sum = {0, 0, 0, 0};
  for (int sp = 0; sp < p.splits; sp += 4) {
    vec_t v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const vec_t*>(src + static_cast<int64_t>(min(sp + i, p.splits - 1)) * slab * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (sp + i < p.splits) sum += v[i];
  }
  f32x4 b = {0.f, 0.f, 0.f, 0.f};
  if (p.bias != nullptr) {
    const u32x2 bw = *reinterpret_cast<const u32x2*>(static_cast<const uint16_t*>(p.bias) + col);
    b = f32x4{lo_bf(bw[0]), hi_bf(bw[0]), lo_bf(bw[1]), hi_bf(bw[1])};
  }
  f32x4 y;
  if constexpr (I8) {
    const float xs = p.x_scale[t];
    const f32x4 ws = *reinterpret_cast<const f32x4*>(p.w_scale + col);
#pragma unroll
    for (int e = 0; e < 4; ++e) y[e] = w8a8_scaled(sum[e], xs, ws[e], b[e]);
  } else {
    y = sum + b;
  }
  return f32x4{bf16_round(y[0]), bf16_round(y[1]), bf16_round(y[2]), bf16_round(y[3])};
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
