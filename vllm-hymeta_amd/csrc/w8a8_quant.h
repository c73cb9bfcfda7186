// Dynamic per-token int8 quantisation of one activation row, shared by every kernel that produces W8A8 inputs (the
// stand-alone quantise launch of the linears and the block norms that quantise their own output) so that they agree bit
// for bit: s = max|x| / 127 (at least 1e-10 / 127), xq = clamp(rint(x * (1 / s)), -128, 127).
#pragma once

#include "common.h"

namespace hymeta {

// The row held in registers: NC chunks of 8 values per thread (chunks t, t + 256, ...), every load issued before the
// first use (a loop over the chunks waits one cache round trip per chunk in each of its two passes), one read of x --
// or none, when the caller (a norm that has just formed the row) hands its registers over: quantise_chunks_int8.
template <int NC>
__device__ __forceinline__ void quantise_chunks_int8(const u32x4 (&v)[NC], int chunks, int8_t* q, float* scale_out, float* s_red) {
  float mx = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i)
    if (static_cast<int>(threadIdx.x) + 256 * i < chunks) {
#pragma unroll
      for (int j = 0; j < 4; ++j) mx = fmaxf(mx, fmaxf(fabsf(lo_bf(v[i][j])), fabsf(hi_bf(v[i][j]))));
    }
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
  __syncthreads();                       // s_red may still be read by the caller's previous reduction
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
  const float scale = fmaxf(mx, 1e-10f) / 127.f;
  if (threadIdx.x == 0) *scale_out = scale;
  const float inv = 1.f / scale;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int ch = threadIdx.x + 256 * i;
    if (ch >= chunks) continue;
    uint32_t w[2] = {0u, 0u};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float f[2] = {lo_bf(v[i][k]), hi_bf(v[i][k])};
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int qi = static_cast<int>(fminf(fmaxf(rintf(f[e] * inv), -128.f), 127.f));
        const int j = 2 * k + e;
        w[j >> 2] |= (static_cast<uint32_t>(qi) & 0xffu) << (8 * (j & 3));
      }
    }
    *reinterpret_cast<u32x2*>(q + 8 * ch) = u32x2{w[0], w[1]};
  }
}

template <int NC>
__device__ __forceinline__ void quantise_row_int8_regs(const uint16_t* x, int chunks, int8_t* q, float* scale_out, float* s_red) {
  u32x4 v[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) v[i] = ld16(x + 8 * min(static_cast<int>(threadIdx.x) + 256 * i, chunks - 1));
  quantise_chunks_int8<NC>(v, chunks, q, scale_out, s_red);
}

// One workgroup of 256 threads per row; thread t handles the 8-element chunks t, t + 256, ... in both passes (so a row
// the same threads wrote just before needs no fence).  s_red: 4 floats of shared memory.
__device__ __forceinline__ void quantise_row_int8(const uint16_t* x, int K, int8_t* q, float* scale_out, float* s_red) {
  const int chunks = K / 8;
  if (chunks <= 512) return quantise_row_int8_regs<2>(x, chunks, q, scale_out, s_red);          // up to 4096 values
  if (chunks <= 3072) return quantise_row_int8_regs<12>(x, chunks, q, scale_out, s_red);        // up to 24576
  float mx = 0.f;
  for (int ch = threadIdx.x; ch < chunks; ch += 256) {
    const u32x4 v = ld16(x + 8 * ch);
#pragma unroll
    for (int i = 0; i < 4; ++i) mx = fmaxf(mx, fmaxf(fabsf(lo_bf(v[i])), fabsf(hi_bf(v[i]))));
  }
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
  __syncthreads();                       // s_red may still be read by the caller's previous reduction
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
  const float scale = fmaxf(mx, 1e-10f) / 127.f;
  if (threadIdx.x == 0) *scale_out = scale;
  const float inv = 1.f / scale;
  for (int ch = threadIdx.x; ch < chunks; ch += 256) {
    const u32x4 v = ld16(x + 8 * ch);
    uint32_t w[2] = {0u, 0u};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float f[2] = {lo_bf(v[i]), hi_bf(v[i])};
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int qi = static_cast<int>(fminf(fmaxf(rintf(f[e] * inv), -128.f), 127.f));
        const int j = 2 * i + e;
        w[j >> 2] |= (static_cast<uint32_t>(qi) & 0xffu) << (8 * (j & 3));
      }
    }
    *reinterpret_cast<u32x2*>(q + 8 * ch) = u32x2{w[0], w[1]};
  }
}

}  // namespace hymeta
