// First linear of the 7B low-rank gate projection (hidden -> 16, no bias) for ONE token, by a workgroup of 256 threads:
// 16 dot products over the hidden size (w0 is 16 rows, read by every workgroup and served by the L2), summed over the
// workgroup and rounded to bf16 as the linear's output is.  Shared by the stand-alone gate kernel and the recurrence
// kernel so that both produce the same bits.
#pragma once

#include "common.h"

namespace hymeta {

// x: the token's row (bf16, hidden values); w0: bf16 [16, hidden] contiguous; s_part: shared [4][16]; s_t: shared [16],
// holds the result after the call (which ends on a barrier).  hidden a multiple of 8, both pointers 16-byte aligned.
__device__ __forceinline__ void lowrank_first_linear(const uint16_t* x, const uint16_t* w0, int hidden, float (*s_part)[16], float* s_t) {
  constexpr int R = 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float acc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = 0.f;
  // two chunks of 8 values per trip, every load of both (2 of x, 32 of w0) issued before the first is used: left to
  // itself the compiler waits for each w0 load in turn, 32 L2 round trips in a row
  const int chunks = hidden / 8;
  for (int ch = tid; ch < chunks; ch += 512) {
    const bool two = ch + 256 < chunks;
    const int ch1 = two ? ch + 256 : ch;
    const u32x4 xv0 = ld16(x + 8 * ch), xv1 = ld16(x + 8 * ch1);
    u32x4 wv0[R], wv1[R];
#pragma unroll
    for (int r = 0; r < R; ++r) wv0[r] = ld16(w0 + static_cast<int64_t>(r) * hidden + 8 * ch);
#pragma unroll
    for (int r = 0; r < R; ++r) wv1[r] = ld16(w0 + static_cast<int64_t>(r) * hidden + 8 * ch1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[r] = fmaf(lo_bf(xv0[i]), lo_bf(wv0[r][i]), acc[r]);
        acc[r] = fmaf(hi_bf(xv0[i]), hi_bf(wv0[r][i]), acc[r]);
      }
    }
    if (two) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc[r] = fmaf(lo_bf(xv1[i]), lo_bf(wv1[r][i]), acc[r]);
          acc[r] = fmaf(hi_bf(xv1[i]), hi_bf(wv1[r][i]), acc[r]);
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) acc[r] += __shfl_xor(acc[r], off);
  }
  if (lane == 0) {
#pragma unroll
    for (int r = 0; r < R; ++r) s_part[wave][r] = acc[r];
  }
  __syncthreads();
  if (tid < R) s_t[tid] = bf16_round(s_part[0][tid] + s_part[1][tid] + s_part[2][tid] + s_part[3][tid]);
  __syncthreads();
}

}  // namespace hymeta
