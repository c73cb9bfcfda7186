// Element-wise glue of the attention layers: NeoX rotary embedding, the 70B
// two-branch RMS-norm combine and the blocks' RMS norm with the residual add.  HBM-bound; 16-byte accesses throughout.
#include "lowrank_gate.h"
#include "common.h"
#include "partial_sums.h"
#include "w8a8_quant.h"

namespace hymeta {

// One workgroup of 256 threads per token; thread i < heads*8 handles 8 pairs (x[8j..], x[64+8j..]).
__global__ __launch_bounds__(256) void rope_neox_kernel(hymeta_rope_args a) {
  const int t = blockIdx.x;
  const int half = a.rot_dim / 2;
  const int chunks = half / 8;                 // 16-byte chunks per half head
  // positions outside the table are the caller's error; they are clamped to its last row rather than read past it
  int64_t pos = a.positions[t] + a.position_offset;
  pos = pos < 0 ? 0 : (pos >= a.max_position ? a.max_position - 1 : pos);
  const uint16_t* cs = reinterpret_cast<const uint16_t*>(a.cos_sin_cache) + pos * a.rot_dim;
  const int total_heads = a.num_heads + (a.k != nullptr ? a.num_kv_heads : 0);
  for (int w = threadIdx.x; w < total_heads * chunks; w += blockDim.x) {
    const int head = w / chunks, ch = w % chunks;
    uint16_t* x = head < a.num_heads
        ? reinterpret_cast<uint16_t*>(a.q) + t * a.q_stride_t + head * a.q_stride_h
        : reinterpret_cast<uint16_t*>(a.k) + t * a.k_stride_t + (head - a.num_heads) * a.k_stride_h;
    const u32x4 xv = ld16(x + 8 * ch), yv = ld16(x + half + 8 * ch);
    const u32x4 cv = ld16(cs + 8 * ch), sv = ld16(cs + half + 8 * ch);
    u32x4 xo, yo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float r[2][2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const float xf = e ? hi_bf(xv[i]) : lo_bf(xv[i]);
        const float yf = e ? hi_bf(yv[i]) : lo_bf(yv[i]);
        const float c = e ? hi_bf(cv[i]) : lo_bf(cv[i]);
        const float s = e ? hi_bf(sv[i]) : lo_bf(sv[i]);
        // bf16 arithmetic: every product and the sum round to bf16
        r[0][e] = bf16_round(xf * c) - bf16_round(yf * s);
        r[1][e] = bf16_round(yf * c) + bf16_round(xf * s);
      }
      xo[i] = pack_bf16x2(r[0][0], r[0][1]);
      yo[i] = pack_bf16x2(r[1][0], r[1][1]);
    }
    st16(x + 8 * ch, xo);
    st16(x + half + 8 * ch, yo);
  }
}

// The fused qkv projection finished from its K slices' partial sums, RoPE included: row t of out [T, N] = the linear's
// result (columns [0, (H + Hkv) D) are the q and k heads and get the rotation, the rest -- v, and whatever rides behind
// it -- is written as it is).  Same arithmetic as the linear's reduce launch followed by rope_neox_kernel.
template <bool I8>
__global__ __launch_bounds__(256) void rope_partials_kernel(hymeta_rope_args a, hymeta_gemm_partials p) {
  const int64_t t = blockIdx.x;
  const int half = a.rot_dim / 2;
  const int chunks = half / 8;
  int64_t pos = a.positions[t] + a.position_offset;
  pos = pos < 0 ? 0 : (pos >= a.max_position ? a.max_position - 1 : pos);
  const uint16_t* cs = reinterpret_cast<const uint16_t*>(a.cos_sin_cache) + pos * a.rot_dim;
  uint16_t* out = reinterpret_cast<uint16_t*>(a.q) + t * a.q_stride_t;
  const int rope_heads = a.num_heads + a.num_kv_heads;
  for (int w = threadIdx.x; w < rope_heads * chunks; w += blockDim.x) {
    const int head = w / chunks, ch = w % chunks;
    const int col = head * a.head_dim + 8 * ch;
    float x[8], y[8];
    const u32x4 cv = ld16(cs + 8 * ch), sv = ld16(cs + half + 8 * ch);
    partial_row8_x2<I8>(p, t, col, col + half, x, y);      // both halves' loads in flight together
    u32x4 xo, yo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float r[2][2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const float xf = x[2 * i + e], yf = y[2 * i + e];
        const float c = e ? hi_bf(cv[i]) : lo_bf(cv[i]);
        const float s = e ? hi_bf(sv[i]) : lo_bf(sv[i]);
        r[0][e] = bf16_round(xf * c) - bf16_round(yf * s);
        r[1][e] = bf16_round(yf * c) + bf16_round(xf * s);
      }
      xo[i] = pack_bf16x2(r[0][0], r[0][1]);
      yo[i] = pack_bf16x2(r[1][0], r[1][1]);
    }
    st16(out + col, xo);
    st16(out + col + half, yo);
  }
  const int first = rope_heads * a.head_dim;
  for (int c = first / 8 + threadIdx.x; c < p.N / 8; c += blockDim.x) {
    float x[8];
    partial_row8<I8>(p, t, 8 * c, x);
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = pack_bf16x2(x[2 * i], x[2 * i + 1]);
    st16(out + 8 * c, o);
  }
}

__device__ __forceinline__ float block_sum_256(float v, float* s_red) {
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) v += __shfl_xor(v, off);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) s_red[wave] = v;
  __syncthreads();
  const float r = s_red[0] + s_red[1] + s_red[2] + s_red[3];
  __syncthreads();
  return r;
}

// One workgroup per token.  MODE 0: write the two local mean-squares.  MODE 1: combine.
template <int MODE>
__global__ __launch_bounds__(256) void hybrid_combine_kernel(hymeta_combine_args a) {
  __shared__ float s_red[4];
  const int t = blockIdx.x;
  const uint16_t* xa = reinterpret_cast<const uint16_t*>(a.o_attn) + t * a.attn_stride_t;
  const uint16_t* xl = reinterpret_cast<const uint16_t*>(a.o_lin) + t * a.lin_stride_t;
  const int chunks = a.hidden_local / 8;
  float var1, var2;
  if (MODE == 0 || a.stats == nullptr) {
    float s1 = 0.f, s2 = 0.f;
    for (int c = threadIdx.x; c < chunks; c += blockDim.x) {
      const u32x4 v1 = ld16(xa + 8 * c), v2 = ld16(xl + 8 * c);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float a0 = lo_bf(v1[i]), a1 = hi_bf(v1[i]), b0 = lo_bf(v2[i]), b1 = hi_bf(v2[i]);
        s1 += a0 * a0 + a1 * a1;
        s2 += b0 * b0 + b1 * b1;
      }
    }
    var1 = block_sum_256(s1, s_red) / a.hidden_local;
    var2 = block_sum_256(s2, s_red) / a.hidden_local;
    if (MODE == 0) {
      if (threadIdx.x == 0) {
        a.stats[2 * t] = var1;
        a.stats[2 * t + 1] = var2;
      }
      return;
    }
  } else {
    var1 = a.stats[2 * t];
    var2 = a.stats[2 * t + 1];
  }
  const float r1 = rsqrtf(var1 + a.eps), r2 = rsqrtf(var2 + a.eps);
  uint16_t* out = reinterpret_cast<uint16_t*>(a.out) + t * a.out_stride_t;
  for (int c = threadIdx.x; c < chunks; c += blockDim.x) {
    const u32x4 v1 = ld16(xa + 8 * c), v2 = ld16(xl + 8 * c);
    const f32x4 w1a = *reinterpret_cast<const f32x4*>(a.weight1 + 8 * c);
    const f32x4 w1b = *reinterpret_cast<const f32x4*>(a.weight1 + 8 * c + 4);
    const f32x4 w2a = *reinterpret_cast<const f32x4*>(a.weight2 + 8 * c);
    const f32x4 w2b = *reinterpret_cast<const f32x4*>(a.weight2 + 8 * c + 4);
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float y[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int j = 2 * i + e;
        const float x1 = e ? hi_bf(v1[i]) : lo_bf(v1[i]);
        const float x2 = e ? hi_bf(v2[i]) : lo_bf(v2[i]);
        const float w1 = j < 4 ? w1a[j] : w1b[j - 4];
        const float w2 = j < 4 ? w2a[j] : w2b[j - 4];
        y[e] = (bf16_round(x1 * r1) * w1 + bf16_round(x2 * r2) * w2) * 0.5f;
      }
      o[i] = pack_bf16x2(y[0], y[1]);
    }
    st16(out + 8 * c, o);
  }
}

// One workgroup per token.  With a residual: z = bf16(x + residual) is written over the residual
// and read back by the same thread for the second pass.
// Q: also the row's dynamic int8 quantisation (xq [T, hidden], x_scale [T]) of the bf16 output, for a W8A8 linear that
// consumes it: the linear then skips its own quantise launch.
template <bool ADD, bool Q>
__global__ __launch_bounds__(256) void add_rmsnorm_kernel(hymeta_add_rmsnorm_args a, int8_t* xq, float* x_scale) {
  __shared__ float s_red[4];
  const int64_t t = blockIdx.x;
  const uint16_t* x = reinterpret_cast<const uint16_t*>(a.x) + t * a.x_stride_t;
  uint16_t* res = ADD ? reinterpret_cast<uint16_t*>(a.residual) + t * a.residual_stride_t : nullptr;
  const uint16_t* w = reinterpret_cast<const uint16_t*>(a.weight);
  uint16_t* out = reinterpret_cast<uint16_t*>(a.out) + t * a.out_stride_t;
  const int chunks = a.hidden / 8;
  float ss = 0.f;
  for (int c = threadIdx.x; c < chunks; c += blockDim.x) {
    u32x4 z = ld16(x + 8 * c);
    if (ADD) {
      const u32x4 r = ld16(res + 8 * c);
#pragma unroll
      for (int i = 0; i < 4; ++i) z[i] = pack_bf16x2(lo_bf(z[i]) + lo_bf(r[i]), hi_bf(z[i]) + hi_bf(r[i]));
      st16(res + 8 * c, z);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float z0 = lo_bf(z[i]), z1 = hi_bf(z[i]);
      ss += z0 * z0 + z1 * z1;
    }
  }
  const float rs = rsqrtf(block_sum_256(ss, s_red) / a.hidden + a.eps);
  const uint16_t* zsrc = ADD ? res : x;
  for (int c = threadIdx.x; c < chunks; c += blockDim.x) {
    const u32x4 z = ld16(zsrc + 8 * c), wv = ld16(w + 8 * c);
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      o[i] = pack_bf16x2(bf16_round(lo_bf(z[i]) * rs) * lo_bf(wv[i]), bf16_round(hi_bf(z[i]) * rs) * hi_bf(wv[i]));
    st16(out + 8 * c, o);
  }
  if constexpr (Q) quantise_row_int8(out, a.hidden, xq + t * a.hidden, x_scale + t, s_red);
}

// add_rmsnorm_kernel<true> whose x is a K-sliced linear's result still in partial sums: slices added in slice order,
// W8A8 scales and the bias applied, the sum rounded to bf16 -- the bits the linear's own reduce launch would have
// written -- then z = bf16(x + residual) as above.
template <bool I8, bool Q>
__global__ __launch_bounds__(256) void add_rmsnorm_partials_kernel(hymeta_add_rmsnorm_args a, hymeta_gemm_partials p, int8_t* xq,
                                                                   float* x_scale) {
  __shared__ float s_red[4];
  const int64_t t = blockIdx.x;
  uint16_t* res = reinterpret_cast<uint16_t*>(a.residual) + t * a.residual_stride_t;
  const uint16_t* w = reinterpret_cast<const uint16_t*>(a.weight);
  uint16_t* out = reinterpret_cast<uint16_t*>(a.out) + t * a.out_stride_t;
  const int chunks = a.hidden / 8;
  if (chunks <= 512) {
    // up to 4096 columns: a thread owns chunks tid and tid + 256; every load of both (residual, norm weight, the slabs
    // of a batch of slices) is in flight before the first is used, and z stays in registers between the two passes
    const int ca = min(static_cast<int>(threadIdx.x), chunks - 1), cb = threadIdx.x + 256 < chunks ? threadIdx.x + 256 : ca;
    const bool has_a = static_cast<int>(threadIdx.x) < chunks, has_b = static_cast<int>(threadIdx.x) + 256 < chunks;
    const u32x4 ra = ld16(res + 8 * ca), rb = ld16(res + 8 * cb);
    const u32x4 wa = ld16(w + 8 * ca), wb = ld16(w + 8 * cb);
    float xa[8], xb[8];
    partial_row8_x2<I8>(p, t, 8 * ca, 8 * cb, xa, xb);
    u32x4 za, zb;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      za[i] = pack_bf16x2(xa[2 * i] + lo_bf(ra[i]), xa[2 * i + 1] + hi_bf(ra[i]));
      zb[i] = pack_bf16x2(xb[2 * i] + lo_bf(rb[i]), xb[2 * i + 1] + hi_bf(rb[i]));
    }
    if (has_a) {
      st16(res + 8 * ca, za);
#pragma unroll
      for (int i = 0; i < 4; ++i) ss += lo_bf(za[i]) * lo_bf(za[i]) + hi_bf(za[i]) * hi_bf(za[i]);
    }
    if (has_b) {
      st16(res + 8 * cb, zb);
#pragma unroll
      for (int i = 0; i < 4; ++i) ss += lo_bf(zb[i]) * lo_bf(zb[i]) + hi_bf(zb[i]) * hi_bf(zb[i]);
    }
    const float rs = rsqrtf(block_sum_256(ss, s_red) / a.hidden + a.eps);
    u32x4 oa, ob;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      oa[i] = pack_bf16x2(bf16_round(lo_bf(za[i]) * rs) * lo_bf(wa[i]), bf16_round(hi_bf(za[i]) * rs) * hi_bf(wa[i]));
      ob[i] = pack_bf16x2(bf16_round(lo_bf(zb[i]) * rs) * lo_bf(wb[i]), bf16_round(hi_bf(zb[i]) * rs) * hi_bf(wb[i]));
    }
    if (has_a) st16(out + 8 * ca, oa);
    if (has_b) st16(out + 8 * cb, ob);
    if constexpr (Q) {       // the row is in registers: quantise it from there
      const u32x4 rowv[2] = {oa, ob};
      quantise_chunks_int8<2>(rowv, chunks, xq + t * a.hidden, x_scale + t, s_red);
      return;
    }
  } else {
    float ss = 0.f;
    for (int c = threadIdx.x; c < chunks; c += blockDim.x) {
      float x[8];
      partial_row8<I8>(p, t, 8 * c, x);
      const u32x4 r = ld16(res + 8 * c);
      u32x4 z;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        z[i] = pack_bf16x2(x[2 * i] + lo_bf(r[i]), x[2 * i + 1] + hi_bf(r[i]));
      st16(res + 8 * c, z);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float z0 = lo_bf(z[i]), z1 = hi_bf(z[i]);
        ss += z0 * z0 + z1 * z1;
      }
    }
    const float rs = rsqrtf(block_sum_256(ss, s_red) / a.hidden + a.eps);
    for (int c = threadIdx.x; c < chunks; c += blockDim.x) {
      const u32x4 z = ld16(res + 8 * c), wv = ld16(w + 8 * c);
      u32x4 o;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        o[i] = pack_bf16x2(bf16_round(lo_bf(z[i]) * rs) * lo_bf(wv[i]), bf16_round(hi_bf(z[i]) * rs) * hi_bf(wv[i]));
      st16(out + 8 * c, o);
    }
  }
  if constexpr (Q) quantise_row_int8(out, a.hidden, xq + t * a.hidden, x_scale + t, s_red);
}

}  // namespace hymeta

using namespace hymeta;

static int add_rmsnorm_partials_impl(const hymeta_add_rmsnorm_args* args, const hymeta_gemm_partials* partials, int8_t* xq,
                                     float* x_scale, hymeta_stream_t stream) {
  if (!hymeta::args_ok(args) || !hymeta::args_ok(partials)) return HYMETA_ERR_INVALID_ARG;
  const hymeta_add_rmsnorm_args& a = *args;
  const hymeta_gemm_partials& p = *partials;
  if (a.num_tokens < 0 || a.hidden <= 0 || p.splits < 1 || !p.slabs) return HYMETA_ERR_INVALID_ARG;
  if (a.num_tokens != p.M || a.hidden != p.N) return HYMETA_ERR_INVALID_ARG;
  if (a.num_tokens == 0) return HYMETA_OK;
  if (!a.residual || !a.weight || !a.out) return HYMETA_ERR_INVALID_ARG;
  if (p.is_int32 && (!p.x_scale || !p.w_scale)) return HYMETA_ERR_INVALID_ARG;
  if (a.hidden % 8 != 0 || (a.out_stride_t | a.residual_stride_t) % 8 != 0) return HYMETA_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(p.slabs) | reinterpret_cast<uintptr_t>(a.residual) | reinterpret_cast<uintptr_t>(a.weight) |
       reinterpret_cast<uintptr_t>(a.out) | reinterpret_cast<uintptr_t>(p.bias) | reinterpret_cast<uintptr_t>(xq)) % 16 != 0)
    return HYMETA_ERR_UNSUPPORTED;
  hipStream_t s = begin_launches(stream);
  const dim3 grid(a.num_tokens), block(256);
  if (xq != nullptr) {
    if (p.is_int32) hipLaunchKernelGGL((add_rmsnorm_partials_kernel<true, true>), grid, block, 0, s, a, p, xq, x_scale);
    else hipLaunchKernelGGL((add_rmsnorm_partials_kernel<false, true>), grid, block, 0, s, a, p, xq, x_scale);
  } else {
    if (p.is_int32) hipLaunchKernelGGL((add_rmsnorm_partials_kernel<true, false>), grid, block, 0, s, a, p, xq, x_scale);
    else hipLaunchKernelGGL((add_rmsnorm_partials_kernel<false, false>), grid, block, 0, s, a, p, xq, x_scale);
  }
  return launch_status();
}

extern "C" int hymeta_add_rmsnorm_partials(const hymeta_add_rmsnorm_args* args, const hymeta_gemm_partials* partials,
                                           hymeta_stream_t stream) {
  return add_rmsnorm_partials_impl(args, partials, nullptr, nullptr, stream);
}

extern "C" int hymeta_rope_neox(const hymeta_rope_args* args, hymeta_stream_t stream) {
  if (!hymeta::args_ok(args)) return HYMETA_ERR_INVALID_ARG;
  const hymeta_rope_args& a = *args;
  if (a.num_tokens < 0 || a.num_heads <= 0 || a.num_kv_heads < 0) return HYMETA_ERR_INVALID_ARG;
  if (a.num_tokens == 0) return HYMETA_OK;
  if (!a.q || !a.positions || !a.cos_sin_cache) return HYMETA_ERR_INVALID_ARG;
  if (a.rot_dim != a.head_dim || a.rot_dim % 16 != 0) return HYMETA_ERR_UNSUPPORTED;
  if ((a.q_stride_t | a.q_stride_h | a.k_stride_t | a.k_stride_h) % 8 != 0) return HYMETA_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(a.q) | reinterpret_cast<uintptr_t>(a.k) |
       reinterpret_cast<uintptr_t>(a.cos_sin_cache)) % 16 != 0) return HYMETA_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(rope_neox_kernel, dim3(a.num_tokens), dim3(256), 0, begin_launches(stream), a);
  return launch_status();
}

extern "C" int hymeta_rope_neox_partials(const hymeta_rope_args* args, const hymeta_gemm_partials* partials,
                                         hymeta_stream_t stream) {
  if (!hymeta::args_ok(args) || !hymeta::args_ok(partials)) return HYMETA_ERR_INVALID_ARG;
  const hymeta_rope_args& a = *args;
  const hymeta_gemm_partials& p = *partials;
  if (a.num_tokens < 0 || a.num_heads <= 0 || a.num_kv_heads < 0 || p.splits < 1 || !p.slabs) return HYMETA_ERR_INVALID_ARG;
  if (a.num_tokens != p.M) return HYMETA_ERR_INVALID_ARG;
  if (a.num_tokens == 0) return HYMETA_OK;
  if (!a.q || !a.positions || !a.cos_sin_cache) return HYMETA_ERR_INVALID_ARG;
  if (p.is_int32 && (!p.x_scale || !p.w_scale)) return HYMETA_ERR_INVALID_ARG;
  if (a.rot_dim != a.head_dim || a.rot_dim % 16 != 0 || p.N % 8 != 0) return HYMETA_ERR_UNSUPPORTED;
  // the output is the projection's own [T, N] buffer: q heads, then k heads, head_dim apart, the row N or more wide
  if (a.q_stride_h != a.head_dim || (a.num_heads + a.num_kv_heads) * a.head_dim > p.N || a.q_stride_t < p.N || a.q_stride_t % 8 != 0)
    return HYMETA_ERR_INVALID_ARG;
  if (a.k != nullptr && (a.k != static_cast<char*>(a.q) + static_cast<int64_t>(a.num_heads) * a.head_dim * 2 ||
                         a.k_stride_t != a.q_stride_t || a.k_stride_h != a.head_dim))
    return HYMETA_ERR_INVALID_ARG;
  if ((reinterpret_cast<uintptr_t>(a.q) | reinterpret_cast<uintptr_t>(a.cos_sin_cache) | reinterpret_cast<uintptr_t>(p.slabs) |
       reinterpret_cast<uintptr_t>(p.bias)) % 16 != 0) return HYMETA_ERR_UNSUPPORTED;
  hipStream_t s = begin_launches(stream);
  if (p.is_int32) hipLaunchKernelGGL(rope_partials_kernel<true>, dim3(a.num_tokens), dim3(256), 0, s, a, p);
  else hipLaunchKernelGGL(rope_partials_kernel<false>, dim3(a.num_tokens), dim3(256), 0, s, a, p);
  return launch_status();
}

static int check_combine(const hymeta_combine_args& a, bool need_out) {
  if (a.num_tokens < 0 || a.hidden_local <= 0) return HYMETA_ERR_INVALID_ARG;
  if (!a.o_attn || !a.o_lin) return HYMETA_ERR_INVALID_ARG;
  if (need_out && (!a.out || !a.weight1 || !a.weight2)) return HYMETA_ERR_INVALID_ARG;
  if (!need_out && !a.stats) return HYMETA_ERR_INVALID_ARG;
  if (a.hidden_local % 8 != 0 || (a.attn_stride_t | a.lin_stride_t | a.out_stride_t) % 8 != 0) return HYMETA_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(a.o_attn) | reinterpret_cast<uintptr_t>(a.o_lin) | reinterpret_cast<uintptr_t>(a.out) |
       reinterpret_cast<uintptr_t>(a.weight1) | reinterpret_cast<uintptr_t>(a.weight2)) % 16 != 0) return HYMETA_ERR_UNSUPPORTED;
  return HYMETA_OK;
}

extern "C" int hymeta_rms_sumsq2(const hymeta_combine_args* args, hymeta_stream_t stream) {
  if (!hymeta::args_ok(args)) return HYMETA_ERR_INVALID_ARG;
  const int rc = check_combine(*args, false);
  if (rc != HYMETA_OK || args->num_tokens == 0) return rc;
  hipLaunchKernelGGL((hybrid_combine_kernel<0>), dim3(args->num_tokens), dim3(256), 0, begin_launches(stream), *args);
  return launch_status();
}

extern "C" int hymeta_hybrid_combine(const hymeta_combine_args* args, hymeta_stream_t stream) {
  if (!hymeta::args_ok(args)) return HYMETA_ERR_INVALID_ARG;
  const int rc = check_combine(*args, true);
  if (rc != HYMETA_OK || args->num_tokens == 0) return rc;
  hipLaunchKernelGGL((hybrid_combine_kernel<1>), dim3(args->num_tokens), dim3(256), 0, begin_launches(stream), *args);
  return launch_status();
}

namespace hymeta {

// One token per workgroup: 16 dot products over the hidden size (w0 is 16 rows, read by every
// workgroup and served by the L2), summed over the workgroup, rounded to bf16 as the first linear's
// output is, then the 16 -> out_features expansion with bias.
__global__ __launch_bounds__(256) void lowrank_gate_kernel(hymeta_lowrank_gate_args a) {
  constexpr int R = 16;
  __shared__ float s_part[4][R];
  __shared__ float s_t[R];
  const int tid = threadIdx.x;
  const uint16_t* x = reinterpret_cast<const uint16_t*>(a.x) + static_cast<int64_t>(blockIdx.x) * a.x_stride_t;
  const uint16_t* w0 = reinterpret_cast<const uint16_t*>(a.w0);
  // the second stage's weights do not depend on the first: fetch this thread's first two rows now, under the first
  // stage's loads, instead of after the workgroup's reduction
  const uint16_t* w1 = reinterpret_cast<const uint16_t*>(a.w1);
  u32x4 w1a[2], w1b[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int o = min(tid + 256 * j, a.out_features - 1);
    w1a[j] = ld16(w1 + static_cast<int64_t>(o) * R), w1b[j] = ld16(w1 + static_cast<int64_t>(o) * R + 8);
  }
  lowrank_first_linear(x, w0, a.hidden, s_part, s_t);
  float t[R];
#pragma unroll
  for (int r = 0; r < R; ++r) t[r] = s_t[r];
  uint16_t* out = reinterpret_cast<uint16_t*>(a.out) + static_cast<int64_t>(blockIdx.x) * a.out_stride_t;
  for (int o = tid, j = 0; o < a.out_features; o += 256, ++j) {
    const u32x4 wa = j < 2 ? w1a[j & 1] : ld16(w1 + static_cast<int64_t>(o) * R);
    const u32x4 wb = j < 2 ? w1b[j & 1] : ld16(w1 + static_cast<int64_t>(o) * R + 8);
    float y = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      y = fmaf(t[2 * i], lo_bf(wa[i]), y);
      y = fmaf(t[2 * i + 1], hi_bf(wa[i]), y);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      y = fmaf(t[8 + 2 * i], lo_bf(wb[i]), y);
      y = fmaf(t[8 + 2 * i + 1], hi_bf(wb[i]), y);
    }
    if (a.bias != nullptr) y += bf2f(reinterpret_cast<const uint16_t*>(a.bias)[o]);
    out[o] = f2bf(y);
  }
}

}  // namespace hymeta

extern "C" int hymeta_lowrank_gate(const hymeta_lowrank_gate_args* args, hymeta_stream_t stream) {
  if (!hymeta::args_ok(args)) return HYMETA_ERR_INVALID_ARG;
  const hymeta_lowrank_gate_args& a = *args;
  if (a.num_tokens < 0 || a.hidden <= 0 || a.out_features <= 0) return HYMETA_ERR_INVALID_ARG;
  if (a.num_tokens == 0) return HYMETA_OK;
  if (!a.x || !a.w0 || !a.w1 || !a.out) return HYMETA_ERR_INVALID_ARG;
  if (a.rank != 16 || a.hidden % 8 != 0 || a.x_stride_t % 8 != 0) return HYMETA_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(a.x) | reinterpret_cast<uintptr_t>(a.w0) | reinterpret_cast<uintptr_t>(a.w1)) % 16 != 0)
    return HYMETA_ERR_UNSUPPORTED;
  hipStream_t s = hymeta::begin_launches(stream);
  hipLaunchKernelGGL(hymeta::lowrank_gate_kernel, dim3(a.num_tokens), dim3(256), 0, s, a);
  return hymeta::launch_status();
}

static int add_rmsnorm_impl(const hymeta_add_rmsnorm_args* args, int8_t* xq, float* x_scale, hymeta_stream_t stream) {
  if (!hymeta::args_ok(args)) return HYMETA_ERR_INVALID_ARG;
  const hymeta_add_rmsnorm_args& a = *args;
  if (a.num_tokens < 0 || a.hidden <= 0) return HYMETA_ERR_INVALID_ARG;
  if (a.num_tokens == 0) return HYMETA_OK;
  if (!a.x || !a.weight || !a.out) return HYMETA_ERR_INVALID_ARG;
  if (a.hidden % 8 != 0 || (a.x_stride_t | a.out_stride_t | (a.residual ? a.residual_stride_t : 0)) % 8 != 0) return HYMETA_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(a.x) | reinterpret_cast<uintptr_t>(a.residual) | reinterpret_cast<uintptr_t>(a.weight) |
       reinterpret_cast<uintptr_t>(a.out) | reinterpret_cast<uintptr_t>(xq)) % 16 != 0) return HYMETA_ERR_UNSUPPORTED;
  hipStream_t s = begin_launches(stream);
  const dim3 grid(a.num_tokens), block(256);
  if (xq != nullptr) {
    if (a.residual != nullptr) hipLaunchKernelGGL((add_rmsnorm_kernel<true, true>), grid, block, 0, s, a, xq, x_scale);
    else hipLaunchKernelGGL((add_rmsnorm_kernel<false, true>), grid, block, 0, s, a, xq, x_scale);
  } else {
    if (a.residual != nullptr) hipLaunchKernelGGL((add_rmsnorm_kernel<true, false>), grid, block, 0, s, a, xq, x_scale);
    else hipLaunchKernelGGL((add_rmsnorm_kernel<false, false>), grid, block, 0, s, a, xq, x_scale);
  }
  return launch_status();
}

extern "C" int hymeta_add_rmsnorm(const hymeta_add_rmsnorm_args* args, hymeta_stream_t stream) {
  return add_rmsnorm_impl(args, nullptr, nullptr, stream);
}

extern "C" int hymeta_add_rmsnorm_int8(const hymeta_add_rmsnorm_args* args, const hymeta_gemm_partials* partials, int8_t* xq,
                                       float* x_scale, hymeta_stream_t stream) {
  if (xq == nullptr || x_scale == nullptr) return HYMETA_ERR_INVALID_ARG;
  if (partials != nullptr && !hymeta::args_ok(partials)) return HYMETA_ERR_INVALID_ARG;
  if (partials != nullptr && partials->splits > 0) return add_rmsnorm_partials_impl(args, partials, xq, x_scale, stream);
  return add_rmsnorm_impl(args, xq, x_scale, stream);
}

namespace hymeta {

// Greedy sampling: index of the largest logit of each row (the lowest index among equals).  A workgroup scans one of
// ``segs`` column segments of a row, 4 columns per thread at a time, and merges its 256 candidates through LDS; with
// more than one segment the candidates (value, index) go to ``part`` [M][segs] and a second launch picks among them.
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* logits, int64_t stride, int n, int segs, float* part_val,
                                                          int* part_idx, int64_t* out) {
  __shared__ float s_val[256];
  __shared__ int s_idx[256];
  const int m = blockIdx.x / segs, seg = blockIdx.x % segs;
  const float* row = logits + static_cast<int64_t>(m) * stride;
  const int per = ((n / 4 + segs - 1) / segs) * 4;            // columns per segment, a multiple of 4
  const int lo = seg * per, hi = min(n, lo + per);
  float best = -INFINITY;
  int best_i = 0x7fffffff;
  for (int c = lo + 4 * threadIdx.x; c < hi; c += 1024) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(row + c);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (v[e] > best) best = v[e], best_i = c + e;       // ascending columns: the first of equals stays
  }
  s_val[threadIdx.x] = best, s_idx[threadIdx.x] = best_i;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (static_cast<int>(threadIdx.x) < off) {
      const float v = s_val[threadIdx.x + off];
      const int i = s_idx[threadIdx.x + off];
      if (v > s_val[threadIdx.x] || (v == s_val[threadIdx.x] && i < s_idx[threadIdx.x])) s_val[threadIdx.x] = v, s_idx[threadIdx.x] = i;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (segs == 1) out[m] = s_idx[0] == 0x7fffffff ? 0 : s_idx[0];
    else part_val[blockIdx.x] = s_val[0], part_idx[blockIdx.x] = s_idx[0];
  }
}

// One wave per row picks among the segments' candidates (segs <= 64).
__global__ __launch_bounds__(64) void argmax_pick_kernel(const float* part_val, const int* part_idx, int segs, int64_t* out) {
  const int m = blockIdx.x, lane = threadIdx.x;
  float v = lane < segs ? part_val[m * segs + lane] : -INFINITY;
  int i = lane < segs ? part_idx[m * segs + lane] : 0x7fffffff;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const float ov = __shfl_xor(v, off);
    const int oi = __shfl_xor(i, off);
    if (ov > v || (ov == v && oi < i)) v = ov, i = oi;
  }
  if (lane == 0) out[m] = i == 0x7fffffff ? 0 : i;
}

}  // namespace hymeta

extern "C" size_t hymeta_argmax_rows_workspace_bytes(int32_t M) { return static_cast<size_t>(M > 0 ? M : 0) * 32 * 8; }

extern "C" int hymeta_argmax_rows(const float* logits, int64_t stride_m, int32_t M, int32_t N, int64_t* out, void* workspace,
                                  size_t workspace_bytes, hymeta_stream_t stream) {
  if (M < 0 || N <= 0) return HYMETA_ERR_INVALID_ARG;
  if (M == 0) return HYMETA_OK;
  if (!logits || !out) return HYMETA_ERR_INVALID_ARG;
  if (N % 4 != 0 || stride_m % 4 != 0 || reinterpret_cast<uintptr_t>(logits) % 16 != 0) return HYMETA_ERR_UNSUPPORTED;
  // few rows of a wide vocabulary: 32 workgroups per row, so that the whole chip reads
  const int segs = workspace != nullptr && workspace_bytes >= hymeta_argmax_rows_workspace_bytes(M) && N >= 32 * 1024 && M <= 1024 &&
                           reinterpret_cast<uintptr_t>(workspace) % 4 == 0 ? 32 : 1;
  float* part_val = static_cast<float*>(workspace);
  int* part_idx = reinterpret_cast<int*>(part_val + static_cast<size_t>(M) * 32);
  hipStream_t s = hymeta::begin_launches(stream);
  hipLaunchKernelGGL(hymeta::argmax_rows_kernel, dim3(M * segs), dim3(256), 0, s, logits, stride_m, N, segs, part_val, part_idx, out);
  if (segs > 1) {
    if (hipGetLastError() != hipSuccess) return HYMETA_ERR_LAUNCH;
    hipLaunchKernelGGL(hymeta::argmax_pick_kernel, dim3(M), dim3(64), 0, s, part_val, part_idx, segs, out);
  }
  return hymeta::launch_status();
}

namespace hymeta {

// Multi-step decoding: every live sequence moves on by one token, in place (one thread per sequence).
__global__ __launch_bounds__(256) void advance_step_kernel(int num_seqs, int num_queries, int block_size, int64_t* input_tokens,
                                                           const int64_t* sampled_token_ids, int64_t* input_positions, int32_t* seq_lens,
                                                           int64_t* slot_mapping, const int32_t* block_tables, int64_t block_table_stride) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= num_seqs) return;
  if (i >= num_queries) {        // graph padding
    slot_mapping[i] = -1;
    return;
  }
  if (input_tokens != nullptr && sampled_token_ids != nullptr) input_tokens[i] = sampled_token_ids[i];
  const int pos = seq_lens[i];   // position of the next token = the length so far
  seq_lens[i] = pos + 1;
  if (input_positions != nullptr) input_positions[i] = pos;
  const int64_t block = block_tables[i * block_table_stride + pos / block_size];
  slot_mapping[i] = block * block_size + pos % block_size;
}

}  // namespace hymeta

extern "C" int hymeta_advance_step(int32_t num_seqs, int32_t num_queries, int32_t block_size, int64_t* input_tokens,
                                   const int64_t* sampled_token_ids, int64_t* input_positions, int32_t* seq_lens, int64_t* slot_mapping,
                                   const int32_t* block_tables, int64_t block_table_stride, hymeta_stream_t stream) {
  if (num_seqs < 0 || num_queries < 0 || num_queries > num_seqs || block_size <= 0) return HYMETA_ERR_INVALID_ARG;
  if (num_seqs == 0) return HYMETA_OK;
  if (!seq_lens || !slot_mapping || !block_tables) return HYMETA_ERR_INVALID_ARG;
  hipLaunchKernelGGL(hymeta::advance_step_kernel, dim3((num_seqs + 255) / 256), dim3(256), 0, hymeta::begin_launches(stream), num_seqs,
                     num_queries, block_size, input_tokens, sampled_token_ids, input_positions, seq_lens, slot_mapping, block_tables,
                     block_table_stride);
  return hymeta::launch_status();
}
