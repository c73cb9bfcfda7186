// Sparse mixture-of-experts block of the 70B model: top-k routing (softmax probability as the
// weight, optionally renormalised), counting sort of the (token, choice) pairs by expert, grouped
// expert GEMMs straight from the per-expert weight stacks, SiLU-and-multiply, and the combine
// with the shared expert.  Everything stays on the device: no host read of the expert counts.
//
//   route    one lane-group per token: softmax over E <= 64 logits, k <= 2 picks, histogram
//   offsets  exclusive prefix over the experts (one workgroup)
//   scatter  pair -> slot offsets[e] + ticket; the order inside an expert is arbitrary, which
//            cannot change a result: rows of a GEMM are independent
//   gemm     out[row] = x[a_rows[row]] W[e]^T for the rows of expert e; 64 columns per workgroup,
//            the 4 waves take alternate 32-deep k-steps (bf16 MFMA 16x16x32, operands straight
//            from global memory: both x rows and W rows are K-contiguous), LDS reduce
//   combine  out[t] = sum_k y[pos[t,k]] (+ shared[t]), y weighted in the down GEMM; pairs of non-local experts
//            (expert parallelism) carry pos = -1 and add nothing
#include "common.h"
#include "w8a8_quant.h"

#include <algorithm>
#include <cstdlib>

namespace hymeta {
namespace {

constexpr int kMaxExperts = 64;
constexpr int kCols = 64;

__device__ __forceinline__ f32x4 mfma16(const bf16x8& a, const bf16x8& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ bf16x8 as_frag(const u32x4& v) { return __builtin_bit_cast(bf16x8, v); }

// The expert GEMMs in two operand types: bf16 (fp32 sums), or int8 weights with per-output-channel scales against
// per-row int8 activations (W8A8: exact int32 sums, the scales applied once on the way out).  A 16-byte fragment piece
// is 8 bf16 or 16 int8 k-values of one row, a 64-byte k-step one v_mfma_f32_16x16x32_bf16 or one v_mfma_i32_16x16x64_i8:
// in BYTES the two walk the operands identically, which is how the kernels below are written.
template <bool I8> struct MoeAcc { typedef f32x4 type; };
template <> struct MoeAcc<true> { typedef i32x4 type; };

template <bool I8>
__device__ __forceinline__ typename MoeAcc<I8>::type moe_mma(const u32x4& a, const u32x4& b, const typename MoeAcc<I8>::type& c) {
  if constexpr (I8) {
    return __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, a), __builtin_bit_cast(i32x4, b), c, 0, 0, 0);
  } else {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(a), as_frag(b), c, 0, 0, 0);
  }
}

// source row of a slot (the token of a pair for the first GEMM, the slot itself for the second)
__device__ __forceinline__ int moe_src_row(const hymeta_moe_gemm_args& a, int slot) {
  return a.a_rows != nullptr ? a.a_rows[slot] / a.a_rows_div : slot;
}

// fp32 value of four summed columns of one row: the sums themselves, or (W8A8) the int32 sums times the row's and the
// columns' scales with the rounding rule every W8A8 kernel here shares
template <bool I8>
__device__ __forceinline__ f32x4 moe_sums_to_f32(const hymeta_moe_gemm_args& a, const typename MoeAcc<I8>::type& s, int e, int slot,
                                                 int col) {
  if constexpr (I8) {
    const float xs = a.x_scale[moe_src_row(a, slot)];
    const f32x4 ws = *reinterpret_cast<const f32x4*>(a.w_scale + static_cast<int64_t>(e) * a.N + col);
    return f32x4{w8a8_scaled(s[0], xs, ws[0], 0.f), w8a8_scaled(s[1], xs, ws[1], 0.f), w8a8_scaled(s[2], xs, ws[2], 0.f),
                 w8a8_scaled(s[3], xs, ws[3], 0.f)};
  } else {
    return s;
  }
}

}  // namespace

// One thread per token (E is small).  counts[e] += 1 per pair.
__global__ __launch_bounds__(256) void moe_route_kernel(hymeta_moe_route_args a) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= a.num_tokens) return;
  float logit[kMaxExperts];
  float mx = -INFINITY;
  for (int e = 0; e < a.num_experts; ++e) {
    const int64_t i = static_cast<int64_t>(t) * a.logits_stride_t + e;
    logit[e] = a.logits_dtype == HYMETA_F32 ? reinterpret_cast<const float*>(a.router_logits)[i]
                                            : bf2f(reinterpret_cast<const uint16_t*>(a.router_logits)[i]);
    mx = fmaxf(mx, logit[e]);
  }
  float denom = 0.f;
  for (int e = 0; e < a.num_experts; ++e) {
    logit[e] = expf(logit[e] - mx);
    denom += logit[e];
  }
  int pick[2] = {0, 0};
  float pw[2] = {0.f, 0.f};
  for (int j = 0; j < a.top_k; ++j) {
    // starts from the first expert not taken: a row of NaN logits compares false everywhere and keeps it, with a NaN
    // weight -- a valid index for the counters below and a NaN row out, never index -1
    int best = (j == 1 && pick[0] == 0) ? 1 : 0;
    float bv = logit[best];
    for (int e = 0; e < a.num_experts; ++e) {
      const bool taken = j == 1 && e == pick[0];
      if (!taken && logit[e] > bv) {   // first maximum wins, as torch.topk on ties of distinct indices
        bv = logit[e];
        best = e;
      }
    }
    pick[j] = best;
    pw[j] = bv / denom;
  }
  if (a.renormalize) {
    float s = 0.f;
    for (int j = 0; j < a.top_k; ++j) s += pw[j];
    for (int j = 0; j < a.top_k; ++j) pw[j] /= s;
  }
  for (int j = 0; j < a.top_k; ++j) {
    a.topk_ids[t * a.top_k + j] = pick[j];
    a.topk_weights[t * a.top_k + j] = pw[j];
    atomicAdd(a.counts + pick[j], 1);
  }
}

__global__ __launch_bounds__(64) void moe_zero_counts_kernel(hymeta_moe_route_args a) {
  if (static_cast<int>(threadIdx.x) < a.num_experts) a.counts[threadIdx.x] = 0;
}

// offsets[0..E] = exclusive prefix of counts; cursors zeroed.
__global__ __launch_bounds__(64) void moe_offsets_kernel(hymeta_moe_route_args a) {
  if (threadIdx.x == 0) {
    int run = 0;
    for (int e = 0; e < a.num_experts; ++e) {
      a.expert_offsets[e] = run;
      run += a.counts[e];
      a.cursors[e] = 0;
    }
    a.expert_offsets[a.num_experts] = run;
  }
}

__global__ __launch_bounds__(256) void moe_scatter_kernel(hymeta_moe_route_args a) {
  const int p = blockIdx.x * 256 + threadIdx.x;   // pair index = token * top_k + choice
  if (p >= a.num_tokens * a.top_k) return;
  const int e = a.topk_ids[p];
  const bool local = e >= a.local_expert_begin && e < a.local_expert_end;
  int pos = -1;
  if (local) {
    pos = a.expert_offsets[e] + atomicAdd(a.cursors + e, 1);
    a.sorted_pair[pos] = p;
  }
  a.pair_pos[p] = pos;
}

// Decode-sized batches: the router's linear layer, the routing, the expert offsets and the scatter in ONE launch.
// grid = tokens.  A workgroup computes its token's E logits (x row times the [E, H] gate weight, fp32 sums of bf16
// products, rounded to bf16 as the bf16 linear layer's output is), one thread routes them as moe_route_kernel does, and
// the workgroup that draws the last ticket counts the pairs per expert in LDS, writes the offsets and scatters the
// pairs.  The ticket word must be zero on entry and is zero again on exit.
// I8: the router of a W8A8 checkpoint (int8 [E, H] with a scale per expert) against the token's int8 row: exact integer
// dot products (v_dot4_i32_i8), scaled and rounded as the W8A8 linear kernels scale and round -- the logits that linear
// would give, bit for bit.
template <bool I8>
__global__ __launch_bounds__(256) void moe_router_kernel(hymeta_moe_router_args a) {
  __shared__ float s_part[4][kMaxExperts];
  __shared__ float s_logit[kMaxExperts];
  __shared__ int s_count[kMaxExperts], s_cur[kMaxExperts], s_off[kMaxExperts + 1];
  __shared__ int s_last;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int t = blockIdx.x;
  const int E = a.num_experts;
  constexpr int ES = I8 ? 1 : 2;                       // bytes per element of x and the gate weight
  const int chunks = a.hidden * ES / 16;               // 16-byte pieces of a row
  const char* xrow = static_cast<const char*>(a.x) + static_cast<int64_t>(t) * a.x_stride_t * ES;
  const char* w = static_cast<const char*>(a.gate_weight);
  const int64_t wpitch = static_cast<int64_t>(a.hidden) * ES;

  for (int e0 = 0; e0 < E; e0 += 16) {
    float acc[16];
    int iacc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f, iacc[j] = 0;
    for (int ch = tid; ch < chunks; ch += 256) {
      const u32x4 xv = ld16g(xrow + 16 * ch);
      u32x4 wv[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) wv[j] = ld16g(w + static_cast<int64_t>(min(e0 + j, E - 1)) * wpitch + 16 * ch);
#pragma unroll
      for (int j = 0; j < 16; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if constexpr (I8) {
            iacc[j] = __builtin_amdgcn_sdot4(static_cast<int>(xv[i]), static_cast<int>(wv[j][i]), iacc[j], false);
          } else {
            acc[j] = __builtin_fmaf(lo_bf(xv[i]), lo_bf(wv[j][i]), acc[j]);
            acc[j] = __builtin_fmaf(hi_bf(xv[i]), hi_bf(wv[j][i]), acc[j]);
          }
        }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if constexpr (I8) {
        int v = iacc[j];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0 && e0 + j < E) s_part[wave][e0 + j] = __int_as_float(v);
      } else {
        float v = acc[j];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0 && e0 + j < E) s_part[wave][e0 + j] = v;
      }
    }
  }
  __syncthreads();
  if (tid < E) {
    float v;
    if constexpr (I8) {
      const int sum = __float_as_int(s_part[0][tid]) + __float_as_int(s_part[1][tid]) + __float_as_int(s_part[2][tid]) +
                      __float_as_int(s_part[3][tid]);
      v = bf16_round(w8a8_scaled(sum, a.x_scale[t], a.gate_scale[tid], 0.f));
    } else {
      v = bf16_round(((s_part[0][tid] + s_part[1][tid]) + s_part[2][tid]) + s_part[3][tid]);
    }
    s_logit[tid] = v;
    if (a.logits_out != nullptr) reinterpret_cast<uint16_t*>(a.logits_out)[static_cast<int64_t>(t) * E + tid] = f2bf(v);
  }
  __syncthreads();
  if (tid == 0) {
    float mx = -INFINITY;
    for (int e = 0; e < E; ++e) mx = fmaxf(mx, s_logit[e]);
    float denom = 0.f;
    for (int e = 0; e < E; ++e) {
      s_logit[e] = expf(s_logit[e] - mx);
      denom += s_logit[e];
    }
    int pick[2] = {0, 0};
    float pw[2] = {0.f, 0.f};
    for (int j = 0; j < a.top_k; ++j) {
      // starts from the first expert not taken: a row of NaN logits compares false everywhere and keeps it, with a
      // NaN weight -- a valid index for the counting below and a NaN row out, never index -1
      int best = (j == 1 && pick[0] == 0) ? 1 : 0;
      float bv = s_logit[best];
      for (int e = 0; e < E; ++e) {
        const bool taken = j == 1 && e == pick[0];
        if (!taken && s_logit[e] > bv) {   // first maximum wins
          bv = s_logit[e];
          best = e;
        }
      }
      pick[j] = best;
      pw[j] = bv / denom;
    }
    if (a.renormalize) {
      float sum = 0.f;
      for (int j = 0; j < a.top_k; ++j) sum += pw[j];
      for (int j = 0; j < a.top_k; ++j) pw[j] /= sum;
    }
    for (int j = 0; j < a.top_k; ++j) {
      __hip_atomic_store(a.topk_ids + t * a.top_k + j, pick[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      a.topk_weights[t * a.top_k + j] = pw[j];
    }
    __threadfence();
    s_last = atomicAdd(a.ticket, 1u) == gridDim.x - 1 ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;

  // the last workgroup: every token's picks are in memory
  __threadfence();
  if (tid < kMaxExperts) s_count[tid] = 0, s_cur[tid] = 0;
  __syncthreads();
  const int pairs = a.num_tokens * a.top_k;
  for (int p = tid; p < pairs; p += 256)
    atomicAdd(&s_count[__hip_atomic_load(a.topk_ids + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)], 1);
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int e = 0; e < E; ++e) {
      s_off[e] = run;
      a.expert_offsets[e] = run;
      run += s_count[e];
    }
    s_off[E] = run;
    a.expert_offsets[E] = run;
    __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  for (int p = tid; p < pairs; p += 256) {
    const int e = __hip_atomic_load(a.topk_ids + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int pos = -1;
    if (e >= a.local_expert_begin && e < a.local_expert_end) {
      pos = s_off[e] + atomicAdd(&s_cur[e], 1);
      a.sorted_pair[pos] = p;
    }
    a.pair_pos[p] = pos;
  }
}

// grid = (N / 64, local experts); rows of expert e = sorted slots [offsets[e], offsets[e + 1]).
// Row tiles of 16 MT rows are walked inside the workgroup (weights of the 64 columns are re-read
// per tile, from L2 after the first).
template <int MT, bool I8>
__global__ __launch_bounds__(256) void moe_gemm_kernel(hymeta_moe_gemm_args a) {
  typedef typename MoeAcc<I8>::type acc_t;
  constexpr int ES = I8 ? 1 : 2;                      // bytes per operand element
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* s_red = reinterpret_cast<float*>(smem_raw);  // [4 waves][16 MT rows][64 cols] of 4-byte sums

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int c = lane >> 4, n = lane & 15;
  const int nb = blockIdx.x;
  const int e = a.local_expert_begin + blockIdx.y;
  const int row_lo = a.expert_offsets[e], row_hi = a.expert_offsets[e + 1];
  if (row_lo >= row_hi) return;

  const int64_t kb = static_cast<int64_t>(a.K) * ES;   // bytes of a weight row
  const char* wbase = static_cast<const char*>(a.weights) + static_cast<int64_t>(e) * a.w_stride_e * ES;
  const char* wrow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) wrow[i] = wbase + static_cast<int64_t>(nb * kCols + 16 * i + n) * kb + 16 * c;
  const int ksteps = static_cast<int>(kb / 64);
  const int nsteps = wave < ksteps ? (ksteps - wave + 3) / 4 : 0;   // this wave's k-steps: wave, wave + 4, ...
  const int last = nsteps - 1;

  for (int r0 = row_lo; r0 < row_hi; r0 += 16 * MT) {
    const char* xrow[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int slot = min(r0 + 16 * mt + n, row_hi - 1);   // rows past the expert's end re-read its last row
      xrow[mt] = static_cast<const char*>(a.x) + static_cast<int64_t>(moe_src_row(a, slot)) * a.x_stride_m * ES + 16 * c;
    }
    acc_t acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[mt][i] = acc_t{0, 0, 0, 0};

    // two register sets, unconditional loads (steps past the end re-read the last step)
    u32x4 wa[4], xa[MT], wb[4], xb[MT];
    auto load = [&](int s, u32x4 (&w)[4], u32x4 (&x)[MT]) {
      const int k0 = 64 * (wave + 4 * min(s, last));
#pragma unroll
      for (int i = 0; i < 4; ++i) w[i] = ld16g(wrow[i] + k0);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) x[mt] = ld16g(xrow[mt] + k0);
    };
    auto fma_step = [&](const u32x4 (&w)[4], const u32x4 (&x)[MT]) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[mt][i] = moe_mma<I8>(x[mt], w[i], acc[mt][i]);
    };
    if (nsteps > 0) {
      load(0, wa, xa);
      for (int s = 0; s < nsteps; s += 2) {
        load(s + 1, wb, xb);
        __builtin_amdgcn_sched_barrier(0);
        fma_step(wa, xa);
        load(s + 2, wa, xa);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < nsteps) fma_step(wb, xb);
      }
    }

    // acc[mt][i][r] = C[row 16 mt + 4c + r][col 16 i + n]
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v;
          if constexpr (I8) v = __int_as_float(acc[mt][i][r]); else v = acc[mt][i][r];
          s_red[(static_cast<size_t>(wave) * 16 * MT + 16 * mt + 4 * c + r) * kCols + 16 * i + n] = v;
        }
    __syncthreads();
    const int cg = tid & 15;
    const int col = nb * kCols + 4 * cg;
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      const int row = (tid >> 4) + 16 * j;
      const int slot = r0 + row;
      if (slot < row_hi) {
        acc_t sum = acc_t{0, 0, 0, 0};
#pragma unroll
        for (int w = 0; w < 4; ++w)
          sum += *reinterpret_cast<const acc_t*>(s_red + (static_cast<size_t>(w) * 16 * MT + row) * kCols + 4 * cg);
        f32x4 s = moe_sums_to_f32<I8>(a, sum, e, slot, col);
        // routing weight applied to the fp32 sums, one rounding (as the fused expert kernel the
        // reference delegates to)
        if (a.row_scale != nullptr) s *= a.row_scale[a.scale_rows != nullptr ? a.scale_rows[slot] : slot];
        const u32x2 o = {pack_bf16x2(s[0], s[1]), pack_bf16x2(s[2], s[3])};
        *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(a.out) + static_cast<int64_t>(slot) * a.out_stride_m + col) = o;
      }
    }
    __syncthreads();
  }
}

// The same product for K a multiple of 64, with the weights fetched as whole 128-byte lines: a load
// instruction of a wave covers 8 weight rows x 128 bytes (two 32-deep k-steps), lands in a wave-private
// LDS image, and the MFMA B fragments are read back from there.  A load in fragment shape (16 rows x
// 64 bytes) asks the memory system for half lines, and a CU keeps a fixed number of line requests in
// flight whatever their size; whole lines carry twice the bytes per request.  The 16-byte pieces of a
// row are XOR-ed with its row number on the way in, so that the fragment reads are free of bank
// conflicts.  The 4 waves take alternate 64-deep pairs of k-steps.
constexpr int kLineImageBytes = kCols * 128;   // per wave

template <int MT, bool I8>
__global__ __launch_bounds__(256) void moe_gemm_lines_kernel(hymeta_moe_gemm_args a) {
  typedef typename MoeAcc<I8>::type acc_t;
  constexpr int ES = I8 ? 1 : 2;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* s_red = reinterpret_cast<float*>(smem_raw);  // [4 waves][16 MT rows][64 cols] of 4-byte sums, after the k loop

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int c = lane >> 4, n = lane & 15;
  const int nb = blockIdx.x;
  const int e = a.local_expert_begin + blockIdx.y;
  const int row_lo = a.expert_offsets[e], row_hi = a.expert_offsets[e + 1];
  if (row_lo >= row_hi) return;

  char* img = smem_raw + wave * kLineImageBytes;
  // load q of a pair covers image rows 8q .. 8q + 7; this lane: row 8q + (lane >> 3), piece
  // (lane & 7) ^ swizzle(row within its 16-row fragment)
  const int lr = lane >> 3, lp = lane & 7;
  const int64_t kb = static_cast<int64_t>(a.K) * ES;   // bytes of a weight row
  const char* wbase = static_cast<const char*>(a.weights) + static_cast<int64_t>(e) * a.w_stride_e * ES +
                      static_cast<int64_t>(nb * kCols + lr) * kb;
  const char* wld[2] = {wbase + 16 * (lp ^ (lr >> 1)), wbase + 8 * kb + 16 * (lp ^ (4 + (lr >> 1)))};
  const int64_t frag_stride = 16 * kb;
  char* img_st = img + lane * 16;
  const char* img_ld[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) img_ld[h] = img + n * 128 + (((4 * h + c) ^ ((n >> 1) & 7)) * 16);

  const int pairs = static_cast<int>(kb / 128);       // 128-byte lines of a row = pairs of 64-byte k-steps
  const int nsteps = wave < pairs ? (pairs - wave + 3) / 4 : 0;   // this wave's pairs: wave, wave + 4, ...
  const int last = nsteps - 1;

  for (int r0 = row_lo; r0 < row_hi; r0 += 16 * MT) {
    const char* xrow[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int slot = min(r0 + 16 * mt + n, row_hi - 1);   // rows past the expert's end re-read its last row
      xrow[mt] = static_cast<const char*>(a.x) + static_cast<int64_t>(moe_src_row(a, slot)) * a.x_stride_m * ES + 16 * c;
    }
    acc_t acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[mt][i] = acc_t{0, 0, 0, 0};

    // two register sets, unconditional loads (pairs past the end re-read the last pair)
    u32x4 wa[8], xa[MT][2], wb[8], xb[MT][2];
    auto load = [&](int s, u32x4 (&w)[8], u32x4 (&x)[MT][2]) {
      const int k0 = 128 * (wave + 4 * min(s, last));
#pragma unroll
      for (int q = 0; q < 8; ++q) w[q] = ld16g(wld[q & 1] + (q >> 1) * frag_stride + k0);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int h = 0; h < 2; ++h) x[mt][h] = ld16g(xrow[mt] + k0 + 64 * h);
    };
    auto fma_pair = [&](const u32x4 (&w)[8], const u32x4 (&x)[MT][2]) {
#pragma unroll
      for (int q = 0; q < 8; ++q) *reinterpret_cast<u32x4*>(img_st + q * 1024) = w[q];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const u32x4 b = *reinterpret_cast<const u32x4*>(img_ld[h] + i * 2048);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) acc[mt][i] = moe_mma<I8>(x[mt][h], b, acc[mt][i]);
        }
    };
    if (nsteps > 0) {
      load(0, wa, xa);
      for (int s = 0; s < nsteps; s += 2) {
        load(s + 1, wb, xb);
        __builtin_amdgcn_sched_barrier(0);
        fma_pair(wa, xa);
        load(s + 2, wa, xa);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < nsteps) fma_pair(wb, xb);
      }
    }
    __syncthreads();   // the images give way to the partial sums

    // acc[mt][i][r] = C[row 16 mt + 4c + r][col 16 i + n]
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v;
          if constexpr (I8) v = __int_as_float(acc[mt][i][r]); else v = acc[mt][i][r];
          s_red[(static_cast<size_t>(wave) * 16 * MT + 16 * mt + 4 * c + r) * kCols + 16 * i + n] = v;
        }
    __syncthreads();
    const int cg = tid & 15;
    const int col = nb * kCols + 4 * cg;
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      const int row = (tid >> 4) + 16 * j;
      const int slot = r0 + row;
      if (slot < row_hi) {
        acc_t sum = acc_t{0, 0, 0, 0};
#pragma unroll
        for (int w = 0; w < 4; ++w)
          sum += *reinterpret_cast<const acc_t*>(s_red + (static_cast<size_t>(w) * 16 * MT + row) * kCols + 4 * cg);
        f32x4 s = moe_sums_to_f32<I8>(a, sum, e, slot, col);
        if (a.row_scale != nullptr) s *= a.row_scale[a.scale_rows != nullptr ? a.scale_rows[slot] : slot];
        int64_t out_row = slot;
        if (a.combine_tokens != nullptr) {
          // one choice per token, every expert local: the row goes straight to its token, the shared expert added
          // as moe_combine_kernel adds it (to the rounded row)
          out_row = a.combine_tokens[slot];
          if (a.combine_shared != nullptr) {
            const u32x2 sh = *reinterpret_cast<const u32x2*>(reinterpret_cast<const uint16_t*>(a.combine_shared) +
                                                             out_row * a.combine_shared_stride_m + col);
            s[0] = bf16_round(s[0]) + lo_bf(sh[0]), s[1] = bf16_round(s[1]) + hi_bf(sh[0]);
            s[2] = bf16_round(s[2]) + lo_bf(sh[1]), s[3] = bf16_round(s[3]) + hi_bf(sh[1]);
          }
        }
        const u32x2 o = {pack_bf16x2(s[0], s[1]), pack_bf16x2(s[2], s[3])};
        *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(a.out) + out_row * a.out_stride_m + col) = o;
      }
    }
    __syncthreads();
  }
}

// Prefill regime (many rows per expert): BM rows x BN columns per workgroup, WM x WN waves that own
// (BM / WM) x (BN / WN) each.  Both operands are K-contiguous; a 64-deep step of each (128 bytes per
// row) is staged once per workgroup through double-buffered LDS images (16-byte chunks XOR-ed with
// the row, so the sixteen rows of a fragment read hit distinct banks): registers -> LDS right
// behind the barrier that frees the buffer, the next step's loads issued at once.  LDS stores run
// at a third of the read rate, so the tile sets the ceiling: 128 x 128 spends more LDS time than
// matrix time per step, 256 x 256 (64 x 128 per wave) about three quarters of it.
// grid = (N / BN, row tiles); the row tiles of all local experts are numbered consecutively and a
// workgroup finds its (expert, tile) from the offsets on the device -- the host only knows the
// bound ceil(max_rows / BM) + experts.
constexpr int kTK = 64;
constexpr int kTilePitch = 2 * kTK;                 // bytes per staged row
constexpr int kTiledMinRowsPerExpert = 8;           // measured: level at 4 rows per expert, 15 % ahead at 16, 2.6x at 512
constexpr int kBigTileMinRowsPerExpert = 192;

template <int BM, int BN, int WM, int WN, bool I8>
__global__ __launch_bounds__(64 * WM * WN, 512 / (64 * WM * WN)) void moe_gemm_tiled_kernel(hymeta_moe_gemm_args a) {
  typedef typename MoeAcc<I8>::type acc_t;
  constexpr int ES = I8 ? 1 : 2;
  constexpr int kThreads = 64 * WM * WN;
  constexpr int MT = BM / WM / 16, NT = BN / WN / 16;       // 16 x 16 tiles per wave
  constexpr int kXBytes = BM * kTilePitch, kWBytes = BN * kTilePitch;
  constexpr int kRowsPerPass = kThreads / 8;
  constexpr int XP = BM / kRowsPerPass, WP = BN / kRowsPerPass;   // staging passes per operand
  __shared__ __attribute__((aligned(16))) char sm[2][kXBytes + kWBytes];   // [buffer][x | w]

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int c = lane >> 4, n = lane & 15;
  const int wm = wave / WN, wn = wave % WN;

  // which expert, which of its row tiles
  int e = -1, row_lo = 0, row_hi = 0;
  {
    int t = blockIdx.y;
    for (int i = a.local_expert_begin; i < a.local_expert_end; ++i) {
      const int lo = a.expert_offsets[i], hi = a.expert_offsets[i + 1];
      const int tiles = (hi - lo + BM - 1) / BM;
      if (t < tiles) {
        e = i;
        row_lo = lo + t * BM;
        row_hi = hi;
        break;
      }
      t -= tiles;
    }
  }
  if (e < 0) return;                                 // workgroup-uniform
  const int col0 = blockIdx.x * BN;

  // staging role: thread = 16-byte chunk sc of rows sr + kRowsPerPass i of both operand tiles
  const int sr = tid >> 3, sc = tid & 7;
  const int64_t kb = static_cast<int64_t>(a.K) * ES;   // bytes of an operand row
  const char* xp[XP];
  const char* wp[WP];
#pragma unroll
  for (int i = 0; i < XP; ++i) {
    const int slot = min(row_lo + sr + kRowsPerPass * i, row_hi - 1);   // rows past the expert's end re-read its last row
    xp[i] = static_cast<const char*>(a.x) + static_cast<int64_t>(moe_src_row(a, slot)) * a.x_stride_m * ES + 16 * sc;
  }
#pragma unroll
  for (int i = 0; i < WP; ++i)
    wp[i] = static_cast<const char*>(a.weights) + static_cast<int64_t>(e) * a.w_stride_e * ES +
            static_cast<int64_t>(col0 + sr + kRowsPerPass * i) * kb + 16 * sc;
  u32x4 xr[XP], wr[WP];
  auto issue = [&](int it) {
#pragma unroll
    for (int i = 0; i < XP; ++i) xr[i] = ld16g(xp[i] + it * kTilePitch);
#pragma unroll
    for (int i = 0; i < WP; ++i) wr[i] = ld16g(wp[i] + it * kTilePitch);
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      const int r = sr + kRowsPerPass * i;
      st16(&sm[buf][r * kTilePitch + 16 * (sc ^ (r & 7))], xr[i]);
    }
#pragma unroll
    for (int i = 0; i < WP; ++i) {
      const int r = sr + kRowsPerPass * i;
      st16(&sm[buf][kXBytes + r * kTilePitch + 16 * (sc ^ (r & 7))], wr[i]);
    }
  };

  acc_t acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = acc_t{0, 0, 0, 0};

  const int nk = static_cast<int>(kb / kTilePitch);
  issue(0);
  commit(0);
  if (nk > 1) issue(1);
  __syncthreads();
  for (int it = 0; it < nk; ++it) {
    const int buf = it & 1;
    if (it + 1 < nk) commit(buf ^ 1);
    if (it + 2 < nk) issue(it + 2);
    const char* ximg = sm[buf] + (16 * MT * wm + n) * kTilePitch;
    const char* wimg = sm[buf] + kXBytes + (16 * NT * wn + n) * kTilePitch;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int off = 16 * ((4 * ks + c) ^ (n & 7));   // rows 16 t + n: (row & 7) = (n & 7)
      u32x4 xf[MT], wf[NT];
#pragma unroll
      for (int t = 0; t < MT; ++t) xf[t] = ld16(ximg + 16 * t * kTilePitch + off);
#pragma unroll
      for (int t = 0; t < NT; ++t) wf[t] = ld16(wimg + 16 * t * kTilePitch + off);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = moe_mma<I8>(xf[mt], wf[nt], acc[mt][nt]);
    }
    __syncthreads();
  }

  // acc[mt][nt][r] = C[row 16 MT wm + 16 mt + 4c + r][col 16 NT wn + 16 nt + n]: routing weight on the
  // fp32 sums, one rounding, staged through LDS (the operand images are free) 128 rows at a time and
  // stored as whole rows
  constexpr int kCPitch = 2 * BN + 16;
  constexpr int kStageRows = 128;
  static_assert(kStageRows * kCPitch <= 2 * (kXBytes + kWBytes), "output staging fits the operand images");
  static_assert(kStageRows % (16 * MT) == 0 && BM % kStageRows == 0, "a wave's rows fall into one staging pass");
  char* cst = &sm[0][0];
  float wsc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
    wsc[nt] = I8 ? a.w_scale[static_cast<int64_t>(e) * a.N + col0 + 16 * NT * wn + 16 * nt + n] : 1.f;
#pragma unroll
  for (int half = 0; half < BM / kStageRows; ++half) {
    if ((16 * MT * wm) / kStageRows == half) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * MT * wm + 16 * mt + 4 * c + r;
          const int slot = min(row_lo + row, row_hi - 1);
          const float sc_row = a.row_scale != nullptr ? a.row_scale[a.scale_rows != nullptr ? a.scale_rows[slot] : slot] : 1.f;
          uint16_t* crow = reinterpret_cast<uint16_t*>(cst + (row - kStageRows * half) * kCPitch) + 16 * NT * wn + n;
          if constexpr (I8) {
            const float xs = a.x_scale[moe_src_row(a, slot)];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) crow[16 * nt] = f2bf(w8a8_scaled(acc[mt][nt][r], xs, wsc[nt], 0.f) * sc_row);
          } else {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) crow[16 * nt] = f2bf(acc[mt][nt][r] * sc_row);
          }
        }
    }
    __syncthreads();
    constexpr int kChunksPerRow = BN / 8;
    for (int id = tid; id < kStageRows * kChunksPerRow; id += kThreads) {
      const int row = id / kChunksPerRow, ch = id % kChunksPerRow;
      const int slot = row_lo + kStageRows * half + row;
      if (slot < row_hi)
        st16(reinterpret_cast<uint16_t*>(a.out) + static_cast<int64_t>(slot) * a.out_stride_m + col0 + 8 * ch,
             ld16(cst + row * kCPitch + 16 * ch));
    }
    if (half + 1 < BM / kStageRows) __syncthreads();
  }
}

// out[r, j] = silu(in[r, j]) * in[r, inter + j]; the product is formed in fp32 and rounded once.
__global__ __launch_bounds__(256) void silu_and_mul_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out,
                                                           const int32_t* __restrict__ num_rows_dev, int num_rows,
                                                           int inter, int64_t in_stride, int64_t out_stride) {
  const int rows = num_rows_dev != nullptr ? min(*num_rows_dev, num_rows) : num_rows;
  const int chunks = inter / 8;
  const int64_t total = static_cast<int64_t>(rows) * chunks;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < total; i += static_cast<int64_t>(gridDim.x) * 256) {
    const int64_t r = i / chunks;
    const int ch = static_cast<int>(i % chunks);
    const u32x4 g = ld16(in + r * in_stride + 8 * ch);
    const u32x4 u = ld16(in + r * in_stride + inter + 8 * ch);
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float g0 = lo_bf(g[j]), g1 = hi_bf(g[j]);
      o[j] = pack_bf16x2(g0 * sigmoidf(g0) * lo_bf(u[j]), g1 * sigmoidf(g1) * hi_bf(u[j]));
    }
    st16(out + r * out_stride + 8 * ch, o);
  }
}

// The activation and the row's dynamic int8 quantisation in one pass (the input of a W8A8 down projection): one workgroup
// per row forms the bf16 values silu_and_mul_kernel would write, keeps them in registers (NC chunks of 8 per thread) and
// quantises them with the rule of every W8A8 input here.
template <int NC>
__global__ __launch_bounds__(256) void silu_and_mul_int8_kernel(const uint16_t* __restrict__ in, int8_t* __restrict__ out_q,
                                                                float* __restrict__ out_scale,
                                                                const int32_t* __restrict__ num_rows_dev, int num_rows, int inter,
                                                                int64_t in_stride) {
  __shared__ float s_red[4];
  const int rows = num_rows_dev != nullptr ? min(*num_rows_dev, num_rows) : num_rows;
  const int r = blockIdx.x;
  if (r >= rows) return;                                   // workgroup-uniform
  const int chunks = inter / 8;
  const uint16_t* row = in + static_cast<int64_t>(r) * in_stride;
  u32x4 g[NC], u[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int ch = min(static_cast<int>(threadIdx.x) + 256 * i, chunks - 1);
    g[i] = ld16(row + 8 * ch);
    u[i] = ld16(row + inter + 8 * ch);
  }
#pragma unroll
  for (int i = 0; i < NC; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float g0 = lo_bf(g[i][j]), g1 = hi_bf(g[i][j]);
      g[i][j] = pack_bf16x2(g0 * sigmoidf(g0) * lo_bf(u[i][j]), g1 * sigmoidf(g1) * hi_bf(u[i][j]));
    }
  quantise_chunks_int8<NC>(g, chunks, out_q + static_cast<int64_t>(r) * inter, out_scale + r, s_red);
}

// out[t] = sum_k y[pos[t,k]] (+ shared[t]); thread = 8 columns of one token.
__global__ __launch_bounds__(256) void moe_combine_kernel(hymeta_moe_combine_args a) {
  const int chunks = a.hidden / 8;
  const int64_t total = static_cast<int64_t>(a.num_tokens) * chunks;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < total; i += static_cast<int64_t>(gridDim.x) * 256) {
    const int t = static_cast<int>(i / chunks);
    const int ch = static_cast<int>(i % chunks);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int k = 0; k < a.top_k; ++k) {
      const int pos = a.pair_pos[t * a.top_k + k];
      if (pos < 0) continue;
      const u32x4 y = ld16(reinterpret_cast<const uint16_t*>(a.y) + static_cast<int64_t>(pos) * a.y_stride_m + 8 * ch);
#pragma unroll
      for (int j = 0; j < 4; ++j) {   // y already carries its routing weight
        acc[2 * j] += lo_bf(y[j]);
        acc[2 * j + 1] += hi_bf(y[j]);
      }
    }
    if (a.shared != nullptr) {
      const u32x4 s = ld16(reinterpret_cast<const uint16_t*>(a.shared) + static_cast<int64_t>(t) * a.shared_stride_m + 8 * ch);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[2 * j] = bf16_round(acc[2 * j]) + lo_bf(s[j]);
        acc[2 * j + 1] = bf16_round(acc[2 * j + 1]) + hi_bf(s[j]);
      }
    }
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = pack_bf16x2(acc[2 * j], acc[2 * j + 1]);
    st16(reinterpret_cast<uint16_t*>(a.out) + static_cast<int64_t>(t) * a.out_stride_m + 8 * ch, o);
  }
}

}  // namespace hymeta

using namespace hymeta;

extern "C" int hymeta_moe_route(const hymeta_moe_route_args* args, hymeta_stream_t stream) {
  if (!hymeta::args_ok(args)) return HYMETA_ERR_INVALID_ARG;
  const hymeta_moe_route_args& a = *args;
  if (a.num_tokens < 0 || a.num_experts <= 0 || a.top_k <= 0) return HYMETA_ERR_INVALID_ARG;
  if (a.num_experts > kMaxExperts || a.top_k > 2 || a.top_k > a.num_experts) return HYMETA_ERR_UNSUPPORTED;
  if (a.logits_dtype != HYMETA_BF16 && a.logits_dtype != HYMETA_F32) return HYMETA_ERR_UNSUPPORTED;
  if (a.local_expert_begin < 0 || a.local_expert_end > a.num_experts || a.local_expert_begin > a.local_expert_end)
    return HYMETA_ERR_INVALID_ARG;
  if (!a.counts || !a.expert_offsets || !a.cursors) return HYMETA_ERR_INVALID_ARG;
  hipStream_t s = begin_launches(stream);
  hipLaunchKernelGGL(moe_zero_counts_kernel, dim3(1), dim3(64), 0, s, a);      // a kernel, not a memset node: see the header
  if (a.num_tokens > 0) {
    if (!a.router_logits || !a.topk_ids || !a.topk_weights || !a.sorted_pair || !a.pair_pos) return HYMETA_ERR_INVALID_ARG;
    hipLaunchKernelGGL(moe_route_kernel, dim3((a.num_tokens + 255) / 256), dim3(256), 0, s, a);
  }
  hipLaunchKernelGGL(moe_offsets_kernel, dim3(1), dim3(64), 0, s, a);
  if (a.num_tokens > 0)
    hipLaunchKernelGGL(moe_scatter_kernel, dim3((a.num_tokens * a.top_k + 255) / 256), dim3(256), 0, s, a);
  return launch_status();
}

extern "C" int hymeta_moe_router(const hymeta_moe_router_args* args, hymeta_stream_t stream) {
  if (!hymeta::args_ok(args)) return HYMETA_ERR_INVALID_ARG;
  const hymeta_moe_router_args& a = *args;
  if (a.num_tokens < 0 || a.num_experts <= 0 || a.top_k <= 0 || a.hidden <= 0) return HYMETA_ERR_INVALID_ARG;
  if (a.num_experts > kMaxExperts || a.top_k > 2 || a.top_k > a.num_experts) return HYMETA_ERR_UNSUPPORTED;
  if (a.operand_dtype != HYMETA_BF16 && a.operand_dtype != HYMETA_I8) return HYMETA_ERR_UNSUPPORTED;
  const bool i8 = a.operand_dtype == HYMETA_I8;
  if (i8 && (!a.x_scale || !a.gate_scale)) return HYMETA_ERR_INVALID_ARG;
  const int per16 = i8 ? 16 : 8;       // elements of a 16-byte piece
  if (a.num_tokens == 0 || a.num_tokens > HYMETA_MOE_ROUTER_MAX_TOKENS || a.hidden % per16 != 0 || a.x_stride_t % per16 != 0)
    return HYMETA_ERR_UNSUPPORTED;
  if (a.local_expert_begin < 0 || a.local_expert_end > a.num_experts || a.local_expert_begin > a.local_expert_end)
    return HYMETA_ERR_INVALID_ARG;
  if (!a.x || !a.gate_weight || !a.topk_ids || !a.topk_weights || !a.expert_offsets || !a.sorted_pair || !a.pair_pos || !a.ticket)
    return HYMETA_ERR_INVALID_ARG;
  if ((reinterpret_cast<uintptr_t>(a.x) | reinterpret_cast<uintptr_t>(a.gate_weight)) % 16 != 0) return HYMETA_ERR_UNSUPPORTED;
  hipStream_t s = begin_launches(stream);
  if (i8) hipLaunchKernelGGL(moe_router_kernel<true>, dim3(a.num_tokens), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(moe_router_kernel<false>, dim3(a.num_tokens), dim3(256), 0, s, a);
  return launch_status();
}

namespace hymeta {
namespace {
enum class MoeGemmKind { kBigTiles, kTiles, kLines, kFragments };

MoeGemmKind moe_gemm_kind(int64_t max_rows, int experts, int K, int N) {
  // tiled grouped GEMM once the experts average more than a few rows (prefill chunks); the large tile when they fill it
  if (max_rows > static_cast<int64_t>(kTiledMinRowsPerExpert) * experts && K % kTK == 0) {
    if (max_rows >= static_cast<int64_t>(kBigTileMinRowsPerExpert) * experts && N % 256 == 0) return MoeGemmKind::kBigTiles;
    if (N % 128 == 0) return MoeGemmKind::kTiles;
  }
  static const bool lines = [] {
    const char* v = getenv("HYMETA_MOE_LINES");
    return v == nullptr || atoi(v) != 0;
  }();
  return lines && K % 64 == 0 ? MoeGemmKind::kLines : MoeGemmKind::kFragments;
}
}  // namespace
}  // namespace hymeta

extern "C" int hymeta_moe_gemm_fuses(int32_t max_rows, int32_t experts, int32_t K, int32_t N) {
  return experts > 0 && hymeta::moe_gemm_kind(max_rows, experts, K, N) == hymeta::MoeGemmKind::kLines ? 1 : 0;
}

extern "C" int hymeta_moe_gemm(const hymeta_moe_gemm_args* args, hymeta_stream_t stream) {
  using namespace hymeta;
  if (!hymeta::args_ok(args)) return HYMETA_ERR_INVALID_ARG;
  const hymeta_moe_gemm_args& a = *args;
  if (a.max_rows < 0 || a.K <= 0 || a.N <= 0) return HYMETA_ERR_INVALID_ARG;
  if (a.local_expert_begin < 0 || a.local_expert_end < a.local_expert_begin) return HYMETA_ERR_INVALID_ARG;
  if (a.max_rows == 0 || a.local_expert_end == a.local_expert_begin) return HYMETA_OK;
  if (!a.x || !a.weights || !a.out || !a.expert_offsets) return HYMETA_ERR_INVALID_ARG;
  if (a.a_rows != nullptr && a.a_rows_div <= 0) return HYMETA_ERR_INVALID_ARG;
  if (a.operand_dtype != HYMETA_BF16 && a.operand_dtype != HYMETA_I8) return HYMETA_ERR_UNSUPPORTED;
  const bool i8 = a.operand_dtype == HYMETA_I8;
  if (i8 && (!a.x_scale || !a.w_scale || reinterpret_cast<uintptr_t>(a.w_scale) % 16 != 0)) return HYMETA_ERR_INVALID_ARG;
  if (a.K % (i8 ? 64 : 32) != 0 || a.N % kCols != 0) return HYMETA_ERR_UNSUPPORTED;
  if ((a.x_stride_m | a.out_stride_m | a.w_stride_e) % (i8 ? 16 : 8) != 0 || a.out_stride_m % 8 != 0) return HYMETA_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(a.x) | reinterpret_cast<uintptr_t>(a.weights) | reinterpret_cast<uintptr_t>(a.out)) % 16 != 0)
    return HYMETA_ERR_UNSUPPORTED;
  const int experts = a.local_expert_end - a.local_expert_begin;
  const MoeGemmKind kind = moe_gemm_kind(a.max_rows, experts, i8 ? a.K / 2 : a.K, a.N);   // the kind goes by the bytes of a row
  if (a.combine_tokens != nullptr || a.combine_shared != nullptr) {
    // the fused combine exists in the decode-sized kernel with whole-line loads only (hymeta_moe_gemm_fuses)
    if (kind != MoeGemmKind::kLines) return HYMETA_ERR_UNSUPPORTED;
    if (a.combine_shared != nullptr && (a.combine_tokens == nullptr || a.combine_shared_stride_m % 8 != 0 ||
                                        reinterpret_cast<uintptr_t>(a.combine_shared) % 16 != 0))
      return HYMETA_ERR_INVALID_ARG;
  }
  hipStream_t s = begin_launches(stream);
  dim3 grid(a.N / kCols, experts), block(256);
  // few rows per expert on average (decode): 16-row tiles; otherwise 64-row tiles
  const bool small = static_cast<int64_t>(a.max_rows) <= 8LL * experts;
#define HYMETA_MOE_BY_TYPE(LAUNCH_BF16, LAUNCH_I8) do { if (i8) { LAUNCH_I8; } else { LAUNCH_BF16; } } while (0)
  if (kind == MoeGemmKind::kBigTiles) {
    const dim3 tgrid(a.N / 256, (a.max_rows + 255) / 256 + experts);
    HYMETA_MOE_BY_TYPE(hipLaunchKernelGGL((moe_gemm_tiled_kernel<256, 256, 4, 2, false>), tgrid, dim3(512), 0, s, a),
                       hipLaunchKernelGGL((moe_gemm_tiled_kernel<256, 256, 4, 2, true>), tgrid, dim3(512), 0, s, a));
    return launch_status();
  }
  if (kind == MoeGemmKind::kTiles) {
    const dim3 tgrid(a.N / 128, (a.max_rows + 127) / 128 + experts);
    HYMETA_MOE_BY_TYPE(hipLaunchKernelGGL((moe_gemm_tiled_kernel<128, 128, 2, 2, false>), tgrid, dim3(256), 0, s, a),
                       hipLaunchKernelGGL((moe_gemm_tiled_kernel<128, 128, 2, 2, true>), tgrid, dim3(256), 0, s, a));
    return launch_status();
  }
  const size_t red1 = 4 * 16 * kCols * sizeof(float), red4 = 4 * 64 * kCols * sizeof(float);
  if (kind == MoeGemmKind::kLines) {
    const size_t images = 4 * static_cast<size_t>(kLineImageBytes);
    if (small) HYMETA_MOE_BY_TYPE(hipLaunchKernelGGL((moe_gemm_lines_kernel<1, false>), grid, block, std::max(images, red1), s, a),
                                  hipLaunchKernelGGL((moe_gemm_lines_kernel<1, true>), grid, block, std::max(images, red1), s, a));
    else HYMETA_MOE_BY_TYPE(hipLaunchKernelGGL((moe_gemm_lines_kernel<4, false>), grid, block, std::max(images, red4), s, a),
                            hipLaunchKernelGGL((moe_gemm_lines_kernel<4, true>), grid, block, std::max(images, red4), s, a));
    return launch_status();
  }
  if (small) HYMETA_MOE_BY_TYPE(hipLaunchKernelGGL((moe_gemm_kernel<1, false>), grid, block, red1, s, a),
                                hipLaunchKernelGGL((moe_gemm_kernel<1, true>), grid, block, red1, s, a));
  else HYMETA_MOE_BY_TYPE(hipLaunchKernelGGL((moe_gemm_kernel<4, false>), grid, block, red4, s, a),
                          hipLaunchKernelGGL((moe_gemm_kernel<4, true>), grid, block, red4, s, a));
#undef HYMETA_MOE_BY_TYPE
  return launch_status();
}

extern "C" int hymeta_silu_and_mul(const void* in, void* out, const int32_t* num_rows_dev, int32_t num_rows,
                                   int32_t inter, int64_t in_stride, int64_t out_stride, hymeta_stream_t stream) {
  if (num_rows < 0 || inter <= 0) return HYMETA_ERR_INVALID_ARG;
  if (num_rows == 0) return HYMETA_OK;
  if (!in || !out) return HYMETA_ERR_INVALID_ARG;
  if (inter % 8 != 0 || (in_stride | out_stride) % 8 != 0 ||
      (reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) % 16 != 0)
    return HYMETA_ERR_UNSUPPORTED;
  const int64_t total = static_cast<int64_t>(num_rows) * (inter / 8);
  int64_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(silu_and_mul_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, begin_launches(stream),
                     static_cast<const uint16_t*>(in), static_cast<uint16_t*>(out), num_rows_dev, num_rows, inter, in_stride,
                     out_stride);
  return launch_status();
}

extern "C" int hymeta_silu_and_mul_int8(const void* in, int8_t* out_q, float* out_scale, const int32_t* num_rows_dev,
                                        int32_t num_rows, int32_t inter, int64_t in_stride, hymeta_stream_t stream) {
  if (num_rows < 0 || inter <= 0) return HYMETA_ERR_INVALID_ARG;
  if (num_rows == 0) return HYMETA_OK;
  if (!in || !out_q || !out_scale) return HYMETA_ERR_INVALID_ARG;
  if (inter % 8 != 0 || inter > 24576 || in_stride % 8 != 0 ||
      (reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out_q)) % 16 != 0)
    return HYMETA_ERR_UNSUPPORTED;
  const uint16_t* src = static_cast<const uint16_t*>(in);
  hipStream_t s = begin_launches(stream);
  const int chunks = inter / 8;
#define HYMETA_SILU_Q(NC) \
  hipLaunchKernelGGL(silu_and_mul_int8_kernel<NC>, dim3(num_rows), dim3(256), 0, s, src, out_q, out_scale, num_rows_dev, num_rows, inter, in_stride)
  if (chunks <= 256) HYMETA_SILU_Q(1);
  else if (chunks <= 768) HYMETA_SILU_Q(3);
  else if (chunks <= 1536) HYMETA_SILU_Q(6);
  else HYMETA_SILU_Q(12);
#undef HYMETA_SILU_Q
  return launch_status();
}

extern "C" int hymeta_moe_combine(const hymeta_moe_combine_args* args, hymeta_stream_t stream) {
  if (!hymeta::args_ok(args)) return HYMETA_ERR_INVALID_ARG;
  const hymeta_moe_combine_args& a = *args;
  if (a.num_tokens < 0 || a.hidden <= 0 || a.top_k <= 0) return HYMETA_ERR_INVALID_ARG;
  if (a.num_tokens == 0) return HYMETA_OK;
  if (!a.y || !a.out || !a.pair_pos) return HYMETA_ERR_INVALID_ARG;
  if (a.hidden % 8 != 0 || (a.y_stride_m | a.out_stride_m | a.shared_stride_m) % 8 != 0 ||
      (reinterpret_cast<uintptr_t>(a.y) | reinterpret_cast<uintptr_t>(a.out) | reinterpret_cast<uintptr_t>(a.shared)) % 16 != 0)
    return HYMETA_ERR_UNSUPPORTED;
  const int64_t total = static_cast<int64_t>(a.num_tokens) * (a.hidden / 8);
  int64_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(moe_combine_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, begin_launches(stream), a);
  return launch_status();
}
