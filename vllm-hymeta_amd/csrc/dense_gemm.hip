// bf16 linear for decode-sized batches (up to 64 rows): out = x W^T + bias with W [N, K] as torch
// stores a Linear weight.  The weights are the traffic: each is read once, straight from global
// memory into the MFMA B operand (both x rows and W rows are K-contiguous, so a lane's 16 bytes are
// a fragment as they stand), the activations come from L2.
// Measured on MI355X with cold weights against the library GEMM: ahead only on deep, narrow
// shapes, where the library does not cut K (7B down projection, K 18944 -> N 3584: 61 vs 91 us at
// 64 rows, 40 vs 48 at one row); behind on the wide ones (every workgroup re-reads the
// activations from L2).  The host side therefore routes only K >= 4 N here.
//
// grid = (N / 64, K slices); 256 threads: the 4 waves take alternate 32-deep k-steps of the
// workgroup's slice with two register sets (the next step's loads in flight during the MFMAs),
// meet in LDS, and the K slices of one 64-column tile meet in fp32 slabs: the last slice to
// arrive sums them in slice order (deterministic), adds the bias and writes bf16.
#include "common.h"
#include "stream_gemm.h"

namespace hymeta {
namespace {

constexpr int kCols = 64;

__device__ __forceinline__ f32x4 mfma16(const bf16x8& a, const bf16x8& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ bf16x8 as_frag(const u32x4& v) { return __builtin_bit_cast(bf16x8, v); }

template <int MT>
__global__ __launch_bounds__(256) void bf16_skinny_gemm_kernel(hymeta_bf16_gemm_args a, int splits, int steps_per_split,
                                                               float* slabs, int* counters) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* s_red = reinterpret_cast<float*>(smem_raw);  // [4 waves][16 MT rows][64 cols]

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int c = lane >> 4, n = lane & 15;
  const int nb = blockIdx.x;
  const int split = blockIdx.y;

  const uint16_t* wrow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
    wrow[i] = reinterpret_cast<const uint16_t*>(a.weight) + static_cast<int64_t>(nb * kCols + 16 * i + n) * a.w_stride_n + 8 * c;
  const uint16_t* xrow[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)   // rows past M re-read the last row; their results are not stored
    xrow[mt] = reinterpret_cast<const uint16_t*>(a.x) + static_cast<int64_t>(min(16 * mt + n, a.M - 1)) * a.x_stride_m + 8 * c;

  const int ks_lo = split * steps_per_split;
  const int ks_hi = min(a.K / 32, ks_lo + steps_per_split);
  const int mine = ks_hi - ks_lo > wave ? (ks_hi - ks_lo - wave + 3) / 4 : 0;   // this wave's k-steps: ks_lo + wave, + 4, ...
  const int last = mine - 1;

  f32x4 acc[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[mt][i] = {0.f, 0.f, 0.f, 0.f};

  // two register sets, unconditional loads (steps past the end re-read the last step)
  u32x4 wa[4], xa[MT], wb[4], xb[MT];
  auto load = [&](int s, u32x4 (&w)[4], u32x4 (&x)[MT]) {
    const int k0 = 32 * (ks_lo + wave + 4 * min(s, last));
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = ld16g(wrow[i] + k0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) x[mt] = ld16g(xrow[mt] + k0);
  };
  auto fma_step = [&](const u32x4 (&w)[4], const u32x4 (&x)[MT]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[mt][i] = mfma16(as_frag(x[mt]), as_frag(w[i]), acc[mt][i]);
  };
  if (mine > 0) {
    load(0, wa, xa);
    for (int s = 0; s < mine; s += 2) {
      load(s + 1, wb, xb);
      __builtin_amdgcn_sched_barrier(0);
      fma_step(wa, xa);
      load(s + 2, wa, xa);
      __builtin_amdgcn_sched_barrier(0);
      if (s + 1 < mine) fma_step(wb, xb);
    }
  }

  // acc[mt][i][r] = C[row 16 mt + 4c + r][col 16 i + n]
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        s_red[(static_cast<size_t>(wave) * 16 * MT + 16 * mt + 4 * c + r) * kCols + 16 * i + n] = acc[mt][i][r];
  __syncthreads();
  const int cg = tid & 15;
  const int col = nb * kCols + 4 * cg;
  f32x4 sum[MT];
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    const int row = (tid >> 4) + 16 * j;
    sum[j] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < 4; ++w)
      sum[j] += *reinterpret_cast<const f32x4*>(s_red + (static_cast<size_t>(w) * 16 * MT + row) * kCols + 4 * cg);
  }
  if (splits > 1) {
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      const int row = (tid >> 4) + 16 * j;
      if (row < a.M) *reinterpret_cast<f32x4*>(slabs + (static_cast<int64_t>(split) * a.M + row) * a.N + col) = sum[j];
    }
    if (!last_arriver(counters + nb, splits, reinterpret_cast<int*>(s_red))) return;
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      const int row = (tid >> 4) + 16 * j;
      sum[j] = {0.f, 0.f, 0.f, 0.f};
      if (row < a.M)
        for (int sp = 0; sp < splits; ++sp)
          sum[j] += *reinterpret_cast<const f32x4*>(slabs + (static_cast<int64_t>(sp) * a.M + row) * a.N + col);
    }
  }
  f32x4 bias = {0.f, 0.f, 0.f, 0.f};
  if (a.bias != nullptr) {
    const u32x2 bw = *reinterpret_cast<const u32x2*>(reinterpret_cast<const uint16_t*>(a.bias) + col);
    bias = f32x4{lo_bf(bw[0]), hi_bf(bw[0]), lo_bf(bw[1]), hi_bf(bw[1])};
  }
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    const int row = (tid >> 4) + 16 * j;
    if (row >= a.M) continue;
    const f32x4 y = sum[j] + bias;
    const u32x2 o = {pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3])};
    *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(a.out) + static_cast<int64_t>(row) * a.out_stride_m + col) = o;
  }
}

}  // namespace
}  // namespace hymeta

using namespace hymeta;

extern "C" int hymeta_bf16_gemm_split_k(int32_t M, int32_t K, int32_t N) {
  if (M <= 0 || K <= 0 || N <= 0) return 1;
  const int64_t tiles = N / kCols > 0 ? N / kCols : 1;
  int64_t want = 256 / tiles;                                   // about one workgroup per compute unit (measured best)
  const int64_t cap = K / 32 / 16 > 1 ? K / 32 / 16 : 1;        // at least four k-steps per wave
  if (want > cap) want = cap;
  if (want < 1) want = 1;
  return static_cast<int>(want);
}

extern "C" size_t hymeta_bf16_gemm_workspace_bytes(int32_t M, int32_t N, int32_t split_k) {
  if (M <= 0 || N <= 0) return 0;
  const size_t counters = kTicketBytes;   // zero at first use, left zero by every call
  if (split_k <= 0) {   // a bound for whatever the library chooses: tiles x slices <= 256 + tiles, or the streaming kernel's slabs
    const size_t tiles = N / kCols > 0 ? N / kCols : 1;
    const size_t old_bound = counters + (256 + tiles) * kCols * static_cast<size_t>(M) * sizeof(float);
    const size_t stream_bound = counters + stream_gemm_workspace_bound(M, N);
    return old_bound > stream_bound ? old_bound : stream_bound;
  }
  if (split_k == 1) return 0;
  return counters + static_cast<size_t>(split_k) * M * N * sizeof(float);
}

extern "C" int hymeta_bf16_gemm_takes(int32_t M, int32_t K, int32_t N, int32_t form) {
  if (M <= 0 || K <= 0 || N <= 0 || M > 256 || K % 32 != 0 || N % kCols != 0 || K > (1 << 28)) return 0;
  StreamPlan plan;
  if (form == 1) return stream_gemm_plan_gated(M, K * 2, N, &plan) ? 1 : 0;
  if (form == 2) return stream_gemm_plan(M, K * 2, N, false, &plan) && plan.splits == 1 ? 1 : 0;
  if (M <= 64) return 1;
  return stream_gemm_plan(M, K * 2, N, true, &plan) ? 1 : 0;
}

static int bf16_gemm_impl(const hymeta_bf16_gemm_args* args, hymeta_stream_t stream, hymeta_gemm_partials* deferred) {
  if (!hymeta::args_ok(args)) return HYMETA_ERR_INVALID_ARG;
  if (deferred != nullptr) *deferred = hymeta::make_args<hymeta_gemm_partials>();
  const hymeta_bf16_gemm_args& a = *args;
  if (a.M < 0 || a.K <= 0 || a.N <= 0) return HYMETA_ERR_INVALID_ARG;
  if (a.M == 0) return HYMETA_OK;
  if (!a.x || !a.weight || !a.out) return HYMETA_ERR_INVALID_ARG;
  if (a.M > 256 || a.K % 32 != 0 || a.N % kCols != 0) return HYMETA_ERR_UNSUPPORTED;
  if ((a.x_stride_m | a.w_stride_n | a.out_stride_m) % 8 != 0) return HYMETA_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(a.x) | reinterpret_cast<uintptr_t>(a.weight) | reinterpret_cast<uintptr_t>(a.out) |
       reinterpret_cast<uintptr_t>(a.bias) | reinterpret_cast<uintptr_t>(a.workspace)) % 16 != 0)
    return HYMETA_ERR_UNSUPPORTED;
  // the streaming kernel (activations shared through LDS, work dealt in equal ranges over the whole chip) takes every
  // shape it covers when the library chooses the decomposition
  if (a.split_k <= 0) {
    StreamPlan plan;
    const bool have_ws = a.workspace != nullptr;
    if (stream_gemm_plan(a.M, a.K * 2, a.N, have_ws, &plan) &&
        (!have_ws || a.workspace_bytes >= kTicketBytes + stream_gemm_workspace_bytes(a.M, a.N, plan))) {
      StreamArgs sa = {};
      sa.x = static_cast<const char*>(a.x), sa.w = static_cast<const char*>(a.weight);
      sa.bias = static_cast<const uint16_t*>(a.bias), sa.out = static_cast<uint16_t*>(a.out);
      sa.slabs = have_ws ? static_cast<char*>(a.workspace) + kTicketBytes : nullptr;
      sa.M = a.M, sa.N = a.N, sa.k_bytes = a.K * 2;
      sa.x_stride_bytes = a.x_stride_m * 2, sa.w_stride_bytes = a.w_stride_n * 2, sa.out_stride = a.out_stride_m;
      if (deferred != nullptr && plan.splits > 1) {
        deferred->slabs = sa.slabs, deferred->splits = plan.splits, deferred->bias = a.bias;
        deferred->M = a.M, deferred->N = a.N;
      }
      return stream_gemm_launch<false>(sa, plan, begin_launches(stream), deferred != nullptr);
    }
  }
  if (a.M > 64) return HYMETA_ERR_UNSUPPORTED;      // 65 to 256 rows: the streaming kernels only
  const int ksteps = a.K / 32;
  int splits = a.split_k > 0 ? a.split_k : (a.workspace != nullptr ? hymeta_bf16_gemm_split_k(a.M, a.K, a.N) : 1);
  if (splits > ksteps) splits = ksteps;
  if (a.N / kCols > kMaxTickets) splits = 1;
  const int sps = (ksteps + splits - 1) / splits;
  splits = (ksteps + sps - 1) / sps;
  const size_t counter_bytes = kTicketBytes;
  if (splits > 1 && (a.workspace == nullptr ||
                     a.workspace_bytes < counter_bytes + static_cast<size_t>(splits) * a.M * a.N * sizeof(float)))
    return HYMETA_ERR_WORKSPACE;
  hipStream_t s = begin_launches(stream);
  int* counters = static_cast<int*>(a.workspace);
  float* slabs = reinterpret_cast<float*>(static_cast<char*>(a.workspace) + counter_bytes);
  const int mt = a.M <= 16 ? 1 : (a.M <= 32 ? 2 : 4);
  const dim3 grid(a.N / kCols, splits), block(256);
  const size_t lds = static_cast<size_t>(4) * 16 * mt * kCols * sizeof(float);
  if (mt == 1) hipLaunchKernelGGL((bf16_skinny_gemm_kernel<1>), grid, block, lds, s, a, splits, sps, slabs, counters);
  else if (mt == 2) hipLaunchKernelGGL((bf16_skinny_gemm_kernel<2>), grid, block, lds, s, a, splits, sps, slabs, counters);
  else hipLaunchKernelGGL((bf16_skinny_gemm_kernel<4>), grid, block, lds, s, a, splits, sps, slabs, counters);
  return launch_status();
}

extern "C" int hymeta_bf16_gemm(const hymeta_bf16_gemm_args* args, hymeta_stream_t stream) {
  return bf16_gemm_impl(args, stream, nullptr);
}

extern "C" int hymeta_bf16_gemm_deferred(const hymeta_bf16_gemm_args* args, hymeta_gemm_partials* partials,
                                         hymeta_stream_t stream) {
  if (!hymeta::args_ok(partials)) return HYMETA_ERR_INVALID_ARG;
  return bf16_gemm_impl(args, stream, partials);
}

// fp32 result of bf16 operands, unrounded: the logits of a bf16 lm_head.  The streaming kernel without K slices only.
extern "C" int hymeta_bf16_gemm_f32out(const hymeta_bf16_gemm_args* args, hymeta_stream_t stream) {
  using namespace hymeta;
  if (!hymeta::args_ok(args)) return HYMETA_ERR_INVALID_ARG;
  const hymeta_bf16_gemm_args& a = *args;
  if (a.M < 0 || a.K <= 0 || a.N <= 0) return HYMETA_ERR_INVALID_ARG;
  if (a.M == 0) return HYMETA_OK;
  if (!a.x || !a.weight || !a.out) return HYMETA_ERR_INVALID_ARG;
  if ((a.x_stride_m | a.w_stride_n) % 8 != 0 || a.out_stride_m % 4 != 0) return HYMETA_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(a.x) | reinterpret_cast<uintptr_t>(a.weight) | reinterpret_cast<uintptr_t>(a.out) |
       reinterpret_cast<uintptr_t>(a.bias)) % 16 != 0)
    return HYMETA_ERR_UNSUPPORTED;
  StreamPlan plan;
  if (a.K > (1 << 28) || !stream_gemm_plan(a.M, a.K * 2, a.N, false, &plan) || plan.splits != 1) return HYMETA_ERR_UNSUPPORTED;
  StreamArgs sa = {};
  sa.x = static_cast<const char*>(a.x), sa.w = static_cast<const char*>(a.weight);
  sa.bias = static_cast<const uint16_t*>(a.bias), sa.out = static_cast<uint16_t*>(a.out);
  sa.M = a.M, sa.N = a.N, sa.k_bytes = a.K * 2, sa.out_f32 = 1;
  sa.x_stride_bytes = a.x_stride_m * 2, sa.w_stride_bytes = a.w_stride_n * 2, sa.out_stride = a.out_stride_m;
  return stream_gemm_launch<false>(sa, plan, begin_launches(stream));
}

// Gate/up projection with the activation in the epilogue: out [M, N / 2] = silu(x gate^T) * (x up^T), weight = [gate rows |
// up rows].  Only the shapes the streaming kernel takes whole (no K slices); others are HYMETA_ERR_UNSUPPORTED and the
// caller runs hymeta_bf16_gemm + hymeta_silu_and_mul, which give the same bits.
extern "C" int hymeta_bf16_gemm_silu_mul(const hymeta_bf16_gemm_args* args, hymeta_stream_t stream) {
  using namespace hymeta;
  if (!hymeta::args_ok(args)) return HYMETA_ERR_INVALID_ARG;
  const hymeta_bf16_gemm_args& a = *args;
  if (a.M < 0 || a.K <= 0 || a.N <= 0 || a.bias != nullptr) return HYMETA_ERR_INVALID_ARG;
  if (a.M == 0) return HYMETA_OK;
  if (!a.x || !a.weight || !a.out) return HYMETA_ERR_INVALID_ARG;
  if ((a.x_stride_m | a.w_stride_n | a.out_stride_m) % 8 != 0) return HYMETA_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(a.x) | reinterpret_cast<uintptr_t>(a.weight) | reinterpret_cast<uintptr_t>(a.out)) % 16 != 0)
    return HYMETA_ERR_UNSUPPORTED;
  StreamPlan plan;
  if (a.K > (1 << 28) || !stream_gemm_plan_gated(a.M, a.K * 2, a.N, &plan)) return HYMETA_ERR_UNSUPPORTED;
  StreamArgs sa = {};
  sa.x = static_cast<const char*>(a.x), sa.w = static_cast<const char*>(a.weight);
  sa.out = static_cast<uint16_t*>(a.out);
  sa.M = a.M, sa.N = a.N, sa.k_bytes = a.K * 2, sa.gated = 1;
  sa.x_stride_bytes = a.x_stride_m * 2, sa.w_stride_bytes = a.w_stride_n * 2, sa.out_stride = a.out_stride_m;
  return stream_gemm_launch<false>(sa, plan, begin_launches(stream));
}
