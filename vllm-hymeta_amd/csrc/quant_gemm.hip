// Fused dequant-GEMMs for the quantised linears around the attention kernels:
// GPTQ 4/8-bit (group-wise scales and zeros) and W8A8 int8.
//
// Decode-sized M (<= 64 rows per workgroup pass) is the design point: the weight
// stream is read once, coalesced, and de-quantised in registers on its way into
// the matrix cores; larger M re-runs the same body per 64-row block.
//
// GPTQ: a packed int32 of qweight holds consecutive k of ONE output column,
// which is exactly what the MFMA B-operand wants per lane (8 k-slots of column
// n), so weights go global -> VGPR -> MFMA with no LDS and no shuffles:
//   4-bit: (q >> 4w) & 0x000F000F | 0x43004300 turns two nibbles into the bf16
//          pair (128 + q_w, 128 + q_{w+4}) in ONE VALU op; the k-order this
//          induces (0,4,1,5,2,6,3,7) is applied to the activation fragment too.
//   8-bit: v_cvt_f32_ubyteN + a byte-permute keep the exact integer in bf16.
// Split-K partial tiles meet in fp32 slabs; the last workgroup to arrive at a tile sums and writes it
// (no atomics on data, no second launch).
// Scales and zeros are applied once per group, outside the MFMA chain:
//   sum_k x w = s * ( sum_k x (c + q)  -  (c + z) * sum_k x ),   c = 128 (4-bit) or 0,
// where sum_k x comes from one extra MFMA against an all-ones B fragment.
#include <type_traits>

#include <cstdlib>

#include "common.h"
#include "stream_gemm.h"
#include "w8a8_quant.h"

namespace hymeta {
namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma16(const bf16x8& a, const bf16x8& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ bf16x8 as_frag(const u32x4& v) { return __builtin_bit_cast(bf16x8, v); }

__device__ __forceinline__ float load_scale(const void* p, int64_t idx, int dtype) {
  const uint16_t raw = reinterpret_cast<const uint16_t*>(p)[idx];
  if (dtype == HYMETA_F16) return static_cast<float>(__builtin_bit_cast(_Float16, raw));
  return bf2f(raw);
}

constexpr int kColsPerBlock = 64;  // output columns per workgroup: lane n owns columns 4n .. 4n+3

}  // namespace

// grid = (N / 64, ceil(M / (16 MT)), splits); 256 threads: the 4 waves take alternate groups of
// the workgroup's K slice; each wave's (group, k-step) sequence is software-pipelined one step
// deep (next step's packed weights and activation fragments are in flight during the MFMAs).
template <int BITS, int MT, int KSTEPS>
__global__ __launch_bounds__(256) void gptq_gemm_kernel(hymeta_gptq_gemm_args a, int splits, int groups_per_split,
                                                        float* slabs, int* counters) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* s_red = reinterpret_cast<float*>(smem_raw);  // [4 waves][16 MT rows][64 cols]

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int c = lane >> 4, n = lane & 15;
  const int nb = blockIdx.x;
  const int m0 = blockIdx.y * 16 * MT;
  const int split = blockIdx.z;
  const int G = a.K / a.group_size;
  const int g_lo = split * groups_per_split;
  const int g_hi = min(G, g_lo + groups_per_split);
  const int col0 = nb * kColsPerBlock + 4 * n;   // this lane's first output column
  constexpr int PACK = 32 / BITS;
  constexpr int QL = BITS == 4 ? 1 : 2;          // 16-byte weight loads per k-step

  f32x4 acc_tot[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc_tot[mt][i] = {0.f, 0.f, 0.f, 0.f};

  const u32x4 ones_raw = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
  const bf16x8 ones = as_frag(ones_raw);
  const uint16_t* xbase = reinterpret_cast<const uint16_t*>(a.x);
  // rows past M read row M - 1 (their results are never stored)
  const uint16_t* xrow[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) xrow[mt] = xbase + static_cast<int64_t>(min(m0 + 16 * mt + n, a.M - 1)) * a.x_stride_m + 8 * c;

  const int first_g = g_lo + wave;
  const int ng = first_g < g_hi ? (g_hi - first_g + 3) / 4 : 0;

  // Register rings, refilled slot by slot as soon as a step has consumed its slot: packed weights
  // (HBM latency, 16 bytes per lane and step) two groups ahead, activation fragments (L2 hits,
  // 16 MT bytes) kXDepth steps ahead, a group's scales and zeros one group ahead.  Control flow is
  // uniform and every load unconditional (steps past the end re-read the last step; an odd group
  // count is padded with a group of zero scale): with conditional issue the compiler cannot count
  // the loads in flight and drains them all at the next use.
  constexpr int kWDepth = 2 * KSTEPS;
  constexpr int kXCap = MT == 1 ? 8 : (MT == 2 ? 4 : 2);
  constexpr int kXDepth = kWDepth < kXCap ? kWDepth : kXCap;
  const int ngp = (ng + 1) & ~1;
  const int last_step = ng * KSTEPS - 1;
  u32x4 qring[kWDepth][QL], xring[kXDepth][MT];
  u32x2 sraw[2];     // 4 scales (16-bit) of the lane's columns
  uint32_t zraw[2];  // packed zeros word holding the lane's columns
  auto step_k0 = [&](int t) {
    const int g = first_g + 4 * (t / KSTEPS);
    return g * a.group_size + 32 * (t % KSTEPS);   // + 8c inside the pointers
  };
  auto load_w = [&](int t, u32x4 (&q)[QL]) {
    const int32_t* qp = a.qweight + static_cast<int64_t>((step_k0(min(t, last_step)) + 8 * c) / PACK) * a.N + col0;
#pragma unroll
    for (int i = 0; i < QL; ++i) q[i] = ld16(qp + static_cast<int64_t>(i) * a.N);
  };
  auto load_x = [&](int t, u32x4 (&x)[MT]) {
    const int k0 = step_k0(min(t, last_step));
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) x[mt] = ld16(xrow[mt] + k0);
  };
  auto load_params = [&](int gi, u32x2& sr, uint32_t& zr) {
    const int g = first_g + 4 * min(gi, ng - 1);
    sr = *reinterpret_cast<const u32x2*>(reinterpret_cast<const uint16_t*>(a.scales) + static_cast<int64_t>(g) * a.N + col0);
    zr = static_cast<uint32_t>(a.qzeros[static_cast<int64_t>(g) * (a.N / PACK) + col0 / PACK]);
  };

  auto do_group = [&](int gi, auto parity) {
    constexpr int PAR = decltype(parity)::value;
    load_params(gi + 1, sraw[PAR ^ 1], zraw[PAR ^ 1]);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc_g[MT][4], acc_x[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      acc_x[mt] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 4; ++i) acc_g[mt][i] = {0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      constexpr int kDummy = 0;
      (void)kDummy;
      const int t = gi * KSTEPS + ks;
      const int wslot = PAR * KSTEPS + ks;
      const int xslot = wslot % kXDepth;
      // ---- weights: 8 k-values x 4 columns per lane
      u32x4 bq[4];
      if constexpr (BITS == 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int w = 0; w < 4; ++w) bq[i][w] = ((qring[wslot][0][i] >> (4 * w)) & 0x000F000Fu) | 0x43004300u;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const uint32_t lohi[2] = {qring[wslot][0][i], qring[wslot][QL - 1][i]};
#pragma unroll
          for (int hw = 0; hw < 2; ++hw) {
            const float f0 = static_cast<float>(lohi[hw] & 0xffu);
            const float f1 = static_cast<float>((lohi[hw] >> 8) & 0xffu);
            const float f2 = static_cast<float>((lohi[hw] >> 16) & 0xffu);
            const float f3 = static_cast<float>(lohi[hw] >> 24);
            // integers <= 255 are exact in bf16: keep the high halves
            bq[i][2 * hw] = __builtin_amdgcn_perm(__float_as_uint(f1), __float_as_uint(f0), 0x07060302u);
            bq[i][2 * hw + 1] = __builtin_amdgcn_perm(__float_as_uint(f3), __float_as_uint(f2), 0x07060302u);
          }
        }
      }
      load_w(t + kWDepth, qring[wslot]);
      __builtin_amdgcn_sched_barrier(0);   // keep the refill here: the scheduler would sink it to its use
      // ---- activations: 16 rows x 8 k per m-tile
      bf16x8 af[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        u32x4 xr = xring[xslot][mt];
        if constexpr (BITS == 4) {
          // k-slot order (0,4,1,5,2,6,3,7), the order the nibble trick emits
          u32x4 xp;
          xp[0] = __builtin_amdgcn_perm(xr[2], xr[0], 0x05040100u);
          xp[1] = __builtin_amdgcn_perm(xr[2], xr[0], 0x07060302u);
          xp[2] = __builtin_amdgcn_perm(xr[3], xr[1], 0x05040100u);
          xp[3] = __builtin_amdgcn_perm(xr[3], xr[1], 0x07060302u);
          xr = xp;
        }
        af[mt] = as_frag(xr);
      }
      load_x(t + kXDepth, xring[xslot]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc_g[mt][i] = mfma16(af[mt], as_frag(bq[i]), acc_g[mt][i]);
        acc_x[mt] = mfma16(af[mt], ones, acc_x[mt]);
      }
    }
    // ---- group epilogue: s * (sum x (c + q) - (c + z) * sum x), zero scale for a padding group
    const bool real = gi < ng;
    const uint32_t zw = zraw[PAR];
    const int first = col0 % PACK;
    float sc[4], zf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint16_t raw = static_cast<uint16_t>(sraw[PAR][i >> 1] >> (16 * (i & 1)));
      const float sv = a.scales_dtype == HYMETA_F16 ? static_cast<float>(__builtin_bit_cast(_Float16, raw)) : bf2f(raw);
      sc[i] = real ? sv : 0.f;
      const uint32_t z = (zw >> (BITS * (first + i))) & ((1u << BITS) - 1u);
      zf[i] = static_cast<float>(z + a.zero_offset) + (BITS == 4 ? 128.f : 0.f);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          acc_tot[mt][i][r] += sc[i] * (acc_g[mt][i][r] - zf[i] * acc_x[mt][r]);
  };

  if (ng > 0) {
#pragma unroll
    for (int p = 0; p < kWDepth; ++p) load_w(p, qring[p]);
#pragma unroll
    for (int p = 0; p < kXDepth; ++p) load_x(p, xring[p]);
    load_params(0, sraw[0], zraw[0]);
    for (int gi = 0; gi < ngp; gi += 2) {
      do_group(gi, std::integral_constant<int, 0>{});
      do_group(gi + 1, std::integral_constant<int, 1>{});
    }
  }

  // ---- reduce the 4 waves: s_red[wave][row][col], row = 16 mt + 4c + r, col = 4n + i
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float* p = s_red + (static_cast<size_t>(wave) * 16 * MT + 16 * mt + 4 * c + r) * kColsPerBlock + 4 * n;
      *reinterpret_cast<f32x4*>(p) = f32x4{acc_tot[mt][0][r], acc_tot[mt][1][r], acc_tot[mt][2][r], acc_tot[mt][3][r]};
    }
  __syncthreads();
  const int cg = tid & 15;          // 4 columns per thread
  const int col = nb * kColsPerBlock + 4 * cg;
  f32x4 sum[MT];
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    const int row = (tid >> 4) + 16 * j;
    sum[j] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < 4; ++w)
      sum[j] += *reinterpret_cast<const f32x4*>(s_red + (static_cast<size_t>(w) * 16 * MT + row) * kColsPerBlock + 4 * cg);
  }
  if (splits > 1) {
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      const int grow = m0 + (tid >> 4) + 16 * j;
      if (grow < a.M) *reinterpret_cast<f32x4*>(slabs + (static_cast<int64_t>(split) * a.M + grow) * a.N + col) = sum[j];
    }
    if (!last_arriver(counters + blockIdx.y * gridDim.x + blockIdx.x, splits, reinterpret_cast<int*>(s_red))) return;
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      const int grow = m0 + (tid >> 4) + 16 * j;
      sum[j] = {0.f, 0.f, 0.f, 0.f};
      if (grow < a.M)
        for (int sp = 0; sp < splits; ++sp)
          sum[j] += *reinterpret_cast<const f32x4*>(slabs + (static_cast<int64_t>(sp) * a.M + grow) * a.N + col);
    }
  }
  f32x4 bias = {0.f, 0.f, 0.f, 0.f};
  if (a.bias != nullptr) {
    const u32x2 bw = *reinterpret_cast<const u32x2*>(reinterpret_cast<const uint16_t*>(a.bias) + col);
    bias = f32x4{lo_bf(bw[0]), hi_bf(bw[0]), lo_bf(bw[1]), hi_bf(bw[1])};
  }
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    const int grow = m0 + (tid >> 4) + 16 * j;
    if (grow >= a.M) continue;
    const f32x4 y = sum[j] + bias;
    const u32x2 o = {pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3])};
    *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(a.out) + static_cast<int64_t>(grow) * a.out_stride_m + col) = o;
  }
}

// ---------------------------------------------------------------------------------------------
// The 33..64-row regime.  There the kernel above loses twice: its waves split K, so each fetches
// the whole 64 x K activation block (four times the bytes of the weights it streams) straight
// from L2 in fragment-shaped pieces, and at 64 rows the loop is bound by instruction issue (one
// de-quantisation instruction per 1.1 weights beside 2 * 64 flop per weight), not by memory.
// This kernel takes 128 columns per workgroup of 8 waves = 2 K halves x 2 row halves x 2 column
// halves: the activation block is staged once per workgroup, two 128-deep chunks per barrier, in
// full 256-byte row pieces, into double-buffered LDS (k already in the order the nibble trick
// emits); a wave owns 32 rows x 64 columns of one chunk, software-pipelines its k-steps by hand
// (operands of step t + 1 are fetched and de-quantised while the matrix cores work on step t) and
// keeps its weights eight k-steps ahead in registers; the K halves meet in LDS at the end.
// ---------------------------------------------------------------------------------------------
constexpr int kWideCols = 128;
constexpr int kWideThreads = 512;
constexpr int kChunkK = 128;                    // k per staged activation chunk = 4 MFMA k-steps
constexpr int kXPitch = kChunkK * 2 + 16;       // bytes per LDS row: 16 rows x 16 B hit 64 distinct banks
constexpr int kXBuf = 64 * kXPitch;

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// How the 64 x 128 output tiles are dealt to workgroups.
//   mode 0: grid (tiles, splits): one (tile, K slice) per workgroup; slices meet in slabs laid out
//           [split][M][N].
//   mode 1: more tiles than compute units: grid (wgs); workgroup w takes the whole tiles
//           w, w + wgs, ... of the first full_rounds * wgs tiles, and the `left` tiles that remain
//           are cut into left_slices K slices each, one per workgroup, so that the last round costs
//           1 / left_slices of a tile instead of a whole one; their slabs are compact:
//           [left tile][slice][64][128].
struct WidePlan {
  int mode, tiles_n, splits, cps, wgs, full_rounds, left, left_slices, left_cps;
};

template <int BITS, int KSTEPS>
__global__ __launch_bounds__(kWideThreads) void gptq_gemm_wide_kernel(hymeta_gptq_gemm_args a, WidePlan plan,
                                                                      float* slabs, int* counters) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 2 * kXBuf];   // [buffer][K half][row][pitch]
  const int njobs = plan.mode == 0 ? 1 : plan.full_rounds + (static_cast<int>(blockIdx.x) < plan.left * plan.left_slices ? 1 : 0);
  for (int job = 0; job < njobs; ++job) {
  int tile, split, splits, cps;
  float* sbase;                      // this tile's slab of slice 0
  int64_t s_split_stride, s_row_stride;
  if (plan.mode == 0) {
    tile = blockIdx.x, split = blockIdx.y, splits = plan.splits, cps = plan.cps;
  } else if (job < plan.full_rounds) {
    tile = blockIdx.x + plan.wgs * job, split = 0, splits = 1, cps = a.K / kChunkK;
  } else {
    tile = plan.wgs * plan.full_rounds + static_cast<int>(blockIdx.x) % plan.left;
    split = blockIdx.x / plan.left, cps = plan.left_cps;
    splits = (a.K / kChunkK + cps - 1) / cps;
  }
  const int nb = tile % plan.tiles_n, mb = tile / plan.tiles_n;
  if (plan.mode == 0) {
    sbase = slabs + static_cast<int64_t>(mb) * 64 * a.N + nb * kWideCols;
    s_split_stride = static_cast<int64_t>(a.M) * a.N, s_row_stride = a.N;
  } else {
    sbase = slabs + static_cast<int64_t>(tile - plan.wgs * plan.full_rounds) * plan.left_slices * (64 * kWideCols);
    s_split_stride = 64 * kWideCols, s_row_stride = kWideCols;
  }
  const bool active = split < splits;          // a slice past the end of K (mode 1, uneven cut) has nothing to do
  if (job > 0) __syncthreads();                // the previous tile's last LDS reads
  if (active) {
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int c = lane >> 4, n = lane & 15;
  // 8 waves = 2 K halves x 2 row halves x 2 column halves: in every round of two chunks the waves
  // of K half kh work on chunk 2 round + kh
  const int kh = wave >> 2, wr = (wave >> 1) & 1, wc = wave & 1;
  const int m0 = mb * 64;
  const int ch_lo = split * cps;
  const int ch_hi = min(a.K / kChunkK, ch_lo + cps);
  const int nch = ch_hi - ch_lo;                  // >= 1
  const int lcol0 = wc * 64 + 4 * n;              // column inside the tile
  const int col0 = nb * kWideCols + lcol0;
  constexpr int PACK = 32 / BITS;
  constexpr int QL = BITS == 4 ? 1 : 2;
  constexpr int GPC = 4 / KSTEPS;                 // groups per chunk
  constexpr int kRing = 8;                        // weight k-steps in flight

  f32x4 acc_tot[2][4];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc_tot[mt][i] = {0.f, 0.f, 0.f, 0.f};
  const u32x4 ones_raw = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
  const bf16x8 ones = as_frag(ones_raw);

  // ---- activation staging, two chunks per round: piece p = rows 16p .. 16p+15; 32 lanes x 16 B =
  // the row's two 256-byte chunk pieces (lanes 0-15: the even chunk, 16-31: the odd one)
  const int xhalf = (tid >> 4) & 1;
  const uint16_t* xsrc[4];
  int xdst[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int row = 16 * p + (tid >> 5);
    xsrc[p] = reinterpret_cast<const uint16_t*>(a.x) + static_cast<int64_t>(min(m0 + row, a.M - 1)) * a.x_stride_m + 8 * (tid & 15);
    xdst[p] = xhalf * kXBuf + row * kXPitch + 16 * (tid & 15);
  }
  u32x4 xs[4];                                    // the next round's chunks, on their way to LDS
  auto load_x = [&](int round, u32x4 (&xr)[4]) {
    const int ch = ch_lo + min(2 * round + xhalf, nch - 1);
#pragma unroll
    for (int p = 0; p < 4; ++p) xr[p] = ld16g(xsrc[p] + ch * kChunkK);
  };
  auto write_x = [&](int buf, const u32x4 (&xr)[4]) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      u32x4 v = xr[p];
      if constexpr (BITS == 4) {
        // k-slot order (0,4,1,5,2,6,3,7), the order the nibble trick emits
        u32x4 t;
        t[0] = __builtin_amdgcn_perm(v[2], v[0], 0x05040100u);
        t[1] = __builtin_amdgcn_perm(v[2], v[0], 0x07060302u);
        t[2] = __builtin_amdgcn_perm(v[3], v[1], 0x05040100u);
        t[3] = __builtin_amdgcn_perm(v[3], v[1], 0x07060302u);
        v = t;
      }
      *reinterpret_cast<u32x4*>(smem + buf * 2 * kXBuf + xdst[p]) = v;
    }
  };
  // A fragments of this wave: rows 32 wr + 16 mt + n, 8 k at 8c of the k-step
  const int afrag_off = kh * kXBuf + (32 * wr + n) * kXPitch + 16 * c;

  // ---- weight ring and group parameters: every load unconditional, indices clamped
  const int nrounds = (nch + 1) / 2;
  const int last_step = 4 * nrounds - 1;
  u32x4 qring[kRing][QL];
  u32x2 sraw[2][GPC];
  uint32_t zraw[2][GPC];
  auto load_w = [&](int t, u32x4 (&q)[QL]) {
    const int tt = min(t, last_step);
    const int k = (ch_lo + min(2 * (tt >> 2) + kh, nch - 1)) * kChunkK + 32 * (tt & 3) + 8 * c;
    const int32_t* qp = a.qweight + static_cast<int64_t>(k / PACK) * a.N + col0;
#pragma unroll
    for (int i = 0; i < QL; ++i) q[i] = ld16g(qp + static_cast<int64_t>(i) * a.N);
  };
  auto load_params = [&](int ci, u32x2 (&sr)[GPC], uint32_t (&zr)[GPC]) {
    const int g0 = (ch_lo + min(2 * min(ci, nrounds - 1) + kh, nch - 1)) * GPC;
#pragma unroll
    for (int gq = 0; gq < GPC; ++gq) {
      const int64_t g = g0 + gq;
      sr[gq] = *(const __attribute__((address_space(1))) u32x2*)(reinterpret_cast<const uint16_t*>(a.scales) + g * a.N + col0);
      zr[gq] = *(const __attribute__((address_space(1))) uint32_t*)(a.qzeros + g * (a.N / PACK) + col0 / PACK);
    }
  };

  // ---- the k-step pipeline.  Operands of step t live in register set t & 1: while the matrix
  // cores work on set t & 1, the same wave reads the next A fragments from LDS, turns the next
  // packed weights into set (t + 1) & 1 and refills their ring slot.  One barrier per chunk,
  // ahead of the MFMAs of the chunk's last step.
  u32x4 bq[2][4];
  bf16x8 af[2][2];
  f32x4 acc_g[2][4], acc_x[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    acc_x[mt] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) acc_g[mt][i] = {0.f, 0.f, 0.f, 0.f};
  }
  // (q >> 4w) & 0x000F000F | 0x43004300 as shift + v_and_or_b32: the mask rides in an SGPR and the
  // exponent pattern in a VGPR (two 32-bit constants do not fit one instruction, and left to
  // itself the compiler spends three instructions per word).
  uint32_t exp_pattern = 0x43004300u;
  asm volatile("" : "+v"(exp_pattern));
  auto dequant = [&](const u32x4 (&q)[QL], u32x4 (&b)[4]) {
    if constexpr (BITS == 4) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const uint32_t sh = w == 0 ? q[0][i] : q[0][i] >> (4 * w);
          asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(b[i][w]) : "v"(sh), "s"(0x000F000Fu), "v"(exp_pattern));
        }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t lohi[2] = {q[0][i], q[QL - 1][i]};
#pragma unroll
        for (int hw = 0; hw < 2; ++hw) {
          const float f0 = static_cast<float>(lohi[hw] & 0xffu);
          const float f1 = static_cast<float>((lohi[hw] >> 8) & 0xffu);
          const float f2 = static_cast<float>((lohi[hw] >> 16) & 0xffu);
          const float f3 = static_cast<float>(lohi[hw] >> 24);
          // integers <= 255 are exact in bf16: keep the high halves
          b[i][2 * hw] = __builtin_amdgcn_perm(__float_as_uint(f1), __float_as_uint(f0), 0x07060302u);
          b[i][2 * hw + 1] = __builtin_amdgcn_perm(__float_as_uint(f3), __float_as_uint(f2), 0x07060302u);
        }
      }
    }
  };
  auto read_af = [&](int buf, int step, bf16x8 (&f)[2]) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
      f[mt] = as_frag(*reinterpret_cast<const u32x4*>(smem + buf * 2 * kXBuf + afrag_off + 16 * mt * kXPitch + 64 * step));
  };
  // Half-dequantisation: B fragments i0, i0 + 1 of the four.
  auto dequant2 = [&](const u32x4 (&q)[QL], u32x4 (&b)[4], int i0) {
    u32x4 full[4];
    dequant(q, full);
    b[i0] = full[i0];
    b[i0 + 1] = full[i0 + 1];
  };
  // Scalar FMAs on purpose: beside MFMAs a packed fp32 FMA costs several times its issue slot.
  float pend_sc[4] = {0.f, 0.f, 0.f, 0.f}, pend_nzs[4] = {0.f, 0.f, 0.f, 0.f};   // s and -s * (c + z)
  auto epilogue = [&](int mt) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float t = acc_tot[mt][i][r];
        asm("v_fmac_f32 %0, %1, %2" : "+v"(t) : "v"(pend_sc[i]), "v"(acc_g[mt][i][r]));
        asm("v_fmac_f32 %0, %1, %2" : "+v"(t) : "v"(pend_nzs[i]), "v"(acc_x[mt][r]));
        acc_tot[mt][i][r] = t;
      }
  };

  // A step is two scheduling blocks the compiler cannot move code across (left alone it sinks
  // the ring refills to the end of the chunk and the ring runs dry):
  //   A: LDS reads of the next A fragments, ring refill of the slot this step's operands came
  //      from, the 5 MFMAs of row tile 0, first half of the next step's weights, and the scales of
  //      row tile 1 of a group that ended with the previous step;
  //   B: the 5 MFMAs of row tile 1, second half of the weights, and at a group's end its
  //      scales and zeros and their application to row tile 0.
  // One barrier per chunk, ahead of the last step's block A.
  auto do_chunk = [&](int ci, auto parity) {
    constexpr int PAR = decltype(parity)::value;
    const bool real = 2 * ci + kh < nch;
    static_for<0, 4>([&](auto sc_) {
      constexpr int S = decltype(sc_)::value;
      constexpr int SP = S & 1;                       // operand set of this step
      constexpr int GQ = S / KSTEPS;                  // group within the chunk
      constexpr bool kFirst = S % KSTEPS == 0, kLast = S % KSTEPS == KSTEPS - 1;
      constexpr int slot = (4 * PAR + S) % kRing, nslot = (4 * PAR + S + 1) % kRing;
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
      if constexpr (S == 3) {
        write_x(PAR ^ 1, xs);
        load_x(ci + 2, xs);
        __syncthreads();
        read_af(PAR ^ 1, 0, af[SP ^ 1]);
      } else {
        read_af(PAR, S + 1, af[SP ^ 1]);
      }
      if constexpr (S == 0) load_params(ci + 1, sraw[PAR ^ 1], zraw[PAR ^ 1]);
      load_w(4 * ci + S + kRing, qring[slot]);
      if constexpr (kFirst) epilogue(1);              // the previous group's row tile 1, before its sums are overwritten
#pragma unroll
      for (int i = 0; i < 4; ++i) acc_g[0][i] = mfma16(af[SP][0], as_frag(bq[SP][i]), kFirst ? zero : acc_g[0][i]);
      acc_x[0] = mfma16(af[SP][0], ones, kFirst ? zero : acc_x[0]);
      dequant2(qring[nslot], bq[SP ^ 1], 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc_g[1][i] = mfma16(af[SP][1], as_frag(bq[SP][i]), kFirst ? zero : acc_g[1][i]);
      acc_x[1] = mfma16(af[SP][1], ones, kFirst ? zero : acc_x[1]);
      dequant2(qring[nslot], bq[SP ^ 1], 2);
      if constexpr (kLast) {
        // s * (sum x (c + q) - (c + z) * sum x); zero scale for a padding chunk
        const uint32_t zw = zraw[PAR][GQ];
        const int first = col0 % PACK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const uint16_t raw = static_cast<uint16_t>(sraw[PAR][GQ][i >> 1] >> (16 * (i & 1)));
          const float sv = a.scales_dtype == HYMETA_F16 ? static_cast<float>(__builtin_bit_cast(_Float16, raw)) : bf2f(raw);
          const float scl = real ? sv : 0.f;
          const uint32_t z = (zw >> (BITS * (first + i))) & ((1u << BITS) - 1u);
          const float zf = static_cast<float>(z + a.zero_offset) + (BITS == 4 ? 128.f : 0.f);
          pend_sc[i] = scl;
          pend_nzs[i] = -scl * zf;
        }
        epilogue(0);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  load_x(0, xs);
#pragma unroll
  for (int p = 0; p < kRing; ++p) load_w(p, qring[p]);
  load_params(0, sraw[0], zraw[0]);
  write_x(0, xs);
  load_x(1, xs);
  __syncthreads();
  read_af(0, 0, af[0]);
  dequant(qring[0], bq[0]);
  for (int ci = 0; ci < nrounds; ci += 2) {
    do_chunk(ci, std::integral_constant<int, 0>{});
    if (ci + 1 < nrounds) do_chunk(ci + 1, std::integral_constant<int, 1>{});
  }
  epilogue(1);   // row tile 1 of the last group

  // ---- the two K halves meet in LDS: [8 accumulator quads][4 waves][64 lanes] x 16 bytes
  __syncthreads();
  if (kh == 1) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        *reinterpret_cast<f32x4*>(smem + (((mt * 4 + i) * 4 + (wave & 3)) * 64 + lane) * 16) = acc_tot[mt][i];
  }
  __syncthreads();
  const bool owner = kh == 0;
  bool writer = owner;
  if (owner) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        acc_tot[mt][i] += *reinterpret_cast<const f32x4*>(smem + (((mt * 4 + i) * 4 + (wave & 3)) * 64 + lane) * 16);
  }

  // ---- acc_tot[mt][i][r] = C[m0 + 32 wr + 16 mt + 4c + r][col0 + i]: 16 lanes x 4 columns per row
  if (splits > 1) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int lrow = 32 * wr + 16 * mt + 4 * c + r;
        if (owner && m0 + lrow < a.M)
          *reinterpret_cast<f32x4*>(sbase + split * s_split_stride + lrow * s_row_stride + lcol0) =
              f32x4{acc_tot[mt][0][r], acc_tot[mt][1][r], acc_tot[mt][2][r], acc_tot[mt][3][r]};
      }
    writer = last_arriver(counters + tile, splits, reinterpret_cast<int*>(smem)) && owner;
    // the last arriver sums the slabs, two slices' loads in flight at a time
    if (writer) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc_tot[mt][i] = {0.f, 0.f, 0.f, 0.f};
    for (int sp = 0; sp < splits; sp += 2) {
      const int sp1 = min(sp + 1, splits - 1);
      const float w1 = sp + 1 < splits ? 1.f : 0.f;
      f32x4 va[2][4], vb[2][4];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int lrow = min(32 * wr + 16 * mt + 4 * c + r, a.M - 1 - m0);
          va[mt][r] = *(const __attribute__((address_space(1))) f32x4*)(sbase + sp * s_split_stride + lrow * s_row_stride + lcol0);
          vb[mt][r] = *(const __attribute__((address_space(1))) f32x4*)(sbase + sp1 * s_split_stride + lrow * s_row_stride + lcol0);
        }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc_tot[mt][i][r] += va[mt][r][i] + w1 * vb[mt][r][i];
    }
    }
  }
  if (writer) {
  f32x4 bias = {0.f, 0.f, 0.f, 0.f};
  if (a.bias != nullptr) {
    const u32x2 bw = *reinterpret_cast<const u32x2*>(reinterpret_cast<const uint16_t*>(a.bias) + col0);
    bias = f32x4{lo_bf(bw[0]), hi_bf(bw[0]), lo_bf(bw[1]), hi_bf(bw[1])};
  }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int grow = m0 + 32 * wr + 16 * mt + 4 * c + r;
      if (grow >= a.M) continue;
      const u32x2 o = {pack_bf16x2(acc_tot[mt][0][r] + bias[0], acc_tot[mt][1][r] + bias[1]),
                       pack_bf16x2(acc_tot[mt][2][r] + bias[2], acc_tot[mt][3][r] + bias[3])};
      *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(a.out) + static_cast<int64_t>(grow) * a.out_stride_m + col0) = o;
    }
  }
  }   // active
  }   // jobs
}

// Whole-matrix dequantisation W[k][n] = bf16(s * (q - z)) for the prefill regime, where a library
// bf16 GEMM on the expanded weights beats re-dequantising them per 64-row block.
// Thread = one packed word row x 4 columns: 16-byte read, PACK rows of 8-byte writes.
template <int BITS>
__global__ __launch_bounds__(256) void gptq_dequant_kernel(const int32_t* __restrict__ qweight, const int32_t* __restrict__ qzeros,
                                                           const void* __restrict__ scales, uint16_t* __restrict__ out, int K, int N,
                                                           int group_size, int scales_dtype, int zero_offset) {
  constexpr int PACK = 32 / BITS;
  const int ncg = N / 4;
  const int64_t idx = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (idx >= static_cast<int64_t>(K / PACK) * ncg) return;
  const int kr = static_cast<int>(idx / ncg);
  const int col = static_cast<int>(idx % ncg) * 4;
  const int g = kr * PACK / group_size;
  const u32x4 qw = ld16(qweight + static_cast<int64_t>(kr) * N + col);
  const uint32_t zw = static_cast<uint32_t>(qzeros[static_cast<int64_t>(g) * (N / PACK) + col / PACK]);
  float sc[4], zf[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    sc[i] = load_scale(scales, static_cast<int64_t>(g) * N + col + i, scales_dtype);
    zf[i] = static_cast<float>(((zw >> (BITS * (col % PACK + i))) & ((1u << BITS) - 1u)) + zero_offset);
  }
#pragma unroll
  for (int j = 0; j < PACK; ++j) {
    float w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = sc[i] * (static_cast<float>((qw[i] >> (BITS * j)) & ((1u << BITS) - 1u)) - zf[i]);
    const u32x2 o = {pack_bf16x2(w[0], w[1]), pack_bf16x2(w[2], w[3])};
    *reinterpret_cast<u32x2*>(out + (static_cast<int64_t>(kr) * PACK + j) * N + col) = o;
  }
}

// The same expansion written as W^T: out[n][k], the layout of a torch Linear weight, which the library's bf16 GEMM
// runs measurably faster at prefill-sized M than the [K, N] one (7B shapes, 32768 rows: 10.4 ms against 11.8 ms a
// layer).  Workgroup = 64 columns x 32 packed rows: lane = column (256-byte row reads of the packed words), each
// thread expands its words into an LDS tile [column][k], which then leaves as whole 32*PACK-element runs of one
// output row (512 or 256 bytes).  Needs N % 64 == 0 and K % (32 * PACK) == 0.
template <int BITS>
__global__ __launch_bounds__(256) void gptq_dequant_nk_kernel(const int32_t* __restrict__ qweight, const int32_t* __restrict__ qzeros,
                                                              const void* __restrict__ scales, uint16_t* __restrict__ out, int K, int N,
                                                              int group_size, int scales_dtype, int zero_offset) {
  constexpr int PACK = 32 / BITS;
  constexpr int TK = 32 * PACK;                 // k values of a tile
  constexpr int PITCH = TK * 2 + 16;            // bytes of an LDS row: 16-byte pad keeps the column-strided writes off one bank group
  __shared__ __attribute__((aligned(16))) unsigned char tile[64 * PITCH];
  const int n0 = blockIdx.x * 64, kp0 = blockIdx.y * 32;
  const int lane_n = threadIdx.x & 63, sub = threadIdx.x >> 6;
  const int n = n0 + lane_n;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int kp = kp0 + sub + 4 * i;
    const int g = kp * PACK / group_size;
    const uint32_t qw = static_cast<uint32_t>(qweight[static_cast<int64_t>(kp) * N + n]);
    const uint32_t zw = static_cast<uint32_t>(qzeros[static_cast<int64_t>(g) * (N / PACK) + n / PACK]);
    const float sc = load_scale(scales, static_cast<int64_t>(g) * N + n, scales_dtype);
    const float zf = static_cast<float>(((zw >> (BITS * (n % PACK))) & ((1u << BITS) - 1u)) + zero_offset);
    uint32_t w[PACK / 2];
#pragma unroll
    for (int j = 0; j < PACK / 2; ++j) {
      const float lo = sc * (static_cast<float>((qw >> (BITS * (2 * j))) & ((1u << BITS) - 1u)) - zf);
      const float hi = sc * (static_cast<float>((qw >> (BITS * (2 * j + 1))) & ((1u << BITS) - 1u)) - zf);
      w[j] = pack_bf16x2(lo, hi);
    }
    unsigned char* dst = tile + lane_n * PITCH + (sub + 4 * i) * (PACK * 2);
    if constexpr (PACK == 8) *reinterpret_cast<u32x4*>(dst) = u32x4{w[0], w[1], w[2], w[3]};
    else *reinterpret_cast<u32x2*>(dst) = u32x2{w[0], w[1]};
  }
  __syncthreads();
  constexpr int CHUNKS = TK * 2 / 16;           // 16-byte pieces of one row of the tile
  for (int c = threadIdx.x; c < 64 * CHUNKS; c += 256) {
    const int row = c / CHUNKS, piece = c % CHUNKS;
    const u32x4 v = *reinterpret_cast<const u32x4*>(tile + row * PITCH + piece * 16);
    *reinterpret_cast<u32x4*>(out + static_cast<int64_t>(n0 + row) * K + static_cast<int64_t>(kp0) * PACK + piece * 8) = v;
  }
}

// ---------------------------------------------------------------------------------------------
// W8A8
// ---------------------------------------------------------------------------------------------
// One workgroup per token row: s = max|x| / 127, xq = clamp(rint(x / s)) (w8a8_quant.h).
__global__ __launch_bounds__(256) void w8a8_quant_kernel(hymeta_w8a8_gemm_args a) {
  __shared__ float s_red[4];
  const int m = blockIdx.x;
  quantise_row_int8(reinterpret_cast<const uint16_t*>(a.x) + static_cast<int64_t>(m) * a.x_stride_m, a.K,
                    a.xq + static_cast<int64_t>(m) * a.K, a.x_scale + m, s_red);
}

// out = bf16(acc * x_scale[m] * w_scale[n] + bias[n]): thread = 4 columns of one row
__global__ __launch_bounds__(256) void w8a8_scale_out_kernel(const int32_t* __restrict__ acc, const float* __restrict__ x_scale,
                                                             const float* __restrict__ w_scale, const uint16_t* __restrict__ bias,
                                                             uint16_t* __restrict__ out, int M, int N, int64_t acc_stride_m,
                                                             int64_t out_stride_m) {
  const int ncg = N / 4;
  const int64_t idx = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (idx >= static_cast<int64_t>(M) * ncg) return;
  const int m = static_cast<int>(idx / ncg), col = static_cast<int>(idx % ncg) * 4;
  const i32x4 v = *reinterpret_cast<const i32x4*>(acc + m * acc_stride_m + col);
  const f32x4 ws = *reinterpret_cast<const f32x4*>(w_scale + col);
  const float xs = x_scale[m];
  f32x4 b = {0.f, 0.f, 0.f, 0.f};
  if (bias != nullptr) {
    const u32x2 bw = *reinterpret_cast<const u32x2*>(bias + col);
    b = f32x4{lo_bf(bw[0]), hi_bf(bw[0]), lo_bf(bw[1]), hi_bf(bw[1])};
  }
  float y[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) y[i] = w8a8_scaled(v[i], xs, ws[i], b[i]);
  *reinterpret_cast<u32x2*>(out + m * out_stride_m + col) = u32x2{pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3])};
}

// grid = (N / 64, ceil(M / (16 MT)), splits); the 4 waves take alternate 64-deep k-steps of the
// workgroup's K slice; slices meet in int32 slabs (exact), the last arriver scales and writes.
template <int MT>
__global__ __launch_bounds__(256) void w8a8_gemm_kernel(hymeta_w8a8_gemm_args a, int splits, int steps_per_split,
                                                        int* slabs, int* counters) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  int* s_red = reinterpret_cast<int*>(smem_raw);  // [4 waves][16 MT rows][64 cols]
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int c = lane >> 4, n = lane & 15;
  const int nb = blockIdx.x;
  const int m0 = blockIdx.y * 16 * MT;

  i32x4 acc[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[mt][i] = {0, 0, 0, 0};

  const int split = blockIdx.z;
  const int ks_lo = split * steps_per_split;
  const int ks_hi = min(a.K / 64, ks_lo + steps_per_split);
  // At 33..64 rows a k-step is 16 MFMAs against two microseconds of load latency, and a launch holds two
  // or three waves per SIMD: each wave keeps a ring of four k-steps' fragments in flight.  Loads are
  // unconditional (a step past the slice re-reads its last step, a row past M re-reads row M - 1 and is
  // zeroed) so that the compiler can count them.
  struct Frags {
    i32x4 b[4];
    i32x4 a[MT];
  };
  const int8_t* wrow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)   // n-tile i: output channels 64 nb + 16 i + n
    wrow[i] = a.qweight + static_cast<int64_t>(nb * kColsPerBlock + 16 * i + n) * a.K + 16 * c;
  const int8_t* xrow[MT];
  bool xlive[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int row = m0 + 16 * mt + n;
    xlive[mt] = row < a.M;
    xrow[mt] = a.xq + static_cast<int64_t>(min(row, a.M - 1)) * a.K + 16 * c;
  }
  auto load = [&](int ks, Frags& f) {
    const int k0 = 64 * min(ks, ks_hi - 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) f.b[i] = __builtin_bit_cast(i32x4, ld16g(wrow[i] + k0));
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) f.a[mt] = __builtin_bit_cast(i32x4, ld16g(xrow[mt] + k0));
  };
  auto mma = [&](const Frags& f) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const i32x4 afrag = xlive[mt] ? f.a[mt] : i32x4{0, 0, 0, 0};
#pragma unroll
      for (int i = 0; i < 4; ++i)
        acc[mt][i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(afrag, f.b[i], acc[mt][i], 0, 0, 0);
    }
  };
  if constexpr (MT < 4) {
    // up to 32 rows the plain loop already streams the weights (4.5-4.8 TB/s on the widest matrix) and
    // the ring measured slower (fewer waves resident)
    for (int ks = ks_lo + wave; ks < ks_hi; ks += 4) {
      Frags f;
      load(ks, f);
      mma(f);
    }
  } else if (ks_lo + wave < ks_hi) {
    Frags f[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) load(ks_lo + wave + 4 * u, f[u]);
    for (int ks = ks_lo + wave; ks < ks_hi; ks += 16) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (ks + 4 * u < ks_hi) mma(f[u]);
        load(ks + 16 + 4 * u, f[u]);
      }
    }
  }
  // acc[mt][i][r] = C[row 16 mt + 4c + r][col 16 i + n]
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        s_red[(static_cast<size_t>(wave) * 16 * MT + 16 * mt + 4 * c + r) * kColsPerBlock + 16 * i + n] = acc[mt][i][r];
  __syncthreads();
  const int cg = tid & 15;
  const int col = nb * kColsPerBlock + 4 * cg;
  i32x4 sum[MT];
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    const int row = (tid >> 4) + 16 * j;
    sum[j] = {0, 0, 0, 0};
#pragma unroll
    for (int w = 0; w < 4; ++w)
      sum[j] += *reinterpret_cast<const i32x4*>(s_red + (static_cast<size_t>(w) * 16 * MT + row) * kColsPerBlock + 4 * cg);
  }
  if (splits > 1) {
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      const int grow = m0 + (tid >> 4) + 16 * j;
      if (grow < a.M) *reinterpret_cast<i32x4*>(slabs + (static_cast<int64_t>(split) * a.M + grow) * a.N + col) = sum[j];
    }
    if (!last_arriver(counters + blockIdx.y * gridDim.x + blockIdx.x, splits, s_red)) return;
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      const int grow = m0 + (tid >> 4) + 16 * j;
      sum[j] = {0, 0, 0, 0};
      if (grow < a.M)
        for (int sp = 0; sp < splits; ++sp)
          sum[j] += *reinterpret_cast<const i32x4*>(slabs + (static_cast<int64_t>(sp) * a.M + grow) * a.N + col);
    }
  }
  const f32x4 ws = *reinterpret_cast<const f32x4*>(a.w_scale + col);
  f32x4 bias = {0.f, 0.f, 0.f, 0.f};
  if (a.bias != nullptr) {
    const u32x2 bw = *reinterpret_cast<const u32x2*>(reinterpret_cast<const uint16_t*>(a.bias) + col);
    bias = f32x4{lo_bf(bw[0]), hi_bf(bw[0]), lo_bf(bw[1]), hi_bf(bw[1])};
  }
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    const int grow = m0 + (tid >> 4) + 16 * j;
    if (grow >= a.M) continue;
    const float xs = a.x_scale[grow];
    float y[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i] = w8a8_scaled(sum[j][i], xs, ws[i], bias[i]);
    const u32x2 o = {pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3])};
    *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(a.out) + static_cast<int64_t>(grow) * a.out_stride_m + col) = o;
  }
}


// ---------------------------------------------------------------------------------------------
// W8A8 at prefill-sized row counts: out = bf16( (xq wq^T) * x_scale[m] * w_scale[n] + bias ) as ONE tiled launch.
// The library path it replaces runs an int8 GEMM that writes its int32 sums to memory and a second launch that reads
// them back to scale them: 10 bytes of traffic per output element that this kernel never makes.
// BM rows x BN columns per workgroup, WM x WN waves that own (BM / WM) x (BN / WN) each, v_mfma_i32_16x16x64_i8.  Both
// operands are K-contiguous int8; a 128-deep step of each (128 bytes per row) is staged once per workgroup through
// double-buffered LDS images (16-byte chunks XOR-ed with the row, so the sixteen rows of a fragment read hit distinct
// banks): registers -> LDS right behind the barrier that frees the buffer, the next step's loads issued at once.  The
// epilogue applies the scales to the exact int32 sums with the rounding rule of every W8A8 kernel here (w8a8_scaled),
// stages the bf16 tile through LDS and stores whole rows.  grid = (N / BN, ceil(M / BM)), column tiles fastest: the
// workgroups that share a row tile of x run together.
// GATED (a fused gate / up projection, weight rows = gate, then up): the tile's first BN / 2 weight rows are gate rows
// and the other half the up rows of the same output columns, so the staged bf16 tile holds both factors of BN / 2 output
// columns side by side and the store pass writes silu(bf16 gate) * bf16 up -- the bits of this kernel followed by
// silu_and_mul, without the [M, N] intermediate (2.5 GB written and read back at 32768 rows of the 7B MLP).
// ---------------------------------------------------------------------------------------------
constexpr int kI8StepBytes = 128;                  // bytes of one staged row = 128 k

template <int BM, int BN, int WM, int WN, bool GATED>
__global__ __launch_bounds__(64 * WM * WN, 512 / (64 * WM * WN)) void w8a8_tiled_kernel(hymeta_w8a8_gemm_args a) {
  constexpr int kThreads = 64 * WM * WN;
  constexpr int MT = BM / WM / 16, NT = BN / WN / 16;       // 16 x 16 tiles per wave
  constexpr int kXBytes = BM * kI8StepBytes, kWBytes = BN * kI8StepBytes;
  constexpr int kRowsPerPass = kThreads / 8;
  constexpr int XP = BM / kRowsPerPass, WP = BN / kRowsPerPass;   // staging passes per operand
  __shared__ __attribute__((aligned(16))) char sm[2][kXBytes + kWBytes];   // [buffer][x | w]

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int c = lane >> 4, n = lane & 15;
  const int wm = wave / WN, wn = wave % WN;
  const int col0 = blockIdx.x * (GATED ? BN / 2 : BN);      // first output column of the tile
  const int row0 = blockIdx.y * BM;
  // weight row (= column of x W^T) behind column t of the tile
  auto wrow = [&](int t) { return GATED ? (t < BN / 2 ? col0 + t : a.N / 2 + col0 + (t - BN / 2)) : col0 + t; };

  // staging role: thread = 16-byte chunk sc of rows sr + kRowsPerPass i of both operand tiles (rows past M re-read the last row)
  const int sr = tid >> 3, sc = tid & 7;
  const int8_t* xp[XP];
  const int8_t* wp[WP];
#pragma unroll
  for (int i = 0; i < XP; ++i) xp[i] = a.xq + static_cast<int64_t>(min(row0 + sr + kRowsPerPass * i, a.M - 1)) * a.K + 16 * sc;
#pragma unroll
  for (int i = 0; i < WP; ++i)
    wp[i] = static_cast<const int8_t*>(a.qweight) + static_cast<int64_t>(wrow(sr + kRowsPerPass * i)) * a.K + 16 * sc;
  u32x4 xr[XP], wr[WP];
  auto issue = [&](int it) {
#pragma unroll
    for (int i = 0; i < XP; ++i) xr[i] = ld16g(xp[i] + it * kI8StepBytes);
#pragma unroll
    for (int i = 0; i < WP; ++i) wr[i] = ld16g(wp[i] + it * kI8StepBytes);
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      const int r = sr + kRowsPerPass * i;
      st16(&sm[buf][r * kI8StepBytes + 16 * (sc ^ (r & 7))], xr[i]);
    }
#pragma unroll
    for (int i = 0; i < WP; ++i) {
      const int r = sr + kRowsPerPass * i;
      st16(&sm[buf][kXBytes + r * kI8StepBytes + 16 * (sc ^ (r & 7))], wr[i]);
    }
  };

  i32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = i32x4{0, 0, 0, 0};

  const int nk = a.K / kI8StepBytes;
  issue(0);
  commit(0);
  if (nk > 1) issue(1);
  __syncthreads();
  for (int it = 0; it < nk; ++it) {
    const int buf = it & 1;
    if (it + 1 < nk) commit(buf ^ 1);
    if (it + 2 < nk) issue(it + 2);
    const char* ximg = sm[buf] + (16 * MT * wm + n) * kI8StepBytes;
    const char* wimg = sm[buf] + kXBytes + (16 * NT * wn + n) * kI8StepBytes;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {          // two 64-deep k-steps per staged step; lane (n, c) holds k = 16 c .. 16 c + 15 of its row
      const int off = 16 * ((4 * ks + c) ^ (n & 7));   // rows 16 t + n: (row & 7) = (n & 7)
      i32x4 xf[MT], wf[NT];
#pragma unroll
      for (int t = 0; t < MT; ++t) xf[t] = __builtin_bit_cast(i32x4, ld16(ximg + 16 * t * kI8StepBytes + off));
#pragma unroll
      for (int t = 0; t < NT; ++t) wf[t] = __builtin_bit_cast(i32x4, ld16(wimg + 16 * t * kI8StepBytes + off));
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(xf[mt], wf[nt], acc[mt][nt], 0, 0, 0);
    }
    __syncthreads();
  }

  // acc[mt][nt][r] = C[row 16 MT wm + 16 mt + 4c + r][col 16 NT wn + 16 nt + n]: scales and bias on the exact int32 sums,
  // one rounding, staged through LDS (the operand images are free) 128 rows at a time and stored as whole rows
  constexpr int kCPitch = 2 * BN + 16;
  constexpr int kStageRows = BM < 128 ? BM : 128;
  static_assert(kStageRows * kCPitch <= 2 * (kXBytes + kWBytes), "output staging fits the operand images");
  static_assert(kStageRows % (16 * MT) == 0 && BM % kStageRows == 0, "a wave's rows fall into one staging pass");
  char* cst = &sm[0][0];
  float wsc[NT], bia[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int col = wrow(16 * NT * wn + 16 * nt + n);
    wsc[nt] = a.w_scale[col];
    bia[nt] = a.bias != nullptr ? bf2f(static_cast<const uint16_t*>(a.bias)[col]) : 0.f;
  }
#pragma unroll
  for (int half = 0; half < BM / kStageRows; ++half) {
    if ((16 * MT * wm) / kStageRows == half) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * MT * wm + 16 * mt + 4 * c + r;
          const float xs = a.x_scale[min(row0 + row, a.M - 1)];
          uint16_t* crow = reinterpret_cast<uint16_t*>(cst + (row - kStageRows * half) * kCPitch) + 16 * NT * wn + n;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) crow[16 * nt] = f2bf(w8a8_scaled(acc[mt][nt][r], xs, wsc[nt], bia[nt]));
        }
    }
    __syncthreads();
    constexpr int kChunksPerRow = (GATED ? BN / 2 : BN) / 8;      // 8-column pieces of a row of the output tile
    for (int id = tid; id < kStageRows * kChunksPerRow; id += kThreads) {
      const int row = id / kChunksPerRow, ch = id % kChunksPerRow;
      const int grow = row0 + kStageRows * half + row;
      if (grow >= a.M) continue;
      u32x4 o = ld16(cst + row * kCPitch + 16 * ch);
      if constexpr (GATED) {
        const u32x4 up = ld16(cst + row * kCPitch + BN + 16 * ch);    // the up factors sit BN / 2 columns on
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float g0 = lo_bf(o[e]), g1 = hi_bf(o[e]);
          o[e] = pack_bf16x2(g0 * sigmoidf(g0) * lo_bf(up[e]), g1 * sigmoidf(g1) * hi_bf(up[e]));
        }
      }
      st16(static_cast<uint16_t*>(a.out) + static_cast<int64_t>(grow) * a.out_stride_m + col0 + 8 * ch, o);
    }
    if (half + 1 < BM / kStageRows) __syncthreads();
  }
}

// Rows from which the tiled kernel runs, and whether the shape fits it (K in whole 128-deep steps, N in whole tiles).
constexpr int kW8A8TiledFromRows = 129;
static bool w8a8_tiled_fits(int M, int K, int N) { return M >= kW8A8TiledFromRows && K % kI8StepBytes == 0 && N % 128 == 0; }

// 256 x 256 tiles (8 waves of 64 x 128) once they fill the chip, 128 x 128 (4 waves of 64 x 64) below
template <bool GATED>
static void w8a8_tiled_launch(const hymeta_w8a8_gemm_args& a, hipStream_t s) {
  const int64_t big_tiles = static_cast<int64_t>(a.N / 256) * ((a.M + 255) / 256);
  if (a.N % 256 == 0 && big_tiles >= 192)
    hipLaunchKernelGGL((w8a8_tiled_kernel<256, 256, 4, 2, GATED>), dim3(a.N / 256, (a.M + 255) / 256), dim3(512), 0, s, a);
  else
    hipLaunchKernelGGL((w8a8_tiled_kernel<128, 128, 2, 2, GATED>), dim3(a.N / 128, (a.M + 127) / 128), dim3(256), 0, s, a);
}

}  // namespace hymeta

using namespace hymeta;

static int mt_for(int M) { return M <= 16 ? 1 : (M <= 32 ? 2 : 4); }

static int64_t gptq_tiles(int M, int N) {
  const int mt = mt_for(M);
  return static_cast<int64_t>(N / kColsPerBlock) * ((M + 16 * mt - 1) / (16 * mt));
}

constexpr int kMaxPairs = 256;    // (tile, K slice) pairs the library's own choice of decomposition ever creates

static bool gptq_use_wide(int M, int K, int N) { return M > 32 && K % kChunkK == 0 && N % kWideCols == 0; }

extern "C" int hymeta_gptq_gemm_split_k(int32_t M, int32_t K, int32_t N, int32_t group_size) {
  if (M <= 0 || K <= 0 || N <= 0 || group_size <= 0) return 1;
  if (gptq_use_wide(M, K, N)) {
    // one workgroup (8 waves) per compute unit; a slice keeps at least seven chunks (measured:
    // below that the slab hand-off costs more than the shorter slice saves)
    const int64_t tiles = static_cast<int64_t>(N / kWideCols) * ((M + 63) / 64);
    int64_t want = 256 / tiles;
    const int64_t cap = K / kChunkK / 7 > 1 ? K / kChunkK / 7 : 1;
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    return static_cast<int>(want);
  }
  int64_t tiles = gptq_tiles(M, N);
  if (tiles < 1) tiles = 1;
  const int G = K / group_size;
  // About one workgroup per compute unit (measured: a workgroup's latency barely depends on its
  // K slice; more slices only add slab traffic, for skinny M too).
  int64_t want = 256 / tiles;
  const int64_t cap = G / 4 > 1 ? G / 4 : 1;       // every wave of a workgroup gets at least one group
  if (want > cap) want = cap;
  if (want < 1) want = 1;
  return static_cast<int>(want);
}

static size_t gptq_counter_bytes(int, int) { return kTicketBytes; }

// workspace = [ticket counters: zero at first use, left zero by every call][fp32 slabs]
extern "C" size_t hymeta_gptq_gemm_workspace_bytes(int32_t M, int32_t N, int32_t split_k) {
  if (M <= 0 || N <= 0 || split_k == 1) return 0;
  // split_k <= 0 (the library chooses): never more (tile, slice) pairs than kMaxPairs, whatever K is
  if (split_k <= 0) {
    const size_t old = static_cast<size_t>(kMaxPairs) * 64 * kWideCols * sizeof(float);
    const size_t streaming = M <= 64 ? stream_gptq_workspace_bound(M, N) : (M <= 256 ? stream_gemm_workspace_bound(M, N) : 0);
    return gptq_counter_bytes(M, N) + (old > streaming ? old : streaming);
  }
  return gptq_counter_bytes(M, N) + static_cast<size_t>(split_k) * M * N * sizeof(float);
}

static int compute_units(void) {
  static int cached = 0;
  if (cached == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
      cached = n;
    else
      cached = 256;
  }
  return cached;
}

static int gptq_gemm_impl(const hymeta_gptq_gemm_args* args, hymeta_stream_t stream, hymeta_gemm_partials* deferred) {
  if (!hymeta::args_ok(args)) return HYMETA_ERR_INVALID_ARG;
  if (deferred != nullptr) *deferred = hymeta::make_args<hymeta_gemm_partials>();
  hymeta_gptq_gemm_args a = *args;
  if (a.M < 0 || a.K <= 0 || a.N <= 0) return HYMETA_ERR_INVALID_ARG;
  if (a.M == 0) return HYMETA_OK;
  if (!a.x || !a.qweight || !a.qzeros || !a.scales || !a.out) return HYMETA_ERR_INVALID_ARG;
  if (a.bits != 4 && a.bits != 8) return HYMETA_ERR_UNSUPPORTED;
  if (a.group_size != 32 && a.group_size != 64 && a.group_size != 128) return HYMETA_ERR_UNSUPPORTED;
  if (a.K % a.group_size != 0 || a.N % kColsPerBlock != 0) return HYMETA_ERR_UNSUPPORTED;
  if (a.scales_dtype != HYMETA_F16 && a.scales_dtype != HYMETA_BF16) return HYMETA_ERR_UNSUPPORTED;
  if (a.zero_offset != 0 && a.zero_offset != 1) return HYMETA_ERR_INVALID_ARG;
  if ((a.x_stride_m | a.out_stride_m) % 8 != 0) return HYMETA_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(a.x) | reinterpret_cast<uintptr_t>(a.qweight) | reinterpret_cast<uintptr_t>(a.out) |
       reinterpret_cast<uintptr_t>(a.bias) | reinterpret_cast<uintptr_t>(a.workspace)) % 16 != 0)
    return HYMETA_ERR_UNSUPPORTED;
  if (reinterpret_cast<uintptr_t>(a.scales) % 8 != 0) return HYMETA_ERR_UNSUPPORTED;
  // decode-sized 4-bit, group 128: the weight-streaming kernel (whole-line loads through a wave-private LDS image, x
  // shared through LDS, one workgroup per compute unit) when the library chooses the decomposition
  static const int stream_from = [] { const char* e = getenv("HYMETA_GPTQ_STREAM_FROM"); return e ? atoi(e) : 1; }();
  if (a.split_k <= 0 && a.M >= stream_from) {
    StreamGptqPlan sp;
    const bool have_ws = a.workspace != nullptr;
    if (stream_gptq_plan(a.M, a.K, a.N, a.bits, a.group_size, have_ws, &sp) &&
        (!have_ws || a.workspace_bytes >= kTicketBytes + stream_gptq_workspace_bytes(a.M, a.N, sp))) {
      StreamGptqArgs sa = {};
      sa.x = static_cast<const char*>(a.x), sa.qweight = a.qweight, sa.qzeros = a.qzeros, sa.scales = a.scales;
      sa.bias = static_cast<const uint16_t*>(a.bias), sa.out = static_cast<uint16_t*>(a.out);
      sa.slabs = have_ws ? reinterpret_cast<char*>(a.workspace) + kTicketBytes : nullptr;
      sa.M = a.M, sa.N = a.N, sa.scales_f16 = a.scales_dtype == HYMETA_F16, sa.zero_offset = a.zero_offset;
      sa.x_stride_bytes = a.x_stride_m * 2, sa.out_stride = a.out_stride_m;
      if (deferred != nullptr && sp.splits > 1) {
        deferred->slabs = sa.slabs, deferred->splits = sp.splits, deferred->bias = a.bias;
        deferred->M = a.M, deferred->N = a.N;
      }
      return stream_gptq_launch(sa, sp, begin_launches(stream), deferred != nullptr);
    }
  }
  // 129 to 256 rows (8-bit: 65 to 256): the 129..256-row streaming kernel with the packed words expanded into its weight
  // image (one pass over the weights for all rows; the kernels below re-read and re-expand them per 64-row block).
  // Measured on the 7B shapes, us per launch, this kernel / the kernels below: 8-bit group 64 at 128 rows gate_up
  // 76 / 103, down 50 / 60, qkv 24 / 25, o 24 / 25; at 64 rows 65 / 63, 42 / 40, 20 / 23, 19 / 21 (level); 4-bit
  // group 128 at 96 / 128 sequences of the 7B step 7.24 / 7.73 ms against 7.18 / 7.59 (those rows stay below).
  // HYMETA_GPTQ_TALL=0 keeps every row count on the kernels below; HYMETA_GPTQ8_TALL_FROM moves the 8-bit threshold.
  static const int tall_on = [] { const char* e = getenv("HYMETA_GPTQ_TALL"); return e ? atoi(e) : 1; }();
  static const int tall8_from = [] { const char* e = getenv("HYMETA_GPTQ8_TALL_FROM"); return e ? atoi(e) : 65; }();
  const int tall_from = a.bits == 8 ? tall8_from : 129;
  if (tall_on && a.split_k <= 0 && a.M >= tall_from && (a.group_size == 64 || a.group_size == 128) && a.K % 64 == 0) {
    StreamPlan tp;
    const bool have_ws = a.workspace != nullptr;
    if (tall_gemm_plan(a.M, a.K * 2, a.N, have_ws, &tp, tall_from) &&
        (!have_ws || a.workspace_bytes >= kTicketBytes + stream_gemm_workspace_bytes(a.M, a.N, tp))) {
      StreamArgs sa = {};
      sa.x = static_cast<const char*>(a.x), sa.w = reinterpret_cast<const char*>(a.qweight);
      sa.qzeros = a.qzeros, sa.scales = a.scales, sa.bits = a.bits;
      sa.scales_f16 = a.scales_dtype == HYMETA_F16, sa.zero_offset = a.zero_offset, sa.group_size = a.group_size;
      sa.bias = static_cast<const uint16_t*>(a.bias), sa.out = static_cast<uint16_t*>(a.out);
      sa.slabs = have_ws ? reinterpret_cast<char*>(a.workspace) + kTicketBytes : nullptr;
      sa.M = a.M, sa.N = a.N, sa.k_bytes = a.K * 2;
      sa.x_stride_bytes = a.x_stride_m * 2, sa.out_stride = a.out_stride_m;
      if (deferred != nullptr && tp.splits > 1) {
        deferred->slabs = sa.slabs, deferred->splits = tp.splits, deferred->bias = a.bias;
        deferred->M = a.M, deferred->N = a.N;
      }
      return tall_gemm_gptq_launch(sa, tp, begin_launches(stream), deferred != nullptr);
    }
  }
  const bool wide = gptq_use_wide(a.M, a.K, a.N);
  const int G = a.K / a.group_size;
  const int units = wide ? a.K / kChunkK : G;       // what a K slice is counted in
  const bool autosplit = a.split_k <= 0;
  int splits = autosplit ? hymeta_gptq_gemm_split_k(a.M, a.K, a.N, a.group_size) : a.split_k;
  if (autosplit && a.workspace == nullptr) splits = 1;
  if (gptq_tiles(a.M, a.N) > kMaxTickets) splits = 1;
  if (splits > units) splits = units;
  const size_t counter_bytes = gptq_counter_bytes(a.M, a.N);
  if (a.workspace != nullptr && a.workspace_bytes < hymeta_gptq_gemm_workspace_bytes(a.M, a.N, autosplit ? 0 : splits))
    return HYMETA_ERR_WORKSPACE;
  if (splits > 1 && a.workspace == nullptr) return HYMETA_ERR_WORKSPACE;
  const int gps = (units + splits - 1) / splits;
  splits = (units + gps - 1) / gps;
  // more 128-column tiles than compute units: whole tiles round by round, the remainder cut in K
  WidePlan plan = {0, a.N / kWideCols, splits, gps, 0, 0, 0, 0, 0};
  if (wide && autosplit && a.workspace != nullptr) {
    const int wgs = compute_units();
    const int tiles = plan.tiles_n * ((a.M + 63) / 64);
    if (tiles > wgs && wgs <= kMaxPairs) {
      plan.mode = 1, plan.wgs = wgs;
      plan.full_rounds = tiles / wgs, plan.left = tiles % wgs;
      if (plan.left > 0) {
        int slices = wgs / plan.left;
        const int cap = units / 4 > 1 ? units / 4 : 1;      // a slice keeps two rounds of two chunks
        if (slices > cap) slices = cap;
        plan.left_cps = (units + slices - 1) / slices;
        plan.left_slices = (units + plan.left_cps - 1) / plan.left_cps;
      }
    }
  }
  hipStream_t s = begin_launches(stream);
  int* counters = reinterpret_cast<int*>(a.workspace);
  float* slabs = a.workspace == nullptr ? nullptr : reinterpret_cast<float*>(reinterpret_cast<char*>(a.workspace) + counter_bytes);
  if (wide) {
    dim3 grid(plan.tiles_n * ((a.M + 63) / 64), splits), block(kWideThreads);
    if (plan.mode == 1) grid = dim3(plan.wgs);
#define HYMETA_LAUNCH_WIDE(B, KS) hipLaunchKernelGGL((gptq_gemm_wide_kernel<B, KS>), grid, block, 0, s, a, plan, slabs, counters)
#define HYMETA_LAUNCH_WIDE_KS(B)                     \
  do {                                               \
    if (a.group_size == 128) HYMETA_LAUNCH_WIDE(B, 4); \
    else if (a.group_size == 64) HYMETA_LAUNCH_WIDE(B, 2); \
    else HYMETA_LAUNCH_WIDE(B, 1);                   \
  } while (0)
    if (a.bits == 4) HYMETA_LAUNCH_WIDE_KS(4); else HYMETA_LAUNCH_WIDE_KS(8);
#undef HYMETA_LAUNCH_WIDE_KS
#undef HYMETA_LAUNCH_WIDE
    return launch_status();
  }
  const int mt = mt_for(a.M);
  dim3 grid(a.N / kColsPerBlock, (a.M + 16 * mt - 1) / (16 * mt), splits), block(256);
  const size_t lds = static_cast<size_t>(4) * 16 * mt * kColsPerBlock * sizeof(float);
#define HYMETA_LAUNCH_GPTQ(B, T, KS) \
  hipLaunchKernelGGL((gptq_gemm_kernel<B, T, KS>), grid, block, lds, s, a, splits, gps, slabs, counters)
#define HYMETA_LAUNCH_GPTQ_KS(B, T)                   \
  do {                                                \
    if (a.group_size == 128) HYMETA_LAUNCH_GPTQ(B, T, 4); \
    else if (a.group_size == 64) HYMETA_LAUNCH_GPTQ(B, T, 2); \
    else HYMETA_LAUNCH_GPTQ(B, T, 1);                 \
  } while (0)
  if (a.bits == 4) {
    if (mt == 1) HYMETA_LAUNCH_GPTQ_KS(4, 1); else if (mt == 2) HYMETA_LAUNCH_GPTQ_KS(4, 2); else HYMETA_LAUNCH_GPTQ_KS(4, 4);
  } else {
    if (mt == 1) HYMETA_LAUNCH_GPTQ_KS(8, 1); else if (mt == 2) HYMETA_LAUNCH_GPTQ_KS(8, 2); else HYMETA_LAUNCH_GPTQ_KS(8, 4);
  }
#undef HYMETA_LAUNCH_GPTQ_KS
#undef HYMETA_LAUNCH_GPTQ
  return launch_status();
}

extern "C" int hymeta_gptq_gemm(const hymeta_gptq_gemm_args* args, hymeta_stream_t stream) {
  return gptq_gemm_impl(args, stream, nullptr);
}

extern "C" int hymeta_gptq_gemm_deferred(const hymeta_gptq_gemm_args* args, hymeta_gemm_partials* partials,
                                         hymeta_stream_t stream) {
  if (!hymeta::args_ok(partials)) return HYMETA_ERR_INVALID_ARG;
  return gptq_gemm_impl(args, stream, partials);
}

extern "C" int hymeta_gptq_dequant(const int32_t* qweight, const int32_t* qzeros, const void* scales, void* out,
                                   int32_t K, int32_t N, int32_t bits, int32_t group_size, int32_t scales_dtype,
                                   int32_t zero_offset, hymeta_stream_t stream) {
  if (!qweight || !qzeros || !scales || !out || K <= 0 || N <= 0) return HYMETA_ERR_INVALID_ARG;
  if (bits != 4 && bits != 8) return HYMETA_ERR_UNSUPPORTED;
  if (group_size <= 0 || group_size % (32 / bits) != 0 || K % group_size != 0 || N % 8 != 0) return HYMETA_ERR_UNSUPPORTED;
  if (scales_dtype != HYMETA_F16 && scales_dtype != HYMETA_BF16) return HYMETA_ERR_UNSUPPORTED;
  if (zero_offset != 0 && zero_offset != 1) return HYMETA_ERR_INVALID_ARG;
  if ((reinterpret_cast<uintptr_t>(qweight) | reinterpret_cast<uintptr_t>(out)) % 16 != 0) return HYMETA_ERR_UNSUPPORTED;
  const int64_t threads = static_cast<int64_t>(K / (32 / bits)) * (N / 4);
  dim3 grid(static_cast<unsigned>((threads + 255) / 256)), block(256);
  hipStream_t s = begin_launches(stream);
  uint16_t* o = static_cast<uint16_t*>(out);
  if (bits == 4) hipLaunchKernelGGL(gptq_dequant_kernel<4>, grid, block, 0, s, qweight, qzeros, scales, o, K, N, group_size, scales_dtype, zero_offset);
  else hipLaunchKernelGGL(gptq_dequant_kernel<8>, grid, block, 0, s, qweight, qzeros, scales, o, K, N, group_size, scales_dtype, zero_offset);
  return launch_status();
}

extern "C" int hymeta_gptq_dequant_nk(const int32_t* qweight, const int32_t* qzeros, const void* scales, void* out,
                                      int32_t K, int32_t N, int32_t bits, int32_t group_size, int32_t scales_dtype,
                                      int32_t zero_offset, hymeta_stream_t stream) {
  if (!qweight || !qzeros || !scales || !out || K <= 0 || N <= 0) return HYMETA_ERR_INVALID_ARG;
  if (bits != 4 && bits != 8) return HYMETA_ERR_UNSUPPORTED;
  const int pack = 32 / bits;
  if (group_size <= 0 || group_size % pack != 0 || K % group_size != 0) return HYMETA_ERR_UNSUPPORTED;
  if (N % 64 != 0 || K % (32 * pack) != 0) return HYMETA_ERR_UNSUPPORTED;
  if (scales_dtype != HYMETA_F16 && scales_dtype != HYMETA_BF16) return HYMETA_ERR_UNSUPPORTED;
  if (zero_offset != 0 && zero_offset != 1) return HYMETA_ERR_INVALID_ARG;
  if (reinterpret_cast<uintptr_t>(out) % 16 != 0) return HYMETA_ERR_UNSUPPORTED;
  dim3 grid(static_cast<unsigned>(N / 64), static_cast<unsigned>(K / (32 * pack))), block(256);
  if (grid.y > 65535u) return HYMETA_ERR_UNSUPPORTED;
  hipStream_t s = begin_launches(stream);
  uint16_t* o = static_cast<uint16_t*>(out);
  if (bits == 4) hipLaunchKernelGGL(gptq_dequant_nk_kernel<4>, grid, block, 0, s, qweight, qzeros, scales, o, K, N, group_size, scales_dtype, zero_offset);
  else hipLaunchKernelGGL(gptq_dequant_nk_kernel<8>, grid, block, 0, s, qweight, qzeros, scales, o, K, N, group_size, scales_dtype, zero_offset);
  return launch_status();
}

extern "C" int hymeta_w8a8_gemm_split_k(int32_t M, int32_t K, int32_t N) {
  if (M <= 0 || K <= 0 || N <= 0) return 1;
  int64_t tiles = gptq_tiles(M, N);
  if (tiles < 1) tiles = 1;
  int64_t want = 256 / tiles;                       // about one workgroup per compute unit
  const int64_t cap = K / 64 / 8 > 1 ? K / 64 / 8 : 1;   // at least two k-steps per wave
  if (want > cap) want = cap;
  if (want < 1) want = 1;
  return static_cast<int>(want);
}

extern "C" size_t hymeta_w8a8_gemm_workspace_bytes(int32_t M, int32_t N, int32_t split_k) {
  if (M <= 0 || N <= 0) return 0;
  if (split_k <= 0) {   // a bound for whatever the library chooses (the streaming kernel's slabs, or the older kernel's)
    const size_t stream_bound = kTicketBytes + stream_gemm_workspace_bound(M, N);
    const size_t old_bound = kTicketBytes + static_cast<size_t>(16) * M * N * sizeof(int);
    return stream_bound > old_bound ? stream_bound : old_bound;
  }
  if (split_k <= 1) return 0;
  // [ticket counters: zero at first use, left zero by every call][int32 slabs [split_k][M][N]]
  return kTicketBytes + static_cast<size_t>(split_k) * M * N * sizeof(int);
}

static int w8a8_gemm_impl(const hymeta_w8a8_gemm_args* args, hymeta_stream_t stream, hymeta_gemm_partials* deferred) {
  if (!hymeta::args_ok(args)) return HYMETA_ERR_INVALID_ARG;
  if (deferred != nullptr) *deferred = hymeta::make_args<hymeta_gemm_partials>();
  const hymeta_w8a8_gemm_args& a = *args;
  if (a.M < 0 || a.K <= 0 || a.N <= 0) return HYMETA_ERR_INVALID_ARG;
  if (a.M == 0) return HYMETA_OK;
  if (!a.qweight || !a.w_scale || !a.out || !a.xq || !a.x_scale) return HYMETA_ERR_INVALID_ARG;   // x NULL: xq / x_scale are inputs
  if (a.K % 64 != 0 || a.N % kColsPerBlock != 0) return HYMETA_ERR_UNSUPPORTED;
  if ((a.x_stride_m | a.out_stride_m) % 8 != 0) return HYMETA_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(a.x) | reinterpret_cast<uintptr_t>(a.qweight) | reinterpret_cast<uintptr_t>(a.out) |
       reinterpret_cast<uintptr_t>(a.bias) | reinterpret_cast<uintptr_t>(a.xq) | reinterpret_cast<uintptr_t>(a.w_scale) |
       reinterpret_cast<uintptr_t>(a.workspace)) % 16 != 0)
    return HYMETA_ERR_UNSUPPORTED;
  // The streaming kernel (activations shared through LDS, whole-line weight loads) takes every shape it covers when the
  // library chooses the decomposition -- with one workgroup per compute unit it is ahead of the 64-column kernel below
  // at every row count (7B shapes at 1 / 16 rows, us: qkv 11.3 / 12.5 against 14.5 / 16.4, o 10.3 / 11.3 against
  // 13.6 / 15.5, gate_up 27.9 / 28.8 against 32.4 / 34.0, down 24.1 / 26.1 against 38.3 / 39.9).
  // HYMETA_W8A8_STREAM_FROM = first row count that goes there.
  static const int stream_from = [] { const char* e = getenv("HYMETA_W8A8_STREAM_FROM"); return e ? atoi(e) : 1; }();
  if (a.split_k <= 0 && a.M >= stream_from) {
    StreamPlan plan;
    const bool have_ws = a.workspace != nullptr;
    if (stream_gemm_plan(a.M, a.K, a.N, have_ws, &plan) &&
        (!have_ws || a.workspace_bytes >= kTicketBytes + stream_gemm_workspace_bytes(a.M, a.N, plan))) {
      hipStream_t s = begin_launches(stream);
      if (a.x != nullptr) {
        hipLaunchKernelGGL(w8a8_quant_kernel, dim3(a.M), dim3(256), 0, s, a);
        if (hipGetLastError() != hipSuccess) return HYMETA_ERR_LAUNCH;
      }
      StreamArgs sa = {};
      sa.x = reinterpret_cast<const char*>(a.xq), sa.w = reinterpret_cast<const char*>(a.qweight);
      sa.bias = static_cast<const uint16_t*>(a.bias), sa.out = static_cast<uint16_t*>(a.out);
      sa.x_scale = a.x_scale, sa.w_scale = a.w_scale;
      sa.slabs = have_ws ? static_cast<char*>(a.workspace) + kTicketBytes : nullptr;
      sa.M = a.M, sa.N = a.N, sa.k_bytes = a.K;
      sa.x_stride_bytes = a.K, sa.w_stride_bytes = a.K, sa.out_stride = a.out_stride_m;
      if (deferred != nullptr && plan.splits > 1) {
        deferred->slabs = sa.slabs, deferred->splits = plan.splits, deferred->is_int32 = 1, deferred->bias = a.bias;
        deferred->x_scale = a.x_scale, deferred->w_scale = a.w_scale;
        deferred->M = a.M, deferred->N = a.N;
      }
      return stream_gemm_launch<true>(sa, plan, s, deferred != nullptr);
    }
  }
  // prefill-sized row counts: the tiled kernel (scales in its epilogue), behind the per-token quantisation when x is given
  static const int tiled_from = [] { const char* e = getenv("HYMETA_W8A8_TILED_FROM"); return e ? atoi(e) : kW8A8TiledFromRows; }();
  if (a.split_k <= 0 && a.M >= tiled_from && w8a8_tiled_fits(a.M, a.K, a.N)) {
    hipStream_t s = begin_launches(stream);
    if (a.x != nullptr) {
      hipLaunchKernelGGL(w8a8_quant_kernel, dim3(a.M), dim3(256), 0, s, a);
      if (hipGetLastError() != hipSuccess) return HYMETA_ERR_LAUNCH;
    }
    w8a8_tiled_launch<false>(a, s);
    return launch_status();
  }
  const int ksteps = a.K / 64;
  int splits = a.split_k > 0 ? a.split_k : (a.workspace != nullptr ? hymeta_w8a8_gemm_split_k(a.M, a.K, a.N) : 1);
  if (splits > ksteps) splits = ksteps;
  if (gptq_tiles(a.M, a.N) > kMaxTickets) splits = 1;
  if (splits > 1 && (a.workspace == nullptr || a.workspace_bytes < hymeta_w8a8_gemm_workspace_bytes(a.M, a.N, splits)))
    return HYMETA_ERR_WORKSPACE;
  const int sps = (ksteps + splits - 1) / splits;
  splits = (ksteps + sps - 1) / sps;
  hipStream_t s = begin_launches(stream);
  int* counters = static_cast<int*>(a.workspace);
  int* slabs = reinterpret_cast<int*>(static_cast<char*>(a.workspace) + kTicketBytes);
  if (a.x != nullptr) {
    hipLaunchKernelGGL(w8a8_quant_kernel, dim3(a.M), dim3(256), 0, s, a);
    if (hipGetLastError() != hipSuccess) return HYMETA_ERR_LAUNCH;
  }
  const int mt = mt_for(a.M);
  dim3 grid(a.N / kColsPerBlock, (a.M + 16 * mt - 1) / (16 * mt), splits), block(256);
  const size_t lds = static_cast<size_t>(4) * 16 * mt * kColsPerBlock * sizeof(int);
  if (mt == 1) hipLaunchKernelGGL((w8a8_gemm_kernel<1>), grid, block, lds, s, a, splits, sps, slabs, counters);
  else if (mt == 2) hipLaunchKernelGGL((w8a8_gemm_kernel<2>), grid, block, lds, s, a, splits, sps, slabs, counters);
  else hipLaunchKernelGGL((w8a8_gemm_kernel<4>), grid, block, lds, s, a, splits, sps, slabs, counters);
  return launch_status();
}

extern "C" int hymeta_w8a8_quantize(const void* x, int8_t* xq, float* x_scale, int32_t M, int32_t K, int64_t x_stride_m,
                                    hymeta_stream_t stream) {
  if (M < 0 || K <= 0) return HYMETA_ERR_INVALID_ARG;
  if (M == 0) return HYMETA_OK;
  if (!x || !xq || !x_scale) return HYMETA_ERR_INVALID_ARG;
  if (K % 8 != 0 || x_stride_m % 8 != 0) return HYMETA_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(xq)) % 16 != 0) return HYMETA_ERR_UNSUPPORTED;
  hymeta_w8a8_gemm_args a = hymeta::make_args<hymeta_w8a8_gemm_args>();
  a.x = x, a.xq = xq, a.x_scale = x_scale, a.M = M, a.K = K, a.x_stride_m = x_stride_m;
  hipLaunchKernelGGL(w8a8_quant_kernel, dim3(M), dim3(256), 0, begin_launches(stream), a);
  return launch_status();
}

extern "C" int hymeta_w8a8_scale_out(const int32_t* acc, const float* x_scale, const float* w_scale, const void* bias, void* out,
                                     int32_t M, int32_t N, int64_t acc_stride_m, int64_t out_stride_m, hymeta_stream_t stream) {
  if (M < 0 || N <= 0) return HYMETA_ERR_INVALID_ARG;
  if (M == 0) return HYMETA_OK;
  if (!acc || !x_scale || !w_scale || !out) return HYMETA_ERR_INVALID_ARG;
  if (N % 4 != 0 || acc_stride_m % 4 != 0 || out_stride_m % 4 != 0) return HYMETA_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(acc) | reinterpret_cast<uintptr_t>(w_scale)) % 16 != 0 ||
      (reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(bias)) % 8 != 0)
    return HYMETA_ERR_UNSUPPORTED;
  const int64_t threads = static_cast<int64_t>(M) * (N / 4);
  hipLaunchKernelGGL(w8a8_scale_out_kernel, dim3(static_cast<unsigned>((threads + 255) / 256)), dim3(256), 0, begin_launches(stream),
                     acc, x_scale, w_scale, static_cast<const uint16_t*>(bias), static_cast<uint16_t*>(out), M, N, acc_stride_m,
                     out_stride_m);
  return launch_status();
}

extern "C" int hymeta_w8a8_gemm(const hymeta_w8a8_gemm_args* args, hymeta_stream_t stream) {
  return w8a8_gemm_impl(args, stream, nullptr);
}

extern "C" int hymeta_w8a8_gemm_deferred(const hymeta_w8a8_gemm_args* args, hymeta_gemm_partials* partials,
                                         hymeta_stream_t stream) {
  if (!hymeta::args_ok(partials)) return HYMETA_ERR_INVALID_ARG;
  return w8a8_gemm_impl(args, stream, partials);
}

// W8A8 gate/up projection with the activation in the epilogue (see hymeta_bf16_gemm_silu_mul): quantises x per row, then
// out [M, N / 2] = silu(gate) * up from the int8 sums.  Decode-sized batches run the streaming kernel, larger ones the
// tiled kernel (K % 128 == 0, N % 128 == 0); other shapes are HYMETA_ERR_UNSUPPORTED (the caller runs hymeta_w8a8_gemm +
// hymeta_silu_and_mul: same bits).
extern "C" int hymeta_w8a8_gemm_silu_mul(const hymeta_w8a8_gemm_args* args, hymeta_stream_t stream) {
  using namespace hymeta;
  if (!hymeta::args_ok(args)) return HYMETA_ERR_INVALID_ARG;
  const hymeta_w8a8_gemm_args& a = *args;
  if (a.M < 0 || a.K <= 0 || a.N <= 0 || a.bias != nullptr) return HYMETA_ERR_INVALID_ARG;
  if (a.M == 0) return HYMETA_OK;
  if (!a.qweight || !a.w_scale || !a.out || !a.xq || !a.x_scale) return HYMETA_ERR_INVALID_ARG;   // x NULL: xq / x_scale are inputs
  if (a.K % 64 != 0 || (a.x_stride_m | a.out_stride_m) % 8 != 0) return HYMETA_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(a.x) | reinterpret_cast<uintptr_t>(a.qweight) | reinterpret_cast<uintptr_t>(a.out) |
       reinterpret_cast<uintptr_t>(a.xq) | reinterpret_cast<uintptr_t>(a.w_scale)) % 16 != 0)
    return HYMETA_ERR_UNSUPPORTED;
  StreamPlan plan;
  const bool streaming = stream_gemm_plan_gated(a.M, a.K, a.N, &plan);
  if (!streaming && !w8a8_tiled_fits(a.M, a.K, a.N)) return HYMETA_ERR_UNSUPPORTED;
  hipStream_t s = begin_launches(stream);
  if (a.x != nullptr) {
    hipLaunchKernelGGL(w8a8_quant_kernel, dim3(a.M), dim3(256), 0, s, a);
    if (hipGetLastError() != hipSuccess) return HYMETA_ERR_LAUNCH;
  }
  if (!streaming) {
    w8a8_tiled_launch<true>(a, s);
    return launch_status();
  }
  StreamArgs sa = {};
  sa.x = reinterpret_cast<const char*>(a.xq), sa.w = reinterpret_cast<const char*>(a.qweight);
  sa.out = static_cast<uint16_t*>(a.out);
  sa.x_scale = a.x_scale, sa.w_scale = a.w_scale;
  sa.M = a.M, sa.N = a.N, sa.k_bytes = a.K, sa.gated = 1;
  sa.x_stride_bytes = a.K, sa.w_stride_bytes = a.K, sa.out_stride = a.out_stride_m;
  return stream_gemm_launch<true>(sa, plan, s);
}
