// Grouped expert GEMM on GPTQ experts (4- or 8-bit, group-wise scales and zeros) for decode-sized row counts per
// expert: out[slot] = x[a_rows[slot]] dequant(W[e]) for the sorted (token, choice) slots of expert e, the routing weight
// applied to the fp32 sums.  A GPTQ checkpoint of the 70B family lists the routed experts' gate / up / down projections
// among its quantised modules (hf_model/hymeta_gptq.py:30-32); the reference runs them through vLLM's quantised
// fused_moe (third party: parity unpinned).
//
// Built on the dense kernel for up to 32 rows (gptq_gemm_kernel in quant_gemm.hip): a packed int32 of qweight holds 8 (4)
// consecutive k of ONE output column, which is what the MFMA B operand wants per lane, so the weights go global -> VGPR
// -> MFMA; 4-bit: (q >> 4w) & 0x000F000F | 0x43004300 = the bf16 pair (128 + q_w, 128 + q_{w+4}) in one VALU op, the k
// order this induces applied to the activation fragment; scales and zeros once per group on the group's partial sums,
//   sum_k x w = s * ( sum_k x (c + q) - (c + z) * sum_k x ),   c = 128 (4-bit) or 0,
// with sum_k x from one extra MFMA against an all-ones fragment.  grid = (N / 64, local experts): a workgroup owns 64
// output columns of one expert over the whole K (the 4 waves take alternate groups, software-pipelined one step deep with
// register rings two groups ahead).  At decode sizes (TILES = false) grid.y = local experts and an expert has one 16-row
// tile (the loop over its tiles runs once); there are N / 64 x (experts hit) workgroups, several per compute unit, so
// no K slices are needed.  With more rows (TILES = true) the 32-row tiles of all local experts are numbered consecutively
// on grid.y and a workgroup finds its (expert, tile) from the offsets, as the tiled bf16 kernel does: a workgroup that
// walked all tiles of its expert would make the launch as long as the busiest expert (measured in the model, where the
// routing is uneven: 1.12 ms per layer at 2080 rows against 0.78 for evenly spread rows).  Rows are gathered through the
// sorted pair list as in the bf16 kernels (moe.hip).
#include <type_traits>

#include "common.h"

namespace hymeta {
namespace {

constexpr int kCols = 64;

__device__ __forceinline__ f32x4 mfma16(const bf16x8& a, const bf16x8& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ bf16x8 as_frag(const u32x4& v) { return __builtin_bit_cast(bf16x8, v); }

template <int BITS, int MT, int KSTEPS, bool TILES>
__global__ __launch_bounds__(256) void moe_gptq_gemm_kernel(hymeta_moe_gptq_gemm_args a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* s_red = reinterpret_cast<float*>(smem_raw);  // [4 waves][16 MT rows][64 cols]

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int c = lane >> 4, n = lane & 15;
  const int nb = blockIdx.x;
  int e, row_lo, row_hi;
  if constexpr (TILES) {
    // which expert, which of its row tiles
    e = -1, row_lo = 0, row_hi = 0;
    int t = blockIdx.y;
    for (int i = a.local_expert_begin; i < a.local_expert_end; ++i) {
      const int lo = a.expert_offsets[i], hi = a.expert_offsets[i + 1];
      const int tiles = (hi - lo + 16 * MT - 1) / (16 * MT);
      if (t < tiles) {
        e = i;
        row_lo = lo + t * 16 * MT;
        row_hi = hi;
        break;
      }
      t -= tiles;
    }
    if (e < 0) return;                                 // workgroup-uniform
  } else {
    e = a.local_expert_begin + blockIdx.y;
    row_lo = a.expert_offsets[e], row_hi = a.expert_offsets[e + 1];
    if (row_lo >= row_hi) return;                      // workgroup-uniform
  }

  constexpr int PACK = 32 / BITS;
  constexpr int QL = BITS == 4 ? 1 : 2;                // 16-byte weight loads per k-step
  const int G = a.K / a.group_size;
  const int col0 = nb * kCols + 4 * n;                 // this lane's first output column
  const int32_t* qweight = a.qweight + static_cast<int64_t>(e) * (a.K / PACK) * a.N;
  const int32_t* qzeros = a.qzeros + static_cast<int64_t>(e) * G * (a.N / PACK);
  const uint16_t* scales = reinterpret_cast<const uint16_t*>(a.scales) + static_cast<int64_t>(e) * G * a.N;
  const u32x4 ones_raw = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
  const bf16x8 ones = as_frag(ones_raw);

  const int first_g = wave;
  const int ng = first_g < G ? (G - first_g + 3) / 4 : 0;   // this wave's groups: wave, wave + 4, ...
  constexpr int kWDepth = 2 * KSTEPS;
  constexpr int kXCap = MT == 1 ? 8 : (MT == 2 ? 4 : 2);
  constexpr int kXDepth = kWDepth < kXCap ? kWDepth : kXCap;
  const int ngp = (ng + 1) & ~1;
  const int last_step = ng * KSTEPS - 1;

  for (int r0 = row_lo; r0 < (TILES ? min(row_hi, row_lo + 16 * MT) : row_hi); r0 += 16 * MT) {
    const uint16_t* xrow[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int slot = min(r0 + 16 * mt + n, row_hi - 1);   // rows past the expert's end re-read its last row
      const int src = a.a_rows != nullptr ? a.a_rows[slot] / a.a_rows_div : slot;
      xrow[mt] = reinterpret_cast<const uint16_t*>(a.x) + static_cast<int64_t>(src) * a.x_stride_m + 8 * c;
    }
    f32x4 acc_tot[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc_tot[mt][i] = {0.f, 0.f, 0.f, 0.f};

    // Register rings, refilled slot by slot as soon as a step has consumed its slot: packed weights two groups ahead,
    // activation fragments kXDepth steps ahead, a group's scales and zeros one group ahead.  Control flow is uniform and
    // every load unconditional (steps past the end re-read the last step; an odd group count is padded with a group of
    // zero scale), so that the compiler can count the loads in flight.
    u32x4 qring[kWDepth][QL], xring[kXDepth][MT];
    u32x2 sraw[2];
    uint32_t zraw[2];
    auto step_k0 = [&](int t) {
      const int g = first_g + 4 * (t / KSTEPS);
      return g * a.group_size + 32 * (t % KSTEPS);     // + 8c inside the pointers
    };
    auto load_w = [&](int t, u32x4 (&q)[QL]) {
      const int32_t* qp = qweight + static_cast<int64_t>((step_k0(min(t, last_step)) + 8 * c) / PACK) * a.N + col0;
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
      sr = *reinterpret_cast<const u32x2*>(scales + static_cast<int64_t>(g) * a.N + col0);
      zr = static_cast<uint32_t>(qzeros[static_cast<int64_t>(g) * (a.N / PACK) + col0 / PACK]);
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
        // ---- activations: 16 rows x 8 k per row tile
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
          for (int r = 0; r < 4; ++r) acc_tot[mt][i][r] += sc[i] * (acc_g[mt][i][r] - zf[i] * acc_x[mt][r]);
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

    // ---- the 4 waves' partial sums meet: s_red[wave][row][col], row = 16 mt + 4c + r, col = 4n + i
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float* p = s_red + (static_cast<size_t>(wave) * 16 * MT + 16 * mt + 4 * c + r) * kCols + 4 * n;
        *reinterpret_cast<f32x4*>(p) = f32x4{acc_tot[mt][0][r], acc_tot[mt][1][r], acc_tot[mt][2][r], acc_tot[mt][3][r]};
      }
    __syncthreads();
    const int cg = tid & 15;          // 4 columns per thread
    const int col = nb * kCols + 4 * cg;
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      const int row = (tid >> 4) + 16 * j;
      const int slot = r0 + row;
      if (slot < row_hi) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < 4; ++w) s += *reinterpret_cast<const f32x4*>(s_red + (static_cast<size_t>(w) * 16 * MT + row) * kCols + 4 * cg);
        // routing weight on the fp32 sums, one rounding
        if (a.row_scale != nullptr) s *= a.row_scale[a.scale_rows != nullptr ? a.scale_rows[slot] : slot];
        const u32x2 o = {pack_bf16x2(s[0], s[1]), pack_bf16x2(s[2], s[3])};
        *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(a.out) + static_cast<int64_t>(slot) * a.out_stride_m + col) = o;
      }
    }
    __syncthreads();
  }
}

// Every local expert's matrix expanded to bf16 W^T [N, K] (the layout of the bf16 expert stacks) in one launch, for
// prefill-sized batches: the tiled bf16 grouped GEMM then runs on the expansion.  grid = (N / 64, K / (16 PACK), experts);
// a workgroup expands 64 columns x 16 packed rows of one expert (128 k at 4 bits: a tensor-parallel shard's K is a whole
// number of groups, not of 256) through an LDS tile [column][k] and stores whole 16 PACK-element runs of one output row.
template <int BITS>
__global__ __launch_bounds__(256) void moe_gptq_expand_kernel(const int32_t* __restrict__ qweight, const int32_t* __restrict__ qzeros,
                                                              const void* __restrict__ scales, uint16_t* __restrict__ out, int K,
                                                              int N, int group_size, int scales_dtype, int zero_offset) {
  constexpr int PACK = 32 / BITS;
  constexpr int TKP = 16;                       // packed rows of a tile
  constexpr int TK = TKP * PACK;
  constexpr int PITCH = TK * 2 + 16;
  __shared__ __attribute__((aligned(16))) unsigned char tile[64 * PITCH];
  const int e = blockIdx.z;
  const int G = K / group_size;
  qweight += static_cast<int64_t>(e) * (K / PACK) * N;
  qzeros += static_cast<int64_t>(e) * G * (N / PACK);
  const uint16_t* sc16 = reinterpret_cast<const uint16_t*>(scales) + static_cast<int64_t>(e) * G * N;
  out += static_cast<int64_t>(e) * N * K;
  const int n0 = blockIdx.x * 64, kp0 = blockIdx.y * TKP;
  const int lane_n = threadIdx.x & 63, sub = threadIdx.x >> 6;
  const int n = n0 + lane_n;
#pragma unroll
  for (int i = 0; i < TKP / 4; ++i) {
    const int kp = kp0 + sub + 4 * i;
    const int g = kp * PACK / group_size;
    const uint32_t qw = static_cast<uint32_t>(qweight[static_cast<int64_t>(kp) * N + n]);
    const uint32_t zw = static_cast<uint32_t>(qzeros[static_cast<int64_t>(g) * (N / PACK) + n / PACK]);
    const uint16_t raw = sc16[static_cast<int64_t>(g) * N + n];
    const float sc = scales_dtype == HYMETA_F16 ? static_cast<float>(__builtin_bit_cast(_Float16, raw)) : bf2f(raw);
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
  constexpr int CHUNKS = TK * 2 / 16;
  for (int ch = threadIdx.x; ch < 64 * CHUNKS; ch += 256) {
    const int row = ch / CHUNKS, piece = ch % CHUNKS;
    const u32x4 v = *reinterpret_cast<const u32x4*>(tile + row * PITCH + piece * 16);
    *reinterpret_cast<u32x4*>(out + static_cast<int64_t>(n0 + row) * K + static_cast<int64_t>(kp0) * PACK + piece * 8) = v;
  }
}

int check_common(int K, int N, int bits, int group_size, int scales_dtype, int zero_offset) {
  if (bits != 4 && bits != 8) return HYMETA_ERR_UNSUPPORTED;
  if (group_size != 32 && group_size != 64 && group_size != 128) return HYMETA_ERR_UNSUPPORTED;
  if (K <= 0 || N <= 0 || K % group_size != 0 || N % kCols != 0) return HYMETA_ERR_UNSUPPORTED;
  if (scales_dtype != HYMETA_F16 && scales_dtype != HYMETA_BF16) return HYMETA_ERR_UNSUPPORTED;
  if (zero_offset != 0 && zero_offset != 1) return HYMETA_ERR_INVALID_ARG;
  return HYMETA_OK;
}

}  // namespace
}  // namespace hymeta

using namespace hymeta;

extern "C" int hymeta_moe_gptq_gemm(const hymeta_moe_gptq_gemm_args* args, hymeta_stream_t stream) {
  if (!hymeta::args_ok(args)) return HYMETA_ERR_INVALID_ARG;
  const hymeta_moe_gptq_gemm_args& a = *args;
  if (a.max_rows < 0) return HYMETA_ERR_INVALID_ARG;
  if (a.local_expert_begin < 0 || a.local_expert_end < a.local_expert_begin) return HYMETA_ERR_INVALID_ARG;
  if (const int rc = check_common(a.K, a.N, a.bits, a.group_size, a.scales_dtype, a.zero_offset)) return rc;
  if (a.max_rows == 0 || a.local_expert_end == a.local_expert_begin) return HYMETA_OK;
  if (!a.x || !a.qweight || !a.qzeros || !a.scales || !a.out || !a.expert_offsets) return HYMETA_ERR_INVALID_ARG;
  if (a.a_rows != nullptr && a.a_rows_div <= 0) return HYMETA_ERR_INVALID_ARG;
  if ((a.x_stride_m | a.out_stride_m) % 8 != 0) return HYMETA_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(a.x) | reinterpret_cast<uintptr_t>(a.qweight) | reinterpret_cast<uintptr_t>(a.out)) % 16 != 0 ||
      reinterpret_cast<uintptr_t>(a.scales) % 8 != 0)
    return HYMETA_ERR_UNSUPPORTED;
  const int experts = a.local_expert_end - a.local_expert_begin;
  hipStream_t s = begin_launches(stream);
  const dim3 grid(a.N / kCols, experts), block(256);
  // few rows per expert on average (decode): one 16-row tile per expert; otherwise 32-row tiles, numbered over all experts
  const bool small = static_cast<int64_t>(a.max_rows) <= 8LL * experts;
  const dim3 tgrid(a.N / kCols, (a.max_rows + 31) / 32 + experts);
  if (tgrid.y > 65535u) return HYMETA_ERR_UNSUPPORTED;
  const size_t lds1 = static_cast<size_t>(4) * 16 * kCols * sizeof(float), lds2 = 2 * lds1;
#define HYMETA_MOE_GPTQ(B, KS)                                                                         \
  do {                                                                                                 \
    if (small) hipLaunchKernelGGL((moe_gptq_gemm_kernel<B, 1, KS, false>), grid, block, lds1, s, a);    \
    else hipLaunchKernelGGL((moe_gptq_gemm_kernel<B, 2, KS, true>), tgrid, block, lds2, s, a);          \
  } while (0)
#define HYMETA_MOE_GPTQ_KS(B)                          \
  do {                                                 \
    if (a.group_size == 128) HYMETA_MOE_GPTQ(B, 4);    \
    else if (a.group_size == 64) HYMETA_MOE_GPTQ(B, 2); \
    else HYMETA_MOE_GPTQ(B, 1);                        \
  } while (0)
  if (a.bits == 4) HYMETA_MOE_GPTQ_KS(4); else HYMETA_MOE_GPTQ_KS(8);
#undef HYMETA_MOE_GPTQ_KS
#undef HYMETA_MOE_GPTQ
  return launch_status();
}

extern "C" int hymeta_moe_gptq_expand(const int32_t* qweight, const int32_t* qzeros, const void* scales, void* out,
                                      int32_t experts, int32_t K, int32_t N, int32_t bits, int32_t group_size,
                                      int32_t scales_dtype, int32_t zero_offset, hymeta_stream_t stream) {
  if (experts < 0) return HYMETA_ERR_INVALID_ARG;
  if (const int rc = check_common(K, N, bits, group_size, scales_dtype, zero_offset)) return rc;
  if (experts == 0) return HYMETA_OK;
  if (!qweight || !qzeros || !scales || !out) return HYMETA_ERR_INVALID_ARG;
  const int pack = 32 / bits;
  if (K % (16 * pack) != 0 || reinterpret_cast<uintptr_t>(out) % 16 != 0) return HYMETA_ERR_UNSUPPORTED;
  const dim3 grid(static_cast<unsigned>(N / 64), static_cast<unsigned>(K / (16 * pack)), static_cast<unsigned>(experts));
  if (grid.y > 65535u || grid.z > 65535u) return HYMETA_ERR_UNSUPPORTED;
  hipStream_t s = begin_launches(stream);
  uint16_t* o = static_cast<uint16_t*>(out);
  if (bits == 4) hipLaunchKernelGGL(moe_gptq_expand_kernel<4>, grid, dim3(256), 0, s, qweight, qzeros, scales, o, K, N, group_size, scales_dtype, zero_offset);
  else hipLaunchKernelGGL(moe_gptq_expand_kernel<8>, grid, dim3(256), 0, s, qweight, qzeros, scales, o, K, N, group_size, scales_dtype, zero_offset);
  return launch_status();
}
