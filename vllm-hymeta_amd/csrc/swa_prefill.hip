// Varlen prefill attention: causal sliding window over the real keys plus an
// always-visible block of meta keys, one softmax (flash-attention forward).
//
// MFMA-bound.  A 512-thread workgroup owns 256 consecutive queries of one
// (sequence, q-head); each of its 8 waves owns 32 of them as two 16-row MFMA
// tiles.  The workgroup walks the visible keys in tiles of 64: all threads
// stage the tile's K and V rows once (global -> registers one tile ahead,
// registers -> LDS right after the barrier that frees the buffer, one barrier
// per tile, two LDS buffers), and every wave reads its operands from that
// shared image:
//   S^T = K Q^T   A = 16 key rows x 32 channels, ds_read_b128 from the K image
//                 (256-byte rows, 16-byte chunks XOR-ed with the row so the
//                 sixteen rows of a read hit distinct banks);
//                 B = the wave's resident Q fragments.
//   O  += P V     A = the S^T accumulators re-packed to bf16: the query is on
//                 the lane and the keys are in registers, which is the
//                 A-operand layout up to a fixed permutation of the key slots;
//                 B = V in the same permuted key order, two transposed 8-byte
//                 LDS reads (ds_read_b64_tr_b16) per fragment from the V image
//                 (288-byte rows: the four rows of a read sit 8 banks apart).
// Keys are read through a block table (chunked / prefix prefill) or from the
// packed [token, kv-head, dim] tensor of the step.  Outputs are normalised,
// staged through LDS and stored as whole 256-byte rows.
#include "common.h"

namespace hymeta {
namespace {

constexpr int kTile = 32;            // granularity required of the meta-key count
constexpr int kTileK = 64;           // keys per step
constexpr int kRowsPerWave = 32;
constexpr int kMaxWaves = 8;
constexpr int kKPitch = 256;
constexpr int kVPitch = 288;
constexpr int kKBytes = kTileK * kKPitch;
constexpr int kVBytes = kTileK * kVPitch;
constexpr int kOPitch = 272;         // output staging rows (bf16 x 128 + pad)
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLazyLog2 = 8.f;     // the softmax reference may trail the true row maximum by a factor 2^8

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

struct Smem {
  char k[2][kKBytes];
  char v[2][kVBytes];
};
static_assert(sizeof(Smem) >= kMaxWaves * kRowsPerWave * kOPitch, "output staging aliases the K/V images");

__device__ __forceinline__ f32x4 mfma16(const bf16x8& a, const bf16x8& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ bf16x8 as_frag(const u32x4& v) { return __builtin_bit_cast(bf16x8, v); }

// max of three as ONE instruction: v_maximum3_f32 (NaN-propagating, so the compiler adds no
// canonicalising v_max per MFMA output as it does for fmaxf, and no wait states as after inline asm).
__device__ __forceinline__ float max3(float a, float b, float c) {
  return __builtin_elementwise_maximum(__builtin_elementwise_maximum(a, b), c);
}

// 4 keys x 16 dims block read column-major: lane i of a 16-lane group gets dim i of the 4 keys.
__device__ __forceinline__ s16x4 tr_read(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
}

}  // namespace

// grid = (ceil(max_seqlen_q / (32 NW)), num_seqs (+1: the meta-query rows), num_heads); NW waves of
// 32 queries per workgroup: 8 by default, 4 when that leaves compute units without a workgroup.
//
// splits > 1 (few workgroups, long key ranges: a short chunk of a long sequence): blockIdx.x also
// numbers the key slice; a workgroup walks only its share of the tiles and leaves an un-normalised
// partial (fp32 O, the reference exponent and the row sum per query) in the workspace; a second
// launch merges the slices in slice order.
template <bool PAGED, int NW>
__global__ __launch_bounds__(64 * NW, 8 / NW) void swa_prefill_kernel(hymeta_swa_prefill_args a, float scale_log2, int splits,
                                                                      int64_t part_rows) {
  constexpr int kRowsPerBlock = kRowsPerWave * NW;
  constexpr int kPasses = 16 / NW;     // staging passes: 4 NW key rows of a 64-key tile per pass

  __shared__ __attribute__((aligned(16))) Smem sm;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int c = lane >> 4;
  const int n = lane & 15;

  const int seq = blockIdx.y;
  const int h = blockIdx.z;
  const int kvh = h / (a.num_heads / a.num_kv_heads);
  const bool meta_rows = seq == a.num_seqs;  // this workgroup computes the meta-query rows
  const int M = a.meta_key != nullptr ? a.num_meta : 0;
  const int meta_tiles = (M + kTileK - 1) / kTileK;
  const int out_row0 = a.meta_query != nullptr ? a.num_meta : 0;

  int Lq, Lk, q_first;        // query count, key count, first query row of the sequence
  const uint16_t* qbase;
  int64_t q_stride_t;
  int64_t out_first;          // output row of the sequence's first query
  int64_t k_first = 0;        // un-paged: first key row of the sequence
  if (meta_rows) {
    Lq = a.num_meta;
    Lk = 0;
    qbase = reinterpret_cast<const uint16_t*>(a.meta_query) + static_cast<int64_t>(h) * a.meta_q_stride_h;
    q_stride_t = a.meta_q_stride_t;
    out_first = 0;
  } else {
    q_first = a.cu_seqlens_q[seq];
    Lq = a.cu_seqlens_q[seq + 1] - q_first;
    if constexpr (PAGED) {
      Lk = a.seq_lens_k[seq];
    } else {
      k_first = a.cu_seqlens_k[seq];
      Lk = a.cu_seqlens_k[seq + 1] - static_cast<int>(k_first);
    }
    qbase = reinterpret_cast<const uint16_t*>(a.q) + static_cast<int64_t>(q_first) * a.q_stride_t + static_cast<int64_t>(h) * a.q_stride_h;
    q_stride_t = a.q_stride_t;
    out_first = out_row0 + q_first;
  }
  const int split = splits > 1 ? blockIdx.x % splits : 0;
  const int blk0 = (splits > 1 ? blockIdx.x / splits : blockIdx.x) * kRowsPerBlock;
  if (blk0 >= Lq) return;                      // workgroup-uniform
  const int row0 = blk0 + wave * kRowsPerWave;  // first query of this wave
  const bool active = row0 < Lq;               // idle waves still stage tiles and meet the barriers
  const int pos_shift = Lk - Lq;  // query r sits at key position r + pos_shift (bottom-right aligned)

  // Q fragments: qf[qt][ks] = Q[row0 + 16qt + n][32ks + 8c .. +7]
  bf16x8 qf[2][4];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int r = row0 + 16 * qt + n;
    const uint16_t* qp = qbase + static_cast<int64_t>(r) * q_stride_t + 8 * c;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      u32x4 raw = {0u, 0u, 0u, 0u};
      if (r < Lq) raw = ld16(qp + 32 * ks);
      qf[qt][ks] = as_frag(raw);
    }
  }

  f32x4 o[2][8];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int t = 0; t < 8; ++t) o[qt][t] = {0.f, 0.f, 0.f, 0.f};
  float m_run[2] = {-INFINITY, -INFINITY};   // reference score of the exponentials (a lazy running maximum)
  float ms_run[2] = {0.f, 0.f};              // m_run * scale_log2, 0 while m_run is -inf
  f32x4 l_acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};   // l_acc[qt][r]: softmax denominator of query 16qt + 4c + r
  const bf16x8 ones = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};

  // Real-key tile ranges: of this wave's 32 queries and of the workgroup's 256.
  const int last_row = min(row0 + kRowsPerWave, Lq) - 1;
  const int p_lo = row0 + pos_shift, p_hi = last_row + pos_shift;
  int tw_lo = 0, tw_hi = -1, T_lo = 0, T_hi = -1;
  if (!meta_rows && Lk > 0) {
    if (active && p_hi >= 0) {
      const int k_lo = (a.window_left >= 0 && p_lo - a.window_left > 0) ? p_lo - a.window_left : 0;
      const int k_hi = p_hi < Lk - 1 ? p_hi : Lk - 1;
      tw_lo = k_lo / kTileK;
      tw_hi = k_hi / kTileK;
    }
    const int P_lo = blk0 + pos_shift, P_hi = min(blk0 + kRowsPerBlock, Lq) - 1 + pos_shift;
    if (P_hi >= 0) {
      const int k_lo = (a.window_left >= 0 && P_lo - a.window_left > 0) ? P_lo - a.window_left : 0;
      const int k_hi = P_hi < Lk - 1 ? P_hi : Lk - 1;
      T_lo = k_lo / kTileK;
      T_hi = k_hi / kTileK;
    }
  }
  const int all_tiles = meta_tiles + (T_hi - T_lo + 1);
  // this workgroup's share of the tile sequence (all of it when there is one slice)
  const int per_split = (all_tiles + splits - 1) / splits;
  const int t_begin = min(split * per_split, all_tiles);
  const int ntiles = min(t_begin + per_split, all_tiles);     // one past the last tile walked here

  const uint16_t* kglob = nullptr;
  const uint16_t* vglob = nullptr;
  const int32_t* bt = nullptr;
  if constexpr (PAGED) {
    kglob = reinterpret_cast<const uint16_t*>(a.key_cache) + static_cast<int64_t>(kvh) * a.cache_stride_h;
    vglob = reinterpret_cast<const uint16_t*>(a.value_cache) + static_cast<int64_t>(kvh) * a.cache_stride_h;
    bt = a.block_tables + static_cast<int64_t>(seq) * a.block_table_stride;
  } else if (!meta_rows) {
    kglob = reinterpret_cast<const uint16_t*>(a.key) + k_first * a.k_stride_t + static_cast<int64_t>(kvh) * a.k_stride_h;
    vglob = reinterpret_cast<const uint16_t*>(a.value) + k_first * a.v_stride_t + static_cast<int64_t>(kvh) * a.v_stride_h;
  }

  // ---- staging role: thread = 16-byte chunk sc of key rows sr + 4 NW i (i < kPasses) of the tile
  const int sr = tid >> 4, sc = tid & 15;
  constexpr int kRowStep = 4 * NW;
  // Loads are unconditional (rows past the end are clamped to the last valid row and zeroed when
  // they are committed), so nothing has to wait for them before the tile's compute.
  u32x4 kreg[kPasses], vreg[kPasses];
  bool rok[kPasses] = {};
  // Un-paged keys: the thread's two row pointers walk down the tile sequence (one 64-bit add per
  // pointer and tile); they are rebuilt only where the sequence of tiles changes tensor (first
  // tile, first real tile after the meta tiles).  klast / vlast: the last valid row of the tensor
  // being walked, what a row past the end reads instead.
  const uint16_t* kptr[kPasses] = {};
  const uint16_t* vptr[kPasses] = {};
  const uint16_t* klast = nullptr;
  const uint16_t* vlast = nullptr;
  int64_t kstep = 0, vstep = 0;
  int prow = 0, plimit = 0;            // row index of kptr[0] in its tensor, and that tensor's row count
  auto issue = [&](int tile) {
    if constexpr (PAGED) {
#pragma unroll
      for (int i = 0; i < kPasses; ++i) {
        const int r = sr + kRowStep * i;
        const uint16_t* kp;
        const uint16_t* vp;
        if (tile < meta_tiles) {
          const int idx = tile * kTileK + r;
          rok[i] = idx < M;
          const int64_t off = static_cast<int64_t>(min(idx, M - 1)) * a.meta_stride_t + static_cast<int64_t>(kvh) * a.meta_stride_h + 8 * sc;
          kp = reinterpret_cast<const uint16_t*>(a.meta_key) + off;
          vp = reinterpret_cast<const uint16_t*>(a.meta_value) + off;
        } else {
          const int p_raw = (T_lo + tile - meta_tiles) * kTileK + r;
          rok[i] = p_raw < Lk;   // rows past the sequence end: stale cache
          const int p = min(p_raw, Lk - 1);
          const int64_t off = static_cast<int64_t>(bt[p / a.block_size]) * a.cache_stride_block +
                              static_cast<int64_t>(p % a.block_size) * a.cache_stride_pos + 8 * sc;
          kp = kglob + off;
          vp = vglob + off;
        }
        kreg[i] = ld16(kp);
        vreg[i] = ld16(vp);
      }
    } else {
      if (tile == t_begin || tile == meta_tiles) { // workgroup-uniform, at most twice per workgroup
        const bool meta = tile < meta_tiles;
        const uint16_t* kb = meta ? reinterpret_cast<const uint16_t*>(a.meta_key) + static_cast<int64_t>(kvh) * a.meta_stride_h : kglob;
        const uint16_t* vb = meta ? reinterpret_cast<const uint16_t*>(a.meta_value) + static_cast<int64_t>(kvh) * a.meta_stride_h : vglob;
        const int64_t ks = meta ? a.meta_stride_t : a.k_stride_t, vs = meta ? a.meta_stride_t : a.v_stride_t;
        plimit = meta ? M : Lk;
        prow = (meta ? tile : T_lo + tile - meta_tiles) * kTileK + sr;
        kstep = ks * kTileK, vstep = vs * kTileK;
#pragma unroll
        for (int i = 0; i < kPasses; ++i) {
          kptr[i] = kb + static_cast<int64_t>(prow + kRowStep * i) * ks + 8 * sc;
          vptr[i] = vb + static_cast<int64_t>(prow + kRowStep * i) * vs + 8 * sc;
        }
        klast = kb + static_cast<int64_t>(plimit - 1) * ks + 8 * sc;
        vlast = vb + static_cast<int64_t>(plimit - 1) * vs + 8 * sc;
      }
#pragma unroll
      for (int i = 0; i < kPasses; ++i) {
        rok[i] = prow + kRowStep * i < plimit;           // rows past the end: next sequence, or outside the tensor
        kreg[i] = ld16(rok[i] ? kptr[i] : klast);
        vreg[i] = ld16(rok[i] ? vptr[i] : vlast);
        kptr[i] += kstep;
        vptr[i] += vstep;
      }
      prow += kTileK;
    }
  };
  auto commit = [&](int buf) {
    const u32x4 z = {0u, 0u, 0u, 0u};
    bool all_ok = true;
#pragma unroll
    for (int i = 0; i < kPasses; ++i) all_ok = all_ok && rok[i];
    if (__all(all_ok)) {
#pragma unroll
      for (int i = 0; i < kPasses; ++i) {
        const int r = sr + kRowStep * i;
        st16(&sm.k[buf][r * kKPitch + 16 * (sc ^ (r & 15))], kreg[i]);
        st16(&sm.v[buf][r * kVPitch + 16 * sc], vreg[i]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < kPasses; ++i) {
        const int r = sr + kRowStep * i;
        st16(&sm.k[buf][r * kKPitch + 16 * (sc ^ (r & 15))], rok[i] ? kreg[i] : z);
        st16(&sm.v[buf][r * kVPitch + 16 * sc], rok[i] ? vreg[i] : z);
      }
    }
  };

  if (ntiles > t_begin) {
    issue(t_begin);
    commit(0);
  }
  if (ntiles > t_begin + 1) issue(t_begin + 1);
  // Pin the Q fragments as landed here: otherwise the first MFMA of EVERY tile waits for all
  // outstanding global loads (the just-issued next tile included) on behalf of the Q loads above.
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(qf[qt][ks]));
  __syncthreads();

  for (int tile = t_begin; tile < ntiles; ++tile) {
    const int buf = (tile - t_begin) & 1;
    // Tile t + 1 (loaded during the previous tile) goes to the other buffer now, behind the barrier
    // that ended its last reads, and tile t + 2 is requested at once: the stores overlap the
    // other waves' matrix work instead of standing between every wave's last MFMA and the barrier.
    if (tile + 1 < ntiles) commit(buf ^ 1);
    if (tile + 2 < ntiles) issue(tile + 2);

    const bool is_meta = tile < meta_tiles;
    const int rt = T_lo + tile - meta_tiles;   // real tile index
    if (active && (is_meta || (rt >= tw_lo && rt <= tw_hi))) {
      const int pos0 = is_meta ? tile * kTileK : rt * kTileK;
      // Every (query, key) pair of the tile visible?  (wave-uniform)
      const bool full = is_meta ? (pos0 + kTileK <= M)
                                : (pos0 + kTileK - 1 <= p_lo && pos0 + kTileK <= Lk &&
                                   (a.window_left < 0 || p_hi - pos0 <= a.window_left));
      const char* kimg = sm.k[buf];
      const char* vimg = sm.v[buf];

      // S^T tiles: st[qt][s][r] = score(key 16s + 4c + r, query 16qt + n)
      f32x4 st[2][4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        bf16x8 kf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) kf[ks] = as_frag(ld16(kimg + (16 * s + n) * kKPitch + 16 * ((4 * ks + c) ^ n)));
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          st[qt][s] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) st[qt][s] = mfma16(kf[ks], qf[qt][ks], st[qt][s]);
        }
      }

      if (!full) {
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          // visible key positions of this lane's query: [lo, hi], as one unsigned range test
          const int qp = row0 + 16 * qt + n + pos_shift;
          int lo = 0, hi = M - 1;
          if (!is_meta) {
            lo = a.window_left >= 0 ? max(qp - a.window_left, 0) : 0;
            hi = min(qp, Lk - 1);
          }
          const bool any = hi >= lo;
          const unsigned span = static_cast<unsigned>(hi - lo);
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int kp = pos0 + 16 * s + 4 * c + r;
              const bool ok = any & (static_cast<unsigned>(kp - lo) <= span);
              st[qt][s][r] = ok ? st[qt][s][r] : -INFINITY;
            }
        }
      }

      // online softmax per query column
      bf16x8 pf[2][2];
      float tnew[2];
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        float tmax = max3(st[qt][0][0], st[qt][0][1], st[qt][0][2]);
        tmax = max3(tmax, st[qt][0][3], st[qt][1][0]);
        tmax = max3(tmax, st[qt][1][1], st[qt][1][2]);
        tmax = max3(tmax, st[qt][1][3], st[qt][2][0]);
        tmax = max3(tmax, st[qt][2][1], st[qt][2][2]);
        tmax = max3(tmax, st[qt][2][3], st[qt][3][0]);
        tmax = max3(tmax, st[qt][3][1], st[qt][3][2]);
        tmax = max3(tmax, st[qt][3][3], st[qt][3][3]);
        // across the four 16-lane groups (the tile's other keys) with the two row-swap
        // instructions: ALU latency on the tile's critical path instead of two LDS round trips
        const auto h16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
        tmax = max3(__uint_as_float(h16[0]), __uint_as_float(h16[1]), tmax);
        const auto h32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
        tnew[qt] = max3(__uint_as_float(h32[0]), __uint_as_float(h32[1]), tmax);
      }
      // The running reference m_run follows the row maximum lazily: it moves (and the running
      // output is rescaled) only when some query's scores exceed it by more than 2^kLazyLog2, so
      // P stays <= 2^kLazyLog2 and most tiles skip the pass over the 64 output registers.  The
      // decision precedes this tile's exponentials and follows the previous tile's P V.
      const bool grow = (tnew[0] - m_run[0]) * scale_log2 > kLazyLog2 || (tnew[1] - m_run[1]) * scale_log2 > kLazyLog2;
      if (__any(grow)) {
        float alpha[2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          const float m_new = max3(m_run[qt], tnew[qt], tnew[qt]);
          const float m_use = m_new == -INFINITY ? 0.f : m_new;
          alpha[qt] = __builtin_amdgcn_exp2f((m_run[qt] - m_use) * scale_log2);
          m_run[qt] = m_new;
          ms_run[qt] = m_use * scale_log2;
        }
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float ar = __shfl(alpha[qt], 4 * c + r);  // o rows are queries 4c + r
#pragma unroll
            for (int t = 0; t < 8; ++t) o[qt][t][r] *= ar;
            l_acc[qt][r] *= ar;
          }
      }
      // P re-packed as the A operand of P V:
      // fragment kh, slot j <-> key 32kh + (j < 4 ? 4c + j : 16 + 4c + j - 4)
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        const float ms = ms_run[qt];
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
          float p[8];
#pragma unroll
          for (int j = 0; j < 8; ++j)
            p[j] = __builtin_amdgcn_exp2f(fmaf(st[qt][2 * kh + (j >> 2)][j & 3], scale_log2, -ms));
          u32x4 pw;
#pragma unroll
          for (int i = 0; i < 4; ++i) pw[i] = pack_bf16x2(p[2 * i], p[2 * i + 1]);
          pf[qt][kh] = as_frag(pw);
        }
      }

      // O += P V : o[qt][t][r] = O[query 16qt + 4c + r][dim 16t + n]
      const char* vlane = vimg + (4 * c + (n >> 2)) * kVPitch + 8 * (n & 3);
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        // row sums of the bf16 P that enters P V, on the matrix cores (B = ones)
        l_acc[0] = mfma16(pf[0][kh], ones, l_acc[0]);
        l_acc[1] = mfma16(pf[1][kh], ones, l_acc[1]);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const s16x4 lo = tr_read(vlane + (32 * kh) * kVPitch + 32 * t);
          const s16x4 hi = tr_read(vlane + (32 * kh + 16) * kVPitch + 32 * t);
          const bf16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          o[0][t] = mfma16(pf[0][kh], vf, o[0][t]);
          o[1][t] = mfma16(pf[1][kh], vf, o[1][t]);
        }
      }
    }

    __syncthreads();
  }

  if (splits > 1) {
    // partial of this key slice: O un-normalised (fp32), per query its reference exponent (the
    // lanes of the first 16-lane group hold the queries' references) and its row sum
    float* po = static_cast<float*>(a.workspace);
    float* pm = po + static_cast<int64_t>(splits) * part_rows * a.num_heads * kHeadDim;
    float* pl = pm + static_cast<int64_t>(splits) * part_rows * a.num_heads;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      if (c == 0) {
        const int qr = row0 + 16 * qt + n;
        if (qr < Lq) pm[(split * part_rows + out_first + qr) * a.num_heads + h] = m_run[qt] == -INFINITY ? -INFINITY : ms_run[qt];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qr = row0 + 16 * qt + 4 * c + r;
        if (qr >= Lq) continue;
        const int64_t prow_ = (split * part_rows + out_first + qr) * a.num_heads + h;
        if (n == 0) pl[prow_] = l_acc[qt][r];
#pragma unroll
        for (int t = 0; t < 8; ++t) po[prow_ * kHeadDim + 16 * t + n] = o[qt][t][r];
      }
    }
    return;
  }

  // epilogue: normalise, stage the wave's 32 x 128 outputs in LDS (the K/V images are free after
  // the loop's last barrier), store whole rows.
  char* ost = reinterpret_cast<char*>(&sm) + wave * (kRowsPerWave * kOPitch);
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float lr = l_acc[qt][r];
      const float inv = lr > 0.f ? 1.f / lr : 0.f;
      uint16_t* orow = reinterpret_cast<uint16_t*>(ost + (16 * qt + 4 * c + r) * kOPitch) + n;
#pragma unroll
      for (int t = 0; t < 8; ++t) orow[16 * t] = f2bf(o[qt][t][r] * inv);
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int id = lane + 64 * i;
    const int row = id >> 4, ch = id & 15;
    const int qr = row0 + row;
    if (qr < Lq) {
      uint16_t* op = reinterpret_cast<uint16_t*>(a.out) + (out_first + qr) * a.out_stride_t +
                     static_cast<int64_t>(h) * a.out_stride_h + 8 * ch;
      st16(op, ld16(ost + row * kOPitch + 16 * ch));
    }
  }
}

// One wave per (output row, head): out = sum_s 2^(m_s - m) O_s / sum_s 2^(m_s - m) l_s, slices in order.
__global__ __launch_bounds__(256) void swa_prefill_merge_kernel(hymeta_swa_prefill_args a, int splits, int64_t part_rows) {
  const int lane = threadIdx.x & 63;
  const int64_t pair = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);     // row * H + head
  if (pair >= part_rows * a.num_heads) return;
  const float* po = static_cast<const float*>(a.workspace);
  const float* pm = po + static_cast<int64_t>(splits) * part_rows * a.num_heads * kHeadDim;
  const float* pl = pm + static_cast<int64_t>(splits) * part_rows * a.num_heads;
  float m = -INFINITY;
  for (int s = 0; s < splits; ++s) m = fmaxf(m, pm[s * part_rows * a.num_heads + pair]);
  float o0 = 0.f, o1 = 0.f, l = 0.f;
  for (int s = 0; s < splits; ++s) {
    const float ms = pm[s * part_rows * a.num_heads + pair];
    if (ms == -INFINITY) continue;           // a slice without a visible key for this query
    const float w = __builtin_amdgcn_exp2f(ms - m);
    const f32x2 v = *reinterpret_cast<const f32x2*>(po + (s * part_rows * a.num_heads + pair) * kHeadDim + 2 * lane);
    o0 += w * v[0];
    o1 += w * v[1];
    l += w * pl[s * part_rows * a.num_heads + pair];
  }
  const float inv = l > 0.f ? 1.f / l : 0.f;
  const int64_t row = pair / a.num_heads;
  const int head = static_cast<int>(pair % a.num_heads);
  uint32_t* op = reinterpret_cast<uint32_t*>(reinterpret_cast<uint16_t*>(a.out) + row * a.out_stride_t +
                                             static_cast<int64_t>(head) * a.out_stride_h) + lane;
  *op = pack_bf16x2(o0 * inv, o1 * inv);
}

}  // namespace hymeta

using namespace hymeta;

constexpr int kMaxPrefillSplits = 8;

static size_t prefill_partial_bytes(int64_t rows, int heads, int splits) {
  return static_cast<size_t>(splits) * rows * heads * (kHeadDim + 2) * sizeof(float);
}

// Key slices for a call: only when 128-query workgroups leave most compute units idle and every
// slice still walks a fair number of tiles (a bound from the window or max_seqlen_k; unknown: 1).
extern "C" int hymeta_meta_swa_prefill_splits(int32_t total_q, int32_t num_heads, int32_t num_meta, int32_t meta_rows,
                                              int32_t window_left, int32_t max_seqlen_k) {
  if (total_q <= 0 || num_heads <= 0) return 1;
  const int64_t wgs256 = (static_cast<int64_t>(total_q) + 255) / 256 * num_heads + static_cast<int64_t>(num_heads) * ((meta_rows + 255) / 256);
  if (wgs256 >= 128) return 1;
  const int64_t wgs128 = (static_cast<int64_t>(total_q) + 127) / 128 * num_heads + static_cast<int64_t>(num_heads) * ((meta_rows + 127) / 128);
  int64_t keys = max_seqlen_k > 0 ? max_seqlen_k : -1;
  if (window_left >= 0 && (keys < 0 || window_left + 128 < keys)) keys = window_left + 128;
  if (keys < 0) return 1;
  const int64_t tiles = (keys + kTileK - 1) / kTileK + 1 + (num_meta + kTileK - 1) / kTileK;
  int64_t splits = 512 / (wgs128 > 0 ? wgs128 : 1);          // two 128-query workgroups fit a compute unit
  if (splits > tiles / 8) splits = tiles / 8;
  if (splits > kMaxPrefillSplits) splits = kMaxPrefillSplits;
  return splits < 2 ? 1 : static_cast<int>(splits);
}

extern "C" size_t hymeta_meta_swa_prefill_workspace_bytes(int32_t out_rows, int32_t num_heads, int32_t num_splits) {
  if (out_rows <= 0 || num_heads <= 0 || num_splits == 1) return 0;
  return prefill_partial_bytes(out_rows, num_heads, num_splits <= 0 ? kMaxPrefillSplits : num_splits);
}

extern "C" int hymeta_meta_swa_prefill(const hymeta_swa_prefill_args* args, hymeta_stream_t stream) {
  if (!hymeta::args_ok(args)) return HYMETA_ERR_INVALID_ARG;
  hymeta_swa_prefill_args a = *args;
  if (a.num_seqs < 0 || a.total_q < 0 || a.num_heads <= 0 || a.num_kv_heads <= 0) return HYMETA_ERR_INVALID_ARG;
  if (a.head_dim != kHeadDim) return HYMETA_ERR_UNSUPPORTED;
  if (a.num_heads % a.num_kv_heads != 0) return HYMETA_ERR_INVALID_ARG;
  if ((a.meta_key == nullptr) != (a.meta_value == nullptr)) return HYMETA_ERR_INVALID_ARG;
  if (a.meta_key == nullptr) a.num_meta = 0;
  if (a.num_meta < 0 || a.num_meta % kTile != 0) return HYMETA_ERR_UNSUPPORTED;
  if (a.meta_query != nullptr && a.num_meta == 0) return HYMETA_ERR_INVALID_ARG;
  const bool have_rows = a.num_seqs > 0 && a.total_q > 0;
  if (!have_rows && a.meta_query == nullptr) return HYMETA_OK;
  if (!a.out) return HYMETA_ERR_INVALID_ARG;
  const bool paged = a.block_tables != nullptr;
  if (have_rows) {
    if (!a.q || !a.cu_seqlens_q) return HYMETA_ERR_INVALID_ARG;
    if (paged) {
      if (!a.key_cache || !a.value_cache || !a.seq_lens_k || a.block_size <= 0) return HYMETA_ERR_INVALID_ARG;
      if (a.block_size % 16 != 0) return HYMETA_ERR_UNSUPPORTED;
    } else if (!a.key || !a.value || !a.cu_seqlens_k) {
      return HYMETA_ERR_INVALID_ARG;
    }
  }
  const int64_t strides[] = {a.q_stride_t, a.q_stride_h, a.k_stride_t, a.k_stride_h, a.v_stride_t, a.v_stride_h,
                             a.out_stride_t, a.out_stride_h, a.cache_stride_block, a.cache_stride_pos,
                             a.cache_stride_h, a.meta_q_stride_t, a.meta_q_stride_h, a.meta_stride_t, a.meta_stride_h};
  for (int64_t st : strides)
    if (st % 8 != 0) return HYMETA_ERR_UNSUPPORTED;
  const uintptr_t ptrs = reinterpret_cast<uintptr_t>(a.q) | reinterpret_cast<uintptr_t>(a.key) |
                         reinterpret_cast<uintptr_t>(a.value) | reinterpret_cast<uintptr_t>(a.key_cache) |
                         reinterpret_cast<uintptr_t>(a.value_cache) | reinterpret_cast<uintptr_t>(a.meta_query) |
                         reinterpret_cast<uintptr_t>(a.meta_key) | reinterpret_cast<uintptr_t>(a.meta_value) |
                         reinterpret_cast<uintptr_t>(a.out);
  if (ptrs % 16 != 0) return HYMETA_ERR_UNSUPPORTED;

  const float scale = a.scale > 0.f ? a.scale : 1.f / sqrtf(static_cast<float>(a.head_dim));
  int max_rows = have_rows ? a.max_seqlen_q : 0;
  if (have_rows && max_rows <= 0) max_rows = a.total_q;
  if (a.meta_query != nullptr && a.num_meta > max_rows) max_rows = a.num_meta;
  const int seqs = have_rows ? a.num_seqs : 0;
  if (!have_rows) a.num_seqs = 0;
  // 256-query workgroups, unless they would leave more than half of the compute units idle:
  // then 128-query ones (the same tiles staged by half as many threads, two workgroups per
  // compute unit; measured 13 % faster on a 2048-query chunk of one 7-head shard, slower once the
  // 256-query workgroups already fill the chip)
  const int64_t wgs256 = (static_cast<int64_t>(have_rows ? a.total_q : 0) + 255) / 256 * a.num_heads +
                         (a.meta_query != nullptr ? static_cast<int64_t>(a.num_heads) * ((a.num_meta + 255) / 256) : 0);
  const int nw = wgs256 < 128 ? 4 : 8;
  const int rows_per_block = kRowsPerWave * nw;
  // key slices: the library's choice needs a workspace that holds it (without one: a single slice)
  const int64_t out_rows = static_cast<int64_t>(have_rows ? a.total_q : 0) + (a.meta_query != nullptr ? a.num_meta : 0);
  int splits = a.num_splits > 0 ? a.num_splits
                                : hymeta_meta_swa_prefill_splits(have_rows ? a.total_q : 0, a.num_heads, a.num_meta,
                                                                 a.meta_query != nullptr ? a.num_meta : 0, a.window_left, a.max_seqlen_k);
  if (splits > kMaxPrefillSplits) return HYMETA_ERR_UNSUPPORTED;
  if (splits > 1 && (a.workspace == nullptr || a.workspace_bytes < prefill_partial_bytes(out_rows, a.num_heads, splits))) {
    if (a.num_splits > 1) return HYMETA_ERR_WORKSPACE;
    splits = 1;
  }
  if (splits > 1 && reinterpret_cast<uintptr_t>(a.workspace) % 16 != 0) return HYMETA_ERR_UNSUPPORTED;
  dim3 grid((max_rows + rows_per_block - 1) / rows_per_block * splits, seqs + (a.meta_query != nullptr ? 1 : 0), a.num_heads);
  hipStream_t s = begin_launches(stream);
  const float sl2 = scale * kLog2e;
  if (paged) {
    if (nw == 8) hipLaunchKernelGGL((swa_prefill_kernel<true, 8>), grid, dim3(512), 0, s, a, sl2, splits, out_rows);
    else hipLaunchKernelGGL((swa_prefill_kernel<true, 4>), grid, dim3(256), 0, s, a, sl2, splits, out_rows);
  } else {
    if (nw == 8) hipLaunchKernelGGL((swa_prefill_kernel<false, 8>), grid, dim3(512), 0, s, a, sl2, splits, out_rows);
    else hipLaunchKernelGGL((swa_prefill_kernel<false, 4>), grid, dim3(256), 0, s, a, sl2, splits, out_rows);
  }
  if (splits > 1) {
    if (hipGetLastError() != hipSuccess) return HYMETA_ERR_LAUNCH;
    const int64_t pairs = out_rows * a.num_heads;
    hipLaunchKernelGGL(swa_prefill_merge_kernel, dim3(static_cast<unsigned>((pairs + 3) / 4)), dim3(256), 0, s, a, splits, out_rows);
  }
  return launch_status();
}
