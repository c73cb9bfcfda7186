// Single-query paged decode attention with a sliding window over the paged
// real keys and an always-visible tile of meta keys, one softmax (split-KV).
//
// HBM-bound: each (sequence, kv-head) streams its visible K and V rows once
// (256 B per row) and the q-heads that share the kv-head ride along as the 16
// columns of an MFMA tile, so the arithmetic costs nothing next to the loads.
// Nothing goes through LDS on the way in:
//   * S^T = K Q^T with v_mfma_f32_16x16x32_bf16: the A operand wants, per lane,
//     16 contiguous bytes of one key row -- exactly a coalesced global load.
//   * The S^T accumulator has the q-head on the lane and keys in registers,
//     which is already the A-operand layout of P for O += P V (the k-slots
//     are a fixed permutation of the tile's 32 keys; V is loaded through the
//     same permutation).
//   * V must be the B operand with the KEY index packed inside a lane, but is
//     stored with the channel index contiguous.  Each lane loads 8 key rows x
//     16 bytes and transposes the 8x8 block of 16-bit values in registers
//     (32 v_perm_b32), which yields the B fragments of 8 MFMAs at once.
// Two workgroup shapes:
//   HKV == 1  a workgroup (4 waves) owns one (sequence, kv-head, split); its waves take alternate
//             32-key tiles and merge their online-softmax states through LDS.
//   HKV  > 1  a workgroup (8 waves) owns one (sequence, split) with ALL its kv-heads: wave =
//             (kv-head, tile parity).  One kv-head's rows are 256 B at a 1-KiB pitch in the cache;
//             the waves of a workgroup move through the tiles together, so the workgroup as a whole
//             reads whole contiguous token rows, which HBM serves at the plain streaming rate
//             (measured with tools/membw: 6.2 TB/s against 5.6 TB/s for one kv-head per workgroup).
// With more than one split the un-normalised partial results go to a workspace and a second kernel
// merges them (letting the workgroup that arrives last merge inside the split kernel was measured
// slower: the agent-scope release / acquire around the ticket costs more than the second launch,
// 23.9 against 19.9 us at 8 sequences, 30.8 against 27.9 us at 16).
#include <cstdint>
#include <cstdlib>
#include <mutex>

#include "common.h"

namespace hymeta {

namespace {

constexpr int kTile = 32;          // keys per tile (two 16-key MFMA sub-tiles)
constexpr int kMaxGroup = 16;      // q-heads per kv-head handled by one workgroup
constexpr float kLog2e = 1.4426950408889634f;

__device__ __forceinline__ f32x4 mfma16(const bf16x8& a, const bf16x8& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ bf16x8 as_frag(const u32x4& v) { return __builtin_bit_cast(bf16x8, v); }

struct SplitGeom {
  int L;        // keys of this sequence
  int kstart;   // first visible real position
  int t0;       // absolute index of the first real tile
  int nreal;    // real tiles
  int ntiles;   // meta tiles + real tiles
  int lo, hi;   // tile range of this split
};

__device__ __forceinline__ SplitGeom split_geom(int L, int window_left, int meta_tiles,
                                                int split, int num_splits) {
  SplitGeom g;
  g.L = L;
  g.kstart = (window_left >= 0 && L - 1 - window_left > 0) ? L - 1 - window_left : 0;
  g.t0 = g.kstart / kTile;
  g.nreal = L > 0 ? (L - 1) / kTile - g.t0 + 1 : 0;
  g.ntiles = meta_tiles + g.nreal;
  const int per = (g.ntiles + num_splits - 1) / num_splits;
  g.lo = split * per;
  g.hi = g.lo + per < g.ntiles ? g.lo + per : g.ntiles;
  return g;
}

// The workgroup-per-sequence shape pays from about 48 sequences up (7B geometry, 4097 keys, us per launch
// incl. the merge of key splits; per-kv-head workgroups / all kv-heads per workgroup, best split count each):
// 16: 31 / 35, 32: 53 / 52, 48: 82 / 72, 64: 100 / 91, 128: 196 / 175, 256: 388 / 336.
inline bool all_heads_shape(int batch, int num_kv_heads) {
  return (num_kv_heads == 2 || num_kv_heads == 4) && batch >= 48;
}

}  // namespace

// HKV == 1: grid.x = batch * num_kv_heads * head_groups; HKV > 1: grid.x = batch.  grid.y = num_splits,
// NW waves per workgroup.
// NT: KV rows are read with the non-temporal hint.  It pays where a workgroup reads whole token rows (all
// kv-heads in the workgroup, or a single kv-head in the cache: 6.7 against 6.2 TB/s) and costs where it
// reads one kv-head's 256-B pieces of wider rows (16 sequences x 4 kv-heads: 31 against 28 us).
template <int NW, int HKV, bool NT>
__global__ __launch_bounds__(NW * 64) void paged_decode_split_kernel(hymeta_paged_decode_args a,
                                                                 int num_splits, int head_groups,
                                                                 float scale_log2) {
  constexpr int D = kHeadDim;
  constexpr int WPH = NW / HKV;    // waves that share one kv-head's tiles
  __shared__ float s_o[NW][kMaxGroup][D];  // per-wave partial outputs
  __shared__ float s_m[NW][kMaxGroup];
  __shared__ float s_l[NW][kMaxGroup];

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
  const int lane = tid % kWave;
  const int c = lane >> 4;   // lane group: k-slot block of the MFMA operands
  const int n = lane & 15;   // MFMA row/column index carried by the lane

  int bid = blockIdx.x;
  const int hg = HKV > 1 ? 0 : bid % head_groups;
  if (HKV == 1) bid /= head_groups;
  const int kvh = HKV > 1 ? wave % HKV : bid % a.num_kv_heads;
  const int b = HKV > 1 ? bid : bid / a.num_kv_heads;
  const int wp = HKV > 1 ? wave / HKV : wave;   // this wave's place among the waves of its kv-head
  const int split = blockIdx.y;
  const int group = a.num_heads / a.num_kv_heads;
  const int head0 = kvh * group + hg * kMaxGroup;        // first q-head of this workgroup
  const int nheads = min(kMaxGroup, group - hg * kMaxGroup);

  const int meta_tiles = a.meta_key != nullptr ? a.num_meta / kTile : 0;
  const SplitGeom geo = split_geom(a.cache_seqlens[b], a.window_left, meta_tiles, split, num_splits);

  // Q as the B operand of S^T = K Q^T: lane holds Q[head n][32*ks + 8c .. +7].
  bf16x8 qf[4];
  {
    const uint16_t* qp = reinterpret_cast<const uint16_t*>(a.q) + b * a.q_stride_b +
                         static_cast<int64_t>(head0 + n) * a.q_stride_h + 8 * c;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      u32x4 raw = {0u, 0u, 0u, 0u};
      if (n < nheads) raw = ld16(qp + 32 * ks);
      qf[ks] = as_frag(raw);
    }
  }

  f32x4 o[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) o[t] = {0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY;  // running max (log2 domain) of head n, replicated over lane groups
  float l_run = 0.f;        // this lane's share of the running denominator

  const uint16_t* kc = reinterpret_cast<const uint16_t*>(a.key_cache) + static_cast<int64_t>(kvh) * a.cache_stride_h;
  const uint16_t* vc = reinterpret_cast<const uint16_t*>(a.value_cache) + static_cast<int64_t>(kvh) * a.cache_stride_h;
  const int32_t* bt = a.block_tables + static_cast<int64_t>(b) * a.block_table_stride;

  // Fused KV write: the workgroup that owns the sequence's last tile (the only reader of the
  // new row; the host uses this form only when one workgroup covers all q-heads of the kv-head)
  // stores the decoded token's K/V row and, instead of waiting for that store to become visible,
  // patches the row into the last tile's registers from the source tensors.
  // The row loads are unconditional (a workgroup without a row to patch re-reads its q) and the
  // store is issued after the tile loop: conditional memory operations ahead of the loop would
  // keep the compiler from counting the loads in flight inside it.
  const bool owner = a.new_key != nullptr && geo.hi == geo.ntiles && geo.hi > geo.lo;
  const int64_t new_slot = owner ? a.slot_mapping[b] : -1;
  const bool patch = new_slot >= 0;
  const uint16_t* qrow = reinterpret_cast<const uint16_t*>(a.q) + b * a.q_stride_b + static_cast<int64_t>(head0) * a.q_stride_h;
  const uint16_t* nk = patch ? reinterpret_cast<const uint16_t*>(a.new_key) + b * a.new_key_stride_b + static_cast<int64_t>(kvh) * a.new_key_stride_h : qrow;
  const uint16_t* nv = patch ? reinterpret_cast<const uint16_t*>(a.new_value) + b * a.new_value_stride_b + static_cast<int64_t>(kvh) * a.new_value_stride_h : qrow;
  u32x4 nkf[4], nvf;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) nkf[ks] = ld16g(nk + 32 * ks + 8 * c);
  nvf = ld16g(nv + 8 * n);

  // K as A-operand rows (16 keys x 64 B per instruction), V as 8 key rows x 16 bytes per lane.
  struct TileRegs {
    u32x4 k[2][4];
    u32x4 v[8];
  };
  // Loads are unconditional: a sub-tile outside the visible range (or a tile past this wave's
  // last) reads the nearest visible sub-tile instead, and the scores / V rows are masked when the
  // tile is consumed.  With conditional issue the compiler could not count the loads in flight
  // and the two-tile register pipeline below would drain at every use.
  const int sub_first = (geo.kstart / 16) * 16;
  const int sub_last = geo.L > 0 ? ((geo.L - 1) / 16) * 16 : 0;
  const int tile_last = geo.hi - 1;
  auto issue = [&](int tile_raw, TileRegs& r) {
    const int tile = min(tile_raw, tile_last);
    const uint16_t* krow[2];
    const uint16_t* vrow[2];
    int64_t row_stride;
    if (tile < meta_tiles) {
      row_stride = a.meta_stride_t;
      const int64_t off = static_cast<int64_t>(tile) * kTile * a.meta_stride_t + static_cast<int64_t>(kvh) * a.meta_stride_h;
      krow[0] = reinterpret_cast<const uint16_t*>(a.meta_key) + off;
      vrow[0] = reinterpret_cast<const uint16_t*>(a.meta_value) + off;
      krow[1] = krow[0] + 16 * row_stride;
      vrow[1] = vrow[0] + 16 * row_stride;
    } else {
      row_stride = a.cache_stride_pos;
      const int pos0 = (geo.t0 + tile - meta_tiles) * kTile;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int p = max(min(pos0 + 16 * s, sub_last), sub_first);
        const int blk = bt[p / a.block_size];
        const int64_t off = static_cast<int64_t>(blk) * a.cache_stride_block + static_cast<int64_t>(p % a.block_size) * a.cache_stride_pos;
        krow[s] = kc + off;
        vrow[s] = vc + off;
      }
    }
    if (NT && tile_raw > tile_last) {
      // past this wave's last tile the loads stay (the compiler counts them) but fetch one hot row:
      // with the non-temporal hint a re-read of the last tile would come from HBM again
      krow[0] = krow[1] = vrow[0] = vrow[1] = qrow;
      row_stride = 0;
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const uint16_t* kp = krow[s] + static_cast<int64_t>(n) * row_stride + 8 * c;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) r.k[s][ks] = NT ? ld16nt(kp + 32 * ks) : ld16g(kp + 32 * ks);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint16_t* vp = vrow[j >> 2] + static_cast<int64_t>(4 * c + (j & 3)) * row_stride + 8 * n;
      r.v[j] = NT ? ld16nt(vp) : ld16g(vp);
    }
  };

  auto consume = [&](int tile, TileRegs& r) {
    if (tile >= geo.hi) return;   // wave-uniform; the registers hold a redundant copy of the last tile
    const bool is_meta = tile < meta_tiles;
    const int pos0 = is_meta ? 0 : (geo.t0 + tile - meta_tiles) * kTile;
    const bool full = is_meta || (pos0 >= geo.kstart && pos0 + kTile <= geo.L);
    if (patch && tile == geo.ntiles - 1) {   // wave-uniform: the tile that holds position L - 1
      const int rn = geo.L - 1 - pos0;       // row of the new token inside the tile
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bool mine = rn == 16 * s + n;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) r.k[s][ks] = mine ? nkf[ks] : r.k[s][ks];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) r.v[j] = (rn == 16 * (j >> 2) + 4 * c + (j & 3)) ? nvf : r.v[j];
    }

    // ---- S^T = K Q^T : st[s][r] = score(key 16s + 4c + r, head n) ----
    f32x4 st[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      st[s] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) st[s] = mfma16(as_frag(r.k[s][ks]), qf[ks], st[s]);
    }

    float sc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) sc[j] = st[j >> 2][j & 3] * scale_log2;
    if (!full) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int p = pos0 + 16 * (j >> 2) + 4 * c + (j & 3);
        const bool ok = p >= geo.kstart && p < geo.L;
        sc[j] = ok ? sc[j] : -INFINITY;
        // Rows outside the visible range may hold anything (NaN included): a zero
        // probability does not neutralise them inside the MFMA, so clear them.
        if (!ok) r.v[j] = u32x4{0u, 0u, 0u, 0u};
      }
    }

    // ---- online softmax over the tile ----
    float tmax = sc[0];
#pragma unroll
    for (int j = 1; j < 8; ++j) tmax = fmaxf(tmax, sc[j]);
    tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    const float m_new = fmaxf(m_run, tmax);
    const float m_use = m_new == -INFINITY ? 0.f : m_new;  // fully masked so far: p = 0
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
    float p[8];
    float psum = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      p[j] = __builtin_amdgcn_exp2f(sc[j] - m_use);
      psum += p[j];
    }
    l_run = l_run * alpha + psum;
    m_run = m_new;

    if (!__all(alpha == 1.f)) {
      // o[t][r] belongs to head 4c + r; its factor lives on the lanes with n == 4c + r.
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const float ar = __shfl(alpha, 4 * c + rr);
#pragma unroll
        for (int t = 0; t < 8; ++t) o[t][rr] *= ar;
      }
    }

    u32x4 pw;
#pragma unroll
    for (int i = 0; i < 4; ++i) pw[i] = pack_bf16x2(p[2 * i], p[2 * i + 1]);
    const bf16x8 pf = as_frag(pw);

    // ---- O += P V : in-register 8x8 transpose of V, then 8 MFMAs ----
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      u32x4 vt;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const uint32_t x = r.v[2 * w][t >> 1];
        const uint32_t y = r.v[2 * w + 1][t >> 1];
        vt[w] = (t & 1) ? __builtin_amdgcn_perm(y, x, 0x07060302u) : __builtin_amdgcn_perm(y, x, 0x05040100u);
      }
      o[t] = mfma16(pf, as_frag(vt), o[t]);
    }
  };

  // Two register sets: the next tile of this wave is in flight while the current one is consumed.
  if (geo.lo + wp < geo.hi) {
    TileRegs ra, rb;
    issue(geo.lo + wp, ra);
    for (int tile = geo.lo + wp; tile < geo.hi; tile += 2 * WPH) {
      issue(tile + WPH, rb);
      __builtin_amdgcn_sched_barrier(0);
      consume(tile, ra);
      issue(tile + 2 * WPH, ra);
      __builtin_amdgcn_sched_barrier(0);
      consume(tile + WPH, rb);
    }
  }

  if (patch && wp == 0 && lane < 32) {   // the new row into the cache (nothing in this launch reads it back)
    const int chunk = lane & 15;
    const bool is_v = lane >= 16;
    uint16_t* dst = const_cast<uint16_t*>(is_v ? vc : kc) + (new_slot / a.block_size) * a.cache_stride_block +
                    (new_slot % a.block_size) * a.cache_stride_pos;
    st16(dst + 8 * chunk, ld16g((is_v ? nv : nk) + 8 * chunk));
  }

  // ---- merge the waves of each kv-head ----
  l_run += __shfl_xor(l_run, 16);
  l_run += __shfl_xor(l_run, 32);
  if (lane < kMaxGroup) {
    s_m[wave][lane] = m_run;
    s_l[wave][lane] = l_run;
  }
#pragma unroll
  for (int t = 0; t < 8; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) s_o[wave][4 * c + r][8 * n + t] = o[t][r];
  }
  __syncthreads();

  float* wo = reinterpret_cast<float*>(a.workspace);
  float* wml = wo + static_cast<int64_t>(a.batch) * a.num_heads * num_splits * D;
  // item = (kv-head of the workgroup, head within the group, 8 channels)
  for (int item = tid; item < HKV * 256; item += NW * 64) {
    const int kk = item >> 8;
    const int hh = (item >> 4) & 15;
    const int d0 = (item & 15) * 8;
    if (hh >= nheads) continue;
    float mw[WPH], m_tot = -INFINITY;
#pragma unroll
    for (int w = 0; w < WPH; ++w) {
      mw[w] = s_m[HKV > 1 ? kk + HKV * w : w][hh];
      m_tot = fmaxf(m_tot, mw[w]);
    }
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    float l_tot = 0.f;
#pragma unroll
    for (int w = 0; w < WPH; ++w) {
      const int ws = HKV > 1 ? kk + HKV * w : w;
      const float f = mw[w] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mw[w] - m_tot);
      l_tot += f * s_l[ws][hh];
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += f * s_o[ws][hh][d0 + i];
    }
    const int head = HKV > 1 ? kk * group + hh : head0 + hh;
    if (num_splits == 1) {
      const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
      u32x4 w;
#pragma unroll
      for (int i = 0; i < 4; ++i) w[i] = pack_bf16x2(acc[2 * i] * inv, acc[2 * i + 1] * inv);
      st16(reinterpret_cast<uint16_t*>(a.out) + b * a.out_stride_b + static_cast<int64_t>(head) * a.out_stride_h + d0, w);
    } else {
      const int64_t unit = (static_cast<int64_t>(b) * a.num_heads + head) * num_splits + split;
      float* po = wo + unit * D + d0;
      *reinterpret_cast<f32x4*>(po) = f32x4{acc[0], acc[1], acc[2], acc[3]};
      *reinterpret_cast<f32x4*>(po + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
      if ((item & 15) == 0) {
        wml[unit * 2] = m_tot;
        wml[unit * 2 + 1] = l_tot;
      }
    }
  }
}

// grid.x = batch * num_heads, 128 threads: one channel per thread.
__global__ __launch_bounds__(128) void paged_decode_merge_kernel(hymeta_paged_decode_args a, int num_splits) {
  constexpr int D = kHeadDim;
  const int d = threadIdx.x;
  const int b = blockIdx.x / a.num_heads;
  const int head = blockIdx.x % a.num_heads;
  const float* wo = reinterpret_cast<const float*>(a.workspace);
  const int64_t unit0 = static_cast<int64_t>(blockIdx.x) * num_splits;
  const float* ml = wo + static_cast<int64_t>(a.batch) * a.num_heads * num_splits * D + unit0 * 2;
  float m_tot = -INFINITY;
  float acc = 0.f, l_tot = 0.f;
  if (num_splits <= 16) {
    // every load of every split in flight at once (a loop over the splits waits one cache round trip per split, twice);
    // the same maxima, factors and sums in the same order as the loops below
    float ms[16], ls[16], os[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int s = min(i, num_splits - 1);
      ms[i] = ml[2 * s], ls[i] = ml[2 * s + 1], os[i] = wo[(unit0 + s) * D + d];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (i < num_splits) m_tot = fmaxf(m_tot, ms[i]);
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (i < num_splits) {
        const float f = ms[i] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(ms[i] - m_tot);
        l_tot += f * ls[i];
        acc += f * os[i];
      }
  } else {
    for (int s = 0; s < num_splits; ++s) m_tot = fmaxf(m_tot, ml[2 * s]);
    for (int s = 0; s < num_splits; ++s) {
      const float ms = ml[2 * s];
      const float f = ms == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(ms - m_tot);
      l_tot += f * ml[2 * s + 1];
      acc += f * wo[(unit0 + s) * D + d];
    }
  }
  const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
  reinterpret_cast<uint16_t*>(a.out)[b * a.out_stride_b + static_cast<int64_t>(head) * a.out_stride_h + d] = f2bf(acc * inv);
}

}  // namespace hymeta

using namespace hymeta;

extern "C" size_t hymeta_meta_swa_paged_decode_workspace_bytes(int32_t batch, int32_t num_heads,
                                                               int32_t head_dim, int32_t num_splits) {
  if (batch <= 0 || num_heads <= 0 || head_dim <= 0 || num_splits <= 1) return 0;
  return static_cast<size_t>(batch) * num_heads * num_splits * (head_dim + 2) * sizeof(float);
}

extern "C" int hymeta_meta_swa_paged_decode_splits(int32_t batch, int32_t num_kv_heads,
                                                   int32_t max_seqlen_k, int32_t window_left,
                                                   int32_t num_meta) {
  if (batch <= 0 || num_kv_heads <= 0) return 1;
  int64_t keys = max_seqlen_k > 0 ? max_seqlen_k : 0;
  if (window_left >= 0 && keys > static_cast<int64_t>(window_left) + 1) keys = static_cast<int64_t>(window_left) + 1;
  const int64_t tiles = (num_meta > 0 ? num_meta / kTile : 0) + (keys + kTile - 1) / kTile + 1;
  if (all_heads_shape(batch, num_kv_heads)) {
    // one round of workgroups: as many key splits as keep (sequences x splits) within the compute units
    // (measured best: 48 -> 5, 64 -> 4, 128 -> 2, 192 and more -> 1), at least 2 tiles per wave
    int64_t want = 256 / batch;
    const int64_t cap = tiles / (2 * (8 / num_kv_heads));
    if (want > cap) want = cap;
    return want < 1 ? 1 : static_cast<int>(want);
  }
  const int64_t units = static_cast<int64_t>(batch) * num_kv_heads;
  // One workgroup per CU already streams at the rate this access pattern allows (measured:
  // 256 workgroups = 5.2 TB/s, more splits only add merge traffic), so split just enough to
  // put a workgroup on every CU.
  int64_t want = (256 + units - 1) / units;
  const int64_t cap = tiles / 8 > 1 ? tiles / 8 : 1;  // at least 8 tiles (2 per wave) per split
  if (want > cap) want = cap;
  if (want < 1) want = 1;
  if (want > 256) want = 256;
  return static_cast<int>(want);
}

namespace {

// HYMETA_PAGED_SHAPE=all|head forces a workgroup shape (A/B timing, tests of both shapes); read once per process.
int forced_shape() {
  static const int shape = [] {
    const char* e = getenv("HYMETA_PAGED_SHAPE");
    return e == nullptr ? 0 : (e[0] == 'a' ? 1 : 2);
  }();
  return shape;
}

// Split-KV partials of hymeta_meta_swa_paged_decode_static: one array per GPU the library runs on, held until
// the process ends.  Nothing in it has to start at zero (every partial the merge kernel reads was written by the
// split kernel of the same call).  In host code the name `scratch` is only a handle for hipGetSymbolAddress:
// the device address is looked up once per device and kept in scratch_ptr[].
static __device__ __align__(256) unsigned char scratch[HYMETA_STATIC_SCRATCH_BYTES];
constexpr int kMaxDevices = 64;
void* scratch_ptr[kMaxDevices] = {};
std::mutex scratch_mutex;

int static_scratch(void** out) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) return HYMETA_ERR_LAUNCH;
  if (dev < 0 || dev >= kMaxDevices) return HYMETA_ERR_UNSUPPORTED;
  std::lock_guard<std::mutex> lock(scratch_mutex);
  if (scratch_ptr[dev] == nullptr) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)) != hipSuccess || p == nullptr) {
      (void)hipGetLastError();
      return HYMETA_ERR_LAUNCH;
    }
    scratch_ptr[dev] = p;
  }
  *out = scratch_ptr[dev];
  return HYMETA_OK;
}

int paged_decode_impl(hymeta_paged_decode_args a, hymeta_stream_t stream, bool use_static_scratch);

}  // namespace

extern "C" int hymeta_meta_swa_paged_decode(const hymeta_paged_decode_args* args, hymeta_stream_t stream) {
  if (!hymeta::args_ok(args)) return HYMETA_ERR_INVALID_ARG;
  return paged_decode_impl(*args, stream, false);
}

extern "C" int hymeta_meta_swa_paged_decode_static(const hymeta_paged_decode_args* args, hymeta_stream_t stream) {
  if (!hymeta::args_ok(args)) return HYMETA_ERR_INVALID_ARG;
  return paged_decode_impl(*args, stream, true);
}

namespace {

int paged_decode_impl(hymeta_paged_decode_args a, hymeta_stream_t stream, bool use_static_scratch) {
  if (a.batch < 0 || a.num_heads <= 0 || a.num_kv_heads <= 0 || a.block_size <= 0) return HYMETA_ERR_INVALID_ARG;
  if (a.batch == 0) return HYMETA_OK;
  if (!a.q || !a.out || !a.key_cache || !a.value_cache || !a.block_tables || !a.cache_seqlens) return HYMETA_ERR_INVALID_ARG;
  if (a.head_dim != kHeadDim) return HYMETA_ERR_UNSUPPORTED;
  if (a.num_heads % a.num_kv_heads != 0) return HYMETA_ERR_INVALID_ARG;
  if (a.block_size % 16 != 0) return HYMETA_ERR_UNSUPPORTED;
  if ((a.meta_key == nullptr) != (a.meta_value == nullptr)) return HYMETA_ERR_INVALID_ARG;
  if (a.meta_key == nullptr) a.num_meta = 0;
  if (a.num_meta < 0 || a.num_meta % kTile != 0) return HYMETA_ERR_UNSUPPORTED;
  const int64_t strides[] = {a.q_stride_b, a.q_stride_h, a.out_stride_b, a.out_stride_h, a.cache_stride_block,
                             a.cache_stride_pos, a.cache_stride_h, a.meta_stride_t, a.meta_stride_h};
  for (int64_t st : strides)
    if (st % 8 != 0) return HYMETA_ERR_UNSUPPORTED;
  uintptr_t ptrs = reinterpret_cast<uintptr_t>(a.q) | reinterpret_cast<uintptr_t>(a.out) |
                   reinterpret_cast<uintptr_t>(a.key_cache) | reinterpret_cast<uintptr_t>(a.value_cache) |
                   reinterpret_cast<uintptr_t>(a.meta_key) | reinterpret_cast<uintptr_t>(a.meta_value);
  if (ptrs % 16 != 0) return HYMETA_ERR_UNSUPPORTED;

  const int max_k = a.max_seqlen_k > 0 ? a.max_seqlen_k : a.max_blocks_per_seq * a.block_size;
  int splits = a.num_splits > 0 ? a.num_splits
                                : hymeta_meta_swa_paged_decode_splits(a.batch, a.num_kv_heads, max_k, a.window_left, a.num_meta);
  if (use_static_scratch) {
    a.workspace = nullptr;
    a.workspace_bytes = 0;
    if (splits > 1) {
      const size_t need = hymeta_meta_swa_paged_decode_workspace_bytes(a.batch, a.num_heads, a.head_dim, splits);
      if (need > HYMETA_STATIC_SCRATCH_BYTES) return HYMETA_ERR_SCRATCH_LIMIT;
      const int rc = static_scratch(&a.workspace);
      if (rc != HYMETA_OK) return rc;
      a.workspace_bytes = HYMETA_STATIC_SCRATCH_BYTES;
    }
  }
  if (splits > 1) {
    const size_t need = hymeta_meta_swa_paged_decode_workspace_bytes(a.batch, a.num_heads, a.head_dim, splits);
    if (a.workspace == nullptr || a.workspace_bytes < need) return HYMETA_ERR_WORKSPACE;
    if (reinterpret_cast<uintptr_t>(a.workspace) % 16 != 0) return HYMETA_ERR_UNSUPPORTED;
  }
  const float scale = a.scale > 0.f ? a.scale : 1.f / sqrtf(static_cast<float>(a.head_dim));
  const int group = a.num_heads / a.num_kv_heads;
  const int head_groups = (group + kMaxGroup - 1) / kMaxGroup;

  hipStream_t s = begin_launches(stream);
  if (a.new_key != nullptr) {
    if (!a.new_value || !a.slot_mapping) return HYMETA_ERR_INVALID_ARG;
    if ((a.new_key_stride_b | a.new_key_stride_h | a.new_value_stride_b | a.new_value_stride_h) % 8 != 0 ||
        (reinterpret_cast<uintptr_t>(a.new_key) | reinterpret_cast<uintptr_t>(a.new_value)) % 16 != 0)
      return HYMETA_ERR_UNSUPPORTED;
    if (head_groups > 1) {
      // several workgroups read the new row: write it with the stand-alone kernel first
      hymeta_reshape_and_cache_args w = hymeta::make_args<hymeta_reshape_and_cache_args>();
      w.key = a.new_key; w.value = a.new_value;
      w.key_cache = const_cast<void*>(a.key_cache); w.value_cache = const_cast<void*>(a.value_cache);
      w.slot_mapping = a.slot_mapping;
      w.num_tokens = a.batch; w.num_kv_heads = a.num_kv_heads; w.head_dim = a.head_dim; w.block_size = a.block_size;
      w.elem_bytes = 2;
      w.num_blocks = INT64_MAX / (a.cache_stride_block > 0 ? a.cache_stride_block : 1) / 4;   // bound unknown here: the caller vouches for its slots
      w.key_stride_t = a.new_key_stride_b; w.key_stride_h = a.new_key_stride_h;
      w.value_stride_t = a.new_value_stride_b; w.value_stride_h = a.new_value_stride_h;
      w.cache_stride_block = a.cache_stride_block; w.cache_stride_pos = a.cache_stride_pos; w.cache_stride_h = a.cache_stride_h;
      const int rc = hymeta_reshape_and_cache(&w, stream);
      if (rc != HYMETA_OK) return rc;
      a.new_key = nullptr;
    }
  }
  bool all_heads = all_heads_shape(a.batch, a.num_kv_heads);
  if (forced_shape() != 0) all_heads = forced_shape() == 1 && (a.num_kv_heads == 2 || a.num_kv_heads == 4);
  all_heads = all_heads && head_groups == 1;
  if (all_heads) {
    dim3 grid(static_cast<unsigned>(a.batch), static_cast<unsigned>(splits));
    if (a.num_kv_heads == 4)
      hipLaunchKernelGGL((paged_decode_split_kernel<8, 4, true>), grid, dim3(512), 0, s, a, splits, 1, scale * kLog2e);
    else
      hipLaunchKernelGGL((paged_decode_split_kernel<8, 2, true>), grid, dim3(512), 0, s, a, splits, 1, scale * kLog2e);
  } else {
    dim3 grid(static_cast<unsigned>(a.batch) * a.num_kv_heads * head_groups, static_cast<unsigned>(splits));
    if (a.num_kv_heads == 1 && a.cache_stride_pos == kHeadDim)   // token rows are contiguous for the one kv-head
      hipLaunchKernelGGL((paged_decode_split_kernel<4, 1, true>), grid, dim3(256), 0, s, a, splits, head_groups, scale * kLog2e);
    else
      hipLaunchKernelGGL((paged_decode_split_kernel<4, 1, false>), grid, dim3(256), 0, s, a, splits, head_groups, scale * kLog2e);
  }
  if (hipGetLastError() != hipSuccess) return HYMETA_ERR_LAUNCH;
  if (splits > 1) {
    hipLaunchKernelGGL(paged_decode_merge_kernel, dim3(static_cast<unsigned>(a.batch) * a.num_heads), dim3(128), 0, s, a, splits);
  }
  return launch_status();
}

}  // namespace
