// Varlen prefill attention: causal sliding window over the real keys plus an
// always-visible block of meta keys, one softmax (flash-attention forward).
//
// MFMA-bound.  A 256-thread workgroup owns 128 consecutive queries of one
// (sequence, q-head); each wave owns 32 of them as two 16-row MFMA tiles and
// walks the visible keys in tiles of 32.  Operand plumbing is the same as in
// the paged decode kernel, with query rows where the decode kernel has heads:
//   S^T = K Q^T   (A = 16 key rows x 32 channels straight from memory,
//                  B = the wave's resident Q fragments)
//   O  += P V     (A = the S^T accumulators re-packed to bf16: the query is on
//                  the lane and the keys are in registers, which is the
//                  A-operand layout; B = V, 8 key rows x 16 bytes per lane,
//                  transposed in registers with v_perm_b32)
// Keys are read through a block table (chunked / prefix prefill) or from the
// packed [token, kv-head, dim] tensor of the step.
#include "common.h"

namespace hymeta {
namespace {

constexpr int kTile = 32;   // keys per step
constexpr int kRowsPerWave = 32;
constexpr int kRowsPerBlock = 128;
constexpr float kLog2e = 1.4426950408889634f;

__device__ __forceinline__ f32x4 mfma16(const bf16x8& a, const bf16x8& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ bf16x8 as_frag(const u32x4& v) { return __builtin_bit_cast(bf16x8, v); }

}  // namespace

// grid = (ceil(max_seqlen_q / 128), num_seqs (+1: the meta-query rows), num_heads)
template <bool PAGED>
__global__ __launch_bounds__(256) void swa_prefill_kernel(hymeta_swa_prefill_args a, float scale_log2) {
  constexpr int D = kHeadDim;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int c = lane >> 4;
  const int n = lane & 15;

  const int seq = blockIdx.y;
  const int h = blockIdx.z;
  const int kvh = h / (a.num_heads / a.num_kv_heads);
  const bool meta_rows = seq == a.num_seqs;  // this workgroup computes the meta-query rows
  const int meta_tiles = a.meta_key != nullptr ? a.num_meta / kTile : 0;
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
  const int row0 = blockIdx.x * kRowsPerBlock + wave * kRowsPerWave;  // first query of this wave
  if (row0 >= Lq) return;
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
  float m_run[2] = {-INFINITY, -INFINITY};
  float l_run[2] = {0.f, 0.f};

  // Real-key tile range visible to this wave's 32 queries.
  const int last_row = min(row0 + kRowsPerWave, Lq) - 1;
  const int p_lo = row0 + pos_shift, p_hi = last_row + pos_shift;
  int t_lo = 0, t_hi = -1;
  if (!meta_rows && Lk > 0 && p_hi >= 0) {
    const int k_lo = (a.window_left >= 0 && p_lo - a.window_left > 0) ? p_lo - a.window_left : 0;
    const int k_hi = p_hi < Lk - 1 ? p_hi : Lk - 1;
    t_lo = k_lo / kTile;
    t_hi = k_hi / kTile;
  }
  const int ntiles = meta_tiles + (t_hi - t_lo + 1);

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

  for (int tile = 0; tile < ntiles; ++tile) {
    const bool is_meta = tile < meta_tiles;
    const uint16_t* krow[2];
    const uint16_t* vrow[2];
    int64_t kstride, vstride;
    int pos0 = 0;
    bool sub_ok[2] = {true, true};
    if (is_meta) {
      kstride = vstride = a.meta_stride_t;
      const int64_t off = static_cast<int64_t>(tile) * kTile * a.meta_stride_t + static_cast<int64_t>(kvh) * a.meta_stride_h;
      krow[0] = reinterpret_cast<const uint16_t*>(a.meta_key) + off;
      vrow[0] = reinterpret_cast<const uint16_t*>(a.meta_value) + off;
      krow[1] = krow[0] + 16 * kstride;
      vrow[1] = vrow[0] + 16 * vstride;
    } else {
      pos0 = (t_lo + tile - meta_tiles) * kTile;
      if constexpr (PAGED) {
        kstride = vstride = a.cache_stride_pos;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int p = pos0 + 16 * s;
          sub_ok[s] = p < Lk;
          int64_t off = 0;
          if (sub_ok[s]) {
            const int blk = bt[p / a.block_size];
            off = static_cast<int64_t>(blk) * a.cache_stride_block + static_cast<int64_t>(p % a.block_size) * a.cache_stride_pos;
          }
          krow[s] = kglob + off;
          vrow[s] = vglob + off;
        }
      } else {
        kstride = a.k_stride_t;
        vstride = a.v_stride_t;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          krow[s] = kglob + static_cast<int64_t>(pos0 + 16 * s) * kstride;
          vrow[s] = vglob + static_cast<int64_t>(pos0 + 16 * s) * vstride;
        }
      }
    }
    // Every (query, key) pair of the tile visible?  (wave-uniform)
    const bool full = is_meta || (pos0 + kTile - 1 <= p_lo && pos0 + kTile <= Lk &&
                                  (a.window_left < 0 || p_hi - pos0 <= a.window_left) &&
                                  row0 + kRowsPerWave <= Lq);

    u32x4 kraw[2][4];
    u32x4 vraw[8];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      // Un-paged rows past the sequence end belong to the next sequence or lie outside the tensor.
      const bool row_ok = PAGED || is_meta ? sub_ok[s] : (pos0 + 16 * s + n < Lk);
      const uint16_t* kp = krow[s] + static_cast<int64_t>(n) * kstride + 8 * c;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) kraw[s][ks] = row_ok ? ld16(kp + 32 * ks) : u32x4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int s = j >> 2;
      const int kk = 4 * c + (j & 3);
      const bool row_ok = PAGED || is_meta ? sub_ok[s] : (pos0 + 16 * s + kk < Lk);
      const uint16_t* vp = vrow[s] + static_cast<int64_t>(kk) * vstride + 8 * n;
      vraw[j] = row_ok ? ld16(vp) : u32x4{0u, 0u, 0u, 0u};
    }

    // S^T tiles: st[qt][s][r] = score(key 16s + 4c + r, query 16qt + n)
    f32x4 st[2][2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        st[qt][s] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) st[qt][s] = mfma16(as_frag(kraw[s][ks]), qf[qt][ks], st[qt][s]);
      }

    float sc[2][8];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int j = 0; j < 8; ++j) sc[qt][j] = st[qt][j >> 2][j & 3] * scale_log2;

    if (!full) {
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        const int qr = row0 + 16 * qt + n;
        const int qp = qr + pos_shift;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int kp = pos0 + 16 * (j >> 2) + 4 * c + (j & 3);
          const bool ok = qr < Lq && kp <= qp && kp < Lk && (a.window_left < 0 || qp - kp <= a.window_left);
          sc[qt][j] = ok ? sc[qt][j] : -INFINITY;
        }
      }
      if constexpr (PAGED) {
        // Cache rows past the sequence end may hold anything; p = 0 does not neutralise NaN/Inf.
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int kp = pos0 + 16 * (j >> 2) + 4 * c + (j & 3);
          if (kp >= Lk) vraw[j] = u32x4{0u, 0u, 0u, 0u};
        }
      }
    }

    // online softmax per query column
    bf16x8 pf[2];
    float alpha[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      float tmax = sc[qt][0];
#pragma unroll
      for (int j = 1; j < 8; ++j) tmax = fmaxf(tmax, sc[qt][j]);
      tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
      const float m_new = fmaxf(m_run[qt], tmax);
      const float m_use = m_new == -INFINITY ? 0.f : m_new;
      alpha[qt] = __builtin_amdgcn_exp2f(m_run[qt] - m_use);
      float p[8], psum = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        p[j] = __builtin_amdgcn_exp2f(sc[qt][j] - m_use);
        psum += p[j];
      }
      l_run[qt] = l_run[qt] * alpha[qt] + psum;
      m_run[qt] = m_new;
      u32x4 pw;
#pragma unroll
      for (int i = 0; i < 4; ++i) pw[i] = pack_bf16x2(p[2 * i], p[2 * i + 1]);
      pf[qt] = as_frag(pw);
    }
    if (!__all(alpha[0] == 1.f && alpha[1] == 1.f)) {
#pragma unroll
      for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float ar = __shfl(alpha[qt], 4 * c + r);  // o rows are queries 4c + r
#pragma unroll
          for (int t = 0; t < 8; ++t) o[qt][t][r] *= ar;
        }
    }

    // O += P V
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      u32x4 vt;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const uint32_t x = vraw[2 * w][t >> 1], y = vraw[2 * w + 1][t >> 1];
        vt[w] = (t & 1) ? __builtin_amdgcn_perm(y, x, 0x07060302u) : __builtin_amdgcn_perm(y, x, 0x05040100u);
      }
      o[0][t] = mfma16(pf[0], as_frag(vt), o[0][t]);
      o[1][t] = mfma16(pf[1], as_frag(vt), o[1][t]);
    }
  }

  // epilogue: normalise and store.  o[qt][t][r] = O[query 16qt + 4c + r][dim 8n + t]
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    float l = l_run[qt];
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float lr = __shfl(l, 4 * c + r);
      const float inv = lr > 0.f ? 1.f / lr : 0.f;
      const int qr = row0 + 16 * qt + 4 * c + r;
      if (qr < Lq) {
        u32x4 w;
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = pack_bf16x2(o[qt][2 * i][r] * inv, o[qt][2 * i + 1][r] * inv);
        uint16_t* op = reinterpret_cast<uint16_t*>(a.out) + (out_first + qr) * a.out_stride_t +
                       static_cast<int64_t>(h) * a.out_stride_h + 8 * n;
        st16(op, w);
      }
    }
  }
}

}  // namespace hymeta

using namespace hymeta;

extern "C" int hymeta_meta_swa_prefill(const hymeta_swa_prefill_args* args, hymeta_stream_t stream) {
  if (args == nullptr) return HYMETA_ERR_INVALID_ARG;
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
  dim3 grid((max_rows + kRowsPerBlock - 1) / kRowsPerBlock, seqs + (a.meta_query != nullptr ? 1 : 0), a.num_heads);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (paged) hipLaunchKernelGGL((swa_prefill_kernel<true>), grid, dim3(256), 0, s, a, scale * kLog2e);
  else hipLaunchKernelGGL((swa_prefill_kernel<false>), grid, dim3(256), 0, s, a, scale * kLog2e);
  return launch_status();
}
