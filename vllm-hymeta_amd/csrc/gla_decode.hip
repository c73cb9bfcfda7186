// GLA single-token decode: S' = bf16(k v^T) + exp(g) * S ; o = bf16(bf16(scale) q) S'  (the rounding points of
// vllm_hymeta/models/my_fused_recurrent.py:59-76).
//
// HBM-bound: per (sequence, head) the kernel streams the 128x128 state once in
// and once out (64 KiB at bf16) and touches 5 vectors of 128 elements.  One
// 256-thread workgroup owns one (sequence, head, column split): every lane
// issues all of its 16-byte state loads up front (8 x 16 B at NSPLIT = 1, so
// a workgroup keeps 32 KiB in flight), then updates, stores and accumulates
// the q-contraction in registers.  Rows of the state are 256 B, so a wave's
// 16-byte-per-lane access covers whole rows (NSPLIT 1) or 128-B half rows
// (NSPLIT 2): fully coalesced either way.
#include <string.h>

#include "lowrank_gate.h"
#include "common.h"
#include "gla_gate.h"
#include "partial_sums.h"

namespace hymeta {

// The gate given in its low-rank form (7B: gk = W1 (W0 x) + b with a rank of 16): ``g`` then holds the rank-16 row of the
// token (bf16, the first linear's rounded output) and the second linear is applied here, one output feature per thread, in
// the arithmetic of hymeta_lowrank_gate's second stage (fma chain in rank order, bias added, rounded to bf16).
__device__ __forceinline__ hymeta_gemm_partials qkv_partials(const hymeta_gla_decode_args& a) {
  hymeta_gemm_partials p = hymeta::make_args<hymeta_gemm_partials>();
  p.slabs = a.qkv_slabs, p.splits = a.qkv_splits, p.bias = a.qkv_bias, p.M = a.batch, p.N = a.qkv_n;
  p.x_scale = a.qkv_x_scale, p.w_scale = a.qkv_w_scale, p.is_int32 = a.qkv_x_scale != nullptr;
  return p;
}

__device__ __forceinline__ float lowrank_gate_from(const hymeta_gla_decode_args& a, const float* t, int feature) {
  const uint16_t* w1 = reinterpret_cast<const uint16_t*>(a.g_w1) + static_cast<int64_t>(feature) * 16;
  const u32x4 wa = ld16(w1), wb = ld16(w1 + 8);
  float y = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    y = fmaf(t[2 * i], lo_bf(wa[i]), y);
    y = fmaf(t[2 * i + 1], hi_bf(wa[i]), y);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    y = fmaf(t[8 + 2 * i], lo_bf(wb[i]), y);
    y = fmaf(t[9 + 2 * i], hi_bf(wb[i]), y);
  }
  if (a.g_bias != nullptr) y += bf2f(reinterpret_cast<const uint16_t*>(a.g_bias)[feature]);
  return bf16_round(y);
}

__device__ __forceinline__ float lowrank_gate_value(const hymeta_gla_decode_args& a, int b, int feature) {
  float t[16];
  if (a.qkv_slabs != nullptr) {
    const hymeta_gemm_partials p = qkv_partials(a);
    float lo[8], hi[8];
    partial_row8<false>(p, b, a.g_col, lo);
    partial_row8<false>(p, b, a.g_col + 8, hi);
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = lo[i], t[8 + i] = hi[i];
  } else {
    const uint16_t* tp = reinterpret_cast<const uint16_t*>(a.g) + b * a.g_stride_b;
    const u32x4 ta = ld16(tp), tb = ld16(tp + 8);
#pragma unroll
    for (int i = 0; i < 4; ++i) t[2 * i] = lo_bf(ta[i]), t[2 * i + 1] = hi_bf(ta[i]), t[8 + 2 * i] = lo_bf(tb[i]), t[9 + 2 * i] = hi_bf(tb[i]);
  }
  return lowrank_gate_from(a, t, feature);
}

template <typename StateT>
struct StateIO;

template <>
struct StateIO<uint16_t> {  // bf16 state: 8 elements = one 16-byte access
  __device__ static void load8(const uint16_t* p, u32x4& raw) { raw = ld16(p); }
  __device__ static void unpack(const u32x4& raw, float (&s)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      s[2 * i] = lo_bf(raw[i]);
      s[2 * i + 1] = hi_bf(raw[i]);
    }
  }
  __device__ static void store8(uint16_t* p, const float (&s)[8]) {
    u32x4 w;
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = pack_bf16x2(s[2 * i], s[2 * i + 1]);
    st16(p, w);
  }
};

template <int NSPLIT, bool STATE_F32>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(7, 8))) void gla_decode_kernel(hymeta_gla_decode_args a, GateParams gp) {
  constexpr int D = kHeadDim;
  constexpr int COLS = D / NSPLIT;      // state columns this workgroup owns
  constexpr int CG = COLS / 8;          // 8-column groups per row
  constexpr int RPP = 256 / CG;         // rows covered per pass
  constexpr int NPASS = D / RPP;

  __shared__ float s_q[D];
  __shared__ float s_k[D];
  __shared__ float s_dec[D];
  __shared__ float s_red[4][COLS];
  __shared__ float s_ssq[2];

  const int tid = threadIdx.x;
  const int b = blockIdx.x / a.num_heads;
  const int h = blockIdx.x % a.num_heads;
  const int split = blockIdx.y;
  const int slot = a.slot_idx[b];
  if (slot < 0) return;  // padded sequence: state and output untouched
  const int kvh = h / (a.num_heads / a.num_kv_heads);

  if (tid < D) {
    const bool from_slabs = a.qkv_slabs != nullptr;
    const hymeta_gemm_partials qp_ = qkv_partials(a);
    const uint16_t* qp = reinterpret_cast<const uint16_t*>(a.q) + b * a.q_stride_b + h * a.q_stride_h;
    const uint16_t* kp = reinterpret_cast<const uint16_t*>(a.k) + b * a.k_stride_b + kvh * a.k_stride_h;
    const float q_raw = from_slabs ? partial_elem(qp_, b, a.q_col + h * D + tid) : bf2f(qp[tid]);
    const float k_raw1 = from_slabs ? partial_elem(qp_, b, a.k_col + kvh * D + tid) : bf2f(kp[tid]);
    float g_raw = 0.f;
    if (a.g_w1 != nullptr) {
      g_raw = lowrank_gate_value(a, b, kvh * D + tid);
    } else if (a.g != nullptr) {
      const int64_t goff = b * a.g_stride_b + kvh * a.g_stride_h + tid;
      g_raw = a.g_dtype == HYMETA_F32 ? reinterpret_cast<const float*>(a.g)[goff]
                                      : bf2f(reinterpret_cast<const uint16_t*>(a.g)[goff]);
    }
    const float lb = a.lower_bound != nullptr ? a.lower_bound[h * D + tid] : 0.f;
    float qs, kb, g;
    apply_gate(gp, q_raw, k_raw1, g_raw, lb, qs, kb, g);
    s_q[tid] = qs;
    s_k[tid] = kb;
    s_dec[tid] = __expf(g);
  }

  const int cg = tid % CG;
  const int r0 = tid / CG;
  const int col0 = split * COLS + cg * 8;

  // v for this lane's 8 columns
  float vf[8];
  {
    if (a.qkv_slabs != nullptr) {
      partial_row8<false>(qkv_partials(a), b, a.v_col + kvh * D + col0, vf);
    } else {
      const uint16_t* vp = reinterpret_cast<const uint16_t*>(a.v) + b * a.v_stride_b + kvh * a.v_stride_h + col0;
      u32x4 raw = ld16(vp);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        vf[2 * i] = lo_bf(raw[i]);
        vf[2 * i + 1] = hi_bf(raw[i]);
      }
    }
  }

  const int64_t sbase = static_cast<int64_t>(slot) * a.state_stride_slot + static_cast<int64_t>(h) * a.state_stride_h;
  float acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) acc[c] = 0.f;

  if constexpr (!STATE_F32) {
    // Rows go through in batches of 2 passes with the next batch's loads issued as soon as a
    // register is consumed; the batch loop is kept rolled so that the kernel stays within 72
    // VGPRs: 7 workgroups per CU, i.e. the 1792 (sequence, head) units of a 64-sequence 7B step
    // are resident in ONE round instead of 1.75.
    constexpr int PB = NPASS < 2 ? NPASS : 2;
    uint16_t* sp = reinterpret_cast<uint16_t*>(a.state) + sbase + col0;
    u32x4 raw[PB];
#pragma unroll
    for (int i = 0; i < PB; ++i) raw[i] = ld16(sp + (r0 + RPP * i) * D);
    __syncthreads();
#pragma unroll 1
    for (int hb = 0; hb < NPASS; hb += PB) {
#pragma unroll
      for (int i = 0; i < PB; ++i) {
        const int row = r0 + RPP * (hb + i);
        const float kr = s_k[row], dr = s_dec[row], qr = s_q[row];
        float s[8];
        StateIO<uint16_t>::unpack(raw[i], s);
        // next batch, same register (the last batch re-reads its own rows: unconditional issue)
        raw[i] = ld16(sp + min(row + RPP * PB, r0 + RPP * (NPASS - 1)) * D);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const float kv = bf16_round(kr * vf[c]);  // outer product is formed in bf16
          s[c] = fmaf(dr, s[c], kv);
          acc[c] = fmaf(qr, s[c], acc[c]);
        }
        StateIO<uint16_t>::store8(sp + row * D, s);
      }
    }
  } else {
    float* sp = reinterpret_cast<float*>(a.state) + sbase + col0;
    f32x4 raw[NPASS][2];
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      const float* p = sp + (r0 + RPP * i) * D;
      raw[i][0] = *reinterpret_cast<const f32x4*>(p);
      raw[i][1] = *reinterpret_cast<const f32x4*>(p + 4);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      const int row = r0 + RPP * i;
      const float kr = s_k[row], dr = s_dec[row], qr = s_q[row];
      f32x4 o0, o1;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const float kv = bf16_round(kr * vf[c]);
        const float sn = fmaf(dr, c < 4 ? raw[i][0][c] : raw[i][1][c - 4], kv);
        acc[c] = fmaf(qr, sn, acc[c]);
        if (c < 4) o0[c] = sn; else o1[c - 4] = sn;
      }
      float* p = sp + row * D;
      *reinterpret_cast<f32x4*>(p) = o0;
      *reinterpret_cast<f32x4*>(p + 4) = o1;
    }
  }

  // Sum the partial contractions of lanes that share a column group.
#pragma unroll
  for (int off = CG; off < kWave; off <<= 1) {
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] += __shfl_xor(acc[c], off);
  }
  const int wave = tid / kWave, lane = tid % kWave;
  if (lane < CG) {
#pragma unroll
    for (int c = 0; c < 8; ++c) s_red[wave][lane * 8 + c] = acc[c];
  }
  __syncthreads();

  uint16_t* op = reinterpret_cast<uint16_t*>(a.out) + b * a.out_stride_b + h * a.out_stride_h + split * COLS;
  if (a.out_norm_weight == nullptr || NSPLIT != 1) {
    if (tid < COLS) {
      const float o = s_red[0][tid] + s_red[1][tid] + s_red[2][tid] + s_red[3][tid];
      op[tid] = f2bf(o);
    }
    return;
  }
  // Per-head RMS norm over the 128 output channels (g_norm), applied to the
  // bf16-rounded output the way the unfused layer sees it.
  float ob = 0.f;
  if (tid < D) {
    ob = bf16_round(s_red[0][tid] + s_red[1][tid] + s_red[2][tid] + s_red[3][tid]);
    float ss = ob * ob;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) ss += __shfl_xor(ss, off);
    if (lane == 0) s_ssq[wave] = ss;
  }
  __syncthreads();
  if (tid < D) {
    const float var = (s_ssq[0] + s_ssq[1]) * (1.f / D);
    const float y = bf16_round(ob * rsqrtf(var + a.out_norm_eps));
    const float w = bf2f(reinterpret_cast<const uint16_t*>(a.out_norm_weight)[tid]);
    op[tid] = f2bf(y * w);
  }
}


// One state per key/value head.  The REP query heads of a group see the same k, v and decay, so
// their states are equal for ever once they start equal; the group's state is streamed once and
// contracted with all REP queries.  Rows, lanes and summation order are those of the kernel above,
// so out is bit-identical to REP per-query-head units.
// NT: the state rows are read with the non-temporal hint (256 sequences x 4 groups: 23.7 -> 21.3-22.2 us;
// at 64 sequences it is level to slightly behind, so small launches keep plain loads).
// GATE_X: the first linear of the low-rank gate runs in this kernel (a.g_w0); a variant of its own because its 32 weight
// loads in flight hold 128 registers and would halve the workgroups a compute unit holds for every other caller.
template <int REP, int NSPLIT, bool STATE_F32, int THREADS, bool NT, bool GATE_X>
__global__ __launch_bounds__(THREADS) void gla_decode_group_kernel(hymeta_gla_decode_args a, GateParams gp) {
  constexpr int D = kHeadDim;
  constexpr int COLS = D / NSPLIT;
  constexpr int CG = COLS / 8;
  constexpr int RPP = THREADS / CG;
  constexpr int NPASS = D / RPP;
  constexpr int NW = THREADS / kWave;

  __shared__ float s_q[REP][D];
  __shared__ float s_k[D];
  __shared__ float s_dec[D];
  __shared__ float s_red[NW][REP][COLS];
  __shared__ float s_in[REP + 2][D];     // inputs taken from the projection's partial sums: q heads, k, v
  __shared__ float s_t[16];              // and the gate's rank-16 row

  const int tid = threadIdx.x;
  const int b = blockIdx.x / a.num_kv_heads;
  const int kvh = blockIdx.x % a.num_kv_heads;
  const int split = blockIdx.y;
  const int slot = a.slot_idx[b];
  if (slot < 0) return;  // padded sequence: state and output untouched
  const int h0 = kvh * REP;

  const int cg = tid % CG;
  const int r0 = tid / CG;
  const int col0 = split * COLS + cg * 8;

  const int64_t sbase = static_cast<int64_t>(slot) * a.state_stride_slot + static_cast<int64_t>(kvh) * a.state_stride_h;
  // the state rows first: every load of the lane's rows is in flight before anything else is fetched, so that the
  // q / k / gate prologue below (loads, exponentials) runs under their latency instead of in front of it
  u32x4 raw[NPASS][STATE_F32 ? 2 : 1];
  char* sp = reinterpret_cast<char*>(a.state) + (sbase + col0) * (STATE_F32 ? 4 : 2);
#pragma unroll
  for (int i = 0; i < NPASS; ++i) {
    const char* p = sp + static_cast<int64_t>(r0 + RPP * i) * D * (STATE_F32 ? 4 : 2);
    raw[i][0] = NT ? ld16nt(p) : ld16(p);
    if constexpr (STATE_F32) raw[i][1] = NT ? ld16nt(p + 16) : ld16(p + 16);
  }

  const bool from_slabs = a.qkv_slabs != nullptr;
  if (from_slabs) {
    // the whole workgroup sums the slices of its (REP + 2) x 128 input values (+ the 16 rank columns) in 4-column pieces
    // (with the gate's first linear in this kernel the projection carries no rank columns)
    const hymeta_gemm_partials qp_ = qkv_partials(a);
    constexpr int kPer = D / 4;
    for (int item = tid; item < (REP + 2) * kPer + (GATE_X ? 0 : 4); item += THREADS) {
      const int row = item / kPer, c4 = item % kPer;
      int col;
      float* dst;
      if (row < REP) col = a.q_col + (h0 + row) * D + 4 * c4, dst = &s_in[row][4 * c4];
      else if (row == REP) col = a.k_col + kvh * D + 4 * c4, dst = &s_in[REP][4 * c4];
      else if (row == REP + 1) col = a.v_col + kvh * D + 4 * c4, dst = &s_in[REP + 1][4 * c4];
      else col = a.g_col + 4 * c4, dst = &s_t[4 * c4];
      *reinterpret_cast<f32x4*>(dst) = qp_.is_int32 ? partial_row4<true>(qp_, b, col) : partial_row4<false>(qp_, b, col);
    }
    __syncthreads();
  }

  // both linears of the low-rank gate here: g is the layer's input row, the first linear's 16 values land in s_t
  constexpr bool gate_from_x = GATE_X;
  if constexpr (GATE_X)
    lowrank_first_linear(reinterpret_cast<const uint16_t*>(a.g) + b * a.g_stride_b, reinterpret_cast<const uint16_t*>(a.g_w0), a.g_hidden,
                         reinterpret_cast<float(*)[16]>(&s_red[0][0][0]), s_t);

  if (tid < D) {
    const uint16_t* kp = reinterpret_cast<const uint16_t*>(a.k) + b * a.k_stride_b + kvh * a.k_stride_h;
    float g_raw = 0.f;
    if (a.g_w1 != nullptr) {
      g_raw = (from_slabs || gate_from_x) ? lowrank_gate_from(a, s_t, kvh * D + tid) : lowrank_gate_value(a, b, kvh * D + tid);
    } else if (a.g != nullptr) {
      const int64_t goff = b * a.g_stride_b + kvh * a.g_stride_h + tid;
      g_raw = a.g_dtype == HYMETA_F32 ? reinterpret_cast<const float*>(a.g)[goff]
                                      : bf2f(reinterpret_cast<const uint16_t*>(a.g)[goff]);
    }
    const float k_raw = from_slabs ? s_in[REP][tid] : bf2f(kp[tid]);
    float q_raw[REP];
#pragma unroll
    for (int r = 0; r < REP; ++r) {
      const uint16_t* qp = reinterpret_cast<const uint16_t*>(a.q) + b * a.q_stride_b + (h0 + r) * a.q_stride_h;
      q_raw[r] = from_slabs ? s_in[r][tid] : bf2f(qp[tid]);
    }
#pragma unroll
    for (int r = 0; r < REP; ++r) {
      float qs, kb, g;
      apply_gate(gp, q_raw[r], k_raw, g_raw, 0.f, qs, kb, g);
      s_q[r][tid] = qs;
      if (r == 0) {
        s_k[tid] = kb;
        s_dec[tid] = __expf(g);
      }
    }
  }

  float vf[8];
  {
    if (from_slabs) {
#pragma unroll
      for (int c = 0; c < 8; ++c) vf[c] = s_in[REP + 1][col0 + c];
    } else {
      const uint16_t* vp = reinterpret_cast<const uint16_t*>(a.v) + b * a.v_stride_b + kvh * a.v_stride_h + col0;
      u32x4 raw = ld16(vp);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        vf[2 * i] = lo_bf(raw[i]);
        vf[2 * i + 1] = hi_bf(raw[i]);
      }
    }
  }

  float acc[REP][8];
#pragma unroll
  for (int r = 0; r < REP; ++r)
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[r][c] = 0.f;

  __syncthreads();
#pragma unroll
  for (int i = 0; i < NPASS; ++i) {
    const int row = r0 + RPP * i;
    const float kr = s_k[row], dr = s_dec[row];
    float s[8];
    if constexpr (STATE_F32) {
#pragma unroll
      for (int c = 0; c < 8; ++c) s[c] = __uint_as_float(raw[i][c >> 2][c & 3]);
    } else {
      StateIO<uint16_t>::unpack(raw[i][0], s);
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float kv = bf16_round(kr * vf[c]);  // outer product is formed in bf16
      s[c] = fmaf(dr, s[c], kv);
    }
    char* p = sp + static_cast<int64_t>(row) * D * (STATE_F32 ? 4 : 2);
    if constexpr (STATE_F32) {
      st16(p, u32x4{__float_as_uint(s[0]), __float_as_uint(s[1]), __float_as_uint(s[2]), __float_as_uint(s[3])});
      st16(p + 16, u32x4{__float_as_uint(s[4]), __float_as_uint(s[5]), __float_as_uint(s[6]), __float_as_uint(s[7])});
    } else {
      StateIO<uint16_t>::store8(reinterpret_cast<uint16_t*>(p), s);
    }
#pragma unroll
    for (int r = 0; r < REP; ++r) {
      const float qr = s_q[r][row];
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[r][c] = fmaf(qr, s[c], acc[r][c]);
    }
  }

  // Sum over the lanes of a wave that share a column group.  Each step hands half of the lane's
  // columns to the partner lane and keeps the other half: the same pairs are added as in a full
  // butterfly, with half the exchanges per step.
  const int wave = tid / kWave, lane = tid % kWave;
  float a4[REP][4], a2[REP][2];
  int cbase = 0;
  {
    const bool hi = (lane & CG) != 0;
#pragma unroll
    for (int r = 0; r < REP; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float send = hi ? acc[r][c] : acc[r][c + 4];
        const float keep = hi ? acc[r][c + 4] : acc[r][c];
        a4[r][c] = keep + __shfl_xor(send, CG);
      }
    cbase += hi ? 4 : 0;
  }
  {
    const bool hi = (lane & (2 * CG)) != 0;
#pragma unroll
    for (int r = 0; r < REP; ++r)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const float send = hi ? a4[r][c] : a4[r][c + 2];
        const float keep = hi ? a4[r][c + 2] : a4[r][c];
        a2[r][c] = keep + __shfl_xor(send, 2 * CG);
      }
    cbase += hi ? 2 : 0;
  }
  if constexpr (CG == 8) {
    const bool hi = (lane & 32) != 0;
#pragma unroll
    for (int r = 0; r < REP; ++r) {
      const float send = hi ? a2[r][0] : a2[r][1];
      const float keep = hi ? a2[r][1] : a2[r][0];
      s_red[wave][r][cg * 8 + cbase + (hi ? 1 : 0)] = keep + __shfl_xor(send, 32);
    }
  } else {
#pragma unroll
    for (int r = 0; r < REP; ++r) {
      s_red[wave][r][cg * 8 + cbase] = a2[r][0];
      s_red[wave][r][cg * 8 + cbase + 1] = a2[r][1];
    }
  }
  __syncthreads();

  if (a.out_norm_weight == nullptr || NSPLIT != 1) {
    for (int idx = tid; idx < REP * COLS; idx += THREADS) {
      const int r = idx / COLS, c = idx % COLS;
      float o = s_red[0][r][c];
#pragma unroll
      for (int w = 1; w < NW; ++w) o += s_red[w][r][c];
      reinterpret_cast<uint16_t*>(a.out)[b * a.out_stride_b + (h0 + r) * a.out_stride_h + split * COLS + c] = f2bf(o);
    }
    return;
  }
  if constexpr (NSPLIT == 1) {
    // Per-head RMS norm over the 128 output channels (g_norm), applied to the bf16-rounded
    // output the way the unfused layer sees it; a wave takes heads wave, wave + NW, ...
    const float w0 = bf2f(reinterpret_cast<const uint16_t*>(a.out_norm_weight)[lane]);
    const float w1 = bf2f(reinterpret_cast<const uint16_t*>(a.out_norm_weight)[lane + kWave]);
    for (int r = wave; r < REP; r += NW) {
      float o0 = s_red[0][r][lane], o1 = s_red[0][r][lane + kWave];
#pragma unroll
      for (int w = 1; w < NW; ++w) {
        o0 += s_red[w][r][lane];
        o1 += s_red[w][r][lane + kWave];
      }
      o0 = bf16_round(o0);
      o1 = bf16_round(o1);
      float ss0 = o0 * o0, ss1 = o1 * o1;
#pragma unroll
      for (int off = 1; off < kWave; off <<= 1) {
        ss0 += __shfl_xor(ss0, off);
        ss1 += __shfl_xor(ss1, off);
      }
      const float rs = rsqrtf((ss0 + ss1) * (1.f / D) + a.out_norm_eps);
      uint16_t* op = reinterpret_cast<uint16_t*>(a.out) + b * a.out_stride_b + (h0 + r) * a.out_stride_h;
      op[lane] = f2bf(bf16_round(o0 * rs) * w0);
      op[lane + kWave] = f2bf(bf16_round(o1 * rs) * w1);
    }
  }
}

}  // namespace hymeta

using namespace hymeta;

namespace {

template <int REP, bool GATE_X>
void launch_group_variant(const hymeta_gla_decode_args& a, const GateParams& gp, bool split2, bool f32, hipStream_t s) {
  const int64_t groups = static_cast<int64_t>(a.batch) * a.num_kv_heads;
  dim3 grid(static_cast<unsigned>(groups), split2 ? 2 : 1);
  if (split2) {
    if (f32) hipLaunchKernelGGL((gla_decode_group_kernel<REP, 2, true, 256, false, GATE_X>), grid, dim3(256), 0, s, a, gp);
    else hipLaunchKernelGGL((gla_decode_group_kernel<REP, 2, false, 256, false, GATE_X>), grid, dim3(256), 0, s, a, gp);
  } else if (groups > 512) {   // more than two workgroups per compute unit: the launch is bandwidth-bound
    if (f32) hipLaunchKernelGGL((gla_decode_group_kernel<REP, 1, true, 256, true, GATE_X>), grid, dim3(256), 0, s, a, gp);
    else hipLaunchKernelGGL((gla_decode_group_kernel<REP, 1, false, 256, true, GATE_X>), grid, dim3(256), 0, s, a, gp);
  } else {   // (512-thread workgroups measured level with 256 at 64 sequences: 9.1 against 9.35 us)
    if (f32) hipLaunchKernelGGL((gla_decode_group_kernel<REP, 1, true, 256, false, GATE_X>), grid, dim3(256), 0, s, a, gp);
    else hipLaunchKernelGGL((gla_decode_group_kernel<REP, 1, false, 256, false, GATE_X>), grid, dim3(256), 0, s, a, gp);
  }
}

template <int REP>
void launch_group(const hymeta_gla_decode_args& a, const GateParams& gp, bool split2, bool f32, hipStream_t s) {
  if (a.g_w0 != nullptr) launch_group_variant<REP, true>(a, gp, split2, f32, s);
  else launch_group_variant<REP, false>(a, gp, split2, f32, s);
}

}  // namespace

extern "C" int hymeta_gla_decode(const hymeta_gla_decode_args* args, hymeta_stream_t stream) {
  if (!hymeta::args_ok(args)) return HYMETA_ERR_INVALID_ARG;
  const hymeta_gla_decode_args& a = *args;
  if (a.batch < 0 || a.num_heads <= 0 || a.num_kv_heads <= 0) return HYMETA_ERR_INVALID_ARG;
  if (a.batch == 0) return HYMETA_OK;
  const bool from_slabs = a.qkv_slabs != nullptr;
  if (from_slabs) {
    if (a.qkv_splits < 1 || a.qkv_n <= 0 || a.qkv_n % 8 != 0 || (a.q_col | a.k_col | a.v_col | a.g_col) % 8 != 0) return HYMETA_ERR_INVALID_ARG;
    if (a.q_col < 0 || a.q_col + a.num_heads * a.head_dim > a.qkv_n || a.k_col < 0 || a.k_col + a.num_kv_heads * a.head_dim > a.qkv_n ||
        a.v_col < 0 || a.v_col + a.num_kv_heads * a.head_dim > a.qkv_n)
      return HYMETA_ERR_INVALID_ARG;
    if (a.g_w1 != nullptr && a.g_w0 == nullptr && (a.g_col < 0 || a.g_col + 16 > a.qkv_n)) return HYMETA_ERR_INVALID_ARG;
    if ((a.qkv_x_scale == nullptr) != (a.qkv_w_scale == nullptr)) return HYMETA_ERR_INVALID_ARG;
    if ((a.gate_mode == HYMETA_GATE_NONE || a.gate_mode == HYMETA_GATE_7B) && !a.g_w1) return HYMETA_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(a.qkv_slabs) | reinterpret_cast<uintptr_t>(a.qkv_bias)) % 16 != 0) return HYMETA_ERR_UNSUPPORTED;
    if (!a.out || !a.state || !a.slot_idx) return HYMETA_ERR_INVALID_ARG;
  } else if (!a.q || !a.k || !a.v || !a.out || !a.state || !a.slot_idx) {
    return HYMETA_ERR_INVALID_ARG;
  }
  if (a.head_dim != kHeadDim) return HYMETA_ERR_UNSUPPORTED;
  if (a.num_heads % a.num_kv_heads != 0) return HYMETA_ERR_INVALID_ARG;
  if (a.g_dtype != HYMETA_BF16 && a.g_dtype != HYMETA_F32) return HYMETA_ERR_UNSUPPORTED;
  if (a.state_dtype != HYMETA_BF16 && a.state_dtype != HYMETA_F32) return HYMETA_ERR_UNSUPPORTED;
  if (a.gate_mode < HYMETA_GATE_NONE || a.gate_mode > HYMETA_GATE_70B_LOWER_BOUND) return HYMETA_ERR_UNSUPPORTED;
  if (!from_slabs && (a.gate_mode == HYMETA_GATE_NONE || a.gate_mode == HYMETA_GATE_7B) && !a.g) return HYMETA_ERR_INVALID_ARG;
  if (a.gate_mode == HYMETA_GATE_70B_LOWER_BOUND && !a.lower_bound) return HYMETA_ERR_INVALID_ARG;
  if (a.g_w1 != nullptr && !from_slabs) {      // low-rank gate: g = bf16 [batch, 16] rows, 16-byte loads
    if (!a.g || a.g_dtype != HYMETA_BF16 || a.g_stride_b % 8 != 0) return HYMETA_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(a.g) | reinterpret_cast<uintptr_t>(a.g_w1)) % 16 != 0) return HYMETA_ERR_UNSUPPORTED;
  }
  if (a.g_w0 != nullptr) {      // both linears in the kernel: g = the layer input, bf16 [batch, g_hidden]
    if (!a.g_w1 || !a.g || a.g_dtype != HYMETA_BF16 || a.g_stride_b % 8 != 0 || a.g_hidden <= 0 || a.g_hidden % 8 != 0) return HYMETA_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(a.g_w0) | reinterpret_cast<uintptr_t>(a.g) | reinterpret_cast<uintptr_t>(a.g_w1)) % 16 != 0) return HYMETA_ERR_UNSUPPORTED;
    const int sh = a.state_heads == 0 ? a.num_heads : a.state_heads;
    if (sh == a.num_heads) return HYMETA_ERR_UNSUPPORTED;   // the kernels of the state per key/value head only
  }
  // 16-byte vector access on v, out and state rows
  if (((from_slabs ? 0 : (a.v_stride_b | a.v_stride_h)) | a.state_stride_slot | a.state_stride_h) % 8 != 0) return HYMETA_ERR_UNSUPPORTED;
  if (((from_slabs ? 0 : reinterpret_cast<uintptr_t>(a.v)) | reinterpret_cast<uintptr_t>(a.state)) % 16 != 0) return HYMETA_ERR_UNSUPPORTED;

  GateParams gp;
  gp.mode = a.gate_mode;
  // the reference multiplies q by the scale in bf16: the scale enters rounded to bf16 (round to nearest even)
  const float scale = a.scale > 0.f ? a.scale : 1.f / sqrtf(static_cast<float>(a.head_dim));
  uint32_t scale_bits;
  memcpy(&scale_bits, &scale, sizeof(scale_bits));
  scale_bits = (scale_bits + 0x7fffu + ((scale_bits >> 16) & 1u)) & 0xffff0000u;
  memcpy(&gp.scale, &scale_bits, sizeof(scale_bits));
  gp.normalizer = a.gate_normalizer > 0.f ? a.gate_normalizer : 16.f;
  gp.clamp = a.gate_clamp > 0.f ? a.gate_clamp : 0.95f;

  const int state_heads = a.state_heads == 0 ? a.num_heads : a.state_heads;
  if (state_heads != a.num_heads && state_heads != a.num_kv_heads) return HYMETA_ERR_INVALID_ARG;
  const int rep = a.num_heads / a.num_kv_heads;
  const bool grouped = state_heads != a.num_heads;
  // a state shared by a group needs a gate that is the same for every query head of the group
  if (grouped && a.gate_mode == HYMETA_GATE_70B_LOWER_BOUND) return HYMETA_ERR_INVALID_ARG;
  if (grouped && rep > 8) return HYMETA_ERR_UNSUPPORTED;
  if (!grouped && from_slabs && a.qkv_x_scale != nullptr) return HYMETA_ERR_UNSUPPORTED;      // int32 slabs: the grouped kernels only
  if (from_slabs && a.qkv_x_scale != nullptr && reinterpret_cast<uintptr_t>(a.qkv_w_scale) % 16 != 0) return HYMETA_ERR_UNSUPPORTED;

  hipStream_t s = begin_launches(stream);
  const bool f32 = a.state_dtype == HYMETA_F32;
  if (grouped) {
    const bool split2 = static_cast<int64_t>(a.batch) * a.num_kv_heads < 512 && a.out_norm_weight == nullptr;
    switch (rep) {
      case 2: launch_group<2>(a, gp, split2, f32, s); break;
      case 3: launch_group<3>(a, gp, split2, f32, s); break;
      case 4: launch_group<4>(a, gp, split2, f32, s); break;
      case 5: launch_group<5>(a, gp, split2, f32, s); break;
      case 6: launch_group<6>(a, gp, split2, f32, s); break;
      case 7: launch_group<7>(a, gp, split2, f32, s); break;
      default: launch_group<8>(a, gp, split2, f32, s); break;
    }
    return launch_status();
  }
  const int64_t units = static_cast<int64_t>(a.batch) * a.num_heads;
  // Few (sequence, head) units: split the state columns over two workgroups so
  // more CUs stream.  The fused norm needs all 128 columns in one workgroup.
  const bool split2 = units < 1024 && a.out_norm_weight == nullptr;
  dim3 block(256);
  if (split2) {
    dim3 grid(static_cast<unsigned>(units), 2);
    if (f32) hipLaunchKernelGGL((gla_decode_kernel<2, true>), grid, block, 0, s, a, gp);
    else hipLaunchKernelGGL((gla_decode_kernel<2, false>), grid, block, 0, s, a, gp);
  } else {
    dim3 grid(static_cast<unsigned>(units), 1);
    if (f32) hipLaunchKernelGGL((gla_decode_kernel<1, true>), grid, block, 0, s, a, gp);
    else hipLaunchKernelGGL((gla_decode_kernel<1, false>), grid, block, 0, s, a, gp);
  }
  return launch_status();
}
