// GLA single-token decode: S' = bf16(k v^T) + exp(g) * S ; o = (scale q) S'.
//
// HBM-bound: per (sequence, head) the kernel streams the 128x128 state once in
// and once out (64 KiB at bf16) and touches 5 vectors of 128 elements.  One
// 256-thread workgroup owns one (sequence, head, column split): every lane
// issues all of its 16-byte state loads up front (8 x 16 B at NSPLIT = 1, so
// a workgroup keeps 32 KiB in flight), then updates, stores and accumulates
// the q-contraction in registers.  Rows of the state are 256 B, so a wave's
// 16-byte-per-lane access covers whole rows (NSPLIT 1) or 128-B half rows
// (NSPLIT 2): fully coalesced either way.
#include "common.h"
#include "gla_gate.h"

namespace hymeta {

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
    const uint16_t* qp = reinterpret_cast<const uint16_t*>(a.q) + b * a.q_stride_b + h * a.q_stride_h;
    const uint16_t* kp = reinterpret_cast<const uint16_t*>(a.k) + b * a.k_stride_b + kvh * a.k_stride_h;
    float g_raw = 0.f;
    if (a.g != nullptr) {
      const int64_t goff = b * a.g_stride_b + kvh * a.g_stride_h + tid;
      g_raw = a.g_dtype == HYMETA_F32 ? reinterpret_cast<const float*>(a.g)[goff]
                                      : bf2f(reinterpret_cast<const uint16_t*>(a.g)[goff]);
    }
    const float lb = a.lower_bound != nullptr ? a.lower_bound[h * D + tid] : 0.f;
    float qs, kb, g;
    apply_gate(gp, bf2f(qp[tid]), bf2f(kp[tid]), g_raw, lb, qs, kb, g);
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
    const uint16_t* vp = reinterpret_cast<const uint16_t*>(a.v) + b * a.v_stride_b + kvh * a.v_stride_h + col0;
    u32x4 raw = ld16(vp);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      vf[2 * i] = lo_bf(raw[i]);
      vf[2 * i + 1] = hi_bf(raw[i]);
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

}  // namespace hymeta

using namespace hymeta;

extern "C" int hymeta_gla_decode(const hymeta_gla_decode_args* args, hymeta_stream_t stream) {
  if (args == nullptr) return HYMETA_ERR_INVALID_ARG;
  const hymeta_gla_decode_args& a = *args;
  if (a.batch < 0 || a.num_heads <= 0 || a.num_kv_heads <= 0) return HYMETA_ERR_INVALID_ARG;
  if (a.batch == 0) return HYMETA_OK;
  if (!a.q || !a.k || !a.v || !a.out || !a.state || !a.slot_idx) return HYMETA_ERR_INVALID_ARG;
  if (a.head_dim != kHeadDim) return HYMETA_ERR_UNSUPPORTED;
  if (a.num_heads % a.num_kv_heads != 0) return HYMETA_ERR_INVALID_ARG;
  if (a.g_dtype != HYMETA_BF16 && a.g_dtype != HYMETA_F32) return HYMETA_ERR_UNSUPPORTED;
  if (a.state_dtype != HYMETA_BF16 && a.state_dtype != HYMETA_F32) return HYMETA_ERR_UNSUPPORTED;
  if (a.gate_mode < HYMETA_GATE_NONE || a.gate_mode > HYMETA_GATE_70B_LOWER_BOUND) return HYMETA_ERR_UNSUPPORTED;
  if ((a.gate_mode == HYMETA_GATE_NONE || a.gate_mode == HYMETA_GATE_7B) && !a.g) return HYMETA_ERR_INVALID_ARG;
  if (a.gate_mode == HYMETA_GATE_70B_LOWER_BOUND && !a.lower_bound) return HYMETA_ERR_INVALID_ARG;
  // 16-byte vector access on v, out and state rows
  if ((a.v_stride_b | a.v_stride_h | a.state_stride_slot | a.state_stride_h) % 8 != 0) return HYMETA_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(a.v) | reinterpret_cast<uintptr_t>(a.state)) % 16 != 0) return HYMETA_ERR_UNSUPPORTED;

  GateParams gp;
  gp.mode = a.gate_mode;
  gp.scale = a.scale > 0.f ? a.scale : 1.f / sqrtf(static_cast<float>(a.head_dim));
  gp.normalizer = a.gate_normalizer > 0.f ? a.gate_normalizer : 16.f;
  gp.clamp = a.gate_clamp > 0.f ? a.gate_clamp : 0.95f;

  hipStream_t s = begin_launches(stream);
  const int64_t units = static_cast<int64_t>(a.batch) * a.num_heads;
  const bool f32 = a.state_dtype == HYMETA_F32;
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
