// GLA chunked prefill, all sequences of a step in one launch.
//
// Recurrence per (sequence, head): S_t = diag(exp g_t) S_{t-1} + k_t^T v_t,
// o_t = (scale q_t) S_t, state S is 128 x 128.
//
// One 256-thread workgroup walks one (sequence, head) in chunks of 32 tokens
// (two 16-token sub-chunks); wave w keeps the fp32 state columns
// [32w, 32w+32) in MFMA accumulators for the whole sequence -- the state
// never leaves registers between chunks.  Per chunk, with P the cumulative
// log-decay inside a sub-chunk and Pe its value at the sub-chunk's end:
//
//   q^ = scale q exp(P)        (<= |q|)   decays a query back to its sub-chunk start
//   k~ = k exp(Pe - P)         (<= |k|)   decays a key forward to its sub-chunk end
//   qv = scale q exp(P - Pe)              diagonal blocks only (see below)
//
//   o   = q^[*dec0] S  +  tril(A) V              A = q k-weights inside the chunk
//   S  <- diag(dec0 dec1) S + (k~0 * dec1 | k~1)^T V
//
// A^T is formed per 16x16 block on the matrix cores: the off-diagonal block is
// k~0 q^1^T (every factor <= 1, no range problem); the two diagonal blocks
// are k~ qv^T, whose q-side exponent P - Pe is positive.  It is clamped at 80
// so nothing overflows; results are exact to bf16 rounding while a channel's
// decay over one 16-token sub-chunk stays above e^-80.
//
// Matrix-core plumbing (v_mfma_f32_16x16x32_bf16), chosen so that no operand
// needs a transposed LDS image:
//   * products contracting over CHANNELS (q^ S, k~ q^T) read 16-byte fragments
//     from row-major [token][channel] LDS tiles; the channel order inside the
//     tile is permuted to the order the state accumulators present their rows;
//   * products contracting over TOKENS (A V, k~^T V) need the token index
//     packed inside a lane: A^T comes out of its MFMA already in that layout,
//     and the V / k~ fragments are built by loading 8 token rows per lane and
//     transposing the 8x8 block of 16-bit values in registers (v_perm_b32);
//   * the fp32 state accumulators are re-packed to bf16 B-fragments in place.
#include "common.h"
#include "gla_gate.h"

namespace hymeta {
namespace {

constexpr int CH = 32;        // tokens per chunk
constexpr int D = kHeadDim;   // 128 channels (k) and 128 value columns
constexpr int LDB = 136;      // bf16 tile row stride in elements (272 B: rows shift 16 B in the banks)
constexpr int LDF = 132;      // fp32 tile row stride in elements

__device__ __forceinline__ f32x4 mfma16(const bf16x8& a, const bf16x8& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ bf16x8 as_frag(const u32x4& v) { return __builtin_bit_cast(bf16x8, v); }

// Position of channel ch inside a permuted [token][channel] tile: the k-slot
// block (ks, c) of an MFMA fragment holds channels 32c + 8r + 2ks + e at slot
// j = 2r + e, which is the row order of the state accumulators.
__device__ __forceinline__ int perm_pos(int ch) {
  const int c = ch >> 5, r = (ch >> 3) & 3, ks = (ch >> 1) & 3, e = ch & 1;
  return 32 * ks + 8 * c + 2 * r + e;
}

// k-slot j of lane group c <-> token inside the 32-token chunk.
__device__ __forceinline__ int slot_token(int c, int j) { return (j < 4) ? 4 * c + j : 16 + 4 * c + (j - 4); }

struct Smem {
  uint16_t qs[CH][LDB];   // q^ (sub-chunk 1 rows also times dec0): A-operand of q^ S
  uint16_t qh1[16][LDB];  // q^ of sub-chunk 1: B-operand of the off-diagonal block
  uint16_t qc[CH][LDB];   // qv: B-operand of the diagonal blocks
  uint16_t kr[CH][LDB];   // k~: A-operand of the A^T blocks               (permuted channels)
  uint16_t ks[CH][LDB];   // k~ (sub-chunk 0 rows times dec1): state update (natural channels)
  uint16_t v[CH][LDB];    // v                                              (natural columns)
  float p[CH][LDF];       // gate -> cumulative log-decay; later the output staging tile
  float dect[D];          // dec0 * dec1
};

}  // namespace

template <bool G_F32, bool STATE_F32>
__global__ __launch_bounds__(256) void gla_prefill_kernel(hymeta_gla_prefill_args a, GateParams gp) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  Smem& sm = *reinterpret_cast<Smem*>(smem_raw);

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int c = lane >> 4;
  const int n = lane & 15;

  const int seq = blockIdx.x / a.num_heads;
  const int h = blockIdx.x % a.num_heads;
  const int kvh = h / (a.num_heads / a.num_kv_heads);
  const int t_begin = a.cu_seqlens[seq];
  const int t_end = a.cu_seqlens[seq + 1];
  const int slot = a.slot_idx[seq];
  if (slot < 0 || t_end <= t_begin) return;

  // ---- initial state into accumulators: s_acc[tp][nt][r] = S[32c + 8r + tp][32w + 2n + nt]
  f32x4 s_acc[8][2];
  {
    const bool first = a.context_lens != nullptr && a.context_lens[seq] == 0;
    const char* src = nullptr;
    if (!first) {
      src = reinterpret_cast<const char*>(a.state) +
            (static_cast<int64_t>(slot) * a.state_stride_slot + static_cast<int64_t>(h) * a.state_stride_h) * (STATE_F32 ? 4 : 2);
    } else if (a.meta_state != nullptr) {
      src = reinterpret_cast<const char*>(a.meta_state) + static_cast<int64_t>(h) * a.meta_state_stride_h * (STATE_F32 ? 4 : 2);
    }
    const int col = 32 * wave + 2 * n;
#pragma unroll
    for (int tp = 0; tp < 8; ++tp) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ch = 32 * c + 8 * r + tp;
        float x0 = 0.f, x1 = 0.f;
        if (src != nullptr) {
          if constexpr (STATE_F32) {
            const float* p = reinterpret_cast<const float*>(src) + ch * D + col;
            x0 = p[0];
            x1 = p[1];
          } else {
            const uint32_t w = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint16_t*>(src) + ch * D + col);
            x0 = lo_bf(w);
            x1 = hi_bf(w);
          }
        }
        s_acc[tp][0][r] = x0;
        s_acc[tp][1][r] = x1;
      }
    }
  }

  // ---- per-thread role in the staging phases: one token, 16 channels
  const int tok = tid >> 3;
  const int g8 = tid & 7;
  const int ch0 = 16 * g8;
  const int sub = tok >> 4;
  const uint16_t* qg = reinterpret_cast<const uint16_t*>(a.q) + static_cast<int64_t>(h) * a.q_stride_h + ch0;
  const uint16_t* kg = reinterpret_cast<const uint16_t*>(a.k) + static_cast<int64_t>(kvh) * a.k_stride_h + ch0;
  const uint16_t* vg = reinterpret_cast<const uint16_t*>(a.v) + static_cast<int64_t>(kvh) * a.v_stride_h + ch0;
  const char* gg = a.g == nullptr ? nullptr
      : reinterpret_cast<const char*>(a.g) + (static_cast<int64_t>(kvh) * a.g_stride_h + ch0) * (G_F32 ? 4 : 2);
  float lbv[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) lbv[e] = a.lower_bound != nullptr ? a.lower_bound[h * D + ch0 + e] : 0.f;
  float nw[16];
  const bool do_norm = a.out_norm_weight != nullptr;
#pragma unroll
  for (int e = 0; e < 16; ++e)
    nw[e] = do_norm ? bf2f(reinterpret_cast<const uint16_t*>(a.out_norm_weight)[ch0 + e]) : 1.f;

  const int nchunks = (t_end - t_begin + CH - 1) / CH;
  for (int ci = 0; ci < nchunks; ++ci) {
    const int tg = t_begin + ci * CH + tok;
    const bool valid = tg < t_end;

    // ---- stage 1: global loads + gate prologue -------------------------------------------
    float qsv[16], kbv[16];
    {
      u32x4 qr[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, kr_[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, vr[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
      float graw[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) graw[e] = 0.f;
      if (valid) {
        const uint16_t* qp = qg + static_cast<int64_t>(tg) * a.q_stride_t;
        const uint16_t* kp = kg + static_cast<int64_t>(tg) * a.k_stride_t;
        const uint16_t* vp = vg + static_cast<int64_t>(tg) * a.v_stride_t;
        qr[0] = ld16(qp); qr[1] = ld16(qp + 8);
        kr_[0] = ld16(kp); kr_[1] = ld16(kp + 8);
        vr[0] = ld16(vp); vr[1] = ld16(vp + 8);
        if (gg != nullptr) {
          if constexpr (G_F32) {
            const float* gp_ = reinterpret_cast<const float*>(gg) + static_cast<int64_t>(tg) * a.g_stride_t;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const f32x4 x = *reinterpret_cast<const f32x4*>(gp_ + 4 * i);
#pragma unroll
              for (int e = 0; e < 4; ++e) graw[4 * i + e] = x[e];
            }
          } else {
            const uint16_t* gp_ = reinterpret_cast<const uint16_t*>(gg) + static_cast<int64_t>(tg) * a.g_stride_t;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              const u32x4 x = ld16(gp_ + 8 * i);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                graw[8 * i + 2 * e] = lo_bf(x[e]);
                graw[8 * i + 2 * e + 1] = hi_bf(x[e]);
              }
            }
          }
        }
      }
      float gv[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const uint32_t qw = qr[e >> 3][(e >> 1) & 3], kw = kr_[e >> 3][(e >> 1) & 3];
        const float qx = (e & 1) ? hi_bf(qw) : lo_bf(qw);
        const float kx = (e & 1) ? hi_bf(kw) : lo_bf(kw);
        float g;
        apply_gate(gp, qx, kx, graw[e], lbv[e], qsv[e], kbv[e], g);
        if (!valid) { qsv[e] = 0.f; kbv[e] = 0.f; g = 0.f; }  // padding: no decay, no contribution
        gv[e] = g;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
        *reinterpret_cast<f32x4*>(&sm.p[tok][ch0 + 4 * i]) = f32x4{gv[4 * i], gv[4 * i + 1], gv[4 * i + 2], gv[4 * i + 3]};
      st16(&sm.v[tok][ch0], vr[0]);
      st16(&sm.v[tok][ch0 + 8], vr[1]);
    }
    __syncthreads();

    // ---- stage 2: inclusive cumulative log-decay inside each 16-token sub-chunk ------------
    {
      const int d = tid & 127, sb = tid >> 7;
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        acc += sm.p[16 * sb + i][d];
        sm.p[16 * sb + i][d] = acc;
      }
    }
    __syncthreads();

    // ---- stage 3: decayed operands ---------------------------------------------------------
    {
      float pe0[16], pe1[16], pt[16];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(&sm.p[15][ch0 + 4 * i]);
        const f32x4 x1 = *reinterpret_cast<const f32x4*>(&sm.p[31][ch0 + 4 * i]);
        const f32x4 xt = *reinterpret_cast<const f32x4*>(&sm.p[tok][ch0 + 4 * i]);
#pragma unroll
        for (int e = 0; e < 4; ++e) { pe0[4 * i + e] = x0[e]; pe1[4 * i + e] = x1[e]; pt[4 * i + e] = xt[e]; }
      }
      float v_qs[16], v_qh[16], v_qc[16], v_kr[16], v_ks[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float pe = sub ? pe1[e] : pe0[e];
        const float qh = qsv[e] * __expf(pt[e]);
        const float kt = kbv[e] * __expf(pe - pt[e]);
        v_qh[e] = qh;
        v_qc[e] = qsv[e] * __expf(fminf(pt[e] - pe, 80.f));
        v_kr[e] = kt;
        v_qs[e] = sub ? qh * __expf(pe0[e]) : qh;
        v_ks[e] = sub ? kt : kt * __expf(pe1[e]);
      }
      if (tok == 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) sm.dect[ch0 + e] = __expf(pe0[e] + pe1[e]);
      }
      // permuted tiles: channel pairs (ch, ch+1) stay adjacent
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int pos = perm_pos(ch0 + 2 * i);
        *reinterpret_cast<uint32_t*>(&sm.qs[tok][pos]) = pack_bf16x2(v_qs[2 * i], v_qs[2 * i + 1]);
        *reinterpret_cast<uint32_t*>(&sm.qc[tok][pos]) = pack_bf16x2(v_qc[2 * i], v_qc[2 * i + 1]);
        *reinterpret_cast<uint32_t*>(&sm.kr[tok][pos]) = pack_bf16x2(v_kr[2 * i], v_kr[2 * i + 1]);
        if (sub) *reinterpret_cast<uint32_t*>(&sm.qh1[tok - 16][pos]) = pack_bf16x2(v_qh[2 * i], v_qh[2 * i + 1]);
      }
      u32x4 w0, w1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        w0[i] = pack_bf16x2(v_ks[2 * i], v_ks[2 * i + 1]);
        w1[i] = pack_bf16x2(v_ks[8 + 2 * i], v_ks[8 + 2 * i + 1]);
      }
      st16(&sm.ks[tok][ch0], w0);
      st16(&sm.ks[tok][ch0 + 8], w1);
    }
    __syncthreads();

    // ---- stage 4: matrix work (per wave: all 32 tokens x its 32 value columns) -------------
    f32x4 o_acc[2][2];
    {
      // V fragments: slot j <-> token slot_token(c, j); column 32w + 2n + nt
      bf16x8 vf[2];
      {
        uint32_t d[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
          d[j] = *reinterpret_cast<const uint32_t*>(&sm.v[slot_token(c, j)][32 * wave + 2 * n]);
        u32x4 b0, b1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          b0[i] = __builtin_amdgcn_perm(d[2 * i + 1], d[2 * i], 0x05040100u);
          b1[i] = __builtin_amdgcn_perm(d[2 * i + 1], d[2 * i], 0x07060302u);
        }
        vf[0] = as_frag(b0);
        vf[1] = as_frag(b1);
      }

      // A^T blocks: rows = key token s (4c + r in the block), column = query token t (n)
      f32x4 at00 = {0, 0, 0, 0}, at01 = {0, 0, 0, 0}, at11 = {0, 0, 0, 0};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int off = 32 * ks + 8 * c;
        const bf16x8 k0 = as_frag(ld16(&sm.kr[n][off]));
        const bf16x8 k1 = as_frag(ld16(&sm.kr[16 + n][off]));
        const bf16x8 q0 = as_frag(ld16(&sm.qc[n][off]));
        const bf16x8 q1 = as_frag(ld16(&sm.qc[16 + n][off]));
        const bf16x8 qh = as_frag(ld16(&sm.qh1[n][off]));
        at00 = mfma16(k0, q0, at00);
        at01 = mfma16(k0, qh, at01);
        at11 = mfma16(k1, q1, at11);
      }
      // causal mask inside the diagonal blocks (key s <= query t), then bf16 A-fragments
      u32x4 a0, a1;
      {
        float m00[4], m11[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool keep = 4 * c + r <= n;
          m00[r] = keep ? at00[r] : 0.f;
          m11[r] = keep ? at11[r] : 0.f;
        }
        a0[0] = pack_bf16x2(m00[0], m00[1]); a0[1] = pack_bf16x2(m00[2], m00[3]); a0[2] = 0u; a0[3] = 0u;
        a1[0] = pack_bf16x2(at01[0], at01[1]); a1[1] = pack_bf16x2(at01[2], at01[3]);
        a1[2] = pack_bf16x2(m11[0], m11[1]); a1[3] = pack_bf16x2(m11[2], m11[3]);
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        o_acc[0][nt] = mfma16(as_frag(a0), vf[nt], f32x4{0, 0, 0, 0});
        o_acc[1][nt] = mfma16(as_frag(a1), vf[nt], f32x4{0, 0, 0, 0});
      }

      // o += q^ S : state accumulators re-packed as B-fragments
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int off = 32 * ks + 8 * c;
        const bf16x8 qa0 = as_frag(ld16(&sm.qs[n][off]));
        const bf16x8 qa1 = as_frag(ld16(&sm.qs[16 + n][off]));
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          u32x4 sb;
#pragma unroll
          for (int i = 0; i < 4; ++i) sb[i] = pack_bf16x2(s_acc[2 * ks][nt][i], s_acc[2 * ks + 1][nt][i]);
          o_acc[0][nt] = mfma16(qa0, as_frag(sb), o_acc[0][nt]);
          o_acc[1][nt] = mfma16(qa1, as_frag(sb), o_acc[1][nt]);
        }
      }

      // S <- diag(dec) S + k~^T V : k~ fragments by in-register 8x8 transpose
      {
        u32x4 kraw[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) kraw[j] = ld16(&sm.ks[slot_token(c, j)][8 * n]);
        float dec[4][8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const f32x4 x0 = *reinterpret_cast<const f32x4*>(&sm.dect[32 * c + 8 * r]);
          const f32x4 x1 = *reinterpret_cast<const f32x4*>(&sm.dect[32 * c + 8 * r + 4]);
#pragma unroll
          for (int e = 0; e < 4; ++e) { dec[r][e] = x0[e]; dec[r][4 + e] = x1[e]; }
        }
#pragma unroll
        for (int tp = 0; tp < 8; ++tp) {
          u32x4 kt;
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            const uint32_t x = kraw[2 * w][tp >> 1], y = kraw[2 * w + 1][tp >> 1];
            kt[w] = (tp & 1) ? __builtin_amdgcn_perm(y, x, 0x07060302u) : __builtin_amdgcn_perm(y, x, 0x05040100u);
          }
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            f32x4 s = s_acc[tp][nt];
#pragma unroll
            for (int r = 0; r < 4; ++r) s[r] *= dec[r][tp];
            s_acc[tp][nt] = mfma16(as_frag(kt), vf[nt], s);
          }
        }
      }
    }
    // output staging: o_acc[tt][nt][r] = o[16tt + 4c + r][32w + 2n + nt]  (aliases the decay tile)
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float* p = &sm.p[16 * tt + 4 * c + r][32 * wave + 2 * n];
        p[0] = o_acc[tt][0][r];
        p[1] = o_acc[tt][1][r];
      }
    }
    __syncthreads();

    // ---- stage 5: (optional per-head RMS norm) and the store ---------------------------------
    {
      float ov[16];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(&sm.p[tok][ch0 + 4 * i]);
#pragma unroll
        for (int e = 0; e < 4; ++e) ov[4 * i + e] = x[e];
      }
      if (do_norm) {
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          ov[e] = bf16_round(ov[e]);
          ss += ov[e] * ov[e];
        }
        ss += __shfl_xor(ss, 1);
        ss += __shfl_xor(ss, 2);
        ss += __shfl_xor(ss, 4);
        const float rs = rsqrtf(ss * (1.f / D) + a.out_norm_eps);
#pragma unroll
        for (int e = 0; e < 16; ++e) ov[e] = bf16_round(ov[e] * rs) * nw[e];
      }
      if (valid) {
        u32x4 w0, w1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          w0[i] = pack_bf16x2(ov[2 * i], ov[2 * i + 1]);
          w1[i] = pack_bf16x2(ov[8 + 2 * i], ov[8 + 2 * i + 1]);
        }
        uint16_t* op = reinterpret_cast<uint16_t*>(a.out) + static_cast<int64_t>(tg) * a.out_stride_t +
                       static_cast<int64_t>(h) * a.out_stride_h + ch0;
        st16(op, w0);
        st16(op + 8, w1);
      }
    }
    __syncthreads();
  }

  // ---- final state back to its slot ---------------------------------------------------------
  {
    char* dst = reinterpret_cast<char*>(a.state) +
                (static_cast<int64_t>(slot) * a.state_stride_slot + static_cast<int64_t>(h) * a.state_stride_h) * (STATE_F32 ? 4 : 2);
    const int col = 32 * wave + 2 * n;
#pragma unroll
    for (int tp = 0; tp < 8; ++tp) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ch = 32 * c + 8 * r + tp;
        if constexpr (STATE_F32) {
          float* p = reinterpret_cast<float*>(dst) + ch * D + col;
          p[0] = s_acc[tp][0][r];
          p[1] = s_acc[tp][1][r];
        } else {
          *reinterpret_cast<uint32_t*>(reinterpret_cast<uint16_t*>(dst) + ch * D + col) =
              pack_bf16x2(s_acc[tp][0][r], s_acc[tp][1][r]);
        }
      }
    }
  }
}

}  // namespace hymeta

using namespace hymeta;

extern "C" int hymeta_gla_prefill_varlen(const hymeta_gla_prefill_args* args, hymeta_stream_t stream) {
  if (args == nullptr) return HYMETA_ERR_INVALID_ARG;
  const hymeta_gla_prefill_args& a = *args;
  if (a.num_seqs < 0 || a.total_tokens < 0 || a.num_heads <= 0 || a.num_kv_heads <= 0) return HYMETA_ERR_INVALID_ARG;
  if (a.num_seqs == 0 || a.total_tokens == 0) return HYMETA_OK;
  if (!a.q || !a.k || !a.v || !a.out || !a.state || !a.slot_idx || !a.cu_seqlens) return HYMETA_ERR_INVALID_ARG;
  if (a.head_dim != kHeadDim) return HYMETA_ERR_UNSUPPORTED;
  if (a.num_heads % a.num_kv_heads != 0) return HYMETA_ERR_INVALID_ARG;
  if (a.g_dtype != HYMETA_BF16 && a.g_dtype != HYMETA_F32) return HYMETA_ERR_UNSUPPORTED;
  if (a.state_dtype != HYMETA_BF16 && a.state_dtype != HYMETA_F32) return HYMETA_ERR_UNSUPPORTED;
  if (a.gate_mode < HYMETA_GATE_NONE || a.gate_mode > HYMETA_GATE_70B_LOWER_BOUND) return HYMETA_ERR_UNSUPPORTED;
  if ((a.gate_mode == HYMETA_GATE_NONE || a.gate_mode == HYMETA_GATE_7B) && !a.g) return HYMETA_ERR_INVALID_ARG;
  if (a.gate_mode == HYMETA_GATE_70B_LOWER_BOUND && !a.lower_bound) return HYMETA_ERR_INVALID_ARG;
  const int64_t strides[] = {a.q_stride_t, a.q_stride_h, a.k_stride_t, a.k_stride_h, a.v_stride_t, a.v_stride_h,
                             a.out_stride_t, a.out_stride_h, a.state_stride_slot, a.state_stride_h,
                             a.meta_state_stride_h, a.g ? a.g_stride_t : 0, a.g ? a.g_stride_h : 0};
  for (int64_t st : strides)
    if (st % 8 != 0) return HYMETA_ERR_UNSUPPORTED;
  const uintptr_t ptrs = reinterpret_cast<uintptr_t>(a.q) | reinterpret_cast<uintptr_t>(a.k) |
                         reinterpret_cast<uintptr_t>(a.v) | reinterpret_cast<uintptr_t>(a.g) |
                         reinterpret_cast<uintptr_t>(a.out) | reinterpret_cast<uintptr_t>(a.state) |
                         reinterpret_cast<uintptr_t>(a.meta_state);
  if (ptrs % 16 != 0) return HYMETA_ERR_UNSUPPORTED;

  GateParams gp;
  gp.mode = a.gate_mode;
  gp.scale = a.scale > 0.f ? a.scale : 1.f / sqrtf(static_cast<float>(a.head_dim));
  gp.normalizer = a.gate_normalizer > 0.f ? a.gate_normalizer : 16.f;
  gp.clamp = a.gate_clamp > 0.f ? a.gate_clamp : 0.95f;

  hipStream_t s = static_cast<hipStream_t>(stream);
  dim3 grid(static_cast<unsigned>(a.num_seqs) * a.num_heads), block(256);
  const size_t lds = sizeof(Smem);
  const bool gf = a.g_dtype == HYMETA_F32, sf = a.state_dtype == HYMETA_F32;
  if (gf && sf) hipLaunchKernelGGL((gla_prefill_kernel<true, true>), grid, block, lds, s, a, gp);
  else if (gf) hipLaunchKernelGGL((gla_prefill_kernel<true, false>), grid, block, lds, s, a, gp);
  else if (sf) hipLaunchKernelGGL((gla_prefill_kernel<false, true>), grid, block, lds, s, a, gp);
  else hipLaunchKernelGGL((gla_prefill_kernel<false, false>), grid, block, lds, s, a, gp);
  return launch_status();
}
