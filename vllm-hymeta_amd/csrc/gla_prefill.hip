// GLA chunked prefill, all sequences of a step in one call.
//
// Recurrence per (sequence, head): S_t = diag(exp g_t) S_{t-1} + k_t^T v_t,
// o_t = (scale q_t) S_t, state S is 128 x 128.
//
// A 512-thread workgroup walks a run of 32-token chunks (two 16-token
// sub-chunks each) of one (sequence, head).  Its eight waves are split by role
// and run one chunk apart through double-buffered LDS tiles, one workgroup
// barrier per chunk:
//
//   producer waves 4..7   (wave p owns channels [32p, 32p+32), lane = token x
//                          16-channel group x sub-chunk)
//       global loads two chunks ahead (registers), gate prologue, the
//       inclusive scan of the log-decay over the 16 tokens of a sub-chunk as a
//       DPP row scan (a DPP row is 16 lanes = the 16 tokens), decayed bf16
//       operand tiles into LDS; and the store of the PREVIOUS chunk's outputs
//       (optional per-head RMS norm) from the fp32 staging tile.
//   consumer waves 0..3   (wave w owns value columns [32w, 32w+32))
//       all matrix-core work of the chunk; the fp32 state stays in MFMA
//       accumulators for the whole run.
//
// Per chunk, with P the cumulative log-decay inside a sub-chunk and Pe its
// value at the sub-chunk's end:
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
//
// Long sequences with few (sequence, head) pairs are cut into segments that
// run in parallel (three launches):
//   A  every segment but a sequence's last: local state U_j (zero start) and
//      the segment's total log-decay, fp32, into the workspace;
//   B  per (sequence, head) the scan X_0 = initial state,
//      X_{j+1} = diag(exp dec_j) X_j + U_j, leaving X_j where U_j was;
//   C  every segment: the full chunk pipeline started from X_j; the last
//      segment stores the final state.
#include "common.h"
#include "gla_gate.h"

namespace hymeta {
namespace {

constexpr int CH = 32;        // tokens per chunk
constexpr int D = kHeadDim;   // 128 channels (k) and 128 value columns
constexpr int LDB = 136;      // bf16 tile row stride in elements (272 B: rows shift 16 B in the banks)
constexpr int LDF = 132;      // fp32 tile row stride in elements
constexpr int kThreads = 512;

constexpr int kFull = 0;       // outputs + state
constexpr int kStateOnly = 1;  // phase A: state contribution of a segment only

__device__ __forceinline__ f32x4 mfma16(const bf16x8& a, const bf16x8& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ bf16x8 as_frag(const u32x4& v) { return __builtin_bit_cast(bf16x8, v); }

// k-slot j of lane group c <-> token inside the 32-token chunk.
__device__ __forceinline__ int slot_token(int c, int j) { return (j < 4) ? 4 * c + j : 16 + 4 * c + (j - 4); }

// x of the lane `N` places lower in the same 16-lane DPP row, 0 for the first N lanes.
template <int N>
__device__ __forceinline__ float row_shr(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x110 + N, 0xf, 0xf, false));
}

// Operand tiles of one chunk.  In the "permuted" tiles channel 32c + 8r + 2ks + e
// sits at position 32ks + 8c + 2r + e, the row order of the state accumulators.
struct Tiles {
  uint16_t qs[CH][LDB];   // q^ (sub-chunk 1 rows also times dec0): A-operand of q^ S      (permuted)
  uint16_t qh1[16][LDB];  // q^ of sub-chunk 1: B-operand of the off-diagonal block        (permuted)
  uint16_t qc[CH][LDB];   // qv: B-operand of the diagonal blocks                          (permuted)
  uint16_t kr[CH][LDB];   // k~: A-operand of the A^T blocks                               (permuted)
  uint16_t ks[CH][LDB];   // k~ (sub-chunk 0 rows times dec1): state update                (natural)
  uint16_t v[CH][LDB];    // v                                                             (natural)
  float dect[D];          // dec0 * dec1
};

struct Smem {
  Tiles t[2];
  float o[2][CH][LDF];    // fp32 outputs of a chunk, consumer -> producer
};

struct Seg {
  int seq, h, kvh, slot;
  int sh;         // head index in the state tensors (h, or the kv-head when a group shares its state)
  int tb;         // first token of the run
  int t_end;      // end of the sequence
  int nchunks;    // chunks in the run
  bool last;      // run ends the sequence
  float* ws_state;  // fp32 [D][D] of this segment in the workspace, or nullptr
  float* ws_dec;    // fp32 [D]
};

// ---------------------------------------------------------------------------------------------
// producer side
// ---------------------------------------------------------------------------------------------
template <bool G_F32>
struct Raw {
  u32x4 q[2], k[2], v[2];
  u32x4 g[G_F32 ? 4 : 2];
};

template <int GATE, bool G_F32, int MODE>
struct Producer {
  const hymeta_gla_prefill_args& a;
  const GateParams& gp;
  const Seg& sg;
  int pw, lane, tok, ch0;   // build role: token in chunk, first of 16 channels
  bool sub;
  const uint16_t *qg, *kg, *vg;
  const char* gg;
  float inv_normalizer;
  float lbv[16];

  __device__ __forceinline__ Producer(const hymeta_gla_prefill_args& a_, const GateParams& gp_, const Seg& sg_, int pw_, int lane_)
      : a(a_), gp(gp_), sg(sg_), pw(pw_), lane(lane_) {
    tok = (lane & 15) + 16 * (lane >> 5);
    sub = (lane >> 5) != 0;
    ch0 = 32 * pw + 16 * ((lane >> 4) & 1);
    qg = reinterpret_cast<const uint16_t*>(a.q) + static_cast<int64_t>(sg.h) * a.q_stride_h + ch0;
    kg = reinterpret_cast<const uint16_t*>(a.k) + static_cast<int64_t>(sg.kvh) * a.k_stride_h + ch0;
    vg = reinterpret_cast<const uint16_t*>(a.v) + static_cast<int64_t>(sg.kvh) * a.v_stride_h + ch0;
    gg = a.g == nullptr ? nullptr
        : reinterpret_cast<const char*>(a.g) + (static_cast<int64_t>(sg.kvh) * a.g_stride_h + ch0) * (G_F32 ? 4 : 2);
    inv_normalizer = 1.f / gp.normalizer;
#pragma unroll
    for (int e = 0; e < 16; ++e)
      lbv[e] = (GATE == HYMETA_GATE_70B_LOWER_BOUND) ? a.lower_bound[sg.h * D + ch0 + e] : 0.f;
  }

  __device__ __forceinline__ void load(int ci, Raw<G_F32>& r) const {
    const int tg = sg.tb + ci * CH + tok;
    const u32x4 z = {0, 0, 0, 0};
    r.q[0] = r.q[1] = r.k[0] = r.k[1] = r.v[0] = r.v[1] = z;
#pragma unroll
    for (int i = 0; i < (G_F32 ? 4 : 2); ++i) r.g[i] = z;
    if (ci < sg.nchunks && tg < sg.t_end) {
      const uint16_t* kp = kg + static_cast<int64_t>(tg) * a.k_stride_t;
      const uint16_t* vp = vg + static_cast<int64_t>(tg) * a.v_stride_t;
      if (MODE == kFull) {
        const uint16_t* qp = qg + static_cast<int64_t>(tg) * a.q_stride_t;
        r.q[0] = ld16(qp); r.q[1] = ld16(qp + 8);
      }
      r.k[0] = ld16(kp); r.k[1] = ld16(kp + 8);
      r.v[0] = ld16(vp); r.v[1] = ld16(vp + 8);
      if constexpr (GATE == HYMETA_GATE_NONE || GATE == HYMETA_GATE_7B) {   // the 70B gates do not read g
        const char* gp_ = gg + static_cast<int64_t>(tg) * a.g_stride_t * (G_F32 ? 4 : 2);
#pragma unroll
        for (int i = 0; i < (G_F32 ? 4 : 2); ++i) r.g[i] = ld16(gp_ + 16 * i);
      }
    }
  }

  // First half of a chunk's build: consumes the raw registers (so that the next loads can be
  // issued into them right away): gate prologue and the V rows.
  __device__ __forceinline__ void gate(int ci, const Raw<G_F32>& r, Tiles& t, float (&qsv)[16], float (&kbv)[16],
                                       float (&pt)[16]) const {
    const bool valid = sg.tb + ci * CH + tok < sg.t_end;
    st16(&t.v[tok][ch0], r.v[0]);
    st16(&t.v[tok][ch0 + 8], r.v[1]);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const uint32_t qw = r.q[e >> 3][(e >> 1) & 3], kw = r.k[e >> 3][(e >> 1) & 3];
      const float qx = (e & 1) ? hi_bf(qw) : lo_bf(qw);
      const float kx = (e & 1) ? hi_bf(kw) : lo_bf(kw);
      float graw;
      if constexpr (G_F32) {
        graw = __uint_as_float(r.g[e >> 2][e & 3]);
      } else {
        const uint32_t gw = r.g[e >> 3][(e >> 1) & 3];
        graw = (e & 1) ? hi_bf(gw) : lo_bf(gw);
      }
      float g;
      apply_gate_fast<GATE>(gp, inv_normalizer, qx, kx, graw, lbv[e], qsv[e], kbv[e], g);
      // padding rows (their raw q is zero): no key contribution, no decay
      kbv[e] = valid ? kbv[e] : 0.f;
      pt[e] = valid ? g : 0.f;
    }
  }

  // Second half: scan, decayed operands, tiles.  ptot accumulates the chunk's total log-decay
  // (phase A only).
  __device__ __forceinline__ void finish(Tiles& t, const float (&qsv)[16], const float (&kbv)[16], float (&pt)[16],
                                         float (&ptot)[16]) const {
    // inclusive scan over the 16 tokens of the sub-chunk (one DPP row)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      float x = pt[e];
      x += row_shr<1>(x);
      x += row_shr<2>(x);
      x += row_shr<4>(x);
      x += row_shr<8>(x);
      pt[e] = x;
    }
    // Row-uniform factors travel by lane permute from the row's last token instead of being
    // recomputed: Pe, exp(Pe) of this sub-chunk and exp(Pe) of the other one.
    const int last_lane = lane | 15;
    u32x4 w0, w1;
    uint32_t w_qs[8], w_qc[8], w_kr[8], w_qh[8];
    float dects[16];
#pragma unroll
    for (int e2 = 0; e2 < 8; ++e2) {
      float o_qs[2], o_qc[2], o_kr[2], o_qh[2], o_ks[2];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int e = 2 * e2 + b;
        const float ept = __expf(pt[e]);
        const float pe = __shfl(pt[e], last_lane);
        const float ee_own = __shfl(ept, last_lane);
        const float ee_oth = __shfl(ept, last_lane ^ 32);
        dects[e] = ee_own * ee_oth;
        if (MODE == kStateOnly) ptot[e] += pe + __shfl(pt[e], last_lane ^ 32);
        const float kt = kbv[e] * __expf(pe - pt[e]);
        o_kr[b] = kt;
        o_ks[b] = kt * (sub ? 1.f : ee_oth);            // sub-chunk 0 keys also cross sub-chunk 1
        if (MODE == kFull) {
          const float qh = qsv[e] * ept;
          o_qh[b] = qh;
          o_qc[b] = qsv[e] * __expf(fminf(pt[e] - pe, 80.f));
          o_qs[b] = qh * (sub ? ee_oth : 1.f);          // sub-chunk 1 queries also cross sub-chunk 0
        }
      }
      const uint32_t ks2 = pack_bf16x2(o_ks[0], o_ks[1]);
      if (e2 < 4) w0[e2] = ks2; else w1[e2 - 4] = ks2;
      w_kr[e2] = pack_bf16x2(o_kr[0], o_kr[1]);
      if (MODE == kFull) {
        w_qs[e2] = pack_bf16x2(o_qs[0], o_qs[1]);
        w_qc[e2] = pack_bf16x2(o_qc[0], o_qc[1]);
        w_qh[e2] = pack_bf16x2(o_qh[0], o_qh[1]);
      }
    }
    if ((lane & 47) == 0) {   // token 0 of sub-chunk 0, both channel groups
#pragma unroll
      for (int i = 0; i < 4; ++i)
        *reinterpret_cast<f32x4*>(&t.dect[ch0 + 4 * i]) = f32x4{dects[4 * i], dects[4 * i + 1], dects[4 * i + 2], dects[4 * i + 3]};
    }
    // permuted tiles: for k-slot block ks the lane's channel pairs (2ks, 2ks+1) and (8+2ks, 9+2ks)
    // land on 4 adjacent positions -> one 8-byte store per block
    const int pbase = 8 * pw + 4 * ((lane >> 4) & 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int pos = 32 * ks + pbase;
      *reinterpret_cast<u32x2*>(&t.kr[tok][pos]) = u32x2{w_kr[ks], w_kr[4 + ks]};
      if (MODE == kFull) {
        *reinterpret_cast<u32x2*>(&t.qs[tok][pos]) = u32x2{w_qs[ks], w_qs[4 + ks]};
        *reinterpret_cast<u32x2*>(&t.qc[tok][pos]) = u32x2{w_qc[ks], w_qc[4 + ks]};
        if (sub) *reinterpret_cast<u32x2*>(&t.qh1[tok - 16][pos]) = u32x2{w_qh[ks], w_qh[4 + ks]};
      }
    }
    st16(&t.ks[tok][ch0], w0);
    st16(&t.ks[tok][ch0 + 8], w1);
  }

  // Chunk ci from the raw registers into tiles t, then the loads of chunk ci_next into them.
  __device__ __forceinline__ void build(int ci, Raw<G_F32>& r, Tiles& t, float (&ptot)[16], int ci_next) const {
    float qsv[16], kbv[16], pt[16];
    gate(ci, r, t, qsv, kbv, pt);
    load(ci_next, r);
    finish(t, qsv, kbv, pt, ptot);
  }
};

// Outputs of chunk ci from the staging tile: producer wave pw stores tokens [8pw, 8pw+8).
__device__ __forceinline__ void store_outputs(const hymeta_gla_prefill_args& a, const Seg& sg, int ci, int pw, int lane,
                                              const float (&o)[CH][LDF], const float (&nw)[16], bool do_norm) {
  const int tok = 8 * pw + (lane >> 3);
  const int c0 = 16 * (lane & 7);
  const int tg = sg.tb + ci * CH + tok;
  float ov[16];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x4 x = *reinterpret_cast<const f32x4*>(&o[tok][c0 + 4 * i]);
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
  if (tg < sg.t_end) {
    u32x4 w0, w1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      w0[i] = pack_bf16x2(ov[2 * i], ov[2 * i + 1]);
      w1[i] = pack_bf16x2(ov[8 + 2 * i], ov[8 + 2 * i + 1]);
    }
    uint16_t* op = reinterpret_cast<uint16_t*>(a.out) + static_cast<int64_t>(tg) * a.out_stride_t +
                   static_cast<int64_t>(sg.h) * a.out_stride_h + c0;
    st16(op, w0);
    st16(op + 8, w1);
  }
}

// ---------------------------------------------------------------------------------------------
// consumer side: one chunk of matrix work; wave w owns value columns [32w, 32w+32).
// s_acc[tp][nt][r] = S[32c + 8r + tp][32w + 2n + nt]
// ---------------------------------------------------------------------------------------------
template <int MODE>
__device__ __forceinline__ void consume_chunk(const Tiles& t, float (&o)[CH][LDF], f32x4 (&s_acc)[8][2], int wave, int lane) {
  const int c = lane >> 4, n = lane & 15;

  // V fragments: slot j <-> token slot_token(c, j); column 32w + 2n + nt
  bf16x8 vf[2];
  {
    uint32_t d[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) d[j] = *reinterpret_cast<const uint32_t*>(&t.v[slot_token(c, j)][32 * wave + 2 * n]);
    u32x4 b0, b1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      b0[i] = __builtin_amdgcn_perm(d[2 * i + 1], d[2 * i], 0x05040100u);
      b1[i] = __builtin_amdgcn_perm(d[2 * i + 1], d[2 * i], 0x07060302u);
    }
    vf[0] = as_frag(b0);
    vf[1] = as_frag(b1);
  }

  if (MODE == kFull) {
    // A^T blocks: rows = key token s (4c + r in the block), column = query token t (n)
    f32x4 at00 = {0, 0, 0, 0}, at01 = {0, 0, 0, 0}, at11 = {0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int off = 32 * ks + 8 * c;
      const bf16x8 k0 = as_frag(ld16(&t.kr[n][off]));
      const bf16x8 k1 = as_frag(ld16(&t.kr[16 + n][off]));
      const bf16x8 q0 = as_frag(ld16(&t.qc[n][off]));
      const bf16x8 q1 = as_frag(ld16(&t.qc[16 + n][off]));
      const bf16x8 qh = as_frag(ld16(&t.qh1[n][off]));
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
    f32x4 o_acc[2][2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      o_acc[0][nt] = mfma16(as_frag(a0), vf[nt], f32x4{0, 0, 0, 0});
      o_acc[1][nt] = mfma16(as_frag(a1), vf[nt], f32x4{0, 0, 0, 0});
    }
    // o += q^ S : state accumulators re-packed as B-fragments
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int off = 32 * ks + 8 * c;
      const bf16x8 qa0 = as_frag(ld16(&t.qs[n][off]));
      const bf16x8 qa1 = as_frag(ld16(&t.qs[16 + n][off]));
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        u32x4 sb;
#pragma unroll
        for (int i = 0; i < 4; ++i) sb[i] = pack_bf16x2(s_acc[2 * ks][nt][i], s_acc[2 * ks + 1][nt][i]);
        o_acc[0][nt] = mfma16(qa0, as_frag(sb), o_acc[0][nt]);
        o_acc[1][nt] = mfma16(qa1, as_frag(sb), o_acc[1][nt]);
      }
    }
    // o_acc[tt][nt][r] = o[16tt + 4c + r][32w + 2n + nt]
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        *reinterpret_cast<float2*>(&o[16 * tt + 4 * c + r][32 * wave + 2 * n]) = float2{o_acc[tt][0][r], o_acc[tt][1][r]};
    }
  }

  // S <- diag(dec) S + k~^T V : k~ fragments by in-register 8x8 transpose
  {
    u32x4 kraw[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) kraw[j] = ld16(&t.ks[slot_token(c, j)][8 * n]);
    float dec[4][8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(&t.dect[32 * c + 8 * r]);
      const f32x4 x1 = *reinterpret_cast<const f32x4*>(&t.dect[32 * c + 8 * r + 4]);
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


// ---------------------------------------------------------------------------------------------
// One state per key/value head: a workgroup walks the chunks of one (sequence, kv-head) and serves
// every query head of the group.  The key side of a chunk (gate, decay scan, k~ tiles, V, state
// update) is built and applied once; the query side (q^ / qv tiles, A^T blocks, outputs) once per
// query head.  The pipeline unit is (chunk, query head): producers build the query tiles of unit
// u + 1 (and, when it opens a chunk, that chunk's key tiles) while the consumers store the outputs
// of unit u - 1 and run the matrix work of unit u.  Arithmetic per (token, head) is that of the per-head kernel, so
// outputs and states are bit-identical to it.
// ---------------------------------------------------------------------------------------------
struct KTiles {
  uint16_t kr[CH][LDB];   // k~: A-operand of the A^T blocks            (permuted)
  uint16_t ks[CH][LDB];   // k~ (sub-chunk 0 rows times dec1): update   (natural)
  uint16_t v[CH][LDB];    // v                                          (natural)
  float dect[D];          // dec0 * dec1
};
struct QTiles {
  uint16_t qs[CH][LDB];   // as in Tiles
  uint16_t qh1[16][LDB];
  uint16_t qc[CH][LDB];
};
struct GSmem {
  KTiles k[2];            // by chunk parity
  QTiles q[2];            // by unit parity
  float o[2][CH][LDF];    // by unit parity
};

template <bool G_F32>
struct RawK {
  u32x4 k[2], v[2];
  u32x4 g[G_F32 ? 4 : 2];
};
struct RawQ {
  u32x4 q[2];
};

template <int GATE, bool G_F32, int MODE>
struct GroupProducer {
  const hymeta_gla_prefill_args& a;
  const GateParams& gp;
  const Seg& sg;            // sg.h = first query head of the group
  int pw, lane, tok, ch0;
  bool sub;
  const uint16_t *qg, *kg, *vg;
  const char* gg;
  float inv_normalizer;
  float f_qh[16], f_qc[16], f_oth[16];   // query-side decay factors of the chunk whose key side was built last

  __device__ __forceinline__ GroupProducer(const hymeta_gla_prefill_args& a_, const GateParams& gp_, const Seg& sg_, int pw_, int lane_)
      : a(a_), gp(gp_), sg(sg_), pw(pw_), lane(lane_) {
    tok = (lane & 15) + 16 * (lane >> 5);
    sub = (lane >> 5) != 0;
    ch0 = 32 * pw + 16 * ((lane >> 4) & 1);
    qg = reinterpret_cast<const uint16_t*>(a.q) + ch0;
    kg = reinterpret_cast<const uint16_t*>(a.k) + static_cast<int64_t>(sg.kvh) * a.k_stride_h + ch0;
    vg = reinterpret_cast<const uint16_t*>(a.v) + static_cast<int64_t>(sg.kvh) * a.v_stride_h + ch0;
    gg = a.g == nullptr ? nullptr
        : reinterpret_cast<const char*>(a.g) + (static_cast<int64_t>(sg.kvh) * a.g_stride_h + ch0) * (G_F32 ? 4 : 2);
    inv_normalizer = 1.f / gp.normalizer;
  }

  __device__ __forceinline__ void load_k(int ci, RawK<G_F32>& r) const {
    const int tg = sg.tb + ci * CH + tok;
    const u32x4 z = {0, 0, 0, 0};
    r.k[0] = r.k[1] = r.v[0] = r.v[1] = z;
#pragma unroll
    for (int i = 0; i < (G_F32 ? 4 : 2); ++i) r.g[i] = z;
    if (ci < sg.nchunks && tg < sg.t_end) {
      const uint16_t* kp = kg + static_cast<int64_t>(tg) * a.k_stride_t;
      const uint16_t* vp = vg + static_cast<int64_t>(tg) * a.v_stride_t;
      r.k[0] = ld16(kp); r.k[1] = ld16(kp + 8);
      r.v[0] = ld16(vp); r.v[1] = ld16(vp + 8);
      if constexpr (GATE == HYMETA_GATE_NONE || GATE == HYMETA_GATE_7B) {
        const char* gp_ = gg + static_cast<int64_t>(tg) * a.g_stride_t * (G_F32 ? 4 : 2);
#pragma unroll
        for (int i = 0; i < (G_F32 ? 4 : 2); ++i) r.g[i] = ld16(gp_ + 16 * i);
      }
    }
  }

  // q rows of unit (chunk ci, r-th query head of the group)
  __device__ __forceinline__ void load_q(int ci, int r_, RawQ& r) const {
    const int tg = sg.tb + ci * CH + tok;
    const u32x4 z = {0, 0, 0, 0};
    r.q[0] = r.q[1] = z;
    if (ci < sg.nchunks && tg < sg.t_end) {
      const uint16_t* qp = qg + static_cast<int64_t>(tg) * a.q_stride_t + static_cast<int64_t>(sg.h + r_) * a.q_stride_h;
      r.q[0] = ld16(qp); r.q[1] = ld16(qp + 8);
    }
  }

  // Key side of chunk ci into t; the loads of chunk ci_next are issued as soon as the raw registers are free.
  __device__ __forceinline__ void build_k(int ci, RawK<G_F32>& r, KTiles& t, float (&ptot)[16], int ci_next) {
    const bool valid = sg.tb + ci * CH + tok < sg.t_end;
    float kbv[16], pt[16];
    st16(&t.v[tok][ch0], r.v[0]);
    st16(&t.v[tok][ch0 + 8], r.v[1]);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const uint32_t kw = r.k[e >> 3][(e >> 1) & 3];
      const float kx = (e & 1) ? hi_bf(kw) : lo_bf(kw);
      float graw;
      if constexpr (G_F32) {
        graw = __uint_as_float(r.g[e >> 2][e & 3]);
      } else {
        const uint32_t gw = r.g[e >> 3][(e >> 1) & 3];
        graw = (e & 1) ? hi_bf(gw) : lo_bf(gw);
      }
      float g, q_unused;
      apply_gate_fast<GATE>(gp, inv_normalizer, 0.f, kx, graw, 0.f, q_unused, kbv[e], g);
      kbv[e] = valid ? kbv[e] : 0.f;
      pt[e] = valid ? g : 0.f;
    }
    load_k(ci_next, r);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      float x = pt[e];
      x += row_shr<1>(x);
      x += row_shr<2>(x);
      x += row_shr<4>(x);
      x += row_shr<8>(x);
      pt[e] = x;
    }
    const int last_lane = lane | 15;
    u32x4 w0, w1;
    uint32_t w_kr[8];
    float dects[16];
#pragma unroll
    for (int e2 = 0; e2 < 8; ++e2) {
      float o_kr[2], o_ks[2];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int e = 2 * e2 + b;
        const float ept = __expf(pt[e]);
        const float pe = __shfl(pt[e], last_lane);
        const float ee_own = __shfl(ept, last_lane);
        const float ee_oth = __shfl(ept, last_lane ^ 32);
        dects[e] = ee_own * ee_oth;
        if (MODE == kStateOnly) ptot[e] += pe + __shfl(pt[e], last_lane ^ 32);
        const float kt = kbv[e] * __expf(pe - pt[e]);
        o_kr[b] = kt;
        o_ks[b] = kt * (sub ? 1.f : ee_oth);
        if (MODE == kFull) {
          f_qh[e] = ept;
          f_qc[e] = __expf(fminf(pt[e] - pe, 80.f));
          f_oth[e] = sub ? ee_oth : 1.f;
        }
      }
      const uint32_t ks2 = pack_bf16x2(o_ks[0], o_ks[1]);
      if (e2 < 4) w0[e2] = ks2; else w1[e2 - 4] = ks2;
      w_kr[e2] = pack_bf16x2(o_kr[0], o_kr[1]);
    }
    if ((lane & 47) == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        *reinterpret_cast<f32x4*>(&t.dect[ch0 + 4 * i]) = f32x4{dects[4 * i], dects[4 * i + 1], dects[4 * i + 2], dects[4 * i + 3]};
    }
    const int pbase = 8 * pw + 4 * ((lane >> 4) & 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      *reinterpret_cast<u32x2*>(&t.kr[tok][32 * ks + pbase]) = u32x2{w_kr[ks], w_kr[4 + ks]};
    st16(&t.ks[tok][ch0], w0);
    st16(&t.ks[tok][ch0 + 8], w1);
  }

  // Query side of one unit from its raw q rows and the chunk's decay factors.
  __device__ __forceinline__ void build_q(const RawQ& r, QTiles& t) const {
    uint32_t w_qs[8], w_qc[8], w_qh[8];
#pragma unroll
    for (int e2 = 0; e2 < 8; ++e2) {
      float o_qs[2], o_qc[2], o_qh[2];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int e = 2 * e2 + b;
        const uint32_t qw = r.q[e >> 3][(e >> 1) & 3];
        const float qx = (e & 1) ? hi_bf(qw) : lo_bf(qw);
        const float qsv = (GATE == HYMETA_GATE_NONE ? qx : fmaxf(qx, 0.f)) * gp.scale;
        const float qh = qsv * f_qh[e];
        o_qh[b] = qh;
        o_qc[b] = qsv * f_qc[e];
        o_qs[b] = qh * f_oth[e];
      }
      w_qs[e2] = pack_bf16x2(o_qs[0], o_qs[1]);
      w_qc[e2] = pack_bf16x2(o_qc[0], o_qc[1]);
      w_qh[e2] = pack_bf16x2(o_qh[0], o_qh[1]);
    }
    const int pbase = 8 * pw + 4 * ((lane >> 4) & 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int pos = 32 * ks + pbase;
      *reinterpret_cast<u32x2*>(&t.qs[tok][pos]) = u32x2{w_qs[ks], w_qs[4 + ks]};
      *reinterpret_cast<u32x2*>(&t.qc[tok][pos]) = u32x2{w_qc[ks], w_qc[4 + ks]};
      if (sub) *reinterpret_cast<u32x2*>(&t.qh1[tok - 16][pos]) = u32x2{w_qh[ks], w_qh[4 + ks]};
    }
  }
};

// V fragments of a chunk: slot j <-> token slot_token(c, j); column 32w + 2n + nt
__device__ __forceinline__ void load_vf(const KTiles& t, int wave, int lane, bf16x8 (&vf)[2]) {
  const int c = lane >> 4, n = lane & 15;
  uint32_t d[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) d[j] = *reinterpret_cast<const uint32_t*>(&t.v[slot_token(c, j)][32 * wave + 2 * n]);
  u32x4 b0, b1;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    b0[i] = __builtin_amdgcn_perm(d[2 * i + 1], d[2 * i], 0x05040100u);
    b1[i] = __builtin_amdgcn_perm(d[2 * i + 1], d[2 * i], 0x07060302u);
  }
  vf[0] = as_frag(b0);
  vf[1] = as_frag(b1);
}

// Outputs of one query head for one chunk (the query half of consume_chunk).
// k~ fragments of a chunk (A-operand of the A^T blocks): the same for every query head of the group.
__device__ __forceinline__ void load_kfr(const KTiles& t, int lane, bf16x8 (&kfr)[2][4]) {
  const int c = lane >> 4, n = lane & 15;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    kfr[0][ks] = as_frag(ld16(&t.kr[n][32 * ks + 8 * c]));
    kfr[1][ks] = as_frag(ld16(&t.kr[16 + n][32 * ks + 8 * c]));
  }
}

__device__ __forceinline__ void consume_q(const bf16x8 (&kfr)[2][4], const QTiles& qt, float (&o)[CH][LDF], const f32x4 (&s_acc)[8][2],
                                          const bf16x8 (&vf)[2], int wave, int lane) {
  const int c = lane >> 4, n = lane & 15;
  f32x4 at00 = {0, 0, 0, 0}, at01 = {0, 0, 0, 0}, at11 = {0, 0, 0, 0};
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int off = 32 * ks + 8 * c;
    const bf16x8 k0 = kfr[0][ks];
    const bf16x8 k1 = kfr[1][ks];
    const bf16x8 q0 = as_frag(ld16(&qt.qc[n][off]));
    const bf16x8 q1 = as_frag(ld16(&qt.qc[16 + n][off]));
    const bf16x8 qh = as_frag(ld16(&qt.qh1[n][off]));
    at00 = mfma16(k0, q0, at00);
    at01 = mfma16(k0, qh, at01);
    at11 = mfma16(k1, q1, at11);
  }
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
  f32x4 o_acc[2][2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    o_acc[0][nt] = mfma16(as_frag(a0), vf[nt], f32x4{0, 0, 0, 0});
    o_acc[1][nt] = mfma16(as_frag(a1), vf[nt], f32x4{0, 0, 0, 0});
  }
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int off = 32 * ks + 8 * c;
    const bf16x8 qa0 = as_frag(ld16(&qt.qs[n][off]));
    const bf16x8 qa1 = as_frag(ld16(&qt.qs[16 + n][off]));
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      u32x4 sb;
#pragma unroll
      for (int i = 0; i < 4; ++i) sb[i] = pack_bf16x2(s_acc[2 * ks][nt][i], s_acc[2 * ks + 1][nt][i]);
      o_acc[0][nt] = mfma16(qa0, as_frag(sb), o_acc[0][nt]);
      o_acc[1][nt] = mfma16(qa1, as_frag(sb), o_acc[1][nt]);
    }
  }
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      *reinterpret_cast<float2*>(&o[16 * tt + 4 * c + r][32 * wave + 2 * n]) = float2{o_acc[tt][0][r], o_acc[tt][1][r]};
  }
}

// S <- diag(dec) S + k~^T V (the state half of consume_chunk).
__device__ __forceinline__ void update_state(const KTiles& t, f32x4 (&s_acc)[8][2], const bf16x8 (&vf)[2], int lane) {
  const int c = lane >> 4, n = lane & 15;
  u32x4 kraw[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) kraw[j] = ld16(&t.ks[slot_token(c, j)][8 * n]);
  float dec[4][8];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const f32x4 x0 = *reinterpret_cast<const f32x4*>(&t.dect[32 * c + 8 * r]);
    const f32x4 x1 = *reinterpret_cast<const f32x4*>(&t.dect[32 * c + 8 * r + 4]);
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

}  // namespace

template <int GATE, bool G_F32, int MODE>
__global__ __launch_bounds__(kThreads) void gla_prefill_group_kernel(hymeta_gla_prefill_args a, GateParams gp, float* workspace,
                                                                     int nseg, int seg_chunks) {
  const bool state_f32 = a.state_dtype == HYMETA_F32;
  __shared__ __attribute__((aligned(16))) GSmem sm;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int SH = a.state_heads;               // == num_kv_heads
  const int rep = a.num_heads / SH;
  const int urep = MODE == kStateOnly ? 1 : rep;   // units per chunk

  Seg sg;
  {
    sg.kvh = blockIdx.x % SH;
    const int rest = blockIdx.x / SH;
    const int j = rest % nseg;
    sg.seq = rest / nseg;
    sg.h = sg.kvh * rep;
    sg.sh = sg.kvh;
    const int t_begin = a.cu_seqlens[sg.seq];
    sg.t_end = a.cu_seqlens[sg.seq + 1];
    sg.slot = a.slot_idx[sg.seq];
    if (sg.slot < 0 || sg.t_end <= t_begin) return;
    const int nch_seq = (sg.t_end - t_begin + CH - 1) / CH;
    const int c0 = j * seg_chunks;
    if (c0 >= nch_seq) return;
    const int c1 = min(c0 + seg_chunks, nch_seq);
    sg.last = c1 == nch_seq;
    if (MODE == kStateOnly && sg.last) return;
    sg.nchunks = c1 - c0;
    sg.tb = t_begin + c0 * CH;
    sg.ws_state = nullptr;
    sg.ws_dec = nullptr;
    if (nseg > 1) {
      const int64_t unit = (static_cast<int64_t>(sg.seq) * SH + sg.kvh) * nseg + j;
      sg.ws_state = workspace + unit * (D * D);
      sg.ws_dec = workspace + static_cast<int64_t>(a.num_seqs) * SH * nseg * (D * D) + unit * D;
    }
  }
  const int U = sg.nchunks * urep;

  if (wave < 4) {
    // ------------------------------------------------------------------ consumers
    const int c = lane >> 4, n = lane & 15;
    const int col = 32 * wave + 2 * n;
    f32x4 s_acc[8][2];
    if (MODE == kStateOnly) {
#pragma unroll
      for (int tp = 0; tp < 8; ++tp) s_acc[tp][0] = s_acc[tp][1] = f32x4{0, 0, 0, 0};
    } else if (nseg > 1) {
#pragma unroll
      for (int tp = 0; tp < 8; ++tp) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float2 x = *reinterpret_cast<const float2*>(sg.ws_state + (32 * c + 8 * r + tp) * D + col);
          s_acc[tp][0][r] = x.x;
          s_acc[tp][1][r] = x.y;
        }
      }
    } else {
      const bool first = a.context_lens != nullptr && a.context_lens[sg.seq] == 0;
      const char* src = nullptr;
      if (!first) {
        src = reinterpret_cast<const char*>(a.state) +
              (static_cast<int64_t>(sg.slot) * a.state_stride_slot + static_cast<int64_t>(sg.sh) * a.state_stride_h) * (state_f32 ? 4 : 2);
      } else if (a.meta_state != nullptr) {
        src = reinterpret_cast<const char*>(a.meta_state) + static_cast<int64_t>(sg.sh) * a.meta_state_stride_h * (state_f32 ? 4 : 2);
      }
#pragma unroll
      for (int tp = 0; tp < 8; ++tp) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ch = 32 * c + 8 * r + tp;
          float x0 = 0.f, x1 = 0.f;
          if (src != nullptr) {
            if (state_f32) {
              const float2 x = *reinterpret_cast<const float2*>(reinterpret_cast<const float*>(src) + ch * D + col);
              x0 = x.x;
              x1 = x.y;
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

    // The consumers also store the outputs of the unit before (staged in fp32 by all four of them): the
    // matrix work of a unit is short, the producers' share is what bounds the pipeline.
    const bool do_norm = MODE == kFull && a.out_norm_weight != nullptr;
    float nw[16];
#pragma unroll
    for (int e = 0; e < 16; ++e)
      nw[e] = do_norm ? bf2f(reinterpret_cast<const uint16_t*>(a.out_norm_weight)[16 * (lane & 7) + e]) : 1.f;

    __syncthreads();   // tiles of unit 0 are ready
    int ci = 0, r = 0;
    int cim = 0, rm = 0;    // (chunk, head) of unit u - 1
    bf16x8 vf[2], kfr[2][4];   // fragments of the chunk's key side, loaded at its first unit
    for (int u = 0; u <= U; ++u) {
      if (MODE == kFull && u >= 1) {
        Seg so = sg;
        so.h = sg.h + rm;
        store_outputs(a, so, cim, wave, lane, sm.o[(u - 1) & 1], nw, do_norm);
        if (++rm == urep) { rm = 0; ++cim; }
      }
      if (u < U) {
        const KTiles& kt = sm.k[ci & 1];
        if (r == 0) {
          load_vf(kt, wave, lane, vf);
          if (MODE == kFull) load_kfr(kt, lane, kfr);
        }
        if (MODE == kFull) consume_q(kfr, sm.q[u & 1], sm.o[u & 1], s_acc, vf, wave, lane);
        if (r == urep - 1) update_state(kt, s_acc, vf, lane);
        if (++r == urep) { r = 0; ++ci; }
      }
      __syncthreads();
    }

    if (MODE == kStateOnly) {
#pragma unroll
      for (int tp = 0; tp < 8; ++tp) {
#pragma unroll
        for (int r2 = 0; r2 < 4; ++r2)
          *reinterpret_cast<float2*>(sg.ws_state + (32 * c + 8 * r2 + tp) * D + col) = float2{s_acc[tp][0][r2], s_acc[tp][1][r2]};
      }
    } else if (sg.last) {
      char* dst = reinterpret_cast<char*>(a.state) +
                  (static_cast<int64_t>(sg.slot) * a.state_stride_slot + static_cast<int64_t>(sg.sh) * a.state_stride_h) * (state_f32 ? 4 : 2);
#pragma unroll
      for (int tp = 0; tp < 8; ++tp) {
#pragma unroll
        for (int r2 = 0; r2 < 4; ++r2) {
          const int ch = 32 * c + 8 * r2 + tp;
          if (state_f32) {
            *reinterpret_cast<float2*>(reinterpret_cast<float*>(dst) + ch * D + col) = float2{s_acc[tp][0][r2], s_acc[tp][1][r2]};
          } else {
            *reinterpret_cast<uint32_t*>(reinterpret_cast<uint16_t*>(dst) + ch * D + col) =
                pack_bf16x2(s_acc[tp][0][r2], s_acc[tp][1][r2]);
          }
        }
      }
    }
  } else {
    // ------------------------------------------------------------------ producers
    const int pw = wave - 4;
    GroupProducer<GATE, G_F32, MODE> P(a, gp, sg, pw, lane);
    float ptot[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) ptot[e] = 0.f;

    RawK<G_F32> rk;
    RawQ rq;
    P.load_k(0, rk);
    if (MODE == kFull) P.load_q(0, 0, rq);
    P.build_k(0, rk, sm.k[0], ptot, 1);
    if (MODE == kFull) {
      P.build_q(rq, sm.q[0]);
      P.load_q(urep > 1 ? 0 : 1, urep > 1 ? 1 : 0, rq);     // unit 1
    }
    __syncthreads();
    int ci1 = urep > 1 ? 0 : 1, r1 = urep > 1 ? 1 : 0;      // (chunk, head) of unit u + 1
    for (int u = 0; u <= U; ++u) {
      if (u + 1 < U) {
        if (r1 == 0) P.build_k(ci1, rk, sm.k[ci1 & 1], ptot, ci1 + 1);
        if (MODE == kFull) {
          P.build_q(rq, sm.q[(u + 1) & 1]);
          int ci2 = ci1, r2 = r1 + 1;                       // unit u + 2
          if (r2 == urep) { r2 = 0; ++ci2; }
          P.load_q(ci2, r2, rq);
        }
      }
      if (++r1 == urep) { r1 = 0; ++ci1; }
      __syncthreads();
    }
    if (MODE == kStateOnly && (lane & 47) == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        *reinterpret_cast<f32x4*>(sg.ws_dec + P.ch0 + 4 * i) = f32x4{ptot[4 * i], ptot[4 * i + 1], ptot[4 * i + 2], ptot[4 * i + 3]};
    }
  }
}

namespace {
}  // namespace

// nseg == 1: one run per (sequence, head), initial state per the context rules.
// nseg  > 1: MODE kStateOnly = phase A, MODE kFull = phase C (initial state X_j from the workspace).
template <int GATE, bool G_F32, int MODE>
__global__ __launch_bounds__(kThreads) void gla_prefill_kernel(hymeta_gla_prefill_args a, GateParams gp, float* workspace,
                                                               int nseg, int seg_chunks) {
  const bool state_f32 = a.state_dtype == HYMETA_F32;
  __shared__ __attribute__((aligned(16))) Smem sm;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;

  Seg sg;
  {
    const int H = a.num_heads;
    sg.h = blockIdx.x % H;
    const int rest = blockIdx.x / H;
    const int j = rest % nseg;
    sg.seq = rest / nseg;
    sg.kvh = sg.h / (H / a.num_kv_heads);
    sg.sh = sg.h;      // one state per query head here; shared states go through gla_prefill_group_kernel
    const int t_begin = a.cu_seqlens[sg.seq];
    sg.t_end = a.cu_seqlens[sg.seq + 1];
    sg.slot = a.slot_idx[sg.seq];
    if (sg.slot < 0 || sg.t_end <= t_begin) return;
    const int nch_seq = (sg.t_end - t_begin + CH - 1) / CH;
    const int c0 = j * seg_chunks;
    if (c0 >= nch_seq) return;
    const int c1 = min(c0 + seg_chunks, nch_seq);
    sg.last = c1 == nch_seq;
    if (MODE == kStateOnly && sg.last) return;   // the last segment's state comes out of the full pass
    sg.nchunks = c1 - c0;
    sg.tb = t_begin + c0 * CH;
    sg.ws_state = nullptr;
    sg.ws_dec = nullptr;
    if (nseg > 1) {
      const int64_t unit = (static_cast<int64_t>(sg.seq) * H + sg.h) * nseg + j;
      sg.ws_state = workspace + unit * (D * D);
      sg.ws_dec = workspace + static_cast<int64_t>(a.num_seqs) * H * nseg * (D * D) + unit * D;
    }
  }

  if (wave < 4) {
    // ------------------------------------------------------------------ consumers
    const int c = lane >> 4, n = lane & 15;
    const int col = 32 * wave + 2 * n;
    f32x4 s_acc[8][2];
    if (MODE == kStateOnly) {
#pragma unroll
      for (int tp = 0; tp < 8; ++tp) s_acc[tp][0] = s_acc[tp][1] = f32x4{0, 0, 0, 0};
    } else if (nseg > 1) {
#pragma unroll
      for (int tp = 0; tp < 8; ++tp) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float2 x = *reinterpret_cast<const float2*>(sg.ws_state + (32 * c + 8 * r + tp) * D + col);
          s_acc[tp][0][r] = x.x;
          s_acc[tp][1][r] = x.y;
        }
      }
    } else {
      const bool first = a.context_lens != nullptr && a.context_lens[sg.seq] == 0;
      const char* src = nullptr;
      if (!first) {
        src = reinterpret_cast<const char*>(a.state) +
              (static_cast<int64_t>(sg.slot) * a.state_stride_slot + static_cast<int64_t>(sg.sh) * a.state_stride_h) * (state_f32 ? 4 : 2);
      } else if (a.meta_state != nullptr) {
        src = reinterpret_cast<const char*>(a.meta_state) + static_cast<int64_t>(sg.sh) * a.meta_state_stride_h * (state_f32 ? 4 : 2);
      }
#pragma unroll
      for (int tp = 0; tp < 8; ++tp) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ch = 32 * c + 8 * r + tp;
          float x0 = 0.f, x1 = 0.f;
          if (src != nullptr) {
            if (state_f32) {
              const float2 x = *reinterpret_cast<const float2*>(reinterpret_cast<const float*>(src) + ch * D + col);
              x0 = x.x;
              x1 = x.y;
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

    __syncthreads();   // tiles of chunk 0 are ready
    for (int ci = 0; ci <= sg.nchunks; ++ci) {
      if (ci < sg.nchunks) consume_chunk<MODE>(sm.t[ci & 1], sm.o[ci & 1], s_acc, wave, lane);
      __syncthreads();
    }

    if (MODE == kStateOnly) {
#pragma unroll
      for (int tp = 0; tp < 8; ++tp) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          *reinterpret_cast<float2*>(sg.ws_state + (32 * c + 8 * r + tp) * D + col) = float2{s_acc[tp][0][r], s_acc[tp][1][r]};
      }
    } else if (sg.last) {
      char* dst = reinterpret_cast<char*>(a.state) +
                  (static_cast<int64_t>(sg.slot) * a.state_stride_slot + static_cast<int64_t>(sg.sh) * a.state_stride_h) * (state_f32 ? 4 : 2);
#pragma unroll
      for (int tp = 0; tp < 8; ++tp) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ch = 32 * c + 8 * r + tp;
          if (state_f32) {
            *reinterpret_cast<float2*>(reinterpret_cast<float*>(dst) + ch * D + col) = float2{s_acc[tp][0][r], s_acc[tp][1][r]};
          } else {
            *reinterpret_cast<uint32_t*>(reinterpret_cast<uint16_t*>(dst) + ch * D + col) =
                pack_bf16x2(s_acc[tp][0][r], s_acc[tp][1][r]);
          }
        }
      }
    }
  } else {
    // ------------------------------------------------------------------ producers
    const int pw = wave - 4;
    Producer<GATE, G_F32, MODE> P(a, gp, sg, pw, lane);
    const bool do_norm = MODE == kFull && a.out_norm_weight != nullptr;
    float nw[16];
#pragma unroll
    for (int e = 0; e < 16; ++e)
      nw[e] = do_norm ? bf2f(reinterpret_cast<const uint16_t*>(a.out_norm_weight)[16 * (lane & 7) + e]) : 1.f;
    float ptot[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) ptot[e] = 0.f;

    Raw<G_F32> raw;
    P.load(0, raw);
    P.build(0, raw, sm.t[0], ptot, 1);
    __syncthreads();
    for (int ci = 0; ci <= sg.nchunks; ++ci) {
      if (ci + 1 < sg.nchunks) P.build(ci + 1, raw, sm.t[(ci + 1) & 1], ptot, ci + 2);
      if (MODE == kFull && ci >= 1) store_outputs(a, sg, ci - 1, pw, lane, sm.o[(ci - 1) & 1], nw, do_norm);
      __syncthreads();
    }
    if (MODE == kStateOnly && (lane & 47) == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        *reinterpret_cast<f32x4*>(sg.ws_dec + P.ch0 + 4 * i) = f32x4{ptot[4 * i], ptot[4 * i + 1], ptot[4 * i + 2], ptot[4 * i + 3]};
    }
  }
}

// Phase B: per (sequence, head) the scan over segments; thread = 4 adjacent state elements.
// H = heads the segment passes ran over: query heads, or state heads when a group shares its state.
template <bool STATE_F32>
__global__ __launch_bounds__(256) void gla_prefill_scan_kernel(hymeta_gla_prefill_args a, float* workspace, int nseg, int seg_chunks,
                                                               int H) {
  const int h = blockIdx.x % H, seq = blockIdx.x / H;
  const int t_begin = a.cu_seqlens[seq], t_end = a.cu_seqlens[seq + 1];
  const int slot = a.slot_idx[seq];
  if (slot < 0 || t_end <= t_begin) return;
  const int nch_seq = (t_end - t_begin + CH - 1) / CH;
  const int j_last = (nch_seq - 1) / seg_chunks;
  const int idx = (blockIdx.y * 256 + threadIdx.x) * 4;   // element of the [D][D] block
  const int ch = idx / D;
  const int sh = h / (H / a.state_heads);     // H == state_heads when a group is one unit of work
  const int64_t unit0 = (static_cast<int64_t>(seq) * H + h) * nseg;
  float* ws = workspace + unit0 * (D * D) + idx;
  const float* wd = workspace + static_cast<int64_t>(a.num_seqs) * H * nseg * (D * D) + unit0 * D + ch;

  f32x4 x = {0, 0, 0, 0};
  const bool first = a.context_lens != nullptr && a.context_lens[seq] == 0;
  const char* src = nullptr;
  if (!first) {
    src = reinterpret_cast<const char*>(a.state) +
          (static_cast<int64_t>(slot) * a.state_stride_slot + static_cast<int64_t>(sh) * a.state_stride_h) * (STATE_F32 ? 4 : 2);
  } else if (a.meta_state != nullptr) {
    src = reinterpret_cast<const char*>(a.meta_state) + static_cast<int64_t>(sh) * a.meta_state_stride_h * (STATE_F32 ? 4 : 2);
  }
  if (src != nullptr) {
    if constexpr (STATE_F32) {
      x = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(src) + idx);
    } else {
      const u32x2 w = *reinterpret_cast<const u32x2*>(reinterpret_cast<const uint16_t*>(src) + idx);
      x = f32x4{lo_bf(w[0]), hi_bf(w[0]), lo_bf(w[1]), hi_bf(w[1])};
    }
  }
  // Segments go through in batches whose contributions are all loaded before the first is used:
  // the recurrence itself is a short chain of FMAs, the loads are what takes time.
  constexpr int kBatch = 8;
  for (int j0 = 0; j0 <= j_last; j0 += kBatch) {
    f32x4 u[kBatch];
    float dl[kBatch];
#pragma unroll
    for (int i = 0; i < kBatch; ++i) {
      u[i] = f32x4{0, 0, 0, 0};
      dl[i] = 0.f;
      if (j0 + i < j_last) {
        u[i] = *reinterpret_cast<const f32x4*>(ws + static_cast<int64_t>(j0 + i) * (D * D));
        dl[i] = wd[(j0 + i) * D];
      }
    }
#pragma unroll
    for (int i = 0; i < kBatch; ++i) {
      const int j = j0 + i;
      if (j <= j_last) *reinterpret_cast<f32x4*>(ws + static_cast<int64_t>(j) * (D * D)) = x;
      if (j < j_last) {
        const float dec = __expf(dl[i]);
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] = dec * x[e] + u[i][e];
      }
    }
  }
}

}  // namespace hymeta

using namespace hymeta;

extern "C" int hymeta_gla_prefill_segments(int32_t num_seqs, int32_t num_heads, int32_t max_seqlen) {
  if (num_seqs <= 0 || num_heads <= 0 || max_seqlen <= 0) return 1;
  const int64_t units = static_cast<int64_t>(num_seqs) * num_heads;
  const int chunks = (max_seqlen + CH - 1) / CH;
  if (units > 128 || chunks < 4) return 1;
  // One workgroup per compute unit in a single round (a workgroup fills a CU: 8 waves at up to 256
  // registers, 128 KB of LDS); more rounds only add the segments' redundant state pass.  Segments
  // keep at least 2 chunks (measured on 2048-token chunks of one sequence: 7 heads 66 us with 8-chunk
  // segments, 42 us with 2-chunk segments; 4 groups of 7 heads 112 against 47 us; one chunk per
  // segment is slower again).
  int nseg = static_cast<int>(256 / units);
  if (nseg > chunks / 2) nseg = chunks / 2;
  return nseg < 1 ? 1 : nseg;
}

extern "C" size_t hymeta_gla_prefill_workspace_bytes(int32_t num_seqs, int32_t num_heads, int32_t state_heads,
                                                     int32_t num_segments) {
  if (num_seqs <= 0 || num_heads <= 0 || num_segments <= 1) return 0;
  const int work_heads = state_heads > 0 && state_heads < num_heads ? state_heads : num_heads;
  return static_cast<size_t>(num_seqs) * work_heads * num_segments * (D * D + D) * sizeof(float);
}

namespace {

template <int GATE, bool G_F32, int MODE>
void launch_one(const hymeta_gla_prefill_args& a, const GateParams& gp, float* ws, int nseg, int seg_chunks, hipStream_t s) {
  dim3 grid(static_cast<unsigned>(a.num_seqs) * a.num_heads * nseg), block(kThreads);
  hipLaunchKernelGGL((gla_prefill_kernel<GATE, G_F32, MODE>), grid, block, 0, s, a, gp, ws, nseg, seg_chunks);
}

template <int GATE, bool G_F32, int MODE>
void launch_group_one(const hymeta_gla_prefill_args& a, const GateParams& gp, float* ws, int nseg, int seg_chunks, hipStream_t s) {
  dim3 grid(static_cast<unsigned>(a.num_seqs) * a.state_heads * nseg), block(kThreads);
  hipLaunchKernelGGL((gla_prefill_group_kernel<GATE, G_F32, MODE>), grid, block, 0, s, a, gp, ws, nseg, seg_chunks);
}

// One workgroup per (sequence, kv-head, segment); the lower-bound gate never comes here.
template <int MODE>
void launch_group_main(const hymeta_gla_prefill_args& a, const GateParams& gp, float* ws, int nseg, int seg_chunks, hipStream_t s) {
  const bool gf = a.g_dtype == HYMETA_F32;
  switch (a.gate_mode) {
    case HYMETA_GATE_7B:
      if (gf) launch_group_one<HYMETA_GATE_7B, true, MODE>(a, gp, ws, nseg, seg_chunks, s);
      else launch_group_one<HYMETA_GATE_7B, false, MODE>(a, gp, ws, nseg, seg_chunks, s);
      break;
    case HYMETA_GATE_70B_CLAMP:
      launch_group_one<HYMETA_GATE_70B_CLAMP, false, MODE>(a, gp, ws, nseg, seg_chunks, s);
      break;
    default:
      if (gf) launch_group_one<HYMETA_GATE_NONE, true, MODE>(a, gp, ws, nseg, seg_chunks, s);
      else launch_group_one<HYMETA_GATE_NONE, false, MODE>(a, gp, ws, nseg, seg_chunks, s);
      break;
  }
}

template <int MODE>
void launch_main(const hymeta_gla_prefill_args& a, const GateParams& gp, float* ws, int nseg, int seg_chunks, hipStream_t s) {
  const bool gf = a.g_dtype == HYMETA_F32;
  switch (a.gate_mode) {
    case HYMETA_GATE_7B:
      if (gf) launch_one<HYMETA_GATE_7B, true, MODE>(a, gp, ws, nseg, seg_chunks, s);
      else launch_one<HYMETA_GATE_7B, false, MODE>(a, gp, ws, nseg, seg_chunks, s);
      break;
    case HYMETA_GATE_70B_CLAMP:   // the 70B gates derive the decay from k: g is not read
      launch_one<HYMETA_GATE_70B_CLAMP, false, MODE>(a, gp, ws, nseg, seg_chunks, s);
      break;
    case HYMETA_GATE_70B_LOWER_BOUND:
      launch_one<HYMETA_GATE_70B_LOWER_BOUND, false, MODE>(a, gp, ws, nseg, seg_chunks, s);
      break;
    default:
      if (gf) launch_one<HYMETA_GATE_NONE, true, MODE>(a, gp, ws, nseg, seg_chunks, s);
      else launch_one<HYMETA_GATE_NONE, false, MODE>(a, gp, ws, nseg, seg_chunks, s);
      break;
  }
}

}  // namespace

extern "C" int hymeta_gla_prefill_varlen(const hymeta_gla_prefill_args* args, hymeta_stream_t stream) {
  if (!hymeta::args_ok(args)) return HYMETA_ERR_INVALID_ARG;
  hymeta_gla_prefill_args a = *args;
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
  if (a.state_heads == 0) a.state_heads = a.num_heads;
  if (a.state_heads != a.num_heads && a.state_heads != a.num_kv_heads) return HYMETA_ERR_INVALID_ARG;
  const bool grouped = a.state_heads != a.num_heads;
  // a state shared by a group needs a gate that is the same for every query head of the group
  if (grouped && a.gate_mode == HYMETA_GATE_70B_LOWER_BOUND) return HYMETA_ERR_INVALID_ARG;
  const int64_t strides[] = {a.q_stride_t, a.q_stride_h, a.k_stride_t, a.k_stride_h, a.v_stride_t, a.v_stride_h,
                             a.out_stride_t, a.out_stride_h, a.state_stride_slot, a.state_stride_h,
                             a.meta_state_stride_h, a.g ? a.g_stride_t : 0, a.g ? a.g_stride_h : 0};
  for (int64_t st : strides)
    if (st % 8 != 0) return HYMETA_ERR_UNSUPPORTED;
  const uintptr_t ptrs = reinterpret_cast<uintptr_t>(a.q) | reinterpret_cast<uintptr_t>(a.k) |
                         reinterpret_cast<uintptr_t>(a.v) | reinterpret_cast<uintptr_t>(a.g) |
                         reinterpret_cast<uintptr_t>(a.out) | reinterpret_cast<uintptr_t>(a.state) |
                         reinterpret_cast<uintptr_t>(a.meta_state) | reinterpret_cast<uintptr_t>(a.workspace);
  if (ptrs % 16 != 0) return HYMETA_ERR_UNSUPPORTED;

  int nseg = a.num_segments;
  if (nseg < 0) return HYMETA_ERR_INVALID_ARG;
  if (a.max_seqlen <= 0) return HYMETA_ERR_INVALID_ARG;
  const int chunks = (a.max_seqlen + CH - 1) / CH;
  // heads that run as separate workgroups: a group that shares its state is one unit of work
  const int work_heads = grouped ? a.state_heads : a.num_heads;
  if (nseg == 0) nseg = a.workspace != nullptr ? hymeta_gla_prefill_segments(a.num_seqs, work_heads, a.max_seqlen) : 1;
  if (nseg > chunks) nseg = chunks;
  if (nseg < 1) nseg = 1;
  // the library's own choice stays within the workspace it was given
  if (a.num_segments == 0 && nseg > 1 &&
      a.workspace_bytes < hymeta_gla_prefill_workspace_bytes(a.num_seqs, a.num_heads, a.state_heads, nseg)) nseg = 1;
  if (nseg > 1) {
    if (a.workspace == nullptr) return HYMETA_ERR_INVALID_ARG;
    if (a.workspace_bytes < hymeta_gla_prefill_workspace_bytes(a.num_seqs, a.num_heads, a.state_heads, nseg)) return HYMETA_ERR_WORKSPACE;
  }
  const int seg_chunks = (chunks + nseg - 1) / nseg;

  GateParams gp;
  gp.mode = a.gate_mode;
  gp.scale = a.scale > 0.f ? a.scale : 1.f / sqrtf(static_cast<float>(a.head_dim));
  gp.normalizer = a.gate_normalizer > 0.f ? a.gate_normalizer : 16.f;
  gp.clamp = a.gate_clamp > 0.f ? a.gate_clamp : 0.95f;

  hipStream_t s = begin_launches(stream);
  float* ws = static_cast<float*>(a.workspace);
  if (nseg > 1) {
    if (grouped) launch_group_main<kStateOnly>(a, gp, ws, nseg, seg_chunks, s);
    else launch_main<kStateOnly>(a, gp, ws, nseg, seg_chunks, s);
    dim3 grid(static_cast<unsigned>(a.num_seqs) * work_heads, D * D / (256 * 4)), block(256);
    if (a.state_dtype == HYMETA_F32)
      hipLaunchKernelGGL((gla_prefill_scan_kernel<true>), grid, block, 0, s, a, ws, nseg, seg_chunks, work_heads);
    else
      hipLaunchKernelGGL((gla_prefill_scan_kernel<false>), grid, block, 0, s, a, ws, nseg, seg_chunks, work_heads);
  }
  if (grouped) launch_group_main<kFull>(a, gp, ws, nseg, seg_chunks, s);
  else launch_main<kFull>(a, gp, ws, nseg, seg_chunks, s);
  return launch_status();
}
