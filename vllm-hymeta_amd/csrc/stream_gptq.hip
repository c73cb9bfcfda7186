// Weight-streaming GEMM for 4-bit GPTQ linears at decode-sized batches (up to 64 rows): out = x dequant(W) (+ bias),
// qweight int32 [K / 8, N] (a word = 8 consecutive k of ONE output column), qzeros int32 [K / G, N / 8], scales
// fp16 | bf16 [K / G, N], group size G = 128 or 64 -- and for 8-bit ones of group size 64 (qweight [K / 4, N], a word = 4 k;
// the 7B recipe of the reference, hf_7B_model/quant_7B.py:32-36), described at the end of this comment.
//
// Built like the bf16 streaming kernel (stream_gemm.hip): a workgroup owns a tile of columns and a K slice, x is staged
// through double-buffered LDS in 256-k chunks (one barrier per chunk), every wave streams the weights of its own
// columns.  The workgroup has 8 waves, two per SIMD: waves w and w + 4 own the same columns and take one each of a
// chunk's two groups of 128 k, so that one wave's de-quantisation (VALU) and LDS round trips run under the other's
// MFMAs; their two partial tiles are added when the tile is written.  What differs further is the weight path:
//   * a wave's columns come in blocks of 32 (one 128-byte line of a qweight row): one load instruction is 8 packed rows
//     x 128 bytes = 64 k of 32 columns, whole lines (lane = row l / 8, 16-byte piece l % 8).  It passes through a
//     wave-private 1 KB LDS image, from which lane (n, cq) of the MFMA layout reads the ONE word that holds its 8
//     k-slots of column n (rows two apart in parity sit in opposite halves of the banks);
//   * (w >> 4 j) & 0x000F000F | 0x43004300 makes the bf16 pair (128 + q_j, 128 + q_{j+4}) in one instruction, 4 per
//     word; the k order this induces inside a word, (0,4,1,5,2,6,3,7), is given to x when it is staged;
//   * scale and zero are applied once per group (of 128 k: once per chunk and wave; of 64 k: twice) to the group's
//     partial sums, outside the MFMA chain:
//       sum_k x w = s * ( sum_k x (128 + q)  -  (128 + z) * sum_k x ),
//     with sum_k x from one extra MFMA per k-step against an all-ones fragment (it lands in the accumulator layout).
// Tiles and waves are dealt the matrix's 32-column blocks in equal shares (a wave owns 1 or 2); narrow matrices are
// cut along K, the slices' fp32 partial tiles go to slabs [slice][M][N] that the bf16 kernel's reduce launch (or the
// consumer) sums.
//
// 8-bit, group 64: in packed rows the geometry is the same -- a unit is 8 packed rows x 128 bytes of a 32-column block,
// a group two units, a chunk of x four units, one group for each half of the waves -- with half the k per row: a unit
// is ONE k-step (32 k), a group 64 k, a chunk of x 128 k (256 bytes a row).  A lane's 8 k-slots of a column are two
// words (packed rows 2 cq and 2 cq + 1), whose bytes become bf16 exactly by conversion (v_cvt_f32_ubyteN + a packed
// round, 12 instructions per fragment against 4 for the nibbles); k stays in natural order, so x is staged as it is;
// zeros enter without the 128 of the nibble trick.
#include <cstdlib>

#include "stream_gemm.h"

namespace hymeta {
namespace {

constexpr int kUnitsPerChunk = 4;               // weight units (8 packed rows each) per chunk
constexpr int kMaxSplits = 16;

__device__ __forceinline__ f32x4 mma_bf16(const u32x4& a, const u32x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ float scale_at(const void* p, int64_t idx, bool f16) {
  const uint16_t raw = static_cast<const uint16_t*>(p)[idx];
  if (f16) return static_cast<float>(__builtin_bit_cast(_Float16, raw));
  return bf2f(raw);
}

// One wave's part of a tile: NP blocks of 32 columns from block ``wave_b0`` on; the tile is blocks [tile_b0, tile_b0 +
// tile_nb).  MT: the wave's 16-row fragments; NPT: the widest wave of the launch (sizes the shared images).
// Waves w and w + 4 own the same columns and split K: one group of every chunk each, over all 16 MT rows.
template <int MT, int NP, int NPT, int BITS, int GU>
__device__ __forceinline__ void gptq_wave(const StreamGptqArgs& a, char* smem, const int wave_b0, const int tile_b0,
                                          const int tile_nb, const int split) {
  // How far ahead the wave asks for things.  Deeper rings (4 chunks) with x and the scales two chunks ahead were measured
  // and change nothing: these launches are bound by their fixed costs on narrow matrices and by instruction issue on
  // wide ones, not by the time a load is in flight.
  constexpr int R = 4;                           // ring of the wave's weight units per block of columns: two chunks ahead
  constexpr int kAheadX = 1;                     // chunks of x held in registers ahead of the shared image
  constexpr int GH = 2 / GU;                     // groups of a chunk this wave takes (GU = units of a group: 1 or 2)
  constexpr int kUnroll = 2;                     // chunks per trip of the main loop: every index above is static inside
  constexpr int kThreads = 512;
  constexpr int kChunkBytes = BITS == 4 ? 512 : 256;      // bytes of one x row per staged chunk (32 packed rows: 256 or 128 k)
  constexpr int kPieces = kChunkBytes / 16;
  constexpr int kStepsPerUnit = BITS == 4 ? 2 : 1;        // k-steps (32 k) of a unit
  constexpr int NF = 2 * NP;                     // 16-column fragments of the wave
  constexpr int kRowsT = 16 * MT;
  constexpr int kPitch = 128 * NPT + 4;          // floats per row of the epilogue image (4 waves x 32 NPT columns)
  constexpr int kBufBytes = kRowsT * kChunkBytes;
  constexpr int kWaveImg = 2 * NPT * 1024;       // a wave's weight images: two alternating sets of NPT units

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = wave >> 2;                    // which of a chunk's two groups this wave takes
  const int lane = tid & 63;
  const int n = lane & 15, cq = lane >> 4;
  char* wimg = smem + 2 * kBufBytes + wave * kWaveImg;
  const int c_lo = static_cast<int>(static_cast<int64_t>(split) * a.chunks / a.splits);
  const int c_hi = static_cast<int>(static_cast<int64_t>(split + 1) * a.chunks / a.splits);

  // x staging: 16-byte piece idx = j * 256 + tid of the [rows][32 pieces] chunk image; the 8 values of a piece are
  // stored in the order the nibble trick emits
  constexpr int kStagePieces = kRowsT * kPieces;
  constexpr int kStage = (kStagePieces + kThreads - 1) / kThreads;
  const bool stages = tid < kStagePieces;       // 16 rows of 8-bit weights: 256 pieces for 512 threads
  const char* xsrc[kStage];
  int xdst[kStage];
#pragma unroll
  for (int j = 0; j < kStage; ++j) {
    const int idx = min(j * kThreads + tid, kStagePieces - 1);
    const int row = idx / kPieces, piece = idx % kPieces;
    xsrc[j] = a.x + static_cast<int64_t>(min(row, a.M - 1)) * a.x_stride_bytes + piece * 16;
    xdst[j] = row * kChunkBytes + ((piece & 16) | ((piece ^ row) & 15)) * 16;      // (8-bit: 16 pieces a row, piece & 16 == 0)
  }
  int aoff[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) aoff[mt] = (16 * mt + n) * kChunkBytes;

  // weight loads: unit = 8 packed rows x 128 bytes of one 32-column block; lane = (row l / 8, piece l % 8)
  const int64_t row_bytes = static_cast<int64_t>(a.N) * 4;
  const int64_t unit_bytes = 8 * row_bytes;
  const char* wl[NP];
  int wdst;                                         // lane's 16 bytes inside a unit image: piece XOR 4 on odd rows
  {
    const int r = lane >> 3, p = lane & 7;
    wdst = r * 128 + ((p ^ (4 * (r & 1))) * 16);
#pragma unroll
    for (int i = 0; i < NP; ++i)
      wl[i] = reinterpret_cast<const char*>(a.qweight) + (static_cast<int64_t>(c_lo) * 32 + 16 * half + r) * row_bytes +
              (static_cast<int64_t>(wave_b0 + i) * 32 + 4 * p) * 4;
  }
  // fragment reads, 4-bit: k-step s of the unit, fragment f of the block: word (row 4 s + cq, column 16 f + n);
  // 8-bit: the unit is one k-step, the lane's words are (rows 2 cq and 2 cq + 1, column 16 f + n): wsrc[s] = the word of row 2 cq + s
  int wsrc[2][2];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const int r = BITS == 4 ? 4 * s + cq : 2 * cq + s, c = 16 * f + n;
      wsrc[s][f] = r * 128 + (((c >> 2) ^ (4 * (r & 1))) * 16) + (c & 3) * 4;
    }

  // the wave's own units, two per chunk: unit v is unit (v & 1) of its group in chunk v / 2 of the slice
  const int nunits = (c_hi - c_lo) * 2;
  u32x4 ring[R][NP];
  auto issue = [&](int slot, int unit) {
    const int v = min(unit, nunits - 1);
    const int64_t off = static_cast<int64_t>((v >> 1) * kUnitsPerChunk + (v & 1)) * unit_bytes;
#pragma unroll
    for (int i = 0; i < NP; ++i) ring[slot][i] = ld16nt(wl[i] + off);
  };
#pragma unroll
  for (int r = 0; r < R; ++r) issue(r, r);
  __builtin_amdgcn_sched_barrier(0);

  // scales and zeros of the lane's column in each fragment, one group ahead
  const bool f16 = a.scales_f16 != 0;
  const int groups_n8 = a.N * BITS / 32;          // words of a row of qzeros
  // two register sets, one per chunk parity (static in the unrolled chunk loop): a set is loaded a chunk before its
  // own and first read at the end of a group of its own -- no copy, no early wait
  // (kept as loaded -- a 16-bit scale, the packed word of zeros -- and converted where they are used: a conversion next
  // to the load would make the wave wait for it, and for every weight load before it, on the spot)
  uint32_t sc[2][GH][NF], zc[2][GH][NF];
  auto load_groups = [&](int ci, uint32_t (&s_)[GH][NF], uint32_t (&z_)[GH][NF]) {      // ci: chunk, counted from c_lo
    const int cg = min(c_lo + ci, a.chunks - 1);
#pragma unroll
    for (int u = 0; u < GH; ++u) {
      const int gg = (2 * cg + half) * GH + u;          // the chunk's groups in k order: half 0's, then half 1's
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const int col = (wave_b0 + (f >> 1)) * 32 + 16 * (f & 1) + n;
        s_[u][f] = static_cast<const uint16_t*>(a.scales)[static_cast<int64_t>(gg) * a.N + col];
        z_[u][f] = static_cast<const uint32_t*>(a.qzeros)[static_cast<int64_t>(gg) * groups_n8 + (col * BITS / 32)];
      }
    }
  };
  const int zshift = BITS * (n & (32 / BITS - 1));      // the lane's column inside a packed word of zeros (16 f + n: same for every f)
  const float zbase = static_cast<float>((BITS == 4 ? 128 : 0) + a.zero_offset);
  load_groups(0, sc[0], zc[0]);

  f32x4 acc[MT][NF];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int f = 0; f < NF; ++f) acc[mt][f] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 xs[kAheadX][kStage];
  auto stage_load = [&](int set, int c) {
    const int64_t off = static_cast<int64_t>(min(c, c_hi - 1)) * kChunkBytes;
#pragma unroll
    for (int j = 0; j < kStage; ++j) xs[set][j] = ld16g(xsrc[j] + off);
  };
  auto stage_store = [&](int set, char* buf) {
#pragma unroll
    for (int j = 0; j < kStage; ++j) {
      const u32x4 r = xs[set][j];
      u32x4 o = r;
      if constexpr (BITS == 4) {     // (k0,k1)(k2,k3)(k4,k5)(k6,k7) -> (k0,k4)(k1,k5)(k2,k6)(k3,k7)
        o[0] = __builtin_amdgcn_perm(r[2], r[0], 0x05040100u);
        o[1] = __builtin_amdgcn_perm(r[2], r[0], 0x07060302u);
        o[2] = __builtin_amdgcn_perm(r[3], r[1], 0x05040100u);
        o[3] = __builtin_amdgcn_perm(r[3], r[1], 0x07060302u);
      }
      if (kStagePieces >= kThreads || stages) *reinterpret_cast<u32x4*>(buf + xdst[j]) = o;
    }
  };
  stage_load(0, c_lo);
  if constexpr (kAheadX == 2) stage_load(1, c_lo + 1);
  stage_store(0, smem);
  __syncthreads();

  const u32x4 ones = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
  // at the top of chunk i (counted from c_lo): the shared image i & 1 holds x of chunk i, the registers those of chunks
  // i + 1 .. i + kAheadX - 1; chunk i + kAheadX is asked for, and chunk i + 1 goes to the other image at the bottom
  for (int c0 = c_lo; c0 < c_hi; c0 += kUnroll) {
#pragma unroll
    for (int cc = 0; cc < kUnroll; ++cc) {
      const int c = c0 + cc;
      if (c >= c_hi) break;
      const char* cur = smem + ((c - c_lo) & 1) * kBufBytes;
      stage_load(cc % kAheadX, c + kAheadX);
      {
        const int gb = cc % 2;                         // the chunk's set of scale / zero registers
        load_groups(c - c_lo + 1, sc[gb ^ 1], zc[gb ^ 1]);
        __builtin_amdgcn_sched_barrier(0);
        const int unit0 = (c - c_lo) * 2;
        // the group's two units (64 k each) of every block go to the wave's images, and their ring slots are refilled
#pragma unroll
        for (int uu = 0; uu < 2; ++uu) {
          const int slot = (2 * cc + uu) % R;
#pragma unroll
          for (int i = 0; i < NP; ++i) *reinterpret_cast<u32x4*>(wimg + uu * (NPT * 1024) + i * 1024 + wdst) = ring[slot][i];
        }
        // keep the refills where they are (see stream_gemm.hip)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int uu = 0; uu < 2; ++uu) issue((2 * cc + uu) % R, unit0 + uu + R);
        __builtin_amdgcn_sched_barrier(0);
        // one group, and in it one block of 32 columns, at a time (its partial sums are the only ones live); the rows' sums
        // of x over the group come with the first block
#pragma unroll
        for (int u = 0; u < GH; ++u) {
        f32x4 sx[MT];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
          f32x4 gacc[MT][2];
          constexpr int kGroupSteps = GU * kStepsPerUnit;
#pragma unroll
          for (int kg = 0; kg < kGroupSteps; ++kg) {      // the group's k-steps; 4-bit: unit ks / 2, half ks % 2 of it; 8-bit: unit ks
            const int ks = u * kGroupSteps + kg;
            const int uu = ks / kStepsPerUnit, s = ks % kStepsPerUnit;
            const char* region = wimg + uu * (NPT * 1024) + i * 1024;
            u32x4 af[MT];
            const int piece = 4 * (2 * kStepsPerUnit * half + ks) + cq;
            const int poff = ((piece & 16) | ((piece ^ n) & 15)) * 16;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) af[mt] = *reinterpret_cast<const u32x4*>(cur + aoff[mt] + poff);
            if (i == 0) {
#pragma unroll
              for (int mt = 0; mt < MT; ++mt) sx[mt] = mma_bf16(af[mt], ones, kg == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : sx[mt]);
            }
#pragma unroll
            for (int f = 0; f < 2; ++f) {
              u32x4 b;
              if constexpr (BITS == 4) {
                const uint32_t w = *reinterpret_cast<const uint32_t*>(region + wsrc[s][f]);
#pragma unroll
                for (int j = 0; j < 4; ++j) b[j] = ((w >> (4 * j)) & 0x000F000Fu) | 0x43004300u;
              } else {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                  const uint32_t w = *reinterpret_cast<const uint32_t*>(region + wsrc[h][f]);
                  b[2 * h] = pack_bf16x2(static_cast<float>(w & 0xFFu), static_cast<float>((w >> 8) & 0xFFu));
                  b[2 * h + 1] = pack_bf16x2(static_cast<float>((w >> 16) & 0xFFu), static_cast<float>(w >> 24));
                }
              }
#pragma unroll
              for (int mt = 0; mt < MT; ++mt) gacc[mt][f] = mma_bf16(af[mt], b, kg == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : gacc[mt][f]);
            }
          }
          // the group's scale and zero, applied to its partial sums
#pragma unroll
          for (int f2 = 0; f2 < 2; ++f2) {
            const int f = 2 * i + f2;
            const float s_ = f16 ? static_cast<float>(__builtin_bit_cast(_Float16, static_cast<uint16_t>(sc[gb][u][f]))) : __uint_as_float(sc[gb][u][f] << 16);
            const float z_ = zbase + static_cast<float>((zc[gb][u][f] >> zshift) & ((1u << BITS) - 1u));
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
              for (int r = 0; r < 4; ++r) acc[mt][f][r] = fmaf(s_, fmaf(-z_, sx[mt][r], gacc[mt][f2][r]), acc[mt][f][r]);
          }
        }
        }
      }
      stage_store((cc + 1) % kAheadX, smem + (((c - c_lo) & 1) ^ 1) * kBufBytes);
      __syncthreads();
    }
  }

  // ---- epilogue: the tile as a row-major fp32 image in LDS; with K split between the halves there is one image for
  // each, added on the way out (half 0 + half 1)
  constexpr int kImg = kRowsT * kPitch;
  float* img = reinterpret_cast<float*>(smem);
  const int img_col0 = 32 * (wave_b0 - tile_b0);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int r = 0; r < 4; ++r) img[half * kImg + (16 * mt + 4 * cq + r) * kPitch + img_col0 + 16 * f + n] = acc[mt][f][r];
  __syncthreads();
  const int vec_per_row = 8 * tile_nb;
  for (int idx = tid; idx < kRowsT * vec_per_row; idx += kThreads) {
    const int row = idx / vec_per_row, c4 = idx % vec_per_row;
    if (row >= a.M) continue;
    const int col = 32 * tile_b0 + 4 * c4;
    f32x4 y = *reinterpret_cast<const f32x4*>(img + row * kPitch + 4 * c4);
    y += *reinterpret_cast<const f32x4*>(img + kImg + row * kPitch + 4 * c4);
    if (a.splits > 1) {
      *reinterpret_cast<f32x4*>(a.slabs + ((static_cast<int64_t>(split) * a.M + row) * a.N + col) * 4) = y;
      continue;
    }
    if (a.bias != nullptr) {
      const u32x2 bw = *reinterpret_cast<const u32x2*>(a.bias + col);
      y += f32x4{lo_bf(bw[0]), hi_bf(bw[0]), lo_bf(bw[1]), hi_bf(bw[1])};
    }
    const u32x2 o = {pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3])};
    *reinterpret_cast<u32x2*>(a.out + static_cast<int64_t>(row) * a.out_stride + col) = o;
  }
}

template <int MT, int NPHI, int BITS, int GU>
__global__ __launch_bounds__(512, 1) void stream_gptq_kernel(StreamGptqArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = blockIdx.x / a.splits, split = blockIdx.x % a.splits;
  const int blocks = a.N / 32;
  const int b_lo = static_cast<int>(static_cast<int64_t>(t) * blocks / a.tiles);
  const int b_hi = static_cast<int>(static_cast<int64_t>(t + 1) * blocks / a.tiles);
  const int c = b_hi - b_lo;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) & 3;      // waves w and w + 4 share their columns
  const int w_lo = b_lo + wave * c / 4, w_hi = b_lo + (wave + 1) * c / 4;
  if (w_hi - w_lo == NPHI) {
    gptq_wave<MT, NPHI, NPHI, BITS, GU>(a, smem, w_lo, b_lo, c, split);
  } else if constexpr (NPHI > 1) {
    gptq_wave<MT, NPHI - 1, NPHI, BITS, GU>(a, smem, w_lo, b_lo, c, split);
  }
}

// The slices' partial tiles summed in slice order (+ bias) -> bf16; thread = 4 columns of a row.
__global__ __launch_bounds__(256) void stream_gptq_reduce_kernel(StreamGptqArgs a) {
  const int64_t idx = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  const int per_row = a.N / 4;
  if (idx >= static_cast<int64_t>(a.M) * per_row) return;
  const int row = static_cast<int>(idx / per_row), col = 4 * static_cast<int>(idx % per_row);
  f32x4 sum = {0.f, 0.f, 0.f, 0.f};
  for (int sp = 0; sp < a.splits; sp += 4) {
    f32x4 part[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      part[i] = *reinterpret_cast<const f32x4*>(a.slabs + ((static_cast<int64_t>(min(sp + i, a.splits - 1)) * a.M + row) * a.N + col) * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (sp + i < a.splits) sum += part[i];
  }
  if (a.bias != nullptr) {
    const u32x2 bw = *reinterpret_cast<const u32x2*>(a.bias + col);
    sum += f32x4{lo_bf(bw[0]), hi_bf(bw[0]), lo_bf(bw[1]), hi_bf(bw[1])};
  }
  const u32x2 o = {pack_bf16x2(sum[0], sum[1]), pack_bf16x2(sum[2], sum[3])};
  *reinterpret_cast<u32x2*>(a.out + static_cast<int64_t>(row) * a.out_stride + col) = o;
}

template <int MT, int NP, int BITS, int GU>
int launch_gptq(const StreamGptqArgs& a, hipStream_t s, bool defer_reduce) {
  constexpr int rows = 16 * MT;
  constexpr size_t img = 2u * static_cast<size_t>(rows) * (128 * NP + 4) * 4;     // K split: one image per half
  constexpr size_t stream = 2u * rows * (BITS == 4 ? 512 : 256) + 8u * 2 * NP * 1024;
  constexpr size_t lds = img > stream ? img : stream;
  auto kern = stream_gptq_kernel<MT, NP, BITS, GU>;
  static bool configured = false;
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)) != hipSuccess)
      (void)hipGetLastError();
    configured = true;
  }
  hipLaunchKernelGGL(kern, dim3(a.tiles * a.splits), dim3(512), lds, s, a);
  if (a.splits > 1 && !defer_reduce) {
    if (hipGetLastError() != hipSuccess) return HYMETA_ERR_LAUNCH;
    const int64_t threads = static_cast<int64_t>(a.M) * (a.N / 4);
    hipLaunchKernelGGL(stream_gptq_reduce_kernel, dim3(static_cast<unsigned>((threads + 255) / 256)), dim3(256), 0, s, a);
  }
  return launch_status();
}

}  // namespace

bool stream_gptq_plan(int M, int K, int N, int bits, int group_size, bool have_workspace, StreamGptqPlan* p) {
  // 4-bit group 128 or 64 (chunks of 256 k), or 8-bit group 64 (chunks of 128 k): 32 packed rows a chunk either way
  const bool four = bits == 4 && (group_size == 128 || group_size == 64), eight = bits == 8 && group_size == 64;
  if ((!four && !eight) || M <= 0 || M > 64 || K % (four ? 256 : 128) != 0 || N % 128 != 0) return false;
  static const int want_forced = [] { const char* e = getenv("HYMETA_GPTQ_STREAM_WGS"); return e ? atoi(e) : 0; }();
  static const int splits_forced = [] { const char* e = getenv("HYMETA_GPTQ_STREAM_SPLITS"); return e ? atoi(e) : 0; }();
  const int want = want_forced > 0 ? want_forced : 256;
  const int chunks = K / (four ? 256 : 128);
  const int blocks = N / 32;                   // 32-column blocks; a wave owns 1 or 2, a tile 4 to 8
  // Matrices wide enough to give every compute unit a tile without K slices have waves that own two blocks: with four
  // row fragments (33 to 64 rows) those have no registers left for them at two waves per SIMD, and such calls stay on
  // the older kernels (quant_gemm.hip).  Measured on the 7B gate_up matrix: level with them up to 16 rows (24 against
  // 23 us; they keep those), ahead at 17 to 32 (30 against 40 us).  HYMETA_GPTQ_STREAM_WIDE=0 keeps every wide matrix
  // on the older kernels, =2 sends all up to 32 rows here.  (8-bit weights at 33 to 64 rows on one block per wave and
  // 296 tiles, with 1 to 4 K slices: 61 to 72 us on the 7B gate_up matrix against 60 on the older kernel.)
  static const int wide_mode = [] { const char* e = getenv("HYMETA_GPTQ_STREAM_WIDE"); return e == nullptr ? 1 : atoi(e); }();
  const bool wide = blocks / 4 >= want;
  // (4-bit weights of group size 64 at 17 to 32 rows: 35.1 against 32.5 us on the older kernel, which keeps them)
  if (wide && (wide_mode == 0 || M > 32 || (M <= 16 && wide_mode != 2) || (four && group_size == 64 && wide_mode != 2))) return false;
  int tiles = blocks / 4 < want ? blocks / 4 : want;
  if (tiles < (blocks + 7) / 8) tiles = (blocks + 7) / 8;
  const int np = ((blocks + tiles - 1) / tiles + 3) / 4;
  int splits = 1;
  if (have_workspace && tiles < want) splits = want / tiles;
  if (have_workspace && splits_forced > 0) splits = splits_forced;
  if (splits > chunks) splits = chunks;
  if (splits > kMaxSplits) splits = kMaxSplits;
  if (splits < 1) splits = 1;
  p->np = np, p->tiles = tiles, p->chunks = chunks, p->splits = splits, p->bits = bits, p->group_units = group_size * bits / 256;
  return np == 1 || (np == 2 && M <= 32);
}

size_t stream_gptq_workspace_bytes(int M, int N, const StreamGptqPlan& p) {
  return p.splits > 1 ? static_cast<size_t>(p.splits) * M * N * 4 : 0;
}

size_t stream_gptq_workspace_bound(int M, int N) { return static_cast<size_t>(kMaxSplits) * M * N * 4 / (N >= 32768 ? 16 : 1); }

int stream_gptq_launch(StreamGptqArgs a, const StreamGptqPlan& p, hipStream_t s, bool defer_reduce) {
  a.tiles = p.tiles, a.chunks = p.chunks, a.splits = p.splits;
  const int mt = a.M <= 16 ? 1 : (a.M <= 32 ? 2 : 4);
#define HYMETA_GPTQ_CASE(MT_, NP_)                                                             \
  if (mt == MT_ && p.np == NP_)                                                                \
    return p.bits == 8 ? launch_gptq<MT_, NP_, 8, 2>(a, s, defer_reduce)                      \
                       : (p.group_units == 1 ? launch_gptq<MT_, NP_, 4, 1>(a, s, defer_reduce) : launch_gptq<MT_, NP_, 4, 2>(a, s, defer_reduce));
  HYMETA_GPTQ_CASE(1, 1) HYMETA_GPTQ_CASE(1, 2) HYMETA_GPTQ_CASE(2, 1) HYMETA_GPTQ_CASE(2, 2) HYMETA_GPTQ_CASE(4, 1)
#undef HYMETA_GPTQ_CASE
  return HYMETA_ERR_UNSUPPORTED;
}

}  // namespace hymeta
