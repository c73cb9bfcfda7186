// Weight-streaming GEMM for decode-sized batches (up to 64 rows): out = x W^T (+ bias), W [N, K] K-contiguous
// as torch stores a Linear weight (bf16), or int8 with per-channel scales against int8 activations (W8A8).
//
// At these row counts the weights are the traffic and every weight is used once: the kernel is a stream of W
// from HBM through the matrix cores, and everything else is arranged so that nothing competes with that stream.
//   * W is read in whole 128-byte lines: one load instruction covers 8 weight rows x 128 bytes (lane = row l / 8,
//     16-byte piece l % 8).  A compute unit keeps about 64 line requests in flight whatever their size, and a
//     load in MFMA fragment shape (16 rows x 64 bytes, a lane's neighbours in other rows) asks for half lines:
//     measured on the 7B gate_up matrix, the bare stream runs at 5.3-6.1 TB/s in whole lines against 3.3-4.3 in
//     fragment shape (tools/membw2.hip).  The lines pass through a wave-private LDS image (ds_write_b128 as
//     loaded, ds_read_b128 as B fragments; pieces XOR-ed with the row so that both are conflict free); each wave
//     keeps a ring of R pairs of k-steps in flight in registers, issued unconditionally so that the compiler can
//     count them (a pair past the end re-reads the last one).
//   * The activations are shared: a workgroup of 4 waves owns 64 rows x (4 x 16 NF) columns, every wave the
//     same k range and its own columns.  x is staged through LDS in full 128-byte lines, 256 k-values (one
//     "chunk" = 8 k-steps) of all rows per buffer, two buffers, one barrier per chunk; fragment-shaped loads
//     of x straight from L2 (16 rows x 64 B per instruction) would double the address-path work of a kernel
//     whose weight loads already have that shape.  16-byte pieces are XOR-ed with the row inside the LDS image,
//     which makes both the staging stores and the fragment reads (ds_read_b128) conflict free.
//   * A workgroup owns one (column tile, K slice).  With one slice it writes the tile; with more, the slices' fp32
//     (int32) partial tiles go to slabs [slice][M][N] and a second small kernel sums them in slice order, adds the
//     bias and writes bf16.  (Tickets and an in-launch last-arriver sum were measured first: at 256-column tiles the
//     agent-scope release / acquire per hand-off and the serial slab reads cost more than the launch they save,
//     99 against 63 us on the 7B gate_up matrix.)  One compute unit pulls 29-35 GB/s of weights with this kernel, so
//     about 150 workgroups already draw what HBM delivers; the host side picks the widest tile (x is re-read once
//     per tile) and the fewest slices that give that many.
#include <cstdlib>

#include "stream_gemm.h"

namespace hymeta {
namespace {

constexpr int kChunkSteps = 8;                 // k-steps (of 64 bytes per row) per staged chunk
constexpr int kChunkBytes = 64 * kChunkSteps;  // 512 bytes of one row
constexpr int kPieces = kChunkBytes / 16;      // 16-byte pieces per row of a chunk
constexpr int kMaxSplits = 16;


template <bool I8> struct Acc { typedef f32x4 type; };
template <> struct Acc<true> { typedef i32x4 type; };

template <bool I8>
__device__ __forceinline__ typename Acc<I8>::type mma(const u32x4& a, const u32x4& b, const typename Acc<I8>::type& c) {
  if constexpr (I8) {
    return __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, a), __builtin_bit_cast(i32x4, b), c, 0, 0, 0);
  } else {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
}

// One wave's part of a workgroup's tile.  The tile is fragments [tile_f0, tile_f0 + tile_nf) of the matrix (a fragment =
// 16 weight rows), this wave owns NF of them from ``wave_f0`` on; the four waves of a workgroup may own NF or NF - 1 each
// (two instantiations, chosen per wave), share the staged x and meet at the same barriers.  K slice ``split``.
// MT: 16-row fragments (rows = 16 MT); NFT: the widest wave of the launch (sizes the shared images); R: ring depth.
template <int MT, int NF, int NFT, int R, bool I8, bool NT>
__device__ __forceinline__ void stream_wave(const StreamArgs& a, char* smem, const int wave_f0, const int tile_f0,
                                            const int tile_nf, const int split) {
  typedef typename Acc<I8>::type acc_t;
  constexpr int kRowsT = 16 * MT;
  constexpr int kPitch = 64 * NFT + 4;           // floats per row of the epilogue image
  constexpr int kBufBytes = kRowsT * kChunkBytes;
  constexpr int kBufs = MT > 4 ? 1 : 2;            // x chunk images: 128 rows leave room for one (a second barrier per chunk instead)
  constexpr int kWaveImg = 2 * NFT * 2048;         // a wave's two weight images (one pair of k-steps each)

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int n = lane & 15, cq = lane >> 4;
  char* wimg = smem + kBufs * kBufBytes + wave * kWaveImg;
  const int c_lo = static_cast<int>(static_cast<int64_t>(split) * a.chunks / a.splits);
  const int c_hi = static_cast<int>(static_cast<int64_t>(split + 1) * a.chunks / a.splits);
  // The last chunk of a row may be partial (K a multiple of 128 bytes, not of 512): x is staged as zeros past the
  // end of the row and the weight loads of those k-steps re-read the row's last whole line, so they add nothing.
  const int piece_off = (tid & (kPieces - 1)) * 16;      // this thread's 16-byte piece inside a chunk (the same for every j)

  // staging geometry: 16-byte piece idx = j * 256 + tid of the [rows][32 pieces] chunk image
  constexpr int kStage = kRowsT * kPieces / 256;   // pieces per thread
  const char* xsrc[kStage];
  int xdst[kStage];
#pragma unroll
  for (int j = 0; j < kStage; ++j) {
    const int idx = j * 256 + tid;
    const int row = idx / kPieces, piece = idx % kPieces;
    xsrc[j] = a.x + static_cast<int64_t>(min(row, a.M - 1)) * a.x_stride_bytes + piece * 16;
    xdst[j] = row * kChunkBytes + ((piece & 16) | ((piece ^ row) & 15)) * 16;
  }
  // fragment reads: row 16 mt + n, piece 4 s + cq
  int aoff[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) aoff[mt] = (16 * mt + n) * kChunkBytes;

  {
    const int nsteps = (c_hi - c_lo) * kChunkSteps;
    // plain: the wave's 16 NF weight rows are consecutive output columns.  Gated: fragment i holds 8 gate rows and the
    // 8 up rows of the same output columns (rows 8 j .. 8 j + 7 of a fragment are sub-load j), so that a lane's
    // neighbour 8 columns on holds the other factor and the product is formed in the epilogue image.
    const int col0 = 16 * wave_f0;
    const int ocol0 = 8 * wave_f0;                           // gated: first output column of the wave
    const int up0 = a.N / 2;
    const int img_col0 = 16 * (wave_f0 - tile_f0);           // the wave's first column in the tile's epilogue image

    // weight loads: group i = 16 rows, sub j = its rows 8j .. 8j+7; lane = (row l / 8, piece l % 8) of a 128-byte line.
    // The lane's piece is XOR-ed with the row (swz) so that the lane-linear LDS image serves conflict-free fragment reads.
    const char* wl[NF][2];
#pragma unroll
    for (int i = 0; i < NF; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int r = 8 * j + (lane >> 3);
        const int wrow = a.gated ? (j ? up0 : 0) + ocol0 + 8 * i + (lane >> 3) : col0 + 16 * i + r;
        wl[i][j] = a.w + static_cast<int64_t>(wrow) * a.w_stride_bytes + static_cast<int64_t>(c_lo) * kChunkBytes +
                   (((lane & 7) ^ ((r >> 1) & 7)) * 16);
      }

    acc_t acc[MT][NF];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int i = 0; i < NF; ++i) acc[mt][i] = acc_t{0, 0, 0, 0};

    // ring of R slots, each a PAIR of k-steps = one 128-byte line of each of the wave's 16 NF rows
    u32x4 ring[R][2 * NF];
    const int npairs = min(nsteps / 2, (a.k_bytes - c_lo * kChunkBytes) / 128);     // whole lines of this slice
    auto issue = [&](int slot, int pair) {
      const int off = 128 * min(pair, npairs - 1);
#pragma unroll
      for (int i = 0; i < NF; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) ring[slot][2 * i + j] = NT ? ld16nt(wl[i][j] + off) : ld16g(wl[i][j] + off);
    };
#pragma unroll
    for (int r = 0; r < R; ++r) issue(r, r);
    __builtin_amdgcn_sched_barrier(0);

    // Deep rings (R > 4) run kUnroll chunks per trip of the loop so that ring slots stay compile-time registers, and
    // hold two chunks of x in registers ahead of the shared image (kAhead), so that neither the weights nor x are
    // waited for a whole memory latency per chunk when a slice has only a few chunks.
    constexpr int kUnroll = R > 4 ? R / 4 : 1;
    constexpr int kAhead = R > 4 ? 2 : 1;
    u32x4 xs[kAhead][kStage];
    auto stage_load = [&](int slot, int c) {
      const int off = min(c, c_hi - 1) * kChunkBytes;
      const bool inside = off + piece_off < a.k_bytes;
#pragma unroll
      for (int j = 0; j < kStage; ++j) xs[slot][j] = inside ? ld16g(xsrc[j] + off) : u32x4{0, 0, 0, 0};
    };
    auto stage_store = [&](int slot, char* buf) {
#pragma unroll
      for (int j = 0; j < kStage; ++j) *reinterpret_cast<u32x4*>(buf + xdst[j]) = xs[slot][j];
    };
    stage_load(0, c_lo);
    if constexpr (kAhead == 2) stage_load(1, c_lo + 1);
    stage_store(0, smem);
    __syncthreads();

    // at the top of chunk i (counted from c_lo): the shared image i & 1 holds chunk i, the registers chunks i + 1 ..
    // i + kAhead - 1; chunk i + kAhead is asked for, and chunk i + 1 goes to the other image at the bottom
    for (int c0 = c_lo; c0 < c_hi; c0 += kUnroll) {
#pragma unroll
      for (int cc = 0; cc < kUnroll; ++cc) {
        const int c = c0 + cc;
        if (c < c_hi) {
          const char* cur = smem + ((c - c_lo) & (kBufs - 1)) * kBufBytes;
          stage_load(kAhead == 2 ? (cc & 1) : 0, c + kAhead);
          __builtin_amdgcn_sched_barrier(0);
          const int pair0 = (c - c_lo) * (kChunkSteps / 2);
#pragma unroll
          for (int p = 0; p < kChunkSteps / 2; ++p) {
            const int slot = (cc * (kChunkSteps / 2) + p) % R;
            char* region = wimg + (p & 1) * (NF * 2048);
#pragma unroll
            for (int q = 0; q < 2 * NF; ++q) *reinterpret_cast<u32x4*>(region + q * 1024 + lane * 16) = ring[slot][q];
            // the scheduler must not sink the refill towards its use a whole ring later (it would, to save registers,
            // and every pair would then wait out a full memory latency)
            __builtin_amdgcn_sched_barrier(0);
            issue(slot, pair0 + p + R);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              u32x4 af[MT], bfr[NF];
              const int piece = 4 * (2 * p + h) + cq;
              const int poff = ((piece & 16) | ((piece ^ n) & 15)) * 16;
#pragma unroll
              for (int mt = 0; mt < MT; ++mt) af[mt] = *reinterpret_cast<const u32x4*>(cur + aoff[mt] + poff);
              const int woff = n * 128 + (((4 * h + cq) ^ ((n >> 1) & 7)) * 16);
#pragma unroll
              for (int i = 0; i < NF; ++i) bfr[i] = *reinterpret_cast<const u32x4*>(region + i * 2048 + woff);
#pragma unroll
              for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int i = 0; i < NF; ++i) acc[mt][i] = mma<I8>(af[mt], bfr[i], acc[mt][i]);
            }
          }
          if constexpr (kBufs == 1) __syncthreads();       // every wave is done with the one image
          stage_store(kAhead == 2 ? ((cc + 1) & 1) : 0, smem + (((c - c_lo) + 1) & (kBufs - 1)) * kBufBytes);
          __syncthreads();
        }
      }
    }

    // ---- epilogue: the tile as a row-major 4-byte image in LDS
    float* img = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int i = 0; i < NF; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v;
          if constexpr (I8) v = __int_as_float(acc[mt][i][r]); else v = acc[mt][i][r];
          img[(16 * mt + 4 * cq + r) * kPitch + img_col0 + 16 * i + n] = v;
        }
    __syncthreads();
    if (a.gated) {
      // image column 16 f + n of fragment f: n < 8 the gate of output column 8 f + n, n >= 8 the up factor of 8 f + n - 8
      for (int idx = tid; idx < kRowsT * 2 * tile_nf; idx += 256) {
        const int row = idx / (2 * tile_nf), v = idx % (2 * tile_nf);     // v: 4-column vector of the tile's 8 tile_nf output columns
        if (row >= a.M) continue;
        const int f = v >> 1, within = 4 * (v & 1);
        const acc_t g4 = *reinterpret_cast<const acc_t*>(img + row * kPitch + 16 * f + within);
        const acc_t u4 = *reinterpret_cast<const acc_t*>(img + row * kPitch + 16 * f + 8 + within);
        const int ocol = 8 * (tile_f0 + f) + within;
        float g[4], u[4];
        if constexpr (I8) {
          const float xs_ = a.x_scale[row];
          const f32x4 wg = *reinterpret_cast<const f32x4*>(a.w_scale + ocol);
          const f32x4 wu = *reinterpret_cast<const f32x4*>(a.w_scale + up0 + ocol);
#pragma unroll
          for (int e = 0; e < 4; ++e) g[e] = w8a8_scaled(g4[e], xs_, wg[e], 0.f), u[e] = w8a8_scaled(u4[e], xs_, wu[e], 0.f);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) g[e] = g4[e], u[e] = u4[e];
        }
        // both factors rounded to bf16 first, as the two-kernel form (linear, then silu_and_mul) holds them
        const uint32_t gp[2] = {pack_bf16x2(g[0], g[1]), pack_bf16x2(g[2], g[3])};
        const uint32_t up[2] = {pack_bf16x2(u[0], u[1]), pack_bf16x2(u[2], u[3])};
        u32x2 o;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const float g0 = lo_bf(gp[e]), g1 = hi_bf(gp[e]);
          o[e] = pack_bf16x2(g0 * sigmoidf(g0) * lo_bf(up[e]), g1 * sigmoidf(g1) * hi_bf(up[e]));
        }
        *reinterpret_cast<u32x2*>(a.out + static_cast<int64_t>(row) * a.out_stride + ocol) = o;
      }
      return;
    }
    const int vec_per_row = 4 * tile_nf;                     // 4-column vectors per row of the tile
    for (int idx = tid; idx < kRowsT * vec_per_row; idx += 256) {
      const int row = idx / vec_per_row, c4 = idx % vec_per_row;
      if (row >= a.M) continue;
      const int col = 16 * tile_f0 + 4 * c4;
      const acc_t sum = *reinterpret_cast<const acc_t*>(img + row * kPitch + 4 * c4);
      if (a.splits > 1) {
        *reinterpret_cast<acc_t*>(a.slabs + ((static_cast<int64_t>(split) * a.M + row) * a.N + col) * 4) = sum;
        continue;
      }
      f32x4 b = {0.f, 0.f, 0.f, 0.f};
      if (a.bias != nullptr) {
        const u32x2 bw = *reinterpret_cast<const u32x2*>(a.bias + col);
        b = f32x4{lo_bf(bw[0]), hi_bf(bw[0]), lo_bf(bw[1]), hi_bf(bw[1])};
      }
      f32x4 y;
      if constexpr (I8) {
        const float xs_ = a.x_scale[row];
        const f32x4 ws = *reinterpret_cast<const f32x4*>(a.w_scale + col);
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = w8a8_scaled(sum[e], xs_, ws[e], b[e]);
      } else {
        y = sum + b;
      }
      if (a.out_f32) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.out) + static_cast<int64_t>(row) * a.out_stride + col) = y;
        continue;
      }
      const u32x2 o = {pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3])};
      *reinterpret_cast<u32x2*>(a.out + static_cast<int64_t>(row) * a.out_stride + col) = o;
    }
  }
}

// The matrix's N / 16 fragments are dealt to the launch's tiles in equal shares (they differ by at most one fragment), and
// a tile's to its four waves likewise: a matrix of any multiple of 64 columns is cut into a chosen number of workgroups
// (one per compute unit) whose loads differ by a sixteenth of a tile at most.  NFHI = the widest wave's fragments.
template <int MT, int NFHI, int R, bool I8, bool NT>
__global__ __launch_bounds__(256, ((NFHI == 4 || MT * NFHI >= 8) ? 1 : 2)) void stream_gemm_kernel(StreamArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = blockIdx.x / a.splits, split = blockIdx.x % a.splits;
  const int frags = a.N / 16;
  const int f_lo = static_cast<int>(static_cast<int64_t>(t) * frags / a.tiles);
  const int f_hi = static_cast<int>(static_cast<int64_t>(t + 1) * frags / a.tiles);
  const int c = f_hi - f_lo;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int w_lo = f_lo + wave * c / 4, w_hi = f_lo + (wave + 1) * c / 4;
  if (w_hi - w_lo == NFHI) {
    stream_wave<MT, NFHI, NFHI, R, I8, NT>(a, smem, w_lo, f_lo, c, split);
  } else if constexpr (NFHI > 1) {
    stream_wave<MT, NFHI - 1, NFHI, R, I8, NT>(a, smem, w_lo, f_lo, c, split);
  }
}

// Sum of the K slices' partial tiles, in slice order; one thread per 4 columns of a row.
template <bool I8>
__global__ __launch_bounds__(256) void stream_gemm_reduce_kernel(StreamArgs a) {
  typedef typename Acc<I8>::type acc_t;
  const int64_t idx = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  const int per_row = a.N / 4;
  if (idx >= static_cast<int64_t>(a.M) * per_row) return;
  const int row = static_cast<int>(idx / per_row), col = 4 * static_cast<int>(idx % per_row);
  acc_t sum = acc_t{0, 0, 0, 0};
  // four slices' loads in flight at a time; the adds stay in slice order
  for (int sp = 0; sp < a.splits; sp += 4) {
    acc_t part[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      part[i] = *reinterpret_cast<const acc_t*>(a.slabs + ((static_cast<int64_t>(min(sp + i, a.splits - 1)) * a.M + row) * a.N + col) * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (sp + i < a.splits) sum += part[i];
  }
  f32x4 b = {0.f, 0.f, 0.f, 0.f};
  if (a.bias != nullptr) {
    const u32x2 bw = *reinterpret_cast<const u32x2*>(a.bias + col);
    b = f32x4{lo_bf(bw[0]), hi_bf(bw[0]), lo_bf(bw[1]), hi_bf(bw[1])};
  }
  f32x4 y;
  if constexpr (I8) {
    const float xs_ = a.x_scale[row];
    const f32x4 ws = *reinterpret_cast<const f32x4*>(a.w_scale + col);
#pragma unroll
    for (int e = 0; e < 4; ++e) y[e] = w8a8_scaled(sum[e], xs_, ws[e], b[e]);
  } else {
    y = sum + b;
  }
  const u32x2 o = {pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3])};
  *reinterpret_cast<u32x2*>(a.out + static_cast<int64_t>(row) * a.out_stride + col) = o;
}

template <int MT, int NF, bool I8, int RING>
int launch_stream(const StreamArgs& a, hipStream_t s, bool defer_reduce) {
  constexpr int R = RING;     // pairs of k-steps in flight per wave
  constexpr size_t img = static_cast<size_t>(16 * MT) * (64 * NF + 4) * 4;                 // epilogue image
  constexpr size_t stream = (MT > 4 ? 1u : 2u) * 16 * MT * kChunkBytes + 4u * 2 * NF * 2048;     // x chunks + the waves' weight images
  constexpr size_t lds = img > stream ? img : stream;
  auto kern = stream_gemm_kernel<MT, NF, R, I8, true>;
  static bool configured = false;
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)) != hipSuccess)
      (void)hipGetLastError();
    configured = true;
  }
  hipLaunchKernelGGL(kern, dim3(a.tiles * a.splits), dim3(256), lds, s, a);
  if (a.splits > 1 && !defer_reduce) {
    if (hipGetLastError() != hipSuccess) return HYMETA_ERR_LAUNCH;
    const int64_t threads = static_cast<int64_t>(a.M) * (a.N / 4);
    hipLaunchKernelGGL(stream_gemm_reduce_kernel<I8>, dim3(static_cast<unsigned>((threads + 255) / 256)), dim3(256), 0, s, a);
  }
  return launch_status();
}

}  // namespace

// Tuning overrides, read once: HYMETA_STREAM_NF (uniform tile width, 1..4), HYMETA_STREAM_WGS (workgroups to aim for),
// HYMETA_STREAM_SPLITS (K slices), HYMETA_STREAM_RING (2 or 4).
static int tune(const char* name) {
  const char* e = getenv(name);
  return e == nullptr ? 0 : atoi(e);
}

bool stream_gemm_plan(int M, int K_bytes, int N, bool have_workspace, StreamPlan* p) {
  p->tall = 0;
  if (tall_gemm_plan(M, K_bytes, N, have_workspace, p)) return true;
  if (M <= 0 || M > 128 || K_bytes % 128 != 0 || K_bytes < kChunkBytes || N % 64 != 0) return false;
  static const int nf_forced = tune("HYMETA_STREAM_NF"), splits_forced = tune("HYMETA_STREAM_SPLITS");
  static const int ring_forced = tune("HYMETA_STREAM_RING"), want_forced = tune("HYMETA_STREAM_WGS");
  const int chunks = (K_bytes + kChunkBytes - 1) / kChunkBytes;
  const int groups = N / 64;
  // One workgroup per compute unit draws the most: a compute unit pulls 29-35 GB/s of weights with this kernel, HBM
  // delivers what about 190 of them ask for, and a second workgroup on a unit only lengthens the launch.
  const int want = want_forced > 0 ? want_forced : 256;
  const int frags = N / 16;
  int tiles, nf;
  if (nf_forced > 0) {                       // uniform tiles of the forced width
    if (groups % nf_forced != 0) return false;
    tiles = groups / nf_forced, nf = nf_forced;
  } else if (groups >= want) {
    // wide matrix: `want` tiles, the fragments dealt out in equal shares; when a wave would pass 4 fragments, the next
    // multiple of `want`: one workgroup runs per compute unit, so the tiles go by in rounds, and a last round that
    // fills a third of the chip (the 152064-row lm_head as 594 tiles of 16 fragments) costs as much as a full one
    const int min_tiles = (frags + 15) / 16;
    tiles = want >= min_tiles ? want : (min_tiles + want - 1) / want * want;
    nf = ((frags + tiles - 1) / tiles + 3) / 4;
  } else {
    // narrow matrix: K slices make up the count.  Tiles of 128 columns for the large ones (x is re-read once per
    // tile, the slabs grow with the slices); 64 columns for the small ones, whose time is mostly fixed cost and
    // which then need the fewest slices.  Measured on the 7B shapes at 16 and 64 rows.
    nf = static_cast<int64_t>(N) * K_bytes >= (64ll << 20) && groups % 2 == 0 ? 2 : 1;
    if (!have_workspace) nf = 1;
    tiles = groups / nf;
  }
  int splits = 1;
  if (have_workspace && tiles < want) splits = want / tiles;
  if (splits > (chunks + 1) / 2) splits = (chunks + 1) / 2;      // slices of at least two chunks
  if (have_workspace && splits_forced > 0) splits = splits_forced;
  if (splits > chunks) splits = chunks;
  if (splits > kMaxSplits) splits = kMaxSplits;
  if (splits < 1) splits = 1;
  p->nf = nf, p->tiles = tiles, p->chunks = chunks, p->splits = splits;
  // bytes in flight per wave = ring x fragments x 2 KB: one-fragment tiles need the deeper ring to cover the memory latency
  // (rings of 8 and 16 pairs with x two chunks ahead -- the kUnroll / kAhead forms of the main loop -- were measured on the
  // narrow 7B matrices: 1 to 7 us slower, the launches are bound by their fixed costs, not by loads in flight)
  p->ring = ring_forced == 2 || ring_forced == 4 ? ring_forced : (p->nf == 1 ? 4 : 2);
  return p->nf <= 4 && tiles <= groups;      // every wave of every tile owns at least one fragment
}

bool stream_gemm_plan_gated(int M, int K_bytes, int N, StreamPlan* p) {
  return N % 128 == 0 && stream_gemm_plan(M, K_bytes, N, false, p) && p->splits == 1;
}

size_t stream_gemm_workspace_bytes(int M, int N, const StreamPlan& p) {
  return p.splits > 1 ? static_cast<size_t>(p.splits) * M * N * 4 : 0;
}

size_t stream_gemm_workspace_bound(int M, int N) {
  return static_cast<size_t>(kMaxSplits) * M * N * 4 / (N >= 16384 ? 4 : 1);   // wide matrices never take more than kMaxSplits / 4 slices
}

int stream_gemm_reduce_launch(const StreamArgs& a, hipStream_t s) {
  const int64_t threads = static_cast<int64_t>(a.M) * (a.N / 4);
  hipLaunchKernelGGL(stream_gemm_reduce_kernel<false>, dim3(static_cast<unsigned>((threads + 255) / 256)), dim3(256), 0, s, a);
  return launch_status();
}

template <bool I8>
int stream_gemm_launch(StreamArgs a, const StreamPlan& p, hipStream_t s, bool defer_reduce) {
  a.tiles = p.tiles, a.chunks = p.chunks, a.splits = p.splits;
  if (p.tall) {
    if (a.gated && (p.splits != 1 || a.bias != nullptr)) return HYMETA_ERR_INVALID_ARG;
    if (a.out_f32 && (p.splits != 1 || a.gated)) return HYMETA_ERR_INVALID_ARG;
    const int rc = tall_gemm_launch<I8>(a, p, s);
    if (rc != HYMETA_OK || a.splits == 1 || defer_reduce) return rc;
    const int64_t threads = static_cast<int64_t>(a.M) * (a.N / 4);
    hipLaunchKernelGGL(stream_gemm_reduce_kernel<I8>, dim3(static_cast<unsigned>((threads + 255) / 256)), dim3(256), 0, s, a);
    return launch_status();
  }
  if (a.k_bytes <= 0 || (a.k_bytes + kChunkBytes - 1) / kChunkBytes != p.chunks) return HYMETA_ERR_INVALID_ARG;
  if (a.gated && (p.splits != 1 || a.bias != nullptr || a.N % 128 != 0)) return HYMETA_ERR_INVALID_ARG;
  if (a.out_f32 && (p.splits != 1 || a.gated)) return HYMETA_ERR_INVALID_ARG;
  const int mt = a.M <= 16 ? 1 : (a.M <= 32 ? 2 : (a.M <= 64 ? 4 : 8));
#define HYMETA_STREAM_CASE(MT_, NF_) \
  if (mt == MT_ && p.nf == NF_) {                                                                   \
    return p.ring == 4 ? launch_stream<MT_, NF_, I8, 4>(a, s, defer_reduce) : launch_stream<MT_, NF_, I8, 2>(a, s, defer_reduce); \
  }
  HYMETA_STREAM_CASE(1, 1) HYMETA_STREAM_CASE(1, 2) HYMETA_STREAM_CASE(1, 3) HYMETA_STREAM_CASE(1, 4)
  HYMETA_STREAM_CASE(2, 1) HYMETA_STREAM_CASE(2, 2) HYMETA_STREAM_CASE(2, 3) HYMETA_STREAM_CASE(2, 4)
  HYMETA_STREAM_CASE(4, 1) HYMETA_STREAM_CASE(4, 2) HYMETA_STREAM_CASE(4, 3) HYMETA_STREAM_CASE(4, 4)
  HYMETA_STREAM_CASE(8, 1) HYMETA_STREAM_CASE(8, 2) HYMETA_STREAM_CASE(8, 3) HYMETA_STREAM_CASE(8, 4)
#undef HYMETA_STREAM_CASE
  return HYMETA_ERR_UNSUPPORTED;
}

template int stream_gemm_launch<false>(StreamArgs, const StreamPlan&, hipStream_t, bool);
template int stream_gemm_launch<true>(StreamArgs, const StreamPlan&, hipStream_t, bool);

}  // namespace hymeta

extern "C" int hymeta_gemm_partials_reduce(const hymeta_gemm_partials* partials, void* out, int64_t out_stride_m,
                                           hymeta_stream_t stream) {
  using namespace hymeta;
  if (!hymeta::args_ok(partials) || out == nullptr) return HYMETA_ERR_INVALID_ARG;
  const hymeta_gemm_partials& p = *partials;
  if (p.splits < 1 || p.M <= 0 || p.N <= 0 || p.N % 4 != 0 || !p.slabs) return HYMETA_ERR_INVALID_ARG;
  if (p.is_int32 && (!p.x_scale || !p.w_scale)) return HYMETA_ERR_INVALID_ARG;
  if (out_stride_m % 4 != 0 || reinterpret_cast<uintptr_t>(out) % 8 != 0) return HYMETA_ERR_UNSUPPORTED;
  StreamArgs a = {};
  a.slabs = const_cast<char*>(static_cast<const char*>(p.slabs));
  a.bias = static_cast<const uint16_t*>(p.bias), a.out = static_cast<uint16_t*>(out);
  a.x_scale = p.x_scale, a.w_scale = p.w_scale;
  a.M = p.M, a.N = p.N, a.splits = p.splits, a.out_stride = out_stride_m;
  const int64_t threads = static_cast<int64_t>(a.M) * (a.N / 4);
  const dim3 grid(static_cast<unsigned>((threads + 255) / 256));
  hipStream_t s = begin_launches(stream);
  if (p.is_int32) hipLaunchKernelGGL(stream_gemm_reduce_kernel<true>, grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL(stream_gemm_reduce_kernel<false>, grid, dim3(256), 0, s, a);
  return launch_status();
}
