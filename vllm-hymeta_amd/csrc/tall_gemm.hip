// Weight-streaming GEMM for 129 to 256 rows: out = x W^T (+ bias), W [N, K] K-contiguous (bf16, or int8 against int8 rows),
// or 4- / 8-bit GPTQ weights (qweight int32 [K bits / 32, N], group size 64 or 128; 8-bit from 65 rows on).
//
// At these row counts a weight is still read once, but each one now feeds 256 multiply-adds: the matrix cores and the
// HBM stream are both near their limits (one compute unit's share of HBM, 31 GB/s, is 8 TFLOP/s of its 9.8 at 256
// rows), and the registers that hold weights or activations on their way to LDS in the decode-sized kernel are needed
// for accumulators.  So nothing is staged through registers here:
//   * both operands go from memory straight into LDS (global_load_lds, 16 bytes per lane, 1 KB per instruction), in
//     whole 128-byte lines: one instruction covers 8 rows x 128 bytes (lane = row l / 8, piece l % 8).  The LDS image
//     of an instruction is lane-linear; the 16-byte pieces are XOR-ed with the row on the SOURCE address, which makes
//     the fragment reads (ds_read_b128, 16 rows x 64 bytes) conflict free.  The unit of the pipeline is a "pair" of
//     k-steps = one 128-byte line of every row of x (32 KB at 256 rows) and of the tile's weight rows (2 KB a fragment);
//   * a workgroup of 8 waves (two per SIMD: one wave's waits at the barrier, on loads and on LDS are the other's MFMA
//     time) owns all rows x (16 nf <= 160) columns x a K slice.  Wave w owns the 16-row fragments w % 4, w % 4 + 4, ...
//     x one half of the tile's columns and reads both from shared images: per k-step 4 A and 5 B fragments for 20
//     MFMAs, 9 LDS reads where waves that split only the columns would need 18;
//   * x is two pairs deep in LDS, the weights three: at the top of pair g every wave waits (counted vmcnt) until
//     everything but the newest weight pair has landed, meets the others at one barrier -- which also says that all of
//     them are done with pair g - 1 -- and only then asks for x of pair g + 1 and the weights of pair g + 2 into the
//     images pair g - 1 used.  x comes from L2 (one pair of latency), the weights from HBM (two).  Past the end of the
//     slice the same instructions re-load the last pair into an idle image, so that the count the wait relies on is the
//     same in every trip.
//   * GPTQ weights (GQ = their bits) do not come by LDS-DMA: a pair of a 160-column tile is 5 or 10 KB of packed words, which
//     the workgroup's threads load (one pair ahead, three 8-k pieces a thread), expand to bf16(scale * (q - zero)) -- the
//     values of the whole-matrix expansion -- and write into one of two weight images behind the barrier, one word per
//     step; the MFMA loop is the bf16 one.
//   * the matrix's N / 16 fragments are dealt to the tiles in equal shares (9 or 10 to each of 256 workgroups for the 7B
//     gate_up matrix); narrow matrices use 128-column tiles and K slices whose fp32 partial tiles go to slabs
//     [slice][M][N], summed by the reduce kernel or by the consumer (the same contract as the decode-sized kernel).
#include <cstdlib>

#include "stream_gemm.h"

namespace hymeta {
namespace {

constexpr int kWant = 256;      // workgroups to aim for: one per compute unit
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

// 16 bytes per lane from global memory into LDS at (wave-uniform) dst + 16 lane.  Written as an asm statement, with
// the waits for it counted by hand (wait_loads_then_barrier): the compiler books such a load on the LDS counter as
// well, and while one is in flight -- here: always -- every wait for a fragment read becomes a wait for ALL of them.
// (The non-temporal hint on the weight requests, which the decode-sized kernel's register loads gain from, was measured
// here too: level on whole-K tiles, 3 to 5 us slower per launch on the K-sliced 7B shapes.)
__device__ __forceinline__ void glds16(const char* src, char* dst) {
  const uint32_t lds = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)dst));
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds), "v"(src) : "memory");
}

template <int N>
__device__ __forceinline__ void wait_loads_then_barrier() {
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

constexpr int kThreads = 512;    // 8 waves, two per SIMD
constexpr int kMaxNf = 10;       // fragments of the widest tile (sizes the images)
constexpr int kWRing = 3;        // weight pairs in LDS
constexpr int kWInstr = 3;       // weight instructions per wave and pair: 8 x 3 >= 2 kMaxNf

// One wave's part of a workgroup's tile.  The tile is fragments [tile_f0, tile_f0 + nf) of the matrix, K slice `split`;
// the wave owns the 16-row fragments rg, rg + 4, ... (rg = wave % 4) x NFW fragments of one half of the tile's columns
// from `wave_f0` on (counted inside the tile).  MTW: row fragments per wave (rows staged = 64 MTW).
//
// Every wave requests an eighth of each pair's x and weight lines.  The counter a wave waits on counts its requests in
// the order they were made: per pair it asks for x (one pair ahead of its use, from L2) first and then for the weights
// (two pairs ahead, from HBM), so that the wait which opens a pair can leave exactly the newest weights in flight.
template <int MTW, int NFW, bool I8, int GQ = 0>
__device__ __forceinline__ void tall_wave(const StreamArgs& a, char* smem, const int tile_f0, const int nf, const int wave_f0,
                                          const int split) {
  typedef typename Acc<I8>::type acc_t;
  constexpr bool GPTQ = GQ != 0;                // GQ: bits of GPTQ weights (4 or 8), 0: bf16 / int8 weights by LDS-DMA
  constexpr int kRows = 64 * MTW;
  constexpr int kXBuf = kRows * 128;            // one pair of every row
  constexpr int kWSlot = kMaxNf * 2048;         // one pair of the tile's weight rows
  constexpr int kXInstr = MTW;                  // x instructions per wave and pair (8 rows each)
  constexpr int kReq = kXInstr + (GPTQ ? 0 : kWInstr);      // requests per wave and pair
  constexpr int kRing = GPTQ ? 2 : kWRing;      // weight images (GPTQ: expanded by the workgroup itself, one pair ahead)
  constexpr int kItems = 3;                     // GPTQ: packed words a thread expands per pair (8 x 16 kMaxNf <= 3 x 512)
  constexpr int kPitch = 16 * kMaxNf + 4;       // floats per row of the epilogue image
  static_assert(kReq <= 2 * NFW, "one request per step");

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rg = wave & 3;
  const int lane = tid & 63;
  const int n = lane & 15, cq = lane >> 4;
  char* const xbuf = smem;
  char* const wbuf = smem + 2 * kXBuf;

  const int pairs_all = a.k_bytes / 128;
  const int p_lo = static_cast<int>(static_cast<int64_t>(split) * pairs_all / a.splits);
  const int p_hi = static_cast<int>(static_cast<int64_t>(split + 1) * pairs_all / a.splits);
  const int P = p_hi - p_lo;
  auto pair_off = [&](int g) { return 128 * min(g, P - 1); };      // past the end: the last pair again, into an idle image

  // This wave's requests for one pair: source of this lane (row l / 8 of the instruction's 8 rows, piece l % 8 XOR-ed
  // with the row's position in its 16-row fragment) and the instruction's place in the pair's image
  const int sub = lane >> 3;
  const int up0 = a.N / 2;
  const char* src[kReq];
  int dst[kReq];
#pragma unroll
  for (int q = 0; q < kXInstr; ++q) {
    const int t = wave + 8 * q;
    const int row = 8 * t + sub;
    src[q] = a.x + static_cast<int64_t>(min(row, a.M - 1)) * a.x_stride_bytes + static_cast<int64_t>(p_lo) * 128 +
             (((lane & 7) ^ (((row & 15) >> 1) & 7)) * 16);
    dst[q] = t * 1024;
  }
#pragma unroll
  for (int u = 0; u < (GPTQ ? 0 : kWInstr); ++u) {
    const int t = min(wave + 8 * u, 2 * nf - 1);        // a wave past the tile's last instruction repeats it
    const int i = t >> 1, j = t & 1, r = 8 * j + sub;
    // plain: a fragment's 16 weight rows are consecutive output columns.  Gated: 8 gate rows, then the 8 up rows of
    // the same output columns, so that the two factors of an output meet in one row of the epilogue image.
    const int wrow = a.gated ? (j ? up0 : 0) + 8 * (tile_f0 + i) + sub : 16 * (tile_f0 + i) + r;
    src[kXInstr + u] = a.w + static_cast<int64_t>(wrow) * a.w_stride_bytes + static_cast<int64_t>(p_lo) * 128 +
                       (((lane & 7) ^ ((r >> 1) & 7)) * 16);
    dst[kXInstr + u] = __builtin_amdgcn_readfirstlane(t) * 1024;
  }
  // request q of the set a wave makes between the barriers that open pairs g and g + 1: x of pair g + 1 (image
  // (g + 1) % 2), then the weights of pair g + 2 (image (g + 2) % 3) -- the images pair g - 1 used
  auto request = [&](int q, int g) {
    if (q >= kReq) return;
    if (q < kXInstr) {
      glds16(src[q] + pair_off(g + 1), xbuf + ((g + 1) & 1) * kXBuf + dst[q]);
    } else {
      glds16(src[q] + pair_off(g + 2), wbuf + ((g + 2) % kRing) * kWSlot + dst[q]);
    }
  };
  // every wave: until all but its newest weight requests have landed, then until every wave's have -- at which point
  // every wave has also issued its last read of the pair before.  GPTQ: everything the wave has asked for (x of the
  // next pair, the packed words of the one after) and its writes of the expanded pair.
  auto landed = [&]() {
    if constexpr (GPTQ) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else wait_loads_then_barrier<kWInstr>();
  };

  // GPTQ: the tile's packed words of a pair are 8 rows (8 k each) x 16 nf columns; a thread takes up to three of them
  // (item = thread + 512 m: consecutive threads, consecutive columns), expands each into the 16-byte piece of its
  // column's 128-byte line in the weight image -- bf16(scale * (q - zero)), the arithmetic of the whole-matrix expansion
  // -- and asks for the same item of the next pair at once: a word is in flight for one pair.
  constexpr int kWordRows = GQ == 8 ? 16 : 8;      // packed rows of a pair (a word holds 32 / GQ consecutive k of a column)
  uint32_t qw[kItems][GQ == 8 ? 2 : 1], qs[kItems], qz[kItems];
  const char* qsrc[kItems];
  int qcol[kItems], qdst[kItems];
  if constexpr (GPTQ) {
#pragma unroll
    for (int m = 0; m < kItems; ++m) {
      const int item = min(tid + kThreads * m, 8 * 16 * nf - 1);      // threads past the last piece repeat it
      const int j = item / (16 * nf), c = item % (16 * nf);            // piece j (8 k) of the tile's column c
      const int i = c >> 4, r = c & 15;
      qcol[m] = a.gated ? (r < 8 ? 0 : up0) + 8 * (tile_f0 + i) + (r & 7) : 16 * (tile_f0 + i) + r;
      qsrc[m] = a.w + (static_cast<int64_t>(p_lo) * kWordRows + j * (kWordRows / 8)) * a.N * 4 + static_cast<int64_t>(qcol[m]) * 4;
      qdst[m] = c * 128 + ((j ^ ((r >> 1) & 7)) * 16);
    }
  }
  // (offsets as 32-bit element counts from per-item base pointers: a 64-bit multiply per load and a division by the
  // group size per pair would be a third of the loop's instructions)
  const int gshift = a.group_size == 128 ? 1 : 0;      // pairs per group = 1 << gshift
  const int word_step = kWordRows * a.N;               // words from a pair to the next
  const uint32_t* qbase[kItems];
  const uint16_t* sbase[kItems];
  const uint32_t* zbase[kItems];
  if constexpr (GPTQ) {
#pragma unroll
    for (int m = 0; m < kItems; ++m) {
      qbase[m] = reinterpret_cast<const uint32_t*>(qsrc[m]);
      sbase[m] = static_cast<const uint16_t*>(a.scales) + qcol[m];
      zbase[m] = static_cast<const uint32_t*>(a.qzeros) + (qcol[m] * GQ / 32);
    }
  }
  auto ask_item = [&](int m, int g) {        // the words, scale and zeros of item m for pair g
    const int pg = min(g, P - 1);
    const uint32_t* w0 = qbase[m] + static_cast<uint32_t>(pg * word_step);
    qw[m][0] = w0[0];
    if constexpr (GQ == 8) qw[m][1] = w0[a.N];
    const uint32_t grp = static_cast<uint32_t>((p_lo + pg) >> gshift);
    qs[m] = sbase[m][grp * static_cast<uint32_t>(a.N)];
    qz[m] = zbase[m][grp * static_cast<uint32_t>(a.N * GQ / 32)];
  };
  auto expand_item = [&](int m, int g) {     // item m as loaded, into the image of pair g
    const float sc = a.scales_f16 ? static_cast<float>(__builtin_bit_cast(_Float16, static_cast<uint16_t>(qs[m])))
                                  : __uint_as_float(qs[m] << 16);
    constexpr int kPack = GPTQ ? 32 / GQ : 8;
    const float zf = static_cast<float>(((qz[m] >> (GQ * (qcol[m] & (kPack - 1)))) & ((1u << GQ) - 1u)) + a.zero_offset);
    u32x4 o;
    const uint32_t lo4 = qw[m][0] & 0x0F0F0F0Fu, hi4 = (qw[m][0] >> 4) & 0x0F0F0F0Fu;      // 4-bit: values 0,2,4,6 and 1,3,5,7 as bytes
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      uint32_t q0, q1;
      if constexpr (GQ == 8) {
        q0 = (qw[m][e >> 1] >> (16 * (e & 1))) & 0xFFu, q1 = (qw[m][e >> 1] >> (16 * (e & 1) + 8)) & 0xFFu;
      } else {
        q0 = (lo4 >> (8 * e)) & 0xFFu, q1 = (hi4 >> (8 * e)) & 0xFFu;      // byte extracts: one conversion instruction each
      }
      o[e] = pack_bf16x2(sc * (static_cast<float>(q0) - zf), sc * (static_cast<float>(q1) - zf));
    }
    *reinterpret_cast<u32x4*>(wbuf + (g & 1) * kWSlot + qdst[m]) = o;
  };

  // fragment reads: row n of a 16-row fragment, piece 4 h + cq of its 128-byte line
  int roff[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) roff[h] = n * 128 + (((4 * h + cq) ^ ((n >> 1) & 7)) * 16);

  acc_t acc[MTW][NFW];
#pragma unroll
  for (int i = 0; i < MTW; ++i)
#pragma unroll
    for (int f = 0; f < NFW; ++f) acc[i][f] = acc_t{0, 0, 0, 0};

  // A k-step's fragments (MTW of x, NFW of the weights) are read into one of two register sets a whole k-step ahead
  // of their use, so that an LDS read has 4 NFW MFMAs of this wave (and as many of the SIMD's other wave) to arrive.
  // The barrier that opens pair g + 1 stands one step into the second k-step of pair g: the first reads of the new
  // pair go out right behind it, under the remaining MFMAs of the old one.
  u32x4 fa[2][MTW], fb[2][NFW];
  auto reads = [&](int g, int h) {
#pragma unroll
    for (int i = 0; i < MTW; ++i)
      fa[h][i] = *reinterpret_cast<const u32x4*>(xbuf + (g & 1) * kXBuf + (rg + 4 * i) * 2048 + roff[h]);
#pragma unroll
    for (int f = 0; f < NFW; ++f)
      fb[h][f] = *reinterpret_cast<const u32x4*>(wbuf + (g % kRing) * kWSlot + (wave_f0 + f) * 2048 + roff[h]);
  };
  auto mfmas = [&](int h, int f) {
#pragma unroll
    for (int i = 0; i < MTW; ++i) acc[i][f] = mma<I8>(fa[h][i], fb[h][f], acc[i][f]);
  };
  // What a wave does at step f (1 .. NFW - 1) behind the barrier that opens pair g + 1: its requests for x of pair
  // g + 2 (and the weights of pair g + 3), the set the next barrier waits for.  GPTQ: per step one item of pair g + 2 expanded and
  // re-asked for g + 3, then all x requests (the wait at the next barrier is for everything).
  auto behind_barrier = [&](int f, int g) {
    if constexpr (GPTQ) {
      if (f - 1 < kItems) {
        expand_item(f - 1, g + 2);
        ask_item(f - 1, g + 3);
      }
      // x last: the compiler counts only its own loads when it waits for an item's words, and a request of this wave
      // still in flight in front of them would make that wait a wait for x
      if (f == NFW - 1) {
#pragma unroll
        for (int q = 0; q < kXInstr; ++q) request(q, g + 1);
      }
    } else {
      request(f - 1, g + 1);
    }
  };
  static_assert(!GPTQ || kItems <= 3, "NFW - 1 >= 3 steps behind the barrier");
  if constexpr (GPTQ) {
    // before the loop: pair 0 expanded, pair 1 in registers, x of pair 0 asked for
#pragma unroll
    for (int q = 0; q < kXInstr; ++q) request(q, -1);
#pragma unroll
    for (int m = 0; m < kItems; ++m) ask_item(m, 0);
#pragma unroll
    for (int m = 0; m < kItems; ++m) {
      expand_item(m, 0);
      ask_item(m, 1);
    }
    landed();                     // pair 0
#pragma unroll
    for (int f = 1; f < NFW; ++f) behind_barrier(f, -1);
  } else {
    // before the loop: x of pair 0, the weights of pairs 0 and 1 (the sets of g = -2 and -1 as far as they exist)
#pragma unroll
    for (int q = 0; q < kXInstr; ++q) request(q, -1);
#pragma unroll
    for (int q = kXInstr; q < kReq; ++q) request(q, -2);
#pragma unroll
    for (int q = kXInstr; q < kReq; ++q) request(q, -1);
    landed();                       // pair 0
#pragma unroll
    for (int q = 0; q < NFW - 1; ++q) request(q, 0);
  }
  reads(0, 0);
  for (int g = 0; g < P; ++g) {
    // the scheduler may not move anything across a step's boundaries (left alone it puts every read right before its
    // use and waits for it)
    __builtin_amdgcn_sched_barrier(0);
    reads(g, 1);
#pragma unroll
    for (int f = 0; f < NFW; ++f) {
      if constexpr (!GPTQ) request(NFW - 1 + f, g);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(0, f);
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_sched_barrier(0);
    mfmas(1, 0);
    __builtin_amdgcn_sched_barrier(0);
    landed();                     // pair g + 1; every wave has issued its last read of pair g: its images are free
    reads(g + 1, 0);
#pragma unroll
    for (int f = 1; f < NFW; ++f) {
      behind_barrier(f, g);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(1, f);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  wait_loads_then_barrier<0>();      // the repeats of the last pair have landed; every wave is done with the images

  // ---- epilogue: 64 rows at a time (row fragment i of every wave) as a row-major 4-byte image in LDS
  float* img = reinterpret_cast<float*>(smem);
  const int vec_per_row = 4 * nf;
#pragma unroll
  for (int i = 0; i < MTW; ++i) {
    if (i > 0) __syncthreads();
#pragma unroll
    for (int f = 0; f < NFW; ++f)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v;
        if constexpr (I8) v = __int_as_float(acc[i][f][r]); else v = acc[i][f][r];
        img[(16 * rg + 4 * cq + r) * kPitch + 16 * (wave_f0 + f) + n] = v;
      }
    __syncthreads();
    const int row0 = 64 * i;
    if (a.gated) {
      // image column 16 f + c of fragment f: c < 8 the gate of output column 8 f + c, c >= 8 the up factor of 8 f + c - 8
      for (int idx = tid; idx < 64 * 2 * nf; idx += kThreads) {
        const int ir = idx / (2 * nf), v = idx % (2 * nf);
        const int row = row0 + ir;
        if (row >= a.M) continue;
        const int f = v >> 1, within = 4 * (v & 1);
        const acc_t g4 = *reinterpret_cast<const acc_t*>(img + ir * kPitch + 16 * f + within);
        const acc_t u4 = *reinterpret_cast<const acc_t*>(img + ir * kPitch + 16 * f + 8 + within);
        const int ocol = 8 * (tile_f0 + f) + within;
        float gg[4], uu[4];
        if constexpr (I8) {
          const float xs_ = a.x_scale[row];
          const f32x4 wg = *reinterpret_cast<const f32x4*>(a.w_scale + ocol);
          const f32x4 wu = *reinterpret_cast<const f32x4*>(a.w_scale + up0 + ocol);
#pragma unroll
          for (int e = 0; e < 4; ++e) gg[e] = w8a8_scaled(g4[e], xs_, wg[e], 0.f), uu[e] = w8a8_scaled(u4[e], xs_, wu[e], 0.f);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) gg[e] = g4[e], uu[e] = u4[e];
        }
        // both factors rounded to bf16 first, as the two-kernel form (linear, then silu_and_mul) holds them
        const uint32_t gp[2] = {pack_bf16x2(gg[0], gg[1]), pack_bf16x2(gg[2], gg[3])};
        const uint32_t up[2] = {pack_bf16x2(uu[0], uu[1]), pack_bf16x2(uu[2], uu[3])};
        u32x2 o;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const float g0 = lo_bf(gp[e]), g1 = hi_bf(gp[e]);
          o[e] = pack_bf16x2(g0 * sigmoidf(g0) * lo_bf(up[e]), g1 * sigmoidf(g1) * hi_bf(up[e]));
        }
        *reinterpret_cast<u32x2*>(a.out + static_cast<int64_t>(row) * a.out_stride + ocol) = o;
      }
      continue;
    }
    for (int idx = tid; idx < 64 * vec_per_row; idx += kThreads) {
      const int ir = idx / vec_per_row, c4 = idx % vec_per_row;
      const int row = row0 + ir;
      if (row >= a.M) continue;
      const int col = 16 * tile_f0 + 4 * c4;
      const acc_t sum = *reinterpret_cast<const acc_t*>(img + ir * kPitch + 4 * c4);
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
        const f32x4 wsc = *reinterpret_cast<const f32x4*>(a.w_scale + col);
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = w8a8_scaled(sum[e], xs_, wsc[e], b[e]);
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

// A tile holds 8 to 10 fragments (the matrix's fragments dealt in equal shares); waves 0..3 own its first
// (nf + 1) / 2, waves 4..7 the rest.
template <int MTW, bool I8, int GQ = 0>
__global__ __launch_bounds__(kThreads) void tall_gemm_kernel(StreamArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = blockIdx.x / a.splits, split = blockIdx.x % a.splits;
  const int frags = a.N / 16;
  const int f_lo = static_cast<int>(static_cast<int64_t>(t) * frags / a.tiles);
  const int f_hi = static_cast<int>(static_cast<int64_t>(t + 1) * frags / a.tiles);
  const int nf = f_hi - f_lo;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int first = (nf + 1) / 2;
  const int mine = wave < 4 ? first : nf - first;
  if (mine == 5) tall_wave<MTW, 5, I8, GQ>(a, smem, f_lo, nf, wave < 4 ? 0 : first, split);
  else tall_wave<MTW, 4, I8, GQ>(a, smem, f_lo, nf, wave < 4 ? 0 : first, split);
}

template <int MTW, bool I8, int GQ = 0>
int launch_tall(const StreamArgs& a, hipStream_t s) {
  constexpr size_t img = static_cast<size_t>(64) * (16 * kMaxNf + 4) * 4;
  constexpr size_t stream = static_cast<size_t>(2) * 64 * MTW * 128 + static_cast<size_t>(GQ ? 2 : kWRing) * kMaxNf * 2048;
  constexpr size_t lds = img > stream ? img : stream;
  auto kern = tall_gemm_kernel<MTW, I8, GQ>;
  static bool configured = false;
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)) != hipSuccess)
      (void)hipGetLastError();
    configured = true;
  }
  hipLaunchKernelGGL(kern, dim3(a.tiles * a.splits), dim3(kThreads), lds, s, a);
  return launch_status();
}

}  // namespace

// HYMETA_TALL_MIN_ROWS: rows from which the tall kernel is planned (default 129: at 128 rows the decode-sized kernel
// is level for int8 and 5 % ahead for bf16 on the 7B step; above 256 turns this kernel off).
static int tall_min_rows() {
  static const int v = [] {
    const char* e = getenv("HYMETA_TALL_MIN_ROWS");
    const int r = e == nullptr ? 0 : atoi(e);
    return r > 0 ? r : 129;
  }();
  return v;
}

bool tall_gemm_plan(int M, int K_bytes, int N, bool have_workspace, StreamPlan* p, int min_rows) {
  if (M < (min_rows > 0 ? min_rows : tall_min_rows()) || M > 256 || K_bytes % 128 != 0 || K_bytes < 512 || N % 64 != 0) return false;
  const int frags = N / 16, pairs = K_bytes / 128;
  int tiles, splits = 1;
  if (frags >= 9 * kWant) {
    // wide matrix: a multiple of `want` tiles of at most 10 fragments (one workgroup runs per compute unit, the tiles
    // go by in rounds)
    const int rounds = (frags + 10 * kWant - 1) / (10 * kWant);
    tiles = rounds * kWant;
  } else {
    // 128-column tiles; K slices make up the count
    tiles = (frags + 7) / 8;
    if (have_workspace && tiles < kWant) splits = kWant / tiles;
    if (splits > pairs / 4) splits = pairs / 4;        // slices of at least four pairs
    if (splits > kMaxSplits) splits = kMaxSplits;
    if (splits < 1) splits = 1;
  }
  const int nf = (frags + tiles - 1) / tiles;
  if (nf < 8 || nf > 10) return false;
  if (tiles * splits < kWant / 2) return false;      // too few workgroups for the chip: a library GEMM is ahead
  p->nf = nf, p->tiles = tiles, p->chunks = (K_bytes + 511) / 512, p->splits = splits, p->ring = 0, p->tall = 1;
  return true;
}

template <bool I8>
int tall_gemm_launch(const StreamArgs& a, const StreamPlan& p, hipStream_t s) {
  if (a.M <= 0 || a.M > 256 || a.k_bytes % 128 != 0 || a.k_bytes / 128 < p.splits) return HYMETA_ERR_INVALID_ARG;
  if (p.nf < 8 || p.nf > kMaxNf) return HYMETA_ERR_INVALID_ARG;
  if (a.M <= 192) return launch_tall<3, I8>(a, s);
  return launch_tall<4, I8>(a, s);
}

int tall_gemm_gptq_launch(const StreamArgs& a0, const StreamPlan& p, hipStream_t s, bool defer_reduce) {
  StreamArgs a = a0;
  a.tiles = p.tiles, a.chunks = p.chunks, a.splits = p.splits;
  if (!p.tall || a.M <= 0 || a.M > 256 || a.k_bytes % 128 != 0 || a.k_bytes / 128 < p.splits || a.out_f32) return HYMETA_ERR_INVALID_ARG;
  if (a.group_size != 64 && a.group_size != 128) return HYMETA_ERR_UNSUPPORTED;
  if (p.nf < 8 || p.nf > kMaxNf || (a.gated && p.splits != 1)) return HYMETA_ERR_INVALID_ARG;
  int rc;
  if (a.bits == 4) rc = a.M <= 192 ? launch_tall<3, false, 4>(a, s) : launch_tall<4, false, 4>(a, s);
  else if (a.bits == 8) rc = a.M <= 128 ? launch_tall<2, false, 8>(a, s) : a.M <= 192 ? launch_tall<3, false, 8>(a, s) : launch_tall<4, false, 8>(a, s);
  else return HYMETA_ERR_UNSUPPORTED;
  if (rc != HYMETA_OK || a.splits == 1 || defer_reduce) return rc;
  return stream_gemm_reduce_launch(a, s);
}

template int tall_gemm_launch<false>(const StreamArgs&, const StreamPlan&, hipStream_t);
template int tall_gemm_launch<true>(const StreamArgs&, const StreamPlan&, hipStream_t);

}  // namespace hymeta
