// Host-side interface of the weight-streaming GEMM (stream_gemm.hip), shared by the bf16 and W8A8 entry points.
#pragma once

#include "common.h"

namespace hymeta {

struct StreamArgs {
  const char* x;          // [M, K] row-major, x_stride_bytes per row (bf16, or int8 for W8A8)
  const char* w;          // [N, K] row-major, w_stride_bytes per row
  const uint16_t* bias;   // bf16 [N] or null
  uint16_t* out;          // bf16 [M, N]
  const float* x_scale;   // W8A8: fp32 [M]
  const float* w_scale;   // W8A8: fp32 [N]
  char* slabs;            // [splits][M][N] 4-byte partial sums (splits > 1)
  int M, N;
  int k_bytes;            // bytes of one row of x / W that hold K (a multiple of 128)
  int64_t x_stride_bytes, w_stride_bytes, out_stride;
  int out_f32;            // 1: out is fp32 [M, N] (out_stride in fp32 elements), the sums are stored unrounded; no K slices
  int gated;              // 1: W = [gate rows | up rows] (N / 2 each), out [M, N / 2] = bf16(silu(bf16(x gate^T)) * bf16(x up^T));
                          // no bias, no K slices
  int tiles, chunks, splits;    // filled from the plan
  // GPTQ weights (bits 4 or 8) for the 129..256-row kernel (tall_gemm_gptq_launch): w = qweight int32 [K * bits / 32, N] (w_stride_bytes
  // unused), k_bytes = 2 K (the bytes of a bf16 row of x)
  const void* qzeros;     // int32 [K / group_size, N * bits / 32]
  const void* scales;     // fp16 | bf16 [K / group_size, N]
  int scales_f16, zero_offset, group_size, bits;
};

struct StreamPlan {
  int nf;        // 16-column fragments of the widest wave (waves own nf or nf - 1: the matrix's fragments are dealt
                 // to the tiles, and a tile's to its four waves, in equal shares)
  int tiles, chunks;   // chunks: 512-byte pieces of a weight row
  int splits;          // K slices per tile (1: the workgroup writes the tile itself)
  int ring;      // pairs of k-steps each wave keeps in flight (2 or 4)
  int tall;      // 1: the plan of the 129..256-row kernel (tall_gemm.hip): nf = fragments of the widest TILE (8..10)
};

// Whether a streaming kernel takes the shape (M <= 128 rows: the decode-sized kernel; 129..256 rows: the tall one; K a multiple of 128 bytes and at least 512, N of 64 columns), and how
// it is cut (one workgroup per compute unit where the shape allows).  Without a workspace every workgroup owns whole tiles (no slabs).
bool stream_gemm_plan(int M, int K_bytes, int N, bool have_workspace, StreamPlan* plan);
// Gated form (see StreamArgs::gated): N counts the gate and the up rows together and must be a multiple of 128.
bool stream_gemm_plan_gated(int M, int K_bytes, int N, StreamPlan* plan);
// Slab bytes of that plan (0 without K slices); no ticket counters are used.
size_t stream_gemm_workspace_bytes(int M, int N, const StreamPlan& plan);
// A bound on stream_gemm_workspace_bytes over every plan the library may choose for (M, N).
size_t stream_gemm_workspace_bound(int M, int N);
// defer_reduce: with K slices, stop after the slabs are written (the caller hands them to a consumer that sums them).
template <bool I8>
int stream_gemm_launch(StreamArgs args, const StreamPlan& plan, hipStream_t stream, bool defer_reduce = false);

// The 129..256-row kernel (tall_gemm.hip); stream_gemm_plan / stream_gemm_launch route to these two.
bool tall_gemm_plan(int M, int K_bytes, int N, bool have_workspace, StreamPlan* plan, int min_rows = 0);      // min_rows 0: 129
template <bool I8>
int tall_gemm_launch(const StreamArgs& args, const StreamPlan& plan, hipStream_t stream);

// The fp32 slabs of a sliced launch summed in slice order (+ bias) -> bf16 out (args as the launch had them).
int stream_gemm_reduce_launch(const StreamArgs& args, hipStream_t stream);
// The same kernel on 4- or 8-bit GPTQ weights (group size 64 or 128): a pair's packed words are expanded into the weight image
// as bf16(scale * (q - zero)), the values hymeta_gptq_dequant writes; plan with tall_gemm_plan(M, 2 K, N, ...).
int tall_gemm_gptq_launch(const StreamArgs& args, const StreamPlan& plan, hipStream_t stream, bool defer_reduce);

// ---- the GPTQ form (stream_gptq.hip): 4-bit with group size 128 or 64 (K a multiple of 256) or 8-bit with group size 64
// (K a multiple of 128); N a multiple of 128 ----
struct StreamGptqArgs {
  const char* x;            // bf16 [M, K], x_stride_bytes per row
  const void* qweight;      // int32 [K bits / 32, N]
  const void* qzeros;       // int32 [K / group, N bits / 32]
  const void* scales;       // fp16 | bf16 [K / group, N]
  const uint16_t* bias;     // bf16 [N] or null
  uint16_t* out;            // bf16 [M, N]
  char* slabs;              // [splits][M][N] fp32 partial sums (splits > 1)
  int M, N;
  int scales_f16, zero_offset;
  int64_t x_stride_bytes, out_stride;
  int tiles, chunks, splits;    // filled from the plan
};

struct StreamGptqPlan {
  int np;               // 32-column blocks of the widest wave (1 or 2)
  int tiles, chunks;    // chunks: pieces of K of 32 packed rows (256 k for 4-bit, 128 k for 8-bit weights)
  int splits;
  int bits;             // 4 (group size 128 or 64) or 8 (group size 64)
  int group_units;      // units of 8 packed rows in a group: 2, or 1 for 4-bit weights of group size 64
};

bool stream_gptq_plan(int M, int K, int N, int bits, int group_size, bool have_workspace, StreamGptqPlan* plan);
size_t stream_gptq_workspace_bytes(int M, int N, const StreamGptqPlan& plan);
size_t stream_gptq_workspace_bound(int M, int N);
int stream_gptq_launch(StreamGptqArgs args, const StreamGptqPlan& plan, hipStream_t stream, bool defer_reduce = false);

}  // namespace hymeta
