// Paged KV write: scatter each token's K and V rows into the block it maps to.
// A byte copy (bit-exact), 16 bytes per lane; consecutive lanes cover one
// token's (kv-heads x head_dim) row, so both the read and the write side are
// contiguous runs of num_kv_heads * head_dim * elem_bytes bytes.
#include "common.h"

namespace hymeta {

__global__ __launch_bounds__(256) void reshape_and_cache_kernel(hymeta_reshape_and_cache_args a,
                                                                int chunks_per_row, int elems_per_chunk) {
  const int64_t total = static_cast<int64_t>(a.num_tokens) * 2 * chunks_per_row;
  const int64_t slots_total = a.num_blocks * a.block_size;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t t = i / (2 * chunks_per_row);
    const int rem = static_cast<int>(i % (2 * chunks_per_row));
    const int which = rem / chunks_per_row;  // 0 = key, 1 = value
    const int chunk = rem % chunks_per_row;
    const int64_t slot = a.slot_mapping[t];
    if (slot < 0 || slot >= slots_total) continue;  // PAD_SLOT_ID
    const int e = chunk * elems_per_chunk;
    const int head = e / a.head_dim;
    const int d = e % a.head_dim;
    const int64_t blk = slot / a.block_size;
    const int64_t pos = slot % a.block_size;
    const int64_t dst_off = blk * a.cache_stride_block + pos * a.cache_stride_pos + head * a.cache_stride_h + d;
    if (which == 0) {
      const char* src = static_cast<const char*>(a.key) + (t * a.key_stride_t + head * a.key_stride_h + d) * a.elem_bytes;
      st16(static_cast<char*>(a.key_cache) + dst_off * a.elem_bytes, ld16(src));
    } else {
      const char* src = static_cast<const char*>(a.value) + (t * a.value_stride_t + head * a.value_stride_h + d) * a.elem_bytes;
      st16(static_cast<char*>(a.value_cache) + dst_off * a.elem_bytes, ld16(src));
    }
  }
}

}  // namespace hymeta

using namespace hymeta;

extern "C" int hymeta_reshape_and_cache(const hymeta_reshape_and_cache_args* args, hymeta_stream_t stream) {
  if (!hymeta::args_ok(args)) return HYMETA_ERR_INVALID_ARG;
  const hymeta_reshape_and_cache_args& a = *args;
  if (a.num_tokens < 0 || a.num_kv_heads <= 0 || a.head_dim <= 0 || a.block_size <= 0 || a.num_blocks < 0)
    return HYMETA_ERR_INVALID_ARG;
  if (a.num_tokens == 0) return HYMETA_OK;
  if (!a.key || !a.value || !a.key_cache || !a.value_cache || !a.slot_mapping) return HYMETA_ERR_INVALID_ARG;
  if (a.elem_bytes != 1 && a.elem_bytes != 2 && a.elem_bytes != 4) return HYMETA_ERR_UNSUPPORTED;
  const int elems_per_chunk = 16 / a.elem_bytes;
  if (a.head_dim % elems_per_chunk != 0) return HYMETA_ERR_UNSUPPORTED;
  const int64_t strides[] = {a.key_stride_t, a.key_stride_h, a.value_stride_t, a.value_stride_h,
                             a.cache_stride_block, a.cache_stride_pos, a.cache_stride_h};
  for (int64_t st : strides)
    if (st % elems_per_chunk != 0) return HYMETA_ERR_UNSUPPORTED;
  const uintptr_t ptrs = reinterpret_cast<uintptr_t>(a.key) | reinterpret_cast<uintptr_t>(a.value) |
                         reinterpret_cast<uintptr_t>(a.key_cache) | reinterpret_cast<uintptr_t>(a.value_cache);
  if (ptrs % 16 != 0) return HYMETA_ERR_UNSUPPORTED;

  const int chunks_per_row = a.num_kv_heads * a.head_dim / elems_per_chunk;
  const int64_t total = static_cast<int64_t>(a.num_tokens) * 2 * chunks_per_row;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(reshape_and_cache_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0,
                     begin_launches(stream), a, chunks_per_row, elems_per_chunk);
  return launch_status();
}
