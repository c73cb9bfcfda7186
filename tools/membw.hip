// Read-bandwidth calibration: streams a buffer with 16-byte loads in a few access shapes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#ifdef NONTEMPORAL   // build with -DNONTEMPORAL: every load carries the non-temporal hint
#define LD(ptr) __builtin_nontemporal_load(ptr)
#else
#define LD(ptr) (*(ptr))
#endif

// plain streaming: grid-stride, 16 B per lane, UNROLL loads in flight per lane
template <int UNROLL>
__global__ __launch_bounds__(256) void stream_read(const u32x4* __restrict__ p, size_t n16, uint32_t* sink) {
  size_t i = (size_t)blockIdx.x * blockDim.x * UNROLL + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x * UNROLL;
  u32x4 acc = {0, 0, 0, 0};
  for (; i + (UNROLL - 1) * blockDim.x < n16; i += stride) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = LD(&p[i + u * blockDim.x]);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc ^= v[u];
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) *sink = 1;
}

// strided rows: each workgroup reads 256-byte rows at a 1024-byte pitch (one kv head of a [tok][4][128] bf16 cache)
template <int UNROLL>
__global__ __launch_bounds__(256) void strided_rows(const char* __restrict__ base, size_t rows_per_wg, uint32_t* sink) {
  // wg -> (chunk, head): 4 heads interleaved
  const int head = blockIdx.x & 3;
  const size_t chunk = blockIdx.x >> 2;
  const char* p = base + chunk * rows_per_wg * 1024 + head * 256;
  const int lane16 = threadIdx.x & 15, row_in = threadIdx.x >> 4;  // 16 rows per pass
  u32x4 acc = {0, 0, 0, 0};
  for (size_t r = 0; r + 16 * UNROLL <= rows_per_wg; r += 16 * UNROLL) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
      v[u] = LD(reinterpret_cast<const u32x4*>(p + (r + u * 16 + row_in) * 1024 + lane16 * 16));
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc ^= v[u];
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) *sink = 1;
}

// attention-tile shapes: a wave reads 32 rows (256 B each, 1024-B pitch) per tile.
// MODE 0: "V shape": lane = (row group, 16-B chunk): one instruction covers 4 whole rows.
// MODE 1: "K shape": lane = (row, 64-B quarter): one instruction covers 16 rows x 64 B.
// MODE 2: K tile in K shape + V tile in V shape from two different buffers (the decode kernel's mix).
// MODE 3: the mix of MODE 2 with the 4 heads of a row range on workgroups 8 apart (the same XCD when
// consecutive workgroups go round the 8 XCDs) instead of on 4 consecutive workgroups.
template <int MODE>
__global__ __launch_bounds__(256) void tile_rows(const char* __restrict__ base, const char* __restrict__ base2,
                                                 size_t rows_per_wg, uint32_t* sink) {
  const int head = MODE == 3 ? (blockIdx.x >> 3) & 3 : blockIdx.x & 3;
  const size_t chunk = MODE == 3 ? (blockIdx.x & 7) | ((blockIdx.x >> 5) << 3) : blockIdx.x >> 2;
  const char* p = base + chunk * rows_per_wg * 1024 + head * 256;
  const char* p2 = base2 + chunk * rows_per_wg * 1024 + head * 256;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane >> 4, n = lane & 15;
  u32x4 acc = {0, 0, 0, 0};
  for (size_t r = wave * 32; r + 32 <= rows_per_wg; r += 4 * 32) {
    u32x4 v[16];
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = LD(reinterpret_cast<const u32x4*>(p + (r + 4 * j + c) * 1024 + n * 16));
#pragma unroll
      for (int j = 0; j < 8; ++j) v[8 + j] = LD(reinterpret_cast<const u32x4*>(p2 + (r + 4 * j + c) * 1024 + n * 16));
    } else if (MODE == 1) {
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) v[4 * s + ks] = LD(reinterpret_cast<const u32x4*>(p + (r + 16 * s + n) * 1024 + ks * 64 + c * 16));
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) v[8 + 4 * s + ks] = LD(reinterpret_cast<const u32x4*>(p2 + (r + 16 * s + n) * 1024 + ks * 64 + c * 16));
    } else {
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) v[4 * s + ks] = LD(reinterpret_cast<const u32x4*>(p + (r + 16 * s + n) * 1024 + ks * 64 + c * 16));
#pragma unroll
      for (int j = 0; j < 8; ++j) v[8 + j] = LD(reinterpret_cast<const u32x4*>(p2 + (r + 4 * j + c) * 1024 + n * 16));
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) acc ^= v[u];
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) *sink = 1;
}

// The K+V mix with one 512-thread workgroup covering all 4 heads of a row range: wave = (head, tile parity),
// so the workgroup as a whole reads contiguous 1-KiB rows.
__global__ __launch_bounds__(512) void tile_rows_allheads(const char* __restrict__ base, const char* __restrict__ base2,
                                                          size_t rows_per_wg, uint32_t* sink) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane >> 4, n = lane & 15;
  const int head = wave & 3, par = wave >> 2;
  const char* p = base + (size_t)blockIdx.x * rows_per_wg * 1024 + head * 256;
  const char* p2 = base2 + (size_t)blockIdx.x * rows_per_wg * 1024 + head * 256;
  u32x4 acc = {0, 0, 0, 0};
  for (size_t r = par * 32; r + 32 <= rows_per_wg; r += 2 * 32) {
    u32x4 v[16];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) v[4 * s + ks] = LD(reinterpret_cast<const u32x4*>(p + (r + 16 * s + n) * 1024 + ks * 64 + c * 16));
#pragma unroll
    for (int j = 0; j < 8; ++j) v[8 + j] = LD(reinterpret_cast<const u32x4*>(p2 + (r + 4 * j + c) * 1024 + n * 16));
#pragma unroll
    for (int u = 0; u < 16; ++u) acc ^= v[u];
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) *sink = 1;
}

template <typename F>
float time_us(F f, int iters) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) f();
  hipEventRecord(a);
  for (int i = 0; i < iters; ++i) f();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms * 1e3f / iters;
}

int main() {
  const size_t bytes = (size_t)4 << 30;  // 4 GiB: far beyond the 256 MiB Infinity Cache
  char* buf; uint32_t* sink;
  hipMalloc(&buf, bytes); hipMalloc(&sink, 4);
  hipMemset(buf, 1, bytes);
  const size_t n16 = bytes / 16;
  for (int grid : {256, 512, 1024, 2048, 4096, 8192}) {
    float us = time_us([&] { hipLaunchKernelGGL(stream_read<8>, dim3(grid), dim3(256), 0, 0, (const u32x4*)buf, n16, sink); }, 5);
    printf("stream_read<8>  grid=%5d  %8.1f us  %7.1f GB/s\n", grid, us, bytes / us / 1e3);
  }
  for (int grid : {1024, 4096}) {
    float us = time_us([&] { hipLaunchKernelGGL(stream_read<4>, dim3(grid), dim3(256), 0, 0, (const u32x4*)buf, n16, sink); }, 5);
    printf("stream_read<4>  grid=%5d  %8.1f us  %7.1f GB/s\n", grid, us, bytes / us / 1e3);
  }
  // strided rows: total rows = bytes/1024; WGs = 4 heads x chunks
  for (size_t rows_per_wg : {4096, 1024, 512}) {
    const size_t chunks = bytes / 1024 / rows_per_wg;
    float us = time_us([&] { hipLaunchKernelGGL(strided_rows<8>, dim3(chunks * 4), dim3(256), 0, 0, buf, rows_per_wg, sink); }, 5);
    printf("strided_rows<8> rows/wg=%5zu wgs=%6zu %8.1f us  %7.1f GB/s\n", rows_per_wg, chunks * 4, us, bytes / us / 1e3);
  }
  {
    const size_t half = bytes / 2;
    const size_t rows_per_wg = 4096;
    const size_t chunks = half / 1024 / rows_per_wg;
    float us0 = time_us([&] { hipLaunchKernelGGL(tile_rows<0>, dim3(chunks * 4), dim3(256), 0, 0, buf, buf + half, rows_per_wg, sink); }, 5);
    float us1 = time_us([&] { hipLaunchKernelGGL(tile_rows<1>, dim3(chunks * 4), dim3(256), 0, 0, buf, buf + half, rows_per_wg, sink); }, 5);
    float us2 = time_us([&] { hipLaunchKernelGGL(tile_rows<2>, dim3(chunks * 4), dim3(256), 0, 0, buf, buf + half, rows_per_wg, sink); }, 5);
    printf("tile_rows V-shape  wgs=%zu %8.1f us %7.1f GB/s\n", chunks * 4, us0, bytes / us0 / 1e3);
    printf("tile_rows K-shape  wgs=%zu %8.1f us %7.1f GB/s\n", chunks * 4, us1, bytes / us1 / 1e3);
    printf("tile_rows K+V mix  wgs=%zu %8.1f us %7.1f GB/s\n", chunks * 4, us2, bytes / us2 / 1e3);
    float us5 = time_us([&] { hipLaunchKernelGGL(tile_rows<3>, dim3(chunks * 4), dim3(256), 0, 0, buf, buf + half, rows_per_wg, sink); }, 5);
    printf("tile_rows K+V mix, heads of a range on one XCD  wgs=%zu %8.1f us %7.1f GB/s\n", chunks * 4, us5, bytes / us5 / 1e3);
    for (size_t rows : {2048, 512, 256}) {     // 32, 8 and 4 tiles per wave
      const size_t wgs = half / 1024 / rows;
      float us3 = time_us([&] { hipLaunchKernelGGL(tile_rows_allheads, dim3(wgs), dim3(512), 0, 0, buf, buf + half, rows, sink); }, 5);
      printf("K+V mix, all heads per wg, rows/wg=%zu wgs=%zu %8.1f us %7.1f GB/s\n", rows, wgs, us3, bytes / us3 / 1e3);
      const size_t chunks2 = half / 1024 / (rows * 2);   // same tiles per wave for the per-head form
      float us4 = time_us([&] { hipLaunchKernelGGL(tile_rows<2>, dim3(chunks2 * 4), dim3(256), 0, 0, buf, buf + half, rows * 2, sink); }, 5);
      printf("K+V mix, one head per wg,  rows/wg=%zu wgs=%zu %8.1f us %7.1f GB/s\n", rows * 2, chunks2 * 4, us4, bytes / us4 / 1e3);
    }
  }
  return 0;
}
