// Read-bandwidth calibration for the weight stream of a skinny GEMM: a [rows][pitch] matrix read by
// workgroups of 4 waves, each wave streaming 64 rows along K, 16 bytes per lane per instruction, in the
// lane -> (row, 16-byte chunk) mappings a kernel could use.  build: hipcc -O3 --offload-arch=gfx950 membw2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define LDNT(p) __builtin_nontemporal_load((const __attribute__((address_space(1))) u32x4*)(p))
#define LDPL(p) (*(const __attribute__((address_space(1))) u32x4*)(p))

// MODE 0: MFMA fragment shape: lane = (row l%16, chunk l/16); one instruction = 16 rows x 64 B, 4 row groups per step
// MODE 1: same bytes per instruction, adjacent lanes contiguous: lane = (row l/4, chunk l%4)
// MODE 2: full lines: lane = (row l/8, chunk l%8): one instruction = 8 rows x 128 B
// MODE 3: MODE 0 with the two 64-byte halves of a line issued back to back
template <int MODE, bool NT, int DEPTH>
__global__ __launch_bounds__(256) void stream(const char* __restrict__ w, int64_t pitch, int kbytes, int ksplit, uint32_t* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = blockIdx.x / ksplit, ks = blockIdx.x % ksplit;
  const int kper = kbytes / ksplit;
  const char* base = w + (int64_t)(tile * 256 + wave * 64) * pitch + (int64_t)ks * kper;
  u32x4 acc = {0, 0, 0, 0};
  // one "step" = 128 bytes of each of the wave's 64 rows = 8 instructions
  const int nsteps = kper / 128;
  const char* p[8];
  for (int j = 0; j < 8; ++j) {
    int row, off;
    if (MODE == 0) { row = 16 * (j >> 1) + (lane & 15); off = 64 * (j & 1) + 16 * (lane >> 4); }
    else if (MODE == 3) { row = 16 * (j >> 1) + (lane & 15); off = 64 * (j & 1) + 16 * (lane >> 4); }
    else if (MODE == 1) { row = 16 * (j >> 1) + (lane >> 2); off = 64 * (j & 1) + 16 * (lane & 3); }
    else { row = 8 * j + (lane >> 3); off = 16 * (lane & 7); }
    p[j] = base + (int64_t)row * pitch + off;
  }
  // MODE 0 issues the first halves of all 4 row groups, then the second halves; MODE 3 / 1 / 2 in index order
  for (int s = 0; s < nsteps; s += DEPTH) {
    u32x4 v[DEPTH][8];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int off = 128 * min(s + d, nsteps - 1);
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const int j = MODE == 0 ? ((jj & 3) * 2 + (jj >> 2)) : jj;
        v[d][jj] = NT ? LDNT(p[j] + off) : LDPL(p[j] + off);
      }
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) acc ^= v[d][jj];
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) *sink = 1;
}

template <int MODE, bool NT, int DEPTH>
void run(const char* name, std::vector<char*>& bufs, int rows, int64_t pitch, int kbytes, int ksplit, uint32_t* sink) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = rows / 256 * ksplit;
  for (auto b : bufs) hipLaunchKernelGGL((stream<MODE, NT, DEPTH>), dim3(grid), dim3(256), 0, 0, b, pitch, kbytes, ksplit, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int it = 0; it < 3; ++it)
    for (auto b : bufs) hipLaunchKernelGGL((stream<MODE, NT, DEPTH>), dim3(grid), dim3(256), 0, 0, b, pitch, kbytes, ksplit, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / (3 * bufs.size());
  printf("%-44s grid %4d depth %d %s: %7.1f us  %5.2f TB/s\n", name, grid, DEPTH, NT ? "nt" : "  ", us, (double)rows * kbytes / us / 1e6);
}

int main(int argc, char** argv) {
  const int rows = 37888, kbytes = 7168;      // the 7B gate_up projection in bf16
  const int64_t pitch = kbytes;
  std::vector<char*> bufs(5);
  for (auto& b : bufs) { hipMalloc(&b, (size_t)rows * pitch); hipMemset(b, 1, (size_t)rows * pitch); }
  uint32_t* sink; hipMalloc(&sink, 4);
  for (int ksplit : {2, 4}) {
    run<0, true, 2>("fragment shape (16 rows x 64 B)", bufs, rows, pitch, kbytes, ksplit, sink);
    run<0, true, 4>("fragment shape (16 rows x 64 B)", bufs, rows, pitch, kbytes, ksplit, sink);
    run<3, true, 2>("fragment shape, line halves back to back", bufs, rows, pitch, kbytes, ksplit, sink);
    run<3, true, 4>("fragment shape, line halves back to back", bufs, rows, pitch, kbytes, ksplit, sink);
    run<3, false, 4>("fragment shape, line halves back to back", bufs, rows, pitch, kbytes, ksplit, sink);
    run<1, true, 2>("16 rows x 64 B, adjacent lanes contiguous", bufs, rows, pitch, kbytes, ksplit, sink);
    run<1, true, 4>("16 rows x 64 B, adjacent lanes contiguous", bufs, rows, pitch, kbytes, ksplit, sink);
    run<2, true, 2>("full lines (8 rows x 128 B)", bufs, rows, pitch, kbytes, ksplit, sink);
    run<2, true, 4>("full lines (8 rows x 128 B)", bufs, rows, pitch, kbytes, ksplit, sink);
    run<2, false, 4>("full lines (8 rows x 128 B)", bufs, rows, pitch, kbytes, ksplit, sink);
  }
  return 0;
}
