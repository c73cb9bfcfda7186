// Read-bandwidth calibration, continued: does a weight stream run faster when one load instruction covers fewer rows and
// longer runs of each?  A [rows][pitch] matrix, workgroups of 4 waves, each wave streaming 64 rows along K with 16
// loads of 16 bytes per lane in flight; one instruction = ROWS rows x (1024 / ROWS) bytes.
// build: hipcc -O3 --offload-arch=gfx950 membw3.hip -o membw3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define LDNT(p) __builtin_nontemporal_load((const __attribute__((address_space(1))) u32x4*)(p))
#define LDPL(p) (*(const __attribute__((address_space(1))) u32x4*)(p))

template <int ROWS, bool NT>
__global__ __launch_bounds__(256) void stream(const char* __restrict__ w, int64_t pitch, int kbytes, int ksplit, uint32_t* sink) {
  constexpr int RUN = 1024 / ROWS;          // bytes of one row per instruction
  constexpr int GROUPS = 64 / ROWS;         // instructions that cover the wave's 64 rows once
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = blockIdx.x / ksplit, ks = blockIdx.x % ksplit;
  const int kper = kbytes / ksplit;
  const char* base = w + (int64_t)(tile * 256 + wave * 64) * pitch + (int64_t)ks * kper +
                     (int64_t)(lane / (RUN / 16)) * pitch + 16 * (lane % (RUN / 16));
  u32x4 acc = {0, 0, 0, 0};
  const int total = kper / RUN * GROUPS;    // instructions of the wave: run index outer, row group inner
  for (int s = 0; s < total; s += 16) {
    u32x4 v[16];
#pragma unroll
    for (int d = 0; d < 16; ++d) {
      const int i = min(s + d, total - 1);
      const char* p = base + (int64_t)(i % GROUPS) * ROWS * pitch + (int64_t)(i / GROUPS) * RUN;
      v[d] = NT ? LDNT(p) : LDPL(p);
    }
#pragma unroll
    for (int d = 0; d < 16; ++d) acc ^= v[d];
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) *sink = 1;
}

template <int ROWS, bool NT>
void run(std::vector<char*>& bufs, int rows, int64_t pitch, int kbytes, int ksplit, uint32_t* sink) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = rows / 256 * ksplit;
  for (auto b : bufs) hipLaunchKernelGGL((stream<ROWS, NT>), dim3(grid), dim3(256), 0, 0, b, pitch, kbytes, ksplit, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int it = 0; it < 3; ++it)
    for (auto b : bufs) hipLaunchKernelGGL((stream<ROWS, NT>), dim3(grid), dim3(256), 0, 0, b, pitch, kbytes, ksplit, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / (3 * bufs.size());
  printf("%2d rows x %4d B per instruction, grid %4d %s: %7.1f us  %5.2f TB/s\n", ROWS, 1024 / ROWS, grid, NT ? "nt" : "  ", us,
         (double)rows * kbytes / us / 1e6);
}

int main() {
  const int rows = 37888, kbytes = 7168;      // the 7B gate_up projection in bf16
  const int64_t pitch = kbytes;
  std::vector<char*> bufs(5);
  for (auto& b : bufs) { hipMalloc(&b, (size_t)rows * pitch); hipMemset(b, 1, (size_t)rows * pitch); }
  uint32_t* sink; hipMalloc(&sink, 4);
  for (int ksplit : {1, 2, 7}) {
    run<8, true>(bufs, rows, pitch, kbytes, ksplit, sink);
    run<8, false>(bufs, rows, pitch, kbytes, ksplit, sink);
    run<4, true>(bufs, rows, pitch, kbytes, ksplit, sink);
    run<2, true>(bufs, rows, pitch, kbytes, ksplit, sink);
    run<2, false>(bufs, rows, pitch, kbytes, ksplit, sink);
    run<1, true>(bufs, rows, pitch, kbytes, ksplit, sink);
    run<1, false>(bufs, rows, pitch, kbytes, ksplit, sink);
  }
  return 0;
}
