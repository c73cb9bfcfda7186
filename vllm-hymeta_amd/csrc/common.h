// Shared device helpers for the gfx950 kernels of libhymeta_hip.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hymeta_hip.h"

namespace hymeta {

// Guard of every entry point: the caller's struct is there and was compiled to the size this library
// was built with (its first field says so), checked before any other field is read.
template <typename T>
inline bool args_ok(const T* a) { return a != nullptr && a->struct_size == sizeof(T); }

// An argument struct the library builds for its own calls.
template <typename T>
__host__ __device__ inline T make_args() {
  T t = {};
  t.struct_size = static_cast<uint32_t>(sizeof(T));
  return t;
}

constexpr int kWave = 64;        // CDNA wavefront width
constexpr int kHeadDim = 128;    // the only head size the Hymeta models use

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));  // MFMA A/B fragment (8 bf16)

__device__ __forceinline__ float bf2f(uint16_t x) {
  return __uint_as_float(static_cast<uint32_t>(x) << 16);
}

// Round-to-nearest-even fp32 -> bf16 (plain cast: v_cvt_pk_bf16_f32, NaN-safe).
__device__ __forceinline__ uint16_t f2bf(float f) {
  __bf16 b = static_cast<__bf16>(f);
  return __builtin_bit_cast(uint16_t, b);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// Two fp32 -> packed bf16 pair, round-to-nearest-even, one v_cvt_pk_bf16_f32.
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

__device__ __forceinline__ float bf16_round(float f) { return bf2f(f2bf(f)); }

// Low / high bf16 of a packed dword as fp32.
__device__ __forceinline__ float lo_bf(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float hi_bf(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

__device__ __forceinline__ u32x4 ld16(const void* p) {
  return *reinterpret_cast<const u32x4*>(p);
}
// 16-byte load through an explicitly GLOBAL pointer.  A pointer chosen at run time between two
// kernel-argument buffers degrades to a flat pointer; flat loads count on both the vector-memory
// and the LDS counters, so the compiler waits for all of them at the first use and a register
// pipeline of loads drains.
__device__ __forceinline__ u32x4 ld16g(const void* p) {
  return *(const __attribute__((address_space(1))) u32x4*)(p);
}
// The same with the non-temporal hint, for data that is read once per launch (the KV cache rows of
// a decode step): at 256 sequences the paged decode streams 6.7 TB/s with it against 6.0 TB/s without.
__device__ __forceinline__ u32x4 ld16nt(const void* p) {
  return __builtin_nontemporal_load((const __attribute__((address_space(1))) u32x4*)(p));
}
__device__ __forceinline__ void st16(void* p, u32x4 v) { *reinterpret_cast<u32x4*>(p) = v; }

__device__ __forceinline__ float logsigmoidf(float x) {
  // log(sigmoid(x)) = min(x, 0) - log1p(exp(-|x|))
  return fminf(x, 0.f) - log1pf(__expf(-fabsf(x)));
}
__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + __expf(-x)); }

// Ticket counters of the split-K GEMMs: the first kTicketBytes of the caller's workspace.  They must
// be zero when a workspace is first used (one hipMemset at allocation); every call leaves them zero
// (the workgroup that draws the last ticket of a tile puts the counter back), so no call spends a
// memset node on them (measured: 5 us per node inside a replayed graph, 0.6 ms of a 5.7 ms decode step).
constexpr size_t kTicketBytes = HYMETA_GEMM_TICKET_BYTES;
constexpr int kMaxTickets = static_cast<int>(kTicketBytes / sizeof(int));

// Split-K meeting point of the skinny GEMMs: every K-slice workgroup of an output tile stores its
// fp32 partial tile ("slab") with plain stores, publishes it with one agent-scope release and
// draws a ticket; the workgroup that draws the last ticket acquires, puts the counter back to zero
// and returns true: it sums the slabs and writes the tile.  `flag` is one int of the caller's
// (already idle) LDS array.
__device__ __forceinline__ bool last_arriver(int* counter, int splits, int* flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int ticket = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = ticket == splits - 1;
    if (last) {
      __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // every other slice has drawn already
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    *flag = last ? 1 : 0;
  }
  __syncthreads();
  return *flag != 0;
}

// Entry points report launch failures through hipGetLastError(), which is sticky per host thread:
// a benign error left behind by an earlier runtime call of the host program (a not-ready event
// or stream query, for instance) must not be mistaken for ours, so the state is cleared first.
inline hipStream_t begin_launches(hymeta_stream_t stream) {
  (void)hipGetLastError();
  return static_cast<hipStream_t>(stream);
}

inline int launch_status() {
  return hipGetLastError() == hipSuccess ? HYMETA_OK : HYMETA_ERR_LAUNCH;
}

// The W8A8 output scaling, written once so that every kernel that applies it rounds the same way whatever the
// compiler would contract: (acc * x_scale) rounded, then one fused multiply-add with the weight scale and the bias
// (bias 0 without one: the fused form then equals the rounded product).
__device__ __forceinline__ float w8a8_scaled(int acc, float x_scale, float w_scale, float bias) {
  return __builtin_fmaf(static_cast<float>(acc) * x_scale, w_scale, bias);
}

}  // namespace hymeta
