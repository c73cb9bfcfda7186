// One-shot all-reduce for the small messages of a tensor-parallel decode step (the row-parallel
// out_proj / down_proj outputs, [<= 256, hidden] bf16, and the [T, 2] fp32 branch variances;
// reference: vllm_hymeta/models/modeling_hymeta.py:158-160,858-861, where vLLM's
// tensor_model_parallel_all_reduce carries them).  A ring moves such a message in 2 (W - 1) hops of
// a few microseconds each; here every rank writes its message straight into a slot of every peer's
// exchange buffer over xGMI (one hop), raises a flag there, waits for the W flags in its OWN buffer
// and sums the W slots in rank order.  Every rank adds the same values in the same order in fp32 and
// rounds once, so the results are bit-identical on all ranks.
//
//   exchange buffer of a rank (device memory allocated uncached, so that neither the writer's nor the
//   reader's L2 keeps lines of it; mapped into the peers through HIP IPC handles):
//     flags [2 parities][kMaxWorld][kMaxBlocks] u32     the epoch a (rank, block) has delivered
//     probe [kMaxWorld] u32 (+ padding)                 set-up only: every peer writes a token here with a runtime copy
//     slots [2 parities][W][slot_bytes]                 the messages, fp32 part first, then bf16
//   state of a rank (ordinary device memory, local): epoch, block ticket, timeouts
//
// The epoch lives in device memory and is advanced by the kernel itself (the last block of a launch),
// so a launch carries no per-call host value and replays from a HIP graph.  Two parities are enough:
// a rank can push epoch e + 2 only after its launch e + 1 has ended, which needed every peer's
// flag e + 1, which a peer raises only in a launch that follows the end of its launch e, the last
// reader of the parity's slots.  That argument needs the launches of one communicator to be ordered on
// every rank (one stream, or one graph): calls that overlap on a rank race.
//
// Every wait is bounded by wall-clock ticks, so that every launch ends whatever the peers do.  A block
// whose wait runs out counts a timeout in the state (a sticky count: nothing on the device ever clears
// it) and writes NaN over its share of the output instead of a sum of whatever the slots held, so a
// missed peer can never pass for a result; the host looks at the count where it synchronises anyway
// (hymeta_p2p_timeouts) and must then stop using the communicator: after a timeout the argument above
// no longer holds -- the rank that gave up can reach epoch e + 2 and overwrite slots and flags of
// parity e that a late peer has not read yet, so the peer times out in turn.
#include "common.h"
#include "partial_sums.h"

#include <cstring>
#include <mutex>
#include <new>
#include <vector>

namespace hymeta {
namespace {

constexpr int kMaxWorld = 8;
constexpr int kMaxBlocks = 64;
constexpr int kThreads = 256;
constexpr size_t kProbeOffset = 2ull * kMaxWorld * kMaxBlocks * sizeof(uint32_t);   // behind the 4 KB of flags
constexpr size_t kFlagBytes = kProbeOffset + 256;    // flags + one probe word per rank (set-up check of the mappings)
constexpr uint32_t kProbeToken = 0xA5C30000u;
constexpr uint64_t kDefaultSpinTicks = 200000000ull;   // 2 s of the 100 MHz wall clock

struct P2pArgs {
  char* peers[kMaxWorld];
  uint32_t* state;          // [0] epoch, [1] ticket, [2] timeouts
  uint16_t* data;           // bf16, n8 units of 8
  float* stats;             // fp32, n_stats values
  int64_t n_stats, units_stats, units;   // units of 16 bytes: the fp32 part first
  int64_t slot_bytes;
  uint64_t spin_ticks;
  int rank, units_per_block;
  float stats_scale;        // the fp32 sums are multiplied by it (1 / W turns the sum of the ranks' means into the mean)
  hymeta_gemm_partials partials;   // MODE 1 / 2: the message is a K-sliced linear's result, still in its partial sums
  uint16_t* out;                   // MODE 1 / 2: where the sum over the ranks goes (bf16 [M, N] contiguous)
};

__device__ __forceinline__ uint32_t* flag_ptr(char* buf, int parity, int rank, int block) {
  return reinterpret_cast<uint32_t*>(buf) + (static_cast<size_t>(parity) * kMaxWorld + rank) * kMaxBlocks + block;
}

// MODE 0: the message is data / stats.  MODE 1 (fp32 slabs) / 2 (int32 slabs, W8A8): the message is the bf16 result
// of a K-sliced linear that has not run its reduce launch; the push adds the slices while it reads them (the bits that
// launch would have written) and the sum over the ranks goes to a.out.
template <int W, int MODE>
__global__ __launch_bounds__(kThreads) void p2p_all_reduce_kernel(P2pArgs a) {
  __shared__ uint32_t s_epoch;
  __shared__ uint32_t s_timed_out;
  const int tid = threadIdx.x, b = blockIdx.x;
  if (tid == 0) {
    s_epoch = __hip_atomic_load(a.state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
    s_timed_out = 0;
  }
  __syncthreads();
  const uint32_t epoch = s_epoch;
  const int parity = epoch & 1;
  const int64_t u_lo = static_cast<int64_t>(b) * a.units_per_block;
  const int64_t u_hi = min(u_lo + a.units_per_block, a.units);
  const size_t slots = kFlagBytes + static_cast<size_t>(parity) * W * a.slot_bytes;

  // push this block's share of the message into slot [rank] of every rank's buffer (its own too)
  for (int64_t u = u_lo + tid; u < u_hi; u += kThreads) {
    u32x4 v;
    if constexpr (MODE != 0) {
      const int64_t t = 8 * u / a.partials.N;
      const int col = static_cast<int>(8 * u - t * a.partials.N);
      float x[8];
      partial_row8<MODE == 2>(a.partials, t, col, x);
      v = {pack_bf16x2(x[0], x[1]), pack_bf16x2(x[2], x[3]), pack_bf16x2(x[4], x[5]), pack_bf16x2(x[6], x[7])};
    } else if (u < a.units_stats) {
      float f[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) f[i] = 4 * u + i < a.n_stats ? a.stats[4 * u + i] : 0.f;
      v = {__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3])};
    } else {
      v = ld16g(a.data + 8 * (u - a.units_stats));
    }
#pragma unroll
    for (int q = 0; q < W; ++q) {
      const int peer = (a.rank + q) % W;     // start with the own buffer, spread the first remote writes over the links
      *reinterpret_cast<u32x4*>(a.peers[peer] + slots + static_cast<size_t>(a.rank) * a.slot_bytes + 16 * u) = v;
    }
  }
  __threadfence_system();
  __syncthreads();
  if (tid < W) {
    __hip_atomic_store(flag_ptr(a.peers[tid], parity, a.rank, b), epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    // wait for rank tid's share of this block in the own buffer
    uint32_t* flag = flag_ptr(a.peers[a.rank], parity, tid, b);
    const uint64_t t0 = wall_clock64();
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != epoch) {
      if (wall_clock64() - t0 > a.spin_ticks) {
        atomicAdd(a.state + 2, 1u);
        s_timed_out = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(4);
    }
  }
  __syncthreads();

  // sum the W slots in rank order; a block that missed a peer writes NaN (fp32 and bf16 quiet NaNs) instead
  const bool poisoned = s_timed_out != 0;
  const char* mine = a.peers[a.rank] + slots;
  for (int64_t u = u_lo + tid; u < u_hi; u += kThreads) {
    u32x4 v[W];
#pragma unroll
    for (int q = 0; q < W; ++q) v[q] = ld16g(mine + static_cast<size_t>(q) * a.slot_bytes + 16 * u);
    if (poisoned) {
      if (u < a.units_stats) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (4 * u + i < a.n_stats) a.stats[4 * u + i] = __uint_as_float(0x7FC00000u);
      } else {
        const u32x4 nan = {0x7FC07FC0u, 0x7FC07FC0u, 0x7FC07FC0u, 0x7FC07FC0u};
        *reinterpret_cast<u32x4*>((MODE != 0 ? a.out : a.data) + 8 * (u - a.units_stats)) = nan;
      }
      continue;
    }
    if (u < a.units_stats) {
      float f[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < W; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] += __uint_as_float(v[q][i]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (4 * u + i < a.n_stats) a.stats[4 * u + i] = f[i] * a.stats_scale;
    } else {
      float f[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < W; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          f[2 * i] += lo_bf(v[q][i]);
          f[2 * i + 1] += hi_bf(v[q][i]);
        }
      const u32x4 o = {pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7])};
      *reinterpret_cast<u32x4*>((MODE != 0 ? a.out : a.data) + 8 * (u - a.units_stats)) = o;
    }
  }

  // the last block of the launch advances the epoch and puts the ticket back
  __syncthreads();
  if (tid == 0) {
    const uint32_t t = atomicAdd(a.state + 1, 1u);
    if (t == gridDim.x - 1) {
      __hip_atomic_store(a.state + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.state, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

template <int MODE>
int launch_exchange(int world, P2pArgs& a, hymeta_stream_t stream) {
  // 4 units per thread and block at least; every rank derives the same grid from the same sizes
  const int64_t want = (a.units + 4 * kThreads - 1) / (4 * kThreads);
  const int blocks = static_cast<int>(want < 1 ? 1 : want > kMaxBlocks ? kMaxBlocks : want);
  a.units_per_block = static_cast<int>((a.units + blocks - 1) / blocks);
  hipStream_t s = begin_launches(stream);
  switch (world) {
    case 1: hipLaunchKernelGGL((p2p_all_reduce_kernel<1, MODE>), dim3(blocks), dim3(kThreads), 0, s, a); break;
    case 2: hipLaunchKernelGGL((p2p_all_reduce_kernel<2, MODE>), dim3(blocks), dim3(kThreads), 0, s, a); break;
    case 4: hipLaunchKernelGGL((p2p_all_reduce_kernel<4, MODE>), dim3(blocks), dim3(kThreads), 0, s, a); break;
    case 8: hipLaunchKernelGGL((p2p_all_reduce_kernel<8, MODE>), dim3(blocks), dim3(kThreads), 0, s, a); break;
    default: return HYMETA_ERR_UNSUPPORTED;
  }
  return launch_status();
}

}  // namespace
}  // namespace hymeta

struct hymeta_p2p_comm {
  int rank = 0, world = 0, device = 0;
  size_t max_bytes = 0, slot_bytes = 0, buffer_bytes = 0;
  size_t alloc_bytes = 0;               // size of the allocation behind the own buffer (a parked one may be larger than asked)
  char* peers[hymeta::kMaxWorld] = {};
  bool opened[hymeta::kMaxWorld] = {};
  bool by_pointer[hymeta::kMaxWorld] = {};   // peers[q] is a buffer of this process that this communicator holds
  bool connected = false;
  uint32_t* state = nullptr;
  uint64_t spin_ticks = hymeta::kDefaultSpinTicks;
};

namespace hymeta {
namespace {
// Exchange buffers are never handed back to the driver while the process lives: a communicator that goes away parks
// its buffer here and a later one on the same device that needs no more takes it over.  (After a series of create /
// destroy cycles that freed uncached buffers, a later ordinary allocation of the process was seen with stretches of a
// kernel's output missing.  The cause is NOT established: a late store of a peer into a buffer that had changed hands
// would look the same as a stale uncached mapping.  Keeping the buffers, and the holder counts below, rule out the
// first; a process makes a handful of communicators, so nothing is lost by keeping them.)
struct ParkedBuffer {
  void* ptr;
  size_t bytes;
  int device;
};
std::mutex g_parked_mutex;
std::vector<ParkedBuffer> g_parked;

// Communicators of THIS process that still hold a pointer into a buffer (its owner, and every in-process peer that was
// connected to it by pointer): a buffer is handed to a new communicator only once nobody holds it any more, so a kernel
// of a communicator that is still alive can never write into a buffer that has changed hands.  (Peers in other
// processes hold IPC mappings, which this count cannot see: their communicators are closed behind a barrier of the
// group, vllm_hymeta/parallel.py.)
struct Held {
  void* ptr;
  int holders;
};
std::vector<Held> g_held;

void hold(void* ptr) {
  std::lock_guard<std::mutex> lock(g_parked_mutex);
  for (Held& h : g_held)
    if (h.ptr == ptr) {
      ++h.holders;
      return;
    }
  g_held.push_back({ptr, 1});
}

void release(void* ptr) {
  std::lock_guard<std::mutex> lock(g_parked_mutex);
  for (size_t i = 0; i < g_held.size(); ++i)
    if (g_held[i].ptr == ptr) {
      if (--g_held[i].holders <= 0) g_held.erase(g_held.begin() + static_cast<std::ptrdiff_t>(i));
      return;
    }
}

bool held_locked(const void* ptr) {
  for (const Held& h : g_held)
    if (h.ptr == ptr) return true;
  return false;
}

// The smallest parked buffer of the device that is large enough and that no communicator of the process still holds.
void* take_parked(size_t bytes, int device, size_t* got_bytes) {
  std::lock_guard<std::mutex> lock(g_parked_mutex);
  size_t best = g_parked.size();
  for (size_t i = 0; i < g_parked.size(); ++i)
    if (g_parked[i].bytes >= bytes && g_parked[i].device == device && !held_locked(g_parked[i].ptr) &&
        (best == g_parked.size() || g_parked[i].bytes < g_parked[best].bytes))
      best = i;
  if (best == g_parked.size()) return nullptr;
  void* p = g_parked[best].ptr;
  *got_bytes = g_parked[best].bytes;
  g_parked.erase(g_parked.begin() + static_cast<std::ptrdiff_t>(best));
  return p;
}

void park(void* ptr, size_t bytes, int device) {
  std::lock_guard<std::mutex> lock(g_parked_mutex);
  g_parked.push_back({ptr, bytes, device});
}
}  // namespace
}  // namespace hymeta

extern "C" int hymeta_p2p_create(int rank, int world, size_t max_message_bytes, hymeta_p2p_comm** comm, void* ipc_handle) {
  using namespace hymeta;
  if (comm == nullptr || world < 1 || world > kMaxWorld || rank < 0 || rank >= world || max_message_bytes == 0)
    return HYMETA_ERR_INVALID_ARG;
  if (world != 1 && world != 2 && world != 4 && world != 8) return HYMETA_ERR_UNSUPPORTED;
  static_assert(sizeof(hipIpcMemHandle_t) == HYMETA_P2P_HANDLE_BYTES, "IPC handle size");
  hymeta_p2p_comm* c = new (std::nothrow) hymeta_p2p_comm();
  if (c == nullptr) return HYMETA_ERR_WORKSPACE;
  c->rank = rank;
  c->world = world;
  c->max_bytes = max_message_bytes;
  c->slot_bytes = (max_message_bytes + 255) / 256 * 256;
  c->buffer_bytes = kFlagBytes + 2 * static_cast<size_t>(world) * c->slot_bytes;
  void* buf = nullptr;
  if (hipGetDevice(&c->device) != hipSuccess) {
    delete c;
    return HYMETA_ERR_WORKSPACE;
  }
  c->alloc_bytes = c->buffer_bytes;
  buf = take_parked(c->buffer_bytes, c->device, &c->alloc_bytes);
  if (buf == nullptr && hipExtMallocWithFlags(&buf, c->buffer_bytes, hipDeviceMallocUncached) != hipSuccess) {
    (void)hipGetLastError();
    delete c;
    return HYMETA_ERR_WORKSPACE;
  }
  c->peers[rank] = static_cast<char*>(buf);
  if (hipMalloc(reinterpret_cast<void**>(&c->state), 4 * sizeof(uint32_t)) != hipSuccess ||
      hipMemset(c->state, 0, 4 * sizeof(uint32_t)) != hipSuccess || hipMemset(buf, 0, kFlagBytes) != hipSuccess ||
      hipDeviceSynchronize() != hipSuccess) {
    park(buf, c->alloc_bytes, c->device);
    if (c->state) (void)hipFree(c->state);
    delete c;
    return HYMETA_ERR_WORKSPACE;
  }
  if (ipc_handle != nullptr) {
    hipIpcMemHandle_t h;
    if (hipIpcGetMemHandle(&h, buf) != hipSuccess) {
      (void)hipGetLastError();
      park(buf, c->alloc_bytes, c->device);
      (void)hipFree(c->state);
      delete c;
      return HYMETA_ERR_UNSUPPORTED;
    }
    memcpy(ipc_handle, &h, sizeof(h));
  }
  hold(buf);
  *comm = c;
  return HYMETA_OK;
}

extern "C" void* hymeta_p2p_local_buffer(const hymeta_p2p_comm* comm) {
  return comm != nullptr ? comm->peers[comm->rank] : nullptr;
}

extern "C" int hymeta_p2p_connect(hymeta_p2p_comm* comm, const void* all_handles, const int32_t* peer_devices) {
  using namespace hymeta;
  if (comm == nullptr || all_handles == nullptr || comm->connected) return HYMETA_ERR_INVALID_ARG;
  // Every way in which a mapping can be unusable must show up HERE as an error code: a kernel that stores through a
  // bad peer pointer faults the GPU.  So: peer access must be reported possible, the handle must open, and a runtime
  // copy of a token into the peer's probe word must succeed (hymeta_p2p_verify then checks that every peer's token
  // has arrived in the own buffer).
  if (peer_devices != nullptr) {
    for (int q = 0; q < comm->world; ++q) {
      if (q == comm->rank || peer_devices[q] == comm->device) continue;
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, comm->device, peer_devices[q]) != hipSuccess || !can) {
        (void)hipGetLastError();
        return HYMETA_ERR_UNSUPPORTED;
      }
    }
  }
  for (int q = 0; q < comm->world; ++q) {
    if (q == comm->rank) continue;
    hipIpcMemHandle_t h;
    memcpy(&h, static_cast<const char*>(all_handles) + static_cast<size_t>(q) * sizeof(h), sizeof(h));
    void* p = nullptr;
    if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess || p == nullptr) {
      (void)hipGetLastError();
      for (int r = 0; r < q; ++r)
        if (comm->opened[r]) {
          (void)hipIpcCloseMemHandle(comm->peers[r]);
          comm->opened[r] = false;
          comm->peers[r] = nullptr;
        }
      return HYMETA_ERR_UNSUPPORTED;
    }
    comm->peers[q] = static_cast<char*>(p);
    comm->opened[q] = true;
  }
  for (int q = 0; q < comm->world; ++q) {
    if (q == comm->rank) continue;
    const uint32_t token = kProbeToken | static_cast<uint32_t>(comm->rank);
    if (hipMemcpy(comm->peers[q] + kProbeOffset + sizeof(uint32_t) * comm->rank, &token, sizeof(token), hipMemcpyHostToDevice) != hipSuccess) {
      (void)hipGetLastError();
      for (int r = 0; r < comm->world; ++r)
        if (comm->opened[r]) {
          (void)hipIpcCloseMemHandle(comm->peers[r]);
          comm->opened[r] = false;
          comm->peers[r] = nullptr;
        }
      return HYMETA_ERR_UNSUPPORTED;
    }
  }
  comm->connected = true;
  return HYMETA_OK;
}

extern "C" int hymeta_p2p_verify(hymeta_p2p_comm* comm) {
  using namespace hymeta;
  if (comm == nullptr || !comm->connected) return HYMETA_ERR_INVALID_ARG;
  uint32_t probe[kMaxWorld] = {};
  if (hipMemcpy(probe, comm->peers[comm->rank] + kProbeOffset, sizeof(probe), hipMemcpyDeviceToHost) != hipSuccess) {
    (void)hipGetLastError();
    return HYMETA_ERR_LAUNCH;
  }
  for (int q = 0; q < comm->world; ++q)
    if (q != comm->rank && comm->opened[q] && probe[q] != (kProbeToken | static_cast<uint32_t>(q))) return HYMETA_ERR_UNSUPPORTED;
  return HYMETA_OK;
}

extern "C" int hymeta_p2p_connect_pointers(hymeta_p2p_comm* comm, void* const* buffers) {
  if (comm == nullptr || buffers == nullptr || comm->connected) return HYMETA_ERR_INVALID_ARG;
  for (int q = 0; q < comm->world; ++q) {
    if (buffers[q] == nullptr) return HYMETA_ERR_INVALID_ARG;
    if (q == comm->rank) {
      if (buffers[q] != comm->peers[q]) return HYMETA_ERR_INVALID_ARG;
      continue;
    }
    comm->peers[q] = static_cast<char*>(buffers[q]);
    comm->by_pointer[q] = true;
    hymeta::hold(buffers[q]);
  }
  comm->connected = true;
  return HYMETA_OK;
}

extern "C" int hymeta_p2p_set_wait_limit(hymeta_p2p_comm* comm, uint64_t ticks_100mhz) {
  if (comm == nullptr || ticks_100mhz == 0) return HYMETA_ERR_INVALID_ARG;
  comm->spin_ticks = ticks_100mhz;
  return HYMETA_OK;
}

extern "C" int hymeta_p2p_all_reduce(hymeta_p2p_comm* comm, void* data_bf16, int64_t n_bf16, float* stats, int64_t n_stats,
                                     float stats_scale, hymeta_stream_t stream) {
  using namespace hymeta;
  if (comm == nullptr || n_bf16 < 0 || n_stats < 0) return HYMETA_ERR_INVALID_ARG;
  if (!comm->connected) return HYMETA_ERR_INVALID_ARG;
  if ((n_bf16 > 0 && data_bf16 == nullptr) || (n_stats > 0 && stats == nullptr)) return HYMETA_ERR_INVALID_ARG;
  if (n_bf16 == 0 && n_stats == 0) return HYMETA_OK;
  if (n_bf16 % 8 != 0 || reinterpret_cast<uintptr_t>(data_bf16) % 16 != 0 || reinterpret_cast<uintptr_t>(stats) % 4 != 0)
    return HYMETA_ERR_UNSUPPORTED;
  P2pArgs a;
  a.units_stats = (n_stats + 3) / 4;
  a.units = a.units_stats + n_bf16 / 8;
  if (static_cast<size_t>(a.units) * 16 > comm->max_bytes) return HYMETA_ERR_SCRATCH_LIMIT;
  for (int q = 0; q < kMaxWorld; ++q) a.peers[q] = q < comm->world ? comm->peers[q] : nullptr;
  a.state = comm->state;
  a.data = static_cast<uint16_t*>(data_bf16);
  a.stats = stats;
  a.n_stats = n_stats;
  a.slot_bytes = static_cast<int64_t>(comm->slot_bytes);
  a.spin_ticks = comm->spin_ticks;
  a.rank = comm->rank;
  a.stats_scale = stats_scale;
  a.partials = hymeta::make_args<hymeta_gemm_partials>();
  a.out = nullptr;
  return launch_exchange<0>(comm->world, a, stream);
}

extern "C" int hymeta_p2p_all_reduce_partials(hymeta_p2p_comm* comm, const hymeta_gemm_partials* partials, void* out_bf16,
                                              hymeta_stream_t stream) {
  using namespace hymeta;
  if (comm == nullptr || !hymeta::args_ok(partials) || out_bf16 == nullptr || !comm->connected) return HYMETA_ERR_INVALID_ARG;
  const hymeta_gemm_partials& p = *partials;
  if (p.splits <= 0 || p.slabs == nullptr || p.M <= 0 || p.N <= 0) return HYMETA_ERR_INVALID_ARG;
  if (p.is_int32 && (p.x_scale == nullptr || p.w_scale == nullptr)) return HYMETA_ERR_INVALID_ARG;
  if (p.N % 8 != 0 || reinterpret_cast<uintptr_t>(out_bf16) % 16 != 0 || reinterpret_cast<uintptr_t>(p.slabs) % 16 != 0 ||
      reinterpret_cast<uintptr_t>(p.bias) % 16 != 0)
    return HYMETA_ERR_UNSUPPORTED;
  P2pArgs a;
  a.units_stats = 0;
  a.n_stats = 0;
  a.units = static_cast<int64_t>(p.M) * p.N / 8;
  if (static_cast<size_t>(a.units) * 16 > comm->max_bytes) return HYMETA_ERR_SCRATCH_LIMIT;
  for (int q = 0; q < kMaxWorld; ++q) a.peers[q] = q < comm->world ? comm->peers[q] : nullptr;
  a.state = comm->state;
  a.data = nullptr;
  a.stats = nullptr;
  a.slot_bytes = static_cast<int64_t>(comm->slot_bytes);
  a.spin_ticks = comm->spin_ticks;
  a.rank = comm->rank;
  a.stats_scale = 1.f;
  a.partials = p;
  a.out = static_cast<uint16_t*>(out_bf16);
  return p.is_int32 ? launch_exchange<2>(comm->world, a, stream) : launch_exchange<1>(comm->world, a, stream);
}

extern "C" int hymeta_p2p_timeouts(hymeta_p2p_comm* comm, uint32_t* count) {
  if (comm == nullptr || count == nullptr) return HYMETA_ERR_INVALID_ARG;
  return hipMemcpy(count, comm->state + 2, sizeof(uint32_t), hipMemcpyDeviceToHost) == hipSuccess ? HYMETA_OK : HYMETA_ERR_LAUNCH;
}

extern "C" int hymeta_p2p_destroy(hymeta_p2p_comm* comm) {
  if (comm == nullptr) return HYMETA_OK;
  (void)hipDeviceSynchronize();
  for (int q = 0; q < comm->world; ++q) {
    if (comm->opened[q]) (void)hipIpcCloseMemHandle(comm->peers[q]);
    if (comm->by_pointer[q]) hymeta::release(comm->peers[q]);
  }
  if (comm->peers[comm->rank]) {
    // parked at once, handed out again only when the in-process peers that point into it are gone too
    hymeta::release(comm->peers[comm->rank]);
    hymeta::park(comm->peers[comm->rank], comm->alloc_bytes, comm->device);
  }
  if (comm->state) (void)hipFree(comm->state);
  delete comm;
  return HYMETA_OK;
}
