// One-shot all-reduce / all-gather for small payloads over peer-mapped device memory (hipIpcMemHandle), MI355X.
//
// Why: the tensor-parallel Llama decode step moves 20 KB - 320 KB per all-reduce, 80 times per token. A ring collective is
// latency-bound there (xGMI is point to point: 7 links x ~153 GB/s per GPU, every hop ~2 us), and an RCCL call cannot be
// captured into the decode step's HIP graph on this stack (parallel.py: TorchDistComm.graph_safe = False). This collective is
// ONE ordinary kernel launch per rank — capturable, replayable, no host round trip:
// (per 16-KB chunk of the payload, one workgroup each:)
//   1. every rank copies its payload into its own STAGING slot (slot = epoch & 1), which all peers have mapped
//   2. release, then writes its epoch into flag[rank] of every peer's flag array (one 4-byte store per peer)
//   3. polls its own flag array until every peer's epoch has arrived, acquire
//   4. reads all staging slots and reduces them IN RANK ORDER (bit-identical result on every rank), or concatenates them
// A slot is reused at epoch e+2: a peer signals e+1 only after it finished reading epoch e, and nobody starts e+2 before it has
// seen every e+1 — no second barrier needed. The epoch lives in device memory and is advanced by the kernel itself, so a graph
// replay needs no new arguments. Cross-rank data moves with system-scope (sc0 sc1) stores and loads: it never sits in a
// non-coherent L2 or in a CU's L1 (MI355X_MICROARCH.md §inter-workgroup visibility). Polling is bounded: a missing peer makes
// the call fail (status word) instead of hanging the GPU.
//
// Reference counterpart: none — the reference is single-device inference; its only collectives are the training-side
// helpers of src/train/dist_utils.py:5-34. This is north_star's "RCCL all-reduce … overlapped" requirement in the form
// that fits 20-KB payloads.
#include "sx_common.h"

namespace sxk_comm {

struct ArP {
  float* data;                 // in/out: n floats (all-reduce) | in: n floats, out = gather_out (all-gather)
  float* gather_out;           // [world][n] or nullptr
  float* const* stage;         // device array [world]: peer r's staging area, 2 slots x cap floats
  unsigned* const* flags;      // device array [world]: peer r's flag array, [cap / chunk][world] epochs
  unsigned* epoch;             // this rank's epoch counters, one per chunk (device)
  unsigned* status;            // != 0 after a timeout
  int n, cap, rank, world, chunk;
  unsigned max_spin;
  int nb;                      // neighbour mode: signal / wait for rank - 1 and rank + 1 only; gather_out[n] = [prev's first half | next's second half]
};

// Cross-rank data moves as 8-byte (two floats) or 4-byte system-scope relaxed atomics: plain global_load/store … sc0 sc1, tracked
// by the compiler's own waitcnt insertion, never served from a non-coherent cache.
__device__ __forceinline__ void st_sys(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ float ld_sys(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void st_sys2(float* p, float2 v) {
  __hip_atomic_store((unsigned long long*)p, __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ float2 ld_sys2(const float* p) {
  return __builtin_bit_cast(float2, __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
}

// One workgroup per CHUNK of the payload (chunk = a fixed number of floats, so a chunk's staging range never moves between
// calls and the slot-reuse argument above holds per chunk); every chunk runs the whole protocol on its own epoch counter and
// its own row of flags, so a 320-KB payload is 20 independent 16-KB exchanges in flight at once instead of one serial one.
__global__ __launch_bounds__(1024) void oneshot_kernel(const ArP p) {
  __shared__ unsigned s_e, s_ok;
  const int tid = threadIdx.x, T = blockDim.x, b = blockIdx.x;
  const int lo = b * p.chunk, cnt = min(p.n - lo, p.chunk);
  if (tid == 0) { s_e = p.epoch[b] + 1u; s_ok = 1u; }
  __syncthreads();
  const unsigned e = s_e;
  const size_t slot = (size_t)(e & 1u) * (size_t)p.cap + (size_t)lo;
  const bool vec = ((p.n | lo) & 1) == 0 && (((uintptr_t)p.data | (uintptr_t)p.gather_out) & 7) == 0;   // cnt even, 8-byte aligned
  // 1. publish
  float* mine = p.stage[p.rank] + slot;
  const float* in = p.data + lo;
  if (vec) for (int i = 2 * tid; i < cnt; i += 2 * T) st_sys2(mine + i, *(const float2*)(in + i));
  else     for (int i = tid; i < cnt; i += T) st_sys(mine + i, in[i]);
  // every wave drains and releases its OWN staging stores at system scope before the barrier: a workgroup barrier alone does
  // not wait for other waves' outstanding global stores, so a single-thread fence would let a peer see the flag first
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // 2. signal every peer, 3. wait for every peer (threads 0 .. world-1, one peer each)
  const bool is_peer = p.nb ? (tid == p.rank - 1 || tid == p.rank + 1) : true;    // (a rank is its own peer in mode 0: the flag it polls)
  if (tid < p.world && is_peer) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(p.flags[tid] + (size_t)b * p.world + p.rank, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned* f = p.flags[p.rank] + (size_t)b * p.world + tid;
    unsigned spins = 0;
    // epochs only grow; (int) difference tolerates wrap-around
    while ((int)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - e) < 0) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > p.max_spin) { s_ok = 0u; break; }
    }
  }
  __syncthreads();
  if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  __syncthreads();
  if (!s_ok) {                                                       // a peer never arrived: report, do not hang
    if (tid == 0) { atomicCAS(p.status, 0u, e ? e : 1u); p.epoch[b] = e; }   // sticky: the FIRST failed epoch stays recorded
    return;
  }
  // 4. reduce in rank order / gather
  if (p.nb) {
    // neighbour exchange: element i < n/2 comes from rank - 1 (its last row), i >= n/2 from rank + 1 (its first row) — the SAME
    // index in the neighbour's slot, so chunk b only reads what the neighbour's chunk b published
    const int half = p.n >> 1;
    float* dst = p.gather_out + lo;
    for (int i = 2 * tid; i < cnt; i += 2 * T) {
      const int src_rank = (lo + i < half) ? p.rank - 1 : p.rank + 1;
      float2 v2 = {0.f, 0.f};
      if (src_rank >= 0 && src_rank < p.world) v2 = ld_sys2(p.stage[src_rank] + slot + i);
      *(float2*)(dst + i) = v2;
    }
  } else if (p.gather_out) {
    for (int r = 0; r < p.world; ++r) {
      const float* src = p.stage[r] + slot;
      float* dst = p.gather_out + (size_t)r * p.n + lo;
      if (vec) for (int i = 2 * tid; i < cnt; i += 2 * T) *(float2*)(dst + i) = ld_sys2(src + i);
      else     for (int i = tid; i < cnt; i += T) dst[i] = ld_sys(src + i);
    }
  } else {
    float* out = p.data + lo;
    if (vec) {
      for (int i = 2 * tid; i < cnt; i += 2 * T) {
        float2 acc = ld_sys2(p.stage[0] + slot + i);
        for (int r = 1; r < p.world; ++r) { const float2 v = ld_sys2(p.stage[r] + slot + i); acc.x += v.x; acc.y += v.y; }
        *(float2*)(out + i) = acc;
      }
    } else {
      for (int i = tid; i < cnt; i += T) {
        float acc = ld_sys(p.stage[0] + slot + i);
        for (int r = 1; r < p.world; ++r) acc += ld_sys(p.stage[r] + slot + i);
        out[i] = acc;
      }
    }
  }
  if (tid == 0) p.epoch[b] = e;
}

}  // namespace sxk_comm
using namespace sxk_comm;

extern "C" int sx_comm_alloc(void** ptr, uint64_t bytes) {
  SX_CHECK(ptr && bytes > 0, "sx_comm_alloc: bad arguments");
  // Fine-grained, uncached device memory (MTYPE UC): these buffers are written by PEER GPUs over xGMI and polled / read by
  // kernels that are already in flight. Ordinary hipMalloc memory is coarse-grained — coherent only at kernel boundaries — so a
  // peer's flag store could sit behind a stale line of the local XCD's L2 no matter how the load is scoped. (Same choice as
  // RCCL's and vLLM's custom-all-reduce signal buffers.)
  hipError_t e = hipExtMallocWithFlags(ptr, bytes, hipDeviceMallocUncached);
  if (e != hipSuccess) { (void)hipGetLastError(); e = hipExtMallocWithFlags(ptr, bytes, hipDeviceMallocFinegrained); }
  SX_CHECK(e == hipSuccess, "sx_comm_alloc: hipExtMallocWithFlags(%llu, uncached | fine-grained) failed: %s", (unsigned long long)bytes, hipGetErrorString(e));
  e = hipMemset(*ptr, 0, bytes);
  SX_CHECK(e == hipSuccess, "sx_comm_alloc: hipMemset failed: %s", hipGetErrorString(e));
  return SX_OK;
}

extern "C" int sx_comm_free(void* ptr) {
  if (ptr) (void)hipFree(ptr);
  return SX_OK;
}

// 64-byte opaque handle of a sx_comm_alloc buffer, to be sent to the peer processes (any byte transport)
extern "C" int sx_ipc_export(void* ptr, unsigned char* handle64) {
  SX_CHECK(ptr && handle64, "sx_ipc_export: null pointer");
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
  hipIpcMemHandle_t h;
  const hipError_t e = hipIpcGetMemHandle(&h, ptr);
  SX_CHECK(e == hipSuccess, "sx_ipc_export: hipIpcGetMemHandle failed: %s (is HSA_ENABLE_IPC_MODE_LEGACY=0 set?)", hipGetErrorString(e));
  memcpy(handle64, &h, 64);
  return SX_OK;
}

extern "C" int sx_ipc_open(const unsigned char* handle64, void** ptr) {
  SX_CHECK(ptr && handle64, "sx_ipc_open: null pointer");
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, 64);
  const hipError_t e = hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess);
  SX_CHECK(e == hipSuccess, "sx_ipc_open: hipIpcOpenMemHandle failed: %s", hipGetErrorString(e));
  return SX_OK;
}

extern "C" int sx_ipc_close(void* ptr) {
  if (ptr) (void)hipIpcCloseMemHandle(ptr);
  return SX_OK;
}

extern "C" int sx_allreduce_oneshot(const sx_oneshot_args* a, void* stream) {
  SX_CHECK(a && a->data && a->stage && a->flags && a->epoch && a->status, "sx_allreduce_oneshot: null pointer");
  SX_CHECK(a->world >= 1 && a->world <= 64 && a->rank >= 0 && a->rank < a->world, "sx_allreduce_oneshot: rank %d / world %d", a->rank, a->world);
  SX_CHECK(a->n > 0 && a->n <= a->cap, "sx_allreduce_oneshot: n=%d exceeds the staging capacity %d", a->n, a->cap);
  ArP p;
  p.data = (float*)a->data; p.gather_out = (float*)a->gather_out;
  p.stage = (float* const*)a->stage; p.flags = (unsigned* const*)a->flags;
  p.epoch = (unsigned*)a->epoch; p.status = (unsigned*)a->status;
  p.n = a->n; p.cap = a->cap; p.rank = a->rank; p.world = a->world;
  p.chunk = a->chunk > 0 ? a->chunk : SX_ONESHOT_CHUNK;
  SX_CHECK((p.chunk & 1) == 0 && a->cap % p.chunk == 0, "sx_allreduce_oneshot: chunk %d must be even and divide the capacity %d", p.chunk, a->cap);
  p.nb = a->mode == 1 ? 1 : 0;
  SX_CHECK(a->mode == 0 || a->mode == 1, "sx_allreduce_oneshot: mode %d", a->mode);
  SX_CHECK(!p.nb || (a->gather_out && a->n % 4 == 0 && (((uintptr_t)a->data | (uintptr_t)a->gather_out) & 7) == 0),
           "sx_allreduce_oneshot: neighbour mode needs gather_out, n %% 4 == 0 and 8-byte aligned buffers");
  p.max_spin = a->max_spin ? a->max_spin : (1u << 24);     // ~16 s of polling: rank skew from lazy module loads is seconds
  hipLaunchKernelGGL(oneshot_kernel, dim3((a->n + p.chunk - 1) / p.chunk), dim3(1024), 0, (hipStream_t)stream, p);
  SX_HIP_LAUNCH_CHECK();
  return SX_OK;
}
