// common.h -- shared host/device helpers of libpvcnn_hip.so (gfx950 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/pvcnn_hip.h"

namespace pvcnn {

constexpr int kWave = 64;                      // CDNA wavefront
constexpr int kLdsBytesPerCU = 160 * 1024;     // gfx950 LDS per CU (and max per workgroup)
constexpr int kNumCU = 256;                    // MI355X

// thread-local error string (api.hip)
void set_error(const char *fmt, ...);
// hipGetLastError() -> return code (+ error string); 0 when clean
int check_launch(const char *what);

__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
// log2(R) when R is a power of two >= 4 (grid rows can then be staged padded in LDS, slab.h), else 0
inline int grid_pad_shift(int R) { int s = 0; while ((1 << s) < R) ++s; return ((1 << s) == R && R >= 4) ? s : 0; }
__host__ __device__ inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// tables longer than this keep the separate reduce launch (one wave peeks a folded table: 64 words per step)
constexpr long kFoldTableMax = 1024;
// out[0] = max over out[1 .. T] of an amax buffer whose table has just been written on stream s (conv3d_bf16.hip)
int launch_amax_reduce(uint32_t *out, long T, hipStream_t s);

// One weight of a batched image refresh (pvcnn_conv3d_weight_split_pair_batch / pvcnn_pwconv_weight_split_pair_batch): ten int64 words
// in device memory, filled on the host by pvcnn_*_weight_split_pair_entry.  Workgroups [row_begin, row_begin + rows_f + rows_b) of the
// launch write this weight's forward and backward-data images, one (padded) output row each.
struct SplitEntry {
  const float *w;
  uint16_t *wts_f;
  int *wexp_f;
  uint16_t *wts_b;
  int *wexp_b;
  long long Co, Ci, rows_f, tm /* 1x1 only: TM_f | TM_b << 32 */, row_begin;
};
static_assert(sizeof(SplitEntry) == 80, "ten 8-byte words");

#ifdef __HIPCC__
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding GLOBAL store of the
// wave (s_waitcnt vmcnt(0): its release fence), which serialises "store a tile, barrier, load the next one" loops on the
// full write latency; kernels whose global stores are never read back in the same launch use this instead, so the stores
// drain while the next phase runs.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// "Last workgroup done" WITHOUT fences.  VALIDATED ON gfx950 ONLY: what orders a published word before the ticket is that returning
// device-scope atomics are performed where all XCDs meet -- an observed property of this chip, not the HIP memory model's wording.  A workgroup PUBLISHES the few words the finalize step needs with returning device-scope
// atomics (publish32 / publish64: an exchange executes at the level where all XCDs meet and its return says it has), then takes a
// ticket; the workgroup that takes the last of `total` tickets reads the published words with device-scope atomic loads (peek32 /
// peek64: they do not hit in this XCD's L2) and finishes the job.  No __threadfence(): on gfx950 an agent-scope release fence writes
// back the XCD's whole L2 -- measured (profiles/ab/r05c_last_workgroup_fences.md): with one fence per workgroup the folded kernels of a
// PVCNN step (~100 k workgroups) cost 6.3 ms MORE than the ~26 launches of ~5 us they replaced.  *ticket must be 0 when the launch starts and is 0
// again when it ends (the last workgroup resets it: graph replays and later launches reuse the word).
// ticket_take: every workgroup, AFTER its published words have returned; contains __syncthreads(); true in every thread of exactly one
// workgroup.
__device__ __forceinline__ void publish32(uint32_t *p, uint32_t v) {
  const uint32_t old = atomicExch(p, v);
  asm volatile("" ::"v"(old) : "memory");                   // the exchange has returned (performed) before anything below is issued
}
__device__ __forceinline__ void publish64(unsigned long long *p, unsigned long long v) {
  const unsigned long long old = atomicExch(p, v);
  asm volatile("" ::"v"(old) : "memory");
}
__device__ __forceinline__ uint32_t peek32(const uint32_t *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long peek64(const unsigned long long *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// ticket_take_wave: the same for ONE wave of the workgroup (all 64 lanes call it, after their publishes have returned; the other waves
// have left: a workgroup of short-lived waves must not wait at a barrier for two atomic round trips -- measured, r05d)
__device__ __forceinline__ bool ticket_take_wave(unsigned *ticket, unsigned total) {
  unsigned t = 0u;
  if ((threadIdx.x & 63) == 0) {
    t = atomicAdd(ticket, 1u);
    if (t == total - 1u) atomicExch(ticket, 0u);
  }
  return (unsigned)__builtin_amdgcn_readfirstlane((int)t) == total - 1u;
}
__device__ __forceinline__ bool ticket_take(unsigned *ticket, unsigned total) {
  __shared__ int s_last_workgroup;
  __syncthreads();                                           // every thread's publishes have returned
  if (threadIdx.x == 0) {
    const unsigned t = atomicAdd(ticket, 1u);
    s_last_workgroup = (t == total - 1u) ? 1 : 0;
    if (t == total - 1u) atomicExch(ticket, 0u);
  }
  __syncthreads();
  return s_last_workgroup != 0;
}

// out[0] = max over the table out[1 .. T] of an amax buffer, by the calling workgroup (every thread calls; <= 1024 threads).  Used by
// the workgroup that takes the last ticket in the kernels that PUBLISH the table (see ticket_take): the entries are peeked.
// (single wave: the 64 lanes of the wave that took the last ticket; T small -- see the callers' bounds)
__device__ __forceinline__ void amax_table_max_wave(uint32_t *out, long T) {
  uint32_t m = 0;
  for (long i = threadIdx.x & 63; i < T; i += 64) m = max(m, peek32(out + 1 + i));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
  if ((threadIdx.x & 63) == 0) out[0] = m;
}
__device__ __forceinline__ void amax_table_max(uint32_t *out, long T) {
  __shared__ uint32_t s_amax_red[16];
  uint32_t m = 0;
  for (long i = threadIdx.x; i < T; i += blockDim.x) m = max(m, peek32(out + 1 + i));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
  if ((threadIdx.x & 63) == 0) s_amax_red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)((blockDim.x + 63) >> 6); ++w) m = max(m, s_amax_red[w]);
    out[0] = m;
  }
}

// (ABI v12) max |x| of the tensor behind an amax buffer, for every thread of the calling workgroup (all of them call; two barriers):
// T > 0: the maximum of the table buf[1 .. T] -- word [0] is not read, its producer may have left it unwritten (PVCNN_TABLE_ONLY:
// no one-workgroup launch behind the table pass); T == 0: buf[0] (a 1-word buffer, or an amax buffer with its word [0]).
__device__ __forceinline__ uint32_t amax_table_value(const uint32_t *__restrict__ buf, long T) {
  if (T <= 0) return buf[0];
  __shared__ uint32_t s_amax_val[17];
  uint32_t m = 0;
  for (long i = threadIdx.x; i < T; i += blockDim.x) m = max(m, buf[1 + i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
  if ((threadIdx.x & 63) == 0) s_amax_val[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)((blockDim.x + 63) >> 6); ++w) m = max(m, s_amax_val[w]);
    s_amax_val[16] = m;
  }
  __syncthreads();
  return s_amax_val[16];
}

// Sum each of 16 per-lane values over the 32 lanes of a half-wave (lanes that differ in bits 0..4) with a
// transposing butterfly: every step halves the number of values a lane carries (the lane keeps one half and
// ships the other), so it takes 8+4+2+1+1 = 16 shuffles instead of 16*5.  Returns, in lane j, the total of
// value index (j >> 1) & 15 (both lanes of a pair hold the same total).
__device__ __forceinline__ float half_wave_sum16(const float (&v)[16], int lane) {
  float a[8], b[4], c[2];
  const bool b4 = lane & 16, b3 = lane & 8, b2 = lane & 4, b1 = lane & 2;
#pragma unroll
  for (int r = 0; r < 8; ++r) a[r] = (b4 ? v[r + 8] : v[r]) + __shfl_xor(b4 ? v[r] : v[r + 8], 16);
#pragma unroll
  for (int r = 0; r < 4; ++r) b[r] = (b3 ? a[r + 4] : a[r]) + __shfl_xor(b3 ? a[r] : a[r + 4], 8);
#pragma unroll
  for (int r = 0; r < 2; ++r) c[r] = (b2 ? b[r + 2] : b[r]) + __shfl_xor(b2 ? b[r] : b[r + 2], 4);
  float d = (b1 ? c[1] : c[0]) + __shfl_xor(b1 ? c[0] : c[1], 2);
  d += __shfl_xor(d, 1);
  return d;
}
// The same for 8 values per lane (7 + 2 shuffles): returns, in lane j, the total of value index (j >> 2) & 7.  Every total is the
// butterfly over the lane bits 4, 3, 2, 1, 0 in that order, like half_wave_sum16's: value q of an 8-value call carries the same
// bits as value q of a 16-value call on the same numbers (fp32 addition commutes; the tree is the same).
__device__ __forceinline__ float half_wave_sum8(const float (&v)[8], int lane) {
  float a[4], b[2];
  const bool b4 = lane & 16, b3 = lane & 8, b2 = lane & 4;
#pragma unroll
  for (int r = 0; r < 4; ++r) a[r] = (b4 ? v[r + 4] : v[r]) + __shfl_xor(b4 ? v[r] : v[r + 4], 16);
#pragma unroll
  for (int r = 0; r < 2; ++r) b[r] = (b3 ? a[r + 2] : a[r]) + __shfl_xor(b3 ? a[r] : a[r + 2], 8);
  float c = (b2 ? b[1] : b[0]) + __shfl_xor(b2 ? b[0] : b[1], 4);
  c += __shfl_xor(c, 2);
  c += __shfl_xor(c, 1);
  return c;
}
#endif

}  // namespace pvcnn

// ---- phase clocks (tools/probe/: a SEPARATE build of a kernel's translation unit with -DPVCNN_PHASE_PROBE; the library itself never
// defines it, there these macros are nothing).  s_memtime per wave at named points of a kernel, the time since the previous point
// added to the slot's accumulator (scalar registers); at the end lane 0 of every wave adds its accumulators to
// pvcnn::phase_probe_buf[slot] and counts itself in slot 31.  A stamp waits for the wave's outstanding LDS / scalar loads
// (s_waitcnt lgkmcnt(0)): an instrumented kernel runs ~10 % slower than the real one; the SHARES are what is read.
#ifdef PVCNN_PHASE_PROBE
namespace pvcnn { static __device__ unsigned long long *phase_probe_buf = nullptr; }     // (one kernel translation unit per probe build)
#ifdef PVCNN_PHASE_PROBE_SETTER
extern "C" __attribute__((visibility("default"))) int pvcnn_probe_set_buffer(void *device_ptr) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(pvcnn::phase_probe_buf), &device_ptr, sizeof(device_ptr));
}
#endif
#define PVCNN_PROBE_BEGIN() unsigned long long probe_t_ = __builtin_amdgcn_s_memtime(); unsigned probe_acc_[16] = {}
#define PVCNN_PROBE(k) do { const unsigned long long probe_n_ = __builtin_amdgcn_s_memtime(); probe_acc_[k] += (unsigned)(probe_n_ - probe_t_); probe_t_ = probe_n_; } while (0)
#define PVCNN_PROBE_END()                                                                                              \
  do {                                                                                                                   \
    if ((threadIdx.x & 63) == 0 && pvcnn::phase_probe_buf != nullptr) {                                                  \
      for (int probe_k_ = 0; probe_k_ < 16; ++probe_k_) atomicAdd(pvcnn::phase_probe_buf + probe_k_, (unsigned long long)probe_acc_[probe_k_]); \
      atomicAdd(pvcnn::phase_probe_buf + 31, 1ull);                                                                      \
    }                                                                                                                    \
  } while (0)
#else
#define PVCNN_PROBE_BEGIN() do { } while (0)
#define PVCNN_PROBE(k) do { } while (0)
#define PVCNN_PROBE_END() do { } while (0)
#endif

#define PVCNN_REQUIRE(cond, msg)                                       \
  do {                                                                 \
    if (!(cond)) {                                                     \
      pvcnn::set_error("%s: %s", __func__, msg);                       \
      return PVCNN_ERR_INVALID_ARGUMENT;                               \
    }                                                                  \
  } while (0)

#define PVCNN_HIP_TRY(expr)                                            \
  do {                                                                 \
    hipError_t e_ = (expr);                                            \
    if (e_ != hipSuccess) {                                            \
      pvcnn::set_error("%s: %s failed: %s", __func__, #expr, hipGetErrorString(e_)); \
      return static_cast<int>(e_);                                     \
    }                                                                  \
  } while (0)
