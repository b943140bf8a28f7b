// mask_select.hip -- the index selection of logits_mask (Frustum-PVCNN) on the device.
//
// Reference: modules/functional/sampling.py:51-84.  After the foreground mask (logits[:,0] < logits[:,1]) the reference
// walks the batch in Python: per cloud a `nonzero()` (device -> host sync), numpy draws on the host
// (np.random.choice without replacement, np.random.shuffle) and an indexed assignment back to the device -- B syncs per
// forward.  Here one workgroup per cloud does it without leaving the GPU:
//   1. stable compaction of the foreground point ids (wave ballots + prefix), k = their count;
//   2. selection of M list positions:
//        k >= M : M distinct candidates, in random order                  [np.random.choice(k, M, replace=False)]
//        0<k<M  : every candidate M / k times + (M % k) distinct extra ones, all shuffled
//                                                        [arange(k).repeat(M // k) ++ choice(k, M % k, False); shuffle]
//        k == 0 : index 0 everywhere (the reference leaves its zero-initialised row)
//      either from caller-supplied `choices` (B, M) -- positions into the candidate list, e.g. numpy's own draws: the
//      parity mode, bit-identical to the reference given the same draws -- or from a counter-based Philox4x32-10 stream
//      keyed by a seed that lives in device memory (no host round trip): "M distinct of k in random order" = the M
//      smallest of k random keys, ordered by key; a shuffle = ordering by fresh random keys.
#include "common.h"

namespace pvcnn {

constexpr int kSelThreads = 1024;
constexpr int kSelMaxN = 8192;        // candidates + keys live in LDS (rank counting is O(k^2 / threads))

__device__ __forceinline__ uint32_t mulhi32(uint32_t a, uint32_t b) { return __umulhi(a, b); }

// Philox4x32-10 (Salmon et al., SC'11): counter (c0..c3), key (k0,k1) -> 4 random words
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = mulhi32(M0, c.x), lo0 = M0 * c.x, hi1 = mulhi32(M1, c.z), lo1 = M1 * c.z;
    c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
    k.x += W0; k.y += W1;
  }
  return c;
}

// rank of (key, id) among n (key, id) pairs in LDS (ties broken by id): O(n) per caller
__device__ __forceinline__ int rank_of(const uint32_t *keys, int n, uint32_t key, int id) {
  int r = 0;
  for (int j = 0; j < n; ++j) {
    const uint32_t kj = keys[j];
    r += (kj < key || (kj == key && j < id)) ? 1 : 0;
  }
  return r;
}

__global__ __launch_bounds__(kSelThreads) void mask_select_kernel(const uint8_t *__restrict__ mask, int N, int M,
                                                                 const int32_t *__restrict__ choices,
                                                                 const int64_t *__restrict__ seed, int32_t *__restrict__ selected,
                                                                 int32_t *__restrict__ count) {
  extern __shared__ __attribute__((aligned(16))) int sel_lds[];
  int *cand = sel_lds;                                        // [N]  foreground point ids, ascending
  uint32_t *keys = reinterpret_cast<uint32_t *>(cand + N);    // [max(N, M)]
  int *entry = reinterpret_cast<int *>(keys + max(N, M));     // [M]   list positions before the shuffle (k < M case)
  __shared__ int wave_cnt[kSelThreads / 64];
  __shared__ int total;
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const uint8_t *mk = mask + (size_t)b * N;
  int32_t *out = selected + (size_t)b * M;

  // ---- 1. stable compaction, chunks of 1024 points ----
  if (tid == 0) total = 0;
  __syncthreads();
  for (int base = 0; base < N; base += kSelThreads) {
    const int i = base + tid;
    const bool fg = i < N && mk[i] != 0;
    const unsigned long long bal = __ballot(fg);
    if (lane == 0) wave_cnt[wave] = __popcll(bal);
    __syncthreads();
    int off = total;
    for (int w = 0; w < wave; ++w) off += wave_cnt[w];
    if (fg) cand[off + __popcll(bal & ((1ull << lane) - 1ull))] = i;
    __syncthreads();
    if (tid == 0) { int t = total; for (int w = 0; w < kSelThreads / 64; ++w) t += wave_cnt[w]; total = t; }
    __syncthreads();
  }
  const int k = total;
  if (count && tid == 0) count[b] = k;
  if (k == 0) {
    for (int m = tid; m < M; m += kSelThreads) out[m] = 0;
    return;
  }
  // ---- 2a. parity mode: the caller's draws ----
  if (choices) {
    const int32_t *ch = choices + (size_t)b * M;
    for (int m = tid; m < M; m += kSelThreads) out[m] = cand[min(max(ch[m], 0), k - 1)];
    return;
  }
  // ---- 2b. device RNG ----
  const uint2 key = make_uint2((uint32_t)seed[0], (uint32_t)((uint64_t)seed[0] >> 32));
  const uint32_t stream = (uint32_t)seed[1];
  for (int i = tid; i < k; i += kSelThreads) keys[i] = philox4x32_10(make_uint4(i, b, stream, 0u), key).x;
  __syncthreads();
  if (k >= M) {
    for (int i = tid; i < k; i += kSelThreads) {
      const int r = rank_of(keys, k, keys[i], i);
      if (r < M) out[r] = cand[i];
    }
    return;
  }
  const int rep = M / k, extra = M - rep * k;
  for (int s = tid; s < rep * k; s += kSelThreads) entry[s] = s / rep;             // arange(k).repeat(M // k)
  for (int i = tid; i < k; i += kSelThreads) {
    const int r = rank_of(keys, k, keys[i], i);
    if (r < extra) entry[rep * k + r] = i;                                           // choice(k, M % k, replace=False)
  }
  __syncthreads();
  for (int s = tid; s < M; s += kSelThreads) keys[s] = philox4x32_10(make_uint4(s, b, stream, 1u), key).x;
  __syncthreads();
  for (int s = tid; s < M; s += kSelThreads) out[rank_of(keys, M, keys[s], s)] = cand[entry[s]];   // shuffle
}

}  // namespace pvcnn

using namespace pvcnn;

extern "C" int pvcnn_mask_select(const uint8_t *mask, int B, int N, int M, const int32_t *choices, const int64_t *seed,
                                 int32_t *selected, int32_t *count, void *stream) {
  PVCNN_REQUIRE(B >= 0 && N >= 0 && M >= 0, "negative size");
  if (B == 0 || M == 0) return 0;
  PVCNN_REQUIRE(selected && (mask || N == 0), "null pointer");
  PVCNN_REQUIRE(choices || seed, "either `choices` (parity mode) or a device `seed` (device RNG) is required");
  PVCNN_REQUIRE(N <= kSelMaxN && M <= kSelMaxN, "N or M beyond the LDS-resident selection (8192)");
  PVCNN_REQUIRE(B <= 65535 * 32767, "batch too large");
  const size_t lds = ((size_t)N + (size_t)(N > M ? N : M) + (size_t)M) * 4;
  if (lds > 64 * 1024) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(mask_select_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("mask_select: LDS attribute: %s", hipGetErrorString(e)); return (int)e; }
  }
  hipLaunchKernelGGL(mask_select_kernel, dim3(B), dim3(kSelThreads), lds, static_cast<hipStream_t>(stream), mask, N, M, choices, seed,
                     selected, count);
  return check_launch("mask_select");
}
