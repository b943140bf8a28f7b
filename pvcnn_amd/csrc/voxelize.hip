// voxelize.hip -- avg_voxelize forward / backward for gfx950.
//
// Reference: voxelization/vox.cu:18-72 (grid_stats + per-point float atomicAdd scatter from B
// workgroups, run-to-run non-deterministic) and vox.cu:86-110 (backward).
//
// Forward here is a per-cloud counting sort followed by an atomic-free CSR gather:
//   vox_prep_kernel   (1 workgroup / cloud, histogram of the R^3 voxels in LDS)
//       ind[b,i], cnt[b,v], start[b,v] = exclusive prefix of cnt, order[b,*] = point ids
//       grouped by voxel, ASCENDING point index inside every voxel (stable);
//   vox_gather_kernel (lane = 4 consecutive voxels, loop over a channel tile)
//       out[b,c,v] = sum over the voxel's points, in ascending point index, of
//                    feat[b,c,p] * (1/cnt) -- each addend pre-multiplied like vox.cu:66-68 --
//       and every voxel (occupied or not) is written exactly once with 16-byte stores.
// Result: no memset pass, no float atomics, bit-reproducible, and bit-identical to a serial
// point-order evaluation (what oracle/pvcnn_oracle.c computes).  HBM traffic = compulsory.
// Grids with R^3 > kMaxLdsVoxels (R > 32) use the atomic fallback at the bottom.
#include "slab.h"

namespace pvcnn {

constexpr int kPrepThreads = 1024;
constexpr int kMaxLdsVoxels = 32768;   // R <= 32: histogram (padded) fits the 160 KiB LDS

// conflict-free padding for the "thread owns 32 consecutive bins" scan phase
__device__ __forceinline__ int pad(int v) { return v + (v >> 5); }

__global__ __launch_bounds__(kPrepThreads) void vox_prep_kernel(
    const int32_t *__restrict__ coords, int N, int R, int S, int32_t *__restrict__ ind,
    int32_t *__restrict__ cnt, int32_t *__restrict__ start, int32_t *__restrict__ order_tmp,
    int32_t *__restrict__ order) {
  extern __shared__ __attribute__((aligned(16))) int hist[];   // pad(S) bins + 32 wave totals
  const int b = blockIdx.x, tid = threadIdx.x;
  const int R2 = R * R;
  const int SP = pad(S) + 1;
  int *wave_tot = hist + SP;
  coords += (size_t)b * 3 * N;
  ind += (size_t)b * N;
  cnt += (size_t)b * S;
  start += (size_t)b * S;
  order_tmp += (size_t)b * N;
  order += (size_t)b * N;

  for (int i = tid; i < SP; i += kPrepThreads) hist[i] = 0;
  __syncthreads();
  // pass 1: voxel id of every point (vox.cu:31) + histogram (vox.cu:32)
  for (int i = tid; i < N; i += kPrepThreads) {
    int v = coords[i] * R2 + coords[i + N] * R + coords[i + 2 * N];
    v = min(max(v, 0), S - 1);   // reference: unchecked (UB when out of range)
    ind[i] = v;
    atomicAdd(&hist[pad(v)], 1);
  }
  __syncthreads();
  // cnt -> global; exclusive scan of the bins (thread t owns bins [t*per, t*per+per))
  for (int v = tid; v < S; v += kPrepThreads) cnt[v] = hist[pad(v)];
  const int per = (S + kPrepThreads - 1) / kPrepThreads;   // <= 32
  const int v0 = tid * per;
  int local = 0;
  for (int k = 0; k < per; ++k)
    if (v0 + k < S) local += hist[pad(v0 + k)];
  // wave inclusive scan of `local`
  int incl = local;
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    const int up = __shfl_up(incl, d);
    if ((tid & (kWave - 1)) >= d) incl += up;
  }
  const int wave = tid >> 6, lane = tid & 63;
  __syncthreads();   // all reads of hist for cnt[] done before anybody rewrites bins
  if (lane == kWave - 1) wave_tot[wave] = incl;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; ++w) base += wave_tot[w];
  int run = base + incl - local;   // exclusive prefix of this thread's first bin
  for (int k = 0; k < per; ++k)
    if (v0 + k < S) {
      const int c = hist[pad(v0 + k)];
      hist[pad(v0 + k)] = run;   // becomes the running cursor of pass 2
      run += c;
    }
  __syncthreads();
  for (int v = tid; v < S; v += kPrepThreads) start[v] = hist[pad(v)];
  __syncthreads();
  // pass 2: unordered placement inside each voxel's segment
  for (int i = tid; i < N; i += kPrepThreads) {
    const int pos = atomicAdd(&hist[pad(ind[i])], 1);
    order_tmp[pos] = i;
  }
  __syncthreads();   // workgroup-scope release/acquire: order_tmp / start visible to the block
  // pass 3: stable rank = number of smaller point ids in my voxel's segment.  Work is per point,
  // so a degenerate cloud (all points in one voxel) costs O(N^2 / threads), never a serial lane.
  for (int i = tid; i < N; i += kPrepThreads) {
    const int v = ind[i];
    const int end = hist[pad(v)];            // cursor finished at start + cnt
    const int beg = start[v];
    int rank = 0;
    for (int q = beg; q < end; ++q) rank += (order_tmp[q] < i) ? 1 : 0;
    order[beg + rank] = i;
  }
}

template <int VEC>
__global__ __launch_bounds__(256) void vox_gather_kernel(
    const float *__restrict__ feat, const int32_t *__restrict__ cnt, const int32_t *__restrict__ start,
    const int32_t *__restrict__ order, int C, int N, int S, int CT, float *__restrict__ out) {
  const int b = blockIdx.z;
  const int v0 = (blockIdx.x * 256 + threadIdx.x) * VEC;
  if (v0 >= S) return;
  int n[VEC], s0[VEC];
  float rcp[VEC];
  if constexpr (VEC == 4) {
    const int4 c4 = ld4(cnt + (size_t)b * S + v0), s4 = ld4(start + (size_t)b * S + v0);
    n[0] = c4.x; n[1] = c4.y; n[2] = c4.z; n[3] = c4.w;
    s0[0] = s4.x; s0[1] = s4.y; s0[2] = s4.z; s0[3] = s4.w;
  } else {
    n[0] = cnt[(size_t)b * S + v0];
    s0[0] = start[(size_t)b * S + v0];
  }
#pragma unroll
  for (int q = 0; q < VEC; ++q) rcp[q] = (n[q] > 0) ? (float)(1.0 / (double)(float)n[q]) : 0.0f;
  const int32_t *ord = order + (size_t)b * N;
  const int c0 = blockIdx.y * CT, c1 = min(C, c0 + CT);
  for (int c = c0; c < c1; ++c) {
    const float *f = feat + ((size_t)b * C + c) * N;
    float acc[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) {
      acc[q] = 0.0f;
      for (int k = 0; k < n[q]; ++k) acc[q] = acc[q] + f[ord[s0[q] + k]] * rcp[q];
    }
    float *o = out + ((size_t)b * C + c) * S + v0;
    if constexpr (VEC == 4) st4(o, acc[0], acc[1], acc[2], acc[3]); else o[0] = acc[0];
  }
}

// ---- fallback for R^3 > kMaxLdsVoxels: global int atomics + float atomics (reference-like) ----
__global__ __launch_bounds__(256) void vox_stats_atomic_kernel(const int32_t *__restrict__ coords, int N,
                                                               int R, int S, int32_t *__restrict__ ind,
                                                               int32_t *__restrict__ cnt) {
  const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const int32_t *co = coords + (size_t)b * 3 * N;
  int v = co[i] * R * R + co[i + N] * R + co[i + 2 * N];
  v = min(max(v, 0), S - 1);
  ind[(size_t)b * N + i] = v;
  atomicAdd(cnt + (size_t)b * S + v, 1);
}

__global__ __launch_bounds__(256) void vox_scatter_atomic_kernel(
    const float *__restrict__ feat, const int32_t *__restrict__ ind, const int32_t *__restrict__ cnt, int C,
    int N, int S, int CT, float *__restrict__ out) {
  const int b = blockIdx.z, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const int pos = ind[(size_t)b * N + i];
  const int cur = cnt[(size_t)b * S + pos];
  if (cur <= 0) return;
  const float rcp = (float)(1.0 / (double)(float)cur);
  const int c0 = blockIdx.y * CT, c1 = min(C, c0 + CT);
  for (int c = c0; c < c1; ++c)
    atomicAdd(out + ((size_t)b * C + c) * S + pos, feat[((size_t)b * C + c) * N + i] * rcp);
}

static size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

}  // namespace pvcnn

using namespace pvcnn;

extern "C" size_t pvcnn_avg_voxelize_fwd_workspace_bytes(int B, int N, int R) {
  if (B <= 0 || N < 0 || R <= 0) return 0;
  const size_t S = (size_t)R * R * R;
  if (S > (size_t)kMaxLdsVoxels) return 16;   // atomic fallback needs no scratch
  return align16((size_t)B * S * 4) + 2 * align16((size_t)B * N * 4) + 16;
}

extern "C" int pvcnn_avg_voxelize_fwd(const float *feat, const int32_t *coords, int B, int C, int N, int R,
                                      float *out, int32_t *ind, int32_t *cnt, void *workspace,
                                      size_t workspace_bytes, void *stream) {
  PVCNN_REQUIRE(B >= 0 && C >= 0 && N >= 0 && R > 0, "negative size");
  PVCNN_REQUIRE((long)R * R * R <= 0x7fffffffL / 4, "resolution too large");
  const int S = R * R * R;
  if (B == 0) return 0;
  PVCNN_REQUIRE(out && ind && cnt && (feat || C == 0 || N == 0) && (coords || N == 0), "null pointer");
  PVCNN_REQUIRE(B <= 65535, "batch > 65535");
  hipStream_t s = static_cast<hipStream_t>(stream);

  if (S > kMaxLdsVoxels) {
    PVCNN_HIP_TRY(hipMemsetAsync(cnt, 0, (size_t)B * S * 4, s));
    PVCNN_HIP_TRY(hipMemsetAsync(out, 0, (size_t)B * C * S * 4, s));
    if (N == 0) return 0;
    hipLaunchKernelGGL(vox_stats_atomic_kernel, dim3(ceil_div(N, 256), B), dim3(256), 0, s, coords, N, R, S, ind, cnt);
    if (int e = check_launch("vox_stats_atomic")) return e;
    if (C == 0) return 0;
    const int CT = 16;
    hipLaunchKernelGGL(vox_scatter_atomic_kernel, dim3(ceil_div(N, 256), ceil_div(C, CT), B), dim3(256), 0, s,
                       feat, ind, cnt, C, N, S, CT, out);
    return check_launch("vox_scatter_atomic");
  }

  PVCNN_REQUIRE(workspace && workspace_bytes >= pvcnn_avg_voxelize_fwd_workspace_bytes(B, N, R),
                "workspace too small (see pvcnn_avg_voxelize_fwd_workspace_bytes)");
  PVCNN_REQUIRE(aligned16(workspace), "workspace must be 16-byte aligned");
  char *ws = static_cast<char *>(workspace);
  int32_t *start = reinterpret_cast<int32_t *>(ws);
  ws += align16((size_t)B * S * 4);
  int32_t *order_tmp = reinterpret_cast<int32_t *>(ws);
  ws += align16((size_t)B * N * 4);
  int32_t *order = reinterpret_cast<int32_t *>(ws);

  const size_t lds = ((size_t)(S + (S >> 5)) + 1 + 32) * sizeof(int);
  if (int e = enable_big_lds(vox_prep_kernel, lds)) { set_error("vox_prep: LDS attribute: %d", e); return e; }
  hipLaunchKernelGGL(vox_prep_kernel, dim3(B), dim3(kPrepThreads), lds, s, coords, N, R, S, ind, cnt, start,
                     order_tmp, order);
  if (int e = check_launch("vox_prep")) return e;
  if (C == 0) return 0;

  const bool vec = (S % 4 == 0) && aligned16(out) && aligned16(cnt) && aligned16(start);
  // channel tile: keep the grid >= ~8 workgroups per CU, amortise the cnt/start/order reads
  int CT = C;
  const int vox_blocks = ceil_div(S, 256 * (vec ? 4 : 1));
  while (CT > 4 && (long)vox_blocks * ceil_div(C, CT) * B < 8L * kNumCU) CT = (CT + 1) / 2;
  const dim3 grid(vox_blocks, ceil_div(C, CT), B);
  if (vec)
    hipLaunchKernelGGL(vox_gather_kernel<4>, grid, dim3(256), 0, s, feat, cnt, start, order, C, N, S, CT, out);
  else
    hipLaunchKernelGGL(vox_gather_kernel<1>, grid, dim3(256), 0, s, feat, cnt, start, order, C, N, S, CT, out);
  return check_launch("vox_gather");
}

extern "C" int pvcnn_avg_voxelize_bwd(const float *grad_y, const int32_t *ind, const int32_t *cnt, int B, int C,
                                      int N, int S, float *grad_x, void *stream) {
  PVCNN_REQUIRE(B >= 0 && C >= 0 && N >= 0 && S > 0, "negative size");
  if (B == 0 || C == 0 || N == 0) return 0;
  PVCNN_REQUIRE(grad_y && ind && cnt && grad_x, "null pointer");
  PVCNN_REQUIRE(B <= 65535, "batch > 65535");
  VoxelMean p{ind, cnt, N, S};
  const bool vec = (N % 4 == 0) && aligned16(ind) && aligned16(grad_x);
  return launch_gather(p, grad_y, grad_x, B, C, /*L=*/S, /*J=*/N, vec, static_cast<hipStream_t>(stream),
                       "avg_voxelize_bwd");
}
