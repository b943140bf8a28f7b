/*
 * pvcnn_hip.h -- C ABI of libpvcnn_hip.so, the MI355X (gfx950) native backend of the PVConv
 * hot path.  It is what the reference's Python seam `modules/functional/backend.py:_backend`
 * (12 pybind functions, modules/functional/src/bindings.cpp:10-37) binds to on AMD hardware.
 *
 * Conventions (all entry points)
 *   - plain pointers and sizes only; no torch / ATen types cross this boundary;
 *   - every pointer is a DEVICE pointer on the current HIP device; tensors are dense,
 *     row-major, channel-major: features (B, C, N), coords (B, 3, N), voxel grids (B, C, R^3);
 *     data fp32, indices int32 (same as the reference, utils.hpp:7-18);
 *   - the library NEVER allocates or frees device memory and never synchronises the host:
 *     outputs and scratch are caller-owned, work is enqueued on `stream` (a hipStream_t,
 *     e.g. torch.cuda.current_stream().cuda_stream; NULL = the null stream);
 *   - every output buffer is FULLY written by the call -- the caller does not pre-zero
 *     anything (the reference's host code zero-fills with torch::zeros first);
 *   - return value: 0 on success; PVCNN_ERR_INVALID_ARGUMENT for a rejected argument;
 *     otherwise the (positive) hipError_t of the failed launch.  The reference instead
 *     prints and exit(-1)s (cuda_utils.cuh:28-37).  pvcnn_last_error_string() describes the
 *     calling thread's most recent failure;
 *   - re-entrant: no global mutable state except the thread-local error string.
 *
 * Numerics contract (what tests/ check against oracle/)
 *   bit-exact : EVERY output -- ind, cnt, inds, ball_query, FPS and 3-NN indices; devoxelize
 *               fwd outs/wgts, 3-NN weights/outputs, grouping/gather fwd; and also all
 *               scatter-adds (avg_voxelize fwd, devoxelize bwd, grouping/gather bwd, 3-NN bwd):
 *               they are evaluated without float atomics, in the serial point-index order
 *               the oracle defines, so results are run-to-run deterministic as well
 *               (the reference's atomicAdd order is undefined);
 *   <= 1e-5   : only the atomic fallback taken beyond 2^20 scatter targets per cloud (R > 101 grids).
 *   Scatter-type calls take caller-owned scratch: `workspace` must hold at least the matching
 *   *_workspace_bytes(...) bytes and be 16-byte aligned.
 */
#ifndef PVCNN_HIP_H_
#define PVCNN_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden; only these entry points are exported. */
#if defined(__GNUC__)
#define PVCNN_API __attribute__((visibility("default")))
#else
#define PVCNN_API
#endif

#define PVCNN_ABI_VERSION 12
#define PVCNN_OK 0
#define PVCNN_ERR_INVALID_ARGUMENT (-1)

/* ABI version of the loaded library (== PVCNN_ABI_VERSION it was built with). */
PVCNN_API int pvcnn_version(void);
/* Description of the calling thread's last error ("" if none).  Never NULL. */
PVCNN_API const char *pvcnn_last_error_string(void);

/* ---- coordinate pre-pass of modules/voxelization.py:16-25 (a dozen torch kernels in the reference) ------------
 * coords (B,3,N) float -> norm_coords (B,3,N) float in [0,R-1] (centred, optionally normalised to the unit cube,
 * scaled by R, clamped) and vox_coords (B,3,N) int32 = round-half-even(norm_coords).  normalize != 0:
 * c / (max ||c||_2 * 2 + eps) + 0.5, else (c + 1) / 2.  A degenerate cloud with eps = 0 gives NaN like the reference. */
PVCNN_API int pvcnn_voxel_coords(const float *coords, int B, int N, int R, int normalize, float eps, float *norm_coords,
                       int32_t *vox_coords, void *stream);

/* The same pre-pass with the two reductions left to the caller (DEFAULT path of pvcnn_amd.modules.Voxelization): the
 * reference computes mean = coords.mean(2) and radius = (coords - mean).norm(dim=1).max(dim=2) with torch ops; calling
 * those same ops on the same device and handing the results to this fused elementwise tail (centre, / (radius*2+eps),
 * + 0.5, * R, clamp, round-half-even -- each step separately rounded like the reference's separate kernels) gives
 * norm_coords / vox_coords BIT-IDENTICAL to modules/voxelization.py:16-25, whatever reduction tree the library uses.
 * coords: rows of a cloud contiguous, clouds coords_batch_stride floats apart (>= 3N: a channel slice of the input);
 * mean (B,3); radius (B) or NULL for normalize=False [(c + 1) / 2]. pvcnn_voxel_coords above (one launch, fp64 mean,
 * exact max: the correctly rounded statistics) remains as an opt-in; it may differ from torch in the last bit of the mean. */
PVCNN_API int pvcnn_voxel_coords_tail(const float *coords, long coords_batch_stride, const float *mean, const float *radius,
                            int B, int N, int R, float eps, float *norm_coords, int32_t *vox_coords, void *stream);

/* ---- avg_voxelize -------------------------------------------------------------------------
 * replaces avg_voxelize_forward  (voxelization/vox.cpp:17-43, kernels vox.cu:18-72)
 *          avg_voxelize_backward (voxelization/vox.cpp:54-76, kernel  vox.cu:86-110)
 * fwd: feat (B,C,N) f32, coords (B,3,N) i32 voxel coordinates in [0,R)
 *      -> out (B,C,R^3) f32, ind (B,N) i32, cnt (B,R^3) i32.
 *      out[b,c,v] = sum_{i: ind[b,i]=v} feat[b,c,i] * (1/cnt[b,v]), summed in ascending i
 *      (deterministic; no float atomics).  `workspace` is scratch of at least
 *      pvcnn_avg_voxelize_fwd_workspace_bytes(B,C,N,R) bytes, 16-byte aligned.
 *      Out-of-range coords are clamped into the grid (the reference has no bounds check).
 * bwd: grad_y (B,C,S) f32, ind (B,N), cnt (B,S) -> grad_x (B,C,N) f32.
 */
PVCNN_API size_t pvcnn_avg_voxelize_fwd_workspace_bytes(int B, int C, int N, int R);
PVCNN_API int pvcnn_avg_voxelize_fwd(const float *feat, const int32_t *coords, int B, int C, int N, int R,
                           float *out, int32_t *ind, int32_t *cnt, void *workspace,
                           size_t workspace_bytes, void *stream);
PVCNN_API int pvcnn_avg_voxelize_bwd(const float *grad_y, const int32_t *ind, const int32_t *cnt, int B,
                           int C, int N, int S, float *grad_x, void *stream);

/* ---- trilinear_devoxelize -----------------------------------------------------------------
 * replaces trilinear_devoxelize_forward  (interpolate/trilinear_devox.cpp:18-55, .cu:21-105)
 *          trilinear_devoxelize_backward (interpolate/trilinear_devox.cpp:67-91, .cu:119-162)
 * fwd: coords (B,3,N) f32 in [0,R-1], feat (B,C,R^3) -> outs (B,C,N); when is_training != 0
 *      also inds (B,8,N) i32 and wgts (B,8,N) f32 (corner order 000..111, z fastest);
 *      when is_training == 0 inds/wgts are not touched and may be NULL.
 * bwd: grad_y (B,C,N), inds, wgts -> grad_x (B,C,R^3) (every element written).
 */
PVCNN_API int pvcnn_trilinear_devox_fwd(const float *coords, const float *feat, int B, int C, int N, int R,
                              int is_training, int32_t *inds, float *wgts, float *outs,
                              void *stream);
PVCNN_API size_t pvcnn_trilinear_devox_bwd_workspace_bytes(int B, int C, int N, int R);
PVCNN_API int pvcnn_trilinear_devox_bwd(const float *grad_y, const int32_t *inds, const float *wgts, int B,
                              int C, int N, int R, float *grad_x, void *workspace,
                              size_t workspace_bytes, void *stream);

/* ---- scatter plans: the counting sort of a scatter depends on its entries only -----------------------------------
 * Both scatters of the voxel branch are "plan, then apply".  The plan -- entries (source index, weight) grouped by
 * target voxel in the reference's serial order, their prefix offsets, and a count-sorted lane assignment -- depends on
 * (voxel coordinates, R) for avg_voxelize and on (inds, wgts) = (point coordinates, R) for the devoxelize backward, NOT
 * on the features, their channel count or the layer: every PVConv that shares coords and R (PVCNN: three at R = 16)
 * applies one plan, forward and backward.  pvcnn_avg_voxelize_fwd / pvcnn_trilinear_devox_bwd above are plan + apply in
 * one call.  *_plan_bytes returns 0 when the grid is too large for a plan (R > 101): use the one-shot calls then.
 * plan: caller-owned, 16-byte aligned, *_plan_bytes(...) bytes, opaque; scratch: *_plan_scratch_bytes(...) bytes, free
 * to reuse once the plan call is enqueued on the same stream.  ind / cnt are avg_voxelize_forward's outputs.
 */
PVCNN_API size_t pvcnn_avg_voxelize_plan_bytes(int B, int N, int R);
PVCNN_API size_t pvcnn_avg_voxelize_plan_scratch_bytes(int B, int N, int R);
PVCNN_API int pvcnn_avg_voxelize_plan(const int32_t *coords, int B, int N, int R, int32_t *ind, int32_t *cnt, void *plan,
                            size_t plan_bytes, void *scratch, size_t scratch_bytes, void *stream);
/* (ABI v11) Both plans of one PVConv geometry in ONE chain of three launches: vox_plan as pvcnn_avg_voxelize_plan(vox_coords)
 * writes it (with ind / cnt), devox_bwd_plan as pvcnn_trilinear_devox_bwd_plan writes it from the (inds, wgts) that
 * pvcnn_trilinear_devox_fwd(norm_coords) emits -- the corner entries are derived from norm_coords (B,3,N) with the same expressions
 * (trilinear_devox.cu:41-75), so the plan exists before any layer has devoxelized.  The two sorts run side by side in each launch
 * (each alone is a latency chain that leaves most of the chip idle).  Plan sizes: the *_plan_bytes queries above;
 * scratch: pvcnn_pvconv_plans_scratch_bytes.  N > 0. */
PVCNN_API size_t pvcnn_pvconv_plans_scratch_bytes(int B, int N, int R);
PVCNN_API int pvcnn_pvconv_plans(const int32_t *vox_coords, const float *norm_coords, int B, int N, int R, int32_t *ind, int32_t *cnt,
                                 void *vox_plan, size_t vox_plan_bytes, void *devox_bwd_plan, size_t devox_bwd_plan_bytes,
                                 void *scratch, size_t scratch_bytes, void *stream);
PVCNN_API int pvcnn_avg_voxelize_apply(const float *feat, const void *plan, size_t plan_bytes, int B, int C, int N, int R,
                             float *out, void *stream);
PVCNN_API size_t pvcnn_trilinear_devox_bwd_plan_bytes(int B, int N, int R);
PVCNN_API size_t pvcnn_trilinear_devox_bwd_plan_scratch_bytes(int B, int N, int R);
PVCNN_API int pvcnn_trilinear_devox_bwd_plan(const int32_t *inds, const float *wgts, int B, int N, int R, void *plan,
                                   size_t plan_bytes, void *scratch, size_t scratch_bytes, void *stream);
PVCNN_API int pvcnn_trilinear_devox_bwd_apply(const float *grad_y, long grad_y_batch_stride, const void *plan, size_t plan_bytes,
                                    int B, int C, int N, int R, float *grad_x, void *stream);

/* ---- ball_query ---------------------------------------------------------------------------
 * replaces ball_query_forward (ball_query/ball_query.cpp:6-30, kernel ball_query.cu:19-50)
 * centers (B,3,M), points (B,3,N) -> out (B,M,U) i32: the first U points (ascending index)
 * with d^2 < radius^2 (radius^2 evaluated in float); short rows padded with the first hit,
 * rows without a hit are all 0.
 */
PVCNN_API int pvcnn_ball_query(const float *centers, const float *points, int B, int N, int M, float radius,
                     int U, int32_t *out, void *stream);

/* ---- grouping / gather --------------------------------------------------------------------
 * replaces grouping_forward/backward        (grouping/grouping.cpp:6-44, grouping.cu:18-77)
 *          gather_features_forward/backward (sampling/sampling.cpp:6-41, sampling.cu:17-66)
 * grouping fwd: features (B,C,N), indices (B,M,U) -> out (B,C,M,U)
 * grouping bwd: grad_y (B,C,M,U), indices -> grad_x (B,C,N)
 * gather   fwd: features (B,C,N), indices (B,M)   -> out (B,C,M)
 * gather   bwd: grad_y (B,C,M), indices -> grad_x (B,C,N)
 * Indices must lie in [0,N).
 */
PVCNN_API int pvcnn_grouping_fwd(const float *features, const int32_t *indices, int B, int C, int N, int M,
                       int U, float *out, void *stream);
PVCNN_API size_t pvcnn_grouping_bwd_workspace_bytes(int B, int C, int N, int M, int U);
PVCNN_API int pvcnn_grouping_bwd(const float *grad_y, const int32_t *indices, int B, int C, int N, int M,
                       int U, float *grad_x, void *workspace, size_t workspace_bytes, void *stream);
PVCNN_API int pvcnn_gather_fwd(const float *features, const int32_t *indices, int B, int C, int N, int M,
                     float *out, void *stream);
PVCNN_API size_t pvcnn_gather_bwd_workspace_bytes(int B, int C, int N, int M);
PVCNN_API int pvcnn_gather_bwd(const float *grad_y, const int32_t *indices, int B, int C, int N, int M,
                     float *grad_x, void *workspace, size_t workspace_bytes, void *stream);

/* ---- furthest point sampling --------------------------------------------------------------
 * replaces furthest_point_sampling_forward (sampling/sampling.cpp:43-58, sampling.cu:86-167)
 * coords (B,3,N) -> indices (B,M) i32, starting from point 0, with the reference's tie rule
 * (lowest (k mod 512, k) among equidistant candidates).  `distances` is optional scratch
 * (B,N) f32: NULL is allowed for N <= PVCNN_FPS_MAX_RESIDENT_POINTS; when given it receives
 * the final point-to-set distances (the reference's caller-visible scratch).
 */
#define PVCNN_FPS_MAX_RESIDENT_POINTS 16384
PVCNN_API int pvcnn_fps(const float *coords, int B, int N, int M, float *distances, int32_t *indices,
              void *stream);

/* ---- foreground selection of logits_mask (Frustum-PVCNN) --------------------------------------------------------
 * replaces the per-cloud Python loop of modules/functional/sampling.py:69-82 (nonzero() sync + numpy draws per cloud).
 * mask (B,N) bytes (non-zero = foreground) -> selected (B,M) int32 point ids:
 *   k = foreground count; k >= M: M distinct foreground points in random order; 0 < k < M: every one M/k times plus M%k
 *   distinct extra ones, shuffled; k == 0: all 0 -- the reference's three cases.
 * Randomness: `choices` (B,M) int32 = positions into the ascending foreground list, supplied by the caller (PARITY MODE:
 * with numpy's own draws the result is bit-identical to the reference), or, when choices is NULL, a Philox4x32-10 stream
 * keyed by `seed` (two int64 in DEVICE memory: key, stream id) -- no host round trip.  count (B) optional: receives k.
 * N, M <= 8192.
 */
PVCNN_API int pvcnn_mask_select(const uint8_t *mask, int B, int N, int M, const int32_t *choices, const int64_t *seed,
                      int32_t *selected, int32_t *count, void *stream);

/* ---- 3-nearest-neighbour interpolation ----------------------------------------------------
 * replaces three_nearest_neighbors_interpolate_forward/backward
 *          (interpolate/neighbor_interpolate.cpp:6-65, neighbor_interpolate.cu:20-170)
 * fwd: points_coords (B,3,N), centers_coords (B,3,M), centers_features (B,C,M)
 *      -> indices (B,3,N) i32, weights (B,3,N) f32, out (B,C,N)
 * bwd: grad_y (B,C,N), indices, weights -> grad_x (B,C,M)
 */
PVCNN_API int pvcnn_three_nn_interp_fwd(const float *points_coords, const float *centers_coords,
                              const float *centers_features, int B, int C, int M, int N,
                              int32_t *indices, float *weights, float *out, void *stream);
PVCNN_API size_t pvcnn_three_nn_interp_bwd_workspace_bytes(int B, int C, int N, int M);
PVCNN_API int pvcnn_three_nn_interp_bwd(const float *grad_y, const int32_t *indices, const float *weights,
                              int B, int C, int N, int M, float *grad_x, void *workspace,
                              size_t workspace_bytes, void *stream);

/* ---- 3x3x3 voxel convolution (PVConv.voxel_layers) -------------------------------------------
 * replaces the nn.Conv3d(k=3, stride 1, padding 1) calls of modules/pvconv.py:20-27 (cuDNN in the
 * reference) with an fp32-MFMA implicit GEMM.  x (B,Ci,R,R,R), y (B,Co,R,R,R), channel-major.
 * weight_transform: w (Co,Ci,3,3,3) -> wt, the layout the kernels consume:
 *     for_bwd_data = 0 : wt (Ci,27,Co)                      [forward: y = conv(x, w) + bias]
 *     for_bwd_data = 1 : wt (Co,27,Ci), taps reversed       [grad_x = conv3d_fwd(grad_y, wt) with
 *                                                            Ci and Co exchanged, bias = NULL]
 * fp32 in / fp32 accumulate (v_mfma_f32_32x32x2_f32); results agree with an fp64 evaluation to
 * fp32 round-off (summation order differs from cuDNN / MIOpen / torch-CPU, as theirs do).
 */
PVCNN_API int pvcnn_conv3d_weight_transform(const float *w, int Co, int Ci, int for_bwd_data, float *wt,
                                            void *stream);
PVCNN_API int pvcnn_conv3d_fwd(const float *x, const float *wt, const float *bias, int B, int Ci, int Co,
                               int R, float *y, void *stream);
/* grad_w (Co,Ci,3,3,3) = sum over batch and voxels of grad_y (B,Co,R^3) x shifted x (B,Ci,R^3);
 * grad_bias (Co) = sum of grad_y over batch and voxels (optional: NULL to skip; it comes for free
 * from the grad_y tiles the kernel stages anyway).  Every element written.  `workspace`:
 * >= pvcnn_conv3d_bwd_weight_workspace_bytes(...) bytes of 16-byte aligned scratch (per-partition
 * partial sums; no float atomics). */
PVCNN_API size_t pvcnn_conv3d_bwd_weight_workspace_bytes(int B, int Ci, int Co, int R);
PVCNN_API int pvcnn_conv3d_bwd_weight(const float *x, const float *grad_y, int B, int Ci, int Co, int R,
                                      float *grad_w, float *grad_bias, void *workspace, size_t workspace_bytes,
                                      void *stream);

/* ---- the same convolution on the bf16 matrix cores (csrc/conv3d_bf16.hip) ---------------------------------------
 * nsplit = 1: bf16 operands, fp32 accumulate (BASELINE configs[4], "bf16 with MFMA 3D conv"; ~4e-3 relative).
 * nsplit = 3: "bf16x3" -- both fp32 operands split exactly into three bf16 pieces, the six significant partial products
 *             accumulated in fp32: fp32-class accuracy (<= 1e-5 vs fp64, like pvcnn_conv3d_fwd) at up to 16/6 = 2.7x
 *             the fp32-MFMA rate.
 * nsplit = 2: "f16x2" -- both fp32 operands scaled by a power of two (x: per workgroup tile, see "amax buffers" below; w: per
 *             output channel) and split into fp16 hi + lo (11 + 11 bits); hi*hi + hi*lo + lo*hi accumulated in fp32 and scaled
 *             back exactly: fp32-class accuracy (<= 1e-5 vs fp64) at 3 MFMAs per k-step.  Needs x_absmax = an amax buffer of the
 *             input.
 * amax buffers (f16x2 operand scales): 1 + T uint32 in device memory, T = B * ceil(L / seg) for x viewed as (B, C, L):
 *             [0]     = bit pattern of max |x| over the whole tensor -- the scale of the backward-weight kernels (they reduce
 *                       over positions, where small entries are negligible next to large ones in fp32 as well);
 *             [1 + t] = bit pattern of max |x| over ALL CHANNELS of position segment t = b * ceil(L / seg) + l / seg.
 *             The forward / backward-data kernels scale the tile of x a workgroup stages by the largest segment that tile
 *             (with its halo) touches: Conv3d seg = R (one z row of the grid, T = B*R*R), 1x1 GEMM seg = 256 points (its
 *             point tile).  RANGE CONTRACT: an output element is fp32-class (operands keep 22 bits) whenever the inputs it
 *             depends on are within 2^-17 of the largest magnitude of ITS OWN tile; smaller inputs keep fewer bits (absolute
 *             error <= 2^-38 of the tile maximum per product).  An outlier or a heavy-tailed gradient therefore costs
 *             precision only inside the tiles it occupies -- tests/test_gpu_range.py.  amax_seg = 0 selects the old single
 *             scale ([0] only; any 1-word buffer from pvcnn_absmax_bits will do).
 *             pvcnn_absmax_tiles computes a buffer in one read of x; pvcnn_bnact_fwd / pvcnn_bnact_bwd_strided emit the buffer
 *             of the tensor they write (no extra pass).
 * weight_split: w (Co,Ci,3,3,3) fp32 -> the kernels' pre-split, pre-swizzled LDS image (opaque, *_split_bytes bytes,
 *             16-byte aligned); for_bwd_data = 1 builds the flipped / channel-transposed image with which
 *             grad_x = conv3d_fwd_split(grad_y, wts, NULL, B, Ci = Co_fwd, Co = Ci_fwd, ...).
 * stats_part: NULL, or (Co, *_split_stats_parts) float pairs of per-workgroup (sum, sum of squares) of (y - bias).
 */
PVCNN_API size_t pvcnn_conv3d_weight_split_bytes(int Co, int Ci, int for_bwd_data, int nsplit);
PVCNN_API int pvcnn_conv3d_weight_split(const float *w, int Co, int Ci, int for_bwd_data, int nsplit, void *wts, void *stream);
/* both f16x2 images (for_bwd_data = 0 and 1) of one weight in ONE launch: a training step needs both, the weights do not change in between */
PVCNN_API int pvcnn_conv3d_weight_split_pair(const float *w, int Co, int Ci, void *wts_fwd, void *wts_bwd, void *stream);
PVCNN_API size_t pvcnn_conv3d_fwd_split_stats_parts(int B, int Co, int R, int nsplit);
/* (ABI v8) which launch shape pvcnn_conv3d_fwd_split takes for this problem: (voxels per workgroup tile) << 8 | weight rows per
 * tile (64, or 32 for the half-height tile of layers with Co <= 32 at R = 32).  Pure function of the sizes; for tests and tools
 * that have to prove a given tile ran (tests/test_gpu_parity_as_benched.py). */
PVCNN_API int pvcnn_conv3d_fwd_split_route(int B, int Ci, int Co, int R, int nsplit);
PVCNN_API int pvcnn_absmax_bits(const float *x, size_t n, void *out, void *stream);
PVCNN_API size_t pvcnn_absmax_tiles_count(int B, long L, int seg);      /* 1 + T words */
/* (ABI v11) `ticket`: NULL, or ONE zeroed 32-bit word in device memory that the call leaves zeroed.  With a ticket the global
 * maximum out[0] is written by the workgroup of the table pass that finishes last (release / acquire at device scope) instead of a
 * one-workgroup launch behind it (tables of <= 1024 words; longer ones keep the launch).  MEASURED (round 5, profiles/ab/r05c_*,
 * r05d_*): correct and bit-identical, but NOT faster on gfx950 -- the published atomics and tickets of every workgroup cross the
 * eight XCDs and cost what the ~5 us launches cost; pvcnn_amd passes NULL unless PVCNN_FOLD_FINALIZE=1.  The word must not be shared
 * with a launch that can run concurrently (another stream); launches on one stream may reuse it.  The same convention: `tickets` of
 * pvcnn_bnact_bwd_strided (C words: the per-channel sums are finalised by the workgroup that writes a channel's last partial),
 * `ticket` of pvcnn_concat_points. */
/* (ABI v12) ticket == PVCNN_TABLE_ONLY: the table out[1 ..] only -- word [0] is left UNWRITTEN and nothing is launched for it (~5 us
 * of launch latency, eleven times per PVCNN step).  For buffers whose every consumer takes the maximum from the table: the f16x2
 * convolutions / GEMMs with amax_seg > 0, and pvcnn_conv3d_bwd_weight_f16 / pvcnn_pwconv_bwd_weight_f16 with x_amax_seg / gy_amax_seg
 * > 0 (their workgroups reduce the table themselves: <= 64 KiB from L2).  The same value of `ticket` in pvcnn_concat_points. */
#define PVCNN_TABLE_ONLY ((void *)1)
PVCNN_API int pvcnn_absmax_tiles(const float *x, int B, int C, long L, int seg, void *out, void *ticket, void *stream);
PVCNN_API int pvcnn_conv3d_fwd_split(const float *x, const void *wts, const float *bias, int B, int Ci, int Co, int R, int nsplit,
                           const void *x_absmax, int amax_seg /* 0 | R */, float *y, float *stats_part, void *stream);

/* Backward-weight in the same f16x2 arithmetic (csrc/conv3d_wgrad_f16.hip), R = 8, 12, 16 or 32 (workspace_bytes returns 0 for
 * any other R: use pvcnn_conv3d_bwd_weight).  x_absmax / gy_absmax: pvcnn_absmax_bits of x and grad_y (word [0] of an amax
 * buffer).  (ABI v8) x_amax_seg = R says x_absmax IS an amax buffer with one maximum per z row (pvcnn_absmax_tiles(x, ..., seg = R),
 * or what pvcnn_bnact_fwd emitted): output rows whose nine neighbouring x rows are all zero -- the empty part of a voxelised
 * cloud -- skip their matrix work (exact: they would add zeros); x_amax_seg = 0: a 1-word buffer, nothing is skipped.
 * (ABI v12) gy_amax_seg = R: gy_absmax IS an amax buffer with one maximum per z row, too; with x_amax_seg / gy_amax_seg = R the
 * global maximum the kernel scales by is taken from the TABLE (word [0] is not read: its producer may have been asked for
 * PVCNN_TABLE_ONLY); 0: from word [0] as before.
 * Deterministic (split-K partials summed in a fixed order), <= 1e-5 vs fp64 like pvcnn_conv3d_bwd_weight. */
PVCNN_API size_t pvcnn_conv3d_bwd_weight_f16_workspace_bytes(int B, int Ci, int Co, int R);
PVCNN_API int pvcnn_conv3d_bwd_weight_f16(const float *x, const float *grad_y, const void *x_absmax, int x_amax_seg, const void *gy_absmax,
                                int gy_amax_seg, int B, int Ci, int Co, int R, float *grad_w, float *grad_bias, void *workspace,
                                size_t workspace_bytes, void *stream);

/* ---- 1x1 convolutions of SharedMLP (point branch, classifier) ------------------------------------
 * replaces the nn.Conv1d / nn.Conv2d (kernel 1) calls of modules/shared_mlp.py:9-25 (cuDNN / cuBLAS in the
 * reference) with fp32-MFMA GEMMs that work on the reference's channel-major layout directly:
 *     x (B,K,N), y (B,M,N);  N = points (Conv2d: N = M_centres * U_neighbours, flattened)
 * pwconv_fwd:  y[b,m,n] = sum_k wt[k,m] * x[b,k,n] + bias[m]   -- wt is the weight k-major, (wt_rows >= K, M):
 *     forward       : wt = pwconv_transpose(w (Co,Ci)): (Ci rounded up to 32, Co), zero tail rows; K = Ci, M = Co
 *     backward-data : wt = w (Co,Ci) as it is, wt_rows = Co; K = Co, M = Ci, bias = NULL, x = grad_y
 *   (rows K..wt_rows-1 must be zero; with wt_rows a multiple of 32 the unguarded fast path is available)
 * pwconv_bwd_weight: grad_w (M,K) = sum_{b,n} grad_y[b,m,n] * x[b,k,n]; grad_bias (M) optional (NULL).
 * fp32 in / fp32 accumulate; results agree with an fp64 evaluation to fp32 round-off.
 */
PVCNN_API int pvcnn_pwconv_transpose(const float *w, int M, int K, float *wt, void *stream);
PVCNN_API int pvcnn_pwconv_fwd(const float *x, const float *wt, int wt_rows, const float *bias, int B, int K, int M,
                     int N, float *y, void *stream);
PVCNN_API size_t pvcnn_pwconv_bwd_weight_workspace_bytes(int B, int K, int M, int N);
PVCNN_API int pvcnn_pwconv_bwd_weight(const float *x, const float *grad_y, int B, int K, int M, int N,
                            float *grad_w, float *grad_bias, void *workspace, size_t workspace_bytes,
                            void *stream);

/* ---- the same 1x1 GEMMs on the bf16 matrix cores (csrc/pointwise_bf16.hip): nsplit = 3 "bf16x3" (fp32-class accuracy, <= 1e-5
 * vs fp64) or nsplit = 1 (bf16 operands); see the Conv3d split entry points for the arithmetic.  weight_split: w (Co,Ci) fp32 -> the
 * kernel's pre-split, pre-swizzled image (opaque, *_split_bytes bytes, 16-byte aligned); for_bwd_data = 1 builds the transposed image
 * with which grad_x = pwconv_fwd_split(grad_y, wts, NULL, B, K = Co, M = Ci, ...). */
PVCNN_API size_t pvcnn_pwconv_weight_split_bytes(int Co, int Ci, int for_bwd_data, int nsplit);
PVCNN_API int pvcnn_pwconv_weight_split(const float *w, int Co, int Ci, int for_bwd_data, int nsplit, void *wts, void *stream);
PVCNN_API int pvcnn_pwconv_weight_split_pair(const float *w, int Co, int Ci, void *wts_fwd, void *wts_bwd, void *stream);
PVCNN_API size_t pvcnn_pwconv_fwd_split_stats_parts(int B, int N);
PVCNN_API int pvcnn_pwconv_fwd_split(const float *x, const void *wts, const float *bias, int B, int K, int M, int N, int nsplit,
                           const void *x_absmax, int amax_seg /* 0 | 256 */, float *y, float *stats_part, void *stream);
/* Backward-weight of the 1x1 convolution in f16x2 (csrc/pointwise_wgrad_f16.hip), N % 4 == 0 (workspace_bytes returns 0 otherwise:
 * use pvcnn_pwconv_bwd_weight).  x (B,K,N), grad_y (B,M,N) -> grad_w (M,K) [, grad_bias (M)]; *_absmax: pvcnn_absmax_bits of the two
 * tensors (word [0] of an amax buffer).  (ABI v12) x_amax_seg / gy_amax_seg > 0: the buffer is an amax buffer with segments of that
 * many points (pvcnn_absmax_tiles_count(B, N, seg) words) and the global maximum is taken from its TABLE, word [0] is not read (see
 * PVCNN_TABLE_ONLY); 0: from word [0].  Deterministic split-K, <= 1e-5 vs fp64. */
PVCNN_API size_t pvcnn_pwconv_bwd_weight_f16_workspace_bytes(int B, int K, int M, int N);
PVCNN_API int pvcnn_pwconv_bwd_weight_f16(const float *x, const float *grad_y, const void *x_absmax, int x_amax_seg, const void *gy_absmax,
                                int gy_amax_seg, int B, int K, int M, int N, float *grad_w, float *grad_bias, void *workspace,
                                size_t workspace_bytes, void *stream);

/* ---- BatchNorm fused with the following ReLU / LeakyReLU -----------------------------------------
 * replaces the (nn.BatchNorm{1,2,3}d, nn.ReLU | nn.LeakyReLU) module pairs of modules/pvconv.py:20-27
 * and modules/shared_mlp.py:20-25 (cuDNN BN + one more elementwise pass each way in the reference).
 * x, y, grad_* are (B, C, S) channel-major; statistics per channel over B*S.  slope = 0 for ReLU.
 * fwd, training != 0: computes mean / rstd (outputs, saved for backward), updates running_mean /
 *      running_var in place with `momentum` (unbiased variance; NULL = do not track), writes y.
 * fwd, training == 0: the caller supplies mean = running_mean and rstd = 1/sqrt(running_var+eps).
 * bwd: grad_x, grad_gamma, grad_beta (batch statistics differentiated through when training != 0).
 * gamma / beta may be NULL (affine = False).  `workspace`: >= pvcnn_bnact_workspace_bytes(B,C,S).
 * y_amax / gx_amax (NULL, or pvcnn_absmax_tiles_count(B, S, amax_seg) words): the amax buffer of the tensor the call writes (y
 *      resp. grad_x) with segments of amax_seg positions, emitted by the apply pass itself -- the f16x2 convolution that consumes
 *      that tensor needs no pass of its own over it.  amax_seg: ANY length in 1..256 on every entry that takes one (a z row of an
 *      R = 12 grid is a segment of 12; a workgroup then owns floor(256 / amax_seg) whole segments) -- the one exception is
 *      pvcnn_bnact_apply_rowmax, whose row-maximum butterfly needs all 256 lanes: amax_seg must DIVIDE 256 there and it says so.
 *      Word [0] (the global maximum) is accumulated with
 *      one atomic per workgroup when it was zeroed beforehand: by the call's own finalize kernel (training != 0, and always in
 *      bwd), or -- amax_zeroed != 0 -- by the pvcnn_bn_finalize call that produced mean / rstd (its zero_words argument, the WHOLE
 *      buffer: on small position counts the pass splits the channels over several workgroups whose table entries meet by atomic
 *      maxima); with training == 0 and amax_zeroed == 0 a one-workgroup reduction of the table is launched behind the pass instead.
 * tickets (pvcnn_bnact_bwd_strided, ABI v11; NULL, or C zeroed 32-bit words the call leaves zeroed -- see pvcnn_absmax_tiles): the
 *      per-channel sums grad_gamma / grad_beta are combined by the workgroup that writes a channel's last partial (the same fp64
 *      combine, the same bits) instead of a finalize launch between the two passes.
 * drop_seed / drop_p (pvcnn_bnact_fwd, pvcnn_bnact_bwd_strided; NULL / 0: off): the nn.Dropout(p) that follows the pair in the
 *      classifier heads (models/utils.py:15-36), fused: fwd writes y = keep ? act(bn(x)) / (1 - p) : 0 (and y's amax buffer of THAT
 *      tensor), bwd takes grad_y as the gradient of the dropped tensor.  keep(e) of element e is a pure function of (e, *drop_seed):
 *      16 bits of a 32-bit integer mixer, recomputed in every pass, never stored (P(keep) = 1 - round(p * 65536) / 65536).
 *      drop_seed: ONE int64 in device memory, drawn by the caller per forward call and handed to the matching backward call.
 *      Requires y_amax / gx_amax (the position-block-major passes) and B * C * S < 2^33.  pvcnn_dropout_keep_mask writes keep(e) for
 *      e = 0 .. numel - 1 as bytes (tests; not on the training path).
 */
PVCNN_API int pvcnn_dropout_keep_mask(const void *drop_seed, float drop_p, long numel, unsigned char *keep, void *stream);
PVCNN_API size_t pvcnn_bnact_workspace_bytes(int B, int C, int S);
PVCNN_API int pvcnn_bnact_fwd(const float *x, const float *gamma, const float *beta, float *running_mean,
                              float *running_var, int B, int C, int S, float eps, float momentum, float slope,
                              int training, float *mean, float *rstd, float *y, void *y_amax, int amax_seg, int amax_zeroed,
                              void *workspace, size_t workspace_bytes, const void *drop_seed, float drop_p, void *stream);
/* Batch statistics only (training): mean / rstd per channel + running-stat update; the first half of bnact_fwd.
 * Used with pvcnn_trilinear_devox_bnact_fwd, which applies BatchNorm + LeakyReLU while it stages the voxel
 * grid into LDS -- out = trilinear_devoxelize(leaky_relu(bn(feat))) without writing the activated grid
 * (PVConv: the last BatchNorm3d + LeakyReLU of voxel_layers followed by the devoxelization, modules/pvconv.py:
 * 25-27,36).  Bit-identical to bnact_fwd followed by trilinear_devox_fwd.  Requires R^3 * 4 bytes <= 160 KiB.
 * addend (B,C,N) or NULL: outs = devoxelized + addend in the gather's store -- PVConv's "voxel branch + point branch"
 * (modules/pvconv.py:38) without a separate read-modify-write pass; one rounded fp32 addition, as in the reference.
 * se_scale (B,C) or NULL: the squeeze-and-excitation factor of SE3d (modules/se.py:17) applied to the activated grid while it is
 * staged, out = trilinear_devoxelize(leaky_relu(bn(feat)) * se_scale[b][c]) -- a second rounded multiplication, as in the reference. */
/* Convolution forward WITH BatchNorm statistics: as pvcnn_conv3d_fwd / pvcnn_pwconv_fwd, and the epilogue also
 * writes per-workgroup partial (sum, sum of squares) of every output channel to stats_part -- (C, nparts) pairs of
 * floats, nparts = *_fwd_stats_parts(...) -- so the BatchNorm that follows needs no pass over y:
 * pvcnn_bn_finalize turns the partials into mean / rstd (fp64 combine) and updates the running statistics. */
PVCNN_API size_t pvcnn_conv3d_fwd_stats_parts(int B, int Co, int R);
PVCNN_API int pvcnn_conv3d_fwd_stats(const float *x, const float *wt, const float *bias, int B, int Ci, int Co, int R,
                           float *y, float *stats_part, void *stream);
PVCNN_API size_t pvcnn_pwconv_fwd_stats_parts(int B, int N);
PVCNN_API int pvcnn_pwconv_fwd_stats(const float *x, const float *wt, int wt_rows, const float *bias, int B, int K, int M,
                           int N, float *y, float *stats_part, void *stream);
/* The epilogue partials are sums of (y - bias) and (y - bias)^2: pass the convolution's bias as `shift` (NULL = no bias) and
 * bn_finalize adds it back to the mean -- a variance from E[a^2] - E[a]^2 stays accurate when the bias dwarfs the spread. */
PVCNN_API int pvcnn_bn_finalize(const float *part, int C, long nparts, double count, float eps, float momentum, const float *shift,
                      float *running_mean, float *running_var, float *mean, float *rstd,
                      void *zero_words /* NULL | the amax buffer the apply pass that follows will fill: */, long zero_count /* its words, set to 0 */,
                      void *num_batches_tracked /* NULL | one int64 incremented by 1 (nn.BatchNorm's counter) */, void *stream);
PVCNN_API int pvcnn_bn_stats(const float *x, float *running_mean, float *running_var, int B, int C, int S,
                   float eps, float momentum, float *mean, float *rstd, void *workspace,
                   size_t workspace_bytes, void *stream);
PVCNN_API int pvcnn_trilinear_devox_bnact_fwd(const float *coords, const float *feat, const float *gamma,
                                    const float *beta, const float *mean, const float *rstd, float slope,
                                    int B, int C, int N, int R, int is_training, int32_t *inds, float *wgts,
                                    const float *addend, const float *se_scale, float *outs, void *stream);
PVCNN_API int pvcnn_bnact_bwd(const float *x, const float *grad_y, const float *gamma, const float *beta,
                              const float *mean, const float *rstd, int B, int C, int S, float slope, int training,
                              float *grad_x, float *grad_gamma, float *grad_beta, void *workspace,
                              size_t workspace_bytes, void *stream);
/* As pvcnn_bnact_bwd / pvcnn_trilinear_devox_bwd, but grad_y may be a channel-slice VIEW of a wider tensor (what
 * the backward of torch.cat hands to each consumer): the C rows of one sample / cloud are contiguous, samples
 * are grad_y_batch_stride elements apart (>= C*S resp. C*N).  Saves the .contiguous() copy of every gradient. */
PVCNN_API int pvcnn_bnact_bwd_strided(const float *x, const float *grad_y, long grad_y_batch_stride, const float *gamma,
                            const float *beta, const float *mean, const float *rstd, int B, int C, int S,
                            float slope, int training, float *grad_x, float *grad_gamma, float *grad_beta,
                            void *gx_amax, int amax_seg, void *workspace, size_t workspace_bytes, const void *drop_seed,
                            float drop_p, void *tickets, void *stream);
/* The two halves of pvcnn_bnact_bwd_strided on their own, for callers that put something between them (PVConv's SE tail,
 * pvcnn_amd/modules/functional/bnact.py: the per-(cloud, channel) sums feed the excitation's backward before the apply pass runs).
 * partial_sums: part (C, B, slices) float pairs, slices = pvcnn_bnact_slices(S): per slice of row (b, c) the sums of g' and g' * xhat
 *   with g' = grad_y * act'(z), z = gamma * xhat + beta, xhat = (x - mean) * rstd; grad_y = NULL means grad_y == 1 (then the two sums
 *   are what SE3d's squeeze needs: sum act(z) = gamma * sum act'(z) xhat + beta * sum act'(z)).
 * bwd_apply: grad_x = gamma * rstd * (g' - sum_beta * inv_count - xhat * sum_gamma * inv_count) [training] or gamma * rstd * g' [eval] with
 *   g' = (grad_y * bc_mul[b][c] + bc_add[b][c]) * act'(z); bc_mul / bc_add (B,C) or NULL (1 / 0); sum_gamma / sum_beta (C) given by
 *   the caller; gx_amax / amax_seg as in pvcnn_bnact_bwd_strided. */
PVCNN_API int pvcnn_bnact_slices(int S);
/* Batched refresh of the f16x2 weight images (pvcnn_conv3d_weight_split_pair / pvcnn_pwconv_weight_split_pair of MANY weights in one
 * launch each): a training step changes every weight once (the optimizer) and needs both images of every layer afterwards -- 13
 * launches per PVCNN step, 56 per PVCNN++ step as per-layer calls.
 * _entry: fills `entry` (HOST memory, 10 int64 words) for one weight and its two image buffers (sized by *_weight_split_bytes(.., 0 / 1,
 *      2)) and returns the number of workgroups it takes (-1: bad argument).  The caller sets entry[9] to the running sum of the counts
 *      of the entries before it, and copies the n x 10 words to device memory.
 * _batch: table = those words on the device, total_rows = the sum of the counts.  Writes exactly what the per-weight calls write. */
PVCNN_API long pvcnn_conv3d_weight_split_pair_entry(const float *w, int Co, int Ci, void *wts_fwd, void *wts_bwd, long long *entry);
PVCNN_API int pvcnn_conv3d_weight_split_pair_batch(const void *table, int n, long total_rows, void *stream);
PVCNN_API long pvcnn_pwconv_weight_split_pair_entry(const float *w, int Co, int Ci, void *wts_fwd, void *wts_bwd, long long *entry);
PVCNN_API int pvcnn_pwconv_weight_split_pair_batch(const void *table, int n, long total_rows, void *stream);
/* (ABI v10) The same for the plain-bf16 images (nsplit = 1: torch.autocast; buffers sized by *_weight_split_bytes(.., 0 / 1, 1)): a
 * Frustum-PVCNN step issued 41 per-layer launches for them. */
PVCNN_API long pvcnn_conv3d_weight_split_pair_entry_bf16(const float *w, int Co, int Ci, void *wts_fwd, void *wts_bwd, long long *entry);
PVCNN_API int pvcnn_conv3d_weight_split_pair_batch_bf16(const void *table, int n, long total_rows, void *stream);
PVCNN_API long pvcnn_pwconv_weight_split_pair_entry_bf16(const float *w, int Co, int Ci, void *wts_fwd, void *wts_bwd, long long *entry);
PVCNN_API int pvcnn_pwconv_weight_split_pair_batch_bf16(const void *table, int n, long total_rows, void *stream);

/* The max over the K neighbours of a centre (modules/pointnet.py:85: `mlp(grouper(...)).max(dim=-1).values` on (B, C, M, K)) and its
 * backward, one streaming pass each.  x: (rows, K) contiguous, 16-byte aligned, rows = B * C * M; K in {4, 8, 16, 32, 64}
 * (pvcnn_neighbor_max_supported; other K: the caller keeps torch's reduction).  out (rows) = the maxima, winners (rows) = their k
 * (ties: the smallest k; a NaN wins against numbers, like torch.max).  bwd: grad_x (rows, K) = grad_out at the winner, 0 elsewhere. */
PVCNN_API int pvcnn_neighbor_max_supported(int K);
PVCNN_API int pvcnn_neighbor_max_fwd(const float *x, long rows, int K, float *out, unsigned char *winners, void *stream);
PVCNN_API int pvcnn_neighbor_max_bwd(const float *grad_out, const unsigned char *winners, long rows, int K, float *grad_x, void *stream);

/* arg-max over long rows: the global max-pool over the points of a cloud (models/s3dis/pvcnn.py:41-43, `features.max(dim=-1)` on
 * (B, C, N)).  x: (rows, K) contiguous, 16-byte aligned, K % 4 == 0; winners (rows) int64 = the first index of the row maximum (a NaN
 * wins against numbers), values (rows) = the maxima or NULL. */
PVCNN_API int pvcnn_row_argmax(const float *x, long rows, int K, long long *winners, float *values, void *stream);

/* (ABI v9) The same arg-max WITHOUT a read of its own: the BatchNorm + activation pass that writes the tensor the max-pool reduces
 * (the last SharedMLP of models/s3dis/pvcnn.py:37-43) emits, per (sample, channel) row, a 64-bit key
 *     (orderable bits of the maximum << 32) | ~(first position of that maximum)
 * into row_keys[b * C + c] by atomic maxima, and pvcnn_row_keys_decode turns the keys into int64 winners (the indices torch.max
 * returns: first index on ties, -0 == +0, a NaN wins) and, unless NULL, the values (read back from y: bit-identical elements).
 * pvcnn_bnact_apply_rowmax is the apply pass of pvcnn_bnact_fwd alone: mean / rstd from pvcnn_bn_finalize, whose zero_words
 * argument must have zeroed BOTH y_amax (pvcnn_absmax_tiles_count(B, S, amax_seg) words) and row_keys (B * C uint64, 8-byte aligned)
 * -- e.g. one buffer holding the two.  S % 256 == 0, amax_seg a multiple of 4 that DIVIDES 256 (4, 8, 16, 32, 64, 128, 256: every lane
 * of a workgroup's 256 positions takes part in the row butterfly), x / y 16-byte aligned. */
/* (ABI v11) y_batch_stride (0 = C * S): the samples of y may be that many elements apart -- the pass writes its output INTO a channel
 * slice of a wider (B, C_total, S) tensor, the concatenation in front of the classifier (models/s3dis/pvcnn.py:45), so that
 * pvcnn_concat_points has nothing to copy for the widest of its sources (`src_amax` there).  pvcnn_row_keys_decode reads the values
 * back from such a y with C and the same stride (C <= 0: a contiguous (rows, S) tensor). */
PVCNN_API int pvcnn_bnact_apply_rowmax(const float *x, const float *gamma, const float *beta, const float *mean, const float *rstd, int B,
                                       int C, int S, float slope, float *y, long y_batch_stride, void *y_amax, int amax_seg,
                                       void *row_keys, void *stream);
PVCNN_API int pvcnn_row_keys_decode(const void *row_keys, const float *y, long rows, int S, long long *winners, float *values,
                                    int C, long y_batch_stride, void *stream);

/* (ABI v10) The box part of Frustum-PointNet's multi-task loss (modules/frustum.py:43-124: FrustumPointNetLoss without the
 * foreground-mask cross entropy) AND its gradient, one launch:
 *   box = huber(|c_t - c|, 2) + huber(|c_t - c_reg|, 1) + CE(heading_scores, h) + CE(size_scores, s)
 *       + w_heading_residual * huber(hrn[h] - h_res_t / heading_bin_width, 1) + w_size_residual * huber(|s_res_t / T[s] - srn[s]|, 1)
 *       + w_corners * huber(min(|corners - corners_t|, |corners - corners_t turned by pi|), 1),   every term a mean over the batch.
 * Inputs (fp32, contiguous): center, center_reg, center_t (B,3); heading_scores, heading_residuals_normalized, heading_residuals
 * (B,NH); size_scores (B,NS); size_residuals_normalized, size_residuals (B,NS,3); heading_bin_id, size_template_id (B) int64;
 * heading_residual_t (B); size_residual_t (B,3); size_templates (NS,3); heading_bin_centers (NH); heading_bin_width = pi / NH in
 * the reference.  Outputs: loss[0] = box; grads (pvcnn_frustum_box_loss_grad_floats(B, NH, NS) floats) = d box / d of the eight network
 * outputs, concatenated: [center 3B | center_reg 3B | heading_scores B*NH | size_scores B*NS | heading_residuals_normalized B*NH |
 * size_residuals_normalized B*NS*3 | heading_residuals B*NH | size_residuals B*NS*3], every element written.  Sub-gradients as
 * autograd takes them (0 at |x| = 0 and at a zero-length vector; an exact tie of the two corner distances splits evenly). */
PVCNN_API size_t pvcnn_frustum_box_loss_grad_floats(int B, int NH, int NS);
PVCNN_API int pvcnn_frustum_box_loss(const float *center, const float *center_reg, const float *heading_scores, const float *size_scores,
                                     const float *heading_residuals_normalized, const float *size_residuals_normalized,
                                     const float *heading_residuals, const float *size_residuals, const long long *heading_bin_id,
                                     const long long *size_template_id, const float *heading_residual_t, const float *size_residual_t,
                                     const float *center_t, const float *size_templates, const float *heading_bin_centers, int B, int NH,
                                     int NS, float heading_bin_width, float w_heading_residual, float w_size_residual, float w_corners,
                                     float *loss, float *grads, void *stream);

/* The excitation of SE3d (modules/se.py:6-17: Linear(C, H, bias=False) + ReLU + Linear(H, C, bias=False) + Sigmoid on the squeezed
 * (B, C) descriptor) between the two reduction passes of PVConv's fused squeeze-and-excitation tail, and its backward.
 * part: (C, B, slices, 2) as pvcnn_bnact_partial_sums writes it (slices = pvcnn_bnact_slices(S)); the sums over the slices are taken here.
 * fwd: part from grad_y == NULL -> a_sum / ax_sum (B, C) = the sums of act'(z) and act'(z) * xhat over the grid (outputs, kept for the
 *      backward); squeezed = (gamma * ax_sum + beta * a_sum) * inv_s, hidden = relu(squeezed W1^T) (B, H), excite = sigmoid(hidden W2^T) (B, C).
 * bwd: part from grad_y = g_y -> P / Q = the sums of g_y act'(z) and g_y act'(z) xhat; -> g_w1 (H, C), g_w2 (C, H), g_mean (B, C) =
 *      dL/dsqueezed * inv_s, and the BatchNorm backward's per-channel sums sum_beta / sum_gamma (C) of g' = (excite g_y + g_mean) act'(z).
 *      workspace: B * (3 C + H) floats.  C <= 2048, H <= 256.  Deterministic (sums over slices and clouds in index order). */
PVCNN_API int pvcnn_se_excite_fwd(const float *part, int slices, const float *gamma, const float *beta, const float *w1, const float *w2,
                        int B, int C, int H, float inv_s, float *a_sum, float *ax_sum, float *squeezed, float *hidden, float *excite,
                        void *stream);
PVCNN_API int pvcnn_se_excite_bwd(const float *part, int slices, const float *a_sum, const float *ax_sum, const float *gamma,
                        const float *beta, const float *squeezed, const float *hidden, const float *excite, const float *w1,
                        const float *w2, int B, int C, int H, float inv_s, float *g_w1, float *g_w2, float *g_mean, float *sum_beta,
                        float *sum_gamma, float *workspace, void *stream);
PVCNN_API int pvcnn_bnact_partial_sums(const float *x, const float *grad_y, long grad_y_batch_stride, const float *gamma, const float *beta,
                             const float *mean, const float *rstd, int B, int C, int S, float slope, float *part, void *stream);
PVCNN_API int pvcnn_bnact_bwd_apply(const float *x, const float *grad_y, long grad_y_batch_stride, const float *gamma, const float *beta,
                          const float *mean, const float *rstd, const float *sum_gamma, const float *sum_beta, const float *bc_mul,
                          const float *bc_add, int B, int C, int S, float slope, int training, float *grad_x, void *gx_amax,
                          int amax_seg, void *stream);
PVCNN_API int pvcnn_trilinear_devox_bwd_strided(const float *grad_y, long grad_y_batch_stride, const int32_t *inds,
                                      const float *wgts, int B, int C, int N, int R, float *grad_x,
                                      void *workspace, size_t workspace_bytes, void *stream);

/* ---- concatenation of per-point feature maps along the channels: torch.cat(features, dim=1) of models/s3dis/pvcnn.py:45 (and
 * models/shapenet/pvcnn.py:41) in one pass that also emits the amax buffer (256-point segments) of its output, i.e. the f16x2 scale
 * table of the classifier GEMM that consumes it.  nsrc <= 8 sources; source i has channels[i] channels, its clouds are bstrides[i]
 * elements apart and pstrides[i] is 1 (rows of N points) or 0 (one value per (cloud, channel), broadcast over the points). */
PVCNN_API int pvcnn_concat_points(const float *const *srcs, const long *bstrides, const int *channels, const int *pstrides,
                        const void *const *src_amax /* (ABI v11) NULL, or per source: NULL = copy it; else the source ALREADY IS its channel
                        slice of out (srcs[i] == out + c0_i * N, bstrides[i] == C_total * N: the pass that produced it wrote it there) and
                        this is its amax buffer (256-point segments), merged into out_amax; nothing is copied for it */,
                        int nsrc, int B, int N, float *out, void *out_amax,
                        void *ticket /* (ABI v11) see pvcnn_absmax_tiles; NULL: a reduce launch */, void *stream);

/* ---- (ABI v11) Linear + BatchNorm1d + ReLU on a handful of rows: the `_linear_bn_relu` blocks of models/utils.py:11-12 -- the cloud
 * descriptor head of models/s3dis/pvcnn.py:22-25 ((B, 1024) -> 256 -> 128) and the dense heads of the Frustum nets -- where B is the
 * batch (16 ... 32 rows).  With so few rows the workgroup that owns an output channel owns that channel's whole batch, so the layer is
 * ONE launch forward (dot products, batch statistics, running statistics + num_batches_tracked, normalise, ReLU; torch modules: 5) and
 * ONE launch backward for everything but grad_x (ReLU mask, the BatchNorm backward, grad_z and the four parameter gradients; torch: 6
 * with the two GEMMs); grad_x = grad_z (rows, Cout) . weight (Cout, Cin) is left to the caller's GEMM.
 * x (rows, Cin), weight (Cout, Cin), z / y / grad_y / grad_z (rows, Cout) row-major fp32; bias / gamma / beta may be NULL;
 * running_mean / running_var NULL = not tracked; num_batches_tracked: NULL or one int64 in device memory (incremented).
 * z (the Linear's output) and mean / rstd are what backward needs.  Training mode only (batch statistics); 2 <= rows <= 64 and
 * rows * Cin * 4 bytes <= 144 KiB: ask pvcnn_dense_bn_relu_supported. */
PVCNN_API int pvcnn_dense_bn_relu_supported(int rows, int Cin, int Cout);
PVCNN_API int pvcnn_dense_bn_relu_fwd(const float *x, const float *weight, const float *bias, const float *gamma, const float *beta,
                                      float *running_mean, float *running_var, void *num_batches_tracked, int rows, int Cin, int Cout,
                                      float eps, float momentum, float *z, float *y, float *mean, float *rstd, void *stream);
PVCNN_API int pvcnn_dense_bn_relu_bwd(const float *x, const float *grad_y, const float *z, const float *mean, const float *rstd,
                                      const float *gamma, const float *beta, int rows, int Cin, int Cout, float *grad_z,
                                      float *grad_weight, float *grad_bias, float *grad_gamma, float *grad_beta, void *stream);

/* ---- the optimizer update of the training step on flat buffers (csrc/optim.hip) ----------------------------------------------
 * replaces torch.optim.Adam's per-tensor update of train.py:96-119 (optimizer.step()) when the parameters share the flat layout of
 * the gradient buckets (pvcnn_amd/dp.py): p, m (exp_avg), v (exp_avg_sq) updated in place, g read; arithmetic of torch.optim.Adam
 * (no amsgrad; weight_decay added to the gradient), fp32.  `step`: one float in device memory = updates done so far (nothing is
 * read back: graph-capturable); inc_step != 0 increments it behind this update (the last buffer of an optimizer step).
 * (ABI v8) `hyper`: five floats in DEVICE memory -- lr, beta1, beta2, eps, weight_decay -- read by the kernel, so that a learning-rate
 * schedule (train.py:121-122: scheduler.step()) reaches launches that were captured into a hipGraph. */
PVCNN_API int pvcnn_adam_step(float *p, const float *g, float *m, float *v, size_t n, float *step, const float *hyper, int inc_step,
                    void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PVCNN_HIP_H_ */
