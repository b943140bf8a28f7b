// ref_api.cpp -- extern "C" doorway to the reference's launcher functions (prototypes come from
// the reference's own .cuh headers, included in place).  Arguments and pre-zeroed outputs are
// exactly what the reference's *.cpp host code passes (oracle/ref_backend.py mirrors it).
#include "cuda_on_cpu.h"

#include "ball_query/ball_query.cuh"
#include "grouping/grouping.cuh"
#include "interpolate/neighbor_interpolate.cuh"
#include "interpolate/trilinear_devox.cuh"
#include "sampling/sampling.cuh"
#include "voxelization/vox.cuh"

#define REF_API extern "C" __attribute__((visibility("default")))

REF_API void ref_avg_voxelize(int b, int c, int n, int r, const int *coords, const float *feat, int *ind, int *cnt, float *out) {
  avg_voxelize(b, c, n, r, r * r, r * r * r, coords, feat, ind, cnt, out);
}
REF_API void ref_avg_voxelize_grad(int b, int c, int n, int s, const int *ind, const int *cnt, const float *grad_y, float *grad_x) {
  avg_voxelize_grad(b, c, n, s, ind, cnt, grad_y, grad_x);
}
REF_API void ref_trilinear_devoxelize(int b, int c, int n, int r, int training, const float *coords, const float *feat, int *inds, float *wgts, float *outs) {
  trilinear_devoxelize(b, c, n, r, r * r, r * r * r, training != 0, coords, feat, inds, wgts, outs);
}
REF_API void ref_trilinear_devoxelize_grad(int b, int c, int n, int r3, const int *inds, const float *wgts, const float *grad_y, float *grad_x) {
  trilinear_devoxelize_grad(b, c, n, r3, inds, wgts, grad_y, grad_x);
}
REF_API void ref_ball_query(int b, int n, int m, float r2, int u, const float *centers, const float *points, int *out) {
  ball_query(b, n, m, r2, u, centers, points, out);
}
REF_API void ref_grouping(int b, int c, int n, int m, int u, const float *features, const int *indices, float *out) {
  grouping(b, c, n, m, u, features, indices, out);
}
REF_API void ref_grouping_grad(int b, int c, int n, int m, int u, const float *grad_y, const int *indices, float *grad_x) {
  grouping_grad(b, c, n, m, u, grad_y, indices, grad_x);
}
REF_API void ref_gather_features(int b, int c, int n, int m, const float *features, const int *indices, float *out) {
  gather_features(b, c, n, m, features, indices, out);
}
REF_API void ref_gather_features_grad(int b, int c, int n, int m, const float *grad_y, const int *indices, float *grad_x) {
  gather_features_grad(b, c, n, m, grad_y, indices, grad_x);
}
// The reference's FPS kernel has a latent race: every thread reads `old = dists_i[0]` after the last
// barrier of step j (sampling.cu:165) and thread 0 overwrites dists_i[0] at step j+1 (:149-150) with
// no barrier in between.  On a GPU the warps read `old` long before thread 0 finishes its point loop;
// a sequential ascending schedule would let thread 0 clobber it first.  Descending order (thread 0
// last) is an equally legal schedule that reads before the overwrite, like the hardware does.
REF_API void ref_furthest_point_sampling(int b, int n, int m, const float *coords, float *distances, int *indices) {
  pvref::set_descending_schedule(true);
  furthest_point_sampling(b, n, m, coords, distances, indices);
  pvref::set_descending_schedule(false);
}
REF_API void ref_three_nn_interpolate(int b, int c, int m, int n, const float *points_coords, const float *centers_coords,
                                      const float *centers_features, int *indices, float *weights, float *out) {
  three_nearest_neighbors_interpolate(b, c, m, n, points_coords, centers_coords, centers_features, indices, weights, out);
}
REF_API void ref_three_nn_interpolate_grad(int b, int c, int n, int m, const float *grad_y, const int *indices, const float *weights, float *grad_x) {
  three_nearest_neighbors_interpolate_grad(b, c, n, m, grad_y, indices, weights, grad_x);
}
