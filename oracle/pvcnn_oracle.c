/*
 * pvcnn_oracle.c -- CPU restatement of the reference's native PVConv hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product (pvcnn_amd/) never imports,
 * links or executes anything in oracle/.
 *
 * Parity status: the reference (mit-han-lab/pvcnn) ships NO tests, golden vectors or CPU
 * implementation of these ops (every entry point is CHECK_CUDA, utils.hpp:7).  This file
 * is pinned by (a) the analytic known-answer tests in tests/test_oracle_kat.py and
 * (b) oracle/_ref -- the reference's own .cu kernels executed on the CPU through an
 * execution-model shim (oracle/build_ref.py) -- see tests/test_oracle_vs_ref.py.
 *
 * Restatement rules:
 *   - loops run in POINT-INDEX order: this defines the canonical fp32 summation order for
 *     every kernel the reference implements with fp32 atomicAdd (order undefined there);
 *   - floating-point contraction (nvcc's default -fmad=true) is pinned with explicit fmaf()
 *     following the rule LLVM/NVVM and GCC both apply to a left-associated sum of products:
 *     the FIRST addition fuses its LEFT product and rounds the right one,
 *         a*b + c*d        ->  fma(a, b, c*d)
 *     and every later "+ e*f" fuses too:  -> fma(e, f, acc).
 *     This was established against oracle/_ref (the reference's own .cu sources built with
 *     -ffp-contract=fast): the opposite association fma(c, d, a*b) does NOT reproduce it.
 *     This file must be compiled with -ffp-contract=off so nothing else is contracted;
 *   - all index tensors are int32, all data fp32, layout channel-major (B, C, N).
 *
 * Every function cites the reference file:line it follows (paths relative to
 * modules/functional/src/ of the reference tree).
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

ORC_API int orc_version(void) { return 1; }

/* One thread per cloud at most (mirrors the reference's one-block-per-cloud launch).  Without the
 * cap a 256-core host spins 256 OpenMP threads for a B=2 loop and starves torch's own thread pool. */
static inline int orc_threads(int b) {
  const int m = omp_get_max_threads();
  return b < 1 ? 1 : (b < m ? b : m);
}

/* ------------------------------------------------------------------------------------
 * avg_voxelize forward: voxelization/vox.cu:18-34 (grid_stats_kernel) and
 * vox.cu:48-72 (avg_voxelize_kernel).  Outputs must be zero on entry (vox.cpp:33-38).
 *   ind[b,i]  = x*r^2 + y*r + z                       (vox.cu:31, no bounds check)
 *   cnt[b,v]  = #{i : ind[b,i] == v}                  (vox.cu:32)
 *   out[b,c,v] += feat[b,c,i] * (1.0 / (float)cnt)   (vox.cu:66-68; each addend is
 *                 pre-multiplied; the divide is a double divide rounded to float)
 * ---------------------------------------------------------------------------------- */
ORC_API void orc_avg_voxelize_fwd(const float *feat, const int32_t *coords, int b, int c, int n,
                                  int r, float *out, int32_t *ind, int32_t *cnt) {
  const int r2 = r * r, s = r2 * r;
#pragma omp parallel for schedule(static) num_threads(orc_threads(b))
  for (int bi = 0; bi < b; ++bi) {
    const int32_t *co = coords + (size_t)bi * n * 3;
    int32_t *in_ = ind + (size_t)bi * n;
    int32_t *cn = cnt + (size_t)bi * s;
    const float *f = feat + (size_t)bi * c * n;
    float *o = out + (size_t)bi * c * s;
    for (int i = 0; i < n; ++i) {
      in_[i] = co[i] * r2 + co[i + n] * r + co[i + n + n];
      cn[in_[i]] += 1;
    }
    for (int i = 0; i < n; ++i) {
      const int pos = in_[i];
      const int cur = cn[pos];
      if (cur > 0) {
        const float div_cur_cnt = (float)(1.0 / (double)(float)cur);
        for (int j = 0; j < c; ++j) {
          const float addend = f[(size_t)j * n + i] * div_cur_cnt;
          o[(size_t)j * s + pos] = o[(size_t)j * s + pos] + addend;
        }
      }
    }
  }
}

/* fp64-accumulated variant: the "true" value both fp32 implementations are measured against. */
ORC_API void orc_avg_voxelize_fwd_f64(const float *feat, const int32_t *ind, const int32_t *cnt,
                                      int b, int c, int n, int s, double *out) {
#pragma omp parallel for schedule(static) num_threads(orc_threads(b))
  for (int bi = 0; bi < b; ++bi) {
    for (int i = 0; i < n; ++i) {
      const int pos = ind[(size_t)bi * n + i];
      const int cur = cnt[(size_t)bi * s + pos];
      if (cur > 0)
        for (int j = 0; j < c; ++j)
          out[((size_t)bi * c + j) * s + pos] += (double)feat[((size_t)bi * c + j) * n + i] / cur;
    }
  }
}

/* avg_voxelize backward: vox.cu:86-110.  grad_x zero on entry (vox.cpp:71-72); every
 * address is hit exactly once, so the atomicAdd is a plain store of one product. */
ORC_API void orc_avg_voxelize_bwd(const float *grad_y, const int32_t *ind, const int32_t *cnt,
                                  int b, int c, int n, int s, float *grad_x) {
#pragma omp parallel for schedule(static) num_threads(orc_threads(b))
  for (int bi = 0; bi < b; ++bi) {
    const int32_t *in_ = ind + (size_t)bi * n;
    const int32_t *cn = cnt + (size_t)bi * s;
    const float *gy = grad_y + (size_t)bi * c * s;
    float *gx = grad_x + (size_t)bi * c * n;
    for (int i = 0; i < n; ++i) {
      const int pos = in_[i];
      const int cur = cn[pos];
      if (cur > 0) {
        const float div_cur_cnt = (float)(1.0 / (double)(float)cur);
        for (int j = 0; j < c; ++j)
          gx[(size_t)j * n + i] = gx[(size_t)j * n + i] + gy[(size_t)j * s + pos] * div_cur_cnt;
      }
    }
  }
}

/* ------------------------------------------------------------------------------------
 * trilinear_devoxelize forward: interpolate/trilinear_devox.cu:21-105.
 * Corner order 000,001,010,011,100,101,110,111 (bits x,y,z; z fastest).  Hi offsets are
 * applied only where the fractional part is > 0 (:64-75), so p in [0, r-1] never reads
 * out of bounds.  The 8-term sum (:98-102) is evaluated left to right and contracted to
 * fma(w0,f0, w1*f1) followed by 6 more fma (see the header).  inds/wgts only in training.
 * ---------------------------------------------------------------------------------- */
static inline void trilinear_setup(float x, float y, float z, int r, int r2, int32_t idx[8],
                                   float w[8]) {
  const float x_lo_f = floorf(x), y_lo_f = floorf(y), z_lo_f = floorf(z);
  const float x_d_1 = x - x_lo_f, y_d_1 = y - y_lo_f, z_d_1 = z - z_lo_f;
  const float x_d_0 = 1.0f - x_d_1, y_d_0 = 1.0f - y_d_1, z_d_0 = 1.0f - z_d_1;
  w[0] = x_d_0 * y_d_0 * z_d_0;
  w[1] = x_d_0 * y_d_0 * z_d_1;
  w[2] = x_d_0 * y_d_1 * z_d_0;
  w[3] = x_d_0 * y_d_1 * z_d_1;
  w[4] = x_d_1 * y_d_0 * z_d_0;
  w[5] = x_d_1 * y_d_0 * z_d_1;
  w[6] = x_d_1 * y_d_1 * z_d_0;
  w[7] = x_d_1 * y_d_1 * z_d_1;
  const int x_lo = (int)x_lo_f, y_lo = (int)y_lo_f, z_lo = (int)z_lo_f;
  const int x_hi = (x_d_1 > 0) ? -1 : 0;
  const int y_hi = (y_d_1 > 0) ? -1 : 0;
  const int z_hi = (z_d_1 > 0) ? 1 : 0;
  idx[0] = x_lo * r2 + y_lo * r + z_lo;
  idx[1] = idx[0] + z_hi;
  idx[2] = idx[0] + (y_hi & r);
  idx[3] = idx[2] + z_hi;
  idx[4] = idx[0] + (x_hi & r2);
  idx[5] = idx[4] + z_hi;
  idx[6] = idx[4] + (y_hi & r);
  idx[7] = idx[6] + z_hi;
}

ORC_API void orc_trilinear_devox_fwd(const float *coords, const float *feat, int b, int c, int n,
                                     int r, int is_training, int32_t *inds, float *wgts,
                                     float *outs) {
  const int r2 = r * r, r3 = r2 * r;
#pragma omp parallel for schedule(static) num_threads(orc_threads(b))
  for (int bi = 0; bi < b; ++bi) {
    const float *co = coords + (size_t)bi * n * 3;
    const float *f = feat + (size_t)bi * c * r3;
    float *o = outs + (size_t)bi * c * n;
    for (int i = 0; i < n; ++i) {
      int32_t idx[8];
      float w[8];
      trilinear_setup(co[i], co[i + n], co[i + n + n], r, r2, idx, w);
      if (is_training) {
        for (int k = 0; k < 8; ++k) {
          wgts[((size_t)bi * 8 + k) * n + i] = w[k];
          inds[((size_t)bi * 8 + k) * n + i] = idx[k];
        }
      }
      for (int j = 0; j < c; ++j) {
        const float *fj = f + (size_t)j * r3;
        float acc = fmaf(w[0], fj[idx[0]], w[1] * fj[idx[1]]);
        for (int k = 2; k < 8; ++k) acc = fmaf(w[k], fj[idx[k]], acc);
        o[(size_t)j * n + i] = acc;
      }
    }
  }
}

ORC_API void orc_trilinear_devox_fwd_f64(const float *coords, const float *feat, int b, int c,
                                         int n, int r, double *outs) {
  const int r2 = r * r, r3 = r2 * r;
#pragma omp parallel for schedule(static) num_threads(orc_threads(b))
  for (int bi = 0; bi < b; ++bi) {
    const float *co = coords + (size_t)bi * n * 3;
    for (int i = 0; i < n; ++i) {
      int32_t idx[8];
      float w[8];
      trilinear_setup(co[i], co[i + n], co[i + n + n], r, r2, idx, w);
      for (int j = 0; j < c; ++j) {
        const float *fj = feat + ((size_t)bi * c + j) * r3;
        double acc = 0.0;
        for (int k = 0; k < 8; ++k) acc += (double)w[k] * (double)fj[idx[k]];
        outs[((size_t)bi * c + j) * n + i] = acc;
      }
    }
  }
}

/* trilinear_devoxelize backward: trilinear_devox.cu:119-162.  grad_x zero on entry
 * (trilinear_devox.cpp:85-86).  Canonical order: point i ascending, channel j, corner k. */
ORC_API void orc_trilinear_devox_bwd(const float *grad_y, const int32_t *inds, const float *wgts,
                                     int b, int c, int n, int r3, float *grad_x) {
#pragma omp parallel for schedule(static) num_threads(orc_threads(b))
  for (int bi = 0; bi < b; ++bi) {
    const int32_t *id = inds + (size_t)bi * n * 8;
    const float *wg = wgts + (size_t)bi * n * 8;
    const float *gy = grad_y + (size_t)bi * c * n;
    float *gx = grad_x + (size_t)bi * c * r3;
    for (int i = 0; i < n; ++i) {
      for (int j = 0; j < c; ++j) {
        const float g = gy[(size_t)j * n + i];
        float *gxj = gx + (size_t)j * r3;
        for (int k = 0; k < 8; ++k) {
          const int p = id[(size_t)k * n + i];
          gxj[p] = gxj[p] + wg[(size_t)k * n + i] * g;
        }
      }
    }
  }
}

ORC_API void orc_trilinear_devox_bwd_f64(const float *grad_y, const int32_t *inds,
                                         const float *wgts, int b, int c, int n, int r3,
                                         double *grad_x) {
#pragma omp parallel for schedule(static) num_threads(orc_threads(b))
  for (int bi = 0; bi < b; ++bi)
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < c; ++j) {
        const double g = grad_y[((size_t)bi * c + j) * n + i];
        for (int k = 0; k < 8; ++k) {
          const int p = inds[((size_t)bi * 8 + k) * n + i];
          grad_x[((size_t)bi * c + j) * r3 + p] += (double)wgts[((size_t)bi * 8 + k) * n + i] * g;
        }
      }
}

/* ------------------------------------------------------------------------------------
 * ball_query: ball_query/ball_query.cu:19-50; r2 = radius*radius in float on the host
 * (ball_query.cpp:24).  Output zero on entry (ball_query.cpp:20-22).  d2 is contracted:
 * fma(dz,dz, fma(dx,dx, dy*dy)).  Strict '<'.  First hit fills all u slots.
 * ---------------------------------------------------------------------------------- */
ORC_API void orc_ball_query(const float *centers, const float *points, int b, int n, int m,
                            float r2, int u, int32_t *out) {
#pragma omp parallel for schedule(static) num_threads(orc_threads(b))
  for (int bi = 0; bi < b; ++bi) {
    const float *pc = points + (size_t)bi * n * 3;
    const float *cc = centers + (size_t)bi * m * 3;
    int32_t *ni = out + (size_t)bi * m * u;
    for (int j = 0; j < m; ++j) {
      const float cx = cc[j], cy = cc[j + m], cz = cc[j + m + m];
      for (int k = 0, cnt = 0; k < n && cnt < u; ++k) {
        const float dx = cx - pc[k], dy = cy - pc[k + n], dz = cz - pc[k + n + n];
        const float d2 = fmaf(dz, dz, fmaf(dx, dx, dy * dy));
        if (d2 < r2) {
          if (cnt == 0)
            for (int v = 0; v < u; ++v) ni[(size_t)j * u + v] = k;
          ni[(size_t)j * u + cnt] = k;
          ++cnt;
        }
      }
    }
  }
}

/* grouping forward/backward: grouping/grouping.cu:18-36, 58-77. */
ORC_API void orc_grouping_fwd(const float *features, const int32_t *indices, int b, int c, int n,
                              int m, int u, float *out) {
#pragma omp parallel for schedule(static) num_threads(orc_threads(b))
  for (int bi = 0; bi < b; ++bi) {
    const float *f = features + (size_t)bi * n * c;
    const int32_t *id = indices + (size_t)bi * m * u;
    float *o = out + (size_t)bi * m * u * c;
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < m; ++j)
        for (int k = 0; k < u; ++k)
          o[((size_t)l * m + j) * u + k] = f[(size_t)l * n + id[(size_t)j * u + k]];
  }
}

/* canonical order for the atomics: (j, k) ascending per channel */
ORC_API void orc_grouping_bwd(const float *grad_y, const int32_t *indices, int b, int c, int n,
                              int m, int u, float *grad_x) {
#pragma omp parallel for schedule(static) num_threads(orc_threads(b))
  for (int bi = 0; bi < b; ++bi) {
    const float *gy = grad_y + (size_t)bi * m * u * c;
    const int32_t *id = indices + (size_t)bi * m * u;
    float *gx = grad_x + (size_t)bi * n * c;
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < m; ++j)
        for (int k = 0; k < u; ++k) {
          float *dst = gx + (size_t)l * n + id[(size_t)j * u + k];
          *dst = *dst + gy[((size_t)l * m + j) * u + k];
        }
  }
}

/* gather forward/backward: sampling/sampling.cu:17-31, 52-66. */
ORC_API void orc_gather_fwd(const float *features, const int32_t *indices, int b, int c, int n,
                            int m, float *out) {
#pragma omp parallel for schedule(static) num_threads(orc_threads(b))
  for (int bi = 0; bi < b; ++bi)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < m; ++j)
        out[((size_t)bi * c + l) * m + j] =
            features[((size_t)bi * c + l) * n + indices[(size_t)bi * m + j]];
}

ORC_API void orc_gather_bwd(const float *grad_y, const int32_t *indices, int b, int c, int n,
                            int m, float *grad_x) {
#pragma omp parallel for schedule(static) num_threads(orc_threads(b))
  for (int bi = 0; bi < b; ++bi)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < m; ++j) {
        float *dst = grad_x + ((size_t)bi * c + l) * n + indices[(size_t)bi * m + j];
        *dst = *dst + grad_y[((size_t)bi * c + l) * m + j];
      }
}

/* ------------------------------------------------------------------------------------
 * furthest point sampling: sampling/sampling.cu:86-167, always launched with 512 threads
 * (:171).  distances must be 1e38f on entry (sampling.cpp:53-54), indices zero.
 * The tie rule is an artefact of the launch shape and is restated literally: slot t owns
 * k = t, t+512, ... with a strict '>' (lowest k of the slot wins ties, :141-144); the
 * 512-slot tree keeps the LEFT operand on ties (:154).  d is nvcc-contracted.
 * ---------------------------------------------------------------------------------- */
#define FPS_SLOTS 512
ORC_API void orc_fps(const float *coords, int b, int n, int m, float *distances,
                     int32_t *indices) {
  if (m <= 0) return;
#pragma omp parallel for schedule(static) num_threads(orc_threads(b))
  for (int bi = 0; bi < b; ++bi) {
    const float *co = coords + (size_t)bi * n * 3;
    float *dist = distances + (size_t)bi * n;
    int32_t *idx = indices + (size_t)bi * m;
    float dists[FPS_SLOTS];
    int dists_i[FPS_SLOTS];
    int old = 0;
    idx[0] = old;
    for (int j = 1; j < m; ++j) {
      const float x1 = co[old], y1 = co[old + n], z1 = co[old + n + n];
      for (int t = 0; t < FPS_SLOTS; ++t) {
        int besti = 0;
        float best = -1;
        for (int k = t; k < n; k += FPS_SLOTS) {
          const float td = dist[k];
          const float ex = co[k] - x1, ey = co[k + n] - y1, ez = co[k + n + n] - z1;
          const float d = fmaf(ez, ez, fmaf(ex, ex, ey * ey));
          const float d2 = fminf(d, td);
          if (d2 != td) dist[k] = d2;
          if (d2 > best) {
            best = d2;
            besti = k;
          }
        }
        dists[t] = best;
        dists_i[t] = besti;
      }
      for (int u = 0; (1 << u) < FPS_SLOTS; ++u)
        for (int t = 0; t < (FPS_SLOTS >> (u + 1)); ++t) {
          const int i1 = (t * 2) << u, i2 = (t * 2 + 1) << u;
          if (dists[i1] < dists[i2]) {
            dists[i1] = dists[i2];
            dists_i[i1] = dists_i[i2];
          }
        }
      old = dists_i[0];
      idx[j] = old;
    }
  }
}

/* ------------------------------------------------------------------------------------
 * 3-NN inverse-squared-distance interpolation:
 * interpolate/neighbor_interpolate.cu:20-75 (neighbour search, running minima kept as
 * double, init 1e40, strict '<'), :61-72 (clamp to [1e-10, 1e10] and product-form
 * weights), :90-116 (interpolation: fma(f3,w3, fma(f1,w1, f2*w2))), :145-170 (backward).
 * ---------------------------------------------------------------------------------- */
ORC_API void orc_three_nn_interp_fwd(const float *points_coords, const float *centers_coords,
                                     const float *centers_features, int b, int c, int m, int n,
                                     int32_t *indices, float *weights, float *out) {
#pragma omp parallel for schedule(static) num_threads(orc_threads(b))
  for (int bi = 0; bi < b; ++bi) {
    const float *pc = points_coords + (size_t)bi * 3 * n;
    const float *cc = centers_coords + (size_t)bi * 3 * m;
    float *w = weights + (size_t)bi * 3 * n;
    int32_t *id = indices + (size_t)bi * 3 * n;
    for (int j = 0; j < n; ++j) {
      const float ux = pc[j], uy = pc[j + n], uz = pc[j + n + n];
      double best0 = 1e40, best1 = 1e40, best2 = 1e40;
      int besti0 = 0, besti1 = 0, besti2 = 0;
      for (int k = 0; k < m; ++k) {
        const float ex = ux - cc[k], ey = uy - cc[k + m], ez = uz - cc[k + m + m];
        const float d = fmaf(ez, ez, fmaf(ex, ex, ey * ey));
        if (d < best2) {
          best2 = d;
          besti2 = k;
          if (d < best1) {
            best2 = best1;
            besti2 = besti1;
            best1 = d;
            besti1 = k;
            if (d < best0) {
              best1 = best0;
              besti1 = besti0;
              best0 = d;
              besti0 = k;
            }
          }
        }
      }
      best0 = fmax(fmin((double)1e10f, best0), (double)1e-10f);
      best1 = fmax(fmin((double)1e10f, best1), (double)1e-10f);
      best2 = fmax(fmin((double)1e10f, best2), (double)1e-10f);
      const float d0d1 = (float)(best0 * best1);
      const float d0d2 = (float)(best0 * best2);
      const float d1d2 = (float)(best1 * best2);
      const float d0d1d2 = 1.0f / (d0d1 + d0d2 + d1d2);
      w[j] = d1d2 * d0d1d2;
      id[j] = besti0;
      w[j + n] = d0d2 * d0d1d2;
      id[j + n] = besti1;
      w[j + n + n] = d0d1 * d0d1d2;
      id[j + n + n] = besti2;
    }
    const float *cf = centers_features + (size_t)bi * m * c;
    float *o = out + (size_t)bi * n * c;
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < n; ++j) {
        const float *cfl = cf + (size_t)l * m;
        o[(size_t)l * n + j] =
            fmaf(cfl[id[j + n + n]], w[j + n + n], fmaf(cfl[id[j]], w[j], cfl[id[j + n]] * w[j + n]));
      }
  }
}

/* canonical order for the atomics: point j ascending, neighbour 1,2,3 per channel */
ORC_API void orc_three_nn_interp_bwd(const float *grad_y, const int32_t *indices,
                                     const float *weights, int b, int c, int n, int m,
                                     float *grad_x) {
#pragma omp parallel for schedule(static) num_threads(orc_threads(b))
  for (int bi = 0; bi < b; ++bi) {
    const float *gy = grad_y + (size_t)bi * n * c;
    const int32_t *id = indices + (size_t)bi * n * 3;
    const float *w = weights + (size_t)bi * n * 3;
    float *gx = grad_x + (size_t)bi * m * c;
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < n; ++j) {
        const float g = gy[(size_t)l * n + j];
        float *gxl = gx + (size_t)l * m;
        gxl[id[j]] = gxl[id[j]] + g * w[j];
        gxl[id[j + n]] = gxl[id[j + n]] + g * w[j + n];
        gxl[id[j + n + n]] = gxl[id[j + n + n]] + g * w[j + n + n];
      }
  }
}
