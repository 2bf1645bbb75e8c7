/*
 * pcp_oracle.c -- CPU restatement (colour path A0-A8, z-buffer cull, scores,
 * top-5 mean).  See pcp_oracle.h: TEST INFRASTRUCTURE ONLY, PARITY UNPINNED.
 *
 * Every function cites the reference lines it restates.  PCP/ =
 * /root/reference/PointCloudProcessor/.  Build: oracle/Makefile
 * (-O2 -ffp-contract=off -fno-fast-math).
 */
#include "pcp_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ */
/* defaults                                                            */
/* ------------------------------------------------------------------ */

/* PCP/src/PointCloudProcessor.cpp:57-62 (K, D), :525 (cull size). */
void orc_default_camera(orc_camera *cam) {
  cam->fx = 4818.200388954926;
  cam->fy = 4819.10345841615;
  cam->cx = 2032.4178620390019;
  cam->cy = 1535.1895959282901;
  cam->k1 = 0.003043514741045163;
  cam->k2 = 0.06634739187544138;
  cam->p1 = -0.000217681797407554;
  cam->p2 = -0.0006654964142658197;
  cam->k3 = 0.0;
  cam->image_width = 4096;
  cam->image_height = 3000;
  cam->cull_width = 4096;
  cam->cull_height = 3000;
}

/* PCP/include/vlcal/calib/view_culling.hpp:12-15; view_culling.cpp:63,157 */
void orc_default_cull_params(orc_cull_params *p) {
  p->enable_depth_buffer_culling = 1;
  p->downsample_factor = 14;
  p->depth_slack = 0.05;
  p->cull_mode = ORC_CULL_ZBUFFER;
  p->match_mode = ORC_MATCH_ROUNDTRIP;
  p->hpr_flip_radius = 90000.0; /* view_culling.hpp:14 */
}

/* PCP/src/PointCloudProcessor.cpp:67-86 */
void orc_default_mls_params(orc_mls_params *p) {
  p->search_radius = 0.03;
  p->sqr_gauss_param = 0.0009;
  p->polynomial_order = 2;
  p->compute_normals = 1;
  p->upsampling = 3;
  p->vgd_iterations = 4;
  p->vgd_voxel_size = 0.001f;
  p->threads = 30;
}

int32_t orc_hardware_threads(void) {
#ifdef _OPENMP
  return (int32_t)omp_get_num_procs();
#else
  return 1;
#endif
}

static int resolve_threads(int32_t t) {
#ifdef _OPENMP
  int hw = omp_get_num_procs();
  if (t <= 0 || t > hw) return hw;
  return t;
#else
  (void)t;
  return 1;
#endif
}

/* ------------------------------------------------------------------ */
/* A1 pose -> matrices                                                 */
/* ------------------------------------------------------------------ */

/* Eigen::Quaterniond::toRotationMatrix (no normalisation) [upstream Eigen 3.3
 * Quaternion.h], reached from t_c2w.rotate(q), PCP/src/PointCloudProcessor.cpp:497. */
static void quat_to_rot(const orc_pose *p, double R[9]) {
  const double tx = 2.0 * p->qx, ty = 2.0 * p->qy, tz = 2.0 * p->qz;
  const double twx = tx * p->qw, twy = ty * p->qw, twz = tz * p->qw;
  const double txx = tx * p->qx, txy = ty * p->qx, txz = tz * p->qx;
  const double tyy = ty * p->qy, tyz = tz * p->qy, tzz = tz * p->qz;
  R[0] = 1.0 - (tyy + tzz);
  R[1] = txy - twz;
  R[2] = txz + twy;
  R[3] = txy + twz;
  R[4] = 1.0 - (txx + tzz);
  R[5] = tyz - twx;
  R[6] = txz - twy;
  R[7] = tyz + twx;
  R[8] = 1.0 - (txx + tyy);
}

/* General fp32 affine inverse: Eigen Transform<float,3,Affine>::inverse(Affine)
 * = [L^-1 | -L^-1 t]  (PCP/src/PointCloudProcessor.cpp:509,518; again at :578 for
 * every match).  [upstream Eigen 3.3.7] LU/InverseImpl.h compute_inverse<.,.,3>:
 * cofactor_3x3<i,j> = m(i1,j1) m(i2,j2) - m(i1,j2) m(i2,j1), i1 = (i+1)%3 ...;
 * det = (cofactors_col0 .* m.col(0)).sum(); inverse(j,i) = cofactor<i,j> * (1/det).
 * A 3-element fp32 sum has no SSE packet (4 floats): Redux.h's redux_novec_unroller
 * <0,3> splits 1 + 2, i.e. p0 + (p1 + p2).  The translation is
 * (-L^-1) * t, a 3x3 * 3x1 coefficient-based product whose 3 fp32 coefficients are
 * each such a sum: (-l0 t0) + ((-l1 t1) + (-l2 t2)). */
static void affine_inverse_f32(const float m[12], float out[12]) {
  const float a = m[0], b = m[1], c = m[2];
  const float d = m[4], e = m[5], f = m[6];
  const float g = m[8], h = m[9], i = m[10];
  const float c00 = e * i - f * h; /* cofactor<0,0> */
  const float c10 = h * c - i * b; /* cofactor<1,0> */
  const float c20 = b * f - c * e; /* cofactor<2,0> */
  const float det = c00 * a + (c10 * d + c20 * g);
  const float inv = 1.0f / det;
  float L[9];
  L[0] = c00 * inv;
  L[1] = c10 * inv;
  L[2] = c20 * inv;
  L[3] = (f * g - d * i) * inv; /* cofactor<0,1> */
  L[4] = (i * a - g * c) * inv; /* cofactor<1,1> */
  L[5] = (c * d - a * f) * inv; /* cofactor<2,1> */
  L[6] = (d * h - e * g) * inv; /* cofactor<0,2> */
  L[7] = (g * b - h * a) * inv; /* cofactor<1,2> */
  L[8] = (a * e - b * d) * inv; /* cofactor<2,2> */
  const float t0 = m[3], t1 = m[7], t2 = m[11];
  for (int r = 0; r < 3; ++r) {
    out[4 * r + 0] = L[3 * r + 0];
    out[4 * r + 1] = L[3 * r + 1];
    out[4 * r + 2] = L[3 * r + 2];
    out[4 * r + 3] = (-L[3 * r + 0]) * t0 + ((-L[3 * r + 1]) * t1 + (-L[3 * r + 2]) * t2);
  }
}

/* exported for the tests (oracle_capi.affine_inverse) */
void orc_affine_inverse_f32(const float m[12], float out[12]) { affine_inverse_f32(m, out); }

/* PCP/src/PointCloudProcessor.cpp:495-519 (same at :186-194). */
void orc_pose_to_matrices(const orc_pose *pose, const double *T_opt, float w2c[12], float c2w[12]) {
  double R[9];
  quat_to_rot(pose, R);
  const double t[3] = {pose->x, pose->y, pose->z};
  if (!T_opt) {
    /* Isometry3d::inverse(): [R^T | -(R^T t)], then cast<float>() (:499-501). */
    for (int r = 0; r < 3; ++r) {
      const double a0 = R[0 + r], a1 = R[3 + r], a2 = R[6 + r]; /* row r of R^T */
      w2c[4 * r + 0] = (float)a0;
      w2c[4 * r + 1] = (float)a1;
      w2c[4 * r + 2] = (float)a2;
      w2c[4 * r + 3] = (float)(-((a0 * t[0] + a1 * t[1]) + a2 * t[2]));
      c2w[4 * r + 0] = (float)R[3 * r + 0];
      c2w[4 * r + 1] = (float)R[3 * r + 1];
      c2w[4 * r + 2] = (float)R[3 * r + 2];
      c2w[4 * r + 3] = (float)t[r];
    }
    return;
  }
  /* (t_c2w * T_opt).cast<float>(), then general fp32 inverse (:506-509,515-518). */
  double M[12];
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 4; ++c) {
      double s = (R[3 * r + 0] * T_opt[0 * 4 + c] + R[3 * r + 1] * T_opt[1 * 4 + c]) + R[3 * r + 2] * T_opt[2 * 4 + c];
      if (c == 3) s = s + t[r] * T_opt[3 * 4 + 3];
      M[4 * r + c] = s;
    }
  }
  for (int k = 0; k < 12; ++k) c2w[k] = (float)M[k];
  affine_inverse_f32(c2w, w2c);
}

/* ------------------------------------------------------------------ */
/* A2 transform                                                        */
/* ------------------------------------------------------------------ */

/* pcl::transformPointCloud(..., Affine3f), PCL 1.10 detail::Transformer::se3
 * SSE path: x*c0 + (y*c1 + (z*c2 + c3)) per output row [upstream].  Call sites
 * PCP/src/PointCloudProcessor.cpp:196,521,549,555. */
static inline void xform_point(const float m[12], float x, float y, float z, float *xc, float *yc, float *zc) {
  *xc = x * m[0] + (y * m[1] + (z * m[2] + m[3]));
  *yc = x * m[4] + (y * m[5] + (z * m[6] + m[7]));
  *zc = x * m[8] + (y * m[9] + (z * m[10] + m[11]));
}

void orc_transform(const float m[12], const float *x, const float *y, const float *z, int64_t n, float *xc,
                   float *yc, float *zc) {
  for (int64_t i = 0; i < n; ++i) xform_point(m, x[i], y[i], z[i], &xc[i], &yc[i], &zc[i]);
}

/* ------------------------------------------------------------------ */
/* A3 projection                                                       */
/* ------------------------------------------------------------------ */

/* PCP/include/camera/pinhole.hpp:13-51; duplicate PCP/include/PointCloudProcessor.hpp:100-123.
 * Left-to-right association exactly as written in the source. */
void orc_project_point(const orc_camera *cam, double xc, double yc, double zc, double *u, double *v) {
  const double xn = xc / zc;
  const double yn = yc / zc;
  const double x2 = xn * xn;
  const double y2 = yn * yn;
  const double r2 = x2 + y2;
  const double r4 = r2 * r2;
  const double r6 = r2 * r4;
  const double rc = ((1.0 + cam->k1 * r2) + cam->k2 * r4) + cam->k3 * r6;
  const double t1 = (2.0 * xn) * yn;
  const double t2 = r2 + 2.0 * x2;
  const double t3 = r2 + 2.0 * y2;
  const double xd = (rc * xn + cam->p1 * t1) + cam->p2 * t2;
  const double yd = (rc * yn + cam->p1 * t3) + cam->p2 * t1;
  *u = cam->fx * xd + cam->cx;
  *v = cam->fy * yd + cam->cy;
}

/* float/double -> int32 with C truncation; values that do not fit (or NaN) are
 * UB in the reference (Appendix B6) and are rejected here. */
static inline int trunc_d(double v, int32_t *out) {
  if (!(v > -2147483648.0 && v < 2147483648.0)) return 0;
  *out = (int32_t)v;
  return 1;
}
static inline int trunc_f(float v, int32_t *out) {
  if (!(v > -2147483648.0f && v < 2147483648.0f)) return 0;
  *out = (int32_t)v;
  return 1;
}

typedef struct projected {
  float xc, yc, zc;
  double range;   /* ||p_c|| fp64 (view_culling.cpp:102,144) */
  int32_t cell;   /* >=0 in map, -2 candidate outside map, -1 rejected */
  int32_t pixel;  /* >=0 colour pixel, -1 rejected */
} projected;

/* view_culling.cpp:27-38 (float->double promote), :76 (z test, B6: z<=0 rejected),
 * :86-90 (project, cast<float>, /14, cast<int>, bounds vs full image_size),
 * :116 (map bounds), :102 (norm).  Colour pixel: PointCloudProcessor.cpp:748-754. */
static inline void project_one(const orc_camera *cam, const orc_cull_params *cp, const float m[12], float x,
                               float y, float z, projected *o) {
  xform_point(m, x, y, z, &o->xc, &o->yc, &o->zc);
  o->cell = -1;
  o->pixel = -1;
  o->range = (double)FLT_MAX;
  if (!(o->zc > 0.0f)) return;
  const double X = (double)o->xc, Y = (double)o->yc, Z = (double)o->zc;
  o->range = sqrt((X * X + Y * Y) + Z * Z);
  double u, v;
  orc_project_point(cam, X, Y, Z, &u, &v);
  /* A4' hidden_points_removal's candidate filter, view_culling.cpp:284-288:
   * project(p).cast<int>() against the full image_size; no map, reported as -2 */
  if (cp->cull_mode != ORC_CULL_ZBUFFER) {
    int32_t ui, vi;
    if (trunc_d(u, &ui) && trunc_d(v, &vi) && ui >= 0 && ui < cam->cull_width && vi >= 0 && vi < cam->cull_height)
      o->cell = -2;
  } else
  /* A4 cell */
  {
    const float ds = (float)cp->downsample_factor;
    const float uf = (float)u, vf = (float)v;
    int32_t cx, cy;
    if (trunc_f(uf / ds, &cx) && trunc_f(vf / ds, &cy)) {
      if (cx >= 0 && cy >= 0 && cx < cam->cull_width && cy < cam->cull_height) {
        const int32_t mw = cam->cull_width / cp->downsample_factor;
        const int32_t mh = cam->cull_height / cp->downsample_factor;
        /* -2 (candidate without a map cell) only matters, and is only reported, when the
         * depth buffer is off: with it on such a point is dropped in pass 2 (:166-169) */
        o->cell = (cx < mw && cy < mh) ? cy * mw + cx : (cp->enable_depth_buffer_culling ? -1 : -2);
      }
    }
  }
  /* A5 pixel */
  {
    int32_t ui, vi;
    if (trunc_d(u, &ui) && trunc_d(v, &vi)) {
      if (ui >= 0 && ui < cam->image_width && vi >= 0 && vi < cam->image_height)
        o->pixel = vi * cam->image_width + ui;
    }
  }
}

void orc_project_frame(const orc_camera *cam, const orc_cull_params *cp, const float w2c[12], const float *x,
                       const float *y, const float *z, int64_t n, int32_t *out_cell, int32_t *out_pixel,
                       float *out_range, float *out_xc, float *out_yc, float *out_zc) {
  for (int64_t i = 0; i < n; ++i) {
    projected p;
    project_one(cam, cp, w2c, x[i], y[i], z[i], &p);
    if (out_cell) out_cell[i] = p.cell;
    if (out_pixel) out_pixel[i] = p.pixel;
    if (out_range) out_range[i] = (float)p.range;
    if (out_xc) out_xc[i] = p.xc;
    if (out_yc) out_yc[i] = p.yc;
    if (out_zc) out_zc[i] = p.zc;
  }
}

/* ------------------------------------------------------------------ */
/* A4 z-buffer cull                                                    */
/* ------------------------------------------------------------------ */

static void depth_map_fill(float *map, int64_t cells) {
  for (int64_t i = 0; i < cells; ++i) map[i] = FLT_MAX; /* view_culling.cpp:64 */
}

/* pass 1, view_culling.cpp:99-125: if (dist > map) continue; map = (float)dist.
 * Sequential result == MIN over f32(dist) (rounding is monotone), so the
 * threaded variant below (private maps + MIN merge) is exactly equivalent. */
static void depth_pass(const orc_camera *cam, const orc_cull_params *cp, const float m[12], const float *x,
                       const float *y, const float *z, int64_t n, float *map, int threads) {
  const int64_t cells = (int64_t)(cam->cull_width / cp->downsample_factor) * (cam->cull_height / cp->downsample_factor);
  depth_map_fill(map, cells);
  if (threads <= 1) {
    for (int64_t i = 0; i < n; ++i) {
      projected p;
      project_one(cam, cp, m, x[i], y[i], z[i], &p);
      if (p.cell < 0) continue;
      if (p.range > (double)map[p.cell]) continue;
      map[p.cell] = (float)p.range;
    }
    return;
  }
#ifdef _OPENMP
  /* shared map, lock-free MIN on the bit pattern of the positive fp32 range
   * (unsigned order == float order), same final map as the sequential loop */
  {
    uint32_t *bits = (uint32_t *)map;
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int64_t i = 0; i < n; ++i) {
      projected p;
      project_one(cam, cp, m, x[i], y[i], z[i], &p);
      if (p.cell < 0) continue;
      const float rf = (float)p.range;
      uint32_t nb;
      memcpy(&nb, &rf, 4);
      uint32_t cur = __atomic_load_n(&bits[p.cell], __ATOMIC_RELAXED);
      while (nb < cur && !__atomic_compare_exchange_n(&bits[p.cell], &cur, nb, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
      }
    }
  }
#endif
}

/* pass 2 keep rule, view_culling.cpp:135-171: keep iff cell in map and
 * !(dist > (double)map + 0.05); without depth-buffer culling every candidate
 * (cell >= 0 or -2) is kept (:92-93, indices.emplace_back). */
static inline int keep_rule(const orc_cull_params *cp, const projected *p, const float *map) {
  if (!cp->enable_depth_buffer_culling || cp->cull_mode != ORC_CULL_ZBUFFER) return p->cell != -1;
  if (p->cell < 0) return 0;
  return !(p->range > (double)map[p->cell] + cp->depth_slack);
}

static inline int zbuf_on(const orc_cull_params *cp) {
  return cp->enable_depth_buffer_culling && cp->cull_mode == ORC_CULL_ZBUFFER;
}

/* ORC_CULL_HPR: the keep mask of ViewCulling::hidden_points_removal for one keyframe (pcp_oracle_hpr.c), n bytes in
 * `hull`; keep_rule's candidate test (cell != -1) is then ANDed with it.  Other modes: hull is left untouched. */
static inline int hull_mask(const orc_camera *cam, const orc_cull_params *cp, const float w2c[12], const float *x,
                            const float *y, const float *z, int64_t n, uint8_t *hull) {
  if (cp->cull_mode != ORC_CULL_HPR) return 0;
  return orc_hpr_frame(cam, w2c, x, y, z, n, cp->hpr_flip_radius, hull, NULL) < 0 ? -1 : 0;
}

int64_t orc_cull_frame(const orc_camera *cam, const orc_cull_params *cp, const float w2c[12], const float *x,
                       const float *y, const float *z, int64_t n, uint8_t *out_keep, float *depth_map,
                       int32_t threads) {
  const int nt = threads == 1 ? 1 : resolve_threads(threads);
  const int64_t cells = (int64_t)(cam->cull_width / cp->downsample_factor) * (cam->cull_height / cp->downsample_factor);
  float *map = depth_map ? depth_map : (float *)malloc((size_t)(cells > 0 ? cells : 1) * sizeof(float));
  if (zbuf_on(cp))
    depth_pass(cam, cp, w2c, x, y, z, n, map, nt);
  else
    depth_map_fill(map, cells);
  int64_t kept = 0;
  uint8_t *hull = cp->cull_mode == ORC_CULL_HPR ? (uint8_t *)malloc((size_t)(n > 0 ? n : 1)) : NULL;
  if (hull) hull_mask(cam, cp, w2c, x, y, z, n, hull);
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nt) reduction(+ : kept)
#endif
  for (int64_t i = 0; i < n; ++i) {
    projected p;
    project_one(cam, cp, w2c, x[i], y[i], z[i], &p);
    const int k = keep_rule(cp, &p, map) && (!hull || hull[i]);
    if (out_keep) out_keep[i] = (uint8_t)k;
    kept += k;
  }
  free(hull);
  if (!depth_map) free(map);
  return kept;
}

/* ------------------------------------------------------------------ */
/* A6 scores                                                           */
/* ------------------------------------------------------------------ */

/* computeOrientationScore PCP/include/PointCloudProcessor.hpp:205-220 (B4: world
 * camera position subtracted from a camera-frame point, reproduced);
 * computeDistanceScore :222-236; final PCP/src/PointCloudProcessor.cpp:588.
 * Identity mode (Appendix B3): the camera-frame point is the transform output. */
void orc_scores(float xc, float yc, float zc, const orc_pose *pose, float *orientation, float *distance,
                float *final_score) {
  const double dx = (double)xc - pose->x, dy = (double)yc - pose->y, dz = (double)zc - pose->z;
  const double sq = (dx * dx + dy * dy) + dz * dz;
  /* Eigen normalized(): v / sqrt(sq) if sq > 0 else v [upstream]; dot with (0,0,1) = z */
  const double cosA = sq > 0.0 ? dz / sqrt(sq) : dz;
  float o = (float)((cosA + 1.0) / 2.0);
  o = 0.2f + 0.8f * o;
  const float dist = sqrtf((xc * xc + yc * yc) + zc * zc);
  const float diff = fabsf(dist - 2.0f);
  float nd = diff / 2.0f;
  if (!(nd < 1.0f)) nd = 1.0f; /* std::min(nd, 1.0f) */
  float d = 1.0f - nd;
  d = 0.2f + 0.8f * d;
  if (orientation) *orientation = o;
  if (distance) *distance = d;
  if (final_score) *final_score = (float)((double)(o + d) / 2.0);
}

/* ------------------------------------------------------------------ */
/* A6 / B3: the reference's camera -> world -> camera round trip        */
/* ------------------------------------------------------------------ */

/* kdtree.radiusSearch(searchPoint, epsilon) with `const float epsilon = 1e-5`
 * (PCP/src/PointCloudProcessor.cpp:482,571): pcl::KdTreeFLANN::radiusSearch takes the
 * radius as double and hands FLANN static_cast<float>(radius * radius) [upstream PCL 1.10
 * kdtree_flann.hpp]. */
static inline float match_radius_sq(void) {
  const double eps = (double)1e-5f;
  return (float)(eps * eps);
}

/* flann::L2_Simple<float> [upstream FLANN 1.9.1 dist.h]: result = 0; result += diff*diff per
 * dimension, fp32; a point is reported iff dist < radius (strict). */
static inline float l2_simple_f32(float ax, float ay, float az, float bx, float by, float bz) {
  const float dx = ax - bx, dy = ay - by, dz = az - bz;
  float r = dx * dx;
  r += dy * dy;
  r += dz * dz;
  return r;
}

/* PCP/src/PointCloudProcessor.cpp:578-579: Eigen::Vector4f ptCamera =
 * transformation_c2w_optimized.inverse() * ptWorld.  [upstream Eigen 3.3.7 Transform.h]
 * Affine * 4-vector = T.affine() (3x4 block, column-major) * v: a coefficient-based
 * product whose rows have no packet access, so each coefficient is a 4-term fp32 sum
 * through Redux.h's redux_novec_unroller<0,4>: (p0 + p1) + (p2 + p3), p3 = m3 * 1.0f. */
static inline void affine_times_point_eigen(const float m[12], float x, float y, float z, float *ox, float *oy,
                                            float *oz) {
  *ox = (m[0] * x + m[1] * y) + (m[2] * z + m[3] * 1.0f);
  *oy = (m[4] * x + m[5] * y) + (m[6] * z + m[7] * 1.0f);
  *oz = (m[8] * x + m[9] * y) + (m[10] * z + m[11] * 1.0f);
}

/* ------------------------------------------------------------------ */
/* A8 top-5 accumulate / finalise                                      */
/* ------------------------------------------------------------------ */

#define ORC_TOPM 5 /* PCP/src/PointCloudProcessor.cpp:615 */

typedef struct top5 {
  float score[ORC_TOPM];
  uint32_t rgb[ORC_TOPM]; /* 0x00RRGGBB */
  int32_t frame[ORC_TOPM];
  int32_t count;
} top5;

/* Streaming equivalent of "collect all, std::sort descending by finalScore, keep
 * 5" (PCP/src/PointCloudProcessor.cpp:612-615) with ties -> lower keyframe index
 * (B8): frames arrive in ascending order, a new entry goes after equal scores. */
static inline void top5_insert(top5 *t, float score, uint32_t rgb, int32_t frame) {
  if (t->count < INT32_MAX) t->count++;
  int pos = ORC_TOPM;
  for (int k = 0; k < ORC_TOPM; ++k) {
    if (t->frame[k] < 0 || score > t->score[k]) {
      pos = k;
      break;
    }
  }
  if (pos == ORC_TOPM) return;
  for (int k = ORC_TOPM - 1; k > pos; --k) {
    t->score[k] = t->score[k - 1];
    t->rgb[k] = t->rgb[k - 1];
    t->frame[k] = t->frame[k - 1];
  }
  t->score[pos] = score;
  t->rgb[pos] = rgb;
  t->frame[pos] = frame;
}

/* smoothColors PCP/src/PointCloudProcessor.cpp:616-629: fp32 sums in sorted
 * order, r/total truncated to uint8; no entries -> (0,0,0) (B7). */
static inline void top5_finalise(const top5 *t, uint8_t rgb[3]) {
  float total = 0.0f, r = 0.0f, g = 0.0f, b = 0.0f;
  int m = 0;
  for (int k = 0; k < ORC_TOPM; ++k) {
    if (t->frame[k] < 0) break;
    const float s = t->score[k];
    r += (float)((t->rgb[k] >> 16) & 0xff) * s;
    g += (float)((t->rgb[k] >> 8) & 0xff) * s;
    b += (float)(t->rgb[k] & 0xff) * s;
    total += s;
    ++m;
  }
  if (m == 0) {
    rgb[0] = rgb[1] = rgb[2] = 0;
    return;
  }
  rgb[0] = (uint8_t)(r / total);
  rgb[1] = (uint8_t)(g / total);
  rgb[2] = (uint8_t)(b / total);
}

int orc_colorize(const orc_camera *cam, const orc_cull_params *cp, const float *x, const float *y, const float *z,
                 int64_t n, const orc_pose *poses, int32_t n_frames, const double *T_opt, int32_t T_opt_stride,
                 const uint8_t *const *images, uint8_t *out_rgb, uint8_t *out_has, int32_t *out_count,
                 float *out_top_score, uint32_t *out_top_rgb, int32_t *out_top_frame, int32_t threads) {
  const int nt = threads == 1 ? 1 : resolve_threads(threads);
  const int64_t cells = (int64_t)(cam->cull_width / cp->downsample_factor) * (cam->cull_height / cp->downsample_factor);
  float *map = (float *)malloc((size_t)(cells > 0 ? cells : 1) * sizeof(float));
  top5 *state = (top5 *)malloc((size_t)(n > 0 ? n : 1) * sizeof(top5));
  if (!map || !state) {
    free(map);
    free(state);
    return -1;
  }
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nt)
#endif
  for (int64_t i = 0; i < n; ++i) {
    state[i].count = 0;
    for (int k = 0; k < ORC_TOPM; ++k) {
      state[i].score[k] = -1.0f;
      state[i].rgb[k] = 0;
      state[i].frame[k] = -1;
    }
  }
  uint8_t *hull = NULL; /* ORC_CULL_HPR: this keyframe's hull vertices */
  for (int32_t f = 0; f < n_frames; ++f) { /* PCP/src/PointCloudProcessor.cpp:488 */
    float w2c[12], c2w[12], c2w_inv[12];
    const double *T = T_opt ? T_opt + (int64_t)T_opt_stride * f : NULL;
    orc_pose_to_matrices(&poses[f], T, w2c, c2w);
    affine_inverse_f32(c2w, c2w_inv); /* transformation_c2w_optimized.inverse(), :578 */
    const float r2 = match_radius_sq();
    if (zbuf_on(cp)) depth_pass(cam, cp, w2c, x, y, z, n, map, nt);
    if (cp->cull_mode == ORC_CULL_HPR) {
      if (!hull) hull = (uint8_t *)malloc((size_t)(n > 0 ? n : 1));
      if (!hull || hull_mask(cam, cp, w2c, x, y, z, n, hull) != 0) {
        free(hull);
        free(map);
        free(state);
        return -1;
      }
    }
    const uint8_t *img = images[f];
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nt)
#endif
    for (int64_t i = 0; i < n; ++i) {
      projected p;
      project_one(cam, cp, w2c, x[i], y[i], z[i], &p);
      if (!keep_rule(cp, &p, map) || (hull && !hull[i])) continue; /* ViewCulling::cull, :527 */
      if (p.pixel < 0) continue;             /* generateColorMap bounds, :748-754 */
      const uint8_t *px = img + (int64_t)p.pixel * 3; /* BGR, :760-762 */
      const uint32_t rgb = ((uint32_t)px[2] << 16) | ((uint32_t)px[1] << 8) | (uint32_t)px[0];
      float sx = p.xc, sy = p.yc, sz = p.zc;
      if (cp->match_mode == ORC_MATCH_ROUNDTRIP) {
        float wx, wy, wz;
        xform_point(c2w, p.xc, p.yc, p.zc, &wx, &wy, &wz);            /* :555 */
        if (!(l2_simple_f32(wx, wy, wz, x[i], y[i], z[i]) < r2)) continue; /* :571: point i does not find itself */
        affine_times_point_eigen(c2w_inv, wx, wy, wz, &sx, &sy, &sz);  /* :578-579 */
      }
      float fs;
      orc_scores(sx, sy, sz, &poses[f], NULL, NULL, &fs); /* :584-588 */
      top5_insert(&state[i], fs, rgb, f);                /* :590-591 */
    }
  }
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nt)
#endif
  for (int64_t i = 0; i < n; ++i) {
    uint8_t rgb[3];
    top5_finalise(&state[i], rgb);
    out_rgb[3 * i + 0] = rgb[0];
    out_rgb[3 * i + 1] = rgb[1];
    out_rgb[3 * i + 2] = rgb[2];
    /* removePointsWithNoColor PCP/include/PointCloudProcessor.hpp:238-252 */
    if (out_has) out_has[i] = (uint8_t)(rgb[0] != 0 || rgb[1] != 0 || rgb[2] != 0);
    if (out_count) out_count[i] = state[i].count;
    for (int k = 0; k < ORC_TOPM; ++k) {
      if (out_top_score) out_top_score[ORC_TOPM * i + k] = state[i].score[k];
      if (out_top_rgb) out_top_rgb[ORC_TOPM * i + k] = state[i].rgb[k];
      if (out_top_frame) out_top_frame[ORC_TOPM * i + k] = state[i].frame[k];
    }
  }
  free(hull);
  free(map);
  free(state);
  return 0;
}

/* ------------------------------------------------------------------ */
/* B3 faithful mode: the reference's own match-back                     */
/* ------------------------------------------------------------------ */

/* Spatial hash over the map points for radiusSearch(1e-5): cells of 4e-5 m, so the ball
 * of one query touches at most 2 cells per axis.  Stands in for pcl::KdTreeFLANN (exact
 * search, eps = 0): the result SET of a radius search does not depend on the index
 * structure, only on the distance functor and the strict comparison. */
typedef struct cell_entry {
  uint64_t key;
  int32_t index;
} cell_entry;

static const double kMatchCell = 4e-5;

static inline uint64_t cell_key(int64_t ix, int64_t iy, int64_t iz) {
  /* 21 bits per axis after an offset of 2^20 cells (+-41.9 m at 4e-5 m); larger maps alias cells,
   * which only costs extra distance tests */
  const uint64_t m = (1ull << 21) - 1;
  return (((uint64_t)(ix + (1 << 20)) & m) << 42) | (((uint64_t)(iy + (1 << 20)) & m) << 21) |
         ((uint64_t)(iz + (1 << 20)) & m);
}

static int cmp_cell_entry(const void *a, const void *b) {
  const cell_entry *x = (const cell_entry *)a, *y = (const cell_entry *)b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  return x->index < y->index ? -1 : (x->index > y->index);
}

typedef struct match {
  float dist;
  int32_t index;
} match;

static int cmp_match(const void *a, const void *b) {
  const match *x = (const match *)a, *y = (const match *)b;
  if (x->dist != y->dist) return x->dist < y->dist ? -1 : 1;
  return x->index < y->index ? -1 : (x->index > y->index);
}

/* pcdColorizationAndSmooth exactly as PCP/src/PointCloudProcessor.cpp:474-602 runs it, with the
 * z-buffer (or HPR-candidate) cull of cp: every kept and coloured sample goes to world coordinates in
 * fp32 (:555), is matched back by radiusSearch(1e-5) over the ORIGINAL cloud (:571) -- to no point, to
 * its own point, or to several points -- and every match receives the sample with scores computed from
 * c2w.inverse() * p_w (:578-588).  Matches are visited in FLANN's sorted order (ascending distance).
 * stats (nullable, 4 values): samples, samples without any match, samples whose own point is not among
 * the matches, credits to points other than the sample's own. */
int orc_colorize_faithful(const orc_camera *cam, const orc_cull_params *cp, const float *x, const float *y,
                          const float *z, int64_t n, const orc_pose *poses, int32_t n_frames, const double *T_opt,
                          int32_t T_opt_stride, const uint8_t *const *images, uint8_t *out_rgb, uint8_t *out_has,
                          int32_t *out_count, float *out_top_score, uint32_t *out_top_rgb, int32_t *out_top_frame,
                          int64_t *stats, int32_t threads) {
  const int nt = threads == 1 ? 1 : resolve_threads(threads);
  const int64_t cells = (int64_t)(cam->cull_width / cp->downsample_factor) * (cam->cull_height / cp->downsample_factor);
  float *map = (float *)malloc((size_t)(cells > 0 ? cells : 1) * sizeof(float));
  top5 *state = (top5 *)malloc((size_t)(n > 0 ? n : 1) * sizeof(top5));
  cell_entry *grid = (cell_entry *)malloc((size_t)(n > 0 ? n : 1) * sizeof(cell_entry));
  uint8_t *vis = (uint8_t *)malloc((size_t)(n > 0 ? n : 1));
  if (!map || !state || !grid || !vis) {
    free(map);
    free(state);
    free(grid);
    free(vis);
    return -1;
  }
  for (int64_t i = 0; i < n; ++i) {
    state[i].count = 0;
    for (int k = 0; k < ORC_TOPM; ++k) {
      state[i].score[k] = -1.0f;
      state[i].rgb[k] = 0;
      state[i].frame[k] = -1;
    }
    grid[i].key = cell_key((int64_t)floor((double)x[i] / kMatchCell), (int64_t)floor((double)y[i] / kMatchCell),
                           (int64_t)floor((double)z[i] / kMatchCell));
    grid[i].index = (int32_t)i;
  }
  qsort(grid, (size_t)n, sizeof(cell_entry), cmp_cell_entry); /* kdtree.setInputCloud(cloud), :481 */
  const float r2 = match_radius_sq();
  const double reach = sqrt((double)r2) * 1.0001 + 1e-9; /* per-axis reach of the ball, padded */
  int64_t st[4] = {0, 0, 0, 0};
  for (int32_t f = 0; f < n_frames; ++f) {
    float w2c[12], c2w[12], c2w_inv[12];
    const double *T = T_opt ? T_opt + (int64_t)T_opt_stride * f : NULL;
    orc_pose_to_matrices(&poses[f], T, w2c, c2w);
    affine_inverse_f32(c2w, c2w_inv);
    if (zbuf_on(cp)) depth_pass(cam, cp, w2c, x, y, z, n, map, nt);
    const uint8_t *img = images[f];
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nt)
#endif
    for (int64_t i = 0; i < n; ++i) { /* cull + generateColorMap: which points are in coloredCloud */
      projected p;
      project_one(cam, cp, w2c, x[i], y[i], z[i], &p);
      vis[i] = (uint8_t)(keep_rule(cp, &p, map) && p.pixel >= 0);
    }
    if (cp->cull_mode == ORC_CULL_HPR) { /* the hull is taken over every candidate, then the colour bounds apply */
      uint8_t *hull = (uint8_t *)malloc((size_t)(n > 0 ? n : 1));
      if (hull && hull_mask(cam, cp, w2c, x, y, z, n, hull) == 0)
        for (int64_t i = 0; i < n; ++i) vis[i] = (uint8_t)(vis[i] && hull[i]);
      free(hull);
    }
    for (int64_t i = 0; i < n; ++i) { /* the loop at :559-594, in coloredCloudInWorld order */
      if (!vis[i]) continue;
      projected p;
      project_one(cam, cp, w2c, x[i], y[i], z[i], &p);
      const uint8_t *px = img + (int64_t)p.pixel * 3;
      const uint32_t rgb = ((uint32_t)px[2] << 16) | ((uint32_t)px[1] << 8) | (uint32_t)px[0];
      float wx, wy, wz;
      xform_point(c2w, p.xc, p.yc, p.zc, &wx, &wy, &wz); /* :555 */
      match found[64];
      int nf = 0;
      const int64_t x0 = (int64_t)floor(((double)wx - reach) / kMatchCell), x1 = (int64_t)floor(((double)wx + reach) / kMatchCell);
      const int64_t y0 = (int64_t)floor(((double)wy - reach) / kMatchCell), y1 = (int64_t)floor(((double)wy + reach) / kMatchCell);
      const int64_t z0 = (int64_t)floor(((double)wz - reach) / kMatchCell), z1 = (int64_t)floor(((double)wz + reach) / kMatchCell);
      for (int64_t cx = x0; cx <= x1; ++cx)
        for (int64_t cy = y0; cy <= y1; ++cy)
          for (int64_t cz = z0; cz <= z1; ++cz) {
            const uint64_t key = cell_key(cx, cy, cz);
            int64_t lo = 0, hi = n;
            while (lo < hi) {
              const int64_t mid = (lo + hi) >> 1;
              if (grid[mid].key < key)
                lo = mid + 1;
              else
                hi = mid;
            }
            for (int64_t e = lo; e < n && grid[e].key == key; ++e) {
              const int32_t j = grid[e].index;
              const float d = l2_simple_f32(wx, wy, wz, x[j], y[j], z[j]);
              if (d < r2 && nf < 64) {
                int dup = 0; /* aliased cell keys could list a point twice */
                for (int q = 0; q < nf; ++q) dup |= found[q].index == j;
                if (!dup) {
                  found[nf].dist = d;
                  found[nf].index = j;
                  ++nf;
                }
              }
            }
          }
      st[0] += 1;
      if (nf == 0) {
        st[1] += 1;
        st[2] += 1;
        continue; /* radiusSearch(...) > 0 fails, :571 */
      }
      qsort(found, (size_t)nf, sizeof(match), cmp_match);
      float sx, sy, sz;
      affine_times_point_eigen(c2w_inv, wx, wy, wz, &sx, &sy, &sz); /* :578-579 */
      float fs;
      orc_scores(sx, sy, sz, &poses[f], NULL, NULL, &fs);
      int self = 0;
      for (int q = 0; q < nf; ++q) {
        top5_insert(&state[found[q].index], fs, rgb, f); /* rgbCloud.addPointData(pointIndex, ...), :591 */
        if (found[q].index == (int32_t)i)
          self = 1;
        else
          st[3] += 1;
      }
      if (!self) st[2] += 1;
    }
  }
  for (int64_t i = 0; i < n; ++i) {
    uint8_t rgb[3];
    top5_finalise(&state[i], rgb);
    out_rgb[3 * i + 0] = rgb[0];
    out_rgb[3 * i + 1] = rgb[1];
    out_rgb[3 * i + 2] = rgb[2];
    if (out_has) out_has[i] = (uint8_t)(rgb[0] != 0 || rgb[1] != 0 || rgb[2] != 0);
    if (out_count) out_count[i] = state[i].count;
    for (int k = 0; k < ORC_TOPM; ++k) {
      if (out_top_score) out_top_score[ORC_TOPM * i + k] = state[i].score[k];
      if (out_top_rgb) out_top_rgb[ORC_TOPM * i + k] = state[i].rgb[k];
      if (out_top_frame) out_top_frame[ORC_TOPM * i + k] = state[i].frame[k];
    }
  }
  if (stats) memcpy(stats, st, sizeof(st));
  free(map);
  free(state);
  free(grid);
  free(vis);
  return 0;
}

/* generateColorMap + generateSegmentMap + transform to world,
 * PCP/src/PointCloudProcessor.cpp:531-551,743-766,783-815. */
int64_t orc_frame_visible(const orc_camera *cam, const orc_cull_params *cp, const orc_pose *pose,
                          const double *T_opt, const float *x, const float *y, const float *z, int64_t n,
                          const uint8_t *image, const uint8_t *mask, int32_t *out_index, uint8_t *out_rgb,
                          uint16_t *out_mask, float *out_xyz_cam, float *out_xyz_world) {
  float w2c[12], c2w[12];
  orc_pose_to_matrices(pose, T_opt, w2c, c2w);
  const int64_t cells = (int64_t)(cam->cull_width / cp->downsample_factor) * (cam->cull_height / cp->downsample_factor);
  float *map = (float *)malloc((size_t)(cells > 0 ? cells : 1) * sizeof(float));
  if (zbuf_on(cp)) depth_pass(cam, cp, w2c, x, y, z, n, map, 1);
  uint8_t *hull = cp->cull_mode == ORC_CULL_HPR ? (uint8_t *)malloc((size_t)(n > 0 ? n : 1)) : NULL;
  if (hull) hull_mask(cam, cp, w2c, x, y, z, n, hull);
  int64_t m = 0;
  for (int64_t i = 0; i < n; ++i) {
    projected p;
    project_one(cam, cp, w2c, x[i], y[i], z[i], &p);
    if (!keep_rule(cp, &p, map) || (hull && !hull[i])) continue;
    if (p.pixel < 0) continue;
    uint8_t r = 0, g = 0, b = 0;
    if (image) {
      const uint8_t *px = image + (int64_t)p.pixel * 3;
      b = px[0];
      g = px[1];
      r = px[2];
    }
    uint16_t mv = 0;
    if (mask) {
      mv = mask[p.pixel]; /* grayImg.at<uchar>(v,u), :803 */
      if (mv == 255) {     /* :805-810 */
        r = 255;
        g = 0;
        b = 0;
      }
    }
    if (out_index) out_index[m] = (int32_t)i;
    if (out_rgb) {
      out_rgb[3 * m + 0] = r;
      out_rgb[3 * m + 1] = g;
      out_rgb[3 * m + 2] = b;
    }
    if (out_mask) out_mask[m] = mv;
    if (out_xyz_cam) {
      out_xyz_cam[3 * m + 0] = p.xc;
      out_xyz_cam[3 * m + 1] = p.yc;
      out_xyz_cam[3 * m + 2] = p.zc;
    }
    if (out_xyz_world) { /* transformPointCloud(c2w), :549,555 */
      xform_point(c2w, p.xc, p.yc, p.zc, &out_xyz_world[3 * m + 0], &out_xyz_world[3 * m + 1],
                  &out_xyz_world[3 * m + 2]);
    }
    ++m;
  }
  free(hull);
  free(map);
  return m;
}

/* ------------------------------------------------------------------ */
/* f4: generateColorMap's 8-bit BGR -> HSV -> BGR round trip            */
/* ------------------------------------------------------------------ */

/* PCP/src/PointCloudProcessor.cpp:722-741: cv::cvtColor(rgb, hsv, COLOR_BGR2HSV); S and V times
 * saturation_scale / brightness_scale (1.0, :728-729) through saturate_cast<uchar>;
 * cv::cvtColor(hsv, out, COLOR_HSV2BGR).  [upstream OpenCV 4.2.0 (osrf/ros:noetic),
 * imgproc/src/color_hsv.simd.hpp: RGB2HSV_b (integer, exact) and HSV2RGB_b -> HSV2RGB_native, the
 * scalar fp32 routine without FMA; OpenCV's SIMD build of the backward half re-associates
 * (v - v*s instead of v*(1 - s)) and may differ by one level on some pixels: parity unpinned.]
 * cvRound = round half to even (lrint under the default rounding mode). */
static inline uint8_t sat_u8_f(float x) {
  long v = lrintf(x);
  return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

void orc_hsv_round_trip(const uint8_t *bgr_in, uint8_t *bgr_out, int64_t n_pixels, float saturation_scale,
                        float brightness_scale) {
  enum { hsv_shift = 12 };
  int sdiv_table[256], hdiv_table180[256];
  sdiv_table[0] = hdiv_table180[0] = 0;
  for (int i = 1; i < 256; i++) {
    sdiv_table[i] = (int)lrint((255 << hsv_shift) / (1. * i));
    hdiv_table180[i] = (int)lrint((180 << hsv_shift) / (6. * i));
  }
  static const int sector_data[][3] = {{1, 3, 0}, {1, 0, 2}, {3, 0, 1}, {0, 2, 1}, {0, 1, 3}, {2, 1, 0}};
  const float hscale = 6.f / 180.f;
  for (int64_t k = 0; k < n_pixels; ++k) {
    /* RGB2HSV_b::operator(), blueIdx = 0, hrange = 180 */
    const int b = bgr_in[3 * k + 0], g = bgr_in[3 * k + 1], r = bgr_in[3 * k + 2];
    int h, s, v = b, vmin = b, vr, vg;
    if (g > v) v = g;
    if (r > v) v = r;
    if (g < vmin) vmin = g;
    if (r < vmin) vmin = r;
    const int diff = v - vmin; /* saturate_cast<uchar>(v - vmin): already in range */
    vr = v == r ? -1 : 0;
    vg = v == g ? -1 : 0;
    s = (diff * sdiv_table[v] + (1 << (hsv_shift - 1))) >> hsv_shift;
    h = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
    h = (h * hdiv_table180[diff] + (1 << (hsv_shift - 1))) >> hsv_shift;
    h += h < 0 ? 180 : 0;
    const uint8_t H = (uint8_t)(h < 0 ? 0 : (h > 255 ? 255 : h));
    /* :733-734 */
    const uint8_t S = sat_u8_f((float)(uint8_t)s * saturation_scale);
    const uint8_t V = sat_u8_f((float)(uint8_t)v * brightness_scale);
    /* HSV2RGB_b: buf = {h, s/255, v/255}; HSV2RGB_native */
    float fh = (float)H;
    const float fs = (float)S * (1.0f / 255.0f), fv = (float)V * (1.0f / 255.0f);
    float fb, fg, fr;
    if (fs == 0) {
      fb = fg = fr = fv;
    } else {
      float tab[4];
      fh *= hscale;
      fh = fmodf(fh, 6.f);
      int sector = (int)floorf(fh);
      fh -= (float)sector;
      if ((unsigned)sector >= 6u) {
        sector = 0;
        fh = 0.f;
      }
      tab[0] = fv;
      tab[1] = fv * (1.f - fs);
      tab[2] = fv * (1.f - fs * fh);
      tab[3] = fv * (1.f - fs * (1.f - fh));
      fb = tab[sector_data[sector][0]];
      fg = tab[sector_data[sector][1]];
      fr = tab[sector_data[sector][2]];
    }
    bgr_out[3 * k + 0] = sat_u8_f(fb * 255.0f);
    bgr_out[3 * k + 1] = sat_u8_f(fg * 255.0f);
    bgr_out[3 * k + 2] = sat_u8_f(fr * 255.0f);
  }
}

/* ------------------------------------------------------------------ */
/* keyframes                                                           */
/* ------------------------------------------------------------------ */

/* markKeyframe PCP/include/PointCloudProcessor.hpp:151-191 (distance-only rule,
 * B9) driven by selectKeyframes PCP/src/PointCloudProcessor.cpp:1050-1075. */
int32_t orc_select_keyframes(const orc_pose *poses, int32_t n, double dist_threshold, int32_t *out_indices) {
  int32_t m = 0, last = -1;
  for (int32_t i = 0; i < n; ++i) {
    int key = 0;
    if (last < 0) {
      key = 1;
    } else {
      const double dx = poses[i].x - poses[last].x;
      const double dy = poses[i].y - poses[last].y;
      const double dz = poses[i].z - poses[last].z;
      key = sqrt(dx * dx + dy * dy + dz * dz) >= dist_threshold;
    }
    if (key) {
      out_indices[m++] = i;
      last = i;
    }
  }
  return m;
}
