/*
 * pcp_oracle_mls.c -- CPU restatement of the enableMLS path: PCL 1.10
 * MovingLeastSquares (radius search + order-2 polynomial fit, SIMPLE
 * projection, NONE and VOXEL_GRID_DILATION upsampling) and
 * StatisticalOutlierRemoval.  See pcp_oracle.h: TEST INFRASTRUCTURE ONLY,
 * PARITY UNPINNED.
 *
 * The reference only configures and calls PCL (PCP/src/cloudSmooth.cpp:109-164,
 * parameters PCP/src/PointCloudProcessor.cpp:67-86); the arithmetic lives in
 * PCL 1.10.0 (libpcl from osrf/ros:noetic, /root/reference/Dockerfile:2), which
 * is not vendored.  Everything below restates PCL's published algorithm
 * (surface/include/pcl/surface/impl/mls.hpp, common/impl/centroid.hpp,
 * common/impl/eigen.hpp, filters/impl/statistical_outlier_removal.hpp,
 * FLANN 1.9.1 L2_Simple) and is marked [upstream].
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
/* uniform grid for fixed-radius / kNN queries (replaces pcl::KdTreeFLANN;
 * results are defined by the distance predicate, not by the tree)        */
/* ------------------------------------------------------------------ */

typedef struct grid {
  float minx, miny, minz;
  float inv_cell;
  int32_t nx, ny, nz;
  int64_t *cell_start; /* ncell+1 */
  int32_t *order;      /* point ids sorted by cell */
} grid;

static inline int32_t clampi(int32_t v, int32_t lo, int32_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

static inline void grid_coords(const grid *g, float x, float y, float z, int32_t *ix, int32_t *iy, int32_t *iz) {
  *ix = clampi((int32_t)floorf((x - g->minx) * g->inv_cell), 0, g->nx - 1);
  *iy = clampi((int32_t)floorf((y - g->miny) * g->inv_cell), 0, g->ny - 1);
  *iz = clampi((int32_t)floorf((z - g->minz) * g->inv_cell), 0, g->nz - 1);
}

static int grid_build(grid *g, const float *x, const float *y, const float *z, int64_t n, float cell) {
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int64_t i = 0; i < n; ++i) {
    if (x[i] < mn[0]) mn[0] = x[i];
    if (y[i] < mn[1]) mn[1] = y[i];
    if (z[i] < mn[2]) mn[2] = z[i];
    if (x[i] > mx[0]) mx[0] = x[i];
    if (y[i] > mx[1]) mx[1] = y[i];
    if (z[i] > mx[2]) mx[2] = z[i];
  }
  if (n == 0) mn[0] = mn[1] = mn[2] = mx[0] = mx[1] = mx[2] = 0.0f;
  /* bound the cell count: grow the cell until the table fits */
  for (;;) {
    const double ex = (double)(mx[0] - mn[0]) / cell + 1.0, ey = (double)(mx[1] - mn[1]) / cell + 1.0,
                 ez = (double)(mx[2] - mn[2]) / cell + 1.0;
    if (ex * ey * ez <= 2.0e8) break;
    cell *= 2.0f;
  }
  g->minx = mn[0];
  g->miny = mn[1];
  g->minz = mn[2];
  g->inv_cell = 1.0f / cell;
  g->nx = (int32_t)floorf((mx[0] - mn[0]) * g->inv_cell) + 1;
  g->ny = (int32_t)floorf((mx[1] - mn[1]) * g->inv_cell) + 1;
  g->nz = (int32_t)floorf((mx[2] - mn[2]) * g->inv_cell) + 1;
  const int64_t ncell = (int64_t)g->nx * g->ny * g->nz;
  g->cell_start = (int64_t *)calloc((size_t)ncell + 1, sizeof(int64_t));
  g->order = (int32_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int32_t));
  if (!g->cell_start || !g->order) return -1;
  for (int64_t i = 0; i < n; ++i) {
    int32_t ix, iy, iz;
    grid_coords(g, x[i], y[i], z[i], &ix, &iy, &iz);
    g->cell_start[((int64_t)iz * g->ny + iy) * g->nx + ix + 1]++;
  }
  for (int64_t c = 0; c < ncell; ++c) g->cell_start[c + 1] += g->cell_start[c];
  int64_t *cursor = (int64_t *)malloc((size_t)ncell * sizeof(int64_t));
  if (!cursor) return -1;
  memcpy(cursor, g->cell_start, (size_t)ncell * sizeof(int64_t));
  for (int64_t i = 0; i < n; ++i) {
    int32_t ix, iy, iz;
    grid_coords(g, x[i], y[i], z[i], &ix, &iy, &iz);
    g->order[cursor[((int64_t)iz * g->ny + iy) * g->nx + ix]++] = (int32_t)i;
  }
  free(cursor);
  return 0;
}

static void grid_free(grid *g) {
  free(g->cell_start);
  free(g->order);
}

/* FLANN L2_Simple<float>: result += diff*diff, dims in order, fp32 [upstream]. */
static inline float sqdist_f32(float ax, float ay, float az, float bx, float by, float bz) {
  const float dx = ax - bx, dy = ay - by, dz = az - bz;
  return (dx * dx + dy * dy) + dz * dz;
}

typedef struct nbr {
  float d;
  int32_t i;
} nbr;

static int nbr_cmp(const void *a, const void *b) {
  const nbr *p = (const nbr *)a, *q = (const nbr *)b;
  if (p->d < q->d) return -1;
  if (p->d > q->d) return 1;
  return (p->i > q->i) - (p->i < q->i);
}

/* pcl::KdTreeFLANN::radiusSearch: dist < f32(radius*radius) strictly, sorted
 * ascending [upstream kdtree_flann.hpp + flann RadiusResultSet]. */
static int64_t radius_search(const grid *g, const float *x, const float *y, const float *z, float qx, float qy,
                             float qz, float radius, float sq_radius, nbr **buf, int64_t *cap) {
  const int32_t reach = (int32_t)ceilf(radius * g->inv_cell);
  int32_t ix, iy, iz;
  grid_coords(g, qx, qy, qz, &ix, &iy, &iz);
  int64_t m = 0;
  for (int32_t cz = clampi(iz - reach, 0, g->nz - 1); cz <= clampi(iz + reach, 0, g->nz - 1); ++cz)
    for (int32_t cy = clampi(iy - reach, 0, g->ny - 1); cy <= clampi(iy + reach, 0, g->ny - 1); ++cy) {
      const int64_t row = ((int64_t)cz * g->ny + cy) * g->nx;
      const int64_t b = g->cell_start[row + clampi(ix - reach, 0, g->nx - 1)];
      const int64_t e = g->cell_start[row + clampi(ix + reach, 0, g->nx - 1) + 1];
      for (int64_t k = b; k < e; ++k) {
        const int32_t j = g->order[k];
        const float d = sqdist_f32(x[j], y[j], z[j], qx, qy, qz);
        if (d < sq_radius) {
          if (m == *cap) {
            *cap = *cap ? *cap * 2 : 256;
            *buf = (nbr *)realloc(*buf, (size_t)*cap * sizeof(nbr));
          }
          (*buf)[m].d = d;
          (*buf)[m].i = j;
          ++m;
        }
      }
    }
  qsort(*buf, (size_t)m, sizeof(nbr), nbr_cmp);
  return m;
}

/* ------------------------------------------------------------------ */
/* pcl::eigen33 (smallest eigenpair) [upstream common/impl/eigen.hpp]     */
/* ------------------------------------------------------------------ */

static void compute_roots2(double b, double c, double roots[3]) {
  roots[0] = 0.0;
  double d = b * b - 4.0 * c;
  if (d < 0.0) d = 0.0;
  const double sd = sqrt(d);
  roots[2] = 0.5 * (b + sd);
  roots[1] = 0.5 * (b - sd);
}

static void swapd(double *a, double *b) {
  const double t = *a;
  *a = *b;
  *b = t;
}

static void compute_roots(const double m[9], double roots[3]) {
  const double c0 = m[0] * m[4] * m[8] + 2.0 * m[1] * m[2] * m[5] - m[0] * m[5] * m[5] - m[4] * m[2] * m[2] -
                    m[8] * m[1] * m[1];
  const double c1 = m[0] * m[4] - m[1] * m[1] + m[0] * m[8] - m[2] * m[2] + m[4] * m[8] - m[5] * m[5];
  const double c2 = m[0] + m[4] + m[8];
  if (fabs(c0) < DBL_EPSILON) {
    compute_roots2(c2, c1, roots);
    return;
  }
  const double s_inv3 = 1.0 / 3.0;
  const double s_sqrt3 = sqrt(3.0);
  const double c2_over_3 = c2 * s_inv3;
  double a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
  if (a_over_3 > 0.0) a_over_3 = 0.0;
  const double half_b = 0.5 * (c0 + c2_over_3 * (2.0 * c2_over_3 * c2_over_3 - c1));
  double q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
  if (q > 0.0) q = 0.0;
  const double rho = sqrt(-a_over_3);
  const double theta = atan2(sqrt(-q), half_b) * s_inv3;
  const double cos_theta = cos(theta);
  const double sin_theta = sin(theta);
  roots[0] = c2_over_3 + 2.0 * rho * cos_theta;
  roots[1] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
  roots[2] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
  if (roots[0] >= roots[1]) swapd(&roots[0], &roots[1]);
  if (roots[1] >= roots[2]) {
    swapd(&roots[1], &roots[2]);
    if (roots[0] >= roots[1]) swapd(&roots[0], &roots[1]);
  }
  if (roots[0] <= 0.0) compute_roots2(c2, c1, roots);
}

/* detail::getLargest3x3Eigenvector: cross products of the rows of (A - lambda I),
 * keep the longest, normalise [upstream]. */
static void largest_3x3_eigenvector(const double s[9], double v[3]) {
  const double r0[3] = {s[0], s[1], s[2]}, r1[3] = {s[3], s[4], s[5]}, r2[3] = {s[6], s[7], s[8]};
  double c[3][3];
  c[0][0] = r0[1] * r1[2] - r0[2] * r1[1];
  c[0][1] = r0[2] * r1[0] - r0[0] * r1[2];
  c[0][2] = r0[0] * r1[1] - r0[1] * r1[0];
  c[1][0] = r0[1] * r2[2] - r0[2] * r2[1];
  c[1][1] = r0[2] * r2[0] - r0[0] * r2[2];
  c[1][2] = r0[0] * r2[1] - r0[1] * r2[0];
  c[2][0] = r1[1] * r2[2] - r1[2] * r2[1];
  c[2][1] = r1[2] * r2[0] - r1[0] * r2[2];
  c[2][2] = r1[0] * r2[1] - r1[1] * r2[0];
  int best = 0;
  double len[3];
  for (int k = 0; k < 3; ++k) len[k] = (c[k][0] * c[k][0] + c[k][1] * c[k][1]) + c[k][2] * c[k][2];
  if (len[1] > len[best]) best = 1;
  if (len[2] > len[best]) best = 2;
  const double l = sqrt(len[best]);
  v[0] = c[best][0] / l;
  v[1] = c[best][1] / l;
  v[2] = c[best][2] / l;
}

static void eigen33_smallest(const double mat[9], double *eigenvalue, double eigenvector[3]) {
  double scale = 0.0;
  for (int k = 0; k < 9; ++k)
    if (fabs(mat[k]) > scale) scale = fabs(mat[k]);
  if (scale <= DBL_MIN) scale = 1.0;
  double s[9], roots[3];
  for (int k = 0; k < 9; ++k) s[k] = mat[k] / scale;
  compute_roots(s, roots);
  *eigenvalue = roots[0] * scale;
  s[0] -= roots[0];
  s[4] -= roots[0];
  s[8] -= roots[0];
  largest_3x3_eigenvector(s, eigenvector);
}

/* ------------------------------------------------------------------ */
/* MLSResult::computeMLSSurface + projectQueryPoint(SIMPLE) [upstream mls.hpp] */
/* ------------------------------------------------------------------ */

typedef struct mls_result {
  double mean[3], normal[3], u_axis[3], v_axis[3];
  double c_vec[6];
  double curvature;
  int32_t num_neighbors;
  int32_t order;
  int valid;
  int fitted;
} mls_result;

static inline double dot3(const double a[3], const double b[3]) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

/* 6x6 Cholesky (Eigen LLT, lower) solve in place; returns 0 if not positive definite. */
static int llt_solve6(double A[36], double b[6], int nc) {
  for (int j = 0; j < nc; ++j) {
    double d = A[j * nc + j];
    for (int k = 0; k < j; ++k) d -= A[j * nc + k] * A[j * nc + k];
    if (!(d > 0.0)) {
      /* Eigen's LLT carries on with sqrt of a non-positive pivot -> NaN; PCL then
       * sees a non-finite c_vec[0] and falls back to the plane projection. */
      for (int k = 0; k < nc; ++k) b[k] = NAN;
      return 0;
    }
    d = sqrt(d);
    A[j * nc + j] = d;
    for (int i = j + 1; i < nc; ++i) {
      double s = A[i * nc + j];
      for (int k = 0; k < j; ++k) s -= A[i * nc + k] * A[j * nc + k];
      A[i * nc + j] = s / d;
    }
  }
  for (int i = 0; i < nc; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= A[i * nc + k] * b[k];
    b[i] = s / A[i * nc + i];
  }
  for (int i = nc - 1; i >= 0; --i) {
    double s = b[i];
    for (int k = i + 1; k < nc; ++k) s -= A[k * nc + i] * b[k];
    b[i] = s / A[i * nc + i];
  }
  return 1;
}

static void mls_fit(const float *x, const float *y, const float *z, int32_t index, const nbr *nn, int64_t K,
                    double search_radius, int order, mls_result *r) {
  /* compute3DCentroid: fp64 sums of fp32 coordinates, divided by K */
  double c[3] = {0, 0, 0};
  for (int64_t k = 0; k < K; ++k) {
    c[0] += x[nn[k].i];
    c[1] += y[nn[k].i];
    c[2] += z[nn[k].i];
  }
  c[0] /= (double)K;
  c[1] /= (double)K;
  c[2] /= (double)K;
  /* computeCovarianceMatrix: un-normalised */
  double C[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int64_t k = 0; k < K; ++k) {
    const double px = (double)x[nn[k].i] - c[0], py = (double)y[nn[k].i] - c[1], pz = (double)z[nn[k].i] - c[2];
    C[4] += py * py;
    C[5] += py * pz;
    C[8] += pz * pz;
    C[0] += px * px;
    C[1] += px * py;
    C[2] += px * pz;
  }
  C[3] = C[1];
  C[6] = C[2];
  C[7] = C[5];
  double ev, n[3];
  eigen33_smallest(C, &ev, n);
  const double q[3] = {(double)x[index], (double)y[index], (double)z[index]};
  r->num_neighbors = (int32_t)K;
  r->order = order;
  r->fitted = 0;
  for (int k = 0; k < 6; ++k) r->c_vec[k] = 0.0;
  if (!isfinite(n[0]) || !isfinite(n[1]) || !isfinite(n[2])) {
    r->valid = 0;
    r->mean[0] = q[0];
    r->mean[1] = q[1];
    r->mean[2] = q[2];
    r->normal[0] = r->normal[1] = r->normal[2] = 0.0;
    r->curvature = 0.0;
    return;
  }
  r->valid = 1;
  const double d = -1.0 * dot3(n, c);
  const double distance = dot3(q, n) + d;
  for (int k = 0; k < 3; ++k) {
    r->mean[k] = q[k] - distance * n[k];
    r->normal[k] = n[k];
  }
  double curv = (C[0] + C[4]) + C[8];
  if (curv != 0.0) curv = fabs(ev / curv);
  r->curvature = curv;
  /* Eigen unitOrthogonal() for Vector3d [upstream OrthoMethods.h] */
  if (!(fabs(n[0]) <= fabs(n[2]) * 1e-12) || !(fabs(n[1]) <= fabs(n[2]) * 1e-12)) {
    const double invnm = 1.0 / sqrt(n[0] * n[0] + n[1] * n[1]);
    r->v_axis[0] = -n[1] * invnm;
    r->v_axis[1] = n[0] * invnm;
    r->v_axis[2] = 0.0;
  } else {
    const double invnm = 1.0 / sqrt(n[1] * n[1] + n[2] * n[2]);
    r->v_axis[0] = 0.0;
    r->v_axis[1] = -n[2] * invnm;
    r->v_axis[2] = n[1] * invnm;
  }
  r->u_axis[0] = n[1] * r->v_axis[2] - n[2] * r->v_axis[1];
  r->u_axis[1] = n[2] * r->v_axis[0] - n[0] * r->v_axis[2];
  r->u_axis[2] = n[0] * r->v_axis[1] - n[1] * r->v_axis[0];

  if (order > 1) {
    const int nc = (order + 1) * (order + 2) / 2;
    if (K >= nc && nc <= 6) {
      const double max_sq_radius = search_radius * search_radius; /* B13 */
      double A[36], b[6];
      memset(A, 0, sizeof(A));
      memset(b, 0, sizeof(b));
      for (int64_t k = 0; k < K; ++k) {
        const double de[3] = {(double)x[nn[k].i] - r->mean[0], (double)y[nn[k].i] - r->mean[1],
                              (double)z[nn[k].i] - r->mean[2]};
        const double w = exp(-dot3(de, de) / max_sq_radius);
        const double uc = dot3(de, r->u_axis), vc = dot3(de, r->v_axis), f = dot3(de, r->normal);
        double P[6];
        int j = 0;
        double u_pow = 1.0;
        for (int ui = 0; ui <= order; ++ui) {
          double v_pow = 1.0;
          for (int vi = 0; vi <= order - ui; ++vi) {
            P[j++] = u_pow * v_pow;
            v_pow *= vc;
          }
          u_pow *= uc;
        }
        for (int a = 0; a < nc; ++a) {
          const double pw = P[a] * w;
          for (int bb = 0; bb < nc; ++bb) A[a * nc + bb] += pw * P[bb];
          b[a] += pw * f;
        }
      }
      llt_solve6(A, b, nc);
      for (int k = 0; k < nc; ++k) r->c_vec[k] = b[k];
      r->fitted = 1;
    }
  }
}

/* MLSResult::projectQueryPoint(SIMPLE, required_neighbors) */
static void mls_project_query(const mls_result *r, int required, double pt[3], double nrm[3]) {
  if (r->order > 1 && r->num_neighbors >= required && r->fitted && isfinite(r->c_vec[0])) {
    for (int k = 0; k < 3; ++k) {
      pt[k] = r->mean[k] + r->c_vec[0] * r->normal[k];
      nrm[k] = r->normal[k] - r->c_vec[r->order + 1] * r->u_axis[k] - r->c_vec[1] * r->v_axis[k];
    }
    const double l = sqrt(dot3(nrm, nrm));
    if (l > 0.0) {
      nrm[0] /= l;
      nrm[1] /= l;
      nrm[2] /= l;
    }
  } else {
    for (int k = 0; k < 3; ++k) {
      pt[k] = r->mean[k];
      nrm[k] = r->normal[k];
    }
  }
}

/* MLSResult::projectPoint(pt, SIMPLE, required_neighbors) used by upsampling */
static void mls_project_point(const mls_result *r, const double p[3], int required, double pt[3], double nrm[3]) {
  const double de[3] = {p[0] - r->mean[0], p[1] - r->mean[1], p[2] - r->mean[2]};
  const double u = dot3(de, r->u_axis), v = dot3(de, r->v_axis);
  double w = 0.0;
  for (int k = 0; k < 3; ++k) nrm[k] = r->normal[k];
  if (r->order > 1 && r->num_neighbors >= required && r->fitted && isfinite(r->c_vec[0])) {
    /* getPolynomialPartialDerivative(u, v) */
    double d_z = 0, d_zu = 0, d_zv = 0;
    double u_pow[4], v_pow[4];
    int j = 0;
    u_pow[0] = v_pow[0] = 1.0;
    for (int ui = 0; ui <= r->order; ++ui) {
      for (int vi = 0; vi <= r->order - ui; ++vi) {
        d_z += u_pow[ui] * v_pow[vi] * r->c_vec[j];
        if (ui >= 1) d_zu += r->c_vec[j] * ui * u_pow[ui - 1] * v_pow[vi];
        if (vi >= 1) d_zv += r->c_vec[j] * vi * u_pow[ui] * v_pow[vi - 1];
        if (ui == 0 && vi < r->order) v_pow[vi + 1] = v_pow[vi] * v;
        ++j;
      }
      if (ui < r->order) u_pow[ui + 1] = u_pow[ui] * u;
    }
    w = d_z;
    for (int k = 0; k < 3; ++k) nrm[k] -= (d_zu * r->u_axis[k] + d_zv * r->v_axis[k]);
    const double l = sqrt(dot3(nrm, nrm));
    if (l > 0.0) {
      nrm[0] /= l;
      nrm[1] /= l;
      nrm[2] /= l;
    }
  }
  for (int k = 0; k < 3; ++k) pt[k] = r->mean[k] + u * r->u_axis[k] + v * r->v_axis[k] + w * r->normal[k];
}

/* pcl::MovingLeastSquares::process / performProcessing, NONE [upstream];
 * configured by PCP/src/cloudSmooth.cpp:124-154. */
static mls_result *mls_compute_all(const float *x, const float *y, const float *z, int64_t n,
                                   const orc_mls_params *p, uint8_t *has) {
  grid g;
  /* cell 0.1 % above r so that reach = 1 cell is safe against fp32 slop in the cell assignment */
  if (grid_build(&g, x, y, z, n, (float)p->search_radius * 1.001f) != 0) return NULL;
  mls_result *res = (mls_result *)calloc((size_t)(n > 0 ? n : 1), sizeof(mls_result));
  const float sq_radius = (float)(p->search_radius * p->search_radius);
  int nt = 1;
#ifdef _OPENMP
  nt = p->threads <= 0 ? omp_get_num_procs() : (p->threads > omp_get_num_procs() ? omp_get_num_procs() : p->threads);
#pragma omp parallel num_threads(nt)
#endif
  {
    nbr *buf = NULL;
    int64_t cap = 0;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1000)
#endif
    for (int64_t i = 0; i < n; ++i) {
      const int64_t K = radius_search(&g, x, y, z, x[i], y[i], z[i], (float)p->search_radius, sq_radius, &buf, &cap);
      has[i] = 0;
      if (K < 3) continue;
      mls_fit(x, y, z, (int32_t)i, buf, K, p->search_radius, p->polynomial_order, &res[i]);
      has[i] = 1;
    }
    free(buf);
  }
  (void)nt;
  grid_free(&g);
  return res;
}

int64_t orc_mls(const float *x, const float *y, const float *z, int64_t n, const orc_mls_params *p, float *out_xyz,
                float *out_normal, float *out_curv, int32_t *out_index) {
  uint8_t *has = (uint8_t *)malloc((size_t)(n > 0 ? n : 1));
  mls_result *res = mls_compute_all(x, y, z, n, p, has);
  if (!res) {
    free(has);
    return -1;
  }
  const int nr_coeff = (p->polynomial_order + 1) * (p->polynomial_order + 2) / 2;
  int64_t m = 0;
  for (int64_t i = 0; i < n; ++i) {
    if (!has[i]) continue;
    double pt[3], nrm[3];
    if (!res[i].valid) { /* invalid plane: keep the input point */
      pt[0] = res[i].mean[0];
      pt[1] = res[i].mean[1];
      pt[2] = res[i].mean[2];
      nrm[0] = nrm[1] = nrm[2] = 0.0;
    } else {
      mls_project_query(&res[i], nr_coeff, pt, nrm);
    }
    if (out_xyz) {
      out_xyz[3 * m + 0] = (float)pt[0];
      out_xyz[3 * m + 1] = (float)pt[1];
      out_xyz[3 * m + 2] = (float)pt[2];
    }
    if (out_normal) {
      out_normal[3 * m + 0] = (float)nrm[0];
      out_normal[3 * m + 1] = (float)nrm[1];
      out_normal[3 * m + 2] = (float)nrm[2];
    }
    if (out_curv) out_curv[m] = (float)res[i].curvature;
    if (out_index) out_index[m] = (int32_t)i;
    ++m;
  }
  free(res);
  free(has);
  return m;
}

/* ------------------------------------------------------------------ */
/* VOXEL_GRID_DILATION [upstream mls.hpp MLSVoxelGrid + performUpsampling] */
/* ------------------------------------------------------------------ */

static int u64_cmp(const void *a, const void *b) {
  const uint64_t p = *(const uint64_t *)a, q = *(const uint64_t *)b;
  return (p > q) - (p < q);
}

static int64_t sort_unique(uint64_t *k, int64_t n) {
  if (n == 0) return 0;
  qsort(k, (size_t)n, sizeof(uint64_t), u64_cmp);
  int64_t m = 1;
  for (int64_t i = 1; i < n; ++i)
    if (k[i] != k[m - 1]) k[m++] = k[i];
  return m;
}

/* origin / extent (nullable / <= 0): the voxel lattice of a LARGER cloud these points are a part of -- its bounding_min_ and
 * its largest extent (MLSVoxelGrid is laid over the bounding box of the cloud it is given) -- so that a region of a map can
 * be restated on the map's own lattice (tests of the streamed chain at full size); null: the points' own box, as PCL. */
static int64_t voxel_dilation_impl(const float *x, const float *y, const float *z, int64_t n, const orc_mls_params *p,
                                   const float *origin, double extent, int64_t capacity, float *out_xyz, float *out_normal,
                                   float *out_curv, int32_t *out_index) {
  if (n == 0) return 0;
  uint8_t *has = (uint8_t *)malloc((size_t)n);
  mls_result *res = mls_compute_all(x, y, z, n, p, has);
  if (!res) {
    free(has);
    return -1;
  }
  /* pcl::getMinMax3D -> bounding_min_/max_ (Vector4f) */
  float bmin[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, bmax[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int64_t i = 0; i < n; ++i) {
    if (x[i] < bmin[0]) bmin[0] = x[i];
    if (y[i] < bmin[1]) bmin[1] = y[i];
    if (z[i] < bmin[2]) bmin[2] = z[i];
    if (x[i] > bmax[0]) bmax[0] = x[i];
    if (y[i] > bmax[1]) bmax[1] = y[i];
    if (z[i] > bmax[2]) bmax[2] = z[i];
  }
  const float vs = p->vgd_voxel_size;
  const float sx = bmax[0] - bmin[0], sy = bmax[1] - bmin[1], sz = bmax[2] - bmin[2];
  double max_size = (double)fmaxf(fmaxf(sx, sy), sz);
  if (origin) {
    bmin[0] = origin[0];
    bmin[1] = origin[1];
    bmin[2] = origin[2];
    if (extent > 0.0) max_size = extent;
  }
  const uint64_t S = (uint64_t)(1.5 * max_size / vs);
  int64_t nk = n;
  uint64_t *keys = (uint64_t *)malloc((size_t)nk * sizeof(uint64_t));
  for (int64_t i = 0; i < n; ++i) {
    /* getCellIndex: (p[i] - bounding_min_[i]) / voxel_size_ in fp32, cast to int */
    const int32_t ix = (int32_t)((x[i] - bmin[0]) / vs), iy = (int32_t)((y[i] - bmin[1]) / vs),
                  iz = (int32_t)((z[i] - bmin[2]) / vs);
    keys[i] = (uint64_t)(int64_t)ix * S * S + (uint64_t)(int64_t)iy * S + (uint64_t)(int64_t)iz;
  }
  nk = sort_unique(keys, nk);
  for (int it = 0; it < p->vgd_iterations; ++it) {
    uint64_t *nw = (uint64_t *)malloc((size_t)nk * 27 * sizeof(uint64_t));
    int64_t m = 0;
    for (int64_t k = 0; k < nk; ++k) {
      /* getIndexIn3D */
      uint64_t id = keys[k];
      const int32_t ix = (int32_t)(S ? id / (S * S) : 0);
      id -= (uint64_t)(int64_t)ix * S * S;
      const int32_t iy = (int32_t)(S ? id / S : 0);
      id -= (uint64_t)(int64_t)iy * S;
      const int32_t iz = (int32_t)id;
      for (int dx = -1; dx <= 1; ++dx)
        for (int dy = -1; dy <= 1; ++dy)
          for (int dz = -1; dz <= 1; ++dz) {
            /* B16: PCL lets a negative neighbour index wrap in its uint64 key
             * arithmetic (MLSVoxelGrid::dilate + getIndexIn1D [upstream]), which
             * decodes to a voxel ~1.5 extents away and emits a few garbage points
             * next to the bounding-box minimum faces.  Normalised: such neighbours
             * are not created. */
            if (ix + dx < 0 || iy + dy < 0 || iz + dz < 0) continue;
            nw[m++] = (uint64_t)(int64_t)(ix + dx) * S * S + (uint64_t)(int64_t)(iy + dy) * S +
                      (uint64_t)(int64_t)(iz + dz);
          }
    }
    free(keys);
    keys = nw;
    nk = sort_unique(keys, m);
  }
  /* nearest input point per voxel centre (tree_->nearestKSearch(p, 1)) */
  grid g;
  grid_build(&g, x, y, z, n, (float)p->search_radius);
  const int nr_coeff = (p->polynomial_order + 1) * (p->polynomial_order + 2) / 2;
  int64_t m = 0;
  for (int64_t k = 0; k < nk; ++k) {
    uint64_t id = keys[k];
    const int32_t ix = (int32_t)(S ? id / (S * S) : 0);
    id -= (uint64_t)(int64_t)ix * S * S;
    const int32_t iy = (int32_t)(S ? id / S : 0);
    id -= (uint64_t)(int64_t)iy * S;
    const int32_t iz = (int32_t)id;
    /* getPosition: float(index) * voxel_size + bounding_min */
    const float px = (float)ix * vs + bmin[0], py = (float)iy * vs + bmin[1], pz = (float)iz * vs + bmin[2];
    /* expanding-ring nearest search on the grid */
    int32_t cx, cy, cz;
    grid_coords(&g, px, py, pz, &cx, &cy, &cz);
    int32_t best = -1;
    float bestd = FLT_MAX;
    const float cell = 1.0f / g.inv_cell;
    int32_t maxr = g.nx > g.ny ? g.nx : g.ny;
    if (g.nz > maxr) maxr = g.nz;
    for (int32_t ring = 0; ring <= maxr; ++ring) {
      if (best >= 0 && ring >= 1) {
        /* rings 0..ring-1 are done: every unvisited point is >= (ring-1)*cell away
         * (also when the query was clamped into the grid); 0.999 absorbs the fp32
         * slop of the cell assignment */
        const float reach = (float)(ring - 1) * cell * 0.999f;
        if (reach * reach > bestd) break;
      }
      for (int32_t zz = cz - ring; zz <= cz + ring; ++zz) {
        if (zz < 0 || zz >= g.nz) continue;
        for (int32_t yy = cy - ring; yy <= cy + ring; ++yy) {
          if (yy < 0 || yy >= g.ny) continue;
          const int shell_yz = (zz == cz - ring || zz == cz + ring || yy == cy - ring || yy == cy + ring);
          for (int32_t xx = cx - ring; xx <= cx + ring; ++xx) {
            if (xx < 0 || xx >= g.nx) continue;
            if (!shell_yz && xx != cx - ring && xx != cx + ring) continue;
            const int64_t c = ((int64_t)zz * g.ny + yy) * g.nx + xx;
            for (int64_t q = g.cell_start[c]; q < g.cell_start[c + 1]; ++q) {
              const int32_t j = g.order[q];
              const float d = sqdist_f32(x[j], y[j], z[j], px, py, pz);
              if (d < bestd || (d == bestd && j < best)) {
                bestd = d;
                best = j;
              }
            }
          }
        }
      }
    }
    if (best < 0) continue;
    if (!has[best] || !res[best].valid) continue; /* mls_results_[input_index].valid == false */
    const double add_point[3] = {(double)px, (double)py, (double)pz};
    double pt[3], nrm[3];
    mls_project_point(&res[best], add_point, 5 * nr_coeff, pt, nrm);
    if (out_xyz && m < capacity) {
      out_xyz[3 * m + 0] = (float)pt[0];
      out_xyz[3 * m + 1] = (float)pt[1];
      out_xyz[3 * m + 2] = (float)pt[2];
    }
    if (out_normal && m < capacity) {
      out_normal[3 * m + 0] = (float)nrm[0];
      out_normal[3 * m + 1] = (float)nrm[1];
      out_normal[3 * m + 2] = (float)nrm[2];
    }
    if (out_curv && m < capacity) out_curv[m] = (float)res[best].curvature;
    if (out_index && m < capacity) out_index[m] = best;
    ++m;
  }
  grid_free(&g);
  free(keys);
  free(res);
  free(has);
  return m;
}

int64_t orc_mls_voxel_dilation(const float *x, const float *y, const float *z, int64_t n, const orc_mls_params *p,
                               int64_t capacity, float *out_xyz, float *out_normal, float *out_curv,
                               int32_t *out_index) {
  return voxel_dilation_impl(x, y, z, n, p, NULL, 0.0, capacity, out_xyz, out_normal, out_curv, out_index);
}

int64_t orc_mls_voxel_dilation_part(const float *x, const float *y, const float *z, int64_t n, const orc_mls_params *p,
                                    const float *origin, double extent, int64_t capacity, float *out_xyz, float *out_normal,
                                    float *out_curv, int32_t *out_index) {
  return voxel_dilation_impl(x, y, z, n, p, origin, extent, capacity, out_xyz, out_normal, out_curv, out_index);
}

/* ------------------------------------------------------------------ */
/* StatisticalOutlierRemoval [upstream statistical_outlier_removal.hpp];
 * configured by PCP/src/cloudSmooth.cpp:109-116,160-164 (k=60, 0.7 sigma). */
/* ------------------------------------------------------------------ */

static void knn_push(nbr *heap, int *size, int k, float d, int32_t i) {
  /* max-heap of the k best (smallest) by (d, i) */
  if (*size < k) {
    int c = (*size)++;
    heap[c].d = d;
    heap[c].i = i;
    while (c > 0) {
      const int pnt = (c - 1) / 2;
      if (nbr_cmp(&heap[pnt], &heap[c]) >= 0) break;
      const nbr t = heap[pnt];
      heap[pnt] = heap[c];
      heap[c] = t;
      c = pnt;
    }
    return;
  }
  nbr cand = {d, i};
  if (nbr_cmp(&cand, &heap[0]) >= 0) return;
  heap[0] = cand;
  int c = 0;
  for (;;) {
    int l = 2 * c + 1, r = l + 1, big = c;
    if (l < k && nbr_cmp(&heap[l], &heap[big]) > 0) big = l;
    if (r < k && nbr_cmp(&heap[r], &heap[big]) > 0) big = r;
    if (big == c) break;
    const nbr t = heap[big];
    heap[big] = heap[c];
    heap[c] = t;
    c = big;
  }
}

int64_t orc_sor(const float *x, const float *y, const float *z, int64_t n, int32_t mean_k, double std_mul,
                uint8_t *out_keep, float *out_distance, double *out_threshold, int32_t threads) {
  if (n == 0) return 0;
  /* grid sized for ~mean_k points per 27-cell neighbourhood */
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int64_t i = 0; i < n; ++i) {
    if (x[i] < mn[0]) mn[0] = x[i];
    if (y[i] < mn[1]) mn[1] = y[i];
    if (z[i] < mn[2]) mn[2] = z[i];
    if (x[i] > mx[0]) mx[0] = x[i];
    if (y[i] > mx[1]) mx[1] = y[i];
    if (z[i] > mx[2]) mx[2] = z[i];
  }
  double vol = fmax((double)(mx[0] - mn[0]), 1e-3) * fmax((double)(mx[1] - mn[1]), 1e-3) * fmax((double)(mx[2] - mn[2]), 1e-3);
  float cell = (float)cbrt(vol / (double)n * 4.0);
  if (!(cell > 1e-4f)) cell = 1e-4f;
  grid g;
  if (grid_build(&g, x, y, z, n, cell) != 0) return -1;
  cell = 1.0f / g.inv_cell;
  const int k = mean_k + 1; /* nearestKSearch(point, mean_k_ + 1): first hit is the point itself */
  float *distances = (float *)malloc((size_t)n * sizeof(float));
  int nt = 1;
#ifdef _OPENMP
  nt = threads <= 0 ? omp_get_num_procs() : (threads > omp_get_num_procs() ? omp_get_num_procs() : threads);
#pragma omp parallel num_threads(nt)
#endif
  {
    nbr *heap = (nbr *)malloc((size_t)k * sizeof(nbr));
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1000)
#endif
    for (int64_t i = 0; i < n; ++i) {
      int size = 0;
      int32_t cx, cy, cz;
      grid_coords(&g, x[i], y[i], z[i], &cx, &cy, &cz);
      int32_t maxr = g.nx > g.ny ? g.nx : g.ny;
      if (g.nz > maxr) maxr = g.nz;
      for (int32_t ring = 0; ring <= maxr; ++ring) {
        if (size == k && ring >= 1) {
          /* every unvisited cell is at least (ring-1)*cell away from the query */
          const float reach = (float)(ring - 1) * cell * 0.999f;
          if (reach * reach > heap[0].d) break;
        }
        for (int32_t zz = cz - ring; zz <= cz + ring; ++zz) {
          if (zz < 0 || zz >= g.nz) continue;
          for (int32_t yy = cy - ring; yy <= cy + ring; ++yy) {
            if (yy < 0 || yy >= g.ny) continue;
            const int shell_yz = (zz == cz - ring || zz == cz + ring || yy == cy - ring || yy == cy + ring);
            for (int32_t xx = cx - ring; xx <= cx + ring; ++xx) {
              if (xx < 0 || xx >= g.nx) continue;
              if (!shell_yz && xx != cx - ring && xx != cx + ring) continue;
              const int64_t c = ((int64_t)zz * g.ny + yy) * g.nx + xx;
              for (int64_t q = g.cell_start[c]; q < g.cell_start[c + 1]; ++q) {
                const int32_t j = g.order[q];
                knn_push(heap, &size, k, sqdist_f32(x[j], y[j], z[j], x[i], y[i], z[i]), j);
              }
            }
          }
        }
      }
      qsort(heap, (size_t)size, sizeof(nbr), nbr_cmp);
      /* dist_sum over k = 1..mean_k of sqrt(nn_dists[k]) (float sqrt, double sum) */
      double dist_sum = 0.0;
      for (int q = 1; q < size; ++q) dist_sum += (double)sqrtf(heap[q].d);
      distances[i] = (float)(dist_sum / (double)mean_k);
    }
    free(heap);
  }
  (void)nt;
  double sum = 0.0, sq_sum = 0.0;
  for (int64_t i = 0; i < n; ++i) {
    sum += (double)distances[i];
    sq_sum += (double)(distances[i] * distances[i]);
  }
  const double mean = sum / (double)n;
  const double variance = (sq_sum - sum * sum / (double)n) / ((double)n - 1.0);
  const double stddev = sqrt(variance);
  const double threshold = mean + std_mul * stddev;
  if (out_threshold) *out_threshold = threshold;
  if (out_distance) memcpy(out_distance, distances, (size_t)n * sizeof(float));
  int64_t kept = 0;
  for (int64_t i = 0; i < n; ++i) {
    const int keep = !((double)distances[i] > threshold);
    if (out_keep) out_keep[i] = (uint8_t)keep;
    kept += keep;
  }
  free(distances);
  grid_free(&g);
  return kept;
}
