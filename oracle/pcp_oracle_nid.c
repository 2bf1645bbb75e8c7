/*
 * pcp_oracle_nid.c -- CPU restatement of the NID extrinsic-refinement cost
 * (SURVEY.md section 8 f1).  See pcp_oracle.h: TEST INFRASTRUCTURE ONLY, PARITY UNPINNED.
 *
 * Restates NIDCost::operator() (PCP/include/vlcal/costs/nid_cost.hpp:42-116) and the sum
 * and domain limit of MultiNIDCost (PCP/src/vlcal/calib/visual_camera_calibration.cpp:86-129).
 * The reference differentiates with ceres::Jet<double,7> through Sophus::Manifold<SE3>;
 * what the optimiser consumes is the gradient in the SE(3) tangent of T * exp(delta),
 * delta = (upsilon, omega) [upstream Sophus / Ceres].  Here the same quantity is obtained
 * with forward-mode dual numbers of dimension 6 seeded at delta = 0:
 *   d(pt_camera)/d upsilon = R,   d(pt_camera)/d omega = -R [p]x.
 *
 * Reference accident reproduced: the image is a 3-channel BGR8 image
 * converted to CV_64FC3 and read with at<double>(y, x), i.e. element x of row y of the
 * interleaved B,G,R doubles: pixel x / 3, channel x % 3 (visual_camera_calibration.cpp:171-173,
 * nid_cost.hpp:87).
 */
#include "pcp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ND 6

typedef struct jet {
  double a;
  double v[ND];
} jet;

static jet jc(double a) {
  jet r;
  r.a = a;
  memset(r.v, 0, sizeof(r.v));
  return r;
}
static jet jadd(jet x, jet y) {
  jet r;
  r.a = x.a + y.a;
  for (int k = 0; k < ND; ++k) r.v[k] = x.v[k] + y.v[k];
  return r;
}
static jet jsub(jet x, jet y) {
  jet r;
  r.a = x.a - y.a;
  for (int k = 0; k < ND; ++k) r.v[k] = x.v[k] - y.v[k];
  return r;
}
static jet jmul(jet x, jet y) {
  jet r;
  r.a = x.a * y.a;
  for (int k = 0; k < ND; ++k) r.v[k] = x.a * y.v[k] + x.v[k] * y.a;
  return r;
}
static jet jdiv(jet x, jet y) {
  jet r;
  r.a = x.a / y.a;
  for (int k = 0; k < ND; ++k) r.v[k] = (x.v[k] - r.a * y.v[k]) / y.a;
  return r;
}
static jet jscale(double s, jet x) {
  jet r;
  r.a = s * x.a;
  for (int k = 0; k < ND; ++k) r.v[k] = s * x.v[k];
  return r;
}
static jet jlog(jet x) {
  jet r;
  r.a = log(x.a);
  for (int k = 0; k < ND; ++k) r.v[k] = x.v[k] / x.a;
  return r;
}

/* pinhole + plumb-bob on jets (PCP/include/camera/pinhole.hpp:13-51) */
static void project_jet(const orc_camera *c, const jet p[3], jet *u, jet *v) {
  const jet xn = jdiv(p[0], p[2]), yn = jdiv(p[1], p[2]);
  const jet x2 = jmul(xn, xn), y2 = jmul(yn, yn);
  const jet r2 = jadd(x2, y2), r4 = jmul(r2, r2), r6 = jmul(r2, r4);
  const jet rc = jadd(jadd(jadd(jc(1.0), jscale(c->k1, r2)), jscale(c->k2, r4)), jscale(c->k3, r6));
  const jet t1 = jmul(jscale(2.0, xn), yn);
  const jet t2 = jadd(r2, jscale(2.0, x2)), t3 = jadd(r2, jscale(2.0, y2));
  const jet xd = jadd(jadd(jmul(rc, xn), jscale(c->p1, t1)), jscale(c->p2, t2));
  const jet yd = jadd(jadd(jmul(rc, yn), jscale(c->p1, t3)), jscale(c->p2, t1));
  *u = jadd(jscale(c->fx, xd), jc(c->cx));
  *v = jadd(jscale(c->fy, yd), jc(c->cy));
}

/* NIDCost::operator() for one keyframe.  pts: camera-frame x y z (fp32) and intensity of the
 * culled cloud; image: BGR8, cam->image_height x cam->image_width.  Returns 0 when the
 * cost is not finite (the functor returns false). */
static int nid_one(const orc_camera *cam, const uint8_t *image, const float *x, const float *y, const float *z,
                   const float *intensity, int64_t n, const double T[16], int bins, jet *out) {
  const int W = cam->image_width, H = cam->image_height;
  /* cubic B-spline basis (nid_cost.hpp:35-39) */
  static const double C[4][4] = {{1.0 / 6, -3.0 / 6, 3.0 / 6, -1.0 / 6},
                                 {4.0 / 6, 0.0, -6.0 / 6, 3.0 / 6},
                                 {1.0 / 6, 3.0 / 6, 3.0 / 6, -3.0 / 6},
                                 {0.0, 0.0, 0.0, 1.0 / 6}};
  jet *hist = (jet *)calloc((size_t)bins * bins, sizeof(jet));
  jet *hist_image = (jet *)calloc((size_t)bins, sizeof(jet));
  double *hist_points = (double *)calloc((size_t)bins, sizeof(double));
  for (int64_t i = 0; i < n; ++i) {
    const double p[3] = {(double)x[i], (double)y[i], (double)z[i]};
    jet pc[3];
    for (int r = 0; r < 3; ++r) {
      const double *R = T + 4 * r;
      pc[r].a = (R[0] * p[0] + R[1] * p[1] + R[2] * p[2]) + R[3];
      /* d/d upsilon = R */
      pc[r].v[0] = R[0];
      pc[r].v[1] = R[1];
      pc[r].v[2] = R[2];
      /* d/d omega = R d(omega x p)/d omega, omega x p = (wy pz - wz py, wz px - wx pz, wx py - wy px):
       * d/d wx = (0, -pz, py), d/d wy = (pz, 0, -px), d/d wz = (-py, px, 0) */
      pc[r].v[3] = -R[1] * p[2] + R[2] * p[1];
      pc[r].v[4] = R[0] * p[2] - R[2] * p[0];
      pc[r].v[5] = -R[0] * p[1] + R[1] * p[0];
    }
    int bin_points = (int)((double)intensity[i] * bins);
    if (bin_points > bins - 1) bin_points = bins - 1;
    if (bin_points < 0) bin_points = 0;
    jet u, v;
    project_jet(cam, pc, &u, &v);
    if (!isfinite(u.a) || !isfinite(v.a) || fabs(u.a) > 1e9 || fabs(v.a) > 1e9) continue;
    const int kx = (int)floor(u.a), ky = (int)floor(v.a);
    if (kx < 0 || ky < 0 || kx >= W || ky >= H) continue; /* outlier */
    hist_points[bin_points] += 1.0;
    const jet s[2] = {jsub(u, jc((double)kx)), jsub(v, jc((double)ky))};
    jet beta[4][2];
    for (int d = 0; d < 2; ++d) {
      const jet s2 = jmul(s[d], s[d]), s3 = jmul(s2, s[d]);
      for (int r = 0; r < 4; ++r)
        beta[r][d] = jadd(jadd(jadd(jc(C[r][0]), jscale(C[r][1], s[d])), jscale(C[r][2], s2)), jscale(C[r][3], s3));
    }
    for (int a = 0; a < 4; ++a) {
      int px = kx - 1 + a;
      px = px < 0 ? 0 : (px > W - 1 ? W - 1 : px);
      for (int b = 0; b < 4; ++b) {
        int py = ky - 1 + b;
        py = py < 0 ? 0 : (py > H - 1 ? H - 1 : py);
        const jet w = jmul(beta[a][0], beta[b][1]);
        /* normalized_image.at<double>(py, px) on a CV_64FC3 matrix: interleaved element px of row py */
        const double pix = (double)image[((int64_t)py * W + px / 3) * 3 + px % 3] / 255.0;
        int bin_image = (int)(pix * bins);
        if (bin_image > bins - 1) bin_image = bins - 1;
        hist[bin_image * bins + bin_points] = jadd(hist[bin_image * bins + bin_points], w);
        hist_image[bin_image] = jadd(hist_image[bin_image], w);
      }
    }
  }
  double sum = 0.0;
  for (int b = 0; b < bins; ++b) sum += hist_points[b];
  jet H_image = jc(0.0), H_ip = jc(0.0);
  double H_points = 0.0;
  for (int b = 0; b < bins; ++b) {
    const jet h = jscale(1.0 / sum, hist_image[b]);
    H_image = jsub(H_image, jmul(h, jlog(jadd(h, jc(1e-6)))));
    const double hp = hist_points[b] / sum;
    H_points -= hp * log(hp + 1e-6);
  }
  for (int b = 0; b < bins * bins; ++b) {
    const jet h = jscale(1.0 / sum, hist[b]);
    H_ip = jsub(H_ip, jmul(h, jlog(jadd(h, jc(1e-6)))));
  }
  const jet MI = jsub(jadd(H_image, jc(H_points)), H_ip);
  *out = jdiv(jsub(H_ip, MI), H_ip);
  free(hist);
  free(hist_image);
  free(hist_points);
  return isfinite(out->a) ? 1 : 0;
}

/* MultiNIDCost: sum over keyframes.  Per keyframe k the culled cloud is
 * x/y/z/intensity[offsets[k] .. offsets[k+1]) and the image images[k].  T = T_camera_lidar
 * (4x4 row-major).  out_grad: 6 doubles (d/d upsilon, d/d omega), nullable.
 * Returns 1 on success, 0 when any keyframe's cost is not finite. */
int orc_nid(const orc_camera *cam, const uint8_t *const *images, int32_t n_frames, const int64_t *offsets,
            const float *x, const float *y, const float *z, const float *intensity, const double T[16], int32_t bins,
            double *out_cost, double *out_grad) {
  jet total = jc(0.0);
  int ok = 1;
  for (int32_t k = 0; k < n_frames; ++k) {
    jet r;
    const int64_t b = offsets[k], e = offsets[k + 1];
    if (!nid_one(cam, images[k], x + b, y + b, z + b, intensity + b, e - b, T, bins, &r)) {
      ok = 0;
      continue;
    }
    total = jadd(total, r);
  }
  if (out_cost) *out_cost = total.a;
  if (out_grad)
    for (int k = 0; k < ND; ++k) out_grad[k] = total.v[k];
  return ok;
}
