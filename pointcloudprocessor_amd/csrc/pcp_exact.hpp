// pcp_exact.hpp -- orientation predicate of the hidden-point-removal hull (pcp_hpr.hip), decided exactly.
//
// orient3d(a, b, c, d) = sign of det [a - d; b - d; c - d]: > 0 iff d lies below the plane through a, b, c when
// they appear counter-clockwise from above.  Two stages: a floating-point evaluation with a forward error bound
// (`orient3d_filtered`, the only stage the per-candidate kernel uses), and an evaluation in expansion arithmetic
// (sums of non-overlapping doubles; two_sum / two_prod building blocks, Grow-Expansion, Scale-Expansion) whose sign
// is the sign of the real determinant whatever the conditioning.  [Shewchuk 1997, "Adaptive Precision Floating-Point
// Arithmetic and Fast Robust Geometric Predicates": the published algorithms, restated.]  The exact stage keeps its
// expansions in private arrays (scratch memory): it runs for a handful of points per keyframe.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace pcp {

struct Vec3d {
  double x, y, z;
};

// floating-point determinant and its permanent; certain iff |det| > 8e-16 * permanent ((7 + 56 eps) eps = 7.77e-16 is
// the bound of this evaluation order [Shewchuk 1997, orient3d stage A]; every operation individually rounded or fused
// -- the bound covers both)
__device__ __forceinline__ double orient3d_det(const Vec3d &a, const Vec3d &b, const Vec3d &c, const Vec3d &d,
                                               double &permanent) {
  const double adx = a.x - d.x, ady = a.y - d.y, adz = a.z - d.z;
  const double bdx = b.x - d.x, bdy = b.y - d.y, bdz = b.z - d.z;
  const double cdx = c.x - d.x, cdy = c.y - d.y, cdz = c.z - d.z;
  const double bdxcdy = bdx * cdy, cdxbdy = cdx * bdy;
  const double cdxady = cdx * ady, adxcdy = adx * cdy;
  const double adxbdy = adx * bdy, bdxady = bdx * ady;
  permanent = (fabs(bdxcdy) + fabs(cdxbdy)) * fabs(adz) + (fabs(cdxady) + fabs(adxcdy)) * fabs(bdz) +
              (fabs(adxbdy) + fabs(bdxady)) * fabs(cdz);
  return adz * (bdxcdy - cdxbdy) + bdz * (cdxady - adxcdy) + cdz * (adxbdy - bdxady);
}

// +1 / -1 when the sign is certain, 0 when the filter cannot tell (which includes a determinant that is exactly 0)
__device__ __forceinline__ int orient3d_filtered(const Vec3d &a, const Vec3d &b, const Vec3d &c, const Vec3d &d) {
  double perm;
  const double det = orient3d_det(a, b, c, d, perm);
  const double err = 8.0e-16 * perm;
  return det > err ? 1 : (-det > err ? -1 : 0);
}

namespace exact {

__device__ __forceinline__ void two_sum(double a, double b, double &x, double &y) {
  const double s = a + b;
  const double bv = s - a;
  const double av = s - bv;
  y = (a - av) + (b - bv);
  x = s;
}

__device__ __forceinline__ void two_diff(double a, double b, double &x, double &y) {
  const double s = a - b;
  const double bv = a - s;
  const double av = s + bv;
  y = (a - av) + (bv - b);
  x = s;
}

__device__ __forceinline__ void two_prod(double a, double b, double &x, double &y) {
  const double p = a * b;
  y = __builtin_fma(a, b, -p);
  x = p;
}

// h = e + f; h may alias e (room for elen + flen components)
__device__ inline int expansion_sum(int elen, const double *e, int flen, const double *f, double *h) {
  if (h != e)
    for (int i = 0; i < elen; ++i) h[i] = e[i];
  int hlen = elen;
  for (int j = 0; j < flen; ++j) {
    double q = f[j];
    for (int i = j; i < hlen; ++i) {
      double s, r;
      two_sum(q, h[i], s, r);
      h[i] = r;
      q = s;
    }
    h[hlen++] = q;
  }
  return hlen;
}

// h = e * b, 2 * elen components
__device__ inline int scale_expansion(int elen, const double *e, double b, double *h) {
  double q, t, T, s, r;
  two_prod(e[0], b, q, h[0]);
  int k = 1;
  for (int i = 1; i < elen; ++i) {
    two_prod(e[i], b, T, t);
    two_sum(q, t, s, r);
    h[k++] = r;
    two_sum(T, s, q, r);
    h[k++] = r;
  }
  h[k++] = q;
  return k;
}

__device__ inline int expansion_product(int elen, const double *e, int flen, const double *f, double *h, double *tmp) {
  int hlen = 0;
  for (int j = 0; j < flen; ++j) {
    const int tl = scale_expansion(elen, e, f[j], tmp);
    if (hlen == 0) {
      for (int i = 0; i < tl; ++i) h[i] = tmp[i];
      hlen = tl;
    } else {
      hlen = expansion_sum(hlen, h, tl, tmp, h);
    }
  }
  return hlen;
}

}  // namespace exact

// the sign of the real determinant
__device__ inline int orient3d_exact(const Vec3d &a, const Vec3d &b, const Vec3d &c, const Vec3d &d) {
  using namespace exact;
  double A[3][2], B[3][2], C[3][2];  // (low, high) parts of a - d, b - d, c - d
  two_diff(a.x, d.x, A[0][1], A[0][0]);
  two_diff(a.y, d.y, A[1][1], A[1][0]);
  two_diff(a.z, d.z, A[2][1], A[2][0]);
  two_diff(b.x, d.x, B[0][1], B[0][0]);
  two_diff(b.y, d.y, B[1][1], B[1][0]);
  two_diff(b.z, d.z, B[2][1], B[2][0]);
  two_diff(c.x, d.x, C[0][1], C[0][0]);
  two_diff(c.y, d.y, C[1][1], C[1][0]);
  two_diff(c.z, d.z, C[2][1], C[2][0]);
  double m1[8], m2[8], minor[16], term[64], tmp[64], acc[192];
  int acclen = 0;
  // det = A.z (B.x C.y - C.x B.y) + B.z (C.x A.y - A.x C.y) + C.z (A.x B.y - B.x A.y)
  for (int t = 0; t < 3; ++t) {
    double(*P)[2] = t == 0 ? A : (t == 1 ? B : C);
    double(*Q)[2] = t == 0 ? B : (t == 1 ? C : A);
    double(*R)[2] = t == 0 ? C : (t == 1 ? A : B);
    const int l1 = expansion_product(2, Q[0], 2, R[1], m1, tmp);
    const int l2 = expansion_product(2, R[0], 2, Q[1], m2, tmp);
    for (int i = 0; i < l2; ++i) m2[i] = -m2[i];
    const int lm = expansion_sum(l1, m1, l2, m2, minor);
    const int lt = expansion_product(lm, minor, 2, P[2], term, tmp);
    if (acclen == 0) {
      for (int i = 0; i < lt; ++i) acc[i] = term[i];
      acclen = lt;
    } else {
      acclen = expansion_sum(acclen, acc, lt, term, acc);
    }
  }
  for (int i = acclen - 1; i >= 0; --i) {
    if (acc[i] > 0.0) return 1;
    if (acc[i] < 0.0) return -1;
  }
  return 0;
}

// the sign of n . (q - p), every operation exact (six exact products of the exact differences, summed as expansions)
__device__ inline int dot_diff_sign_exact(const Vec3d &n, const Vec3d &q, const Vec3d &p) {
  using namespace exact;
  double acc[12], term[2];
  int len = 0;
  const double nn[3] = {n.x, n.y, n.z}, qq[3] = {q.x, q.y, q.z}, pp[3] = {p.x, p.y, p.z};
  for (int k = 0; k < 3; ++k) {
    double hi, lo;
    two_diff(qq[k], pp[k], hi, lo);
    two_prod(nn[k], lo, term[1], term[0]);
    len = len == 0 ? (acc[0] = term[0], acc[1] = term[1], 2) : expansion_sum(len, acc, 2, term, acc);
    two_prod(nn[k], hi, term[1], term[0]);
    len = expansion_sum(len, acc, 2, term, acc);
  }
  for (int i = len - 1; i >= 0; --i) {
    if (acc[i] > 0.0) return 1;
    if (acc[i] < 0.0) return -1;
  }
  return 0;
}

// filter, then the exact evaluation; *used_exact (nullable) counts the latter
__device__ inline int orient3d_sign(const Vec3d &a, const Vec3d &b, const Vec3d &c, const Vec3d &d, int *used_exact) {
  const int s = orient3d_filtered(a, b, c, d);
  if (s != 0) return s;
  if (used_exact) ++*used_exact;
  return orient3d_exact(a, b, c, d);
}

}  // namespace pcp
