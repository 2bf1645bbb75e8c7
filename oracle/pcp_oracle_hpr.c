/*
 * pcp_oracle_hpr.c -- CPU restatement of ViewCulling::hidden_points_removal
 * (PCP/src/vlcal/calib/view_culling.cpp:266-334), the cull the reference binary runs (:46).
 *
 * TEST INFRASTRUCTURE ONLY, PARITY UNPINNED (see pcp_oracle.h).
 *
 * The reference flips every candidate about a sphere of radius 90000 (:291-292,
 * view_culling.hpp:14), appends the origin (:297) and asks qhull ("qhull ", defaults) for the
 * convex hull; visible = the hull's vertices other than the origin (:316-329).  qhull is a
 * third-party library that is not under /root/reference (cloned at HEAD by the reference's
 * Dockerfile:21-27, unpinned) and not in this image as a C library, so the hull is restated
 * here from its definition: a candidate is visible iff its flipped point is an EXTREME POINT
 * of conv(flipped points + origin).  qhull computes the same set up to its round-off
 * treatment: with the default options (no Qt / QJ; C-0 merging in 3-d) a point within
 * ~DISTround (1e-10 m at these magnitudes) of a facet spanned by others is a "coplanar
 * point", not a vertex.  This file decides every orientation test exactly (floating-point
 * filter, then expansion arithmetic after Shewchuk 1997, "Adaptive Precision Floating-Point
 * Arithmetic and Fast Robust Geometric Predicates" -- the algorithms are restated, no code is
 * taken), so its vertex set is the exact one; tests compare it with scipy's bundled qhull_r
 * and list the points within the stated tolerance of the hull on which the two differ.
 *
 * Algorithm: quickhull with conflict lists (Barber, Dobkin, Huhdanpaa 1996 -- the published
 * algorithm qhull implements), triangular facets, exact predicates, points exactly on a
 * facet's plane treated as not above it (counted in stats).  Exact duplicates: only the
 * lowest index of a group of identical flipped points can be a vertex.
 */
#define _GNU_SOURCE /* qsort_r */
#include "pcp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* exact orientation predicate                                         */
/* ------------------------------------------------------------------ */

static inline void two_sum(double a, double b, double *x, double *y) {
  const double s = a + b;
  const double bv = s - a;
  const double av = s - bv;
  *y = (a - av) + (b - bv);
  *x = s;
}

static inline void two_diff(double a, double b, double *x, double *y) {
  const double s = a - b;
  const double bv = a - s;
  const double av = s + bv;
  *y = (a - av) + (bv - b);
  *x = s;
}

static inline void two_prod(double a, double b, double *x, double *y) {
  const double p = a * b;
  *y = fma(a, b, -p); /* exact error of the rounded product */
  *x = p;
}

/* expansions: arrays of doubles, non-overlapping, increasing magnitude; value = their exact sum */

/* h = e + f (Expansion-Sum); h may alias e when h has room for elen + flen components */
static int expansion_sum(int elen, const double *e, int flen, const double *f, double *h) {
  if (h != e) memcpy(h, e, (size_t)elen * sizeof(double));
  int hlen = elen;
  for (int j = 0; j < flen; ++j) {
    /* grow h[j ..] by f[j]: the lower j components are already final */
    double q = f[j];
    for (int i = j; i < hlen; ++i) {
      double s, r;
      two_sum(q, h[i], &s, &r);
      h[i] = r;
      q = s;
    }
    h[hlen++] = q;
  }
  return hlen;
}

/* h = e * b (Scale-Expansion); 2 * elen components */
static int scale_expansion(int elen, const double *e, double b, double *h) {
  double q, t, T, s, r;
  two_prod(e[0], b, &q, &h[0]);
  int k = 1;
  for (int i = 1; i < elen; ++i) {
    two_prod(e[i], b, &T, &t);
    two_sum(q, t, &s, &r);
    h[k++] = r;
    two_sum(T, s, &q, &r);
    h[k++] = r;
  }
  h[k++] = q;
  return k;
}

/* h = e * f for short expansions */
static int expansion_product(int elen, const double *e, int flen, const double *f, double *h, double *tmp) {
  int hlen = 0;
  for (int j = 0; j < flen; ++j) {
    const int tl = scale_expansion(elen, e, f[j], tmp);
    if (hlen == 0) {
      memcpy(h, tmp, (size_t)tl * sizeof(double));
      hlen = tl;
    } else {
      hlen = expansion_sum(hlen, h, tl, tmp, h);
    }
  }
  return hlen;
}

static void negate_expansion(int n, double *e) {
  for (int i = 0; i < n; ++i) e[i] = -e[i];
}

static int expansion_sign(int n, const double *e) {
  for (int i = n - 1; i >= 0; --i) {
    if (e[i] > 0.0) return 1;
    if (e[i] < 0.0) return -1;
  }
  return 0;
}

/* sign of det [a-d; b-d; c-d], every step exact */
static int orient3d_exact(const double *a, const double *b, const double *c, const double *d) {
  double A[3][2], B[3][2], Cc[3][2]; /* (lo, hi) of a-d, b-d, c-d per axis */
  for (int k = 0; k < 3; ++k) {
    two_diff(a[k], d[k], &A[k][1], &A[k][0]);
    two_diff(b[k], d[k], &B[k][1], &B[k][0]);
    two_diff(c[k], d[k], &Cc[k][1], &Cc[k][0]);
  }
  double m1[8], m2[8], minor[16], term[64], tmp[64], acc[192], tmp2[64];
  int acclen = 0;
  /* det = A.z (B.x C.y - C.x B.y) + B.z (C.x A.y - A.x C.y) + C.z (A.x B.y - B.x A.y) */
  double(*rows[3][3])[2] = {{A, B, Cc}, {B, Cc, A}, {Cc, A, B}};
  for (int t = 0; t < 3; ++t) {
    double(*P)[2] = rows[t][0], (*Q)[2] = rows[t][1], (*R)[2] = rows[t][2];
    const int l1 = expansion_product(2, Q[0], 2, R[1], m1, tmp2); /* Q.x R.y */
    const int l2 = expansion_product(2, R[0], 2, Q[1], m2, tmp2); /* R.x Q.y */
    negate_expansion(l2, m2);
    const int lm = expansion_sum(l1, m1, l2, m2, minor);
    const int lt = expansion_product(lm, minor, 2, P[2], term, tmp);
    if (acclen == 0) {
      memcpy(acc, term, (size_t)lt * sizeof(double));
      acclen = lt;
    } else {
      acclen = expansion_sum(acclen, acc, lt, term, acc);
    }
  }
  return expansion_sign(acclen, acc);
}

typedef struct hull_stats {
  int64_t filtered;   /* orientation tests decided by the floating-point filter */
  int64_t exact;      /* ... that needed the exact evaluation */
  int64_t zero;       /* ... whose exact value is zero (four coplanar points) */
} hull_stats;

/* Shewchuk's orientation: > 0 iff d is below the plane through a, b, c (counter-clockwise seen from above).
 * *approx receives the floating-point determinant (for "furthest point" choices only). */
static int orient3d(const double *a, const double *b, const double *c, const double *d, double *approx,
                    hull_stats *st) {
  const double adx = a[0] - d[0], ady = a[1] - d[1], adz = a[2] - d[2];
  const double bdx = b[0] - d[0], bdy = b[1] - d[1], bdz = b[2] - d[2];
  const double cdx = c[0] - d[0], cdy = c[1] - d[1], cdz = c[2] - d[2];
  const double bdxcdy = bdx * cdy, cdxbdy = cdx * bdy;
  const double cdxady = cdx * ady, adxcdy = adx * cdy;
  const double adxbdy = adx * bdy, bdxady = bdx * ady;
  const double det = adz * (bdxcdy - cdxbdy) + bdz * (cdxady - adxcdy) + cdz * (adxbdy - bdxady);
  const double permanent = (fabs(bdxcdy) + fabs(cdxbdy)) * fabs(adz) + (fabs(cdxady) + fabs(adxcdy)) * fabs(bdz) +
                           (fabs(adxbdy) + fabs(bdxady)) * fabs(cdz);
  if (approx) *approx = det;
  const double errbound = 8.0e-16 * permanent; /* (7 + 56 eps) eps = 7.77e-16 [Shewchuk 1997, orient3d stage A] */
  if (det > errbound) {
    if (st) st->filtered++;
    return 1;
  }
  if (-det > errbound) {
    if (st) st->filtered++;
    return -1;
  }
  const int s = orient3d_exact(a, b, c, d);
  if (st) {
    st->exact++;
    if (s == 0) st->zero++;
  }
  return s;
}

int orc_orient3d(const double a[3], const double b[3], const double c[3], const double d[3], int32_t exact_only) {
  return exact_only ? orient3d_exact(a, b, c, d) : orient3d(a, b, c, d, NULL, NULL);
}

/* ------------------------------------------------------------------ */
/* quickhull                                                           */
/* ------------------------------------------------------------------ */

typedef struct facet {
  int32_t v[3];     /* counter-clockwise seen from outside */
  int32_t nb[3];    /* nb[i]: the facet across edge (v[i], v[(i+1)%3]) */
  int32_t out_head; /* conflict list (points strictly above this facet), -1 = empty */
  int32_t far_pt;
  double far_d;     /* normalised height of far_pt above the plane (approximate; choice only) */
  double inv_len;   /* 1 / |(v1-v0) x (v2-v0)| (approximate) */
  int32_t mark;
  uint8_t alive;
} facet;

typedef struct hull {
  const double *p; /* n x 3 */
  int64_t n;
  facet *f;
  int32_t nf, cap;
  int32_t *next; /* conflict-list links */
  hull_stats st;
} hull;

static int32_t new_facet(hull *h, int32_t a, int32_t b, int32_t c) {
  if (h->nf == h->cap) {
    h->cap = h->cap * 2 + 64;
    h->f = (facet *)realloc(h->f, (size_t)h->cap * sizeof(facet));
  }
  facet *g = &h->f[h->nf];
  g->v[0] = a;
  g->v[1] = b;
  g->v[2] = c;
  g->nb[0] = g->nb[1] = g->nb[2] = -1;
  g->out_head = -1;
  g->far_pt = -1;
  g->far_d = 0.0;
  g->mark = -1;
  g->alive = 1;
  const double *A = h->p + 3 * (int64_t)a, *B = h->p + 3 * (int64_t)b, *C = h->p + 3 * (int64_t)c;
  const double ux = B[0] - A[0], uy = B[1] - A[1], uz = B[2] - A[2];
  const double vx = C[0] - A[0], vy = C[1] - A[1], vz = C[2] - A[2];
  const double nx = uy * vz - uz * vy, ny = uz * vx - ux * vz, nz = ux * vy - uy * vx;
  const double len = sqrt(nx * nx + ny * ny + nz * nz);
  g->inv_len = len > 0.0 ? 1.0 / len : 1.0;
  return h->nf++;
}

/* strictly above (outside) facet g?  *height: approximate normalised height */
static inline int above(hull *h, const facet *g, int32_t q, double *height) {
  double det;
  const int s = orient3d(h->p + 3 * (int64_t)g->v[0], h->p + 3 * (int64_t)g->v[1], h->p + 3 * (int64_t)g->v[2],
                         h->p + 3 * (int64_t)q, &det, &h->st);
  if (height) *height = -det * g->inv_len;
  return s < 0;
}

static inline void push_point(hull *h, facet *g, int32_t q, double height) {
  h->next[q] = g->out_head;
  g->out_head = q;
  if (g->far_pt < 0 || height > g->far_d) {
    g->far_pt = q;
    g->far_d = height;
  }
}

static int cmp_point_lex(const void *a, const void *b, void *ctx) {
  const double *p = (const double *)ctx;
  const int32_t i = *(const int32_t *)a, j = *(const int32_t *)b;
  for (int k = 0; k < 3; ++k) {
    const double x = p[3 * (int64_t)i + k], y = p[3 * (int64_t)j + k];
    if (x != y) return x < y ? -1 : 1;
  }
  return i < j ? -1 : (i > j);
}

/* Extreme points of conv(points): is_vertex[i] = 1 iff point i is a vertex of the hull (exact arithmetic; see the
 * header of this file for coplanar points and duplicates).  stats (nullable, 4 values): filtered tests, exact tests,
 * exact zeros, duplicates removed.  Returns the number of vertices, or -1 (fewer than 4 points / all points
 * coplanar: qhull fails on such input and the reference then returns no visible point, view_culling.cpp:307-312),
 * or -2 (allocation failure). */
int64_t orc_convex_hull_vertices(const double *points, int64_t n, uint8_t *is_vertex, int64_t *stats) {
  if (stats) stats[0] = stats[1] = stats[2] = stats[3] = 0;
  if (is_vertex && n > 0) memset(is_vertex, 0, (size_t)n);
  if (n < 4 || n > INT32_MAX - 8) return -1;
  hull H;
  memset(&H, 0, sizeof H);
  H.p = points;
  H.n = n;
  H.next = (int32_t *)malloc((size_t)n * sizeof(int32_t));
  int32_t *order = (int32_t *)malloc((size_t)n * sizeof(int32_t));
  uint8_t *dup = (uint8_t *)calloc((size_t)n, 1);
  int32_t *start_of = (int32_t *)malloc((size_t)n * sizeof(int32_t));
  int32_t *end_of = (int32_t *)malloc((size_t)n * sizeof(int32_t));
  int32_t *stack = NULL, *visible = NULL, *horizon = NULL, *fresh = NULL, *work = NULL;
  int64_t result = -2;
  if (!H.next || !order || !dup || !start_of || !end_of) goto done;
  /* duplicates: the lowest index of a group of identical points stands for the group */
  int64_t n_dup = 0;
  for (int64_t i = 0; i < n; ++i) order[i] = (int32_t)i;
  qsort_r(order, (size_t)n, sizeof(int32_t), cmp_point_lex, (void *)points);
  for (int64_t k = 1; k < n; ++k) {
    const double *a = points + 3 * (int64_t)order[k - 1], *b = points + 3 * (int64_t)order[k];
    if (a[0] == b[0] && a[1] == b[1] && a[2] == b[2]) {
      dup[order[k]] = 1; /* sorted by index inside a group: order[k-1] < order[k] */
      ++n_dup;
    }
  }
  if (stats) stats[3] = n_dup;
  /* initial simplex: i0 lowest x, i1 farthest from it, i2 farthest from their line, i3 farthest from their plane */
  int32_t i0 = -1, i1 = -1, i2 = -1, i3 = -1;
  for (int64_t i = 0; i < n; ++i)
    if (!dup[i] && (i0 < 0 || points[3 * i] < points[3 * (int64_t)i0])) i0 = (int32_t)i;
  double best = -1.0;
  const double *P0 = points + 3 * (int64_t)i0;
  for (int64_t i = 0; i < n; ++i) {
    if (dup[i]) continue;
    const double dx = points[3 * i] - P0[0], dy = points[3 * i + 1] - P0[1], dz = points[3 * i + 2] - P0[2];
    const double d2 = dx * dx + dy * dy + dz * dz;
    if (d2 > best) {
      best = d2;
      i1 = (int32_t)i;
    }
  }
  result = -1;
  if (i1 < 0 || !(best > 0.0)) goto done;
  const double *P1 = points + 3 * (int64_t)i1;
  const double ex = P1[0] - P0[0], ey = P1[1] - P0[1], ez = P1[2] - P0[2];
  best = -1.0;
  for (int64_t i = 0; i < n; ++i) {
    if (dup[i]) continue;
    const double dx = points[3 * i] - P0[0], dy = points[3 * i + 1] - P0[1], dz = points[3 * i + 2] - P0[2];
    const double cx = ey * dz - ez * dy, cy = ez * dx - ex * dz, cz = ex * dy - ey * dx;
    const double a2 = cx * cx + cy * cy + cz * cz;
    if (a2 > best) {
      best = a2;
      i2 = (int32_t)i;
    }
  }
  if (i2 < 0 || !(best > 0.0)) goto done;
  const double *P2 = points + 3 * (int64_t)i2;
  best = -1.0;
  for (int64_t i = 0; i < n; ++i) {
    if (dup[i]) continue;
    double det;
    orient3d(P0, P1, P2, points + 3 * i, &det, NULL);
    if (fabs(det) > best) {
      best = fabs(det);
      i3 = (int32_t)i;
    }
  }
  int s3 = i3 >= 0 ? orient3d(P0, P1, P2, points + 3 * (int64_t)i3, NULL, &H.st) : 0;
  if (s3 == 0) { /* the floating-point maximum is exactly coplanar: look for any point that is not */
    for (int64_t i = 0; i < n && s3 == 0; ++i) {
      if (dup[i]) continue;
      s3 = orient3d(P0, P1, P2, points + 3 * i, NULL, &H.st);
      if (s3 != 0) i3 = (int32_t)i;
    }
    if (s3 == 0) goto done; /* flat input */
  }
  result = -2;
  /* s3 > 0: i3 below plane (i0, i1, i2) counter-clockwise from above, so (i0, i1, i2) faces away from i3 */
  int32_t a = i0, b = i1, c = i2, d = i3;
  if (s3 < 0) {
    int32_t t = b;
    b = c;
    c = t;
  }
  {
    const int32_t f0 = new_facet(&H, a, b, c), f1 = new_facet(&H, a, d, b), f2 = new_facet(&H, b, d, c),
                  f3 = new_facet(&H, c, d, a);
    if (!H.f) goto done;
    /* neighbours: nb[i] across (v[i], v[i+1]) */
    H.f[f0].nb[0] = f1; /* a b */
    H.f[f0].nb[1] = f2; /* b c */
    H.f[f0].nb[2] = f3; /* c a */
    H.f[f1].nb[0] = f3; /* a d */
    H.f[f1].nb[1] = f2; /* d b */
    H.f[f1].nb[2] = f0; /* b a */
    H.f[f2].nb[0] = f1; /* b d */
    H.f[f2].nb[1] = f3; /* d c */
    H.f[f2].nb[2] = f0; /* c b */
    H.f[f3].nb[0] = f2; /* c d */
    H.f[f3].nb[1] = f1; /* d a */
    H.f[f3].nb[2] = f0; /* a c */
  }
  for (int64_t i = 0; i < n; ++i) {
    if (dup[i] || i == i0 || i == i1 || i == i2 || i == i3) continue;
    for (int k = 0; k < 4; ++k) {
      double hgt;
      if (above(&H, &H.f[k], (int32_t)i, &hgt)) {
        push_point(&H, &H.f[k], (int32_t)i, hgt);
        break;
      }
    }
  }
  int32_t stack_cap = 1024, stack_n = 0, vis_cap = 1024, hor_cap = 1024, fresh_cap = 1024, work_cap = 1024;
  stack = (int32_t *)malloc((size_t)stack_cap * sizeof(int32_t));
  visible = (int32_t *)malloc((size_t)vis_cap * sizeof(int32_t));
  horizon = (int32_t *)malloc((size_t)hor_cap * 3 * sizeof(int32_t));
  fresh = (int32_t *)malloc((size_t)fresh_cap * sizeof(int32_t));
  work = (int32_t *)malloc((size_t)work_cap * sizeof(int32_t));
  if (!stack || !visible || !horizon || !fresh || !work) goto done;
  for (int k = 0; k < 4; ++k)
    if (H.f[k].out_head >= 0) stack[stack_n++] = k;
  int32_t stamp = 0;
#define GROW(arr, cap, need, width)                                                     \
  if ((need) > (cap)) {                                                                 \
    (cap) = (need) * 2;                                                                 \
    (arr) = (int32_t *)realloc((arr), (size_t)(cap) * (width) * sizeof(int32_t));       \
    if (!(arr)) goto done;                                                              \
  }
  while (stack_n > 0) {
    const int32_t fi = stack[--stack_n];
    if (!H.f[fi].alive || H.f[fi].out_head < 0) continue;
    const int32_t e = H.f[fi].far_pt;
    ++stamp;
    /* visible region: flood fill from fi over facets e is strictly above */
    int32_t nvis = 0, nwork = 0, nhor = 0;
    H.f[fi].mark = stamp;
    work[nwork++] = fi;
    while (nwork > 0) {
      const int32_t g = work[--nwork];
      GROW(visible, vis_cap, nvis + 1, 1);
      visible[nvis++] = g;
      for (int k = 0; k < 3; ++k) {
        const int32_t nbk = H.f[g].nb[k];
        if (H.f[nbk].mark == stamp) continue; /* already known visible */
        if (H.f[nbk].mark == -stamp - 2 || !above(&H, &H.f[nbk], e, NULL)) {
          H.f[nbk].mark = -stamp - 2; /* known not visible in this round */
          GROW(horizon, hor_cap, nhor + 1, 3);
          horizon[3 * nhor + 0] = H.f[g].v[k];
          horizon[3 * nhor + 1] = H.f[g].v[(k + 1) % 3];
          horizon[3 * nhor + 2] = nbk;
          ++nhor;
        } else {
          H.f[nbk].mark = stamp;
          GROW(work, work_cap, nwork + 1, 1);
          work[nwork++] = nbk;
        }
      }
    }
    /* cone of new facets (u, v, e) over the horizon edges u -> v */
    GROW(fresh, fresh_cap, nhor, 1);
    for (int32_t k = 0; k < nhor; ++k) {
      const int32_t u = horizon[3 * k], v = horizon[3 * k + 1], behind = horizon[3 * k + 2];
      const int32_t nfi = new_facet(&H, u, v, e);
      if (!H.f) goto done;
      fresh[k] = nfi;
      H.f[nfi].nb[0] = behind;
      for (int j = 0; j < 3; ++j) /* the facet behind the horizon now borders the new one across (v, u) */
        if (H.f[behind].v[j] == v && H.f[behind].v[(j + 1) % 3] == u) H.f[behind].nb[j] = nfi;
      start_of[u] = nfi;
      end_of[v] = nfi;
    }
    for (int32_t k = 0; k < nhor; ++k) {
      const int32_t u = horizon[3 * k], v = horizon[3 * k + 1];
      H.f[fresh[k]].nb[1] = start_of[v]; /* across (v, e): the new facet whose horizon edge starts at v */
      H.f[fresh[k]].nb[2] = end_of[u];   /* across (e, u): the one whose horizon edge ends at u */
    }
    /* conflict lists of the visible facets go to the new facets (or inside the hull) */
    for (int32_t k = 0; k < nvis; ++k) {
      facet *g = &H.f[visible[k]];
      int32_t q = g->out_head;
      while (q >= 0) {
        const int32_t nq = H.next[q];
        if (q != e) {
          for (int32_t j = 0; j < nhor; ++j) {
            double hgt;
            if (above(&H, &H.f[fresh[j]], q, &hgt)) {
              push_point(&H, &H.f[fresh[j]], q, hgt);
              break;
            }
          }
        }
        q = nq;
      }
      g->out_head = -1;
      g->alive = 0;
    }
    for (int32_t k = 0; k < nhor; ++k)
      if (H.f[fresh[k]].out_head >= 0) {
        GROW(stack, stack_cap, stack_n + 1, 1);
        stack[stack_n++] = fresh[k];
      }
  }
#undef GROW
  result = 0;
  for (int32_t k = 0; k < H.nf; ++k) {
    if (!H.f[k].alive) continue;
    for (int j = 0; j < 3; ++j) {
      if (is_vertex && !is_vertex[H.f[k].v[j]]) {
        is_vertex[H.f[k].v[j]] = 1;
        ++result;
      }
    }
  }
  if (stats) {
    stats[0] = H.st.filtered;
    stats[1] = H.st.exact;
    stats[2] = H.st.zero;
  }
done:
  free(H.next);
  free(H.f);
  free(order);
  free(dup);
  free(start_of);
  free(end_of);
  free(stack);
  free(visible);
  free(horizon);
  free(fresh);
  free(work);
  return result;
}

/* ------------------------------------------------------------------ */
/* hidden_points_removal                                               */
/* ------------------------------------------------------------------ */

/* view_culling.cpp:291-292: pt_norm = pt.head<3>().norm(); flipped = (pt + 2.0 * (max_z - pt_norm) * pt / pt_norm).head<3>()
 * on the fp64-promoted camera coordinates (:27-38).  Eigen evaluates the expression per coefficient:
 * x + ((2.0 * (R - norm)) * x) / norm; norm of a 3-vector block = sqrt((x^2 + y^2) + z^2) [upstream Eigen 3.3.7 Redux.h,
 * the same reduction as the z-buffer routine's range, DESIGN.md section 3]. */
void orc_hpr_flip(const float *xc, const float *yc, const float *zc, int64_t m, double flip_radius, double *out_flipped) {
  for (int64_t i = 0; i < m; ++i) {
    const double X = (double)xc[i], Y = (double)yc[i], Z = (double)zc[i];
    const double norm = sqrt((X * X + Y * Y) + Z * Z);
    const double s = 2.0 * (flip_radius - norm);
    out_flipped[3 * i + 0] = X + (s * X) / norm;
    out_flipped[3 * i + 1] = Y + (s * Y) / norm;
    out_flipped[3 * i + 2] = Z + (s * Z) / norm;
  }
}

/* ViewCulling::hidden_points_removal for one keyframe over the whole cloud: out_keep[i] = 1 iff point i is a candidate
 * (z > 0 and the truncated pixel inside the cull size, :276-288) and its flipped image is a hull vertex (:316-329).
 * Fewer than 3 candidates, or a flat candidate set: qhull fails and the reference returns no point (:307-312).
 * stats (nullable, 5 values): candidates, filtered tests, exact tests, exact zeros, duplicates.  Returns the number
 * kept, or -2 on allocation failure.  Output order: the reference lists the visible points in qhull's vertex-list
 * order; a keep mask in input order is what the drop-in boundary hands on (pcp_hip.h). */
int64_t orc_hpr_frame(const orc_camera *cam, const float w2c[12], const float *x, const float *y, const float *z,
                      int64_t n, double flip_radius, uint8_t *out_keep, int64_t *stats) {
  orc_cull_params cp;
  orc_default_cull_params(&cp);
  cp.cull_mode = ORC_CULL_HPR_CANDIDATES;
  if (stats) stats[0] = stats[1] = stats[2] = stats[3] = stats[4] = 0;
  if (out_keep && n > 0) memset(out_keep, 0, (size_t)n);
  int32_t *cell = (int32_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int32_t));
  float *xc = (float *)malloc((size_t)(n > 0 ? n : 1) * 3 * sizeof(float));
  if (!cell || !xc) {
    free(cell);
    free(xc);
    return -2;
  }
  float *yc = xc + n, *zc = yc + n;
  orc_project_frame(cam, &cp, w2c, x, y, z, n, cell, NULL, NULL, xc, yc, zc);
  int64_t m = 0;
  for (int64_t i = 0; i < n; ++i)
    if (cell[i] != -1) {
      xc[m] = xc[i];
      yc[m] = yc[i];
      zc[m] = zc[i];
      cell[m] = (int32_t)i; /* m <= i: compaction in place */
      ++m;
    }
  if (stats) stats[0] = m;
  int64_t kept = 0;
  if (m >= 3) {
    /* the m flipped candidates in xyz triples, then the origin (:297) */
    double *fl = (double *)malloc((size_t)(m + 1) * 3 * sizeof(double));
    uint8_t *vtx = (uint8_t *)malloc((size_t)(m + 1));
    if (!fl || !vtx) {
      free(fl);
      free(vtx);
      free(cell);
      free(xc);
      return -2;
    }
    orc_hpr_flip(xc, yc, zc, m, flip_radius, fl);
    fl[3 * m] = fl[3 * m + 1] = fl[3 * m + 2] = 0.0;
    int64_t hs[4];
    const int64_t nv = orc_convex_hull_vertices(fl, m + 1, vtx, hs);
    if (stats) {
      stats[1] = hs[0];
      stats[2] = hs[1];
      stats[3] = hs[2];
      stats[4] = hs[3];
    }
    if (nv == -2) kept = -2;
    if (nv >= 0)
      for (int64_t k = 0; k < m; ++k)
        if (vtx[k]) {
          if (out_keep) out_keep[cell[k]] = 1;
          ++kept;
        }
    free(fl);
    free(vtx);
  }
  free(cell);
  free(xc);
  return kept;
}
