// pcp_hpr.hip -- ViewCulling::hidden_points_removal on the device (PCP/src/vlcal/calib/view_culling.cpp:266-334):
// the cull the reference binary runs (:46).  Candidates (z > 0, truncated pixel inside the cull size, :276-288) are
// flipped about a sphere of radius hidden_points_removal_max_z = 90000 (:291-292, view_culling.hpp:14), the origin is
// appended (:297) and the visible points are the vertices of the convex hull of that set other than the origin
// (qhull, :302-329).
//
// gfx950 only.  Build with -ffp-contract=off (the flip is the reference's arithmetic; the error bounds below hold with
// or without contraction).
//
// What is computed.  A candidate is visible iff its flipped point p is an EXTREME POINT of S = {flipped candidates} +
// {origin}: iff some plane through p has every other point of S strictly on one side.  That is a per-point question,
// and it is answered per point -- no hull data structure is built:
//
//   visible  <=>  exists n :  n . (q - p) < 0  for every q in S \ {p}                                   (1)
//   hidden   <=>  p lies in the simplex spanned by at most four other points of S (Caratheodory)          (2)
//
// One wavefront per candidate searches for a witness of (1) or (2) in floating point and then CHECKS it with a
// forward error bound; both kinds of witness are proofs, so a point the kernel decides is decided correctly whatever
// the search did.  The search: write n = e0 + s1 e1 + s2 e2 in a frame at p (e0 radial), so that (1) is a
// two-dimensional linear feasibility problem in s: every q contributes the half-plane s . D_q < E_q with
// D_q = ((q-p).e1, (q-p).e2), E_q = -(q-p).e0.  The wavefront keeps the feasible polygon of the half-planes seen so far
// (one vertex per lane), takes its vertex mean as the trial normal, tests the other points against it -- 64 at a time,
// one per lane -- and clips the polygon with every point the trial normal does not clear.  A trial normal that clears
// everything is a witness of (1); a polygon clipped to nothing names three half-planes with an empty intersection, and
// the tetrahedron (origin, q_a, q_b, q_c) of their points is the witness of (2).
//
// Locality.  All flipped points lie within metres of a sphere of radius ~1.8e5 m, inside the camera's cone.  A plane
// through p that is nearly tangent to that sphere leaves it by rho sep^2 / 2 at chordal distance sep, so only points
// within a few pixels of p can reach it.  Candidates are binned by their gnomonic coordinates (X/Z, Y/Z) into cells
// (about 8 points each, 8 x 8 cells to a coarse cell); a cell holds its centre direction u_c, a bound r_c of the chord
// between u_c and any direction inside, and an upper bound rho_c of the norms of its points.  For q in the cell
//     n . q = |q| |n| cos(n, q) <= rho_c |n| (1 - max(0, |n/|n| - u_c| - r_c)^2 / 2)
// and the cell is cleared when that is below n . p.  A wavefront tests 64 coarse cells, then the 64 cells of a coarse
// cell it could not clear, then the points of the cells that remain.
//
// Points neither witness settles (a trial normal within round-off of another point's plane, a tetrahedron test whose
// determinant the filter cannot sign: margins ~1e-12 m) go to k_hpr_exact, one wavefront each, which repeats the search
// over all candidates and checks its certificates with the exact orientation predicate (pcp_exact.hpp).  What even that
// cannot certify -- only exactly degenerate input (four coplanar points) can get there -- is counted (`unresolved`,
// pcp_hpr_stats) and classified hidden, as qhull classifies points on a facet ("coplanar points" are not vertices).
// Exact duplicates: the lowest input index of a group of identical flipped points stands for the group.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <type_traits>

#include "pcp_device.hpp"
#include "pcp_exact.hpp"
#include "pcp_scan.hpp"

namespace pcp {

constexpr int kHprBlock = 256;
constexpr int kHprCoarse = 8;           // fine cells per coarse cell edge: 64 fine cells = one wavefront of tests
constexpr int kHprMid = 4;              // ... per "mid" cell edge: 16 fine cells = one row of tests of the 16-lane passes
constexpr double kHprTargetPerCell = 8.0;
constexpr int64_t kHprMaxCells = int64_t(1) << 22;
constexpr int kHprMaxRestarts = 96;
constexpr int kStatStride = 32, kStatCopies = 64;  // counters: block 0 + kStatCopies copies of 32 words
// The tallies of the passes every candidate goes through cost what they count: an atomic or three per candidate from every
// wavefront of k_hpr_quick / k_hpr_radial / k_hpr_tilt were 11.5 of the 182 ms of kernels in a run of C3 and 9 % of the pass
// (device-scope atomics execute beyond the L2; profiles/r05_hpr_tally_ab.log).  Now: visible / hidden are counted ONCE, from
// the final states (k_hpr_writeback / k_hpr_set_bits, a ballot per wavefront); the work of the 16-lane passes (trial normals,
// batches of point tests) goes into word kStatPacked of the copies as ONE atomic per wavefront: normals in bits 0..23,
// batches above (a copy holds a 64th of one keyframe's tallies: < 2^24 normals).
constexpr int kStatPacked = 5;
constexpr int kStatPackedShift = 24;
constexpr double kHprBox = 1073741824.0;  // half-width of the initial box of trial normals (2^30 rad of tilt)
constexpr double kPointSlack = 1.0e-15;  // |fl(n . (q - p)) - exact| <= 4.44e-16 sum |n_i (q_i - p_i)| (see test_range)

enum : int32_t { kStHidden = 0, kStVisible = 1, kStUndecided = 2 };
enum : int { kSearchVisible = 1, kSearchEmpty = 2, kSearchFail = 3, kSearchHiddenDup = 4 };

struct HprGrid {
  double a0, b0, h, inv_h;
  int32_t gw, gh, cgw, cgh;
  int32_t mgw, mgh;         // mid cells (kHprMid x kHprMid fine cells): the upper level of the 16-lane passes
  double r_fine, r_coarse, r_mid;  // chord bound between a cell's centre direction and any direction inside it
  double a_reach;           // sqrt(1 + max |A|^2) over the grid: bounds 1 / u_z of every candidate direction
  const double *rho_max;    // device: upper bound of |q| over all candidates (k_hpr_cells)
  int32_t m;
};

struct HprArrays {
  const double *sx, *sy, *sz;   // flipped candidates, cell order
  const int32_t *sidx;          // input index of the candidate (duplicate rule)
  const int32_t *scell;         // fine cell of the candidate
  const int32_t *cstart;        // fine cells: first candidate (cells + 1 entries)
  const double *crho;           // fine cells: upper bound of |q| (0 = empty)
  const double *cdir;           // fine cells: centre direction, SoA x[cells] y[cells] z[cells]
  const double *Crho, *Cdir;    // coarse cells
  const unsigned long long *crep;  // fine cells: (upper bits of the largest |q| | place of that candidate), 0 = empty; nullable
  // the same bounds as ONE 16-byte record per cell, for the 16-lane passes (k_hpr_radial, k_hpr_tilt), whose time is the
  // round trips of their cell tests: centre direction rounded to fp32 (off by <= 1.1e-7 in chord: kCell4Slack is added to
  // the cell's radius) and the norm bound rounded UP to fp32 (0 = empty)
  const float4 *cell4, *Cell4;
  const float4 *Mid4;  // mid cells: 4 x 4 fine cells, the same record
};
constexpr double kCell4Slack = 2.5e-7;

// ------------------------------------------------------------------------------------------------------------------
// wavefront helpers
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return static_cast<int>(threadIdx.x & 63u); }

// Reductions over the wavefront in DPP steps (row_shr 1, 2, 4, 8 inside the rows of 16 lanes, then row_bcast:15 and
// row_bcast:31 across them -- the sequence of LLVM's own wave reductions on gfx9): the total lands in lane 63 and is read
// from there.  Six steps of two register moves and one operation, no LDS round trip (the butterfly through
// ds_bpermute took ~700 cycles of latency per reduction, four reductions per cut of the polygon).
template <int kCtrl, int kRowMask>
__device__ __forceinline__ double dpp_move(double identity, double v) {
  const long long iv = __double_as_longlong(v), id = __double_as_longlong(identity);
  const int lo = __builtin_amdgcn_update_dpp(static_cast<int>(id), static_cast<int>(iv), kCtrl, kRowMask, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(static_cast<int>(id >> 32), static_cast<int>(iv >> 32), kCtrl, kRowMask, 0xf, false);
  return __longlong_as_double((static_cast<long long>(hi) << 32) | static_cast<unsigned int>(lo));
}

template <typename Op>
__device__ __forceinline__ double wave_reduce(double v, double identity, Op op) {
  v = op(v, dpp_move<0x111, 0xf>(identity, v));  // row_shr:1
  v = op(v, dpp_move<0x112, 0xf>(identity, v));  // row_shr:2
  v = op(v, dpp_move<0x114, 0xf>(identity, v));  // row_shr:4
  v = op(v, dpp_move<0x118, 0xf>(identity, v));  // row_shr:8
  v = op(v, dpp_move<0x142, 0xa>(identity, v));  // row_bcast:15 into rows 1 and 3
  v = op(v, dpp_move<0x143, 0xc>(identity, v));  // row_bcast:31 into rows 2 and 3
  const long long t = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane(static_cast<int>(t), 63), hi = __builtin_amdgcn_readlane(static_cast<int>(t >> 32), 63);
  return __longlong_as_double((static_cast<long long>(hi) << 32) | static_cast<unsigned int>(lo));
}

__device__ __forceinline__ double wave_sum(double v) {
  return wave_reduce(v, 0.0, [](double a, double b) { return a + b; });
}
__device__ __forceinline__ double wave_min(double v) {
  return wave_reduce(v, INFINITY, [](double a, double b) { return fmin(a, b); });
}
__device__ __forceinline__ double wave_max(double v) {
  return wave_reduce(v, -INFINITY, [](double a, double b) { return fmax(a, b); });
}

// 1 / b to ~1 ulp without the IEEE division's scaling and fix-up steps (v_rcp_f64 and two Newton steps; b is a moderate,
// non-zero number everywhere it is used).  For the SEARCH only -- frames, trial normals, polygon vertices: what the search
// finds is checked by the certificates, whose own arithmetic (differences, products, sums) has no division in it.
__device__ __forceinline__ double quick_rcp(double b) {
  double r = __builtin_amdgcn_rcp(b);
  r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
  return r;
}

// the value a lane holds, for a lane index every lane agrees on (a scalar register: v_readlane, no LDS permute)
__device__ __forceinline__ double lane_value(double v, int lane) {
  const int l = __builtin_amdgcn_readfirstlane(lane);
  const long long t = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane(static_cast<int>(t), l), hi = __builtin_amdgcn_readlane(static_cast<int>(t >> 32), l);
  return __longlong_as_double((static_cast<long long>(hi) << 32) | static_cast<unsigned int>(lo));
}
__device__ __forceinline__ int32_t lane_value(int32_t v, int lane) {
  return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(lane));
}

__device__ __forceinline__ unsigned long long order_key(double d) {
  const unsigned long long b = static_cast<unsigned long long>(__double_as_longlong(d));
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

inline double key_to_double(unsigned long long k) {
  const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  double d;
  std::memcpy(&d, &b, 8);
  return d;
}

// ------------------------------------------------------------------------------------------------------------------
// the feasible polygon of the half-planes seen so far: vertex i in lane i, eid = the constraint whose boundary line
// carries the edge from vertex i to vertex i + 1 (ids >= 0: candidates in cell order; < 0: the initial box)
// ------------------------------------------------------------------------------------------------------------------
struct Polygon {
  double vx, vy;
  double lx, ly, le;  // the line of the edge from this vertex to the next: s . (lx, ly) = le, inside <=
  int32_t eid;
  int nv;  // wave-uniform
};

__device__ __forceinline__ void polygon_box(Polygon &P, double B) {
  const int l = lane_id();
  // counter-clockwise square; edge ids -1 .. -4: top, left, bottom, right
  P.vx = (l == 0 || l == 3) ? B : -B;
  P.vy = (l == 0 || l == 1) ? B : -B;
  P.lx = l == 1 ? -1.0 : (l == 3 ? 1.0 : 0.0);
  P.ly = l == 0 ? 1.0 : (l == 2 ? -1.0 : 0.0);
  P.le = B;
  P.eid = -1 - l;
  P.nv = 4;
}

// where the lines s . (ax, ay) = ae and s . (bx, by) = be meet; false when they are parallel in floating point
__device__ __forceinline__ bool line_meet(double ax, double ay, double ae, double bx, double by, double be, double &x,
                                          double &y) {
  const double det = ax * by - ay * bx;
  if (det == 0.0) return false;
  const double inv = quick_rcp(det);
  x = (ae * by - ay * be) * inv;
  y = (ax * be - ae * bx) * inv;
  return isfinite(x) && isfinite(y);
}

// Clip with s . D <= E.  0: nothing cut, 1: cut, 2: nothing left (cert_a / cert_b = the two constraints that meet in
// the vertex nearest to the half-plane: with `id` their intersection is empty), 3: the inside vertices are not one
// cyclic run (round-off), or the polygon would need a 65th vertex.
__device__ __forceinline__ int polygon_clip(Polygon &P, double Dx, double Dy, double E, int32_t id, int32_t &cert_a,
                                            int32_t &cert_b) {
  const int l = lane_id();
  const int nv = P.nv;
  const bool valid = l < nv;
  const double ax = Dx * P.vx, ay = Dy * P.vy;
  const double val = valid ? (ax + ay) - E : 0.0;
  // a vertex within the rounding of its own evaluation counts as inside: a vertex this very half-plane created
  // earlier must not be cut again (the clip would never settle)
  const double tol = 4.0e-16 * ((fabs(ax) + fabs(ay)) + fabs(E));
  const unsigned long long m_valid = nv >= 64 ? ~0ull : ((1ull << nv) - 1ull);
  // ... and so does a vertex that lies on this half-plane's own line by construction (an end of an edge it made
  // earlier), however ill-conditioned the intersection that placed it
  const int32_t eid_prev = __shfl(P.eid, (l + nv - 1) % nv, 64);
  const bool own = P.eid == id || eid_prev == id;
  const unsigned long long m_in = __ballot(valid && (val <= tol || own)) & m_valid;
  if (m_in == m_valid) return 0;
  if (m_in == 0ull) {
    const double best = wave_min(valid ? val : INFINITY);
    const unsigned long long at = __ballot(valid && val == best);
    const int bl = static_cast<int>(__builtin_ctzll(at));
    cert_a = lane_value(P.eid, bl);
    cert_b = lane_value(P.eid, (bl + nv - 1) % nv);
    return 2;
  }
  const int cnt = __popcll(m_in);
  if (cnt + 2 > 64) return 4;
  const unsigned long long prev = ((m_in << 1) | (m_in >> (nv - 1))) & m_valid;  // bit i = inside(i - 1)
  const unsigned long long starts = m_in & ~prev;
  if (__popcll(starts) != 1) return 3;
  const int a = static_cast<int>(__builtin_ctzll(starts));
  const int b = (a + cnt - 1) % nv, b1 = (b + 1) % nv, a0 = (a + nv - 1) % nv;
  const double vbx = lane_value(P.vx, b), vby = lane_value(P.vy, b), valb = lane_value(val, b);
  const double vb1x = lane_value(P.vx, b1), vb1y = lane_value(P.vy, b1), valb1 = lane_value(val, b1);
  const double va0x = lane_value(P.vx, a0), va0y = lane_value(P.vy, a0), vala0 = lane_value(val, a0);
  const double vax = lane_value(P.vx, a), vay = lane_value(P.vy, a), vala = lane_value(val, a);
  const int32_t eid_a0 = lane_value(P.eid, a0);
  // The new vertices are where the half-plane's line meets the lines of the two edges it crosses -- from the LINES, not
  // by interpolating between the edges' end points: an edge of the initial box is 2^31 long, and a point near s = 0
  // interpolated between end points that far apart is only good to 1e-7.
  const double lbx = lane_value(P.lx, b), lby = lane_value(P.ly, b), lbe = lane_value(P.le, b);
  const double l0x = lane_value(P.lx, a0), l0y = lane_value(P.ly, a0), l0e = lane_value(P.le, a0);
  double x1x, x1y, x2x, x2y;
  if (!line_meet(lbx, lby, lbe, Dx, Dy, E, x1x, x1y)) {
    const double t1 = fmin(fmax(valb / (valb - valb1), 0.0), 1.0);  // valb <= tol < valb1
    x1x = vbx + t1 * (vb1x - vbx);
    x1y = vby + t1 * (vb1y - vby);
  }
  if (!line_meet(l0x, l0y, l0e, Dx, Dy, E, x2x, x2y)) {
    const double t2 = fmin(fmax(vala0 / (vala0 - vala), 0.0), 1.0);  // vala <= tol < vala0
    x2x = va0x + t2 * (vax - va0x);
    x2y = va0y + t2 * (vay - va0y);
  }
  const int src = (a + l) % nv;
  double nvx = __shfl(P.vx, src, 64), nvy = __shfl(P.vy, src, 64);
  int32_t neid = __shfl(P.eid, src, 64);
  double nlx = __shfl(P.lx, src, 64), nly = __shfl(P.ly, src, 64), nle = __shfl(P.le, src, 64);
  if (l == cnt) {
    nvx = x1x;
    nvy = x1y;
    neid = id;
    nlx = Dx;
    nly = Dy;
    nle = E;
  } else if (l == cnt + 1) {
    nvx = x2x;
    nvy = x2y;
    neid = eid_a0;
    nlx = l0x;
    nly = l0y;
    nle = l0e;
  }
  P.vx = nvx;
  P.vy = nvy;
  P.lx = nlx;
  P.ly = nly;
  P.le = nle;
  P.eid = neid;
  P.nv = cnt + 2;
  return 1;
}

// ------------------------------------------------------------------------------------------------------------------
// the search of one candidate
// ------------------------------------------------------------------------------------------------------------------
struct Search {
  Vec3d p;           // the candidate's flipped point
  Vec3d e0, e1, e2;  // frame at p: e0 radial
  int32_t self;      // its place in cell order
  int32_t self_idx;  // its input index
  // trial normal
  Vec3d n, nh;
  double nn_hi, hp_lo;
  // outcome of a pass
  bool changed, uncertain_left;
  int32_t cert_a, cert_b, cert_c;
  int status;  // 0 running, kSearchEmpty, kSearchFail, kSearchHiddenDup
  int fail_code;
  unsigned long long tests;
};

__device__ __forceinline__ void search_frame(Search &S) {
  const double irho = quick_rcp(sqrt(S.p.x * S.p.x + S.p.y * S.p.y + S.p.z * S.p.z));
  S.e0 = {S.p.x * irho, S.p.y * irho, S.p.z * irho};
  // e1 = the x axis made orthogonal to e0 (the candidates lie in the camera's cone about +z, so e0 is never near x)
  double ax = 1.0 - S.e0.x * S.e0.x, ay = -S.e0.x * S.e0.y, az = -S.e0.x * S.e0.z;
  const double ian = quick_rcp(sqrt(ax * ax + ay * ay + az * az));
  S.e1 = {ax * ian, ay * ian, az * ian};
  S.e2 = {S.e0.y * S.e1.z - S.e0.z * S.e1.y, S.e0.z * S.e1.x - S.e0.x * S.e1.z, S.e0.x * S.e1.y - S.e0.y * S.e1.x};
}

// Trial normal: the point of the polygon nearest to s = 0 (the least tilted plane the constraints so far allow), moved
// up to kNudge into the polygon so that it is strictly inside.  Least tilt keeps the trial plane nearly tangent to the
// sphere of flipped points, which is what keeps the set of points that can reach it small (file header, "Locality"),
// whatever the size of the initial box.
constexpr double kNudge = 1.0e-6;

__device__ __forceinline__ void search_witness(Search &S, const Polygon &P) {
  const int l = lane_id();
  const bool valid = l < P.nv;
  const int nxt = (l + 1 < P.nv) ? l + 1 : 0;
  const double wx = __shfl(P.vx, nxt, 64), wy = __shfl(P.vy, nxt, 64);
  const double ex = wx - P.vx, ey = wy - P.vy;
  const double l2 = ex * ex + ey * ey;
  double t = l2 > 0.0 ? -(P.vx * ex + P.vy * ey) * quick_rcp(l2) : 0.0;
  t = fmin(fmax(t, 0.0), 1.0);
  const double cx = P.vx + t * ex, cy = P.vy + t * ey;
  const double d2 = valid ? cx * cx + cy * cy : INFINITY;
  const double dmin2 = wave_min(d2);
  // counter-clockwise polygon: s = 0 is inside iff it is to the left of every edge
  const bool inside = !__ballot(valid && !(P.vx * wy - P.vy * wx > 0.0));
  double sx = 0.0, sy = 0.0;
  if (!(inside && dmin2 > kNudge * kNudge)) {
    // b = the boundary point nearest to s = 0.  Every vertex pulled in to within kNudge of b stays on its segment
    // from b, hence in the polygon, and so does their mean: a point of the polygon next to b whose direction from b
    // does not depend on how far away the far vertices (the initial box) are.
    const int src = static_cast<int>(__builtin_ctzll(__ballot(valid && d2 == dmin2)));
    const double bx = lane_value(cx, src), by = lane_value(cy, src);
    const double gx = P.vx - bx, gy = P.vy - by;
    const double len = sqrt(gx * gx + gy * gy);
    const double f = len > kNudge ? kNudge * quick_rcp(len) : 1.0;
    sx = bx + wave_sum(valid ? f * gx : 0.0) / P.nv;
    sy = by + wave_sum(valid ? f * gy : 0.0) / P.nv;
  }
  S.n = {S.e0.x + (sx * S.e1.x + sy * S.e2.x), S.e0.y + (sx * S.e1.y + sy * S.e2.y), S.e0.z + (sx * S.e1.z + sy * S.e2.z)};
  const double nn = sqrt(S.n.x * S.n.x + S.n.y * S.n.y + S.n.z * S.n.z);
  const double inv = quick_rcp(nn);  // n / |n| to a few ulp: the cell bound carries 1e-12 of slack for it
  S.nh = {S.n.x * inv, S.n.y * inv, S.n.z * inv};
  S.nn_hi = nn * (1.0 + 1.0e-14);
  const double hp = S.n.x * S.p.x + S.n.y * S.p.y + S.n.z * S.p.z;
  S.hp_lo = hp * (1.0 - 1.0e-13);  // n . p > 0: the origin is on the inner side of every trial plane
}

// Candidates [k0, k1) of the cell order against the trial normal, 64 at a time (one per lane).  A point is cleared when
//   fl(n . (q - p)) < -1e-15 sum |n_i fl(q_i - p_i)|:
// the subtraction, the product and the two additions of a term are each within 2^-53 relative, so the computed value
// is within 4.44e-16 sum |n_i (q_i - p_i)| of the real one.  While some lane's point is not cleared, the worst of them
// (largest n . d relative to its bound) clips the polygon and the trial normal is taken again at once -- the 64
// differences stay in registers, so the wavefront solves the batch's own little feasibility problem without touching
// memory.  Points cleared by an earlier trial normal of the pass are re-tested by the caller's next pass.
__device__ __forceinline__ void test_range(Search &S, Polygon &P, const HprArrays &A, int32_t k0, int32_t k1) {
  const int l = lane_id();
  for (int32_t base = k0; base < k1 && S.status == 0; base += 64) {
    const int32_t k = base + l;
    const bool active = k < k1 && k != S.self;
    double dx = 0.0, dy = 0.0, dz = 0.0;
    if (active) {
      dx = A.sx[k] - S.p.x;
      dy = A.sy[k] - S.p.y;
      dz = A.sz[k] - S.p.z;
    }
    S.tests += 1;
    const bool dup = active && dx == 0.0 && dy == 0.0 && dz == 0.0;
    if (__ballot(dup)) {
      // identical flipped points: the lowest input index stands for the group
      if (__ballot(dup && A.sidx[k] < S.self_idx)) {
        S.status = kSearchHiddenDup;
        return;
      }
    }
    bool open = active && !dup;  // this lane's point may still clip
    for (int guard = 0;; ++guard) {
      if (guard > 192) {  // every cut removes a vertex or retires a lane: 64 lanes cannot need this many
        S.fail_code = 1;
        S.status = kSearchFail;
        return;
      }
      const double tx = S.n.x * dx, ty = S.n.y * dy, tz = S.n.z * dz;
      const double t = (tx + ty) + tz;
      const double T = (fabs(tx) + fabs(ty)) + fabs(tz);
      const bool bad = open && !(t < -kPointSlack * T);
      if (!__ballot(bad)) break;
      // which one is "worst" only steers the search: the ratio in fp32
      const double score = bad ? static_cast<double>(static_cast<float>(t) * __builtin_amdgcn_rcpf(static_cast<float>(T))) : -INFINITY;
      const double worst = wave_max(score);
      const int src = static_cast<int>(__builtin_ctzll(__ballot(bad && score == worst)));
      const double qx = lane_value(dx, src), qy = lane_value(dy, src), qz = lane_value(dz, src);
      const double Dx = (qx * S.e1.x + qy * S.e1.y) + qz * S.e1.z;
      const double Dy = (qx * S.e2.x + qy * S.e2.y) + qz * S.e2.z;
      const double E = -((qx * S.e0.x + qy * S.e0.y) + qz * S.e0.z);
      int32_t ca = 0, cb = 0;
      const int r = polygon_clip(P, Dx, Dy, E, base + src, ca, cb);
      if (r == 1) {
        S.changed = true;
        search_witness(S, P);
      } else if (r == 2) {
        S.status = kSearchEmpty;
        S.cert_a = ca;
        S.cert_b = cb;
        S.cert_c = base + src;
        return;
      } else if (r >= 3) {
        S.fail_code = r;
        S.status = kSearchFail;
        return;
      } else {
        // the whole polygon satisfies this point's half-plane, yet the trial normal (a point of the polygon) does not
        // clear it: the point is within round-off of the trial plane and there is no cut to move the plane away
        S.uncertain_left = true;
        if (l == src) open = false;
      }
    }
  }
}

// is every point of the cell on the inner side of the trial plane?  (file header, "Locality")
__device__ __forceinline__ bool cell_cleared(const Search &S, double ux, double uy, double uz, double rho, double r) {
  const double dx = S.nh.x - ux, dy = S.nh.y - uy, dz = S.nh.z - uz;
  const double sep = sqrt((dx * dx + dy * dy) + dz * dz);
  const double sl = fmax(sep - r - 1.0e-12, 0.0);
  return rho * S.nn_hi * (1.0 - 0.5 * sl * sl) < S.hp_lo;
}

// The same bound with the chord taken in fp32: sep only enters through max(sep - r, 0), which must not be OVER-estimated, so
// the fp32 square root (1 ulp) is scaled down by 1 - 3e-7 and the square under it is the fp64 one rounded down.  An fp64
// square root is ~25 instructions, and k_hpr_radial -- 78 % vector-ALU busy -- takes one per cell it looks at.
__device__ __forceinline__ bool cell_cleared_f32sep(double nhx, double nhy, double nhz, double nn_hi, double hp_lo, double ux,
                                                    double uy, double uz, double rho, double r) {
  const double dx = nhx - ux, dy = nhy - uy, dz = nhz - uz;
  const float sep_lo = __builtin_sqrtf(__double2float_rd((dx * dx + dy * dy) + dz * dz)) * (1.0f - 3.0e-7f);
  const double sl = fmax(static_cast<double>(sep_lo) - r - 1.0e-12, 0.0);
  return rho * nn_hi * (1.0 - 0.5 * sl * sl) < hp_lo;
}

// The candidates that could reach the plane (S.nh, S.nn_hi, S.hp_lo), as runs of the cell order handed to
// range(k0, k1); range returns false to stop.  traverse_near: the 3 x 3 cells around the candidate's own -- that is
// where the binding constraints are; traverse_all: every cell the bound cannot clear (the near cells again included).
template <typename RangeFn>
__device__ __forceinline__ void traverse_near(const Search &S, const HprArrays &A, const HprGrid &G, RangeFn range) {
  const int32_t cell = A.scell[S.self];
  const int32_t ci = cell % G.gw, cj = cell / G.gw;
  for (int dj = -1; dj <= 1; ++dj) {
    const int32_t rj = cj + dj;
    if (rj < 0 || rj >= G.gh) continue;
    const int32_t c0 = rj * G.gw + max(ci - 1, 0), c1 = rj * G.gw + min(ci + 1, G.gw - 1);
    if (!range(A.cstart[c0], A.cstart[c1 + 1])) return;
  }
}

// Which coarse cells can hold a point that reaches the plane at all?  With rho_max >= |q| for every candidate, a point
// can only fail n . q < n . p when the chord between its direction u and n / |n| is at most
//     sep = sqrt(2 (1 - hp_lo / (rho_max |n|)))                                   (0 when p itself tops the plane).
// Gnomonic coordinates A = (u_x, u_y) / u_z of two directions differ by at most |u - u'| (1 + |A'|) / u_z, and
// 1 / u_z <= a_reach for every candidate, so such a point lies within R = sep (1 + |A_n|) a_reach of A_n, the gnomonic
// image of n.  The window is the coarse cells that disc touches; a plane tilted beyond the grid's reach (n_z <= 0) gets
// the whole grid.
struct Window {
  int32_t i0, i1, j0, j1;  // coarse cells, inclusive; i0 > i1: empty
};

// kEdge: fine cells per edge of the upper-level cells the window is in (kHprCoarse: cgw x cgh of them; kHprMid: mgw x mgh)
template <int kEdge = kHprCoarse>
__device__ __forceinline__ Window reach_window(const Search &S, const HprGrid &G) {
  // (only a SUPERSET of the cells is asked for: reciprocals instead of divisions, |a| + |b| for the norm, a cell of margin
  // for the rounding of the products -- this routine was a quarter of k_hpr_radial's vector instructions)
  const int32_t ugw = kEdge == kHprCoarse ? G.cgw : G.mgw, ugh = kEdge == kHprCoarse ? G.cgh : G.mgh;
  Window W = {0, ugw - 1, 0, ugh - 1};
  const double ratio = S.hp_lo * quick_rcp(*G.rho_max * S.nn_hi) * (1.0 - 1.0e-14);
  if (ratio >= 1.0) return {1, 0, 1, 0};
  if (!(S.nh.z > 1.0e-3)) return W;
  const double sep = sqrt(2.0 * (1.0 - ratio)) * (1.0 + 1.0e-9) + 1.0e-12;
  const double inz = quick_rcp(S.nh.z);
  const double ax = S.nh.x * inz, ay = S.nh.y * inz;  // to ~1 ulp; R below carries 1e-9 of slack
  const double R = sep * (1.0 + (fabs(ax) + fabs(ay))) * G.a_reach * (1.0 + 1.0e-9);
  const double iH = G.inv_h * (1.0 / kEdge);
  const double fi0 = floor((ax - R - G.a0) * iH) - 1.0, fi1 = floor((ax + R - G.a0) * iH) + 1.0;
  const double fj0 = floor((ay - R - G.b0) * iH) - 1.0, fj1 = floor((ay + R - G.b0) * iH) + 1.0;
  if (!(fi0 == fi0 && fi1 == fi1 && fj0 == fj0 && fj1 == fj1)) return W;  // NaN: no window
  if (fi1 < 0.0 || fj1 < 0.0 || fi0 > ugw - 1 || fj0 > ugh - 1) return {1, 0, 1, 0};
  W.i0 = static_cast<int32_t>(fmax(fi0, 0.0));
  W.j0 = static_cast<int32_t>(fmax(fj0, 0.0));
  W.i1 = static_cast<int32_t>(fmin(fi1, static_cast<double>(ugw - 1)));
  W.j1 = static_cast<int32_t>(fmin(fj1, static_cast<double>(ugh - 1)));
  return W;
}

// t / w for 0 <= t < 2^24, 0 < w < 2^24 without the integer division's expansion: the float quotient is within one of the
// true one, two comparisons settle it
__device__ __forceinline__ int32_t small_div(int32_t t, int32_t w) {
  int32_t q = static_cast<int32_t>(static_cast<float>(t) * __builtin_amdgcn_rcpf(static_cast<float>(w)));
  q -= q * w > t ? 1 : 0;
  q += (q + 1) * w <= t ? 1 : 0;
  return q;
}

// skip_near: leave out the 3 x 3 cells traverse_near covers (run_search only calls this after a near pass that changed
// nothing: those cells are cleared for the current trial normal already; the exact path enumerates everything)
template <bool kSkipNear = false, typename RangeFn>
__device__ __forceinline__ void traverse_all(const Search &S, const HprArrays &A, const HprGrid &G, RangeFn range) {
  const int l = lane_id();
  const int32_t own = kSkipNear ? A.scell[S.self] : 0;
  const int32_t oi = own % G.gw, oj = own / G.gw;
  const int32_t n_coarse = G.cgw * G.cgh, n_fine = G.gw * G.gh;
  const Window W = reach_window(S, G);
  if (W.i0 > W.i1 || W.j0 > W.j1) return;
  const int32_t ww = W.i1 - W.i0 + 1, wn = ww * (W.j1 - W.j0 + 1);
  for (int32_t cb = 0; cb < wn; cb += 64) {
    const int32_t t = cb + l;
    const int32_t C = t < wn ? (W.j0 + t / ww) * G.cgw + W.i0 + t % ww : -1;
    bool open = false;
    if (C >= 0) {
      const double rho = A.Crho[C];
      open = rho > 0.0 && !cell_cleared(S, A.Cdir[C], A.Cdir[n_coarse + C], A.Cdir[2 * n_coarse + C], rho, G.r_coarse);
    }
    unsigned long long open_c = __ballot(open);
    while (open_c) {
      const int32_t Cc = lane_value(C, static_cast<int>(__builtin_ctzll(open_c)));
      open_c &= open_c - 1ull;
      const int32_t fi = (Cc % G.cgw) * kHprCoarse + (l & 7), fj = (Cc / G.cgw) * kHprCoarse + (l >> 3);
      bool fopen = false;
      int32_t f = 0;
      if (fi < G.gw && fj < G.gh && !(kSkipNear && abs(fi - oi) <= 1 && abs(fj - oj) <= 1)) {
        f = fj * G.gw + fi;
        const double rho = A.crho[f];
        fopen = rho > 0.0 && !cell_cleared(S, A.cdir[f], A.cdir[n_fine + f], A.cdir[2 * n_fine + f], rho, G.r_fine);
      }
      unsigned long long open_f = __ballot(fopen);
      while (open_f) {
        // a run of open cells side by side in one row of the coarse cell is one run of the cell order: one range, full
        // batches, instead of a batch of ~8 points per cell
        const int src = static_cast<int>(__builtin_ctzll(open_f));
        const int room = 8 - (src & 7);
        const int run = min(room, static_cast<int>(__builtin_ctzll(~(open_f >> src))));
        open_f &= ~(((1ull << run) - 1ull) << src);
        const int32_t ff = lane_value(f, src);
        if (!range(A.cstart[ff], A.cstart[ff + run])) return;
      }
    }
  }
}

// Runs the search from a box of half-width B.  Returns kSearchVisible (the last trial normal cleared every point;
// `uncertain_left` tells whether some only within round-off), kSearchEmpty (cert_a / cert_b / cert_c), kSearchFail or
// kSearchHiddenDup.
__device__ __forceinline__ int run_search(Search &S, Polygon &P, const HprArrays &A, const HprGrid &G, double B,
                                          int max_restarts, unsigned long long &restarts) {
  polygon_box(P, B);
  S.status = 0;
  search_witness(S, P);
  for (int it = 0; it < max_restarts; ++it) {
    S.changed = false;
    S.uncertain_left = false;
    // the near cells first; once they have moved the trial normal, what it cleared before must be seen again
    traverse_near(S, A, G, [&](int32_t k0, int32_t k1) {
      test_range(S, P, A, k0, k1);
      return S.status == 0;
    });
    if (S.status == 0 && !S.changed)
      traverse_all<true>(S, A, G, [&](int32_t k0, int32_t k1) {
        test_range(S, P, A, k0, k1);
        return S.status == 0 && !S.changed;
      });
    restarts += 1;
    if (S.status != 0) return S.status;
    if (!S.changed) return kSearchVisible;
  }
  S.fail_code = 2;
  return kSearchFail;
}

__device__ __forceinline__ Vec3d load_point(const HprArrays &A, int32_t k) { return {A.sx[k], A.sy[k], A.sz[k]}; }

// p inside (or on) the tetrahedron (origin, a, b, c)?  Filtered signs: 1 yes, 0 cannot tell (or no).
// The eight orientations of the general test -- for each face, the side of the opposite vertex and the side of p -- with one
// vertex at the origin: orient(o, b, c, a) = orient(o, c, a, b) = orient(o, a, b, c) = -det [a; b; c] = -orient(a, b, c, o), and
// orient(o, y, z, p) = -det [p; y; z].  So: D = det [a; b; c] certain and non-zero, and det [a - p; b - p; c - p], det [p; b; c],
// det [p; c; a], det [p; a; b] certain and of D's sign.  The four determinants through the origin are orient3d_det's own formula with
// d = 0 (differences that are exact: the same forward error bound, 8e-16 x the permanent) and share their 2 x 2 minors: six
// minors instead of twenty-four -- the quick certificate runs this test up to twelve times per candidate.
__device__ __forceinline__ int tetra_contains_filtered(const Vec3d &p, const Vec3d &a, const Vec3d &b, const Vec3d &c) {
  // M(u, v) = u.x v.y - v.x u.y and P(u, v) = |u.x v.y| + |v.x u.y|
  const double bxcy = b.x * c.y, cxby = c.x * b.y, cxay = c.x * a.y, axcy = a.x * c.y, axby = a.x * b.y, bxay = b.x * a.y;
  const double Mbc = bxcy - cxby, Mca = cxay - axcy, Mab = axby - bxay;
  const double Pbc = fabs(bxcy) + fabs(cxby), Pca = fabs(cxay) + fabs(axcy), Pab = fabs(axby) + fabs(bxay);
  const double D = a.z * Mbc + b.z * Mca + c.z * Mab;
  const double permD = Pbc * fabs(a.z) + Pca * fabs(b.z) + Pab * fabs(c.z);
  const int sD = D > 8.0e-16 * permD ? 1 : (-D > 8.0e-16 * permD ? -1 : 0);
  if (sD == 0) return 0;
  const double pxay = p.x * a.y, axpy = a.x * p.y, pxby = p.x * b.y, bxpy = b.x * p.y, pxcy = p.x * c.y, cxpy = c.x * p.y;
  const double Mpa = pxay - axpy, Mpb = pxby - bxpy, Mpc = pxcy - cxpy;  // M(a, p) = -M(p, a): a negation is exact
  const double Ppa = fabs(pxay) + fabs(axpy), Ppb = fabs(pxby) + fabs(bxpy), Ppc = fabs(pxcy) + fabs(cxpy);
  auto sign_of = [](double det, double perm) { return det > 8.0e-16 * perm ? 1 : (-det > 8.0e-16 * perm ? -1 : 0); };
  // det [p; b; c] = p.z M(b, c) + b.z M(c, p) + c.z M(p, b), and cyclically
  const int s1 = sign_of(p.z * Mbc + b.z * (-Mpc) + c.z * Mpb, Pbc * fabs(p.z) + Ppc * fabs(b.z) + Ppb * fabs(c.z));
  if (s1 != sD) return 0;
  const int s2 = sign_of(p.z * Mca + c.z * (-Mpa) + a.z * Mpc, Pca * fabs(p.z) + Ppa * fabs(c.z) + Ppc * fabs(a.z));
  if (s2 != sD) return 0;
  const int s3 = sign_of(p.z * Mab + a.z * (-Mpb) + b.z * Mpa, Pab * fabs(p.z) + Ppb * fabs(a.z) + Ppa * fabs(b.z));
  if (s3 != sD) return 0;
  return orient3d_filtered(a, b, c, p) == sD ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------------------------

// The candidates of one keyframe straight from the SORTED copy of the cloud (the order the batched passes walk): the filter
// of view_culling.cpp:276-288 (project_point's cull cell), the flip, the gnomonic coordinates and their bounds in one kernel,
// appended wavefront by wavefront to the candidate arrays -- no flag array of the whole map, no compaction, no second
// gather of the coordinates by index, and the host learns the count and the bounds in ONE wait.  place[] keeps the point's
// place in the sorted order (the whole-run bits are addressed by it), index[] its input index (the duplicate rule and the
// flags of pcp_cull_frame).  The order of the candidates is whatever the appends make it; no result depends on it.
constexpr int kStatCandidates = 20;  // block 0 of the tallies: number of candidates (a cache line away from the bounds)
constexpr int kStatSearch = 21;      // block 0 of the tallies: length of the list of the 64-lane searches
__global__ __launch_bounds__(kHprBlock) void k_hpr_candidates(const float *__restrict__ x, const float *__restrict__ y,
                                                              const float *__restrict__ z, int64_t n, DevCamera cam, DevFrame fr,
                                                              const int32_t *__restrict__ perm, double flip_radius,
                                                              int64_t stride, int32_t *__restrict__ index,
                                                              int32_t *__restrict__ place, double *__restrict__ f64,
                                                              unsigned long long *__restrict__ stats,
                                                              const uint32_t *__restrict__ tile_mask, int32_t mask_words,
                                                              int32_t frame) {
  const int64_t j = static_cast<int64_t>(blockIdx.x) * kHprBlock + threadIdx.x;
  bool cand = false;
  float xc = 0.0f, yc = 0.0f, zc = 0.0f;
  // tile_mask (the whole-run pass; null for a single keyframe): the tile x keyframe masks of the batched passes as the tile
  // level left them (pcp_colour.hip k_tile_mask_dense) -- a cleared bit says that no point of the tile (this wavefront's 64
  // points of the sorted copy) can pass the filter in this keyframe, so the wavefront neither reads nor projects them
  bool visit = j < n;
  if (tile_mask && visit) visit = ((tile_mask[(j >> 6) * mask_words + (frame >> 5)] >> (frame & 31)) & 1u) != 0u;
  if (visit) {
    const Projected p = project_point(cam, fr.w2c, x[j], y[j], z[j]);
    cand = p.cell != -1;
    xc = p.xc;
    yc = p.yc;
    zc = p.zc;
  }
  const unsigned long long m = __ballot(cand);
  unsigned long long amin = ~0ull, amax = 0ull, bmin = ~0ull, bmax = 0ull;
  // places in the candidate arrays: ONE atomic per workgroup that holds candidates (an atomic per wavefront on the one
  // counter queues up in its L2 channel: ~15 k of them per keyframe cost more than the projection).  Most workgroups hold
  // none -- the cloud is in spatial order, a keyframe sees a few per cent of it -- and leave at the first barrier.
  __shared__ uint32_t wave_count[kHprBlock / 64];
  __shared__ unsigned long long block_base;
  __shared__ unsigned long long part[kHprBlock / 64][4];
  const int w = static_cast<int>(threadIdx.x >> 6);
  if (lane_id() == 0) wave_count[w] = static_cast<uint32_t>(__popcll(m));
  if (!__syncthreads_or(m != 0ull ? 1 : 0)) return;
  if (threadIdx.x == 0) {
    uint32_t total = 0;
    for (int k = 0; k < kHprBlock / 64; ++k) total += wave_count[k];
    block_base = atomicAdd(&stats[kStatCandidates], static_cast<unsigned long long>(total));
  }
  __syncthreads();
  if (m) {
    unsigned long long base = block_base;
    for (int k = 0; k < w; ++k) base += wave_count[k];
    if (cand) {
      const int64_t k = static_cast<int64_t>(base) + __popcll(m & ((1ull << lane_id()) - 1ull));
      // pt_norm = pt.head<3>().norm(); flipped = pt + 2.0 * (max_z - pt_norm) * pt / pt_norm, per coefficient
      // x + ((2.0 * (R - norm)) * x) / norm on the promoted camera coordinates (view_culling.cpp:291-292)
      const double X = xc, Y = yc, Z = zc;
      const double norm = sqrt((X * X + Y * Y) + Z * Z);
      const double s = 2.0 * (flip_radius - norm);
      const double fx = X + (s * X) / norm, fy = Y + (s * Y) / norm, fz = Z + (s * Z) / norm;
      const double ifz = quick_rcp(fz);  // (the gnomonic coordinates only bin the candidates: no IEEE division needed)
      const double a = fx * ifz, b = fy * ifz;
      f64[k] = fx;
      f64[stride + k] = fy;
      f64[2 * stride + k] = fz;
      f64[3 * stride + k] = a;
      f64[4 * stride + k] = b;
      f64[5 * stride + k] = sqrt((fx * fx + fy * fy) + fz * fz) * (1.0 + 1.0e-15);  // an upper bound of the norm
      index[k] = perm ? perm[j] : static_cast<int32_t>(j);
      place[k] = static_cast<int32_t>(j);
      amin = amax = order_key(a);
      bmin = bmax = order_key(b);
    }
  }
  // bounds: per wavefront (a wavefront without a candidate skips the 24 64-bit shuffles), then one set of atomics per
  // workgroup
  if (m) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      amin = min(amin, static_cast<unsigned long long>(__shfl_xor(static_cast<long long>(amin), o, 64)));
      amax = max(amax, static_cast<unsigned long long>(__shfl_xor(static_cast<long long>(amax), o, 64)));
      bmin = min(bmin, static_cast<unsigned long long>(__shfl_xor(static_cast<long long>(bmin), o, 64)));
      bmax = max(bmax, static_cast<unsigned long long>(__shfl_xor(static_cast<long long>(bmax), o, 64)));
    }
  }
  if (lane_id() == 0) {
    part[w][0] = amin;
    part[w][1] = amax;
    part[w][2] = bmin;
    part[w][3] = bmax;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < kHprBlock / 64; ++k) {
      amin = min(amin, part[k][0]);
      amax = max(amax, part[k][1]);
      bmin = min(bmin, part[k][2]);
      bmax = max(bmax, part[k][3]);
    }
    // kStatCopies copies of the bounds, each on a line of its own (thousands of atomics on ONE line queue up in its L2
    // channel: ~7 ns each, 60-120 us per keyframe); the minima as maxima of the inverted keys, so that zeroed memory is
    // the identity of all four; the host folds the copies
    unsigned long long *mine = stats + kStatStride * (1 + (blockIdx.x % kStatCopies)) + 24;
    if (amax != 0ull) {
      atomicMax(&mine[0], ~amin);
      atomicMax(&mine[1], amax);
      atomicMax(&mine[2], ~bmin);
      atomicMax(&mine[3], bmax);
    }
  }
}

// The number of candidates and the folded bounds, written straight into pinned HOST memory behind k_hpr_candidates, the
// sequence number last: the host polls that word instead of waiting for a device-to-host copy and its event (the copy
// engine's round trip was the larger part of a keyframe's one host wait, and its latency depends on what else the process
// has initialised: the same pass took 0.148 s alone and 0.163 s once torch had touched the device).
struct HprCounts {
  unsigned long long count, inv_amin, amax, inv_bmin, bmax, seq;
};
// `host` must be COHERENT (fine-grained) pinned memory: hipHostMallocCoherent (hpr_begin allocates it that way).
__global__ __launch_bounds__(64) void k_hpr_publish(const unsigned long long *__restrict__ stats, unsigned long long seq,
                                                    volatile HprCounts *__restrict__ host) {
  const int l = lane_id();
  unsigned long long v[4] = {0ull, 0ull, 0ull, 0ull};
  for (int c = 1 + l; c <= kStatCopies; c += 64)
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = max(v[k], stats[kStatStride * c + 24 + k]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = max(v[k], static_cast<unsigned long long>(__shfl_xor(static_cast<long long>(v[k]), o, 64)));
  if (l == 0) {
    host->count = stats[kStatCandidates];
    host->inv_amin = v[0];
    host->amax = v[1];
    host->inv_bmin = v[2];
    host->bmax = v[3];
    __threadfence_system();
    host->seq = seq;
    __threadfence_system();
  }
}

__device__ __forceinline__ int32_t hpr_cell_of(const HprGrid &G, double a, double b) {
  const int32_t ci = min(G.gw - 1, max(0, static_cast<int32_t>((a - G.a0) * G.inv_h)));
  const int32_t cj = min(G.gh - 1, max(0, static_cast<int32_t>((b - G.b0) * G.inv_h)));
  return cj * G.gw + ci;
}

__global__ __launch_bounds__(kHprBlock) void k_hpr_count(const double *__restrict__ ga, const double *__restrict__ gb,
                                                         HprGrid G, int32_t *__restrict__ cell, int32_t *__restrict__ count) {
  const int32_t k = static_cast<int32_t>(blockIdx.x) * kHprBlock + static_cast<int32_t>(threadIdx.x);
  if (k >= G.m) return;
  const int32_t c = hpr_cell_of(G, ga[k], gb[k]);
  cell[k] = c;
  atomicAdd(&count[c], 1);
}

__global__ __launch_bounds__(kHprBlock) void k_hpr_scatter(const double *__restrict__ px, const double *__restrict__ py,
                                                           const double *__restrict__ pz, const double *__restrict__ rho,
                                                           const int32_t *__restrict__ index, const int32_t *__restrict__ place,
                                                           const int32_t *__restrict__ cell,
                                                           int32_t m, const int32_t *__restrict__ cstart,
                                                           int32_t *__restrict__ cursor, double *__restrict__ sx,
                                                           double *__restrict__ sy, double *__restrict__ sz,
                                                           int32_t *__restrict__ sidx, int32_t *__restrict__ splace,
                                                           int32_t *__restrict__ scell,
                                                           unsigned long long *__restrict__ crho_bits,
                                                           unsigned long long *__restrict__ crep) {
  const int32_t k = static_cast<int32_t>(blockIdx.x) * kHprBlock + static_cast<int32_t>(threadIdx.x);
  if (k >= m) return;
  const int32_t c = cell[k];
  const int32_t pos = cstart[c] + atomicAdd(&cursor[c], 1);
  sx[pos] = px[k];
  sy[pos] = py[k];
  sz[pos] = pz[k];
  sidx[pos] = index[k];
  splace[pos] = place[k];
  scell[pos] = c;
  const unsigned long long rb = static_cast<unsigned long long>(__double_as_longlong(rho[k]));
  // (positive doubles order as integers; a candidate that would not raise what it reads stays away from the atomic)
  if (rb > __atomic_load_n(&crho_bits[c], __ATOMIC_RELAXED)) atomicMax(&crho_bits[c], rb);
  // a representative of the cell for k_hpr_quick: (one of) its outermost candidates, by place in the cell order
  if (crep) {
    const unsigned long long rep = (rb & ~0x3ffffffull) | static_cast<unsigned long long>(pos);
    if (rep > __atomic_load_n(&crep[c], __ATOMIC_RELAXED)) atomicMax(&crep[c], rep);
  }
}

// centre directions of the fine cells; coarse cells: centre direction and the largest rho of their fine cells
__global__ __launch_bounds__(kHprBlock) void k_hpr_cells(HprGrid G, const unsigned long long *__restrict__ crho_bits,
                                                         double *__restrict__ cdir, double *__restrict__ Crho,
                                                         double *__restrict__ Cdir, unsigned long long *__restrict__ rho_max_bits,
                                                         float4 *__restrict__ cell4, float4 *__restrict__ Cell4,
                                                         float4 *__restrict__ Mid4) {
  const int32_t t = static_cast<int32_t>(blockIdx.x) * kHprBlock + static_cast<int32_t>(threadIdx.x);
  const int32_t n_fine = G.gw * G.gh, n_coarse = G.cgw * G.cgh;
  if (t < G.mgw * G.mgh) {
    const int32_t Mi = t % G.mgw, Mj = t / G.mgw;
    unsigned long long best = 0ull;
    for (int dj = 0; dj < kHprMid; ++dj)
      for (int di = 0; di < kHprMid; ++di) {
        const int32_t fi = Mi * kHprMid + di, fj = Mj * kHprMid + dj;
        if (fi < G.gw && fj < G.gh) best = max(best, crho_bits[fj * G.gw + fi]);
      }
    const double a = G.a0 + (static_cast<double>(Mi) + 0.5) * (G.h * kHprMid);
    const double b = G.b0 + (static_cast<double>(Mj) + 0.5) * (G.h * kHprMid);
    const double inv = 1.0 / sqrt((a * a + b * b) + 1.0);
    Mid4[t] = make_float4(static_cast<float>(a * inv), static_cast<float>(b * inv), static_cast<float>(inv),
                          best ? __double2float_ru(__longlong_as_double(static_cast<long long>(best))) : 0.0f);
  }
  if (t < n_fine) {
    const double a = G.a0 + (static_cast<double>(t % G.gw) + 0.5) * G.h, b = G.b0 + (static_cast<double>(t / G.gw) + 0.5) * G.h;
    const double inv = 1.0 / sqrt((a * a + b * b) + 1.0);
    cdir[t] = a * inv;
    cdir[n_fine + t] = b * inv;
    cdir[2 * n_fine + t] = inv;
    const unsigned long long rb = crho_bits[t];
    cell4[t] = make_float4(static_cast<float>(a * inv), static_cast<float>(b * inv), static_cast<float>(inv),
                           rb ? __double2float_ru(__longlong_as_double(static_cast<long long>(rb))) : 0.0f);
  }
  if (t < n_coarse) {
    const int32_t Ci = t % G.cgw, Cj = t / G.cgw;
    unsigned long long best = 0ull;
    for (int dj = 0; dj < kHprCoarse; ++dj)
      for (int di = 0; di < kHprCoarse; ++di) {
        const int32_t fi = Ci * kHprCoarse + di, fj = Cj * kHprCoarse + dj;
        if (fi < G.gw && fj < G.gh) best = max(best, crho_bits[fj * G.gw + fi]);
      }
    Crho[t] = __longlong_as_double(static_cast<long long>(best));
    if (best) atomicMax(rho_max_bits, best);
    const double a = G.a0 + (static_cast<double>(Ci) + 0.5) * (G.h * kHprCoarse);
    const double b = G.b0 + (static_cast<double>(Cj) + 0.5) * (G.h * kHprCoarse);
    const double inv = 1.0 / sqrt((a * a + b * b) + 1.0);
    Cdir[t] = a * inv;
    Cdir[n_coarse + t] = b * inv;
    Cdir[2 * n_coarse + t] = inv;
    Cell4[t] = make_float4(static_cast<float>(a * inv), static_cast<float>(b * inv), static_cast<float>(inv),
                           best ? __double2float_ru(__longlong_as_double(static_cast<long long>(best))) : 0.0f);
  }
}

// The other easy majority: a candidate deep behind a surface.  The outermost candidates of three cells around its own span
// a triangle; if p lies in the tetrahedron (origin, a, b, c) it is not a hull vertex -- the witness a search would end with,
// checked with the same filtered determinants, found without a search.  (A point ON the surface is not in it: the
// triangle's plane passes below the sphere by the sagitta of the cells' spacing.)  One LANE per candidate: the test is a few
// hundred scalar operations, which a wavefront per candidate would spend 64 times over.  On hidden-heavy keyframes of C3 it
// settles 88-94 % of the hidden candidates (75-89 % with the first four triangles alone; profiles/hpr_quick_sim.py); the rest
// stay kStUndecided for k_hpr_radial / k_hpr_decide.
__global__ __launch_bounds__(kHprBlock) void k_hpr_quick(HprArrays A, HprGrid G, uint8_t *__restrict__ state,
                                                         unsigned long long *__restrict__ stats) {
  const int32_t j = static_cast<int32_t>(blockIdx.x) * kHprBlock + static_cast<int32_t>(threadIdx.x);
  bool hit = false;
  if (j < G.m) {
    const Vec3d p = load_point(A, j);
    const int32_t cell = A.scell[j];
    const int32_t ci = cell % G.gw, cj = cell / G.gw;
    auto rep_of = [&](int di, int dj) -> int32_t {
      const int32_t i = ci + di, jj = cj + dj;
      if (i < 0 || jj < 0 || i >= G.gw || jj >= G.gh) return -1;
      const unsigned long long v = A.crep[jj * G.gw + i];
      return v ? static_cast<int32_t>(v & 0x3ffffffull) : -1;
    };
    // four triangles of neighbouring cells that enclose the candidate's own cell
    const int8_t tri[4][6] = {{-1, -1, 1, -1, 0, 1}, {-1, 1, 1, 1, 0, -1}, {-1, -1, -1, 1, 1, 0}, {1, -1, 1, 1, -1, 0}};
    for (int t = 0; t < 4 && !hit; ++t) {
      const int32_t ia = rep_of(tri[t][0], tri[t][1]);
      const int32_t ib = rep_of(tri[t][2], tri[t][3]);
      const int32_t ic = rep_of(tri[t][4], tri[t][5]);
      if (ia < 0 || ib < 0 || ic < 0 || ia == j || ib == j || ic == j) continue;
      hit = tetra_contains_filtered(p, load_point(A, ia), load_point(A, ib), load_point(A, ic)) != 0;
    }
    // then the fan around the outermost candidate of the candidate's OWN cell: with two consecutive cells of the ring it
    // spans the small triangles that lie right above a candidate just behind the surface (on C3 the four triangles leave
    // 11-25 % of a keyframe's hidden candidates to the searches, the fan half of that)
    const int32_t io = hit ? -1 : rep_of(0, 0);
    if (io >= 0 && io != j) {
      const Vec3d o = load_point(A, io);
      const int8_t ring[9][2] = {{-1, -1}, {0, -1}, {1, -1}, {1, 0}, {1, 1}, {0, 1}, {-1, 1}, {-1, 0}, {-1, -1}};
      int32_t ia = rep_of(ring[0][0], ring[0][1]);
      Vec3d a = load_point(A, max(ia, 0));
      for (int t = 1; t < 9 && !hit; ++t) {
        const int32_t ib = rep_of(ring[t][0], ring[t][1]);
        const Vec3d b = load_point(A, max(ib, 0));
        if (ia >= 0 && ib >= 0 && ia != j && ib != j) hit = tetra_contains_filtered(p, o, a, b) != 0;
        ia = ib;
        a = b;
      }
    }
    state[j] = static_cast<uint8_t>(hit ? kStHidden : kStUndecided);
  }
  (void)stats;  // (its verdicts are counted with everyone's, from the final states)
}

// ------------------------------------------------------------------------------------------------------------------
// The easy majority first: a candidate whose RADIAL plane (n = p / |p|, the first trial normal of every search) already
// has every other point strictly on its inner side is a hull vertex, and nearly every visible candidate is of that kind
// (the tangent plane of the flipped sphere leaves its neighbours by |q| delta^2 / 2).  That needs no polygon and no cuts --
// only the traversal -- so four candidates share a wavefront here, 16 lanes each: the wave-uniform arithmetic of a search
// (frame, window, bounds) is paid once per four candidates, the cell and point tests run 16 at a time.  Same certificate as
// k_hpr_decide's (same plane, same rounding-proof point test, same cell bound); whatever this pass cannot certify -- an
// uncleared point, a duplicate -- stays kStUndecided and k_hpr_decide searches it as before.
// ------------------------------------------------------------------------------------------------------------------
// `todo` (nullable): the candidates k_hpr_quick left undecided, as a list (k_hpr_list; its length in the tallies): the four rows
// of a wavefront then all have work on keyframes where most candidates are hidden and certified already.
// ---- the point test of the 16- and 64-lane passes ----
// A point q is strictly on the inner side of the trial plane when n . (q - p) < 0 EXACTLY.  Computed: d = fl(q - p) per
// coordinate, t = fma(n_z, d_z, fma(n_y, d_y, fl(n_x d_x))); |t - exact| <= 4.44e-16 T with T = sum |n_i d_i| (the bound the
// searches have used since round 3, kPointSlack = 1e-15 covers it with or without the fused form).  The test t < -1e-15 T costs
// three multiplies, two adds with |.|, a multiply and a compare per point beside t itself -- and T <= |n| |q - p| <= nn_hi x
// 2 rho_max (Cauchy-Schwarz; rho_max bounds every candidate's norm) is a constant of the trial plane: a point with
// t < -1e-15 nn_hi 2 rho_max (~ -4e-10 m for an untilted plane) is clear whatever its T.  Clearances are millimetres to
// metres, so the relative test -- and the test for an identical point, whose t is 0 -- only run for the lanes the absolute
// one leaves (round 5: 7 vector instructions per tested point instead of 17).
__device__ __forceinline__ double point_side(const Vec3d &n, double dx, double dy, double dz) {
  return __builtin_fma(n.z, dz, __builtin_fma(n.y, dy, n.x * dx));
}
__device__ __forceinline__ double abs_clear_of(double nn_hi, double two_rho) { return -kPointSlack * (nn_hi * two_rho); }
// the relative test (T returned for the callers that rank violators)
__device__ __forceinline__ bool point_cleared_rel(const Vec3d &n, double dx, double dy, double dz, double t, double &T) {
  T = (fabs(n.x * dx) + fabs(n.y * dy)) + fabs(n.z * dz);
  return t < -kPointSlack * T;
}

// ---- the least-norm tilt of a trial plane (shared by k_hpr_radial<true> and k_hpr_tilt; described at k_hpr_tilt) ----
constexpr double kTiltMargin = 1.0e-9;  // the trial tilt stays this far (rad) inside every half-plane it knows: n . d <= -1e-9 |D| against a rounding bound of ~1e-13
constexpr int kTiltMaxSteps = 200;      // half-planes added per search (a dense cluster next to the candidate: 41 seen on C3)
constexpr int kTiltMaxPasses = 8;       // traversals per search
constexpr int kTiltWideWindow = 128;    // mid cells: a wider window is walked in coarse cells

// the maximum over the 16 lanes of a row, in every lane of the row (four rotations inside the row)
__device__ __forceinline__ float row_max16(float v) {
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false)));  // row_ror:1
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, false)));  // row_ror:2
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false)));  // row_ror:4
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false)));  // row_ror:8
  return v;
}

struct TiltLine {
  double x, y, f;  // unit normal and offset: inside is s . (x, y) <= f (the margin is in f)
  int32_t id;      // the point whose half-plane it is (place in cell order)
};

struct Tilt {
  double sx, sy;
  int na;
  TiltLine a, b;
};

__device__ __forceinline__ bool tilt_inside(const TiltLine &c, double sx, double sy) {
  const double u = c.x * sx, v = c.y * sy;
  return (u + v) <= c.f + 1.0e-12 * ((fabs(u) + fabs(v)) + fabs(c.f));
}

// 1: s moved (active set replaced); 2: (a, b, c) have nothing in common; 3: no conclusion
__device__ __forceinline__ int tilt_add(Tilt &T, double Dx, double Dy, double E, int32_t id) {
  const double len = sqrt(Dx * Dx + Dy * Dy);
  if (!(len > 0.0) || !isfinite(len)) return 3;
  const double inv = quick_rcp(len);
  TiltLine c = {Dx * inv, Dy * inv, E * inv - kTiltMargin, id};
  // the three places where c can be tight; the least-norm one that the other lines allow
  double bx = 0.0, by = 0.0, bn = INFINITY;
  int pick = -1;
  {
    const double x = c.f * c.x, y = c.f * c.y;
    if ((T.na < 1 || tilt_inside(T.a, x, y)) && (T.na < 2 || tilt_inside(T.b, x, y))) {
      bx = x;
      by = y;
      bn = x * x + y * y;
      pick = 0;
    }
  }
  if (T.na >= 1) {
    double x, y;
    if (line_meet(c.x, c.y, c.f, T.a.x, T.a.y, T.a.f, x, y) && (T.na < 2 || tilt_inside(T.b, x, y))) {
      const double n2 = x * x + y * y;
      if (n2 < bn) {
        bx = x;
        by = y;
        bn = n2;
        pick = 1;
      }
    }
  }
  if (T.na >= 2) {
    double x, y;
    if (line_meet(c.x, c.y, c.f, T.b.x, T.b.y, T.b.f, x, y) && tilt_inside(T.a, x, y)) {
      const double n2 = x * x + y * y;
      if (n2 < bn) {
        bx = x;
        by = y;
        bn = n2;
        pick = 2;
      }
    }
  }
  if (pick < 0) return T.na == 2 ? 2 : 3;
  if (!(bn < kHprBox * kHprBox)) return 3;
  if (pick == 0) {
    T.a = c;
    T.na = 1;
  } else if (pick == 1) {
    T.b = c;
    T.na = 2;
  } else {
    T.a = c;
    T.na = 2;
  }
  T.sx = bx;
  T.sy = by;
  return 1;
}

__device__ __forceinline__ void tilt_normal(Search &S, const Tilt &T) {
  S.n = {S.e0.x + (T.sx * S.e1.x + T.sy * S.e2.x), S.e0.y + (T.sx * S.e1.y + T.sy * S.e2.y), S.e0.z + (T.sx * S.e1.z + T.sy * S.e2.z)};
  const double nn = sqrt((S.n.x * S.n.x + S.n.y * S.n.y) + S.n.z * S.n.z);
  const double inv = quick_rcp(nn);
  S.nh = {S.n.x * inv, S.n.y * inv, S.n.z * inv};
  S.nn_hi = nn * (1.0 + 1.0e-14);
  S.hp_lo = ((S.n.x * S.p.x + S.n.y * S.p.y) + S.n.z * S.p.z) * (1.0 - 1.0e-13);
}

constexpr int kStatRadial = 22;  // block 0 of the tallies: length of that list
// kOneStep (round 5): the same pass for the candidates the radial plane FAILED, with the plane tilted once.  What the radial
// pass leaves undecided are mostly visible candidates whose supporting plane leans with the local surface, and on C3 a visible
// search of k_hpr_tilt adds 1.1-2.4 half-planes on average: most of them are settled by the FIRST one.  So before the searches
// (LDS records, a dual active set, 168 VGPRs, 3 wavefronts per SIMD) this pass probes the 3 x 3 near cells with the radial
// plane, takes the worst violator's half-plane, moves to the least tilted plane that clears it by the searches' margin
// (tilt_add on an empty active set) and runs the ordinary traversal with THAT plane: a plane that has every other point
// strictly on its inner side is a witness whatever produced it (same point test, same cell bound).  Whatever it cannot
// certify stays undecided for k_hpr_tilt.  `stat_len`: the word of block 0 that holds the length of `todo`.
template <bool kOneStep, bool kTally /* PCP_HPR_DEBUG: trial normals and batches counted (the counting is 6-15 % of the pass) */>
__global__ __launch_bounds__(kHprBlock) void k_hpr_radial(HprArrays A, HprGrid G, uint8_t *__restrict__ state,
                                                          const int32_t *__restrict__ todo, unsigned long long *__restrict__ stats,
                                                          int32_t stat_len) {
  const int lane = lane_id();
  const int rl = lane & 15, row_base = lane & 48;
  const int32_t u = static_cast<int32_t>(blockIdx.x) * (kHprBlock / 16) + static_cast<int32_t>(threadIdx.x >> 4);
  const int32_t count = todo ? static_cast<int32_t>(stats[stat_len]) : G.m;
  if (static_cast<int32_t>(blockIdx.x) * (kHprBlock / 16) >= count) return;  // (the grid covers every candidate)
  const bool have = u < count;
  const int32_t j = have ? (todo ? todo[u] : u) : 0;
  auto row_mask = [&](bool b) -> uint32_t { return static_cast<uint32_t>((__ballot(b) >> row_base) & 0xffffull); };
  Search S;
  S.self = have ? j : 0;
  S.p = load_point(A, S.self);
  S.self_idx = A.sidx[S.self];
  search_frame(S);
  S.n = S.e0;
  {
    // e0 = p / |p| to a few ulp (search_frame): |n| = 1 within 1e-15, so n stands for its own unit vector (the cell bound
    // carries 1e-12 of slack for that) and 1 + 1e-13 bounds its norm -- no second square root and reciprocal per candidate
    S.nh = S.n;
    S.nn_hi = 1.0 + 1.0e-13;
    S.hp_lo = ((S.n.x * S.p.x + S.n.y * S.p.y) + S.n.z * S.p.z) * (1.0 - 1.0e-13);
  }
  bool open = have && state[S.self] == kStUndecided;  // the row may still certify its candidate (k_hpr_quick may have settled it)
  const bool mine_to_write = open;
  bool hidden_dup = false;
  unsigned long long batches = 0;
  const double two_rho = 2.0 * *G.rho_max;
  double abs_clear = abs_clear_of(S.nn_hi, two_rho);  // t below this: the point is strictly inside whatever its T (point_side)
  const int32_t cell = A.scell[S.self];
  const int32_t ci = cell % G.gw, cj = cell / G.gw;
  if (kOneStep) {
    // the worst violator of the radial plane among the 3 x 3 near cells (largest n . d relative to its rounding bound, as the
    // searches choose), one per lane, then over the row
    float best = -INFINITY;
    double bx = 0.0, by = 0.0, bz = 0.0;
    int32_t bk = -1;
    for (int dj = -1; dj <= 1; ++dj) {
      const int32_t rj = cj + dj;
      const bool in = rj >= 0 && rj < G.gh;
      const int32_t c0 = (in ? rj : cj) * G.gw + max(ci - 1, 0), c1 = (in ? rj : cj) * G.gw + min(ci + 1, G.gw - 1);
      const int32_t k0 = A.cstart[c0], k1 = A.cstart[c1 + 1];
      for (int32_t base = k0; __ballot(in && open && base < k1); base += 16) {
        const int32_t k = base + rl;
        if (in && open && k < k1 && k != S.self) {
          const double dx = A.sx[k] - S.p.x, dy = A.sy[k] - S.p.y, dz = A.sz[k] - S.p.z;
          const double t = point_side(S.n, dx, dy, dz);
          double T = 0.0;
          if (!(t < abs_clear) && !(dx == 0.0 && dy == 0.0 && dz == 0.0) && !point_cleared_rel(S.n, dx, dy, dz, t, T)) {
            const float score = static_cast<float>(t) * __builtin_amdgcn_rcpf(static_cast<float>(T));
            if (score > best) {
              best = score;
              bx = dx;
              by = dy;
              bz = dz;
              bk = k;
            }
          }
        }
        if (kTally && in && open && base < k1 && rl == 0) batches += 1;
      }
    }
    const float worst = row_max16(best);
    const uint32_t at = row_mask(open && bk >= 0 && best == worst);
    if (!at) {
      open = false;  // nothing near violates the radial plane (a far point did): the searches' business
    } else {
      const int src = row_base + __builtin_ctz(at);
      const double qx = __shfl(bx, src, 64), qy = __shfl(by, src, 64), qz = __shfl(bz, src, 64);
      const double Dx = (qx * S.e1.x + qy * S.e1.y) + qz * S.e1.z;
      const double Dy = (qx * S.e2.x + qy * S.e2.y) + qz * S.e2.z;
      const double E = -((qx * S.e0.x + qy * S.e0.y) + qz * S.e0.z);
      Tilt T;
      T.sx = T.sy = 0.0;
      T.na = 0;
      T.a = T.b = {0.0, 0.0, 0.0, -1};
      if (tilt_add(T, Dx, Dy, E, __shfl(bk, src, 64)) == 1) {
        tilt_normal(S, T);
        abs_clear = abs_clear_of(S.nn_hi, two_rho);
      } else {
        open = false;
      }
    }
  }
  // points [k0, k1) of the cell order against the plane, 16 at a time; `go`: this row takes part
  auto test_points = [&](bool go, int32_t k0, int32_t k1) {
    for (int32_t base = k0; __ballot(go && open && base < k1); base += 16) {
      const int32_t k = base + rl;
      const bool active = go && open && k < k1 && k != S.self;
      bool bad = false, dup_lower = false;
      if (active) {
        const double dx = A.sx[k] - S.p.x, dy = A.sy[k] - S.p.y, dz = A.sz[k] - S.p.z;
        const double t = point_side(S.n, dx, dy, dz);
        if (!(t < abs_clear)) {  // rare: not clear by the absolute bound (a duplicate has t = 0)
          double T;
          if (dx == 0.0 && dy == 0.0 && dz == 0.0)
            dup_lower = A.sidx[k] < S.self_idx;  // identical flipped points: the lowest input index stands for the group
          else
            bad = !point_cleared_rel(S.n, dx, dy, dz, t, T);
        }
      }
      if (kTally && go && open && base < k1 && rl == 0) batches += 1;
      if (__ballot(bad || dup_lower)) {  // (rare: one ballot in front of the two row masks)
        if (row_mask(dup_lower)) {
          hidden_dup = true;
          open = false;
        }
        if (row_mask(bad)) open = false;
      }
    }
  };
  // the 3 x 3 cells around the candidate's own
  for (int dj = -1; dj <= 1; ++dj) {
    const int32_t rj = cj + dj;
    const bool in = rj >= 0 && rj < G.gh;
    const int32_t c0 = (in ? rj : cj) * G.gw + max(ci - 1, 0), c1 = (in ? rj : cj) * G.gw + min(ci + 1, G.gw - 1);
    test_points(in, A.cstart[c0], A.cstart[c1 + 1]);
  }
  // every other cell the bound cannot clear: mid cells (4 x 4 fine cells) of the window, 16 at a time, then the 16 fine cells
  // of a mid cell that stays open -- one row of tests each.  (Through the coarse cells of the 64-lane search an open cell
  // cost four rows of fine tests, most of them on cells nowhere near the plane: k_hpr_radial is bound by its arithmetic.)
  const Window W = reach_window<kHprMid>(S, G);
  const bool has_window = open && W.i0 <= W.i1 && W.j0 <= W.j1;
  const int32_t ww = has_window ? W.i1 - W.i0 + 1 : 1, wn = has_window ? ww * (W.j1 - W.j0 + 1) : 0;
  for (int32_t cb = 0; __ballot(open && cb < wn); cb += 16) {
    const int32_t t = cb + rl;
    const int32_t tq = small_div(t, ww);
    const int32_t Ci = W.i0 + (t - tq * ww), Cj = W.j0 + tq;  // (kept apart: the fine cells need them, not the linear index)
    const int32_t C = (open && t < wn) ? Cj * G.mgw + Ci : -1;
    bool copen = false;
    if (C >= 0) {
      const float4 c4 = A.Mid4[C];
      copen = c4.w > 0.0f && !cell_cleared_f32sep(S.nh.x, S.nh.y, S.nh.z, S.nn_hi, S.hp_lo, c4.x, c4.y, c4.z, c4.w, G.r_mid + kCell4Slack);
    }
    uint32_t open_c = row_mask(copen);
    while (__ballot(open && open_c != 0u)) {
      const bool go_c = open && open_c != 0u;
      const int bc = go_c ? __builtin_ctz(open_c) : 0;
      open_c &= open_c - 1u;
      const int32_t Cci = __shfl(Ci, row_base + bc, 64), Ccj = __shfl(Cj, row_base + bc, 64);
      {
        const int32_t fi = (go_c ? Cci : 0) * kHprMid + (rl & 3), fj = (go_c ? Ccj : 0) * kHprMid + (rl >> 2);
        bool fopen = false;
        int32_t f = 0;
        if (go_c && fi < G.gw && fj < G.gh && !(abs(fi - ci) <= 1 && abs(fj - cj) <= 1)) {
          f = fj * G.gw + fi;
          const float4 c4 = A.cell4[f];
          fopen = c4.w > 0.0f && !cell_cleared_f32sep(S.nh.x, S.nh.y, S.nh.z, S.nn_hi, S.hp_lo, c4.x, c4.y, c4.z, c4.w, G.r_fine + kCell4Slack);
        }
        uint32_t open_f = row_mask(fopen);
        while (__ballot(open && open_f != 0u)) {
          const bool go_f = open && open_f != 0u;
          const int bf = go_f ? __builtin_ctz(open_f) : 0;
          // open cells side by side in one row of the mid cell are one run of the cell order: one range, fuller batches
          const int len = go_f ? min(kHprMid - (bf & (kHprMid - 1)), __builtin_ctz(~(open_f >> bf))) : 1;
          open_f &= ~(((1u << len) - 1u) << bf);
          const int32_t ff = __shfl(f, row_base + bf, 64);
          test_points(go_f, go_f ? A.cstart[ff] : 0, go_f ? A.cstart[ff + len] : 0);
        }
      }
    }
  }
  unsigned long long work = 0;  // this row's trial normal (if it decided) and batches, packed
  if (mine_to_write && rl == 0) {
    const int32_t out = hidden_dup ? kStHidden : (open ? kStVisible : kStUndecided);
    state[j] = static_cast<uint8_t>(out);
    work = (out != kStUndecided ? 1ull : 0ull) + (batches << kStatPackedShift);
  }
  if (kTally) {  // the wavefront's four rows as one atomic
    work += __shfl_xor(work, 16, 64);
    work += __shfl_xor(work, 32, 64);
    if (lane == 0 && work) atomicAdd(&stats[kStatStride * (1 + (blockIdx.x % kStatCopies)) + kStatPacked], work);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// The search itself, 16 lanes per candidate (round 4).  What the radial pass leaves are candidates whose least-tilted
// supporting plane is not the radial one (or that have none).  Their feasibility problem -- find s with s . D_q < E_q for
// every q -- is solved here as the LEAST-NORM problem  min |s|^2  s.t.  s . D_q <= E_q - margin |D_q|  by the dual
// active-set method: the trial tilt s is the least-norm point of at most two "active" half-planes (s = 0: none, the radial
// plane); a point the trial plane does not clear adds its half-plane, the least-norm point of (active + new) is taken from
// the three places where the new line can be tight (alone, or meeting one of the active lines), and the half-planes that
// are tight there become the active set.  |s| grows strictly with every step, so no active set comes back; three
// half-planes with nothing in common end the search with the same witness the polygon search ends with -- p inside the
// tetrahedron (origin, q_a, q_b, q_c) -- CHECKED by the same filtered determinants, and a trial plane is a witness of
// visibility only after one whole traversal in which it did not move: the same rounding-proof point test and cell bound
// as k_hpr_radial / k_hpr_decide.  So the state of a search is two doubles and two lines -- no polygon, no cross-lane
// bookkeeping --, four searches share a wavefront, and the arithmetic every lane of a 64-lane search repeated for one
// candidate (frame, trial normal, window) now serves four.  Whatever ends otherwise (round-off, a parallel pair, a cap)
// stays kStUndecided for k_hpr_decide, whose polygon remembers every half-plane it has seen.
// ------------------------------------------------------------------------------------------------------------------
constexpr int kStatTilt = 23;  // block 0 of the tallies: length of k_hpr_tilt's list
constexpr int kStatOneStep = 30;   // block 0: length of the list of k_hpr_radial<true>
constexpr int kStatTiltCont = 29;  // block 0: length of the list of searches the 16-lane rows handed on to a wavefront each
// PCP_HPR_DEBUG only: searches by the binary logarithm of their round trips [0..15], the round trips of each class [16..31], and
// per wavefront the rows' round trips summed [32] against (rows x the longest row's) [33] (what lockstep rows cost),
// wavefronts [34], rows of cell tests [35]
__device__ unsigned long long g_tilt_hist[40];
// The state of a search lives in LDS, one record per row (every lane of the row writes the same values, so each thread
// reads what it wrote itself): carried in registers through the eight loop levels below it cost a copy per level -- the
// first build of this kernel took 263 VGPRs (one wavefront per SIMD) for ~100 registers of state.
struct TiltRow {
  double n[3], nh[3], nn_hi, hp_lo;  // trial normal (tilt_normal)
  double abs_clear;                  // ... and the absolute bound of its point test (point_side)
  double e0[3], e1[3], e2[3];        // frame at p
  double sx, sy;                     // trial tilt
  double ax, ay, af, bx, by, bf;     // active lines
  int32_t aid, bid, na, last_id;
};

// A search handed on (round 5).  The search is a serial chain of round trips -- a batch of loads, a test, a ballot --, a launch
// lasts as long as its longest chain, and a chain of 16-lane steps can be a thousand long while the bulk takes 4-60 (the
// launches were tails: 1.06 resident wavefronts per SIMD of 3, profiles/r04_hpr_pmc.json).  So a row stops after `budget` round
// trips and writes down what the search has found so far -- the trial tilt and the active half-planes, which IS its state: the
// least-norm point of at most two half-planes of real points, |s| only ever grows -- and a second launch continues every such
// search on a whole wavefront: the same algorithm on rows of 64 lanes (a quarter of the round trips, coarse cells of 8 x 8
// fine ones as the upper level), starting its passes over with the plane it was handed.  Every verdict still rests on the
// same certificates (one whole pass with an unmoved plane / the checked tetrahedron).
struct TiltCont {
  double sx, sy, ax, ay, af, bx, by, bf;
  int32_t j, na, aid, bid;
};

#define TILT_STORE_NORMAL(R, S) \
  do {                         \
    R.n[0] = S.n.x;            \
    R.n[1] = S.n.y;            \
    R.n[2] = S.n.z;            \
    R.nh[0] = S.nh.x;          \
    R.nh[1] = S.nh.y;          \
    R.nh[2] = S.nh.z;          \
    R.nn_hi = S.nn_hi;         \
    R.hp_lo = S.hp_lo;         \
    R.abs_clear = abs_clear_of(S.nn_hi, two_rho); \
  } while (0)

#ifndef PCP_TILT_WPE
#define PCP_TILT_WPE 3
#endif
// Workgroups of ONE wavefront (round 5).  A workgroup's wave slots are given back when its LAST wavefront ends, so with four
// wavefronts to a workgroup the lockstep group was sixteen searches, not four: the slots of the three wavefronts that had
// finished stood empty behind the longest (1.06 resident wavefronts per SIMD of 3, profiles/r04_hpr_pmc.json; k_hpr_decide
// had shown the same in round 3).
#ifndef PCP_TILT_BLOCK
#define PCP_TILT_BLOCK 64
#endif
constexpr int kTiltBlock = PCP_TILT_BLOCK;
// kRow = 16: four searches per wavefront, from the list `todo`, each with a budget of round trips; kRow = 64: one search per
// wavefront, from the records `cont` the first launch wrote (budget: none).
template <bool kDebug, int kRow>
__global__ __launch_bounds__(kTiltBlock) __attribute__((amdgpu_waves_per_eu(PCP_TILT_WPE, PCP_TILT_WPE))) void k_hpr_tilt(
    HprArrays A, HprGrid G, uint8_t *__restrict__ state, const int32_t *__restrict__ todo, unsigned long long *__restrict__ stats,
    int32_t wide_window, int32_t *__restrict__ left_over, TiltCont *__restrict__ cont, int32_t budget) {
  static_assert(kRow == 16 || kRow == 64, "rows of 16 or 64 lanes");
  constexpr int kRowsPerWave = 64 / kRow, kRowsPerBlock = kTiltBlock / kRow;
  // Really in LDS: a compiler barrier in front of every group of reads keeps the compiler from forwarding the stores to them
  // (and so from carrying the record in registers after all); not `volatile`, which turns the accesses into flat ones with an
  // address pair per field (60 VGPRs).
  __shared__ TiltRow rows[kRowsPerBlock];
#define R rows[threadIdx.x / kRow]
#define TILT_FENCE() asm volatile("" ::: "memory")
  const int lane = lane_id();
  const int rl = lane & (kRow - 1), row_base = lane & (64 - kRow);
  const int32_t count = static_cast<int32_t>(stats[kRow == 16 ? kStatTilt : kStatTiltCont]);
  const int32_t rows_total = static_cast<int32_t>(gridDim.x) * kRowsPerBlock;
  using mask_t = typename std::conditional<kRow == 64, unsigned long long, uint32_t>::type;  // a row's lanes as bits
  auto row_mask = [&](bool b) -> mask_t {
    const unsigned long long m = __ballot(b);
    return static_cast<mask_t>(kRow == 64 ? m : ((m >> row_base) & 0xffffull));
  };
  auto first_bit = [](mask_t m) -> int { return kRow == 64 ? static_cast<int>(__builtin_ctzll(m)) : __builtin_ctz(static_cast<uint32_t>(m)); };
  auto row_max = [&](float v) -> float {
    v = row_max16(v);
    if (kRow == 64) {
      const int iv = __float_as_int(v);
      v = fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(iv, 0)), __int_as_float(__builtin_amdgcn_readlane(iv, 16))),
                fmaxf(__int_as_float(__builtin_amdgcn_readlane(iv, 32)), __int_as_float(__builtin_amdgcn_readlane(iv, 48))));
    }
    return v;
  };
  const int32_t n_coarse = G.cgw * G.cgh;
  const double two_rho = 2.0 * *G.rho_max;
  for (int32_t u0 = static_cast<int32_t>(blockIdx.x) * kRowsPerBlock + static_cast<int32_t>(threadIdx.x >> 6) * kRowsPerWave; u0 < count;
       u0 += rows_total) {  // (u0: the first row of this wavefront; uniform over the wavefront)
    const int32_t u = u0 + lane / kRow;
    const bool have = u < count;
    TiltCont c0 = {};
    if (kRow == 64 && have) c0 = cont[u];
    const int32_t j = have ? (kRow == 16 ? todo[u] : c0.j) : 0;
    const int32_t self = j, self_idx = A.sidx[j];
    const Vec3d p = load_point(A, j);
    {
      Search S;
      S.p = p;
      search_frame(S);
      Tilt T;
      T.sx = T.sy = 0.0;
      T.na = 0;
      T.a = T.b = {0.0, 0.0, 0.0, -1};
      if (kRow == 64 && have) {  // the search so far
        T.sx = c0.sx;
        T.sy = c0.sy;
        T.na = c0.na;
        T.a = {c0.ax, c0.ay, c0.af, c0.aid};
        T.b = {c0.bx, c0.by, c0.bf, c0.bid};
      }
      tilt_normal(S, T);
      TILT_STORE_NORMAL(R, S);
      R.e0[0] = S.e0.x; R.e0[1] = S.e0.y; R.e0[2] = S.e0.z;
      R.e1[0] = S.e1.x; R.e1[1] = S.e1.y; R.e1[2] = S.e1.z;
      R.e2[0] = S.e2.x; R.e2[1] = S.e2.y; R.e2[2] = S.e2.z;
      R.sx = T.sx;
      R.sy = T.sy;
      R.ax = T.a.x; R.ay = T.a.y; R.af = T.a.f; R.aid = T.a.id;
      R.bx = T.b.x; R.by = T.b.y; R.bf = T.b.f; R.bid = T.b.id;
      R.na = T.na;
      R.last_id = -1;
      TILT_FENCE();
    }
    bool run = have;        // the row is still searching
    bool changed = false;   // the trial plane moved during this pass
    int outcome = 0;        // 0 none, 1 visible, 2 hidden duplicate, 3 empty (a, b, c below), 4 gave up, 5 handed on (budget)
    int32_t ca = -1, cb = -1, cc = -1;
    int steps = 0, passes = 0, why4 = 0, wide = 0, last_wn = 0;
    int32_t trips = 0;      // round trips of this row so far: batches of point tests + rows of cell tests (row-uniform)
    unsigned long long batches = 0, csteps = 0;  // (tallies, lane 0 of the row; csteps: PCP_HPR_DEBUG only)
    auto stop = [&](int why) {
      outcome = why;
      run = false;
    };
    // one more half-plane: the difference vector (qx, qy, qz) of point qid.  The record in LDS is read, advanced and written back.
    auto add_half_plane = [&](double qx, double qy, double qz, int32_t qid) {
      TILT_FENCE();
      Search S;
      S.p = p;
      S.e0 = {R.e0[0], R.e0[1], R.e0[2]};
      S.e1 = {R.e1[0], R.e1[1], R.e1[2]};
      S.e2 = {R.e2[0], R.e2[1], R.e2[2]};
      const double Dx = (qx * S.e1.x + qy * S.e1.y) + qz * S.e1.z;
      const double Dy = (qx * S.e2.x + qy * S.e2.y) + qz * S.e2.z;
      const double E = -((qx * S.e0.x + qy * S.e0.y) + qz * S.e0.z);
      Tilt T;
      T.sx = R.sx;
      T.sy = R.sy;
      T.na = R.na;
      T.a = {R.ax, R.ay, R.af, R.aid};
      T.b = {R.bx, R.by, R.bf, R.bid};
      const int32_t pa = T.a.id, pb = T.b.id;
      const int r = tilt_add(T, Dx, Dy, E, qid);
      if (r == 1) {
        changed = true;
        tilt_normal(S, T);
        TILT_STORE_NORMAL(R, S);
        R.sx = T.sx;
        R.sy = T.sy;
        R.na = T.na;
        R.ax = T.a.x; R.ay = T.a.y; R.af = T.a.f; R.aid = T.a.id;
        R.bx = T.b.x; R.by = T.b.y; R.bf = T.b.f; R.bid = T.b.id;
        TILT_FENCE();
      } else if (r == 2) {
        ca = pa;
        cb = pb;
        cc = qid;
        stop(3);
      } else {
        why4 = 2;
        stop(4);
      }
    };
    // points [k0, k1) of the cell order against the trial plane, kRow at a time; `go`: this row takes part
    auto test_points = [&](bool go, int32_t k0, int32_t k1) {
      for (int32_t base = k0; __ballot(go && run && base < k1); base += kRow) {
        if (go && run && base < k1 && trips >= budget) stop(5);  // (row-uniform)
        const bool on = go && run && base < k1;
        if (on) ++trips;
        const int32_t k = base + rl;
        bool active = on && k < k1 && k != self;
        double dx = 0.0, dy = 0.0, dz = 0.0;
        if (active) {
          dx = A.sx[k] - p.x;
          dy = A.sy[k] - p.y;
          dz = A.sz[k] - p.z;
        }
        if (kDebug && on && rl == 0) batches += 1;
        for (int guard = 0;; ++guard) {
          TILT_FENCE();
          const Vec3d nn = {R.n[0], R.n[1], R.n[2]};
          const double t = point_side(nn, dx, dy, dz);
          bool bad = run && active && !(t < R.abs_clear);  // (point_side: clear by the absolute bound whatever its T)
          if (!__ballot(bad)) break;
          // rare: the relative test, and the test for an identical flipped point (t = 0: it never passes the absolute one) --
          // the lowest input index stands for a group of identical points
          double Tt = 1.0;
          bool dup_lower = false;
          if (bad) {
            if (dx == 0.0 && dy == 0.0 && dz == 0.0) {
              dup_lower = A.sidx[k] < self_idx;
              active = false;  // not a constraint: out of this batch
              bad = false;
            } else {
              bad = !point_cleared_rel(nn, dx, dy, dz, t, Tt);
            }
          }
          if (row_mask(dup_lower) && on) stop(2);
          bad = bad && run;
          if (!__ballot(bad)) break;
          const mask_t rm = row_mask(bad);
          if (rm) {
            // the worst of the row's points (largest n . d relative to its bound; the choice only steers)
            const float score = bad ? static_cast<float>(t) * __builtin_amdgcn_rcpf(static_cast<float>(Tt)) : -INFINITY;
            const float worst = row_max(score);
            const mask_t at = row_mask(bad && score == worst);
            const int src = row_base + first_bit(at ? at : rm);
            const double qx = __shfl(dx, src, 64), qy = __shfl(dy, src, 64), qz = __shfl(dz, src, 64);
            const int32_t qid = base + (src - row_base);
            // a point that is not cleared right after its own half-plane was added lies within round-off of every plane the
            // margin allows: not for this pass
            TILT_FENCE();
            const int32_t last_id = R.last_id;
            if (qid == last_id || ++steps > kTiltMaxSteps || guard > 24) {
              why4 = qid == last_id ? 0 : 1;
              stop(4);
            } else {
              R.last_id = qid;
              add_half_plane(qx, qy, qz, qid);
            }
          }
        }
      }
    };
    auto cleared = [&](double ux, double uy, double uz, double rho, double r) -> bool {
      TILT_FENCE();
      return cell_cleared_f32sep(R.nh[0], R.nh[1], R.nh[2], R.nn_hi, R.hp_lo, ux, uy, uz, rho, r);  // on the row's record
    };
    const int32_t cell = A.scell[self];
    const int32_t ci = cell % G.gw, cj = cell / G.gw;
    for (int pass = 0; __ballot(run); ++pass) {
      if (run && pass >= kTiltMaxPasses) {
        why4 = 3;
        stop(4);
      }
      if (run) ++passes;
      // the 3 x 3 cells around the candidate's own -- that is where the binding half-planes are -- until the plane rests there
      bool again = run;
      for (int nit = 0; __ballot(again); ++nit) {
        changed = false;
#pragma unroll 1
        for (int dj = -1; dj <= 1; ++dj) {
          const int32_t rj = cj + dj;
          const bool in = rj >= 0 && rj < G.gh;
          const int32_t c0n = (in ? rj : cj) * G.gw + max(ci - 1, 0), c1n = (in ? rj : cj) * G.gw + min(ci + 1, G.gw - 1);
          test_points(in && again, A.cstart[c0n], A.cstart[c1n + 1]);
        }
        if (run && again && changed && nit >= 12) {
          why4 = 3;
          stop(4);
        }
        again = again && run && changed;
      }
      // (rows still running: the near cells accept the plane, changed == false)
      // every other cell the bound cannot clear.  A point in there that moves the plane does not end the sweep: the rest of
      // the window is taken with the new plane (its half-planes are collected in this sweep instead of one per pass) and
      // the next pass starts over -- a plane is only a witness after a whole pass in which it did not move.
      // Rows of 16: the window in mid cells (4 x 4 fine cells: an open one costs ONE row of fine tests) -- unless the plane is
      // tilted so far that the window is wide: then the upper level is the coarse cells (8 x 8: a quarter of the upper rows, four
      // rows of fine tests per open cell).  Per row: `big` selects the level.  Rows of 64: always coarse cells (one row each).
      Window W;
      bool big = kRow == 64;
      {
        TILT_FENCE();
        Search S;
        S.nh = {R.nh[0], R.nh[1], R.nh[2]};
        S.nn_hi = R.nn_hi;
        S.hp_lo = R.hp_lo;
        if (kRow == 64) {
          W = reach_window<kHprCoarse>(S, G);
        } else {
          W = reach_window<kHprMid>(S, G);
          big = run && W.i0 <= W.i1 && (W.i1 - W.i0 + 1) * (W.j1 - W.j0 + 1) > wide_window;
          if (big) {  // the same window in coarse cells (a superset: the mid window's corners, halved)
            W.i0 >>= 1;
            W.j0 >>= 1;
            W.i1 = min(W.i1 >> 1, G.cgw - 1);
            W.j1 = min(W.j1 >> 1, G.cgh - 1);
          }
        }
      }
      const int32_t ugw = big ? G.cgw : G.mgw;
      const int eshift = big ? 3 : 2;                          // log2 of the upper cell's edge in fine cells
      const int fsteps = big ? 64 / kRow : 1;                  // rows of fine cells per upper cell
      const float4 *upper = big ? A.Cell4 : A.Mid4;
      const double r_upper = (big ? G.r_coarse : G.r_mid) + kCell4Slack;
      const bool has_window = run && W.i0 <= W.i1 && W.j0 <= W.j1;
      const int32_t ww = has_window ? W.i1 - W.i0 + 1 : 1, wn = has_window ? ww * (W.j1 - W.j0 + 1) : 0;
      if (kDebug) {
        if (wn > 256) ++wide;
        if (run) last_wn = wn;
      }
      for (int32_t cbk = 0; __ballot(run && cbk < wn); cbk += kRow) {
        if (run && cbk < wn && trips >= budget) stop(5);
        if (run && cbk < wn) ++trips;
        const int32_t t = cbk + rl;
        const int32_t tq = small_div(t, ww);
        const int32_t Ci = W.i0 + (t - tq * ww), Cj = W.j0 + tq;
        if (kDebug && run && cbk < wn && rl == 0) ++csteps;
        const int32_t C = (run && t < wn) ? Cj * ugw + Ci : -1;
        bool copen = false;
        if (C >= 0) {
          const float4 c4 = upper[C];
          copen = c4.w > 0.0f && !cleared(c4.x, c4.y, c4.z, c4.w, r_upper);
        }
        mask_t open_c = row_mask(copen);
        while (__ballot(run && open_c != 0)) {
          const bool go_c = run && open_c != 0;
          const int bc = go_c ? first_bit(open_c) : 0;
          open_c &= open_c - 1;
          const int32_t Cci = __shfl(Ci, row_base + bc, 64), Ccj = __shfl(Cj, row_base + bc, 64);
          for (int q4 = 0; __ballot(go_c && run && q4 < fsteps); ++q4) {  // the fine cells of the upper cell, kRow at a time
            if (go_c && run && q4 < fsteps && trips >= budget) stop(5);
            const bool go_q = go_c && run && q4 < fsteps;
            if (go_q) ++trips;
            if (kDebug && go_q && rl == 0) ++csteps;
            const int fidx = q4 * kRow + rl;
            const int32_t fi = ((go_q ? Cci : 0) << eshift) + (fidx & ((1 << eshift) - 1)), fj = ((go_q ? Ccj : 0) << eshift) + (fidx >> eshift);
            bool fopen = false;
            int32_t f = 0;
            if (go_q && fi < G.gw && fj < G.gh && !(abs(fi - ci) <= 1 && abs(fj - cj) <= 1)) {
              f = fj * G.gw + fi;
              const float4 c4 = A.cell4[f];
              fopen = c4.w > 0.0f && !cleared(c4.x, c4.y, c4.z, c4.w, G.r_fine + kCell4Slack);
            }
            mask_t open_f = row_mask(fopen);
            while (__ballot(run && open_f != 0)) {
              const bool go_f = run && open_f != 0;
              const int bf = go_f ? first_bit(open_f) : 0;
              // open cells side by side in one row of the upper cell are one run of the cell order: one range, fuller batches
              // (a cell holds ~8 candidates: alone it fills half a row of 16 lanes, an eighth of a row of 64)
              const int wide = 1 << eshift;
              const mask_t rest = static_cast<mask_t>(~(open_f >> bf));
              const int len = go_f ? min(wide - (bf & (wide - 1)), rest ? first_bit(rest) : wide) : 1;
              open_f &= static_cast<mask_t>(~(((static_cast<mask_t>(1) << (len - 1) << 1) - 1) << bf));
              const int32_t ff = __shfl(f, row_base + bf, 64);
              test_points(go_f, go_f ? A.cstart[ff] : 0, go_f ? A.cstart[ff + len] : 0);
            }
          }
        }
      }
      if (run && !changed) stop(1);  // one whole pass with a plane that did not move: every other point is strictly inside
    }
    // the witness of "hidden": p in the tetrahedron (origin, a, b, c) of the three half-planes with nothing in common
    int32_t out = kStUndecided;
    if (outcome == 1) out = kStVisible;
    else if (outcome == 2) out = kStHidden;
    if (__ballot(outcome == 3)) {
      if (outcome == 3 && ca >= 0 && cb >= 0 && cc >= 0 &&
          tetra_contains_filtered(p, load_point(A, ca), load_point(A, cb), load_point(A, cc)))
        out = kStHidden;
    }
    unsigned long long work = 0;
    if (have && rl == 0) {
      unsigned long long *mine = stats + kStatStride * (1 + (blockIdx.x % kStatCopies));
      (void)mine;
      work = static_cast<unsigned long long>(steps + 1) + (((batches * kRow + 63ull) / 64ull) << kStatPackedShift);
      if (out != kStUndecided) {
        state[j] = static_cast<uint8_t>(out);
      } else if (kRow == 16 && outcome == 5) {
        // out of round trips: the search continues on a wavefront of its own, from the plane it has reached
        TILT_FENCE();
        TiltCont c;
        c.sx = R.sx; c.sy = R.sy;
        c.ax = R.ax; c.ay = R.ay; c.af = R.af;
        c.bx = R.bx; c.by = R.by; c.bf = R.bf;
        c.j = j; c.na = R.na; c.aid = R.aid; c.bid = R.bid;
        cont[atomicAdd(&stats[kStatTiltCont], 1ull)] = c;
      } else {
        // what this pass gives up on (a handful per keyframe) goes straight onto the list of the polygon search
        left_over[atomicAdd(&stats[kStatSearch], 1ull)] = j;
      }
      if (kDebug) {  // PCP_HPR_DEBUG: what became of the searches (words 9.. of the copies; nothing else uses them)
        if (outcome == 4)
          printf("hpr: tilt%d gave up on %d: why %d passes %d steps %d batches %llu |s| %.3g na %d window %d coarse cells of %d, p = (%.17g, %.17g, %.17g)\n",
                 kRow, j, why4, passes, steps, batches, sqrt(R.sx * R.sx + R.sy * R.sy), R.na, last_wn, n_coarse, p.x, p.y, p.z);
        atomicAdd(&mine[9], 1ull);
        atomicAdd(&mine[10 + min(outcome, 4)], 1ull);  // 10 none 11 visible 12 duplicate 13 empty 14 gave up / handed on
        if (outcome == 3 && out == kStHidden) atomicAdd(&mine[15], 1ull);
        if (outcome == 4) atomicAdd(&mine[16 + min(why4, 3)], 1ull);  // 16 same point again 17 step cap 18 no conclusion 19 pass cap
        atomicAdd(&mine[20], static_cast<unsigned long long>(passes));
        atomicAdd(&mine[21], static_cast<unsigned long long>(steps));
        atomicAdd(&mine[22], batches);
        atomicMax(&mine[28], batches);
        if (outcome == 5) atomicAdd(&mine[29], 1ull);
        atomicAdd(&mine[30], static_cast<unsigned long long>(wide));
        const int hb = trips ? min(31 - __clz(trips), 15) : 0;
        atomicAdd(&g_tilt_hist[hb], 1ull);
        atomicAdd(&g_tilt_hist[16 + hb], static_cast<unsigned long long>(trips));
        atomicAdd(&g_tilt_hist[35], csteps);
      }
    }
    if (kDebug) {
      if (kRow == 16) {
        work += __shfl_xor(work, 16, 64);
        work += __shfl_xor(work, 32, 64);
      }
      if (lane == 0 && work) atomicAdd(&stats[kStatStride * (1 + (blockIdx.x % kStatCopies)) + kStatPacked], work);
    }
    if (kDebug) {
      unsigned long long tsum = 0, tmax = 0;
#pragma unroll
      for (int r4 = 0; r4 < kRowsPerWave; ++r4) {
        const unsigned long long tr = static_cast<unsigned long long>(__shfl(trips, r4 * kRow, 64));
        tsum += tr;
        tmax = max(tmax, tr);
      }
      if (lane == 0) {
        atomicAdd(&g_tilt_hist[32], tsum);
        atomicAdd(&g_tilt_hist[33], kRowsPerWave * tmax);
        atomicAdd(&g_tilt_hist[34], 1ull);
      }
    }
  }
}

#undef R
#undef TILT_STORE_NORMAL
#undef TILT_FENCE

// stats: [0] hidden [1] visible (both from the final states) [2] - [3] trial normals [4] batches of 64 point tests (polygon and exact
// searches; the 16-lane passes': packed in [5] = kStatPacked of the copies)
// [6] unresolved [7] exact predicate evaluations [8] length of the list for k_hpr_exact
// The candidates the passes in front left undecided, as a list (any order): the searches then run on wavefronts that all have
// work.  Launched over every candidate, nine wavefronts in ten found theirs decided and left after one load -- and the
// dispatcher could not refill the slots as fast as they emptied: 1.15 resident wavefronts per SIMD of the 3 the registers
// allow, vector ALU busy 28 % (profiles/r03m_hpr_pmc.json).  One atomic per 1024 candidates.
constexpr int kHprListPer = 4;   // candidates per lane
__global__ __launch_bounds__(kHprBlock) void k_hpr_list(const uint8_t *__restrict__ state, int32_t m, int32_t *__restrict__ list,
                                                        unsigned long long *__restrict__ length) {
  __shared__ int32_t ws[kHprBlock / 64];
  __shared__ unsigned long long block_base;
  const int32_t base = (static_cast<int32_t>(blockIdx.x) * kHprBlock + static_cast<int32_t>(threadIdx.x)) * kHprListPer;
  bool und[kHprListPer];
  int32_t c = 0;
#pragma unroll
  for (int k = 0; k < kHprListPer; ++k) {
    und[k] = base + k < m && state[base + k] == kStUndecided;
    c += und[k] ? 1 : 0;
  }
  int32_t total;
  const int32_t ex = scan_block_exclusive(c, &total, ws);
  if (total == 0) return;  // uniform
  if (threadIdx.x == 0) block_base = atomicAdd(length, static_cast<unsigned long long>(total));
  __syncthreads();
  int32_t at = static_cast<int32_t>(block_base) + ex;
#pragma unroll
  for (int k = 0; k < kHprListPer; ++k)
    if (und[k]) list[at++] = base + k;
}

__device__ __forceinline__ void hpr_decide_one(const HprArrays &A, const HprGrid &G, int32_t j, uint8_t *__restrict__ state,
                                               int32_t *__restrict__ undecided, unsigned long long *__restrict__ stats,
                                               int32_t force_exact) {
  Search S;
  S.p = load_point(A, j);
  S.self = j;
  S.self_idx = A.sidx[j];
  S.tests = 0;
  S.fail_code = 0;
  search_frame(S);
  Polygon P;
  unsigned long long restarts = 0;
  int32_t out = kStUndecided;
  if (!force_exact) {
    // the box holds every plane that misses the ray from the origin through p by 1e-9 rad.  The trial normal is the
    // least tilted one the constraints allow, so the size of the box costs nothing.
    const int r = run_search(S, P, A, G, kHprBox, kHprMaxRestarts, restarts);
    int why = 0;
    if (r == kSearchHiddenDup) {
      out = kStHidden;
    } else if (r == kSearchVisible) {
      if (!S.uncertain_left) out = kStVisible; else why = 10;
    } else if (r == kSearchEmpty && S.cert_a >= 0 && S.cert_b >= 0) {
      if (tetra_contains_filtered(S.p, load_point(A, S.cert_a), load_point(A, S.cert_b), load_point(A, S.cert_c)))
        out = kStHidden;
      else why = 11;
    } else if (r == kSearchEmpty) why = 12; else why = 13 + (S.fail_code & 7);
    if (why && lane_id() == 0) atomicAdd(&stats[why], 1ull);  // rare
  }
  if (lane_id() == 0) {
    state[j] = static_cast<uint8_t>(out);
    // the tallies of all wavefronts on one cache line would queue up behind each other in the L2 (measured: 38 ns per
    // candidate whatever its work): kStatCopies copies, each on lines of its own, summed by the host
    unsigned long long *mine = stats + kStatStride * (1 + (blockIdx.x % kStatCopies));
    atomicAdd(&mine[3], restarts);
    atomicAdd(&mine[4], S.tests);
    if (out == kStUndecided) undecided[atomicAdd(&stats[8], 1ull)] = j;
  }
}

// one wavefront per entry of the list, the wavefronts of a capped grid striding over it (its length is read on the device);
// workgroups of ONE wavefront: with four, a workgroup's slots were refilled only as fast as whole workgroups could be placed
// (1.38 resident wavefronts per SIMD of 2, profiles/r03q_hpr_pmc.json)
constexpr int32_t kHprDecideGrid = 65536;
constexpr int32_t kHprTiltGrid = 4096;  // workgroups of k_hpr_tilt (16 rows each) striding over its list
constexpr int32_t kHprOneStepGrid = 1 << 20;  // (k_hpr_radial does not stride: its grid covers the list's upper bound, extra workgroups leave at once)
constexpr int32_t kHprTilt64Grid = 768;  // workgroups of its continuation (4 rows of 64 each): every resident slot at 3 wavefronts per SIMD
constexpr int32_t kTiltBudget = 128;     // round trips of a search on a row of 16 lanes before it is handed on (PCP_TILT_BUDGET; 32: pass +12 %, 48-128 and never: within 1 %)
#ifndef PCP_DECIDE_WPE
#define PCP_DECIDE_WPE 2  // 214 VGPRs, nothing spilled; at 3 wavefronts per SIMD (168 VGPRs) 63 registers went to scratch: hull pass 0.312 -> 0.285 s once the searches ran from a list
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PCP_DECIDE_WPE, PCP_DECIDE_WPE))) void k_hpr_decide(HprArrays A, HprGrid G, uint8_t *__restrict__ state,
                                                          const int32_t *__restrict__ todo, int32_t *__restrict__ undecided,
                                                          unsigned long long *__restrict__ stats, int32_t force_exact) {
  const int32_t count = static_cast<int32_t>(stats[kStatSearch]);
  for (int32_t u = static_cast<int32_t>(blockIdx.x); u < count; u += static_cast<int32_t>(gridDim.x))
    hpr_decide_one(A, G, todo[u], state, undecided, stats, force_exact);
}

// ---- the exact path ----

// Does the trial normal S.n have every other point of S strictly on its inner side, EXACTLY?  Cells are cleared by the
// bound of the file header (rigorous for this n); a point the floating-point evaluation cannot clear is signed in
// expansion arithmetic.  The origin is inside because n . p > 0 (checked the same way).
__device__ inline bool witness_supports(const Search &S, const HprArrays &A, const HprGrid &G, int *n_exact) {
  const Vec3d o = {0.0, 0.0, 0.0};
  if (dot_diff_sign_exact(S.n, o, S.p) >= 0) return false;
  bool ok_all = true;
  const int l = lane_id();
  traverse_all(S, A, G, [&](int32_t k0, int32_t k1) {
    for (int32_t base = k0; base < k1; base += 64) {
      const int32_t k = base + l;
      bool ok = true;
      if (k < k1 && k != S.self) {
        const Vec3d q = load_point(A, k);
        const double dx = q.x - S.p.x, dy = q.y - S.p.y, dz = q.z - S.p.z;
        if (!(dx == 0.0 && dy == 0.0 && dz == 0.0)) {  // a duplicate p stands for (a lower index was found by the search)
          const double tx = S.n.x * dx, ty = S.n.y * dy, tz = S.n.z * dz;
          const double t = (tx + ty) + tz, T = (fabs(tx) + fabs(ty)) + fabs(tz);
          if (!(t < -kPointSlack * T)) {
            ++*n_exact;
            ok = dot_diff_sign_exact(S.n, q, S.p) < 0;
          }
        }
      }
      if (__ballot(!ok)) {
        ok_all = false;
        return false;
      }
    }
    return true;
  });
  return ok_all;
}

// All of S strictly on the origin's side of the plane (p, a, b)?  (Then p, a, b span a facet of the hull and p is a
// vertex.)  Cells are cleared against the floating-point normal n_f = (a - p) x (b - p) with room for its distance from
// the real normal: |n_f - n| <= 6 eps |a - p| |b - p|, so n . (q - p) < 0 follows from n_f . q < n_f . p - slack with
// slack = 1e-15 |a - p| |b - p| reach, reach >= |q - p| for every q.  Points of the cells that remain are signed exactly.
__device__ inline bool plane_supports(const Search &S0, const HprArrays &A, const HprGrid &G, double reach, int32_t ia,
                                      int32_t ib, int *n_exact) {
  const Vec3d o = {0.0, 0.0, 0.0};
  const Vec3d a = load_point(A, ia), b = load_point(A, ib);
  const int so = orient3d_sign(S0.p, a, b, o, n_exact);
  if (so == 0) return false;
  Search S = S0;
  const double ux = a.x - S.p.x, uy = a.y - S.p.y, uz = a.z - S.p.z;
  const double vx = b.x - S.p.x, vy = b.y - S.p.y, vz = b.z - S.p.z;
  Vec3d n = {uy * vz - uz * vy, uz * vx - ux * vz, ux * vy - uy * vx};
  double hp = n.x * S.p.x + n.y * S.p.y + n.z * S.p.z;
  if (hp < 0.0) {
    n = {-n.x, -n.y, -n.z};
    hp = -hp;
  }
  const double nn = sqrt(n.x * n.x + n.y * n.y + n.z * n.z);
  const double lu = sqrt(ux * ux + uy * uy + uz * uz), lv = sqrt(vx * vx + vy * vy + vz * vz);
  const double slack = 1.0e-15 * lu * lv * reach;
  const bool use_cells = nn > 0.0 && hp * (1.0 - 1.0e-13) - slack > 0.0;
  S.nh = {n.x / nn, n.y / nn, n.z / nn};
  S.nn_hi = nn * (1.0 + 1.0e-14);
  S.hp_lo = hp * (1.0 - 1.0e-13) - slack;
  bool ok_all = true;
  const int l = lane_id();
  auto range = [&](int32_t k0, int32_t k1) {
    for (int32_t base = k0; base < k1; base += 64) {
      const int32_t k = base + l;
      bool ok = true;
      if (k < k1 && k != S.self && k != ia && k != ib) {
        const Vec3d q = load_point(A, k);
        const bool same_p = q.x == S.p.x && q.y == S.p.y && q.z == S.p.z;  // a duplicate the duplicate rule lets p stand for
        const bool same_ab = (q.x == a.x && q.y == a.y && q.z == a.z) || (q.x == b.x && q.y == b.y && q.z == b.z);
        if (!same_p && !same_ab) ok = orient3d_sign(S.p, a, b, q, n_exact) == so;
      }
      if (__ballot(!ok)) {
        ok_all = false;
        return false;
      }
    }
    return true;
  };
  if (use_cells)
    traverse_all(S, A, G, range);
  else
    range(0, G.m);
  return ok_all;
}

// p in the closed tetrahedron (origin, a, b, c), the tetrahedron not flat
__device__ inline bool tetra_contains_exact(const Vec3d &p, const Vec3d &a, const Vec3d &b, const Vec3d &c, int *n_exact) {
  const Vec3d o = {0.0, 0.0, 0.0};
  const int s0 = orient3d_sign(a, b, c, o, n_exact);
  if (s0 == 0) return false;
  const int t0 = orient3d_sign(a, b, c, p, n_exact);
  if (t0 != 0 && t0 != s0) return false;
  const int s1 = orient3d_sign(o, b, c, a, n_exact), t1 = orient3d_sign(o, b, c, p, n_exact);
  if (s1 == 0 || (t1 != 0 && t1 != s1)) return false;
  const int s2 = orient3d_sign(o, c, a, b, n_exact), t2 = orient3d_sign(o, c, a, p, n_exact);
  if (s2 == 0 || (t2 != 0 && t2 != s2)) return false;
  const int s3 = orient3d_sign(o, a, b, c, n_exact), t3 = orient3d_sign(o, a, b, p, n_exact);
  if (s3 == 0 || (t3 != 0 && t3 != s3)) return false;
  return true;
}

constexpr int kHprExactIds = 16;

// one undecided candidate (place j of the cell order), one wavefront
__device__ void hpr_exact_one(const HprArrays &A, const HprGrid &G, double reach, int32_t j, uint8_t *__restrict__ state,
                              unsigned long long *__restrict__ stats, int32_t *ids) {
  const int l = lane_id();
  Search S;
  S.p = load_point(A, j);
  S.self = j;
  S.self_idx = A.sidx[j];
  S.tests = 0;
  S.fail_code = 0;
  search_frame(S);
  Polygon P;
  unsigned long long restarts = 0;
  int n_exact = 0;
  int32_t out = -1;
  // the search again, with more patience
  const int r = run_search(S, P, A, G, kHprBox, 4 * kHprMaxRestarts, restarts);
  int n_ids = 0;
  if (r == kSearchHiddenDup) {
    out = kStHidden;
  } else {
    // the constraints that bound what is left of the polygon (and, when nothing is left, the three that emptied it)
    if (r == kSearchEmpty) {
      if (l == 0) {
        ids[0] = S.cert_c;
        ids[1] = S.cert_a;
        ids[2] = S.cert_b;
      }
      n_ids = 3;
    }
    __syncthreads();
    for (int i = 0; i < P.nv && n_ids < kHprExactIds; ++i) {
      const int32_t e = __shfl(P.eid, i, 64);
      bool seen = false;
      for (int t = 0; t < n_ids; ++t) seen = seen || ids[t] == e;
      if (!seen) {
        if (l == 0) ids[n_ids] = e;
        ++n_ids;
      }
      __syncthreads();
    }
    if (r == kSearchVisible) {
      // the last trial normal, signed exactly where floating point could not; failing that, the polygon's vertex mean
      // (a point deep inside what the search left)
      if (witness_supports(S, A, G, &n_exact)) {
        out = kStVisible;
      } else {
        const bool valid = l < P.nv;
        const double mx = wave_sum(valid ? P.vx : 0.0) / P.nv, my = wave_sum(valid ? P.vy : 0.0) / P.nv;
        if (fabs(mx) < 64.0 && fabs(my) < 64.0) {
          S.n = {S.e0.x + (mx * S.e1.x + my * S.e2.x), S.e0.y + (mx * S.e1.y + my * S.e2.y),
                 S.e0.z + (mx * S.e1.z + my * S.e2.z)};
          const double nn = sqrt(S.n.x * S.n.x + S.n.y * S.n.y + S.n.z * S.n.z);
          S.nh = {S.n.x / nn, S.n.y / nn, S.n.z / nn};
          S.nn_hi = nn * (1.0 + 1.0e-14);
          S.hp_lo = (S.n.x * S.p.x + S.n.y * S.p.y + S.n.z * S.p.z) * (1.0 - 1.0e-13);
          if (S.hp_lo > 0.0 && witness_supports(S, A, G, &n_exact)) out = kStVisible;
        }
      }
    }
    if (out < 0 && (r == kSearchVisible || r == kSearchFail)) {
      // a facet of the hull at p: the planes through p and two constraints that meet in a vertex of the polygon
      for (int i = 0; i < P.nv && out < 0; ++i) {
        const int32_t ea = __shfl(P.eid, (i + P.nv - 1) % P.nv, 64), eb = __shfl(P.eid, i, 64);
        if (ea >= 0 && eb >= 0 && ea != eb && plane_supports(S, A, G, reach, ea, eb, &n_exact)) out = kStVisible;
      }
    }
    if (out < 0) {
      // a simplex (origin, a, b, c) around p, among the binding constraints: triples dealt out over the lanes
      int found = 0;
      int t = 0;
      for (int ia = 0; ia < n_ids; ++ia)
        for (int ib = ia + 1; ib < n_ids; ++ib)
          for (int ic = ib + 1; ic < n_ids; ++ic, ++t) {
            if ((t & 63) != l || found) continue;
            const int32_t a = ids[ia], b = ids[ib], c = ids[ic];
            if (a < 0 || b < 0 || c < 0) continue;
            if (tetra_contains_exact(S.p, load_point(A, a), load_point(A, b), load_point(A, c), &n_exact)) found = 1;
          }
      if (__ballot(found)) out = kStHidden;
    }
    if (out < 0 && r == kSearchEmpty) {
      // the polygon emptied in floating point but no simplex holds p: any pair of binding constraints as a facet
      for (int ia = 0; ia < n_ids && out < 0; ++ia)
        for (int ib = ia + 1; ib < n_ids && out < 0; ++ib)
          if (ids[ia] >= 0 && ids[ib] >= 0 && plane_supports(S, A, G, reach, ids[ia], ids[ib], &n_exact)) out = kStVisible;
    }
  }
  const unsigned long long exact_total = static_cast<unsigned long long>(wave_sum(static_cast<double>(n_exact)));
  if (l == 0) {
    if (out < 0) {
      atomicAdd(&stats[6], 1ull);  // unresolved: classified hidden (file header)
      out = kStHidden;
    }
    state[j] = static_cast<uint8_t>(out);
    atomicAdd(&stats[3], restarts);
    atomicAdd(&stats[4], S.tests);
    atomicAdd(&stats[7], exact_total);
  }
}

// (a fixed grid whose wavefronts stride over the list; its length is read where k_hpr_decide counted it -- stats[8] -- so the
// host neither waits for the count nor adds launches, however long the list)
constexpr int32_t kHprExactGrid = 64;  // (a handful of candidates per keyframe; every wavefront of the launch reserves 2.8 KB of scratch per lane)
__global__ __launch_bounds__(64) void k_hpr_exact(HprArrays A, HprGrid G, double reach, const int32_t *__restrict__ undecided,
                                                  uint8_t *__restrict__ state, unsigned long long *__restrict__ stats) {
  __shared__ int32_t ids[kHprExactIds];
  const unsigned long long count = stats[8];
  for (unsigned long long u = blockIdx.x; u < count; u += gridDim.x) {
    hpr_exact_one(A, G, reach, undecided[u], state, stats, ids);
    __syncthreads();
  }
}

// whole-run bits (pcp_colour.hip, hull_bits): bit of the keyframe at the sorted place of every hull vertex (plane cleared)
// (tally: the counters of the keyframe, or null -- the kernel that reads the final states first counts them: visible / hidden)
__device__ __forceinline__ void tally_states(bool in, uint8_t st, unsigned long long *__restrict__ tally) {
  if (!tally) return;
  const unsigned long long v = __ballot(in && st == kStVisible), h = __ballot(in && st == kStHidden);
  if (lane_id() == 0) {
    unsigned long long *mine = tally + kStatStride * (1 + (blockIdx.x % kStatCopies));
    if (v) atomicAdd(&mine[kStVisible], static_cast<unsigned long long>(__popcll(v)));
    if (h) atomicAdd(&mine[kStHidden], static_cast<unsigned long long>(__popcll(h)));
  }
}

__global__ __launch_bounds__(kHprBlock) void k_hpr_set_bits(const uint8_t *__restrict__ state, const int32_t *__restrict__ splace,
                                                            int32_t m, uint32_t *__restrict__ word, uint32_t bit,
                                                            unsigned long long *__restrict__ tally) {
  const int32_t k = static_cast<int32_t>(blockIdx.x) * kHprBlock + static_cast<int32_t>(threadIdx.x);
  const uint8_t st = k < m ? state[k] : uint8_t(kStUndecided);
  if (k < m && st == kStVisible) atomicOr(word + splace[k], bit);
  tally_states(k < m, st, tally);
}

// keep flags (input order): the candidate flags become the visible flags (keep == null: the count alone)
__global__ __launch_bounds__(kHprBlock) void k_hpr_writeback(const uint8_t *__restrict__ state, const int32_t *__restrict__ sidx,
                                                             int32_t m, uint8_t *__restrict__ keep,
                                                             unsigned long long *__restrict__ tally) {
  const int32_t k = static_cast<int32_t>(blockIdx.x) * kHprBlock + static_cast<int32_t>(threadIdx.x);
  const uint8_t st = k < m ? state[k] : uint8_t(kStUndecided);
  if (k < m && keep) keep[sidx[k]] = st == kStVisible ? 1 : 0;
  tally_states(k < m, st, tally);
}

// zeroes `words` 8-byte words (the per-keyframe clears of the hull: a kernel of the stream it belongs to -- hipMemsetAsync goes
// through the runtime's own fill path)
__global__ __launch_bounds__(kHprBlock) void k_hpr_zero(unsigned long long *__restrict__ p, int64_t words) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kHprBlock + threadIdx.x; i < words; i += static_cast<int64_t>(gridDim.x) * kHprBlock) p[i] = 0ull;
}

// ------------------------------------------------------------------------------------------------------------------
// host
// ------------------------------------------------------------------------------------------------------------------
static inline uint32_t hpr_blocks(int64_t n) { return static_cast<uint32_t>(std::max<int64_t>(1, div_up(n, kHprBlock))); }

static int hpr_scan(pcp_context *ctx, HprLane &L, int32_t *counts, int64_t entries) {
  const int64_t tiles = div_up(entries, kScanTile);
  PCP_HIP_TRY(ctx, L.tiles.ensure(static_cast<size_t>(tiles) + 4));
  hipLaunchKernelGGL(k_scan_tile_sums, dim3(static_cast<uint32_t>(tiles)), dim3(kScanBlock), 0, L.stream, counts, entries,
                     L.tiles.p);
  hipLaunchKernelGGL(k_scan_tile_offsets, dim3(1), dim3(kScanSingle), 0, L.stream, L.tiles.p, tiles,
                     static_cast<unsigned long long *>(nullptr));
  hipLaunchKernelGGL(k_scan_apply, dim3(static_cast<uint32_t>(tiles)), dim3(kScanBlock), 0, L.stream, counts, entries,
                     L.tiles.p, counts);
  PCP_HIP_TRY(ctx, hipGetLastError());
  return PCP_OK;
}

constexpr size_t kStatWords = kStatStride * (1 + kStatCopies);
static_assert(kStatWords * sizeof(unsigned long long) <= pcp_context::kReadbackBytes, "readback scratch");

// hidden_points_removal of one keyframe, first half: the candidates (the filter of view_culling.cpp:276-288) and their
// flipped points from the sorted copy of the cloud, queued on `stream` with the lane's buffers; the count and the bounds
// start their way to the lane's pinned readback, an event behind them.  Nothing waits here.
// Two outputs of the keyframe, each optional: d_flags (n bytes on the device, input order: 1 = hull vertex), and the
// keyframe's bit in a CLEARED plane of the whole-run bits (hull_plane[place in the sorted order] |= bit).
int hpr_begin(pcp_context *ctx, HprLane &L, hipStream_t stream, bool timed, int32_t frame, uint8_t *d_flags,
              uint32_t *hull_plane, uint32_t bit, const uint32_t *tile_mask = nullptr) {
  const int64_t n = ctx->n;
  L.busy = false;
  if (n == 0) return PCP_OK;
  if (n >= (int64_t(1) << 31)) return set_error(ctx, PCP_ERR_RANGE, "hidden_points_removal: %lld points (fewer than 2^31 are handled)", (long long)n);
  const size_t cap = static_cast<size_t>(n);  // every point may be a candidate
  const size_t plane = (static_cast<size_t>(n) + 3) & ~size_t(3);
  L.stream = stream;
  L.frame = frame;
  L.d_flags = d_flags;
  L.hull_plane = hull_plane;
  L.bit = bit;
  if (!L.readback) {
    // fine-grained (coherent) pinned memory, asked for explicitly: the host polls a word k_hpr_publish writes while the stream is
    // still running; with non-coherent pinned memory (HIP_HOST_COHERENT=0 or a runtime default) the kernel's
    // __threadfence_system() writes are only guaranteed visible when the kernel ends
    if (hipHostMalloc(&L.readback, pcp_context::kReadbackBytes, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) {
      L.readback = nullptr;
      return set_error(ctx, PCP_ERR_NOMEM, "hidden_points_removal: no pinned readback for a lane");
    }
    std::memset(L.readback, 0, pcp_context::kReadbackBytes);
    L.seq = 0;
  }
  // doubles: px py pz ga gb rho | sx sy sz, `cap` apart; ints: candidate's input index | its place in the sorted order
  PCP_HIP_TRY(ctx, L.f64.ensure(9 * cap + 16));
  PCP_HIP_TRY(ctx, L.index.ensure(2 * cap + 16));
  PCP_HIP_TRY(ctx, L.stats.ensure(kStatWords));
  double *px = L.f64.p;
  int32_t *cidx = L.index.p, *cplace = cidx + cap;
  unsigned long long *stats = L.stats.p;
  hipLaunchKernelGGL(k_hpr_zero, dim3(static_cast<uint32_t>(div_up(kStatWords, kHprBlock))), dim3(kHprBlock), 0, stream, L.stats.p,
                     static_cast<int64_t>(kStatWords));
  const DevFrame &fr = ctx->hframes[static_cast<size_t>(frame)];
  {
    LaunchTimer t(timed ? ctx : nullptr, PCP_K_HPR);
    hipLaunchKernelGGL(k_hpr_candidates, dim3(hpr_blocks(n)), dim3(kHprBlock), 0, stream, ctx->sxyz.p, ctx->sxyz.p + plane,
                       ctx->sxyz.p + 2 * plane, n, ctx->dcam, fr, ctx->perm.p, ctx->cull.hpr_flip_radius, static_cast<int64_t>(cap),
                       cidx, cplace, px, stats, tile_mask, ctx->mask_words, frame);
    PCP_HIP_TRY(ctx, hipGetLastError());
  }
  // the number of candidates (block 0) and the bounds (folded from the copies in words 24..27 of the other blocks) into the
  // lane's pinned readback, the keyframe's sequence number behind them
  L.seq += 1;
  hipLaunchKernelGGL(k_hpr_publish, dim3(1), dim3(64), 0, stream, stats, L.seq, static_cast<volatile HprCounts *>(L.readback));
  PCP_HIP_TRY(ctx, hipGetLastError());
  L.busy = true;
  return PCP_OK;
}

// second half: waits for the lane's count and bounds (the one host wait of a keyframe), sizes the gnomonic grid and queues
// binning, the certificate passes, the searches, the exact path and the outputs behind the first half.
int hpr_finish(pcp_context *ctx, HprLane &L, bool timed) {
  if (!L.busy) return PCP_OK;
  L.busy = false;
  const int64_t n = ctx->n;
  const size_t cap = static_cast<size_t>(n);
  int rc = PCP_OK;
  hipStream_t stream = L.stream;
  pcp_context *tctx = timed ? ctx : nullptr;
  const int32_t frame = L.frame;
  (void)frame;
  uint8_t *d_flags = L.d_flags;
  uint32_t *hull_plane = L.hull_plane;
  const uint32_t bit = L.bit;
  double *px = L.f64.p, *py = px + cap, *pz = py + cap, *ga = pz + cap, *gb = ga + cap, *rho = gb + cap;
  double *sx = rho + cap, *sy = sx + cap, *sz = sy + cap;
  int32_t *cidx = L.index.p, *cplace = cidx + cap;
  unsigned long long *stats = L.stats.p;
  // The one host wait of a keyframe: polling the sequence number k_hpr_publish writes last (an error on the stream ends it)
  const volatile HprCounts *hc = static_cast<const volatile HprCounts *>(L.readback);
  {
    const auto t_wait = std::chrono::steady_clock::now();
    for (uint64_t spins = 0; hc->seq != L.seq; ++spins) {
      __builtin_ia32_pause();
      if ((spins & 0xfffffu) == 0xfffffu) {  // now and then: is the stream still alive?
        const hipError_t q = hipStreamQuery(stream);
        if (q != hipSuccess && q != hipErrorNotReady) PCP_HIP_TRY(ctx, q);
        if (q == hipSuccess && hc->seq != L.seq) {
          __sync_synchronize();
          if (hc->seq != L.seq) return set_error(ctx, PCP_ERR_DEVICE, "hidden_points_removal: the candidates' count never arrived");
        }
      }
    }
    __sync_synchronize();
    ctx->hpr_host_wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_wait).count();
  }
  const int64_t m64 = static_cast<int64_t>(hc->count);
  unsigned long long hb[4] = {~hc->inv_amin, hc->amax, ~hc->inv_bmin, hc->bmax};  // the minima were kept as maxima of the inverted keys
  std::memset(ctx->hpr_stats, 0, sizeof(ctx->hpr_stats));
  ctx->hpr_stats_pending = false;
  ctx->hpr_stats[9] = m64;
  ctx->hpr_last_lane = static_cast<int32_t>(&L - ctx->hpr_lane);
  if (d_flags) PCP_HIP_TRY(ctx, hipMemsetAsync(d_flags, 0, static_cast<size_t>(n), stream));  // the hull vertices are set below
  if (m64 < 3) {
    // qhull needs dim + 1 points (here: three candidates and the origin); with fewer it fails and the reference
    // returns no visible point (view_culling.cpp:307-312)
    return PCP_OK;
  }
  const int32_t m = static_cast<int32_t>(m64);
  const size_t sm = static_cast<size_t>(m);
  // ints: cell | sidx splace scell | undecided | todo (the list of the searches)
  PCP_HIP_TRY(ctx, L.i32.ensure(6 * sm + 16));
  int32_t *cell = L.i32.p, *sidx = cell + sm, *splace = sidx + sm, *scell = splace + sm, *undecided = scell + sm;
  int32_t *todo = undecided + sm;
  PCP_HIP_TRY(ctx, L.state.ensure(sm + 16));
  const double amin = key_to_double(hb[0]), amax = key_to_double(hb[1]), bmin = key_to_double(hb[2]), bmax = key_to_double(hb[3]);
  if (!(std::isfinite(amin) && std::isfinite(amax) && std::isfinite(bmin) && std::isfinite(bmax)))
    return set_error(ctx, PCP_ERR_INVALID, "hidden_points_removal: non-finite flipped coordinates (flip radius %g)",
                     ctx->cull.hpr_flip_radius);
  HprGrid G{};
  G.m = m;
  G.a0 = amin;
  G.b0 = bmin;
  {
    const double wa = amax - amin, wb = bmax - bmin;
    double h = std::sqrt(kHprTargetPerCell * std::max(wa, 1e-12) * std::max(wb, 1e-12) / static_cast<double>(m));
    h = std::max(h, std::max(wa, wb) * 1e-6);
    h = std::max(h, 1e-9);
    for (;;) {
      G.gw = static_cast<int32_t>(std::min(wa / h, 1e9)) + 1;
      G.gh = static_cast<int32_t>(std::min(wb / h, 1e9)) + 1;
      if (static_cast<int64_t>(G.gw) * G.gh <= kHprMaxCells) break;
      h *= 1.5;
    }
    G.h = h;
    G.inv_h = 1.0 / h;
    G.cgw = (G.gw + kHprCoarse - 1) / kHprCoarse;
    G.cgh = (G.gh + kHprCoarse - 1) / kHprCoarse;
    G.mgw = (G.gw + kHprMid - 1) / kHprMid;
    G.mgh = (G.gh + kHprMid - 1) / kHprMid;
    // the gnomonic plane z = 1 projects onto the unit sphere without stretching any distance, so half a cell diagonal
    // bounds the chord from the centre direction; the slack covers a coordinate that rounding put on a cell's edge
    G.r_fine = 0.5 * h * std::sqrt(2.0) * (1.0 + 1e-9) + 1e-12;
    G.r_coarse = 0.5 * h * kHprCoarse * std::sqrt(2.0) * (1.0 + 1e-9) + 1e-12;
    G.r_mid = 0.5 * h * kHprMid * std::sqrt(2.0) * (1.0 + 1e-9) + 1e-12;
    const double ax = std::max(std::fabs(amin), std::fabs(amin + G.gw * h)), ay = std::max(std::fabs(bmin), std::fabs(bmin + G.gh * h));
    G.a_reach = std::sqrt(1.0 + ax * ax + ay * ay) * (1.0 + 1e-9);
    G.rho_max = reinterpret_cast<const double *>(L.stats.p + 28);
  }
  const int64_t n_fine = static_cast<int64_t>(G.gw) * G.gh, n_coarse = static_cast<int64_t>(G.cgw) * G.cgh;
  // per cell, in ONE allocation whose zeroed part comes first (one memset instead of three: each small launch costs its
  // 3 us and a 5-10 us gap, 256 times per run): rho bits (n_fine) | representative (n_fine) | count / start (n_fine + 2)
  // and cursor (n_fine) as int32 | centre directions (3 n_fine), coarse rho / directions (4 n_coarse)
  const size_t nf = static_cast<size_t>(n_fine), nc = static_cast<size_t>(n_coarse);
  const size_t int_words = (2 * nf + 16 + 1) / 2;  // the int32 part, in 8-byte words
  const size_t nm = static_cast<size_t>(G.mgw) * static_cast<size_t>(G.mgh);
  PCP_HIP_TRY(ctx, L.cells_d.ensure(2 * nf + int_words + 3 * nf + 4 * nc + 2 * nf + 2 * nc + 2 * nm + 18));
  double *crho = L.cells_d.p;
  unsigned long long *crep_all = reinterpret_cast<unsigned long long *>(crho + nf);
  int32_t *cstart = reinterpret_cast<int32_t *>(crho + 2 * nf), *cursor = cstart + n_fine + 2;
  double *cdir = crho + 2 * nf + int_words, *Crho = cdir + 3 * nf, *Cdir = Crho + nc;
  size_t off4 = 5 * nf + int_words + 4 * nc;
  off4 += off4 & 1;  // 16-byte aligned
  float4 *cell4 = reinterpret_cast<float4 *>(crho + off4), *Cell4 = cell4 + nf, *Mid4 = Cell4 + nc;
  hipLaunchKernelGGL(k_hpr_zero, dim3(static_cast<uint32_t>(std::min<int64_t>(div_up(static_cast<int64_t>(2 * nf + int_words), kHprBlock), 1024))),
                     dim3(kHprBlock), 0, stream, reinterpret_cast<unsigned long long *>(crho), static_cast<int64_t>(2 * nf + int_words));
  // PCP_HPR_QUICK=0 / PCP_HPR_RADIAL=0: without the two passes in front of the search (results identical; the place of a
  // representative needs 26 bits)
  const char *qe = std::getenv("PCP_HPR_QUICK");
  const bool quick = !(qe && qe[0] == '0') && m < (1 << 26);
  unsigned long long *crep = quick ? crep_all : nullptr;
  HprArrays A{sx, sy, sz, sidx, scell, cstart, crho, cdir, Crho, Cdir, crep, cell4, Cell4, Mid4};
  // PCP_HPR_FORCE_EXACT=1 (tests): every candidate takes the exact path; read per call
  const char *fe = std::getenv("PCP_HPR_FORCE_EXACT");
  const bool force_exact = fe && fe[0] == '1';
  {
    LaunchTimer t(tctx, PCP_K_HPR);
    hipLaunchKernelGGL(k_hpr_count, dim3(hpr_blocks(m)), dim3(kHprBlock), 0, stream, ga, gb, G, cell, cstart);
    if ((rc = hpr_scan(ctx, L, cstart, n_fine + 1)) != PCP_OK) return rc;
    hipLaunchKernelGGL(k_hpr_scatter, dim3(hpr_blocks(m)), dim3(kHprBlock), 0, stream, px, py, pz, rho, cidx, cplace,
                       cell, m, cstart, cursor, sx, sy, sz, sidx, splace, scell,
                       reinterpret_cast<unsigned long long *>(crho), crep);
    hipLaunchKernelGGL(k_hpr_cells, dim3(hpr_blocks(std::max(n_fine, n_coarse))), dim3(kHprBlock), 0, stream, G,
                       reinterpret_cast<const unsigned long long *>(crho), cdir, Crho, Cdir, stats + 28, cell4, Cell4, Mid4);
    // PCP_HPR_RADIAL=0: every candidate through k_hpr_decide (results identical)
    const char *re = std::getenv("PCP_HPR_RADIAL");
    if (quick && !force_exact)
      hipLaunchKernelGGL(k_hpr_quick, dim3(hpr_blocks(m)), dim3(kHprBlock), 0, stream, A, G, L.state.p, stats);
    else
      PCP_HIP_TRY(ctx, hipMemsetAsync(L.state.p, kStUndecided, sm, stream));
    if (!(force_exact || (re && re[0] == '0'))) {
      const int32_t *radial_todo = nullptr;
      if (quick) {  // rows for the candidates the quick certificate left, and only for them
        hipLaunchKernelGGL(k_hpr_list, dim3(static_cast<uint32_t>(div_up(m, kHprBlock * kHprListPer))), dim3(kHprBlock), 0, stream,
                           L.state.p, m, undecided, stats + kStatRadial);  // (`undecided` is free until the searches)
        radial_todo = undecided;
      }
      const bool tally = std::getenv("PCP_HPR_DEBUG") != nullptr;  // (the 16-lane passes count their work only then)
      hipLaunchKernelGGL((tally ? k_hpr_radial<false, true> : k_hpr_radial<false, false>), dim3(static_cast<uint32_t>(div_up(m, kHprBlock / 16))),
                         dim3(kHprBlock), 0, stream, A, G, L.state.p, radial_todo, stats, static_cast<int32_t>(kStatRadial));
      // PCP_HPR_ONESTEP=0: without the once-tilted plane for what the radial plane failed (results identical)
      const char *oe = std::getenv("PCP_HPR_ONESTEP");
      if (!(oe && oe[0] == '0')) {
        hipLaunchKernelGGL(k_hpr_list, dim3(static_cast<uint32_t>(div_up(m, kHprBlock * kHprListPer))), dim3(kHprBlock), 0, stream,
                           L.state.p, m, todo, stats + kStatOneStep);
        hipLaunchKernelGGL((tally ? k_hpr_radial<true, true> : k_hpr_radial<true, false>),
                           dim3(static_cast<uint32_t>(std::min<int64_t>(div_up(m, kHprBlock / 16), kHprOneStepGrid))),
                           dim3(kHprBlock), 0, stream, A, G, L.state.p, todo, stats, static_cast<int32_t>(kStatOneStep));
      }
    }
    std::vector<uint8_t> dbg_before;
    if (std::getenv("PCP_HPR_DEBUG")) {  // what the two passes in front left to the searches
      dbg_before.resize(sm);
      (void)hipMemcpyAsync(dbg_before.data(), L.state.p, sm, hipMemcpyDeviceToHost, stream);
      (void)hipStreamSynchronize(stream);
    }
    // PCP_HPR_TILT=0: without the least-norm searches in front of the polygon search (results identical)
    const char *te = std::getenv("PCP_HPR_TILT");
    if (!force_exact && !(te && te[0] == '0')) {
      const bool dbg = std::getenv("PCP_HPR_DEBUG") != nullptr;
      // PCP_TILT_BUDGET: round trips a search gets on a row of 16 lanes before it is handed on to a wavefront of its own
      // (0: never handed on -- the form of round 4)
      const char *be = std::getenv("PCP_TILT_BUDGET");
      int32_t budget = be ? std::atoi(be) : kTiltBudget;
      if (budget <= 0) budget = INT32_MAX;
      const int32_t wide_window = std::getenv("PCP_TILT_WIDE") ? std::atoi(std::getenv("PCP_TILT_WIDE")) : kTiltWideWindow;
      PCP_HIP_TRY(ctx, L.cont.ensure((sizeof(TiltCont) / sizeof(double)) * sm + 16));
      TiltCont *cont = reinterpret_cast<TiltCont *>(L.cont.p);
      auto debug_hist = [&](const char *what, int rows_per_wave) {
        if (!dbg) return;
        (void)hipStreamSynchronize(stream);
        unsigned long long hist[40] = {0}, zero[40] = {0};
        (void)hipMemcpyFromSymbol(hist, HIP_SYMBOL(g_tilt_hist), sizeof(hist));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tilt_hist), zero, sizeof(zero));
        fprintf(stderr, "hpr: tilt (%s): searches by log2(round trips):", what);
        for (int b = 0; b < 16; ++b) fprintf(stderr, " %llu", hist[b]);
        fprintf(stderr, "\nhpr: tilt (%s): round trips of each class:", what);
        for (int b = 0; b < 16; ++b) fprintf(stderr, " %llu", hist[16 + b]);
        fprintf(stderr, "\nhpr: tilt (%s): %llu wavefronts, rows' round trips (point batches + cell rows) %llu of %llu row slots (%d x the longest row); cell rows %llu\n",
                what, hist[34], hist[32], hist[33], rows_per_wave, hist[35]);
      };
      hipLaunchKernelGGL(k_hpr_list, dim3(static_cast<uint32_t>(div_up(m, kHprBlock * kHprListPer))), dim3(kHprBlock), 0, stream,
                         L.state.p, m, todo, stats + kStatTilt);
      // (`cell`, the candidates' cells in arrival order, is free after k_hpr_scatter: the list of what the searches give up on)
      hipLaunchKernelGGL((dbg ? k_hpr_tilt<true, 16> : k_hpr_tilt<false, 16>),
                         dim3(static_cast<uint32_t>(std::min<int64_t>(div_up(m, kTiltBlock / 16), kHprTiltGrid * (kHprBlock / kTiltBlock)))),
                         dim3(kTiltBlock), 0, stream, A, G, L.state.p, todo, stats, wide_window, cell, cont, budget);
      debug_hist("rows of 16", 4);
      if (budget != INT32_MAX) {
        hipLaunchKernelGGL((dbg ? k_hpr_tilt<true, 64> : k_hpr_tilt<false, 64>),
                           dim3(static_cast<uint32_t>(std::min<int64_t>(div_up(m, kTiltBlock / 64), kHprTilt64Grid * (kHprBlock / kTiltBlock)))),
                           dim3(kTiltBlock), 0, stream, A, G, L.state.p, static_cast<const int32_t *>(nullptr), stats, wide_window, cell,
                           cont, INT32_MAX);
        debug_hist("rows of 64", 1);
      }
    }
    // the list of the 64-lane search: what k_hpr_tilt gave up on (it appended them itself: no third listing launch), or, without
    // that pass, every candidate still undecided
    const bool tilted = !force_exact && !(te && te[0] == '0');
    const int32_t *decide_todo = cell;
    if (!tilted) {
      hipLaunchKernelGGL(k_hpr_list, dim3(static_cast<uint32_t>(div_up(m, kHprBlock * kHprListPer))), dim3(kHprBlock), 0, stream,
                         L.state.p, m, todo, stats + kStatSearch);
      decide_todo = todo;
    }
    // (after k_hpr_tilt a few dozen candidates are left: a grid of 64 K one-wavefront workgroups that find nothing costs 15 us)
    hipLaunchKernelGGL(k_hpr_decide, dim3(static_cast<uint32_t>(std::min<int64_t>(m, tilted ? 2048 : kHprDecideGrid))), dim3(64),
                       0, stream, A, G, L.state.p, decide_todo, undecided, stats, force_exact ? 1 : 0);
    if (!dbg_before.empty()) {
      std::vector<uint8_t> after(sm);
      (void)hipMemcpyAsync(after.data(), L.state.p, sm, hipMemcpyDeviceToHost, stream);
      (void)hipStreamSynchronize(stream);
      size_t und = 0, to_vis = 0, to_hid = 0, left = 0;
      for (size_t k = 0; k < sm; ++k)
        if (dbg_before[k] == kStUndecided) {
          ++und;
          if (after[k] == kStVisible) ++to_vis;
          else if (after[k] == kStHidden) ++to_hid;
          else ++left;
        }
      fprintf(stderr, "hpr: %d candidates, %zu searched: %zu visible, %zu hidden, %zu to the exact path\n", m, und, to_vis, to_hid, left);
    }
    PCP_HIP_TRY(ctx, hipGetLastError());
  }
  {
    // the exact path for what the searches left undecided (a handful per keyframe): a fixed grid that reads the count on
    // the device and strides over the list; the tallies stay on the device until pcp_hpr_stats asks for them -- no host
    // round trip in here
    LaunchTimer t(tctx, PCP_K_HPR);
    hipLaunchKernelGGL(k_hpr_exact, dim3(static_cast<uint32_t>(std::min<int32_t>(m, kHprExactGrid))), dim3(64), 0, stream, A, G,
                       4.0 * std::fabs(ctx->cull.hpr_flip_radius) + 1.0e4, undecided, L.state.p, stats);
    PCP_HIP_TRY(ctx, hipGetLastError());
  }
  {
    LaunchTimer t(tctx, PCP_K_HPR);
    if (d_flags || !hull_plane)
      hipLaunchKernelGGL(k_hpr_writeback, dim3(hpr_blocks(m)), dim3(kHprBlock), 0, stream, L.state.p, sidx, m, d_flags, stats);
    if (hull_plane)
      hipLaunchKernelGGL(k_hpr_set_bits, dim3(hpr_blocks(m)), dim3(kHprBlock), 0, stream, L.state.p, splace, m, hull_plane, bit,
                         d_flags ? static_cast<unsigned long long *>(nullptr) : stats);
    PCP_HIP_TRY(ctx, hipGetLastError());
  }
  ctx->hpr_stats[8] = n_fine;
  ctx->hpr_stats_pending = true;  // [0..7] are summed from the device tallies when pcp_hpr_stats is called
  if (std::getenv("PCP_HPR_DEBUG")) {
    unsigned long long dbg[24];
    (void)hipMemcpy(dbg, stats, sizeof(dbg), hipMemcpyDeviceToHost);
    // why candidates left the floating-point path: a point within round-off of the last trial plane / a simplex the
    // filter could not sign / an emptied polygon that still had an edge of the initial box / searches that gave up
    // (inner-loop guard, restart cap, non-contiguous clip, more than 64 polygon vertices)
    {
      std::vector<unsigned long long> all(kStatWords);
      (void)hipMemcpy(all.data(), stats, kStatWords * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      unsigned long long w[14] = {0}, bmax = 0, big = 0, wide = 0, bigsum = 0;
      for (int c = 1; c <= kStatCopies; ++c) {
        for (int k = 0; k < 14; ++k) w[k] += all[static_cast<size_t>(c * kStatStride + 9 + k)];
        bmax = std::max(bmax, all[static_cast<size_t>(c * kStatStride + 28)]);
        big += all[static_cast<size_t>(c * kStatStride + 29)];
        wide += all[static_cast<size_t>(c * kStatStride + 30)];
        bigsum += all[static_cast<size_t>(c * kStatStride + 31)];
      }
      (void)bigsum;
      fprintf(stderr, "hpr: tilt: longest search %llu batches; %llu searches handed on to a wavefront each; %llu passes with a window of > 256 upper cells\n",
              bmax, big, wide);
      fprintf(stderr, "hpr: tilt: %llu searches: visible %llu duplicate %llu empty %llu (certified %llu) gave up %llu (same point %llu, step cap %llu, "
              "no conclusion %llu, pass cap %llu); passes %llu steps %llu batches %llu\n", w[0], w[2], w[3], w[4], w[6], w[5], w[7], w[8], w[9],
              w[10], w[11], w[12], w[13]);
    }
    fprintf(stderr, "hpr: to the exact path: uncertain_left %llu tetra_filter %llu box_cert %llu fail: guard %llu restarts %llu noncontig %llu overflow %llu\n", dbg[10], dbg[11], dbg[12], dbg[14], dbg[15], dbg[16], dbg[17]);
  }
  return PCP_OK;
}

// one keyframe, both halves on the context's stream with lane 0 (the single-keyframe calls: pcp_cull_frame, pcp_frame_visible, NID)
int hpr_run(pcp_context *ctx, int32_t frame, uint8_t *d_flags, uint32_t *hull_plane, uint32_t bit) {
  HprLane &L = ctx->hpr_lane[0];
  int rc = hpr_begin(ctx, L, ctx->stream, true, frame, d_flags, hull_plane, bit);
  if (rc != PCP_OK) return rc;
  return hpr_finish(ctx, L, true);
}

// The hulls of keyframes [f0, f1) into the whole-run bits, `lanes` keyframes in flight, each lane on a stream of its own; the
// host walks the keyframes in order and gives the next one to a lane that is free (see below), finishing the keyframe that
// lane holds before it begins the next one there.  The lanes start behind everything queued on the context's stream so far
// (the cleared planes) and the context's stream continues behind the last kernel of every lane.
// tile_mask: the batched passes' tile x keyframe masks of these keyframes as the tile level left them (not yet refined by the
// depth pass, whose rule also asks for a colour pixel), or null.
int hpr_run_range(pcp_context *ctx, int32_t f0, int32_t f1, int32_t lanes, const uint32_t *tile_mask) {
  if (f1 <= f0 || ctx->n == 0) return PCP_OK;
  lanes = std::max(1, std::min<int32_t>(std::min<int32_t>(lanes, pcp_context::kHprMaxLanes), f1 - f0));
  int rc = PCP_OK;
  if (!ctx->hpr_fork) PCP_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->hpr_fork, hipEventDisableTiming));
  LaunchTimer t(ctx, PCP_K_HPR);  // the whole pass as one bracket on the context's stream
  ctx->hpr_host_wait_s = 0.0;
  const auto t_pass = std::chrono::steady_clock::now();
  PCP_HIP_TRY(ctx, hipEventRecord(ctx->hpr_fork, ctx->stream));  // (nothing is queued on a lane yet: an early return is safe)
  // every lane holds the scratch of a keyframe whose every point may be a candidate (80 B per map point): where the device
  // has no room for all of them, fewer keyframes are in flight
  {
    const size_t cap = static_cast<size_t>(ctx->n);
    for (int32_t k = 1; k < lanes; ++k) {
      HprLane &L = ctx->hpr_lane[k];
      if (L.f64.ensure(9 * cap + 16) != hipSuccess || L.index.ensure(2 * cap + 16) != hipSuccess) {
        (void)hipGetLastError();
        L.f64.release();
        L.index.release();
        lanes = k;
        break;
      }
    }
  }
  // From here on an error does not return: it is kept in `rc` and the function falls through to the drain-and-join block --
  // nothing may stay queued on a lane whose buffers the next call reuses on another stream, no lane may stay `busy`.
  auto keep = [&](hipError_t e, const char *what) {
    if (e != hipSuccess && rc == PCP_OK) rc = set_error(ctx, PCP_ERR_DEVICE, "hidden_points_removal: %s: %s", what, hipGetErrorString(e));
  };
  int32_t forked = 0;  // lanes whose stream waits behind the fork (the others are left alone)
  for (int32_t k = 0; k < lanes && rc == PCP_OK; ++k) {
    HprLane &L = ctx->hpr_lane[k];
    if (!L.own_stream) keep(hipStreamCreateWithFlags(&L.own_stream, hipStreamNonBlocking), "lane stream");
    if (rc == PCP_OK && !ctx->hpr_join[k]) keep(hipEventCreateWithFlags(&ctx->hpr_join[k], hipEventDisableTiming), "join event");
    if (rc == PCP_OK) keep(hipStreamWaitEvent(L.own_stream, ctx->hpr_fork, 0), "fork");
    if (rc == PCP_OK) forked = k + 1;
  }
  // A lane is free for the next keyframe when it holds none, or when the count of the one it holds has arrived -- its stream
  // is in order, so everything queued there before has run.  The next keyframe goes to the first free lane, looked for from the
  // one after the lane served last: a keyframe whose searches take milliseconds holds up its own lane only (handing the
  // keyframes out in turn made the host wait for that lane's next count while the other lanes ran dry).
  auto arrived = [](const HprLane &L) {
    return !L.busy || static_cast<const volatile HprCounts *>(L.readback)->seq == L.seq;
  };
  int32_t turn = 0;
  for (int32_t f = f0; f < f1 && rc == PCP_OK; ++f) {
    int32_t pick = -1;
    const auto t_wait = std::chrono::steady_clock::now();
    for (uint32_t spins = 0; pick < 0 && spins < (1u << 22); ++spins) {
      for (int32_t k = 0; k < lanes && pick < 0; ++k)
        if (arrived(ctx->hpr_lane[(turn + k) % lanes])) pick = (turn + k) % lanes;
      if (pick < 0) __builtin_ia32_pause();
    }
    ctx->hpr_host_wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_wait).count();
    if (pick < 0) pick = turn;  // nothing for a long while: hpr_finish waits for this lane and looks after its stream
    HprLane &L = ctx->hpr_lane[pick];
    turn = (pick + 1) % lanes;
    if ((rc = hpr_finish(ctx, L, false)) != PCP_OK) break;
    uint32_t *plane = ctx->hull_bits.p + static_cast<size_t>(f >> 5) * static_cast<size_t>(ctx->n);
    rc = hpr_begin(ctx, L, L.own_stream, false, f, nullptr, plane, 1u << (f & 31), tile_mask);
  }
  // the keyframes still in flight, whichever count arrives first; then the join (also after an error: nothing may stay queued
  // on a lane whose buffers the next call reuses on another stream)
  for (int32_t left = lanes; left > 0; --left) {
    int32_t pick = -1;
    for (uint32_t spins = 0; pick < 0 && spins < (1u << 22); ++spins) {
      for (int32_t k = 0; k < lanes && pick < 0; ++k)
        if (ctx->hpr_lane[k].busy && arrived(ctx->hpr_lane[k])) pick = k;
      if (pick < 0) __builtin_ia32_pause();
    }
    if (pick < 0)
      for (int32_t k = 0; k < lanes && pick < 0; ++k)
        if (ctx->hpr_lane[k].busy) pick = k;
    if (pick < 0) break;  // no lane holds a keyframe any more
    const int rcl = hpr_finish(ctx, ctx->hpr_lane[pick], false);
    if (rc == PCP_OK) rc = rcl;
  }
  for (int32_t k = 0; k < forked; ++k) {
    // the context's stream continues behind the lane; where the join cannot even be queued, the lane is waited for here
    hipError_t e = hipEventRecord(ctx->hpr_join[k], ctx->hpr_lane[k].own_stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, ctx->hpr_join[k], 0);
    if (e != hipSuccess) {
      (void)hipStreamSynchronize(ctx->hpr_lane[k].own_stream);
      keep(e, "join");
    }
  }
  for (int32_t k = 0; k < pcp_context::kHprMaxLanes; ++k) ctx->hpr_lane[k].busy = false;
  if (std::getenv("PCP_HPR_HOST_TIMING"))
    fprintf(stderr, "hpr: pass of %d keyframes on %d lanes: host %.1f ms, of which %.1f ms waiting for counts\n", f1 - f0, lanes,
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t_pass).count() * 1e3, ctx->hpr_host_wait_s * 1e3);
  return rc;
}

}  // namespace pcp

namespace pcp {
// (pcp_create loads every code object of the library up front: see preload_code_objects in pcp_context.hip)
hipError_t preload_hpr() {
  hipFuncAttributes a;
  return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&k_hpr_zero));
}
}  // namespace pcp

using namespace pcp;

extern "C" {

int pcp_hpr_stats(pcp_context *ctx, int64_t out[10]) {
  if (!ctx || !out) return PCP_ERR_INVALID;
  const HprLane &last = ctx->hpr_lane[ctx->hpr_last_lane];
  if (ctx->hpr_stats_pending && last.stats.p) {
    // the tallies of the last run: block 0 + kStatCopies per-workgroup copies (k_hpr_decide), summed here
    PCP_HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::vector<unsigned long long> hall(kStatStride * (1 + kStatCopies));
    PCP_HIP_TRY(ctx, hipMemcpyAsync(hall.data(), last.stats.p, hall.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    PCP_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    unsigned long long hs[9];
    for (int k = 0; k < 9; ++k) {
      hs[k] = hall[static_cast<size_t>(k)];
      if (k != 8 && k != kStatPacked)
        for (int c = 1; c <= kStatCopies; ++c) hs[k] += hall[static_cast<size_t>(c * kStatStride + k)];
    }
    for (int c = 1; c <= kStatCopies; ++c) {  // the 16-lane passes' work, packed (kStatPacked)
      const unsigned long long w = hall[static_cast<size_t>(c * kStatStride + kStatPacked)];
      hs[3] += w & ((1ull << kStatPackedShift) - 1ull);
      hs[4] += w >> kStatPackedShift;
    }
    ctx->hpr_stats[0] = static_cast<int64_t>(hs[1]);  // visible, as finally classified (the exact path moved its points out of "undecided")
    ctx->hpr_stats[1] = static_cast<int64_t>(hs[0]);  // hidden
    ctx->hpr_stats[2] = static_cast<int64_t>(hs[8]);  // sent to the exact path
    for (int k = 3; k <= 7; ++k) ctx->hpr_stats[k] = static_cast<int64_t>(hs[k]);
    ctx->hpr_stats_pending = false;
  }
  for (int i = 0; i < 10; ++i) out[i] = ctx->hpr_stats[i];
  return PCP_OK;
}

}  // extern "C"
