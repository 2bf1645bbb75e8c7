// pcp_mls.hip -- the enableMLS path on gfx950: pcl::MovingLeastSquares
// (radius search r, order-2 weighted polynomial fit, SIMPLE projection,
// upsampling NONE) as CloudSmooth::process configures it
// (PCP/src/cloudSmooth.cpp:124-154, values PCP/src/PointCloudProcessor.cpp:67-86).
//
// The kd-tree of the reference is replaced by a uniform grid (cell >= r) built
// on the device: cell histogram -> exclusive scan -> scatter -> per-cell order
// fix-up (so results do not depend on atomic arrival order).  Points are then
// processed in cell order: the 64 lanes of a wavefront sit in one or two cells
// and walk the same 9 contiguous runs of neighbour candidates, so their loads
// are wave-wide broadcasts served by L1/L2.
//
// Numerics (Appendix A7): neighbour test in fp32 exactly as FLANN's L2_Simple
// ((dx*dx + dy*dy) + dz*dz < f32(r*r)); everything after it in fp64.  Moments
// are accumulated relative to the query point (exact differences of fp32
// values), which removes the reference's second neighbour sweep without losing
// digits.  Parity bar: 1e-4 relative on positions, normals up to sign.
#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <limits>

#include "pcp_internal.hpp"
#include "pcp_scan.hpp"

// The MLS / SOR stage is tolerance-gated (1e-4 relative, Appendix A9), not bit-gated: FMA
// contraction is allowed here.  The decisions that must be exact (neighbour membership,
// voxel indices, voxel positions) use explicit __f*_rn intrinsics, which never contract.
#pragma clang fp contract(fast)

namespace pcp {

constexpr int kMB = 256;
constexpr double kMaxGridCells = 536870912.0;      // 2^29: dense table of cell starts
constexpr double kMaxSparseCells = 34359738368.0;  // 2^35: bitmap (4 GiB) + running popcounts (2 GiB)

struct GridDesc {
  float minx, miny, minz, inv_cell;
  int32_t nx, ny, nz;
  int32_t reach;  // cells to visit on each side: ceil(r / cell)
  // Sparse form (nullptr: dense form, the table of cell starts has one entry per cell).  Grids of more than 2^29 cells keep
  // table entries for the OCCUPIED cells only; a bitmap with one bit per cell and the running popcount per 64-bit word
  // give the number of occupied cells before a cell -- its place in the table (1.5 bits per cell instead of 32).
  const unsigned long long *occ;
  const int32_t *occ_rank;
};

// number of occupied cells before cell c (sparse form)
__device__ __forceinline__ int32_t cell_rank(const GridDesc &g, int64_t c) {
  const unsigned long long bits = g.occ[c >> 6];
  return g.occ_rank[c >> 6] + static_cast<int32_t>(__popcll(bits & ((1ull << (c & 63)) - 1ull)));
}

// Entry (zz, yy, xx) of the table of cell starts = number of points in the cells before that cell (xx may be nx: the
// first cell of the next row).  The cells of a row are consecutive in either form, so start(x0) .. start(x1 + 1) is the
// run of candidates of the cells x0 .. x1 of a row.
__device__ __forceinline__ int32_t cell_start(const GridDesc &g, const int32_t *__restrict__ start, int32_t zz, int32_t yy,
                                              int32_t xx) {
  if (!g.occ) return start[(zz * g.ny + yy) * g.nx + xx];
  return start[cell_rank(g, (static_cast<int64_t>(zz) * g.ny + yy) * g.nx + xx)];
}

__device__ __forceinline__ void grid_coords(const GridDesc &g, float x, float y, float z, int32_t &ix, int32_t &iy,
                                            int32_t &iz) {
  ix = min(max(static_cast<int32_t>(floorf((x - g.minx) * g.inv_cell)), 0), g.nx - 1);
  iy = min(max(static_cast<int32_t>(floorf((y - g.miny) * g.inv_cell)), 0), g.ny - 1);
  iz = min(max(static_cast<int32_t>(floorf((z - g.minz) * g.inv_cell)), 0), g.nz - 1);
}

// cell id and arrival rank of every input point; histogram in `count`.  The views are spatially ordered (Morton copy of
// the upload, survivors of it), so a wavefront's 64 points fall into a handful of cells: the lanes of one cell share one
// atomic (ranks by lane, i.e. by index) instead of queueing 64 returning atomics on a few addresses (545 -> 60 us per
// 10 M points); after kAggRounds distinct cells the remaining lanes go alone (an unordered view would have 64 of them).
constexpr int kAggRounds = 8;
__global__ __launch_bounds__(kMB) void k_grid_count(const float *__restrict__ x, const float *__restrict__ y,
                                                    const float *__restrict__ z, int64_t n, GridDesc g,
                                                    int32_t *__restrict__ cell, int32_t *__restrict__ rank,
                                                    int32_t *__restrict__ count) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kMB + threadIdx.x;
  int32_t c = -1;
  if (i < n) {
    int32_t ix, iy, iz;
    grid_coords(g, x[i], y[i], z[i], ix, iy, iz);
    c = (iz * g.ny + iy) * g.nx + ix;
    cell[i] = c;
  }
  const int lane = threadIdx.x & 63;
  const unsigned long long below = (1ull << lane) - 1ull;
  bool todo = c >= 0;
  int32_t r = 0;
  for (int round = 0; round < kAggRounds; ++round) {
    const unsigned long long open = __ballot(todo);
    if (!open) break;
    const int32_t c0 = __shfl(c, __ffsll(open) - 1, 64);
    const bool mine = todo && c == c0;
    const unsigned long long same = __ballot(mine);
    if (mine) {
      int32_t base = 0;
      if ((same & below) == 0) base = atomicAdd(count + c0, static_cast<int32_t>(__popcll(same)));  // lowest lane of the cell
      base = __builtin_amdgcn_readfirstlane(base);
      r = base + static_cast<int32_t>(__popcll(same & below));
      todo = false;
    }
  }
  if (todo) r = atomicAdd(count + c, 1);
  if (i < n) rank[i] = r;
}

// Sparse form, step 1: one bit per occupied cell
__global__ __launch_bounds__(kMB) void k_grid_mark(const float *__restrict__ x, const float *__restrict__ y,
                                                   const float *__restrict__ z, int64_t n, GridDesc g,
                                                   unsigned long long *__restrict__ occ) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kMB + threadIdx.x;
  if (i >= n) return;
  int32_t ix, iy, iz;
  grid_coords(g, x[i], y[i], z[i], ix, iy, iz);
  const int64_t c = (static_cast<int64_t>(iz) * g.ny + iy) * g.nx + ix;
  const unsigned long long bit = 1ull << (c & 63);
  if (!(occ[c >> 6] & bit)) atomicOr(occ + (c >> 6), bit);  // spatially ordered views: most bits are set already
}
// step 2: set bits per word (scanned in place into the running popcount)
__global__ __launch_bounds__(kMB) void k_grid_popc(const unsigned long long *__restrict__ occ, int64_t words,
                                                   int32_t *__restrict__ count) {
  const int64_t w = static_cast<int64_t>(blockIdx.x) * kMB + threadIdx.x;
  if (w < words) count[w] = static_cast<int32_t>(__popcll(occ[w]));
}
// step 3: as k_grid_count, with the cell's place among the occupied cells as its id
__global__ __launch_bounds__(kMB) void k_grid_count_sparse(const float *__restrict__ x, const float *__restrict__ y,
                                                           const float *__restrict__ z, int64_t n, GridDesc g,
                                                           int32_t *__restrict__ cell, int32_t *__restrict__ rank,
                                                           int32_t *__restrict__ count) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kMB + threadIdx.x;
  int32_t c = -1;
  if (i < n) {
    int32_t ix, iy, iz;
    grid_coords(g, x[i], y[i], z[i], ix, iy, iz);
    c = cell_rank(g, (static_cast<int64_t>(iz) * g.ny + iy) * g.nx + ix);
    cell[i] = c;
  }
  const int lane = threadIdx.x & 63;
  const unsigned long long below = (1ull << lane) - 1ull;
  bool todo = c >= 0;
  int32_t r = 0;
  for (int round = 0; round < kAggRounds; ++round) {
    const unsigned long long open = __ballot(todo);
    if (!open) break;
    const int32_t c0 = __shfl(c, __ffsll(open) - 1, 64);
    const bool mine = todo && c == c0;
    const unsigned long long same = __ballot(mine);
    if (mine) {
      int32_t base = 0;
      if ((same & below) == 0) base = atomicAdd(count + c0, static_cast<int32_t>(__popcll(same)));
      base = __builtin_amdgcn_readfirstlane(base);
      r = base + static_cast<int32_t>(__popcll(same & below));
      todo = false;
    }
  }
  if (todo) r = atomicAdd(count + c, 1);
  if (i < n) rank[i] = r;
}

// surface-density probe: every `stride`-th point is binned and the cells that receive their first point are counted
// (the estimate only sizes the SOR grid: any value gives the same, exact, result)
__global__ __launch_bounds__(kMB) void k_grid_probe(const float *__restrict__ x, const float *__restrict__ y,
                                                    const float *__restrict__ z, int64_t n, int64_t stride, GridDesc g,
                                                    int32_t *__restrict__ count, unsigned long long *__restrict__ occupied) {
  const int64_t i = (static_cast<int64_t>(blockIdx.x) * kMB + threadIdx.x) * stride;
  bool first = false;
  if (i < n) {
    int32_t ix, iy, iz;
    grid_coords(g, x[i], y[i], z[i], ix, iy, iz);
    first = atomicAdd(count + ((iz * g.ny + iy) * g.nx + ix), 1) == 0;
  }
  // one atomic per workgroup (atomics on one address queue up in its L2 channel, ~7 ns each: one per wavefront of every
  // 8th of 10 M points was 0.1 ms)
  __shared__ uint32_t firsts[kMB / 64];
  const unsigned long long m = __ballot(first);
  if ((threadIdx.x & 63) == 0) firsts[threadIdx.x >> 6] = static_cast<uint32_t>(__popcll(m));
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t c = 0;
    for (int k = 0; k < kMB / 64; ++k) c += firsts[k];
    if (c) atomicAdd(occupied, static_cast<unsigned long long>(c));
  }
}

__global__ __launch_bounds__(kMB) void k_grid_scatter(int64_t n, const int32_t *__restrict__ cell,
                                                      const int32_t *__restrict__ rank,
                                                      const int32_t *__restrict__ start,
                                                      int32_t *__restrict__ order) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kMB + threadIdx.x;
  if (i >= n) return;
  order[start[cell[i]] + rank[i]] = static_cast<int32_t>(i);
}

// make the order inside every cell ascending in the input index (deterministic
// neighbour order), then gather the coordinates into cell order.  One lane per
// slot p of the scattered array: its point i = order_in[p] counts the members of
// its cell [b, e) that precede it.  The workgroup's 256 slots sit in LDS, and a
// cell holds a few dozen points, so nearly every comparison is an LDS read; the
// parts of a cell outside the window come from memory.  (The first version went
// through global memory for every member: 1.4 ms per 10 M-point SOR grid.)
__global__ __launch_bounds__(kMB) void k_grid_order(const float *__restrict__ x, const float *__restrict__ y,
                                                    const float *__restrict__ z, int64_t n, int64_t plane,
                                                    const int32_t *__restrict__ cell,
                                                    const int32_t *__restrict__ start,
                                                    const int32_t *__restrict__ order_in,
                                                    int32_t *__restrict__ order_out, float *__restrict__ sxyz) {
  __shared__ int32_t win[kMB];
  const int64_t w0 = static_cast<int64_t>(blockIdx.x) * kMB;
  const int64_t p = w0 + threadIdx.x;
  const int32_t i = p < n ? order_in[p] : 0x7fffffff;
  win[threadIdx.x] = i;
  __syncthreads();
  if (p >= n) return;
  const int32_t c = cell[i];
  const int64_t b = start[c], e = start[c + 1];
  int32_t before = 0;
  const int64_t lb = max(b, w0), le = min(e, w0 + kMB);  // the part of the cell inside the window
  for (int64_t k = b; k < lb; ++k) before += order_in[k] < i ? 1 : 0;
  for (int64_t k = lb; k < le; ++k) before += win[k - w0] < i ? 1 : 0;
  for (int64_t k = max(le, b); k < e; ++k) before += order_in[k] < i ? 1 : 0;
  const int64_t j = b + before;
  order_out[j] = i;
  sxyz[j] = x[i];
  sxyz[plane + j] = y[i];
  sxyz[2 * plane + j] = z[i];
}

// 1 / x and 1 / sqrt(x) in fp64 to an ulp or two: the hardware estimate and Newton steps, without the range scaling, the exact
// residual correction and the special-case fix-ups of the IEEE sequences (~8 instead of ~28 instructions).  For the fit's own
// arithmetic (tolerance-gated against the oracle: 3 um, 1e-4), whose operands are covariances, lengths and pivots -- normal
// numbers; zero, negative and non-finite arguments give infinities / NaNs that the callers' guards (ok, isfinite) catch as before.
__device__ __forceinline__ double mls_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-x, r, 1.0);
  return __builtin_fma(r, e, r);
}
__device__ __forceinline__ double mls_rsqrt(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  const double e = __builtin_fma(-(x * y), y, 1.0);
  return __builtin_fma(y * e, __builtin_fma(0.375, e, 0.5), y);
}

// ---- pcl::eigen33 smallest eigenpair (common/impl/eigen.hpp) [upstream] --------
__device__ __forceinline__ void roots2(double b, double c, double &r0, double &r1, double &r2) {
  r0 = 0.0;
  double d = b * b - 4.0 * c;
  if (d < 0.0) d = 0.0;
  const double sd = sqrt(d);
  r2 = 0.5 * (b + sd);
  r1 = 0.5 * (b - sd);
}

__device__ __forceinline__ void swap2(double &a, double &b) {
  const double t = a;
  a = b;
  b = t;
}

// (inlined: as a call it cost the kernel 96 bytes of scratch per lane and a save / restore around it -- fit 4.25 -> 4.12 ms)
__device__ __forceinline__ void smallest_eigenpair(const double m[6] /* xx xy xz yy yz zz */, double &ev, double n[3]) {
  double scale = fmax(fmax(fmax(fabs(m[0]), fabs(m[1])), fmax(fabs(m[2]), fabs(m[3]))), fmax(fabs(m[4]), fabs(m[5])));
  if (scale <= DBL_MIN) scale = 1.0;
  const double inv_scale = mls_rcp(scale);
  const double a00 = m[0] * inv_scale, a01 = m[1] * inv_scale, a02 = m[2] * inv_scale, a11 = m[3] * inv_scale,
               a12 = m[4] * inv_scale, a22 = m[5] * inv_scale;
  const double c0 = a00 * a11 * a22 + 2.0 * a01 * a02 * a12 - a00 * a12 * a12 - a11 * a02 * a02 - a22 * a01 * a01;
  const double c1 = a00 * a11 - a01 * a01 + a00 * a22 - a02 * a02 + a11 * a22 - a12 * a12;
  const double c2 = a00 + a11 + a22;
  double r0, r1, r2;
  if (fabs(c0) < DBL_EPSILON) {
    roots2(c2, c1, r0, r1, r2);
  } else {
    const double inv3 = 1.0 / 3.0;
    const double sqrt3 = sqrt(3.0);
    const double c2_3 = c2 * inv3;
    double a_3 = (c1 - c2 * c2_3) * inv3;
    if (a_3 > 0.0) a_3 = 0.0;
    const double half_b = 0.5 * (c0 + c2_3 * (2.0 * c2_3 * c2_3 - c1));
    double q = half_b * half_b + a_3 * a_3 * a_3;
    if (q > 0.0) q = 0.0;
    const double rho = sqrt(-a_3);
    const double theta = atan2(sqrt(-q), half_b) * inv3;
    const double ct = cos(theta), st = sin(theta);
    r0 = c2_3 + 2.0 * rho * ct;
    r1 = c2_3 - rho * (ct + sqrt3 * st);
    r2 = c2_3 - rho * (ct - sqrt3 * st);
    if (r0 >= r1) swap2(r0, r1);
    if (r1 >= r2) {
      swap2(r1, r2);
      if (r0 >= r1) swap2(r0, r1);
    }
    if (r0 <= 0.0) roots2(c2, c1, r0, r1, r2);
  }
  ev = r0 * scale;
  // getLargest3x3Eigenvector of (A - r0 I): longest cross product of two rows
  const double s00 = a00 - r0, s11 = a11 - r0, s22 = a22 - r0;
  const double k0x = a01 * a12 - a02 * s11, k0y = a02 * a01 - s00 * a12, k0z = s00 * s11 - a01 * a01;  // row0 x row1
  const double k1x = a01 * s22 - a02 * a12, k1y = a02 * a02 - s00 * s22, k1z = s00 * a12 - a01 * a02;  // row0 x row2
  const double k2x = s11 * s22 - a12 * a12, k2y = a12 * a02 - a01 * s22, k2z = a01 * a12 - s11 * a02;  // row1 x row2
  const double l0 = (k0x * k0x + k0y * k0y) + k0z * k0z;
  const double l1 = (k1x * k1x + k1y * k1y) + k1z * k1z;
  const double l2 = (k2x * k2x + k2y * k2y) + k2z * k2z;
  double vx = k0x, vy = k0y, vz = k0z, l = l0;
  if (l1 > l) {
    vx = k1x; vy = k1y; vz = k1z; l = l1;
  }
  if (l2 > l) {
    vx = k2x; vy = k2y; vz = k2z; l = l2;
  }
  const double inv_len = mls_rsqrt(l);  // (l == 0: infinity, the components NaN as with the division by zero)
  n[0] = vx * inv_len;
  n[1] = vy * inv_len;
  n[2] = vz * inv_len;
}

// doubles kept per input point for the upsampling stage (MLSResult)
constexpr int kMlsState = 22;  // mean3 normal3 u3 v3 c6 curvature K valid(+fitted) pad
constexpr int kRowStride = 8;  // floats per fitted row in ctx->m_tmp: xyz, normal, curvature, pad -- one 32-byte sector

struct MlsArgs {
  const float *sx, *sy, *sz;     // cell-sorted coordinates
  const int32_t *order;          // cell-sorted -> view index
  const int32_t *remap;          // view index -> caller's index (nullable)
  const int32_t *start;          // cell starts (ncell + 1)
  int64_t n;
  GridDesc g;
  float sq_radius;               // f32(r*r): kdtree_flann radiusSearch [upstream]
  double inv_sq_radius;          // 1 / (r*r), weight exp(-d^2 / r^2) (B13)
  int32_t order_poly;
  float *tmp;                    // kRowStride floats per input point (32-byte aligned rows)
  uint8_t *flag;                 // per input point
  double *state;                 // kMlsState doubles per input point (nullable)
  int32_t q_begin, q_end;        // only queries with q_begin <= input index < q_end are fitted (query sharding by index)
  int64_t j_begin, j_end;        // the places of the cell order this launch covers (query sharding by slabs)
};

__device__ __forceinline__ float sqdist_f32(float ax, float ay, float az, float bx, float by, float bz) {
  const float dx = __fsub_rn(ax, bx), dy = __fsub_rn(ay, by), dz = __fsub_rn(az, bz);
  return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// neighbour list of one lane: up to kMaxNbr entries (run << 12 | offset in run) in LDS, column
// layout [entry][lane].  With the candidates compacted first, both accumulation sweeps run with
// every lane busy (in the plain double loop only the ~35 % of lanes whose current candidate is
// inside the radius do fp64 work).  Lanes whose neighbourhood does not fit (K > kMaxNbr, a run
// longer than 4096 points, or a grid with reach > 1) take the plain loops.
#ifndef PCP_MLS_MAXNBR
#define PCP_MLS_MAXNBR 87
#endif
constexpr int kMaxNbr = PCP_MLS_MAXNBR;  // 13.3 KB of LDS per wavefront: 12 wavefronts per CU, what the 140 VGPRs allow (96: 11 wavefronts, fit +3 %;
                             // 72: the 4.6 % of lanes with more neighbours take the plain loops, fit +38 %)
constexpr int kMaxRun = 9;
constexpr int kFitBlock = 64;

// kBuf: the cell-sorted coordinate planes are read through buffer descriptors (uniform base, 32-bit byte offset per
// lane: one shift per candidate instead of three 64-bit addresses); needs n < 2^30 points.
template <bool kBuf>
__global__ __launch_bounds__(kFitBlock) void k_mls_fit(MlsArgs a) {
  __shared__ uint16_t nbr_code[kMaxNbr + 1][kFitBlock];  // + the row that takes the writes once the list is full
  __shared__ int32_t run_base[kMaxRun][kFitBlock];
  const uint32_t plane_bytes = kBuf ? static_cast<uint32_t>(a.n) * 4u : 0u;
  const auto rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.sx), 0, plane_bytes, 0x00020000);
  const auto rsy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.sy), 0, plane_bytes, 0x00020000);
  const auto rsz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.sz), 0, plane_bytes, 0x00020000);
  auto fetch = [&](int32_t k, float &x, float &y, float &z) {
    if constexpr (kBuf) {
      const uint32_t off = static_cast<uint32_t>(k) * 4u;
      x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsx, off, 0, 0));
      y = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsy, off, 0, 0));
      z = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsz, off, 0, 0));
    } else {
      x = a.sx[k];
      y = a.sy[k];
      z = a.sz[k];
    }
  };
  const int tid = threadIdx.x;
  const int64_t j = a.j_begin + static_cast<int64_t>(blockIdx.x) * kFitBlock + threadIdx.x;
  if (j >= a.j_end) return;
  const int32_t i = a.remap ? a.remap[a.order[j]] : a.order[j];
  if (i < a.q_begin || i >= a.q_end) {  // another shard's query
    a.flag[i] = 0;
    return;
  }
  const float qx = a.sx[j], qy = a.sy[j], qz = a.sz[j];
  int32_t cx, cy, cz;
  grid_coords(a.g, qx, qy, qz, cx, cy, cz);
  const int32_t x0 = max(cx - a.g.reach, 0), x1 = min(cx + a.g.reach, a.g.nx - 1);
  const int32_t y0 = max(cy - a.g.reach, 0), y1 = min(cy + a.g.reach, a.g.ny - 1);
  const int32_t z0 = max(cz - a.g.reach, 0), z1 = min(cz + a.g.reach, a.g.nz - 1);

  // sweep 0: radius test (fp32, exact FLANN form), neighbour count, compact list.  The run bounds
  // are fetched first (18 independent loads), and the candidates of a run are tested four at a time
  // (12 coordinate loads in flight: this scan is latency-bound, not arithmetic-bound).
  int32_t K = 0;
  bool fast = a.g.reach == 1;
  if (fast) {
    int32_t rb[kMaxRun], re[kMaxRun];
#pragma unroll
    for (int32_t r = 0; r < kMaxRun; ++r) {
      const int32_t zz = z0 + r / 3, yy = y0 + r % 3;
      const bool in = zz <= z1 && yy <= y1;  // fewer rows at the grid border: empty runs keep the (z, y) order
      rb[r] = in ? cell_start(a.g, a.start, min(zz, z1), min(yy, y1), x0) : 0;
      re[r] = in ? cell_start(a.g, a.start, min(zz, z1), min(yy, y1), x1 + 1) : 0;
    }
#pragma unroll
    for (int32_t r = 0; r < kMaxRun; ++r) {
      const int32_t b = rb[r], e = re[r];
      run_base[r][tid] = b;
      if (e - b > 4096) fast = false;
      // branch-free: every candidate is written to the list's next free row (a hit keeps it by moving on, a miss is
      // overwritten by the next candidate); row kMaxNbr takes what comes after the list is full
      auto append = [&](bool hit, int32_t k) {
        // (k - b < 4096 whenever the list is used: a longer run clears `fast`; unmasked, the code is one scalar add)
        nbr_code[min(K, kMaxNbr)][tid] = static_cast<uint16_t>((r << 12) + (k - b));
        K += hit ? 1 : 0;
      };
      if constexpr (kBuf) {
        // four candidates per trip: one 16-byte load per coordinate plane (dword alignment suffices for buffer loads),
        // packed fp32 arithmetic on two candidates at a time, each component rounded as sqdist_f32 does; the ragged
        // end of a run is one more trip whose missing candidates sit at infinity
#pragma clang fp contract(off)
        typedef float v2f __attribute__((ext_vector_type(2)));
        typedef uint32_t v4u __attribute__((ext_vector_type(4)));
        const v2f qx2 = {qx, qx}, qy2 = {qy, qy}, qz2 = {qz, qz};
        auto d2_of = [&](v2f px, v2f py, v2f pz) {
          const v2f dx = px - qx2, dy = py - qy2, dz = pz - qz2;
          return (dx * dx + dy * dy) + dz * dz;
        };
        auto load4 = [&](decltype(rsx) rs, int32_t k, float (&v)[4]) {
          const v4u w = __builtin_amdgcn_raw_buffer_load_b128(rs, static_cast<uint32_t>(k) * 4u, 0, 0);
#pragma unroll
          for (int c = 0; c < 4; ++c) v[c] = __uint_as_float(w[c]);
        };
        int32_t k = b;
        for (; k + 4 <= e; k += 4) {
          float X[4], Y[4], Z[4];
          load4(rsx, k, X);
          load4(rsy, k, Y);
          load4(rsz, k, Z);
          const v2f da = d2_of(v2f{X[0], X[1]}, v2f{Y[0], Y[1]}, v2f{Z[0], Z[1]});
          const v2f db = d2_of(v2f{X[2], X[3]}, v2f{Y[2], Y[3]}, v2f{Z[2], Z[3]});
          append(da[0] < a.sq_radius, k);
          append(da[1] < a.sq_radius, k + 1);
          append(db[0] < a.sq_radius, k + 2);
          append(db[1] < a.sq_radius, k + 3);
        }
        if (k < e) {
          const int32_t rem = e - k;
          float X[4], Y[4], Z[4];
          load4(rsx, k, X);
          load4(rsy, k, Y);
          load4(rsz, k, Z);
          const v2f da = d2_of(v2f{X[0], rem > 1 ? X[1] : INFINITY}, v2f{Y[0], Y[1]}, v2f{Z[0], Z[1]});
          const v2f db = d2_of(v2f{rem > 2 ? X[2] : INFINITY, INFINITY}, v2f{Y[2], Y[3]}, v2f{Z[2], Z[3]});
          append(da[0] < a.sq_radius, k);
          append(da[1] < a.sq_radius, k + 1);
          append(db[0] < a.sq_radius, k + 2);
        }
      } else {
        for (int32_t k = b; k < e; k += 4) {
          const int32_t k1 = min(k + 1, e - 1), k2 = min(k + 2, e - 1), k3 = min(k + 3, e - 1);
          float ax, ay, az, bx, by, bz, cx_, cy_, cz_, dx_, dy_, dz_;
          fetch(k, ax, ay, az);
          fetch(k1, bx, by, bz);
          fetch(k2, cx_, cy_, cz_);
          fetch(k3, dx_, dy_, dz_);
          append(sqdist_f32(ax, ay, az, qx, qy, qz) < a.sq_radius, k);
          append(k + 1 < e && sqdist_f32(bx, by, bz, qx, qy, qz) < a.sq_radius, k + 1);
          append(k + 2 < e && sqdist_f32(cx_, cy_, cz_, qx, qy, qz) < a.sq_radius, k + 2);
          append(k + 3 < e && sqdist_f32(dx_, dy_, dz_, qx, qy, qz) < a.sq_radius, k + 3);
        }
      }
    }
  } else {
    for (int32_t zz = z0; zz <= z1; ++zz)
      for (int32_t yy = y0; yy <= y1; ++yy) {
        const int32_t b = cell_start(a.g, a.start, zz, yy, x0), e = cell_start(a.g, a.start, zz, yy, x1 + 1);
        for (int32_t k = b; k < e; ++k)
          if (sqdist_f32(a.sx[k], a.sy[k], a.sz[k], qx, qy, qz) < a.sq_radius) ++K;
      }
  }
  fast = fast && K <= kMaxNbr;
  // visit every neighbour once: body(px, py, pz)
  auto for_each_neighbour = [&](auto &&body) {
    if (fast) {
      // the next two neighbours' coordinates are in flight while the current one is accumulated
      auto index_of = [&](int32_t t) {
        const uint32_t code = nbr_code[min(t, K - 1)][tid];
        return run_base[code >> 12][tid] + static_cast<int32_t>(code & 4095u);
      };
      float ax, ay, az, bx, by, bz;
      fetch(index_of(0), ax, ay, az);
      fetch(index_of(1), bx, by, bz);
      for (int32_t t = 0; t < K; ++t) {
        const float px = ax, py = ay, pz = az;
        ax = bx; ay = by; az = bz;
        fetch(index_of(t + 2), bx, by, bz);
        body(px, py, pz);
      }
    } else {
      for (int32_t zz = z0; zz <= z1; ++zz)
        for (int32_t yy = y0; yy <= y1; ++yy) {
          const int32_t b = cell_start(a.g, a.start, zz, yy, x0), e = cell_start(a.g, a.start, zz, yy, x1 + 1);
          for (int32_t k = b; k < e; ++k) {
            const float px = a.sx[k], py = a.sy[k], pz = a.sz[k];
            if (sqdist_f32(px, py, pz, qx, qy, qz) < a.sq_radius) body(px, py, pz);
          }
        }
    }
  };

  // From here on the arithmetic is the fit's own fp64 work on fp32 data, compared with the oracle within tolerances (3 um,
  // 1e-4), not bit for bit -- PCL's Eigen code is itself free to fuse -- so the multiply-adds of the two sweeps are written
  // as fma (the compiler had fused them already: the object code did not change).  The neighbour SELECTION above (fp32,
  // FLANN's exact form) decides index sets and keeps its individually rounded operations.
  // sweep 1: moments of (p - q)
  double s1x = 0, s1y = 0, s1z = 0, sxx = 0, sxy = 0, sxz = 0, syy = 0, syz = 0, szz = 0;
  if (K >= 3)
    for_each_neighbour([&](float px, float py, float pz) {
      const double dx = static_cast<double>(px) - static_cast<double>(qx);
      const double dy = static_cast<double>(py) - static_cast<double>(qy);
      const double dz = static_cast<double>(pz) - static_cast<double>(qz);
      s1x += dx; s1y += dy; s1z += dz;
      sxx = fma(dx, dx, sxx); sxy = fma(dx, dy, sxy); sxz = fma(dx, dz, sxz);
      syy = fma(dy, dy, syy); syz = fma(dy, dz, syz); szz = fma(dz, dz, szz);
    });
  if (K < 3) {  // MovingLeastSquares::performProcessing skips the point
    a.flag[i] = 0;
    return;
  }
  const double invK = mls_rcp(static_cast<double>(K));
  const double mx = s1x * invK, my = s1y * invK, mz = s1z * invK;  // centroid - q
  double C[6];
  C[0] = sxx - s1x * mx;
  C[1] = sxy - s1x * my;
  C[2] = sxz - s1x * mz;
  C[3] = syy - s1y * my;
  C[4] = syz - s1y * mz;
  C[5] = szz - s1z * mz;
  double ev, nrm[3];
  smallest_eigenpair(C, ev, nrm);
  // rows of kRowStride = 8 floats (xyz, normal, curvature, pad): ONE aligned 32-byte sector each, written by two 16-byte stores --
  // the row lands in view-index order (scattered), and seven 4-byte stores into a 28-byte row cost six partial sector writes
  // (WRITE_SIZE 1.69 GB for 0.28 GB of rows, profiles/r04_pmc.json)
  float4 *out4 = reinterpret_cast<float4 *>(a.tmp + static_cast<int64_t>(i) * kRowStride);
  double *st = a.state ? a.state + static_cast<int64_t>(i) * kMlsState : nullptr;
  const double Qx = qx, Qy = qy, Qz = qz;
  if (!isfinite(nrm[0]) || !isfinite(nrm[1]) || !isfinite(nrm[2])) {
    out4[0] = make_float4(qx, qy, qz, 0.0f);
    out4[1] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    a.flag[i] = 1;
    if (st) {
      for (int k = 0; k < kMlsState; ++k) st[k] = 0.0;
      st[0] = Qx; st[1] = Qy; st[2] = Qz;
    }
    return;
  }
  // projected query point: distance = (q - centroid) . n
  const double distance = -((mx * nrm[0] + my * nrm[1]) + mz * nrm[2]);
  const double meanx = Qx - distance * nrm[0], meany = Qy - distance * nrm[1], meanz = Qz - distance * nrm[2];
  double curv = (C[0] + C[3]) + C[5];
  if (curv != 0.0) curv = fabs(ev / curv);
  // Darboux frame: v = n.unitOrthogonal(), u = n x v (Eigen OrthoMethods.h)
  double vx, vy, vz;
  if (!(fabs(nrm[0]) <= fabs(nrm[2]) * 1e-12) || !(fabs(nrm[1]) <= fabs(nrm[2]) * 1e-12)) {
    const double inv = mls_rsqrt(nrm[0] * nrm[0] + nrm[1] * nrm[1]);
    vx = -nrm[1] * inv; vy = nrm[0] * inv; vz = 0.0;
  } else {
    const double inv = mls_rsqrt(nrm[1] * nrm[1] + nrm[2] * nrm[2]);
    vx = 0.0; vy = -nrm[2] * inv; vz = nrm[1] * inv;
  }
  const double ux = nrm[1] * vz - nrm[2] * vy, uy = nrm[2] * vx - nrm[0] * vz, uz = nrm[0] * vy - nrm[1] * vx;

  double c[6] = {0, 0, 0, 0, 0, 0};
  bool fitted = false;
  if (a.order_poly > 1 && K >= 6) {
    // sweep 2: weighted normal equations of the 6 monomials (1, v, v^2, u, uv, u^2)
    double A00 = 0, A01 = 0, A02 = 0, A03 = 0, A04 = 0, A05 = 0, A11 = 0, A12 = 0, A13 = 0, A14 = 0, A15 = 0, A22 = 0,
           A23 = 0, A24 = 0, A25 = 0, A33 = 0, A34 = 0, A35 = 0, A44 = 0, A45 = 0, A55 = 0;
    double b0 = 0, b1 = 0, b2 = 0, b3 = 0, b4 = 0, b5 = 0;
    for_each_neighbour([&](float px, float py, float pz) {
      const double dx = static_cast<double>(px) - meanx, dy = static_cast<double>(py) - meany,
                   dz = static_cast<double>(pz) - meanz;
      // The Gaussian weight through the fp32 hardware exponential: the argument lies in [-1.1, 0], so v_exp_f32's
      // ~2e-7 relative error moves the fitted surface by ~1e-7 of its millimetre-scale offset -- five orders below
      // the 3 um parity bar -- while ocml's fp64 exp costs ~40 of this loop's ~100 VALU instructions per neighbour
      // (k_mls_fit 10 M points: 5.99 -> 5.5 ms).
      const double d2 = fma(dz, dz, fma(dy, dy, dx * dx));
      const double w = static_cast<double>(__expf(static_cast<float>(-d2 * a.inv_sq_radius)));
      const double uc = fma(dz, uz, fma(dy, uy, dx * ux));
      const double vc = fma(dz, vz, fma(dy, vy, dx * vx));
      const double f = fma(dz, nrm[2], fma(dy, nrm[1], dx * nrm[0]));
      const double p1 = vc, p2 = vc * vc, p3 = uc, p4 = uc * vc, p5 = uc * uc;
      const double w1 = w * p1, w2 = w * p2, w3 = w * p3, w4 = w * p4, w5 = w * p5;
      // the 21 entries of P W P^T are sums of w u^a v^b with a + b <= 4: only 15 distinct monomials
      A00 += w; A01 += w1; A02 += w2; A03 += w3; A04 += w4; A05 += w5;
      A12 = fma(w1, p2, A12); A14 = fma(w1, p4, A14); A15 = fma(w1, p5, A15);
      A22 = fma(w2, p2, A22); A24 = fma(w2, p4, A24); A25 = fma(w2, p5, A25);
      A35 = fma(w3, p5, A35);
      A45 = fma(w4, p5, A45);
      A55 = fma(w5, p5, A55);
      b0 = fma(w, f, b0); b1 = fma(w1, f, b1); b2 = fma(w2, f, b2); b3 = fma(w3, f, b3); b4 = fma(w4, f, b4);
      b5 = fma(w5, f, b5);
    });
    A11 = A02;  // w v^2
    A13 = A04;  // w u v
    A33 = A05;  // w u^2
    A23 = A14;  // w u v^2
    A34 = A15;  // w u^2 v
    A44 = A25;  // w u^2 v^2
    // LLT (lower) + forward / backward substitution, fully unrolled in registers
    bool ok = true;
    double L00, L10, L20, L30, L40, L50, L11, L21, L31, L41, L51, L22, L32, L42, L52, L33, L43, L53, L44, L54, L55;
    double d;
    // the divisions by the six pivots are multiplications by their reciprocals (27 -> 6 divisions)
    double i0, i1, i2, i3, i4, i5;
    d = A00; ok = ok && d > 0.0; i0 = mls_rsqrt(d); L00 = d * i0;
    L10 = A01 * i0; L20 = A02 * i0; L30 = A03 * i0; L40 = A04 * i0; L50 = A05 * i0;
    d = A11 - L10 * L10; ok = ok && d > 0.0; i1 = mls_rsqrt(d); L11 = d * i1;
    L21 = (A12 - L20 * L10) * i1; L31 = (A13 - L30 * L10) * i1; L41 = (A14 - L40 * L10) * i1;
    L51 = (A15 - L50 * L10) * i1;
    d = A22 - L20 * L20 - L21 * L21; ok = ok && d > 0.0; i2 = mls_rsqrt(d); L22 = d * i2;
    L32 = (A23 - L30 * L20 - L31 * L21) * i2; L42 = (A24 - L40 * L20 - L41 * L21) * i2;
    L52 = (A25 - L50 * L20 - L51 * L21) * i2;
    d = A33 - L30 * L30 - L31 * L31 - L32 * L32; ok = ok && d > 0.0; i3 = mls_rsqrt(d); L33 = d * i3;
    L43 = (A34 - L40 * L30 - L41 * L31 - L42 * L32) * i3; L53 = (A35 - L50 * L30 - L51 * L31 - L52 * L32) * i3;
    d = A44 - L40 * L40 - L41 * L41 - L42 * L42 - L43 * L43; ok = ok && d > 0.0; i4 = mls_rsqrt(d); L44 = d * i4;
    L54 = (A45 - L50 * L40 - L51 * L41 - L52 * L42 - L53 * L43) * i4;
    d = A55 - L50 * L50 - L51 * L51 - L52 * L52 - L53 * L53 - L54 * L54; ok = ok && d > 0.0; i5 = mls_rsqrt(d); L55 = d * i5;
    if (ok) {
      const double y0_ = b0 * i0;
      const double y1_ = (b1 - L10 * y0_) * i1;
      const double y2_ = (b2 - L20 * y0_ - L21 * y1_) * i2;
      const double y3_ = (b3 - L30 * y0_ - L31 * y1_ - L32 * y2_) * i3;
      const double y4_ = (b4 - L40 * y0_ - L41 * y1_ - L42 * y2_ - L43 * y3_) * i4;
      const double y5_ = (b5 - L50 * y0_ - L51 * y1_ - L52 * y2_ - L53 * y3_ - L54 * y4_) * i5;
      c[5] = y5_ * i5;
      c[4] = (y4_ - L54 * c[5]) * i4;
      c[3] = (y3_ - L43 * c[4] - L53 * c[5]) * i3;
      c[2] = (y2_ - L32 * c[3] - L42 * c[4] - L52 * c[5]) * i2;
      c[1] = (y1_ - L21 * c[2] - L31 * c[3] - L41 * c[4] - L51 * c[5]) * i1;
      c[0] = (y0_ - L10 * c[1] - L20 * c[2] - L30 * c[3] - L40 * c[4] - L50 * c[5]) * i0;
    } else {
      c[0] = c[1] = c[2] = c[3] = c[4] = c[5] = NAN;  // Eigen's LLT yields NaN; PCL then uses the plane
    }
    fitted = true;
  }
  // MLSResult::projectQueryPoint(SIMPLE, nr_coeff)
  double ox = meanx, oy = meany, oz = meanz, nx = nrm[0], ny = nrm[1], nz = nrm[2];
  if (fitted && isfinite(c[0])) {
    ox = meanx + c[0] * nrm[0];
    oy = meany + c[0] * nrm[1];
    oz = meanz + c[0] * nrm[2];
    nx = nrm[0] - c[3] * ux - c[1] * vx;
    ny = nrm[1] - c[3] * uy - c[1] * vy;
    nz = nrm[2] - c[3] * uz - c[1] * vz;
    const double l2 = (nx * nx + ny * ny) + nz * nz;
    if (l2 > 0.0) {
      const double il = mls_rsqrt(l2);
      nx *= il; ny *= il; nz *= il;
    }
  }
  out4[0] = make_float4(static_cast<float>(ox), static_cast<float>(oy), static_cast<float>(oz), static_cast<float>(nx));
  out4[1] = make_float4(static_cast<float>(ny), static_cast<float>(nz), static_cast<float>(curv), 0.0f);
  a.flag[i] = 1;
  if (st) {
    st[0] = meanx; st[1] = meany; st[2] = meanz;
    st[3] = nrm[0]; st[4] = nrm[1]; st[5] = nrm[2];
    st[6] = ux; st[7] = uy; st[8] = uz;
    st[9] = vx; st[10] = vy; st[11] = vz;
    for (int k = 0; k < 6; ++k) st[12 + k] = c[k];
    st[18] = curv;
    st[19] = static_cast<double>(K);
    st[20] = fitted ? 2.0 : 1.0;  // 0 invalid, 1 valid plane only, 2 polynomial fitted
    st[21] = 0.0;
  }
}

// pcp_close_pairs: per point, is there ANOTHER map point closer than r (fp32 L2_Simple distance, strict <)?
__global__ __launch_bounds__(kMB) void k_close_pairs(const float *__restrict__ sx, const float *__restrict__ sy,
                                                     const float *__restrict__ sz, const int32_t *__restrict__ start,
                                                     int64_t n, GridDesc g, float sq_radius,
                                                     unsigned long long *__restrict__ count) {
  const int64_t j = static_cast<int64_t>(blockIdx.x) * kMB + threadIdx.x;
  bool close = false;
  if (j < n) {
    const float qx = sx[j], qy = sy[j], qz = sz[j];
    int32_t cx, cy, cz;
    grid_coords(g, qx, qy, qz, cx, cy, cz);
    for (int32_t zz = max(cz - g.reach, 0); zz <= min(cz + g.reach, g.nz - 1) && !close; ++zz)
      for (int32_t yy = max(cy - g.reach, 0); yy <= min(cy + g.reach, g.ny - 1) && !close; ++yy) {
        const int32_t b = cell_start(g, start, zz, yy, max(cx - g.reach, 0)), e = cell_start(g, start, zz, yy, min(cx + g.reach, g.nx - 1) + 1);
        for (int32_t k = b; k < e; ++k)
          if (k != j && sqdist_f32(sx[k], sy[k], sz[k], qx, qy, qz) < sq_radius) close = true;
      }
  }
  const unsigned long long m = __ballot(close);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(count, static_cast<unsigned long long>(__popcll(m)));
}

// dense outputs by compacted index list
__global__ __launch_bounds__(kMB) void k_mls_gather(const float *__restrict__ tmp, const int32_t *__restrict__ index,
                                                    int64_t m, float *__restrict__ xyz, float *__restrict__ normal,
                                                    float *__restrict__ curv) {
  const int64_t k = static_cast<int64_t>(blockIdx.x) * kMB + threadIdx.x;
  if (k >= m) return;
  const float *t = tmp + static_cast<int64_t>(index[k]) * kRowStride;
  xyz[3 * k + 0] = t[0]; xyz[3 * k + 1] = t[1]; xyz[3 * k + 2] = t[2];
  normal[3 * k + 0] = t[3]; normal[3 * k + 1] = t[4]; normal[3 * k + 2] = t[5];
  curv[k] = t[6];
}

// ---------------------------------------------------------------------------
// VOXEL_GRID_DILATION upsampling (pcl::MovingLeastSquares::performUpsampling +
// MLSVoxelGrid [upstream mls.hpp]; configured at PCP/src/cloudSmooth.cpp:144-147,
// values PCP/src/PointCloudProcessor.cpp:78-81).
//
// PCL keeps the voxels in a std::map keyed ix*S^2 + iy*S + iz and dilates it
// `iterations` times by the 26-neighbourhood.  Here the voxel set is a dense bitmap
// over the cloud's bounding box, bit index ((ix*NY + iy)*NZ + iz): ascending bit
// index is ascending PCL key, so the output order is the reference's.  k dilations
// of a voxel are the (2k+1)^3 cube around it (restricted to non-negative indices,
// Appendix B16), which every input point stamps directly with atomicOr.
// ---------------------------------------------------------------------------
struct VoxelDesc {
  float bminx, bminy, bminz, vs;
  int32_t NX, NY, NZ, it;
  int64_t words;  // 32-bit words of the bitmap
};

__device__ __forceinline__ void voxel_of(const VoxelDesc &v, float x, float y, float z, int32_t &ix, int32_t &iy,
                                         int32_t &iz) {
  // MLSVoxelGrid::getCellIndex: (p - bounding_min) / voxel_size in fp32, truncated
  ix = static_cast<int32_t>(__fdiv_rn(__fsub_rn(x, v.bminx), v.vs));
  iy = static_cast<int32_t>(__fdiv_rn(__fsub_rn(y, v.bminy), v.vs));
  iz = static_cast<int32_t>(__fdiv_rn(__fsub_rn(z, v.bminz), v.vs));
}

__global__ __launch_bounds__(kMB) void k_voxel_stamp(const float *__restrict__ x, const float *__restrict__ y,
                                                     const float *__restrict__ z, int64_t n, VoxelDesc v,
                                                     uint32_t *__restrict__ bitmap) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kMB + threadIdx.x;
  if (i >= n) return;
  const float px = x[i];
  if (!isfinite(px)) return;  // MLSVoxelGrid ctor skips non-finite points
  int32_t ix, iy, iz;
  voxel_of(v, px, y[i], z[i], ix, iy, iz);
  const int32_t z0 = max(iz - v.it, 0), z1 = min(iz + v.it, v.NZ - 1);
  for (int32_t cx = max(ix - v.it, 0); cx <= min(ix + v.it, v.NX - 1); ++cx)
    for (int32_t cy = max(iy - v.it, 0); cy <= min(iy + v.it, v.NY - 1); ++cy) {
      const int64_t l0 = (static_cast<int64_t>(cx) * v.NY + cy) * v.NZ + z0;
      const int64_t l1 = l0 + (z1 - z0);  // inclusive
      for (int64_t wd = l0 >> 5; wd <= (l1 >> 5); ++wd) {
        const int64_t b0 = max(l0, wd << 5) - (wd << 5), b1 = min(l1, (wd << 5) + 31) - (wd << 5);
        const uint32_t m = ((b1 - b0 + 1) >= 32 ? 0xffffffffu : ((1u << (b1 - b0 + 1)) - 1u)) << b0;
        atomicOr(bitmap + wd, m);
      }
    }
}

// set bits per tile of kScanTile bitmap words (a workgroup per tile, four words per lane; grid-stride: a bitmap of more than
// 2^32 words does not fit one launch's 32-bit work-item count).  The per-WORD counts and their prefix used to be an array
// of their own, as large as the bitmap (60 GB for the room at 1 mm) and 99 % zeros; the prefix inside a tile is now taken
// where it is needed (k_voxel_expand).
__global__ __launch_bounds__(kScanBlock) void k_voxel_tile_counts(const uint32_t *__restrict__ bitmap, int64_t words,
                                                                int32_t *__restrict__ tile_count) {
  __shared__ int32_t ws[kScanBlock / 64];
  const int64_t tiles = (words + kScanTile - 1) / kScanTile;
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t base = tile * kScanTile + threadIdx.x * 4;
    int32_t c = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (base + k < words) c += __popc(bitmap[base + k]);
    int32_t total;
    (void)scan_block_exclusive(c, &total, ws);
    if (threadIdx.x == 0) tile_count[tile] = total;
  }
}


// ---------------------------------------------------------------------------
// The voxel set as BRICKS (round 4): memory that follows the surface.  The dense bitmap above costs one bit per voxel of the
// bounding box -- 60 GB for a 12 x 10 x 4 m room at 1 mm, and a 40 x 40 x 5 m map does not fit at all, where PCL's std::map
// only grows with the occupied voxels.  Here the box is cut into bricks of 16 x 16 x 16 voxels (512 B of bits each) and only
// the bricks a stamped cube reaches exist: one occupancy bit per brick place, linear index (bx NBY + by) NBZ + bz, with the
// running popcount per 32-bit word, so that a brick's storage slot is its RANK among the occupied places (two loads and a
// popcount; no allocation atomics, and slots are numbered in the same order on every run).
//
// Inside a brick bit ((lx 16 + ly) 16 + lz): a row of 16 z-neighbours is half a word.  PCL's key order ix, iy, iz has iz
// fastest, so for a fixed ix the 16 columns iy = 16 by .. 16 by + 15 are ONE contiguous key range: the "strip" (ix, by), whose
// bits are the rows lx = ix & 15 of the bricks (ix >> 4, by, 0 .. NBZ - 1) -- consecutive storage slots.  Strips in the order
// ix NBY + by are the key order; they play the part the tiles of 1024 bitmap words play in the dense form: counted
// (k_brick_strip_counts), cut into chunks by the host, expanded into the list of occupied voxels (k_brick_expand) in
// ascending key order, and k_voxel_emit takes it from there unchanged.
// ---------------------------------------------------------------------------
constexpr int kBrick = 16;
constexpr int kBrickWords = kBrick * kBrick * kBrick / 32;  // 128
constexpr int kBrickColumnMax = 4096;  // bricks along z (k_brick_expand keeps a column's list in LDS): 65 m at 1 mm

__device__ __forceinline__ int lane_of() { return static_cast<int>(threadIdx.x & 63u); }

struct BrickDesc {
  VoxelDesc v;
  int32_t NBX, NBY, NBZ;
  const uint32_t *occ;   // one bit per brick place
  const int32_t *rank;   // occupied places in the words before this one
  uint32_t *bits;        // kBrickWords words per occupied brick, in rank order
};

__device__ __forceinline__ int64_t brick_place(const BrickDesc &B, int32_t bx, int32_t by, int32_t bz) {
  return (static_cast<int64_t>(bx) * B.NBY + by) * B.NBZ + bz;
}

__device__ __forceinline__ int32_t brick_slot(const BrickDesc &B, int64_t place) {
  const int64_t w = place >> 5;
  return B.rank[w] + __popc(B.occ[w] & ((1u << (place & 31)) - 1u));
}

// every point marks the bricks its dilated cube reaches (at most 2 x 2 x 2 for cubes of up to 17 voxels)
__global__ __launch_bounds__(kMB) void k_brick_mark(const float *__restrict__ x, const float *__restrict__ y,
                                                    const float *__restrict__ z, int64_t n, BrickDesc B, uint32_t *__restrict__ occ) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kMB + threadIdx.x;
  if (i >= n) return;
  const float px = x[i];
  if (!isfinite(px)) return;  // MLSVoxelGrid ctor skips non-finite points
  int32_t ix, iy, iz;
  voxel_of(B.v, px, y[i], z[i], ix, iy, iz);
  const int32_t bx0 = max(ix - B.v.it, 0) >> 4, bx1 = min(ix + B.v.it, B.v.NX - 1) >> 4;
  const int32_t by0 = max(iy - B.v.it, 0) >> 4, by1 = min(iy + B.v.it, B.v.NY - 1) >> 4;
  const int32_t bz0 = max(iz - B.v.it, 0) >> 4, bz1 = min(iz + B.v.it, B.v.NZ - 1) >> 4;
  for (int32_t bx = bx0; bx <= bx1; ++bx)
    for (int32_t by = by0; by <= by1; ++by)
      for (int32_t bz = bz0; bz <= bz1; ++bz) {
        const int64_t place = brick_place(B, bx, by, bz);
        const uint32_t m = 1u << (place & 31);
        if (!(occ[place >> 5] & m)) atomicOr(occ + (place >> 5), m);  // (neighbouring points reach the same bricks: read first)
      }
}

__global__ __launch_bounds__(kScanBlock) void k_brick_popc(const uint32_t *__restrict__ occ, int64_t words, int32_t *__restrict__ cnt) {
  for (int64_t w = static_cast<int64_t>(blockIdx.x) * kScanBlock + threadIdx.x; w < words; w += static_cast<int64_t>(gridDim.x) * kScanBlock)
    cnt[w] = __popc(occ[w]);
}

// the dilated cube of every point, stamped into its bricks (k_voxel_stamp on the brick storage)
__global__ __launch_bounds__(kMB) void k_brick_stamp(const float *__restrict__ x, const float *__restrict__ y,
                                                     const float *__restrict__ z, int64_t n, BrickDesc B) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kMB + threadIdx.x;
  if (i >= n) return;
  const float px = x[i];
  if (!isfinite(px)) return;
  int32_t ix, iy, iz;
  voxel_of(B.v, px, y[i], z[i], ix, iy, iz);
  const int32_t x0 = max(ix - B.v.it, 0), x1 = min(ix + B.v.it, B.v.NX - 1);
  const int32_t y0 = max(iy - B.v.it, 0), y1 = min(iy + B.v.it, B.v.NY - 1);
  const int32_t z0 = max(iz - B.v.it, 0), z1 = min(iz + B.v.it, B.v.NZ - 1);
  for (int32_t bz = z0 >> 4; bz <= (z1 >> 4); ++bz) {
    // the cube's z-run inside this layer of bricks: bits [l0, l1] of a 16-bit row
    const int32_t l0 = max(z0, bz << 4) - (bz << 4), l1 = min(z1, (bz << 4) + 15) - (bz << 4);
    const uint32_t row = ((1u << (l1 - l0 + 1)) - 1u) << l0;
    for (int32_t bx = x0 >> 4; bx <= (x1 >> 4); ++bx)
      for (int32_t by = y0 >> 4; by <= (y1 >> 4); ++by) {
        uint32_t *bits = B.bits + static_cast<int64_t>(brick_slot(B, brick_place(B, bx, by, bz))) * kBrickWords;
        const int32_t cx0 = max(x0, bx << 4), cx1 = min(x1, (bx << 4) + 15);
        const int32_t cy0 = max(y0, by << 4), cy1 = min(y1, (by << 4) + 15);
        for (int32_t cx = cx0; cx <= cx1; ++cx)
          for (int32_t cy = cy0; cy <= cy1; ++cy) {
            const int32_t r = ((cx & 15) << 4) | (cy & 15);
            const uint32_t m = row << ((r & 1) << 4);
            uint32_t *wd = bits + (r >> 1);
            if ((*wd & m) != m) atomicOr(wd, m);
          }
      }
  }
}

// the occupied bricks of brick column (bx, by): their number and the storage slot of the first (the column's places are NBZ
// consecutive bits of the occupancy bitmap, its bricks consecutive slots).  All lanes of the wavefront call it.
__device__ __forceinline__ int32_t brick_column(const BrickDesc &B, int64_t col, int32_t *slot0) {
  const int64_t b0 = col * B.NBZ, b1 = b0 + B.NBZ;  // bits [b0, b1)
  const int64_t w0 = b0 >> 5, w1 = (b1 + 31) >> 5;
  int32_t nb = 0;
  for (int64_t w = w0 + lane_of(); w < w1; w += 64) {
    uint32_t bits = B.occ[w];
    if (w == w0) bits &= ~((1u << (b0 & 31)) - 1u);
    if (w == w1 - 1 && (b1 & 31)) bits &= (1u << (b1 & 31)) - 1u;
    nb += __popc(bits);
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) nb += __shfl_xor(nb, o, 64);
  *slot0 = nb ? brick_slot(B, b0) : 0;
  return nb;
}

// set bits per strip (ix, by), and per plane ix in 64 bits: one wavefront per brick column, lane = (lx, quarter of the 16 rows)
__global__ __launch_bounds__(64) void k_brick_strip_counts(BrickDesc B, int32_t *__restrict__ strip_count,
                                                           unsigned long long *__restrict__ plane_count) {
  const int64_t cols = static_cast<int64_t>(B.NBX) * B.NBY;
  const int lane = lane_of();
  for (int64_t col = blockIdx.x; col < cols; col += gridDim.x) {
    int32_t slot0;
    const int32_t nb = brick_column(B, col, &slot0);
    if (nb == 0) continue;  // (the counts were zeroed)
    int32_t c = 0;
    for (int32_t k = 0; k < nb; ++k) {
      const uint2 wv = *reinterpret_cast<const uint2 *>(B.bits + static_cast<int64_t>(slot0 + k) * kBrickWords + lane * 2);
      c += __popc(wv.x) + __popc(wv.y);
    }
    c += __shfl_xor(c, 1, 64);
    c += __shfl_xor(c, 2, 64);  // words [lx 8, lx 8 + 8) = the 16 rows of lx: four lanes
    const int32_t bx = static_cast<int32_t>(col / B.NBY), by = static_cast<int32_t>(col % B.NBY);
    const int32_t ix = (bx << 4) + (lane >> 2);
    if ((lane & 3) == 0 && ix < B.v.NX && c > 0) {
      strip_count[static_cast<int64_t>(ix) * B.NBY + by] = c;
      atomicAdd(plane_count + ix, static_cast<unsigned long long>(c));
    }
  }
}

// the occupied voxels of strips [s0, s0 + strips) in key order: one wavefront per strip, lane = (column ly, k-th brick of the
// column), 16 z-neighbours each; strip_first = exclusive prefix of the strips' counts inside the chunk
__global__ __launch_bounds__(64) void k_brick_expand(BrickDesc B, const int32_t *__restrict__ strip_count,
                                                     const int32_t *__restrict__ strip_first, int64_t s0, int64_t strips,
                                                     int64_t *__restrict__ vox) {
  __shared__ int16_t bz_of[kBrickColumnMax];
  const int lane = lane_of();
  for (int64_t t = blockIdx.x; t < strips; t += gridDim.x) {
    const int64_t s = s0 + t;
    if (strip_count[s] == 0) continue;
    const int32_t ix = static_cast<int32_t>(s / B.NBY), by = static_cast<int32_t>(s % B.NBY);
    const int64_t col = static_cast<int64_t>(ix >> 4) * B.NBY + by;
    // the column's bricks: bz of the k-th one, in LDS
    const int64_t b0 = col * B.NBZ, b1 = b0 + B.NBZ, w0 = b0 >> 5, w1 = (b1 + 31) >> 5;
    int32_t nb = 0;
    __syncthreads();  // (one wavefront: orders the reuse of bz_of)
    for (int64_t wb = w0; wb < w1; wb += 64) {
      const int64_t w = wb + lane;
      uint32_t bits = w < w1 ? B.occ[w] : 0u;
      if (w == w0) bits &= ~((1u << (b0 & 31)) - 1u);
      if (w == w1 - 1 && (b1 & 31)) bits &= (1u << (b1 & 31)) - 1u;
      const int32_t c = __popc(bits);
      int32_t incl = c;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int32_t u = __shfl_up(incl, o, 64);
        if (lane >= o) incl += u;
      }
      int32_t at = nb + incl - c;
      while (bits) {
        const int32_t bit = __builtin_ctz(bits);
        bits &= bits - 1u;
        if (at < kBrickColumnMax) bz_of[at] = static_cast<int16_t>((w << 5) + bit - b0);
        ++at;
      }
      nb += __shfl(incl, 63, 64);
    }
    __syncthreads();
    const int32_t slot0 = brick_slot(B, b0);
    const int32_t lx = ix & 15;
    int64_t out = strip_first[t];
    const int32_t rows = 16 * nb;
    for (int32_t r0 = 0; r0 < rows; r0 += 64) {
      const int32_t r = r0 + lane;
      uint32_t bits = 0u;
      int32_t ly = 0, k = 0;
      if (r < rows) {
        ly = r / nb;
        k = r % nb;
        const int32_t rr = (lx << 4) | ly;
        bits = (B.bits[static_cast<int64_t>(slot0 + k) * kBrickWords + (rr >> 1)] >> ((rr & 1) << 4)) & 0xffffu;
      }
      const int32_t c = __popc(bits);
      int32_t incl = c;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int32_t u = __shfl_up(incl, o, 64);
        if (lane >= o) incl += u;
      }
      int64_t at = out + incl - c;
      const int64_t key0 = (static_cast<int64_t>(ix) * B.v.NY + ((by << 4) + ly)) * B.v.NZ + (static_cast<int64_t>(bz_of[k < kBrickColumnMax ? k : 0]) << 4);
      while (bits) {
        vox[at++] = key0 + __builtin_ctz(bits);
        bits &= bits - 1u;
      }
      out += __shfl(incl, 63, 64);
    }
  }
}

struct VoxelEmitArgs {
  const uint32_t *bitmap;
  VoxelDesc v;
  // neighbour grid of the MLS stage (cell-sorted coordinates)
  const float *sx, *sy, *sz;
  const int32_t *order, *start;
  const int32_t *remap;  // view index -> caller's index (nullable)
  GridDesc g;
  float dmax;           // upper bound of the distance from a voxel position to its nearest input point
  const int64_t *vox;   // occupied voxels in key order (k_voxel_expand)
  int64_t total;
  const double *state;  // kMlsState doubles per input point
  int32_t order_poly, required_neighbors;
  float *xyz, *normal, *curv;
  int32_t *index;
  uint8_t *valid;
  uint32_t *max_dx;  // nullable: max over the emitted points of |x of the point - x of its voxel position| (bits of a float >= 0)
  int32_t dry;       // 1: nothing is stored -- the launch is for max_dx alone (sweep 0 of the streamed chain); 2: positions and
                     // validity only (its sweep 1, which needs no normals, curvatures or source indices)
  int32_t block_step;  // workgroup b takes voxels [b * block_step * kMB, ... + kMB): 1 = every voxel, 8 = a sample of one in eight
};

// the occupied voxels of bitmap words [word_base, word_base + words) in key order: a workgroup per tile of kScanTile words,
// four consecutive words per lane, the prefix of the set bits inside the tile by a block scan, the tile's first place from
// the exclusive prefix of the tile counts (tile_first).  Stores only (the bitmap is ~1 % full: the search of k_voxel_emit
// must not run at this granularity).
__global__ __launch_bounds__(kScanBlock) void k_voxel_expand(const uint32_t *__restrict__ bitmap, const int32_t *__restrict__ tile_first,
                                                           int64_t words, int64_t word_base, int64_t *__restrict__ vox) {
  __shared__ int32_t ws[kScanBlock / 64];
  const int64_t tiles = (words + kScanTile - 1) / kScanTile;
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t base = tile * kScanTile + threadIdx.x * 4;
    uint32_t bits[4];
    int32_t c = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      bits[k] = base + k < words ? bitmap[base + k] : 0u;
      c += __popc(bits[k]);
    }
    int32_t total;
    int64_t out = static_cast<int64_t>(tile_first[tile]) + scan_block_exclusive(c, &total, ws);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      uint32_t b = bits[k];
      while (b) {
        vox[out++] = ((base + k + word_base) << 5) + __builtin_ctz(b);
        b &= b - 1u;
      }
    }
  }
}

// one lane per occupied voxel (consecutive lanes = consecutive keys = neighbouring voxels of one z-row, so the
// nearest-point searches of a wavefront read the same cells)
// (returns |x of the emitted point - x of its voxel position|, 0 for a voxel without output)
__device__ __forceinline__ float voxel_emit_one(const VoxelEmitArgs &a, int64_t out) {
  const int64_t L = a.vox[out];
  const int32_t iz = static_cast<int32_t>(L % a.v.NZ);
  const int64_t q = L / a.v.NZ;
  const int32_t iy = static_cast<int32_t>(q % a.v.NY), ix = static_cast<int32_t>(q / a.v.NY);
  // MLSVoxelGrid::getPosition: float(index) * voxel_size + bounding_min
  const float px = __fadd_rn(__fmul_rn(static_cast<float>(ix), a.v.vs), a.v.bminx);
  const float py = __fadd_rn(__fmul_rn(static_cast<float>(iy), a.v.vs), a.v.bminy);
  const float pz = __fadd_rn(__fmul_rn(static_cast<float>(iz), a.v.vs), a.v.bminz);
  // tree_->nearestKSearch(p, 1): closest input point (fp32 L2_Simple); ties -> lower index.  The point that
  // stamped this voxel lies within `a.dmax` of p, so the nearest one does too: only the cells meeting the
  // box p +- dmax are visited (grid_coords is monotone in each coordinate).
  int32_t x0, y0, z0, x1, y1, z1;
  grid_coords(a.g, px - a.dmax, py - a.dmax, pz - a.dmax, x0, y0, z0);
  grid_coords(a.g, px + a.dmax, py + a.dmax, pz + a.dmax, x1, y1, z1);
  int32_t best = -1;
  float bestd = FLT_MAX;
  for (int32_t zz = z0; zz <= z1; ++zz)
    for (int32_t yy = y0; yy <= y1; ++yy) {
      const int32_t s0 = cell_start(a.g, a.start, zz, yy, x0), s1 = cell_start(a.g, a.start, zz, yy, x1 + 1);
      for (int32_t k = s0; k < s1; ++k) {
        const float d = sqdist_f32(a.sx[k], a.sy[k], a.sz[k], px, py, pz);
        if (d <= bestd) {  // the index is only needed for the rare candidates that can win
          const int32_t id = a.remap ? a.remap[a.order[k]] : a.order[k];
          if (d < bestd || id < best) {
            bestd = d;
            best = id;
          }
        }
      }
    }
  bool ok = best >= 0;
  const double *st = a.state + static_cast<int64_t>(ok ? best : 0) * kMlsState;
  ok = ok && st[20] >= 1.0;  // mls_results_[input_index].valid
  float ox = 0.0f;
  if (ok) {
    // MLSResult::projectPoint(pt, SIMPLE, 5 * nr_coeff)
    const double dx = static_cast<double>(px) - st[0], dy = static_cast<double>(py) - st[1],
                 dz = static_cast<double>(pz) - st[2];
    const double u = (dx * st[6] + dy * st[7]) + dz * st[8];
    const double vv = (dx * st[9] + dy * st[10]) + dz * st[11];
    double wgt = 0.0, nx = st[3], ny = st[4], nz = st[5];
    if (a.order_poly > 1 && st[19] >= static_cast<double>(a.required_neighbors) && st[20] >= 2.0 && isfinite(st[12])) {
      // getPolynomialPartialDerivative: monomials 1, v, v^2, u, uv, u^2
      const double c0 = st[12], c1 = st[13], c2 = st[14], c3 = st[15], c4 = st[16], c5 = st[17];
      wgt = c0 + vv * c1 + (vv * vv) * c2 + u * c3 + (u * vv) * c4 + (u * u) * c5;
      if (a.dry == 0) {
        const double zu = c3 + c4 * vv + c5 * 2.0 * u;
        const double zv = c1 + c2 * 2.0 * vv + c4 * u;
        nx -= zu * st[6] + zv * st[9];
        ny -= zu * st[7] + zv * st[10];
        nz -= zu * st[8] + zv * st[11];
        const double l = sqrt((nx * nx + ny * ny) + nz * nz);
        if (l > 0.0) {
          nx /= l; ny /= l; nz /= l;
        }
      }
    }
    ox = static_cast<float>(st[0] + u * st[6] + vv * st[9] + wgt * st[3]);
    if (a.dry != 1) {
      a.xyz[3 * out + 0] = ox;
      a.xyz[3 * out + 1] = static_cast<float>(st[1] + u * st[7] + vv * st[10] + wgt * st[4]);
      a.xyz[3 * out + 2] = static_cast<float>(st[2] + u * st[8] + vv * st[11] + wgt * st[5]);
    }
    if (a.dry == 0) {
      a.normal[3 * out + 0] = static_cast<float>(nx);
      a.normal[3 * out + 1] = static_cast<float>(ny);
      a.normal[3 * out + 2] = static_cast<float>(nz);
      a.curv[out] = static_cast<float>(st[18]);
      a.index[out] = best;
    }
  }
  if (a.dry != 1) a.valid[out] = ok ? 1 : 0;
  float d = ok ? fabsf(ox - px) : 0.0f;
  if (!(d == d)) d = INFINITY;  // a NaN position: no bound
  return d;
}

__global__ __launch_bounds__(kMB) void k_voxel_emit(VoxelEmitArgs a) {
  const int64_t out = static_cast<int64_t>(blockIdx.x) * a.block_step * kMB + threadIdx.x;
  const float d = out < a.total ? voxel_emit_one(a, out) : 0.0f;
  if (a.max_dx) {
    // how far (along x, the axis the streamed chain cuts the key order by) the projection moved a point from its voxel: the
    // maximum over the launch (every lane takes part in the reduction)
    uint32_t b = __float_as_uint(d);
    for (int o = 32; o >= 1; o >>= 1) b = max(b, static_cast<uint32_t>(__shfl_xor(static_cast<int>(b), o, 64)));
    // (only a wavefront that would raise the maximum it reads goes to the atomic: four million atomics on one address were
    // 40 ms of a 53 ms launch)
    if ((threadIdx.x & 63) == 0 && b > __atomic_load_n(a.max_dx, __ATOMIC_RELAXED)) atomicMax(a.max_dx, b);
  }
}

// in-place style compaction of the voxel outputs when some voxels were dropped
__global__ __launch_bounds__(kMB) void k_voxel_compact(const int32_t *__restrict__ keep_index, int64_t m,
                                                       const float *__restrict__ xyz, const float *__restrict__ normal,
                                                       const float *__restrict__ curv, const int32_t *__restrict__ index,
                                                       float *__restrict__ oxyz, float *__restrict__ onormal,
                                                       float *__restrict__ ocurv, int32_t *__restrict__ oindex) {
  const int64_t k = static_cast<int64_t>(blockIdx.x) * kMB + threadIdx.x;
  if (k >= m) return;
  const int64_t s_ = keep_index[k];
  for (int c = 0; c < 3; ++c) oxyz[3 * k + c] = xyz[3 * s_ + c];
  if (!onormal) return;  // (positions only: sweep 1 of the streamed chain)
  for (int c = 0; c < 3; ++c) onormal[3 * k + c] = normal[3 * s_ + c];
  ocurv[k] = curv[s_];
  oindex[k] = index[s_];
}

// ---------------------------------------------------------------------------
// pcl::StatisticalOutlierRemoval (filters/impl/statistical_outlier_removal.hpp
// [upstream]; configured at PCP/src/cloudSmooth.cpp:109-116,160-164: k = 60, 0.7 sigma).
// Per point: mean distance to its mean_k nearest neighbours (the query itself is hit
// 0 of nearestKSearch(k+1) and is skipped), then keep iff distance <= mean + mul*sigma.
// One lane per point; its k+1 best squared distances live in a max-heap in LDS
// (column layout heap[e][lane]: conflict-free).  Cells are visited ring by ring until
// the heap's maximum is closer than the next ring.
// ---------------------------------------------------------------------------
constexpr int kSorBlock = 128;

// kHeapArity-ary max-heap: with 4 children per node three levels below the root hold k + 1 = 61 entries,
// and the children of a node are read with independent LDS loads -- the sift-down is a chain of dependent
// LDS round trips, so its depth (three, against a binary heap's six) is what counts (binary 35.1 ms,
// 4-ary 30.0 ms, 8-ary 30.0 ms for the two SOR passes of the C3 cloud).
constexpr int kHeapArity = 4;
__device__ __forceinline__ void heap_push(float *heap, int &size, int k, float d) {
  // heap[e * kSorBlock] : e-th slot of this lane's max-heap
  if (size < k) {
    int c = size++;
    while (c > 0) {
      const int pnt = (c - 1) / kHeapArity;
      const float pv = heap[pnt * kSorBlock];
      if (pv >= d) break;
      heap[c * kSorBlock] = pv;
      c = pnt;
    }
    heap[c * kSorBlock] = d;
    return;
  }
  if (!(d < heap[0])) return;
  int c = 0;
  for (;;) {
    const int l = kHeapArity * c + 1;
    if (l >= k) break;
    float v[kHeapArity];
#pragma unroll
    for (int i = 0; i < kHeapArity; ++i) v[i] = heap[(l + i < k ? l + i : l) * kSorBlock];  // past the end: slot l again
    int big = l;
    float bv = v[0];
#pragma unroll
    for (int i = 1; i < kHeapArity; ++i)
      if (v[i] > bv) {
        bv = v[i];
        big = l + i;
      }
    if (bv <= d) break;
    heap[c * kSorBlock] = bv;
    c = big;
  }
  heap[c * kSorBlock] = d;
}

__global__ __launch_bounds__(kSorBlock) void k_sor_mean_distance(const float *__restrict__ sx, const float *__restrict__ sy,
                                                                 const float *__restrict__ sz,
                                                                 const int32_t *__restrict__ order,
                                                                 const int32_t *__restrict__ remap,
                                                                 const int32_t *__restrict__ start, int64_t n, GridDesc g,
                                                                 int32_t mean_k, float *__restrict__ distances,
                                                                 const int32_t *__restrict__ list, int64_t list_n,
                                                                 const uint8_t *__restrict__ only_flagged,
                                                                 float *__restrict__ kth = nullptr) {
  // kth (nullable, all three distance kernels): an upper bound of the SQUARED distance to the (mean_k + 1)-th nearest point,
  // under the index `distances` uses -- how far the neighbourhood of the point reaches (the streamed chain checks it against
  // the halo of its chunks)
  extern __shared__ float sor_heap[];
  // list == nullptr: every point; else only the cell-sorted positions named by the list (k_sor_select's leftovers),
  // and of those only the ones k_sor_wave left flagged (only_flagged[j] == 2) when that array is given
  const int64_t t = static_cast<int64_t>(blockIdx.x) * kSorBlock + threadIdx.x;
  if (t >= (list ? list_n : n)) return;
  const int64_t j = list ? list[t] : t;
  if (only_flagged && only_flagged[j] != 2) return;
  float *heap = sor_heap + threadIdx.x;
  const int k = mean_k + 1;
  const float qx = sx[j], qy = sy[j], qz = sz[j];
  int32_t cx, cy, cz;
  grid_coords(g, qx, qy, qz, cx, cy, cz);
  const float cell = 1.0f / g.inv_cell;
  const int32_t maxr = max(g.nx, max(g.ny, g.nz));
  int size = 0;
  // candidates of one run of cells, four at a time (12 coordinate loads in flight)
  auto scan = [&](int32_t b, int32_t e) {
    for (int32_t q = b; q < e; q += 4) {
      const int32_t q1 = min(q + 1, e - 1), q2 = min(q + 2, e - 1), q3 = min(q + 3, e - 1);
      const float d0 = sqdist_f32(sx[q], sy[q], sz[q], qx, qy, qz);
      const float d1 = sqdist_f32(sx[q1], sy[q1], sz[q1], qx, qy, qz);
      const float d2 = sqdist_f32(sx[q2], sy[q2], sz[q2], qx, qy, qz);
      const float d3 = sqdist_f32(sx[q3], sy[q3], sz[q3], qx, qy, qz);
      heap_push(heap, size, k, d0);
      if (q + 1 < e) heap_push(heap, size, k, d1);
      if (q + 2 < e) heap_push(heap, size, k, d2);
      if (q + 3 < e) heap_push(heap, size, k, d3);
    }
  };
  for (int32_t ring = 0; ring <= maxr; ++ring) {
    if (size == k && ring >= 1) {
      // every unvisited point is at least (ring - 1) cells away; 0.999 absorbs the fp32 cell-assignment slop
      const float reach = static_cast<float>(ring - 1) * cell * 0.999f;
      if (reach * reach > heap[0]) break;
    }
    if (ring > 1 && static_cast<int64_t>(2 * ring + 1) * (2 * ring + 1) > n / 32) {
      // a stray point far from the rest of the cloud: the shells of empty rows cost more than every point once
      size = 0;
      scan(0, static_cast<int32_t>(n));
      break;
    }
    for (int32_t zz = max(cz - ring, 0); zz <= min(cz + ring, g.nz - 1); ++zz)
      for (int32_t yy = max(cy - ring, 0); yy <= min(cy + ring, g.ny - 1); ++yy) {
        const bool shell_yz = zz == cz - ring || zz == cz + ring || yy == cy - ring || yy == cy + ring;
        if (shell_yz) {
          scan(cell_start(g, start, zz, yy, max(cx - ring, 0)), cell_start(g, start, zz, yy, min(cx + ring, g.nx - 1) + 1));
        } else {
          if (cx - ring >= 0) scan(cell_start(g, start, zz, yy, cx - ring), cell_start(g, start, zz, yy, cx - ring + 1));
          if (cx + ring < g.nx && ring > 0) scan(cell_start(g, start, zz, yy, cx + ring), cell_start(g, start, zz, yy, cx + ring + 1));
        }
      }
  }
  // sum of sqrt over the hits except hit 0 (the smallest, the query itself)
  double sum = 0.0;
  float smallest = FLT_MAX;
  for (int e = 0; e < size; ++e) {
    const float d = heap[e * kSorBlock];
    sum += static_cast<double>(sqrtf(d));
    smallest = fminf(smallest, d);
  }
  if (size > 0) sum -= static_cast<double>(sqrtf(smallest));
  distances[remap ? remap[order[j]] : order[j]] = static_cast<float>(sum / static_cast<double>(mean_k));
  if (kth) kth[remap ? remap[order[j]] : order[j]] = size == k ? heap[0] : INFINITY;  // the heap's root: the largest of the k + 1
}

// Selection without a heap (the common case), one wavefront per 64 consecutive cell-sorted points.
//
// All points closer than `limit` (= 0.999 grid cells) lie in the 3x3x3 cells around the query, and the cell size is
// chosen so that this ball holds ~1.5 (k + 1) points on average.   A histogram pass counts the candidates inside the
// ball in 32 equal-width bins of the squared distance (16-bit counters in LDS, [bin][lane]; a 33rd row takes the
// candidates outside the ball, so the pass has no test and no branch per candidate); the bin in which the count reaches
// k + 1 is the boundary bin.  Where the cloud is denser than average that bin is crowded: it is then histogrammed
// again into 32 sub-bins (up to two refinements), every level classifying a candidate with the same arithmetic.  The
// last pass appends every candidate up to and including the boundary bin to a short per-lane list in LDS; whenever a
// list is nearly full the wavefront drains the lists in a dense loop (sqrt and the fp64 sum of the entries below the
// boundary with every lane busy; the boundary bin's members, at most kSelList, move to a second list), and at the
// end the smallest missing members are picked.  A lane whose ball holds fewer than k + 1 points, or whose boundary
// stays crowded (many equal distances), is flagged and redone by the heap kernel: the result is always the exact sum
// over the k + 1 nearest, minus the nearest (the query itself).  (The fp64 additions are exact -- 24-bit significands
// within a few binades, < 2^8 terms -- so the order of the terms does not matter.)
//
// Each lane walks the 9 runs of its own 27 cells, four candidates per trip: three 16-byte buffer loads (one per
// coordinate plane; the descriptor keeps the base in SGPRs), packed fp32 arithmetic on two candidates at a time (each
// component rounded as the scalar form: (dx dx + dy dy) + dz dz, no contraction).  A run's ragged end is one more trip
// whose missing candidates sit at infinity (outside every ball).  ~9 VALU instructions per candidate and pass; the
// first version (one dword load per coordinate, scalar arithmetic, a test and a branch per candidate, sqrt inside the
// candidate loop) issued 25.
// Two shapes of the selection (template arguments kBins, kList of k_sor_select):
//   32 bins, 16 members -- scanned clouds: a boundary bin holds ~3 of the ball's ~90 candidates, the lists are 6.4 KB of LDS per
//     wavefront (4.6 wavefronts per SIMD resident);
//   64 bins, 32 members -- the upsampled clouds of VOXEL_GRID_DILATION (round 5): up to nine voxels of a column project onto
//     nearly the same surface point, so the distances to a query come in clusters; with 32 bins most wavefronts held a lane
//     whose boundary bin overflowed 16 members and walked their candidates a third time (refinement).  Measured on the
//     425 M-row cloud of the reference's chain at 1 M input points (profiles/vgd_sor_probe.py, PCP_SEL_BINS / PCP_SEL_LIST
//     builds): 32/16 781 ms of outlier removal, 32/32 689, 64/16 753, 64/32 566, 96/40 569, 64/48 650, 128/32 723, 128/64 760;
//     on the scanned 10 M-point map 64/32 costs 9.0 ms against 7.3.
constexpr int kSelDrain = 9;    // entries of the per-lane list of the last pass
constexpr int kSelLevels = 3;
constexpr int kSelWave = 64;
typedef float sel_v2f __attribute__((ext_vector_type(2)));
typedef float sel_v4f __attribute__((ext_vector_type(4)));
typedef uint32_t sel_v4u __attribute__((ext_vector_type(4)));

#ifndef PCP_SEL_WPE
#define PCP_SEL_WPE 6
#endif
// kOneDescriptor: the three coordinate planes end below 2^32 bytes (n below ~ 3.5e8) and are read through one buffer descriptor;
// else (up to 2^30 points: the dilated clouds of VOXEL_GRID_DILATION) through one descriptor per plane.
template <bool kOneDescriptor, int kSelBins /* + 1 row for candidates outside the ball */, int kSelList /* members of the boundary bin a lane can pick from */>
__global__ __launch_bounds__(kSelWave) __attribute__((amdgpu_waves_per_eu(PCP_SEL_WPE, 8))) void k_sor_select(const float *__restrict__ sx, const float *__restrict__ sy,
                                                         const float *__restrict__ sz,
                                                         const int32_t *__restrict__ order,
                                                         const int32_t *__restrict__ remap,
                                                         const int32_t *__restrict__ start, int64_t n, GridDesc g,
                                                         int32_t mean_k, float *__restrict__ distances,
                                                         uint8_t *__restrict__ redo, int64_t j_begin, int64_t j_end,
                                                         float *__restrict__ kth, int32_t row_begin, int32_t row_end) {
  // [row_begin, row_end) (the shape for upsampled clouds only): the caller's indices whose distances are wanted -- the streamed
  // chain computes a chunk's own rows against chunk + halo; the halo's rows are candidates, not queries
  // [j_begin, j_end): the slab of the cell order this launch covers (the whole cloud, or one GPU's share: pcp_sor_partial)
#pragma clang fp contract(off)
  // the bins ((kSelBins + 1) x 32 words) share the lists' space
  constexpr int kSelLdsWords = (kSelDrain + kSelList) * kSelWave > (kSelBins + 1) * 32 ? (kSelDrain + kSelList) * kSelWave : (kSelBins + 1) * 32;
  __shared__ uint32_t sel_lds[kSelLdsWords];
  float *list = reinterpret_cast<float *>(sel_lds);            // [kSelDrain][64]
  float *members = list + kSelDrain * kSelWave;                // [kSelList][64]
  const int lane = threadIdx.x;
  const int64_t t = j_begin + static_cast<int64_t>(blockIdx.x) * kSelWave + lane;
  const bool live = t < j_end;
  const int64_t j = min(t, j_end - 1);  // lanes past the end shadow the last point and write nothing
  const int k = mean_k + 1;
  const float qx = sx[j], qy = sy[j], qz = sz[j];
  int32_t cx, cy, cz;
  grid_coords(g, qx, qy, qz, cx, cy, cz);
  const float cell = 1.0f / g.inv_cell;
  // every point closer than g.reach cells lies in the (2 reach + 1)^3 cells; 0.999: fp32 slop of the cell assignment.
  // (reach = 1 unless PCP_SOR_REACH says otherwise: see sor_run)
  const int32_t sr = g.reach;
  const float limit = static_cast<float>(sr) * cell * 0.999f, limit2 = limit * limit;
  const int32_t x0 = max(cx - sr, 0), x1 = min(cx + sr, g.nx - 1);
  const int32_t y0 = max(cy - sr, 0), y1 = min(cy + sr, g.ny - 1);
  const int32_t z0 = max(cz - sr, 0), z1 = min(cz + sr, g.nz - 1);
  // level l splits [lo[l], lo[l] + 32 / sc[l]) into 32 bins; bnd[l] = its boundary bin (levels > `level` unused)
  float lo[kSelLevels] = {0.0f, 0.0f, 0.0f}, sc[kSelLevels] = {static_cast<float>(kSelBins) / limit2, 0.0f, 0.0f};
  int bnd[kSelLevels] = {0, 0, 0};
  const float sc0 = sc[0];
  // -1 below the boundary, +1 above it (or outside the ball), 0 in the boundary bin of the deepest level so far;
  // `deepest` returns that level's bin for the histogram pass
  auto classify = [&](float d, int levels, int &deepest) -> int {
    const float t0 = d * sc0;
    if (!(t0 < static_cast<float>(kSelBins))) return 1;
#pragma unroll
    for (int l = 0; l < kSelLevels; ++l) {  // static indices: lo / sc / bnd stay in registers
      if (l >= levels) break;
      const int b = l == 0 ? static_cast<int>(t0) : min(max(static_cast<int>((d - lo[l]) * sc[l]), 0), kSelBins - 1);
      if (l == levels - 1) {
        deepest = b;
        return 0;
      }
      if (b < bnd[l]) return -1;
      if (b > bnd[l]) return 1;
    }
    return 0;
  };
  // The candidate coordinates come through buffer descriptors (one per coordinate plane: uniform base in SGPRs, 32-bit
  // byte offset per lane; a 16-byte load needs dword alignment only).  The planes hold n < 2^30 floats.
  // ONE descriptor over the three planes (x first, z last: one allocation, its end below 2^32 bytes), the y and z planes through
  // the instruction's scalar offset, which the range check includes: three descriptors did not stay in the scalar registers, and
  // half of each was rebuilt in front of every load.  A load that runs past the end of the x or y plane reads the start of the
  // next one, past the z plane zeros: such candidates are masked by their x (the ragged end below).
  const uint32_t plane_bytes = static_cast<uint32_t>(n) * 4u;
  const uint32_t ry = kOneDescriptor ? static_cast<uint32_t>(reinterpret_cast<const char *>(sy) - reinterpret_cast<const char *>(sx)) : 1u;
  const uint32_t rz = kOneDescriptor ? static_cast<uint32_t>(reinterpret_cast<const char *>(sz) - reinterpret_cast<const char *>(sx)) : 2u;
  const auto rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(sx), 0, kOneDescriptor ? rz + plane_bytes : plane_bytes, 0x00020000);
  const auto dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(sy), 0, plane_bytes, 0x00020000);  // (!kOneDescriptor)
  const auto dz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(sz), 0, plane_bytes, 0x00020000);
  const sel_v2f qx2 = {qx, qx}, qy2 = {qy, qy}, qz2 = {qz, qz};
  auto d2_of = [&](sel_v2f px, sel_v2f py, sel_v2f pz) {
    const sel_v2f dx = px - qx2, dy = py - qy2, dz = pz - qz2;
    return (dx * dx + dy * dy) + dz * dz;
  };
  // plane: 0u (x), ry, rz
  auto load4 = [&](uint32_t plane, uint32_t q) {
    sel_v4u v;
    if constexpr (kOneDescriptor)
      v = __builtin_amdgcn_raw_buffer_load_b128(rx, q * 4u, plane, 0);
    else
      v = plane == 0u ? __builtin_amdgcn_raw_buffer_load_b128(rx, q * 4u, 0, 0)
                      : (plane == 1u ? __builtin_amdgcn_raw_buffer_load_b128(dy, q * 4u, 0, 0)
                                     : __builtin_amdgcn_raw_buffer_load_b128(dz, q * 4u, 0, 0));
    return sel_v4f{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
  };
  // squared distances of this lane's candidates, four per call: four(d of candidates 0 1, d of candidates 2 3)
  auto for_candidates = [&](bool active, auto &&four) {
    if (!active) return;
    for (int32_t zz = z0; zz <= z1; ++zz)
      for (int32_t yy = y0; yy <= y1; ++yy) {
        uint32_t q = static_cast<uint32_t>(cell_start(g, start, zz, yy, x0));
        const uint32_t e = static_cast<uint32_t>(cell_start(g, start, zz, yy, x1 + 1));
        if (q + 4 <= e) {
          // the next four candidates are on their way while these four are tested
          sel_v4f X = load4(0u, q), Y = load4(ry, q), Z = load4(rz, q);
          for (;;) {
            q += 4;
            const bool more = q + 4 <= e;
            sel_v4f Xn = X, Yn = Y, Zn = Z;
            if (more) {
              Xn = load4(0u, q);
              Yn = load4(ry, q);
              Zn = load4(rz, q);
            }
            four(d2_of(sel_v2f{X[0], X[1]}, sel_v2f{Y[0], Y[1]}, sel_v2f{Z[0], Z[1]}),
                 d2_of(sel_v2f{X[2], X[3]}, sel_v2f{Y[2], Y[3]}, sel_v2f{Z[2], Z[3]}));
            if (!more) break;
            X = Xn;
            Y = Yn;
            Z = Zn;
          }
        }
        if (q < e) {  // 1..3 candidates left: the others at infinity (loads past the planes' end read zeros)
          const uint32_t rem = e - q;
          const sel_v4f X = load4(0u, q), Y = load4(ry, q), Z = load4(rz, q);
          four(d2_of(sel_v2f{X[0], rem > 1 ? X[1] : INFINITY}, sel_v2f{Y[0], Y[1]}, sel_v2f{Z[0], Z[1]}),
               d2_of(sel_v2f{rem > 2 ? X[2] : INFINITY, INFINITY}, sel_v2f{Y[2], Y[3]}, sel_v2f{Z[2], Z[3]}));
        }
      }
  };
  // 16-bit counters: lanes l and l + 32 share a dword (a row of 32 dwords: conflict-free whatever the bins)
  const uint32_t bump_by = lane < 32 ? 1u : 0x10000u, bump_at = static_cast<uint32_t>(lane & 31) * 4u;
  auto bump = [&](uint32_t b) {  // one no-return LDS atomic
    atomicAdd(reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(sel_lds) + b * 128u + bump_at), bump_by);
  };
  auto clear_bins = [&]() {
    for (int w = lane; w < (kSelBins + 1) * 32; w += kSelWave) sel_lds[w] = 0;
  };
  auto bin_count = [&](int b) -> int {
    const uint32_t v = sel_lds[b * 32 + (lane & 31)];
    return static_cast<int>(lane < 32 ? (v & 0xffffu) : (v >> 16));
  };
  bool bad = false, sparse = false, skip = false;
  {
    // a counter cannot wrap (and carry into the lane it shares a dword with): the 27 cells hold fewer candidates than
    // it can count, else the heap kernel takes the lane
    uint32_t total = 0;
    for (int32_t zz = z0; zz <= z1; ++zz)
      for (int32_t yy = y0; yy <= y1; ++yy) {
        total += static_cast<uint32_t>(cell_start(g, start, zz, yy, x1 + 1) - cell_start(g, start, zz, yy, x0));
      }
    bad = total > 60000u;
    if constexpr (kSelBins == 64) {
      const int32_t at_q = remap ? remap[order[j]] : order[j];
      skip = at_q < row_begin || at_q >= row_end;
      bad = bad || skip;
    }
  }
  int below = 0, level = 0, crowd = 0;
  // level 0 with the plain bin arithmetic (on a uniform cloud no lane needs more)
  {
    clear_bins();
    const sel_v2f sc2 = {sc0, sc0}, top2 = {static_cast<float>(kSelBins), static_cast<float>(kSelBins)};
    for_candidates(!bad, [&](sel_v2f da, sel_v2f db) {
      const sel_v2f ta = __builtin_elementwise_min(da * sc2, top2), tb = __builtin_elementwise_min(db * sc2, top2);  // NaN -> 32
      bump(static_cast<uint32_t>(ta[0]));
      bump(static_cast<uint32_t>(ta[1]));
      bump(static_cast<uint32_t>(tb[0]));
      bump(static_cast<uint32_t>(tb[1]));
    });
    int boundary = -1;
    for (int b = 0; b < kSelBins; ++b) {
      const int c = bin_count(b);
      if (boundary < 0) {
        if (below + c >= k) {
          boundary = b;
          crowd = c;
        } else {
          below += c;
        }
      }
    }
    if (boundary < 0 && !bad) sparse = bad = true;  // fewer than k + 1 points inside the ball
    bnd[0] = max(boundary, 0);
  }
  // refinements, only in wavefronts that hold a lane with a crowded boundary bin
  while (__any(!bad && crowd > kSelList)) {
    const bool mine = !bad && crowd > kSelList;
    if (mine && level + 1 == kSelLevels) bad = true;  // still crowded after the refinements (many equal distances)
    const bool go = mine && !bad;
    if (go) {
#pragma unroll
      for (int l = 0; l + 1 < kSelLevels; ++l)
        if (l == level) {
          lo[l + 1] = lo[l] + static_cast<float>(bnd[l]) / sc[l];
          sc[l + 1] = sc[l] * static_cast<float>(kSelBins);
        }
      ++level;
    }
    clear_bins();
    auto one = [&](float d) {
      int bb = 0;
      if (classify(d, level + 1, bb) == 0) bump(static_cast<uint32_t>(bb));
    };
    for_candidates(go, [&](sel_v2f da, sel_v2f db) {
      one(da[0]);
      one(da[1]);
      one(db[0]);
      one(db[1]);
    });
    if (go) {
      int boundary = -1;
      for (int b = 0; b < kSelBins; ++b) {
        const int c = bin_count(b);
        if (boundary < 0) {
          if (below + c >= k) {
            boundary = b;
            crowd = c;
          } else {
            below += c;
          }
        }
      }
#pragma unroll
      for (int l = 0; l < kSelLevels; ++l)
        if (l == level) bnd[l] = boundary;
    }
  }
  int last_bnd = 0;
#pragma unroll
  for (int l = 0; l < kSelLevels; ++l)
    if (l == level) last_bnd = bnd[l];
  // last pass: candidates up to and including the boundary bin go to the list; drained when a lane's list fills up
  double sum = 0.0;
  float smallest = FLT_MAX;
  int listed = 0, picked = 0;
  const bool plain = !__any(level > 0);  // wave-uniform
  const float below_t = static_cast<float>(bnd[0]);  // plain: bin < bnd[0]  <=>  d sc0 < bnd[0]
  // correctly rounded sqrt of d == 0 or d >= 2^-96: the hardware estimate (1 ulp) moved to the neighbour the exact
  // residuals ask for -- sqrtf() without its rescaling of tiny arguments and its class test (9 instructions, not 22);
  // a lane that meets 0 < d < 2^-96 goes to the heap kernel, which calls sqrtf()
  bool tiny = false;
  auto sqrt_rn = [&](float d) {
    tiny = tiny || (__float_as_uint(d) - 1u) < 0x0F7FFFFFu;
    float r = __builtin_amdgcn_sqrtf(d);
    const float r_dn = __uint_as_float(__float_as_uint(r) - 1u), r_up = __uint_as_float(__float_as_uint(r) + 1u);
    const float e_dn = __fmaf_rn(-r_dn, r, d), e_up = __fmaf_rn(-r_up, r, d);
    r = e_dn <= 0.0f ? r_dn : r;
    r = e_up > 0.0f ? r_up : r;
    return r;
  };
  auto drain = [&]() {
    for (int e = 0; e < listed; ++e) {
      const float d = list[e * kSelWave + lane];
      bool is_below;
      if (plain) {
        is_below = d * sc0 < below_t;
      } else {
        int bb = 0;
        const int c = classify(d, level + 1, bb);
        is_below = c < 0 || (c == 0 && bb < last_bnd);
      }
      if (is_below) {
        sum += static_cast<double>(sqrt_rn(d));
        smallest = fminf(smallest, d);
      } else {
        if (picked < kSelList) members[picked * kSelWave + lane] = d;
        ++picked;
      }
    }
    listed = 0;
  };
  auto append = [&](float d) {
    list[listed * kSelWave + lane] = d;
    ++listed;
  };
  if (plain) {
    const sel_v2f sc2 = {sc0, sc0};
    const float keep_below = static_cast<float>(bnd[0] + 1);  // bin <= bnd[0]  <=>  d sc0 < bnd[0] + 1
    for_candidates(!bad, [&](sel_v2f da, sel_v2f db) {
      const sel_v2f ta = da * sc2, tb = db * sc2;
      if (ta[0] < keep_below) append(da[0]);
      if (ta[1] < keep_below) append(da[1]);
      if (tb[0] < keep_below) append(db[0]);
      if (tb[1] < keep_below) append(db[1]);
      if (__any(listed > kSelDrain - 4)) drain();
    });
  } else {
    auto one = [&](float d) {
      int bb = 0;
      int c = classify(d, level + 1, bb);
      if (c == 0 && bb > last_bnd) c = 1;
      if (c <= 0) append(d);
    };
    for_candidates(!bad, [&](sel_v2f da, sel_v2f db) {
      one(da[0]);
      one(da[1]);
      one(db[0]);
      one(db[1]);
      if (__any(listed > kSelDrain - 4)) drain();
    });
  }
  drain();
  if (picked > kSelList) bad = true;  // cannot happen (the passes classify alike); never trust a list that overflowed
  if (!bad) {  // (`tiny` is settled inside)
    // the k - below smallest members of the boundary bin (picked == crowd <= kSelList)
    for (int need = k - below; need > 0; --need) {
      int at = 0;
      float best = members[lane];
      for (int e = 1; e < picked; ++e) {
        const float v = members[e * kSelWave + lane];
        if (v < best) {
          best = v;
          at = e;
        }
      }
      sum += static_cast<double>(sqrt_rn(best));
      smallest = fminf(smallest, best);
      members[at * kSelWave + lane] = FLT_MAX;
    }
    sum -= static_cast<double>(sqrt_rn(smallest));  // hit 0 of nearestKSearch(k + 1) is the query itself
    if (live && !tiny) {
      const int32_t at_i = remap ? remap[order[j]] : order[j];
      distances[at_i] = static_cast<float>(sum / static_cast<double>(mean_k));
      if (kth) kth[at_i] = limit2;  // the k + 1 nearest were found inside the ball of one cell
    }
  }
  // 1: the ball of one cell holds too few points (k_sor_wave starts with two cells); 3: any other reason
  if (live) redo[j] = skip ? 0 : (sparse ? 1 : ((bad || tiny) ? 3 : 0));
}

// The lanes k_sor_select flags (sparse spots and borders of a surface: fewer than k + 1 points within one cell; dense
// spots: a crowded boundary bin), one WAVEFRONT per point.  The heap kernel gives such a point one lane, and its
// ring-by-ring walk with a dependent LDS heap update per candidate makes a wavefront as slow as its slowest lane
// (1.1 ms for 0.35 % of a 10 M-point cloud).  Here the 64 lanes share the point's candidates:
//   1. a block of (2R + 1)^3 cells, R = 1, 2, 3, 4, 6, 9 ...: lane r fetches the bounds of row r, the rows that hold
//      points are walked eight at a time (lane i tests candidate b + i: coalesced loads, 24 in flight), and the
//      squared distances below (0.999 R cells)^2 -- every point that close lies inside the block -- are packed into
//      LDS.  Fewer than k + 1 of them: next R.
//   2. the (k + 1)-th smallest of the cached values bit by bit (non-negative floats order as unsigned integers): per
//      bit one ballot count over the entries that still match the prefix; it stops as soon as the entries still in
//      play are exactly the ones missing.
//   3. sqrt of the entries below the prefix (+ the ones in play, or the missing multiple of the one remaining value),
//      exact fp64 wave sum, minus the nearest (the point itself).
// A point whose block holds more values than the cache stays flagged (2) for the heap kernel.  A stray point -- its block
// grows until walking the rows costs more than reading the whole cloud -- takes every point as a candidate instead.
constexpr int kWsCap = 1024;
constexpr int kWsRows = 8;
__global__ __launch_bounds__(kSelWave) void k_sor_wave(const float *__restrict__ sx, const float *__restrict__ sy,
                                                       const float *__restrict__ sz, const int32_t *__restrict__ order,
                                                       const int32_t *__restrict__ remap,
                                                       const int32_t *__restrict__ start, int64_t n, GridDesc g,
                                                       int32_t mean_k, float *__restrict__ distances,
                                                       uint8_t *__restrict__ redo, const int32_t *__restrict__ list,
                                                       int64_t list_n, float *__restrict__ kth,
                                                       unsigned long long *__restrict__ tally /* nullable: PCP_SOR_WAVE_STATS */) {
#pragma clang fp contract(off)
  __shared__ float cache[kWsCap];
  const int lane = threadIdx.x;
  if (static_cast<int64_t>(blockIdx.x) >= list_n) return;
  const unsigned long long t_start = tally ? wall_clock64() : 0ull;
  int32_t passes = 0;
  auto tally_up = [&](bool overflow, bool was_far) {  // per number of block passes: wavefronts, 100 MHz ticks
    if (tally && lane == 0) {
      const int slot = min(passes, 15) + (was_far ? 16 : 0) + (overflow ? 32 : 0);
      atomicAdd(&tally[2 * slot], 1ull);
      atomicAdd(&tally[2 * slot + 1], wall_clock64() - t_start);
    }
  };
  const int32_t j = list[blockIdx.x];
  const int k = mean_k + 1;
  const float qx = sx[j], qy = sy[j], qz = sz[j];
  int32_t cx, cy, cz;
  grid_coords(g, qx, qy, qz, cx, cy, cz);
  const float cell = 1.0f / g.inv_cell;
  const int32_t maxr = max(g.nx, max(g.ny, g.nz));
  const unsigned long long lanes_below = (1ull << lane) - 1ull;
  int32_t M = 0;
  float T0 = 0.0f;   // squared distance below which the current pass caches a candidate
  bool far = false;  // the block of cells has grown past the point where walking its rows beats reading every point
  auto take = [&](float d) {  // d = +inf for lanes without a candidate
    const bool in = d < T0;
    const unsigned long long m = __ballot(in);
    const int32_t at = M + static_cast<int32_t>(__popcll(m & lanes_below));
    if (in && at < kWsCap) cache[at] = d;
    M += static_cast<int32_t>(__popcll(m));
  };
  // (a point k_sor_select found too few neighbours for within one cell starts with two)
  // (in units of the selection's own ball, g.reach cells: a sparse spot starts with two of them)
  for (int32_t R = redo[j] == 1 ? 2 * g.reach : g.reach;; R = R < 4 * g.reach ? R + g.reach : R + R / 2) {
    const bool whole = R >= maxr;  // the block is the grid: every point is a candidate
    const float lim = static_cast<float>(R) * cell * 0.999f;  // 0.999: fp32 slop of the cell assignment
    T0 = whole ? INFINITY : lim * lim;
    if (!whole && static_cast<int64_t>(2 * R + 1) * (2 * R + 1) > n / 32) {  // a stray point far from the rest of the cloud
      far = true;
      break;
    }
    const int32_t x0 = max(cx - R, 0), x1 = min(cx + R, g.nx - 1);
    const int32_t y0 = max(cy - R, 0), y1 = min(cy + R, g.ny - 1);
    const int32_t z0 = max(cz - R, 0), z1 = min(cz + R, g.nz - 1);
    const int32_t wy = y1 - y0 + 1, nrows = wy * (z1 - z0 + 1);
    M = 0;
    passes += 1;
    for (int32_t r0 = 0; r0 < nrows; r0 += kSelWave) {
      const int32_t r = r0 + lane;
      int32_t b = 0, len = 0;
      if (r < nrows) {
        const int32_t zz = z0 + r / wy, yy = y0 + r % wy;
        b = cell_start(g, start, zz, yy, x0);
        len = cell_start(g, start, zz, yy, x1 + 1) - b;
      }
      unsigned long long todo = __ballot(len > 0);
      while (todo) {
        int32_t rb[kWsRows], rl[kWsRows];
#pragma unroll
        for (int u = 0; u < kWsRows; ++u) {
          rb[u] = 0;
          rl[u] = 0;
          if (todo) {
            const int src = __ffsll(todo) - 1;
            todo &= todo - 1;
            rb[u] = __builtin_amdgcn_readlane(b, src);
            rl[u] = __builtin_amdgcn_readlane(len, src);
          }
        }
        float d[kWsRows];
#pragma unroll
        for (int u = 0; u < kWsRows; ++u) {
          d[u] = INFINITY;
          if (lane < rl[u]) d[u] = sqdist_f32(sx[rb[u] + lane], sy[rb[u] + lane], sz[rb[u] + lane], qx, qy, qz);
        }
#pragma unroll
        for (int u = 0; u < kWsRows; ++u)
          if (rl[u] > 0) take(d[u]);
#pragma unroll
        for (int u = 0; u < kWsRows; ++u)  // long rows: the rest, 64 at a time
          for (int32_t off = kSelWave; off < rl[u]; off += kSelWave) {
            const int32_t q = rb[u] + off + lane;
            take(off + lane < rl[u] ? sqdist_f32(sx[q], sy[q], sz[q], qx, qy, qz) : INFINITY);
          }
      }
    }
    if (M >= k || whole) break;
  }
  if (far) {
    // Every point is a candidate (a linear, coalesced sweep of the cell-sorted planes: cheaper than visiting millions of
    // empty rows one pair of table entries at a time).  The radius of the last block is doubled until at least k points lie
    // within it; should that hold more values than the cache, the radius is bisected down.
    auto count_within = [&](float t) {
      int32_t c = 0;
      for (int64_t i0 = 0; i0 < n; i0 += 4 * kSelWave) {
        float d[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int64_t i = i0 + u * kSelWave + lane;
          d[u] = i < n ? sqdist_f32(sx[i], sy[i], sz[i], qx, qy, qz) : INFINITY;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) c += d[u] < t ? 1 : 0;
      }
      for (int o = 32; o >= 1; o >>= 1) c += __shfl_xor(c, o, 64);
      return c;
    };
    float lo = 0.0f;  // fewer than k points within lo
    int32_t c = count_within(T0);
    while (c < k && T0 < INFINITY) {
      lo = T0;
      T0 = T0 * 4.0f;  // overflows to +inf: every point
      c = count_within(T0);
    }
    for (int it = 0; it < 40 && c > kWsCap; ++it) {
      const float mid = T0 < INFINITY ? lo + (T0 - lo) * 0.5f : fmaxf(lo * 4.0f, 1.0f);
      if (!(mid > lo && mid < T0)) break;  // no float in between: many equal distances (the heap kernel's case)
      const int32_t cm = count_within(mid);
      if (cm < k) {
        lo = mid;
      } else {
        T0 = mid;
        c = cm;
      }
    }
    M = 0;
    for (int64_t i0 = 0; i0 < n; i0 += kSelWave) {
      const int64_t i = i0 + lane;
      take(i < n ? sqdist_f32(sx[i], sy[i], sz[i], qx, qy, qz) : INFINITY);
    }
  }
  if (M > kWsCap) {  // uniform
    if (lane == 0) redo[j] = 2;
    tally_up(true, far);
    return;
  }
  // the k-th smallest cached value (1-based; the query itself is among them), bit by bit from the top
  uint32_t prefix = 0, decided = 0;  // bits of the value fixed so far, and which bits those are
  __builtin_amdgcn_wave_barrier();  // the cache is complete: no LDS access moves across this point
  int32_t need = min(k, M), in_play = M;  // fewer than k values (a cloud of < k points): all of them
  auto decide = [&](int bit, int32_t zeros) {  // zeros: entries matching the prefix whose `bit` is 0
    if (need <= zeros) {
      in_play = zeros;
    } else {
      prefix |= 1u << bit;
      need -= zeros;
      in_play -= zeros;
    }
    decided |= 1u << bit;
  };
  if (M > k && M <= 4 * kSelWave) {
    // at most four entries per lane, kept in registers (finite values never match all-ones)
    uint32_t u[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) u[i] = i * kSelWave + lane < M ? __float_as_uint(cache[i * kSelWave + lane]) : 0xffffffffu;
    for (int bit = 30; bit >= 0 && in_play != need; --bit) {
      const uint32_t probe = decided | (1u << bit);
      int32_t zeros = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) zeros += static_cast<int32_t>(__popcll(__ballot((u[i] & probe) == prefix)));
      decide(bit, zeros);
    }
  } else if (M > k && M <= 8 * kSelWave) {
    // a sparse spot's ball of two cells holds 250-400 values: eight per lane, still in registers (the loop over the
    // cache below re-reads every value for every bit)
    uint32_t u[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) u[i] = i * kSelWave + lane < M ? __float_as_uint(cache[i * kSelWave + lane]) : 0xffffffffu;
    for (int bit = 30; bit >= 0 && in_play != need; --bit) {
      const uint32_t probe = decided | (1u << bit);
      int32_t zeros = 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) zeros += static_cast<int32_t>(__popcll(__ballot((u[i] & probe) == prefix)));
      decide(bit, zeros);
    }
  } else if (M > k) {
    for (int bit = 30; bit >= 0 && in_play != need; --bit) {
      const uint32_t probe = decided | (1u << bit);
      int32_t zeros = 0;
      for (int32_t e0 = 0; e0 < M; e0 += kSelWave) {
        const int32_t e = e0 + lane;
        const uint32_t u = e < M ? __float_as_uint(cache[e]) : 0xffffffffu;
        zeros += static_cast<int32_t>(__popcll(__ballot((u & probe) == prefix)));
      }
      decide(bit, zeros);
    }
  }
  // entries below the prefix are among the k nearest; of the entries in play, all (in_play == need), or `need` copies
  // of the one value they share (every bit decided)
  const bool all_in_play = in_play == need;
  double sum = 0.0;
  float smallest = FLT_MAX;
  for (int32_t e0 = 0; e0 < M; e0 += kSelWave) {
    const int32_t e = e0 + lane;
    if (e < M) {
      const float d = cache[e];
      const uint32_t u = __float_as_uint(d) & decided;
      if (u < prefix || (u == prefix && all_in_play)) sum += static_cast<double>(sqrtf(d));
      smallest = fminf(smallest, d);
    }
  }
  for (int o = 32; o >= 1; o >>= 1) {
    sum += __shfl_xor(sum, o, 64);
    smallest = fminf(smallest, __shfl_xor(smallest, o, 64));
  }
  if (!all_in_play) sum += static_cast<double>(need) * static_cast<double>(sqrtf(__uint_as_float(prefix)));
  sum -= static_cast<double>(sqrtf(smallest));  // hit 0 of nearestKSearch(k + 1) is the query itself
  if (lane == 0) {
    distances[remap ? remap[order[j]] : order[j]] = static_cast<float>(sum / static_cast<double>(mean_k));
    // (every cached value is below T0, and at least k of them were: the k + 1 nearest lie within sqrt(T0); a cloud of fewer
    // than k points: no bound needed, every point was seen -- T0 is then +inf anyway)
    if (kth) kth[remap ? remap[order[j]] : order[j]] = T0;
    redo[j] = 0;
  }
  tally_up(false, far);
}

// sum and sum of squares (fp32 squares, as the reference) of the distances, fp64 accumulation.  One pair of partial sums
// per CHUNK of kSorChunk consecutive places of the cell order (a workgroup per chunk, every lane a fixed sub-sequence, a
// fixed reduction tree), added up over the chunks in a fixed order by k_sor_threshold: no atomics, so the threshold is the
// same on every run -- and the same however the chunks are dealt out to GPUs (pcp_sor_partial: a GPU's slab of the cell
// order is whole chunks; the concatenation of the slabs' pairs is the array one GPU computes).
constexpr int64_t kSorChunk = 16384;
__global__ __launch_bounds__(kMB) void k_sor_stats(const float *__restrict__ distances, const int32_t *__restrict__ order,
                                                   const int32_t *__restrict__ remap, int64_t n, int64_t first_chunk,
                                                   double *__restrict__ partial /* [chunks of the cloud][2] */) {
  __shared__ double sh[2][kMB / 64];
  const int64_t chunk = first_chunk + blockIdx.x;
  const int64_t lo = chunk * kSorChunk, hi = min(lo + kSorChunk, n);
  double s = 0.0, q = 0.0;
  for (int64_t j = lo + threadIdx.x; j < hi; j += kMB) {
    const float d = distances[remap ? remap[order[j]] : order[j]];
    s += static_cast<double>(d);
    q += static_cast<double>(__fmul_rn(d, d));
  }
  for (int o = 32; o >= 1; o >>= 1) {
    s += __shfl_xor(s, o, 64);
    q += __shfl_xor(q, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    sh[0][threadIdx.x >> 6] = s;
    sh[1][threadIdx.x >> 6] = q;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ts = 0.0, tq = 0.0;
    for (int k = 0; k < kMB / 64; ++k) {
      ts += sh[0][k];
      tq += sh[1][k];
    }
    partial[2 * chunk] = ts;
    partial[2 * chunk + 1] = tq;
  }
}

// mean + std_mul * stddev of the distances from the workgroups' partial sums (statistical_outlier_removal.hpp [upstream]), on the device:
// the host does not have to wait for the sums between the two kernels.  Individually rounded IEEE operations, as the
// host form compiles.
__global__ __launch_bounds__(64) void k_sor_threshold(const double *__restrict__ partial, int64_t blocks, int64_t n, double std_mul,
                                                      double *__restrict__ threshold) {
#pragma clang fp contract(off)
  double s = 0.0, q = 0.0;
  for (int64_t b = threadIdx.x; b < blocks; b += 64) {
    s += partial[2 * b];
    q += partial[2 * b + 1];
  }
  for (int o = 32; o >= 1; o >>= 1) {
    s += __shfl_xor(s, o, 64);
    q += __shfl_xor(q, o, 64);
  }
  if (threadIdx.x == 0) {
    const double dn = static_cast<double>(n);
    const double mean = s / dn;
    const double variance = (q - s * s / dn) / (dn - 1.0);
    *threshold = mean + std_mul * sqrt(variance);
  }
}

// keep flags (under the indices `distances` uses) of the points at places [begin, end) of the cell order
__global__ __launch_bounds__(kMB) void k_sor_classify(const float *__restrict__ distances, const int32_t *__restrict__ order,
                                                      const int32_t *__restrict__ remap, int64_t begin, int64_t end,
                                                      const double *__restrict__ threshold, uint8_t *__restrict__ keep) {
  const int64_t j = begin + static_cast<int64_t>(blockIdx.x) * kMB + threadIdx.x;
  if (j >= end) return;
  const int64_t i = remap ? remap[order[j]] : order[j];
  keep[i] = !(static_cast<double>(distances[i]) > *threshold) ? 1 : 0;
}

// a cloud on the device the smoothing stages operate on (the uploaded map, or an
// intermediate of pcp_cloud_smooth); mn/mx = its bounding box
struct CloudView {
  const float *x, *y, *z;
  int64_t n;
  float mn[3], mx[3];
  // nullptr, or view index -> caller's point index: the uploaded map is walked through its
  // Morton-ordered copy (cell binning then permutes nearby memory only) and results are
  // reported under the caller's indices
  const int32_t *remap;
};

__device__ __forceinline__ uint32_t ordered_bits(float f) {
  const uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// bounding box as order-preserving uint keys: box[a * kBoxStride] = min of axis a, box[(3 + a) * kBoxStride] = max
constexpr int kBoxStride = 64;
__global__ __launch_bounds__(kMB) void k_bbox(const float *__restrict__ x, const float *__restrict__ y,
                                              const float *__restrict__ z, int64_t n, uint32_t *__restrict__ box) {
  uint32_t lo[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, hi[3] = {0u, 0u, 0u};
  // four points per lane and step (16-B loads: the planes are 16-B aligned); the first version's one dword per lane
  // and 38 dependent trips per lane took 293 us for 10 M points
  const int64_t quads = n >> 2;
  for (int64_t q = static_cast<int64_t>(blockIdx.x) * kMB + threadIdx.x; q < quads; q += static_cast<int64_t>(gridDim.x) * kMB) {
    const float4 vx = reinterpret_cast<const float4 *>(x)[q], vy = reinterpret_cast<const float4 *>(y)[q],
                 vz = reinterpret_cast<const float4 *>(z)[q];
    const float px[4] = {vx.x, vx.y, vx.z, vx.w}, py[4] = {vy.x, vy.y, vy.z, vy.w}, pz[4] = {vz.x, vz.y, vz.z, vz.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t k[3] = {ordered_bits(px[e]), ordered_bits(py[e]), ordered_bits(pz[e])};
      for (int a = 0; a < 3; ++a) {
        lo[a] = min(lo[a], k[a]);
        hi[a] = max(hi[a], k[a]);
      }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {  // ragged tail
    const int64_t i = (quads << 2) + threadIdx.x;
    const uint32_t k[3] = {ordered_bits(x[i]), ordered_bits(y[i]), ordered_bits(z[i])};
    for (int a = 0; a < 3; ++a) {
      lo[a] = min(lo[a], k[a]);
      hi[a] = max(hi[a], k[a]);
    }
  }
  // wavefront, then workgroup, then one atomic per workgroup and bound on a cache line of its own (kBoxStride words
  // apart): the first version's 6 atomics per wavefront on one line -- 98 k of them, serialised at the L2 -- took
  // 1.1 ms of the kernel's 1.12 ms at 10 M points
  __shared__ uint32_t part[6][kMB / 64];
  for (int a = 0; a < 3; ++a) {
    for (int o = 32; o >= 1; o >>= 1) {
      lo[a] = min(lo[a], static_cast<uint32_t>(__shfl_xor(static_cast<int>(lo[a]), o, 64)));
      hi[a] = max(hi[a], static_cast<uint32_t>(__shfl_xor(static_cast<int>(hi[a]), o, 64)));
    }
    if ((threadIdx.x & 63) == 0) {
      part[a][threadIdx.x >> 6] = lo[a];
      part[3 + a][threadIdx.x >> 6] = hi[a];
    }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    uint32_t v = part[threadIdx.x][0];
    for (int w = 1; w < kMB / 64; ++w) v = threadIdx.x < 3 ? min(v, part[threadIdx.x][w]) : max(v, part[threadIdx.x][w]);
    if (threadIdx.x < 3)
      atomicMin(box + threadIdx.x * kBoxStride, v);
    else
      atomicMax(box + threadIdx.x * kBoxStride, v);
  }
}

// survivors of a view, in view order: out[k] = view[pos[k]], index[k] = the caller's index of that point
__global__ __launch_bounds__(kMB) void k_gather_view(const float *__restrict__ x, const float *__restrict__ y,
                                                     const float *__restrict__ z, const int32_t *__restrict__ remap,
                                                     const int32_t *__restrict__ pos, int64_t m, float *__restrict__ ox,
                                                     float *__restrict__ oy, float *__restrict__ oz,
                                                     int32_t *__restrict__ index) {
  const int64_t k = static_cast<int64_t>(blockIdx.x) * kMB + threadIdx.x;
  if (k >= m) return;
  const int32_t i = pos[k];
  ox[k] = x[i];
  oy[k] = y[i];
  oz[k] = z[i];
  index[k] = remap ? remap[i] : i;
}

// rows that survive (flag) announce themselves under their caller's index: mark[index] = 1, where[index] = row
__global__ __launch_bounds__(kMB) void k_mark_rows(const uint8_t *__restrict__ flag, const int32_t *__restrict__ index,
                                                   int64_t m, uint8_t *__restrict__ mark, int32_t *__restrict__ where) {
  const int64_t r = static_cast<int64_t>(blockIdx.x) * kMB + threadIdx.x;
  if (r >= m || !flag[r]) return;
  const int32_t i = index[r];
  mark[i] = 1;
  where[i] = static_cast<int32_t>(r);
}

// positions of the fitted rows as SoA planes: out[k] = rows[index[k]].xyz
__global__ __launch_bounds__(kMB) void k_rows_xyz(const float *__restrict__ rows, const int32_t *__restrict__ index, int64_t m,
                                                  float *__restrict__ ox, float *__restrict__ oy, float *__restrict__ oz) {
  const int64_t k = static_cast<int64_t>(blockIdx.x) * kMB + threadIdx.x;
  if (k >= m) return;
  const float *t = rows + static_cast<int64_t>(index[k]) * kRowStride;
  ox[k] = t[0];
  oy[k] = t[1];
  oz[k] = t[2];
}

// out[k] = map[in[k]]
__global__ __launch_bounds__(kMB) void k_remap_index_to(const int32_t *__restrict__ in, int64_t m, const int32_t *__restrict__ map,
                                                        int32_t *__restrict__ out) {
  const int64_t k = static_cast<int64_t>(blockIdx.x) * kMB + threadIdx.x;
  if (k < m) out[k] = map[in[k]];
}

// the chain's survivors in the caller's order: result t is the point list[t]; where[] names its row among the fitted
// rows, row_view[] that row's place in rows[]
__global__ __launch_bounds__(kMB) void k_final_rows(const int32_t *__restrict__ list, int64_t kept, const int32_t *__restrict__ where,
                                                    const int32_t *__restrict__ row_view, const float *__restrict__ rows,
                                                    float *__restrict__ xyz, float *__restrict__ normal, float *__restrict__ curv,
                                                    int32_t *__restrict__ index) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * kMB + threadIdx.x;
  if (t >= kept) return;
  const int32_t i = list[t];
  const float *r = rows + static_cast<int64_t>(row_view[where[i]]) * kRowStride;
  xyz[3 * t + 0] = r[0]; xyz[3 * t + 1] = r[1]; xyz[3 * t + 2] = r[2];
  normal[3 * t + 0] = r[3]; normal[3 * t + 1] = r[4]; normal[3 * t + 2] = r[5];
  curv[t] = r[6];
  index[t] = i;
}

// AoS xyz[3m] -> SoA planes
__global__ __launch_bounds__(kMB) void k_deinterleave(const float *__restrict__ xyz, int64_t m, float *__restrict__ ox,
                                                      float *__restrict__ oy, float *__restrict__ oz) {
  const int64_t k = static_cast<int64_t>(blockIdx.x) * kMB + threadIdx.x;
  if (k >= m) return;
  ox[k] = xyz[3 * k + 0];
  oy[k] = xyz[3 * k + 1];
  oz[k] = xyz[3 * k + 2];
}

// index[k] = map[index[k]]
__global__ __launch_bounds__(kMB) void k_remap_index(int32_t *__restrict__ index, int64_t m,
                                                     const int32_t *__restrict__ map) {
  const int64_t k = static_cast<int64_t>(blockIdx.x) * kMB + threadIdx.x;
  if (k < m) index[k] = map[index[k]];
}

static inline uint32_t blocks_of(int64_t n) { return static_cast<uint32_t>(std::max<int64_t>(1, div_up(n, kMB))); }

// device-wide exclusive scan of counts[0..m) into out[0..m], out[m] = total
static int exclusive_scan(pcp_context *ctx, int32_t *counts, int64_t m) {
  const int64_t tiles = std::max<int64_t>(1, div_up(m + 1, kScanTile));
  PCP_HIP_TRY(ctx, ctx->s_tiles.ensure(static_cast<size_t>(tiles) + 4));
  hipLaunchKernelGGL(k_scan_tile_sums, dim3(static_cast<uint32_t>(tiles)), dim3(kScanBlock), 0, ctx->stream, counts,
                     m + 1, ctx->s_tiles.p);
  hipLaunchKernelGGL(k_scan_tile_offsets, dim3(1), dim3(kScanSingle), 0, ctx->stream, ctx->s_tiles.p, tiles,
                     static_cast<unsigned long long *>(nullptr));
  hipLaunchKernelGGL(k_scan_apply, dim3(static_cast<uint32_t>(tiles)), dim3(kScanBlock), 0, ctx->stream, counts, m + 1,
                     ctx->s_tiles.p, counts);
  PCP_HIP_TRY(ctx, hipGetLastError());
  return PCP_OK;
}

// uniform grid over a cloud view, cell edge >= `cell`; fills ctx->g_*
// geometry_only: just the grid description and a large enough cell table (the density probe of sor_run fills it).
static int build_grid(pcp_context *ctx, const CloudView &cv, float cell, float radius, GridDesc *out,
                      bool geometry_only = false) {
  const int64_t n = cv.n;
  GridDesc g{};
  const float *mn = cv.mn, *mx = cv.mx;
  for (int a = 0; a < 3; ++a)
    if (!(std::fabs(mn[a]) <= FLT_MAX) || !(std::fabs(mx[a]) <= FLT_MAX) || !(cell > 0.0f))
      return set_error(ctx, PCP_ERR_INVALID, "the smoothing stages need finite coordinates (bounding box %g .. %g on axis %d)",
                       static_cast<double>(mn[a]), static_cast<double>(mx[a]), a);
  // Up to kMaxGridCells cells the table of cell starts is dense (one int32 per cell, 2 GiB at most); up to
  // kMaxSparseCells it holds the occupied cells only, found through a bitmap with running popcounts (GridDesc; 6 GiB at
  // most).  A box that needs more cells than that at the wanted edge gets a coarser grid: the searches stay exact, every
  // doubling of the edge multiplies the candidates per query by up to 8 (a map with a stray point kilometres away).
  // PCP_GRID_SPARSE=1 / 0 forces the sparse / dense form (tests).
  const char *form_env = std::getenv("PCP_GRID_SPARSE");  // read per call: the tests flip it inside one process
  const int force_form = form_env ? (form_env[0] == '1' ? 1 : 0) : -1;
  const double cap = (force_form == 0 || geometry_only) ? kMaxGridCells : kMaxSparseCells;
  double cells = 0.0;
  for (int doubling = 0;; ++doubling) {  // bound the table: grow the cell until it fits
    const double ex = static_cast<double>(mx[0] - mn[0]) / cell + 1.0, ey = static_cast<double>(mx[1] - mn[1]) / cell + 1.0,
                 ez = static_cast<double>(mx[2] - mn[2]) / cell + 1.0;
    cells = ex * ey * ez;
    if (cells <= cap) break;
    if (doubling > 300) return set_error(ctx, PCP_ERR_INVALID, "no uniform grid fits this cloud's bounding box");
    cell *= 2.0f;
  }
  const bool sparse = !geometry_only && (force_form == 1 || cells > kMaxGridCells);
  g.minx = mn[0];
  g.miny = mn[1];
  g.minz = mn[2];
  g.inv_cell = 1.0f / cell;
  g.nx = static_cast<int32_t>(floorf((mx[0] - mn[0]) * g.inv_cell)) + 1;
  g.ny = static_cast<int32_t>(floorf((mx[1] - mn[1]) * g.inv_cell)) + 1;
  g.nz = static_cast<int32_t>(floorf((mx[2] - mn[2]) * g.inv_cell)) + 1;
  g.reach = static_cast<int32_t>(ceilf(radius * g.inv_cell));
  const int64_t ncell = static_cast<int64_t>(g.nx) * g.ny * g.nz;
  const size_t sn = static_cast<size_t>(n);
  const size_t plane = (sn + 3) & ~size_t(3);
  PCP_HIP_TRY(ctx, ctx->g_cell.ensure(sn + 4));
  PCP_HIP_TRY(ctx, ctx->g_rank.ensure(sn + 4));
  PCP_HIP_TRY(ctx, ctx->g_order.ensure(2 * sn + 8));
  {
    // k_sor_select<true> reads the planes in 16-byte pieces through one descriptor: a run's ragged end reads up to three
    // floats past the run -- the next cell's points, the up-to-3 floats of padding behind a plane, the start of the next plane
    // -- and masks those candidates by their x alone.  What it reads there must never be a NaN bit pattern (inf + NaN would
    // slip past the mask as NaN): the buffer only ever holds finite coordinates (non-finite clouds are refused) or, right
    // after an allocation, whatever hipMalloc left -- so a NEW allocation is zeroed once, and the padding stays non-NaN for
    // the buffer's life.
    const size_t before = ctx->g_xyz.count;  // (ensure() only ever grows: a changed count is a new allocation)
    PCP_HIP_TRY(ctx, ctx->g_xyz.ensure(3 * plane + 4));
    if (ctx->g_xyz.count != before) PCP_HIP_TRY(ctx, hipMemsetAsync(ctx->g_xyz.p, 0, ctx->g_xyz.count * sizeof(float), ctx->stream));
  }
  if (geometry_only) {
    PCP_HIP_TRY(ctx, ctx->g_start.ensure(static_cast<size_t>(ncell) + 8));
    *out = g;
    return PCP_OK;
  }
  const float *x = cv.x, *y = cv.y, *z = cv.z;
  int64_t entries = ncell;  // entries of the table of cell starts (+ 1 for the total)
  if (sparse) {
    // occupied cells: one bit each; running popcount per word; the table gets one entry per set bit
    const int64_t words = ncell / 64 + 2;  // the lookups reach cell id ncell (one past the last)
    PCP_HIP_TRY(ctx, ctx->g_occ.ensure(static_cast<size_t>(words) + 2));
    PCP_HIP_TRY(ctx, ctx->g_occ_rank.ensure(static_cast<size_t>(words) + 8));
    PCP_HIP_TRY(ctx, hipMemsetAsync(ctx->g_occ.p, 0, (static_cast<size_t>(words) + 2) * 8, ctx->stream));
    g.occ = ctx->g_occ.p;
    g.occ_rank = ctx->g_occ_rank.p;
    {
      LaunchTimer t(ctx, PCP_K_MLS_GRID);
      hipLaunchKernelGGL(k_grid_mark, dim3(blocks_of(n)), dim3(kMB), 0, ctx->stream, x, y, z, n, g, ctx->g_occ.p);
      hipLaunchKernelGGL(k_grid_popc, dim3(blocks_of(words)), dim3(kMB), 0, ctx->stream, ctx->g_occ.p, words, ctx->g_occ_rank.p);
      int rc = exclusive_scan(ctx, ctx->g_occ_rank.p, words);
      if (rc != PCP_OK) return rc;
    }
    int32_t occupied = 0;
    PCP_HIP_TRY(ctx, hipMemcpyAsync(&occupied, ctx->g_occ_rank.p + words, 4, hipMemcpyDeviceToHost, ctx->stream));
    PCP_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    entries = occupied;
  }
  PCP_HIP_TRY(ctx, ctx->g_start.ensure(static_cast<size_t>(entries) + 8));
  PCP_HIP_TRY(ctx, hipMemsetAsync(ctx->g_start.p, 0, (static_cast<size_t>(entries) + 8) * 4, ctx->stream));
  {
    LaunchTimer t(ctx, PCP_K_MLS_GRID);
    if (sparse)
      hipLaunchKernelGGL(k_grid_count_sparse, dim3(blocks_of(n)), dim3(kMB), 0, ctx->stream, x, y, z, n, g, ctx->g_cell.p,
                         ctx->g_rank.p, ctx->g_start.p);
    else
      hipLaunchKernelGGL(k_grid_count, dim3(blocks_of(n)), dim3(kMB), 0, ctx->stream, x, y, z, n, g, ctx->g_cell.p,
                         ctx->g_rank.p, ctx->g_start.p);
    int rc = exclusive_scan(ctx, ctx->g_start.p, entries);
    if (rc != PCP_OK) return rc;
    hipLaunchKernelGGL(k_grid_scatter, dim3(blocks_of(n)), dim3(kMB), 0, ctx->stream, n, ctx->g_cell.p, ctx->g_rank.p,
                       ctx->g_start.p, ctx->g_order.p + sn + 4);
    hipLaunchKernelGGL(k_grid_order, dim3(blocks_of(n)), dim3(kMB), 0, ctx->stream, x, y, z, n,
                       static_cast<int64_t>(plane), ctx->g_cell.p, ctx->g_start.p, ctx->g_order.p + sn + 4,
                       ctx->g_order.p, ctx->g_xyz.p);
    PCP_HIP_TRY(ctx, hipGetLastError());
  }
  *out = g;
  return PCP_OK;
}

// keep the `kept` result rows (of m) that keep_index names, in order.  They go into the context's second set of result
// buffers, sized like the first (m, not kept), and the sets are swapped: either set then serves the next run without a
// hipMalloc / hipFree pair (fresh buffers per call cost the enableMLS chain 1.5 ms).
static int compact_results(pcp_context *ctx, const int32_t *keep_index, int64_t m, int64_t kept, bool positions_only = false) {
  const size_t sm = static_cast<size_t>(m);
  PCP_HIP_TRY(ctx, ctx->mls_alt_xyz.ensure(std::max(3 * sm + 4, ctx->mls_xyz.count)));
  if (positions_only) {  // (the other arrays hold nothing: they stay where they are)
    if (kept > 0) {
      hipLaunchKernelGGL(k_voxel_compact, dim3(blocks_of(kept)), dim3(kMB), 0, ctx->stream, keep_index, kept, ctx->mls_xyz.p,
                         static_cast<const float *>(nullptr), static_cast<const float *>(nullptr), static_cast<const int32_t *>(nullptr),
                         ctx->mls_alt_xyz.p, static_cast<float *>(nullptr), static_cast<float *>(nullptr), static_cast<int32_t *>(nullptr));
      PCP_HIP_TRY(ctx, hipGetLastError());
    }
    std::swap(ctx->mls_xyz, ctx->mls_alt_xyz);
    return PCP_OK;
  }
  PCP_HIP_TRY(ctx, ctx->mls_alt_normal.ensure(std::max(3 * sm + 4, ctx->mls_normal.count)));
  PCP_HIP_TRY(ctx, ctx->mls_alt_curv.ensure(std::max(sm + 4, ctx->mls_curv.count)));
  PCP_HIP_TRY(ctx, ctx->mls_alt_index.ensure(std::max(sm + 4, ctx->mls_index.count)));
  if (kept > 0) {
    hipLaunchKernelGGL(k_voxel_compact, dim3(blocks_of(kept)), dim3(kMB), 0, ctx->stream, keep_index, kept, ctx->mls_xyz.p,
                       ctx->mls_normal.p, ctx->mls_curv.p, ctx->mls_index.p, ctx->mls_alt_xyz.p, ctx->mls_alt_normal.p,
                       ctx->mls_alt_curv.p, ctx->mls_alt_index.p);
    PCP_HIP_TRY(ctx, hipGetLastError());
  }
  std::swap(ctx->mls_xyz, ctx->mls_alt_xyz);
  std::swap(ctx->mls_normal, ctx->mls_alt_normal);
  std::swap(ctx->mls_curv, ctx->mls_alt_curv);
  std::swap(ctx->mls_index, ctx->mls_alt_index);
  return PCP_OK;
}

// performUpsampling(VOXEL_GRID_DILATION) on the fitted surfaces in ctx->m_state.  Two steps: vgd_prepare stamps the
// dilated voxel set into the bitmap and counts it (per bitmap word on the device, per tile of 1024 words on the host,
// in 64 bits); vgd_emit turns a range of bitmap words -- a slab of the key order -- into output points.  The one-shot
// form emits the whole range; pcp_mls_stream_* emit it in chunks, which is how a voxel set above 2^31 points, or above
// what the device can hold as output (78 B per point), is produced at all: the reference's own configuration (1 mm
// voxels, 4 dilations, PointCloudProcessor.cpp:78-81) makes ~3.8e9 points of the 10 M-point C3 map.
struct VgdStream {  // plain data: kept in ctx->vgd_blob between pcp_mls_stream_begin and _next
  VoxelDesc v;
  GridDesc g;
  const int32_t *remap;
  size_t plane;
  int32_t order_poly;
  int32_t bricks;         // 1: the voxel set is in bricks (chunks are runs of strips), 0: dense bitmap (runs of words)
  int32_t NBX, NBY, NBZ;
};

// PCP_VGD_DENSE=1: the dense bitmap of rounds 2-3 (one bit per voxel of the bounding box); results identical
static bool vgd_dense_form() {
  const char *e = std::getenv("PCP_VGD_DENSE");
  return e && e[0] == '1';
}

static BrickDesc brick_desc(pcp_context *ctx, const VoxelDesc &v) {
  BrickDesc B{};
  B.v = v;
  B.NBX = (v.NX + kBrick - 1) / kBrick;
  B.NBY = (v.NY + kBrick - 1) / kBrick;
  B.NBZ = (v.NZ + kBrick - 1) / kBrick;
  B.occ = ctx->v_occ.p;
  B.rank = ctx->v_rank.p;
  B.bits = ctx->v_bitmap.p;
  return B;
}

// The dilated voxel set in bricks: occupancy marks, ranks, storage sized from the number of occupied bricks, stamps,
// counts per strip and per plane.  `plane_counts` (NX entries, 64-bit) come to the host; the strips' counts stay on the
// device (ctx->v_offsets) and the chunk planner fetches a plane's strips only where a chunk boundary falls inside it.
static int vgd_prepare_bricks(pcp_context *ctx, const CloudView &cv, const VoxelDesc &v0, std::vector<unsigned long long> *plane_counts,
                              unsigned long long *out_total, bool *fits) {
  *fits = false;
  VoxelDesc v = v0;
  v.words = 0;
  BrickDesc B = brick_desc(ctx, v);
  if (B.NBZ > kBrickColumnMax) return PCP_OK;  // (taller than the brick form handles: the caller falls back to the dense form)
  const int64_t places = static_cast<int64_t>(B.NBX) * B.NBY * B.NBZ;
  const int64_t occ_words = (places + 31) / 32 + 1;
  const int64_t strips = static_cast<int64_t>(v.NX) * B.NBY;
  size_t free_b = 0, total_b = 0;
  PCP_HIP_TRY(ctx, hipMemGetInfo(&free_b, &total_b));
  const double have = static_cast<double>(free_b) + 4.0 * (static_cast<double>(ctx->v_occ.count) + static_cast<double>(ctx->v_rank.count) +
                                                            static_cast<double>(ctx->v_offsets.count) + static_cast<double>(ctx->v_bitmap.count));
  if (8.0 * static_cast<double>(occ_words) + 4.0 * static_cast<double>(strips) > 0.5 * have) return PCP_OK;
  PCP_HIP_TRY(ctx, ctx->v_occ.ensure(static_cast<size_t>(occ_words) + 8));
  PCP_HIP_TRY(ctx, ctx->v_rank.ensure(static_cast<size_t>(occ_words) + 8));
  PCP_HIP_TRY(ctx, ctx->v_offsets.ensure(static_cast<size_t>(strips) + 8));
  PCP_HIP_TRY(ctx, ctx->v_plane.ensure(static_cast<size_t>(v.NX) + 8));
  PCP_HIP_TRY(ctx, hipMemsetAsync(ctx->v_occ.p, 0, static_cast<size_t>(occ_words) * 4, ctx->stream));
  B = brick_desc(ctx, v);
  const int64_t n = cv.n;
  const int64_t tiles = std::max<int64_t>(1, div_up(occ_words, kScanTile));
  PCP_HIP_TRY(ctx, ctx->s_tiles.ensure(static_cast<size_t>(tiles) + 16));
  unsigned long long *d_total = reinterpret_cast<unsigned long long *>(ctx->v_plane.p);  // (word 0 until the planes are counted)
  {
    LaunchTimer t(ctx, PCP_K_MLS_VOXEL);
    hipLaunchKernelGGL(k_brick_mark, dim3(blocks_of(n)), dim3(kMB), 0, ctx->stream, cv.x, cv.y, cv.z, n, B, ctx->v_occ.p);
    hipLaunchKernelGGL(k_brick_popc, dim3(scan_grid(div_up(occ_words, kScanBlock))), dim3(kScanBlock), 0, ctx->stream, ctx->v_occ.p,
                       occ_words, ctx->v_rank.p);
    hipLaunchKernelGGL(k_scan_tile_sums, dim3(scan_grid(tiles)), dim3(kScanBlock), 0, ctx->stream, ctx->v_rank.p, occ_words, ctx->s_tiles.p);
    hipLaunchKernelGGL(k_scan_tile_offsets, dim3(1), dim3(kScanSingle), 0, ctx->stream, ctx->s_tiles.p, tiles, d_total);
    hipLaunchKernelGGL(k_scan_apply, dim3(scan_grid(tiles)), dim3(kScanBlock), 0, ctx->stream, ctx->v_rank.p, occ_words, ctx->s_tiles.p,
                       ctx->v_rank.p);
    PCP_HIP_TRY(ctx, hipGetLastError());
  }
  unsigned long long n_bricks = 0;
  {
    unsigned long long *dst = ctx->readback ? static_cast<unsigned long long *>(ctx->readback) : &n_bricks;  // pinned: no staging
    PCP_HIP_TRY(ctx, hipMemcpyAsync(dst, d_total, 8, hipMemcpyDeviceToHost, ctx->stream));
    PCP_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    n_bricks = *dst;
  }
  const double brick_bytes = static_cast<double>(n_bricks) * kBrickWords * 4.0;
  PCP_HIP_TRY(ctx, hipMemGetInfo(&free_b, &total_b));
  if (n_bricks >= (1ull << 31) || brick_bytes > 0.8 * (static_cast<double>(free_b) + 4.0 * static_cast<double>(ctx->v_bitmap.count)))
    return set_error(ctx, PCP_ERR_NOMEM, "pcp_mls_process: the dilated voxel set needs %llu bricks of 16^3 voxels (%.3g GB) and does not "
                     "fit the device; use a larger vgd_voxel_size or crop the cloud", n_bricks, brick_bytes / 1e9);
  const size_t bw = static_cast<size_t>(n_bricks) * kBrickWords;
  PCP_HIP_TRY(ctx, ctx->v_bitmap.ensure(bw + 8));
  PCP_HIP_TRY(ctx, hipMemsetAsync(ctx->v_bitmap.p, 0, (bw + 8) * 4, ctx->stream));
  PCP_HIP_TRY(ctx, hipMemsetAsync(ctx->v_offsets.p, 0, static_cast<size_t>(strips) * 4, ctx->stream));
  PCP_HIP_TRY(ctx, hipMemsetAsync(ctx->v_plane.p, 0, static_cast<size_t>(v.NX) * 8, ctx->stream));
  B = brick_desc(ctx, v);
  {
    LaunchTimer t(ctx, PCP_K_MLS_VOXEL);
    hipLaunchKernelGGL(k_brick_stamp, dim3(blocks_of(n)), dim3(kMB), 0, ctx->stream, cv.x, cv.y, cv.z, n, B);
    const int64_t cols = static_cast<int64_t>(B.NBX) * B.NBY;
    hipLaunchKernelGGL(k_brick_strip_counts, dim3(static_cast<uint32_t>(std::min<int64_t>(cols, int64_t(1) << 20))), dim3(64), 0, ctx->stream, B,
                       ctx->v_offsets.p, reinterpret_cast<unsigned long long *>(ctx->v_plane.p));
    PCP_HIP_TRY(ctx, hipGetLastError());
  }
  plane_counts->resize(static_cast<size_t>(v.NX));
  PCP_HIP_TRY(ctx, hipMemcpyAsync(plane_counts->data(), ctx->v_plane.p, plane_counts->size() * 8, hipMemcpyDeviceToHost, ctx->stream));
  PCP_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  unsigned long long total = 0;
  for (unsigned long long c : *plane_counts) total += c;
  *out_total = total;
  *fits = true;
  return PCP_OK;
}

// chunks of the brick form = maximal runs of whole strips that hold at most `capacity` voxels (a strip alone holds at most
// 16 NZ): whole planes while they fit; where a boundary falls inside a plane, that plane's strip counts are fetched
static int vgd_plan_bricks(pcp_context *ctx, const VoxelDesc &v, int32_t NBY, const std::vector<unsigned long long> &plane_counts,
                           int64_t capacity, std::vector<int64_t> *chunks) {
  chunks->clear();
  std::vector<int32_t> strip(static_cast<size_t>(NBY));
  int64_t s_begin = 0, cnt = 0;
  auto close = [&](int64_t s_end) {
    if (cnt > 0) {
      chunks->push_back(s_begin);
      chunks->push_back(s_end);
      chunks->push_back(cnt);
    }
    s_begin = s_end;
    cnt = 0;
  };
  for (int32_t ix = 0; ix < v.NX; ++ix) {
    const int64_t pc = static_cast<int64_t>(plane_counts[static_cast<size_t>(ix)]);
    if (pc == 0) continue;
    if (cnt + pc <= capacity) {
      cnt += pc;
      continue;
    }
    PCP_HIP_TRY(ctx, hipMemcpy(strip.data(), ctx->v_offsets.p + static_cast<int64_t>(ix) * NBY, static_cast<size_t>(NBY) * 4, hipMemcpyDeviceToHost));
    for (int32_t by = 0; by < NBY; ++by) {
      const int64_t c = strip[static_cast<size_t>(by)];
      if (cnt + c > capacity && cnt > 0) close(static_cast<int64_t>(ix) * NBY + by);
      cnt += c;
    }
  }
  close(static_cast<int64_t>(v.NX) * NBY);
  return PCP_OK;
}

// the voxel grid of the cloud: MLSVoxelGrid's origin and the largest cell index per axis + the dilation's reach
static int vgd_describe(pcp_context *ctx, const CloudView &cv, const pcp_mls_params *p, VoxelDesc *out_v) {
  VoxelDesc v{};
  v.bminx = cv.mn[0];
  v.bminy = cv.mn[1];
  v.bminz = cv.mn[2];
  v.vs = p->vgd_voxel_size;
  v.it = p->vgd_iterations;
  // largest cell index per axis (float division as MLSVoxelGrid::getCellIndex) + dilation reach
  const int64_t mx = static_cast<int64_t>((cv.mx[0] - v.bminx) / v.vs) + v.it + 1;
  const int64_t my = static_cast<int64_t>((cv.mx[1] - v.bminy) / v.vs) + v.it + 1;
  const int64_t mz = static_cast<int64_t>((cv.mx[2] - v.bminz) / v.vs) + v.it + 1;
  if (mx >= (int64_t(1) << 31) || my >= (int64_t(1) << 31) || mz >= (int64_t(1) << 31))
    return set_error(ctx, PCP_ERR_NOMEM, "pcp_mls_process: the %lld x %lld x %lld voxel grid at %.4g m has an axis of 2^31 voxels or more; "
                     "use a larger vgd_voxel_size or crop the cloud", (long long)mx, (long long)my, (long long)mz, static_cast<double>(v.vs));
  v.NX = static_cast<int32_t>(mx);
  v.NY = static_cast<int32_t>(my);
  v.NZ = static_cast<int32_t>(mz);
  v.words = (static_cast<int64_t>(v.NX) * v.NY * v.NZ + 31) / 32;
  *out_v = v;
  return PCP_OK;
}

static int vgd_prepare(pcp_context *ctx, const CloudView &cv, const pcp_mls_params *p, VoxelDesc *out_v,
                       std::vector<int32_t> *tile_counts, unsigned long long *out_total) {
  const int64_t n = cv.n;
  VoxelDesc v{};
  int rcd = vgd_describe(ctx, cv, p, &v);
  if (rcd != PCP_OK) return rcd;
  const double bits = static_cast<double>(v.NX) * static_cast<double>(v.NY) * static_cast<double>(v.NZ);
  size_t free_b = 0, total_b = 0;
  PCP_HIP_TRY(ctx, hipMemGetInfo(&free_b, &total_b));
  // the bitmap (bits / 8; the per-tile counts are 1 / 8192 of that) must fit with room for the outputs
  const double have = static_cast<double>(free_b) + static_cast<double>(ctx->v_bitmap.count) * 4.0;
  if (bits / 8.0 > 0.6 * have)
    return set_error(ctx, PCP_ERR_NOMEM,
                     "pcp_mls_process: the %d x %d x %d voxel grid (%.3g voxels at %.4g m) does not fit the device as a dense bitmap; "
                     "use a larger vgd_voxel_size or crop the cloud",
                     v.NX, v.NY, v.NZ, bits, static_cast<double>(v.vs));
  const size_t sw = static_cast<size_t>(v.words);
  PCP_HIP_TRY(ctx, ctx->v_bitmap.ensure(sw + 8));
  PCP_HIP_TRY(ctx, hipMemsetAsync(ctx->v_bitmap.p, 0, (sw + 8) * 4, ctx->stream));
  const int64_t tiles = std::max<int64_t>(1, div_up(v.words, kScanTile));
  PCP_HIP_TRY(ctx, ctx->v_offsets.ensure(static_cast<size_t>(tiles) + 8));  // set bits per tile of the whole bitmap
  {
    LaunchTimer t(ctx, PCP_K_MLS_VOXEL);
    hipLaunchKernelGGL(k_voxel_stamp, dim3(blocks_of(n)), dim3(kMB), 0, ctx->stream, cv.x, cv.y, cv.z, n, v, ctx->v_bitmap.p);
    hipLaunchKernelGGL(k_voxel_tile_counts, dim3(scan_grid(tiles)), dim3(kScanBlock), 0, ctx->stream, ctx->v_bitmap.p, v.words,
                       ctx->v_offsets.p);
    PCP_HIP_TRY(ctx, hipGetLastError());
  }
  // the tile counts to the host (the caller cuts the key range into chunks from them); one pass for the total
  std::vector<int32_t> &sums = *tile_counts;
  sums.resize(static_cast<size_t>(tiles));
  PCP_HIP_TRY(ctx, hipMemcpyAsync(sums.data(), ctx->v_offsets.p, sums.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
  PCP_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  unsigned long long total = 0;
  for (int32_t c : sums) total += static_cast<unsigned long long>(c);
  *out_total = total;
  *out_v = v;
  return PCP_OK;
}

// the voxels of bitmap words [word0, word1) (word0 a multiple of kScanTile), `count` of them: results in ctx->mls_*
// sample_step > 0: nothing is emitted -- one workgroup of voxels in `sample_step` is projected for max_dx alone (*out_m = 0)
// positions_only: the rows' positions alone (ctx->mls_xyz; the other result arrays are left as they are)
static int vgd_emit(pcp_context *ctx, const VgdStream &S, int64_t word0, int64_t word1, int64_t count, int64_t *out_m,
                    uint32_t *max_dx = nullptr, int32_t sample_step = 0, bool positions_only = false) {
  const size_t st = static_cast<size_t>(count);
  size_t free_b = 0, total_b = 0;
  PCP_HIP_TRY(ctx, hipMemGetInfo(&free_b, &total_b));
  auto short_of_memory = [&]() {
    return static_cast<double>(st) * 78.0 > static_cast<double>(free_b) + static_cast<double>(ctx->mls_xyz.count) * 4.0 * 2.4;
  };
  if (short_of_memory() && ctx->css_next < 0 && !ctx->css_building && ctx->css_dist.p) {
    // the memory an ended stream of the whole chain still holds (4 B per row of its upsampled cloud) goes first
    ctx->css_dist.release();
    PCP_HIP_TRY(ctx, hipMemGetInfo(&free_b, &total_b));
  }
  if (short_of_memory())
    return set_error(ctx, PCP_ERR_NOMEM, "pcp_mls_process: %lld upsampled points do not fit the device memory "
                     "(pcp_mls_stream_begin / _next emit them in chunks)", (long long)count);
  if (sample_step == 0) {  // (a dry run stores nothing)
    PCP_HIP_TRY(ctx, ctx->mls_xyz.ensure(3 * st + 4));
    if (!positions_only) {
      PCP_HIP_TRY(ctx, ctx->mls_normal.ensure(3 * st + 4));
      PCP_HIP_TRY(ctx, ctx->mls_curv.ensure(st + 4));
      PCP_HIP_TRY(ctx, ctx->mls_index.ensure(st + 4));
    }
    PCP_HIP_TRY(ctx, ctx->m_flag.ensure(st + 16));
  }
  PCP_HIP_TRY(ctx, ctx->v_vox.ensure(st + 4));
  int64_t m = count;
  if (count > 0) {
    // dense form: words of the bitmap, counted per tile of kScanTile words; brick form: strips, counted one by one
    const int64_t words = word1 - word0;
    const int64_t tiles = S.bricks ? words : std::max<int64_t>(1, div_up(words, kScanTile));
    // the chunk's counts -> the first place of every tile / strip inside the chunk (exclusive prefix, on a copy: the counts
    // of the whole set stay for the other chunks)
    const int64_t tiles2 = std::max<int64_t>(1, div_up(tiles, kScanTile));
    PCP_HIP_TRY(ctx, ctx->s_tiles.ensure(static_cast<size_t>(tiles + tiles2) + 8));
    int32_t *tile_first = ctx->s_tiles.p, *level2 = ctx->s_tiles.p + tiles + 4;
    PCP_HIP_TRY(ctx, hipMemcpyAsync(tile_first, ctx->v_offsets.p + (S.bricks ? word0 : word0 / kScanTile), static_cast<size_t>(tiles) * 4,
                                    hipMemcpyDeviceToDevice, ctx->stream));
    VoxelEmitArgs e{};
    e.bitmap = ctx->v_bitmap.p;
    e.v = S.v;
    e.sx = ctx->g_xyz.p;
    e.sy = ctx->g_xyz.p + S.plane;
    e.sz = ctx->g_xyz.p + 2 * S.plane;
    e.order = ctx->g_order.p;
    e.remap = S.remap;
    e.start = ctx->g_start.p;
    e.g = S.g;
    // a dilated voxel's corner lies within sqrt(3) * (it + 1) voxels of the point that stamped it (1 % + 1 um of
    // slack for the fp32 roundings of getCellIndex / getPosition)
    e.dmax = static_cast<float>(1.7321 * (S.v.it + 1) * static_cast<double>(S.v.vs) * 1.01 + 1e-6);
    e.vox = ctx->v_vox.p;
    e.total = count;
    e.state = ctx->m_state.p;
    e.order_poly = S.order_poly;
    const int32_t nr_coeff = (S.order_poly + 1) * (S.order_poly + 2) / 2;
    e.required_neighbors = 5 * nr_coeff;
    e.xyz = ctx->mls_xyz.p;
    e.normal = ctx->mls_normal.p;
    e.curv = ctx->mls_curv.p;
    e.index = ctx->mls_index.p;
    e.valid = ctx->m_flag.p;
    e.max_dx = max_dx;
    e.dry = sample_step > 0 ? 1 : (positions_only ? 2 : 0);
    e.block_step = std::max(1, sample_step);
    {
      LaunchTimer t(ctx, PCP_K_MLS_VOXEL);
      hipLaunchKernelGGL(k_scan_tile_sums, dim3(scan_grid(tiles2)), dim3(kScanBlock), 0, ctx->stream, tile_first, tiles, level2);
      hipLaunchKernelGGL(k_scan_tile_offsets, dim3(1), dim3(kScanSingle), 0, ctx->stream, level2, tiles2,
                         static_cast<unsigned long long *>(nullptr));
      hipLaunchKernelGGL(k_scan_apply, dim3(scan_grid(tiles2)), dim3(kScanBlock), 0, ctx->stream, tile_first, tiles, level2, tile_first);
      if (S.bricks) {
        BrickDesc B = brick_desc(ctx, S.v);
        hipLaunchKernelGGL(k_brick_expand, dim3(static_cast<uint32_t>(std::min<int64_t>(words, int64_t(1) << 20))), dim3(64), 0, ctx->stream, B,
                           ctx->v_offsets.p, tile_first, word0, words, ctx->v_vox.p);
      } else {
        hipLaunchKernelGGL(k_voxel_expand, dim3(scan_grid(tiles)), dim3(kScanBlock), 0, ctx->stream, ctx->v_bitmap.p + word0, tile_first,
                           words, word0, ctx->v_vox.p);
      }
      hipLaunchKernelGGL(k_voxel_emit, dim3(static_cast<uint32_t>(div_up(div_up(count, kMB), e.block_step))), dim3(kMB), 0, ctx->stream, e);
      PCP_HIP_TRY(ctx, hipGetLastError());
    }
    if (sample_step > 0) {
      ctx->mls_count = 0;
      if (out_m) *out_m = 0;
      return PCP_OK;
    }
    // voxels whose nearest point has no valid fit are skipped by PCL: compact if any
    PCP_HIP_TRY(ctx, ctx->s_cell.ensure(st + 4));
    int64_t kept = 0;
    int rc = compact_flags(ctx, ctx->m_flag.p, count, ctx->s_cell.p, count, &kept);
    if (rc != PCP_OK) return rc;
    if (kept != count) {
      LaunchTimer t(ctx, PCP_K_MLS_VOXEL);
      if ((rc = compact_results(ctx, ctx->s_cell.p, count, kept, positions_only)) != PCP_OK) return rc;
      m = kept;
    }
  }
  ctx->mls_count = positions_only ? 0 : m;  // (nothing to fetch from a positions-only emission)
  if (out_m) *out_m = m;
  return PCP_OK;
}

static VgdStream vgd_stream_of(const CloudView &cv, const pcp_mls_params *p, const GridDesc &g, const VoxelDesc &v) {
  VgdStream S{};
  S.v = v;
  S.g = g;
  S.remap = cv.remap;
  S.plane = (static_cast<size_t>(cv.n) + 3) & ~size_t(3);
  S.order_poly = p->polynomial_order;
  S.bricks = 0;
  return S;
}

// counts the dilated voxel set in the form that fits -- bricks unless PCP_VGD_DENSE=1 or the grid is taller than the brick
// form handles -- and cuts it into chunks of at most `capacity` voxels (capacity <= 0: one chunk)
static int vgd_count_and_plan(pcp_context *ctx, const CloudView &cv, const pcp_mls_params *p, const GridDesc &g, int64_t capacity,
                              VgdStream *out_S, std::vector<int64_t> *chunks, unsigned long long *out_total) {
  VoxelDesc v{};
  int rc = vgd_describe(ctx, cv, p, &v);
  if (rc != PCP_OK) return rc;
  chunks->clear();
  if (!vgd_dense_form()) {
    std::vector<unsigned long long> planes;
    bool fits = false;
    if ((rc = vgd_prepare_bricks(ctx, cv, v, &planes, out_total, &fits)) != PCP_OK) return rc;
    if (fits) {
      VgdStream S = vgd_stream_of(cv, p, g, v);
      const BrickDesc B = brick_desc(ctx, v);
      S.bricks = 1;
      S.NBX = B.NBX;
      S.NBY = B.NBY;
      S.NBZ = B.NBZ;
      *out_S = S;
      const int64_t cap = capacity > 0 ? capacity : std::numeric_limits<int64_t>::max();
      if ((rc = vgd_plan_bricks(ctx, v, B.NBY, planes, cap, chunks)) != PCP_OK) return rc;
      // a strip holds up to 16 NZ voxels and is never split: where one alone exceeds the capacity (a solid block of points
      // and a capacity near the 32 768 minimum) the dense form, whose unit is 32 768 keys, takes over
      bool ok = true;
      for (size_t k = 2; k < chunks->size(); k += 3) ok = ok && (*chunks)[k] <= cap;
      if (ok) return PCP_OK;
      chunks->clear();
    }
  }
  std::vector<int32_t> tile_counts;
  if ((rc = vgd_prepare(ctx, cv, p, &v, &tile_counts, out_total)) != PCP_OK) return rc;
  *out_S = vgd_stream_of(cv, p, g, v);
  // chunks = maximal runs of whole tiles that hold at most `capacity` voxels (a tile alone holds <= 32 768)
  const int64_t tiles = static_cast<int64_t>(tile_counts.size());
  const int64_t cap = capacity > 0 ? capacity : std::numeric_limits<int64_t>::max();
  int64_t t0 = 0, cnt = 0;
  for (int64_t t = 0; t <= tiles; ++t) {
    const int64_t c = t < tiles ? tile_counts[static_cast<size_t>(t)] : 0;
    if (t == tiles || (cnt + c > cap && t > t0)) {
      if (cnt > 0) {
        chunks->push_back(t0 * kScanTile);
        chunks->push_back(std::min<int64_t>(t * kScanTile, v.words));
        chunks->push_back(cnt);
      }
      t0 = t;
      cnt = 0;
    }
    cnt += c;
  }
  return PCP_OK;
}

static int voxel_grid_dilation(pcp_context *ctx, const CloudView &cv, const pcp_mls_params *p, const GridDesc &g,
                               int64_t *out_count) {
  VgdStream S{};
  std::vector<int64_t> chunks;
  unsigned long long total = 0;
  int rc = vgd_count_and_plan(ctx, cv, p, g, 0, &S, &chunks, &total);
  if (rc != PCP_OK) return rc;
  if (total >= (1ull << 31))
    return set_error(ctx, PCP_ERR_NOMEM, "pcp_mls_process: %llu dilated voxels exceed the 2^31 points one result holds "
                     "(pcp_mls_stream_begin / _next emit them in chunks)", total);
  int64_t m = 0;
  if (chunks.empty()) {  // an empty set: the result buffers still have to exist
    if ((rc = vgd_emit(ctx, S, 0, 0, 0, &m)) != PCP_OK) return rc;
  } else if ((rc = vgd_emit(ctx, S, chunks[0], chunks[1], chunks[2], &m)) != PCP_OK) {
    return rc;
  }
  if (out_count) *out_count = m;
  return PCP_OK;
}

// pcl::StatisticalOutlierRemoval / MovingLeastSquares skip non-finite points one by one; this library refuses the cloud
// (the reference's maps come out of a PCD file of a LiDAR odometry: finite)
static int require_finite_cloud(pcp_context *ctx, const char *who) {
  if (ctx->nonfinite_points > 0)
    return set_error(ctx, PCP_ERR_INVALID, "%s: %lld uploaded points have a NaN or infinite coordinate", who,
                     (long long)ctx->nonfinite_points);
  return PCP_OK;
}

static CloudView uploaded_view(const pcp_context *ctx) {
  CloudView cv{};
  const size_t plane = (static_cast<size_t>(ctx->n) + 3) & ~size_t(3);
  cv.x = ctx->sxyz.p;
  cv.y = ctx->sxyz.p + plane;
  cv.z = ctx->sxyz.p + 2 * plane;
  cv.remap = ctx->perm.p;
  cv.n = ctx->n;
  for (int a = 0; a < 3; ++a) {
    cv.mn[a] = ctx->host_min[static_cast<size_t>(a)];
    cv.mx[a] = ctx->host_max[static_cast<size_t>(a)];
  }
  return cv;
}

static int check_mls_params(pcp_context *ctx, const pcp_mls_params *p) {
  if (!p) return set_error(ctx, PCP_ERR_INVALID, "pcp_mls: NULL params");
  // MovingLeastSquares::process refuses these (mls.hpp) [upstream]
  if (!(p->search_radius > 0.0) || !(p->sqr_gauss_param > 0.0))
    return set_error(ctx, PCP_ERR_INVALID, "pcp_mls: search_radius and sqr_gauss_param must be > 0");
  if (p->polynomial_order < 0 || p->polynomial_order > 2)
    return set_error(ctx, PCP_ERR_INVALID, "pcp_mls: polynomial_order %d unsupported (0..2; the reference uses 2)",
                     p->polynomial_order);
  if (p->upsampling != 0 && p->upsampling != 3)
    return set_error(ctx, PCP_ERR_INVALID, "pcp_mls: upsampling %d unsupported (0 NONE, 3 VOXEL_GRID_DILATION)",
                     p->upsampling);
  if (p->upsampling == 3 && (!(p->vgd_voxel_size > 0.0f) || p->vgd_iterations < 0 || p->vgd_iterations > 15))
    return set_error(ctx, PCP_ERR_INVALID, "pcp_mls: vgd_voxel_size must be > 0 and vgd_iterations in 0..15");
  return PCP_OK;
}

// the grid the emission of the dilated voxels searches the nearest input point in (see mls_run); leaves *g alone when the
// fit's own grid serves (PCP_VGD_GRID=fit, or cells of 2 dmax would not be finer)
static int emission_grid(pcp_context *ctx, const CloudView &cv, const pcp_mls_params *p, GridDesc *g) {
  const char *ge = std::getenv("PCP_VGD_GRID");
  if (ge && ge[0] == 'f') return PCP_OK;
  const float dmax = static_cast<float>(1.7321 * (p->vgd_iterations + 1) * static_cast<double>(p->vgd_voxel_size) * 1.01 + 1e-6);
  const float cell_emit = 2.0f * dmax;  // measured at C3: 439 ms of emission (1.5 dmax: 449, 3: 474, 1: 604; the fit's grid: 543)
  if (cell_emit < static_cast<float>(p->search_radius)) return build_grid(ctx, cv, cell_emit, cell_emit, g);
  return PCP_OK;
}

// MovingLeastSquares::process on a cloud view; results in ctx->mls_* (index = view index)
// keep_rows: leave the fitted rows in ctx->m_tmp (7 floats at the view index mls_index names) instead of gathering them
// into the result arrays -- pcp_cloud_smooth picks the survivors of its last filter straight from there
static int mls_run(pcp_context *ctx, const CloudView &cv, const pcp_mls_params *p, int64_t *out_count,
                   int64_t q_begin = 0, int64_t q_end = -1, bool keep_rows = false, int64_t stream_capacity = 0,
                   int32_t slab = 0, int32_t n_slabs = 1) {
  const int64_t n = cv.n;
  ctx->mls_count = 0;
  ctx->vgd_next = -1;  // a stream of an earlier call rests on the grid and the fits this call replaces
  ctx->css_next = -1;
  if (out_count) *out_count = 0;
  if (n == 0) return PCP_OK;
  const size_t sn = static_cast<size_t>(n);
  const size_t plane = (sn + 3) & ~size_t(3);
  GridDesc g;
  // cell edge 0.1 % above r: two points closer than r then differ by < 1 in every cell
  // coordinate even with the fp32 slop of the cell assignment, so reach 1 suffices
  int rc = build_grid(ctx, cv, static_cast<float>(p->search_radius) * 1.001f, static_cast<float>(p->search_radius), &g);
  if (rc != PCP_OK) return rc;
  PCP_HIP_TRY(ctx, ctx->m_tmp.ensure(kRowStride * sn + 8));
  PCP_HIP_TRY(ctx, ctx->m_flag.ensure(sn + 16));
  MlsArgs a{};
  a.sx = ctx->g_xyz.p;
  a.sy = ctx->g_xyz.p + plane;
  a.sz = ctx->g_xyz.p + 2 * plane;
  a.order = ctx->g_order.p;
  a.remap = cv.remap;
  a.start = ctx->g_start.p;
  a.n = n;
  a.g = g;
  a.sq_radius = static_cast<float>(p->search_radius * p->search_radius);
  a.inv_sq_radius = 1.0 / (p->search_radius * p->search_radius);
  a.order_poly = p->polynomial_order;
  a.tmp = ctx->m_tmp.p;
  a.flag = ctx->m_flag.p;
  a.state = nullptr;
  a.q_begin = static_cast<int32_t>(q_begin);
  a.q_end = static_cast<int32_t>(q_end < 0 ? n : q_end);
  // a slab of the cell order (whole wavefronts: the deal divides the work whatever the caller's point order is)
  a.j_begin = n_slabs > 1 ? (n * slab / n_slabs) / kFitBlock * kFitBlock : 0;
  a.j_end = (n_slabs > 1 && slab + 1 < n_slabs) ? (n * (slab + 1) / n_slabs) / kFitBlock * kFitBlock : n;
  if (n_slabs > 1) PCP_HIP_TRY(ctx, hipMemsetAsync(ctx->m_flag.p, 0, sn, ctx->stream));  // the other slabs' points: no output
  if (p->upsampling == 3) {
    PCP_HIP_TRY(ctx, ctx->m_state.ensure(static_cast<size_t>(kMlsState) * sn + 8));
    // points skipped by the fit (< 3 neighbours) must read as "invalid" later
    PCP_HIP_TRY(ctx, hipMemsetAsync(ctx->m_state.p, 0, static_cast<size_t>(kMlsState) * sn * sizeof(double), ctx->stream));
    a.state = ctx->m_state.p;
  }
  {
    LaunchTimer t(ctx, PCP_K_MLS_FIT);
    const uint32_t fit_blocks = static_cast<uint32_t>(div_up(a.j_end - a.j_begin, kFitBlock));
    if (fit_blocks > 0 && n < (int64_t(1) << 30))
      hipLaunchKernelGGL(k_mls_fit<true>, dim3(fit_blocks), dim3(kFitBlock), 0, ctx->stream, a);
    else if (fit_blocks > 0)
      hipLaunchKernelGGL(k_mls_fit<false>, dim3(fit_blocks), dim3(kFitBlock), 0, ctx->stream, a);
    PCP_HIP_TRY(ctx, hipGetLastError());
  }
  if (p->upsampling == 3) {
    // The emission looks for every dilated voxel's nearest input point inside the box p +- dmax (dmax = the reach of the
    // dilation, < 9 mm for the reference's 1 mm x 4): the fit's own grid (3 cm cells) makes that ~85 candidates per voxel,
    // a grid with cells of 2 dmax ~25 -- the fit is done (its results sit under the input indices), so the cell tables are
    // rebuilt for the emission.  PCP_VGD_GRID=fit keeps the fit's grid (results identical).
    if ((rc = emission_grid(ctx, cv, p, &g)) != PCP_OK) return rc;
  }
  if (p->upsampling == 3 && stream_capacity > 0) {
    // pcp_mls_stream_begin: count the voxel set and cut its key range into chunks (whole strips of the brick form, whole
    // tiles of the dense one)
    VgdStream S{};
    unsigned long long total = 0;
    if ((rc = vgd_count_and_plan(ctx, cv, p, g, stream_capacity, &S, &ctx->vgd_chunks, &total)) != PCP_OK) return rc;
    ctx->vgd_blob.assign(reinterpret_cast<const uint8_t *>(&S), reinterpret_cast<const uint8_t *>(&S) + sizeof(S));
    ctx->vgd_next = 0;
    if (out_count) *out_count = static_cast<int64_t>(total);
    return PCP_OK;
  }
  if (p->upsampling == 3) return voxel_grid_dilation(ctx, cv, p, g, out_count);
  // points with < 3 neighbours are dropped; output keeps the input order
  PCP_HIP_TRY(ctx, ctx->mls_index.ensure(sn + 4));
  int64_t m = 0;
  if ((rc = compact_flags(ctx, ctx->m_flag.p, n, ctx->mls_index.p, n, &m)) != PCP_OK) return rc;
  PCP_HIP_TRY(ctx, ctx->mls_xyz.ensure(3 * static_cast<size_t>(m) + 4));
  PCP_HIP_TRY(ctx, ctx->mls_normal.ensure(3 * static_cast<size_t>(m) + 4));
  PCP_HIP_TRY(ctx, ctx->mls_curv.ensure(static_cast<size_t>(m) + 4));
  if (m > 0 && !keep_rows) {
    LaunchTimer t(ctx, PCP_K_MLS_FIT);
    hipLaunchKernelGGL(k_mls_gather, dim3(blocks_of(m)), dim3(kMB), 0, ctx->stream, ctx->m_tmp.p, ctx->mls_index.p, m,
                       ctx->mls_xyz.p, ctx->mls_normal.p, ctx->mls_curv.p);
    PCP_HIP_TRY(ctx, hipGetLastError());
  }
  ctx->mls_count = m;
  if (out_count) *out_count = m;
  return PCP_OK;
}

// StatisticalOutlierRemoval on a cloud view: keep flags in ctx->m_flag (view order)
// view_order: distances and keep flags under the view's own indices (ctx->m_flag[view index]) instead of the caller's
// Slabs (pcp_sor_partial / pcp_sor_finish): the places [chunk c0, chunk c1) of the cell order are this GPU's queries (see
// k_sor_stats); `classify` false stops after the chunk sums (they sit in ctx->m_sums from double 4 on).
static int sor_classify(pcp_context *ctx, int64_t n, const int32_t *remap, double std_mul, int64_t j_begin, int64_t j_end);
// clustered: the cloud is an upsampled one (VOXEL_GRID_DILATION): the selection takes its shape for clustered distances
// area_density > 0: points per unit area of the cloud's surface, known from how the cloud was made (the upsampled clouds: at
// least 2 it + 1 voxels of a column per vs^2) -- the ball is sized from it instead of from the density probe
static int sor_run(pcp_context *ctx, const CloudView &cv, int32_t mean_k, double std_mul, bool view_order = false,
                   int32_t slab = 0, int32_t n_slabs = 1, bool classify = true, float *kth = nullptr, bool clustered = false,
                   double area_density = 0.0, int64_t row_begin = 0, int64_t row_end = -1) {
  if (const char *ce = std::getenv("PCP_SOR_CLUSTERED")) clustered = ce[0] == '1';  // (tests: either shape on any cloud)
  const int32_t *remap = view_order ? nullptr : cv.remap;
  const int64_t n = cv.n;
  if (n == 0) return PCP_OK;
  ctx->vgd_next = -1;  // (pcp_mls_stream_next searches the grid this call rebuilds; pcp_cloud_smooth_stream_* rebuild theirs per chunk)
  const int64_t n_chunks = div_up(n, kSorChunk);
  const int64_t c0 = n_chunks * slab / n_slabs, c1 = n_chunks * (slab + 1) / n_slabs;
  const int64_t q_begin = c0 * kSorChunk, q_end = std::min<int64_t>(c1 * kSorChunk, n);  // places of the cell order
  const bool whole = n_slabs == 1;
  const size_t sn = static_cast<size_t>(n);
  const size_t plane = (sn + 3) & ~size_t(3);
  // cell edge: first a volume-based guess, then refined from the number of occupied cells so that
  // a 3x3x3 neighbourhood of a surface patch holds ~2.5 (k + 1) points
  const double vol = std::max<double>(cv.mx[0] - cv.mn[0], 1e-3) * std::max<double>(cv.mx[1] - cv.mn[1], 1e-3) *
                     std::max<double>(cv.mx[2] - cv.mn[2], 1e-3);
  float cell = static_cast<float>(std::cbrt(vol / static_cast<double>(n) * 4.0));
  if (!(cell > 1e-4f)) cell = 1e-4f;
  GridDesc g;
  int rc = PCP_OK;
  if (area_density > 0.0) {
    // The density is known.  (The probe below reads the upsampled clouds 3.5x too thin -- their ball then held ~320 points
    // instead of ~90 and the outlier removal of the reference's 425 M-row cloud took 566 ms; sized from the voxel structure:
    // profiles/r05_vgd_sor_ball.log.)
    // (the density handed in is a lower bound -- tilted and noisy sheets carry more voxels per area --: 1.2 (k + 1) points by
    // the bound leave 0.05 % of the 425 M rows to the wavefront kernel; 1.5: 414 ms, 1.2: 385, 1.0: 354, 0.8: 331)
    // (the streamed chain moves it from chunk to chunk by the share of rows the selection flagged: ctx->css_ball)
    double ball = ctx->css_building && ctx->css_ball > 0.0 ? ctx->css_ball : 1.2;
    if (const char *e = std::getenv("PCP_SOR_BALL")) ball = atof(e);
    double final_cell = std::sqrt(ball * (mean_k + 1) / (3.14159265358979 * area_density));
    if (!(final_cell > 1e-7) || !(final_cell < 1e30)) final_cell = static_cast<double>(cell);
    // (cells of 1 / PCP_SOR_REACH of the ball's radius: as below)
    int sel_reach = 1;
    if (const char *e = std::getenv("PCP_SOR_REACH")) sel_reach = std::max(1, std::min(4, atoi(e)));
    rc = build_grid(ctx, cv, static_cast<float>(final_cell / sel_reach), static_cast<float>(final_cell * 0.9999), &g);
    if (rc != PCP_OK) return rc;
  } else {
    // density probe on every 8th point, every 32nd of a large cloud (cell edge from the sub-sample's own volume guess):
    // occupied cells -> points per unit area of the surface.  It only sizes the grid; the kNN result does not depend on it.
    // (One returning atomic per probed point: 150-180 us for every 8th of 10 M points, twice per smoothing chain.)
    const int64_t stride = n >= 3200000 ? 32 : (n >= 400000 ? 8 : 1);
    const int64_t probe_n = div_up(n, stride);
    float pcell = static_cast<float>(std::cbrt(vol / static_cast<double>(probe_n) * 4.0));
    if (!(pcell > 1e-4f)) pcell = 1e-4f;
    if ((rc = build_grid(ctx, cv, pcell, pcell, &g, /*geometry_only=*/true)) != PCP_OK) return rc;
    const int64_t ncell = static_cast<int64_t>(g.nx) * g.ny * g.nz;
    PCP_HIP_TRY(ctx, ctx->s_counter.ensure(4));
    PCP_HIP_TRY(ctx, hipMemsetAsync(ctx->s_counter.p, 0, 8, ctx->stream));
    PCP_HIP_TRY(ctx, hipMemsetAsync(ctx->g_start.p, 0, (static_cast<size_t>(ncell) + 8) * 4, ctx->stream));
    {
      LaunchTimer t(ctx, PCP_K_MLS_GRID);
      hipLaunchKernelGGL(k_grid_probe, dim3(blocks_of(probe_n)), dim3(kMB), 0, ctx->stream, cv.x, cv.y, cv.z, n, stride, g,
                         ctx->g_start.p, ctx->s_counter.p);
      PCP_HIP_TRY(ctx, hipGetLastError());
    }
    unsigned long long occ = 0;
    {
      unsigned long long *dst = ctx->readback ? static_cast<unsigned long long *>(ctx->readback) : &occ;  // pinned: no staging
      PCP_HIP_TRY(ctx, hipMemcpyAsync(dst, ctx->s_counter.p, 8, hipMemcpyDeviceToHost, ctx->stream));
      PCP_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
      occ = *dst;
    }
    const double c0 = 1.0 / g.inv_cell;
    double final_cell = static_cast<double>(cell);
    if (occ > 0) {
      const double per_area = static_cast<double>(probe_n) * static_cast<double>(stride) /
                              (static_cast<double>(occ) * c0 * c0);  // points per unit area
      // cell edge such that the ball of one cell radius holds ~1.5 (k + 1) points of a surface of this density (a flat
      // optimum: 1.35 .. 1.8 within 2 %; smaller balls flag more points for k_sor_wave, larger ones test more candidates)
      double ball = 1.5;
      if (const char *e = std::getenv("PCP_SOR_BALL")) ball = atof(e);
      const double want = std::sqrt(ball * (mean_k + 1) / (3.14159265358979 * per_area));
      if (want > 0.0 && want < 1e30) final_cell = want;
    }
    // final_cell is the radius of the selection's ball; the cells are 1 / PCP_SOR_REACH of it.  Measured at C3: reach 1
    // (27 cells, 9 runs per lane) sor 7.33 ms / grids 0.99 ms; reach 2 (125 half-size cells, 25 runs, 31 % fewer candidates
    // on a surface) 7.64 / 1.65 ms; reach 3 9.88 / 3.01 ms -- the runs' fixed costs and ragged ends outweigh the candidates saved
    int sel_reach = 1;
    if (const char *e = std::getenv("PCP_SOR_REACH")) sel_reach = std::max(1, std::min(4, atoi(e)));
    // (0.9999: reach = ceil(radius / cell) must not round up to sel_reach + 1)
    rc = build_grid(ctx, cv, static_cast<float>(final_cell / sel_reach), static_cast<float>(final_cell * 0.9999), &g);
    if (rc != PCP_OK) return rc;
  }
  PCP_HIP_TRY(ctx, ctx->s_dist.ensure(sn + 8));
  PCP_HIP_TRY(ctx, ctx->m_flag.ensure(sn + 16));
  const int64_t chunks = div_up(n, kSorChunk);
  PCP_HIP_TRY(ctx, ctx->m_sums.ensure(4 + 2 * static_cast<size_t>(chunks)));
  float *dist = ctx->s_dist.p;
  const size_t heap_lds = static_cast<size_t>(mean_k + 1) * kSorBlock * sizeof(float);
  const char *heap_only = std::getenv("PCP_SOR_HEAP_ONLY");
  // the selection kernel addresses the coordinate planes through buffer descriptors (32-bit byte offsets)
  // (the three planes of ctx->g_xyz are one allocation which one buffer descriptor must span: fewer than 2^32 bytes)
  const bool use_select = (!(heap_only && heap_only[0] == '1') || !whole) && n < (int64_t(1) << 30);
  // (PCP_SOR_THREE_DESCRIPTORS=1: the form of the large clouds on any cloud -- the tests compare the two)
  const char *three = std::getenv("PCP_SOR_THREE_DESCRIPTORS");
  const bool one_descriptor = 2 * static_cast<int64_t>(plane) + n < (int64_t(1) << 30) && !(three && three[0] == '1');
  if (!whole && !(use_select && mean_k + 1 <= 250))
    return set_error(ctx, PCP_ERR_INVALID, "pcp_sor_partial: slabs need mean_k <= 249 and fewer than 2^30 points");
  if (use_select && mean_k + 1 <= 250) {
    // selection kernel for every point, heap kernel for the few it flags (sparse spots, borders of a surface)
    PCP_HIP_TRY(ctx, ctx->s_cell.ensure(sn + 8));
    if (!whole) PCP_HIP_TRY(ctx, hipMemsetAsync(ctx->m_flag.p, 0, sn, ctx->stream));  // nothing to redo outside the slab
    if (q_end > q_begin) {
      LaunchTimer t(ctx, PCP_K_SOR);
      auto sel = one_descriptor ? k_sor_select<true, 32, 16> : k_sor_select<false, 32, 16>;
      if (clustered) sel = one_descriptor ? k_sor_select<true, 64, 32> : k_sor_select<false, 64, 32>;
      hipLaunchKernelGGL(sel, dim3(static_cast<uint32_t>(div_up(q_end - q_begin, kSelWave))), dim3(kSelWave), 0, ctx->stream,
                         ctx->g_xyz.p, ctx->g_xyz.p + plane, ctx->g_xyz.p + 2 * plane, ctx->g_order.p, remap,
                         ctx->g_start.p, n, g, mean_k, dist, ctx->m_flag.p, q_begin, q_end, kth, static_cast<int32_t>(row_begin),
                         static_cast<int32_t>(row_end < 0 ? n : row_end));
      PCP_HIP_TRY(ctx, hipGetLastError());
    }
    int64_t redo = 0;
    if ((rc = compact_flags(ctx, ctx->m_flag.p, n, ctx->s_cell.p, n, &redo)) != PCP_OK) return rc;
    ctx->sor_redo_fraction = static_cast<double>(redo) / static_cast<double>(n);
    if (redo > 0) {
      // one wavefront per flagged point; the heap kernel (one lane per point) for the few whose block of cells
      // overflows the wavefront's cache
      LaunchTimer t(ctx, PCP_K_SOR);
      const char *no_wave = std::getenv("PCP_SOR_NO_WAVE");
      const bool wave = !(no_wave && no_wave[0] == '1');
      // PCP_SOR_WAVE_STATS=1: what the flagged points cost, by the number of block passes each needed (stderr, one line per run)
      static const bool wave_stats = [] { const char *e = std::getenv("PCP_SOR_WAVE_STATS"); return e && e[0] == '1'; }();
      unsigned long long *tally = nullptr;
      if (wave && wave_stats) {
        PCP_HIP_TRY(ctx, ctx->s_u32.ensure(2 * 2 * 48 + 8));
        tally = reinterpret_cast<unsigned long long *>(ctx->s_u32.p);
        PCP_HIP_TRY(ctx, hipMemsetAsync(tally, 0, 2 * 48 * 8, ctx->stream));
      }
      if (wave)
        hipLaunchKernelGGL(k_sor_wave, dim3(static_cast<uint32_t>(redo)), dim3(kSelWave), 0, ctx->stream, ctx->g_xyz.p,
                           ctx->g_xyz.p + plane, ctx->g_xyz.p + 2 * plane, ctx->g_order.p, remap, ctx->g_start.p, n, g, mean_k,
                           dist, ctx->m_flag.p, ctx->s_cell.p, redo, kth, tally);
      if (tally) {
        unsigned long long h[2 * 48];
        PCP_HIP_TRY(ctx, hipMemcpyAsync(h, tally, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
        PCP_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        std::fprintf(stderr, "[pcp] k_sor_wave n=%lld flagged=%lld cell=%.5f reach=%d:", (long long)n, (long long)redo, 1.0 / g.inv_cell, g.reach);
        for (int k = 0; k < 48; ++k)
          if (h[2 * k]) std::fprintf(stderr, " %s%s%d:%llux%.1fus", k >= 32 ? "o" : "", (k & 16) ? "f" : "", k & 15, h[2 * k],
                                     static_cast<double>(h[2 * k + 1]) / static_cast<double>(h[2 * k]) * 0.01);
        std::fprintf(stderr, "\n");
      }
      hipLaunchKernelGGL(k_sor_mean_distance, dim3(static_cast<uint32_t>(div_up(redo, kSorBlock))), dim3(kSorBlock), heap_lds,
                         ctx->stream, ctx->g_xyz.p, ctx->g_xyz.p + plane, ctx->g_xyz.p + 2 * plane, ctx->g_order.p,
                         remap, ctx->g_start.p, n, g, mean_k, dist, ctx->s_cell.p, redo,
                         wave ? ctx->m_flag.p : static_cast<const uint8_t *>(nullptr), kth);
      PCP_HIP_TRY(ctx, hipGetLastError());
    }
  } else {
    ctx->sor_redo_fraction = 0.0;
    LaunchTimer t(ctx, PCP_K_SOR);
    hipLaunchKernelGGL(k_sor_mean_distance, dim3(static_cast<uint32_t>(div_up(n, kSorBlock))), dim3(kSorBlock), heap_lds,
                       ctx->stream, ctx->g_xyz.p, ctx->g_xyz.p + plane, ctx->g_xyz.p + 2 * plane, ctx->g_order.p,
                       remap, ctx->g_start.p, n, g, mean_k, dist, static_cast<const int32_t *>(nullptr), int64_t(0),
                       static_cast<const uint8_t *>(nullptr), kth);
    PCP_HIP_TRY(ctx, hipGetLastError());
  }
  {
    // the chunks of this GPU's slab
    LaunchTimer t(ctx, PCP_K_SOR);
    if (c1 > c0)
      hipLaunchKernelGGL(k_sor_stats, dim3(static_cast<uint32_t>(c1 - c0)), dim3(kMB), 0, ctx->stream, dist, ctx->g_order.p,
                         remap, n, c0, ctx->m_sums.p + 4);
    PCP_HIP_TRY(ctx, hipGetLastError());
  }
  return classify ? sor_classify(ctx, n, remap, std_mul, q_begin, q_end) : PCP_OK;
}

// threshold from the chunk sums of the WHOLE cloud (ctx->m_sums), keep flags (ctx->m_flag, zero elsewhere) of the points
// at places [j_begin, j_end) of the cell order (ctx->g_order: the grid of the sor_run that left the distances)
static int sor_classify(pcp_context *ctx, int64_t n, const int32_t *remap, double std_mul, int64_t j_begin, int64_t j_end) {
  LaunchTimer t(ctx, PCP_K_SOR);
  double *partial = ctx->m_sums.p + 4, *threshold = ctx->m_sums.p;
  hipLaunchKernelGGL(k_sor_threshold, dim3(1), dim3(64), 0, ctx->stream, partial, div_up(n, kSorChunk), n, std_mul, threshold);
  if (j_begin > 0 || j_end < n) PCP_HIP_TRY(ctx, hipMemsetAsync(ctx->m_flag.p, 0, static_cast<size_t>(n), ctx->stream));
  if (j_end > j_begin)
    hipLaunchKernelGGL(k_sor_classify, dim3(blocks_of(j_end - j_begin)), dim3(kMB), 0, ctx->stream, ctx->s_dist.p, ctx->g_order.p,
                       remap, j_begin, j_end, threshold, ctx->m_flag.p);
  PCP_HIP_TRY(ctx, hipGetLastError());
  return PCP_OK;
}


// ---- pcp_cloud_smooth_stream_*: the trailing StatisticalOutlierRemoval over the chunked voxel dilation ----

// (sum, sum of fp32 squares) of d[0, n) per chunk of kSorChunk consecutive ROWS, the arithmetic and the tree of k_sor_stats
__global__ __launch_bounds__(kMB) void k_rows_stats(const float *__restrict__ d, int64_t n, double *__restrict__ partial) {
  __shared__ double sh[2][kMB / 64];
  const int64_t chunk = blockIdx.x;
  const int64_t lo = chunk * kSorChunk, hi = min(lo + kSorChunk, n);
  double s = 0.0, q = 0.0;
  for (int64_t j = lo + threadIdx.x; j < hi; j += kMB) {
    const float v = d[j];
    s += static_cast<double>(v);
    q += static_cast<double>(__fmul_rn(v, v));
  }
  for (int o = 32; o >= 1; o >>= 1) {
    s += __shfl_xor(s, o, 64);
    q += __shfl_xor(q, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    sh[0][threadIdx.x >> 6] = s;
    sh[1][threadIdx.x >> 6] = q;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ts = 0.0, tq = 0.0;
    for (int k = 0; k < kMB / 64; ++k) {
      ts += sh[0][k];
      tq += sh[1][k];
    }
    partial[2 * chunk] = ts;
    partial[2 * chunk + 1] = tq;
  }
}

// keep flags of rows [0, m) by their stored distance (the rule of k_sor_classify); optionally only counts the survivors
__global__ __launch_bounds__(kMB) void k_rows_classify(const float *__restrict__ d, int64_t m, const double *__restrict__ threshold,
                                                       uint8_t *__restrict__ keep, unsigned long long *__restrict__ count) {
  int64_t i = static_cast<int64_t>(blockIdx.x) * kMB + threadIdx.x;
  unsigned long long mine = 0;
  for (; i < m; i += static_cast<int64_t>(gridDim.x) * kMB) {
    const bool k = !(static_cast<double>(d[i]) > *threshold);
    if (keep) keep[i] = k ? 1 : 0;
    mine += k ? 1ull : 0ull;
  }
  if (count) {
    for (int o = 32; o >= 1; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(count, mine);
  }
}

// order-preserving map of a float onto unsigned integers (negative margins included)
__device__ __forceinline__ uint32_t float_key(float f) {
  const uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
inline float float_of_key(uint32_t k) {
  const uint32_t b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  float f;
  std::memcpy(&f, &b, 4);
  return f;
}

// The smallest margin of a chunk's own rows [r0, r1) of the emitted cloud (xyz interleaved): how far the ball of a row --
// centre x, radius sqrt(kth) -- stays from the first plane the emission did NOT hold on either side (x_lo / x_hi = the voxel
// positions of those planes; use_lo / use_hi: there is such a plane).  A row of a missing plane lies within `max displacement`
// of its voxel position, so a margin above that proves every neighbour of every row was present.
__global__ __launch_bounds__(kMB) void k_rows_margin(const float *__restrict__ xyz, const float *__restrict__ kth, int64_t r0, int64_t r1,
                                                     double x_lo, double x_hi, int use_lo, int use_hi, uint32_t *__restrict__ min_key) {
  float m = INFINITY;
  for (int64_t i = r0 + static_cast<int64_t>(blockIdx.x) * kMB + threadIdx.x; i < r1; i += static_cast<int64_t>(gridDim.x) * kMB) {
    const double x = static_cast<double>(xyz[3 * i]);
    const double r = sqrt(static_cast<double>(kth[i])) * (1.0 + 1.0e-6);
    double mm = INFINITY;
    if (use_lo) mm = fmin(mm, (x - r) - x_lo);
    if (use_hi) mm = fmin(mm, x_hi - (x + r));
    if (!(mm == mm)) mm = -INFINITY;  // NaN: no proof
    m = fminf(m, __double2float_rd(mm));
  }
  uint32_t k = float_key(m);
  for (int o = 32; o >= 1; o >>= 1) k = min(k, static_cast<uint32_t>(__shfl_xor(static_cast<int>(k), o, 64)));
  if ((threadIdx.x & 63) == 0) atomicMin(min_key, k);
}

struct SmoothStream {  // plain data, kept in ctx->css_blob between pcp_cloud_smooth_stream_begin and _next
  pcp_mls_params p;
  VgdStream S;
  CloudView cv1;  // the survivors of the first filter (planes in ctx->c_xyz)
  int64_t total_rows, kept_rows, rows_computed;
  double threshold, max_dx, min_margin;
  int32_t halo, redone;
  double sampled_dx;  // the largest displacement sweep 0 saw on its sample of the voxels (sizes the halo; max_dx proves it)
  double seconds[4];  // host clock of _begin: first filter + fit + voxel set, device allocations (inside the other three), sweep 0, sweep 1 + threshold
  double alloc_bytes;  // device memory allocated during _begin
};

// the grid the emission searches, rebuilt (the outlier removal of a chunk overwrites it): the same call sequence as mls_run's
static int stream_grid(pcp_context *ctx, const CloudView &cv, const pcp_mls_params *p, GridDesc *g) {
  const char *ge = std::getenv("PCP_VGD_GRID");
  const float dmax = static_cast<float>(1.7321 * (p->vgd_iterations + 1) * static_cast<double>(p->vgd_voxel_size) * 1.01 + 1e-6);
  const float cell_emit = 2.0f * dmax;
  if (!(ge && ge[0] == 'f') && cell_emit < static_cast<float>(p->search_radius)) return build_grid(ctx, cv, cell_emit, cell_emit, g);
  return build_grid(ctx, cv, static_cast<float>(p->search_radius) * 1.001f, static_cast<float>(p->search_radius), g);
}

// bounding box of three device planes -> view
static int view_of(pcp_context *ctx, const float *x, const float *y, const float *z, int64_t n, CloudView *cv) {
  cv->x = x;
  cv->y = y;
  cv->z = z;
  cv->remap = nullptr;
  cv->n = n;
  for (int a = 0; a < 3; ++a) cv->mn[a] = cv->mx[a] = 0.0f;
  if (n == 0) return PCP_OK;
  PCP_HIP_TRY(ctx, ctx->s_u32.ensure(6 * kBoxStride));
  uint32_t box[6 * kBoxStride];
  for (int a = 0; a < 6; ++a)
    for (int w = 0; w < kBoxStride; ++w) box[a * kBoxStride + w] = a < 3 ? 0xffffffffu : 0u;
  PCP_HIP_TRY(ctx, hipMemcpyAsync(ctx->s_u32.p, box, sizeof(box), hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_bbox, dim3(static_cast<uint32_t>(std::max<int64_t>(1, std::min<int64_t>(div_up(n, 8 * kMB), 2048)))),
                     dim3(kMB), 0, ctx->stream, x, y, z, n, ctx->s_u32.p);
  PCP_HIP_TRY(ctx, hipGetLastError());
  {
    static_assert(sizeof(box) <= pcp_context::kReadbackBytes, "readback scratch");
    uint32_t *dst = ctx->readback ? static_cast<uint32_t *>(ctx->readback) : box;  // pinned: no staging
    PCP_HIP_TRY(ctx, hipMemcpyAsync(dst, ctx->s_u32.p, sizeof(box), hipMemcpyDeviceToHost, ctx->stream));
    PCP_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (dst != box) std::memcpy(box, dst, sizeof(box));
  }
  auto decode = [](uint32_t k) {
    const uint32_t b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    float f;
    std::memcpy(&f, &b, 4);
    return f;
  };
  for (int a = 0; a < 3; ++a) {
    cv->mn[a] = decode(box[a * kBoxStride]);
    cv->mx[a] = decode(box[(3 + a) * kBoxStride]);
  }
  return PCP_OK;
}

// The first StatisticalOutlierRemoval of CloudSmooth::process (cloudSmooth.cpp:109-116) and its survivors as a new device
// cloud (planes in ctx->c_xyz; ctx->c_index[i] = the caller's index of point i of that cloud).  The intermediate clouds stay
// in the order of the view they come from (the Morton-ordered copy of the upload): their grids are then built from
// spatially ordered input (cell histogram, scatter and in-cell ordering touch neighbouring memory: -1.2 ms per chain
// against clouds in the caller's order); the caller's order comes back in the last compaction.
static int smooth_first_filter(pcp_context *ctx, const pcp_mls_params *p, const CloudView &cv0, CloudView *out_cv1, int64_t *out_n1) {
  int rc;
  *out_n1 = 0;
  if ((rc = sor_run(ctx, cv0, p->sor_mean_k, p->sor_std_mul, /*view_order=*/true)) != PCP_OK) return rc;
  PCP_HIP_TRY(ctx, ctx->c_index.ensure(2 * (static_cast<size_t>(cv0.n) + 4)));
  int32_t *c_pos = ctx->c_index.p + static_cast<size_t>(cv0.n) + 4;  // view positions of the survivors
  int64_t n1 = 0;
  if ((rc = compact_flags(ctx, ctx->m_flag.p, cv0.n, c_pos, cv0.n, &n1)) != PCP_OK) return rc;
  if (n1 == 0) return PCP_OK;
  const size_t plane1 = (static_cast<size_t>(n1) + 3) & ~size_t(3);
  PCP_HIP_TRY(ctx, ctx->c_xyz.ensure(3 * plane1 + 4));
  float *x1 = ctx->c_xyz.p, *y1 = ctx->c_xyz.p + plane1, *z1 = ctx->c_xyz.p + 2 * plane1;
  hipLaunchKernelGGL(k_gather_view, dim3(blocks_of(n1)), dim3(kMB), 0, ctx->stream, cv0.x, cv0.y, cv0.z, cv0.remap, c_pos, n1,
                     x1, y1, z1, ctx->c_index.p);  // c_index: the caller's indices of cloud 1
  PCP_HIP_TRY(ctx, hipGetLastError());
  // cloud 1 is a subset of the upload: without upsampling the upload's box serves (any enclosing box gives the same grid
  // searches); the voxel dilation counts its voxels from the cloud's own box (getMinMax3D, mls.hpp [upstream])
  CloudView cv1 = cv0;
  cv1.x = x1;
  cv1.y = y1;
  cv1.z = z1;
  cv1.n = n1;
  cv1.remap = nullptr;
  if (p->upsampling != 0 && (rc = view_of(ctx, x1, y1, z1, n1, &cv1)) != PCP_OK) return rc;
  *out_cv1 = cv1;
  *out_n1 = n1;
  return PCP_OK;
}

}  // namespace pcp

namespace pcp {
// (pcp_create loads every code object of the library up front: see preload_code_objects in pcp_context.hip)
hipError_t preload_mls() {
  hipFuncAttributes a;
  return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&k_deinterleave));
}
}  // namespace pcp

using namespace pcp;

extern "C" {

int pcp_mls_process(pcp_context *ctx, const pcp_mls_params *p, int64_t *out_count) {
  if (!ctx) return PCP_ERR_INVALID;
  int rc = check_mls_params(ctx, p);
  if (rc != PCP_OK) return rc;
  if (!ctx->xyz.p) return set_error(ctx, PCP_ERR_STATE, "pcp_mls_process: no cloud uploaded");
  if (int rcf = require_finite_cloud(ctx, "pcp_mls_process")) return rcf;
  PCP_HIP_TRY(ctx, hipSetDevice(ctx->device));
  return mls_run(ctx, uploaded_view(ctx), p, out_count);
}

int pcp_mls_stream_begin(pcp_context *ctx, const pcp_mls_params *p, int64_t chunk_capacity, int64_t *out_total,
                         int32_t *out_chunks) {
  if (!ctx) return PCP_ERR_INVALID;
  int rc = check_mls_params(ctx, p);
  if (rc != PCP_OK) return rc;
  if (p->upsampling != 3) return set_error(ctx, PCP_ERR_INVALID, "pcp_mls_stream_begin: upsampling must be VOXEL_GRID_DILATION (3)");
  if (chunk_capacity < 32768 || chunk_capacity >= (int64_t(1) << 31))
    return set_error(ctx, PCP_ERR_INVALID, "pcp_mls_stream_begin: chunk_capacity must be in [32768, 2^31)");
  if (!ctx->xyz.p) return set_error(ctx, PCP_ERR_STATE, "pcp_mls_stream_begin: no cloud uploaded");
  if (int rcf = require_finite_cloud(ctx, "pcp_mls_stream_begin")) return rcf;
  PCP_HIP_TRY(ctx, hipSetDevice(ctx->device));
  ctx->vgd_next = -1;
  int64_t total = 0;
  if ((rc = mls_run(ctx, uploaded_view(ctx), p, &total, 0, -1, false, chunk_capacity)) != PCP_OK) return rc;
  if (out_total) *out_total = total;
  if (out_chunks) *out_chunks = static_cast<int32_t>(ctx->vgd_chunks.size() / 3);
  return PCP_OK;
}

int pcp_mls_stream_next(pcp_context *ctx, int64_t *out_count) {
  if (!ctx || !out_count) return PCP_ERR_INVALID;
  *out_count = 0;
  if (ctx->vgd_next < 0 || ctx->vgd_blob.size() != sizeof(VgdStream))
    return set_error(ctx, PCP_ERR_STATE, "pcp_mls_stream_next: no stream (call pcp_mls_stream_begin)");
  PCP_HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t k = static_cast<size_t>(ctx->vgd_next) * 3;
  if (k >= ctx->vgd_chunks.size()) {  // past the last chunk
    ctx->mls_count = 0;
    return PCP_OK;
  }
  VgdStream S;
  std::memcpy(&S, ctx->vgd_blob.data(), sizeof(S));
  int64_t m = 0;
  const int rc = vgd_emit(ctx, S, ctx->vgd_chunks[k], ctx->vgd_chunks[k + 1], ctx->vgd_chunks[k + 2], &m);
  if (rc != PCP_OK) return rc;
  ctx->vgd_next += 1;
  *out_count = m;
  return PCP_OK;
}

int pcp_mls_stream_seek(pcp_context *ctx, int32_t chunk) {
  if (!ctx) return PCP_ERR_INVALID;
  if (ctx->vgd_next < 0 || ctx->vgd_blob.empty())
    return set_error(ctx, PCP_ERR_STATE, "pcp_mls_stream_seek: no stream (call pcp_mls_stream_begin)");
  const int64_t chunks = static_cast<int64_t>(ctx->vgd_chunks.size() / 3);
  if (chunk < 0 || chunk > chunks)
    return set_error(ctx, PCP_ERR_RANGE, "pcp_mls_stream_seek: chunk %d outside 0..%lld", chunk, (long long)chunks);
  ctx->vgd_next = chunk;
  return PCP_OK;
}

int pcp_mls_process_shard(pcp_context *ctx, const pcp_mls_params *p, int64_t index_begin, int64_t index_end,
                          int64_t *out_count) {
  if (!ctx) return PCP_ERR_INVALID;
  int rc = check_mls_params(ctx, p);
  if (rc != PCP_OK) return rc;
  if (!ctx->xyz.p) return set_error(ctx, PCP_ERR_STATE, "pcp_mls_process_shard: no cloud uploaded");
  if (int rcf = require_finite_cloud(ctx, "pcp_mls_process_shard")) return rcf;
  if (p->upsampling != 0)
    return set_error(ctx, PCP_ERR_INVALID, "pcp_mls_process_shard: query sharding supports upsampling NONE only");
  if (index_begin < 0 || index_end > ctx->n || index_begin > index_end)
    return set_error(ctx, PCP_ERR_RANGE, "pcp_mls_process_shard: query range [%lld,%lld) outside 0..%lld",
                     (long long)index_begin, (long long)index_end, (long long)ctx->n);
  PCP_HIP_TRY(ctx, hipSetDevice(ctx->device));
  return mls_run(ctx, uploaded_view(ctx), p, out_count, index_begin, index_end);
}

int pcp_mls_process_slab(pcp_context *ctx, const pcp_mls_params *p, int32_t slab, int32_t n_slabs, int64_t *out_count) {
  if (!ctx) return PCP_ERR_INVALID;
  int rc = check_mls_params(ctx, p);
  if (rc != PCP_OK) return rc;
  if (!ctx->xyz.p) return set_error(ctx, PCP_ERR_STATE, "pcp_mls_process_slab: no cloud uploaded");
  if (int rcf = require_finite_cloud(ctx, "pcp_mls_process_slab")) return rcf;
  if (p->upsampling != 0)
    return set_error(ctx, PCP_ERR_INVALID, "pcp_mls_process_slab: query sharding supports upsampling NONE only");
  if (n_slabs < 1 || slab < 0 || slab >= n_slabs) return set_error(ctx, PCP_ERR_RANGE, "pcp_mls_process_slab: slab %d of %d", slab, n_slabs);
  PCP_HIP_TRY(ctx, hipSetDevice(ctx->device));
  return mls_run(ctx, uploaded_view(ctx), p, out_count, 0, -1, false, 0, slab, n_slabs);
}

int pcp_mls_fetch(pcp_context *ctx, int64_t capacity, float *out_xyz, float *out_normal, float *out_curvature,
                  int32_t *out_index) {
  if (!ctx) return PCP_ERR_INVALID;
  if (capacity < 0) return set_error(ctx, PCP_ERR_INVALID, "pcp_mls_fetch: negative capacity");
  PCP_HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t m = static_cast<size_t>(std::min<int64_t>(capacity, ctx->mls_count));
  if (m > 0) {
    if (out_xyz) PCP_HIP_TRY(ctx, hipMemcpyAsync(out_xyz, ctx->mls_xyz.p, 3 * m * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (out_normal)
      PCP_HIP_TRY(ctx, hipMemcpyAsync(out_normal, ctx->mls_normal.p, 3 * m * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (out_curvature)
      PCP_HIP_TRY(ctx, hipMemcpyAsync(out_curvature, ctx->mls_curv.p, m * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (out_index) PCP_HIP_TRY(ctx, hipMemcpyAsync(out_index, ctx->mls_index.p, m * 4, hipMemcpyDeviceToHost, ctx->stream));
  }
  PCP_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return PCP_OK;
}

int pcp_sor(pcp_context *ctx, int32_t mean_k, double std_mul, uint8_t *out_keep, int64_t *out_kept) {
  if (!ctx) return PCP_ERR_INVALID;
  if (!ctx->xyz.p) return set_error(ctx, PCP_ERR_STATE, "pcp_sor: no cloud uploaded");
  if (int rcf = require_finite_cloud(ctx, "pcp_sor")) return rcf;
  if (mean_k < 1 || mean_k > 254) return set_error(ctx, PCP_ERR_INVALID, "pcp_sor: mean_k %d out of range (1..254)", mean_k);
  PCP_HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int64_t n = ctx->n;
  if (out_kept) *out_kept = 0;
  if (n == 0) return PCP_OK;
  ctx->sor_distances_live = false;
  ctx->sor_partial_slab = ctx->sor_partial_slabs = -1;
  int rc = sor_run(ctx, uploaded_view(ctx), mean_k, std_mul);
  if (rc != PCP_OK) return rc;
  ctx->sor_distances_live = true;
  if (out_kept) {
    int64_t kept = 0;
    if ((rc = compact_flags(ctx, ctx->m_flag.p, n, nullptr, 0, &kept)) != PCP_OK) return rc;
    *out_kept = kept;
  }
  if (out_keep)
    PCP_HIP_TRY(ctx, hipMemcpyAsync(out_keep, ctx->m_flag.p, static_cast<size_t>(n), hipMemcpyDeviceToHost, ctx->stream));
  PCP_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return PCP_OK;
}

int64_t pcp_sor_chunk_points(void) { return kSorChunk; }

static int check_sor_slab(pcp_context *ctx, const char *who, int32_t slab, int32_t n_slabs) {
  if (!ctx->xyz.p) return set_error(ctx, PCP_ERR_STATE, "%s: no cloud uploaded", who);
  if (n_slabs < 1 || slab < 0 || slab >= n_slabs)
    return set_error(ctx, PCP_ERR_RANGE, "%s: slab %d of %d", who, slab, n_slabs);
  return PCP_OK;
}

int pcp_sor_partial(pcp_context *ctx, int32_t mean_k, int32_t slab, int32_t n_slabs, int64_t capacity, double *out_chunk_sums,
                    int64_t *out_first_chunk, int64_t *out_chunks) {
  if (!ctx) return PCP_ERR_INVALID;
  int rc = check_sor_slab(ctx, "pcp_sor_partial", slab, n_slabs);
  if (rc != PCP_OK) return rc;
  if (int rcf = require_finite_cloud(ctx, "pcp_sor_partial")) return rcf;
  if (mean_k < 1 || mean_k > 249) return set_error(ctx, PCP_ERR_INVALID, "pcp_sor_partial: mean_k %d out of range (1..249)", mean_k);
  const int64_t n = ctx->n, n_chunks = div_up(n, kSorChunk);
  const int64_t c0 = n_chunks * slab / n_slabs, c1 = n_chunks * (slab + 1) / n_slabs;
  if (out_first_chunk) *out_first_chunk = c0;
  if (out_chunks) *out_chunks = c1 - c0;
  if (capacity < c1 - c0 || (c1 > c0 && !out_chunk_sums))
    return set_error(ctx, PCP_ERR_RANGE, "pcp_sor_partial: room for %lld chunks, the slab has %lld", (long long)capacity, (long long)(c1 - c0));
  PCP_HIP_TRY(ctx, hipSetDevice(ctx->device));
  ctx->sor_distances_live = false;
  ctx->sor_partial_slab = ctx->sor_partial_slabs = -1;
  if (n == 0) return PCP_OK;
  // (a slab without chunks still builds the grid: pcp_sor_finish reads the cell order from it)
  if ((rc = sor_run(ctx, uploaded_view(ctx), mean_k, 0.0, false, slab, n_slabs, /*classify=*/false)) != PCP_OK) return rc;
  if (c1 > c0)
    PCP_HIP_TRY(ctx, hipMemcpyAsync(out_chunk_sums, ctx->m_sums.p + 4 + 2 * c0, static_cast<size_t>(c1 - c0) * 2 * sizeof(double),
                                    hipMemcpyDefault, ctx->stream));  // host or device memory
  PCP_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  ctx->sor_partial_slab = slab;
  ctx->sor_partial_slabs = n_slabs;
  return PCP_OK;
}

int pcp_sor_finish(pcp_context *ctx, double std_mul, const double *all_chunk_sums, int64_t n_chunks, int32_t slab, int32_t n_slabs,
                   uint8_t *out_keep, int64_t *out_kept) {
  if (!ctx) return PCP_ERR_INVALID;
  int rc = check_sor_slab(ctx, "pcp_sor_finish", slab, n_slabs);
  if (rc != PCP_OK) return rc;
  const int64_t n = ctx->n;
  if (out_kept) *out_kept = 0;
  if (n == 0) return PCP_OK;
  if (n_chunks != div_up(n, kSorChunk) || !all_chunk_sums)
    return set_error(ctx, PCP_ERR_INVALID, "pcp_sor_finish: %lld chunk sums given, the cloud has %lld chunks", (long long)n_chunks, (long long)div_up(n, kSorChunk));
  if (!ctx->s_dist.p || !ctx->m_sums.p || ctx->m_sums.count < 4 + 2 * static_cast<size_t>(n_chunks) || slab != ctx->sor_partial_slab ||
      n_slabs != ctx->sor_partial_slabs)
    return set_error(ctx, PCP_ERR_STATE, "pcp_sor_finish: slab %d of %d is not the slab of the last pcp_sor_partial of this context", slab, n_slabs);
  PCP_HIP_TRY(ctx, hipSetDevice(ctx->device));
  PCP_HIP_TRY(ctx, hipMemcpyAsync(ctx->m_sums.p + 4, all_chunk_sums, static_cast<size_t>(n_chunks) * 2 * sizeof(double),
                                  hipMemcpyDefault, ctx->stream));  // host or device memory
  const int64_t c0 = n_chunks * slab / n_slabs, c1 = n_chunks * (slab + 1) / n_slabs;
  const int64_t j0 = c0 * kSorChunk, j1 = std::min<int64_t>(c1 * kSorChunk, n);
  PCP_HIP_TRY(ctx, hipMemsetAsync(ctx->m_flag.p, 0, static_cast<size_t>(n), ctx->stream));
  if ((rc = sor_classify(ctx, n, uploaded_view(ctx).remap, std_mul, j0, j1)) != PCP_OK) return rc;
  if (out_kept) {
    int64_t kept = 0;
    if ((rc = compact_flags(ctx, ctx->m_flag.p, n, nullptr, 0, &kept)) != PCP_OK) return rc;
    *out_kept = kept;
  }
  if (out_keep) PCP_HIP_TRY(ctx, hipMemcpyAsync(out_keep, ctx->m_flag.p, static_cast<size_t>(n), hipMemcpyDefault, ctx->stream));
  PCP_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return PCP_OK;
}

// CloudSmooth::process end to end on the device (cloudSmooth.cpp:109-164):
// SOR -> MovingLeastSquares (+ upsampling) -> SOR.  Results through pcp_mls_fetch;
// out_index refers to the uploaded cloud.
int pcp_cloud_smooth(pcp_context *ctx, const pcp_mls_params *p, int64_t *out_count) {
  if (!ctx) return PCP_ERR_INVALID;
  int rc = check_mls_params(ctx, p);
  if (rc != PCP_OK) return rc;
  if (!ctx->xyz.p) return set_error(ctx, PCP_ERR_STATE, "pcp_cloud_smooth: no cloud uploaded");
  if (int rcf = require_finite_cloud(ctx, "pcp_cloud_smooth")) return rcf;
  if (p->sor_mean_k < 1 || p->sor_mean_k > 254)
    return set_error(ctx, PCP_ERR_INVALID, "pcp_cloud_smooth: sor_mean_k %d out of range (1..254)", p->sor_mean_k);
  PCP_HIP_TRY(ctx, hipSetDevice(ctx->device));
  ctx->mls_count = 0;
  ctx->sor_distances_live = false;
  ctx->sor_partial_slab = ctx->sor_partial_slabs = -1;
  if (out_count) *out_count = 0;
  const CloudView cv0 = uploaded_view(ctx);
  if (cv0.n == 0) return PCP_OK;
  CloudView cv1;
  int64_t n1 = 0;
  if ((rc = smooth_first_filter(ctx, p, cv0, &cv1, &n1)) != PCP_OK) return rc;
  if (n1 == 0) return PCP_OK;
  int32_t *c_pos = ctx->c_index.p + static_cast<size_t>(cv0.n) + 4;  // (free again: reused for the rows' caller indices)
  // MLS (cloudSmooth.cpp:124-154).  Without upsampling the fitted rows stay where the fit wrote them (7 floats per point
  // of cloud 1): the second filter only needs their positions, and the survivors are picked from there at the end.
  const bool plain = p->upsampling == 0;
  int64_t m = 0;
  if ((rc = mls_run(ctx, cv1, p, &m, 0, -1, /*keep_rows=*/plain)) != PCP_OK) return rc;
  if (m == 0) return PCP_OK;
  const size_t plane2 = (static_cast<size_t>(m) + 3) & ~size_t(3);
  PCP_HIP_TRY(ctx, ctx->c_xyz2.ensure(3 * plane2 + 4));
  float *x2 = ctx->c_xyz2.p, *y2 = ctx->c_xyz2.p + plane2, *z2 = ctx->c_xyz2.p + 2 * plane2;
  int32_t *row_caller = c_pos;  // the caller's index of every fitted row (c_pos is free again)
  if (plain) {
    hipLaunchKernelGGL(k_remap_index_to, dim3(blocks_of(m)), dim3(kMB), 0, ctx->stream, ctx->mls_index.p, m, ctx->c_index.p,
                       row_caller);
    hipLaunchKernelGGL(k_rows_xyz, dim3(blocks_of(m)), dim3(kMB), 0, ctx->stream, ctx->m_tmp.p, ctx->mls_index.p, m, x2, y2, z2);
  } else {
    // source indices back to the uploaded cloud
    hipLaunchKernelGGL(k_remap_index, dim3(blocks_of(m)), dim3(kMB), 0, ctx->stream, ctx->mls_index.p, m, ctx->c_index.p);
    hipLaunchKernelGGL(k_deinterleave, dim3(blocks_of(m)), dim3(kMB), 0, ctx->stream, ctx->mls_xyz.p, m, x2, y2, z2);
  }
  PCP_HIP_TRY(ctx, hipGetLastError());
  // 2nd SOR on the MLS output (cloudSmooth.cpp:160-164)
  CloudView cv2;
  if ((rc = view_of(ctx, x2, y2, z2, m, &cv2)) != PCP_OK) return rc;
  // (the upsampled cloud: every surface patch of vs^2 carries a column of at least 2 it + 1 voxels)
  const double dens2 = plain ? 0.0 : (2.0 * p->vgd_iterations + 1.0) / (static_cast<double>(p->vgd_voxel_size) * p->vgd_voxel_size);
  if ((rc = sor_run(ctx, cv2, p->sor_mean_k, p->sor_std_mul, false, 0, 1, true, nullptr, /*clustered=*/!plain, dens2)) != PCP_OK) return rc;
  int64_t kept = 0;
  if (plain) {
    // survivors back in the caller's order (ascending index, as a filter chain on the input cloud leaves them): every
    // surviving row marks its index; the ordered compaction of the marks lists the indices, `where` names their rows
    const size_t sn0 = static_cast<size_t>(cv0.n);
    PCP_HIP_TRY(ctx, ctx->c_mark.ensure(sn0 + 16));
    PCP_HIP_TRY(ctx, ctx->c_where.ensure(sn0 + 4));
    PCP_HIP_TRY(ctx, ctx->s_cell.ensure(sn0 + 4));
    PCP_HIP_TRY(ctx, hipMemsetAsync(ctx->c_mark.p, 0, sn0, ctx->stream));
    hipLaunchKernelGGL(k_mark_rows, dim3(blocks_of(m)), dim3(kMB), 0, ctx->stream, ctx->m_flag.p, row_caller, m, ctx->c_mark.p,
                       ctx->c_where.p);
    PCP_HIP_TRY(ctx, hipGetLastError());
    if ((rc = compact_flags(ctx, ctx->c_mark.p, cv0.n, ctx->s_cell.p, cv0.n, &kept)) != PCP_OK) return rc;
    const size_t sk = static_cast<size_t>(kept);
    PCP_HIP_TRY(ctx, ctx->mls_xyz.ensure(3 * sk + 4));
    PCP_HIP_TRY(ctx, ctx->mls_normal.ensure(3 * sk + 4));
    PCP_HIP_TRY(ctx, ctx->mls_curv.ensure(sk + 4));
    PCP_HIP_TRY(ctx, ctx->mls_alt_index.ensure(std::max(sk + 4, ctx->mls_index.count)));
    if (kept > 0) {
      hipLaunchKernelGGL(k_final_rows, dim3(blocks_of(kept)), dim3(kMB), 0, ctx->stream, ctx->s_cell.p, kept, ctx->c_where.p,
                         ctx->mls_index.p, ctx->m_tmp.p, ctx->mls_xyz.p, ctx->mls_normal.p, ctx->mls_curv.p,
                         ctx->mls_alt_index.p);
      PCP_HIP_TRY(ctx, hipGetLastError());
    }
    std::swap(ctx->mls_index, ctx->mls_alt_index);
  } else {
    // upsampled clouds: voxel order, rows dropped by the 2nd SOR removed
    PCP_HIP_TRY(ctx, ctx->s_cell.ensure(static_cast<size_t>(m) + 4));
    if ((rc = compact_flags(ctx, ctx->m_flag.p, m, ctx->s_cell.p, m, &kept)) != PCP_OK) return rc;
    if (kept != m && (rc = compact_results(ctx, ctx->s_cell.p, m, kept)) != PCP_OK) return rc;
  }
  ctx->mls_count = kept;
  if (out_count) *out_count = kept;
  return PCP_OK;
}

// ---- the streamed chain (include/pcp_hip.h) ----

namespace pcp {

// The mean k-NN distances of `m` rows (interleaved positions at rows_xyz) of which rows [r0, r1) are a chunk's own and the
// others its halo -- planes [ea, eb] of the voxel grid are present --: the own rows' distances into css_dist at row0,
// *out_margin = how far their neighbourhoods stay from the first missing planes (k_rows_margin).
static int css_rows_distances(pcp_context *ctx, SmoothStream &st, const float *rows_xyz, int64_t m, int64_t r0, int64_t r1, int64_t ea,
                              int64_t eb, int64_t row0, float *out_margin) {
  const int64_t NX = st.S.v.NX;
  int rc;
  const size_t plane2 = (static_cast<size_t>(m) + 3) & ~size_t(3);
  PCP_HIP_TRY(ctx, ctx->c_xyz2.ensure(3 * plane2 + 4));
  float *x2 = ctx->c_xyz2.p, *y2 = ctx->c_xyz2.p + plane2, *z2 = ctx->c_xyz2.p + 2 * plane2;
  hipLaunchKernelGGL(k_deinterleave, dim3(blocks_of(m)), dim3(kMB), 0, ctx->stream, rows_xyz, m, x2, y2, z2);
  PCP_HIP_TRY(ctx, hipGetLastError());
  CloudView cv2;
  if ((rc = view_of(ctx, x2, y2, z2, m, &cv2)) != PCP_OK) return rc;
  PCP_HIP_TRY(ctx, ctx->s_kth.ensure(static_cast<size_t>(m) + 8));
  const double dens2 = (2.0 * st.p.vgd_iterations + 1.0) / (static_cast<double>(st.p.vgd_voxel_size) * st.p.vgd_voxel_size);
  if ((rc = sor_run(ctx, cv2, st.p.sor_mean_k, st.p.sor_std_mul, false, 0, 1, /*classify=*/false, ctx->s_kth.p, /*clustered=*/true, dens2, r0, r1)) != PCP_OK) return rc;
  // MLSVoxelGrid::getPosition of the first missing plane on either side (fp32, as k_voxel_emit forms it)
  const float xl = static_cast<float>(ea - 1) * st.S.v.vs + st.S.v.bminx, xh = static_cast<float>(eb + 1) * st.S.v.vs + st.S.v.bminx;
  PCP_HIP_TRY(ctx, hipMemsetAsync(ctx->css_words.p + 1, 0xff, 4, ctx->stream));
  hipLaunchKernelGGL(k_rows_margin, dim3(static_cast<uint32_t>(std::min<int64_t>(div_up(r1 - r0, kMB), 4096))), dim3(kMB), 0, ctx->stream,
                     rows_xyz, ctx->s_kth.p, r0, r1, static_cast<double>(xl), static_cast<double>(xh), ea > 0 ? 1 : 0,
                     eb < NX - 1 ? 1 : 0, ctx->css_words.p + 1);
  PCP_HIP_TRY(ctx, hipGetLastError());
  PCP_HIP_TRY(ctx, hipMemcpyAsync(ctx->css_dist.p + row0, ctx->s_dist.p + r0, static_cast<size_t>(r1 - r0) * sizeof(float),
                                  hipMemcpyDeviceToDevice, ctx->stream));
  uint32_t key = 0;
  PCP_HIP_TRY(ctx, hipMemcpyAsync(&key, ctx->css_words.p + 1, 4, hipMemcpyDeviceToHost, ctx->stream));
  PCP_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  *out_margin = float_of_key(key);
  return PCP_OK;
}

// Sweep 1 of one chunk: the voxels of planes [ia, ib] emitted together with `H` planes on either side, the mean k-NN
// distances of every emitted row, the chunk's own rows' distances into css_dist at row0.  *out_rows = the chunk's own rows,
// *out_margin as above, *out_ext_rows = rows computed.
static int css_sweep1_chunk(pcp_context *ctx, SmoothStream &st, const std::vector<unsigned long long> &planes, int64_t ia, int64_t ib,
                            int64_t H, int64_t row0, int64_t *out_rows, float *out_margin, int64_t *out_ext_rows) {
  const int64_t NX = st.S.v.NX, NBY = st.S.NBY;
  const int64_t ea = std::max<int64_t>(0, ia - H), eb = std::min<int64_t>(NX - 1, ib + H);
  *out_margin = INFINITY;
  unsigned long long a = 0, core = 0, ext = 0;
  for (int64_t ix = ea; ix <= eb; ++ix) {
    const unsigned long long c = planes[static_cast<size_t>(ix)];
    ext += c;
    if (ix < ia) a += c;
    else if (ix <= ib) core += c;
  }
  if (ext >= (1ull << 31) - (1ull << 20))
    return set_error(ctx, PCP_ERR_RANGE, "pcp_cloud_smooth_stream_begin: a chunk and its halo hold %llu voxels (2^31 is the limit of one "
                     "emission): use a smaller chunk_capacity", ext);
  int rc;
  GridDesc g;
  if ((rc = stream_grid(ctx, st.cv1, &st.p, &g)) != PCP_OK) return rc;
  st.S.g = g;
  int64_t m = 0;
  if ((rc = vgd_emit(ctx, st.S, ea * NBY, (eb + 1) * NBY, static_cast<int64_t>(ext), &m, ctx->css_words.p, 0, /*positions_only=*/true)) != PCP_OK) return rc;
  // the chunk's own rows inside the (compacted, order-preserving) emission: voxels without a valid fit give no row
  int64_t r0 = 0, r1 = 0;
  if (a > 0 && (rc = compact_flags(ctx, ctx->m_flag.p, static_cast<int64_t>(a), nullptr, 0, &r0)) != PCP_OK) return rc;
  if ((rc = compact_flags(ctx, ctx->m_flag.p, static_cast<int64_t>(a + core), nullptr, 0, &r1)) != PCP_OK) return rc;
  *out_rows = r1 - r0;
  *out_ext_rows = m;
  if (r1 == r0) return PCP_OK;
  return css_rows_distances(ctx, st, ctx->mls_xyz.p, m, r0, r1, ea, eb, row0, out_margin);
}

}  // namespace pcp

int pcp_cloud_smooth_stream_begin(pcp_context *ctx, const pcp_mls_params *p, int64_t chunk_capacity, int64_t *out_total_rows,
                                  int64_t *out_kept_rows, int32_t *out_chunks) {
  if (!ctx) return PCP_ERR_INVALID;
  int rc = check_mls_params(ctx, p);
  if (rc != PCP_OK) return rc;
  if (p->upsampling != 3) return set_error(ctx, PCP_ERR_INVALID, "pcp_cloud_smooth_stream_begin: upsampling must be VOXEL_GRID_DILATION (3)");
  if (chunk_capacity < 4096 || chunk_capacity >= (int64_t(1) << 31))
    return set_error(ctx, PCP_ERR_INVALID, "pcp_cloud_smooth_stream_begin: chunk_capacity must be in [4096, 2^31)");
  if (!ctx->xyz.p) return set_error(ctx, PCP_ERR_STATE, "pcp_cloud_smooth_stream_begin: no cloud uploaded");
  if (int rcf = require_finite_cloud(ctx, "pcp_cloud_smooth_stream_begin")) return rcf;
  if (p->sor_mean_k < 1 || p->sor_mean_k > 254)
    return set_error(ctx, PCP_ERR_INVALID, "pcp_cloud_smooth_stream_begin: sor_mean_k %d out of range (1..254)", p->sor_mean_k);
  PCP_HIP_TRY(ctx, hipSetDevice(ctx->device));
  ctx->mls_count = 0;
  ctx->css_next = -1;
  ctx->css_chunks.clear();
  ctx->sor_distances_live = false;
  ctx->sor_partial_slab = ctx->sor_partial_slabs = -1;
  if (out_total_rows) *out_total_rows = 0;
  if (out_kept_rows) *out_kept_rows = 0;
  if (out_chunks) *out_chunks = 0;
  SmoothStream st{};
  st.p = *p;
  const CloudView cv0 = uploaded_view(ctx);
  struct Building {
    pcp_context *c;
    explicit Building(pcp_context *c_) : c(c_) { c->css_building = true; }
    ~Building() { c->css_building = false; }
  } building(ctx);
  auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const AllocTally alloc_before = alloc_tally();
  double t_mark = now();
  auto lap = [&](int k) {  // (every phase ends in a readback: the stream is idle here)
    const double t = now();
    st.seconds[k] += t - t_mark;
    t_mark = t;
  };
  auto publish = [&]() {
    ctx->css_blob.assign(reinterpret_cast<const uint8_t *>(&st), reinterpret_cast<const uint8_t *>(&st) + sizeof(st));
    ctx->css_next = 0;
  };
  if (cv0.n == 0) {
    publish();
    return PCP_OK;
  }
  int64_t n1 = 0;
  if ((rc = smooth_first_filter(ctx, p, cv0, &st.cv1, &n1)) != PCP_OK) return rc;
  if (n1 == 0) {
    publish();
    return PCP_OK;
  }
  // the fits and the dilated voxel set (counted per plane of the first axis), as pcp_mls_stream_begin leaves them
  int64_t total_voxels = 0;
  // (the strip-wise chunk plan of that call is not used -- the chain cuts by whole planes --: its capacity is set so that
  // the brick form is never left for a strip's sake)
  if ((rc = mls_run(ctx, st.cv1, p, &total_voxels, 0, -1, false, (int64_t(1) << 31) - 1)) != PCP_OK) return rc;
  if (ctx->vgd_blob.size() != sizeof(VgdStream)) return set_error(ctx, PCP_ERR_STATE, "pcp_cloud_smooth_stream_begin: no voxel set");
  std::memcpy(&st.S, ctx->vgd_blob.data(), sizeof(VgdStream));
  ctx->vgd_next = -1;  // (that stream's own chunk plan is not used)
  if (!st.S.bricks)
    return set_error(ctx, PCP_ERR_INVALID, "pcp_cloud_smooth_stream_begin: the voxel set is in its dense form (PCP_VGD_DENSE=1, or a grid taller "
                     "than the brick form handles); the streamed chain cuts the brick form by planes");
  const int64_t NX = st.S.v.NX;
  std::vector<unsigned long long> planes(static_cast<size_t>(NX));
  PCP_HIP_TRY(ctx, hipMemcpyAsync(planes.data(), ctx->v_plane.p, planes.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
  PCP_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  // chunks = maximal runs of whole planes that hold at most chunk_capacity voxels
  std::vector<int64_t> &ch = ctx->css_chunks;  // per chunk: ia, ib, voxels, row0, rows
  {
    int64_t ia = -1;
    unsigned long long cnt = 0;
    for (int64_t ix = 0; ix < NX; ++ix) {
      const unsigned long long c = planes[static_cast<size_t>(ix)];
      if (c > static_cast<unsigned long long>(chunk_capacity))
        return set_error(ctx, PCP_ERR_RANGE, "pcp_cloud_smooth_stream_begin: plane %lld of the voxel grid holds %llu voxels, more than "
                         "chunk_capacity %lld", (long long)ix, c, (long long)chunk_capacity);
      if (ia >= 0 && cnt + c > static_cast<unsigned long long>(chunk_capacity)) {
        ch.insert(ch.end(), {ia, ix - 1, static_cast<int64_t>(cnt), 0, 0});
        ia = -1;
        cnt = 0;
      }
      if (c > 0 && ia < 0) ia = ix;
      cnt += c;
    }
    if (ia >= 0 && cnt > 0) ch.insert(ch.end(), {ia, NX - 1, static_cast<int64_t>(cnt), 0, 0});
  }
  const size_t n_chunks = ch.size() / 5;
  PCP_HIP_TRY(ctx, ctx->css_dist.ensure(static_cast<size_t>(total_voxels) + 8));
  PCP_HIP_TRY(ctx, ctx->css_words.ensure(32));
  PCP_HIP_TRY(ctx, hipMemsetAsync(ctx->css_words.p, 0, 32 * 4, ctx->stream));
  PCP_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  lap(0);
  // The halo must outreach the largest displacement D = max |x of a row - x of its voxel position| of ANY row: a row of a missing
  // plane lies within D of its plane, a chunk's own row within D of the chunk, so a halo of H planes leaves a margin of
  // H vs - 2 D - (k-NN radius).  D itself is only known when every row has been emitted -- which sweep 1 does anyway (every
  // voxel is some chunk's own): it keeps the maximum, and the margins are checked against THAT afterwards.  Sweep 0 only
  // has to size the halo well enough that the check passes: it projects a sample of the voxels (one workgroup in
  // kCssSample, nothing stored: a tenth of an emission; emitting every chunk for D alone was 16 % of the chain).
  constexpr int32_t kCssSample = 8;
  {
    GridDesc g;
    if ((rc = stream_grid(ctx, st.cv1, &st.p, &g)) != PCP_OK) return rc;
    st.S.g = g;
    for (size_t c = 0; c < n_chunks; ++c) {
      int64_t m0 = 0;
      if ((rc = vgd_emit(ctx, st.S, ch[5 * c] * st.S.NBY, (ch[5 * c + 1] + 1) * st.S.NBY, ch[5 * c + 2], &m0, ctx->css_words.p, kCssSample)) != PCP_OK) return rc;
    }
  }
  auto read_max_dx = [&](double *out) -> int {
    uint32_t dx_bits = 0;
    float v = 0.0f;
    PCP_HIP_TRY(ctx, hipMemcpyAsync(&dx_bits, ctx->css_words.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    PCP_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    std::memcpy(&v, &dx_bits, 4);
    if (!(v >= 0.0f) || !std::isfinite(v))
      return set_error(ctx, PCP_ERR_INVALID, "pcp_cloud_smooth_stream_begin: a row of the upsampled cloud has no finite position");
    *out = static_cast<double>(v);
    return PCP_OK;
  };
  if ((rc = read_max_dx(&st.sampled_dx)) != PCP_OK) return rc;
  lap(2);
  // ... halo in planes: twice the displacement (the sample's, and a quarter on top for what it missed) and 16 voxels for the
  // k-NN radius of the dense upsampled surface (1-3 voxels); CHECKED per chunk below against the displacement of all rows,
  // widened where the check fails (PCP_CSS_HALO: another first guess -- the tests force a failure)
  int64_t H = static_cast<int64_t>(std::ceil(2.0 * 1.25 * st.sampled_dx / static_cast<double>(st.S.v.vs))) + 16;
  if (const char *he = std::getenv("PCP_CSS_HALO")) H = std::max(1, std::atoi(he));
  st.halo = static_cast<int32_t>(std::min<int64_t>(H, NX));
  {
    // Every buffer that holds a row (or a voxel) of a chunk and its halo, sized ONCE for the largest chunk: grown chunk by chunk
    // they were allocated several times over -- 206 GB of hipMalloc in a first call on the 10 M-point map, 1.8 of its 4.1 s
    // (a device allocation costs 20-40 ms per GB here; pcp_cloud_smooth_stream_stats [9], [12]).
    unsigned long long most = 0, most_own = 0;  // voxels of the largest chunk with its halo / on its own
    for (size_t c = 0; c < n_chunks; ++c) {
      const int64_t ea = std::max<int64_t>(0, ch[5 * c] - H), eb = std::min<int64_t>(NX - 1, ch[5 * c + 1] + H);
      unsigned long long e = 0;
      for (int64_t ix = ea; ix <= eb; ++ix) e += planes[static_cast<size_t>(ix)];
      most = std::max(most, e);
      most_own = std::max(most_own, static_cast<unsigned long long>(ch[5 * c + 2]));
    }
    if (most < (1ull << 31)) {  // (beyond: css_sweep1_chunk reports the chunk that is too large)
      const size_t n = static_cast<size_t>(most), plane = (n + 3) & ~size_t(3), own = static_cast<size_t>(most_own);
      PCP_HIP_TRY(ctx, ctx->mls_xyz.ensure(3 * n + 4));
      PCP_HIP_TRY(ctx, ctx->mls_alt_xyz.ensure(3 * n + 4));
      // (sweep 1 emits positions only; normals, curvatures and source indices are sweep 2's: a chunk without its halo)
      PCP_HIP_TRY(ctx, ctx->mls_normal.ensure(3 * own + 4));
      PCP_HIP_TRY(ctx, ctx->mls_curv.ensure(own + 4));
      PCP_HIP_TRY(ctx, ctx->mls_index.ensure(own + 4));
      PCP_HIP_TRY(ctx, ctx->mls_alt_normal.ensure(3 * own + 4));
      PCP_HIP_TRY(ctx, ctx->mls_alt_curv.ensure(own + 4));
      PCP_HIP_TRY(ctx, ctx->mls_alt_index.ensure(own + 4));
      PCP_HIP_TRY(ctx, ctx->m_flag.ensure(n + 16));
      PCP_HIP_TRY(ctx, ctx->v_vox.ensure(n + 4));
      PCP_HIP_TRY(ctx, ctx->s_cell.ensure(n + 8));
      PCP_HIP_TRY(ctx, ctx->c_xyz2.ensure(3 * plane + 4));
      PCP_HIP_TRY(ctx, ctx->s_kth.ensure(n + 8));
      PCP_HIP_TRY(ctx, ctx->g_cell.ensure(n + 4));
      PCP_HIP_TRY(ctx, ctx->g_rank.ensure(n + 4));
      PCP_HIP_TRY(ctx, ctx->g_order.ensure(2 * n + 8));
      {
        const size_t before = ctx->g_xyz.count;  // (a new allocation of these planes is zeroed: build_grid says why)
        PCP_HIP_TRY(ctx, ctx->g_xyz.ensure(3 * plane + 4));
        if (ctx->g_xyz.count != before) PCP_HIP_TRY(ctx, hipMemsetAsync(ctx->g_xyz.p, 0, ctx->g_xyz.count * sizeof(float), ctx->stream));
      }
      PCP_HIP_TRY(ctx, ctx->s_dist.ensure(n + 8));
    }
  }
  std::vector<float> margin(n_chunks, INFINITY);
  int64_t row0 = 0;
  // The ball of the trailing filter's selection (in (k + 1) rows by the density bound of the voxel structure; it only sizes the
  // grid, the distances do not depend on it): the rows of a dilated cloud sit on a lattice of voxel columns, and how many columns a
  // ball catches jumps with its radius -- on the 10 M-point map a ball of 1.2 leaves 10 % of the rows with fewer than k + 1
  // neighbours (a wavefront each in k_sor_wave: 0.5 s of the chain), 1.9 leaves 1.4 %, while the map at a tenth of the density is
  // best served by 1.0-1.2 (profiles/r05_css_ball_sweep.log).  So the ball follows what the last chunk reported.
  if (!(ctx->css_ball > 0.0)) ctx->css_ball = 1.2;
  for (size_t c = 0; c < n_chunks; ++c) {
    int64_t rows = 0, ext_rows = 0;
    if ((rc = css_sweep1_chunk(ctx, st, planes, ch[5 * c], ch[5 * c + 1], H, row0, &rows, &margin[c], &ext_rows)) != PCP_OK) return rc;
    ch[5 * c + 3] = row0;
    ch[5 * c + 4] = rows;
    row0 += rows;
    st.rows_computed += ext_rows;
    if (rows > 0) {
      const double flagged = ctx->sor_redo_fraction * static_cast<double>(ext_rows) / static_cast<double>(rows);  // (of the own rows)
      if (std::getenv("PCP_CSS_DEBUG")) std::fprintf(stderr, "[pcp] chunk %zu: ball %.3f, %.4f of the own rows flagged\n", c, ctx->css_ball, flagged);
      if (flagged > 0.03) ctx->css_ball = std::min(2.4, ctx->css_ball * 1.26);
      else if (flagged < 0.003) ctx->css_ball = std::max(1.0, ctx->css_ball / 1.12);
    }
  }
  st.total_rows = row0;
  st.min_margin = INFINITY;
  if ((rc = read_max_dx(&st.max_dx)) != PCP_OK) return rc;  // every row has been emitted: THE displacement
  const int64_t H_needed = static_cast<int64_t>(std::ceil(2.0 * st.max_dx / static_cast<double>(st.S.v.vs))) + 16;
  for (size_t c = 0; c < n_chunks; ++c) {
    int64_t Hc = H;
    // a margin at or below the displacement: some neighbourhood may reach rows the halo did not hold -- again, wider
    while (!(static_cast<double>(margin[c]) > st.max_dx * (1.0 + 1e-6) + 1e-9)) {
      if (Hc >= NX) return set_error(ctx, PCP_ERR_RANGE, "pcp_cloud_smooth_stream_begin: the neighbourhood of a row of chunk %zu has no bound "
                                     "(margin %g m against a displacement of %g m with every plane emitted)", c, (double)margin[c], st.max_dx);
      Hc = std::max(2 * Hc, Hc < H_needed ? H_needed : 0);
      int64_t rows = 0, ext_rows = 0;
      if ((rc = css_sweep1_chunk(ctx, st, planes, ch[5 * c], ch[5 * c + 1], Hc, ch[5 * c + 3], &rows, &margin[c], &ext_rows)) != PCP_OK) return rc;
      if (rows != ch[5 * c + 4]) return set_error(ctx, PCP_ERR_STATE, "pcp_cloud_smooth_stream_begin: chunk %zu changed its rows", c);
      st.redone += 1;
      st.rows_computed += ext_rows;
      st.halo = static_cast<int32_t>(std::max<int64_t>(st.halo, std::min<int64_t>(Hc, NX)));
    }
    if (ch[5 * c + 4] > 0) st.min_margin = std::min(st.min_margin, static_cast<double>(margin[c]));
  }
  if (st.total_rows > 0) {
    // mean + mul * stddev over ALL rows in row order (chunks of kSorChunk rows, fixed trees: the same on every run)
    const int64_t blocks = div_up(st.total_rows, kSorChunk);
    PCP_HIP_TRY(ctx, ctx->m_sums.ensure(4 + 2 * static_cast<size_t>(blocks)));
    hipLaunchKernelGGL(k_rows_stats, dim3(static_cast<uint32_t>(blocks)), dim3(kMB), 0, ctx->stream, ctx->css_dist.p, st.total_rows,
                       ctx->m_sums.p + 4);
    hipLaunchKernelGGL(k_sor_threshold, dim3(1), dim3(64), 0, ctx->stream, ctx->m_sums.p + 4, blocks, st.total_rows, p->sor_std_mul,
                       ctx->m_sums.p);
    unsigned long long *d_count = reinterpret_cast<unsigned long long *>(ctx->css_words.p + 4);
    hipLaunchKernelGGL(k_rows_classify, dim3(static_cast<uint32_t>(std::min<int64_t>(div_up(st.total_rows, kMB), 1 << 16))), dim3(kMB), 0,
                       ctx->stream, ctx->css_dist.p, st.total_rows, ctx->m_sums.p, static_cast<uint8_t *>(nullptr), d_count);
    PCP_HIP_TRY(ctx, hipGetLastError());
    unsigned long long kept = 0;
    PCP_HIP_TRY(ctx, hipMemcpyAsync(&st.threshold, ctx->m_sums.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    PCP_HIP_TRY(ctx, hipMemcpyAsync(&kept, d_count, 8, hipMemcpyDeviceToHost, ctx->stream));
    PCP_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    st.kept_rows = static_cast<int64_t>(kept);
  }
  lap(3);
  st.seconds[1] = alloc_tally().seconds - alloc_before.seconds;
  st.alloc_bytes = alloc_tally().bytes - alloc_before.bytes;
  publish();
  if (out_total_rows) *out_total_rows = st.total_rows;
  if (out_kept_rows) *out_kept_rows = st.kept_rows;
  if (out_chunks) *out_chunks = static_cast<int32_t>(n_chunks);
  return PCP_OK;
}

int pcp_cloud_smooth_stream_next(pcp_context *ctx, int64_t *out_count) {
  if (!ctx || !out_count) return PCP_ERR_INVALID;
  *out_count = 0;
  if (ctx->css_next < 0 || ctx->css_blob.size() != sizeof(SmoothStream))
    return set_error(ctx, PCP_ERR_STATE, "pcp_cloud_smooth_stream_next: no stream (call pcp_cloud_smooth_stream_begin; any other "
                     "smoothing call ends a stream)");
  PCP_HIP_TRY(ctx, hipSetDevice(ctx->device));
  ctx->mls_count = 0;
  SmoothStream st;
  std::memcpy(&st, ctx->css_blob.data(), sizeof(st));
  const std::vector<int64_t> &ch = ctx->css_chunks;
  // (chunks whose every voxel lacks a valid fit have no rows: skipped)
  while (static_cast<size_t>(ctx->css_next) * 5 < ch.size() && ch[static_cast<size_t>(ctx->css_next) * 5 + 4] == 0) ctx->css_next += 1;
  const size_t c = static_cast<size_t>(ctx->css_next);
  if (c * 5 >= ch.size()) return PCP_OK;  // past the last chunk
  const int64_t ia = ch[5 * c], ib = ch[5 * c + 1], voxels = ch[5 * c + 2], row0 = ch[5 * c + 3], rows = ch[5 * c + 4];
  const int64_t next_before = ctx->css_next;
  int rc;
  double *d_thr = reinterpret_cast<double *>(ctx->css_words.p + 8);
  GridDesc g;
  if ((rc = stream_grid(ctx, st.cv1, &st.p, &g)) != PCP_OK) return rc;  // (build_grid does not go through mls_run: the stream stays)
  st.S.g = g;
  int64_t m = 0;
  if ((rc = vgd_emit(ctx, st.S, ia * st.S.NBY, (ib + 1) * st.S.NBY, voxels, &m)) != PCP_OK) return rc;
  if (m != rows) return set_error(ctx, PCP_ERR_STATE, "pcp_cloud_smooth_stream_next: chunk %zu emitted %lld rows, sweep 1 saw %lld", c,
                                  (long long)m, (long long)rows);
  // source indices back to the uploaded cloud (ctx->c_index: the caller's indices of cloud 1), keep flags by the stored distance
  PCP_HIP_TRY(ctx, hipMemcpyAsync(d_thr, &st.threshold, 8, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_remap_index, dim3(blocks_of(m)), dim3(kMB), 0, ctx->stream, ctx->mls_index.p, m, ctx->c_index.p);
  hipLaunchKernelGGL(k_rows_classify, dim3(static_cast<uint32_t>(std::min<int64_t>(div_up(m, kMB), 1 << 16))), dim3(kMB), 0, ctx->stream,
                     ctx->css_dist.p + row0, m, d_thr, ctx->m_flag.p, static_cast<unsigned long long *>(nullptr));
  PCP_HIP_TRY(ctx, hipGetLastError());
  int64_t kept = 0;
  PCP_HIP_TRY(ctx, ctx->s_cell.ensure(static_cast<size_t>(m) + 4));
  if ((rc = compact_flags(ctx, ctx->m_flag.p, m, ctx->s_cell.p, m, &kept)) != PCP_OK) return rc;
  if (kept != m && (rc = compact_results(ctx, ctx->s_cell.p, m, kept)) != PCP_OK) return rc;
  ctx->mls_count = kept;
  ctx->css_next = next_before + 1;
  *out_count = kept;
  return PCP_OK;
}

int pcp_cloud_smooth_stream_end(pcp_context *ctx) {
  if (!ctx) return PCP_ERR_INVALID;
  PCP_HIP_TRY(ctx, hipSetDevice(ctx->device));
  PCP_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  ctx->css_next = -1;
  ctx->css_chunks.clear();
  ctx->css_dist.release();
  return PCP_OK;
}

int pcp_cloud_smooth_stream_stats(pcp_context *ctx, double out[13]) {
  if (!ctx || !out) return PCP_ERR_INVALID;
  if (ctx->css_blob.size() != sizeof(SmoothStream)) return set_error(ctx, PCP_ERR_STATE, "pcp_cloud_smooth_stream_stats: no stream");
  SmoothStream st;
  std::memcpy(&st, ctx->css_blob.data(), sizeof(st));
  out[0] = st.halo;
  out[1] = st.redone;
  out[2] = st.threshold;
  out[3] = st.max_dx;
  out[4] = st.min_margin;
  out[5] = static_cast<double>(st.rows_computed);
  out[6] = st.sampled_dx;
  out[7] = 4.0 * static_cast<double>(ctx->css_dist.count);
  for (int k = 0; k < 4; ++k) out[8 + k] = st.seconds[k];
  out[12] = st.alloc_bytes;
  return PCP_OK;
}

int pcp_close_pairs(pcp_context *ctx, double radius, int64_t *points_with_close_neighbour) {
  if (!ctx || !points_with_close_neighbour) return PCP_ERR_INVALID;
  if (!ctx->xyz.p) return set_error(ctx, PCP_ERR_STATE, "pcp_close_pairs: no cloud uploaded");
  if (int rcf = require_finite_cloud(ctx, "pcp_close_pairs")) return rcf;
  if (!(radius > 0.0)) return set_error(ctx, PCP_ERR_INVALID, "pcp_close_pairs: radius must be > 0");
  PCP_HIP_TRY(ctx, hipSetDevice(ctx->device));
  *points_with_close_neighbour = 0;
  const CloudView cv = uploaded_view(ctx);
  if (cv.n < 2) return PCP_OK;
  GridDesc g;
  // Cell edge: the radius, but never finer than ~8 cells per point.  A micrometre radius on a map tens of metres across
  // would otherwise take the finest grid there is (2^35 cells: 6 GiB of bitmap and running popcounts, a 4 GiB memset
  // and a scan over 2^29 words, all for one count -- ADVICE r2); with the coarser cell the search is as exact (reach =
  // ceil(radius / cell) = 1, every pair closer than the radius lies in adjacent cells) and the table stays a few
  // entries per point.
  const double vol = std::max<double>(cv.mx[0] - cv.mn[0], 1e-3) * std::max<double>(cv.mx[1] - cv.mn[1], 1e-3) *
                     std::max<double>(cv.mx[2] - cv.mn[2], 1e-3);
  const float by_density = static_cast<float>(std::cbrt(vol / (8.0 * static_cast<double>(cv.n))));
  const float cell = std::max(static_cast<float>(radius) * 1.001f, by_density);
  int rc = build_grid(ctx, cv, cell, static_cast<float>(radius), &g);
  if (rc != PCP_OK) return rc;
  const size_t plane = (static_cast<size_t>(cv.n) + 3) & ~size_t(3);
  PCP_HIP_TRY(ctx, ctx->s_counter.ensure(4));
  PCP_HIP_TRY(ctx, hipMemsetAsync(ctx->s_counter.p, 0, 8, ctx->stream));
  hipLaunchKernelGGL(k_close_pairs, dim3(blocks_of(cv.n)), dim3(kMB), 0, ctx->stream, ctx->g_xyz.p, ctx->g_xyz.p + plane,
                     ctx->g_xyz.p + 2 * plane, ctx->g_start.p, cv.n, g, static_cast<float>(radius * radius), ctx->s_counter.p);
  PCP_HIP_TRY(ctx, hipGetLastError());
  unsigned long long c = 0;
  PCP_HIP_TRY(ctx, hipMemcpyAsync(&c, ctx->s_counter.p, 8, hipMemcpyDeviceToHost, ctx->stream));
  PCP_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  *points_with_close_neighbour = static_cast<int64_t>(c);
  // a micrometre radius takes the finest grid there is: do not keep gigabytes of bitmap for a one-off check
  if (ctx->g_occ.count > (size_t(1) << 25)) {
    ctx->g_occ.release();
    ctx->g_occ_rank.release();
  }
  return PCP_OK;
}

int pcp_sor_distances(pcp_context *ctx, int64_t capacity, float *out_distance) {
  if (!ctx || !out_distance) return PCP_ERR_INVALID;
  if (!ctx->sor_distances_live) return set_error(ctx, PCP_ERR_STATE, "pcp_sor_distances: no pcp_sor result to read");
  if (capacity < ctx->n) return set_error(ctx, PCP_ERR_RANGE, "pcp_sor_distances: capacity %lld < %lld points", (long long)capacity, (long long)ctx->n);
  PCP_HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (ctx->n > 0) {
    PCP_HIP_TRY(ctx, hipMemcpyAsync(out_distance, ctx->s_dist.p, static_cast<size_t>(ctx->n) * 4, hipMemcpyDeviceToHost, ctx->stream));
    PCP_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  return PCP_OK;
}

int pcp_sor_redo_fraction(pcp_context *ctx, double *fraction) {
  if (!ctx || !fraction) return PCP_ERR_INVALID;
  *fraction = ctx->sor_redo_fraction;
  return PCP_OK;
}

}  // extern "C"
