// pcp_colour.hip -- HIP kernels and C-ABI entry points of the colour path:
// per-keyframe SE(3) transform, pinhole projection, z-buffer MIN pass,
// visibility, colour / mask lookup, view scores and the per-point top-5 mean.
//
// gfx950 only.  Build with -ffp-contract=off (see pcp_device.hpp).
//
// Data layout in HBM
//   cloud        SoA fp32 x[n] y[n] z[n] (12 B/point), twice: input order for the
//                per-keyframe calls, Morton order for the batched run
//   frames       DevFrame[F], 192 B each, read through the scalar cache
//   images       uint32 per pixel  B | G<<8 | R<<16 | mask<<24  (one gather
//                fetches colour and segmentation mask)
//   depth maps   uint32[F][mh*mw], the bit pattern of a positive fp32 range:
//                unsigned MIN == float MIN, so atomicMin is exactly the
//                reference's sequential running minimum (view_culling.cpp:117-123)
//   cand bits    uint32[(F+31)/32][n]: bit f of word (f>>5, j) = point j is a cull
//                candidate with an in-image colour pixel in keyframe f
//   top-5 state  SoA score[5][n] rgb[5][n] frame[5][n] count[n]
#include <algorithm>
#include <cfloat>
#include <cstdlib>

#include "pcp_device.hpp"
#include "pcp_hsv.hpp"
#include "pcp_scan.hpp"

namespace pcp {

constexpr int kBlock = 256;
constexpr uint32_t kFltMaxBits = 0x7f7fffffu;  // FLT_MAX, view_culling.cpp:64

// ---------------------------------------------------------------------------
// K1: single-keyframe projection (the roofline kernel): 12 B read, 4 B cell +
// 4 B range written per point (optionally pixel and camera coordinates).
// 4 points per lane: dwordx4 loads / stores, 1 KiB per wave-instruction.
// ---------------------------------------------------------------------------
template <bool kCommon>
__device__ __forceinline__ DevCamera common_camera(const DevCamera &in) {
  DevCamera c = in;
  if constexpr (kCommon) {
    c.cull_mode = PCP_CULL_ZBUFFER;
    c.enable_zbuf = 1;
    c.pretest = 1;
    c.ds_fast = 1;
  }
  return c;
}
inline bool is_common_camera(const DevCamera &c) {
  return c.cull_mode == PCP_CULL_ZBUFFER && c.enable_zbuf == 1 && c.pretest == 1 && c.ds_fast == 1;
}

struct ProjectOut {
  int32_t *cell;
  int32_t *pixel;
  float *range;
  float *xc, *yc, *zc;
};

__device__ __forceinline__ void project_store_one(const DevCamera &cam, const DevFrame &fr, float x, float y, float z,
                                                  int32_t &cell, int32_t &pixel, float &range, float &xc, float &yc,
                                                  float &zc) {
  const Projected p = project_point<true, false>(cam, fr.w2c, x, y, z);
  cell = p.cell;
  pixel = p.pixel;
  range = p.cell != -1 ? static_cast<float>(range64(p.xc, p.yc, p.zc)) : FLT_MAX;
  xc = p.xc;
  yc = p.yc;
  zc = p.zc;
}

template <bool kCommon>
__global__ __launch_bounds__(kBlock) void k_project_frame(const float *__restrict__ x, const float *__restrict__ y,
                                                          const float *__restrict__ z, int64_t n, DevCamera cam_in,
                                                          DevFrame fr, ProjectOut out) {
  const DevCamera cam = common_camera<kCommon>(cam_in);
  const int64_t q = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;  // quad index
  const int64_t i0 = q * 4;
  if (i0 >= n) return;
  if (i0 + 3 < n) {
    // Non-temporal loads and stores: every byte of this kernel is touched once per launch (12 B read, 8 B written per
    // point), so keeping lines in L2 / the Infinity Cache only evicts what other kernels could reuse.  On a 40 M-point
    // cloud (763 MiB per launch): 142 -> 132 us, 5.63 -> 6.04 TB/s (loads alone: no gain; stores alone: 139 us).
    typedef float nt_f4 __attribute__((ext_vector_type(4)));
    typedef int nt_i4 __attribute__((ext_vector_type(4)));
    const nt_f4 ax = __builtin_nontemporal_load(reinterpret_cast<const nt_f4 *>(x + i0));
    const nt_f4 ay = __builtin_nontemporal_load(reinterpret_cast<const nt_f4 *>(y + i0));
    const nt_f4 az = __builtin_nontemporal_load(reinterpret_cast<const nt_f4 *>(z + i0));
    const float4 vx = make_float4(ax.x, ax.y, ax.z, ax.w), vy = make_float4(ay.x, ay.y, ay.z, ay.w),
                 vz = make_float4(az.x, az.y, az.z, az.w);
    int4 cell, pixel;
    float4 range, xc, yc, zc;
    project_store_one(cam, fr, vx.x, vy.x, vz.x, cell.x, pixel.x, range.x, xc.x, yc.x, zc.x);
    project_store_one(cam, fr, vx.y, vy.y, vz.y, cell.y, pixel.y, range.y, xc.y, yc.y, zc.y);
    project_store_one(cam, fr, vx.z, vy.z, vz.z, cell.z, pixel.z, range.z, xc.z, yc.z, zc.z);
    project_store_one(cam, fr, vx.w, vy.w, vz.w, cell.w, pixel.w, range.w, xc.w, yc.w, zc.w);
    if (out.cell) {
      const nt_i4 c4 = {cell.x, cell.y, cell.z, cell.w};
      __builtin_nontemporal_store(c4, reinterpret_cast<nt_i4 *>(out.cell + i0));
    }
    if (out.range) {
      const nt_f4 r4 = {range.x, range.y, range.z, range.w};
      __builtin_nontemporal_store(r4, reinterpret_cast<nt_f4 *>(out.range + i0));
    }
    if (out.pixel) *reinterpret_cast<int4 *>(out.pixel + i0) = pixel;
    if (out.xc) {
      *reinterpret_cast<float4 *>(out.xc + i0) = xc;
      *reinterpret_cast<float4 *>(out.yc + i0) = yc;
      *reinterpret_cast<float4 *>(out.zc + i0) = zc;
    }
  } else {
    for (int64_t i = i0; i < n; ++i) {
      int32_t cell, pixel;
      float range, xc, yc, zc;
      project_store_one(cam, fr, x[i], y[i], z[i], cell, pixel, range, xc, yc, zc);
      if (out.cell) out.cell[i] = cell;
      if (out.range) out.range[i] = range;
      if (out.pixel) out.pixel[i] = pixel;
      if (out.xc) {
        out.xc[i] = xc;
        out.yc[i] = yc;
        out.zc[i] = zc;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// K0: tile x keyframe visibility masks.  A tile is 64 consecutive Morton-ordered
// points = one wavefront of the batched kernels; its bounding sphere is computed
// at upload.  Bit f of tile_mask[tile][f >> 5] is CLEARED only when every point of
// the tile is certain to be rejected by the reference in keyframe f, proved in fp64
// with outward-padded interval arithmetic:
//   (a) the sphere, transformed with the fp32 matrix the reference uses and
//       inflated by the matrix's spectral-norm bound and the fp32 rounding slack of
//       the per-point transform, lies in z <= 0; or
//   (b) it lies in z > 0 and the interval image of its normalised coordinates under
//       the distortion polynomial misses the acceptance box of both the cull-cell
//       and the colour-pixel rule (same box as surely_rejected()).
// Anything else (straddling z = 0, fold-back distortion, huge tiles) keeps the bit.
// ---------------------------------------------------------------------------
struct Interval {
  double lo, hi;
};
__device__ __forceinline__ Interval iv_add(Interval a, Interval b) { return {a.lo + b.lo, a.hi + b.hi}; }
__device__ __forceinline__ Interval iv_scale(double k, Interval a) {
  return k >= 0.0 ? Interval{k * a.lo, k * a.hi} : Interval{k * a.hi, k * a.lo};
}
__device__ __forceinline__ Interval iv_mul(Interval a, Interval b) {
  const double p0 = a.lo * b.lo, p1 = a.lo * b.hi, p2 = a.hi * b.lo, p3 = a.hi * b.hi;
  return {fmin(fmin(p0, p1), fmin(p2, p3)), fmax(fmax(p0, p1), fmax(p2, p3))};
}
__device__ __forceinline__ Interval iv_sqr(Interval a) {
  const double l = fabs(a.lo), h = fabs(a.hi);
  const double mx = fmax(l, h), mn = (a.lo <= 0.0 && a.hi >= 0.0) ? 0.0 : fmin(l, h);
  return {mn * mn, mx * mx};
}
// a >= 0 and b >= 0 (squares, and reciprocals of positive numbers): the corner products are the bounds
__device__ __forceinline__ Interval iv_mul_pos(Interval a, Interval b) { return {a.lo * b.lo, a.hi * b.hi}; }
// b > 0: the sign of each bound of a picks its factor
__device__ __forceinline__ Interval iv_mul_by_pos(Interval a, Interval b) {
  return {a.lo * (a.lo >= 0.0 ? b.lo : b.hi), a.hi * (a.hi >= 0.0 ? b.hi : b.lo)};
}
// 1 / x of a positive normal x, within 1e-14 relative (the hardware estimate and two Newton steps; the intervals are padded by
// 1e-12 wherever it is used: an exactly rounded quotient costs three times the instructions)
__device__ __forceinline__ double iv_rcp_pos(double b) {
  double r = __builtin_amdgcn_rcp(b);
  r = fma(fma(-b, r, 1.0), r, r);
  r = fma(fma(-b, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ Interval iv_pad(Interval a) {  // absorbs the fp64 rounding of the interval arithmetic
  const double e = 1e-12 * (fabs(a.lo) + fabs(a.hi)) + 1e-300;
  return {a.lo - e, a.hi + e};
}

// Returns 1 when every point of the tile is certain to be rejected (exactness-critical, see above),
// 2 when the whole tile images strictly inside the acceptance box (a performance hint only: the
// per-point fp32 rejection test could not reject anything there, so the depth pass skips it), else 0.
__device__ __forceinline__ int tile_classify(const DevCamera &c, const DevFrame &fr, float4 sph) {
  const double cx = sph.x, cy = sph.y, cz = sph.z;
  const float *m = fr.w2c;
  const double X = (m[0] * cx + m[1] * cy) + (m[2] * cz + m[3]);
  const double Y = (m[4] * cx + m[5] * cy) + (m[6] * cz + m[7]);
  const double Z = (m[8] * cx + m[9] * cy) + (m[10] * cz + m[11]);
  // |M (p - c)| <= s |p - c|; fp32 transform rounding <= 3 * 2^-24 * sum of term magnitudes
  const double mag = fr.norm_bound * (fabs(cx) + fabs(cy) + fabs(cz) + 3.0 * sph.w) + fabs(static_cast<double>(m[3])) +
                     fabs(static_cast<double>(m[7])) + fabs(static_cast<double>(m[11])) + 1e-3;
  const double rho = fr.norm_bound * static_cast<double>(sph.w) * (1.0 + 1e-6) + 1e-6 * mag;
  if (!(rho >= 0.0) || !isfinite(X + Y + Z + rho)) return 0;
  if (Z + rho <= 0.0) return 1;   // (a) entirely behind the camera
  if (Z - rho <= 0.0) return 0;  // straddles z = 0: no bound on x / z
  const Interval zi{Z - rho, Z + rho};
  if (!(zi.lo >= 1e-290)) return 0;  // (a subnormal depth: no reciprocal estimate, no verdict)
  const Interval iz{iv_rcp_pos(zi.hi) * (1.0 - 1e-13), iv_rcp_pos(zi.lo) * (1.0 + 1e-13)};
  const Interval xn = iv_pad(iv_mul_by_pos(Interval{X - rho, X + rho}, iz));
  const Interval yn = iv_pad(iv_mul_by_pos(Interval{Y - rho, Y + rho}, iz));
  const Interval x2 = iv_sqr(xn), y2 = iv_sqr(yn);
  const Interval r2 = iv_add(x2, y2);
  const Interval r4 = iv_mul_pos(r2, r2);
  const Interval r6 = iv_mul_pos(r2, r4);
  Interval rc = iv_add(iv_add(iv_scale(c.k1, r2), iv_scale(c.k2, r4)), iv_scale(c.k3, r6));
  rc.lo += 1.0;
  rc.hi += 1.0;
  const Interval t1 = iv_scale(2.0, iv_mul(xn, yn));
  const Interval t2 = iv_add(r2, iv_scale(2.0, x2));
  const Interval t3 = iv_add(r2, iv_scale(2.0, y2));
  const Interval xd = iv_pad(iv_add(iv_add(iv_mul(rc, xn), iv_scale(c.p1, t1)), iv_scale(c.p2, t2)));
  const Interval yd = iv_pad(iv_add(iv_add(iv_mul(rc, yn), iv_scale(c.p1, t3)), iv_scale(c.p2, t1)));
  Interval u = iv_pad(iv_scale(c.fx, xd)), v = iv_pad(iv_scale(c.fy, yd));
  u.lo += c.cx; u.hi += c.cx;
  v.lo += c.cy; v.hi += c.cy;
  if (!isfinite(u.lo + u.hi + v.lo + v.hi)) return 0;
  // the box already carries a 0.5 px margin (pcp_set_camera)
  if (u.hi < static_cast<double>(c.u_lo) || u.lo > static_cast<double>(c.u_hi) ||
      v.hi < static_cast<double>(c.v_lo) || v.lo > static_cast<double>(c.v_hi))
    return 1;
  return (u.lo > static_cast<double>(c.u_lo) + 1.0 && u.hi < static_cast<double>(c.u_hi) - 1.0 &&
          v.lo > static_cast<double>(c.v_lo) + 1.0 && v.hi < static_cast<double>(c.v_hi) - 1.0)
             ? 2
             : 0;
}

// Two launches: first the spheres of groups of 16 tiles against every keyframe, then the tiles, which only test
// keyframes their group survived -- most groups are rejected as a whole.
constexpr int kTileGroup = 16;

// Group level, flat form: one lane per (group, keyframe); the 32 lanes of a half wavefront are the bits of one mask
// word, gathered by a ballot.  (A lane per (group, word) gives 1 224 wavefronts at C3, each looping over
// 32 keyframes: barely more than one wavefront per SIMD, 48 us of pure latency; this form has 32 times the lanes.)
__global__ __launch_bounds__(kBlock) void k_group_mask_flat(const float4 *__restrict__ spheres, int64_t groups, DevCamera cam,
                                                            const DevFrame *__restrict__ frames, int32_t n_frames,
                                                            int32_t w0, int32_t w1, int32_t words,
                                                            uint32_t *__restrict__ group_mask,
                                                            uint32_t *__restrict__ group_inside, int32_t cull_enabled,
                                                            int32_t *__restrict__ zeroed, int32_t n_zeroed,
                                                            unsigned long long *__restrict__ far_maps, int64_t far_cells) {
  const int64_t idx = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (idx < n_zeroed) zeroed[idx] = 0;  // the counters of the stage's last kernels (sort_tiles_by_work)
  // ... and the squared-range maps of the pass at "nothing seen yet" (the byte pattern 0x7f: a large finite double), here instead
  // of a fill launch of the runtime in front of the stage (6.6 us and a 6 us gap per step)
  for (int64_t c = idx; c < far_cells; c += static_cast<int64_t>(gridDim.x) * kBlock) far_maps[c] = 0x7f7f7f7f7f7f7f7full;
  const int32_t nw = w1 - w0;
  const int64_t gw = idx >> 5;  // word of this half wavefront, counted over the launch
  const int64_t group = gw / nw;
  const bool live = group < groups;
  const int32_t w = w0 + static_cast<int32_t>(gw - group * nw);
  const int32_t f = (w << 5) + static_cast<int32_t>(idx & 31);
  int cls = 1;
  if (live && f < n_frames) cls = cull_enabled ? tile_classify(cam, frames[f], spheres[group]) : 0;
  // a group that images wholly inside the acceptance box (and in front of the camera): none of its tiles can be rejected
  // -- a tile's interval image contains the images of its points, which lie in the box -- so the tile level has nothing to
  // decide for that pair
  const unsigned long long m = __ballot(cls != 1), in = __ballot(cls == 2);
  if (live && (idx & 31) == 0) {
    group_mask[group * words + w] = static_cast<uint32_t>((threadIdx.x & 32) ? (m >> 32) : m);
    group_inside[group * words + w] = static_cast<uint32_t>((threadIdx.x & 32) ? (in >> 32) : in);
  }
}

// Tile level, dense form: one wavefront per (group of 16 tiles, 32-keyframe word), lane = slot * 16 + tile-in-group.
// The keyframes of the word that the group survived (its parent mask) are taken four at a time, so all 64 lanes
// classify (tile, keyframe) pairs until the word is exhausted (a lane per (tile, word) ran at 32 % lane utilisation:
// every lane looped over its own number of surviving keyframes; a wavefront per group with all its words: 9 768
// wavefronts at C3, barely more than the chip holds at once, the longest deciding the kernel).  The verdicts are
// gathered with wave ballots; the 16 lanes of slot 0 own their tiles' mask words.
__global__ __launch_bounds__(kBlock) void k_tile_mask_dense(const float4 *__restrict__ spheres, int64_t tiles,
                                                            DevCamera cam, const DevFrame *__restrict__ frames,
                                                            int32_t n_frames, int32_t w0, int32_t w1, int32_t words,
                                                            const uint32_t *__restrict__ group_mask,
                                                            const uint32_t *__restrict__ group_inside,
                                                            uint32_t *__restrict__ tile_mask,
                                                            uint32_t *__restrict__ inside_mask, int32_t cull_enabled) {
  const int lane = threadIdx.x & 63;
  const int32_t nw = w1 - w0;
  const int64_t gw = static_cast<int64_t>(blockIdx.x) * (kBlock / 64) + (threadIdx.x >> 6);
  const int64_t group = gw / nw;
  if (group >= (tiles + kTileGroup - 1) / kTileGroup) return;  // whole wavefronts leave together
  const int32_t w = w0 + static_cast<int32_t>(gw - group * nw);
  const int t = lane & 15, slot = lane >> 4;
  const int64_t tile = group * kTileGroup + t;
  const bool have = tile < tiles;
  uint32_t todo = __builtin_amdgcn_readfirstlane(group_mask[group * words + w]);
  const int32_t nb = n_frames - (w << 5);
  if (nb < 32) todo &= nb <= 0 ? 0u : ((1u << nb) - 1u);
  // keyframes in which the whole group images inside the box: every tile keeps the pair (and skips the pre-test) unexamined
  const uint32_t whole = __builtin_amdgcn_readfirstlane(group_inside[group * words + w]) & todo;
  todo &= ~whole;
  uint32_t word = whole, inside = whole;
  if (todo) {
    const float4 sph = spheres[have ? tile : tiles - 1];
    while (todo) {
      int32_t bit[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        bit[k] = todo ? __builtin_ctz(todo) : -1;
        todo &= todo - (todo ? 1u : 0u);
      }
      const int32_t b = slot == 0 ? bit[0] : slot == 1 ? bit[1] : slot == 2 ? bit[2] : bit[3];
      int cls = 1;
      if (b >= 0 && have) cls = cull_enabled ? tile_classify(cam, frames[(w << 5) + b], sph) : 0;
      const unsigned long long keep = __ballot(cls != 1), ins = __ballot(cls == 2);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (bit[k] >= 0) {
          word |= static_cast<uint32_t>((keep >> (k * 16 + t)) & 1ull) << bit[k];
          inside |= static_cast<uint32_t>((ins >> (k * 16 + t)) & 1ull) << bit[k];
        }
      }
    }
  }
  if (slot == 0 && have) {
    tile_mask[tile * words + w] = word;
    if (inside_mask) inside_mask[tile * words + w] = inside;
  }
}

// Longest-work-first order of the tiles (counting sort on the number of keyframes a tile walks, descending).  The
// batched passes run one wavefront-sized workgroup per tile in this order: the number of keyframes per tile ranges
// from 0 to several times the mean, and in cloud order the last heavy tiles leave most of the chip idle at the end
// of a pass (list-scheduling the measured C3 weights: 1.14x the ideal makespan in cloud order with 4-tile
// workgroups, 1.0003x longest-first with 1-tile workgroups).  Ties are placed in arrival order: any permutation is
// correct, the results do not depend on it.
// Two launches (they were four and a memset: each small launch costs its 5 us and a gap): k_tile_work_hist counts a tile's
// keyframes (the scheduling weight of its wavefront) and bins them; k_work_scatter turns the histogram into offsets itself,
// in every workgroup, and places the tiles.  The histogram and the bins' cursors are zeroed by the first kernel of the stage
// (k_group_mask_flat).
constexpr int kWorkBins = 1024;
constexpr int kWorkPerBlock = 1024;  // tiles per workgroup of the two sorting kernels (4 per lane)
// workgroup-local histogram in LDS, then one global atomic per occupied bin (many tiles share a weight: per-tile
// global atomics on the same few addresses serialise)
__global__ __launch_bounds__(kBlock) void k_tile_work_hist(const uint32_t *__restrict__ tile_mask, int64_t tiles, int32_t w0,
                                                          int32_t w1, int32_t words, int32_t *__restrict__ tile_work,
                                                          int32_t *__restrict__ hist) {
  __shared__ int32_t cnt[kWorkBins];
  for (int b = threadIdx.x; b < kWorkBins; b += kBlock) cnt[b] = 0;
  __syncthreads();
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kWorkPerBlock;
  for (int q = 0; q < kWorkPerBlock / kBlock; ++q) {
    const int64_t t = base + q * kBlock + threadIdx.x;
    if (t < tiles) {
      int32_t visits = 0;
      for (int32_t w = w0; w < w1; ++w) visits += __builtin_popcount(tile_mask[t * words + w]);
      tile_work[t] = visits;
      atomicAdd(&cnt[min(visits, kWorkBins - 1)], 1);
    }
  }
  __syncthreads();
  for (int b = threadIdx.x; b < kWorkBins; b += kBlock)
    if (cnt[b]) atomicAdd(&hist[b], cnt[b]);
}
__device__ __forceinline__ int32_t block_exclusive_scan(int32_t v, int32_t *total);
// hist: the finished histogram; cursor[b] (zeroed): tiles of bin b placed so far
__global__ __launch_bounds__(kBlock) void k_work_scatter(const int32_t *__restrict__ work, int64_t tiles,
                                                        const int32_t *__restrict__ hist, int32_t *__restrict__ cursor,
                                                        int32_t *__restrict__ order) {
  __shared__ int32_t cnt[kWorkBins];    // local count, then the workgroup's first slot of the bin
  __shared__ int32_t first[kWorkBins];  // number of tiles with more work than the bin
  for (int b = threadIdx.x; b < kWorkBins; b += kBlock) cnt[b] = 0;
  {
    // heaviest bin first: thread t takes bins kWorkBins - 1 - 4 t ... kWorkBins - 4 - 4 t
    constexpr int kPer = kWorkBins / kBlock;
    int32_t h[kPer], mine = 0;
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      h[k] = hist[kWorkBins - 1 - (static_cast<int>(threadIdx.x) * kPer + k)];
      mine += h[k];
    }
    int32_t total = 0;
    int32_t run = block_exclusive_scan(mine, &total);
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      first[kWorkBins - 1 - (static_cast<int>(threadIdx.x) * kPer + k)] = run;
      run += h[k];
    }
  }
  __syncthreads();
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kWorkPerBlock;
  int32_t bin[kWorkPerBlock / kBlock], rank[kWorkPerBlock / kBlock];
#pragma unroll
  for (int q = 0; q < kWorkPerBlock / kBlock; ++q) {
    const int64_t t = base + q * kBlock + threadIdx.x;
    bin[q] = t < tiles ? min(work[t], kWorkBins - 1) : -1;
    rank[q] = bin[q] >= 0 ? atomicAdd(&cnt[bin[q]], 1) : 0;
  }
  __syncthreads();
  for (int b = threadIdx.x; b < kWorkBins; b += kBlock)
    if (cnt[b]) cnt[b] = first[b] + atomicAdd(&cursor[b], cnt[b]);
  __syncthreads();
#pragma unroll
  for (int q = 0; q < kWorkPerBlock / kBlock; ++q)
    if (bin[q] >= 0) order[cnt[bin[q]] + rank[q]] = static_cast<int32_t>(base + q * kBlock + threadIdx.x);
}

// 256-thread launches without a tile order (single-keyframe calls, culling switched off): workgroups are handed
// to the 8 XCDs round-robin (workgroup b runs on XCD b % 8) and every XCD has its own 4 MiB L2; this remap makes
// every run of kXcdChunk consecutive workgroups of the Morton-ordered cloud execute on ONE XCD, so a patch of space,
// and of every image it projects into, is served by one L2 instead of eight.  Measured at C3: neutral to +2 %.
// Whole contiguous eighths per XCD were 40 % slower (the work per tile varies, the XCDs finish far apart).
constexpr uint32_t kXcdChunk = 16;
__device__ __forceinline__ uint32_t xcd_chunked_block() {
  constexpr uint32_t kXcd = 8, kSuper = kXcd * kXcdChunk;
  const uint32_t b = blockIdx.x;
  if (b >= (gridDim.x / kSuper) * kSuper) return b;  // ragged tail: identity
  const uint32_t x = b % kXcd, slot = b / kXcd;
  return ((slot / kXcdChunk) * kXcd + x) * kXcdChunk + slot % kXcdChunk;
}

// ---------------------------------------------------------------------------
// K2: z-buffer MIN pass, keyframes [f0, f1).  One lane per point, one wavefront
// per tile; the point stays in registers across the keyframe loop (12 B read per
// point per pass) and the wavefront only visits keyframes its tile mask keeps.
// tile_mask == nullptr: visit every keyframe of the range.
// ---------------------------------------------------------------------------
// The z-buffer holds, per cell, min over the points of f32(sqrt(s)), s = (X X + Y Y) + Z Z in fp64 (view_culling.cpp:102).
// s -> f32(RN(sqrt(s))) is monotone, so that minimum is f32(sqrt(min s)): the pass takes the minimum of s itself (the bit
// pattern of a non-negative double orders as an unsigned integer) and k_depth_finish takes ONE square root per cell -- the
// correctly rounded fp64 sqrt (a 16-cycle v_rsq_f64 and ~20 more instructions) leaves the per-visit path.
constexpr unsigned long long kDepthEmpty = 0x7f7f7f7f7f7f7f7full;  // hipMemset pattern; above every finite s, below NaN
__device__ __forceinline__ void depth_min(unsigned long long *__restrict__ map, int32_t cell, unsigned long long bits) {
  // a plain (possibly stale, hence >= current) read filters most atomics
  if (bits < map[cell]) atomicMin(map + cell, bits);
}
__device__ __forceinline__ double sumsq64(float xc, float yc, float zc) {
  const double X = xc, Y = yc, Z = zc;
  return (X * X + Y * Y) + Z * Z;
}
// range map of the reference from the squared-range map: f32(sqrt(min s)); an empty cell keeps FLT_MAX, and so does a cell
// whose nearest point is farther than FLT_MAX (the reference's `dist < map` is false for an infinite fp32 range)
__global__ __launch_bounds__(kBlock) void k_depth_finish(const unsigned long long *__restrict__ sq, int64_t count,
                                                         uint32_t *__restrict__ range) {
  int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (; i < count; i += stride) {
    const unsigned long long b = sq[i];
    float r = FLT_MAX;
    if (b != kDepthEmpty) r = fminf(static_cast<float>(sqrt(__longlong_as_double(static_cast<long long>(b)))), FLT_MAX);
    range[i] = __float_as_uint(r);
  }
}

__device__ __forceinline__ uint32_t range_bits(int32_t w, int32_t f0, int32_t f1) {
  const int32_t fb = max(f0, w << 5), fe = min(f1, (w << 5) + 32);
  const int32_t nb = fe - fb;
  return nb <= 0 ? 0u : (((nb >= 32) ? 0xffffffffu : ((1u << nb) - 1u)) << (fb & 31));
}

// kCommon: the configuration every batched run of the CLI and the bench uses (z-buffer cull with its depth buffer on, fp32
// rejection test and exact short divisions enabled) is compiled with those switches as constants -- the mode fields, the
// hidden_points_removal bounds and the branches on them leave the scalar register file (see common_camera).
template <bool kCommon>
__global__ __launch_bounds__(kBlock) void k_depth_pass(const float *__restrict__ x, const float *__restrict__ y,
                                                       const float *__restrict__ z, int64_t n, DevCamera cam_in,
                                                       const DevFrame *__restrict__ frames, int32_t f0, int32_t f1,
                                                       unsigned long long *__restrict__ depth, int64_t cells,
                                                       int32_t depth_first_frame, uint32_t *__restrict__ tile_mask,
                                                       const uint32_t *__restrict__ tile_inside, int32_t words,
                                                       const int32_t *__restrict__ tile_order) {
  DevCamera cam = common_camera<kCommon>(cam_in);
  // The fp32 rejection test only pays where whole wavefronts fail it.  Behind the tile masks few do -- a quarter of the visits of a
  // C3 step hold no candidate at all (5.62 % of the pairs survive the tile level, 4.24 % this pass's refinement below) -- and every
  // other visit outside the `inside` pairs paid its 45 instructions for nothing:
  // depth pass 0.635 -> 0.568 ms without it.  (The other configurations keep it, among them the single-keyframe calls, which
  // have no masks.)
  if constexpr (kCommon) cam.pretest = 0;
  // per-wavefront combining table: slot = cell & 63 holds min over (cell << 32 | range bits)
  __shared__ unsigned long long combine[kBlock];
  const int lane = threadIdx.x & 63;
  unsigned long long *tbl = combine + (threadIdx.x & ~63);
  // tile_order: wavefront-sized workgroups, longest-work-first; else 256-thread workgroups in (XCD-chunked) cloud order
  const int64_t j = tile_order ? static_cast<int64_t>(tile_order[blockIdx.x]) * 64 + threadIdx.x
                               : static_cast<int64_t>(xcd_chunked_block()) * kBlock + threadIdx.x;
  const bool live = j < n;
  // a wavefront wholly past the cloud (tail of a 256-thread workgroup) has no tile: its mask words do not exist
  if (!__ballot(live)) return;
  const float px = live ? x[j] : 0.0f, py = live ? y[j] : 0.0f, pz = live ? z[j] : 0.0f;
  const int64_t tile = __builtin_amdgcn_readfirstlane(static_cast<int32_t>(j >> 6));
  // every camera coordinate of this wavefront is finite (divide_xy_by_z)
  const bool finite_sure = cam.frames_bounded != 0 && __all(fabsf(px) <= 0x1p40f && fabsf(py) <= 0x1p40f && fabsf(pz) <= 0x1p40f);
  // The map's atomic of a visit waits for a plain read of the cell (depth_min), and that read for the memory: the read is
  // issued at the end of the visit and the atomic follows behind the NEXT visit's projection, which needs nothing from memory.
  unsigned long long *held_at = nullptr;  // the held lane's cell in its keyframe's map (null: this lane holds nothing)
  unsigned long long held_bits = 0ull, held_seen = 0ull;
  auto settle_held = [&]() {
    if (held_at && held_bits < held_seen) atomicMin(held_at, held_bits);
    held_at = nullptr;
  };
  for (int32_t w = f0 >> 5; w <= (f1 - 1) >> 5; ++w) {
    const uint32_t in_range = range_bits(w, f0, f1);
    uint32_t todo = in_range;
    if (tile_mask) todo &= tile_mask[tile * words + w];
    todo = __builtin_amdgcn_readfirstlane(todo);
    // pairs whose tile images inside the acceptance box: the fp32 rejection test is skipped (it could not reject)
    const uint32_t inside = tile_inside ? __builtin_amdgcn_readfirstlane(tile_inside[tile * words + w]) : 0u;
    uint32_t seen = 0u;  // keyframes in which some lane can be coloured
    while (todo) {
      const int32_t b = __builtin_ctz(todo);
      const int32_t f = (w << 5) + b;
      todo &= todo - 1u;
      const DevFrame &fr = frames[f];
      const Projected p = project_point(cam, fr.w2c, px, py, pz, ((inside >> b) & 1u) == 0u, finite_sure);
      const bool in_map = live && p.cell >= 0;
      const bool cand = live && p.pixel >= 0 && (cam.enable_zbuf ? p.cell >= 0 : p.cell != -1);
      if (__ballot(cand)) seen |= 1u << b;
      settle_held();
      if (cam.enable_zbuf && __ballot(in_map)) {
        // wave-level combine: lanes of a tile hit a handful of cells, one atomic per cell suffices.  (Reading the
        // cell's current value first and skipping the square root, the combine and the atomic for points that cannot
        // lower it -- s >= m * m -- was slower, 0.79 -> 0.82 ms: some lane of the wavefront nearly always stays, so the
        // wavefront pays for the whole path anyway, plus the extra gather.)
        unsigned long long *map = depth + static_cast<uint64_t>(static_cast<uint32_t>(f - depth_first_frame)) * static_cast<uint32_t>(cells);
        // the table elects by the upper half of s's bit pattern (monotone in s): lanes of a cell that tie there (ranges
        // within 1e-6 of each other) all go to the map, which settles them in full precision
        unsigned long long key = ~0ull, sbits = 0ull;
        if (in_map) {
          sbits = static_cast<unsigned long long>(__double_as_longlong(sumsq64(p.xc, p.yc, p.zc)));
          key = (static_cast<unsigned long long>(static_cast<uint32_t>(p.cell)) << 32) | (sbits >> 32);
        }
        tbl[lane] = ~0ull;
        __builtin_amdgcn_wave_barrier();
        if (in_map) atomicMin(&tbl[p.cell & 63], key);
        __builtin_amdgcn_wave_barrier();
        if (in_map) {
          const unsigned long long got = tbl[p.cell & 63];
          // winner of its cell, or a cell that lost its slot to a smaller cell id
          if (got == key || static_cast<uint32_t>(got >> 32) != static_cast<uint32_t>(p.cell)) {
            held_at = map + p.cell;
            held_bits = sbits;
            held_seen = *held_at;  // (possibly stale, hence >= the cell's value: the read filters most atomics)
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    // drop the pairs in which no lane is a colouring candidate: the colour pass skips them
    if (tile_mask && lane == 0 && live) {
      uint32_t *dst = tile_mask + tile * words + w;
      *dst = (*dst & ~in_range) | seen;
    }
  }
  settle_held();
}

// ---------------------------------------------------------------------------
// K3: single-keyframe keep mask (pass 2 of view_culling), flags in input order.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_visibility(const float *__restrict__ x, const float *__restrict__ y,
                                                       const float *__restrict__ z, int64_t n, DevCamera cam,
                                                       DevFrame fr, const uint32_t *__restrict__ depth,
                                                       const int32_t *__restrict__ perm,
                                                       uint8_t *__restrict__ keep, int32_t require_pixel) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= n) return;
  const Projected p = project_point(cam, fr.w2c, x[i], y[i], z[i]);
  bool k = keep_rule(cam, p, depth);
  if (require_pixel) k = k && p.pixel >= 0;
  // the cloud is walked in Morton order, the flags land in input order: `keep` is zeroed by the caller and only the
  // few per cent of kept points are scattered (scattering every byte cost 4x the projection itself)
  if (k) keep[perm ? perm[i] : i] = 1;
}

// ---------------------------------------------------------------------------
// K4: visibility + colour + scores + top-5 over keyframes [f0, f1), same tile
// mask walk as K2.  flags: bit0 load state, bit1 store state, bit2 write packed result.
// ---------------------------------------------------------------------------
// !(r > lim), r = RN(sqrt(s)) the reference's fp64 range (view_culling.cpp:144,157), decided from s where that is certain:
// with L = fl(lim lim) = lim^2 (1 + d), |d| <= 2^-53, and lim > 0,
//   s > L (1 + 2^-40)  =>  sqrt(s) > lim (1 + 2^-42) > lim + ulp(lim) / 2  =>  r >= next(lim) > lim;
//   s < L (1 - 2^-40)  =>  sqrt(s) < lim                                   =>  r <= lim (rounding is monotone);
// in between (2^-39 of the candidates), for lim <= 0 and for non-finite values the square root is taken as before.
__device__ __forceinline__ bool keep_by_depth(float xc, float yc, float zc, double lim) {
  const double s = sumsq64(xc, yc, zc);
  const double L = lim * lim;
  bool keep = s < L * (1.0 - 0x1p-40);
  const bool sure = lim > 0.0 && (keep || s > L * (1.0 + 0x1p-40));
  if (!sure) keep = !(sqrt(s) > lim);
  return keep;
}

struct TopState {
  float *score;
  uint32_t *rgb;
  int32_t *frame;
  int32_t *count;
};

// kMatch: 0 = match mode read from the camera block, else the mode itself (PCP_MATCH_IDENTITY 1 ... see match_constant)
// kOneShot: flags == 4 or 12 (no top-5 state loaded or stored; packed result written in input order, or -- bit 8 -- in the
// sorted order the pass walks, un-permuted by whatever reads it): the usual whole-run call
template <bool kCommon, int kMatch, bool kOneShot>
__global__ __launch_bounds__(kBlock) void k_colour_pass(const float *__restrict__ x, const float *__restrict__ y,
                                                        const float *__restrict__ z, int64_t n, DevCamera cam_in,
                                                        const DevFrame *__restrict__ frames, int32_t f0, int32_t f1,
                                                        const uint32_t *__restrict__ depth, int64_t cells,
                                                        const uint32_t *__restrict__ tile_mask, int32_t words,
                                                        const uint32_t *__restrict__ images, int64_t image_px,
                                                        TopState st, const int32_t *__restrict__ perm,
                                                        uint32_t *__restrict__ rgba, int32_t flags_in,
                                                        const int32_t *__restrict__ tile_order_in,
                                                        const uint32_t *__restrict__ hull_bits_in) {
  DevCamera cam = common_camera<kCommon>(cam_in);
  if constexpr (kMatch == 1) cam.match_mode = PCP_MATCH_IDENTITY;
  if constexpr (kMatch == 2) cam.match_mode = PCP_MATCH_ROUNDTRIP;
  // the common configuration walks the cloud order and has no hull bits (both are null then: the host checks)
  const int32_t *__restrict__ tile_order = kCommon ? nullptr : tile_order_in;
  const uint32_t *__restrict__ hull_bits = kCommon ? nullptr : hull_bits_in;
  const int32_t flags = kOneShot ? (4 | (flags_in & 8)) : flags_in;
  const int64_t j = tile_order ? static_cast<int64_t>(tile_order[blockIdx.x]) * 64 + threadIdx.x
                               : static_cast<int64_t>(xcd_chunked_block()) * kBlock + threadIdx.x;
  const bool live = j < n;
  // a wavefront wholly past the cloud (tail of a 256-thread workgroup) has no tile: its mask words do not exist
  if (!__ballot(live)) return;
  const float px = live ? x[j] : 0.0f, py = live ? y[j] : 0.0f, pz = live ? z[j] : 0.0f;
  const int64_t tile = __builtin_amdgcn_readfirstlane(static_cast<int32_t>(j >> 6));
  // every camera coordinate of this wavefront is finite (divide_xy_by_z)
  const bool finite_sure = cam.frames_bounded != 0 && __all(fabsf(px) <= 0x1p40f && fabsf(py) <= 0x1p40f && fabsf(pz) <= 0x1p40f);
  Top5 t;
  t.init();
  if ((flags & 1) && live) {
    t.s0 = st.score[0 * n + j]; t.s1 = st.score[1 * n + j]; t.s2 = st.score[2 * n + j];
    t.s3 = st.score[3 * n + j]; t.s4 = st.score[4 * n + j];
    t.c0 = st.rgb[0 * n + j]; t.c1 = st.rgb[1 * n + j]; t.c2 = st.rgb[2 * n + j];
    t.c3 = st.rgb[3 * n + j]; t.c4 = st.rgb[4 * n + j];
    t.f0 = st.frame[0 * n + j]; t.f1 = st.frame[1 * n + j]; t.f2 = st.frame[2 * n + j];
    t.f3 = st.frame[3 * n + j]; t.f4 = st.frame[4 * n + j];
    t.count = st.count[j];
  }
  for (int32_t w = f0 >> 5; w <= (f1 - 1) >> 5; ++w) {
    uint32_t todo = range_bits(w, f0, f1);
    if (tile_mask) todo &= tile_mask[tile * words + w];
    todo = __builtin_amdgcn_readfirstlane(todo);
    // PCP_CULL_HPR: bit f of this word = the point is a hull vertex of keyframe f (pcp_hpr.hip): the cull's verdict
    const uint32_t hull_word = (hull_bits && live) ? hull_bits[static_cast<int64_t>(w) * n + j] : 0xffffffffu;
    while (todo) {
      const int32_t f = (w << 5) + __builtin_ctz(todo);
      todo &= todo - 1u;
      const DevFrame &fr = frames[f];
      // the refined masks only keep pairs with a candidate lane: the fp32 rejection test cannot skip the wavefront
      const Projected p = project_point<false>(cam, fr.w2c, px, py, pz, true, finite_sure);
      const bool cand = live && p.pixel >= 0 && (cam.enable_zbuf ? p.cell >= 0 : p.cell != -1) &&
                        ((hull_word >> (f & 31)) & 1u);
      if (cand) {
        // (cells and image_px are below 2^31 -- a cell / pixel index is an int32 --, f is not negative: a 32 x 32-bit product)
        const uint32_t dbits = cam.enable_zbuf ? depth[static_cast<uint64_t>(static_cast<uint32_t>(f)) * static_cast<uint32_t>(cells) + static_cast<uint32_t>(p.cell)] : 0u;
        // A4 keep rule (view_culling.cpp:135-171)
        bool keep = true;
        if (cam.enable_zbuf) keep = keep_by_depth(p.xc, p.yc, p.zc, static_cast<double>(__uint_as_float(dbits)) + cam.slack);
        float sx = p.xc, sy = p.yc, sz = p.zc;
        if (cam.match_mode == PCP_MATCH_ROUNDTRIP && keep) keep = roundtrip_sample(cam, fr, px, py, pz, sx, sy, sz);
        // only samples that pass the keep rule fetch their texel: nearly every fetch is a 64-B sector of its own
        // (neighbouring candidates image a median 4 px apart at 1920x1080, 11 px at 4096x3000), and the depth test
        // drops 51 % / 37 % of the candidates (profiles/r02_candidates.json).  Issuing both gathers up front (one
        // latency instead of two) was slower: colour pass 1.01 -> 0.96 ms at 1920x1080, 1.70 -> 1.48 ms at 4096x3000
        // with the dependent fetch (enough wavefronts hide the latency).
        if (keep) {
          // texel = B | G<<8 | R<<16 | mask<<24; its low 24 bits are 0x00RRGGBB (PointCloudProcessor.cpp:760-762)
          const uint32_t texel = images[static_cast<uint64_t>(static_cast<uint32_t>(f)) * static_cast<uint32_t>(image_px) + static_cast<uint32_t>(p.pixel)];
          t.insert(final_score(sx, sy, sz, fr.px, fr.py, fr.pz), texel & 0xffffffu, f);
        }
      }
    }
  }
  if (!live) return;
  if (flags & 2) {
    st.score[0 * n + j] = t.s0; st.score[1 * n + j] = t.s1; st.score[2 * n + j] = t.s2;
    st.score[3 * n + j] = t.s3; st.score[4 * n + j] = t.s4;
    st.rgb[0 * n + j] = t.c0; st.rgb[1 * n + j] = t.c1; st.rgb[2 * n + j] = t.c2;
    st.rgb[3 * n + j] = t.c3; st.rgb[4 * n + j] = t.c4;
    st.frame[0 * n + j] = t.f0; st.frame[1 * n + j] = t.f1; st.frame[2 * n + j] = t.f2;
    st.frame[3 * n + j] = t.f3; st.frame[4 * n + j] = t.f4;
    st.count[j] = t.count;
  }
  // The packed result.  Straight into input order (flags & 8 clear: PCP_RESULT_UNPERMUTE=0, the form of rounds 2-4) it is a
  // scattered 4-B store per point, each a 32-B partial write in HBM: WRITE_SIZE 316 MB for 40 MB of results
  // (profiles/r04_pmc.json) -- it retires under this VALU-bound kernel for free, but it is 8x the bytes.  flags & 8: coalesced
  // stores in the sorted order the pass walks (40 MB); the un-permutation is then fused into whatever READS the result
  // (round 5, PCP_RESULT_UNPERMUTE=2; slower, see unpermute_mode): the download (k_unpermute on the COPY stream, beside the next step's passes, then the copy engine),
  // the byte-splitting kernel of pcp_colorize's outputs, or -- for a caller that asks for the device array -- a kernel of its own.  PCP_RESULT_UNPERMUTE=1: sorted stores + that kernel right after
  // the pass (45 MB, +35 us per step: the first form of this alternative).  Results identical in all three.
  if (flags & 4) rgba[(flags & 8) ? j : static_cast<int64_t>(perm[j])] = t.finalise();
}

// out[i] = sorted[inv_perm[i]]: coalesced index loads and result stores; the gathers hit a 4n-byte buffer that the
// producing kernel has just written (L2 / Infinity Cache)
__global__ __launch_bounds__(kBlock) void k_unpermute(const uint32_t *__restrict__ sorted, const int32_t *__restrict__ inv_perm,
                                                      int64_t n, uint32_t *__restrict__ out) {
  const int64_t q = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) * 4;
  if (q + 3 < n) {
    const int4 k = *reinterpret_cast<const int4 *>(inv_perm + q);
    *reinterpret_cast<uint4 *>(out + q) = make_uint4(sorted[k.x], sorted[k.y], sorted[k.z], sorted[k.w]);
  } else {
    for (int64_t i = q; i < n; ++i) out[i] = sorted[inv_perm[i]];
  }
}

// finalise from stored state (multi-batch runs)
__global__ __launch_bounds__(kBlock) void k_finalise(int64_t n, TopState st, const int32_t *__restrict__ perm,
                                                     uint32_t *__restrict__ rgba, int32_t sorted_out) {
  const int64_t j = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (j >= n) return;
  Top5 t;
  t.s0 = st.score[0 * n + j]; t.s1 = st.score[1 * n + j]; t.s2 = st.score[2 * n + j];
  t.s3 = st.score[3 * n + j]; t.s4 = st.score[4 * n + j];
  t.c0 = st.rgb[0 * n + j]; t.c1 = st.rgb[1 * n + j]; t.c2 = st.rgb[2 * n + j];
  t.c3 = st.rgb[3 * n + j]; t.c4 = st.rgb[4 * n + j];
  t.f0 = st.frame[0 * n + j]; t.f1 = st.frame[1 * n + j]; t.f2 = st.frame[2 * n + j];
  t.f3 = st.frame[3 * n + j]; t.f4 = st.frame[4 * n + j];
  t.count = st.count[j];
  rgba[sorted_out ? j : static_cast<int64_t>(perm[j])] = t.finalise();  // see k_colour_pass
}

// ---------------------------------------------------------------------------
// arithmetic self-test (pcp_selftest_arithmetic): short exact divisions vs `/`
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t v) {  // splitmix64 finaliser
  v += 0x9e3779b97f4a7c15ull;
  v = (v ^ (v >> 30)) * 0xbf58476d1ce4e5b9ull;
  v = (v ^ (v >> 27)) * 0x94d049bb133111ebull;
  return v ^ (v >> 31);
}
__device__ __forceinline__ bool same_f64(double a, double b) {
  if (a != a && b != b) return true;
  if (a == 0.0 && b == 0.0) return true;  // the sign of a zero quotient is not observable downstream
  return __double_as_longlong(a) == __double_as_longlong(b);
}

__global__ __launch_bounds__(kBlock) void k_selftest_div64(int64_t samples, uint64_t seed,
                                                           unsigned long long *__restrict__ bad) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= samples) return;
  const uint64_t a = mix64(seed ^ static_cast<uint64_t>(i) * 3u), b = mix64(a);
  uint32_t bx = static_cast<uint32_t>(a), by = static_cast<uint32_t>(a >> 32), bz = static_cast<uint32_t>(b);
  const uint32_t mode = static_cast<uint32_t>(b >> 32) & 3u;
  if (mode != 0u) {
    // exponents of a metre-scale scene (mode 0 keeps the raw patterns: every exponent, NaN, inf, denormals)
    const uint32_t span = mode == 1u ? 12u : 40u;
    bx = (bx & 0x807fffffu) | ((127u - span / 2u + (bx >> 23) % span) << 23);
    by = (by & 0x807fffffu) | ((127u - span / 2u + (by >> 23) % span) << 23);
    bz = (bz & 0x807fffffu) | ((127u - span / 2u + (bz >> 23) % span) << 23);
  }
  const float xc = __uint_as_float(bx), yc = __uint_as_float(by), zc = fabsf(__uint_as_float(bz));
  if (!(zc > 0.0f)) return;  // the projection is only entered with z > 0
  double xn, yn;
  divide_xy_by_z(xc, yc, zc, xn, yn);
  const double xr = static_cast<double>(xc) / static_cast<double>(zc);
  const double yr = static_cast<double>(yc) / static_cast<double>(zc);
  if (!same_f64(xn, xr) || !same_f64(yn, yr)) atomicAdd(bad, 1ull);
}

__global__ __launch_bounds__(kBlock) void k_selftest_div32(DevCamera cam, unsigned long long *__restrict__ bad) {
  // 2^32 bit patterns: 2^24 lanes x 256 patterns each
  const uint32_t lane = blockIdx.x * kBlock + threadIdx.x;
  uint32_t wrong = 0u;
  for (uint32_t k = 0; k < 256u; ++k) {
    const float x = __uint_as_float((k << 24) | lane);
    const float a = div_by_ds(cam, x), b = x / cam.ds_f;
    const float ax = fabsf(x);
    if (ax >= 0x1p-40f && ax <= 0x1p60f) {  // the window in which div_by_ds is the quotient itself
      if (!((a != a && b != b) || __float_as_uint(a) == __float_as_uint(b))) ++wrong;
    } else {  // elsewhere: what cull_cell makes of it, along either axis
      if (cell_of_quotient(a, cam.cull_wf) != cell_of_quotient(b, cam.cull_wf)) ++wrong;
      if (cell_of_quotient(a, cam.cull_hf) != cell_of_quotient(b, cam.cull_hf)) ++wrong;
    }
  }
  if (wrong) atomicAdd(bad, static_cast<unsigned long long>(wrong));
}

// ---------------------------------------------------------------------------
// small utility kernels
// ---------------------------------------------------------------------------
// packed result r | g<<8 | b<<16 | has<<24  ->  rgb[3n] (r, g, b) and has[n]
// inv_perm (nullable): `packed` is in sorted order -- point i's word is packed[inv_perm[i]] (the un-permutation fused in)
__global__ __launch_bounds__(kBlock) void k_unpack_result(const uint32_t *__restrict__ packed, int64_t n,
                                                         uint8_t *__restrict__ rgb, uint8_t *__restrict__ has,
                                                         const int32_t *__restrict__ inv_perm) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= n) return;
  const uint32_t v = packed[inv_perm ? inv_perm[i] : i];
  rgb[3 * i + 0] = static_cast<uint8_t>(v & 0xffu);
  rgb[3 * i + 1] = static_cast<uint8_t>((v >> 8) & 0xffu);
  rgb[3 * i + 2] = static_cast<uint8_t>((v >> 16) & 0xffu);
  has[i] = static_cast<uint8_t>(v >> 24);
}

__global__ __launch_bounds__(kBlock) void k_fill_u32(uint32_t *__restrict__ p, int64_t n, uint32_t v) {
  int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (; i < n; i += stride) p[i] = v;
}

// out[perm[j]] = in[j]
__global__ __launch_bounds__(kBlock) void k_scatter_u32(const uint32_t *__restrict__ in,
                                                        const int32_t *__restrict__ perm, int64_t n,
                                                        uint32_t *__restrict__ out) {
  const int64_t j = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (j < n) out[perm[j]] = in[j];
}

// BGR8 rows -> packed texels, keeping the mask byte
// hsv_tables != nullptr: generateColorMap's 8-bit BGR -> HSV -> BGR round trip on the way (pcp_hsv.hpp)
__global__ __launch_bounds__(kBlock) void k_pack_bgr(const uint8_t *__restrict__ bgr, int64_t row_stride, int32_t w,
                                                     int32_t h, uint32_t *__restrict__ texels, int32_t clear_mask,
                                                     const int32_t *__restrict__ hsv_tables, float sat_scale,
                                                     float val_scale) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= static_cast<int64_t>(w) * h) return;
  const int32_t v = static_cast<int32_t>(i / w), u = static_cast<int32_t>(i - static_cast<int64_t>(v) * w);
  const uint8_t *p = bgr + static_cast<int64_t>(v) * row_stride + 3 * u;
  const uint32_t keep = clear_mask ? 0u : (texels[i] & 0xff000000u);
  uint32_t b = p[0], g = p[1], r = p[2];
  if (hsv_tables) hsv_round_trip(hsv_tables, hsv_tables + 256, sat_scale, val_scale, b, g, r);
  texels[i] = keep | b | (g << 8) | (r << 16);
}

// The same, 16 pixels per lane: three 16-B loads (48 B of BGR) -> four 16-B stores (16 texels).  Needs 16-B aligned
// rows and a width that is a multiple of 16.  The source may be device memory or PINNED HOST memory mapped into the
// device's address space: the loads then cross PCIe themselves (no staging copy, no DMA engine, one launch per image),
// 1 KiB per wave-instruction.
__global__ __launch_bounds__(kBlock) void k_pack_bgr16(const uint8_t *__restrict__ bgr, int64_t row_stride, int32_t w,
                                                       int32_t h, uint32_t *__restrict__ texels, int32_t clear_mask,
                                                       const int32_t *__restrict__ hsv_tables, float sat_scale,
                                                       float val_scale) {
  const int32_t groups = w >> 4;  // 16-pixel groups per row
  const int64_t n_groups = static_cast<int64_t>(groups) * h;
  const int64_t g = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  uint4 a, b, c;
  if (row_stride == 3 * static_cast<int64_t>(w)) {
    // tightly packed rows: the workgroup's 256 groups are 12 KiB of consecutive bytes.  Consecutive lanes load
    // consecutive 16-B pieces (every 64-B request fully used -- this matters when the bytes cross PCIe) and the
    // pieces are dealt out through LDS.
    __shared__ uint4 piece[3 * kBlock];
    const int64_t base = static_cast<int64_t>(blockIdx.x) * (3 * kBlock);
    const int64_t pieces = 3 * n_groups;
    const uint4 *src = reinterpret_cast<const uint4 *>(bgr);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int64_t i = base + q * kBlock + threadIdx.x;
      if (i < pieces) piece[q * kBlock + threadIdx.x] = src[i];
    }
    __syncthreads();
    if (g >= n_groups) return;
    a = piece[3 * threadIdx.x];
    b = piece[3 * threadIdx.x + 1];
    c = piece[3 * threadIdx.x + 2];
  } else {
    if (g >= n_groups) return;
    const int32_t vr = static_cast<int32_t>(g / groups), gr = static_cast<int32_t>(g - static_cast<int64_t>(vr) * groups);
    const uint4 *src = reinterpret_cast<const uint4 *>(bgr + static_cast<int64_t>(vr) * row_stride + static_cast<int64_t>(gr) * 48);
    a = src[0];
    b = src[1];
    c = src[2];
  }
  const int32_t v = static_cast<int32_t>(g / groups), gu = static_cast<int32_t>(g - static_cast<int64_t>(v) * groups);
  const uint32_t in[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w};
  uint4 *dst = reinterpret_cast<uint4 *>(texels + static_cast<int64_t>(v) * w + static_cast<int64_t>(gu) * 16);
  uint32_t out[16];
#pragma unroll
  for (int q = 0; q < 4; ++q) {  // 4 pixels = 3 dwords: B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3
    const uint32_t d0 = in[3 * q], d1 = in[3 * q + 1], d2 = in[3 * q + 2];
    out[4 * q + 0] = d0 & 0xffffffu;
    out[4 * q + 1] = (d0 >> 24) | ((d1 & 0xffffu) << 8);
    out[4 * q + 2] = (d1 >> 16) | ((d2 & 0xffu) << 16);
    out[4 * q + 3] = d2 >> 8;
  }
  if (hsv_tables) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      uint32_t bb = out[k] & 0xffu, gg = (out[k] >> 8) & 0xffu, rr = out[k] >> 16;
      hsv_round_trip(hsv_tables, hsv_tables + 256, sat_scale, val_scale, bb, gg, rr);
      out[k] = bb | (gg << 8) | (rr << 16);
    }
  }
  if (!clear_mask) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint4 old = dst[q];
      out[4 * q + 0] |= old.x & 0xff000000u;
      out[4 * q + 1] |= old.y & 0xff000000u;
      out[4 * q + 2] |= old.z & 0xff000000u;
      out[4 * q + 3] |= old.w & 0xff000000u;
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) dst[q] = make_uint4(out[4 * q], out[4 * q + 1], out[4 * q + 2], out[4 * q + 3]);
}

__global__ __launch_bounds__(kBlock) void k_pack_mask(const uint8_t *__restrict__ gray, int64_t row_stride, int32_t w,
                                                      int32_t h, uint32_t *__restrict__ texels) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= static_cast<int64_t>(w) * h) return;
  const int32_t v = static_cast<int32_t>(i / w), u = static_cast<int32_t>(i - static_cast<int64_t>(v) * w);
  texels[i] = (texels[i] & 0x00ffffffu) | (static_cast<uint32_t>(gray[static_cast<int64_t>(v) * row_stride + u]) << 24);
}

// ---- ordered compaction of a byte flag array (tile = 1024 flags) -------------
constexpr int kTile = 1024;

__device__ __forceinline__ int32_t block_exclusive_scan(int32_t v, int32_t *total) {
  __shared__ int32_t wave_sum[kBlock / 64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int32_t incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int32_t t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wave_sum[wid] = incl;
  __syncthreads();
  int32_t base = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < kBlock / 64; ++k) {
    if (k < wid) base += wave_sum[k];
    tot += wave_sum[k];
  }
  __syncthreads();
  *total = tot;
  return base + incl - v;
}

__global__ __launch_bounds__(kBlock) void k_tile_count(const uint8_t *__restrict__ flags, int64_t n,
                                                       int32_t *__restrict__ tile_count) {
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kTile + threadIdx.x * 4;
  int32_t c = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (base + k < n) c += flags[base + k] ? 1 : 0;
  int32_t total;
  (void)block_exclusive_scan(c, &total);
  if (threadIdx.x == 0) tile_count[blockIdx.x] = total;
}

// single block: exclusive scan of tile counts in place, grand total to *total (scan_single_block, pcp_scan.hpp)
constexpr int kScanTilesBlock = kScanSingle;
__global__ __launch_bounds__(kScanTilesBlock) void k_scan_tiles(int32_t *__restrict__ tile_count, int64_t tiles,
                                                                unsigned long long *__restrict__ total) {
  scan_single_block(tile_count, tiles, total);
}

__global__ __launch_bounds__(kBlock) void k_tile_scatter(const uint8_t *__restrict__ flags, int64_t n,
                                                         const int32_t *__restrict__ tile_offset,
                                                         int32_t *__restrict__ out_index, int64_t capacity) {
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kTile + threadIdx.x * 4;
  int32_t c = 0;
  bool fl[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    fl[k] = base + k < n && flags[base + k];
    c += fl[k] ? 1 : 0;
  }
  int32_t total;
  int64_t pos = tile_offset[blockIdx.x] + block_exclusive_scan(c, &total);
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (fl[k]) {
      if (pos < capacity) out_index[pos] = static_cast<int32_t>(base + k);
      ++pos;
    }
}

// records of the visible points of one keyframe, by compacted index list
__global__ __launch_bounds__(kBlock) void k_gather_visible(const float *__restrict__ x, const float *__restrict__ y,
                                                           const float *__restrict__ z, DevCamera cam, DevFrame fr,
                                                           const int32_t *__restrict__ index, int64_t m,
                                                           const uint32_t *__restrict__ image, int32_t has_image,
                                                           int32_t has_mask, uint32_t *__restrict__ out_rgbm,
                                                           float *__restrict__ out_cam, float *__restrict__ out_world) {
  const int64_t k = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (k >= m) return;
  const int32_t i = index[k];
  const Projected p = project_point(cam, fr.w2c, x[i], y[i], z[i]);
  uint32_t r = 0, g = 0, b = 0, mv = 0;
  if (p.pixel >= 0) {
    const uint32_t texel = image[p.pixel];
    if (has_image) {
      b = texel & 0xffu;
      g = (texel >> 8) & 0xffu;
      r = (texel >> 16) & 0xffu;
    }
    if (has_mask) {
      mv = texel >> 24;  // generateSegmentMap, PointCloudProcessor.cpp:803-810
      if (mv == 255u) {
        r = 255u;
        g = 0u;
        b = 0u;
      }
    }
  }
  out_rgbm[k] = r | (g << 8) | (b << 16) | (mv << 24);
  if (out_cam) {
    out_cam[3 * k + 0] = p.xc;
    out_cam[3 * k + 1] = p.yc;
    out_cam[3 * k + 2] = p.zc;
  }
  if (out_world) {  // transformPointCloud(c2w), PointCloudProcessor.cpp:549,555
    float wx, wy, wz;
    xform(fr.c2w, p.xc, p.yc, p.zc, wx, wy, wz);
    out_world[3 * k + 0] = wx;
    out_world[3 * k + 1] = wy;
    out_world[3 * k + 2] = wz;
  }
}

// ---------------------------------------------------------------------------
// host helpers
// ---------------------------------------------------------------------------
static inline int64_t cells_of(const pcp_context *ctx) { return static_cast<int64_t>(ctx->dcam.mw) * ctx->dcam.mh; }
static inline size_t plane_of(const pcp_context *ctx) { return (static_cast<size_t>(ctx->n) + 3) & ~size_t(3); }
static inline uint32_t blocks_for(int64_t n) { return static_cast<uint32_t>(std::max<int64_t>(1, div_up(n, kBlock))); }

static int check_ready(pcp_context *ctx, const char *who, bool need_frames) {
  if (!ctx) return PCP_ERR_INVALID;
  if (!ctx->have_camera) return set_error(ctx, PCP_ERR_STATE, "%s: pcp_set_camera has not been called", who);
  if (ctx->xyz.p == nullptr && ctx->n == 0 && ctx->xyz.count == 0)
    return set_error(ctx, PCP_ERR_STATE, "%s: no cloud uploaded", who);
  if (need_frames && ctx->n_frames <= 0) return set_error(ctx, PCP_ERR_STATE, "%s: pcp_set_frames has not been called", who);
  hipError_t e = hipSetDevice(ctx->device);
  if (e != hipSuccess) return set_error(ctx, PCP_ERR_DEVICE, "%s: hipSetDevice failed: %s", who, hipGetErrorString(e));
  return PCP_OK;
}

static int check_frame(pcp_context *ctx, const char *who, int32_t frame) {
  if (frame < 0 || frame >= ctx->n_frames)
    return set_error(ctx, PCP_ERR_RANGE, "%s: keyframe %d out of range (0..%d)", who, frame, ctx->n_frames - 1);
  return PCP_OK;
}

static int fill_u32(pcp_context *ctx, uint32_t *p, int64_t n, uint32_t v) {
  if (n <= 0) return PCP_OK;
  LaunchTimer t(ctx, PCP_K_MISC);
  const uint32_t grid = static_cast<uint32_t>(std::min<int64_t>(div_up(n, kBlock), 8192));
  hipLaunchKernelGGL(k_fill_u32, dim3(grid), dim3(kBlock), 0, ctx->stream, p, n, v);
  PCP_HIP_TRY(ctx, hipGetLastError());
  return PCP_OK;
}

static int ensure_images(pcp_context *ctx) {
  const size_t px = static_cast<size_t>(ctx->dcam.img_w) * ctx->dcam.img_h;
  const size_t need = px * static_cast<size_t>(ctx->n_frames) + 4;
  if (ctx->images.count < need) {
    PCP_HIP_TRY(ctx, ctx->images.ensure(need));
    PCP_HIP_TRY(ctx, hipMemsetAsync(ctx->images.p, 0, need * 4, ctx->stream));
    ctx->image_set.assign(static_cast<size_t>(ctx->n_frames), 0);
    ctx->mask_set.assign(static_cast<size_t>(ctx->n_frames), 0);
  }
  return PCP_OK;
}

int wait_images(pcp_context *ctx, int32_t f0, int32_t f1) {
  ctx->texels_touched = true;
  if (ctx->image_pending.empty()) return PCP_OK;
  // every lane is in order: per lane, waiting for the range's most recently queued keyframe covers the others
  for (int lane = 0; lane < pcp_context::kUploadLanes; ++lane) {
    int32_t last = -1;
    for (int32_t f = std::max(f0, 0); f < f1 && f < static_cast<int32_t>(ctx->image_pending.size()); ++f) {
      const size_t sf = static_cast<size_t>(f);
      if (ctx->image_pending[sf] && ctx->image_lane[sf] == lane &&
          (last < 0 || ctx->image_seq[sf] > ctx->image_seq[static_cast<size_t>(last)]))
        last = f;
    }
    if (last < 0) continue;
    PCP_HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->image_event[static_cast<size_t>(last)], 0));
    const uint64_t seq = ctx->image_seq[static_cast<size_t>(last)];
    for (size_t f = 0; f < ctx->image_pending.size(); ++f)
      if (ctx->image_pending[f] && ctx->image_lane[f] == lane && ctx->image_seq[f] <= seq) ctx->image_pending[f] = 0;
  }
  return PCP_OK;
}

// tile_work = the keyframes every tile walks in words [w0, w1) of its mask, and the tiles sorted by it, heaviest first, into
// `order` (ctx->work_hist: histogram | cursors, zeroed by k_group_mask_flat earlier in the stage)
static int sort_tiles_by_work(pcp_context *ctx, int32_t w0, int32_t w1, int32_t *order) {
  const uint32_t sort_blocks = static_cast<uint32_t>(div_up(ctx->n_tiles, static_cast<int64_t>(kWorkPerBlock)));
  hipLaunchKernelGGL(k_tile_work_hist, dim3(sort_blocks), dim3(kBlock), 0, ctx->stream, ctx->tile_mask.p, ctx->n_tiles, w0, w1,
                     ctx->mask_words, ctx->tile_work.p, ctx->work_hist.p);
  hipLaunchKernelGGL(k_work_scatter, dim3(sort_blocks), dim3(kBlock), 0, ctx->stream, ctx->tile_work.p, ctx->n_tiles,
                     ctx->work_hist.p, ctx->work_hist.p + kWorkBins, order);
  PCP_HIP_TRY(ctx, hipGetLastError());
  return PCP_OK;
}

static int ensure_hull_bits(pcp_context *ctx) {
  const size_t words = static_cast<size_t>((ctx->n_frames + 31) / 32);
  const size_t need = words * static_cast<size_t>(ctx->n) + 4;
  if (ctx->hull_bits.count < need || ctx->hull_valid.size() != static_cast<size_t>(ctx->n_frames)) {
    PCP_HIP_TRY(ctx, ctx->hull_bits.ensure(need));
    ctx->hull_valid.assign(static_cast<size_t>(ctx->n_frames), 0);
  }
  return PCP_OK;
}

static int ensure_depth(pcp_context *ctx) {
  const size_t need = static_cast<size_t>(cells_of(ctx)) * ctx->n_frames + 4;
  if (ctx->depth.count < need) {
    PCP_HIP_TRY(ctx, ctx->depth.ensure(need));
    std::fill(ctx->depth_valid.begin(), ctx->depth_valid.end(), uint8_t(0));
  }
  const int32_t words = (ctx->n_frames + 31) / 32;
  const size_t need_mask = static_cast<size_t>(words) * static_cast<size_t>(std::max<int64_t>(ctx->n_tiles, 1)) + 4;
  if (ctx->tile_mask.count < need_mask || ctx->mask_words != words) {
    PCP_HIP_TRY(ctx, ctx->tile_mask.ensure(need_mask));
    PCP_HIP_TRY(ctx, ctx->tile_inside.ensure(need_mask));
    PCP_HIP_TRY(ctx, ctx->tile_work.ensure(static_cast<size_t>(std::max<int64_t>(ctx->n_tiles, 1)) + 4));
    PCP_HIP_TRY(ctx, ctx->tile_order.ensure(static_cast<size_t>(std::max<int64_t>(ctx->n_tiles, 1)) + 4));
    PCP_HIP_TRY(ctx, ctx->work_hist.ensure(2 * kWorkBins + 4));
    ctx->tile_order_live = false;
    ctx->mask_words = words;
    std::fill(ctx->depth_valid.begin(), ctx->depth_valid.end(), uint8_t(0));
  }
  return PCP_OK;
}

// PCP_RESULT_UNPERMUTE: 0 (default) = the colour pass scatters its result into input order; 1 = sorted stores + an un-permuting
// kernel right after the pass; 2 = sorted stores, un-permuted by whatever reads the result.  Read per call: tests flip it
// inside one process.  Measured at C3 in round 5 (100 steps, driver protocol): 0: 1.345 ms per step; 1: 1.56 ms (the gather
// of 10 M random words is a 0.17 ms kernel); 2: 1.45 ms (the same kernel on the copy stream, beside the next step's passes,
// then the copy engine; with the kernel writing the pinned buffer over PCIe itself: 2.06 ms).  The un-permutation is 10 M
// random 4-byte accesses wherever it is put; as scattered stores of the VALU-bound pass it costs 0.045 ms and 8x the bytes.
static int unpermute_mode() {
  const char *e = std::getenv("PCP_RESULT_UNPERMUTE");
  if (!e || !e[0]) return 0;
  return e[0] == '0' ? 0 : (e[0] == '1' ? 1 : 2);
}
static bool unpermute_results() { return unpermute_mode() == 1; }

static int ensure_state(pcp_context *ctx) {
  const size_t sn = static_cast<size_t>(ctx->n);
  PCP_HIP_TRY(ctx, ctx->top_score.ensure(kTopM * sn + 4));
  PCP_HIP_TRY(ctx, ctx->top_rgb.ensure(kTopM * sn + 4));
  PCP_HIP_TRY(ctx, ctx->top_frame.ensure(kTopM * sn + 4));
  PCP_HIP_TRY(ctx, ctx->view_count.ensure(sn + 4));
  PCP_HIP_TRY(ctx, ctx->rgba2[0].ensure(sn + 4));
  PCP_HIP_TRY(ctx, ctx->rgba2[1].ensure(sn + 4));
  PCP_HIP_TRY(ctx, ctx->rgba_sorted.ensure(sn + 4));
  return PCP_OK;
}

// ordered compaction of a byte flag array into an index list (nullable); returns the count
int compact_flags(pcp_context *ctx, const uint8_t *flags, int64_t n, int32_t *out_index, int64_t capacity,
                  int64_t *count) {
  const int64_t tiles = std::max<int64_t>(1, div_up(n, kTile));
  PCP_HIP_TRY(ctx, ctx->s_tiles.ensure(static_cast<size_t>(tiles) + 4));
  PCP_HIP_TRY(ctx, ctx->s_counter.ensure(4));
  {
    LaunchTimer t(ctx, PCP_K_MISC);
    hipLaunchKernelGGL(k_tile_count, dim3(static_cast<uint32_t>(tiles)), dim3(kBlock), 0, ctx->stream, flags, n,
                       ctx->s_tiles.p);
    hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(kScanTilesBlock), 0, ctx->stream, ctx->s_tiles.p, tiles, ctx->s_counter.p);
    if (out_index)
      hipLaunchKernelGGL(k_tile_scatter, dim3(static_cast<uint32_t>(tiles)), dim3(kBlock), 0, ctx->stream, flags, n,
                         ctx->s_tiles.p, out_index, capacity);
    PCP_HIP_TRY(ctx, hipGetLastError());
  }
  unsigned long long total = 0;
  unsigned long long *dst = ctx->readback ? static_cast<unsigned long long *>(ctx->readback) : &total;  // pinned: no staging
  PCP_HIP_TRY(ctx, hipMemcpyAsync(dst, ctx->s_counter.p, sizeof(total), hipMemcpyDeviceToHost, ctx->stream));
  PCP_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  *count = static_cast<int64_t>(*dst);
  return PCP_OK;
}

// depth map of one keyframe into ctx->s_u32 (scratch)
static int single_frame_depth(pcp_context *ctx, int32_t frame) {
  const int64_t cells = cells_of(ctx);
  PCP_HIP_TRY(ctx, ctx->s_u32.ensure(static_cast<size_t>(cells) + 4));
  if (ctx->depth_from_batch) {
    // one index shard of a larger map: the batched maps, MIN-merged across the shards by the caller, are the maps
    // of the whole cloud (pcp_set_depth_source)
    if (!ctx->depth.p || static_cast<size_t>(frame) >= ctx->depth_valid.size() || !ctx->depth_valid[static_cast<size_t>(frame)])
      return set_error(ctx, PCP_ERR_STATE, "PCP_DEPTH_BATCHED: pcp_depth_pass has not covered keyframe %d", frame);
    PCP_HIP_TRY(ctx, hipMemcpyAsync(ctx->s_u32.p, ctx->depth.p + static_cast<int64_t>(frame) * cells,
                                    static_cast<size_t>(cells) * 4, hipMemcpyDeviceToDevice, ctx->stream));
    return PCP_OK;
  }
  if (ctx->n > 0 && ctx->dcam.enable_zbuf) {
    const size_t plane = plane_of(ctx);
    PCP_HIP_TRY(ctx, ctx->depth_sq.ensure(static_cast<size_t>(cells) + 4));
    PCP_HIP_TRY(ctx, hipMemsetAsync(ctx->depth_sq.p, 0x7f, static_cast<size_t>(cells) * 8, ctx->stream));
    LaunchTimer t(ctx, PCP_K_DEPTH);
    hipLaunchKernelGGL(k_depth_pass<false>, dim3(blocks_for(ctx->n)), dim3(kBlock), 0, ctx->stream, ctx->sxyz.p,
                       ctx->sxyz.p + plane, ctx->sxyz.p + 2 * plane, ctx->n, ctx->dcam, ctx->frames.p, frame, frame + 1,
                       ctx->depth_sq.p, cells, frame, static_cast<uint32_t *>(nullptr),
                       static_cast<const uint32_t *>(nullptr), 0, static_cast<const int32_t *>(nullptr));
    hipLaunchKernelGGL(k_depth_finish, dim3(blocks_for(cells)), dim3(kBlock), 0, ctx->stream, ctx->depth_sq.p, cells,
                       ctx->s_u32.p);
    PCP_HIP_TRY(ctx, hipGetLastError());
    return PCP_OK;
  }
  return fill_u32(ctx, ctx->s_u32.p, cells, kFltMaxBits);
}

// clears the flags of points without a colour pixel (generateColorMap's bounds, PointCloudProcessor.cpp:748-754)
__global__ __launch_bounds__(kBlock) void k_require_pixel(const float *__restrict__ x, const float *__restrict__ y,
                                                          const float *__restrict__ z, int64_t n, DevCamera cam, DevFrame fr,
                                                          uint8_t *__restrict__ keep) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= n || !keep[i]) return;
  if (project_point(cam, fr.w2c, x[i], y[i], z[i]).pixel < 0) keep[i] = 0;
}

// PCP_CULL_HPR, whole run: bit `bit` of word[j] = keep flag of the point at place j of the sorted order.  The keep flags
// are in input order and few of them are set (the hull vertices among the candidates of one keyframe: 1-4 % of the map), so
// the plane's bit is cleared with a streaming pass (or the whole plane with a memset when its 32 keyframes are all being
// produced) and the set flags are scattered through inv_perm -- the first form read keep[perm[j]] for every place: a random
// byte gather over the whole map, 117 us per keyframe at 10 M points.
__global__ __launch_bounds__(kBlock) void k_hull_clear(uint32_t *__restrict__ word, int64_t n, uint32_t bit) {
  const int64_t j = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (j < n) word[j] &= ~bit;
}

__global__ __launch_bounds__(kBlock) void k_hull_set(const uint8_t *__restrict__ keep, const int32_t *__restrict__ inv_perm,
                                                     int64_t n, uint32_t *__restrict__ word, uint32_t bit) {
  const int64_t i0 = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) * 4;
  if (i0 >= n) return;
  uint32_t four = 0;
  if (i0 + 4 <= n) {
    four = *reinterpret_cast<const uint32_t *>(keep + i0);
  } else {
    for (int k = 0; k < 4 && i0 + k < n; ++k) four |= static_cast<uint32_t>(keep[i0 + k]) << (8 * k);
  }
  if (four == 0) return;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if ((four >> (8 * k)) & 0xffu) atomicOr(word + inv_perm[i0 + k], bit);
}

// ctx->s_keep (input order) -> bit f of the hull bits; plane_clear: the plane of f has just been zeroed
static int store_hull_bits(pcp_context *ctx, int32_t f, bool plane_clear) {
  uint32_t *plane = ctx->hull_bits.p + static_cast<size_t>(f >> 5) * static_cast<size_t>(ctx->n);
  const uint32_t bit = 1u << (f & 31);
  LaunchTimer t(ctx, PCP_K_HPR);
  if (!plane_clear) hipLaunchKernelGGL(k_hull_clear, dim3(blocks_for(ctx->n)), dim3(kBlock), 0, ctx->stream, plane, ctx->n, bit);
  hipLaunchKernelGGL(k_hull_set, dim3(blocks_for((ctx->n + 3) / 4)), dim3(kBlock), 0, ctx->stream, ctx->s_keep.p, ctx->inv_perm.p,
                     ctx->n, plane, bit);
  PCP_HIP_TRY(ctx, hipGetLastError());
  return PCP_OK;
}

// the inverse: keep flags (input order) of one keyframe from the hull bits (pcp_hull_flags_import on an index shard)
__global__ __launch_bounds__(kBlock) void k_flags_from_hull_bits(const uint32_t *__restrict__ word, uint32_t bit,
                                                                 const int32_t *__restrict__ perm, int64_t n,
                                                                 uint8_t *__restrict__ keep) {
  const int64_t j = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (j < n) keep[perm[j]] = (word[j] & bit) ? 1 : 0;
}

// ViewCulling::cull of one keyframe as byte flags in input order (ctx->s_keep): the z-buffer routine's pass 2 against the
// map in ctx->s_u32 (single_frame_depth), or hidden_points_removal (candidate filter, then the hull: pcp_hpr.hip).
// require_pixel: only points that generateColorMap can colour.
static int frame_keep_flags(pcp_context *ctx, int32_t frame, bool require_pixel) {
  const int64_t n = ctx->n;
  PCP_HIP_TRY(ctx, ctx->s_keep.ensure(static_cast<size_t>(n) + 16));
  if (n == 0) return PCP_OK;
  const size_t plane = plane_of(ctx);
  const bool hull = ctx->cull.cull_mode == PCP_CULL_HPR;
  const bool have_bits = hull && ctx->hull_bits.p && static_cast<size_t>(frame) < ctx->hull_valid.size() && ctx->hull_valid[static_cast<size_t>(frame)];
  if (hull && (ctx->depth_from_batch || have_bits)) {
    // the keyframe's verdicts are in the whole-run bits already: imported (one index shard of a larger map: the hull was taken
    // over the WHOLE map elsewhere), or left there by this context's own hull pass (pcp_depth_pass) -- nothing is recomputed
    if (!have_bits)
      return set_error(ctx, PCP_ERR_STATE, "PCP_CULL_HPR on an index shard: pcp_hull_flags_import has not covered keyframe %d", frame);
    // pcp_hpr_stats describes the keyframe asked for LAST: this one was not recomputed, so there are no tallies of it (the
    // ones of the whole-run pass's last lane would be another keyframe's): zeros, candidates = -1
    std::memset(ctx->hpr_stats, 0, sizeof(ctx->hpr_stats));
    ctx->hpr_stats[9] = -1;
    ctx->hpr_stats_pending = false;
    LaunchTimer t(ctx, PCP_K_VISIBILITY);
    hipLaunchKernelGGL(k_flags_from_hull_bits, dim3(blocks_for(n)), dim3(kBlock), 0, ctx->stream,
                       ctx->hull_bits.p + static_cast<size_t>(frame >> 5) * static_cast<size_t>(n), 1u << (frame & 31), ctx->perm.p,
                       n, ctx->s_keep.p);
    if (require_pixel)
      hipLaunchKernelGGL(k_require_pixel, dim3(blocks_for(n)), dim3(kBlock), 0, ctx->stream, ctx->xyz.p, ctx->xyz.p + plane,
                         ctx->xyz.p + 2 * plane, n, ctx->dcam, ctx->hframes[static_cast<size_t>(frame)], ctx->s_keep.p);
    PCP_HIP_TRY(ctx, hipGetLastError());
    return PCP_OK;
  }
  if (!hull) {
    LaunchTimer t(ctx, PCP_K_VISIBILITY);
    PCP_HIP_TRY(ctx, hipMemsetAsync(ctx->s_keep.p, 0, static_cast<size_t>(n), ctx->stream));
    hipLaunchKernelGGL(k_visibility, dim3(blocks_for(n)), dim3(kBlock), 0, ctx->stream, ctx->sxyz.p,
                       ctx->sxyz.p + plane, ctx->sxyz.p + 2 * plane, n, ctx->dcam,
                       ctx->hframes[static_cast<size_t>(frame)], ctx->s_u32.p, ctx->perm.p, ctx->s_keep.p,
                       require_pixel ? 1 : 0);
    PCP_HIP_TRY(ctx, hipGetLastError());
    return PCP_OK;
  }
  // the hull is taken over EVERY candidate (view_culling.cpp:276-288 knows nothing of the image's own size; pcp_hpr.hip
  // finds them itself); the colour bounds apply to what it keeps
  int rc = hpr_run(ctx, frame, ctx->s_keep.p, nullptr, 0u);
  if (rc != PCP_OK) return rc;
  if (require_pixel) {
    LaunchTimer t(ctx, PCP_K_VISIBILITY);
    hipLaunchKernelGGL(k_require_pixel, dim3(blocks_for(n)), dim3(kBlock), 0, ctx->stream, ctx->xyz.p, ctx->xyz.p + plane,
                       ctx->xyz.p + 2 * plane, n, ctx->dcam, ctx->hframes[static_cast<size_t>(frame)], ctx->s_keep.p);
    PCP_HIP_TRY(ctx, hipGetLastError());
  }
  return PCP_OK;
}

// ViewCulling::cull for one keyframe, device side: ordered index list of the kept points into
// `d_index` (capacity entries), count to *count.  Used by pcp_cull_frame-like paths and the NID stage.
int cull_frame_indices(pcp_context *ctx, int32_t frame, int32_t *d_index, int64_t capacity, int64_t *count) {
  const int64_t n = ctx->n;
  *count = 0;
  if (n == 0) return PCP_OK;
  int rc = single_frame_depth(ctx, frame);
  if (rc != PCP_OK) return rc;
  if ((rc = frame_keep_flags(ctx, frame, false)) != PCP_OK) return rc;
  return compact_flags(ctx, ctx->s_keep.p, n, d_index, capacity, count);
}

}  // namespace pcp

namespace pcp {
// (pcp_create loads every code object of the library up front: see preload_code_objects in pcp_context.hip)
hipError_t preload_colour() {
  hipFuncAttributes a;
  return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&k_fill_u32));
}
}  // namespace pcp

using namespace pcp;

extern "C" {

// Both forms run on the upload stream: copy into the staging buffer, pack kernel (with the HSV round trip when
// pcp_set_image_adjust enabled it), event.  `bgr` may be a host pointer (pinned for a real overlap) or a device
// pointer (hipMemcpyDefault: e.g. frames all-gathered over xGMI by the multi-GPU driver).
// block_bytes > 0: `bgr` is the first of block_frames keyframes (block_stride bytes apart) in pinned host memory; the whole
// block is copied into the lane's staging buffer by one DMA and every keyframe of it is packed from there.
static int upload_image_impl(pcp_context *ctx, const char *who, int32_t frame, const uint8_t *bgr, int64_t row_stride_bytes,
                             bool wait, size_t block_bytes = 0, int64_t block_stride = 0, int32_t block_frames = 1) {
  int rc = check_ready(ctx, who, true);
  if (rc != PCP_OK) return rc;
  if ((rc = check_frame(ctx, who, frame)) != PCP_OK) return rc;
  const int32_t w = ctx->dcam.img_w, h = ctx->dcam.img_h;
  if (!bgr || row_stride_bytes < 3 * static_cast<int64_t>(w))
    return set_error(ctx, PCP_ERR_INVALID, "%s: NULL image or row stride < 3*width", who);
  const bool fresh = ctx->images.count < static_cast<size_t>(w) * h * static_cast<size_t>(ctx->n_frames) + 4;
  if ((rc = ensure_images(ctx)) != PCP_OK) return rc;
  if (!ctx->upload_stream[0]) {
    for (int l = 0; l < pcp_context::kUploadLanes; ++l)
      PCP_HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->upload_stream[l], hipStreamNonBlocking));
    PCP_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->texels_idle, hipEventDisableTiming));
  }
  const size_t nf = static_cast<size_t>(ctx->n_frames);
  if (ctx->image_event.size() < nf) {
    const size_t old = ctx->image_event.size();
    ctx->image_event.resize(nf, nullptr);
    for (size_t k = old; k < nf; ++k) PCP_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->image_event[k], hipEventDisableTiming));
  }
  if (ctx->image_pending.size() != nf) {
    ctx->image_pending.assign(nf, 0);
    ctx->image_lane.assign(nf, 0);
    ctx->image_seq.assign(nf, 0);
  }
  const size_t sf = static_cast<size_t>(frame);
  // a keyframe uploaded again goes to the lane of its previous upload while that one may still be in flight (two
  // lanes writing the same texels would race); otherwise the lanes take turns
  const int lane = (ctx->image_pending[sf] && block_bytes == 0) ? ctx->image_lane[sf]
                                                                : static_cast<int>(ctx->upload_turn % pcp_context::kUploadLanes);
  if (block_bytes > 0)  // a block takes one lane: uploads of its keyframes still in flight on the other lane come first
    for (int32_t k = 0; k < block_frames; ++k)
      if (ctx->image_pending[sf + static_cast<size_t>(k)] && ctx->image_lane[sf + static_cast<size_t>(k)] != lane)
        PCP_HIP_TRY(ctx, hipStreamWaitEvent(ctx->upload_stream[lane], ctx->image_event[sf + static_cast<size_t>(k)], 0));
  // kernels of the compute stream that read or write texels (a colour pass still sampling the previous image of this
  // keyframe, a mask pack, the clearing of a fresh buffer) come first, on every lane
  if (fresh || ctx->texels_touched) {
    PCP_HIP_TRY(ctx, hipEventRecord(ctx->texels_idle, ctx->stream));
    for (int l = 0; l < pcp_context::kUploadLanes; ++l) ctx->lane_must_wait[l] = true;
    ctx->texels_touched = false;
  }
  if (ctx->lane_must_wait[lane]) {
    PCP_HIP_TRY(ctx, hipStreamWaitEvent(ctx->upload_stream[lane], ctx->texels_idle, 0));
    ctx->lane_must_wait[lane] = false;
  }
  hipStream_t us = ctx->upload_stream[lane];
  const size_t bytes = static_cast<size_t>(row_stride_bytes) * h;
  const int64_t px = static_cast<int64_t>(w) * h;
  // Where do the pack kernel's loads go?  Device memory and pinned (device-mapped) host memory are read in place;
  // pageable host memory goes through the lane's staging buffer.
  const uint8_t *src = nullptr;
  if (block_bytes > 0) {
    PCP_HIP_TRY(ctx, ctx->upload_stage[lane].ensure(block_bytes + 16));
    PCP_HIP_TRY(ctx, hipMemcpyAsync(ctx->upload_stage[lane].p, bgr, block_bytes, hipMemcpyHostToDevice, us));
    src = ctx->upload_stage[lane].p;
  } else {
    static const bool direct = [] {
      const char *e = std::getenv("PCP_UPLOAD_DIRECT");
      return !(e && e[0] == '0');
    }();
    hipPointerAttribute_t attr{};
    if (hipPointerGetAttributes(&attr, bgr) == hipSuccess &&
        (attr.type == hipMemoryTypeDevice || (direct && attr.type == hipMemoryTypeHost)) && attr.devicePointer) {
      src = static_cast<const uint8_t *>(attr.devicePointer);
      if (attr.type == hipMemoryTypeDevice) {
        // bytes produced on the device (a collective on the context's stream): this lane starts after that work
        PCP_HIP_TRY(ctx, hipEventRecord(ctx->texels_idle, ctx->stream));
        PCP_HIP_TRY(ctx, hipStreamWaitEvent(us, ctx->texels_idle, 0));
      }
    } else {
      (void)hipGetLastError();  // an unregistered host pointer is not an error here
    }
  }
  if (!src) {
    PCP_HIP_TRY(ctx, ctx->upload_stage[lane].ensure(bytes + 16));
    PCP_HIP_TRY(ctx, hipMemcpyAsync(ctx->upload_stage[lane].p, bgr, bytes, hipMemcpyDefault, us));
    src = ctx->upload_stage[lane].p;
  }
  const int32_t *tables = ctx->adjust_images ? ctx->hsv_tables.p : static_cast<const int32_t *>(nullptr);
  ++ctx->upload_turn;  // one turn of the lanes per call (a block is one call)
  for (int32_t k = 0; k < block_frames; ++k) {
    const int32_t fk = frame + k;
    const uint8_t *sk = src + static_cast<int64_t>(k) * block_stride;
    uint32_t *dst = ctx->images.p + static_cast<int64_t>(fk) * px;
    const int32_t clear_mask = ctx->mask_set[fk] ? 0 : 1;
    if ((w & 15) == 0 && (row_stride_bytes & 15) == 0 && (reinterpret_cast<uintptr_t>(sk) & 15u) == 0)
      hipLaunchKernelGGL(k_pack_bgr16, dim3(blocks_for(px / 16)), dim3(kBlock), 0, us, sk, row_stride_bytes, w, h, dst,
                         clear_mask, tables, ctx->saturation_scale, ctx->brightness_scale);
    else
      hipLaunchKernelGGL(k_pack_bgr, dim3(blocks_for(px)), dim3(kBlock), 0, us, sk, row_stride_bytes, w, h, dst, clear_mask,
                         tables, ctx->saturation_scale, ctx->brightness_scale);
    PCP_HIP_TRY(ctx, hipGetLastError());
    PCP_HIP_TRY(ctx, hipEventRecord(ctx->image_event[static_cast<size_t>(fk)], us));
    ctx->image_pending[static_cast<size_t>(fk)] = 1;
    ctx->image_lane[static_cast<size_t>(fk)] = static_cast<uint8_t>(lane);
    ctx->image_seq[static_cast<size_t>(fk)] = ++ctx->upload_seq;  // queue position: a lane is in order
    ctx->image_set[static_cast<size_t>(fk)] = 1;
  }
  if (wait) {  // the host buffer may be reused by the caller
    PCP_HIP_TRY(ctx, hipStreamSynchronize(us));
    for (size_t f = 0; f < nf; ++f)
      if (ctx->image_lane[f] == lane) ctx->image_pending[f] = 0;
  }
  return PCP_OK;
}

int pcp_upload_image(pcp_context *ctx, int32_t frame, const uint8_t *bgr, int64_t row_stride_bytes) {
  return upload_image_impl(ctx, "pcp_upload_image", frame, bgr, row_stride_bytes, true);
}

int pcp_upload_image_async(pcp_context *ctx, int32_t frame, const uint8_t *bgr, int64_t row_stride_bytes) {
  return upload_image_impl(ctx, "pcp_upload_image_async", frame, bgr, row_stride_bytes, false);
}

// `count` keyframes that sit one after the other in pinned host memory (frame_stride_bytes apart): the copy engine moves
// them in blocks of up to kBlockUploadBytes into a staging buffer of the lane (one hipMemcpyAsync per block: 57.6 GB/s
// measured, against 52 GB/s for the pack kernels reading pinned memory in place and 43-46 GB/s for one copy per
// keyframe), the pack kernels of the block's keyframes follow on the same stream, the next block goes to the other lane.
// SURVEY 8(d)(i)'s boundary at C3: 29.8 ms instead of 32.5 (PCIe floor 27.7 ms).  Asynchronous like
// pcp_upload_image_async: the host memory must stay valid and unchanged until a synchronising call returns.
constexpr size_t kBlockUploadBytes = size_t(128) << 20;

int pcp_upload_images_block(pcp_context *ctx, int32_t first_frame, int32_t count, const uint8_t *bgr, int64_t row_stride_bytes,
                            int64_t frame_stride_bytes) {
  int rc = check_ready(ctx, "pcp_upload_images_block", true);
  if (rc != PCP_OK) return rc;
  if (count < 0 || first_frame < 0 || first_frame + count > ctx->n_frames)
    return set_error(ctx, PCP_ERR_RANGE, "pcp_upload_images_block: keyframes [%d, %d) outside 0..%d", first_frame, first_frame + count,
                     ctx->n_frames);
  if (count == 0) return PCP_OK;
  const int32_t w = ctx->dcam.img_w, h = ctx->dcam.img_h;
  if (!bgr || row_stride_bytes < 3 * static_cast<int64_t>(w) || frame_stride_bytes < row_stride_bytes * h)
    return set_error(ctx, PCP_ERR_INVALID, "pcp_upload_images_block: NULL block, row stride < 3*width or frame stride < rows");
  hipPointerAttribute_t attr{};
  const bool host_pinned = hipPointerGetAttributes(&attr, bgr) == hipSuccess && attr.type == hipMemoryTypeHost;
  if (!host_pinned) {
    // device memory is read in place and pageable memory is staged per keyframe: the single-keyframe path serves both
    (void)hipGetLastError();
    for (int32_t k = 0; k < count; ++k)
      if ((rc = upload_image_impl(ctx, "pcp_upload_images_block", first_frame + k, bgr + static_cast<int64_t>(k) * frame_stride_bytes,
                                  row_stride_bytes, false)) != PCP_OK)
        return rc;
    return PCP_OK;
  }
  const int32_t per_block = static_cast<int32_t>(std::max<int64_t>(1, static_cast<int64_t>(kBlockUploadBytes) / frame_stride_bytes));
  for (int32_t b0 = 0; b0 < count; b0 += per_block) {
    const int32_t b1 = std::min(count, b0 + per_block);
    // the first keyframe goes through the single-keyframe path's bookkeeping (streams, events, lane choice, ordering against
    // the compute stream) with the block's staging buffer as its source; the others of the block follow on its lane
    const size_t bytes = static_cast<size_t>(b1 - b0 - 1) * static_cast<size_t>(frame_stride_bytes) + static_cast<size_t>(row_stride_bytes) * h;
    if ((rc = upload_image_impl(ctx, "pcp_upload_images_block", first_frame + b0, bgr + static_cast<int64_t>(b0) * frame_stride_bytes,
                                row_stride_bytes, false, bytes, frame_stride_bytes, b1 - b0)) != PCP_OK)
      return rc;
  }
  return PCP_OK;
}

int pcp_set_image_adjust(pcp_context *ctx, int32_t enable, float saturation_scale, float brightness_scale) {
  if (!ctx) return PCP_ERR_INVALID;
  if (!(saturation_scale >= 0.0f) || !(brightness_scale >= 0.0f))
    return set_error(ctx, PCP_ERR_INVALID, "pcp_set_image_adjust: scales must be >= 0");
  PCP_HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (enable && !ctx->hsv_tables.p) {
    int32_t tab[512];
    hsv_build_tables(tab, tab + 256);
    PCP_HIP_TRY(ctx, ctx->hsv_tables.ensure(512));
    PCP_HIP_TRY(ctx, hipMemcpyAsync(ctx->hsv_tables.p, tab, sizeof(tab), hipMemcpyHostToDevice, ctx->stream));
    PCP_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // `tab` lives on this stack frame
  }
  ctx->adjust_images = enable != 0;
  ctx->saturation_scale = saturation_scale;
  ctx->brightness_scale = brightness_scale;
  return PCP_OK;
}

int pcp_download_image(pcp_context *ctx, int32_t frame, uint8_t *out_bgr, uint8_t *out_mask) {
  int rc = check_ready(ctx, "pcp_download_image", true);
  if (rc != PCP_OK) return rc;
  if ((rc = check_frame(ctx, "pcp_download_image", frame)) != PCP_OK) return rc;
  if (!ctx->images.p || !(ctx->image_set[static_cast<size_t>(frame)] || ctx->mask_set[static_cast<size_t>(frame)]))
    return set_error(ctx, PCP_ERR_STATE, "pcp_download_image: nothing uploaded for keyframe %d", frame);
  if ((rc = wait_images(ctx, frame, frame + 1)) != PCP_OK) return rc;
  const int32_t w = ctx->dcam.img_w, h = ctx->dcam.img_h;
  const size_t px = static_cast<size_t>(w) * h;
  std::vector<uint32_t> texels(px);
  PCP_HIP_TRY(ctx, hipMemcpyAsync(texels.data(), ctx->images.p + static_cast<int64_t>(frame) * static_cast<int64_t>(px),
                                  px * 4, hipMemcpyDeviceToHost, ctx->stream));
  PCP_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  for (size_t i = 0; i < px; ++i) {
    const uint32_t t = texels[i];
    if (out_bgr) {
      out_bgr[3 * i + 0] = static_cast<uint8_t>(t & 0xffu);
      out_bgr[3 * i + 1] = static_cast<uint8_t>((t >> 8) & 0xffu);
      out_bgr[3 * i + 2] = static_cast<uint8_t>((t >> 16) & 0xffu);
    }
    if (out_mask) out_mask[i] = static_cast<uint8_t>(t >> 24);
  }
  return PCP_OK;
}

int pcp_upload_mask(pcp_context *ctx, int32_t frame, const uint8_t *gray, int64_t row_stride_bytes) {
  int rc = check_ready(ctx, "pcp_upload_mask", true);
  if (rc != PCP_OK) return rc;
  if ((rc = check_frame(ctx, "pcp_upload_mask", frame)) != PCP_OK) return rc;
  const int32_t w = ctx->dcam.img_w, h = ctx->dcam.img_h;
  if (!gray || row_stride_bytes < static_cast<int64_t>(w))
    return set_error(ctx, PCP_ERR_INVALID, "pcp_upload_mask: NULL mask or row stride < width");
  if ((rc = ensure_images(ctx)) != PCP_OK) return rc;
  if ((rc = wait_images(ctx, frame, frame + 1)) != PCP_OK) return rc;  // the mask byte shares its word with the colour
  const size_t bytes = static_cast<size_t>(row_stride_bytes) * h;
  PCP_HIP_TRY(ctx, ctx->s_keep.ensure(bytes + 16));
  PCP_HIP_TRY(ctx, hipMemcpyAsync(ctx->s_keep.p, gray, bytes, hipMemcpyDefault, ctx->stream));
  const int64_t px = static_cast<int64_t>(w) * h;
  {
    LaunchTimer t(ctx, PCP_K_MISC);
    hipLaunchKernelGGL(k_pack_mask, dim3(blocks_for(px)), dim3(kBlock), 0, ctx->stream, ctx->s_keep.p,
                       row_stride_bytes, w, h, ctx->images.p + static_cast<int64_t>(frame) * px);
    PCP_HIP_TRY(ctx, hipGetLastError());
  }
  PCP_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  ctx->mask_set[static_cast<size_t>(frame)] = 1;
  return PCP_OK;
}

int pcp_project_frame(pcp_context *ctx, int32_t frame, int32_t *out_cell, int32_t *out_pixel, float *out_range,
                      float *out_xyz_cam) {
  int rc = check_ready(ctx, "pcp_project_frame", true);
  if (rc != PCP_OK) return rc;
  if ((rc = check_frame(ctx, "pcp_project_frame", frame)) != PCP_OK) return rc;
  const int64_t n = ctx->n;
  const size_t plane = plane_of(ctx);
  PCP_HIP_TRY(ctx, ctx->s_cell.ensure(plane + 4));
  PCP_HIP_TRY(ctx, ctx->s_range.ensure(plane + 4));
  if (out_pixel) PCP_HIP_TRY(ctx, ctx->s_pixel.ensure(plane + 4));
  if (out_xyz_cam) PCP_HIP_TRY(ctx, ctx->s_cam.ensure(3 * plane + 4));
  if (n == 0) return PCP_OK;
  // The kernel walks the Morton-ordered copy (wave-uniform early-outs) and leaves its
  // results in that order on the device; host outputs are scattered back to input order.
  ProjectOut o{};
  o.cell = ctx->s_cell.p;
  o.range = ctx->s_range.p;
  o.pixel = out_pixel ? ctx->s_pixel.p : nullptr;
  o.xc = out_xyz_cam ? ctx->s_cam.p : nullptr;
  o.yc = out_xyz_cam ? ctx->s_cam.p + plane : nullptr;
  o.zc = out_xyz_cam ? ctx->s_cam.p + 2 * plane : nullptr;
  {
    LaunchTimer t(ctx, PCP_K_PROJECT);
    hipLaunchKernelGGL(is_common_camera(ctx->dcam) ? k_project_frame<true> : k_project_frame<false>, dim3(blocks_for(div_up(n, 4))), dim3(kBlock), 0, ctx->stream, ctx->sxyz.p,
                       ctx->sxyz.p + plane, ctx->sxyz.p + 2 * plane, n, ctx->dcam,
                       ctx->hframes[static_cast<size_t>(frame)], o);
    PCP_HIP_TRY(ctx, hipGetLastError());
  }
  if (!(out_cell || out_range || out_pixel || out_xyz_cam)) return PCP_OK;
  const size_t sn = static_cast<size_t>(n);
  PCP_HIP_TRY(ctx, ctx->s_u32.ensure(plane + 4));
  auto fetch = [&](const void *sorted, void *host) -> int {
    {
      LaunchTimer t(ctx, PCP_K_MISC);
      hipLaunchKernelGGL(k_scatter_u32, dim3(blocks_for(n)), dim3(kBlock), 0, ctx->stream,
                         static_cast<const uint32_t *>(sorted), ctx->perm.p, n, ctx->s_u32.p);
      PCP_HIP_TRY(ctx, hipGetLastError());
    }
    PCP_HIP_TRY(ctx, hipMemcpyAsync(host, ctx->s_u32.p, sn * 4, hipMemcpyDeviceToHost, ctx->stream));
    PCP_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PCP_OK;
  };
  if (out_cell && (rc = fetch(o.cell, out_cell)) != PCP_OK) return rc;
  if (out_range && (rc = fetch(o.range, out_range)) != PCP_OK) return rc;
  if (out_pixel && (rc = fetch(o.pixel, out_pixel)) != PCP_OK) return rc;
  if (out_xyz_cam) {
    if ((rc = fetch(o.xc, out_xyz_cam)) != PCP_OK) return rc;
    if ((rc = fetch(o.yc, out_xyz_cam + sn)) != PCP_OK) return rc;
    if ((rc = fetch(o.zc, out_xyz_cam + 2 * sn)) != PCP_OK) return rc;
  }
  return PCP_OK;
}

int pcp_cull_frame(pcp_context *ctx, int32_t frame, uint8_t *out_keep, int64_t *out_kept, float *out_depth_map) {
  int rc = check_ready(ctx, "pcp_cull_frame", true);
  if (rc != PCP_OK) return rc;
  if ((rc = check_frame(ctx, "pcp_cull_frame", frame)) != PCP_OK) return rc;
  const int64_t n = ctx->n;
  if ((rc = single_frame_depth(ctx, frame)) != PCP_OK) return rc;
  if ((rc = frame_keep_flags(ctx, frame, false)) != PCP_OK) return rc;
  if (out_kept) {
    int64_t cnt = 0;
    if (n > 0 && (rc = compact_flags(ctx, ctx->s_keep.p, n, nullptr, 0, &cnt)) != PCP_OK) return rc;
    *out_kept = cnt;
  }
  if (out_keep && n > 0)  // host memory, or device memory of this GPU (the multi-GPU host exchanges the flags with RCCL)
    PCP_HIP_TRY(ctx, hipMemcpyAsync(out_keep, ctx->s_keep.p, static_cast<size_t>(n), hipMemcpyDefault, ctx->stream));
  if (out_depth_map)
    PCP_HIP_TRY(ctx, hipMemcpyAsync(out_depth_map, ctx->s_u32.p, static_cast<size_t>(cells_of(ctx)) * 4,
                                    hipMemcpyDeviceToHost, ctx->stream));
  PCP_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return PCP_OK;
}

int pcp_frame_visible(pcp_context *ctx, int32_t frame, int64_t capacity, int32_t *out_index, uint8_t *out_rgb,
                      uint16_t *out_mask, float *out_xyz_cam, float *out_xyz_world, int64_t *out_count) {
  int rc = check_ready(ctx, "pcp_frame_visible", true);
  if (rc != PCP_OK) return rc;
  if ((rc = check_frame(ctx, "pcp_frame_visible", frame)) != PCP_OK) return rc;
  if (capacity < 0) return set_error(ctx, PCP_ERR_INVALID, "pcp_frame_visible: negative capacity");
  const int64_t n = ctx->n;
  if (out_count) *out_count = 0;
  if (n == 0) return PCP_OK;
  if ((rc = ensure_images(ctx)) != PCP_OK) return rc;
  if ((rc = wait_images(ctx, frame, frame + 1)) != PCP_OK) return rc;
  if ((rc = single_frame_depth(ctx, frame)) != PCP_OK) return rc;
  const size_t plane = plane_of(ctx);
  PCP_HIP_TRY(ctx, ctx->s_cell.ensure(plane + 4));
  if ((rc = frame_keep_flags(ctx, frame, true)) != PCP_OK) return rc;
  int64_t m = 0;
  if ((rc = compact_flags(ctx, ctx->s_keep.p, n, ctx->s_cell.p, static_cast<int64_t>(plane), &m)) != PCP_OK) return rc;
  if (out_count) *out_count = m;
  const int64_t take = std::min(m, capacity);
  if (take == 0) return PCP_OK;
  const size_t st = static_cast<size_t>(take);
  PCP_HIP_TRY(ctx, ctx->s_range.ensure(st + 4));  // packed rgbm as raw words
  const bool want_cam = out_xyz_cam != nullptr, want_world = out_xyz_world != nullptr;
  PCP_HIP_TRY(ctx, ctx->s_cam.ensure((want_cam ? 3 * st : 0) + (want_world ? 3 * st : 0) + 4));
  float *d_cam = want_cam ? ctx->s_cam.p : nullptr;
  float *d_world = want_world ? ctx->s_cam.p + (want_cam ? 3 * st : 0) : nullptr;
  const int64_t px = static_cast<int64_t>(ctx->dcam.img_w) * ctx->dcam.img_h;
  {
    LaunchTimer t(ctx, PCP_K_MISC);
    hipLaunchKernelGGL(k_gather_visible, dim3(blocks_for(take)), dim3(kBlock), 0, ctx->stream, ctx->xyz.p,
                       ctx->xyz.p + plane, ctx->xyz.p + 2 * plane, ctx->dcam, ctx->hframes[static_cast<size_t>(frame)],
                       ctx->s_cell.p, take, ctx->images.p + static_cast<int64_t>(frame) * px,
                       static_cast<int32_t>(ctx->image_set[static_cast<size_t>(frame)]),
                       static_cast<int32_t>(ctx->mask_set[static_cast<size_t>(frame)]),
                       reinterpret_cast<uint32_t *>(ctx->s_range.p), d_cam, d_world);
    PCP_HIP_TRY(ctx, hipGetLastError());
  }
  std::vector<uint32_t> rgbm;
  if (out_rgb || out_mask) {
    rgbm.resize(st);
    PCP_HIP_TRY(ctx, hipMemcpyAsync(rgbm.data(), ctx->s_range.p, st * 4, hipMemcpyDeviceToHost, ctx->stream));
  }
  if (out_index) PCP_HIP_TRY(ctx, hipMemcpyAsync(out_index, ctx->s_cell.p, st * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (want_cam) PCP_HIP_TRY(ctx, hipMemcpyAsync(out_xyz_cam, d_cam, 3 * st * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (want_world) PCP_HIP_TRY(ctx, hipMemcpyAsync(out_xyz_world, d_world, 3 * st * 4, hipMemcpyDeviceToHost, ctx->stream));
  PCP_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  for (size_t k = 0; k < rgbm.size(); ++k) {
    if (out_rgb) {
      out_rgb[3 * k + 0] = static_cast<uint8_t>(rgbm[k] & 0xffu);
      out_rgb[3 * k + 1] = static_cast<uint8_t>((rgbm[k] >> 8) & 0xffu);
      out_rgb[3 * k + 2] = static_cast<uint8_t>((rgbm[k] >> 16) & 0xffu);
    }
    if (out_mask) out_mask[k] = static_cast<uint16_t>(rgbm[k] >> 24);
  }
  return PCP_OK;
}

int pcp_depth_pass(pcp_context *ctx, int32_t frame_begin, int32_t frame_end) {
  int rc = check_ready(ctx, "pcp_depth_pass", true);
  if (rc != PCP_OK) return rc;
  if (frame_begin < 0 || frame_end > ctx->n_frames || frame_begin > frame_end)
    return set_error(ctx, PCP_ERR_RANGE, "pcp_depth_pass: keyframe range [%d,%d) outside 0..%d", frame_begin, frame_end,
                     ctx->n_frames);
  if (frame_begin == frame_end) return PCP_OK;
  if ((rc = ensure_depth(ctx)) != PCP_OK) return rc;
  const int64_t cells = cells_of(ctx);
  // the pass takes the minimum of the SQUARED ranges (k_depth_finish turns them into the reference's fp32 range maps)
  const int64_t map_cells = static_cast<int64_t>(frame_end - frame_begin) * cells;
  const bool z_maps = ctx->n > 0 && ctx->dcam.enable_zbuf;
  if (z_maps) {
    PCP_HIP_TRY(ctx, ctx->depth_sq.ensure(static_cast<size_t>(map_cells) + 4));  // (set to "far" by the first kernel of the stage)
  } else if ((rc = fill_u32(ctx, ctx->depth.p + static_cast<int64_t>(frame_begin) * cells, map_cells, kFltMaxBits)) != PCP_OK) {
    return rc;
  }
  if (ctx->n > 0) {
    const size_t plane = plane_of(ctx);
    // tile x keyframe masks for the words this range touches
    const int32_t w0 = frame_begin >> 5, w1 = ((frame_end - 1) >> 5) + 1;
    static const bool cull_tiles = [] {
      const char *e = std::getenv("PCP_DISABLE_TILE_CULL");
      return !(e && e[0] == '1');
    }();
    {
      LaunchTimer t(ctx, PCP_K_TILE_MASK);
      const int64_t groups = div_up(ctx->n_tiles, kTileGroup);
      const float4 *tile_sph = reinterpret_cast<const float4 *>(ctx->tile_sphere.p);
      const size_t gwords = static_cast<size_t>(groups) * ctx->mask_words;
      PCP_HIP_TRY(ctx, ctx->group_mask.ensure(2 * gwords + 8));  // keep bits | "wholly inside" bits
      uint32_t *group_inside = ctx->group_mask.p + gwords + 4;
      hipLaunchKernelGGL(k_group_mask_flat, dim3(std::max(blocks_for(groups * (w1 - w0) * 32), blocks_for(2 * kWorkBins))),
                         dim3(kBlock), 0, ctx->stream, tile_sph + ctx->n_tiles, groups, ctx->dcam, ctx->frames.p, ctx->n_frames,
                         w0, w1, ctx->mask_words, ctx->group_mask.p, group_inside, cull_tiles ? 1 : 0, ctx->work_hist.p,
                         2 * kWorkBins, z_maps ? reinterpret_cast<unsigned long long *>(ctx->depth_sq.p) : nullptr,
                         z_maps ? map_cells : int64_t(0));
      hipLaunchKernelGGL(k_tile_mask_dense, dim3(static_cast<uint32_t>(div_up(groups * (w1 - w0), kBlock / 64))), dim3(kBlock),
                         0, ctx->stream, tile_sph, ctx->n_tiles, ctx->dcam, ctx->frames.p, ctx->n_frames, w0, w1,
                         ctx->mask_words, ctx->group_mask.p, group_inside, ctx->tile_mask.p, ctx->tile_inside.p, cull_tiles ? 1 : 0);
      // longest-work-first order of the tiles for this pass
      if ((rc = sort_tiles_by_work(ctx, w0, w1, ctx->tile_order.p)) != PCP_OK) return rc;
      ctx->tile_order_live = true;
    }
    if (ctx->cull.cull_mode == PCP_CULL_HPR) {
      // hidden_points_removal has no depth map to pass on: its verdict per (point, keyframe) is a bit (pcp_hpr.hip), taken
      // here keyframe by keyframe; the colour pass reads the bits where the z-buffer routine reads the maps.  The hulls come
      // BEFORE the depth kernel: their candidate kernels skip the tiles the masks cleared, and the depth kernel is about to
      // refine the masks by a rule of its own (a candidate lane must also have a colour pixel) which the hull does not share.
      int rch = ensure_hull_bits(ctx);
      if (rch != PCP_OK) return rch;
      // an index shard (PCP_DEPTH_BATCHED) cannot take a hull: its bits come through pcp_hull_flags_import
      if (!ctx->depth_from_batch) {
        // first every bit this range is about to write is cleared (whole planes by one memset where all 32 keyframes of the
        // plane are coming, single bits otherwise), then the hulls -- several keyframes in flight on lanes of their own
        // (pcp_hpr.hip hpr_run_range; PCP_HPR_LANES, default 4: the keyframes are independent and one keyframe's kernels
        // leave most of the chip idle), each setting its bit at the sorted place of every hull vertex
        int32_t clear_until = frame_begin;  // keyframes below this one have their plane zeroed already
        for (int32_t f = frame_begin; f < frame_end; ++f) {
          uint32_t *hull_plane = ctx->hull_bits.p + static_cast<size_t>(f >> 5) * static_cast<size_t>(ctx->n);
          if ((f & 31) == 0 && std::min(f + 32, ctx->n_frames) <= frame_end) {  // all keyframes of this plane are coming
            PCP_HIP_TRY(ctx, hipMemsetAsync(hull_plane, 0, static_cast<size_t>(ctx->n) * 4, ctx->stream));
            clear_until = std::min(f + 32, ctx->n_frames);
          }
          if (f >= clear_until) {
            LaunchTimer t(ctx, PCP_K_HPR);
            hipLaunchKernelGGL(k_hull_clear, dim3(blocks_for(ctx->n)), dim3(kBlock), 0, ctx->stream, hull_plane, ctx->n, 1u << (f & 31));
          }
        }
        PCP_HIP_TRY(ctx, hipGetLastError());
        const char *le = std::getenv("PCP_HPR_LANES");
        const int32_t lanes = le && le[0] >= '1' && le[0] <= '8' ? le[0] - '0' : 4;
        if ((rc = hpr_run_range(ctx, frame_begin, frame_end, lanes, ctx->tile_mask.p)) != PCP_OK) return rc;
        // (the single-keyframe calls read these bits instead of taking the keyframe's hull again)
        for (int32_t f = frame_begin; f < frame_end; ++f) ctx->hull_valid[static_cast<size_t>(f)] = 1;
      }
    }
    {
      LaunchTimer t(ctx, PCP_K_DEPTH);
      hipLaunchKernelGGL(is_common_camera(ctx->dcam) ? k_depth_pass<true> : k_depth_pass<false>,
                         dim3(static_cast<uint32_t>(ctx->n_tiles)), dim3(64), 0, ctx->stream, ctx->sxyz.p,
                         ctx->sxyz.p + plane, ctx->sxyz.p + 2 * plane, ctx->n, ctx->dcam, ctx->frames.p, frame_begin,
                         frame_end, ctx->depth_sq.p, cells, frame_begin, ctx->tile_mask.p, ctx->tile_inside.p,
                         ctx->mask_words, ctx->tile_order.p);
      if (z_maps)
        hipLaunchKernelGGL(k_depth_finish, dim3(static_cast<uint32_t>(std::min<int64_t>(div_up(map_cells, kBlock), 4096))),
                           dim3(kBlock), 0, ctx->stream, ctx->depth_sq.p, map_cells,
                           ctx->depth.p + static_cast<int64_t>(frame_begin) * cells);
      PCP_HIP_TRY(ctx, hipGetLastError());
    }
  }
  for (int32_t f = frame_begin; f < frame_end; ++f) ctx->depth_valid[static_cast<size_t>(f)] = 1;
  return PCP_OK;
}

int pcp_depth_maps_device(pcp_context *ctx, void **device_ptr, int64_t *n_floats) {
  int rc = check_ready(ctx, "pcp_depth_maps_device", true);
  if (rc != PCP_OK) return rc;
  if (ctx->cull.cull_mode == PCP_CULL_HPR)
    return set_error(ctx, PCP_ERR_STATE, "pcp_depth_maps_device: PCP_CULL_HPR has no depth maps to merge across point shards "
                     "(the hull needs the whole map on one GPU)");
  if ((rc = ensure_depth(ctx)) != PCP_OK) return rc;
  if (device_ptr) *device_ptr = ctx->depth.p;
  if (n_floats) *n_floats = cells_of(ctx) * ctx->n_frames;
  return PCP_OK;
}

int pcp_set_depth_source(pcp_context *ctx, int32_t source) {
  if (!ctx) return PCP_ERR_INVALID;
  if (source != PCP_DEPTH_OWN && source != PCP_DEPTH_BATCHED)
    return set_error(ctx, PCP_ERR_INVALID, "pcp_set_depth_source: unknown source %d", source);
  ctx->depth_from_batch = source == PCP_DEPTH_BATCHED;
  return PCP_OK;
}

int pcp_hull_flags_import(pcp_context *ctx, int32_t frame, const uint8_t *keep) {
  int rc = check_ready(ctx, "pcp_hull_flags_import", true);
  if (rc != PCP_OK) return rc;
  if ((rc = check_frame(ctx, "pcp_hull_flags_import", frame)) != PCP_OK) return rc;
  if (ctx->cull.cull_mode != PCP_CULL_HPR || !ctx->depth_from_batch)
    return set_error(ctx, PCP_ERR_STATE, "pcp_hull_flags_import: needs cull_mode PCP_CULL_HPR and pcp_set_depth_source(PCP_DEPTH_BATCHED)");
  if (!keep && ctx->n > 0) return set_error(ctx, PCP_ERR_INVALID, "pcp_hull_flags_import: NULL flags");
  if ((rc = ensure_hull_bits(ctx)) != PCP_OK) return rc;
  if (ctx->n > 0) {
    PCP_HIP_TRY(ctx, ctx->s_keep.ensure(static_cast<size_t>(ctx->n) + 16));
    PCP_HIP_TRY(ctx, hipMemcpyAsync(ctx->s_keep.p, keep, static_cast<size_t>(ctx->n), hipMemcpyDefault, ctx->stream));
    if ((rc = store_hull_bits(ctx, frame, false)) != PCP_OK) return rc;
    PCP_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // the caller's buffer is free again
  }
  ctx->hull_valid[static_cast<size_t>(frame)] = 1;
  return PCP_OK;
}

int pcp_download_depth_map(pcp_context *ctx, int32_t frame, float *out_depth_map) {
  int rc = check_ready(ctx, "pcp_download_depth_map", true);
  if (rc != PCP_OK) return rc;
  if ((rc = check_frame(ctx, "pcp_download_depth_map", frame)) != PCP_OK) return rc;
  if (!out_depth_map) return set_error(ctx, PCP_ERR_INVALID, "pcp_download_depth_map: NULL output");
  if (!ctx->depth.p || !ctx->depth_valid[static_cast<size_t>(frame)])
    return set_error(ctx, PCP_ERR_STATE, "pcp_download_depth_map: pcp_depth_pass has not covered keyframe %d", frame);
  const int64_t cells = cells_of(ctx);
  PCP_HIP_TRY(ctx, hipMemcpyAsync(out_depth_map, ctx->depth.p + static_cast<int64_t>(frame) * cells,
                                  static_cast<size_t>(cells) * 4, hipMemcpyDeviceToHost, ctx->stream));
  PCP_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return PCP_OK;
}

int pcp_colour_reset(pcp_context *ctx) {
  if (!ctx) return PCP_ERR_INVALID;
  ctx->colour_state_live = false;
  ctx->colour_result_live = false;
  return PCP_OK;
}

// next result buffer (input order); if an asynchronous download still reads it, the kernels wait
// for that copy only
static int begin_result(pcp_context *ctx, uint32_t **dst) {
  const int32_t cur = ctx->rgba_cur ^ 1;
  PCP_HIP_TRY(ctx, ctx->rgba2[cur].ensure(static_cast<size_t>(ctx->n) + 4));
  if (ctx->copy_pending[cur]) {
    PCP_HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->copy_done[cur], 0));
    ctx->copy_pending[cur] = false;
  }
  *dst = ctx->rgba2[cur].p;
  return PCP_OK;
}

static int colour_pass_impl(pcp_context *ctx, int32_t frame_begin, int32_t frame_end, bool one_shot,
                            uint32_t *result = nullptr) {
  int rc = check_ready(ctx, "pcp_colour_pass", true);
  if (rc != PCP_OK) return rc;
  if (frame_begin < 0 || frame_end > ctx->n_frames || frame_begin > frame_end)
    return set_error(ctx, PCP_ERR_RANGE, "pcp_colour_pass: keyframe range [%d,%d) outside 0..%d", frame_begin, frame_end,
                     ctx->n_frames);
  for (int32_t f = frame_begin; f < frame_end; ++f) {
    if (!ctx->depth.p || !ctx->depth_valid[static_cast<size_t>(f)])
      return set_error(ctx, PCP_ERR_STATE, "pcp_colour_pass: pcp_depth_pass has not covered keyframe %d", f);
    if (!ctx->images.p || !ctx->image_set[static_cast<size_t>(f)])
      return set_error(ctx, PCP_ERR_STATE, "pcp_colour_pass: no image uploaded for keyframe %d", f);
    if (ctx->cull.cull_mode == PCP_CULL_HPR && ctx->depth_from_batch &&
        (static_cast<size_t>(f) >= ctx->hull_valid.size() || !ctx->hull_valid[static_cast<size_t>(f)]))
      return set_error(ctx, PCP_ERR_STATE, "pcp_colour_pass: PCP_CULL_HPR on an index shard, pcp_hull_flags_import has not covered keyframe %d", f);
  }
  if ((rc = ensure_state(ctx)) != PCP_OK) return rc;
  if ((rc = wait_images(ctx, frame_begin, frame_end)) != PCP_OK) return rc;
  if (ctx->n == 0) return PCP_OK;
  if (frame_begin == frame_end) {
    if (one_shot && result) PCP_HIP_TRY(ctx, hipMemsetAsync(result, 0, static_cast<size_t>(ctx->n) * 4, ctx->stream));
    return PCP_OK;
  }
  const size_t plane = plane_of(ctx);
  TopState st{ctx->top_score.p, ctx->top_rgb.p, ctx->top_frame.p, ctx->view_count.p};
  int32_t flags = 0;
  if (ctx->colour_state_live) flags |= 1;
  if (one_shot)
    flags |= 4;
  else
    flags |= 2;
  {
    LaunchTimer t(ctx, PCP_K_COLOUR);
    // the colour pass keeps the cloud order (256-thread workgroups, XCD-chunked): it does not end in a tail of heavy
    // tiles (longest-first order: no gain at 1920x1080), and its texel gathers want neighbouring tiles on the same L2
    // (longest-first order at 4096x3000: 1.96 -> 2.18 ms)
    const bool ordered = false;
    const bool common = is_common_camera(ctx->dcam) && !ordered && ctx->cull.cull_mode != PCP_CULL_HPR;
    // (mode 1 writes the scratch buffer a download of mode 2 may still be filling on the copy stream: tests flip the modes)
    if (one_shot && unpermute_mode() == 1) PCP_HIP_TRY(ctx, hipStreamSynchronize(ctx->copy_stream));
    const int32_t launch_flags = flags | ((one_shot && unpermute_mode() != 0) ? 8 : 0);
    ctx->last_pass_sorted = one_shot && unpermute_mode() == 2;  // the result stays in sorted order: end_result marks the buffer
    auto kernel = k_colour_pass<false, 0, false>;
    if (common && ctx->dcam.match_mode == PCP_MATCH_IDENTITY)
      kernel = (launch_flags & ~8) == 4 ? k_colour_pass<true, 1, true> : k_colour_pass<true, 1, false>;
    if (common && ctx->dcam.match_mode == PCP_MATCH_ROUNDTRIP)
      kernel = (launch_flags & ~8) == 4 ? k_colour_pass<true, 2, true> : k_colour_pass<true, 2, false>;
    hipLaunchKernelGGL(kernel, dim3(ordered ? static_cast<uint32_t>(ctx->n_tiles) : blocks_for(ctx->n)),
                       dim3(ordered ? 64 : kBlock), 0, ctx->stream, ctx->sxyz.p, ctx->sxyz.p + plane,
                       ctx->sxyz.p + 2 * plane, ctx->n, ctx->dcam, ctx->frames.p, frame_begin, frame_end, ctx->depth.p,
                       cells_of(ctx), ctx->tile_mask.p, ctx->mask_words, ctx->images.p,
                       static_cast<int64_t>(ctx->dcam.img_w) * ctx->dcam.img_h, st, ctx->perm.p,
                       (one_shot && unpermute_results()) ? ctx->rgba_sorted.p : result,
                       launch_flags, ordered ? ctx->tile_order.p : static_cast<const int32_t *>(nullptr),
                       ctx->cull.cull_mode == PCP_CULL_HPR ? ctx->hull_bits.p : static_cast<const uint32_t *>(nullptr));
    if (one_shot && unpermute_results())
      hipLaunchKernelGGL(k_unpermute, dim3(blocks_for(div_up(ctx->n, 4))), dim3(kBlock), 0, ctx->stream, ctx->rgba_sorted.p,
                         ctx->inv_perm.p, ctx->n, result);
    PCP_HIP_TRY(ctx, hipGetLastError());
  }
  if (!one_shot) ctx->colour_state_live = true;
  return PCP_OK;
}

int pcp_colour_pass(pcp_context *ctx, int32_t frame_begin, int32_t frame_end) {
  return colour_pass_impl(ctx, frame_begin, frame_end, false);
}

static int end_result(pcp_context *ctx, uint8_t *out_rgb, uint8_t *out_has) {
  const int64_t n = ctx->n;
  const int32_t cur = ctx->rgba_cur ^ 1;
  uint32_t *dst = ctx->rgba2[cur].p;
  PCP_HIP_TRY(ctx, hipEventRecord(ctx->result_ready[cur], ctx->stream));
  ctx->rgba_cur = cur;
  ctx->colour_result_live = true;
  ctx->result_sorted[cur] = ctx->last_pass_sorted;  // the producer left the words in sorted order: readers un-permute
  ctx->last_pass_sorted = false;
  if ((out_rgb || out_has) && n > 0) {
    // split the packed words on the device: 3 + 1 bytes per point cross PCIe, and no host loop over the points
    const size_t sn = static_cast<size_t>(n);
    PCP_HIP_TRY(ctx, ctx->s_keep.ensure(4 * sn + 16));
    uint8_t *d_rgb = ctx->s_keep.p, *d_has = ctx->s_keep.p + 3 * sn;
    hipLaunchKernelGGL(k_unpack_result, dim3(blocks_for(n)), dim3(kBlock), 0, ctx->stream, dst, n, d_rgb, d_has,
                       ctx->result_sorted[cur] ? ctx->inv_perm.p : static_cast<const int32_t *>(nullptr));
    PCP_HIP_TRY(ctx, hipGetLastError());
    if (out_rgb) PCP_HIP_TRY(ctx, hipMemcpyAsync(out_rgb, d_rgb, 3 * sn, hipMemcpyDeviceToHost, ctx->stream));
    if (out_has) PCP_HIP_TRY(ctx, hipMemcpyAsync(out_has, d_has, sn, hipMemcpyDeviceToHost, ctx->stream));
    PCP_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  return PCP_OK;
}

int pcp_colour_finalise(pcp_context *ctx, uint8_t *out_rgb, uint8_t *out_has, int32_t *out_count, float *out_top_score,
                        uint32_t *out_top_rgb, int32_t *out_top_frame) {
  int rc = check_ready(ctx, "pcp_colour_finalise", false);
  if (rc != PCP_OK) return rc;
  const int64_t n = ctx->n;
  if ((rc = ensure_state(ctx)) != PCP_OK) return rc;
  const size_t sn = static_cast<size_t>(n);
  uint32_t *result = nullptr;
  if ((rc = begin_result(ctx, &result)) != PCP_OK) return rc;
  if (!ctx->colour_state_live && n > 0) {
    // no keyframe processed: empty lists
    PCP_HIP_TRY(ctx, hipMemsetAsync(result, 0, sn * 4, ctx->stream));
    if ((rc = fill_u32(ctx, reinterpret_cast<uint32_t *>(ctx->top_score.p), kTopM * n, 0xbf800000u)) != PCP_OK) return rc;
    PCP_HIP_TRY(ctx, hipMemsetAsync(ctx->top_rgb.p, 0, kTopM * sn * 4, ctx->stream));
    PCP_HIP_TRY(ctx, hipMemsetAsync(ctx->top_frame.p, 0xff, kTopM * sn * 4, ctx->stream));
    PCP_HIP_TRY(ctx, hipMemsetAsync(ctx->view_count.p, 0, sn * 4, ctx->stream));
  } else if (n > 0) {
    TopState st{ctx->top_score.p, ctx->top_rgb.p, ctx->top_frame.p, ctx->view_count.p};
    LaunchTimer t(ctx, PCP_K_COLOUR);
    const bool un = unpermute_results();
    ctx->last_pass_sorted = unpermute_mode() == 2;
    hipLaunchKernelGGL(k_finalise, dim3(blocks_for(n)), dim3(kBlock), 0, ctx->stream, n, st, ctx->perm.p,
                       un ? ctx->rgba_sorted.p : result, (un || ctx->last_pass_sorted) ? 1 : 0);
    if (un)
      hipLaunchKernelGGL(k_unpermute, dim3(blocks_for(div_up(n, 4))), dim3(kBlock), 0, ctx->stream, ctx->rgba_sorted.p,
                         ctx->inv_perm.p, n, result);
    PCP_HIP_TRY(ctx, hipGetLastError());
  }
  if ((rc = end_result(ctx, out_rgb, out_has)) != PCP_OK) return rc;
  if ((out_count || out_top_score || out_top_rgb || out_top_frame) && n > 0) {
    std::vector<int32_t> perm(sn), cnt;
    std::vector<uint32_t> tmp(kTopM * sn);
    PCP_HIP_TRY(ctx, hipMemcpyAsync(perm.data(), ctx->perm.p, sn * 4, hipMemcpyDeviceToHost, ctx->stream));
    auto fetch5 = [&](const void *src, void *dst_v) -> int {
      PCP_HIP_TRY(ctx, hipMemcpyAsync(tmp.data(), src, kTopM * sn * 4, hipMemcpyDeviceToHost, ctx->stream));
      PCP_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
      uint32_t *dst = static_cast<uint32_t *>(dst_v);
      for (size_t j = 0; j < sn; ++j)
        for (int k = 0; k < kTopM; ++k) dst[static_cast<size_t>(perm[j]) * kTopM + k] = tmp[static_cast<size_t>(k) * sn + j];
      return PCP_OK;
    };
    if (out_top_score && (rc = fetch5(ctx->top_score.p, out_top_score)) != PCP_OK) return rc;
    if (out_top_rgb && (rc = fetch5(ctx->top_rgb.p, out_top_rgb)) != PCP_OK) return rc;
    if (out_top_frame && (rc = fetch5(ctx->top_frame.p, out_top_frame)) != PCP_OK) return rc;
    if (out_count) {
      cnt.resize(sn);
      PCP_HIP_TRY(ctx, hipMemcpyAsync(cnt.data(), ctx->view_count.p, sn * 4, hipMemcpyDeviceToHost, ctx->stream));
      PCP_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
      for (size_t j = 0; j < sn; ++j) out_count[perm[j]] = cnt[j];
    }
  }
  return PCP_OK;
}

int pcp_colorize(pcp_context *ctx, uint8_t *out_rgb, uint8_t *out_has) {
  int rc = check_ready(ctx, "pcp_colorize", true);
  if (rc != PCP_OK) return rc;
  pcp_colour_reset(ctx);
  if ((rc = pcp_depth_pass(ctx, 0, ctx->n_frames)) != PCP_OK) return rc;
  uint32_t *result = nullptr;
  if ((rc = begin_result(ctx, &result)) != PCP_OK) return rc;
  if ((rc = colour_pass_impl(ctx, 0, ctx->n_frames, true, result)) != PCP_OK) return rc;
  return end_result(ctx, out_rgb, out_has);
}

int pcp_colorize_from_depth(pcp_context *ctx, uint8_t *out_rgb, uint8_t *out_has) {
  int rc = check_ready(ctx, "pcp_colorize_from_depth", true);
  if (rc != PCP_OK) return rc;
  pcp_colour_reset(ctx);
  uint32_t *result = nullptr;
  if ((rc = begin_result(ctx, &result)) != PCP_OK) return rc;
  if ((rc = colour_pass_impl(ctx, 0, ctx->n_frames, true, result)) != PCP_OK) return rc;
  return end_result(ctx, out_rgb, out_has);
}

// The current result as a device array in INPUT order (callers that read rgba2[cur] themselves): where the producer left it in
// sorted order, one un-permuting kernel into the scratch buffer, which then becomes the result buffer.
static int result_in_input_order(pcp_context *ctx) {
  const int32_t cur = ctx->rgba_cur;
  if (!ctx->result_sorted[cur]) return PCP_OK;
  if (ctx->copy_pending[cur]) {  // a download of this buffer may still be reading it on the copy stream
    PCP_HIP_TRY(ctx, hipEventSynchronize(ctx->copy_done[cur]));
    ctx->copy_pending[cur] = false;
  }
  PCP_HIP_TRY(ctx, hipStreamSynchronize(ctx->copy_stream));  // (... or still writing the scratch)
  PCP_HIP_TRY(ctx, ctx->rgba_sorted.ensure(static_cast<size_t>(ctx->n) + 4));
  hipLaunchKernelGGL(k_unpermute, dim3(blocks_for(div_up(ctx->n, 4))), dim3(kBlock), 0, ctx->stream, ctx->rgba2[cur].p,
                     ctx->inv_perm.p, ctx->n, ctx->rgba_sorted.p);
  PCP_HIP_TRY(ctx, hipGetLastError());
  std::swap(ctx->rgba2[cur], ctx->rgba_sorted);
  ctx->result_sorted[cur] = false;
  PCP_HIP_TRY(ctx, hipEventRecord(ctx->result_ready[cur], ctx->stream));
  return PCP_OK;
}

int pcp_download_result_packed(pcp_context *ctx, uint32_t *out_rgba) {
  if (!ctx) return PCP_ERR_INVALID;
  if (!ctx->colour_result_live)
    return set_error(ctx, PCP_ERR_STATE, "pcp_download_result_packed: no result (call pcp_colorize / pcp_colour_finalise)");
  if (!out_rgba && ctx->n > 0) return set_error(ctx, PCP_ERR_INVALID, "pcp_download_result_packed: NULL output");
  if (ctx->n > 0) {
    int rc = result_in_input_order(ctx);
    if (rc != PCP_OK) return rc;
    PCP_HIP_TRY(ctx, hipMemcpyAsync(out_rgba, ctx->rgba2[ctx->rgba_cur].p, static_cast<size_t>(ctx->n) * 4,
                                    hipMemcpyDeviceToHost, ctx->stream));
  }
  PCP_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return PCP_OK;
}

int pcp_download_result_packed_async(pcp_context *ctx, uint32_t *out_rgba) {
  if (!ctx) return PCP_ERR_INVALID;
  if (!ctx->colour_result_live)
    return set_error(ctx, PCP_ERR_STATE, "pcp_download_result_packed_async: no result (call pcp_colorize / pcp_colour_finalise)");
  if (!out_rgba && ctx->n > 0) return set_error(ctx, PCP_ERR_INVALID, "pcp_download_result_packed_async: NULL output");
  const int32_t cur = ctx->rgba_cur;
  PCP_HIP_TRY(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->result_ready[cur], 0));
  if (ctx->n > 0 && ctx->result_sorted[cur]) {
    // the result sits in the sorted order the colour pass walks: the un-permutation rides on the copy stream -- a kernel gathers
    // sorted[inv_perm[i]] into a device scratch in input order (~35 us at 10 M points, beside the next step's passes), the copy
    // engine takes it from there.  (Measured and left out: the kernel writing the caller's pinned buffer itself over PCIe, in
    // place of the blit -- 19.5 GB/s against the copy engine's 55: a step of 2.06 ms instead of 1.30.)
    PCP_HIP_TRY(ctx, ctx->rgba_sorted.ensure(static_cast<size_t>(ctx->n) + 4));
    uint32_t *target = ctx->rgba_sorted.p;
    hipLaunchKernelGGL(k_unpermute, dim3(blocks_for(div_up(ctx->n, 4))), dim3(kBlock), 0, ctx->copy_stream, ctx->rgba2[cur].p,
                       ctx->inv_perm.p, ctx->n, target);
    PCP_HIP_TRY(ctx, hipGetLastError());
    PCP_HIP_TRY(ctx, hipMemcpyAsync(out_rgba, target, static_cast<size_t>(ctx->n) * 4, hipMemcpyDeviceToHost, ctx->copy_stream));
  } else if (ctx->n > 0) {
    PCP_HIP_TRY(ctx, hipMemcpyAsync(out_rgba, ctx->rgba2[cur].p, static_cast<size_t>(ctx->n) * 4, hipMemcpyDeviceToHost,
                                    ctx->copy_stream));
  }
  PCP_HIP_TRY(ctx, hipEventRecord(ctx->copy_done[cur], ctx->copy_stream));
  ctx->copy_pending[cur] = true;
  return PCP_OK;
}

int pcp_download_wait_previous(pcp_context *ctx) {
  if (!ctx) return PCP_ERR_INVALID;
  const int32_t prev = ctx->rgba_cur ^ 1;
  if (ctx->copy_pending[prev]) PCP_HIP_TRY(ctx, hipEventSynchronize(ctx->copy_done[prev]));
  return PCP_OK;
}

int pcp_tile_mask_density(pcp_context *ctx, double *kept_fraction) {
  if (!ctx || !kept_fraction) return PCP_ERR_INVALID;
  if (!ctx->tile_mask.p || ctx->n_tiles == 0 || ctx->n_frames == 0)
    return set_error(ctx, PCP_ERR_STATE, "pcp_tile_mask_density: no masks (call pcp_depth_pass)");
  std::vector<uint32_t> h(static_cast<size_t>(ctx->n_tiles) * ctx->mask_words);
  PCP_HIP_TRY(ctx, hipMemcpyAsync(h.data(), ctx->tile_mask.p, h.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
  PCP_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  uint64_t bits = 0;
  for (uint32_t w : h) bits += static_cast<uint64_t>(__builtin_popcount(w));
  *kept_fraction = static_cast<double>(bits) / (static_cast<double>(ctx->n_tiles) * ctx->n_frames);
  return PCP_OK;
}

int pcp_tile_masks(pcp_context *ctx, int64_t *tiles, int32_t *mask_words, uint32_t *out_words) {
  if (!ctx) return PCP_ERR_INVALID;
  if (!ctx->tile_mask.p || ctx->n_tiles == 0 || ctx->n_frames == 0)
    return set_error(ctx, PCP_ERR_STATE, "pcp_tile_masks: no masks (call pcp_depth_pass)");
  if (tiles) *tiles = ctx->n_tiles;
  if (mask_words) *mask_words = ctx->mask_words;
  if (out_words) {
    PCP_HIP_TRY(ctx, hipMemcpyAsync(out_words, ctx->tile_mask.p, static_cast<size_t>(ctx->n_tiles) * ctx->mask_words * 4,
                                    hipMemcpyDeviceToHost, ctx->stream));
    PCP_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  return PCP_OK;
}

int pcp_colour_result_device(pcp_context *ctx, void **device_ptr, int64_t *n_words) {
  if (!ctx) return PCP_ERR_INVALID;
  if (!ctx->colour_result_live)
    return set_error(ctx, PCP_ERR_STATE, "pcp_colour_result_device: no result (call pcp_colorize / pcp_colour_finalise)");
  if (ctx->n > 0) {
    const int rc = result_in_input_order(ctx);
    if (rc != PCP_OK) return rc;
  }
  if (device_ptr) *device_ptr = ctx->rgba2[ctx->rgba_cur].p;
  if (n_words) *n_words = ctx->n;
  return PCP_OK;
}

int pcp_selftest_arithmetic(pcp_context *ctx, int64_t samples, uint64_t seed, int64_t *mismatches_fp64,
                            int64_t *mismatches_fp32) {
  if (!ctx) return PCP_ERR_INVALID;
  if (!ctx->have_camera) return set_error(ctx, PCP_ERR_STATE, "pcp_selftest_arithmetic: pcp_set_camera has not been called");
  PCP_HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (samples < 0) return set_error(ctx, PCP_ERR_INVALID, "pcp_selftest_arithmetic: negative sample count");
  PCP_HIP_TRY(ctx, ctx->s_counter.ensure(4));
  unsigned long long *bad = ctx->s_counter.p;
  PCP_HIP_TRY(ctx, hipMemsetAsync(bad, 0, 16, ctx->stream));
  if (samples > 0)
    hipLaunchKernelGGL(k_selftest_div64, dim3(blocks_for(samples)), dim3(kBlock), 0, ctx->stream, samples, seed, bad);
  hipLaunchKernelGGL(k_selftest_div32, dim3((1u << 24) / kBlock), dim3(kBlock), 0, ctx->stream, ctx->dcam, bad + 1);
  PCP_HIP_TRY(ctx, hipGetLastError());
  unsigned long long h[2] = {0, 0};
  PCP_HIP_TRY(ctx, hipMemcpyAsync(h, bad, 16, hipMemcpyDeviceToHost, ctx->stream));
  PCP_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (mismatches_fp64) *mismatches_fp64 = static_cast<int64_t>(h[0]);
  if (mismatches_fp32) *mismatches_fp32 = static_cast<int64_t>(h[1]);
  return PCP_OK;
}

}  // extern "C"
