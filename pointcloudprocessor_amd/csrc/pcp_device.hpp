// pcp_device.hpp -- device arithmetic of the colour path, stated operation by
// operation as SURVEY.md Appendix A fixes it.  This translation unit MUST be
// built with -ffp-contract=off (no FMA contraction): every fp32 / fp64 multiply
// and add below is individually rounded, exactly as the reference's x86-64
// baseline build executes them.  Division and sqrt are the correctly rounded
// IEEE forms (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt; fp64
// divide / sqrt expand to correctly rounded sequences on gfx950).
#pragma once

#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>

#include "pcp_internal.hpp"

namespace pcp {

// A2: pcl::transformPointCloud(Affine3f), PCL 1.10 SSE association
// x*c0 + (y*c1 + (z*c2 + c3)); call sites PointCloudProcessor.cpp:196,521,549,555.
__device__ __forceinline__ void xform(const float *__restrict__ m, float x, float y, float z, float &xc, float &yc,
                                      float &zc) {
  xc = x * m[0] + (y * m[1] + (z * m[2] + m[3]));
  yc = x * m[4] + (y * m[5] + (z * m[6] + m[7]));
  zc = x * m[8] + (y * m[9] + (z * m[10] + m[11]));
}

// Correctly rounded X/Z and Y/Z (fp64) for operands that are promoted finite floats, Z > 0.
// This is the compiler's own IEEE f64 division sequence (v_rcp_f64, two Newton steps on the
// reciprocal, quotient, exact residual, one correction: the Markstein scheme LLVM emits for
// fdiv double) with the two divisions sharing the reciprocal and with the range scaling
// (v_div_scale / v_div_fixup) dropped: promoted floats have exponents in [-149, 128], far inside
// the window in which those instructions pass their operands through unchanged.  The only
// difference from the two `/` it replaces is the sign of a zero quotient, which no later
// operation of the projection can observe.  Non-finite operands take the plain divisions.
// pcp_selftest_arithmetic() compares both forms on the device.
// finite_sure (wave-uniform): the caller knows the operands are finite -- a point below 2^40 in every coordinate under
// matrices below 2^40 in every entry (DevCamera::frames_bounded) -- and the test per call is left out.
__device__ __forceinline__ void divide_xy_by_z(float xc, float yc, float zc, double &xn, double &yn, bool finite_sure = false) {
  const double X = xc, Y = yc, Z = zc;
  if (finite_sure || (xc - xc) + (yc - yc) + (zc - zc) == 0.0f) {  // all three finite
    double r = __builtin_amdgcn_rcp(Z);
    double e = __builtin_fma(-Z, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-Z, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double qx = X * r, qy = Y * r;
    const double rx = __builtin_fma(-Z, qx, X), ry = __builtin_fma(-Z, qy, Y);
    xn = __builtin_fma(rx, r, qx);
    yn = __builtin_fma(ry, r, qy);
  } else {
    xn = X / Z;
    yn = Y / Z;
  }
}

// A3: PinholeProjection::operator() + distort, pinhole.hpp:13-51 (duplicate
// PointCloudProcessor.hpp:100-123), fp64, left-to-right as written.
template <bool kShortDiv = true>
__device__ __forceinline__ void project_uv(const DevCamera &c, float xc, float yc, float zc, double &u, double &v,
                                           bool finite_sure = false) {
  double xn, yn;
  if (kShortDiv && c.ds_fast) {
    if (finite_sure)  // (two copies of the sequence: a scalar branch instead of a test per lane)
      divide_xy_by_z(xc, yc, zc, xn, yn, true);
    else
      divide_xy_by_z(xc, yc, zc, xn, yn);
  } else {
    xn = static_cast<double>(xc) / static_cast<double>(zc);
    yn = static_cast<double>(yc) / static_cast<double>(zc);
  }
  const double x2 = xn * xn;
  const double y2 = yn * yn;
  const double r2 = x2 + y2;
  const double r4 = r2 * r2;
  const double r6 = r2 * r4;
  const double rc = ((1.0 + c.k1 * r2) + c.k2 * r4) + c.k3 * r6;
  const double t1 = (2.0 * xn) * yn;
  const double t2 = r2 + 2.0 * x2;
  const double t3 = r2 + 2.0 * y2;
  const double xd = (rc * xn + c.p1 * t1) + c.p2 * t2;
  const double yd = (rc * yn + c.p1 * t3) + c.p2 * t1;
  u = c.fx * xd + c.cx;
  v = c.fy * yd + c.cy;
}

// Correctly rounded x / ds (fp32) by a constant divisor: with r = RN(1 / ds) precomputed,
// q0 = RN(x r) is within 2 ulp, one residual correction makes it faithful, and a second one
// (exact residual by FMA, Markstein's theorem: r correctly rounded, q faithful) returns
// RN(x / ds).  The residuals are exact only away from the underflow / overflow ranges: for
// 2^-40 <= |x| <= 2^60 the result IS RN(x / ds) (pcp_selftest_arithmetic checks every such x against `/`
// for the configured ds).  Outside that window the value may differ from the quotient, but not what cull_cell
// makes of it (ds in [2^-20, 2^20], see ds_fast): below the window |x / ds| < 2^-20 and the sequence stays as
// small -- inside (-1, cull size), truncated to 0; above it |x / ds| > 2^40 and the sequence returns a value beyond
// +-2^39, an infinity or a NaN -- rejected like the quotient.  The self-test checks that decision for every other x.
__device__ __forceinline__ float div_by_ds(const DevCamera &c, float x) {
  if (c.ds_fast) {
    float q = x * c.ds_rcp;
    float e = __builtin_fmaf(-c.ds_f, q, x);
    q = __builtin_fmaf(e, c.ds_rcp, q);
    e = __builtin_fmaf(-c.ds_f, q, x);
    return __builtin_fmaf(e, c.ds_rcp, q);
  }
  return x / c.ds_f;
}
// what cull_cell makes of a quotient along one axis (`size` = the cull size as fp32): -1 rejected, else the cell coordinate
__device__ __forceinline__ int32_t cell_of_quotient(float q, float size) {
  // C truncation: 0 <= (int)t < W  <=>  -1 < t < W (W an integer below 2^24); NaN / inf / out-of-int32 fail
  return ((q > -1.0f) & (q < size)) ? static_cast<int32_t>(q) : -1;
}

// A4 cell: (project(p).cast<float>() / 14).cast<int>(), bounds vs the FULL cull
// size (view_culling.cpp:86-90, sic) then vs the /14 map (:116,:155).
// Returns cy*mw+cx, -1 (rejected) or, with the depth buffer off, -2 (candidate outside the map).
// Values that do not fit an int32 (UB in the reference, Appendix B6) are rejected.
template <bool kShortDiv = true>
__device__ __forceinline__ int32_t cull_cell(const DevCamera &c, double u, double v) {
  if (c.cull_mode == PCP_CULL_HPR_CANDIDATES) {
    // hidden_points_removal's filter, view_culling.cpp:284-288: project(p).cast<int>() against the FULL cull size.
    // There is no map in this mode: a candidate is reported as -2 ("candidate without a map cell") and kept.
    return (u > -1.0 && u < c.cull_wd && v > -1.0 && v < c.cull_hd) ? -2 : -1;
  }
  const float cxf = kShortDiv ? div_by_ds(c, static_cast<float>(u)) : static_cast<float>(u) / c.ds_f;
  const float cyf = kShortDiv ? div_by_ds(c, static_cast<float>(v)) : static_cast<float>(v) / c.ds_f;
  const int32_t cx = cell_of_quotient(cxf, c.cull_wf), cy = cell_of_quotient(cyf, c.cull_hf);
  if ((cx < 0) | (cy < 0)) return -1;
  // -2 (candidate without a map cell) is only reported when the depth buffer is off
  return ((cx < c.mw) & (cy < c.mh)) ? cy * c.mw + cx : (c.enable_zbuf ? -1 : -2);
}

// A5 pixel: static_cast<int>(fx*xd+cx) with C truncation, bounds vs the actual
// image (PointCloudProcessor.cpp:752-754).  -1 when rejected.
__device__ __forceinline__ int32_t colour_pixel(const DevCamera &c, double u, double v) {
  if (!(u > -1.0 && u < c.img_wd && v > -1.0 && v < c.img_hd)) return -1;  // same equivalence as above
  return static_cast<int32_t>(v) * c.img_w + static_cast<int32_t>(u);
}

// ||p_c|| in fp64 on the promoted fp32 camera coordinates (view_culling.cpp:102,144).
__device__ __forceinline__ double range64(float xc, float yc, float zc) {
  const double X = xc, Y = yc, Z = zc;
  return sqrt((X * X + Y * Y) + Z * Z);
}

// Conservative fp32 rejection test.  Returns true only when the reference's fp64
// projection of (xc, yc, zc), zc > 0, is CERTAIN to fail both the cull-cell rule
// and the colour-pixel rule, so that skipping the fp64 path cannot change any result.
//
// The distortion polynomial g(xn, yn) is evaluated in fp32 (FMAs: this arithmetic is
// ours, not the reference's) together with S = the same polynomial with every monomial
// replaced by its absolute value.  Error budget against the reference's fp64 value:
//   inputs: xn~ = xc * rcp(zc), relative error <= 2^-22 + 2^-24 (v_rcp_f32 is 1 ulp);
//           a relative perturbation e of (xn, yn) moves a degree-<=7 polynomial by at
//           most 7 e S  ->  <= 2.2 * 2^-20 S
//   rounding: <= 24 fp32 operations                      ->  <= 1.5 * 2^-20 S
//   total <= 0.93 * 2^-18 S; the test uses 2^-16 S (4x margin), the same on the final
//   u = fx xd + cx, and the acceptance box is widened by 0.5 px.
// Non-finite intermediates make every comparison false, i.e. "not rejected".
__device__ __forceinline__ bool surely_rejected(const DevCamera &c, float xc, float yc, float zc) {
  const float rz = __builtin_amdgcn_rcpf(zc);
  const float xn = xc * rz, yn = yc * rz;
  const float x2 = xn * xn, y2 = yn * yn;
  const float r2 = x2 + y2, r4 = r2 * r2, r6 = r2 * r4;
  const float rc = __builtin_fmaf(c.qk3, r6, __builtin_fmaf(c.qk2, r4, __builtin_fmaf(c.qk1, r2, 1.0f)));
  // the absolute-valued coefficients are |q..| as source modifiers of the same scalar registers (no copies held)
  const float ra = __builtin_fmaf(fabsf(c.qk3), r6, __builtin_fmaf(fabsf(c.qk2), r4, __builtin_fmaf(fabsf(c.qk1), r2, 1.0f)));
  const float t1 = 2.0f * xn * yn;
  const float t2 = __builtin_fmaf(2.0f, x2, r2), t3 = __builtin_fmaf(2.0f, y2, r2);
  const float at1 = fabsf(t1);
  const float xd = __builtin_fmaf(c.qp2, t2, __builtin_fmaf(c.qp1, t1, rc * xn));
  const float yd = __builtin_fmaf(c.qp2, t1, __builtin_fmaf(c.qp1, t3, rc * yn));
  const float sx = __builtin_fmaf(fabsf(c.qp2), t2, __builtin_fmaf(fabsf(c.qp1), at1, ra * fabsf(xn)));
  const float sy = __builtin_fmaf(fabsf(c.qp2), at1, __builtin_fmaf(fabsf(c.qp1), t3, ra * fabsf(yn)));
  const float u = __builtin_fmaf(c.qfx, xd, c.qcx), v = __builtin_fmaf(c.qfy, yd, c.qcy);
  constexpr float kErr = 1.52587890625e-05f;  // 2^-16
  const float eu = kErr * __builtin_fmaf(fabsf(c.qfx), sx, fabsf(c.qcx));
  const float ev = kErr * __builtin_fmaf(fabsf(c.qfy), sy, fabsf(c.qcy));
  // any of u + eu < u_lo, u - eu > u_hi, v + ev < v_lo, v - ev > v_hi -- as the sign of the largest excess (v_max3_f32 drops a
  // NaN operand as the comparison would be false; four NaNs compare false below)
  const float over = fmaxf(fmaxf(c.u_lo - (u + eu), (u - eu) - c.u_hi), fmaxf(c.v_lo - (v + ev), (v - ev) - c.v_hi));
  return over > 0.0f;
}

struct Projected {
  float xc, yc, zc;
  int32_t cell;   // >=0, -2, -1
  int32_t pixel;  // >=0, -1
};

// transform + z test (B6: z <= 0 rejected) + projection + both truncation rules.
// kPretest = false: the caller already knows the wavefront holds candidates (colour pass over
// the refined tile masks), so the fp32 rejection test could not skip anything.
// kShortDiv = false keeps the plain `/` (single-keyframe kernel: HBM-bound, and four points per lane
// make the extra code paths cost more registers than the divisions save).
template <bool kPretest = true, bool kShortDiv = true>
__device__ __forceinline__ Projected project_point(const DevCamera &c, const float *__restrict__ m, float x, float y,
                                                   float z, bool pretest_here = true, bool finite_sure = false) {
  Projected p;
  xform(m, x, y, z, p.xc, p.yc, p.zc);
  p.cell = -1;
  p.pixel = -1;
  // pretest_here is wave-uniform (a tile-level hint); the rejection test never changes a result
  if (p.zc > 0.0f && !(kPretest && pretest_here && c.pretest && surely_rejected(c, p.xc, p.yc, p.zc))) {
    double u, v;
    project_uv<kShortDiv>(c, p.xc, p.yc, p.zc, u, v, finite_sure);
    p.cell = cull_cell<kShortDiv>(c, u, v);
    p.pixel = colour_pixel(c, u, v);
  }
  return p;
}

// A4 keep rule (view_culling.cpp:135-171).  depth = this keyframe's map.
__device__ __forceinline__ bool keep_rule(const DevCamera &c, const Projected &p, const uint32_t *__restrict__ depth) {
  if (!c.enable_zbuf) return p.cell != -1;
  if (p.cell < 0) return false;
  const double r = range64(p.xc, p.yc, p.zc);
  const double lim = static_cast<double>(__uint_as_float(depth[p.cell])) + c.slack;
  return !(r > lim);
}

// B3: what the reference does to a coloured sample before scoring it (PointCloudProcessor.cpp:555,571-579).
//   p_w  = transformPointCloud(c2w) p_c            fp32, PCL SSE association (:555)
//   kdtree.radiusSearch(p_w, 1e-5) over the ORIGINAL cloud: flann::L2_Simple<float> distance
//          ((dx^2 + dy^2) + dz^2, fp32), reported iff < f32(1e-5^2) -- evaluated here for the sample's own map
//          point (wx0, wy0, wz0), which is the one match the reference finds unless map points lie < ~20 um apart
//   p_c' = c2w.inverse() * (p_w, 1)                Eigen 3.3.7 Affine3f * Vector4f: per row a 4-term fp32 sum
//          split 2 + 2 by Redux.h's scalar unroller: (m0 x + m1 y) + (m2 z + m3 * 1)
// Returns false when the point does not find itself (the reference then drops the sample); else (xc, yc, zc)
// become p_c'.
__device__ __forceinline__ bool roundtrip_sample(const DevCamera &c, const DevFrame &fr, float wx0, float wy0, float wz0,
                                                 float &xc, float &yc, float &zc) {
  float wx, wy, wz;
  xform(fr.c2w, xc, yc, zc, wx, wy, wz);
  const float dx = wx - wx0, dy = wy - wy0, dz = wz - wz0;
  float d2 = dx * dx;
  d2 += dy * dy;
  d2 += dz * dz;
  if (!(d2 < c.match_r2)) return false;
  const float *m = fr.c2w_inv;
  xc = (m[0] * wx + m[1] * wy) + (m[2] * wz + m[3]);
  yc = (m[4] * wx + m[5] * wy) + (m[6] * wz + m[7]);
  zc = (m[8] * wx + m[9] * wy) + (m[10] * wz + m[11]);
  return true;
}

// A6 scores: computeOrientationScore hpp:205-220 (B4 reproduced),
// computeDistanceScore hpp:222-236, final cpp:588; (xc, yc, zc) is p_c (PCP_MATCH_IDENTITY) or p_c' (PCP_MATCH_ROUNDTRIP).
//
// The orientation score is f32((cosA + 1) / 2) with cosA = RN(dz / RN(sqrt(sq))) in fp64: a correctly rounded square root and
// a correctly rounded division (~45 instructions) whose result is rounded to fp32 right away.  c = dz * y, y = 1 / sqrt(sq)
// from the hardware estimate and one third-order correction (7 instructions), is within 5e-16 of the real quotient, and so is
// the reference's cosA within 2.3e-16; the two values of (cosA + 1) / 2 differ by less than 1e-15.  When t - 4e-15 and
// t + 4e-15 round to the same fp32 number the reference's value rounds to it too; otherwise (1.3e-7 of the samples) and for
// magnitudes outside [2^-100, 2^100] the square root and the division are taken as written.
__device__ __forceinline__ float final_score(float xc, float yc, float zc, double px, double py, double pz) {
  const double dx = static_cast<double>(xc) - px;
  const double dy = static_cast<double>(yc) - py;
  const double dz = static_cast<double>(zc) - pz;
  const double sq = (dx * dx + dy * dy) + dz * dz;
  // (taken on every lane, accepted where it holds: one rare branch instead of two)
  const double y0 = __builtin_amdgcn_rsq(sq);
  const double e = __builtin_fma(-(sq * y0), y0, 1.0);
  const double y = __builtin_fma(y0 * e, __builtin_fma(0.375, e, 0.5), y0);
  const double t = (dz * y + 1.0) * 0.5;
  const float lo = static_cast<float>(t - 4e-15), hi = static_cast<float>(t + 4e-15);
  float o = lo;
  const bool settled = (sq > 0x1p-100) & (sq < 0x1p100) & (lo == hi);
  if (!settled) {
    const double cosA = sq > 0.0 ? dz / sqrt(sq) : dz;
    o = static_cast<float>((cosA + 1.0) / 2.0);
  }
  o = 0.2f + 0.8f * o;
  const float dist = sqrtf((xc * xc + yc * yc) + zc * zc);
  const float diff = fabsf(dist - 2.0f);
  float nd = diff / 2.0f;
  nd = nd < 1.0f ? nd : 1.0f;
  float d = 1.0f - nd;
  d = 0.2f + 0.8f * d;
  return static_cast<float>(static_cast<double>(o + d) / 2.0);
}

// Per-point top-5 held in registers across the keyframe loop (A8).  Streaming
// equivalent of "append all, std::sort descending, take 5" with ties -> lower
// keyframe index (B8): keyframes arrive ascending, a new entry goes behind equals.
struct Top5 {
  float s0, s1, s2, s3, s4;
  uint32_t c0, c1, c2, c3, c4;
  int32_t f0, f1, f2, f3, f4;
  int32_t count;
  __device__ __forceinline__ void init() {
    s0 = s1 = s2 = s3 = s4 = -1.0f;
    c0 = c1 = c2 = c3 = c4 = 0u;
    f0 = f1 = f2 = f3 = f4 = -1;
    count = 0;
  }
  __device__ __forceinline__ void insert(float s, uint32_t c, int32_t f) {
    count += 1;
    // scores are >= 0.2 and empty slots hold -1, so "s > slot" also fills empties
    if (s > s4) {
      s4 = s; c4 = c; f4 = f;
      if (s4 > s3) {
        float ts = s3; s3 = s4; s4 = ts;
        uint32_t tc = c3; c3 = c4; c4 = tc;
        int32_t tf = f3; f3 = f4; f4 = tf;
        if (s3 > s2) {
          ts = s2; s2 = s3; s3 = ts;
          tc = c2; c2 = c3; c3 = tc;
          tf = f2; f2 = f3; f3 = tf;
          if (s2 > s1) {
            ts = s1; s1 = s2; s2 = ts;
            tc = c1; c1 = c2; c2 = tc;
            tf = f1; f1 = f2; f2 = tf;
            if (s1 > s0) {
              ts = s0; s0 = s1; s1 = ts;
              tc = c0; c0 = c1; c1 = tc;
              tf = f0; f0 = f1; f1 = tf;
            }
          }
        }
      }
    }
  }
  // smoothColors PointCloudProcessor.cpp:616-629: fp32 sums in sorted order,
  // truncation to uint8, no entries -> (0,0,0) (B7).  Packed r | g<<8 | b<<16 | has<<24.
  __device__ __forceinline__ uint32_t finalise() const {
    float total = 0.0f, r = 0.0f, g = 0.0f, b = 0.0f;
    const float ss[5] = {s0, s1, s2, s3, s4};
    const uint32_t cc[5] = {c0, c1, c2, c3, c4};
    const int32_t ff[5] = {f0, f1, f2, f3, f4};
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      if (ff[k] >= 0) {
        const float s = ss[k];
        r += static_cast<float>((cc[k] >> 16) & 0xffu) * s;
        g += static_cast<float>((cc[k] >> 8) & 0xffu) * s;
        b += static_cast<float>(cc[k] & 0xffu) * s;
        total += s;
      }
    }
    if (f0 < 0) return 0u;
    const uint32_t ri = static_cast<uint32_t>(r / total) & 0xffu;
    const uint32_t gi = static_cast<uint32_t>(g / total) & 0xffu;
    const uint32_t bi = static_cast<uint32_t>(b / total) & 0xffu;
    const uint32_t has = (ri | gi | bi) ? 1u : 0u;  // removePointsWithNoColor hpp:238-252
    return ri | (gi << 8) | (bi << 16) | (has << 24);
  }
};

}  // namespace pcp
