// pcp_hsv.hpp -- the 8-bit BGR -> HSV -> BGR round trip generateColorMap applies to every keyframe
// (PointCloudProcessor.cpp:722-741), as OpenCV 4.2.0 (the version osrf/ros:noetic ships, /root/reference/
// Dockerfile:2) computes it for CV_8UC3: imgproc/src/color_hsv.simd.hpp RGB2HSV_b / HSV2RGB_b [upstream: OpenCV is
// not under /root/reference; restated from its published source].
//
// Forward (exact, integer): v = max, diff = max - min, s = (diff * sdiv[v] + 2^11) >> 12,
//   h = (sector term) * hdiv180[diff], rounded the same way, + 180 when negative;
//   sdiv[i] = cvRound((255 << 12) / (1. * i)), hdiv180[i] = cvRound((180 << 12) / (6. * i)), [0] = 0.
// Middle: S, V <- saturate_cast<uchar>(x * scale) (fp32 product, cvRound = round half to even, clamp).
// Backward (fp32, OpenCV's scalar routine HSV2RGB_native; every operation individually rounded -- this file is
// built with -ffp-contract=off): s, v *= 1.f / 255.f; s == 0 -> grey; else h *= 6.f / 180, h = fmod(h, 6),
//   sector = floor(h), h -= sector, tab = {v, v (1 - s), v (1 - s h), v (1 - s (1 - h))}, channel = tab[...] and
//   saturate_cast<uchar>(c * 255.f).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace pcp {

constexpr int kHsvShift = 12;

// host: the two tables exactly as RGB2HSV_b's static initialiser builds them (fp64 quotient, cvRound)
inline void hsv_build_tables(int32_t sdiv[256], int32_t hdiv180[256]) {
  sdiv[0] = hdiv180[0] = 0;
  for (int i = 1; i < 256; ++i) {
    sdiv[i] = static_cast<int32_t>(__builtin_rint((255 << kHsvShift) / (1. * i)));
    hdiv180[i] = static_cast<int32_t>(__builtin_rint((180 << kHsvShift) / (6. * i)));
  }
}

__device__ __forceinline__ uint32_t sat_u8_from_float(float x) {  // cv::saturate_cast<uchar>(float): cvRound, clamp
  const int v = __float2int_rn(x);
  return static_cast<uint32_t>(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// b, g, r in / out (0..255)
__device__ __forceinline__ void hsv_round_trip(const int32_t *__restrict__ sdiv, const int32_t *__restrict__ hdiv180,
                                               float sat_scale, float val_scale, uint32_t &b8, uint32_t &g8,
                                               uint32_t &r8) {
  const int b = static_cast<int>(b8), g = static_cast<int>(g8), r = static_cast<int>(r8);
  // ---- RGB2HSV_b (bidx = 0: BGR input, hrange = 180)
  int v = b > g ? b : g;
  v = v > r ? v : r;
  int vmin = b < g ? b : g;
  vmin = vmin < r ? vmin : r;
  const int diff = v - vmin;
  const int vr = v == r ? -1 : 0, vg = v == g ? -1 : 0;
  const int s = (diff * sdiv[v] + (1 << (kHsvShift - 1))) >> kHsvShift;
  int h = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
  h = (h * hdiv180[diff] + (1 << (kHsvShift - 1))) >> kHsvShift;
  h += h < 0 ? 180 : 0;
  const uint32_t H = static_cast<uint32_t>(h < 0 ? 0 : (h > 255 ? 255 : h));  // saturate_cast<uchar>(h)
  // ---- hsv.at<Vec3b>[1] = saturate_cast<uchar>(S * saturation_scale), [2] likewise (:733-734)
  const uint32_t S = sat_u8_from_float(static_cast<float>(s & 0xff) * sat_scale);
  const uint32_t V = sat_u8_from_float(static_cast<float>(v) * val_scale);
  // ---- HSV2RGB_b -> HSV2RGB_native (hscale = 6.f / 180)
  float fh = static_cast<float>(H);
  const float fs = static_cast<float>(S) * (1.0f / 255.0f);
  const float fv = static_cast<float>(V) * (1.0f / 255.0f);
  float fb, fg, fr;
  if (fs == 0.0f) {
    fb = fg = fr = fv;
  } else {
    const float hscale = 6.0f / 180.0f;
    fh *= hscale;
    fh = fmodf(fh, 6.0f);
    int sector = static_cast<int>(floorf(fh));
    fh -= static_cast<float>(sector);
    if (static_cast<unsigned>(sector) >= 6u) {
      sector = 0;
      fh = 0.0f;
    }
    const float t0 = fv;
    const float t1 = fv * (1.0f - fs);
    const float t2 = fv * (1.0f - fs * fh);
    const float t3 = fv * (1.0f - fs * (1.0f - fh));
    // sector_data = {{1,3,0}, {1,0,2}, {3,0,1}, {0,2,1}, {0,1,3}, {2,1,0}} -> (b, g, r)
    switch (sector) {
      case 0: fb = t1; fg = t3; fr = t0; break;
      case 1: fb = t1; fg = t0; fr = t2; break;
      case 2: fb = t3; fg = t0; fr = t1; break;
      case 3: fb = t0; fg = t2; fr = t1; break;
      case 4: fb = t0; fg = t1; fr = t3; break;
      default: fb = t2; fg = t1; fr = t0; break;
    }
  }
  b8 = sat_u8_from_float(fb * 255.0f);
  g8 = sat_u8_from_float(fg * 255.0f);
  r8 = sat_u8_from_float(fr * 255.0f);
}

}  // namespace pcp
