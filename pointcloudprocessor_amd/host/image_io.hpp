// image_io.hpp -- the image side of the path without OpenCV:
//   cv::imread(path)                       -> read_image_bgr   (JPEG / PNG / PPM)
//   cv::imread(path, cv::IMREAD_GRAYSCALE) -> read_image_gray  (PNG / JPEG / PGM)
// (PCP/src/PointCloudProcessor.cpp:716,775).
//
// JPEG: baseline / extended-sequential Huffman, 8 bit, 1 or 3 components, sampling 1x1, 2x1, 2x2
// (4:4:4, 4:2:2, 4:2:0), restart intervals.  The arithmetic is libjpeg's, which is what OpenCV links:
// dequantisation inside the "islow" integer IDCT (jidctint.c), "fancy" triangle upsampling
// (jdsample.c h2v1 / h2v2), fixed-point YCbCr->RGB tables (jdcolor.c) [upstream libjpeg 6b /
// libjpeg-turbo, restated].  tests/test_image_io.py checks the decoder bit for bit against
// Pillow (libjpeg-turbo) on this image.  Progressive / arithmetic / CMYK files are rejected.
//
// PNG: 8 / 16 bit, gray / RGB / palette / alpha, non-interlaced; zlib inflate + the five
// scanline filters (RFC 2083).  16-bit samples are reduced to their high byte and alpha is
// dropped, as cv::imread's default flags do.
#pragma once

#include <zlib.h>

#include <cstdint>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace pcp_amd {

struct Image8 {
  int width = 0, height = 0, channels = 0;  // channels: 1 (gray) or 3 (B,G,R)
  std::vector<uint8_t> data;                // tightly packed rows
  bool empty() const { return data.empty(); }
};

namespace detail {

inline std::vector<uint8_t> read_file(const std::string &path) {
  std::ifstream in(path, std::ios::binary);
  if (!in) return {};
  return std::vector<uint8_t>((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
}

// ---------------------------------------------------------------- JPEG ------------------
struct JpegComponent {
  int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
  int blocks_w = 0, blocks_h = 0;      // padded to whole MCUs
  int down_w = 0, down_h = 0;          // downsampled_width / height (real samples)
  std::vector<uint8_t> plane;          // blocks_w*8 x blocks_h*8
  int pred = 0;
};

struct Huff {
  // canonical code -> symbol by code length (JPEG Annex C / F.2.2.3)
  int mincode[17], maxcode[18], valptr[17];
  uint8_t vals[256];
  bool present = false;
};

class JpegDecoder {
 public:
  explicit JpegDecoder(const std::vector<uint8_t> &buf) : d(buf) {}

  bool decode(Image8 &out, bool want_gray) {
    if (d.size() < 4 || d[0] != 0xFF || d[1] != 0xD8) return false;
    pos = 2;
    bool have_frame = false;
    for (;;) {
      int m = next_marker();
      if (m < 0) return false;
      if (m == 0xD9) break;  // EOI
      if (m == 0xDA) {       // SOS
        if (!have_frame || !read_sos()) return false;
        if (!decode_scan()) return false;
        break;  // baseline: one scan carries everything
      }
      const size_t len = be16(pos);
      if (len < 2 || pos + len > d.size()) return false;
      const size_t seg = pos + 2, end = pos + len;
      if (m == 0xC0 || m == 0xC1) {
        if (!read_sof(seg, end)) return false;
        have_frame = true;
      } else if (m == 0xC2 || (m >= 0xC5 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) {
        return false;  // progressive / lossless / arithmetic: not supported
      } else if (m == 0xC4) {
        if (!read_dht(seg, end)) return false;
      } else if (m == 0xDB) {
        if (!read_dqt(seg, end)) return false;
      } else if (m == 0xDD) {
        restart_interval = static_cast<int>(be16(seg));
      } else if (m == 0xEE) {  // Adobe: transform flag
        if (end - seg >= 12 && std::memcmp(&d[seg], "Adobe", 5) == 0) adobe_transform = d[seg + 11];
      }
      pos = end;
    }
    return finish(out, want_gray);
  }

 private:
  const std::vector<uint8_t> &d;
  size_t pos = 0;
  int width = 0, height = 0, ncomp = 0, hmax = 1, vmax = 1;
  JpegComponent comp[3];
  uint16_t quant[4][64] = {};
  Huff dc[4], ac[4];
  int restart_interval = 0;
  int adobe_transform = -1;
  // bit reader
  uint32_t bitbuf = 0;
  int bitcnt = 0;
  bool hit_marker = false;

  size_t be16(size_t p) const { return (static_cast<size_t>(d[p]) << 8) | d[p + 1]; }

  int next_marker() {
    while (pos + 1 < d.size()) {
      if (d[pos] != 0xFF) {
        ++pos;
        continue;
      }
      while (pos < d.size() && d[pos] == 0xFF) ++pos;
      if (pos >= d.size()) return -1;
      const int m = d[pos++];
      if (m != 0) return m;
    }
    return -1;
  }

  bool read_sof(size_t p, size_t end) {
    if (end - p < 6 || d[p] != 8) return false;  // 8-bit precision only
    height = static_cast<int>(be16(p + 1));
    width = static_cast<int>(be16(p + 3));
    ncomp = d[p + 5];
    if ((ncomp != 1 && ncomp != 3) || width <= 0 || height <= 0 || end - p < static_cast<size_t>(6 + 3 * ncomp)) return false;
    hmax = vmax = 1;
    for (int c = 0; c < ncomp; ++c) {
      comp[c].id = d[p + 6 + 3 * c];
      comp[c].h = d[p + 7 + 3 * c] >> 4;
      comp[c].v = d[p + 7 + 3 * c] & 15;
      comp[c].tq = d[p + 8 + 3 * c];
      if (comp[c].h < 1 || comp[c].h > 2 || comp[c].v < 1 || comp[c].v > 2 || comp[c].tq > 3) return false;
      hmax = std::max(hmax, comp[c].h);
      vmax = std::max(vmax, comp[c].v);
    }
    if (ncomp == 3 && (comp[1].h != 1 || comp[1].v != 1 || comp[2].h != 1 || comp[2].v != 1)) return false;
    if (ncomp == 1) comp[0].h = comp[0].v = hmax = vmax = 1;
    const int mcux = (width + 8 * hmax - 1) / (8 * hmax), mcuy = (height + 8 * vmax - 1) / (8 * vmax);
    for (int c = 0; c < ncomp; ++c) {
      comp[c].blocks_w = mcux * comp[c].h;
      comp[c].blocks_h = mcuy * comp[c].v;
      comp[c].down_w = (width * comp[c].h + hmax - 1) / hmax;
      comp[c].down_h = (height * comp[c].v + vmax - 1) / vmax;
      comp[c].plane.assign(static_cast<size_t>(comp[c].blocks_w) * 8 * comp[c].blocks_h * 8, 0);
    }
    return true;
  }

  bool read_dqt(size_t p, size_t end) {
    static const uint8_t zz[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                   41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                   30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
    while (p < end) {
      const int pq = d[p] >> 4, tq = d[p] & 15;
      ++p;
      if (tq > 3 || p + (pq ? 128 : 64) > end) return false;
      for (int k = 0; k < 64; ++k) {
        quant[tq][zz[k]] = pq ? static_cast<uint16_t>(be16(p)) : d[p];
        p += pq ? 2 : 1;
      }
    }
    return true;
  }

  bool read_dht(size_t p, size_t end) {
    while (p < end) {
      const int tc = d[p] >> 4, th = d[p] & 15;
      ++p;
      if (tc > 1 || th > 3 || p + 16 > end) return false;
      Huff &h = tc ? ac[th] : dc[th];
      int counts[17] = {0}, total = 0;
      for (int l = 1; l <= 16; ++l) total += counts[l] = d[p + l - 1];
      p += 16;
      if (total > 256 || p + total > end) return false;
      std::memcpy(h.vals, &d[p], static_cast<size_t>(total));
      p += total;
      int code = 0, k = 0;
      for (int l = 1; l <= 16; ++l) {
        h.valptr[l] = k;
        h.mincode[l] = code;
        code += counts[l];
        k += counts[l];
        h.maxcode[l] = counts[l] ? code - 1 : -1;
        code <<= 1;
      }
      h.maxcode[17] = 0x7fffffff;
      h.present = true;
    }
    return true;
  }

  bool read_sos() {
    const size_t len = be16(pos);
    const size_t p = pos + 2;
    if (pos + len > d.size()) return false;
    const int ns = d[p];
    if (ns != ncomp) return false;  // baseline files from cameras / OpenCV are single-scan interleaved
    for (int s = 0; s < ns; ++s) {
      const int id = d[p + 1 + 2 * s];
      bool found = false;
      for (int c = 0; c < ncomp; ++c)
        if (comp[c].id == id) {
          comp[c].td = d[p + 2 + 2 * s] >> 4;
          comp[c].ta = d[p + 2 + 2 * s] & 15;
          found = comp[c].td < 4 && comp[c].ta < 4 && dc[comp[c].td].present && ac[comp[c].ta].present;
        }
      if (!found) return false;
    }
    pos += len;
    return true;
  }

  // ---- entropy decoding ----
  void fill() {
    while (bitcnt <= 24) {
      int b = 0;
      if (!hit_marker && pos < d.size()) {
        b = d[pos];
        if (b == 0xFF) {
          const int n = pos + 1 < d.size() ? d[pos + 1] : 0xD9;
          if (n == 0) {
            pos += 2;
          } else {
            hit_marker = true;  // leave the marker in place, feed zeros
            b = 0;
          }
        } else {
          ++pos;
        }
      }
      bitbuf |= static_cast<uint32_t>(b) << (24 - bitcnt);
      bitcnt += 8;
    }
  }
  int getbits(int n) {
    if (n == 0) return 0;
    if (bitcnt < n) fill();
    const int v = static_cast<int>(bitbuf >> (32 - n));
    bitbuf <<= n;
    bitcnt -= n;
    return v;
  }
  int decode_huff(const Huff &h) {
    int code = 0;
    for (int l = 1; l <= 16; ++l) {
      code = (code << 1) | getbits(1);
      if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
    }
    return -1;
  }
  static int extend(int v, int n) { return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v; }

  bool decode_block(JpegComponent &c, int16_t blk[64]) {
    static const uint8_t zz[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                   41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                   30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
    std::memset(blk, 0, 64 * sizeof(int16_t));
    const int t = decode_huff(dc[c.td]);
    if (t < 0 || t > 11) return false;
    c.pred += t ? extend(getbits(t), t) : 0;
    blk[0] = static_cast<int16_t>(c.pred);
    for (int k = 1; k < 64;) {
      const int rs = decode_huff(ac[c.ta]);
      if (rs < 0) return false;
      const int r = rs >> 4, s = rs & 15;
      if (s == 0) {
        if (r != 15) break;  // EOB
        k += 16;
        continue;
      }
      k += r;
      if (k > 63) return false;
      blk[zz[k]] = static_cast<int16_t>(extend(getbits(s), s));
      ++k;
    }
    return true;
  }

  // jidctint.c jpeg_idct_islow: dequantise, 2-pass LL&M integer IDCT, range limit
  static void idct_islow(const int16_t *coef, const uint16_t *q, uint8_t *out, int stride) {
    constexpr int CONST_BITS = 13, PASS1_BITS = 2;
    constexpr int32_t F_0_298 = 2446, F_0_390 = 3196, F_0_541 = 4433, F_0_765 = 6270, F_0_899 = 7373, F_1_175 = 9633,
                      F_1_501 = 12299, F_1_847 = 15137, F_1_961 = 16069, F_2_053 = 16819, F_2_562 = 20995,
                      F_3_072 = 25172;
    auto descale = [](int32_t x, int n) { return (x + (1 << (n - 1))) >> n; };
    int32_t ws[64];
    for (int col = 0; col < 8; ++col) {
      const int16_t *in = coef + col;
      const uint16_t *qq = q + col;
      int32_t *w = ws + col;
      if (in[8] == 0 && in[16] == 0 && in[24] == 0 && in[32] == 0 && in[40] == 0 && in[48] == 0 && in[56] == 0) {
        const int32_t dcv = static_cast<int32_t>(in[0]) * qq[0] * (1 << PASS1_BITS);
        for (int r = 0; r < 8; ++r) w[8 * r] = dcv;
        continue;
      }
      int32_t z2 = in[16] * qq[16], z3 = in[48] * qq[48];
      int32_t z1 = (z2 + z3) * F_0_541;
      int32_t tmp2 = z1 + z3 * (-F_1_847), tmp3 = z1 + z2 * F_0_765;
      z2 = in[0] * qq[0];
      z3 = in[32] * qq[32];
      int32_t tmp0 = (z2 + z3) * (1 << CONST_BITS), tmp1 = (z2 - z3) * (1 << CONST_BITS);
      const int32_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
      tmp0 = in[56] * qq[56];
      tmp1 = in[40] * qq[40];
      tmp2 = in[24] * qq[24];
      tmp3 = in[8] * qq[8];
      z1 = tmp0 + tmp3;
      z2 = tmp1 + tmp2;
      z3 = tmp0 + tmp2;
      int32_t z4 = tmp1 + tmp3;
      const int32_t z5 = (z3 + z4) * F_1_175;
      tmp0 *= F_0_298;
      tmp1 *= F_2_053;
      tmp2 *= F_3_072;
      tmp3 *= F_1_501;
      z1 *= -F_0_899;
      z2 *= -F_2_562;
      z3 *= -F_1_961;
      z4 *= -F_0_390;
      z3 += z5;
      z4 += z5;
      tmp0 += z1 + z3;
      tmp1 += z2 + z4;
      tmp2 += z2 + z3;
      tmp3 += z1 + z4;
      w[0] = descale(tmp10 + tmp3, CONST_BITS - PASS1_BITS);
      w[56] = descale(tmp10 - tmp3, CONST_BITS - PASS1_BITS);
      w[8] = descale(tmp11 + tmp2, CONST_BITS - PASS1_BITS);
      w[48] = descale(tmp11 - tmp2, CONST_BITS - PASS1_BITS);
      w[16] = descale(tmp12 + tmp1, CONST_BITS - PASS1_BITS);
      w[40] = descale(tmp12 - tmp1, CONST_BITS - PASS1_BITS);
      w[24] = descale(tmp13 + tmp0, CONST_BITS - PASS1_BITS);
      w[32] = descale(tmp13 - tmp0, CONST_BITS - PASS1_BITS);
    }
    auto clamp8 = [](int32_t v) { return static_cast<uint8_t>(v < 0 ? 0 : (v > 255 ? 255 : v)); };
    for (int row = 0; row < 8; ++row) {
      const int32_t *w = ws + 8 * row;
      uint8_t *o = out + row * stride;
      int32_t z2 = w[2], z3 = w[6];
      int32_t z1 = (z2 + z3) * F_0_541;
      int32_t tmp2 = z1 + z3 * (-F_1_847), tmp3 = z1 + z2 * F_0_765;
      int32_t tmp0 = (w[0] + w[4]) * (1 << CONST_BITS), tmp1 = (w[0] - w[4]) * (1 << CONST_BITS);
      const int32_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
      tmp0 = w[7];
      tmp1 = w[5];
      tmp2 = w[3];
      tmp3 = w[1];
      z1 = tmp0 + tmp3;
      z2 = tmp1 + tmp2;
      z3 = tmp0 + tmp2;
      int32_t z4 = tmp1 + tmp3;
      const int32_t z5 = (z3 + z4) * F_1_175;
      tmp0 *= F_0_298;
      tmp1 *= F_2_053;
      tmp2 *= F_3_072;
      tmp3 *= F_1_501;
      z1 *= -F_0_899;
      z2 *= -F_2_562;
      z3 *= -F_1_961;
      z4 *= -F_0_390;
      z3 += z5;
      z4 += z5;
      tmp0 += z1 + z3;
      tmp1 += z2 + z4;
      tmp2 += z2 + z3;
      tmp3 += z1 + z4;
      constexpr int S = CONST_BITS + PASS1_BITS + 3;
      o[0] = clamp8(descale(tmp10 + tmp3, S) + 128);
      o[7] = clamp8(descale(tmp10 - tmp3, S) + 128);
      o[1] = clamp8(descale(tmp11 + tmp2, S) + 128);
      o[6] = clamp8(descale(tmp11 - tmp2, S) + 128);
      o[2] = clamp8(descale(tmp12 + tmp1, S) + 128);
      o[5] = clamp8(descale(tmp12 - tmp1, S) + 128);
      o[3] = clamp8(descale(tmp13 + tmp0, S) + 128);
      o[4] = clamp8(descale(tmp13 - tmp0, S) + 128);
    }
  }

  bool decode_scan() {
    const int mcux = (width + 8 * hmax - 1) / (8 * hmax), mcuy = (height + 8 * vmax - 1) / (8 * vmax);
    bitbuf = 0;
    bitcnt = 0;
    hit_marker = false;
    int16_t blk[64];
    int todo = restart_interval;
    int expected_rst = 0;
    for (int my = 0; my < mcuy; ++my)
      for (int mx = 0; mx < mcux; ++mx) {
        if (restart_interval && todo == 0) {
          // byte-align, expect RSTn
          bitbuf = 0;
          bitcnt = 0;
          hit_marker = false;
          while (pos + 1 < d.size() && !(d[pos] == 0xFF && d[pos + 1] >= 0xD0 && d[pos + 1] <= 0xD7)) ++pos;
          if (pos + 1 >= d.size()) return false;
          if ((d[pos + 1] & 7) != expected_rst) return false;
          expected_rst = (expected_rst + 1) & 7;
          pos += 2;
          for (int c = 0; c < ncomp; ++c) comp[c].pred = 0;
          todo = restart_interval;
        }
        for (int c = 0; c < ncomp; ++c)
          for (int by = 0; by < comp[c].v; ++by)
            for (int bx = 0; bx < comp[c].h; ++bx) {
              if (!decode_block(comp[c], blk)) return false;
              const int stride = comp[c].blocks_w * 8;
              uint8_t *o = comp[c].plane.data() + static_cast<size_t>((my * comp[c].v + by) * 8) * stride +
                           static_cast<size_t>(mx * comp[c].h + bx) * 8;
              idct_islow(blk, quant[comp[c].tq], o, stride);
            }
        if (restart_interval) --todo;
      }
    return true;
  }

  // jdsample.c: fullsize copy, h2v1_fancy_upsample, h2v2_fancy_upsample into a width x height plane.
  // Edge rows / columns replicate the last REAL sample (downsampled_width / height), as the
  // decompressor's main controller does.
  std::vector<uint8_t> upsample(const JpegComponent &c) const {
    std::vector<uint8_t> out(static_cast<size_t>(width) * height);
    const int stride = c.blocks_w * 8;
    const uint8_t *p = c.plane.data();
    if (c.h == hmax && c.v == vmax) {
      for (int y = 0; y < height; ++y) std::memcpy(&out[static_cast<size_t>(y) * width], p + static_cast<size_t>(y) * stride, width);
      return out;
    }
    const int dw = c.down_w, dh = c.down_h;
    std::vector<uint8_t> row(static_cast<size_t>(dw) * 2 + 4);
    auto emit = [&](int y) {
      if (y < height) std::memcpy(&out[static_cast<size_t>(y) * width], row.data(), width);
    };
    if (c.h * 2 == hmax && c.v == vmax) {  // h2v1
      for (int y = 0; y < height; ++y) {
        const uint8_t *in = p + static_cast<size_t>(y) * stride;
        if (dw > 2) {
          int k = 0;
          int inv = in[0];
          row[k++] = static_cast<uint8_t>(inv);
          row[k++] = static_cast<uint8_t>((inv * 3 + in[1] + 2) >> 2);
          for (int i = 1; i < dw - 1; ++i) {
            inv = in[i] * 3;
            row[k++] = static_cast<uint8_t>((inv + in[i - 1] + 1) >> 2);
            row[k++] = static_cast<uint8_t>((inv + in[i + 1] + 2) >> 2);
          }
          inv = in[dw - 1];
          row[k++] = static_cast<uint8_t>((inv * 3 + in[dw - 2] + 1) >> 2);
          row[k++] = static_cast<uint8_t>(inv);
        } else {
          for (int i = 0; i < dw; ++i) row[2 * i] = row[2 * i + 1] = in[i];
        }
        emit(y);
      }
      return out;
    }
    if (c.h * 2 == hmax && c.v * 2 == vmax) {  // h2v2
      for (int r = 0; r < dh; ++r) {
        const uint8_t *in0 = p + static_cast<size_t>(r) * stride;
        for (int v = 0; v < 2; ++v) {
          const int rn = v == 0 ? (r > 0 ? r - 1 : 0) : (r + 1 < dh ? r + 1 : dh - 1);
          const uint8_t *in1 = p + static_cast<size_t>(rn) * stride;
          if (dw > 2) {
            int k = 0;
            int thiscol = in0[0] * 3 + in1[0], nextcol = in0[1] * 3 + in1[1], lastcol;
            row[k++] = static_cast<uint8_t>((thiscol * 4 + 8) >> 4);
            row[k++] = static_cast<uint8_t>((thiscol * 3 + nextcol + 7) >> 4);
            lastcol = thiscol;
            thiscol = nextcol;
            for (int i = 2; i < dw; ++i) {
              nextcol = in0[i] * 3 + in1[i];
              row[k++] = static_cast<uint8_t>((thiscol * 3 + lastcol + 8) >> 4);
              row[k++] = static_cast<uint8_t>((thiscol * 3 + nextcol + 7) >> 4);
              lastcol = thiscol;
              thiscol = nextcol;
            }
            row[k++] = static_cast<uint8_t>((thiscol * 3 + lastcol + 8) >> 4);
            row[k++] = static_cast<uint8_t>((thiscol * 4 + 7) >> 4);
          } else {
            for (int i = 0; i < dw; ++i) row[2 * i] = row[2 * i + 1] = in0[i];  // int_upsample (box)
          }
          emit(2 * r + v);
        }
      }
      return out;
    }
    throw std::runtime_error("unsupported chroma sampling");
  }

  bool finish(Image8 &out, bool want_gray) {
    out.width = width;
    out.height = height;
    const size_t px = static_cast<size_t>(width) * height;
    const std::vector<uint8_t> Y = upsample(comp[0]);
    if (ncomp == 1 || want_gray) {
      // IMREAD_GRAYSCALE on a colour JPEG: libjpeg delivers the luma plane (out_color_space = JCS_GRAYSCALE)
      if (want_gray) {
        out.channels = 1;
        out.data = Y;
      } else {
        out.channels = 3;
        out.data.resize(3 * px);
        for (size_t i = 0; i < px; ++i) out.data[3 * i] = out.data[3 * i + 1] = out.data[3 * i + 2] = Y[i];
      }
      return true;
    }
    const std::vector<uint8_t> Cb = upsample(comp[1]), Cr = upsample(comp[2]);
    // jdcolor.c build_ycc_rgb_table: SCALEBITS 16
    int cr_r[256], cb_b[256];
    int32_t cr_g[256], cb_g[256];
    for (int i = 0; i < 256; ++i) {
      const int x = i - 128;
      cr_r[i] = static_cast<int>((91881LL * x + 32768) >> 16);    // FIX(1.40200)
      cb_b[i] = static_cast<int>((116130LL * x + 32768) >> 16);   // FIX(1.77200)
      cr_g[i] = static_cast<int32_t>(-46802LL * x);               // -FIX(0.71414)
      cb_g[i] = static_cast<int32_t>(-22554LL * x + 32768);       // -FIX(0.34414) + ONE_HALF
    }
    auto clamp8 = [](int v) { return static_cast<uint8_t>(v < 0 ? 0 : (v > 255 ? 255 : v)); };
    out.channels = 3;
    out.data.resize(3 * px);
    for (size_t i = 0; i < px; ++i) {
      const int y = Y[i], cb = Cb[i], cr = Cr[i];
      out.data[3 * i + 2] = clamp8(y + cr_r[cr]);
      out.data[3 * i + 1] = clamp8(y + static_cast<int>((cb_g[cb] + cr_g[cr]) >> 16));
      out.data[3 * i + 0] = clamp8(y + cb_b[cb]);
    }
    return true;
  }
};

// ---------------------------------------------------------------- PNG -------------------
inline bool decode_png(const std::vector<uint8_t> &d, Image8 &out, bool want_gray) {
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  if (d.size() < 33 || std::memcmp(d.data(), sig, 8) != 0) return false;
  size_t p = 8;
  int w = 0, h = 0, depth = 0, ctype = 0, interlace = 0;
  std::vector<uint8_t> idat, palette;
  auto be32 = [&](size_t q) { return (static_cast<uint32_t>(d[q]) << 24) | (d[q + 1] << 16) | (d[q + 2] << 8) | d[q + 3]; };
  while (p + 12 <= d.size()) {
    const uint32_t len = be32(p);
    if (p + 12 + len > d.size()) return false;
    const char *type = reinterpret_cast<const char *>(&d[p + 4]);
    const size_t body = p + 8;
    if (!std::memcmp(type, "IHDR", 4)) {
      w = static_cast<int>(be32(body));
      h = static_cast<int>(be32(body + 4));
      depth = d[body + 8];
      ctype = d[body + 9];
      interlace = d[body + 12];
    } else if (!std::memcmp(type, "PLTE", 4)) {
      palette.assign(d.begin() + body, d.begin() + body + len);
    } else if (!std::memcmp(type, "IDAT", 4)) {
      idat.insert(idat.end(), d.begin() + body, d.begin() + body + len);
    } else if (!std::memcmp(type, "IEND", 4)) {
      break;
    }
    p += 12 + len;
  }
  if (w <= 0 || h <= 0 || interlace != 0 || (depth != 8 && depth != 16)) return false;
  const int samples = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
  if (!samples || (ctype == 3 && (depth != 8 || palette.size() < 3))) return false;
  const int bpp = samples * depth / 8;
  const size_t rowbytes = static_cast<size_t>(w) * bpp;
  std::vector<uint8_t> raw((rowbytes + 1) * h);
  uLongf rawlen = static_cast<uLongf>(raw.size());
  if (uncompress(raw.data(), &rawlen, idat.data(), static_cast<uLong>(idat.size())) != Z_OK || rawlen != raw.size()) return false;
  std::vector<uint8_t> prev(rowbytes, 0), cur(rowbytes);
  out.width = w;
  out.height = h;
  out.channels = want_gray ? 1 : 3;
  out.data.resize(static_cast<size_t>(w) * h * out.channels);
  for (int y = 0; y < h; ++y) {
    const uint8_t *src = &raw[(rowbytes + 1) * y];
    const int ft = src[0];
    for (size_t i = 0; i < rowbytes; ++i) {
      const int a = i >= static_cast<size_t>(bpp) ? cur[i - bpp] : 0, b = prev[i], c = i >= static_cast<size_t>(bpp) ? prev[i - bpp] : 0;
      int v = src[1 + i];
      switch (ft) {
        case 0: break;
        case 1: v += a; break;
        case 2: v += b; break;
        case 3: v += (a + b) >> 1; break;
        case 4: {
          const int pa = std::abs(b - c), pb = std::abs(a - c), pc = std::abs(a + b - 2 * c);
          v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
          break;
        }
        default: return false;
      }
      cur[i] = static_cast<uint8_t>(v);
    }
    const int step = depth / 8;  // 16-bit: keep the high byte
    for (int x = 0; x < w; ++x) {
      const uint8_t *s = &cur[static_cast<size_t>(x) * bpp];
      uint8_t r, g, bl;
      if (ctype == 0 || ctype == 4) {
        r = g = bl = s[0];
      } else if (ctype == 3) {
        const size_t pi = static_cast<size_t>(s[0]) * 3;
        if (pi + 2 >= palette.size() + 0 && pi + 2 > palette.size() - 1) return false;
        r = palette[pi];
        g = palette[pi + 1];
        bl = palette[pi + 2];
      } else {
        r = s[0];
        g = s[step];
        bl = s[2 * step];
      }
      uint8_t *o = &out.data[(static_cast<size_t>(y) * w + x) * out.channels];
      if (want_gray) {
        // cv::cvtColor RGB->gray, 8 bit fixed point (B 1868, G 9617, R 4899, shift 14)
        o[0] = (r == g && g == bl) ? r : static_cast<uint8_t>((r * 4899 + g * 9617 + bl * 1868 + 8192) >> 14);
      } else {
        o[0] = bl;
        o[1] = g;
        o[2] = r;
      }
    }
    prev.swap(cur);
  }
  return true;
}

inline bool decode_pnm(const std::vector<uint8_t> &d, Image8 &out, bool want_gray) {
  if (d.size() < 7 || d[0] != 'P' || (d[1] != '5' && d[1] != '6')) return false;
  const int ch = d[1] == '6' ? 3 : 1;
  size_t p = 2;
  auto next_int = [&]() {
    for (;;) {
      while (p < d.size() && std::isspace(d[p])) ++p;
      if (p < d.size() && d[p] == '#') {
        while (p < d.size() && d[p] != '\n') ++p;
      } else {
        break;
      }
    }
    int v = 0;
    while (p < d.size() && std::isdigit(d[p])) v = v * 10 + (d[p++] - '0');
    return v;
  };
  const int w = next_int(), h = next_int(), maxv = next_int();
  ++p;
  if (w <= 0 || h <= 0 || maxv != 255 || p + static_cast<size_t>(w) * h * ch > d.size()) return false;
  out.width = w;
  out.height = h;
  out.channels = want_gray ? 1 : 3;
  out.data.resize(static_cast<size_t>(w) * h * out.channels);
  for (size_t i = 0; i < static_cast<size_t>(w) * h; ++i) {
    const uint8_t *s = &d[p + i * ch];
    const uint8_t r = s[0], g = ch == 3 ? s[1] : s[0], b = ch == 3 ? s[2] : s[0];
    if (want_gray) {
      out.data[i] = (ch == 1) ? r : static_cast<uint8_t>((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14);
    } else {
      out.data[3 * i] = b;
      out.data[3 * i + 1] = g;
      out.data[3 * i + 2] = r;
    }
  }
  return true;
}

inline bool decode_any(const std::string &path, Image8 &out, bool want_gray) {
  const std::vector<uint8_t> buf = read_file(path);
  if (buf.size() < 4) return false;
  try {
    if (buf[0] == 0xFF && buf[1] == 0xD8) {
      JpegDecoder dec(buf);
      return dec.decode(out, want_gray);
    }
    if (buf[0] == 0x89 && buf[1] == 'P') return decode_png(buf, out, want_gray);
    if (buf[0] == 'P') return decode_pnm(buf, out, want_gray);
  } catch (const std::exception &) {
    return false;
  }
  return false;
}

}  // namespace detail

// cv::imread(path): 3-channel BGR; empty image on failure (as cv::Mat::empty())
inline Image8 read_image_bgr(const std::string &path) {
  Image8 img;
  if (!detail::decode_any(path, img, false)) img = Image8{};
  return img;
}

// cv::imread(path, cv::IMREAD_GRAYSCALE)
inline Image8 read_image_gray(const std::string &path) {
  Image8 img;
  if (!detail::decode_any(path, img, true)) img = Image8{};
  return img;
}

}  // namespace pcp_amd
