// pcd_io.hpp -- the PCD subset PointCloudProcessor reads and writes, without PCL.
//
// Reader: .pcd v0.7, DATA ascii | binary | binary_compressed (LZF), any field list
// that contains x y z (float32), optional intensity (float32) and rgb (packed).
// What the reference calls: pcl::io::loadPCDFile<PointXYZI> (PCP/src/PointCloudProcessor.cpp:112,148,
// PCP/src/cloudSmooth.cpp:92).
//
// Writer: the ASCII layout of pcl::PCDWriter::writeASCII (PCL 1.10 pcd_io.hpp
// [upstream]): header lines VERSION/FIELDS/SIZE/TYPE/COUNT/WIDTH/HEIGHT/VIEWPOINT/
// POINTS/DATA, then one point per line, floats printed with 8 significant digits
// (ostream precision 8 == "%.8g"), the `rgb` field as the packed uint32 with TYPE U.
// Call sites: savePCDFileASCII :135 (x y z intensity), writeASCII :217 (culled cloud,
// x y z intensity), :542 (x y z rgb segmentMask), :920 (x y z rgb).
#pragma once

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <algorithm>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

namespace pcp_amd {

struct XYZICloud {  // pcl::PointCloud<pcl::PointXYZI>
  std::vector<float> x, y, z, intensity;
  size_t size() const { return x.size(); }
  void resize(size_t n) {
    x.resize(n);
    y.resize(n);
    z.resize(n);
    intensity.resize(n);
  }
  void push_back(float a, float b, float c, float i) {
    x.push_back(a);
    y.push_back(b);
    z.push_back(c);
    intensity.push_back(i);
  }
};

struct PcdField {
  std::string name;
  int size = 4;
  char type = 'F';
  int count = 1;
  int offset = 0;
};

namespace detail {
// LZF decompression (Marc Lehmann's format, the codec of PCL's binary_compressed PCD files [upstream lzf.cpp]):
// a control byte < 32 starts a literal run of ctrl + 1 bytes; otherwise a back reference of length
// (ctrl >> 5) + 2 (plus an extension byte when the 3-bit length is 7) at distance ((ctrl & 31) << 8 | next) + 1.
inline bool lzf_decompress(const unsigned char *in, size_t in_len, unsigned char *out, size_t out_len) {
  size_t ip = 0, op = 0;
  while (ip < in_len) {
    unsigned ctrl = in[ip++];
    if (ctrl < 32) {
      const size_t run = ctrl + 1;
      if (ip + run > in_len || op + run > out_len) return false;
      std::memcpy(out + op, in + ip, run);
      ip += run;
      op += run;
    } else {
      size_t len = ctrl >> 5;
      if (len == 7) {
        if (ip >= in_len) return false;
        len += in[ip++];
      }
      if (ip >= in_len) return false;
      const size_t dist = ((ctrl & 31u) << 8 | in[ip++]) + 1;
      len += 2;
      if (dist > op || op + len > out_len) return false;
      for (size_t k = 0; k < len; ++k, ++op) out[op] = out[op - dist];  // may overlap: byte by byte
    }
  }
  return op == out_len;
}
}  // namespace detail

inline int loadPCDFile(const std::string &path, XYZICloud &cloud) {
  std::ifstream in(path, std::ios::binary);
  if (!in) return -1;
  std::vector<PcdField> fields;
  size_t points = 0, width = 0, height = 1;
  std::string data_mode, line;
  while (std::getline(in, line)) {
    if (!line.empty() && line.back() == '\r') line.pop_back();
    if (line.empty() || line[0] == '#') continue;
    std::istringstream ls(line);
    std::string key;
    ls >> key;
    if (key == "FIELDS" || key == "COLUMNS") {
      std::string f;
      while (ls >> f) {
        PcdField pf;
        pf.name = f;
        fields.push_back(pf);
      }
    } else if (key == "SIZE") {
      for (auto &f : fields) ls >> f.size;
    } else if (key == "TYPE") {
      for (auto &f : fields) ls >> f.type;
    } else if (key == "COUNT") {
      for (auto &f : fields) ls >> f.count;
    } else if (key == "WIDTH") {
      ls >> width;
    } else if (key == "HEIGHT") {
      ls >> height;
    } else if (key == "POINTS") {
      ls >> points;
    } else if (key == "DATA") {
      ls >> data_mode;
      break;
    }
  }
  if (fields.empty() || data_mode.empty()) return -1;
  if (points == 0) points = width * height;
  int off = 0, ix = -1, iy = -1, iz = -1, ii = -1;
  for (size_t k = 0; k < fields.size(); ++k) {
    fields[k].offset = off;
    off += fields[k].size * fields[k].count;
    if (fields[k].name == "x") ix = static_cast<int>(k);
    if (fields[k].name == "y") iy = static_cast<int>(k);
    if (fields[k].name == "z") iz = static_cast<int>(k);
    if (fields[k].name == "intensity") ii = static_cast<int>(k);
  }
  if (ix < 0 || iy < 0 || iz < 0) return -1;
  for (int k : {ix, iy, iz})
    if (fields[static_cast<size_t>(k)].type != 'F' || fields[static_cast<size_t>(k)].size != 4) return -1;
  cloud.resize(points);
  if (data_mode == "ascii") {
    // column index of every scalar
    std::vector<int> first_col(fields.size());
    int col = 0;
    for (size_t k = 0; k < fields.size(); ++k) {
      first_col[k] = col;
      col += fields[k].count;
    }
    std::vector<std::string> tok;
    for (size_t p = 0; p < points; ++p) {
      if (!std::getline(in, line)) return -1;
      tok.clear();
      std::istringstream ls(line);
      std::string t;
      while (ls >> t) tok.push_back(t);
      if (static_cast<int>(tok.size()) < col) return -1;
      auto val = [&](int f) { return std::strtof(tok[static_cast<size_t>(first_col[static_cast<size_t>(f)])].c_str(), nullptr); };
      cloud.x[p] = val(ix);
      cloud.y[p] = val(iy);
      cloud.z[p] = val(iz);
      cloud.intensity[p] = ii >= 0 ? val(ii) : 0.0f;
    }
  } else if (data_mode == "binary") {
    std::vector<char> rec(static_cast<size_t>(off));
    for (size_t p = 0; p < points; ++p) {
      in.read(rec.data(), off);
      if (!in) return -1;
      std::memcpy(&cloud.x[p], rec.data() + fields[static_cast<size_t>(ix)].offset, 4);
      std::memcpy(&cloud.y[p], rec.data() + fields[static_cast<size_t>(iy)].offset, 4);
      std::memcpy(&cloud.z[p], rec.data() + fields[static_cast<size_t>(iz)].offset, 4);
      if (ii >= 0 && fields[static_cast<size_t>(ii)].size == 4)
        std::memcpy(&cloud.intensity[p], rec.data() + fields[static_cast<size_t>(ii)].offset, 4);
      else
        cloud.intensity[p] = 0.0f;
    }
  } else if (data_mode == "binary_compressed") {
    // PCDWriter::writeBinaryCompressed [upstream pcd_io.cpp]: uint32 compressed size, uint32 uncompressed size,
    // then the LZF stream of the cloud stored field by field (all x, all y, ... : SoA), each field `points`
    // entries of size * count bytes
    uint32_t csize = 0, usize = 0;
    in.read(reinterpret_cast<char *>(&csize), 4);
    in.read(reinterpret_cast<char *>(&usize), 4);
    if (!in || usize != static_cast<uint64_t>(off) * points) return -1;
    std::vector<unsigned char> comp(csize), raw(usize);
    in.read(reinterpret_cast<char *>(comp.data()), csize);
    if (!in || !detail::lzf_decompress(comp.data(), csize, raw.data(), usize)) return -1;
    auto column = [&](int f) { return raw.data() + static_cast<size_t>(fields[static_cast<size_t>(f)].offset) * points; };
    std::memcpy(cloud.x.data(), column(ix), points * 4);
    std::memcpy(cloud.y.data(), column(iy), points * 4);
    std::memcpy(cloud.z.data(), column(iz), points * 4);
    if (ii >= 0 && fields[static_cast<size_t>(ii)].size == 4 && fields[static_cast<size_t>(ii)].count == 1)
      std::memcpy(cloud.intensity.data(), column(ii), points * 4);
    else
      std::fill(cloud.intensity.begin(), cloud.intensity.end(), 0.0f);
  } else {
    return -1;
  }
  return 0;
}

namespace detail {
// "%.8g" of a float, the text `ostream << float` produces at precision 8 (PCDWriter::writeASCII [upstream]).
// Values printed in fixed notation (decimal exponent -4 .. 7, i.e. every coordinate of a metre-scale map) are
// formatted with exact integer arithmetic: |v| = m * 2^e, the 8 significant digits are round-half-even of
// m * 5^k * 2^(e + k), k = 7 - exponent, which fits 128 bits.  Everything else goes through snprintf.
// host/format_selftest.cpp compares the two on random bit patterns (tests/test_cli.py).
inline bool put_float_fixed(std::string &out, float v) {
  static const double kPow10[] = {1e-4, 1e-3, 1e-2, 1e-1, 1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8};
  static const uint64_t kPow5[] = {1ull, 5ull, 25ull, 125ull, 625ull, 3125ull, 15625ull, 78125ull, 390625ull,
                                   1953125ull, 9765625ull, 48828125ull};
  const double a = std::fabs(static_cast<double>(v));
  if (!(a >= 1e-4 && a < 1e8)) return false;
  int X = -4;
  while (a >= kPow10[X + 5]) ++X;  // 10^X <= a < 10^(X+1)
  uint32_t bits;
  std::memcpy(&bits, &v, 4);
  const int be = static_cast<int>((bits >> 23) & 0xffu);
  const uint64_t m = (bits & 0x7fffffu) | (be ? 0x800000u : 0u);
  const int e = (be ? be : 1) - 150;  // |v| = m * 2^e
  const int k = 7 - X;                // 0 .. 11
  const unsigned __int128 num = static_cast<unsigned __int128>(m) * kPow5[k];
  const int sh = e + k;
  uint64_t N;
  if (sh >= 0) {
    N = static_cast<uint64_t>(num << sh);
  } else {
    const int r = -sh;  // at most 149 + ... in principle; a >= 1e-4 keeps it below 64 + 24
    if (r >= 100) return false;
    const unsigned __int128 q = num >> r, rem = num & ((static_cast<unsigned __int128>(1) << r) - 1);
    const unsigned __int128 half = static_cast<unsigned __int128>(1) << (r - 1);
    N = static_cast<uint64_t>(q);
    if (rem > half || (rem == half && (N & 1u))) ++N;  // round half to even on the exact value
  }
  if (N >= 100000000ull) {  // carried into the next decade
    N = 10000000ull;
    ++X;
    if (X >= 8) return false;  // exponent notation
  }
  char d[8];
  for (int i = 7; i >= 0; --i) {
    d[i] = static_cast<char>('0' + N % 10);
    N /= 10;
  }
  int last = 7;
  while (last > 0 && d[last] == '0') --last;  // %g strips trailing zeros
  char buf[24];
  int n = 0;
  if (bits >> 31) buf[n++] = '-';
  if (X >= 0) {
    for (int i = 0; i <= X; ++i) buf[n++] = d[i];
    if (last > X) {
      buf[n++] = '.';
      for (int i = X + 1; i <= last; ++i) buf[n++] = d[i];
    }
  } else {
    buf[n++] = '0';
    buf[n++] = '.';
    for (int i = 0; i < -X - 1; ++i) buf[n++] = '0';
    for (int i = 0; i <= last; ++i) buf[n++] = d[i];
  }
  out.append(buf, static_cast<size_t>(n));
  return true;
}
inline void put_float_printf(std::string &out, float v) {
  if (std::isnan(v)) {
    out += "nan";
    return;
  }
  char buf[40];
  std::snprintf(buf, sizeof(buf), "%.8g", static_cast<double>(v));
  out += buf;
}
inline void put_float(std::string &out, float v) {
  if (!put_float_fixed(out, v)) put_float_printf(out, v);
}
inline std::string header(const char *fields, const char *sizes, const char *types, const char *counts, size_t n) {
  std::ostringstream h;
  h << "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS " << fields << "\nSIZE " << sizes << "\nTYPE "
    << types << "\nCOUNT " << counts << "\nWIDTH " << n << "\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS " << n
    << "\nDATA ascii\n";
  return h.str();
}
// Formats rows [0, n) with `row(out, i)` on all host cores (contiguous slices, one buffer per thread) and writes
// header + slices in order: the bytes are those of a sequential writer, the wall time is not (the reference's
// PCDWriter::writeASCII formats one value at a time on one thread and dominates its runs at scale).
template <class Row>
inline int write_rows(const std::string &path, const std::string &head, size_t n, size_t bytes_per_row_hint, Row row) {
  // pcl::PCDWriter::writeASCII (PCL 1.10 pcd_io.hpp) throws pcl::IOException on an empty cloud before touching the
  // file; PCLException's what() is ": <description>" when no file / function / line is attached.  The reference's
  // main catches it and exits with -2 (main.cpp:64-68) -- e.g. a keyframe whose mask image is missing.
  if (n == 0) throw std::runtime_error(": [pcl::PCDWriter::writeASCII] Input point cloud has no data!");
  unsigned threads = std::thread::hardware_concurrency();
  if (const char *e = std::getenv("PCP_WRITER_THREADS")) threads = static_cast<unsigned>(std::max(1, std::atoi(e)));
  threads = std::max(1u, std::min(threads, 32u));
  if (n < 20000) threads = 1;
  std::vector<std::string> part(threads);
  auto work = [&](unsigned t) {
    const size_t b = n * t / threads, e = n * (t + 1) / threads;
    std::string &s = part[t];
    s.reserve((e - b) * bytes_per_row_hint);
    for (size_t i = b; i < e; ++i) row(s, i);
  };
  std::vector<std::thread> pool;
  for (unsigned t = 1; t < threads; ++t) pool.emplace_back(work, t);
  work(0);
  for (auto &th : pool) th.join();
  std::ofstream out(path, std::ios::binary);
  if (!out) return -1;
  out.write(head.data(), static_cast<std::streamsize>(head.size()));
  for (const auto &s : part) out.write(s.data(), static_cast<std::streamsize>(s.size()));
  return out ? 0 : -1;
}
inline void put_uint(std::string &s, uint32_t v) {  // "%u"
  char buf[12];
  int n = 12;
  do {
    buf[--n] = static_cast<char>('0' + v % 10u);
    v /= 10u;
  } while (v);
  s.append(buf + n, static_cast<size_t>(12 - n));
}
inline void put_rgb(std::string &s, const uint8_t *rgb) {  // packed 0xAARRGGBB with A = 255 (PointXYZRGB ctor)
  put_uint(s, 0xff000000u | (static_cast<uint32_t>(rgb[0]) << 16) | (static_cast<uint32_t>(rgb[1]) << 8) | rgb[2]);
}
}  // namespace detail

// x y z intensity (pcl::PointXYZI)
inline int writeASCII_XYZI(const std::string &path, const float *x, const float *y, const float *z, const float *intensity,
                           size_t n) {
  return detail::write_rows(path, detail::header("x y z intensity", "4 4 4 4", "F F F F", "1 1 1 1", n), n, 48,
                            [=](std::string &s, size_t i) {
                              detail::put_float(s, x[i]);
                              s += ' ';
                              detail::put_float(s, y[i]);
                              s += ' ';
                              detail::put_float(s, z[i]);
                              s += ' ';
                              detail::put_float(s, intensity ? intensity[i] : 0.0f);
                              s += '\n';
                            });
}

// x y z rgb (pcl::PointXYZRGB): rgb as packed uint32
inline int writeASCII_XYZRGB(const std::string &path, const float *x, const float *y, const float *z, const uint8_t *rgb,
                             size_t n) {
  return detail::write_rows(path, detail::header("x y z rgb", "4 4 4 4", "F F F U", "1 1 1 1", n), n, 48,
                            [=](std::string &s, size_t i) {
                              detail::put_float(s, x[i]);
                              s += ' ';
                              detail::put_float(s, y[i]);
                              s += ' ';
                              detail::put_float(s, z[i]);
                              s += ' ';
                              detail::put_rgb(s, rgb + 3 * i);
                              s += '\n';
                            });
}

// x y z rgb segmentMask (PointXYZRGBMask, PCP/include/FrameData.hpp:68-87)
inline int writeASCII_XYZRGBMask(const std::string &path, const float *xyz /* 3 per point */, const uint8_t *rgb,
                                 const uint16_t *mask, size_t n) {
  return detail::write_rows(path, detail::header("x y z rgb segmentMask", "4 4 4 4 2", "F F F U U", "1 1 1 1 1", n), n, 56,
                            [=](std::string &s, size_t i) {
                              for (int c = 0; c < 3; ++c) {
                                detail::put_float(s, xyz[3 * i + static_cast<size_t>(c)]);
                                s += ' ';
                              }
                              detail::put_rgb(s, rgb + 3 * i);
                              s += ' ';
                              detail::put_uint(s, mask[i]);
                              s += '\n';
                            });
}

// x y z normal_x normal_y normal_z curvature (pcl::PointNormal), savePCDFile default = ASCII (cloudSmooth.cpp:181)
inline int writeASCII_PointNormal(const std::string &path, const float *xyz, const float *normal, const float *curv,
                                  size_t n) {
  return detail::write_rows(path,
                            detail::header("x y z normal_x normal_y normal_z curvature", "4 4 4 4 4 4 4", "F F F F F F F",
                                           "1 1 1 1 1 1 1", n),
                            n, 96, [=](std::string &s, size_t i) {
                              for (int c = 0; c < 3; ++c) {
                                detail::put_float(s, xyz[3 * i + static_cast<size_t>(c)]);
                                s += ' ';
                              }
                              for (int c = 0; c < 3; ++c) {
                                detail::put_float(s, normal[3 * i + static_cast<size_t>(c)]);
                                s += ' ';
                              }
                              detail::put_float(s, curv[i]);
                              s += '\n';
                            });
}

}  // namespace pcp_amd
