// image_dump.cpp -- decodes an image with host/image_io.hpp and writes "w h c\n" + raw bytes
// (BGR or gray) to the output path; used by tests/test_image_io.py to compare with Pillow.
#include <cstdio>
#include <cstring>

#include "image_io.hpp"

int main(int argc, char **argv) {
  if (argc < 3) {
    std::fprintf(stderr, "usage: image_dump <in> <out> [gray]\n");
    return 2;
  }
  const bool gray = argc > 3 && std::strcmp(argv[3], "gray") == 0;
  const pcp_amd::Image8 img = gray ? pcp_amd::read_image_gray(argv[1]) : pcp_amd::read_image_bgr(argv[1]);
  if (img.empty()) {
    std::fprintf(stderr, "decode failed: %s\n", argv[1]);
    return 1;
  }
  FILE *f = std::fopen(argv[2], "wb");
  if (!f) return 2;
  std::fprintf(f, "%d %d %d\n", img.width, img.height, img.channels);
  std::fwrite(img.data.data(), 1, img.data.size(), f);
  std::fclose(f);
  return 0;
}
