// shim_selftest.cpp -- exercises the C++ shim end to end on a tiny hand-made scene and
// prints results as text; tests/test_host_shim.py compares them with the oracle on a
// GPU box and checks the exit-code convention without one.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>

#include "pcp_shim.hpp"

int main(int argc, char **argv) {
  try {
    const int n = argc > 1 ? std::atoi(argv[1]) : 2000;
    pcp_amd::Device dev(0);
    pcp_camera cam;
    pcp_default_camera(&cam);
    cam.fx = cam.fy = 188.2083;
    cam.cx = 80.0;
    cam.cy = 45.0;
    cam.image_width = cam.cull_width = 160;
    cam.image_height = cam.cull_height = 90;
    dev.setCamera(cam);
    // AoS points as pcl::PointXYZI lays them out (x y z pad | intensity pad pad pad), 32 B
    struct P {
      float x, y, z, pad, intensity, p1, p2, p3;
    };
    std::vector<P> pts(static_cast<size_t>(n));
    uint32_t s = 12345u;
    auto rnd = [&]() {
      s = s * 1664525u + 1013904223u;
      return static_cast<float>(s >> 8) / 16777216.0f;
    };
    for (auto &p : pts) {  // a wall at z = 3 and a nearer occluding strip at z = 1.5
      const bool strip = rnd() < 0.3f;
      p.x = (rnd() - 0.5f) * (strip ? 0.4f : 3.0f);
      p.y = (rnd() - 0.5f) * 1.6f;
      p.z = strip ? 1.5f : 3.0f;
      p.intensity = rnd();
    }
    dev.uploadCloudAoS(pts.data(), n, sizeof(P));
    std::vector<pcp_pose> poses = {{0, 0, 0, 1, 0, 0, 0}, {0.2, 0, 0, 1, 0, 0, 0}};
    dev.setKeyframes(poses);
    std::vector<uint8_t> img(160 * 90 * 3);
    for (int f = 0; f < 2; ++f) {
      for (size_t i = 0; i < img.size(); ++i) img[i] = static_cast<uint8_t>((i * 7 + f * 31) % 251 + 1);
      dev.uploadImage(f, img.data(), 160 * 3);
    }
    pcp_amd::ViewCulling vc(dev);
    const auto kept = vc.cull(0);
    pcp_amd::Colorizer col(dev);
    std::vector<uint8_t> rgb, has;
    col.colorize(rgb, has);
    const auto vis = col.frameVisible(1);
    long coloured = 0, checksum = 0;
    for (size_t i = 0; i < has.size(); ++i) {
      coloured += has[i];
      checksum += rgb[3 * i] + 3 * rgb[3 * i + 1] + 7 * rgb[3 * i + 2];
    }
    std::printf("kept0 %zu coloured %ld checksum %ld visible1 %zu\n", kept.size(), coloured, checksum, vis.index.size());
    if (argc > 2) {
      // CloudSmooth::process, one-shot and streamed (the last two stages in chunks of at most argv[2] voxels): the same rows
      pcp_mls_params mp;
      pcp_default_mls_params(&mp);
      mp.vgd_voxel_size = 0.02f;
      mp.vgd_iterations = 1;
      mp.search_radius = 0.25;
      mp.sqr_gauss_param = 0.0625;
      mp.sor_mean_k = 12;
      pcp_amd::CloudSmooth sm(dev, mp);
      const pcp_amd::SmoothedCloud one = sm.processWithOutlierRemoval();
      size_t at = 0;
      bool same = true;
      int64_t total = 0;
      int chunks = 0;
      const int64_t kept_rows = sm.processWithOutlierRemovalStreamed(std::atoll(argv[2]), [&](const pcp_amd::SmoothedCloud &c) {
        ++chunks;
        for (size_t i = 0; i < c.index.size(); ++i, ++at)
          same = same && at < one.index.size() && c.index[i] == one.index[at] && c.xyz[3 * i] == one.xyz[3 * at] &&
                 c.xyz[3 * i + 1] == one.xyz[3 * at + 1] && c.xyz[3 * i + 2] == one.xyz[3 * at + 2] && c.curvature[i] == one.curvature[at];
      }, &total);
      std::printf("smooth rows %zu streamed %lld of %lld in %d chunks same %d\n", one.index.size(), (long long)kept_rows, (long long)total,
                  chunks, same && at == one.index.size() ? 1 : 0);
    }
    std::cout << "Processing completed successfully." << std::endl;
  } catch (const std::exception &e) {
    // same convention as PCP/src/main.cpp:64-68
    std::cerr << "Unhandled Exception reached the top of main: " << e.what() << ", application will now exit" << std::endl;
    return -2;
  }
  return 0;
}
