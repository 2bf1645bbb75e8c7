// pcp_shim.hpp -- C++ host shim above the C ABI (include/pcp_hip.h).
//
// Mirrors the reference's operator interface for the hot path so that the
// existing C++ binary keeps its call structure:
//
//   pcp_amd::ViewCulling::cull        <- vlcal::ViewCulling::cull           (view_culling.hpp:34)
//   pcp_amd::Colorizer::colorize      <- pcdColorizationAndSmooth inner loop (PointCloudProcessor.cpp:488-596)
//   pcp_amd::Colorizer::frameVisible  <- generateColorMap + generateSegmentMap (:531-551)
//   pcp_amd::CloudSmooth::process     <- CloudSmooth::process               (cloudSmooth.cpp:77-185)
//
// Error convention: the ABI never throws; this shim rethrows std::runtime_error
// so that main.cpp:64-68 still maps failures to exit code -2.
// Header-only, depends on the C ABI alone (no PCL / Eigen / OpenCV types): the
// caller passes raw pointers taken from its own containers.
#pragma once

#include <algorithm>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "pcp_hip.h"

namespace pcp_amd {

class Device {
 public:
  explicit Device(int ordinal = 0) {
    if (pcp_create(ordinal, &ctx_) != PCP_OK) throw std::runtime_error(std::string("pcp_hip: ") + pcp_last_error(nullptr));
  }
  ~Device() { pcp_destroy(ctx_); }
  Device(const Device &) = delete;
  Device &operator=(const Device &) = delete;
  pcp_context *get() const { return ctx_; }
  void check(int rc) const {
    if (rc != PCP_OK) throw std::runtime_error(std::string("pcp_hip: ") + pcp_last_error(ctx_));
  }

  // `cloud` of the reference: pcl::PointCloud<pcl::PointXYZI>::points.data(), stride sizeof(pcl::PointXYZI) == 32
  void uploadCloudAoS(const void *points, int64_t n, int64_t stride_bytes) { check(pcp_upload_cloud_aos(ctx_, points, n, stride_bytes)); }
  void uploadCloud(const float *x, const float *y, const float *z, int64_t n) { check(pcp_upload_cloud(ctx_, x, y, z, n)); }
  // pcl::PointXYZI::intensity of every point (the NID stage bins it, nid_cost.hpp:55-56)
  void uploadIntensity(const float *intensity, int64_t n) { check(pcp_upload_intensity(ctx_, intensity, n)); }
  int64_t cloudSize() const { return pcp_cloud_size(ctx_); }

  // K_camera_coefficients / D_camera (PointCloudProcessor.cpp:57-62) + ViewCulling image size (:525)
  void setCamera(const pcp_camera &cam, const pcp_cull_params *cull = nullptr) { check(pcp_set_camera(ctx_, &cam, cull)); }
  // selectedKeyframes' poses; T_camera_lidar_optimized when NID / manual guess ran (:504-519)
  void setKeyframes(const std::vector<pcp_pose> &poses, const double *T_opt = nullptr, int T_opt_stride = 0) {
    check(pcp_set_frames(ctx_, poses.data(), static_cast<int32_t>(poses.size()), T_opt, T_opt_stride));
  }
  // generateColorMap's cvtColor(BGR2HSV) -> S, V scaling -> cvtColor(HSV2BGR) (PointCloudProcessor.cpp:722-741) done
  // by the library while it packs an uploaded image: enable it and hand over cv::imread's pixels, or leave it off
  // and hand over `adjusted_image`
  void setImageAdjust(bool enable, float saturation_scale = 1.0f, float brightness_scale = 1.0f) {
    check(pcp_set_image_adjust(ctx_, enable ? 1 : 0, saturation_scale, brightness_scale));
  }
  // cv::Mat rgb (CV_8UC3; after the HSV round trip unless setImageAdjust(true)): data, step
  void uploadImage(int keyframe, const uint8_t *bgr, int64_t step) { check(pcp_upload_image(ctx_, keyframe, bgr, step)); }
  // decoded frames in pinned memory: queues the copy only; the buffer must outlive the next synchronising call
  void uploadImageAsync(int keyframe, const uint8_t *bgr, int64_t step) { check(pcp_upload_image_async(ctx_, keyframe, bgr, step)); }
  // cv::Mat grayImg (CV_8UC1)
  void uploadMask(int keyframe, const uint8_t *gray, int64_t step) { check(pcp_upload_mask(ctx_, keyframe, gray, step)); }

 private:
  pcp_context *ctx_ = nullptr;
};

// vlcal::ViewCulling with the z-buffer routine (view_culling.cpp:52-174)
class ViewCulling {
 public:
  explicit ViewCulling(Device &dev) : dev_(dev) {}
  // indices of the kept points in input order (what `sample(points, point_indices)` consumes)
  std::vector<int32_t> cull(int keyframe) const {
    std::vector<uint8_t> keep(static_cast<size_t>(dev_.cloudSize()));
    int64_t kept = 0;
    dev_.check(pcp_cull_frame(dev_.get(), keyframe, keep.data(), &kept, nullptr));
    std::vector<int32_t> idx;
    idx.reserve(static_cast<size_t>(kept));
    for (size_t i = 0; i < keep.size(); ++i)
      if (keep[i]) idx.push_back(static_cast<int32_t>(i));
    return idx;
  }

 private:
  Device &dev_;
};

struct VisiblePoints {  // one keyframe's coloredCloud / scanInBodyWithRGBandMask
  std::vector<int32_t> index;
  std::vector<uint8_t> rgb;      // 3 per point
  std::vector<uint16_t> mask;    // segmentMask
  std::vector<float> xyz_cam;    // 3 per point, camera frame
  std::vector<float> xyz_world;  // 3 per point, transformPointCloud(c2w)
};

class Colorizer {
 public:
  explicit Colorizer(Device &dev) : dev_(dev) {}
  // rgbCloud.cloudWithSmoothedColor after smoothColors: rgb (3 per input point) and the
  // removePointsWithNoColor keep flag
  void colorize(std::vector<uint8_t> &rgb, std::vector<uint8_t> &has) const {
    const size_t n = static_cast<size_t>(dev_.cloudSize());
    rgb.resize(3 * n);
    has.resize(n);
    dev_.check(pcp_colorize(dev_.get(), rgb.data(), has.data()));
  }
  // One cull per keyframe as a rule.  The outputs are sized by the largest keyframe seen so far (a keyframe sees a few
  // per cent of the map; sizing them for the whole cloud meant 33 B x n of zero-filled vectors per call, 1.65 GB at
  // 50 M points -- ADVICE r2); a keyframe that needs more reports its count and is fetched again with room for it.
  VisiblePoints frameVisible(int keyframe) const {
    const size_t n = static_cast<size_t>(dev_.cloudSize());
    VisiblePoints v;
    size_t cap = std::min(n, std::max<size_t>(capacity_, size_t(1) << 16));
    for (;;) {
      v.index.resize(cap);
      v.rgb.resize(3 * cap);
      v.mask.resize(cap);
      v.xyz_cam.resize(3 * cap);
      v.xyz_world.resize(3 * cap);
      int64_t m = 0;
      dev_.check(pcp_frame_visible(dev_.get(), keyframe, static_cast<int64_t>(cap), v.index.data(), v.rgb.data(), v.mask.data(),
                                   v.xyz_cam.data(), v.xyz_world.data(), &m));
      const size_t sm = static_cast<size_t>(m);
      if (sm > cap) {  // the library wrote the first `cap` records and reported the true count
        cap = std::min(n, sm + sm / 8);
        continue;
      }
      capacity_ = std::max(capacity_, sm + sm / 8);
      v.index.resize(sm);
      v.rgb.resize(3 * sm);
      v.mask.resize(sm);
      v.xyz_cam.resize(3 * sm);
      v.xyz_world.resize(3 * sm);
      return v;
    }
  }

 private:
  Device &dev_;
  mutable size_t capacity_ = 0;  // records of the largest keyframe so far (+ 1/8)
};

// vlcal::VisualLiDARCalibration (PCP/src/calibrate.cpp:42-126): NID-based refinement of
// T_camera_lidar on the z-buffer-culled keyframe clouds
class VisualLiDARCalibration {
 public:
  explicit VisualLiDARCalibration(Device &dev) : dev_(dev) {}
  // returns T_camera_lidar_optimized as a 4x4 row-major matrix (identity initial guess, 16 bins,
  // <= 10 outer iterations: calibrate.cpp:45-51, visual_camera_calibration.hpp:17,28)
  std::vector<double> calibrate(double *final_cost = nullptr) const {
    int64_t pts = 0;
    dev_.check(pcp_nid_prepare(dev_.get(), &pts));
    const double I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    std::vector<double> T(16);
    int32_t evals = 0;
    dev_.check(pcp_nid_optimize(dev_.get(), I, 16, 10, T.data(), final_cost, &evals));
    return T;
  }

 private:
  Device &dev_;
};

struct SmoothedCloud {  // pcl::PointCloud<pcl::PointNormal> mls_points + corresponding_input_indices
  std::vector<float> xyz, normal, curvature;
  std::vector<int32_t> index;
};

class CloudSmooth {
 public:
  explicit CloudSmooth(Device &dev) : dev_(dev) { pcp_default_mls_params(&params_); }
  CloudSmooth(Device &dev, const pcp_mls_params &p) : dev_(dev), params_(p) {}
  void initialize(const pcp_mls_params &p) { params_ = p; }  // CloudSmooth::initialize(MLSParameters)
  // the whole CloudSmooth::process: SOR -> MLS (+ upsampling) -> SOR (cloudSmooth.cpp:109-164)
  SmoothedCloud processWithOutlierRemoval() const { return run(true); }
  // pcl::MovingLeastSquares::process alone
  SmoothedCloud process() const { return run(false); }
  // The whole CloudSmooth::process through the streamed form (pcp_cloud_smooth_stream_begin / _next): any size of upsampled
  // cloud -- the reference's VOXEL_GRID_DILATION 1 mm x 4 (PointCloudProcessor.cpp:78-81) makes ~2.8e9 rows of a 10 M-point
  // map, more than one result holds.  sink(const SmoothedCloud &chunk) receives the survivors of the trailing outlier
  // removal chunk by chunk, in the order the one-shot form returns them; total_rows (nullable): rows before that filter.
  // Returns the rows kept.
  // hold_device_memory: leave the stream's distances on the device (4 B per row: 11 GB for a 10 M-point map) for the next call
  // instead of freeing them (pcp_cloud_smooth_stream_end).
  template <class Sink>
  int64_t processWithOutlierRemovalStreamed(int64_t chunk_capacity, Sink &&sink, int64_t *total_rows = nullptr,
                                            bool hold_device_memory = false) const {
    int64_t total = 0, kept = 0;
    int32_t chunks = 0;
    dev_.check(pcp_cloud_smooth_stream_begin(dev_.get(), &params_, chunk_capacity, &total, &kept, &chunks));
    if (total_rows) *total_rows = total;
    SmoothedCloud s;
    for (;;) {
      int64_t m = 0;
      dev_.check(pcp_cloud_smooth_stream_next(dev_.get(), &m));
      if (m == 0) break;
      fetch(m, s);
      sink(static_cast<const SmoothedCloud &>(s));
    }
    if (!hold_device_memory) dev_.check(pcp_cloud_smooth_stream_end(dev_.get()));
    return kept;
  }

 private:
  void fetch(int64_t m, SmoothedCloud &s) const {
    const size_t sm = static_cast<size_t>(m);
    s.xyz.resize(3 * sm);
    s.normal.resize(3 * sm);
    s.curvature.resize(sm);
    s.index.resize(sm);
    dev_.check(pcp_mls_fetch(dev_.get(), m, s.xyz.data(), s.normal.data(), s.curvature.data(), s.index.data()));
  }
  SmoothedCloud run(bool with_sor) const {
    int64_t m = 0;
    int rc = with_sor ? pcp_cloud_smooth(dev_.get(), &params_, &m) : pcp_mls_process(dev_.get(), &params_, &m);
    if (with_sor && rc == PCP_ERR_NOMEM && params_.upsampling == 3) {
      // more upsampled points than one result holds: the streamed chain, gathered on the host (the caller asked for one cloud)
      SmoothedCloud all;
      processWithOutlierRemovalStreamed(int64_t(1) << 28, [&](const SmoothedCloud &c) {
        all.xyz.insert(all.xyz.end(), c.xyz.begin(), c.xyz.end());
        all.normal.insert(all.normal.end(), c.normal.begin(), c.normal.end());
        all.curvature.insert(all.curvature.end(), c.curvature.begin(), c.curvature.end());
        all.index.insert(all.index.end(), c.index.begin(), c.index.end());
      });
      return all;
    }
    dev_.check(rc);
    SmoothedCloud s;
    const size_t sm = static_cast<size_t>(m);
    s.xyz.resize(3 * sm);
    s.normal.resize(3 * sm);
    s.curvature.resize(sm);
    s.index.resize(sm);
    dev_.check(pcp_mls_fetch(dev_.get(), m, s.xyz.data(), s.normal.data(), s.curvature.data(), s.index.data()));
    return s;
  }

  Device &dev_;
  pcp_mls_params params_;
};

}  // namespace pcp_amd
