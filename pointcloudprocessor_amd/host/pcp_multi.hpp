// pcp_multi.hpp -- the N GPUs of one node behind the operator names of pcp_shim.hpp (C++ host, one process).
//
// BASELINE.json north_star / SURVEY.md 8(e): the map is sharded by contiguous point-index range, keyframes and
// images are replicated, per-point results (top-5 lists, colours) never leave their GPU, and the ONE exchange on the
// data path is an all-reduce(MIN) of the per-keyframe depth maps -- here a grouped ncclAllReduce(ncclFloat, ncclMin)
// over xGMI on the device pointers pcp_depth_maps_device() hands out (ranges are positive finite floats, so the float
// MIN equals the uint-bits MIN the kernels used).  Keyframe images cross PCIe ONCE (to GPU 0) and reach the other
// GPUs by ncclBroadcast; every GPU packs them from its own device buffer.
//
// Everything the reference's per-keyframe stages need follows from the merged maps: with PCP_DEPTH_BATCHED the
// single-keyframe calls of every shard use them, so cull() / frameVisible() / cameraCoordinates() concatenate the
// shards' outputs (contiguous index ranges: input order is preserved) into exactly the single-GPU result.
//
// With one GPU nothing of RCCL is touched: the calls go straight to the one Device.
//
// PCP_MULTI_REHEARSAL=1 (tests on a one-GPU box; not a measurement): the N shards are N contexts on GPU 0 and the
// two collectives are emulated through the host (MIN of the downloaded maps written back to every shard; device
// copies for the broadcast).  Everything else -- sharding, PCP_DEPTH_BATCHED, the stitching of the outputs -- is the
// code that runs on N GPUs.
#pragma once

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "pcp_shim.hpp"

namespace pcp_amd {

class MultiDevice {
 public:
  explicit MultiDevice(int n_gpus) {
    if (n_gpus < 1) throw std::runtime_error("pcp_multi: --gpus must be >= 1");
    if (const char *e = std::getenv("PCP_MULTI_REHEARSAL")) rehearsal_ = e[0] == '1' && n_gpus > 1;
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || have < (rehearsal_ ? 1 : n_gpus))
      throw std::runtime_error("pcp_multi: " + std::to_string(n_gpus) + " GPUs requested, " + std::to_string(have) +
                               " visible; libpcp_hip has no CPU fallback");
    for (int i = 0; i < n_gpus; ++i) dev_.emplace_back(new Device(ordinal(i)));
    if (n_gpus == 1) return;
    stream_.resize(static_cast<size_t>(n_gpus));
    stage_.assign(static_cast<size_t>(n_gpus) * kStageSlots, nullptr);
    std::vector<int> ids(static_cast<size_t>(n_gpus));
    for (int i = 0; i < n_gpus; ++i) {
      ids[static_cast<size_t>(i)] = i;
      hip(hipSetDevice(ordinal(i)), "hipSetDevice");
      hip(hipStreamCreateWithFlags(&stream_[static_cast<size_t>(i)], hipStreamNonBlocking), "hipStreamCreate");
      // the library's kernels and RCCL share one stream per GPU: ordered without host synchronisation
      dev_[static_cast<size_t>(i)]->check(pcp_set_stream(dev_[static_cast<size_t>(i)]->get(), stream_[static_cast<size_t>(i)]));
      dev_[static_cast<size_t>(i)]->check(pcp_set_depth_source(dev_[static_cast<size_t>(i)]->get(), PCP_DEPTH_BATCHED));
    }
    if (rehearsal_) return;
    comm_.resize(static_cast<size_t>(n_gpus));
    nccl(ncclCommInitAll(comm_.data(), n_gpus, ids.data()), "ncclCommInitAll");
  }
  ~MultiDevice() {
    for (size_t i = 0; i < comm_.size(); ++i) (void)ncclCommDestroy(comm_[i]);
    for (size_t i = 0; i < stage_.size(); ++i)
      if (stage_[i]) {
        (void)hipSetDevice(ordinal(static_cast<int>(i / kStageSlots)));
        (void)hipFree(stage_[i]);
      }
    dev_.clear();  // the contexts go before their streams
    for (size_t i = 0; i < stream_.size(); ++i) {
      (void)hipSetDevice(ordinal(static_cast<int>(i)));
      (void)hipStreamDestroy(stream_[i]);
    }
  }
  MultiDevice(const MultiDevice &) = delete;
  MultiDevice &operator=(const MultiDevice &) = delete;

  int size() const { return static_cast<int>(dev_.size()); }
  Device &device(int i) { return *dev_[static_cast<size_t>(i)]; }
  int64_t cloudSize() const { return n_; }
  // contiguous index range of shard r: the first n % N shards hold one point more (= pipeline.shard_bounds)
  int64_t shardBegin(int r) const {
    const int64_t N = size(), base = n_ / N, rem = n_ % N;
    return r * base + std::min<int64_t>(r, rem);
  }

  // The arrays must stay valid until the next uploadCloud (hidden_points_removal over several GPUs uploads the WHOLE map
  // to every GPU once the cull mode is known, see setCamera).
  void uploadCloud(const float *x, const float *y, const float *z, int64_t n) {
    n_ = n;
    hx_ = x;
    hy_ = y;
    hz_ = z;
    for (int r = 0; r < size(); ++r) {
      const int64_t lo = shardBegin(r), hi = shardBegin(r + 1);
      device(r).uploadCloud(x + lo, y + lo, z + lo, hi - lo);
    }
    for (auto &h : hull_) h->uploadCloud(x, y, z, n);
    depth_ready_ = false;
  }
  // PCP_CULL_HPR with N > 1: a keyframe's hull is taken over EVERY candidate of the map (view_culling.cpp:291-329), so an
  // index shard cannot decide its own points.  Every GPU gets a second context that holds the whole map (no images);
  // keyframe f's hull is taken by GPU f mod N there, and its verdicts (one flag per map point; slices exchanged on the devices,
  // hullPassAll) are handed to the shards (pcp_hull_flags_import), which colour / dump their points from them exactly as
  // they do from the merged depth maps of the z-buffer routine.
  void setCamera(const pcp_camera &cam, const pcp_cull_params *cull = nullptr) {
    cam_ = cam;
    for (auto &d : dev_) d->setCamera(cam, cull);
    hpr_ = cull && cull->cull_mode == PCP_CULL_HPR && size() > 1;
    if (hpr_) {
      if (hull_.empty()) {
        for (int r = 0; r < size(); ++r) {
          hull_.emplace_back(new Device(ordinal(r)));
          if (hx_) hull_.back()->uploadCloud(hx_, hy_, hz_, n_);
        }
      }
      for (auto &h : hull_) h->setCamera(cam, cull);
    } else {
      hull_.clear();
    }
    depth_ready_ = false;
  }
  void setKeyframes(const std::vector<pcp_pose> &poses, const double *T_opt = nullptr, int T_opt_stride = 0) {
    n_frames_ = static_cast<int>(poses.size());
    for (auto &d : dev_) d->setKeyframes(poses, T_opt, T_opt_stride);
    for (auto &h : hull_) h->setKeyframes(poses, T_opt, T_opt_stride);
    depth_ready_ = false;
  }
  void setImageAdjust(bool enable, float saturation_scale = 1.0f, float brightness_scale = 1.0f) {
    for (auto &d : dev_) d->setImageAdjust(enable, saturation_scale, brightness_scale);
  }
  // cv::Mat rgb / grayImg of one keyframe: over PCIe once, to the other GPUs over xGMI
  void uploadImage(int keyframe, const uint8_t *bgr, int64_t step) { replicate(keyframe, bgr, step, cam_.image_height, false); }
  void uploadMask(int keyframe, const uint8_t *gray, int64_t step) { replicate(keyframe, gray, step, cam_.image_height, true); }

  // z-buffer MIN pass of every shard over all keyframes + the all-reduce(MIN) across the shards
  void depthPassAll() {
    if (size() == 1) return;  // pcp_colorize / pcp_cull_frame build their own maps
    for (int r = 0; r < size(); ++r) device(r).check(pcp_depth_pass(device(r).get(), 0, n_frames_));
    if (hpr_) {  // no depth maps in this mode: the hulls of the whole map, keyframes dealt out round-robin
      hullPassAll();
      depth_ready_ = true;
      return;
    }
    if (rehearsal_) {
      std::vector<uint32_t> merged, part;
      std::vector<void *> ptr(static_cast<size_t>(size()));
      int64_t count = 0;
      for (int r = 0; r < size(); ++r) {
        device(r).check(pcp_depth_maps_device(device(r).get(), &ptr[static_cast<size_t>(r)], &count));
        device(r).check(pcp_synchronize(device(r).get()));
        part.resize(static_cast<size_t>(count));
        hip(hipMemcpy(part.data(), ptr[static_cast<size_t>(r)], part.size() * 4, hipMemcpyDeviceToHost), "hipMemcpy(depth maps)");
        if (r == 0)
          merged = part;
        else
          for (size_t k = 0; k < merged.size(); ++k) merged[k] = std::min(merged[k], part[k]);  // positive floats: uint order
      }
      for (int r = 0; r < size(); ++r)
        hip(hipMemcpy(ptr[static_cast<size_t>(r)], merged.data(), merged.size() * 4, hipMemcpyHostToDevice), "hipMemcpy(depth maps)");
      depth_ready_ = true;
      return;
    }
    nccl(ncclGroupStart(), "ncclGroupStart");
    for (int r = 0; r < size(); ++r) {
      void *p = nullptr;
      int64_t count = 0;
      device(r).check(pcp_depth_maps_device(device(r).get(), &p, &count));
      nccl(ncclAllReduce(p, p, static_cast<size_t>(count), ncclFloat, ncclMin, comm_[static_cast<size_t>(r)],
                         stream_[static_cast<size_t>(r)]),
           "ncclAllReduce(depth maps, MIN)");
    }
    nccl(ncclGroupEnd(), "ncclGroupEnd");
    depth_ready_ = true;
  }

  // pcl::PointXYZI::intensity of every point, sharded like the cloud
  void uploadIntensity(const float *intensity, int64_t n) {
    if (n != n_) throw std::runtime_error("pcp_multi: intensity for " + std::to_string(n) + " points, cloud has " + std::to_string(n_));
    for (int r = 0; r < size(); ++r) device(r).uploadIntensity(intensity + shardBegin(r), shardBegin(r + 1) - shardBegin(r));
  }

  // VisualLiDARCalibration::calibrate (calibrate.cpp:42-126) over the shards: every GPU accumulates the keyframes' joint
  // histograms of its points, one all-reduce(SUM) per cost evaluation adds them (8 B x F x 1808 at 16 bins), GPU 0's
  // context turns the sums into cost and gradient for the BFGS loop.  Identity initial guess, 16 bins, <= 10 outer
  // iterations as the one-GPU shim.
  std::vector<double> calibrate(double *final_cost = nullptr) {
    if (size() == 1) return VisualLiDARCalibration(device(0)).calibrate(final_cost);
    if (!depth_ready_) depthPassAll();  // the shards' culls read the merged maps
    for (int r = 0; r < size(); ++r) {
      int64_t pts = 0;
      device(r).check(pcp_nid_prepare(device(r).get(), &pts));
    }
    const double I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    std::vector<double> T(16);
    int32_t evals = 0;
    nid_error_.clear();
    const int rc = pcp_nid_optimize_with(device(0).get(), &MultiDevice::nidEvaluate, this, I, 16, 10, T.data(), final_cost, &evals);
    if (!nid_error_.empty()) throw std::runtime_error(nid_error_);
    device(0).check(rc);
    return T;
  }

  // pcdColorizationAndSmooth: rgb (3 per input point) and the removePointsWithNoColor flag, input order
  void colorize(std::vector<uint8_t> &rgb, std::vector<uint8_t> &has) {
    rgb.resize(3 * static_cast<size_t>(n_));
    has.resize(static_cast<size_t>(n_));
    if (size() == 1) {
      device(0).check(pcp_colorize(device(0).get(), rgb.data(), has.data()));
      return;
    }
    if (!depth_ready_) depthPassAll();
    for (int r = 0; r < size(); ++r)  // queued on every GPU before any host wait
      device(r).check(pcp_colorize_from_depth(device(r).get(), nullptr, nullptr));
    // every shard's packed colours start for the host at once (each GPU's copy stream), then the host waits shard by shard
    std::vector<uint32_t> all(static_cast<size_t>(n_));
    for (int r = 0; r < size(); ++r)
      device(r).check(pcp_download_result_packed_async(device(r).get(), all.data() + shardBegin(r)));
    for (int r = 0; r < size(); ++r) {
      const int64_t lo = shardBegin(r), m = shardBegin(r + 1) - lo;
      device(r).check(pcp_synchronize(device(r).get()));
      const uint32_t *packed = all.data() + lo;
      for (int64_t i = 0; i < m; ++i) {
        const uint32_t v = packed[static_cast<size_t>(i)];
        uint8_t *o = rgb.data() + 3 * (lo + i);
        o[0] = static_cast<uint8_t>(v & 0xffu);
        o[1] = static_cast<uint8_t>((v >> 8) & 0xffu);
        o[2] = static_cast<uint8_t>((v >> 16) & 0xffu);
        has[static_cast<size_t>(lo + i)] = static_cast<uint8_t>(v >> 24);
      }
    }
  }

  // ViewCulling::cull of one keyframe: kept indices into the whole cloud, input order
  std::vector<int32_t> cull(int keyframe) {
    if (size() > 1 && !depth_ready_) depthPassAll();
    std::vector<int32_t> idx;
    std::vector<uint8_t> keep;
    for (int r = 0; r < size(); ++r) {
      const int64_t lo = shardBegin(r), m = shardBegin(r + 1) - lo;
      keep.resize(static_cast<size_t>(m));
      int64_t kept = 0;
      device(r).check(pcp_cull_frame(device(r).get(), keyframe, keep.data(), &kept, nullptr));
      for (int64_t i = 0; i < m; ++i)
        if (keep[static_cast<size_t>(i)]) idx.push_back(static_cast<int32_t>(lo + i));
    }
    return idx;
  }

  // transformPointCloud(*cloud, *cloudInCameraPose, w2c) of one keyframe: SoA x[n] y[n] z[n]
  void cameraCoordinates(int keyframe, std::vector<float> &cam) {
    const size_t n = static_cast<size_t>(n_);
    cam.resize(3 * n);
    std::vector<float> part;
    for (int r = 0; r < size(); ++r) {
      const int64_t lo = shardBegin(r), m = shardBegin(r + 1) - lo;
      part.resize(3 * static_cast<size_t>(m));
      device(r).check(pcp_project_frame(device(r).get(), keyframe, nullptr, nullptr, nullptr, part.data()));
      for (int a = 0; a < 3; ++a)
        std::copy(part.begin() + a * m, part.begin() + (a + 1) * m, cam.begin() + static_cast<int64_t>(a * n) + lo);
    }
  }

  // one keyframe's coloredCloud / scanInBodyWithRGBandMask over the whole map, input order
  VisiblePoints frameVisible(int keyframe) {
    if (size() > 1 && !depth_ready_) depthPassAll();
    VisiblePoints all;
    for (int r = 0; r < size(); ++r) {
      Colorizer c(device(r));
      VisiblePoints v = c.frameVisible(keyframe);
      const int32_t lo = static_cast<int32_t>(shardBegin(r));
      for (int32_t i : v.index) all.index.push_back(lo + i);
      all.rgb.insert(all.rgb.end(), v.rgb.begin(), v.rgb.end());
      all.mask.insert(all.mask.end(), v.mask.begin(), v.mask.end());
      all.xyz_cam.insert(all.xyz_cam.end(), v.xyz_cam.begin(), v.xyz_cam.end());
      all.xyz_world.insert(all.xyz_world.end(), v.xyz_world.begin(), v.xyz_world.end());
    }
    return all;
  }

 private:
  // pcp_nid_eval_fn: the sharded MultiNIDCost at T
  static int nidEvaluate(void *user, const double T[16], int32_t bins, double *cost, double grad6[6], int32_t *valid) {
    MultiDevice &m = *static_cast<MultiDevice *>(user);
    try {
      for (int r = 0; r < m.size(); ++r) m.device(r).check(pcp_nid_accumulate(m.device(r).get(), T, bins));
      std::vector<void *> ptr(static_cast<size_t>(m.size()));
      int64_t count = 0;
      for (int r = 0; r < m.size(); ++r)
        m.device(r).check(pcp_nid_histograms_device(m.device(r).get(), &ptr[static_cast<size_t>(r)], &count));
      if (m.rehearsal_) {
        std::vector<double> sum(static_cast<size_t>(count), 0.0), part(static_cast<size_t>(count));
        for (int r = 0; r < m.size(); ++r) {
          m.device(r).check(pcp_synchronize(m.device(r).get()));
          hip(hipMemcpy(part.data(), ptr[static_cast<size_t>(r)], part.size() * 8, hipMemcpyDeviceToHost), "hipMemcpy(NID histograms)");
          for (size_t k = 0; k < sum.size(); ++k) sum[k] += part[k];
        }
        for (int r = 0; r < m.size(); ++r)
          hip(hipMemcpy(ptr[static_cast<size_t>(r)], sum.data(), sum.size() * 8, hipMemcpyHostToDevice), "hipMemcpy(NID histograms)");
      } else {
        nccl(ncclGroupStart(), "ncclGroupStart");
        for (int r = 0; r < m.size(); ++r)
          nccl(ncclAllReduce(ptr[static_cast<size_t>(r)], ptr[static_cast<size_t>(r)], static_cast<size_t>(count), ncclDouble, ncclSum,
                             m.comm_[static_cast<size_t>(r)], m.stream_[static_cast<size_t>(r)]),
               "ncclAllReduce(NID histograms, SUM)");
        nccl(ncclGroupEnd(), "ncclGroupEnd");
      }
      // the contexts' streams carry the all-reduce: pcp_nid_finish's copy is ordered behind it
      m.device(0).check(pcp_nid_finish(m.device(0).get(), bins, cost, grad6, valid));
      return PCP_OK;
    } catch (const std::exception &e) {
      m.nid_error_ = e.what();
      return PCP_ERR_STATE;
    }
  }

  // hidden_points_removal of every keyframe on the whole-map contexts: GPU o takes the keyframes f = o (mod N), batch by batch
  // (one host thread per GPU for the hulls: the call synchronises with the host).  A batch's verdicts -- one flag per MAP
  // point -- stay on the device (pcp_cull_frame writes them to a device buffer): GPU o sends every shard s the slice of its
  // index range, one grouped ncclSend / ncclRecv exchange per batch over xGMI, and the shards import their slices from
  // device memory (pcp_hull_flags_import), on the stream they arrived on.  (Until round 4 the flags went down to the host and
  // up again, n bytes per keyframe each way.)  PCP_MULTI_REHEARSAL: the same slices by device copies on the one GPU.
  void hullPassAll() {
    const int N = size();
    if (n_ == 0) return;
    // per GPU: the owner's flags of the whole map, and room for the N slices a shard receives in one batch
    std::vector<uint8_t *> flags(static_cast<size_t>(N), nullptr), slices(static_cast<size_t>(N), nullptr);
    auto release = [&]() {
      for (int r = 0; r < N; ++r) {
        (void)hipSetDevice(ordinal(r));
        if (flags[static_cast<size_t>(r)]) (void)hipFree(flags[static_cast<size_t>(r)]);
        if (slices[static_cast<size_t>(r)]) (void)hipFree(slices[static_cast<size_t>(r)]);
      }
    };
    const int64_t longest = shardBegin(1) - shardBegin(0);  // the first shards hold one point more
    try {
      for (int r = 0; r < N; ++r) {
        hip(hipSetDevice(ordinal(r)), "hipSetDevice");
        hip(hipMalloc(reinterpret_cast<void **>(&flags[static_cast<size_t>(r)]), static_cast<size_t>(n_)), "hipMalloc(hull flags)");
        hip(hipMalloc(reinterpret_cast<void **>(&slices[static_cast<size_t>(r)]), static_cast<size_t>(longest) * N + 16), "hipMalloc(hull slices)");
      }
      for (int f0 = 0; f0 < n_frames_; f0 += N) {
        const int owners = std::min(N, n_frames_ - f0);
        // the hulls of this batch, GPU o keyframe f0 + o
        std::vector<std::string> failure(static_cast<size_t>(owners));
        std::vector<std::thread> th;
        for (int o = 0; o < owners; ++o)
          th.emplace_back([&, o] {
            try {
              Device &h = *hull_[static_cast<size_t>(o)];
              h.check(pcp_cull_frame(h.get(), f0 + o, flags[static_cast<size_t>(o)], nullptr, nullptr));  // device memory; synchronised
            } catch (const std::exception &e) {
              failure[static_cast<size_t>(o)] = e.what();
            }
          });
        for (auto &t : th) t.join();
        for (const auto &f : failure)
          if (!f.empty()) throw std::runtime_error(f);
        // slice s of owner o's flags -> GPU s
        if (!rehearsal_) nccl(ncclGroupStart(), "ncclGroupStart");
        for (int o = 0; o < owners; ++o)
          for (int s = 0; s < N; ++s) {
            const int64_t lo = shardBegin(s), len = shardBegin(s + 1) - lo;
            if (len == 0) continue;
            uint8_t *dst = slices[static_cast<size_t>(s)] + static_cast<int64_t>(o) * longest;
            const uint8_t *src = flags[static_cast<size_t>(o)] + lo;
            if (rehearsal_ || o == s) {
              hip(hipSetDevice(ordinal(s)), "hipSetDevice");
              hip(hipMemcpyAsync(dst, src, static_cast<size_t>(len), hipMemcpyDeviceToDevice, stream_[static_cast<size_t>(s)]), "hipMemcpyAsync(hull flags)");
            } else {
              nccl(ncclSend(src, static_cast<size_t>(len), ncclUint8, s, comm_[static_cast<size_t>(o)], stream_[static_cast<size_t>(o)]), "ncclSend(hull flags)");
              nccl(ncclRecv(dst, static_cast<size_t>(len), ncclUint8, o, comm_[static_cast<size_t>(s)], stream_[static_cast<size_t>(s)]), "ncclRecv(hull flags)");
            }
          }
        if (!rehearsal_) nccl(ncclGroupEnd(), "ncclGroupEnd");
        // (the shards' contexts run on stream_[s]: the imports are ordered behind the arrivals)
        for (int s = 0; s < N; ++s)
          for (int o = 0; o < owners; ++o)
            device(s).check(pcp_hull_flags_import(device(s).get(), f0 + o, slices[static_cast<size_t>(s)] + static_cast<int64_t>(o) * longest));
        // the owners' buffers are rewritten by the next batch's hulls (another stream): every copy out of them must be done
        for (int r = 0; r < N; ++r) {
          hip(hipSetDevice(ordinal(r)), "hipSetDevice");
          hip(hipStreamSynchronize(stream_[static_cast<size_t>(r)]), "hipStreamSynchronize");
        }
      }
      for (int s = 0; s < N; ++s) device(s).check(pcp_synchronize(device(s).get()));
    } catch (...) {
      release();
      throw;
    }
    release();
  }

  static void hip(hipError_t e, const char *what) {
    if (e != hipSuccess) throw std::runtime_error(std::string("pcp_multi: ") + what + ": " + hipGetErrorString(e));
  }
  static void nccl(ncclResult_t r, const char *what) {
    if (r != ncclSuccess) throw std::runtime_error(std::string("pcp_multi: ") + what + ": " + ncclGetErrorString(r));
  }

  // One keyframe image (or mask) to every GPU: over PCIe once (to GPU 0), to the others over xGMI (ncclBroadcast on the
  // GPUs' streams), then packed into texels by every library from its device copy.  A ring of kStageSlots staging buffers
  // per GPU: the copy of keyframe k + 1 crosses PCIe while keyframe k is broadcast and packed, nothing waits on the host
  // until a slot comes round again (every kStageSlots keyframes: one pcp_synchronize per GPU) -- the one-buffer version
  // waited for the pack of every keyframe before the next copy could start ("What's weak" 11).
  static constexpr int kStageSlots = 4;
  void replicate(int keyframe, const uint8_t *host, int64_t step, int rows, bool mask) {
    if (size() == 1) {
      if (mask)
        device(0).uploadMask(keyframe, host, step);
      else
        device(0).uploadImage(keyframe, host, step);
      return;
    }
    const size_t bytes = static_cast<size_t>(step) * static_cast<size_t>(rows);
    if (bytes > stage_bytes_) {
      drainStages();
      for (int r = 0; r < size(); ++r) {
        hip(hipSetDevice(ordinal(r)), "hipSetDevice");
        for (int k = 0; k < kStageSlots; ++k) {
          uint8_t *&p = stage_[static_cast<size_t>(r * kStageSlots + k)];
          if (p) hip(hipFree(p), "hipFree");
          hip(hipMalloc(reinterpret_cast<void **>(&p), bytes), "hipMalloc(image staging)");
        }
      }
      stage_bytes_ = bytes;
    }
    if (stage_used_ == kStageSlots) drainStages();  // the oldest slot is about to be overwritten: its pack must be done
    const int slot = stage_next_;
    stage_next_ = (stage_next_ + 1) % kStageSlots;
    stage_used_ += 1;
    auto buf = [&](int r) { return stage_[static_cast<size_t>(r * kStageSlots + slot)]; };
    hip(hipSetDevice(ordinal(0)), "hipSetDevice");
    hip(hipMemcpyAsync(buf(0), host, bytes, hipMemcpyHostToDevice, stream_[0]), "hipMemcpyAsync(image)");
    if (rehearsal_) {
      hip(hipStreamSynchronize(stream_[0]), "hipStreamSynchronize");
      for (int r = 1; r < size(); ++r)
        hip(hipMemcpyAsync(buf(r), buf(0), bytes, hipMemcpyDeviceToDevice, stream_[static_cast<size_t>(r)]),
            "hipMemcpyAsync(image, rehearsal)");
    } else {
      nccl(ncclGroupStart(), "ncclGroupStart");
      for (int r = 0; r < size(); ++r)
        nccl(ncclBroadcast(buf(0), buf(r), bytes, ncclUint8, 0, comm_[static_cast<size_t>(r)], stream_[static_cast<size_t>(r)]),
             "ncclBroadcast(image)");
      nccl(ncclGroupEnd(), "ncclGroupEnd");
    }
    // device pointers: the library orders its pack kernel after the broadcast queued on the same stream.  Images take the
    // asynchronous form (the slot stays untouched until drainStages); the mask upload synchronises by itself.
    for (int r = 0; r < size(); ++r) {
      if (mask)
        device(r).uploadMask(keyframe, buf(r), step);
      else
        device(r).uploadImageAsync(keyframe, buf(r), step);
    }
    // the host buffer belongs to the caller again when this returns: the PCIe copy must have left it
    hip(hipStreamSynchronize(stream_[0]), "hipStreamSynchronize(image copy)");
  }
  // every queued pack has read its staging slot
  void drainStages() {
    if (stage_used_ == 0) return;
    for (int r = 0; r < size(); ++r) device(r).check(pcp_synchronize(device(r).get()));
    stage_used_ = 0;
  }

  int ordinal(int shard) const { return rehearsal_ ? 0 : shard; }

  bool rehearsal_ = false;
  bool hpr_ = false;
  const float *hx_ = nullptr, *hy_ = nullptr, *hz_ = nullptr;
  std::vector<std::unique_ptr<Device>> hull_;  // PCP_CULL_HPR, N > 1: the whole map on every GPU
  std::vector<std::unique_ptr<Device>> dev_;
  std::vector<hipStream_t> stream_;
  std::vector<ncclComm_t> comm_;
  std::vector<uint8_t *> stage_;  // [gpu][slot]
  size_t stage_bytes_ = 0;
  int stage_next_ = 0, stage_used_ = 0;
  pcp_camera cam_{};
  int64_t n_ = 0;
  int n_frames_ = 0;
  bool depth_ready_ = false;
  std::string nid_error_;
};

// CloudSmooth::process (cloudSmooth.cpp:77-185) over the GPUs of a node, SURVEY.md 8(e): the cloud on EVERY GPU, the work
// dealt out -- MLS queries by index range (pcp_mls_process_shard), the dilated voxel set by chunks of its key order
// (pcp_mls_stream_*, chunk c to GPU c mod N) -- and the pieces put together in order on the host, which needs them anyway to
// write <stem>_mls.pcd.  The two StatisticalOutlierRemoval brackets run on GPU 0: each is milliseconds of work whose
// sharding would cost an exchange of the intermediate cloud.  One GPU: pcp_cloud_smooth.
class MultiCloudSmooth {
 public:
  explicit MultiCloudSmooth(int n_gpus) {
    const char *rehearsal = std::getenv("PCP_MULTI_REHEARSAL");
    const bool one_gpu = rehearsal && rehearsal[0] == '1';
    for (int r = 0; r < std::max(n_gpus, 1); ++r) dev_.emplace_back(new Device(one_gpu ? 0 : r));
    pcp_default_mls_params(&params_);
    if (n_gpus > 1 && !one_gpu) {
      // the exchanges of the outlier removal (chunk sums, keep flags) are RCCL collectives on device memory
      std::vector<int> ids(static_cast<size_t>(n_gpus));
      stream_.resize(static_cast<size_t>(n_gpus));
      for (int r = 0; r < n_gpus; ++r) {
        ids[static_cast<size_t>(r)] = r;
        check_hip(hipSetDevice(r), "hipSetDevice");
        check_hip(hipStreamCreateWithFlags(&stream_[static_cast<size_t>(r)], hipStreamNonBlocking), "hipStreamCreate");
      }
      comm_.resize(static_cast<size_t>(n_gpus));
      check_nccl(ncclCommInitAll(comm_.data(), n_gpus, ids.data()), "ncclCommInitAll");
    }
  }
  ~MultiCloudSmooth() {
    for (size_t i = 0; i < comm_.size(); ++i) (void)ncclCommDestroy(comm_[i]);
    dev_.clear();
    for (size_t i = 0; i < stream_.size(); ++i) {
      (void)hipSetDevice(static_cast<int>(i));
      (void)hipStreamDestroy(stream_[i]);
    }
  }
  MultiCloudSmooth(const MultiCloudSmooth &) = delete;
  MultiCloudSmooth &operator=(const MultiCloudSmooth &) = delete;
  void initialize(const pcp_mls_params &p) { params_ = p; }
  int size() const { return static_cast<int>(dev_.size()); }

  SmoothedCloud processWithOutlierRemoval(const float *x, const float *y, const float *z, int64_t n) {
    if (size() == 1) {
      dev_[0]->uploadCloud(x, y, z, n);
      return CloudSmooth(*dev_[0], params_).processWithOutlierRemoval();
    }
    // SOR 1 (cloudSmooth.cpp:109-116)
    std::vector<int32_t> idx1 = outlierRemoval(x, y, z, n);
    std::vector<float> cx(idx1.size()), cy(idx1.size()), cz(idx1.size());
    for (size_t k = 0; k < idx1.size(); ++k) {
      cx[k] = x[idx1[k]];
      cy[k] = y[idx1[k]];
      cz[k] = z[idx1[k]];
    }
    // MLS (+ upsampling) on the survivors, on every GPU (:124-154)
    const int64_t m1 = static_cast<int64_t>(idx1.size());
    for (auto &d : dev_) d->uploadCloud(cx.data(), cy.data(), cz.data(), m1);
    SmoothedCloud s;
    auto append = [&](Device &d, int64_t count) {
      const size_t at = s.curvature.size(), c = static_cast<size_t>(count);
      s.xyz.resize(3 * (at + c));
      s.normal.resize(3 * (at + c));
      s.curvature.resize(at + c);
      s.index.resize(at + c);
      d.check(pcp_mls_fetch(d.get(), count, s.xyz.data() + 3 * at, s.normal.data() + 3 * at, s.curvature.data() + at,
                            s.index.data() + at));
    };
    if (params_.upsampling == 0) {
      // queries by slabs of the stage's own spatial order (1 / N of the work whatever order the points come in); every
      // slab's rows arrive in input order, a point is fitted by exactly one slab: an N-way merge by source index
      std::vector<int64_t> counts(static_cast<size_t>(size()));
      for (int r = 0; r < size(); ++r)  // queued on every GPU before the first fetch waits
        dev_[static_cast<size_t>(r)]->check(pcp_mls_process_slab(dev_[static_cast<size_t>(r)]->get(), &params_, r, size(), &counts[static_cast<size_t>(r)]));
      std::vector<SmoothedCloud> part(static_cast<size_t>(size()));
      size_t total = 0;
      for (int r = 0; r < size(); ++r) {
        SmoothedCloud &q = part[static_cast<size_t>(r)];
        const size_t c = static_cast<size_t>(counts[static_cast<size_t>(r)]);
        q.xyz.resize(3 * c);
        q.normal.resize(3 * c);
        q.curvature.resize(c);
        q.index.resize(c);
        dev_[static_cast<size_t>(r)]->check(pcp_mls_fetch(dev_[static_cast<size_t>(r)]->get(), counts[static_cast<size_t>(r)], q.xyz.data(),
                                                          q.normal.data(), q.curvature.data(), q.index.data()));
        total += c;
      }
      s.xyz.resize(3 * total);
      s.normal.resize(3 * total);
      s.curvature.resize(total);
      s.index.resize(total);
      std::vector<size_t> head(static_cast<size_t>(size()), 0);
      for (size_t k = 0; k < total; ++k) {
        int best = -1;
        for (int r = 0; r < size(); ++r) {
          const SmoothedCloud &q = part[static_cast<size_t>(r)];
          if (head[static_cast<size_t>(r)] < q.index.size() &&
              (best < 0 || q.index[head[static_cast<size_t>(r)]] < part[static_cast<size_t>(best)].index[head[static_cast<size_t>(best)]]))
            best = r;
        }
        const SmoothedCloud &q = part[static_cast<size_t>(best)];
        const size_t h = head[static_cast<size_t>(best)]++;
        for (int c = 0; c < 3; ++c) {
          s.xyz[3 * k + static_cast<size_t>(c)] = q.xyz[3 * h + static_cast<size_t>(c)];
          s.normal[3 * k + static_cast<size_t>(c)] = q.normal[3 * h + static_cast<size_t>(c)];
        }
        s.curvature[k] = q.curvature[h];
        s.index[k] = q.index[h];
      }
    } else {
      int32_t chunks = 0;
      int64_t voxels = 0;
      for (auto &d : dev_) {  // the same plan on every GPU (same cloud, same capacity)
        int64_t total = 0;
        int32_t c = 0;
        d->check(pcp_mls_stream_begin(d->get(), &params_, int64_t(1) << 26, &total, &c));
        if (&d != &dev_[0] && c != chunks) throw std::runtime_error("pcp_multi: the GPUs disagree about the voxel chunks");
        chunks = c;
        voxels = total;
      }
      // The upsampled cloud goes through the host here and back onto every GPU for the last filter: fine up to what one upload
      // holds.  Beyond that (the reference's 1 mm x 4 on a 10 M-point map makes 2.8e9 rows: 90 GB of host memory, more rows than
      // a cloud has indices) the whole chain runs in its streamed form on the first GPU -- the same rows, one GPU's 2.7 s
      // (pcp_cloud_smooth_stream_*; its chunks over several GPUs: DESIGN.md section 8).  PCP_MULTI_STREAM_ABOVE: another limit (tests).
      int64_t stream_above = int64_t(1) << 30;
      if (const char *e = std::getenv("PCP_MULTI_STREAM_ABOVE")) stream_above = std::max<long long>(1, std::atoll(e));
      if (voxels > stream_above) {
        dev_[0]->uploadCloud(x, y, z, n);
        SmoothedCloud all;
        CloudSmooth(*dev_[0], params_).processWithOutlierRemovalStreamed(int64_t(1) << 28, [&](const SmoothedCloud &c) {
          all.xyz.insert(all.xyz.end(), c.xyz.begin(), c.xyz.end());
          all.normal.insert(all.normal.end(), c.normal.begin(), c.normal.end());
          all.curvature.insert(all.curvature.end(), c.curvature.begin(), c.curvature.end());
          all.index.insert(all.index.end(), c.index.begin(), c.index.end());
        });
        return all;
      }
      for (int32_t c = 0; c < chunks; ++c) {
        Device &d = *dev_[static_cast<size_t>(c % size())];
        int64_t count = 0;
        d.check(pcp_mls_stream_seek(d.get(), c));
        d.check(pcp_mls_stream_next(d.get(), &count));
        append(d, count);
      }
    }
    // SOR 2 on the smoothed cloud (:158-164)
    const size_t m2 = s.curvature.size();
    std::vector<float> sx(m2), sy(m2), sz(m2);
    for (size_t k = 0; k < m2; ++k) {
      sx[k] = s.xyz[3 * k];
      sy[k] = s.xyz[3 * k + 1];
      sz[k] = s.xyz[3 * k + 2];
    }
    const std::vector<int32_t> idx2 = outlierRemoval(sx.data(), sy.data(), sz.data(), static_cast<int64_t>(m2));
    SmoothedCloud out;
    out.xyz.resize(3 * idx2.size());
    out.normal.resize(3 * idx2.size());
    out.curvature.resize(idx2.size());
    out.index.resize(idx2.size());
    for (size_t k = 0; k < idx2.size(); ++k) {
      const size_t j = static_cast<size_t>(idx2[k]);
      for (int c = 0; c < 3; ++c) {
        out.xyz[3 * k + static_cast<size_t>(c)] = s.xyz[3 * j + static_cast<size_t>(c)];
        out.normal[3 * k + static_cast<size_t>(c)] = s.normal[3 * j + static_cast<size_t>(c)];
      }
      out.curvature[k] = s.curvature[j];
      out.index[k] = idx1[static_cast<size_t>(s.index[j])];  // index into the caller's cloud
    }
    return out;
  }

 private:
  // pcl::StatisticalOutlierRemoval over all GPUs (cloudSmooth.cpp:109-116,160-164): the cloud on every GPU, the queries
  // dealt out by slabs of the filter's own spatial order (pcp_sor_partial: whole wavefronts, whole statistic chunks), the
  // chunk sums of the slabs put together -- the array one GPU computes, so the threshold is the one-GPU threshold bit for
  // bit -- and every GPU classifies its own slab (pcp_sor_finish: flags under the caller's indices, zero for the other
  // slabs' points).  The exchange is ceil(n / 16384) pairs of doubles and the flags: RCCL all-reduces on device memory (through
  // the host only in the one-GPU rehearsal); the calls synchronise with the host, hence one host thread per GPU.  Returns
  // the indices kept, ascending.
  std::vector<int32_t> outlierRemoval(const float *x, const float *y, const float *z, int64_t n) {
    const int N = size();
    const int64_t chunk = pcp_sor_chunk_points(), chunks = (n + chunk - 1) / chunk;
    std::vector<double> sums(static_cast<size_t>(2 * std::max<int64_t>(chunks, 1)));
    std::vector<std::vector<uint8_t>> keep(static_cast<size_t>(N), std::vector<uint8_t>(static_cast<size_t>(n)));
    std::vector<std::string> failure(static_cast<size_t>(N));
    auto on_every_gpu = [&](auto &&body) {
      std::vector<std::thread> th;
      for (int r = 0; r < N; ++r)
        th.emplace_back([&, r] {
          try {
            body(r, *dev_[static_cast<size_t>(r)]);
          } catch (const std::exception &e) {
            failure[static_cast<size_t>(r)] = e.what();
          }
        });
      for (auto &t : th) t.join();
      for (const auto &f : failure)
        if (!f.empty()) throw std::runtime_error(f);
    };
    if (comm_.empty()) {
      // PCP_MULTI_REHEARSAL (one GPU): the slabs' arrays put together on the host
      on_every_gpu([&](int r, Device &d) {
        d.uploadCloud(x, y, z, n);
        std::vector<double> mine(static_cast<size_t>(2 * std::max<int64_t>(chunks, 1)));
        int64_t first = 0, cnt = 0;
        d.check(pcp_sor_partial(d.get(), params_.sor_mean_k, r, N, chunks, mine.data(), &first, &cnt));
        std::copy(mine.begin(), mine.begin() + 2 * cnt, sums.begin() + 2 * first);  // disjoint ranges: no lock
      });
      on_every_gpu([&](int r, Device &d) {
        d.check(pcp_sor_finish(d.get(), params_.sor_std_mul, sums.data(), chunks, r, N, keep[static_cast<size_t>(r)].data(), nullptr));
      });
    } else {
      // N GPUs: the chunk sums and the keep flags never visit the host in between.  Every GPU holds a zeroed array of all
      // chunk sums and writes its slab's pairs into it (pcp_sor_partial, device memory); one ncclAllReduce(SUM) puts the slabs
      // together -- every pair has ONE non-zero contribution, so the sums are the one-GPU array bit for bit --; every GPU
      // classifies its slab into a device flag array (zero for the other slabs' points) and one ncclAllReduce(MAX) is their
      // OR; GPU 0's copy comes down once.
      std::vector<double *> dsum(static_cast<size_t>(N), nullptr);
      std::vector<uint8_t *> dkeep(static_cast<size_t>(N), nullptr);
      auto release = [&]() {
        for (int r = 0; r < N; ++r) {
          (void)hipSetDevice(r);
          if (dsum[static_cast<size_t>(r)]) (void)hipFree(dsum[static_cast<size_t>(r)]);
          if (dkeep[static_cast<size_t>(r)]) (void)hipFree(dkeep[static_cast<size_t>(r)]);
        }
      };
      const size_t sum_bytes = static_cast<size_t>(2 * std::max<int64_t>(chunks, 1)) * sizeof(double);
      try {
        on_every_gpu([&](int r, Device &d) {
          check_hip(hipSetDevice(r), "hipSetDevice");
          check_hip(hipMalloc(reinterpret_cast<void **>(&dsum[static_cast<size_t>(r)]), sum_bytes), "hipMalloc(chunk sums)");
          check_hip(hipMalloc(reinterpret_cast<void **>(&dkeep[static_cast<size_t>(r)]), static_cast<size_t>(std::max<int64_t>(n, 1))), "hipMalloc(keep flags)");
          check_hip(hipMemset(dsum[static_cast<size_t>(r)], 0, sum_bytes), "hipMemset(chunk sums)");
          d.uploadCloud(x, y, z, n);
          int64_t first = 0, cnt = 0;
          d.check(pcp_sor_partial(d.get(), params_.sor_mean_k, r, N, chunks, dsum[static_cast<size_t>(r)] + 2 * (chunks * r / N), &first, &cnt));
        });
        check_nccl(ncclGroupStart(), "ncclGroupStart");
        for (int r = 0; r < N; ++r)
          check_nccl(ncclAllReduce(dsum[static_cast<size_t>(r)], dsum[static_cast<size_t>(r)], static_cast<size_t>(2 * chunks), ncclDouble, ncclSum,
                                   comm_[static_cast<size_t>(r)], stream_[static_cast<size_t>(r)]), "ncclAllReduce(SOR chunk sums, SUM)");
        check_nccl(ncclGroupEnd(), "ncclGroupEnd");
        on_every_gpu([&](int r, Device &d) {
          check_hip(hipSetDevice(r), "hipSetDevice");
          check_hip(hipStreamSynchronize(stream_[static_cast<size_t>(r)]), "hipStreamSynchronize");
          d.check(pcp_sor_finish(d.get(), params_.sor_std_mul, dsum[static_cast<size_t>(r)], chunks, r, N, dkeep[static_cast<size_t>(r)], nullptr));
        });
        check_nccl(ncclGroupStart(), "ncclGroupStart");
        for (int r = 0; r < N; ++r)
          check_nccl(ncclAllReduce(dkeep[static_cast<size_t>(r)], dkeep[static_cast<size_t>(r)], static_cast<size_t>(n), ncclUint8, ncclMax,
                                   comm_[static_cast<size_t>(r)], stream_[static_cast<size_t>(r)]), "ncclAllReduce(SOR keep flags, MAX)");
        check_nccl(ncclGroupEnd(), "ncclGroupEnd");
        check_hip(hipSetDevice(0), "hipSetDevice");
        check_hip(hipMemcpyAsync(keep[0].data(), dkeep[0], static_cast<size_t>(n), hipMemcpyDeviceToHost, stream_[0]), "hipMemcpyAsync(keep flags)");
        for (int r = 0; r < N; ++r) {
          check_hip(hipSetDevice(r), "hipSetDevice");
          check_hip(hipStreamSynchronize(stream_[static_cast<size_t>(r)]), "hipStreamSynchronize");
        }
      } catch (...) {
        release();
        throw;
      }
      release();
    }
    std::vector<int32_t> idx;
    for (int64_t i = 0; i < n; ++i) {
      uint8_t k = 0;
      for (int r = 0; r < N; ++r) k |= keep[static_cast<size_t>(r)][static_cast<size_t>(i)];
      if (k) idx.push_back(static_cast<int32_t>(i));
    }
    return idx;
  }

  static void check_hip(hipError_t e, const char *what) {
    if (e != hipSuccess) throw std::runtime_error(std::string("pcp_multi: ") + what + ": " + hipGetErrorString(e));
  }
  static void check_nccl(ncclResult_t r, const char *what) {
    if (r != ncclSuccess) throw std::runtime_error(std::string("pcp_multi: ") + what + ": " + ncclGetErrorString(r));
  }

  std::vector<std::unique_ptr<Device>> dev_;
  std::vector<hipStream_t> stream_;  // N > 1 on real GPUs: RCCL's stream per GPU
  std::vector<ncclComm_t> comm_;
  pcp_mls_params params_;
};

}  // namespace pcp_amd
