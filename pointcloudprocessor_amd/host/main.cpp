// main.cpp -- `PointCloudProcessor` command line on top of libpcp_hip.so.
//
// Same flags, same odometry / PCD inputs, same output files and exit codes as the
// reference binary (PCP/src/main.cpp:7-71, PCP/src/PointCloudProcessor.cpp:1007-1032):
//   --point_cloud_path/-p --odometry_path/-o --images_folder/-i --mask_image_folder/-m
//   --output_path/-t --enableMLS --enableNIDOptimize --enableInitialGuessManual --help/-h
// Stages kept on the host: odometry parsing (:965-1005), keyframe selection (:1050-1075,
// hpp:151-191), trajectory crop (:92-136), PCD reading / ASCII writing.  Hot path on the
// GPU through pcp_shim.hpp.
//
// Differences (this image has no OpenCV):
//   * images are decoded by host/image_io.hpp: <images_folder><ts>.jpg (baseline JPEG, libjpeg's
//     arithmetic, bit-exact with Pillow / libjpeg-turbo here) and masks <mask_folder><ts>.png;
//     <ts>.ppm / <ts>.pgm are accepted when the .jpg / .png is absent.
//   * the 8-bit BGR->HSV->BGR round trip of generateColorMap (:722-741, Appendix B5) is applied by the
//     library while it packs the decoded image (pcp_set_image_adjust: OpenCV 4.2's integer forward routine
//     exactly, its scalar float backward routine); the NID stage reads the raw pixels, as calibrate.cpp does.
//   * --enableNIDOptimize runs the NID cost on the GPU with a BFGS on SE(3) in place of
//     ceres::Solve (same cost, gradient, domain limits and outer loop; not Ceres' line search).
//   * --enableInitialGuessManual is accepted and rejected with an exception (exit -2): the
//     interactive GUI is out of scope.
//   * --cull zbuffer|hpr|hpr_candidates (new, default zbuffer): which of ViewCulling's two routines decides visibility.
//     zbuffer = view_culling (view_culling.cpp:52-174, the routine north_star names; its call is commented out at :43);
//     hpr = hidden_points_removal (:266-334), the routine the reference binary actually calls (:46), flip + convex
//     hull on the GPU; hpr_candidates = only its candidate filter (:276-288), a frustum cull.
//   * --mlsVoxelSize v, --mlsDilationIterations k, --mlsUpsampling none|vgd (new): the three MLSParameters the reference
//     hard-codes (upsampling VOXEL_GRID_DILATION, 0.001 m, 4 iterations, PointCloudProcessor.cpp:78-81; the defaults here);
//     at 1 mm x 4 every input point becomes up to 729 output points.
//   * --gpus N (new, default 1): the map is sharded by point index over N GPUs of this node (pcp_multi.hpp: one
//     process, N contexts, RCCL all-reduce(MIN) of the depth maps over xGMI, images broadcast over xGMI); every
//     output file is identical to the one-GPU run.  The NID refinement sums its joint histograms over the shards
//     (same optimum, last-digit differences in the printed cost); --enableMLS deals the MLS queries / the dilated voxel
//     chunks out over the GPUs (MultiCloudSmooth), the two outlier-removal brackets run on GPU 0.
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <malloc.h>
#include <memory>
#include <sstream>
#include <thread>
#include <exception>
#include <mutex>
#include <condition_variable>

#include "image_io.hpp"
#include "pcd_io.hpp"
#include "pcp_multi.hpp"
#include "pcp_shim.hpp"

namespace fs = std::filesystem;
using namespace pcp_amd;

// Wall-clock split of a run (PCP_CLI_TIMING=<file>: one JSON object, seconds): where the time of the command line goes --
// the reference has no such report; `bench.py`'s cli_e2e leg reads it (at scale the ASCII I/O dominates).
struct PhaseClock {
  std::vector<std::pair<std::string, double>> phases;
  std::mutex mu;
  using clock = std::chrono::steady_clock;
  static double since(clock::time_point t0) { return std::chrono::duration<double>(clock::now() - t0).count(); }
  void add(const std::string &name, double s) {
    std::lock_guard<std::mutex> lk(mu);
    for (auto &p : phases)
      if (p.first == name) {
        p.second += s;
        return;
      }
    phases.emplace_back(name, s);
  }
  void write(const char *path, double total) {
    std::ofstream f(path);
    f << "{";
    for (const auto &p : phases) f << "\"" << p.first << "\": " << p.second << ", ";
    f << "\"total\": " << total << "}\n";
  }
};
static PhaseClock g_clock;
struct Phase {  // adds its lifetime to a named phase
  std::string name;
  PhaseClock::clock::time_point t0 = PhaseClock::clock::now();
  explicit Phase(const char *n) : name(n) {}
  ~Phase() { g_clock.add(name, PhaseClock::since(t0)); }
};

struct Frame {  // FrameData (PCP/include/FrameData.hpp:89-126)
  std::string imagePath, maskImagePath;
  double imageTimestamp = 0;
  pcp_pose pose{};
};

struct Options {
  std::string pointCloudPath, odometryPath, imagesFolder, maskImageFolder, outputPath = ".";
  bool have_p = false, have_o = false, have_i = false;
  bool enableMLS = false, enableNIDOptimize = false, enableInitialGuessManual = false;
  bool help = false;
  bool skip_filtered_dumps = false;
  int gpus = 1;
  int cull_mode = PCP_CULL_ZBUFFER;
  float mls_voxel_size = -1.0f;  // < 0: the reference's constants (PointCloudProcessor.cpp:67-86)
  int mls_dilation_iterations = -1;
  int mls_upsampling = -1;
};

static bool parse_bool(const std::string &v) {  // boost::program_options bool semantics
  std::string s;
  for (char c : v) s += static_cast<char>(std::tolower(c));
  if (s == "1" || s == "true" || s == "yes" || s == "on") return true;
  if (s == "0" || s == "false" || s == "no" || s == "off") return false;
  throw std::runtime_error("the argument ('" + v + "') for a boolean option is invalid");
}

static Options parse(int argc, char **argv) {
  Options o;
  for (int k = 1; k < argc; ++k) {
    std::string a = argv[k], val;
    bool has_val = false;
    const size_t eq = a.find('=');
    if (a.rfind("--", 0) == 0 && eq != std::string::npos) {
      val = a.substr(eq + 1);
      a = a.substr(0, eq);
      has_val = true;
    }
    auto next = [&]() -> std::string {
      if (has_val) return val;
      if (k + 1 >= argc) throw std::runtime_error("the required argument for option '" + a + "' is missing");
      return argv[++k];
    };
    if (a == "--help" || a == "-h") o.help = true;
    else if (a == "--point_cloud_path" || a == "-p") { o.pointCloudPath = next(); o.have_p = true; }
    else if (a == "--odometry_path" || a == "-o") { o.odometryPath = next(); o.have_o = true; }
    else if (a == "--images_folder" || a == "-i") { o.imagesFolder = next(); o.have_i = true; }
    else if (a == "--mask_image_folder" || a == "-m") o.maskImageFolder = next();
    else if (a == "--output_path" || a == "-t") o.outputPath = next();
    else if (a == "--enableMLS") o.enableMLS = parse_bool(next());
    else if (a == "--enableNIDOptimize") o.enableNIDOptimize = parse_bool(next());
    else if (a == "--enableInitialGuessManual") o.enableInitialGuessManual = parse_bool(next());
    else if (a == "--skip_filtered_dumps") o.skip_filtered_dumps = parse_bool(next());
    else if (a == "--gpus") o.gpus = std::stoi(next());
    else if (a == "--mlsVoxelSize") o.mls_voxel_size = std::stof(next());
    else if (a == "--mlsDilationIterations") o.mls_dilation_iterations = std::stoi(next());
    else if (a == "--mlsUpsampling") {
      const std::string v = next();
      if (v == "none") o.mls_upsampling = 0;
      else if (v == "vgd") o.mls_upsampling = 3;
      else throw std::runtime_error("the argument ('" + v + "') for option '--mlsUpsampling' is invalid (none, vgd)");
    }
    else if (a == "--cull") {
      const std::string v = next();
      if (v == "zbuffer") o.cull_mode = PCP_CULL_ZBUFFER;
      else if (v == "hpr") o.cull_mode = PCP_CULL_HPR;
      else if (v == "hpr_candidates") o.cull_mode = PCP_CULL_HPR_CANDIDATES;
      else throw std::runtime_error("the argument ('" + v + "') for option '--cull' is invalid (zbuffer, hpr, hpr_candidates)");
    }
    else throw std::runtime_error("unrecognised option '" + a + "'");
  }
  return o;
}

static void usage(std::ostream &os) {
  os << "Allowed options:\n"
        "  -h [ --help ]                         Produce help message\n"
        "  -p [ --point_cloud_path ] arg         Path to the point cloud data file\n"
        "  -o [ --odometry_path ] arg            Path to odometry data file\n"
        "  -i [ --images_folder ] arg            Path to directory containing images\n"
        "  -m [ --mask_image_folder ] arg        Path to directory for segmented images\n"
        "  -t [ --output_path ] arg (=.)         Path to save processed output\n"
        "  --enableMLS arg (=0)                  Enable MLS smoothing\n"
        "  --enableNIDOptimize arg (=0)          Enable NID-based camera pose optimization\n"
        "  --enableInitialGuessManual arg (=0)   Enable manual pickup point based camera pose optimization\n";
}

class Processor {
 public:
  explicit Processor(const Options &o) : opt(o), enableMaskSegmentation(!o.maskImageFolder.empty()) {}
  ~Processor() {
    stopDecoders();
    if (device_thread.joinable()) device_thread.join();
  }

  void process() {  // PointCloudProcessor::process, PointCloudProcessor.cpp:1007-1032
    // The GPU context takes a quarter of a second to come up (HIP runtime, code objects, queues): it is created on a thread of
    // its own while this one reads the odometry and the map and writes the crop, and the keyframes' decoders start as soon as
    // the keyframes are known -- by the time the device is ready the first images wait decoded (end to end, 1 M points x 32
    // keyframes from a tmpfs: 0.60 -> 0.42-0.43 s without the per-keyframe dumps, 0.69 -> 0.48 s with them; profiles/r05b_cli_e2e_probe.log).  Same calls in the same order on this thread.
    device_thread = std::thread([this]() {
      try {
        gpu.reset(new MultiDevice(opt.gpus));
      } catch (...) {
        device_error = std::current_exception();
      }
    });
    loadImagesAndOdometry();
    loadPointCloud();
    generateResultStorageFolder();
    selectKeyframes();
    if (!opt.enableNIDOptimize) startDecoders();  // (with the NID stage the first consumer wants the images unadjusted and again later: decoded on demand)
    setupDevice();
    if (!opt.skip_filtered_dumps) viewCullingAndSaveFilteredPcds();
    if (opt.enableNIDOptimize)
      applyNIDBasedPoseOptimization();
    else if (opt.enableInitialGuessManual)
      throw std::runtime_error("the manual initial-guess GUI is not part of this build");
    pcdColorizationAndSmooth();
  }

 private:
  Options opt;
  bool enableMaskSegmentation;
  std::vector<Frame> frames, keyframes;
  XYZICloud cloud;
  std::unique_ptr<MultiDevice> gpu;
  std::thread device_thread;
  std::exception_ptr device_error;
  int img_w = 0, img_h = 0;
  bool images_uploaded = false, images_adjusted = false;
  std::vector<uint8_t> mask_missing;
  std::vector<double> T_camera_lidar_optimized;

  void loadImagesAndOdometry() {  // :965-1005
    Phase ph("odometry_s");
    std::ifstream vo(opt.odometryPath);
    std::string line;
    while (std::getline(vo, line)) {
      std::istringstream iss(line);
      double ts, x, y, z, qw, qx, qy, qz;
      if (!(iss >> ts >> x >> y >> z >> qw >> qx >> qy >> qz)) break;  // stop at the first malformed line
      Frame f;
      f.pose = {x, y, z, qw, qx, qy, qz};
      f.imageTimestamp = ts;
      const std::string stem = opt.imagesFolder + std::to_string(ts);  // "%f": 6 decimals (:981)
      f.imagePath = stem + ".jpg";
      if (!fs::exists(f.imagePath)) f.imagePath = stem + ".ppm";
      if (!fs::exists(f.imagePath)) continue;  // skip this frame if its image does not exist (:984-987)
      if (enableMaskSegmentation) {
        f.maskImagePath = opt.maskImageFolder + std::to_string(ts) + ".png";  // :991
        if (!fs::exists(f.maskImagePath) && fs::exists(opt.maskImageFolder + std::to_string(ts) + ".pgm"))
          f.maskImagePath = opt.maskImageFolder + std::to_string(ts) + ".pgm";
      }
      frames.push_back(f);
    }
  }

  void loadPointCloud() {  // :92-154
    double mn[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, mx[3] = {DBL_MIN, DBL_MIN, DBL_MIN};  // DBL_MIN: sic (B10)
    for (const auto &f : frames) {
      const double p[3] = {f.pose.x, f.pose.y, f.pose.z};
      for (int a = 0; a < 3; ++a) {
        mn[a] = std::min(mn[a], p[a]);
        mx[a] = std::max(mx[a], p[a]);
      }
    }
    for (int a = 0; a < 3; ++a) {
      mn[a] -= 2.0;
      mx[a] += 2.0;
    }
    XYZICloud original;
    {
      Phase ph("pcd_read_s");
      if (loadPCDFile(opt.pointCloudPath, original) == -1) throw std::runtime_error("Couldn't read point cloud file.");
    }
    std::cout << "Start crop pcd..." << std::endl;
    // pcl::CropBox with Vector4f(min), Vector4f(max): keep min <= p <= max (fp32 bounds)
    const float fmn[3] = {static_cast<float>(mn[0]), static_cast<float>(mn[1]), static_cast<float>(mn[2])};
    const float fmx[3] = {static_cast<float>(mx[0]), static_cast<float>(mx[1]), static_cast<float>(mx[2])};
    XYZICloud cropped;
    const auto t_crop = PhaseClock::clock::now();
    for (size_t i = 0; i < original.size(); ++i) {
      const float p[3] = {original.x[i], original.y[i], original.z[i]};
      if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) continue;
      bool in = true;
      for (int a = 0; a < 3; ++a) in = in && !(p[a] < fmn[a]) && !(p[a] > fmx[a]);
      if (in) cropped.push_back(p[0], p[1], p[2], original.intensity[i]);
    }
    std::cout << "Loaded point cloud with " << original.size() << " points." << std::endl;
    std::cout << "Cropped point cloud with " << cropped.size() << " points." << std::endl;
    const std::string cropPath = opt.outputPath + "scans-crop.pcd";  // outputPath must end in '/' (:131)
    writeASCII_XYZI(cropPath, cropped.x.data(), cropped.y.data(), cropped.z.data(), cropped.intensity.data(), cropped.size());
    std::cout << "Cropped point cloud saved to: " << cropPath << std::endl;
    g_clock.add("crop_and_write_ascii_s", PhaseClock::since(t_crop));
    Phase ph_mls(opt.enableMLS ? "enable_mls_stage_s" : "cloud_move_s");
    if (opt.enableMLS) {
      // CloudSmooth re-reads the ASCII crop it was handed (cloudSmooth.cpp:92): 8 significant digits
      XYZICloud crop8;
      if (loadPCDFile(cropPath, crop8) == -1) {
        std::cerr << "Couldn't read file " << cropPath << std::endl;
        return;
      }
      if (device_thread.joinable()) device_thread.join();  // (one thread at a time creates contexts: pcp_create sets process-wide defaults)
      MultiCloudSmooth smooth(opt.gpus);  // --gpus N: MLS queries / voxel chunks dealt out over the GPUs (pcp_multi.hpp)
      pcp_mls_params mp;
      pcp_default_mls_params(&mp);  // PointCloudProcessor.cpp:67-86
      if (opt.mls_voxel_size > 0.0f) mp.vgd_voxel_size = opt.mls_voxel_size;
      if (opt.mls_dilation_iterations >= 0) mp.vgd_iterations = opt.mls_dilation_iterations;
      if (opt.mls_upsampling >= 0) mp.upsampling = opt.mls_upsampling;
      smooth.initialize(mp);
      SmoothedCloud s = smooth.processWithOutlierRemoval(crop8.x.data(), crop8.y.data(), crop8.z.data(),
                                                         static_cast<int64_t>(crop8.size()));
      const std::string mlsPath = fs::path(cropPath).stem().string() + "_mls.pcd";  // CWD-relative, sic (B14)
      writeASCII_PointNormal(mlsPath, s.xyz.data(), s.normal.data(), s.curvature.data(), s.curvature.size());
      cloud.resize(s.curvature.size());
      for (size_t i = 0; i < cloud.size(); ++i) {
        cloud.x[i] = s.xyz[3 * i];
        cloud.y[i] = s.xyz[3 * i + 1];
        cloud.z[i] = s.xyz[3 * i + 2];
        cloud.intensity[i] = 0.0f;  // PointNormal carries no intensity (copyPointCloud, :144)
      }
      if (const char *dump = std::getenv("PCP_CLI_DUMP_SMOOTHED")) {
        // test hook: the smoothed cloud as the colour stage receives it (raw fp32 xyz triples; the ASCII file above
        // carries 8 significant digits, one short of a float's round trip)
        std::ofstream df(dump, std::ios::binary);
        df.write(reinterpret_cast<const char *>(s.xyz.data()), static_cast<std::streamsize>(s.xyz.size() * sizeof(float)));
      }
    } else {
      cloud = std::move(original);  // the reference reloads the same file (:148)
      std::cout << "Loaded point cloud with " << cloud.size() << " points." << std::endl;
    }
  }

  void generateResultStorageFolder() {  // :1034-1048
    const fs::path dir(opt.outputPath + "filtered_pcd/");
    if (fs::exists(dir)) fs::remove_all(dir);
    fs::create_directories(dir);
  }

  void selectKeyframes() {  // :1050-1075 + markKeyframe hpp:151-191 (distance rule only, B9)
    keyframes.clear();
    int last_idx = -1;
    for (size_t i = 0; i < frames.size(); ++i) {
      bool key = last_idx < 0;
      if (key) std::cout << "First frame is always a keyframe." << std::endl;
      if (!key) {
        const auto &a = frames[i].pose, &b = frames[static_cast<size_t>(last_idx)].pose;
        const double dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
        key = std::sqrt(dx * dx + dy * dy + dz * dz) >= 0.1;
      }
      if (key) {
        keyframes.push_back(frames[i]);
        last_idx = static_cast<int>(i);
      }
    }
  }

  void setupDevice() {
    Phase ph("device_setup_and_cloud_upload_s");
    if (device_thread.joinable()) device_thread.join();
    if (device_error) std::rethrow_exception(device_error);
    if (!gpu) gpu.reset(new MultiDevice(opt.gpus));
    gpu->uploadCloud(cloud.x.data(), cloud.y.data(), cloud.z.data(), static_cast<int64_t>(cloud.size()));
    // image size from the first keyframe image; cull size stays the reference's {4096,3000} (:206,:525)
    if (!keyframes.empty() && decoders) {  // (the decoders are on it already)
      std::unique_lock<std::mutex> lk(decoders->mu);
      decoders->cv.wait(lk, [&] { return decoders->ready[0] != 0; });
      if (decoders->img[0].empty()) throw std::runtime_error("Failed to read image from: " + keyframes[0].imagePath);
      img_w = decoders->img[0].width;
      img_h = decoders->img[0].height;
    } else if (!keyframes.empty()) {
      const Image8 first = read_image_bgr(keyframes[0].imagePath);
      if (first.empty()) throw std::runtime_error("Failed to read image from: " + keyframes[0].imagePath);
      img_w = first.width;
      img_h = first.height;
    }
    pcp_camera cam;
    pcp_default_camera(&cam);
    if (!keyframes.empty()) {
      cam.image_width = img_w;
      cam.image_height = img_h;
    }
    pcp_cull_params cull;
    pcp_default_cull_params(&cull);
    cull.cull_mode = opt.cull_mode;
    gpu->setCamera(cam, &cull);
    std::vector<pcp_pose> poses;
    for (const auto &k : keyframes) poses.push_back(k.pose);
    gpu->setKeyframes(poses);
    if (gpu->size() == 1) {
      // the reference credits a sample to EVERY map point within 10 um of it (radiusSearch, :571); the library to the
      // sample's own point.  Say so when the map holds points that close together (duplicates of merged scans).
      int64_t close = 0;
      if (pcp_close_pairs(gpu->device(0).get(), 2.5e-5, &close) != PCP_OK)  // a map with NaN / infinite points: no grid
        std::cerr << "Warning: " << pcp_last_error(gpu->device(0).get()) << "; the close-pair check is skipped." << std::endl;
      if (close > 0)
        std::cerr << "Warning: " << close << " map points have another point within 25 um; the reference would let them "
                  << "share colour samples (PointCloudProcessor.cpp:571), this build does not." << std::endl;
    }
  }

  void applyNIDBasedPoseOptimization() {  // :156-164 -> calibrate.cpp:42-126
    double cost = 0.0;
    // one GPU: VisualLiDARCalibration on the context; several: the keyframes' joint histograms are summed over the point
    // shards (MultiDevice::calibrate), no GPU ever holds the whole map
    gpu->uploadIntensity(cloud.intensity.data(), static_cast<int64_t>(cloud.size()));
    uploadImages(false);  // VisualLiDARCalibration reads the images itself, without generateColorMap's adjustment
    T_camera_lidar_optimized = gpu->calibrate(&cost);
    std::printf("Final cost: %.3f\n--- T_camera_lidar ---\n", cost);
    for (int r = 0; r < 4; ++r)
      std::printf("%g %g %g %g\n", T_camera_lidar_optimized[4 * r], T_camera_lidar_optimized[4 * r + 1],
                  T_camera_lidar_optimized[4 * r + 2], T_camera_lidar_optimized[4 * r + 3]);
    {  // full-precision copy of the result next to the outputs (not written by the reference)
      std::ofstream tf(opt.outputPath + "T_camera_lidar_optimized.txt");
      char buf[64];
      for (int k = 0; k < 16; ++k) {
        std::snprintf(buf, sizeof(buf), "%.17g%c", T_camera_lidar_optimized[static_cast<size_t>(k)], (k % 4 == 3) ? '\n' : ' ');
        tf << buf;
      }
    }
    std::vector<pcp_pose> poses;
    for (const auto &k : keyframes) poses.push_back(k.pose);
    gpu->setKeyframes(poses, T_camera_lidar_optimized.data(), 0);  // the enableNIDOptimize branch, :504-509
    images_uploaded = false;                                        // set_frames drops the images
  }

  void viewCullingAndSaveFilteredPcds() {  // :178-224
    const size_t n = cloud.size();
    std::vector<float> cam(3 * n);
    for (size_t k = 0; k < keyframes.size(); ++k) {
      const auto t_gpu = PhaseClock::clock::now();
      const std::vector<int32_t> kept = gpu->cull(static_cast<int>(k));
      gpu->cameraCoordinates(static_cast<int>(k), cam);
      g_clock.add("filtered_dumps_gpu_s", PhaseClock::since(t_gpu));
      Phase ph("filtered_dumps_write_ascii_s");
      std::vector<float> x(kept.size()), y(kept.size()), z(kept.size()), in(kept.size());
      for (size_t q = 0; q < kept.size(); ++q) {
        const size_t i = static_cast<size_t>(kept[q]);
        x[q] = cam[i];
        y[q] = cam[n + i];
        z[q] = cam[2 * n + i];
        in[q] = cloud.intensity[i];
      }
      const std::string path =
          opt.outputPath + "filtered_pcd/" + std::to_string(keyframes[k].imageTimestamp) + "_beforeNID" + ".pcd";
      if (writeASCII_XYZI(path, x.data(), y.data(), z.data(), in.data(), kept.size()) == -1)
        throw std::runtime_error("Couldn't save filtered point cloud to PCD file.");
      std::cout << "Before NID optimization: view culling pcd saved to: " << path << ", the point size is "
                << kept.size() << std::endl;
    }
  }

  // cv::imread of every keyframe (and mask) on all host cores -- the decoders are pure functions of the file -- while
  // this thread uploads them in keyframe order; at most `window` decoded keyframes are held at a time (a 4096x3000
  // frame is 37 MB).  The reference decodes one image per keyframe iteration on its one thread.  The decoders may be started
  // ahead of their consumer (startDecoders: process() does, while the device comes up): uploadImages then finds them running.
  struct Decoders {
    std::vector<Image8> img, gray;
    std::vector<uint8_t> ready;
    std::mutex mu;
    std::condition_variable cv;
    size_t n = 0, next = 0, uploaded = 0, window = 0;
    bool stop = false;
    std::vector<std::thread> pool;
  };
  std::unique_ptr<Decoders> decoders;

  void startDecoders() {
    if (decoders) return;
    decoders.reset(new Decoders);
    Decoders &D = *decoders;
    D.n = keyframes.size();
    unsigned threads = std::max(1u, std::min(std::thread::hardware_concurrency(), 16u));
    if (const char *e = std::getenv("PCP_DECODE_THREADS")) threads = static_cast<unsigned>(std::max(1, std::atoi(e)));
    threads = static_cast<unsigned>(std::min<size_t>(threads, std::max<size_t>(D.n, 1)));
    D.window = 2 * static_cast<size_t>(threads) + 2;
    D.img.resize(D.n);
    D.gray.resize(D.n);
    D.ready.assign(D.n, 0);
    auto worker = [this]() {
      Decoders &W = *decoders;
      for (;;) {
        size_t k;
        {
          std::unique_lock<std::mutex> lk(W.mu);
          W.cv.wait(lk, [&] { return W.stop || W.next >= W.n || W.next < W.uploaded + W.window; });
          if (W.stop || W.next >= W.n) return;
          k = W.next++;
        }
        const auto t_dec = PhaseClock::clock::now();
        Image8 a = read_image_bgr(keyframes[k].imagePath);  // cv::imread, :716
        Image8 b;
        if (enableMaskSegmentation) b = read_image_gray(keyframes[k].maskImagePath);  // cv::IMREAD_GRAYSCALE, :775
        g_clock.add("images_decode_thread_seconds", PhaseClock::since(t_dec));  // summed over the decoder threads
        {
          std::lock_guard<std::mutex> lk(W.mu);
          W.img[k] = std::move(a);
          W.gray[k] = std::move(b);
          W.ready[k] = 1;
        }
        W.cv.notify_all();
      }
    };
    for (unsigned t = 0; t < threads; ++t) D.pool.emplace_back(worker);
  }

  void stopDecoders() {
    if (!decoders) return;
    {
      std::lock_guard<std::mutex> lk(decoders->mu);
      decoders->stop = true;
    }
    decoders->cv.notify_all();
    for (auto &th : decoders->pool) th.join();
    decoders.reset();
  }

  void uploadImages(bool adjusted) {
    if (images_uploaded && images_adjusted == adjusted) {
      stopDecoders();
      return;
    }
    Phase ph_all("images_decode_and_upload_wall_s");  // decoders on the host threads, uploads on this one, overlapped
    gpu->setImageAdjust(adjusted);  // cvtColor(BGR2HSV) ... cvtColor(HSV2BGR), :722-741, fused into the upload
    const size_t n = keyframes.size();
    mask_missing.assign(n, 0);
    startDecoders();  // (running already when process() started them ahead)
    Decoders &D = *decoders;
    try {
      for (size_t k = 0; k < n; ++k) {
        {
          std::unique_lock<std::mutex> lk(D.mu);
          D.cv.wait(lk, [&] { return D.ready[k] != 0; });
        }
        std::cout << "Reading image from: " << keyframes[k].imagePath << std::endl;
        if (D.img[k].empty() || D.img[k].width != img_w || D.img[k].height != img_h)
          throw std::runtime_error("Failed to read image from: " + keyframes[k].imagePath);
        {
          Phase ph_up("images_upload_calls_s");
          gpu->uploadImage(static_cast<int>(k), D.img[k].data.data(), static_cast<int64_t>(D.img[k].width) * 3);
        }
        if (enableMaskSegmentation) {
          std::cout << "Reading segment mask image from: " << keyframes[k].maskImagePath << std::endl;
          if (!D.gray[k].empty() && D.gray[k].width == img_w && D.gray[k].height == img_h)
            gpu->uploadMask(static_cast<int>(k), D.gray[k].data.data(), D.gray[k].width);
          else
            mask_missing[k] = 1;  // generateSegmentMap logs it and returns an empty cloud, :776-781
        }
        {
          std::lock_guard<std::mutex> lk(D.mu);
          D.img[k] = Image8();
          D.gray[k] = Image8();
          D.uploaded = k + 1;
        }
        D.cv.notify_all();
      }
    } catch (...) {
      stopDecoders();
      throw;
    }
    stopDecoders();
    images_uploaded = true;
    images_adjusted = adjusted;
  }

  void pcdColorizationAndSmooth() {  // :474-602
    uploadImages(true);
    std::vector<float> wx, wy, wz;  // cloudInWorldWithRGBandMask
    std::vector<float> wxyz;
    std::vector<uint8_t> wrgb;
    std::vector<uint16_t> wmask;
    if (enableMaskSegmentation) {
      for (size_t k = 0; k < keyframes.size(); ++k) {
        VisiblePoints v;
        if (mask_missing[k])  // :779-780: message, empty scanInBodyWithRGBandMask -> PCDWriter throws below (exit -2)
          std::cout << "Failed to read image from: " << keyframes[k].maskImagePath << std::endl;
        else {
          Phase ph("frame_visible_gpu_s");
          v = gpu->frameVisible(static_cast<int>(k));
        }
        Phase ph_w("rgb_mask_dumps_write_ascii_s");
        const std::string path =
            opt.outputPath + "filtered_pcd/" + std::to_string(keyframes[k].imageTimestamp) + "_rgb-mask" + ".pcd";
        if (writeASCII_XYZRGBMask(path, v.xyz_cam.data(), v.rgb.data(), v.mask.data(), v.index.size()) == -1)
          throw std::runtime_error("Couldn't save filtered point cloud to PCD file.");
        std::cout << "Filtered point cloud saved to: " << path << ", the point size is " << v.index.size() << std::endl;
        wxyz.insert(wxyz.end(), v.xyz_world.begin(), v.xyz_world.end());
        wrgb.insert(wrgb.end(), v.rgb.begin(), v.rgb.end());
        wmask.insert(wmask.end(), v.mask.begin(), v.mask.end());
      }
    }
    std::vector<uint8_t> rgb, has;
    {
      Phase ph("colourise_gpu_s");
      gpu->colorize(rgb, has);  // smoothColors + removePointsWithNoColor flag
    }
    Phase ph_w("final_pcd_write_ascii_s");
    XYZICloud out;
    std::vector<uint8_t> out_rgb;
    for (size_t i = 0; i < cloud.size(); ++i)
      if (has[i]) {
        out.push_back(cloud.x[i], cloud.y[i], cloud.z[i], 0.0f);
        out_rgb.insert(out_rgb.end(), {rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]});
      }
    if (enableMaskSegmentation && !wmask.empty()) {  // saveColorizedPointCloud(rgbCloud, withMask), :933-960
      const std::string path = opt.outputPath + "cloudInWorldWithRGBandMask.pcd";
      if (writeASCII_XYZRGBMask(path, wxyz.data(), wrgb.data(), wmask.data(), wmask.size()) == -1)
        throw std::runtime_error("Couldn't save colorized and segment colored point cloud.");
      std::cout << "All colored and segment colored cloud saved to: " << path << std::endl;
    }
    if (out.size() > 0) {  // :912-929
      const std::string path = opt.outputPath + "cloudInWorldWithRGB.pcd";
      if (writeASCII_XYZRGB(path, out.x.data(), out.y.data(), out.z.data(), out_rgb.data(), out.size()) == -1)
        throw std::runtime_error("Couldn't save colorized point cloud.");
      std::cout << "All colored cloud saved to: " << path << std::endl;
    }
  }
};

int main(int argc, char **argv) {
  const auto t_main = PhaseClock::clock::now();
  // Decoded keyframes are 6-37 MB buffers that live for one upload each.  glibc serves such sizes by mmap / munmap until a
  // freed block has raised its threshold: sixteen decoder threads then fault fresh pages in and tear mappings down while this
  // thread's uploads pin and unpin theirs -- all under one address-space lock.  Measured (profiles/r05_cli_e2e_probe_before.log / _after.log, profiles/cli_e2e_probe.py): the
  // 32 uploads of a 1 M-point / 32-keyframe run took 0.50 s when nothing large had been freed before (--skip_filtered_dumps 1)
  // and 0.03 s otherwise.  The threshold is set up front: the buffers come from the heaps and are reused.
  (void)mallopt(M_MMAP_THRESHOLD, 32 << 20);   // (glibc's ceiling)
  (void)mallopt(M_TRIM_THRESHOLD, 1 << 30);
  struct TimingAtExit {  // also after an exception: the phases reached so far
    PhaseClock::clock::time_point t0;
    ~TimingAtExit() {
      if (const char *path = std::getenv("PCP_CLI_TIMING")) g_clock.write(path, PhaseClock::since(t0));
    }
  } timing_at_exit{t_main};
  try {
    const Options o = parse(argc, argv);
    if (o.help) {
      usage(std::cout);
      return 1;
    }
    if (o.have_p && o.have_o && o.have_i) {
      Processor processor(o);
      processor.process();
      std::cout << "Processing completed successfully." << std::endl;
    } else {
      std::cerr << "Error: Missing required arguments." << std::endl;
      usage(std::cerr);
      return -1;
    }
  } catch (const std::exception &e) {
    std::cerr << "Unhandled Exception reached the top of main: " << e.what() << ", application will now exit"
              << std::endl;
    return -2;
  }
  return 0;
}
