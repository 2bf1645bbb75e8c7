// pcp_internal.hpp -- context, device parameter blocks and launch/timing helpers
// shared by the translation units of libpcp_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "pcp_hip.h"

namespace pcp {

// ---- device-side parameter blocks (passed by value as kernel arguments, so they
// land in SGPRs through the kernarg segment; nothing here is per-lane) ----------
struct DevCamera {
  double fx, fy, cx, cy;
  double k1, k2, p1, p2, k3;
  double slack;  // 0.05, view_culling.cpp:157
  float ds_f;    // 14.0f
  double img_wd, img_hd;  // image size as fp64 / cull size as fp32: bounds of the truncation rules
  float cull_wf, cull_hf;
  float ds_rcp;  // RN(1 / ds_f): exact constant division (pcp_device.hpp div_by_ds)
  int32_t ds_fast;  // 1 when ds_f is in the range div_by_ds is proven for
  int32_t ds;
  int32_t img_w, img_h;
  int32_t cull_w, cull_h;
  int32_t mw, mh;  // cull_w/ds, cull_h/ds
  int32_t enable_zbuf;
  int32_t cull_mode;   // PCP_CULL_ZBUFFER / PCP_CULL_HPR_CANDIDATES (enable_zbuf is 0 with the latter; PCP_CULL_HPR runs
                       // the candidate filter here and the hull in pcp_hpr.hip)
  int32_t match_mode;  // PCP_MATCH_IDENTITY / PCP_MATCH_ROUNDTRIP
  double cull_wd, cull_hd;  // cull size as fp64: bounds of hidden_points_removal's (int)u, (int)v rule
  float match_r2;      // f32(1e-5 * 1e-5): radiusSearch(epsilon) squared radius, PointCloudProcessor.cpp:482,571
  int32_t pretest;  // 1: run the conservative fp32 rejection test before the fp64 projection
  int32_t frames_bounded;  // 1: no entry of any keyframe's w2c exceeds 2^40 in magnitude (pcp_set_frames; see divide_xy_by_z)
  // fp32 copies for the rejection test (pcp_device.hpp surely_rejected): the coefficients (their absolute
  // values are source modifiers of the same registers) and the (u, v) box outside of which BOTH the cell rule
  // and the pixel rule reject, widened by 0.5 px
  float qfx, qfy, qcx, qcy;
  float qk1, qk2, qk3, qp1, qp2;
  float u_lo, u_hi, v_lo, v_hi;
};

// One keyframe: w2c / c2w 3x4 row-major fp32 (A1), the pose translation used by
// computeOrientationScore (hpp:207, B4) and an upper bound of the spectral norm of
// w2c's linear part (tile culling), and c2w.inverse() as the reference recomputes it in fp32 for every
// match (PointCloudProcessor.cpp:578).  192 B: three 64-B scalar-cache lines.
struct DevFrame {
  float w2c[12];
  float c2w[12];
  double px, py, pz;
  double norm_bound;
  float c2w_inv[12];
  float pad_[4];
};
static_assert(sizeof(DevFrame) == 192, "DevFrame layout");

// Per-point top-5 state, SoA over points: score[k][n], rgb[k][n], frame[k][n], count[n].
constexpr int kTopM = 5;  // PointCloudProcessor.cpp:615

struct TimingSlot {
  double total_ms = 0.0;
  int64_t launches = 0;
};

struct PendingEvent {
  hipEvent_t start, stop;
  int32_t kernel;
};

// seconds and bytes of the device allocations of this process (hipMalloc / hipFree inside DevBuf), for diagnostics
struct AllocTally {
  double seconds = 0.0, bytes = 0.0;
};
inline AllocTally &alloc_tally() {
  static AllocTally t;
  return t;
}
struct AllocClock {
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  ~AllocClock() { alloc_tally().seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};

template <typename T>
struct DevBuf {
  T *p = nullptr;
  size_t count = 0;
  hipError_t ensure(size_t n) {
    if (n <= count && p) return hipSuccess;
    AllocClock clk;  // (device allocations cost 20-40 ms per GB on this platform: the diagnostics of the long calls report them)
    if (p) (void)hipFree(p);
    p = nullptr;
    count = 0;
    if (n == 0) return hipSuccess;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), n * sizeof(T));
    if (e == hipSuccess) {
      count = n;
      alloc_tally().bytes += static_cast<double>(n) * sizeof(T);
    }
    return e;
  }
  void release() {
    if (!p) return;
    AllocClock clk;
    (void)hipFree(p);
    p = nullptr;
    count = 0;
  }
};

// hidden_points_removal: the scratch of ONE keyframe's hull (pcp_hpr.hip).  The keyframes of a run are independent, so the
// whole-run pass keeps several of them in flight, each on a lane of its own: a stream, the buffers and a pinned readback
// into which the device publishes the candidates' count and bounds (the one host wait of a keyframe polls it).  Lane 0
// also serves the single-keyframe calls, on the context's stream.
struct HprLane {
  hipStream_t own_stream = nullptr;  // created on first use by the whole-run pass
  hipStream_t stream = nullptr;      // the stream the keyframe in flight was queued on
  unsigned long long seq = 0;        // sequence number of the keyframe whose counts the readback is waited for
  DevBuf<int32_t> index, i32, tiles;
  DevBuf<double> f64, cells_d, cont;  // cont: the searches k_hpr_tilt hands on (TiltCont records)
  DevBuf<uint8_t> state;
  DevBuf<unsigned long long> stats;
  void *readback = nullptr;  // pinned, kReadbackBytes
  // the keyframe between hpr_begin and hpr_finish
  bool busy = false;
  int32_t frame = -1;
  uint8_t *d_flags = nullptr;
  uint32_t *hull_plane = nullptr;
  uint32_t bit = 0;
  void release() {
    index.release();
    i32.release();
    tiles.release();
    f64.release();
    cells_d.release();
    cont.release();
    state.release();
    stats.release();
    if (readback) (void)hipHostFree(readback);
    readback = nullptr;
    if (own_stream) (void)hipStreamDestroy(own_stream);
    own_stream = nullptr;
  }
};

}  // namespace pcp

struct pcp_context {
  int32_t device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  // pinned host scratch for the small device-to-host readbacks (counts, bounds, tallies): a copy into pageable memory is
  // staged by the runtime and costs 20-30 us more per wait; 64 KB, allocated by pcp_create (nullptr: pageable fallback)
  void *readback = nullptr;
  static constexpr size_t kReadbackBytes = 65536;
  mutable std::string error;

  // configuration
  bool have_camera = false;
  pcp_camera camera{};
  pcp_cull_params cull{};
  pcp::DevCamera dcam{};

  // cloud: original order (per-keyframe drop-in calls) and spatially sorted copy
  // (batched run + MLS); perm[j] = original index of sorted point j.
  int64_t n = 0;
  int64_t nonfinite_points = 0;  // uploaded points with a NaN or infinite coordinate (the smoothing stages refuse them)
  pcp::DevBuf<float> xyz;    // x[n] y[n] z[n]
  pcp::DevBuf<float> sxyz;   // sorted x[n] y[n] z[n]
  pcp::DevBuf<int32_t> perm; // n
  pcp::DevBuf<int32_t> inv_perm;  // n: inv_perm[perm[j]] = j (un-permutes per-point results with coalesced stores)
  pcp::DevBuf<uint32_t> rgba_sorted;  // packed result in Morton order, before the un-permute
  std::vector<float> host_min = {0, 0, 0}, host_max = {0, 0, 0};

  // frames
  int32_t n_frames = 0;
  std::vector<pcp_pose> poses;
  std::vector<pcp::DevFrame> hframes;
  pcp::DevBuf<pcp::DevFrame> frames;
  pcp::DevBuf<uint32_t> images;  // n_frames * img_h * img_w  (B | G<<8 | R<<16 | mask<<24)
  std::vector<uint8_t> image_set, mask_set;
  // pcp_upload_image_async: kUploadLanes streams of their own, taken in turn, each with one staging buffer (a stream
  // is in order): the copy of keyframe i + 1 crosses PCIe while keyframe i is packed, and both overlap the compute
  // stream.  An event per keyframe for the consumers, and an event of the compute stream that the next uploads wait
  // for when kernels that touch the texels were queued since the last one.
  static constexpr int kUploadLanes = 2;
  hipStream_t upload_stream[kUploadLanes] = {nullptr, nullptr};
  pcp::DevBuf<uint8_t> upload_stage[kUploadLanes];
  std::vector<hipEvent_t> image_event;   // per keyframe, recorded on its lane after its pack kernel
  std::vector<uint8_t> image_pending;    // 1: the consumer has not yet made its stream wait for image_event[f]
  std::vector<uint8_t> image_lane;       // lane of the keyframe's latest upload
  std::vector<uint64_t> image_seq;       // queue position of the keyframe's latest upload
  bool lane_must_wait[kUploadLanes] = {false, false};  // texels_idle not yet waited for on this lane
  uint64_t upload_seq = 0;   // queue position of the latest packed keyframe
  uint64_t upload_turn = 0;  // upload calls so far: the lanes take turns per call (a block of keyframes is one call)
  hipEvent_t texels_idle = nullptr;      // recorded on the compute stream
  bool texels_touched = false;           // compute-stream work on the texel buffer since the last wait
  // generateColorMap's 8-bit BGR -> HSV -> BGR round trip, fused into the pack kernel (pcp_set_image_adjust)
  bool adjust_images = false;
  float saturation_scale = 1.0f, brightness_scale = 1.0f;
  pcp::DevBuf<int32_t> hsv_tables;  // sdiv_table[256], hdiv_table180[256] (OpenCV RGB2HSV_b)

  // depth maps [n_frames][mh*mw] as uint view of positive floats
  pcp::DevBuf<uint32_t> depth;
  pcp::DevBuf<unsigned long long> depth_sq;  // per cell, min of the squared fp64 range (bit pattern) during a depth pass
  std::vector<uint8_t> depth_valid;
  bool depth_from_batch = false;  // pcp_set_depth_source: single-keyframe calls use the batched (merged) maps

  // tiles = wavefront-sized runs of 64 Morton-ordered points: bounding spheres
  // (x, y, z, radius) and the tile x keyframe visibility masks [tile][mask_words]
  int64_t n_tiles = 0;
  pcp::DevBuf<float> tile_sphere;
  pcp::DevBuf<uint32_t> tile_mask, group_mask;
  pcp::DevBuf<uint32_t> tile_inside;  // pairs whose whole tile images inside the acceptance box (a hint: skip the pre-test)
  int32_t mask_words = 0;
  // longest-work-first order of the tiles for the batched passes (pcp_colour.hip k_work_*)
  pcp::DevBuf<int32_t> tile_work, tile_order, work_hist;
  bool tile_order_live = false;

  // per-point colour state (sorted order) and packed results
  pcp::DevBuf<float> top_score;     // 5*n
  pcp::DevBuf<uint32_t> top_rgb;    // 5*n
  pcp::DevBuf<int32_t> top_frame;   // 5*n
  pcp::DevBuf<int32_t> view_count;  // n
  // packed results in input order, double-buffered so that the device-to-host copy of one
  // run (copy stream) overlaps the kernels of the next
  pcp::DevBuf<uint32_t> rgba2[2];
  int32_t rgba_cur = 0;
  hipStream_t copy_stream = nullptr;
  hipEvent_t result_ready[2] = {nullptr, nullptr}, copy_done[2] = {nullptr, nullptr};
  bool copy_pending[2] = {false, false};
  bool colour_state_live = false;
  bool colour_result_live = false;
  bool result_sorted[2] = {false, false};  // rgba2[k] holds its words in the sorted (Morton) order of the passes: readers un-permute
  bool last_pass_sorted = false;           // ... as the pass that has just run left them (end_result moves it to the buffer)

  // scratch for the single-frame calls
  pcp::DevBuf<int32_t> s_cell, s_pixel;
  pcp::DevBuf<float> s_range, s_cam;
  pcp::DevBuf<uint8_t> s_keep;
  pcp::DevBuf<uint32_t> s_u32;
  pcp::DevBuf<unsigned long long> s_counter;
  pcp::DevBuf<int32_t> s_tiles;

  // hidden_points_removal (pcp_hpr.hip): candidate list, flipped points (candidate and cell order), cells, states
  static constexpr int kHprMaxLanes = 8;
  pcp::HprLane hpr_lane[kHprMaxLanes];
  int32_t hpr_last_lane = 0;  // whose tallies pcp_hpr_stats reads
  double hpr_host_wait_s = 0.0;  // host time of the last whole-run pass spent waiting for the keyframes' counts (PCP_HPR_DEBUG prints it)
  hipEvent_t hpr_fork = nullptr, hpr_join[kHprMaxLanes] = {};
  pcp::DevBuf<uint32_t> hull_bits;  // whole run: uint32[(F + 31) / 32][n], bit f & 31 of word (f >> 5, j) = point j (Morton
                                    // order) is a hull vertex of keyframe f
  std::vector<uint8_t> hull_valid;  // per keyframe: hull bits imported (index shards)
  int64_t hpr_stats[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  bool hpr_stats_pending = false;  // the tallies of the last hull still sit on the device (h_stats)

  // MLS: uniform grid (cell id / in-cell rank per point, cell starts, cell-sorted
  // order + coordinates), per-input-point results, compacted outputs
  pcp::DevBuf<int32_t> g_cell, g_rank, g_start, g_order;
  pcp::DevBuf<unsigned long long> g_occ;  // sparse grids: one bit per cell
  pcp::DevBuf<int32_t> g_occ_rank;        // ... and the set bits before each 64-bit word
  pcp::DevBuf<float> g_xyz;      // cell-sorted x[n] y[n] z[n]
  pcp::DevBuf<float> m_tmp;      // 8 floats per input point (xyz, normal, curvature, pad: one 32-byte sector), input order
  pcp::DevBuf<float> s_dist;     // StatisticalOutlierRemoval: mean kNN distance per point
  pcp::DevBuf<double> m_state;   // per-point MLSResult (mean, axes, c_vec ...) for upsampling
  pcp::DevBuf<uint8_t> m_flag;   // n
  pcp::DevBuf<double> m_sums;    // SOR statistics
  pcp::DevBuf<int32_t> c_index;  // pcp_cloud_smooth: survivors of the 1st SOR (indices into the uploaded cloud)
  int32_t sor_partial_slab = -1, sor_partial_slabs = -1;  // the slab of the cell order whose mean distances the last pcp_sor_partial left in s_dist
  bool sor_distances_live = false;  // s_dist holds the mean distances of the last pcp_sor (caller's order)
  pcp::DevBuf<uint8_t> c_mark;    // pcp_cloud_smooth: per uploaded point, survives the whole chain
  pcp::DevBuf<int32_t> c_where;   // ... and the result row that holds it
  pcp::DevBuf<float> c_xyz, c_xyz2;  // pcp_cloud_smooth: intermediate clouds (SoA)
  pcp::DevBuf<uint32_t> v_bitmap;  // dilated voxel set: the bits of the occupied bricks (or the dense bitmap over the bounding box, PCP_VGD_DENSE=1)
  pcp::DevBuf<int32_t> v_offsets;  // set bits per strip of the brick form (per tile of 1024 words of the dense bitmap)
  pcp::DevBuf<uint32_t> v_occ;     // brick form: one bit per brick place
  pcp::DevBuf<int32_t> v_rank;     // brick form: occupied places before each word of v_occ
  pcp::DevBuf<unsigned long long> v_plane;  // brick form: voxels per plane ix
  pcp::DevBuf<int64_t> v_vox;      // occupied voxels (linear index) in key order
  pcp::DevBuf<float> mls_xyz, mls_normal, mls_curv;
  pcp::DevBuf<int32_t> mls_index;
  // second set: pcp_cloud_smooth compacts the survivors of its last SOR into it and swaps the sets
  pcp::DevBuf<float> mls_alt_xyz, mls_alt_normal, mls_alt_curv;
  pcp::DevBuf<int32_t> mls_alt_index;
  int64_t mls_count = 0;
  // pcp_mls_stream_*: the plan of a chunked VOXEL_GRID_DILATION emission (pcp_mls.hip VgdStream; word0, word1, count per chunk)
  std::vector<uint8_t> vgd_blob;
  std::vector<int64_t> vgd_chunks;
  int64_t vgd_next = -1;
  // pcp_cloud_smooth_stream_*: the whole chain with the trailing outlier removal over the chunked emission (pcp_mls.hip
  // SmoothStream; per chunk: first plane, last plane, voxels, first result row, result rows)
  std::vector<uint8_t> css_blob;
  std::vector<int64_t> css_chunks;
  int64_t css_next = -1;
  pcp::DevBuf<float> css_dist;      // mean kNN distance of EVERY row of the dilated cloud (4 B x ~3.8e9 at C3)
  pcp::DevBuf<float> s_kth;         // per row of a chunk: bound of the squared distance to its (k + 1)-th nearest
  pcp::DevBuf<uint32_t> css_words;  // device scalars of the stream (max displacement, margins, counts)
  bool css_building = false;        // inside pcp_cloud_smooth_stream_begin
  double css_ball = 0.0;            // the trailing filter's ball, in (k + 1) rows by the voxel structure's density bound, as the last chunks left it (0: default)
  double sor_redo_fraction = 0.0;  // diagnostic: share of points the SOR selection kernel handed to the heap kernel

  // NID stage (section 8 f1): per-point intensity, per-keyframe culled clouds in camera
  // coordinates (x, y, z, intensity), chunked so that a workgroup sees one keyframe
  pcp::DevBuf<float> intensity;
  bool have_intensity = false;
  pcp::DevBuf<float> nid_pts;        // float4 per entry, NaN intensity = padding
  pcp::DevBuf<int32_t> nid_chunk_kf; // keyframe of every chunk
  pcp::DevBuf<double> nid_hist;      // [keyframe][bins*bins*7 + bins]
  int64_t nid_chunks = 0, nid_points = 0;
  int32_t nid_frames = 0;
  int32_t nid_hist_bins = 0;  // bins of the histograms pcp_nid_accumulate left in nid_hist

  // measurement
  bool timing = false;
  pcp::TimingSlot slots[PCP_K_COUNT];
  std::vector<pcp::PendingEvent> pending;
  std::vector<hipEvent_t> event_pool;
};

namespace pcp {

int set_error(const pcp_context *ctx, int code, const char *fmt, ...);
// one per translation unit with kernels: forces the runtime to load that unit's code object (pcp_context.hip preload_code_objects)
hipError_t preload_colour();
hipError_t preload_mls();
hipError_t preload_nid();
hipError_t preload_hpr();
void set_global_error(const char *fmt, ...);

#define PCP_HIP_TRY(ctx, expr)                                                                       \
  do {                                                                                               \
    hipError_t e__ = (expr);                                                                         \
    if (e__ != hipSuccess)                                                                           \
      return pcp::set_error((ctx), e__ == hipErrorOutOfMemory ? PCP_ERR_NOMEM : PCP_ERR_DEVICE,       \
                            "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
  } while (0)

// RAII bracket: records start/stop events around a launch when timing is on.
struct LaunchTimer {
  pcp_context *ctx;
  PendingEvent ev{};
  bool active = false;
  LaunchTimer(pcp_context *c, int32_t kernel);
  ~LaunchTimer();
};

int drain_timing(pcp_context *ctx);

// make ctx->stream wait for the asynchronous uploads of keyframes [f0, f1) that are still in flight, and note that
// the compute stream is about to touch the texel buffer (pcp_colour.hip)
int wait_images(pcp_context *ctx, int32_t f0, int32_t f1);

// ordered compaction of a device byte-flag array (pcp_colour.hip): index list (nullable) + count
int compact_flags(pcp_context *ctx, const uint8_t *flags, int64_t n, int32_t *out_index, int64_t capacity,
                  int64_t *count);

// ViewCulling::cull of one keyframe on the device: ordered index list of kept points (pcp_colour.hip)
int cull_frame_indices(pcp_context *ctx, int32_t frame, int32_t *d_index, int64_t capacity, int64_t *count);

// hidden_points_removal's hull over the candidate flags of one keyframe (pcp_hpr.hip): flags (input order, device)
// in: 1 = candidate; out: 1 = hull vertex
int hpr_run(pcp_context *ctx, int32_t frame, uint8_t *d_flags, uint32_t *hull_plane, uint32_t bit);
// the hulls of keyframes [f0, f1) into the (cleared) whole-run bits, `lanes` keyframes in flight on streams of their own
int hpr_run_range(pcp_context *ctx, int32_t f0, int32_t f1, int32_t lanes, const uint32_t *tile_mask = nullptr);

inline int64_t div_up(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace pcp
