/*
 * pcp_hip.h -- C ABI of libpcp_hip.so: the MI355X (gfx950) implementation of
 * PointCloudProcessor's colourisation / view-culling / MLS hot path.
 *
 * The reference (ChunLI-666/PointCloudProcessor @2024_10_08) has no FFI or
 * plugin interface: the path is reached through C++ member calls.  Every entry
 * point below names the reference call site it replaces (PCP/ =
 * PointCloudProcessor/ in the reference tree); INTEGRATION.md shows the shim a
 * maintainer adds at each site.
 *
 * Conventions
 *  - plain C types, caller-owned host buffers, no exceptions across the
 *    boundary: every call returns PCP_OK (0) or a negative error class and
 *    pcp_last_error() holds the message (the C++ shim rethrows
 *    std::runtime_error so that main.cpp:64-68 still maps it to exit code -2);
 *  - the library owns all device memory behind the opaque handle; no host
 *    pointer is retained after a call returns;
 *  - one calling thread per handle (the reference makes every hot-path call
 *    from its main thread, PointCloudProcessor.cpp:1007-1032);
 *  - there is no CPU fallback: without a usable HIP device pcp_create() fails.
 *  - results: pixel / cell indices, depth maps and keep masks are bit-exact
 *    with the CPU restatement in oracle/; colours and MLS outputs within 1e-4
 *    relative.
 */
#ifndef PCP_HIP_H
#define PCP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCP_ABI_VERSION 6 /* 2: pcp_cull_params grew cull_mode / match_mode; pcp_set_image_adjust
                             3: PCP_CULL_HPR, pcp_cull_params.hpr_flip_radius, pcp_hpr_stats
                             4: entry points added, no layout changed: pcp_sor_partial / pcp_sor_finish /
                                pcp_sor_chunk_points, pcp_hull_flags_import; PCP_DEPTH_BATCHED accepts PCP_CULL_HPR
                             5: no entry point or layout changed; pcp_cull_frame's out_keep, pcp_sor_partial's out_chunk_sums,
                                pcp_sor_finish's all_chunk_sums / out_keep may be DEVICE memory of the context's GPU (the
                                multi-GPU host exchanges them with RCCL instead of through the host)
                             6: entry points added: pcp_cloud_smooth_stream_begin / _next / _end / _stats (the whole enableMLS chain
                                with its trailing outlier removal over a chunked voxel dilation); pcp_hpr_stats reports
                                candidates = -1 after a call served from the whole-run bits */

#define PCP_OK 0
#define PCP_ERR_INVALID (-1) /* bad argument */
#define PCP_ERR_STATE (-2)   /* call order: camera / cloud / frames / images missing */
#define PCP_ERR_DEVICE (-3)  /* HIP runtime failure */
#define PCP_ERR_NOMEM (-4)   /* host or device allocation failed */
#define PCP_ERR_RANGE (-5)   /* frame index or capacity out of range */

typedef struct pcp_context pcp_context;

/* PCP/include/FrameData.hpp:9-12 (Pose); odometry line "ts x y z qw qx qy qz",
 * PCP/src/PointCloudProcessor.cpp:970-978. */
typedef struct pcp_pose {
  double x, y, z, qw, qx, qy, qz;
} pcp_pose;

/* K_camera / D_camera, PCP/src/PointCloudProcessor.cpp:57-62; image size used by
 * generateColorMap (:754); ViewCulling image_size hard-coded {4096,3000} (:206,:525). */
typedef struct pcp_camera {
  double fx, fy, cx, cy;
  double k1, k2, p1, p2, k3;
  int32_t image_width, image_height;
  int32_t cull_width, cull_height;
} pcp_camera;

/* Which of ViewCulling's two routines decides the candidates (pcp_cull_params.cull_mode). */
#define PCP_CULL_ZBUFFER 0 /* ViewCulling::view_culling, view_culling.cpp:52-174 (the routine north_star names; its
                              call is commented out at :43) */
#define PCP_CULL_HPR_CANDIDATES 1 /* ONLY the candidate filter of ViewCulling::hidden_points_removal, view_culling.cpp:
                              276-288: z > 0 and 0 <= (int)u < cull_width and 0 <= (int)v < cull_height, every candidate
                              kept -- a frustum cull, no occlusion test.  It is a SUPERSET of what the reference binary
                              keeps: at map density the hull below drops most of these candidates (C3 scene, 10 M points:
                              between 3 % and 90 % of a keyframe's candidates, profiles/r03_hpr_retention.json). */
#define PCP_CULL_HPR 2 /* ViewCulling::hidden_points_removal, view_culling.cpp:266-334, the routine the reference binary
                              calls (:46): the candidates above, flipped about a sphere of radius hpr_flip_radius
                              (:291-292), the origin appended (:297); visible = the vertices of the convex hull of that
                              set other than the origin (qhull, :302-329).  Computed on the device, per candidate, with
                              checked certificates (csrc/pcp_hpr.hip); the keep set is the exact set of extreme points,
                              from which qhull's differs by the points within its round-off (~1e-10 m) of a facet.  Kept
                              points are reported in input order (the reference lists them in qhull's vertex order). */

/* How a visible sample is credited to map points (pcp_cull_params.match_mode), PointCloudProcessor.cpp:554-592. */
#define PCP_MATCH_IDENTITY 0 /* the sample of point i is credited to point i, scores from the transform output p_c
                              (SURVEY.md Appendix B3 "identity mode"; ~2 % faster steps) */
#define PCP_MATCH_ROUNDTRIP 1 /* (default) the reference's arithmetic: p_w = c2w p_c in fp32 (:555), the sample is dropped unless
                              |p_w - p_i|^2 < f32(1e-5^2) in fp32 (what radiusSearch(1e-5) tests for point i itself,
                              :571), scores from p_c' = c2w.inverse() p_w in fp32 (:578-579).  Samples that the
                              reference's kd-tree would ALSO credit to other map points closer than 10 um to p_w are
                              not replicated (needs map points < ~20 um apart). */

/* vlcal::ViewCullingParams, PCP/include/vlcal/calib/view_culling.hpp:10-19, plus
 * the constants 14 (view_culling.cpp:63) and 0.05 (:157). */
typedef struct pcp_cull_params {
  int32_t enable_depth_buffer_culling;
  int32_t downsample_factor;
  double depth_slack;
  int32_t cull_mode;  /* PCP_CULL_ZBUFFER (default) / PCP_CULL_HPR_CANDIDATES / PCP_CULL_HPR */
  int32_t match_mode; /* PCP_MATCH_ROUNDTRIP (default) / PCP_MATCH_IDENTITY */
  double hpr_flip_radius; /* ViewCullingParams::hidden_points_removal_max_z = 90000 (view_culling.hpp:14) */
} pcp_cull_params;

/* MLSParameters, PCP/include/cloudSmooth.hpp:21-36; values
 * PCP/src/PointCloudProcessor.cpp:67-86. */
typedef struct pcp_mls_params {
  double search_radius;
  double sqr_gauss_param;
  int32_t polynomial_order;
  int32_t compute_normals;
  int32_t upsampling; /* 0 NONE, 3 VOXEL_GRID_DILATION (cloudSmooth.hpp enum order) */
  int32_t vgd_iterations;
  float vgd_voxel_size;
  int32_t sor_mean_k;   /* 60  (PointCloudProcessor.cpp:84) */
  double sor_std_mul;   /* 0.7 (PointCloudProcessor.cpp:86) */
} pcp_mls_params;

/* kernel ids for pcp_timing_get() */
enum {
  PCP_K_PROJECT = 0,    /* single-frame projection (the roofline kernel) */
  PCP_K_DEPTH = 1,      /* batched z-buffer MIN pass */
  PCP_K_COLOUR = 2,     /* batched visibility + colour + score + top-5 pass */
  PCP_K_VISIBILITY = 3, /* single-frame keep mask */
  PCP_K_MLS_GRID = 4,   /* MLS cell binning / counting sort */
  PCP_K_MLS_FIT = 5,    /* MLS radius search + polynomial fit + projection */
  PCP_K_MISC = 6,       /* fills, compaction, permutation */
  PCP_K_SOR = 7,        /* StatisticalOutlierRemoval kNN mean distance */
  PCP_K_MLS_VOXEL = 8,  /* VOXEL_GRID_DILATION upsampling */
  PCP_K_TILE_MASK = 9,  /* tile x keyframe visibility masks (conservative culling) */
  PCP_K_NID = 10,       /* NID joint histograms (value + SE(3) tangent gradient) */
  PCP_K_HPR = 11,       /* hidden_points_removal: flip, binning, per-candidate hull membership */
  PCP_K_COUNT = 12
};

/* ---- lifecycle ---------------------------------------------------------- */
int pcp_abi_version(void);
/* device: HIP device ordinal.  Fails (PCP_ERR_DEVICE) when no GPU is usable. */
int pcp_create(int32_t device, pcp_context **out);
void pcp_destroy(pcp_context *ctx);
/* message of the last failing call on ctx (or of pcp_create when ctx == NULL) */
const char *pcp_last_error(const pcp_context *ctx);
/* run on an externally owned hipStream_t (e.g. torch's current stream); NULL = own stream */
int pcp_set_stream(pcp_context *ctx, void *hip_stream);
int pcp_synchronize(pcp_context *ctx);

/* ---- configuration ------------------------------------------------------ */
void pcp_default_camera(pcp_camera *cam);
void pcp_default_cull_params(pcp_cull_params *p);
void pcp_default_mls_params(pcp_mls_params *p);
/* replaces create_camera + ViewCulling ctor, PointCloudProcessor.cpp:522-525 */
int pcp_set_camera(pcp_context *ctx, const pcp_camera *cam, const pcp_cull_params *cull);

/* ---- cloud -------------------------------------------------------------- */
/* SoA fp32 upload of the map (`cloud`, PointCloudProcessor.cpp:148). */
int pcp_upload_cloud(pcp_context *ctx, const float *x, const float *y, const float *z, int64_t n);
/* AoS upload straight from pcl::PointCloud<PointXYZI>::points.data(): x,y,z are
 * the first three floats of every `stride_bytes` record (32 for PointXYZI). */
int pcp_upload_cloud_aos(pcp_context *ctx, const void *points, int64_t n, int64_t stride_bytes);
int64_t pcp_cloud_size(const pcp_context *ctx);

/* ---- frames ------------------------------------------------------------- */
/* Host helper: pose -> (w2c, c2w) 3x4 row-major fp32, PointCloudProcessor.cpp:495-519.
 * T_opt: NULL, or a 4x4 row-major fp64 T_camera_lidar_optimized (:504-519). */
int pcp_pose_to_matrices(const pcp_pose *pose, const double *T_opt, float w2c[12], float c2w[12]);
/* Keyframe poses of the run (selectKeyframes output).  T_opt: NULL, 16 doubles
 * (stride 0, NID result) or n_frames*16 (stride 16, per-keyframe manual guess). */
int pcp_set_frames(pcp_context *ctx, const pcp_pose *poses, int32_t n_frames, const double *T_opt,
                   int32_t T_opt_stride);
int32_t pcp_frame_count(const pcp_context *ctx);
/* Decoded BGR8 image of one keyframe (what generateColorMap holds after the HSV
 * round trip, PointCloudProcessor.cpp:716-741), image_height rows of image_width
 * pixels, row_stride_bytes apart (cv::Mat::step). */
int pcp_upload_image(pcp_context *ctx, int32_t frame, const uint8_t *bgr, int64_t row_stride_bytes);
/* The same without waiting for the copy: returns once the transfer is queued on the context's stream.  The
 * host buffer must stay valid and unchanged until pcp_synchronize (or any synchronising call) returns; from
 * pinned memory a sequence of keyframes streams at the PCIe rate with the packing kernels in between. */
int pcp_upload_image_async(pcp_context *ctx, int32_t frame, const uint8_t *bgr, int64_t row_stride_bytes);
/* `count` keyframes first_frame ... that sit one after the other in PINNED host memory, frame_stride_bytes apart (a pinned
 * arena the decoder fills, cv::Mat headers over it): the copy engine moves them in blocks of <= 128 MB (one DMA per block:
 * the PCIe rate, which neither one copy per keyframe nor kernels reading pinned memory in place reach), each block's
 * keyframes are packed from the device copy, consecutive blocks alternate between two streams.  Asynchronous like
 * pcp_upload_image_async.  Pageable or device memory is accepted and handled keyframe by keyframe. */
int pcp_upload_images_block(pcp_context *ctx, int32_t first_frame, int32_t count, const uint8_t *bgr, int64_t row_stride_bytes,
                            int64_t frame_stride_bytes);
/* Both upload calls also take a DEVICE pointer for `bgr` (e.g. frames broadcast or all-gathered over xGMI by a
 * multi-GPU host): the transfer is then ordered after everything already queued on the context's stream
 * (pcp_set_stream), so a collective that produced the bytes on that stream needs no host synchronisation.  Pinned
 * (device-mapped) host memory is read in place by the pack kernel; pageable memory goes through a staging copy. */
/* The image adjustment generateColorMap applies to every keyframe before sampling it
 * (PointCloudProcessor.cpp:722-741): cv::cvtColor(BGR2HSV) on 8-bit pixels, S and V multiplied by
 * saturation_scale / brightness_scale (both 1.0 in the reference, :728-729) with saturate_cast<uchar>, and
 * cv::cvtColor(HSV2BGR).  8-bit HSV is lossy, so this is not the identity even at scale 1.0.
 * enable != 0: images handed to pcp_upload_image / _async afterwards are RAW decoded pixels (cv::imread output)
 * and the library applies the round trip while packing them (fused into the pack kernel, no extra pass).
 * enable == 0 (default): the caller passes what generateColorMap holds AFTER its own cvtColor calls.
 * Arithmetic: the forward half is OpenCV 4.2's integer routine (RGB2HSV_b, hdiv / sdiv tables, h range 180),
 * restated exactly; the backward half is OpenCV 4.2's scalar float routine (HSV2RGB_native, no FMA), whose
 * SIMD / FMA builds may differ from it by one level on some pixels [upstream, parity unpinned: DESIGN.md]. */
int pcp_set_image_adjust(pcp_context *ctx, int32_t enable, float saturation_scale, float brightness_scale);
/* diagnostic: the pixels of one keyframe as the kernels sample them (after pcp_set_image_adjust's round trip when it
 * was enabled at upload): out_bgr image_height*image_width*3 tightly packed, out_mask image_height*image_width
 * (either nullable) */
int pcp_download_image(pcp_context *ctx, int32_t frame, uint8_t *out_bgr, uint8_t *out_mask);
/* gray8 segmentation mask (cv::IMREAD_GRAYSCALE, PointCloudProcessor.cpp:775) */
int pcp_upload_mask(pcp_context *ctx, int32_t frame, const uint8_t *gray, int64_t row_stride_bytes);

/* ---- single keyframe (drop-in for the calls inside the per-keyframe loop) -- */
/* transformPointCloud + project (PointCloudProcessor.cpp:521, view_culling.cpp:86-90,
 * PointCloudProcessor.cpp:748-754).  All outputs nullable, length n, input order:
 *   out_cell  z-buffer cell cy*mw+cx; -1 rejected; -2 = candidate outside the /14 map,
 *             reported only when enable_depth_buffer_culling == 0 (where it is kept)
 *   out_pixel colour pixel v*image_width+u; -1 rejected
 *   out_range f32(||p_c||), valid where out_cell != -1 (FLT_MAX elsewhere)
 *   out_xyz_cam 3*n floats, SoA (x[n] y[n] z[n]) camera coordinates
 * With every output NULL the kernel still runs and leaves cell/range on the
 * device (used by bench.py to time the kernel without PCIe traffic). */
int pcp_project_frame(pcp_context *ctx, int32_t frame, int32_t *out_cell, int32_t *out_pixel, float *out_range,
                      float *out_xyz_cam);
/* ViewCulling::cull with the z-buffer routine (view_culling.cpp:23-50,52-174).
 * out_keep n bytes (nullable), out_depth_map (H/14)*(W/14) floats (nullable). */
int pcp_cull_frame(pcp_context *ctx, int32_t frame, uint8_t *out_keep, int64_t *out_kept, float *out_depth_map);
/* cull + generateColorMap + generateSegmentMap + transform to world
 * (PointCloudProcessor.cpp:527-551): the kept and coloured points of one keyframe
 * in input order.  rgb after the mask==255 -> (255,0,0) override when a mask was
 * uploaded.  All outputs nullable; capacity in points; *out_count = true count. */
int pcp_frame_visible(pcp_context *ctx, int32_t frame, int64_t capacity, int32_t *out_index, uint8_t *out_rgb,
                      uint16_t *out_mask, float *out_xyz_cam, float *out_xyz_world, int64_t *out_count);

/* diagnostic: the last hidden_points_removal run on ctx (PCP_CULL_HPR; the latest keyframe of a batched call):
 * out[0] visible, [1] hidden (both counted from the final verdicts), [2] candidates that went to the exact path (neither
 * floating-point certificate held), [3] trial normals, [4] batches of 64 point tests -- of the polygon and exact searches; the
 * 16-lane passes in front of them, which settle nearly every candidate, count theirs only under PCP_HPR_DEBUG=1 (the counting
 * was 9 % of a hull pass) --, [5] reserved (0), [6] UNRESOLVED (no exact
 * certificate either: exactly degenerate input such as four coplanar flipped points; classified hidden), [7] exact
 * predicate evaluations, [8] grid cells, [9] candidates.
 * After a single-keyframe call (pcp_cull_frame, pcp_frame_visible, NID) that was SERVED FROM THE WHOLE-RUN BITS -- a
 * pcp_depth_pass of this context had taken the keyframe's hull already, or pcp_hull_flags_import had brought it -- nothing
 * was recomputed and there are no tallies of that keyframe: out[0..8] = 0 and out[9] = -1. */
int pcp_hpr_stats(pcp_context *ctx, int64_t out[10]);

/* ---- whole run (pcdColorizationAndSmooth, PointCloudProcessor.cpp:474-602) -- */
/* z-buffer MIN pass for keyframes [frame_begin, frame_end) over the local points */
int pcp_depth_pass(pcp_context *ctx, int32_t frame_begin, int32_t frame_end);
/* device address of the n_frames*(H/14)*(W/14) fp32 depth maps, for the
 * all-reduce(MIN) across point shards (multi-GPU), and its length in floats */
int pcp_depth_maps_device(pcp_context *ctx, void **device_ptr, int64_t *n_floats);
int pcp_download_depth_map(pcp_context *ctx, int32_t frame, float *out_depth_map);
/* Where the single-keyframe calls (pcp_cull_frame, pcp_frame_visible, pcp_nid_prepare) take a keyframe's depth map
 * from.  PCP_DEPTH_OWN (default): each call builds it from the uploaded points, as ViewCulling::view_culling does.
 * PCP_DEPTH_BATCHED: they use the maps pcp_depth_pass left behind -- for a context that holds one index shard of the
 * map, after the caller's all-reduce(MIN) across shards, these ARE the maps of the whole cloud, so the per-keyframe
 * outputs of the shards concatenate to the single-GPU result.  The keyframe must have been covered by pcp_depth_pass. */
#define PCP_DEPTH_OWN 0
#define PCP_DEPTH_BATCHED 1
int pcp_set_depth_source(pcp_context *ctx, int32_t source);
/* PCP_CULL_HPR over index shards: a keyframe's hull is taken over EVERY candidate of the map (view_culling.cpp:291-329), so
 * a context that holds one shard (PCP_DEPTH_BATCHED) cannot decide its points.  The verdicts come from a context that holds
 * the whole map (pcp_cull_frame there returns them, input order) and are handed to the shard here: keep[i] != 0 = point i
 * of THIS context is a hull vertex of `frame` (n flags, host or device memory).  pcp_colour_pass and the single-keyframe
 * calls of the shard then read them where the z-buffer routine reads the merged depth maps. */
int pcp_hull_flags_import(pcp_context *ctx, int32_t frame, const uint8_t *keep);
int pcp_colour_reset(pcp_context *ctx);
/* visibility + colour lookup + scores + per-point top-5 for [frame_begin, frame_end) */
int pcp_colour_pass(pcp_context *ctx, int32_t frame_begin, int32_t frame_end);
/* smoothColors + removePointsWithNoColor flag (PointCloudProcessor.cpp:604-631,
 * hpp:238-252).  out_rgb n*3 (r,g,b), out_has n; optional: out_count n (#views),
 * out_top_score/out_top_rgb(0x00RRGGBB)/out_top_frame n*5 (desc, -1 padded). */
int pcp_colour_finalise(pcp_context *ctx, uint8_t *out_rgb, uint8_t *out_has, int32_t *out_count,
                        float *out_top_score, uint32_t *out_top_rgb, int32_t *out_top_frame);
/* reset + depth_pass(all) + colour_pass(all) + finalise */
int pcp_colorize(pcp_context *ctx, uint8_t *out_rgb, uint8_t *out_has);
/* visibility + colour + top-5 + smoothColors in one launch over ALL keyframes,
 * given depth maps that already cover them (after pcp_depth_pass and, across
 * point shards, the all-reduce(MIN)).  out_* nullable. */
int pcp_colorize_from_depth(pcp_context *ctx, uint8_t *out_rgb, uint8_t *out_has);
/* packed result r | g<<8 | b<<16 | has<<24, n words, one plain device-to-host copy */
int pcp_download_result_packed(pcp_context *ctx, uint32_t *out_rgba);
/* the same copy on the library's copy stream: returns at once, the result buffer is
 * double-buffered so the next run's kernels overlap this transfer; out_rgba (pinned
 * memory for a real overlap) is valid after pcp_synchronize(). */
int pcp_download_result_packed_async(pcp_context *ctx, uint32_t *out_rgba);
/* Blocks the host until the asynchronous download issued BEFORE the latest one has landed (no-op when there is
 * none): the consumer of a two-buffer ring calls it before reusing the older buffer.  It does not wait for any
 * kernel, so the device stays busy, and it keeps the host at most one run ahead of the device -- an unbounded
 * lead (hundreds of queued commands) was measured to slow the steps by 13 %. */
int pcp_download_wait_previous(pcp_context *ctx);
/* device address of the packed per-point result (r | g<<8 | b<<16 | has<<24),
 * valid after pcp_colour_finalise / pcp_colorize, for device-side gathers */
int pcp_colour_result_device(pcp_context *ctx, void **device_ptr, int64_t *n_words);

/* ---- MLS (CloudSmooth::process, PCP/src/cloudSmooth.cpp:77-185) ---------- */
/* pcl::MovingLeastSquares on the uploaded cloud (radius search + order-2 fit +
 * SIMPLE projection; upsampling NONE or VOXEL_GRID_DILATION).  Results stay on
 * the device; *out_count = number of output points. */
int pcp_mls_process(pcp_context *ctx, const pcp_mls_params *p, int64_t *out_count);
/* Multi-GPU form: the whole cloud is uploaded on every rank and this rank fits
 * only the queries index_begin <= i < index_end (upsampling NONE).  Outputs as pcp_mls_process. */
int pcp_mls_process_shard(pcp_context *ctx, const pcp_mls_params *p, int64_t index_begin, int64_t index_end,
                          int64_t *out_count);
/* The same deal by SLABS of the stage's own spatial order (consecutive places of its cell-sorted cloud, whole wavefronts):
 * slab `slab` of `n_slabs` is 1 / n_slabs of the work whatever order the caller's points come in (an index range only
 * divides the work when the caller's order is spatially coherent).  Outputs as pcp_mls_process: the slab's rows in input
 * order; the slabs' results merged by source index are the unsharded result. */
int pcp_mls_process_slab(pcp_context *ctx, const pcp_mls_params *p, int32_t slab, int32_t n_slabs, int64_t *out_count);
/* VOXEL_GRID_DILATION in chunks.  One result holds fewer than 2^31 points and must fit the device (78 B per point); the
 * reference's own configuration (1 mm voxels, 4 dilations, PointCloudProcessor.cpp:78-81) turns every input point into up
 * to 729 output points -- ~3.8e9 for a 10 M-point map -- and pcp_mls_process then fails with PCP_ERR_NOMEM.  The
 * streamed form fits the surfaces once, counts the dilated voxel set in 64 bits (*out_total) and cuts its ascending key
 * order into *out_chunks chunks of at most chunk_capacity voxels; every pcp_mls_stream_next emits the next chunk into
 * the result buffers (read with pcp_mls_fetch; *out_count = its points, 0 after the last chunk).  The chunks in order
 * are exactly what one pcp_mls_process would return.  No other call on ctx may come between begin and the last next. */
int pcp_mls_stream_begin(pcp_context *ctx, const pcp_mls_params *p, int64_t chunk_capacity, int64_t *out_total,
                         int32_t *out_chunks);
int pcp_mls_stream_next(pcp_context *ctx, int64_t *out_count);
/* The chunk the next pcp_mls_stream_next emits (0 .. chunks; a chunk may be emitted more than once): several GPUs that
 * hold the same cloud and began the same stream deal the chunks out among themselves (chunk c to GPU c mod N). */
int pcp_mls_stream_seek(pcp_context *ctx, int32_t chunk);
/* xyz / normal 3*m floats AoS, curvature m, source index m (input order for
 * NONE; ascending voxel key for VOXEL_GRID_DILATION). */
int pcp_mls_fetch(pcp_context *ctx, int64_t capacity, float *out_xyz, float *out_normal, float *out_curvature,
                  int32_t *out_index);
/* CloudSmooth::process end to end on the device: SOR(sor_mean_k, sor_std_mul) ->
 * MovingLeastSquares (+ upsampling) -> SOR, cloudSmooth.cpp:109-164.  Results through
 * pcp_mls_fetch; out_index refers to the uploaded cloud. */
int pcp_cloud_smooth(pcp_context *ctx, const pcp_mls_params *p, int64_t *out_count);

/* CloudSmooth::process WHOLE for clouds whose dilated voxel set exceeds one result (PCP/src/cloudSmooth.cpp:109-164 with the
 * reference's own MLS configuration, PointCloudProcessor.cpp:67-86: VOXEL_GRID_DILATION 1 mm x 4 makes ~3.8e9 points of a
 * 10 M-point map): StatisticalOutlierRemoval -> MovingLeastSquares + upsampling -> StatisticalOutlierRemoval ON THE UPSAMPLED
 * CLOUD (:160-164), the last two stages streamed over chunks of the voxel key order (whole planes of the first axis).
 *   _begin: first filter, fit, voxel set; sweep 0 projects a sample of the voxels to size the halo; then sweep 1 -- every chunk
 *           is emitted together with a halo of neighbouring planes, the mean k-NN distances of its own rows are computed against
 *           chunk + halo and kept on the device (4 B per row of the whole upsampled cloud), (sum, sum of squares) are taken over
 *           ALL rows in row order, the filter's threshold follows.  The halo is CHECKED, not assumed: a row's neighbourhood (bound
 *           of the distance to its (k + 1)-th nearest) must end inside the part of space whose rows the halo is guaranteed to
 *           hold, given the largest displacement any row has from its voxel (taken over every row sweep 1 emitted -- all of them);
 *           a chunk that fails is redone with a wider halo.  The distances are therefore the ones the one-shot pcp_cloud_smooth
 *           computes, bit for bit; the threshold is summed in another order (row order instead of the cell order of one big
 *           grid) and agrees to rounding (~1e-16 relative).
 *           out_total_rows = rows of the upsampled cloud before the last filter, out_kept_rows = after it.
 *   _next:  sweep 2 -- the next chunk's own rows are emitted again, classified by their stored distance and compacted:
 *           *out_count survivors in key order, fetched with pcp_mls_fetch (out_index refers to the uploaded cloud); 0 after
 *           the last chunk.  The concatenation over the chunks is what pcp_cloud_smooth returns when the cloud fits one result.
 *   _end:   ends the stream and frees the distances (4 B per row: 11 GB for a 10 M-point map).  Optional: the next stream of
 *           the context reuses them, a one-shot upsampling call that would not fit without them takes them, pcp_destroy frees them.
 * chunk_capacity: most voxels per chunk, own rows (>= 4096; every plane of the voxel grid must fit). */
int pcp_cloud_smooth_stream_begin(pcp_context *ctx, const pcp_mls_params *p, int64_t chunk_capacity, int64_t *out_total_rows,
                                  int64_t *out_kept_rows, int32_t *out_chunks);
int pcp_cloud_smooth_stream_next(pcp_context *ctx, int64_t *out_count);
int pcp_cloud_smooth_stream_end(pcp_context *ctx);
/* diagnostic of the last pcp_cloud_smooth_stream_begin: out[0] halo in planes (as finally used, the widest), [1] chunks redone
 * with a wider halo, [2] threshold of the last filter, [3] largest |x displacement| of a row from its voxel (m; over all rows),
 * [4] smallest margin of any chunk (m; > [3] proves the halo), [5] rows computed including halos, [6] the displacement sweep 0's
 * sample saw (m; sized the halo), [7] bytes of device memory the stream holds at the time of the call, [8..11] host-clock
 * seconds of _begin: first filter + fit + voxel set, [9] device allocations (hipMalloc / hipFree: they lie inside the other
 * three; seconds on a first call, none afterwards), sweep 0, sweep 1 + threshold, [12] bytes allocated during _begin. */
int pcp_cloud_smooth_stream_stats(pcp_context *ctx, double out[13]);
/* pcl::StatisticalOutlierRemoval (k, std_mul) keep mask of the uploaded cloud,
 * cloudSmooth.cpp:109-116,160-164.
 * The smoothing entry points (pcp_sor, pcp_mls_process[_shard], pcp_cloud_smooth, pcp_close_pairs) need finite
 * coordinates: a cloud with NaN or infinite points is refused with PCP_ERR_INVALID (PCL's filters skip such points one
 * by one; the projection / colour entry points accept them and reject the points, Appendix B6). */
int pcp_sor(pcp_context *ctx, int32_t mean_k, double std_mul, uint8_t *out_keep, int64_t *out_kept);
/* Multi-GPU form of pcp_sor (the cloud on every GPU, the queries dealt out).  The queries are dealt out by
 * SLABS of the filter's own spatial order (consecutive places of its cell-sorted cloud), not by the caller's indices: a
 * wavefront holds 64 consecutive places, so a slab is whole wavefronts and 1 / n_slabs of the work whatever order the
 * caller's points come in.  The filter's threshold is mean + std_mul * stddev of ALL mean distances (cloudSmooth.cpp:113-115 ->
 * statistical_outlier_removal.hpp [upstream]), so a slab cannot classify alone; the statistics are kept as one (sum, sum of
 * squares) pair per chunk of pcp_sor_chunk_points() consecutive places, each chunk summed in a fixed order by the GPU whose
 * slab holds it (slabs are whole chunks):
 *   pcp_sor_partial  mean distances of slab `slab` of `n_slabs` and its chunk sums: 2 doubles per chunk into out_chunk_sums,
 *                    *out_first_chunk / *out_chunks = where they belong among the ceil(n / chunk) pairs of the cloud;
 *   (the caller puts the slabs' arrays together: the array one GPU computes)
 *   pcp_sor_finish   threshold from all chunk sums, keep flags of the slab's points: out_keep[i] (n bytes, the caller's
 *                    indices, 0 for every point of another slab -- the OR of the slabs' arrays is the filter's mask).
 * pcp_sor itself is partial + finish with one slab: the sharded keep mask equals it bit for bit. */
int64_t pcp_sor_chunk_points(void);
int pcp_sor_partial(pcp_context *ctx, int32_t mean_k, int32_t slab, int32_t n_slabs, int64_t capacity, double *out_chunk_sums,
                    int64_t *out_first_chunk, int64_t *out_chunks);
int pcp_sor_finish(pcp_context *ctx, double std_mul, const double *all_chunk_sums, int64_t n_chunks, int32_t slab, int32_t n_slabs,
                   uint8_t *out_keep, int64_t *out_kept);

/* ---- NID extrinsic refinement (VisualLiDARCalibration::calibrate, PCP/src/calibrate.cpp:42-126) -- */
/* per-point intensity of the uploaded cloud (pcl::PointXYZI::intensity), needed by the NID stage */
int pcp_upload_intensity(pcp_context *ctx, const float *intensity, int64_t n);
/* Builds every keyframe's NID input on the device: its z-buffer-culled points in camera
 * coordinates with intensity (the content of <ts>_beforeNID.pcd, PointCloudProcessor.cpp:178-224).
 * Needs camera, cloud, intensity, keyframes and images. */
int pcp_nid_prepare(pcp_context *ctx, int64_t *out_points);
/* MultiNIDCost at T = T_camera_lidar (4x4 row-major fp64): sum over keyframes of NIDCost
 * (nid_cost.hpp:42-116), and its gradient in the SE(3) tangent of T * exp(delta),
 * delta = (upsilon, omega).  T_init (nullable): the initial guess whose +-0.2 m / 2 deg
 * neighbourhood bounds the domain (visual_camera_calibration.cpp:100-105); outside it, or when
 * a keyframe's cost is not finite, *valid = 0. */
int pcp_nid_evaluate(pcp_context *ctx, const double T[16], const double *T_init, int32_t bins, double *cost,
                     double grad6[6], int32_t *valid);
/* VisualCameraCalibration::calibrate (visual_camera_calibration.cpp:49-80): up to
 * max_outer_iterations runs of a BFGS minimisation on SE(3) (stands in for ceres::Solve),
 * stopping when the pose moves by < 1 cm and < 1 deg.  T_out feeds pcp_set_frames' T_opt. */
int pcp_nid_optimize(pcp_context *ctx, const double T_init[16], int32_t bins, int32_t max_outer_iterations,
                     double T_out[16], double *final_cost, int32_t *evaluations);

/* The same cost over an index-sharded map (one context per GPU, each holding a slice of the cloud and of the
 * intensities; keyframes and images replicated; depth maps MIN-merged, PCP_DEPTH_BATCHED).  A keyframe's joint histogram
 * (nid_cost.hpp:55-93) is a sum over its points, so: every shard accumulates its own histograms at T
 * (pcp_nid_accumulate), the shards' histograms are added in place -- an all-reduce(SUM) over the `count` doubles at
 * pcp_nid_histograms_device, queued on the context's stream or completed before the next call -- and pcp_nid_finish
 * turns the summed histograms into cost and gradient, the same numbers on every shard.  On one context
 * pcp_nid_evaluate == pcp_nid_accumulate + pcp_nid_finish (plus the domain test against T_init). */
int pcp_nid_accumulate(pcp_context *ctx, const double T[16], int32_t bins);
int pcp_nid_histograms_device(pcp_context *ctx, void **device_ptr, int64_t *count);
int pcp_nid_finish(pcp_context *ctx, int32_t bins, double *cost, double grad6[6], int32_t *valid);
/* pcp_nid_optimize with the cost supplied by the caller (the sharded evaluation above, driven by the host that owns the
 * shards): eval returns PCP_OK and fills cost / grad6 / valid for a T inside the domain; the loop, its tolerances and the
 * domain test are those of pcp_nid_optimize.  ctx only carries the error text. */
typedef int (*pcp_nid_eval_fn)(void *user, const double T[16], int32_t bins, double *cost, double grad6[6], int32_t *valid);
int pcp_nid_optimize_with(pcp_context *ctx, pcp_nid_eval_fn eval, void *user, const double T_init[16], int32_t bins,
                          int32_t max_outer_iterations, double T_out[16], double *final_cost, int32_t *evaluations);

/* ---- precondition of the match-back (PointCloudProcessor.cpp:480-482,571) ------------------------------- */
/* Number of map points that have ANOTHER map point closer than `radius` (fp32 squared distance, strict <, as
 * kdtree.radiusSearch compares).  The reference credits a visible sample to every map point within 1e-5 m of the
 * sample's fp32 world position; libpcp_hip credits the sample's own point only (PCP_MATCH_ROUNDTRIP / _IDENTITY).  The
 * two agree when no two map points can both lie within 1e-5 m of one sample: call this with radius = 2.5e-5 (the
 * match radius plus twice the largest fp32 round-trip error of maps within +-50 m) and expect 0; a non-zero count
 * (duplicated or near-duplicated points, e.g. un-deduplicated scan accumulations) names how many points may receive
 * a neighbour's samples in the reference and not here. */
int pcp_close_pairs(pcp_context *ctx, double radius, int64_t *points_with_close_neighbour);

/* ---- measurement -------------------------------------------------------- */
/* When enabled every kernel launch is bracketed by hipEvents on the context's
 * stream; totals are read back with pcp_timing_get (which synchronises). */
int pcp_timing_enable(pcp_context *ctx, int32_t on);
int pcp_timing_reset(pcp_context *ctx);
int pcp_timing_get(pcp_context *ctx, int32_t kernel_id, double *total_ms, int64_t *launches);
const char *pcp_kernel_name(int32_t kernel_id);
/* diagnostic: fraction of (tile, keyframe) pairs the conservative culling keeps (after pcp_depth_pass) */
int pcp_tile_mask_density(pcp_context *ctx, double *kept_fraction);
/* diagnostic: the (tile, keyframe) masks themselves, tiles x mask_words uint32 (bit f & 31 of word f >> 5); either
 * output may be NULL to query the sizes */
int pcp_tile_masks(pcp_context *ctx, int64_t *tiles, int32_t *mask_words, uint32_t *out_words);
/* diagnostic: share of the points of the last pcp_sor / pcp_cloud_smooth SOR pass that the selection kernel handed
 * to the heap kernel (fewer than mean_k + 1 neighbours within one grid cell, or a crowded boundary bin) */
int pcp_sor_redo_fraction(pcp_context *ctx, double *fraction);
/* diagnostic: the mean distance to the mean_k nearest neighbours of every uploaded point, as the last pcp_sor computed
 * it (the quantity StatisticalOutlierRemoval thresholds; statistical_outlier_removal.hpp [upstream] keeps it private).
 * Valid until the next call that smooths or filters; capacity >= the number of uploaded points. */
int pcp_sor_distances(pcp_context *ctx, int64_t capacity, float *out_distance);
/* diagnostic: the kernels replace three IEEE divisions of the projection (pinhole.hpp:17-18 x/z, y/z in fp64;
 * view_culling.cpp:88 u/14, v/14 in fp32) by shorter sequences that are proven to return the same correctly
 * rounded quotients (pcp_device.hpp).  This runs both forms on the device and counts disagreements:
 * `samples` pseudo-random (x, y, z) float triples (all exponents, z > 0) for the fp64 pair, and EVERY fp32 bit
 * pattern for the division by the configured downsample factor.  Both counts must be 0. */
int pcp_selftest_arithmetic(pcp_context *ctx, int64_t samples, uint64_t seed, int64_t *mismatches_fp64,
                            int64_t *mismatches_fp32);

#ifdef __cplusplus
}
#endif
#endif /* PCP_HIP_H */
