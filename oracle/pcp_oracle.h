/*
 * pcp_oracle.h -- CPU restatement of PointCloudProcessor's colourisation /
 * view-culling / MLS hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED: the reference (ChunLI-666/PointCloudProcessor @2024_10_08)
 * ships no tests, golden vectors or fixtures for this path and cannot be
 * built in this image (PCL, Eigen, OpenCV, qhull, Boost absent), so this
 * restatement is pinned only by (i) line-by-line reading of the cited
 * reference sources, (ii) an independent numpy twin (oracle/np_oracle.py)
 * and (iii) analytic known-answer tests.  Statements about upstream PCL /
 * Eigen internals are marked [upstream].
 *
 * Shorthand: PCP/ = /root/reference/PointCloudProcessor/.
 * All fp32 / fp64 operations are individually rounded: build with
 * -ffp-contract=off (see oracle/Makefile).
 */
#ifndef PCP_ORACLE_H
#define PCP_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* PCP/include/FrameData.hpp:9-12 */
typedef struct orc_pose {
  double x, y, z, qw, qx, qy, qz;
} orc_pose;

/* K/D: PCP/src/PointCloudProcessor.cpp:57-62; cull size :525; image size :754 */
typedef struct orc_camera {
  double fx, fy, cx, cy;
  double k1, k2, p1, p2, k3;
  int32_t image_width, image_height; /* actual image (generateColorMap bounds) */
  int32_t cull_width, cull_height;   /* ViewCulling image_size, ref {4096,3000} */
} orc_camera;

/* PCP/include/vlcal/calib/view_culling.hpp:10-19, view_culling.cpp:63,157 */
#define ORC_CULL_ZBUFFER 0        /* ViewCulling::view_culling, view_culling.cpp:52-174 */
#define ORC_CULL_HPR_CANDIDATES 1 /* candidate filter of hidden_points_removal, view_culling.cpp:276-288, keep all */
#define ORC_CULL_HPR 2            /* hidden_points_removal, view_culling.cpp:266-334: filter + hull (pcp_oracle_hpr.c) */
#define ORC_MATCH_IDENTITY 0      /* Appendix B3: the sample of point i goes to point i, scores from p_c */
#define ORC_MATCH_ROUNDTRIP 1     /* fp32 world round trip + the 10 um self-match test + scores from
                                     c2w.inverse() * p_w (PointCloudProcessor.cpp:555,571-579), no cross-credit */
typedef struct orc_cull_params {
  int32_t enable_depth_buffer_culling; /* ref: true */
  int32_t downsample_factor;           /* ref: 14 */
  double depth_slack;                  /* ref: 0.05 */
  int32_t cull_mode;                   /* ORC_CULL_* */
  int32_t match_mode;                  /* ORC_MATCH_* (orc_colorize; orc_colorize_faithful ignores it) */
  double hpr_flip_radius;              /* ref: 90000 (view_culling.hpp:14) */
} orc_cull_params;

/* PCP/include/cloudSmooth.hpp:21-36, values PCP/src/PointCloudProcessor.cpp:67-86 */
typedef struct orc_mls_params {
  double search_radius;     /* 0.03 */
  double sqr_gauss_param;   /* 0.0009 (unused by the PCL 1.10 fit, B13) */
  int32_t polynomial_order; /* 2 */
  int32_t compute_normals;  /* 1 */
  int32_t upsampling;       /* 0 NONE, 3 VOXEL_GRID_DILATION */
  int32_t vgd_iterations;   /* 4 */
  float vgd_voxel_size;     /* 0.001 */
  int32_t threads;          /* OpenMP threads, <=0: all */
} orc_mls_params;

void orc_default_camera(orc_camera *cam);
void orc_default_cull_params(orc_cull_params *p);
void orc_default_mls_params(orc_mls_params *p);

/* A1: pose -> (w2c, c2w) as 3x4 row-major fp32.  T_opt: optional 4x4 row-major
 * fp64 extrinsic correction (NID / manual branch, cpp:504-519). */
void orc_pose_to_matrices(const orc_pose *pose, const double *T_opt, float w2c[12], float c2w[12]);

/* A2: pcl::transformPointCloud, PCL 1.10 SSE association [upstream]. */
void orc_transform(const float m[12], const float *x, const float *y, const float *z, int64_t n,
                   float *xc, float *yc, float *zc);

/* A3: pinhole + plumb-bob, fp64. */
void orc_project_point(const orc_camera *cam, double xc, double yc, double zc, double *u, double *v);

/* A2+A3+A4(candidate test)+A5(pixel): per-point outputs, all nullable.
 *   out_cell : cy*mw+cx for z-buffer candidates inside the map, -1 rejected;
 *              -2 = candidate outside the downsampled map, reported only when
 *              enable_depth_buffer_culling == 0 (where such points are kept)
 *   out_pixel: vi*image_width+ui (colour lookup), -1 rejected
 *   out_range: f32(||p_c||) (valid where z_c > 0, else FLT_MAX)
 */
void orc_project_frame(const orc_camera *cam, const orc_cull_params *cp, const float w2c[12],
                       const float *x, const float *y, const float *z, int64_t n,
                       int32_t *out_cell, int32_t *out_pixel, float *out_range,
                       float *out_xc, float *out_yc, float *out_zc);

/* A4: z-buffer cull of one frame.  depth_map (mh*mw floats, nullable) receives
 * the final map; out_keep (n bytes).  Returns the number kept. */
int64_t orc_cull_frame(const orc_camera *cam, const orc_cull_params *cp, const float w2c[12],
                       const float *x, const float *y, const float *z, int64_t n,
                       uint8_t *out_keep, float *depth_map, int32_t threads);

/* A6: scores for one camera-frame point. */
void orc_scores(float xc, float yc, float zc, const orc_pose *pose, float *orientation, float *distance,
                float *final_score);

/* A1..A8 whole colourisation.  images: F pointers to tightly packed BGR8
 * image_height*image_width*3.  T_opt: NULL, or 16 doubles (global, stride 0),
 * or F*16 doubles (per keyframe, stride 16).
 * Outputs (nullable except out_rgb/out_has):
 *   out_rgb   n*3 (r,g,b), out_has n (1 iff rgb != 0,0,0, i.e. survives
 *   removePointsWithNoColor), out_count n (#views, capped at INT32_MAX),
 *   out_top_score n*5 (desc, -1 padded), out_top_rgb n*5 (0x00RRGGBB),
 *   out_top_frame n*5 (-1 padded).
 */
int orc_colorize(const orc_camera *cam, const orc_cull_params *cp, const float *x, const float *y,
                 const float *z, int64_t n, const orc_pose *poses, int32_t n_frames, const double *T_opt,
                 int32_t T_opt_stride, const uint8_t *const *images, uint8_t *out_rgb, uint8_t *out_has,
                 int32_t *out_count, float *out_top_score, uint32_t *out_top_rgb, int32_t *out_top_frame,
                 int32_t threads);

/* The same with the reference's own match-back (Appendix B3 "faithful mode"): fp32 world round trip,
 * radiusSearch(1e-5) over the original cloud (every match is credited, including other points and
 * none at all), scores from c2w.inverse() * p_w.  stats (nullable): {samples, without any match,
 * own point not matched, credits to other points}. */
int orc_colorize_faithful(const orc_camera *cam, const orc_cull_params *cp, const float *x, const float *y,
                          const float *z, int64_t n, const orc_pose *poses, int32_t n_frames, const double *T_opt,
                          int32_t T_opt_stride, const uint8_t *const *images, uint8_t *out_rgb, uint8_t *out_has,
                          int32_t *out_count, float *out_top_score, uint32_t *out_top_rgb, int32_t *out_top_frame,
                          int64_t *stats, int32_t threads);

/* Eigen Transform<float,3,Affine>::inverse() of a 3x4 row-major fp32 matrix (tests) */
void orc_affine_inverse_f32(const float m[12], float out[12]);

/* A5 mask branch + per-frame visible list (generateColorMap + generateSegmentMap,
 * cpp:531-551).  For one frame: for every kept & coloured point, in input order,
 * emits index, rgb (after the 255 -> (255,0,0) override when mask given), mask
 * value and camera / world coordinates.  mask may be NULL.  Returns count. */
int64_t orc_frame_visible(const orc_camera *cam, const orc_cull_params *cp, const orc_pose *pose,
                          const double *T_opt, const float *x, const float *y, const float *z, int64_t n,
                          const uint8_t *image, const uint8_t *mask, int32_t *out_index, uint8_t *out_rgb,
                          uint16_t *out_mask, float *out_xyz_cam, float *out_xyz_world);

/* A7: MovingLeastSquares, upsampling NONE, SIMPLE projection.
 * Outputs sized n: xyz (3n), normal (3n), curvature (n), src index (n).
 * Returns the number of output points (points with <3 neighbours are dropped),
 * in input order. */
int64_t orc_mls(const float *x, const float *y, const float *z, int64_t n, const orc_mls_params *p,
                float *out_xyz, float *out_normal, float *out_curv, int32_t *out_index);

/* A7.8: VOXEL_GRID_DILATION upsampling.  Call with out_* NULL to get the count. */
int64_t orc_mls_voxel_dilation(const float *x, const float *y, const float *z, int64_t n,
                               const orc_mls_params *p, int64_t capacity, float *out_xyz, float *out_normal,
                               float *out_curv, int32_t *out_index);
/* the same on the voxel lattice of a larger cloud the points are a part of (origin = its bounding_min_, extent = its largest
 * extent): test infrastructure for restating a REGION of a map */
int64_t orc_mls_voxel_dilation_part(const float *x, const float *y, const float *z, int64_t n, const orc_mls_params *p,
                                    const float *origin, double extent, int64_t capacity, float *out_xyz, float *out_normal,
                                    float *out_curv, int32_t *out_index);

/* f4: the 8-bit BGR -> HSV -> BGR round trip of generateColorMap (PCP/src/PointCloudProcessor.cpp:722-741),
 * OpenCV 4.2 arithmetic [upstream]: n_pixels tightly packed BGR8 pixels in, the adjusted pixels out. */
void orc_hsv_round_trip(const uint8_t *bgr_in, uint8_t *bgr_out, int64_t n_pixels, float saturation_scale,
                        float brightness_scale);

/* keyframe rule, PCP/include/PointCloudProcessor.hpp:151-191, cpp:1050-1075 */
int32_t orc_select_keyframes(const orc_pose *poses, int32_t n, double dist_threshold, int32_t *out_indices);

/* StatisticalOutlierRemoval (k, std_mul) [upstream], cloudSmooth.cpp:109-116.
 * out_keep n bytes; out_distance n floats (mean kNN distance), out_threshold: both
 * nullable; returns number kept. */
int64_t orc_sor(const float *x, const float *y, const float *z, int64_t n, int32_t mean_k, double std_mul,
                uint8_t *out_keep, float *out_distance, double *out_threshold, int32_t threads);

/* NID cost: sum over keyframes of NIDCost (nid_cost.hpp:42-116) on the culled,
 * camera-frame clouds, and its gradient in the SE(3) tangent of T * exp(delta).  See pcp_oracle_nid.c. */
int orc_nid(const orc_camera *cam, const uint8_t *const *images, int32_t n_frames, const int64_t *offsets,
            const float *x, const float *y, const float *z, const float *intensity, const double T[16], int32_t bins,
            double *out_cost, double *out_grad);

/* a4: ViewCulling::hidden_points_removal (view_culling.cpp:266-334), see pcp_oracle_hpr.c.
 * orc_orient3d: sign of det [a-d; b-d; c-d], exact.  orc_convex_hull_vertices: exact extreme points of n points
 * (xyz triples); stats (nullable, 4): filtered / exact / exactly-zero orientation tests, duplicates; returns the vertex
 * count, -1 for fewer than 4 points or a flat set, -2 out of memory.  orc_hpr_flip: :291-292.  orc_hpr_frame: keep mask
 * (input order) of one keyframe; stats (nullable, 5): candidates, then the four above. */
int orc_orient3d(const double a[3], const double b[3], const double c[3], const double d[3], int32_t exact_only);
int64_t orc_convex_hull_vertices(const double *points, int64_t n, uint8_t *is_vertex, int64_t *stats);
void orc_hpr_flip(const float *xc, const float *yc, const float *zc, int64_t m, double flip_radius, double *out_flipped);
int64_t orc_hpr_frame(const orc_camera *cam, const float w2c[12], const float *x, const float *y, const float *z,
                      int64_t n, double flip_radius, uint8_t *out_keep, int64_t *stats);

int32_t orc_hardware_threads(void);

#ifdef __cplusplus
}
#endif
#endif
