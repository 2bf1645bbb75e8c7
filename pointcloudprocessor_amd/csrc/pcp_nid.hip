// pcp_nid.hip -- NID (normalised information distance) extrinsic refinement, SURVEY.md 8 f1:
// the cost of NIDCost::operator() (PCP/include/vlcal/costs/nid_cost.hpp:42-116) summed over
// keyframes as MultiNIDCost does (PCP/src/vlcal/calib/visual_camera_calibration.cpp:86-129),
// with its gradient in the SE(3) tangent of T * exp(delta) (what ceres::GradientProblem with
// Sophus::Manifold<SE3> feeds its BFGS line search, :220-225), and a host BFGS on that manifold
// standing in for ceres::Solve (:245), wrapped in the outer loop of
// VisualCameraCalibration::calibrate (:49-80).
//
// Inputs per keyframe = what the reference reads back from <ts>_beforeNID.pcd: the z-buffer
// culled points in camera coordinates with their intensity (PointCloudProcessor.cpp:178-224),
// and the keyframe's raw BGR image scaled to [0,1] (visual_camera_calibration.cpp:171-173).
//
// GPU part: one workgroup per chunk of 4096 points of one keyframe accumulates the 16x16 joint
// histogram of (image bin, intensity bin) with cubic B-spline weights -- value and 6 tangent
// derivatives per cell -- in LDS (ds_add_f64), then adds it to the keyframe's histogram in HBM
// with fp64 atomics.  Entropies, NID and the chain rule are finished on the host in fp64
// (a few thousand numbers per keyframe).
//
// Not reproduced: the 8-significant-digit round trip of the culled clouds through ASCII PCD
// files, and Ceres' exact line-search trajectory (its optimiser is not restated; the cost and
// gradient it consumes are, and are checked against oracle/pcp_oracle_nid.c).
// Reference accident reproduced: the 3-channel image is read as if single-channel,
// at<double>(y, x) == channel x % 3 of pixel x / 3 (nid_cost.hpp:87 on a CV_64FC3 matrix).
#include <algorithm>
#include <cmath>
#include <cstring>

#include "pcp_internal.hpp"

namespace pcp {

constexpr int kNB = 256;
constexpr int kNidChunk = 4096;  // points per workgroup
constexpr int kNidComp = 7;      // value + 6 tangent derivatives

// camera coordinates + intensity of the kept points of one keyframe into its padded segment
__global__ __launch_bounds__(kNB) void k_nid_gather(const float *__restrict__ x, const float *__restrict__ y,
                                                    const float *__restrict__ z, const float *__restrict__ intensity,
                                                    const int32_t *__restrict__ index, int64_t m, int64_t padded,
                                                    DevFrame fr, float4 *__restrict__ out) {
  const int64_t k = static_cast<int64_t>(blockIdx.x) * kNB + threadIdx.x;
  if (k >= padded) return;
  float4 o;
  if (k < m) {
    const int32_t i = index[k];
    const float px = x[i], py = y[i], pz = z[i];
    const float *mm = fr.w2c;  // pcl::transformPointCloud association (A2), -ffp-contract=off build
    o.x = px * mm[0] + (py * mm[1] + (pz * mm[2] + mm[3]));
    o.y = px * mm[4] + (py * mm[5] + (pz * mm[6] + mm[7]));
    o.z = px * mm[8] + (py * mm[9] + (pz * mm[10] + mm[11]));
    o.w = intensity[i];
  } else {
    o.x = o.y = o.z = 0.0f;
    o.w = __uint_as_float(0x7fc00000u);  // NaN marks padding
  }
  out[k] = o;
}

// The histogram / entropy arithmetic below is tolerance-gated (checked against the oracle's
// dual-number gradient), so FMA contraction is allowed from here on; k_nid_gather above is not.
#pragma clang fp contract(fast)

struct NidArgs {
  const float4 *pts;
  const int32_t *chunk_kf;
  const uint32_t *images;  // texels B | G<<8 | R<<16 | mask<<24
  int64_t image_px;
  int32_t W, H, bins;
  double fx, fy, cx, cy, k1, k2, p1, p2, k3;
  double T[12];  // T_camera_lidar, 3x4 row-major
  double *hist;  // [keyframe][bins*bins*7 + bins]
};

__device__ __forceinline__ void bspline(double s, double b[4], double db[4]) {
  // rows of spline_coeffs / 6 applied to (1, s, s^2, s^3) (nid_cost.hpp:35-39,74-79)
  const double s2 = s * s, s3 = s2 * s;
  b[0] = (1.0 - 3.0 * s + 3.0 * s2 - s3) / 6.0;
  b[1] = (4.0 - 6.0 * s2 + 3.0 * s3) / 6.0;
  b[2] = (1.0 + 3.0 * s + 3.0 * s2 - 3.0 * s3) / 6.0;
  b[3] = s3 / 6.0;
  db[0] = (-3.0 + 6.0 * s - 3.0 * s2) / 6.0;
  db[1] = (-12.0 * s + 9.0 * s2) / 6.0;
  db[2] = (3.0 + 6.0 * s - 9.0 * s2) / 6.0;
  db[3] = 3.0 * s2 / 6.0;
}

__global__ __launch_bounds__(kNB) void k_nid_hist(NidArgs a) {
  extern __shared__ double lh[];  // bins*bins*7 + bins
  const int cells = a.bins * a.bins;
  const int lsize = cells * kNidComp + a.bins;
  for (int k = threadIdx.x; k < lsize; k += kNB) lh[k] = 0.0;
  __syncthreads();
  const int32_t kf = a.chunk_kf[blockIdx.x];
  const uint32_t *img = a.images + static_cast<int64_t>(kf) * a.image_px;
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kNidChunk;
  for (int it = 0; it < kNidChunk / kNB; ++it) {
    const float4 q = a.pts[base + it * kNB + threadIdx.x];
    if (isnan(q.w)) continue;  // padding
    const double px = q.x, py = q.y, pz = q.z;
    // pt_camera = T * p and its tangent derivatives: d/d upsilon = R, d/d omega = R d(omega x p)/d omega
    double pc[3], dpc[3][6];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const double R0 = a.T[4 * r], R1 = a.T[4 * r + 1], R2 = a.T[4 * r + 2];
      pc[r] = (R0 * px + R1 * py + R2 * pz) + a.T[4 * r + 3];
      dpc[r][0] = R0;
      dpc[r][1] = R1;
      dpc[r][2] = R2;
      dpc[r][3] = -R1 * pz + R2 * py;
      dpc[r][4] = R0 * pz - R2 * px;
      dpc[r][5] = -R0 * py + R1 * px;
    }
    int bp = static_cast<int>(static_cast<double>(q.w) * a.bins);
    bp = max(0, min(a.bins - 1, bp));
    // projection (pinhole.hpp:13-51) with its 2x3 Jacobian
    const double iz = 1.0 / pc[2];
    const double xn = pc[0] * iz, yn = pc[1] * iz;
    const double x2 = xn * xn, y2 = yn * yn, r2 = x2 + y2, r4 = r2 * r2, r6 = r2 * r4;
    const double rc = 1.0 + a.k1 * r2 + a.k2 * r4 + a.k3 * r6;
    const double drc = a.k1 + 2.0 * a.k2 * r2 + 3.0 * a.k3 * r4;  // d rc / d r2
    const double xd = rc * xn + a.p1 * (2.0 * xn * yn) + a.p2 * (r2 + 2.0 * x2);
    const double yd = rc * yn + a.p1 * (r2 + 2.0 * y2) + a.p2 * (2.0 * xn * yn);
    const double u = a.fx * xd + a.cx, v = a.fy * yd + a.cy;
    if (!(fabs(u) < 1e9) || !(fabs(v) < 1e9)) continue;
    const int kx = static_cast<int>(floor(u)), ky = static_cast<int>(floor(v));
    if (kx < 0 || ky < 0 || kx >= a.W || ky >= a.H) continue;  // num_outliers++ (nid_cost.hpp:66-69)
    const double xd_xn = rc + 2.0 * x2 * drc + 2.0 * a.p1 * yn + 6.0 * a.p2 * xn;
    const double xd_yn = 2.0 * xn * yn * drc + 2.0 * a.p1 * xn + 2.0 * a.p2 * yn;
    const double yd_xn = 2.0 * xn * yn * drc + 2.0 * a.p1 * xn + 2.0 * a.p2 * yn;
    const double yd_yn = rc + 2.0 * y2 * drc + 6.0 * a.p1 * yn + 2.0 * a.p2 * xn;
    // d(xn, yn)/d pc = (iz, 0, -xn iz), (0, iz, -yn iz)
    double du[6], dv[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const double dxn = (dpc[0][k] - xn * dpc[2][k]) * iz;
      const double dyn = (dpc[1][k] - yn * dpc[2][k]) * iz;
      du[k] = a.fx * (xd_xn * dxn + xd_yn * dyn);
      dv[k] = a.fy * (yd_xn * dxn + yd_yn * dyn);
    }
    atomicAdd(&lh[cells * kNidComp + bp], 1.0);  // hist_points[bin_points]++
    double bx[4], dbx[4], by[4], dby[4];
    bspline(u - static_cast<double>(kx), bx, dbx);
    bspline(v - static_cast<double>(ky), by, dby);
    // The 16 taps of a point share its point bin and its du / dv, and neighbouring taps mostly fall in the same image
    // bin: the weights (w, dw/du, dw/dv) of a run of equal bins are summed in registers and flushed as 7 atomics per
    // run instead of 7 per tap (hist_k += du_k * sum(wu) + dv_k * sum(wv)).
    int run_bin = -1;
    double rw = 0.0, rwu = 0.0, rwv = 0.0;
    auto flush = [&]() {
      if (run_bin < 0) return;
      double *cell = lh + (run_bin * a.bins + bp) * kNidComp;
      atomicAdd(cell, rw);
#pragma unroll
      for (int k = 0; k < 6; ++k) atomicAdd(cell + 1 + k, rwu * du[k] + rwv * dv[k]);
    };
#pragma unroll
    for (int j = 0; j < 4; ++j) {  // rows outermost: the four taps of a row are horizontal neighbours
      const int sy = max(0, min(a.H - 1, ky - 1 + j));
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int sx = max(0, min(a.W - 1, kx - 1 + i));
        // normalized_image.at<double>(sy, sx) of a CV_64FC3 matrix: channel sx % 3 of pixel sx / 3
        const uint32_t texel = img[static_cast<int64_t>(sy) * a.W + sx / 3];
        const double pix = static_cast<double>((texel >> (8 * (sx % 3))) & 0xffu) / 255.0;
        const int bi = min(static_cast<int>(pix * a.bins), a.bins - 1);
        if (bi != run_bin) {
          flush();
          run_bin = bi;
          rw = rwu = rwv = 0.0;
        }
        rw += bx[i] * by[j];
        rwu += dbx[i] * by[j];
        rwv += bx[i] * dby[j];
      }
    }
    flush();
  }
  __syncthreads();
  double *gh = a.hist + static_cast<int64_t>(kf) * lsize;
  for (int k = threadIdx.x; k < lsize; k += kNB)
    if (lh[k] != 0.0) atomicAdd(gh + k, lh[k]);
}

// ---- small SE(3) algebra on the host ------------------------------------------------------
static void se3_exp(const double d[6], double M[16]) {  // Sophus SE3::exp, delta = (upsilon, omega)
  const double wx = d[3], wy = d[4], wz = d[5];
  const double th2 = wx * wx + wy * wy + wz * wz, th = std::sqrt(th2);
  const double K[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
  double K2[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) K2[3 * r + c] = K[3 * r] * K[c] + K[3 * r + 1] * K[3 + c] + K[3 * r + 2] * K[6 + c];
  double A, B, Cc;
  if (th < 1e-8) {
    A = 1.0 - th2 / 6.0;
    B = 0.5 - th2 / 24.0;
    Cc = 1.0 / 6.0 - th2 / 120.0;
  } else {
    A = std::sin(th) / th;
    B = (1.0 - std::cos(th)) / th2;
    Cc = (th - std::sin(th)) / (th2 * th);
  }
  double R[9], V[9];
  for (int k = 0; k < 9; ++k) {
    const double I = (k % 4 == 0) ? 1.0 : 0.0;
    R[k] = I + A * K[k] + B * K2[k];
    V[k] = I + B * K[k] + Cc * K2[k];
  }
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) M[4 * r + c] = R[3 * r + c];
    M[4 * r + 3] = V[3 * r] * d[0] + V[3 * r + 1] * d[1] + V[3 * r + 2] * d[2];
  }
  M[12] = M[13] = M[14] = 0.0;
  M[15] = 1.0;
}

static void mat_mul(const double A[16], const double B[16], double C[16]) {
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) {
      double s = 0.0;
      for (int k = 0; k < 4; ++k) s += A[4 * r + k] * B[4 * k + c];
      C[4 * r + c] = s;
    }
}

static void rigid_inverse(const double A[16], double out[16]) {
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) out[4 * r + c] = A[4 * c + r];
    out[4 * r + 3] = -(A[0 * 4 + r] * A[3] + A[1 * 4 + r] * A[7] + A[2 * 4 + r] * A[11]);
  }
  out[12] = out[13] = out[14] = 0.0;
  out[15] = 1.0;
}

// MultiNIDCost's domain: |translation| <= 0.2 m and rotation angle <= 2 deg around the initial
// guess (visual_camera_calibration.cpp:100-105)
static bool inside_limits(const double init[16], const double T[16], double *dt = nullptr, double *dr = nullptr) {
  double inv[16], delta[16];
  rigid_inverse(init, inv);
  mat_mul(inv, T, delta);
  const double t = std::sqrt(delta[3] * delta[3] + delta[7] * delta[7] + delta[11] * delta[11]);
  const double tr = delta[0] + delta[5] + delta[10];
  const double ang = std::acos(std::max(-1.0, std::min(1.0, 0.5 * (tr - 1.0))));
  if (dt) *dt = t;
  if (dr) *dr = ang;
  return !(t > 0.2 || ang > 2.0 * M_PI / 180.0);
}

// the keyframes' joint histograms of this context's points at T, into ctx->nid_hist
static int nid_accumulate(pcp_context *ctx, const double T[16], int32_t bins) {
  const int cells = bins * bins;
  const int lsize = cells * kNidComp + bins;
  const size_t hsize = static_cast<size_t>(ctx->nid_frames) * lsize;
  PCP_HIP_TRY(ctx, ctx->nid_hist.ensure(hsize + 8));
  PCP_HIP_TRY(ctx, hipMemsetAsync(ctx->nid_hist.p, 0, hsize * sizeof(double), ctx->stream));
  NidArgs a{};
  a.pts = reinterpret_cast<const float4 *>(ctx->nid_pts.p);
  a.chunk_kf = ctx->nid_chunk_kf.p;
  a.images = ctx->images.p;
  a.image_px = static_cast<int64_t>(ctx->dcam.img_w) * ctx->dcam.img_h;
  a.W = ctx->dcam.img_w;
  a.H = ctx->dcam.img_h;
  a.bins = bins;
  a.fx = ctx->dcam.fx;
  a.fy = ctx->dcam.fy;
  a.cx = ctx->dcam.cx;
  a.cy = ctx->dcam.cy;
  a.k1 = ctx->dcam.k1;
  a.k2 = ctx->dcam.k2;
  a.p1 = ctx->dcam.p1;
  a.p2 = ctx->dcam.p2;
  a.k3 = ctx->dcam.k3;
  for (int k = 0; k < 12; ++k) a.T[k] = T[k];
  a.hist = ctx->nid_hist.p;
  if (ctx->nid_chunks > 0) {
    LaunchTimer t(ctx, PCP_K_NID);
    hipLaunchKernelGGL(k_nid_hist, dim3(static_cast<uint32_t>(ctx->nid_chunks)), dim3(kNB), lsize * sizeof(double),
                       ctx->stream, a);
    PCP_HIP_TRY(ctx, hipGetLastError());
  }
  ctx->nid_hist_bins = bins;
  return PCP_OK;
}

// cost and tangent gradient from the histograms in ctx->nid_hist (this context's, or the sum over the shards)
static int nid_finish(pcp_context *ctx, int32_t bins, double *cost, double grad[6], bool *finite) {
  const int cells = bins * bins;
  const int lsize = cells * kNidComp + bins;
  const size_t hsize = static_cast<size_t>(ctx->nid_frames) * lsize;
  std::vector<double> h(hsize);
  PCP_HIP_TRY(ctx, hipMemcpyAsync(h.data(), ctx->nid_hist.p, hsize * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  PCP_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  // entropies and the chain rule (nid_cost.hpp:95-108), per keyframe, summed (MultiNIDCost :118-123)
  double total = 0.0, g[6] = {0, 0, 0, 0, 0, 0};
  bool ok = true;
  for (int32_t f = 0; f < ctx->nid_frames; ++f) {
    const double *hk = h.data() + static_cast<size_t>(f) * lsize;
    const double *hp = hk + cells * kNidComp;
    double sum = 0.0;
    for (int b = 0; b < bins; ++b) sum += hp[b];
    double H_i = 0, H_p = 0, H_ip = 0, dH_i[6] = {0, 0, 0, 0, 0, 0}, dH_ip[6] = {0, 0, 0, 0, 0, 0};
    for (int b = 0; b < bins; ++b) {
      const double p = hp[b] / sum;
      H_p -= p * std::log(p + 1e-6);
    }
    for (int bi = 0; bi < bins; ++bi) {
      double hi = 0.0, dhi[6] = {0, 0, 0, 0, 0, 0};
      for (int bpn = 0; bpn < bins; ++bpn) {
        const double *c = hk + (bi * bins + bpn) * kNidComp;
        const double v = c[0] / sum;
        const double dl = std::log(v + 1e-6) + v / (v + 1e-6);
        H_ip -= v * std::log(v + 1e-6);
        hi += c[0];
        for (int k = 0; k < 6; ++k) {
          dH_ip[k] -= dl * c[1 + k] / sum;
          dhi[k] += c[1 + k];
        }
      }
      const double v = hi / sum;
      const double dl = std::log(v + 1e-6) + v / (v + 1e-6);
      H_i -= v * std::log(v + 1e-6);
      for (int k = 0; k < 6; ++k) dH_i[k] -= dl * dhi[k] / sum;
    }
    const double MI = H_i + H_p - H_ip;
    const double nid = (H_ip - MI) / H_ip;
    if (!std::isfinite(nid)) {
      ok = false;
      continue;
    }
    total += nid;
    // NID = (2 H_ip - H_i - H_p) / H_ip
    for (int k = 0; k < 6; ++k) g[k] += ((2.0 * dH_ip[k] - dH_i[k]) * H_ip - (2.0 * H_ip - H_i - H_p) * dH_ip[k]) / (H_ip * H_ip);
  }
  *cost = total;
  for (int k = 0; k < 6; ++k) grad[k] = g[k];
  *finite = ok;
  return PCP_OK;
}

// one evaluation of the summed cost and tangent gradient at T
static int nid_eval(pcp_context *ctx, const double T[16], int32_t bins, double *cost, double grad[6], bool *finite) {
  int rc = nid_accumulate(ctx, T, bins);
  if (rc != PCP_OK) return rc;
  return nid_finish(ctx, bins, cost, grad, finite);
}

}  // namespace pcp

namespace pcp {
// (pcp_create loads every code object of the library up front: see preload_code_objects in pcp_context.hip)
hipError_t preload_nid() {
  hipFuncAttributes a;
  return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&k_nid_gather));
}
}  // namespace pcp

using namespace pcp;

extern "C" {

int pcp_upload_intensity(pcp_context *ctx, const float *intensity, int64_t n) {
  if (!ctx) return PCP_ERR_INVALID;
  if (!ctx->xyz.p) return set_error(ctx, PCP_ERR_STATE, "pcp_upload_intensity: no cloud uploaded");
  if (n != ctx->n) return set_error(ctx, PCP_ERR_INVALID, "pcp_upload_intensity: %lld values for a cloud of %lld points",
                                    (long long)n, (long long)ctx->n);
  if (n > 0 && !intensity) return set_error(ctx, PCP_ERR_INVALID, "pcp_upload_intensity: NULL intensity");
  PCP_HIP_TRY(ctx, hipSetDevice(ctx->device));
  PCP_HIP_TRY(ctx, ctx->intensity.ensure(static_cast<size_t>(n) + 4));
  if (n > 0)
    PCP_HIP_TRY(ctx, hipMemcpyAsync(ctx->intensity.p, intensity, static_cast<size_t>(n) * 4, hipMemcpyHostToDevice, ctx->stream));
  PCP_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  ctx->have_intensity = true;
  ctx->nid_chunks = 0;
  return PCP_OK;
}

int pcp_nid_prepare(pcp_context *ctx, int64_t *out_points) {
  if (!ctx) return PCP_ERR_INVALID;
  if (!ctx->have_camera || !ctx->xyz.p || ctx->n_frames <= 0)
    return set_error(ctx, PCP_ERR_STATE, "pcp_nid_prepare: camera, cloud and keyframes must be set");
  if (!ctx->have_intensity) return set_error(ctx, PCP_ERR_STATE, "pcp_nid_prepare: pcp_upload_intensity has not been called");
  for (int32_t f = 0; f < ctx->n_frames; ++f)
    if (!ctx->images.p || !ctx->image_set[static_cast<size_t>(f)])
      return set_error(ctx, PCP_ERR_STATE, "pcp_nid_prepare: no image uploaded for keyframe %d", f);
  PCP_HIP_TRY(ctx, hipSetDevice(ctx->device));
  {
    const int rcw = pcp::wait_images(ctx, 0, ctx->n_frames);  // asynchronous uploads still in flight
    if (rcw != PCP_OK) return rcw;
  }
  const int64_t n = ctx->n;
  const size_t plane = (static_cast<size_t>(n) + 3) & ~size_t(3);
  PCP_HIP_TRY(ctx, ctx->s_cell.ensure(plane + 4));
  // one cull per keyframe; the point buffer is sized from the first keyframe and grown (contents kept) if the
  // estimate falls short
  std::vector<int32_t> chunk_kf;
  int64_t at = 0, total = 0;
  size_t capacity = ctx->nid_pts.count / 4;  // points (float4)
  for (int32_t f = 0; f < ctx->n_frames; ++f) {
    int64_t m = 0;
    int rc = cull_frame_indices(ctx, f, ctx->s_cell.p, n, &m);
    if (rc != PCP_OK) return rc;
    const int64_t padded = div_up(m, kNidChunk) * kNidChunk;
    if (static_cast<size_t>(at + padded) > capacity) {
      const size_t want = std::max<size_t>(static_cast<size_t>(at + padded),
                                           static_cast<size_t>((at + padded) * 1.3 * ctx->n_frames / (f + 1)) + kNidChunk);
      DevBuf<float> bigger;
      PCP_HIP_TRY(ctx, bigger.ensure(want * 4 + 16));
      if (at > 0)
        PCP_HIP_TRY(ctx, hipMemcpyAsync(bigger.p, ctx->nid_pts.p, static_cast<size_t>(at) * 16, hipMemcpyDeviceToDevice, ctx->stream));
      PCP_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
      std::swap(ctx->nid_pts, bigger);
      bigger.release();
      capacity = want;
    }
    if (padded > 0) {
      hipLaunchKernelGGL(k_nid_gather, dim3(static_cast<uint32_t>(div_up(padded, kNB))), dim3(kNB), 0, ctx->stream,
                         ctx->xyz.p, ctx->xyz.p + plane, ctx->xyz.p + 2 * plane, ctx->intensity.p, ctx->s_cell.p, m,
                         padded, ctx->hframes[static_cast<size_t>(f)],
                         reinterpret_cast<float4 *>(ctx->nid_pts.p) + at);
      PCP_HIP_TRY(ctx, hipGetLastError());
      PCP_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // s_cell is reused by the next keyframe
    }
    for (int64_t c = 0; c < padded / kNidChunk; ++c) chunk_kf.push_back(f);
    at += padded;
    total += m;
  }
  PCP_HIP_TRY(ctx, ctx->nid_chunk_kf.ensure(chunk_kf.size() + 4));
  if (!chunk_kf.empty())
    PCP_HIP_TRY(ctx, hipMemcpyAsync(ctx->nid_chunk_kf.p, chunk_kf.data(), chunk_kf.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  PCP_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  ctx->nid_chunks = static_cast<int64_t>(chunk_kf.size());
  ctx->nid_points = total;
  ctx->nid_frames = ctx->n_frames;
  if (out_points) *out_points = total;
  return PCP_OK;
}

int pcp_nid_evaluate(pcp_context *ctx, const double T[16], const double *T_init, int32_t bins, double *cost,
                     double grad6[6], int32_t *valid) {
  if (!ctx) return PCP_ERR_INVALID;
  if (!T || !cost) return set_error(ctx, PCP_ERR_INVALID, "pcp_nid_evaluate: NULL argument");
  if (bins < 2 || bins > 16) return set_error(ctx, PCP_ERR_INVALID, "pcp_nid_evaluate: bins %d out of range (2..16)", bins);
  if (ctx->nid_frames <= 0 || ctx->nid_frames != ctx->n_frames)
    return set_error(ctx, PCP_ERR_STATE, "pcp_nid_evaluate: pcp_nid_prepare has not been called for these keyframes");
  PCP_HIP_TRY(ctx, hipSetDevice(ctx->device));
  double g[6];
  bool finite = true;
  if (T_init && !inside_limits(T_init, T)) {  // MultiNIDCost returns false
    *cost = 0.0;
    if (grad6) std::memset(grad6, 0, 6 * sizeof(double));
    if (valid) *valid = 0;
    return PCP_OK;
  }
  int rc = nid_eval(ctx, T, bins, cost, g, &finite);
  if (rc != PCP_OK) return rc;
  if (grad6) std::memcpy(grad6, g, sizeof(g));
  if (valid) *valid = finite ? 1 : 0;
  return PCP_OK;
}

}  // extern "C"

namespace pcp {
// VisualCameraCalibration::calibrate's loop around any evaluator of the cost (one context, or the host that owns the shards)
template <class Eval>
static int nid_optimize_loop(pcp_context *ctx, Eval &&nid_eval_at, const double T_init[16], int32_t bins,
                             int32_t max_outer_iterations, double T_out[16], double *final_cost, int32_t *evaluations) {
  double T[16];
  std::memcpy(T, T_init, sizeof(T));
  int32_t evals = 0;
  double fbest = 0.0;
  for (int32_t outer = 0; outer < std::max(1, max_outer_iterations); ++outer) {
    // inner: BFGS on the SE(3) manifold (stands in for ceres::Solve, line_search_direction_type BFGS)
    double init[16], x[16];
    std::memcpy(init, T, sizeof(T));
    std::memcpy(x, T, sizeof(T));
    double f, g[6];
    bool fin;
    int rc = nid_eval_at(x, &f, g, &fin);
    if (rc != PCP_OK) return rc;
    ++evals;
    if (!fin) return set_error(ctx, PCP_ERR_STATE, "pcp_nid_optimize: the cost is not finite at the initial guess");
    double Hm[36];
    for (int k = 0; k < 36; ++k) Hm[k] = (k % 7 == 0) ? 1.0 : 0.0;
    for (int it = 0; it < 50; ++it) {  // ceres default max_num_iterations
      double gmax = 0.0;
      for (int k = 0; k < 6; ++k) gmax = std::max(gmax, std::fabs(g[k]));
      if (gmax <= 1e-10) break;  // gradient_tolerance
      double d[6];
      double slope = 0.0;
      for (int r = 0; r < 6; ++r) {
        d[r] = 0.0;
        for (int c = 0; c < 6; ++c) d[r] -= Hm[6 * r + c] * g[c];
        slope += d[r] * g[r];
      }
      if (!(slope < 0.0)) {  // not a descent direction: restart from steepest descent
        for (int k = 0; k < 36; ++k) Hm[k] = (k % 7 == 0) ? 1.0 : 0.0;
        slope = 0.0;
        for (int k = 0; k < 6; ++k) {
          d[k] = -g[k];
          slope -= g[k] * g[k];
        }
      }
      // Armijo backtracking; a step outside MultiNIDCost's domain counts as a failed trial
      double alpha = (it == 0) ? std::min(1.0, 1.0 / gmax) * 1e-2 : 1.0;
      double fn = f, gn[6], xn[16];
      bool accepted = false;
      for (int ls = 0; ls < 20; ++ls) {
        double step[6], E[16];
        for (int k = 0; k < 6; ++k) step[k] = alpha * d[k];
        se3_exp(step, E);
        mat_mul(x, E, xn);
        bool fin2 = false;
        if (inside_limits(init, xn)) {
          rc = nid_eval_at(xn, &fn, gn, &fin2);
          if (rc != PCP_OK) return rc;
          ++evals;
        }
        if (fin2 && fn <= f + 1e-4 * alpha * slope) {
          accepted = true;
          break;
        }
        alpha *= 0.5;
      }
      if (!accepted) break;
      // BFGS update of the inverse Hessian with s = alpha d, y = g_new - g
      double s[6], y[6], sy = 0.0;
      for (int k = 0; k < 6; ++k) {
        s[k] = alpha * d[k];
        y[k] = gn[k] - g[k];
        sy += s[k] * y[k];
      }
      if (sy > 1e-14) {
        double Hy[6], yHy = 0.0;
        for (int r = 0; r < 6; ++r) {
          Hy[r] = 0.0;
          for (int c = 0; c < 6; ++c) Hy[r] += Hm[6 * r + c] * y[c];
        }
        for (int k = 0; k < 6; ++k) yHy += y[k] * Hy[k];
        for (int r = 0; r < 6; ++r)
          for (int c = 0; c < 6; ++c)
            Hm[6 * r + c] += (1.0 + yHy / sy) * s[r] * s[c] / sy - (Hy[r] * s[c] + s[r] * Hy[c]) / sy;
      }
      const double df = std::fabs(fn - f);
      std::memcpy(x, xn, sizeof(x));
      std::memcpy(g, gn, sizeof(g));
      const double fprev = f;
      f = fn;
      if (df <= 1e-6 * std::fabs(fprev)) break;  // function_tolerance
    }
    // outer loop convergence (visual_camera_calibration.cpp:66-78; thresholds hpp:23-24)
    double dt = 0.0, dr = 0.0;
    inside_limits(x, T, &dt, &dr);
    std::memcpy(T, x, sizeof(T));
    fbest = f;
    if (dt < 0.01 && dr < 1.0 * M_PI / 180.0) break;
  }
  std::memcpy(T_out, T, sizeof(T));
  if (final_cost) *final_cost = fbest;
  if (evaluations) *evaluations = evals;
  (void)bins;
  return PCP_OK;
}
}  // namespace pcp

extern "C" {

int pcp_nid_optimize(pcp_context *ctx, const double T_init[16], int32_t bins, int32_t max_outer_iterations,
                     double T_out[16], double *final_cost, int32_t *evaluations) {
  if (!ctx) return PCP_ERR_INVALID;
  if (!T_init || !T_out) return set_error(ctx, PCP_ERR_INVALID, "pcp_nid_optimize: NULL argument");
  if (bins < 2 || bins > 16) return set_error(ctx, PCP_ERR_INVALID, "pcp_nid_optimize: bins %d out of range (2..16)", bins);
  if (ctx->nid_frames <= 0 || ctx->nid_frames != ctx->n_frames)
    return set_error(ctx, PCP_ERR_STATE, "pcp_nid_optimize: pcp_nid_prepare has not been called for these keyframes");
  PCP_HIP_TRY(ctx, hipSetDevice(ctx->device));
  return nid_optimize_loop(
      ctx, [&](const double *T, double *f, double *g, bool *fin) { return nid_eval(ctx, T, bins, f, g, fin); }, T_init, bins,
      max_outer_iterations, T_out, final_cost, evaluations);
}

int pcp_nid_optimize_with(pcp_context *ctx, pcp_nid_eval_fn eval, void *user, const double T_init[16], int32_t bins,
                          int32_t max_outer_iterations, double T_out[16], double *final_cost, int32_t *evaluations) {
  if (!ctx) return PCP_ERR_INVALID;
  if (!eval || !T_init || !T_out) return set_error(ctx, PCP_ERR_INVALID, "pcp_nid_optimize_with: NULL argument");
  if (bins < 2 || bins > 16) return set_error(ctx, PCP_ERR_INVALID, "pcp_nid_optimize_with: bins %d out of range (2..16)", bins);
  return nid_optimize_loop(
      ctx,
      [&](const double *T, double *f, double *g, bool *fin) {
        int32_t valid = 0;
        const int rc = eval(user, T, bins, f, g, &valid);
        *fin = valid != 0;
        if (rc != PCP_OK) return set_error(ctx, rc, "pcp_nid_optimize_with: the evaluator failed (%d)", rc);
        return PCP_OK;
      },
      T_init, bins, max_outer_iterations, T_out, final_cost, evaluations);
}

int pcp_nid_accumulate(pcp_context *ctx, const double T[16], int32_t bins) {
  if (!ctx) return PCP_ERR_INVALID;
  if (!T) return set_error(ctx, PCP_ERR_INVALID, "pcp_nid_accumulate: NULL argument");
  if (bins < 2 || bins > 16) return set_error(ctx, PCP_ERR_INVALID, "pcp_nid_accumulate: bins %d out of range (2..16)", bins);
  if (ctx->nid_frames <= 0 || ctx->nid_frames != ctx->n_frames)
    return set_error(ctx, PCP_ERR_STATE, "pcp_nid_accumulate: pcp_nid_prepare has not been called for these keyframes");
  PCP_HIP_TRY(ctx, hipSetDevice(ctx->device));
  return nid_accumulate(ctx, T, bins);
}

int pcp_nid_histograms_device(pcp_context *ctx, void **device_ptr, int64_t *count) {
  if (!ctx || !device_ptr || !count) return PCP_ERR_INVALID;
  if (ctx->nid_frames <= 0 || ctx->nid_hist_bins <= 0 || !ctx->nid_hist.p)
    return set_error(ctx, PCP_ERR_STATE, "pcp_nid_histograms_device: pcp_nid_accumulate has not been called");
  *device_ptr = ctx->nid_hist.p;
  *count = static_cast<int64_t>(ctx->nid_frames) * (ctx->nid_hist_bins * ctx->nid_hist_bins * kNidComp + ctx->nid_hist_bins);
  return PCP_OK;
}

int pcp_nid_finish(pcp_context *ctx, int32_t bins, double *cost, double grad6[6], int32_t *valid) {
  if (!ctx) return PCP_ERR_INVALID;
  if (!cost) return set_error(ctx, PCP_ERR_INVALID, "pcp_nid_finish: NULL argument");
  if (ctx->nid_frames <= 0 || ctx->nid_hist_bins != bins || !ctx->nid_hist.p)
    return set_error(ctx, PCP_ERR_STATE, "pcp_nid_finish: no histograms accumulated with %d bins", bins);
  PCP_HIP_TRY(ctx, hipSetDevice(ctx->device));
  double g[6];
  bool finite = true;
  int rc = nid_finish(ctx, bins, cost, g, &finite);
  if (rc != PCP_OK) return rc;
  if (grad6) std::memcpy(grad6, g, sizeof(g));
  if (valid) *valid = finite ? 1 : 0;
  return PCP_OK;
}

}  // extern "C"
