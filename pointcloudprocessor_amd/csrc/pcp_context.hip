// pcp_context.hip -- handle lifecycle, configuration, uploads, host-side pose
// algebra (A1) and the hipEvent timing facility of libpcp_hip.so.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <new>
#include <vector>

#include "pcp_internal.hpp"
#include "pcp_scan.hpp"

namespace pcp {

static thread_local std::string g_error;

int set_error(const pcp_context *ctx, int code, const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx)
    ctx->error = buf;
  else
    g_error = buf;
  return code;
}

void set_global_error(const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_error = buf;
}

LaunchTimer::LaunchTimer(pcp_context *c, int32_t kernel) : ctx(c) {
  if (!ctx || !ctx->timing) return;  // (nullptr: an untimed launch, e.g. on a lane's stream)
  auto grab = [&](hipEvent_t &e) {
    if (!ctx->event_pool.empty()) {
      e = ctx->event_pool.back();
      ctx->event_pool.pop_back();
      return true;
    }
    return hipEventCreate(&e) == hipSuccess;
  };
  if (!grab(ev.start)) return;
  if (!grab(ev.stop)) {
    ctx->event_pool.push_back(ev.start);
    return;
  }
  ev.kernel = kernel;
  active = hipEventRecord(ev.start, ctx->stream) == hipSuccess;
}

LaunchTimer::~LaunchTimer() {
  if (!active) return;
  (void)hipEventRecord(ev.stop, ctx->stream);
  ctx->pending.push_back(ev);
}

int drain_timing(pcp_context *ctx) {
  for (auto &p : ctx->pending) {
    float ms = 0.0f;
    hipError_t e = hipEventSynchronize(p.stop);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, p.start, p.stop);
    if (e == hipSuccess && p.kernel >= 0 && p.kernel < PCP_K_COUNT) {
      ctx->slots[p.kernel].total_ms += ms;
      ctx->slots[p.kernel].launches += 1;
    }
    ctx->event_pool.push_back(p.start);
    ctx->event_pool.push_back(p.stop);
  }
  ctx->pending.clear();
  return PCP_OK;
}

// ---- A1: pose -> matrices (PointCloudProcessor.cpp:495-519) --------------------
// Eigen::Quaterniond::toRotationMatrix without normalisation, fp64.
static void rotation_from_quaternion(const pcp_pose &p, double R[3][3]) {
  const double tx = 2.0 * p.qx, ty = 2.0 * p.qy, tz = 2.0 * p.qz;
  const double twx = tx * p.qw, twy = ty * p.qw, twz = tz * p.qw;
  const double txx = tx * p.qx, txy = ty * p.qx, txz = tz * p.qx;
  const double tyy = ty * p.qy, tyz = tz * p.qy, tzz = tz * p.qz;
  R[0][0] = 1.0 - (tyy + tzz);
  R[0][1] = txy - twz;
  R[0][2] = txz + twy;
  R[1][0] = txy + twz;
  R[1][1] = 1.0 - (txx + tzz);
  R[1][2] = tyz - twx;
  R[2][0] = txz - twy;
  R[2][1] = tyz + twx;
  R[2][2] = 1.0 - (txx + tyy);
}

// Eigen Transform<float,3,Affine>::inverse(): [L^-1 | -L^-1 t], all fp32 (PointCloudProcessor.cpp:509,518,578).
// Eigen 3.3.7 LU/InverseImpl.h compute_inverse<3>: cofactor_3x3<i,j> = m(i1,j1) m(i2,j2) - m(i1,j2) m(i2,j1),
// det = (cofactors_col0 .* m.col(0)).sum(), inverse(j,i) = cofactor<i,j> / det; a 3-term fp32 sum has no SSE
// packet and goes through Redux.h's scalar unroller as p0 + (p1 + p2); the translation is the coefficient-based
// product (-L^-1) * t with the same 3-term sums.
static void invert_affine_f32(const float in[12], float out[12]) {
  const float a = in[0], b = in[1], c = in[2];
  const float d = in[4], e = in[5], f = in[6];
  const float g = in[8], h = in[9], i = in[10];
  const float k00 = e * i - f * h, k10 = h * c - i * b, k20 = b * f - c * e;  // cofactors of column 0
  const float det = k00 * a + (k10 * d + k20 * g);
  const float inv = 1.0f / det;
  const float L[3][3] = {{k00 * inv, k10 * inv, k20 * inv},
                         {(f * g - d * i) * inv, (i * a - g * c) * inv, (c * d - a * f) * inv},
                         {(d * h - e * g) * inv, (g * b - h * a) * inv, (a * e - b * d) * inv}};
  for (int r = 0; r < 3; ++r) {
    out[4 * r + 0] = L[r][0];
    out[4 * r + 1] = L[r][1];
    out[4 * r + 2] = L[r][2];
    out[4 * r + 3] = (-L[r][0]) * in[3] + ((-L[r][1]) * in[7] + (-L[r][2]) * in[11]);
  }
}

static void matrices_from_pose(const pcp_pose &pose, const double *T, float w2c[12], float c2w[12]) {
  double R[3][3];
  rotation_from_quaternion(pose, R);
  const double t[3] = {pose.x, pose.y, pose.z};
  if (!T) {
    // Isometry3d::inverse() = [R^T | -(R^T t)] in fp64, then cast<float>()
    for (int r = 0; r < 3; ++r) {
      const double a0 = R[0][r], a1 = R[1][r], a2 = R[2][r];
      w2c[4 * r + 0] = static_cast<float>(a0);
      w2c[4 * r + 1] = static_cast<float>(a1);
      w2c[4 * r + 2] = static_cast<float>(a2);
      w2c[4 * r + 3] = static_cast<float>(-((a0 * t[0] + a1 * t[1]) + a2 * t[2]));
      c2w[4 * r + 0] = static_cast<float>(R[r][0]);
      c2w[4 * r + 1] = static_cast<float>(R[r][1]);
      c2w[4 * r + 2] = static_cast<float>(R[r][2]);
      c2w[4 * r + 3] = static_cast<float>(t[r]);
    }
    return;
  }
  // (t_c2w * T_camera_lidar_optimized).cast<float>(), then the general fp32 inverse
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c) {
      double s = (R[r][0] * T[0 * 4 + c] + R[r][1] * T[1 * 4 + c]) + R[r][2] * T[2 * 4 + c];
      if (c == 3) s = s + t[r] * T[15];
      c2w[4 * r + c] = static_cast<float>(s);
    }
  invert_affine_f32(c2w, w2c);
}

// ---------------------------------------------------------------------------
// Cloud upload, on the device: bounding box, 30-bit Morton keys, a stable LSD radix sort (4 passes of
// 8 bits), the Morton-ordered copy, and the bounding spheres of the 64-point tiles and of the groups of 16
// tiles.  (The first version did all of this on one host thread: 275 ms for 10 M points.)
// ---------------------------------------------------------------------------
constexpr int kUpBlock = 256;

__device__ __forceinline__ uint32_t spread10(uint32_t v) {
  v &= 0x3ffu;
  v = (v | (v << 16)) & 0x030000ffu;
  v = (v | (v << 8)) & 0x0300f00fu;
  v = (v | (v << 4)) & 0x030c30c3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}

// per-workgroup min / max of the coordinates (NaN ignored, as std::min / std::max do on the host)
__global__ __launch_bounds__(kUpBlock) void k_up_bbox(const float *__restrict__ x, const float *__restrict__ y,
                                                     const float *__restrict__ z, int64_t n,
                                                     float *__restrict__ partial /* [blocks][6] */,
                                                     unsigned long long *__restrict__ nonfinite) {
  __shared__ float sh[6][kUpBlock / 64];
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  // the box of the finite coordinates (the projection takes non-finite points as they come; the smoothing stages
  // refuse a cloud that has any: their count goes to *nonfinite)
  uint32_t bad = 0;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kUpBlock + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * kUpBlock) {
    const float v[3] = {x[i], y[i], z[i]};
    bool fin = true;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const bool f = fabsf(v[a]) <= FLT_MAX;  // false for NaN and +-inf
      fin = fin && f;
      if (f && v[a] < lo[a]) lo[a] = v[a];
      if (f && v[a] > hi[a]) hi[a] = v[a];
    }
    bad += fin ? 0u : 1u;
  }
  {
    const unsigned long long any = __ballot(bad != 0);
    if (any) {  // rare
      for (int o = 32; o >= 1; o >>= 1) bad += __shfl_xor(bad, o, 64);
      if ((threadIdx.x & 63) == 0) atomicAdd(nonfinite, static_cast<unsigned long long>(bad));
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a)
    for (int o = 32; o >= 1; o >>= 1) {
      lo[a] = fminf(lo[a], __shfl_xor(lo[a], o, 64));
      hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o, 64));
    }
  if ((threadIdx.x & 63) == 0)
    for (int a = 0; a < 3; ++a) {
      sh[a][threadIdx.x >> 6] = lo[a];
      sh[3 + a][threadIdx.x >> 6] = hi[a];
    }
  __syncthreads();
  if (threadIdx.x < 6) {
    float v = sh[threadIdx.x][0];
    for (int k = 1; k < kUpBlock / 64; ++k) v = threadIdx.x < 3 ? fminf(v, sh[threadIdx.x][k]) : fmaxf(v, sh[threadIdx.x][k]);
    partial[static_cast<int64_t>(blockIdx.x) * 6 + threadIdx.x] = v;
  }
}

// 30-bit index of the cell (ix, iy, iz), 10 bits each, along the 3-D Hilbert curve (Skilling, "Programming the Hilbert
// curve", AIP Conf. Proc. 707, 2004: axes to transpose, then the bits interleaved, x most significant).  Consecutive
// indices are always face-adjacent cells: a run of 64 sorted points never straddles one of the Z-order curve's jumps.
__device__ __forceinline__ uint32_t hilbert30(uint32_t x, uint32_t y, uint32_t z) {
  uint32_t X[3] = {x, y, z};
  for (uint32_t Q = 512u; Q > 1u; Q >>= 1) {
    const uint32_t P = Q - 1u;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (X[i] & Q) {
        X[0] ^= P;
      } else {
        const uint32_t t = (X[0] ^ X[i]) & P;
        X[0] ^= t;
        X[i] ^= t;
      }
    }
  }
  X[1] ^= X[0];
  X[2] ^= X[1];
  uint32_t t = 0u;
  for (uint32_t Q = 512u; Q > 1u; Q >>= 1)
    if (X[2] & Q) t ^= Q - 1u;
  X[0] ^= t;
  X[1] ^= t;
  X[2] ^= t;
  return (spread10(X[0]) << 2) | (spread10(X[1]) << 1) | spread10(X[2]);
}

// Spatial order for the batched run: 64 consecutive points (one wavefront) then fall in few z-buffer cells
// and mostly share their keyframe visibility.  Hilbert order by default (PCP_CLOUD_ORDER=morton: Z order, as rounds 1-3
// shipped it): on the C3 scene the bounding spheres of the 64-point tiles shrink from 8.4 to 6.0 cm on average, and the
// few tiles that straddle a jump of the Z curve (radius up to metres) disappear.
__global__ __launch_bounds__(kUpBlock) void k_up_keys(const float *__restrict__ x, const float *__restrict__ y,
                                                     const float *__restrict__ z, int64_t n, float mnx, float mny,
                                                     float mnz, float scx, float scy, float scz,
                                                     uint32_t *__restrict__ key, int32_t *__restrict__ val, int32_t hilbert) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kUpBlock + threadIdx.x;
  if (i >= n) return;
  // non-finite coordinates convert to 0 / saturate: any key is fine, the order only affects speed
  const uint32_t ix = static_cast<uint32_t>(fminf(fmaxf((x[i] - mnx) * scx, 0.0f), 1023.0f));
  const uint32_t iy = static_cast<uint32_t>(fminf(fmaxf((y[i] - mny) * scy, 0.0f), 1023.0f));
  const uint32_t iz = static_cast<uint32_t>(fminf(fmaxf((z[i] - mnz) * scz, 0.0f), 1023.0f));
  key[i] = hilbert ? hilbert30(ix, iy, iz) : (spread10(ix) | (spread10(iy) << 1) | (spread10(iz) << 2));
  val[i] = static_cast<int32_t>(i);
}

// one radix pass, part 1: digit histogram of every 256-key workgroup, digit-major (hist[d][block]) so that the
// exclusive scan of the whole table yields each (digit, workgroup) run's first output slot
__global__ __launch_bounds__(kUpBlock) void k_up_radix_hist(const uint32_t *__restrict__ key, int64_t n, int32_t shift,
                                                           int32_t *__restrict__ hist, int64_t blocks) {
  __shared__ int32_t cnt[256];
  cnt[threadIdx.x] = 0;
  __syncthreads();
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kUpBlock + threadIdx.x;
  if (i < n) atomicAdd(&cnt[(key[i] >> shift) & 0xffu], 1);
  __syncthreads();
  hist[static_cast<int64_t>(threadIdx.x) * blocks + blockIdx.x] = cnt[threadIdx.x];
}

// part 2: stable scatter.  A lane's rank among the equal digits of its wavefront comes from eight ballots (one
// per digit bit); wavefronts of a workgroup are ordered through per-wavefront digit counts in LDS.
__global__ __launch_bounds__(kUpBlock) void k_up_radix_scatter(const uint32_t *__restrict__ key_in,
                                                              const int32_t *__restrict__ val_in, int64_t n,
                                                              int32_t shift, const int32_t *__restrict__ offs,
                                                              int64_t blocks, uint32_t *__restrict__ key_out,
                                                              int32_t *__restrict__ val_out) {
  __shared__ int32_t wcount[kUpBlock / 64][256];
  __shared__ int32_t wbase[kUpBlock / 64][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int k = 0; k < kUpBlock / 64; ++k) wcount[k][threadIdx.x] = 0;
  __syncthreads();
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kUpBlock + threadIdx.x;
  const bool valid = i < n;
  const uint32_t k = valid ? key_in[i] : 0u;
  const int32_t v = valid ? val_in[i] : 0;
  const uint32_t digit = (k >> shift) & 0xffu;
  unsigned long long peers = __ballot(valid);
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const bool bit = (digit >> b) & 1u;
    const unsigned long long m = __ballot(bit);
    peers &= bit ? m : ~m;
  }
  const int rank = __popcll(peers & ((1ull << lane) - 1ull));
  if (valid && rank == 0) wcount[wave][digit] = __popcll(peers);
  __syncthreads();
  {
    int32_t run = offs[static_cast<int64_t>(threadIdx.x) * blocks + blockIdx.x];  // this thread = one digit
    for (int w = 0; w < kUpBlock / 64; ++w) {
      wbase[w][threadIdx.x] = run;
      run += wcount[w][threadIdx.x];
    }
  }
  __syncthreads();
  if (valid) {
    const int64_t pos = static_cast<int64_t>(wbase[wave][digit]) + rank;
    key_out[pos] = k;
    val_out[pos] = v;
  }
}

__global__ __launch_bounds__(kUpBlock) void k_up_gather(const float *__restrict__ x, const float *__restrict__ y,
                                                       const float *__restrict__ z, const int32_t *__restrict__ perm,
                                                       int64_t n, float *__restrict__ sx, float *__restrict__ sy,
                                                       float *__restrict__ sz, int32_t *__restrict__ inv_perm) {
  const int64_t j = static_cast<int64_t>(blockIdx.x) * kUpBlock + threadIdx.x;
  if (j >= n) return;
  const int32_t i = perm[j];
  sx[j] = x[i];
  sy[j] = y[i];
  sz[j] = z[i];
  inv_perm[i] = static_cast<int32_t>(j);
}

// bounding sphere of `span` consecutive Morton points per workgroup-slice: span = 64 (one wavefront per tile) or
// 1024 (one workgroup per group of 16 tiles).  Centre = box centre rounded to fp32, radius = largest member
// distance to that fp32 centre (fp64), inflated by 1e-6 and rounded up: it dominates every member's distance.
template <int kSpan>
__global__ __launch_bounds__(kUpBlock) void k_up_spheres(const float *__restrict__ sx, const float *__restrict__ sy,
                                                        const float *__restrict__ sz, int64_t n,
                                                        float4 *__restrict__ out, int64_t count) {
  constexpr int kLanes = kSpan >= kUpBlock ? kUpBlock : kSpan;  // lanes cooperating on one sphere
  constexpr int kPer = kSpan / kLanes;                           // points per lane
  constexpr int kSpheresPerBlock = kUpBlock / kLanes;
  __shared__ double red[7][kUpBlock / 64];
  const int sub = threadIdx.x / kLanes, l = threadIdx.x % kLanes;
  const int64_t sphere = static_cast<int64_t>(blockIdx.x) * kSpheresPerBlock + sub;
  const int64_t b = sphere * kSpan;
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  float px[kPer], py[kPer], pz[kPer];
  bool have[kPer];
#pragma unroll
  for (int q = 0; q < kPer; ++q) {
    const int64_t j = b + static_cast<int64_t>(q) * kLanes + l;
    have[q] = sphere < count && j < n;
    px[q] = have[q] ? sx[j] : 0.0f;
    py[q] = have[q] ? sy[j] : 0.0f;
    pz[q] = have[q] ? sz[j] : 0.0f;
    if (have[q]) {
      lo[0] = fmin(lo[0], static_cast<double>(px[q])); hi[0] = fmax(hi[0], static_cast<double>(px[q]));
      lo[1] = fmin(lo[1], static_cast<double>(py[q])); hi[1] = fmax(hi[1], static_cast<double>(py[q]));
      lo[2] = fmin(lo[2], static_cast<double>(pz[q])); hi[2] = fmax(hi[2], static_cast<double>(pz[q]));
    }
  }
  // reduce over the cooperating lanes: within the wavefront by shuffles, across wavefronts (kSpan = 1024) via LDS
#pragma unroll
  for (int a = 0; a < 3; ++a)
    for (int o = (kLanes < 64 ? kLanes : 64) / 2; o >= 1; o >>= 1) {
      lo[a] = fmin(lo[a], __shfl_xor(lo[a], o, 64));
      hi[a] = fmax(hi[a], __shfl_xor(hi[a], o, 64));
    }
  if (kLanes > 64) {
    if ((threadIdx.x & 63) == 0)
      for (int a = 0; a < 3; ++a) {
        red[a][threadIdx.x >> 6] = lo[a];
        red[3 + a][threadIdx.x >> 6] = hi[a];
      }
    __syncthreads();
    for (int a = 0; a < 3; ++a)
      for (int k = 0; k < kUpBlock / 64; ++k) {
        lo[a] = fmin(lo[a], red[a][k]);
        hi[a] = fmax(hi[a], red[3 + a][k]);
      }
    __syncthreads();
  }
  float c[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) c[a] = static_cast<float>(0.5 * (lo[a] + hi[a]));
  double r2 = 0.0;
#pragma unroll
  for (int q = 0; q < kPer; ++q)
    if (have[q]) {
      const double dx = static_cast<double>(px[q]) - static_cast<double>(c[0]);
      const double dy = static_cast<double>(py[q]) - static_cast<double>(c[1]);
      const double dz = static_cast<double>(pz[q]) - static_cast<double>(c[2]);
      r2 = fmax(r2, (dx * dx + dy * dy) + dz * dz);  // fmax drops NaN members: they never project anyway
    }
  for (int o = (kLanes < 64 ? kLanes : 64) / 2; o >= 1; o >>= 1) r2 = fmax(r2, __shfl_xor(r2, o, 64));
  if (kLanes > 64) {
    if ((threadIdx.x & 63) == 0) red[6][threadIdx.x >> 6] = r2;
    __syncthreads();
    for (int k = 0; k < kUpBlock / 64; ++k) r2 = fmax(r2, red[6][k]);
  }
  if (l == 0 && sphere < count) {
    float r = static_cast<float>(sqrt(r2) * (1.0 + 1e-6));
    r = isfinite(r) ? __uint_as_float(__float_as_uint(r) + 1u) : r;  // next float up
    out[sphere] = make_float4(c[0], c[1], c[2], r);
  }
}

static inline uint32_t up_blocks(int64_t n) { return static_cast<uint32_t>(std::max<int64_t>(1, (n + kUpBlock - 1) / kUpBlock)); }

// x y z of `points + i * stride` -> SoA planes (stride % 4 == 0)
__global__ __launch_bounds__(kUpBlock) void k_up_deinterleave(const uint32_t *__restrict__ aos, int64_t stride_words,
                                                             int64_t n, float *__restrict__ x, float *__restrict__ y,
                                                             float *__restrict__ z) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kUpBlock + threadIdx.x;
  if (i >= n) return;
  const uint32_t *p = aos + i * stride_words;
  x[i] = __uint_as_float(p[0]);
  y[i] = __uint_as_float(p[1]);
  z[i] = __uint_as_float(p[2]);
}

// Either (x, y, z) SoA host arrays, or `aos` = n records of `stride` bytes starting with x y z (fp32).
static int store_cloud(pcp_context *ctx, const float *x, const float *y, const float *z, int64_t n,
                       const void *aos = nullptr, int64_t stride = 0) {
  PCP_HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t sn = static_cast<size_t>(n);
  // pad every SoA plane to a multiple of 4 floats so float4 loads stay aligned
  const size_t plane = (sn + 3) & ~size_t(3);
  PCP_HIP_TRY(ctx, ctx->xyz.ensure(3 * plane + 4));
  PCP_HIP_TRY(ctx, ctx->sxyz.ensure(3 * plane + 4));
  PCP_HIP_TRY(ctx, ctx->perm.ensure(sn + 4));
  ctx->n = n;
  ctx->nonfinite_points = 0;
  ctx->sor_distances_live = false;
  ctx->sor_partial_slab = ctx->sor_partial_slabs = -1;
  ctx->n_tiles = 0;
  ctx->tile_order_live = false;
  ctx->have_intensity = false;
  ctx->nid_chunks = 0;
  ctx->colour_state_live = false;
  ctx->colour_result_live = false;
  ctx->mls_count = 0;
  ctx->vgd_next = ctx->css_next = -1;  // the streams of the smoothing stage belong to the cloud that is being replaced
  ctx->css_ball = 0.0;
  std::fill(ctx->depth_valid.begin(), ctx->depth_valid.end(), uint8_t(0));
  ctx->hull_valid.clear();
  for (int a = 0; a < 3; ++a) ctx->host_min[a] = ctx->host_max[a] = 0.0f;
  if (n == 0) return PCP_OK;
  hipStream_t st = ctx->stream;
  float *dx = ctx->xyz.p, *dy = ctx->xyz.p + plane, *dz = ctx->xyz.p + 2 * plane;
  PCP_HIP_TRY(ctx, hipMemsetAsync(ctx->sxyz.p, 0, (3 * plane + 4) * 4, st));  // plane padding reads as zeros
  DevBuf<uint32_t> raw;
  if (aos) {
    // the records cross PCIe as they are and are taken apart on the device
    PCP_HIP_TRY(ctx, raw.ensure(static_cast<size_t>(n) * static_cast<size_t>(stride) / 4 + 4));
    PCP_HIP_TRY(ctx, hipMemcpyAsync(raw.p, aos, static_cast<size_t>(n) * static_cast<size_t>(stride), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_up_deinterleave, dim3(up_blocks(n)), dim3(kUpBlock), 0, st, raw.p, stride / 4, n, dx, dy, dz);
  } else {
    PCP_HIP_TRY(ctx, hipMemcpyAsync(dx, x, sn * 4, hipMemcpyHostToDevice, st));
    PCP_HIP_TRY(ctx, hipMemcpyAsync(dy, y, sn * 4, hipMemcpyHostToDevice, st));
    PCP_HIP_TRY(ctx, hipMemcpyAsync(dz, z, sn * 4, hipMemcpyHostToDevice, st));
  }
  // ---- bounding box ----
  const uint32_t bb_blocks = std::min<uint32_t>(up_blocks(n), 1024u);
  DevBuf<float> partial;
  PCP_HIP_TRY(ctx, partial.ensure(static_cast<size_t>(bb_blocks) * 6));
  PCP_HIP_TRY(ctx, ctx->s_counter.ensure(4));
  PCP_HIP_TRY(ctx, hipMemsetAsync(ctx->s_counter.p, 0, 8, st));
  hipLaunchKernelGGL(k_up_bbox, dim3(bb_blocks), dim3(kUpBlock), 0, st, dx, dy, dz, n, partial.p, ctx->s_counter.p);
  std::vector<float> hp(static_cast<size_t>(bb_blocks) * 6);
  unsigned long long nonfinite = 0;
  PCP_HIP_TRY(ctx, hipMemcpyAsync(hp.data(), partial.p, hp.size() * 4, hipMemcpyDeviceToHost, st));
  PCP_HIP_TRY(ctx, hipMemcpyAsync(&nonfinite, ctx->s_counter.p, 8, hipMemcpyDeviceToHost, st));
  PCP_HIP_TRY(ctx, hipStreamSynchronize(st));
  ctx->nonfinite_points = static_cast<int64_t>(nonfinite);
  partial.release();
  raw.release();
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (uint32_t b = 0; b < bb_blocks; ++b)
    for (int a = 0; a < 3; ++a) {
      mn[a] = std::min(mn[a], hp[static_cast<size_t>(b) * 6 + a]);
      mx[a] = std::max(mx[a], hp[static_cast<size_t>(b) * 6 + 3 + a]);
    }
  float sc[3];
  for (int a = 0; a < 3; ++a) {
    if (mn[a] > mx[a]) mn[a] = mx[a] = 0.0f;  // no finite coordinate at all
    ctx->host_min[a] = mn[a];
    ctx->host_max[a] = mx[a];
    const float ext = mx[a] - mn[a];
    sc[a] = (ext > 0.0f && ext < FLT_MAX) ? 1023.999f / ext : 0.0f;
  }
  // ---- Morton keys + stable LSD radix sort (key, input index) ----
  const int64_t blocks = up_blocks(n);
  DevBuf<uint32_t> key_a, key_b;
  DevBuf<int32_t> val_b, hist;
  PCP_HIP_TRY(ctx, key_a.ensure(sn + 4));
  PCP_HIP_TRY(ctx, key_b.ensure(sn + 4));
  PCP_HIP_TRY(ctx, val_b.ensure(sn + 4));
  const int64_t hm = 256 * blocks;
  PCP_HIP_TRY(ctx, hist.ensure(static_cast<size_t>(hm) + 8));
  const int64_t scan_tiles = std::max<int64_t>(1, (hm + 1 + kScanTile - 1) / kScanTile);
  PCP_HIP_TRY(ctx, ctx->s_tiles.ensure(static_cast<size_t>(scan_tiles) + 4));
  const char *order_env = std::getenv("PCP_CLOUD_ORDER");
  const int32_t hilbert = !(order_env && order_env[0] == 'm');
  hipLaunchKernelGGL(k_up_keys, dim3(up_blocks(n)), dim3(kUpBlock), 0, st, dx, dy, dz, n, mn[0], mn[1], mn[2], sc[0], sc[1],
                     sc[2], key_a.p, ctx->perm.p, hilbert);
  uint32_t *kin = key_a.p, *kout = key_b.p;
  int32_t *vin = ctx->perm.p, *vout = val_b.p;
  for (int pass = 0; pass < 4; ++pass) {  // 30 key bits: 4 passes of 8 (an even count: the result lands in perm)
    const int32_t shift = 8 * pass;
    hipLaunchKernelGGL(k_up_radix_hist, dim3(static_cast<uint32_t>(blocks)), dim3(kUpBlock), 0, st, kin, n, shift, hist.p, blocks);
    PCP_HIP_TRY(ctx, hipMemsetAsync(hist.p + hm, 0, sizeof(int32_t), st));
    hipLaunchKernelGGL(k_scan_tile_sums, dim3(static_cast<uint32_t>(scan_tiles)), dim3(kScanBlock), 0, st, hist.p, hm + 1,
                       ctx->s_tiles.p);
    hipLaunchKernelGGL(k_scan_tile_offsets, dim3(1), dim3(kScanSingle), 0, st, ctx->s_tiles.p, scan_tiles,
                       static_cast<unsigned long long *>(nullptr));
    hipLaunchKernelGGL(k_scan_apply, dim3(static_cast<uint32_t>(scan_tiles)), dim3(kScanBlock), 0, st, hist.p, hm + 1,
                       ctx->s_tiles.p, hist.p);
    hipLaunchKernelGGL(k_up_radix_scatter, dim3(static_cast<uint32_t>(blocks)), dim3(kUpBlock), 0, st, kin, vin, n, shift,
                       hist.p, blocks, kout, vout);
    std::swap(kin, kout);
    std::swap(vin, vout);
  }
  PCP_HIP_TRY(ctx, hipGetLastError());
  // ---- Morton-ordered copy and the bounding spheres ----
  PCP_HIP_TRY(ctx, ctx->inv_perm.ensure(plane + 4));
  hipLaunchKernelGGL(k_up_gather, dim3(up_blocks(n)), dim3(kUpBlock), 0, st, dx, dy, dz, ctx->perm.p, n, ctx->sxyz.p,
                     ctx->sxyz.p + plane, ctx->sxyz.p + 2 * plane, ctx->inv_perm.p);
  const int64_t tiles = (n + 63) / 64, groups = (tiles + 15) / 16;
  PCP_HIP_TRY(ctx, ctx->tile_sphere.ensure(static_cast<size_t>(tiles + groups) * 4 + 4));
  float4 *sph = reinterpret_cast<float4 *>(ctx->tile_sphere.p);
  hipLaunchKernelGGL(k_up_spheres<64>, dim3(static_cast<uint32_t>((tiles + 3) / 4)), dim3(kUpBlock), 0, st, ctx->sxyz.p,
                     ctx->sxyz.p + plane, ctx->sxyz.p + 2 * plane, n, sph, tiles);
  hipLaunchKernelGGL(k_up_spheres<1024>, dim3(static_cast<uint32_t>(groups)), dim3(kUpBlock), 0, st, ctx->sxyz.p,
                     ctx->sxyz.p + plane, ctx->sxyz.p + 2 * plane, n, sph + tiles, groups);
  PCP_HIP_TRY(ctx, hipGetLastError());
  ctx->n_tiles = tiles;
  PCP_HIP_TRY(ctx, hipStreamSynchronize(st));  // the host arrays may be reused by the caller
  key_a.release();
  key_b.release();
  val_b.release();
  hist.release();
  return PCP_OK;
}

}  // namespace pcp

using namespace pcp;

extern "C" {

int pcp_abi_version(void) { return PCP_ABI_VERSION; }

int pcp_create(int32_t device, pcp_context **out) {
  if (!out) {
    set_global_error("pcp_create: out is NULL");
    return PCP_ERR_INVALID;
  }
  *out = nullptr;
  // The hull pass of hidden_points_removal keeps several keyframes in flight on streams of their own (pcp_hpr.hip
  // hpr_run_range); the HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and
  // two streams on one queue run one after the other (C3: hull pass 0.17 s on 4 queues, 0.15 s on 8).  The variable is read
  // when the runtime initialises, so this only takes effect when pcp_create is the process's first HIP call (the C++ host);
  // a host that initialises HIP earlier sets it itself (bench.py does).  Never overrides a value the caller chose.
  (void)setenv("GPU_MAX_HW_QUEUES", "8", 0);
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    set_global_error("pcp_create: no HIP device available (%s); libpcp_hip has no CPU fallback",
                     e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    return PCP_ERR_DEVICE;
  }
  if (device < 0 || device >= count) {
    set_global_error("pcp_create: device %d out of range (0..%d)", device, count - 1);
    return PCP_ERR_INVALID;
  }
  e = hipSetDevice(device);
  if (e == hipSuccess) {
    // The HIP runtime loads a translation unit's code object at the first launch of one of its kernels (deferred loading), on
    // the calling thread: tens of milliseconds of host work per unit, wherever that first launch happens to fall.  All five
    // units are loaded here, where a caller expects set-up time; the first call of every entry point then costs what the
    // others cost.
    hipFuncAttributes a;
    const hipError_t pl[] = {hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&k_up_bbox)), preload_colour(), preload_mls(),
                             preload_nid(), preload_hpr()};
    for (hipError_t x : pl)
      if (x != hipSuccess && e == hipSuccess) e = x;
  }
  if (e != hipSuccess) {
    set_global_error("pcp_create: hipSetDevice(%d) / loading the gfx950 code objects failed: %s", device, hipGetErrorString(e));
    return PCP_ERR_DEVICE;
  }
  pcp_context *ctx = new (std::nothrow) pcp_context();
  if (!ctx) {
    set_global_error("pcp_create: out of host memory");
    return PCP_ERR_NOMEM;
  }
  ctx->device = device;
  e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    set_global_error("pcp_create: hipStreamCreate failed: %s", hipGetErrorString(e));
    delete ctx;
    return PCP_ERR_DEVICE;
  }
  ctx->stream = ctx->own_stream;
  {
    // The download stream gets the highest stream priority: HIP pools its hardware queues per priority,
    // so the device-to-host copies never share a queue with (and serialise behind) the compute stream,
    // whichever stream pcp_set_stream later names and however many streams the host program owns.
    int prio_least = 0, prio_greatest = 0;
    e = hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&ctx->copy_stream, hipStreamNonBlocking, prio_greatest);
  }
  for (int k = 0; k < 2 && e == hipSuccess; ++k) {
    e = hipEventCreateWithFlags(&ctx->result_ready[k], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->copy_done[k], hipEventDisableTiming);
  }
  if (e != hipSuccess) {
    set_global_error("pcp_create: copy stream / events: %s", hipGetErrorString(e));
    pcp_destroy(ctx);
    return PCP_ERR_DEVICE;
  }
  if (hipHostMalloc(&ctx->readback, pcp_context::kReadbackBytes, hipHostMallocDefault) != hipSuccess) {
    ctx->readback = nullptr;  // the readbacks then go through pageable memory
    (void)hipGetLastError();
  }
  pcp_default_camera(&ctx->camera);
  pcp_default_cull_params(&ctx->cull);
  *out = ctx;
  return PCP_OK;
}

void pcp_destroy(pcp_context *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
  for (hipStream_t us : ctx->upload_stream)
    if (us) {
      (void)hipStreamSynchronize(us);
      (void)hipStreamDestroy(us);
    }
  for (auto e : ctx->image_event)
    if (e) (void)hipEventDestroy(e);
  if (ctx->texels_idle) (void)hipEventDestroy(ctx->texels_idle);
  for (auto &b : ctx->upload_stage) b.release();
  drain_timing(ctx);
  for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
  ctx->xyz.release();
  ctx->sxyz.release();
  ctx->perm.release();
  ctx->inv_perm.release();
  ctx->rgba_sorted.release();
  ctx->frames.release();
  ctx->images.release();
  ctx->hsv_tables.release();
  ctx->depth.release();
  ctx->depth_sq.release();
  ctx->tile_sphere.release();
  ctx->tile_mask.release();
  ctx->tile_inside.release();
  ctx->tile_work.release();
  ctx->tile_order.release();
  ctx->work_hist.release();
  ctx->group_mask.release();
  ctx->top_score.release();
  ctx->top_rgb.release();
  ctx->top_frame.release();
  ctx->view_count.release();
  ctx->rgba2[0].release();
  ctx->rgba2[1].release();
  for (int k = 0; k < 2; ++k) {
    if (ctx->result_ready[k]) (void)hipEventDestroy(ctx->result_ready[k]);
    if (ctx->copy_done[k]) (void)hipEventDestroy(ctx->copy_done[k]);
  }
  if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
  ctx->s_cell.release();
  ctx->s_pixel.release();
  ctx->s_range.release();
  ctx->s_cam.release();
  ctx->s_keep.release();
  ctx->s_u32.release();
  ctx->s_counter.release();
  ctx->s_tiles.release();
  ctx->g_cell.release();
  ctx->g_rank.release();
  ctx->g_start.release();
  ctx->g_occ.release();
  ctx->g_occ_rank.release();
  ctx->g_order.release();
  ctx->g_xyz.release();
  ctx->m_tmp.release();
  ctx->s_dist.release();
  ctx->css_dist.release();
  ctx->s_kth.release();
  ctx->css_words.release();
  ctx->m_state.release();
  ctx->m_flag.release();
  ctx->intensity.release();
  for (auto &lane : ctx->hpr_lane) lane.release();
  if (ctx->hpr_fork) (void)hipEventDestroy(ctx->hpr_fork);
  for (auto &e : ctx->hpr_join)
    if (e) (void)hipEventDestroy(e);
  ctx->hull_bits.release();
  ctx->nid_pts.release();
  ctx->nid_chunk_kf.release();
  ctx->nid_hist.release();
  ctx->m_sums.release();
  ctx->c_index.release();
  ctx->c_xyz.release();
  ctx->c_mark.release();
  ctx->c_where.release();
  ctx->c_xyz2.release();
  ctx->v_bitmap.release();
  ctx->v_occ.release();
  ctx->v_rank.release();
  ctx->v_plane.release();
  ctx->v_vox.release();
  ctx->v_offsets.release();
  ctx->mls_xyz.release();
  ctx->mls_normal.release();
  ctx->mls_curv.release();
  ctx->mls_index.release();
  ctx->mls_alt_xyz.release();
  ctx->mls_alt_normal.release();
  ctx->mls_alt_curv.release();
  ctx->mls_alt_index.release();
  if (ctx->readback) (void)hipHostFree(ctx->readback);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  delete ctx;
}

const char *pcp_last_error(const pcp_context *ctx) { return ctx ? ctx->error.c_str() : g_error.c_str(); }

int pcp_set_stream(pcp_context *ctx, void *hip_stream) {
  if (!ctx) return PCP_ERR_INVALID;
  PCP_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  ctx->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->own_stream;
  return PCP_OK;
}

int pcp_synchronize(pcp_context *ctx) {
  if (!ctx) return PCP_ERR_INVALID;
  for (hipStream_t us : ctx->upload_stream)
    if (us) PCP_HIP_TRY(ctx, hipStreamSynchronize(us));
  std::fill(ctx->image_pending.begin(), ctx->image_pending.end(), uint8_t(0));
  PCP_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  PCP_HIP_TRY(ctx, hipStreamSynchronize(ctx->copy_stream));
  ctx->copy_pending[0] = ctx->copy_pending[1] = false;
  return PCP_OK;
}

// PointCloudProcessor.cpp:57-62,525
void pcp_default_camera(pcp_camera *cam) {
  cam->fx = 4818.200388954926;
  cam->fy = 4819.10345841615;
  cam->cx = 2032.4178620390019;
  cam->cy = 1535.1895959282901;
  cam->k1 = 0.003043514741045163;
  cam->k2 = 0.06634739187544138;
  cam->p1 = -0.000217681797407554;
  cam->p2 = -0.0006654964142658197;
  cam->k3 = 0.0;
  cam->image_width = 4096;
  cam->image_height = 3000;
  cam->cull_width = 4096;
  cam->cull_height = 3000;
}

// view_culling.hpp:12-15, view_culling.cpp:63,157
void pcp_default_cull_params(pcp_cull_params *p) {
  p->enable_depth_buffer_culling = 1;
  p->downsample_factor = 14;
  p->depth_slack = 0.05;
  p->cull_mode = PCP_CULL_ZBUFFER;
  p->match_mode = PCP_MATCH_ROUNDTRIP;  // the reference's arithmetic (PointCloudProcessor.cpp:555,571-579); +2 % of a step
  p->hpr_flip_radius = 90000.0;         // view_culling.hpp:14
}

// PointCloudProcessor.cpp:67-86
void pcp_default_mls_params(pcp_mls_params *p) {
  p->search_radius = 0.03;
  p->sqr_gauss_param = 0.0009;
  p->polynomial_order = 2;
  p->compute_normals = 1;
  p->upsampling = 3;
  p->vgd_iterations = 4;
  p->vgd_voxel_size = 0.001f;
  p->sor_mean_k = 60;
  p->sor_std_mul = 0.7;
}

int pcp_set_camera(pcp_context *ctx, const pcp_camera *cam, const pcp_cull_params *cull) {
  if (!ctx || !cam) return set_error(ctx, PCP_ERR_INVALID, "pcp_set_camera: NULL argument");
  pcp_cull_params cp;
  if (cull)
    cp = *cull;
  else
    pcp_default_cull_params(&cp);
  if (cam->image_width <= 0 || cam->image_height <= 0 || cam->cull_width <= 0 || cam->cull_height <= 0)
    return set_error(ctx, PCP_ERR_INVALID, "pcp_set_camera: image / cull size must be positive");
  if (cp.downsample_factor <= 0) return set_error(ctx, PCP_ERR_INVALID, "pcp_set_camera: downsample_factor must be > 0");
  if (cp.cull_mode != PCP_CULL_ZBUFFER && cp.cull_mode != PCP_CULL_HPR_CANDIDATES && cp.cull_mode != PCP_CULL_HPR)
    return set_error(ctx, PCP_ERR_INVALID, "pcp_set_camera: unknown cull_mode %d", cp.cull_mode);
  if (cp.cull_mode == PCP_CULL_HPR && !(cp.hpr_flip_radius > 0.0 && cp.hpr_flip_radius < 1e300))
    return set_error(ctx, PCP_ERR_INVALID, "pcp_set_camera: hpr_flip_radius must be positive and finite");
  if (cp.match_mode != PCP_MATCH_IDENTITY && cp.match_mode != PCP_MATCH_ROUNDTRIP)
    return set_error(ctx, PCP_ERR_INVALID, "pcp_set_camera: unknown match_mode %d", cp.match_mode);
  if (cam->cull_width > (1 << 24) || cam->cull_height > (1 << 24))
    return set_error(ctx, PCP_ERR_INVALID, "pcp_set_camera: cull size above 2^24 is not supported");
  if (static_cast<int64_t>(cam->image_width) * cam->image_height >= (int64_t(1) << 31))
    return set_error(ctx, PCP_ERR_INVALID, "pcp_set_camera: image too large for int32 pixel indices");
  ctx->camera = *cam;
  ctx->cull = cp;
  DevCamera &d = ctx->dcam;
  d.fx = cam->fx;
  d.fy = cam->fy;
  d.cx = cam->cx;
  d.cy = cam->cy;
  d.k1 = cam->k1;
  d.k2 = cam->k2;
  d.p1 = cam->p1;
  d.p2 = cam->p2;
  d.k3 = cam->k3;
  d.slack = cp.depth_slack;
  d.ds = cp.downsample_factor;
  d.ds_f = static_cast<float>(cp.downsample_factor);
  d.ds_rcp = 1.0f / d.ds_f;  // IEEE division on the host: the correctly rounded reciprocal
  d.ds_fast = (d.ds_f >= 0x1p-20f && d.ds_f <= 0x1p20f) ? 1 : 0;
  if (const char *e = std::getenv("PCP_DISABLE_FAST_EXACT")) d.ds_fast = (e[0] == '1') ? 0 : d.ds_fast;
  d.img_w = cam->image_width;
  d.img_h = cam->image_height;
  d.cull_w = cam->cull_width;
  d.cull_h = cam->cull_height;
  d.img_wd = static_cast<double>(cam->image_width);
  d.img_hd = static_cast<double>(cam->image_height);
  d.cull_wf = static_cast<float>(cam->cull_width);  // exact: <= 2^24
  d.cull_hf = static_cast<float>(cam->cull_height);
  d.mw = cam->cull_width / cp.downsample_factor;
  d.mh = cam->cull_height / cp.downsample_factor;
  // the kernels of the colour path only know the candidate filter; the hull is a stage of its own (pcp_hpr.hip)
  d.cull_mode = cp.cull_mode == PCP_CULL_HPR ? PCP_CULL_HPR_CANDIDATES : cp.cull_mode;
  d.match_mode = cp.match_mode;
  d.cull_wd = static_cast<double>(cam->cull_width);
  d.cull_hd = static_cast<double>(cam->cull_height);
  {
    // kdtree.radiusSearch(searchPoint, epsilon = 1e-5f): PCL squares the radius in fp64 and hands FLANN the fp32
    // value [upstream KdTreeFLANN::radiusSearch], PointCloudProcessor.cpp:482,571
    const double eps = static_cast<double>(1e-5f);
    d.match_r2 = static_cast<float>(eps * eps);
  }
  // hidden_points_removal has no depth buffer: every candidate is kept
  d.enable_zbuf = (cp.enable_depth_buffer_culling && cp.cull_mode == PCP_CULL_ZBUFFER) ? 1 : 0;
  // conservative fp32 rejection test (pcp_device.hpp): parameters
  d.pretest = 1;
  d.qfx = static_cast<float>(cam->fx);
  d.qfy = static_cast<float>(cam->fy);
  d.qcx = static_cast<float>(cam->cx);
  d.qcy = static_cast<float>(cam->cy);
  d.qk1 = static_cast<float>(cam->k1);
  d.qk2 = static_cast<float>(cam->k2);
  d.qk3 = static_cast<float>(cam->k3);
  d.qp1 = static_cast<float>(cam->p1);
  d.qp2 = static_cast<float>(cam->p2);
  {
    // cell rule accepts trunc(f32(u)/ds) in [0, mw) (depth buffer on) or [0, cull_w) (off);
    // pixel rule accepts (int)u in [0, img_w).  Box = union, +-0.5 px.
    // (hidden_points_removal's rule accepts (int)u in [0, cull_w): inside the same box with cw = cull_w)
    const float ds = static_cast<float>(cp.downsample_factor);
    const bool hpr = cp.cull_mode != PCP_CULL_ZBUFFER;
    const float cw = hpr ? static_cast<float>(d.cull_w) : ds * static_cast<float>(d.enable_zbuf ? d.mw : d.cull_w);
    const float ch = hpr ? static_cast<float>(d.cull_h) : ds * static_cast<float>(d.enable_zbuf ? d.mh : d.cull_h);
    d.u_lo = -(ds + 0.5f);
    d.v_lo = -(ds + 0.5f);
    d.u_hi = std::max(cw, static_cast<float>(d.img_w)) + 0.5f;
    d.v_hi = std::max(ch, static_cast<float>(d.img_h)) + 0.5f;
  }
  if (const char *e = std::getenv("PCP_DISABLE_PRETEST")) d.pretest = (e[0] == '1') ? 0 : 1;

  ctx->have_camera = true;
  for (hipStream_t us : ctx->upload_stream)
    if (us) (void)hipStreamSynchronize(us);  // uploads sized by the previous camera
  std::fill(ctx->image_pending.begin(), ctx->image_pending.end(), uint8_t(0));
  // images / depth maps are sized by the camera: drop them
  ctx->image_set.assign(ctx->image_set.size(), 0);
  ctx->mask_set.assign(ctx->mask_set.size(), 0);
  std::fill(ctx->depth_valid.begin(), ctx->depth_valid.end(), uint8_t(0));
  ctx->hull_valid.clear();
  ctx->colour_state_live = false;
  ctx->colour_result_live = false;
  return PCP_OK;
}

int pcp_upload_cloud(pcp_context *ctx, const float *x, const float *y, const float *z, int64_t n) {
  if (!ctx) return PCP_ERR_INVALID;
  if (n < 0 || n >= (int64_t(1) << 31)) return set_error(ctx, PCP_ERR_INVALID, "pcp_upload_cloud: n=%lld out of range", (long long)n);
  if (n > 0 && (!x || !y || !z)) return set_error(ctx, PCP_ERR_INVALID, "pcp_upload_cloud: NULL coordinate array");
  return store_cloud(ctx, x, y, z, n);
}

int pcp_upload_cloud_aos(pcp_context *ctx, const void *points, int64_t n, int64_t stride_bytes) {
  if (!ctx) return PCP_ERR_INVALID;
  if (n < 0 || n >= (int64_t(1) << 31)) return set_error(ctx, PCP_ERR_INVALID, "pcp_upload_cloud_aos: n out of range");
  if (stride_bytes < 12) return set_error(ctx, PCP_ERR_INVALID, "pcp_upload_cloud_aos: stride must be >= 12 bytes");
  if (n > 0 && !points) return set_error(ctx, PCP_ERR_INVALID, "pcp_upload_cloud_aos: NULL points");
  if (stride_bytes % 4 == 0 && (reinterpret_cast<uintptr_t>(points) & 3u) == 0)
    return store_cloud(ctx, nullptr, nullptr, nullptr, n, points, stride_bytes);
  // odd strides / unaligned records: take them apart on the host
  std::vector<float> x(static_cast<size_t>(n)), y(static_cast<size_t>(n)), z(static_cast<size_t>(n));
  const uint8_t *base = static_cast<const uint8_t *>(points);
  for (int64_t i = 0; i < n; ++i) {
    float v[3];
    std::memcpy(v, base + i * stride_bytes, 12);
    x[static_cast<size_t>(i)] = v[0];
    y[static_cast<size_t>(i)] = v[1];
    z[static_cast<size_t>(i)] = v[2];
  }
  return store_cloud(ctx, x.data(), y.data(), z.data(), n);
}

int64_t pcp_cloud_size(const pcp_context *ctx) { return ctx ? ctx->n : -1; }

int pcp_pose_to_matrices(const pcp_pose *pose, const double *T_opt, float w2c[12], float c2w[12]) {
  if (!pose || !w2c || !c2w) {
    set_global_error("pcp_pose_to_matrices: NULL argument");
    return PCP_ERR_INVALID;
  }
  matrices_from_pose(*pose, T_opt, w2c, c2w);
  return PCP_OK;
}

int pcp_set_frames(pcp_context *ctx, const pcp_pose *poses, int32_t n_frames, const double *T_opt,
                   int32_t T_opt_stride) {
  if (!ctx) return PCP_ERR_INVALID;
  if (n_frames < 0 || (n_frames > 0 && !poses)) return set_error(ctx, PCP_ERR_INVALID, "pcp_set_frames: bad poses");
  if (T_opt && T_opt_stride != 0 && T_opt_stride != 16)
    return set_error(ctx, PCP_ERR_INVALID, "pcp_set_frames: T_opt_stride must be 0 (global) or 16 (per keyframe)");
  PCP_HIP_TRY(ctx, hipSetDevice(ctx->device));
  for (hipStream_t us : ctx->upload_stream)
    if (us) PCP_HIP_TRY(ctx, hipStreamSynchronize(us));  // uploads of the previous keyframe set
  std::fill(ctx->image_pending.begin(), ctx->image_pending.end(), uint8_t(0));
  ctx->poses.assign(poses, poses + n_frames);
  ctx->hframes.resize(static_cast<size_t>(n_frames));
  for (int32_t f = 0; f < n_frames; ++f) {
    DevFrame &d = ctx->hframes[static_cast<size_t>(f)];
    const double *T = T_opt ? T_opt + static_cast<int64_t>(T_opt_stride) * f : nullptr;
    matrices_from_pose(poses[f], T, d.w2c, d.c2w);
    invert_affine_f32(d.c2w, d.c2w_inv);  // transformation_c2w_optimized.inverse(), PointCloudProcessor.cpp:578
    d.pad_[0] = d.pad_[1] = d.pad_[2] = d.pad_[3] = 0.0f;
    d.px = poses[f].x;
    d.py = poses[f].y;
    d.pz = poses[f].z;
    {
      // spectral norm of the 3x3 linear part L of w2c (as stored, fp32): lambda_max(L^T L)
      // <= ||L^T L||_inf; 1 for a rotation, larger for un-normalised quaternions
      double g[3][3];
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) {
          g[a][b] = 0.0;
          for (int k = 0; k < 3; ++k) g[a][b] += static_cast<double>(d.w2c[4 * k + a]) * static_cast<double>(d.w2c[4 * k + b]);
        }
      double rowmax = 0.0;
      for (int a = 0; a < 3; ++a) rowmax = std::max(rowmax, std::fabs(g[a][0]) + std::fabs(g[a][1]) + std::fabs(g[a][2]));
      d.norm_bound = std::sqrt(rowmax) * (1.0 + 1e-9);
    }
  }
  {
    // camera coordinates of a point below 2^40 in every coordinate are finite under such matrices: the batched passes then
    // skip the per-visit test for non-finite operands in front of the short division (pcp_device.hpp divide_xy_by_z)
    bool bounded = true;
    for (int32_t f = 0; f < n_frames; ++f)
      for (int k = 0; k < 12; ++k) bounded = bounded && std::fabs(ctx->hframes[static_cast<size_t>(f)].w2c[k]) <= 0x1p40f;
    ctx->dcam.frames_bounded = bounded ? 1 : 0;
  }
  PCP_HIP_TRY(ctx, ctx->frames.ensure(static_cast<size_t>(n_frames) + 1));
  if (n_frames > 0)
    PCP_HIP_TRY(ctx, hipMemcpyAsync(ctx->frames.p, ctx->hframes.data(), sizeof(DevFrame) * n_frames,
                                    hipMemcpyHostToDevice, ctx->stream));
  PCP_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  ctx->n_frames = n_frames;
  ctx->image_set.assign(static_cast<size_t>(n_frames), 0);
  ctx->mask_set.assign(static_cast<size_t>(n_frames), 0);
  ctx->depth_valid.assign(static_cast<size_t>(n_frames), 0);
  ctx->hull_valid.clear();
  ctx->colour_state_live = false;
  ctx->colour_result_live = false;
  return PCP_OK;
}

int32_t pcp_frame_count(const pcp_context *ctx) { return ctx ? ctx->n_frames : -1; }

int pcp_timing_enable(pcp_context *ctx, int32_t on) {
  if (!ctx) return PCP_ERR_INVALID;
  PCP_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  drain_timing(ctx);
  ctx->timing = on != 0;
  return PCP_OK;
}

int pcp_timing_reset(pcp_context *ctx) {
  if (!ctx) return PCP_ERR_INVALID;
  PCP_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  drain_timing(ctx);
  for (auto &s : ctx->slots) s = TimingSlot{};
  return PCP_OK;
}

int pcp_timing_get(pcp_context *ctx, int32_t kernel_id, double *total_ms, int64_t *launches) {
  if (!ctx) return PCP_ERR_INVALID;
  if (kernel_id < 0 || kernel_id >= PCP_K_COUNT) return set_error(ctx, PCP_ERR_RANGE, "pcp_timing_get: bad kernel id %d", kernel_id);
  PCP_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  drain_timing(ctx);
  if (total_ms) *total_ms = ctx->slots[kernel_id].total_ms;
  if (launches) *launches = ctx->slots[kernel_id].launches;
  return PCP_OK;
}

const char *pcp_kernel_name(int32_t kernel_id) {
  static const char *names[PCP_K_COUNT] = {"project_frame", "depth_pass", "colour_pass", "visibility", "mls_grid",
                                           "mls_fit",       "misc",       "sor",         "mls_voxel",   "tile_mask",
                                           "nid_hist",      "hpr"};
  return (kernel_id >= 0 && kernel_id < PCP_K_COUNT) ? names[kernel_id] : "?";
}

}  // extern "C"
