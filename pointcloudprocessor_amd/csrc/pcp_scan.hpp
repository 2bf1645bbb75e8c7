// pcp_scan.hpp -- device-wide exclusive prefix sum over int32 counts (three
// launches: tile sums, single-block scan of the sums, per-tile apply).  Used by
// the MLS cell binning; 64-wide wavefront scans through __shfl_up.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace pcp {

constexpr int kScanBlock = 256;
constexpr int kScanTile = 1024;  // 4 items per lane

__device__ __forceinline__ int32_t scan_block_exclusive(int32_t v, int32_t *total, int32_t *wave_sum /* [4] LDS */) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int32_t incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int32_t t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wave_sum[wid] = incl;
  __syncthreads();
  int32_t base = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < kScanBlock / 64; ++k) {
    if (k < wid) base += wave_sum[k];
    tot += wave_sum[k];
  }
  __syncthreads();
  *total = tot;
  return base + incl - v;
}

// A launch holds fewer than 2^32 work-items (the dispatch packet counts them in 32 bits), i.e. fewer than 2^24
// workgroups of 256: the per-tile kernels walk the tiles with the grid's stride and their launches are capped
// (scan_grid), so that arrays of more than 2^34 entries -- the voxel bitmap of a map at 1 mm -- are covered.
constexpr int64_t kScanMaxGrid = int64_t(1) << 22;
static inline uint32_t scan_grid(int64_t tiles) {
  return static_cast<uint32_t>(tiles < 1 ? 1 : (tiles > kScanMaxGrid ? kScanMaxGrid : tiles));
}

static __global__ __launch_bounds__(kScanBlock) void k_scan_tile_sums(const int32_t *__restrict__ in, int64_t n,
                                                                      int32_t *__restrict__ tile_sum) {
  __shared__ int32_t ws[kScanBlock / 64];
  const int64_t tiles = (n + kScanTile - 1) / kScanTile;
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t base = tile * kScanTile + threadIdx.x * 4;
    int32_t c = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (base + k < n) c += in[base + k];
    int32_t total;
    (void)scan_block_exclusive(c, &total, ws);
    if (threadIdx.x == 0) tile_sum[tile] = total;
  }
}

// Exclusive scan of `tiles` counts in place by ONE workgroup of kScanSingle threads, grand total to *total (nullable): two
// sweeps over the counts with the wavefronts working independently inside a sweep (sweep 1: sums per wavefront and chunk of
// 1024 counts; their prefix by one wavefront; sweep 2: the counts again, scanned inside the wavefront on top of that prefix)
// -- three barriers per 8 192 counts.  The first form (a block scan with two barriers per 256 counts) took 25 us for the
// 9 766 tiles of a 10 M-point cloud, once or twice per keyframe of a cull and several times per smoothing chain.
constexpr int kScanSingle = 1024;
__device__ __forceinline__ void scan_single_block(int32_t *__restrict__ data, int64_t tiles, unsigned long long *__restrict__ total) {
  constexpr int kWaves = kScanSingle / 64, kChunks = 8;  // 8 192 counts per round of three barriers (small, so that ordinary sizes take several rounds and the tests cover the carry)
  __shared__ long long ws[kChunks * kWaves];
  __shared__ long long carry_s;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  long long carry = 0;
  for (int64_t sb = 0; sb < tiles; sb += static_cast<int64_t>(kChunks) * kScanSingle) {
    const int nch = static_cast<int>(min(static_cast<int64_t>(kChunks), (tiles - sb + kScanSingle - 1) / kScanSingle));
    for (int c = 0; c < nch; ++c) {
      const int64_t i = sb + static_cast<int64_t>(c) * kScanSingle + threadIdx.x;
      int32_t v = i < tiles ? data[i] : 0;
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
      if (lane == 0) ws[c * kWaves + wid] = v;
    }
    __syncthreads();
    if (wid == 0) {
      const int entries = nch * kWaves, per = (entries + 63) / 64;
      long long s_ = 0;
      for (int k = 0; k < per; ++k) {
        const int e = lane * per + k;
        if (e < entries) s_ += ws[e];
      }
      long long incl = s_;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const long long t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
      }
      long long run = carry + incl - s_;
      for (int k = 0; k < per; ++k) {
        const int e = lane * per + k;
        if (e < entries) {
          const long long v = ws[e];
          ws[e] = run;
          run += v;
        }
      }
      if (lane == 63) carry_s = carry + incl;
    }
    __syncthreads();
    for (int c = 0; c < nch; ++c) {
      const int64_t i = sb + static_cast<int64_t>(c) * kScanSingle + threadIdx.x;
      const int32_t v = i < tiles ? data[i] : 0;
      int32_t incl = v;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int32_t t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
      }
      if (i < tiles) data[i] = static_cast<int32_t>(ws[c * kWaves + wid] + incl - v);
    }
    carry = carry_s;
    __syncthreads();
  }
  if (threadIdx.x == 0 && total) *total = static_cast<unsigned long long>(carry);
}

// single block: exclusive scan of tile sums in place; grand total to *total
static __global__ __launch_bounds__(kScanSingle) void k_scan_tile_offsets(int32_t *__restrict__ tile_sum, int64_t tiles,
                                                                          unsigned long long *__restrict__ total) {
  scan_single_block(tile_sum, tiles, total);
}

// out[i] = tile_offset[tile] + exclusive prefix inside the tile (in and out may alias)
static __global__ __launch_bounds__(kScanBlock) void k_scan_apply(const int32_t *in, int64_t n,
                                                                  const int32_t *__restrict__ tile_offset,
                                                                  int32_t *out) {
  __shared__ int32_t ws[kScanBlock / 64];
  const int64_t tiles = (n + kScanTile - 1) / kScanTile;
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t base = tile * kScanTile + threadIdx.x * 4;
    int32_t v[4];
    int32_t c = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[k] = base + k < n ? in[base + k] : 0;
      c += v[k];
    }
    int32_t total;
    int32_t run = tile_offset[tile] + scan_block_exclusive(c, &total, ws);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (base + k < n) out[base + k] = run;
      run += v[k];
    }
  }
}

}  // namespace pcp
