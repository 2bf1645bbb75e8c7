"""Where do the idle lanes of k_sor_select come from?  The run lengths of the 9 runs of every query on the synthetic scene (the
grid sized as sor_run sizes it), and the lane utilisation that the raggedness of those runs ALONE would give: one lane per
query (64 consecutive cell-sorted queries per wavefront: the shipped kernel), 4 or 8 lanes per query.  CPU only.
python profiles/sim_lane_util.py [points]"""
import os, numpy as np, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudprocessor_amd import synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
x, y, z, _ = synth.make_cloud(N)
mn = np.array([x.min(), y.min(), z.min()]); mx = np.array([x.max(), y.max(), z.max()])
vol = np.prod(np.maximum(mx - mn, 1e-3))
def grid(cell):
    ix = np.floor((x - mn[0]) / cell).astype(np.int64); iy = np.floor((y - mn[1]) / cell).astype(np.int64); iz = np.floor((z - mn[2]) / cell).astype(np.int64)
    nx, ny, nz = ix.max() + 1, iy.max() + 1, iz.max() + 1
    return ix, iy, iz, nx, ny, nz
# density probe as sor_run: every 8th point
stride = 8
pc = np.cbrt(vol / (N / stride) * 4.0)
ix, iy, iz, nx, ny, nz = grid(pc)
key = ((iz * ny + iy) * nx + ix)[::stride]
occ = len(np.unique(key))
per_area = N / (occ * pc * pc)
k = 60
cell = np.sqrt(1.5 * (k + 1) / (np.pi * per_area))
print("cell", cell, "per_area", per_area)
ix, iy, iz, nx, ny, nz = grid(cell)
key = (iz * ny + iy) * nx + ix
order = np.argsort(key, kind="stable")
ks = key[order]
ncell = nx * ny * nz
counts = np.bincount(ks, minlength=ncell + 1)
start = np.concatenate([[0], np.cumsum(counts)])
cx, cy, cz = ix[order], iy[order], iz[order]
# run lengths for the 9 runs of each query
runs = np.zeros((9, N), np.int32)
r = 0
for dz in (-1, 0, 1):
    for dy in (-1, 0, 1):
        zz, yy = cz + dz, cy + dy
        ok = (zz >= 0) & (zz < nz) & (yy >= 0) & (yy < ny)
        x0 = np.maximum(cx - 1, 0); x1 = np.minimum(cx + 1, nx - 1)
        base = (np.clip(zz, 0, nz - 1) * ny + np.clip(yy, 0, ny - 1)) * nx
        runs[r] = np.where(ok, start[base + x1 + 1] - start[base + x0], 0)
        r += 1
tot = runs.sum(0)
print("mean candidates", tot.mean(), "points per occupied cell", N / (counts > 0).sum())
def util_lane(group):
    m = (N // group) * group
    # current scheme: trips per run = ceil(len/4), wave executes max over lanes per run
    t = np.ceil(runs[:, :m] / 4.0).reshape(9, -1, group)
    wave = t.max(2).sum(0)             # per-wave trips (sum over runs of max over lanes)
    useful = (runs[:, :m] / 4.0).reshape(9, -1, group).sum(2).sum(0) / group
    return useful.sum() / wave.sum()
def util_quad():
    m = (N // 16) * 16
    t = np.ceil(runs[:, :m] / 16.0).reshape(9, -1, 16)   # trips of a quad-lane for query
    wave = t.max(2).sum(0)
    useful = (runs[:, :m] / 16.0).reshape(9, -1, 16).sum(2).sum(0) / 16
    return useful.sum() / wave.sum()
def util_oct():
    m = (N // 8) * 8
    t = np.ceil(runs[:, :m] / 32.0).reshape(9, -1, 8)
    wave = t.max(2).sum(0)
    useful = (runs[:, :m] / 32.0).reshape(9, -1, 8).sum(2).sum(0) / 8
    return useful.sum() / wave.sum()
print("lane utilisation, 1 lane per query (64 per wave):", round(util_lane(64), 3))
print("4 lanes per query (16 per wave):", round(util_quad(), 3))
print("8 lanes per query (8 per wave):", round(util_oct(), 3))
