"""The big shapes of BASELINE.json under pytest: one rank's share of configs[3] (6.25 M points x
1024 keyframes) and of configs[4] (12.5 M points x 2048 keyframes + segmentation masks) at 1920x1080, and the MLS /
SOR stages at 10 M points.  The oracle cannot run these sizes in seconds, so they are checked through size-independent
properties -- run-to-run determinism, invariance under point-index sharding with MIN-merged depth maps (the multi-GPU
scheme, incl. the per-keyframe calls on PCP_DEPTH_BATCHED) -- plus oracle comparisons on sub-samples whose result does
not depend on the rest of the cloud."""
import numpy as np
import pytest

from conftest import cam_struct

pytestmark = pytest.mark.gpu


def _engine(cd, x, y, z, poses, images, masks=None, batched_depth=False):
    from pointcloudprocessor_amd import pipeline

    eng = pipeline.HipEngine(0)
    eng.configure(cd)
    eng.upload_cloud(x, y, z)
    eng.ctx.set_frames(poses)
    for f in range(len(poses)):
        eng.ctx.upload_image(f, images[f % len(images)])
        if masks is not None:
            eng.ctx.upload_mask(f, masks[f % len(masks)])
    if batched_depth:
        eng.ctx.set_depth_source(True)
    return eng


def _sharding_properties(cd, x, y, z, poses, images, masks, probe_frames):
    """determinism; 2 index shards + MIN-merged maps == unsharded, for the colours and for the per-keyframe calls."""
    import torch

    from pointcloudprocessor_amd import pipeline

    n = len(x)
    eng = _engine(cd, x, y, z, poses, images, masks)
    a = eng.ctx.colorize()
    b = eng.ctx.colorize()
    assert np.array_equal(a["rgb"], b["rgb"]) and np.array_equal(a["has"], b["has"])
    assert 0.2 * n < int(a["has"].sum()) < n
    depth_full = {f: eng.ctx.download_depth_map(f) for f in probe_frames}
    vis_full = {f: eng.ctx.frame_visible(f) for f in probe_frames[:2]}
    keep_full = {f: eng.ctx.cull_frame(f)[0] for f in probe_frames[:2]}
    eng.close()
    engs, maps = [], []
    for r in range(2):
        lo, hi = pipeline.shard_bounds(n, r, 2)
        e = _engine(cd, x[lo:hi], y[lo:hi], z[lo:hi], poses, images, masks, batched_depth=True)
        e.depth_pass()
        engs.append(e)
        maps.append(e.depth_maps_tensor())
    merged = torch.minimum(maps[0], maps[1])  # the all-reduce(MIN)
    for f, ref in depth_full.items():
        cells = ref.size
        assert np.array_equal(merged[f * cells:(f + 1) * cells].cpu().numpy().view(np.uint32), ref.reshape(-1).view(np.uint32)), f
    for t in maps:
        t.copy_(merged)
    torch.cuda.synchronize()
    parts = [e.colour_from_depth() for e in engs]
    assert np.array_equal(np.concatenate([q["rgb"] for q in parts]), a["rgb"])
    assert np.array_equal(np.concatenate([q["has"] for q in parts]), a["has"])
    # the per-keyframe calls of the shards (PCP_DEPTH_BATCHED: merged maps) stitch to the unsharded outputs
    for f in probe_frames[:2]:
        lo1 = pipeline.shard_bounds(n, 1, 2)[0]
        v = [e.ctx.frame_visible(f) for e in engs]
        assert np.array_equal(np.concatenate([v[0]["index"], v[1]["index"] + lo1]), vis_full[f]["index"]), f
        for k in ("rgb", "mask", "xyz_cam", "xyz_world"):
            assert np.array_equal(np.concatenate([v[0][k], v[1][k]]), vis_full[f][k]), (f, k)
        keep = np.concatenate([e.ctx.cull_frame(f)[0] for e in engs])
        assert np.array_equal(keep, keep_full[f]), f
    for e in engs:
        e.close()
    return a


def test_config3_rank_share_6M25_x_1024(oracle):
    """One rank's share of configs[3] (50 M x 1024 over 8 GPUs): properties at full size, and a 200 k-point
    sub-sample x all 1024 keyframes against the oracle (depth maps of a sub-sample are its own: bit-exact colours)."""
    from pointcloudprocessor_amd import capi, synth

    cd = synth.camera_dict("cfg")
    N, F = 6_250_000, 1024
    x, y, z, _ = synth.make_cloud(N)
    poses, _ = synth.make_trajectory(F)
    images = [synth.make_image(f, cd["image_width"], cd["image_height"]) for f in range(8)]
    _sharding_properties(cd, x, y, z, poses, images, None, [0, 517, 1023])
    n = 200_000
    ctx = capi.Context(0)
    ctx.set_camera(cam_struct(capi, cd))
    ctx.upload_cloud(x[:n], y[:n], z[:n])
    ctx.set_frames(poses)
    for f in range(F):
        ctx.upload_image(f, images[f % 8])
    got = ctx.colorize()
    ref = oracle.colorize(cam_struct(oracle, cd), oracle.default_cull_params(), x[:n], y[:n], z[:n], poses,
                          [images[f % 8] for f in range(F)], threads=4, want_top=False)  # 3 x 1024 short OpenMP regions:
    # more threads than the box's CPU share turn every barrier into a scheduler wait (260 s with 16 threads)
    assert np.array_equal(got["rgb"], ref["rgb"]) and np.array_equal(got["has"], ref["has"])
    assert ref["has"].sum() > 60_000
    ctx.close()


def test_config4_rank_share_12M5_x_2048_with_masks():
    """One rank's share of configs[4] (100 M x 2048 + masks over 8 GPUs): 17 GB of texels per context, 64 mask words
    per tile; determinism, sharding invariance, masked per-keyframe dumps stitched from the shards."""
    from pointcloudprocessor_amd import synth

    cd = synth.camera_dict("cfg")
    N, F = 12_500_000, 2048
    x, y, z, _ = synth.make_cloud(N)
    poses, _ = synth.make_trajectory(F)
    W, H = cd["image_width"], cd["image_height"]
    images = [synth.make_image(f, W, H) for f in range(8)]
    masks = [synth.make_mask(f, W, H) for f in range(8)]
    a = _sharding_properties(cd, x, y, z, poses, images, masks, [5, 1030, 2047])
    assert int(a["has"].sum()) > 0.5 * N


@pytest.fixture(scope="module")
def cloud10m():
    from pointcloudprocessor_amd import synth

    x, y, z, _ = synth.make_cloud(10_000_000)
    return x, y, z


def test_mls_10M_shards_determinism_and_oracle_slab(cloud10m, oracle):
    """MLS (NONE) at configs[2]'s size: run-to-run equality, query shards concatenate to the full result, and the
    points of a slab whose whole neighbourhood lies inside the slab against the oracle run on the slab alone."""
    from pointcloudprocessor_amd import capi

    x, y, z = cloud10m
    n = len(x)
    ctx = capi.Context(0)
    ctx.set_camera(capi.default_camera())
    ctx.upload_cloud(x, y, z)
    mp = capi.default_mls_params()
    mp.upsampling = 0
    full = ctx.mls_fetch(ctx.mls_process(mp))
    again = ctx.mls_fetch(ctx.mls_process(mp))
    for k in ("index", "xyz", "normal", "curvature"):
        assert np.array_equal(full[k], again[k]), k
    assert len(full["index"]) > 0.99 * n and np.all(np.diff(full["index"]) > 0)
    cuts = [0, 3_333_333, 7_000_001, n]
    parts = [ctx.mls_fetch(ctx.mls_process_shard(mp, a, b)) for a, b in zip(cuts[:-1], cuts[1:])]
    for k in ("index", "xyz", "normal", "curvature"):
        assert np.array_equal(np.concatenate([p[k] for p in parts]), full[k]), k
    ctx.close()
    # oracle on the slab 0 < x < 1.2 (about 1 M points); compare where the r = 0.03 ball is inside the slab
    slab = np.nonzero((x > 0.0) & (x < 1.2))[0]
    op = oracle.default_mls_params()
    op.upsampling = 0
    op.threads = 0
    ref = oracle.mls(x[slab], y[slab], z[slab], op)
    ref_idx = slab[ref["index"]]
    inner = (x[ref_idx] > 0.04) & (x[ref_idx] < 1.16)
    pos = np.searchsorted(full["index"], ref_idx[inner])
    assert np.array_equal(full["index"][pos], ref_idx[inner])  # every interior point the oracle fits, the GPU fits
    assert inner.sum() > 500_000
    assert np.abs(full["xyz"][pos].astype(np.float64) - ref["xyz"][inner]).max() <= 3e-6
    sgn = np.sign((full["normal"][pos] * ref["normal"][inner]).sum(axis=1))
    assert np.abs(full["normal"][pos] * sgn[:, None] - ref["normal"][inner]).max() <= 1e-4
    np.testing.assert_allclose(full["curvature"][pos], ref["curvature"][inner], rtol=1e-4, atol=1e-9)


def test_sor_and_cloud_smooth_10M(cloud10m, oracle):
    """StatisticalOutlierRemoval at 10 M points: the keep flags of a slab's interior are consistent with the oracle's
    exact mean kNN distances under ONE threshold (the global mean + 0.7 sigma is a property of the whole cloud, the
    distances are local); the whole SOR -> MLS -> SOR chain is deterministic and its index set nests as it must."""
    from pointcloudprocessor_amd import capi

    x, y, z = cloud10m
    n = len(x)
    ctx = capi.Context(0)
    ctx.set_camera(capi.default_camera())
    ctx.upload_cloud(x, y, z)
    keep, kept = ctx.sor(60, 0.7)
    keep2, kept2 = ctx.sor(60, 0.7)
    assert np.array_equal(keep, keep2) and kept == kept2 == int(keep.sum())
    assert 0.5 * n < kept < n
    slab = np.nonzero((x > 0.0) & (x < 0.5))[0]  # ~400 k points
    _, _, dist, _ = oracle.sor(x[slab], y[slab], z[slab], 60, 0.7, threads=0, details=True)
    inner = (x[slab] > 0.1) & (x[slab] < 0.4)  # 60 nearest neighbours lie well within 0.1 m at this density
    d_in, k_in = dist[inner], keep[slab][inner].astype(bool)
    assert k_in.sum() > 50_000 and (~k_in).sum() > 5_000
    # one threshold separates them: every kept distance <= every dropped distance (fp32 ties aside)
    assert d_in[k_in].max() <= d_in[~k_in].min() * (1 + 1e-6)
    mp = capi.default_mls_params()
    mp.upsampling = 0
    a = ctx.mls_fetch(ctx.cloud_smooth(mp))
    b = ctx.mls_fetch(ctx.cloud_smooth(mp))
    for k in ("index", "xyz", "normal", "curvature"):
        assert np.array_equal(a[k], b[k]), k
    assert 0.3 * n < len(a["index"]) < kept
    assert np.all(keep[a["index"]] == 1)  # survivors of the chain survived the first SOR
    assert np.all(np.diff(a["index"]) > 0)
    ctx.close()


def _region_check(oracle, x, y, z, keep1, region, thr):
    """The survivors of the streamed chain whose source point lies in a region of the map, against the oracle's stages on that
    region: the first filter's survivors within `outer` of the centre -> the oracle's MLS + VOXEL_GRID_DILATION on the MAP's
    voxel lattice -> the oracle's mean 60-NN distances of those rows -> kept iff distance <= the chain's threshold (the one
    number that needs the whole map).  Compared per source point within `inner` of the centre (their fits, voxels and
    neighbourhoods are complete 0.12 m inside the region): the number of surviving rows and their positions."""
    c, inner, outer, got_idx, got_xyz = region
    k1 = keep1.astype(bool)
    box = k1 & (np.abs(x - c[0]) < outer) & (np.abs(y - c[1]) < outer) & (np.abs(z - c[2]) < outer)
    sub = np.nonzero(box)[0]
    assert 2_000 < len(sub) < 60_000, len(sub)
    op = oracle.default_mls_params()
    assert op.upsampling == 3 and op.vgd_iterations == 4
    op.threads = 16
    origin = np.array([x[k1].min(), y[k1].min(), z[k1].min()], np.float32)
    extent = float(max(np.float32(x[k1].max()) - origin[0], np.float32(y[k1].max()) - origin[1], np.float32(z[k1].max()) - origin[2]))
    r = oracle.mls_voxel_dilation_part(x[sub], y[sub], z[sub], op, origin, extent)
    rx = r["xyz"]
    _, _, dist, _ = oracle.sor(rx[:, 0].copy(), rx[:, 1].copy(), rx[:, 2].copy(), 60, 0.7, threads=0, details=True)
    src = sub[r["index"]]
    core = (np.abs(x[src] - c[0]) < inner) & (np.abs(y[src] - c[1]) < inner) & (np.abs(z[src] - c[2]) < inner)
    want_keep = core & ~(dist.astype(np.float64) > thr)
    n_src = len(x)
    want = np.bincount(src[want_keep], minlength=n_src)
    gcore = (np.abs(x[got_idx] - c[0]) < inner) & (np.abs(y[got_idx] - c[1]) < inner) & (np.abs(z[got_idx] - c[2]) < inner)
    got = np.bincount(got_idx[gcore], minlength=n_src)
    sources = np.nonzero((want > 0) | (got > 0))[0]
    assert len(sources) > 300 and int(want.sum()) > 100_000
    # (a row whose distance sits within the fit's tolerance of the threshold may fall on either side: a handful in 10^5)
    differing = int(np.abs(want - got).sum())
    assert differing <= 2e-4 * want.sum(), (differing, int(want.sum()), int(got.sum()))
    # positions: every surviving row of the GPU lies within 3 um (1e-4 r) of an oracle row of the same source
    same = sources[want[sources] == got[sources]][:200]
    for s_ in same:
        a = np.sort(got_xyz[gcore & (got_idx == s_)].astype(np.float64), axis=0)
        b = np.sort(rx[want_keep & (src == s_)].astype(np.float64), axis=0)
        assert np.abs(a - b).max() <= 3.0e-6 + 1e-9, s_
    return int(want.sum()), differing, len(sources)


def test_streamed_chain_whole_map_is_chunking_invariant(cloud10m, monkeypatch, oracle):
    """CloudSmooth::process with the reference's own MLS configuration on the WHOLE 10 M-point map (2.8e9 upsampled rows:
    pcp_cloud_smooth_stream_*): the chain cut into 11 and into 22 chunks, and with the ball of the trailing filter's selection
    held fixed instead of adapted, gives the same rows before the last filter, the same threshold to the last bit, the same
    number of survivors and the same checksums over their source indices and positions; every survivor's source point
    survived the first outlier removal; each chunk's halo was PROVEN (margin above the largest displacement), nothing had
    to be redone.  And against the ORACLE on a region of the map (0.4 m across: its stages restated on the map's own voxel
    lattice): the surviving rows of every source point in the region's core and their positions."""
    import ctypes as C

    from pointcloudprocessor_amd import capi

    x, y, z = cloud10m
    ctx = capi.Context(0)
    ctx.set_camera(capi.default_camera())
    ctx.upload_cloud(x, y, z)
    keep1, _ = ctx.sor(60, 0.7)
    vp = capi.default_mls_params()
    runs = []
    # a region of the map for the comparison with the oracle's stages (below): 0.2 m around a surviving point in mid-map
    i0 = int(np.nonzero(keep1)[0][len(x) // 3])
    centre, inner, outer = (float(x[i0]), float(y[i0]), float(z[i0])), 0.08, 0.20
    in_region = (np.abs(x - centre[0]) < inner) & (np.abs(y - centre[1]) < inner) & (np.abs(z - centre[2]) < inner)
    reg_idx, reg_xyz = [], []
    for cap, ball in ((1 << 28, None), (1 << 27, None), (1 << 28, "1.2")):
        if ball is not None:
            monkeypatch.setenv("PCP_SOR_BALL", ball)  # (the selection's ball fixed instead of following the flagged share)
        rows, kept, chunks = ctx.cloud_smooth_stream_begin(vp, cap)
        st = ctx.cloud_smooth_stream_stats()
        got = 0
        idx_sum = 0
        x_sum = 0  # (of the positions' bit patterns: exact whatever the chunking)
        nested = True
        while True:
            m = ctx.cloud_smooth_stream_next()
            if m == 0:
                break
            idx = np.empty(m, np.int32)
            xyz = np.empty((m, 3), np.float32)
            ctx._check(ctx.lib.pcp_mls_fetch(ctx.h, C.c_int64(m), xyz.ctypes.data_as(C.c_void_p), None, None, idx.ctypes.data_as(C.c_void_p)))
            got += m
            idx_sum += int(idx.sum(dtype=np.int64))
            x_sum += int(xyz.view(np.uint32).sum(dtype=np.uint64))
            nested = nested and bool(np.all(keep1[idx[:: 997]] == 1))
            if len(runs) == 0:
                sel = in_region[idx]
                if sel.any():
                    reg_idx.append(idx[sel].copy())
                    reg_xyz.append(xyz[sel].copy())
            del idx, xyz
        runs.append(dict(rows=rows, kept=kept, got=got, chunks=chunks, idx_sum=idx_sum, x_sum=x_sum, thr=st["threshold"], st=st, nested=nested))
    a, b, c = runs
    assert a["chunks"] == c["chunks"] < b["chunks"] and a["rows"] == b["rows"] == c["rows"] > 2_000_000_000
    assert a["kept"] == a["got"] == b["kept"] == b["got"] == c["kept"] == c["got"] and 0.5 * a["rows"] < a["kept"] < a["rows"]
    for o in (b, c):
        assert a["thr"] == o["thr"] and a["idx_sum"] == o["idx_sum"] and a["x_sum"] == o["x_sum"]
    for r in runs:
        assert r["nested"] and r["st"]["chunks_redone"] == 0 and r["st"]["min_margin_m"] > r["st"]["max_displacement_m"] > 0
    ctx.close()
    # ... and against the oracle on a region of the map (MLS + dilation on the map's own voxel lattice, distances, the chain's
    # threshold): the surviving rows per source point and their positions
    rows_cmp, differing, n_sources = _region_check(oracle, x, y, z, keep1, (centre, inner, outer, np.concatenate(reg_idx), np.concatenate(reg_xyz)),
                                                   a["thr"])
    print(f"region check: {rows_cmp} surviving rows of {n_sources} source points compared, {differing} differ")
