"""GPU parity of StatisticalOutlierRemoval (pcp_sor) against the CPU restatement:
keep flags equal except for points whose mean kNN distance sits within 1e-6
relative of the threshold (fp64 summation order)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _check(ctx, oracle, x, y, z, mean_k, mul):
    ctx.upload_cloud(x, y, z)
    keep_g, kept_g = ctx.sor(mean_k, mul)
    keep_r, kept_r, dist, thr = oracle.sor(x, y, z, mean_k, mul, threads=8, details=True)
    diff = np.nonzero(keep_g != keep_r)[0]
    assert np.all(np.abs(dist[diff] - thr) <= 1e-6 * thr), (len(diff), dist[diff][:5], thr)
    assert abs(kept_g - kept_r) <= len(diff)
    assert kept_g == int(keep_g.sum())
    # the thresholded quantity itself, bit for bit: the k + 1 nearest are selected exactly, each sqrt is correctly
    # rounded and the fp64 sum of <= 255 fp32 terms within a few binades is exact in any order
    assert np.array_equal(ctx.sor_distances().view(np.uint32), dist.astype(np.float32).view(np.uint32))
    return kept_r


def test_sor_matches_oracle_on_surface_with_outliers(gpu_ctx_factory, oracle):
    rng = np.random.default_rng(4)
    n = 40000
    a = rng.uniform(-0.5, 0.5, (n, 2))
    pts = np.stack([a[:, 0], a[:, 1], 0.1 * np.sin(6 * a[:, 0]) + rng.normal(0, 1e-3, n)], 1)
    out = rng.uniform(-0.5, 0.5, (300, 3))
    pts = np.concatenate([pts, out]).astype(np.float32)
    pts = pts[rng.permutation(len(pts))]
    ctx = gpu_ctx_factory()
    kept = _check(ctx, oracle, pts[:, 0].copy(), pts[:, 1].copy(), pts[:, 2].copy(), 60, 0.7)
    assert 0.5 * len(pts) < kept < len(pts)
    _check(ctx, oracle, pts[:, 0].copy(), pts[:, 1].copy(), pts[:, 2].copy(), 8, 1.5)


def test_sor_on_scene_crop(gpu_ctx_factory, oracle):
    from pointcloudprocessor_amd import synth

    x, y, z, _ = synth.make_cloud(2_000_000)
    sel = (x > 1.0) & (x < 3.0) & (y > -5.1) & (y < -3.5) & (z < 1.5)
    ctx = gpu_ctx_factory()
    kept = _check(ctx, oracle, x[sel], y[sel], z[sel], 60, 0.7)
    assert kept > 1000


def test_sor_edge_cases(gpu_ctx_factory):
    from pointcloudprocessor_amd import capi

    ctx = gpu_ctx_factory()
    e = np.zeros(0, np.float32)
    ctx.upload_cloud(e, e, e)
    keep, kept = ctx.sor()
    assert kept == 0 and len(keep) == 0
    ctx.upload_cloud(np.float32([0, 1, 2]), np.float32([0, 0, 0]), np.float32([0, 0, 0]))
    with pytest.raises(capi.PcpError):
        ctx.sor(mean_k=0)
    # a cloud with NaN / infinite coordinates: PCL's filters skip such points one by one, this library refuses the cloud
    # (and must not spin on an infinite bounding box); the colour path takes the same upload
    rng = np.random.default_rng(2)
    pts = rng.uniform(-1, 1, (5000, 3)).astype(np.float32)
    for bad in (np.nan, np.inf, -np.inf):
        q = pts.copy()
        q[17, 1] = bad
        ctx.upload_cloud(q[:, 0].copy(), q[:, 1].copy(), q[:, 2].copy())
        for call in (lambda: ctx.sor(), lambda: ctx.mls_process(capi.default_mls_params()),
                     lambda: ctx.cloud_smooth(capi.default_mls_params()), lambda: ctx.close_pairs(1e-5)):
            with pytest.raises(capi.PcpError, match="NaN or infinite"):
                call()
    ctx.upload_cloud(pts[:, 0].copy(), pts[:, 1].copy(), pts[:, 2].copy())
    assert ctx.sor()[1] > 0


@pytest.mark.parametrize("heap_only", ["0", "1"])
def test_sor_selection_and_heap_kernels_agree_with_oracle(gpu_ctx_factory, oracle, monkeypatch, heap_only):
    """The selection kernel (+ heap kernel for the lanes it flags) and the heap kernel alone are both the exact
    k + 1 nearest: each is checked against the oracle on a cloud with dense, sparse and border regions."""
    monkeypatch.setenv("PCP_SOR_HEAP_ONLY", heap_only)
    rng = np.random.default_rng(11)
    dense = np.stack([rng.uniform(0, 0.5, 60000), rng.uniform(0, 0.5, 60000), rng.normal(0, 5e-4, 60000)], 1)
    sparse = np.stack([rng.uniform(0.5, 2.0, 6000), rng.uniform(0, 1.5, 6000), rng.normal(0, 5e-4, 6000)], 1)
    stray = rng.uniform(-1, 3, (200, 3))
    pts = np.concatenate([dense, sparse, stray]).astype(np.float32)
    pts = pts[rng.permutation(len(pts))]
    ctx = gpu_ctx_factory()
    from pointcloudprocessor_amd import capi

    ctx.set_camera(capi.default_camera())
    kept = _check(ctx, oracle, pts[:, 0].copy(), pts[:, 1].copy(), pts[:, 2].copy(), 60, 0.7)
    frac = ctx.sor_redo_fraction()
    assert (frac == 0.0) if heap_only == "1" else (0.0 < frac < 0.6)
    assert 0.5 * len(pts) < kept < len(pts)
    ctx.close()


def test_sor_tiny_and_zero_distances(gpu_ctx_factory, oracle):
    """Duplicates (d = 0) and a cluster around the origin whose squared distances are subnormal in fp32: the selection
    kernel's short sqrt does not cover 0 < d < 2^-96 and must hand those lanes to the heap kernel (sqrtf)."""
    rng = np.random.default_rng(21)
    n = 30000
    a = rng.uniform(-0.5, 0.5, (n, 2))
    pts = np.stack([a[:, 0], a[:, 1], rng.normal(0, 1e-3, n)], 1)
    pts[:400] = pts[400:800]  # exact duplicates
    tiny = rng.normal(0, 1e-21, (300, 3))  # squared distances ~1e-42 .. 1e-40
    pts = np.concatenate([pts, tiny]).astype(np.float32)
    pts = pts[rng.permutation(len(pts))]
    ctx = gpu_ctx_factory()
    _check(ctx, oracle, pts[:, 0].copy(), pts[:, 1].copy(), pts[:, 2].copy(), 60, 0.7)
    assert ctx.sor_redo_fraction() > 0.0


def test_sor_flagged_point_paths(gpu_ctx_factory, oracle):
    """The three ways out of the selection kernel: a stray point whose block of cells grows to the whole grid, a pile
    of identical points that overflows the wavefront kernel's cache (heap kernel), and thin spots that need two or three
    cells -- every distance bit-equal to the oracle's."""
    rng = np.random.default_rng(33)
    n = 30000
    a = rng.uniform(-0.5, 0.5, (n, 2))
    sheet = np.stack([a[:, 0], a[:, 1], rng.normal(0, 1e-3, n)], 1)
    thin = np.stack([rng.uniform(0.5, 1.5, 1500), rng.uniform(-0.5, 0.5, 1500), rng.normal(0, 1e-3, 1500)], 1)
    pile = np.repeat(sheet[:1], 3000, axis=0)  # 3000 copies of one point: a boundary bin no refinement can thin out
    stray = np.array([[40.0, -35.0, 12.0], [-60.0, 3.0, -7.0]])
    pts = np.concatenate([sheet, thin, pile, stray]).astype(np.float32)
    pts = pts[rng.permutation(len(pts))]
    ctx = gpu_ctx_factory()
    _check(ctx, oracle, pts[:, 0].copy(), pts[:, 1].copy(), pts[:, 2].copy(), 60, 0.7)
    assert ctx.sor_redo_fraction() > 0.05


def test_sor_fewer_points_than_neighbours(gpu_ctx_factory, oracle):
    """A cloud of 25 points with mean_k = 60: nearestKSearch returns what there is; every point is flagged and its
    block is the whole grid."""
    rng = np.random.default_rng(5)
    pts = rng.uniform(-1, 1, (25, 3)).astype(np.float32)
    ctx = gpu_ctx_factory()
    _check(ctx, oracle, pts[:, 0].copy(), pts[:, 1].copy(), pts[:, 2].copy(), 60, 0.7)
    _check(ctx, oracle, pts[:, 0].copy(), pts[:, 1].copy(), pts[:, 2].copy(), 24, 1.0)


@pytest.mark.parametrize("mean_k", [127, 200, 254])
def test_sor_large_neighbour_counts(gpu_ctx_factory, oracle, mean_k):
    """mean_k up to the ABI's limit: the selection kernel's lists do not depend on k (<= 250 neighbours), beyond that the
    heap kernel runs alone with 130 KB of LDS per workgroup."""
    rng = np.random.default_rng(1)
    n = 20000
    a = rng.uniform(-0.5, 0.5, (n, 2))
    pts = np.stack([a[:, 0], a[:, 1], rng.normal(0, 1e-3, n)], 1).astype(np.float32)
    ctx = gpu_ctx_factory()
    _check(ctx, oracle, pts[:, 0].copy(), pts[:, 1].copy(), pts[:, 2].copy(), mean_k, 1.0)
    assert (ctx.sor_redo_fraction() == 0.0) == (mean_k + 1 > 250)


@pytest.mark.parametrize("form", ["0", "1"])
def test_sor_dense_and_sparse_grid_forms(gpu_ctx_factory, oracle, monkeypatch, form):
    """The table of cell starts as one entry per cell (dense) and as entries of the occupied cells found through a bitmap
    with running popcounts (sparse, what a grid of more than 2^29 cells gets): same distances, bit for bit."""
    monkeypatch.setenv("PCP_GRID_SPARSE", form)
    rng = np.random.default_rng(12)
    n = 50000
    a = rng.uniform(-0.5, 0.5, (n, 2))
    pts = np.stack([a[:, 0], a[:, 1], 0.05 * np.cos(5 * a[:, 1]) + rng.normal(0, 1e-3, n)], 1)
    pts = np.concatenate([pts, rng.uniform(-1, 1, (200, 3))]).astype(np.float32)
    ctx = gpu_ctx_factory()
    _check(ctx, oracle, pts[:, 0].copy(), pts[:, 1].copy(), pts[:, 2].copy(), 60, 0.7)


@pytest.mark.parametrize("three", ["0", "1"])
def test_sor_selection_reads_the_planes_through_one_descriptor_or_three(gpu_ctx_factory, oracle, monkeypatch, three):
    """k_sor_select reads the three coordinate planes through ONE buffer descriptor while they end below 2^32 bytes, and
    through one descriptor per plane above (the dilated clouds of VOXEL_GRID_DILATION: up to 2^30 points);
    PCP_SOR_THREE_DESCRIPTORS=1 takes the second form on any cloud.  A size that is no multiple of four, so that the last
    loads run past the planes' ends."""
    monkeypatch.setenv("PCP_SOR_THREE_DESCRIPTORS", three)
    rng = np.random.default_rng(21)
    n = 50003
    a = rng.uniform(-0.5, 0.5, (n, 2))
    pts = np.stack([a[:, 0], a[:, 1], 0.05 * np.sin(7 * a[:, 0] * a[:, 1]) + rng.normal(0, 1e-3, n)], 1)
    pts = np.concatenate([pts, rng.uniform(-0.6, 0.6, (150, 3))]).astype(np.float32)
    ctx = gpu_ctx_factory()
    _check(ctx, oracle, pts[:, 0].copy(), pts[:, 1].copy(), pts[:, 2].copy(), 60, 0.7)
    assert ctx.sor_redo_fraction() < 0.5  # the selection kernel did the work


@pytest.mark.parametrize("clustered", ["0", "1"])
@pytest.mark.parametrize("three", ["0", "1"])
def test_sor_selection_shapes_give_the_same_distances(gpu_ctx_factory, oracle, monkeypatch, clustered, three):
    """k_sor_select has two shapes -- 32 bins / 16 members for scanned clouds, 64 / 32 for the upsampled clouds of
    VOXEL_GRID_DILATION, whose distances come in clusters (PCP_SOR_CLUSTERED forces one on any cloud) --: the same mean
    distances (bit for bit against the oracle) on a scanned sheet AND on a cloud of stacked near-duplicates, in both
    descriptor forms."""
    monkeypatch.setenv("PCP_SOR_CLUSTERED", clustered)
    monkeypatch.setenv("PCP_SOR_THREE_DESCRIPTORS", three)
    rng = np.random.default_rng(23)
    n = 6000
    a = rng.uniform(-0.25, 0.25, (n, 2))
    base = np.stack([a[:, 0], a[:, 1], 0.03 * np.sin(9 * a[:, 0]) + rng.normal(0, 5e-4, n)], 1)
    # every point nine times, the copies within 20 um of each other (a voxel column projected onto the surface)
    stack = (base[:, None, :] + rng.normal(0, 2e-5, (n, 9, 3))).reshape(-1, 3)
    pts = np.concatenate([stack, rng.uniform(-0.3, 0.3, (100, 3))]).astype(np.float32)
    pts = pts[rng.permutation(len(pts))]
    ctx = gpu_ctx_factory()
    _check(ctx, oracle, pts[:, 0].copy(), pts[:, 1].copy(), pts[:, 2].copy(), 60, 0.7)
    assert ctx.sor_redo_fraction() < 0.5  # the selection kernel did the work


def test_sor_stray_points_far_from_the_cloud(gpu_ctx_factory, oracle):
    """Points hundreds of metres from a 1 m sheet: the bounding box needs 10^12 cells at the wanted edge (sparse form,
    coarser cells), and a stray point's block of cells would have to grow over millions of empty rows -- it reads every
    point instead.  An earlier build took 45 s for this cloud."""
    import time

    rng = np.random.default_rng(3)
    sheet = np.stack([rng.uniform(0, 1, 40000), rng.uniform(0, 1, 40000), rng.normal(0, 1e-3, 40000)], 1)
    pts = np.concatenate([sheet, [[300.0, -200.0, 50.0]], rng.uniform(-1000, 1000, (20, 3))]).astype(np.float32)
    ctx = gpu_ctx_factory()
    t = time.time()
    _check(ctx, oracle, pts[:, 0].copy(), pts[:, 1].copy(), pts[:, 2].copy(), 60, 0.7)
    assert time.time() - t < 20.0  # oracle included


@pytest.mark.parametrize("shards", [2, 3, 5])
@pytest.mark.parametrize("shuffled", [False, True])
def test_sor_slabs_equal_the_one_gpu_filter(gpu_ctx_factory, shards, shuffled):
    """pcp_sor_partial / pcp_sor_finish (the queries dealt out, cloudSmooth.cpp:109-116) by slabs of the filter's
    own spatial order: chunk sums put together, every slab classifies its own points, the OR of the masks == pcp_sor bit for
    bit -- whether the caller's point order is spatially coherent or shuffled."""
    from pointcloudprocessor_amd import synth

    x, y, z, _ = synth.make_cloud(150_000, seed=11)
    if shuffled:
        perm = np.random.default_rng(3).permutation(len(x))
        x, y, z = x[perm].copy(), y[perm].copy(), z[perm].copy()
    n = len(x)
    ref = gpu_ctx_factory()
    ref.upload_cloud(x, y, z)
    keep_ref, kept_ref = ref.sor(60, 0.7)
    c = ref.sor_chunk_points()
    chunks = (n + c - 1) // c
    ctxs, all_sums, covered = [], np.zeros((chunks, 2)), np.zeros(chunks, bool)
    for r in range(shards):
        ctx = gpu_ctx_factory()
        ctx.upload_cloud(x, y, z)
        first, sums = ctx.sor_partial(60, r, shards)
        assert not covered[first:first + len(sums)].any()
        covered[first:first + len(sums)] = True
        all_sums[first:first + len(sums)] = sums
        ctxs.append(ctx)
    assert covered.all()
    keep = np.zeros(n, np.uint8)
    total = 0
    for r in range(shards):
        k, kept = ctxs[r].sor_finish(0.7, all_sums, r, shards)
        assert not (keep & k).any() and kept == int(k.sum())
        keep |= k
        total += kept
    assert np.array_equal(keep, keep_ref) and total == kept_ref
    with pytest.raises(Exception):
        ctxs[0].sor_finish(0.7, all_sums, 1, shards)  # not the slab of this context's last partial


def test_sor_slab_is_its_share_of_the_work(gpu_ctx_factory):
    """A slab of a SHUFFLED cloud is its share of the filter's work COUNT: whole chunks of 16 384 consecutive places of
    the stage's own cell order (whole wavefronts), a quarter of them each -- which dealing the queries out by the
    caller's index ranges could not give (every wavefront would hold points of every range).  The wall-clock form of
    this check (a quarter slab < 0.45 of the whole) is a probe, not a gate: profiles/sor_slab_share_probe.py."""
    from pointcloudprocessor_amd import synth

    x, y, z, _ = synth.make_cloud(2_000_000, seed=4)
    perm = np.random.default_rng(5).permutation(len(x))
    x, y, z = x[perm].copy(), y[perm].copy(), z[perm].copy()
    ctx = gpu_ctx_factory()
    ctx.upload_cloud(x, y, z)
    first0, sums0 = ctx.sor_partial(60, 0, 1)
    chunks = len(sums0)
    assert first0 == 0 and chunks == -(-len(x) // ctx.sor_chunk_points())
    covered = np.zeros(chunks, bool)
    for r in range(4):
        first, sums = ctx.sor_partial(60, r, 4)
        assert abs(len(sums) - chunks / 4) <= 1, (r, len(sums), chunks)  # a quarter of the chunks, i.e. of the wavefronts
        assert not covered[first:first + len(sums)].any()
        covered[first:first + len(sums)] = True
        assert np.array_equal(sums, sums0[first:first + len(sums)])  # the same chunk sums, bit for bit
    assert covered.all()


def test_sor_sharded_python_host_single_rank(gpu_ctx_factory):
    """pipeline.CloudSmooth.outlier_removal_sharded at world 1 (the collective-free path) == ctx.sor()."""
    from pointcloudprocessor_amd import pipeline, synth

    x, y, z, _ = synth.make_cloud(60_000, seed=5)
    ctx = gpu_ctx_factory()
    ctx.upload_cloud(x, y, z)
    keep_ref, _ = ctx.sor(60, 0.7)

    class _Engine:
        pass

    eng = _Engine()
    eng.ctx = ctx
    cs = pipeline.CloudSmooth(eng)
    keep = cs.outlier_removal_sharded(len(x), 0, 1)
    assert np.array_equal(keep, keep_ref)
