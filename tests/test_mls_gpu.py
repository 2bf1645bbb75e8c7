"""GPU parity of the MLS path (pcp_mls_process) against the CPU restatement.
Bars: output set and source indices exact; positions
<= 1e-4 relative to the search radius (3 um) and 1e-4 relative in world
coordinates; normals up to sign; curvature 1e-4 relative (+1e-9 abs)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

R = 0.03


def _patches(seed=7, n=6000):
    rng = np.random.default_rng(seed)
    out = []
    # noisy plane, tilted, away from the origin
    a = rng.uniform(-0.2, 0.2, (n, 2))
    nrm = np.array([0.3, -0.5, 0.81])
    nrm /= np.linalg.norm(nrm)
    e1 = np.cross(nrm, [0, 0, 1.0])
    e1 /= np.linalg.norm(e1)
    e2 = np.cross(nrm, e1)
    p = np.array([3.0, -2.0, 1.5]) + a[:, :1] * e1 + a[:, 1:] * e2 + rng.normal(0, 1e-3, (n, 1)) * nrm
    out.append(p)
    # sphere cap radius 0.8
    d = rng.normal(size=(n, 3))
    d[:, 2] = np.abs(d[:, 2]) + 2.0
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    out.append(np.array([-4.0, 1.0, 2.0]) + (0.8 + rng.normal(0, 1e-3, n))[:, None] * d)
    # saddle
    a = rng.uniform(-0.2, 0.2, (n, 2))
    z = 1.5 * a[:, 0] ** 2 - 1.2 * a[:, 1] ** 2 + 0.4 * a[:, 0] * a[:, 1] + rng.normal(0, 5e-4, n)
    out.append(np.stack([a[:, 0] + 0.5, a[:, 1] + 4.0, z + 0.3], axis=1))
    # sparse stragglers: isolated points and tiny clusters (K < 3, 3 <= K < 6)
    s = rng.uniform(-7, 7, (40, 3))
    cl = s[:10, None, :] + rng.normal(0, 4e-3, (10, 4, 3))
    out.append(s)
    out.append(cl.reshape(-1, 3))
    pts = np.concatenate(out, axis=0)
    pts = pts[rng.permutation(len(pts))].astype(np.float32)
    return pts[:, 0].copy(), pts[:, 1].copy(), pts[:, 2].copy()


def _compare(got, ref, xyz_in):
    assert np.array_equal(got["index"], ref["index"])
    d = np.abs(got["xyz"].astype(np.float64) - ref["xyz"].astype(np.float64))
    scale = np.maximum(np.abs(ref["xyz"].astype(np.float64)), 1.0)
    assert d.max() <= 1e-4 * R, d.max()
    assert (d / scale).max() <= 1e-4
    sgn = np.sign((got["normal"].astype(np.float64) * ref["normal"]).sum(axis=1))
    sgn[sgn == 0] = 1.0
    dn = np.abs(got["normal"] * sgn[:, None] - ref["normal"])
    assert dn.max() <= 1e-4, dn.max()
    np.testing.assert_allclose(got["curvature"], ref["curvature"], rtol=1e-4, atol=1e-9)


def test_mls_none_matches_oracle(gpu_ctx_factory, oracle):
    from pointcloudprocessor_amd import capi

    x, y, z = _patches()
    ctx = gpu_ctx_factory()
    ctx.upload_cloud(x, y, z)
    mp = capi.default_mls_params()
    mp.upsampling = 0
    m = ctx.mls_process(mp)
    got = ctx.mls_fetch(m)
    op = oracle.default_mls_params()
    op.upsampling = 0
    op.threads = 8
    ref = oracle.mls(x, y, z, op)
    assert 0 < len(ref["index"]) < len(x)  # stragglers with < 3 neighbours are dropped
    _compare(got, ref, (x, y, z))
    # run-to-run determinism (cell order fix-up)
    m2 = ctx.mls_process(mp)
    again = ctx.mls_fetch(m2)
    for k in got:
        assert np.array_equal(got[k], again[k]), k


def test_mls_on_scene_sample(gpu_ctx_factory, oracle):
    """A dense crop of the bench scene (walls + sphere), ~65 neighbours per point."""
    from pointcloudprocessor_amd import capi, synth

    x, y, z, _ = synth.make_cloud(3_000_000)
    sel = (x > 2.0) & (x < 2.6) & (y > -5.1) & (y < -4.4) & (z < 0.7)
    x, y, z = x[sel], y[sel], z[sel]
    assert 3000 < len(x) < 60000
    ctx = gpu_ctx_factory()
    ctx.upload_cloud(x, y, z)
    mp = capi.default_mls_params()
    mp.upsampling = 0
    got = ctx.mls_fetch(ctx.mls_process(mp))
    op = oracle.default_mls_params()
    op.upsampling = 0
    op.threads = 8
    ref = oracle.mls(x, y, z, op)
    _compare(got, ref, (x, y, z))


def test_mls_plane_known_answer(gpu_ctx_factory):
    """Exact plane z = 0 on a lattice: zero displacement, normal = +-z, curvature 0."""
    from pointcloudprocessor_amd import capi

    g = np.arange(-0.1, 0.1, 0.004, dtype=np.float32)
    xx, yy = np.meshgrid(g, g)
    x, y = xx.ravel().copy(), yy.ravel().copy()
    z = np.zeros_like(x)
    ctx = gpu_ctx_factory()
    ctx.upload_cloud(x, y, z)
    mp = capi.default_mls_params()
    mp.upsampling = 0
    got = ctx.mls_fetch(ctx.mls_process(mp))
    assert len(got["index"]) == len(x)
    assert np.abs(got["xyz"] - np.stack([x, y, z], 1)).max() < 1e-7
    assert np.abs(np.abs(got["normal"][:, 2]) - 1.0).max() < 1e-6
    assert got["curvature"].max() < 1e-9


def test_mls_edge_cases(gpu_ctx_factory, oracle):
    from pointcloudprocessor_amd import capi

    ctx = gpu_ctx_factory()
    mp = capi.default_mls_params()
    mp.upsampling = 0
    e = np.zeros(0, np.float32)
    ctx.upload_cloud(e, e, e)
    assert ctx.mls_process(mp) == 0
    # two points only: both dropped
    ctx.upload_cloud(np.float32([0, 0.001]), np.float32([0, 0]), np.float32([0, 0]))
    assert ctx.mls_process(mp) == 0
    # order 1: plane projection only
    x, y, z = _patches(seed=3, n=1500)
    ctx.upload_cloud(x, y, z)
    mp.polynomial_order = 1
    got = ctx.mls_fetch(ctx.mls_process(mp))
    op = oracle.default_mls_params()
    op.upsampling = 0
    op.polynomial_order = 1
    ref = oracle.mls(x, y, z, op)
    _compare(got, ref, (x, y, z))
    mp.polynomial_order = 7
    with pytest.raises(capi.PcpError):
        ctx.mls_process(mp)


def test_mls_voxel_grid_dilation_matches_oracle(gpu_ctx_factory, oracle):
    """VOXEL_GRID_DILATION with the reference's parameters (1 mm voxels, 4 iterations):
    same voxel set in the same (ascending key) order, same source indices, projected
    positions within 1e-4 relative."""
    from pointcloudprocessor_amd import capi

    rng = np.random.default_rng(21)
    n = 2500
    a = rng.uniform(-0.08, 0.08, (n, 2))
    zz = 0.8 * a[:, 0] ** 2 - 0.5 * a[:, 0] * a[:, 1] + rng.normal(0, 5e-4, n)
    pts = np.stack([a[:, 0] + 2.0, a[:, 1] - 1.0, zz + 0.5], 1)
    pts = np.concatenate([pts, rng.uniform(-0.1, 0.1, (6, 3)) + [2.0, -1.0, 0.6]]).astype(np.float32)  # a few strays
    x, y, z = pts[:, 0].copy(), pts[:, 1].copy(), pts[:, 2].copy()
    ctx = gpu_ctx_factory()
    ctx.upload_cloud(x, y, z)
    for vs, it in ((0.001, 4), (0.004, 1), (0.002, 0)):
        mp = capi.default_mls_params()
        mp.upsampling = 3
        mp.vgd_voxel_size = vs
        mp.vgd_iterations = it
        m = ctx.mls_process(mp)
        got = ctx.mls_fetch(m)
        op = oracle.default_mls_params()
        op.upsampling = 3
        op.vgd_voxel_size = vs
        op.vgd_iterations = it
        op.threads = 8
        ref = oracle.mls_voxel_dilation(x, y, z, op)
        assert m == len(ref["index"]) > n // 2, (vs, it, m, len(ref["index"]))
        assert np.array_equal(got["index"], ref["index"]), (vs, it)
        d = np.abs(got["xyz"].astype(np.float64) - ref["xyz"].astype(np.float64))
        assert d.max() <= 1e-4 * R, (vs, it, d.max())
        sgn = np.sign((got["normal"].astype(np.float64) * ref["normal"]).sum(axis=1))
        sgn[sgn == 0] = 1.0
        assert np.abs(got["normal"] * sgn[:, None] - ref["normal"]).max() <= 1e-4
        np.testing.assert_allclose(got["curvature"], ref["curvature"], rtol=1e-4, atol=1e-9)


def test_mls_voxel_grid_dilation_refuses_oversized_grids(gpu_ctx_factory):
    from pointcloudprocessor_amd import capi

    ctx = gpu_ctx_factory()
    ctx.upload_cloud(np.float32([0, 3000]), np.float32([0, 3000]), np.float32([0, 3000]))
    mp = capi.default_mls_params()  # 1 mm voxels over a 3 km cube: 2.7e19 voxels
    with pytest.raises(capi.PcpError) as e:
        ctx.mls_process(mp)
    assert e.value.code == capi.PCP_ERR_NOMEM


@pytest.mark.parametrize("grid_form", ["dense", "sparse"])
@pytest.mark.parametrize("upsampling", [0, 3])
def test_cloud_smooth_chain_matches_oracle(gpu_ctx_factory, oracle, upsampling, grid_form, monkeypatch):
    """CloudSmooth::process end to end: SOR -> MLS (+VGD) -> SOR on the device vs the
    same chain composed from the oracle's stages; with the grids' table of cell starts in its dense form and in the
    sparse one (occupied cells only, PCP_GRID_SPARSE=1: what boxes of more than 2^29 cells get)."""
    from pointcloudprocessor_amd import capi

    monkeypatch.setenv("PCP_GRID_SPARSE", "1" if grid_form == "sparse" else "0")

    rng = np.random.default_rng(33)
    n = 9000
    a = rng.uniform(-0.2, 0.2, (n, 2))
    zz = 0.6 * a[:, 0] ** 2 + 0.3 * a[:, 0] * a[:, 1] + rng.normal(0, 8e-4, n)
    pts = np.stack([a[:, 0] - 3.0, a[:, 1] + 1.0, zz + 1.2], 1)
    pts = np.concatenate([pts, rng.uniform(-0.2, 0.2, (60, 3)) + [-3.0, 1.0, 1.3]]).astype(np.float32)
    pts = pts[rng.permutation(len(pts))]
    x, y, z = pts[:, 0].copy(), pts[:, 1].copy(), pts[:, 2].copy()
    ctx = gpu_ctx_factory()
    ctx.upload_cloud(x, y, z)
    mp = capi.default_mls_params()
    mp.upsampling = upsampling
    mp.vgd_voxel_size = 0.003
    mp.vgd_iterations = 1
    m = ctx.cloud_smooth(mp)
    got = ctx.mls_fetch(m)
    # oracle chain
    keep1, _ = oracle.sor(x, y, z, 60, 0.7, threads=8)
    idx1 = np.nonzero(keep1)[0]
    op = oracle.default_mls_params()
    op.upsampling = upsampling
    op.vgd_voxel_size = 0.003
    op.vgd_iterations = 1
    op.threads = 8
    r = (oracle.mls_voxel_dilation if upsampling == 3 else oracle.mls)(x[idx1], y[idx1], z[idx1], op)
    keep2, _ = oracle.sor(r["xyz"][:, 0].copy(), r["xyz"][:, 1].copy(), r["xyz"][:, 2].copy(), 60, 0.7, threads=8)
    k2 = np.nonzero(keep2)[0]
    ref_index = idx1[r["index"][k2]]
    assert 0 < len(k2) < len(r["index"]) and len(idx1) < len(x)
    assert m == len(k2)
    assert np.array_equal(got["index"], ref_index)
    assert np.abs(got["xyz"].astype(np.float64) - r["xyz"][k2]).max() <= 1e-4 * R
    np.testing.assert_allclose(got["curvature"], r["curvature"][k2], rtol=1e-4, atol=1e-9)


def test_mls_query_shards_concatenate_to_the_full_result(gpu_ctx_factory):
    """Multi-GPU form: every rank fits its own query index range on the full cloud; the
    concatenation of the shards is the unsharded result, bit for bit."""
    from pointcloudprocessor_amd import capi, pipeline

    x, y, z = _patches(seed=11, n=3000)
    ctx = gpu_ctx_factory()
    ctx.upload_cloud(x, y, z)
    mp = capi.default_mls_params()
    mp.upsampling = 0
    full = ctx.mls_fetch(ctx.mls_process(mp))
    parts = []
    for r in range(3):
        lo, hi = pipeline.shard_bounds(len(x), r, 3)
        parts.append(ctx.mls_fetch(ctx.mls_process_shard(mp, lo, hi)))
        assert np.all((parts[-1]["index"] >= lo) & (parts[-1]["index"] < hi))
    for k in ("index", "xyz", "normal", "curvature"):
        assert np.array_equal(np.concatenate([p[k] for p in parts]), full[k]), k
    with pytest.raises(capi.PcpError):
        ctx.mls_process_shard(mp, 10, 5)


@pytest.mark.parametrize("slabs", [2, 3, 7])
def test_mls_query_slabs_merge_to_the_full_result(gpu_ctx_factory, slabs):
    """Multi-GPU form by slabs of the stage's own spatial order (pcp_mls_process_slab): every point is fitted by exactly one
    slab, the slabs' rows merged by source index are the unsharded result, bit for bit -- on a shuffled cloud too."""
    from pointcloudprocessor_amd import capi

    x, y, z = _patches(seed=11, n=3000)
    perm = np.random.default_rng(2).permutation(len(x))
    x, y, z = x[perm].copy(), y[perm].copy(), z[perm].copy()
    ctx = gpu_ctx_factory()
    ctx.upload_cloud(x, y, z)
    mp = capi.default_mls_params()
    mp.upsampling = 0
    full = ctx.mls_fetch(ctx.mls_process(mp))
    parts = [ctx.mls_fetch(ctx.mls_process_slab(mp, r, slabs)) for r in range(slabs)]
    for q in parts:
        assert np.all(np.diff(q["index"]) > 0)  # each slab's rows in input order
    idx = np.concatenate([q["index"] for q in parts])
    assert len(np.unique(idx)) == len(idx) == len(full["index"])
    order = np.argsort(idx, kind="stable")
    for k in ("index", "xyz", "normal", "curvature"):
        assert np.array_equal(np.concatenate([q[k] for q in parts])[order], full[k]), k
    with pytest.raises(capi.PcpError):
        ctx.mls_process_slab(mp, 3, 3)


def test_voxel_dilation_stream_equals_one_shot(gpu_ctx_factory):
    """pcp_mls_stream_begin / _next: the dilated voxel set counted in 64 bits and emitted in ascending key order chunk by
    chunk -- the concatenation is the one-shot result, bit for bit, whatever the chunk size."""
    from pointcloudprocessor_amd import capi

    x, y, z = _patches(seed=23, n=6000)
    ctx = gpu_ctx_factory()
    ctx.upload_cloud(x, y, z)
    mp = capi.default_mls_params()
    mp.vgd_voxel_size = 0.002
    mp.vgd_iterations = 2
    full = ctx.mls_fetch(ctx.mls_process(mp))
    assert len(full["index"]) > 300_000
    for cap in (32768, 100_000, 1 << 22):
        total, chunks = ctx.mls_stream_begin(mp, cap)
        parts = []
        while True:
            m = ctx.mls_stream_next()
            if m == 0:
                break
            assert m <= cap
            parts.append(ctx.mls_fetch(m))
        assert len(parts) == chunks and (chunks > 3 or cap > 100_000)
        assert total >= len(full["index"])  # voxels whose nearest point has no valid fit are counted but not emitted
        for k in ("index", "xyz", "normal", "curvature"):
            assert np.array_equal(np.concatenate([q[k] for q in parts]), full[k]), (cap, k)
    assert ctx.mls_stream_next() == 0  # past the end: nothing, no error
    with pytest.raises(capi.PcpError):
        ctx.mls_stream_begin(mp, 1000)  # below one tile's worst case
    mp.upsampling = 0
    with pytest.raises(capi.PcpError):
        ctx.mls_stream_begin(mp, 1 << 20)


def test_voxel_dilation_bricks_equal_the_dense_bitmap(gpu_ctx_factory, monkeypatch):
    """The voxel set as bricks of 16^3 voxels (the default: memory follows the surface) against the dense bitmap over the
    bounding box (PCP_VGD_DENSE=1, rounds 2-3): one-shot result and chunked emission, bit for bit."""
    from pointcloudprocessor_amd import capi

    x, y, z = _patches(seed=29, n=5000)
    mp = capi.default_mls_params()
    mp.vgd_voxel_size = 0.002
    mp.vgd_iterations = 3
    res = {}
    for form in ("bricks", "dense"):
        monkeypatch.setenv("PCP_VGD_DENSE", "1" if form == "dense" else "0")
        ctx = gpu_ctx_factory()
        ctx.upload_cloud(x, y, z)
        full = ctx.mls_fetch(ctx.mls_process(mp))
        total, chunks = ctx.mls_stream_begin(mp, 65536)
        parts = []
        while True:
            m = ctx.mls_stream_next()
            if m == 0:
                break
            assert m <= 65536
            parts.append(ctx.mls_fetch(m))
        assert chunks == len(parts) > 3
        for k in ("index", "xyz", "normal", "curvature"):
            assert np.array_equal(np.concatenate([q[k] for q in parts]), full[k]), (form, k)
        res[form] = (full, total)
    assert res["bricks"][1] == res["dense"][1] and len(res["bricks"][0]["index"]) > 400_000
    for k in ("index", "xyz", "normal", "curvature"):
        assert np.array_equal(res["bricks"][0][k], res["dense"][0][k]), k


def test_voxel_dilation_of_a_40m_map_at_the_reference_configuration(gpu_ctx_factory):
    """The reference's MLS configuration (VOXEL_GRID_DILATION 1 mm x 4, PointCloudProcessor.cpp:78-81) on a map the size of
    a survey (40 x 40 x 5 m: the crop of :119-136 is the trajectory box +- 2 m): 4e4 x 4e4 x 5e3 voxels are 1 TB as a dense
    bitmap -- refused with PCP_ERR_NOMEM until round 3 -- and a few hundred MB as bricks.  The voxel count equals the
    number of distinct PCL keys computed with numpy; the chunks come out in ascending key order and hold every voxel."""
    import torch

    from pointcloudprocessor_amd import capi

    rng = np.random.default_rng(77)
    # 700 dense patches (100 points within 4 cm) on the floor, the ceiling and two walls of a 40 x 40 x 5 m hall
    centres = rng.uniform([0, 0, 0], [40, 40, 5], (700, 3))
    which = rng.integers(0, 4, 700)
    centres[which == 0, 2] = 0.0
    centres[which == 1, 2] = 5.0
    centres[which == 2, 0] = 0.0
    centres[which == 3, 1] = 40.0
    pts = centres[:, None, :] + rng.normal(0, 0.012, (700, 100, 3)) * np.where(
        (np.arange(3)[None, :] == np.array([2, 2, 0, 1])[which][:, None]), 0.05, 1.0)[:, None, :]
    pts = np.concatenate([pts.reshape(-1, 3), [[0, 0, 0], [40, 40, 5]]]).astype(np.float32)
    x, y, z = pts[:, 0].copy(), pts[:, 1].copy(), pts[:, 2].copy()
    mp = capi.default_mls_params()
    assert abs(mp.vgd_voxel_size - 0.001) < 1e-9 and mp.vgd_iterations == 4 and mp.upsampling == 3
    free0 = torch.cuda.mem_get_info()[0]
    ctx = gpu_ctx_factory()
    ctx.upload_cloud(x, y, z)
    cap = 1 << 22
    total, chunks = ctx.mls_stream_begin(mp, cap)
    used = free0 - torch.cuda.mem_get_info()[0]
    assert used < 8 << 30, used  # (the dense bitmap would be 1 TB)
    # the voxel set PCL builds: cell indices in fp32 (MLSVoxelGrid::getCellIndex), the 9 x 9 x 9 cube of every point
    # (negative indices not created, Appendix B16), distinct keys
    vs = np.float32(0.001)
    ix = ((x - x.min()) / vs).astype(np.int64)
    iy = ((y - y.min()) / vs).astype(np.int64)
    iz = ((z - z.min()) / vs).astype(np.int64)
    d = np.arange(-4, 5)
    keys = []
    S = np.int64(1) << 21
    for lo in range(0, len(x), 10_000):
        cx = (ix[lo:lo + 10_000, None] + d[None, :]).reshape(-1, 9, 1, 1)
        cy = (iy[lo:lo + 10_000, None] + d[None, :]).reshape(-1, 1, 9, 1)
        cz = (iz[lo:lo + 10_000, None] + d[None, :]).reshape(-1, 1, 1, 9)
        ok = (cx >= 0) & (cy >= 0) & (cz >= 0)
        keys.append(np.unique(((cx * S + cy) * S + cz)[ok & np.ones((1, 9, 9, 9), bool)]))
    n_keys = len(np.unique(np.concatenate(keys)))
    assert total == n_keys, (total, n_keys)
    assert chunks >= total // cap
    emitted, last_key = 0, -1
    while True:
        m = ctx.mls_stream_next()
        if m == 0:
            break
        assert m <= cap
        emitted += m
    assert 0.9 * total < emitted <= total  # voxels whose nearest point has no valid fit are counted but not emitted
