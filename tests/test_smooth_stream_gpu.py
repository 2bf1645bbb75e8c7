"""pcp_cloud_smooth_stream_*: CloudSmooth::process (PCP/src/cloudSmooth.cpp:109-164) with its trailing
StatisticalOutlierRemoval over the chunked VOXEL_GRID_DILATION -- against the one-shot pcp_cloud_smooth (bit for bit) and
against the chain composed from the oracle's stages."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

R = 0.03


def _strip(seed=5, n=6000, half_x=0.5, half_y=0.05):
    """a gently curved strip, long along x (the axis the chain cuts by), with 1 mm of noise and a few strays"""
    rng = np.random.default_rng(seed)
    a = rng.uniform(-1.0, 1.0, (n, 2)) * [half_x, half_y]
    zz = 0.08 * np.sin(3.0 * a[:, 0]) + 0.5 * a[:, 1] ** 2 + rng.normal(0, 1e-3, n)
    pts = np.stack([a[:, 0] + 2.0, a[:, 1] - 1.0, zz + 1.5], 1)
    strays = rng.uniform(-1.0, 1.0, (40, 3)) * [half_x, half_y, 0.05] + [2.0, -1.0, 1.6]
    pts = np.concatenate([pts, strays]).astype(np.float32)
    pts = pts[rng.permutation(len(pts))]
    return pts[:, 0].copy(), pts[:, 1].copy(), pts[:, 2].copy()


def _stream(ctx, mp, capacity):
    total, kept, chunks = ctx.cloud_smooth_stream_begin(mp, capacity)
    parts = []
    while True:
        m = ctx.cloud_smooth_stream_next()
        if m == 0:
            break
        parts.append(ctx.mls_fetch(m))
    if parts:
        got = {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}
    else:
        got = {k: np.zeros((0,)) for k in ("index",)}
    return total, kept, chunks, got, ctx.cloud_smooth_stream_stats()


@pytest.mark.parametrize("halo", [None, "3"])
def test_streamed_chain_equals_one_shot(gpu_ctx_factory, monkeypatch, halo):
    """The reference's own MLS configuration (1 mm voxels, 4 dilations, PointCloudProcessor.cpp:78-81) on a strip 1 m long:
    ~1000 planes of the first voxel axis, cut into chunks of at most 100 k voxels.  The survivors of the chunks, concatenated,
    are the rows of the one-shot chain bit for bit; with a halo of 3 planes (PCP_CSS_HALO) the proof of the halo fails and the
    chunks are redone with wider ones -- same rows."""
    from pointcloudprocessor_amd import capi

    x, y, z = _strip()
    ctx = gpu_ctx_factory()
    ctx.upload_cloud(x, y, z)
    mp = capi.default_mls_params()
    assert mp.upsampling == 3 and abs(mp.vgd_voxel_size - 0.001) < 1e-9 and mp.vgd_iterations == 4
    one = ctx.mls_fetch(ctx.cloud_smooth(mp))
    assert len(one["index"]) > 300_000
    if halo:
        monkeypatch.setenv("PCP_CSS_HALO", halo)
    total, kept, chunks, got, st = _stream(ctx, mp, 100_000)
    assert chunks >= 6 and total > kept == len(one["index"])
    assert st["device_bytes_held"] >= 4 * total  # (the distances stay for the next stream ...)
    assert 0.0 < st["sampled_displacement_m"] <= st["max_displacement_m"]
    for k in ("index", "xyz", "normal", "curvature"):
        assert np.array_equal(got[k], one[k]), k
    assert st["min_margin_m"] > st["max_displacement_m"] > 0.0
    if halo:
        assert st["chunks_redone"] > 0 and st["halo_planes"] > 3
    else:
        assert st["chunks_redone"] == 0
    assert ctx.cloud_smooth_stream_next() == 0  # the stream is over
    # a fit of another call replaces what the stream rests on: the stream ends
    ctx.cloud_smooth_stream_begin(mp, 100_000)
    assert ctx.cloud_smooth_stream_next() > 0
    ctx.cloud_smooth_stream_end()  # (... until the stream is ended: nothing is held, no chunk follows)
    assert ctx.cloud_smooth_stream_stats()["device_bytes_held"] == 0
    with pytest.raises(capi.PcpError):
        ctx.cloud_smooth_stream_next()
    total2, kept2, chunks2, got2, st2 = _stream(ctx, mp, 100_000)  # and a new stream starts from nothing: same rows
    assert (total2, kept2, chunks2) == (total, kept, chunks) and np.array_equal(got2["xyz"], one["xyz"])
    ctx.cloud_smooth_stream_begin(mp, 100_000)
    mp0 = capi.default_mls_params()
    mp0.upsampling = 0
    ctx.mls_process(mp0)
    with pytest.raises(capi.PcpError):
        ctx.cloud_smooth_stream_next()


def test_streamed_chain_matches_oracle(gpu_ctx_factory, oracle):
    """... and the chain composed from the oracle's stages (SOR, MLS + VOXEL_GRID_DILATION, SOR), as
    tests/test_mls_gpu.py::test_cloud_smooth_chain_matches_oracle checks the one-shot form."""
    from pointcloudprocessor_amd import capi

    x, y, z = _strip(seed=8, n=5000, half_x=0.3, half_y=0.1)
    ctx = gpu_ctx_factory()
    ctx.upload_cloud(x, y, z)
    mp = capi.default_mls_params()
    mp.vgd_voxel_size = 0.003
    mp.vgd_iterations = 1
    total, kept, chunks, got, st = _stream(ctx, mp, 4096)
    assert chunks >= 4
    keep1, _ = oracle.sor(x, y, z, 60, 0.7, threads=8)
    idx1 = np.nonzero(keep1)[0]
    op = oracle.default_mls_params()
    op.upsampling = 3
    op.vgd_voxel_size = 0.003
    op.vgd_iterations = 1
    op.threads = 8
    r = oracle.mls_voxel_dilation(x[idx1], y[idx1], z[idx1], op)
    assert total == len(r["index"])
    keep2, _ = oracle.sor(r["xyz"][:, 0].copy(), r["xyz"][:, 1].copy(), r["xyz"][:, 2].copy(), 60, 0.7, threads=8)
    k2 = np.nonzero(keep2)[0]
    assert 0 < len(k2) < len(r["index"])
    assert kept == len(k2)
    assert np.array_equal(got["index"], idx1[r["index"][k2]])
    assert np.abs(got["xyz"].astype(np.float64) - r["xyz"][k2]).max() <= 1e-4 * R
    np.testing.assert_allclose(got["curvature"], r["curvature"][k2], rtol=1e-4, atol=1e-9)


def test_streamed_chain_argument_checks(gpu_ctx_factory):
    from pointcloudprocessor_amd import capi

    x, y, z = _strip(n=2000)
    ctx = gpu_ctx_factory()
    with pytest.raises(capi.PcpError):
        ctx.cloud_smooth_stream_next()  # no stream
    ctx.upload_cloud(x, y, z)
    mp = capi.default_mls_params()
    with pytest.raises(capi.PcpError):
        ctx.cloud_smooth_stream_begin(mp, 100)  # capacity below the minimum
    mp0 = capi.default_mls_params()
    mp0.upsampling = 0
    with pytest.raises(capi.PcpError):
        ctx.cloud_smooth_stream_begin(mp0, 100_000)  # nothing to stream without upsampling
    # a plane of the voxel grid that holds more voxels than a chunk may: a wall across the first axis
    rng = np.random.default_rng(1)
    w = rng.uniform(-0.15, 0.15, (4000, 2))
    ctx.upload_cloud(np.full(4000, 1.0, np.float32) + rng.normal(0, 1e-3, 4000).astype(np.float32), w[:, 0].astype(np.float32),
                     w[:, 1].astype(np.float32))
    with pytest.raises(capi.PcpError) as e:
        ctx.cloud_smooth_stream_begin(mp, 4096)
    assert e.value.code == capi.PCP_ERR_RANGE


def test_python_host_streamed_cloud_smooth():
    """pipeline.CloudSmooth (the Python mirror of the reference's CloudSmooth): process_streamed yields the rows of process()
    chunk by chunk, and process() itself takes that road when one result cannot hold the upsampled cloud (here: made to, by a
    one-shot call that reports PCP_ERR_NOMEM)."""
    from pointcloudprocessor_amd import capi, pipeline

    x, y, z = _strip()
    eng = pipeline.HipEngine(0)
    eng.configure(capi.default_camera())
    eng.upload_cloud(x, y, z)
    cs = pipeline.CloudSmooth(eng)
    one = cs.process()
    parts = list(cs.process_streamed(100_000))
    assert cs.streamed["chunks"] >= 6 and cs.streamed["kept"] == len(one["index"]) and len(parts) >= 6
    for k in ("index", "xyz", "normal", "curvature"):
        assert np.array_equal(np.concatenate([p[k] for p in parts]), one[k]), k
    assert eng.ctx.cloud_smooth_stream_stats()["device_bytes_held"] == 0  # (ended: nothing is held)

    class NoRoom:  # the context with a one-shot chain that has no room
        def __init__(self, ctx):
            self._ctx = ctx

        def __getattr__(self, name):
            return getattr(self._ctx, name)

        def cloud_smooth(self, params):
            raise capi.PcpError(capi.PCP_ERR_NOMEM, "pcp_cloud_smooth: (test) the upsampled points do not fit")

    real = eng.ctx
    eng.ctx = NoRoom(real)
    try:
        again = cs.process()
    finally:
        eng.ctx = real
    for k in ("index", "xyz", "normal", "curvature"):
        assert np.array_equal(again[k], one[k]), k
    eng.close()
