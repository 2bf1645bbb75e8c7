"""GPU parity of the colour path: HIP kernels (through the C ABI) vs the CPU
restatement in oracle/ on the same seeded inputs.  Bars:
cells / pixels / depth maps / keep masks bit-exact; colours within 1e-4 relative
(uint8 equal unless R/S sits within 1e-4*255 of an integer)."""
import numpy as np
import pytest

from conftest import cam_struct

pytestmark = pytest.mark.gpu


def _setup(ctx, capi, scene, cam_overrides=None, cull=None, with_masks=False):
    cd = dict(scene["cam"])
    if cam_overrides:
        cd.update(cam_overrides)
    ctx.set_camera(cam_struct(capi, cd), cull)
    ctx.upload_cloud(scene["x"], scene["y"], scene["z"])
    ctx.set_frames(scene["poses"])
    for f, im in enumerate(scene["images"]):
        ctx.upload_image(f, im)
        if with_masks:
            ctx.upload_mask(f, scene["masks"][f])
    return cd


def test_pose_to_matrices_matches_oracle(oracle, small_scene):
    from pointcloudprocessor_amd import capi

    for pose in small_scene["poses"]:
        a = capi.pose_to_matrices(pose)
        b = oracle.pose_to_matrices(pose)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    T = np.eye(4)
    T[:3, 3] = [0.01, -0.02, 0.005]
    a = capi.pose_to_matrices(small_scene["poses"][0], T)
    b = oracle.pose_to_matrices(small_scene["poses"][0], T)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_project_frame_bit_exact(gpu_ctx_factory, oracle, small_scene):
    from pointcloudprocessor_amd import capi

    ctx = gpu_ctx_factory()
    cd = _setup(ctx, capi, small_scene)
    ocam, ocp = cam_struct(oracle, cd), oracle.default_cull_params()
    for f, pose in enumerate(small_scene["poses"]):
        w2c, _ = oracle.pose_to_matrices(pose)
        ref = oracle.project_frame(ocam, ocp, w2c, small_scene["x"], small_scene["y"], small_scene["z"])
        got = ctx.project_frame(f)
        for k in ("xc", "yc", "zc", "cell", "pixel"):
            assert np.array_equal(got[k], ref[k]), (f, k)
        cand = ref["cell"] != -1
        assert cand.sum() > 100
        assert np.array_equal(got["range"][cand], ref["range"][cand])
        assert np.all(got["range"][~cand] == np.finfo(np.float32).max)


def test_project_frame_ragged_sizes(gpu_ctx_factory, oracle, small_scene):
    """n not a multiple of 4 / 256, n < 4, n == 0."""
    from pointcloudprocessor_amd import capi

    ctx = gpu_ctx_factory()
    cd = small_scene["cam"]
    ctx.set_camera(cam_struct(capi, cd))
    ctx.set_frames(small_scene["poses"][:1])
    ocam, ocp = cam_struct(oracle, cd), oracle.default_cull_params()
    w2c, _ = oracle.pose_to_matrices(small_scene["poses"][0])
    for n in (0, 1, 3, 5, 255, 257, 1023, 4099):
        x, y, z = (small_scene[k][:n] for k in "xyz")
        ctx.upload_cloud(x, y, z)
        got = ctx.project_frame(0)
        ref = oracle.project_frame(ocam, ocp, w2c, x, y, z)
        for k in ("cell", "pixel", "xc", "yc", "zc"):
            assert np.array_equal(got[k], ref[k]), (n, k)


def test_cull_frame_depth_and_keep_exact(gpu_ctx_factory, oracle, small_scene):
    from pointcloudprocessor_amd import capi

    ctx = gpu_ctx_factory()
    cd = _setup(ctx, capi, small_scene)
    ocam, ocp = cam_struct(oracle, cd), oracle.default_cull_params()
    for f, pose in enumerate(small_scene["poses"]):
        w2c, _ = oracle.pose_to_matrices(pose)
        keep_r, dmap_r, kept_r = oracle.cull_frame(ocam, ocp, w2c, small_scene["x"], small_scene["y"], small_scene["z"])
        keep_g, dmap_g, kept_g = ctx.cull_frame(f)
        assert np.array_equal(dmap_g.view(np.uint32), dmap_r.view(np.uint32)), f
        assert np.array_equal(keep_g, keep_r), f
        assert kept_g == kept_r
        assert 0 < kept_r < (dmap_r < 1e30).sum() * 50


def test_cull_frame_without_depth_buffer(gpu_ctx_factory, oracle, small_scene):
    from pointcloudprocessor_amd import capi

    ctx = gpu_ctx_factory()
    cull = capi.default_cull_params()
    cull.enable_depth_buffer_culling = 0
    cd = _setup(ctx, capi, small_scene, cull=cull)
    ocp = oracle.default_cull_params()
    ocp.enable_depth_buffer_culling = 0
    w2c, _ = oracle.pose_to_matrices(small_scene["poses"][2])
    keep_r, _, kept_r = oracle.cull_frame(cam_struct(oracle, cd), ocp, w2c, small_scene["x"], small_scene["y"],
                                          small_scene["z"])
    keep_g, _, kept_g = ctx.cull_frame(2)
    assert np.array_equal(keep_g, keep_r) and kept_g == kept_r


def _colour_close(got_rgb, ref):
    """uint8 equal except where the fp32 quotient is within 1e-4*255 of an integer."""
    diff = got_rgb.astype(np.int32) - ref["rgb"].astype(np.int32)
    bad = np.nonzero(diff.any(axis=1))[0]
    for i in bad:
        s = ref["top_score"][i]
        c = ref["top_rgb"][i]
        m = (ref["top_frame"][i] >= 0)
        tot = float(np.sum(s[m].astype(np.float64)))
        for ch, sh in enumerate((16, 8, 0)):
            q = float(np.sum(((c[m] >> sh) & 0xFF).astype(np.float64) * s[m])) / tot
            assert abs(int(got_rgb[i, ch]) - int(ref["rgb"][i, ch])) <= 1 and abs(q - round(q)) < 1e-4 * 255, (i, ch, q)
    return len(bad)


def test_colorize_matches_oracle(gpu_ctx_factory, oracle, small_scene):
    from pointcloudprocessor_amd import capi

    ctx = gpu_ctx_factory()
    cd = _setup(ctx, capi, small_scene)
    ref = oracle.colorize(cam_struct(oracle, cd), oracle.default_cull_params(), small_scene["x"], small_scene["y"],
                          small_scene["z"], small_scene["poses"], small_scene["images"])
    assert ref["has"].sum() > 500 and ref["count"].max() >= 3
    # staged run: depth maps, state, top-5 lists
    ctx.colour_reset()
    ctx.depth_pass()
    for f, pose in enumerate(small_scene["poses"]):
        w2c, _ = oracle.pose_to_matrices(pose)
        _, dmap_r, _ = oracle.cull_frame(cam_struct(oracle, cd), oracle.default_cull_params(), w2c, small_scene["x"],
                                         small_scene["y"], small_scene["z"])
        assert np.array_equal(ctx.download_depth_map(f).view(np.uint32), dmap_r.view(np.uint32)), f
    ctx.colour_pass()
    got = ctx.colour_finalise(want_top=True)
    assert np.array_equal(got["count"], ref["count"])
    assert np.array_equal(got["top_frame"], ref["top_frame"])
    assert np.array_equal(got["top_rgb"], ref["top_rgb"])
    np.testing.assert_allclose(got["top_score"], ref["top_score"], rtol=1e-4, atol=0)
    _colour_close(got["rgb"], ref)
    assert np.array_equal(got["has"], (got["rgb"] != 0).any(axis=1).astype(np.uint8))
    # one-shot run gives the same colours
    one = ctx.colorize()
    assert np.array_equal(one["rgb"], got["rgb"]) and np.array_equal(one["has"], got["has"])
    # two batches of keyframes give the same result as one
    ctx.colour_reset()
    ctx.depth_pass(0, 3)
    ctx.depth_pass(3, 6)
    ctx.colour_pass(0, 2)
    ctx.colour_pass(2, 6)
    two = ctx.colour_finalise(want_top=True)
    for k in ("rgb", "has", "count", "top_frame", "top_rgb", "top_score"):
        assert np.array_equal(two[k], got[k]), k


def test_frame_visible_with_masks(gpu_ctx_factory, oracle, small_scene):
    from pointcloudprocessor_amd import capi

    ctx = gpu_ctx_factory()
    cd = _setup(ctx, capi, small_scene, with_masks=True)
    ocam, ocp = cam_struct(oracle, cd), oracle.default_cull_params()
    for f in (0, 3, 5):
        ref = oracle.frame_visible(ocam, ocp, small_scene["poses"][f], small_scene["x"], small_scene["y"],
                                   small_scene["z"], small_scene["images"][f], small_scene["masks"][f])
        got = ctx.frame_visible(f)
        assert got["count"] == len(ref["index"]) > 50
        for k in ("index", "rgb", "mask", "xyz_cam", "xyz_world"):
            assert np.array_equal(got[k], ref[k]), (f, k)
    # capacity smaller than the count: truncated prefix, true count reported
    got = ctx.frame_visible(0, capacity=10)
    ref = oracle.frame_visible(ocam, ocp, small_scene["poses"][0], small_scene["x"], small_scene["y"],
                               small_scene["z"], small_scene["images"][0], small_scene["masks"][0])
    assert got["count"] == len(ref["index"]) and np.array_equal(got["index"], ref["index"][:10])


def test_errors_are_loud(gpu_ctx_factory, small_scene):
    from pointcloudprocessor_amd import capi

    ctx = gpu_ctx_factory()
    with pytest.raises(capi.PcpError) as e:
        ctx.n = 0
        ctx.project_frame(0)
    assert e.value.code == capi.PCP_ERR_STATE
    ctx.set_camera(cam_struct(capi, small_scene["cam"]))
    ctx.upload_cloud(small_scene["x"], small_scene["y"], small_scene["z"])
    ctx.set_frames(small_scene["poses"])
    with pytest.raises(capi.PcpError) as e:
        ctx.project_frame(99)
    assert e.value.code == capi.PCP_ERR_RANGE
    with pytest.raises(capi.PcpError) as e:
        ctx.colour_pass()
    assert e.value.code == capi.PCP_ERR_STATE


def test_optimised_extrinsic_branches(gpu_ctx_factory, oracle, small_scene):
    """T_camera_lidar_optimized: one global matrix (NID branch, PointCloudProcessor.cpp:504-509)
    and one per keyframe (manual-guess branch, :510-519): general fp32 affine inverse."""
    from pointcloudprocessor_amd import capi

    rng = np.random.default_rng(2)

    def small_T():
        a = rng.normal(0, 0.01, 3)
        K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
        T = np.eye(4)
        T[:3, :3] = np.eye(3) + K + 0.5 * K @ K
        T[:3, 3] = rng.normal(0, 0.03, 3)
        return T

    F = len(small_scene["poses"])
    cd = small_scene["cam"]
    for T_opt in (small_T(), np.stack([small_T() for _ in range(F)])):
        ctx = gpu_ctx_factory()
        ctx.set_camera(cam_struct(capi, cd))
        ctx.upload_cloud(small_scene["x"], small_scene["y"], small_scene["z"])
        ctx.set_frames(small_scene["poses"], T_opt=T_opt)
        for f, im in enumerate(small_scene["images"]):
            ctx.upload_image(f, im)
        got = ctx.colorize()
        ref = oracle.colorize(cam_struct(oracle, cd), oracle.default_cull_params(), small_scene["x"], small_scene["y"],
                              small_scene["z"], small_scene["poses"], small_scene["images"], T_opt=T_opt)
        assert np.array_equal(got["rgb"], ref["rgb"]) and np.array_equal(got["has"], ref["has"])
        assert ref["has"].sum() > 300
        ctx.close()


@pytest.mark.parametrize("ds,slack,cull", [(7, 0.05, None), (1, 0.0, None), (14, 0.2, (700, 400)), (20, 0.05, (300, 200))])
def test_cull_parameter_variants(gpu_ctx_factory, oracle, small_scene, ds, slack, cull):
    """Other downsample factors / slacks and a cull size that differs from the image size
    (the reference hard-codes {4096,3000} whatever the image is, PointCloudProcessor.cpp:525)."""
    from pointcloudprocessor_amd import capi

    cd = dict(small_scene["cam"])
    if cull:
        cd["cull_width"], cd["cull_height"] = cull
    cp_g, cp_o = capi.default_cull_params(), oracle.default_cull_params()
    for cp in (cp_g, cp_o):
        cp.downsample_factor = ds
        cp.depth_slack = slack
    ctx = gpu_ctx_factory()
    ctx.set_camera(cam_struct(capi, cd), cp_g)
    ctx.upload_cloud(small_scene["x"], small_scene["y"], small_scene["z"])
    ctx.set_frames(small_scene["poses"])
    for f, im in enumerate(small_scene["images"]):
        ctx.upload_image(f, im)
    ocam = cam_struct(oracle, cd)
    for f in (0, 3):
        w2c, _ = oracle.pose_to_matrices(small_scene["poses"][f])
        keep_r, dmap_r, kept_r = oracle.cull_frame(ocam, cp_o, w2c, small_scene["x"], small_scene["y"], small_scene["z"])
        keep_g, dmap_g, kept_g = ctx.cull_frame(f)
        assert dmap_g.shape == dmap_r.shape
        assert np.array_equal(dmap_g.view(np.uint32), dmap_r.view(np.uint32)) and np.array_equal(keep_g, keep_r)
    got = ctx.colorize()
    ref = oracle.colorize(ocam, cp_o, small_scene["x"], small_scene["y"], small_scene["z"], small_scene["poses"],
                          small_scene["images"])
    assert np.array_equal(got["rgb"], ref["rgb"]) and np.array_equal(got["has"], ref["has"])
    ctx.close()


def test_non_finite_points_are_rejected_not_fatal(gpu_ctx_factory, oracle, small_scene):
    """PCL clouds may carry NaN / inf points (is_dense == false): they never project
    (z > 0 is false / projection not finite) and must not disturb their neighbours."""
    from pointcloudprocessor_amd import capi

    x, y, z = (small_scene[k].copy() for k in "xyz")
    x[10], y[200], z[3000] = np.nan, np.inf, -np.inf
    x[4000] = y[4000] = z[4000] = np.nan
    z[5000] = 3.0e38
    cd = small_scene["cam"]
    ctx = gpu_ctx_factory()
    ctx.set_camera(cam_struct(capi, cd))
    ctx.upload_cloud(x, y, z)
    ctx.set_frames(small_scene["poses"])
    for f, im in enumerate(small_scene["images"]):
        ctx.upload_image(f, im)
    got = ctx.colorize()
    ref = oracle.colorize(cam_struct(oracle, cd), oracle.default_cull_params(), x, y, z, small_scene["poses"],
                          small_scene["images"])
    assert np.array_equal(got["rgb"], ref["rgb"]) and np.array_equal(got["has"], ref["has"])
    assert not got["has"][[10, 200, 3000, 4000]].any()
    w2c, _ = oracle.pose_to_matrices(small_scene["poses"][1])
    p_ref = oracle.project_frame(cam_struct(oracle, cd), oracle.default_cull_params(), w2c, x, y, z)
    p_got = ctx.project_frame(1)
    assert np.array_equal(p_got["cell"], p_ref["cell"]) and np.array_equal(p_got["pixel"], p_ref["pixel"])
    ctx.close()


@pytest.mark.parametrize("far", ["pose", "points"])
def test_huge_finite_operands_keep_the_general_division(gpu_ctx_factory, oracle, small_scene, far):
    """The batched passes leave out the per-visit test for non-finite camera coordinates when every coordinate of a tile's points
    and every entry of every keyframe's matrix is below 2^40 (DevCamera::frames_bounded, pcp_device.hpp divide_xy_by_z).  A pose
    translated by 1e30 switches that off for the whole run; points at 1e12 (just above 2^40) for their tiles.  Same colours as
    the oracle either way."""
    from pointcloudprocessor_amd import capi

    x, y, z = (small_scene[k].copy() for k in "xyz")
    poses = [list(p) for p in small_scene["poses"]]
    if far == "pose":
        poses[1][0] = 1.0e30
        poses[1][2] = -3.0e29
    else:
        x[100:164] = 1.2e12
        z[7000] = -1.15e12
        y[9000] = 1.0e12  # below 2^40: stays on the short path, far outside every image
    cd = small_scene["cam"]
    ctx = gpu_ctx_factory()
    ctx.set_camera(cam_struct(capi, cd))
    ctx.upload_cloud(x, y, z)
    ctx.set_frames(poses)
    for f, im in enumerate(small_scene["images"]):
        ctx.upload_image(f, im)
    got = ctx.colorize()
    ref = oracle.colorize(cam_struct(oracle, cd), oracle.default_cull_params(), x, y, z, poses, small_scene["images"])
    assert np.array_equal(got["rgb"], ref["rgb"]) and np.array_equal(got["has"], ref["has"])
    assert got["has"].sum() > 100
    ctx.close()


@pytest.mark.parametrize("stride", [12, 16, 32, 13])
def test_upload_aos_equals_soa(gpu_ctx_factory, small_scene, stride):
    """pcl::PointXYZI records (x y z first, any stride) uploaded as they are give the same cloud as SoA arrays;
    strides that are not a multiple of 4 take the host path."""
    from pointcloudprocessor_amd import capi

    x, y, z = small_scene["x"], small_scene["y"], small_scene["z"]
    n = len(x)
    raw = np.zeros((n, stride), np.uint8)
    raw[:, 0:4] = x.view(np.uint8).reshape(n, 4)
    raw[:, 4:8] = y.view(np.uint8).reshape(n, 4)
    raw[:, 8:12] = z.view(np.uint8).reshape(n, 4)
    raw[:, 12:] = 0xAB
    outs = []
    for aos in (False, True):
        ctx = gpu_ctx_factory()
        ctx.set_camera(cam_struct(capi, small_scene["cam"]))
        if aos:
            ctx.upload_cloud_aos(raw)
        else:
            ctx.upload_cloud(x, y, z)
        ctx.set_frames(small_scene["poses"])
        outs.append(ctx.project_frame(2))
        ctx.close()
    for k in ("cell", "pixel", "xc", "yc", "zc"):
        assert np.array_equal(outs[0][k], outs[1][k]), k


def test_async_download_ring_and_mask_diagnostics(gpu_ctx_factory, small_scene):
    """Two-buffer consumer: async downloads of consecutive runs, pcp_download_wait_previous before a buffer is reused,
    equal to the synchronous download; pcp_tile_masks agrees with pcp_tile_mask_density."""
    import torch

    from pointcloudprocessor_amd import capi

    ctx = gpu_ctx_factory()
    _setup(ctx, capi, small_scene)
    n = len(small_scene["x"])
    ctx.download_wait_previous()  # nothing pending: a no-op
    ring = [torch.empty(n, dtype=torch.int32).pin_memory() for _ in range(2)]
    for s in range(5):
        ctx.colorize(download=False)
        ctx.download_result_packed_async(ring[s & 1].data_ptr())
        ctx.download_wait_previous()
    ctx.synchronize()
    ref = ctx.download_result_packed()
    assert np.array_equal(ring[0].numpy().view(np.uint32), ref) and np.array_equal(ring[1].numpy().view(np.uint32), ref)
    masks = ctx.tile_masks()
    assert masks.shape == ((n + 63) // 64, 1) and masks.dtype == np.uint32
    bits = np.unpackbits(masks.view(np.uint8)).sum()
    assert abs(bits / (masks.shape[0] * len(small_scene["poses"])) - ctx.tile_mask_density()) < 1e-12
    assert (masks >> len(small_scene["poses"])).max() == 0  # no bits beyond the keyframe count
    ctx.close()


@pytest.mark.parametrize("n", [256 * 7 + 1, 256 * 11 + 64, 256 * 3 + 33])
def test_tail_workgroup_with_dead_wavefronts(gpu_ctx_factory, oracle, small_scene, n):
    """n % 256 in (0, 64] leaves up to three wavefronts of the last 256-thread workgroup without a tile; with more than
    32 keyframes (several mask words per tile) they once read tile-mask words past the allocation."""
    from pointcloudprocessor_amd import capi, synth

    F = 70
    ctx = gpu_ctx_factory()
    cd = small_scene["cam"]
    ctx.set_camera(cam_struct(capi, cd))
    x, y, z = (small_scene[k][:n] for k in "xyz")
    ctx.upload_cloud(x, y, z)
    poses, _ = synth.make_trajectory(F)
    ctx.set_frames(poses)
    imgs = [synth.make_image(f, cd["image_width"], cd["image_height"]) for f in range(F)]
    for f, im in enumerate(imgs):
        ctx.upload_image(f, im)
    got = ctx.colorize()
    ref = oracle.colorize(cam_struct(oracle, cd), oracle.default_cull_params(), x, y, z, poses, imgs)
    assert ref["has"].sum() > 100
    _colour_close(got["rgb"], ref)
    assert np.array_equal(got["has"] > 0, ref["has"] > 0)


@pytest.mark.parametrize("cull_mode,match_mode", [(1, 0), (0, 1), (1, 1)])
def test_hpr_candidates_and_roundtrip_modes_match_oracle(gpu_ctx_factory, oracle, small_scene, cull_mode, match_mode):
    """PCP_CULL_HPR_CANDIDATES (view_culling.cpp:276-288) and PCP_MATCH_ROUNDTRIP (PointCloudProcessor.cpp:555,
    571-579): keep masks, per-keyframe visible lists and the whole colour run against the oracle in the same modes --
    everything bit-equal, scores included."""
    from pointcloudprocessor_amd import capi

    ctx = gpu_ctx_factory()
    cull = capi.default_cull_params()
    cull.cull_mode, cull.match_mode = cull_mode, match_mode
    cd = _setup(ctx, capi, small_scene, cull=cull, with_masks=True)
    ocam, ocp = cam_struct(oracle, cd), oracle.default_cull_params()
    ocp.cull_mode, ocp.match_mode = cull_mode, match_mode
    s = small_scene
    for f in (0, 4):
        w2c, _ = oracle.pose_to_matrices(s["poses"][f])
        keep_r, _, kept_r = oracle.cull_frame(ocam, ocp, w2c, s["x"], s["y"], s["z"])
        keep_g, _, kept_g = ctx.cull_frame(f)
        assert np.array_equal(keep_g, keep_r) and kept_g == kept_r > 100
        ref = oracle.frame_visible(ocam, ocp, s["poses"][f], s["x"], s["y"], s["z"], s["images"][f], s["masks"][f])
        got = ctx.frame_visible(f)
        for k in ("index", "rgb", "mask", "xyz_cam", "xyz_world"):
            assert np.array_equal(got[k], ref[k]), (f, k)
    ref = oracle.colorize(ocam, ocp, s["x"], s["y"], s["z"], s["poses"], s["images"])
    ctx.colour_reset()
    ctx.depth_pass()
    ctx.colour_pass()
    got = ctx.colour_finalise(want_top=True)
    for k in ("count", "top_frame", "top_rgb", "top_score", "rgb", "has"):
        assert np.array_equal(got[k], ref[k]), k
    one = ctx.colorize()
    assert np.array_equal(one["rgb"], ref["rgb"])
    if cull_mode == 1:  # no occlusion test: more samples than the z-buffer run
        zb = oracle.colorize(ocam, oracle.default_cull_params(), s["x"], s["y"], s["z"], s["poses"], s["images"],
                             want_top=False)
        assert ref["count"].sum() > zb["count"].sum()


def test_result_unpermute_switch_gives_identical_results(gpu_ctx_factory, small_scene, monkeypatch):
    """PCP_RESULT_UNPERMUTE=1: the colour pass stores its packed result in Morton order with coalesced stores and a
    second kernel un-permutes it (8x fewer bytes written, 35 us slower per step at 10 M points): same bits."""
    from pointcloudprocessor_amd import capi

    ctx = gpu_ctx_factory()
    _setup(ctx, capi, small_scene)
    base = ctx.colorize()
    ctx.colour_reset()
    ctx.depth_pass()
    ctx.colour_pass(0, 3)
    ctx.colour_pass(3, 6)
    base2 = ctx.colour_finalise()
    monkeypatch.setenv("PCP_RESULT_UNPERMUTE", "1")
    alt = ctx.colorize()
    ctx.colour_reset()
    ctx.depth_pass()
    ctx.colour_pass(0, 3)
    ctx.colour_pass(3, 6)
    alt2 = ctx.colour_finalise()
    for a, b in ((base, alt), (base2, alt2), (base, base2)):
        assert np.array_equal(a["rgb"], b["rgb"]) and np.array_equal(a["has"], b["has"])
    assert base["has"].sum() > 500


def test_sorted_result_is_unpermuted_by_every_reader(gpu_ctx_factory, small_scene, monkeypatch):
    """PCP_RESULT_UNPERMUTE=2 (round 5; measured slower than the scattered store, kept as a switch): the colour pass stores its packed result in the sorted order it walks and
    whatever reads the result un-permutes it -- the byte outputs of pcp_colorize, the packed downloads (synchronous; on the copy
    stream into pinned and into pageable memory: an un-permuting kernel into a scratch buffer, then the copy engine), the
    device array.  All equal to the scattered form of rounds 2-4 (PCP_RESULT_UNPERMUTE=0)."""
    import torch

    from pointcloudprocessor_amd import capi

    ctx = gpu_ctx_factory()
    _setup(ctx, capi, small_scene)
    monkeypatch.setenv("PCP_RESULT_UNPERMUTE", "0")
    base = ctx.colorize()
    want = ctx.download_result_packed()
    assert np.array_equal(want & 0xFF, base["rgb"][:, 0]) and np.array_equal(want >> 24, base["has"])
    monkeypatch.setenv("PCP_RESULT_UNPERMUTE", "2")
    n = len(want)
    got = ctx.colorize()  # the byte-splitting kernel gathers through inv_perm
    assert np.array_equal(got["rgb"], base["rgb"]) and np.array_equal(got["has"], base["has"])
    pinned = torch.zeros(n, dtype=torch.int32).pin_memory()
    for rep in range(3):  # both result buffers in turn
        ctx.colorize(download=False)
        ctx.download_result_packed_async(pinned.data_ptr())
        ctx.download_wait_previous()
        ctx.synchronize()
        ctx.colorize(download=False)
        ctx.download_wait_previous()
        assert np.array_equal(pinned.numpy().view(np.uint32), want), rep
        pinned.zero_()
    pageable = np.zeros(n, np.uint32)
    ctx.colorize(download=False)
    ctx.download_result_packed_async(pageable.ctypes.data)
    ctx.colorize(download=False)
    ctx.download_wait_previous()
    ctx.synchronize()
    assert np.array_equal(pageable, want)
    ctx.colorize(download=False)
    assert np.array_equal(ctx.download_result_packed(), want)  # synchronous: un-permuted on the device first
    ctx.colorize(download=False)
    ptr, words = ctx.colour_result_device()  # the device array in input order
    assert words == n
    dev = torch.zeros(n, dtype=torch.int32, device="cuda")
    ctx.synchronize()
    import ctypes

    hip = ctypes.CDLL("libamdhip64.so")
    assert hip.hipMemcpy(ctypes.c_void_p(dev.data_ptr()), ctypes.c_void_p(ptr), ctypes.c_size_t(4 * n), 3) == 0  # device to device
    assert np.array_equal(dev.cpu().numpy().view(np.uint32), want)
    # the multi-batch path (k_finalise) too
    ctx.colour_reset()
    ctx.depth_pass()
    ctx.colour_pass(0, 3)
    ctx.colour_pass(3, 6)
    fin = ctx.colour_finalise()
    assert np.array_equal(fin["rgb"], base["rgb"]) and np.array_equal(fin["has"], base["has"])
    assert np.array_equal(ctx.download_result_packed(), want)


def test_async_uploads_from_pinned_memory_are_ordered_against_the_passes(gpu_ctx_factory, oracle, small_scene):
    """pcp_upload_image_async from pinned host memory (read in place by the pack kernel, two upload lanes, an event per
    keyframe): colour batches wait for their own keyframes only, and a re-upload of a keyframe waits for the colour
    pass that still samples its previous image.  Results equal the synchronous path / the oracle bit for bit."""
    import torch

    from pointcloudprocessor_amd import capi

    s = small_scene
    cd = s["cam"]
    F = len(s["poses"])
    H, W = cd["image_height"], cd["image_width"]
    ctx = gpu_ctx_factory()
    ctx.set_camera(cam_struct(capi, cd))
    ctx.upload_cloud(s["x"], s["y"], s["z"])
    ctx.set_frames(s["poses"])
    stage = torch.empty((2 * F, H, W, 3), dtype=torch.uint8).pin_memory()
    snp = stage.numpy()
    for f in range(F):
        snp[f] = s["images"][f]
        snp[F + f] = s["images"][(f + 1) % F][::-1, ::-1]  # a second, different set
    ocam, ocp = cam_struct(oracle, cd), oracle.default_cull_params()
    refs = [oracle.colorize(ocam, ocp, s["x"], s["y"], s["z"], s["poses"], [snp[k * F + f] for f in range(F)], want_top=False)
            for k in range(2)]
    assert (refs[0]["rgb"] != refs[1]["rgb"]).any()
    for rep in range(3):
        for k in range(2):
            ctx.colour_reset()
            for f in range(F):  # queued back to back; the set of the previous round may still be sampled
                ctx.upload_image_async(f, snp[k * F + f])
            ctx.depth_pass()
            for f0 in range(0, F, 2):
                ctx.colour_pass(f0, min(F, f0 + 2))
            got = ctx.colour_finalise()
            assert np.array_equal(got["rgb"], refs[k]["rgb"]) and np.array_equal(got["has"], refs[k]["has"]), (rep, k)
            one = ctx.colorize()  # one-shot pass over the same texels
            assert np.array_equal(one["rgb"], refs[k]["rgb"])
    # device pointers are accepted too (frames gathered over xGMI by a multi-GPU host)
    dev = stage[:F].to("cuda:0")
    torch.cuda.synchronize()
    for f in range(F):
        ctx.upload_image_async_ptr(f, dev[f].data_ptr(), W * 3)
    got = ctx.colorize()
    assert np.array_equal(got["rgb"], refs[0]["rgb"])
    ctx.synchronize()


@pytest.mark.parametrize("z1,z2", [(1.25, 1.3125), (1.0, 1.09375), (1.7109375, 1.84375)])
def test_keep_rule_within_rounding_of_the_depth_limit(gpu_ctx_factory, oracle, z1, z2):
    """The colour pass decides `!(r > depth + slack)` from the squared range and takes the square root only when
    s is within 2^-40 of the squared limit (keep_by_depth).  Two points on the optical axis (ranges z1 < z2, exact in
    fp32) and slacks that put the limit z1 + slack exactly on z2, one and 256 ulps of fp64 to either side of it, and
    clearly off: every outcome must be the reference's (view_culling.cpp:144,157)."""
    from pointcloudprocessor_amd import capi, synth

    cd = synth.camera_dict("tiny")
    img = synth.make_image(0, cd["image_width"], cd["image_height"])
    pose = np.array([[0, 0, 0, 0, 0, 0, 1]], np.float64)  # identity: camera frame == world frame
    x = np.array([0.0, 0.0, 0.3], np.float32)  # the third point keeps the bounding box from degenerating
    y = np.array([0.0, 0.0, 0.2], np.float32)
    z = np.array([z1, z2, 1.5], np.float32)
    assert float(z[0]) == z1 and float(z[1]) == z2
    gap, ulp = z2 - z1, 2.0 ** -52
    outcomes = []
    for slack in (gap - 256 * ulp, gap - ulp, gap, gap + ulp, gap + 256 * ulp, 0.5 * gap, 2.0 * gap):
        cp_g, cp_o = capi.default_cull_params(), oracle.default_cull_params()
        cp_g.depth_slack = cp_o.depth_slack = slack
        ctx = gpu_ctx_factory()
        ctx.set_camera(cam_struct(capi, cd), cp_g)
        ctx.upload_cloud(x, y, z)
        ctx.set_frames(pose)
        ctx.upload_image(0, img)
        ctx.colour_reset()
        ctx.depth_pass()
        ctx.colour_pass()
        got = ctx.colour_finalise(want_top=True)
        ref = oracle.colorize(cam_struct(oracle, cd), cp_o, x, y, z, pose, [img])
        assert np.array_equal(got["count"], ref["count"]), (slack, got["count"], ref["count"])
        assert np.array_equal(got["rgb"], ref["rgb"]) and np.array_equal(got["has"], ref["has"])
        keep, dmap, _ = ctx.cull_frame(0)
        keep_r, dmap_r, _ = oracle.cull_frame(cam_struct(oracle, cd), cp_o, oracle.pose_to_matrices(pose[0])[0], x, y, z)
        assert np.array_equal(keep, keep_r) and np.array_equal(dmap.view(np.uint32), dmap_r.view(np.uint32))
        assert ref["count"][0] == 1
        outcomes.append(int(ref["count"][1]))
        ctx.close()
    assert outcomes == [0, 0, 1, 1, 1, 0, 1]  # dropped below the limit, kept on it and above


def test_block_uploads_from_a_pinned_arena(gpu_ctx_factory, oracle, small_scene, monkeypatch):
    """pcp_upload_images_block: runs of keyframes copied by one DMA per block and packed from the device copy, mixed with
    single-keyframe uploads of the same keyframes on the other lane, blocks smaller than a run (several DMAs), pageable
    and device memory as the source -- colours equal the oracle's for whichever image set was uploaded last."""
    import torch

    from pointcloudprocessor_amd import capi

    s = small_scene
    cd = s["cam"]
    F = len(s["poses"])
    H, W = cd["image_height"], cd["image_width"]
    ctx = gpu_ctx_factory()
    ctx.set_camera(cam_struct(capi, cd))
    ctx.upload_cloud(s["x"], s["y"], s["z"])
    ctx.set_frames(s["poses"])
    stage = torch.empty((2 * F, H, W, 3), dtype=torch.uint8).pin_memory()
    snp = stage.numpy()
    for f in range(F):
        snp[f] = s["images"][f]
        snp[F + f] = s["images"][(f + 1) % F][::-1, ::-1]
    ocam, ocp = cam_struct(oracle, cd), oracle.default_cull_params()
    refs = [oracle.colorize(ocam, ocp, s["x"], s["y"], s["z"], s["poses"], [snp[k * F + f] for f in range(F)], want_top=False)
            for k in range(2)]
    for rep in range(3):
        for k in range(2):
            ctx.colour_reset()
            if rep == 1:  # some keyframes singly first: the block's lane must wait for uploads in flight on the other lane
                for f in (1, 4):
                    ctx.upload_image_async(f, snp[(1 - k) * F + f])
            ctx.upload_images_block(0, snp[k * F:k * F + 2])
            ctx.upload_images_block(2, snp[k * F + 2:(k + 1) * F])
            ctx.depth_pass()
            for f0 in range(0, F, 2):
                ctx.colour_pass(f0, min(F, f0 + 2))
            got = ctx.colour_finalise()
            assert np.array_equal(got["rgb"], refs[k]["rgb"]) and np.array_equal(got["has"], refs[k]["has"]), (rep, k)
    # pageable memory and device memory take the keyframe-by-keyframe path
    ctx.upload_images_block(0, np.ascontiguousarray(snp[F:2 * F]).copy())
    assert np.array_equal(ctx.colorize()["rgb"], refs[1]["rgb"])
    ctx.synchronize()
    with pytest.raises(capi.PcpError):
        ctx.upload_images_block(F - 1, snp[:2])
