"""The conservative fp32 rejection test (pcp_device.hpp surely_rejected) must never
change a result: cells and pixels stay bit-exact with the oracle for distortion
models of every sign, for points hugging the acceptance box, for extreme angles and
for tiny / huge depths."""
import numpy as np
import pytest

from conftest import cam_struct

pytestmark = pytest.mark.gpu

CAMS = {
    "reference": {},
    "barrel": dict(k1=-0.28, k2=0.07, p1=0.0012, p2=-0.0009, k3=-0.004),
    "pincushion": dict(k1=0.35, k2=0.4, p1=-0.01, p2=0.02, k3=0.2),
    "tangential": dict(k1=0.0, k2=0.0, p1=0.05, p2=-0.04, k3=0.0),
    "no_distortion": dict(k1=0.0, k2=0.0, p1=0.0, p2=0.0, k3=0.0),
    "fold_back": dict(k1=-0.9, k2=0.0, p1=0.0, p2=0.0, k3=0.0),  # projection folds back into the image at large r
}


def _edge_points(cam, rng, n):
    """Camera-frame points whose undistorted projection hugs the acceptance box edges."""
    W, H = cam["image_width"], cam["image_height"]
    edges_u = np.array([-14.5, -14.0, -1.0, 0.0, W - 1, W, W + 0.5, 14 * (W // 14), 14 * (W // 14) + 0.5])
    edges_v = np.array([-14.5, -14.0, -1.0, 0.0, H - 1, H, H + 0.5, 14 * (H // 14), 14 * (H // 14) + 0.5])
    u = rng.choice(edges_u, n) + rng.normal(0, 0.3, n)
    v = rng.uniform(-30, H + 30, n)
    swap = rng.random(n) < 0.5
    u2 = np.where(swap, rng.uniform(-30, W + 30, n), u)
    v2 = np.where(swap, rng.choice(edges_v, n) + rng.normal(0, 0.3, n), v)
    z = np.exp(rng.uniform(np.log(0.05), np.log(80.0), n))
    x = (u2 - cam["cx"]) / cam["fx"] * z
    y = (v2 - cam["cy"]) / cam["fy"] * z
    return x, y, z


@pytest.mark.parametrize("name", list(CAMS))
def test_pretest_never_changes_results(gpu_ctx_factory, oracle, name):
    from pointcloudprocessor_amd import capi, synth

    rng = np.random.default_rng(abs(hash(name)) % (2 ** 31))
    cd = dict(synth.camera_dict("tiny"))
    cd.update(CAMS[name])
    n = 60000
    # camera-frame clouds: uniform directions incl. grazing angles, plus box-edge huggers,
    # plus extreme depths; identity pose so camera frame == world frame
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    r = np.exp(rng.uniform(np.log(1e-3), np.log(1e4), n))
    pts = d * r[:, None]
    ex, ey, ez = _edge_points(cd, rng, n)
    graze = np.stack([rng.normal(0, 5, n), rng.normal(0, 5, n), np.abs(rng.normal(0, 1e-4, n)) + 1e-30], 1)
    allp = np.concatenate([pts, np.stack([ex, ey, ez], 1), graze]).astype(np.float32)
    x, y, z = allp[:, 0].copy(), allp[:, 1].copy(), allp[:, 2].copy()
    pose = [0, 0, 0, 1, 0, 0, 0]
    for zbuf in (1, 0):
        ctx = gpu_ctx_factory()
        cull = capi.default_cull_params()
        cull.enable_depth_buffer_culling = zbuf
        ctx.set_camera(cam_struct(capi, cd), cull)
        ctx.upload_cloud(x, y, z)
        ctx.set_frames([pose])
        got = ctx.project_frame(0)
        ocp = oracle.default_cull_params()
        ocp.enable_depth_buffer_culling = zbuf
        w2c, _ = oracle.pose_to_matrices(pose)
        ref = oracle.project_frame(cam_struct(oracle, cd), ocp, w2c, x, y, z)
        assert np.array_equal(got["cell"], ref["cell"]), (name, zbuf, np.nonzero(got["cell"] != ref["cell"])[0][:5])
        assert np.array_equal(got["pixel"], ref["pixel"]), (name, zbuf)
        cand = ref["cell"] != -1
        assert np.array_equal(got["range"][cand], ref["range"][cand])
        assert (ref["cell"] >= 0).sum() > 1000 and (ref["pixel"] >= 0).sum() > 1000
        ctx.close()


def test_pretest_on_scene_all_keyframes(gpu_ctx_factory, oracle):
    """1 M scene points x 8 keyframes with the 1920x1080 camera: every index exact."""
    from pointcloudprocessor_amd import capi, synth

    cd = synth.camera_dict("cfg")
    x, y, z, _ = synth.make_cloud(1_000_000)
    poses, _ = synth.make_trajectory(8)
    ctx = gpu_ctx_factory()
    ctx.set_camera(cam_struct(capi, cd))
    ctx.upload_cloud(x, y, z)
    ctx.set_frames(poses)
    ocam, ocp = cam_struct(oracle, cd), oracle.default_cull_params()
    for f in range(8):
        w2c, _ = oracle.pose_to_matrices(poses[f])
        ref = oracle.project_frame(ocam, ocp, w2c, x, y, z)
        got = ctx.project_frame(f, want_cam=False)
        assert np.array_equal(got["cell"], ref["cell"]) and np.array_equal(got["pixel"], ref["pixel"]), f
        keep_r, dmap_r, kept_r = oracle.cull_frame(ocam, ocp, w2c, x, y, z, 8)
        keep_g, dmap_g, kept_g = ctx.cull_frame(f)
        assert np.array_equal(dmap_g.view(np.uint32), dmap_r.view(np.uint32)) and np.array_equal(keep_g, keep_r)


@pytest.mark.parametrize("name", ["reference", "barrel", "fold_back", "tangential"])
def test_tile_culling_never_changes_colours(gpu_ctx_factory, oracle, name):
    """Whole batched run (tile masks + pre-test) against the oracle for several
    distortion models: depth maps bit-exact, top-5 lists and colours equal."""
    from pointcloudprocessor_amd import capi, synth

    cd = dict(synth.camera_dict("tiny"))
    cd.update(CAMS[name])
    x, y, z, _ = synth.make_cloud(150_000, seed=5)
    poses, _ = synth.make_trajectory(40)
    poses = poses[::5]  # 8 well-separated keyframes
    W, H = cd["image_width"], cd["image_height"]
    imgs = [synth.make_image(f, W, H) for f in range(len(poses))]
    ctx = gpu_ctx_factory()
    ctx.set_camera(cam_struct(capi, cd))
    ctx.upload_cloud(x, y, z)
    ctx.set_frames(poses)
    for f, im in enumerate(imgs):
        ctx.upload_image(f, im)
    ocam, ocp = cam_struct(oracle, cd), oracle.default_cull_params()
    ref = oracle.colorize(ocam, ocp, x, y, z, poses, imgs, threads=8)
    ctx.colour_reset()
    ctx.depth_pass()
    for f in range(len(poses)):
        w2c, _ = oracle.pose_to_matrices(poses[f])
        _, dmap, _ = oracle.cull_frame(ocam, ocp, w2c, x, y, z, 8)
        assert np.array_equal(ctx.download_depth_map(f).view(np.uint32), dmap.view(np.uint32)), (name, f)
    ctx.colour_pass()
    got = ctx.colour_finalise(want_top=True)
    assert np.array_equal(got["count"], ref["count"]), name
    assert np.array_equal(got["top_frame"], ref["top_frame"]) and np.array_equal(got["top_rgb"], ref["top_rgb"])
    assert np.array_equal(got["top_score"], ref["top_score"])
    assert np.array_equal(got["rgb"], ref["rgb"]) and np.array_equal(got["has"], ref["has"])
    assert ref["has"].sum() > 2000
    ctx.close()


def test_colorize_1m_points_16_keyframes_cfg_camera(gpu_ctx_factory, oracle):
    from pointcloudprocessor_amd import capi, synth

    cd = synth.camera_dict("cfg")
    x, y, z, _ = synth.make_cloud(1_000_000)
    poses, _ = synth.make_trajectory(16)
    imgs = [synth.make_image(f, 1920, 1080) for f in range(16)]
    ctx = gpu_ctx_factory()
    ctx.set_camera(cam_struct(capi, cd))
    ctx.upload_cloud(x, y, z)
    ctx.set_frames(poses)
    for f, im in enumerate(imgs):
        ctx.upload_image(f, im)
    got = ctx.colorize()
    ref = oracle.colorize(cam_struct(oracle, cd), oracle.default_cull_params(), x, y, z, poses, imgs, threads=8,
                          want_top=False)
    assert np.array_equal(got["rgb"], ref["rgb"]) and np.array_equal(got["has"], ref["has"])
    assert ref["has"].sum() > 50_000
    ctx.close()
