"""BASELINE.json's smaller configurations as parity cases (bench.py measures configs[2]):

  configs[0]  100 k points x 1 keyframe, the reference's own camera (4096x3000, PointCloudProcessor.cpp:57-60), no MLS
  configs[1]  1 M points x 32 keyframes @1920x1080, no MLS

Every per-point output of the C ABI against the oracle on the same seeded inputs: cells, pixels, ranges, depth
maps and keep masks bit-exact, top-5 lists and colours bit-exact.
"""
import numpy as np
import pytest

from conftest import cam_struct

pytestmark = pytest.mark.gpu


def _scene(n, frames, camera):
    from pointcloudprocessor_amd import synth

    cd = synth.camera_dict(camera)
    x, y, z, _ = synth.make_cloud(n)
    poses, _ = synth.make_trajectory(frames)
    imgs = [synth.make_image(f, cd["image_width"], cd["image_height"]) for f in range(frames)]
    return cd, x, y, z, poses, imgs


def test_config0_100k_points_one_keyframe_reference_camera(gpu_ctx_factory, oracle):
    from pointcloudprocessor_amd import capi

    cd, x, y, z, poses, imgs = _scene(100_000, 1, "ref")
    ocam, ocp = cam_struct(oracle, cd), oracle.default_cull_params()
    ctx = gpu_ctx_factory()
    ctx.set_camera(cam_struct(capi, cd))
    ctx.upload_cloud(x, y, z)
    ctx.set_frames(poses)
    ctx.upload_image(0, imgs[0])
    w2c, _ = oracle.pose_to_matrices(poses[0])
    p_ref = oracle.project_frame(ocam, ocp, w2c, x, y, z)
    p_got = ctx.project_frame(0)
    for k in ("cell", "pixel"):
        assert np.array_equal(p_got[k], p_ref[k]), k
    cand = p_ref["cell"] != -1
    assert np.array_equal(p_got["range"][cand].view(np.uint32), p_ref["range"][cand].view(np.uint32))
    keep_r, dmap_r, _ = oracle.cull_frame(ocam, ocp, w2c, x, y, z)
    keep_g, dmap_g, _ = ctx.cull_frame(0)
    assert np.array_equal(keep_g, keep_r)
    assert np.array_equal(dmap_g.view(np.uint32), dmap_r.view(np.uint32))
    ref = oracle.colorize(ocam, ocp, x, y, z, poses, imgs, threads=4, want_top=True)
    ctx.depth_pass()
    ctx.colour_reset()
    ctx.colour_pass()
    got = ctx.colour_finalise(want_top=True)
    for k in ("count", "top_frame", "top_rgb", "top_score", "rgb", "has"):
        assert np.array_equal(got[k], ref[k]), k
    assert ref["has"].sum() > 1000
    ctx.close()


def test_config1_1m_points_32_keyframes(gpu_ctx_factory, oracle):
    from pointcloudprocessor_amd import capi

    cd, x, y, z, poses, imgs = _scene(1_000_000, 32, "cfg")
    ocam, ocp = cam_struct(oracle, cd), oracle.default_cull_params()
    ctx = gpu_ctx_factory()
    ctx.set_camera(cam_struct(capi, cd))
    ctx.upload_cloud(x, y, z)
    ctx.set_frames(poses)
    for f, im in enumerate(imgs):
        ctx.upload_image(f, im)
    ref = oracle.colorize(ocam, ocp, x, y, z, poses, imgs, threads=8, want_top=True)
    ctx.depth_pass()
    for f in (0, 7, 19, 31):
        w2c, _ = oracle.pose_to_matrices(poses[f])
        _, dmap_r, _ = oracle.cull_frame(ocam, ocp, w2c, x, y, z)
        assert np.array_equal(ctx.download_depth_map(f).view(np.uint32), dmap_r.view(np.uint32)), f
    ctx.colour_reset()
    ctx.colour_pass()
    got = ctx.colour_finalise(want_top=True)
    for k in ("count", "top_frame", "top_rgb", "top_score", "rgb", "has"):
        assert np.array_equal(got[k], ref[k]), k
    one_shot = ctx.colorize()
    assert np.array_equal(one_shot["rgb"], ref["rgb"]) and np.array_equal(one_shot["has"], ref["has"])
    assert ref["has"].sum() > 100_000
    ctx.close()


def test_many_keyframes_multi_word_masks_and_unaligned_chunks(gpu_ctx_factory, oracle):
    """300 keyframes = 10 tile-mask words with a ragged last word; depth and colour passes cut at boundaries that
    are not multiples of 32 must give the one-shot result, which must equal the oracle's."""
    from pointcloudprocessor_amd import capi, synth

    cd = synth.camera_dict("tiny")
    n, F = 60_000, 300
    x, y, z, _ = synth.make_cloud(n)
    poses, _ = synth.make_trajectory(F)
    imgs = [synth.make_image(f, cd["image_width"], cd["image_height"]) for f in range(F)]
    ocam, ocp = cam_struct(oracle, cd), oracle.default_cull_params()
    ref = oracle.colorize(ocam, ocp, x, y, z, poses, imgs, threads=8, want_top=True)
    ctx = gpu_ctx_factory()
    ctx.set_camera(cam_struct(capi, cd))
    ctx.upload_cloud(x, y, z)
    ctx.set_frames(poses)
    for f, im in enumerate(imgs):
        ctx.upload_image(f, im)
    one = ctx.colorize()
    assert np.array_equal(one["rgb"], ref["rgb"]) and np.array_equal(one["has"], ref["has"])
    cuts = [0, 17, 64, 65, 150, 255, 256, 299, 300]
    for a, b in zip(cuts[:-1], cuts[1:]):
        ctx.depth_pass(a, b)
    for f in (0, 16, 17, 64, 149, 150, 299):
        w2c, _ = oracle.pose_to_matrices(poses[f])
        _, dmap_r, _ = oracle.cull_frame(ocam, ocp, w2c, x, y, z)
        assert np.array_equal(ctx.download_depth_map(f).view(np.uint32), dmap_r.view(np.uint32)), f
    ctx.colour_reset()
    for a, b in ((0, 33), (33, 34), (34, 200), (200, 300)):
        ctx.colour_pass(a, b)
    got = ctx.colour_finalise(want_top=True)
    for k in ("count", "top_frame", "top_rgb", "top_score", "rgb", "has"):
        assert np.array_equal(got[k], ref[k]), k
    assert ref["has"].sum() > 5000
    ctx.close()


def test_config4_shape_2048_keyframes_with_masks(gpu_ctx_factory, oracle):
    """configs[4]'s keyframe count (2048 = 64 tile-mask words) and its segmentation masks on one rank's share of the
    work, shrunk in points only: colours against the oracle, masked per-keyframe dumps for keyframes in the first,
    a middle and the last word."""
    from pointcloudprocessor_amd import capi, synth

    cd = synth.camera_dict("tiny")
    n, F = 40_000, 2048
    W, H = cd["image_width"], cd["image_height"]
    x, y, z, _ = synth.make_cloud(n)
    poses, _ = synth.make_trajectory(F)
    base_imgs = [synth.make_image(f, W, H) for f in range(8)]
    base_masks = [synth.make_mask(f, W, H) for f in range(8)]
    imgs = [base_imgs[f % 8] for f in range(F)]
    ocam, ocp = cam_struct(oracle, cd), oracle.default_cull_params()
    ctx = gpu_ctx_factory()
    ctx.set_camera(cam_struct(capi, cd))
    ctx.upload_cloud(x, y, z)
    ctx.set_frames(poses)
    for f in range(F):
        ctx.upload_image(f, imgs[f])
        ctx.upload_mask(f, base_masks[f % 8])
    got = ctx.colorize()
    ref = oracle.colorize(ocam, ocp, x, y, z, poses, imgs, threads=8, want_top=False)
    assert np.array_equal(got["rgb"], ref["rgb"]) and np.array_equal(got["has"], ref["has"])
    assert ref["has"].sum() > 5000
    for f in (3, 1000, 2047):
        r = oracle.frame_visible(ocam, ocp, poses[f], x, y, z, imgs[f], base_masks[f % 8])
        g = ctx.frame_visible(f)
        assert g["count"] == len(r["index"])
        for k in ("index", "rgb", "mask", "xyz_cam", "xyz_world"):
            assert np.array_equal(g[k], r[k]), (f, k)
    ctx.close()
