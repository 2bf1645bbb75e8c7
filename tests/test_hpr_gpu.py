"""PCP_CULL_HPR on the device (csrc/pcp_hpr.hip) = ViewCulling::hidden_points_removal (view_culling.cpp:266-334): the
hull vertices among the flipped candidates, through the C ABI, against the committed goldens (scipy's qhull_r) and the
oracle's exact quickhull.  Both sides decide exactly, so the keep masks must be EQUAL, not close; `unresolved` (points
the exact path could not certify: exactly degenerate input only) must be 0 on every scene here."""
import os

import numpy as np
import pytest

from conftest import cam_struct

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CAM_KEYS = ["fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "k3", "image_width", "image_height", "cull_width",
            "cull_height"]


def load(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


def cam_of(mod, a):
    return cam_struct(mod, {k: (int(v) if k.endswith(("width", "height")) else float(v)) for k, v in zip(CAM_KEYS, a)})


def hull_ctx(factory, capi, cam, x, y, z, poses):
    cull = capi.default_cull_params()
    cull.cull_mode = capi.CULL_HPR
    ctx = factory()
    ctx.set_camera(cam, cull)
    ctx.upload_cloud(x, y, z)
    ctx.set_frames(np.asarray(poses, np.float64).reshape(-1, 7))
    return ctx


def test_g3_sparse_golden(gpu_ctx_factory):
    from pointcloudprocessor_amd import capi

    g = load("g3_hpr.npz")
    ctx = hull_ctx(gpu_ctx_factory, capi, cam_of(capi, g["camera"]), g["x"], g["y"], g["z"], g["pose"])
    keep, _, kept = ctx.cull_frame(0)
    st = ctx.hpr_stats()
    assert np.array_equal(np.nonzero(keep)[0], g["visible"]) and kept == len(g["visible"])
    assert st["candidates"] == len(g["candidates"]) and st["unresolved"] == 0
    assert st["visible"] + st["hidden"] == st["candidates"]
    ctx.close()


def test_g3b_dense_golden(gpu_ctx_factory, oracle):
    """61 532 candidates at map density: qhull keeps 26 022 (42 %) -- the hull is an occlusion cull, not a no-op."""
    from pointcloudprocessor_amd import capi

    g = load("g3b_hpr_dense.npz")
    n = len(g["x"])
    ctx = hull_ctx(gpu_ctx_factory, capi, cam_of(capi, g["camera"]), g["x"], g["y"], g["z"], g["pose"])
    keep, _, kept = ctx.cull_frame(0)
    st = ctx.hpr_stats()
    want = np.unpackbits(g["visible_bits"])[:n]
    assert np.array_equal(keep, want), f"{int((keep != want).sum())} points differ from qhull's vertex set"
    assert kept == int(g["n_visible"]) == 26022 and st["candidates"] == n and st["unresolved"] == 0
    w2c, _ = oracle.pose_to_matrices(g["pose"])
    okeep, _ = oracle.hpr_frame(cam_of(oracle, g["camera"]), w2c, g["x"], g["y"], g["z"])
    assert np.array_equal(keep, okeep)
    ctx.close()


@pytest.mark.parametrize("camname", ["cfg", "ref"])
def test_scene_equals_oracle(gpu_ctx_factory, oracle, camname):
    """300 k points, 6 keyframes, both cameras: every keep mask equals the exact hull of the oracle."""
    from pointcloudprocessor_amd import capi, synth

    cd = synth.camera_dict(camname)
    x, y, z, _ = synth.make_cloud(300_000)
    poses, _ = synth.make_trajectory(6)
    ctx = hull_ctx(gpu_ctx_factory, capi, cam_struct(capi, cd), x, y, z, poses)
    ocam = cam_struct(oracle, cd)
    dropped = 0
    for f in range(6):
        keep, _, kept = ctx.cull_frame(f)
        st = ctx.hpr_stats()
        w2c, _ = oracle.pose_to_matrices(poses[f])
        okeep, ost = oracle.hpr_frame(ocam, w2c, x, y, z)
        assert np.array_equal(keep, okeep), (camname, f, int((keep != okeep).sum()))
        assert st["unresolved"] == 0 and st["candidates"] == ost["candidates"] and kept == ost["kept"]
        dropped += ost["candidates"] - ost["kept"]
    assert dropped > 0
    ctx.close()


def test_exact_path_alone_decides_the_same(gpu_ctx_factory, oracle, monkeypatch):
    """PCP_HPR_FORCE_EXACT=1 sends every candidate to k_hpr_exact (certificates checked with the exact predicate):
    same keep masks."""
    from pointcloudprocessor_amd import capi, synth

    cd = synth.camera_dict("cfg")
    x, y, z, _ = synth.make_cloud(120_000)
    poses, _ = synth.make_trajectory(3)
    ctx = hull_ctx(gpu_ctx_factory, capi, cam_struct(capi, cd), x, y, z, poses)
    ocam = cam_struct(oracle, cd)
    for f in range(3):
        monkeypatch.setenv("PCP_HPR_FORCE_EXACT", "1")
        keep, _, _ = ctx.cull_frame(f)
        st = ctx.hpr_stats()
        monkeypatch.delenv("PCP_HPR_FORCE_EXACT")
        w2c, _ = oracle.pose_to_matrices(poses[f])
        okeep, ost = oracle.hpr_frame(ocam, w2c, x, y, z)
        assert st["exact_path"] == st["candidates"] == ost["candidates"] and st["unresolved"] == 0
        assert np.array_equal(keep, okeep)
    ctx.close()


def test_near_degenerate_points(gpu_ctx_factory, oracle):
    """Points placed within 1e-12 .. 1e-7 m of facets of the hull of the flipped set (on either side): the
    floating-point certificates cannot settle the closest ones, the exact path must, and the result is the oracle's."""
    from pointcloudprocessor_amd import capi, synth

    cd = synth.camera_dict("cfg")
    rng = np.random.default_rng(9)
    x, y, z, _ = synth.make_cloud(40_000)
    poses, _ = synth.make_trajectory(3)
    ocam = cam_struct(oracle, cd)
    w2c, _ = oracle.pose_to_matrices(poses[1])
    p = oracle.project_frame(ocam, _hpr_candidates_params(oracle), w2c, x, y, z)
    cand = np.nonzero(p["cell"] != -1)[0]
    cx, cy, cz = p["xc"][cand], p["yc"][cand], p["zc"][cand]
    # camera-frame cloud with the identity pose (the transform of the identity returns the coordinates unchanged);
    # add points interpolated between neighbouring candidates: their flipped images land next to hull facets / edges
    k = len(cand)
    a, b = rng.integers(0, k, 4000), rng.integers(0, k, 4000)
    t = rng.random(4000).astype(np.float32)
    near = np.abs(cx[a] / cz[a] - cx[b] / cz[b]) + np.abs(cy[a] / cz[a] - cy[b] / cz[b]) < 0.02
    a, b, t = a[near], b[near], t[near]
    ex = np.concatenate([cx, cx[a] * (1 - t) + cx[b] * t])
    ey = np.concatenate([cy, cy[a] * (1 - t) + cy[b] * t])
    ez = np.concatenate([cz, cz[a] * (1 - t) + cz[b] * t])
    ident = np.array([0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0])
    ctx = hull_ctx(gpu_ctx_factory, capi, cam_struct(capi, cd), ex, ey, ez, ident)
    keep, _, _ = ctx.cull_frame(0)
    st = ctx.hpr_stats()
    w2i, _ = oracle.pose_to_matrices(ident)
    okeep, ost = oracle.hpr_frame(ocam, w2i, ex, ey, ez)
    assert np.array_equal(keep, okeep), int((keep != okeep).sum())
    assert st["unresolved"] == 0 and ost["zero"] == 0
    ctx.close()


def _hpr_candidates_params(oracle):
    cp = oracle.default_cull_params()
    cp.cull_mode = oracle.CULL_HPR_CANDIDATES
    return cp


def test_duplicates_and_tiny_candidate_sets(gpu_ctx_factory, oracle):
    from pointcloudprocessor_amd import capi, synth

    cd = synth.camera_dict("cfg")
    x, y, z, _ = synth.make_cloud(30_000)
    poses, _ = synth.make_trajectory(2)
    # exact duplicates of 500 points, appended: the lowest index of a group stands for it
    dup = np.random.default_rng(5).choice(len(x), 500, replace=False)
    xd, yd, zd = np.concatenate([x, x[dup]]), np.concatenate([y, y[dup]]), np.concatenate([z, z[dup]])
    ctx = hull_ctx(gpu_ctx_factory, capi, cam_struct(capi, cd), xd, yd, zd, poses)
    ocam = cam_struct(oracle, cd)
    for f in range(2):
        keep, _, _ = ctx.cull_frame(f)
        w2c, _ = oracle.pose_to_matrices(poses[f])
        okeep, ost = oracle.hpr_frame(ocam, w2c, xd, yd, zd)
        assert np.array_equal(keep, okeep) and not keep[len(x):].any()
        assert ctx.hpr_stats()["unresolved"] == 0
    ctx.close()
    # fewer than three candidates: qhull fails, the reference returns nothing (view_culling.cpp:307-312); three: all
    ident = np.array([0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0])
    pts = np.array([[0.1, 0.0, 2.0], [-0.1, 0.05, 2.5], [0.0, -0.1, 3.0], [0.0, 0.0, -1.0]], np.float32)
    for m, want in ((2, 0), (3, 3)):
        sel = np.concatenate([pts[:m], pts[3:]])  # the last point is behind the camera: never a candidate
        c2 = hull_ctx(gpu_ctx_factory, capi, cam_struct(capi, cd), sel[:, 0], sel[:, 1], sel[:, 2], ident)
        keep, _, kept = c2.cull_frame(0)
        w2i, _ = oracle.pose_to_matrices(ident)
        okeep, _ = oracle.hpr_frame(ocam, w2i, sel[:, 0], sel[:, 1], sel[:, 2])
        assert kept == want and np.array_equal(keep, okeep)
        c2.close()


def test_frame_visible_and_whole_run_in_hull_mode(gpu_ctx_factory, oracle, small_scene):
    """The single-keyframe record call and the whole colourisation with cull_mode = PCP_CULL_HPR against the oracle's
    ORC_CULL_HPR: visible lists, colours, masks, top-5 lists."""
    from pointcloudprocessor_amd import capi

    s = small_scene
    cull = capi.default_cull_params()
    cull.cull_mode = capi.CULL_HPR
    ctx = gpu_ctx_factory()
    ctx.set_camera(cam_struct(capi, s["cam"]), cull)
    ctx.upload_cloud(s["x"], s["y"], s["z"])
    ctx.set_frames(s["poses"])
    for f, (im, mk) in enumerate(zip(s["images"], s["masks"])):
        ctx.upload_image(f, im)
        ctx.upload_mask(f, mk)
    ocam = cam_struct(oracle, s["cam"])
    ocp = oracle.default_cull_params()
    ocp.cull_mode = oracle.CULL_HPR
    for f in range(len(s["poses"])):
        got = ctx.frame_visible(f)
        ref = oracle.frame_visible(ocam, ocp, s["poses"][f], s["x"], s["y"], s["z"], s["images"][f], s["masks"][f])
        assert np.array_equal(got["index"], ref["index"]) and np.array_equal(got["rgb"], ref["rgb"])
        assert np.array_equal(got["mask"], ref["mask"]) and np.array_equal(got["xyz_cam"], ref["xyz_cam"])
    ref = oracle.colorize(ocam, ocp, s["x"], s["y"], s["z"], s["poses"], s["images"])
    col = ctx.colorize()
    assert np.array_equal(col["has"], ref["has"]) and np.array_equal(col["rgb"], ref["rgb"])
    ctx.depth_pass()
    ctx.colour_reset()
    ctx.colour_pass(0, 2)
    ctx.colour_pass(2, len(s["poses"]))
    r = ctx.colour_finalise(want_top=True)
    for k in ("rgb", "has", "count", "top_score", "top_rgb", "top_frame"):
        assert np.array_equal(r[k], ref[k]), k
    # not the z-buffer's answer: the two culls keep different view lists
    zref = oracle.colorize(ocam, oracle.default_cull_params(), s["x"], s["y"], s["z"], s["poses"], s["images"])
    assert not np.array_equal(zref["count"], ref["count"])
    ctx.close()


def test_hull_mode_over_index_shards(gpu_ctx_factory, small_scene):
    """A keyframe's hull is taken over every candidate of the map, so an index shard cannot decide its own points: in
    PCP_CULL_HPR a shard (PCP_DEPTH_BATCHED) fails loudly until the verdicts of a whole-map context have been handed to
    it (pcp_hull_flags_import); with them its colours and per-keyframe records equal the whole-map run's, slice by slice."""
    from pointcloudprocessor_amd import capi

    s = small_scene
    F, n = len(s["poses"]), len(s["x"])
    cull = capi.default_cull_params()
    cull.cull_mode = capi.CULL_HPR

    def make(lo, hi, shard):
        ctx = gpu_ctx_factory()
        if shard:
            ctx.set_depth_source(True)
        ctx.set_camera(cam_struct(capi, s["cam"]), cull)
        ctx.upload_cloud(s["x"][lo:hi], s["y"][lo:hi], s["z"][lo:hi])
        ctx.set_frames(s["poses"])
        for f, (im, mk) in enumerate(zip(s["images"], s["masks"])):
            ctx.upload_image(f, im)
            ctx.upload_mask(f, mk)
        return ctx

    full = make(0, n, False)
    ref = full.colorize()
    flags = [full.cull_frame(f)[0] for f in range(F)]
    vis = [full.frame_visible(f) for f in range(F)]
    bounds = [0, n // 3, n // 3 + n // 4, n]
    rgb, has = [], []
    idx = [[] for _ in range(F)]
    vrgb = [[] for _ in range(F)]
    for r in range(3):
        lo, hi = bounds[r], bounds[r + 1]
        sh = make(lo, hi, True)
        sh.depth_pass()
        with pytest.raises(capi.PcpError):
            sh.colorize_from_depth()
        with pytest.raises(capi.PcpError):
            sh.cull_frame(0)
        for f in range(F):
            sh.hull_flags_import(f, flags[f][lo:hi])
        col = sh.colorize_from_depth()
        rgb.append(col["rgb"])
        has.append(col["has"])
        for f in range(F):
            keep, _, kept = sh.cull_frame(f)
            assert np.array_equal(keep, flags[f][lo:hi]) and kept == int(flags[f][lo:hi].sum())
            v = sh.frame_visible(f)
            idx[f].append(v["index"] + lo)
            vrgb[f].append(v["rgb"])
        sh.close()
    assert np.array_equal(np.concatenate(rgb), ref["rgb"]) and np.array_equal(np.concatenate(has), ref["has"])
    assert ref["has"].sum() > 100
    for f in range(F):
        assert np.array_equal(np.concatenate(idx[f]), vis[f]["index"])
        assert np.array_equal(np.concatenate(vrgb[f]), vis[f]["rgb"])
    full.close()


def test_whole_run_bits_in_chunks_of_keyframes(gpu_ctx_factory, oracle):
    """The whole-run hull bits over more than one 32-keyframe plane: one pcp_depth_pass over every keyframe (whole planes are
    zeroed by a memset) against the same pass in ranges that cut planes in two (single bits cleared by a kernel), and both
    against the oracle's colours."""
    from pointcloudprocessor_amd import capi, synth

    cd = synth.camera_dict("tiny")
    F = 40
    x, y, z, _ = synth.make_cloud(30000)
    poses, _ = synth.make_trajectory(F)
    imgs = [synth.make_image(f, cd["image_width"], cd["image_height"]) for f in range(F)]
    cull = capi.default_cull_params()
    cull.cull_mode = capi.CULL_HPR
    ctx = gpu_ctx_factory()
    ctx.set_camera(cam_struct(capi, cd), cull)
    ctx.upload_cloud(x, y, z)
    ctx.set_frames(poses)
    for f, im in enumerate(imgs):
        ctx.upload_image(f, im)
    whole = ctx.colorize()
    results = []
    for cuts in ((0, F), (0, 20, F), (0, 7, 33, 39, F)):
        # (the bits of the round before are still there: every round must clear what it does not set)
        for a, b in zip(cuts[:-1], cuts[1:]):
            ctx.depth_pass(a, b)
        ctx.colour_reset()
        ctx.colour_pass(0, F)
        results.append(ctx.colour_finalise(want_top=True))
    for r in results:
        assert np.array_equal(r["rgb"], whole["rgb"]) and np.array_equal(r["has"], whole["has"])
        for k in ("count", "top_score", "top_frame"):
            assert np.array_equal(r[k], results[0][k]), k
    ocam = cam_struct(oracle, cd)
    ocp = oracle.default_cull_params()
    ocp.cull_mode = oracle.CULL_HPR
    ref = oracle.colorize(ocam, ocp, x, y, z, poses, imgs)
    assert np.array_equal(whole["rgb"], ref["rgb"]) and np.array_equal(whole["has"], ref["has"])
    ctx.close()


@pytest.mark.parametrize("lanes", ["1", "3", "8"])
def test_whole_run_bits_do_not_depend_on_the_keyframes_in_flight(gpu_ctx_factory, oracle, monkeypatch, lanes):
    """The whole-run hull pass keeps several keyframes in flight on lanes of their own (PCP_HPR_LANES, default 4): with one,
    three (ranges shorter than the lanes, a count that does not divide) and eight lanes the colours are the oracle's, the
    per-keyframe verdicts read back from the pass's bits equal pcp_cull_frame on a context that never ran the pass, and the
    16-lane search switched off (PCP_HPR_TILT=0) changes nothing."""
    from pointcloudprocessor_amd import capi, synth

    cd = synth.camera_dict("tiny")
    F = 11
    x, y, z, _ = synth.make_cloud(40000, seed=11)
    poses, _ = synth.make_trajectory(F)
    imgs = [synth.make_image(f, cd["image_width"], cd["image_height"]) for f in range(F)]
    cull = capi.default_cull_params()
    cull.cull_mode = capi.CULL_HPR
    monkeypatch.setenv("PCP_HPR_LANES", lanes)

    def ctx_of():
        c = gpu_ctx_factory()
        c.set_camera(cam_struct(capi, cd), cull)
        c.upload_cloud(x, y, z)
        c.set_frames(poses)
        for f, im in enumerate(imgs):
            c.upload_image(f, im)
        return c

    ctx = ctx_of()
    got = ctx.colorize()
    ocam = cam_struct(oracle, cd)
    ocp = oracle.default_cull_params()
    ocp.cull_mode = oracle.CULL_HPR
    ref = oracle.colorize(ocam, ocp, x, y, z, poses, imgs)
    assert np.array_equal(got["rgb"], ref["rgb"]) and np.array_equal(got["has"], ref["has"])
    # ranges shorter than the lanes, and the verdicts of the pass read back per keyframe
    ctx.depth_pass(0, 2)
    ctx.depth_pass(2, F)
    fresh = ctx_of()  # never ran the pass: pcp_cull_frame takes each hull itself
    for f in (0, 1, 5, F - 1):
        a, b = ctx.cull_frame(f), fresh.cull_frame(f)
        assert np.array_equal(a[0], b[0]) and a[2] == b[2], f
    monkeypatch.setenv("PCP_HPR_TILT", "0")
    again = ctx.colorize()
    assert np.array_equal(again["rgb"], got["rgb"]) and np.array_equal(again["has"], got["has"])
    ctx.close()
    fresh.close()
