"""Appendix B3 "faithful mode" of the oracle (the reference's own fp32 world round trip + 10 um kd-tree
match-back, PointCloudProcessor.cpp:554-592) and the hidden_points_removal candidate filter
(view_culling.cpp:276-288): C restatement vs the numpy twin vs the committed goldens g4b / g3, and the
measurement VERDICT r1 asked for -- how far the product's index-identity scoring is from the reference's
arithmetic.  No GPU needed."""
import numpy as np

from test_oracle_cpu import cam_from_array, load

KEYS = ("rgb", "has", "count", "top_score", "top_rgb", "top_frame")
STAT = ("samples", "unmatched", "self_missed", "cross_credits")


def identity_vs_faithful(ident, fa):
    """The numbers DESIGN.md quotes: relative finalScore difference over the slots both modes fill with the same
    keyframe, uint8 channels that differ, points whose top-5 membership / order differs."""
    same = (ident["top_frame"] == fa["top_frame"]) & (fa["top_frame"] >= 0)
    rel = np.abs(ident["top_score"][same].astype(np.float64) - fa["top_score"][same]) / fa["top_score"][same]
    member = (np.sort(ident["top_frame"], axis=1) != np.sort(fa["top_frame"], axis=1)).any(axis=1)
    order = (ident["top_frame"] != fa["top_frame"]).any(axis=1)
    chan = ident["rgb"] != fa["rgb"]
    return dict(max_rel_score=float(rel.max()) if rel.size else 0.0, score_bits_differ=int((ident["top_score"][same] != fa["top_score"][same]).sum()),
                scores=int(same.sum()), channels_differ=int(chan.sum()), channels=int(3 * (fa["has"] > 0).sum()),
                max_channel_delta=int(np.abs(ident["rgb"].astype(int) - fa["rgb"].astype(int)).max()),
                membership_differs=int(member.sum()), order_differs=int(order.sum()),
                count_differs=int((ident["count"] != fa["count"]).sum()))


def test_affine_inverse_matches_twin_and_inverts(oracle):
    from oracle import np_oracle as npo

    rng = np.random.default_rng(3)
    for _ in range(50):
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        R = npo.quat_to_rot(*q) * rng.uniform(0.98, 1.02)  # also un-normalised quaternions
        m = np.concatenate([R, rng.uniform(-20, 20, (3, 1))], axis=1).astype(np.float32)
        a = oracle.affine_inverse(m).reshape(3, 4)
        assert np.array_equal(a, npo.affine_inverse_f32(m))
        full = np.vstack([m.astype(np.float64), [0, 0, 0, 1]])
        assert np.allclose(np.vstack([a.astype(np.float64), [0, 0, 0, 1]]) @ full, np.eye(4), atol=2e-5)


def test_g4b_faithful_golden_and_roundtrip_mode(oracle):
    g4, gb = load("g4_colour.npz"), load("g4b_faithful.npz")
    cam, _ = cam_from_array(oracle, g4["camera"])
    cp = oracle.default_cull_params()
    imgs = list(g4["images"])
    fa = oracle.colorize_faithful(cam, cp, g4["x"], g4["y"], g4["z"], g4["poses"], imgs)
    for k in KEYS:
        assert np.array_equal(fa[k], gb[k]), k
    assert [fa["stats"][k] for k in STAT] == list(gb["stats"])
    # on separated points within +-8 m the kd-tree match is the point itself: the round-trip mode (no neighbour
    # search, what the GPU implements) is the faithful mode bit for bit
    assert fa["stats"]["unmatched"] == fa["stats"]["self_missed"] == fa["stats"]["cross_credits"] == 0
    cp.match_mode = oracle.MATCH_ROUNDTRIP
    rt = oracle.colorize(cam, cp, g4["x"], g4["y"], g4["z"], g4["poses"], imgs)
    for k in KEYS:
        assert np.array_equal(rt[k], fa[k]), k
    # identity mode against the reference's arithmetic: scores agree to fp32 rounding, the top-5 lists hold the same
    # keyframes in the same order, uint8 colours differ by at most one level
    cp.match_mode = oracle.MATCH_IDENTITY
    ident = oracle.colorize(cam, cp, g4["x"], g4["y"], g4["z"], g4["poses"], imgs)
    d = identity_vs_faithful(ident, fa)
    print("g4 identity vs faithful:", d)
    assert d["max_rel_score"] < 1e-6 and d["membership_differs"] == 0 and d["count_differs"] == 0
    assert d["max_channel_delta"] <= 1


def test_g4b_far_map_loses_samples_and_cross_credits(oracle):
    """150 m from the origin the fp32 ulp (15 um) exceeds the 10 um match radius: the reference loses samples; points
    4 um apart receive each other's samples.  C restatement == numpy twin == golden; the round-trip mode reproduces
    the losses (the self-match test) but, by construction, not the cross-credits."""
    g4, gb = load("g4_colour.npz"), load("g4b_faithful.npz")
    cam, _ = cam_from_array(oracle, g4["camera"])
    cp = oracle.default_cull_params()
    imgs = list(g4["images"])
    x, y, z, poses = gb["far_x"], gb["far_y"], gb["far_z"], gb["far_poses"]
    fa = oracle.colorize_faithful(cam, cp, x, y, z, poses, imgs)
    for k in KEYS:
        assert np.array_equal(fa[k], gb["far_" + k]), k
    assert [fa["stats"][k] for k in STAT] == list(gb["far_stats"])
    assert fa["stats"]["unmatched"] > 100 and fa["stats"]["cross_credits"] > 5
    cp.match_mode = oracle.MATCH_ROUNDTRIP
    rt = oracle.colorize(cam, cp, x, y, z, poses, imgs)
    # every difference between the two is a point that took part in a cross-credit
    differs = np.nonzero((rt["count"] != fa["count"]) | (rt["top_frame"] != fa["top_frame"]).any(axis=1))[0]
    assert 0 < len(differs) <= 2 * fa["stats"]["cross_credits"]
    # samples the reference loses are lost in round-trip mode too (identity mode keeps them)
    cp.match_mode = oracle.MATCH_IDENTITY
    ident = oracle.colorize(cam, cp, x, y, z, poses, imgs)
    assert ident["count"].sum() == fa["stats"]["samples"]
    assert rt["count"].sum() == fa["stats"]["samples"] - fa["stats"]["self_missed"]


def test_faithful_equals_roundtrip_on_baseline_config0(oracle):
    """configs[0] (100 k points x 1 keyframe) and a 100 k x 8 slice of configs[1]'s scene at 1920x1080: the scene of
    SURVEY 8(d) keeps points >= 50 um apart inside +-8 m, so faithful == round trip; identity differs only in score
    rounding (the figures go to DESIGN.md)."""
    from conftest import cam_struct
    from pointcloudprocessor_amd import synth

    cd = synth.camera_dict("cfg")
    cam = cam_struct(oracle, cd)
    x, y, z, _ = synth.make_cloud(100_000)
    poses, _ = synth.make_trajectory(8)
    imgs = [synth.make_image(f, cd["image_width"], cd["image_height"]) for f in range(8)]
    for nf in (1, 8):
        cp = oracle.default_cull_params()
        fa = oracle.colorize_faithful(cam, cp, x, y, z, poses[:nf], imgs[:nf], threads=0)
        assert fa["stats"]["unmatched"] == fa["stats"]["self_missed"] == fa["stats"]["cross_credits"] == 0
        cp.match_mode = oracle.MATCH_ROUNDTRIP
        rt = oracle.colorize(cam, cp, x, y, z, poses[:nf], imgs[:nf], threads=0)
        for k in KEYS:
            assert np.array_equal(rt[k], fa[k]), (nf, k)
        cp.match_mode = oracle.MATCH_IDENTITY
        ident = oracle.colorize(cam, cp, x, y, z, poses[:nf], imgs[:nf], threads=0)
        d = identity_vs_faithful(ident, fa)
        print(f"100k x {nf} @1920x1080 identity vs faithful:", d, fa["stats"])
        assert d["max_rel_score"] < 1e-6 and d["count_differs"] == 0 and d["max_channel_delta"] <= 1
        assert d["membership_differs"] <= 2


def test_g3_hpr_candidates_and_qhull(oracle):
    """The candidate filter in front of qhull (PCP_CULL_HPR_CANDIDATES): C restatement == numpy twin == golden;
    everything qhull keeps is a candidate, and the golden records how many candidates it drops."""
    from oracle import np_oracle as npo

    g = load("g3_hpr.npz")
    cam, cd = cam_from_array(oracle, g["camera"])
    cp = oracle.default_cull_params()
    cp.cull_mode = oracle.CULL_HPR_CANDIDATES
    w2c, _ = oracle.pose_to_matrices(g["pose"])
    keep, dmap, kept = oracle.cull_frame(cam, cp, w2c, g["x"], g["y"], g["z"])
    cand = np.nonzero(keep)[0]
    assert np.array_equal(cand, g["candidates"]) and kept == len(cand)
    mask, _ = npo.hpr_candidates(cd, w2c.reshape(3, 4), g["x"], g["y"], g["z"])
    assert np.array_equal(np.nonzero(mask)[0], cand)
    assert np.all(dmap == np.finfo(np.float32).max)  # no depth buffer in this mode
    p = oracle.project_frame(cam, cp, w2c, g["x"], g["y"], g["z"])
    assert np.array_equal(p["cell"] == -2, keep.astype(bool)) and np.all((p["cell"] == -2) | (p["cell"] == -1))
    assert set(g["visible"]) <= set(cand)
    dropped = len(cand) - len(g["visible"])
    print(f"g3: {len(cand)} candidates, qhull (flip radius 90000) keeps {len(g['visible'])}, drops {dropped}")
    assert dropped <= max(2, len(cand) // 100)
    # the z-buffer (the product default) is the stricter cull
    assert set(g["zbuffer_keep"]) - set(cand) == set() or len(set(g["zbuffer_keep"]) - set(cand)) < 20
