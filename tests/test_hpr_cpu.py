"""ViewCulling::hidden_points_removal (view_culling.cpp:266-334) in the oracle: the exact orientation predicate against
rational arithmetic, the exact quickhull against scipy's bundled qhull_r (the library family the reference links) on
random sets and on the committed goldens g3 (sparse: 418 candidates) and g3b (map density: 61 532 candidates), and the
cases qhull fails on.  No GPU."""
import os
from fractions import Fraction

import numpy as np
import pytest

from conftest import cam_struct

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CAM_KEYS = ["fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "k3", "image_width", "image_height", "cull_width",
            "cull_height"]
QHULL_TOLERANCE = 1e-9  # metres: qhull decides with round-off tolerances (DISTround ~ 1e-10 at |x| ~ 1.8e5), this file exactly


def load(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


def cam_of(mod, a):
    return cam_struct(mod, {k: (int(v) if k.endswith(("width", "height")) else float(v)) for k, v in zip(CAM_KEYS, a)})


def rational_orient(a, b, c, d):
    A = [[Fraction(float(p[k])) - Fraction(float(d[k])) for k in range(3)] for p in (a, b, c)]
    det = (A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0])
           + A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]))
    return (det > 0) - (det < 0)


def hull_depth(points, which):
    """Signed distance (negative inside) of points[which] to the hull of the others -- for the qhull tolerance clause."""
    from scipy.spatial import ConvexHull

    keep = np.ones(len(points), bool)
    keep[which] = False
    h = ConvexHull(points[keep])
    return float(np.max(h.equations[:, :3] @ points[which] + h.equations[:, 3]))


def compare_with_qhull(points, is_vertex_exact, vertices_qhull):
    """The exact vertex set equals qhull's except for points within QHULL_TOLERANCE of the hull of the rest."""
    q = np.zeros(len(points), bool)
    q[vertices_qhull] = True
    diff = np.nonzero(q != is_vertex_exact.astype(bool))[0]
    for i in diff:
        assert abs(hull_depth(points, i)) < QHULL_TOLERANCE, f"point {i}: exact {is_vertex_exact[i]} qhull {q[i]}"
    return len(diff)


def test_orientation_predicate_is_exact(oracle):
    rng = np.random.default_rng(1)
    zeros = 0
    for t in range(1500):
        a, b, c = rng.normal(size=(3, 3)) * rng.choice([1.0, 1e5, 1e-3])
        if t % 3 == 0:  # nearly / exactly coplanar
            w = rng.random(3)
            w /= w.sum()
            d = w[0] * a + w[1] * b + w[2] * c
            if t % 2 == 0:
                d = np.nextafter(d, d + rng.normal(size=3))
        elif t % 3 == 1:
            d = a.copy() if t % 2 else (a + b) / 2
        else:
            d = rng.normal(size=3)
        want = rational_orient(a, b, c, d)
        zeros += want == 0
        assert oracle.orient3d(a, b, c, d) == want
        assert oracle.orient3d(a, b, c, d, exact_only=True) == want
    assert zeros > 100  # the exactly degenerate inputs were really exercised


def test_quickhull_equals_qhull_on_random_sets(oracle):
    from scipy.spatial import ConvexHull

    rng = np.random.default_rng(2)
    for n in (4, 5, 8, 30, 200, 3000):
        for rep in range(12 if n < 1000 else 2):
            P = rng.normal(size=(n, 3))
            if rep % 3 == 1:
                P /= np.linalg.norm(P, axis=1, keepdims=True)  # every point a vertex
            if rep % 3 == 2:
                P[:, 2] *= 1e-3
            v, nv, st = oracle.convex_hull_vertices(P)
            assert nv == int(v.sum()) and st["zero"] == 0
            assert compare_with_qhull(P, v, ConvexHull(P).vertices) == 0


def test_flat_and_tiny_inputs_fail_like_qhull(oracle):
    """qh_new_qhull returns an error for fewer than dim + 1 points and for flat input; the reference then returns no
    visible point (view_culling.cpp:307-312)."""
    P = np.random.default_rng(3).normal(size=(10, 3))
    assert oracle.convex_hull_vertices(P[:3])[1] == -1
    Q = P.copy()
    Q[:, 2] = 0.0
    v, nv, _ = oracle.convex_hull_vertices(Q)
    assert nv == -1 and not v.any()


def test_duplicates_lowest_index_stands(oracle):
    rng = np.random.default_rng(4)
    P = rng.normal(size=(50, 3))
    P /= np.linalg.norm(P, axis=1, keepdims=True)
    D = np.concatenate([P, P[[3, 7, 7]]])
    v, nv, st = oracle.convex_hull_vertices(D)
    assert st["duplicates"] == 3 and nv == 50 and v[:50].all() and not v[50:].any()


def test_g3_sparse_golden(oracle):
    g = load("g3_hpr.npz")
    cam = cam_of(oracle, g["camera"])
    w2c, _ = oracle.pose_to_matrices(g["pose"])
    keep, st = oracle.hpr_frame(cam, w2c, g["x"], g["y"], g["z"])
    assert st["candidates"] == len(g["candidates"]) and st["zero"] == 0
    assert np.array_equal(np.nonzero(keep)[0], g["visible"])


def test_g3b_dense_golden(oracle):
    """At map density the hull is an occlusion cull: 61 532 candidates, 26 022 vertices (42 %)."""
    g = load("g3b_hpr_dense.npz")
    cam = cam_of(oracle, g["camera"])
    w2c, _ = oracle.pose_to_matrices(g["pose"])
    n = len(g["x"])
    keep, st = oracle.hpr_frame(cam, w2c, g["x"], g["y"], g["z"])
    want = np.unpackbits(g["visible_bits"])[:n].astype(bool)
    assert st["candidates"] == n and int(g["n_visible"]) == int(want.sum()) == 26022
    flipped = np.concatenate([oracle.hpr_flip(g["x"], g["y"], g["z"]), np.zeros((1, 3))])
    differing = compare_with_qhull(flipped, np.append(keep, 1), np.append(np.nonzero(want)[0], n))
    assert differing == 0, "today's qhull_r and the exact hull agree on every point of this fixture"
    assert st["zero"] == 0 and st["duplicates"] == 0


def test_numpy_twin_agrees(oracle):
    from oracle import np_oracle as npo
    from pointcloudprocessor_amd import synth

    cd = synth.camera_dict("cfg")
    x, y, z, _ = synth.make_cloud(150_000)
    poses, _ = synth.make_trajectory(4)
    cam = cam_struct(oracle, cd)
    for f in range(4):
        w2c, _ = npo.pose_to_matrices(poses[f])
        keep, st = oracle.hpr_frame(cam, w2c, x, y, z)
        assert np.array_equal(np.nonzero(keep)[0], npo.hpr_frame(cd, w2c, x, y, z)) and st["kept"] < st["candidates"]


def test_cull_mode_hpr_in_the_colour_path(oracle):
    """ORC_CULL_HPR through orc_cull_frame / orc_colorize: the hull's keep mask replaces the z-buffer rule."""
    from pointcloudprocessor_amd import synth

    cd = synth.camera_dict("tiny")
    x, y, z, _ = synth.make_cloud(60_000)
    poses, _ = synth.make_trajectory(3)
    imgs = [synth.make_image(f, cd["image_width"], cd["image_height"]) for f in range(3)]
    cam = cam_struct(oracle, cd)
    cp = oracle.default_cull_params()
    cp.cull_mode = oracle.CULL_HPR
    w2c, _ = oracle.pose_to_matrices(poses[1])
    keep, _, kept = oracle.cull_frame(cam, cp, w2c, x, y, z)
    hk, st = oracle.hpr_frame(cam, w2c, x, y, z)
    assert np.array_equal(keep, hk) and kept == st["kept"]
    cz = oracle.default_cull_params()
    a = oracle.colorize(cam, cp, x, y, z, poses, imgs)
    b = oracle.colorize(cam, cz, x, y, z, poses, imgs)
    assert a["has"].sum() > 0 and not np.array_equal(a["count"], b["count"])  # a different cull: different view lists
