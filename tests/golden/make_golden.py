#!/usr/bin/env python3
"""Generates the committed golden vectors under tests/golden/ with the numpy twin
(oracle/np_oracle.py) and, for the HPR variant, scipy's bundled qhull_r.

The reference ships no fixtures and cannot be built or run in this image
(SURVEY.md 8c), so these vectors do NOT come from the reference: they pin the
restatement against itself across languages (numpy here, C in
oracle/pcp_oracle.c, HIP in csrc/) and across time.  Parity stays "unpinned"
in the sense of the task statement; see DESIGN.md.

    python tests/golden/make_golden.py            # rewrites every .npz file
    python tests/golden/make_golden.py g7 g8 g9   # only the named ones (zip timestamps make rewrites non-identical)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import np_oracle as npo  # noqa: E402
from pointcloudprocessor_amd import synth  # noqa: E402


def nano_camera():
    d = dict(fx=188.2083, fy=188.2083, cx=80.0, cy=45.0)
    d.update(synth.REF_D)
    d.update(image_width=160, image_height=90, cull_width=160, cull_height=90)
    return d


ONLY = set(sys.argv[1:])


def save(name, **arrays):
    if ONLY and name.split("_")[0] not in ONLY:
        return
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def cam_array(cam):
    keys = ["fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "k3", "image_width", "image_height", "cull_width",
            "cull_height"]
    return np.array([cam[k] for k in keys], np.float64)


def main():
    rng = np.random.default_rng(20241008)
    poses, _ = synth.make_trajectory(6)

    # g1: projection with the REFERENCE constants (4096x3000), points in front of one pose
    cam = synth.camera_dict("ref")
    x, y, z, _ = synth.make_cloud(3000, seed=11)
    w2c, c2w = npo.pose_to_matrices(poses[0])
    p = npo.project_frame(cam, w2c, x, y, z)
    T = np.eye(4)
    T[:3, :3] = npo.quat_to_rot(0.9998, 0.01, -0.012, 0.008)
    T[:3, 3] = [0.02, -0.015, 0.03]
    w2c_T, c2w_T = npo.pose_to_matrices(poses[1], T)
    save("g1_projection.npz", camera=cam_array(cam), pose=poses[0], x=x, y=y, z=z, w2c=w2c, c2w=c2w, xc=p["xc"],
         yc=p["yc"], zc=p["zc"], u=p["u"], v=p["v"], cell=p["cell"], pixel=p["pixel"], range=p["range"],
         pose_T=poses[1], T_opt=T, w2c_T=w2c_T, c2w_T=c2w_T)

    # g2: z-buffer cull with real occluders, nano camera
    cam = nano_camera()
    x, y, z, _ = synth.make_cloud(20000, seed=12)
    keeps, dmaps = [], []
    for f in range(3):
        w2c, _ = npo.pose_to_matrices(poses[f])
        keep, dmap, _ = npo.cull_frame(cam, w2c, x, y, z)
        keeps.append(keep)
        dmaps.append(dmap)
    save("g2_zbuffer.npz", camera=cam_array(cam), poses=poses[:3], x=x, y=y, z=z, keep=np.array(keeps),
         depth=np.array(dmaps))

    # g3: HPR (the cull that is ACTIVE in the reference) on the same points, frame 0
    w2c, _ = npo.pose_to_matrices(poses[0])
    hpr = npo.hpr_frame(cam, w2c, x, y, z)
    cand, _ = npo.hpr_candidates(cam, w2c, x, y, z)  # the filter in front of qhull (view_culling.cpp:276-288)
    save("g3_hpr.npz", camera=cam_array(cam), pose=poses[0], x=x, y=y, z=z, visible=hpr.astype(np.int32),
         zbuffer_keep=np.nonzero(keeps[0])[0].astype(np.int32), candidates=np.nonzero(cand)[0].astype(np.int32))

    # g3b: the same cull at map density -- the candidates of keyframe 1 of the 2 M-point scene at the 1920x1080 camera
    # (61 532 of them; qhull keeps 42 %).  The fixture holds the candidates' CAMERA coordinates as a cloud of their own
    # with the identity pose: the transform of the identity (x * 1 + (y * 0 + (z * 0 + 0))) returns every coordinate
    # unchanged, so candidates, flipped points and hull are those of the 2 M-point scene without shipping it.
    if not ONLY or "g3b" in ONLY:
        camd = synth.camera_dict("cfg")
        xs, ys, zs, _ = synth.make_cloud(2_000_000)
        poses8, _ = synth.make_trajectory(8)
        w2c_d, _ = npo.pose_to_matrices(poses8[1])
        cand_d, pd = npo.hpr_candidates(camd, w2c_d, xs, ys, zs)
        ci = np.nonzero(cand_d)[0]
        vis_full = npo.hpr_frame(camd, w2c_d, xs, ys, zs)
        cx, cy, cz = pd["xc"][ci], pd["yc"][ci], pd["zc"][ci]
        ident = np.array([0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0])
        w2c_i, _ = npo.pose_to_matrices(ident)
        vis_i = npo.hpr_frame(camd, w2c_i, cx, cy, cz)
        assert np.array_equal(ci[vis_i], vis_full), "the candidate-only cloud must reproduce the scene's hull"
        vis_mask = np.zeros(len(ci), bool)
        vis_mask[vis_i] = True
        save("g3b_hpr_dense.npz", camera=cam_array(camd), pose=ident, x=cx, y=cy, z=cz,
             visible_bits=np.packbits(vis_mask), n_visible=np.int64(len(vis_i)))

    # g4: 6-keyframe colour run, procedural images
    imgs = [synth.make_image(f, cam["image_width"], cam["image_height"], seed=5) for f in range(6)]
    col = npo.colorize(cam, x, y, z, poses, imgs)
    save("g4_colour.npz", camera=cam_array(cam), poses=poses, x=x, y=y, z=z, images=np.array(imgs), rgb=col["rgb"],
         has=col["has"], count=col["count"], top_score=col["top_score"], top_rgb=col["top_rgb"],
         top_frame=col["top_frame"])

    # g4b: the reference's own match-back (Appendix B3 faithful mode) on g4's scene, and on a variant built to make
    # it differ from index identity: the map moved 150 m away (fp32 ulp 15 um > the 10 um match radius: samples get
    # lost) with 300 points duplicated 4 um beside their originals (cross-credits).  Inputs of (a) are g4's.
    fa = npo.colorize_faithful(cam, x, y, z, poses, imgs)
    xs = (x.astype(np.float64) + 150.0).astype(np.float32)
    dup = np.random.default_rng(4242).choice(len(x), 300, replace=False)  # own generator: g5+ keep their streams
    xs2 = np.concatenate([xs, xs[dup]])
    ys2 = np.concatenate([y, (y[dup].astype(np.float64) + 4e-6).astype(np.float32)])
    zs2 = np.concatenate([z, z[dup]])
    poses_s = poses.copy()
    poses_s[:, 0] += 150.0
    fs = npo.colorize_faithful(cam, xs2, ys2, zs2, poses_s, imgs)
    st = lambda d: np.array([d["stats"][k] for k in ("samples", "unmatched", "self_missed", "cross_credits")], np.int64)
    save("g4b_faithful.npz", rgb=fa["rgb"], has=fa["has"], count=fa["count"], top_score=fa["top_score"],
         top_rgb=fa["top_rgb"], top_frame=fa["top_frame"], stats=st(fa),
         far_x=xs2, far_y=ys2, far_z=zs2, far_poses=poses_s, far_rgb=fs["rgb"], far_has=fs["has"], far_count=fs["count"],
         far_top_score=fs["top_score"], far_top_rgb=fs["top_rgb"], far_top_frame=fs["top_frame"], far_stats=st(fs))

    # g5: MLS on plane / sphere / saddle patches
    n = 700
    a = rng.uniform(-0.12, 0.12, (n, 2))
    plane = np.stack([a[:, 0] + 1.0, a[:, 1] - 2.0, 0.2 * a[:, 0] - 0.1 * a[:, 1] + 0.7 + rng.normal(0, 1e-3, n)], 1)
    d = rng.normal(size=(n, 3))
    d[:, 2] = np.abs(d[:, 2]) + 3.0
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    sphere = np.array([-2.0, 0.5, 1.0]) + (0.6 + rng.normal(0, 1e-3, n))[:, None] * d
    a = rng.uniform(-0.12, 0.12, (n, 2))
    saddle = np.stack([a[:, 0], a[:, 1] + 3.0, 2.0 * a[:, 0] ** 2 - 1.5 * a[:, 1] ** 2 + rng.normal(0, 5e-4, n)], 1)
    stray = rng.uniform(-5, 5, (12, 3))
    pts = np.concatenate([plane, sphere, saddle, stray]).astype(np.float32)
    pts = pts[rng.permutation(len(pts))]
    m = npo.mls(pts[:, 0], pts[:, 1], pts[:, 2])
    save("g5_mls.npz", x=pts[:, 0], y=pts[:, 1], z=pts[:, 2], xyz=m["xyz"], normal=m["normal"],
         curvature=m["curvature"], index=m["index"])

    # g6: odometry / keyframe goldens
    od_poses, ts = synth.make_trajectory(12, spacing=0.06)  # every second pose passes the 0.1 m rule
    lines = []
    for t, p_ in zip(ts, od_poses):
        lines.append(synth.odometry_line(t, p_).rstrip("\n"))
    key = [0]
    for i in range(1, len(od_poses)):
        if np.linalg.norm(od_poses[i, :3] - od_poses[key[-1], :3]) >= 0.1:
            key.append(i)
    save("g6_odometry.npz", text=np.array("\n".join(lines) + "\n"), poses=od_poses, ts=ts,
         keyframes=np.array(key, np.int32))

    # ---- the "next" rows (SURVEY.md 8 f): SOR, voxel-grid dilation, NID cost ----
    rng = np.random.default_rng(20241009)
    # g7: StatisticalOutlierRemoval (k = 20 on 900 points: brute-force twin) -- a bumpy sheet plus strays
    n = 900
    a = rng.uniform(-0.15, 0.15, (n, 2))
    sheet = np.stack([a[:, 0] + 0.5, a[:, 1] - 1.0, 0.3 * np.sin(9 * a[:, 0]) * a[:, 1] + rng.normal(0, 8e-4, n)], 1)
    stray = np.stack([rng.uniform(0.3, 0.7, 25), rng.uniform(-1.2, -0.8, 25), rng.uniform(-0.1, 0.1, 25)], 1)
    pts = np.concatenate([sheet, stray]).astype(np.float32)
    pts = pts[rng.permutation(len(pts))]
    keep, dist, thr = npo.sor(pts[:, 0], pts[:, 1], pts[:, 2], 20, 0.7)
    save("g7_sor.npz", x=pts[:, 0], y=pts[:, 1], z=pts[:, 2], mean_k=np.int32(20), std_mul=np.float64(0.7), keep=keep,
         distance=dist, threshold=np.float64(thr))

    # g8: VOXEL_GRID_DILATION (4 mm voxels, 2 iterations on a 400-point quadric patch)
    n = 400
    a = rng.uniform(-0.06, 0.06, (n, 2))
    pts = np.stack([a[:, 0] - 0.3, a[:, 1] + 0.2, 0.2 * a[:, 0] ** 2 - 0.1 * a[:, 0] * a[:, 1] + rng.normal(0, 3e-4, n)],
                   1).astype(np.float32)
    v = npo.mls_voxel_dilation(pts[:, 0], pts[:, 1], pts[:, 2], 0.03, 2, 0.004, 2)
    save("g8_voxel_dilation.npz", x=pts[:, 0], y=pts[:, 1], z=pts[:, 2], voxel=np.float32(0.004), iterations=np.int32(2),
         xyz=v["xyz"], normal=v["normal"], curvature=v["curvature"], index=v["index"])

    # g9: NID cost of two keyframes at three extrinsics (value; the gradient is checked by finite differences of it)
    cam = nano_camera()
    n = 3000
    zc = rng.uniform(1, 4, n).astype(np.float32)
    xc = (rng.uniform(-0.4, 0.4, n) * zc).astype(np.float32)
    yc = (rng.uniform(-0.22, 0.22, n) * zc).astype(np.float32)
    inten = rng.random(n).astype(np.float32)
    imgs = np.stack([synth.make_image(k, 160, 90) for k in range(2)])
    off = np.array([0, n // 2, n], np.int64)
    Ts, costs = [], []
    for d in ([0, 0, 0, 0, 0, 0], [0.01, -0.02, 0.005, 0.003, -0.002, 0.004], [-0.05, 0.03, 0.02, -0.01, 0.015, -0.02]):
        d = np.array(d, np.float64)
        K = np.array([[0, -d[5], d[4]], [d[5], 0, -d[3]], [-d[4], d[3], 0]])
        T = np.eye(4)
        T[:3, :3] = np.eye(3) + K + 0.5 * K @ K
        T[:3, 3] = (np.eye(3) + 0.5 * K) @ d[:3]
        Ts.append(T)
        costs.append(sum(npo.nid_cost(cam, imgs[k], xc[off[k]:off[k + 1]], yc[off[k]:off[k + 1]], zc[off[k]:off[k + 1]],
                                      inten[off[k]:off[k + 1]], T) for k in range(2)))
    save("g9_nid.npz", camera=cam_array(cam), x=xc, y=yc, z=zc, intensity=inten, images=imgs, offsets=off,
         T=np.stack(Ts), cost=np.array(costs, np.float64))


if __name__ == "__main__":
    main()
