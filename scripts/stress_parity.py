#!/usr/bin/env python3
"""Randomised parity stress (not collected by pytest): many seeds x distortion models x camera sets, the HIP path
against the oracle -- colours, top-5 lists, depth maps of a few keyframes.  Prints one line per case; exit 1 on the
first mismatch.     python scripts/stress_parity.py [cases] [points] [keyframes]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import cam_struct  # noqa: E402
from oracle import oracle_capi as oc  # noqa: E402
from pointcloudprocessor_amd import capi, synth  # noqa: E402

DIST = [{}, dict(k1=-0.28, k2=0.07, p1=0.0012, p2=-0.0009, k3=-0.004), dict(k1=0.35, k2=0.4, p1=-0.01, p2=0.02, k3=0.2),
        dict(k1=0.0, k2=0.0, p1=0.05, p2=-0.04, k3=0.0), dict(k1=0.0, k2=0.0, p1=0.0, p2=0.0, k3=0.0),
        dict(k1=-0.9, k2=0.0, p1=0.0, p2=0.0, k3=0.0)]


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 300_000
    F = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    oc.build()
    for case in range(cases):
        rng = np.random.default_rng(1000 + case)
        cd = dict(synth.camera_dict(["tiny", "cfg"][case % 2]))
        cd.update(DIST[case % len(DIST)])
        cd["fx"] *= rng.uniform(0.6, 1.6)
        cd["fy"] *= rng.uniform(0.6, 1.6)
        cd["cx"] += rng.uniform(-40, 40)
        cd["cy"] += rng.uniform(-40, 40)
        W, H = cd["image_width"], cd["image_height"]
        x, y, z, _ = synth.make_cloud(n, seed=77 + case)
        if case % 4 == 3:  # a sprinkle of non-finite and far-away points (PCL clouds with is_dense == false)
            bad = rng.choice(n, n // 1000, replace=False)
            x[bad[0::4]] = np.nan
            y[bad[1::4]] = np.inf
            z[bad[2::4]] = -np.inf
            x[bad[3::4]] *= 1e6
        poses, _ = synth.make_trajectory(F)
        poses = poses[rng.permutation(F)]  # keyframe order must not matter to the machinery
        imgs = [synth.make_image(int(rng.integers(0, 64)), W, H) for _ in range(8)]
        imgs = [imgs[f % 8] for f in range(F)]
        cull = capi.default_cull_params()
        ocull = oc.default_cull_params()
        ds = int(rng.choice([14, 14, 7, 20]))
        slack = float(rng.choice([0.05, 0.05, 0.0, 0.2]))
        cull.downsample_factor = ocull.downsample_factor = ds
        cull.depth_slack = ocull.depth_slack = slack
        # the cull routine and the match-back arithmetic: mostly the defaults (z-buffer, fp32 round trip)
        cull.cull_mode = ocull.cull_mode = int(rng.choice([0, 0, 0, 1]))
        cull.match_mode = ocull.match_mode = int(rng.choice([1, 1, 0]))
        adjust = bool(rng.integers(0, 2))  # generateColorMap's HSV round trip inside the upload
        T_opt = None
        if case % 3 == 2:
            T_opt = np.eye(4)
            T_opt[:3, 3] = rng.normal(0, 0.02, 3)
        ctx = capi.Context(0)
        ctx.set_camera(cam_struct(capi, cd), cull)
        ctx.upload_cloud(x, y, z)
        ctx.set_frames(poses, T_opt)
        ctx.set_image_adjust(adjust)
        for f, im in enumerate(imgs):
            ctx.upload_image(f, im)
        if adjust:
            adj = {id(im): oc.hsv_round_trip(im) for im in imgs[:8]}
            imgs = [adj[id(im)] for im in imgs]
        ref = oc.colorize(cam_struct(oc, cd), ocull, x, y, z, poses, imgs, T_opt=T_opt, threads=oc.hardware_threads(), want_top=True)
        ctx.depth_pass()
        ctx.colour_reset()
        ctx.colour_pass()
        got = ctx.colour_finalise(want_top=True)
        ok = all(np.array_equal(got[k], ref[k]) for k in ("count", "top_frame", "top_rgb", "top_score", "rgb", "has"))
        one = ctx.colorize()
        ok = ok and np.array_equal(one["rgb"], ref["rgb"]) and np.array_equal(one["has"], ref["has"])
        for f in (0, F // 2, F - 1):
            w2c, _ = oc.pose_to_matrices(poses[f], T_opt)
            keep_r, dmap, _ = oc.cull_frame(cam_struct(oc, cd), ocull, w2c, x, y, z)
            if ocull.cull_mode == 0:  # the HPR-candidate mode has no depth maps
                ok = ok and np.array_equal(ctx.download_depth_map(f).view(np.uint32), dmap.view(np.uint32))
            ok = ok and np.array_equal(ctx.cull_frame(f)[0], keep_r)
        ctx.close()
        print(f"case {case:3d} cam={'tiny' if case % 2 == 0 else 'cfg '} dist={case % len(DIST)} ds={ds:2d} slack={slack:.2f} "
              f"T_opt={'y' if T_opt is not None else 'n'} cull={ocull.cull_mode} match={ocull.match_mode} hsv={int(adjust)} coloured={int(ref['has'].sum()):7d}  {'ok' if ok else 'MISMATCH'}", flush=True)
        if not ok:
            sys.exit(1)
    print("all cases identical")


if __name__ == "__main__":
    main()
