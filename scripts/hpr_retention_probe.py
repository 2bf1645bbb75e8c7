#!/usr/bin/env python3
"""How much of a keyframe's HPR candidate set qhull's convex hull keeps at map density (VERDICT r2 "What's weak" #2).

CPU only, test infrastructure (it runs the numpy twin + scipy's qhull_r, so it lives under tests/; not collected by
pytest).  For `--frames` keyframes of the C3 scene (10 M points, 256 keyframes) and both cameras: the candidates of
ViewCulling::hidden_points_removal (view_culling.cpp:276-288), the hull vertices among the spherically flipped
candidates (:291-329, flip radius 90000), and what the z-buffer routine (:52-174) keeps of the same cloud.

    python scripts/hpr_retention_probe.py --points 10000000 --frames 16 > profiles/r03_hpr_retention.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import np_oracle as npo  # noqa: E402
from pointcloudprocessor_amd import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--points", type=int, default=10_000_000)
ap.add_argument("--keyframes", type=int, default=256)
ap.add_argument("--frames", type=int, default=16)
ap.add_argument("--cameras", default="cfg,ref")
args = ap.parse_args()

x, y, z, _ = synth.make_cloud(args.points)
poses, _ = synth.make_trajectory(args.keyframes)
out = {"points": args.points, "keyframes": args.keyframes, "flip_radius": 90000.0, "cameras": {}}
for camname in args.cameras.split(","):
    cam = synth.camera_dict(camname)
    rows = []
    for f in range(0, args.keyframes, max(1, args.keyframes // args.frames)):
        w2c, _ = npo.pose_to_matrices(poses[f])
        t0 = time.time()
        vis = npo.hpr_frame(cam, w2c, x, y, z)
        t_h = time.time() - t0
        cand, _ = npo.hpr_candidates(cam, w2c, x, y, z)
        keep = npo.cull_frame(cam, w2c, x, y, z)[0]
        m = int(np.count_nonzero(cand))
        rows.append({"keyframe": f, "candidates": m, "hull_vertices": int(len(vis)),
                     "kept_fraction": round(len(vis) / max(m, 1), 4), "zbuffer_keeps": int(np.count_nonzero(keep)),
                     "seconds": round(t_h, 2)})
        print(camname, rows[-1], file=sys.stderr, flush=True)
    tot_c = sum(r["candidates"] for r in rows)
    tot_v = sum(r["hull_vertices"] for r in rows)
    out["cameras"][camname] = {"image": [cam["image_width"], cam["image_height"]], "frames": rows,
                               "candidates_total": tot_c, "hull_vertices_total": tot_v,
                               "kept_fraction": round(tot_v / max(tot_c, 1), 4),
                               "kept_fraction_min": min(r["kept_fraction"] for r in rows),
                               "kept_fraction_max": max(r["kept_fraction"] for r in rows)}
print(json.dumps(out, indent=1))
