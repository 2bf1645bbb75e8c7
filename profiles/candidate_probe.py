#!/usr/bin/env python3
"""How many (point, keyframe) pairs are cull candidates with a colour pixel, and how many of them survive the depth
test (the samples that fetch a texel): 10 M points, 16 keyframes sampled from the 256, both cameras."""
import os
import json
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401

from pointcloudprocessor_amd import capi, synth  # noqa: E402

N, F = 10_000_000, 256
x, y, z, _ = synth.make_cloud(N)
poses, _ = synth.make_trajectory(F)
out = {}
for camname in ("cfg", "ref"):
    cam = synth.camera_dict(camname)
    ctx = capi.Context(0)
    ctx.set_camera(capi.camera_from_dict(cam))
    ctx.upload_cloud(x, y, z)
    ctx.set_frames(poses)
    cand = kept = 0
    frames = list(range(0, F, 16))
    for f in frames:
        p = ctx.project_frame(f, want_cam=False)
        c = (p["cell"] >= 0) & (p["pixel"] >= 0)
        keep, _, _ = ctx.cull_frame(f)
        cand += int(c.sum())
        kept += int((keep.astype(bool) & (p["pixel"] >= 0)).sum())
        if f == frames[0]:
            # pixel spacing of neighbouring candidates: median distance to the nearest other candidate (sample)
            pix = p["pixel"][c][:20000]
            u, v = pix % cam["image_width"], pix // cam["image_width"]
            from scipy.spatial import cKDTree
            d, _ = cKDTree(np.stack([u, v], 1)).query(np.stack([u, v], 1), k=2)
            out[camname + "_median_nn_px_in_20k_sample"] = float(np.median(d[:, 1]))
    out[camname] = {"frames": len(frames), "candidates_per_frame": cand / len(frames), "kept_per_frame": kept / len(frames),
                    "kept_fraction": kept / cand, "candidates_all_256": cand / len(frames) * F, "kept_all_256": kept / len(frames) * F}
    ctx.close()
print(json.dumps(out))
