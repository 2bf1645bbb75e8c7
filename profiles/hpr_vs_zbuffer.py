#!/usr/bin/env python3
"""What switching the cull costs in colours: the C3 colourisation (10 M points x 256 keyframes @1920x1080) with
cull_mode = PCP_CULL_HPR (hidden_points_removal, the routine the reference binary calls) against PCP_CULL_ZBUFFER (the
routine north_star names), both on the GPU.  Prints one JSON object (profiles/r03_hpr_vs_zbuffer.json)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudprocessor_amd import capi, synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
F = int(sys.argv[2]) if len(sys.argv) > 2 else 256
cam = synth.camera_dict("cfg")
x, y, z, _ = synth.make_cloud(N)
poses, _ = synth.make_trajectory(F)
imgs = [synth.make_image(f, cam["image_width"], cam["image_height"]) for f in range(F)]
res = {}
for name, mode in (("zbuffer", capi.CULL_ZBUFFER), ("hpr", capi.CULL_HPR), ("hpr_candidates", capi.CULL_HPR_CANDIDATES)):
    cull = capi.default_cull_params()
    cull.cull_mode = mode
    with capi.Context(0) as ctx:
        ctx.set_camera(capi.camera_from_dict(cam), cull)
        ctx.upload_cloud(x, y, z)
        ctx.set_frames(poses)
        for f, im in enumerate(imgs):
            ctx.upload_image(f, im)
        ctx.synchronize()
        t0 = time.perf_counter()
        ctx.depth_pass()
        ctx.synchronize()
        t1 = time.perf_counter()
        ctx.colour_reset()
        ctx.colour_pass()
        r = ctx.colour_finalise(want_top=True)
        t2 = time.perf_counter()
        res[name] = dict(rgb=r["rgb"], has=r["has"], count=r["count"], cull_s=t1 - t0, colour_s=t2 - t1)
        if mode == capi.CULL_HPR:
            res[name]["last_frame_stats"] = ctx.hpr_stats()
out = {"points": N, "keyframes": F, "camera": "cfg 1920x1080"}
zb = res["zbuffer"]
for name in ("hpr", "hpr_candidates"):
    a = res[name]
    both = (a["has"] > 0) & (zb["has"] > 0)
    d = np.abs(a["rgb"][both].astype(int) - zb["rgb"][both].astype(int)).max(axis=1)
    out[name] = {
        "coloured": int((a["has"] > 0).sum()), "coloured_zbuffer": int((zb["has"] > 0).sum()),
        "coloured_in_both": int(both.sum()), "only_here": int(((a["has"] > 0) & ~(zb["has"] > 0)).sum()),
        "only_zbuffer": int((~(a["has"] > 0) & (zb["has"] > 0)).sum()),
        "views_total": int(a["count"].astype(np.int64).sum()), "views_total_zbuffer": int(zb["count"].astype(np.int64).sum()),
        "same_colour_fraction_of_both": round(float((d == 0).mean()), 4),
        "max_channel_diff_le_8_fraction": round(float((d <= 8).mean()), 4),
        "max_channel_diff_mean": round(float(d.mean()), 2), "max_channel_diff_p95": int(np.percentile(d, 95)),
        "cull_seconds": round(a["cull_s"], 3), "colour_seconds": round(a["colour_s"], 3),
    }
    if "last_frame_stats" in a:
        out[name]["last_keyframe_hull_stats"] = a["last_frame_stats"]
out["zbuffer"] = {"cull_seconds": round(zb["cull_s"], 4), "colour_seconds": round(zb["colour_s"], 4)}
print(json.dumps(out, indent=1))
