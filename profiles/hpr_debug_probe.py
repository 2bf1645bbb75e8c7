#!/usr/bin/env python3
"""PCP_HPR_DEBUG tallies of a few keyframes of C3 (what each pass of the hull leaves to the next)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PCP_HPR_DEBUG"] = "1"
from pointcloudprocessor_amd import capi, synth
cam = synth.camera_dict("cfg")
x, y, z, _ = synth.make_cloud(10_000_000)
poses, _ = synth.make_trajectory(256)
cull = capi.default_cull_params()
cull.cull_mode = capi.CULL_HPR
with capi.Context(0) as ctx:
    ctx.set_camera(capi.camera_from_dict(cam), cull)
    ctx.upload_cloud(x, y, z)
    ctx.set_frames(poses)
    for f in (0, 100, 144, 192):
        print(f"--- keyframe {f}", file=sys.stderr, flush=True)
        ctx.cull_frame(f)
        print(ctx.hpr_stats(), file=sys.stderr, flush=True)
