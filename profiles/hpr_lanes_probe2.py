#!/usr/bin/env python3
"""Whole-run hull pass of C3 by keyframes in flight (PCP_HPR_LANES is read per pass).  python3 profiles/hpr_lanes_probe2.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudprocessor_amd import capi, synth
cam = synth.camera_dict("cfg")
x, y, z, _ = synth.make_cloud(10_000_000)
poses, _ = synth.make_trajectory(256)
cull = capi.default_cull_params()
cull.cull_mode = capi.CULL_HPR
out = {}
with capi.Context(0) as ctx:
    ctx.set_camera(capi.camera_from_dict(cam), cull)
    ctx.upload_cloud(x, y, z)
    ctx.set_frames(poses)
    for lanes in (4, 2, 3, 5, 6, 8, 4):
        os.environ["PCP_HPR_LANES"] = str(lanes)
        ctx.depth_pass(); ctx.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); ctx.depth_pass(); ctx.synchronize(); ts.append(round(time.perf_counter() - t0, 4))
        out.setdefault(str(lanes), []).append(ts)
print(json.dumps(out))
