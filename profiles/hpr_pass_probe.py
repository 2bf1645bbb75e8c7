#!/usr/bin/env python3
"""The whole-run hull pass of C3 (10 M points, 256 keyframes, PCP_CULL_HPR), twice (allocations, then the measured one),
for a rocprofv3 kernel trace.   PCP_HPR_LANES=4 python3 profiles/hpr_pass_probe.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
    ctx.depth_pass()
    ctx.synchronize()
    t0 = time.perf_counter()
    ctx.depth_pass()
    ctx.synchronize()
    print(json.dumps({"lanes": os.environ.get("PCP_HPR_LANES", "default"), "hull_pass_s": round(time.perf_counter() - t0, 4)}))
