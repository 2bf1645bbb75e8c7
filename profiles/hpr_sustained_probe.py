#!/usr/bin/env python3
"""The whole-run hull pass of C3 repeated for ~6 s: does its time drift with the clock the device holds under a sustained load?"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
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
    ts = []
    for _ in range(36):
        t0 = time.perf_counter()
        ctx.depth_pass()
        ctx.synchronize()
        ts.append(round(time.perf_counter() - t0, 4))
print(json.dumps({"hull_pass_s": ts}))
