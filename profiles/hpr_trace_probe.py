#!/usr/bin/env python3
"""hidden_points_removal at C3 (10 M points, 1920x1080): eight keyframes of pcp_cull_frame for a rocprofv3 kernel trace
(per-kernel share of the call), and the whole-run hull pass of 256 keyframes timed by the host.
python3 profiles/hpr_trace_probe.py [whole]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
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
    ctx.cull_frame(0)
    ctx.synchronize()
    res = {}
    ts = []
    for f in (0, 32, 64, 100, 144, 192, 224, 255):
        t0 = time.perf_counter()
        keep, _, kept = ctx.cull_frame(f)
        ts.append(round((time.perf_counter() - t0) * 1e3, 2))
    res["cull_frame_ms"] = ts
    if len(sys.argv) > 1 and sys.argv[1] == "whole":
        ctx.depth_pass()
        ctx.synchronize()
        t0 = time.perf_counter()
        ctx.depth_pass()
        ctx.synchronize()
        res["hull_pass_256_keyframes_s"] = round(time.perf_counter() - t0, 3)
    print(json.dumps(res))
