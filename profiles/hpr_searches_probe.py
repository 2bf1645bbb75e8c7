#!/usr/bin/env python3
"""Per keyframe of C3: candidates and searches handed to k_hpr_tilt (PCP_HPR_DEBUG lines on stderr are parsed by the caller).
python3 profiles/hpr_searches_probe.py 2> log"""
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
    for f in range(256):
        print(f"--- keyframe {f}", file=sys.stderr, flush=True)
        ctx.cull_frame(f)
