#!/usr/bin/env python3
"""Why the hull pass takes 0.167 s inside bench.py and 0.148 s alone: the same pass after (a) torch.cuda initialised, (b) pinned
buffers, (c) a second context holding the workload's 256 keyframe images."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch
from pointcloudprocessor_amd import capi, pipeline, synth
cam = synth.camera_dict("cfg")
x, y, z, _ = synth.make_cloud(10_000_000)
poses, _ = synth.make_trajectory(256)
cull = capi.default_cull_params()
cull.cull_mode = capi.CULL_HPR
res = {}
def hull(tag):
    with capi.Context(0) as ctx:
        ctx.set_camera(capi.camera_from_dict(cam), cull)
        ctx.upload_cloud(x, y, z)
        ctx.set_frames(poses)
        ctx.depth_pass(); ctx.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); ctx.depth_pass(); ctx.synchronize(); ts.append(round(time.perf_counter() - t0, 4))
    res[tag] = ts
hull("plain")
torch.cuda.set_device(0); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
hull("torch_cuda_initialised")
pinned = [torch.empty(10_000_000, dtype=torch.int32).pin_memory() for _ in range(2)]
hull("pinned_buffers")
eng = pipeline.HipEngine(0)
eng.configure(cam, capi.default_cull_params())
eng.upload_cloud(x, y, z)
eng.ctx.set_frames(poses)
img = synth.make_image(0, cam["image_width"], cam["image_height"])
for f in range(256):
    eng.ctx.upload_image(f, img)
eng.ctx.colorize()
hull("main_context_alive")
print(json.dumps(res))
