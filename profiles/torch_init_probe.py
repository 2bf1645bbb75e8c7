#!/usr/bin/env python3
"""Does an initialised torch.cuda slow the library's kernels?  The colour step (20 passes) and the hull pass, before and after
torch touches the device (and after it pins host memory)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch
from pointcloudprocessor_amd import capi, synth
cam = synth.camera_dict("cfg")
x, y, z, _ = synth.make_cloud(10_000_000)
poses, _ = synth.make_trajectory(256)
ctx = capi.Context(0)
ctx.set_camera(capi.camera_from_dict(cam))
ctx.upload_cloud(x, y, z)
ctx.set_frames(poses)
imgs = [synth.make_image(f, cam["image_width"], cam["image_height"]) for f in range(8)]
for f in range(256):
    ctx.upload_image(f, imgs[f % 8])
res = {}
def step(tag):
    for _ in range(5):
        ctx.colorize(download=False)
    ctx.synchronize()
    ts = []
    for _ in range(3):
        t = time.perf_counter()
        for _ in range(20):
            ctx.colorize(download=False)
        ctx.synchronize()
        ts.append(round((time.perf_counter() - t) / 20 * 1e3, 4))
    res[tag] = ts
step("before_torch_cuda")
torch.cuda.set_device(0); a = torch.zeros(1, device="cuda"); torch.cuda.synchronize()
step("after_torch_cuda_init")
p = torch.empty(10_000_000, dtype=torch.int32).pin_memory()
step("after_pinning_40MB")
print(json.dumps(res))
