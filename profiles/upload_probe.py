#!/usr/bin/env python3
"""Upload path probe: 256 BGR8 keyframes from pinned host memory through pcp_upload_image_async (copy + pack per
keyframe on the upload lanes), against one plain pinned H2D copy of the same bytes.  python profiles/upload_probe.py [cfg|ref]"""
import os
import json
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pointcloudprocessor_amd import capi, synth  # noqa: E402

cam = synth.camera_dict(sys.argv[1] if len(sys.argv) > 1 else "cfg")
F = int(sys.argv[2]) if len(sys.argv) > 2 else 256
W, H = cam["image_width"], cam["image_height"]
ctx = capi.Context(0)
ctx.set_camera(capi.camera_from_dict(cam))
x, y, z, _ = synth.make_cloud(100000)
ctx.upload_cloud(x, y, z)
poses, _ = synth.make_trajectory(F)
ctx.set_frames(poses)
stage = torch.empty((F, H, W, 3), dtype=torch.uint8).pin_memory()
snp = stage.numpy()
img = synth.make_image(0, W, H)
for f in range(F):
    snp[f] = img
dev = torch.empty(stage.shape, dtype=torch.uint8, device="cuda:0")
dev.copy_(stage, non_blocking=True)
torch.cuda.synchronize()
t = time.perf_counter()
dev.copy_(stage, non_blocking=True)
torch.cuda.synchronize()
plain = time.perf_counter() - t
res = {"bytes": stage.numel(), "plain_h2d_ms": plain * 1e3, "plain_GBps": stage.numel() / plain / 1e9}
for rep in range(3):
    ctx.synchronize()
    t = time.perf_counter()
    for f in range(F):
        ctx.upload_image_async(f, snp[f])
    t_host = time.perf_counter() - t
    ctx.synchronize()
    dt = time.perf_counter() - t
    res[f"upload_ms_{rep}"] = dt * 1e3
    res[f"host_queue_ms_{rep}"] = t_host * 1e3
res["upload_GBps"] = stage.numel() / dt / 1e9
print(json.dumps(res))
