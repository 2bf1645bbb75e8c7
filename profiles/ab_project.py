#!/usr/bin/env python3
"""k_project_frame on a 40 M-point cloud (outside the Infinity Cache) for the library PCP_HIP_LIBRARY names."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401

from pointcloudprocessor_amd import capi, synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40_000_000
cam = synth.camera_dict("cfg")
x, y, z, _ = synth.make_cloud(10_000_000)
reps = -(-N // len(x))
rx = np.concatenate([x + np.float32(k * 1e-4) for k in range(reps)])[:N]
ry = np.concatenate([y + np.float32(k * 1e-4) for k in range(reps)])[:N]
rz = np.concatenate([z + np.float32(k * 1e-4) for k in range(reps)])[:N]
poses, _ = synth.make_trajectory(256)
ctx = capi.Context(0)
ctx.set_camera(capi.camera_from_dict(cam))
ctx.upload_cloud(rx, ry, rz)
ctx.set_frames(poses)
for f in range(4):
    ctx.project_frame(f, device_only=True)
out = {}
for rep in range(3):
    ctx.timing_enable(True)
    ctx.timing_reset()
    for f in range(64):
        ctx.project_frame(f, device_only=True)
    ms, n = ctx.timing_get(capi.K_PROJECT)
    out[f"avg_us_{rep}"] = round(ms / n * 1e3, 2)
out["GBps"] = round(20 * N / (ms / n * 1e-3) / 1e9, 1)
out["lib"] = os.path.basename(os.environ.get("PCP_HIP_LIBRARY", "default"))
print(json.dumps(out))
