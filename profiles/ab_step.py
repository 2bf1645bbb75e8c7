#!/usr/bin/env python3
"""Step / kernel times of the colour path for the library PCP_HIP_LIBRARY names (A/B of kernel variants):
python profiles/ab_step.py [cfg|ref] [points] [frames]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401

from pointcloudprocessor_amd import capi, synth  # noqa: E402

cam = synth.camera_dict(sys.argv[1] if len(sys.argv) > 1 else "cfg")
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
F = int(sys.argv[3]) if len(sys.argv) > 3 else 256
x, y, z, _ = synth.make_cloud(N)
poses, _ = synth.make_trajectory(F)
ctx = capi.Context(0)
ctx.set_camera(capi.camera_from_dict(cam))
ctx.upload_cloud(x, y, z)
ctx.set_frames(poses)
imgs = [synth.make_image(f, cam["image_width"], cam["image_height"]) for f in range(8)]
for f in range(F):
    ctx.upload_image(f, imgs[f % 8])
for _ in range(3):
    ctx.colorize(download=False)
ctx.synchronize()
t = time.perf_counter()
for _ in range(20):
    ctx.colorize(download=False)
ctx.synchronize()
step = (time.perf_counter() - t) / 20
ctx.timing_enable(True)
ctx.timing_reset()
for _ in range(5):
    ctx.colorize(download=False)
ctx.synchronize()
out = ctx.download_result_packed()
res = {"lib": os.environ.get("PCP_HIP_LIBRARY", "default"), "camera": sys.argv[1] if len(sys.argv) > 1 else "cfg",
       "step_ms": round(step * 1e3, 4),
       "kernels_ms": {ctx.kernel_name(k): round(ctx.timing_get(k)[0] / 5, 4) for k in (capi.K_TILE_MASK, capi.K_DEPTH, capi.K_COLOUR, capi.K_MISC)},
       "coloured": int(((out >> 24) & 1).sum()), "checksum": int(out.astype(np.uint64).sum())}
print(json.dumps(res))
