#!/usr/bin/env python3
"""SURVEY 8(d)(i) boundary, pieces timed separately: images from pinned host memory -> colours on the host.
python profiles/pipeline_probe.py"""
import os
import json
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pointcloudprocessor_amd import capi, synth  # noqa: E402

N, F = 10_000_000, 256
cam = synth.camera_dict("cfg")
W, H = cam["image_width"], cam["image_height"]
x, y, z, _ = synth.make_cloud(N)
poses, _ = synth.make_trajectory(F)
ctx = capi.Context(0)
ctx.set_camera(capi.camera_from_dict(cam))
ctx.upload_cloud(x, y, z)
ctx.set_frames(poses)
stage = torch.empty((F, H, W, 3), dtype=torch.uint8).pin_memory()
snp = stage.numpy()
for f in range(8):
    snp[f] = synth.make_image(f, W, H)
for f in range(8, F):
    snp[f] = snp[f % 8]
pinned = torch.empty(N, dtype=torch.int32).pin_memory()


def run(batch, uploads=True, compute=True, download=True):
    ctx.colour_reset()
    if uploads:
        for f in range(F):
            ctx.upload_image_async(f, snp[f])
    if compute:
        ctx.depth_pass()
        if batch >= F:
            ctx.colorize_from_depth(download=False)
        else:
            for f0 in range(0, F, batch):
                ctx.colour_pass(f0, min(F, f0 + batch))
            ctx.colour_finalise(download=False)
        if download:
            ctx.download_result_packed(out_ptr=pinned.data_ptr())
    ctx.synchronize()


def timed(**kw):
    run(**kw)
    ts = []
    for _ in range(3):
        t = time.perf_counter()
        run(**kw)
        ts.append((time.perf_counter() - t) * 1e3)
    return round(min(ts), 2)


res = {"uploads_only": timed(batch=32, compute=False)}
for b in (16, 32, 64, 128, 256):
    res[f"compute_only_batch{b}"] = timed(batch=b, uploads=False)
    res[f"pipelined_batch{b}"] = timed(batch=b)
print(json.dumps(res))
