#!/usr/bin/env python3
"""SURVEY 8(d)(i) boundary (keyframes start in pinned host memory): the shipped per-keyframe in-place reads of pinned memory
against DMA copies of blocks of keyframes into a device staging buffer + pcp_upload_image_async from device pointers.
python3 profiles/block_upload_probe.py [keyframes per block ...]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from pointcloudprocessor_amd import capi, pipeline, synth

N, F = 10_000_000, 256
cam = synth.camera_dict("cfg")
W, H = cam["image_width"], cam["image_height"]
x, y, z, _ = synth.make_cloud(N)
poses, _ = synth.make_trajectory(F)
eng = pipeline.HipEngine(0)
eng.configure(cam, None)
eng.upload_cloud(x, y, z)
eng.ctx.set_frames(poses)
stage = torch.empty((F, H, W, 3), dtype=torch.uint8).pin_memory()
snp = stage.numpy()
for f in range(8):
    snp[f] = synth.make_image(f, W, H)
for f in range(8, F):
    snp[f] = snp[f % 8]
out = torch.empty(N, dtype=torch.int32).pin_memory()
dev = torch.empty(stage.shape, dtype=torch.uint8, device="cuda:0")
copy_stream = torch.cuda.Stream()
batch = 64


def finish():
    eng.ctx.colour_finalise(download=False)
    eng.ctx.download_result_packed(out_ptr=out.data_ptr())


def in_place():
    eng.ctx.colour_reset()
    for f in range(F):
        eng.ctx.upload_image_async(f, snp[f])
    eng.ctx.depth_pass()
    for f0 in range(0, F, batch):
        eng.ctx.colour_pass(f0, min(F, f0 + batch))
    finish()


def blocks(per):
    eng.ctx.colour_reset()
    evs = []
    with torch.cuda.stream(copy_stream):
        for b0 in range(0, F, per):
            dev[b0:b0 + per].copy_(stage[b0:b0 + per], non_blocking=True)
            e = torch.cuda.Event()
            e.record(copy_stream)
            evs.append(e)
    eng.ctx.depth_pass()  # needs no image: runs under the first copy
    done = 0
    for k, b0 in enumerate(range(0, F, per)):
        evs[k].synchronize()  # the block is on the device; the next ones keep coming
        for f in range(b0, min(F, b0 + per)):
            eng.ctx.upload_image_async_ptr(f, dev[f].data_ptr(), 3 * W)
        up = min(F, b0 + per)
        while done + batch <= up or (up == F and done < F):
            eng.ctx.colour_pass(done, min(F, done + batch))
            done = min(F, done + batch)
    finish()


res = {}
for name, fn in [("in_place", in_place)] + [(f"blocks_{p}", (lambda p=p: blocks(p))) for p in [int(v) for v in (sys.argv[1:] or ["16", "32", "64"])]]:
    fn()
    eng.ctx.synchronize()
    ts = []
    for _ in range(5):
        t = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t) * 1e3)
    res[name] = [round(v, 2) for v in ts]
    ref = np.array(out.numpy(), copy=True)
    if name == "in_place":
        base = ref
    else:
        assert np.array_equal(ref, base), name
print(json.dumps(res))
