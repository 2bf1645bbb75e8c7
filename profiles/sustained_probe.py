"""Step time over a long run (clock / power behaviour under sustained load): prints the mean step time of
consecutive 20-step blocks, with and without the result download.   python profiles/sustained_probe.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pointcloudprocessor_amd import pipeline, synth  # noqa: E402

cam = synth.camera_dict("cfg")
N, F = 10_000_000, 256
eng = pipeline.HipEngine(0)
eng.configure(cam)
x, y, z, _ = synth.make_cloud(N)
eng.upload_cloud(x, y, z)
poses, _ = synth.make_trajectory(F)
eng.ctx.set_frames(poses)
for f in range(F):
    eng.ctx.upload_image(f, synth.make_image(f, 1920, 1080))
col = pipeline.PointCloudColorizer(eng)
pinned = [torch.empty(N, dtype=torch.int32).pin_memory() for _ in range(2)]
for download in (True, False):
    eng.ctx.synchronize()
    time.sleep(1.0)
    out = []
    for block in range(20):
        t0 = time.perf_counter()
        for s in range(20):
            col.run(download=False)
            if download:
                eng.ctx.download_result_packed_async(pinned[s & 1].data_ptr())
        eng.ctx.synchronize()
        out.append((time.perf_counter() - t0) / 20 * 1e3)
    print("download" if download else "no download", " ".join(f"{v:.2f}" for v in out))

# how far the host may run ahead of the device: synchronise every `every` steps (0 = never inside the 200 steps)
for every in (0, 50, 20, 8, 4, 2, 1):
    eng.ctx.synchronize()
    time.sleep(0.5)
    t0 = time.perf_counter()
    for s in range(200):
        col.run(download=False)
        eng.ctx.download_result_packed_async(pinned[s & 1].data_ptr())
        if every and (s + 1) % every == 0:
            eng.ctx.synchronize()
    eng.ctx.synchronize()
    print(f"sync every {every:3d}: {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms/step")
