import sys, time, json
import numpy as np
sys.path.insert(0, '.')
import torch
from pointcloudprocessor_amd import capi, pipeline, synth
N, F = 10_000_000, 256
cam = synth.camera_dict(sys.argv[1] if len(sys.argv) > 1 else "cfg")
x, y, z, _ = synth.make_cloud(N)
poses, _ = synth.make_trajectory(F)
res = {}
for mode in (0, 1):
    ctx = capi.Context(0)
    cull = capi.default_cull_params(); cull.match_mode = mode
    ctx.set_camera(capi.camera_from_dict(cam), cull)
    ctx.upload_cloud(x, y, z); ctx.set_frames(poses)
    for f in range(F):
        ctx.upload_image(f, synth.make_image(f, cam["image_width"], cam["image_height"]))
    for _ in range(3):
        ctx.colorize(download=False)
    ctx.synchronize()
    ctx.timing_enable(True); ctx.timing_reset()
    t = time.perf_counter()
    for _ in range(10):
        ctx.colorize(download=False)
    ctx.synchronize()
    dt = (time.perf_counter() - t) / 10
    res[mode] = dict(step_ms=dt * 1e3, colour_ms=ctx.timing_get(capi.K_COLOUR)[0] / 10, depth_ms=ctx.timing_get(capi.K_DEPTH)[0] / 10)
    out = ctx.download_result_packed()
    res[mode]["coloured"] = int(((out >> 24) & 1).sum()); res[mode]["sum"] = int(out.astype(np.uint64).sum())
    ctx.close()
print(json.dumps(res))
