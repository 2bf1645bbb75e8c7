#!/usr/bin/env python3
"""PCP_CULL_HPR against the oracle's exact quickhull on the largest candidate sets of the C3 scene: the reference's
4096x3000 camera, keyframes 80 and 192 (1.3 and 1.7 M candidates).  Not collected by pytest (each oracle run takes seconds)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle_capi as oc, np_oracle as npo
from pointcloudprocessor_amd import capi, synth

cam = synth.camera_dict("ref")
ocam = oc.Camera()
for k, _t in oc.Camera._fields_:
    setattr(ocam, k, cam[k])
x, y, z, _ = synth.make_cloud(10_000_000)
poses, _ = synth.make_trajectory(256)
cull = capi.default_cull_params()
cull.cull_mode = capi.CULL_HPR
with capi.Context(0) as ctx:
    ctx.set_camera(capi.camera_from_dict(cam), cull)
    ctx.upload_cloud(x, y, z)
    ctx.set_frames(poses)
    for f in (80, 192):
        ctx.cull_frame(f)
        ctx.synchronize()
        t0 = time.perf_counter()
        keep, _, kept = ctx.cull_frame(f)
        t1 = time.perf_counter()
        st = ctx.hpr_stats()
        w2c, _ = npo.pose_to_matrices(poses[f])
        okeep, ost = oc.hpr_frame(ocam, w2c, x, y, z)
        t2 = time.perf_counter()
        print(json.dumps({"frame": f, "candidates": st["candidates"], "gpu_ms": round((t1 - t0) * 1e3, 2), "oracle_s": round(t2 - t1, 2),
                          "kept_gpu": int(kept), "kept_oracle": ost["kept"], "mismatch": int((keep != okeep).sum()),
                          "exact_path": st["exact_path"], "unresolved": st["unresolved"]}), flush=True)
