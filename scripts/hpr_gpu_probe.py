#!/usr/bin/env python3
"""GPU hull (PCP_CULL_HPR) against the oracle's exact quickhull on the synthetic scene; not collected by pytest."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle_capi as oc, np_oracle as npo
from pointcloudprocessor_amd import capi, synth

sizes = [int(v) for v in (sys.argv[1:] or ["20000", "200000", "2000000"])]
for camname in ("cfg",):
    cam = synth.camera_dict(camname)
    ocam = oc.Camera()
    for k, _t in oc.Camera._fields_:
        setattr(ocam, k, cam[k])
    for N in sizes:
        x, y, z, _ = synth.make_cloud(N)
        poses, _ = synth.make_trajectory(8 if N < 5_000_000 else 256)
        cull = capi.default_cull_params()
        cull.cull_mode = capi.CULL_HPR
        with capi.Context(0) as ctx:
            ctx.set_camera(capi.camera_from_dict(cam), cull)
            ctx.upload_cloud(x, y, z)
            ctx.set_frames(poses)
            for f in ((1, 3) if N < 5_000_000 else (0, 100, 144)):
                ctx.cull_frame(f)  # warm-up (allocations)
                ctx.synchronize()
                t0 = time.perf_counter()
                keep, _, kept = ctx.cull_frame(f)
                t1 = time.perf_counter()
                st = ctx.hpr_stats()
                w2c, _ = npo.pose_to_matrices(poses[f])
                t2 = time.perf_counter()
                okeep, ost = oc.hpr_frame(ocam, w2c, x, y, z)
                t3 = time.perf_counter()
                diff = np.nonzero(keep != okeep)[0]
                print(json.dumps({"N": N, "frame": f, "gpu_ms": round((t1 - t0) * 1e3, 2), "oracle_s": round(t3 - t2, 2),
                                  "kept_gpu": int(kept), "kept_oracle": ost["kept"], "mismatch": int(len(diff)),
                                  "first": diff[:8].tolist(), "stats": st}), flush=True)
