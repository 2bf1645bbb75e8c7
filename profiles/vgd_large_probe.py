#!/usr/bin/env python3
"""VOXEL_GRID_DILATION at the reference's configuration (1 mm x 4) on a map larger than a room: the C3 room (12 x 10 x 4 m)
tiled 3 x 4 -> 36 x 40 x 4 m, 12 M points.  As a dense bitmap the 36 000 x 40 000 x 4 000 voxel box is 720 GB; as bricks of
16^3 voxels the memory follows the surface.  Prints one JSON line.   python3 profiles/vgd_large_probe.py [points_per_tile]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pointcloudprocessor_amd import capi, synth

per = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
x0, y0, z0, _ = synth.make_cloud(per)
xs, ys, zs = [], [], []
for i in range(3):
    for j in range(4):
        xs.append(x0 + np.float32(12.0 * i))
        ys.append(y0 + np.float32(10.0 * j))
        zs.append(z0)
x, y, z = np.concatenate(xs), np.concatenate(ys), np.concatenate(zs)
mp = capi.default_mls_params()
free0 = torch.cuda.mem_get_info()[0]
res = {"points": int(len(x)), "box_m": [float(x.max() - x.min()), float(y.max() - y.min()), float(z.max() - z.min())]}
with capi.Context(0) as ctx:
    ctx.set_camera(capi.default_camera())
    ctx.upload_cloud(x, y, z)
    cap = 1 << 28
    for rep in range(2):
        t0 = time.perf_counter()
        total, chunks = ctx.mls_stream_begin(mp, cap)
        ctx.synchronize()
        t_b = time.perf_counter() - t0
        t0 = time.perf_counter()
        emitted = 0
        while True:
            m = ctx.mls_stream_next()
            if m == 0:
                break
            emitted += m
        ctx.synchronize()
        t_e = time.perf_counter() - t0
        res["first_call" if rep == 0 else "steady"] = {"fit_stamp_count_s": round(t_b, 3), "emit_s": round(t_e, 3)}
    res.update({"voxels": int(total), "emitted": int(emitted), "chunks": int(chunks),
                "device_memory_used_GB": round((free0 - torch.cuda.mem_get_info()[0]) / 1e9, 2),
                "dense_bitmap_would_be_GB": round(float(np.prod([(a.max() - a.min()) / 0.001 for a in (x, y, z)])) / 8e9, 1),
                "Moutputs_per_s_steady": round(emitted / (res["steady"]["fit_stamp_count_s"] + res["steady"]["emit_s"]) / 1e6, 1)})
print(json.dumps(res))
