#!/usr/bin/env python3
"""The reference's own MLS configuration (VOXEL_GRID_DILATION 1 mm x 4, PointCloudProcessor.cpp:78-81) on the whole C3 map:
pcp_mls_stream_begin (fit + stamp + count) and the chunked emission, for a rocprofv3 kernel trace.
python3 profiles/vgd_stream_probe.py"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudprocessor_amd import capi, synth

x, y, z, _ = synth.make_cloud(10_000_000)
ctx = capi.Context(0)
ctx.set_camera(capi.default_camera())
ctx.upload_cloud(x, y, z)
vp = capi.default_mls_params()
res = {}
for rep in range(2):
    t = time.perf_counter()
    total, chunks = ctx.mls_stream_begin(vp, 1 << 28)
    ctx.synchronize()
    tb = time.perf_counter() - t
    t = time.perf_counter()
    emitted = 0
    while True:
        m = ctx.mls_stream_next()
        if m == 0:
            break
        emitted += m
    ctx.synchronize()
    res[f"run{rep}"] = {"begin_ms": round(tb * 1e3, 1), "emit_ms": round((time.perf_counter() - t) * 1e3, 1), "voxels": int(total),
                        "chunks": int(chunks)}
print(json.dumps(res))
