#!/usr/bin/env python3
"""The streamed chain (pcp_cloud_smooth_stream_*) on the whole C3 map: begin / emit seconds and its diagnostics.
python3 profiles/css_probe.py [capacity_log2] [calls]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudprocessor_amd import capi, synth
cap = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 28)
x, y, z, _ = synth.make_cloud(10_000_000)
ctx = capi.Context(0)
ctx.set_camera(capi.default_camera())
ctx.upload_cloud(x, y, z)
vp = capi.default_mls_params()
out = []
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
for rep in range(reps):
    t = time.perf_counter(); rows, kept, chunks = ctx.cloud_smooth_stream_begin(vp, cap); ctx.synchronize(); tb = time.perf_counter() - t
    t = time.perf_counter(); got = 0
    while True:
        m = ctx.cloud_smooth_stream_next()
        if m == 0: break
        got += m
    ctx.synchronize(); te = time.perf_counter() - t
    out.append({"begin_s": round(tb, 3), "emit_s": round(te, 3), "rows": rows, "kept": kept, "got": got, "chunks": chunks, **ctx.cloud_smooth_stream_stats()})
print(json.dumps(out))
