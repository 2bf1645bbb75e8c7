#!/usr/bin/env python3
"""MLS alone (upsampling NONE, r = 0.03, order 2) on the 10 M-point C3 map, three runs: the workload of the `mls` leg of
bench.py, for the PMC passes of collect_mls.sh.  python profiles/mls_probe.py [points]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401

from pointcloudprocessor_amd import capi, synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
x, y, z, _ = synth.make_cloud(N)
ctx = capi.Context(0)
ctx.set_camera(capi.default_camera())
ctx.upload_cloud(x, y, z)
mp = capi.default_mls_params()
mp.upsampling = 0
ctx.mls_process(mp)
ctx.synchronize()
ts = []
for _ in range(3):
    t = time.perf_counter()
    m = ctx.mls_process(mp)
    ctx.synchronize()
    ts.append((time.perf_counter() - t) * 1e3)
print(json.dumps({"points": N, "outputs": int(m), "ms": [round(v, 2) for v in ts]}))
