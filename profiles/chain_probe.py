#!/usr/bin/env python3
"""enableMLS=1 chain (SOR -> MLS -> SOR, cloudSmooth.cpp:109-164) at 10 M points: wall time and per-kernel-group
times.  python profiles/chain_probe.py [points]"""
import os
import json
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
ctx.cloud_smooth(mp)
ctx.synchronize()
res = {"points": N}
ts = []
for _ in range(3):
    t = time.perf_counter()
    m = ctx.cloud_smooth(mp)
    ctx.synchronize()
    ts.append((time.perf_counter() - t) * 1e3)
res["chain_ms"] = [round(v, 2) for v in ts]
res["outputs"] = int(m)
if len(sys.argv) > 2 and sys.argv[2] == "chain-only":  # (profiles/chain_timeline.sh: the trace ends with a chain)
    print(json.dumps(res))
    sys.exit(0)
ctx.timing_enable(True)
ctx.timing_reset()
ctx.cloud_smooth(mp)
ctx.synchronize()
res["kernels_ms"] = {ctx.kernel_name(k): round(ctx.timing_get(k)[0], 3) for k in (capi.K_SOR, capi.K_MLS_GRID, capi.K_MLS_FIT, capi.K_MISC)}
res["sor_heap_fraction"] = round(ctx.sor_redo_fraction(), 5)
ctx.timing_enable(False)
t = time.perf_counter()
ctx.mls_process(mp)
ctx.synchronize()
t = time.perf_counter()
ctx.mls_process(mp)
ctx.synchronize()
res["mls_alone_ms"] = round((time.perf_counter() - t) * 1e3, 2)
print(json.dumps(res))
