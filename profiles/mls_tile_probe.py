#!/usr/bin/env python3
"""k_mls_fit, tile form against gather form (PCP_MLS_TILE=0) on the 10 M-point C3 map: results compared bit for bit, kernel
times from the library's own events.  python profiles/mls_tile_probe.py [points]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from pointcloudprocessor_amd import capi, synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
x, y, z, _ = synth.make_cloud(N)
ctx = capi.Context(0)
ctx.set_camera(capi.default_camera())
ctx.upload_cloud(x, y, z)
mp = capi.default_mls_params()
mp.upsampling = 0
res = {}
out = {}
for mode in ("0", "1"):
    os.environ["PCP_MLS_TILE"] = mode
    ctx.mls_process(mp)
    ctx.synchronize()
    ctx.timing_enable(True)
    ts, ks = [], []
    for _ in range(3):
        ctx.timing_reset()
        t = time.perf_counter()
        m = ctx.mls_process(mp)
        ctx.synchronize()
        ts.append(round((time.perf_counter() - t) * 1e3, 2))
        ks.append(round(ctx.timing_get(capi.K_MLS_FIT)[0], 3))
    ctx.timing_enable(False)
    out[mode] = ctx.mls_fetch(int(m))
    res["tile" if mode == "1" else "gather"] = {"ms": ts, "fit_ms": ks, "outputs": int(m)}
same = all(np.array_equal(out["0"][k].view(np.uint8), out["1"][k].view(np.uint8)) for k in out["0"])
res["identical"] = bool(same)
print(json.dumps(res))
