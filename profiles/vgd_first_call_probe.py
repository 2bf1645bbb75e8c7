import os, sys, time, json
sys.path.insert(0, os.getcwd())
import numpy as np
from pointcloudprocessor_amd import capi, synth
warm = len(sys.argv) > 1 and sys.argv[1] == "warm"
x, y, z, _ = synth.make_cloud(10_000_000)
ctx = capi.Context(0)
ctx.set_camera(capi.default_camera())
if warm:
    os.environ["PCP_SOR_THREE_DESCRIPTORS"] = "1"
    ctx.upload_cloud(x[:200000].copy(), y[:200000].copy(), z[:200000].copy())
    ctx.sor(60, 0.7)
    ctx.synchronize()
    os.environ["PCP_SOR_THREE_DESCRIPTORS"] = "0"
ctx.upload_cloud(x[::10].copy(), y[::10].copy(), z[::10].copy())
vp = capi.default_mls_params()
runs = []
for _ in range(3):
    t = time.perf_counter(); m = ctx.cloud_smooth(vp); ctx.synchronize(); runs.append(round((time.perf_counter() - t) * 1e3, 1))
print(json.dumps({"warm": warm, "runs_ms": runs, "outputs": int(m)}))
