#!/usr/bin/env python3
"""The trailing StatisticalOutlierRemoval of the reference's chain on the dilated cloud (1 M input points -> 425 M rows): share of
rows the selection kernel hands on, kernel-group times.  python3 profiles/vgd_sor_probe.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudprocessor_amd import capi, synth
x, y, z, _ = synth.make_cloud(10_000_000)
ctx = capi.Context(0)
ctx.set_camera(capi.default_camera())
ctx.upload_cloud(x[::10], y[::10], z[::10])
vp = capi.default_mls_params()
ctx.cloud_smooth(vp); ctx.synchronize()
ctx.timing_enable(True); ctx.timing_reset()
t = time.perf_counter(); m = ctx.cloud_smooth(vp); ctx.synchronize(); t = time.perf_counter() - t
res = {"outputs": int(m), "ms": round(t * 1e3, 1), "sor_redo_fraction_last": ctx.sor_redo_fraction(),
       "kernels_ms": {ctx.kernel_name(k): round(ctx.timing_get(k)[0], 3) for k in (capi.K_SOR, capi.K_MLS_GRID, capi.K_MLS_FIT, capi.K_MLS_VOXEL, capi.K_MISC)}}
print(json.dumps(res))
