#!/usr/bin/env python3
"""k_mls_fit with different prefetch depths (builds under variants/, chosen by PCP_HIP_LIBRARY in a child process each): MLS
alone and the enableMLS chain on the 10 M-point C3 map, results hashed.  python profiles/mls_ahead_probe.py lib.so ..."""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, time, json, hashlib
sys.path.insert(0, %r)
import numpy as np
from pointcloudprocessor_amd import capi, synth
x, y, z, _ = synth.make_cloud(10_000_000)
ctx = capi.Context(0)
ctx.set_camera(capi.default_camera())
ctx.upload_cloud(x, y, z)
mp = capi.default_mls_params()
mp.upsampling = 0
ctx.mls_process(mp); ctx.synchronize()
ctx.timing_enable(True)
ts, ks = [], []
for _ in range(4):
    ctx.timing_reset()
    t = time.perf_counter(); m = ctx.mls_process(mp); ctx.synchronize()
    ts.append(round((time.perf_counter() - t) * 1e3, 2)); ks.append(round(ctx.timing_get(capi.K_MLS_FIT)[0], 3))
out = ctx.mls_fetch(int(m))
h = hashlib.sha256()
for k in ("xyz", "normal", "curvature", "index"):
    h.update(out[k].tobytes())
cs = []
ctx.timing_enable(False)
for _ in range(3):
    t = time.perf_counter(); mc = ctx.cloud_smooth(mp); ctx.synchronize()
    cs.append(round((time.perf_counter() - t) * 1e3, 2))
print(json.dumps({"mls_ms": ts, "fit_ms": ks, "chain_ms": cs, "chain_outputs": int(mc), "sha": h.hexdigest()[:16]}))
''' % ROOT
for lib in sys.argv[1:]:
    env = dict(os.environ, PCP_HIP_LIBRARY=os.path.abspath(lib))
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
    print(os.path.basename(lib), r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-500:], flush=True)
