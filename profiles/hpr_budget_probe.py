#!/usr/bin/env python3
"""The whole-run hull pass of C3 (10 M points, 256 keyframes, PCP_CULL_HPR) by the round-trip budget of k_hpr_tilt's 16-lane rows
(PCP_TILT_BUDGET; 0 = searches are never handed on: the form of round 4).  The verdicts of every budget are compared with the
first one's (bit planes read back per keyframe for 8 keyframes).   python3 profiles/hpr_budget_probe.py [budgets...]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pointcloudprocessor_amd import capi, synth

budgets = [int(a) for a in sys.argv[1:]] or [0, 32, 48, 64, 96, 128]
cam = synth.camera_dict("cfg")
x, y, z, _ = synth.make_cloud(10_000_000)
poses, _ = synth.make_trajectory(256)
cull = capi.default_cull_params()
cull.cull_mode = capi.CULL_HPR
out = {}
ref = None
with capi.Context(0) as ctx:
    ctx.set_camera(capi.camera_from_dict(cam), cull)
    ctx.upload_cloud(x, y, z)
    ctx.set_frames(poses)
    for b in budgets:
        os.environ["PCP_TILT_BUDGET"] = str(b)
        ctx.depth_pass()
        ctx.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            ctx.depth_pass()
            ctx.synchronize()
            ts.append(round(time.perf_counter() - t0, 4))
        keeps = [ctx.cull_frame(f)[0].copy() for f in (0, 37, 100, 144, 192, 200, 230, 255)]
        if ref is None:
            ref = keeps
        same = all(np.array_equal(a, b_) for a, b_ in zip(ref, keeps))
        out[str(b)] = {"hull_pass_s": ts, "kept": [int(k.sum()) for k in keeps], "same_verdicts_as_first": bool(same)}
        print(json.dumps({b: out[str(b)]}), file=sys.stderr, flush=True)
print(json.dumps(out))
