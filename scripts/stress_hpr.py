#!/usr/bin/env python3
"""Randomised stress of PCP_CULL_HPR (not collected by pytest): seeds x cloud sizes x cameras x keyframes x flip radii, the
GPU hull (quick certificate, radial pre-pass, search, exact path) against the oracle's exact quickhull; also with the two
passes switched off.  Prints one line per case; exit 1 on the first mismatch.   python scripts/stress_hpr.py [cases]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle_capi as oc  # noqa: E402
from pointcloudprocessor_amd import capi, synth  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    oc.build()
    bad = 0
    for case in range(cases):
        rng = np.random.default_rng(7000 + case)
        camname = ["tiny", "cfg", "ref"][case % 3]
        cam = synth.camera_dict(camname)
        n = int(rng.choice([30_000, 120_000, 400_000, 1_000_000]))
        x, y, z, _ = synth.make_cloud(n, seed=100 + case)
        if case % 5 == 4:  # exact duplicates and a coarse lattice (degenerate flips): the duplicate rule, the exact path
            x[: n // 50] = x[n // 50: 2 * (n // 50)]
            y[: n // 50] = y[n // 50: 2 * (n // 50)]
            z[: n // 50] = z[n // 50: 2 * (n // 50)]
            x, y, z = (np.round(v * 64) / 64 for v in (x, y, z))
            x, y, z = x.astype(np.float32), y.astype(np.float32), z.astype(np.float32)
        poses, _ = synth.make_trajectory(64)
        radius = float(rng.choice([90000.0, 90000.0, 3000.0, 1.0e6]))
        ocam = oc.Camera()
        for k, _t in oc.Camera._fields_:
            setattr(ocam, k, cam[k])
        cull = capi.default_cull_params()
        cull.cull_mode = capi.CULL_HPR
        cull.hpr_flip_radius = radius
        frames = [int(f) for f in rng.choice(64, 3, replace=False)]
        refs = {}
        for f in frames:
            w2c, _ = oc.pose_to_matrices(poses[f])
            refs[f], _ = oc.hpr_frame(ocam, w2c, x, y, z, radius)
        for variant in (0, 1):
            os.environ["PCP_HPR_QUICK"] = "1" if variant == 0 else "0"
            os.environ["PCP_HPR_RADIAL"] = "1" if variant == 0 else "0"
            with capi.Context(0) as ctx:
                ctx.set_camera(capi.camera_from_dict(cam), cull)
                ctx.upload_cloud(x, y, z)
                ctx.set_frames(poses)
                for f in frames:
                    keep, _, kept = ctx.cull_frame(f)
                    st = ctx.hpr_stats()
                    ok = np.array_equal(keep, refs[f])
                    print(f"case {case:3d} cam={camname:4s} n={n:8d} R={radius:9.0f} kf={f:3d} passes={'on ' if variant == 0 else 'off'} "
                          f"candidates={st['candidates']:8d} kept={kept:8d} exact={st['exact_path']:4d} unresolved={st['unresolved']} "
                          f"{'ok' if ok else 'MISMATCH'}", flush=True)
                    if not ok:
                        bad += 1
    print("all cases identical" if bad == 0 else f"{bad} MISMATCHES")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
