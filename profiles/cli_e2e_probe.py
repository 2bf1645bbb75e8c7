#!/usr/bin/env python3
"""Where the command line's time goes when the per-keyframe dumps are skipped: the same dataset, forms and decoder threads
varied.  python3 profiles/cli_e2e_probe.py"""
import json, os, shutil, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from PIL import Image
from pointcloudprocessor_amd import host_build, synth

exe = host_build.build()["PointCloudProcessor"]
d = tempfile.mkdtemp(prefix="pcp_cli_probe_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
n, F, W, H = 1_000_000, 32, 1920, 1080
x, y, z, inten = synth.make_cloud(n)
hdr = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\n"
       f"WIDTH {n}\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS {n}\nDATA binary\n")
with open(os.path.join(d, "scans.pcd"), "wb") as fh:
    fh.write(hdr.encode()); fh.write(np.stack([x, y, z, inten], 1).astype(np.float32).tobytes())
poses, ts = synth.make_trajectory(F)
with open(os.path.join(d, "odo.txt"), "w") as fh:
    for k, (t, p) in enumerate(zip(ts, poses)):
        fh.write(synth.odometry_line(t, p))
        Image.fromarray(synth.make_image(k, W, H)[:, :, ::-1]).save(os.path.join(d, "%f.jpg" % t), quality=92)
res = []
k = 0
for skip, env_extra in ((1, {}), (0, {}), (1, {}), (1, {"PCP_DECODE_THREADS": "4"}), (1, {"PCP_DECODE_THREADS": "1"}), (0, {"PCP_DECODE_THREADS": "4"}),
                        (1, {"PCP_UPLOAD_DIRECT": "0"}), (1, {})):
    out = os.path.join(d, f"o{k}") + "/"; k += 1
    os.makedirs(out)
    env = dict(os.environ, PCP_CLI_TIMING=out + "t.json", **env_extra)
    t0 = time.perf_counter()
    p = subprocess.run([exe, "-p", d + "/scans.pcd", "-o", d + "/odo.txt", "-i", d + "/", "-t", out, "--skip_filtered_dumps", str(skip)],
                       capture_output=True, text=True, env=env, cwd=out)
    wall = time.perf_counter() - t0
    ph = json.load(open(out + "t.json")) if p.returncode == 0 else {"error": p.stderr[-200:]}
    res.append({"skip": skip, "env": env_extra, "wall": round(wall, 3), **{a: round(b, 3) for a, b in ph.items() if isinstance(b, float)}})
    print(json.dumps(res[-1]), flush=True)
    shutil.rmtree(out, ignore_errors=True)
shutil.rmtree(d, ignore_errors=True)
