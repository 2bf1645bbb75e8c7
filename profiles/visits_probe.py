import ctypes as C, json, os, sys
sys.path.insert(0, os.getcwd())
os.environ["PCP_HIP_LIBRARY"] = os.path.join(os.getcwd(), "pointcloudprocessor_amd/lib/libpcp_hip_dbg.so")
from pointcloudprocessor_amd import capi, synth
cam = synth.camera_dict("cfg")
x, y, z, _ = synth.make_cloud(10_000_000)
poses, _ = synth.make_trajectory(256)
ctx = capi.Context(0)
ctx.set_camera(capi.camera_from_dict(cam))
ctx.upload_cloud(x, y, z)
ctx.set_frames(poses)
img = synth.make_image(0, cam["image_width"], cam["image_height"])
for f in range(256):
    ctx.upload_image(f, img)
ctx.colorize(download=False)
ctx.synchronize()
out = (C.c_ulonglong * 4)()
capi.load().pcp_debug_visits(out)
v = list(out)
print(json.dumps({"visits": v[0], "visits_without_a_kept_lane": v[1], "share": round(v[1] / v[0], 4), "candidates": v[2], "kept": v[3],
                  "kept_share": round(v[3] / v[2], 4), "lanes_per_visit": round(v[2] / v[0], 1)}))
