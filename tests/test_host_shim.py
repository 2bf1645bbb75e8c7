"""The C++ host shim (pointcloudprocessor_amd/host/pcp_shim.hpp) above the C ABI:
builds with g++, keeps the reference's exit-code convention without a GPU, and on a
GPU box reproduces the oracle on its built-in scene."""
import re
import subprocess

import numpy as np
import pytest


def _exe():
    from pointcloudprocessor_amd import _build, host_build

    _build.build()
    return host_build.build()["pcp_shim_selftest"]


def test_shim_builds_and_maps_errors_to_exit_minus_2():
    import torch

    exe = _exe()
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by the gpu test")
    p = subprocess.run([exe], capture_output=True, text=True)
    assert p.returncode == 254  # == (unsigned char)-2, PCP/src/main.cpp:64-68
    assert "Unhandled Exception reached the top of main" in p.stderr and "no CPU fallback" in p.stderr


def _selftest_scene(n):
    """The scene shim_selftest.cpp builds (same LCG), for the oracle."""
    s = np.uint32(12345)
    pts = np.zeros((n, 3), np.float32)

    def rnd():
        nonlocal s
        s = np.uint32((int(s) * 1664525 + 1013904223) & 0xFFFFFFFF)
        return np.float32(int(s) >> 8) / np.float32(16777216.0)

    for i in range(n):
        strip = rnd() < np.float32(0.3)
        pts[i, 0] = (rnd() - np.float32(0.5)) * np.float32(0.4 if strip else 3.0)
        pts[i, 1] = (rnd() - np.float32(0.5)) * np.float32(1.6)
        pts[i, 2] = np.float32(1.5 if strip else 3.0)
        rnd()
    imgs = []
    for f in range(2):
        i = np.arange(160 * 90 * 3, dtype=np.int64)
        imgs.append(((i * 7 + f * 31) % 251 + 1).astype(np.uint8).reshape(90, 160, 3))
    return pts, imgs


@pytest.mark.gpu
def test_shim_selftest_matches_oracle(oracle):
    n = 2000
    p = subprocess.run([_exe(), str(n)], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    m = re.search(r"kept0 (\d+) coloured (\d+) checksum (\d+) visible1 (\d+)", p.stdout)
    assert m, p.stdout
    kept0, coloured, checksum, visible1 = map(int, m.groups())
    pts, imgs = _selftest_scene(n)
    cam = oracle.default_camera()
    cam.fx = cam.fy = 188.2083
    cam.cx, cam.cy = 80.0, 45.0
    cam.image_width = cam.cull_width = 160
    cam.image_height = cam.cull_height = 90
    cp = oracle.default_cull_params()
    poses = np.array([[0, 0, 0, 1, 0, 0, 0], [0.2, 0, 0, 1, 0, 0, 0]], np.float64)
    x, y, z = pts[:, 0].copy(), pts[:, 1].copy(), pts[:, 2].copy()
    w2c, _ = oracle.pose_to_matrices(poses[0])
    _, _, kept_ref = oracle.cull_frame(cam, cp, w2c, x, y, z)
    ref = oracle.colorize(cam, cp, x, y, z, poses, imgs)
    vis = oracle.frame_visible(cam, cp, poses[1], x, y, z, imgs[1])
    rgb = ref["rgb"].astype(np.int64)
    assert kept0 == kept_ref and coloured == int(ref["has"].sum())
    assert checksum == int((rgb[:, 0] + 3 * rgb[:, 1] + 7 * rgb[:, 2]).sum())
    assert visible1 == len(vis["index"])
    assert 0 < kept_ref < n  # the strip occludes part of the wall


@pytest.mark.gpu
def test_shim_streamed_smoothing_equals_one_shot():
    """pcp_amd::CloudSmooth::processWithOutlierRemovalStreamed (pcp_cloud_smooth_stream_begin / _next: the trailing outlier
    removal over the chunked voxel dilation) hands its sink the rows processWithOutlierRemoval returns, chunk by chunk."""
    p = subprocess.run([_exe(), "4000", "4096"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    m = re.search(r"smooth rows (\d+) streamed (\d+) of (\d+) in (\d+) chunks same (\d)", p.stdout)
    assert m, p.stdout
    rows, kept, total, chunks, same = map(int, m.groups())
    assert same == 1 and kept == rows > 1000 and total > kept and chunks >= 3
