"""ctypes view of oracle/libpcp_oracle.so (the CPU restatement).

TEST INFRASTRUCTURE ONLY -- see oracle/pcp_oracle.h.  Importable from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; never from the
product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpcp_oracle.so")


class Pose(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("x", "y", "z", "qw", "qx", "qy", "qz")]


class Camera(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "k3")] + [
        (k, C.c_int32) for k in ("image_width", "image_height", "cull_width", "cull_height")
    ]


class CullParams(C.Structure):
    _fields_ = [
        ("enable_depth_buffer_culling", C.c_int32),
        ("downsample_factor", C.c_int32),
        ("depth_slack", C.c_double),
        ("cull_mode", C.c_int32),   # 0 z-buffer, 1 hidden_points_removal's candidate filter, 2 hidden_points_removal
        ("match_mode", C.c_int32),  # 0 identity, 1 fp32 world round trip + self-match
        ("hpr_flip_radius", C.c_double),
    ]


CULL_ZBUFFER, CULL_HPR_CANDIDATES, CULL_HPR = 0, 1, 2
MATCH_IDENTITY, MATCH_ROUNDTRIP = 0, 1


class MLSParams(C.Structure):
    _fields_ = [
        ("search_radius", C.c_double),
        ("sqr_gauss_param", C.c_double),
        ("polynomial_order", C.c_int32),
        ("compute_normals", C.c_int32),
        ("upsampling", C.c_int32),
        ("vgd_iterations", C.c_int32),
        ("vgd_voxel_size", C.c_float),
        ("threads", C.c_int32),
    ]


def build(force: bool = False) -> str:
    """Compile the restatement with the committed Makefile (gcc, -ffp-contract=off)."""
    srcs = [os.path.join(_HERE, f) for f in ("pcp_oracle.c", "pcp_oracle_mls.c", "pcp_oracle_nid.c", "pcp_oracle_hpr.c",
                                            "pcp_oracle.h", "Makefile")]
    stale = force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs
    )
    if stale:
        subprocess.run(["make", "-C", _HERE, "libpcp_oracle.so"], check=True, capture_output=True)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        # PCP_ORACLE_LIBRARY: e.g. the sanitizer build (make -C oracle libpcp_oracle_asan.so)
        L = C.CDLL(os.environ.get("PCP_ORACLE_LIBRARY") or _LIB_PATH)
        fp = C.POINTER(C.c_float)
        L.orc_cull_frame.restype = C.c_int64
        L.orc_frame_visible.restype = C.c_int64
        L.orc_mls.restype = C.c_int64
        L.orc_mls_voxel_dilation.restype = C.c_int64
        L.orc_mls_voxel_dilation_part.restype = C.c_int64
        L.orc_sor.restype = C.c_int64
        L.orc_select_keyframes.restype = C.c_int32
        L.orc_hardware_threads.restype = C.c_int32
        L.orc_colorize.restype = C.c_int
        L.orc_colorize_faithful.restype = C.c_int
        L.orc_affine_inverse_f32.restype = None
        L.orc_convex_hull_vertices.restype = C.c_int64
        L.orc_hpr_frame.restype = C.c_int64
        L.orc_orient3d.restype = C.c_int
        L.orc_hpr_flip.restype = None
        _ = fp
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t=None):
    if a is None:
        return None
    return a.ctypes.data_as(C.c_void_p)


def default_camera(width: int | None = None, height: int | None = None) -> Camera:
    cam = Camera()
    lib().orc_default_camera(C.byref(cam))
    if width is not None:
        cam.image_width = cam.cull_width = int(width)
    if height is not None:
        cam.image_height = cam.cull_height = int(height)
    return cam


def default_cull_params() -> CullParams:
    p = CullParams()
    lib().orc_default_cull_params(C.byref(p))
    return p


def default_mls_params() -> MLSParams:
    p = MLSParams()
    lib().orc_default_mls_params(C.byref(p))
    return p


def make_pose(v) -> Pose:
    return Pose(*[float(t) for t in v])


def poses_array(poses: np.ndarray):
    poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 7)
    arr = (Pose * len(poses))()
    C.memmove(arr, poses.ctypes.data, poses.nbytes)
    return arr


def pose_to_matrices(pose, T_opt=None):
    w2c = np.zeros(12, np.float32)
    c2w = np.zeros(12, np.float32)
    p = make_pose(pose)
    T = None if T_opt is None else np.ascontiguousarray(T_opt, np.float64).reshape(16)
    lib().orc_pose_to_matrices(C.byref(p), _p(T), _p(w2c), _p(c2w))
    return w2c, c2w


def project_frame(cam, cp, w2c, x, y, z):
    x, y, z = _f32(x), _f32(y), _f32(z)
    n = len(x)
    cell = np.empty(n, np.int32)
    pix = np.empty(n, np.int32)
    rng = np.empty(n, np.float32)
    xc = np.empty(n, np.float32)
    yc = np.empty(n, np.float32)
    zc = np.empty(n, np.float32)
    w2c = _f32(w2c)
    lib().orc_project_frame(C.byref(cam), C.byref(cp), _p(w2c), _p(x), _p(y), _p(z), C.c_int64(n), _p(cell),
                            _p(pix), _p(rng), _p(xc), _p(yc), _p(zc))
    return dict(cell=cell, pixel=pix, range=rng, xc=xc, yc=yc, zc=zc)


def cull_frame(cam, cp, w2c, x, y, z, threads: int = 1):
    x, y, z = _f32(x), _f32(y), _f32(z)
    n = len(x)
    mw = cam.cull_width // cp.downsample_factor
    mh = cam.cull_height // cp.downsample_factor
    keep = np.empty(n, np.uint8)
    dmap = np.empty(mw * mh, np.float32)
    w2c = _f32(w2c)
    kept = lib().orc_cull_frame(C.byref(cam), C.byref(cp), _p(w2c), _p(x), _p(y), _p(z), C.c_int64(n), _p(keep),
                                _p(dmap), C.c_int32(threads))
    return keep, dmap.reshape(mh, mw), int(kept)


def orient3d(a, b, c, d, exact_only: bool = False) -> int:
    """Sign of det [a-d; b-d; c-d] (> 0: d below the plane through a, b, c counter-clockwise from above), decided
    exactly; exact_only skips the floating-point filter."""
    a, b, c, d = (np.ascontiguousarray(v, np.float64).reshape(3) for v in (a, b, c, d))
    return int(lib().orc_orient3d(_p(a), _p(b), _p(c), _p(d), C.c_int32(1 if exact_only else 0)))


def hpr_flip(xc, yc, zc, flip_radius: float = 90000.0):
    """view_culling.cpp:291-292 on fp32 camera coordinates -> (m, 3) fp64 flipped points."""
    xc, yc, zc = _f32(xc), _f32(yc), _f32(zc)
    out = np.empty((len(xc), 3), np.float64)
    lib().orc_hpr_flip(_p(xc), _p(yc), _p(zc), C.c_int64(len(xc)), C.c_double(flip_radius), _p(out))
    return out


def convex_hull_vertices(points):
    """Exact extreme points of an (n, 3) fp64 point set: (is_vertex uint8[n], count or -1 for flat / too few points,
    stats dict)."""
    pts = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
    is_v = np.zeros(len(pts), np.uint8)
    st = np.zeros(4, np.int64)
    nv = lib().orc_convex_hull_vertices(_p(pts), C.c_int64(len(pts)), _p(is_v), _p(st))
    return is_v, int(nv), dict(filtered=int(st[0]), exact=int(st[1]), zero=int(st[2]), duplicates=int(st[3]))


def hpr_frame(cam, w2c, x, y, z, flip_radius: float = 90000.0):
    """ViewCulling::hidden_points_removal (view_culling.cpp:266-334) for one keyframe: keep mask in input order, stats."""
    x, y, z = _f32(x), _f32(y), _f32(z)
    n = len(x)
    keep = np.zeros(n, np.uint8)
    st = np.zeros(5, np.int64)
    w2c = _f32(w2c)
    kept = lib().orc_hpr_frame(C.byref(cam), _p(w2c), _p(x), _p(y), _p(z), C.c_int64(n), C.c_double(flip_radius), _p(keep),
                               _p(st))
    if kept < 0:
        raise MemoryError("orc_hpr_frame")
    return keep, dict(candidates=int(st[0]), filtered=int(st[1]), exact=int(st[2]), zero=int(st[3]), duplicates=int(st[4]),
                      kept=int(kept))


def scores(xc, yc, zc, pose):
    o = C.c_float()
    d = C.c_float()
    f = C.c_float()
    p = make_pose(pose)
    lib().orc_scores(C.c_float(xc), C.c_float(yc), C.c_float(zc), C.byref(p), C.byref(o), C.byref(d), C.byref(f))
    return o.value, d.value, f.value


def colorize(cam, cp, x, y, z, poses, images, T_opt=None, threads: int = 1, want_top: bool = True):
    x, y, z = _f32(x), _f32(y), _f32(z)
    n = len(x)
    poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 7)
    F = len(poses)
    parr = poses_array(poses)
    imgs = [np.ascontiguousarray(im, np.uint8) for im in images]
    assert len(imgs) == F
    for im in imgs:
        assert im.shape == (cam.image_height, cam.image_width, 3), im.shape
    iptr = (C.c_void_p * F)(*[im.ctypes.data for im in imgs])
    rgb = np.zeros((n, 3), np.uint8)
    has = np.zeros(n, np.uint8)
    cnt = np.zeros(n, np.int32)
    ts = np.zeros((n, 5), np.float32) if want_top else None
    tr = np.zeros((n, 5), np.uint32) if want_top else None
    tf = np.zeros((n, 5), np.int32) if want_top else None
    T = None
    stride = 0
    if T_opt is not None:
        T = np.ascontiguousarray(T_opt, np.float64)
        stride = 16 if T.size == 16 * F and F > 1 else 0
        T = T.reshape(-1)
    rc = lib().orc_colorize(C.byref(cam), C.byref(cp), _p(x), _p(y), _p(z), C.c_int64(n), parr, C.c_int32(F),
                            _p(T), C.c_int32(stride), iptr, _p(rgb), _p(has), _p(cnt), _p(ts), _p(tr), _p(tf),
                            C.c_int32(threads))
    if rc != 0:
        raise RuntimeError("orc_colorize failed")
    return dict(rgb=rgb, has=has, count=cnt, top_score=ts, top_rgb=tr, top_frame=tf)


def colorize_faithful(cam, cp, x, y, z, poses, images, T_opt=None, threads: int = 1):
    """The reference's own match-back (Appendix B3): fp32 world round trip, radiusSearch(1e-5) over the original
    cloud, scores from c2w.inverse() * p_w.  Adds `stats` = dict(samples, unmatched, self_missed, cross_credits)."""
    x, y, z = _f32(x), _f32(y), _f32(z)
    n = len(x)
    poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 7)
    F = len(poses)
    parr = poses_array(poses)
    imgs = [np.ascontiguousarray(im, np.uint8) for im in images]
    assert len(imgs) == F
    iptr = (C.c_void_p * F)(*[im.ctypes.data for im in imgs])
    rgb = np.zeros((n, 3), np.uint8)
    has = np.zeros(n, np.uint8)
    cnt = np.zeros(n, np.int32)
    ts = np.zeros((n, 5), np.float32)
    tr = np.zeros((n, 5), np.uint32)
    tf = np.zeros((n, 5), np.int32)
    st = np.zeros(4, np.int64)
    T = None
    stride = 0
    if T_opt is not None:
        T = np.ascontiguousarray(T_opt, np.float64)
        stride = 16 if T.size == 16 * F and F > 1 else 0
        T = T.reshape(-1)
    rc = lib().orc_colorize_faithful(C.byref(cam), C.byref(cp), _p(x), _p(y), _p(z), C.c_int64(n), parr, C.c_int32(F),
                                     _p(T), C.c_int32(stride), iptr, _p(rgb), _p(has), _p(cnt), _p(ts), _p(tr), _p(tf),
                                     _p(st), C.c_int32(threads))
    if rc != 0:
        raise RuntimeError("orc_colorize_faithful failed")
    return dict(rgb=rgb, has=has, count=cnt, top_score=ts, top_rgb=tr, top_frame=tf,
                stats=dict(samples=int(st[0]), unmatched=int(st[1]), self_missed=int(st[2]), cross_credits=int(st[3])))


def affine_inverse(m):
    m = _f32(m).reshape(12)
    out = np.zeros(12, np.float32)
    lib().orc_affine_inverse_f32(_p(m), _p(out))
    return out


def hsv_round_trip(bgr, saturation_scale: float = 1.0, brightness_scale: float = 1.0):
    """generateColorMap's 8-bit BGR -> HSV -> BGR round trip (PointCloudProcessor.cpp:722-741) of an (..., 3) image."""
    a = np.ascontiguousarray(bgr, np.uint8)
    out = np.empty_like(a)
    lib().orc_hsv_round_trip.restype = None
    lib().orc_hsv_round_trip(_p(a), _p(out), C.c_int64(a.size // 3), C.c_float(saturation_scale),
                             C.c_float(brightness_scale))
    return out


def frame_visible(cam, cp, pose, x, y, z, image, mask=None, T_opt=None):
    x, y, z = _f32(x), _f32(y), _f32(z)
    n = len(x)
    idx = np.empty(n, np.int32)
    rgb = np.empty((n, 3), np.uint8)
    mv = np.empty(n, np.uint16)
    cam_xyz = np.empty((n, 3), np.float32)
    wrd_xyz = np.empty((n, 3), np.float32)
    p = make_pose(pose)
    image = None if image is None else np.ascontiguousarray(image, np.uint8)
    mask = None if mask is None else np.ascontiguousarray(mask, np.uint8)
    T = None if T_opt is None else np.ascontiguousarray(T_opt, np.float64).reshape(16)
    m = lib().orc_frame_visible(C.byref(cam), C.byref(cp), C.byref(p), _p(T), _p(x), _p(y), _p(z), C.c_int64(n),
                                _p(image), _p(mask), _p(idx), _p(rgb), _p(mv), _p(cam_xyz), _p(wrd_xyz))
    m = int(m)
    return dict(index=idx[:m], rgb=rgb[:m], mask=mv[:m], xyz_cam=cam_xyz[:m], xyz_world=wrd_xyz[:m])


def mls(x, y, z, params: MLSParams):
    x, y, z = _f32(x), _f32(y), _f32(z)
    n = len(x)
    xyz = np.empty((n, 3), np.float32)
    nrm = np.empty((n, 3), np.float32)
    curv = np.empty(n, np.float32)
    idx = np.empty(n, np.int32)
    m = int(lib().orc_mls(_p(x), _p(y), _p(z), C.c_int64(n), C.byref(params), _p(xyz), _p(nrm), _p(curv), _p(idx)))
    if m < 0:
        raise RuntimeError("orc_mls failed")
    return dict(xyz=xyz[:m], normal=nrm[:m], curvature=curv[:m], index=idx[:m])


def mls_voxel_dilation(x, y, z, params: MLSParams):
    x, y, z = _f32(x), _f32(y), _f32(z)
    n = len(x)
    m = int(lib().orc_mls_voxel_dilation(_p(x), _p(y), _p(z), C.c_int64(n), C.byref(params), C.c_int64(0), None,
                                         None, None, None))
    if m < 0:
        raise RuntimeError("orc_mls_voxel_dilation failed")
    xyz = np.empty((m, 3), np.float32)
    nrm = np.empty((m, 3), np.float32)
    curv = np.empty(m, np.float32)
    idx = np.empty(m, np.int32)
    lib().orc_mls_voxel_dilation(_p(x), _p(y), _p(z), C.c_int64(n), C.byref(params), C.c_int64(m), _p(xyz),
                                 _p(nrm), _p(curv), _p(idx))
    return dict(xyz=xyz, normal=nrm, curvature=curv, index=idx)


def mls_voxel_dilation_part(x, y, z, params: MLSParams, origin, extent: float):
    """mls_voxel_dilation of a REGION of a larger cloud on that cloud's voxel lattice: origin = the cloud's bounding minimum
    (3 floats), extent = its largest extent."""
    x, y, z = _f32(x), _f32(y), _f32(z)
    n = len(x)
    org = _f32(np.asarray(origin, np.float32))
    args = (_p(x), _p(y), _p(z), C.c_int64(n), C.byref(params), _p(org), C.c_double(extent))
    m = int(lib().orc_mls_voxel_dilation_part(*args, C.c_int64(0), None, None, None, None))
    if m < 0:
        raise RuntimeError("orc_mls_voxel_dilation_part failed")
    xyz = np.empty((m, 3), np.float32)
    nrm = np.empty((m, 3), np.float32)
    curv = np.empty(m, np.float32)
    idx = np.empty(m, np.int32)
    lib().orc_mls_voxel_dilation_part(*args, C.c_int64(m), _p(xyz), _p(nrm), _p(curv), _p(idx))
    return dict(xyz=xyz, normal=nrm, curvature=curv, index=idx)


def sor(x, y, z, mean_k: int = 60, std_mul: float = 0.7, threads: int = 0, details: bool = False):
    x, y, z = _f32(x), _f32(y), _f32(z)
    n = len(x)
    keep = np.empty(n, np.uint8)
    dist = np.empty(n, np.float32)
    thr = C.c_double()
    kept = int(lib().orc_sor(_p(x), _p(y), _p(z), C.c_int64(n), C.c_int32(mean_k), C.c_double(std_mul), _p(keep),
                             _p(dist), C.byref(thr), C.c_int32(threads)))
    if details:
        return keep, kept, dist, thr.value
    return keep, kept


def nid(cam, images, offsets, x, y, z, intensity, T, bins: int = 16):
    """MultiNIDCost value and SE(3)-tangent gradient (6,) at T (4x4)."""
    x, y, z, intensity = _f32(x), _f32(y), _f32(z), _f32(intensity)
    offsets = np.ascontiguousarray(offsets, np.int64)
    F = len(offsets) - 1
    imgs = [np.ascontiguousarray(im, np.uint8) for im in images]
    iptr = (C.c_void_p * F)(*[im.ctypes.data for im in imgs])
    T = np.ascontiguousarray(T, np.float64).reshape(16)
    cost = C.c_double()
    grad = np.zeros(6, np.float64)
    ok = lib().orc_nid(C.byref(cam), iptr, C.c_int32(F), _p(offsets), _p(x), _p(y), _p(z), _p(intensity), _p(T),
                       C.c_int32(bins), C.byref(cost), _p(grad))
    return cost.value, grad, bool(ok)


def select_keyframes(poses, dist_threshold: float = 0.1):
    poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 7)
    parr = poses_array(poses)
    out = np.empty(len(poses), np.int32)
    m = lib().orc_select_keyframes(parr, C.c_int32(len(poses)), C.c_double(dist_threshold), _p(out))
    return out[:m]


def hardware_threads() -> int:
    """Host cores this process may actually use: min(affinity mask, cgroup cpu quota)."""
    import math

    n = int(lib().orc_hardware_threads())
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, math.ceil(int(parts[0]) / int(parts[1]))))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                        n = min(n, max(1, math.ceil(q / int(g.read().split()[0]))))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)
