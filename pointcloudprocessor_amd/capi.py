"""ctypes binding of include/pcp_hip.h (libpcp_hip.so).

This is the Python view of the drop-in boundary used by tests and bench.py; the
C++ host shim (pointcloudprocessor_amd/host/) binds the same symbols.  There is
no fallback: if the library is missing or no GPU is usable, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
import re

import numpy as np

from . import _build

# error classes (pcp_hip.h)
PCP_OK = 0
PCP_ERR_INVALID = -1
PCP_ERR_STATE = -2
PCP_ERR_DEVICE = -3
PCP_ERR_NOMEM = -4
PCP_ERR_RANGE = -5

NID_EVAL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double),
                          C.POINTER(C.c_int32))
K_PROJECT, K_DEPTH, K_COLOUR, K_VISIBILITY, K_MLS_GRID, K_MLS_FIT, K_MISC, K_SOR, K_MLS_VOXEL, K_TILE_MASK, K_NID, K_HPR = range(12)
K_COUNT = 12


class PcpError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"pcp error {code}: {msg}")
        self.code = code


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
        ("cull_mode", C.c_int32),   # CULL_ZBUFFER / CULL_HPR_CANDIDATES / CULL_HPR
        ("match_mode", C.c_int32),  # MATCH_IDENTITY / MATCH_ROUNDTRIP
        ("hpr_flip_radius", C.c_double),  # hidden_points_removal_max_z, view_culling.hpp:14
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
        ("sor_mean_k", C.c_int32),
        ("sor_std_mul", C.c_double),
    ]


def declared_symbols() -> list[str]:
    """Every function include/pcp_hip.h declares (parsed from the header)."""
    with open(os.path.join(_build.INCLUDE, "pcp_hip.h")) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pcp_[a-z0-9_]+)\s*\(", text)))


_lib = None


def load(build_if_needed: bool = True) -> C.CDLL:
    """dlopen libpcp_hip.so (building it in-tree first when stale)."""
    global _lib
    if _lib is not None:
        return _lib
    if build_if_needed:
        try:
            _build.build()
        except Exception:
            if not os.path.exists(_build.LIB_PATH):
                raise
    if not os.path.exists(_build.LIB_PATH):
        raise PcpError(PCP_ERR_DEVICE, f"{_build.LIB_PATH} is missing: the HIP extension is not built")
    # One HIP runtime per process: torch ships its own libamdhip64.so.7 and must initialise it
    # before another copy of the same SONAME is mapped (otherwise torch later reports "No HIP
    # GPUs are available").  Importing torch first makes libpcp_hip.so bind to torch's copy.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    # PCP_HIP_LIBRARY: an alternative build of the same ABI (kernel experiments); never a CPU stand-in
    L = C.CDLL(os.environ.get("PCP_HIP_LIBRARY") or _build.LIB_PATH)
    L.pcp_last_error.restype = C.c_char_p
    L.pcp_last_error.argtypes = [C.c_void_p]
    L.pcp_kernel_name.restype = C.c_char_p
    L.pcp_cloud_size.restype = C.c_int64
    L.pcp_sor_chunk_points.restype = C.c_int64
    L.pcp_sor_chunk_points.argtypes = []
    L.pcp_cloud_size.argtypes = [C.c_void_p]
    L.pcp_frame_count.restype = C.c_int32
    L.pcp_frame_count.argtypes = [C.c_void_p]
    L.pcp_destroy.restype = None
    L.pcp_destroy.argtypes = [C.c_void_p]
    for name in ("pcp_default_camera", "pcp_default_cull_params", "pcp_default_mls_params"):
        getattr(L, name).restype = None
    _lib = L
    return L


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def default_camera() -> Camera:
    cam = Camera()
    load().pcp_default_camera(C.byref(cam))
    return cam


def camera_from_dict(d: dict) -> Camera:
    cam = Camera()
    for k, _ in Camera._fields_:
        setattr(cam, k, d[k])
    return cam


def default_cull_params() -> CullParams:
    p = CullParams()
    load().pcp_default_cull_params(C.byref(p))
    return p


def default_mls_params() -> MLSParams:
    p = MLSParams()
    load().pcp_default_mls_params(C.byref(p))
    return p


def pose_to_matrices(pose, T_opt=None):
    p = Pose(*[float(v) for v in pose])
    w2c = np.zeros(12, np.float32)
    c2w = np.zeros(12, np.float32)
    T = None if T_opt is None else np.ascontiguousarray(T_opt, np.float64).reshape(16)
    rc = load().pcp_pose_to_matrices(C.byref(p), _ptr(T), _ptr(w2c), _ptr(c2w))
    if rc != PCP_OK:
        raise PcpError(rc, load().pcp_last_error(None).decode())
    return w2c, c2w


class Context:
    """Owns one pcp_context (one GPU)."""

    def __init__(self, device: int = 0):
        self.lib = load()
        h = C.c_void_p()
        rc = self.lib.pcp_create(C.c_int32(device), C.byref(h))
        if rc != PCP_OK:
            raise PcpError(rc, self.lib.pcp_last_error(None).decode())
        self.h = h
        self.n = 0
        self.n_frames = 0
        self.camera = None
        self.cull = None

    # -- plumbing ---------------------------------------------------------
    def _check(self, rc: int):
        if rc != PCP_OK:
            raise PcpError(rc, self.lib.pcp_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.pcp_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr: int | None):
        self._check(self.lib.pcp_set_stream(self.h, C.c_void_p(stream_ptr or 0)))

    def synchronize(self):
        self._check(self.lib.pcp_synchronize(self.h))

    # -- configuration ----------------------------------------------------
    def set_camera(self, cam: Camera, cull: CullParams | None = None):
        self.camera = cam
        self.cull = cull if cull is not None else default_cull_params()
        self._check(self.lib.pcp_set_camera(self.h, C.byref(cam), C.byref(self.cull)))

    @property
    def map_shape(self):
        ds = self.cull.downsample_factor
        return (self.camera.cull_height // ds, self.camera.cull_width // ds)

    def upload_cloud(self, x, y, z):
        x = np.ascontiguousarray(x, np.float32)
        y = np.ascontiguousarray(y, np.float32)
        z = np.ascontiguousarray(z, np.float32)
        assert len(x) == len(y) == len(z)
        self._check(self.lib.pcp_upload_cloud(self.h, _ptr(x), _ptr(y), _ptr(z), C.c_int64(len(x))))
        self.n = len(x)

    def upload_cloud_aos(self, pts: np.ndarray):
        pts = np.ascontiguousarray(pts)
        self._check(self.lib.pcp_upload_cloud_aos(self.h, _ptr(pts), C.c_int64(pts.shape[0]),
                                                  C.c_int64(pts.strides[0])))
        self.n = pts.shape[0]

    def set_frames(self, poses, T_opt=None):
        poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 7)
        F = len(poses)
        arr = (Pose * max(F, 1))()
        if F:
            C.memmove(arr, poses.ctypes.data, poses.nbytes)
        T = None
        stride = 0
        if T_opt is not None:
            T = np.ascontiguousarray(T_opt, np.float64).reshape(-1)
            stride = 16 if (T.size == 16 * F and F > 1) else 0
        self._check(self.lib.pcp_set_frames(self.h, arr, C.c_int32(F), _ptr(T), C.c_int32(stride)))
        self.n_frames = F

    def upload_image(self, frame: int, bgr: np.ndarray):
        bgr = np.ascontiguousarray(bgr, np.uint8)
        assert bgr.shape == (self.camera.image_height, self.camera.image_width, 3), bgr.shape
        self._check(self.lib.pcp_upload_image(self.h, C.c_int32(frame), _ptr(bgr), C.c_int64(bgr.strides[0])))

    def upload_image_async(self, frame: int, bgr: np.ndarray):
        """Queues the transfer only: `bgr` (ideally pinned) must stay alive and unchanged until synchronize()."""
        assert bgr.dtype == np.uint8 and bgr.flags.c_contiguous
        assert bgr.shape == (self.camera.image_height, self.camera.image_width, 3), bgr.shape
        self._check(self.lib.pcp_upload_image_async(self.h, C.c_int32(frame), _ptr(bgr), C.c_int64(bgr.strides[0])))

    def upload_images_block(self, first_frame: int, block: np.ndarray):
        """Keyframes first_frame ... from one (count, H, W, 3) array (ideally pinned): DMA in blocks, packed on the device."""
        assert block.dtype == np.uint8 and block.flags.c_contiguous and block.ndim == 4
        assert block.shape[1:] == (self.camera.image_height, self.camera.image_width, 3), block.shape
        self._check(self.lib.pcp_upload_images_block(self.h, C.c_int32(first_frame), C.c_int32(block.shape[0]), _ptr(block),
                                                     C.c_int64(block.strides[1]), C.c_int64(block.strides[0])))

    def upload_image_async_ptr(self, frame: int, ptr: int, row_stride_bytes: int):
        """The same from a raw address: pinned host memory or DEVICE memory (e.g. frames all-gathered over xGMI)."""
        self._check(self.lib.pcp_upload_image_async(self.h, C.c_int32(frame), C.c_void_p(ptr), C.c_int64(row_stride_bytes)))

    def set_image_adjust(self, enable: bool = True, saturation_scale: float = 1.0, brightness_scale: float = 1.0):
        """generateColorMap's 8-bit BGR -> HSV -> BGR round trip applied to the images uploaded from now on."""
        self._check(self.lib.pcp_set_image_adjust(self.h, C.c_int32(1 if enable else 0), C.c_float(saturation_scale),
                                                  C.c_float(brightness_scale)))

    def download_image(self, frame: int):
        """(bgr (H, W, 3), mask (H, W)) of one keyframe as the kernels sample it."""
        hh, ww = self.camera.image_height, self.camera.image_width
        bgr = np.empty((hh, ww, 3), np.uint8)
        mask = np.empty((hh, ww), np.uint8)
        self._check(self.lib.pcp_download_image(self.h, C.c_int32(frame), _ptr(bgr), _ptr(mask)))
        return bgr, mask

    def upload_mask(self, frame: int, gray: np.ndarray):
        gray = np.ascontiguousarray(gray, np.uint8)
        assert gray.shape == (self.camera.image_height, self.camera.image_width), gray.shape
        self._check(self.lib.pcp_upload_mask(self.h, C.c_int32(frame), _ptr(gray), C.c_int64(gray.strides[0])))

    # -- single keyframe --------------------------------------------------
    def project_frame(self, frame: int, want_pixel=True, want_cam=True, device_only=False):
        if device_only:
            self._check(self.lib.pcp_project_frame(self.h, C.c_int32(frame), None, None, None, None))
            return None
        n = self.n
        cell = np.empty(n, np.int32)
        rng = np.empty(n, np.float32)
        pix = np.empty(n, np.int32) if want_pixel else None
        cam = np.empty((3, n), np.float32) if want_cam else None
        self._check(self.lib.pcp_project_frame(self.h, C.c_int32(frame), _ptr(cell), _ptr(pix), _ptr(rng), _ptr(cam)))
        out = dict(cell=cell, range=rng)
        if want_pixel:
            out["pixel"] = pix
        if want_cam:
            out.update(xc=cam[0], yc=cam[1], zc=cam[2])
        return out

    def cull_frame(self, frame: int):
        keep = np.empty(self.n, np.uint8)
        mh, mw = self.map_shape
        dmap = np.empty(mh * mw, np.float32)
        kept = C.c_int64()
        self._check(self.lib.pcp_cull_frame(self.h, C.c_int32(frame), _ptr(keep), C.byref(kept), _ptr(dmap)))
        return keep, dmap.reshape(mh, mw), kept.value

    def cull_frame_into(self, frame: int, device_ptr: int) -> int:
        """pcp_cull_frame with the n keep flags written to DEVICE memory of this context's GPU (ABI v5): the multi-GPU hosts
        exchange them with RCCL.  Returns the number kept."""
        kept = C.c_int64()
        self._check(self.lib.pcp_cull_frame(self.h, C.c_int32(frame), C.c_void_p(device_ptr), C.byref(kept), None))
        return kept.value

    def hull_flags_import_ptr(self, frame: int, device_ptr: int):
        """pcp_hull_flags_import from device memory (n flags of THIS context's points)."""
        self._check(self.lib.pcp_hull_flags_import(self.h, C.c_int32(frame), C.c_void_p(device_ptr)))

    def hpr_stats(self) -> dict:
        """Counters of the last hidden_points_removal run (pcp_hpr_stats)."""
        out = np.zeros(10, np.int64)
        self._check(self.lib.pcp_hpr_stats(self.h, _ptr(out)))
        keys = ("visible", "hidden", "exact_path", "trial_normals", "test_batches", "reserved", "unresolved",
                "exact_evaluations", "cells", "candidates")
        return {k: int(v) for k, v in zip(keys, out)}

    def frame_visible(self, frame: int, capacity: int | None = None):
        cap = self.n if capacity is None else capacity
        idx = np.empty(cap, np.int32)
        rgb = np.empty((cap, 3), np.uint8)
        mv = np.empty(cap, np.uint16)
        cam = np.empty((cap, 3), np.float32)
        wrd = np.empty((cap, 3), np.float32)
        cnt = C.c_int64()
        self._check(self.lib.pcp_frame_visible(self.h, C.c_int32(frame), C.c_int64(cap), _ptr(idx), _ptr(rgb), _ptr(mv),
                                               _ptr(cam), _ptr(wrd), C.byref(cnt)))
        m = min(cnt.value, cap)
        return dict(index=idx[:m], rgb=rgb[:m], mask=mv[:m], xyz_cam=cam[:m], xyz_world=wrd[:m], count=cnt.value)

    # -- whole run --------------------------------------------------------
    def depth_pass(self, f0: int = 0, f1: int | None = None):
        self._check(self.lib.pcp_depth_pass(self.h, C.c_int32(f0), C.c_int32(self.n_frames if f1 is None else f1)))

    def depth_maps_device(self):
        p = C.c_void_p()
        n = C.c_int64()
        self._check(self.lib.pcp_depth_maps_device(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def set_depth_source(self, batched: bool):
        """PCP_DEPTH_BATCHED: the single-keyframe calls use the maps pcp_depth_pass left (MIN-merged across shards)."""
        self._check(self.lib.pcp_set_depth_source(self.h, C.c_int32(1 if batched else 0)))

    def hull_flags_import(self, frame: int, keep):
        """PCP_CULL_HPR on an index shard: the verdicts of keyframe `frame` for this context's points, from a whole-map context."""
        a = np.ascontiguousarray(keep, np.uint8)
        assert len(a) == self.n
        self._check(self.lib.pcp_hull_flags_import(self.h, C.c_int32(frame), _ptr(a)))

    def download_depth_map(self, frame: int):
        mh, mw = self.map_shape
        d = np.empty(mh * mw, np.float32)
        self._check(self.lib.pcp_download_depth_map(self.h, C.c_int32(frame), _ptr(d)))
        return d.reshape(mh, mw)

    def colour_reset(self):
        self._check(self.lib.pcp_colour_reset(self.h))

    def colour_pass(self, f0: int = 0, f1: int | None = None):
        self._check(self.lib.pcp_colour_pass(self.h, C.c_int32(f0), C.c_int32(self.n_frames if f1 is None else f1)))

    def colour_finalise(self, want_top: bool = False, download: bool = True):
        n = self.n
        rgb = np.empty((n, 3), np.uint8) if download else None
        has = np.empty(n, np.uint8) if download else None
        cnt = np.empty(n, np.int32) if want_top else None
        ts = np.empty((n, 5), np.float32) if want_top else None
        tr = np.empty((n, 5), np.uint32) if want_top else None
        tf = np.empty((n, 5), np.int32) if want_top else None
        self._check(self.lib.pcp_colour_finalise(self.h, _ptr(rgb), _ptr(has), _ptr(cnt), _ptr(ts), _ptr(tr), _ptr(tf)))
        return dict(rgb=rgb, has=has, count=cnt, top_score=ts, top_rgb=tr, top_frame=tf)

    def colorize(self, download: bool = True):
        n = self.n
        rgb = np.empty((n, 3), np.uint8) if download else None
        has = np.empty(n, np.uint8) if download else None
        self._check(self.lib.pcp_colorize(self.h, _ptr(rgb), _ptr(has)))
        return dict(rgb=rgb, has=has)

    def colorize_from_depth(self, download: bool = True):
        n = self.n
        rgb = np.empty((n, 3), np.uint8) if download else None
        has = np.empty(n, np.uint8) if download else None
        self._check(self.lib.pcp_colorize_from_depth(self.h, _ptr(rgb), _ptr(has)))
        return dict(rgb=rgb, has=has)

    def download_result_packed(self, out: np.ndarray | None = None, out_ptr: int | None = None):
        """n uint32 words r | g<<8 | b<<16 | has<<24 (out_ptr: e.g. a pinned host buffer)."""
        if out_ptr is not None:
            self._check(self.lib.pcp_download_result_packed(self.h, C.c_void_p(out_ptr)))
            return None
        if out is None:
            out = np.empty(self.n, np.uint32)
        self._check(self.lib.pcp_download_result_packed(self.h, _ptr(out)))
        return out

    def download_result_packed_async(self, out_ptr: int):
        """Enqueue the device-to-host copy on the copy stream; valid after synchronize()."""
        self._check(self.lib.pcp_download_result_packed_async(self.h, C.c_void_p(out_ptr)))

    def download_wait_previous(self):
        self._check(self.lib.pcp_download_wait_previous(self.h))

    def colour_result_device(self):
        p = C.c_void_p()
        n = C.c_int64()
        self._check(self.lib.pcp_colour_result_device(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    # -- MLS --------------------------------------------------------------
    def mls_process(self, params: MLSParams) -> int:
        cnt = C.c_int64()
        self._check(self.lib.pcp_mls_process(self.h, C.byref(params), C.byref(cnt)))
        return cnt.value

    def mls_stream_begin(self, params: "MLSParams", chunk_capacity: int):
        """(total voxels, chunks) of a chunked VOXEL_GRID_DILATION emission (pcp_mls_stream_begin)."""
        total = C.c_int64()
        chunks = C.c_int32()
        self._check(self.lib.pcp_mls_stream_begin(self.h, C.byref(params), C.c_int64(chunk_capacity), C.byref(total),
                                                  C.byref(chunks)))
        return total.value, chunks.value

    def mls_stream_next(self) -> int:
        m = C.c_int64()
        self._check(self.lib.pcp_mls_stream_next(self.h, C.byref(m)))
        return m.value

    def cloud_smooth_stream_begin(self, params: "MLSParams", chunk_capacity: int):
        """CloudSmooth::process whole, its last two stages streamed (pcp_cloud_smooth_stream_begin): (rows of the upsampled
        cloud, rows the trailing outlier removal keeps, chunks)."""
        total, kept = C.c_int64(), C.c_int64()
        chunks = C.c_int32()
        self._check(self.lib.pcp_cloud_smooth_stream_begin(self.h, C.byref(params), C.c_int64(chunk_capacity), C.byref(total),
                                                           C.byref(kept), C.byref(chunks)))
        return total.value, kept.value, chunks.value

    def cloud_smooth_stream_next(self) -> int:
        """Survivors of the next chunk (fetch them with mls_fetch); 0 after the last one."""
        m = C.c_int64()
        self._check(self.lib.pcp_cloud_smooth_stream_next(self.h, C.byref(m)))
        return m.value

    def cloud_smooth_stream_stats(self) -> dict:
        out = (C.c_double * 13)()
        self._check(self.lib.pcp_cloud_smooth_stream_stats(self.h, out))
        return {"halo_planes": int(out[0]), "chunks_redone": int(out[1]), "threshold": float(out[2]),
                "max_displacement_m": float(out[3]), "min_margin_m": float(out[4]), "rows_computed": int(out[5]),
                "sampled_displacement_m": float(out[6]), "device_bytes_held": int(out[7]),
                "begin_seconds": {"filter_fit_voxels": round(out[8], 4), "allocations": round(out[9], 4),
                                  "sweep0": round(out[10], 4), "sweep1_threshold": round(out[11], 4)},
                "allocated_GB": round(out[12] / 1e9, 2)}

    def cloud_smooth_stream_end(self):
        """Ends the stream and frees the device memory it holds (pcp_cloud_smooth_stream_end)."""
        self._check(self.lib.pcp_cloud_smooth_stream_end(self.h))

    def mls_stream_seek(self, chunk: int):
        self._check(self.lib.pcp_mls_stream_seek(self.h, C.c_int32(chunk)))

    def mls_process_shard(self, params: MLSParams, index_begin: int, index_end: int) -> int:
        cnt = C.c_int64()
        self._check(self.lib.pcp_mls_process_shard(self.h, C.byref(params), C.c_int64(index_begin), C.c_int64(index_end),
                                                   C.byref(cnt)))
        return cnt.value

    def mls_process_slab(self, params: MLSParams, slab: int, n_slabs: int) -> int:
        """Queries of one slab of the stage's own spatial order (1 / n_slabs of the work whatever the caller's point order)."""
        cnt = C.c_int64()
        self._check(self.lib.pcp_mls_process_slab(self.h, C.byref(params), C.c_int32(slab), C.c_int32(n_slabs), C.byref(cnt)))
        return cnt.value

    def cloud_smooth(self, params: MLSParams) -> int:
        cnt = C.c_int64()
        self._check(self.lib.pcp_cloud_smooth(self.h, C.byref(params), C.byref(cnt)))
        return cnt.value

    def mls_fetch(self, count: int):
        xyz = np.empty((count, 3), np.float32)
        nrm = np.empty((count, 3), np.float32)
        curv = np.empty(count, np.float32)
        idx = np.empty(count, np.int32)
        self._check(self.lib.pcp_mls_fetch(self.h, C.c_int64(count), _ptr(xyz), _ptr(nrm), _ptr(curv), _ptr(idx)))
        return dict(xyz=xyz, normal=nrm, curvature=curv, index=idx)

    def sor(self, mean_k: int = 60, std_mul: float = 0.7):
        keep = np.empty(self.n, np.uint8)
        kept = C.c_int64()
        self._check(self.lib.pcp_sor(self.h, C.c_int32(mean_k), C.c_double(std_mul), _ptr(keep), C.byref(kept)))
        return keep, kept.value

    def sor_chunk_points(self) -> int:
        return int(self.lib.pcp_sor_chunk_points())

    def sor_partial(self, mean_k: int, slab: int, n_slabs: int):
        """Slab `slab` of `n_slabs` of sor(): (first chunk, (sum, sum of squares) per chunk of the slab)."""
        c = self.sor_chunk_points()
        chunks = (self.n + c - 1) // c
        out = np.zeros((max(chunks, 1), 2), np.float64)
        first, cnt = C.c_int64(), C.c_int64()
        self._check(self.lib.pcp_sor_partial(self.h, C.c_int32(mean_k), C.c_int32(slab), C.c_int32(n_slabs), C.c_int64(len(out)),
                                             _ptr(out), C.byref(first), C.byref(cnt)))
        return first.value, out[:cnt.value].copy()

    def sor_finish(self, std_mul: float, all_chunk_sums: np.ndarray, slab: int, n_slabs: int):
        """Keep flags of the slab's points (n bytes under the caller's indices, 0 for the other slabs' points)."""
        sums = np.ascontiguousarray(all_chunk_sums, np.float64)
        keep = np.empty(self.n, np.uint8)
        kept = C.c_int64()
        self._check(self.lib.pcp_sor_finish(self.h, C.c_double(std_mul), _ptr(sums), C.c_int64(len(sums)), C.c_int32(slab),
                                            C.c_int32(n_slabs), _ptr(keep), C.byref(kept)))
        return keep, kept.value

    def close_pairs(self, radius: float = 2.5e-5) -> int:
        """Map points with another map point closer than `radius` (precondition of the index match-back: expect 0)."""
        cnt = C.c_int64()
        self._check(self.lib.pcp_close_pairs(self.h, C.c_double(radius), C.byref(cnt)))
        return cnt.value

    # -- NID extrinsic refinement -----------------------------------------
    def sor_distances(self) -> np.ndarray:
        """mean distance to the mean_k nearest neighbours per uploaded point, from the last sor() call"""
        out = np.empty(self.n, np.float32)
        self._check(self.lib.pcp_sor_distances(self.h, C.c_int64(len(out)), _ptr(out)))
        return out

    def sor_redo_fraction(self) -> float:
        v = C.c_double()
        self._check(self.lib.pcp_sor_redo_fraction(self.h, C.byref(v)))
        return v.value

    def upload_intensity(self, intensity):
        a = np.ascontiguousarray(intensity, np.float32)
        self._check(self.lib.pcp_upload_intensity(self.h, _ptr(a), C.c_int64(len(a))))

    def nid_prepare(self) -> int:
        cnt = C.c_int64()
        self._check(self.lib.pcp_nid_prepare(self.h, C.byref(cnt)))
        return cnt.value

    def nid_evaluate(self, T, T_init=None, bins: int = 16):
        T = np.ascontiguousarray(T, np.float64).reshape(16)
        Ti = None if T_init is None else np.ascontiguousarray(T_init, np.float64).reshape(16)
        cost = C.c_double()
        grad = np.zeros(6, np.float64)
        valid = C.c_int32()
        self._check(self.lib.pcp_nid_evaluate(self.h, _ptr(T), _ptr(Ti), C.c_int32(bins), C.byref(cost), _ptr(grad),
                                              C.byref(valid)))
        return cost.value, grad, bool(valid.value)

    def nid_optimize(self, T_init, bins: int = 16, max_outer_iterations: int = 10):
        Ti = np.ascontiguousarray(T_init, np.float64).reshape(16)
        out = np.zeros(16, np.float64)
        cost = C.c_double()
        evals = C.c_int32()
        self._check(self.lib.pcp_nid_optimize(self.h, _ptr(Ti), C.c_int32(bins), C.c_int32(max_outer_iterations), _ptr(out),
                                              C.byref(cost), C.byref(evals)))
        return out.reshape(4, 4), cost.value, evals.value

    # NID over an index-sharded map: accumulate -> all-reduce(SUM) of the histograms -> finish
    def nid_accumulate(self, T, bins: int = 16):
        T = np.ascontiguousarray(T, np.float64).reshape(16)
        self._check(self.lib.pcp_nid_accumulate(self.h, _ptr(T), C.c_int32(bins)))

    def nid_histograms_device(self):
        p = C.c_void_p()
        n = C.c_int64()
        self._check(self.lib.pcp_nid_histograms_device(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def nid_finish(self, bins: int = 16):
        cost = C.c_double()
        grad = np.zeros(6, np.float64)
        valid = C.c_int32()
        self._check(self.lib.pcp_nid_finish(self.h, C.c_int32(bins), C.byref(cost), _ptr(grad), C.byref(valid)))
        return cost.value, grad, bool(valid.value)

    def nid_optimize_with(self, evaluate, T_init, bins: int = 16, max_outer_iterations: int = 10):
        """pcp_nid_optimize_with: evaluate(T 4x4, bins) -> (cost, grad6, valid), e.g. the sharded evaluation."""
        errors = []

        @NID_EVAL_FN
        def trampoline(_user, T, b, cost, grad, valid):
            try:
                c, g, ok = evaluate(np.ctypeslib.as_array(T, (16,)).copy().reshape(4, 4), int(b))
                cost[0] = float(c)
                for k in range(6):
                    grad[k] = float(g[k])
                valid[0] = 1 if ok else 0
                return 0
            except Exception as e:  # noqa: BLE001 -- reported through the status code, re-raised below
                errors.append(e)
                return -3

        Ti = np.ascontiguousarray(T_init, np.float64).reshape(16)
        out = np.zeros(16, np.float64)
        cost = C.c_double()
        evals = C.c_int32()
        rc = self.lib.pcp_nid_optimize_with(self.h, trampoline, None, _ptr(Ti), C.c_int32(bins), C.c_int32(max_outer_iterations),
                                            _ptr(out), C.byref(cost), C.byref(evals))
        if errors:
            raise errors[0]
        self._check(rc)
        return out.reshape(4, 4), cost.value, evals.value

    # -- measurement ------------------------------------------------------
    def timing_enable(self, on: bool = True):
        self._check(self.lib.pcp_timing_enable(self.h, C.c_int32(1 if on else 0)))

    def timing_reset(self):
        self._check(self.lib.pcp_timing_reset(self.h))

    def timing_get(self, kernel_id: int):
        ms = C.c_double()
        cnt = C.c_int64()
        self._check(self.lib.pcp_timing_get(self.h, C.c_int32(kernel_id), C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value

    def tile_mask_density(self) -> float:
        v = C.c_double()
        self._check(self.lib.pcp_tile_mask_density(self.h, C.byref(v)))
        return v.value

    def selftest_arithmetic(self, samples: int = 1 << 26, seed: int = 1):
        """(fp64 mismatches, fp32 mismatches) of the short exact division sequences against `/`."""
        a, b = C.c_int64(), C.c_int64()
        self._check(self.lib.pcp_selftest_arithmetic(self.h, C.c_int64(samples), C.c_uint64(seed), C.byref(a), C.byref(b)))
        return a.value, b.value

    def tile_masks(self) -> np.ndarray:
        """(tiles, mask_words) uint32: the tile x keyframe masks as the last depth pass left them."""
        t, w = C.c_int64(), C.c_int32()
        self._check(self.lib.pcp_tile_masks(self.h, C.byref(t), C.byref(w), None))
        out = np.empty((t.value, w.value), np.uint32)
        self._check(self.lib.pcp_tile_masks(self.h, None, None, _ptr(out)))
        return out

    def kernel_name(self, kernel_id: int) -> str:
        return self.lib.pcp_kernel_name(C.c_int32(kernel_id)).decode()
