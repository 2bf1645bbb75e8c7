"""Synthetic scene of SURVEY.md section 8(d): room + occluders, Lissajous
trajectory, procedural BGR8 images and disc masks.  Pure numpy, seeded, used by
tests, bench.py and smoke() (the reference ships no data sets).

Camera sets:
  "ref": reference constants, PCP/src/PointCloudProcessor.cpp:57-60, 4096x3000
         (cull size :525).
  "cfg": BASELINE.json's 1920x1080 -- reference K scaled by 1920/4096, principal
         point centred, same distortion.
"""
from __future__ import annotations

import numpy as np

SEED = 20241008

REF_K = dict(fx=4818.200388954926, fy=4819.10345841615, cx=2032.4178620390019, cy=1535.1895959282901)
REF_D = dict(k1=0.003043514741045163, k2=0.06634739187544138, p1=-0.000217681797407554,
             p2=-0.0006654964142658197, k3=0.0)


def camera_dict(kind: str = "cfg") -> dict:
    if kind == "ref":
        d = dict(REF_K)
        d.update(REF_D)
        d.update(image_width=4096, image_height=3000, cull_width=4096, cull_height=3000)
        return d
    if kind == "cfg":
        d = dict(fx=2258.5, fy=2258.5, cx=960.0, cy=540.0)
        d.update(REF_D)
        d.update(image_width=1920, image_height=1080, cull_width=1920, cull_height=1080)
        return d
    if kind == "tiny":  # small images for unit tests: same optics as cfg, /4
        d = dict(fx=564.625, fy=564.625, cx=240.0, cy=135.0)
        d.update(REF_D)
        d.update(image_width=480, image_height=270, cull_width=480, cull_height=270)
        return d
    raise ValueError(kind)


# ---------------------------------------------------------------------------
# geometry
# ---------------------------------------------------------------------------

ROOM = np.array([12.0, 10.0, 4.0])  # x in [-6,6], y in [-5,5], z in [0,4]


def _surfaces(rng: np.random.Generator):
    """Returns a list of (kind, params, area)."""
    surf = []
    hx, hy, hz = ROOM[0] / 2, ROOM[1] / 2, ROOM[2]
    # room: 6 rectangles (origin, edge u, edge v, inward normal)
    rects = [
        ((-hx, -hy, 0), (ROOM[0], 0, 0), (0, ROOM[1], 0), (0, 0, 1)),  # floor
        ((-hx, -hy, hz), (ROOM[0], 0, 0), (0, ROOM[1], 0), (0, 0, -1)),  # ceiling
        ((-hx, -hy, 0), (ROOM[0], 0, 0), (0, 0, hz), (0, 1, 0)),
        ((-hx, hy, 0), (ROOM[0], 0, 0), (0, 0, hz), (0, -1, 0)),
        ((-hx, -hy, 0), (0, ROOM[1], 0), (0, 0, hz), (1, 0, 0)),
        ((hx, -hy, 0), (0, ROOM[1], 0), (0, 0, hz), (-1, 0, 0)),
    ]
    for o, u, v, nrm in rects:
        u = np.array(u, float)
        v = np.array(v, float)
        surf.append(("rect", (np.array(o, float), u, v, np.array(nrm, float)), np.linalg.norm(np.cross(u, v))))
    # 6 spheres radius 0.5-1.0
    for _ in range(6):
        r = rng.uniform(0.5, 1.0)
        c = np.array([rng.uniform(-hx + 1.5, hx - 1.5), rng.uniform(-hy + 1.5, hy - 1.5), rng.uniform(r, hz - r)])
        surf.append(("sphere", (c, r), 4 * np.pi * r * r))
    # 4 box pillars 0.4-0.8 m square, floor to ceiling: 4 side faces each
    for _ in range(4):
        w = rng.uniform(0.4, 0.8)
        cx, cy = rng.uniform(-hx + 1.0, hx - 1.0), rng.uniform(-hy + 1.0, hy - 1.0)
        x0, x1, y0, y1 = cx - w / 2, cx + w / 2, cy - w / 2, cy + w / 2
        faces = [
            ((x0, y0, 0), (w, 0, 0), (0, 0, hz), (0, -1, 0)),
            ((x0, y1, 0), (w, 0, 0), (0, 0, hz), (0, 1, 0)),
            ((x0, y0, 0), (0, w, 0), (0, 0, hz), (-1, 0, 0)),
            ((x1, y0, 0), (0, w, 0), (0, 0, hz), (1, 0, 0)),
        ]
        for o, u, v, nrm in faces:
            u = np.array(u, float)
            v = np.array(v, float)
            surf.append(("rect", (np.array(o, float), u, v, np.array(nrm, float)), np.linalg.norm(np.cross(u, v))))
    return surf


def make_cloud(n: int, seed: int = SEED, noise_sigma: float = 1e-3, spatial_sort: bool = False):
    """n points sampled uniformly by area on the scene, 1 mm Gaussian noise along
    the normal, fp32 SoA.  Returns (x, y, z, intensity)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    surf = _surfaces(rng)
    areas = np.array([s[2] for s in surf])
    counts = rng.multinomial(n, areas / areas.sum())
    chunks = []
    for (kind, prm, _), m in zip(surf, counts):
        if m == 0:
            continue
        if kind == "rect":
            o, u, v, nrm = prm
            a = rng.random(m)[:, None]
            b = rng.random(m)[:, None]
            p = o + a * u + b * v
            p = p + rng.normal(0.0, noise_sigma, m)[:, None] * nrm
        else:
            c, r = prm
            d = rng.normal(size=(m, 3))
            d /= np.linalg.norm(d, axis=1, keepdims=True)
            p = c + (r + rng.normal(0.0, noise_sigma, m))[:, None] * d
        chunks.append(p)
    pts = np.concatenate(chunks, axis=0) if chunks else np.zeros((0, 3))
    perm = rng.permutation(len(pts))  # LiDAR maps are not surface-ordered
    pts = pts[perm]
    if spatial_sort:
        key = (np.floor((pts[:, 0] + 8) / 0.25) * 4096 + np.floor((pts[:, 1] + 8) / 0.25)) * 64 + np.floor(
            (pts[:, 2] + 1) / 0.25)
        pts = pts[np.argsort(key, kind="stable")]
    pts = pts.astype(np.float32)
    inten = rng.random(len(pts)).astype(np.float32)
    return (np.ascontiguousarray(pts[:, 0]), np.ascontiguousarray(pts[:, 1]), np.ascontiguousarray(pts[:, 2]),
            inten)


# ---------------------------------------------------------------------------
# trajectory
# ---------------------------------------------------------------------------

def _path(s):
    x = 3.5 * np.sin(2 * np.pi * s)
    y = 2.5 * np.sin(2 * np.pi * 2.03 * s + 0.5)
    z = 1.5 + 0.5 * np.sin(2 * np.pi * 3.07 * s)
    return np.stack([x, y, z], axis=-1)


def _rot_to_quat(R):
    """3x3 rotation -> (qw,qx,qy,qz), unit."""
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = np.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        q = np.array([(R[2, 1] - R[1, 2]) / s, 0.25 * s, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s])
    elif R[1, 1] > R[2, 2]:
        s = np.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        q = np.array([(R[0, 2] - R[2, 0]) / s, (R[0, 1] + R[1, 0]) / s, 0.25 * s, (R[1, 2] + R[2, 1]) / s])
    else:
        s = np.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        q = np.array([(R[1, 0] - R[0, 1]) / s, (R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, 0.25 * s])
    return q / np.linalg.norm(q)


def make_trajectory(n_frames: int, seed: int = SEED, spacing: float = 0.12):
    """F poses (x,y,z,qw,qx,qy,qz) as the odometry file stores them: positions and
    quaternions rounded to 8 decimals, timestamps to 6 (the producer's format,
    PCP/scripts/make_vo_odom_for_fastlio.py:126).  Every pose passes the 0.1 m keyframe rule."""
    rng = np.random.Generator(np.random.PCG64(seed + 1))
    # fine sampling, then arc-length resampling
    ds = 1e-4
    need = spacing * (n_frames + 2)
    s_hi = max(1.0, need / 25.0)
    while True:
        s = np.arange(0.0, s_hi, ds)
        p = _path(s)
        seg = np.linalg.norm(np.diff(p, axis=0), axis=1)
        arc = np.concatenate([[0.0], np.cumsum(seg)])
        if arc[-1] >= need:
            break
        s_hi *= 1.5
    targets = spacing * np.arange(n_frames)
    s_at = np.interp(targets, arc, s)
    pos = _path(s_at)
    tan = _path(s_at + 1e-5) - _path(s_at - 1e-5)
    tan /= np.linalg.norm(tan, axis=1, keepdims=True)
    poses = np.zeros((n_frames, 7))
    up = np.array([0.0, 0.0, 1.0])
    for i in range(n_frames):
        fwd = tan[i]
        yaw, pitch = np.deg2rad(rng.uniform(-15, 15, 2))
        right = np.cross(fwd, up)
        if np.linalg.norm(right) < 1e-6:
            right = np.array([1.0, 0.0, 0.0])
        right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        # jitter: rotate about down (yaw) then about right (pitch)
        fwd = np.cos(yaw) * fwd + np.sin(yaw) * right
        right = np.cross(down, fwd)
        fwd = np.cos(pitch) * fwd + np.sin(pitch) * down
        down = np.cross(fwd, right)
        R = np.stack([right, down, fwd], axis=1)  # camera axes as world columns (x right, y down, z fwd)
        q = _rot_to_quat(R)
        poses[i, :3] = np.round(pos[i], 8)
        poses[i, 3:] = np.round(q, 8)
    ts = np.round(1700000000.0 + 0.1 * np.arange(n_frames), 6)
    return poses, ts


def odometry_line(t: float, pose) -> str:
    """One line of vo_interpolated_odom.txt exactly as the reference's producer writes it
    (PCP/scripts/make_vo_odom_for_fastlio.py:126): `ts x y z qw qx qy qz`, 6 / 8 decimals."""
    return "%.6f %.8f %.8f %.8f %.8f %.8f %.8f %.8f\n" % (t, *[float(v) for v in pose])


def write_odometry(path, ts, poses) -> None:
    with open(path, "w") as f:
        for t, p in zip(ts, poses):
            f.write(odometry_line(t, p))


# ---------------------------------------------------------------------------
# images / masks
# ---------------------------------------------------------------------------

def _hash32(a: np.ndarray) -> np.ndarray:
    a = a.astype(np.uint32, copy=True)
    a ^= a >> np.uint32(16)
    a *= np.uint32(0x7FEB352D)
    a ^= a >> np.uint32(15)
    a *= np.uint32(0x846CA68B)
    a ^= a >> np.uint32(16)
    return a


_NOISE_CACHE: dict = {}


def _noise_plane(width: int, height: int, seed: int) -> np.ndarray:
    """(H,W) uint32 hash noise, computed once per (size, seed)."""
    key = (width, height, seed)
    if key not in _NOISE_CACHE:
        v, u = np.meshgrid(np.arange(height, dtype=np.uint32), np.arange(width, dtype=np.uint32), indexing="ij")
        _NOISE_CACHE.clear()
        _NOISE_CACHE[key] = _hash32((v * np.uint32(width) + u) ^ np.uint32(seed & 0xFFFFFFFF))
    return _NOISE_CACHE[key]


def make_image(frame: int, width: int, height: int, seed: int = SEED) -> np.ndarray:
    """Procedural BGR8 (H,W,3): frame-seeded smooth gradient + 6-bit hash noise
    (a frame-dependent roll of one cached noise plane); stands for the image
    *after* the HSV round trip (Appendix B5).  Green is never 0, so a coloured
    point can never be mistaken for the never-seen (0,0,0)."""
    h0 = int(_hash32(np.array([(seed * 7919 + frame * 104729 + 13) & 0xFFFFFFFF], dtype=np.uint32))[0])
    noise = np.roll(_noise_plane(width, height, seed), (h0 % height, (h0 >> 12) % width), axis=(0, 1))
    u = np.arange(width, dtype=np.uint32)[None, :]
    v = np.arange(height, dtype=np.uint32)[:, None]
    img = np.empty((height, width, 3), np.uint8)
    for c in range(3):
        a = (h0 >> (8 * c)) & 0xFF
        grad = (u * np.uint32(3 + c)) // np.uint32(16) + (v * np.uint32(5 - c)) // np.uint32(16) + np.uint32(a)
        img[:, :, c] = ((grad + ((noise >> np.uint32(8 * c)) & np.uint32(0x3F))) & np.uint32(0xFF)).astype(np.uint8)
    img[:, :, 1] |= 1
    return img


def make_mask(frame: int, width: int, height: int, seed: int = SEED) -> np.ndarray:
    """gray8 (H,W): 255 inside 3 random discs, else 0."""
    rng = np.random.Generator(np.random.PCG64(seed * 31 + frame))
    v, u = np.meshgrid(np.arange(height), np.arange(width), indexing="ij")
    m = np.zeros((height, width), np.uint8)
    for _ in range(3):
        cu, cv = rng.uniform(0, width), rng.uniform(0, height)
        r = rng.uniform(0.05, 0.2) * min(width, height)
        m[(u - cu) ** 2 + (v - cv) ** 2 < r * r] = 255
    return m
