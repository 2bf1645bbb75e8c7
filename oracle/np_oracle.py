"""numpy twin of the CPU restatement (stages A0-A8 and A7).

TEST INFRASTRUCTURE ONLY, PARITY UNPINNED (see oracle/pcp_oracle.h).  Written
independently of pcp_oracle.c from the same reference lines so that the two
restatements check each other; it also generates the committed golden vectors
(tests/golden/make_golden.py).  numpy fp32 / fp64 array ops are individually
rounded (no FMA contraction), which is the arithmetic model of Appendix A.

PCP/ = /root/reference/PointCloudProcessor/.
"""
from __future__ import annotations

import numpy as np

f32 = np.float32
f64 = np.float64
FLT_MAX = np.finfo(np.float32).max


# --------------------------------------------------------------------------- A1
def quat_to_rot(qw, qx, qy, qz):
    """Eigen Quaterniond::toRotationMatrix, no normalisation [upstream]."""
    tx, ty, tz = 2.0 * qx, 2.0 * qy, 2.0 * qz
    twx, twy, twz = tx * qw, ty * qw, tz * qw
    txx, txy, txz = tx * qx, ty * qx, tz * qx
    tyy, tyz, tzz = ty * qy, tz * qy, tz * qz
    return np.array([
        [1.0 - (tyy + tzz), txy - twz, txz + twy],
        [txy + twz, 1.0 - (txx + tzz), tyz - twx],
        [txz - twy, tyz + twx, 1.0 - (txx + tyy)],
    ], dtype=f64)


def pose_to_matrices(pose, T_opt=None):
    """PCP/src/PointCloudProcessor.cpp:495-519 -> (w2c, c2w) 3x4 fp32."""
    x, y, z, qw, qx, qy, qz = [float(v) for v in pose]
    R = quat_to_rot(qw, qx, qy, qz)
    t = np.array([x, y, z], f64)
    if T_opt is None:
        Rt = R.T
        tr = -((Rt[:, 0] * t[0] + Rt[:, 1] * t[1]) + Rt[:, 2] * t[2])
        w2c = np.concatenate([Rt, tr[:, None]], axis=1).astype(f32)
        c2w = np.concatenate([R, t[:, None]], axis=1).astype(f32)
        return w2c, c2w
    T = np.asarray(T_opt, f64).reshape(4, 4)
    M = np.zeros((3, 4), f64)
    for r in range(3):
        for c in range(4):
            s = (R[r, 0] * T[0, c] + R[r, 1] * T[1, c]) + R[r, 2] * T[2, c]
            if c == 3:
                s = s + t[r] * T[3, 3]
            M[r, c] = s
    c2w = M.astype(f32)
    return affine_inverse_f32(c2w), c2w


def affine_inverse_f32(m):
    """General fp32 affine inverse, Eigen Transform<float,3,Affine>::inverse() [upstream Eigen 3.3.7
    LU/InverseImpl.h compute_inverse<3>: cofactors, det expanded along COLUMN 0; a 3-term fp32 sum goes through
    Redux.h's scalar unroller as p0 + (p1 + p2); translation = (-Linv) * t with the same 3-term sums]."""
    m = np.asarray(m, f32).reshape(3, 4)

    def cof(i, j):
        i1, i2, j1, j2 = (i + 1) % 3, (i + 2) % 3, (j + 1) % 3, (j + 2) % 3
        return m[i1, j1] * m[i2, j2] - m[i1, j2] * m[i2, j1]

    col0 = [cof(0, 0), cof(1, 0), cof(2, 0)]
    det = col0[0] * m[0, 0] + (col0[1] * m[1, 0] + col0[2] * m[2, 0])
    inv = f32(1.0) / det
    L = np.zeros((3, 3), f32)
    for j in range(3):
        L[0, j] = col0[j] * inv
    for j in range(3):
        L[1, j] = cof(j, 1) * inv
        L[2, j] = cof(j, 2) * inv
    t = m[:, 3]
    out = np.zeros((3, 4), f32)
    out[:, :3] = L
    out[:, 3] = (-L[:, 0]) * t[0] + ((-L[:, 1]) * t[1] + (-L[:, 2]) * t[2])
    return out


# --------------------------------------------------------------------------- A2
def transform(m, x, y, z):
    """pcl::transformPointCloud, PCL 1.10 SSE association [upstream]."""
    m = np.asarray(m, f32).reshape(3, 4)
    x, y, z = (np.asarray(a, f32) for a in (x, y, z))
    out = []
    for r in range(3):
        out.append(x * m[r, 0] + (y * m[r, 1] + (z * m[r, 2] + m[r, 3])))
    return out


# --------------------------------------------------------------------------- A3
def project(cam: dict, xc, yc, zc):
    """PCP/include/camera/pinhole.hpp:13-51, fp64, left-to-right."""
    X, Y, Z = (np.asarray(a, f64) for a in (xc, yc, zc))
    with np.errstate(all="ignore"):
        xn = X / Z
        yn = Y / Z
        x2 = xn * xn
        y2 = yn * yn
        r2 = x2 + y2
        r4 = r2 * r2
        r6 = r2 * r4
        rc = ((1.0 + cam["k1"] * r2) + cam["k2"] * r4) + cam["k3"] * r6
        t1 = (2.0 * xn) * yn
        t2 = r2 + 2.0 * x2
        t3 = r2 + 2.0 * y2
        xd = (rc * xn + cam["p1"] * t1) + cam["p2"] * t2
        yd = (rc * yn + cam["p1"] * t3) + cam["p2"] * t1
        u = cam["fx"] * xd + cam["cx"]
        v = cam["fy"] * yd + cam["cy"]
    return u, v


def _trunc_ok(a):
    a = np.asarray(a)
    return np.isfinite(a) & (a > -2147483648.0) & (a < 2147483648.0)


def project_frame(cam: dict, w2c, x, y, z, ds: int = 14, enable_zbuffer: bool = True):
    """A2-A5 per point: cam coords, (u,v), cell, pixel, range."""
    xc, yc, zc = transform(w2c, x, y, z)
    front = zc > 0
    u, v = project(cam, xc, yc, zc)
    X, Y, Z = xc.astype(f64), yc.astype(f64), zc.astype(f64)
    rng = np.sqrt((X * X + Y * Y) + Z * Z)
    W, H = cam["cull_width"], cam["cull_height"]
    mw, mh = W // ds, H // ds
    with np.errstate(all="ignore"):
        cxf = u.astype(f32) / f32(ds)
        cyf = v.astype(f32) / f32(ds)
    ok = front & _trunc_ok(cxf) & _trunc_ok(cyf)
    cxi = np.where(ok, np.trunc(np.where(ok, cxf, 0)), -1).astype(np.int64)
    cyi = np.where(ok, np.trunc(np.where(ok, cyf, 0)), -1).astype(np.int64)
    cand = ok & (cxi >= 0) & (cyi >= 0) & (cxi < W) & (cyi < H)
    inmap = cand & (cxi < mw) & (cyi < mh)
    # -2 (candidate without a map cell) is only reported when the depth buffer is off
    cell = np.where(inmap, cyi * mw + cxi, np.where(cand & (not enable_zbuffer), -2, -1)).astype(np.int32)
    okp = front & _trunc_ok(u) & _trunc_ok(v)
    ui = np.where(okp, np.trunc(np.where(okp, u, 0)), -1).astype(np.int64)
    vi = np.where(okp, np.trunc(np.where(okp, v, 0)), -1).astype(np.int64)
    inimg = okp & (ui >= 0) & (ui < cam["image_width"]) & (vi >= 0) & (vi < cam["image_height"])
    pixel = np.where(inimg, vi * cam["image_width"] + ui, -1).astype(np.int32)
    range_f = np.where(front, rng, FLT_MAX).astype(f32)
    return dict(xc=xc, yc=yc, zc=zc, u=u, v=v, cell=cell, pixel=pixel, range=range_f, range64=rng)


# --------------------------------------------------------------------------- A4
def cull_frame(cam: dict, w2c, x, y, z, ds: int = 14, slack: float = 0.05, enable_zbuffer: bool = True):
    """view_culling.cpp:52-174: returns keep mask, depth map, projection dict."""
    p = project_frame(cam, w2c, x, y, z, ds, enable_zbuffer)
    mw, mh = cam["cull_width"] // ds, cam["cull_height"] // ds
    dmap = np.full(mw * mh, FLT_MAX, f32)
    if not enable_zbuffer:
        return p["cell"] != -1, dmap.reshape(mh, mw), p
    inmap = p["cell"] >= 0
    np.minimum.at(dmap, p["cell"][inmap], p["range"][inmap])
    keep = np.zeros(len(p["cell"]), bool)
    keep[inmap] = ~(p["range64"][inmap] > dmap[p["cell"][inmap]].astype(f64) + slack)
    return keep, dmap.reshape(mh, mw), p


# --------------------------------------------------------------------------- A6
def scores(xc, yc, zc, pose):
    """hpp:205-236 + cpp:588 (identity mode)."""
    xc, yc, zc = (np.asarray(a, f32) for a in (xc, yc, zc))
    px, py, pz = float(pose[0]), float(pose[1]), float(pose[2])
    dx, dy, dz = xc.astype(f64) - px, yc.astype(f64) - py, zc.astype(f64) - pz
    sq = (dx * dx + dy * dy) + dz * dz
    with np.errstate(all="ignore"):
        cosA = np.where(sq > 0, dz / np.sqrt(sq), dz)
    o = ((cosA + 1.0) / 2.0).astype(f32)
    o = f32(0.2) + f32(0.8) * o
    dist = np.sqrt((xc * xc + yc * yc) + zc * zc).astype(f32)
    nd = np.abs(dist - f32(2.0)) / f32(2.0)
    nd = np.where(nd < f32(1.0), nd, f32(1.0)).astype(f32)
    d = f32(1.0) - nd
    d = f32(0.2) + f32(0.8) * d
    final = ((o + d).astype(f64) / 2.0).astype(f32)
    return o, d, final


# --------------------------------------------------------------------------- A8
def colorize(cam: dict, x, y, z, poses, images, ds: int = 14, slack: float = 0.05, enable_zbuffer: bool = True,
             T_opt=None, roundtrip: bool = False):
    """Collect-all + stable sort (the reference's shape, cpp:590-591,604-631).  roundtrip: the reference's fp32
    world round trip, 10 um self-match test and scores from c2w.inverse() p_w (cpp:555,571-579) instead of the
    identity mode of Appendix B3 (no neighbour search: see colorize_faithful for that)."""
    n = len(x)
    x, y, z = (np.asarray(a, f32) for a in (x, y, z))
    eps = f64(f32(1e-5))
    r2 = f32(eps * eps)
    lists = [[] for _ in range(n)]
    for f, pose in enumerate(poses):
        T = None
        if T_opt is not None:
            T = np.asarray(T_opt, f64).reshape(-1, 16)
            T = T[f if len(T) > 1 else 0]
        w2c, c2w = pose_to_matrices(pose, T)
        keep, _, p = cull_frame(cam, w2c, x, y, z, ds, slack, enable_zbuffer)
        sel = np.nonzero(keep & (p["pixel"] >= 0))[0]
        if len(sel) == 0:
            continue
        sx, sy, sz = p["xc"][sel], p["yc"][sel], p["zc"][sel]
        if roundtrip:
            wx, wy, wz = transform(c2w, sx, sy, sz)
            dx, dy, dz = wx - x[sel], wy - y[sel], wz - z[sel]
            d = dx * dx
            d = d + dy * dy
            d = d + dz * dz
            found = d < r2
            sel, wx, wy, wz = sel[found], wx[found], wy[found], wz[found]
            if len(sel) == 0:
                continue
            sx, sy, sz = affine_times_point_eigen(affine_inverse_f32(c2w), wx, wy, wz)
        img = np.asarray(images[f], np.uint8).reshape(-1, 3)
        bgr = img[p["pixel"][sel]]
        _, _, fin = scores(sx, sy, sz, pose)
        for k, i in enumerate(sel):
            lists[i].append((float(fin[k]), int(bgr[k, 2]), int(bgr[k, 1]), int(bgr[k, 0]), f))
    rgb = np.zeros((n, 3), np.uint8)
    count = np.zeros(n, np.int32)
    top_score = np.full((n, 5), -1.0, f32)
    top_rgb = np.zeros((n, 5), np.uint32)
    top_frame = np.full((n, 5), -1, np.int32)
    for i, lst in enumerate(lists):
        count[i] = len(lst)
        if not lst:
            continue
        lst = sorted(lst, key=lambda e: -e[0])  # stable: ties keep ascending frame order (B8)
        lst = lst[:5]
        tot = f32(0)
        acc = [f32(0), f32(0), f32(0)]
        for k, (s, r, g, b, fr) in enumerate(lst):
            s = f32(s)
            acc[0] = acc[0] + f32(r) * s
            acc[1] = acc[1] + f32(g) * s
            acc[2] = acc[2] + f32(b) * s
            tot = tot + s
            top_score[i, k] = s
            top_rgb[i, k] = (r << 16) | (g << 8) | b
            top_frame[i, k] = fr
        for c in range(3):
            rgb[i, c] = np.uint8(int(acc[c] / tot))
    has = (rgb != 0).any(axis=1).astype(np.uint8)
    return dict(rgb=rgb, has=has, count=count, top_score=top_score, top_rgb=top_rgb, top_frame=top_frame)


def hpr_candidates(cam: dict, w2c, x, y, z):
    """The candidate filter of hidden_points_removal (view_culling.cpp:276-288): z > 0 and
    0 <= (int)u < W, 0 <= (int)v < H against the full cull size.  Returns the mask and the projection."""
    p = project_frame(cam, w2c, x, y, z)
    u, v = p["u"], p["v"]
    ok = (p["zc"] > 0) & _trunc_ok(u) & _trunc_ok(v)
    ui = np.where(ok, np.trunc(np.where(ok, u, 0)), -1)
    vi = np.where(ok, np.trunc(np.where(ok, v, 0)), -1)
    return ok & (ui >= 0) & (ui < cam["cull_width"]) & (vi >= 0) & (vi < cam["cull_height"]), p


def affine_times_point_eigen(m, x, y, z):
    """Eigen Affine3f * Vector4f (PointCloudProcessor.cpp:579): per row (m0 x + m1 y) + (m2 z + m3 * 1) in fp32
    [upstream Eigen 3.3.7: coefficient-based 3x4 * 4x1 product, 4-term sum split 2 + 2 by Redux.h]."""
    m = np.asarray(m, f32).reshape(3, 4)
    x, y, z = (np.asarray(a, f32) for a in (x, y, z))
    return [(m[r, 0] * x + m[r, 1] * y) + (m[r, 2] * z + m[r, 3] * f32(1.0)) for r in range(3)]


def colorize_faithful(cam: dict, x, y, z, poses, images, ds: int = 14, slack: float = 0.05, T_opt=None,
                      hpr_candidates_mode: bool = False):
    """pcdColorizationAndSmooth with the reference's own match-back (Appendix B3 faithful mode,
    PointCloudProcessor.cpp:554-592): p_w = c2w p_c (fp32, PCL association), radiusSearch(1e-5) over the original
    cloud (fp32 L2_Simple distance, strict <, ascending distance), every match credited with scores from
    c2w.inverse() p_w.  Neighbour candidates come from scipy's cKDTree (a superset within a padded radius) and are
    filtered with the fp32 rule, so the result set does not depend on the tree."""
    from scipy.spatial import cKDTree

    x, y, z = (np.asarray(a, f32) for a in (x, y, z))
    n = len(x)
    P64 = np.stack([x, y, z], axis=1).astype(f64)
    tree = cKDTree(P64)
    eps = f64(f32(1e-5))
    r2 = f32(eps * eps)
    lists = [[] for _ in range(n)]
    stats = dict(samples=0, unmatched=0, self_missed=0, cross_credits=0)
    for f, pose in enumerate(poses):
        T = None
        if T_opt is not None:
            T = np.asarray(T_opt, f64).reshape(-1, 16)
            T = T[f if len(T) > 1 else 0]
        w2c, c2w = pose_to_matrices(pose, T)
        inv = affine_inverse_f32(c2w)
        if hpr_candidates_mode:
            keep, p = hpr_candidates(cam, w2c, x, y, z)
        else:
            keep, _, p = cull_frame(cam, w2c, x, y, z, ds, slack, True)
        sel = np.nonzero(keep & (p["pixel"] >= 0))[0]
        if len(sel) == 0:
            continue
        img = np.asarray(images[f], np.uint8).reshape(-1, 3)
        bgr = img[p["pixel"][sel]]
        wx, wy, wz = transform(c2w, p["xc"][sel], p["yc"][sel], p["zc"][sel])
        sx, sy, sz = affine_times_point_eigen(inv, wx, wy, wz)
        _, _, fin = scores(sx, sy, sz, pose)
        near = tree.query_ball_point(np.stack([wx, wy, wz], axis=1).astype(f64), float(eps) * 1.01 + 1e-9)
        for k, i in enumerate(sel):
            stats["samples"] += 1
            cand = np.array(sorted(near[k]), np.int64)
            if len(cand):
                dx, dy, dz = wx[k] - x[cand], wy[k] - y[cand], wz[k] - z[cand]
                d = dx * dx
                d = d + dy * dy
                d = d + dz * dz
                hit = d < r2
                cand, d = cand[hit], d[hit]
                cand = cand[np.lexsort((cand, d))]
            if len(cand) == 0:
                stats["unmatched"] += 1
                stats["self_missed"] += 1
                continue
            if i not in cand:
                stats["self_missed"] += 1
            for j in cand:
                if j != i:
                    stats["cross_credits"] += 1
                lists[j].append((float(fin[k]), int(bgr[k, 2]), int(bgr[k, 1]), int(bgr[k, 0]), f))
    out = _finalise_lists(lists, n)
    out["stats"] = stats
    return out


def _finalise_lists(lists, n):
    """smoothColors + removePointsWithNoColor on collected (score, r, g, b, frame) lists (cpp:604-631)."""
    rgb = np.zeros((n, 3), np.uint8)
    count = np.zeros(n, np.int32)
    top_score = np.full((n, 5), -1.0, f32)
    top_rgb = np.zeros((n, 5), np.uint32)
    top_frame = np.full((n, 5), -1, np.int32)
    for i, lst in enumerate(lists):
        count[i] = len(lst)
        if not lst:
            continue
        lst = sorted(lst, key=lambda e: -e[0])[:5]  # stable: ties keep arrival order (B8)
        tot = f32(0)
        acc = [f32(0), f32(0), f32(0)]
        for k, (s, r, g, b, fr) in enumerate(lst):
            s = f32(s)
            acc[0] = acc[0] + f32(r) * s
            acc[1] = acc[1] + f32(g) * s
            acc[2] = acc[2] + f32(b) * s
            tot = tot + s
            top_score[i, k] = s
            top_rgb[i, k] = (r << 16) | (g << 8) | b
            top_frame[i, k] = fr
        for c in range(3):
            rgb[i, c] = np.uint8(int(acc[c] / tot))
    has = (rgb != 0).any(axis=1).astype(np.uint8)
    return dict(rgb=rgb, has=has, count=count, top_score=top_score, top_rgb=top_rgb, top_frame=top_frame)


# --------------------------------------------------------------------------- f4
def hsv_round_trip(bgr, saturation_scale: float = 1.0, brightness_scale: float = 1.0):
    """generateColorMap's image adjustment (PointCloudProcessor.cpp:722-741): 8-bit BGR2HSV (OpenCV 4.2 RGB2HSV_b,
    integer), S / V scaling through saturate_cast<uchar>, HSV2BGR (HSV2RGB_native, scalar fp32, no FMA)
    [upstream OpenCV 4.2.0, color_hsv.simd.hpp].  Vectorised over an (..., 3) uint8 array."""
    a = np.asarray(bgr, np.uint8)
    shape = a.shape
    a = a.reshape(-1, 3).astype(np.int64)
    b, g, r = a[:, 0], a[:, 1], a[:, 2]
    idx = np.arange(1, 256, dtype=f64)
    sdiv = np.concatenate([[0], np.rint((255 << 12) / (1.0 * idx)).astype(np.int64)])
    hdiv = np.concatenate([[0], np.rint((180 << 12) / (6.0 * idx)).astype(np.int64)])
    v = np.maximum(np.maximum(b, g), r)
    diff = v - np.minimum(np.minimum(b, g), r)
    s = (diff * sdiv[v] + 2048) >> 12
    h = np.where(v == r, g - b, np.where(v == g, b - r + 2 * diff, r - g + 4 * diff))
    h = (h * hdiv[diff] + 2048) >> 12  # arithmetic shift of a negative product = floor, as in C
    h = np.where(h < 0, h + 180, h)
    H = np.clip(h, 0, 255)

    def sat(x):  # saturate_cast<uchar>(float): round half to even, clamp
        return np.clip(np.rint(x.astype(f32)).astype(np.int64), 0, 255)

    S = sat(s.astype(f32) * f32(saturation_scale))
    V = sat(v.astype(f32) * f32(brightness_scale))
    inv255 = f32(1.0) / f32(255.0)
    fs = S.astype(f32) * inv255
    fv = V.astype(f32) * inv255
    fh = H.astype(f32) * (f32(6.0) / f32(180.0))
    fh = np.fmod(fh, f32(6.0)).astype(f32)
    sector = np.floor(fh).astype(np.int64)
    fh = fh - sector.astype(f32)
    bad = (sector < 0) | (sector >= 6)
    sector = np.where(bad, 0, sector)
    fh = np.where(bad, f32(0), fh).astype(f32)
    one = f32(1.0)
    tab = np.stack([fv, fv * (one - fs), fv * (one - fs * fh), fv * (one - fs * (one - fh))], axis=1).astype(f32)
    sector_data = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])
    pick = sector_data[sector]
    rows = np.arange(len(tab))
    out = np.stack([tab[rows, pick[:, c]] for c in range(3)], axis=1)
    grey = fs == 0
    out[grey] = fv[grey, None]
    return sat(out * f32(255.0)).astype(np.uint8).reshape(shape)


# --------------------------------------------------------------------------- A7
def _eigen33_smallest(C):
    """Smallest eigenpair of a symmetric 3x3 (any accurate solver, Appendix A7.3)."""
    w, V = np.linalg.eigh(C)
    return w[0], V[:, 0]


def mls_results(x, y, z, radius: float = 0.03, order: int = 2):
    """Per input point the MLSResult PCL caches (mean, normal, u, v, c_vec, curvature, K, fitted) or None when
    the point has fewer than 3 neighbours (MovingLeastSquares::performProcessing skips it) [upstream];
    brute-force neighbours (small n only)."""
    x, y, z = (np.asarray(a, f32) for a in (x, y, z))
    n = len(x)
    P = np.stack([x, y, z], axis=1)
    sq_r = f32(radius * radius)
    nr_coeff = (order + 1) * (order + 2) // 2
    res = []
    for i in range(n):
        d = P - P[i]
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        nn = np.nonzero(d2 < sq_r)[0]
        K = len(nn)
        if K < 3:
            res.append(None)
            continue
        nb = P[nn].astype(f64)
        c = nb.sum(axis=0) / K
        dm = nb - c
        C = dm.T @ dm
        ev, nrm = _eigen33_smallest(C)
        q = P[i].astype(f64)
        dist = q @ nrm - c @ nrm
        mean = q - dist * nrm
        tr = np.trace(C)
        curv = abs(ev / tr) if tr != 0 else 0.0
        if not (abs(nrm[0]) <= abs(nrm[2]) * 1e-12) or not (abs(nrm[1]) <= abs(nrm[2]) * 1e-12):
            inv = 1.0 / np.sqrt(nrm[0] ** 2 + nrm[1] ** 2)
            v = np.array([-nrm[1] * inv, nrm[0] * inv, 0.0])
        else:
            inv = 1.0 / np.sqrt(nrm[1] ** 2 + nrm[2] ** 2)
            v = np.array([0.0, -nrm[2] * inv, nrm[1] * inv])
        u = np.cross(nrm, v)
        cvec, fitted = np.zeros(nr_coeff), False
        if order > 1 and K >= nr_coeff:
            de = nb - mean
            w = np.exp(-(de * de).sum(axis=1) / (radius * radius))
            uc, vc, fz = de @ u, de @ v, de @ nrm
            rows = []
            for ui in range(order + 1):
                for vi in range(order - ui + 1):
                    rows.append(uc ** ui * vc ** vi)
            Pm = np.stack(rows, axis=0)
            A = (Pm * w) @ Pm.T
            b = (Pm * w) @ fz
            try:
                L = np.linalg.cholesky(A)
                cvec = np.linalg.solve(L.T, np.linalg.solve(L, b))
            except np.linalg.LinAlgError:
                cvec = np.full(nr_coeff, np.nan)
            fitted = True
        res.append(dict(query=q, mean=mean, normal=nrm, u=u, v=v, c_vec=cvec, curvature=curv, K=K, fitted=fitted))
    return res


def _mls_project(r, p, order: int, required: int):
    """MLSResult::projectPoint(p, SIMPLE, required) [upstream]: polynomial height and its gradient at the (u, v) of p
    when the fit exists and the point had `required` neighbours, else the plane."""
    de = p - r["mean"]
    u, v = de @ r["u"], de @ r["v"]
    w, nrm = 0.0, r["normal"].copy()
    if order > 1 and r["K"] >= required and r["fitted"] and np.isfinite(r["c_vec"][0]):
        dz = dzu = dzv = 0.0
        j = 0
        for ui in range(order + 1):
            for vi in range(order - ui + 1):
                c = r["c_vec"][j]
                dz += u ** ui * v ** vi * c
                if ui >= 1:
                    dzu += c * ui * u ** (ui - 1) * v ** vi
                if vi >= 1:
                    dzv += c * vi * u ** ui * v ** (vi - 1)
                j += 1
        w = dz
        nrm = nrm - (dzu * r["u"] + dzv * r["v"])
        l = np.linalg.norm(nrm)
        if l > 0:
            nrm = nrm / l
    return r["mean"] + u * r["u"] + v * r["v"] + w * r["normal"], nrm


def mls(x, y, z, radius: float = 0.03, order: int = 2):
    """PCL MovingLeastSquares, NONE upsampling, SIMPLE projection of the query point itself [upstream]."""
    nr_coeff = (order + 1) * (order + 2) // 2
    out_xyz, out_n, out_c, out_i = [], [], [], []
    for i, r in enumerate(mls_results(x, y, z, radius, order)):
        if r is None:
            continue
        pt, nn_out = r["mean"], r["normal"]
        if r["fitted"] and np.isfinite(r["c_vec"][0]):  # projectQueryPoint: u = v = 0
            pt = r["mean"] + r["c_vec"][0] * r["normal"]
            nn_out = r["normal"] - r["c_vec"][order + 1] * r["u"] - r["c_vec"][1] * r["v"]
            nn_out = nn_out / np.linalg.norm(nn_out)
        out_xyz.append(pt)
        out_n.append(nn_out)
        out_c.append(r["curvature"])
        out_i.append(i)
    return dict(xyz=np.array(out_xyz, f64).reshape(-1, 3).astype(f32),
                normal=np.array(out_n, f64).reshape(-1, 3).astype(f32),
                curvature=np.array(out_c, f64).astype(f32), index=np.array(out_i, np.int32))


def mls_voxel_dilation(x, y, z, radius: float = 0.03, order: int = 2, voxel: float = 0.001, iterations: int = 4):
    """performUpsampling(VOXEL_GRID_DILATION) [upstream mls.hpp MLSVoxelGrid]: voxelise at `voxel` from the
    bounding-box minimum, dilate the 26-neighbourhood `iterations` times (negative indices are not created,
    SURVEY.md B16), and for every voxel in ascending key order project its position onto the polynomial of the
    nearest input point (needs 5 * nr_coeff neighbours, else onto its plane)."""
    x, y, z = (np.asarray(a, f32) for a in (x, y, z))
    P = np.stack([x, y, z], axis=1)
    res = mls_results(x, y, z, radius, order)
    bmin, bmax = P.min(axis=0), P.max(axis=0)
    vs = f32(voxel)
    S = int(1.5 * float(np.max(bmax - bmin)) / float(vs))
    cells = set()
    for i in range(len(x)):
        ix, iy, iz = (int(np.trunc((P[i, k] - bmin[k]) / vs)) for k in range(3))  # fp32 arithmetic, C truncation
        cells.add((ix, iy, iz))
    for _ in range(iterations):
        grown = set()
        for (ix, iy, iz) in cells:
            for dx in (-1, 0, 1):
                for dy in (-1, 0, 1):
                    for dz in (-1, 0, 1):
                        if ix + dx >= 0 and iy + dy >= 0 and iz + dz >= 0:
                            grown.add((ix + dx, iy + dy, iz + dz))
        cells = grown
    nr_coeff = (order + 1) * (order + 2) // 2
    out_xyz, out_n, out_c, out_i = [], [], [], []
    for (ix, iy, iz) in sorted(cells, key=lambda c: (c[0] * S + c[1]) * S + c[2]):
        p = np.array([f32(ix) * vs + bmin[0], f32(iy) * vs + bmin[1], f32(iz) * vs + bmin[2]], f32)
        d = P - p
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]  # fp32 L2_Simple
        best = int(np.argmin(d2))  # first minimum = lowest index
        r = res[best]
        if r is None:
            continue
        pt, nrm = _mls_project(r, p.astype(f64), order, 5 * nr_coeff)
        out_xyz.append(pt)
        out_n.append(nrm)
        out_c.append(r["curvature"])
        out_i.append(best)
    return dict(xyz=np.array(out_xyz, f64).reshape(-1, 3).astype(f32),
                normal=np.array(out_n, f64).reshape(-1, 3).astype(f32),
                curvature=np.array(out_c, f64).astype(f32), index=np.array(out_i, np.int32))


def sor(x, y, z, mean_k: int = 60, std_mul: float = 0.7):
    """pcl::StatisticalOutlierRemoval [upstream statistical_outlier_removal.hpp]: mean distance to the mean_k nearest
    neighbours (fp32 squared distances, the query itself is hit 0), then keep iff distance <= mean + mul * stddev.
    Brute force (small n only).  Returns (keep, distances fp32, threshold)."""
    x, y, z = (np.asarray(a, f32) for a in (x, y, z))
    n = len(x)
    P = np.stack([x, y, z], axis=1)
    dist = np.zeros(n, f32)
    for i in range(n):
        d = P - P[i]
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        nn = np.sort(d2)[1:mean_k + 1]
        dist[i] = f32(np.sqrt(nn).astype(f64).sum() / mean_k)  # sqrt(float) summed in double
    s = dist.astype(f64).sum()
    sq = (dist * dist).astype(f64).sum()  # fp32 squares, double accumulation
    mean = s / n
    var = (sq - s * s / n) / (n - 1)
    thr = mean + std_mul * np.sqrt(var)
    return (dist.astype(f64) <= thr).astype(np.uint8), dist, thr


# --------------------------------------------------------------------------- f1
_BSPLINE = np.array([[1.0, -3.0, 3.0, -1.0], [4.0, 0.0, -6.0, 3.0], [1.0, 3.0, 3.0, -3.0], [0.0, 0.0, 0.0, 1.0]]) / 6.0


def nid_cost(cam: dict, image, x, y, z, intensity, T, bins: int = 16):
    """NIDCost::operator() value for one keyframe (PCP/include/vlcal/costs/nid_cost.hpp:42-116): x, y, z, intensity =
    the culled cloud, T = T_camera_lidar (4x4), image BGR8 read with the reference's channel accident (element x of
    the interleaved row, visual_camera_calibration.cpp:171-173)."""
    P = np.stack([np.asarray(a, f32).astype(f64) for a in (x, y, z)], axis=1)
    T = np.asarray(T, f64).reshape(4, 4)
    pc = P @ T[:3, :3].T + T[:3, 3]
    u, v = project(cam, pc[:, 0], pc[:, 1], pc[:, 2])
    W, H = cam["image_width"], cam["image_height"]
    bin_pts = np.clip((np.asarray(intensity, f32).astype(f64) * bins).astype(np.int64), 0, bins - 1)
    ok = np.isfinite(u) & np.isfinite(v) & (np.abs(u) <= 1e9) & (np.abs(v) <= 1e9)
    kx, ky = np.floor(np.where(ok, u, -1.0)).astype(np.int64), np.floor(np.where(ok, v, -1.0)).astype(np.int64)
    ok &= (kx >= 0) & (ky >= 0) & (kx < W) & (ky < H)
    u, v, kx, ky, bin_pts = u[ok], v[ok], kx[ok], ky[ok], bin_pts[ok]
    hist_points = np.bincount(bin_pts, minlength=bins).astype(f64)
    su, sv = u - kx, v - ky
    bu = _BSPLINE @ np.stack([np.ones_like(su), su, su ** 2, su ** 3])  # (4, m)
    bv = _BSPLINE @ np.stack([np.ones_like(sv), sv, sv ** 2, sv ** 3])
    flat = np.asarray(image, np.uint8).reshape(H, W * 3)  # the CV_64FC3 row as the reference indexes it
    hist = np.zeros((bins, bins))
    for a in range(4):
        px = np.clip(kx - 1 + a, 0, W - 1)
        for b in range(4):
            py = np.clip(ky - 1 + b, 0, H - 1)
            pix = flat[py, px].astype(f64) / 255.0
            bin_img = np.minimum((pix * bins).astype(np.int64), bins - 1)
            np.add.at(hist, (bin_img, bin_pts), bu[a] * bv[b])
    total = hist_points.sum()
    h_img = hist.sum(axis=1) / total
    h_pts = hist_points / total
    h_ip = hist / total
    H_image = -(h_img * np.log(h_img + 1e-6)).sum()
    H_points = -(h_pts * np.log(h_pts + 1e-6)).sum()
    H_ip = -(h_ip * np.log(h_ip + 1e-6)).sum()
    MI = H_image + H_points - H_ip
    return (H_ip - MI) / H_ip


def hpr_frame(cam: dict, w2c, x, y, z, flip_radius: float = 90000.0):
    """The ACTIVE reference cull (Katz HPR, view_culling.cpp:266-334) through
    scipy's bundled qhull_r -- CPU-only documented alternative (Appendix B1)."""
    from scipy.spatial import ConvexHull

    p = project_frame(cam, w2c, x, y, z)
    u, v = p["u"], p["v"]
    ok = (p["zc"] > 0) & _trunc_ok(u) & _trunc_ok(v)
    ui = np.where(ok, np.trunc(np.where(ok, u, 0)), -1)
    vi = np.where(ok, np.trunc(np.where(ok, v, 0)), -1)
    cand = ok & (ui >= 0) & (ui < cam["cull_width"]) & (vi >= 0) & (vi < cam["cull_height"])
    idx = np.nonzero(cand)[0]
    if len(idx) < 3:
        # qhull needs dim + 1 = 4 points (three candidates and the origin): it fails on fewer and the reference then
        # returns no visible point (view_culling.cpp:307-312)
        return idx[:0]
    pts = np.stack([p["xc"][idx], p["yc"][idx], p["zc"][idx]], axis=1).astype(f64)
    X, Y, Z = pts[:, 0], pts[:, 1], pts[:, 2]
    nrm = np.sqrt((X * X + Y * Y) + Z * Z)[:, None]  # Eigen's norm of a 3-vector block, as the z-buffer's range
    flipped = pts + (2.0 * (flip_radius - nrm) * pts) / nrm  # per coefficient x + ((2 (R - norm)) x) / norm
    try:
        hull = ConvexHull(np.concatenate([flipped, np.zeros((1, 3))], axis=0))
    except Exception:  # QhullError: flat input -> qh_new_qhull returns non-zero -> no visible point (:307-312)
        return idx[:0]
    vis = np.array(sorted(int(k) for k in hull.vertices if k < len(idx)), dtype=np.int64)
    return idx[vis]
